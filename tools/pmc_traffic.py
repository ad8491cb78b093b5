"""Collate two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the MI355X guide prescribes) of
`python bench.py --task T ...` into profiles/r01_traffic_all_tasks.json: median over the step-kernel launches, counter
unit KB, FETCH_SIZE / 0.5039 (gfx950 calibration, profiles/r01_traffic.json + tools/pmc_calibrate.py), WRITE_SIZE exact.

usage: python tools/pmc_traffic.py TASK KERNEL_SUBSTRING ENVS ALGORITHMIC_BYTES_PER_ENV_STEP fetch.csv write.csv
"""
import csv
import json
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
FETCH_CALIBRATION = 0.5039


def median_counter(path, kernel, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
    assert vals, (path, kernel, counter)
    return statistics.median(vals), len(vals)


def main():
    task, kernel, envs, alg, fetch_csv, write_csv = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
    f, nf = median_counter(fetch_csv, kernel, "FETCH_SIZE")
    w, nw = median_counter(write_csv, kernel, "WRITE_SIZE")
    path = ROOT / "profiles" / "r01_traffic_all_tasks.json"
    d = json.load(open(path))
    read, write = f * 1024 / FETCH_CALIBRATION, w * 1024
    d["tasks"][task] = {"kernel": kernel, "envs": envs, "launches": [nf, nw], "raw_KB": {"FETCH_SIZE": f, "WRITE_SIZE": w},
                        "corrected_bytes_per_launch": {"read": read, "write": write, "total": read + write},
                        "algorithmic_bytes_per_launch": alg * envs}
    json.dump(d, open(path, "w"), indent=1)
    print(task, "read %.3f MB write %.3f MB per launch (algorithmic %.3f MB)" % (read / 1e6, write / 1e6, alg * envs / 1e6))


if __name__ == "__main__":
    main()
