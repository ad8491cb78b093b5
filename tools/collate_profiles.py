#!/usr/bin/env python3
"""Copies the summaries of a tools/profile_round.sh run (gpurun_out/<tag>/) into profiles/ (tracked) and derives the two
JSON files bench.py reads: profiles/<round>_traffic_all_tasks.json (HBM bytes per step launch from the FETCH_SIZE /
WRITE_SIZE passes, FETCH_SIZE corrected by the gfx950 calibration 0.5039 of profiles/r01_traffic.json) and
profiles/<round>_reach_valu.json (VALU busy share of a wavefront's lifetime for the default Robot-Reach launch).

usage: python tools/collate_profiles.py r4p r04
"""
import csv
import json
import shutil
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
FETCH_CALIBRATION = 0.5039
STEP_KERNELS = {"robot_reach": "rr::kernel3", "point_mass_reach": "pm::kernel<false>", "robot_push_button": "bp::kernel<false", "robot_planar_push": "pp::kernel<false>",
                "robot_push_button_articulated": "bg::kernel<false>"}
ALG_BYTES = {"robot_reach": 571, "point_mass_reach": 267, "robot_push_button": 643, "robot_planar_push": 843, "robot_push_button_articulated": 1123}  # mjs_algorithmic_bytes_per_env_step (abi 2)


def find(d, suffix):
    hits = sorted(Path(d).rglob(f"*{suffix}"))
    return hits[0] if hits else None


def counter_rows(path, kernel, counter):
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]


def main():
    tag, rnd = sys.argv[1], sys.argv[2]
    src, dst = ROOT / "gpurun_out" / tag, ROOT / "profiles"
    for d in sorted(src.glob("kt_*")):
        if d.is_dir():
            f = find(d, "kernel_stats.csv")
            if f:
                shutil.copy(f, dst / f"{rnd}_final_{d.name[3:]}_kernel_stats.csv")
    lines = []
    for j in sorted(src.glob("kt_*.json")):
        txt = [l for l in j.read_text().splitlines() if l.startswith("{")]
        if txt:
            lines.append(json.dumps({"run": j.stem[3:], **json.loads(txt[-1])}))
    (dst / f"{rnd}_final_bench_lines.jsonl").write_text("\n".join(lines) + "\n")
    traffic = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `python bench.py --task T --steps 100 --warmup 10 "
                         "--no-cpu-baseline` (Planar-Push --steps 30), median over the step launches, counter unit KB; FETCH_SIZE / 0.5039 (gfx950 "
                         "calibration of profiles/r01_traffic.json, tools/pmc_calibrate.py), WRITE_SIZE exact", "tasks": {}}
    for task, kern in STEP_KERNELS.items():
        fd, wd = src / f"pmc_{task}_FETCH_SIZE", src / f"pmc_{task}_WRITE_SIZE"
        ff, wf = find(fd, "counter_collection.csv") if fd.exists() else None, find(wd, "counter_collection.csv") if wd.exists() else None
        if not ff or not wf:
            continue
        shutil.copy(ff, dst / f"{rnd}_pmc_fetch_{task}.csv")
        shutil.copy(wf, dst / f"{rnd}_pmc_write_{task}.csv")
        fr, wr = counter_rows(ff, kern, "FETCH_SIZE"), counter_rows(wf, kern, "WRITE_SIZE")
        if not fr or not wr:
            continue
        f, w = statistics.median(fr), statistics.median(wr)
        read, write = f * 1024 / FETCH_CALIBRATION, w * 1024
        traffic["tasks"][task] = {"kernel": kern, "envs": 4096, "launches": [len(fr), len(wr)], "raw_KB": {"FETCH_SIZE": f, "WRITE_SIZE": w},
                                  "corrected_bytes_per_launch": {"read": read, "write": write, "total": read + write},
                                  "algorithmic_bytes_per_launch": ALG_BYTES[task] * 4096}
        print(task, "read %.3f MB write %.3f MB per launch (algorithmic %.3f MB)" % (read / 1e6, write / 1e6, ALG_BYTES[task] * 4096 / 1e6))
    json.dump(traffic, open(dst / f"{rnd}_traffic_all_tasks.json", "w"), indent=1)
    # bench.py replays the COMMITTED traffic figure in its line; the lines of this very run were printed before the figure existed (or
    # with the previous round's): put this run's figure in, so that the jsonl is self-consistent
    run_task = {"reach": "robot_reach", "reach_driver": "robot_reach", "pointmass": "point_mass_reach", "button": "robot_push_button",
                "push": "robot_planar_push", "button_articulated": "robot_push_button_articulated"}
    fixed = []
    for l in (dst / f"{rnd}_final_bench_lines.jsonl").read_text().splitlines():
        d = json.loads(l)
        t = traffic["tasks"].get(run_task.get(d["run"], ""))
        if t and d.get("roofline"):
            d["roofline"]["traffic"] = t["corrected_bytes_per_launch"]["total"]
        fixed.append(json.dumps(d))
    (dst / f"{rnd}_final_bench_lines.jsonl").write_text("\n".join(fixed) + "\n")
    vd = src / "pmc_reach_valu"
    vf = find(vd, "counter_collection.csv") if vd.exists() else None
    if vf:
        shutil.copy(vf, dst / f"{rnd}_reach_pmc_valu.csv")
        med = {c: statistics.median(counter_rows(vf, "rr::kernel3", c)) for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES")}
        waves = med["SQ_WAVES"]
        out = {"method": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -- python bench.py --steps 100 "
                         f"--warmup 10 --no-cpu-baseline; medians over the rr::kernel3 launches (profiles/{rnd}_reach_pmc_valu.csv)",
               "envs": 4096, "waves": waves, "valu_insts_per_wave": med["SQ_INSTS_VALU"] / waves, "wave_quad_cycles": med["SQ_WAVE_CYCLES"] / waves,
               "valu_busy_quad_cycles_per_wave": med["SQ_ACTIVE_INST_VALU"] / waves,
               "valu_busy_frac_of_wave_lifetime": med["SQ_ACTIVE_INST_VALU"] / med["SQ_WAVE_CYCLES"],
               "clocks_per_valu_inst": 4 * med["SQ_ACTIVE_INST_VALU"] / med["SQ_INSTS_VALU"]}
        json.dump(out, open(dst / f"{rnd}_reach_valu.json", "w"), indent=1)
        print("reach VALU:", out)


if __name__ == "__main__":
    main()
