#!/bin/bash
# Planar-Push HBM-side traffic (run on the GPU box): FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes.
# usage: bash tools/pmc_push_traffic.sh <out-dir> [bench args...]
set -o pipefail
out=$1; shift
mkdir -p $out
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $out/$c -o pmc --output-format csv -- python3 bench.py --task robot_planar_push --steps 30 --warmup 10 --no-cpu-baseline "$@" > $out/$c.json 2> $out/$c.log || echo "pmc $c failed"
done
python3 - "$out" <<'PY'
import csv, sys, collections
out = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f"{out}/{c}/pmc_counter_collection.csv")) if "kernelILb0" in r["Kernel_Name"] or "kernel<false>" in r["Kernel_Name"]]
    v = v[len(v) // 4:]
    res[c] = sum(v) / max(1, len(v))
# FETCH_SIZE / WRITE_SIZE are in KB on this profiler; gfx950 correction of the guide: FETCH_SIZE reads 0.5039 of the bytes
fetch_mb, write_mb = res["FETCH_SIZE"] / 0.5039 / 1024, res["WRITE_SIZE"] / 1024
print(f"per launch: fetch {fetch_mb:.1f} MB (corrected), write {write_mb:.1f} MB, total {fetch_mb + write_mb:.1f} MB")
PY
