#!/bin/bash
# HBM traffic of the articulated-gripper kernel for one or more diagnostic libraries (gpurun -- 'bash tools/pmc_art_traffic.sh <tag> <lib.so> ...'):
# FETCH_SIZE and WRITE_SIZE in separate passes of bench.py's articulated workload; prints the medians per launch (FETCH corrected by 0.5039).
set -o pipefail
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  export MJS_LIB=$GRAFT_REPO_ROOT/$lib
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $out/${name}_$c -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --task robot_push_button --gripper-model articulated --steps 30 --warmup 5 --no-cpu-baseline > $out/${name}_$c.log 2>&1 || echo "pmc $name $c failed"
  done
done
python3 - <<PY
import csv, glob, statistics
for f in sorted(glob.glob("$out/*/*counter_collection.csv")):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("void bg::kernel<false>")]
    name = f.split("/")[-2]
    k = 1024 / 0.5039 if "FETCH" in name else 1024
    print(name, "median %.1f MB per launch over %d launches" % (statistics.median(v) * k / 1e6, len(v)))
PY
