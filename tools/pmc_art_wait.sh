#!/bin/bash
# Where the articulated-gripper kernel's cycles go (run on the GPU box: gpurun -- 'bash tools/pmc_art_wait.sh <tag> <lib.so> ...'):
# SQ wait / activity counters per instruction class for tools/art_bench.py, counter passes only, one pass per group.
set -o pipefail
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  export MJS_LIB=$GRAFT_REPO_ROOT/$lib
  k=0
  for counters in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
                  "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY" \
                  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU"; do
    k=$((k+1))
    rocprofv3 --pmc $counters -d $out/${name}_$k -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/art_bench.py 4096 > $out/${name}_$k.log 2>&1 || echo "pmc $name $k failed"
  done
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$out/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "bg" in r["Kernel_Name"] and "Lb0" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f.split("/")[-2], {k: round(sum(v) / len(v)) for k, v in acc.items()}, "launches", {k: len(v) for k, v in acc.items()}.popitem()[1] if acc else 0)
PY
