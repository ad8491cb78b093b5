#!/bin/bash
# Instruction-cache view of a kernel (run on the GPU box: gpurun -- 'bash tools/pmc_icache.sh <tag>'): SQC_ICACHE_* and the SQ issue
# counters for the articulated-gripper kernel (tools/art_bench.py) and, as a control, for the Robot-Reach bench. Counter passes
# only (no trace domains), separate passes per counter group; CSVs land under gpurun_out/<tag>/.
set -o pipefail
tag=${1:-r4pmc}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters, program...
  local name=$1 counters=$2; shift 2
  rocprofv3 --pmc $counters -d $out/$name -o pmc --output-format csv -- "$@" > $out/$name.log 2>&1 || echo "pmc $name failed"
}
run art_icache "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" python3 $GRAFT_REPO_ROOT/tools/art_bench.py 4096
run art_issue "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_IFETCH" python3 $GRAFT_REPO_ROOT/tools/art_bench.py 4096


find $out -name "*counter_collection.csv" | head
