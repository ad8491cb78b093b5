#!/bin/bash
# Round profiles (run on the GPU box: gpurun -- 'bash tools/profile_round.sh r3p'): rocprofv3 kernel-trace statistics of the
# default bench command and of the secondary workloads, and the PMC passes (FETCH_SIZE and WRITE_SIZE in SEPARATE passes, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes; the SQ counters in a third). Counter passes never combine with the
# hip / hsa / memory-copy trace domains. Outputs land under gpurun_out/<tag>/; tools/collate_profiles.py copies the summaries
# into profiles/.
set -o pipefail
tag=${1:-r4p}
part=${2:-all}   # stats | pmc | art | reach | all (a gpurun call is limited to 20 minutes: run the two halves separately; art = the articulated-gripper lines only)
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
stats() {  # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -d $out/kt_$name -o $name --output-format csv -- python3 bench.py "$@" > $out/kt_$name.json 2> $out/kt_$name.log || echo "kernel-trace $name failed"
}
pmc() {  # name, counters (quoted), bench args...
  local name=$1 counters=$2; shift 2
  rocprofv3 --pmc $counters -d $out/pmc_$name -o pmc --output-format csv -- python3 bench.py "$@" > $out/pmc_$name.json 2> $out/pmc_$name.log || echo "pmc $name failed"
}
if [ "$part" == "reach" ]; then
stats reach --steps 2000 --warmup 200
stats reach_driver --steps 20 --warmup 5 --no-cpu-baseline
pmc robot_reach_FETCH_SIZE FETCH_SIZE --task robot_reach --steps 100 --warmup 10 --no-cpu-baseline
pmc robot_reach_WRITE_SIZE WRITE_SIZE --task robot_reach --steps 100 --warmup 10 --no-cpu-baseline
pmc reach_valu "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" --steps 100 --warmup 10 --no-cpu-baseline
exit 0
fi
if [ "$part" == "art" ]; then
stats button_articulated --task robot_push_button --gripper-model articulated --steps 30 --warmup 5
pmc robot_push_button_articulated_FETCH_SIZE FETCH_SIZE --task robot_push_button --gripper-model articulated --steps 30 --warmup 5 --no-cpu-baseline
pmc robot_push_button_articulated_WRITE_SIZE WRITE_SIZE --task robot_push_button --gripper-model articulated --steps 30 --warmup 5 --no-cpu-baseline
ls -R $out | head -20
exit 0
fi
if [ "$part" != "pmc" ]; then
stats reach --steps 2000 --warmup 200
stats reach_driver --steps 20 --warmup 5 --no-cpu-baseline
stats push --task robot_planar_push --steps 150 --warmup 30 --no-cpu-baseline
stats push_box --task robot_planar_push --block-shape box --steps 150 --warmup 30 --no-cpu-baseline
stats button --task robot_push_button --steps 500 --warmup 50 --no-cpu-baseline
stats button_visual --task robot_push_button --envs-per-gpu 2048 --visual 64 --steps 200 --warmup 20 --no-cpu-baseline
stats pointmass --task point_mass_reach --steps 1000 --warmup 100 --no-cpu-baseline
stats button_articulated --task robot_push_button --gripper-model articulated --steps 30 --warmup 5 --no-cpu-baseline
fi
if [ "$part" != "stats" ]; then
for task in robot_reach point_mass_reach robot_push_button; do
  pmc ${task}_FETCH_SIZE FETCH_SIZE --task $task --steps 100 --warmup 10 --no-cpu-baseline
  pmc ${task}_WRITE_SIZE WRITE_SIZE --task $task --steps 100 --warmup 10 --no-cpu-baseline
done
pmc robot_push_button_articulated_FETCH_SIZE FETCH_SIZE --task robot_push_button --gripper-model articulated --steps 30 --warmup 5 --no-cpu-baseline
pmc robot_push_button_articulated_WRITE_SIZE WRITE_SIZE --task robot_push_button --gripper-model articulated --steps 30 --warmup 5 --no-cpu-baseline
pmc robot_planar_push_FETCH_SIZE FETCH_SIZE --task robot_planar_push --steps 30 --warmup 10 --no-cpu-baseline
pmc robot_planar_push_WRITE_SIZE WRITE_SIZE --task robot_planar_push --steps 30 --warmup 10 --no-cpu-baseline
pmc reach_valu "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" --steps 100 --warmup 10 --no-cpu-baseline
fi
ls -R $out | head -80
