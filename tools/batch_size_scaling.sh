#!/bin/bash
out=gpurun_out/${1:-r3k}; mkdir -p $out
run() { python bench.py "$@" --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'].split(':')[0], d['config']['envs_per_gpu'], round(d['value']/1e6,3), round(d['ms_per_step']*1000,1))"; }
for n in 4096 16384 65536 262144; do run --envs-per-gpu $n --steps 300 --warmup 30; done
for n in 4096 65536 262144; do run --task point_mass_reach --envs-per-gpu $n --steps 300 --warmup 30; done
for n in 4096 16384 65536; do run --task robot_push_button --envs-per-gpu $n --steps 300 --warmup 130; done
for n in 4096 16384 65536; do run --task robot_planar_push --envs-per-gpu $n --steps 60 --warmup 10; done
for n in 4096 16384; do run --task robot_planar_push --block-shape box --envs-per-gpu $n --steps 60 --warmup 10; done
for n in 2048 8192; do run --task robot_push_button --envs-per-gpu $n --visual 64 --steps 100 --warmup 20; done
for n in 4096 16384 65536; do run --task robot_push_button --gripper-model articulated --envs-per-gpu $n --steps 60 --warmup 10; done
