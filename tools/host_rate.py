#!/usr/bin/env python3
"""Host-side cost of the two batched front ends (development tool, run on the GPU box):
  * HipVectorEnv.step_flat: enqueue cost per call (zero-copy torch path)
  * HipSB3VecEnv.step_wait at 4096 envs: the whole SB3 surface per step (action upload, kernel, ONE device->host copy of the output
    arena, numpy views, per-env infos dicts) - VERDICT r2 item 9 asks for <= 1 ms."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import mujoco_sim_amd as m  # noqa: E402
from mujoco_sim_amd.sb3_vec_env import HipSB3VecEnv  # noqa: E402

for task, A in (("robot_reach", 3), ("point_mass_reach", 2)):
    small = m.HipVectorEnv(task, 64, seed=1)
    small.reset()
    lo, hi = np.asarray(small.spec.action_low), np.asarray(small.spec.action_high)
    a64 = torch.from_numpy(np.random.RandomState(0).uniform(lo, hi, (64, A))).cuda()
    for _ in range(200):
        small.step_flat(a64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5000):
        small.step_flat(a64)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(task, "HipVectorEnv.step_flat host cost: %.2f us per call (enqueue loop), %.2f us incl. drain" % ((t1 - t0) / 5000 * 1e6, (t2 - t0) / 5000 * 1e6))
    small.close()
for task in ("robot_reach", "robot_push_button"):
    N = 4096
    env = HipSB3VecEnv(task, N, seed=3)
    env.reset()
    from bench import make_actions  # the bench's action distribution (Button-Push: joint targets q_home +- 0.2, not the +-3.14 box)

    acts = make_actions(task, 64, N, "cpu", 1).numpy().astype(np.float32)
    for k in range(20):
        env.step(acts[k % 64])
    t0 = time.perf_counter()
    n = 300
    for k in range(n):
        env.step(acts[k % 64])
    dt = (time.perf_counter() - t0) / n
    print(f"{task} HipSB3VecEnv.step at {N} envs: {dt * 1e6:.0f} us per step (kernel + copies + numpy + {N} infos dicts) = {N / dt / 1e6:.2f} M env-steps/s through the SB3 surface")
    env.close()
