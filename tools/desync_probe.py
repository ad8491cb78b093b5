#!/usr/bin/env python3
"""Development probe (GPU box): what episodes that end at DIFFERENT times cost. bench.py's episodes are synchronous (all envs reset in the same
launch); here the time row of every env is shifted by a random phase after the first reset, so that ~1 % of the envs end in every control step,
and the mean launch time of the following 200 steps is compared with the synchronous run's. Usage: desync_probe.py [task ...]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import mujoco_sim_amd as m  # noqa: E402
from bench import make_actions  # noqa: E402

N = 4096
TIME_ROW = 12  # S_TIME of the three robot scenes
for task in sys.argv[1:] or ["robot_reach", "robot_push_button", "robot_planar_push"]:
    # variants: Robot-Reach / Button-Push 3 = reset workgroups; Planar-Push 1 = round 3's behaviour (settle steps inside the step launch), 0 = prefetched episodes
    for desync, variant in ((False, 0), (True, 0)) + (((False, 3), (True, 3)) if task in ("robot_reach", "robot_push_button") else ((False, 1), (True, 1))):
        kw = {"max_episode_steps": 100} if task == "robot_planar_push" else {}
        venv = m.HipVectorEnv(task, N, seed=0, kernel_variant=variant, **kw)
        acts = make_actions(task, 64, N, "cuda", 1)
        venv.reset()
        if desync:
            st = venv.get_state()
            phase = torch.from_numpy(np.floor(np.random.RandomState(5).uniform(0, 100, N))).to(st.device)
            if task == "robot_planar_push":
                st[16] = phase  # S_STEP: the step counter ends the episode there (max_episode_steps)
            else:
                st[TIME_ROW] = phase * 0.1
            venv.set_state(st)
        T0, T = (120, 200) if task != "robot_planar_push" else (110, 100)
        for t in range(T0):
            venv.step_flat(acts[t % 64])
        torch.cuda.synchronize()
        ev = []
        for t in range(T):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); venv.step_flat(acts[t % 64]); b.record()
            ev.append((a, b))
        torch.cuda.synchronize()
        us = np.array([a.elapsed_time(b) for a, b in ev]) * 1e3
        ends = int((venv._buf["step_type"] == 2).sum())
        print(f"{task:20s} variant {variant} {'episode ends spread over the steps' if desync else 'synchronous episodes            '}: mean {us.mean():8.1f} us per launch (median {np.median(us):8.1f}, max {us.max():8.1f}); "
              f"envs ending in the last step: {ends}", flush=True)
        venv.close()
