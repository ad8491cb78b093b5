"""Development check (GPU box): the rectangle-walk camera kernels against the 8x8-tile walk (kernel_variant 1), byte for byte,
over 1024 envs per task, twelve rounds of full-range random actions and a final batch of arbitrary joint states injected with
set_state (cameras at odd poses, arms through the floor). Prints the number of mismatching bytes (expected: 0)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import mujoco_sim_amd as m
tot = 0
for task in ("robot_push_button", "robot_reach", "robot_planar_push"):
    N = 1024
    a = m.HipVectorEnv(task, N, seed=5)
    b = m.HipVectorEnv(task, N, seed=5, kernel_variant=1)
    a.reset(); b.reset()
    rng = np.random.RandomState(9)
    cams = (0, 1) if task == "robot_push_button" else (0,)
    lo, hi = np.asarray(a.action_low, dtype=np.float64), np.asarray(a.action_high, dtype=np.float64)
    for rnd in range(12):
        for cam in cams:
            for hh, ww in ((64, 64), (32, 48), (96, 96)):
                ia, ib = a.render(hh, ww, camera=cam), b.render(hh, ww, camera=cam)
                nd = int((ia != ib).sum().item())
                tot += nd
                if nd:
                    bad = (ia != ib).flatten(1).any(1).nonzero().flatten()[:5].tolist()
                    print("MISMATCH", task, rnd, cam, hh, ww, nd, bad)
        for _ in range(5):
            act = torch.as_tensor(lo + rng.uniform(0, 1, (N, a.action_dim)) * (hi - lo), device="cuda")
            a.step(act); b.step(act)
    # arbitrary joint states through set_state: random joints in +-pi (cameras at odd poses, arm through the floor)
    st = a.get_state()
    st[0:6] = torch.as_tensor(rng.uniform(-3.1, 3.1, (6, N)), device="cuda")
    a.set_state(st); b.set_state(st)
    for cam in cams:
        ia, ib = a.render(64, 64, camera=cam), b.render(64, 64, camera=cam)
        nd = int((ia != ib).sum().item()); tot += nd
        print(task, "random joints cam", cam, "mismatching bytes", nd)
    a.close(); b.close()
print("total mismatching bytes", tot)
