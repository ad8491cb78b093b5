#!/usr/bin/env python3
"""GPU microbenchmark (development tool): device time of the analytic IK entry points, one wavefront per 64 poses.
mjs_ur5e_tcp_to_joints = the top-down closed form every task uses; mjs_debug_ur5e_ik = the general closest-of-8 search."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from mujoco_sim_amd import _native as nat  # noqa: E402

L = nat.lib()
n = 64 * 64
rs = np.random.RandomState(0)
pos = torch.from_numpy(rs.uniform([-0.1, -0.6, 0.02], [0.1, -0.4, 0.2], (n, 3))).cuda()
guess = torch.from_numpy(np.tile(np.array([1.2, -1.0, 1.5, 1.1, 1.57, -0.3]), (n, 1)) + rs.uniform(-0.1, 0.1, (n, 6))).cuda()
q = torch.empty(n, 6, dtype=torch.float64, device="cuda")
ok = torch.empty(n, dtype=torch.uint8, device="cuda")
T = torch.zeros(n, 12, dtype=torch.float64, device="cuda")
T[:, 0] = 1; T[:, 4] = -1; T[:, 8] = -1
T[:, 9:12] = pos + torch.tensor([0, 0, 0.174], dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream


def timeit(f, reps=200):
    for _ in range(20):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


t_td = timeit(lambda: L.mjs_ur5e_tcp_to_joints(C.c_void_p(pos.data_ptr()), C.c_void_p(guess.data_ptr()), C.c_void_p(q.data_ptr()), C.c_void_p(ok.data_ptr()), n, C.c_void_p(stream)))
q1 = q.clone()
t_gen = timeit(lambda: L.mjs_debug_ur5e_ik(C.c_void_p(T.data_ptr()), C.c_void_p(guess.data_ptr()), C.c_void_p(q.data_ptr()), C.c_void_p(ok.data_ptr()), n, C.c_void_p(stream)))
print(f"top-down closed form {t_td:.2f} us per launch, general search {t_gen:.2f} us per launch ({n} poses, 64 wavefronts); max |dq| between them {float((q1 - q).abs().max()):.2e}")
