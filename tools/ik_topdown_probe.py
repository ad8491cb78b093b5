#!/usr/bin/env python3
"""Top-down TCP -> joints on the device (mjs_ur5e_tcp_to_joints: the closed form every task's EEF action goes through) against
the oracle's exhaustive closest-of-8 search, on targets FAR outside any task's action box and guesses anywhere in the joint
ranges (run on the GPU box). Prints the disagreeing cases."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import mujoco_sim_amd as m  # noqa: E402
from oracle import oracle as om  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rs = np.random.RandomState(3)
pos = rs.uniform([-0.9, -1.0, -0.3], [0.9, 0.4, 0.8], (n, 3))
venv = m.HipVectorEnv("robot_reach", 256, seed=1)
obs, _ = venv.reset()
q_reset = venv.flat_obs.cpu().numpy()[:, 3:9]
guess = q_reset[rs.randint(0, 256, n)] + rs.normal(0, 0.2, (n, 6))
guess[n // 2:] = rs.uniform(-3.1, 3.1, (n - n // 2, 6))
q, ok = venv.tcp_to_joints(pos, guess)
q, ok = q.cpu().numpy(), ok.cpu().numpy()
tcp_z = 0.174
R = np.array([[1.0, 0, 0], [0, -1, 0], [0, 0, -1]])  # quaternion (x, y, z, w) = (1, 0, 0, 0)
bad = []
for i in range(n):
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = pos[i] - R[:, 2] * tcp_z
    qo = om.ur5e_ik_closest(T, guess[i])
    if qo is None:
        if ok[i]:
            bad.append((i, "device found one, oracle none"))
    elif not ok[i]:
        bad.append((i, "oracle found one, device none"))
    elif np.abs(q[i] - qo).max() > 1e-7:
        bad.append((i, f"differ by {np.abs(q[i] - qo).max():.3g}: device {np.round(q[i], 4)} oracle {np.round(qo, 4)} |dev-guess| {np.linalg.norm(q[i] - guess[i]):.6f} |ora-guess| {np.linalg.norm(qo - guess[i]):.6f}"))
print(f"{n} cases, {int(ok.sum())} solvable on the device, {len(bad)} disagreements")
for i, why in bad[:25]:
    print(i, np.round(pos[i], 4), np.round(guess[i], 3), why)
