"""Generates tests/golden/*.npz from the CPU oracle (oracle/), which is the only executable
statement of the reference path available in this pipeline: the reference itself cannot be
imported (mujoco / dm_control / ur_analytic_ik are absent third-party wheels). The vectors are
therefore REGRESSION pins of the oracle + the numpy-RandomState-pinned reset draws, not outputs
of the reference ("parity unpinned").

Run:  python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402

OUT = Path(__file__).resolve().parent


def actions_for(task, T, N, seed=12345):
    rs = np.random.RandomState(seed)
    if task == oracle.TASK_POINTMASS:
        return rs.uniform(-0.05, 0.05, (T, N, 2)).astype(np.float32).astype(np.float64)
    return rs.uniform([-0.1, -0.6, 0.02], [0.1, -0.4, 0.2], (T, N, 3))


def run(task, N, T, base_seed, autoreset):
    b = oracle.OracleBatch(task, N, base_seed, autoreset=autoreset)
    acts = actions_for(task, T, N)
    r0 = b.reset()
    keys = ["obs", "reward", "discount", "step_type", "terminated", "truncated", "is_success", "ncon"]
    traj = {k: [] for k in keys}
    for t in range(T):
        r = b.step(acts[t])
        for k in keys:
            traj[k].append(r[k])
    return dict(actions=acts, reset_obs=r0["obs"], **{k: np.stack(v) for k, v in traj.items()})


if __name__ == "__main__":
    np.savez_compressed(OUT / "pointmass_n8_t70_seed2025.npz", **run(oracle.TASK_POINTMASS, 8, 70, 2025, oracle.AUTORESET_NEXT_STEP))
    np.savez_compressed(OUT / "robot_reach_n8_t110_seed2025.npz", **run(oracle.TASK_ROBOT_REACH, 8, 110, 2025, oracle.AUTORESET_NEXT_STEP))
    print("golden fixtures written to", OUT)
