#!/usr/bin/env python3
"""Calibration workload for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters in THIS library's
access pattern (8 B per lane, coalesced, struct-of-arrays): mjs_get_state on a handle large enough
to exceed the 256 MiB Infinity Cache copies a KNOWN number of bytes
(read 16*8*N + N, written 17*8*N). Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
and compare with the printed byte counts (MI355X_MICROARCH.md, HBM section)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import mujoco_sim_amd as m  # noqa: E402

N = 1 << 21
venv = m.HipVectorEnv("robot_reach", N, seed=1)
for _ in range(5):
    s = venv.get_state()
torch.cuda.synchronize()
print(f"calibration: get_state_kernel N={N}: read {16 * 8 * N + N} B, written {17 * 8 * N} B per launch")
venv.close()
