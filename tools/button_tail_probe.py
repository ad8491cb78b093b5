"""Development probe: what happens in the first control steps after a Button-Push reset under bench.py's action distribution
(joint targets q_home +- U(0.2))? Prints, per step, how many envs have contacts, joints beyond their range, fault bits, and
the kernel time."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import mujoco_sim_amd as m  # noqa: E402
from bench import make_actions  # noqa: E402

N = 4096
venv = m.HipVectorEnv("robot_push_button", N, seed=0)
acts = make_actions("robot_push_button", 24, N, "cuda", 1)
venv.reset()
for t in range(24):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = venv.step(acts[t])
    e1.record()
    torch.cuda.synchronize()
    st = venv.get_state().cpu().numpy()
    q, v = st[0:6], st[6:12]
    info = out[-1]
    ncon, fault = info["ncon"].cpu().numpy(), info["fault"].cpu().numpy()
    print(t, "ms %.3f" % e0.elapsed_time(e1), "ncon>0:", None if ncon is None else int((ncon > 0).sum()),
          "fault bits:", None if fault is None else {b: int(((fault & b) != 0).sum()) for b in (1, 2, 4, 8, 16)},
          "|q|max per joint:", np.round(np.abs(q).max(axis=1), 2), "|v| mean:", np.round(np.abs(v).mean(), 2))
