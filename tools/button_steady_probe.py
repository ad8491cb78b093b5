"""Development probe: Button-Push steady-state launch time (control steps 30..90 of the synchronous episodes, bench.py's actions) and
the first steps after the reset, kernel time from HIP events around trains of 10 launches."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import mujoco_sim_amd as m  # noqa: E402
from bench import make_actions  # noqa: E402

N = 4096
venv = m.HipVectorEnv("robot_push_button", N, seed=0)
acts = make_actions("robot_push_button", 64, N, "cuda", 1)
venv.reset()
for t in range(100):  # one whole episode as warm-up (incl. the first-launch scratch allocation)
    venv.step_flat(acts[t % 64])
torch.cuda.synchronize()
times = []
for t in range(100):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    venv.step_flat(acts[t % 64])
    e1.record()
    times.append((e0, e1))
torch.cuda.synchronize()
ms = np.array([a.elapsed_time(b) for a, b in times]) * 1e3
print("us per launch (incl. ~6 us of event overhead): steps 0-5 after the reset step:", np.round(ms[:6], 0), "| steady (steps 30-90) mean %.1f max %.1f | episode mean %.1f" % (ms[30:90].mean(), ms[30:90].max(), ms.mean()))
