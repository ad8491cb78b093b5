"""Development probe (diagnostic build -DMJS_STAMPS, MJS_LIB=mujoco_sim_amd/lib/libmjsim_stamps.so): phase cycles of the Button-Push step launch
K control steps after a reset under bench.py's actions (K = 5: robust path without a constraint stage; K = 12: row-free path). The library
prints the stamps of the LAST launch when the handle is destroyed."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import mujoco_sim_amd as m  # noqa: E402
from bench import make_actions  # noqa: E402

N = 4096
for K in [int(x) for x in (sys.argv[1:] or ["2", "5", "12"])]:
    venv = m.HipVectorEnv("robot_push_button", N, seed=0)
    acts = make_actions("robot_push_button", 24, N, "cuda", 1)
    venv.reset()
    for t in range(K):
        venv.step(acts[t])
    torch.cuda.synchronize()
    print(f"--- last launch = control step {K - 1} after the reset", file=sys.stderr, flush=True)
    venv.close()
