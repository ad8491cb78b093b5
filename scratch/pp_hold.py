import sys, numpy as np, torch
sys.path.insert(0, ".")
import mujoco_sim_amd as m
N = 4096
venv = m.HipVectorEnv("robot_planar_push", N, seed=2025)
venv.reset(); torch.cuda.synchronize()
hold = venv.flat_obs[:, :2].clone().contiguous()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for a, b in evs:
    a.record(); venv.step_flat(hold); b.record()
torch.cuda.synchronize()
print("hold ms", np.round([a.elapsed_time(b) for a, b in evs], 2))
venv.close()
