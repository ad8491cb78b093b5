import sys, time, torch
sys.path.insert(0, "/root/repo")
import mujoco_sim_amd as m
import numpy as np
for task, A in (("robot_reach", 3), ("point_mass_reach", 2)):
    venv = m.HipVectorEnv(task, 4096, seed=1)
    venv.reset()
    lo, hi = np.asarray(venv.spec.action_low), np.asarray(venv.spec.action_high)
    a = torch.from_numpy(np.random.RandomState(0).uniform(lo, hi, (4096, A))).cuda()
    for _ in range(200): venv.step_flat(a)
    torch.cuda.synchronize()
    # enqueue rate with a tiny env count (GPU work negligible -> host-bound)
    small = m.HipVectorEnv(task, 64, seed=1); small.reset(); a64 = a[:64].contiguous()
    for _ in range(200): small.step_flat(a64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5000): small.step_flat(a64)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(task, "host cost per step_flat call: %.2f us (enqueue loop), %.2f us incl. drain" % ((t1 - t0) / 5000 * 1e6, (t2 - t0) / 5000 * 1e6))
