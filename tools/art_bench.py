"""Launch time of the articulated-gripper Button-Push kernel (mjs_gripper14.h) at a few batch sizes (holding pose, gripper half open)."""
import sys, time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import mujoco_sim_amd as m

for n in [int(x) for x in (sys.argv[1:] or ["4096"])]:
    v = m.HipVectorEnv("robot_push_button", n, seed=5, action_type="absolute_joint_action", gripper_model="articulated")
    v.reset()
    a = torch.zeros(n, 7, dtype=torch.float64, device="cuda")
    a[:, :6] = v._buf["obs"][:, :6]
    a[:, 6] = 0.04
    for k in range(3):
        v.step(a)
    torch.cuda.synchronize()
    ts = []
    for k in range(30):
        t0 = time.time()
        v.step(a)
        torch.cuda.synchronize()
        ts.append(time.time() - t0)
    ts.sort()
    dt = ts[len(ts) // 2]
    print(f"envs {n}: median {dt * 1e3:.2f} ms per step (min {ts[0] * 1e3:.2f}, max {ts[-1] * 1e3:.2f}) = {n / dt / 1e6:.3f} M env-steps/s", flush=True)
    v.close()
