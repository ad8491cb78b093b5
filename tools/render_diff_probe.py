"""Development probe (GPU box): share of image bytes that differ between the HIP cameras and the CPU restatement (oracle/om_render.c), per scene,
camera and size. (Planar-Push with mesh blocks: the probe does not copy the per-episode block draws to the oracle, its figures there mean nothing.)"""
import sys
import numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import mujoco_sim_amd as m
import oracle
oracle.build()
def report(name, venv, ob, cams):
    for cam in cams:
        for res in (64, 96):
            g = venv.render(res, res, camera=cam).cpu().numpy().astype(np.int16)
            c = ob.render(res, res, camera=cam).astype(np.int16)
            d = np.abs(g - c)
            print(f"{name} camera {cam} {res}x{res}: differing bytes {(d > 0).mean():.2e}, max {d.max()}")
N = 64
venv = m.HipVectorEnv("robot_push_button", N, seed=11); ob = oracle.OracleBatch(oracle.TASK_BUTTON_PUSH, N, 11, nthreads=8)
venv.reset(); ob.reset()
rs = np.random.RandomState(1)
home = np.array([-0.5, -0.5, 0.5, -0.5, -0.5, -0.5]) * np.pi
for t in range(8):
    a = np.concatenate([home + rs.uniform(-0.3, 0.3, (N, 6)), rs.uniform(0, 0.085, (N, 1))], axis=1)
    venv.step(torch.from_numpy(a)); ob.step(a)
try:
    report("Button-Push", venv, ob, (0, 1))
except TypeError as e:
    print("render signature:", e)

for task, tid, kw, okw in (("robot_reach", oracle.TASK_ROBOT_REACH, {}, {}), ("robot_planar_push", oracle.TASK_PLANAR_PUSH, {"block_shape": "box"}, {"block_shape": 1}), ("robot_planar_push", oracle.TASK_PLANAR_PUSH, {}, {})):
    venv = m.HipVectorEnv(task, 32, seed=5, **kw); ob = oracle.OracleBatch(tid, 32, 5, nthreads=8, **okw)
    venv.reset(); ob.reset()
    lo, hi = np.asarray(venv.spec.action_low), np.asarray(venv.spec.action_high)
    for t in range(4):
        a = rs.uniform(lo, hi, (32, len(lo))); venv.step(torch.from_numpy(a)); ob.step(a)
    for res in (64, 96, 128):
        g = venv.render(res, res).cpu().numpy().astype(np.int16); c = ob.render(res, res).astype(np.int16); d = np.abs(g - c)
        print(f"{task} {kw} {res}x{res}: differing bytes {(d > 0).mean():.2e}, max {d.max()}")
