#!/usr/bin/env python3
"""GPU probe (development tool): arm links on the floor, HIP robust path vs the CPU oracle.
  reach : Robot-Reach, shoulder-lift offsets injected with mjs_set_state so that links start near / in the floor,
          all three kernel variants; prints per-step max |obs| error, ncon agreement, fault words
  button: Button-Push with uniform full-range joint actions (the registered action space), alive statistics"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import mujoco_sim_amd as m  # noqa: E402
import oracle  # noqa: E402


def gpu(venv):
    b = venv._buf
    return {k: b[k].cpu().numpy().copy() for k in ("obs", "reward", "step_type", "ncon", "fault")}


def inject(venv, q, v, s_warm):
    gs = venv.get_state().clone()
    gs[0:6] = torch.from_numpy(q.T)
    gs[6:12] = torch.from_numpy(v.T)
    gs[s_warm:s_warm + 6] = 0.0                      # the oracle's debug setter zeroes qacc_warmstart
    gs[-1] = torch.from_numpy((gs[-1].cpu().numpy().astype(np.uint8) | 16).astype(np.float64))  # FLAG_WARM_VALID
    venv.set_state(gs)


def reach(T=8):
    N = 64
    for variant in (0, 1, 2):
        venv = m.HipVectorEnv("robot_reach", N, seed=5, kernel_variant=variant)
        ob = oracle.OracleBatch(oracle.TASK_ROBOT_REACH, N, 5, nthreads=8)
        venv.reset()
        o = ob.reset()
        q = o["obs"][:, 3:9].copy()
        tcp0 = o["obs"][:, 0:3].copy()
        q[:, 1] += np.linspace(-0.2, 1.6, N)
        v = np.zeros((N, 6))
        ob.set_robot_state(q, v)
        inject(venv, q, v, 16)
        worst = np.zeros(N)
        for t in range(T):
            a = tcp0
            venv.step(torch.from_numpy(a))
            o = ob.step(a)
            g = gpu(venv)
            err = np.abs(g["obs"] - o["obs"]).max(axis=1)
            worst = np.maximum(worst, err)
            print(f"variant {variant} step {t}: max err {err.max():.3e} (env {err.argmax()}), ncon equal {np.array_equal(g['ncon'], o['ncon'])}, "
                  f"ncon>0 envs {(o['ncon'] > 0).sum()}, gpu fault words {sorted(set(g['fault'].tolist()))}, "
                  f"oracle bad {int(o['fault'].sum())} arm_floor {int(ob.arm_floor_seen().sum())}")
            if not np.array_equal(g["ncon"], o["ncon"]):
                bad = np.nonzero(g["ncon"] != o["ncon"])[0][:8]
                print("   ncon mismatch envs", bad, g["ncon"][bad], o["ncon"][bad])
        print(f"variant {variant}: envs with err > 1e-7: {np.nonzero(worst > 1e-7)[0].tolist()}")
        venv.close()


def button(T=12, N=512):
    venv = m.HipVectorEnv("robot_push_button", N, seed=41, autoreset="disabled")
    ob = oracle.OracleBatch(oracle.TASK_BUTTON_PUSH, N, 41, autoreset=2, nthreads=8)
    rs = np.random.RandomState(17)
    venv.reset()
    ob.reset()
    alive = np.ones(N, bool)
    for t in range(T):
        a = np.concatenate([rs.uniform(-3.14, 3.14, (N, 6)), rs.uniform(0, 0.085, (N, 1))], axis=1)
        venv.step(torch.from_numpy(a))
        o = ob.step(a)
        g = gpu(venv)
        err = np.abs(g["obs"] - o["obs"]).max(axis=1)
        seen = ob.arm_floor_seen()
        drop = alive & ((err > 1e-7) | o["fault"].astype(bool) | ((g["fault"] & (1 | 8 | 16)) > 0))
        print(f"step {t}: alive {alive.mean():.3f} arm_floor envs {int(seen.sum())} err>1e-7 among alive {int((alive & (err > 1e-7)).sum())} "
              f"(of which arm-floor {int((alive & (err > 1e-7) & seen).sum())}) gpu fault bits among alive: "
              f"1:{int(((g['fault'] & 1) > 0)[alive].sum())} 4:{int(((g['fault'] & 4) > 0)[alive].sum())} 8:{int(((g['fault'] & 8) > 0)[alive].sum())} 16:{int(((g['fault'] & 16) > 0)[alive].sum())} "
              f"oracle bad {int(o['fault'][alive].sum())} ncon equal among alive {np.array_equal(g['ncon'][alive & ~drop], o['ncon'][alive & ~drop])} median err {np.median(err[alive]):.2e}")
        alive &= ~drop
    print("alive at end", alive.mean())


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "detail"):
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    if which in ("reach", "both"):
        reach()
    if which in ("button", "both"):
        button()


def button_detail(T=12, N=512):
    """per-env detail of every env-step whose observation differs from the oracle's by > 1e-7"""
    np.set_printoptions(precision=5, linewidth=220, suppress=True)
    venv = m.HipVectorEnv("robot_push_button", N, seed=41, autoreset="disabled")
    ob = oracle.OracleBatch(oracle.TASK_BUTTON_PUSH, N, 41, autoreset=2, nthreads=8)
    rs = np.random.RandomState(17)
    venv.reset()
    o = ob.reset()
    alive = np.ones(N, bool)
    prev = o["obs"].copy()
    for t in range(T):
        a = np.concatenate([rs.uniform(-3.14, 3.14, (N, 6)), rs.uniform(0, 0.085, (N, 1))], axis=1)
        venv.step(torch.from_numpy(a))
        o = ob.step(a)
        g = gpu(venv)
        seen = ob.arm_floor_seen()
        err = np.abs(g["obs"] - o["obs"]).max(axis=1)
        for i in np.nonzero(alive & (err > 1e-7))[0]:
            print(f"--- step {t} env {i}: err {err[i]:.3e} gpu fault {g['fault'][i]} ncon gpu/oracle {g['ncon'][i]}/{o['ncon'][i]} oracle bad {o['fault'][i]} arm_floor_seen {seen[i]}")
            print("   q before ", prev[i, :6], " tcp before", prev[i, 6:9], " switch", prev[i, 9:12])
            print("   action   ", a[i])
            print("   q gpu    ", g["obs"][i, :6], " tcp", g["obs"][i, 6:9])
            print("   q oracle ", o["obs"][i, :6], " tcp", o["obs"][i, 6:9])
        alive &= ~((err > 1e-7) | o["fault"].astype(bool) | ((g["fault"] & (1 | 8 | 16)) > 0))
        prev = o["obs"].copy()


def reach_detail():
    np.set_printoptions(precision=5, linewidth=220, suppress=True)
    N = 64
    venv = m.HipVectorEnv("robot_reach", N, seed=5)
    ob = oracle.OracleBatch(oracle.TASK_ROBOT_REACH, N, 5, nthreads=8)
    venv.reset()
    o = ob.reset()
    q = o["obs"][:, 3:9].copy()
    tcp0 = o["obs"][:, 0:3].copy()
    q[:, 1] += np.linspace(-0.2, 1.6, N)
    v = np.zeros((N, 6))
    ob.set_robot_state(q, v)
    inject(venv, q, v, 16)
    venv.step(torch.from_numpy(tcp0))
    o = ob.step(tcp0)
    g = gpu(venv)
    err = np.abs(g["obs"] - o["obs"]).max(axis=1)
    for i in range(N):
        print(i, "offset %.3f" % np.linspace(-0.2, 1.6, N)[i], "fault", g["fault"][i], "err %.2e" % err[i], "q", g["obs"][i, 3:9])


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "detail":
    reach_detail()
    button_detail()
