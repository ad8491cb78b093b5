#!/usr/bin/env python3
"""Large-batch parity soak (development tool, run on the GPU box): the GPU tests' comparisons at batch sizes and horizons the
test suite does not afford, on the inputs that reach the rarely taken paths. Every env against the oracle, every step; an env
leaves a comparison when the oracle reports a bad state, the device reports bit 1 / 8 / 16, or (Robot-Reach) a command's closest
IK solution is an exact tie. Prints one summary line per scenario; exit code 1 if any scenario saw a divergence."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import mujoco_sim_amd as m  # noqa: E402
import oracle  # noqa: E402
from test_gpu_parity import _gpu_result, _top_down_ik_is_a_tie  # noqa: E402

oracle.build()
NT = 16
import os  # noqa: E402
SEED_SHIFT = int(os.environ.get("SOAK_SEED", "0"))  # other seeds for the envs and the action streams
bad_total = 0


def run(name, task, tid, N, T, actions, seed, tie_check=False, venv_kw=None, ob_kw=None, atol=1e-7, obs_sink=None):
    global bad_total
    t0 = time.time()
    seed += 1000 * SEED_SHIFT
    venv = m.HipVectorEnv(task, N, seed=seed, **(venv_kw or {}))
    ob = oracle.OracleBatch(tid, N, seed, nthreads=NT, **(ob_kw or {}))
    venv.reset()
    o = ob.reset()
    if obs_sink is not None:
        obs_sink["obs"] = o["obs"]
    g = _gpu_result(venv)
    assert np.abs(g["obs"] - o["obs"]).max() < 1e-9, "reset"
    alive = np.ones(N, bool)
    n_div = n_rows = n_tie = n_guard = n_last = 0
    worst = 0.0
    for t in range(T):
        a = actions(t)
        if tie_check:
            q_now = o["obs"][:, 3:9]
            for i in np.nonzero(alive)[0]:
                if _top_down_ik_is_a_tie(oracle, a[i], q_now[i]):
                    alive[i] = False
                    n_tie += 1
        venv.step(torch.from_numpy(a))
        o = ob.step(a)
        if obs_sink is not None:
            obs_sink["obs"] = o["obs"]
        g = _gpu_result(venv)
        n_guard += int(((g["fault"] & (8 | 16)) > 0)[alive].sum())
        alive &= ~o["fault"] & ~(g["fault"] & (1 | 8 | 16)).astype(bool) & (np.abs(o["obs"]).max(axis=1) < 50)
        d = np.abs(g["obs"] - o["obs"]).max(axis=1)
        badenv = np.nonzero(alive & ~(d <= atol))[0]
        if badenv.size:
            print(f"  {name} step {t}: {badenv.size} env(s) beyond {atol}: {badenv[:8]} max {d[badenv].max():.3g} fault {g['fault'][badenv[:8]]} ncon {g['ncon'][badenv[:8]]} / {o['ncon'][badenv[:8]]}")
        worst = max(worst, float(d[alive & (d <= atol)].max(initial=0.0)))
        n_div += badenv.size
        alive[badenv] = False
        for k in ("step_type", "terminated", "truncated", "is_success", "ncon"):
            mism = np.nonzero(alive & (np.asarray(g[k]).astype(np.int64) != np.asarray(o[k]).astype(np.int64)))[0]
            if mism.size:
                print(f"  {name} step {t}: {k} differs for env(s) {mism[:8]}")
                n_div += mism.size
                alive[mism] = False
        n_rows += int(((g["fault"] & 4) > 0)[alive].sum())
        n_last += int((np.asarray(g["step_type"]) == 2).sum())
    venv.close()
    bad_total += n_div
    print(f"{name}: {N} envs x {T} steps, alive {alive.mean():.4f}, divergences {n_div}, rows env-steps {n_rows}, guard reports {n_guard}, ik ties {n_tie}, "
          f"episode ends {n_last}, worst accepted |d obs| {worst:.2e}, {time.time() - t0:.0f} s", flush=True)


rs = np.random.RandomState(123 + SEED_SHIFT)
which = sys.argv[1:] or ["reach_box", "reach_success", "reach_wild", "button_eef", "button_joint_full", "button_joint_nominal", "button_articulated", "planar_push", "pointmass"]
if "reach_box" in which:
    run("Robot-Reach, workspace actions, 3 episodes", "robot_reach", oracle.TASK_ROBOT_REACH, 4096, 250, lambda t: rs.uniform([-0.1, -0.6, 0.02], [0.1, -0.4, 0.2], (4096, 3)), 11)
if "reach_success" in which:
    # episodes that end at different times (terminate_on_success): three quarters of the envs servo to their target; the host picks the reset-groups kernel
    state = {}
    def seek(t, N=4096):
        a = rs.uniform([-0.1, -0.6, 0.02], [0.1, -0.4, 0.2], (N, 3))
        if "obs" in state:
            k = np.arange(N) % 4 != 0
            a[k] = state["obs"][k, 9:12] + rs.normal(0, 0.002, (int(k.sum()), 3))
        return a
    run("Robot-Reach, success-terminated episodes (reset groups)", "robot_reach", oracle.TASK_ROBOT_REACH, 4096, 300, seek, 18,
        venv_kw=dict(terminate_on_success=True), ob_kw=dict(terminate_on_success=True), obs_sink=state)
if "reach_wild" in which:
    def wild(t, N=2048):
        a = rs.uniform([-0.6, -0.9, -0.15], [0.6, 0.1, 0.5], (N, 3))
        a[: N // 4] = rs.uniform([-0.1, -0.6, 0.02], [0.1, -0.4, 0.2], (N // 4, 3))
        return a
    run("Robot-Reach, actions far outside the box", "robot_reach", oracle.TASK_ROBOT_REACH, 2048, 120, wild, 12, tie_check=True)
if "button_eef" in which:
    run("Button-Push, EEF actions, same-step reset", "robot_push_button", oracle.TASK_BUTTON_PUSH, 2048, 220,
        lambda t: rs.uniform([-0.2, -0.6, 0.02, 0.0], [0.2, -0.3, 0.3, 0.085], (2048, 4)), 13,
        venv_kw=dict(autoreset="same_step", action_type="absolute_eef_action"), ob_kw=dict(autoreset=1, action_type=1))
if "button_joint_full" in which:
    run("Button-Push, full-range joint actions", "robot_push_button", oracle.TASK_BUTTON_PUSH, 2048, 110,
        lambda t: np.concatenate([rs.uniform(-3.14, 3.14, (2048, 6)), rs.uniform(0, 0.085, (2048, 1))], axis=1), 14)
if "button_joint_nominal" in which:
    nominal = np.array([-1.57, -1.57, 1.57, -1.57, -1.57, 0.0, 0.04])
    run("Button-Push, joint actions around the nominal pose", "robot_push_button", oracle.TASK_BUTTON_PUSH, 2048, 220,
        lambda t: nominal + rs.uniform(-1, 1, (2048, 7)) * np.array([0.6, 0.4, 0.4, 0.4, 0.4, 0.6, 0.04]), 15)
if "button_articulated" in which:
    # the articulated 2F-85 (mjs_gripper14.h: three wavefronts per env group): EEF actions over the workspace with the fingers opening and
    # closing (pads on the switch / the floor where the targets go low), same-step resets; joint actions around the nominal pose
    run("Button-Push, articulated gripper, EEF actions, same-step reset", "robot_push_button", oracle.TASK_BUTTON_PUSH, 1024, 110,
        lambda t: rs.uniform([-0.2, -0.6, 0.05, 0.0], [0.2, -0.3, 0.3, 0.085], (1024, 4)), 21,
        venv_kw=dict(autoreset="same_step", action_type="absolute_eef_action", gripper_model="articulated"), ob_kw=dict(autoreset=1, action_type=1, gripper_model=1), atol=1e-6)
    nominal14 = np.array([-1.57, -1.57, 1.57, -1.57, -1.57, 0.0, 0.04])
    run("Button-Push, articulated gripper, joint actions around the nominal pose", "robot_push_button", oracle.TASK_BUTTON_PUSH, 1024, 110,
        lambda t: nominal14 + rs.uniform(-1, 1, (1024, 7)) * np.array([0.4, 0.3, 0.3, 0.3, 0.3, 0.6, 0.04]), 22,
        venv_kw=dict(gripper_model="articulated"), ob_kw=dict(gripper_model=1), atol=1e-6)
def run_push(name, shape, N=2048, LIMIT=40, T=85, seed=31):
    """Planar-Push with the conditioning rule of tests/test_gpu_parity.py::_check_ill_conditioned_envs: a second oracle whose resets are perturbed by
    1e-13 m measures every env's sensitivity; calm envs (< 1e-10) are held to 1e-6 (worst value reported) and exact flags / contact counts, the others to the episode
    bookkeeping exactly, no bad state, and an error of at most max(1e-6, 100 x their own sensitivity)."""
    global bad_total
    import ctypes as C
    t0 = time.time()
    knob = C.c_double.in_dll(oracle.lib(), "om_dbg_perturb")
    venv = m.HipVectorEnv("robot_planar_push", N, seed=seed + SEED_SHIFT, max_episode_steps=LIMIT, block_shape=shape)
    okw = dict(nthreads=16, max_episode_steps=LIMIT, block_shape=1 if shape == "box" else 0)
    ob, ob2 = oracle.OracleBatch(oracle.TASK_PLANAR_PUSH, N, seed + SEED_SHIFT, **okw), oracle.OracleBatch(oracle.TASK_PLANAR_PUSH, N, seed + SEED_SHIFT, **okw)
    venv.reset()
    o = ob.reset()
    knob.value = 1e-13
    o2 = ob2.reset()
    knob.value = 0.0
    sens = np.abs(o["obs"] - o2["obs"]).max(axis=1) > 1e-10
    prs = np.random.RandomState(9 + SEED_SHIFT)
    n_div = n_last = 0
    worst_calm = 0.0
    frac = []
    for t in range(T):
        tcp, blk = o["obs"][:, :2], o["obs"][:, 5:7]
        a = tcp + np.clip(blk - tcp, -0.02, 0.02) + prs.uniform(-0.004, 0.004, (N, 2))
        venv.step(torch.from_numpy(a))
        o = ob.step(a)
        knob.value = 1e-13
        o2 = ob2.step(a)
        knob.value = 0.0
        dev = np.abs(o["obs"] - o2["obs"]).max(axis=1)
        fresh = (o["step_type"] != 1) & (o2["step_type"] != 1)
        sens = np.where(fresh, dev > 1e-12, sens | (dev > 1e-10))
        g = _gpu_result(venv)
        err = np.abs(g["obs"] - o["obs"]).max(axis=1)
        calm_bad = ~sens & (err > 1e-6)  # (the solvers stop at MuJoCo's tolerance 1e-8 on the scaled gradient: calm envs differ by up to 1e-7)
        for k in ("step_type", "terminated", "truncated", "is_success", "ncon"):
            calm_bad |= ~sens & (np.asarray(g[k]).astype(np.int64) != np.asarray(o[k]).astype(np.int64))
        same = np.asarray(o["step_type"]) == np.asarray(o2["step_type"])
        wild_bad = sens & same & (err > np.maximum(1e-6, 100 * dev))
        for k in ("step_type", "terminated", "truncated"):
            wild_bad |= sens & (np.asarray(g[k]).astype(np.int64) != np.asarray(o[k]).astype(np.int64))
        wild_bad |= sens & ((np.asarray(g["fault"]).astype(np.int64) & 1) > 0)
        if calm_bad.any() or wild_bad.any():
            print(f"  {name} step {t}: calm envs off {np.nonzero(calm_bad)[0][:8]} (|d obs| {err[calm_bad][:8]}, own sensitivity {dev[calm_bad][:8]}), "
                  f"ill-conditioned envs beyond their bound {np.nonzero(wild_bad)[0][:8]}")
        n_div += int(calm_bad.sum() + wild_bad.sum())
        worst_calm = max(worst_calm, float(err[calm_bad].max(initial=0.0)))
        sens = sens | calm_bad  # reported once: the env sits out the rest of its episode like an ill-conditioned one
        n_last += int((np.asarray(g["step_type"]) == 2).sum())
        frac.append(sens.mean())
    venv.close()
    bad_total += n_div
    print(f"{name}: {N} envs x {T} steps, env-episodes off their bound {n_div} (worst |d obs| of a calm env {worst_calm:.2e}), ill-conditioned fraction mean {np.mean(frac):.4f} "
          f"max {np.max(frac):.4f}, episode ends {n_last}, {time.time() - t0:.0f} s", flush=True)


if "planar_push" in which:
    run_push("Planar-Push, mesh blocks, 2 episodes", "mesh")
    run_push("Planar-Push, box blocks, 2 episodes", "box", seed=33)
if "pointmass" in which:
    run("Pointmass-Reach, 4 episodes", "point_mass_reach", oracle.TASK_POINTMASS, 4096, 420, lambda t: rs.uniform(-0.1, 0.1, (4096, 2)).astype(np.float32).astype(np.float64), 16, atol=1e-9)
print("soak:", "OK" if bad_total == 0 else f"{bad_total} divergence(s)")
sys.exit(1 if bad_total else 0)
