"""Development soak (GPU box): Button-Push parity with the oracle over more seeds, envs and steps than the test suite runs
(both action types, aimed presses, random gripper commands; obs / reward / flags / ncon at the tests' tolerances and the
gripper's driver state at 1e-10). Usage: python tools/parity_soak.py [n_seeds]. Prints one line per run; exits 1 on a mismatch."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import mujoco_sim_amd as m  # noqa: E402
import oracle as om  # noqa: E402
import test_gpu_parity as T  # noqa: E402


def run(seed, action_type, autoreset, N=256, steps=150):
    venv = m.HipVectorEnv("robot_push_button", N, seed=seed, autoreset=autoreset, action_type=action_type)
    ob = om.OracleBatch(om.TASK_BUTTON_PUSH, N, seed, autoreset={"next_step": 0, "same_step": 1}[autoreset], nthreads=8,
                        action_type={"absolute_joint_action": 0, "absolute_eef_action": 1}[action_type])
    acts = T._button_actions(action_type, steps, N, seed=seed + 7)
    venv.reset()
    o = ob.reset()
    rs = np.random.RandomState(seed + 1)
    n_contact = n_active = n_rows = 0
    for t in range(steps):
        a = acts[t].copy()
        if action_type == "absolute_eef_action":
            aim = rs.uniform(size=N) < 0.5
            a[aim, :3] = o["obs"][aim, 9:12] + rs.uniform([-0.03, -0.03, -0.04], [0.03, 0.03, 0.05], (N, 3))[aim]
        venv.step(torch.from_numpy(a))
        o = ob.step(a)
        g = T._gpu_result(venv)
        T._compare(t, g, o)
        np.testing.assert_allclose(venv.get_state().cpu().numpy()[16:18].T, ob.get_gripper(), rtol=0, atol=1e-10, err_msg=f"gripper step {t}")
        n_contact += int((o["ncon"] > 0).sum())
        n_active += int((o["obs"][:, 12] > 0.5).sum())
        n_rows += int(((g["fault"] & 4) != 0).sum())
        assert not (g["fault"] & (8 | 16)).any(), ("unsupported contact / fast-path violation", t)
    venv.close()
    print(f"seed {seed} {action_type} {autoreset}: ok ({n_contact} env-steps with contacts, {n_rows} with rows, {n_active} with an active switch)", flush=True)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    for s in range(n):
        for at in ("absolute_eef_action", "absolute_joint_action"):
            run(1000 + 37 * s, at, "next_step" if s % 2 == 0 else "same_step")
    print("soak passed")
