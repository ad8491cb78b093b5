"""Pins for the CPU oracle (no GPU needed).

The physics of the reference lives in absent third-party wheels, so the oracle is pinned by
everything that CAN be checked here:
  * numpy's RandomState itself (the reference seeds with it, dmc2gym.py:126-131),
  * float64 time-limit crossings of the composer loop,
  * the reference's own test tolerances re-expressed against the oracle
    (test/test_ur_frame_matches_real.py:29, test/test_ur_control_api.py:26-28,80-82,
    test/test_gym_envs.py:21-36),
  * closed-form answers (weld spring response, servo interpolation),
  * committed golden vectors (regression).
"""
import ctypes as C

import numpy as np
import pytest

GOLDEN = __import__("pathlib").Path(__file__).parent / "golden"


# ---------------------------------------------------------------- RNG: pinned against numpy
@pytest.mark.parametrize("seed", [0, 1, 2024, 2025, 123456789, 2**32 - 1])
def test_rng_matches_numpy_randomstate(oracle_mod, seed):
    r = oracle_mod.OracleRng(seed)
    rs = np.random.RandomState(seed)
    # > 624 words so that the state regeneration (twist) is exercised twice
    ours = [r.uniform(-0.45, 0.45) for _ in range(700)]
    ref = [rs.uniform(-0.45, 0.45) for _ in range(700)]
    assert ours == ref


def test_reset_draws_known_answers(oracle_mod):
    # SURVEY.md App. A.6 (computed there with numpy only): goal_x, goal_y, point_x, point_y
    expect = {
        2025: (-0.3280606526898344, 0.34906653245734015, 0.3893450758978522, -0.048988652357164375),
        2024: (0.07921306700585812, 0.1791978729134242, -0.2806632359653446, -0.4105722926278217),
        0: (0.04393215353459229, 0.1936704297351775, 0.09248703846447953, 0.040394864697207156),
    }
    for seed, (gx, gy, px, py) in expect.items():
        b = oracle_mod.OracleBatch(oracle_mod.TASK_POINTMASS, 1, seed)
        obs = b.reset()["obs"][0]
        assert tuple(obs) == (px, py, gx, gy)
    # Robot-Reach: robot xyz then target xyz over the spawn box (robot_reach.py:143-150)
    b = oracle_mod.OracleBatch(oracle_mod.TASK_ROBOT_REACH, 1, 2025)
    obs = b.reset()["obs"][0]
    assert tuple(obs[9:12]) == (-0.010886367190480972, -0.5223528907772035, 0.06636735835420862)
    # TCP lands on the sampled robot position up to the DH-vs-model offset (~1 mm; reference tolerance 1e-2)
    assert np.allclose(obs[0:3], (-0.07290236726440764, -0.4224296594539244, 0.18786901517957044), atol=1e-2)
    assert not np.allclose(obs[0:3], (-0.07290236726440764, -0.4224296594539244, 0.18786901517957044), atol=1e-5)


# ------------------------------------------------------ composer loop: time-limit crossings
def test_time_limit_crossing_pointmass(oracle_mod):
    # time += 0.02 five times per step: 4.99999... after 50 steps -> truncation at control step 51
    b = oracle_mod.OracleBatch(oracle_mod.TASK_POINTMASS, 1, 3, reward_type=2)
    b.reset()
    # keep far from the goal so that only the time limit can end the episode
    steps = 0
    while True:
        r = b.step(np.zeros((1, 2)))
        steps += 1
        if r["step_type"][0] == 2:
            break
        assert steps < 60
    assert steps == 51 and r["truncated"][0] and not r["terminated"][0] and r["discount"][0] == 1.0
    # next step is the auto-reset: FIRST, action ignored
    r = b.step(np.full((1, 2), 0.05))
    assert r["step_type"][0] == 0 and not r["truncated"][0] and not r["terminated"][0]


def test_time_limit_crossing_robot(oracle_mod):
    b = oracle_mod.OracleBatch(oracle_mod.TASK_ROBOT_REACH, 1, 3)
    obs = b.reset()["obs"]
    a = obs[:, 0:3].copy()
    for k in range(1, 101):
        r = b.step(a)
        if k < 100:
            assert r["step_type"][0] == 1, k
    assert r["step_type"][0] == 2 and r["truncated"][0] and not r["terminated"][0]


# --------------------------------------------- reference test re-expressions (UR kinematics)


def test_sim_ur_frame_matches_real(oracle_mod):
    # test/test_ur_frame_matches_real.py: at q=0 the model's attachment_site == analytic FK (atol 1e-2).
    # The model chain (include/mjs_scene_spec.h) is evaluated here independently with numpy.
    import re

    spec = (GOLDEN.parents[1] / "include" / "mjs_scene_spec.h").read_text()

    def arr(name, shape):
        m = re.search(name + r"\[[^=]*=\s*\{(.*?)\};", spec, re.S)
        vals = [float(x) for x in re.findall(r"-?\d+\.?\d*(?:e-?\d+)?", re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S))]
        return np.array(vals).reshape(shape)

    pos, quat = arr("MJS_UR_BODY_POS", (7, 3)), arr("MJS_UR_BODY_QUAT", (7, 4))
    fpos, fquat = arr(r"MJS_UR_FLANGE_POS", (3,)), arr(r"MJS_UR_FLANGE_QUAT", (4,))

    def q2m(q):
        w, x, y, z = q / np.linalg.norm(q)
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    T = np.eye(4)
    for b in range(7):
        A = np.eye(4)
        A[:3, :3], A[:3, 3] = q2m(quat[b]), pos[b]
        T = T @ A  # joint angles are zero
    A = np.eye(4)
    A[:3, :3], A[:3, 3] = q2m(fquat), fpos
    T = T @ A
    FK = oracle_mod.ur5e_fk_dh(np.zeros(6))
    assert np.allclose(T, FK, atol=1e-2), f"{T}\nvs\n{FK}"
    # known answer of the real UR5e DH table at q = 0
    assert np.allclose(FK[:3, 3], [-0.8172, -0.2329, 0.0628], atol=1e-12)


def test_ik_round_trip_and_closest(oracle_mod):
    rs = np.random.RandomState(7)
    for _ in range(500):
        q = rs.uniform(-3.0, 3.0, 6)
        T = oracle_mod.ur5e_fk_dh(q)
        sols = oracle_mod.ur5e_ik_all(T)
        assert len(sols) >= 1
        for s in sols:  # every returned solution reproduces the pose
            assert np.abs(oracle_mod.ur5e_fk_dh(s) - T).max() < 1e-9
        qc = oracle_mod.ur5e_ik_closest(T, q)
        assert np.abs(qc - q).max() < 1e-6  # the generating configuration is its own closest solution
    # unreachable pose -> no solution (robot.py:119-120 returns None)
    T = np.eye(4)
    T[:3, 3] = [2.0, 0.0, 0.5]
    assert oracle_mod.ur5e_ik_closest(T, np.zeros(6)) is None


# ---- the reference's component tests of the Robot control API, re-expressed LITERALLY on the oracle: a bare UR5e (no
# arena, no end effector: TCP = flange), the XML's default timestep 0.002, the reference's start configuration, target,
# substep counts and atol = 1e-2 on every number the reference compares. Same three tests on the HIP path:
# tests/test_gpu_parity.py::test_reference_moveJ / _moveJ_IK / _servoL.
REF_START_JOINTS = np.array([0.0, -0.5, 0.5, -0.5, -0.5, -0.5]) * np.pi
REF_TARGET_POSE = np.array([0.1, 0.3, 0.5, 1, 0, 0, 0])


def _pose_close(pose, target, atol):
    # position + quaternion, the quaternion up to its double-cover sign: at a rotation of ~pi the sign spatialmath's log
    # map returns follows sub-tolerance residuals (oracle/om_robot_api.c)
    return np.allclose(pose[:3], target[:3], atol=atol) and (np.allclose(pose[3:], target[3:], atol=atol) or np.allclose(pose[3:], -target[3:], atol=atol))


def test_reference_moveJ(oracle_mod):
    """test/test_ur_control_api.py:7-28: from the default qpos (zeros, ctrl zeros) moveJ(joints, speed 1.0) and 10 000 x
    (before_substep; physics.step()) -> joints within atol 1e-2"""
    st, _, ok = oracle_mod.ur_robot_run(oracle_mod.ur_robot_state(np.zeros(6), ctrl=np.zeros(6)), REF_START_JOINTS, oracle_mod.UR_CMD_MOVEJ, 1.0, 10000)
    assert ok and np.allclose(st[0:6], REF_START_JOINTS, atol=1e-2), st[0:6]
    assert st[18] == pytest.approx(20.0, abs=1e-6) and st[19] == 1.0  # is_finished(physics.timestep()) never fires (robot.py:271)


def test_reference_moveJ_IK(oracle_mod):
    """test/test_ur_control_api.py:31-54: set_joint_positions(joints); movej_IK(target pose, 1.0); 6000 substeps ->
    get_tcp_pose within atol 1e-2 on all 7 numbers"""
    st, pose, ok = oracle_mod.ur_robot_run(oracle_mod.ur_robot_state(REF_START_JOINTS), REF_TARGET_POSE, oracle_mod.UR_CMD_MOVEJ_IK, 1.0, 6000)
    assert ok and _pose_close(pose, REF_TARGET_POSE, 1e-2), pose


def test_reference_servoL(oracle_mod):
    """test/test_ur_control_api.py:57-82: 20 x (servoL(target pose, 0.2) + 100 substeps) -> get_tcp_pose within atol 1e-2"""
    st = oracle_mod.ur_robot_state(REF_START_JOINTS)
    for _ in range(20):
        st, pose, ok = oracle_mod.ur_robot_run(st, REF_TARGET_POSE, oracle_mod.UR_CMD_SERVOL, 0.2, 100)
        assert ok
    assert _pose_close(pose, REF_TARGET_POSE, 1e-2), pose


def test_servoL_converges_through_the_task(oracle_mod):
    # the same convergence through the Robot-Reach task (gripper payload, dt 0.005, top-down orientation)
    b = oracle_mod.OracleBatch(oracle_mod.TASK_ROBOT_REACH, 1, 11)
    b.reset()
    target = np.array([[0.05, -0.45, 0.1]])
    for _ in range(40):
        r = b.step(target)
    assert np.allclose(r["obs"][0, 0:3], target[0], atol=1e-2)
    assert r["ncon"][0] == 0 and not r["fault"][0] and not r["ik_failed"][0]


def test_determinism_of_env(oracle_mod):
    # test/test_gym_envs.py:21-36: same seed -> same reset obs (atol 1e-6); different seed differs
    for task in (oracle_mod.TASK_POINTMASS, oracle_mod.TASK_ROBOT_REACH):
        b = oracle_mod.OracleBatch(task, 1, 0)
        b.seed(2025)
        o1 = b.reset()["obs"]
        b.seed(2025)
        o2 = b.reset()["obs"]
        b.seed(2024)
        o3 = b.reset()["obs"]
        assert np.allclose(o1, o2, atol=1e-6) and (o1 == o2).all()
        assert not np.allclose(o1, o3, atol=1e-6)


# ------------------------------------------------------------------ closed-form physics answers
def test_weld_first_substep_closed_form(oracle_mod):
    # One env, zero action for a step (body == mocap -> nothing moves), then a 0.05 step in x only:
    # first-substep acceleration a = D*K*imp*r/(m + D), D = 1/R, R = (1-imp)/imp * invweight,
    # invweight = (1/m + 1/m + 0)/3, refsafe time constant 2*dt (SURVEY.md App. A.2 with the
    # body_invweight0 averaging of mj_setConst).
    b = oracle_mod.OracleBatch(oracle_mod.TASK_POINTMASS, 1, 5, reward_type=2)
    o0 = b.reset()["obs"][0].copy()
    r = b.step(np.zeros((1, 2)))
    assert (r["obs"][0, :2] == o0[:2]).all()
    assert r["reward"][0] == -np.hypot(o0[0] - o0[2], o0[1] - o0[3])
    m, dt, rr = 0.1, 0.02, 0.05
    imp, dmax, tc = 0.95, 0.95, 2 * dt
    K, Bc = 1 / (dmax**2 * tc**2), 2 / (dmax * tc)
    D = 1 / ((1 - imp) / imp * (2 / m / 3))
    x, v = 0.0, 0.0
    # dm_control legacy stepping (mj_step2 then mj_step1): the first substep after before_step still
    # uses the constraint rows built from the OLD mocap position (residual 0 here) -> 4 driven substeps
    for sub in range(5):
        aref = -Bc * (-v) - K * imp * ((rr if sub > 0 else 0.0) - x)
        a = -D * aref / (m + D)  # row J = -1: minimise m a^2/2 + D (-a - aref)^2/2
        v += dt * a
        x += dt * v
    # only valid while |r| > solimp width (impedance saturated at 0.95) and away from the walls
    if abs(o0[0]) < 0.35:
        r = b.step(np.array([[rr, 0.0]]))
        assert np.isclose(r["obs"][0, 0] - o0[0], x, rtol=0, atol=1e-12)
        assert r["obs"][0, 1] == o0[1]


def test_pointmass_wall_contacts_detected(oracle_mod):
    b = oracle_mod.OracleBatch(oracle_mod.TASK_POINTMASS, 4, 2025, reward_type=2, time_limit=1e9)
    r = b.reset()
    assert (r["ncon"] == 1).all()  # sphere resting exactly on the ground plane: detected, not active
    for _ in range(40):
        r = b.step(np.tile([[0.05, 0.05]], (4, 1)))
    # pushed into the +x/+y corner: ground + two walls; the soft wall contacts balance the soft weld
    # (same solref/solimp) half-way between the wall surface (0.45) and the clipped mocap target (0.5)
    assert (r["ncon"] == 3).all()
    assert np.allclose(r["obs"][:, :2], 0.475, atol=1e-5)


# ------------------------------------------------------------------------- golden regression
@pytest.mark.parametrize("name,task", [("pointmass_n8_t70_seed2025", 0), ("robot_reach_n8_t110_seed2025", 1), ("button_push_eef_n8_t80_seed2025", 3),
                                       ("planar_push_n8_t70", 2), ("planar_push5_n4_t36", 2), ("planar_push_mesh_n8_t70", 2)])
def test_oracle_matches_golden(oracle_mod, name, task):
    g = np.load(GOLDEN / f"{name}.npz")
    N, T = g["actions"].shape[1], g["actions"].shape[0]
    seed = int(g["base_seed"]) if "base_seed" in g else 2025
    five = name.startswith("planar_push5")  # the reference's default of 5 blocks
    b = oracle_mod.OracleBatch(task, N, seed, action_type=1 if task == 3 else None, max_episode_steps=(14 if five else 25) if task == 2 else None,
                               n_objects=5 if five else None, nthreads=4,
                               block_shape=None if task != 2 else (oracle_mod.BLOCKS_MESH if "mesh" in name else oracle_mod.BLOCKS_BOX))
    r0 = b.reset()
    assert np.array_equal(r0["obs"], g["reset_obs"])
    for t in range(T):
        r = b.step(g["actions"][t])
        np.testing.assert_allclose(r["obs"], g["obs"][t], rtol=0, atol=1e-12)
        np.testing.assert_allclose(r["reward"], g["reward"][t], rtol=0, atol=1e-12)
        for k in ("step_type", "terminated", "truncated", "is_success", "ncon"):
            assert np.array_equal(r[k], g[k][t]), (k, t)
    assert (g["step_type"] == 2).any() and (g["step_type"] == 0).any()  # fixtures cover episode ends + auto-resets


def test_oracle_render_camera_model(oracle_mod):
    """Own image definition: pinhole camera in MuJoCo's convention, top-down at
    z = 2.4 with fovy 30 deg (point_reach.py:22). The red sphere and the green target box project
    where the state says; outside the 1 m arena the background is black."""
    b = oracle_mod.OracleBatch(oracle_mod.TASK_POINTMASS, 1, 2025)
    obs = b.reset()["obs"][0]
    img = b.render(128, 128)[0]
    assert img.shape == (128, 128, 3) and img.dtype == np.uint8
    f = 0.5 * 128 / np.tan(np.radians(15.0))

    def pix(x, y, z):
        return int(64 - f * y / (2.4 - z)), int(64 + f * x / (2.4 - z))  # row, col

    r, c = pix(obs[0], obs[1], 0.05)
    assert img[r, c, 0] > 150 and img[r, c, 0] > 1.5 * img[r, c, 1]  # pointmass: red, blended over the floor
    r, c = pix(obs[2], obs[3], 0.065)
    assert img[r, c, 1] > 100 and img[r, c, 0] < 40 and img[r, c, 2] < 40  # target: green box top
    assert (img[0, 0] == 0).all() and (img[127, 127] == 0).all()  # beyond the arena: nothing
    centre = img[60:68, 60:68].reshape(-1, 3).astype(int)
    assert (centre[:, 2] > centre[:, 0]).all()  # bluish checker floor


def test_oracle_render_robot_scene(oracle_mod):
    """Robot-Reach scene camera (robot_reach.py:52: pos (0,-1.1,0.5), quat (-0.7,-0.35,0,0), fovy 70):
    the white target site and the black gripper stand-in project where the state says."""
    b = oracle_mod.OracleBatch(oracle_mod.TASK_ROBOT_REACH, 1, 2025)
    obs = b.reset()["obs"][0]
    H = 160
    img = b.render(H, H)[0]
    q = np.array([-0.7, -0.35, 0.0, 0.0]); q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    f = 0.5 * H / np.tan(np.radians(35.0))

    def pix(p):
        pc = R.T @ (np.asarray(p) - np.array([0.0, -1.1, 0.5]))  # camera frame: looks along -z
        return int(H / 2 - f * pc[1] / -pc[2]), int(H / 2 + f * pc[0] / -pc[2])

    r, c = pix(obs[9:12])
    assert (img[r, c] > 150).all(), img[r, c]  # white target sphere
    tcp = obs[0:3]
    r, c = pix(tcp + np.array([0.0, 0.0, 0.08]))  # inside the gripper stand-in, above the TCP (tool points down)
    assert (img[r, c] < 40).all(), img[r, c]
    assert (img[H - 1, 0] > 30).all() and abs(int(img[H - 1, 0, 0]) - int(img[H - 1, 0, 2])) < 3  # grey floor


# ------------------------------------------------------------------------------------ Button-Push
def test_button_push_reset_draw_order(oracle_mod):
    """initialize_episode (robot_push_button.py:126-134): robot spawn xyz, then switch xyz, each three sequential
    uniform() calls (spaces.py:24-31) of RandomState(seed); switch observable = button xpos + 0.5*size[1] on all axes
    (switch.py:86-87); the IK places the TCP on the drawn position (to the DH-vs-MJCF model mismatch)."""
    for seed in (0, 5, 2025):
        rs = np.random.RandomState(seed)
        robot = [rs.uniform(lo, hi) for lo, hi in ((-0.2, 0.2), (-0.6, -0.3), (0.02, 0.3))]
        switch = [rs.uniform(lo, hi) for lo, hi in ((-0.2, 0.2), (-0.6, -0.3), (0.0, 0.1))]
        o = oracle_mod.OracleBatch(oracle_mod.TASK_BUTTON_PUSH, 1, seed).reset()["obs"][0]
        np.testing.assert_allclose(o[6:9], robot, atol=1.5e-3)  # DH IK vs MJCF body chain: < 1 mm apart, as in the reference's own frame test
        np.testing.assert_allclose(o[9:12], np.array(switch) + [0.01, 0.01, 0.05 + 0.01], atol=1e-15)
        assert o[12] == 0.0


def _demo_eef(obs):
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", GOLDEN / "make_golden.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.demo_actions(obs)


def test_button_push_scripted_policy_solves_task(oracle_mod):
    """The reference's own demonstration policy is the behavioural known answer for the contact + touch-sensor +
    switch chain: approach, press (a touch force inside [5, 200] N flips the switch on each rising edge, switch.py:51-60),
    retreat to the end pose -> sparse reward 1, terminated with discount 0 (robot_push_button.py:167-219). With the
    rigid gripper stand-in (D-1) the press force can leave and re-enter the band within one control step, so an
    episode shows an odd number of flips when it succeeds; most episodes do."""
    N = 16
    b = oracle_mod.OracleBatch(oracle_mod.TASK_BUTTON_PUSH, N, 100, autoreset=2, action_type=1, nthreads=4)
    r = b.reset()
    toggles = np.zeros(N, int)
    active = np.zeros(N, bool)
    done = np.zeros(N, bool)
    success = np.zeros(N, bool)
    touched = np.zeros(N, bool)
    for t in range(100):
        r = b.step(_demo_eef(r["obs"]))
        now = r["obs"][:, 12] > 0.5
        toggles += (now != active) & ~done
        active = now
        touched |= r["ncon"] > 0
        newly = (r["step_type"] == 2) & ~done
        won = newly & r["is_success"]
        assert (r["reward"][won] == 1.0).all() and (r["discount"][won] == 0.0).all() and r["terminated"][won].all()
        lost = newly & ~r["is_success"]
        assert t == 99 or not lost.any()  # the only other way out is the time limit ...
        assert (r["reward"][lost] == 0.0).all() and (r["discount"][lost] == 1.0).all() and r["truncated"][lost].all()
        assert (r["reward"][~newly & ~done] == 0.0).all()
        assert (np.linalg.norm(r["obs"][won, 6:9] - [-0.3, -0.2, 0.3], axis=1) < 0.05).all() and now[won].all()
        success |= won
        done |= newly
    assert done.all() and touched.all()
    assert success.sum() >= N // 2
    assert (toggles[success] % 2 == 1).all()


def test_button_push_time_limit_truncates(oracle_mod):
    # an idle policy never presses: episode ends by the 100-step time limit with discount 1 (truncated)
    b = oracle_mod.OracleBatch(oracle_mod.TASK_BUTTON_PUSH, 1, 3, action_type=1)
    r = b.reset()
    hold = np.concatenate([r["obs"][0, 6:9], [0.0]])[None]
    for t in range(100):
        r = b.step(hold)
        assert (r["step_type"][0] == 2) == (t == 99)
    assert r["truncated"][0] and not r["terminated"][0] and r["discount"][0] == 1.0 and r["reward"][0] == 0.0
    np.testing.assert_allclose(r["obs"][0, 6:9], hold[0, :3], atol=4e-3)  # the servo holds the pose (payload sag + DH/MJCF offset only)


# ------------------------------------------------------------------------------------ Planar-Push
def _convex(L, t1, s1, p1, R1, t2, s2, p2, R2):
    out = np.zeros(7)
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (s1, p1, R1, s2, p2, R2)]
    hit = L.om_debug_convex(t1, a[0].ctypes.data_as(C.c_void_p), a[1].ctypes.data_as(C.c_void_p), a[2].ctypes.data_as(C.c_void_p),
                            t2, a[3].ctypes.data_as(C.c_void_p), a[4].ctypes.data_as(C.c_void_p), a[5].ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return bool(hit), out[0], out[1:4], out[4:7]


def _rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def test_convex_penetration_known_answers(oracle_mod):
    """Own MPR for the convex pairs of the Planar-Push scene (role of mjc_Convex/libccd): closed-form cases, then the
    defining properties on random poses: translating geom2 by depth along the normal separates the pair, half of it
    does not, and the contact point lies inside both (depth-inflated) shapes."""
    L = oracle_mod.lib()
    BOX, CYL, I = 6, 5, np.eye(3)
    h, cyl = [0.02, 0.02, 0.019], [0.02, 0.05, 0.0]
    hit, depth, n, pos = _convex(L, BOX, h, [0, 0, 0], I, BOX, h, [0.03, 0.001, 0.002], I)
    assert hit and abs(depth - 0.01) < 1e-9 and np.allclose(n, [1, 0, 0], atol=1e-9) and abs(pos[0] - 0.015) < 1e-9
    assert not _convex(L, BOX, h, [0, 0, 0], I, BOX, h, [0.0401, 0, 0], I)[0]
    hit, depth, n, pos = _convex(L, CYL, cyl, [-0.035, 0.003, 0.05], I, BOX, h, [0, 0, 0.019], I)   # cylinder side vs box face
    assert hit and abs(depth - 0.005) < 1e-7 and np.allclose(n, [1, 0, 0], atol=1e-5) and abs(pos[0] + 0.0175) < 1e-7
    hit, depth, n, pos = _convex(L, CYL, cyl, [0.005, 0, 0.038 + 0.05 - 0.002], I, BOX, h, [0, 0, 0.019], I)  # flat cap on the top face
    assert hit and abs(depth - 0.002) < 1e-9 and np.allclose(n, [0, 0, -1], atol=1e-9) and 0.036 - 1e-12 <= pos[2] <= 0.038 + 1e-12  # inside the overlap slab
    assert not _convex(L, CYL, cyl, [-0.0401, 0, 0.05], I, BOX, h, [0, 0, 0.019], I)[0]

    def inside(tp, s, p, R, x, tol):
        loc = R.T @ (x - p)
        if tp == BOX:
            return bool(np.all(np.abs(loc) <= np.array(s) + tol))
        return np.hypot(loc[0], loc[1]) <= s[0] + tol and abs(loc[2]) <= s[1] + tol

    rs = np.random.RandomState(0)
    hits = 0
    for t in range(600):
        R1, R2 = _rot(rs.normal(size=3), rs.uniform(0, 3)), _rot(rs.normal(size=3), rs.uniform(0, 3))
        p2 = rs.uniform(-0.05, 0.05, 3)
        t1, s1 = (CYL, cyl) if t % 2 else (BOX, h)
        hit, depth, n, pos = _convex(L, t1, s1, [0, 0, 0], R1, BOX, h, p2, R2)
        if not hit:
            continue
        hits += 1
        assert depth >= 0 and abs(np.linalg.norm(n) - 1) < 1e-12
        assert not _convex(L, t1, s1, [0, 0, 0], R1, BOX, h, p2 + (depth + 1e-4) * n, R2)[0]
        assert _convex(L, t1, s1, [0, 0, 0], R1, BOX, h, p2 + 0.5 * depth * n, R2)[0]
        assert inside(t1, s1, np.zeros(3), R1, pos, depth + 1e-6) and inside(BOX, h, p2, R2, pos, depth + 1e-6)
    assert hits > 200


@pytest.mark.parametrize("mesh", [True, False])
def test_planar_push_reset_and_settle(oracle_mod, mesh):
    """robot_planar_push.py:144-176 (intended semantics): draw order = per block (category, colour, scale) of
    GoogleBlockProp.sample_random_object (initialize_episode_mjcf; mesh blocks only), then robot xyz, target xyz, then block xyz
    per block (re-drawn together while anything touches), 150 physics steps to settle: blocks end flat on the floor, 4 vertex /
    corner contacts each, at their drawn xy; physics time = 150 * 0.005."""
    for seed in (2025, 3):
        b = oracle_mod.OracleBatch(oracle_mod.TASK_PLANAR_PUSH, 1, seed, block_shape=oracle_mod.BLOCKS_MESH if mesh else oracle_mod.BLOCKS_BOX)
        r = b.reset()
        rs = np.random.RandomState(seed)
        if mesh:
            shapes = [(int(rs.uniform(0, 4)), int(rs.uniform(0, 6)), rs.uniform(0.8, 1.2)) for _ in range(2)]
            cat, col, sc = b.block_shapes()
            assert [(cat[0, k], col[0, k], sc[0, k]) for k in range(2)] == shapes
        robot = [rs.uniform(lo, hi) for lo, hi in ((-0.2, 0.2), (-0.6, -0.3), (0.02, 0.02))]
        target = [rs.uniform(lo, hi) for lo, hi in ((-0.15, 0.15), (-0.55, -0.35), (0.001, 0.005))]
        blocks = [[rs.uniform(lo, hi) for lo, hi in ((-0.15, 0.15), (-0.55, -0.35), (0.05, 0.2))] for _ in range(2)]
        o = r["obs"][0]
        np.testing.assert_allclose(o[0:3], robot, atol=1.5e-3)
        np.testing.assert_allclose(o[3:5], target[:2], atol=1e-15)
        qpos, qvel, tm = b.get_state()
        apart = np.linalg.norm(np.array(blocks[0]) - np.array(blocks[1])) > 0.08  # first draw accepted when nothing overlaps
        if apart:
            np.testing.assert_allclose(o[5:9], np.array(blocks)[:, :2].ravel(), atol=1e-6 if not mesh else 2e-3)  # a mesh block creeps a little while settling
        assert r["ncon"][0] == 8 and abs(tm[0] - 0.75) < 1e-12
        for i in range(2):
            z, quat = qpos[0, 6 + 7 * i + 2], qpos[0, 6 + 7 * i + 3: 6 + 7 * i + 7]
            assert -1e-3 < z < 1e-4 and abs(abs(quat[0]) - 1) < (1e-6 if not mesh else 1e-3)   # resting on the floor (soft contact: tiny penetration), upright
        assert np.abs(qvel[0, 6:]).max() < 1e-3


def test_planar_push_free_body_closed_forms(oracle_mod):
    """Free-joint integration known answers (no contact): semi-implicit Euler free fall
    z_k = z_0 - g dt^2 k (k + 1) / 2, x_k = x_0 + v_x k dt; a block spinning about a principal axis keeps its
    body-frame angular velocity and turns by w k dt (quaternion integration on the local angular velocity)."""
    b = oracle_mod.OracleBatch(oracle_mod.TASK_PLANAR_PUSH, 1, 1)
    b.reset()
    qpos, qvel, _ = b.get_state()
    qpos[0, 6:13] = [0.1, -0.45, 0.6, 1, 0, 0, 0]
    qpos[0, 13:20] = [-0.1, -0.45, 0.7, 1, 0, 0, 0]
    qvel[:] = 0
    qvel[0, 6] = 0.2          # block 0 drifts in x
    qvel[0, 12 + 5] = 3.0     # block 1 spins about its local z (a principal axis)
    b.set_state(qpos, qvel)
    k, dt, g = 40, 0.005, 9.81
    b.substeps(k)
    q, v, _ = b.get_state()
    np.testing.assert_allclose(q[0, 8], 0.6 - g * dt * dt * k * (k + 1) / 2, atol=1e-12)
    np.testing.assert_allclose(q[0, 6], 0.1 + 0.2 * k * dt, atol=1e-12)
    np.testing.assert_allclose(v[0, 8], -g * k * dt, atol=1e-12)
    ang = 3.0 * k * dt
    np.testing.assert_allclose(q[0, 16:20], [np.cos(ang / 2), 0, 0, np.sin(ang / 2)], atol=1e-12)
    np.testing.assert_allclose(v[0, 12 + 3: 12 + 6], [0, 0, 3.0], atol=1e-12)


def test_planar_push_pushing_moves_the_block(oracle_mod):
    # the cylinder EEF driven through a block's position pushes it along (contact solver path, condim-4 pyramids)
    b = oracle_mod.OracleBatch(oracle_mod.TASK_PLANAR_PUSH, 4, 2025, nthreads=4)
    r = b.reset()
    start = r["obs"][:, 5:7].copy()
    saw_eef_contact = False
    for t in range(80):
        tcp, blk = r["obs"][:, :2], r["obs"][:, 5:7]
        a = tcp + np.clip(blk - tcp, -0.02, 0.02)
        r = b.step(a)
        saw_eef_contact |= bool((r["ncon"] > 8).any() or (r["ncon"] < 8).any())
        assert not r["fault"].any()
    moved = np.linalg.norm(r["obs"][:, 5:7] - start, axis=1)
    assert saw_eef_contact and (moved > 0.02).all(), moved
    assert (r["reward"] < 0).all()  # dense reward = 0.1 * (-mean distance - 0.1 * nearest)


def test_planar_push_step_limit_truncates(oracle_mod):
    b = oracle_mod.OracleBatch(oracle_mod.TASK_PLANAR_PUSH, 1, 5, max_episode_steps=7)
    r = b.reset()
    hold = r["obs"][:, :2].copy()
    for t in range(7):
        r = b.step(hold)
        assert (r["step_type"][0] == 2) == (t == 6)
    assert r["truncated"][0] and not r["terminated"][0] and r["discount"][0] == 1.0


def test_joint_space_move_converges_like_reference_test(oracle_mod):
    """test/test_ur_control_api.py:7-28 (test_moveJ) re-expressed on the path the tasks use: a joint-space target
    (the reference's home pose, robot.py:307) commanded through servoJ control step after control step is reached within
    the reference's own tolerance 1e-2 rad (position servos + gripper payload sag)."""
    target = np.array([0.0, -0.5, 0.5, -0.5, -0.5, -0.5]) * np.pi
    b = oracle_mod.OracleBatch(oracle_mod.TASK_BUTTON_PUSH, 3, 12, action_type=0)
    r = b.reset()
    a = np.tile(np.concatenate([target, [0.0]]), (3, 1))
    for _ in range(40):
        r = b.step(a)
    assert np.abs(r["obs"][:, :6] - target).max() < 1e-2
    assert not r["fault"].any()


def test_scene_constants_match_reference_data_files():
    """The scene constants tagged [REF] and the block hull tables against the DATA FILES the reference holds
    (mujoco_sim/mjcf/walled_pointmass_arena.xml, google_language_table_blocks/*.xml, *.obj), read into
    tests/golden/reference_scene_data.json by tests/golden/make_reference_data_pins.py in the build container."""
    import json
    import re

    ref = json.loads((GOLDEN / "reference_scene_data.json").read_text())
    inc = GOLDEN.parents[1] / "include"
    spec, hulls = (inc / "mjs_scene_spec.h").read_text(), (inc / "mjs_block_hulls.h").read_text()

    def arr(text, name, shape=None):
        m = re.search(r"\b" + name + r"\b(?:\[[^=]*)?\s*=\s*(\{.*?\}|[^;]*);", text, re.S)
        assert m, name
        body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
        vals = np.array([float(x) for x in re.findall(r"-?\d+\.?\d*(?:e-?\d+)?", body.replace("f,", ",").replace("f}", "}"))])
        return vals.reshape(shape) if shape else vals

    a = ref["arena"]
    # walled arena: ground half-size, wall planes at +-0.5 facing inwards with half-height 0.02, lights, colours
    lo, hi, wz = arr(spec, "MJS_PM_ARENA_LO")[0], arr(spec, "MJS_PM_ARENA_HI")[0], arr(spec, "MJS_PM_WALL_Z")[0]
    assert a["planes"]["ground"]["size"][:2] == [hi, hi] and lo == -hi
    for name, axis, sign in (("wall_x", 0, -1), ("wall_y", 1, -1), ("wall_neg_x", 0, 1), ("wall_neg_y", 1, 1)):
        w = a["planes"][name]
        assert w["pos"][axis] == sign * hi and w["pos"][2] == wz and w["zaxis"][axis] == -sign and w["size"][2] == wz, name
    np.testing.assert_allclose(arr(spec, "MJS_PM_LIGHT_POS", (2, 3)), a["lights"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(arr(spec, "MJS_PM_GRID_RGB1"), a["grid_rgb1"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(arr(spec, "MJS_PM_GRID_RGB2"), a["grid_rgb2"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(arr(spec, "MJS_PM_WALL_RGB"), a["decoration_rgba"][:3], rtol=0, atol=1e-7)
    # blocks: mesh geoms turned z-up by quat (1 1 0 0); the hull tables' bounding boxes are the meshes' (a convex hull keeps
    # the extremes), a hull has no more vertices than its mesh, every hull vertex lies inside the mesh's box
    ncat = int(re.search(r"#define MJS_HULL_NCAT (\d+)", hulls).group(1))
    maxv = int(re.search(r"#define MJS_HULL_MAXV (\d+)", hulls).group(1))
    cats = ("cube", "moon", "pentagon", "star")
    assert ncat == len(cats) == len(ref["blocks"])
    nv = arr(hulls, "MJS_HULL_NV").astype(int)
    blo, bhi = arr(hulls, "MJS_HULL_BOX_LO", (ncat, 3)), arr(hulls, "MJS_HULL_BOX_HI", (ncat, 3))
    vert = arr(hulls, "MJS_HULL_VERT", (ncat, maxv, 3))
    for c, cat in enumerate(cats):
        b = ref["blocks"][cat]
        assert b["geom_type"] == "mesh" and b["quat"] == [1.0, 1.0, 0.0, 0.0], cat
        assert 4 <= nv[c] <= b["n_vertices"], cat
        np.testing.assert_allclose(blo[c], b["body_frame_bbox_lo"], rtol=0, atol=1e-12, err_msg=cat)
        np.testing.assert_allclose(bhi[c], b["body_frame_bbox_hi"], rtol=0, atol=1e-12, err_msg=cat)
        v = vert[c, :nv[c]]
        np.testing.assert_allclose(v.min(0), b["body_frame_bbox_lo"], rtol=0, atol=1e-12, err_msg=cat)
        np.testing.assert_allclose(v.max(0), b["body_frame_bbox_hi"], rtol=0, atol=1e-12, err_msg=cat)


def test_scene_constants_match_reference_source_literals():
    """Every [REF] constant of include/mjs_scene_spec.h that restates a numeric literal of the reference's task / entity
    sources against the numbers tests/golden/make_reference_data_pins.py read from those lines (file and line recorded in
    tests/golden/reference_scene_data.json)."""
    import json
    import re

    lit = json.loads((GOLDEN / "reference_scene_data.json").read_text())["literals"]
    spec = (GOLDEN.parents[1] / "include" / "mjs_scene_spec.h").read_text()

    def const(name):
        m = re.search(r"\b" + name + r"\b(?:\[[^=]*)?\s*=\s*(\{.*?\}|[^;]*);", spec, re.S)
        assert m, name
        body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S).replace("f,", ",").replace("f}", "}")
        return [float(x) for x in re.findall(r"-?\d+\.?\d*(?:e-?\d+)?", re.sub(r"(?<=\d)f\b", "", body))]

    def box(prefix):  # EuclideanSpace((x0, x1), (y0, y1), (z0, z1)) -> LO, HI
        lo, hi = const(prefix + "_LO"), const(prefix + "_HI")
        return [lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]]

    checked = 0
    for name, pin in lit.items():
        where = f"{name} vs {pin['file']}:{pin['line']}"
        if name.startswith("MJS_"):
            got = [v for n in name.split() for v in const(n)]
        elif name.endswith("_SPACE"):
            got = box({"BP_ROBOT_SPACE": "MJS_BP_ROBOT_SPACE", "BP_SWITCH_SPACE": "MJS_BP_SWITCH_SPACE", "RR_SPACE": "MJS_RR_SPACE",
                       "PP_ROBOT_SPACE": "MJS_PP_ROBOT_SPACE", "PP_OBJECT_SPACE": "MJS_PP_OBJECT_SPACE", "PP_TARGET_SPACE": "MJS_PP_TARGET_SPACE"}[name])
        elif name == "SW_BOX_SIZE":      # box geom half sizes = box_size / 2, button radius = box_size / 2.5 (switch.py:27,34)
            got = [2 * const("MJS_SW_BOX_HALF")[0]]
            assert abs(const("MJS_SW_BUTTON_RADIUS")[0] - pin["numbers"][0] / 2.5) < 1e-15, where
        elif name == "SW_BOX_HEIGHT":    # button geom and site at z = box_height (switch.py:36,39)
            got = const("MJS_SW_BUTTON_Z")
        elif name.startswith("BLOCK_COLOR_"):
            idx = ("RED", "BLUE", "GREEN", "YELLOW", "ORANGE", "PURPLE").index(name[len("BLOCK_COLOR_"):])
            got = const("MJS_BLOCK_COLORS")[3 * idx:3 * idx + 3] + [1.0]
        else:
            raise AssertionError(f"unmapped pin {name}")
        assert len(got) == len(pin["numbers"]), where
        np.testing.assert_allclose(got, pin["numbers"], rtol=0, atol=1e-7, err_msg=where)  # float32 colour literals
        checked += 1
    assert checked == len(lit) >= 40


def test_static_equilibrium_closed_form(oracle_mod):
    """Closed-form pin of gravity compensation, the position servos and the kinematic chain together: a stand-alone UR5e held
    at q_home by its servos. Bare arm: every arm body has gravcomp 1 (robot.py:312-318), so nothing sags (q == ctrl). With
    the 0.925 kg gripper lump (not compensated) the servos settle where kp (ctrl - q) = -J_com(q)^T m g, evaluated here
    independently with numpy from include/mjs_scene_spec.h (no oracle code in the right-hand side)."""
    import re

    om = oracle_mod
    spec = (GOLDEN.parents[1] / "include" / "mjs_scene_spec.h").read_text()

    def arr(name, shape):
        m = re.search(r"\b" + name + r"\b(?:\[[^=]*)?\s*=\s*(\{.*?\}|[^;]*);", spec, re.S)
        vals = [float(x) for x in re.findall(r"-?\d+\.?\d*(?:e-?\d+)?", re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S))]
        return np.array(vals).reshape(shape)

    pos, quat, axis = arr("MJS_UR_BODY_POS", (7, 3)), arr("MJS_UR_BODY_QUAT", (7, 4)), arr("MJS_UR_JNT_AXIS", (6, 3))
    fpos, fquat, kp = arr("MJS_UR_FLANGE_POS", (3,)), arr("MJS_UR_FLANGE_QUAT", (4,)), arr("MJS_UR_ACT_KP", (6,))
    mass, ipos, home = arr("MJS_G2F85_MASS", (1,))[0], arr("MJS_G2F85_IPOS", (3,)), arr("MJS_UR_HOME_Q", (6,))
    grav = arr("MJS_GRAVITY_Z", (1,))[0]

    def q2m(q):
        w, x, y, z = q / np.linalg.norm(q)
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    def rot(ax, ang):  # Rodrigues
        ax = ax / np.linalg.norm(ax)
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K

    def gravity_torque(q):
        R, p = q2m(quat[0]), pos[0].copy()
        anchors, axes = [], []
        for b in range(1, 7):
            p = p + R @ pos[b]
            R = R @ q2m(quat[b])
            anchors.append(p.copy())
            axes.append(R @ axis[b - 1])
            R = R @ rot(axis[b - 1], q[b - 1])
        pf, Rf = p + R @ fpos, R @ q2m(fquat)
        com = pf + Rf @ ipos
        f = np.array([0.0, 0.0, mass * grav])
        return np.array([np.dot(np.cross(axes[j], com - anchors[j]), f) for j in range(6)])

    for eef in (om.UR_EEF_NONE, om.UR_EEF_GRIPPER):
        st, _, _ = om.ur_robot_run(om.ur_robot_state(home), np.zeros(7), om.UR_CMD_NONE, 0.0, 6000, eef=eef, dt=0.002)
        q, v = st[0:6], st[6:12]
        assert np.abs(v).max() < 1e-9
        if eef == om.UR_EEF_NONE:
            np.testing.assert_allclose(q, home, rtol=0, atol=1e-10)  # gravcomp: the bare arm does not sag
        else:
            tau = gravity_torque(q)
            assert np.abs(tau).max() > 0.5  # the lump does load the joints
            np.testing.assert_allclose(kp * (home - q), -tau, rtol=0, atol=1e-6)


def test_reach_fast_path_guard_assumptions_hold_in_the_action_box(oracle_mod):
    """The Robot-Reach fast path (mjs_reach.h, "the fast path's guard") assumes about the task's OWN action box
    (robot_reach.py:187-201, x in [-0.1, 0.1], y in [-0.6, -0.4], z in [0.02, 0.2], tool down): every configuration the
    oracle reaches under uniform in-box targets keeps the arm's collision geoms (all but the two that cannot move vertically)
    at least CLEAR_MARGIN = 0.10 m above the floor, and the joint-space travel measure of a control step,
    sum_j (|q1_j - q0_j| + 0.05 |v_j|)^2 with q1 the IK target, stays below TRAVEL2_MAX = 2.5. Checked here on the CPU with an
    independent numpy FK of include/mjs_scene_spec.h (so that a change of the scene constants that breaks the assumption is
    caught without a GPU)."""
    import re
    from pathlib import Path

    spec = (Path(__file__).resolve().parents[1] / "include" / "mjs_scene_spec.h").read_text()

    def arr(name, shape):
        m = re.search(name + r"(?:\[[^\]]*\])*\s*=\s*\{(.*?)\};", spec, re.S)
        body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
        return np.array([float(x) for x in re.findall(r"-?\d+\.?\d*(?:e-?\d+)?", body)]).reshape(shape)

    BP, BQ, AX = arr("MJS_UR_BODY_POS", (7, 3)), arr("MJS_UR_BODY_QUAT", (7, 4)), arr("MJS_UR_JNT_AXIS", (6, 3))
    CB, CP, CQ, CS = arr("MJS_UR_COL_BODY", (10,)).astype(int), arr("MJS_UR_COL_POS", (10, 3)), arr("MJS_UR_COL_QUAT", (10, 4)), arr("MJS_UR_COL_SIZE", (10, 2))

    def q2m(q):
        w, x, y, z = q / np.linalg.norm(q)
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    def rot(ax, a):
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K

    def clearance(q):
        R, p, Rs, ps = np.eye(3), np.zeros(3), [], []
        for b in range(7):
            p = p + R @ BP[b]
            R = R @ q2m(BQ[b])
            if b > 0:
                R = R @ rot(AX[b - 1], q[b - 1])
            Rs.append(R.copy()); ps.append(p.copy())
        m = 1e9
        for g in range(2, 10):  # geoms 0, 1 and the shoulder-side end of geom 2 sit on the first two joint axes: fixed heights (mjs_arm_stage.h)
            b = CB[g]
            gp, ax = ps[b] + Rs[b] @ CP[g], (Rs[b] @ q2m(CQ[g]))[:, 2]
            ends = [gp[2] + CS[g][1] * ax[2]] if g == 2 else [gp[2] - CS[g][1] * abs(ax[2])]
            m = min(m, min(ends) - CS[g][0])
        return m

    N, T = 64, 110
    b = oracle_mod.OracleBatch(oracle_mod.TASK_ROBOT_REACH, N, 2025, nthreads=8)
    o = b.reset()
    acts = np.random.RandomState(12345).uniform([-0.1, -0.6, 0.02], [0.1, -0.4, 0.2], (T, N, 3))
    q, qold = o["obs"][:, 3:9].copy(), o["obs"][:, 3:9].copy()
    min_clr, max_t2 = 1e9, 0.0
    for t in range(T):
        v = (q - qold) / 0.1
        for i in range(0, N, 4):
            min_clr = min(min_clr, clearance(q[i]))
            Tm = np.eye(4)
            Tm[:3, :3] = np.diag([1.0, -1.0, -1.0])
            Tm[:3, 3] = acts[t, i] + np.array([0, 0, 0.174])
            r = oracle_mod.ur5e_ik_closest(Tm, q[i])
            q1 = r[0] if isinstance(r, tuple) else r
            if q1 is not None:
                d = np.abs(np.asarray(q1) - q[i]) + 0.05 * np.abs(v[i])
                max_t2 = max(max_t2, float((d * d).sum()))
        qold = q.copy()
        o = b.step(acts[t])
        q = o["obs"][:, 3:9].copy()
        fresh = o["step_type"] == 0
        qold[fresh] = q[fresh]
    assert min_clr > 0.14, min_clr     # CLEAR_MARGIN 0.10 with room (the wrist_3 capsule, 0.154 m at the lowest target)
    assert max_t2 < 1.6, max_t2        # TRAVEL2_MAX 2.5 with room
