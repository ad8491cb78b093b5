"""the articulated Robotiq 2F-85 in the CPU oracle (no GPU needed).

The reference loads the gripper's MJCF from an absent package (entities/eef/gripper.py:5,37), so the model is re-authored
([MEN] constants, include/mjs_scene_spec.h MJS_G85_*) and the physics stays "parity unpinned". What the reference itself
holds about the gripper is checked here against the oracle's model:
  * eight hinge joints in the order of ``joint_home_positions`` (gripper.py:40), one actuator, two driver joints coupled by
    an equality (gripper.py:62-66),
  * ``joint_home_positions`` is a rest configuration (gripper.py:40),
  * the driver range 0..0.8 maps to an 85 mm stroke, approximated by ``_joint_angle_to_finger_distance`` (gripper.py:73-75),
  * ``move(w)`` (gripper.py:79-84) brings ``get_finger_opening`` (gripper.py:76-77) to w,
  * the TCP offset 0.174 m (gripper.py:46-48) is where the closed finger tips are,
plus the solver-side known answers of what the oracle had to learn for it: connect / joint-coupling rows (loop closure by
Newton on the rows' own Jacobian), the fixed-tendon actuator, elliptic friction cones (KKT residual, forces inside the cone).
"""
import numpy as np
import pytest

HOME = np.array([0.0, 0.0, 0.005, -0.01, 0.0, 0.0, 0.005, -0.01])  # gripper.py:40
OPEN, MAX_DRIVER = 0.085, 0.8                                     # gripper.py:50-52, :38
# geoms of the articulated Button-Push scene: 0 floor, 1..10 arm, 11 / 12 right pad boxes, 13 / 14 left pad boxes, 15 switch box, 16 button
G_RPAD1, G_LPAD1, PAD_HALF_Y = 11, 13, 0.004
EQ, LIMIT, ELLIPTIC = 0, 3, 7


def _env(oracle_mod, n=1, seed=5, **kw):
    return oracle_mod.OracleBatch(oracle_mod.TASK_BUTTON_PUSH, n, seed, gripper_model=1, autoreset=oracle_mod.AUTORESET_DISABLED, **kw)


def _ref_opening(theta):
    return OPEN * (1 - np.sin(theta) / np.sin(MAX_DRIVER))  # gripper.py:73-75


def _ctrl_of(w):
    return np.arcsin((1 - w / OPEN) * np.sin(MAX_DRIVER)) / MAX_DRIVER * 255  # gripper.py:77-84


def _pad_gap(b, i=0):
    pr, _ = b.geom_pose(i, G_RPAD1)
    pl, _ = b.geom_pose(i, G_LPAD1)
    return np.linalg.norm(pr - pl) - 2 * PAD_HALF_Y


def test_model_structure_matches_what_the_reference_holds(oracle_mod):
    b = _env(oracle_mod)
    d = b.model_dims()
    assert d["nq"] == d["nv"] == 6 + len(HOME) and d["njnt"] == 14  # gripper.py:40: eight gripper joints (all hinges)
    assert d["nu"] == 7 and d["neq"] == 3 and d["cone"] == 1        # six servos + fingers_actuator; 2 connects + the driver coupling; elliptic
    b.reset()
    e = b.efc(0)
    assert (e["type"][:7] == EQ).all() and np.abs(e["pos"][:7]).max() < 1e-15  # both loops are closed at qpos0 by construction
    reduced = oracle_mod.OracleBatch(oracle_mod.TASK_BUTTON_PUSH, 1, 5)
    assert reduced.model_dims()["nv"] == 6 and reduced.model_dims()["cone"] == 0  # D-1b stays the default


def _close_loops(b, theta, start=None):
    """Newton on the seven equality rows with the rows' own Jacobian: right driver = theta, couplers at their upper limit 0
    (where the spring_link spring holds them), unknowns = left driver + both (spring_link, follower) pairs. Start: the parallel
    linkage's guess (spring_link = theta, follower = -theta) or a neighbouring solution."""
    qp, qv, _ = b.get_state()
    q = qp[0].copy()
    q[6:] = [theta, 0, theta, -theta, theta, 0, theta, -theta] if start is None else start[6:]
    q[6] = theta
    free = [8, 9, 10, 12, 13]
    res = None
    for _ in range(12):
        b.set_state(q[None], np.zeros((1, 14)))
        e = b.efc(0)
        r, J = e["pos"][:7], e["J"][:7][:, free]
        res = np.abs(r).max()
        if res < 1e-13:
            break
        q[free] -= np.linalg.lstsq(J, r, rcond=None)[0]
    return q, res


def test_four_bar_loops_close_over_the_driver_range(oracle_mod):
    """connect rows (residual AND Jacobian): Newton with the rows' Jacobian closes both finger linkages to 1e-9 at every
    driver angle, quadratically (a wrong Jacobian would crawl); the closed linkage is the parallel one: follower = -spring_link
    to a few mrad (the pads stay parallel), left = right (the mirrored finger), the left driver follows the joint equality."""
    b = _env(oracle_mod)
    b.reset()
    for theta in np.linspace(0.0, MAX_DRIVER, 9):
        q, res = _close_loops(b, theta)
        assert res <= 1e-9, (theta, res)
        assert abs(q[10] - theta) < 1e-12                      # right_driver = left_driver (polycoef 0 1 0 0 0)
        np.testing.assert_allclose(q[8:10], q[12:14], atol=1e-9)
        assert abs(q[8] - theta) < 0.02 and abs(q[9] + q[8]) < 0.04, (theta, q[6:])  # |AE| = 57.50 mm vs |BD'| = 57.49 mm: a parallelogram to 2 degrees
        _, mat = b.geom_pose(0, G_RPAD1)
        _, mat0 = b.geom_pose(0, G_LPAD1)
        assert abs(np.dot(mat[:, 1], mat0[:, 1]) + 1) < 5e-3   # pad faces opposite each other, parallel to ~4 degrees at most


def test_finger_opening_against_the_reference_formula(oracle_mod):
    """`_joint_angle_to_finger_distance` (gripper.py:73-75) is, in the reference's words, a hacky linearisation: exact at both
    ends of the stroke, a few mm off in between. The kinematic opening of the closed linkage (gap between the pads' inner faces):
    85 mm at 0, ~0 at 0.8, within 8 mm of the formula at 0.4; monotone; finger tips 13-14 mm further out when closed (the
    reference's TCP offset 0.174 is the CLOSED tips, gripper.py:46-48)."""
    b = _env(oracle_mod)
    b.reset()
    gaps, tips = [], []
    for theta in (0.0, 0.4, 0.8):
        _close_loops(b, theta)
        gaps.append(_pad_gap(b))
        pr, mat = b.geom_pose(0, G_RPAD1)
        qp, _, _ = b.get_state()
        # distance of the pad's far end from the flange along the gripper axis: pad centre + half length, flange = site 0 (obs gives the TCP)
        tips.append(pr + mat[:, 2] * 0.009375)
    # at 0.8 the (kinematic) far-end boxes overlap by 2.5 mm: the pads meet at 0.79 rad, just before the driver's stop
    assert abs(gaps[0] - OPEN) < 1e-3 and abs(gaps[2]) < 3e-3 and gaps[0] > gaps[1] > gaps[2]
    for theta, g in zip((0.0, 0.4, 0.8), gaps):
        assert abs(g - _ref_opening(theta)) < 8e-3, (theta, g, _ref_opening(theta))
    assert abs(gaps[1] - _ref_opening(0.4)) > 2e-3  # ... and it IS an approximation (6.7 mm at mid stroke)
    stroke = np.linalg.norm(tips[2] - tips[0])
    assert 0.040 < stroke < 0.048  # 42.6 mm inwards + 13.8 mm outwards per finger


def test_home_positions_are_a_rest_configuration(oracle_mod):
    """gripper.py:40: joint_home_positions = (0, 0, 0.005, -0.01) per finger. From there, with ctrl = 0 (reset(): gripper.py:86-93),
    the fingers stay put: every joint within 7 mrad of home after 2 s, velocities gone. (The oracle's own equilibrium is
    (0.003, 0.0004, 0.004, -0.005): the spring_link spring against the open actuator and the coupler's stop.)"""
    b = _env(oracle_mod)
    b.reset()
    qp, _, _ = b.get_state()
    q = qp[0].copy()
    q[6:] = HOME
    b.set_state(q[None], np.zeros((1, 14)))
    b.set_ctrl(6, 0.0)
    b.substeps(400)
    qp, qv, _ = b.get_state()
    assert np.abs(qp[0, 6:] - HOME).max() < 7e-3, qp[0, 6:]
    assert np.abs(qv[0, 6:]).max() < 1e-6
    assert np.abs(qp[0, :6] - q[:6]).max() < 5e-3  # and the arm holds its pose under the gripper's weight


def test_move_reaches_the_commanded_opening(oracle_mod):
    """Robotiq2f85.move(w) (gripper.py:79-84) -> fingers_actuator on the tendon 0.5 (right + left driver) -> at rest
    get_finger_opening (gripper.py:76-77, the reference's own read-back) = w to 1 mm; both drivers equal (joint equality); the
    real gap between the pads follows (monotone, open ~85 mm, closed < 1 mm: the pads meet just before the driver's stop)."""
    b = _env(oracle_mod)
    b.reset()
    last_gap = 1.0
    for w in (0.085, 0.06, 0.04, 0.02, 0.0):
        b.set_ctrl(6, _ctrl_of(w))
        b.substeps(500)
        qp, qv, _ = b.get_state()
        assert np.abs(qv[0, 6:]).max() < 1e-6
        assert abs(_ref_opening(qp[0, 6]) - w) < 1e-3, (w, qp[0, 6])
        assert abs(qp[0, 6] - qp[0, 10]) < 2e-4
        gap = _pad_gap(b)
        assert gap < last_gap
        last_gap = gap
    assert last_gap < 1e-3
    b.set_ctrl(6, 0.0)
    b.substeps(500)
    assert abs(_pad_gap(b) - OPEN) < 1.5e-3


def test_tendon_actuator_force_and_clamp(oracle_mod):
    """fingers_actuator: force = 0.3137255 ctrl - 100 len - 10 vel on the tendon, clamped to +-5 N, spread 0.5 / 0.5 on the two
    drivers. At rest and open, ctrl = 255 saturates it: qfrc_smooth of both drivers jumps by exactly 0.5 * 5 N."""
    b = _env(oracle_mod)
    b.reset()
    b.substeps(300)
    qp, qv, _ = b.get_state()
    b.set_ctrl(6, 0.0)
    b.set_state(qp, qv)
    _, f0, _ = b.dynamics(0)
    length = 0.5 * (qp[0, 6] + qp[0, 10])
    b.set_ctrl(6, 255.0)
    b.set_state(qp, qv)
    _, f1, _ = b.dynamics(0)
    unclamped0 = -100 * length - 10 * 0.5 * (qv[0, 6] + qv[0, 10])
    np.testing.assert_allclose(f1[[6, 10]] - f0[[6, 10]], 0.5 * (5.0 - unclamped0), atol=1e-9)
    others = [k for k in range(14) if k not in (6, 10)]
    np.testing.assert_allclose(f1[others], f0[others], atol=1e-12)


def test_elliptic_cone_solution_is_a_kkt_point_inside_the_cone(oracle_mod):
    """Pads pressed on the floor and dragged sideways (EEF actions below the floor): elliptic condim-3 contacts in all three
    zones. At every inspected state: M qacc - qfrc_smooth = J^T f (the Newton solver converged with the cone Hessian), the
    contact forces lie in the friction cone |f_t / mu_pad| <= f_n (elliptic cones are exact, whatever impratio), normal
    forces push, and sliding contacts sit ON the cone. Pad friction 0.7 / 0.6 has priority over the floor's 1.0."""
    b = _env(oracle_mod, n=4, seed=11, action_type=1)
    r = b.reset()
    rs = np.random.RandomState(0)
    seen_slide, seen_stick, seen_rows = 0, 0, 0
    tcp = r["obs"][:, 6:9].copy()
    for t in range(30):
        goal = tcp.copy()
        goal[:, 2] = -0.004                       # closed pads 4 mm into the floor
        goal[:, :2] += rs.uniform(-0.03, 0.03, (4, 2)) if t > 8 else 0.0
        diff = np.clip(goal - tcp, -0.05, 0.05)
        r = b.step(np.concatenate([tcp + diff, np.zeros((4, 1))], axis=1))
        tcp = r["obs"][:, 6:9].copy()
        qp, qv, _ = b.get_state()
        b.set_state(qp, qv)  # mj_forward: rows, forces and accelerations of ONE state (after a step the rows are the new state's, the forces the old one's)
        for i in range(4):
            e = b.efc(i)
            M, fs, qacc = b.dynamics(i)
            resid = M @ qacc - fs - e["J"].T @ e["force"]
            assert np.abs(resid).max() < 2e-5 * max(1.0, np.abs(fs).max()), (t, i, np.abs(resid).max())
            rows = np.where(e["type"] == ELLIPTIC)[0]
            for k in rows[::3]:
                fn, ft = e["force"][k], e["force"][k + 1:k + 3]
                mu = 0.7 if fn == 0 else None
                seen_rows += 1
                assert fn >= -1e-12
                ratio = np.linalg.norm(ft) / max(fn, 1e-300)
                assert ratio <= 0.7 + 1e-6, (t, i, ratio)   # pad_box1 0.7 / pad_box2 0.6 / arm capsule on floor 1.0 never on these rows' bodies... bounded by the largest pad value
                if fn > 1e-6:
                    seen_slide += ratio > 0.6 - 1e-6
                    seen_stick += ratio < 0.3
    assert seen_rows > 50 and seen_slide > 5 and seen_stick > 5, (seen_rows, seen_slide, seen_stick)


def test_scripted_policy_solves_with_the_articulated_gripper(oracle_mod):
    """robot_push_button.py:231-300: "add gripper, which is always closed" - the closed pads' ends are the TCP (0.174 m), so
    the policy's press (TCP 10 mm below the button's top) lands the pads on the button. Every episode succeeds with an odd number
    of switch flips; with the stand-in spheres of D-1b about half of them did."""
    import importlib.util
    from pathlib import Path

    spec = importlib.util.spec_from_file_location("make_golden", Path(__file__).parent / "golden" / "make_golden.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    N = 16
    b = _env(oracle_mod, n=N, seed=100, action_type=1, nthreads=8)
    r = b.reset()
    done, success, toggles, active = np.zeros(N, bool), np.zeros(N, bool), np.zeros(N, int), np.zeros(N, bool)
    for t in range(100):
        r = b.step(mod.demo_actions(r["obs"]))
        now = r["obs"][:, 12] > 0.5
        toggles += (now != active) & ~done
        active = now
        newly = (r["step_type"] == 2) & ~done
        won = newly & r["is_success"]
        assert (r["reward"][won] == 1.0).all() and (r["discount"][won] == 0.0).all() and r["terminated"][won].all()
        success |= won
        done |= newly
        assert not r["fault"].any()
    assert success.sum() >= N - 1, success.sum()
    assert (toggles[success] % 2 == 1).all()


@pytest.mark.parametrize("action_type", [0, 1])
def test_articulated_matches_its_golden(oracle_mod, action_type):
    """regression pin of the nv = 14 oracle (tests/golden/make_golden.py; not a reference output)"""
    from pathlib import Path

    name = {0: "button_push_art_joint_n8_t60_seed2025.npz", 1: "button_push_art_eef_n8_t80_seed2025.npz"}[action_type]
    fx = np.load(Path(__file__).parent / "golden" / name)
    N = fx["actions"].shape[1]
    kw = {"time_limit": 4.0} if action_type == 0 else {}  # the joint-action fixture truncates at 40 steps (make_golden.run_button_joint)
    b = oracle_mod.OracleBatch(oracle_mod.TASK_BUTTON_PUSH, N, 2025, action_type=action_type, gripper_model=1, **kw)
    r = b.reset()
    np.testing.assert_allclose(r["obs"], fx["reset_obs"], rtol=0, atol=1e-12)
    for t in range(fx["actions"].shape[0]):
        r = b.step(fx["actions"][t])
        np.testing.assert_allclose(r["obs"], fx["obs"][t], rtol=0, atol=1e-10, err_msg=f"step {t}")
        np.testing.assert_allclose(r["reward"], fx["reward"][t], rtol=0, atol=1e-10)
        for k in ("step_type", "terminated", "truncated", "is_success", "ncon"):
            assert np.array_equal(np.asarray(r[k]).astype(int), fx[k][t].astype(int)), (k, t)
