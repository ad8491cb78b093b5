"""collision pairs that neither the oracle nor the kernels evaluate. This file holds what can be SHOWN about them on
the CPU: the scene camera's collidable geoms (camera.py:78-88: a 0.09 x 0.025 x 0.025 box and a 12.5 mm lens sphere, static, at the
camera position (0, -1.1, 0.5) of Robot-Reach and Planar-Push, robot_reach.py:52,78 / robot_planar_push.py:43,70).
  * Planar-Push (UR5e + CylinderEEF): no collision geom of the arm or the tool reaches the camera in ANY joint configuration (bounding
    spheres, sampled + hill-climbed below: >= 3 cm clear), so the arm-camera pairs MuJoCo would list are vacuous.
  * Button-Push: the camera stands at (0, -1.7, 0.7) (robot_push_button.py:51-52), 0.63 m further away: vacuous for the gripper too.
  * Robot-Reach: the 2F-85 reaches 7 cm further than the cylinder tool, which the bound below does not cover; there the registered
    action space does (robot_reach.py:108,183-203: TCP targets in x +-0.1, y -0.6..-0.4, z 0.02..0.2, at least 0.58 m from the camera;
    the test asserts that arithmetic). An arm flung at the camera through mjs_set_state would pass through it: D-8."""
import numpy as np

CAM = np.array([0.0, -1.1, 0.5])          # FRONT_TILTED_CAMERA_CONFIG, the closest camera of the three robot scenes
CAM_BOUND = float(np.linalg.norm([0.045, 0.0125, 0.0125]))  # the box's half diagonal (the lens sphere lies inside it)


def _clearance(ob, q, n_geoms):
    """min over the arm's collision geoms + the tool of (distance of the geom's centre to the camera - the geom's bounding radius -
    the camera's), for the joint configurations q [n, 6] (a lower bound of the true clearance)."""
    n = q.shape[0]
    qp, qv, _ = ob.get_state()
    qp[:n, :6] = q
    qv[:] = 0
    ob.set_state(qp, qv)
    shapes = [ob.geom_shape(g) for g in range(1, n_geoms)]
    out = np.full(n, np.inf)
    for i in range(n):
        for g, (typ, _body, size) in zip(range(1, n_geoms), shapes):
            bound = size[0] + size[1] if typ == 3 else float(np.hypot(size[0], size[1]))  # capsule: half length + radius; cylinder: hypot
            pos, _ = ob.geom_pose(i, g)
            out[i] = min(out[i], np.linalg.norm(pos - CAM) - bound - CAM_BOUND)
    return out


def test_scene_camera_is_out_of_the_arms_reach(oracle_mod):
    N = 1024
    ob = oracle_mod.OracleBatch(oracle_mod.TASK_PLANAR_PUSH, N, 3, nthreads=8, block_shape=1)   # UR5e + CylinderEEF (TCP 0.1 m from the flange)
    ob.reset()
    n_geoms = 12  # floor, ten arm proxies, the CylinderEEF
    assert [ob.geom_shape(g)[0] for g in range(1, n_geoms)] == [3] * 9 + [5, 5]
    rs = np.random.RandomState(0)
    best_q, best = None, np.inf
    for _ in range(24):   # 24 k uniformly random configurations (all joints are periodic within their ranges)
        q = rs.uniform(-np.pi, np.pi, (N, 6))
        c = _clearance(ob, q, n_geoms)
        k = int(np.argmin(c))
        if c[k] < best:
            best, best_q = float(c[k]), q[k].copy()
    sigma = 0.3
    for _ in range(12):   # hill climbing from the closest one
        q = best_q + rs.normal(0, sigma, (N, 6))
        q[0] = best_q
        c = _clearance(ob, q, n_geoms)
        k = int(np.argmin(c))
        if c[k] < best:
            best, best_q = float(c[k]), q[k].copy()
        sigma *= 0.7
    assert best > 0.03, (best, best_q)
    ob.close()
    # Robot-Reach: the action box (robot_reach.py:108) against the camera; the gripper and the wrist lie within 0.3 m of the TCP
    box_lo, box_hi = np.array([-0.1, -0.6, 0.02]), np.array([0.1, -0.4, 0.2])
    nearest = np.clip(CAM, box_lo, box_hi)
    assert np.linalg.norm(nearest - CAM) - 0.3 - CAM_BOUND > 0.2
