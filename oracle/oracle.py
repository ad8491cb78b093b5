"""ctypes front-end of the C oracle (oracle/*.c). TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "liboracle.so"

TASK_POINTMASS, TASK_ROBOT_REACH, TASK_PLANAR_PUSH, TASK_BUTTON_PUSH = 0, 1, 2, 3
ACTION_ABS_JOINT, ACTION_ABS_EEF = 0, 1
STEP_FIRST, STEP_MID, STEP_LAST = 0, 1, 2
REW_SPARSE, REW_DENSE_POTENTIAL, REW_DENSE_NEG_DISTANCE, REW_DENSE_BIASED_NEG_DISTANCE = 0, 1, 2, 3
AUTORESET_NEXT_STEP, AUTORESET_SAME_STEP, AUTORESET_DISABLED = 0, 1, 2
MAXOBS = 16


class TaskConfig(C.Structure):
    _fields_ = [
        ("task", C.c_int),
        ("reward_type", C.c_int),
        ("autoreset", C.c_int),
        ("time_limit", C.c_double),
        ("terminate_on_success", C.c_int),
        ("action_type", C.c_int),
        ("button_disturbances", C.c_int),
        ("n_objects", C.c_int),
        ("max_episode_steps", C.c_int),
        ("block_shape", C.c_int),
        ("gripper_model", C.c_int),
    ]


class StepOut(C.Structure):
    _fields_ = [
        ("obs", C.c_double * MAXOBS),
        ("terminal_obs", C.c_double * MAXOBS),
        ("reward", C.c_double),
        ("discount", C.c_double),
        ("step_type", C.c_int),
        ("terminated", C.c_int),
        ("truncated", C.c_int),
        ("is_success", C.c_int),
        ("ncon", C.c_int),
        ("fault", C.c_int),
        ("ik_failed", C.c_int),
    ]


_STEP_DTYPE = np.dtype(
    {
        "names": ["obs", "terminal_obs", "reward", "discount", "step_type", "terminated", "truncated", "is_success", "ncon", "fault", "ik_failed"],
        "formats": [(np.float64, MAXOBS), (np.float64, MAXOBS), np.float64, np.float64] + [np.int32] * 7,
        "offsets": [0, 8 * MAXOBS, 16 * MAXOBS, 16 * MAXOBS + 8] + [16 * MAXOBS + 16 + 4 * i for i in range(7)],
        "itemsize": C.sizeof(StepOut),
    }
)


class _Rng(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("pos", C.c_int)]


def build(force: bool = False) -> Path:
    """Compile oracle/*.c with gcc (Makefile in this directory)."""
    srcs = [*_HERE.glob("*.c"), _HERE / "mjs_oracle.h", _HERE.parent / "include" / "mjs_scene_spec.h"]
    stale = not _LIB_PATH.exists() or any(s.stat().st_mtime > _LIB_PATH.stat().st_mtime for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", str(_HERE), "-B" if force else "-s"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists() or os.access(_HERE / "Makefile", os.R_OK) and _sources_newer():
            build()
        L = C.CDLL(str(_LIB_PATH))
        L.om_default_config.argtypes = [C.c_int, C.POINTER(TaskConfig)]
        L.om_obs_dim.argtypes = [C.c_int]
        L.om_action_dim.argtypes = [C.c_int]
        L.om_batch_create.argtypes = [C.POINTER(TaskConfig), C.c_int, C.c_uint32]
        L.om_batch_create.restype = C.c_void_p
        L.om_batch_destroy.argtypes = [C.c_void_p]
        L.om_batch_env.argtypes = [C.c_void_p, C.c_int]
        L.om_batch_env.restype = C.c_void_p
        L.om_batch_reset.argtypes = [C.c_void_p, C.c_void_p]
        L.om_batch_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.om_env_seed.argtypes = [C.c_void_p, C.c_uint32]
        L.om_render_pointmass.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.om_render_robot.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.om_render_camera.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.om_rng_seed.argtypes = [C.POINTER(_Rng), C.c_uint32]
        L.om_rng_uniform.argtypes = [C.POINTER(_Rng), C.c_double, C.c_double]
        L.om_rng_uniform.restype = C.c_double
        L.om_rng_u32.argtypes = [C.POINTER(_Rng)]
        L.om_rng_u32.restype = C.c_uint32
        L.om_ur5e_fk_dh.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.om_ur5e_ik_all.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.om_ur5e_ik_closest.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        assert L.om_sizeof_step_out() == C.sizeof(StepOut), "StepOut layout mismatch"
        _lib = L
    return _lib


def _sources_newer() -> bool:
    if not _LIB_PATH.exists():
        return True
    t = _LIB_PATH.stat().st_mtime
    return any(s.stat().st_mtime > t for s in [*_HERE.glob("*.c"), *_HERE.glob("*.h")])


def _dptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class OracleRng:
    """numpy-legacy MT19937 stream of the oracle (for pinning against numpy)."""

    def __init__(self, seed: int):
        self._r = _Rng()
        lib().om_rng_seed(C.byref(self._r), seed & 0xFFFFFFFF)

    def uniform(self, lo: float, hi: float) -> float:
        return lib().om_rng_uniform(C.byref(self._r), lo, hi)

    def u32(self) -> int:
        return lib().om_rng_u32(C.byref(self._r))


def ur5e_fk_dh(q) -> np.ndarray:
    q = np.ascontiguousarray(q, dtype=np.float64)
    T = np.zeros(16)
    lib().om_ur5e_fk_dh(_dptr(q), _dptr(T))
    return T.reshape(4, 4)


def ur5e_ik_all(T) -> np.ndarray:
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    sols = np.zeros((8, 6))
    n = lib().om_ur5e_ik_all(_dptr(T), _dptr(sols))
    return sols[:n]


def ur5e_ik_closest(T, q_guess):
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
    g = np.ascontiguousarray(q_guess, dtype=np.float64)
    out = np.zeros(6)
    ok = lib().om_ur5e_ik_closest(_dptr(T), _dptr(g), _dptr(out))
    return out if ok else None


BLOCKS_MESH, BLOCKS_BOX = 0, 1
UR_STATE = 34
UR_CMD_NONE, UR_CMD_MOVEJ, UR_CMD_MOVEJ_IK, UR_CMD_SERVOL, UR_CMD_SERVOJ = 0, 1, 2, 3, 4
UR_EEF_NONE, UR_EEF_GRIPPER = 0, 1


def ur_robot_state(q, ctrl=None):
    """state block of a stand-alone UR5e at rest at q (Robot.set_joint_positions: ctrl = q; model creation: ctrl = 0)"""
    st = np.zeros(UR_STATE)
    st[0:6] = q
    st[12:18] = q if ctrl is None else ctrl
    return st


def ur_robot_run(state, target, command, param, n_substeps, eef=UR_EEF_NONE, dt=0.002):
    """Robot control API + n x (before_substep; Physics.step) on the oracle; returns (new state, tcp pose[7], ok)"""
    L = lib()
    L.om_ur_robot_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, C.c_void_p]
    st = np.array(state, dtype=np.float64)
    tgt = np.zeros(7)
    tgt[: len(target)] = target
    pose = np.zeros(7)
    ok = L.om_ur_robot_run(st.ctypes.data, tgt.ctypes.data, command, float(param), int(n_substeps), int(eef), float(dt), pose.ctypes.data)
    return st, pose, bool(ok)


class OracleBatch:
    """N independent oracle envs, env i seeded ``base_seed + i`` (reach_sac.py:84)."""

    def __init__(self, task: int, n: int, base_seed: int = 0, *, reward_type: int | None = None,
                 autoreset: int = AUTORESET_NEXT_STEP, time_limit: float | None = None,
                 terminate_on_success: bool = False, nthreads: int = 1, action_type: int | None = None,
                 button_disturbances: bool = False, n_objects: int | None = None, max_episode_steps: int | None = None,
                 block_shape: int | None = None, gripper_model: int = 0):
        L = lib()
        cfg = TaskConfig()
        L.om_default_config(task, C.byref(cfg))
        if reward_type is not None:
            cfg.reward_type = reward_type
        if time_limit is not None:
            cfg.time_limit = time_limit
        cfg.autoreset = autoreset
        cfg.terminate_on_success = int(terminate_on_success)
        if action_type is not None:
            cfg.action_type = action_type
        cfg.button_disturbances = int(button_disturbances)
        if n_objects is not None:
            cfg.n_objects = n_objects
        if max_episode_steps is not None:
            cfg.max_episode_steps = max_episode_steps
        if block_shape is not None:
            cfg.block_shape = block_shape
        cfg.gripper_model = int(gripper_model)
        self.cfg, self.task, self.n, self.nthreads = cfg, task, n, nthreads
        L.om_obs_dim_for.argtypes = [C.c_void_p]
        self.obs_dim, self.action_dim = L.om_obs_dim_for(C.byref(cfg)), L.om_action_dim(task)
        if task == TASK_BUTTON_PUSH and cfg.action_type == ACTION_ABS_EEF:
            self.action_dim = 4
        self._h = L.om_batch_create(C.byref(cfg), n, base_seed & 0xFFFFFFFF)
        self._out = np.zeros(n, dtype=_STEP_DTYPE)

    def seed(self, base_seed: int):
        for i in range(self.n):
            lib().om_env_seed(lib().om_batch_env(self._h, i), (base_seed + i) & 0xFFFFFFFF)

    def _result(self):
        o = self._out
        return {
            "obs": o["obs"][:, : self.obs_dim].copy(),
            "terminal_obs": o["terminal_obs"][:, : self.obs_dim].copy(),
            "reward": o["reward"].copy(),
            "discount": o["discount"].copy(),
            "step_type": o["step_type"].copy(),
            "terminated": o["terminated"].astype(bool),
            "truncated": o["truncated"].astype(bool),
            "is_success": o["is_success"].astype(bool),
            "ncon": o["ncon"].copy(),
            "fault": o["fault"].astype(bool),
            "ik_failed": o["ik_failed"].astype(bool),
        }

    def reset(self):
        lib().om_batch_reset(self._h, self._out.ctypes.data)
        return self._result()

    def reset_envs(self, indices):
        """reset some envs now (what mjs_reset does with a mask); the other envs' last results stay in the returned dict"""
        L = lib()
        L.om_env_reset.argtypes = [C.c_void_p, C.c_void_p]
        L.om_env_reset.restype = None
        for i in indices:
            L.om_env_reset(L.om_batch_env(self._h, int(i)), self._out[int(i):int(i) + 1].ctypes.data)
        return self._result()

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float64).reshape(self.n, self.action_dim)
        lib().om_batch_step(self._h, a.ctypes.data, self._out.ctypes.data, self.nthreads)
        return self._result()

    def set_robot_state(self, q, v):
        """debug: overwrite joint positions/velocities of every robot env ([N,6] each)"""
        q = np.ascontiguousarray(q, dtype=np.float64); v = np.ascontiguousarray(v, dtype=np.float64)
        L = lib()
        L.om_debug_set_robot_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        for i in range(self.n):
            L.om_debug_set_robot_state(L.om_batch_env(self._h, i), q[i].ctypes.data, v[i].ctypes.data)

    def set_block_shapes(self, cats, colors, scales):
        """debug (Planar-Push): block b of env i becomes mesh category cats[i, b] at scales[i, b] ([N, n_objects] each)"""
        L = lib()
        L.om_debug_set_block_shape.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double]
        cats, colors, scales = np.asarray(cats), np.asarray(colors), np.asarray(scales, dtype=np.float64)
        for i in range(self.n):
            for b in range(cats.shape[1]):
                L.om_debug_set_block_shape(L.om_batch_env(self._h, i), b, int(cats[i, b]), int(colors[i, b]), float(scales[i, b]))

    def block_shapes(self):
        """debug (Planar-Push): (category, colour, scale) arrays [N, 5] of the current episode"""
        import struct  # noqa: F401
        L = lib()
        L.om_debug_get_block_shape.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        cat, col, sc = np.zeros((self.n, 5), np.int32), np.zeros((self.n, 5), np.int32), np.zeros((self.n, 5))
        for i in range(self.n):
            L.om_debug_get_block_shape(L.om_batch_env(self._h, i), cat[i].ctypes.data, col[i].ctypes.data, sc[i].ctypes.data)
        return cat, col, sc

    def arm_floor_seen(self) -> np.ndarray:
        """debug: per env, did an arm link touch the floor in any substep since the last call (sticky flag, cleared)"""
        L = lib()
        L.om_debug_arm_floor_seen.argtypes = [C.c_void_p]
        return np.array([bool(L.om_debug_arm_floor_seen(L.om_batch_env(self._h, i))) for i in range(self.n)])

    def get_state(self):
        """debug: (qpos [N, nq], qvel [N, nv], time [N]) of every env"""
        L = lib()
        L.om_debug_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        qp, qv, tm = np.zeros((self.n, 48)), np.zeros((self.n, 36)), np.zeros(self.n)
        nq = 0
        for i in range(self.n):
            nq = L.om_debug_get_state(L.om_batch_env(self._h, i), qp[i].ctypes.data, qv[i].ctypes.data, tm[i:].ctypes.data)
        nv = self.model_dims()["nv"]
        return qp[:, :nq].copy(), qv[:, :nv].copy(), tm

    # ---- debug hooks of the articulated-gripper tests
    def model_dims(self):
        L = lib()
        L.om_debug_model_dims.argtypes = [C.c_void_p, C.c_void_p]
        out = np.zeros(8, np.int32)
        L.om_debug_model_dims(L.om_batch_env(self._h, 0), out.ctypes.data)
        return dict(zip(("nq", "nv", "nbody", "ngeom", "neq", "nu", "njnt", "cone"), out.tolist()))

    def efc(self, i: int, maxrows: int = 128):
        """constraint rows of env i's current state: dict(pos, J [nefc, nv], type, force, aref, D)"""
        L = lib()
        L.om_debug_efc.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6
        nv = self.model_dims()["nv"]
        pos, J, typ = np.zeros(maxrows), np.zeros((maxrows, nv)), np.zeros(maxrows, np.int32)
        force, aref, D = np.zeros(maxrows), np.zeros(maxrows), np.zeros(maxrows)
        n = L.om_debug_efc(L.om_batch_env(self._h, i), maxrows, pos.ctypes.data, J.ctypes.data, typ.ctypes.data, force.ctypes.data, aref.ctypes.data, D.ctypes.data)
        n = min(n, maxrows)
        return {"pos": pos[:n], "J": J[:n], "type": typ[:n], "force": force[:n], "aref": aref[:n], "D": D[:n]}

    def geom_pose(self, i: int, g: int):
        L = lib()
        L.om_debug_geom_pose.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        pos, mat = np.zeros(3), np.zeros(9)
        L.om_debug_geom_pose(L.om_batch_env(self._h, i), g, pos.ctypes.data, mat.ctypes.data)
        return pos, mat.reshape(3, 3)

    def geom_shape(self, g: int):
        """debug: (type, body, size[3]) of geom g of the model (OM_GEOM_* codes: 0 plane, 2 sphere, 3 capsule, 5 cylinder, 6 box, 7 mesh)"""
        L = lib()
        L.om_debug_geom_shape.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        tb, size = np.zeros(2, np.int32), np.zeros(3)
        L.om_debug_geom_shape(L.om_batch_env(self._h, 0), g, tb.ctypes.data, size.ctypes.data)
        return int(tb[0]), int(tb[1]), size

    def set_ctrl(self, u: int, values):
        L = lib()
        L.om_debug_set_ctrl.argtypes = [C.c_void_p, C.c_int, C.c_double]
        v = np.broadcast_to(np.asarray(values, dtype=np.float64), (self.n,))
        for i in range(self.n):
            L.om_debug_set_ctrl(L.om_batch_env(self._h, i), u, float(v[i]))

    def dynamics(self, i: int):
        L = lib()
        L.om_debug_get_dynamics.argtypes = [C.c_void_p] * 4
        nv = self.model_dims()["nv"]
        M, f, a = np.zeros((nv, nv)), np.zeros(nv), np.zeros(nv)
        L.om_debug_get_dynamics(L.om_batch_env(self._h, i), M.ctypes.data, f.ctypes.data, a.ctypes.data)
        return M, f, a

    def get_gripper(self):
        """Button-Push: (driver angle, driver velocity) of the reduced 2F-85 per env, [N, 2]."""
        L = lib()
        L.om_debug_get_gripper.argtypes = [C.c_void_p, C.c_void_p]
        L.om_debug_get_gripper.restype = None
        L.om_batch_env.restype = C.c_void_p
        L.om_batch_env.argtypes = [C.c_void_p, C.c_int]
        out = np.zeros((self.n, 2))
        for i in range(self.n):
            L.om_debug_get_gripper(L.om_batch_env(self._h, i), out[i].ctypes.data)
        return out

    def set_gripper(self, theta_vel):
        L = lib()
        L.om_debug_set_gripper.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.om_debug_set_gripper.restype = None
        L.om_batch_env.restype = C.c_void_p
        L.om_batch_env.argtypes = [C.c_void_p, C.c_int]
        tv = np.asarray(theta_vel, dtype=np.float64).reshape(self.n, 2)
        for i in range(self.n):
            L.om_debug_set_gripper(L.om_batch_env(self._h, i), float(tv[i, 0]), float(tv[i, 1]))

    def set_state(self, qpos, qvel):
        """debug: overwrite qpos [N, nq] / qvel [N, nv] of every env (ctrl = arm joints), then mj_forward"""
        L = lib()
        L.om_debug_set_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        qpos = np.ascontiguousarray(qpos, dtype=np.float64); qvel = np.ascontiguousarray(qvel, dtype=np.float64)
        for i in range(self.n):
            L.om_debug_set_state(L.om_batch_env(self._h, i), qpos[i].ctypes.data, qvel[i].ctypes.data)

    def substeps(self, n: int):
        """debug: n raw Physics.step() of every env with the current ctrl"""
        L = lib()
        L.om_debug_substeps.argtypes = [C.c_void_p, C.c_int]
        for i in range(self.n):
            L.om_debug_substeps(L.om_batch_env(self._h, i), n)

    def render(self, height: int, width: int, camera: int = 0) -> np.ndarray:
        """camera images of all envs (0 = scene camera, 1 = Button-Push wrist camera): uint8 [N, H, W, 3]"""
        img = np.zeros((self.n, height, width, 3), dtype=np.uint8)
        for i in range(self.n):
            lib().om_render_camera(lib().om_batch_env(self._h, i), camera, height, width, img[i].ctypes.data)
        return img

    def close(self):
        if self._h:
            lib().om_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
