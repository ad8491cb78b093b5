"""ctypes binding of libmjsim.so (C ABI in include/mjsim.h) and its in-tree hipcc build.

There is NO CPU fallback: if the shared library is missing or no HIP device is visible the
product path raises. The oracle under ``oracle/`` is test infrastructure and is never imported
from here.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

_PKG = Path(__file__).resolve().parent
_ROOT = _PKG.parent
LIB_PATH = Path(os.environ["MJS_LIB"]) if os.environ.get("MJS_LIB") else _PKG / "lib" / "libmjsim.so"  # MJS_LIB: diagnostic builds
_SOURCES = [*sorted((_PKG / "csrc").glob("*")), _ROOT / "include" / "mjsim.h", _ROOT / "include" / "mjs_scene_spec.h"]

TASK_POINTMASS_REACH, TASK_ROBOT_REACH, TASK_PLANAR_PUSH, TASK_BUTTON_PUSH = 0, 1, 2, 3
ACTION_ABS_JOINT, ACTION_ABS_EEF = 0, 1
REW_SPARSE, REW_DENSE_POTENTIAL, REW_DENSE_NEG_DISTANCE, REW_DENSE_BIASED_NEG_DISTANCE = 0, 1, 2, 3
STEP_FIRST, STEP_MID, STEP_LAST = 0, 1, 2
AUTORESET_NEXT_STEP, AUTORESET_SAME_STEP, AUTORESET_DISABLED = 0, 1, 2
FAULT_BAD_STATE, FAULT_IK_FAILED, FAULT_LIMIT_COLDSTART = 1, 2, 4

EXPORTED_SYMBOLS = [
    "mjs_version", "mjs_obs_dim", "mjs_action_dim", "mjs_action_dim_for", "mjs_state_dim", "mjs_env_obs_dim", "mjs_env_state_dim", "mjs_algorithmic_bytes_per_env_step", "mjs_substeps",
    "mjs_create", "mjs_destroy", "mjs_last_error", "mjs_seed", "mjs_reset", "mjs_step", "mjs_rollout",
    "mjs_get_state", "mjs_set_state", "mjs_get_rng_state", "mjs_set_rng_state", "mjs_render", "mjs_debug_ur5e_ik", "mjs_ur5e_tcp_to_joints",
]


class MjsConfig(C.Structure):
    _fields_ = [
        ("task", C.c_int32), ("num_envs", C.c_int32), ("device", C.c_int32), ("reward_type", C.c_int32),
        ("autoreset", C.c_int32), ("terminate_on_success", C.c_int32), ("env_index_offset", C.c_int32),
        ("kernel_variant", C.c_int32), ("time_limit", C.c_double), ("action_type", C.c_int32), ("button_disturbances", C.c_int32), ("n_objects", C.c_int32), ("max_episode_steps", C.c_int32),
    ]


class MjsOutputs(C.Structure):
    _fields_ = [
        ("obs", C.c_void_p), ("terminal_obs", C.c_void_p), ("reward", C.c_void_p), ("discount", C.c_void_p),
        ("terminated", C.c_void_p), ("truncated", C.c_void_p), ("is_success", C.c_void_p), ("step_type", C.c_void_p),
        ("fault", C.c_void_p), ("ncon", C.c_void_p),
    ]


class MjsError(RuntimeError):
    pass


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile mujoco_sim_amd/csrc/mjsim.hip for gfx950 into mujoco_sim_amd/lib/libmjsim.so."""
    if not force and LIB_PATH.exists() and all(s.stat().st_mtime <= LIB_PATH.stat().st_mtime for s in _SOURCES):
        return LIB_PATH
    LIB_PATH.parent.mkdir(parents=True, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # max-ilp: the kernels run one wavefront per SIMD, so schedule for ILP, not occupancy (+2% measured)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-comment",
           "-mllvm", "-amdgpu-sched-strategy=max-ilp",
           "-o", str(LIB_PATH), str(_PKG / "csrc" / "mjsim.hip")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise MjsError(f"hipcc failed:\n{res.stderr}")
    if verbose:
        print(res.stderr)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """Load libmjsim.so (building it first if the sources are newer). Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        try:
            build()
        except Exception as e:  # noqa: BLE001
            raise MjsError(f"libmjsim.so is missing and could not be built ({e}); there is no CPU fallback") from e
    L = C.CDLL(str(LIB_PATH))
    L.mjs_version.restype = C.c_char_p
    for name in ("mjs_obs_dim", "mjs_action_dim", "mjs_state_dim", "mjs_algorithmic_bytes_per_env_step", "mjs_substeps"):
        getattr(L, name).argtypes = [C.c_int]
        getattr(L, name).restype = C.c_int
    for name in ("mjs_env_obs_dim", "mjs_env_state_dim"):
        getattr(L, name).argtypes = [C.c_void_p]
        getattr(L, name).restype = C.c_int
    L.mjs_action_dim_for.argtypes = [C.c_int, C.c_int]
    L.mjs_action_dim_for.restype = C.c_int
    L.mjs_create.argtypes = [C.POINTER(MjsConfig), C.POINTER(C.c_void_p)]
    L.mjs_destroy.argtypes = [C.c_void_p]
    L.mjs_destroy.restype = None
    L.mjs_last_error.argtypes = [C.c_void_p]
    L.mjs_last_error.restype = C.c_char_p
    L.mjs_seed.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.mjs_reset.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(MjsOutputs), C.c_void_p]
    L.mjs_step.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(MjsOutputs), C.c_void_p]
    L.mjs_rollout.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(MjsOutputs), C.c_void_p]
    L.mjs_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.mjs_set_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.mjs_debug_ur5e_ik.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.mjs_ur5e_tcp_to_joints.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.mjs_render.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.mjs_get_rng_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mjs_set_rng_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = L
    return L


def check(rc: int, handle=None):
    if rc != 0:
        msg = lib().mjs_last_error(handle)
        raise MjsError(f"libmjsim error {rc}: {msg.decode() if msg else '?'}")
