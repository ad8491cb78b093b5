"""ctypes binding of libmjsim.so (C ABI in include/mjsim.h) and its in-tree hipcc build.

There is NO CPU fallback: if the shared library is missing or no HIP device is visible the
product path raises. The oracle under ``oracle/`` is test infrastructure and is never imported
from here.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import re
import subprocess
from pathlib import Path

_PKG = Path(__file__).resolve().parent
_ROOT = _PKG.parent
LIB_PATH = Path(os.environ["MJS_LIB"]) if os.environ.get("MJS_LIB") else _PKG / "lib" / "libmjsim.so"  # MJS_LIB: diagnostic builds


def _sources() -> list[Path]:
    """Every file the library is compiled from: csrc/*.{h,hip} and include/*.h."""
    return [*sorted(p for p in (_PKG / "csrc").glob("*") if p.suffix in (".h", ".hip")), *sorted((_ROOT / "include").glob("*.h"))]


TASK_POINTMASS_REACH, TASK_ROBOT_REACH, TASK_PLANAR_PUSH, TASK_BUTTON_PUSH = 0, 1, 2, 3
ACTION_ABS_JOINT, ACTION_ABS_EEF = 0, 1
REW_SPARSE, REW_DENSE_POTENTIAL, REW_DENSE_NEG_DISTANCE, REW_DENSE_BIASED_NEG_DISTANCE = 0, 1, 2, 3
STEP_FIRST, STEP_MID, STEP_LAST = 0, 1, 2
AUTORESET_NEXT_STEP, AUTORESET_SAME_STEP, AUTORESET_DISABLED = 0, 1, 2
FAULT_BAD_STATE, FAULT_IK_FAILED, FAULT_LIMIT_COLDSTART, FAULT_UNSUPPORTED_CONTACT, FAULT_FASTPATH_VIOLATED = 1, 2, 4, 8, 16
BLOCKS_MESH, BLOCKS_BOX = 0, 1
GRIPPER_REDUCED, GRIPPER_ARTICULATED = 0, 1
VARIANT_DEFAULT, VARIANT_SINGLE_WAVE, VARIANT_TWO_ROLES, VARIANT_RESET_GROUPS = 0, 1, 2, 3
UR_STATE = 34
UR_CMD_NONE, UR_CMD_MOVEJ, UR_CMD_MOVEJ_IK, UR_CMD_SERVOL, UR_CMD_SERVOJ = 0, 1, 2, 3, 4
UR_EEF_NONE, UR_EEF_GRIPPER = 0, 1

EXPORTED_SYMBOLS = [
    "mjs_version", "mjs_obs_dim", "mjs_action_dim", "mjs_action_dim_for", "mjs_state_dim", "mjs_env_obs_dim", "mjs_env_state_dim", "mjs_algorithmic_bytes_per_env_step", "mjs_substeps",
    "mjs_create", "mjs_destroy", "mjs_last_error", "mjs_seed", "mjs_reset", "mjs_step", "mjs_rollout",
    "mjs_get_state", "mjs_set_state", "mjs_get_rng_state", "mjs_set_rng_state", "mjs_render", "mjs_debug_ur5e_ik", "mjs_ur5e_tcp_to_joints", "mjs_ur5e_robot_run",
]


class MjsConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("task", C.c_int32), ("num_envs", C.c_int32), ("device", C.c_int32), ("reward_type", C.c_int32),
        ("autoreset", C.c_int32), ("terminate_on_success", C.c_int32), ("env_index_offset", C.c_int32),
        ("kernel_variant", C.c_int32), ("time_limit", C.c_double), ("action_type", C.c_int32), ("button_disturbances", C.c_int32), ("n_objects", C.c_int32), ("max_episode_steps", C.c_int32), ("block_shape", C.c_int32),
        ("gripper_model", C.c_int32), ("reserved0", C.c_int32),
    ]

    def __init__(self, *args, **kw):  # struct_size = sizeof(mjs_config) of THIS binding: mjs_create refuses a mismatch (abi 2+)
        super().__init__(*args, **kw)
        if "struct_size" not in kw:
            self.struct_size = C.sizeof(MjsConfig)


class MjsOutputs(C.Structure):
    _fields_ = [
        ("obs", C.c_void_p), ("terminal_obs", C.c_void_p), ("reward", C.c_void_p), ("discount", C.c_void_p),
        ("terminated", C.c_void_p), ("truncated", C.c_void_p), ("is_success", C.c_void_p), ("step_type", C.c_void_p),
        ("fault", C.c_void_p), ("ncon", C.c_void_p),
    ]


class MjsError(RuntimeError):
    pass


def source_hash() -> str:
    """sha256 over the names and bytes of every file the library is compiled from (csrc/ + the two headers). The build
    embeds it (``-DMJS_SOURCE_HASH``) and ``mjs_version()`` reports it, so a binary that does not belong to the sources
    next to it is detected whatever the file times say (a copied tree, a git checkout, the gpurun snapshot)."""
    h = hashlib.sha256()
    for s in _sources():
        h.update(s.name.encode() + b"\0" + s.read_bytes() + b"\0")
    return h.hexdigest()[:16]


def built_hash(path: Path | None = None) -> str | None:
    """The source hash embedded in an existing library (None if the file is absent or carries none)."""
    path = Path(path or LIB_PATH)
    if not path.exists():
        return None
    m = re.search(rb"mjsim-hip [^\0]* src=([0-9a-f]{16}|unhashed)", path.read_bytes())
    return m.group(1).decode() if m else None


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile mujoco_sim_amd/csrc/mjsim.hip for gfx950 into mujoco_sim_amd/lib/libmjsim.so (skipped when the existing
    library already embeds the hash of the current sources)."""
    want = source_hash()
    if not force and built_hash() == want:
        return LIB_PATH
    LIB_PATH.parent.mkdir(parents=True, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # max-ilp: the kernels run one wavefront per SIMD, so schedule for ILP, not occupancy (+2% measured).
    # enable-ipra=0: with inter-procedural register allocation the call to the (rare, register-hungry) robust path clobbers every
    # AGPR in the step kernels' eyes, and their hot loops then spill to scratch instead of to AGPRs: Robot-Reach 39.7 -> 36.8 us,
    # Button-Push 110.9 -> 93.9 us per launch (profiles/r03_c_ipra_ab.txt); Planar-Push / Pointmass unchanged.
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-comment", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-enable-ipra=0",
             f'-DMJS_SOURCE_HASH="{want}"']
    src = str(_PKG / "csrc" / "mjsim.hip")
    # the same compilation once more, to device assembly, for the ISA lint (_isa_lint.py: a code-generation fault of this
    # compiler that the flags above provoke around out-of-line calls); both run side by side
    asm_path = LIB_PATH.with_suffix(".gfx950.s")
    asm = subprocess.Popen([hipcc, *flags, "-S", "--cuda-device-only", "-Wno-unused-command-line-argument", "-o", str(asm_path), src],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    res = subprocess.run([hipcc, *flags, "-fPIC", "-shared", "-o", str(LIB_PATH), src], capture_output=True, text=True)
    asm_err = asm.communicate()[1]
    if res.returncode != 0 or asm.returncode != 0:
        LIB_PATH.unlink(missing_ok=True)
        raise MjsError(f"hipcc failed:\n{res.stderr}\n{asm_err}")
    from ._isa_lint import lint
    findings = lint(asm_path.read_text().split("\n"))
    asm_path.unlink(missing_ok=True)
    if findings:
        LIB_PATH.unlink(missing_ok=True)
        raise MjsError("ISA lint: register copies ahead of an exec-mask restore (see mujoco_sim_amd/_isa_lint.py): "
                       + "; ".join(f"{fn} {label}: {copies[:2]}" for fn, label, copies, _ in findings))
    if verbose:
        print(res.stderr)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """Load libmjsim.so. On the first load the hash embedded in the binary is compared with the hash of the sources
    next to it: a missing or stale library is rebuilt, and if that is impossible the load FAILS (no stale binary is
    ever used silently, and there is no CPU fallback). MJS_LIB (diagnostic builds) bypasses the check."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.environ.get("MJS_LIB") and built_hash() != source_hash():
        try:
            build()
        except Exception as e:  # noqa: BLE001
            raise MjsError(f"libmjsim.so is missing or older than its sources (embedded {built_hash()}, sources "
                           f"{source_hash()}) and could not be rebuilt ({e}); there is no CPU fallback") from e
    L = C.CDLL(str(LIB_PATH))
    L.mjs_version.restype = C.c_char_p
    if not os.environ.get("MJS_LIB") and f"src={source_hash()}".encode() not in L.mjs_version():
        raise MjsError(f"loaded {LIB_PATH} reports {L.mjs_version()!r}, sources hash to {source_hash()}")
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
    L.mjs_ur5e_robot_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.mjs_render.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.mjs_get_rng_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mjs_set_rng_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = L
    return L


def check(rc: int, handle=None):
    if rc != 0:
        msg = lib().mjs_last_error(handle)
        raise MjsError(f"libmjsim error {rc}: {msg.decode() if msg else '?'}")
