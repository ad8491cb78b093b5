"""C-ABI checks that need no GPU: the library builds, loads and exports every symbol
include/mjsim.h declares; on a GPU-less host the product fails loudly (no CPU fallback)."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def native():
    from mujoco_sim_amd import _native

    _native.build()
    return _native


def test_library_exports_every_declared_symbol(native):
    header = (ROOT / "include" / "mjsim.h").read_text()
    declared = set(re.findall(r"\b(mjs_[a-z0-9_]+)\s*\(", header))
    assert declared == set(native.EXPORTED_SYMBOLS), declared ^ set(native.EXPORTED_SYMBOLS)
    L = C.CDLL(str(native.LIB_PATH))
    for name in sorted(declared):
        assert hasattr(L, name), f"libmjsim.so does not export {name}"


def test_loaded_library_matches_its_sources(native, tmp_path, monkeypatch):
    """Stale-binary guard: the library embeds the sha256 of csrc/ + include/ it was compiled from; the loader compares
    it with the sources next to it on first load, rebuilds on mismatch and raises when it cannot."""
    L = native.lib()
    assert f"src={native.source_hash()}".encode() in L.mjs_version()
    assert native.built_hash() == native.source_hash()
    # a binary whose embedded hash differs from the sources is never loaded silently: with no compiler, lib() raises
    stale = tmp_path / "libmjsim.so"
    stale.write_bytes(native.LIB_PATH.read_bytes().replace(native.source_hash().encode(), b"0123456789abcdef"))
    assert native.built_hash(stale) == "0123456789abcdef"
    monkeypatch.setattr(native, "LIB_PATH", stale)
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setenv("HIPCC", "/nonexistent/hipcc")
    with pytest.raises(native.MjsError, match="older than its sources"):
        native.lib()


def test_static_queries(native):
    L = native.lib()
    assert L.mjs_version().startswith(b"mjsim-hip")
    assert (L.mjs_obs_dim(0), L.mjs_action_dim(0), L.mjs_substeps(0)) == (4, 2, 5)
    assert (L.mjs_obs_dim(1), L.mjs_action_dim(1), L.mjs_substeps(1)) == (12, 3, 20)
    assert (L.mjs_obs_dim(2), L.mjs_action_dim(2), L.mjs_substeps(2), L.mjs_state_dim(2)) == (9, 2, 20, 2 * 47 + 1 + 6 + 12 + 1)   # Planar-Push, 2 block slots: current world + prepared next episode (+ progress, set-point, cos / sin) + flags
    assert (L.mjs_obs_dim(3), L.mjs_action_dim(3), L.mjs_substeps(3)) == (13, 7, 20)                            # Button-Push
    assert (L.mjs_action_dim_for(3, 0), L.mjs_action_dim_for(3, 1), L.mjs_action_dim_for(3, 2), L.mjs_action_dim_for(1, 5)) == (7, 4, -1, 3)
    assert L.mjs_obs_dim(99) == -1
    # algorithmic bytes per env-step, recomputed from the SoA layout
    assert L.mjs_algorithmic_bytes_per_env_step(0) == 8 * 13 + 8 * 11 + 2 + 16 + 32 + 25
    # Robot-Reach: q6 v6 time target3 + the carried cos6 sin6 read, the same minus the target written (the qacc_warmstart rows are
    # touched by the robust path only)
    assert L.mjs_algorithmic_bytes_per_env_step(1) == 8 * (16 + 12) + 8 * (13 + 12) + 2 + 24 + 96 + 25
    # state blocks (rows + the flag row): Robot-Reach q6 v6 time target3 qacc_warmstart6 cos6 sin6; Button-Push q6 v6 time switch3
    # gripper2 qacc_warmstart6 cos6 sin6
    assert (L.mjs_state_dim(1), L.mjs_state_dim(3)) == (35, 37)


def test_create_rejects_bad_arguments(native):
    L = native.lib()
    h = C.c_void_p()
    cfg = native.MjsConfig(task=7, num_envs=4, device=0, reward_type=-1, autoreset=0)
    assert L.mjs_create(C.byref(cfg), C.byref(h)) == -5  # MJS_ERR_UNSUPPORTED
    cfg = native.MjsConfig(task=0, num_envs=0, device=0, reward_type=-1, autoreset=0)
    assert L.mjs_create(C.byref(cfg), C.byref(h)) == -1  # MJS_ERR_INVALID_ARG
    assert L.mjs_create(None, C.byref(h)) == -1
    cfg = native.MjsConfig(task=3, num_envs=4, device=0, reward_type=-1, autoreset=0, action_type=9)
    assert L.mjs_create(C.byref(cfg), C.byref(h)) == -1  # bad action_type
    cfg = native.MjsConfig(task=2, num_envs=4, device=0, reward_type=-1, autoreset=0, n_objects=6)
    assert L.mjs_create(C.byref(cfg), C.byref(h)) == -1 and b"n_objects" in L.mjs_last_error(None)
    cfg = native.MjsConfig(task=1, num_envs=4, device=0, reward_type=-1, autoreset=7)
    assert L.mjs_create(C.byref(cfg), C.byref(h)) == -1 and b"autoreset" in L.mjs_last_error(None)
    # abi 2: a caller built against another header (different mjs_config) is refused before any field is read
    cfg = native.MjsConfig(task=1, num_envs=4, device=0, reward_type=-1, autoreset=0, struct_size=C.sizeof(native.MjsConfig) - 4)
    assert L.mjs_create(C.byref(cfg), C.byref(h)) == -1 and b"struct_size" in L.mjs_last_error(None)
    assert b"abi 3" in L.mjs_version()


def test_no_cpu_fallback_without_gpu(native):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import mujoco_sim_amd as m

    with pytest.raises(m.MjsError):
        m.HipVectorEnv("robot_reach", 4)
    L = native.lib()
    h = C.c_void_p()
    cfg = native.MjsConfig(task=1, num_envs=4, device=0, reward_type=-1, autoreset=0)
    assert L.mjs_create(C.byref(cfg), C.byref(h)) == -2  # MJS_ERR_NO_DEVICE
    assert b"no CPU path" in L.mjs_last_error(None)


def test_product_never_imports_oracle():
    # the oracle is test infrastructure: nothing under mujoco_sim_amd/ may reference it
    for p in (ROOT / "mujoco_sim_amd").rglob("*"):
        if p.is_file() and p.suffix in {".py", ".h", ".hip", ".cpp"}:
            txt = p.read_text()
            assert "import oracle" not in txt and "from oracle" not in txt and "mjs_oracle.h" not in txt, p
