"""The generated straight-line UR5e dynamics (tools/gen_ur5e_dynamics.py -> csrc/mjs_ur5e_dyn_gen.h: link-local
CRBA + RNE with the structural zeros folded, what the Robot-Reach and Button-Push kernels execute) against the
oracle's generic engine (composite-rigid-body + recursive Newton-Euler over the body tree, om_engine.c) on random
states: joint-space inertia, bias forces and the mj_setConst constants. Compiled for the host with g++."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def gen_lib(tmp_path_factory):
    out = tmp_path_factory.mktemp("gen") / "libgen_dyn_host.so"
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", str(ROOT / "tests/support/gen_dyn_host.cpp"), "-o", str(out)], check=True)
    return C.CDLL(str(out))


@pytest.mark.parametrize("variant", [0, 1])
def test_generated_dynamics_match_generic_engine(oracle_mod, gen_lib, variant):
    L = oracle_mod.lib()
    rs = np.random.RandomState(7 + variant)
    worst_M = worst_b = 0.0
    for _ in range(200):
        q = rs.uniform(-3.1, 3.1, 6)
        v = rs.uniform(-3.0, 3.0, 6)
        Mg, bg = np.zeros(36), np.zeros(6)
        gen_lib.gen_dynamics(variant, q.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), Mg.ctypes.data_as(C.c_void_p), bg.ctypes.data_as(C.c_void_p))
        Mo, bo, consts = np.zeros(36), np.zeros(6), np.zeros(9)
        if variant == 0:
            L.om_debug_reach_dynamics(q.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), Mo.ctypes.data_as(C.c_void_p), bo.ctypes.data_as(C.c_void_p))
        else:
            L.om_debug_button_dynamics(q.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), Mo.ctypes.data_as(C.c_void_p), bo.ctypes.data_as(C.c_void_p),
                                       consts.ctypes.data_as(C.c_void_p))
        Mo = Mo.reshape(6, 6) - 0.1 * np.eye(6)  # the engine's M includes the joint armature (MJS_UR_ARMATURE)
        worst_M = max(worst_M, np.abs(Mg.reshape(6, 6) - Mo).max())
        worst_b = max(worst_b, np.abs(bg - bo).max())
    assert worst_M < 1e-13 and worst_b < 5e-12, (worst_M, worst_b)
    if variant == 1:  # constants the contact / limit rows use
        g = np.zeros(9)
        gen_lib.gen_constants(1, g.ctypes.data_as(C.c_void_p))
        np.testing.assert_allclose(g[7:9], consts[0:2], rtol=1e-12)   # EEF body invweight0
        np.testing.assert_allclose(g[6], consts[2], rtol=1e-12)       # meaninertia
        np.testing.assert_allclose(g[0:6], consts[3:9], rtol=1e-12)   # dof_invweight0


@pytest.mark.parametrize("variant,task", [(0, 1), (1, 3), (2, 2)])
def test_generated_link_body_invweights_match_generic_engine(oracle_mod, gen_lib, variant, task):
    """The arm-floor contact rows (mjs_arm_stage.h) take their diagApprox from <V>_LINK_BODY_INVWEIGHT0: the generator's
    values against the oracle's mj_setConst on the full model of each scene (base and shoulder link: 0 / mjMINVAL, their
    COMs sit on the first joint's axis at qpos0)."""
    L = oracle_mod.lib()
    g, o = np.zeros(7), np.zeros(7)
    gen_lib.gen_link_invweights(variant, g.ctypes.data_as(C.c_void_p))
    L.om_debug_link_invweights(task, o.ctypes.data_as(C.c_void_p))
    np.testing.assert_allclose(np.maximum(g, 1e-15)[2:], o[2:], rtol=1e-10)
    assert (o[:2] <= 1e-12).all() and (g[:2] <= 1e-12).all(), (o, g)


def test_generated_factor_inverse_block_solves_the_implicit_system(gen_lib):
    """ur5e_MW_gen (the Robot-Reach role-0 block as one scheduled expression graph: M + dd -> U D U^T -> V = U^-1) with
    rr::apply_inverse's two mat-vecs equals numpy's solve of (M(q) + diag(dd)) x = b; M from ur5e_M_gen."""
    rs = np.random.RandomState(11)
    worst = 0.0
    for _ in range(200):
        q = rs.uniform(-3.1, 3.1, 6)
        dd = 0.1 + rs.uniform(0.0, 1.0, 6) * (rs.uniform(size=6) < 0.7)   # armature + dt * kd of the unclamped actuators
        b = rs.uniform(-50, 50, 6)
        x = np.zeros(6)
        gen_lib.gen_solve(q.ctypes.data_as(C.c_void_p), dd.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p))
        Mg, bg = np.zeros(36), np.zeros(6)
        gen_lib.gen_dynamics(0, q.ctypes.data_as(C.c_void_p), np.zeros(6).ctypes.data_as(C.c_void_p), Mg.ctypes.data_as(C.c_void_p), bg.ctypes.data_as(C.c_void_p))
        ref = np.linalg.solve(Mg.reshape(6, 6) + np.diag(dd), b)
        worst = max(worst, np.abs(x - ref).max() / max(1.0, np.abs(ref).max()))
    assert worst < 1e-11, worst
