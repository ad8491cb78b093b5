"""The ISA lint every build of libmjsim.so runs on its device assembly (mujoco_sim_amd/_isa_lint.py): vector register copies
ahead of the exec-mask restore of a divergent join (a code-generation fault of the toolchain, found in round 3 as an endless
loop of rr::solo_control_step on the GPU). The signature on hand-made assembly; the real build is linted by build()."""
from mujoco_sim_amd._isa_lint import lint

FAULTY = """
_ZN2rr17solo_control_stepENS_6SoloInENS_2WsE:
\ts_and_saveexec_b64 s[0:1], vcc
\ts_cbranch_execz .LBB44_21
; %bb.20:
\tv_mul_f64 v[4:5], v[4:5], v[0:1]
\ts_or_b64 s[4:5], s[2:3], exec
.LBB44_21:                              ;   in Loop: Header=BB44_4 Depth=1
\ts_mov_b32 s53, s97
\tv_accvgpr_write_b32 a42, v222
\ts_mov_b64 s[96:97], s[98:99]
\ts_or_b64 exec, exec, s[0:1]
\tv_mov_b32_e32 v200, v138
.Lfunc_end44:
"""

SOUND = """
_ZN2pm6kernelILb0EEEv12KernelParams:
.LBB49_41:
\tv_mov_b64_e32 v[18:19], v[94:95]
\ts_and_saveexec_b64 s[10:11], s[24:25]
; %bb.42:
\tv_mov_b64_e32 v[18:19], v[142:143]
; %bb.43:
\ts_or_b64 exec, exec, s[10:11]
.LBB49_44:
\ts_mov_b32 s53, s97
\ts_or_b64 exec, exec, s[8:9]
\tv_accvgpr_write_b32 a42, v222
.LBB49_45:
\tv_fma_f64 v[12:13], -v[76:77], v[200:201], v[12:13]
\tv_accvgpr_write_b32 a42, v222
\ts_or_b64 exec, exec, s[8:9]
.Lfunc_end49:
"""


def test_copy_ahead_of_the_exec_restore_is_found():
    found = lint(FAULTY.split("\n"))
    assert len(found) == 1
    fn, label, copies, restore = found[0]
    assert fn.startswith("_ZN2rr17solo_control_step") and label == ".LBB44_21"
    assert copies == ["v_accvgpr_write_b32 a42, v222"] and restore == "s_or_b64 exec, exec, s[0:1]"


def test_sound_joins_pass():
    # a copy ahead of a saveexec (the `then` block of a nested if), a copy after the restore, real work ahead of a restore
    assert lint(SOUND.split("\n")) == []


WIDER = """
_ZN2bg9st_forcesEPdm:
.LBB7_12:
\ts_load_dwordx2 s[4:5], s[0:1], 0x10
\ts_waitcnt lgkmcnt(0)
\ts_mul_i32 s6, s6, s7
\tscratch_load_dwordx2 v[76:77], off, s6 ; 8-byte Folded Reload
\ts_mov_b64 exec, s[8:9]
.LBB7_13:
\ts_bfe_u32 s2, s3, 0x10008
\tv_accvgpr_read_b32 v5, a9
\ts_or_saveexec_b64 s[10:11], s[12:13]
.Lfunc_end7:
"""


def test_round4_widening_scalar_loads_and_other_restore_forms():
    """a split copy placed after a scalar instruction the old table did not list (an s_load, an s_mul), and restores
    written as `s_mov_b64 exec` / `s_or_saveexec_b64`, are findings too."""
    found = lint(WIDER.split("\n"))
    assert [f[1] for f in found] == [".LBB7_12", ".LBB7_13"]
    assert found[0][3].startswith("s_mov_b64 exec") and found[1][3].startswith("s_or_saveexec_b64")
