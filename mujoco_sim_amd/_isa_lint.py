#!/usr/bin/env python3
"""ISA lint of the device code: register copies placed AHEAD of the exec-mask restore of a divergent join.

hipcc (ROCm 7.2's LLVM) allocates SGPRs before VGPRs. When the SGPR pass has put its own split copies at the top of the join
block of a divergent `if` (scalar copies may legally sit ahead of the `s_or_b64 exec, exec, sX` that re-enables the lanes
which skipped the `if`), the VGPR pass's search for "the first instruction after the block prologue" stops at those scalar
copies, and its vector split copies (v_accvgpr_write aN, vM; v_mov; spills) land ahead of the exec restore as well. They then
run for the `then` lanes only, and the other lanes read a stale register afterwards. Seen with -enable-ipra=0 around calls of
the out-of-line constraint stages (round 3: the sub-step counter of rr::solo_control_step lived in such an AGPR across the
call: an endless loop on the GPU for the lanes that skipped `if (lazy)`). The signature is mechanical:

  .LBB<n>:                               a block label ...
      s_mov_b32 ..., v_readlane ...      ... scalar instructions only, and
      v_accvgpr_write_b32 a42, v222      <- >= 1 vector register copy / spill / reload
      s_or_b64 exec, exec, s[0:1]        ... up to the exec restore

tools/isa_lint.py file.s [...]: one line per finding, exit code 1 if there is any. mujoco_sim_amd._native.build() runs it on
the device assembly of every build and refuses the library on a finding."""
import re
import sys

COPY_LIKE = re.compile(r"v_accvgpr_(write|read)_b32|v_mov_b(32|64)_e32 v\S+, v|(scratch|buffer)_(store|load)\S* .*Folded (Spill|Reload)")
# Round 4: the scan no longer stops at the first scalar instruction it does not know: ANY scalar instruction that is not a
# branch and does not write exec may sit between the label and the restore (an s_load, an s_waitcnt, address arithmetic ...), and the
# restore itself may be spelled `s_or_b64 exec, exec, sX`, `s_mov_b64 exec, sX` or `s_or_saveexec_b64` (the `else` entry).
SCALAR_OK = re.compile(r"s_(?!cbranch|branch|endpgm|setpc|swappc|call|barrier|trap|sleep)\w+\s|v_(readlane|writelane|readfirstlane)_b32")
# (`s_or_saveexec_b64 sX, -1` is the whole-wave prologue / epilogue of an out-of-line function - callee-saved VGPR spills of ALL lanes -
# not a join: only the register form counts)
EXEC_RESTORE = re.compile(r"s_or_b64 exec, exec, |s_mov_b64 exec, s|s_or_saveexec_b64 s\[\d+:\d+\], s")


def lint(lines):
    findings, fn = [], None
    for k, l in enumerate(lines):
        m = re.match(r"^([A-Za-z_][\w.$]*):", l)
        if m:
            fn = m.group(1)
        if not l.startswith(".LBB"):
            continue
        copies, j = [], k + 1
        while j < len(lines):
            t = lines[j].strip()
            j += 1
            if not t or t.startswith(";"):
                continue
            if EXEC_RESTORE.match(t):
                if copies:
                    findings.append((fn, l.split(":")[0], copies, t))
                break
            if "exec" in t.split(";")[0]:  # any other reader / writer of the mask ends the block prologue
                break
            if COPY_LIKE.match(t):
                copies.append(t.split(";")[0].strip())
            elif not SCALAR_OK.match(t):
                break
    return findings


def main(paths):
    n = 0
    for path in paths:
        with open(path) as f:
            for fn, label, copies, restore in lint(f.read().split("\n")):
                n += 1
                print(f"{path}: {fn} {label}: {len(copies)} vector copy/spill instruction(s) ahead of `{restore}`: {copies[:3]}")
    print(f"isa_lint: {n} finding(s)")
    return 1 if n else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
