"""Diagnostic A/B builds (development tool): copy csrc/ + include/ to a scratch tree, apply textual substitutions, compile to
ab/libmjsim_<name>.so (git-ignored, travels with gpurun). Usage:
  python tools/ab_build.py NAME [--flag=-fsome-flag ...] [--sub FILE 'old' 'new' ...]
Load with MJS_LIB=ab/libmjsim_NAME.so (bypasses the source-hash check: never for tests or published numbers)."""
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def main():
    name = sys.argv[1]
    flags, subs, i = [], [], 2
    while i < len(sys.argv):
        a = sys.argv[i]
        if a.startswith("--flag="):
            flags.append(a[len("--flag="):]); i += 1
        elif a == "--sub":
            subs.append(tuple(sys.argv[i + 1:i + 4])); i += 4
        else:
            raise SystemExit(f"bad argument {a}")
    tmp = Path(tempfile.mkdtemp(prefix="mjs_ab_"))
    shutil.copytree(ROOT / "mujoco_sim_amd" / "csrc", tmp / "mujoco_sim_amd" / "csrc")
    shutil.copytree(ROOT / "include", tmp / "include")
    for f, old, new in subs:
        p = tmp / f
        s = p.read_text()
        if old not in s:
            raise SystemExit(f"{f}: pattern not found: {old[:60]!r}")
        p.write_text(s.replace(old, new))
    out = ROOT / "ab" / f"libmjsim_{name}.so"
    out.parent.mkdir(exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-comment", "-mllvm",
           "-amdgpu-sched-strategy=max-ilp", *([] if "--ipra" in flags else ["-mllvm", "-enable-ipra=0"]), f'-DMJS_SOURCE_HASH="ab-{name}"', *[f for f in flags if f != "--ipra"], "-o", str(out), str(tmp / "mujoco_sim_amd" / "csrc" / "mjsim.hip")]
    subprocess.run(cmd, check=True)
    shutil.rmtree(tmp)
    print(out)


if __name__ == "__main__":
    main()
