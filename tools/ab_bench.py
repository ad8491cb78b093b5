#!/usr/bin/env python3
"""Same-session A/B timing of several libmjsim builds (different -D flags / flags): each library is
loaded in its OWN subprocess (ctypes cannot reload a library), rounds are interleaved
(A B C A B C ...) so that clock/thermal drift hits all arms equally, and the per-arm median of the
HIP-event kernel time is reported. Usage: python tools/ab_bench.py name=path.so[:variant] ..."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
rounds = int(os.environ.get("AB_ROUNDS", "5"))
arms = []
for a in sys.argv[1:]:
    name, rest = a.split("=")
    path, _, variant = rest.partition(":")
    arms.append((name, path, variant or "0"))
res = {n: [] for n, _, _ in arms}
for r in range(rounds):
    for name, path, variant in arms:
        env = dict(os.environ, MJS_LIB=str(Path(path).resolve()))
        extra = os.environ.get("AB_ARGS", "--steps 1500 --warmup 100").split()  # e.g. AB_ARGS="--task robot_push_button --steps 500 --warmup 50"
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), *extra, "--no-cpu-baseline", "--variant", variant],
                             env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        res[name].append((d["roofline"]["kernel_ms"] * 1e3, d["value"] / 1e6))
for name, v in res.items():
    ks = sorted(x[0] for x in v)
    vs = sorted(x[1] for x in v)
    print(f"{name:28s} kernel_us median {ks[len(ks)//2]:7.2f}  min {ks[0]:7.2f}  max {ks[-1]:7.2f} | Menv-steps/s median {vs[len(vs)//2]:7.2f}")
