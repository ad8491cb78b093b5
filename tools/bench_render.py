#!/usr/bin/env python3
"""Times the camera render kernels (the visual configs): N envs x (res x res x 3) uint8 per launch, HIP events on
the launch stream.  usage: bench_render.py [N] [res] [task] [camera] [kernel_variant: 0 = default, 1 = tile walk]"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import mujoco_sim_amd as m  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
res = int(sys.argv[2]) if len(sys.argv) > 2 else 64
task = sys.argv[3] if len(sys.argv) > 3 else "point_mass_reach"
camera = int(sys.argv[4]) if len(sys.argv) > 4 else 0
variant = int(sys.argv[5]) if len(sys.argv) > 5 else 0
venv = m.HipVectorEnv(task, N, seed=2025, kernel_variant=variant)
venv.reset()
out = torch.empty(N, res, res, 3, dtype=torch.uint8, device="cuda")
for _ in range(20):
    venv.render(res, res, out=out, camera=camera)
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
for a, b in evs:
    a.record()
    venv.render(res, res, out=out, camera=camera)
    b.record()
torch.cuda.synchronize()
ms = float(np.median([a.elapsed_time(b) for a, b in evs]))
nbytes = out.numel()
print(json.dumps({"task": task, "camera": camera, "variant": variant, "envs": N, "res": res, "ms": ms, "bytes_written": nbytes,
                  "GBps": nbytes / ms / 1e6, "frac_of_8TBps": nbytes / ms / 1e6 / 8000, "images_per_s": N / ms * 1e3}))
