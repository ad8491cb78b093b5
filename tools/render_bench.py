"""Render-only timing (development tool): Button-Push at BASELINE config 5's shape (2048 envs, 64x64), the scene camera and
the wrist camera timed separately with events on the launch stream after a few random steps (arms in varied poses).
Run on the GPU box: python tools/render_bench.py [--envs 2048] [--res 64] [--reps 50]. MJS_LIB selects a diagnostic build."""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import mujoco_sim_amd as m  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=2048)
    ap.add_argument("--res", type=int, default=64)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--task", default="robot_push_button")
    a = ap.parse_args()
    venv = m.HipVectorEnv(a.task, a.envs, seed=11)
    venv.reset()
    g = torch.Generator(device="cuda").manual_seed(3)
    for _ in range(8):
        venv.step(torch.rand(a.envs, venv.action_dim, device="cuda", dtype=torch.float64, generator=g) * 2 - 1)
    out = {}
    cams = [0, 1] if a.task == "robot_push_button" else [0]
    for cam in cams:
        img = torch.empty(a.envs, a.res, a.res, 3, dtype=torch.uint8, device="cuda")
        for _ in range(5):
            venv.render(a.res, a.res, out=img, camera=cam)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            venv.render(a.res, a.res, out=img, camera=cam)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        out[f"camera{cam}"] = {"ms": round(ms, 4), "GB/s": round(img.numel() / ms / 1e6, 1), "checksum": int(img.to(torch.int64).sum().item())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
