"""Child process of tests/test_gpu_parity.py::test_rccl_backend_initialises_and_gathers_a_rollout_block: one rank, backend "nccl"
(= RCCL on ROCm), a real dist.all_gather_into_tensor on a device rollout block (gather_rollout returns early at world size 1,
so the collective is called directly here). Prints one JSON line."""
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.distributed as dist

import mujoco_sim_amd as m


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1])
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    N, T = 256, 8
    venv = m.HipVectorEnv("robot_reach", N, seed=2025)
    venv.reset()
    acts = torch.rand(T, N, 3, dtype=torch.float64, device="cuda") * torch.tensor([0.2, 0.2, 0.18], device="cuda", dtype=torch.float64) + torch.tensor([-0.1, -0.6, 0.02], device="cuda", dtype=torch.float64)
    block = venv.rollout(acts)["obs"]                       # [T, N, 12] on the device
    moved = block.movedim(1, 0).contiguous()                # what gather_rollout hands to the collective
    out = torch.empty_like(moved)
    dist.all_gather_into_tensor(out, moved)                 # RCCL, world size 1
    t = torch.tensor([3.5], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                # bench.py's max-over-ranks
    dist.barrier()
    torch.cuda.synchronize()
    ok = bool(torch.equal(out, moved)) and float(t.item()) == 3.5
    print(json.dumps({"backend": dist.get_backend(), "world": dist.get_world_size(), "gather_identical": ok, "bytes": int(moved.numel() * 8)}))
    dist.destroy_process_group()
    venv.close()


if __name__ == "__main__":
    main()
