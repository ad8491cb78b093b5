"""Stage profile of the articulated-gripper kernel: builds a DIAGNOSTIC library (-DMJS_BG_PROFILE: per-stage shader clocks written
over the observations) next to the shipped one and prints the mean clocks per control step of each stage."""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
out = Path(sys.argv[1] if len(sys.argv) > 1 else "/tmp/libmjsim_prof.so")
if not out.exists():
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-comment", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-enable-ipra=0",
                    "-DMJS_BG_PROFILE", "-fPIC", "-shared", "-o", str(out), str(ROOT / "mujoco_sim_amd/csrc/mjsim.hip")], check=True)
if len(sys.argv) > 2 and sys.argv[2] == "build":
    sys.exit(0)
os.environ["MJS_LIB"] = str(out)
sys.path.insert(0, str(ROOT))
import torch
import mujoco_sim_amd as m
n = 4096
v = m.HipVectorEnv("robot_push_button", n, seed=5, action_type="absolute_joint_action", gripper_model="articulated")
v.reset()
a = torch.zeros(n, 7, dtype=torch.float64, device="cuda")
st = v.get_state()
a[:, :6] = st[0:6].T
a[:, 6] = 0.04
for k in range(3):
    v.step(a)
prof = v._buf["obs"][:, :7].cpu().numpy()

names = ["kinematics", "crb+factor", "collision", "rows", "velocity", "forces+solve", "integrate"]
tot = prof.mean(axis=0).sum()
for k, nm in enumerate(names):
    print(f"{nm:14s} {prof[:, k].mean():12.0f} clk  {100 * prof[:, k].mean() / tot:5.1f} %   max {prof[:, k].max():12.0f}")
dbg = v._buf["obs"][:, 7:13].cpu().numpy()
for k, nm in enumerate(["(spare)", "solver: newton_direction", "solver: line search", "solver: update pass", "forces: factor-solve of M", "forces: the whole solver"]):
    print(f"  {nm:48s} {dbg[:, k].mean():12.0f} clk  {100 * dbg[:, k].mean() / tot:5.1f} %")
print("total", tot, "clocks per control step (clock64 = 100 MHz wall clock on gfx9: x21 for shader cycles)")
