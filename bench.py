#!/usr/bin/env python3
"""bench.py — env-steps/sec of the Robot-Reach hot path (BASELINE.json configs[2]: UR5e 6-DoF,
4096 envs per MI355X, joint-space obs). One "step" = one control step of all envs of this rank
(IK + 20 physics substeps + obs/reward/termination + auto-reset), one kernel launch.

  python bench.py --gpus N --steps K --warmup W
      N > 1: either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or run
      plainly, in which case this process spawns the N rank processes itself and never touches the GPU.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     : algorithmic HBM bytes per launch / HIP-event kernel time vs 8 TB/s
  cpu_baseline : the CPU oracle (oracle/, a scalar C restatement; kind "port") timed on this
                 box's host cores on a bounded sample of the same workload (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def measured_traffic(task, n_local):
    """HBM bytes per launch from the committed PMC passes (profiles/r03_traffic_all_tasks.json: rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs of this very command, FETCH corrected per the gfx950 calibration of
    profiles/r01_traffic.json). Only valid for the configuration it was measured on; otherwise None."""
    for name in ("r04_traffic_all_tasks.json", "r03_traffic_all_tasks.json"):  # measured on the kernels of THIS round (tools/profile_round.sh + collate_profiles.py)
        try:
            d = json.load(open(ROOT / "profiles" / name))["tasks"][task]
            if n_local == d["envs"]:
                return d["corrected_bytes_per_launch"]["total"]
        except Exception:  # noqa: BLE001
            pass
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--task", default="robot_reach", choices=["robot_reach", "point_mass_reach", "robot_push_button", "robot_planar_push"])
    ap.add_argument("--envs-per-gpu", type=int, default=4096, help="weak scaling: envs per GPU; strong scaling: envs of the WHOLE job (split over the ranks)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --envs-per-gpu envs on every GPU; strong: --envs-per-gpu envs in total, 1/N of them per rank "
                         "(north_star's '4096 parallel Robot-Reach envs at 1/2/4/8 MI355X' read literally)")
    ap.add_argument("--n-objects", type=int, default=2, help="Planar-Push blocks: 2 = BASELINE config 4, 5 = the reference's dataclass default")
    ap.add_argument("--block-shape", default="mesh", choices=["mesh", "box"], help="Planar-Push blocks: the reference's meshes (default) or round 1's box stand-in")
    ap.add_argument("--gripper-model", default="reduced", choices=["reduced", "articulated"],
                    help="Button-Push: the one-coordinate 2F-85 of DESIGN.md D-1b (BASELINE config 5's kernel) or the articulated 2F-85 (SURVEY 8 f-1, nv = 14)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant for A/B profiling (0 = default)")
    ap.add_argument("--visual", type=int, default=0, metavar="RES",
                    help="also render the task's camera(s) at RESxRES every step (BASELINE config 5: Button-Push, 64)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="time box of the CPU oracle sample")
    ap.add_argument("--stub", action="store_true",
                    help="launcher self-test: gloo on CPU with a no-physics stand-in env (tests/test_host_logic.py); never a measurement")
    return ap.parse_args()


def make_actions(task, T, N, device, seed):
    # SURVEY.md §8d synthetic inputs: uniform in the task's action / workspace box
    rs = np.random.RandomState(seed)
    if task == "point_mass_reach":
        a = rs.uniform(-0.05, 0.05, (T, N, 2)).astype(np.float32).astype(np.float64)
    elif task == "robot_planar_push":
        # SURVEY.md section 8d, cfg 4: absolute TCP xy uniform in the robot workspace x in [-0.2, 0.2], y in [-0.6, -0.3]
        a = rs.uniform([-0.2, -0.6], [0.2, -0.3], (T, N, 2))
    elif task == "robot_push_button":
        # SURVEY.md section 8d, cfg 5: joint targets q_home +- U(0.2) (robot.py:307) + gripper U(0, 0.085); 7-D absolute
        # joint actions are the registered action type
        home = np.array([-0.5, -0.5, 0.5, -0.5, -0.5, -0.5]) * np.pi
        a = np.concatenate([home + rs.uniform(-0.2, 0.2, (T, N, 6)), rs.uniform(0.0, 0.085, (T, N, 1))], axis=2)
    else:
        a = rs.uniform([-0.1, -0.6, 0.02], [0.1, -0.4, 0.2], (T, N, 3))
    return torch.from_numpy(a).to(device)


def host_cores():
    """Threads for the CPU leg: affinity, capped by the cgroup CPU quota and by the GPU box's
    per-GPU CPU share (16; override with MJS_BENCH_CORES)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:  # noqa: BLE001
        pass
    return max(1, min(n, int(os.environ.get("MJS_BENCH_CORES", "16"))))


def cpu_baseline(task, n_envs, seconds, n_objects=2, block_shape="mesh", gripper_model="reduced"):
    import oracle

    tid = {"robot_reach": oracle.TASK_ROBOT_REACH, "point_mass_reach": oracle.TASK_POINTMASS, "robot_push_button": oracle.TASK_BUTTON_PUSH,
           "robot_planar_push": oracle.TASK_PLANAR_PUSH}[task]
    cores = host_cores()
    kw = {"n_objects": n_objects, "block_shape": oracle.BLOCKS_BOX if block_shape == "box" else oracle.BLOCKS_MESH} if task == "robot_planar_push" else {}
    if task == "robot_push_button" and gripper_model == "articulated":
        kw["gripper_model"] = 1
    b = oracle.OracleBatch(tid, n_envs, 2025, nthreads=cores, **kw)
    b.reset()
    acts = make_actions(task, 8, n_envs, "cpu", 12345).numpy()
    b.step(acts[0])  # warm
    steps, t0 = 0, time.perf_counter()
    while True:
        b.step(acts[steps % 8])
        steps += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or steps >= 20000:
            break
    return {"value": steps * n_envs / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n_envs} envs x {steps} control steps ({dt:.1f} s), oracle/ C restatement, OpenMP over envs"}


def mujoco_probe():
    """Is a real MuJoCo importable on THIS box (SURVEY section 8c last row, BASELINE.md baseline A)? Asked with find_spec only:
    nothing of the reference travels, and nothing is imported unless it is there. Round 3's answer on the GPU pool: absent
    (profiles/r03_a_mujoco_probe.txt), so the CPU column is the oracle (kind "port") and the physics stay parity-unpinned."""
    import importlib.util

    found = {}
    for mod in ("mujoco", "dm_control"):
        try:
            found[mod] = importlib.util.find_spec(mod) is not None
        except Exception:  # noqa: BLE001
            found[mod] = False
    out = {"mujoco_on_box": found["mujoco"], "dm_control_on_box": found["dm_control"]}
    if found["mujoco"]:
        try:
            import mujoco  # noqa: PLC0415

            out["mujoco_version"] = getattr(mujoco, "__version__", "?")
            out["mujoco_pointmass"] = mujoco_pointmass_baseline(mujoco)
        except Exception as e:  # noqa: BLE001
            out["mujoco_error"] = repr(e)[:200]
    return out


def mujoco_pointmass_baseline(mujoco, seconds=5.0):
    """Baseline A on the build's OWN MJCF of the Pointmass scene (tests/golden/pointmass_scene.xml, authored from
    walled_pointmass_arena.xml:11-20, pointmass.py:52-66, point_reach.py:79-93): raw mj_step rate of one MjData on one host
    thread (dmc2gym.py:136's Physics.step without the Python around it). Only runs where `import mujoco` works."""
    model = mujoco.MjModel.from_xml_path(str(ROOT / "tests" / "golden" / "pointmass_scene.xml"))
    data = mujoco.MjData(model)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(1000):
            mujoco.mj_step(model, data)
        n += 1000
    dt = time.perf_counter() - t0
    return {"mj_steps_per_s": n / dt, "env_steps_per_s": n / dt / 5, "cores": 1, "sample": f"{n} mj_step calls, 1 MjData, 1 thread"}


def single_env_figures(device, n_steps=2000):
    """BASELINE config 1 in the reference's own style (environments/tasks/utils.py:6-17: n_steps iterations of
    `reset if done else step(action_space.sample())`, resets included, seconds per iteration): Pointmass-Reach, ONE env,
    through the single-env adapter. (a) the HIP path behind the reference's DMCEnvironmentAdapter surface (one kernel launch +
    one device->host read-back per step: host-bound); (b) the oracle, one env, one host thread, called from Python."""
    import mujoco_sim_amd as m
    import oracle

    env = m.make("mujoco_sim/point_mass_reach_state-v0", device=str(device))
    env.seed(2025)
    done, t0 = True, time.perf_counter()
    for _ in range(n_steps):
        if done:
            env.reset()
            done = False
        else:
            _, _, term, trunc, _ = env.step(env.action_space.sample())
            done = term or trunc
    hip = (time.perf_counter() - t0) / n_steps
    env.close()
    ob = oracle.OracleBatch(oracle.TASK_POINTMASS, 1, 2025, nthreads=1, autoreset=2)
    rs = np.random.RandomState(2025)
    done, t0 = True, time.perf_counter()
    for _ in range(n_steps):
        if done:
            ob.reset()
            done = False
        else:
            o = ob.step(rs.uniform(-0.05, 0.05, (1, 2)).astype(np.float32).astype(np.float64))
            done = bool(o["step_type"][0] == 2)
    cpu = (time.perf_counter() - t0) / n_steps
    return {"workload": f"Pointmass-Reach, 1 env, {n_steps} iterations of reset-if-done-else-random-step (tasks/utils.py:6-17)",
            "hip_adapter_s_per_step": hip, "hip_adapter_env_steps_per_s": 1.0 / hip,
            "oracle_s_per_step": cpu, "oracle_env_steps_per_s": 1.0 / cpu,
            "note": "HIP: one launch + read-back per step behind DMCEnvironmentAdapter (host-bound); oracle: the C restatement, 1 thread, via ctypes"}


class _StubVectorEnv:
    """Launcher self-test stand-in (``--stub``, CPU, gloo): the same surface bench.py drives, NO physics and no GPU.
    Its only purpose is to let a GPU-less test exercise the rank fan-out, the barrier / max-over-ranks timing and the
    rollout gather; the JSON line it produces is marked ``"stub": true`` and is never a measurement."""

    def __init__(self, n_local, rank):
        self.num_envs, self.obs_dim, self.algorithmic_bytes_per_env_step, self.rank = n_local, 12, 379, rank
        self._buf = {"obs": torch.zeros(n_local, 12, dtype=torch.float64), "fault": torch.zeros(n_local, dtype=torch.uint8)}

    def reset(self):
        self._buf["obs"].zero_()

    def step_flat(self, a):
        self._buf["obs"][:, 0] += 1.0

    def rollout(self, a, keep=("obs",)):
        T = a.shape[0]
        o = torch.zeros(T, self.num_envs, 12, dtype=torch.float64)
        o[..., 0] = self.rank
        o[..., 1] = torch.arange(self.num_envs, dtype=torch.float64)
        return {"obs": o}

    def close(self):
        pass


def launch_ranks(args) -> int:
    """``python bench.py --gpus N`` outside torchrun: fan out to N fresh rank processes (the counterpart of the
    reference's ``SubprocVecEnv([create_env(rank=i, seed=...) ...])``, scripts/sb3/reach_sac.py:93-96, one env shard per
    process with per-rank seeds ``seed + rank``-style global seeding). THIS process never touches the GPU (no HIP call,
    no ``torch.cuda`` query): it only spawns ``python bench.py ...`` children with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set, relays rank 0's JSON line and returns non-zero if any rank fails."""
    import socket
    import subprocess

    if not args.stub:  # compile once here (hipcc only, no GPU): N children would otherwise race N links to the same output path
        from mujoco_sim_amd import _native

        _native.build()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    import threading

    out0, rc = "", 0
    deadline = time.time() + float(os.environ.get("MJS_BENCH_LAUNCH_TIMEOUT", "1500"))
    chunks: list[str] = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)  # drain rank 0's pipe while polling
    reader.start()
    # poll ALL children: the first rank that dies ends the job at once (its siblings would otherwise sit in the rendezvous
    # until the timeout), each remaining child is stopped by its own handle
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        if any(c not in (None, 0) for c in codes):
            break
        if time.time() > deadline:
            rc = 124
            break
        time.sleep(0.05)
    for p in procs:
        if p.poll() is None and (rc or any(c not in (None, 0) for c in codes)):
            p.kill()
    for p in procs:
        p.wait()
    reader.join(timeout=10)
    out0 = "".join(chunks)
    for r, p in enumerate(procs):
        if p.poll() is None:  # a rank outlived a failed / timed-out sibling: stop exactly that child
            p.kill()
            p.wait()
        if p.returncode != 0:
            print(f"bench.py: rank {r} exited with code {p.returncode}", file=sys.stderr)
            rc = rc or (p.returncode if p.returncode > 0 else 1)
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if rc == 0 and len(lines) != 1:
        print(f"bench.py: rank 0 printed {len(lines)} JSON lines", file=sys.stderr)
        rc = 1
    if lines:
        print(lines[-1], flush=True)
    return rc


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))
    from mujoco_sim_amd import distributed as D

    rank, local_rank, world = D.init_process_group(backend="gloo" if args.stub else None)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.scaling == "strong":  # the job's env count is fixed, every rank takes 1/N of it (global seeds as always)
        if args.envs_per_gpu % world:
            raise SystemExit(f"bench.py: --scaling strong needs --envs-per-gpu ({args.envs_per_gpu}) divisible by --gpus ({world})")
        n_local = args.envs_per_gpu // world
    else:
        n_local = args.envs_per_gpu
    n_global = n_local * world
    if args.stub:
        device = torch.device("cpu")
        venv = _StubVectorEnv(n_local, rank)
        sync = lambda: None  # noqa: E731
        substeps = 20
    else:
        device = torch.device(f"cuda:{local_rank}")
        torch.cuda.set_device(device)
        import mujoco_sim_amd as m

        extra = {"n_objects": args.n_objects, "block_shape": args.block_shape} if args.task == "robot_planar_push" else {}
        if args.task == "robot_push_button" and args.gripper_model == "articulated":
            extra["gripper_model"] = "articulated"
        # global seeds: env i of the whole job <- RandomState(2025 + i) whatever the rank count (reach_sac.py:84 seeds sub-env rank with seed + rank)
        venv = m.HipVectorEnv(args.task, n_local, device=device, seed=2025, env_index_offset=rank * n_local, kernel_variant=args.variant, **extra)
        sync = lambda: torch.cuda.synchronize(device)  # noqa: E731
        substeps = venv._lib.mjs_substeps(venv.spec.task_id)
    venv.reset()
    chunk = 64  # distinct action slabs resident in HBM, cycled
    acts = make_actions(args.task, chunk, n_local, device, 12345 + rank)
    cams = ([0, 1] if args.task == "robot_push_button" else [0]) if args.visual else []
    imgs = [torch.empty(n_local, args.visual, args.visual, 3, dtype=torch.uint8, device=device) for _ in cams]

    def one_step(i):
        venv.step_flat(acts[i % chunk])
        for cam, img in zip(cams, imgs):  # visual observations: what Camera.get_rgb_image does per control step
            venv.render(args.visual, args.visual, out=img, camera=cam)

    for i in range(args.warmup):
        one_step(i)
    D.barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(i)
    sync()
    D.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = D.max_over_ranks(elapsed, device=device)

    # per-launch kernel time with HIP events on the launch stream (torch's current stream). Each event pair brackets a
    # short train of back-to-back launches so that the ~4 us cost of the event records themselves is amortised and the
    # figure is the kernel's own duration (it then agrees with rocprofv3's per-kernel average, profiles/).
    if args.stub:
        kernel_ms = elapsed / args.steps * 1e3
    else:
        train = 10 if not cams else 1
        n_ev = min(40, max(4, args.steps // train))
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
        for g, (a, b) in enumerate(evs):
            a.record()
            for k in range(train):
                venv.step_flat(acts[(g * train + k) % chunk])
            b.record()
        torch.cuda.synchronize(device)
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in evs])) / train
    faults = int(venv._buf["fault"].max().item())

    # N > 1: the one collective of the design — all-gather of a rollout block once per 64-step chunk (RCCL over xGMI;
    # gloo in the stub) — outside the timed step region and reported on its own; every rank checks that its own shard
    # came back bit-identical at its global position.
    gather = None
    if world > 1:
        block = venv.rollout(acts, keep=("obs",))["obs"]  # [64, n_local, obs_dim]
        full = D.gather_rollout(block)
        ok = tuple(full.shape) == (chunk, n_global, block.shape[2]) and bool(torch.equal(full[:, rank * n_local:(rank + 1) * n_local], block))
        D.barrier()
        sync()
        tg = time.perf_counter()
        reps = 5
        for _ in range(reps):
            full = D.gather_rollout(block)
        sync()
        D.barrier()
        tg = D.max_over_ranks((time.perf_counter() - tg) / reps, device=device)
        ok = D.max_over_ranks(0.0 if ok else 1.0, device=device) == 0.0
        nbytes = block.numel() * block.element_size()
        gather = {"chunk_steps": chunk, "bytes_per_rank": nbytes, "ms_per_chunk": tg * 1e3, "ms_per_step_amortised": tg * 1e3 / chunk,
                  "recv_GBps_per_rank": nbytes * (world - 1) / tg / 1e9, "backend": torch.distributed.get_backend(), "shards_bit_identical": ok}
        if not ok:
            raise SystemExit("bench.py: gathered rollout block does not contain this rank's shard at its global position")

    if rank == 0:
        env_steps = args.steps * n_global
        value = env_steps / elapsed
        bytes_per_env_step = venv.algorithmic_bytes_per_env_step
        bytes_per_launch = bytes_per_env_step * n_local
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": {"robot_reach": "env-steps/sec at N_envs=4096, Robot-Reach", "point_mass_reach": "env-steps/sec, Pointmass-Reach",
                       "robot_push_button": "env-steps/sec, Button-Push (" + (f"scene + wrist camera {args.visual}x{args.visual}" if cams else "state obs") + ")",
                       "robot_planar_push": f"env-steps/sec, Planar-Push, {args.n_objects} objects (state obs)"}[args.task],
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.task}: {n_local} envs per GPU, {substeps} substeps/step, "
                                   f"{'state obs' if not cams else 'visual obs'}, {'joint targets q_home +- U(0.2) + gripper opening U(0, 0.085)' if args.task == 'robot_push_button' else 'uniform workspace actions'}, next-step auto-reset"
                                   + (f", + {len(cams)} camera image(s) {args.visual}x{args.visual} per step" if cams else "")
                                   + (", articulated Robotiq 2F-85 (nv = 14)" if (args.task == "robot_push_button" and args.gripper_model == "articulated") else ""),
                       "envs_per_gpu": n_local, "envs_total": n_global, "parallelism": f"env-sharded x{world}, no collective in the step",
                       "scaling_mode": ("weak: envs_per_gpu fixed, envs_total grows with the GPU count" if args.scaling == "weak" else
                                        "strong: envs_total fixed, envs_per_gpu = envs_total / n_gpus (a launch is latency-bound below ~16k envs per GPU: expect ~1x)")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None if (args.stub or (args.task == "robot_planar_push" and args.n_objects != 2)) else measured_traffic(args.task + ("_articulated" if (args.task == "robot_push_button" and args.gripper_model == "articulated") else ""), n_local), "kernel_ms": kernel_ms, "algorithmic_bytes_per_env_step": bytes_per_env_step,
                         "note": "state fits in L2 at this size; the kernel is bound by per-lane FP64 dependency chains (DESIGN.md)"},
            "faults": faults,
        }
        if args.stub:
            line["stub"] = True
            line["data"] = "STUB: launcher self-test on CPU, no physics, not a measurement"
        if gather is not None:
            line["rollout_gather"] = gather
        if args.task == "robot_reach" and not args.stub:
            # informational: the bound that actually applies. FP64 operation count per env-step:
            # 20 substeps x (generated M + bias code 725 ops + servo/actuators/U D U^T/inverse/integration ~330) + IK ~3000
            # + FK/observables ~500; peak = 1024 SIMDs x 16 FP64 FMA lanes x 2 x ~2.4 GHz (public spec 78.6 TFLOP/s;
            # measured here: 4 cycles per wave64 FP64 instruction per SIMD). 4096 envs occupy 128 of the 1024 SIMDs.
            flops = 20 * (725 + 330) + 3000 + 500
            tf = flops * value / 1e12
            line["roofline"]["valu_fp64"] = {"flops_per_env_step": flops, "achieved_tflops": tf, "peak_tflops": 78.6, "frac": tf / 78.6,
                                             "simds_occupied_frac": min(1.0, (2 if args.variant == 2 else 1 if args.variant == 1 else 3) * (n_local / 64) / 1024)}
            try:  # PMC view of this very launch shape (profiles/r03_reach_valu.json): VALU busy share of a wavefront's lifetime
                v = json.load(open(ROOT / "profiles" / ("r04_reach_valu.json" if (ROOT / "profiles" / "r04_reach_valu.json").exists() else "r03_reach_valu.json")))
                if v["envs"] == n_local:
                    line["roofline"]["valu_fp64"]["valu_busy_frac_on_occupied_simds_pmc"] = v["valu_busy_frac_of_wave_lifetime"]
            except Exception:  # noqa: BLE001
                pass
        if world == 1 and not args.no_cpu_baseline and not args.stub:
            line["cpu_baseline"] = cpu_baseline(args.task, n_local, args.cpu_seconds, args.n_objects, args.block_shape, args.gripper_model)
            line["cpu_baseline"].update(mujoco_probe())  # "mujoco_on_box": asked on every run, acted on only when true
            if args.task == "robot_reach":
                line["config1_single_env"] = single_env_figures(device)
        print(json.dumps(line))
    venv.close()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
