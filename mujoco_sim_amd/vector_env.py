"""HipVectorEnv — N environments of one task stepped by one HIP kernel per control step.

Host-side mirror of the reference's step/reset boundary
(mujoco_sim/environments/dmc2gym.py:133-163) with a gymnasium ``VectorEnv``-shaped surface:
``reset(seed=...) -> (obs, info)``, ``step(actions) -> (obs, reward, terminated, truncated,
info)``; everything stays on the GPU as torch tensors (float64 like MuJoCo's mjtNum).
PyTorch is plumbing only (device memory + streams); the arithmetic is libmjsim.so.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from dataclasses import dataclass

import numpy as np
import torch

from . import _native as nat
from .spaces import Box, Dict, batch_box

# reward / observation type strings of the reference (point_reach.py:11-17, robot_reach.py:37-41)
SPARSE_REWARD = "sparse_reward"
DENSE_POTENTIAL_REWARD = "dense_potential_reward"
DENSE_NEG_DISTANCE_REWARD = "dense_negative_distance_reward"
DENSE_BIASED_NEG_DISTANCE_REWARD = "dense_biased_negative_distance_reward"
STATE_OBS = "state_observations"
VISUAL_OBS = "visual_observations"
# Button-Push action types (robot_push_button.py:28-29)
ABS_EEF_ACTION = "absolute_eef_action"
ABS_JOINT_ACTION = "absolute_joint_action"
_ACTION_IDS = {ABS_JOINT_ACTION: nat.ACTION_ABS_JOINT, ABS_EEF_ACTION: nat.ACTION_ABS_EEF}

_REWARD_IDS = {
    SPARSE_REWARD: nat.REW_SPARSE,
    DENSE_POTENTIAL_REWARD: nat.REW_DENSE_POTENTIAL,
    DENSE_NEG_DISTANCE_REWARD: nat.REW_DENSE_NEG_DISTANCE,
    DENSE_BIASED_NEG_DISTANCE_REWARD: nat.REW_DENSE_BIASED_NEG_DISTANCE,
}
_AUTORESET_IDS = {"next_step": nat.AUTORESET_NEXT_STEP, "same_step": nat.AUTORESET_SAME_STEP, "disabled": nat.AUTORESET_DISABLED}


@dataclass(frozen=True)
class TaskSpec:
    name: str
    task_id: int
    obs_layout: tuple  # ((key, start, length), ...) in the reference's observation-dict order
    action_low: tuple
    action_high: tuple
    action_dtype: type  # dtype of the dm_env action spec
    reward_types: tuple
    control_timestep: float
    physics_timestep: float
    max_control_steps: int


TASKS = {
    # point_reach.py:115-118 (STATE_OBS), :204-209 action spec, :24-28 timing
    "point_mass_reach": TaskSpec(
        "point_mass_reach", nat.TASK_POINTMASS_REACH,
        (("pointmass/position", 0, 2), ("goal_position", 2, 2)),
        (-0.05, -0.05), (0.05, 0.05), np.float32,
        (SPARSE_REWARD, DENSE_POTENTIAL_REWARD, DENSE_NEG_DISTANCE_REWARD, DENSE_BIASED_NEG_DISTANCE_REWARD),
        0.1, 0.02, 50),
    # robot_reach.py:134-137 + robot.py:296-298 (BASELINE "joint-space obs"), :187-201 action spec, :59-64 timing
    "robot_reach": TaskSpec(
        "robot_reach", nat.TASK_ROBOT_REACH,
        (("ur5e/tcp_position", 0, 3), ("ur5e/joint_configuration", 3, 6), ("target_position", 9, 3)),
        (-0.1, -0.6, 0.02), (0.1, -0.4, 0.2), np.float64,
        (SPARSE_REWARD, DENSE_NEG_DISTANCE_REWARD),
        0.1, 0.005, 100),
    # robot_push_button.py:113-119 observables (which robot observable is enabled follows the action type, the
    # kernel always writes both), switch.py:86-108 (the Switch model is an unnamed mjcf root -> "unnamed_model/");
    # :176-203 action spec (ABS_JOINT default; ABS_EEF bounds are substituted in HipVectorEnv), :38-40 timing
    # robot_planar_push.py:120-126,134-138 (STATE obs: tcp_position, target_position = site.pos[:2], block_positions = body
    # xpos[:2] per block), :222-228 action spec ([-1, 1]^2, used directly as absolute TCP xy in metres, :197-201), :49-53 timing
    "robot_planar_push": TaskSpec(
        "robot_planar_push", nat.TASK_PLANAR_PUSH,
        (("ur5e/tcp_position", 0, 3), ("target_position", 3, 2), ("block_positions", 5, 4)),
        (-1.0, -1.0), (1.0, 1.0), np.float32,
        (SPARSE_REWARD, DENSE_NEG_DISTANCE_REWARD),
        0.1, 0.005, 500),
    "robot_push_button": TaskSpec(
        "robot_push_button", nat.TASK_BUTTON_PUSH,
        (("ur5e/joint_configuration", 0, 6), ("ur5e/tcp_position", 6, 3), ("unnamed_model/position", 9, 3), ("unnamed_model/active", 12, 1)),
        (-3.14,) * 6 + (0.0,), (3.14,) * 6 + (0.085,), np.float64,
        (SPARSE_REWARD,),
        0.1, 0.005, 100),
}
_BUTTON_EEF_ACTION_BOUNDS = ((-0.2, -0.6, 0.02, 0.0), (0.2, -0.3, 0.3, 0.085))  # robot_push_button.py:79,177-192


def _as_uint8_ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def visual_observation_layout(task: str, action_type: str | None = None, image_resolution: int = 64, use_wrist_camera: bool = True):
    """Keys, shapes and dtypes of a task's VISUAL observation dict in the reference's order (host logic, no device needed):
    proprioception + camera image(s) - point_reach.py:119-121, robot_reach.py:139-141, robot_push_button.py:113-124; dict order =
    composer's entity order (robot, wrist camera on the robot, scene camera). The wrist camera's key is what the reference's code
    produces at HEAD, ``ur5e/Camera/rgb_image`` (setting ``observable.name`` at robot_push_button.py:95 does not rename a key);
    the LeRobot configs still say ``ur5e_WristCamera_rgb_image`` (scripts/lerobot/configs/act_robot_button_push.yaml:24,47): see
    ``mujoco_sim_amd.recording.REFERENCE_YAML_KEY_ALIASES``."""
    if task == "point_mass_reach":
        state_keys = ("pointmass/position",)
    elif task == "robot_push_button":
        state_keys = ("ur5e/joint_configuration",) if (action_type or ABS_JOINT_ACTION) == ABS_JOINT_ACTION else ("ur5e/tcp_position",)
    else:
        state_keys = ("ur5e/tcp_position",)
    r = int(image_resolution)
    out = [(k, (n,), np.float64) for k, _, n in TASKS[task].obs_layout if k in state_keys]
    if use_wrist_camera and task == "robot_push_button":
        out.append(("ur5e/Camera/rgb_image", (r, r, 3), np.uint8))
    out.append(("Camera/rgb_image", (r, r, 3), np.uint8))
    return out


class HipVectorEnv:
    """Batched environment on one GPU. One handle = one process = one device."""

    def __init__(self, task: str, num_envs: int, device: str | int | torch.device = "cuda:0", seed: int | None = None,
                 autoreset: str = "next_step", reward_type: str | None = None, time_limit: float | None = None,
                 terminate_on_success: bool = False, env_index_offset: int = 0, kernel_variant: int | None = None,
                 observation_type: str = STATE_OBS, image_resolution: int = 64, action_type: str | None = None,
                 button_disturbances: bool = False, use_wrist_camera: bool = True, n_objects: int | None = None,
                 max_episode_steps: int | None = None, block_shape: str = "mesh", global_num_envs: int | None = None,
                 gripper_model: str = "reduced"):
        if task not in TASKS:
            raise ValueError(f"unknown task {task!r}; available: {sorted(TASKS)}")
        self.spec = TASKS[task]
        if task == "robot_push_button":
            action_type = action_type or ABS_JOINT_ACTION
            if action_type not in _ACTION_IDS:
                raise AssertionError(f"action_type must be one of {tuple(_ACTION_IDS)}")  # robot_push_button.py:33
        elif action_type is not None:
            raise ValueError(f"task {task!r} has no action_type option")
        self.action_type = action_type
        if reward_type is not None and reward_type not in self.spec.reward_types:
            raise AssertionError(f"reward_type {reward_type!r} not in {self.spec.reward_types}")  # reference asserts (point_reach.py:68)
        if autoreset not in _AUTORESET_IDS:
            raise ValueError(f"autoreset must be one of {sorted(_AUTORESET_IDS)}")
        if block_shape not in ("mesh", "box"):
            raise ValueError("block_shape must be 'mesh' (the reference's GoogleBlockProp meshes) or 'box' (round 1's stand-in)")
        self.block_shape = block_shape
        if gripper_model not in ("reduced", "articulated"):
            raise ValueError("gripper_model must be 'reduced' (DESIGN.md D-1b, nv = 6) or 'articulated' (the 2F-85 of gripper.py:36-98, nv = 14)")
        if gripper_model == "articulated" and task != "robot_push_button":
            raise ValueError("the articulated 2F-85 is built for robot_push_button only")
        self.gripper_model = gripper_model
        self.device = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise nat.MjsError("HipVectorEnv needs a HIP device (torch device 'cuda:N'); there is no CPU path")
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", self._dev_index)  # explicit ordinal: buffers, kernel launches and streams all name the same GPU
        self.num_envs = int(num_envs)
        self.autoreset = autoreset
        self.env_index_offset = int(env_index_offset)
        self._lib = nat.lib()
        # The default step kernel is chosen from the env count of the whole JOB, not of this shard (include/mjsim.h: bitwise shard
        # invariance holds only if every handle runs the same kernel shape; Robot-Reach switches shape above 16384 envs per GPU).
        # A shard (env_index_offset != 0, or a rank of a sharded job) passes the job's total as `global_num_envs`; without it the
        # lower bound env_index_offset + num_envs is all the handle knows.
        self.global_num_envs = int(global_num_envs) if global_num_envs is not None else self.env_index_offset + self.num_envs
        if self.global_num_envs < self.env_index_offset + self.num_envs:
            raise ValueError("global_num_envs is smaller than env_index_offset + num_envs")
        if kernel_variant is None:
            if task == "robot_reach" and self.global_num_envs > 16384:
                kernel_variant = nat.VARIANT_TWO_ROLES  # what MJS_VARIANT_DEFAULT runs on a handle of the whole job's size
            elif task == "robot_reach" and terminate_on_success and autoreset == "next_step":
                # episodes that end at different times: resets on workgroups of their own (mjsim.h)
                kernel_variant = nat.VARIANT_RESET_GROUPS
            else:
                kernel_variant = 0
        cfg = nat.MjsConfig(task=self.spec.task_id, num_envs=self.num_envs, device=self._dev_index,
                            reward_type=_REWARD_IDS[reward_type] if reward_type else -1, autoreset=_AUTORESET_IDS[autoreset],
                            terminate_on_success=int(terminate_on_success), env_index_offset=self.env_index_offset, kernel_variant=int(kernel_variant),
                            time_limit=float(time_limit) if time_limit is not None else -1.0,
                            action_type=_ACTION_IDS.get(action_type, 0), button_disturbances=int(bool(button_disturbances)),
                            n_objects=int(n_objects or 0), max_episode_steps=int(max_episode_steps or 0),
                            block_shape=nat.BLOCKS_BOX if block_shape == "box" else nat.BLOCKS_MESH,
                            gripper_model=nat.GRIPPER_ARTICULATED if gripper_model == "articulated" else nat.GRIPPER_REDUCED)
        h = C.c_void_p()
        nat.check(self._lib.mjs_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.obs_dim = self._lib.mjs_env_obs_dim(h)  # Planar-Push: 2 block slots for n_objects <= 2, 5 for 3..5
        self.action_dim = self._lib.mjs_action_dim_for(self.spec.task_id, _ACTION_IDS.get(action_type, 0))
        self.state_dim = self._lib.mjs_env_state_dim(h)
        self.algorithmic_bytes_per_env_step = self._lib.mjs_algorithmic_bytes_per_env_step(self.spec.task_id)
        if task == "robot_planar_push" and self.state_dim != self._lib.mjs_state_dim(self.spec.task_id):  # 5 block slots: same formula, wider rows
            S = (self.state_dim - 2 - 18) // 2  # rows of ONE world (the state also carries the next-episode slot, its progress row and its servo set-point)
            self.algorithmic_bytes_per_env_step = 8 * S + 8 * (S - 3) + 2 + 8 * self.action_dim + 8 * self.obs_dim + 25
        elif self.state_dim != self._lib.mjs_state_dim(self.spec.task_id):  # Button-Push with the articulated gripper: wider state
            S = self.state_dim - 1
            self.algorithmic_bytes_per_env_step = 8 * S + 8 * (S - 3) + 2 + 8 * self.action_dim + 8 * self.obs_dim + 25
        N, dev = self.num_envs, self.device
        # every output field is a typed view into ONE device arena (64-byte aligned slots), so that a host-side consumer (the SB3
        # adapter's numpy replay buffer, scripts/sb3/reach_sac.py:93-131) needs ONE device->host copy per step for all of them
        fields = (("obs", torch.float64, (N, self.obs_dim)), ("terminal_obs", torch.float64, (N, self.obs_dim)), ("reward", torch.float64, (N,)),
                  ("discount", torch.float64, (N,)), ("ncon", torch.int32, (N,)), ("terminated", torch.uint8, (N,)), ("truncated", torch.uint8, (N,)),
                  ("is_success", torch.uint8, (N,)), ("step_type", torch.uint8, (N,)), ("fault", torch.uint8, (N,)))
        self._arena_layout, off = [], 0
        for name, dt, shape in fields:
            nbytes = int(np.prod(shape)) * torch.empty((), dtype=dt).element_size()
            self._arena_layout.append((name, dt, shape, off, nbytes))
            off += (nbytes + 63) // 64 * 64
        self._arena = torch.zeros(off, dtype=torch.uint8, device=dev)
        self._buf = {name: self._arena[o:o + nb].view(dt).view(shape) for name, dt, shape, o, nb in self._arena_layout}
        self._buf["discount"].fill_(1.0)
        self._out = self._make_outputs(self._buf)
        self._out_ref = C.byref(self._out)  # step_flat's hot path: no per-call ctypes object construction
        self._mjs_step = self._lib.mjs_step
        # enabled observables: everything the kernel writes, except that Button-Push enables only the robot
        # observable matching its action type (robot_push_button.py:113-117)
        self.action_low, self.action_high = self.spec.action_low, self.spec.action_high
        self._state_layout = self.spec.obs_layout
        if task == "robot_planar_push":  # block_positions has 2 entries per configured block (robot_planar_push.py:178-179)
            self.n_objects = int(n_objects or 2)
            self._state_layout = self.spec.obs_layout[:2] + (("block_positions", 5, 2 * self.n_objects),)
        if task == "robot_push_button":
            drop = "ur5e/tcp_position" if action_type == ABS_JOINT_ACTION else "ur5e/joint_configuration"
            self._state_layout = tuple(e for e in self.spec.obs_layout if e[0] != drop)
            if action_type == ABS_EEF_ACTION:
                self.action_low, self.action_high = _BUTTON_EEF_ACTION_BOUNDS
        # spaces (dmc2gym.py:55-63,90,99-101)
        self.single_observation_space = Dict(OrderedDict(
            (k, Box(-np.inf, np.inf, shape=(n,), dtype=np.float64)) for k, _, n in self._state_layout))
        self.single_action_space = Box(np.asarray(self.action_low, dtype=np.float32), np.asarray(self.action_high, dtype=np.float32), dtype=np.float32)
        self.observation_space = Dict(OrderedDict((k, batch_box(s, N)) for k, s in self.single_observation_space.items()))
        self.action_space = batch_box(self.single_action_space, N)
        # visual observations (point_reach.py:119-121: camera image + pointmass position)
        if observation_type not in (STATE_OBS, VISUAL_OBS):
            raise AssertionError(f"observation_type must be one of {(STATE_OBS, VISUAL_OBS)}")
        self.observation_type = observation_type
        self.image_resolution = int(image_resolution)
        self._img = None
        self._wrist_img = None
        self.use_wrist_camera = bool(use_wrist_camera) and task == "robot_push_button"
        if observation_type == VISUAL_OBS:
            r = self.image_resolution
            self._img = torch.zeros(N, r, r, 3, dtype=torch.uint8, device=dev)
            # point_reach.py:119-121 / robot_reach.py:139-141 / robot_push_button.py:113-124: proprioception +
            # camera image(s); dict order = composer's entity order (robot, wrist camera on the robot, scene camera)
            layout = visual_observation_layout(task, action_type, r, self.use_wrist_camera)
            self._visual_keys = tuple(k for k, shape, dt in layout if dt != np.uint8)
            if self.use_wrist_camera:
                self._wrist_img = torch.zeros(N, r, r, 3, dtype=torch.uint8, device=dev)
            spaces = [(k, Box(0, 255, shape=shape, dtype=np.uint8) if dt == np.uint8 else Box(-np.inf, np.inf, shape=shape, dtype=np.float64)) for k, shape, dt in layout]
            self.single_observation_space = Dict(OrderedDict(spaces))
            self.observation_space = Dict(OrderedDict((k, batch_box(s, N)) for k, s in self.single_observation_space.items()))
        if seed is not None:
            self.seed(seed)

    # ------------------------------------------------------------------ plumbing
    @staticmethod
    def _make_outputs(buf) -> nat.MjsOutputs:
        return nat.MjsOutputs(**{k: C.c_void_p(v.data_ptr()) if v is not None else None for k, v in buf.items()})

    def _stream(self):
        # the caller's current stream, by its raw handle (the Stream-object route costs ~4 us per step on the host)
        if _RAW_STREAM is not None:
            return C.c_void_p(_RAW_STREAM(self._dev_index))
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _obs_dict(self, flat):
        if self._img is not None and flat is self._buf["obs"]:
            # entity observables first (pointmass position, camera image), as in the reference's dict order
            self.render(self.image_resolution, self.image_resolution, out=self._img)
            d = OrderedDict((k, flat[..., s:s + n]) for k, s, n in self.spec.obs_layout if k in self._visual_keys)
            if self._wrist_img is not None:
                d["ur5e/Camera/rgb_image"] = self.render(self.image_resolution, self.image_resolution, out=self._wrist_img, camera=1)
            d["Camera/rgb_image"] = self._img
            return d
        return OrderedDict((k, flat[..., s:s + n]) for k, s, n in self._state_layout)

    def _info(self, b):
        return {"is_success": b["is_success"], "discount": b["discount"], "step_type": b["step_type"], "fault": b["fault"],
                "ncon": b["ncon"], "terminal_observation": self._obs_dict(b["terminal_obs"])}

    # ----------------------------------------------------------------------- API
    def seed(self, seed: int):
        """env i <- np.random.RandomState(seed + env_index_offset + i) (dmc2gym.py:126-131; reach_sac.py:84)."""
        nat.check(self._lib.mjs_seed(self._h, C.c_uint32(int(seed) & 0xFFFFFFFF), self._stream()), self._h)

    def reset(self, seed: int | None = None, options: dict | None = None, mask: torch.Tensor | None = None):
        if seed is not None:
            self.seed(seed)
        m = None
        if mask is not None:
            m = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        nat.check(self._lib.mjs_reset(self._h, _as_uint8_ptr(m), C.byref(self._out), self._stream()), self._h)
        return self._obs_dict(self._buf["obs"]), {}

    def step(self, actions, copy: bool = False):
        """One control step of all envs. The returned tensors are VIEWS of the handle's persistent output buffers (the
        next ``step`` / ``reset`` overwrites them): a collector that keeps them across steps must pass ``copy=True`` (or
        clone what it keeps); ``step_flat`` is the zero-copy path."""
        a = torch.as_tensor(actions, device=self.device).to(torch.float64).contiguous()
        if a.shape != (self.num_envs, self.action_dim):
            raise AssertionError(f"actions must have shape {(self.num_envs, self.action_dim)}, got {tuple(a.shape)}")  # cf. point_reach.py:158
        nat.check(self._lib.mjs_step(self._h, C.c_void_p(a.data_ptr()), C.byref(self._out), self._stream()), self._h)
        b = self._buf
        obs, reward, info = self._obs_dict(b["obs"]), b["reward"], self._info(b)
        if copy:
            obs = OrderedDict((k, v.clone()) for k, v in obs.items())
            reward = reward.clone()
            info = {k: (OrderedDict((kk, vv.clone()) for kk, vv in v.items()) if isinstance(v, dict) else v.clone()) for k, v in info.items()}
        return obs, reward, b["terminated"].bool(), b["truncated"].bool(), info

    def step_flat(self, actions_f64: torch.Tensor):
        """Zero-overhead variant: float64 CUDA actions in, raw output buffers out (no copies)."""
        # argtypes are declared (c_void_p): plain ints convert without building ctypes objects
        stream = _RAW_STREAM(self._dev_index) if _RAW_STREAM is not None else torch.cuda.current_stream(self.device).cuda_stream
        rc = self._mjs_step(self._h, actions_f64.data_ptr(), self._out_ref, stream)
        if rc != 0:
            nat.check(rc, self._h)
        return self._buf

    def rollout(self, actions: torch.Tensor, keep: tuple = ("obs", "reward", "terminated", "truncated", "is_success", "step_type", "fault", "ncon", "discount")):
        """Open-loop rollout of T control steps: actions [T, N, A] -> dict of [T, N, ...] tensors."""
        a = torch.as_tensor(actions, device=self.device).to(torch.float64).contiguous()
        T = a.shape[0]
        assert a.shape == (T, self.num_envs, self.action_dim)
        buf = {k: (torch.zeros((T,) + tuple(v.shape), dtype=v.dtype, device=self.device) if k in keep else None) for k, v in self._buf.items()}
        out = self._make_outputs(buf)
        nat.check(self._lib.mjs_rollout(self._h, C.c_void_p(a.data_ptr()), T, C.byref(out), self._stream()), self._h)
        return {k: v for k, v in buf.items() if v is not None}

    def get_state(self) -> torch.Tensor:
        s = torch.empty(self.state_dim, self.num_envs, dtype=torch.float64, device=self.device)
        nat.check(self._lib.mjs_get_state(self._h, C.c_void_p(s.data_ptr()), self._stream()), self._h)
        return s

    def set_state(self, state: torch.Tensor):
        s = state.to(device=self.device, dtype=torch.float64).contiguous()
        assert s.shape == (self.state_dim, self.num_envs)
        nat.check(self._lib.mjs_set_state(self._h, C.c_void_p(s.data_ptr()), self._stream()), self._h)

    def render(self, height: int = 64, width: int = 64, out: torch.Tensor | None = None, camera: int = 0) -> torch.Tensor:
        """RGB images of all envs, uint8 [N, H, W, 3] on the GPU (replaces Camera.get_rgb_image /
        physics.render, entities/camera.py:94-103; own ray caster, D-6). camera 0 = the task's scene camera,
        1 = the Button-Push wrist camera."""
        if out is None:
            out = torch.empty(self.num_envs, height, width, 3, dtype=torch.uint8, device=self.device)
        assert out.shape == (self.num_envs, height, width, 3) and out.dtype == torch.uint8 and out.is_contiguous()
        nat.check(self._lib.mjs_render(self._h, int(camera), height, width, C.c_void_p(out.data_ptr()), self._stream()), self._h)
        return out

    def tcp_to_joints(self, tcp_positions, joint_guess):
        """Robot.get_joint_positions_from_tcp_pose for the top-down orientation (robot.py:140-151), batched on
        the device: ([n, 3], [n, 6]) -> (q [n, 6], ok [n] bool); q = the guess where no IK solution exists."""
        pos = torch.as_tensor(tcp_positions, device=self.device).to(torch.float64).contiguous()
        guess = torch.as_tensor(joint_guess, device=self.device).to(torch.float64).contiguous()
        n = pos.shape[0]
        assert pos.shape == (n, 3) and guess.shape == (n, 6)
        q = torch.empty(n, 6, dtype=torch.float64, device=self.device)
        ok = torch.empty(n, dtype=torch.uint8, device=self.device)
        nat.check(self._lib.mjs_ur5e_tcp_to_joints(C.c_void_p(pos.data_ptr()), C.c_void_p(guess.data_ptr()), C.c_void_p(q.data_ptr()),
                                                   C.c_void_p(ok.data_ptr()), n, self._stream()), self._h)
        return q, ok.bool()

    def get_rng_state(self):
        """(mt uint32 [624, N], pos int32 [N]) — the per-env numpy-legacy MT19937 streams."""
        mt = torch.empty(624, self.num_envs, dtype=torch.int32, device=self.device)
        pos = torch.empty(self.num_envs, dtype=torch.int32, device=self.device)
        nat.check(self._lib.mjs_get_rng_state(self._h, C.c_void_p(mt.data_ptr()), C.c_void_p(pos.data_ptr()), self._stream()), self._h)
        return mt, pos

    def set_rng_state(self, rng_state):
        mt, pos = (t.to(device=self.device, dtype=torch.int32).contiguous() for t in rng_state)
        assert mt.shape == (624, self.num_envs) and pos.shape == (self.num_envs,)
        nat.check(self._lib.mjs_set_rng_state(self._h, C.c_void_p(mt.data_ptr()), C.c_void_p(pos.data_ptr()), self._stream()), self._h)

    @property
    def flat_obs(self) -> torch.Tensor:
        return self._buf["obs"]

    def close(self):
        if getattr(self, "_h", None):
            torch.cuda.synchronize(self.device)
            self._lib.mjs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
