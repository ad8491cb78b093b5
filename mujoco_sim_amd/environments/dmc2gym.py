"""dm_env -> gymnasium adapter, same interface as the reference's
``mujoco_sim/environments/dmc2gym.py:78-168`` (``DMCEnvironmentAdapter``), backed by the HIP
engine instead of dm_control + MuJoCo.

``HipEnvironment`` plays the role of ``dm_control.composer.Environment`` for ONE environment
(``step``/``reset`` -> dm_env-style ``TimeStep``, ``action_spec``/``observation_spec``,
``task``); it is a 1-env view of a :class:`HipVectorEnv`. For throughput use
:class:`HipVectorEnv` directly.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import NamedTuple

import numpy as np
import torch

from .. import _native as nat
from ..spaces import Box, Dict
from ..vector_env import STATE_OBS, HipVectorEnv
from .tasks.point_reach import ArraySpec, BoundedArraySpec

try:  # pragma: no cover
    import gymnasium as _gym

    _EnvBase = _gym.Env
except Exception:  # noqa: BLE001
    _EnvBase = object


class TimeStep(NamedTuple):
    step_type: int
    reward: float | None
    discount: float | None
    observation: dict

    def first(self):
        return self.step_type == nat.STEP_FIRST

    def mid(self):
        return self.step_type == nat.STEP_MID

    def last(self):
        return self.step_type == nat.STEP_LAST


class HipEnvironment:
    """Single-environment dm_env-style front end (role of composer.Environment)."""

    def __init__(self, task, time_limit: float = float("inf"), device="cuda:0", strip_singleton_obs_buffer_dim: bool = True,
                 random_state=None):
        self.task = task
        self._time_limit = time_limit
        kwargs = {}
        if hasattr(task, "config") and getattr(task.config, "terminate_on_success", False):
            kwargs["terminate_on_success"] = True
        if task.task_name == "robot_planar_push":
            kwargs["n_objects"] = task.config.n_objects
            kwargs["max_episode_steps"] = task.config.max_control_steps_per_episode
        if hasattr(task, "action_type") and task.task_name == "robot_push_button":
            kwargs["action_type"] = task.action_type
            kwargs["button_disturbances"] = task.button_disturbances
            kwargs["use_wrist_camera"] = task.use_wrist_camera
        self._venv = HipVectorEnv(task.task_name, 1, device=device, autoreset="next_step", reward_type=task.reward_type,
                                  time_limit=(1e300 if np.isinf(time_limit) else time_limit),
                                  observation_type=getattr(task, "observation_type", STATE_OBS),
                                  image_resolution=getattr(task, "image_resolution", 64), **kwargs)
        self._random_state = None
        # one pinned host mirror of the env's output arena and one pinned action buffer (the SB3 adapter's scheme): a step costs
        # ONE host->device copy (the action) and ONE device->host copy (every output field) instead of a sync per scalar
        v = self._venv
        self._host_arena = torch.empty(v._arena.shape, dtype=torch.uint8, pin_memory=True)
        self._host = {name: self._host_arena[o:o + nb].view(dt).view(shape).numpy() for name, dt, shape, o, nb in v._arena_layout}
        self._act_host = torch.empty(1, v.action_dim, dtype=torch.float64, pin_memory=True)
        self._act_dev = torch.empty(1, v.action_dim, dtype=torch.float64, device=v.device)
        self._visual = v._img is not None
        self._layout = [e for e in v.spec.obs_layout if e[0] in v._visual_keys] if self._visual else list(v._state_layout)
        if random_state is not None:
            self.seed(random_state)

    # composer.Environment stores a RandomState; only integer seeding is supported on the device
    def seed(self, seed: int):
        self._venv.seed(int(seed))

    def control_timestep(self):
        return self.task.control_timestep

    def action_spec(self):
        return self.task.action_spec()

    def observation_spec(self):
        spec = OrderedDict()
        for k, box in self._venv.single_observation_space.items():
            if box.dtype == np.uint8:  # RGBObservable.array_spec (entities/camera.py:146-150)
                spec[k] = BoundedArraySpec(box.shape, np.uint8, 0, 255, name=k)
            else:
                spec[k] = ArraySpec(box.shape, np.float64, name=k)
        return spec

    def _timestep(self) -> TimeStep:
        v = self._venv
        images = None
        if self._visual:  # renders the camera image(s) on the env's stream; they are copied on their own
            images = [(k, t) for k, t in v._obs_dict(v._buf["obs"]).items() if t.dtype == torch.uint8]
        self._host_arena.copy_(v._arena, non_blocking=True)
        if images is not None:
            images = [(k, t[0].cpu().numpy().copy()) for k, t in images]
        torch.cuda.current_stream(v.device).synchronize()
        h = self._host
        st = int(h["step_type"][0])
        flat = h["obs"][0]
        obs = OrderedDict((k, flat[s:s + n].copy()) for k, s, n in self._layout)
        if images is not None:
            for k, img in images:
                obs[k] = img
            obs = OrderedDict((k, obs[k]) for k in v.single_observation_space.keys())  # the reference's dict order
        self._is_success = bool(h["is_success"][0])
        if st == nat.STEP_FIRST:
            return TimeStep(st, None, None, obs)
        return TimeStep(st, float(h["reward"][0]), float(h["discount"][0]), obs)

    def reset(self) -> TimeStep:
        self._venv.reset()
        return self._timestep()

    def step(self, action) -> TimeStep:
        a = np.asarray(action)
        assert a.shape == (self._venv.action_dim,)  # point_reach.py:158 / robot_reach.py:167
        self._act_host.numpy()[0] = a
        self._act_dev.copy_(self._act_host, non_blocking=True)
        self._venv.step_flat(self._act_dev)
        return self._timestep()

    @property
    def is_success(self) -> bool:
        return self._is_success

    def close(self):
        self._venv.close()


def convert_spec_to_box(s):
    """dmc2gym.py:55-63"""
    if hasattr(s, "minimum"):
        zeros = np.zeros(s.shape, dtype=s.dtype)
        return Box(s.minimum + zeros, s.maximum + zeros, dtype=s.dtype)
    bound = np.inf * np.ones(s.shape, dtype=s.dtype)
    return Box(-bound, bound, dtype=s.dtype)


def _convert_specs_to_flattened_box(specs, dtype):
    """dmc2gym.py:18-52: concatenated flat bounds, Box dtype float32."""
    mins, maxs = [], []
    for s in specs:
        assert s.dtype == np.float64 or s.dtype == np.float32
        dim = int(np.prod(s.shape))
        if hasattr(s, "minimum"):
            zeros = np.zeros(dim, dtype=dtype)
            mins.append(s.minimum + zeros)
            maxs.append(s.maximum + zeros)
        else:
            bound = np.inf * np.ones(dim, dtype=dtype)
            mins.append(-bound)
            maxs.append(bound)
    low = np.concatenate(mins, axis=0).astype(dtype)
    high = np.concatenate(maxs, axis=0).astype(dtype)
    assert low.shape == high.shape
    return Box(low, high, dtype=np.float32)


def _flatten_obs(obs: dict) -> np.ndarray:
    """dmc2gym.py:66-75"""
    return np.concatenate([np.array([v]) if np.isscalar(v) else v.ravel() for v in obs.values()], axis=0)


class DMCEnvironmentAdapter(_EnvBase):
    """Same constructor, properties and step/reset/seed semantics as dmc2gym.py:78-168."""

    def __init__(self, env: HipEnvironment, flatten_observation_space: bool = False, render_camera_id=-1,
                 render_dims: tuple = (256, 256)):
        self.flatten_observation_space = flatten_observation_space
        self._camera_id = render_camera_id
        self.render_dims = render_dims
        self._env = env
        self._action_space = _convert_specs_to_flattened_box([self._env.action_spec()], np.float32)
        if flatten_observation_space:
            self._observation_space = _convert_specs_to_flattened_box(self._env.observation_spec().values(), np.float64)
        else:
            self._observation_space = Dict(OrderedDict((k, convert_spec_to_box(s)) for k, s in self._env.observation_spec().items()))

    def __getattr__(self, name):
        if name.startswith("__") or name == "_env":
            raise AttributeError(name)
        return getattr(self._env, name)

    def _get_obs(self, time_step):
        obs = time_step.observation
        return _flatten_obs(obs) if self.flatten_observation_space else obs

    @property
    def dmc_env(self):
        return self._env

    @property
    def observation_space(self):
        return self._observation_space

    @property
    def action_space(self):
        return self._action_space

    def seed(self, seed):
        self._env.seed(seed)
        self._action_space.seed(seed)
        self._observation_space.seed(seed)

    def step(self, action):
        info = {}
        time_step = self._env.step(action)
        reward = time_step.reward
        obs = self._get_obs(time_step)
        truncated = time_step.last() and time_step.discount > 0
        terminated = time_step.last() and time_step.discount == 0
        if self._env.task.task_name in ("point_mass_reach", "robot_push_button"):
            # only tasks that define is_goal_reached report it (dmc2gym.py:149-150; point_reach.py:195-196,
            # robot_push_button.py:205-210)
            info["is_success"] = self._env.is_success * 1.0
        info["discount"] = time_step.discount
        return obs, reward, terminated, truncated, info

    def reset(self, seed: int = None, options: dict = None):
        if seed is not None:
            self.seed(seed)
        time_step = self._env.reset()
        return self._get_obs(time_step), {}

    def render(self, mode="rgb_array"):
        """dmc2gym.py:165-168. The reference renders camera `render_camera_id` (default -1: MuJoCo's free
        camera); here every id maps to the task's scene camera (own ray caster)."""
        assert mode == "rgb_array", "only support rgb_array mode, given %s" % mode
        height, width = self.render_dims
        return self._env._venv.render(height, width)[0].cpu().numpy()

    def close(self):
        self._env.close()
