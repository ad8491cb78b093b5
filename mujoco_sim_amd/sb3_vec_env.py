"""Stable-Baselines3 ``VecEnv`` surface over :class:`HipVectorEnv`.

The reference trains with ``DummyVecEnv`` / ``SubprocVecEnv`` over N single-env adapters
(scripts/sb3/reach_sac.py:53-96, scripts/sb3/planar_push.py:55-98: one process per env, sub-env
``rank`` seeded ``seed + rank``). This class is the drop-in for that construction: the same
attributes and methods SB3's algorithms call (``num_envs``, ``observation_space``,
``action_space``, ``reset``, ``step_async`` / ``step_wait`` / ``step``, ``seed``, ``close``,
``get_attr`` / ``set_attr`` / ``env_method``, ``env_is_wrapped``), with SB3's conventions:
numpy dict observations, same-step auto-reset with ``infos[i]["terminal_observation"]``,
``infos[i]["TimeLimit.truncated"]``, ``dones = terminated | truncated``. All N envs live on one
GPU and are stepped by one kernel launch; the host traffic per step is ONE device->host copy of the
output arena (observations, rewards, flags, terminal observations: all fields are views of one
device buffer) into one pinned staging buffer, what SB3's numpy replay buffer needs anyway; the
per-env ``infos`` dicts are built from one vectorised ``is_success`` array and only envs that
ended get the extra keys (tools/host_rate.py: host time per ``step_wait``).

Subclasses ``stable_baselines3.common.vec_env.VecEnv`` when SB3 is importable (it is not part
of this repository's environment), otherwise it is duck-typed.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from .vector_env import HipVectorEnv

try:  # pragma: no cover - depends on the environment
    from stable_baselines3.common.vec_env import VecEnv as _Base  # type: ignore

    _HAVE_SB3 = True
except Exception:  # noqa: BLE001
    _Base = object
    _HAVE_SB3 = False


class HipSB3VecEnv(_Base):
    def __init__(self, task: str, num_envs: int, device="cuda:0", seed: int | None = None, **task_kwargs):
        self.venv = HipVectorEnv(task, num_envs, device=device, seed=seed, autoreset="same_step", **task_kwargs)
        obs_space, act_space = self.venv.single_observation_space, self.venv.single_action_space
        if _HAVE_SB3:
            super().__init__(num_envs, obs_space, act_space)
        else:
            self.num_envs, self.observation_space, self.action_space = num_envs, obs_space, act_space
        self.render_mode = None
        self._actions = None
        # VecMonitor-equivalent episode statistics (what SB3's Monitor wrapper of scripts/sb3/reach_sac.py:77 reports)
        self._ep_return = np.zeros(num_envs, dtype=np.float64)
        self._ep_length = np.zeros(num_envs, dtype=np.int64)
        self._lo = torch.as_tensor(np.asarray(self.venv.action_low), dtype=torch.float64, device=self.venv.device)
        self._hi = torch.as_tensor(np.asarray(self.venv.action_high), dtype=torch.float64, device=self.venv.device)
        # pinned host mirror of the env's output arena + numpy views of its fields (re-used every step)
        self._host_arena = torch.empty(self.venv._arena.shape, dtype=torch.uint8, pin_memory=True)
        self._host = {name: self._host_arena[o:o + nb].view(dt).view(shape).numpy() for name, dt, shape, o, nb in self.venv._arena_layout}
        self._act_host = torch.empty(num_envs, self.venv.action_dim, dtype=torch.float64, pin_memory=True)
        self._act_dev = torch.empty(num_envs, self.venv.action_dim, dtype=torch.float64, device=self.venv.device)

    # ------------------------------------------------------------------ helpers
    def _obs_numpy(self, flat: torch.Tensor):
        # the observation dict the spaces advertise: state keys, or (VISUAL_OBS) proprioception + rendered camera image(s)
        return OrderedDict((k, v.cpu().numpy().copy()) for k, v in self.venv._obs_dict(flat).items())

    def _terminal_obs_numpy(self, term_obs_host, i):
        keys = self.venv._visual_keys if self.venv._img is not None else None
        layout = [e for e in self.venv.spec.obs_layout if e[0] in keys] if keys is not None else self.venv._state_layout
        # (the cameras show the post-reset state after a same-step reset: the terminal frame is not re-rendered)
        return OrderedDict((k, term_obs_host[i, s:s + n].copy()) for k, s, n in layout)

    # ---------------------------------------------------------------- VecEnv API
    def seed(self, seed: int | None = None):
        if seed is not None:
            self.venv.seed(seed)  # env i <- RandomState(seed + i), as create_env(rank, seed, ...) does
        return [None if seed is None else seed + i for i in range(self.num_envs)]

    def reset(self):
        self.venv.reset()
        self._ep_return[:] = 0
        self._ep_length[:] = 0
        return self._obs_numpy(self.venv.flat_obs)

    def step_async(self, actions):
        # one pinned staging buffer, one host->device copy on the env's stream
        self._act_host.numpy()[...] = np.asarray(actions, dtype=np.float64).reshape(self.num_envs, self.venv.action_dim)
        self._act_dev.copy_(self._act_host, non_blocking=True)
        self._actions = self._act_dev

    def step_wait(self):
        b = self.venv.step_flat(self._actions)
        visual = self.venv._img is not None
        if visual:
            obs = self._obs_numpy(b["obs"])  # renders the camera image(s); image tensors are copied on their own
        # ONE device->host copy for every output field
        self._host_arena.copy_(self.venv._arena, non_blocking=True)
        torch.cuda.current_stream(self.venv.device).synchronize()
        h = self._host
        if not visual:
            flat = h["obs"]
            obs = OrderedDict((k, flat[:, s:s + n].copy()) for k, s, n in self.venv._state_layout)
        reward = h["reward"].astype(np.float32)
        terminated = h["terminated"].astype(bool)
        truncated = h["truncated"].astype(bool)
        dones = terminated | truncated
        # SB3 wants one dict per env (a shared dict would alias): built from ONE vectorised array; extra keys only where an episode ended
        infos = [{"is_success": x} for x in h["is_success"].astype(np.float64).tolist()]
        self._ep_return += reward
        self._ep_length += 1
        if dones.any():
            term_obs = h["terminal_obs"]
            for i in np.nonzero(dones)[0]:
                infos[i]["terminal_observation"] = self._terminal_obs_numpy(term_obs, i)
                infos[i]["TimeLimit.truncated"] = bool(truncated[i] and not terminated[i])
                infos[i]["episode"] = {"r": float(self._ep_return[i]), "l": int(self._ep_length[i])}  # Monitor's keys
                self._ep_return[i] = 0
                self._ep_length[i] = 0
        return obs, reward, dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.venv.close()

    def get_attr(self, attr_name, indices=None):
        n = self.num_envs if indices is None else len(self._indices(indices))
        return [getattr(self.venv, attr_name)] * n

    def set_attr(self, attr_name, value, indices=None):
        setattr(self.venv, attr_name, value)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        n = self.num_envs if indices is None else len(self._indices(indices))
        return [getattr(self.venv, method_name)(*method_args, **method_kwargs)] * n

    def env_is_wrapped(self, wrapper_class, indices=None):
        n = self.num_envs if indices is None else len(self._indices(indices))
        return [False] * n

    def _indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        return [indices] if isinstance(indices, int) else list(indices)

    def get_images(self):
        return list(self.venv.render(256, 256).cpu().numpy())

    def render(self, mode=None):
        return self.get_images()[0]
