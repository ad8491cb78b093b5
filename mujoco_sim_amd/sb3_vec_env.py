"""Stable-Baselines3 ``VecEnv`` surface over :class:`HipVectorEnv`.

The reference trains with ``DummyVecEnv`` / ``SubprocVecEnv`` over N single-env adapters
(scripts/sb3/reach_sac.py:53-96, scripts/sb3/planar_push.py:55-98: one process per env, sub-env
``rank`` seeded ``seed + rank``). This class is the drop-in for that construction: the same
attributes and methods SB3's algorithms call (``num_envs``, ``observation_space``,
``action_space``, ``reset``, ``step_async`` / ``step_wait`` / ``step``, ``seed``, ``close``,
``get_attr`` / ``set_attr`` / ``env_method``, ``env_is_wrapped``), with SB3's conventions:
numpy dict observations, same-step auto-reset with ``infos[i]["terminal_observation"]``,
``infos[i]["TimeLimit.truncated"]``, ``dones = terminated | truncated``. All N envs live on one
GPU and are stepped by one kernel launch; the only host traffic per step is the batched copy
SB3's numpy replay buffer needs anyway.

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
        self._actions = torch.as_tensor(np.asarray(actions), device=self.venv.device).to(torch.float64)

    def step_wait(self):
        b = self.venv.step_flat(self._actions.contiguous())
        # one device->host copy per field (what a numpy replay buffer needs)
        obs = self._obs_numpy(b["obs"])
        reward = b["reward"].cpu().numpy().astype(np.float32)
        terminated = b["terminated"].cpu().numpy().astype(bool)
        truncated = b["truncated"].cpu().numpy().astype(bool)
        success = b["is_success"].cpu().numpy()
        dones = terminated | truncated
        infos = [{"is_success": float(success[i])} for i in range(self.num_envs)]
        self._ep_return += reward
        self._ep_length += 1
        if dones.any():
            term_obs = b["terminal_obs"].cpu().numpy()
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
