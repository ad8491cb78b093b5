"""Pointmass-Reach task description (host side).

Mirrors the constructor surface of the reference's
``mujoco_sim/environments/tasks/point_reach.py:31-113`` (same kwargs, same assertion
behaviour, same module-level constants). The scene, physics and reward logic themselves run in
the HIP kernel (csrc/mjs_pointmass.h); this class only carries configuration.
"""
from __future__ import annotations

import numpy as np

from ...vector_env import (DENSE_BIASED_NEG_DISTANCE_REWARD, DENSE_NEG_DISTANCE_REWARD, DENSE_POTENTIAL_REWARD, SPARSE_REWARD,
                           STATE_OBS, VISUAL_OBS)

REWARD_TYPES = (SPARSE_REWARD, DENSE_POTENTIAL_REWARD, DENSE_NEG_DISTANCE_REWARD, DENSE_BIASED_NEG_DISTANCE_REWARD)
OBSERVATION_TYPES = (STATE_OBS, VISUAL_OBS)

PHYSICS_TIMESTEP = 0.02
CONTROL_TIMESTEP = 0.1
MAX_CONTROL_STEPS_PER_EPISODE = 50
GOAL_DISTANCE_THRESHOLD = 0.02
MAX_STEP_SIZE = 0.05


class PointMassReachTask:
    MAX_CONTROL_STEPS_PER_EPISODE = MAX_CONTROL_STEPS_PER_EPISODE
    CONTROL_TIMESTEP = CONTROL_TIMESTEP
    task_name = "point_mass_reach"

    def __init__(self, reward_type: str = DENSE_BIASED_NEG_DISTANCE_REWARD, observation_type: str = VISUAL_OBS,
                 image_resolution: int = 64) -> None:
        assert reward_type in REWARD_TYPES
        assert observation_type in OBSERVATION_TYPES
        self.reward_type = reward_type
        self.observation_type = observation_type
        self.image_resolution = image_resolution
        self.physics_timestep = PHYSICS_TIMESTEP
        self.control_timestep = CONTROL_TIMESTEP

    def action_spec(self, physics=None):
        bound = np.array([MAX_STEP_SIZE, MAX_STEP_SIZE])
        return BoundedArraySpec((2,), np.float32, -bound, bound)

    def create_random_policy(self):
        spec = self.action_spec()

        def random_policy(time_step):
            return np.random.uniform(spec.minimum, spec.maximum, spec.shape)

        return random_policy

    # point_reach.py:227-240 (the reference spells it `create_demonstation_policy`; both names are provided)
    def demonstration_actions(self, venv, noise: float = 0.0):
        """Batched form for a HipVectorEnv of this task: the step towards the goal, rescaled so that its largest
        component is MAX_STEP_SIZE (optionally multiplied by 1 + N(0, noise) first, as the reference does)."""
        import torch

        obs = venv.flat_obs  # [N, 4] = pointmass position (2), goal position (2)
        action = obs[:, 2:4] - obs[:, 0:2]
        if noise > 0:
            action = action * (1 + noise * torch.randn_like(action))
        return action * (MAX_STEP_SIZE / action.abs().amax(dim=1, keepdim=True).clamp_min(1e-300))

    def create_demonstration_policy(self, environment, noise: float = 0.0):
        venv = environment._venv if hasattr(environment, "_venv") else environment

        def policy(time_step=None):
            return self.demonstration_actions(venv, noise)[0].cpu().numpy().astype(np.float32)

        return policy

    create_demonstation_policy = create_demonstration_policy


class BoundedArraySpec:
    """Stand-in for dm_env.specs.BoundedArray (shape, dtype, minimum, maximum)."""

    def __init__(self, shape, dtype, minimum, maximum, name=None):
        self.shape, self.dtype, self.name = tuple(shape), np.dtype(dtype), name
        self.minimum = np.broadcast_to(np.asarray(minimum, dtype=self.dtype), self.shape).copy()
        self.maximum = np.broadcast_to(np.asarray(maximum, dtype=self.dtype), self.shape).copy()


class ArraySpec:
    """Stand-in for dm_env.specs.Array."""

    def __init__(self, shape, dtype, name=None):
        self.shape, self.dtype, self.name = tuple(shape), np.dtype(dtype), name
