"""Planar-Push task description (host side).

Mirrors ``mujoco_sim/environments/tasks/robot_planar_push.py:28-73`` (RobotPushConfig dataclass: same fields and
defaults: ``n_objects`` = 5, the registered env and BASELINE config 4 use 2) and the task surface at :76-241 (``action_spec``, ``create_random_policy``). Physics and task logic run in
csrc/mjs_push.h with the INTENDED semantics where the reference is broken at HEAD (seeded object
draws, ``episode_step`` limit); blocks are box stand-ins for the cube mesh.
"""
from __future__ import annotations

import dataclasses

import numpy as np

from .point_reach import BoundedArraySpec


@dataclasses.dataclass
class RobotPushConfig:
    SPARSE_REWARD = "sparse_reward"
    DENSE_NEG_DISTANCE_REWARD = "dense_negative_distance_reward"
    STATE_OBS = "state_observations"
    VISUAL_OBS = "visual_observations"
    REWARD_TYPES = (SPARSE_REWARD, DENSE_NEG_DISTANCE_REWARD)
    OBSERVATION_TYPES = (STATE_OBS, VISUAL_OBS)

    reward_type: str = None
    observation_type: str = None
    max_step_size: float = 0.05
    physics_timestep: float = 0.005
    control_timestep: float = 0.1
    max_control_steps_per_episode: int = 500
    goal_distance_threshold: float = 0.02
    image_resolution: int = 64
    nearest_object_reward_coefficient: float = 0.1
    target_radius = 0.05
    n_objects: int = 5  # robot_planar_push.py:61; 1..2 run the 2-slot kernel, 3..5 the 5-slot kernel

    def __post_init__(self):
        self.reward_type = self.reward_type or RobotPushConfig.DENSE_NEG_DISTANCE_REWARD
        self.observation_type = self.observation_type or RobotPushConfig.STATE_OBS
        assert self.observation_type in RobotPushConfig.OBSERVATION_TYPES
        assert self.reward_type in RobotPushConfig.REWARD_TYPES
        if not 1 <= self.n_objects <= 5:
            raise NotImplementedError("n_objects must be 1..5 (MJS_PP_MAX_OBJECTS)")
        if self.nearest_object_reward_coefficient != 0.1 or self.physics_timestep != 0.005 or self.control_timestep != 0.1:
            raise NotImplementedError("timesteps and the reward coefficient are compiled-in scene constants")


class RobotPushTask:
    task_name = "robot_planar_push"

    def __init__(self, config: RobotPushConfig | None = None) -> None:
        self.config = config or RobotPushConfig()
        self.physics_timestep = self.config.physics_timestep
        self.control_timestep = self.config.control_timestep
        self.reward_type = self.config.reward_type
        self.observation_type = self.config.observation_type
        self.image_resolution = self.config.image_resolution

    @property
    def CONTROL_TIMESTEP(self):
        return self.config.control_timestep

    def action_spec(self, physics=None):
        # robot_planar_push.py:222-228: [-1, 1]^2; before_step uses the value as the absolute TCP xy in metres (:197-201)
        return BoundedArraySpec((2,), np.float32, [-1.0, -1.0], [1.0, 1.0])

    def create_random_policy(self):
        spec = self.action_spec()

        def random_policy(time_step):
            return np.random.uniform(spec.minimum, spec.maximum, spec.shape).astype(np.float32)

        return random_policy
