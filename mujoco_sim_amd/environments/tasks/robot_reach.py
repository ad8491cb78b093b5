"""Robot-Reach task description (host side).

Mirrors ``mujoco_sim/environments/tasks/robot_reach.py:33-83`` (RobotReachConfig dataclass:
same fields and defaults) and the task constructor at :85-132. Physics and task logic run in
csrc/mjs_reach.h.
"""
from __future__ import annotations

import dataclasses

import numpy as np

from .point_reach import BoundedArraySpec


@dataclasses.dataclass
class RobotReachConfig:
    SPARSE_REWARD = "sparse_reward"
    DENSE_NEG_DISTANCE_REWARD = "dense_negative_distance_reward"
    STATE_OBS = "state_observations"
    VISUAL_OBS = "visual_observations"
    REL_EEF_ACTION = "relative_eef_action"
    ABS_EEF_ACTION = "absolute_eef_action"
    REL_JOIN_ACTION = "relative_joint_action"
    ABS_JOIN_ACTION = "absolute_joint_action"
    REWARD_TYPES = (SPARSE_REWARD, DENSE_NEG_DISTANCE_REWARD)
    OBSERVATION_TYPES = (STATE_OBS, VISUAL_OBS)
    ACTION_TYPES = (REL_EEF_ACTION, ABS_EEF_ACTION, REL_JOIN_ACTION, ABS_JOIN_ACTION)

    reward_type: str = None
    observation_type: str = None
    action_type: str = None
    max_step_size: float = 0.05
    physics_timestep: float = 0.005
    control_timestep: float = 0.1
    max_control_steps_per_episode: int = 100
    image_resolution: int = 96
    goal_distance_threshold: float = 0.02
    target_radius = 0.03
    # opt-in: the reference task never terminates on success
    terminate_on_success: bool = False

    def __post_init__(self):
        self.reward_type = self.reward_type or RobotReachConfig.DENSE_NEG_DISTANCE_REWARD
        self.observation_type = self.observation_type or RobotReachConfig.STATE_OBS
        self.action_type = self.action_type or RobotReachConfig.ABS_EEF_ACTION
        assert self.observation_type in RobotReachConfig.OBSERVATION_TYPES
        assert self.reward_type in RobotReachConfig.REWARD_TYPES
        assert self.action_type in RobotReachConfig.ACTION_TYPES


class RobotReachTask:
    task_name = "robot_reach"

    def __init__(self, config: RobotReachConfig | None = None) -> None:
        self.config = config or RobotReachConfig()
        self.physics_timestep = self.config.physics_timestep
        self.control_timestep = self.config.control_timestep
        self.reward_type = self.config.reward_type
        self.observation_type = self.config.observation_type
        self.image_resolution = self.config.image_resolution
        if self.config.action_type != RobotReachConfig.ABS_EEF_ACTION:
            # the reference's before_step only implements the absolute-EEF path (robot_reach.py:159-169)
            raise NotImplementedError("only ABS_EEF_ACTION is implemented (as in the reference)")

    @property
    def CONTROL_TIMESTEP(self):
        return self.config.control_timestep

    def action_spec(self, physics=None):
        return BoundedArraySpec((3,), np.float64, [-0.1, -0.6, 0.02], [0.1, -0.4, 0.2])

    def create_random_policy(self):
        spec = self.action_spec()

        def random_policy(time_step):
            return np.random.uniform(spec.minimum, spec.maximum, spec.shape)

        return random_policy
