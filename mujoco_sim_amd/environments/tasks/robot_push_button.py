"""Button-Push task description (host side).

Mirrors ``mujoco_sim/environments/tasks/robot_push_button.py:20-108`` (class constants, constructor
arguments and defaults), ``:172-203`` (action spec), ``:221-229`` (random policy) and ``:231-296``
(scripted demonstration policy). Physics, the switch state machine and the task logic run in
csrc/mjs_button.h; the policies here only read the env's observation buffer.
"""
from __future__ import annotations

import numpy as np
import torch

from .point_reach import BoundedArraySpec


class RobotPushButtonTask:
    task_name = "robot_push_button"

    SPARSE_REWARD = "sparse_reward"
    STATE_OBS = "state_observations"
    VISUAL_OBS = "visual_observations"
    ABS_EEF_ACTION = "absolute_eef_action"
    ABS_JOINT_ACTION = "absolute_joint_action"
    REWARD_TYPES = SPARSE_REWARD
    OBSERVATION_TYPES = (STATE_OBS, VISUAL_OBS)
    ACTION_TYPES = (ABS_EEF_ACTION, ABS_JOINT_ACTION)

    MAX_STEP_SIZE: float = 0.05
    PHYSICS_TIMESTEP: float = 0.005
    CONTROL_TIMESTEP: float = 0.1
    MAX_CONTROL_STEPS_PER_EPISODE: int = 100
    GOAL_DISTANCE_THRESHOLD: float = 0.05
    TARGET_RADIUS = 0.03

    def __init__(self, reward_type: str = SPARSE_REWARD, observation_type: str = VISUAL_OBS, action_type: str = ABS_JOINT_ACTION,
                 image_resolution: int = 96, use_wrist_camera: bool = True, button_disturbances: bool = False) -> None:
        assert reward_type == RobotPushButtonTask.SPARSE_REWARD
        assert observation_type in RobotPushButtonTask.OBSERVATION_TYPES
        assert action_type in RobotPushButtonTask.ACTION_TYPES
        self.reward_type = reward_type
        self.observation_type = observation_type
        self.action_type = action_type
        self.image_resolution = image_resolution
        self.use_wrist_camera = use_wrist_camera
        self.button_disturbances = button_disturbances
        self.physics_timestep = self.PHYSICS_TIMESTEP
        self.control_timestep = self.CONTROL_TIMESTEP
        self.robot_end_position = np.array([-0.3, -0.2, 0.3])
        # scene / wrist camera poses are fixed in include/mjs_scene_spec.h (MJS_BP_CAM_*, MJS_WCAM_*): the
        # reference's constructor defaults (robot_push_button.py:52-57)

    def action_spec(self, physics=None):
        if self.action_type == RobotPushButtonTask.ABS_EEF_ACTION:
            return BoundedArraySpec((4,), np.float64, [-0.2, -0.6, 0.02, 0.0], [0.2, -0.3, 0.3, 0.085])
        return BoundedArraySpec((7,), np.float64, [-3.14] * 6 + [0.0], [3.14] * 6 + [0.085])

    def create_random_policy(self):
        spec = self.action_spec()

        def random_policy(time_step):
            return np.random.uniform(spec.minimum, spec.maximum, spec.shape)

        return random_policy

    # ------------------------------------------------------------------ scripted demonstration policy
    def demonstration_actions(self, venv) -> torch.Tensor:
        """One batched step of the reference's demonstration policy for every env of `venv`
        (a HipVectorEnv of this task): approach above the switch, press straight down, then move to the end
        pose once the switch is active; Cartesian speed limit 0.5 m/s; gripper closed."""
        obs = venv.flat_obs  # [N, 13] = joints(6) tcp(3) switch position(3) active(1), all on the device
        q, tcp, sw, active = obs[:, 0:6], obs[:, 6:9], obs[:, 9:12], obs[:, 12] > 0.5
        planar = torch.linalg.norm(tcp[:, :2] - sw[:, :2], dim=1)
        press = (~active) & (tcp[:, 2] > sw[:, 2]) & (planar < 0.01)
        goal = sw.clone()
        goal[:, 2] += 0.05                                                   # phase 1: hover above the button ...
        too_low = tcp[:, 2] < sw[:, 2] + 0.02
        goal[:, :2] = torch.where(too_low[:, None], tcp[:, :2], goal[:, :2])  # ... rising first when below its top
        goal = torch.where(press[:, None], sw, goal)                         # phase 2: straight down onto it
        end = torch.as_tensor(self.robot_end_position, dtype=obs.dtype, device=obs.device).repeat(len(obs), 1)
        end[:, 2] = torch.where(planar < 0.05, sw[:, 2] + 0.1, end[:, 2])    # phase 3: leave without re-touching
        goal = torch.where(active[:, None], end, goal)
        diff = goal - tcp
        biggest = diff.abs().amax(dim=1)
        limit = 0.5 * self.control_timestep
        diff = diff * torch.where(biggest > limit, limit / biggest.clamp_min(1e-300), torch.ones_like(biggest))[:, None]
        target = tcp + diff
        closed = torch.zeros(len(obs), 1, dtype=obs.dtype, device=obs.device)
        if self.action_type == RobotPushButtonTask.ABS_JOINT_ACTION:
            joints, _ = venv.tcp_to_joints(target, q)
            return torch.cat([joints, closed], dim=1)
        return torch.cat([target, closed], dim=1)

    def create_demonstration_policy(self, environment):
        """Single-env form with the reference's signature: policy(time_step) -> action (numpy)."""
        venv = environment._venv if hasattr(environment, "_venv") else environment

        def demonstration_policy(time_step=None):
            return self.demonstration_actions(venv)[0].cpu().numpy()

        return demonstration_policy
