"""Host-side view of the reduced Robotiq 2F-85 of the Button-Push kernel.

The reference's ``Robotiq2f85`` (``mujoco_sim/entities/eef/gripper.py:36-98``) wraps the menagerie MJCF; what it computes itself
is restated here on batches: ``tcp_offset``, ``open_distance``, the opening <-> driver-angle maps and ``move``'s ctrl value.
The driver angle and velocity of every env are rows 16-17 of the Button-Push state (``mjs_get_state``); the physics (actuator,
finger-tip contacts) runs in ``csrc/mjs_button.h``, never here.
"""
from __future__ import annotations

import torch

OPEN_DISTANCE = 0.085           # gripper.py:50-52
MAX_DRIVER_JOINT_ANGLE = 0.8    # gripper.py:38
TCP_OFFSET = (0.0, 0.0, 0.174)  # gripper.py:46-48
CTRL_MAX = 255.0                # gripper.py:82-84
STATE_ROW_ANGLE, STATE_ROW_VELOCITY = 16, 17


def joint_angle_to_finger_distance(joint_angle: torch.Tensor) -> torch.Tensor:
    """gripper.py:73-75"""
    sin_max = torch.sin(torch.tensor(MAX_DRIVER_JOINT_ANGLE, dtype=joint_angle.dtype, device=joint_angle.device))
    return OPEN_DISTANCE * (1 - torch.sin(joint_angle) / sin_max)


def finger_distance_to_joint_angle(finger_distance: torch.Tensor) -> torch.Tensor:
    """gripper.py:77-78"""
    sin_max = torch.sin(torch.tensor(MAX_DRIVER_JOINT_ANGLE, dtype=finger_distance.dtype, device=finger_distance.device))
    return torch.arcsin((1 - finger_distance / OPEN_DISTANCE) * sin_max)


def move_ctrl(finger_distance: torch.Tensor) -> torch.Tensor:
    """The fingers_actuator ctrl ``Robotiq2f85.move`` writes (gripper.py:80-84); the kernel applies the same map to the last
    action component (argument of the arcsin and ctrl clamped to their ranges, as MuJoCo clamps ctrl to ctrlrange)."""
    arg = ((1 - finger_distance / OPEN_DISTANCE) * torch.sin(torch.tensor(MAX_DRIVER_JOINT_ANGLE, dtype=finger_distance.dtype))).clamp(-1, 1)
    return (torch.arcsin(arg) / MAX_DRIVER_JOINT_ANGLE * CTRL_MAX).clamp(0, CTRL_MAX)


class Robotiq2f85Batch:
    """``get_finger_opening`` / driver state of every env of a Button-Push ``HipVectorEnv`` (gripper.py:64-78)."""

    open_distance = OPEN_DISTANCE
    max_driver_joint_angle = MAX_DRIVER_JOINT_ANGLE
    tcp_offset = TCP_OFFSET

    def __init__(self, venv):
        if venv.spec.name != "robot_push_button":
            raise ValueError("only the Button-Push scene carries the reduced 2F-85")
        self._venv = venv

    def driver_state(self) -> torch.Tensor:
        """[N, 2]: driver joint angle (rad) and velocity (rad/s)"""
        return self._venv.get_state()[STATE_ROW_ANGLE:STATE_ROW_VELOCITY + 1].T.contiguous()

    def get_finger_opening(self) -> torch.Tensor:
        return joint_angle_to_finger_distance(self.driver_state()[:, 0])
