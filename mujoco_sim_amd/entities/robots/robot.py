"""Batched mirror of the reference's ``Robot`` / ``UR5e`` control API on the HIP path.

``mujoco_sim/entities/robots/robot.py:41-321``: a position-controlled UR5e mimicking the UR control box —
``moveJ`` / ``movej_IK`` / ``servoL`` / ``servoJ`` plan a 2-waypoint joint trajectory, ``before_substep`` writes its
interpolated set-point into the position servos' ``ctrl`` and ``Physics.step`` integrates; ``get_tcp_pose`` /
``get_joint_positions`` read the state back. The reference's own tests drive exactly this component
(``test/test_ur_control_api.py:7-82``: ``UR5e()`` + raw ``mjcf.Physics``, the XML's default timestep, no task).

Here ``n`` robots live on one GPU, one wavefront lane each; the arithmetic is ``mjs_ur5e_robot_run`` of libmjsim.so
(include/mjsim.h), the same device code the task kernels run. ``substeps(k)`` = k x (before_substep; physics.step).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ... import _native as nat

HOME_JOINT_POSITIONS = np.array([-0.5, -0.5, 0.5, -0.5, -0.5, -0.5]) * np.pi  # robot.py:307


class UR5eBatch:
    """n stand-alone UR5e arms (``eef=None``: bare flange, TCP = flange; ``eef="gripper"``: lumped 2F-85, TCP 0.174 m
    down the flange z axis). Poses are ``xyz + scalar-last quaternion`` (type_aliases.py:6-10)."""

    home_joint_positions = HOME_JOINT_POSITIONS
    max_joint_speed = 1.0  # rad/s (robot.py:308)

    def __init__(self, n: int = 1, device: str | torch.device = "cuda:0", eef: str | None = None, physics_timestep: float = 0.002):
        if not torch.cuda.is_available():
            raise nat.MjsError("UR5eBatch needs a HIP device; there is no CPU path")
        if eef not in (None, "gripper"):
            raise ValueError("eef must be None or 'gripper'")
        self.n, self.device, self.dt = int(n), torch.device(device), float(physics_timestep)
        self._eef = nat.UR_EEF_GRIPPER if eef == "gripper" else nat.UR_EEF_NONE
        self._lib = nat.lib()
        self._state = torch.zeros(self.n, nat.UR_STATE, dtype=torch.float64, device=self.device)  # qpos0 = 0, ctrl = 0
        self._pose = torch.zeros(self.n, 7, dtype=torch.float64, device=self.device)
        self._status = torch.ones(self.n, dtype=torch.uint8, device=self.device)
        self._pending = None  # (command, target[n, 7], param): applied at the start of the next substeps() call

    # ------------------------------------------------------------------ state access
    def set_joint_positions(self, joint_positions):
        """robot.py:185-189: qpos = q, qvel = 0, ctrl = q, trajectory cleared"""
        q = self._as(joint_positions, 6)
        self._state.zero_()
        self._state[:, 0:6] = q
        self._state[:, 12:18] = q
        self._pending = None

    def get_joint_positions(self) -> torch.Tensor:
        return self._state[:, 0:6].clone()

    def time(self) -> torch.Tensor:
        return self._state[:, 18].clone()

    def get_tcp_pose(self) -> torch.Tensor:
        self._run(nat.UR_CMD_NONE, None, 1.0, 0)
        return self._pose.clone()

    # ------------------------------------------------------------------ IK helpers (robot.py:113-124,176-183)
    @property
    def tcp_offset_z(self) -> float:
        return 0.174 if self._eef == nat.UR_EEF_GRIPPER else 0.0  # gripper.py:46-48 / bare flange

    def get_joint_positions_from_tcp_pose(self, tcp_pose, current_joints=None):
        """robot.py:113-121: TCP pose (xyz + scalar-last quaternion) -> flange pose -> inverse_kinematics_closest to ``current_joints``
        (default: the home joints). Returns (joints [n, 6], found [n] bool); the reference returns None where no solution exists."""
        pose = self._as(tcp_pose, 7)
        guess = self._as(self.home_joint_positions if current_joints is None else current_joints, 6)
        x, y, z, w = (pose[:, 3 + k] for k in range(4))
        nrm = torch.sqrt(x * x + y * y + z * z + w * w)
        x, y, z, w = x / nrm, y / nrm, z / nrm, w / nrm
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                         2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                         2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=1)
        t = pose[:, 0:3] - R.view(-1, 3, 3)[:, :, 2] * self.tcp_offset_z  # flange = TCP * inv(T_tcp_in_flange), robot.py:138-151
        T = torch.cat([R, t], dim=1).contiguous()
        q = torch.empty(self.n, 6, dtype=torch.float64, device=self.device)
        ok = torch.empty(self.n, dtype=torch.uint8, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            nat.check(self._lib.mjs_debug_ur5e_ik(C.c_void_p(T.data_ptr()), C.c_void_p(guess.data_ptr()), C.c_void_p(q.data_ptr()), C.c_void_p(ok.data_ptr()),
                                                  self.n, C.c_void_p(stream)))
        return q, ok.bool()

    def is_pose_reachable(self, tcp_pose) -> torch.Tensor:
        """robot.py:123-124"""
        return self.get_joint_positions_from_tcp_pose(tcp_pose)[1]

    def set_tcp_pose(self, pose):
        """robot.py:176-183: IK from the current joints; robots whose pose is unreachable keep their state (the reference passes silently)"""
        self._flush()
        q, ok = self.get_joint_positions_from_tcp_pose(pose, self.get_joint_positions())
        new = self._state.clone()
        new[ok] = 0.0
        new[ok, 0:6] = q[ok]
        new[ok, 12:18] = q[ok]
        self._state.copy_(new)
        return ok

    def is_moving(self) -> torch.Tensor:
        """robot.py:274-275: a joint trajectory is set. (It stays set once a command was given: before_substep hands
        ``physics.timestep()`` to ``is_finished``, robot.py:271, so the reference never clears it either.)"""
        self._flush()
        return self._state[:, 19] != 0

    def moveL(self, tcp_pose, speed: float):
        raise NotImplementedError("moveL not implemented")  # robot.py:193-194

    # ------------------------------------------------------------------ control API (robot.py:198-259)
    def moveJ(self, target_joint_positions, speed: float):
        self._flush()
        self._pending = (nat.UR_CMD_MOVEJ, self._pad(self._as(target_joint_positions, 6)), float(speed))

    def movej_IK(self, tcp_pose, speed: float):
        self._flush()
        self._pending = (nat.UR_CMD_MOVEJ_IK, self._as(tcp_pose, 7), float(speed))

    def servoL(self, tcp_pose, time: float):
        self._flush()
        self._pending = (nat.UR_CMD_SERVOL, self._as(tcp_pose, 7), float(time))

    def servoJ(self, target_joint_positions, time: float):
        self._flush()
        self._pending = (nat.UR_CMD_SERVOJ, self._pad(self._as(target_joint_positions, 6)), float(time))

    def substeps(self, k: int):
        """k x (Robot.before_substep; physics.step()), robot.py:261-272"""
        cmd, tgt, param = self._pending or (nat.UR_CMD_NONE, None, 1.0)
        self._pending = None
        self._run(cmd, tgt, param, int(k))

    @property
    def ik_ok(self) -> torch.Tensor:
        """per robot: the last IK-based command found a solution (movej_IK prints "IK failed", servoL raises ValueError)"""
        return (self._status & 1).bool()

    @property
    def status(self) -> torch.Tensor:
        return self._status.clone()

    # ------------------------------------------------------------------ plumbing
    def _flush(self):
        if self._pending is not None:  # a command replaced before any substep still planned from the same state
            self.substeps(0)

    def _as(self, x, width):
        t = torch.as_tensor(np.asarray(x, dtype=np.float64) if not torch.is_tensor(x) else x, dtype=torch.float64, device=self.device)
        if t.dim() == 1:
            t = t.expand(self.n, width)
        assert t.shape == (self.n, width), t.shape
        return t.contiguous()

    def _pad(self, q):
        return torch.cat([q, torch.zeros(self.n, 1, dtype=torch.float64, device=self.device)], dim=1).contiguous()

    def _run(self, command, target, param, k):
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            nat.check(self._lib.mjs_ur5e_robot_run(C.c_void_p(self._state.data_ptr()), C.c_void_p(target.data_ptr()) if target is not None else None,
                                                   int(command), float(param), int(k), self._eef, self.dt, C.c_void_p(self._pose.data_ptr()),
                                                   C.c_void_p(self._status.data_ptr()), self.n, C.c_void_p(stream)))
