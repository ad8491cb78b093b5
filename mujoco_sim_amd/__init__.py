"""mujoco_sim_amd — MI355X-native batched replacement for the per-step hot path of
tlpss/mujoco-sim (physics step + task obs/reward/termination + the dmc2gym step/reset wrapper).

Two front ends over the same C ABI (include/mjsim.h -> libmjsim.so, hand-written HIP):
  * :class:`HipVectorEnv` — N envs on one GPU, torch tensors in/out (the fast path);
  * :class:`DMCEnvironmentAdapter` + registry ids — the reference's own single-env gymnasium
    surface (mujoco_sim/__init__.py:19-40, environments/dmc2gym.py:78-168).
"""
from __future__ import annotations

from functools import partial

from . import _native
from ._native import MjsError, build  # noqa: F401
from .environments.dmc2gym import DMCEnvironmentAdapter, HipEnvironment, TimeStep  # noqa: F401
from .environments.tasks.point_reach import PointMassReachTask
from .environments.tasks.robot_reach import RobotReachConfig, RobotReachTask
from .environments.tasks.robot_push_button import RobotPushButtonTask
from .environments.tasks.robot_planar_push import RobotPushConfig, RobotPushTask
from .recording import LeRobotDatasetRecorder  # noqa: F401
from .vector_env import TASKS, HipVectorEnv  # noqa: F401

__version__ = "0.1.0"


def make_point_mass_reach_env(task_class, max_steps, device="cuda:0", **kwargs):
    """mujoco_sim/__init__.py:19-23: task -> Environment(time_limit = max_steps * CONTROL_TIMESTEP) -> adapter."""
    task = task_class(**kwargs)
    env = HipEnvironment(task, time_limit=max_steps * task.CONTROL_TIMESTEP, device=device)
    return DMCEnvironmentAdapter(env, flatten_observation_space=False)


def _make_robot_planar_push_env(device="cuda:0", **kwargs):
    """scripts/sb3/planar_push.py:66-74: task -> Environment (no time limit: the task counts its steps) -> adapter"""
    task = RobotPushTask(RobotPushConfig(**kwargs))
    env = HipEnvironment(task, device=device)
    return DMCEnvironmentAdapter(env, flatten_observation_space=False)


def _make_robot_reach_env(max_steps=100, device="cuda:0", **kwargs):
    task = RobotReachTask(RobotReachConfig(**kwargs))
    env = HipEnvironment(task, time_limit=max_steps * task.CONTROL_TIMESTEP, device=device)
    return DMCEnvironmentAdapter(env, flatten_observation_space=False)


# registry ids of the reference (mujoco_sim/__init__.py:26-40): point_mass_reach-v0 is the VISUAL
# variant (64x64 top-down camera image + position); robot_push_button_visual-v0 is joints + wrist and scene camera images.
# The two *_state ids are additions for the state-observation configs of BASELINE.json.
registry = {
    "mujoco_sim/point_mass_reach-v0": (partial(make_point_mass_reach_env, PointMassReachTask, max_steps=50),
                                       {"observation_type": "visual_observations", "image_resolution": 64}),
    "mujoco_sim/point_mass_reach_state-v0": (partial(make_point_mass_reach_env, PointMassReachTask, max_steps=50),
                                             {"observation_type": "state_observations"}),
    "mujoco_sim/robot_reach_state-v0": (_make_robot_reach_env, {}),
    # mujoco_sim/__init__.py:31-39
    "mujoco_sim/robot_push_button_visual-v0": (partial(make_point_mass_reach_env, RobotPushButtonTask, max_steps=100),
                                               {"observation_type": RobotPushButtonTask.VISUAL_OBS, "image_resolution": 96,
                                                "action_type": RobotPushButtonTask.ABS_JOINT_ACTION}),
    "mujoco_sim/robot_planar_push_state-v0": (_make_robot_planar_push_env, {"n_objects": 2}),  # robot_planar_push.py:315 / BASELINE config 4
    "mujoco_sim/robot_push_button_state-v0": (partial(make_point_mass_reach_env, RobotPushButtonTask, max_steps=100),
                                              {"observation_type": RobotPushButtonTask.STATE_OBS, "action_type": RobotPushButtonTask.ABS_JOINT_ACTION}),
}


def make(env_id: str, **kwargs):
    """gymnasium.make equivalent for the ids above (kwargs override the registered ones)."""
    if env_id not in registry:
        raise KeyError(f"unknown env id {env_id!r}; registered: {sorted(registry)}")
    entry, default_kwargs = registry[env_id]
    return entry(**{**default_kwargs, **kwargs})


def make_vec(task: str, num_envs: int, **kwargs) -> HipVectorEnv:
    return HipVectorEnv(task, num_envs, **kwargs)


try:  # pragma: no cover - only when gymnasium is installed
    import gymnasium as _gym

    for _id, (_entry, _kw) in registry.items():
        if _id not in _gym.registry:
            _gym.register(id=_id, entry_point=_entry, kwargs=_kw)
except Exception:  # noqa: BLE001
    pass
