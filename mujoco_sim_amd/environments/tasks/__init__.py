from .point_reach import PointMassReachTask  # noqa: F401
from .robot_reach import RobotReachConfig, RobotReachTask  # noqa: F401
