"""CPU float64 ORACLE (test infrastructure only) — see oracle/mjs_oracle.h.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; the product (``mujoco_sim_amd``) never does.
PARITY UNPINNED for the physics (MuJoCo / dm_control / ur_analytic_ik are absent
third-party dependencies of the reference); pinned pieces are listed in DESIGN.md.
"""
from .oracle import (  # noqa: F401
    AUTORESET_DISABLED,
    AUTORESET_NEXT_STEP,
    AUTORESET_SAME_STEP,
    TASK_POINTMASS,
    TASK_ROBOT_REACH,
    TASK_BUTTON_PUSH,
    TASK_PLANAR_PUSH,
    ACTION_ABS_JOINT,
    ACTION_ABS_EEF,
    OracleBatch,
    OracleRng,
    build,
    lib,
    ur5e_fk_dh,
    ur5e_ik_all,
    ur5e_ik_closest,
    UR_STATE, UR_CMD_NONE, UR_CMD_MOVEJ, UR_CMD_MOVEJ_IK, UR_CMD_SERVOL, UR_CMD_SERVOJ, UR_EEF_NONE, UR_EEF_GRIPPER,
    BLOCKS_MESH, BLOCKS_BOX,
    ur_robot_state,
    ur_robot_run,
)
