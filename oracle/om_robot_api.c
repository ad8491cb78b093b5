/* om_robot_api.c — CPU float64 ORACLE (test infrastructure only; see mjs_oracle.h) for the Robot entity's control API:
 * moveJ / movej_IK / servoL / servoJ / before_substep / get_tcp_pose of
 * /root/reference/mujoco_sim/entities/robots/robot.py:113-272 on a stand-alone UR5e, which is what the reference's own
 * component tests drive (/root/reference/test/test_ur_control_api.py:7-82: Robot entity + raw mjcf.Physics, no task).
 * PARITY UNPINNED for the physics underneath (mjs_oracle.h); what IS held here are those tests' tolerances. */
#include <math.h>
#include <string.h>

#include "../include/mjs_scene_spec.h"
#include "mjs_oracle.h"

/* SE3Container.orientation_as_quaternion (SE3Container.py:102-106): rotation -> (angle, unit axis) by spatialmath's
 * tr2angvec / trlog [third-party, recalled: theta = acos((tr-1)/2) in [0, pi], axis = vex(R - R^T) / (2 sin theta); the
 * theta = pi case takes the column of R + I with the largest diagonal entry] -> UnitQuaternion.AngVec = [cos(theta/2),
 * sin(theta/2) axis] -> scalar-LAST. The scalar part is therefore >= 0; at theta ~ pi the sign of the axis follows
 * sub-tolerance residuals of R - R^T (the double cover q ~ -q), which is why the tests compare such poses up to sign. */
void om_rotation_to_quat_xyzw(const double* R, double* q) {
  double tr = R[0] + R[4] + R[8];
  double c = 0.5 * (tr - 1.0);
  if (c > 1.0) c = 1.0;
  if (c < -1.0) c = -1.0;
  double theta = acos(c);
  double ax[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  double n = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
  if (theta < 1e-12 || (n < 1e-12 && c > 0)) { q[0] = q[1] = q[2] = 0; q[3] = 1; return; }
  if (n < 1e-9) { /* theta = pi: R = 2 a a^T - I */
    int k = 0;
    if (R[4] > R[0]) k = 1;
    if (R[8] > R[4 * k]) k = 2;
    double col[3] = {R[k] + (k == 0), R[3 + k] + (k == 1), R[6 + k] + (k == 2)};
    double m = sqrt(2.0 * (1.0 + R[4 * k]));
    for (int i = 0; i < 3; i++) ax[i] = col[i] / m;
  } else {
    for (int i = 0; i < 3; i++) ax[i] /= n;
  }
  double s = sin(0.5 * theta);
  q[0] = s * ax[0]; q[1] = s * ax[1]; q[2] = s * ax[2]; q[3] = cos(0.5 * theta);
}

/* robot.py:113-121,138-151: TCP pose -> flange pose (tcp_in_flange = translation tcp_z along the flange z) -> IK closest */
static int tcp_pose_to_joints7(const double* pose, double tcp_z, const double* q_guess, double* q_out) {
  double x = pose[3], y = pose[4], z = pose[5], w = pose[6];
  double n = sqrt(x * x + y * y + z * z + w * w);
  x /= n; y /= n; z /= n; w /= n;
  double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
  double T[16] = {R[0], R[1], R[2], pose[0] - R[2] * tcp_z, R[3], R[4], R[5], pose[1] - R[5] * tcp_z,
                  R[6], R[7], R[8], pose[2] - R[8] * tcp_z, 0, 0, 0, 1};
  return om_ur5e_ik_closest(T, q_guess, q_out);
}

int om_ur_robot_run(double* st, const double* target, int command, double param, int n_substeps, int eef, double dt, double* tcp_pose_out) {
  static __thread om_model m;
  static __thread om_data d;
  static __thread int built_eef = -1;
  static __thread double built_dt = 0;
  if (built_eef != eef || built_dt != dt) { om_build_ur5e_alone(&m, eef == OM_UR_EEF_GRIPPER, dt); built_eef = eef; built_dt = dt; }
  const double tcp_z = eef == OM_UR_EEF_GRIPPER ? MJS_G2F85_TCP_Z : 0.0; /* robot.py:104-107: the bare flange has no TCP offset */
  double *q = st, *v = st + 6, *ctrl = st + 12, *time = st + 18, *active = st + 19, *q0 = st + 20, *q1 = st + 26, *t0 = st + 32, *t1 = st + 33;
  om_reset_data(&m, &d);
  memcpy(d.qpos, q, sizeof(double) * 6);
  memcpy(d.qvel, v, sizeof(double) * 6);
  memcpy(d.ctrl, ctrl, sizeof(double) * 6);
  d.time = *time;
  om_forward(&m, &d);
  int ok = 1;
  if (command != OM_UR_CMD_NONE) {
    double tgt[6];
    int have = 1;
    if (command == OM_UR_CMD_MOVEJ_IK || command == OM_UR_CMD_SERVOL) have = tcp_pose_to_joints7(target, tcp_z, d.qpos, tgt); /* robot.py:204-205,219-220 */
    else memcpy(tgt, target, sizeof tgt);
    if (!have) ok = 0; /* movej_IK: print + return; servoL: raise (the trajectory is left as it was) */
    else {
      double span = param; /* servoJ / servoL: robot.py:254-258 */
      if (command == OM_UR_CMD_MOVEJ || command == OM_UR_CMD_MOVEJ_IK) { /* robot.py:211-216: time = max |dq| / speed */
        double mx = 0;
        for (int j = 0; j < 6; j++) mx = fmax(mx, fabs(tgt[j] - d.qpos[j]));
        span = mx / param;
      }
      for (int j = 0; j < 6; j++) { q0[j] = d.qpos[j]; q1[j] = tgt[j]; }
      *t0 = d.time; *t1 = d.time + span; *active = 1;
    }
  }
  for (int s = 0; s < n_substeps; s++) {
    if (*active != 0) { /* Robot.before_substep, robot.py:261-263; joint_trajectory.py:33-47 */
      double t = fmin(fmax(d.time, *t0), *t1);
      for (int j = 0; j < 6; j++) d.ctrl[j] = q0[j] + (q1[j] - q0[j]) * (t - *t0) / (*t1 - *t0);
      if (dt >= *t1) *active = 0; /* robot.py:271: is_finished is handed physics.timestep(), i.e. dt (joint_trajectory.py:57-59) */
    }
    om_physics_step(&m, &d);
  }
  memcpy(q, d.qpos, sizeof(double) * 6);
  memcpy(v, d.qvel, sizeof(double) * 6);
  memcpy(ctrl, d.ctrl, sizeof(double) * 6);
  *time = d.time;
  if (tcp_pose_out) { /* robot.py:153-168 */
    const double* p = d.site_xpos[0];
    const double* R = d.site_xmat[0];
    for (int k = 0; k < 3; k++) tcp_pose_out[k] = p[k] + R[3 * k + 2] * tcp_z;
    om_rotation_to_quat_xyzw(R, tcp_pose_out + 3);
  }
  return ok;
}
