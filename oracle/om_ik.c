/* om_ik.c — ORACLE (test infrastructure): closed-form UR5e kinematics.
 *
 * Restates the third-party `ur_analytic_ik` package (ur-analytic-ik on PyPI,
 * unpinned in the reference's setup.py:16; absent from /root/reference). Call
 * sites: entities/robots/robot.py:33-37 (`ur5e.inverse_kinematics_closest(pose,
 * *q_guess)`, first element used at :121) and
 * test/test_ur_frame_matches_real.py:25-27 (`ur5e.forward_kinematics`).
 * Published algorithm: K. P. Hawkins, "Analytic Inverse Kinematics for the
 * Universal Robots UR-5/UR-10 Arms" (2013), standard DH table of the UR5e.
 *
 * Candidate validity is analytic (acos domains, reach); the FK round trip of every
 * returned solution is asserted in tests/test_oracle_known_answers.py.
 * "closest": joint angles are first mapped to the 2*pi-equivalent nearest to the
 * guess (kept inside [-2pi, 2pi]), then the minimum Euclidean joint distance wins
 * (recalled behaviour; DESIGN.md deviation D-7). PARITY UNPINNED.
 */
#include <math.h>
#include <string.h>

#include "../include/mjs_scene_spec.h"
#include "mjs_oracle.h"

static void dh(double T[16], double th, double d, double a, double al) {
  double ct = cos(th), st = sin(th), ca = cos(al), sa = sin(al);
  double M[16] = {ct, -st * ca, st * sa, a * ct, st, ct * ca, -ct * sa, a * st, 0, sa, ca, d, 0, 0, 0, 1};
  memcpy(T, M, sizeof M);
}
static void mul44(double* R, const double* A, const double* B) {
  double M[16];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double s = 0;
      for (int k = 0; k < 4; k++) s += A[4 * i + k] * B[4 * k + j];
      M[4 * i + j] = s;
    }
  memcpy(R, M, sizeof M);
}
static void inv44(double* R, const double* A) {
  /* rigid transform inverse */
  double M[16] = {A[0], A[4], A[8], 0, A[1], A[5], A[9], 0, A[2], A[6], A[10], 0, 0, 0, 0, 1};
  for (int i = 0; i < 3; i++) M[4 * i + 3] = -(M[4 * i] * A[3] + M[4 * i + 1] * A[7] + M[4 * i + 2] * A[11]);
  memcpy(R, M, sizeof M);
}

static const double HALF_PI = 1.5707963267948966;

void om_ur5e_fk_dh(const double q[6], double T[16]) {
  const double d[6] = {MJS_UR_DH_D1, 0, 0, MJS_UR_DH_D4, MJS_UR_DH_D5, MJS_UR_DH_D6};
  const double a[6] = {0, MJS_UR_DH_A2, MJS_UR_DH_A3, 0, 0, 0};
  const double al[6] = {HALF_PI, 0, 0, HALF_PI, -HALF_PI, 0};
  double A[16];
  dh(T, q[0], d[0], a[0], al[0]);
  for (int i = 1; i < 6; i++) {
    dh(A, q[i], d[i], a[i], al[i]);
    mul44(T, T, A);
  }
}

static double wrap_pi(double x) {
  while (x > M_PI) x -= 2 * M_PI;
  while (x <= -M_PI) x += 2 * M_PI;
  return x;
}
static double clamp1(double x, int* ok) {
  if (x > 1.0) { if (x > 1.0 + 1e-9) *ok = 0; return 1.0; }
  if (x < -1.0) { if (x < -1.0 - 1e-9) *ok = 0; return -1.0; }
  return x;
}

int om_ur5e_ik_all(const double T[16], double sols[8][6]) {
  const double d1 = MJS_UR_DH_D1, a2 = MJS_UR_DH_A2, a3 = MJS_UR_DH_A3, d4 = MJS_UR_DH_D4, d5 = MJS_UR_DH_D5, d6 = MJS_UR_DH_D6;
  int n = 0;
  double p05x = T[3] - d6 * T[2], p05y = T[7] - d6 * T[6];
  double rxy = sqrt(p05x * p05x + p05y * p05y);
  if (rxy < fabs(d4)) return 0;
  double psi = atan2(p05y, p05x), phi = acos(d4 / rxy);
  double Tinv[16];
  inv44(Tinv, T);
  for (int s1 = 0; s1 < 2; s1++) {
    double th1 = psi + (s1 ? -phi : phi) + HALF_PI;
    double c1 = cos(th1), sn1 = sin(th1);
    int ok5 = 1;
    double c5 = clamp1((T[3] * sn1 - T[7] * c1 - d4) / d6, &ok5);
    if (!ok5) continue;
    for (int s5 = 0; s5 < 2; s5++) {
      double th5 = (s5 ? -1 : 1) * acos(c5);
      double sn5 = sin(th5), th6;
      if (fabs(sn5) < 1e-12) th6 = 0; /* wrist singularity: theta6 free, choose 0 */
      else {
        double X60x = Tinv[0], X60y = Tinv[4], Y60x = Tinv[1], Y60y = Tinv[5];
        th6 = atan2((-X60y * sn1 + Y60y * c1) / sn5, (X60x * sn1 - Y60x * c1) / sn5);
      }
      /* T14 = inv(T01) * T06 * inv(T45*T56) */
      double T01[16], T45[16], T56[16], T46[16], T14[16], tmp[16];
      dh(T01, th1, d1, 0, HALF_PI);
      dh(T45, th5, d5, 0, -HALF_PI);
      dh(T56, th6, d6, 0, 0);
      mul44(T46, T45, T56);
      inv44(tmp, T01);
      mul44(T14, tmp, T);
      inv44(tmp, T46);
      mul44(T14, T14, tmp);
      /* frame 1 -> 4 is planar in x,y of frame 1 (z of frames 1..3 parallel) */
      double px = T14[3], py = T14[7];
      double r2 = px * px + py * py;
      int ok3 = 1;
      double c3 = clamp1((r2 - a2 * a2 - a3 * a3) / (2 * a2 * a3), &ok3);
      if (!ok3) continue;
      for (int s3 = 0; s3 < 2; s3++) {
        double th3 = (s3 ? -1 : 1) * acos(c3);
        double th2 = atan2(py, px) - atan2(a3 * sin(th3), a2 + a3 * cos(th3));
        double T12[16], T23[16], T13[16], T34[16];
        dh(T12, th2, 0, a2, 0);
        dh(T23, th3, 0, a3, 0);
        mul44(T13, T12, T23);
        inv44(tmp, T13);
        mul44(T34, tmp, T14);
        double th4 = atan2(T34[4], T34[0]);
        double q[6] = {wrap_pi(th1), wrap_pi(th2), wrap_pi(th3), wrap_pi(th4), wrap_pi(th5), wrap_pi(th6)};
        int finite = 1;
        for (int k = 0; k < 6; k++) finite &= isfinite(q[k]) ? 1 : 0;
        if (finite) { memcpy(sols[n], q, sizeof q); n++; }
      }
    }
  }
  return n;
}

int om_ur5e_ik_closest(const double T[16], const double q_guess[6], double q_out[6]) {
  double sols[8][6];
  int n = om_ur5e_ik_all(T, sols);
  if (n == 0) return 0;
  double best = INFINITY;
  for (int s = 0; s < n; s++) {
    double q[6], dist = 0;
    for (int j = 0; j < 6; j++) {
      q[j] = sols[s][j];
      double alt = q[j] + (q_guess[j] > q[j] ? 2 * M_PI : -2 * M_PI);
      if (fabs(alt - q_guess[j]) < fabs(q[j] - q_guess[j]) && fabs(alt) <= 2 * M_PI) q[j] = alt;
      dist += (q[j] - q_guess[j]) * (q[j] - q_guess[j]);
    }
    if (dist < best) { best = dist; memcpy(q_out, q, sizeof q); }
  }
  return 1;
}
