/* om_engine.c — ORACLE (test infrastructure): scalar float64 restatement of the
 * MuJoCo pipeline stages the four scenes exercise, i.e. what
 * `Physics.step()` does underneath the reference's step path
 * (environments/dmc2gym.py:136 -> composer.Environment.step -> Physics.step;
 * direct call sites robot_planar_push.py:160-161, test/test_ur_control_api.py:21).
 *
 * MuJoCo itself is a third-party dependency absent from /root/reference
 * (unpinned in setup.py:11-20); this file restates its published algorithms
 * (MuJoCo documentation, "Computation" chapter; Featherstone 2008 for CRBA/RNEA):
 *   kinematics -> composite-rigid-body inertia -> collision -> constraint rows
 *   (equality, joint limit, pyramidal contact) with impedance/reference
 *   acceleration -> bias forces (RNE), gravity compensation, affine actuators ->
 *   unconstrained acceleration -> primal Newton solver with exact line search ->
 *   semi-implicit Euler / implicitfast integration.
 * Spatial quantities are expressed in world axes about the world origin (MuJoCo
 * uses the kinematic tree's subtree-COM; a rounding-level difference).
 * PARITY UNPINNED — see mjs_oracle.h.
 */
#include <math.h>
#include <string.h>

#include "../include/mjs_scene_spec.h"
#include "mjs_oracle.h"
#include "../include/mjs_block_hulls.h"

/* ------------------------------------------------------------ small math */
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static double norm3(const double* a) { return sqrt(dot3(a, a)); }
static void mulMatVec3(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  double y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  double z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void mulQuat(double* r, const double* a, const double* b) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void negQuat(double* r, const double* q) { r[0] = q[0]; r[1] = -q[1]; r[2] = -q[2]; r[3] = -q[3]; }
static void normQuat(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MJS_MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void quat2Mat(double* m, const double* q) {
  double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  double q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3], q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2 * (q12 - q03); m[2] = 2 * (q13 + q02); m[3] = 2 * (q12 + q03);
  m[5] = 2 * (q23 - q01); m[6] = 2 * (q13 - q02); m[7] = 2 * (q23 + q01);
}
static void rotVecQuat(double* r, const double* v, const double* q) {
  double m[9];
  quat2Mat(m, q);
  mulMatVec3(r, m, v);
}
static void axisAngle2Quat(double* q, const double* axis, double angle) {
  double s = sin(angle * 0.5);
  q[0] = cos(angle * 0.5); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}

/* ----------------------------------------------------------- kinematics */
/* mj_kinematics: body, joint-anchor, geom and site poses from qpos */
static void om_kinematics(const om_model* m, om_data* d) {
  d->xpos[0][0] = d->xpos[0][1] = d->xpos[0][2] = 0;
  d->xquat[0][0] = 1; d->xquat[0][1] = d->xquat[0][2] = d->xquat[0][3] = 0;
  quat2Mat(d->xmat[0], d->xquat[0]);
  memcpy(d->xipos[0], d->xpos[0], sizeof(double) * 3);
  memcpy(d->ximat[0], d->xmat[0], sizeof(double) * 9);
  for (int b = 1; b < m->nbody; b++) {
    double xpos[3], xquat[4], tmp[3];
    int p = m->body_parent[b];
    if (m->body_mocapid[b] >= 0) {
      memcpy(xpos, d->mocap_pos[m->body_mocapid[b]], sizeof xpos);
      memcpy(xquat, d->mocap_quat[m->body_mocapid[b]], sizeof xquat);
      normQuat(xquat);
    } else {
      int jadr = m->body_jntadr[b], jnum = m->body_jntnum[b];
      if (jnum == 1 && m->jnt_type[jadr] == OM_JNT_FREE) {
        int qa = m->jnt_qposadr[jadr];
        memcpy(xpos, d->qpos + qa, sizeof xpos);
        memcpy(xquat, d->qpos + qa + 3, sizeof xquat);
        normQuat(xquat);
        memcpy(d->xanchor[jadr], xpos, sizeof xpos);
        d->xaxis[jadr][0] = 0; d->xaxis[jadr][1] = 0; d->xaxis[jadr][2] = 1;
      } else {
        mulMatVec3(tmp, d->xmat[p], m->body_pos[b]);
        for (int k = 0; k < 3; k++) xpos[k] = d->xpos[p][k] + tmp[k];
        mulQuat(xquat, d->xquat[p], m->body_quat[b]);
        for (int j = jadr; j < jadr + jnum; j++) {
          int qa = m->jnt_qposadr[j];
          rotVecQuat(d->xaxis[j], m->jnt_axis[j], xquat);
          rotVecQuat(tmp, m->jnt_pos[j], xquat);
          for (int k = 0; k < 3; k++) d->xanchor[j][k] = xpos[k] + tmp[k];
          if (m->jnt_type[j] == OM_JNT_SLIDE) {
            double dq = d->qpos[qa] - m->qpos0[qa];
            for (int k = 0; k < 3; k++) xpos[k] += d->xaxis[j][k] * dq;
          } else if (m->jnt_type[j] == OM_JNT_HINGE) {
            double qloc[4], q2[4];
            axisAngle2Quat(qloc, m->jnt_axis[j], d->qpos[qa] - m->qpos0[qa]);
            mulQuat(q2, xquat, qloc);
            memcpy(xquat, q2, sizeof q2);
            rotVecQuat(tmp, m->jnt_pos[j], xquat);
            for (int k = 0; k < 3; k++) xpos[k] = d->xanchor[j][k] - tmp[k];
          }
        }
      }
    }
    normQuat(xquat);
    memcpy(d->xpos[b], xpos, sizeof xpos);
    memcpy(d->xquat[b], xquat, sizeof xquat);
    quat2Mat(d->xmat[b], xquat);
    mulMatVec3(tmp, d->xmat[b], m->body_ipos[b]);
    for (int k = 0; k < 3; k++) d->xipos[b][k] = xpos[k] + tmp[k];
    double iq[4];
    mulQuat(iq, xquat, m->body_iquat[b]);
    quat2Mat(d->ximat[b], iq);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_body[g];
    double tmp[3], q[4];
    mulMatVec3(tmp, d->xmat[b], m->geom_pos[g]);
    for (int k = 0; k < 3; k++) d->geom_xpos[g][k] = d->xpos[b][k] + tmp[k];
    mulQuat(q, d->xquat[b], m->geom_quat[g]);
    quat2Mat(d->geom_xmat[g], q);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_body[s];
    double tmp[3], q[4];
    mulMatVec3(tmp, d->xmat[b], m->site_pos[s]);
    for (int k = 0; k < 3; k++) d->site_xpos[s][k] = d->xpos[b][k] + tmp[k];
    mulQuat(q, d->xquat[b], m->site_quat[s]);
    quat2Mat(d->site_xmat[s], q);
  }
}

/* motion subspace of every dof, world axes, about the world origin (role of cdof) */
static void om_subspaces(const om_model* m, om_data* d) {
  for (int j = 0; j < m->njnt; j++) {
    int da = m->jnt_dofadr[j], b = m->jnt_body[j];
    if (m->jnt_type[j] == OM_JNT_HINGE) {
      memcpy(d->S[da], d->xaxis[j], sizeof(double) * 3);
      cross3(d->S[da] + 3, d->xanchor[j], d->xaxis[j]);
    } else if (m->jnt_type[j] == OM_JNT_SLIDE) {
      d->S[da][0] = d->S[da][1] = d->S[da][2] = 0;
      memcpy(d->S[da] + 3, d->xaxis[j], sizeof(double) * 3);
    } else if (m->jnt_type[j] == OM_JNT_FREE) {
      /* translational dofs: world axes; rotational dofs: body-local axes through xpos */
      for (int k = 0; k < 3; k++) {
        memset(d->S[da + k], 0, sizeof(double) * 6);
        d->S[da + k][3 + k] = 1;
        double ax[3] = {d->xmat[b][k], d->xmat[b][3 + k], d->xmat[b][6 + k]};
        memcpy(d->S[da + 3 + k], ax, sizeof ax);
        cross3(d->S[da + 3 + k] + 3, d->xpos[b], ax);
      }
    }
  }
}

/* spatial inertia about the world origin: I (sym 3x3 as xx,xy,xz,yy,yz,zz), h = m*c, mass */
typedef struct { double I[6], h[3], mass; } sinertia;
static void om_body_inertia(const om_model* m, const om_data* d, int b, sinertia* s) {
  const double* R = d->ximat[b];
  const double* di = m->body_inertia[b];
  const double* c = d->xipos[b];
  double mass = m->body_mass[b];
  /* R diag(di) R^T */
  double Ic[6];
  int idx = 0;
  for (int r = 0; r < 3; r++)
    for (int cc = r; cc < 3; cc++)
      Ic[idx++] = R[3 * r] * di[0] * R[3 * cc] + R[3 * r + 1] * di[1] * R[3 * cc + 1] + R[3 * r + 2] * di[2] * R[3 * cc + 2];
  double c2 = dot3(c, c);
  s->I[0] = Ic[0] + mass * (c2 - c[0] * c[0]);
  s->I[1] = Ic[1] - mass * c[0] * c[1];
  s->I[2] = Ic[2] - mass * c[0] * c[2];
  s->I[3] = Ic[3] + mass * (c2 - c[1] * c[1]);
  s->I[4] = Ic[4] - mass * c[1] * c[2];
  s->I[5] = Ic[5] + mass * (c2 - c[2] * c[2]);
  for (int k = 0; k < 3; k++) s->h[k] = mass * c[k];
  s->mass = mass;
}
/* spatial momentum/force f = I * v, v = [ang, lin] */
static void sinertia_mul(double* f, const sinertia* s, const double* v) {
  double hv[3], wh[3];
  cross3(hv, s->h, v + 3);
  cross3(wh, v, s->h);
  f[0] = s->I[0] * v[0] + s->I[1] * v[1] + s->I[2] * v[2] + hv[0];
  f[1] = s->I[1] * v[0] + s->I[3] * v[1] + s->I[4] * v[2] + hv[1];
  f[2] = s->I[2] * v[0] + s->I[4] * v[1] + s->I[5] * v[2] + hv[2];
  for (int k = 0; k < 3; k++) f[3 + k] = s->mass * v[3 + k] + wh[k];
}
/* spatial cross products: motion x motion, motion x* force */
static void cross_motion(double* r, const double* v, const double* s) {
  double a[3], b[3], c[3];
  cross3(a, v, s);
  cross3(b, v, s + 3);
  cross3(c, v + 3, s);
  for (int k = 0; k < 3; k++) { r[k] = a[k]; r[3 + k] = b[k] + c[k]; }
}
static void cross_force(double* r, const double* v, const double* f) {
  double a[3], b[3], c[3];
  cross3(a, v, f);
  cross3(b, v + 3, f + 3);
  cross3(c, v, f + 3);
  for (int k = 0; k < 3; k++) { r[k] = a[k] + b[k]; r[3 + k] = c[k]; }
}

/* mj_crb + armature -> dense M; then Cholesky (role of mj_factorM) */
static int chol_factor(int n, double A[][OM_MAXV], double L[][OM_MAXV]) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j <= i; j++) {
      double s = A[i][j];
      for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
      if (i == j) {
        if (s < MJS_MINVAL) return 0;
        L[i][i] = sqrt(s);
      } else
        L[i][j] = s / L[j][j];
    }
  return 1;
}
static void chol_solve(int n, double L[][OM_MAXV], double* x) {
  for (int i = 0; i < n; i++) {
    double s = x[i];
    for (int k = 0; k < i; k++) s -= L[i][k] * x[k];
    x[i] = s / L[i][i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = x[i];
    for (int k = i + 1; k < n; k++) s -= L[k][i] * x[k];
    x[i] = s / L[i][i];
  }
}

static void om_crb(const om_model* m, om_data* d) {
  sinertia crb[OM_MAXBODY];
  for (int b = 0; b < m->nbody; b++) om_body_inertia(m, d, b, &crb[b]);
  for (int b = m->nbody - 1; b > 0; b--) {
    int p = m->body_parent[b];
    for (int k = 0; k < 6; k++) crb[p].I[k] += crb[b].I[k];
    for (int k = 0; k < 3; k++) crb[p].h[k] += crb[b].h[k];
    crb[p].mass += crb[b].mass;
  }
  for (int i = 0; i < m->nv; i++)
    for (int j = 0; j < m->nv; j++) d->M[i][j] = 0;
  for (int i = 0; i < m->nv; i++) {
    double f[6];
    sinertia_mul(f, &crb[m->dof_body[i]], d->S[i]);
    for (int j = i; j >= 0; j = m->dof_parent[j]) {
      double v = 0;
      for (int k = 0; k < 6; k++) v += d->S[j][k] * f[k];
      d->M[i][j] = d->M[j][i] = v;
    }
    d->M[i][i] += m->dof_armature[i];
  }
  chol_factor(m->nv, d->M, d->Lm);
}

/* translational / rotational Jacobian of a world point attached to body b */
static void om_jac(const om_model* m, const om_data* d, int b, const double* p, double jt[3][OM_MAXV], double jr[3][OM_MAXV]) {
  for (int k = 0; k < 3; k++)
    for (int i = 0; i < m->nv; i++) { jt[k][i] = 0; jr[k][i] = 0; }
  while (b > 0 && m->body_dofnum[b] == 0) b = m->body_parent[b];
  if (b <= 0) return;
  for (int i = m->body_dofadr[b] + m->body_dofnum[b] - 1; i >= 0; i = m->dof_parent[i]) {
    double wxp[3];
    cross3(wxp, d->S[i], p);
    for (int k = 0; k < 3; k++) { jt[k][i] = d->S[i][3 + k] + wxp[k]; jr[k][i] = d->S[i][k]; }
  }
}

/* ------------------------------------------------------------ collision */
static void make_frame(double* frame) {
  /* mju_makeFrame: frame[0:3] is the normal; build two tangents */
  double n = norm3(frame);
  for (int k = 0; k < 3; k++) frame[k] /= n;
  double y[3];
  if (frame[1] > -0.5 && frame[1] < 0.5) { y[0] = 0; y[1] = 1; y[2] = 0; } else { y[0] = 0; y[1] = 0; y[2] = 1; }
  double dp = dot3(frame, y);
  for (int k = 0; k < 3; k++) y[k] -= dp * frame[k];
  n = norm3(y);
  for (int k = 0; k < 3; k++) frame[3 + k] = y[k] / n;
  cross3(frame + 6, frame, frame + 3);
}

static int add_contact(const om_model* m, om_data* d, int g1, int g2, double dist, const double* pos, const double* normal) {
  if (d->ncon >= OM_MAXCON) return 0;
  om_contact* c = &d->contact[d->ncon];
  memset(c, 0, sizeof *c);
  c->dist = dist;
  memcpy(c->pos, pos, sizeof(double) * 3);
  memcpy(c->frame, normal, sizeof(double) * 3);
  make_frame(c->frame);
  c->geom1 = g1; c->geom2 = g2;
  /* mj_contactParam: the geom of higher priority decides; equal priority: condim and friction = element-wise max, solref / solimp
   * mixed with the weights solmix1 : solmix2 = 1 : 1 (both solref in the standard, positive format) */
  double fr[3];
  if (m->geom_priority[g1] != m->geom_priority[g2]) {
    const int gp = m->geom_priority[g1] > m->geom_priority[g2] ? g1 : g2;
    c->dim = m->geom_condim[gp];
    for (int k = 0; k < 3; k++) fr[k] = m->geom_friction[gp][k];
    memcpy(c->solref, m->geom_solref[gp], sizeof c->solref);
    memcpy(c->solimp, m->geom_solimp[gp], sizeof c->solimp);
  } else {
    c->dim = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
    for (int k = 0; k < 3; k++) fr[k] = fmax(m->geom_friction[g1][k], m->geom_friction[g2][k]);
    for (int k = 0; k < 2; k++) c->solref[k] = 0.5 * m->geom_solref[g1][k] + 0.5 * m->geom_solref[g2][k];
    for (int k = 0; k < 5; k++) c->solimp[k] = 0.5 * m->geom_solimp[g1][k] + 0.5 * m->geom_solimp[g2][k];
  }
  c->friction[0] = c->friction[1] = fr[0]; c->friction[2] = fr[1]; c->friction[3] = c->friction[4] = fr[2];
  c->includemargin = 0.0; /* margin - gap, both default 0 */
  c->exclude = (c->dist >= c->includemargin);
  c->efc_address = -1;
  d->ncon++;
  return 1;
}

/* plane (g1) vs sphere/capsule/cylinder (g2); normal points from plane into the body */
static void collide_plane(const om_model* m, om_data* d, int g1, int g2) {
  const double* pm = d->geom_xmat[g1];
  double n[3] = {pm[2], pm[5], pm[8]};
  const double* pp = d->geom_xpos[g1];
  const double* gp = d->geom_xpos[g2];
  const double* gm = d->geom_xmat[g2];
  const double* sz = m->geom_size[g2];
  double margin = 0.0, tmp[3], pos[3];
  if (m->geom_type[g2] == OM_GEOM_SPHERE) {
    for (int k = 0; k < 3; k++) tmp[k] = gp[k] - pp[k];
    double cd = dot3(tmp, n);
    if (cd > margin + sz[0]) return;
    double dist = cd - sz[0];
    for (int k = 0; k < 3; k++) pos[k] = gp[k] - n[k] * (sz[0] + 0.5 * dist);
    add_contact(m, d, g1, g2, dist, pos, n);
  } else if (m->geom_type[g2] == OM_GEOM_CAPSULE) {
    /* two end spheres (mjc_PlaneCapsule) */
    double axis[3] = {gm[2], gm[5], gm[8]};
    for (int e = -1; e <= 1; e += 2) {
      double c[3];
      for (int k = 0; k < 3; k++) c[k] = gp[k] + e * sz[1] * axis[k];
      for (int k = 0; k < 3; k++) tmp[k] = c[k] - pp[k];
      double cd = dot3(tmp, n);
      if (cd > margin + sz[0]) continue;
      double dist = cd - sz[0];
      for (int k = 0; k < 3; k++) pos[k] = c[k] - n[k] * (sz[0] + 0.5 * dist);
      add_contact(m, d, g1, g2, dist, pos, n);
    }
  } else if (m->geom_type[g2] == OM_GEOM_CYLINDER) {
    /* mjc_PlaneCylinder: deepest rim point, the opposite end of that generator, then two more points of the lower disk: A = +side first,
     * then B = -side (engine_collision_primitive.c as recalled); degenerate tests against mjMINVAL */
    double axis[3] = {gm[2], gm[5], gm[8]};
    for (int k = 0; k < 3; k++) tmp[k] = gp[k] - pp[k];
    double dist0 = dot3(tmp, n);
    double prjaxis = dot3(n, axis);
    if (prjaxis > 0) { for (int k = 0; k < 3; k++) axis[k] = -axis[k]; prjaxis = -prjaxis; }
    double vec[3];
    for (int k = 0; k < 3; k++) vec[k] = axis[k] * prjaxis - n[k];
    double len = norm3(vec);
    if (len < MJS_MINVAL) {
      /* disk parallel to plane: pick x-axis of the cylinder scaled by radius */
      for (int k = 0; k < 3; k++) vec[k] = gm[3 * k] * sz[0];
    } else {
      for (int k = 0; k < 3; k++) vec[k] *= sz[0] / len;
    }
    double prjvec = dot3(vec, n);
    for (int k = 0; k < 3; k++) axis[k] *= sz[1];
    prjaxis *= sz[1];
    int cnt = 0;
    /* first contact: deepest */
    double dd = dist0 + prjaxis + prjvec;
    if (dd > margin) return;
    for (int k = 0; k < 3; k++) pos[k] = gp[k] + vec[k] + axis[k] - n[k] * dd * 0.5;
    cnt += add_contact(m, d, g1, g2, dd, pos, n);
    /* second: opposite end of the axis, same side */
    dd = dist0 - prjaxis + prjvec;
    if (dd <= margin) {
      for (int k = 0; k < 3; k++) pos[k] = gp[k] + vec[k] - axis[k] - n[k] * dd * 0.5;
      cnt += add_contact(m, d, g1, g2, dd, pos, n);
    }
    /* two side points of the lower disk */
    double side[3];
    cross3(side, vec, axis);
    double sl = norm3(side);
    if (sl > MJS_MINVAL) {
      for (int k = 0; k < 3; k++) side[k] *= sz[0] * sqrt(3.0) * 0.5 / sl;
      dd = dist0 + prjaxis - 0.5 * prjvec;
      if (dd <= margin) {
        for (int s = 1; s >= -1; s -= 2) {
          for (int k = 0; k < 3; k++) pos[k] = gp[k] + s * side[k] + axis[k] - 0.5 * vec[k] - n[k] * dd * 0.5;
          cnt += add_contact(m, d, g1, g2, dd, pos, n);
        }
      }
    }
    (void)cnt;
  } else if (m->geom_type[g2] == OM_GEOM_BOX) {
    /* mjc_PlaneBox: every corner at or below the plane (margin 0) is a contact, at most 4; corner order
     * (-,-,-), (+,-,-), (-,+,-), (+,+,-), (-,-,+), ... (x fastest) */
    int cnt = 0;
    for (int i = 0; i < 8 && cnt < 4; i++) {
      double loc[3] = {(i & 1 ? sz[0] : -sz[0]), (i & 2 ? sz[1] : -sz[1]), (i & 4 ? sz[2] : -sz[2])}, corner[3];
      mulMatVec3(corner, gm, loc);
      for (int k = 0; k < 3; k++) corner[k] += gp[k];
      for (int k = 0; k < 3; k++) tmp[k] = corner[k] - pp[k];
      double dist = dot3(tmp, n);
      if (dist > margin) continue;
      for (int k = 0; k < 3; k++) pos[k] = corner[k] - n[k] * dist * 0.5;
      cnt += add_contact(m, d, g1, g2, dist, pos, n);
    }
  } else if (m->geom_type[g2] == OM_GEOM_MESH) {
    /* plane vs mesh = plane vs the vertices of its convex hull. ASSUMED rule (MuJoCo's mjc_PlaneConvex is third-party and
     * absent): the hull vertices in the table's farthest-point order, each one at or below the plane (margin 0)
     * is a contact, at most 4 - the shape of mjc_PlaneBox's rule, with a vertex order that spreads the four over the
     * touching face. dist = height of the vertex, pos = vertex - n dist / 2 (as for a box corner). */
    const int cat = m->geom_mesh[g2];
    const double sc = m->geom_mesh_scale[g2];
    int cnt = 0;
    for (int i = 0; i < MJS_HULL_NV[cat] && cnt < 4; i++) {
      double loc[3], vert[3];
      for (int k = 0; k < 3; k++) loc[k] = MJS_HULL_VERT_C[cat][i][k] * sc;
      mulMatVec3(vert, gm, loc);
      for (int k = 0; k < 3; k++) vert[k] += gp[k];
      for (int k = 0; k < 3; k++) tmp[k] = vert[k] - pp[k];
      double dist = dot3(tmp, n);
      if (dist > margin) continue;
      for (int k = 0; k < 3; k++) pos[k] = vert[k] - n[k] * dist * 0.5;
      cnt += add_contact(m, d, g1, g2, dist, pos, n);
    }
  }
}

/* sphere (g1) vs sphere-like point set of g2: shared tail of the sphere tests. Normal points from
 * geom1 to geom2 (MuJoCo convention). */
static void sphere_point(const om_model* m, om_data* d, int g1, int g2, const double* c1, double r1, const double* p2, double r2) {
  double v[3] = {p2[0] - c1[0], p2[1] - c1[1], p2[2] - c1[2]};
  double len = norm3(v), nrm[3], pos[3];
  double dist = len - r1 - r2;
  if (dist > 0.0) return; /* margin 0 */
  if (len < MJS_MINVAL) { nrm[0] = 1; nrm[1] = 0; nrm[2] = 0; } else for (int k = 0; k < 3; k++) nrm[k] = v[k] / len;
  for (int k = 0; k < 3; k++) pos[k] = c1[k] + nrm[k] * (r1 + 0.5 * dist);
  add_contact(m, d, g1, g2, dist, pos, nrm);
}

/* mjc_SphereCylinder: side / cap / rim cases */
static void collide_sphere_cylinder(const om_model* m, om_data* d, int g1, int g2) {
  const double* c1 = d->geom_xpos[g1];
  const double* c2 = d->geom_xpos[g2];
  const double* m2 = d->geom_xmat[g2];
  double r1 = m->geom_size[g1][0], r2 = m->geom_size[g2][0], h2 = m->geom_size[g2][1];
  double axis[3] = {m2[2], m2[5], m2[8]}, vec[3], a[3];
  for (int k = 0; k < 3; k++) vec[k] = c1[k] - c2[k];
  double x = dot3(vec, axis);
  for (int k = 0; k < 3; k++) a[k] = vec[k] - axis[k] * x;
  double a2 = dot3(a, a);
  if (fabs(x) <= h2) { /* side: sphere vs the point of the axis at the same height, radius r2 */
    double p[3];
    for (int k = 0; k < 3; k++) p[k] = c2[k] + axis[k] * x;
    sphere_point(m, d, g1, g2, c1, r1, p, r2);
  } else if (a2 <= r2 * r2) { /* cap: plane through the cap, normal along the axis */
    double sgn = x > 0 ? 1.0 : -1.0, nrm[3], pos[3];
    double dist = fabs(x) - h2 - r1;
    if (dist > 0.0) return;
    for (int k = 0; k < 3; k++) nrm[k] = -sgn * axis[k]; /* from the sphere towards the cylinder */
    for (int k = 0; k < 3; k++) pos[k] = c1[k] + nrm[k] * (r1 + 0.5 * dist);
    add_contact(m, d, g1, g2, dist, pos, nrm);
  } else { /* rim: closest point of the cap circle */
    double sgn = x > 0 ? 1.0 : -1.0, p[3], la = sqrt(a2);
    for (int k = 0; k < 3; k++) p[k] = c2[k] + axis[k] * h2 * sgn + a[k] / la * r2;
    sphere_point(m, d, g1, g2, c1, r1, p, 0.0);
  }
}

/* mjc_SphereBox: clamp the centre into the box */
static void collide_sphere_box(const om_model* m, om_data* d, int g1, int g2) {
  const double* c1 = d->geom_xpos[g1];
  const double* c2 = d->geom_xpos[g2];
  const double* m2 = d->geom_xmat[g2];
  const double* sz = m->geom_size[g2];
  double r1 = m->geom_size[g1][0], rel[3], loc[3], cl[3];
  for (int k = 0; k < 3; k++) rel[k] = c1[k] - c2[k];
  for (int k = 0; k < 3; k++) loc[k] = m2[k] * rel[0] + m2[3 + k] * rel[1] + m2[6 + k] * rel[2]; /* R^T rel */
  int inside = 1;
  for (int k = 0; k < 3; k++) {
    cl[k] = fmin(fmax(loc[k], -sz[k]), sz[k]);
    if (cl[k] != loc[k]) inside = 0;
  }
  if (!inside) {
    double pw[3];
    for (int k = 0; k < 3; k++) pw[k] = c2[k] + m2[3 * k] * cl[0] + m2[3 * k + 1] * cl[1] + m2[3 * k + 2] * cl[2];
    sphere_point(m, d, g1, g2, c1, r1, pw, 0.0);
  } else { /* centre inside the box: push out through the nearest face */
    int ax = 0;
    double best = INFINITY, sgn = 1;
    for (int k = 0; k < 3; k++) {
      double dpos = sz[k] - loc[k], dneg = sz[k] + loc[k];
      if (dpos < best) { best = dpos; ax = k; sgn = 1; }
      if (dneg < best) { best = dneg; ax = k; sgn = -1; }
    }
    double nrm[3], pos[3], dist = -(best + r1);
    for (int k = 0; k < 3; k++) nrm[k] = -sgn * m2[3 * k + ax]; /* from the sphere towards the box interior */
    for (int k = 0; k < 3; k++) pos[k] = c1[k] + nrm[k] * (r1 + 0.5 * dist);
    add_contact(m, d, g1, g2, dist, pos, nrm);
  }
}

/* ------------------------------------------------------------------ convex-convex (own MPR)
 * Minkowski portal refinement (Snethen, "XenoCollide", Game Programming Gems 7) on the Minkowski difference
 * geom1 - geom2 for the convex pairs of the Planar-Push scene (cylinder-box, box-box); the role of MuJoCo's
 * mjc_Convex -> libccd ccdMPRPenetration for mesh / cylinder pairs. One contact per pair: normal = the final
 * portal's normal (pointing from geom1 to geom2), depth = distance of that portal's plane from the origin,
 * position = midpoint of the two witness points (barycentric weights of the origin ray in the portal).
 * Plain + - * / sqrt arithmetic, compiled without FMA contraction: the HIP kernel repeats the same operations. */
typedef struct { double v[3], a[3], b[3]; } mpr_vert; /* v = a - b, a on geom1, b on geom2 */
/* Flat-on-flat and line contacts (a block lying on the floor pushed by a vertical cylinder, block on block) make many
 * of the sign tests below exact ties in exact arithmetic; biased thresholds decide every such tie the same way
 * whatever the rounding noise of the inputs, which keeps the contact point a deterministic function of the pose.
 * MPR_EPS_DIR: components of unit directions; MPR_EPS_LEN: products with one length (~1e-2 m);
 * MPR_EPS_VOL: triple products of Minkowski-difference points (~1e-5 m^3). */
#define MPR_EPS_DIR 1e-10
#define MPR_EPS_LEN 1e-13
#define MPR_EPS_VOL 1e-16
#define MPR_EPS_TIE 1e-12

static void support_geom(const om_model* m, const om_data* d, int g, const double* dir, double* out) {
  const double* gp = d->geom_xpos[g];
  const double* gm = d->geom_xmat[g];
  const double* sz = m->geom_size[g];
  double loc[3], res[3];
  for (int k = 0; k < 3; k++) loc[k] = gm[k] * dir[0] + gm[3 + k] * dir[1] + gm[6 + k] * dir[2]; /* R^T dir */
  if (m->geom_type[g] == OM_GEOM_BOX) {
    for (int k = 0; k < 3; k++) res[k] = loc[k] >= -MPR_EPS_DIR ? sz[k] : -sz[k];
  } else if (m->geom_type[g] == OM_GEOM_MESH) {
    /* hull vertex with the largest projection on the direction; structural ties (a face or an edge square to the direction) go to
     * the LOWEST table index among the vertices within MPR_EPS_TIE of the maximum, whatever the rounding noise (two passes: the
     * rule does not depend on the order of evaluation, so the kernels may split the scan over lanes) */
    const int cat = m->geom_mesh[g];
    const double sc = m->geom_mesh_scale[g];
    const double ls[3] = {loc[0] * sc, loc[1] * sc, loc[2] * sc};
    double best = -1e300;
    int idx = 0;
    for (int i = 0; i < MJS_HULL_NV[cat]; i++) {
      double pr = ls[0] * MJS_HULL_VERT_C[cat][i][0] + ls[1] * MJS_HULL_VERT_C[cat][i][1] + ls[2] * MJS_HULL_VERT_C[cat][i][2];
      if (pr > best) best = pr;
    }
    for (int i = 0; i < MJS_HULL_NV[cat]; i++) {
      double pr = ls[0] * MJS_HULL_VERT_C[cat][i][0] + ls[1] * MJS_HULL_VERT_C[cat][i][1] + ls[2] * MJS_HULL_VERT_C[cat][i][2];
      if (pr >= best - MPR_EPS_TIE) { idx = i; break; }
    }
    for (int k = 0; k < 3; k++) res[k] = MJS_HULL_VERT_C[cat][idx][k] * sc;
  } else { /* cylinder, axis = local z */
    double len = sqrt(loc[0] * loc[0] + loc[1] * loc[1]);
    if (len > MPR_EPS_DIR) { res[0] = sz[0] * loc[0] / len; res[1] = sz[0] * loc[1] / len; } else { res[0] = 0; res[1] = 0; }
    res[2] = loc[2] >= -MPR_EPS_DIR ? sz[1] : -sz[1];
  }
  mulMatVec3(out, gm, res);
  for (int k = 0; k < 3; k++) out[k] += gp[k];
}
static void mpr_support(const om_model* m, const om_data* d, int g1, int g2, const double* dir, mpr_vert* s) {
  double nd[3] = {-dir[0], -dir[1], -dir[2]};
  support_geom(m, d, g1, dir, s->a);
  support_geom(m, d, g2, nd, s->b);
  for (int k = 0; k < 3; k++) s->v[k] = s->a[k] - s->b[k];
}
static int normalize3(double* v) {
  double n = norm3(v);
  if (n < 1e-14) return 0;
  for (int k = 0; k < 3; k++) v[k] /= n;
  return 1;
}
static void any_perpendicular(double* out, const double* v) {
  double ax[3] = {0, 0, 0};
  int k = fabs(v[0]) <= fabs(v[1]) ? (fabs(v[0]) <= fabs(v[2]) ? 0 : 2) : (fabs(v[1]) <= fabs(v[2]) ? 1 : 2);
  ax[k] = 1;
  cross3(out, v, ax);
}
/* returns 1 and fills (depth >= 0, normal geom1->geom2, pos) when the geoms overlap */
long om_dbg_mpr_calls = 0, om_dbg_mpr_iters = 0, om_dbg_mpr_max = 0, om_dbg_mpr_hits = 0;
static int mpr_penetration(const om_model* m, const om_data* d, int g1, int g2, double* depth, double* normal, double* pos) {
  om_dbg_mpr_calls++;
  mpr_vert v0, v1, v2, v3, v4;
  double dir[3], t1[3], t2[3];
  for (int k = 0; k < 3; k++) { v0.a[k] = d->geom_xpos[g1][k]; v0.b[k] = d->geom_xpos[g2][k]; v0.v[k] = v0.a[k] - v0.b[k]; }
  if (norm3(v0.v) < 1e-12) v0.v[0] = 1e-5; /* coincident centres: any interior direction */
  for (int k = 0; k < 3; k++) dir[k] = -v0.v[k];
  normalize3(dir);
  mpr_support(m, d, g1, g2, dir, &v1);
  if (dot3(v1.v, dir) <= 0) return 0;
  cross3(dir, v0.v, v1.v);
  if (!normalize3(dir)) { any_perpendicular(dir, v0.v); normalize3(dir); } /* origin on the line v0-v1 */
  mpr_support(m, d, g1, g2, dir, &v2);
  if (dot3(v2.v, dir) <= 0) return 0;
  for (int k = 0; k < 3; k++) { t1[k] = v1.v[k] - v0.v[k]; t2[k] = v2.v[k] - v0.v[k]; }
  cross3(dir, t1, t2);
  if (!normalize3(dir)) return 0;
  if (dot3(dir, v0.v) > MPR_EPS_LEN) { mpr_vert t = v1; v1 = v2; v2 = t; for (int k = 0; k < 3; k++) dir[k] = -dir[k]; }
  /* portal discovery */
  for (int it = 0;; it++) {
    if (it >= MJS_MPR_MAX_ITER) return 0;
    mpr_support(m, d, g1, g2, dir, &v3);
    if (dot3(v3.v, dir) <= 0) return 0;
    int cont = 0;
    cross3(t1, v1.v, v3.v);
    if (dot3(t1, v0.v) < -MPR_EPS_VOL) { v2 = v3; cont = 1; }
    else {
      cross3(t1, v3.v, v2.v);
      if (dot3(t1, v0.v) < -MPR_EPS_VOL) { v1 = v3; cont = 1; }
    }
    if (!cont) break;
    for (int k = 0; k < 3; k++) { t1[k] = v1.v[k] - v0.v[k]; t2[k] = v2.v[k] - v0.v[k]; }
    cross3(dir, t1, t2);
    if (!normalize3(dir)) return 0;
  }
  /* portal refinement */
  int hit = 0;
  for (int it = 0; it < MJS_MPR_MAX_ITER; it++) {
    om_dbg_mpr_iters++; if (it + 1 > om_dbg_mpr_max) om_dbg_mpr_max = it + 1;
    for (int k = 0; k < 3; k++) { t1[k] = v2.v[k] - v1.v[k]; t2[k] = v3.v[k] - v1.v[k]; }
    cross3(dir, t1, t2);
    if (!normalize3(dir)) return 0;
    if (dot3(dir, v1.v) >= -MPR_EPS_LEN) hit = 1; /* the origin is on the inner side of the portal: the shapes overlap */
    mpr_support(m, d, g1, g2, dir, &v4);
    double reach = dot3(v4.v, dir);
    if (!hit && reach < 0) return 0; /* the support plane separates the origin */
    for (int k = 0; k < 3; k++) t1[k] = v4.v[k] - v3.v[k];
    double progress = dot3(t1, dir);
    if (progress <= MJS_MPR_TOLERANCE || it == MJS_MPR_MAX_ITER - 1) {
      if (!hit) return 0;
      /* converged: the portal v1 v2 v3 lies on the surface of the Minkowski difference */
      *depth = dot3(v1.v, dir);
      for (int k = 0; k < 3; k++) normal[k] = dir[k]; /* portal normal points away from v0 = c1 - c2: geom1 -> geom2 */
      /* barycentric coordinates of the origin in the tetrahedron (v0, v1, v2, v3) by Cramer's rule:
       * [v1 v2 v3] v0 - [v0 v2 v3] v1 + [v0 v1 v3] v2 - [v0 v1 v2] v3 = 0 */
      double c23[3], c13[3], c12[3], b0, b1, b2, b3, sum;
      cross3(c23, v2.v, v3.v); cross3(c13, v1.v, v3.v); cross3(c12, v1.v, v2.v);
      b0 = dot3(v1.v, c23); b1 = -dot3(v0.v, c23); b2 = dot3(v0.v, c13); b3 = -dot3(v0.v, c12);
      sum = b0 + b1 + b2 + b3;
      if (fabs(sum) < 1e-30) { b0 = 0; b1 = b2 = b3 = 1; sum = 3; } /* degenerate: portal centroid */
      for (int k = 0; k < 3; k++) {
        double pa = (b0 * v0.a[k] + b1 * v1.a[k] + b2 * v2.a[k] + b3 * v3.a[k]) / sum;
        double pb = (b0 * v0.b[k] + b1 * v1.b[k] + b2 * v2.b[k] + b3 * v3.b[k]) / sum;
        pos[k] = 0.5 * (pa + pb);
      }
      return 1;
    }
    /* expand the portal with v4: keep the face the ray origin->(-v0) passes through */
    cross3(t1, v4.v, v0.v);
    if (dot3(v1.v, t1) > MPR_EPS_VOL) {
      if (dot3(v2.v, t1) > MPR_EPS_VOL) v1 = v4; else v3 = v4;
    } else {
      if (dot3(v3.v, t1) > MPR_EPS_VOL) v2 = v4; else v1 = v4;
    }
  }
  return 0;
}
static double geom_rbound(const om_model* m, int g) {
  const double* sz = m->geom_size[g];
  if (m->geom_type[g] == OM_GEOM_BOX) return sqrt(sz[0] * sz[0] + sz[1] * sz[1] + sz[2] * sz[2]);
  if (m->geom_type[g] == OM_GEOM_MESH) return MJS_HULL_RBOUND[m->geom_mesh[g]] * m->geom_mesh_scale[g];
  return sqrt(sz[0] * sz[0] + sz[1] * sz[1]); /* cylinder */
}
static void collide_convex(const om_model* m, om_data* d, int g1, int g2) {
  double diff[3], depth, normal[3], pos[3];
  for (int k = 0; k < 3; k++) diff[k] = d->geom_xpos[g2][k] - d->geom_xpos[g1][k];
  double bound = geom_rbound(m, g1) + geom_rbound(m, g2);
  if (dot3(diff, diff) > bound * bound) return; /* bounding-sphere filter (margin 0) */
  if (!mpr_penetration(m, d, g1, g2, &depth, normal, pos)) return;
  add_contact(m, d, g1, g2, -depth, pos, normal);
}

/* debug hook for tests: penetration of two convex geoms (type, size[3], pos[3], xmat[9] row-major each);
 * returns 1 when overlapping and fills out = {depth, normal[3], pos[3]} */
int om_debug_convex(int type1, const double* size1, const double* pos1, const double* mat1, int type2, const double* size2, const double* pos2,
                    const double* mat2, double* out) {
  static __thread om_model m;
  static __thread om_data d;
  m.ngeom = 2;
  m.geom_type[0] = type1; m.geom_type[1] = type2;
  memcpy(m.geom_size[0], size1, sizeof(double) * 3); memcpy(m.geom_size[1], size2, sizeof(double) * 3);
  memcpy(d.geom_xpos[0], pos1, sizeof(double) * 3); memcpy(d.geom_xpos[1], pos2, sizeof(double) * 3);
  memcpy(d.geom_xmat[0], mat1, sizeof(double) * 9); memcpy(d.geom_xmat[1], mat2, sizeof(double) * 9);
  return mpr_penetration(&m, &d, 0, 1, out, out + 1, out + 4);
}

static int body_weld(const om_model* m, int b) { return m->body_weldid[b]; }

static void om_collision(const om_model* m, om_data* d) {
  d->ncon = 0;
  for (int g1 = 0; g1 < m->ngeom; g1++)
    for (int g2 = g1 + 1; g2 < m->ngeom; g2++) {
      int b1 = m->geom_body[g1], b2 = m->geom_body[g2];
      int w1 = body_weld(m, b1), w2 = body_weld(m, b2);
      if (w1 == w2) continue; /* same (welded) body, incl. static-static */
      if (!((m->geom_contype[g1] & m->geom_conaffinity[g2]) || (m->geom_contype[g2] & m->geom_conaffinity[g1]))) continue;
      if (w1 != 0 && w2 != 0) {
        int p1 = body_weld(m, m->body_parent[w1]), p2 = body_weld(m, m->body_parent[w2]);
        if (w1 == p2 || w2 == p1) continue; /* parent-child filter */
      }
      int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
      int ga = g1, gb = g2; /* MuJoCo orders a pair by geom type */
      if (t1 > t2) { ga = g2; gb = g1; int tt = t1; t1 = t2; t2 = tt; }
      if (t1 == OM_GEOM_PLANE && t2 != OM_GEOM_PLANE) collide_plane(m, d, ga, gb);
      else if (t1 == OM_GEOM_SPHERE && t2 == OM_GEOM_CYLINDER) collide_sphere_cylinder(m, d, ga, gb);
      else if (t1 == OM_GEOM_SPHERE && t2 == OM_GEOM_BOX) collide_sphere_box(m, d, ga, gb);
      else if ((t1 == OM_GEOM_CYLINDER || t1 == OM_GEOM_BOX || t1 == OM_GEOM_MESH) && (t2 == OM_GEOM_BOX || t2 == OM_GEOM_MESH)) collide_convex(m, d, ga, gb);
      /* other pairs (capsule-capsule, capsule-box, ...): not evaluated */
    }
}

/* ----------------------------------------------------- constraint rows */
static int add_row(om_data* d, int nv, const double* J, double pos, double margin, int type, int id) {
  if (d->nefc >= OM_MAXEFC) return -1;
  int r = d->nefc++;
  memcpy(d->efc_J[r], J, sizeof(double) * nv);
  d->efc_pos[r] = pos; d->efc_margin[r] = margin; d->efc_type[r] = type; d->efc_id[r] = id;
  return r;
}

static void om_make_constraint(const om_model* m, om_data* d) {
  int nv = m->nv;
  d->nefc = d->ne = d->nl = 0;
  double jt1[3][OM_MAXV], jr1[3][OM_MAXV], jt2[3][OM_MAXV], jr2[3][OM_MAXV], J[OM_MAXV];
  /* equality (mj_instantiateEquality): connect, weld, joint coupling, in the order of definition */
  for (int e = 0; e < m->neq; e++) {
    int b1 = m->eq_body1[e], b2 = m->eq_body2[e];
    const double* data = m->eq_data[e];
    if (m->eq_type[e] == OM_EQ_CONNECT) {
      /* mjEQ_CONNECT: the anchor (data[0:3] in body1, data[3:6] in body2: the same world point at qpos0) must coincide: 3 rows */
      double p1[3], p2[3], tmp[3];
      mulMatVec3(tmp, d->xmat[b1], data);
      for (int k = 0; k < 3; k++) p1[k] = d->xpos[b1][k] + tmp[k];
      mulMatVec3(tmp, d->xmat[b2], data + 3);
      for (int k = 0; k < 3; k++) p2[k] = d->xpos[b2][k] + tmp[k];
      om_jac(m, d, b1, p1, jt1, jr1);
      om_jac(m, d, b2, p2, jt2, jr2);
      for (int k = 0; k < 3; k++) {
        for (int i = 0; i < nv; i++) J[i] = jt1[k][i] - jt2[k][i];
        add_row(d, nv, J, p1[k] - p2[k], 0, OM_CNSTR_EQUALITY, e);
      }
      continue;
    }
    if (m->eq_type[e] == OM_EQ_JOINT) {
      /* mjEQ_JOINT: q1 - q1_0 = polycoef(q2 - q2_0), one row; eq_body1/2 hold the JOINT ids */
      const int q1a = m->jnt_qposadr[b1], q2a = m->jnt_qposadr[b2];
      const double x1 = d->qpos[q1a] - m->qpos0[q1a], x2 = d->qpos[q2a] - m->qpos0[q2a];
      const double poly = data[0] + x2 * (data[1] + x2 * (data[2] + x2 * (data[3] + x2 * data[4])));
      const double deriv = data[1] + x2 * (2 * data[2] + x2 * (3 * data[3] + x2 * 4 * data[4]));
      memset(J, 0, sizeof(double) * nv);
      J[m->jnt_dofadr[b1]] = 1;
      J[m->jnt_dofadr[b2]] = -deriv;
      add_row(d, nv, J, x1 - poly, 0, OM_CNSTR_EQUALITY, e);
      continue;
    }
    double pos0[3], pos1[3], tmp[3], cpos[6];
    mulMatVec3(tmp, d->xmat[b1], data + 3);
    for (int k = 0; k < 3; k++) pos0[k] = d->xpos[b1][k] + tmp[k];
    mulMatVec3(tmp, d->xmat[b2], data);
    for (int k = 0; k < 3; k++) pos1[k] = d->xpos[b2][k] + tmp[k];
    for (int k = 0; k < 3; k++) cpos[k] = pos0[k] - pos1[k];
    om_jac(m, d, b1, pos0, jt1, jr1);
    om_jac(m, d, b2, pos1, jt2, jr2);
    double torquescale = data[10];
    double quat[4], quat1[4], quat2[4];
    mulQuat(quat, d->xquat[b1], data + 6);
    negQuat(quat1, d->xquat[b2]);
    mulQuat(quat2, quat1, quat);
    for (int k = 0; k < 3; k++) cpos[3 + k] = quat2[1 + k] * torquescale;
    for (int k = 0; k < 3; k++) {
      for (int i = 0; i < nv; i++) J[i] = jt1[k][i] - jt2[k][i];
      add_row(d, nv, J, cpos[k], 0, OM_CNSTR_EQUALITY, e);
    }
    /* rotational rows: 0.5 * neg(q2) * (jr1-jr2) * q1 * relpose, axis part, scaled */
    double Jrot[3][OM_MAXV];
    for (int i = 0; i < nv; i++) {
      double ax[4] = {0, jr1[0][i] - jr2[0][i], jr1[1][i] - jr2[1][i], jr1[2][i] - jr2[2][i]};
      double q3[4], q4[4];
      mulQuat(q3, ax, quat);
      mulQuat(q4, quat1, q3);
      for (int k = 0; k < 3; k++) Jrot[k][i] = 0.5 * q4[1 + k] * torquescale;
    }
    for (int k = 0; k < 3; k++) add_row(d, nv, Jrot[k], cpos[3 + k], 0, OM_CNSTR_EQUALITY, e);
  }
  d->ne = d->nefc;
  /* joint limits (mj_instantiateLimit): lower side first */
  for (int j = 0; j < m->njnt; j++) {
    if (!m->jnt_limited[j]) continue;
    if (m->jnt_type[j] != OM_JNT_HINGE && m->jnt_type[j] != OM_JNT_SLIDE) continue;
    double value = d->qpos[m->jnt_qposadr[j]], margin = m->jnt_margin[j];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (m->jnt_range[j][(side + 1) / 2] - value);
      if (dist < margin) {
        memset(J, 0, sizeof(double) * nv);
        J[m->jnt_dofadr[j]] = -side;
        add_row(d, nv, J, dist, margin, OM_CNSTR_LIMIT_JOINT, j);
      }
    }
  }
  d->nl = d->nefc - d->ne;
  /* contacts (mj_instantiateContact), pyramidal cone */
  for (int c = 0; c < d->ncon; c++) {
    om_contact* con = &d->contact[c];
    if (con->exclude) continue;
    int b1 = m->geom_body[con->geom1], b2 = m->geom_body[con->geom2];
    om_jac(m, d, b1, con->pos, jt1, jr1);
    om_jac(m, d, b2, con->pos, jt2, jr2);
    /* rows of the contact frame applied to (jac2 - jac1) */
    double Jc[6][OM_MAXV];
    int allzero = 1;
    for (int r = 0; r < 3; r++)
      for (int i = 0; i < nv; i++) {
        double vt = 0, vr = 0;
        for (int k = 0; k < 3; k++) {
          vt += con->frame[3 * r + k] * (jt2[k][i] - jt1[k][i]);
          vr += con->frame[3 * r + k] * (jr2[k][i] - jr1[k][i]);
        }
        Jc[r][i] = vt; Jc[3 + r][i] = vr;
        if (vt != 0 || (con->dim > 3 && vr != 0)) allzero = 0;
      }
    if (allzero) { con->exclude = 3; continue; } /* no dofs */
    con->efc_address = d->nefc;
    if (con->dim == 1) {
      add_row(d, nv, Jc[0], con->dist, con->includemargin, OM_CNSTR_CONTACT_FRICTIONLESS, c);
    } else if (m->cone == OM_CONE_ELLIPTIC) {
      /* elliptic cone: the contact frame's own rows - normal, then tangents (torsion, rolling for condim 4, 6); only the normal
       * row carries the distance and the margin */
      add_row(d, nv, Jc[0], con->dist, con->includemargin, OM_CNSTR_CONTACT_ELLIPTIC, c);
      for (int k = 1; k < con->dim; k++) add_row(d, nv, Jc[k], 0, 0, OM_CNSTR_CONTACT_ELLIPTIC, c);
    } else {
      for (int k = 1; k < con->dim; k++) {
        /* Jc row order: normal, tangent1, tangent2, torsion(=rot normal), roll1, roll2 */
        const double* Jk = (k < 3) ? Jc[k] : Jc[k];
        double mu = con->friction[k - 1];
        for (int s = 1; s >= -1; s -= 2) {
          for (int i = 0; i < nv; i++) J[i] = Jc[0][i] + s * mu * Jk[i];
          add_row(d, nv, J, con->dist, con->includemargin, OM_CNSTR_CONTACT_PYRAMIDAL, c);
        }
      }
    }
  }
}

/* mj_makeImpedance helper */
static void get_impedance(const double* solimp, double pos, double margin, double* imp, double* impP) {
  if (solimp[0] == solimp[1] || solimp[2] <= MJS_MINVAL) { *imp = 0.5 * (solimp[0] + solimp[1]); *impP = 0; return; }
  double x = (pos - margin) / solimp[2], sgn = 1;
  if (x < 0) { x = -x; sgn = -1; }
  if (x >= 1 || x <= 0) { *imp = (x >= 1 ? solimp[1] : solimp[0]); *impP = 0; return; }
  double y, yP;
  if (solimp[4] == 1) { y = x; yP = 1; }
  else if (x <= solimp[3]) {
    double a = 1 / pow(solimp[3], solimp[4] - 1);
    y = a * pow(x, solimp[4]); yP = solimp[4] * a * pow(x, solimp[4] - 1);
  } else {
    double b = 1 / pow(1 - solimp[3], solimp[4] - 1);
    y = 1 - b * pow(1 - x, solimp[4]); yP = solimp[4] * b * pow(1 - x, solimp[4] - 1);
  }
  *imp = solimp[0] + y * (solimp[1] - solimp[0]);
  *impP = yP * sgn * (solimp[1] - solimp[0]) / solimp[2];
}

/* mj_diagApprox + mj_makeImpedance + mj_referenceConstraint (aref needs efc_vel) */
static void om_make_impedance(const om_model* m, om_data* d) {
  int nv = m->nv;
  /* diagApprox from body/dof inverse weights at qpos0 */
  for (int i = 0; i < d->nefc; i++) {
    int id = d->efc_id[i];
    if (d->efc_type[i] == OM_CNSTR_EQUALITY && m->eq_type[id] == OM_EQ_JOINT) {
      d->efc_diagApprox[i] = m->dof_invweight0[m->jnt_dofadr[m->eq_body1[id]]] + m->dof_invweight0[m->jnt_dofadr[m->eq_body2[id]]];
    } else if (d->efc_type[i] == OM_CNSTR_EQUALITY && m->eq_type[id] == OM_EQ_CONNECT) {
      int b1 = m->eq_body1[id], b2 = m->eq_body2[id];
      double tran = m->body_invweight0[b1][0] + m->body_invweight0[b2][0];
      for (int k = 0; k < 3; k++) d->efc_diagApprox[i + k] = tran;
      i += 2;
    } else if (d->efc_type[i] == OM_CNSTR_EQUALITY) {
      int b1 = m->eq_body1[id], b2 = m->eq_body2[id];
      double tran = m->body_invweight0[b1][0] + m->body_invweight0[b2][0];
      double rot = m->body_invweight0[b1][1] + m->body_invweight0[b2][1];
      for (int k = 0; k < 3; k++) { d->efc_diagApprox[i + k] = tran; d->efc_diagApprox[i + 3 + k] = rot; }
      i += 5;
    } else if (d->efc_type[i] == OM_CNSTR_LIMIT_JOINT) {
      d->efc_diagApprox[i] = m->dof_invweight0[m->jnt_dofadr[id]];
    } else {
      const om_contact* con = &d->contact[id];
      int b1 = m->geom_body[con->geom1], b2 = m->geom_body[con->geom2];
      double tran = m->body_invweight0[b1][0] + m->body_invweight0[b2][0];
      double rot = m->body_invweight0[b1][1] + m->body_invweight0[b2][1];
      if (d->efc_type[i] == OM_CNSTR_CONTACT_FRICTIONLESS) d->efc_diagApprox[i] = tran;
      else if (d->efc_type[i] == OM_CNSTR_CONTACT_ELLIPTIC) {
        /* only the normal row's value survives: the friction rows' R is derived from the normal row's below */
        for (int j = 0; j < con->dim; j++) d->efc_diagApprox[i + j] = j < 3 ? tran : rot;
        i += con->dim - 1;
      } else {
        int nrow = 2 * (con->dim - 1);
        for (int j = 0; j < nrow; j++) {
          double fri = con->friction[j / 2];
          d->efc_diagApprox[i + j] = tran + fri * fri * (j < 4 ? tran : rot);
        }
        i += nrow - 1;
      }
    }
  }
  for (int i = 0; i < d->nefc;) {
    int id = d->efc_id[i], dim = 1;
    double pos = d->efc_pos[i], solref[2], solimp[5];
    memcpy(solref, m->solref, sizeof solref);
    memcpy(solimp, m->solimp, sizeof solimp);
    if (d->efc_type[i] == OM_CNSTR_EQUALITY) {
      memcpy(solref, m->eq_solref[id], sizeof solref);
      memcpy(solimp, m->eq_solimp[id], sizeof solimp);
      dim = m->eq_type[id] == OM_EQ_WELD ? 6 : m->eq_type[id] == OM_EQ_CONNECT ? 3 : 1;
      if (dim > 1) { /* getposdim: weld / connect use the norm of the 6- / 3-vector residual */
        pos = 0;
        for (int k = 0; k < dim; k++) pos += d->efc_pos[i + k] * d->efc_pos[i + k];
        pos = sqrt(pos);
      }
    } else if (d->efc_type[i] == OM_CNSTR_LIMIT_JOINT) {
      memcpy(solref, m->jnt_solref[id], sizeof solref);
      memcpy(solimp, m->jnt_solimp[id], sizeof solimp);
    } else if (d->efc_type[i] == OM_CNSTR_CONTACT_ELLIPTIC) {
      const om_contact* con = &d->contact[id];
      memcpy(solref, con->solref, sizeof solref);
      memcpy(solimp, con->solimp, sizeof solimp);
      dim = con->dim;
      pos = fabs(d->efc_pos[i]); /* norm of (dist, 0, ..): the impedance is even in pos - margin */
    } else if (d->efc_type[i] == OM_CNSTR_CONTACT_PYRAMIDAL) {
      const om_contact* con = &d->contact[id];
      memcpy(solref, con->solref, sizeof solref);
      memcpy(solimp, con->solimp, sizeof solimp);
      dim = 2 * (con->dim - 1);
    } else if (d->efc_type[i] == OM_CNSTR_CONTACT_FRICTIONLESS) {
      memcpy(solref, d->contact[id].solref, sizeof solref);
      memcpy(solimp, d->contact[id].solimp, sizeof solimp);
    }
    /* refsafe */
    if (solref[0] > 0 && solref[0] < 2 * m->dt) solref[0] = 2 * m->dt;
    double imp, impP;
    get_impedance(solimp, pos, d->efc_margin[i], &imp, &impP);
    double K, B, dmax = solimp[1];
    if (solref[0] > 0) {
      K = 1 / fmax(MJS_MINVAL, dmax * dmax * solref[0] * solref[0] * solref[1] * solref[1]);
      B = 2 / fmax(MJS_MINVAL, dmax * solref[0]);
    } else { K = -solref[0] / fmax(MJS_MINVAL, dmax * dmax); B = -solref[1] / fmax(MJS_MINVAL, dmax); }
    for (int j = 0; j < dim; j++) {
      d->efc_R[i + j] = fmax(MJS_MINVAL, (1 - imp) * d->efc_diagApprox[i + j] / imp);
      d->efc_KBIP[i + j][0] = K; d->efc_KBIP[i + j][1] = B; d->efc_KBIP[i + j][2] = imp; d->efc_KBIP[i + j][3] = impP;
    }
    if (d->efc_type[i] == OM_CNSTR_CONTACT_PYRAMIDAL) {
      om_contact* con = &d->contact[id];
      con->mu = con->friction[0] * sqrt(1 / m->impratio);
      double Rpy = 2 * con->mu * con->mu * d->efc_R[i];
      for (int j = 0; j < dim; j++) d->efc_R[i + j] = Rpy;
    }
    if (d->efc_type[i] == OM_CNSTR_CONTACT_ELLIPTIC) {
      /* friction rows: R_tangent = R_normal / impratio, the other friction dimensions scaled so that the cone keeps its shape
       * (R_j = R_tangent mu_1^2 / mu_j^2); the cone's slope in the solver's variables becomes mu = mu_1 sqrt(R_tangent / R_normal) */
      om_contact* con = &d->contact[id];
      d->efc_R[i + 1] = d->efc_R[i] / fmax(MJS_MINVAL, m->impratio);
      for (int j = 2; j < dim; j++) d->efc_R[i + j] = d->efc_R[i + 1] * con->friction[0] * con->friction[0] / (con->friction[j - 1] * con->friction[j - 1]);
      con->mu = con->friction[0] * sqrt(d->efc_R[i + 1] / d->efc_R[i]);
    }
    for (int j = 0; j < dim; j++) d->efc_D[i + j] = 1 / d->efc_R[i + j];
    i += dim;
  }
  (void)nv;
}

static void om_reference_constraint(const om_model* m, om_data* d) {
  for (int i = 0; i < d->nefc; i++) {
    double v = 0;
    for (int k = 0; k < m->nv; k++) v += d->efc_J[i][k] * d->qvel[k];
    d->efc_vel[i] = v;
    d->efc_aref[i] = -d->efc_KBIP[i][1] * v - d->efc_KBIP[i][0] * d->efc_KBIP[i][2] * (d->efc_pos[i] - d->efc_margin[i]);
  }
}

/* ------------------------------------------------ velocity / force stages */
static void om_com_vel(const om_model* m, om_data* d) {
  memset(d->cvel[0], 0, sizeof d->cvel[0]);
  for (int b = 1; b < m->nbody; b++) {
    memcpy(d->cvel[b], d->cvel[m->body_parent[b]], sizeof d->cvel[b]);
    for (int i = m->body_dofadr[b]; i < m->body_dofadr[b] + m->body_dofnum[b]; i++)
      for (int k = 0; k < 6; k++) d->cvel[b][k] += d->S[i][k] * d->qvel[i];
  }
}

/* mj_rne with flg_acc=0: Coriolis/centrifugal + gravity */
static void om_rne_bias(const om_model* m, om_data* d) {
  double cacc[OM_MAXBODY][6], cfrc[OM_MAXBODY][6];
  memset(cacc[0], 0, sizeof cacc[0]);
  for (int k = 0; k < 3; k++) cacc[0][3 + k] = -m->gravity[k];
  memset(cfrc[0], 0, sizeof cfrc[0]);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parent[b];
    double v[6];
    memcpy(cacc[b], cacc[p], sizeof cacc[b]);
    memcpy(v, d->cvel[p], sizeof v);
    for (int i = m->body_dofadr[b]; i < m->body_dofadr[b] + m->body_dofnum[b]; i++) {
      double sd[6];
      int j = m->dof_jnt[i];
      if (m->jnt_type[j] == OM_JNT_FREE && i - m->jnt_dofadr[j] < 3) memset(sd, 0, sizeof sd);
      else if (m->jnt_type[j] == OM_JNT_FREE) cross_motion(sd, d->cvel[b], d->S[i]); /* local axes move with the body */
      else cross_motion(sd, v, d->S[i]);
      for (int k = 0; k < 6; k++) { cacc[b][k] += sd[k] * d->qvel[i]; v[k] += d->S[i][k] * d->qvel[i]; }
    }
    sinertia I;
    om_body_inertia(m, d, b, &I);
    double Ia[6], Iv[6], vIv[6];
    sinertia_mul(Ia, &I, cacc[b]);
    sinertia_mul(Iv, &I, d->cvel[b]);
    cross_force(vIv, d->cvel[b], Iv);
    for (int k = 0; k < 6; k++) cfrc[b][k] = Ia[k] + vIv[k];
  }
  for (int b = m->nbody - 1; b > 0; b--)
    for (int k = 0; k < 6; k++) cfrc[m->body_parent[b]][k] += cfrc[b][k];
  for (int i = 0; i < m->nv; i++) {
    double v = 0;
    for (int k = 0; k < 6; k++) v += d->S[i][k] * cfrc[m->dof_body[i]][k];
    d->qfrc_bias[i] = v;
  }
}

/* mj_passive: joint damping is added implicitly by the integrators' derivative;
 * here: damper force + gravity compensation (mj_gravcomp) */
static void om_passive(const om_model* m, om_data* d) {
  for (int i = 0; i < m->nv; i++) d->qfrc_passive[i] = -m->dof_damping[i] * d->qvel[i];
  /* joint springs (hinge / slide): -stiffness (q - springref) */
  for (int j = 0; j < m->njnt; j++)
    if (m->jnt_stiffness[j] != 0 && (m->jnt_type[j] == OM_JNT_HINGE || m->jnt_type[j] == OM_JNT_SLIDE))
      d->qfrc_passive[m->jnt_dofadr[j]] -= m->jnt_stiffness[j] * (d->qpos[m->jnt_qposadr[j]] - m->qpos_spring[m->jnt_qposadr[j]]);
  for (int b = 1; b < m->nbody; b++) {
    if (m->body_gravcomp[b] == 0) continue;
    double F[3], jt[3][OM_MAXV], jr[3][OM_MAXV];
    for (int k = 0; k < 3; k++) F[k] = -m->gravity[k] * m->body_mass[b] * m->body_gravcomp[b];
    om_jac(m, d, b, d->xipos[b], jt, jr);
    for (int i = 0; i < m->nv; i++) d->qfrc_passive[i] += jt[0][i] * F[0] + jt[1][i] * F[1] + jt[2][i] * F[2];
  }
}

/* mj_fwdActuation: fixed gain, affine bias, joint transmission, force clamp */
static void om_actuation(const om_model* m, om_data* d) {
  for (int i = 0; i < m->nv; i++) d->qfrc_actuator[i] = 0;
  for (int u = 0; u < m->nu; u++) {
    int j = m->act_jnt[u];
    double ctrl = d->ctrl[u];
    if (m->act_ctrllimited[u]) ctrl = fmin(fmax(ctrl, m->act_ctrlrange[u][0]), m->act_ctrlrange[u][1]);
    double len, vel;
    if (m->act_trntype[u] == OM_TRN_TENDON) {
      /* fixed tendon (mj_tendon + mj_transmission): length = sum coef q, the moment arm is the coefficient vector */
      len = 0; vel = 0;
      for (int k = 0; k < m->nv; k++)
        if (m->tendon_coef[j][k] != 0) { len += m->tendon_coef[j][k] * d->qpos[m->jnt_qposadr[m->dof_jnt[k]]]; vel += m->tendon_coef[j][k] * d->qvel[k]; }
    } else { len = d->qpos[m->jnt_qposadr[j]]; vel = d->qvel[m->jnt_dofadr[j]]; }
    double f = m->act_gain[u] * ctrl + m->act_bias[u][0] + m->act_bias[u][1] * len + m->act_bias[u][2] * vel;
    if (m->act_forcelimited[u]) f = fmin(fmax(f, m->act_forcerange[u][0]), m->act_forcerange[u][1]);
    d->actuator_force[u] = f;
    if (m->act_trntype[u] == OM_TRN_TENDON) {
      for (int k = 0; k < m->nv; k++) d->qfrc_actuator[k] += m->tendon_coef[j][k] * f;
    } else
      d->qfrc_actuator[m->jnt_dofadr[j]] += f;
  }
}

/* ------------------------------------------------------ Newton solver */
typedef struct {
  double cost, gauss;
} cstate;

/* elliptic cone of one contact in the solver's variables (PrimalUpdateConstraint): with z = jar of the contact's rows,
 * U = (mu z_0, f_1 z_1, .., f_{dim-1} z_{dim-1}), N = U_0, T = |U_1..|, the row weights D_j = D_0 f_j^2 / mu^2 make the quadratic
 * cost isotropic in U (weight D_0 / mu^2). Zones: top N >= mu T (no force), bottom mu N + T <= 0 (all rows quadratic), middle:
 * squared distance to the cone N = mu T, cost = Dm (N - mu T)^2 / 2 with Dm = D_0 / (mu^2 (1 + mu^2)). */
typedef struct { double N, T, mu, Dm, U[6]; int zone; } cone_state;
static void cone_eval(const om_contact* con, const double* z, double D0, cone_state* c) {
  c->mu = con->mu;
  c->U[0] = z[0] * con->mu;
  double tt = 0;
  for (int j = 1; j < con->dim; j++) { c->U[j] = z[j] * con->friction[j - 1]; tt += c->U[j] * c->U[j]; }
  c->N = c->U[0]; c->T = sqrt(tt);
  c->Dm = D0 / (con->mu * con->mu * (1 + con->mu * con->mu));
  c->zone = (c->N >= con->mu * c->T) ? 0 : (con->mu * c->N + c->T <= 0) ? 2 : 1;
}

static double constraint_update(const om_model* m, om_data* d, const double* jar, const double* qacc, const double* Ma, int* active, double* force) {
  double cost = 0;
  for (int i = 0; i < d->nefc; i++) {
    if (d->efc_type[i] == OM_CNSTR_CONTACT_ELLIPTIC) {
      om_contact* con = &d->contact[d->efc_id[i]];
      cone_state c;
      cone_eval(con, jar + i, d->efc_D[i], &c);
      con->zone = c.zone;
      for (int j = 0; j < con->dim; j++) { active[i + j] = (c.zone == 2); force[i + j] = 0; }
      if (c.zone == 2) {
        for (int j = 0; j < con->dim; j++) { force[i + j] = -d->efc_D[i + j] * jar[i + j]; cost += 0.5 * d->efc_D[i + j] * jar[i + j] * jar[i + j]; }
      } else if (c.zone == 1) {
        const double NT = c.N - c.mu * c.T;
        cost += 0.5 * c.Dm * NT * NT;
        force[i] = -c.Dm * NT * c.mu;
        for (int j = 1; j < con->dim; j++) force[i + j] = -force[i] / c.T * c.U[j] * con->friction[j - 1];
      }
      i += con->dim - 1;
      continue;
    }
    int act = (d->efc_type[i] == OM_CNSTR_EQUALITY) ? 1 : (jar[i] < 0);
    active[i] = act;
    force[i] = act ? -d->efc_D[i] * jar[i] : 0;
    if (act) cost += 0.5 * d->efc_D[i] * jar[i] * jar[i];
  }
  double gauss = 0;
  for (int i = 0; i < m->nv; i++) gauss += (Ma[i] - d->qfrc_smooth[i]) * (qacc[i] - d->qacc_smooth[i]);
  return cost + 0.5 * gauss;
}

/* exact minimiser of the piecewise-quadratic cost along `search` (role of PrimalSearch) */
/* debug statistics (tests / tuning only) */
int om_dbg_cold_start = 0; /* test knob: ignore qacc_warmstart */
long om_dbg_ls_calls = 0, om_dbg_ls_iters = 0, om_dbg_ls_max = 0, om_dbg_newton_calls = 0, om_dbg_newton_iters = 0, om_dbg_newton_max = 0;

/* 1-D Newton with bracketing on the piecewise-quadratic cost along `search`. Stops like MuJoCo's
 * PrimalSearch: when the directional derivative drops below gtol = tolerance * ls_tolerance *
 * |search| / scale (ls_tolerance = 0.01, at most ls_iterations = 50 evaluations). */
static double line_search(const om_model* m, const om_data* d, const double* jar, const double* jv, double g1, double g2, double gtol) {
  double alpha = 0, lo = 0, hi = INFINITY;
  om_dbg_ls_calls++;
  for (int it = 0; it < 50; it++) {
    om_dbg_ls_iters++;
    if (it + 1 > om_dbg_ls_max) om_dbg_ls_max = it + 1;
    double d1 = g1 + alpha * g2, d2 = g2;
    for (int i = 0; i < d->nefc; i++) {
      if (d->efc_type[i] == OM_CNSTR_CONTACT_ELLIPTIC) {
        /* the cone's cost along the ray (PrimalEval): N(alpha) linear, T(alpha)^2 quadratic in alpha */
        const om_contact* con = &d->contact[d->efc_id[i]];
        const int dim = con->dim;
        const double mu = con->mu, U0 = jar[i] * mu, V0 = jv[i] * mu;
        double UU = 0, UV = 0, VV = 0;
        for (int j = 1; j < dim; j++) {
          const double u = jar[i + j] * con->friction[j - 1], v = jv[i + j] * con->friction[j - 1];
          UU += u * u; UV += u * v; VV += v * v;
        }
        const double N = U0 + alpha * V0, Tsq = UU + alpha * (2 * UV + alpha * VV), T = Tsq > 0 ? sqrt(Tsq) : 0;
        if (N >= mu * T) { /* top zone: nothing */
        } else if (mu * N + T <= 0) { /* bottom zone: every row quadratic */
          for (int j = 0; j < dim; j++) {
            const double x = jar[i + j] + alpha * jv[i + j];
            d1 += d->efc_D[i + j] * x * jv[i + j]; d2 += d->efc_D[i + j] * jv[i + j] * jv[i + j];
          }
        } else { /* middle zone */
          const double Dm = d->efc_D[i] / (mu * mu * (1 + mu * mu));
          const double N1 = V0, T1 = (UV + alpha * VV) / T, T2 = VV / T - (UV + alpha * VV) * (UV + alpha * VV) / (T * T * T);
          const double NT = N - mu * T, NT1 = N1 - mu * T1;
          d1 += Dm * NT * NT1; d2 += Dm * (NT1 * NT1 - NT * mu * T2);
        }
        i += dim - 1;
        continue;
      }
      double x = jar[i] + alpha * jv[i];
      if (d->efc_type[i] == OM_CNSTR_EQUALITY || x < 0) { d1 += d->efc_D[i] * x * jv[i]; d2 += d->efc_D[i] * jv[i] * jv[i]; }
    }
    if (fabs(d1) < gtol) break;
    if (d1 < 0) lo = alpha; else hi = alpha;
    if (d2 <= 0) break;
    double step = -d1 / d2;
    double next = alpha + step;
    if (!(next > lo && next < hi)) next = isfinite(hi) ? 0.5 * (lo + hi) : (alpha > 0 ? 2 * alpha : 1.0);
    if (fabs(next - alpha) <= 1e-15 * fmax(1.0, fabs(alpha))) { alpha = next; break; }
    alpha = next;
  }
  (void)m;
  return alpha;
}

static void om_solve_constraint(const om_model* m, om_data* d) {
  int nv = m->nv, nefc = d->nefc;
  if (nefc == 0) {
    memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
    memset(d->qfrc_constraint, 0, sizeof(double) * nv);
    d->solver_niter = 0;
    return;
  }
  static __thread double jar[OM_MAXEFC], jv[OM_MAXEFC], force[OM_MAXEFC], H[OM_MAXV][OM_MAXV], Lh[OM_MAXV][OM_MAXV];
  static __thread int active[OM_MAXEFC];
  double qacc[OM_MAXV], Ma[OM_MAXV], grad[OM_MAXV], search[OM_MAXV], Mv[OM_MAXV];
  /* warmstart: pick the cheaper of qacc_warmstart and qacc_smooth */
  double best = INFINITY;
  for (int trial = om_dbg_cold_start ? 1 : 0; trial < 2; trial++) {
    const double* q0 = trial == 0 ? d->qacc_warmstart : d->qacc_smooth;
    double ma[OM_MAXV], jr[OM_MAXEFC];
    for (int i = 0; i < nv; i++) { ma[i] = 0; for (int k = 0; k < nv; k++) ma[i] += d->M[i][k] * q0[k]; }
    for (int i = 0; i < nefc; i++) { jr[i] = -d->efc_aref[i]; for (int k = 0; k < nv; k++) jr[i] += d->efc_J[i][k] * q0[k]; }
    double c = constraint_update(m, d, jr, q0, ma, active, force);
    if (c < best) { best = c; memcpy(qacc, q0, sizeof(double) * nv); }
  }
  for (int i = 0; i < nv; i++) { Ma[i] = 0; for (int k = 0; k < nv; k++) Ma[i] += d->M[i][k] * qacc[k]; }
  for (int i = 0; i < nefc; i++) { jar[i] = -d->efc_aref[i]; for (int k = 0; k < nv; k++) jar[i] += d->efc_J[i][k] * qacc[k]; }
  double cost = constraint_update(m, d, jar, qacc, Ma, active, force);
  double scale = 1 / (m->meaninertia * (nv > 1 ? nv : 1));
  int iter = 0;
  for (; iter < m->iterations; iter++) {
    /* gradient, Hessian, Newton direction */
    for (int i = 0; i < nv; i++) {
      grad[i] = Ma[i] - d->qfrc_smooth[i];
      for (int r = 0; r < nefc; r++) grad[i] -= d->efc_J[r][i] * force[r];
    }
    for (int i = 0; i < nv; i++)
      for (int j = 0; j <= i; j++) {
        double h = d->M[i][j];
        for (int r = 0; r < nefc; r++)
          if (active[r]) h += d->efc_J[r][i] * d->efc_D[r] * d->efc_J[r][j];
        H[i][j] = H[j][i] = h;
      }
    /* contacts on the cone surface (middle zone): H += Jc^T Hc Jc with the dim x dim Hessian of Dm (N - mu T)^2 / 2 in the
     * contact's own rows (the role of HessianCone): with v = (mu, -mu f_j t_j / T), t_j = f_j z_j,
     * Hc = Dm [ v v^T - mu (N - mu T) (diag(0, f_j^2) / T - (0, f_j t_j)(0, f_k t_k)^T / T^3) ] */
    for (int r = 0; r < nefc; r++) {
      if (d->efc_type[r] != OM_CNSTR_CONTACT_ELLIPTIC) continue;
      const om_contact* con = &d->contact[d->efc_id[r]];
      const int dim = con->dim;
      if (con->zone == 1) {
        cone_state c;
        cone_eval(con, jar + r, d->efc_D[r], &c);
        double v[6], w[6], Hc[6][6];
        const double NT = c.N - c.mu * c.T;
        v[0] = c.mu; w[0] = 0;
        for (int j = 1; j < dim; j++) { w[j] = con->friction[j - 1] * c.U[j]; v[j] = -c.mu * w[j] / c.T; }
        for (int a = 0; a < dim; a++)
          for (int b = 0; b < dim; b++) {
            double curv = -w[a] * w[b] / (c.T * c.T * c.T);
            if (a == b && a > 0) curv += con->friction[a - 1] * con->friction[a - 1] / c.T;
            Hc[a][b] = c.Dm * (v[a] * v[b] - c.mu * NT * curv);
          }
        for (int i = 0; i < nv; i++)
          for (int j = 0; j <= i; j++) {
            double h = 0;
            for (int a = 0; a < dim; a++) {
              double t = 0;
              for (int b = 0; b < dim; b++) t += Hc[a][b] * d->efc_J[r + b][j];
              h += d->efc_J[r + a][i] * t;
            }
            H[i][j] += h;
            if (i != j) H[j][i] += h;
          }
      }
      r += dim - 1;
    }
    if (!chol_factor(nv, H, Lh)) break;
    for (int i = 0; i < nv; i++) search[i] = -grad[i];
    chol_solve(nv, Lh, search);
    /* line search */
    double g1 = 0, g2 = 0, snorm = 0;
    for (int i = 0; i < nv; i++) { Mv[i] = 0; for (int k = 0; k < nv; k++) Mv[i] += d->M[i][k] * search[k]; }
    for (int i = 0; i < nv; i++) { g1 += search[i] * (Ma[i] - d->qfrc_smooth[i]); g2 += search[i] * Mv[i]; snorm += search[i] * search[i]; }
    if (sqrt(snorm) < MJS_MINVAL) break;
    for (int r = 0; r < nefc; r++) { jv[r] = 0; for (int k = 0; k < nv; k++) jv[r] += d->efc_J[r][k] * search[k]; }
    double alpha = line_search(m, d, jar, jv, g1, g2, m->tolerance * 0.01 * sqrt(snorm) / scale);
    if (alpha == 0) break;
    for (int i = 0; i < nv; i++) { qacc[i] += alpha * search[i]; Ma[i] += alpha * Mv[i]; }
    for (int r = 0; r < nefc; r++) jar[r] += alpha * jv[r];
    double oldcost = cost;
    cost = constraint_update(m, d, jar, qacc, Ma, active, force);
    double gn = 0;
    for (int i = 0; i < nv; i++) {
      double g = Ma[i] - d->qfrc_smooth[i];
      for (int r = 0; r < nefc; r++) g -= d->efc_J[r][i] * force[r];
      gn += g * g;
    }
    double improvement = scale * (oldcost - cost), gradient = scale * sqrt(gn);
    if (improvement < m->tolerance || gradient < m->tolerance) { iter++; break; }
  }
  d->solver_niter = iter;
  om_dbg_newton_calls++; om_dbg_newton_iters += iter; if (iter > om_dbg_newton_max) om_dbg_newton_max = iter;
  memcpy(d->qacc, qacc, sizeof(double) * nv);
  memcpy(d->efc_force, force, sizeof(double) * nefc);
  for (int i = 0; i < nv; i++) {
    double f = 0;
    for (int r = 0; r < nefc; r++) f += d->efc_J[r][i] * force[r];
    d->qfrc_constraint[i] = f;
  }
}

/* ------------------------------------------------------------ pipeline */
static void om_fwd_position(const om_model* m, om_data* d) {
  om_kinematics(m, d);
  om_subspaces(m, d);
  om_crb(m, d);
  om_collision(m, d);
  om_make_constraint(m, d);
  om_make_impedance(m, d);
}
static void om_fwd_velocity(const om_model* m, om_data* d) {
  om_com_vel(m, d);
  om_passive(m, d);
  om_rne_bias(m, d);
  om_reference_constraint(m, d);
}
static void om_fwd_acceleration(const om_model* m, om_data* d) {
  for (int i = 0; i < m->nv; i++) {
    d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
    d->qacc_smooth[i] = d->qfrc_smooth[i];
  }
  chol_solve(m->nv, d->Lm, d->qacc_smooth);
}

static void integrate_pos(const om_model* m, om_data* d, double dt) {
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == OM_JNT_FREE) {
      for (int k = 0; k < 3; k++) d->qpos[qa + k] += dt * d->qvel[da + k];
      double w[3] = {d->qvel[da + 3], d->qvel[da + 4], d->qvel[da + 5]};
      double ang = norm3(w);
      if (ang >= MJS_MINVAL) {
        double ax[3] = {w[0] / ang, w[1] / ang, w[2] / ang}, qr[4], qn[4];
        axisAngle2Quat(qr, ax, ang * dt);
        normQuat(d->qpos + qa + 3);
        mulQuat(qn, d->qpos + qa + 3, qr);
        memcpy(d->qpos + qa + 3, qn, sizeof qn);
      }
      normQuat(d->qpos + qa + 3);
    } else
      d->qpos[qa] += dt * d->qvel[da];
  }
}

static void om_check_state(const om_model* m, om_data* d) {
  for (int i = 0; i < m->nq; i++)
    if (!isfinite(d->qpos[i]) || fabs(d->qpos[i]) > MJS_MAXVAL) d->warning_bad = 1;
  for (int i = 0; i < m->nv; i++) {
    if (!isfinite(d->qvel[i]) || fabs(d->qvel[i]) > MJS_MAXVAL) d->warning_bad = 1;
    if (!isfinite(d->qacc[i]) || fabs(d->qacc[i]) > MJS_MAXVAL) d->warning_bad = 1;
  }
}

void om_step1(const om_model* m, om_data* d) {
  om_fwd_position(m, d);
  om_fwd_velocity(m, d);
}

/* mj_sensorAcc, touch sensor: sum of the normal forces of the contacts that involve a geom of the
 * site's body and whose contact point lies inside the (cylinder) site volume */
static void om_sensor_touch(const om_model* m, om_data* d) {
  d->touch_force = 0;
  if (m->touch_site < 0) return;
  int s = m->touch_site, sb = m->site_body[s];
  for (int c = 0; c < d->ncon; c++) {
    const om_contact* con = &d->contact[c];
    if (con->efc_address < 0) continue;
    if (m->geom_body[con->geom1] != sb && m->geom_body[con->geom2] != sb) continue;
    double rel[3], loc[3];
    for (int k = 0; k < 3; k++) rel[k] = con->pos[k] - d->site_xpos[s][k];
    for (int k = 0; k < 3; k++) loc[k] = d->site_xmat[s][k] * rel[0] + d->site_xmat[s][3 + k] * rel[1] + d->site_xmat[s][6 + k] * rel[2];
    if (loc[0] * loc[0] + loc[1] * loc[1] > m->touch_size[0] * m->touch_size[0] || fabs(loc[2]) > m->touch_size[1]) continue;
    int nrow = (con->dim == 1 || m->cone == OM_CONE_ELLIPTIC) ? 1 : 2 * (con->dim - 1);
    double fn = 0;
    for (int r = 0; r < nrow; r++) fn += d->efc_force[con->efc_address + r]; /* pyramid: normal = sum of edges; elliptic: the normal row */
    d->touch_force += fn;
  }
}

void om_step2(const om_model* m, om_data* d) {
  int nv = m->nv;
  om_actuation(m, d);
  om_fwd_acceleration(m, d);
  om_solve_constraint(m, d);
  om_sensor_touch(m, d);
  om_check_state(m, d);
  memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv);
  double qacc[OM_MAXV];
  if (m->integrator == OM_INT_IMPLICITFAST) {
    /* mj_implicit (fast): (M - dt*dF/dv) qacc = qfrc_smooth + qfrc_constraint; dF/dv from
     * joint damping and the velocity term of unclamped affine actuators (mjd_smooth_vel) */
    double A[OM_MAXV][OM_MAXV], L[OM_MAXV][OM_MAXV];
    for (int i = 0; i < nv; i++) {
      for (int j = 0; j < nv; j++) A[i][j] = d->M[i][j];
      A[i][i] += m->dt * m->dof_damping[i];
      qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
    }
    for (int u = 0; u < m->nu; u++) {
      if (m->act_forcelimited[u] && (d->actuator_force[u] <= m->act_forcerange[u][0] || d->actuator_force[u] >= m->act_forcerange[u][1])) continue;
      if (m->act_trntype[u] == OM_TRN_TENDON) { /* d qfrc / d qvel = moment^T bias_vel moment */
        const double* co = m->tendon_coef[m->act_jnt[u]];
        for (int a = 0; a < nv; a++)
          if (co[a] != 0)
            for (int b = 0; b < nv; b++)
              if (co[b] != 0) A[a][b] -= m->dt * m->act_bias[u][2] * co[a] * co[b];
        continue;
      }
      int da = m->jnt_dofadr[m->act_jnt[u]];
      A[da][da] -= m->dt * m->act_bias[u][2];
    }
    chol_factor(nv, A, L);
    chol_solve(nv, L, qacc);
  } else {
    /* mj_Euler; implicit-in-damping variant when any dof_damping > 0 */
    int damped = 0;
    for (int i = 0; i < nv; i++) if (m->dof_damping[i] > 0) damped = 1;
    if (damped) {
      double A[OM_MAXV][OM_MAXV], L[OM_MAXV][OM_MAXV];
      for (int i = 0; i < nv; i++) {
        for (int j = 0; j < nv; j++) A[i][j] = d->M[i][j];
        A[i][i] += m->dt * m->dof_damping[i];
        qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
      }
      chol_factor(nv, A, L);
      chol_solve(nv, L, qacc);
    } else
      memcpy(qacc, d->qacc, sizeof(double) * nv);
  }
  for (int i = 0; i < nv; i++) d->qvel[i] += m->dt * qacc[i];
  integrate_pos(m, d, m->dt);
  d->time += m->dt;
}

void om_forward(const om_model* m, om_data* d) {
  om_step1(m, d);
  om_actuation(m, d);
  om_fwd_acceleration(m, d);
  om_solve_constraint(m, d);
  om_sensor_touch(m, d);
}

/* dm_control Physics.step() legacy mode (non-RK4): mj_step2 then mj_step1, so that
 * positions/contacts are in sync with the new state */
void om_physics_step(const om_model* m, om_data* d) {
  om_step2(m, d);
  om_step1(m, d);
}

void om_reset_data(const om_model* m, om_data* d) {
  memset(d, 0, sizeof *d);
  memcpy(d->qpos, m->qpos0, sizeof(double) * m->nq);
  for (int b = 0; b < m->nbody; b++)
    if (m->body_mocapid[b] >= 0) {
      memcpy(d->mocap_pos[m->body_mocapid[b]], m->body_pos[b], sizeof(double) * 3);
      memcpy(d->mocap_quat[m->body_mocapid[b]], m->body_quat[b], sizeof(double) * 4);
    }
}

/* mj_setConst: body/dof inverse weights and mean inertia at qpos0 */
void om_set_const(om_model* m) {
  static __thread om_data d;
  om_reset_data(m, &d);
  /* weld ids */
  m->body_weldid[0] = 0;
  for (int b = 1; b < m->nbody; b++) m->body_weldid[b] = (m->body_jntnum[b] == 0) ? m->body_weldid[m->body_parent[b]] : b;
  for (int e = 0; e < m->neq; e++) {
    /* weld relpose from qpos0 when not specified (quat all zero): done by the scene builder */
  }
  om_kinematics(m, &d);
  /* connect: the anchor is given in body1 (data[0:3]); the compiler stores the same world point at qpos0 in body2's frame (data[3:6]) */
  for (int e = 0; e < m->neq; e++)
    if (m->eq_type[e] == OM_EQ_CONNECT) {
      const int b1 = m->eq_body1[e], b2 = m->eq_body2[e];
      double p[3], tmp[3];
      mulMatVec3(tmp, d.xmat[b1], m->eq_data[e]);
      for (int k = 0; k < 3; k++) p[k] = d.xpos[b1][k] + tmp[k] - d.xpos[b2][k];
      for (int k = 0; k < 3; k++) m->eq_data[e][3 + k] = d.xmat[b2][k] * p[0] + d.xmat[b2][3 + k] * p[1] + d.xmat[b2][6 + k] * p[2];
    }
  om_subspaces(m, &d);
  om_crb(m, &d);
  int nv = m->nv;
  double tr = 0;
  for (int i = 0; i < nv; i++) tr += d.M[i][i];
  m->meaninertia = nv > 0 ? tr / nv : 1.0;
  for (int b = 0; b < m->nbody; b++) {
    m->body_invweight0[b][0] = m->body_invweight0[b][1] = 0;
    if (m->body_weldid[b] == 0) continue;
    double jt[3][OM_MAXV], jr[3][OM_MAXV];
    om_jac(m, &d, b, d.xipos[b], jt, jr);
    double tsum = 0, rsum = 0;
    for (int k = 0; k < 3; k++) {
      double x[OM_MAXV];
      memcpy(x, jt[k], sizeof(double) * nv);
      chol_solve(nv, d.Lm, x);
      for (int i = 0; i < nv; i++) tsum += jt[k][i] * x[i];
      memcpy(x, jr[k], sizeof(double) * nv);
      chol_solve(nv, d.Lm, x);
      for (int i = 0; i < nv; i++) rsum += jr[k][i] * x[i];
    }
    m->body_invweight0[b][0] = fmax(MJS_MINVAL, tsum / 3);
    m->body_invweight0[b][1] = fmax(MJS_MINVAL, rsum / 3);
  }
  for (int j = 0; j < m->njnt; j++) {
    int da = m->jnt_dofadr[j];
    int nd = m->jnt_type[j] == OM_JNT_FREE ? 6 : 1;
    double diag[6];
    for (int k = 0; k < nd; k++) {
      double x[OM_MAXV];
      memset(x, 0, sizeof x);
      x[da + k] = 1;
      chol_solve(nv, d.Lm, x);
      diag[k] = x[da + k];
    }
    if (nd == 6) {
      double t = (diag[0] + diag[1] + diag[2]) / 3, r = (diag[3] + diag[4] + diag[5]) / 3;
      for (int k = 0; k < 3; k++) { m->dof_invweight0[da + k] = t; m->dof_invweight0[da + 3 + k] = r; }
    } else
      m->dof_invweight0[da] = diag[0];
  }
}
