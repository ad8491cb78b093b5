// Host build of the GENERATED UR5e dynamics header (mujoco_sim_amd/csrc/mjs_ur5e_dyn_gen.h) for the CPU test
// tests/test_generated_dynamics.py: the same straight-line code the kernels run, compiled by g++.
#include <cmath>
#define MJS_DEV static inline
#define MJS_SCHED_PIN ((void)0)
#define MJS_GEN_RCP(x) (1.0 / (x))
#include "../../mujoco_sim_amd/csrc/mjs_ur5e_dyn_gen.h"

extern "C" {
// variant 0 = "ur5e" (Robot-Reach), 1 = "ur5e_bp" (Button-Push); M_out = 6x6 full (no armature), bias_out = 6
void gen_dynamics(int variant, const double* q, const double* v, double* M_out, double* bias_out) {
  double c[6], s[6], M[21], b[6];
  for (int j = 0; j < 6; j++) { c[j] = std::cos(q[j]); s[j] = std::sin(q[j]); }
  if (variant == 0) { ur5e_M_gen(c, s, M); ur5e_bias_gen(c, s, v, b); }
  else { ur5e_bp_M_gen(c, s, M); ur5e_bp_bias_gen(c, s, v, b); }
  for (int i = 0; i < 6; i++) {
    for (int j = 0; j <= i; j++) M_out[6 * i + j] = M_out[6 * j + i] = M[i * (i + 1) / 2 + j];
    bias_out[i] = b[i];
  }
}
// x = (M(q) + diag(dd))^-1 b through the generated factor / inverse block of the Robot-Reach variant (ur5e_MW_gen)
void gen_solve(const double* q, const double* dd, const double* b, double* x) {
  double c[6], s[6], W[15], Dinv[6], z[6];
  for (int j = 0; j < 6; j++) { c[j] = std::cos(q[j]); s[j] = std::sin(q[j]); }
  ur5e_MW_gen(c, s, dd, W, Dinv);
  for (int i = 0; i < 6; i++) {  // rr::apply_inverse: x = V^T (Dinv .* (V b)), V(i, j) = W[j (j - 1) / 2 + i]
    double y = b[i];
    for (int j = i + 1; j < 6; j++) y += W[j * (j - 1) / 2 + i] * b[j];
    z[i] = y * Dinv[i];
  }
  for (int i = 0; i < 6; i++) {
    double v = z[i];
    for (int k = 0; k < i; k++) v += W[i * (i - 1) / 2 + k] * z[k];
    x[i] = v;
  }
}
// mj_setConst constants emitted next to the code: dof_invweight0[6], meaninertia, eef body invweight0[2]
void gen_constants(int variant, double* out) {
  for (int j = 0; j < 6; j++) out[j] = variant == 0 ? UR5E_DOF_INVWEIGHT0[j] : UR5E_BP_DOF_INVWEIGHT0[j];
  out[6] = variant == 0 ? UR5E_MEANINERTIA : UR5E_BP_MEANINERTIA;
  out[7] = variant == 0 ? UR5E_EEF_BODY_INVWEIGHT0[0] : UR5E_BP_EEF_BODY_INVWEIGHT0[0];
  out[8] = variant == 0 ? UR5E_EEF_BODY_INVWEIGHT0[1] : UR5E_BP_EEF_BODY_INVWEIGHT0[1];
}
// body_invweight0 (translation) of the arm's own link bodies 0..6, emitted for the arm-floor contact rows: variant 0 Reach,
// 1 Button-Push, 2 Planar-Push
void gen_link_invweights(int variant, double* out7) {
  for (int b = 0; b < 7; b++) out7[b] = variant == 0 ? UR5E_LINK_BODY_INVWEIGHT0[b] : variant == 1 ? UR5E_BP_LINK_BODY_INVWEIGHT0[b] : UR5E_PP_LINK_BODY_INVWEIGHT0[b];
}
}
