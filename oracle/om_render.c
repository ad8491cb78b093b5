/* om_render.c — ORACLE (test infrastructure): CPU restatement of the scene-camera render.
 *
 * The reference obtains images from MuJoCo's OpenGL renderer through
 * Camera.get_rgb_image (entities/camera.py:94-103); that pipeline is third-party and cannot be
 * reproduced here ("rendered images cannot match OpenGL output"). What this file
 * pins is the build's OWN image definition — pinhole camera in MuJoCo's convention (looks along
 * local -z, fovy vertical), the scene's analytic primitives from
 * mjcf/walled_pointmass_arena.xml:12-19, entities/pointmass.py:52-55, point_reach.py:91-93,
 * Blinn-Phong with MuJoCo's default headlight / light / material parameters — so that the HIP
 * kernel can be checked bit-for-bit (float32, only + - * / sqrt, no FMA contraction).
 * PARITY UNPINNED against the reference's pixels.
 */
#include <math.h>
#include <string.h>

#include "../include/mjs_scene_spec.h"
#include "mjs_oracle.h"
#include "../include/mjs_block_hulls.h"

typedef struct { float x, y, z; } v3;
static v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 vmul(float s, v3 a) { return V(s * a.x, s * a.y, s * a.z); }
static float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 vnorm(v3 a) { float inv = 1.0f / sqrtf(vdot(a, a)); return V(a.x * inv, a.y * inv, a.z * inv); }
static float pow2k(float x, int k) { for (int i = 0; i < k; i++) x = x * x; return x; }

typedef struct { float t; v3 n, rgb; } surf;

static void rect_z(v3 o, v3 d, float z0, float hx, float hy, v3 rgb, int checker, surf* s) {
  if (d.z == 0.0f) return;
  float t = (z0 - o.z) / d.z;
  if (!(t > 0.0f) || !(t < s->t)) return;
  float x = o.x + t * d.x, y = o.y + t * d.y;
  if (x < -hx || x > hx || y < -hy || y > hy) return;
  s->t = t; s->n = V(0, 0, 1);
  if (checker) {
    int cx = x >= 0.0f, cy = y >= 0.0f;
    s->rgb = ((cx + cy) & 1) ? V(MJS_PM_GRID_RGB2[0], MJS_PM_GRID_RGB2[1], MJS_PM_GRID_RGB2[2]) : V(MJS_PM_GRID_RGB1[0], MJS_PM_GRID_RGB1[1], MJS_PM_GRID_RGB1[2]);
  } else s->rgb = rgb;
}
static void wall(v3 o, v3 d, int axis, float c, float sign, float half, float z0, float z1, v3 rgb, surf* s) {
  float oa = axis == 0 ? o.x : o.y, da = axis == 0 ? d.x : d.y;
  if (da == 0.0f) return;
  float t = (c - oa) / da;
  if (!(t > 0.0f) || !(t < s->t)) return;
  float other = axis == 0 ? o.y + t * d.y : o.x + t * d.x;
  float z = o.z + t * d.z;
  if (other < -half || other > half || z < z0 || z > z1) return;
  s->t = t; s->n = axis == 0 ? V(sign, 0, 0) : V(0, sign, 0); s->rgb = rgb;
}
static void sphere(v3 o, v3 d, v3 c, float r, v3 rgb, surf* s) {
  v3 oc = vsub(o, c);
  float b = vdot(oc, d), cc = vdot(oc, oc) - r * r;
  float disc = b * b - cc;
  if (disc < 0.0f) return;
  float t = -b - sqrtf(disc);
  if (!(t > 0.0f) || !(t < s->t)) return;
  s->t = t;
  s->n = vnorm(vsub(vadd(o, vmul(t, d)), c));
  s->rgb = rgb;
}
static void cube(v3 o, v3 d, v3 c, float h, v3 rgb, surf* s) {
  float tmin = 0.0f, tmax = s->t, sg = 0.0f;
  int ax = -1;
  float oo[3] = {o.x - c.x, o.y - c.y, o.z - c.z}, dd[3] = {d.x, d.y, d.z};
  for (int k = 0; k < 3; k++) {
    if (dd[k] == 0.0f) { if (oo[k] < -h || oo[k] > h) return; continue; }
    float t1 = (-h - oo[k]) / dd[k], t2 = (h - oo[k]) / dd[k], sgn = -1.0f;
    if (t1 > t2) { float tmp = t1; t1 = t2; t2 = tmp; sgn = 1.0f; }
    if (t1 > tmin) { tmin = t1; ax = k; sg = sgn; }
    if (t2 < tmax) tmax = t2;
    if (tmin > tmax) return;
  }
  if (ax < 0 || !(tmin > 0.0f) || !(tmin < s->t)) return;
  s->t = tmin;
  s->n = ax == 0 ? V(sg, 0, 0) : ax == 1 ? V(0, sg, 0) : V(0, 0, sg);
  s->rgb = rgb;
}
static v3 shade(v3 p, v3 n, v3 eye, v3 rgb, const float (*lights)[3], int nlight) {
  v3 v = vnorm(vsub(eye, p));
  if (vdot(n, v) < 0.0f) n = vmul(-1.0f, n);
  float diff = 0.0f, spec = 0.0f;
  float ndl = vdot(n, v);
  if (ndl > 0.0f) {
    diff = diff + MJS_HEADLIGHT_DIFFUSE * ndl;
    spec = spec + MJS_HEADLIGHT_SPECULAR * pow2k(ndl, MJS_MATERIAL_SHININESS_POW2);
  }
  for (int k = 0; k < nlight; k++) {
    v3 lv = vsub(V(lights[k][0], lights[k][1], lights[k][2]), p);
    /* outside the 45 degree cone (cos^2 = 1/2), decided on the un-normalised vector */
    if (lv.z <= 0.0f || lv.z * lv.z < MJS_LIGHT_CUTOFF_COS2 * vdot(lv, lv)) continue;
    v3 l = vnorm(lv);
    float spotcos = l.z;
    float spot = pow2k(spotcos, 3) * pow2k(spotcos, 1);
    float nl = vdot(n, l);
    if (nl > 0.0f) {
      diff = diff + MJS_LIGHT_DIFFUSE * nl * spot;
      float ndh = vdot(n, vnorm(vadd(l, v)));
      if (ndh > 0.0f) spec = spec + MJS_LIGHT_SPECULAR * pow2k(ndh, MJS_MATERIAL_SHININESS_POW2) * spot;
    }
  }
  float k = MJS_HEADLIGHT_AMBIENT + diff, sp = MJS_MATERIAL_SPECULAR * spec;
  return V(rgb.x * k + sp, rgb.y * k + sp, rgb.z * k + sp);
}
static uint8_t to_u8(float c) {
  c = c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c);
  return (uint8_t)(int)(c * 255.0f + 0.5f);
}

/* scene camera image of a Pointmass env: out uint8 [H, W, 3] */
void om_render_pointmass(const om_env* e, int H, int W, uint8_t* out) {
  const double* q = MJS_PM_CAM_QUAT;
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
  double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                 2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
  float right[3], up[3], back[3];
  for (int k = 0; k < 3; k++) { right[k] = (float)R[3 * k]; up[k] = (float)R[3 * k + 1]; back[k] = (float)R[3 * k + 2]; }
  float tan_half = (float)tan(MJS_PM_CAM_FOVY * 3.14159265358979323846 / 360.0);
  v3 eye = V((float)MJS_PM_CAM_POS[0], (float)MJS_PM_CAM_POS[1], (float)MJS_PM_CAM_POS[2]);
  float qx = (float)e->d.qpos[0], qy = (float)e->d.qpos[1], tx = (float)e->target_pos[0], ty = (float)e->target_pos[1];
  float mx = (float)e->d.mocap_pos[0][0], my = (float)e->d.mocap_pos[0][1];
  v3 wallrgb = V(MJS_PM_WALL_RGB[0], MJS_PM_WALL_RGB[1], MJS_PM_WALL_RGB[2]);
  float hi = (float)MJS_PM_ARENA_HI, wz = (float)MJS_PM_WALL_Z;
  float aspect = (float)W / (float)H;
  for (int row = 0; row < H; row++)
    for (int col = 0; col < W; col++) {
      float px = (2.0f * ((float)col + 0.5f) / (float)W - 1.0f) * tan_half * aspect;
      float py = (1.0f - 2.0f * ((float)row + 0.5f) / (float)H) * tan_half;
      v3 d = vnorm(V(px * right[0] + py * up[0] - back[0], px * right[1] + py * up[1] - back[1], px * right[2] + py * up[2] - back[2]));
      surf s;
      s.t = INFINITY; s.n = V(0, 0, 1); s.rgb = V(0, 0, 0);
      rect_z(eye, d, 0.0f, hi, hi, V(0, 0, 0), 1, &s);
      wall(eye, d, 0, -hi, 1.0f, hi, 0.0f, 2.0f * wz, wallrgb, &s);
      wall(eye, d, 1, -hi, 1.0f, hi, 0.0f, 2.0f * wz, wallrgb, &s);
      wall(eye, d, 0, hi, -1.0f, hi, 0.0f, 2.0f * wz, wallrgb, &s);
      wall(eye, d, 1, hi, -1.0f, hi, 0.0f, 2.0f * wz, wallrgb, &s);
      cube(eye, d, V(tx, ty, (float)(MJS_PM_RADIUS / 2)), MJS_PM_TARGET_HALF, V(MJS_PM_TARGET_RGB[0], MJS_PM_TARGET_RGB[1], MJS_PM_TARGET_RGB[2]), &s);
      sphere(eye, d, V(mx, my, 0.0f), MJS_PM_MOCAP_SITE_RADIUS, V(MJS_SITE_DEFAULT_RGB[0], MJS_SITE_DEFAULT_RGB[1], MJS_SITE_DEFAULT_RGB[2]), &s);
      v3 c = V(0, 0, 0);
      if (s.t < INFINITY) c = shade(vadd(eye, vmul(s.t, d)), s.n, eye, s.rgb, MJS_PM_LIGHT_POS, 2);
      surf b;
      b.t = s.t; b.n = V(0, 0, 1); b.rgb = V(0, 0, 0);
      sphere(eye, d, V(qx, qy, (float)MJS_PM_RADIUS), (float)MJS_PM_RADIUS, V(MJS_PM_SPHERE_RGBA[0], MJS_PM_SPHERE_RGBA[1], MJS_PM_SPHERE_RGBA[2]), &b);
      if (b.t < s.t) {
        v3 sc = shade(vadd(eye, vmul(b.t, d)), b.n, eye, b.rgb, MJS_PM_LIGHT_POS, 2);
        float a = MJS_PM_SPHERE_RGBA[3];
        c = V(a * sc.x + (1.0f - a) * c.x, a * sc.y + (1.0f - a) * c.y, a * sc.z + (1.0f - a) * c.z);
      }
      uint8_t* o = out + ((size_t)row * W + col) * 3;
      o[0] = to_u8(c.x); o[1] = to_u8(c.y); o[2] = to_u8(c.z);
    }
}

/* ------------------------------------------------------------------ robot scenes */
static void capsule(v3 o, v3 d, v3 pa, v3 pb, float r, v3 rgb, surf* s) {
  v3 ba = vsub(pb, pa), oa = vsub(o, pa);
  float baba = vdot(ba, ba), bard = vdot(ba, d), baoa = vdot(ba, oa), rdoa = vdot(d, oa), oaoa = vdot(oa, oa);
  float a = baba - bard * bard, b = baba * rdoa - baoa * bard, c = baba * oaoa - baoa * baoa - r * r * baba;
  float h = b * b - a * c;
  if (h < 0.0f) return;
  if (a > 0.0f) {
    float t = (-b - sqrtf(h)) / a;
    float y = baoa + t * bard;
    if (y > 0.0f && y < baba) {
      if (!(t > 0.0f) || !(t < s->t)) return;
      s->t = t;
      s->n = vmul(1.0f / r, vsub(vadd(oa, vmul(t, d)), vmul(y / baba, ba)));
      s->rgb = rgb;
      return;
    }
    v3 oc = y <= 0.0f ? oa : vsub(o, pb);
    float bb = vdot(d, oc), cc = vdot(oc, oc) - r * r;
    float hh = bb * bb - cc;
    if (hh > 0.0f) {
      float tc = -bb - sqrtf(hh);
      if (!(tc > 0.0f) || !(tc < s->t)) return;
      s->t = tc;
      s->n = vmul(1.0f / r, vadd(oc, vmul(tc, d)));
      s->rgb = rgb;
    }
  }
}
static void cylinder(v3 o, v3 d, v3 pa, v3 pb, float r, v3 rgb, surf* s) {
  v3 ba = vsub(pb, pa), oa = vsub(o, pa);
  float baba = vdot(ba, ba), bard = vdot(ba, d), baoa = vdot(ba, oa);
  float k2 = baba - bard * bard, k1 = baba * vdot(oa, d) - baoa * bard, k0 = baba * vdot(oa, oa) - baoa * baoa - r * r * baba;
  float h = k1 * k1 - k2 * k0;
  if (h < 0.0f) return;
  if (k2 > 0.0f) {
    float t = (-k1 - sqrtf(h)) / k2;
    float y = baoa + t * bard;
    if (y > 0.0f && y < baba) {
      if (!(t > 0.0f) || !(t < s->t)) return;
      s->t = t;
      s->n = vmul(1.0f / r, vsub(vadd(oa, vmul(t, d)), vmul(y / baba, ba)));
      s->rgb = rgb;
      return;
    }
  }
  if (bard == 0.0f) return;
  float tc = ((bard < 0.0f ? baba : 0.0f) - baoa) / bard;
  if (!(tc > 0.0f) || !(tc < s->t)) return;
  v3 q = vadd(oa, vmul(tc, d));
  float yc = bard < 0.0f ? baba : 0.0f;
  v3 radial = vsub(q, vmul(yc / baba, ba));
  if (vdot(radial, radial) > r * r) return;
  s->t = tc;
  float inv = 1.0f / sqrtf(baba);
  s->n = vmul(bard < 0.0f ? inv : -inv, ba);
  s->rgb = rgb;
}
static void obb(v3 o, v3 d, v3 c, v3 u, v3 v, v3 half, v3 rgb, surf* s) {
  v3 w = V(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
  v3 oc = vsub(o, c);
  float oo[3] = {vdot(oc, u), vdot(oc, v), vdot(oc, w)}, dd[3] = {vdot(d, u), vdot(d, v), vdot(d, w)}, hh[3] = {half.x, half.y, half.z};
  float tmin = 0.0f, tmax = s->t, sg = 0.0f;
  int ax = -1;
  for (int k = 0; k < 3; k++) {
    if (dd[k] == 0.0f) { if (oo[k] < -hh[k] || oo[k] > hh[k]) return; continue; }
    float t1 = (-hh[k] - oo[k]) / dd[k], t2 = (hh[k] - oo[k]) / dd[k], sgn = -1.0f;
    if (t1 > t2) { float tmp = t1; t1 = t2; t2 = tmp; sgn = 1.0f; }
    if (t1 > tmin) { tmin = t1; ax = k; sg = sgn; }
    if (t2 < tmax) tmax = t2;
    if (tmin > tmax) return;
  }
  if (ax < 0 || !(tmin > 0.0f) || !(tmin < s->t)) return;
  s->t = tmin;
  s->n = vmul(sg, ax == 0 ? u : ax == 1 ? v : w);
  s->rgb = rgb;
}
static v3 Vd(const double* p) { return V((float)p[0], (float)p[1], (float)p[2]); }

static void quat_to_mat(const double* q, double* R) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}

/* Robot scenes (arm drawn by its collision proxies + stand-ins): Robot-Reach scene camera
 * (robot_reach.py:52), Button-Push scene camera (robot_push_button.py:51-52,87-89) and wrist camera
 * (robot_push_button.py:53-54,90-96; camera = 1). out uint8 [H, W, 3] */
static void render_robot_scene(const om_env* e, int camera, int H, int W, uint8_t* out) {
  const om_model* m = &e->m;
  const om_data* dd = &e->d;
  const int button = e->cfg.task == OM_TASK_BUTTON_PUSH, push = e->cfg.task == OM_TASK_PLANAR_PUSH;
  /* flange frame (site 0) */
  const double* sp = dd->site_xpos[0];
  const double* sm = dd->site_xmat[0];
  double fx[3], fy[3], fz[3];
  for (int k = 0; k < 3; k++) { fx[k] = sm[3 * k]; fy[k] = sm[3 * k + 1]; fz[k] = sm[3 * k + 2]; }
  /* wrist camera pose */
  double Rc[9], wpos[3], wright[3], wup[3], wback[3];
  quat_to_mat(MJS_WCAM_QUAT, Rc);
  for (int k = 0; k < 3; k++) {
    wpos[k] = ((sp[k] + MJS_WCAM_POS[0] * fx[k]) + MJS_WCAM_POS[1] * fy[k]) + MJS_WCAM_POS[2] * fz[k];
    wright[k] = (Rc[0] * fx[k] + Rc[3] * fy[k]) + Rc[6] * fz[k];
    wup[k] = (Rc[1] * fx[k] + Rc[4] * fy[k]) + Rc[7] * fz[k];
    wback[k] = (Rc[2] * fx[k] + Rc[5] * fy[k]) + Rc[8] * fz[k];
  }
  /* camera used for this image */
  double R[9];
  quat_to_mat(button ? MJS_BP_CAM_QUAT : MJS_RR_CAM_QUAT, R);
  const double* cpos = button ? MJS_BP_CAM_POS : MJS_RR_CAM_POS;
  float right[3], up[3], back[3];
  v3 eye;
  double fovy;
  if (camera == 1) {
    for (int k = 0; k < 3; k++) { right[k] = (float)wright[k]; up[k] = (float)wup[k]; back[k] = (float)wback[k]; }
    eye = Vd(wpos);
    fovy = MJS_WCAM_FOVY;
  } else {
    for (int k = 0; k < 3; k++) { right[k] = (float)R[3 * k]; up[k] = (float)R[3 * k + 1]; back[k] = (float)R[3 * k + 2]; }
    eye = Vd(cpos);
    fovy = button ? MJS_BP_CAM_FOVY : MJS_RR_CAM_FOVY;
  }
  float tan_half = (float)tan(fovy * 3.14159265358979323846 / 360.0);
  /* primitive list, float32 */
  v3 cap_a[MJS_UR_NCOLGEOM], cap_b[MJS_UR_NCOLGEOM];
  for (int g = 0; g < MJS_UR_NCOLGEOM; g++) {
    const double* gp = dd->geom_xpos[1 + g];
    const double* gm = dd->geom_xmat[1 + g];
    double half = m->geom_size[1 + g][1], a[3], b[3];
    for (int k = 0; k < 3; k++) { a[k] = gp[k] + -half * gm[3 * k + 2]; b[k] = gp[k] + half * gm[3 * k + 2]; }
    cap_a[g] = Vd(a); cap_b[g] = Vd(b);
  }
  double bc[3];
  for (int k = 0; k < 3; k++) bc[k] = sp[k] + (double)MJS_G2F85_STANDIN_HALF[2] * fz[k];
  v3 box_c = Vd(bc), box_u = Vd(fx), box_v = Vd(fy);
  v3 tgt = Vd(e->target_pos);
  /* Button-Push extras */
  double Rs[9], su[3], sv[3], swc[3], ba[3], bb[3];
  quat_to_mat(MJS_BP_CAM_QUAT, Rs);
  for (int k = 0; k < 3; k++) { su[k] = Rs[3 * k]; sv[k] = Rs[3 * k + 1]; }
  const double* swp = button ? m->body_pos[m->nbody - 1] : MJS_RR_TARGET_DEFAULT_POS;
  swc[0] = swp[0]; swc[1] = swp[1]; swc[2] = swp[2] + MJS_SW_BOX_HALF;
  ba[0] = bb[0] = swp[0]; ba[1] = bb[1] = swp[1];
  ba[2] = swp[2] + MJS_SW_BUTTON_Z - MJS_SW_BUTTON_HALF; bb[2] = swp[2] + MJS_SW_BUTTON_Z + MJS_SW_BUTTON_HALF;
  const v3 cam_half = V((float)MJS_CAM_BOX_HALF[0], (float)MJS_CAM_BOX_HALF[1], (float)MJS_CAM_BOX_HALF[2]);
  const v3 cam_rgb = V(MJS_CAM_BODY_RGB[0], MJS_CAM_BODY_RGB[1], MJS_CAM_BODY_RGB[2]);
  const float* brgb = e->switch_active ? MJS_SW_BUTTON_RGB_ON : MJS_SW_BUTTON_RGB_OFF;
  float aspect = (float)W / (float)H;
  for (int row = 0; row < H; row++)
    for (int col = 0; col < W; col++) {
      float px = (2.0f * ((float)col + 0.5f) / (float)W - 1.0f) * tan_half * aspect;
      float py = (1.0f - 2.0f * ((float)row + 0.5f) / (float)H) * tan_half;
      v3 d = vnorm(V(px * right[0] + py * up[0] - back[0], px * right[1] + py * up[1] - back[1], px * right[2] + py * up[2] - back[2]));
      surf s;
      s.t = INFINITY; s.n = V(0, 0, 1); s.rgb = V(0, 0, 0);
      rect_z(eye, d, 0.0f, (float)MJS_ROBOT_ARENA_HALF, (float)MJS_ROBOT_ARENA_HALF, V(MJS_RR_FLOOR_RGB[0], MJS_RR_FLOOR_RGB[1], MJS_RR_FLOOR_RGB[2]), 0, &s);
      for (int g = 0; g < MJS_UR_NCOLGEOM; g++) {
        const float* c = MJS_UR_COL_IS_JOINT[g] ? MJS_UR_URBLUE : MJS_UR_LINKGRAY;
        if (g == 3) /* test order = the GPU primitive list: shoulder/upper arm, base stand-in, forearm/wrist, gripper */
          cylinder(eye, d, V(0, 0, 0), V(0, 0, (float)(2.0 * MJS_UR_BASE_STANDIN[1])), MJS_UR_BASE_STANDIN[0], V(MJS_UR_JOINTGRAY[0], MJS_UR_JOINTGRAY[1], MJS_UR_JOINTGRAY[2]), &s);
        if (MJS_UR_COL_TYPE[g] == 3) capsule(eye, d, cap_a[g], cap_b[g], (float)MJS_UR_COL_SIZE[g][0], V(c[0], c[1], c[2]), &s);
        else cylinder(eye, d, cap_a[g], cap_b[g], (float)MJS_UR_COL_SIZE[g][0], V(c[0], c[1], c[2]), &s);
      }
      if (push) {
        /* CylinderEEF (geom after the arm proxies), target site disc, blocks (box geoms of the free bodies) */
        const int ge = 1 + MJS_UR_NCOLGEOM;
        const double* gp = dd->geom_xpos[ge];
        const double* gm = dd->geom_xmat[ge];
        double a[3], b[3], ta[3], tb[3];
        for (int k = 0; k < 3; k++) { a[k] = gp[k] + -MJS_CYL_HALFLEN * gm[3 * k + 2]; b[k] = gp[k] + MJS_CYL_HALFLEN * gm[3 * k + 2]; }
        cylinder(eye, d, Vd(a), Vd(b), (float)MJS_CYL_RADIUS, V(MJS_CYL_RGB[0], MJS_CYL_RGB[1], MJS_CYL_RGB[2]), &s);
        for (int k = 0; k < 3; k++) { ta[k] = e->target_pos[k]; tb[k] = e->target_pos[k]; }
        ta[2] = e->target_pos[2] - (double)MJS_PP_TARGET_HALF_HEIGHT; tb[2] = e->target_pos[2] + (double)MJS_PP_TARGET_HALF_HEIGHT;
        cylinder(eye, d, Vd(ta), Vd(tb), (float)MJS_PP_TARGET_RADIUS, V(MJS_PP_TARGET_RGB[0], MJS_PP_TARGET_RGB[1], MJS_PP_TARGET_RGB[2]), &s);
        for (int i = 0; i < e->cfg.n_objects; i++) {
          const double* bp = dd->geom_xpos[ge + 1 + i];
          const double* bm = dd->geom_xmat[ge + 1 + i];
          double u[3] = {bm[0], bm[3], bm[6]}, w[3] = {bm[1], bm[4], bm[7]};
          if (e->cfg.block_shape == OM_BLOCKS_MESH) {
            /* a mesh block is drawn as the bounding box of its hull (scaled), in its sampled colour (D-6: own ray caster) */
            const int cat = e->block_cat[i], body = e->m.geom_body[ge + 1 + i];
            const double sc = e->block_scale[i];
            double ctr[3], bc[3];
            for (int k = 0; k < 3; k++) bc[k] = 0.5 * (MJS_HULL_BOX_LO[cat][k] + MJS_HULL_BOX_HI[cat][k]) * sc;
            for (int k = 0; k < 3; k++) ctr[k] = dd->xpos[body][k] + (bm[3 * k] * bc[0] + bm[3 * k + 1] * bc[1] + bm[3 * k + 2] * bc[2]);
            obb(eye, d, Vd(ctr), Vd(u), Vd(w), V((float)(0.5 * (MJS_HULL_BOX_HI[cat][0] - MJS_HULL_BOX_LO[cat][0]) * sc), (float)(0.5 * (MJS_HULL_BOX_HI[cat][1] - MJS_HULL_BOX_LO[cat][1]) * sc),
                                                    (float)(0.5 * (MJS_HULL_BOX_HI[cat][2] - MJS_HULL_BOX_LO[cat][2]) * sc)),
                V(MJS_BLOCK_COLORS[e->block_color[i]][0], MJS_BLOCK_COLORS[e->block_color[i]][1], MJS_BLOCK_COLORS[e->block_color[i]][2]), &s);
          } else
          obb(eye, d, Vd(bp), Vd(u), Vd(w), V((float)MJS_BLOCK_HALF[0], (float)MJS_BLOCK_HALF[1], (float)MJS_BLOCK_HALF[2]),
              V(MJS_BLOCK_RGB[i][0], MJS_BLOCK_RGB[i][1], MJS_BLOCK_RGB[i][2]), &s);
        }
      } else
      obb(eye, d, box_c, box_u, box_v, V(MJS_G2F85_STANDIN_HALF[0], MJS_G2F85_STANDIN_HALF[1], MJS_G2F85_STANDIN_HALF[2]), V(MJS_UR_BLACK[0], MJS_UR_BLACK[1], MJS_UR_BLACK[2]), &s);
      if (push) {
      } else if (!button) {
        sphere(eye, d, tgt, MJS_RR_TARGET_RADIUS, V(MJS_RR_TARGET_RGB[0], MJS_RR_TARGET_RGB[1], MJS_RR_TARGET_RGB[2]), &s);
      } else {
        obb(eye, d, Vd(wpos), Vd(wright), Vd(wup), cam_half, cam_rgb, &s);
        sphere(eye, d, Vd(wpos), (float)MJS_CAM_SPHERE_RADIUS, cam_rgb, &s);
        obb(eye, d, Vd(MJS_BP_CAM_POS), Vd(su), Vd(sv), cam_half, cam_rgb, &s);
        sphere(eye, d, Vd(MJS_BP_CAM_POS), (float)MJS_CAM_SPHERE_RADIUS, cam_rgb, &s);
        obb(eye, d, Vd(swc), V(1, 0, 0), V(0, 1, 0), V((float)MJS_SW_BOX_HALF, (float)MJS_SW_BOX_HALF, (float)MJS_SW_BOX_HALF), V(MJS_SW_BOX_RGB[0], MJS_SW_BOX_RGB[1], MJS_SW_BOX_RGB[2]), &s);
        cylinder(eye, d, Vd(ba), Vd(bb), (float)MJS_SW_BUTTON_RADIUS, V(brgb[0], brgb[1], brgb[2]), &s);
      }
      v3 c = V(0, 0, 0);
      if (s.t < INFINITY) c = shade(vadd(eye, vmul(s.t, d)), s.n, eye, s.rgb, MJS_RR_LIGHT_POS, 6);
      uint8_t* o = out + ((size_t)row * W + col) * 3;
      o[0] = to_u8(c.x); o[1] = to_u8(c.y); o[2] = to_u8(c.z);
    }
}
void om_render_robot(const om_env* e, int H, int W, uint8_t* out) { render_robot_scene(e, 0, H, W, out); }
void om_render_camera(const om_env* e, int camera, int H, int W, uint8_t* out) {
  if (e->cfg.task == OM_TASK_POINTMASS) om_render_pointmass(e, H, W, out);
  else render_robot_scene(e, camera, H, W, out);
}
