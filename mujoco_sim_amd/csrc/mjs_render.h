// mjs_render.h — fixed-camera RGB render kernels.
//
// Replaces Camera.get_rgb_image -> physics.render(height, width, camera_id) (reference:
// entities/camera.py:94-103; observable spec uint8 (H, W, 3), camera.py:146-150) for the scene
// cameras of the tasks. The reference renders with MuJoCo's OpenGL pipeline, which cannot be
// reproduced pixel-for-pixel (deviation D-6): this is a ray caster over the scene's analytic
// primitives with a Blinn-Phong model fed by MuJoCo's default light/material parameters.
// One thread per pixel, one workgroup row per env; output [N, H, W, 3] uint8 is the HBM-write
// stream that bounds the visual configs (BASELINE config 5: 12 KB per 64x64 image).
//
// Arithmetic is float32 restricted to + - * / sqrt and comparisons with FMA contraction disabled,
// so the image is bit-identical to the independent CPU restatement in oracle/om_render.c.
#pragma once
#include "mjs_kernel_common.h"
#include "mjs_pointmass.h"
#include "mjs_reach.h"
#include "mjs_button.h"
#include "mjs_push.h"

namespace rend {

#pragma clang fp contract(off)

struct Cam {
  float pos[3];
  float right[3], up[3], back[3];  // camera local +x, +y, +z axes in world coordinates
  float tan_half;                  // tan(fovy / 2)
};

struct RenderParams {
  int N, H, W;
  const double* state;  // [state_dim][N]
  Cam cam;              // fixed camera, or (env_cams != nullptr) only tan_half is used
  const float* env_cams;  // [N][12] per-env camera pose (pos, right, up, back): body-mounted cameras
  int nprim;              // robot scenes: primitives per env
  uint8_t* out;  // [N, H, W, 3]
};

struct F3 {
  float x, y, z;
};
MJS_DEV F3 f3(float x, float y, float z) { return F3{x, y, z}; }
MJS_DEV F3 add(F3 a, F3 b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }
MJS_DEV F3 sub(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
MJS_DEV F3 mul(float s, F3 a) { return F3{s * a.x, s * a.y, s * a.z}; }
MJS_DEV float dotf(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
MJS_DEV F3 normalize(F3 a) {
  float inv = 1.0f / sqrtf(dotf(a, a));  // one IEEE division, then multiplies (same on the CPU side)
  return F3{a.x * inv, a.y * inv, a.z * inv};
}
MJS_DEV float pow_pow2(float x, int k) {  // x^(2^k)
  for (int i = 0; i < k; i++) x = x * x;
  return x;
}

struct Surf {
  float t;  // ray parameter of the nearest hit so far (or +inf)
  F3 n;     // surface normal
  F3 rgb;
};

MJS_DEV void hit_rect_z(F3 o, F3 d, float z0, float hx, float hy, F3 rgb, bool checker, Surf& s) {
  // horizontal rectangle centred on the z axis at height z0
  if (d.z == 0.0f) return;
  float t = (z0 - o.z) / d.z;
  if (!(t > 0.0f) || !(t < s.t)) return;
  float x = o.x + t * d.x, y = o.y + t * d.y;
  if (x < -hx || x > hx || y < -hy || y > hy) return;
  s.t = t;
  s.n = f3(0, 0, 1);
  if (checker) {
    // builtin 2x2 checker, one repeat per unit length (texuniform), centred on the plane
    int cx = x >= 0.0f ? 1 : 0;
    int cy = y >= 0.0f ? 1 : 0;
    s.rgb = ((cx + cy) & 1) ? f3(MJS_PM_GRID_RGB2[0], MJS_PM_GRID_RGB2[1], MJS_PM_GRID_RGB2[2])
                            : f3(MJS_PM_GRID_RGB1[0], MJS_PM_GRID_RGB1[1], MJS_PM_GRID_RGB1[2]);
  } else
    s.rgb = rgb;
}
// vertical wall rectangle: plane {axis coordinate == c} with normal sign*e_axis, extent |other| <= half, z in [z0, z1]
MJS_DEV void hit_wall(F3 o, F3 d, int axis, float c, float sign, float half, float z0, float z1, F3 rgb, Surf& s) {
  float oa = axis == 0 ? o.x : o.y, da = axis == 0 ? d.x : d.y;
  if (da == 0.0f) return;
  float t = (c - oa) / da;
  if (!(t > 0.0f) || !(t < s.t)) return;
  float other = axis == 0 ? o.y + t * d.y : o.x + t * d.x;
  float z = o.z + t * d.z;
  if (other < -half || other > half || z < z0 || z > z1) return;
  s.t = t;
  s.n = axis == 0 ? f3(sign, 0, 0) : f3(0, sign, 0);
  s.rgb = rgb;
}
MJS_DEV void hit_sphere(F3 o, F3 d, F3 c, float r, F3 rgb, Surf& s) {
  F3 oc = sub(o, c);
  float b = dotf(oc, d), cc = dotf(oc, oc) - r * r;
  float disc = b * b - cc;
  if (disc < 0.0f) return;
  float t = -b - sqrtf(disc);
  if (!(t > 0.0f) || !(t < s.t)) return;
  s.t = t;
  F3 p = add(o, mul(t, d));
  s.n = normalize(sub(p, c));
  s.rgb = rgb;
}
MJS_DEV void hit_aabb(F3 o, F3 d, F3 c, float h, F3 rgb, Surf& s) {
  // axis-aligned cube of half-size h: slab test, entry face gives the normal
  float tmin = 0.0f, tmax = s.t;
  int ax = -1;
  float sg = 0.0f;
  const float oo[3] = {o.x - c.x, o.y - c.y, o.z - c.z}, dd[3] = {d.x, d.y, d.z};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (dd[k] == 0.0f) {
      if (oo[k] < -h || oo[k] > h) return;
      continue;
    }
    float t1 = (-h - oo[k]) / dd[k], t2 = (h - oo[k]) / dd[k];
    float sgn = -1.0f;
    if (t1 > t2) { float tmp = t1; t1 = t2; t2 = tmp; sgn = 1.0f; }
    if (t1 > tmin) { tmin = t1; ax = k; sg = sgn; }
    if (t2 < tmax) tmax = t2;
    if (tmin > tmax) return;
  }
  if (ax < 0 || !(tmin > 0.0f) || !(tmin < s.t)) return;
  s.t = tmin;
  s.n = ax == 0 ? f3(sg, 0, 0) : ax == 1 ? f3(0, sg, 0) : f3(0, 0, sg);
  s.rgb = rgb;
}

// normalisation for the shading of BODY-MOUNTED cameras (the Button-Push wrist camera, every pixel of which is shaded per env):
// v_rsq_f32 and one Newton step instead of an IEEE square root and an IEEE division (together ~20 instructions, up to 13 times
// per pixel: shading was 59 of the wrist camera's 132 us). Within 1 ulp of normalize(); the CPU restatement keeps 1 / sqrtf, so
// a colour byte differs by one level where the value sits on a rounding boundary: 2e-5 of the wrist images' bytes
// (profiles/r03_k_camera_hoists.txt; the test's bound is 2e-4 and stays). The fixed scene cameras, whose images are
// byte-identical to the restatement's, and the Pointmass scene keep normalize().
MJS_DEV F3 normalize_fast(F3 a) {
  const float x = dotf(a, a);
  float y = __builtin_amdgcn_rsqf(x);
  y = y * (1.5f - 0.5f * x * y * y);  // (a second step changes no image: what is left is 1 / sqrtf's own double rounding)
  return mul(y, a);
}
// Blinn-Phong with MuJoCo's default headlight + the scene's positional spot lights (dir 0 0 -1)
template <int NLIGHT, bool FAST = false>
MJS_DEV F3 shade(F3 p, F3 n, F3 eye, F3 rgb, const float (*lights)[3]) {
  auto unit = [](F3 a) { return FAST ? normalize_fast(a) : normalize(a); };
  F3 v = unit(sub(eye, p));
  if (dotf(n, v) < 0.0f) n = mul(-1.0f, n);
  float diff = 0.0f, spec = 0.0f;
  {  // headlight at the camera
    float ndl = dotf(n, v);
    if (ndl > 0.0f) {
      diff = diff + MJS_HEADLIGHT_DIFFUSE * ndl;
      spec = spec + MJS_HEADLIGHT_SPECULAR * pow_pow2(ndl, MJS_MATERIAL_SHININESS_POW2);
    }
  }
#pragma unroll
  for (int k = 0; k < NLIGHT; k++) {
    F3 lv = sub(f3(lights[k][0], lights[k][1], lights[k][2]), p);
    // outside the 45 degree cone (cos^2 = 1/2), decided on the un-normalised vector
    if (lv.z <= 0.0f || lv.z * lv.z < MJS_LIGHT_CUTOFF_COS2 * dotf(lv, lv)) continue;
    F3 l = unit(lv);
    float spotcos = l.z;  // cos between -l and the light direction (0,0,-1)
    float spot = pow_pow2(spotcos, 3) * pow_pow2(spotcos, 1);  // exponent 10
    float ndl = dotf(n, l);
    if (ndl > 0.0f) {
      diff = diff + MJS_LIGHT_DIFFUSE * ndl * spot;
      F3 hv = unit(add(l, v));
      float ndh = dotf(n, hv);
      if (ndh > 0.0f) spec = spec + MJS_LIGHT_SPECULAR * pow_pow2(ndh, MJS_MATERIAL_SHININESS_POW2) * spot;
    }
  }
  float k = MJS_HEADLIGHT_AMBIENT + diff, sp = MJS_MATERIAL_SPECULAR * spec;
  return F3{rgb.x * k + sp, rgb.y * k + sp, rgb.z * k + sp};
}

MJS_DEV uint8_t to_u8(float c) {
  c = c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c);
  return (uint8_t)(int)(c * 255.0f + 0.5f);
}

MJS_DEV F3 pixel_ray_axes(const RenderParams& p, int row, int col, const float* right, const float* up, const float* back) {
  float aspect = (float)p.W / (float)p.H;
  float px = (2.0f * ((float)col + 0.5f) / (float)p.W - 1.0f) * p.cam.tan_half * aspect;
  float py = (1.0f - 2.0f * ((float)row + 0.5f) / (float)p.H) * p.cam.tan_half;
  F3 d = f3(px * right[0] + py * up[0] - back[0], px * right[1] + py * up[1] - back[1], px * right[2] + py * up[2] - back[2]);
  return normalize(d);
}
MJS_DEV F3 pixel_ray(const RenderParams& p, int row, int col) { return pixel_ray_axes(p, row, col, p.cam.right, p.cam.up, p.cam.back); }

// Pointmass-Reach scene: walled_pointmass_arena.xml:12-19 + pointmass sphere + target / mocap sites
__global__ __launch_bounds__(256) void pointmass_kernel(RenderParams p) {
  const int env = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= p.H * p.W) return;
  const int row = pix / p.W, col = pix - row * p.W;
  const size_t N = p.N;
  const double* st = p.state + env;
  const float qx = (float)st[pm::S_QX * N], qy = (float)st[pm::S_QY * N], tx = (float)st[pm::S_TX * N], ty = (float)st[pm::S_TY * N];
  const float mx = (float)st[pm::S_MX * N], my = (float)st[pm::S_MY * N];
  const F3 eye = f3(p.cam.pos[0], p.cam.pos[1], p.cam.pos[2]);
  const F3 d = pixel_ray(p, row, col);
  const F3 wall = f3(MJS_PM_WALL_RGB[0], MJS_PM_WALL_RGB[1], MJS_PM_WALL_RGB[2]);
  const float hi = (float)MJS_PM_ARENA_HI, wz = (float)MJS_PM_WALL_Z;
  Surf s;
  s.t = INFINITY;
  s.n = f3(0, 0, 1);
  s.rgb = f3(0, 0, 0);
  hit_rect_z(eye, d, 0.0f, hi, hi, f3(0, 0, 0), true, s);
  hit_wall(eye, d, 0, -hi, 1.0f, hi, 0.0f, 2.0f * wz, wall, s);
  hit_wall(eye, d, 1, -hi, 1.0f, hi, 0.0f, 2.0f * wz, wall, s);
  hit_wall(eye, d, 0, hi, -1.0f, hi, 0.0f, 2.0f * wz, wall, s);
  hit_wall(eye, d, 1, hi, -1.0f, hi, 0.0f, 2.0f * wz, wall, s);
  hit_aabb(eye, d, f3(tx, ty, (float)(MJS_PM_RADIUS / 2)), MJS_PM_TARGET_HALF, f3(MJS_PM_TARGET_RGB[0], MJS_PM_TARGET_RGB[1], MJS_PM_TARGET_RGB[2]), s);
  hit_sphere(eye, d, f3(mx, my, 0.0f), MJS_PM_MOCAP_SITE_RADIUS, f3(MJS_SITE_DEFAULT_RGB[0], MJS_SITE_DEFAULT_RGB[1], MJS_SITE_DEFAULT_RGB[2]), s);
  F3 c = f3(0, 0, 0);  // background
  if (s.t < INFINITY) c = shade<2>(add(eye, mul(s.t, d)), s.n, eye, s.rgb, MJS_PM_LIGHT_POS);
  // translucent pointmass sphere blended over whatever is behind it
  Surf b;
  b.t = s.t;
  b.n = f3(0, 0, 1);
  b.rgb = f3(0, 0, 0);
  hit_sphere(eye, d, f3(qx, qy, (float)MJS_PM_RADIUS), (float)MJS_PM_RADIUS, f3(MJS_PM_SPHERE_RGBA[0], MJS_PM_SPHERE_RGBA[1], MJS_PM_SPHERE_RGBA[2]), b);
  if (b.t < s.t) {
    F3 sc = shade<2>(add(eye, mul(b.t, d)), b.n, eye, b.rgb, MJS_PM_LIGHT_POS);
    const float a = MJS_PM_SPHERE_RGBA[3];
    c = F3{a * sc.x + (1.0f - a) * c.x, a * sc.y + (1.0f - a) * c.y, a * sc.z + (1.0f - a) * c.z};
  }
  uint8_t* o = p.out + ((size_t)env * p.H * p.W + pix) * 3;
  o[0] = to_u8(c.x);
  o[1] = to_u8(c.y);
  o[2] = to_u8(c.z);
}

// ------------------------------------------------------------------------ robot scenes
// Two stages: (1) one thread per env turns the joint state into a world-space primitive list
// (float32 records, the arm's collision proxies + stand-ins); (2) one thread per pixel
// ray-casts that list. Same arithmetic restrictions as above; the only CPU/GPU difference left is
// the float64 forward kinematics feeding stage 1 (1e-16 before the float32 rounding).
constexpr int PRIM_FLOATS = 20;
enum { PRIM_SPHERE = 1, PRIM_CAPSULE = 2, PRIM_CYLINDER = 3, PRIM_BOX = 4 };
// arm records: shoulder/upper-arm proxies g0..g2, base stand-in, forearm/wrist proxies g3..g9, gripper stand-in
constexpr int ARM_NREC = MJS_UR_NCOLGEOM + 2;
constexpr int RR_NPRIM = ARM_NREC + 1;  // + target site
constexpr int MAX_NPRIM = 32;           // the scene kernel keeps one candidate bit per primitive

// conservative bounding-sphere radius of a primitive (about p0 for spheres/boxes, about the segment midpoint
// for capsules/cylinders), with slack for float32 rounding: used only to SKIP exact tests that cannot hit
MJS_DEV float bound_radius(float r) { return r * 1.02f + 1.0e-3f; }
MJS_DEV void put3(float* dst, V3 v) { dst[0] = (float)v.x; dst[1] = (float)v.y; dst[2] = (float)v.z; }
MJS_DEV void put_rgb(float* dst, const float* rgb) { dst[0] = rgb[0]; dst[1] = rgb[1]; dst[2] = rgb[2]; }

// arm proxies + base stand-in + gripper stand-in: ARM_NREC records in the order given above
MJS_DEV void arm_prims(const rr::Chain& c, float* out);

__global__ __launch_bounds__(64) void reach_prims_kernel(const double* state, float* prims, int N) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= N) return;
  double q[6];
#pragma unroll
  for (int j = 0; j < 6; j++) q[j] = state[(size_t)(rr::S_Q + j) * N + i];
  rr::Chain c;
  rr::fk(q, c);
  float* out = prims + (size_t)i * RR_NPRIM * PRIM_FLOATS;
  arm_prims(c, out);
  {  // target site
    float* pr = out + ARM_NREC * PRIM_FLOATS;
    pr[0] = (float)PRIM_SPHERE;
    pr[1] = (float)state[(size_t)(rr::S_TARGET + 0) * N + i];
    pr[2] = (float)state[(size_t)(rr::S_TARGET + 1) * N + i];
    pr[3] = (float)state[(size_t)(rr::S_TARGET + 2) * N + i];
    pr[13] = MJS_RR_TARGET_RADIUS;
    pr[17] = bound_radius(MJS_RR_TARGET_RADIUS);
    put_rgb(pr + 14, MJS_RR_TARGET_RGB);
  }
}

// rotation matrix (row-major) of a MuJoCo quaternion (w, x, y, z), normalised first
MJS_DEV void quat_to_mat(const double* q, double* R) {
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
MJS_DEV void camera_body_prims(V3 pos, V3 ax, V3 ay, float* box, float* lens) {
  box[0] = (float)PRIM_BOX;
  put3(box + 1, pos); put3(box + 4, ax); put3(box + 7, ay);
  box[10] = (float)MJS_CAM_BOX_HALF[0]; box[11] = (float)MJS_CAM_BOX_HALF[1]; box[12] = (float)MJS_CAM_BOX_HALF[2];
  box[17] = bound_radius((float)(MJS_CAM_BOX_HALF[0] + MJS_CAM_BOX_HALF[1] + MJS_CAM_BOX_HALF[2]));
  put_rgb(box + 14, MJS_CAM_BODY_RGB);
  lens[0] = (float)PRIM_SPHERE;
  put3(lens + 1, pos);
  lens[13] = (float)MJS_CAM_SPHERE_RADIUS;
  lens[17] = bound_radius((float)MJS_CAM_SPHERE_RADIUS);
  put_rgb(lens + 14, MJS_CAM_BODY_RGB);
}

// Button-Push scene (robot_push_button.py:66-96): arm + stand-ins, wrist camera body, scene camera body,
// switch box and button. Also writes the wrist camera pose of every env (cams [N][12]).
constexpr int BP_NPRIM = ARM_NREC + 2 + 2 + 2;  // arm, wrist camera body, scene camera body, switch box + button
static_assert(BP_NPRIM <= MAX_NPRIM && RR_NPRIM <= MAX_NPRIM, "one candidate bit per primitive");
__global__ __launch_bounds__(64) void button_prims_kernel(const double* state, const uint8_t* flags, float* prims, float* cams, int N) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= N) return;
  double q[6];
#pragma unroll
  for (int j = 0; j < 6; j++) q[j] = state[(size_t)(bp::S_Q + j) * N + i];
  rr::Chain c;
  rr::fk(q, c);
  float* out = prims + (size_t)i * BP_NPRIM * PRIM_FLOATS;
  arm_prims(c, out);
  // flange frame: x = wrist_3 x, y = -wrist_3 z, z = wrist_3 y (MJS_UR_FLANGE_QUAT), origin on wrist_3 y
  const M3 R6 = c.R[6];
  const V3 fx = R6.cx, fy = -R6.cz, fz = R6.cy;
  const V3 fp = madd(c.p[6], MJS_UR_FLANGE_POS[1], R6.cy);
  double Rc[9];
  quat_to_mat(MJS_WCAM_QUAT, Rc);
  const V3 wpos = madd(madd(madd(fp, MJS_WCAM_POS[0], fx), MJS_WCAM_POS[1], fy), MJS_WCAM_POS[2], fz);
  const V3 wright = madd(madd(Rc[0] * fx, Rc[3], fy), Rc[6], fz);  // camera local x, y, z axes in the world
  const V3 wup = madd(madd(Rc[1] * fx, Rc[4], fy), Rc[7], fz);
  const V3 wback = madd(madd(Rc[2] * fx, Rc[5], fy), Rc[8], fz);
  camera_body_prims(wpos, wright, wup, out + ARM_NREC * PRIM_FLOATS, out + (ARM_NREC + 1) * PRIM_FLOATS);
  float* cm = cams + (size_t)i * 12;
  put3(cm, wpos); put3(cm + 3, wright); put3(cm + 6, wup); put3(cm + 9, wback);
  {  // scene camera body (seen by the wrist camera only)
    double Rs[9];
    quat_to_mat(MJS_BP_CAM_QUAT, Rs);
    camera_body_prims(v3(MJS_BP_CAM_POS[0], MJS_BP_CAM_POS[1], MJS_BP_CAM_POS[2]), v3(Rs[0], Rs[3], Rs[6]), v3(Rs[1], Rs[4], Rs[7]),
                      out + (ARM_NREC + 2) * PRIM_FLOATS, out + (ARM_NREC + 3) * PRIM_FLOATS);
  }
  const V3 sw = v3(state[(size_t)(bp::S_SWITCH + 0) * N + i], state[(size_t)(bp::S_SWITCH + 1) * N + i], state[(size_t)(bp::S_SWITCH + 2) * N + i]);
  {  // switch box
    float* pr = out + (ARM_NREC + 4) * PRIM_FLOATS;
    pr[0] = (float)PRIM_BOX;
    put3(pr + 1, v3(sw.x, sw.y, sw.z + MJS_SW_BOX_HALF));
    put3(pr + 4, v3(1, 0, 0)); put3(pr + 7, v3(0, 1, 0));
    pr[10] = pr[11] = pr[12] = (float)MJS_SW_BOX_HALF;
    pr[17] = bound_radius((float)(3 * MJS_SW_BOX_HALF));
    put_rgb(pr + 14, MJS_SW_BOX_RGB);
  }
  {  // button: red, green while the switch is active (switch.py:59)
    float* pr = out + (ARM_NREC + 5) * PRIM_FLOATS;
    pr[0] = (float)PRIM_CYLINDER;
    put3(pr + 1, v3(sw.x, sw.y, sw.z + MJS_SW_BUTTON_Z - MJS_SW_BUTTON_HALF));
    put3(pr + 4, v3(sw.x, sw.y, sw.z + MJS_SW_BUTTON_Z + MJS_SW_BUTTON_HALF));
    pr[13] = (float)MJS_SW_BUTTON_RADIUS;
    pr[17] = bound_radius((float)(MJS_SW_BUTTON_RADIUS + MJS_SW_BUTTON_HALF));
    put_rgb(pr + 14, (flags[i] & bp::FLAG_SWITCH_ACTIVE) ? MJS_SW_BUTTON_RGB_ON : MJS_SW_BUTTON_RGB_OFF);
  }
}

// Planar-Push scene (robot_planar_push.py:81-117): arm proxies + base stand-in, CylinderEEF, target site disc, blocks
constexpr int PP_NPRIM = ARM_NREC + 1 + MJS_PP_MAX_OBJECTS;  // capacity; a launch uses ARM_NREC + 1 + block slots of its handle
static_assert(PP_NPRIM <= MAX_NPRIM, "one candidate bit per primitive");
__global__ __launch_bounds__(64) void push_prims_kernel(const double* state, float* prims, int N, int nb, int nslots) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= N) return;
  double q[6];
#pragma unroll
  for (int j = 0; j < 6; j++) q[j] = state[(size_t)(pp::S_Q + j) * N + i];
  rr::Chain c;
  rr::fk(q, c);
  float* out = prims + (size_t)i * (ARM_NREC + 1 + nslots) * PRIM_FLOATS;
  arm_prims(c, out);
  {  // CylinderEEF replaces the gripper stand-in record: axis = flange z = wrist_3 y
    float* pr = out + (ARM_NREC - 1) * PRIM_FLOATS;
    const V3 centre = madd(c.p[6], MJS_UR_FLANGE_POS[1] + MJS_CYL_POS_Z, c.R[6].cy);
    pr[0] = (float)PRIM_CYLINDER;
    put3(pr + 1, madd(centre, -MJS_CYL_HALFLEN, c.R[6].cy));
    put3(pr + 4, madd(centre, MJS_CYL_HALFLEN, c.R[6].cy));
    pr[13] = (float)MJS_CYL_RADIUS;
    pr[17] = bound_radius((float)(MJS_CYL_RADIUS + MJS_CYL_HALFLEN));
    put_rgb(pr + 14, MJS_CYL_RGB);
  }
  {  // target site: a thin white disc on the floor
    float* pr = out + ARM_NREC * PRIM_FLOATS;
    const double tx = state[(size_t)(pp::S_TARGET + 0) * N + i], ty = state[(size_t)(pp::S_TARGET + 1) * N + i], tz = state[(size_t)(pp::S_TARGET + 2) * N + i];
    pr[0] = (float)PRIM_CYLINDER;
    put3(pr + 1, v3(tx, ty, tz - (double)MJS_PP_TARGET_HALF_HEIGHT));
    put3(pr + 4, v3(tx, ty, tz + (double)MJS_PP_TARGET_HALF_HEIGHT));
    pr[13] = (float)MJS_PP_TARGET_RADIUS;
    pr[17] = bound_radius((float)MJS_PP_TARGET_RADIUS + MJS_PP_TARGET_HALF_HEIGHT);
    put_rgb(pr + 14, MJS_PP_TARGET_RGB);
  }
  for (int b = 0; b < nslots; b++) {
    float* pr = out + (ARM_NREC + 1 + b) * PRIM_FLOATS;
    const double* bs = state + (size_t)(pp::S_BLOCK + pp::BLOCK_DIM * b) * N + i;
    double qn[4] = {bs[3 * (size_t)N], bs[4 * (size_t)N], bs[5 * (size_t)N], bs[6 * (size_t)N]};
    const double nrm = sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
#pragma unroll
    for (int k = 0; k < 4; k++) qn[k] = qn[k] / nrm;
    const M3 R = pp::quat_to_m3(qn);
    const V3 origin = v3(bs[0], bs[(size_t)N], bs[2 * (size_t)N]);
    pr[0] = (float)PRIM_BOX;
    put3(pr + 4, R.cx); put3(pr + 7, R.cy);
    const double shape = bs[13 * (size_t)N], sc = bs[14 * (size_t)N];
    if (shape >= 0.0 && b < nb) {
      // a mesh block is drawn as the bounding box of its hull (scaled), in its sampled colour (D-6: own ray caster)
      const int cat = ((int)shape) & 3, col = ((int)shape) >> 3;
      const double bx = 0.5 * (MJS_HULL_BOX_LO[cat][0] + MJS_HULL_BOX_HI[cat][0]) * sc, by = 0.5 * (MJS_HULL_BOX_LO[cat][1] + MJS_HULL_BOX_HI[cat][1]) * sc,
                   bz = 0.5 * (MJS_HULL_BOX_LO[cat][2] + MJS_HULL_BOX_HI[cat][2]) * sc;
      put3(pr + 1, v3(origin.x + (R.cx.x * bx + R.cy.x * by + R.cz.x * bz), origin.y + (R.cx.y * bx + R.cy.y * by + R.cz.y * bz), origin.z + (R.cx.z * bx + R.cy.z * by + R.cz.z * bz)));
      const float hx = (float)(0.5 * (MJS_HULL_BOX_HI[cat][0] - MJS_HULL_BOX_LO[cat][0]) * sc), hy = (float)(0.5 * (MJS_HULL_BOX_HI[cat][1] - MJS_HULL_BOX_LO[cat][1]) * sc),
                  hz = (float)(0.5 * (MJS_HULL_BOX_HI[cat][2] - MJS_HULL_BOX_LO[cat][2]) * sc);
      pr[10] = hx; pr[11] = hy; pr[12] = hz;
      pr[17] = bound_radius(hx + hy + hz);
      put_rgb(pr + 14, MJS_BLOCK_COLORS[col < 6 ? col : 5]);
    } else {
      put3(pr + 1, b < nb ? madd(origin, MJS_BLOCK_GEOM_Z, R.cz) : v3(0, 0, -10.0));  // unused slot: out of sight
      pr[10] = (float)MJS_BLOCK_HALF[0]; pr[11] = (float)MJS_BLOCK_HALF[1]; pr[12] = (float)MJS_BLOCK_HALF[2];
      pr[17] = bound_radius((float)(MJS_BLOCK_HALF[0] + MJS_BLOCK_HALF[1] + MJS_BLOCK_HALF[2]));
      put_rgb(pr + 14, MJS_BLOCK_RGB[b]);
    }
  }
}

MJS_DEV void arm_prims(const rr::Chain& c, float* out) {
#pragma unroll
  for (int g = 0; g < MJS_UR_NCOLGEOM; g++) {
    float* pr = out + (g < 3 ? g : g + 1) * PRIM_FLOATS;
    const int b = MJS_UR_COL_BODY[g];
    const M3 R = c.R[b];
    V3 gp = madd(madd(madd(c.p[b], MJS_UR_COL_POS[g][0], R.cx), MJS_UR_COL_POS[g][1], R.cy), MJS_UR_COL_POS[g][2], R.cz);
    V3 axis = (MJS_UR_COL_QUAT[g][1] != 0.0) ? -R.cy : R.cz;
    const double half = MJS_UR_COL_SIZE[g][1];
    pr[0] = MJS_UR_COL_TYPE[g] == 3 ? (float)PRIM_CAPSULE : (float)PRIM_CYLINDER;
    put3(pr + 1, madd(gp, -half, axis));
    put3(pr + 4, madd(gp, half, axis));
    pr[13] = (float)MJS_UR_COL_SIZE[g][0];
    pr[17] = bound_radius((float)(half + MJS_UR_COL_SIZE[g][0]));
    put_rgb(pr + 14, MJS_UR_COL_IS_JOINT[g] ? MJS_UR_URBLUE : MJS_UR_LINKGRAY);
  }
  {  // base stand-in: vertical cylinder on the floor
    float* pr = out + 3 * PRIM_FLOATS;
    pr[0] = (float)PRIM_CYLINDER;
    put3(pr + 1, v3(0, 0, 0));
    put3(pr + 4, v3(0, 0, 2.0 * MJS_UR_BASE_STANDIN[1]));
    pr[13] = MJS_UR_BASE_STANDIN[0];
    pr[17] = bound_radius(MJS_UR_BASE_STANDIN[0] + MJS_UR_BASE_STANDIN[1]);
    put_rgb(pr + 14, MJS_UR_JOINTGRAY);
  }
  {  // gripper stand-in: box in the flange frame (x = wrist_3 x, y = -wrist_3 z, z = wrist_3 y)
    float* pr = out + (ARM_NREC - 1) * PRIM_FLOATS;
    const M3 R = c.R[6];
    V3 centre = madd(c.p[6], MJS_UR_FLANGE_POS[1] + (double)MJS_G2F85_STANDIN_HALF[2], R.cy);
    pr[0] = (float)PRIM_BOX;
    put3(pr + 1, centre);
    put3(pr + 4, R.cx);   // box axis u (half[0])
    put3(pr + 7, -R.cz);  // box axis v (half[1]); w = u x v = flange z (half[2])
    pr[10] = MJS_G2F85_STANDIN_HALF[0]; pr[11] = MJS_G2F85_STANDIN_HALF[1]; pr[12] = MJS_G2F85_STANDIN_HALF[2];
    pr[17] = bound_radius(MJS_G2F85_STANDIN_HALF[0] + MJS_G2F85_STANDIN_HALF[1] + MJS_G2F85_STANDIN_HALF[2]);
    put_rgb(pr + 14, MJS_UR_BLACK);
  }
}

// capsule / capped cylinder between pa and pb (closed-form ray tests)
MJS_DEV void hit_capsule(F3 o, F3 d, F3 pa, F3 pb, float r, F3 rgb, Surf& s) {
  F3 ba = sub(pb, pa), oa = sub(o, pa);
  float baba = dotf(ba, ba), bard = dotf(ba, d), baoa = dotf(ba, oa), rdoa = dotf(d, oa), oaoa = dotf(oa, oa);
  float a = baba - bard * bard, b = baba * rdoa - baoa * bard, c = baba * oaoa - baoa * baoa - r * r * baba;
  float h = b * b - a * c;
  if (h < 0.0f) return;
  if (a > 0.0f) {
    float t = (-b - sqrtf(h)) / a;
    float y = baoa + t * bard;
    if (y > 0.0f && y < baba) {
      if (!(t > 0.0f) || !(t < s.t)) return;
      s.t = t;
      F3 pn = sub(add(oa, mul(t, d)), mul(y / baba, ba));
      s.n = mul(1.0f / r, pn);
      s.rgb = rgb;
      return;
    }
    // end caps
    F3 oc = y <= 0.0f ? oa : sub(o, pb);
    float bb = dotf(d, oc), cc = dotf(oc, oc) - r * r;
    float hh = bb * bb - cc;
    if (hh > 0.0f) {
      float tc = -bb - sqrtf(hh);
      if (!(tc > 0.0f) || !(tc < s.t)) return;
      s.t = tc;
      s.n = mul(1.0f / r, add(oc, mul(tc, d)));
      s.rgb = rgb;
    }
  }
}
MJS_DEV void hit_cylinder(F3 o, F3 d, F3 pa, F3 pb, float r, F3 rgb, Surf& s) {
  F3 ba = sub(pb, pa), oa = sub(o, pa);
  float baba = dotf(ba, ba), bard = dotf(ba, d), baoa = dotf(ba, oa);
  float k2 = baba - bard * bard, k1 = baba * dotf(oa, d) - baoa * bard, k0 = baba * dotf(oa, oa) - baoa * baoa - r * r * baba;
  float h = k1 * k1 - k2 * k0;
  if (h < 0.0f) return;
  if (k2 > 0.0f) {
    float t = (-k1 - sqrtf(h)) / k2;
    float y = baoa + t * bard;
    if (y > 0.0f && y < baba) {
      if (!(t > 0.0f) || !(t < s.t)) return;
      s.t = t;
      s.n = mul(1.0f / r, sub(add(oa, mul(t, d)), mul(y / baba, ba)));
      s.rgb = rgb;
      return;
    }
  }
  // caps
  if (bard == 0.0f) return;
  float tc = ((bard < 0.0f ? baba : 0.0f) - baoa) / bard;  // the cap facing the ray
  if (!(tc > 0.0f) || !(tc < s.t)) return;
  F3 q = add(oa, mul(tc, d));
  float yc = bard < 0.0f ? baba : 0.0f;
  F3 radial = sub(q, mul(yc / baba, ba));
  if (dotf(radial, radial) > r * r) return;
  s.t = tc;
  float inv = 1.0f / sqrtf(baba);
  s.n = mul(bard < 0.0f ? inv : -inv, ba);
  s.rgb = rgb;
}
MJS_DEV void hit_obb(F3 o, F3 d, F3 c, F3 u, F3 v, F3 half, F3 rgb, Surf& s) {
  F3 w = F3{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
  F3 oc = sub(o, c);
  const float oo[3] = {dotf(oc, u), dotf(oc, v), dotf(oc, w)}, dd[3] = {dotf(d, u), dotf(d, v), dotf(d, w)}, hh[3] = {half.x, half.y, half.z};
  float tmin = 0.0f, tmax = s.t, sg = 0.0f;
  int ax = -1;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (dd[k] == 0.0f) {
      if (oo[k] < -hh[k] || oo[k] > hh[k]) return;
      continue;
    }
    float t1 = (-hh[k] - oo[k]) / dd[k], t2 = (hh[k] - oo[k]) / dd[k], sgn = -1.0f;
    if (t1 > t2) { float tmp = t1; t1 = t2; t2 = tmp; sgn = 1.0f; }
    if (t1 > tmin) { tmin = t1; ax = k; sg = sgn; }
    if (t2 < tmax) tmax = t2;
    if (tmin > tmax) return;
  }
  if (ax < 0 || !(tmin > 0.0f) || !(tmin < s.t)) return;
  s.t = tmin;
  s.n = mul(sg, ax == 0 ? u : ax == 1 ? v : w);
  s.rgb = rgb;
}

// A sphere or box that contains the eye can never be hit (hit_sphere: the near root is negative; hit_obb: no slab is
// entered at t > 0), so both kernels drop it before any per-ray work. This is the camera's own body (camera_body_prims
// puts a box and a lens sphere AT the camera position): without the rule every ray of the image runs both exact tests.
MJS_DEV bool eye_inside(const float* pr, F3 eye) {
  const int type = (int)pr[0];
  const F3 oc = sub(eye, f3(pr[1], pr[2], pr[3]));
  if (type == PRIM_SPHERE) return dotf(oc, oc) < pr[13] * pr[13] * 0.999f;
  if (type != PRIM_BOX) return false;
  const F3 u = f3(pr[4], pr[5], pr[6]), v = f3(pr[7], pr[8], pr[9]);
  const F3 w = F3{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
  return fabsf(dotf(oc, u)) < pr[10] * 0.999f && fabsf(dotf(oc, v)) < pr[11] * 0.999f && fabsf(dotf(oc, w)) < pr[12] * 0.999f;
}

// One wavefront per 8x8 pixel tile (4 tiles per workgroup). The env's primitive list is staged in LDS once per
// workgroup; lane k of every wavefront tests primitive k against the cone that bounds the tile's 64 rays, and the
// ballot of that test is the list of primitives the wavefront walks (typically 0-3 of ~20). All of this only
// SKIPS exact tests that cannot hit: the image is unchanged.
__global__ __launch_bounds__(256) void robot_scene_kernel(RenderParams p, const float* prims) {
  __shared__ float lds_prims[MAX_NPRIM * PRIM_FLOATS];
  const int env = blockIdx.y;
  const int nprim = p.nprim;
  {
    const float* pe = prims + (size_t)env * nprim * PRIM_FLOATS;
    for (int k = threadIdx.x; k < nprim * PRIM_FLOATS; k += 256) lds_prims[k] = pe[k];
  }
  __syncthreads();
  const int tiles_x = (p.W + 7) >> 3;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int row0 = (tile / tiles_x) * 8, col0 = (tile % tiles_x) * 8;
  const int row = row0 + (lane >> 3), col = col0 + (lane & 7);
  const bool inside = row < p.H && col < p.W;
  F3 eye, d;
  const float* right = p.cam.right;
  const float* up = p.cam.up;
  const float* back = p.cam.back;
  if (p.env_cams) {  // body-mounted camera: pose computed per env by the primitive stage
    const float* cm = p.env_cams + (size_t)env * 12;
    eye = f3(cm[0], cm[1], cm[2]);
    right = cm + 3; up = cm + 6; back = cm + 9;
  } else {
    eye = f3(p.cam.pos[0], p.cam.pos[1], p.cam.pos[2]);
  }
  d = pixel_ray_axes(p, row, col, right, up, back);
  // cone bounding the tile's rays: axis = ray through the tile centre, cos(half-angle) = min over the lanes
  F3 dc;
  float cosa;
  {
    float aspect = (float)p.W / (float)p.H;
    float px = (2.0f * ((float)col0 + 4.0f) / (float)p.W - 1.0f) * p.cam.tan_half * aspect;
    float py = (1.0f - 2.0f * ((float)row0 + 4.0f) / (float)p.H) * p.cam.tan_half;
    dc = normalize(f3(px * right[0] + py * up[0] - back[0], px * right[1] + py * up[1] - back[1], px * right[2] + py * up[2] - back[2]));
    cosa = dotf(d, dc);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) cosa = fminf(cosa, __shfl_xor(cosa, m));
    cosa = cosa - 1.0e-5f;
  }
  unsigned cand = 0;
  {
    bool c = false;
    if (lane < nprim) {
      const float* pr = lds_prims + lane * PRIM_FLOATS;
      const int type = (int)pr[0];
      const F3 p0 = f3(pr[1], pr[2], pr[3]), p1 = f3(pr[4], pr[5], pr[6]);
      const F3 oc = sub((type == PRIM_CAPSULE || type == PRIM_CYLINDER) ? mul(0.5f, add(p0, p1)) : p0, eye);
      const float L2 = dotf(oc, oc), br = pr[17];
      if (L2 <= br * br) c = !eye_inside(pr, eye);  // the eye is inside the bound (inside the primitive itself: never hit)
      else {
        const float L = sqrtf(L2), cost = dotf(oc, dc) / L, sinb = br / L;
        const float cosb = sqrtf(fmaxf(0.0f, 1.0f - sinb * sinb)), sina = sqrtf(fmaxf(0.0f, 1.0f - cosa * cosa));
        c = cost >= cosa * cosb - sina * sinb - 1.0e-4f;  // angle(oc, axis) <= half-angle + asin(R / L)
      }
    }
    cand = (unsigned)__ballot(c);
  }
  Surf s;
  s.t = INFINITY;
  s.n = f3(0, 0, 1);
  s.rgb = f3(0, 0, 0);
  hit_rect_z(eye, d, 0.0f, (float)MJS_ROBOT_ARENA_HALF, (float)MJS_ROBOT_ARENA_HALF, f3(MJS_RR_FLOOR_RGB[0], MJS_RR_FLOOR_RGB[1], MJS_RR_FLOOR_RGB[2]), false, s);
  while (cand) {  // ascending primitive index = the oracle's test order
    const int k = __ffs(cand) - 1;
    cand &= cand - 1;
    const float* pr = lds_prims + k * PRIM_FLOATS;
    const int type = (int)pr[0];
    const F3 p0 = f3(pr[1], pr[2], pr[3]), p1 = f3(pr[4], pr[5], pr[6]);
    // per-ray bounding-sphere reject
    const F3 oc = sub((type == PRIM_CAPSULE || type == PRIM_CYLINDER) ? mul(0.5f, add(p0, p1)) : p0, eye);
    const float along = dotf(oc, d), off2 = dotf(oc, oc) - along * along, br = pr[17];
    if (!(inside && off2 <= br * br && along + br > 0.0f)) continue;
    const F3 rgb = f3(pr[14], pr[15], pr[16]);
    if (type == PRIM_SPHERE) hit_sphere(eye, d, p0, pr[13], rgb, s);
    else if (type == PRIM_CAPSULE) hit_capsule(eye, d, p0, p1, pr[13], rgb, s);
    else if (type == PRIM_CYLINDER) hit_cylinder(eye, d, p0, p1, pr[13], rgb, s);
    else hit_obb(eye, d, p0, p1, f3(pr[7], pr[8], pr[9]), f3(pr[10], pr[11], pr[12]), rgb, s);
  }
  if (!inside) return;
  F3 c = f3(0, 0, 0);
  if (s.t < INFINITY)  // body-mounted camera: the fast normalisation, as the rectangle walk (normalize_fast)
    c = p.env_cams ? shade<6, true>(add(eye, mul(s.t, d)), s.n, eye, s.rgb, MJS_RR_LIGHT_POS) : shade<6>(add(eye, mul(s.t, d)), s.n, eye, s.rgb, MJS_RR_LIGHT_POS);
  uint8_t* o = p.out + ((size_t)env * p.H * p.W + (size_t)row * p.W + col) * 3;
  o[0] = to_u8(c.x);
  o[1] = to_u8(c.y);
  o[2] = to_u8(c.z);
}


// ---- rectangle walk: one workgroup per env image, or per band of rows of a larger one (H and W multiples of 8) ------------
// The 8x8-tile walk above makes every tile test every primitive whose bounding sphere touches the tile's ray cone. For
// primitives a few cm from the lens (the wrist camera sits on the flange next to the gripper and the last wrist links) the
// spheres contain the eye or reach the camera plane and the cone test keeps them for every tile; at 64x64 a tile is also
// ~30 cm wide at the arm's distance from the scene camera, so most exact tests are for rays that miss. Here each primitive
// gets its pixel rectangle once per image (prim_rect: tangent extents of the bounding spheres, or the primitive's box
// clipped at the near plane), a tile's candidate list is the ballot of "rectangle overlaps the tile", and the walk itself
// is the tile kernel's: ascending primitive index, strict-< update, normals kept in registers. Tiles are taken from an
// LDS counter by whichever wavefront is free (the arm covers a few columns of tiles: a static split leaves three
// wavefronts waiting for the fourth), colours are packed into LDS and written out as whole dwords.
//
// Fixed scene cameras: everything that does not depend on the env is computed ONCE per (camera, H, W) by
// scene_background_kernel and kept in HBM/L2 (20 B per pixel): the normalised pixel ray, the floor's ray parameter and
// the shaded floor / background colour. A fixed camera sees the same floor in every env, so only tiles a primitive's
// rectangle touches are visited and only pixels a primitive wins are shaded; the rest is a copy of the table.
// Same arithmetic per ray, per test and per shaded pixel as the tile kernel, evaluated by the same functions: the image
// is identical (tests: test_scene_camera_kernels_agree_byte_for_byte, both cameras, varied poses).
constexpr int RECT_WALK_MAX_PIXELS = 64 * 64;  // 16 KB of packed colours in LDS
struct Background {
  const float4* ray;    // [H*W] normalised ray direction (xyz) and the floor's ray parameter (w; +inf = no floor there)
  const uint32_t* rgb;  // [H*W] shaded floor / background colour, r | g << 8 | b << 16
};
MJS_DEV uint32_t pack_rgb(F3 c) { return (uint32_t)to_u8(c.x) | ((uint32_t)to_u8(c.y) << 8) | ((uint32_t)to_u8(c.z) << 16); }
__global__ __launch_bounds__(256) void scene_background_kernel(RenderParams p, float4* ray, uint32_t* rgb) {
  const int px = blockIdx.x * 256 + threadIdx.x;
  if (px >= p.H * p.W) return;
  const int row = px / p.W, col = px - row * p.W;
  const F3 eye = f3(p.cam.pos[0], p.cam.pos[1], p.cam.pos[2]);
  const F3 d = pixel_ray(p, row, col);
  Surf s;
  s.t = INFINITY;
  s.n = f3(0, 0, 1);
  s.rgb = f3(0, 0, 0);
  hit_rect_z(eye, d, 0.0f, (float)MJS_ROBOT_ARENA_HALF, (float)MJS_ROBOT_ARENA_HALF, f3(MJS_RR_FLOOR_RGB[0], MJS_RR_FLOOR_RGB[1], MJS_RR_FLOOR_RGB[2]), false, s);
  F3 c = f3(0, 0, 0);
  if (s.t < INFINITY) c = shade<6>(add(eye, mul(s.t, d)), s.n, eye, s.rgb, MJS_RR_LIGHT_POS);
  ray[px] = make_float4(d.x, d.y, d.z, s.t);
  rgb[px] = pack_rgb(c);
}

// Tan-space bounding rectangle (x / depth, y / depth) of an oriented box clipped at the near plane depth = zn, for
// primitives whose bounding spheres reach the camera plane (the gripper next to the wrist camera): the extremes of a
// linear-fractional function over a convex polytope lie at its vertices, so the rectangle of the clipped box's vertices
// (corners in front of the plane + edge crossings) bounds every ray that can hit it at depth >= zn. Returns false if
// the whole box is behind the plane. C, A, B, D: box centre and scaled half-axes in camera coordinates (x right, y up,
// z depth).
MJS_DEV bool clipped_box_rect(const float* C, const float* A, const float* B, const float* D, float zn, float& xlo, float& xhi, float& ylo, float& yhi) {
  xlo = ylo = INFINITY;
  xhi = yhi = -INFINITY;
  bool any = false;
  for (int i = 0; i < 8; i++) {
    const float sa = (i & 1) ? 1.0f : -1.0f, sb = (i & 2) ? 1.0f : -1.0f, sd = (i & 4) ? 1.0f : -1.0f;
    const float P[3] = {C[0] + sa * A[0] + sb * B[0] + sd * D[0], C[1] + sa * A[1] + sb * B[1] + sd * D[1], C[2] + sa * A[2] + sb * B[2] + sd * D[2]};
    if (!(P[2] > zn)) continue;
    any = true;
    const float iz = 1.0f / P[2];
    xlo = fminf(xlo, P[0] * iz); xhi = fmaxf(xhi, P[0] * iz);
    ylo = fminf(ylo, P[1] * iz); yhi = fmaxf(yhi, P[1] * iz);
    for (int e = 0; e < 3; e++) {  // the three edges leaving this corner: towards a corner at or behind the plane?
      const float* E = e == 0 ? A : e == 1 ? B : D;
      const float se = -2.0f * (e == 0 ? sa : e == 1 ? sb : sd);
      const float Q2 = P[2] + se * E[2];
      if (Q2 > zn) continue;
      const float f = (P[2] - zn) / (P[2] - Q2);  // in (0, 1]
      const float X = (P[0] + f * se * E[0]) / zn, Y = (P[1] + f * se * E[1]) / zn;
      xlo = fminf(xlo, X); xhi = fmaxf(xhi, X);
      ylo = fminf(ylo, Y); yhi = fmaxf(yhi, Y);
    }
  }
  return any;
}
// the primitive as an oriented box (centre c, unit axes u v w, half sizes h), inflated for float32 safety
MJS_DEV void prim_box(const float* pr, F3 eye, F3& c, F3& u, F3& v, F3& w, F3& h) {
  const int type = (int)pr[0];
  if (type == PRIM_BOX) {
    c = f3(pr[1], pr[2], pr[3]); u = f3(pr[4], pr[5], pr[6]); v = f3(pr[7], pr[8], pr[9]);
    w = F3{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
    h = f3(pr[10], pr[11], pr[12]);
  } else if (type == PRIM_SPHERE) {
    c = f3(pr[1], pr[2], pr[3]); u = f3(1, 0, 0); v = f3(0, 1, 0); w = f3(0, 0, 1);
    h = f3(pr[13], pr[13], pr[13]);
  } else {  // capsule / cylinder: the box around the segment and its radius (a capsule's round caps reach a radius further)
    const F3 pa = f3(pr[1], pr[2], pr[3]), pb = f3(pr[4], pr[5], pr[6]), ba = sub(pb, pa);
    const float len = sqrtf(dotf(ba, ba));
    c = mul(0.5f, add(pa, pb));
    u = len > 1.0e-6f ? mul(1.0f / len, ba) : f3(0, 0, 1);
    // radial axes: v towards the eye (a round primitive has no preferred ones; this choice keeps the box's corners, which
    // stick out of the round surface, away from an eye that is close to it)
    const F3 ce = sub(eye, c);
    const F3 perp = sub(ce, mul(dotf(ce, u), u));
    const F3 e = fabsf(u.x) < 0.9f ? f3(1, 0, 0) : f3(0, 1, 0);
    v = dotf(perp, perp) > 1.0e-10f ? normalize(perp) : normalize(F3{u.y * e.z - u.z * e.y, u.z * e.x - u.x * e.z, u.x * e.y - u.y * e.x});
    w = F3{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
    h = f3(0.5f * len + (type == PRIM_CAPSULE ? pr[13] : 0.0f), pr[13], pr[13]);
  }
  h = f3(h.x * 1.02f + 1.0e-3f, h.y * 1.02f + 1.0e-3f, h.z * 1.02f + 1.0e-3f);
}

MJS_DEV void exact_test(const float* pr, F3 eye, F3 d, Surf& s) {
  const int type = (int)pr[0];
  const F3 p0 = f3(pr[1], pr[2], pr[3]), p1 = f3(pr[4], pr[5], pr[6]);
  const F3 rgb = f3(pr[14], pr[15], pr[16]);
  if (type == PRIM_SPHERE) hit_sphere(eye, d, p0, pr[13], rgb, s);
  else if (type == PRIM_CAPSULE) hit_capsule(eye, d, p0, p1, pr[13], rgb, s);
  else if (type == PRIM_CYLINDER) hit_cylinder(eye, d, p0, p1, pr[13], rgb, s);
  else hit_obb(eye, d, p0, p1, f3(pr[7], pr[8], pr[9]), f3(pr[10], pr[11], pr[12]), rgb, s);
}
// The exact tests again with everything that does not depend on the ray taken out: for one env image the eye is fixed, so
// ba, oa, their dot products and the whole constant term of the quadratic (capsule / cylinder), oc and cc (sphere), the third
// axis and the eye in box coordinates (box) are the same for every pixel. prim_consts evaluates them once per primitive with
// the expressions of hit_* above (same operations in the same order: the rays' results are bitwise those of exact_test,
// test_scene_camera_kernels_agree_byte_for_byte compares the two walks); exact_test_pre reads them from LDS.
constexpr int PRIM_CONSTS = 12;
MJS_DEV void prim_consts(const float* pr, F3 o, float* pc) {
  const int type = (int)pr[0];
  const F3 p0 = f3(pr[1], pr[2], pr[3]), p1 = f3(pr[4], pr[5], pr[6]);
  if (type == PRIM_SPHERE) {
    const float r = pr[13];
    F3 oc = sub(o, p0);
    float cc = dotf(oc, oc) - r * r;
    pc[0] = oc.x; pc[1] = oc.y; pc[2] = oc.z; pc[3] = cc;
  } else if (type == PRIM_CAPSULE || type == PRIM_CYLINDER) {
    const float r = pr[13];
    F3 ba = sub(p1, p0), oa = sub(o, p0);
    float baba = dotf(ba, ba), baoa = dotf(ba, oa), oaoa = dotf(oa, oa);
    float c = baba * oaoa - baoa * baoa - r * r * baba;
    pc[0] = ba.x; pc[1] = ba.y; pc[2] = ba.z; pc[3] = oa.x; pc[4] = oa.y; pc[5] = oa.z; pc[6] = baba; pc[7] = baoa; pc[8] = c;
    pc[9] = 1.0f / r;             // the normals' scale (a division per hit)
    pc[10] = 1.0f / sqrtf(baba);  // cylinder caps
  } else {
    const F3 u = p1, v = f3(pr[7], pr[8], pr[9]);
    F3 w = F3{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
    F3 oc = sub(o, p0);
    pc[0] = w.x; pc[1] = w.y; pc[2] = w.z; pc[3] = dotf(oc, u); pc[4] = dotf(oc, v); pc[5] = dotf(oc, w);
  }
}
MJS_DEV void exact_test_pre(const float* pr, const float* pc, F3 o, F3 d, Surf& s) {
  const int type = (int)pr[0];
  const F3 rgb = f3(pr[14], pr[15], pr[16]);
  if (type == PRIM_SPHERE) {
    const F3 c = f3(pr[1], pr[2], pr[3]), oc = f3(pc[0], pc[1], pc[2]);
    float b = dotf(oc, d), cc = pc[3];
    float disc = b * b - cc;
    if (disc < 0.0f) return;
    float t = -b - sqrtf(disc);
    if (!(t > 0.0f) || !(t < s.t)) return;
    s.t = t;
    F3 p = add(o, mul(t, d));
    s.n = normalize(sub(p, c));
    s.rgb = rgb;
  } else if (type == PRIM_CAPSULE) {
    const float r = pr[13];
    const F3 pb = f3(pr[4], pr[5], pr[6]), ba = f3(pc[0], pc[1], pc[2]), oa = f3(pc[3], pc[4], pc[5]);
    const float baba = pc[6], baoa = pc[7], c = pc[8];
    float bard = dotf(ba, d), rdoa = dotf(d, oa);
    float a = baba - bard * bard, b = baba * rdoa - baoa * bard;
    float h = b * b - a * c;
    if (h < 0.0f) return;
    if (a > 0.0f) {
      float t = (-b - sqrtf(h)) / a;
      float y = baoa + t * bard;
      if (y > 0.0f && y < baba) {
        if (!(t > 0.0f) || !(t < s.t)) return;
        s.t = t;
        F3 pn = sub(add(oa, mul(t, d)), mul(y / baba, ba));
        s.n = mul(pc[9], pn);
        s.rgb = rgb;
        return;
      }
      // end caps
      F3 oc = y <= 0.0f ? oa : sub(o, pb);
      float bb = dotf(d, oc), cc = dotf(oc, oc) - r * r;
      float hh = bb * bb - cc;
      if (hh > 0.0f) {
        float tc = -bb - sqrtf(hh);
        if (!(tc > 0.0f) || !(tc < s.t)) return;
        s.t = tc;
        s.n = mul(pc[9], add(oc, mul(tc, d)));
        s.rgb = rgb;
      }
    }
  } else if (type == PRIM_CYLINDER) {
    const float r = pr[13];
    const F3 ba = f3(pc[0], pc[1], pc[2]), oa = f3(pc[3], pc[4], pc[5]);
    const float baba = pc[6], baoa = pc[7], k0 = pc[8];
    float bard = dotf(ba, d);
    float k2 = baba - bard * bard, k1 = baba * dotf(oa, d) - baoa * bard;
    float h = k1 * k1 - k2 * k0;
    if (h < 0.0f) return;
    if (k2 > 0.0f) {
      float t = (-k1 - sqrtf(h)) / k2;
      float y = baoa + t * bard;
      if (y > 0.0f && y < baba) {
        if (!(t > 0.0f) || !(t < s.t)) return;
        s.t = t;
        s.n = mul(pc[9], sub(add(oa, mul(t, d)), mul(y / baba, ba)));
        s.rgb = rgb;
        return;
      }
    }
    // caps
    if (bard == 0.0f) return;
    float tc = ((bard < 0.0f ? baba : 0.0f) - baoa) / bard;  // the cap facing the ray
    if (!(tc > 0.0f) || !(tc < s.t)) return;
    F3 q = add(oa, mul(tc, d));
    float yc = bard < 0.0f ? baba : 0.0f;
    F3 radial = sub(q, mul(yc / baba, ba));
    if (dotf(radial, radial) > r * r) return;
    s.t = tc;
    float inv = pc[10];
    s.n = mul(bard < 0.0f ? inv : -inv, ba);
    s.rgb = rgb;
  } else {
    const F3 u = f3(pr[4], pr[5], pr[6]), v = f3(pr[7], pr[8], pr[9]), w = f3(pc[0], pc[1], pc[2]);
    const float oo[3] = {pc[3], pc[4], pc[5]}, dd[3] = {dotf(d, u), dotf(d, v), dotf(d, w)}, hh[3] = {pr[10], pr[11], pr[12]};
    float tmin = 0.0f, tmax = s.t, sg = 0.0f;
    int ax = -1;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (dd[k] == 0.0f) {
        if (oo[k] < -hh[k] || oo[k] > hh[k]) return;
        continue;
      }
      float t1 = (-hh[k] - oo[k]) / dd[k], t2 = (hh[k] - oo[k]) / dd[k], sgn = -1.0f;
      if (t1 > t2) { float tmp = t1; t1 = t2; t2 = tmp; sgn = 1.0f; }
      if (t1 > tmin) { tmin = t1; ax = k; sg = sgn; }
      if (t2 < tmax) tmax = t2;
      if (tmin > tmax) return;
    }
    if (ax < 0 || !(tmin > 0.0f) || !(tmin < s.t)) return;
    s.t = tmin;
    s.n = mul(sg, ax == 0 ? u : ax == 1 ? v : w);
    s.rgb = rgb;
  }
}
// Pixel rectangle [r0, r1] x [c0, c1] that bounds every pixel whose ray can hit the primitive (empty: r1 < r0). Far
// primitives: tangent extents of the bounding sphere(s); primitives whose spheres reach the camera plane (the gripper and
// the last wrist links next to the wrist camera): the primitive's box clipped at 1 mm depth; both when both apply.
// Conservative by construction: it only SKIPS exact tests that cannot hit.
MJS_DEV void prim_rect(const RenderParams& p, const float* pr, F3 eye, const float* right, const float* up, const float* back, int* rect) {
  const int type = (int)pr[0];
  const bool two = type == PRIM_CAPSULE || type == PRIM_CYLINDER;
  const float rad = two ? bound_radius(pr[13]) : pr[17];
  const float aspect = (float)p.W / (float)p.H, tx = p.cam.tan_half * aspect, ty = p.cam.tan_half;
  float xlo = INFINITY, xhi = -INFINITY, ylo = INFINITY, yhi = -INFINITY;
  bool everything = false, behind = true;  // behind: the whole bound lies behind the camera plane
  float zmin = INFINITY;
  for (int e = 0; e < (two ? 2 : 1); e++) {
    const F3 v = sub(f3(pr[1 + 3 * e], pr[2 + 3 * e], pr[3 + 3 * e]), eye);
    const float x = v.x * right[0] + v.y * right[1] + v.z * right[2], y = v.x * up[0] + v.y * up[1] + v.z * up[2];
    const float z = -(v.x * back[0] + v.y * back[1] + v.z * back[2]);
    zmin = fminf(zmin, z);
    if (z + rad * 1.05f + 1.0e-3f > 0.0f) behind = false;
    if (!(z > rad * 1.05f + 1.0e-3f)) { everything = true; continue; }  // the bound reaches the camera plane
    const float den = z * z - rad * rad;
    const float sx = rad * sqrtf(x * x + den), sy = rad * sqrtf(y * y + den);
    xlo = fminf(xlo, (x * z - sx) / den); xhi = fmaxf(xhi, (x * z + sx) / den);
    ylo = fminf(ylo, (y * z - sy) / den); yhi = fmaxf(yhi, (y * z + sy) / den);
  }
  if (!behind && zmin < 4.0f * rad) {  // near the camera plane: the spheres' tangent extents are loose or void, clip the box
    F3 bc, bu, bv, bw, bh;
    prim_box(pr, eye, bc, bu, bv, bw, bh);
    const F3 rel = sub(bc, eye), R = f3(right[0], right[1], right[2]), U = f3(up[0], up[1], up[2]), Bk = f3(back[0], back[1], back[2]);
    const float lu = dotf(rel, bu), lv = dotf(rel, bv), lw = dotf(rel, bw);
    // the eye within 5 mm of the box: rays may hit it nearer than the clip depth, keep what the spheres gave
    const bool near = fabsf(lu) < bh.x + 5.0e-3f && fabsf(lv) < bh.y + 5.0e-3f && fabsf(lw) < bh.z + 5.0e-3f;
    if (!near) {
      const float Cc[3] = {dotf(rel, R), dotf(rel, U), -dotf(rel, Bk)};
      const float Ac[3] = {bh.x * dotf(bu, R), bh.x * dotf(bu, U), -bh.x * dotf(bu, Bk)};
      const float Bc[3] = {bh.y * dotf(bv, R), bh.y * dotf(bv, U), -bh.y * dotf(bv, Bk)};
      const float Dc[3] = {bh.z * dotf(bw, R), bh.z * dotf(bw, U), -bh.z * dotf(bw, Bk)};
      float bxlo, bxhi, bylo, byhi;
      if (!clipped_box_rect(Cc, Ac, Bc, Dc, 1.0e-3f, bxlo, bxhi, bylo, byhi)) behind = true;
      else if (everything) { xlo = bxlo; xhi = bxhi; ylo = bylo; yhi = byhi; everything = false; }
      else { xlo = fmaxf(xlo, bxlo); xhi = fminf(xhi, bxhi); ylo = fmaxf(ylo, bylo); yhi = fminf(yhi, byhi); }
    }
  }
  int r0 = 0, r1 = p.H - 1, c0 = 0, c1 = p.W - 1;
  if (!everything) {
    // px = (2 (col + 0.5) / W - 1) tx,  py = (1 - 2 (row + 0.5) / H) ty   (pixel_ray_axes)
    const float cl = (xlo / tx + 1.0f) * 0.5f * (float)p.W - 0.5f, ch = (xhi / tx + 1.0f) * 0.5f * (float)p.W - 0.5f;
    const float rl = (1.0f - yhi / ty) * 0.5f * (float)p.H - 0.5f, rh = (1.0f - ylo / ty) * 0.5f * (float)p.H - 0.5f;
    // wholly outside the image (with the one-pixel pad), or an empty intersection of the two bounds
    if (!(ch >= -1.5f) || !(cl <= (float)p.W + 0.5f) || !(rh >= -1.5f) || !(rl <= (float)p.H + 0.5f) || cl > ch || rl > rh) behind = true;
    // clamp in float first (the extents can be huge for bounds near the camera plane), then pad by a pixel
    c0 = (int)floorf(fminf(fmaxf(cl, -1.0f), (float)p.W)) - 1; c1 = (int)ceilf(fminf(fmaxf(ch, -1.0f), (float)p.W)) + 1;
    r0 = (int)floorf(fminf(fmaxf(rl, -1.0f), (float)p.H)) - 1; r1 = (int)ceilf(fminf(fmaxf(rh, -1.0f), (float)p.H)) + 1;
    c0 = max(c0, 0); r0 = max(r0, 0); c1 = min(c1, p.W - 1); r1 = min(r1, p.H - 1);
  }
  if (behind || eye_inside(pr, eye)) r1 = r0 - 1;  // nothing of it can be seen: empty rectangle
  rect[0] = r0; rect[1] = r1; rect[2] = c0; rect[3] = c1;
}

template <bool FIXED>  // FIXED: the task's scene camera with its ray / floor table (Background); otherwise p.env_cams
__global__ __launch_bounds__(256) void robot_scene_rect_walk_kernel(RenderParams p, const float* prims, Background bg, int band_rows) {
  __shared__ float lds_prims[MAX_NPRIM * PRIM_FLOATS];
  __shared__ int bbox[MAX_NPRIM * 4];
  __shared__ uint32_t image[RECT_WALK_MAX_PIXELS];  // packed colours, written out coalesced at the end
  __shared__ int next_tile;                        // tiles are taken in turn by whichever wavefront is free
  // body-mounted camera: the image-plane coordinates of a column / a row (pixel_ray_axes' px, py: two IEEE divisions per
  // pixel) are the same for every pixel of that column / row: computed once per workgroup with the same expressions, read
  // from LDS per pixel (bitwise the same rays; W <= RECT_WALK_MAX_PIXELS / 8)
  // per primitive, the same for every pixel of the env's image: centre of the bounding sphere relative to the eye and its
  // squared length (the per-ray reject of the candidate walk computed them per pixel)
  __shared__ float bound_oc[MAX_NPRIM][4];
  __shared__ float pconst[MAX_NPRIM][PRIM_CONSTS];  // prim_consts: the ray-independent part of each primitive's exact test
  __shared__ float pxtab[FIXED ? 1 : RECT_WALK_MAX_PIXELS / 8];
  __shared__ float pytab[FIXED ? 1 : RECT_WALK_MAX_PIXELS / 8];
  // one workgroup per (env, band of rows): images above 4096 pixels are cut into bands of band_rows rows (a multiple of 8,
  // band_rows * W <= 4096) that stage their 16 KB of colours each; a band is contiguous in the output
  const int env = blockIdx.x, nprim = p.nprim, tid = threadIdx.x, lane = tid & 63;
  const int band0 = blockIdx.y * band_rows, rows = min(band_rows, p.H - band0);
  const int npix = rows * p.W, pix0 = band0 * p.W;
  {
    const float* pe = prims + (size_t)env * nprim * PRIM_FLOATS;
    for (int k = tid; k < nprim * PRIM_FLOATS; k += 256) lds_prims[k] = pe[k];
    if (tid == 0) next_tile = 0;
    if (FIXED)  // the env-independent image first; only tiles a primitive touches are revisited
      for (int k = tid; k < npix; k += 256) image[k] = bg.rgb[pix0 + k];
    else {
      const float aspect = (float)p.W / (float)p.H;
      for (int col = tid; col < p.W; col += 256) pxtab[col] = (2.0f * ((float)col + 0.5f) / (float)p.W - 1.0f) * p.cam.tan_half * aspect;
      for (int r = tid; r < rows; r += 256) pytab[r] = (1.0f - 2.0f * ((float)(band0 + r) + 0.5f) / (float)p.H) * p.cam.tan_half;
    }
  }
  F3 eye = f3(p.cam.pos[0], p.cam.pos[1], p.cam.pos[2]);
  const float* right = p.cam.right;
  const float* up = p.cam.up;
  const float* back = p.cam.back;
  if (!FIXED) {
    const float* cm = p.env_cams + (size_t)env * 12;
    eye = f3(cm[0], cm[1], cm[2]);
    right = cm + 3; up = cm + 6; back = cm + 9;
  }
  __syncthreads();
  if (tid < nprim) {
    {
      const float* pr = lds_prims + tid * PRIM_FLOATS;
      const int type = (int)pr[0];
      const F3 p0 = f3(pr[1], pr[2], pr[3]), p1 = f3(pr[4], pr[5], pr[6]);
      const F3 oc = sub((type == PRIM_CAPSULE || type == PRIM_CYLINDER) ? mul(0.5f, add(p0, p1)) : p0, eye);
      bound_oc[tid][0] = oc.x; bound_oc[tid][1] = oc.y; bound_oc[tid][2] = oc.z; bound_oc[tid][3] = dotf(oc, oc);
      prim_consts(pr, eye, pconst[tid]);
    }
    prim_rect(p, lds_prims + tid * PRIM_FLOATS, eye, right, up, back, bbox + 4 * tid);
    // A capped cylinder that FOLLOWS a capsule on the same segment with the same radius (the UR5e's last two collision
    // proxies, MJS_UR_COL_* 8 and 9) can never win a pixel: it lies inside the capsule, its side hits have the capsule's
    // ray parameter to the bit (same expressions in hit_capsule / hit_cylinder) and lose the strict-< update to the lower
    // index, its flat caps lie behind the capsule's round ones, and both share one bounding sphere. Its exact test only
    // costs time - most of all in the wrist camera's image, where the pair sits 1-4 cm from the lens: empty rectangle.
    if (tid > 0) {
      const float* a = lds_prims + (tid - 1) * PRIM_FLOATS;
      const float* b = lds_prims + tid * PRIM_FLOATS;
      if ((int)a[0] == PRIM_CAPSULE && (int)b[0] == PRIM_CYLINDER && a[1] == b[1] && a[2] == b[2] && a[3] == b[3] && a[4] == b[4] && a[5] == b[5] && a[6] == b[6] &&
          a[13] == b[13])
        bbox[4 * tid + 1] = bbox[4 * tid] - 1;
    }
  }
  __syncthreads();
  const int tiles_x = p.W >> 3, ntiles = npix >> 6;
  for (;;) {
    int tile = 0;
    if (lane == 0) tile = atomicAdd(&next_tile, 1);
    tile = __builtin_amdgcn_readfirstlane(tile);
    if (tile >= ntiles) break;
    const int row0 = band0 + (tile / tiles_x) * 8, col0 = (tile % tiles_x) * 8;
    const int row = row0 + (lane >> 3), col = col0 + (lane & 7);
    bool c = false;
    if (lane < nprim) c = bbox[4 * lane] <= row0 + 7 && bbox[4 * lane + 1] >= row0 && bbox[4 * lane + 2] <= col0 + 7 && bbox[4 * lane + 3] >= col0;
    unsigned cand = (unsigned)__ballot(c);
    F3 d;
    Surf s;
    s.t = INFINITY;
    s.n = f3(0, 0, 1);
    s.rgb = f3(0, 0, 0);
    float t_floor = INFINITY;
    if (FIXED) {  // the env-independent part comes from the table: a tile no primitive touches is a copy
      if (!cand) continue;
      const float4 r4 = bg.ray[row * p.W + col];
      d = f3(r4.x, r4.y, r4.z);
      t_floor = r4.w;
      s.t = t_floor;  // the floor's hit bounds the walk exactly as in the sequential test order (floor first)
    } else {
      const float px = pxtab[col], py = pytab[row - band0];
      d = normalize(f3(px * right[0] + py * up[0] - back[0], px * right[1] + py * up[1] - back[1], px * right[2] + py * up[2] - back[2]));  // = pixel_ray_axes
      hit_rect_z(eye, d, 0.0f, (float)MJS_ROBOT_ARENA_HALF, (float)MJS_ROBOT_ARENA_HALF, f3(MJS_RR_FLOOR_RGB[0], MJS_RR_FLOOR_RGB[1], MJS_RR_FLOOR_RGB[2]), false, s);
    }
    while (cand) {  // ascending primitive index = the oracle's test order
      const int k = __ffs(cand) - 1;
      cand &= cand - 1;
      const float* pr = lds_prims + k * PRIM_FLOATS;
      const F3 oc = f3(bound_oc[k][0], bound_oc[k][1], bound_oc[k][2]);
      const float along = dotf(oc, d), off2 = bound_oc[k][3] - along * along, br = pr[17];  // per-ray bounding-sphere reject
      if (!(off2 <= br * br && along + br > 0.0f)) continue;
      if (along - br > s.t) continue;  // every point of the bound lies beyond the nearest hit so far: it cannot pass the strict-< update
      exact_test_pre(pr, pconst[k], eye, d, s);
    }
    if (FIXED) {
      if (s.t < t_floor) image[(row - band0) * p.W + col] = pack_rgb(shade<6>(add(eye, mul(s.t, d)), s.n, eye, s.rgb, MJS_RR_LIGHT_POS));  // a primitive won the pixel
    } else {
      F3 cc = f3(0, 0, 0);
      if (s.t < INFINITY) cc = shade<6, true>(add(eye, mul(s.t, d)), s.n, eye, s.rgb, MJS_RR_LIGHT_POS);
      image[(row - band0) * p.W + col] = pack_rgb(cc);
    }
  }
  __syncthreads();
  uint32_t* o32 = reinterpret_cast<uint32_t*>(p.out + ((size_t)env * p.H * p.W + pix0) * 3);  // pix0 * 3 bytes: a multiple of 4 (W % 8 == 0)
  for (int q = tid; q < (npix >> 2); q += 256) {
    const uint32_t a = image[4 * q], b = image[4 * q + 1], c = image[4 * q + 2], d = image[4 * q + 3];
    o32[3 * q + 0] = a | (b << 24);
    o32[3 * q + 1] = (b >> 8) | (c << 16);
    o32[3 * q + 2] = (c >> 16) | (d << 8);
  }
}

}  // namespace rend
