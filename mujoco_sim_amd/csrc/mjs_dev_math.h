// mjs_dev_math.h — small float64 vector helpers for the gfx950 kernels (one env per lane,
// everything lives in VGPRs after full unrolling).
#pragma once
#include <hip/hip_runtime.h>

#define MJS_DEV __device__ __forceinline__

struct V3 {
  double x, y, z;
};
MJS_DEV V3 v3(double x, double y, double z) { return V3{x, y, z}; }
MJS_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
MJS_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
MJS_DEV V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
MJS_DEV V3 operator*(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
MJS_DEV V3& operator+=(V3& a, V3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
MJS_DEV double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
MJS_DEV V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// a + s*b
MJS_DEV V3 madd(V3 a, double s, V3 b) { return V3{a.x + s * b.x, a.y + s * b.y, a.z + s * b.z}; }

// rotation matrix stored as its three columns (world images of the local x, y, z axes)
struct M3 {
  V3 cx, cy, cz;
};
// R * Ry(angle): columns (c*cx - s*cz, cy, s*cx + c*cz)
MJS_DEV M3 mul_rot_y(M3 R, double c, double s) { return M3{madd(c * R.cx, -s, R.cz), R.cy, madd(s * R.cx, c, R.cz)}; }
// R * Rz(angle): columns (c*cx + s*cy, -s*cx + c*cy, cz)
MJS_DEV M3 mul_rot_z(M3 R, double c, double s) { return M3{madd(c * R.cx, s, R.cy), madd(c * R.cy, -s, R.cx), R.cz}; }
// R * Ry(+90 deg) (MJCF quat "1 0 1 0"): columns (-cz, cy, cx)
MJS_DEV M3 mul_quarter_y(M3 R) { return M3{-R.cz, R.cy, R.cx}; }

// spatial inertia about the world origin in world axes: symmetric I, h = m*c, mass
struct SI {
  double xx, xy, xz, yy, yz, zz;
  V3 h;
  double m;
};
MJS_DEV SI operator+(SI a, SI b) {
  return SI{a.xx + b.xx, a.xy + b.xy, a.xz + b.xz, a.yy + b.yy, a.yz + b.yz, a.zz + b.zz, a.h + b.h, a.m + b.m};
}
// spatial vector [angular, linear]
struct SV {
  V3 w, v;
};
MJS_DEV SV operator+(SV a, SV b) { return SV{a.w + b.w, a.v + b.v}; }
MJS_DEV SV operator*(double s, SV a) { return SV{s * a.w, s * a.v}; }
MJS_DEV double dot(SV a, SV b) { return dot(a.w, b.w) + dot(a.v, b.v); }
MJS_DEV V3 sym_mul(const SI& I, V3 w) {
  return V3{I.xx * w.x + I.xy * w.y + I.xz * w.z, I.xy * w.x + I.yy * w.y + I.yz * w.z, I.xz * w.x + I.yz * w.y + I.zz * w.z};
}
// momentum / force = I * motion
MJS_DEV SV si_mul(const SI& I, SV a) { return SV{sym_mul(I, a.w) + cross(I.h, a.v), I.m * a.v + cross(a.w, I.h)}; }
// motion x motion
MJS_DEV SV cross_motion(SV v, SV s) { return SV{cross(v.w, s.w), cross(v.w, s.v) + cross(v.v, s.w)}; }
// motion x* force
MJS_DEV SV cross_force(SV v, SV f) { return SV{cross(v.w, f.w) + cross(v.v, f.v), cross(v.w, f.v)}; }

MJS_DEV double clampd(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }
MJS_DEV bool bad_value(double x) { return !(fabs(x) <= 1e10); }  // NaN, inf or > mjMAXVAL
