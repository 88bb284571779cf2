// AffineSpace.h -- owl::common::affine3f: 3x3 linear part (columns vx,vy,vz) + translation p,
// the 4x3 column-major layout OWL_MATRIX_FORMAT_OWL describes (12 floats).
#pragma once
#include "owl/common/math/vec.h"

namespace owl {
namespace common {

struct linear3f {
  vec3f vx, vy, vz;
  inline __both__ linear3f() : vx(1.f, 0.f, 0.f), vy(0.f, 1.f, 0.f), vz(0.f, 0.f, 1.f) {}
  inline __both__ linear3f(const vec3f &a, const vec3f &b, const vec3f &c) : vx(a), vy(b), vz(c) {}
  inline __both__ float det() const { return dot(vx, cross(vy, vz)); }
  inline __both__ linear3f inverse() const {
    const float d = 1.f / det();
    const vec3f r0 = cross(vy, vz) * d, r1 = cross(vz, vx) * d, r2 = cross(vx, vy) * d;  // rows of the inverse
    return linear3f(vec3f(r0.x, r1.x, r2.x), vec3f(r0.y, r1.y, r2.y), vec3f(r0.z, r1.z, r2.z));
  }
};
inline __both__ vec3f operator*(const linear3f &l, const vec3f &v) { return l.vx * v.x + l.vy * v.y + l.vz * v.z; }
inline __both__ linear3f operator*(const linear3f &a, const linear3f &b) { return linear3f(a * b.vx, a * b.vy, a * b.vz); }

struct affine3f {
  linear3f l;
  vec3f p;
  inline __both__ affine3f() : l(), p(0.f) {}
  inline __both__ affine3f(const linear3f &l_, const vec3f &p_) : l(l_), p(p_) {}
  static inline __both__ affine3f translate(const vec3f &t) { return affine3f(linear3f(), t); }
  static inline __both__ affine3f scale(const vec3f &s) {
    return affine3f(linear3f(vec3f(s.x, 0.f, 0.f), vec3f(0.f, s.y, 0.f), vec3f(0.f, 0.f, s.z)), vec3f(0.f));
  }
  static inline __both__ affine3f rotate(const vec3f &axis_, float angle) {
    const vec3f u = normalize(axis_);
    const float s = sinf(angle), c = cosf(angle), t = 1.f - c;
    return affine3f(linear3f(vec3f(u.x * u.x * t + c, u.x * u.y * t + u.z * s, u.x * u.z * t - u.y * s),
                             vec3f(u.x * u.y * t - u.z * s, u.y * u.y * t + c, u.y * u.z * t + u.x * s),
                             vec3f(u.x * u.z * t + u.y * s, u.y * u.z * t - u.x * s, u.z * u.z * t + c)),
                    vec3f(0.f));
  }
};
inline __both__ vec3f xfmPoint(const affine3f &a, const vec3f &v) { return a.l * v + a.p; }
inline __both__ vec3f xfmVector(const affine3f &a, const vec3f &v) { return a.l * v; }
inline __both__ affine3f operator*(const affine3f &a, const affine3f &b) { return affine3f(a.l * b.l, a.l * b.p + a.p); }
inline __both__ affine3f rcp(const affine3f &a) {
  const linear3f il = a.l.inverse();
  return affine3f(il, -(il * a.p));
}

}  // namespace common
}  // namespace owl
