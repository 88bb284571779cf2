// box.h -- owl::common::box_t<vec>: axis-aligned box with the extend/including/contains set the
// samples' bounds programs use (deviceCode.cu:45-47 `box3f().extend(a).extend(b)`).
#pragma once
#include "owl/common/math/vec.h"

namespace owl {
namespace common {

template <typename T>
inline __both__ T empty_bounds_lower();
template <typename T>
inline __both__ T empty_bounds_upper();
template <>
inline __both__ float empty_bounds_lower<float>() { return +INFINITY; }
template <>
inline __both__ float empty_bounds_upper<float>() { return -INFINITY; }
template <>
inline __both__ double empty_bounds_lower<double>() { return +INFINITY; }
template <>
inline __both__ double empty_bounds_upper<double>() { return -INFINITY; }
template <>
inline __both__ int32_t empty_bounds_lower<int32_t>() { return INT32_MAX; }
template <>
inline __both__ int32_t empty_bounds_upper<int32_t>() { return INT32_MIN; }
template <>
inline __both__ uint32_t empty_bounds_lower<uint32_t>() { return UINT32_MAX; }
template <>
inline __both__ uint32_t empty_bounds_upper<uint32_t>() { return 0; }

template <typename V>
struct box_t {
  typedef V vec_t;
  typedef typename V::scalar_t scalar_t;
  enum { dims = V::dims };
  V lower, upper;

  inline __both__ box_t() : lower(empty_bounds_lower<scalar_t>()), upper(empty_bounds_upper<scalar_t>()) {}
  inline __both__ explicit box_t(const V &p) : lower(p), upper(p) {}
  inline __both__ box_t(const V &lo, const V &hi) : lower(lo), upper(hi) {}

  inline __both__ box_t &extend(const V &p) {
    lower = min(lower, p);
    upper = max(upper, p);
    return *this;
  }
  inline __both__ box_t &extend(const box_t &b) {
    lower = min(lower, b.lower);
    upper = max(upper, b.upper);
    return *this;
  }
  inline __both__ box_t including(const V &p) const { return box_t(min(lower, p), max(upper, p)); }
  inline __both__ box_t including(const box_t &b) const { return box_t(min(lower, b.lower), max(upper, b.upper)); }
  inline __both__ bool contains(const V &p) const { return !(any_less_than(p, lower) || any_greater_than(p, upper)); }
  inline __both__ bool overlaps(const box_t &b) const {
    return !(any_less_than(b.upper, lower) || any_greater_than(b.lower, upper));
  }
  inline __both__ bool empty() const { return any_less_than(upper, lower); }
  inline __both__ V center() const { return (lower + upper) / scalar_t(2); }
  inline __both__ V span() const { return upper - lower; }
  inline __both__ V size() const { return upper - lower; }
  inline __both__ typename V::scalar_t volume() const {
    V s = size();
    scalar_t v = s[0];
    for (int i = 1; i < dims; i++) v *= s[i];
    return v;
  }
};

template <typename V>
inline __both__ bool operator==(const box_t<V> &a, const box_t<V> &b) { return a.lower == b.lower && a.upper == b.upper; }
template <typename V>
inline std::ostream &operator<<(std::ostream &o, const box_t<V> &b) { return o << "[" << b.lower << ":" << b.upper << "]"; }

typedef box_t<vec2f> box2f;
typedef box_t<vec3f> box3f;
typedef box_t<vec4f> box4f;
typedef box_t<vec2i> box2i;
typedef box_t<vec3i> box3i;
typedef box_t<vec3d> box3d;

}  // namespace common
}  // namespace owl
