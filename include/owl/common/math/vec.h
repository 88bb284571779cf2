// vec.h -- owl::common::vec_t<T,N> for N = 2,3,4: the small vector algebra the OWL samples and
// device programs use (vec3f(float), vec +- scalar, dot/cross/normalize, float3 conversion).
#pragma once
#include "owl/common/owl-common.h"

namespace owl {
namespace common {

template <typename T, int N>
struct vec_t;

#define OWL_VEC_COMMON(N)                                                               \
  typedef T scalar_t;                                                                   \
  enum { dims = N };                                                                    \
  inline __both__ T &operator[](size_t i) { return (&x)[i]; }                           \
  inline __both__ const T &operator[](size_t i) const { return (&x)[i]; }

template <typename T>
struct vec_t<T, 2> {
  OWL_VEC_COMMON(2)
  T x, y;
  inline __both__ vec_t() {}
  inline __both__ vec_t(T s) : x(s), y(s) {}
  inline __both__ vec_t(T x_, T y_) : x(x_), y(y_) {}
  template <typename O>
  inline __both__ explicit vec_t(const vec_t<O, 2> &o) : x((T)o.x), y((T)o.y) {}
#if defined(__HIPCC__)
  inline __both__ vec_t(const float2 &v) : x((T)v.x), y((T)v.y) {}
  inline __both__ vec_t(const int2 &v) : x((T)v.x), y((T)v.y) {}
  inline __both__ vec_t(const uint2 &v) : x((T)v.x), y((T)v.y) {}
  inline __both__ explicit vec_t(const uint3 &v) : x((T)v.x), y((T)v.y) {}
  inline __both__ operator float2() const { return make_float2((float)x, (float)y); }
#endif
};

template <typename T>
struct vec_t<T, 3> {
  OWL_VEC_COMMON(3)
  T x, y, z;
  inline __both__ vec_t() {}
  inline __both__ vec_t(T s) : x(s), y(s), z(s) {}
  inline __both__ vec_t(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
  inline __both__ vec_t(const vec_t<T, 2> &xy, T z_) : x(xy.x), y(xy.y), z(z_) {}
  template <typename O>
  inline __both__ explicit vec_t(const vec_t<O, 3> &o) : x((T)o.x), y((T)o.y), z((T)o.z) {}
  inline __both__ explicit vec_t(const vec_t<T, 4> &o);
#if defined(__HIPCC__)
  inline __both__ vec_t(const float3 &v) : x((T)v.x), y((T)v.y), z((T)v.z) {}
  inline __both__ vec_t(const int3 &v) : x((T)v.x), y((T)v.y), z((T)v.z) {}
  inline __both__ vec_t(const uint3 &v) : x((T)v.x), y((T)v.y), z((T)v.z) {}
  inline __both__ explicit vec_t(const float4 &v) : x((T)v.x), y((T)v.y), z((T)v.z) {}
  inline __both__ operator float3() const { return make_float3((float)x, (float)y, (float)z); }
#endif
};

template <typename T>
struct vec_t<T, 4> {
  OWL_VEC_COMMON(4)
  T x, y, z, w;
  inline __both__ vec_t() {}
  inline __both__ vec_t(T s) : x(s), y(s), z(s), w(s) {}
  inline __both__ vec_t(T x_, T y_, T z_, T w_) : x(x_), y(y_), z(z_), w(w_) {}
  inline __both__ vec_t(const vec_t<T, 3> &v, T w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
  template <typename O>
  inline __both__ explicit vec_t(const vec_t<O, 4> &o) : x((T)o.x), y((T)o.y), z((T)o.z), w((T)o.w) {}
#if defined(__HIPCC__)
  inline __both__ vec_t(const float4 &v) : x((T)v.x), y((T)v.y), z((T)v.z), w((T)v.w) {}
  inline __both__ operator float4() const { return make_float4((float)x, (float)y, (float)z, (float)w); }
#endif
};
#undef OWL_VEC_COMMON

template <typename T>
inline __both__ vec_t<T, 3>::vec_t(const vec_t<T, 4> &o) : x(o.x), y(o.y), z(o.z) {}

// ---- componentwise operators, generated for N = 2,3,4 --------------------------------------
#define OWL_VEC_BINOP(op)                                                                          \
  template <typename T>                                                                            \
  inline __both__ vec_t<T, 2> operator op(const vec_t<T, 2> &a, const vec_t<T, 2> &b) {            \
    return vec_t<T, 2>(a.x op b.x, a.y op b.y);                                                    \
  }                                                                                                \
  template <typename T>                                                                            \
  inline __both__ vec_t<T, 3> operator op(const vec_t<T, 3> &a, const vec_t<T, 3> &b) {            \
    return vec_t<T, 3>(a.x op b.x, a.y op b.y, a.z op b.z);                                        \
  }                                                                                                \
  template <typename T>                                                                            \
  inline __both__ vec_t<T, 4> operator op(const vec_t<T, 4> &a, const vec_t<T, 4> &b) {            \
    return vec_t<T, 4>(a.x op b.x, a.y op b.y, a.z op b.z, a.w op b.w);                            \
  }                                                                                                \
  template <typename T, int N>                                                                     \
  inline __both__ vec_t<T, N> operator op(const vec_t<T, N> &a, const T &s) {                      \
    return a op vec_t<T, N>(s);                                                                    \
  }                                                                                                \
  template <typename T, int N>                                                                     \
  inline __both__ vec_t<T, N> operator op(const T &s, const vec_t<T, N> &b) {                      \
    return vec_t<T, N>(s) op b;                                                                    \
  }                                                                                                \
  template <typename T, int N>                                                                     \
  inline __both__ vec_t<T, N> &operator op##=(vec_t<T, N> &a, const vec_t<T, N> &b) {              \
    a = a op b;                                                                                    \
    return a;                                                                                      \
  }                                                                                                \
  template <typename T, int N>                                                                     \
  inline __both__ vec_t<T, N> &operator op##=(vec_t<T, N> &a, const T &s) {                        \
    a = a op vec_t<T, N>(s);                                                                       \
    return a;                                                                                      \
  }
OWL_VEC_BINOP(+)
OWL_VEC_BINOP(-)
OWL_VEC_BINOP(*)
OWL_VEC_BINOP(/)
#undef OWL_VEC_BINOP

template <typename T>
inline __both__ vec_t<T, 2> operator-(const vec_t<T, 2> &a) { return vec_t<T, 2>(-a.x, -a.y); }
template <typename T>
inline __both__ vec_t<T, 3> operator-(const vec_t<T, 3> &a) { return vec_t<T, 3>(-a.x, -a.y, -a.z); }
template <typename T>
inline __both__ vec_t<T, 4> operator-(const vec_t<T, 4> &a) { return vec_t<T, 4>(-a.x, -a.y, -a.z, -a.w); }

template <typename T, int N>
inline __both__ bool operator==(const vec_t<T, N> &a, const vec_t<T, N> &b) {
  for (int i = 0; i < N; i++)
    if (!(a[i] == b[i])) return false;
  return true;
}
template <typename T, int N>
inline __both__ bool operator!=(const vec_t<T, N> &a, const vec_t<T, N> &b) { return !(a == b); }

// scalar min/max that also work in device code
template <typename T>
inline __both__ T owl_min(T a, T b) { return b < a ? b : a; }
template <typename T>
inline __both__ T owl_max(T a, T b) { return a < b ? b : a; }
#if defined(__HIPCC__)
// float min/max follow fminf/fmaxf (what CUDA's min/max overloads for float do)
template <>
inline __both__ float owl_min<float>(float a, float b) { return fminf(a, b); }
template <>
inline __both__ float owl_max<float>(float a, float b) { return fmaxf(a, b); }
#endif

#define OWL_VEC_FN2(name, expr)                                                           \
  template <typename T>                                                                   \
  inline __both__ vec_t<T, 2> name(const vec_t<T, 2> &a, const vec_t<T, 2> &b) {          \
    return vec_t<T, 2>(expr(a.x, b.x), expr(a.y, b.y));                                   \
  }                                                                                       \
  template <typename T>                                                                   \
  inline __both__ vec_t<T, 3> name(const vec_t<T, 3> &a, const vec_t<T, 3> &b) {          \
    return vec_t<T, 3>(expr(a.x, b.x), expr(a.y, b.y), expr(a.z, b.z));                   \
  }                                                                                       \
  template <typename T>                                                                   \
  inline __both__ vec_t<T, 4> name(const vec_t<T, 4> &a, const vec_t<T, 4> &b) {          \
    return vec_t<T, 4>(expr(a.x, b.x), expr(a.y, b.y), expr(a.z, b.z), expr(a.w, b.w));   \
  }
OWL_VEC_FN2(min, owl_min)
OWL_VEC_FN2(max, owl_max)
#undef OWL_VEC_FN2

template <typename T, int N>
inline __both__ T dot(const vec_t<T, N> &a, const vec_t<T, N> &b) {
  T s = a[0] * b[0];
  for (int i = 1; i < N; i++) s += a[i] * b[i];
  return s;
}
template <typename T>
inline __both__ vec_t<T, 3> cross(const vec_t<T, 3> &a, const vec_t<T, 3> &b) {
  return vec_t<T, 3>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <typename T, int N>
inline __both__ T length(const vec_t<T, N> &v) { return (T)sqrtf((float)dot(v, v)); }
template <int N>
inline __both__ double length(const vec_t<double, N> &v) { return sqrt(dot(v, v)); }
template <typename T, int N>
inline __both__ vec_t<T, N> normalize(const vec_t<T, N> &v) { return v * (T(1) / length(v)); }
template <typename T, int N>
inline __both__ T reduce_min(const vec_t<T, N> &v) {
  T m = v[0];
  for (int i = 1; i < N; i++) m = owl_min(m, v[i]);
  return m;
}
template <typename T, int N>
inline __both__ T reduce_max(const vec_t<T, N> &v) {
  T m = v[0];
  for (int i = 1; i < N; i++) m = owl_max(m, v[i]);
  return m;
}
template <typename T, int N>
inline __both__ bool any_less_than(const vec_t<T, N> &a, const vec_t<T, N> &b) {
  for (int i = 0; i < N; i++)
    if (a[i] < b[i]) return true;
  return false;
}
template <typename T, int N>
inline __both__ bool any_greater_than(const vec_t<T, N> &a, const vec_t<T, N> &b) {
  for (int i = 0; i < N; i++)
    if (a[i] > b[i]) return true;
  return false;
}

template <typename T, int N>
inline std::ostream &operator<<(std::ostream &o, const vec_t<T, N> &v) {
  o << "(";
  for (int i = 0; i < N; i++) o << (i ? "," : "") << v[i];
  return o << ")";
}

#define OWL_VEC_TYPEDEFS(T, s)   \
  typedef vec_t<T, 2> vec2##s;   \
  typedef vec_t<T, 3> vec3##s;   \
  typedef vec_t<T, 4> vec4##s;
OWL_VEC_TYPEDEFS(float, f)
OWL_VEC_TYPEDEFS(double, d)
OWL_VEC_TYPEDEFS(int32_t, i)
OWL_VEC_TYPEDEFS(uint32_t, ui)
OWL_VEC_TYPEDEFS(int64_t, l)
OWL_VEC_TYPEDEFS(uint64_t, ul)
OWL_VEC_TYPEDEFS(int8_t, c)
OWL_VEC_TYPEDEFS(uint8_t, uc)
#undef OWL_VEC_TYPEDEFS

}  // namespace common
using namespace owl::common;
}  // namespace owl
