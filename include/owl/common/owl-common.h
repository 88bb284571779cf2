// owl-common.h -- host/device qualifiers and the small utilities shared by the owl math headers.
// MI355X build of the OWL programming surface: __both__ functions compile for the host (g++ or
// hipcc host pass) and, under hipcc, for gfx950.
#pragma once

#include <math.h>
#include <stdint.h>

#ifdef __cplusplus
#include <algorithm>
#include <cmath>
#include <iostream>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
#endif

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define __owl_device __device__
#define __owl_host __host__
#else
#define __owl_device
#define __owl_host
#endif
#define __both__ __owl_host __owl_device

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define OWL_NOTIMPLEMENTED throw std::runtime_error(std::string(__PRETTY_FUNCTION__) + " not implemented")

// ANSI colours used by the samples' LOG macros
#define OWL_TERMINAL_RED "\033[0;31m"
#define OWL_TERMINAL_GREEN "\033[0;32m"
#define OWL_TERMINAL_LIGHT_GREEN "\033[1;32m"
#define OWL_TERMINAL_YELLOW "\033[1;33m"
#define OWL_TERMINAL_BLUE "\033[0;34m"
#define OWL_TERMINAL_LIGHT_BLUE "\033[1;34m"
#define OWL_TERMINAL_MAGENTA "\033[0;35m"
#define OWL_TERMINAL_LIGHT_MAGENTA "\033[0;95m"
#define OWL_TERMINAL_CYAN "\033[0;36m"
#define OWL_TERMINAL_LIGHT_RED "\033[1;31m"
#define OWL_TERMINAL_BOLD "\033[1;1m"
#define OWL_TERMINAL_RESET "\033[0m"
#define OWL_TERMINAL_DEFAULT OWL_TERMINAL_RESET

#define OWL_ALIGN(n) __attribute__((aligned(n)))
#define MAYBE_UNUSED __attribute__((unused))

#ifdef __cplusplus
namespace owl {
namespace common {

template <typename T>
inline __both__ T divRoundUp(T a, T b) { return (a + b - 1) / b; }
template <typename T>
inline __both__ T clamp(T v, T lo, T hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline __both__ float saturate(float f) { return f < 0.f ? 0.f : (f > 1.f ? 1.f : f); }
inline __both__ float rcp(float f) { return 1.f / f; }
inline __both__ double rcp(double d) { return 1. / d; }

inline std::string prettyNumber(size_t s) {
  const char *unit[] = {"", "K", "M", "G", "T"};
  double v = (double)s;
  int u = 0;
  while (v >= 1000. && u < 4) {
    v /= 1000.;
    u++;
  }
  std::ostringstream o;
  o.precision(3);
  o << v << unit[u];
  return o.str();
}

}  // namespace common
}  // namespace owl
#endif
