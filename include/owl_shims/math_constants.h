/* math_constants.h -- shim for code that includes CUDA's constants header */
#ifndef OWL_SHIM_MATH_CONSTANTS_H
#define OWL_SHIM_MATH_CONSTANTS_H
#include <math.h>
#define CUDART_INF_F INFINITY
#define CUDART_INF ((double)INFINITY)
#define CUDART_PI_F 3.141592654f
#endif
