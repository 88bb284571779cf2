/* driver_types.h -- shim: cudaGraphicsResource_t for the (unsupported) graphics-interop prototype */
#ifndef OWL_SHIM_DRIVER_TYPES_H
#define OWL_SHIM_DRIVER_TYPES_H
#include <cuda.h>
typedef struct owlShimGraphicsResource *cudaGraphicsResource_t;
#endif
