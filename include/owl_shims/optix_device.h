/* optix_device.h -- shim: the device-side optix* intrinsics, implemented by the MI355X runtime
 * in owl/device_runtime.h (reference device code includes <optix_device.h> directly,
 * samples/s01-trueknn/deviceCode.cu:19). */
#ifndef OWL_SHIM_OPTIX_DEVICE_H
#define OWL_SHIM_OPTIX_DEVICE_H
#include <optix.h>
#if defined(__HIPCC__)
#include "owl/device_runtime.h"
#endif
#endif
