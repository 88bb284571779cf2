// owl.h -- umbrella header of the OWL programming surface on MI355X.
// Host translation units get the C-ABI (owl_host.h); translation units compiled by hipcc for
// gfx950 additionally get the device-side program model (owl_device.h), as the reference does for
// nvcc (owl/include/owl/owl.h:27-29).
#pragma once
#include "owl/owl_host.h"
#if defined(__HIPCC__) && defined(__cplusplus)
#include "owl/owl_device.h"
#endif
