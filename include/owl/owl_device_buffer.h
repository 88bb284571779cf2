// owl_device_buffer.h -- what a variable of kind OWL_BUFFER looks like to device code.
// Same three fields and layout (4-byte type, padding, 8-byte count, 8-byte pointer = 24 bytes) as the
// reference's owl/include/owl/owl_device_buffer.h:27-33; filled by the host runtime when the variable
// struct is materialised (owlraytracing_amd/csrc/owl_runtime.cpp, rec::DeviceBufferVar).
#pragma once
#include "owl_host.h"

namespace owl {
namespace device {

struct Buffer {
  OWLDataType type;
  size_t count;
  void *data;
};

}  // namespace device
}  // namespace owl
