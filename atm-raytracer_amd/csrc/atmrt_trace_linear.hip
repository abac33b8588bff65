// General tracer of the Rectilinear generator (scenes with objects) for atmospheres made of Linear temperature functions only.
#include "atmrt_march_impl.h"

namespace atmrt {
ATMRT_INSTANTIATE_TRACE(false)
} // namespace atmrt
