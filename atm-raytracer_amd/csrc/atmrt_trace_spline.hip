// General tracer of the Rectilinear generator (scenes with objects) for atmospheres with Spline temperature functions.
#include "atmrt_march_impl.h"

namespace atmrt {
ATMRT_INSTANTIATE_TRACE(true)
} // namespace atmrt
