// Rectilinear march / general tracer for atmospheres with Spline temperature functions (quadrature path of n(h) inlined).
#include "atmrt_march_impl.h"

namespace atmrt {
ATMRT_INSTANTIATE_MARCH(true)
} // namespace atmrt
