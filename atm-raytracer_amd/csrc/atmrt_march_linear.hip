// Rectilinear march / general tracer for atmospheres made of Linear temperature functions only (US-76 and most configs).
#include "atmrt_march_impl.h"

namespace atmrt {
ATMRT_INSTANTIATE_MARCH(false)

// the non-template entry points pick the variant by whether the compiled atmosphere has Spline (cubic) segments
extern template void launch_rect_march_t<true>(const Frame&, Workspace&, const DensePlanes&, hipStream_t, hipEvent_t);
extern template void launch_multi_fill_t<true>(const Frame&, Workspace&, uint64_t, const DensePlanes&, const PackedHits&, hipStream_t);
extern template void launch_rect_trace_count_t<true>(const Frame&, Workspace&, const DensePlanes&, hipStream_t);
extern template void launch_rect_trace_objects_t<true>(const Frame&, Workspace&, const DensePlanes&, uint64_t, hipStream_t);
extern template void launch_rect_trace_fill_t<true>(const Frame&, Workspace&, uint64_t, const DensePlanes&, const PackedHits&, hipStream_t);
extern template void launch_rect_trace_objects_t<false>(const Frame&, Workspace&, const DensePlanes&, uint64_t, hipStream_t); // atmrt_trace_linear.hip
extern template void launch_rect_trace_fill_t<false>(const Frame&, Workspace&, uint64_t, const DensePlanes&, const PackedHits&, hipStream_t);

void launch_rect_march(const Frame& f, Workspace& ws, const DensePlanes& out, hipStream_t stream, hipEvent_t ev_marched) {
  if (f.atm_cubic) launch_rect_march_t<true>(f, ws, out, stream, ev_marched);
  else launch_rect_march_t<false>(f, ws, out, stream, ev_marched);
}
void launch_multi_fill(const Frame& f, Workspace& ws, uint64_t n_hits, const DensePlanes& dense, const PackedHits& packed,
                       hipStream_t stream) {
  if (f.atm_cubic) launch_multi_fill_t<true>(f, ws, n_hits, dense, packed, stream);
  else launch_multi_fill_t<false>(f, ws, n_hits, dense, packed, stream);
}
void launch_rect_trace_count(const Frame& f, Workspace& ws, const DensePlanes& out, hipStream_t stream) {
  if (f.atm_cubic) launch_rect_trace_count_t<true>(f, ws, out, stream);
  else launch_rect_trace_count_t<false>(f, ws, out, stream);
}
void launch_rect_trace_objects(const Frame& f, Workspace& ws, const DensePlanes& out, uint64_t n_rays, hipStream_t stream) {
  if (f.atm_cubic) launch_rect_trace_objects_t<true>(f, ws, out, n_rays, stream);
  else launch_rect_trace_objects_t<false>(f, ws, out, n_rays, stream);
}
void launch_rect_trace_fill(const Frame& f, Workspace& ws, uint64_t n_hits, const DensePlanes& dense, const PackedHits& packed,
                            hipStream_t stream) {
  if (f.atm_cubic) launch_rect_trace_fill_t<true>(f, ws, n_hits, dense, packed, stream);
  else launch_rect_trace_fill_t<false>(f, ws, n_hits, dense, packed, stream);
}
} // namespace atmrt

#ifdef ATMRT_TIMELINE
extern "C" int atmrt_debug_timeline(unsigned long long* dst, size_t n_words) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(atmrt::g_timeline), n_words * sizeof(unsigned long long));
}
extern "C" int atmrt_debug_slices(unsigned long long* dst, size_t n_words) {
  int rc = (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(atmrt::g_slices), n_words * sizeof(unsigned long long));
  unsigned long long zero = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(atmrt::g_slices), &zero, sizeof zero);
  return rc;
}
#endif
