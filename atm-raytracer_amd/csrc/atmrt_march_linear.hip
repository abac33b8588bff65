// Rectilinear march / general tracer for atmospheres made of Linear temperature functions only (US-76 and most configs).
#include "atmrt_march_impl.h"

namespace atmrt {
ATMRT_INSTANTIATE_MARCH(false)

// the non-template entry points pick the variant by whether the compiled atmosphere has Spline (cubic) segments
extern template void launch_rect_march_t<true>(const Frame&, Workspace&, const DensePlanes&, hipStream_t, hipEvent_t);
extern template void launch_multi_fill_t<true>(const Frame&, Workspace&, uint64_t, const DensePlanes&, const PackedHits&, hipStream_t);
extern template void launch_rect_march3_t<true>(const Frame&, Workspace&, const DensePlanes&, hipStream_t);
extern template void launch_rect_trace_queue_t<true>(const Frame&, Workspace&, const DensePlanes&, hipStream_t);
extern template void launch_rect_trace_fill_t<true>(const Frame&, Workspace&, uint64_t, const DensePlanes&, const PackedHits&, hipStream_t);
extern template void launch_rect_trace_queue_t<false>(const Frame&, Workspace&, const DensePlanes&, hipStream_t); // atmrt_trace_linear.hip
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
void launch_rect_trace_count(const Frame& f, Workspace& ws, const DensePlanes& out, hipStream_t stream, hipStream_t stream2,
                             hipEvent_t ev_fork, hipEvent_t ev_join) {
  const size_t n = (size_t)f.wl * f.h;
  (void)hipMemsetAsync(ws.object_rays, 0xff, n * sizeof(uint32_t), stream); // QUEUE_EMPTY; counters [11], [14], [15] are zero (frame start)
  (void)hipEventRecord(ev_fork, stream);
  (void)hipStreamWaitEvent(stream2, ev_fork, 0);
  // The march first, its consumer second.  Two streams of a process may share a hardware queue, which runs its kernels in order: a
  // consumer ahead of the march it waits for would wait forever.  In this order no consumer precedes its own march in any queue,
  // and a cycle through another context's kernels would need one of them to have been submitted before itself.  (The consumer
  // still gives up after 30 s without its march ending — counters[12] — rather than hang the GPU.)
  if (f.atm_cubic) launch_rect_march3_t<true>(f, ws, out, stream);
  else launch_rect_march3_t<false>(f, ws, out, stream);
  if (f.atm_cubic) launch_rect_trace_queue_t<true>(f, ws, out, stream2);
  else launch_rect_trace_queue_t<false>(f, ws, out, stream2);
  (void)hipEventRecord(ev_join, stream2);
  (void)hipStreamWaitEvent(stream, ev_join, 0);
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
