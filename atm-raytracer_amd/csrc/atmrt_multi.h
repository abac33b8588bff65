// atmrt_multi.h — internal interface between atmrt_api.hip (one device) and atmrt_multi.hip (several devices / ranks).
#pragma once

#include <functional>

#include "atmrt_ctx.h"

namespace atmrt {

// Pixel-column tiles: rank g of G owns columns [g W / G, (g + 1) W / G) (integer division, so widths differ by at most
// one column and any width can be sharded).
inline int shard_begin(int width, int rank, int world) { return (int)((int64_t)rank * width / world); }

// This context's tile of a frame `width` columns wide: the whole width without a comm.
void comm_columns(const atmrt_ctx* c, int width, int* c0, int* c1);

// A parent of several devices forwards a call to every child (on the child's own worker thread) and returns the first failure,
// whose message it adopts.
int multi_forward(atmrt_ctx* parent, const std::function<int(atmrt_ctx*)>& fn);
atmrt_ctx* multi_child(atmrt_ctx* parent, int i);
int multi_size(const atmrt_ctx* parent);
void multi_destroy(atmrt_ctx* parent); // joins the workers and destroys the children
void comm_destroy(atmrt_ctx* c);

int multi_generate(atmrt_ctx* parent, atmrt_result_t* out);
int multi_draw_image(atmrt_ctx* parent, const atmrt_coloring_t* coloring, uint8_t* rgb);
int multi_last_timings(atmrt_ctx* parent, atmrt_timings_t* out);
int multi_last_stats(atmrt_ctx* parent, atmrt_frame_stats_t* out);

// implemented in atmrt_api.hip, used by atmrt_multi.hip
int api_create_plain(atmrt_ctx** out, int device_ordinal);
int api_create_fail(int code, const std::string& msg); // records the message atmrt_last_error(NULL) returns
// One frame of this context's tile: first-hit planes (and, when `want_packed` or the frame has lists, the packed trace points in
// c->last_hits / c->last_offset) left in HBM; `dense` null = the context's own buffer.
int api_generate_tile(atmrt_ctx* c, const DensePlanes* dense, bool want_packed, uint64_t* n_hits, uint64_t* ray_steps, double* device_ms);

} // namespace atmrt
