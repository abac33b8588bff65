/* atmrt.h — C ABI of the MI355X-native per-pixel ray-marching path of atm-raytracer.
 *
 * The reference has no FFI: its plug-in surface for this path is the Rust trait
 *     pub trait Generator { fn generate(&self) -> Vec<Vec<ResultPixel>>; }
 * (src/generator/generators/mod.rs:82-84), implemented by FastGenerator (fast.rs:21-108),
 * RectilinearGenerator (rectilinear.rs:23-76) and InterpolatingRectilinearGenerator
 * (interpolating_rectilinear.rs:110-162), each built from (&Params, &Terrain).  The entry points
 * below are what a Rust `impl Generator for HipGenerator` binds (INTEGRATION.md shows the stub):
 * plain pointers, sizes and #[repr(C)]-compatible PODs; all angles in DEGREES and all lengths in
 * METRES exactly as in the reference's `Params` (src/generator/params.rs:496-505).
 *
 * Error convention: every call returns 0 on success or a negative atmrt_status; the message is
 * available from atmrt_last_error().  The library never aborts the process (the reference
 * panics: terrain/mod.rs:45,71,117, params.rs:681-691; a shim may turn a non-zero status into
 * panic!/Err(String) to keep that behaviour).  A context is single-owner and not re-entrant.
 */
#ifndef ATMRT_H
#define ATMRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ATMRT_ABI_VERSION 5

typedef enum atmrt_status {
  ATMRT_OK = 0,
  ATMRT_ERR_INVALID_ARGUMENT = -1, /* bad pointer, size or enum value */
  ATMRT_ERR_NO_DEVICE = -2,        /* no gfx950 device / HIP runtime unavailable: never falls back to CPU */
  ATMRT_ERR_HIP = -3,              /* a HIP call or kernel failed (message carries hipGetErrorString) */
  ATMRT_ERR_IO = -4,               /* terrain directory or file unreadable (terrain/mod.rs:70-71) */
  ATMRT_ERR_FORMAT = -5,           /* a file in the terrain directory is not a DTED tile (terrain/mod.rs:113-118) */
  ATMRT_ERR_STATE = -6,            /* call order violated (e.g. generate before set_params) */
  ATMRT_ERR_UNSUPPORTED = -7       /* reserved; not returned since ABI 3: the fixed capacities of ABI 2 (12 trace points per step, 64
                                      corner points per interpolating pixel) now have unbounded routes */
} atmrt_status;

/* EarthModel, src/utils/earth_model/mod.rs:19-28 (same order as the Rust enum). */
typedef enum atmrt_earth_kind {
  ATMRT_EARTH_SIMPLE_SPHERE = 0,
  ATMRT_EARTH_SPHERICAL = 1,             /* uses .radius */
  ATMRT_EARTH_ELLIPSOID = 2,             /* uses .a, .b */
  ATMRT_EARTH_WGS84 = 3,
  ATMRT_EARTH_AZIMUTHAL_EQUIDISTANT = 4,
  ATMRT_EARTH_FLAT_DISTORTED = 5,
  ATMRT_EARTH_OBSERVER_AE = 6,           /* uses .radius as proj_radius */
  ATMRT_EARTH_SIMPLE_OBSERVER_AE = 7
} atmrt_earth_kind;

typedef struct atmrt_earth_model {
  int32_t kind; /* atmrt_earth_kind */
  int32_t _pad;
  double radius;
  double a;
  double b;
} atmrt_earth_model_t;

/* Altitude, params.rs:17-30. */
typedef enum atmrt_altitude_kind { ATMRT_ALT_ABSOLUTE = 0, ATMRT_ALT_RELATIVE = 1 } atmrt_altitude_kind;

/* Position, params.rs:32-40. */
typedef struct atmrt_position {
  double latitude;
  double longitude;
  int32_t altitude_kind; /* atmrt_altitude_kind */
  int32_t _pad;
  double altitude;
} atmrt_position_t;

/* Frame, params.rs:145-155. */
typedef struct atmrt_frame {
  double direction;
  double tilt;
  double fov;
  double max_distance;
} atmrt_frame_t;

/* GeneratorDef, params.rs:387-392 (same order). */
typedef enum atmrt_generator_kind {
  ATMRT_GEN_FAST = 0,
  ATMRT_GEN_INTERPOLATING_RECTILINEAR = 1,
  ATMRT_GEN_RECTILINEAR = 2
} atmrt_generator_kind;

/* The subset of `Params` (params.rs:496-505) that reaches the generators. */
typedef struct atmrt_params {
  atmrt_position_t position;   /* view.position */
  atmrt_frame_t frame;         /* view.frame */
  atmrt_earth_model_t earth;   /* model; env.shape is derived from it (earth_model/mod.rs:95-112) */
  double wavelength;           /* env.wavelength [m] */
  double simulation_step;      /* [m] */
  double terrain_alpha;        /* scene.terrain_alpha */
  int32_t straight_rays;       /* bool */
  int32_t generator;           /* atmrt_generator_kind, output.generator */
  uint16_t width;              /* output.width  (u16 as in params.rs:398-402) */
  uint16_t height;             /* output.height */
  uint16_t col_begin;          /* pixel-column shard [col_begin, col_end) computed by this context; 0,0 means the whole width. */
  uint16_t col_end;            /*   Leave 0,0 on a multi-device context / a rank context: the library assigns the tiles itself. */
} atmrt_params_t;

/* AtmosphereDef of crate atm-refraction 0.6 (schema: reference README.md:283-323): a pressure fixed point, a list of
 * temperature functions (the first from -inf, every next one from its `altitude` upwards), each either `Linear{gradient}`
 * or `Spline{boundary_condition, points}`, and — when every function is Linear — a temperature fixed point.  Both lists are
 * `Vec`s in the reference (params.rs:453-454) and unbounded here: pointer + count, borrowed for the duration of the call that
 * takes the definition (atmrt_set_atmosphere copies what it needs). */
typedef enum atmrt_temp_function_kind { ATMRT_TEMP_LINEAR = 0, ATMRT_TEMP_SPLINE = 1 } atmrt_temp_function_kind;
typedef enum atmrt_spline_boundary {
  ATMRT_SPLINE_NATURAL = 0,            /* second derivative 0 at both ends */
  ATMRT_SPLINE_DERIVATIVES = 1,        /* first derivatives bc[0], bc[1] at the ends */
  ATMRT_SPLINE_SECOND_DERIVATIVES = 2  /* second derivatives bc[0], bc[1] at the ends */
} atmrt_spline_boundary;
typedef struct atmrt_temp_function {
  int32_t kind;      /* atmrt_temp_function_kind */
  int32_t boundary;  /* enum atmrt_spline_boundary; Spline only */
  double altitude;   /* applies for h >= altitude; ignored for the first function */
  double gradient;   /* Linear: dT/dh [K/m] */
  double bc[2];      /* Spline boundary values */
  int32_t n_points;  /* Spline: >= 2, strictly increasing altitudes */
  int32_t _pad;
  const double* point_altitude;    /* [n_points] */
  const double* point_temperature; /* [n_points] */
} atmrt_temp_function_t;
typedef struct atmrt_atmosphere {
  double pressure_altitude;          /* pressure fixed point */
  double pressure;                   /* [Pa] */
  double temperature_altitude;       /* temperature_fixed_point (used only when has_temperature_fixed_point) */
  double temperature;                /* [K] */
  int32_t has_temperature_fixed_point;
  int32_t n_functions;               /* >= 1 */
  const atmrt_temp_function_t* functions; /* [n_functions] */
} atmrt_atmosphere_t;

/* Scene objects, src/object/mod.rs:19-75,119-131 after ConfShape::into_shape. */
typedef enum atmrt_object_kind { ATMRT_OBJ_FRUSTUM = 0, ATMRT_OBJ_BILLBOARD = 1 } atmrt_object_kind;
typedef struct atmrt_object {
  int32_t kind; /* atmrt_object_kind; Cylinder = Frustum{r1=r2}, Cone = Frustum{r2=0} (object/mod.rs:44-54) */
  int32_t _pad;
  atmrt_position_t position;
  double r1, r2;       /* frustum bottom / top radius */
  double height;       /* frustum or billboard height */
  double width;        /* billboard width */
  double color[4];     /* frustum RGBA in [0,1]; alpha defaults to 1.0 in the YAML (object/mod.rs:140-146) */
  const uint8_t* texture_rgba; /* billboard texture, row-major top row first, 4 bytes per texel; borrowed during the call */
  uint32_t texture_width;
  uint32_t texture_height;
} atmrt_object_t;

/* PixelColor tag of a trace point, generators/mod.rs:45-49. */
typedef enum atmrt_color_tag { ATMRT_COLOR_TERRAIN = 0, ATMRT_COLOR_RGBA = 1 } atmrt_color_tag;

/* Vec<Vec<ResultPixel>> (generators/mod.rs:13-30) flattened to structure-of-arrays.
 * Pixel p = y * width + (x - col_begin), row-major like result[y][x] (fast.rs:52-92).
 * Trace points of pixel p are hits [hit_offset[p], hit_offset[p] + hit_count[p]) in march order. */
typedef struct atmrt_result {
  uint32_t width;  /* columns in this shard */
  uint32_t height;
  uint64_t n_pixels;
  uint64_t n_hits;
  double* azimuth;         /* [n_pixels] degrees */
  double* elevation_angle; /* [n_pixels] degrees */
  uint32_t* hit_count;     /* [n_pixels] */
  uint64_t* hit_offset;    /* [n_pixels] */
  double* lat;             /* [n_hits] TracePoint.lat */
  double* lon;
  double* distance;
  double* elevation;
  double* path_length;
  double* normal;          /* [n_hits][3] */
  uint32_t* color_tag;     /* [n_hits] atmrt_color_tag */
  double* rgba;            /* [n_hits][4]; Terrain(alpha) -> {0,0,0,alpha} */
  uint64_t ray_steps;      /* sample pairs examined under the reference's termination rule (utils.rs:211-287) */
  double device_ms;        /* device time of the generate call, HIP events */
} atmrt_result_t;

/* Device-resident first-hit planes of one shard, written by atmrt_generate_device (caller-owned
 * device memory, e.g. torch tensors; the library only writes them).  Used by bench.py and by the
 * multi-GPU path so that results stay in HBM for the RCCL all-gather. */
typedef struct atmrt_device_planes {
  double* azimuth;         /* [H][Wshard] */
  double* elevation_angle;
  uint32_t* hit_count;     /* total trace points of the pixel */
  double* lat;             /* first trace point; NaN where hit_count == 0 */
  double* lon;
  double* distance;
  double* elevation;
  double* path_length;
  double* normal;          /* [3][H][Wshard] planar */
} atmrt_device_planes_t;

typedef struct atmrt_ctx atmrt_ctx;

/* ---- lifetime ---------------------------------------------------------------------------- */
int atmrt_abi_version(void);
/* "source_hash: <sha256/16 of csrc + Makefile + this header>; march_units: <flags>; calling_units: <flags>; all: <flags>; arch: gfx950":
 * what the library was built from and with which code-generation flags (the calling units must carry -enable-ipra=0). */
const char* atmrt_build_info(void);
/* device_ordinal: HIP device index (LOCAL_RANK under torchrun).  Fails with ATMRT_ERR_NO_DEVICE
 * when no GPU is present — there is no CPU path in this library. */
int atmrt_ctx_create(atmrt_ctx** out, int device_ordinal);
void atmrt_ctx_destroy(atmrt_ctx* ctx);
/* Message of the last failing call on ctx (or of the last failing atmrt_ctx_create when ctx == NULL). */
const char* atmrt_last_error(const atmrt_ctx* ctx);

/* ---- terrain: replaces Terrain::{from_folder,get_elev} (terrain/mod.rs:55-126) + crate dted 0.2 */
/* Scan a terrain directory (Terrain::from_folder, terrain/mod.rs:66-118): every entry must be a DTED file or be named like a
 * GeoTIFF tile ((N|S)dd(E|W)ddd, 16-bit single band, at least 3601 x 3601 samples); anything else fails as the reference
 * panics.  A GeoTIFF that cannot be decoded leaves its cell without terrain, like the reference's lazy load. */
int atmrt_terrain_load_dir(atmrt_ctx* ctx, const char* path, int32_t* n_files);
/* Register one 1-degree cell directly.  posts: n_lat rows (south to north) of n_lon posts (west to east). */
int atmrt_terrain_add_tile(atmrt_ctx* ctx, int32_t lat0, int32_t lon0, int32_t n_lat, int32_t n_lon,
                           const int16_t* posts);
int atmrt_terrain_clear(atmrt_ctx* ctx);
/* Batched Terrain::get_elev on the device: valid[i] = 0 where the reference returns None. */
int atmrt_terrain_get_elev(atmrt_ctx* ctx, size_t n, const double* lat, const double* lon, double* elev,
                           uint8_t* valid);

/* ---- configuration ------------------------------------------------------------------------ */
void atmrt_params_default(atmrt_params_t* p);         /* Config::default, params.rs:481-494 */
void atmrt_atmosphere_us76(atmrt_atmosphere_t* a);    /* AtmosphereDef::us_76; `functions` points at a table inside the library */
int atmrt_set_params(atmrt_ctx* ctx, const atmrt_params_t* p);
int atmrt_set_atmosphere(atmrt_ctx* ctx, const atmrt_atmosphere_t* a);
/* scene.objects, in order (object/mod.rs:156-190).  Billboard textures are copied during the call. */
int atmrt_objects_set(atmrt_ctx* ctx, const atmrt_object_t* objects, size_t n);

/* ---- the path ----------------------------------------------------------------------------- */
/* Generator::generate for the generator named in params (generators/mod.rs:82-84): Fast (fast.rs:22-98), Rectilinear
 * (rectilinear.rs:24-60) or InterpolatingRectilinear (interpolating_rectilinear.rs:110-162), with or without scene objects and
 * for any terrain_alpha.  The result is library-allocated host memory — one page-locked block the arrays point into, so the
 * device-to-host copy runs at PCIe speed; release it (as a whole) with atmrt_result_free, which keeps the block for the next frame. */
int atmrt_generate(atmrt_ctx* ctx, atmrt_result_t* out);
void atmrt_result_free(atmrt_result_t* r);
/* Same computation, results left in HBM in caller-provided planes; ray_steps/device_ms optional.  The planes must stay
 * allocated until the next generate call on ctx or until the last atmrt_draw_image* / atmrt_last_hits_device call for this
 * frame, whichever comes first: those read the first-hit planes of an opaque frame in place. */
int atmrt_generate_device(atmrt_ctx* ctx, const atmrt_device_planes_t* planes, uint64_t* ray_steps,
                          double* device_ms);

/* The complete trace-point lists of the frame atmrt_generate_device just produced (translucent terrain, scenes with objects,
 * InterpolatingRectilinear), copied device-to-device into caller-owned arrays laid out like the hit arrays of atmrt_result_t:
 * what a multi-GPU host gathers after the hit_count planes.  dst == NULL only queries *n_hits.  An opaque frame has
 * no list beyond its planes: ATMRT_ERR_STATE. */
typedef struct atmrt_device_hits {
  uint64_t capacity;     /* entries each hit array can hold */
  uint64_t* hit_offset;  /* [H][Wshard] index of each pixel's first trace point */
  double* lat;
  double* lon;
  double* distance;
  double* elevation;
  double* path_length;
  double* normal;        /* [n][3] */
  uint32_t* color_tag;
  double* rgba;          /* [n][4] */
} atmrt_device_hits_t;
int atmrt_last_hits_device(atmrt_ctx* ctx, const atmrt_device_hits_t* dst, uint64_t* n_hits);

/* Device time of each phase of the last atmrt_generate / atmrt_generate_device call, measured with HIP
 * events recorded on the library's own streams (a caller's events on another stream cannot see them). */
typedef struct atmrt_timings {
  double total_ms;     /* first launch to last launch of the call */
  double profile_ms;   /* Fast phase A: k_fast_columns + k_terrain_profile   (utils.rs:176-199) */
  double paths_ms;     /* Fast phase B: k_fast_paths, concurrent with phase A (utils.rs:136-174) */
  double intersect_ms; /* Fast phase C: k_fast_intersect, summed over its segments (they run while later path segments of phase B are integrated) (utils.rs:201-289) */
  double march_ms;     /* Rectilinear: k_rect_march                           (rectilinear.rs:161-185) */
  double finalize_ms;  /* trace-point epilogue: k_fast_finalize / k_rect_finalize */
  double pack_ms;      /* scan + packing / multi-hit fill, when requested */
  uint64_t ray_steps;
  uint64_t n_hits;
} atmrt_timings_t;
int atmrt_last_timings(atmrt_ctx* ctx, atmrt_timings_t* out);

/* How often the last frame left the fast routes of the device path (results are the same either way; tests use this to
 * prove that a workload really exercises the fall-back routes). */
typedef struct atmrt_frame_stats {
  uint64_t unlisted_rays;     /* Rectilinear, scenes with objects: rays with more candidate objects than the per-ray list
                                 holds (24) — they test every object at every sample (is_close, frustum.rs:103-114) */
  uint64_t unlisted_columns;  /* Fast / InterpolatingRectilinear: columns with more candidates than the per-column list (64) */
  uint64_t retraced_pixels;   /* Rectilinear: pixels with more trace points than the 4 slots of the counting pass: their further
                                 points come out of an overflow arena (marched / traced a second time only when the arena is full or
                                 big_steps > 0) */
  uint64_t big_steps;         /* steps that produced more trace points than the in-register step list (12): sorted in HBM */
  uint64_t big_blend_pixels;  /* InterpolatingRectilinear: pixels whose four lattice corners hold more than 4 trace points together
                                 (blended over a member arena in HBM; any number of points) */
  uint64_t terrain_lookups;   /* Rectilinear march: Terrain::get_elev evaluations actually performed.  A sample whose ray is
                                 above the highest post of the mosaic (+ 1 m) is above the terrain for certain, so its geodesic
                                 point and lookup are not evaluated — every ODE step, sign test and result stays the same;
                                 ray_steps keeps counting every step */
  uint64_t object_rays;       /* Rectilinear, scenes with objects: rays left to the general tracer — since round 4 only those of
                                 wavefronts with more candidate objects than the wavefront's list holds (96), or of an earth model
                                 without the geometric pre-filter; every other ray stays with the lean march */
  uint64_t object_steps;      /* Rectilinear, scenes with objects: ray-steps the lean march handed to its out-of-line object step
                                 (the step lies inside a candidate object's distance interval and enters its height band) */
} atmrt_frame_stats_t;
int atmrt_last_stats(atmrt_ctx* ctx, atmrt_frame_stats_t* out);
/* Fault injection for tests of the error paths: the next atmrt_generate / atmrt_generate_device on ctx runs its kernels and then
 * fails with ATMRT_ERR_HIP (once).  After ANY failed frame atmrt_draw_image* and atmrt_last_hits_device return ATMRT_ERR_STATE
 * until a frame succeeds: the failed frame has already reused the buffers of the one before it. */
int atmrt_debug_fail_next_frame(atmrt_ctx* ctx);
/* How the Rectilinear march of a frame (or column tile) of width x height pixels with `samples` terrain samples per ray
 * (ceil(max_distance / simulation_step)) is planned; no device needed.  out[0] = 1 when the time-sliced march is used (tiles of at
 * most 4 Mpixel without scene objects), out[1] = ray groups, out[2] = the FIFO's capacity in entries, out[3] = bytes
 * of slice state, out[4] = the bound on the slices a ray can need after the first, out[5] = steps per slice. */
int atmrt_debug_march_plan(int32_t width, int32_t height, int32_t samples, int32_t n_objects, uint64_t out[6]);

/* ---- renderer compositing + colouring on the device (src/renderer/mod.rs:367-414, src/coloring) -- */
typedef enum atmrt_coloring_kind { ATMRT_COLORING_SIMPLE = 0, ATMRT_COLORING_SHADING = 1 } atmrt_coloring_kind;
typedef enum atmrt_palette { ATMRT_PALETTE_LEGACY = 0, ATMRT_PALETTE_IMPROVED = 1 } atmrt_palette;
/* `Coloring` (params.rs:216-229) + view.fog_distance (params.rs:305). */
typedef struct atmrt_coloring {
  int32_t kind;          /* atmrt_coloring_kind */
  int32_t palette;       /* enum atmrt_palette; Shading only */
  double water_level;
  double max_distance;   /* Simple: frame.max_distance */
  double ambient_light;  /* Shading */
  double light_dir[3];   /* Shading: unit vector in the world frame */
  int32_t has_fog;       /* view.fog_distance.is_some() */
  int32_t _pad;
  double fog_distance;
} atmrt_coloring_t;
/* ConfColoring::into_coloring (params.rs:231-277): light_zenith_angle / light_dir in degrees, relative to view.frame.direction. */
int atmrt_coloring_from_conf(const atmrt_params_t* params, int32_t kind, double water_level, double ambient_light,
                             double light_zenith_angle, double light_dir, int32_t palette, int32_t has_fog,
                             double fog_distance, atmrt_coloring_t* out);
/* renderer::draw_image for the frame of the last atmrt_generate / atmrt_generate_device call on this context (its trace
 * points are still in HBM): rgb is host memory [height][width][3] of the shard, row-major like ImageBuffer. */
int atmrt_draw_image(atmrt_ctx* ctx, const atmrt_coloring_t* coloring, uint8_t* rgb);
/* Same, into caller-provided device memory (3 B per pixel instead of 88 B per pixel to gather across GPUs). */
int atmrt_draw_image_device(atmrt_ctx* ctx, const atmrt_coloring_t* coloring, uint8_t* rgb_device);

/* ---- several GPUs of one node --------------------------------------------------------------------------------
 * The reference calls `generator.generate()` ONCE per frame (src/generator/mod.rs:72-86, trait at generators/mod.rs:82-84), so the
 * multi-GPU path lives BELOW this ABI: pixels are independent (rectilinear.rs:32-37), the image is cut into pixel-column tiles —
 * device / rank g of G computes columns [g W / G, (g + 1) W / G) of every row against its own copy of the terrain mosaic — and the
 * only exchange is the finished frame.  Two ways to get there, same code underneath:
 *
 *  (1) ONE PROCESS, SEVERAL DEVICES: atmrt_ctx_create_multi(devices, n) returns a context that every entry point of this header
 *      accepts.  Each device gets a sub-context and a host thread.  atmrt_generate returns the whole [H][W] frame in host memory:
 *      every device copies its tile straight into the one page-locked block (strided device-to-host copies over its own PCIe
 *      link; no collective, the consumer being the host).  atmrt_generate_image_device leaves the whole frame in the HBM of EVERY
 *      device: one ncclAllGather (RCCL over xGMI) of the tiles' planes + a permutation kernel into the [H][W] planes.
 *  (2) ONE PROCESS PER GPU (torchrun, MPI): every rank creates a plain context, rank 0 obtains atmrt_comm_unique_id and hands it to
 *      the others through whatever the launcher offers, every rank calls atmrt_ctx_comm_init_rank (which also runs the
 *      communicator's first collective, a probe with a time-out, so that a fabric that does not work is an error HERE, where the
 *      host can still take atmrt_ctx_comm_init_external*); atmrt_generate_image_device is then collective over the ranks.  atmrt_generate / atmrt_generate_device on such a context return the rank's own tile.
 *
 * Frames whose pixels hold several trace points (terrain_alpha < 1, scene objects, InterpolatingRectilinear) also exchange the
 * variable-length lists: atmrt_image_hits_device (count -> scan -> offset on the device, one all-gather of the lists).
 * RCCL is resolved with dlopen("librccl.so.1") on first use: a process that already carries one under that SONAME (PyTorch's
 * bundled librccl) shares it — one RCCL, one HIP runtime per process — any other finds ROCm's through the library's RUNPATH. */
#define ATMRT_COMM_ID_BYTES 128
/* ncclGetUniqueId: 128 opaque bytes for the other ranks' atmrt_ctx_comm_init_rank. */
int atmrt_comm_unique_id(uint8_t id[ATMRT_COMM_ID_BYTES]);
/* Joins this context (one device) to `world` ranks that share every frame from now on: ncclCommInitRank, collective over the
 * ranks.  Parameters keep describing the WHOLE image (col_begin = col_end = 0). */
int atmrt_ctx_comm_init_rank(atmrt_ctx* ctx, const uint8_t id[ATMRT_COMM_ID_BYTES], int32_t rank, int32_t world);
/* The same with the host's own transport in place of RCCL (an MPI all-gather without GPU awareness, a shared-memory ring, gloo,
 * a test double): the callback must deliver, on every rank, rank i's `bytes_per_rank` bytes at recv_host + i * bytes_per_rank and
 * return 0.  Both pointers are page-locked HOST memory owned by the library, which stages the tile out of and the gathered
 * tiles back into HBM around the call. */
typedef int (*atmrt_all_gather_fn)(void* user, const void* send_host, void* recv_host, size_t bytes_per_rank);
int atmrt_ctx_comm_init_external(atmrt_ctx* ctx, int32_t rank, int32_t world, atmrt_all_gather_fn all_gather, void* user);
/* The same for a transport that moves DEVICE memory itself (a GPU-aware MPI, another RCCL communicator the host already owns):
 * both pointers are in this context's HBM; the library's stream has drained when the callback is entered and the gathered bytes
 * must be in place when it returns. */
typedef int (*atmrt_all_gather_device_fn)(void* user, const void* send_device, void* recv_device, size_t bytes_per_rank);
int atmrt_ctx_comm_init_external_device(atmrt_ctx* ctx, int32_t rank, int32_t world, atmrt_all_gather_device_fn all_gather, void* user);
/* One context over n_devices HIP devices of this process (1): terrain, parameters, atmosphere and objects set on it reach every
 * device.  A device may be listed more than once (its tiles then are exchanged by device-to-device copies; RCCL needs distinct
 * devices).  ATMRT_GATHER=peer forces that route, ATMRT_GATHER=rccl makes a failure to set RCCL up an error instead of a fallback. */
int atmrt_ctx_create_multi(atmrt_ctx** out, const int32_t* devices, int32_t n_devices);
/* 1 for a plain context, n_devices for a multi-device one. */
int atmrt_ctx_device_count(const atmrt_ctx* ctx);
/* Generator::generate with the WHOLE frame left in HBM: `image` holds one set of [H][W] planes per device of the context (a plain
 * or rank context: one; a multi-device context: device_count sets, entry i in the memory of devices[i]; a set whose azimuth
 * pointer is NULL is skipped).  ray_steps: of this rank (rank context) / of all devices (multi-device context). */
int atmrt_generate_image_device(atmrt_ctx* ctx, const atmrt_device_planes_t* image, uint64_t* ray_steps, double* device_ms);
/* The complete trace-point lists of that frame in the image's pixel order p = y W + x — atmrt_last_hits_device for the whole
 * image — on every device: `dst` like `image` above (entry i on devices[i]; hit_offset is [H][W]).
 *   dst == NULL: only *n_hits, the image's total — every rank has known it since the frame's own collective (each tile's slab
 *     carries its count): no communication, any rank may ask at any time.
 *   dst != NULL: COLLECTIVE — every rank of the frame (every device of a multi-device context: one entry each) must make the
 *     call; ONE all-gather of the packed lists, then count -> scan -> offset on the device.  A rank (or device entry) that wants
 *     nothing for itself passes hit_offset == NULL and still takes part.  A rank's own mistake (capacity < *n_hits, a NULL array
 *     while the image has trace points, no image planes assembled on it) is reported to that rank AFTER it has taken part, so
 *     the others never wait for it.  An image without a single trace point has no collective at all: hit_offset is zero-filled,
 *     the list arrays are not touched and may be NULL.
 * Frames without lists (opaque terrain, no objects, not InterpolatingRectilinear): ATMRT_ERR_STATE on every rank. */
int atmrt_image_hits_device(atmrt_ctx* ctx, const atmrt_device_hits_t* dst, uint64_t* n_hits);
/* renderer::draw_image of every tile + an all-gather of the 3 B/pixel RGB8 tiles instead of the 84 B/pixel planes: rgb_device[i]
 * is [H][W][3] on devices[i] (NULL entries skipped). */
int atmrt_draw_image_gathered_device(atmrt_ctx* ctx, const atmrt_coloring_t* coloring, uint8_t* const* rgb_device);
/* What the exchange of the last atmrt_generate_image_device / atmrt_generate (multi-device) cost, from HIP events on the
 * library's streams (slowest device). */
typedef enum atmrt_gather_route {
  ATMRT_ROUTE_NONE = 0,     /* one device: nothing to exchange */
  ATMRT_ROUTE_RCCL = 1,     /* ncclAllGather over xGMI */
  ATMRT_ROUTE_PEER = 2,     /* device-to-device copies inside one process */
  ATMRT_ROUTE_EXTERNAL = 3, /* the host's transport (atmrt_ctx_comm_init_external) */
  ATMRT_ROUTE_HOST = 4,     /* atmrt_generate on a multi-device context: strided copies into the host block */
  ATMRT_ROUTE_EXTERNAL_DEVICE = 5 /* the host's device-memory transport (atmrt_ctx_comm_init_external_device) */
} atmrt_gather_route;
typedef struct atmrt_comm_timings {
  double gather_ms;         /* the collective (or the copies) */
  double assemble_ms;       /* permutation of the gathered tiles into the [H][W] planes (route HOST: the merge of the lists on the host threads, wall clock) */
  double tile_ms_max;       /* slowest device's generate time */
  double tile_ms_min;       /* fastest device's */
  uint64_t bytes_per_rank;  /* what each rank contributed to the collective */
  int32_t world;
  int32_t route;            /* atmrt_gather_route */
  int32_t collectives;      /* data-path collectives of the last frame (1 for an image, + 1 for its lists) */
  int32_t _pad;
} atmrt_comm_timings_t;
int atmrt_last_comm_timings(atmrt_ctx* ctx, atmrt_comm_timings_t* out);
/* 1 when RCCL can be loaded in this process (no device work, not collective): what every rank of a launcher checks — and agrees
 * on — BEFORE any of them enters the collective atmrt_ctx_comm_init_rank. */
int atmrt_comm_available(void);
/* The pixel columns [*col_begin, *col_end) of a tile: of the last frame that was exchanged, else of the next one.  A multi-device
 * context: of device `index`; a rank context: of rank `index` (index < 0: its own); a plain context: its whole width or shard.
 * Tiles start equal (rank g of G: [g W / G, (g + 1) W / G), inner boundaries on multiples of 64 columns when tiles are at least 128
 * wide) and are RE-CUT by the library after a frame whose slowest tile took
 * more than 1 % longer than the mean — every rank from the same gathered tile times, so all agree without a message
 * (ATMRT_TILE_BALANCE=0 keeps them equal; 1 forces re-cutting where it is off by default: devices listed twice, the host-buffer
 * transport). */
int atmrt_ctx_tile_columns(atmrt_ctx* ctx, int32_t index, int32_t* col_begin, int32_t* col_end);
/* The re-cutting rule itself, a pure function: cols (n_tiles + 1 ascending boundaries, 0 .. width) and each tile's time ->
 * boundaries that would have equalised the times had the cost per column been constant inside each tile, on multiples of 64 columns
 * (a wavefront of the marching kernels is 64 consecutive pixels: tiles whose width is not a multiple of 64 march 15 % slower) when
 * every tile is at least 128 wide; cols_out = cols when the re-cut would not shorten the slowest tile by 1 % under that model. */
int atmrt_tiles_rebalance(int32_t width, int32_t n_tiles, const int32_t* cols, const double* tile_ms, int32_t* cols_out);
/* Test hooks.  set_tiling: the next frames use exactly these n = world + 1 boundaries (NULL: back to the library's own); on a
 * rank context every rank must be given the same.  fail_next_collective: the nth data-path collective from now (1 = the next)
 * fails on device `index` of a multi-device context / on this rank — before it is enqueued, as a refused ncclAllGather would. */
int atmrt_debug_set_tiling(atmrt_ctx* ctx, const int32_t* cols, int32_t n);
int atmrt_debug_fail_next_collective(atmrt_ctx* ctx, int32_t index, int32_t nth);

/* ---- integrator / sampler harnesses (the reference's diagnostic subcommands) ---------------- */
/* output-ray-paths (src/ray_path.rs:65-103): for each elevation angle [deg] step the ray n_steps
 * times from height h0 with `step` metres; x and h are [n_angles][n_steps+1] including the start. */
int atmrt_ray_paths(atmrt_ctx* ctx, double h0, size_t n_angles, const double* angles_deg, int32_t straight,
                    double step, size_t n_steps, double* x, double* h);
/* output-atm (src/atm_printer.rs:37-46): T [K], p [Pa], refractive index n and dn/dh at altitudes. */
int atmrt_atmosphere_sample(atmrt_ctx* ctx, size_t n, const double* altitude, double* temperature,
                            double* pressure, double* n_index, double* dn_dh);
/* DirectionalCalc::coords_at_dist (directional_calc.rs:5-7) for the context's earth model. */
int atmrt_coords_at_dist(atmrt_ctx* ctx, double lat0, double lon0, double dir_deg, size_t n, const double* dist,
                         double* lat, double* lon);

/* ---- the metadata file (src/generator/mod.rs:20-45, read back by src/viewer/mod.rs:17-29) -------- */
/* bincode-1 encoding of `result: Vec<Vec<ResultPixel>>` exactly as serde derives it (generators/mod.rs:13-49): u64 lengths,
 * u32 enum tags, f64 little-endian; layout in csrc/atmrt_metadata.hip.  vector3_len_prefix != 0 writes nalgebra's Vector3 as a
 * sequence (u64 3 + 3 f64, what nalgebra 0.32's ArrayStorage serializer is believed to emit; crate absent, UNPINNED), 0 as
 * three bare f64.  Host-only functions (no device work, callable without a GPU).  dst == NULL queries *n_bytes. */
int atmrt_result_encode_bincode(const atmrt_result_t* r, int32_t vector3_len_prefix, uint8_t* dst, size_t capacity,
                                size_t* n_bytes);
/* The inverse: `out` is library-allocated (release with atmrt_result_free); *consumed = bytes read from src.  Rows of unequal
 * length, an unknown PixelColor tag or a truncated buffer give ATMRT_ERR_FORMAT. */
int atmrt_result_decode_bincode(const uint8_t* src, size_t n_bytes, int32_t vector3_len_prefix, atmrt_result_t* out,
                                size_t* consumed);

/* The deterministic elementary functions of the device path (csrc/detmath.h), element-wise on host arrays: the same
 * instruction sequences the marching kernels execute.  The bit-exactness claim of this library rests on them returning, on
 * gfx950, exactly what the host build of detmath.h returns (and, for DIV / DIV_R / SQRT_INRANGE, what IEEE division and square
 * root return inside their documented operand range); tests/test_gpu_detmath.py checks that on 1e7 operands per function.
 * b may be NULL for one-operand functions, out1 may be NULL unless op is SINCOS or POW3. */
typedef enum atmrt_math_probe_op {
  ATMRT_PROBE_DIV = 0,          /* dm_div(a, b): division without the range scaling / fix-up steps */
  ATMRT_PROBE_DIV_R = 1,        /* dm_div_r(a, b, RN(1/b)): division by a tabulated reciprocal */
  ATMRT_PROBE_SQRT_INRANGE = 2, /* dm_sqrt_inrange(a) */
  ATMRT_PROBE_EXP = 3,
  ATMRT_PROBE_LOG = 4,
  ATMRT_PROBE_POW = 5,          /* dm_pow(a, b) */
  ATMRT_PROBE_SINCOS = 6,       /* out0 = sin, out1 = cos */
  ATMRT_PROBE_ASIN = 7,
  ATMRT_PROBE_ATAN2 = 8,        /* dm_atan2(a, b) */
  ATMRT_PROBE_IEEE_DIV = 9,     /* the compiler's a / b */
  ATMRT_PROBE_IEEE_SQRT = 10,   /* the compiler's sqrt(a) */
  ATMRT_PROBE_ATAN = 11,
  ATMRT_PROBE_TAN = 12,
  ATMRT_PROBE_POW3 = 13,        /* out0 = dm_pow(a, b) through the three-point form of the stepping kernels; out1 = sum of
                                   the other two points (a * 0.99999981, a * 1.00000019) */
  ATMRT_PROBE_DIV3 = 14,        /* dm_div3: out0 = a / (b (1 - 2^-22)), out1 = a / (b (1 + 2^-21)), both with their reciprocal seeded
                                   from b's; the centre quotient a / b is ATMRT_PROBE_DIV's */
  ATMRT_PROBE_DIV3_SEEDED = 15, /* dm_div3_seeded without a seed for b (the p / T site of a tight atmosphere segment): the DIV3 outputs
                                   for divisors b (1 -+ 2^-22), no vote */
  ATMRT_PROBE_DIV3_SEED_Z = 16, /* dm_div3_seeded with b = Z close to 1 (|1 - Z| <= 2^-10.5) and 2 - Z as the seed of its reciprocal:
                                   out0 = a / b, out1 = a / (b (1 + 2^-22)) */
  ATMRT_PROBE_DIV_SEED_N = 17,  /* dm_div_seeded(a, 1 + b, 1 - b) for 0 <= b <= 2^-10.5: a / n with n = 1 + (n - 1) */
  ATMRT_PROBE_POW3_SHARED = 18  /* the three-point pow of a TIGHT segment (shared table rows, csrc/detmath.h): out0 = dm_pow(a, b),
                                   out1 = dm_pow(a (1 - 2^-22), b) + dm_pow(a (1 + 2^-22), b) */
} atmrt_math_probe_op;
int atmrt_math_probe(atmrt_ctx* ctx, int32_t op, size_t n, const double* a, const double* b, double* out0, double* out1);

#ifdef __cplusplus
}
#endif
#endif /* ATMRT_H */
