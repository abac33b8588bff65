/* generators.c — shared pixel tracer + Fast / Rectilinear generators.  ORACLE (test infrastructure).
 * Line-by-line restatement of src/generator/generators/{utils,fast,rectilinear}.rs.  It does the
 * reference's work in the reference's order (eager normals and object filter per terrain sample,
 * per-column / per-row caches), which is also what bench.py times as the CPU baseline.
 */
#include "oracle_internal.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

const char* oracle_flavour(void) { return OM_FLAVOUR; }

/* Checker-only economy: compute only the rows y with y % stride == phase of the next frames (Fast and Rectilinear; the other
 * rows stay zeroed: no trace points, 0 ray-steps).  Full-size frames with a thousand objects cost the reference's eager
 * per-sample object filter (utils.rs:74-80) ~0.3 ms per sample; tests sample rows instead of shrinking the frame. */
static int g_row_stride = 1, g_row_phase = 0;
void oracle_set_row_filter(int stride, int phase) {
  g_row_stride = stride > 0 ? stride : 1;
  g_row_phase = phase;
}
static int row_selected(int y) { return g_row_stride == 1 || y % g_row_stride == g_row_phase; }

/* ---- utils.rs ----------------------------------------------------------------------------- */

typedef struct { double dist, elev, path_length; } path_elem; /* utils.rs:55-60 */

typedef struct { /* utils.rs:62-69 */
  double lat, lon, elev;
  ovec3 normal;
  int n_close;
  int* close; /* objects_close, ascending indices */
} terrain_data;

typedef struct {
  const atmrt_params_t* params;
  const oracle_terrain* terrain;
  const oracle_object* objects;
  size_t n_objects;
  oracle_env_atm atm;
  int spherical;
  double radius;
  double alt; /* Altitude::abs of the observer, params.rs:23-30 */
} gen_ctx;

static double elev_or_zero(const oracle_terrain* t, double lat, double lon) {
  double e;
  return oracle_terrain_get_elev(t, lat, lon, &e) ? e : 0.0; /* .unwrap_or(0.0) */
}

/* find_normal, utils.rs:15-40 */
ovec3 oracle_find_normal(const atmrt_earth_model_t* model, double lat, double lon, const oracle_terrain* terrain) {
  const double DIFF = 15.0;
  oracle_dircalc ns_calc, ew_calc;
  double nlat, nlon, slat, slon, elat, elon, wlat, wlon, diff_ew, diff_ns, len;
  ovec3 dir_north, dir_east, dir_up, vec_ns, vec_ew, normal;
  oracle_dircalc_new(model, lat, lon, 0.0, &ns_calc);
  oracle_dircalc_new(model, lat, lon, 90.0, &ew_calc);
  oracle_coords_at_dist(&ns_calc, DIFF, &nlat, &nlon);
  oracle_coords_at_dist(&ns_calc, -DIFF, &slat, &slon);
  oracle_coords_at_dist(&ew_calc, DIFF, &elat, &elon);
  oracle_coords_at_dist(&ew_calc, -DIFF, &wlat, &wlon);
  oracle_world_directions(model, lat, lon, &dir_north, &dir_east, &dir_up);
  diff_ew = elev_or_zero(terrain, elat, elon) - elev_or_zero(terrain, wlat, wlon);
  diff_ns = elev_or_zero(terrain, nlat, nlon) - elev_or_zero(terrain, slat, slon);
  vec_ns.x = 2.0 * DIFF * dir_north.x + diff_ns * dir_up.x;
  vec_ns.y = 2.0 * DIFF * dir_north.y + diff_ns * dir_up.y;
  vec_ns.z = 2.0 * DIFF * dir_north.z + diff_ns * dir_up.z;
  vec_ew.x = 2.0 * DIFF * dir_east.x + diff_ew * dir_up.x;
  vec_ew.y = 2.0 * DIFF * dir_east.y + diff_ew * dir_up.y;
  vec_ew.z = 2.0 * DIFF * dir_east.z + diff_ew * dir_up.z;
  normal.x = vec_ew.y * vec_ns.z - vec_ew.z * vec_ns.y;
  normal.y = vec_ew.z * vec_ns.x - vec_ew.x * vec_ns.z;
  normal.z = vec_ew.x * vec_ns.y - vec_ew.y * vec_ns.x;
  len = om_sqrt(normal.x * normal.x + normal.y * normal.y + normal.z * normal.z);
  normal.x /= len;
  normal.y /= len;
  normal.z /= len;
  return normal;
}

/* calc_dist, utils.rs:42-53 */
static double calc_dist(const gen_ctx* g, oracle_ray_state old_state, oracle_ray_state new_state) {
  double dx = new_state.x - old_state.x;
  double dh = new_state.h - old_state.h;
  if (!g->spherical) return om_sqrt(dx * dx + dh * dh);
  {
    double avg_h = (new_state.h + old_state.h) / 2.0;
    double dx2 = dx / g->radius * (avg_h + g->radius);
    return om_sqrt(dx2 * dx2 + dh * dh);
  }
}

/* TerrainData::from_lat_lon, utils.rs:72-88 */
static terrain_data terrain_data_from_lat_lon(const gen_ctx* g, double lat, double lon) {
  terrain_data td;
  size_t i;
  td.normal = oracle_find_normal(&g->params->earth, lat, lon, g->terrain);
  td.n_close = 0;
  td.close = NULL;
  for (i = 0; i < g->n_objects; i++) {
    if (oracle_object_is_close(&g->objects[i], &g->params->earth, g->params->simulation_step, lat, lon)) {
      if (!td.close) td.close = (int*)malloc(g->n_objects * sizeof(int));
      td.close[td.n_close++] = (int)i;
    }
  }
  td.lat = lat;
  td.lon = lon;
  td.elev = elev_or_zero(g->terrain, lat, lon);
  return td;
}

typedef struct { /* TracingState, utils.rs:91-96 */
  terrain_data td;
  double ray_elev, dist, path_len;
} tracing_state;

/* TracingState::interpolate, utils.rs:108-125 */
static tracing_state ts_interpolate(const tracing_state* a, const tracing_state* b, double prop) {
  tracing_state r;
  r.td.lat = a->td.lat + (b->td.lat - a->td.lat) * prop;
  r.td.lon = a->td.lon + (b->td.lon - a->td.lon) * prop;
  r.td.elev = a->td.elev + (b->td.elev - a->td.elev) * prop;
  r.td.normal.x = a->td.normal.x + (b->td.normal.x - a->td.normal.x) * prop;
  r.td.normal.y = a->td.normal.y + (b->td.normal.y - a->td.normal.y) * prop;
  r.td.normal.z = a->td.normal.z + (b->td.normal.z - a->td.normal.z) * prop;
  r.td.n_close = 0;
  r.td.close = NULL;
  r.ray_elev = a->ray_elev + (b->ray_elev - a->ray_elev) * prop;
  r.dist = a->dist + (b->dist - a->dist) * prop;
  r.path_len = a->path_len + (b->path_len - a->path_len) * prop;
  return r;
}

typedef struct { /* TracePoint, generators/mod.rs:21-30 */
  double lat, lon, distance, elevation, path_length;
  ovec3 normal;
  uint32_t tag;
  double rgba[4];
} trace_point;

typedef struct {
  trace_point* v;
  size_t n, cap;
} tp_vec;

static void tp_push(tp_vec* v, const trace_point* p) {
  if (v->n == v->cap) {
    v->cap = v->cap ? 2 * v->cap : 4;
    v->v = (trace_point*)realloc(v->v, v->cap * sizeof(trace_point));
  }
  v->v[v->n++] = *p;
}

/* A stream of (TerrainData, PathElem): returns 0 when exhausted. */
typedef int (*sample_next_fn)(void* it, terrain_data* td, path_elem* pe);

typedef struct {
  double prop;
  trace_point tp;
} step_hit;

/* get_single_pixel, utils.rs:201-289.  Returns the number of loop iterations (ray-steps). */
static uint64_t get_single_pixel(void* it, sample_next_fn next, const gen_ctx* g, double terrain_alpha, tp_vec* result) {
  terrain_data first_terrain;
  path_elem first_path;
  tracing_state old_ts, new_ts;
  uint64_t steps = 0;
  step_hit* step_result = NULL;
  size_t step_cap = 0;
  if (!next(it, &first_terrain, &first_path)) return 0; /* the reference unwraps; streams are never empty */
  old_ts.td = first_terrain;
  old_ts.ray_elev = first_path.elev;
  old_ts.dist = 0.0;
  old_ts.path_len = 0.0;
  for (;;) {
    terrain_data td;
    path_elem pe;
    int finish = 0;
    size_t n_step = 0, i, j;
    double diff1, diff2;
    if (!next(it, &td, &pe)) break;
    steps++;
    new_ts.td = td;
    new_ts.ray_elev = pe.elev;
    new_ts.dist = pe.dist;
    new_ts.path_len = pe.path_length;
    diff1 = old_ts.ray_elev - old_ts.td.elev;
    diff2 = new_ts.ray_elev - new_ts.td.elev;
    if (step_cap < 1 + 4 * g->n_objects) {
      step_cap = 1 + 4 * g->n_objects;
      step_result = (step_hit*)realloc(step_result, step_cap * sizeof(step_hit));
    }
    if (diff1 * diff2 < 0.0) {
      double prop = diff1 / (diff1 - diff2);
      tracing_state in = ts_interpolate(&old_ts, &new_ts, prop);
      step_hit* h = &step_result[n_step++];
      h->prop = prop;
      h->tp.lat = in.td.lat;
      h->tp.lon = in.td.lon;
      h->tp.distance = in.dist;
      h->tp.elevation = in.td.elev;
      h->tp.path_length = in.path_len;
      h->tp.normal = in.td.normal;
      h->tp.tag = ATMRT_COLOR_TERRAIN;
      h->tp.rgba[0] = h->tp.rgba[1] = h->tp.rgba[2] = 0.0;
      h->tp.rgba[3] = terrain_alpha;
      if (terrain_alpha == 1.0) finish = 1;
    }
    if (new_ts.td.n_close != 0 || old_ts.td.n_close != 0) {
      /* union of the two index lists; the reference iterates a HashSet (arbitrary order, only
       * ties in prop are affected) — ascending index order is one valid instance */
      int ia = 0, ib = 0;
      ocoords c1, c2;
      c1.lat = old_ts.td.lat; c1.lon = old_ts.td.lon; c1.elev = old_ts.ray_elev; /* ray_coords, utils.rs:127-133 */
      c2.lat = new_ts.td.lat; c2.lon = new_ts.td.lon; c2.elev = new_ts.ray_elev;
      while (ia < old_ts.td.n_close || ib < new_ts.td.n_close) {
        int idx;
        oracle_collision col[4];
        int nc, k;
        if (ib >= new_ts.td.n_close || (ia < old_ts.td.n_close && old_ts.td.close[ia] <= new_ts.td.close[ib])) {
          idx = old_ts.td.close[ia];
          if (ib < new_ts.td.n_close && new_ts.td.close[ib] == idx) ib++;
          ia++;
        } else {
          idx = new_ts.td.close[ib++];
        }
        nc = oracle_object_collision(&g->objects[idx], &g->params->earth, c1, c2, col);
        for (k = 0; k < nc; k++) {
          tracing_state in;
          step_hit* h;
          if (col[k].color[3] == 0.0) continue;
          in = ts_interpolate(&old_ts, &new_ts, col[k].prop);
          h = &step_result[n_step++];
          h->prop = col[k].prop;
          h->tp.lat = in.td.lat;
          h->tp.lon = in.td.lon;
          h->tp.distance = in.dist;
          h->tp.elevation = in.ray_elev;
          h->tp.path_length = in.path_len;
          h->tp.normal = col[k].normal;
          h->tp.tag = ATMRT_COLOR_RGBA;
          memcpy(h->tp.rgba, col[k].color, sizeof h->tp.rgba);
          if (col[k].color[3] == 1.0) {
            finish = 1;
            break;
          }
        }
      }
    }
    /* step_result.sort_by(prop): stable insertion sort */
    for (i = 1; i < n_step; i++) {
      step_hit key = step_result[i];
      j = i;
      while (j > 0 && step_result[j - 1].prop > key.prop) {
        step_result[j] = step_result[j - 1];
        j--;
      }
      step_result[j] = key;
    }
    for (i = 0; i < n_step; i++) tp_push(result, &step_result[i].tp);
    if (finish) break;
    old_ts = new_ts;
  }
  free(step_result);
  return steps;
}

/* gen_path_cache, utils.rs:136-174 */
static path_elem* gen_path_cache(const gen_ctx* g, double ray_elev_deg, size_t* n_out) {
  const atmrt_params_t* p = g->params;
  oracle_stepper ray;
  size_t cap = (size_t)(p->frame.max_distance / p->simulation_step) + 8, n = 0;
  path_elem* path = (path_elem*)malloc(cap * sizeof(path_elem));
  oracle_ray_state ray_state;
  double path_length = 0.0;
  oracle_stepper_init(&ray, &g->atm, g->spherical, g->radius, g->alt, om_to_radians(ray_elev_deg), p->straight_rays,
                      p->simulation_step);
  path[n].dist = 0.0;
  path[n].elev = g->alt;
  path[n].path_length = 0.0;
  n++;
  ray_state.x = 0.0;
  ray_state.h = g->alt;
  ray_state.dh = 0.0;
  for (;;) {
    oracle_ray_state new_ray_state = oracle_stepper_next(&ray);
    path_length += calc_dist(g, ray_state, new_ray_state);
    if (n == cap) {
      cap *= 2;
      path = (path_elem*)realloc(path, cap * sizeof(path_elem));
    }
    path[n].dist = new_ray_state.x;
    path[n].elev = new_ray_state.h;
    path[n].path_length = path_length;
    n++;
    if (ray_state.x > p->frame.max_distance || ray_state.h < -1000.0) break;
    ray_state = new_ray_state;
  }
  *n_out = n;
  return path;
}

/* gen_terrain_cache, utils.rs:176-199 */
static terrain_data* gen_terrain_cache(const gen_ctx* g, double dir, size_t* n_out) {
  const atmrt_params_t* p = g->params;
  double distance = 0.0;
  size_t cap = (size_t)(p->frame.max_distance / p->simulation_step) + 2, n = 0;
  terrain_data* result = (terrain_data*)malloc(cap * sizeof(terrain_data));
  oracle_dircalc calc;
  oracle_dircalc_new(&p->earth, p->position.latitude, p->position.longitude, dir, &calc);
  while (distance < p->frame.max_distance) {
    double lat, lon;
    oracle_coords_at_dist(&calc, distance, &lat, &lon);
    if (n == cap) {
      cap *= 2;
      result = (terrain_data*)realloc(result, cap * sizeof(terrain_data));
    }
    result[n++] = terrain_data_from_lat_lon(g, lat, lon);
    distance += p->simulation_step;
  }
  *n_out = n;
  return result;
}

/* ---- fast.rs ------------------------------------------------------------------------------ */

/* get_ray_elev, fast.rs:111-118 */
static double fast_ray_elev(const atmrt_params_t* p, uint16_t y) {
  double width = (double)p->width, height = (double)p->height;
  double aspect = width / height;
  double yy = (double)(int16_t)((int16_t)y - (int16_t)p->height / 2) / height;
  return p->frame.tilt - yy * p->frame.fov / aspect;
}
/* get_ray_dir, fast.rs:120-125 */
static double fast_ray_dir(const atmrt_params_t* p, uint16_t x) {
  double width = (double)p->width;
  double xx = (double)(int16_t)((int16_t)x - (int16_t)p->width / 2) / width;
  return p->frame.direction + xx * p->frame.fov;
}

typedef struct {
  const terrain_data* t;
  const path_elem* p;
  size_t i, n;
} zip_iter;

static int zip_next(void* it, terrain_data* td, path_elem* pe) {
  zip_iter* z = (zip_iter*)it;
  if (z->i >= z->n) return 0;
  *td = z->t[z->i];
  *pe = z->p[z->i];
  z->i++;
  return 1;
}

typedef struct {
  double azimuth, elevation_angle;
  tp_vec tps;
  uint64_t steps;
} pixel_out;

/* FastGenerator::generate, fast.rs:22-98 */
static void generate_fast(const gen_ctx* g, int c0, int c1, pixel_out* px) {
  const atmrt_params_t* p = g->params;
  int w = c1 - c0, h = p->height, x, y;
  terrain_data** terrain_cache = (terrain_data**)malloc((size_t)w * sizeof(*terrain_cache));
  size_t* terrain_n = (size_t*)malloc((size_t)w * sizeof(size_t));
  path_elem** path_cache = (path_elem**)malloc((size_t)h * sizeof(*path_cache));
  size_t* path_n = (size_t*)malloc((size_t)h * sizeof(size_t));
#pragma omp parallel for schedule(dynamic, 1)
  for (x = 0; x < w; x++) terrain_cache[x] = gen_terrain_cache(g, fast_ray_dir(p, (uint16_t)(c0 + x)), &terrain_n[x]);
#pragma omp parallel for schedule(dynamic, 1)
  for (y = 0; y < h; y++) {
    path_cache[y] = NULL;
    if (row_selected(y)) path_cache[y] = gen_path_cache(g, fast_ray_elev(p, (uint16_t)y), &path_n[y]);
  }
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (y = 0; y < h; y++) {
    for (x = 0; x < w; x++) {
      pixel_out* o = &px[(size_t)y * w + x];
      zip_iter z;
      double azimuth;
      if (!row_selected(y)) continue;
      z.t = terrain_cache[x];
      z.p = path_cache[y];
      z.i = 0;
      z.n = terrain_n[x] < path_n[y] ? terrain_n[x] : path_n[y]; /* Iterator::zip */
      o->steps = get_single_pixel(&z, zip_next, g, p->terrain_alpha, &o->tps);
      azimuth = fast_ray_dir(p, (uint16_t)(c0 + x));
      if (azimuth < 0.0) azimuth += 360.0;
      else if (azimuth >= 360.0) azimuth -= 360.0;
      o->elevation_angle = fast_ray_elev(p, (uint16_t)y);
      o->azimuth = azimuth;
    }
  }
  for (x = 0; x < w; x++) {
    size_t i;
    for (i = 0; i < terrain_n[x]; i++) free(terrain_cache[x][i].close);
    free(terrain_cache[x]);
  }
  for (y = 0; y < h; y++) free(path_cache[y]);
  free(terrain_cache);
  free(terrain_n);
  free(path_cache);
  free(path_n);
}

/* ---- rectilinear.rs ----------------------------------------------------------------------- */

typedef struct { double elevation, direction; } ray_params; /* radians, rectilinear.rs:62-66 */

/* RectilinearGenerator::get_ray_params, rectilinear.rs:78-100.
 * nalgebra Matrix4::from_euler_angles(roll, pitch, yaw) builds Rz(yaw) Ry(pitch) Rx(roll);
 * transform_vector applies the upper-left 3x3 (the homogeneous row is 0 0 0 1). */
static ray_params rect_ray_params(const atmrt_params_t* p, uint16_t px, uint16_t py) {
  double width = (double)p->width;
  double x = (double)(int16_t)((int16_t)px - (int16_t)p->width / 2);
  double y = (double)(int16_t)((int16_t)py - (int16_t)p->height / 2);
  double z = width / 2.0 / om_tan(om_to_radians(p->frame.fov) / 2.0);
  double roll = 0.0, pitch = -om_to_radians(p->frame.tilt), yaw = om_to_radians(p->frame.direction);
  double sr = om_sin(roll), cr = om_cos(roll), sp = om_sin(pitch), cp = om_cos(pitch), sy = om_sin(yaw), cy = om_cos(yaw);
  double m00 = cy * cp, m01 = cy * sp * sr - sy * cr, m02 = cy * sp * cr + sy * sr;
  double m10 = sy * cp, m11 = sy * sp * sr + cy * cr, m12 = sy * sp * cr - cy * sr;
  double m20 = -sp, m21 = cp * sr, m22 = cp * cr;
  /* Vector3::new(z, x, -y): [forward, right, up] */
  double vx = z, vy = x, vz = -y;
  double dx = m00 * vx + m01 * vy + m02 * vz;
  double dy = m10 * vx + m11 * vy + m12 * vz;
  double dz = m20 * vx + m21 * vy + m22 * vz;
  double len = om_sqrt(dx * dx + dy * dy + dz * dz);
  ray_params r;
  dx /= len;
  dy /= len;
  dz /= len;
  r.direction = om_atan2(dy, dx);
  r.elevation = om_asin(dz);
  return r;
}

typedef struct { /* PathIterator, rectilinear.rs:118-125 */
  const gen_ctx* g;
  double path_length;
  oracle_ray_state ray_state;
  oracle_stepper ray;
  oracle_dircalc dist_calc;
  terrain_data pending_free[2];
  int n_pending;
} path_iterator;

/* PathIterator::next, rectilinear.rs:161-185 */
static int path_iter_next(void* it, terrain_data* td, path_elem* pe) {
  path_iterator* s = (path_iterator*)it;
  const atmrt_params_t* p = s->g->params;
  double lat, lon;
  oracle_ray_state new_state;
  pe->dist = s->ray_state.x;
  pe->elev = s->ray_state.h;
  pe->path_length = s->path_length;
  if (pe->dist > p->frame.max_distance || pe->elev < -1000.0) return 0;
  oracle_coords_at_dist(&s->dist_calc, pe->dist, &lat, &lon);
  *td = terrain_data_from_lat_lon(s->g, lat, lon);
  /* keep at most the two live samples' object lists; free older ones */
  if (s->n_pending == 2) {
    free(s->pending_free[0].close);
    s->pending_free[0] = s->pending_free[1];
    s->n_pending = 1;
  }
  s->pending_free[s->n_pending++] = *td;
  new_state = oracle_stepper_next(&s->ray);
  s->path_length += calc_dist(s->g, s->ray_state, new_state);
  s->ray_state = new_state;
  return 1;
}

/* RectilinearGenerator::{generate,gen_pixel}, rectilinear.rs:24-60,102-116 */
static void generate_rectilinear(const gen_ctx* g, int c0, int c1, pixel_out* px) {
  const atmrt_params_t* p = g->params;
  int w = c1 - c0, h = p->height, x, y;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (y = 0; y < h; y++) {
    for (x = 0; x < w; x++) {
      pixel_out* o = &px[(size_t)y * w + x];
      ray_params rp = rect_ray_params(p, (uint16_t)(c0 + x), (uint16_t)y);
      path_iterator it;
      int k;
      if (!row_selected(y)) continue;
      it.g = g;
      it.path_length = 0.0;
      it.n_pending = 0;
      it.ray_state.x = 0.0;
      it.ray_state.h = g->alt;
      it.ray_state.dh = 0.0;
      oracle_stepper_init(&it.ray, &g->atm, g->spherical, g->radius, g->alt, rp.elevation, p->straight_rays,
                          p->simulation_step);
      oracle_dircalc_new(&p->earth, p->position.latitude, p->position.longitude, om_to_degrees(rp.direction),
                         &it.dist_calc);
      o->steps = get_single_pixel(&it, path_iter_next, g, p->terrain_alpha, &o->tps);
      for (k = 0; k < it.n_pending; k++) free(it.pending_free[k].close);
      o->elevation_angle = om_to_degrees(rp.elevation);
      o->azimuth = om_to_degrees(rp.direction);
    }
  }
}

/* ---- interpolating_rectilinear.rs --------------------------------------------------------- */

typedef struct { int32_t elev_index, dir_index; } cache_coords; /* :170-174 */

typedef struct { /* ResultPixel of a lattice point, Cache::get_pixel :80-107 */
  cache_coords key;
  double azimuth, elevation_angle;
  tp_vec tps;
  uint64_t steps;
} lattice_pixel;

static int cmp_i32(const void* a, const void* b) {
  int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
  return x < y ? -1 : x > y;
}
static int cmp_key(const void* a, const void* b) {
  const cache_coords* x = (const cache_coords*)a;
  const cache_coords* y = (const cache_coords*)b;
  if (x->elev_index != y->elev_index) return x->elev_index < y->elev_index ? -1 : 1;
  return x->dir_index < y->dir_index ? -1 : x->dir_index > y->dir_index;
}
static int cmp_lattice(const void* a, const void* b) {
  return cmp_key(&((const lattice_pixel*)a)->key, &((const lattice_pixel*)b)->key);
}

/* TracePoint::interpolate, generators/mod.rs:33-43 + PixelColor::interpolate :67-79 */
static trace_point tp_interpolate(const trace_point* a, const trace_point* b, double c) {
  trace_point r;
  int k;
  r.lat = a->lat * (1.0 - c) + b->lat * c;
  r.lon = a->lon * (1.0 - c) + b->lon * c;
  r.distance = a->distance * (1.0 - c) + b->distance * c;
  r.elevation = a->elevation * (1.0 - c) + b->elevation * c;
  r.path_length = a->path_length * (1.0 - c) + b->path_length * c;
  r.normal.x = a->normal.x * (1.0 - c) + b->normal.x * c;
  r.normal.y = a->normal.y * (1.0 - c) + b->normal.y * c;
  r.normal.z = a->normal.z * (1.0 - c) + b->normal.z * c;
  if (a->tag == ATMRT_COLOR_TERRAIN && b->tag == ATMRT_COLOR_TERRAIN) {
    r.tag = ATMRT_COLOR_TERRAIN;
    r.rgba[0] = r.rgba[1] = r.rgba[2] = 0.0;
    r.rgba[3] = a->rgba[3] * (1.0 - c) + b->rgba[3] * c;
  } else if (a->tag == ATMRT_COLOR_RGBA && b->tag == ATMRT_COLOR_RGBA) {
    r.tag = ATMRT_COLOR_RGBA;
    for (k = 0; k < 4; k++) r.rgba[k] = a->rgba[k] * (1.0 - c) + b->rgba[k] * c;
  } else {
    const trace_point* t = a->tag == ATMRT_COLOR_TERRAIN ? a : b;
    r.tag = ATMRT_COLOR_TERRAIN;
    r.rgba[0] = r.rgba[1] = r.rgba[2] = 0.0;
    r.rgba[3] = t->rgba[3];
  }
  return r;
}

/* interpolate_two_adjacent :339-350, _two_diagonal :352-364, _three :366-380, _four :382-393 */
static int interp_two_adjacent(const trace_point* e0, const trace_point* e1, double rem_elev, double rem_dir, trace_point* out) {
  if (rem_elev >= 0.5) return 0;
  *out = tp_interpolate(e0, e1, rem_dir);
  return 1;
}
static int interp_two_diagonal(const trace_point* e0, const trace_point* e1, double rem_elev, double rem_dir, trace_point* out) {
  double coeff;
  if ((rem_elev >= 0.5 && rem_dir < 0.5) || (rem_elev < 0.5 && rem_dir >= 0.5)) return 0;
  coeff = rem_elev * rem_dir / (rem_elev * rem_dir + (1.0 - rem_elev) * (1.0 - rem_dir));
  *out = tp_interpolate(e0, e1, coeff);
  return 1;
}
static int interp_three(const trace_point* e0, const trace_point* e1, const trace_point* e2, double rem_elev, double rem_dir, trace_point* out) {
  double sum;
  trace_point in;
  if (rem_elev >= 0.5 && rem_dir >= 0.5) return 0;
  sum = 1.0 - rem_elev + rem_elev * (1.0 - rem_dir);
  in = tp_interpolate(e0, e1, rem_dir);
  *out = tp_interpolate(&in, e2, rem_elev * (1.0 - rem_dir) / sum);
  return 1;
}

/* interpolate_trace_points, interpolating_rectilinear.rs:267-337 */
static int interp_group(const trace_point* e[4], double re, double rd, trace_point* out) {
  int mask = (e[0] ? 1 : 0) | (e[1] ? 2 : 0) | (e[2] ? 4 : 0) | (e[3] ? 8 : 0);
  switch (mask) {
    case 0: return 0;
    case 1: if (re < 0.5 && rd < 0.5) { *out = *e[0]; return 1; } return 0;
    case 2: if (re < 0.5 && rd >= 0.5) { *out = *e[1]; return 1; } return 0;
    case 4: if (re >= 0.5 && rd < 0.5) { *out = *e[2]; return 1; } return 0;
    case 8: if (re >= 0.5 && rd >= 0.5) { *out = *e[3]; return 1; } return 0;
    case 1 | 2: return interp_two_adjacent(e[0], e[1], re, rd, out);
    case 1 | 4: return interp_two_adjacent(e[0], e[2], rd, re, out);
    case 1 | 8: return interp_two_diagonal(e[0], e[3], re, rd, out);
    case 2 | 4: return interp_two_diagonal(e[1], e[2], re, 1.0 - rd, out);
    case 2 | 8: return interp_two_adjacent(e[1], e[3], 1.0 - rd, re, out);
    case 4 | 8: return interp_two_adjacent(e[2], e[3], 1.0 - re, rd, out);
    case 1 | 2 | 4: return interp_three(e[0], e[1], e[2], re, rd, out);
    case 1 | 2 | 8: return interp_three(e[1], e[0], e[3], re, 1.0 - rd, out);
    case 1 | 4 | 8: return interp_three(e[0], e[3], e[2], 1.0 - re, rd, out);
    case 2 | 4 | 8: return interp_three(e[3], e[2], e[1], 1.0 - re, 1.0 - rd, out);
    default: {
      trace_point i1 = tp_interpolate(e[0], e[1], rd);
      trace_point i2 = tp_interpolate(e[2], e[3], rd);
      *out = tp_interpolate(&i1, &i2, re);
      return 1;
    }
  }
}

/* interpolate, :395-418 with collect_trace_points :213-243 and match_sequence :245-265 */
static void interp_pixel(const lattice_pixel* const px4[4], double rem_elev, double rem_dir, double step_size, pixel_out* o) {
  /* groups of (corner index, trace point) */
  typedef struct { int corner; const trace_point* tp; } member;
  size_t total = px4[0]->tps.n + px4[1]->tps.n + px4[2]->tps.n + px4[3]->tps.n;
  member* members = (member*)malloc((total + 1) * sizeof(member));
  int* group_of = (int*)malloc((total + 1) * sizeof(int));
  size_t n_members = 0, i, k;
  int n_groups = 0, c, gi;
  for (c = 0; c < 4; c++) {
    for (k = 0; k < px4[c]->tps.n; k++) {
      const trace_point* tp = &px4[c]->tps.v[k];
      int found = -1;
      /* first group (in creation order) that has any close point of the same class */
      for (gi = 0; gi < n_groups && found < 0; gi++)
        for (i = 0; i < n_members; i++)
          if (group_of[i] == gi && om_fabs(tp->distance - members[i].tp->distance) < step_size &&
              tp->tag == members[i].tp->tag) {
            found = gi;
            break;
          }
      if (found < 0) found = n_groups++;
      members[n_members].corner = c;
      members[n_members].tp = tp;
      group_of[n_members] = found;
      n_members++;
    }
  }
  for (gi = 0; gi < n_groups; gi++) {
    const trace_point* e[4] = {NULL, NULL, NULL, NULL};
    trace_point outp;
    for (i = 0; i < n_members; i++)
      if (group_of[i] == gi) e[members[i].corner] = members[i].tp; /* later entries overwrite, :247-263 */
    if (interp_group(e, rem_elev, rem_dir, &outp)) tp_push(&o->tps, &outp);
  }
  o->elevation_angle = px4[0]->elevation_angle * (1.0 - rem_elev) * (1.0 - rem_dir) +
                       px4[1]->elevation_angle * (1.0 - rem_elev) * rem_dir +
                       px4[2]->elevation_angle * rem_elev * (1.0 - rem_dir) + px4[3]->elevation_angle * rem_elev * rem_dir;
  o->azimuth = px4[0]->azimuth * (1.0 - rem_elev) * (1.0 - rem_dir) + px4[1]->azimuth * (1.0 - rem_elev) * rem_dir +
               px4[2]->azimuth * rem_elev * (1.0 - rem_dir) + px4[3]->azimuth * rem_elev * rem_dir;
  free(members);
  free(group_of);
}

/* InterpolatingRectilinearGenerator::generate :110-162 with gen_fov_data :453-522.  The
 * reference memoises lattice pixels in RwLock<HashMap>s; the values are pure functions of the key,
 * so computing every referenced key once gives identical results. */
static void generate_interpolating(const gen_ctx* g, int c0, int c1, pixel_out* px) {
  const atmrt_params_t* p = g->params;
  int W = p->width, H = p->height, w = c1 - c0, x, y;
  const double full = om_to_radians(360.0);
  const double min_diff = om_to_radians(p->frame.fov) / (double)p->width / 3.0;
  ray_params* table = (ray_params*)malloc((size_t)W * H * sizeof(ray_params));
  double min_elev_step = 1.0 / 0.0, min_dir_step = 1.0 / 0.0;
  cache_coords* keys;
  double* rems;
  size_t n_px = (size_t)w * H, n_keys, i, n_lat;
  lattice_pixel* lattice;
  int32_t *dir_idx, *elev_idx;
  size_t n_dir, n_elev;
  terrain_data** tcache;
  size_t* tcache_n;
  path_elem** pcache;
  size_t* pcache_n;
  for (y = 0; y < H; y++)
    for (x = 0; x < W; x++) table[(size_t)y * W + x] = rect_ray_params(p, (uint16_t)x, (uint16_t)y);
  for (x = 0; x < W; x++) {
    double mn = full, last = table[x].elevation;
    for (y = 1; y < H; y++) {
      double next = table[(size_t)y * W + x].elevation;
      double diff = om_fabs(next - last);
      if (diff < min_diff) diff = min_diff;
      if (diff < mn) mn = diff;
      last = next;
    }
    if (mn < min_elev_step) min_elev_step = mn;
  }
  min_elev_step *= 1.5;
  for (y = 0; y < H; y++) {
    double mn = full, last = table[(size_t)y * W].direction;
    for (x = 1; x < W; x++) {
      double next = table[(size_t)y * W + x].direction;
      double diff = om_fabs(next - last);
      if (diff > full) diff -= full;
      if (diff < min_diff) diff = min_diff;
      if (diff < mn) mn = diff;
      last = next;
    }
    if (mn < min_dir_step) min_dir_step = mn;
  }
  min_dir_step *= 1.5;
  /* FovData::cache_coords :186-204 for every pixel of the shard */
  keys = (cache_coords*)malloc(4 * n_px * sizeof(cache_coords));
  rems = (double*)malloc(2 * n_px * sizeof(double));
  for (y = 0; y < H; y++)
    for (x = 0; x < w; x++) {
      ray_params rp = table[(size_t)y * W + c0 + x];
      size_t pi = (size_t)y * w + x;
      double ef = rp.elevation / min_elev_step, df = rp.direction / min_dir_step;
      int32_t ei = (int32_t)om_floor(ef), di = (int32_t)om_floor(df);
      int s;
      rems[2 * pi] = ef - (double)ei;
      rems[2 * pi + 1] = df - (double)di;
      for (s = 0; s < 4; s++) { /* SEQUENCE = (0,0) (0,1) (1,0) (1,1) */
        keys[4 * pi + s].elev_index = ei + (s >> 1);
        keys[4 * pi + s].dir_index = di + (s & 1);
      }
    }
  /* unique lattice points, unique directions, unique elevations */
  n_keys = 4 * n_px;
  lattice = (lattice_pixel*)calloc(n_keys, sizeof(lattice_pixel));
  for (i = 0; i < n_keys; i++) lattice[i].key = keys[i];
  qsort(lattice, n_keys, sizeof(lattice_pixel), cmp_lattice);
  n_lat = 0;
  for (i = 0; i < n_keys; i++)
    if (n_lat == 0 || cmp_key(&lattice[n_lat - 1].key, &lattice[i].key)) lattice[n_lat++] = lattice[i];
  dir_idx = (int32_t*)malloc(n_lat * sizeof(int32_t));
  elev_idx = (int32_t*)malloc(n_lat * sizeof(int32_t));
  for (i = 0; i < n_lat; i++) {
    dir_idx[i] = lattice[i].key.dir_index;
    elev_idx[i] = lattice[i].key.elev_index;
  }
  qsort(dir_idx, n_lat, sizeof(int32_t), cmp_i32);
  qsort(elev_idx, n_lat, sizeof(int32_t), cmp_i32);
  n_dir = n_elev = 0;
  for (i = 0; i < n_lat; i++) {
    if (n_dir == 0 || dir_idx[n_dir - 1] != dir_idx[i]) dir_idx[n_dir++] = dir_idx[i];
    if (n_elev == 0 || elev_idx[n_elev - 1] != elev_idx[i]) elev_idx[n_elev++] = elev_idx[i];
  }
  tcache = (terrain_data**)malloc(n_dir * sizeof(*tcache));
  tcache_n = (size_t*)malloc(n_dir * sizeof(size_t));
  pcache = (path_elem**)malloc(n_elev * sizeof(*pcache));
  pcache_n = (size_t*)malloc(n_elev * sizeof(size_t));
  {
    long k;
#pragma omp parallel for schedule(dynamic, 1)
    for (k = 0; k < (long)n_dir; k++) /* Cache::get_terrain_cache :61-78 */
      tcache[k] = gen_terrain_cache(g, om_to_degrees((double)dir_idx[k] * min_dir_step), &tcache_n[k]);
#pragma omp parallel for schedule(dynamic, 1)
    for (k = 0; k < (long)n_elev; k++) /* Cache::get_path_cache :44-59 */
      pcache[k] = gen_path_cache(g, om_to_degrees((double)elev_idx[k] * min_elev_step), &pcache_n[k]);
#pragma omp parallel for schedule(dynamic, 16)
    for (k = 0; k < (long)n_lat; k++) { /* Cache::get_pixel :80-107 */
      lattice_pixel* lp = &lattice[k];
      int32_t* dp = (int32_t*)bsearch(&lp->key.dir_index, dir_idx, n_dir, sizeof(int32_t), cmp_i32);
      int32_t* ep = (int32_t*)bsearch(&lp->key.elev_index, elev_idx, n_elev, sizeof(int32_t), cmp_i32);
      size_t di = (size_t)(dp - dir_idx), ei = (size_t)(ep - elev_idx);
      zip_iter z;
      double azimuth;
      z.t = tcache[di];
      z.p = pcache[ei];
      z.i = 0;
      z.n = tcache_n[di] < pcache_n[ei] ? tcache_n[di] : pcache_n[ei];
      lp->steps = get_single_pixel(&z, zip_next, g, p->terrain_alpha, &lp->tps);
      azimuth = om_to_degrees((double)lp->key.dir_index * min_dir_step);
      if (azimuth < 0.0) azimuth += 360.0;
      else if (azimuth >= 360.0) azimuth -= 360.0;
      lp->azimuth = azimuth;
      lp->elevation_angle = om_to_degrees((double)lp->key.elev_index * min_elev_step);
    }
#pragma omp parallel for schedule(dynamic, 64)
    for (k = 0; k < (long)n_px; k++) {
      const lattice_pixel* px4[4];
      int s;
      for (s = 0; s < 4; s++) {
        lattice_pixel probe;
        probe.key = keys[4 * k + s];
        px4[s] = (const lattice_pixel*)bsearch(&probe, lattice, n_lat, sizeof(lattice_pixel), cmp_lattice);
      }
      interp_pixel(px4, rems[2 * k], rems[2 * k + 1], p->simulation_step, &px[k]);
    }
  }
  /* ray-steps: every memoised lattice pixel is traced once */
  for (i = 0; i < n_lat; i++) {
    px[0].steps += lattice[i].steps;
    free(lattice[i].tps.v);
  }
  for (i = 0; i < n_dir; i++) {
    size_t k;
    for (k = 0; k < tcache_n[i]; k++) free(tcache[i][k].close);
    free(tcache[i]);
  }
  for (i = 0; i < n_elev; i++) free(pcache[i]);
  free(tcache); free(tcache_n); free(pcache); free(pcache_n);
  free(dir_idx); free(elev_idx); free(lattice); free(keys); free(rems); free(table);
}

/* ---- entry point -------------------------------------------------------------------------- */

int oracle_generate(const atmrt_params_t* params, const atmrt_atmosphere_t* atm, const oracle_terrain* terrain,
                    const atmrt_object_t* objects, size_t n_objects, int n_threads, atmrt_result_t* out) {
  gen_ctx g;
  oracle_object* objs = NULL;
  pixel_out* px;
  int c0 = params->col_begin, c1 = params->col_end, w, h = params->height;
  size_t i, n_px, n_hits = 0, off = 0;
  if (c0 == 0 && c1 == 0) c1 = params->width;
  if (c1 <= c0 || c1 > params->width || h <= 0) return -1;
  w = c1 - c0;
  memset(&g, 0, sizeof g);
  g.params = params;
  g.terrain = terrain;
  if (oracle_atm_compile(atm, params->wavelength, &g.atm)) {
    oracle_atm_free(&g.atm);
    return -1;
  }
  g.spherical = oracle_to_shape(&params->earth, &g.radius);
  /* Altitude::abs, params.rs:23-30 */
  g.alt = params->position.altitude_kind == ATMRT_ALT_ABSOLUTE
              ? params->position.altitude
              : elev_or_zero(terrain, params->position.latitude, params->position.longitude) + params->position.altitude;
  if (n_objects) {
    objs = (oracle_object*)calloc(n_objects, sizeof(oracle_object));
    for (i = 0; i < n_objects; i++) {
      const atmrt_object_t* s = &objects[i];
      oracle_object* o = &objs[i];
      o->kind = s->kind;
      o->lat = s->position.latitude;
      o->lon = s->position.longitude;
      /* ConfObject::into_serializable_object, object/mod.rs:164-183 */
      o->elev = s->position.altitude_kind == ATMRT_ALT_ABSOLUTE
                    ? s->position.altitude
                    : elev_or_zero(terrain, s->position.latitude, s->position.longitude) + s->position.altitude;
      o->r1 = s->r1; o->r2 = s->r2; o->height = s->height; o->width = s->width;
      memcpy(o->color, s->color, sizeof o->color);
      o->tex = s->texture_rgba; o->tex_w = s->texture_width; o->tex_h = s->texture_height;
      if (o->kind == ATMRT_OBJ_BILLBOARD && (!o->tex || o->tex_w < 2 || o->tex_h < 2)) { free(objs); oracle_atm_free(&g.atm); return -1; }
    }
  }
  g.objects = objs;
  g.n_objects = n_objects;
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#else
  (void)n_threads;
#endif
  n_px = (size_t)w * h;
  px = (pixel_out*)calloc(n_px, sizeof(pixel_out));
  switch (params->generator) {
    case ATMRT_GEN_FAST: generate_fast(&g, c0, c1, px); break;
    case ATMRT_GEN_RECTILINEAR: generate_rectilinear(&g, c0, c1, px); break;
    default: generate_interpolating(&g, c0, c1, px); break;
  }
  memset(out, 0, sizeof *out);
  out->width = (uint32_t)w;
  out->height = (uint32_t)h;
  out->n_pixels = n_px;
  for (i = 0; i < n_px; i++) n_hits += px[i].tps.n;
  out->n_hits = n_hits;
  out->azimuth = (double*)malloc(n_px * sizeof(double));
  out->elevation_angle = (double*)malloc(n_px * sizeof(double));
  out->hit_count = (uint32_t*)malloc(n_px * sizeof(uint32_t));
  out->hit_offset = (uint64_t*)malloc(n_px * sizeof(uint64_t));
  out->lat = (double*)malloc((n_hits + 1) * sizeof(double));
  out->lon = (double*)malloc((n_hits + 1) * sizeof(double));
  out->distance = (double*)malloc((n_hits + 1) * sizeof(double));
  out->elevation = (double*)malloc((n_hits + 1) * sizeof(double));
  out->path_length = (double*)malloc((n_hits + 1) * sizeof(double));
  out->normal = (double*)malloc((n_hits + 1) * 3 * sizeof(double));
  out->color_tag = (uint32_t*)malloc((n_hits + 1) * sizeof(uint32_t));
  out->rgba = (double*)malloc((n_hits + 1) * 4 * sizeof(double));
  for (i = 0; i < n_px; i++) {
    size_t k;
    out->azimuth[i] = px[i].azimuth;
    out->elevation_angle[i] = px[i].elevation_angle;
    out->hit_count[i] = (uint32_t)px[i].tps.n;
    out->hit_offset[i] = off;
    out->ray_steps += px[i].steps;
    for (k = 0; k < px[i].tps.n; k++, off++) {
      const trace_point* t = &px[i].tps.v[k];
      out->lat[off] = t->lat;
      out->lon[off] = t->lon;
      out->distance[off] = t->distance;
      out->elevation[off] = t->elevation;
      out->path_length[off] = t->path_length;
      out->normal[3 * off + 0] = t->normal.x;
      out->normal[3 * off + 1] = t->normal.y;
      out->normal[3 * off + 2] = t->normal.z;
      out->color_tag[off] = t->tag;
      memcpy(&out->rgba[4 * off], t->rgba, 4 * sizeof(double));
    }
    free(px[i].tps.v);
  }
  free(px);
  free(objs);
  oracle_atm_free(&g.atm);
  return 0;
}

void oracle_result_free(atmrt_result_t* r) {
  free(r->azimuth); free(r->elevation_angle); free(r->hit_count); free(r->hit_offset);
  free(r->lat); free(r->lon); free(r->distance); free(r->elevation); free(r->path_length);
  free(r->normal); free(r->color_tag); free(r->rgba);
  memset(r, 0, sizeof *r);
}

/* output-ray-paths inner loop, ray_path.rs:65-103 (sampling every step; the caller decimates) */
int oracle_ray_paths(const atmrt_params_t* params, const atmrt_atmosphere_t* atm, double h0, size_t n_angles,
                     const double* angles_deg, int straight, double step, size_t n_steps, double* x, double* h) {
  oracle_env_atm env;
  double radius;
  int spherical = oracle_to_shape(&params->earth, &radius);
  size_t a, k;
  if (oracle_atm_compile(atm, params->wavelength, &env)) {
    oracle_atm_free(&env);
    return -1;
  }
  for (a = 0; a < n_angles; a++) {
    oracle_stepper s;
    oracle_stepper_init(&s, &env, spherical, radius, h0, om_to_radians(angles_deg[a]), straight, step);
    x[a * (n_steps + 1)] = 0.0;
    h[a * (n_steps + 1)] = h0;
    for (k = 1; k <= n_steps; k++) {
      oracle_ray_state st = oracle_stepper_next(&s);
      x[a * (n_steps + 1) + k] = st.x;
      h[a * (n_steps + 1) + k] = st.h;
    }
  }
  oracle_atm_free(&env);
  return 0;
}
