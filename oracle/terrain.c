/* terrain.c — tile store, DTED reader/writer, bilinear sampler.  ORACLE (test infrastructure).
 *
 * In-repo parts restated from src/terrain/mod.rs (tile map keyed by integer degrees, lazy load
 * is irrelevant to results) and tile.rs.  The DTED parser and `DtedData::get_elev` live in crate
 * `dted` 0.2 (source absent, PARITY UNPINNED): the parser follows the public format
 * MIL-PRF-89020B (UHL/DSI/ACC headers, 0xAA data records, big-endian signed-magnitude posts) and
 * the sampler is modelled on the reference's only in-repo bilinear sampler,
 * GeoTiffWrapper::get_elev (terrain/geotiff.rs:61-100), generalised from 3600 to (n-1) intervals.
 * Void posts (-32767) are returned as stored.
 */
#include "oracle.h"
#include "oracle_math.h"

#include <dirent.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

oracle_terrain* oracle_terrain_new(void) { return (oracle_terrain*)calloc(1, sizeof(oracle_terrain)); }

void oracle_terrain_free(oracle_terrain* t) {
  int i;
  if (!t) return;
  for (i = 0; i < t->n_tiles; i++) free(t->tiles[i].posts);
  free(t->tiles);
  free(t);
}

int oracle_terrain_add_tile(oracle_terrain* t, int lat0, int lon0, int n_lat, int n_lon, const int16_t* posts) {
  oracle_tile* tile;
  int i;
  if (n_lat < 2 || n_lon < 2 || !posts) return -1;
  /* HashMap::insert replaces an existing key (terrain/mod.rs:93-96) */
  for (i = 0; i < t->n_tiles; i++)
    if (t->tiles[i].lat0 == lat0 && t->tiles[i].lon0 == lon0) break;
  if (i == t->n_tiles) {
    if (t->n_tiles == t->cap) {
      t->cap = t->cap ? 2 * t->cap : 16;
      t->tiles = (oracle_tile*)realloc(t->tiles, (size_t)t->cap * sizeof(oracle_tile));
    }
    t->n_tiles++;
    t->tiles[i].posts = NULL;
  }
  tile = &t->tiles[i];
  free(tile->posts);
  tile->lat0 = lat0;
  tile->lon0 = lon0;
  tile->n_lat = n_lat;
  tile->n_lon = n_lon;
  tile->posts = (int16_t*)malloc((size_t)n_lat * n_lon * sizeof(int16_t));
  memcpy(tile->posts, posts, (size_t)n_lat * n_lon * sizeof(int16_t));
  return 0;
}

/* Rust `f as i16`: saturating, NaN -> 0 */
static int sat_i16(double f) {
  if (f != f) return 0;
  if (f <= -32768.0) return -32768;
  if (f >= 32767.0) return 32767;
  return (int)f;
}
/* Rust `f as usize`: saturating, NaN -> 0, negatives -> 0 */
static long sat_usize(double f) {
  if (f != f || f <= 0.0) return 0;
  if (f >= 2147483647.0) return 2147483647L;
  return (long)f;
}

/* Tile::get_elev -> DtedData::get_elev (tile.rs:28-30; model geotiff.rs:61-100) */
static int tile_get_elev(const oracle_tile* t, double lat, double lon, double* out) {
  double min_lat = (double)t->lat0, min_lon = (double)t->lon0;
  double max_lat = min_lat + 1.0, max_lon = min_lon + 1.0;
  double flat, flon, lat_frac, lon_frac, e00, e01, e10, e11;
  long lat_int, lon_int;
  if (lat < min_lat || lat > max_lat || lon < min_lon || lon > max_lon) return 0;
  flat = (lat - min_lat) * (double)(t->n_lat - 1);
  flon = (lon - min_lon) * (double)(t->n_lon - 1);
  lat_int = sat_usize(flat);
  lon_int = sat_usize(flon);
  lat_frac = flat - (double)lat_int;
  lon_frac = flon - (double)lon_int;
  /* handle the edge case of max lat/lon (geotiff.rs:77-85) */
  if (lat_int == t->n_lat - 1) {
    lat_int -= 1;
    lat_frac += 1.0;
  }
  if (lon_int == t->n_lon - 1) {
    lon_int -= 1;
    lon_frac += 1.0;
  }
  e00 = (double)t->posts[lat_int * t->n_lon + lon_int];
  e01 = (double)t->posts[(lat_int + 1) * t->n_lon + lon_int];
  e10 = (double)t->posts[lat_int * t->n_lon + lon_int + 1];
  e11 = (double)t->posts[(lat_int + 1) * t->n_lon + lon_int + 1];
  *out = e00 * (1.0 - lon_frac) * (1.0 - lat_frac) + e01 * (1.0 - lon_frac) * lat_frac +
         e10 * lon_frac * (1.0 - lat_frac) + e11 * lon_frac * lat_frac;
  return 1;
}

/* Terrain::get_elev, terrain/mod.rs:120-126 */
int oracle_terrain_get_elev(const oracle_terrain* t, double latitude, double longitude, double* elev) {
  int lat = sat_i16(om_floor(latitude));
  int lon = sat_i16(om_floor(longitude));
  int i;
  for (i = 0; i < t->n_tiles; i++)
    if (t->tiles[i].lat0 == lat && t->tiles[i].lon0 == lon) return tile_get_elev(&t->tiles[i], latitude, longitude, elev);
  return 0;
}

/* ---- DTED (MIL-PRF-89020B) ---------------------------------------------------------------- */

#define DTED_DATA_OFFSET 3428 /* UHL 80 + DSI 648 + ACC 2700 */

static int parse_int(const unsigned char* p, int n) {
  int v = 0, i;
  for (i = 0; i < n; i++) {
    if (p[i] < '0' || p[i] > '9') return -1;
    v = v * 10 + (p[i] - '0');
  }
  return v;
}

/* DDDMMSSH -> degrees; returns 0 on success */
static int parse_angle(const unsigned char* p, double* deg) {
  int d = parse_int(p, 3), m = parse_int(p + 3, 2), s = parse_int(p + 5, 2);
  double v;
  if (d < 0 || m < 0 || s < 0) return -1;
  v = (double)d + (double)m / 60.0 + (double)s / 3600.0;
  if (p[7] == 'S' || p[7] == 'W') v = -v;
  else if (p[7] != 'N' && p[7] != 'E') return -1;
  *deg = v;
  return 0;
}

int oracle_dted_read(const char* path, int* lat0, int* lon0, int* n_lat, int* n_lon, int16_t** posts) {
  FILE* f = fopen(path, "rb");
  unsigned char uhl[80];
  unsigned char* rec = NULL;
  double olat, olon;
  int nlon, nlat, i, j, rc = -1;
  int16_t* out = NULL;
  if (!f) return -1;
  if (fread(uhl, 1, 80, f) != 80 || memcmp(uhl, "UHL1", 4) != 0) goto done;
  if (parse_angle(uhl + 4, &olon) || parse_angle(uhl + 12, &olat)) goto done;
  nlon = parse_int(uhl + 47, 4);
  nlat = parse_int(uhl + 51, 4);
  if (nlon < 2 || nlat < 2) goto done;
  if (posts) {
    size_t rec_size = 12 + 2 * (size_t)nlat;
    rec = (unsigned char*)malloc(rec_size);
    out = (int16_t*)malloc((size_t)nlat * nlon * sizeof(int16_t));
    if (fseek(f, DTED_DATA_OFFSET, SEEK_SET)) goto done;
    for (j = 0; j < nlon; j++) {
      if (fread(rec, 1, rec_size, f) != rec_size || rec[0] != 0xAA) goto done;
      for (i = 0; i < nlat; i++) {
        unsigned v = ((unsigned)rec[8 + 2 * i] << 8) | rec[9 + 2 * i];
        int e = (int)(v & 0x7fff);
        if (v & 0x8000) e = -e; /* signed magnitude */
        out[(size_t)i * nlon + j] = (int16_t)e;
      }
    }
    *posts = out;
    out = NULL;
  }
  /* `f64::from(header.origin_lat) as i16` — truncation (terrain/mod.rs:91-92) */
  *lat0 = sat_i16(olat);
  *lon0 = sat_i16(olon);
  *n_lat = nlat;
  *n_lon = nlon;
  rc = 0;
done:
  free(rec);
  free(out);
  fclose(f);
  return rc;
}

static void fmt_angle(char* dst, int deg, int is_lat) {
  char hemi = is_lat ? (deg < 0 ? 'S' : 'N') : (deg < 0 ? 'W' : 'E');
  char buf[16];
  snprintf(buf, sizeof buf, "%03d0000%c", abs(deg), hemi);
  memcpy(dst, buf, 8);
}

int oracle_dted_write(const char* path, int lat0, int lon0, int n_lat, int n_lon, const int16_t* posts) {
  FILE* f = fopen(path, "wb");
  unsigned char* hdr;
  unsigned char* rec;
  size_t rec_size = 12 + 2 * (size_t)n_lat;
  char num[16];
  int i, j;
  if (!f) return -1;
  hdr = (unsigned char*)malloc(DTED_DATA_OFFSET);
  memset(hdr, ' ', DTED_DATA_OFFSET);
  memcpy(hdr, "UHL1", 4);
  fmt_angle((char*)hdr + 4, lon0, 0);
  fmt_angle((char*)hdr + 12, lat0, 1);
  snprintf(num, sizeof num, "%04d", 36000 / (n_lon - 1)); /* interval in tenths of arc seconds */
  memcpy(hdr + 20, num, 4);
  snprintf(num, sizeof num, "%04d", 36000 / (n_lat - 1));
  memcpy(hdr + 24, num, 4);
  memcpy(hdr + 28, "NA  ", 4);
  memcpy(hdr + 32, "U  ", 3);
  snprintf(num, sizeof num, "%04d", n_lon);
  memcpy(hdr + 47, num, 4);
  snprintf(num, sizeof num, "%04d", n_lat);
  memcpy(hdr + 51, num, 4);
  hdr[55] = '0';
  memcpy(hdr + 80, "DSIU", 4);
  memcpy(hdr + 80 + 648, "ACC", 3);
  fwrite(hdr, 1, DTED_DATA_OFFSET, f);
  free(hdr);
  rec = (unsigned char*)malloc(rec_size);
  for (j = 0; j < n_lon; j++) {
    unsigned sum = 0;
    size_t k;
    rec[0] = 0xAA;
    rec[1] = (unsigned char)((j >> 16) & 0xff);
    rec[2] = (unsigned char)((j >> 8) & 0xff);
    rec[3] = (unsigned char)(j & 0xff);
    rec[4] = (unsigned char)((j >> 8) & 0xff);
    rec[5] = (unsigned char)(j & 0xff);
    rec[6] = 0;
    rec[7] = 0;
    for (i = 0; i < n_lat; i++) {
      int e = posts[(size_t)i * n_lon + j];
      unsigned v = e < 0 ? (0x8000u | (unsigned)(-e)) : (unsigned)e;
      rec[8 + 2 * i] = (unsigned char)(v >> 8);
      rec[9 + 2 * i] = (unsigned char)(v & 0xff);
    }
    for (k = 0; k < rec_size - 4; k++) sum += rec[k];
    rec[rec_size - 4] = (unsigned char)(sum >> 24);
    rec[rec_size - 3] = (unsigned char)(sum >> 16);
    rec[rec_size - 2] = (unsigned char)(sum >> 8);
    rec[rec_size - 1] = (unsigned char)sum;
    fwrite(rec, 1, rec_size, f);
  }
  free(rec);
  fclose(f);
  return 0;
}

/* Terrain::from_folder, terrain/mod.rs:66-83: every directory entry must be a terrain file */
int oracle_terrain_load_dir(oracle_terrain* t, const char* path) {
  DIR* d = opendir(path);
  struct dirent* ent;
  int files = 0;
  if (!d) return -1;
  while ((ent = readdir(d)) != NULL) {
    char full[4096];
    int lat0, lon0, n_lat, n_lon;
    int16_t* posts = NULL;
    if (!strcmp(ent->d_name, ".") || !strcmp(ent->d_name, "..")) continue;
    snprintf(full, sizeof full, "%s/%s", path, ent->d_name);
    if (oracle_dted_read(full, &lat0, &lon0, &n_lat, &n_lon, &posts)) {
      closedir(d);
      return -2; /* "Could not buffer terrain file" panic, terrain/mod.rs:117 */
    }
    oracle_terrain_add_tile(t, lat0, lon0, n_lat, n_lon, posts);
    free(posts);
    files++;
  }
  closedir(d);
  return files;
}
