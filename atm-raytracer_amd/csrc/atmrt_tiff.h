// atmrt_tiff.h — host-only reader for single-band 16-bit elevation GeoTIFFs, the reference's second terrain format
// (src/terrain/geotiff.rs wraps the absent crate geotiff-rs; src/terrain/mod.rs:100-118 accepts a file as GeoTIFF when its name
// matches (N|S)dd(E|W)ddd and the DTED header parse failed).
//
// What the reference needs from a tile (geotiff.rs:61-100): get_pixel(x, y) for x, y in 0..=3600 with x the longitude index
// and y the LATITUDE index, bilinear on the 3600-interval grid.  So a tile is the file's first 3601 x 3601 samples and file
// row y is latitude row y, exactly as the reference indexes it (whether geotiff-rs flips rows internally cannot be checked —
// the crate is absent; DESIGN.md lists this under the unpinned choices).
//
// Supported here: classic TIFF (not BigTIFF), little or big endian, one sample per pixel, 16 bits (signed or unsigned; values
// above 32767 saturate), strips or tiles, compression none / LZW / Deflate (zlib) / PackBits, predictor none or horizontal.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace atmrt_tiff {

struct Reader {
  const uint8_t* p = nullptr;
  size_t n = 0;
  bool big = false;
  bool ok(size_t off, size_t len) const { return off <= n && len <= n - off; }
  uint16_t u16(size_t off) const { return big ? (uint16_t)(p[off] << 8 | p[off + 1]) : (uint16_t)(p[off] | p[off + 1] << 8); }
  uint32_t u32(size_t off) const {
    return big ? ((uint32_t)p[off] << 24 | (uint32_t)p[off + 1] << 16 | (uint32_t)p[off + 2] << 8 | p[off + 3])
               : ((uint32_t)p[off] | (uint32_t)p[off + 1] << 8 | (uint32_t)p[off + 2] << 16 | (uint32_t)p[off + 3] << 24);
  }
};

// values of one IFD entry as u32 (types BYTE 1, SHORT 3, LONG 4)
inline bool entry_values(const Reader& r, size_t entry, std::vector<uint32_t>& out) {
  const uint16_t type = r.u16(entry + 2);
  const uint32_t count = r.u32(entry + 4);
  const size_t size = type == 1 ? 1 : type == 3 ? 2 : type == 4 ? 4 : 0;
  if (!size || count > (1u << 26)) return false;
  size_t off = entry + 8;
  if ((size_t)count * size > 4) {
    off = r.u32(entry + 8);
    if (!r.ok(off, (size_t)count * size)) return false;
  }
  out.resize(count);
  for (uint32_t i = 0; i < count; i++)
    out[i] = size == 1 ? r.p[off + i] : size == 2 ? r.u16(off + 2 * (size_t)i) : r.u32(off + 4 * (size_t)i);
  return true;
}

// TIFF LZW (MSB-first codes of 9..12 bits, ClearCode 256, EndOfInformation 257, "early change")
inline bool lzw_decode(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect) {
  struct Ent {
    int32_t prev;
    uint8_t ch;
    uint32_t len;
  };
  std::vector<Ent> tab(4096);
  for (int i = 0; i < 256; i++) tab[i] = Ent{-1, (uint8_t)i, 1};
  int next = 258, bits = 9, old = -1;
  uint32_t acc = 0;
  int nacc = 0;
  size_t pos = 0;
  out.clear();
  out.reserve(expect);
  std::vector<uint8_t> tmp;
  while (out.size() < expect) {
    while (nacc < bits) {
      if (pos >= n) return out.size() >= expect;
      acc = acc << 8 | src[pos++];
      nacc += 8;
    }
    const int code = (int)(acc >> (nacc - bits)) & ((1 << bits) - 1);
    nacc -= bits;
    if (code == 257) break;
    if (code == 256) {
      next = 258;
      bits = 9;
      old = -1;
      continue;
    }
    int cur = code;
    if (old < 0) {
      if (code >= 256) return false;
      out.push_back((uint8_t)code);
      old = code;
      continue;
    }
    uint8_t first;
    if (code < next) {
      tmp.resize(tab[code].len);
      for (int c = code, k = (int)tab[code].len - 1; c >= 0; c = tab[c].prev, k--) tmp[k] = tab[c].ch;
      first = tmp[0];
    } else if (code == next) { // KwKwK
      tmp.resize(tab[old].len + 1);
      for (int c = old, k = (int)tab[old].len - 1; c >= 0; c = tab[c].prev, k--) tmp[k] = tab[c].ch;
      first = tmp[0];
      tmp[tab[old].len] = first;
    } else {
      return false;
    }
    out.insert(out.end(), tmp.begin(), tmp.end());
    if (next < 4096) {
      tab[next] = Ent{old, first, tab[old].len + 1};
      next++;
      if (next + 1 >= (1 << bits) && bits < 12) bits++;
    }
    old = cur;
  }
  return out.size() >= expect;
}

inline bool packbits_decode(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect) {
  out.clear();
  out.reserve(expect);
  size_t i = 0;
  while (i < n && out.size() < expect) {
    const int8_t c = (int8_t)src[i++];
    if (c >= 0) {
      const size_t len = (size_t)c + 1;
      if (i + len > n) return false;
      out.insert(out.end(), src + i, src + i + len);
      i += len;
    } else if (c != -128) {
      if (i >= n) return false;
      out.insert(out.end(), (size_t)(1 - c), src[i++]);
    }
  }
  return out.size() >= expect;
}

// One strip or tile -> raw bytes (rows x row_bytes)
inline bool decode_chunk(int compression, const uint8_t* src, size_t n, size_t expect, std::vector<uint8_t>& out) {
  if (compression == 1) {
    if (n < expect) return false;
    out.assign(src, src + expect);
    return true;
  }
  if (compression == 5) return lzw_decode(src, n, out, expect);
  if (compression == 32773) return packbits_decode(src, n, out, expect);
  if (compression == 8 || compression == 32946) {
    out.resize(expect);
    uLongf len = (uLongf)expect;
    int rc = uncompress(out.data(), &len, src, (uLong)n);
    return (rc == Z_OK || rc == Z_BUF_ERROR) && len == expect;
  }
  return false;
}

// Reads the first `want` x `want` samples of the image as int16 (row-major in file order).  Returns false with a message when the
// file is not a TIFF this reader supports or is smaller than want x want.
inline bool read_dem(const std::string& path, int want, std::vector<int16_t>& posts, std::string& err) {
  FILE* fp = fopen(path.c_str(), "rb");
  if (!fp) {
    err = "cannot open";
    return false;
  }
  std::vector<uint8_t> buf;
  fseek(fp, 0, SEEK_END);
  long sz = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  if (sz < 8 || sz > (1L << 31)) {
    fclose(fp);
    err = "not a TIFF (size)";
    return false;
  }
  buf.resize((size_t)sz);
  size_t got = fread(buf.data(), 1, buf.size(), fp);
  fclose(fp);
  if (got != buf.size()) {
    err = "short read";
    return false;
  }
  Reader r;
  r.p = buf.data();
  r.n = buf.size();
  if (buf[0] == 'I' && buf[1] == 'I') r.big = false;
  else if (buf[0] == 'M' && buf[1] == 'M') r.big = true;
  else {
    err = "not a TIFF (byte order mark)";
    return false;
  }
  if (r.u16(2) != 42) {
    err = "not a classic TIFF (BigTIFF is not supported)";
    return false;
  }
  size_t ifd = r.u32(4);
  if (!r.ok(ifd, 2)) {
    err = "bad IFD offset";
    return false;
  }
  const uint16_t n_entries = r.u16(ifd);
  if (!r.ok(ifd + 2, (size_t)n_entries * 12)) {
    err = "truncated IFD";
    return false;
  }
  uint32_t width = 0, height = 0, bits = 1, compression = 1, spp = 1, rows_per_strip = 0xffffffffu, sample_format = 1, predictor = 1,
           tile_w = 0, tile_h = 0, planar = 1;
  std::vector<uint32_t> offsets, counts, tile_offsets, tile_counts, v;
  for (uint16_t i = 0; i < n_entries; i++) {
    const size_t e = ifd + 2 + (size_t)i * 12;
    const uint16_t tag = r.u16(e);
    if (tag != 256 && tag != 257 && tag != 258 && tag != 259 && tag != 273 && tag != 277 && tag != 278 && tag != 279 && tag != 284 &&
        tag != 317 && tag != 322 && tag != 323 && tag != 324 && tag != 325 && tag != 339)
      continue;
    if (!entry_values(r, e, v) || v.empty()) {
      err = "unsupported IFD entry type for tag " + std::to_string(tag);
      return false;
    }
    switch (tag) {
      case 256: width = v[0]; break;
      case 257: height = v[0]; break;
      case 258: bits = v[0]; break;
      case 259: compression = v[0]; break;
      case 273: offsets = v; break;
      case 277: spp = v[0]; break;
      case 278: rows_per_strip = v[0]; break;
      case 279: counts = v; break;
      case 284: planar = v[0]; break;
      case 317: predictor = v[0]; break;
      case 322: tile_w = v[0]; break;
      case 323: tile_h = v[0]; break;
      case 324: tile_offsets = v; break;
      case 325: tile_counts = v; break;
      case 339: sample_format = v[0]; break;
    }
  }
  (void)planar;
  if (spp != 1 || bits != 16 || (sample_format != 1 && sample_format != 2)) {
    err = "only single-band 16-bit integer rasters are supported";
    return false;
  }
  if (predictor != 1 && predictor != 2) {
    err = "unsupported predictor";
    return false;
  }
  if ((int64_t)width < want || (int64_t)height < want || width > 65535 || height > 65535) {
    err = "raster is " + std::to_string(width) + " x " + std::to_string(height) + ", need at least " + std::to_string(want) + " x " +
          std::to_string(want);
    return false;
  }
  posts.assign((size_t)want * want, 0);
  auto sample = [&](const uint8_t* q) -> int16_t {
    uint16_t u = r.big ? (uint16_t)(q[0] << 8 | q[1]) : (uint16_t)(q[0] | q[1] << 8);
    if (sample_format == 1) return u > 32767 ? (int16_t)32767 : (int16_t)u;
    return (int16_t)u;
  };
  // undo the horizontal predictor in place: 16-bit samples accumulate modulo 2^16 in the file's byte order
  auto unpredict = [&](std::vector<uint8_t>& raw, size_t rows, size_t row_samples) {
    if (predictor != 2) return;
    for (size_t y = 0; y < rows; y++) {
      uint8_t* row = raw.data() + y * row_samples * 2;
      uint16_t acc = 0;
      for (size_t x = 0; x < row_samples; x++) {
        uint16_t d = r.big ? (uint16_t)(row[2 * x] << 8 | row[2 * x + 1]) : (uint16_t)(row[2 * x] | row[2 * x + 1] << 8);
        acc = (uint16_t)(acc + d);
        if (r.big) {
          row[2 * x] = (uint8_t)(acc >> 8);
          row[2 * x + 1] = (uint8_t)acc;
        } else {
          row[2 * x] = (uint8_t)acc;
          row[2 * x + 1] = (uint8_t)(acc >> 8);
        }
      }
    }
  };
  std::vector<uint8_t> raw;
  if (tile_w && tile_h && !tile_offsets.empty()) {
    const uint32_t tx = (width + tile_w - 1) / tile_w, ty = (height + tile_h - 1) / tile_h;
    if (tile_offsets.size() < (size_t)tx * ty || tile_counts.size() < (size_t)tx * ty) {
      err = "tile table too short";
      return false;
    }
    for (uint32_t j = 0; j * tile_h < (uint32_t)want; j++)
      for (uint32_t i = 0; i * tile_w < (uint32_t)want; i++) {
        const size_t t = (size_t)j * tx + i;
        if (!r.ok(tile_offsets[t], tile_counts[t]) ||
            !decode_chunk((int)compression, r.p + tile_offsets[t], tile_counts[t], (size_t)tile_w * tile_h * 2, raw)) {
          err = "cannot decode tile " + std::to_string(t);
          return false;
        }
        unpredict(raw, tile_h, tile_w);
        for (uint32_t y = 0; y < tile_h && j * tile_h + y < (uint32_t)want; y++)
          for (uint32_t x = 0; x < tile_w && i * tile_w + x < (uint32_t)want; x++)
            posts[(size_t)(j * tile_h + y) * want + i * tile_w + x] = sample(raw.data() + ((size_t)y * tile_w + x) * 2);
      }
    return true;
  }
  if (offsets.empty() || counts.size() < offsets.size()) {
    err = "no strip table";
    return false;
  }
  if (rows_per_strip == 0 || rows_per_strip > height) rows_per_strip = height;
  for (size_t s = 0; s < offsets.size() && s * rows_per_strip < (size_t)want; s++) {
    const uint32_t y0 = (uint32_t)(s * rows_per_strip);
    const uint32_t rows = y0 + rows_per_strip <= height ? rows_per_strip : height - y0;
    if (!r.ok(offsets[s], counts[s]) || !decode_chunk((int)compression, r.p + offsets[s], counts[s], (size_t)rows * width * 2, raw)) {
      err = "cannot decode strip " + std::to_string(s);
      return false;
    }
    unpredict(raw, rows, width);
    for (uint32_t y = 0; y < rows && y0 + y < (uint32_t)want; y++)
      for (uint32_t x = 0; x < (uint32_t)want; x++) posts[(size_t)(y0 + y) * want + x] = sample(raw.data() + ((size_t)y * width + x) * 2);
  }
  return true;
}

// GeoTiffWrapper::coords_from_name (geotiff.rs:15-31): the first (N|S)digits(E|W)digits in the file name
inline bool coords_from_name(const std::string& file_name, int& lat, int& lon) {
  for (size_t i = 0; i < file_name.size(); i++) {
    if (file_name[i] != 'N' && file_name[i] != 'S') continue;
    size_t j = i + 1;
    long a = 0;
    while (j < file_name.size() && file_name[j] >= '0' && file_name[j] <= '9' && a < 100000) a = a * 10 + (file_name[j++] - '0');
    if (j == i + 1 || j >= file_name.size() || (file_name[j] != 'E' && file_name[j] != 'W')) continue;
    size_t k = j + 1;
    long b = 0;
    while (k < file_name.size() && file_name[k] >= '0' && file_name[k] <= '9' && b < 100000) b = b * 10 + (file_name[k++] - '0');
    if (k == j + 1) continue;
    if (a > 32767 || b > 32767) return false; // i16::from_str fails
    lat = file_name[i] == 'S' ? -(int)a : (int)a;
    lon = file_name[j] == 'W' ? -(int)b : (int)b;
    return true;
  }
  return false;
}

} // namespace atmrt_tiff
