"""SURVEY §8(f) rank 4, GeoTIFF terrain tiles (src/terrain/geotiff.rs over the absent crate geotiff-rs): the host-side TIFF reader
decodes every layout it claims (strips / tiles, none / LZW / Deflate / PackBits, horizontal predictor, both byte orders, signed
and unsigned samples), the file-name rule is the reference's regex, and — on the GPU — a directory mixing a DTED file and a
GeoTIFF gives the elevations of geotiff.rs:61-100 (file row = latitude index)."""
import ctypes as C
import os
import struct
import zlib

import numpy as np
import pytest
from PIL import Image

import cbuild


@pytest.fixture(scope="module")
def tiff():
    return C.CDLL(cbuild.tiff_export())


def read(lib, path, want):
    out = np.zeros((want, want), dtype=np.int16)
    err = C.create_string_buffer(256)
    ok = lib.t_tiff_read(str(path).encode(), want, out.ctypes.data_as(C.c_void_p), err, 256)
    return (out if ok else None), err.value.decode()


def dem(n, seed=3):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:n, 0:n]
    return (900 + 500 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + rng.integers(-40, 40, (n, n))).astype(np.int16)


def write_tiled(path, a, tile=64, big=False, compress=False):
    """A classic tiled TIFF written by hand (PIL writes strips only): int16, optional Deflate, either byte order."""
    h, w = a.shape
    e = ">" if big else "<"
    tx, ty = -(-w // tile), -(-h // tile)
    chunks = []
    for j in range(ty):
        for i in range(tx):
            t = np.zeros((tile, tile), dtype=np.int16)
            blk = a[j * tile:(j + 1) * tile, i * tile:(i + 1) * tile]
            t[:blk.shape[0], :blk.shape[1]] = blk
            raw = t.astype(e + "i2").tobytes()
            chunks.append(zlib.compress(raw) if compress else raw)
    entries = [(256, 4, [w]), (257, 4, [h]), (258, 3, [16]), (259, 3, [8 if compress else 1]), (262, 3, [1]), (277, 3, [1]),
               (322, 4, [tile]), (323, 4, [tile]), (339, 3, [2])]
    n_entries = len(entries) + 2
    ifd_off = 8
    data_off = ifd_off + 2 + 12 * n_entries + 4
    off_table = data_off
    cnt_table = off_table + 4 * len(chunks)
    pos = cnt_table + 4 * len(chunks)
    offs = []
    for c in chunks:
        offs.append(pos)
        pos += len(c)
    entries += [(324, 4, offs), (325, 4, [len(c) for c in chunks])]
    entries.sort()
    out = bytearray((b"MM" if big else b"II") + struct.pack(e + "HI", 42, ifd_off) + struct.pack(e + "H", n_entries))
    for tag, typ, vals in entries:
        out += struct.pack(e + "HHI", tag, typ, len(vals))
        if len(vals) == 1:
            out += struct.pack(e + ("HH" if typ == 3 else "I"), *((vals[0], 0) if typ == 3 else (vals[0],)))
        else:
            out += struct.pack(e + "I", off_table if tag == 324 else cnt_table)
    out += struct.pack(e + "I", 0)
    out += struct.pack(e + f"{len(offs)}I", *offs) + struct.pack(e + f"{len(chunks)}I", *[len(c) for c in chunks])
    for c in chunks:
        out += c
    open(path, "wb").write(bytes(out))


@pytest.mark.parametrize("compression", [None, "tiff_lzw", "tiff_adobe_deflate", "packbits"])
def test_strip_layouts_written_by_pil(tiff, tmp_path, compression):
    a = dem(300)
    img = Image.fromarray(a.astype(np.uint16))  # unsigned samples; all values positive here
    p = tmp_path / "N46E008.tif"
    img.save(p, compression=compression)
    got, err = read(tiff, p, 300)
    assert got is not None, err
    assert np.array_equal(got, a)
    sub, _ = read(tiff, p, 257)  # the first want x want samples
    assert np.array_equal(sub, a[:257, :257])
    assert read(tiff, p, 301)[0] is None  # too small for the request


def test_lzw_with_horizontal_predictor(tiff, tmp_path):
    a = dem(200, seed=5)
    p = tmp_path / "pred.tif"
    Image.fromarray(a.astype(np.uint16)).save(p, compression="tiff_lzw", tiffinfo={317: 2})
    got, err = read(tiff, p, 200)
    assert got is not None, err
    assert np.array_equal(got, a)


@pytest.mark.parametrize("big,compress", [(False, False), (True, False), (False, True), (True, True)])
def test_tiled_signed_both_byte_orders(tiff, tmp_path, big, compress):
    a = dem(150, seed=7) - 1200  # negative elevations: signed samples
    assert a.min() < 0
    p = tmp_path / "tiled.tif"
    write_tiled(p, a, tile=64, big=big, compress=compress)
    got, err = read(tiff, p, 150)
    assert got is not None, err
    assert np.array_equal(got, a)


def test_rejects_what_it_does_not_support(tiff, tmp_path):
    p = tmp_path / "rgb.tif"
    Image.fromarray(np.zeros((8, 8, 3), dtype=np.uint8)).save(p)
    assert read(tiff, p, 8)[0] is None
    q = tmp_path / "junk.tif"
    q.write_bytes(b"not a tiff at all")
    assert read(tiff, q, 8)[0] is None
    f = tmp_path / "float.tif"
    Image.fromarray(np.zeros((8, 8), dtype=np.float32)).save(f)
    assert read(tiff, f, 8)[0] is None


@pytest.mark.parametrize("name,want", [("N46E008.tif", (46, 8)), ("srtm_S03W071_v3.tiff", (-3, -71)), ("n46e008.tif", None),
                                       ("ALPSMLC30_N046E008_DSM.tif", (46, 8)), ("N99999E1.tif", None), ("readme.txt", None),
                                       ("xNyyN12E3", (12, 3))])
def test_tile_coordinates_from_the_file_name(tiff, name, want):
    """GeoTiffWrapper::coords_from_name (geotiff.rs:15-31): leftmost (N|S)digits(E|W)digits, case-sensitive, i16 range."""
    lat, lon = C.c_int(), C.c_int()
    ok = tiff.t_tiff_coords(name.encode(), C.byref(lat), C.byref(lon))
    assert (bool(ok), (lat.value, lon.value) if ok else None) == (want is not None, want)


@pytest.mark.gpu
def test_directory_with_dted_and_geotiff(tmp_path, oracle_det):
    """Terrain::from_folder (terrain/mod.rs:66-118): a DTED tile and a 3601 x 3601 GeoTIFF tile side by side; the GeoTIFF cell is
    sampled with geotiff.rs's bilinear on the 3600-interval grid, file row = latitude index."""
    from atm_raytracer_amd import generators, synth
    ctx = generators.Context(0)
    tiles = synth.synth_tiles([46], [8, 9], level=2)
    (k_dted, dted_posts), (k_tif, tif_posts) = sorted(tiles.items())
    synth.write_dted(str(tmp_path / "n46_e008_1arc_v3.dt2"), k_dted[0], k_dted[1], dted_posts)
    assert tif_posts.shape == (3601, 3601)
    Image.fromarray(np.ascontiguousarray(tif_posts).astype(np.uint16)).save(tmp_path / f"N{k_tif[0]:02d}E{k_tif[1]:03d}.tif",
                                                                                           compression="tiff_adobe_deflate")
    (tmp_path / "N47E008.tif").write_bytes(b"II*\x00 truncated")  # undecodable: the cell stays empty, no error (lazy-load semantics)
    terrain = generators.Terrain.from_folder(str(tmp_path), ctx)
    rng = np.random.default_rng(9)
    lat = np.concatenate([rng.uniform(46.0, 47.0, 400), [46.0, 47.0, 46.5, 47.3]])
    lon = np.concatenate([rng.uniform(8.0, 10.0, 400), [9.0, 10.0, 9.999999, 8.5]])
    got, valid = terrain.get_elev(lat, lon)
    assert terrain.n_files == 3
    t = oracle_det.terrain_new(tiles)
    want = [oracle_det.get_elev(t, float(a), float(b)) for a, b in zip(lat, lon)]
    oracle_det.terrain_free(t)
    for g, v, w in zip(got, valid, want):
        assert (w is None and not v) or (v and g == w)
    assert valid.sum() >= 400 and not valid[-1]  # (47.3, 8.5) lies in the cell of the undecodable file
    ctx.close()
