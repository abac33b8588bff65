"""Synthetic inputs of SURVEY.md §8(d): seeded terrain on the DTED lattice, DTED file writer and
the scenes S1..S5 that tests and bench.py use (the reference ships no data)."""
import os

import numpy as np

from . import _abi
from .config import Config

SEED = 1234


def _vnoise(u, v, seed):
    """Value noise: bilinear blend of a seeded lattice of uniforms."""
    iu, iv = np.floor(u).astype(np.int64), np.floor(v).astype(np.int64)
    fu, fv = u - iu, v - iv

    def lattice(i, j):
        h = (i * 73856093) ^ (j * 19349663) ^ (seed * 83492791)
        h = (h ^ (h >> 13)) * 1274126177
        return ((h ^ (h >> 16)) & 0xFFFF) / 65535.0 * 2.0 - 1.0

    a, b = lattice(iu, iv), lattice(iu + 1, iv)
    c, d = lattice(iu, iv + 1), lattice(iu + 1, iv + 1)
    su, sv = fu * fu * (3 - 2 * fu), fv * fv * (3 - 2 * fv)
    return (a * (1 - su) + b * su) * (1 - sv) + (c * (1 - su) + d * su) * sv


def _vnoise_grid(u_row, v_col, seed):
    """_vnoise(u_row + 0 * v_col, v_col + 0 * u_row, seed) for a row vector u_row [1][n] and a column vector v_col [n][1]: the
    same arithmetic per element, but the lattice hash is evaluated once per lattice cell instead of once per post."""
    iu, iv = np.floor(u_row).astype(np.int64), np.floor(v_col).astype(np.int64)
    fu, fv = u_row - iu, v_col - iv
    ui, uinv = np.unique(iu.ravel(), return_inverse=True)
    vi, vinv = np.unique(iv.ravel(), return_inverse=True)

    def lattice(i, j):
        h = (i * 73856093) ^ (j * 19349663) ^ (seed * 83492791)
        h = (h ^ (h >> 13)) * 1274126177
        return ((h ^ (h >> 16)) & 0xFFFF) / 65535.0 * 2.0 - 1.0

    ue = np.append(ui, ui[-1] + 1)  # cells and their right / upper neighbours
    ve = np.append(vi, vi[-1] + 1)
    assert np.array_equal(ue[:-1] + 1, ue[1:]) and np.array_equal(ve[:-1] + 1, ve[1:])
    lat = lattice(ue[None, :], ve[:, None])
    r, c = vinv.ravel(), uinv.ravel()
    lo, hi = lat[r], lat[r + 1]  # [n][cells]: rows first, then the column gather along the contiguous axis
    a, b, cc, d = lo[:, c], lo[:, c + 1], hi[:, c], hi[:, c + 1]
    su, sv = fu * fu * (3 - 2 * fu), fv * fv * (3 - 2 * fv)
    return (a * (1 - su) + b * su) * (1 - sv) + (cc * (1 - su) + d * su) * sv


def synth_tile(lat0, lon0, level=1, seed=SEED, mosaic=(44, 6, 5, 5)):
    """int16 posts [n][n] (south->north, west->east) of one 1-degree cell.  (u, v) is the fractional
    position over the whole mosaic (lat_min, lon_min, n_lat_cells, n_lon_cells) so shared edges agree."""
    n = {1: 1201, 2: 3601}[level] if level in (1, 2) else int(level)
    lat = lat0 + np.arange(n) / (n - 1)
    lon = lon0 + np.arange(n) / (n - 1)
    v_all = ((lat - mosaic[0]) / mosaic[2])[:, None]
    u = ((lon - mosaic[1]) / mosaic[3])[None, :]
    out = np.empty((n, n), dtype=np.int16)
    for r0 in range(0, n, 128):  # row blocks that stay in cache (a level-2 tile is 13 M posts); element-wise arithmetic
        v = v_all[r0:r0 + 128]
        e = (800.0 + 600.0 * np.sin(2 * np.pi * 3 * u) * np.cos(2 * np.pi * 2 * v) + 300.0 * np.sin(2 * np.pi * 11 * (u + v))
             + 120.0 * _vnoise_grid(64 * u, 64 * v, seed))
        out[r0:r0 + 128] = np.clip(np.rint(e), 0, 4000).astype(np.int16)
    return out


def synth_tiles(lat_range, lon_range, level=1, seed=SEED):
    return {(la, lo): synth_tile(la, lo, level, seed) for la in lat_range for lo in lon_range}


def write_dted(path, lat0, lon0, posts):
    """Minimal DTED writer (MIL-PRF-89020B: UHL/DSI/ACC, 0xAA records, big-endian signed magnitude)."""
    posts = np.asarray(posts, dtype=np.int16)
    n_lat, n_lon = posts.shape
    hdr = bytearray(b" " * 3428)

    def ang(deg, is_lat):
        hemi = ("S" if deg < 0 else "N") if is_lat else ("W" if deg < 0 else "E")
        return f"{abs(deg):03d}0000{hemi}".encode()

    hdr[0:4] = b"UHL1"
    hdr[4:12] = ang(lon0, False)
    hdr[12:20] = ang(lat0, True)
    hdr[20:24] = f"{36000 // (n_lon - 1):04d}".encode()
    hdr[24:28] = f"{36000 // (n_lat - 1):04d}".encode()
    hdr[28:32] = b"NA  "
    hdr[32:35] = b"U  "
    hdr[47:51] = f"{n_lon:04d}".encode()
    hdr[51:55] = f"{n_lat:04d}".encode()
    hdr[55:56] = b"0"
    hdr[80:84] = b"DSIU"
    hdr[728:731] = b"ACC"
    mag = np.abs(posts.astype(np.int32)).astype(np.uint16) | np.where(posts < 0, 0x8000, 0).astype(np.uint16)
    with open(path, "wb") as f:
        f.write(bytes(hdr))
        for j in range(n_lon):
            body = bytes([0xAA, (j >> 16) & 0xFF, (j >> 8) & 0xFF, j & 0xFF, (j >> 8) & 0xFF, j & 0xFF, 0, 0])
            body += mag[:, j].astype(">u2").tobytes()
            f.write(body + int(sum(body)).to_bytes(4, "big"))


def write_terrain_dir(path, tiles):
    os.makedirs(path, exist_ok=True)
    for (la, lo), posts in tiles.items():
        name = f"{'n' if la >= 0 else 's'}{abs(la):02d}_{'e' if lo >= 0 else 'w'}{abs(lo):03d}.dt{1 if posts.shape[0] == 1201 else 2}"
        write_dted(os.path.join(path, name), la, lo, posts)


def scene(name, width=None, height=None, generator="Fast", step=None, level=1, **over):
    """Scenes S1..S5 / headline of SURVEY.md §8(d).  Returns (Config, tiles dict)."""
    base = {
        "S1": dict(tiles=None, pos=(0.5, 0.5, ("Absolute", 100.0)), dir=90.0, fov=60.0, tilt=-5.0, w=256, h=128, step=100.0,
                   maxd=50_000.0, straight=True),
        "S2": dict(tiles=([46], [8]), pos=(46.5, 8.5, ("Relative", 50.0)), dir=0.0, fov=60.0, tilt=0.0, w=1024, h=512,
                   step=100.0, maxd=200_000.0, straight=False),
        "S3": dict(tiles=([45, 46, 47], [7, 8, 9]), pos=(46.5, 8.5, ("Relative", 500.0)), dir=0.0, fov=120.0, tilt=-3.0,
                   w=4096, h=2048, step=50.0, maxd=200_000.0, straight=False),
        "S4": dict(tiles=([44, 45, 46, 47, 48], [6, 7, 8, 9, 10]), pos=(46.5, 8.5, ("Relative", 500.0)), dir=0.0, fov=120.0,
                   tilt=-3.0, w=8192, h=4096, step=100.0, maxd=200_000.0, straight=False),
    }
    base["headline"] = dict(base["S3"], step=100.0)
    b = base[name]
    d = {
        "view": {"position": {"latitude": b["pos"][0], "longitude": b["pos"][1], "altitude": {b["pos"][2][0]: b["pos"][2][1]}},
                 "frame": {"direction": b["dir"], "fov": b["fov"], "tilt": b["tilt"], "max_distance": b["maxd"]}},
        "earth_shape": {"Spherical": {"radius": 6371000.0}},
        "straight_rays": b["straight"],
        "simulation_step": step if step is not None else b["step"],
        "output": {"width": width or b["w"], "height": height or b["h"], "generator": generator},
    }
    for k, v in over.items():
        if k in ("earth_shape", "straight_rays", "simulation_step", "wavelength", "atmosphere"):
            d[k] = v
        elif k == "terrain_alpha":
            d.setdefault("scene", {})["terrain_alpha"] = v
        elif k in ("direction", "fov", "tilt", "max_distance"):
            d["view"]["frame"][k] = v
        else:
            raise KeyError(k)
    cfg = Config.from_dict(d)
    tiles = synth_tiles(*b["tiles"], level=level) if b["tiles"] else {}
    return cfg, tiles


def checker_texture(n=64, seed=4321):
    """64x64 RGBA checker with a transparent border ring and a translucent band (exercises alpha 0 / partial / 1)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:n, 0:n]
    tex = np.zeros((n, n, 4), dtype=np.uint8)
    cell = ((xx // 8) + (yy // 8)) % 2
    tex[..., 0] = np.where(cell, 230, 40)
    tex[..., 1] = np.where(cell, 200, 60) + rng.integers(0, 20, (n, n))
    tex[..., 2] = np.where(cell, 30, 180)
    tex[..., 3] = 255
    tex[n // 2 - 4:n // 2 + 4, :, 3] = 128   # translucent band
    tex[:3, :, 3] = 0                        # fully transparent ring: skipped trace points (utils.rs:258-260)
    tex[-3:, :, 3] = 0
    tex[:, :3, 3] = 0
    tex[:, -3:, 3] = 0
    return np.ascontiguousarray(tex)


def add_objects(cfg, n_cyl=700, n_bill=300, seed=4321, dist=(1_000.0, 150_000.0), spread_deg=60.0, radius=(5.0, 50.0),
                height=(20.0, 200.0), bill_w=(20.0, 200.0), bill_h=(10.0, 100.0)):
    """Scene S5 of SURVEY.md §8(d): objects placed by default_rng(seed) uniformly in azimuth (direction +- spread)
    and distance from the observer, on the terrain (Relative 0)."""
    import ctypes as C
    rng = np.random.default_rng(seed)
    p = cfg.params
    lat0, lon0 = np.radians(p.position.latitude), np.radians(p.position.longitude)
    R = 6371000.0
    tex = checker_texture()
    cfg._keepalive.append(tex)
    objs = []
    for i in range(n_cyl + n_bill):
        az = np.radians(p.frame.direction + rng.uniform(-spread_deg, spread_deg))
        s = rng.uniform(*dist) / R
        lat = np.arcsin(np.sin(lat0) * np.cos(s) + np.cos(lat0) * np.sin(s) * np.cos(az))
        lon = lon0 + np.arctan2(np.sin(az) * np.sin(s) * np.cos(lat0), np.cos(s) - np.sin(lat0) * np.sin(lat))
        o = _abi.Object()
        o.position.latitude, o.position.longitude = float(np.degrees(lat)), float(np.degrees(lon))
        o.position.altitude_kind, o.position.altitude = _abi.ALT_RELATIVE, 0.0
        if i < n_cyl:
            kind = i % 3  # cylinder, cone, frustum
            r = float(rng.uniform(*radius))
            o.kind, o.r1, o.height = _abi.OBJ_FRUSTUM, r, float(rng.uniform(*height))
            o.r2 = r if kind == 0 else 0.0 if kind == 1 else 0.4 * r
            col = rng.uniform(0.0, 1.0, 3)
            o.color[0], o.color[1], o.color[2] = (float(v) for v in col)
            o.color[3] = 1.0 if rng.uniform() < 0.5 else 0.5
        else:
            o.kind, o.width, o.height = _abi.OBJ_BILLBOARD, float(rng.uniform(*bill_w)), float(rng.uniform(*bill_h))
            o.texture_rgba = tex.ctypes.data_as(C.POINTER(C.c_uint8))
            o.texture_width = o.texture_height = tex.shape[0]
        objs.append(o)
    cfg.objects = objs
    return cfg
