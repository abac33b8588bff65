"""The reference's metadata file: writer and reader.

`generator::output_metadata` (src/generator/mod.rs:26-45) writes gzip(bincode::serialize(&AllData { params, result })) and
`view` loads it (src/viewer/mod.rs:17-29).  bincode 1.x (Cargo.toml:9) with default options is a fixed layout — little-endian
fixed-width integers, f64 as 8 bytes, `bool` one byte, `String`/`Vec` a u64 length + contents, `Option` a u8 tag + value,
structs their fields in declaration order, enums a u32 variant index + the variant's fields — so every type DEFINED IN THE
REFERENCE REPOSITORY has a known encoding, written here field by field with the struct each line restates.

What is NOT known (the byte ranges that stay unpinned; DESIGN.md §6 lists them):

* `Params.env: atm_refraction::Environment` (params.rs:501) — `EarthShape`, `Atmosphere` and their serde layout live in crate
  atm-refraction 0.6, absent from /root/reference.  bincode is not self-describing, so nothing after `env` can be located by
  the reference's reader unless these bytes are exactly right.  This writer emits a DOCUMENTED STAND-IN for the segment
  (`encode_env`: magic "ATMRTENV", the earth shape, the AtmosphereDef the frame was computed with, the wavelength) and takes an
  `env_encoder` hook — the converter stub: once the crate's layout is known, a function producing its bytes from a Config
  makes the whole file readable by `atm-raytracer view`; nothing else changes.
* `nalgebra::Vector3<f64>` (TracePoint.normal, Coloring::Shading.light_dir): nalgebra 0.32's ArrayStorage serializer is
  believed to emit a sequence (u64 length 3 + elements); `vector3_len_prefix=False` writes three bare f64 instead.

The `result` half (Vec<Vec<ResultPixel>>, the bulk of the file) is encoded / decoded by the library
(atmrt_result_encode_bincode / atmrt_result_decode_bincode, csrc/atmrt_metadata.hip)."""
import ctypes as C
import struct
import zlib

import numpy as np

from . import _abi, _lib

ENV_MAGIC = b"ATMRTENV"
EARTH_NAMES = ["SimpleSphere", "Spherical", "Ellipsoid", "Wgs84", "AzimuthalEquidistant", "FlatDistorted", "ObserverAe", "SimpleObserverAe"]


# ---- primitive encoders (bincode 1, default options) ---------------------------------------------
def _u8(v): return struct.pack("<B", int(v))
def _u16(v): return struct.pack("<H", int(v))
def _u32(v): return struct.pack("<I", int(v))
def _u64(v): return struct.pack("<Q", int(v))
def _f64(*v): return struct.pack("<%dd" % len(v), *[float(x) for x in v])
def _string(s): b = s.encode("utf-8"); return _u64(len(b)) + b
def _option(v, enc): return _u8(0) if v is None else _u8(1) + enc(v)
def _vector3(v, prefix): return (_u64(3) if prefix else b"") + _f64(*v)


class _Cursor:
    def __init__(self, data, pos=0):
        self.d, self.p = data, pos

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.d, self.p)
        self.p += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def string(self):
        n = self.take("Q")
        s = bytes(self.d[self.p:self.p + n]).decode("utf-8")
        self.p += n
        return s

    def option(self, dec):
        return dec() if self.take("B") else None

    def vector3(self, prefix):
        if prefix and self.take("Q") != 3:
            raise ValueError("Vector3 length prefix is not 3")
        return list(self.take("3d"))


# ---- Params (params.rs:496-505), field by field ---------------------------------------------------
def _earth_model(e):  # EarthModel, earth_model/mod.rs:18-28
    k = int(e.kind)
    if EARTH_NAMES[k] in ("Spherical", "ObserverAe"):
        return _u32(k) + _f64(e.radius)
    if EARTH_NAMES[k] == "Ellipsoid":
        return _u32(k) + _f64(e.a, e.b)
    return _u32(k)


def _ticks(ticks):  # Vec<Tick> / Vec<VerticalTick>, params.rs:306-367
    out = _u64(len(ticks))
    for t in ticks:
        if t[0] == "Single":
            out += _u32(0) + _f64(t[1]) + _u32(t[2]) + _u8(t[3])
        else:
            out += _u32(1) + _f64(t[1], t[2]) + _u32(t[3]) + _u8(t[4])
    return out


def encode_env(cfg):
    """Stand-in for `env: Environment { shape, atmosphere, wavelength }` (crate atm-refraction absent; see module docstring).
    magic, u32 version, EarthShape {0 Spherical{radius} | 1 Flat} as EarthModel::to_shape gives it (earth_model/mod.rs:95-112),
    the AtmosphereDef (reference README.md:283-323) and the wavelength."""
    p, a = cfg.params, cfg.atmosphere
    name = EARTH_NAMES[p.earth.kind]
    if name in ("SimpleSphere",):
        shape = _u32(0) + _f64(6371000.0)
    elif name == "Spherical":
        shape = _u32(0) + _f64(p.earth.radius)
    elif name in ("Ellipsoid", "Wgs84"):
        ea, eb = (6378137.0, 6356752.314245) if name == "Wgs84" else (p.earth.a, p.earth.b)
        shape = _u32(0) + _f64((2.0 * ea + eb) / 3.0)
    else:
        shape = _u32(1)
    out = ENV_MAGIC + _u32(1) + shape
    out += _f64(a.pressure_altitude, a.pressure) + _u8(a.has_temperature_fixed_point) + _f64(a.temperature_altitude, a.temperature)
    out += _u64(a.n_functions)
    for k in range(a.n_functions):
        fn = a.functions[k]
        out += _f64(fn.altitude) + _u32(fn.kind)
        if fn.kind == _abi.TEMP_LINEAR:
            out += _f64(fn.gradient)
        else:
            out += _u32(fn.boundary) + _f64(fn.bc[0], fn.bc[1]) + _u64(fn.n_points)
            for i in range(fn.n_points):
                out += _f64(fn.point_altitude[i], fn.point_temperature[i])
    return out + _f64(p.wavelength)


def decode_env(cur):
    if bytes(cur.d[cur.p:cur.p + 8]) != ENV_MAGIC:
        raise ValueError("the env segment is not this package's stand-in (a file written by the reference needs an env decoder)")
    cur.p += 8
    env = {"version": cur.take("I")}
    env["shape"] = {"Spherical": {"radius": cur.take("d")}} if cur.take("I") == 0 else "Flat"
    pa, pr = cur.take("2d")
    has = cur.take("B")
    ta, tt = cur.take("2d")
    fns = []
    for _ in range(cur.take("Q")):
        alt, kind = cur.take("d"), cur.take("I")
        if kind == _abi.TEMP_LINEAR:
            fns.append({"altitude": alt, "Linear": {"gradient": cur.take("d")}})
        else:
            boundary, bc0, bc1, n = cur.take("I"), *cur.take("2d"), cur.take("Q")
            fns.append({"altitude": alt, "Spline": {"boundary": boundary, "bc": [bc0, bc1], "points": [list(cur.take("2d")) for _ in range(n)]}})
    env["atmosphere"] = {"pressure": {"altitude": pa, "pressure": pr}, "functions": fns,
                         "temperature_fixed_point": {"altitude": ta, "temperature": tt} if has else None}
    env["wavelength"] = cur.take("d")
    return env


def encode_params(cfg, object_elevations, coloring, vector3_len_prefix=True, env_encoder=encode_env):
    """bincode of `Params` (params.rs:496-505) for a Config.  object_elevations[i] = Altitude::abs of object i (what
    into_serializable_object stores in Coords.elev, object/mod.rs:164-183); coloring = atmrt_coloring_t from into_coloring."""
    p = cfg.params
    out = b""
    # scene: Scene { terrain_folder: String, objects: Vec<SerializableObject>, #[serde(skip)] callable_objects, terrain_alpha } :107-114
    out += _string(cfg.terrain_folder) + _u64(len(cfg.objects))
    for o, elev, tex in zip(cfg.objects, object_elevations, cfg.texture_paths or [None] * len(cfg.objects)):
        out += _f64(o.position.latitude, o.position.longitude, elev)  # position: Coords { lat, lon, elev }, utils/mod.rs:8-13
        if o.kind == _abi.OBJ_FRUSTUM:  # shape: Shape, object/mod.rs:119-131
            out += _u32(0) + _f64(o.r1, o.r2, o.height)
        else:  # Billboard { width, height, texture: Image { #[serde(skip)] image, path } } :77-83
            out += _u32(1) + _f64(o.width, o.height) + _string(tex or "")
        out += _f64(*o.color)  # color: Color { r, g, b, a } :133-140
    out += _f64(p.terrain_alpha)
    # view: View { position, frame, coloring, fog_distance } :287-293
    out += _f64(p.position.latitude, p.position.longitude)  # Position :32-40
    out += _u32(0 if p.position.altitude_kind == _abi.ALT_ABSOLUTE else 1) + _f64(p.position.altitude)  # Altitude :16-21
    out += _f64(p.frame.direction, p.frame.tilt, p.frame.fov, p.frame.max_distance)  # Frame :144-154
    if coloring.kind == _abi.COLORING_SIMPLE:  # Coloring :215-228
        out += _u32(0) + _f64(coloring.water_level, coloring.max_distance)
    else:
        out += _u32(1) + _f64(coloring.water_level, coloring.ambient_light) + _vector3(list(coloring.light_dir), vector3_len_prefix)
        out += _u32(coloring.palette)  # ColorPalette { Legacy, Improved }, coloring/shading.rs:9-14
    out += _option(coloring.fog_distance if coloring.has_fog else None, _f64)
    out += _earth_model(p.earth)  # model: EarthModel
    out += env_encoder(cfg)       # env: Environment — UNPINNED, see module docstring
    out += _u8(1 if p.straight_rays else 0) + _f64(p.simulation_step)
    # output: Output { file, file_metadata, width, height, ticks, vertical_ticks, show_eye_level, show_flat_horizon, generator } :394-414
    o = cfg.output
    out += _string(o["file"]) + _option(o["file_metadata"], _string) + _u16(p.width) + _u16(p.height)
    out += _ticks(o["ticks"]) + _ticks(o["vertical_ticks"]) + _u8(o["show_eye_level"]) + _u8(o["show_flat_horizon"])
    out += _u32(p.generator)  # GeneratorDef { Fast, InterpolatingRectilinear, Rectilinear } :386-391
    return out


def decode_params(data, pos=0, vector3_len_prefix=True, env_decoder=decode_env):
    """Inverse of encode_params: (dict mirroring `Params`, position after it)."""
    c = _Cursor(data, pos)
    scene = {"terrain_folder": c.string(), "objects": []}
    for _ in range(c.take("Q")):
        lat, lon, elev = c.take("3d")
        if c.take("I") == 0:
            r1, r2, h = c.take("3d")
            shape = {"Frustum": {"r1": r1, "r2": r2, "height": h}}
        else:
            w, h = c.take("2d")
            shape = {"Billboard": {"width": w, "height": h, "texture": {"path": c.string()}}}
        r, g, b, a = c.take("4d")
        scene["objects"].append({"position": {"lat": lat, "lon": lon, "elev": elev}, "shape": shape, "color": {"r": r, "g": g, "b": b, "a": a}})
    scene["terrain_alpha"] = c.take("d")
    lat, lon = c.take("2d")
    alt = {("Absolute", "Relative")[c.take("I")]: c.take("d")}
    d, t, f, m = c.take("4d")
    view = {"position": {"latitude": lat, "longitude": lon, "altitude": alt}, "frame": {"direction": d, "tilt": t, "fov": f, "max_distance": m}}
    if c.take("I") == 0:
        wl, md = c.take("2d")
        view["coloring"] = {"Simple": {"water_level": wl, "max_distance": md}}
    else:
        wl, al = c.take("2d")
        ld = c.vector3(vector3_len_prefix)
        view["coloring"] = {"Shading": {"water_level": wl, "ambient_light": al, "light_dir": ld, "palette": ("Legacy", "Improved")[c.take("I")]}}
    view["fog_distance"] = c.option(lambda: c.take("d"))
    k = c.take("I")
    name = EARTH_NAMES[k]
    if name == "Spherical":
        model = {name: {"radius": c.take("d")}}
    elif name == "ObserverAe":
        model = {name: {"proj_radius": c.take("d")}}
    elif name == "Ellipsoid":
        a, b = c.take("2d")
        model = {name: {"a": a, "b": b}}
    else:
        model = name
    env = env_decoder(c)
    straight, step = bool(c.take("B")), c.take("d")

    def ticks(angle_key):
        out = []
        for _ in range(c.take("Q")):
            if c.take("I") == 0:
                out.append({"Single": {angle_key: c.take("d"), "size": c.take("I"), "labelled": bool(c.take("B"))}})
            else:
                bias, stp = c.take("2d")
                out.append({"Multiple": {"bias": bias, "step": stp, "size": c.take("I"), "labelled": bool(c.take("B"))}})
        return out

    output = {"file": c.string(), "file_metadata": c.option(c.string), "width": c.take("H"), "height": c.take("H")}
    output["ticks"] = ticks("azimuth")
    output["vertical_ticks"] = ticks("elevation")
    output["show_eye_level"], output["show_flat_horizon"] = bool(c.take("B")), bool(c.take("B"))
    output["generator"] = ("Fast", "InterpolatingRectilinear", "Rectilinear")[c.take("I")]
    return {"scene": scene, "view": view, "model": model, "env": env, "straight_rays": straight, "simulation_step": step,
            "output": output}, c.p


# ---- result: Vec<Vec<ResultPixel>> through the library ---------------------------------------------
def encode_result(res, vector3_len_prefix=True):
    """bincode of `result` for a ResultPixels dict (see _abi.result_to_numpy) -> numpy uint8 array."""
    lib = _lib.load()
    r, keep = _abi.numpy_to_result(res)
    n = C.c_size_t()
    rc = lib.atmrt_result_encode_bincode(C.byref(r), int(vector3_len_prefix), None, 0, C.byref(n))
    if rc:
        raise ValueError(f"atmrt_result_encode_bincode: status {rc}")
    buf = np.empty(n.value, dtype=np.uint8)
    rc = lib.atmrt_result_encode_bincode(C.byref(r), int(vector3_len_prefix), buf.ctypes.data, buf.size, C.byref(n))
    del keep
    if rc:
        raise ValueError(f"atmrt_result_encode_bincode: status {rc}")
    return buf


def decode_result(data, pos=0, vector3_len_prefix=True):
    """Inverse of encode_result: (ResultPixels-style dict, position after it)."""
    lib = _lib.load()
    buf = np.frombuffer(data, dtype=np.uint8)[pos:]
    res, used = _abi.Result(), C.c_size_t()
    rc = lib.atmrt_result_decode_bincode(buf.ctypes.data, buf.size, int(vector3_len_prefix), C.byref(res), C.byref(used))
    if rc:
        raise ValueError(f"not a bincode Vec<Vec<ResultPixel>> (status {rc})")
    try:
        return _abi.result_to_numpy(res), pos + used.value
    finally:
        lib.atmrt_result_free(C.byref(res))


# ---- the file ---------------------------------------------------------------------------------------
def object_elevations(cfg, terrain):
    """Altitude::abs of every scene object (object/mod.rs:166-175) through Terrain::get_elev on the device."""
    elevs = []
    rel = [i for i, o in enumerate(cfg.objects) if o.position.altitude_kind == _abi.ALT_RELATIVE]
    ground = {}
    if rel:
        e, valid = terrain.get_elev([cfg.objects[i].position.latitude for i in rel], [cfg.objects[i].position.longitude for i in rel])
        ground = {i: (float(e[j]) if valid[j] else 0.0) for j, i in enumerate(rel)}  # .unwrap_or(0.0), params.rs:27
    for i, o in enumerate(cfg.objects):
        elevs.append(o.position.altitude if i not in ground else ground[i] + o.position.altitude)
    return elevs


def write_metadata(path, cfg, res, coloring, object_elevs=(), vector3_len_prefix=True, env_encoder=encode_env, level=6):
    """generator::output_metadata (src/generator/mod.rs:26-45): gzip(bincode(AllData { params, result })).

    NOT interchangeable with the reference as long as `env_encoder` is this module's stand-in: `Params.env` is
    `atm_refraction::Environment`, whose serde layout lives in a crate that is absent here; bincode is not self-describing, so the
    reference's `view` stops at that segment and cannot reach anything behind it, the whole `result` included.  Only
    read_metadata() of this package reads such a file; a warning says so every time one is written."""
    if env_encoder is encode_env:
        import warnings
        warnings.warn(f"{path}: the `env` segment is this package's stand-in (crate atm-refraction absent): the file is readable by "
                      "atm_raytracer_amd.metadata.read_metadata only, NOT by the reference's `atm-raytracer view`; pass env_encoder= "
                      "with the crate's bincode layout to write an interchangeable file", stacklevel=2)
    z = zlib.compressobj(level, zlib.DEFLATED, 31)  # wbits 31: gzip container, what libflate::gzip::Encoder writes
    with open(path, "wb") as f:
        f.write(z.compress(encode_params(cfg, list(object_elevs), coloring, vector3_len_prefix, env_encoder)))
        f.write(z.compress(memoryview(encode_result(res, vector3_len_prefix))))
        f.write(z.flush())


def read_metadata(path, vector3_len_prefix=True, env_decoder=decode_env):
    """viewer::run's load (src/viewer/mod.rs:17-29): gunzip + bincode::deserialize::<AllData> -> {"params", "result"}."""
    with open(path, "rb") as f:
        data = zlib.decompress(f.read(), 31)
    params, pos = decode_params(data, 0, vector3_len_prefix, env_decoder)
    result, pos = decode_result(data, pos, vector3_len_prefix)
    if pos != len(data):
        raise ValueError(f"{len(data) - pos} trailing bytes after AllData")
    return {"params": params, "result": result}
