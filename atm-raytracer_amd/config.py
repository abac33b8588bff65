"""YAML configuration with the reference's schema and defaults (src/generator/params.rs).

`parse_config(path)` / `Config.from_dict` mirror `params::parse_config` (params.rs:678-692): every
field has the reference's serde default; enums use serde's externally tagged form
(`earth_shape: {Spherical: {radius: 6371000}}`, `altitude: {Relative: 2}`, `generator: Fast`).
Only the part of the schema that reaches the generators is interpreted; renderer-only keys
(coloring, ticks, fog, file names) are accepted and ignored, unknown top-level keys raise like
serde's `deny`-less parse would not — they are ignored too.
"""
import ctypes as C

import yaml

from . import _abi


class ConfigError(ValueError):
    pass


def _altitude(node, default=(_abi.ALT_RELATIVE, 1.0)):  # Altitude, params.rs:17-21; default Relative(1.0) :42-44
    if node is None:
        return default
    if isinstance(node, dict) and len(node) == 1:
        (k, v), = node.items()
        if k == "Absolute":
            return _abi.ALT_ABSOLUTE, float(v)
        if k == "Relative":
            return _abi.ALT_RELATIVE, float(v)
    raise ConfigError(f"altitude must be {{Absolute: x}} or {{Relative: x}}, got {node!r}")


def _position(node):  # Position, params.rs:32-40
    node = node or {}
    p = _abi.Position()
    p.latitude = float(node.get("latitude", 0.0))
    p.longitude = float(node.get("longitude", 0.0))
    p.altitude_kind, p.altitude = _altitude(node.get("altitude"))
    return p


def _earth(node):  # EarthModel, earth_model/mod.rs:18-28; default Spherical{6371000}, params.rs:467-471
    e = _abi.EarthModel()
    if node is None:
        e.kind, e.radius = _abi.EARTH_KINDS["Spherical"], 6_371_000.0
        return e
    if isinstance(node, str):
        if node not in ("SimpleSphere", "Wgs84", "AzimuthalEquidistant", "FlatDistorted", "SimpleObserverAe"):
            raise ConfigError(f"unknown earth_shape {node!r}")
        e.kind = _abi.EARTH_KINDS[node]
        return e
    if isinstance(node, dict) and len(node) == 1:
        (k, v), = node.items()
        if k == "Spherical":
            e.kind, e.radius = _abi.EARTH_KINDS[k], float(v["radius"])
            return e
        if k == "Ellipsoid":
            e.kind, e.a, e.b = _abi.EARTH_KINDS[k], float(v["a"]), float(v["b"])
            return e
        if k == "ObserverAe":  # field is proj_radius in the code (earth_model/mod.rs:26)
            e.kind, e.radius = _abi.EARTH_KINDS[k], float(v["proj_radius"])
            return e
    raise ConfigError(f"unknown earth_shape {node!r}")


def us76():
    """AtmosphereDef::us_76 (params.rs:453)."""
    alts = [0.0, 11000.0, 20000.0, 32000.0, 47000.0, 51000.0, 71000.0]
    grads = [-0.0065, 0.0, 0.001, 0.0028, 0.0, -0.0028, -0.002]
    a = _abi.Atmosphere.new(len(alts))
    a.pressure_altitude, a.pressure = 0.0, 101325.0
    a.temperature_altitude, a.temperature, a.has_temperature_fixed_point = 0.0, 288.15, 1
    for k, (h, g) in enumerate(zip(alts, grads)):
        a.functions[k].kind, a.functions[k].altitude, a.functions[k].gradient = _abi.TEMP_LINEAR, h, g
    return a


def _temperature_function(node, a, index):
    """`Linear{gradient}` or `Spline{boundary_condition, points}` (reference README.md:290-316) into function `index` of `a`."""
    fn = a.functions[index]
    if isinstance(node, dict) and len(node) == 1:
        (k, v), = node.items()
        if k == "Linear":
            fn.kind, fn.gradient = _abi.TEMP_LINEAR, float(v["gradient"])
            return
        if k == "Spline":
            fn.kind = _abi.TEMP_SPLINE
            bc = v.get("boundary_condition", "Natural")
            if isinstance(bc, str):
                if bc != "Natural":
                    raise ConfigError(f"unknown boundary_condition {bc!r}")
                fn.boundary = _abi.SPLINE_BOUNDARY["Natural"]
            else:
                (bk, bv), = bc.items()
                if bk not in ("Derivatives", "SecondDerivatives") or len(bv) != 2:
                    raise ConfigError(f"unknown boundary_condition {bc!r}")
                fn.boundary, fn.bc[0], fn.bc[1] = _abi.SPLINE_BOUNDARY[bk], float(bv[0]), float(bv[1])
            pts = v["points"]
            if len(pts) < 2:
                raise ConfigError("a Spline needs at least 2 points")
            a.set_points(index, [p[0] for p in pts], [p[1] for p in pts])
            return
    raise ConfigError(f"unknown temperature function {node!r}")


def _atmosphere(node):  # AtmosphereDef schema, reference README.md:283-323
    if node is None:
        return us76()
    pr = node["pressure"]
    functions = [(0.0, node["first_temperature_function"])] + [(float(nf["altitude"]), nf["function"]) for nf in node.get("next_functions", []) or []]
    a = _abi.Atmosphere.new(len(functions))
    a.pressure_altitude, a.pressure = float(pr["altitude"]), float(pr["pressure"])
    for k, (h, f) in enumerate(functions):
        a.functions[k].altitude = h
        _temperature_function(f, a, k)
    fixed = node.get("temperature_fixed_point")
    has_spline = any(a.functions[k].kind == _abi.TEMP_SPLINE for k in range(a.n_functions))
    if fixed is not None:
        a.temperature_altitude, a.temperature, a.has_temperature_fixed_point = float(fixed["altitude"]), float(fixed["temperature"]), 1
    elif not has_spline:
        raise ConfigError("temperature_fixed_point is required when every temperature function is Linear")
    return a


def _object(node, load_texture):  # ConfObject, object/mod.rs:156-161
    o = _abi.Object()
    o.position = _position(node.get("position"))
    col = node.get("color") or {}
    o.color[0], o.color[1], o.color[2] = float(col.get("r", 0)), float(col.get("g", 0)), float(col.get("b", 0))
    o.color[3] = float(col.get("a", 1.0))  # default_alpha, object/mod.rs:144-146
    (k, v), = node["shape"].items()
    keep = None
    if k == "Cylinder":  # ConfShape::into_shape, object/mod.rs:42-75
        o.kind, o.r1, o.r2, o.height = _abi.OBJ_FRUSTUM, float(v["radius"]), float(v["radius"]), float(v["height"])
    elif k == "Cone":
        o.kind, o.r1, o.r2, o.height = _abi.OBJ_FRUSTUM, float(v["radius"]), 0.0, float(v["height"])
    elif k == "Frustum":
        o.kind, o.r1, o.r2, o.height = _abi.OBJ_FRUSTUM, float(v["r1"]), float(v["r2"]), float(v["height"])
    elif k == "Billboard":
        o.kind, o.width, o.height = _abi.OBJ_BILLBOARD, float(v["width"]), float(v["height"])
        tex = load_texture(v["texture_path"])  # uint8 [h][w][4]
        keep = tex
        o.texture_rgba = tex.ctypes.data_as(C.POINTER(C.c_uint8))
        o.texture_height, o.texture_width = tex.shape[0], tex.shape[1]
    else:
        raise ConfigError(f"unknown shape {k!r}")
    return o, keep


def _ticks(nodes, angle_key):
    """Vec<Tick> / Vec<VerticalTick> (params.rs:306-367): externally tagged `Single{azimuth|elevation, size, labelled}` or
    `Multiple{bias, step, size, labelled}`; every field is required (no serde default).  Kept for the metadata file only."""
    out = []
    for node in nodes or []:
        (k, v), = node.items()
        if k == "Single":
            out.append(("Single", float(v[angle_key]), int(v["size"]), bool(v["labelled"])))
        elif k == "Multiple":
            out.append(("Multiple", float(v["bias"]), float(v["step"]), int(v["size"]), bool(v["labelled"])))
        else:
            raise ConfigError(f"unknown tick {k!r}")
    return out


def _load_texture(path):
    import numpy as np
    from PIL import Image

    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGBA"), dtype=np.uint8))


def _coloring(view):
    """ConfColoring (params.rs:164-213) + fog_distance (:285): returned as the argument tuple of into_coloring."""
    node = view.get("coloring")
    fog = view.get("fog_distance")
    conf = {"kind": _abi.COLORING_SHADING, "water_level": 0.0, "ambient_light": 0.4, "light_zenith_angle": 45.0, "light_dir": 0.0,
            "palette": _abi.PALETTES["Improved"], "has_fog": 0 if fog is None else 1, "fog_distance": 0.0 if fog is None else float(fog)}
    if node is None:
        return conf  # ConfColoring::default = Shading{0, 0.4, 45, 0, Improved}, params.rs:203-213
    (k, v), = node.items()
    v = v or {}
    conf["water_level"] = float(v.get("water_level", 0.0))
    if k == "Simple":
        conf["kind"] = _abi.COLORING_SIMPLE
    elif k == "Shading":
        conf["ambient_light"] = float(v.get("ambient_light", 0.4))
        conf["light_zenith_angle"] = float(v.get("light_zenith_angle", 45.0))
        conf["light_dir"] = float(v.get("light_dir", 0.0))
        pal = v.get("palette", "Improved")
        if pal not in _abi.PALETTES:
            raise ConfigError(f"unknown palette {pal!r}")
        conf["palette"] = _abi.PALETTES[pal]
    else:
        raise ConfigError(f"unknown coloring {k!r}")
    return conf


class Config:
    """Config (params.rs:447-465) reduced to what reaches the generators (+ the colouring of the renderer)."""

    def __init__(self):
        self.coloring = _coloring({})
        self.params = _abi.Params()
        self.atmosphere = us76()
        self.objects = []
        self._keepalive = []
        self.terrain_folder = "./terrain"  # default_terrain_folder, params.rs:72-74
        self.texture_paths = []  # per object: ConfShape::Billboard.texture_path or None
        # Output (params.rs:394-445): renderer-only fields, carried into the metadata file
        self.output = {"file": "./output.png", "file_metadata": None, "ticks": [], "vertical_ticks": [], "show_eye_level": False,
                       "show_flat_horizon": False}
        p = self.params
        p.position = _position(None)
        p.frame.direction, p.frame.tilt, p.frame.fov, p.frame.max_distance = 0.0, 0.0, 30.0, 150_000.0
        p.earth = _earth(None)
        p.wavelength, p.simulation_step, p.terrain_alpha = 530e-9, 50.0, 1.0
        p.straight_rays, p.generator = 0, _abi.GENERATORS["Fast"]
        p.width, p.height = 640, 480

    @classmethod
    def from_dict(cls, d, load_texture=_load_texture):
        d = d or {}
        c = cls()
        p = c.params
        scene = d.get("scene") or {}
        c.terrain_folder = scene.get("terrain_folder", c.terrain_folder)
        p.terrain_alpha = float(scene.get("terrain_alpha", 1.0))
        for node in scene.get("objects") or []:
            o, keep = _object(node, load_texture)
            c.objects.append(o)
            c._keepalive.append(keep)
            (shape_kind, shape), = node["shape"].items()
            c.texture_paths.append(str(shape["texture_path"]) if shape_kind == "Billboard" else None)
        view = d.get("view") or {}
        c.coloring = _coloring(view)
        p.position = _position(view.get("position"))
        fr = view.get("frame") or {}
        p.frame.direction = float(fr.get("direction", 0.0))
        p.frame.tilt = float(fr.get("tilt", 0.0))
        p.frame.fov = float(fr.get("fov", 30.0))
        p.frame.max_distance = float(fr.get("max_distance", 150_000.0))
        p.earth = _earth(d.get("earth_shape"))
        c.atmosphere = _atmosphere(d.get("atmosphere"))
        p.wavelength = float(d.get("wavelength", 530e-9))
        p.straight_rays = 1 if d.get("straight_rays", False) else 0
        p.simulation_step = float(d.get("simulation_step", 50.0))
        out = d.get("output") or {}
        w, h = int(out.get("width", 640)), int(out.get("height", 480))
        if not (0 <= w <= 65535 and 0 <= h <= 65535):
            raise ConfigError("width/height must fit u16")  # params.rs:398-402
        p.width, p.height = w, h
        c.output = {"file": str(out.get("file", "./output.png")), "file_metadata": out.get("file_metadata"),
                    "ticks": _ticks(out.get("ticks"), "azimuth"), "vertical_ticks": _ticks(out.get("vertical_ticks"), "elevation"),
                    "show_eye_level": bool(out.get("show_eye_level", False)),
                    "show_flat_horizon": bool(out.get("show_flat_horizon", False))}
        gen = out.get("generator", "Fast")
        if gen not in _abi.GENERATORS:
            raise ConfigError(f"unknown generator {gen!r}")
        p.generator = _abi.GENERATORS[gen]
        return c


def parse_config(path):
    """params::parse_config (params.rs:678-692)."""
    with open(path) as f:
        return Config.from_dict(yaml.safe_load(f))
