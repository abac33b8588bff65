"""ctypes mirror of include/atmrt.h (the C ABI of the HIP library).

Field order and types must match the header exactly; tests/test_abi.py checks sizeof() of every
struct against the compiled library (atmrt_abi_sizeof).
"""
import ctypes as C

TEMP_LINEAR, TEMP_SPLINE = 0, 1
SPLINE_BOUNDARY = {"Natural": 0, "Derivatives": 1, "SecondDerivatives": 2}

# atmrt_status
OK, ERR_INVALID_ARGUMENT, ERR_NO_DEVICE, ERR_HIP, ERR_IO, ERR_FORMAT, ERR_STATE, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6, -7

# atmrt_earth_kind  (EarthModel, src/utils/earth_model/mod.rs:19-28)
EARTH_KINDS = {
    "SimpleSphere": 0, "Spherical": 1, "Ellipsoid": 2, "Wgs84": 3,
    "AzimuthalEquidistant": 4, "FlatDistorted": 5, "ObserverAe": 6, "SimpleObserverAe": 7,
}
ALT_ABSOLUTE, ALT_RELATIVE = 0, 1
# atmrt_generator_kind (GeneratorDef, params.rs:387-392)
GENERATORS = {"Fast": 0, "InterpolatingRectilinear": 1, "Rectilinear": 2}
OBJ_FRUSTUM, OBJ_BILLBOARD = 0, 1
COLOR_TERRAIN, COLOR_RGBA = 0, 1


class EarthModel(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("radius", C.c_double), ("a", C.c_double), ("b", C.c_double)]


class Position(C.Structure):
    _fields_ = [("latitude", C.c_double), ("longitude", C.c_double), ("altitude_kind", C.c_int32), ("_pad", C.c_int32),
                ("altitude", C.c_double)]


class Frame(C.Structure):
    _fields_ = [("direction", C.c_double), ("tilt", C.c_double), ("fov", C.c_double), ("max_distance", C.c_double)]


class Params(C.Structure):
    _fields_ = [("position", Position), ("frame", Frame), ("earth", EarthModel), ("wavelength", C.c_double),
                ("simulation_step", C.c_double), ("terrain_alpha", C.c_double), ("straight_rays", C.c_int32),
                ("generator", C.c_int32), ("width", C.c_uint16), ("height", C.c_uint16), ("col_begin", C.c_uint16),
                ("col_end", C.c_uint16)]


class TempFunction(C.Structure):
    """atmrt_temp_function_t.  The spline points are borrowed pointers: set them with set_points(), which keeps the arrays alive
    on the owning Atmosphere."""
    _fields_ = [("kind", C.c_int32), ("boundary", C.c_int32), ("altitude", C.c_double), ("gradient", C.c_double),
                ("bc", C.c_double * 2), ("n_points", C.c_int32), ("_pad", C.c_int32),
                ("point_altitude", C.POINTER(C.c_double)), ("point_temperature", C.POINTER(C.c_double))]


class Atmosphere(C.Structure):
    """atmrt_atmosphere_t: any number of temperature functions, any number of spline points (pointer + count, like the `Vec`s
    of AtmosphereDef).  Build with Atmosphere.new(n): the function table and the point arrays are owned by the Python object."""
    _fields_ = [("pressure_altitude", C.c_double), ("pressure", C.c_double), ("temperature_altitude", C.c_double),
                ("temperature", C.c_double), ("has_temperature_fixed_point", C.c_int32), ("n_functions", C.c_int32),
                ("functions", C.POINTER(TempFunction))]

    @classmethod
    def new(cls, n_functions):
        a = cls()
        a._table = (TempFunction * max(1, n_functions))()
        a._points = []
        a.functions = C.cast(a._table, C.POINTER(TempFunction))
        a.n_functions = n_functions
        return a

    def set_points(self, k, altitudes, temperatures):
        n = len(altitudes)
        xa, ya = (C.c_double * n)(*[float(v) for v in altitudes]), (C.c_double * n)(*[float(v) for v in temperatures])
        self._points.append((xa, ya))
        fn = self.functions[k]
        fn.n_points = n
        fn.point_altitude, fn.point_temperature = C.cast(xa, C.POINTER(C.c_double)), C.cast(ya, C.POINTER(C.c_double))


class Object(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("position", Position), ("r1", C.c_double), ("r2", C.c_double),
                ("height", C.c_double), ("width", C.c_double), ("color", C.c_double * 4),
                ("texture_rgba", C.POINTER(C.c_uint8)), ("texture_width", C.c_uint32), ("texture_height", C.c_uint32)]


class Result(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("n_pixels", C.c_uint64), ("n_hits", C.c_uint64),
                ("azimuth", C.POINTER(C.c_double)), ("elevation_angle", C.POINTER(C.c_double)),
                ("hit_count", C.POINTER(C.c_uint32)), ("hit_offset", C.POINTER(C.c_uint64)),
                ("lat", C.POINTER(C.c_double)), ("lon", C.POINTER(C.c_double)), ("distance", C.POINTER(C.c_double)),
                ("elevation", C.POINTER(C.c_double)), ("path_length", C.POINTER(C.c_double)),
                ("normal", C.POINTER(C.c_double)), ("color_tag", C.POINTER(C.c_uint32)), ("rgba", C.POINTER(C.c_double)),
                ("ray_steps", C.c_uint64), ("device_ms", C.c_double)]


class DevicePlanes(C.Structure):
    _fields_ = [("azimuth", C.c_void_p), ("elevation_angle", C.c_void_p), ("hit_count", C.c_void_p), ("lat", C.c_void_p),
                ("lon", C.c_void_p), ("distance", C.c_void_p), ("elevation", C.c_void_p), ("path_length", C.c_void_p),
                ("normal", C.c_void_p)]


class DeviceHits(C.Structure):
    _fields_ = [("capacity", C.c_uint64), ("hit_offset", C.c_void_p), ("lat", C.c_void_p), ("lon", C.c_void_p), ("distance", C.c_void_p),
                ("elevation", C.c_void_p), ("path_length", C.c_void_p), ("normal", C.c_void_p), ("color_tag", C.c_void_p),
                ("rgba", C.c_void_p)]


class Coloring(C.Structure):
    _fields_ = [("kind", C.c_int32), ("palette", C.c_int32), ("water_level", C.c_double), ("max_distance", C.c_double),
                ("ambient_light", C.c_double), ("light_dir", C.c_double * 3), ("has_fog", C.c_int32), ("_pad", C.c_int32),
                ("fog_distance", C.c_double)]


COLORING_SIMPLE, COLORING_SHADING = 0, 1
PALETTES = {"Legacy": 0, "Improved": 1}


def numpy_to_result(res):
    """Inverse of result_to_numpy: an atmrt_result_t whose pointers borrow the numpy arrays (keep `res` alive)."""
    import numpy as np
    r = Result()
    keep = {}

    def ptr(key, dtype, ctype):
        a = np.ascontiguousarray(res[key], dtype=dtype)
        if a.size == 0:
            a = np.zeros(1, dtype=dtype)
        keep[key] = a
        return a.ctypes.data_as(C.POINTER(ctype))

    r.width, r.height = int(res["width"]), int(res["height"])
    r.n_pixels, r.n_hits = r.width * r.height, int(res["n_hits"])
    r.azimuth, r.elevation_angle = ptr("azimuth", np.float64, C.c_double), ptr("elevation_angle", np.float64, C.c_double)
    r.hit_count, r.hit_offset = ptr("hit_count", np.uint32, C.c_uint32), ptr("hit_offset", np.uint64, C.c_uint64)
    for k in ("lat", "lon", "distance", "elevation", "path_length", "normal", "rgba"):
        setattr(r, k, ptr(k, np.float64, C.c_double))
    r.color_tag = ptr("color_tag", np.uint32, C.c_uint32)
    r.ray_steps = int(res["ray_steps"])
    return r, keep


class CommTimings(C.Structure):
    _fields_ = [("gather_ms", C.c_double), ("assemble_ms", C.c_double), ("tile_ms_max", C.c_double), ("tile_ms_min", C.c_double),
                ("bytes_per_rank", C.c_uint64), ("world", C.c_int32), ("route", C.c_int32), ("collectives", C.c_int32), ("_pad", C.c_int32)]


ROUTES = {0: "none", 1: "rccl", 2: "peer", 3: "external", 4: "host", 5: "external-device"}
COMM_ID_BYTES = 128
# atmrt_all_gather_fn(user, send_host, recv_host, bytes_per_rank) -> int
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)


class Timings(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("profile_ms", C.c_double), ("paths_ms", C.c_double), ("intersect_ms", C.c_double),
                ("march_ms", C.c_double), ("finalize_ms", C.c_double), ("pack_ms", C.c_double), ("ray_steps", C.c_uint64),
                ("n_hits", C.c_uint64)]


class FrameStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("unlisted_rays", "unlisted_columns", "retraced_pixels", "big_steps", "big_blend_pixels", "terrain_lookups", "object_rays", "object_steps")]


def result_to_numpy(res):
    """Copy an atmrt_result_t (library-owned) into a dict of numpy arrays."""
    import numpy as np

    n_px, n_hits = int(res.n_pixels), int(res.n_hits)

    def arr(ptr, n, dtype):
        if n == 0:
            return np.zeros(0, dtype=dtype)
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)

    h, w = int(res.height), int(res.width)
    out = {
        "width": w, "height": h, "n_hits": n_hits, "ray_steps": int(res.ray_steps), "device_ms": float(res.device_ms),
        "azimuth": arr(res.azimuth, n_px, np.float64).reshape(h, w),
        "elevation_angle": arr(res.elevation_angle, n_px, np.float64).reshape(h, w),
        "hit_count": arr(res.hit_count, n_px, np.uint32).reshape(h, w),
        "hit_offset": arr(res.hit_offset, n_px, np.uint64).reshape(h, w),
        "lat": arr(res.lat, n_hits, np.float64), "lon": arr(res.lon, n_hits, np.float64),
        "distance": arr(res.distance, n_hits, np.float64), "elevation": arr(res.elevation, n_hits, np.float64),
        "path_length": arr(res.path_length, n_hits, np.float64),
        "normal": arr(res.normal, 3 * n_hits, np.float64).reshape(n_hits, 3),
        "color_tag": arr(res.color_tag, n_hits, np.uint32),
        "rgba": arr(res.rgba, 4 * n_hits, np.float64).reshape(n_hits, 4),
    }
    return out
