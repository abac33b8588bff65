"""Loader of the HIP C-ABI library (csrc/libatmrt.so).  Fails loudly: there is no CPU fallback."""
import ctypes as C
import os
import subprocess

from . import _abi

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# ATMRT_LIB: an experimental build of the library (tools/dev_build.sh) instead of the shipped one — development only
LIB_PATH = os.environ.get("ATMRT_LIB") or os.path.join(CSRC, "libatmrt.so")

_lib = None


class AtmrtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"atmrt status {status}: {message}")
        self.status = status
        self.message = message


def source_hash():
    """sha256 over the sources libatmrt.so is built from (csrc/*.h, csrc/*.hip, csrc/Makefile, include/atmrt.h): profile summaries under
    profiles/ carry the hash of the sources their counters were collected with, and bench.py only quotes counters whose hash
    matches the tree it runs from."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip")))
    files.append(os.path.join(CSRC, "Makefile"))  # the per-unit code generation flags live there
    files.append(os.path.join(os.path.dirname(CSRC), "..", "include", "atmrt.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build_info():
    """{'source_hash': ..., 'march_units': flags, 'calling_units': flags, 'all': flags, 'arch': ...} of the LOADED library."""
    text = load().atmrt_build_info().decode()
    return dict(part.split(": ", 1) for part in text.split("; ") if ": " in part)


def build(force=False):
    """Compile libatmrt.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-s", "-C", CSRC, "clean"], check=True)
    subprocess.run(["make", "-s", "-j8", "-C", CSRC], check=True)
    return LIB_PATH


def load():
    """dlopen libatmrt.so and declare every entry point of include/atmrt.h."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64 and cannot
    # initialise after the system copy has claimed the device ("No HIP GPUs are available").  Importing torch first
    # makes libatmrt.so bind to the runtime torch uses, so both share devices, streams and allocations.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C {CSRC}` (hipcc, gfx950); "
                          "this package has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, i32, dbl, sz = C.c_void_p, C.c_int32, C.c_double, C.c_size_t
    pd = C.POINTER(C.c_double)
    sig = {
        "atmrt_abi_version": (C.c_int, []),
        "atmrt_abi_sizeof": (sz, [C.c_int]),
        "atmrt_build_info": (C.c_char_p, []),
        "atmrt_ctx_create": (C.c_int, [C.POINTER(vp), C.c_int]),
        "atmrt_ctx_destroy": (None, [vp]),
        "atmrt_last_error": (C.c_char_p, [vp]),
        "atmrt_terrain_load_dir": (C.c_int, [vp, C.c_char_p, C.POINTER(i32)]),
        "atmrt_terrain_add_tile": (C.c_int, [vp, i32, i32, i32, i32, vp]),
        "atmrt_terrain_clear": (C.c_int, [vp]),
        "atmrt_terrain_get_elev": (C.c_int, [vp, sz, vp, vp, vp, vp]),
        "atmrt_params_default": (None, [C.POINTER(_abi.Params)]),
        "atmrt_atmosphere_us76": (None, [C.POINTER(_abi.Atmosphere)]),
        "atmrt_set_params": (C.c_int, [vp, C.POINTER(_abi.Params)]),
        "atmrt_set_atmosphere": (C.c_int, [vp, C.POINTER(_abi.Atmosphere)]),
        "atmrt_objects_set": (C.c_int, [vp, C.POINTER(_abi.Object), sz]),
        "atmrt_generate": (C.c_int, [vp, C.POINTER(_abi.Result)]),
        "atmrt_result_free": (None, [C.POINTER(_abi.Result)]),
        "atmrt_generate_device": (C.c_int, [vp, C.POINTER(_abi.DevicePlanes), C.POINTER(C.c_uint64), pd]),
        "atmrt_last_hits_device": (C.c_int, [vp, C.POINTER(_abi.DeviceHits), C.POINTER(C.c_uint64)]),
        "atmrt_last_timings": (C.c_int, [vp, C.POINTER(_abi.Timings)]),
        "atmrt_last_stats": (C.c_int, [vp, C.POINTER(_abi.FrameStats)]),
        "atmrt_debug_fail_next_frame": (C.c_int, [vp]),
        "atmrt_debug_march_plan": (C.c_int, [i32, i32, i32, i32, C.POINTER(C.c_uint64)]),
        "atmrt_coloring_from_conf": (C.c_int, [C.POINTER(_abi.Params), i32, dbl, dbl, dbl, dbl, i32, i32, dbl, C.POINTER(_abi.Coloring)]),
        "atmrt_draw_image": (C.c_int, [vp, C.POINTER(_abi.Coloring), vp]),
        "atmrt_draw_image_device": (C.c_int, [vp, C.POINTER(_abi.Coloring), vp]),
        "atmrt_ray_paths": (C.c_int, [vp, dbl, sz, vp, i32, dbl, sz, vp, vp]),
        "atmrt_atmosphere_sample": (C.c_int, [vp, sz, vp, vp, vp, vp, vp]),
        "atmrt_coords_at_dist": (C.c_int, [vp, dbl, dbl, dbl, sz, vp, vp, vp]),
        "atmrt_math_probe": (C.c_int, [vp, i32, sz, vp, vp, vp, vp]),
        "atmrt_result_encode_bincode": (C.c_int, [C.POINTER(_abi.Result), i32, vp, sz, C.POINTER(sz)]),
        "atmrt_result_decode_bincode": (C.c_int, [vp, sz, i32, C.POINTER(_abi.Result), C.POINTER(sz)]),
        # several GPUs (include/atmrt.h)
        "atmrt_comm_unique_id": (C.c_int, [vp]),
        "atmrt_ctx_comm_init_rank": (C.c_int, [vp, vp, i32, i32]),
        "atmrt_ctx_comm_init_external": (C.c_int, [vp, i32, i32, _abi.ALL_GATHER_FN, vp]),
        "atmrt_ctx_comm_init_external_device": (C.c_int, [vp, i32, i32, _abi.ALL_GATHER_FN, vp]),
        "atmrt_ctx_create_multi": (C.c_int, [C.POINTER(vp), C.POINTER(i32), i32]),
        "atmrt_ctx_device_count": (C.c_int, [vp]),
        "atmrt_generate_image_device": (C.c_int, [vp, C.POINTER(_abi.DevicePlanes), C.POINTER(C.c_uint64), pd]),
        "atmrt_image_hits_device": (C.c_int, [vp, C.POINTER(_abi.DeviceHits), C.POINTER(C.c_uint64)]),
        "atmrt_draw_image_gathered_device": (C.c_int, [vp, C.POINTER(_abi.Coloring), C.POINTER(vp)]),
        "atmrt_last_comm_timings": (C.c_int, [vp, C.POINTER(_abi.CommTimings)]),
        "atmrt_comm_available": (C.c_int, []),
        "atmrt_ctx_tile_columns": (C.c_int, [vp, i32, C.POINTER(i32), C.POINTER(i32)]),
        "atmrt_tiles_rebalance": (C.c_int, [i32, i32, C.POINTER(i32), pd, C.POINTER(i32)]),
        "atmrt_debug_set_tiling": (C.c_int, [vp, C.POINTER(i32), i32]),
        "atmrt_debug_fail_next_collective": (C.c_int, [vp, i32, i32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if L.atmrt_abi_version() != 5:
        raise ImportError("libatmrt.so ABI version mismatch")
    _lib = L
    return L


EXPORTED = ["atmrt_abi_version", "atmrt_build_info", "atmrt_ctx_create", "atmrt_ctx_destroy", "atmrt_last_error", "atmrt_terrain_load_dir",
            "atmrt_terrain_add_tile", "atmrt_terrain_clear", "atmrt_terrain_get_elev", "atmrt_params_default",
            "atmrt_atmosphere_us76", "atmrt_set_params", "atmrt_set_atmosphere", "atmrt_objects_set", "atmrt_generate",
            "atmrt_result_free", "atmrt_generate_device", "atmrt_last_hits_device", "atmrt_last_timings", "atmrt_last_stats", "atmrt_debug_fail_next_frame", "atmrt_debug_march_plan", "atmrt_coloring_from_conf", "atmrt_draw_image",
            "atmrt_draw_image_device", "atmrt_ray_paths", "atmrt_atmosphere_sample",
            "atmrt_coords_at_dist", "atmrt_math_probe", "atmrt_result_encode_bincode",
            "atmrt_result_decode_bincode", "atmrt_comm_unique_id", "atmrt_ctx_comm_init_rank", "atmrt_ctx_comm_init_external", "atmrt_ctx_comm_init_external_device",
            "atmrt_ctx_create_multi", "atmrt_ctx_device_count", "atmrt_generate_image_device", "atmrt_image_hits_device",
            "atmrt_draw_image_gathered_device", "atmrt_last_comm_timings", "atmrt_comm_available", "atmrt_ctx_tile_columns",
            "atmrt_tiles_rebalance", "atmrt_debug_set_tiling", "atmrt_debug_fail_next_collective"]
