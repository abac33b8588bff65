"""Host-side mirror of the reference's generator interface above the C ABI.

Reference                                   here
------------------------------------------  -----------------------------------------------
Terrain::from_folder / get_elev             Terrain.from_folder / Terrain.get_elev
  (src/terrain/mod.rs:66-83,120-126)
Config::into_params(&terrain) -> Params     Params(config)
trait Generator { fn generate(&self) }      Generator.generate() -> ResultPixels
  FastGenerator::new(&params,&terrain,..)     FastGenerator(params, terrain)
  RectilinearGenerator::new(..)               RectilinearGenerator(params, terrain)
  InterpolatingRectilinearGenerator::new(..)  InterpolatingRectilinearGenerator(params, terrain)
generator::generate's match on GeneratorDef  make_generator(params, terrain)
  (src/generator/mod.rs:72-78)
"""
import ctypes as C
import os

import numpy as np

from . import _abi, _lib
from ._lib import AtmrtError
from .config import Config


class Context:
    """Owns one atmrt_ctx: one HIP device, or — Context.multi([...]) — several devices of this process behind one handle
    (the library cuts every frame into pixel-column tiles, one per device; include/atmrt.h "several GPUs of one node")."""

    def __init__(self, device=None, devices=None):
        self.lib = _lib.load()
        h = C.c_void_p()
        if devices is not None:
            devices = [int(d) for d in devices]
            arr = (C.c_int32 * len(devices))(*devices)
            rc = self.lib.atmrt_ctx_create_multi(C.byref(h), arr, len(devices))
            device = devices[0] if devices else 0
        else:
            if device is None:
                device = int(os.environ.get("LOCAL_RANK", "0"))
            rc = self.lib.atmrt_ctx_create(C.byref(h), device)
        if rc != 0:
            raise AtmrtError(rc, self.lib.atmrt_last_error(None).decode())
        self.handle = h
        self.device = device
        self.devices = devices or [device]
        self._transport = None

    @classmethod
    def multi(cls, devices):
        return cls(devices=devices)

    # ---- one process per GPU: this context becomes rank `rank` of `world` ranks that share every frame -------------------
    def comm_unique_id(self):
        """ncclGetUniqueId through the library: 128 bytes for the other ranks' comm_init_rank."""
        buf = (C.c_uint8 * _abi.COMM_ID_BYTES)()
        rc = self.lib.atmrt_comm_unique_id(buf)
        if rc != 0:
            raise AtmrtError(rc, self.lib.atmrt_last_error(None).decode())
        return bytes(buf)

    def comm_init_rank(self, unique_id, rank, world):
        buf = (C.c_uint8 * _abi.COMM_ID_BYTES).from_buffer_copy(unique_id)
        self.check(self.lib.atmrt_ctx_comm_init_rank(self.handle, buf, rank, world))

    def comm_init_external(self, rank, world, all_gather):
        """all_gather(send: memoryview, recv: memoryview) moves host bytes between the ranks (MPI, gloo, a test double)."""
        def thunk(_user, send, recv, nbytes):
            try:
                all_gather((C.c_uint8 * nbytes).from_address(send), (C.c_uint8 * (nbytes * world)).from_address(recv))
                return 0
            except Exception as exc:  # an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                self._transport_error = exc
                return 1
        self._transport = _abi.ALL_GATHER_FN(thunk)  # keep the trampoline alive as long as the context
        self.check(self.lib.atmrt_ctx_comm_init_external(self.handle, rank, world, self._transport, None))

    def comm_init_external_device(self, rank, world, all_gather):
        """all_gather(send_ptr, recv_ptr, nbytes) moves DEVICE memory (e.g. through the torch.distributed communicator the host
        already owns: torch_device_all_gather below) and returns when the gathered bytes are in place."""
        def thunk(_user, send, recv, nbytes):
            try:
                all_gather(send, recv, nbytes)
                return 0
            except Exception as exc:
                import traceback
                traceback.print_exc()
                self._transport_error = exc
                return 1
        self._transport = _abi.ALL_GATHER_FN(thunk)
        self.check(self.lib.atmrt_ctx_comm_init_external_device(self.handle, rank, world, self._transport, None))

    def comm_timings(self):
        t = _abi.CommTimings()
        self.check(self.lib.atmrt_last_comm_timings(self.handle, C.byref(t)))
        out = {k: getattr(t, k) for k, _ in _abi.CommTimings._fields_ if k != "_pad"}
        out["route"] = _abi.ROUTES.get(out["route"], out["route"])
        return out

    def tile_columns(self, index=-1):
        """[c0, c1) of a tile (atmrt_ctx_tile_columns): of the last exchanged frame, else of the next one."""
        c0, c1 = C.c_int32(), C.c_int32()
        self.check(self.lib.atmrt_ctx_tile_columns(self.handle, index, C.byref(c0), C.byref(c1)))
        return c0.value, c1.value

    def set_tiling(self, cols):
        """Test hook (atmrt_debug_set_tiling): the next frames use exactly these world + 1 column boundaries; None: the library's own."""
        if cols is None:
            self.check(self.lib.atmrt_debug_set_tiling(self.handle, None, 0))
        else:
            arr = (C.c_int32 * len(cols))(*[int(v) for v in cols])
            self.check(self.lib.atmrt_debug_set_tiling(self.handle, arr, len(cols)))

    def fail_next_collective(self, index=0, nth=1):
        """Test hook (atmrt_debug_fail_next_collective)."""
        self.check(self.lib.atmrt_debug_fail_next_collective(self.handle, index, nth))

    def check(self, rc):
        if rc != 0:
            raise AtmrtError(rc, self.lib.atmrt_last_error(self.handle).decode())

    def close(self):
        if getattr(self, "handle", None):
            self.lib.atmrt_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _DeviceBytes:
    """A raw device pointer as a __cuda_array_interface__ object, so that torch can wrap it without a copy."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def torch_device_all_gather(dist, device):
    """An all_gather for Context.comm_init_external_device over torch.distributed's own communicator (RCCL on device tensors)."""
    import torch

    def all_gather(send, recv, nbytes):
        world = dist.get_world_size()
        s = torch.as_tensor(_DeviceBytes(send, nbytes), device=device)
        r = torch.as_tensor(_DeviceBytes(recv, nbytes * world), device=device)
        dist.all_gather_into_tensor(r, s)
        torch.cuda.synchronize(device)
    return all_gather


class Terrain:
    """Terrain (src/terrain/mod.rs:55-57): tiles keyed by integer (lat, lon), resident in HBM."""

    def __init__(self, ctx=None):
        self.ctx = ctx or Context()
        self.n_files = 0

    @classmethod
    def from_folder(cls, path, ctx=None):
        t = cls(ctx)
        n = C.c_int32()
        t.ctx.check(t.ctx.lib.atmrt_terrain_load_dir(t.ctx.handle, os.fsencode(path), C.byref(n)))
        t.n_files = n.value
        return t

    @classmethod
    def from_tiles(cls, tiles, ctx=None):
        """tiles: {(lat0, lon0): int16 array [n_lat][n_lon], south->north, west->east}."""
        t = cls(ctx)
        for (lat0, lon0), posts in tiles.items():
            t.add_tile(lat0, lon0, posts)
        return t

    def add_tile(self, lat0, lon0, posts):
        posts = np.ascontiguousarray(posts, dtype=np.int16)
        self.ctx.check(self.ctx.lib.atmrt_terrain_add_tile(self.ctx.handle, lat0, lon0, posts.shape[0], posts.shape[1],
                                                          posts.ctypes.data))
        self.n_files += 1

    def get_elev(self, lat, lon):
        """Batched Terrain::get_elev; returns (elev, valid) arrays; valid=False where the reference returns None."""
        lat = np.ascontiguousarray(np.atleast_1d(lat), dtype=np.float64)
        lon = np.ascontiguousarray(np.atleast_1d(lon), dtype=np.float64)
        elev = np.zeros_like(lat)
        valid = np.zeros(lat.shape, dtype=np.uint8)
        self.ctx.check(self.ctx.lib.atmrt_terrain_get_elev(self.ctx.handle, lat.size, lat.ctypes.data, lon.ctypes.data,
                                                          elev.ctypes.data, valid.ctypes.data))
        return elev, valid.astype(bool)


class Params:
    """`Params` (params.rs:496-505): Config resolved against a Terrain."""

    def __init__(self, config: Config):
        self.config = config
        self.pod = config.params
        self.atmosphere = config.atmosphere
        self.objects = config.objects


class ResultPixels(dict):
    """Vec<Vec<ResultPixel>> as structure-of-arrays (see _abi.result_to_numpy).  `pixel(y, x)` rebuilds one
    ResultPixel {elevation_angle, azimuth, trace_points[]} (generators/mod.rs:13-30)."""

    def pixel(self, y, x):
        off, cnt = int(self["hit_offset"][y, x]), int(self["hit_count"][y, x])
        tps = []
        for k in range(off, off + cnt):
            tps.append({"lat": self["lat"][k], "lon": self["lon"][k], "distance": self["distance"][k],
                        "elevation": self["elevation"][k], "path_length": self["path_length"][k],
                        "normal": self["normal"][k], "color_tag": int(self["color_tag"][k]), "rgba": self["rgba"][k]})
        return {"elevation_angle": self["elevation_angle"][y, x], "azimuth": self["azimuth"][y, x], "trace_points": tps}


class Generator:
    """trait Generator (generators/mod.rs:82-84)."""

    KIND = None

    def __init__(self, params: Params, terrain: Terrain):
        self.params = params
        self.terrain = terrain
        self.ctx = terrain.ctx

    def _configure(self):
        pod = _abi.Params.from_buffer_copy(self.params.pod)
        if self.KIND is not None:
            pod.generator = self.KIND
        self.ctx.check(self.ctx.lib.atmrt_set_params(self.ctx.handle, C.byref(pod)))
        self.ctx.check(self.ctx.lib.atmrt_set_atmosphere(self.ctx.handle, C.byref(self.params.atmosphere)))
        objs = self.params.objects
        arr = (_abi.Object * max(1, len(objs)))(*objs)
        self.ctx.check(self.ctx.lib.atmrt_objects_set(self.ctx.handle, arr, len(objs)))
        return pod

    def generate(self) -> ResultPixels:
        self._configure()
        res = _abi.Result()
        self.ctx.check(self.ctx.lib.atmrt_generate(self.ctx.handle, C.byref(res)))
        try:
            return ResultPixels(_abi.result_to_numpy(res))
        finally:
            self.ctx.lib.atmrt_result_free(C.byref(res))

    def generate_device(self, planes: "_abi.DevicePlanes"):
        """Leave the first-hit planes in HBM (caller-owned device memory).  Returns (ray_steps, device_ms)."""
        self._configure()
        steps, ms = C.c_uint64(), C.c_double()
        self.ctx.check(self.ctx.lib.atmrt_generate_device(self.ctx.handle, C.byref(planes), C.byref(steps), C.byref(ms)))
        return steps.value, ms.value


    def generate_image_device(self, images):
        """The WHOLE [H][W] frame left in HBM on every device of the context (atmrt_generate_image_device): `images` is one
        _abi.DevicePlanes per device (a single one for a plain or rank context).  Returns (ray_steps, device_ms)."""
        self._configure()
        if isinstance(images, _abi.DevicePlanes):
            images = [images]
        arr = (_abi.DevicePlanes * len(images))(*images)
        steps, ms = C.c_uint64(), C.c_double()
        self.ctx.check(self.ctx.lib.atmrt_generate_image_device(self.ctx.handle, arr, C.byref(steps), C.byref(ms)))
        return steps.value, ms.value

    def image_hits_device(self, height, width, skip=()):
        """The trace-point lists of the frame generate_image_device just produced, in the image's pixel order, as torch tensors on
        every device of the context (one dict for a plain or rank context, a list for a multi-device one; the devices whose index
        is in `skip` take part in the exchange but get nothing: None): atmrt_image_hits_device.  The total is known to every rank
        since the frame's own collective (no communication); the fill is ONE collective over the ranks."""
        import torch
        n = C.c_uint64()
        self.ctx.check(self.ctx.lib.atmrt_image_hits_device(self.ctx.handle, None, C.byref(n)))
        devs = self.ctx.devices
        ts = [None if i in skip else _hit_tensors(n.value, height, width, torch.device("cuda", d)) for i, d in enumerate(devs)]
        pods = (_abi.DeviceHits * len(ts))(*[_abi.DeviceHits() if t is None else
                                             _abi.DeviceHits(capacity=n.value, **{k: v.data_ptr() for k, v in t.items()}) for t in ts])
        self.ctx.check(self.ctx.lib.atmrt_image_hits_device(self.ctx.handle, pods, None))
        return ts[0] if len(ts) == 1 else ts

    def last_hits_device(self, height, width):
        """Complete trace-point lists of the frame generate_device just produced, as torch tensors on the context's device:
        {hit_offset [H][W], lat, lon, distance, elevation, path_length, normal [n][3], color_tag, rgba [n][4]}."""
        import torch
        n = C.c_uint64()
        self.ctx.check(self.ctx.lib.atmrt_last_hits_device(self.ctx.handle, None, C.byref(n)))
        n = n.value
        t = _hit_tensors(n, height, width, torch.device("cuda", self.ctx.device))
        pod = _abi.DeviceHits(capacity=n, **{k: v.data_ptr() for k, v in t.items()})
        self.ctx.check(self.ctx.lib.atmrt_last_hits_device(self.ctx.handle, C.byref(pod), None))
        return t

    def last_stats(self):
        """atmrt_frame_stats_t of the last frame: how often it left the fast routes of the device path."""
        t = _abi.FrameStats()
        self.ctx.check(self.ctx.lib.atmrt_last_stats(self.ctx.handle, C.byref(t)))
        return {k: getattr(t, k) for k, _ in _abi.FrameStats._fields_}

    def last_timings(self):
        t = _abi.Timings()
        self.ctx.check(self.ctx.lib.atmrt_last_timings(self.ctx.handle, C.byref(t)))
        return {k: getattr(t, k) for k, _ in _abi.Timings._fields_}


def _hit_tensors(n, height, width, dev):
    """Device arrays laid out like the hit arrays of atmrt_result_t (atmrt_device_hits_t)."""
    import torch
    f64 = dict(dtype=torch.float64, device=dev)
    t = {k: torch.empty(n, **f64) for k in ("lat", "lon", "distance", "elevation", "path_length")}
    t["normal"] = torch.empty((n, 3), **f64)
    t["rgba"] = torch.empty((n, 4), **f64)
    t["color_tag"] = torch.empty(n, dtype=torch.int32, device=dev)
    t["hit_offset"] = torch.empty((height, width), dtype=torch.int64, device=dev)
    return t


def image_planes(height, width, dev):
    """[H][W] planes of a whole frame in HBM + the atmrt_device_planes_t that points at them."""
    import torch
    f64 = dict(dtype=torch.float64, device=dev)
    t = {k: torch.empty((height, width), **f64) for k in ("azimuth", "elevation_angle", "lat", "lon", "distance", "elevation", "path_length")}
    t["normal"] = torch.empty((3, height, width), **f64)
    t["hit_count"] = torch.empty((height, width), dtype=torch.int32, device=dev)
    return t, _abi.DevicePlanes(**{k: v.data_ptr() for k, v in t.items()})


class FastGenerator(Generator):
    KIND = _abi.GENERATORS["Fast"]


class RectilinearGenerator(Generator):
    KIND = _abi.GENERATORS["Rectilinear"]


class InterpolatingRectilinearGenerator(Generator):
    KIND = _abi.GENERATORS["InterpolatingRectilinear"]


def make_generator(params: Params, terrain: Terrain) -> Generator:
    """The `match params.output.generator` of generator::generate (src/generator/mod.rs:72-78)."""
    return {0: FastGenerator, 1: InterpolatingRectilinearGenerator, 2: RectilinearGenerator}[params.pod.generator](params, terrain)


# ---- renderer::draw_image (src/renderer/mod.rs:385-414) on the device -----------------------------
def into_coloring(lib, params_pod, conf):
    """ConfColoring::into_coloring (params.rs:231-277) -> atmrt_coloring_t."""
    col = _abi.Coloring()
    rc = lib.atmrt_coloring_from_conf(C.byref(params_pod), conf["kind"], conf["water_level"], conf["ambient_light"],
                                      conf["light_zenith_angle"], conf["light_dir"], conf["palette"], conf["has_fog"],
                                      conf["fog_distance"], C.byref(col))
    if rc != 0:
        raise AtmrtError(rc, "invalid colouring configuration")
    return col


def draw_image(ctx, coloring, width, height):
    """Composite the frame of the last generate() on `ctx` into an RGB8 image [height][width][3]."""
    rgb = np.zeros((height, width, 3), dtype=np.uint8)
    ctx.check(ctx.lib.atmrt_draw_image(ctx.handle, C.byref(coloring), rgb.ctypes.data))
    return rgb


# ---- integrator / sampler harnesses (ray_path.rs, atm_printer.rs, elev_profile.rs) -------------
def ray_paths(ctx, h0, angles_deg, step, n_steps, straight=False):
    ang = np.ascontiguousarray(angles_deg, dtype=np.float64)
    x = np.zeros((ang.size, n_steps + 1))
    h = np.zeros((ang.size, n_steps + 1))
    ctx.check(ctx.lib.atmrt_ray_paths(ctx.handle, h0, ang.size, ang.ctypes.data, int(straight), step, n_steps,
                                      x.ctypes.data, h.ctypes.data))
    return x, h


def atmosphere_sample(ctx, altitudes):
    alt = np.ascontiguousarray(altitudes, dtype=np.float64)
    outs = [np.zeros_like(alt) for _ in range(4)]
    ctx.check(ctx.lib.atmrt_atmosphere_sample(ctx.handle, alt.size, alt.ctypes.data, *[o.ctypes.data for o in outs]))
    return dict(zip(("temperature", "pressure", "n", "dn_dh"), outs))


def coords_at_dist(ctx, lat0, lon0, dir_deg, dists):
    d = np.ascontiguousarray(dists, dtype=np.float64)
    lat, lon = np.zeros_like(d), np.zeros_like(d)
    ctx.check(ctx.lib.atmrt_coords_at_dist(ctx.handle, lat0, lon0, dir_deg, d.size, d.ctypes.data, lat.ctypes.data,
                                           lon.ctypes.data))
    return lat, lon
