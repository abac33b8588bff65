"""Pixel-column tile sharding across the GPUs of one node: rank g of G owns columns
[g*W/G, (g+1)*W/G); rows and the terrain mosaic are replicated (pixels are independent, rectilinear.rs:32-37).

TEST INFRASTRUCTURE, not product code: the product's exchange is C++ below the C ABI (csrc/atmrt_multi.hip:
atmrt_ctx_create_multi, atmrt_ctx_comm_init_rank, atmrt_generate_image_device, atmrt_image_hits_device — hand-written against
RCCL) and cannot run without a GPU.  This module is the torch.distributed statement of the same layout, the CPU-testable model
that tests/test_distributed_cpu.py runs at world size 2 over gloo; nothing outside tests/ imports it.

The only exchange is at the end of the frame and it is ONE collective: every rank's result planes live in one contiguous
slab (`PlaneSlab`: 7 f64 planes + the planar normal + hit_count, 84 B per pixel), `ImageGather` all-gathers the slabs with a
single all_gather_into_tensor and permutes the rank-major slabs into the [H][W] row-major image `result[y][x]` (fast.rs:52-92)
— the permutation is part of the step, so what a caller times is the time to the finished image.  Frames whose pixels hold
several trace points (translucent terrain, scene objects — BASELINE config 5) additionally exchange the variable-length lists:
`gather_hits` needs one small all-gather of the per-rank totals (the only host synchronisation: buffer sizes must be known),
one all-gather of the lists packed into one padded f64 block, and one device-side gather into the image's pixel order.

xGMI is point-to-point (7 links per GPU): a shard of W/8 columns is 92 MB at the 8192x4096 size, one link's worth per peer, so the
direct all-gather is bound by a single link (~153 GB/s peak) and costs a few milliseconds against hundreds for the march."""
import torch

F64_PLANES = ("azimuth", "elevation_angle", "lat", "lon", "distance", "elevation", "path_length")
HIT_COLUMNS = (("lat", 1), ("lon", 1), ("distance", 1), ("elevation", 1), ("path_length", 1), ("normal", 3), ("rgba", 4), ("color_tag", 1))


def column_shard(width, rank, world):
    """Columns [c0, c1) of rank `rank`.  Equal widths are required: all_gather_into_tensor needs the same count on every rank
    (unequal shards would hang or corrupt the collective), so a width that does not divide is refused before any collective."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    if width % world != 0:
        raise ValueError(f"image width {width} is not a multiple of the {world} ranks: pixel-column tiles must be equal "
                         f"(use a width that divides, e.g. {width - width % world})")
    wl = width // world
    return rank * wl, (rank + 1) * wl


class PlaneSlab:
    """The first-hit planes of one shard ([H][wl] each, what atmrt_generate_device writes) as views of ONE contiguous buffer."""

    def __init__(self, height, wl, device):
        self.height, self.wl = height, wl
        n = height * wl
        self.n_f64 = 10 * n                       # 7 planes + normal [3][H][wl]
        self.nbytes = self.n_f64 * 8 + n * 4      # + hit_count (int32)
        self.buf = torch.empty(self.nbytes, dtype=torch.uint8, device=device)
        self.planes = self.views(self.buf, 1)
        self.planes = {k: v[0] for k, v in self.planes.items()}

    def views(self, buf, g):
        """{name: [g, (3,) H, wl]} views of a buffer holding g slabs back to back."""
        h, wl, n = self.height, self.wl, self.height * self.wl
        b = buf.view(g, self.nbytes)
        f = b[:, :self.n_f64 * 8].view(torch.float64).view(g, 10, h, wl)  # every slab is a multiple of 4 bytes; f64 part first (8-aligned)
        out = {k: f[:, i] for i, k in enumerate(F64_PLANES)}
        out["normal"] = f[:, 7:10]
        out["hit_count"] = b[:, self.n_f64 * 8:].view(torch.int32).view(g, h, wl)
        return out

    def device_planes(self):
        """atmrt_device_planes_t pointing into the slab."""
        from atm_raytracer_amd import _abi
        return _abi.DevicePlanes(**{k: v.data_ptr() for k, v in self.planes.items()})


class ImageGather:
    """One all-gather per frame: slabs -> gathered [G][slab] -> image planes [H][W] (normal [3][H][W])."""

    def __init__(self, slab: PlaneSlab, world, via_host=False):
        if slab.nbytes % 8 != 0:  # an odd pixel count would leave the f64 planes of every second slab misaligned in `gathered`
            raise ValueError("height x shard width must be even")
        self.slab, self.world, self.via_host = slab, world, via_host
        dev = slab.buf.device
        h, w = slab.height, slab.wl * world
        self.gathered = torch.empty(world * slab.nbytes, dtype=torch.uint8, device=dev)
        self.image = {k: torch.empty((h, w), dtype=torch.float64, device=dev) for k in F64_PLANES}
        self.image["normal"] = torch.empty((3, h, w), dtype=torch.float64, device=dev)
        self.image["hit_count"] = torch.empty((h, w), dtype=torch.int32, device=dev)

    def __call__(self, dist):
        s = self.slab
        if self.via_host:  # rehearsal over gloo with device tensors: stage through host memory (a functional check only)
            host = torch.empty(self.gathered.shape, dtype=torch.uint8)
            dist.all_gather_into_tensor(host, s.buf.cpu())
            self.gathered.copy_(host)
        else:
            dist.all_gather_into_tensor(self.gathered, s.buf)  # the frame's single collective
        g, h, wl = self.world, s.height, s.wl
        for k, v in s.views(self.gathered, g).items():  # [G, (3,) H, wl] -> [(3,) H, G, wl]: rank-major shards to image rows
            dst = self.image[k]
            dst.view(*dst.shape[:-1], g, wl).copy_(v.movedim(0, -2))
        return self.image


def assemble(gathered):
    """[G, ..., H, wl] (rank-major, as all_gather_into_tensor fills it) -> [..., H, G*wl] row-major image."""
    g = gathered.shape[0]
    moved = gathered.movedim(0, -2)                       # [..., H, G, wl]
    return moved.reshape(*moved.shape[:-2], g * gathered.shape[-1])


def pack_hits(hits, n_rows):
    """{name: [n, ...]} trace-point arrays -> one f64 block [n_rows, 13] (color_tag as 0.0 / 1.0), zero-padded."""
    n = hits["lat"].shape[0]
    block = torch.zeros((n_rows, 13), dtype=torch.float64, device=hits["lat"].device)
    c = 0
    for name, width in HIT_COLUMNS:
        block[:n, c:c + width] = hits[name].reshape(n, width).to(torch.float64)
        c += width
    return block


def unpack_hits(block):
    out, c = {}, 0
    for name, width in HIT_COLUMNS:
        v = block[:, c:c + width]
        out[name] = v.reshape(-1).contiguous() if width == 1 else v.contiguous()
        c += width
    out["color_tag"] = out["color_tag"].to(torch.int32)
    return out


def gather_hits(hit_count_image, hits_local, world, dist, via_host=False):
    """Variable-length trace-point lists of the column shards -> the lists of the whole image in its pixel order p = y*W + x.

    hit_count_image: [H, W] counts of the WHOLE image (the hit_count plane `ImageGather` has just produced); hits_local:
    {name: tensor [n_local, ...]} in the shard's own pixel order (what atmrt_last_hits_device returns).
    Returns (hit_offset [H, W] int64, {name: tensor [n_total, ...]}).  One host synchronisation (the per-rank totals, needed to
    size the buffers), two collectives, no per-rank loop: every index below is computed for all ranks at once on the device."""
    h, w = hit_count_image.shape
    wl = w // world
    dev = hit_count_image.device
    counts = hit_count_image.to(torch.int64)
    n_local = torch.tensor([hits_local["lat"].shape[0]], dtype=torch.int64, device=dev)
    n_rank_t = torch.empty(world, dtype=torch.int64, device=dev)
    if via_host:
        tmp = torch.empty(world, dtype=torch.int64)
        dist.all_gather_into_tensor(tmp, n_local.cpu())
        n_rank_t.copy_(tmp)
    else:
        dist.all_gather_into_tensor(n_rank_t, n_local)
    n_rank = n_rank_t.tolist()                                  # the one synchronisation
    n_total, n_max = sum(n_rank), max(max(n_rank), 1)
    block = pack_hits(hits_local, n_max)
    buf = torch.empty((world * n_max, 13), dtype=torch.float64, device=dev)
    if via_host:
        tmp = torch.empty(buf.shape, dtype=torch.float64)
        dist.all_gather_into_tensor(tmp, block.cpu())
        buf.copy_(tmp)
    else:
        dist.all_gather_into_tensor(buf, block)
    flat = counts.reshape(-1)
    goff = torch.cumsum(flat, 0) - flat                         # exclusive scan in the image's pixel order
    shard_major = counts.view(h, world, wl).permute(1, 0, 2).reshape(world, h * wl)  # every rank's counts in ITS pixel order
    loff = torch.cumsum(shard_major, 1) - shard_major + (torch.arange(world, device=dev) * n_max).view(world, 1)
    src_off = loff.view(world, h, wl).permute(1, 0, 2).reshape(-1)  # start of each image pixel's points inside `buf`
    pix = torch.repeat_interleave(torch.arange(h * w, device=dev), flat, output_size=n_total)
    src = src_off[pix] + (torch.arange(n_total, device=dev) - goff[pix])
    return goff.reshape(h, w), unpack_hits(buf[src])
