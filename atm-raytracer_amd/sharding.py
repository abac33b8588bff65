"""Pixel-column tile sharding across the GPUs of one node: rank g of G owns columns
[g*W/G, (g+1)*W/G); rows and the terrain mosaic are replicated; the only exchange is one all-gather per
result plane at the end of the frame — and, for frames whose pixels hold several trace points (translucent terrain, scene
objects), one all-gather per trace-point array after the hit_count planes."""
import torch


def column_shard(width, rank, world):
    return rank * width // world, (rank + 1) * width // world


def all_gather_planes(local, world, dist):
    """local: {name: tensor [..., H, wl]} with equal wl on every rank -> {name: tensor [..., H, W]}."""
    out = {}
    for k, v in local.items():
        v = v.contiguous()
        # concatenated-along-dim-0 output layout: accepted by both RCCL ("nccl") and gloo
        buf = torch.empty((world * v.shape[0],) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        dist.all_gather_into_tensor(buf, v)
        out[k] = assemble(buf.view((world,) + tuple(v.shape)))
    return out


def assemble(gathered):
    """[G, ..., H, wl] (rank-major, as all_gather_into_tensor fills it) -> [..., H, G*wl] row-major image."""
    g = gathered.shape[0]
    moved = gathered.movedim(0, -2)                       # [..., H, G, wl]
    return moved.reshape(*moved.shape[:-2], g * gathered.shape[-1])


def all_gather_hits(hit_count_local, hits_local, world, dist):
    """Variable-length trace-point lists of column shards -> the lists of the whole image in its pixel order p = y*W + x.

    hit_count_local: [H, wl] counts of this rank's shard; hits_local: {name: tensor [n_local, ...]} ordered by the shard's own
    pixel order (what atmrt_last_hits_device returns).  Steps: all-gather the hit_count planes, exclusive-scan
    them to global offsets, all-gather the arrays padded to the largest shard, scatter every shard's points to their places.
    Returns (hit_count [H, W], hit_offset [H, W] int64, {name: tensor [n_total, ...]})."""
    h, wl = hit_count_local.shape
    dev = hit_count_local.device
    counts = all_gather_planes({"hit_count": hit_count_local}, world, dist)["hit_count"].to(torch.int64)  # [H, W]
    w = wl * world
    flat = counts.reshape(-1)
    goff = torch.cumsum(flat, 0) - flat                                                       # exclusive scan, global pixel order
    n_total = int(flat.sum().item())
    n_rank = [int(counts[:, g * wl:(g + 1) * wl].sum().item()) for g in range(world)]
    n_max = max(max(n_rank), 1)
    out = {}
    dest = []
    for g in range(world):  # where shard g's k-th point goes: offset of its pixel in the whole image + its index within the pixel
        c = counts[:, g * wl:(g + 1) * wl].reshape(-1)
        loff = torch.cumsum(c, 0) - c
        pix = torch.repeat_interleave(torch.arange(h * wl, device=dev), c)
        j = torch.arange(n_rank[g], device=dev) - loff[pix]
        gp = (pix // wl) * w + g * wl + (pix % wl)
        dest.append(goff[gp] + j)
    for k, v in hits_local.items():
        pad = torch.zeros((n_max,) + tuple(v.shape[1:]), dtype=v.dtype, device=dev)
        pad[:v.shape[0]] = v
        buf = torch.empty((world * n_max,) + tuple(v.shape[1:]), dtype=v.dtype, device=dev)
        dist.all_gather_into_tensor(buf, pad)
        full = torch.empty((n_total,) + tuple(v.shape[1:]), dtype=v.dtype, device=dev)
        for g in range(world):
            full[dest[g]] = buf[g * n_max:g * n_max + n_rank[g]]
        out[k] = full
    return counts.to(hit_count_local.dtype), goff.reshape(h, w), out
