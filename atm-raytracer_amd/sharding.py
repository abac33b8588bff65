"""Pixel-column tile sharding across the GPUs of one node: rank g of G owns columns
[g*W/G, (g+1)*W/G); rows and the terrain mosaic are replicated; the only exchange is one all-gather per
result plane at the end of the frame."""
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
