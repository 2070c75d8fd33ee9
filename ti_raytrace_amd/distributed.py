"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

The reference is single-device.  PT_RGB pixels are independent (per-pixel
RNG stream, per-pixel film word), so the path shards by pixel tile with NO collective on
the data path: rank r renders the tiles t with t % world == r of the linear pixel index
(``tirt_film_create``), every rank holds the full scene + BVH, and the framebuffer is
combined once, after the last sample, by a sum-reduce of the zero-padded films
(12.6 MB at 1024^2: ~0.1 ms on a 7 x 153 GB/s xGMI fabric, so it is issued once per job,
not per frame).
"""
import numpy as np


def tile_owner(p, tile_size, world):
    """Rank that renders linear pixel index p (same rule as the device code)."""
    return (p // tile_size) % world


def local_pixel_count(W, H, rank, world, tile_size):
    NP = W * H
    ntiles = (NP + tile_size - 1) // tile_size
    total = 0
    for t in range(rank, ntiles, world):
        total += min(NP, (t + 1) * tile_size) - t * tile_size
    return total


def reduce_film_tensor(film, dst=0, force=False):
    """Sum-reduce a film tensor (any device/backend) onto ``dst``; no-op without a process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force):
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film


def warmup(ctx, W, H, force=False):
    """Creates the RCCL communicator and runs one film-sized reduce on a scratch tensor, so that the
    first real reduce does not pay for communicator set-up."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)):
        return
    t = torch.zeros((W, H, 3), dtype=torch.float32, device=torch.device("cuda", ctx.device_id))
    dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()


def reduce_film(ctx, W, H, dst=0, force=False):
    """Export this rank's hdr film into a torch CUDA tensor (device-to-device copy through the
    C-ABI), reduce it over RCCL onto ``dst`` and, on ``dst``, import the sum back as the
    context's film.  Returns the tensor (None when running on a single GPU)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)):
        return None
    film = torch.empty((W, H, 3), dtype=torch.float32, device=torch.device("cuda", ctx.device_id))
    ctx.film_export_device(film.data_ptr())
    torch.cuda.synchronize()
    reduce_film_tensor(film, dst, force)
    torch.cuda.synchronize()
    if dist.get_rank() == dst:
        ctx.film_import_device(film.data_ptr())
    return film
