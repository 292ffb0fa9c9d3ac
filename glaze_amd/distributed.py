"""Multi-GPU plumbing: one process per GPU, image tiles sharded over ranks, one reduce of the float
HDR accumulator to rank 0 (RCCL over xGMI when the process group backend is "nccl").

The reference is single-device (lib/src/vulkan/device.rs:252-321); this is the build's addition
(SURVEY 5.8 / 8e).  Every pixel's RNG, path state and accumulator texel are private and the launch
sequence (seed_i, offset_i) is identical on all ranks, so rank r simply renders the 64x64 tiles t with
t % world == r (`glz_renderer_set_partition`) and the sum over ranks of the zero-elsewhere frames is
bit-identical to a single-GPU render.  There is no exchange during rendering.
"""
import numpy as np

from . import abi


def tile_owner(width, height, world):
    """uint16 HxW map of the rank owning each pixel (64x64 tiles, tile t -> t % world)."""
    out = np.zeros((height, width), np.uint16)
    abi.check(abi.lib().glz_host_tile_owner(width, height, world, out.ctypes.data))
    return out


def chain_owner(width, height, rank, world, chains=0):
    """(number of launch chains, uint16 HxW map of the chain rendering each pixel of `rank`; 0xFFFF = another rank's pixel)."""
    out = np.zeros((height, width), np.uint16)
    n = abi.check(abi.lib().glz_host_chain_owner(width, height, rank, world, chains, out.ctypes.data))
    return n, out


def reduce_frame(frame, dst=0, group=None, force=False):
    """Sum-reduces a full-frame RGBA32F torch tensor (zero outside the caller's tiles) onto `dst`.

    With backend "nccl" this is one ncclReduce(sum, float) of W*H*4 floats over xGMI (33 MB at 1080p);
    with "gloo" (CPU tests) the same call runs on host tensors.  A one-rank group has nothing to exchange and is skipped
    unless `force` asks for the collective anyway (the GPU test that runs RCCL on a one-GPU box).
    """
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1):
        dist.reduce(frame, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return frame
