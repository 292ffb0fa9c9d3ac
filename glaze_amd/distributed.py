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


def gather_frame(renderer, frame, which=0, dst=0, group=None):
    """The packed-tile exchange of a one-process-per-GPU job: every rank hands over its own tiles only (1 / world of the
    frame's bytes: ncclSend / ncclRecv pairs inside one group = `dist.gather` on the "nccl" backend, each pair on its own
    xGMI link) and `dst` puts them into `frame` (a full-frame RGBA32F CUDA tensor); the other ranks' `frame` is untouched.

    `renderer` must have its partition set to (rank, world) of the group.  Bit-identical to `reduce_frame` of the
    zero-padded frames.
    """
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n_max = renderer.packed_pixels(0, world)      # rank 0 owns the most tiles; everybody sends that many (gather wants equal sizes)
    # The buffers are kept between calls (a frame is gathered once per timed region: no allocation inside it) and made with
    # torch.empty: a torch.zeros would enqueue its fill on torch's CURRENT stream, while export_packed writes the same buffer on the
    # renderer's own non-blocking stream -- nothing orders the two, and a late fill would wipe the tiles.  scatter_packed only reads
    # the packed_pixels(r) pixels rank r really owns, so the tail of a shorter rank's buffer is never looked at.
    key = (str(frame.device), n_max, world, rank == dst, id(group))
    cache = getattr(renderer, "_gather_buffers", None)
    if cache is None or cache[0] != key:
        packed = torch.empty((n_max, 4), dtype=torch.float32, device=frame.device)
        # one buffer for everybody's part: the receiving side then takes ONE call (scatter_packed_all), not a kernel + a synchronisation per rank
        whole = torch.empty((world, n_max, 4), dtype=torch.float32, device=frame.device) if rank == dst else None
        parts = [whole[r] for r in range(world)] if rank == dst else None
        cache = (key, packed, parts, whole)
        renderer._gather_buffers = cache          # (about a frame's worth on `dst`; release_gather_buffers() drops them)
    _, packed, parts, whole = cache
    if frame.is_cuda:
        # whatever torch / RCCL still have in flight on `packed` on torch's CURRENT stream (the allocator's work, the previous call's send on
        # a rank that is not `dst`) is done before the renderer's stream writes it again -- that stream only, not the whole device
        torch.cuda.current_stream(frame.device).synchronize()
    renderer.export_packed(which, packed.data_ptr())      # synchronised on return (the renderer's stream has finished writing `packed`)
    dist.gather(packed, parts, dst=dst, group=group)
    if rank == dst:
        if frame.is_cuda:
            torch.cuda.current_stream(frame.device).synchronize()      # the gather's receives are done: the scatter below runs on the renderer's own stream
        # every rank's tiles, dst's own among them (the gather put `packed` into parts[dst]), to their place: the tiles of all ranks cover the frame
        renderer.scatter_packed_all(world, whole.data_ptr(), n_max, frame.data_ptr())
    return frame


def release_gather_buffers(renderer):
    """Drops the staging buffers gather_frame keeps on the renderer between calls."""
    if getattr(renderer, "_gather_buffers", None) is not None:
        renderer._gather_buffers = None
