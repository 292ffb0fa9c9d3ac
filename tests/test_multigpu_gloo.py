"""N > 1 path on CPU: world_size-2 `gloo` run of the sharding + reduce logic bench.py uses on RCCL.

Each rank holds the frame it would have rendered -- the full frame masked to its own 64x64 tiles (tile t -> rank
t % world, `glz_host_tile_owner`, the same rule `glz_renderer_set_partition` applies on the device) -- and
`glaze_amd.distributed.reduce_frame` sums them onto rank 0.  Because tiles are disjoint the reduced frame must be
bit-identical to the single-process frame.  The pixel data comes from the CPU oracle (there is no GPU here).
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from glaze_amd.distributed import reduce_frame, tile_owner
    from glaze_amd.scenes import cube_scene
    from oracle.pyoracle import OracleRenderer, OracleScene

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        r = OracleRenderer(OracleScene(cube_scene()), w, h, threads=1)
        r.set_depth(2)
        r.draw(2)
        full = r.read_hdr()
        own = tile_owner(w, h, world)
        mine = np.where((own == rank)[..., None], full, 0.0).astype(np.float32)
        assert mine[own != rank].sum() == 0
        frame = torch.from_numpy(mine.copy())
        dist.barrier()
        reduce_frame(frame, dst=0)
        dist.barrier()
        if rank == 0:
            np.save(os.path.join(out_dir, "reduced.npy"), frame.numpy())
            np.save(os.path.join(out_dir, "full.npy"), full)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_gloo_reduce_is_bit_identical(tmp_path):
    import torch.multiprocessing as mp
    w, h = 200, 136
    mp.spawn(_worker, args=(2, _free_port(), w, h, str(tmp_path)), nprocs=2, join=True)
    reduced = np.load(str(tmp_path / "reduced.npy"))
    full = np.load(str(tmp_path / "full.npy"))
    assert reduced.shape == (h, w, 4)
    assert np.array_equal(reduced.view(np.uint32), full.view(np.uint32))
    assert (reduced[..., 3] == 4.0).all()
