"""CPU-only, 2 processes over gloo: the N>1 path of the batch driver (frame sharding + the single
gather of finished planes) is correct by construction."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from jxlatte_amd import shard


def test_frames_of_rank_partition():
    for n in (0, 1, 7, 8, 64):
        for world in (1, 2, 3, 8):
            seen = sorted(i for r in range(world) for i in shard.frames_of_rank(n, r, world))
            assert seen == list(range(n))
    assert shard.frames_of_rank(64, 3, 8) == [3, 11, 19, 27, 35, 43, 51, 59]  # 8 frames per GPU (config C5)


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    idx = shard.frames_of_rank(n_frames, rank, world)
    # stand-in for decoded planes: frame i is filled with a pattern that identifies (i, channel)
    local = torch.stack([torch.full((3, 4, 6), float(i)) + torch.arange(3).view(3, 1, 1) * 0.25 for i in idx]) if idx else torch.zeros((0, 3, 4, 6))
    out = shard.gather_planes(local, n_frames, rank, world)
    if rank == 0:
        q.put(out.numpy())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [1, 5, 8])
def test_gather_two_ranks_gloo(n_frames):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert out.shape == (n_frames, 3, 4, 6)
    for i in range(n_frames):
        for c in range(3):
            assert np.all(out[i, c] == i + 0.25 * c)
