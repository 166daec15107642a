"""Frame-level sharding across GPUs: frames of a batch are independent, so
rank r of `world` takes frames r, r + world, ... -- no communication during compute -- and the finished
pixel planes are gathered to rank 0 with ONE collective (RCCL when the tensors live on the GPU, gloo in
the CPU tests)."""
import torch
import torch.distributed as dist


def frames_of_rank(n_frames, rank, world):
    """indices of the frames rank `rank` decodes (round-robin, as frame i -> GPU i mod world)"""
    return list(range(rank, n_frames, world))


def gather_planes(local, n_frames, rank, world, dst=0):
    """local: tensor [k, 3, H, W] with this rank's finished frames (k = len(frames_of_rank)); returns on `dst`
    the tensor [n_frames, 3, H, W] in frame order, None elsewhere. Ranks may hold different k (ragged tail):
    every rank pads to the maximum count so that a single dist.gather suffices."""
    kmax = (n_frames + world - 1) // world
    pad = torch.zeros((kmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    if world == 1:
        return pad[:n_frames]
    recv = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, recv, dst=dst)
    if rank != dst:
        return None
    out = torch.empty((n_frames,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = frames_of_rank(n_frames, r, world)
        if idx:
            out[idx] = recv[r][:len(idx)]
    return out
