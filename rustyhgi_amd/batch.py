"""Batch split of independent frames over the GPUs of one node.

The reference codes one frame per call and frames never interact (src/encoder.rs:39), so a batch
shards by frame with NO data-path collective: every rank produces (or is fed) its own frames and
codes them locally.  Collectives (RCCL when the backend is "nccl", i.e. on GPUs; gloo in the CPU
tests) carry only what the split itself needs:
  * broadcast of the coding parameters -- quantizer table, its error bound, levels -- from rank 0;
  * all-gather of small per-rank statistics at the end.
Moving pixels between GPUs would cost more than coding them where they are (xGMI ~1 TB/s out of
one GPU versus ~5 TB/s of HBM per GPU), so it is never done for throughput.  scatter_frames /
gather_frames exist for the one case where the frames really do start on a single GPU, and for the
separately labelled "xgmi_scatter_gather" measurement of bench.py --xgmi-scatter.
"""
import numpy as np


def shard(global_frames, world, rank):
    """Contiguous block of frames owned by `rank`: (first, count).  Blocks differ by at most one frame."""
    base, extra = divmod(int(global_frames), int(world))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def pack_params(table, error, levels):
    """256-entry quantizer table + error bound + levels as one 258-byte message."""
    msg = np.zeros(258, np.uint8)
    msg[:256] = np.asarray(table, np.uint8)
    msg[256], msg[257] = int(error), int(levels)
    return msg


def unpack_params(msg):
    msg = np.asarray(msg, np.uint8)
    return msg[:256].copy(), int(msg[256]), int(msg[257])


def broadcast_params(dist, device, table=None, error=0, levels=0, src=0):
    """Rank `src` supplies (table, error, levels); every rank returns them.  dist=None: single process."""
    import torch
    msg = torch.zeros(258, dtype=torch.uint8, device=device)
    if dist is None or dist.get_rank() == src:
        msg.copy_(torch.from_numpy(pack_params(table, error, levels)))
    if dist is not None:
        dist.broadcast(msg, src=src)
    return unpack_params(msg.cpu().numpy())


def gather_stats(dist, stats):
    """All-gather a small int64 vector from every rank -> (world, len) numpy array."""
    import torch
    if dist is None:
        return stats.detach().cpu().numpy()[None, :]
    out = [torch.zeros_like(stats) for _ in range(dist.get_world_size())]
    dist.all_gather(out, stats)
    return torch.stack(out).cpu().numpy()


def max_over_ranks(dist, value, device):
    """Largest `value` (a float, e.g. elapsed seconds) over the ranks."""
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _through_host(dist, tensor):
    """gloo moves host memory only (its CUDA support stops at broadcast and all-reduce): with that backend -- the CPU tests,
    bench.py --share-gpu -- device frames are staged through host memory.  RCCL takes the device tensors as they are."""
    return dist.get_backend() == "gloo" and tensor.is_cuda


def scatter_frames(dist, all_frames, mine, src=0):
    """Rank `src` holds `all_frames` (world * F frames, rank-major); every rank receives its F frames in `mine`.
    Point-to-point under the hood (one xGMI link per peer), so the source GPU's links bound it."""
    import torch
    if dist is None:
        mine.copy_(all_frames)
        return
    world, rank = dist.get_world_size(), dist.get_rank()
    host = _through_host(dist, mine)
    chunks = None
    if rank == src:
        assert all_frames.shape[0] == world * mine.shape[0], "all_frames must hold world * F frames"
        chunks = [(c.cpu() if host else c).contiguous() for c in all_frames.chunk(world)]
    if host:
        buf = torch.empty(mine.shape, dtype=mine.dtype)
        dist.scatter(buf, chunks, src=src)
        mine.copy_(buf)
    else:
        dist.scatter(mine, chunks, src=src)


def gather_frames(dist, mine, all_frames, dst=0):
    """Inverse of scatter_frames: rank `dst` ends up with every rank's frames, rank-major."""
    import torch
    if dist is None:
        all_frames.copy_(mine)
        return
    world, rank = dist.get_world_size(), dist.get_rank()
    host = _through_host(dist, mine)
    send = mine.cpu() if host else mine
    if rank == dst:
        if host:
            parts = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(world)]
            dist.gather(send, parts, dst=dst)
            for p, c in zip(parts, all_frames.chunk(world)):
                c.copy_(p)
        else:
            parts = [c for c in all_frames.chunk(world)]
            assert all(p.is_contiguous() for p in parts)
            dist.gather(send, parts, dst=dst)
    else:
        dist.gather(send, None, dst=dst)
