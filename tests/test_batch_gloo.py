"""The multi-GPU path of bench.py, rehearsed with two gloo ranks on the CPU: frame sharding,
parameter broadcast and statistics gather (rustyhgi_amd/batch.py).  The ranks code their frames with
the oracle (the GPU library cannot run here); rank 0 checks that the gathered result equals the
single-process result over the whole batch."""
import os
import socket

import numpy as np
import pytest

from conftest import SEED0
from rustyhgi_amd import batch


def test_shard_partitions_the_batch():
    for frames in (0, 1, 7, 8, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [batch.shard(frames, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == frames
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f1 == f0 + c0
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert batch.shard(512, 8, 3) == (192, 64)         # BASELINE config C3: 64 frames per GPU


def test_params_round_trip():
    table = (np.arange(256) * 7 % 256).astype(np.uint8)
    t, e, l = batch.unpack_params(batch.pack_params(table, 20, 4))
    assert (t == table).all() and (e, l) == (20, 4)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, frames, w, h, out):
    import torch
    import torch.distributed as dist
    from oracle import hgi_oracle as O
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        if rank == 0:
            lut, err = O.linear_lut(O.MEDIUM)
            table, error, levels = batch.broadcast_params(dist, dev, lut, err, 4)
        else:
            table, error, levels = batch.broadcast_params(dist, dev)
        first, count = batch.shard(frames, world, rank)
        sq = mx = chk = 0
        for f in range(first, first + count):          # frames are produced where they are coded
            img = O.synth(O.SYNTH_RAMP, SEED0 + 3, f, w, h)
            grid = O.encode(img, levels, table)
            dec = O.decode(grid, levels)
            s, _, m = O.sq_error(img, dec)
            sq, mx, chk = sq + s, max(mx, m), chk + int(grid.astype(np.int64).sum())
        stats = batch.gather_stats(dist, torch.tensor([sq, mx, chk, first, count], dtype=torch.int64))
        slowest = batch.max_over_ranks(dist, 1.0 + rank, dev)
        if rank == 0:
            np.save(out, np.concatenate([stats.reshape(-1), [int(slowest), error, levels]]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_batch_matches_single_process(tmp_path, oracle):
    import torch.multiprocessing as mp
    frames, w, h, world = 5, 160, 96, 2
    out = str(tmp_path / "stats.npy")
    mp.spawn(_worker, args=(world, _free_port(), frames, w, h, out), nprocs=world, join=True)
    got = np.load(out)
    stats, (slowest, error, levels) = got[:-3].reshape(world, 5), got[-3:]
    assert (int(slowest), int(error), int(levels)) == (world, 20, 4)
    assert stats[:, 3].tolist() == [0, 3] and stats[:, 4].tolist() == [3, 2]
    lut = oracle.linear_lut(oracle.MEDIUM)[0]
    sq = mx = chk = 0
    for f in range(frames):
        img = oracle.synth(oracle.SYNTH_RAMP, SEED0 + 3, f, w, h)
        grid = oracle.encode(img, 4, lut)
        s, _, m = oracle.sq_error(img, oracle.decode(grid, 4))
        sq, mx, chk = sq + s, max(mx, m), chk + int(grid.astype(np.int64).sum())
    assert int(stats[:, 0].sum()) == sq and int(stats[:, 1].max()) == mx <= 20 and int(stats[:, 2].sum()) == chk


def _scatter_worker(rank, world, port, F, w, h, out):
    import torch
    import torch.distributed as dist
    from oracle import hgi_oracle as O
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lut = O.linear_lut(O.MEDIUM)[0]
        allf = allo = None
        if rank == 0:      # every frame starts and ends on rank 0 (the labelled xGMI variant of bench.py)
            allf = torch.from_numpy(np.stack([O.synth(O.SYNTH_RAMP, SEED0 + 3, f, w, h) for f in range(world * F)]))
            allo = torch.zeros_like(allf)
        mine = torch.zeros((F, h, w), dtype=torch.uint8)
        batch.scatter_frames(dist, allf, mine)
        coded = torch.from_numpy(np.stack([O.decode(O.encode(m.numpy(), 4, lut), 4) for m in mine]))
        batch.gather_frames(dist, coded, allo)
        if rank == 0:
            np.save(out, allo.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_scatter_code_gather_two_ranks(tmp_path, oracle):
    """frames scattered from rank 0, coded where they land, gathered back in rank-major order"""
    import torch
    import torch.multiprocessing as mp
    F, w, h, world = 2, 96, 64, 2
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_scatter_worker, args=(world, _free_port(), F, w, h, out), nprocs=world, join=True)
    got = np.load(out)
    lut = oracle.linear_lut(oracle.MEDIUM)[0]
    for f in range(world * F):
        img = oracle.synth(oracle.SYNTH_RAMP, SEED0 + 3, f, w, h)
        assert np.array_equal(got[f], oracle.decode(oracle.encode(img, 4, lut), 4)), f
    # single process: plain copies
    a = torch.arange(24, dtype=torch.uint8).reshape(2, 3, 4)
    b, c = torch.zeros_like(a), torch.zeros_like(a)
    batch.scatter_frames(None, a, b)
    batch.gather_frames(None, b, c)
    assert torch.equal(a, c)
