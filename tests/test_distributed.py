"""world_size-2 gloo tests (CPU) of the N>1 plumbing used by bench.py: shard partition, max-over-ranks timing,
cap all-gather and whole-job throughput aggregation."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from olavm_amd import sharding
    lo, hi = sharding.shard_range(94, rank, world)
    # each rank "measures" a different step time; the job time is the max
    t = sharding.max_over_ranks([1.0 + rank, 5.0 - rank])
    # per-rank cap slice: the cosets this rank owns (2 cap entries per coset)
    mine = [c for c in range(8) if sharding.coset_owner(c, world) == rank]
    cap = np.array([[c * 2 + k, rank, 0, 0] for c in mine for k in range(2)], dtype=np.int64)
    gathered = sharding.all_gather_caps(cap)
    dist.barrier()
    q.put((rank, lo, hi, t, [g.numpy().tolist() for g in gathered]))
    dist.destroy_process_group()


def test_two_rank_gloo_plumbing():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, t0, g0), (r1, lo1, hi1, t1, g1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 47, 47, 94)            # columns partitioned without overlap
    assert t0 == t1 == [2.0, 5.0]                             # max over ranks, identical everywhere
    assert g0 == g1                                           # every rank sees the same gathered caps
    full = {}
    for rank_slice in g0:
        for e in rank_slice:
            full[e[0]] = e[1]
    assert sorted(full) == list(range(16))                    # all 16 cap entries present exactly once
    assert all(full[i] == (i // 2) * 2 // 8 for i in range(16))   # entries 2c, 2c+1 came from the owner of coset c
    assert [e[0] for rank_slice in g0 for e in rank_slice] == list(range(16))   # rank-ordered concatenation IS the cap


def test_partition_and_aggregate():
    from olavm_amd import sharding
    for total in (1, 7, 8, 94, 184):
        for world in (1, 2, 4, 8):
            spans = [sharding.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    assert sharding.aggregate_throughput(94, 16.0 * (1 << 22), 8, 10, 0.06) == pytest.approx(94 * 16 * 4194304 * 8 * 10 / 0.06 / 1e9)
