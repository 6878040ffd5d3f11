"""Two ranks sharing the single test GPU: the coset-sharded commitment end to end -- each rank commits its
half of the LDE cosets, the cap slices travel through a real torch.distributed all-gather (gloo here; RCCL when every
rank has its own GPU) and every rank ends up with the cap of the single-GPU commitment."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from olavm_amd import sharding
    from olavm_amd.backend import Backend
    from tests.oracle_lib import rand_field
    vals = rand_field(np.random.default_rng(7), (6, 1 << 12))      # the trace is replicated: same seed on every rank
    be = Backend(device=0)
    batch, cap = sharding.commit_sharded(be, rank, world, cols=vals)
    full = be.commit(vals) if rank == 0 else None
    ok = bool(np.array_equal(cap, full.cap())) if full is not None else True
    lo, hi = sharding.coset_range(rank, world)
    q.put((rank, ok, cap.tolist(), (lo, hi), batch.rate_bits))
    batch.free()
    if full is not None:
        full.free()
    be.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_gpu_sharded_commit():
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (r0, ok0, cap0, rng0, rb0), (r1, ok1, cap1, rng1, rb1) = res
    assert ok0                                  # gathered cap == single-GPU cap (checked on rank 0)
    assert cap0 == cap1                         # identical on both ranks
    assert (rng0, rng1) == ((0, 4), (4, 8)) and rb0 == rb1 == 2
