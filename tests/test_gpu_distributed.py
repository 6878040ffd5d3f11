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


def _prove_worker(rank, world, port, q, log_n=12, real=False, hasher="poseidon"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from olavm_amd.air import ola_tables as T
    from olavm_amd.backend import Backend
    from tests import tracegen
    if real:
        # a real execution against the full-size fixed tables: live cross-table products on sharded and replicated tables
        from olavm_amd.air import miniexec as M
        blob = T.ola_stark().blob()
        traces, params, compress = M.instance(M.memory_program(600), range_bits=16, limb_bits=8, max_steps=1 << 20)
    else:
        blob = T.ola_stark(range_bits=4, limb_bits=2).blob()
        # 2^12-row (or 2^15-row: the large-transform kernels) tables: CPU, memory and Poseidon run on the coset partition,
        # the small / low-degree ones replicated
        traces, params, compress = tracegen.empty_program_instance(log_n=log_n, live=np.random.default_rng(12))
    be = Backend(device=0, hasher=hasher)
    be.set_shard(rank, world)
    sharded = be.prove_with_traces(blob, traces, params, compress)
    calls = be.shard_calls
    single = None
    dist.barrier()
    if rank == 0:
        be.set_shard(0, 1)
        single = be.prove_with_traces(blob, traces, params, compress)
    q.put((rank, sharded, single, calls, [int(x) for x in params]))
    be.close()
    dist.barrier()
    dist.destroy_process_group()


# (round 6: the 8-rank and the 4-rank Poseidon cases left the default run -- 100 s of process start-up on the one shared GPU; 8 ranks
# stay covered by test_coset_partitioned_proof_of_a_real_execution[8] and by the 8-rank multi-device context tests, 4 ranks by the
# Blake3 case.  OLA_FULL_SUITE=1 brings them back.)
_FULL = os.environ.get("OLA_FULL_SUITE") == "1"
@pytest.mark.parametrize("world,log_n,hasher", [(2, 12, "poseidon"), (2, 15, "poseidon"), (2, 12, "blake3"), (4, 12, "blake3")]
                         + ([(4, 12, "poseidon"), (8, 12, "poseidon")] if _FULL else []))
def test_coset_partitioned_proof_equals_the_single_gpu_proof(world, log_n, hasher, oracle):
    """SURVEY 8e end to end: `world` ranks (sharing the one test GPU, gloo for the exchanges) each prove with their share
    of the cosets; every rank's AllProof bytes equal the single-GPU proof, which the oracle verifier accepts."""
    import torch.multiprocessing as mp
    from olavm_amd.air import ola_tables as T
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_prove_worker, args=(r, world, port, q, log_n, False, hasher)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    single = res[0][2]
    assert single is not None and len(single) > 1000
    for rank, sharded, _, calls, _p in res:
        assert sharded == single, "rank %d produced different proof bytes" % rank
        # every table of 2^12 rows and more runs on the coset partition: per table one all-gather of the trace values, 3 of cap
        # slices, 2 quotient planes, the opening evaluations, the first FRI layer's cap slices, 3 of opened rows and one of the
        # first FRI layer's opened leaves
        assert calls > 0 and calls % 12 == 0, calls
    blob = T.ola_stark(range_bits=4, limb_bits=2).blob()
    with oracle.hasher(hasher):          # either of the reference's hash configurations (plonk/config.rs:112-161)
        rc, why = oracle.verify_all_proof(blob, single, res[0][4])
    assert rc == 0, why


@pytest.mark.parametrize("world", [2, 8])
def test_coset_partitioned_proof_of_a_real_execution(oracle, world):
    """2 and 8 ranks prove the executor's memory program (2^14 CPU rows, 2^15-row program table) against ola_stark() with its
    full-size range-check and bitwise tables: bytes equal to the single-GPU proof, verifier accepts.  Every large table is on
    the coset partition -- the program, range-check and bitwise tables too, whose quotients live on fewer cosets than ranks
    (only the owners of those cosets evaluate them) -- and the trace values reach the ranks by column-sharded upload + all-gather."""
    import torch.multiprocessing as mp
    from olavm_amd.air import ola_tables as T
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_prove_worker, args=(r, world, port, q, 0, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    single = res[0][2]
    assert single is not None and len(single) > 1000
    for rank, sharded, _, calls, _p in res:
        assert sharded == single, "rank %d produced different proof bytes" % rank
        assert calls >= 12 and calls % 12 == 0, calls          # tables on the coset partition (12 exchanges each)
    rc, why = oracle.verify_all_proof(T.ola_stark().blob(), single, res[0][4])
    assert rc == 0, why
