"""ONE context spanning several GPUs (ola_gpu_init_multi, SURVEY 8(b) Threading): one process, one ola_prove_with_traces
call, the coset partition and its exchanges inside the library.  The test box has one GPU, so the logical ranks alias device 0:
the worker threads, per-rank streams and pools, the event-ordered peer all-gather and the partition logic are all the real
thing, only the copies are device-local instead of xGMI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small_instance():
    from olavm_amd.air import ola_tables as T
    from tests import tracegen
    blob = T.ola_stark(range_bits=4, limb_bits=2).blob()
    traces, params, compress = tracegen.empty_program_instance(log_n=12, live=np.random.default_rng(12))
    return blob, traces, params, compress


@pytest.mark.parametrize("hasher", ["poseidon", "blake3"])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_multi_device_context_proof_equals_the_single_gpu_proof(world, hasher, small_instance, oracle):
    from olavm_amd.backend import Backend
    blob, traces, params, compress = small_instance
    one = Backend(device=0, hasher=hasher)
    single = one.prove_with_traces(blob, traces, params, compress)
    one.close()
    be = Backend(devices=[0] * world, hasher=hasher)
    assert be.device_count() == world
    be.proof_stats(enable=True)
    multi = be.prove_with_traces(blob, traces, params, compress)
    st = be.proof_stats()
    again = be.prove_with_traces(blob, traces, params, compress)      # the context is reusable, events and pools included
    be.close()
    assert multi == single, "the multi-device context produced different proof bytes"
    assert again == single
    # three tables of the instance are on the partition (CPU, memory, Poseidon at 2^12 rows): 12 exchanges each, done by the
    # library's own all-gather
    assert st["peer_exchanges"] > 0 and st["peer_exchanges"] % 12 == 0, st
    assert st["peer_bytes_moved"] > 0 and st["exchanges"] == st["peer_exchanges"], st
    with oracle.hasher(hasher):
        rc, why = oracle.verify_all_proof(blob, single, [int(x) for x in params])
    assert rc == 0, why


def test_multi_device_context_real_execution_8_ranks(oracle):
    """An executed program against the full-size fixed tables, 8 logical ranks in one process: the program, range-check and bitwise
    tables are on the partition too, with quotients that live on fewer cosets than there are ranks."""
    from olavm_amd.air import miniexec as M, ola_tables as T
    from olavm_amd.backend import Backend
    blob = T.ola_stark().blob()
    traces, params, compress = M.instance(M.memory_program(600), range_bits=16, limb_bits=8, max_steps=1 << 20)
    one = Backend(device=0)
    single = one.prove_with_traces(blob, traces, params, compress)
    one.close()
    be = Backend(devices=[0] * 8)
    multi = be.prove_with_traces(blob, traces, params, compress)
    be.close()
    assert multi == single
    rc, why = oracle.verify_all_proof(blob, single, [int(x) for x in params])
    assert rc == 0, why


def test_multi_device_context_large_transform_sizes():
    """2^15-row tables (the three-launch transform kernels, 2^18-point LDEs) on 2 logical ranks."""
    from olavm_amd.air import ola_tables as T
    from olavm_amd.backend import Backend
    from tests import tracegen
    blob = T.ola_stark(range_bits=4, limb_bits=2).blob()
    traces, params, compress = tracegen.empty_program_instance(log_n=15, live=np.random.default_rng(15))
    one = Backend(device=0)
    single = one.prove_with_traces(blob, traces, params, compress)
    one.close()
    be = Backend(devices=[0, 0])
    assert be.prove_with_traces(blob, traces, params, compress) == single
    be.close()


def test_multi_device_context_other_entry_points_and_errors(small_instance):
    """Outside ola_prove_with_traces a multi-device context behaves like a single-device one on devices[0]; a trace that violates
    its AIR fails on every rank without hanging the others; bad arguments are refused."""
    from olavm_amd.backend import Backend, OlaGpuError
    from tests.oracle_lib import rand_field
    blob, traces, params, compress = small_instance
    be = Backend(devices=[0, 0])
    vals = rand_field(np.random.default_rng(5), (4, 1 << 10))
    one = Backend(device=0)
    a, b = be.commit(vals), one.commit(vals)
    assert np.array_equal(a.cap(), b.cap())
    a.free(); b.free(); one.close()
    bad = [np.array(t, copy=True) for t in traces]
    from olavm_amd.air import ola_tables as T
    from tests import tracegen
    bad[0][T.COL_OPCODE, 3] = (int(bad[0][T.COL_OPCODE, 3]) + 1) % tracegen.P      # one wrong cell of the CPU table (the oracle's check_constraints flags it)
    with pytest.raises(OlaGpuError, match="not divisible") as ei:
        be.prove_with_traces(blob, bad, params, compress)
    assert ei.value.code == -4 and "rank" in str(ei.value), ei.value      # reported with the rank that saw it
    assert be.prove_with_traces(blob, traces, params, compress)[:4] == (12).to_bytes(4, "little")   # still usable
    from olavm_amd.backend import ALL_GATHER_FN
    cb = ALL_GATHER_FN(lambda user, send, recv, nbytes: 1)
    assert be.lib.ola_set_shard(be.ctx, 0, 2, cb, None) == -1      # the partition of a multi-device context is its own
    be.close()
    with pytest.raises(OlaGpuError):
        Backend(devices=[0, 0, 0])                    # 1, 2, 4 or 8
    with pytest.raises(OlaGpuError):
        Backend(devices=[0, 99])


def test_partition_accounting_of_a_single_gpu_proof(small_instance):
    """ola_gpu_proof_stats / ola_gpu_phase_stats on ONE GPU: the bracketed (dividable) kernel time is positive and below the wall
    time, the exchanges a partitioned run would perform are counted (12 per table of 2^12 rows and more), the per-family counters add up to the instance (every leaf and node of every tree of the proof), and the
    same proof on a 2-rank context reports exactly that many real exchanges."""
    from olavm_amd.backend import Backend
    blob, traces, params, compress = small_instance
    be = Backend(device=0)
    be.proof_stats(enable=True)
    proof = be.prove_with_traces(blob, traces, params, compress)
    st, ph = be.proof_stats(enable=False), be.phase_stats()
    be.close()
    sharded = st["sharded_ms_upto2"] + st["sharded_ms_upto4"] + st["sharded_ms_upto8"]
    assert 0 < sharded < st["wall_ms"], st
    on_partition = sum(1 for t in traces if t.shape[1] >= (1 << 12))
    assert st["exchanges"] == 12 * on_partition and st["exchange_bytes"] > on_partition * (1 << 12) * 8, st
    assert st["peer_exchanges"] == 0
    ms, calls, byts = ph["leaf_hash"]
    assert ms > 0 and calls > 0 and byts > 0
    assert ph["merkle_levels"][1] > 0 and ph["lde"][0] > 0 and ph["quotient"][1] > 0 and ph["fri_fold"][0] > 0
    two = Backend(devices=[0, 0])
    two.proof_stats(enable=True)
    assert two.prove_with_traces(blob, traces, params, compress) == proof
    st2 = two.proof_stats()
    two.close()
    assert st2["peer_exchanges"] == st["exchanges"] == st2["exchanges"], (st, st2)


def test_peer_all_gather_by_itself(tmp_path):
    """tests/gpu_peer_gather_check.cpp: the library's all-gather (olavm_amd/csrc/peer_group.h) without the prover around it -- four rank
    threads, blocks of 512 B ... 64 MB, every rank's result equal to the expected concatenation, also when the send block is
    overwritten right after the exchange."""
    import os
    import shutil
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    # built by __graft_entry__.build() next to the library (the box of the round-end run has no hipcc); rebuilt here when it can be
    exe = os.path.join(os.path.dirname(here), "olavm_amd", "lib", "gpu_peer_gather_check")
    if shutil.which("hipcc"):
        exe = os.path.join(str(tmp_path), "gpu_peer_gather_check")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(here, "gpu_peer_gather_check.cpp")],
                              stderr=subprocess.DEVNULL)
    assert os.path.exists(exe), "olavm_amd/lib/gpu_peer_gather_check is missing: run __graft_entry__.build()"
    r = subprocess.run([exe, "4"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "all ok" in r.stdout and "MISMATCH" not in r.stdout, r.stdout + r.stderr


def test_rccl_carrier_one_rank_communicator_round_trips():
    """OLA_COLLECTIVE=rccl: librccl.so is loaded on request (never linked), ncclCommInitAll over the context's devices, every
    exchange an ncclAllGather on the rank's stream (olavm_amd/csrc/rccl_carrier.h).  What one GPU can exercise: a 1-device
    context gets a 1-rank communicator and its gather returns the rank's own block, for a 512-byte cap block and a 64 MB one."""
    from olavm_amd.backend import Backend, OlaGpuError
    be = Backend(devices=[0], collective="rccl")
    try:
        c = be.collective()
        assert c["carrier"] == "rccl" and c["ranks"] == 1 and c["note"].startswith("RCCL "), c
        for size in (512, 64 << 20):
            ms, bad = be.all_gather_check("rccl", size, reps=3)
            assert bad == 0 and ms >= 0, (size, ms, bad)
        with pytest.raises(OlaGpuError):
            be.all_gather_check("peer", 512)          # a single-device context has no peers
    finally:
        be.close()
    plain = Backend(device=0)
    assert plain.collective()["carrier"] == "none"
    with pytest.raises(OlaGpuError) as e:
        plain.all_gather_check("rccl", 512)
    assert "OLA_COLLECTIVE=rccl" in str(e.value)
    plain.close()


def test_rccl_refused_on_aliased_devices_with_a_clear_reason(small_instance):
    """Logical ranks sharing one physical GPU cannot form an RCCL communicator: the context says so, keeps the library's peer
    carrier, and proves the same bytes."""
    from olavm_amd.backend import Backend, OlaGpuError
    blob, traces, params, compress = small_instance
    one = Backend(device=0)
    single = one.prove_with_traces(blob, traces, params, compress)
    one.close()
    be = Backend(devices=[0, 0], collective="rccl")
    try:
        c = be.collective()
        assert c["carrier"] == "peer" and c["ranks"] == 2, c
        assert "refused" in c["note"] and "share device 0" in c["note"], c
        assert be.prove_with_traces(blob, traces, params, compress) == single
        ms, bad = be.all_gather_check("peer", 512, reps=5)
        assert bad == 0
        with pytest.raises(OlaGpuError) as e:
            be.all_gather_check("rccl", 512)
        assert "share device 0" in str(e.value)
    finally:
        be.close()
    with pytest.raises(OlaGpuError):
        Backend(devices=[0, 0], collective="mpi")


def test_two_physical_gpus_both_carriers_and_the_callers_device():
    """Needs two GPUs (skipped on the one-GPU test box): distinct ordinals, so hipMemcpyPeerAsync / hipDeviceEnablePeerAccess /
    cross-device event waits and -- under OLA_COLLECTIVE=rccl -- a 2-rank communicator really run; the caller's current HIP device
    must be what it was after prove / sync / trim / free."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box")
    from olavm_amd.air import ola_tables as T
    from olavm_amd.backend import Backend
    from tests import tracegen
    blob = T.ola_stark(range_bits=4, limb_bits=2).blob()
    traces, params, compress = tracegen.empty_program_instance(log_n=12, live=np.random.default_rng(12))
    torch.cuda.set_device(0)
    one = Backend(device=0)
    single = one.prove_with_traces(blob, traces, params, compress)
    one.close()
    for carrier in ("peer", "rccl"):
        be = Backend(devices=[0, 1], collective=carrier)
        assert be.collective()["carrier"] == carrier, be.collective()
        assert torch.cuda.current_device() == 0
        assert be.prove_with_traces(blob, traces, params, compress) == single
        assert torch.cuda.current_device() == 0
        be.sync()
        assert torch.cuda.current_device() == 0
        be.trim()
        assert torch.cuda.current_device() == 0
        for size in (512, 64 << 20):
            ms, bad = be.all_gather_check(carrier, size, reps=5)
            assert bad == 0, (carrier, size)
        be.close()
        assert torch.cuda.current_device() == 0
