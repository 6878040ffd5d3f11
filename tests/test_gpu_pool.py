"""The block hand-over between the single-device contexts of one process that share a GPU (csrc/device_ctx.h, DeviceCtx::adopt):
a context that needs a block takes a fitting one out of an idle sibling's cache before it goes to the driver.  Checked through the
C ABI's own figures (ola_gpu_memory_stats): the blocks move, the proofs do not change, OLA_POOL_SHARE=0 keeps a context out."""
import pytest

pytestmark = pytest.mark.gpu


def _instance():
    from olavm_amd.air import ola_tables as T
    from olavm_amd.air import tracegen
    blob = T.ola_stark().blob()
    traces, params, compress = tracegen.empty_program_instance(log_n=10, range_bits=16, limb_bits=8, log_n_cpu=18, log_n_mem=18)
    return blob, traces, params, compress


def test_idle_context_hands_its_cached_blocks_to_a_sibling(oracle, monkeypatch):
    from olavm_amd.backend import Backend
    blob, traces, params, compress = _instance()
    monkeypatch.delenv("OLA_POOL_SHARE", raising=False)
    a = Backend(device=0)
    b = Backend(device=0, hasher="blake3")
    try:
        pa = a.prove_with_traces(blob, traces, params, compress)
        a.sync()
        held_a = a.memory_stats()["reserved"]
        assert held_a > (1 << 30)                       # a 2^18-row proof leaves a few GB in the cache
        # the sibling's first proof is served out of that cache ...
        assert b.memory_stats()["reserved"] < held_a // 16
        pb = b.prove_with_traces(blob, traces, params, compress)
        b.sync()
        after_a, held_b = a.memory_stats()["reserved"], b.memory_stats()["reserved"]
        assert after_a < held_a // 2, (held_a, after_a)
        assert held_b >= held_a - after_a
        # ... and a takes the blocks back for its next one; the bytes of either configuration do not depend on who held what
        assert a.prove_with_traces(blob, traces, params, compress) == pa
        a.sync()
        assert b.memory_stats()["reserved"] < held_b
        assert b.prove_with_traces(blob, traces, params, compress) == pb
        rc, why = oracle.verify_all_proof(blob, pa, params)
        assert rc == 0, why
        with oracle.hasher("blake3"):
            rc, why = oracle.verify_all_proof(blob, pb, params)
            assert rc == 0, why
        # both pools together stay near ONE proof's worth instead of two
        assert a.memory_stats()["reserved"] + b.memory_stats()["reserved"] < held_a * 3 // 2
    finally:
        a.close()
        b.close()


def test_pool_share_switch_keeps_a_context_out(monkeypatch):
    from olavm_amd.backend import Backend
    blob, traces, params, compress = _instance()
    monkeypatch.delenv("OLA_POOL_SHARE", raising=False)
    a = Backend(device=0)
    monkeypatch.setenv("OLA_POOL_SHARE", "0")
    loner = Backend(device=0)
    monkeypatch.delenv("OLA_POOL_SHARE", raising=False)
    try:
        pa = a.prove_with_traces(blob, traces, params, compress)
        a.sync()
        held_a = a.memory_stats()["reserved"]
        assert loner.prove_with_traces(blob, traces, params, compress) == pa          # same configuration, same bytes
        loner.sync()
        assert a.memory_stats()["reserved"] == held_a                                 # took nothing ...
        held_l = loner.memory_stats()["reserved"]
        assert a.prove_with_traces(blob, traces, params, compress) == pa
        a.sync()
        assert loner.memory_stats()["reserved"] == held_l                             # ... and lends nothing
    finally:
        a.close()
        loner.close()
