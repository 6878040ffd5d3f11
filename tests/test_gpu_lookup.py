"""Device-side permuted_cols (ola_permuted_cols / ola_permuted_cols_dev) against the oracle's sequential
restatement of circuits/src/stark/lookup.rs:68-132, bit for bit."""
import numpy as np
import pytest

from tests.lookup_cases import cases, P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from olavm_amd.backend import Backend
    b = Backend(device=0)
    yield b
    b.close()


def test_permuted_cols_matches_oracle_on_every_shape_of_lookup(be, oracle):
    rng = np.random.default_rng(77)
    for name, a, b in cases(rng):
        gi, gt = be.permuted_cols(a, b)
        oi, ot = oracle.permuted_cols(a, b)
        assert np.array_equal(gi, oi), name
        assert np.array_equal(gt, ot), name


def test_permuted_cols_random_small_alphabets(be, oracle):
    """Dense coverage of the stack behaviour: tiny alphabets make every interleaving of surplus inputs and surplus table
    values occur (including pops on an empty stack and left-over inputs after the table is exhausted)."""
    rng = np.random.default_rng(5)
    for trial in range(300):
        n = int(rng.integers(1, 40))
        k = int(rng.integers(1, 8))
        a = rng.integers(0, k, n).astype(np.uint64)
        b = rng.integers(0, k + int(rng.integers(0, 3)), n).astype(np.uint64)
        gi, gt = be.permuted_cols(a, b)
        oi, ot = oracle.permuted_cols(a, b)
        assert np.array_equal(gi, oi) and np.array_equal(gt, ot), (trial, a.tolist(), b.tolist(), gt.tolist(), ot.tolist())


def test_permuted_cols_of_the_rangecheck_and_program_tables(be, oracle):
    """The columns the trace generators actually permute: 16-bit limbs against the fixed 0..65535 column (2^17 rows), and a
    program table's executed / listed instruction digests."""
    rng = np.random.default_rng(9)
    n = 1 << 17
    fixed = np.concatenate([np.arange(1 << 16, dtype=np.uint64), np.full(n - (1 << 16), (1 << 16) - 1, dtype=np.uint64)])
    limbs = rng.integers(0, 1 << 16, n).astype(np.uint64)
    limbs[: n // 4] = rng.integers(0, 50, n // 4)        # small values dominate, as with real range checks
    gi, gt = be.permuted_cols(limbs, fixed)
    oi, ot = oracle.permuted_cols(limbs, fixed)
    assert np.array_equal(gi, oi) and np.array_equal(gt, ot)
    from olavm_amd.air import miniexec as M, ola_tables as T
    traces, _, _ = M.instance(M.fibonacci(200))
    prog = traces[10]
    gi, gt = be.permuted_cols(prog[T.COL_PROG_EXEC_COMP_PROG], prog[T.COL_PROG_COMP_PROG])
    assert np.array_equal(gi, prog[T.COL_PROG_EXEC_COMP_PROG_PERM]) and np.array_equal(gt, prog[T.COL_PROG_COMP_PROG_PERM])


def test_permuted_cols_large_resident(be, oracle):
    """2^22 rows, operands resident in HBM (ola_permuted_cols_dev)."""
    import torch
    rng = np.random.default_rng(3)
    n = 1 << 22
    a = rng.integers(0, n // 3, n).astype(np.uint64)
    b = np.arange(n, dtype=np.uint64)
    da, db = torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(b.view(np.int64)).cuda()
    di, dt = torch.empty_like(da), torch.empty_like(da)
    be.permuted_cols_dev(da.data_ptr(), db.data_ptr(), n, di.data_ptr(), dt.data_ptr())
    oi, ot = oracle.permuted_cols(a, b)
    assert np.array_equal(di.cpu().numpy().view(np.uint64), oi)
    assert np.array_equal(dt.cpu().numpy().view(np.uint64), ot)
