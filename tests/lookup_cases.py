"""Input / table pairs for the permuted_cols tests (CPU: oracle vs the Python generator; GPU: device vs oracle)."""
import numpy as np

P = 0xFFFFFFFF00000001


def cases(rng, sizes=(1, 2, 3, 8, 37, 256, 1000, 4096)):
    """-> list of (name, inputs, table), uint64 arrays of equal length."""
    out = []
    for n in sizes:
        fixed = np.arange(n, dtype=np.uint64)
        # a range-check-like lookup: inputs drawn from the table, many repeats, table = 0..n-1
        out.append(("lookup n=%d" % n, rng.integers(0, n, n).astype(np.uint64), fixed))
        # few distinct inputs -> long runs of repeats, deep stack of unused table values
        out.append(("few-distinct n=%d" % n, rng.integers(0, max(1, n // 8), n).astype(np.uint64), fixed))
        # inputs all equal to the LARGEST table value: the table runs out first, left-over inputs take the stack bottom-up
        out.append(("all-max n=%d" % n, np.full(n, n - 1, dtype=np.uint64), fixed))
        out.append(("all-min n=%d" % n, np.zeros(n, dtype=np.uint64), fixed))
        # table with repeated values, inputs partly absent from it (the generator must not assume a valid lookup)
        tab = np.sort(rng.integers(0, max(2, n // 2), n).astype(np.uint64) * np.uint64(3))
        out.append(("dup-table n=%d" % n, rng.integers(0, 2 * n + 3, n).astype(np.uint64), tab))
        # inputs above every table value / below every table value
        out.append(("inputs-above n=%d" % n, rng.integers(5 * n, 6 * n + 1, n).astype(np.uint64), fixed))
        out.append(("inputs-below n=%d" % n, rng.integers(0, 3, n).astype(np.uint64), fixed + np.uint64(10)))
        # alternating surplus: blocks of repeated inputs between blocks of unused table values, stack empties repeatedly
        blk = np.repeat(np.arange(0, n, 4, dtype=np.uint64), 4)[:n]
        out.append(("blocks n=%d" % n, blk, fixed))
        out.append(("blocks-shifted n=%d" % n, (blk + np.uint64(2)) % np.uint64(max(n, 1)), fixed))
        # full-width field elements, some non-canonical words (>= p) that must compare as their canonical value
        big = rng.integers(0, 2**63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
        tab = big.copy()
        rng.shuffle(tab)
        noncanon = big.copy()
        noncanon[: n // 2] = np.uint64(P) + (noncanon[: n // 2] % np.uint64(2**32 - 1))
        out.append(("wide n=%d" % n, big[rng.integers(0, n, n)], tab))
        out.append(("noncanonical n=%d" % n, noncanon, (noncanon[::-1] % np.uint64(P)).copy()))
    return out
