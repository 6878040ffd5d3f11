"""Deterministic full-size test inputs generated on the device (torch is plumbing here: device memory only).

splitmix_columns: a splitmix64 stream per column, reduced to canonical form -- the inputs SURVEY 8(d) config 2 names.
adversarial_columns: a tiling of the values where Goldilocks reductions change branch
({0, 1, p-1, 2^32-1, 2^32, 0xFFFFFFFF00000000}).
"""
P = 0xFFFFFFFF00000001
MASK64 = (1 << 64) - 1


def s64(x):
    x &= MASK64
    return x - (1 << 64) if x >> 63 else x


def splitmix_columns(torch, cols, n, first_col=0, device="cuda"):
    """(cols, n) int64 tensor holding canonical field elements: element i of column c is splitmix64 output i of the
    stream seeded with 0x01A5EED + c, minus p when it is >= p."""
    i = torch.arange(1, n + 1, dtype=torch.int64, device=device)
    seed = torch.arange(first_col, first_col + cols, dtype=torch.int64, device=device) + 0x01A5EED
    z = seed[:, None] + i[None, :] * s64(0x9E3779B97F4A7C15)

    def lsr(v, s):
        return (v >> s) & ((1 << (64 - s)) - 1)
    z = (z ^ lsr(z, 30)) * s64(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * s64(0x94D049BB133111EB)
    z = z ^ lsr(z, 31)
    # unsigned z >= p  <=>  signed z in [p - 2^64, -1]
    return torch.where((z < 0) & (z >= s64(P)), z - s64(P), z)


def adversarial_columns(torch, cols, n, device="cuda"):
    vals = [0, 1, P - 1, 2**32 - 1, 2**32, 0xFFFFFFFF00000000]
    base = torch.tensor([s64(v) for v in vals], dtype=torch.int64, device=device)
    idx = (torch.arange(n, device=device)[None, :] + torch.arange(cols, device=device)[:, None] * 5) % len(vals)
    return base[idx]


def canonical(torch, x):
    return torch.where((x < 0) & (x >= s64(P)), x - s64(P), x)
