"""Rank-level plumbing for the multi-GPU runs (one process per GPU, torch.distributed; backend "nccl" = RCCL on the GPU
box, "gloo" in the CPU tests).  The hot path shards by independent units (columns for the standalone NTT, LDE cosets for
a full proof -- SURVEY F9); there is no data-path collective, only the barrier/max-reduce used for timing and, for a
proof, the all-gather of Merkle caps (16 x 4 u64 per tree)."""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous block partition of `total` units; the first `total % world` ranks get one extra."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def coset_owner(coset, world):
    """LDE coset -> rank for world in {1,2,4,8} with rate_bits = 3 (8 cosets): coset c lives on rank c mod world."""
    return coset % world


def max_over_ranks(values, device="cpu"):
    """Element-wise max of a list of floats across ranks (timing: the slowest rank defines the step)."""
    t = torch.tensor(values, dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t]


def all_gather_caps(cap, device="cpu"):
    """All-gather of per-rank Merkle cap slices (int64 view of canonical u64 digests); returns rank-ordered list."""
    t = torch.as_tensor(cap, dtype=torch.int64, device=device).contiguous()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [t]
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return out


def aggregate_throughput(units_per_rank, bytes_per_unit, world, steps, elapsed_s):
    """Whole-job GB/s for a weak-scaling run: every rank processed `units_per_rank` units per step."""
    return units_per_rank * bytes_per_unit * world * steps / elapsed_s / 1e9
