"""Rank-level plumbing for the multi-GPU runs (one process per GPU, torch.distributed; backend "nccl" = RCCL on the GPU
box, "gloo" in the CPU tests).  The hot path shards by independent units (columns for the standalone NTT, LDE cosets for
a full proof -- SURVEY F9).  The NTT bench has no data-path collective, only the barrier/max-reduce used for timing; a
commitment all-gathers its Merkle cap slices (16 x 4 u64 per tree) and a full proof on the coset partition additionally
exchanges quotient values and opened query rows through Backend.set_shard's callback (olavm_amd/backend.py)."""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous block partition of `total` units; the first `total % world` ranks get one extra."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def coset_range(rank, world, rate_bits=3):
    """Leaf-order cosets (blocks of n commitment leaves) owned by `rank`: a contiguous run, so that the rank's Merkle
    sub-trees are a contiguous slice of the cap (ola_commit_values_shard)."""
    per = (1 << rate_bits) // world
    assert per * world == (1 << rate_bits), "world must divide the number of cosets"
    return rank * per, (rank + 1) * per


def coset_owner(coset, world, rate_bits=3):
    """LDE coset (leaf-order block) -> owning rank for world in {1,2,4,8}."""
    return coset * world >> rate_bits


def commit_sharded(be, rank, world, cols=None, dev_ptr=None, ncols=None, log_n=None, device="cpu"):
    """PolynomialBatch::from_values across `world` GPUs: this rank commits its cosets (Backend.commit_shard), the cap
    slices are all-gathered (the only collective of the commitment: 512 bytes in total) and concatenated in rank order.
    -> (local Batch, full cap as a (16, 4) uint64 array, identical on every rank and to the single-GPU commitment)."""
    import numpy as np
    b = be.commit_shard(cols, rank, world, dev_ptr=dev_ptr, ncols=ncols, log_n=log_n)
    parts = all_gather_caps(b.cap().view(np.int64), device=device)
    cap = np.concatenate([p.cpu().numpy().view(np.uint64) for p in parts])
    return b, cap


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
