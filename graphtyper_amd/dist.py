"""Multi-GPU plumbing: reads shard across ranks (no data-path collective), the per-(haplotype,sample) integer score
vectors are summed with one all-reduce per tensor (RCCL over xGMI on GPUs, gloo in the CPU tests).

The reference's only parallelism is threads owning disjoint BAM pools whose results meet in per-pool files
(src/typer/caller.cpp:253-482); because every per-read effect is an integer addition a sum over ranks
followed by the saturation rules of gtx_scores_finalize gives the same numbers as one sequential pass."""
import numpy as np


def shard_bounds(n_items, world, rank):
    """contiguous shard [lo, hi) of n_items for `rank`; sizes differ by at most one"""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_pairs_by_name(name_ids, world):
    """rank of every record such that both mates of a pair (equal name id) land on the same rank and a rank gets a
    contiguous block of the position-sorted stream where possible: a record goes where the first record with its name
    went."""
    name_ids = np.asarray(name_ids)
    n = len(name_ids)
    first_seen = {}
    owner = np.empty(n, np.int32)
    for i, nm in enumerate(name_ids.tolist()):
        j = first_seen.setdefault(nm, i)
        owner[i] = min(world - 1, j * world // max(n, 1))
    return owner


def reduce_scores(dist, tensors, group=None):
    """in-place sum over ranks of the accumulator tensors (int32 / int64 views of the u32 / u64 counters)"""
    for t in tensors:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return tensors
