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
    went.

    What sharding is exact for: the stateful parts of the record stream -- duplicate reuse (a record equal in position
    and sequence to its predecessor takes over the predecessor's alignment, hts_parallel_reader.cpp:666-684) and the
    coverage filter of SV calling (per-sample bin counts, :594-633) -- depend on which records are neighbours.  Run
    gtx_stream_push over the WHOLE stream (on every rank, or once) and shard its OUTPUT, the alignment tasks and score
    items, as tests/test_dist_gloo.py and bench.py do: that gives the single-process accumulators exactly.  Sharding the
    records before the stream is exact only without duplicates at shard boundaries and without the coverage filter."""
    name_ids = np.asarray(name_ids)
    n = len(name_ids)
    _, first_index, inverse = np.unique(name_ids, return_index=True, return_inverse=True)
    first = first_index[inverse]  # position of the first record with this name
    return np.minimum(world - 1, first * world // max(n, 1)).astype(np.int32)


def reduce_scores(dist, tensors, group=None):
    """in-place sum over ranks of the accumulator tensors (int32 / int64 views of the u32 / u64 counters)"""
    for t in tensors:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return tensors
