"""N>1 path on CPU: two gloo ranks each score their shard of the items (through the host emulation of the score
kernel), all-reduce the accumulators and must end with exactly the single-process result."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import harness
import scenarios
from graphtyper_amd import lib as gtx
from graphtyper_amd.dist import reduce_scores, shard_bounds, shard_pairs_by_name


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 100, 101):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_mates_stay_together():
    names = np.array([5, 9, 5, 7, 9, 7, 1, 1])
    owner = shard_pairs_by_name(names, 2)
    for nm in set(names.tolist()):
        assert len(set(owner[names == nm].tolist())) == 1


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    ref, recs, codes, rec = scenarios.paired_case("snp25", n_ref=8000, n_pairs=30, region_begin=0, n_samples=2)
    b = harness.EmuBackend(gtx.graph_from_records(ref, recs))
    st = gtx.Stream(b.ctx.params, 1)
    a_seq, a_meta, items = st.push(rec, gtx.pack_nibbles(codes))
    records = np.load(os.path.join(tmp, "records.npy"))
    lo, hi = shard_bounds(len(items), world, rank)
    acc = b.score(items[lo:hi], records, 2)
    tensors = [torch.from_numpy(a.view(np.int64 if a.dtype == np.uint64 else np.int32)) for a in
               (acc.log_score, acc.gt_cov, acc.hap_u32, acc.stat_u64, acc.stat_u32, acc.conn_near)]
    reduce_scores(dist, tensors)
    # the far-pair connection log is not reduced: every rank keeps its entries
    np.save(os.path.join(tmp, "log_%d.npy" % rank), acc.conn_log[:6 * int(acc.conn_count[0])])
    if rank == 0:
        np.save(os.path.join(tmp, "reduced.npy"), np.concatenate([t.numpy().astype(np.int64) for t in tensors]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_reduce_equals_single_pass(tmp_path):
    gtx.build()
    ref, recs, codes, rec = scenarios.paired_case("snp25", n_ref=8000, n_pairs=30, region_begin=0, n_samples=2)
    b = harness.EmuBackend(gtx.graph_from_records(ref, recs))
    st = gtx.Stream(b.ctx.params, 1)
    a_seq, a_meta, items = st.push(rec, gtx.pack_nibbles(codes))
    records = b.align(a_seq, a_meta)
    np.save(tmp_path / "records.npy", records)
    acc = b.score(items, records, 2)
    parts = (acc.log_score, acc.gt_cov, acc.hap_u32, acc.stat_u64, acc.stat_u32, acc.conn_near)
    single = np.concatenate([a.astype(np.int64) for a in parts])
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    reduced = np.load(tmp_path / "reduced.npy")
    assert np.array_equal(single, reduced)
    assert single.sum() > 0 and acc.conn_near.sum() > 0
    # reduced counters + the ranks' logs side by side = the single-process state (connections included)
    both = harness.Accumulators(b.ctx, 2)
    at = 0
    for dst in (both.log_score, both.gt_cov, both.hap_u32, both.stat_u64, both.stat_u32, both.conn_near):
        dst[...] = reduced[at:at + len(dst)].astype(dst.dtype)
        at += len(dst)
    logs = np.concatenate([np.load(tmp_path / ("log_%d.npy" % r)) for r in range(2)])
    both.conn_log[:len(logs)] = logs
    both.conn_count[0] = len(logs) // 6
    assert int(acc.conn_count[0]) == len(logs) // 6 > 0
    assert np.array_equal(harness.canonical_scores(b.ctx, acc), harness.canonical_scores(b.ctx, both))


def _replay_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_saturation import saturation_inputs
    ref_s, recs, rb, codes, rec = saturation_inputs()
    b = harness.EmuBackend(gtx.graph_from_records(ref_s, recs, region_begin=rb))
    st = gtx.Stream(b.ctx.params, 1)
    a_seq, a_meta, items = st.push(rec, gtx.pack_nibbles(codes))
    records = np.load(os.path.join(tmp, "records.npy"))
    lo, hi = shard_bounds(len(items), world, rank)
    acc = b.score(items[lo:hi], records, 2)
    alone = int(acc.hap_u32.reshape(2, b.ctx.n_hap, 4)[0, 1, 0])
    tensors = [torch.from_numpy(a.view(np.int64 if a.dtype == np.uint64 else np.int32)) for a in
               (acc.log_score, acc.gt_cov, acc.hap_u32, acc.stat_u64, acc.stat_u32, acc.conn_near)]
    reduce_scores(dist, tensors)  # (in place: acc now holds the sums, like every rank's block behind gtx_scores_reduce)
    assert alone < 0xFFFF < int(acc.hap_u32.reshape(2, b.ctx.n_hap, 4)[0, 1, 0])
    mine = b.score_replay_log(items[lo:hi], records, acc, item_base=lo)  # gtx_scores_replay_log
    logs = [None] * world
    dist.all_gather_object(logs, mine.tobytes())  # (the entries are plain data)
    entries = np.concatenate([np.frombuffer(x, gtx.REPLAY_ENTRY) for x in logs])
    assert b.score_replay_apply(acc, entries) == 1  # gtx_scores_replay_apply, on every rank: the same block everywhere
    np.save(os.path.join(tmp, "replayed_%d.npy" % rank), harness.canonical_scores(b.ctx, acc))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_saturation_replay(tmp_path):
    """reads sharded over two gloo ranks drive one (haplotype, sample) past the guard of explain_to_score only in the SUM: the ranks
    log their calls on that cell against the summed block, exchange the logs and replay them -- every rank ends with the oracle's
    sequential result over all reads (the bit-identity caveat of the exchange step)"""
    gtx.build()
    from test_saturation import saturation_inputs, oracle_scores
    ref_s, recs, rb, codes, rec = saturation_inputs()
    want, _ = oracle_scores(ref_s, recs, rb, codes, rec)
    b = harness.EmuBackend(gtx.graph_from_records(ref_s, recs, region_begin=rb))
    st = gtx.Stream(b.ctx.params, 1)
    a_seq, a_meta, items = st.push(rec, gtx.pack_nibbles(codes))
    np.save(tmp_path / "records.npy", b.align(a_seq, a_meta))
    port = 31500 + os.getpid() % 2000
    mp.spawn(_replay_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        got = np.load(tmp_path / ("replayed_%d.npy" % r))
        assert len(got) == len(want) and np.array_equal(got, want), r
