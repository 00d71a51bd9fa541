"""The sequential saturation guard of Haplotype::explain_to_score (src/graph/haplotype.cpp:560): once a (haplotype, sample)
has max_log_score within epsilon of 0xFFFF no further read is added, and which reads are dropped depends on their order
(a read with a small epsilon may still go in after one with a large epsilon was refused).  gtx_score_batch adds without
the guard; gtx_scores_replay logs the explain_to_score calls of the cells at the guard, sorts them back into call order
and replays them on the host.  ~10 000 reads of one sample over one SNP (different alleles, error counts, MAPQ, clipped
reads: epsilons 4..8) against the oracle's sequential Genotyper."""
import numpy as np

import harness
import scenarios
from graphtyper_amd import lib as gtx
from graphtyper_amd import synth
from oracle_lib import Oracle


def saturation_inputs(n_reads=10500):
    """(reference, records, region begin, read codes, stream records): ~10 000 reads of sample 0 over the middle SNP"""
    rb = 70000
    rng = np.random.default_rng(5)
    ref = synth.make_reference(2400, seed=31)
    recs = synth.make_snp_records(ref, 700, seed=4, region_begin=rb)  # sites at ~350, 1050, 1750
    site = recs[1][0] - rb
    alt = "ACGT".index(recs[1][2][0])
    codes, pos = [], []
    for i in range(n_reads):
        start = int(rng.integers(site - 140, site - 9))
        r = ref[start:start + 150].copy()
        if rng.random() < 0.45:
            r[site - start] = alt
        for _ in range(int(rng.choice([0, 0, 0, 1, 2, 3]))):  # mismatches lower epsilon
            j = int(rng.integers(0, 150))
            if j != site - start:
                r[j] = (int(r[j]) + 1) % 4
        codes.append(synth._CODE_OF_BASE[r])
        pos.append(start + rb)
    order = np.argsort(np.array(pos), kind="stable")
    codes = np.ascontiguousarray(np.array(codes, np.uint8)[order])
    pos = np.array(pos, np.int64)[order]
    mapq = np.where(rng.random(n_reads) < 0.2, 10, 60)
    sample = (rng.random(n_reads) < 0.03).astype(np.int64)  # nearly everything in sample 0: only its cell saturates
    rec = scenarios.stream_records(n_reads, pos, mapq=mapq, sample=sample)
    return synth.bases_to_str(ref), recs, rb, codes, rec


def oracle_scores(ref_s, recs, rb, codes, rec):
    o = Oracle(ref_s, recs, region_begin=rb)
    og = o.genotyper(2, 1)
    og.push(list(codes), flags=rec["flag"], tid=rec["tid"], mtid=rec["mtid"], pos=rec["pos"], isize=rec["isize"], mapq=rec["mapq"],
            score_diff=rec["score_diff"], name=rec["name_id"], sample=rec["sample"], rg=rec["rg"])
    return og.scores(), og.calls()


def saturation_case(Backend, n_reads=10500):
    ref_s, recs, rb, codes, rec = saturation_inputs(n_reads)
    o = Oracle(ref_s, recs, region_begin=rb)
    b = Backend(gtx.graph_from_records(ref_s, recs, region_begin=rb))
    og = o.genotyper(2, 1)
    og.push(list(codes), flags=rec["flag"], tid=rec["tid"], mtid=rec["mtid"], pos=rec["pos"], isize=rec["isize"], mapq=rec["mapq"],
            score_diff=rec["score_diff"], name=rec["name_id"], sample=rec["sample"], rg=rec["rg"])
    want = og.scores()
    st = gtx.Stream(b.ctx.params, 1)
    a_seq, a_meta, items = st.push(rec, gtx.pack_nibbles(codes))
    records = b.align(a_seq, a_meta)
    acc = b.score(items, records, 2)
    cells = acc.hap_u32.reshape(2, b.ctx.n_hap, 4)
    assert cells[0, 1, 0] > 0xFFFF and cells[1].max() < 0xFFFF - 8, "the scenario must drive exactly one sample over the guard"
    unordered = acc.log_score.copy()
    assert b.score_replay(items, records, acc) == 1
    assert b.score_replay(items, records, acc) == 0  # (a replayed cell is marked and left alone)
    got = harness.canonical_scores(b.ctx, acc)  # (finalize accepts the replayed cell)
    assert len(got) == len(want)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, "score streams differ at words %s" % bad[:10]
    assert not np.array_equal(unordered, acc.log_score), "the replay changed nothing"
    cells = acc.hap_u32.reshape(2, b.ctx.n_hap, 4)
    assert 0xFFFF - 8 <= cells[0, 1, 0] < 0xFFFF
    # the calls are made from the replayed rows
    phred, calls = b.calls(acc, 2)
    assert np.array_equal(harness.canonical_calls(b.ctx, phred, calls, 2), og.calls())


def test_saturation_guard_is_replayed_in_call_order():
    saturation_case(harness.EmuBackend)


def two_rank_replay_case(Backend):
    """the reads sharded over two "ranks" of one process: each scores its half into a block of its own, the blocks are added (what
    gtx_scores_reduce leaves on every rank), each rank logs what ITS items did to the cells at the guard of the sum
    (gtx_scores_replay_log, item_base = where its items stand in the stream), the logs side by side are replayed
    (gtx_scores_replay_apply): the oracle's sequential result over all reads"""
    ref_s, recs, rb, codes, rec = saturation_inputs()
    want, want_calls = oracle_scores(ref_s, recs, rb, codes, rec)
    b = Backend(gtx.graph_from_records(ref_s, recs, region_begin=rb))
    st = gtx.Stream(b.ctx.params, 1)
    a_seq, a_meta, items = st.push(rec, gtx.pack_nibbles(codes))
    records = b.align(a_seq, a_meta)
    cut = len(items) * 2 // 5
    halves = [(0, items[:cut]), (cut, items[cut:])]
    accs = [b.score(it, records, 2) for _, it in halves]
    for half in accs:  # (neither rank alone is over the guard by much, or at all: only the sum says which cells to replay)
        assert half.hap_u32.reshape(2, b.ctx.n_hap, 4)[0, 1, 0] < 0xFFFF
    total = harness.Accumulators(b.ctx, 2)
    for dst, x, y in zip(total.arrays(), accs[0].arrays(), accs[1].arrays()):
        dst[...] = x + y
    assert total.hap_u32.reshape(2, b.ctx.n_hap, 4)[0, 1, 0] > 0xFFFF
    logs = [b.score_replay_log(it, records, total, item_base=base) for base, it in halves]
    assert all(len(l) > 1000 for l in logs) and int(logs[1]["item"].min()) >= cut > int(logs[0]["item"].max())
    entries = np.concatenate(logs[::-1])  # (any order: apply sorts them into call order)
    assert b.score_replay_apply(total, entries) == 1
    got = harness.canonical_scores(b.ctx, total)
    assert len(got) == len(want) and np.array_equal(got, want)
    phred, calls = b.calls(total, 2)
    assert np.array_equal(harness.canonical_calls(b.ctx, phred, calls, 2), want_calls)
    # ... and it matters which order: the second rank's reads first is another result
    swapped = harness.Accumulators(b.ctx, 2)
    for dst, x, y in zip(swapped.arrays(), accs[0].arrays(), accs[1].arrays()):
        dst[...] = x + y
    wrong = [b.score_replay_log(it, records, swapped, item_base=base) for base, it in ((len(items) - cut, halves[0][1]), (0, halves[1][1]))]
    b.score_replay_apply(swapped, np.concatenate(wrong))
    assert not np.array_equal(swapped.log_score, total.log_score)


def test_two_rank_replay_on_the_emulation():
    two_rank_replay_case(harness.EmuBackend)
