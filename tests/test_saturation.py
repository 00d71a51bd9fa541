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


def saturation_case(Backend, n_reads=10500):
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
    ref_s = synth.bases_to_str(ref)
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
