"""Pools of samples put together (the merge half of vcf_merge_and_break, /root/reference/src/typer/vcf_operations.cpp:480-620: the
calls of the pools' samples side by side, `var.stats.add_stats(next_vcf_var.stats)`): two pools genotyped on their own -- each its
own record stream, alignment, scoring, calls -- give, with their per-sample arrays concatenated and their site statistics added, the
VCF text of all samples genotyped at once, which is the oracle's.  (Every accumulator is a sum over reads: that is also why
gtx_scores_reduce can add the blocks of ranks that shard the READS; this is the other split, by SAMPLES.)"""
import numpy as np
import pytest

import harness
import scenarios
from graphtyper_amd import lib as gtx
from oracle_lib import Oracle


def genotype(backend, codes, rec, n_samples):
    st = gtx.Stream(backend.ctx.params, 1)
    a_seq, a_meta, items = st.push(rec, gtx.pack_nibbles(codes))
    records = backend.align(a_seq, a_meta)
    acc = backend.score(items, records, n_samples)
    phred, calls = backend.calls(acc, n_samples)
    return acc, phred, calls


@pytest.mark.parametrize("kind", ["snp25", "indel"])
def test_two_pools_of_samples_merge_into_the_text_of_one_run(kind):
    rb = 310000
    ref, recs, codes, rec = scenarios.paired_case(kind, n_ref=30000, n_pairs=1600, region_begin=rb, n_samples=4)
    oracle = Oracle(ref, recs, region_begin=rb)
    og = oracle.genotyper(4, 1)
    og.push(list(codes), flags=rec["flag"], tid=rec["tid"], mtid=rec["mtid"], pos=rec["pos"], isize=rec["isize"], mapq=rec["mapq"],
            score_diff=rec["score_diff"], name=rec["name_id"], sample=rec["sample"], rg=rec["rg"])
    names = ["SAMP%02d" % i for i in range(4)]
    want = og.vcf_records("chrT", names)
    backend = harness.EmuBackend(gtx.graph_from_records(ref, recs, region_begin=rb))
    ctx = backend.ctx
    # all four samples at once
    acc, phred, calls = genotype(backend, codes, rec, 4)
    assert ctx.vcf_records("chrT", names, acc.gt_cov, acc.stat_u64, acc.stat_u32, phred, calls) == want
    # two pools: samples 0-1 and 2-3, each numbered from 0 inside its pool
    parts = []
    for pool in ((0, 1), (2, 3)):
        mine = np.isin(rec["sample"], pool)
        r = rec[mine].copy()
        r["sample"] -= pool[0]
        parts.append(genotype(backend, codes[mine], r, 2))
    (a0, p0, c0), (a1, p1, c1) = parts
    merged = ctx.vcf_records("chrT", names, np.concatenate([a0.gt_cov, a1.gt_cov]), a0.stat_u64 + a1.stat_u64, a0.stat_u32 + a1.stat_u32,
                             np.concatenate([p0, p1]), np.concatenate([c0, c1]))
    assert merged == want and merged.count(b"\n") == ctx.n_hap + 1 and b"\t0/1:" in merged
    # not vacuous: a pool alone says something else
    assert ctx.vcf_records("chrT", names[:2], a0.gt_cov, a0.stat_u64, a0.stat_u32, p0, c0) != ctx.vcf_records("chrT", names[:2], a1.gt_cov, a1.stat_u64, a1.stat_u32, p1, c1)
