"""Genotyper::merge_from (oracle/gto.hpp; test infrastructure): a read set pushed through several oracle Genotypers in
contiguous shards and summed must be the read set pushed through one -- scores, SampleCalls, VCF text.  This is what lets
tests/test_gpu_full_size.py put all 10 M reads of BASELINE cfg2 through the oracle on the host's cores."""
import numpy as np
import pytest

import scenarios
from oracle_lib import Oracle, pack_reads, sharded_genotyper


@pytest.mark.parametrize("kind,threads", [("snp100", 3), ("snp25", 5), ("indel", 2)])
def test_sharded_oracle_equals_one_pass(kind, threads):
    ref, recs, codes, pos = scenarios.synthetic_case(kind, n_ref=30000, n_reads=3000, region_begin=7000)
    order = np.argsort(pos, kind="stable")
    codes, pos = codes[order], np.asarray(pos)[order]
    samples = (np.arange(len(codes)) % 2).astype(np.int32)
    o = Oracle(ref, recs, region_begin=7000)
    one = o.genotyper(2, 1)
    one.push(None, pos=np.ascontiguousarray(pos, np.int64), packed=pack_reads(list(codes)), sample=samples)
    many, used = sharded_genotyper(o, codes, pos, n_samples=2, samples=samples, threads=threads)
    assert used == 1  # (3 000 reads: the helper does not shard below 10 000 reads per thread) ...
    # ... so cut by hand as well
    cuts = [len(codes) * k // threads for k in range(threads + 1)]
    parts = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        g = o.genotyper(2, 1)
        g.push(None, pos=np.ascontiguousarray(pos[a:b], np.int64), packed=pack_reads(list(codes[a:b])), sample=samples[a:b])
        parts.append(g)
    for g in parts[1:]:
        parts[0].merge(g)
    for got in (many, parts[0]):
        assert np.array_equal(got.scores(), one.scores())
        assert np.array_equal(got.calls(), one.calls())
        assert got.vcf_records("chrT", ["A", "B"]) == one.vcf_records("chrT", ["A", "B"])
        assert got.counts()["records"] == one.counts()["records"]
    assert one.scores().sum() > 0 and (one.calls() > 0).any()


def test_merge_refuses_the_saturation_guard():
    """a cell whose summed max_log_score comes within 8 of 0xFFFF needs the reference's sequential order: refused"""
    ref, recs, codes, pos = scenarios.synthetic_case("snp100", n_ref=2000, n_reads=400, region_begin=0, err=0.0, n_rate=0.0)
    o = Oracle(ref, recs)
    reads = list(codes)
    a, b = o.genotyper(1, 1), o.genotyper(1, 1)
    # an error-free read over a site adds 8 to the cell's max_log_score; 400 reads of 150 bp over 2 kb put ~30 on a site per
    # push, the guard stands at 0xFFFF - 8: 150 pushes bring each shard past half of it
    for g in (a, b):
        for _ in range(150):
            g.push(None, pos=np.zeros(len(reads), np.int64), packed=pack_reads(reads))
    with pytest.raises(RuntimeError, match="saturation guard"):
        a.merge(b)
