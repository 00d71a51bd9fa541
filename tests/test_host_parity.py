"""Host-side product code (graph flattening, special positions, backward-enumeration index builder) against the
CPU oracle.  No GPU needed: contexts are created with device=-1 (inspection only; no compute entry point works)."""
import numpy as np
import pytest

import scenarios
from fixtures import contig
from graphtyper_amd import lib as gtx
from graphtyper_amd import synth
from oracle_lib import Oracle


@pytest.fixture(scope="module", autouse=True)
def _built():
    gtx.build()


def _compare(reference, records, region_begin=0):
    o = Oracle(reference, records, region_begin=region_begin)
    g = gtx.graph_from_records(reference, records, region_begin=region_begin)
    og = o.graph()
    # node tables
    for k in ("ref_order", "ref_len", "ref_nvar", "var_order", "var_len", "var_out_ref"):
        assert np.array_equal(g[k], og[k]), k
    inner = og["ref_nvar"] > 0
    assert np.array_equal(g["ref_first_var"][inner], og["ref_first_var"][inner])
    assert g["dna"].tobytes() == og["dna"].tobytes()
    c = gtx.Context(g, device=-1)
    rr, ap = c.special_positions()
    assert np.array_equal(rr, og["ref_reach_poses"]) and np.array_equal(ap, og["actual_poses"])
    # index: same keys, same label count per key, same labels in the same bucket order
    k1, c1, l1 = o.index_dump()
    k2, c2, l2 = c.index_dump()
    assert np.array_equal(k1, k2)
    assert np.array_equal(c1, c2)
    assert np.array_equal(l1, l2)
    return o, c


@pytest.mark.parametrize("chrom", ["chr1", "chr2", "chr3", "chr4", "chr9", "chr10"])
def test_index_test_contigs(chrom):
    ref, recs = contig(chrom)
    _compare(ref, recs)


def test_events_prune_walks():
    # chr9: the alt at 5 carries anti event 2, the insertion at 10 carries event 2 (test/index/test_index.cpp:352-398)
    ref, recs = contig("chr9")
    _, c = _compare(ref, recs)
    from oracle_lib import to_uint64
    assert c.index_get(to_uint64("AGGGGAGTGGGGGGGGGGGGGGGGGGGGGGGG")) == []
    assert len(c.index_get(to_uint64("G" * 32))) == 36


def test_synthetic_snp_graph():
    ref = synth.make_reference(120000, seed=5)
    recs = synth.make_snp_records(ref, every=100, seed=9, region_begin=1000000)
    _compare(synth.bases_to_str(ref), recs, region_begin=1000000)


def test_synthetic_indel_graph():
    ref = synth.make_reference(60000, seed=6)
    recs = synth.make_indel_records(ref, every=37, seed=3, region_begin=5000)
    _compare(synth.bases_to_str(ref), recs, region_begin=5000)


def test_dense_adjacent_sites_hit_the_pruning_rules():
    # SNPs every 3 bp: walks cross up to 11 sites, so entry_has_too_many_nonrefs (indexer.cpp:13-20) decides a lot
    ref = synth.make_reference(900, seed=8)
    recs = synth.make_snp_records(ref, every=3, seed=2, first=40)
    _compare(synth.bases_to_str(ref), recs)
    # multi-allelic sites with different allele lengths right next to each other (special positions in labels)
    s = synth.bases_to_str(ref)
    recs = []
    p = 50
    rng = np.random.default_rng(4)
    while p < 800:
        r = s[p]
        alts = sorted({r + "ACGT"[rng.integers(4)], "ACGT"[("ACGT".index(r) + 1) % 4], r + "GG" + "ACGT"[rng.integers(4)]})
        recs.append((p, r, alts, None))
        p += int(rng.integers(1, 9))
    _compare(s, recs)


def test_reference_with_n_runs():
    ref, recs = contig("chr4")
    s = "ACGTTGCA" * 20 + "N" * 7 + ref + "NN" + "TTGACCA" * 15
    recs = [(p + 167, r, a, i) for p, r, a, i in recs]
    _compare(s, recs)


def test_unsupported_graph_is_refused_loudly():
    """70 alleles are fine (tests/test_wide_sites.py aligns and scores over such sites); a view that claims more alleles
    than the reference's MAX_NUMBER_OF_HAPLOTYPES is outside the envelope and refused by name"""
    ref = "ACGT" * 100
    alts = sorted({"A" + "C" * k for k in range(1, 70)})
    g = gtx.graph_from_records(ref, [(40, "A", alts, None)])
    c = gtx.Context(g, device=-1)
    assert int(c.hap_cnum.max()) == 70
    # (gtx_graph_build caps a record at 2558 alternative alleles like graph.cpp:260-263, so the view is made by hand)
    n = 2600
    dna = ref[:40] + "".join("ACGT"[k % 4] for k in range(n)) + ref[41:]
    view = dict(ref_order=np.array([1, 42], np.uint32), ref_len=np.array([40, len(ref) - 41], np.uint32),
                ref_dna_off=np.array([0, 40 + n], np.uint32), ref_nvar=np.array([n, 0], np.uint32), ref_first_var=np.array([0, n], np.uint32),
                var_order=np.full(n, 41, np.uint32), var_len=np.ones(n, np.uint32), var_dna_off=np.arange(40, 40 + n, dtype=np.uint32),
                var_out_ref=np.ones(n, np.uint32), dna=np.frombuffer(dna.encode(), np.uint8))
    with pytest.raises(gtx.GtxError) as e:
        gtx.Context(view, device=-1)
    assert "MAX_NUMBER_OF_HAPLOTYPES" in str(e.value) and e.value.status == 4


def test_no_cpu_path():
    ref, recs = contig("chr1")
    c = gtx.Context(gtx.graph_from_records(ref, recs), device=-1)
    dummy = np.zeros(64, np.uint8)
    rc = gtx.lib().gtx_align_batch(c.h, gtx._p(dummy), 16, gtx._p(dummy), 1, gtx._p(dummy), 64, None)
    assert rc == 2  # GTX_ERR_NO_DEVICE
    # the library's own host loop has no other place to compute either
    import ctypes as C
    paths = (C.c_char_p * 1)(b"/nonexistent.bam")
    buf, st = gtx.ScoreBuffers(), gtx.PipelineStats()
    buf.n_samples = 1
    assert gtx.lib().gtx_pipeline_run(c.h, paths, 1, 1, None, 1024, 64, 1000, C.byref(buf), C.byref(st)) == 2
    assert gtx.lib().gtx_pipeline_run(c.h, paths, 0, 1, None, 1024, 64, 1000, C.byref(buf), C.byref(st)) != 0  # (no files: a bad argument)


def test_near_pair_layout():
    """gtx_ctx_near_pairs: windows of the dense connection counters = the haplotypes whose order is < 100 above
    (hts_parallel_reader.cpp:800-801), offsets = running sum of cnum(h) * alleles of the window"""
    for kind, aav in (("snp25", False), ("snp100", False), ("cluster", True)):
        ref, recs, _, _ = scenarios.synthetic_case(kind, n_ref=20000, n_reads=1, region_begin=0)
        ctx = gtx.Context(gtx.graph_from_records(ref, recs, add_all_variants=aav), device=-1)
        order, cnum = ctx.hap_order.astype(np.int64), ctx.hap_cnum.astype(np.int64)
        total = 0
        for h in range(ctx.n_hap):
            last = h
            while last + 1 < ctx.n_hap and order[last + 1] < order[h] + 100:
                last += 1
            assert int(ctx.near_last[h]) == last and int(ctx.near_off[h]) == total
            total += int(cnum[h] * cnum[h + 1:last + 1].sum())
        assert total == ctx.total_near
        assert (ctx.total_near == 0) == (kind == "snp100")  # (sites exactly 100 apart are not near)
