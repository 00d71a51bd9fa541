"""Pins the CPU oracle against the reference's own known-answer tests (cited per test).
Runs without a GPU."""
import ctypes as C

import numpy as np
import pytest

from fixtures import contig
from oracle_lib import INVALID_ID, SPECIAL_START, Oracle, encode, lib, to_dna_str, to_uint64, _p


# ---------------------------------------------------------------- test/utilities/test_utilities.cpp
def test_converting_reads():  # :17-31
    read = "TTTCCCCAGGTTTCCCCAGGTTTCCCCAGGTTTGCCCAGGTTTCCCCAGGTTTCCCCTTTGGA"
    k1, k2 = "TTTCCCCAGGTTTCCCCAGGTTTCCCCAGGTT", "TTGCCCAGGTTTCCCCAGGTTTCCCCTTTGGA"
    out = np.zeros(8, np.uint64)
    codes = encode(read)
    assert lib().gto_to_uint64_vec(_p(codes), C.c_long(len(codes)), C.c_long(0), _p(out), C.c_long(8)) == 1
    assert int(out[0]) == to_uint64(k1) and to_dna_str(int(out[0])) == k1
    assert lib().gto_to_uint64_vec(_p(codes), C.c_long(len(codes)), C.c_long(31), _p(out), C.c_long(8)) == 1
    assert int(out[0]) == to_uint64(k2) and to_dna_str(int(out[0])) == k2


@pytest.mark.parametrize("kmer,expect", [  # :33-83
    ("ATTCCCCAGGTTTCCCCAGGTTTCCCCAGGTA", "CGT"), ("TTTCCCCAGGTTTCCCCAGGTTTCCCCAGGTC", "AGT"),
    ("CTTCCCCAGGTTTCCCCAGGTTTCCCCAGGTG", "ACT"), ("GATCCCCAGGTTTCCCCAGGTTTCCCCAGGTT", "ACG")])
def test_mismatches_of_last_base(kmer, expect):
    out = np.zeros(3, np.uint64)
    lib().gto_mismatches_of_last_base(C.c_uint64(to_uint64(kmer)), _p(out))
    assert [to_dna_str(int(x)) for x in out] == [kmer[:-1] + c for c in expect]


@pytest.mark.parametrize("kmer,expect", [  # :85-131
    ("ATTCCCCAGGTTTCCCCAGGTTTCCCCAGGTA", "CGT"), ("CTTCCCCAGGTTTCCCCAGGTTTCCCCAGGTC", "AGT"),
    ("GTTCCCCAGGTTTCCCCAGGTTTCCCCAGGTG", "ACT"), ("TTTCCCCAGGTTTCCCCAGGTTTCCCCAGGTA", "ACG")])
def test_mismatches_of_first_base(kmer, expect):
    out = np.zeros(3, np.uint64)
    lib().gto_mismatches_of_first_base(C.c_uint64(to_uint64(kmer)), _p(out))
    assert [to_dna_str(int(x)) for x in out] == [c + kmer[1:] for c in expect]


def test_hamming_distance_1():  # :133-162
    a = "A" * 32
    out = np.zeros(96, np.uint64)
    lib().gto_hamming1(C.c_uint64(to_uint64(a)), _p(out))
    got = {to_dna_str(int(x)) for x in out}
    assert a not in got and len(got) == 96
    for pos, c in [(31, "C"), (28, "G"), (25, "T"), (17, "T"), (16, "C"), (16, "G"), (16, "T"), (15, "T"), (0, "T"),
                   (4, "G"), (11, "C")]:
        assert a[:pos] + c + a[pos + 1:] in got
    # order contract used by query_index_hamming_distance1_without_index (type_conversions.cpp:272-288)
    key = to_uint64("ACGT" * 8)
    lib().gto_hamming1(C.c_uint64(key), _p(out))
    for bb in range(32):
        for m in (1, 2, 3):
            assert int(out[bb * 3 + m - 1]) == key ^ (m << (2 * bb))


# ---------------------------------------------------------------- test/utilities/test_kmer_help_functions.cpp
def test_get_num_kmers():  # :20-47
    L = lib()
    assert [L.gto_get_num_kmers(n) for n in (32, 62, 63, 64, 93, 94, 95)] == [1, 1, 2, 2, 2, 3, 3]
    assert L.gto_get_num_kmers(31) == 0 and L.gto_get_num_kmers(150) == 4 and L.gto_get_num_kmers(151) == 4


def test_get_ith_kmer():  # :49-71 (offset of the centred i-th k-mer)
    L = lib()
    assert L.gto_ith_kmer_offset(32, 0) == 0 and L.gto_ith_kmer_offset(33, 0) == 0 and L.gto_ith_kmer_offset(34, 0) == 1
    s62 = "AAAACAAAAGAAACCAAAAGAAAACAAAAGATAAAACAAAAGAAAACAAAAGAAAACAAAAG"
    o = L.gto_ith_kmer_offset(len(s62), 0)
    assert s62[o:o + 32] == "AAAAGAAAACAAAAGATAAAACAAAAGAAAAC"
    s63 = s62 + "A"
    assert s63[L.gto_ith_kmer_offset(63, 0):][:32] == "AAAACAAAAGAAACCAAAAGAAAACAAAAGAT"
    assert s63[L.gto_ith_kmer_offset(63, 1):][:32] == "TAAAACAAAAGAAAACAAAAGAAAACAAAAGA"


def _keys(read, i):
    codes = encode(read)
    out = np.zeros(512, np.uint64)
    n = lib().gto_to_uint64_vec(_p(codes), C.c_long(len(codes)), C.c_long(i), _p(out), C.c_long(512))
    return [to_dna_str(int(x)) for x in out[:n]]


def test_iupac_reads():  # :73-116
    r = "ACCGGGGTTAAAATTGAAAACCCCTAAAATTGAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA"
    assert _keys(r, 0) == ["ACCGGGGTTAAAATTGAAAACCCCTAAAATTG"]
    assert _keys(r, 10) == ["AAATTGAAAACCCCTAAAATTGAAAAAAAAAA"]
    r = "ACCGGGGTTAAAATTGAAAACCCCTAAAATTNAAAAAAAAAAAAAAAAAAAAAAAAAWAAAAAAAAAATTTTTTTBTTTTTTTTTTTTTTTTTTT"
    p = "ACCGGGGTTAAAATTGAAAACCCCTAAAATT"
    assert _keys(r, 0) == [p + "T", p + "A", p + "C", p + "G"]
    assert _keys(r, 32) == ["AAAAAAAAAAAAAAAAAAAAAAAAATAAAAAA", "A" * 32]
    assert _keys(r, 63) == ["AAAAATTTTTTTTTTTTTTTTTTTTTTTTTTT", "AAAAATTTTTTTCTTTTTTTTTTTTTTTTTTT",
                            "AAAAATTTTTTTGTTTTTTTTTTTTTTTTTTT"]
    assert _keys("NNNNNNNNNNNNAAAAAAAAAAAAAAAAAAAAAA", 0) == []


# ---------------------------------------------------------------- test/typer/test_path.cpp:50-65
def test_two_reference_paths_merge():
    out = np.zeros(5, np.uint32)
    lib().gto_path_merge_two_ref_labels(1, 32, 0, 31, 32, 43, 31, 62, _p(out))
    assert list(out) == [63, 1, 43, 0, 0]


# ---------------------------------------------------------------- test/index/test_index.cpp
def _oracle(chrom):
    ref, recs = contig(chrom)
    return Oracle(ref, recs), ref


def test_index_chr1():  # :17-81
    o, ref = _oracle("chr1")
    assert o.all_ref() == "AGGTTTCCCCAGGTTTCCCCAGGTTTCCCCAGGTTTCCCCAGGTTTCCCCAGGTTTCCCCTTTGGA" == ref
    assert o.index_check()
    a, b = "AGGTTTCCCCAGGTTTCCCCAGGTTTCCCCAG", "AGGTTTCCCCAGGTTTCCCCAGGTTTCCCCTT"
    c, d = "TTCCCCAGGTTTCCCCAGGTTTCCCCTTTGGA", "GGTTTCCCCAGGTTTCCCCAGGTTTGCCCAGG"
    assert [len(o.index_get(k)) for k in (a, b, c, d)] == [3, 1, 1, 1]
    assert o.index_get(a) == [(1, 32, INVALID_ID), (11, 42, 0), (21, 52, 0)]
    assert o.index_get(b) == [(31, 62, 0)]
    assert o.index_get(d) == [(12, 43, 1)]


def test_index_chr1_lookups_of_the_retired_aligner_test():
    """test/typer/test_gyper_aligner.cpp:28-96 (a test of the reference's former aligner class, no longer built upstream): its
    four lookups on the chr1 graph still state what the index holds -- the repeated k-mer at three places, a unique
    reference k-mer, the k-mer over the G allele at the same place, and a k-mer that is nowhere.  (That class counted
    positions from 0; the index of today counts from 1, test/index/test_index.cpp:17-81 above.)"""
    o, ref = _oracle("chr1")
    common = o.index_get("TTTCCCCAGGTTTCCCCAGGTTTCCCCAGGTT")
    assert len(common) == 3
    assert sorted(s - 1 for s, _, _ in common) == [3, 13, 23] and sorted(e - 1 for _, e, _ in common) == [34, 44, 54]
    unique = o.index_get("TTCCCCAGGTTTCCCCAGGTTTCCCCTTTGGA")
    assert [(s - 1, e - 1) for s, e, _ in unique] == [(34, 65)]
    on_variant = o.index_get("TTGCCCAGGTTTCCCCAGGTTTCCCCTTTGGA")
    assert [(s - 1, e - 1) for s, e, _ in on_variant] == [(34, 65)] and on_variant[0][2] != unique[0][2]
    assert o.index_get("A" * 32) == []


def test_index_chr2():  # :83-143
    o, ref = _oracle("chr2")
    assert o.all_ref() == ref and o.index_check()
    a = "CCCCAGGTTTCCCCAGGTTTCCCCAGGTTTCC"
    assert o.index_get(a) == [(1, 32, 0), (1, 32, 2), (11, 42, INVALID_ID), (21, 52, INVALID_ID)]
    assert o.index_get("CCCCAGGTTTCCCCAGGTTTCCCCAGGTTTGG") == [(31, 62, INVALID_ID)]
    for k in ("CACCAGGTTTCCCCAGGTTTCCCCAGGTTTCC", "CCACAGGTTTCCCCAGGTTTCCCCAGGTTTCC", "CAACAGGTTTCCCCAGGTTTCCCCAGGTTTCC"):
        assert len(o.index_get(k)) == 2


def test_index_chr3():  # :145-209
    o, ref = _oracle("chr3")
    assert o.all_ref() == ref and o.index_check()
    assert o.index_get("AAAACAAAATAAAACAAAATAAAAGAAAACAA")[0][:2] == (1, 32)
    assert len(o.index_get("AAAACAAAATAAAACAAAATAAAAGAAAACAA")) == 1
    assert o.index_get("AAAACAAAATAAAACAAAATAAAAGAAAACGA") == [(1, SPECIAL_START, 2), (1, 32, 1)]
    assert o.index_get("AAAATAAAACAAAATAAAAGAAAACATTATAA") == [(31, 62, 0), (SPECIAL_START, 62, 2)]
    assert o.index_get("AAATAAAACAAAATAAAAGAAAACATTATAAA") == [(32, 63, INVALID_ID)]


def test_index_chr4():  # :211-244
    o, ref = _oracle("chr4")
    assert o.all_ref() == ref and o.index_check()
    assert o.index_get("AAAACAAAATAAAACAAAATAAAAGAAAACAA") == [(1, 32, 0)]
    assert o.index_get("ATAACAAAATAAAACAAAATAAAAGAAAACAA") == [(1, 32, 1)]


@pytest.mark.parametrize("chrom", ["chr9", "chr10"])
def test_index_events(chrom):  # :352-446  (GT_ID / GT_ANTI_HAPLOTYPE; GT_HAPLOTYPE is not parsed)
    o, _ = _oracle(chrom)
    assert o.index_check()
    assert len(o.index_get("G" * 32)) == 36
    lab = o.index_get("GGGGGAGTGGGGGGGGGGGGGGGGGGGGGGGG")
    assert len(lab) == 1 and lab[0][2] == 3
    lab = o.index_get("GGGGGGGTGGGGGGGGGGGGGGGGGGGGGGGG")
    assert sorted(l[2] for l in lab) == [0, 2]
    assert len(o.index_get("AGGGGGGTGGGGGGGGGGGGGGGGGGGGGGGG")) == 2
    if chrom == "chr9":
        assert len(o.index_get("AGGGGAGTGGGGGGGGGGGGGGGGGGGGGGGG")) == 0
    else:
        lab = o.index_get("AGGGGGAGTGGGGGGGGGGGGGGGGGGGGGGG")
        assert sorted(l[2] for l in lab) == [1, 3]


# ---------------------------------------------------------------- test/index/test_index.cpp:246-312 (an SV deletion)
def test_index_chr5():
    from fixtures import sv_contig
    ref, recs = sv_contig("chr5")
    o = Oracle(ref, recs, is_sv_graph=True)  # (create_test_graph's fourth argument is construct_graph's is_sv_graph)
    assert o.index_check()
    assert o.all_ref() == "A" * 70 + "C" * 70 + "G" * 70 + "T" * 70
    K = 32
    assert len(o.index_get(("A" * 32))) == 40
    l1 = o.index_get(("A" * 31 + "G"))
    assert len(l1) == 1 and (l1[0][0], l1[0][1]) == (40, SPECIAL_START)
    l2 = o.index_get(("A" * 30 + "GG"))
    assert len(l2) == 1 and (l2[0][0], l2[0][1]) == (41, SPECIAL_START + 1)
    l3 = o.index_get(("A" + "G" * 31))
    assert len(l3) == 1 and (l3[0][0], l3[0][1]) == (70, SPECIAL_START + 30)
    l4 = o.index_get(("G" * 32))
    assert len(l4) == 2 * (71 - K)
    assert sum(1 for lb in l4 if lb[0] == SPECIAL_START + 1) == 1
    assert len(o.index_get(("T" * 32))) == 2 * (71 - K)


# ---------------------------------------------------------------- test/graph/test_haplotypes.cpp:12-40
def test_haplotype_with_one_genotype():
    """two records at one position merge into one site: Graph::get_all_haplotypes() has one haplotype with three alleles
    (REQUIRE(haps.size() == 1); REQUIRE(haps[0].get_genotype_num() == 3)); the product's builder and layout agree"""
    from graphtyper_amd import lib as gtx
    recs = [(1, "GTACG", ["G"], "."), (1, "G", ["K"], ".")]
    og = Oracle("SGTACGEEF", recs).graph()
    assert int((og["ref_nvar"] > 0).sum()) == 1 and int(og["ref_nvar"].max()) == 3
    ctx = gtx.Context(gtx.graph_from_records("SGTACGEEF", recs), device=-1)
    assert ctx.n_hap == 1 and list(ctx.hap_cnum) == [3] and ctx.total_tri == 6 and ctx.total_allele == 3
