"""Structural-variant alleles of the constructor: the product (gtx_graph_from_files, graphtyper_amd/csrc/gtx_files.cpp)
against the line-by-line restatement in tests/sv_constructor.py feeding the oracle -- node tables, sequences, special
positions and the whole index must agree.  Every SV type and every case of the reference's builders
(src/graph/constructor.cpp:312-1207) is driven at least once; the reference's own SV tests are commented out
(test/graph/test_constructor.cpp:278-411), so this is product == restatement, not product == reference vectors."""
import os

import numpy as np
import pytest

import sv_constructor
from fixtures import GOLDEN, read_fasta
from graphtyper_amd import lib as gtx
from oracle_lib import Oracle


@pytest.fixture(scope="module", autouse=True)
def _built():
    gtx.build()


def _write(tmp_path, seqs, lines):
    fa = tmp_path / "ref.fa"
    with open(fa, "w") as f:
        for name, s in seqs.items():
            f.write(">%s\n" % name)
            for i in range(0, len(s), 60):
                f.write(s[i:i + 60] + "\n")
    vcf = tmp_path / "in.vcf"
    vcf.write_text("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n" + "\n".join(lines) + "\n")
    return str(fa), str(vcf)


def _compare(fa_path, vcf_path, seqs, lines, chrom, region=None):
    region = region or chrom
    g, (rb, re_), table = gtx.graph_from_files(fa_path, vcf_path, region, is_sv_graph=True, with_sv_table=True)
    recs, want_table = sv_constructor.sv_records(seqs, lines, chrom, region_begin=rb, region_end=re_ if ":" in region and "-" in region else 0xFFFFFFFF,
                                                 with_table=True)
    # Graph::SVs (what the calls' post-processing reads: type, breakpoint model, the other breakpoint, sizes, sequences)
    assert table.split("\n") == want_table.split("\n")
    assert table.count("\n") >= g["dna"].tobytes().decode().count("<SV:")  # (a breakpoint record behind the region is dropped, its SV stays registered)
    ref = "".join(c if c in "ACGT" else "N" for c in seqs[chrom].upper())[rb:re_]
    o = Oracle(ref, recs, region_begin=rb, is_sv_graph=True, extend_prefix=True)
    og = o.graph()
    for k in ("ref_order", "ref_len", "ref_nvar", "var_order", "var_len", "var_out_ref"):
        assert np.array_equal(g[k], og[k]), (k, g[k][:12], og[k][:12])
    assert g["dna"].tobytes() == og["dna"].tobytes()
    c = gtx.Context(g, device=-1, is_sv_graph=True)
    rr, ap = c.special_positions()
    assert np.array_equal(rr, og["ref_reach_poses"]) and np.array_equal(ap, og["actual_poses"])
    k1, c1, l1 = o.index_dump()
    k2, c2, l2 = c.index_dump()
    assert np.array_equal(k1, k2) and np.array_equal(c1, c2) and np.array_equal(l1, l2)
    return g, recs


def _random_contigs(seed, n=6000):
    rng = np.random.default_rng(seed)
    return {"chrA": "".join("ACGT"[i] for i in rng.integers(0, 4, n)), "chrB": "".join("ACGT"[i] for i in rng.integers(0, 4, n // 2))}


def test_every_sv_type_and_case(tmp_path):
    seqs = _random_contigs(1)
    a = seqs["chrA"]
    ins_long = "".join("ACGT"[i] for i in np.random.default_rng(2).integers(0, 4, 400))
    ins_short = ins_long[:40]
    lines = [
        # deletions: plain, with an inserted sequence, ALU-tagged
        "chrA\t301\t.\t%s\t<DEL>\t0\t.\tSVTYPE=DEL;SVSIZE=120" % a[300],
        "chrA\t601\t.\t%s\t<DEL>\t0\t.\tSVTYPE=DEL;SVLEN=-75;SVINSSEQ=ACGTTGCA" % a[600],
        "chrA\t801\t.\t%s\t<DEL:ME:ALU>\t0\t.\tSVTYPE=DEL:ME:ALU;SVSIZE=300;END=1101" % a[800],
        # insertions: long and short sequence, from an origin (long / short), incomplete (both, left only, right only)
        "chrA\t1301\t.\t%s\t<INS>\t0\t.\tSVTYPE=INS;SVLEN=400;SEQ=%s" % (a[1300], ins_long),
        "chrA\t1501\t.\t%s\t<INS>\t0\t.\tSVTYPE=INS;SVLEN=40;SEQ=%s" % (a[1500], ins_short),
        "chrA\t1701\t.\t%s\t<INS>\t0\t.\tSVTYPE=INS;SVSIZE=500;ORSTART=4001;OREND=4500" % a[1700],
        "chrA\t1901\t.\t%s\t<INS>\t0\t.\tSVTYPE=INS;SVSIZE=60;ORSTART=4601;OREND=4660" % a[1900],
        "chrA\t2101\t.\t%s\t<INS>\t0\t.\tSVTYPE=INS;LEFT_SVINSSEQ=%s;RIGHT_SVINSSEQ=%s" % (a[2100], ins_long[:200], ins_long[200:390]),
        "chrA\t2201\t.\t%s\t<INS>\t0\t.\tSVTYPE=INS;LEFT_SVINSSEQ=%s" % (a[2200], ins_long[10:90]),
        "chrA\t2301\t.\t%s\t<INS>\t0\t.\tSVTYPE=INS;RIGHT_SVINSSEQ=%s" % (a[2300], ins_long[20:70]),
        "chrA\t2351\t.\t%s\t<INS:ME:ALU>\t0\t.\tSVTYPE=INS:ME:ALU;SVSIZE=10" % a[2350],  # (skipped by the reference)
        # duplications: tandem long / short (a second record at the other breakpoint), with ORSTART, with OREND
        "chrA\t2501\t.\t%s\t<DUP>\t0\t.\tSVTYPE=DUP;SVLEN=400" % a[2500],
        "chrA\t3001\t.\t%s\t<DUP>\t0\t.\tSVTYPE=DUP;SVSIZE=30;SVINSSEQ=TTG" % a[3000],
        "chrA\t3201\t.\t%s\t<DUP>\t0\t.\tSVTYPE=DUP;SVSIZE=80;ORSTART=5001" % a[3200],
        "chrA\t3301\t.\t%s\t<DUP>\t0\t.\tSVTYPE=DUP;SVSIZE=80;OREND=5300" % a[3300],
        # inversions: tandem long / short, INV3, INV5
        "chrA\t3501\t.\t%s\t<INV>\t0\t.\tSVTYPE=INV;SVLEN=350" % a[3500],
        "chrA\t3951\t.\t%s\t<INV>\t0\t.\tSVTYPE=INV;SVSIZE=25" % a[3950],
        "chrA\t4051\t.\t%s\t<INV>\t0\t.\tSVTYPE=INV;SVSIZE=200;INV3" % a[4050],
        "chrA\t4301\t.\t%s\t<INV>\t0\t.\tSVTYPE=INV;SVSIZE=100;INV5;SVINSSEQ=AC" % a[4300],
        # breakends: the four orientations, two of them with inserted bases, one to another contig
        "chrA\t4701\t.\t%s\t%sAC[chrB:1001[\t0\t.\tSVTYPE=BND" % (a[4700], a[4700]),
        "chrA\t4801\t.\t%s\t[chrA:5401[GT%s\t0\t.\tSVTYPE=BND" % (a[4800], a[4800]),
        "chrA\t4901\t.\t%s\t]chrB:2001]%s\t0\t.\tSVTYPE=BND" % (a[4900], a[4900]),
        "chrA\t5001\t.\t%s\t%s]chrA:5601]\t0\t.\tSVTYPE=BND" % (a[5000], a[5000]),
        # small variants between them stay small variants
        "chrA\t5101\t.\t%s\t%s\t0\t.\t." % (a[5100], "ACGT"[("ACGT".index(a[5100]) + 1) % 4]),
        "chrA\t5151\t.\t%s\t%s\t0\t.\t." % (a[5150:5154], a[5150]),
    ]
    fa, vcf = _write(tmp_path, seqs, lines)
    g, recs = _compare(fa, vcf, seqs, lines, "chrA")
    tags = g["dna"].tobytes().decode().count("<SV:")
    assert tags == 31, tags  # one or two breakpoint alleles per SV: 3 DEL + 12 INS + 6 DUP + 6 INV + 4 BND
    assert sum(1 for r in recs if r[3] == "SV=1") == 26  # the 22 SV lines that build + the second records of the four tandem DUP / INV


def test_plain_long_indels_become_svs(tmp_path):
    """transform_sv_records: in an SV graph a plain deletion / insertion of 50 bases or more is rewritten into <DEL> / <INS>
    (also when its first bases differ: the base in front becomes the padding base); 49 bases stay a small variant;
    an insertion that repeats its neighbourhood turns into a duplication"""
    seqs = _random_contigs(3)
    a = seqs["chrA"]
    ins = "".join("ACGT"[i] for i in np.random.default_rng(4).integers(0, 4, 70))
    lines = [
        "chrA\t501\t.\t%s\t%s\t0\t.\t." % (a[500:561], a[500]),                      # 60 bp deletion
        "chrA\t801\t.\t%s\t%s\t0\t.\tXX=1" % (a[800:860], "G" if a[800] != "G" else "T"),  # first bases differ
        "chrA\t1201\t.\t%s\t%s\t0\t.\t." % (a[1200], a[1200] + ins),                 # 70 bp insertion
        "chrA\t1501\t.\t%s\t%s\t0\t.\t." % (a[1500], ("G" if a[1500] != "G" else "T") + ins[:55]),  # first bases differ
        "chrA\t1801\t.\t%s\t%s\t0\t.\t." % (a[1800:1850], a[1800]),                  # 49 bp: a small variant
        "chrA\t2201\t.\t%s\t%s\t0\t.\t." % (a[2200], a[2200] + a[2201:2261]),        # repeats what follows: a duplication
        "chrA\t2601\t.\t%s\t%s\t0\t.\t." % (a[2600], a[2600] + a[2540:2600]),        # repeats what precedes: moved duplication
    ]
    fa, vcf = _write(tmp_path, seqs, lines)
    g, recs = _compare(fa, vcf, seqs, lines, "chrA")
    kinds = [r[3] for r in recs]
    assert kinds.count("SV=1") == 8 and kinds.count(".") == 1  # (each duplication: two records)


def test_fixture_sv_contigs_chr5_chr6_chr7():
    """the SV lines of the reference's own test VCF (DEL, DUP + INV, INS:ME:ALU)"""
    seqs = read_fasta()
    lines = [l.rstrip("\n") for l in open(os.path.join(GOLDEN, "index_test.vcf")) if not l.startswith("#")]
    for chrom in ("chr5", "chr6", "chr7"):
        _compare(os.path.join(GOLDEN, "index_test.fa"), os.path.join(GOLDEN, "index_test.vcf"), seqs, lines, chrom)


def test_bad_sv_lines_are_errors_not_graphs(tmp_path):
    seqs = _random_contigs(5, 2000)
    a = seqs["chrA"]
    for info, status in (("SVSIZE=70", 1), ("SVTYPE=DEL;SVSIZE=7x", 1)):
        fa, vcf = _write(tmp_path, seqs, ["chrA\t301\t.\t%s\t<DEL>\t0\t.\t%s" % (a[300], info)])
        with pytest.raises(gtx.GtxError) as e:
            gtx.graph_from_files(fa, vcf, "chrA", is_sv_graph=True)
        assert e.value.status == status
    fa, vcf = _write(tmp_path, seqs, ["chrA\t301\t.\t%s\t<DEL>\t0\t.\tSVTYPE=DEL;SVSIZE=70" % a[300]])
    with pytest.raises(gtx.GtxError) as e:
        gtx.graph_from_files(fa, vcf, "chrA", is_sv_graph=False)  # (the reference exits: an SV in a non-SV graph)
    assert e.value.status == 4


def test_sv_that_starts_inside_the_region_and_ends_behind_it(tmp_path):
    """a tandem duplication / inversion moves its second breakpoint record by SVLEN (constructor.cpp:727-871, 873-1031): when
    the SV starts inside the region and ends behind it, that record lies at or behind region_end and Graph::add_genomic_region
    drops it (graph.cpp:72-79) -- the region's graph is still built, from the records that stay"""
    seqs = _random_contigs(9, 4000)
    a = seqs["chrA"]
    lines = [
        "chrA\t301\t.\t%s\t%s\t0\t.\t." % (a[300], "ACGT"[("ACGT".index(a[300]) + 1) % 4]),
        "chrA\t901\t.\t%s\t<DUP>\t0\t.\tSVTYPE=DUP;SVLEN=500" % a[900],
        "chrA\t1001\t.\t%s\t<INV>\t0\t.\tSVTYPE=INV;SVLEN=400" % a[1000],
    ]
    fa, vcf = _write(tmp_path, seqs, lines)
    g_all, _ = _compare(fa, vcf, seqs, lines, "chrA", region="chrA:1-2000")   # both breakpoints of both SVs
    g_cut, recs = _compare(fa, vcf, seqs, lines, "chrA", region="chrA:1-1200")  # second breakpoints at 1400 / 1401: dropped
    assert len(g_cut["ref_order"]) < len(g_all["ref_order"]) and len(g_cut["var_order"]) >= 6
