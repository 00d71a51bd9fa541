"""VCF text of a genotyped region: gtx_vcf_records (graphtyper_amd/csrc/gtx_vcf.cpp) against the
oracle's restatement of Vcf::add_haplotype -> Variant::scan_calls / generate_infos -> Vcf::write_record (oracle/gto_vcf.hpp),
byte for byte, on stream scenarios deep enough to reach every FILTER and the logistic models; plus what can be checked
without the oracle: the binning table against the reference's own constants, the HWE p-value against its definition, and
the arithmetic relations between the fields of a record.  The alignment / scoring in front runs through the emulation here
and on the device in tests/test_gpu_parity.py (same run_stream)."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

import harness
import scenarios
from graphtyper_amd import lib as gtx
from oracle_lib import Oracle, lib as olib
from test_emu_parity import run_stream

HERE = os.path.dirname(os.path.abspath(__file__))


def test_binned_pl_equals_the_reference_table():
    want = json.load(open(os.path.join(HERE, "golden", "binned_pl.json")))["binned_pl"]
    L = olib()
    L.gto_binned_pl.restype = C.c_uint16
    assert [int(L.gto_binned_pl(C.c_uint(i))) for i in range(256)] == want


def test_excess_het_p_value_is_the_exact_test():
    """p_hwe_excess_het (snp_hwe.cpp): P(#het >= observed | allele counts) under Hardy-Weinberg, Wigginton et al. 2005"""
    L = olib()
    L.gto_p_hwe_excess_het.restype = C.c_double

    def exact(het, hom1, hom2):
        n = het + hom1 + hom2
        na = 2 * min(hom1, hom2) + het  # copies of the rarer allele
        nb = 2 * n - na
        if het == 0 and (hom1 == 0 or hom2 == 0):
            return 1.0

        def logp(h):  # probability of h heterozygotes given na, nb
            return (h * math.log(2) + math.lgamma(n + 1) - math.lgamma((na - h) // 2 + 1) - math.lgamma(h + 1) - math.lgamma((nb - h) // 2 + 1)
                    + math.lgamma(na + 1) + math.lgamma(nb + 1) - math.lgamma(2 * n + 1))
        hs = range(na % 2, na + 1, 2)
        tot = sum(math.exp(logp(h)) for h in hs)
        return min(1.0, sum(math.exp(logp(h)) for h in hs if h >= het) / tot)
    for het, a, b in [(0, 5, 0), (3, 10, 1), (10, 3, 3), (50, 20, 30), (1, 0, 0), (7, 100, 0), (0, 4, 4), (120, 500, 9)]:
        got = L.gto_p_hwe_excess_het(C.c_int(het), C.c_int(a), C.c_int(b))
        assert abs(got - exact(het, a, b)) < 1e-9 * max(1.0, got), (het, a, b, got, exact(het, a, b))


def _parse(text):
    lines = text.decode().split("\n")
    assert lines[-1] == "" and lines[0].startswith("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO")
    names = lines[0].split("\t")[9:]
    out = []
    for l in lines[1:-1]:
        f = l.split("\t")
        assert len(f) == 9 + len(names), l
        info = dict(kv.split("=", 1) for kv in f[7].split(";"))
        assert list(info) == sorted(info), "INFO keys are written in std::map order"
        out.append(dict(chrom=f[0], pos=int(f[1]), id=f[2], ref=f[3], alts=f[4].split(","), qual=int(f[5]), filt=f[6], info=info, fmt=f[8],
                        samples=[s.split(":") for s in f[9:]]))
    return names, out


def _check_records(records, binned):
    """relations the reference's code implies between the fields of one record"""
    bins = sorted(set(binned))
    for r in records:
        n_all = 1 + len(r["alts"])
        assert r["fmt"] == "GT:AD:MD:DP:GQ:PL"
        ac = [0] * n_all
        genotyped = 0
        qual_floor = 0
        for gt, ad, md, dp, gq, pl in r["samples"]:
            ad = [int(x) for x in ad.split(",")]
            pl = [int(x) for x in pl.split(",")]
            assert len(ad) == n_all and len(pl) == n_all * (n_all + 1) // 2
            assert int(dp) == sum(ad) + int(md)
            assert all(p in bins for p in pl) and int(gq) in bins and int(gq) <= 99
            qual_floor += pl[0]
            if gt != "./.":
                genotyped += 1
                a, b = (int(x) for x in gt.split("/"))
                assert a <= b and pl[b * (b + 1) // 2 + a] == 0
            else:
                assert not any(pl)
                a = b = 0
            ac[a] += 1
            ac[b] += 1
        assert int(r["info"]["AN"]) == 2 * genotyped
        assert [int(x) for x in r["info"]["AC"].split(",")] == ac[1:]
        assert r["qual"] >= 0 and (r["qual"] == 0) == (qual_floor == 0)  # QUAL sums the unbinned PL[0]
        assert int(r["info"]["RefLen"]) == len(r["ref"])
        assert r["id"].startswith("%s:%d:%s" % (r["chrom"], r["pos"], r["info"]["VarType"]))
        assert ("LowQUAL" in r["filt"]) == (r["qual"] < 10)
        if r["info"]["ABHet"] != "-1":
            assert ("LowABHet" in r["filt"]) == (float(r["info"]["ABHet"]) < 0.175)
        if int(r["info"]["AN"]) >= 6:
            assert ("LowQD" in r["filt"]) == (float(r["info"]["QD"]) < 6.0)
            assert ("LowAAScore" in r["filt"]) == (not any(float(x) > 0.15 for x in r["info"]["AAScore"].split(",")))


@pytest.mark.parametrize("kind", ["snp100", "indel", "cluster"])
def test_vcf_text_deep(kind):
    rb = 310000
    if kind != "snp100":  # unpaired reads cut from haplotypes that carry the alleles
        ref, recs, codes, pos = scenarios.synthetic_case(kind, n_ref=9000, n_reads=5000, region_begin=rb, seed=3)
        kw = dict(add_all_variants=True) if kind == "cluster" else {}
        order = np.argsort(pos, kind="stable")
        rec = scenarios.stream_records(len(codes), pos, sample=np.arange(len(codes)) % 4)[order]
        codes, n_samples = codes[order], 4
    else:
        ref, recs, codes, rec = scenarios.paired_case(kind, n_ref=6000, n_pairs=2400, region_begin=rb, n_samples=4, lowq_frac=0.03)
        kw, n_samples = {}, 4
    o = Oracle(ref, recs, region_begin=rb, **kw)
    b = harness.EmuBackend(gtx.graph_from_records(ref, recs, region_begin=rb, **kw))
    run_stream(b, o, codes, rec, n_samples=n_samples)  # compares the text with the oracle's, whole region and a filtered part
    names, records = _parse(run_stream.vcf_full)
    assert names == ["SAMP%02d" % i for i in range(n_samples)] and len(records) == b.ctx.n_hap
    binned = json.load(open(os.path.join(HERE, "golden", "binned_pl.json")))["binned_pl"]
    _check_records(records, binned)
    filters = set(x for r in records for x in r["filt"].split(";"))
    assert "PASS" in filters and len(filters) >= 3, filters
    assert any(float(x) > 0.15 for r in records for x in r["info"]["AAScore"].split(","))
    assert any(s[0] not in ("0/0", "./.") and s[0][0] == s[0][2] for r in records for s in r["samples"]), "no homozygous alt call"
    if kind == "cluster":
        assert max(len(r["alts"]) for r in records) >= 5 and any(r["info"]["VarType"] == "XG" for r in records)
    if kind == "indel":
        assert any(r["info"]["VarType"] == "IG" for r in records)


def test_vcf_text_without_samples_and_of_sv_graphs():
    ref, recs, codes, rec = scenarios.paired_case("snp100", n_ref=3000, n_pairs=10, region_begin=1000)
    g = gtx.graph_from_records(ref, recs, region_begin=1000)
    c = gtx.Context(g, device=-1)
    z = lambda n, t: np.zeros(max(1, n), t)
    text = c.vcf_records("chr1", [], z(0, np.uint32), z(c.n_hap + 2 * c.total_allele, np.uint64), z(c.n_hap + 6 * c.total_allele, np.uint32),
                         z(0, np.uint8), z(0, gtx.SAMPLE_CALL))
    lines = text.decode().split("\n")
    assert lines[0] == "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO" and len(lines) == c.n_hap + 2
    assert all(l.split("\t")[6] == "." and l.count("\t") == 7 for l in lines[1:-1])
    # an SV graph without SV alleles and without samples: the sites go through the SV post-processing untouched and, with
    # nobody called, every one of them is dropped (vcf_operations.cpp:640-652) -- the column line is all there is
    sv = gtx.Context(g, device=-1, is_sv_graph=True)
    text = sv.vcf_records("chr1", [], z(0, np.uint32), z(c.n_hap + 2 * c.total_allele, np.uint64), z(c.n_hap + 6 * c.total_allele, np.uint32),
                          z(0, np.uint8), z(0, gtx.SAMPLE_CALL), sv_table="")
    assert text.decode() == "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"


def test_records_written_by_a_team_of_host_threads_are_the_same_bytes(monkeypatch):
    """large jobs are cut into site ranges for host threads: the text must not depend on the team"""
    from graphtyper_amd import synth
    ref = synth.make_reference(300000, seed=42)
    recs = synth.make_snp_records(ref, 300, seed=7, region_begin=1000)
    ctx = gtx.Context(gtx.graph_from_records(synth.bases_to_str(ref), recs, region_begin=1000), device=-1)
    ns = 250
    nh, ta, tt = ctx.n_hap, ctx.total_allele, ctx.total_tri
    assert nh * (ns + 1) >= 200000
    rng = np.random.default_rng(1)
    gt_cov = rng.integers(0, 30, size=ns * ta).astype(np.uint32)
    stat_u64 = rng.integers(0, 1000, size=nh + 2 * ta).astype(np.uint64)
    stat_u32 = rng.integers(0, 1000, size=nh + 6 * ta).astype(np.uint32)
    phred = rng.integers(0, 255, size=ns * tt).astype(np.uint8)
    calls = np.zeros(ns * nh, gtx.SAMPLE_CALL)
    calls["gt_second"] = rng.integers(0, 2, size=ns * nh)
    calls["ref_total_depth"] = rng.integers(0, 40, size=ns * nh)
    calls["alt_total_depth"] = rng.integers(0, 40, size=ns * nh)
    calls["gq"] = rng.integers(0, 99, size=ns * nh)
    names = ["S%04d" % i for i in range(ns)]
    texts = []
    for team in ("1", "3", "7"):
        monkeypatch.setenv("GTX_HOST_THREADS", team)
        texts.append(ctx.vcf_records("chr1", names, gt_cov, stat_u64, stat_u32, phred, calls))
    assert texts[0] == texts[1] == texts[2] and texts[0].count(b"\n") == nh + 1


def test_header_is_the_references_text():
    """gtx_vcf_header: the ##INFO / ##FORMAT / ##FILTER lines are the reference's own output for them (tests/golden/
    vcf_header_definitions.txt, made by make_vcf_header.py from Vcf::write_header's literals), the lines around them follow
    vcf.cpp:526-548 and 691-758"""
    import os
    golden = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vcf_header_definitions.txt"), "rb").read()
    text = gtx.vcf_header("20260927", "2.7.7", [("chr1", 1000), ("chr20", 64444167)], ["A", "B"], git_branch="master", git_sha1="abc")
    head = (b"##fileformat=VCFv4.2\n##fileDate=20260927\n##source=Graphtyper\n##graphtyperVersion=2.7.7\n##graphtyperGitBranch=master\n"
            b"##graphtyperSHA1=abc\n##contig=<ID=chr1,length=1000>\n##contig=<ID=chr20,length=64444167>\n")
    assert text == head + golden + b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tA\tB\n"
    assert golden.count(b"\n") == 83 and b"##INFO=<ID=SVMODEL" in golden
    text = gtx.vcf_header("20260927", "2.7.7", [], ["A"], dirty=True, drop_genotypes=True)
    assert b"##graphtyperVersion=2.7.7-dirty\n" in text and text.endswith(b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")


def test_bgzf_members():
    """gtx_bgzf_compress: members of at most 0xff00 bytes with the BC field and their own size in it, the input back through
    zlib, and the end-of-file member htslib writes"""
    import struct
    import zlib
    rng = np.random.default_rng(5)
    data = bytes(rng.integers(65, 70, size=200000).astype(np.uint8))
    blob = gtx.bgzf_compress(data, level=6)
    eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    assert blob.endswith(eof)
    at, out, sizes = 0, b"", []
    while at < len(blob):
        assert blob[at:at + 4] == b"\x1f\x8b\x08\x04" and blob[at + 12:at + 16] == b"BC\x02\x00"
        bsize = struct.unpack("<H", blob[at + 16:at + 18])[0] + 1
        payload = zlib.decompress(blob[at + 18:at + bsize - 8], -15)
        crc, isize = struct.unpack("<II", blob[at + bsize - 8:at + bsize])
        assert isize == len(payload) <= 0xff00 and crc == (zlib.crc32(payload) & 0xFFFFFFFF)
        out += payload
        sizes.append(isize)
        at += bsize
    assert out == data and sizes[-1] == 0 and sizes[:3] == [0xff00] * 3
    assert gtx.bgzf_compress(b"", with_eof=True) == eof and gtx.bgzf_compress(b"", with_eof=False) == b""
    # a large input is deflated on several threads, each a run of consecutive members: the bytes are those of one thread
    big = bytes(rng.integers(65, 91, size=3_000_000).astype(np.uint8)) + data
    for level in (1, 6):
        one_by_one = b"".join(gtx.bgzf_compress(big[at:at + 0xff00], level=level, with_eof=False) for at in range(0, len(big), 0xff00))
        assert gtx.bgzf_compress(big, level=level, with_eof=False) == one_by_one
        assert gtx.bgzf_compress(big, level=level, with_eof=True) == one_by_one + eof
    # the marker is the same 28 bytes at every level (deflating nothing at level 0 would give a 31-byte stored-block member)
    for level in (0, 1, 9):
        blob = gtx.bgzf_compress(data[:70000], level=level)
        assert blob.endswith(eof) and len(blob) > len(eof)
        assert gtx.bgzf_compress(b"", level=level, with_eof=True) == eof


def test_final_vcf_breaks_merged_snp_clusters_down():
    """gtx_vcf_records_final on a graph whose SNPs (one every 7 bases) were merged into sites of up to 16 alleles of up to 22 bases, all
    of one length per site: break_multi_snps (variant.cpp:1996-2111) takes them apart position by position -- the text is the oracle's
    (compared inside run_stream, both modes: no site here needs paw::Skyr), every record is a SNP, alleles nobody is called with are
    gone, and a site without a called alternative allele leaves no record"""
    rb = 310000
    ref, recs, codes, pos = scenarios.synthetic_case("snp7", n_ref=6000, n_reads=4000, region_begin=rb, seed=3)
    g = gtx.graph_from_records(ref, recs, region_begin=rb, add_all_variants=True)
    assert int(g["ref_nvar"].max()) >= 8 and int(g["var_len"].max()) >= 10
    order = np.argsort(pos, kind="stable")
    rec = scenarios.stream_records(len(codes), pos, sample=np.arange(len(codes)) % 3)[order]
    o = Oracle(ref, recs, region_begin=rb, add_all_variants=True)
    b = harness.EmuBackend(g)
    run_stream(b, o, codes[order], rec, n_samples=3)
    names, final = _parse(run_stream.final)
    _, whole = _parse(run_stream.vcf_full)
    assert len(final) > len(whole) and all(len(r["ref"]) == 1 and all(len(a) == 1 for a in r["alts"]) for r in final)
    for r in final:
        ac = [int(x) for x in r["info"]["AC"].split(",")]
        assert all(x > 0 for x in ac) or len(ac) == 1  # (break_multi_snps keeps the alleles somebody is called with)
        assert ref[r["pos"] - rb - 1] == r["ref"]
    snp_positions = {p0 + 1 for p0, rf, alts, _ in recs}
    assert {r["pos"] for r in final} <= snp_positions


def test_final_vcf_of_a_snp_graph_is_the_called_part_of_the_records():
    """on a graph of bi-allelic SNPs a broken-down variant is the variant itself: the final file holds exactly the records of
    gtx_vcf_records whose alternative allele somebody is called with and generate_infos does not call bad"""
    rb = 310000
    ref, recs, codes, rec = scenarios.paired_case("snp100", n_ref=6000, n_pairs=2400, region_begin=rb, n_samples=4, lowq_frac=0.03)
    o = Oracle(ref, recs, region_begin=rb)
    b = harness.EmuBackend(gtx.graph_from_records(ref, recs, region_begin=rb))
    run_stream(b, o, codes, rec, n_samples=4)
    whole = run_stream.vcf_full.split(b"\n")[1:-1]
    final = run_stream.final.split(b"\n")[1:-1]
    assert 0 < len(final) < len(whole) and set(final) <= set(whole)
    dropped = [l for l in whole if l not in set(final)]
    for l in dropped:
        f = l.decode().split("\t")
        info = dict(kv.split("=", 1) for kv in f[7].split(";"))
        assert info["AC"] == "0" or float(info["QDalt"]) < 1.0 or int(info["MaxAAS"]) < 2, l[:200]


def test_final_vcf_of_indel_sites_nobody_carries():
    """break_down_skyr (variant.cpp:2113-2190) hands paw::Skyr the reference allele in place of every alternative allele nobody is
    called with (:2137-2155): when that is all of a site's alleles there is nothing to find and the site leaves no record -- the one
    case of that function that does not hang on the absent library.  A SNP graph with two deletions and an insertion that none of the
    reads carries: the final file WITHOUT no_variant_overlapping is made (it was refused), equals the oracle's (inside run_stream)
    and holds no record at those sites; a site somebody carries is still refused (test_emu_parity's indel scenarios)."""
    rb = 310000
    ref, recs, codes, rec = scenarios.paired_case("snp100", n_ref=6000, n_pairs=2400, region_begin=rb, n_samples=4, lowq_frac=0.03)
    snps = {p0 for p0, _, _, _ in recs}
    extra = []
    for p0, kind in ((rb + 1234, "del"), (rb + 2950, "ins"), (rb + 4321, "del")):
        while any(abs(p0 - q) < 40 for q in snps):
            p0 += 1
        at = p0 - rb
        if kind == "del":
            extra.append((p0, ref[at:at + 5], [ref[at]], None))
        else:
            extra.append((p0, ref[at], [ref[at] + "GATTACA"], None))
    recs2 = sorted(list(recs) + extra, key=lambda r: r[0])
    o = Oracle(ref, recs2, region_begin=rb)
    b = harness.EmuBackend(gtx.graph_from_records(ref, recs2, region_begin=rb))
    run_stream(b, o, codes, rec, n_samples=4)  # (compares gtx_vcf_records_final with the oracle's in both modes; neither may refuse)
    assert set(run_stream.final_by_mode) == {True, False}, "the mode without no_variant_overlapping was refused"
    final = run_stream.final_by_mode[False].split(b"\n")[1:-1]
    whole = run_stream.vcf_full.split(b"\n")[1:-1]
    at = {int(l.split(b"\t")[1]) for l in final}
    assert len(final) > 10 and not any(p0 + 1 in at for p0, _, _, _ in extra)
    assert any(int(l.split(b"\t")[1]) == extra[0][0] + 1 for l in whole)  # (the whole records have the sites)
