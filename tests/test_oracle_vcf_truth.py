"""The oracle's VCF writers held to what their output must satisfy whatever wrote it -- no product involved (this file is part of
the kill suite of tests/oracle_mutants/: a misreading of Variant::scan_calls / generate_infos / write_record, of the break-down or
of vcf_merge_and_filter in oracle/gto_vcf.hpp, gto_sv.hpp has to fail HERE, not merely differ from the product).

The checks are relations between fields of one record that the reference's text implies (AN / AC / QUAL / DP / filters against
the sample columns, the binning table, the thresholds of the FILTER column applied to the INFO values as printed), relations
between the three files one genotyped region gives (records -> final file -> sites), and a sample simulated at 40x whose true
genotypes are known: the calls must be them."""
import json
import os

import numpy as np
import pytest

import scenarios
from oracle_lib import Oracle
from test_vcf_text import _parse, _check_records

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(kind, rb=310000, n_samples=4):
    if kind == "snp100":
        ref, recs, codes, rec = scenarios.paired_case(kind, n_ref=6000, n_pairs=2400, region_begin=rb, n_samples=n_samples, lowq_frac=0.03)
        kw = {}
    else:
        ref, recs, codes, pos = scenarios.synthetic_case(kind, n_ref=9000 if kind != "snp7" else 6000, n_reads=5000, region_begin=rb, seed=3)
        kw = dict(add_all_variants=True) if kind in ("cluster", "snp7") else {}
        order = np.argsort(pos, kind="stable")
        rec = scenarios.stream_records(len(codes), pos, sample=np.arange(len(codes)) % n_samples)[order]
        codes = codes[order]
    o = Oracle(ref, recs, region_begin=rb, **kw)
    og = o.genotyper(n_samples, 1)
    og.push(list(codes), flags=rec["flag"], tid=rec["tid"], mtid=rec["mtid"], pos=rec["pos"], isize=rec["isize"], mapq=rec["mapq"],
            score_diff=rec["score_diff"], name=rec["name_id"], sample=rec["sample"], rg=rec["rg"])
    return ref, recs, o, og, ["SAMP%02d" % i for i in range(n_samples)]


def _more_relations(records, n_samples):
    for r in records:
        info = r["info"]
        ad = np.array([[int(x) for x in s[1].split(",")] for s in r["samples"]])
        md = np.array([int(s[2]) for s in r["samples"]])
        n_all = 1 + len(r["alts"])
        genotyped = [s[0] != "./." for s in r["samples"]]
        gts = [tuple(int(x) for x in s[0].split("/")) if s[0] != "./." else (0, 0) for s in r["samples"]]
        assert int(info["SeqDepth"]) == int(ad.sum() + md.sum())
        assert [int(x) for x in info["MaxAAS"].split(",")] == [int(ad[:, a].max()) for a in range(1, n_all)]
        nhet = [sum(1 for g in gts if (g[0] == a) != (g[1] == a)) for a in range(1, n_all)]
        nhomalt = [sum(1 for g in gts if g == (a, a)) for a in range(1, n_all)]
        assert [int(x) for x in info["NHet"].split(",")] == nhet and [int(x) for x in info["NHomAlt"].split(",")] == nhomalt
        assert [int(x) for x in info["NHomRef"].split(",")] == [n_samples - h - m for h, m in zip(nhet, nhomalt)]
        assert int(info["PASS_AN"]) <= int(info["AN"]) and int(info["AN"]) == 2 * sum(genotyped)
        g4 = lambda x: "%.4g" % x  # (the reference writes these through a stream with precision 4: four significant digits, as TEXT)
        if int(info["AN"]):
            assert info["PASS_ratio"] == g4(int(info["PASS_AN"]) / int(info["AN"]))
            assert info["AF"].split(",") == [g4(int(ac) / int(info["AN"])) for ac in info["AC"].split(",")]
        assert int(info["RefLen"]) == len(r["ref"])
        # MaxAASR: the largest share of a sample's unique depth an alternative allele has (scan_calls, variant.cpp:300-312), as text
        tot = ad.sum(axis=1)
        assert info["MaxAASR"].split(",") == ["%.4g" % max([ad[s, a] / tot[s] for s in range(len(ad)) if tot[s] > 0] or [0.0]) for a in range(1, n_all)]
        # MQ: the root of the mean squared mapping quality, rounded half away from zero (variant.cpp:810-830)
        if "MQsquared" in info:
            assert info["MQ"] == (str(int(np.floor(np.sqrt(int(info["MQsquared"]) / int(info["SeqDepth"])) + 0.5))) if int(info["SeqDepth"]) else "0")
        # the strand counts of the alleles add up to the two totals
        sbf, sbr = [int(x) for x in info["SBF"].split(",")], [int(x) for x in info["SBR"].split(",")]
        assert sbf == [a + b for a, b in zip(map(int, info["SBF1"].split(",")), map(int, info["SBF2"].split(",")))]
        assert sbr == [a + b for a, b in zip(map(int, info["SBR1"].split(",")), map(int, info["SBR2"].split(",")))]
        if sum(sbf) + sum(sbr):
            assert info["SB"] == g4(sum(sbf) / (sum(sbf) + sum(sbr)))
        # SBAlt: the same over the alternative alleles only (variant.cpp:704-720)
        assert info["SBAlt"] == (g4(sum(sbf[1:]) / (sum(sbf[1:]) + sum(sbr[1:]))) if sum(sbf[1:]) + sum(sbr[1:]) else "-1")
        het = [(g, row) for g, row in zip(gts, ad) if g[0] != g[1]]
        if het and info["ABHet"] != "-1":
            first, second = sum(int(row[g[0]]) for g, row in het), sum(int(row[g[1]]) for g, row in het)
            assert info["ABHet"] == g4(second / (first + second))
        hom = [(g, row) for g, row in zip(gts, ad) if g[0] == g[1]]
        if info["ABHom"] != "-1":
            called, total = sum(int(row[g[0]]) for g, row in hom), sum(int(row.sum()) for g, row in hom)
            assert info["ABHom"] == g4(called / total)
        if int(info["AN"]) >= 6:
            assert ("LowABHom" in r["filt"]) == (info["ABHom"] != "-1" and float(info["ABHom"]) < 0.85)


@pytest.mark.parametrize("kind", ["snp100", "indel", "cluster"])
def test_records_satisfy_the_relations_of_their_fields(kind):
    ref, recs, o, og, names = _run(kind)
    got_names, records = _parse(og.vcf_records("chrT", names))
    assert got_names == names and len(records) > 20
    binned = json.load(open(os.path.join(HERE, "golden", "binned_pl.json")))["binned_pl"]
    _check_records(records, binned)
    _more_relations(records, len(names))
    assert any(r["filt"] == "PASS" for r in records) and any(r["filt"] != "PASS" for r in records)
    pos = [r["pos"] for r in records]
    assert pos == sorted(pos)
    for r in records:  # the alleles are the graph's: the reference allele is the reference
        assert ref[r["pos"] - 310000 - 1:r["pos"] - 310000 - 1 + len(r["ref"])] == r["ref"]


@pytest.mark.parametrize("kind", ["snp100", "snp7"])
def test_final_file_and_sites_follow_from_the_records(kind):
    rb = 310000
    ref, recs, o, og, names = _run(kind, rb)
    _, whole = _parse(og.vcf_records("chrT", names))
    _, final = _parse(og.vcf_records_final("chrT", names, ref, rb + 1, no_variant_overlapping=True))
    assert final == _parse(og.vcf_records_final("chrT", names, ref, rb + 1, no_variant_overlapping=False))[1]  # (alleles of one length per site)
    binned = json.load(open(os.path.join(HERE, "golden", "binned_pl.json")))["binned_pl"]
    _check_records(final, binned)
    _more_relations(final, len(names))
    assert all(len(r["ref"]) == 1 and all(len(a) == 1 for a in r["alts"]) for r in final)  # broken down to SNPs
    assert all(all(int(x) > 0 for x in r["info"]["AC"].split(",")) for r in final)            # by the alleles somebody is called with
    assert [r["pos"] for r in final] == sorted(r["pos"] for r in final) and all(ref[r["pos"] - rb - 1] == r["ref"] for r in final)
    if kind == "snp100":  # bi-allelic SNPs: a broken-down variant is the variant
        by_pos = {r["pos"]: r for r in whole}
        for r in final:
            w = by_pos[r["pos"]]
            assert (w["alts"], w["qual"], w["filt"], w["info"], w["samples"]) == (r["alts"], r["qual"], r["filt"], r["info"], r["samples"])
        kept = {r["pos"] for r in final}
        for w in whole:
            if w["pos"] not in kept:  # dropped: nobody called with it, or generate_infos calls its alternative allele bad
                assert w["info"]["AC"] == "0" or float(w["info"]["QDalt"]) < 1.0 or int(w["info"]["MaxAAS"]) < 2
            else:
                assert int(w["info"]["AC"]) > 0 and float(w["info"]["QDalt"]) >= 1.0 and int(w["info"]["MaxAAS"]) >= 2
    else:  # merged SNP clusters: more records than sites, every one of them a SNP of the input
        assert len(final) > len(whole)
        assert {r["pos"] for r in final} <= {p0 + 1 for p0, _, _, _ in recs}
    # ---- sites: one line per kept allele, numbered over all alleles of all sites in order
    lines = og.vcf_sites("chrT").decode().split("\n")
    assert lines[0] == "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO"
    n_alts_before, want_ids = 0, []
    for w in whole:
        qd = [float(x) for x in w["info"]["QDalt"].split(",")]
        aas = [int(x) for x in w["info"]["MaxAAS"].split(",")]
        for a in range(len(w["alts"])):
            if qd[a] >= 1.0 and aas[a] >= 2:
                want_ids.append((w["pos"], w["alts"][a], n_alts_before + a + 1))
        n_alts_before += len(w["alts"])
    got_ids = []
    for l in lines[1:-1]:
        f = l.split("\t")
        assert f[5] == "0" and f[6] == "."
        info = dict(kv.split("=", 1) for kv in f[7].split(";"))
        got_ids.append((int(f[1]), f[4], int(info["GT_ID"])))
    assert got_ids == want_ids and len(got_ids) > 10


def test_a_40x_sample_is_called_as_it_was_simulated():
    """error-free reads at ~40x from two known haplotypes (a third of the SNPs on one of them, a third on both): the GT column of
    every record is the pair of alleles the haplotypes carry, AD splits the depth accordingly, and the final file -- the variants
    broken down, records nobody is called with dropped -- holds exactly the sites where the sample is not homozygous reference"""
    from graphtyper_amd import synth
    n_ref, rb, read_len = 20000, 300000, 150
    n_reads = 40 * n_ref // read_len
    rng = np.random.default_rng(9)
    ref = synth.make_reference(n_ref, seed=77)
    recs = synth.make_snp_records(ref, 100, seed=5, region_begin=rb)
    pos = np.array([p - rb for p, _, _, _ in recs])
    alt = np.array(["ACGT".index(a[0]) for _, _, a, _ in recs], np.uint8)
    kind = rng.integers(0, 3, size=len(recs))  # 0: nobody, 1: one haplotype, 2: both
    haps = [ref.copy(), ref.copy()]
    haps[0][pos[kind >= 1]] = alt[kind >= 1]
    haps[1][pos[kind == 2]] = alt[kind == 2]
    which = rng.integers(0, 2, size=n_reads)
    start = np.sort(rng.integers(1, n_ref - read_len, size=n_reads))
    reads = [synth._CODE_OF_BASE[haps[h][s:s + read_len]] for h, s in zip(which, start)]
    ref_s = synth.bases_to_str(ref)
    og = Oracle(ref_s, recs, region_begin=rb).genotyper(1, 1)
    og.push(reads, pos=start + rb)
    _, records = _parse(og.vcf_records("chrT", ["S"]))
    assert len(records) == len(recs)
    inner = lambda k: 200 <= pos[k] <= n_ref - 200  # (thin coverage at the region's ends)
    for k, r in enumerate(records):
        if not inner(k):
            continue
        gt, ad = r["samples"][0][0], [int(x) for x in r["samples"][0][1].split(",")]
        assert gt == ("0/0", "0/1", "1/1")[kind[k]], (r["pos"], gt, kind[k])
        assert (ad[1] == 0) if kind[k] == 0 else (ad[0] == 0) if kind[k] == 2 else (ad[0] > 3 and ad[1] > 3), (r["pos"], ad, kind[k])
        assert r["pos"] == recs[k][0] + 1 and r["ref"] == recs[k][1] and r["alts"] == list(recs[k][2])
    _, final = _parse(og.vcf_records_final("chrT", ["S"], ref_s, rb + 1))
    want = [recs[k][0] + 1 for k in range(len(recs)) if kind[k] > 0 and inner(k)]
    got = [r["pos"] for r in final if 200 <= r["pos"] - rb - 1 <= n_ref - 200]
    assert got == want and len(want) > 50
