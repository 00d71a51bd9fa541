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


# ---- two records worked out by hand, every INFO key (src/graph/variant.cpp:230-1096, var_stats.cpp:53-141, logistic_constants.hpp).
# A read of 151 bases without mismatch adds 2^-(12-4) to the genotypes that hold its allele twice, one less to those that hold it once
# (haplotype.cpp:462-585): with r reference and a alternative reads ref/ref = 8 r, ref/alt = 7 (r + a), alt/alt = 8 a, and the phred
# value of a genotype is 3.0103 x (the best - its own), rounded, 255 at most.
def _one_site_records(n_alts, rows, n_samples=3, final=False):
    import numpy as np
    from graphtyper_amd import synth
    from oracle_lib import Oracle
    from test_vcf_text import _parse
    ref = synth.make_reference(1200, seed=5)
    rb, site = 30000, 600
    o = Oracle(synth.bases_to_str(ref), [(rb + site, "ACGT"[ref[site]], ["ACGT"[(ref[site] + k) % 4] for k in range(1, n_alts + 1)], None)], region_begin=rb)
    og = o.genotyper(n_samples, 1)
    s0 = site - 75
    reads = []
    for sample, allele, flag, mapq, score_diff, mismatches, *clipped in rows:
        r = ref[s0:s0 + 151].copy()
        r[75] = (ref[site] + allele) % 4
        if clipped:
            r[151 - clipped[0]:] = 3 - r[151 - clipped[0]:]  # complemented: no k-mer of it is found and no walk gets through it
        for p in (10, 50, 110, 140)[:mismatches]:  # one in a k-mer at most: the k-mer is still found, one substitution away
            r[p] = (r[p] + 1) % 4
        reads.append(synth._CODE_OF_BASE[r])
    n = len(rows)
    og.push(reads, flags=np.array([r[2] for r in rows], np.uint16), mapq=np.array([r[3] for r in rows], np.uint8), score_diff=np.array([r[4] for r in rows], np.uint8),
            pos=np.full(n, s0 + rb, np.int64), sample=np.array([r[0] for r in rows], np.int32), rg=np.zeros(n, np.int32))
    og.finish()
    _, records = _parse(og.vcf_records("chrT", ["A", "B", "C"][:n_samples]))
    assert len(records) == 1
    if final:
        return records[0], _parse(og.vcf_records_final("chrT", ["A", "B", "C"][:n_samples], synth.bases_to_str(ref), rb + 1))[1]
    return records[0]


def _sigmoid(x):
    import math
    return 1.0 / (1.0 + math.exp(-x))


def test_every_info_key_of_a_record_with_one_alternative_allele_by_hand():
    F, R = 0, 16
    rows = [(0, 0, F, 60, 0, 0), (0, 0, R, 60, 0, 0), (0, 1, F, 60, 5, 0), (0, 1, F, 60, 0, 0)]                 # A: 2 + 2
    rows += [(1, 0, F, 60, 0, 0)] * 3 + [(1, 0, R, 60, 0, 0)]                                                  # B: 4 + 0
    rows += [(2, 1, F, 60, 3, 0)] + [(2, 1, F, 60, 0, 0)] * 4 + [(2, 1, R, 30, 0, 0)] + [(2, 1, R, 60, 0, 0)] * 6  # C: 0 + 12
    r = _one_site_records(1, rows)
    # A: 16 / 28 / 16 -> PL 36, 0, 36; B: 32 / 28 / 0 -> 0, 12, 96; C: 0 / 84 / 96 -> 289 -> 255, 36, 0.  In the text PL and GQ are
    # binned (binned_pl.hpp: 33-37 are written as 35, 80-112 as 99); QUAL and the INFO keys use the values themselves
    assert r["samples"] == [["0/1", "2,2", "0", "4", "35", "35,0,35"], ["0/0", "4,0", "0", "4", "12", "0,12,99"], ["1/1", "0,12", "0", "12", "35", "255,35,0"]]
    assert r["qual"] == 36 + 0 + 255 and r["filt"] == "PASS"
    qd = (36 + 250) / (2 + 10)   # per sample with PL[0] > 0: min(25 x depth, PL[0]) over depth = min(10, alternative reads): A 36 / 2, C 250 / 10
    want = {
        "AC": "3", "AN": "6", "AF": "0.5", "NHomRef": "1", "NHet": "1", "NHomAlt": "1", "PexcessHet": "1",
        "PASS_AC": "3", "PASS_AN": "4", "PASS_ratio": "0.6667",              # GQ 36, 12, 36: A and C pass (>= 30)
        "MaxAAS": "12", "MaxAASR": "1", "SeqDepth": "20", "RefLen": "1", "VarType": "SG",
        "ABHet": "0.5", "ABHetMulti": "0.5,0.5", "ABHom": "1", "ABHomMulti": "1,1",
        "SBF": "4,7", "SBR": "2,7", "SBF1": "0,0", "SBF2": "4,7", "SBR1": "0,0", "SBR2": "2,7", "SB": "0.55", "SBAlt": "0.5",
        "MQsquared": str(19 * 3600 + 900), "MQ": "59", "MQSal": "21600,47700", "MQalt": "58",   # sqrt(69300 / 20) = 58.9, sqrt(47700 / 14) = 58.4
        "SDal": "0,8", "SDalt": "0.571429", "MMal": "0,0", "MMalt": "0", "CR": "0", "CRal": "0,0", "CRalt": "0",
        "QD": "23.83", "QDalt": "23.83",
        # logistic_constants.hpp:52-92: ABHom 1 is in the last bin; no strand bias (7 of 14 reverse); score difference round(8 / 14) = 1
        "AAScore": "%.4g" % _sigmoid(-6.347426707 + 3.930106559 + 1 * 0.014572295 + qd * 0.065221319 + 58 * 0.055973424),
        # :8-50: ABHet 0.5 and SBAlt 0.5 are in the bins without a term; every sample is genotyped
        "LOGF": "%.4g" % _sigmoid(-29.28908 + 1.0 * 23.12909 + 59 * 0.01024 + (2 / 3) * 0.85320 + 1.0 * 4.91178 + qd * 0.23215),
    }
    assert r["info"] == want


def test_every_info_key_of_a_record_with_two_alternative_alleles_by_hand():
    """A without a read; B 3 x reference + 1 x first alternative; C 1 x first + 5 x second alternative, those five with four mismatches
    (2^-(8-4) each) and a score difference of 2"""
    F, R = 0, 16
    rows = [(1, 0, F, 60, 0, 0)] * 2 + [(1, 0, R, 60, 0, 0), (1, 1, R, 60, 7, 0)]
    rows += [(2, 1, F, 60, 0, 0)] + [(2, 2, F, 60, 2, 4)] * 4 + [(2, 2, R, 60, 2, 4)]
    r = _one_site_records(2, rows)
    # genotypes in the order 0/0 0/1 1/1 0/2 1/2 2/2.  B: 24 28 8 21 7 0 -> PL 12 0 60 21 63 84.  C: 0 7 8 15 22 20 -> 66 45 42 21 0 6
    # (written in the bins of binned_pl.hpp: 18-22 as 20, 38-44 as 40, 45-54 as 50, 55-67 as 60, 80-112 as 99; no PL: no genotype)
    assert r["samples"] == [["./.", "0,0,0", "0", "0", "0", "0,0,0,0,0,0"], ["0/1", "3,1,0", "0", "4", "12", "12,0,60,20,60,99"],
                            ["1/2", "0,1,5", "0", "6", "6", "60,50,40,20,0,6"]]
    assert r["qual"] == 12 + 66
    qd = (12 + 66) / (1 + 6)
    aa2 = _sigmoid(-6.347426707 + 2.214801195 + 0.6 * -0.25233400 + 2.6 * -0.04129973 + 2 * 0.014572295 + 8.4 * 0.065221319 + 60 * 0.055973424)
    aa2 *= (1.0 - (2.6 - 1.5) / 20.0) * (1.0 - (2.6 - 2.5) / 40.0)   # more than 1.5 mismatches per read (x 100 / 151), more than 2.5 with the clipped bases
    want = {
        "AC": "2,1", "AN": "4", "AF": "0.5,0.25",                        # 0/0 (not genotyped: no PL), 0/1, 1/2 over two genotyped samples
        "NHomRef": "1,2", "NHet": "2,1", "NHomAlt": "0,0", "PexcessHet": "0.8,1",
        "PASS_AC": "0,0", "PASS_AN": "0", "PASS_ratio": "0",             # GQ 0, 12, 6
        "MaxAAS": "1,5", "MaxAASR": "0.25,0.8333", "SeqDepth": "10", "RefLen": "1", "VarType": "SG",
        "ABHet": "0.6",                                                  # second called allele over both: (1 + 5) / (3 + 1 + 1 + 5)
        "ABHetMulti": "0.25,0.8,0.1667",                                 # reads NOT of the allele over all, in the samples called with it: 1/4, (3 + 5)/10, 1/6
        "ABHom": "-1", "ABHomMulti": "-1,-1,-1",                         # the one homozygous call has no read
        "SBF": "2,1,4", "SBR": "1,1,1", "SBF1": "0,0,0", "SBF2": "2,1,4", "SBR1": "0,0,0", "SBR2": "1,1,1", "SB": "0.7", "SBAlt": "0.7143",
        "MQsquared": "36000", "MQ": "60", "MQSal": "10800,7200,18000", "MQalt": "60,60",
        "SDal": "0,7,10", "SDalt": "3.5,2", "MMal": "0,0,130", "MMalt": "0,2.6",   # 4 x 1000 / 151 = 26 per read
        "CR": "0", "CRal": "0,0,0", "CRalt": "0,0",
        "QD": "11.14",
        "QDalt": "9,8.4",   # first: B min(25, lowest PL without it = 12) + C min(25, 6) over 1 + 1 reads; second: C min(125, 42) over 5
        "AAScore": "0,%.4g" % aa2,                                       # the first has no sample with two reads
        # ABHom unknown counts as 0.985; ABHet 0.6 and SBAlt 0.71 have terms; two of three samples genotyped; none passes
        "LOGF": "%.4g" % _sigmoid(-29.28908 + 0.985 * 23.12909 + 60 * 0.01024 + (2 / 3) * 4.91178 + qd * 0.23215 - 1.05013 - 0.41332),
    }
    assert r["info"] == want


@pytest.mark.parametrize("n_reads", [1, 2])
def test_every_info_key_of_a_record_with_one_sample_and_one_or_two_reads_by_hand(n_reads):
    """the smallest records: one sample, one or two reads of the alternative allele (forward, mapping quality 40, score difference 4).
    alt/alt 8 n, ref/alt 7 n, ref/ref 0 -> PL 24 n, 3 n, 0; an allele needs a sample with two reads for an AAScore"""
    r = _one_site_records(1, [(0, 1, 0, 40, 4, 0)] * n_reads, n_samples=1)
    n = n_reads
    assert r["samples"] == [["1/1", "0,%d" % n, "0", str(n), "3" if n == 1 else "6", "25,3,0" if n == 1 else "50,6,0"]] and r["qual"] == 24 * n
    aa = 0.0 if n == 1 else _sigmoid(-6.347426707 + 3.930106559 + 1.0 * -0.25233400 + 4 * 0.014572295 + 24.0 * 0.065221319 + 40 * 0.055973424)  # every read forward
    want = {
        "AC": "2", "AN": "2", "AF": "1", "NHomRef": "0", "NHet": "0", "NHomAlt": "1", "PexcessHet": "1", "PASS_AC": "0", "PASS_AN": "0", "PASS_ratio": "0",
        "MaxAAS": str(n), "MaxAASR": "1", "SeqDepth": str(n), "RefLen": "1", "VarType": "SG",
        "ABHet": "-1", "ABHetMulti": "-1,-1", "ABHom": "1", "ABHomMulti": "-1,1",
        "SBF": "0,%d" % n, "SBR": "0,0", "SBF1": "0,0", "SBF2": "0,%d" % n, "SBR1": "0,0", "SBR2": "0,0", "SB": "1", "SBAlt": "1",
        "MQsquared": str(1600 * n), "MQ": "40", "MQSal": "0,%d" % (1600 * n), "MQalt": "40",
        "SDal": "0,%d" % (4 * n), "SDalt": "4", "MMal": "0,0", "MMalt": "0", "CR": "0", "CRal": "0,0", "CRalt": "0",
        "QD": "24", "QDalt": "24", "AAScore": "%.4g" % aa,
        "LOGF": "%.4g" % _sigmoid(-29.28908 + 23.12909 + 40 * 0.01024 + 4.91178 + 24.0 * 0.23215 - 1.60844),  # SBAlt 1: the last bin; no heterozygous call: ABHet counts as 0.5
    }
    assert r["info"] == want


def test_the_info_keys_of_clipped_reads_by_hand():
    """three reads of the alternative allele, the last 26 bases of two of them unalignable: those count 2^-(12-3-4) = 5 (4 for ref/alt),
    alt/alt 8 + 5 + 5, ref/alt 7 + 4 + 4 -> PL 54, 9, 0; CR is their number, CRal 26 x 1000 / 151 = 172 per read"""
    r = _one_site_records(1, [(0, 1, 0, 60, 0, 0), (0, 1, 0, 60, 0, 0, 26), (0, 1, 0, 60, 0, 0, 26)], n_samples=1)
    assert r["samples"] == [["1/1", "0,3", "0", "3", "9", "50,9,0"]] and r["qual"] == 54
    cr = 344 / 3 / 10.0
    aa = _sigmoid(-6.347426707 + 3.930106559 + 1.0 * -0.25233400 + 18.0 * 0.065221319 + cr * -0.01934834 + 60 * 0.055973424) * (1.0 - (cr - 2.5) / 40.0)
    i = r["info"]
    assert (i["CR"], i["CRal"], i["CRalt"], i["QD"], i["QDalt"], i["SeqDepth"], i["MaxAAS"]) == ("2", "0,344", "11.4667", "18", "18", "3", "3")
    assert i["AAScore"] == "%.4g" % aa
    assert i["LOGF"] == "%.4g" % _sigmoid(-29.28908 + 23.12909 + (2 / 3) * -10.22658 + 60 * 0.01024 + 4.91178 + 18.0 * 0.23215 - 1.60844)


def _two_base_site(alts, rows):
    """a site whose reference allele is the two bases at 30601; rows = (sample, allele) per read"""
    from graphtyper_amd import synth
    ref = synth.make_reference(1200, seed=5)
    s = synth.bases_to_str(ref)
    rb, site = 30000, 600
    assert s[site:site + 2] == "TT"
    og = Oracle(s, [(rb + site, "TT", alts, None)], region_begin=rb).genotyper(2, 1)
    s0 = site - 75
    reads = []
    for _, a in rows:
        r = ref[s0:s0 + 151].copy()
        if a:
            r[75], r[76] = ("ACGT".index(c) for c in alts[a - 1])
        reads.append(synth._CODE_OF_BASE[r])
    n = len(rows)
    og.push(reads, flags=np.zeros(n, np.uint16), mapq=np.full(n, 60, np.uint8), score_diff=np.zeros(n, np.uint8), pos=np.full(n, s0 + rb, np.int64),
            sample=np.array([r[0] for r in rows], np.int32), rg=np.zeros(n, np.int32))
    og.finish()
    return _parse(og.vcf_records("chrT", ["A", "B"]))[1], _parse(og.vcf_records_final("chrT", ["A", "B"], s, rb + 1))[1]


def test_a_site_of_two_bases_is_taken_apart_by_hand():
    """break_down_variant / break_multi_snps (variant.cpp:1652-1713, :1996-2111).  TT -> AA, TA; A: 2 x TT + 2 x AA, B: 3 x TA.
    Genotypes 0/0 0/1 1/1 0/2 1/2 2/2 -- A: 16 28 16 14 14 0 -> PL 36 0 36 42 42 84; B: 0 0 0 21 21 24 -> 72 72 72 9 9 0.
    First base: T, A, T -> alleles T, A with the third allele counted as the reference; second base: T, A, A.  A new genotype gets
    the smallest PL of the old ones that become it, a new allele the reads of the old ones"""
    rows = [(0, 0), (0, 0), (0, 1), (0, 1), (1, 2), (1, 2), (1, 2)]
    whole, final = _two_base_site(["AA", "TA"], rows)
    assert [(r["pos"], r["ref"], r["alts"], r["qual"]) for r in whole] == [(30601, "TT", ["AA", "TA"], 36 + 72)]
    assert whole[0]["samples"] == [["0/1", "2,2,0", "0", "4", "35", "35,0,35,40,40,99"], ["2/2", "0,0,3", "0", "3", "9", "75,75,75,9,9,0"]]
    assert [(r["pos"], r["ref"], r["alts"], r["qual"], r["id"]) for r in final] == [(30601, "T", ["A"], 36, "chrT:30601:SG"), (30602, "T", ["A"], 36 + 72, "chrT:30602:SG")]
    first, second = final
    assert first["samples"] == [["0/1", "2,2", "0", "4", "35", "35,0,35"], ["0/0", "3,0", "0", "3", "9", "0,9,75"]]        # B: min(72, 9, 0), min(72, 9), 72
    assert second["samples"] == [["0/1", "2,2", "0", "4", "35", "35,0,35"], ["1/1", "0,3", "0", "3", "9", "75,9,0"]]
    keys = ("AC", "AN", "MaxAAS", "SBF", "MQSal", "NHomRef", "NHet", "NHomAlt", "SeqDepth", "ABHomMulti", "QD")
    assert [first["info"][k] for k in keys] == ["1", "4", "2", "5,2", "18000,7200", "1", "1", "0", "7", "1,-1", "18"]       # QD: A alone, 36 / 2
    assert [second["info"][k] for k in keys] == ["3", "4", "3", "2,5", "7200,18000", "0", "1", "1", "7", "-1,1", "21.6"]   # (36 + 72) / (2 + 3)


def test_a_site_of_two_bases_whose_alleles_start_alike_by_hand():
    """TT -> TA, TC: nothing differs at the first base, the second one has three alleles and the calls stay as they are"""
    rows = [(0, 0), (0, 0), (0, 1), (0, 1), (1, 2), (1, 2), (1, 2)]
    whole, final = _two_base_site(["TA", "TC"], rows)
    assert [(r["pos"], r["ref"], r["alts"]) for r in whole] == [(30601, "TT", ["TA", "TC"])]
    assert [(r["pos"], r["ref"], r["alts"], r["qual"]) for r in final] == [(30602, "T", ["A", "C"], 36 + 72)]
    assert final[0]["samples"] == whole[0]["samples"] == [["0/1", "2,2,0", "0", "4", "35", "35,0,35,40,40,99"], ["2/2", "0,0,3", "0", "3", "9", "75,75,75,9,9,0"]]
    # an allele nobody is called with is left out: without B's reads the third allele goes, and its PL with it
    whole, final = _two_base_site(["TA", "TC"], rows[:4])
    assert whole[0]["alts"] == ["TA", "TC"] and whole[0]["samples"][0] == ["0/1", "2,2,0", "0", "4", "35", "35,0,35,40,40,99"]
    assert [(r["pos"], r["ref"], r["alts"]) for r in final] == [(30602, "T", ["A"])] and final[0]["samples"][0] == ["0/1", "2,2", "0", "4", "35", "35,0,35"]


def test_a_site_whose_allele_no_sample_has_two_reads_of_is_not_in_the_final_file():
    """variant.cpp:1036-1063 with vcf_operations.cpp:480-732: an alternative allele is "good" when some sample has two reads of it (and
    its QD is 1 or more); a record without a good one is left out of the file genotype() writes"""
    one, final = _one_site_records(1, [(0, 1, 0, 60, 0, 0)], n_samples=1, final=True)
    assert one["info"]["MaxAAS"] == "1" and final == []
    two, final = _one_site_records(1, [(0, 1, 0, 60, 0, 0)] * 2, n_samples=1, final=True)
    assert two["info"]["MaxAAS"] == "2" and [(r["pos"], r["samples"]) for r in final] == [(30601, two["samples"])]
    # one read in each of two samples is not two reads in one
    spread, final = _one_site_records(1, [(0, 1, 0, 60, 0, 0), (1, 1, 0, 60, 0, 0)], n_samples=2, final=True)
    assert spread["info"]["MaxAAS"] == "1" and spread["info"]["AC"] == "4" and final == []
    # of two alternative alleles one good one keeps the record, with both
    both, final = _one_site_records(2, [(0, 1, 0, 60, 0, 0)] * 2 + [(0, 2, 0, 60, 0, 0)], n_samples=1, final=True)
    assert both["info"]["MaxAAS"] == "2,1" and [r["alts"] for r in final] == [both["alts"]]
