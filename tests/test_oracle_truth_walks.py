"""The walks at the ends of a read (Graph::get_labels_forward / get_labels_backward, src/graph/graph.cpp:1187-1701, through
GenotypePaths::walk_read_ends / walk_read_starts, src/typer/genotype_paths.cpp:483-621) held to simulated truth -- part of the kill
suite of tests/oracle_mutants/: the mechanical audit found the backward walk not executed once by the ground-truth tests (an
error-free read's first k-mer is always found in the index; only the last 26 bases of a 151-base read are walked).

Reads are drawn from two known haplotypes over a graph of SNPs, insertions and deletions.  Every read carries two substitution
errors in its first 32 bases (never on a base of a variant allele): its first k-mer is then two mismatches from anything in
the index, the chain starts at read offset 31, and the first 32 bases -- through whatever sites they hold, indels included -- are
reached by walk_read_starts alone (4 mismatches allowed there, genotype_paths.cpp:584).  A second set of reads has its two
errors in bases 93..124 instead: the LAST k-mer is lost and 58 bases are walked forward.  The truth: every read that holds a
site with ten bases to spare on both sides supports the allele it was drawn with and no other, so the AD column of the VCF text
lies between that count and the count of all reads that touch the site; the genotype is the haplotypes' pair of alleles."""
import numpy as np
import pytest

from graphtyper_amd import synth
from oracle_lib import Oracle
from test_vcf_text import _parse

READ_LEN = 151


def _case(seed, with_indels):
    rng = np.random.default_rng(seed)
    n_ref, rb = 12000, 200000
    ref = synth.make_reference(n_ref, seed=500 + seed)
    recs, kinds = [], []
    p = 300
    while p < n_ref - 300:
        kind = int(rng.integers(0, 4)) if with_indels else 0
        if kind <= 1:  # SNP
            recs.append((rb + p, "ACGT"[ref[p]], ["ACGT"[(ref[p] + 1 + rng.integers(0, 3)) % 4]], None))
        elif kind == 2:  # insertion of 1-5 bases that do not continue the reference (no second placement of the same event)
            ins = rng.integers(0, 4, size=int(rng.integers(1, 6)), dtype=np.uint8)
            ins[0] = (ref[p + 1] + 1 + rng.integers(0, 3)) % 4
            ins[-1] = (ref[p] + 1) % 4 if ins[-1] == ref[p] else ins[-1]
            recs.append((rb + p, "ACGT"[ref[p]], ["ACGT"[ref[p]] + synth.bases_to_str(ins)], None))
        else:  # deletion of 1-5 bases, likewise not shiftable
            dl = int(rng.integers(1, 6))
            while ref[p + dl] == ref[p] or ref[p + dl + 1] == ref[p + 1]:
                p += 1
            recs.append((rb + p, synth.bases_to_str(ref[p:p + dl + 1]), ["ACGT"[ref[p]]], None))
        kinds.append(kind)
        p += int(rng.integers(38, 64))
    # two haplotypes; per site the allele of each
    alleles = rng.integers(0, 2, size=(2, len(recs)))
    haps, spans = [], []
    for h in range(2):
        out, tag, span = [], [], []
        at = 0
        for k, (pos, r, alts, _) in enumerate(recs):
            q = pos - rb
            out.extend(ref[at:q])
            tag.extend([-1] * (q - at))
            a = r if alleles[h, k] == 0 else alts[0]
            span.append((len(out), len(out) + len(a)))
            out.extend("ACGT".index(c) for c in a)
            tag.extend([k] * len(a))
            at = q + len(r)
        out.extend(ref[at:])
        tag.extend([-1] * (n_ref - at))
        haps.append((np.array(out, np.uint8), np.array(tag), span))
    return ref, rb, recs, alleles, haps


def _reads(haps, rng, n_reads, error_window):
    reads, origin = [], []
    for _ in range(n_reads):
        h = int(rng.integers(0, 2))
        seq, tag, _ = haps[h]
        s = int(rng.integers(1, len(seq) - READ_LEN - 1))
        r = seq[s:s + READ_LEN].copy()
        ok = [i for i in range(*error_window) if tag[s + i] == -1 and tag[s + i - 1] == -1 and tag[s + i + 1] == -1]
        a, b = rng.choice(ok, size=2, replace=False)
        for i in (a, b):
            r[i] = (r[i] + 1 + rng.integers(0, 3)) % 4
        reads.append(r)
        origin.append((h, s))
    return reads, origin


def _bounds(recs, alleles, haps, origin, margin=10):
    """per site and allele: reads that hold the site with `margin` bases to spare on both sides / reads that touch it at all"""
    lo = np.zeros((len(recs), 2), int)
    hi = np.zeros((len(recs), 2), int)
    for h, s in origin:
        span = haps[h][2]
        for k, (b, e) in enumerate(span):
            if e <= s or b >= s + READ_LEN:
                continue
            a = alleles[h, k]
            hi[k, a] += 1
            if b - s >= margin and s + READ_LEN - e >= margin:
                lo[k, a] += 1
    return lo, hi


@pytest.mark.parametrize("with_indels,error_window", [(False, (1, 31)), (True, (1, 31)), (True, (94, 124)), (False, (94, 124))])
def test_reads_that_need_a_walk_support_the_allele_they_were_drawn_with(with_indels, error_window):
    ref, rb, recs, alleles, haps = _case(3 if with_indels else 4, with_indels)
    rng = np.random.default_rng(77)
    reads, origin = _reads(haps, rng, 40 * len(ref) // READ_LEN, error_window)
    order = np.argsort([s for _, s in origin], kind="stable")
    reads, origin = [reads[i] for i in order], [origin[i] for i in order]
    og = Oracle(synth.bases_to_str(ref), recs, region_begin=rb).genotyper(1, 1)
    og.push([synth._CODE_OF_BASE[r] for r in reads], pos=np.array([s for _, s in origin]) + rb)
    _, records = _parse(og.vcf_records("chrT", ["S"]))
    assert len(records) == len(recs)
    lo, hi = _bounds(recs, alleles, haps, origin)
    checked = 0
    for k, r in enumerate(records):
        assert r["pos"] == recs[k][0] + 1 and r["ref"] == recs[k][1] and r["alts"] == list(recs[k][2])
        ad = [int(x) for x in r["samples"][0][1].split(",")]
        for a in (0, 1):
            assert lo[k, a] <= ad[a] <= hi[k, a], (k, recs[k], a, ad, lo[k].tolist(), hi[k].tolist())
        gt = tuple(sorted(int(alleles[h, k]) for h in range(2)))
        if lo[k, gt[0]] >= 8 and lo[k, gt[1]] >= 8:
            assert r["samples"][0][0] == "%d/%d" % gt, (k, recs[k], r["samples"][0], gt)
            checked += 1
    assert checked > 0.9 * len(recs)
    # (what the window of errors costs: the reads whose site lies in it are a fifth of those over a site -- were the walk lost,
    #  every site's depth would be below its lower bound)
    assert lo.sum() > 0.8 * hi.sum()
