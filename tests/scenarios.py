"""Seeded inputs shared by the emulation (CPU) and GPU parity tests."""
import numpy as np

from fixtures import contig
from graphtyper_amd import lib as gtx
from graphtyper_amd import synth
from oracle_lib import encode

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def mutate(s, positions, rng):
    s = list(s)
    for p in positions:
        s[p] = "ACGT"[("ACGT".index(s[p]) + int(rng.integers(1, 4))) % 4] if s[p] in "ACGT" else "A"
    return "".join(s)


def haplotype_strings(ref, recs):
    """every allele combination of a few-record contig: list of (sequence)"""
    seqs = [""]
    cur = 0
    for pos, r, alts, _ in recs:
        seqs = [s + ref[cur:pos] for s in seqs]
        seqs = [s + a for s in seqs for a in [r] + [x for x in alts if not x.startswith("<")]]
        cur = pos + len(r)
    return [s + ref[cur:] for s in seqs]


def contig_reads(chrom, seed=1):
    """hand-made 63-66 bp reads over one index_test contig: every allele combination, with 0-2 mismatches, an N,
    and the reverse complement"""
    rng = np.random.default_rng(seed)
    ref, recs = contig(chrom)
    reads = []
    for h in haplotype_strings(ref, recs):
        if "N" in h:
            h = h[:h.index("N")]
        for L in (63, 64, len(h)):
            if L > len(h) or L < 63:
                continue
            for start in {0, len(h) - L}:
                s = h[start:start + L]
                reads.append(s)
                reads.append(mutate(s, rng.choice(L, 1, replace=False), rng))
                reads.append(mutate(s, rng.choice(L, 2, replace=False), rng))
                n = list(s)
                n[int(rng.integers(L))] = "N"
                reads.append("".join(n))
                reads.append(revcomp(s))
    return ref, recs, reads


def synthetic_case(kind, n_ref=40000, n_reads=300, seed=0, region_begin=1000000, read_len=150, err=0.005, n_rate=0.001):
    ref = synth.make_reference(n_ref, seed=seed + 100)
    if kind == "snp1k":
        recs = synth.make_snp_records(ref, 1000, seed=seed + 1, region_begin=region_begin)
    elif kind == "snp100":
        recs = synth.make_snp_records(ref, 100, seed=seed + 2, region_begin=region_begin)
    elif kind == "snp25":
        recs = synth.make_snp_records(ref, 25, seed=seed + 3, region_begin=region_begin)
    elif kind == "indel":
        recs = synth.make_indel_records(ref, 60, seed=seed + 4, region_begin=region_begin)
    elif kind == "cluster":  # merged multi-allelic sites: build the graph with add_all_variants=True
        recs = synth.make_cluster_records(ref, 150, seed=seed + 8, region_begin=region_begin)
    elif kind == "cfg3":  # SNP every 100 bp, a tenth of the sites short indels with a SNP close by (merged by add_all_variants)
        recs = synth.make_cfg3_records(ref, 100, seed=seed + 14, region_begin=region_begin)
    elif kind == "snp7":  # > 8 variant sites per read and wide graph walks: second-pass territory
        recs = synth.make_snp_records(ref, 7, seed=seed + 9, region_begin=region_begin)
    elif kind == "rows":  # rows of 6..10 SNP sites every second, third or fourth base, a row every 300 bases: seven or eight
        # separate sites under one k-mer give its neighbours 49 / 64 labels (round 5: more than the general pass' table held)
        rng = np.random.default_rng(seed + 21)
        recs = []
        for at in range(200, n_ref - 200, 300):
            step, count = int(rng.integers(2, 5)), int(rng.integers(6, 11))
            for p in range(at, at + step * count, step):
                alts = ["ACGT"[(ref[p] + k) % 4] for k in range(1, int(rng.integers(1, 3)) + 1)]
                recs.append((region_begin + p, "ACGT"[ref[p]], alts, None))
    elif kind == "repeat":
        # tandem repeat: 24 copies (16 of them diverged by 1.5 %) of a 180 bp unit in the middle of the region -> a read from it seeds at
        # dozens of places, results have more paths than a record slot holds
        rng = np.random.default_rng(seed + 10)
        unit = rng.integers(0, 4, size=180).astype(np.uint8)
        at = n_ref // 2
        for c in range(24):
            u = unit.copy()
            e = (rng.random(len(u)) < 0.015) & (c % 3 != 0)  # every third copy is exact
            u[e] = (u[e] + rng.integers(1, 4, size=int(e.sum()))) % 4
            ref[at + c * 180:at + (c + 1) * 180] = u
        recs = synth.make_snp_records(ref, 40, seed=seed + 11, region_begin=region_begin)
    elif kind == "neardup":
        # near-duplicate segments: 300 bp stretches copied elsewhere one substitution every ~40 bp apart (a 32-mer of the copy
        # is the original's or its Hamming-1 neighbour: what the per-position flags of the hinted pass have to see through),
        # plus homopolymer runs; SNPs every 100 bp on top
        rng = np.random.default_rng(seed + 12)
        n_seg = max(4, n_ref // 4000)
        for _ in range(n_seg):
            src = int(rng.integers(0, n_ref - 300))
            seg = ref[src:src + 300].copy()
            for _copy in range(int(rng.integers(1, 3))):
                dst = int(rng.integers(0, n_ref - 300))
                c = seg.copy()
                at = np.arange(int(rng.integers(5, 40)), 300, 40)
                c[at] = (c[at] + rng.integers(1, 4, size=len(at))) % 4
                ref[dst:dst + 300] = c
        for _ in range(n_seg):
            at = int(rng.integers(0, n_ref - 80))
            ref[at:at + int(rng.integers(20, 60))] = int(rng.integers(0, 4))
        recs = synth.make_snp_records(ref, 100, seed=seed + 13, region_begin=region_begin)
    elif kind == "satellite":
        # low-complexity repeats, where one exact k-mer has hundreds of places and the chains of a read multiply: a 280-bp
        # homopolymer, two copies (270 and 286 bp, 1 % diverged) of one dinucleotide repeat, a 2 kb array of a 171-bp unit
        # (0.5 % diverged copies), a trinucleotide repeat; SNPs every 50 bp, also inside the repeats.  The reference keeps
        # every chain (genotype_paths.cpp:294-352 has no limit): reads from these places are what the exact pass is for.
        rng = np.random.default_rng(seed + 15)
        assert n_ref >= 16000
        spots = {}
        at = 2000
        ref[at:at + 280] = 1
        spots["homopolymer"] = (at, 280)
        di = np.tile(np.array([0, 2], np.uint8), 143)
        for k, (where, size) in enumerate(((5000, 270), (8000, 286))):
            c = di[:size].copy()
            e = rng.random(size) < 0.01
            c[e] = (c[e] + rng.integers(1, 4, size=int(e.sum()))) % 4
            ref[where:where + size] = c
            spots["dinucleotide%d" % k] = (where, size)
        unit = rng.integers(0, 4, size=171).astype(np.uint8)
        at = 10000
        for c in range(12):
            u = unit.copy()
            e = (rng.random(171) < 0.005) & (c % 2 == 1)
            u[e] = (u[e] + rng.integers(1, 4, size=int(e.sum()))) % 4
            ref[at + c * 171:at + (c + 1) * 171] = u
        spots["array171"] = (at, 12 * 171)
        ref[14000:14000 + 240] = np.tile(np.array([1, 0, 3], np.uint8), 80)
        spots["trinucleotide"] = (14000, 240)
        recs = synth.make_snp_records(ref, 50, seed=seed + 16, region_begin=region_begin)
        codes, pos = [], []
        per = max(1, n_reads // len(spots))
        for k, (name, (where, size)) in enumerate(sorted(spots.items())):
            lo, hi = where - 300, where + size + 300
            sub = [r for r in recs if lo < r[0] - region_begin < hi - 2]
            c, p = synth.make_reads(ref[lo:hi], sub, per, read_len=read_len, seed=seed + 17 + k, err=err, n_rate=n_rate,
                                    region_begin=region_begin + lo, rev_frac=0.0)
            codes.append(c)
            pos.append(p)
        codes, pos = np.concatenate(codes), np.concatenate(pos)
        order = np.argsort(pos, kind="stable")
        return synth.bases_to_str(ref), recs, np.ascontiguousarray(codes[order]), pos[order]
    else:
        raise ValueError(kind)
    if kind == "repeat":  # reads from the repeat and its flanks only
        lo, hi = n_ref // 2 - 400, n_ref // 2 + 24 * 180 + 400
        sub = [r for r in recs if lo < r[0] - region_begin < hi - 2]
        codes, pos = synth.make_reads(ref[lo:hi], sub, n_reads, read_len=read_len, seed=seed + 5, err=err, n_rate=n_rate,
                                      region_begin=region_begin + lo, rev_frac=0.0)
        return synth.bases_to_str(ref), recs, codes, pos
    codes, pos = synth.make_reads(ref, recs, n_reads, read_len=read_len, seed=seed + 5, err=err, n_rate=n_rate,
                                  region_begin=region_begin, rev_frac=0.0)
    return synth.bases_to_str(ref), recs, codes, pos


def stream_records(n, pos, flags=None, mapq=None, sample=None, name=None, isize=None, l_qseq=150, tid=0, mtid=0, score_diff=None):
    r = np.zeros(n, gtx.STREAM_RECORD)
    r["pos"] = pos
    r["l_qseq"] = l_qseq
    r["tid"] = tid
    r["mtid"] = mtid
    r["mapq"] = 60 if mapq is None else mapq
    r["flag"] = 0 if flags is None else flags
    r["sample"] = 0 if sample is None else sample
    r["name_id"] = np.arange(n) if name is None else name
    r["isize"] = 0 if isize is None else isize
    r["score_diff"] = 0 if score_diff is None else score_diff
    return r


def paired_case(kind="snp100", n_ref=30000, n_pairs=150, seed=0, region_begin=500000, read_len=150, n_samples=2,
                discordant_frac=0.1, dup_frac=0.05, lowq_frac=0.1):
    """position-sorted stream of FR pairs + a few unpaired reads, duplicates, low-MAPQ and filtered records.
    Returns (reference string, records, codes [n, L], STREAM_RECORD array)."""
    rng = np.random.default_rng(seed + 77)
    ref = synth.make_reference(n_ref, seed=seed + 200)
    every = {"snp1k": 1000, "snp100": 100, "snp25": 25}.get(kind, 100)
    recs = synth.make_snp_records(ref, every, seed=seed + 6, region_begin=region_begin) if kind != "indel" else \
        synth.make_indel_records(ref, 60, seed=seed + 7, region_begin=region_begin)
    # haplotype 1 of every sample = reference with a random half of the SNPs (substitutions only keeps coordinates)
    rows = []
    for i in range(n_pairs):
        sample = int(rng.integers(n_samples))
        hap = ref.copy()
        if kind != "indel":
            take = np.random.default_rng(1000 + sample).random(len(recs)) < 0.5
            if rng.random() < 0.5:
                for (p, r, alts, _), t in zip(recs, take):
                    if t:
                        hap[p - region_begin] = "ACGT".index(alts[0])
        ins = int(np.clip(rng.normal(400, 50), read_len + 10, 900))
        start = int(rng.integers(0, n_ref - ins))
        a = hap[start:start + read_len].copy()
        b = hap[start + ins - read_len:start + ins].copy()
        for x in (a, b):
            e = rng.random(read_len) < 0.005
            x[e] = (x[e] + rng.integers(1, 4, size=int(e.sum()))) % 4
        mapq = 10 if rng.random() < lowq_frac else 60
        kindp = rng.random()
        name = i
        if kindp < discordant_frac:  # same strand -> both orientations are aligned
            if rng.random() < 0.5:  # second mate stored on the other strand: only its reverse orientation aligns
                b = (3 - b[::-1]).astype(np.uint8)
            rows.append((start, a, 1 | 64, ins, mapq, sample, name))
            rows.append((start + ins - read_len, b, 1 | 128, -ins, mapq, sample, name))
        elif kindp < discordant_frac + 0.1:  # unpaired
            rows.append((start, a, 0, 0, mapq, sample, name))
        else:
            rows.append((start, a, 1 | 2 | 32 | 64, ins, mapq, sample, name))
            rows.append((start + ins - read_len, b, 1 | 2 | 16 | 128, -ins, mapq, sample, name))
        if rng.random() < dup_frac:  # an unpaired PCR duplicate of the first mate right behind it (other name, other sample)
            rows.append((start, a, 0, 0, 60, int(rng.integers(n_samples)), 10_000_000 + i))
        if rng.random() < 0.05:  # a record the flag filter drops (secondary)
            rows.append((start, a, 256, 0, 60, sample, 20_000_000 + i))
    rows.sort(key=lambda r: r[0])  # stable: duplicates stay behind their original
    n = len(rows)
    codes = np.zeros((n, read_len), np.uint8)
    rec = np.zeros(n, gtx.STREAM_RECORD)
    for i, (p, bases, flag, isize, mapq, sample, name) in enumerate(rows):
        codes[i] = np.array([1, 2, 4, 8], np.uint8)[bases]
        rec[i] = (flag, mapq, int(rng.integers(0, 60)), 0, 0, p + region_begin, isize, read_len, 0, sample, name, 0, 0, 0, 0)
    return synth.bases_to_str(ref), recs, codes, rec


def sv_case(n_ref=60000, n_del=8, n_ins=4, n_samples=4, pairs_per_sv=30, background_pairs=200, seed=0, read_len=150):
    """cfg5-like input: synth.make_sv_case (shared with bench.py's cfg5 leg)"""
    return synth.make_sv_case(n_ref=n_ref, n_del=n_del, n_ins=n_ins, n_samples=n_samples, pairs_per_sv=pairs_per_sv,
                              background_pairs=background_pairs, seed=seed, read_len=read_len)


def wide_site_case(seed=0, region_begin=20000, n_ref=3000, step=3, err=0.004):
    """graph with sites of more than 64 alleles (allele sets beyond the 64-bit masks of the front passes; the reference
    allows MAX_NUMBER_OF_HAPLOTYPES = 2560): site A = one record with 99 insertion alleles (100 alleles), site B = a 12-bp
    deletion overlapped by 6 SNP records of 3 alternative alleles each -- add_all_variants merges overlapping records into
    every combination (2 * 4^5 = 2048 alleles) until the product reaches 2559, the sixth SNP then joins by
    VarRecord::merge_one_path (graph.cpp:119-124) --, plus ordinary SNPs.  Reads are tiled over both sites from four
    haplotypes that carry alleles with high numbers.  Returns (reference string, records, codes, pos0, (pA, pB))."""
    rng = np.random.default_rng(seed + 4242)
    ref = synth.make_reference(n_ref, seed=seed + 300)
    b2s = synth.bases_to_str
    pA, pB = 700, 1500
    ins = []
    while len(ins) < 99:
        s = b2s(rng.integers(0, 4, size=int(rng.integers(5, 13)), dtype=np.uint8))
        if s not in ins:
            ins.append(s)
    recs = [(pA + region_begin, b2s(ref[pA:pA + 1]), [b2s(ref[pA:pA + 1]) + s for s in ins], None)]
    recs.append((pB + region_begin, b2s(ref[pB:pB + 13]), [b2s(ref[pB:pB + 1])], None))
    snp_at = [pB + 1, pB + 3, pB + 5, pB + 7, pB + 9, pB + 11]
    for q in snp_at:
        recs.append((q + region_begin, b2s(ref[q:q + 1]), ["ACGT"[(int(ref[q]) + k) % 4] for k in (1, 2, 3)], None))
    for q in list(range(100, 600, 90)) + list(range(900, 1400, 110)) + list(range(1700, n_ref - 100, 130)):
        recs.append((q + region_begin, b2s(ref[q:q + 1]), ["ACGT"[(int(ref[q]) + 1) % 4]], None))
    recs.sort(key=lambda r: r[0])

    def haplotype(ins_k, deletion, snp_choice):
        h = [ref[:pA + 1]]
        if ins_k is not None:
            h.append(np.array(["ACGT".index(c) for c in ins[ins_k]], np.uint8))
        h.append(ref[pA + 1:pB + 1])
        if not deletion:
            mid = ref[pB + 1:pB + 13].copy()
            for q, k in zip(snp_at, snp_choice):
                if k:
                    mid[q - pB - 1] = (int(ref[q]) + k) % 4
            h.append(mid)
        h.append(ref[pB + 13:])
        return np.concatenate(h)
    haps = [haplotype(None, False, (0,) * 6), haplotype(70, False, (3, 3, 2, 0, 1, 0)), haplotype(5, True, None),
            haplotype(98, False, (1, 2, 3, 3, 3, 2)), haplotype(64, False, (0, 0, 0, 0, 3, 3))]
    codes, pos = [], []
    for h in haps:
        for centre in (pA, pB):
            for start in range(max(0, centre - 170), centre + 30, step):
                r = h[start:start + 150].copy()
                e = rng.random(150) < err
                r[e] = (r[e] + rng.integers(1, 4, size=int(e.sum()))) % 4
                codes.append(synth._CODE_OF_BASE[r])
                pos.append(start + region_begin)
    order = np.argsort(np.array(pos), kind="stable")
    return b2s(ref), recs, np.ascontiguousarray(np.array(codes, np.uint8)[order]), np.array(pos, np.int64)[order], (pA, pB)
