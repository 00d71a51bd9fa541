"""Synthetic inputs for the hot path: uniform random reference, SNP / indel variant records,
diploid 150 bp reads with substitution errors and Ns.  All seeds fixed; numpy only."""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_CODE_OF_BASE = np.array([1, 2, 4, 8], dtype=np.uint8)  # A C G T -> BAM/IUPAC 4-bit code


def make_reference(n, seed=42):
    """n i.i.d. uniform bases as a uint8 array of base indices (0..3)"""
    return np.random.default_rng(seed).integers(0, 4, size=n, dtype=np.uint8)


def bases_to_str(b):
    return _ACGT[b].tobytes().decode()


def make_snp_records(ref, every=1000, seed=7, region_begin=0, first=None):
    """one biallelic SNP every `every` bp; returns [(pos0, ref, [alt], None)] with contig coordinates"""
    rng = np.random.default_rng(seed)
    first = every // 2 if first is None else first
    pos = np.arange(first, len(ref) - 1, every)
    alt = (ref[pos] + rng.integers(1, 4, size=len(pos), dtype=np.uint8)) % 4
    return [(int(p) + region_begin, "ACGT"[ref[p]], ["ACGT"[a]], None) for p, a in zip(pos, alt)]


def make_indel_records(ref, every=500, seed=11, region_begin=0, max_len=6):
    """alternating SNP / insertion / deletion records, spaced so that none overlaps the next (no merging needed)"""
    rng = np.random.default_rng(seed)
    recs = []
    p = every // 2
    kind = 0
    while p + max_len + 2 < len(ref):
        if kind % 3 == 0:
            a = (ref[p] + rng.integers(1, 4)) % 4
            recs.append((p + region_begin, "ACGT"[ref[p]], ["ACGT"[a]], None))
        elif kind % 3 == 1:
            ins = rng.integers(0, 4, size=int(rng.integers(1, max_len + 1)), dtype=np.uint8)
            recs.append((p + region_begin, "ACGT"[ref[p]], ["ACGT"[ref[p]] + bases_to_str(ins)], None))
        else:
            dl = int(rng.integers(1, max_len + 1))
            recs.append((p + region_begin, bases_to_str(ref[p:p + dl + 1]), ["ACGT"[ref[p]]], None))
        kind += 1
        p += every
    return recs


def make_reads(ref, records, n, read_len=150, seed=123, err=0.005, n_rate=0.001, region_begin=0, rev_frac=0.0):
    """n reads drawn from a diploid sample (each record het with p=0.5 on haplotype 1, haplotype 0 = reference).
    Returns (codes [n, read_len] uint8 4-bit codes, pos0 [n] int64 contig coordinates of the read start on the reference).
    Only substitution/indel records produced by make_snp_records / make_indel_records are understood."""
    rng = np.random.default_rng(seed)
    # haplotype 1 = reference with a random half of the records applied
    hap, at = [], []  # at: the reference coordinate under every base of haplotype 1 (what a mapper reports as POS)
    cur = 0
    take = rng.random(len(records)) < 0.5
    for (p, r, alts, _), t in zip(records, take):
        p -= region_begin
        hap.append(ref[cur:p])
        at.append(np.arange(cur, p))
        if t:
            hap.append(np.array(["ACGT".index(c) for c in alts[0]], dtype=np.uint8))
            at.append(p + np.minimum(np.arange(len(alts[0])), len(r) - 1))
        else:
            hap.append(ref[p:p + len(r)])
            at.append(np.arange(p, p + len(r)))
        cur = p + len(r)
    hap.append(ref[cur:])
    at.append(np.arange(cur, len(ref)))
    hap1 = np.concatenate(hap)
    at1 = np.concatenate(at)
    haps = [ref, hap1]
    which = rng.integers(0, 2, size=n)
    out = np.zeros((n, read_len), dtype=np.uint8)
    pos = np.zeros(n, dtype=np.int64)
    for h in (0, 1):
        idx = np.nonzero(which == h)[0]
        src = haps[h]
        start = rng.integers(0, len(src) - read_len, size=len(idx))
        gather = start[:, None] + np.arange(read_len)[None, :]
        out[idx] = src[gather]
        pos[idx] = (start if h == 0 else at1[start]) + region_begin
    # substitution errors
    e = rng.random(out.shape) < err
    out = np.where(e, (out + rng.integers(1, 4, size=out.shape, dtype=np.uint8)) % 4, out).astype(np.uint8)
    codes = _CODE_OF_BASE[out]
    if rev_frac > 0:
        rv = rng.random(n) < rev_frac
        rc = _CODE_OF_BASE[3 - out[:, ::-1]]
        codes = np.where(rv[:, None], rc, codes)
    nmask = rng.random(codes.shape) < n_rate
    codes = np.where(nmask, np.uint8(15), codes).astype(np.uint8)
    return np.ascontiguousarray(codes), pos


def plant_repeats(ref, seed=21, homopolymers=40, short_tandems=40, arrays=4, near_duplicates=200):
    """Overwrites stretches of `ref` (in place) with what a real chromosome has and an i.i.d. reference lacks: homopolymer runs
    of 20-300 bp, di- / trinucleotide repeats of 50-300 bp, satellite arrays (12 copies of a 171-bp unit, every other copy
    0.5 % diverged) and near-duplicate segments (300 bp copied elsewhere with a substitution every ~40 bp).  Places are drawn
    uniformly; returns [(kind, start, length)]."""
    rng = np.random.default_rng(seed)
    n = len(ref)
    spots = []
    for _ in range(homopolymers):
        size = int(rng.choice([20, 35, 60, 100, 180, 300]))
        at = int(rng.integers(0, n - size))
        ref[at:at + size] = int(rng.integers(0, 4))
        spots.append(("homopolymer", at, size))
    for _ in range(short_tandems):
        unit = rng.integers(0, 4, size=int(rng.integers(2, 4))).astype(np.uint8)
        if len(set(unit.tolist())) == 1:
            unit[0] = (unit[0] + 1) % 4
        size = int(rng.choice([50, 90, 150, 300]))
        at = int(rng.integers(0, n - size))
        c = np.tile(unit, size // len(unit) + 1)[:size]
        e = rng.random(size) < 0.01
        c[e] = (c[e] + rng.integers(1, 4, size=int(e.sum()))) % 4
        ref[at:at + size] = c
        spots.append(("tandem", at, size))
    for _ in range(arrays):
        unit = rng.integers(0, 4, size=171).astype(np.uint8)
        at = int(rng.integers(0, n - 12 * 171))
        for c in range(12):
            u = unit.copy()
            e = (rng.random(171) < 0.005) & (c % 2 == 1)
            u[e] = (u[e] + rng.integers(1, 4, size=int(e.sum()))) % 4
            ref[at + c * 171:at + (c + 1) * 171] = u
        spots.append(("array", at, 12 * 171))
    for _ in range(near_duplicates):
        src, dst = int(rng.integers(0, n - 300)), int(rng.integers(0, n - 300))
        c = ref[src:src + 300].copy()
        at = np.arange(int(rng.integers(5, 40)), 300, 40)
        c[at] = (c[at] + rng.integers(1, 4, size=len(at))) % 4
        ref[dst:dst + 300] = c
        spots.append(("near-duplicate", dst, 300))
    return spots


def make_cluster_records(ref, every=500, seed=13, region_begin=0):
    """clusters of three biallelic sites a few bp apart (SNP, SNP, 1-6 bp insertion or deletion): with add_all_variants
    every cluster merges into one multi-allelic site (the cfg3-like graph)"""
    rng = np.random.default_rng(seed)
    recs = []
    p = every // 2
    while p + 30 < len(ref):
        q = p
        for k in range(3):
            if k < 2:
                a = (ref[q] + rng.integers(1, 4)) % 4
                recs.append((q + region_begin, "ACGT"[ref[q]], ["ACGT"[a]], None))
                q += int(rng.integers(2, 6))
            elif rng.random() < 0.5:
                ins = rng.integers(0, 4, size=int(rng.integers(1, 7)), dtype=np.uint8)
                recs.append((q + region_begin, "ACGT"[ref[q]], ["ACGT"[ref[q]] + bases_to_str(ins)], None))
            else:
                dl = int(rng.integers(1, 7))
                recs.append((q + region_begin, bases_to_str(ref[q:q + dl + 1]), ["ACGT"[ref[q]]], None))
        p += every
    return recs


def make_cfg3_records(ref, every=100, indel_frac=0.1, seed=17, region_begin=0):
    """SURVEY.md 8(d), the SNP+indel graph of cfg3: a biallelic site every `every` bp, `indel_frac` of them 1-10 bp insertions
    or deletions instead of SNPs, each of those with a SNP 2-9 bp behind it so that add_all_variants (graph.cpp:81-167) merges
    the two into one multi-allelic site.  Records do not overlap (make_reads applies them one after the other)."""
    rng = np.random.default_rng(seed)
    recs = []
    p = every // 2
    while p + 30 < len(ref):
        if rng.random() < indel_frac:
            if rng.random() < 0.5:
                ins = rng.integers(0, 4, size=int(rng.integers(1, 11)), dtype=np.uint8)
                recs.append((p + region_begin, "ACGT"[ref[p]], ["ACGT"[ref[p]] + bases_to_str(ins)], None))
                q = p + int(rng.integers(2, 10))
            else:
                dl = int(rng.integers(1, 11))
                recs.append((p + region_begin, bases_to_str(ref[p:p + dl + 1]), ["ACGT"[ref[p]]], None))
                q = p + dl + int(rng.integers(2, 10))
            a = (ref[q] + rng.integers(1, 4)) % 4
            recs.append((q + region_begin, "ACGT"[ref[q]], ["ACGT"[a]], None))
        else:
            a = (ref[p] + rng.integers(1, 4)) % 4
            recs.append((p + region_begin, "ACGT"[ref[p]], ["ACGT"[a]], None))
        p += every
    return recs


def make_sv_case(n_ref=60000, n_del=8, n_ins=4, n_samples=4, pairs_per_sv=30, background_pairs=200, seed=0, read_len=150):
    """cfg5-like input: a random contig, SV deletions of 50..5 000 bp and SV insertions whose sequence is
    longer than the 152-bp breakpoint window (so both breakpoint alleles are pure insertion sequence), FR pairs drawn from
    diploid samples around every SV (each sample carries an SV on its second haplotype with p = 0.4) plus background pairs.
    Returns ({contig: sequence}, VCF data lines, codes [n, L], STREAM_RECORD array sorted by position)."""
    from . import lib as gtx
    rng = np.random.default_rng(seed + 500)
    ref = make_reference(n_ref, seed=seed + 501)
    ref_s = bases_to_str(ref)
    n_sv = n_del + n_ins
    slots = np.linspace(2000, n_ref - 8000, n_sv).astype(int)
    kinds = np.array(["DEL"] * n_del + ["INS"] * n_ins)
    rng.shuffle(kinds)
    svs, lines = [], []
    for p, k in zip(slots, kinds):
        p = int(p)
        if k == "DEL":
            size = int(rng.choice([50, 75, 150, 400, 1200, 5000]))
            size = min(size, int(n_ref - p - 1000))
            lines.append("chrS\t%d\t.\t%s\t<DEL>\t0\t.\tSVTYPE=DEL;SVSIZE=%d" % (p + 1, ref_s[p], size))
            svs.append((p, "DEL", size, None))
        else:
            ins = rng.integers(0, 4, size=int(rng.integers(170, 420))).astype(np.uint8)
            lines.append("chrS\t%d\t.\t%s\t<INS>\t0\t.\tSVTYPE=INS;SVLEN=%d;SEQ=%s" % (p + 1, ref_s[p], len(ins), bases_to_str(ins)))
            svs.append((p, "INS", len(ins), ins))
    # second haplotype of every sample: (sequence, reference coordinate of every base)
    haps = []
    for s in range(n_samples):
        take = rng.random(n_sv) < 0.4
        parts, coords, cur = [], [], 0
        for (p, k, size, ins), t in zip(svs, take):
            parts.append(ref[cur:p + 1])
            coords.append(np.arange(cur, p + 1))
            cur = p + 1
            if t and k == "DEL":
                cur = p + 1 + size
            elif t:
                parts.append(ins)
                coords.append(np.full(len(ins), p))
        parts.append(ref[cur:])
        coords.append(np.arange(cur, n_ref))
        haps.append((np.concatenate(parts), np.concatenate(coords)))
    rows = []
    name = 0

    def pair(sample, around):
        nonlocal name
        seq, coord = (ref, np.arange(n_ref)) if rng.random() < 0.5 else haps[sample]
        ins = int(np.clip(rng.normal(400, 50), read_len + 10, 900))
        if around is None:
            start = int(rng.integers(0, len(seq) - ins))
        else:
            at = int(np.searchsorted(coord, around))
            start = int(np.clip(at - rng.integers(20, ins - 20), 0, len(seq) - ins))
        a, b = seq[start:start + read_len].copy(), seq[start + ins - read_len:start + ins].copy()
        for x in (a, b):
            e = rng.random(read_len) < 0.004
            x[e] = (x[e] + rng.integers(1, 4, size=int(e.sum()))) % 4
        pa, pb = int(coord[start]), int(coord[start + ins - read_len])
        mapq = 60 if rng.random() < 0.93 else 10
        rows.append((pa, a, 1 | 2 | 32 | 64, pb - pa + read_len, mapq, sample, name, pb))
        rows.append((pb, b, 1 | 2 | 16 | 128, -(pb - pa + read_len), mapq, sample, name, pa))
        name += 1

    for (p, k, size, _) in svs:
        for _ in range(pairs_per_sv):
            pair(int(rng.integers(n_samples)), p + (size // 2 if k == "DEL" and rng.random() < 0.5 else 0))
    for _ in range(background_pairs):
        pair(int(rng.integers(n_samples)), None)
    rows.sort(key=lambda r: r[0])
    n = len(rows)
    codes = np.zeros((n, read_len), np.uint8)
    rec = np.zeros(n, gtx.STREAM_RECORD)
    M = 0
    for i, (p, bases, flag, isize, mapq, sample, nm, mpos) in enumerate(rows):
        codes[i] = np.array([1, 2, 4, 8], np.uint8)[bases]
        rec[i] = (flag, mapq, int(rng.integers(0, 60)), 0, 0, p, isize, read_len, 0, sample, nm, mpos, 1, read_len << 4 | M, read_len << 4 | M)
    return {"chrS": ref_s}, lines, codes, rec


def write_fixed_bam(path, contig, contig_len, sample, codes, pos0, mapq=60, flag=0, threads=8, level=1):
    """A position-sorted BAM of unpaired reads of one length, written without a per-record Python loop (benchmarks: the
    pipeline leg of bench.py reads such files back through gtx_reads).  codes: [n, L] 4-bit BAM codes, pos0: [n] 0-based
    positions (sorted).  Records: name r%09d, cigar LM, qualities 30, no aux fields; BGZF members of 236 records, deflated
    on `threads` threads (zlib releases the GIL)."""
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    codes = np.ascontiguousarray(codes, np.uint8)
    n, L = codes.shape
    nb = (L + 1) // 2
    name_len = 11  # "r%09d" + NUL
    rec = np.dtype([("block_size", "<i4"), ("refid", "<i4"), ("pos", "<i4"), ("l_read_name", "u1"), ("mapq", "u1"), ("bin", "<u2"),
                    ("n_cigar", "<u2"), ("flag", "<u2"), ("l_seq", "<i4"), ("next_refid", "<i4"), ("next_pos", "<i4"), ("tlen", "<i4"),
                    ("name", "u1", name_len), ("cigar", "<u4"), ("seq", "u1", nb), ("qual", "u1", L)])
    out = np.zeros(n, rec)
    out["block_size"] = rec.itemsize - 4
    out["pos"] = pos0
    out["l_read_name"] = name_len
    out["mapq"] = mapq
    beg = np.asarray(pos0, np.int64)
    end = beg + L - 1
    b = np.zeros(n, np.int64)
    done = np.zeros(n, bool)
    for shift, first in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):  # reg2bin (SAM spec 5.3)
        hit = ~done & ((beg >> shift) == (end >> shift))
        b[hit] = first + (beg[hit] >> shift)
        done |= hit
    out["bin"] = b
    out["n_cigar"] = 1
    out["flag"] = flag
    out["l_seq"] = L
    out["next_refid"] = -1
    out["next_pos"] = -1
    idx = np.arange(n, dtype=np.int64)
    out["name"][:, 0] = ord("r")
    for k in range(9):
        out["name"][:, 1 + k] = 48 + (idx // 10 ** (8 - k)) % 10
    out["cigar"] = L << 4
    padded = codes if L % 2 == 0 else np.concatenate([codes, np.zeros((n, 1), np.uint8)], axis=1)
    out["seq"] = (padded[:, 0::2] << 4) | padded[:, 1::2]
    out["qual"] = 30
    header_text = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:%s\tLN:%d\n@RG\tID:%s\tSM:%s\n" % (contig, contig_len, sample, sample)
    head = b"BAM\1" + struct.pack("<i", len(header_text)) + header_text.encode() + struct.pack("<i", 1)
    head += struct.pack("<i", len(contig) + 1) + contig.encode() + b"\0" + struct.pack("<i", contig_len)

    def member(chunk):
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = c.compress(chunk) + c.flush()
        return struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25) + comp + \
            struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))

    raw = out.view(np.uint8).reshape(n, rec.itemsize)
    per = max(1, 65000 // rec.itemsize)
    chunks = [head] + [raw[a:a + per].tobytes() for a in range(0, n, per)] + [b""]
    with ThreadPoolExecutor(max(1, threads)) as pool:
        members = list(pool.map(member, chunks))
    with open(path, "wb") as f:
        for m in members:
            f.write(m)
    return rec.itemsize


def make_genome_like_reference(n, seed=1999, segdups=8, segdup_len=(5000, 20000)):
    """A reference with the structure an i.i.d. one lacks and a human chromosome has, in the proportions of one: a background drawn from
    an order-5 Markov chain (its transition table random but fixed, GC ~ 41 %, CpG depleted), ~45 % of the bases in copies of 24
    interspersed repeat families (consensus of 280-320 bp -- Alu-like -- or 900-2 000 bp -- L1 fragments --, every copy 5-20 %
    diverged from its consensus: substitutions and a few short indels, some copies truncated, half of them reverse-complemented),
    ~3 % short tandem repeats (units of 1-6 bp, 5-40 copies, a substitution every ~60 bp), and 8 segmental duplications of 5-20 kb,
    1-2 % diverged.  Returns (bases uint8 [n], {"interspersed": fraction, "str": fraction, "segdup": fraction, "families": k})."""
    rng = np.random.default_rng(seed)
    # ---- background: order-5 Markov chain
    table = rng.dirichlet(np.array([2.9, 2.1, 2.1, 2.9]) * 3.0, size=4 ** 5)
    cg = np.arange(4 ** 5) % 4 == 1  # context ends in C: G after it is rare (CpG depletion)
    table[cg, 2] *= 0.25
    table /= table.sum(1, keepdims=True)
    cum = np.cumsum(table, axis=1)
    u = rng.random(n)
    ref = np.empty(n, np.uint8)
    ref[:5] = rng.integers(0, 4, 5)
    ctx = 0
    for i in range(5):
        ctx = (ctx * 4 + int(ref[i])) & 1023
    for i in range(5, n):
        b = int(np.searchsorted(cum[ctx], u[i]))
        b = 3 if b > 3 else b
        ref[i] = b
        ctx = ((ctx * 4) + b) & 1023
    covered = np.zeros(n, bool)

    def diverge(seq, rate):
        seq = seq.copy()
        m = rng.random(len(seq)) < rate
        seq[m] = (seq[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) % 4
        out, i = [], 0
        for at in np.nonzero(rng.random(len(seq)) < rate / 12.0)[0]:  # a short indel now and then
            out.append(seq[i:at])
            if rng.random() < 0.5:
                out.append(rng.integers(0, 4, size=int(rng.integers(1, 4)), dtype=np.uint8))
                i = at
            else:
                i = min(len(seq), at + int(rng.integers(1, 4)))
        out.append(seq[i:])
        return np.concatenate(out)

    # ---- interspersed repeats
    families = [rng.integers(0, 4, size=int(rng.integers(280, 321)) if k < 16 else int(rng.integers(900, 2001)), dtype=np.uint8) for k in range(24)]
    target = int(0.45 * n)
    placed = 0
    while placed < target:
        fam = families[int(rng.integers(0, len(families)))]
        copy = diverge(fam, float(rng.uniform(0.05, 0.20)))
        if rng.random() < 0.3:  # truncated at its 5' end, as retrotransposed copies are
            copy = copy[int(rng.integers(0, len(copy) // 2)):]
        if rng.random() < 0.5:
            copy = (3 - copy)[::-1]
        at = int(rng.integers(0, n - len(copy)))
        if covered[at:at + len(copy)].any():
            continue
        ref[at:at + len(copy)] = copy
        covered[at:at + len(copy)] = True
        placed += len(copy)
    inter = placed
    # ---- short tandem repeats
    strs = 0
    while strs < int(0.03 * n):
        unit = rng.integers(0, 4, size=int(rng.integers(1, 7)), dtype=np.uint8)
        size = len(unit) * int(rng.integers(5, 41))
        at = int(rng.integers(0, n - size))
        if covered[at:at + size].any():
            continue
        run = np.tile(unit, size // len(unit))
        m = rng.random(size) < 1.0 / 60.0
        run[m] = (run[m] + 1) % 4
        ref[at:at + size] = run
        covered[at:at + size] = True
        strs += size
    # ---- segmental duplications (copied over whatever is there: they carry their repeats with them)
    dup = 0
    for _ in range(segdups):
        size = int(rng.integers(segdup_len[0], segdup_len[1] + 1))
        src, dst = int(rng.integers(0, n - size)), int(rng.integers(0, n - size))
        if abs(src - dst) < size:
            continue
        seg = ref[src:src + size].copy()
        m = rng.random(size) < float(rng.uniform(0.01, 0.02))
        seg[m] = (seg[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) % 4
        ref[dst:dst + size] = seg
        dup += size
    return ref, {"interspersed": inter / float(n), "str": strs / float(n), "segdup": dup / float(n), "families": len(families)}


def make_mapped_reads(ref, records, n, read_len=150, seed=123, err=0.005, n_rate=0.001, indel_err=0.0005, clip_frac=0.03, bad_hint_frac=0.02,
                      region_begin=0):
    """Reads as a mapper hands them over (make_reads draws the clean case): substitutions and Ns as there, plus per base `indel_err`
    single-base insertions / deletions (the read stays read_len long: bases behind the event shift), `clip_frac` of the reads with
    5-30 bases of foreign sequence at one end (a soft clip: the mapper's POS is the first ALIGNED base, so the read's first base lies
    that many positions in front of it -- the hint is made the way gtx_stream_push makes it, POS minus the leading clip), and
    `bad_hint_frac` of the hints wrong: half of them off by 1-40 positions, half somewhere else in the region.  Returns (codes,
    hint pos0, {"indel_reads": k, "clipped": k, "bad_hints": k})."""
    rng = np.random.default_rng(seed)
    codes, pos = make_reads(ref, records, n, read_len=read_len + 8, seed=seed, err=err, n_rate=n_rate, region_begin=region_begin)
    out = codes[:, :read_len].copy()
    # ---- indel errors: one event per affected read
    has = np.nonzero(rng.random(n) < 1.0 - (1.0 - indel_err) ** read_len)[0]
    if len(has):
        at = rng.integers(1, read_len - 1, size=len(has))
        ins = rng.random(len(has)) < 0.5
        j = np.arange(read_len)[None, :]
        src = np.where(ins[:, None], j - (j > at[:, None]), j + (j >= at[:, None]))  # an inserted base shifts what follows back, a deleted one forward
        rows = codes[has[:, None], src]
        new_base = _CODE_OF_BASE[rng.integers(0, 4, size=len(has))]
        rows[ins, at[ins]] = new_base[ins]
        out[has] = rows
    # ---- soft clips
    clipped = np.nonzero(rng.random(n) < clip_frac)[0]
    if len(clipped):
        k = rng.integers(5, 31, size=len(clipped))
        front = rng.random(len(clipped)) < 0.5
        j = np.arange(read_len)[None, :]
        mask = np.where(front[:, None], j < k[:, None], j >= read_len - k[:, None])  # (the other bases keep their places: the hint stays the read's first base)
        junk = _CODE_OF_BASE[rng.integers(0, 4, size=(len(clipped), read_len))]
        out[clipped] = np.where(mask, junk, out[clipped])
    # ---- wrong hints
    hint = pos.copy()
    bad = np.nonzero(rng.random(n) < bad_hint_frac)[0]
    near = bad[: len(bad) // 2]
    hint[near] += rng.integers(1, 41, size=len(near)) * rng.choice([-1, 1], size=len(near))
    far = bad[len(bad) // 2:]
    hint[far] = rng.integers(0, len(ref) - read_len, size=len(far)) + region_begin
    hint = np.clip(hint, region_begin, region_begin + len(ref) - read_len)
    return out, hint, {"indel_reads": int(len(has)), "clipped": int(len(clipped)), "bad_hints": int(len(bad))}
