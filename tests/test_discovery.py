"""Variant discovery, first slice (include/gtx.h: gtx_disc_*): the per-sample first pass over the reads of a region
(run_first_pass, src/typer/caller.cpp:488-1186) -- events read off the CIGARs with their support, the phase counts, the two
support filters.  The oracle (oracle/gto_discovery.hpp) restates the reference's loop; it has no reference vectors (the
reference tests nothing of discovery), so the CPU tests hold it to cases worked out by hand from the reference's text, and
the GPU test holds the device kernel + host bookkeeping to the oracle on simulated alignments."""
import ctypes as C

import numpy as np
import pytest

from graphtyper_amd import lib as gtx
from oracle_lib import lib as olib, _p

CODE = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
OPS = {c: i for i, c in enumerate("MIDNSHP=X")}


def cig(*ops):
    return [(n << 4) | OPS[o] for o, n in ops]


def oracle_first_pass(reference, region_begin, reads, bucket_size=50):
    """reads: list of dict(pos, flag, mapq, cigar [words], seq str, qual [ints])"""
    L = olib()
    L.gto_first_pass.restype = C.c_long
    n = len(reads)
    pos = np.array([r["pos"] for r in reads], np.int32)
    flag = np.array([r["flag"] for r in reads], np.uint16)
    mapq = np.array([r["mapq"] for r in reads], np.uint8)
    cg = np.array([w for r in reads for w in r["cigar"]] + [0], np.uint32)
    cg_off = np.cumsum([0] + [len(r["cigar"]) for r in reads]).astype(np.uint32)
    codes = np.array([CODE[c] for r in reads for c in r["seq"]] + [0], np.uint8)
    qual = np.array([q for r in reads for q in r["qual"]] + [0], np.uint8)
    c_off = np.cumsum([0] + [len(r["seq"]) for r in reads]).astype(np.uint32)
    cap = 1 << 16
    while True:
        out = np.zeros(cap, np.uint32)
        w = L.gto_first_pass(reference.encode(), C.c_long(region_begin), C.c_long(bucket_size), C.c_long(n), _p(pos), _p(flag), _p(mapq), _p(cg),
                             _p(cg_off), _p(codes), _p(qual), _p(c_off), _p(out), C.c_long(cap))
        assert w >= 0, L.gto_last_error()
        if w <= cap:
            return out[:w]
        cap = int(w)


def parse(words):
    """word stream -> [(pos, type, seq, support dict, phase [(pos, type, seq, count)])]"""
    out, i = [], 0
    names = ["hq", "lq", "proper", "first", "reversed", "clipped", "max_mapq", "max_distance", "u1", "u2", "u3", "span", "realign", "good", "log_qual"]

    def ev():
        nonlocal i
        p, t, ln = int(words[i]), chr(int(words[i + 1])), int(words[i + 2])
        s = "".join(chr(int(x)) for x in words[i + 3:i + 3 + ln])
        i += 3 + ln
        return p, t, s
    while i < len(words):
        p, t, s = ev()
        sup = {k: int(np.int32(words[i + j])) for j, k in enumerate(names)}
        nph = int(words[i + 15])
        i += 16
        ph = []
        for _ in range(nph):
            q = ev()
            ph.append(q + (int(words[i]),))
            i += 1
        out.append((p, t, s, sup, ph))
    return out


def _read(pos, seq, cigar, flag=2 | 1, mapq=60, qual=30):
    return dict(pos=pos, flag=flag, mapq=mapq, cigar=cigar, seq=seq, qual=[qual] * len(seq))


def test_a_well_supported_snp_survives_and_a_lone_one_does_not():
    rng = np.random.default_rng(1)
    ref = "".join("ACGT"[i] for i in rng.integers(0, 4, 400))
    rb = 1000
    snp_at = 200
    alt = "ACGT"[("ACGT".index(ref[snp_at]) + 1) % 4]
    reads = []
    for k in range(12):  # 12 reads over the SNP, different starts, both strands, both mates
        start = 120 + 5 * k
        seq = list(ref[start:start + 100])
        seq[snp_at - start] = alt
        reads.append(_read(rb + start, "".join(seq), cig(("M", 100)), flag=1 | 2 | (16 if k % 2 else 0) | (64 if k % 3 else 128)))
    lone = list(ref[100:200])
    lone[50] = "ACGT"[("ACGT".index(lone[50]) + 2) % 4]
    reads.insert(0, _read(rb + 100, "".join(lone), cig(("M", 100))))
    reads.sort(key=lambda r: r["pos"])
    ev = parse(oracle_first_pass(ref, rb, reads))
    assert [(p, t, s) for p, t, s, _, _ in ev] == [(rb + snp_at, "X", alt)]
    sup = ev[0][3]
    assert sup["hq"] == 12 and sup["lq"] == 0 and sup["proper"] == 12 and sup["reversed"] == 6 and sup["first"] == 8
    assert (sup["u1"], sup["u2"], sup["u3"]) == (rb + 120, rb + 125, rb + 130) and sup["max_mapq"] == 60
    assert sup["max_distance"] == max(min(snp_at - (120 + 5 * k), 99 - (snp_at - (120 + 5 * k))) for k in range(12))


def test_span_of_an_insertion_into_a_repeat():
    """bucket.cpp:100-160: the span of a new indel is how far it can be shifted -- an A inserted in front of AAAAAA spans the run"""
    ref = "CGTACGTTGCA" + "AAAAAA" + "CGTGCATGCATTGCAGTCA" * 6
    rb = 0
    ins_at = 11  # in front of the run of six A
    reads = []
    for k in range(10):
        start = k
        seq = ref[start:ins_at] + "A" + ref[ins_at:start + 80]
        reads.append(_read(start, seq, cig(("M", ins_at - start), ("I", 1), ("M", len(seq) - (ins_at - start) - 1)),
                           flag=1 | 2 | (16 if k % 2 else 0) | 64))
    ev = parse(oracle_first_pass(ref, rb, reads))
    ins = [e for e in ev if e[1] == "I"]
    assert len(ins) == 1 and ins[0][:3] == (ins_at, "I", "A")
    assert ins[0][3]["span"] == 6 + 1 and ins[0][3]["hq"] == 10 and ins[0][3]["realign"] == 1


def simulate(seed, n_reads=3000, ref_len=20000, read_len=150):
    """alignments of reads drawn from a diploid sample with SNPs and short indels, as a mapper would report them: CIGARs with
    I / D / soft clips, qualities, pair flags, a few very noisy reads, reads at the region's edges"""
    rng = np.random.default_rng(seed)
    rb = int(rng.choice([0, 5000, 1000000]))
    ref = rng.integers(0, 4, ref_len)
    for _ in range(ref_len // 1500):  # homopolymers and short tandem repeats: indels that can be shifted
        at = int(rng.integers(0, ref_len - 60))
        if rng.random() < 0.5:
            ref[at:at + int(rng.integers(5, 20))] = int(rng.integers(0, 4))
        else:
            unit = rng.integers(0, 4, int(rng.integers(2, 5)))
            ref[at:at + len(unit) * 6] = np.tile(unit, 6)
    ref_s = "".join("ACGT"[i] for i in ref)
    if rng.random() < 0.5:
        ref_s = ref_s[:7000] + "N" * 30 + ref_s[7030:]
    # variants of the sample: (pos, kind, payload), sorted, non-overlapping
    var = []
    p = 200
    while p < ref_len - 300:
        k = rng.random()
        if k < 0.6:
            var.append((p, "X", "ACGT"[(int(ref[p]) + int(rng.integers(1, 4))) % 4]))
        elif k < 0.8:
            var.append((p, "I", "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(1, 9))))))
        else:
            var.append((p, "D", int(rng.integers(1, 9))))
        p += int(rng.integers(60, 500))
    het = rng.random(len(var)) < 0.5
    reads = []
    for _ in range(n_reads):
        hap = int(rng.integers(0, 2))
        start = int(rng.integers(-100, ref_len - 20))
        # walk the reference from `start`, applying the haplotype's variants, until read_len bases are out
        seq, cigar, rp = [], [], max(start, 0)
        pos0 = rp
        ops = []
        vi = next((i for i, v in enumerate(var) if v[0] >= rp), len(var))
        while len(seq) < read_len and rp < ref_len:
            if vi < len(var) and var[vi][0] == rp and (hap == 1 or not het[vi]):
                v = var[vi]
                vi += 1
                if v[1] == "X":
                    seq.append(v[2])
                    ops.append("M")
                    rp += 1
                elif v[1] == "I" and ops:
                    for c in v[2]:
                        seq.append(c)
                        ops.append("I")
                elif v[1] == "D" and ops:
                    ops.extend("D" * v[2])
                    rp += v[2]
                else:
                    seq.append(ref_s[rp])
                    ops.append("M")
                    rp += 1
                continue
            if vi < len(var) and var[vi][0] == rp:
                vi += 1
            seq.append(ref_s[rp])
            ops.append("M")
            rp += 1
        seq = seq[:read_len]
        n_out, cut = 0, len(ops)
        for i, o in enumerate(ops):  # cut the operations where the read ends
            if o != "D":
                n_out += 1
            if n_out == len(seq):
                cut = i + 1
                break
        ops = ops[:cut]
        while ops and ops[-1] == "D":
            ops.pop()
        if len(seq) < 40 or not ops:
            continue
        seq = list(seq)
        noisy = rng.random() < 0.02
        err = rng.random(len(seq)) < (0.12 if noisy else 0.004)
        for i in np.nonzero(err)[0]:
            if seq[i] in "ACGT":
                seq[i] = "ACGT"[("ACGT".index(seq[i]) + int(rng.integers(1, 4))) % 4]
        for i in np.nonzero(rng.random(len(seq)) < 0.002)[0]:
            seq[i] = "N"
        if rng.random() < 0.1:  # soft clip: the first / last bases are not aligned
            n = int(rng.integers(1, 30))
            k = 0
            if rng.random() < 0.5:
                while k < len(ops) and n > 0:  # at the front: M / I become S, D disappear, the position moves
                    if ops[k] in "MI":
                        n -= 1
                        if ops[k] == "M":
                            pos0 += 1
                        ops[k] = "S"
                    else:
                        pos0 += 1
                        ops[k] = ""
                    k += 1
                while k < len(ops) and ops[k] in "DI":
                    if ops[k] == "D":
                        pos0 += 1
                        ops[k] = ""
                    else:
                        ops[k] = "S"
                    k += 1
            else:
                k = len(ops) - 1
                while k >= 0 and n > 0:
                    if ops[k] in "MI":
                        n -= 1
                        ops[k] = "S"
                    else:
                        ops[k] = ""
                    k -= 1
                while k >= 0 and ops[k] in "DI":
                    ops[k] = "" if ops[k] == "D" else "S"
                    k -= 1
            ops = [o for o in ops if o]
        words, prev, cnt = [], None, 0
        for o in ops + [None]:
            if o == prev:
                cnt += 1
            else:
                if prev is not None:
                    words.append((cnt << 4) | OPS[prev])
                prev, cnt = o, 1
        if not any((w & 15) == 0 for w in words):
            continue
        flag = int(rng.choice([0, 1 | 2 | 64, 1 | 2 | 128 | 16, 1 | 64 | 16, 1 | 2 | 128 | 32, 1 | 2 | 64 | 16 | 1024]))
        qual = np.where(rng.random(len(seq)) < 0.15, rng.integers(2, 25, len(seq)), rng.integers(25, 41, len(seq))).astype(int).tolist()
        reads.append(dict(pos=rb + pos0, flag=flag, mapq=int(rng.choice([60, 60, 60, 37, 12, 0, 255])), cigar=words if rng.random() > 0.01 else [],
                          seq="".join(seq), qual=qual))
    reads.sort(key=lambda r: r["pos"])
    if rng.random() < 0.5:  # a read in front of the region and one behind its end
        reads.insert(0, _read(max(rb - 40, 0) if rb else 0, ref_s[:60], cig(("M", 60))))
        if rb:
            reads[0]["pos"] = rb - 40
        reads.append(_read(rb + ref_len, "ACGT" * 20, cig(("M", 80))))
        reads.append(_read(rb + ref_len - 50, ref_s[-50:], cig(("M", 50))))
    return ref_s, rb, reads


def test_oracle_runs_over_simulated_alignments():
    ref, rb, reads = simulate(3)
    ev = parse(oracle_first_pass(ref, rb, reads))
    kinds = [t for _, t, _, _, _ in ev]
    assert kinds.count("X") > 10 and kinds.count("I") > 2 and kinds.count("D") > 2
    assert any(e[3]["span"] > 1 for e in ev if e[1] != "X") and any(e[4] for e in ev)
    assert oracle_first_pass(ref, rb, reads, bucket_size=50).tolist() == oracle_first_pass(ref, rb, reads, bucket_size=777).tolist()


def oracle_full(reference, region_begin, reads, bucket_size=50, file_i=0):
    """the pass to its end through the oracle -> (result words, events, read states): the last two in the product's layouts"""
    L = olib()
    L.gto_first_pass_full.restype = C.c_long
    n = len(reads)
    pos = np.array([r["pos"] for r in reads], np.int32)
    flag = np.array([r["flag"] for r in reads], np.uint16)
    mapq = np.array([r["mapq"] for r in reads], np.uint8)
    cg = np.array([w for r in reads for w in r["cigar"]] + [0], np.uint32)
    cg_off = np.cumsum([0] + [len(r["cigar"]) for r in reads]).astype(np.uint32)
    codes = np.array([CODE[c] for r in reads for c in r["seq"]] + [0], np.uint8)
    qual = np.array([q for r in reads for q in r["qual"]] + [0], np.uint8)
    c_off = np.cumsum([0] + [len(r["seq"]) for r in reads]).astype(np.uint32)
    cap, ev_cap = 1 << 16, 64 * max(n, 1)
    while True:
        out = np.zeros(cap, np.uint32)
        events = np.zeros(ev_cap, gtx.DISC_EVENT)
        read_out = np.zeros(max(n, 1), gtx.DISC_READ_OUT)
        n_ev = C.c_long()
        w = L.gto_first_pass_full(reference.encode(), C.c_long(region_begin), C.c_long(bucket_size), C.c_long(file_i), C.c_long(n), _p(pos), _p(flag), _p(mapq),
                                  _p(cg), _p(cg_off), _p(codes), _p(qual), _p(c_off), _p(out), C.c_long(cap), _p(events), C.c_long(ev_cap), C.byref(n_ev),
                                  _p(read_out))
        if w == -2:
            ev_cap = int(n_ev.value)
            continue
        assert w >= 0, L.gto_last_error()
        if w <= cap:
            return out[:w], events[:n_ev.value], read_out[:n]
        cap = int(w)


def oracle_merge(a, b):
    L = olib()
    L.gto_disc_merge.restype = C.c_long
    a, b = np.ascontiguousarray(a, np.uint32), np.ascontiguousarray(b, np.uint32)
    out = np.zeros(len(a) + len(b) + 16, np.uint32)
    w = L.gto_disc_merge(_p(a), C.c_long(len(a)), _p(b), C.c_long(len(b)), _p(out), C.c_long(len(out)))
    assert 0 <= w <= len(out), L.gto_last_error()
    return out[:w]


def _host_inputs(reads):
    n = len(reads)
    stride = max(16, (max(len(r["seq"]) for r in reads) + 31) // 32 * 16)
    codes = np.zeros((n, stride * 2), np.uint8)
    dr = np.zeros(n, gtx.DISC_READ)
    cg = []
    for i, r in enumerate(reads):
        codes[i, :len(r["seq"])] = [CODE[c] for c in r["seq"]]
        dr[i] = (r["pos"], r["flag"], r["mapq"], 0, len(r["seq"]), len(r["cigar"]), len(cg))
        cg.extend(r["cigar"])
    return dr, np.array(cg + [0], np.uint32), gtx.pack_nibbles(codes, stride=stride), stride


def product_host_stage(reference, region_begin, reads, events, read_out, bucket_size=50, file_i=None, handle=None):
    """gtx_disc_first_pass (file_i None) or gtx_disc_first_pass_haplotypes over events made elsewhere (a device, or the oracle)"""
    L = gtx.lib()
    dr, cg, nib, stride = _host_inputs(reads)
    h = handle
    if h is None:
        h = C.c_void_p()
        gtx.check(L.gtx_disc_create(reference.encode(), len(reference), region_begin, -1, C.byref(h)))
    events, read_out = np.ascontiguousarray(events, gtx.DISC_EVENT), np.ascontiguousarray(read_out, gtx.DISC_READ_OUT)
    n_words, words = C.c_uint64(), np.zeros(1 << 16, np.uint32)
    while True:
        if file_i is None:
            rc = L.gtx_disc_first_pass(h, _p(dr), _p(cg), _p(read_out), len(reads), _p(events), len(events), _p(nib), stride, bucket_size, _p(words), len(words),
                                       C.byref(n_words))
        else:
            rc = L.gtx_disc_first_pass_haplotypes(h, _p(dr), _p(cg), _p(read_out), len(reads), _p(events), C.c_uint64(len(events)), _p(nib), stride, bucket_size,
                                                  C.c_int32(file_i), _p(words), C.c_uint64(len(words)), C.byref(n_words))
        if rc == 5 and n_words.value > len(words):
            words = np.zeros(int(n_words.value), np.uint32)
            continue
        break
    if handle is None:
        L.gtx_disc_destroy(h)
    assert rc == 0, gtx.lib().gtx_last_error()
    return words[:n_words.value]


def product_merge(a, b):
    L = gtx.lib()
    a, b = np.ascontiguousarray(a, np.uint32), np.ascontiguousarray(b, np.uint32)
    out, n = np.zeros(len(a) + len(b) + 16, np.uint32), C.c_uint64()
    gtx.check(L.gtx_disc_merge(_p(a), C.c_uint64(len(a)), _p(b), C.c_uint64(len(b)), _p(out), C.c_uint64(len(out)), C.byref(n)))
    return out[:n.value]


def parse_result(words):
    """result words -> (indels [(pos, type, seq)], {event: (ever set, always set)})"""
    i = 0

    def ev():
        nonlocal i
        p, t, ln = int(words[i]), chr(int(words[i + 1])), int(words[i + 2])
        s = "".join(chr(int(x)) for x in words[i + 3:i + 3 + ln])
        i += 3 + ln
        return (p, t, s)
    indels = []
    n = int(words[i]); i += 1
    for _ in range(n):
        e = ev()
        i += 16
        np_ = int(words[i]); i += 1
        for _ in range(np_):
            ev(); i += 1
        indels.append(e)
    haps = {}
    n = int(words[i]); i += 1
    for _ in range(n):
        e = ev()
        sets = []
        for _ in range(2):
            m = int(words[i]); i += 1
            sets.append({ev() for _ in range(m)})
        haps[e] = tuple(sets)
    assert i == len(words)
    return indels, haps


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_host_stages_equal_the_oracle(seed):
    """the product's host stages over the events the oracle's walk makes (no device): the state behind the two filters, the pass to
    its end (haplotype map, indels), and the merge of three files' results in their order"""
    ref, rb, reads = simulate(seed, n_reads=3000 if seed % 2 else 1200, read_len=150 if seed != 4 else 250)
    want_full, events, read_out = oracle_full(ref, rb, reads, file_i=0)
    assert np.array_equal(product_host_stage(ref, rb, reads, events, read_out), oracle_first_pass(ref, rb, reads))
    got_full = product_host_stage(ref, rb, reads, events, read_out, file_i=0)
    assert np.array_equal(got_full, want_full)
    indels, haps = parse_result(got_full)
    assert len(haps) > 20 and any(ever for ever, _ in haps.values())
    assert all(t != "X" for _, t, _ in indels) and all(always <= ever for ever, always in haps.values())
    # three "files": the same region seen by other reads; merged in order
    acc_o, acc_p = np.zeros(0, np.uint32), np.zeros(0, np.uint32)
    for f in range(3):
        _, _, rf = simulate(seed if f == 0 else 100 * seed + f, n_reads=1500, read_len=150)
        wf, ev_f, ro_f = oracle_full(ref, rb, rf, file_i=f)
        pf = product_host_stage(ref, rb, rf, ev_f, ro_f, file_i=f)
        assert np.array_equal(pf, wf)
        acc_o, acc_p = oracle_merge(acc_o, wf), product_merge(acc_p, pf)
        assert np.array_equal(acc_p, acc_o)
    indels, haps = parse_result(acc_p)
    assert len(haps) > 20
    with pytest.raises(gtx.GtxError):
        product_merge(acc_p[:-1], acc_p)  # a cut stream is refused


def test_two_snps_of_one_haplotype_travel_together():
    """caller.cpp:1186-1365 worked by hand: 12 reads carry two SNPs five and sixty positions apart on one haplotype, 12 other reads
    carry neither.  Every read that covers a pair of them shows both: support / coverage / support_ratio = 12 / 24 / 0.5 = 1 > 0.78
    -> "ever together" for (first, second) and (first, third), (second, third); "always" only for the pair within ten positions.
    Merged with a second sample that has the first SNP alone, its "always" set is emptied (the intersection), "ever" stays."""
    rng = np.random.default_rng(7)
    ref = "".join("ACGT"[i] for i in rng.integers(0, 4, 600))
    rb, at = 2000, (300, 305, 360)
    alts = ["ACGT"[("ACGT".index(ref[p]) + 1) % 4] for p in at]

    def sample(carry, seed):
        reads = []
        for k in range(24):
            start = 230 + 2 * k
            seq = list(ref[start:start + 150])
            if k % 2 == 0:
                for p, a in zip(at, alts):
                    if p in carry:
                        seq[p - start] = a
            reads.append(_read(rb + start, "".join(seq), cig(("M", 150)), flag=1 | 2 | (16 if (k // 2) % 2 else 0) | (64 if k % 3 else 128)))
        return reads
    one = sample(set(at), 1)
    words, events, read_out = oracle_full(ref, rb, one, file_i=0)
    got = product_host_stage(ref, rb, one, events, read_out, file_i=0)
    assert np.array_equal(got, words)
    indels, haps = parse_result(got)
    e = [(rb + p, "X", a) for p, a in zip(at, alts)]
    assert indels == [] and set(haps) == set(e)
    assert haps[e[0]] == ({e[1], e[2]}, {e[1]}) and haps[e[1]] == ({e[2]}, set()) and haps[e[2]] == (set(), set())
    two = sample({at[0]}, 2)
    w2, ev2, ro2 = oracle_full(ref, rb, two, file_i=1)
    p2 = product_host_stage(ref, rb, two, ev2, ro2, file_i=1)
    assert np.array_equal(p2, w2) and parse_result(p2)[1] == {e[0]: (set(), set())}
    merged = product_merge(got, p2)
    assert np.array_equal(merged, oracle_merge(words, w2))
    assert parse_result(merged)[1] == {e[0]: ({e[1], e[2]}, set()), e[1]: ({e[2]}, set()), e[2]: (set(), set())}
    # the other order: the second sample's lone SNP first, then the sample with all three -- new events keep of their "always" what the
    # accumulated map has not seen (the first SNP has been seen: it is not in anybody's set anyway), the known one takes the intersection
    other = product_merge(p2, got)
    assert np.array_equal(other, oracle_merge(w2, words)) and parse_result(other)[1] == parse_result(merged)[1]


def test_no_device_no_events():
    h = C.c_void_p()
    gtx.check(gtx.lib().gtx_disc_create(b"ACGTACGT", 8, 0, -1, C.byref(h)))
    d = np.zeros(64, np.uint8)
    assert gtx.lib().gtx_disc_events_batch(h, _p(d), 16, _p(d), 32, _p(d), _p(d), 1, _p(d), 1, _p(d), _p(d), None) == 2  # GTX_ERR_NO_DEVICE
    gtx.lib().gtx_disc_destroy(h)


def product_first_pass(reference, region_begin, reads, bucket_size=50, event_cap=None, with_haplotypes=False):
    import torch
    L = gtx.lib()
    n = len(reads)
    stride = max(16, (max(len(r["seq"]) for r in reads) + 31) // 32 * 16)
    codes = np.zeros((n, stride * 2), np.uint8)
    qual = np.zeros((n, stride * 2), np.uint8)
    dr = np.zeros(n, gtx.DISC_READ)
    cg = []
    for i, r in enumerate(reads):
        codes[i, :len(r["seq"])] = [CODE[c] for c in r["seq"]]
        qual[i, :len(r["seq"])] = r["qual"]
        dr[i] = (r["pos"], r["flag"], r["mapq"], 0, len(r["seq"]), len(r["cigar"]), len(cg))
        cg.extend(r["cigar"])
    cg = np.array(cg + [0], np.uint32)
    nib = gtx.pack_nibbles(codes, stride=stride)
    planes = gtx.pack_planes(nib, stride)
    h = C.c_void_p()
    gtx.check(L.gtx_disc_create(reference.encode(), len(reference), region_begin, 0, C.byref(h)))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to("cuda:0")  # noqa: E731
    d_planes, d_qual, d_reads, d_cigar = dev(planes), dev(qual), dev(dr), dev(cg)
    cap = event_cap or 64 * n
    d_events = torch.zeros(cap * gtx.DISC_EVENT.itemsize, dtype=torch.uint8, device="cuda:0")
    d_counts = torch.zeros(2, dtype=torch.int32, device="cuda:0")
    d_out = torch.zeros(n * gtx.DISC_READ_OUT.itemsize, dtype=torch.uint8, device="cuda:0")
    gtx.check(L.gtx_disc_events_batch(h, d_planes.data_ptr(), stride, d_qual.data_ptr(), stride * 2, d_reads.data_ptr(), d_cigar.data_ptr(), n,
                                      d_events.data_ptr(), cap, d_counts.data_ptr(), d_out.data_ptr(), None))
    torch.cuda.synchronize()
    counts = d_counts.cpu().numpy()
    events = d_events.cpu().numpy().view(gtx.DISC_EVENT)
    read_out = d_out.cpu().numpy().view(gtx.DISC_READ_OUT)
    n_words = C.c_uint64()
    words = np.zeros(1 << 16, np.uint32)
    while True:
        rc = L.gtx_disc_first_pass(h, _p(dr), _p(cg), _p(read_out), n, _p(events), min(int(counts[0]), cap), _p(nib), stride, bucket_size, _p(words),
                                   len(words), C.byref(n_words))
        if rc == 5 and n_words.value > len(words):
            words = np.zeros(int(n_words.value), np.uint32)
            continue
        break
    full = None
    if rc == 0 and with_haplotypes:  # the pass to its end over the same device events (the handle has the reference)
        full = product_host_stage(reference, region_begin, reads, events[:min(int(counts[0]), cap)], read_out, bucket_size, file_i=3, handle=h)
    L.gtx_disc_destroy(h)
    return (rc, words[:n_words.value], counts, full) if with_haplotypes else (rc, words[:n_words.value], counts)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_device_first_pass_equals_the_oracle(seed):
    ref, rb, reads = simulate(seed, n_reads=6000 if seed % 2 else 2500, read_len=150 if seed != 4 else 250)
    want = oracle_first_pass(ref, rb, reads)
    rc, got, counts, full = product_first_pass(ref, rb, reads, with_haplotypes=True)
    assert rc == 0 and counts[1] == 0 and counts[0] > len(reads) // 4
    assert np.array_equal(full, oracle_full(ref, rb, reads, file_i=3)[0])  # ... and the pass to its end: indels + haplotype map
    assert len(got) == len(want) and np.array_equal(got, want), "first differing word %s" % np.nonzero(got[:min(len(got), len(want))] != want[:min(len(got), len(want))])[0][:5]
    assert len(parse(got)) > 20
    if seed == 1:  # an event buffer that is too small is reported, not silently cut
        rc, _, counts = product_first_pass(ref, rb, reads, event_cap=100)
        assert counts[1] > 0 and rc == 5
