"""The read pre-filter (gtx_bam_shrink, graphtyper_amd/csrc/gtx_shrink.inl) against the oracle's restatement of the reference's
bamshrink (oracle/gto_shrink.hpp <- /root/reference/src/utilities/bamshrink.cpp:64-1045): the record stream of the output file,
byte for byte, on random coordinate-sorted BAM files that reach every branch (pairs, single reads, unmapped mates, mates on
other contigs, adapters, soft / hard clips, indels, Ns at the ends, AS / XS / WS scores in every integer type, full bins)."""
import ctypes as C
import gzip
import os
import struct

import numpy as np
import pytest

import bam_writer as bw
import oracle_lib
from graphtyper_amd import lib as gtx

REFS = [("chr1", 200000), ("chr2", 150000), ("chr3", 50000)]
HEADER = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in REFS) + "@RG\tID:rg1\tSM:s1\n@RG\tID:rg2\tSM:s1\n@PG\tID:x\n"


def _olib():
    L = oracle_lib.lib()
    L.gto_bam_shrink.restype = C.c_long
    L.gto_bam_shrink.argtypes = [C.c_char_p, C.c_long, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_long), C.c_double, C.c_int, C.POINTER(C.c_long),
                                 C.c_char_p, C.c_long]
    L.gto_shrink_header.restype = C.c_long
    L.gto_shrink_header.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_long]
    return L


def oracle_shrink(stream, intervals, p):
    """intervals: [(tid, begin, end)]; p: gtx ShrinkParams.  The reference's bamshrink(): SUPER_HI_DEPTH 1000 without a coverage."""
    L = _olib()
    opts = (C.c_long * 11)(p.max_frag_len, p.min_num_matching, p.filter_mapq0, p.no_filter_on_coverage, p.min_read_len, p.min_read_len_low_mapq,
                           p.min_unpaired_read_len, p.as_filter_threshold, 2 if p.avg_cov_by_readlen > 0 else 1000, p.sam_flag_filter,
                           p.change_read_names)
    cov = p.avg_cov_by_readlen if p.avg_cov_by_readlen > 0 else 0.30000001
    read_num = C.c_long(0)
    out = b""
    buf = C.create_string_buffer(len(stream) + 1024)
    for tid, b, e in intervals:
        n = L.gto_bam_shrink(stream, len(stream), tid, b, e, opts, cov, int(len(intervals) == 1), C.byref(read_num), buf, len(buf))
        assert n >= 0, n
        out += buf.raw[:n]
    return out


def split_bam(path):
    """(header text, [(name, length)], record stream) of a BAM file"""
    data = gzip.decompress(open(path, "rb").read())
    assert data[:4] == b"BAM\1"
    l_text, = struct.unpack_from("<i", data, 4)
    text = data[8:8 + l_text].decode()
    at = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, at)
    at += 4
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", data, at)
        name = data[at + 4:at + 4 + l_name - 1].decode()
        length, = struct.unpack_from("<i", data, at + 4 + l_name)
        refs.append((name, length))
        at += 8 + l_name
    return text, refs, data[at:]


def parse_records(stream):
    out, at = [], 0
    while at < len(stream):
        block, = struct.unpack_from("<i", stream, at)
        b = stream[at + 4:at + 4 + block]
        tid, pos, l_name, mapq, bin_, n_cigar, flag, l_seq, mtid, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", b, 0)
        name = b[32:32 + l_name - 1].decode("latin1")
        o = 32 + l_name
        cigar = [(w >> 4, "MIDNSHP=X"[w & 15]) for w in struct.unpack_from("<%dI" % n_cigar, b, o)]
        o += 4 * n_cigar
        seq = b[o:o + (l_seq + 1) // 2]
        o += (l_seq + 1) // 2
        qual = b[o:o + l_seq]
        out.append(dict(tid=tid, pos=pos, name=name, mapq=mapq, bin=bin_, flag=flag, l_seq=l_seq, mtid=mtid, mpos=mpos, tlen=tlen, cigar=cigar,
                        seq=seq, qual=qual, aux=b[o + l_seq:]))
        at += 4 + block
    return out


def _cigar(rng, length, kind):
    """a CIGAR that consumes `length` read bases (hard clips aside)"""
    if kind == 0:
        return [("M", length)]
    if kind == 1:  # soft clips
        a, b = int(rng.integers(0, 20)), int(rng.integers(0, 20))
        out = ([("S", a)] if a else []) + [("M", length - a - b)] + ([("S", b)] if b else [])
        return out
    if kind == 2:  # an insertion and a deletion
        a = int(rng.integers(10, length - 30))
        i, d = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        return [("M", a), ("I", i), ("M", 10), ("D", d), ("M", length - a - i - 10)]
    if kind == 3:  # hard clips around it
        return [("H", 5), ("M", length), ("H", 7)]
    if kind == 4:  # a deletion right behind the first bases (in front once Ns or adapters are cut)
        a = int(rng.integers(1, 4))
        return [("M", a), ("D", 3), ("M", length - a)]
    a, b = int(rng.integers(20, min(60, length - 20))), int(rng.integers(0, 8))  # heavily clipped
    return [("S", a), ("M", length - a - b)] + ([("S", b)] if b else [])


def _ref_span(cigar):
    return sum(n for op, n in cigar if op in "MDN=X")


def _aux(rng, p_rg=0.9):
    out = [("RG", "Z", "rg1" if rng.random() < 0.5 else "rg2")] if rng.random() < p_rg else []
    if rng.random() < 0.3:
        out.insert(0, ("NM", "C", int(rng.integers(0, 5))))
    if rng.random() < 0.3:
        out.append(("MD", "Z", "100"))
    if rng.random() < 0.8:
        typ = "cCsSiI"[int(rng.integers(0, 6))]
        out.append(("AS", typ, int(rng.integers(40, 127))))
        if rng.random() < 0.7:
            out.append(("XS", "cCsSiI"[int(rng.integers(0, 6))], int(rng.integers(0, 127))))
        if rng.random() < 0.2:
            out.append(("WS", "C", int(rng.integers(40, 127))))
    if rng.random() < 0.1:
        out.append(("XT", "A", "U"))
    if rng.random() < 0.1:
        out.append(("XF", "f", 1.5))
    if rng.random() < 0.1 and p_rg <= 1:
        out.append(("XB", "B", ("c", [1, 2, 3])))  # an array ends the reference's walk: what follows is not looked at
        out.append(("XS", "C", 120))
    rng.shuffle(out)
    return out


def random_file(path, seed, n_pairs=900, deep=False, index=True, p_rg=0.9):
    rng = np.random.default_rng(seed)
    recs = []  # (tid, pos, order, bytes, end)

    def codes(length, n_front=0, n_back=0):
        c = rng.choice(np.array([1, 2, 4, 8], np.uint8), length)
        if rng.random() < 0.05:
            c[int(rng.integers(0, length))] = int(rng.choice([3, 5, 15]))
        c[:n_front] = 15
        if n_back:
            c[length - n_back:] = 15
        return c

    def quals(length):
        r = rng.random()
        if r < 0.8:
            return rng.integers(15, 41, length)
        if r < 0.9:
            return rng.integers(2, 25, length)
        return np.full(length, 2)

    def mapq():
        r = rng.random()
        return 60 if r < 0.7 else int(rng.integers(0, 61))

    def add(name, flag, tid, pos, mq, cigar, mtid, mpos, tlen, length, n_front=0, n_back=0):
        end = pos + (_ref_span(cigar) if not flag & 4 and _ref_span(cigar) else 1)
        recs.append((tid if tid >= 0 else 1 << 30, pos, len(recs), bw.record(name, flag, tid, pos, mq, cigar, mtid, mpos, tlen, codes(length, n_front, n_back),
                                                                             aux=_aux(rng, p_rg), qual=quals(length)), (tid, pos, end)))

    for k in range(n_pairs):
        name = "read%d" % k if rng.random() < 0.9 else "r\xe9%d" % k  # (a name with a byte above 127: the hash takes chars as signed)
        tid = 0 if rng.random() < 0.8 else int(rng.integers(0, 3))
        span = 4000 if deep else 30000
        pos = int(rng.integers(9000, 9000 + span))
        length = int(rng.choice([151, 151, 151, 120, 100, 80, 60]))
        kind = rng.random()
        nf = int(rng.integers(1, 6)) if rng.random() < 0.06 else 0
        nb = int(rng.integers(1, 6)) if rng.random() < 0.06 else 0
        if kind < 0.62:  # a pair
            r = rng.random()
            frag = int(rng.integers(length - 40, length + 10)) if r < 0.12 else int(rng.integers(200, 700)) if r < 0.95 else int(rng.integers(900, 1500))
            c1, c2 = _cigar(rng, length, int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5]))), _cigar(rng, length, int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5])))
            p2 = max(pos + frag - _ref_span(c2), pos - 5 if rng.random() < 0.3 else pos)
            tlen = p2 + _ref_span(c2) - pos
            mq1, mq2 = mapq(), mapq()
            f1, f2 = 1 | 2 | 32 | 64, 1 | 2 | 16 | 128
            if rng.random() < 0.04:  # the same strand
                f1 &= ~32
                f2 &= ~16
            if rng.random() < 0.03:  # a duplicate / a secondary alignment
                f2 |= int(rng.choice([1024, 256, 2048, 512]))
            if rng.random() < 0.03:  # the insert size's sign against the strand
                tlen = -tlen
            add(name, f1, tid, pos, mq1, c1, tid, p2, tlen, length, nf, nb)
            if rng.random() < 0.95:  # (else: the mate is missing)
                add(name, f2, tid, p2, mq2, c2, tid, pos, -tlen, length, nb, nf)
        elif kind < 0.72:  # mate unmapped: both records at the mapped read's position
            c1 = _cigar(rng, length, int(rng.choice([0, 0, 1])))
            rev = rng.random() < 0.5
            add(name, 1 | 8 | 64 | (16 if rev else 0), tid, pos, mapq(), c1, tid, pos, 0, length, nf, nb)
            add(name, 1 | 4 | 128 | (32 if rev else 0), tid, pos, 0, [], tid, pos, 0, length, nb, nf)
        elif kind < 0.8:  # mate on another contig
            c1 = _cigar(rng, length, 0)
            add(name, 1 | 32 | 64, tid, pos, mapq(), c1, (tid + 1) % 3, int(rng.integers(0, 40000)), 0, length)
        elif kind < 0.95:  # single read
            c1 = _cigar(rng, length, int(rng.choice([0, 0, 1, 2, 3, 4])))
            add(name, 16 if rng.random() < 0.5 else 0, tid, pos, mapq(), c1, -1, -1, 0, length, nf, nb)
        else:  # unmapped, unplaced
            add(name, 4, -1, -1, 0, [], -1, -1, 0, length)
    recs.sort(key=lambda r: (r[0], r[1], r[2]))
    bw.write_bam(path, REFS, HEADER, [r[3] for r in recs], index=[r[4] for r in recs] if index else None)
    return b"".join(r[3] for r in recs)


CASES = [
    # (seed, pairs, deep, intervals [(tid, begin, end)], parameter overrides)
    (1, 900, False, [(0, 12000, 32000)], {}),
    (2, 900, False, [(0, 9500, 20000)], dict(avg_cov_by_readlen=0.2)),
    (3, 1500, True, [(0, 9000, 14000)], dict(avg_cov_by_readlen=0.05)),             # full bins
    (4, 1500, True, [(0, 9000, 14000)], dict(avg_cov_by_readlen=0.03)),             # ... and "super high depth"
    (5, 1500, True, [(0, 9000, 14000)], dict(no_filter_on_coverage=1)),
    (6, 900, False, [(0, 10000, 15000), (0, 30000, 36000), (1, 9000, 30000)], {}),    # several intervals: the header is copied
    (7, 900, False, [(1, 9000, 39000)], dict(change_read_names=0)),
    (8, 900, False, [(0, 12000, 32000)], dict(filter_mapq0=0, min_read_len=50, min_read_len_low_mapq=60, min_unpaired_read_len=60, min_num_matching=30)),
    (9, 900, False, [(0, 12000, 32000)], dict(max_frag_len=400, as_filter_threshold=10)),
    (10, 900, False, [(2, 0, 49999)], dict(sam_flag_filter=0)),
    (11, 400, False, [(0, 100000, 120000)], {}),                                     # nothing there
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "seed%d" % c[0])
def test_output_equals_the_oracles(tmp_path, case):
    seed, n_pairs, deep, intervals, over = case
    path = str(tmp_path / "in.bam")
    stream = random_file(path, seed, n_pairs, deep)
    par = gtx.shrink_params(**over)
    out = str(tmp_path / "out.bam")
    stats = gtx.bam_shrink(path, [(REFS[t][0], b, e) for t, b, e in intervals], out, par)
    text, refs, got = split_bam(out)
    want = oracle_shrink(stream, intervals, par)
    if got != want:
        g, w = parse_records(got), parse_records(want)
        for k, (a, b) in enumerate(zip(g, w)):
            assert a == b, "record %d" % k
        assert len(g) == len(w)
    assert got == want
    n_out = len(parse_records(got))
    assert stats["records_written"] == n_out
    if seed not in (4, 11):
        assert n_out > 20
    if len(intervals) == 1:
        L = _olib()
        buf = C.create_string_buffer(len(HEADER) + 16)
        n = L.gto_shrink_header(HEADER.encode(), REFS[intervals[0][0]][0].encode(), buf, len(buf))
        assert text == buf.raw[:n].decode()
        assert refs == [REFS[intervals[0][0]]]
        assert all(r["tid"] == 0 for r in parse_records(got))
    else:
        assert text == HEADER and refs == REFS


def test_what_the_filter_does_to_the_records(tmp_path):
    """properties of the output that follow from the reference's text, whatever the oracle says"""
    path = str(tmp_path / "in.bam")
    hard_clipped = {r["name"] for r in parse_records(random_file(path, 21, 1200, False)) if any(op == "H" for _, op in r["cigar"])}
    out = str(tmp_path / "out.bam")
    stats = gtx.bam_shrink(path, [("chr1", 12000, 32000)], out, gtx.shrink_params(change_read_names=0))
    _, _, stream = split_bam(out)
    recs = parse_records(stream)
    assert stats["records_written"] == len(recs) > 100
    assert [r["pos"] for r in recs] == sorted(r["pos"] for r in recs)                  # sorted by begin position
    names = {}
    for r in recs:
        assert set(r["qual"]) <= {30, 11}                                               # binarizeQual
        assert all(op != "H" for _, op in r["cigar"][:1] + r["cigar"][-1:]) or len(r["cigar"]) == 1
        assert r["l_seq"] >= 75 and not (r["flag"] & 3840)
        if not r["flag"] & 4 and r["cigar"] and r["name"] not in hard_clipped:
            # the CIGAR still describes the bases (with hard clips in front of Ns the reference cuts the clip instead: :423-482)
            assert sum(n for n, op in r["cigar"] if op in "MIS=X") == r["l_seq"]
        tags, at = [], 0
        while at < len(r["aux"]):
            tag, typ = r["aux"][at:at + 2].decode(), chr(r["aux"][at + 2])
            tags.append(tag)
            at += 3 + ({"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[typ] if typ != "Z" else r["aux"].index(b"\0", at + 3) - at - 2)
        assert set(tags) <= {"RG", "AS", "XS", "WS"}
        names.setdefault(r["name"], []).append(r)
    assert all(len(rs) <= 2 for rs in names.values())
    singles = [r for r in recs if not r["flag"] & 1]
    assert singles and all(r["mtid"] == -1 and r["mpos"] == -1 and not r["flag"] & (2 | 8 | 32) for r in singles)


def test_the_output_is_read_back_by_the_ingest(tmp_path):
    path = str(tmp_path / "in.bam")
    random_file(path, 22, 600, False, p_rg=1.1)
    out = str(tmp_path / "s1.bam")
    stats = gtx.bam_shrink(path, [("chr1", 12000, 32000)], out)
    reads = gtx.Reads([out])
    assert reads.samples == ["s1"]
    n = 0
    while True:
        recs, seq = reads.next(256, seq_stride=80)
        if len(recs) == 0:
            break
        n += len(recs)
    reads.close()
    assert n == stats["records_written"] > 0


def test_without_an_index_the_file_is_scanned(tmp_path):
    a, b = str(tmp_path / "a.bam"), str(tmp_path / "b.bam")
    random_file(a, 23, 500, False, index=True)
    random_file(b, 23, 500, False, index=False)
    gtx.bam_shrink(a, [("chr1", 15000, 25000)], a + ".out")
    gtx.bam_shrink(b, [("chr1", 15000, 25000)], b + ".out")
    assert split_bam(a + ".out") == split_bam(b + ".out")


def test_bad_arguments_are_refused(tmp_path):
    path = str(tmp_path / "in.bam")
    random_file(path, 24, 50, False)
    with pytest.raises(Exception):
        gtx.bam_shrink(path, [("chrX", 0, 100)], str(tmp_path / "o.bam"))
    with pytest.raises(Exception):
        gtx.bam_shrink(str(tmp_path / "missing.bam"), [("chr1", 0, 100)], str(tmp_path / "o.bam"))
    with pytest.raises(Exception):
        gtx.bam_shrink(path, [], str(tmp_path / "o.bam"))
    open(str(tmp_path / "junk.bam"), "wb").write(b"not a bam file at all")
    with pytest.raises(Exception):
        gtx.bam_shrink(str(tmp_path / "junk.bam"), [("chr1", 0, 100)], str(tmp_path / "o.bam"))
    assert not os.path.exists(str(tmp_path / "o.bam"))


def test_short_names_and_the_name_hash():
    """decimal_to_read_name_string (bamshrink.cpp:34-61) worked by hand: digits '!'..'?' then 'A'..'~', lowest first"""
    # through the product: 95 single reads get the names of 0..94
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        recs = []
        for k in range(95):
            recs.append(bw.record("n%d" % k, 0, 0, 1000 + 2 * k, 60, [("M", 100)], -1, -1, 0, np.full(100, 1, np.uint8), qual=np.full(100, 30)))
        bw.write_bam(d + "/in.bam", REFS, HEADER, recs)
        gtx.bam_shrink(d + "/in.bam", [("chr1", 0, 5000)], d + "/out.bam", gtx.shrink_params(no_filter_on_coverage=1))
        names = [r["name"] for r in parse_records(split_bam(d + "/out.bam")[2])]
    assert names[0] == "!" and names[30] == "?" and names[31] == "A" and names[92] == "~" and names[93] == "!\"" and names[94] == "\"\""


def test_intervals_from_a_file(tmp_path):
    """gtx_bam_shrink_multi (bamshrink_multi + readIntervals, bamshrink.cpp:1047-1130, 1352-1371): 1-based lines, neighbours
    closer than 2 x maxFragLen are one interval, unsorted or empty files are refused"""
    path = str(tmp_path / "in.bam")
    stream = random_file(path, 31, 900, False)
    iv = str(tmp_path / "iv.txt")
    open(iv, "w").write("chr1 10001 12000\nchr1 13001 15000\nchr1 30001 36000\nchr2 9001 30000\n")  # the first two merge (1000 apart)
    par = gtx.shrink_params()
    out = str(tmp_path / "out.bam")
    gtx.bam_shrink_multi(path, iv, out, par)
    text, refs, got = split_bam(out)
    assert got == oracle_shrink(stream, [(0, 10000, 14999), (0, 30000, 35999), (1, 9000, 29999)], par) and len(parse_records(got)) > 50
    assert text == HEADER and refs == REFS
    open(iv, "w").write("chr1 10001 12000\n")  # one interval: the header keeps that contig only
    gtx.bam_shrink_multi(path, iv, out, par)
    text, refs, got = split_bam(out)
    assert refs == [REFS[0]] and got == oracle_shrink(stream, [(0, 10000, 11999)], par)
    # the reference asks the stream for its end before it uses an interval: without a newline behind it the last line is left out
    open(iv, "w").write("chr1 10001 12000\nchr2 9001 30000")
    gtx.bam_shrink_multi(path, iv, out, par)
    assert split_bam(out)[2] == oracle_shrink(stream, [(0, 10000, 11999)], par)
    for bad in ("chr1 20001 22000\nchr1 10001 12000\n", "", "chr1 x y\n"):
        open(iv, "w").write(bad)
        with pytest.raises(gtx.GtxError):
            gtx.bam_shrink_multi(path, iv, out, par)
    with pytest.raises(gtx.GtxError):
        gtx.bam_shrink_multi(path, str(tmp_path / "missing.txt"), out, par)
