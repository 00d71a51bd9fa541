"""The coordinate index of a bgzip VCF (gtx_tabix_build / gtx_tabix_start, graphtyper_amd/csrc/gtx_tabix.cpp; the reference:
htslib's tbx_index_build / tabix reads, src/typer/vcf.cpp:1308-1321, src/graph/constructor.cpp:163-176).  htslib is not in the
tree, so the files are held to the tabix and CSI specifications by a reader written here from those specifications: every
region query through the index (bins -> chunks -> records) returns exactly the records a scan of the text finds."""
import gzip
import os
import struct

import numpy as np
import pytest

import bam_writer as bw
from graphtyper_amd import lib as gtx

HEADER = "##fileformat=VCFv4.2\n##contig=<ID=chrA>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"


def make_vcf(rng, n=3000, contigs=(("chrA", 3_000_000), ("chrB", 200_000), ("chr10", 90_000_000))):
    lines, recs = [], []  # recs: (contig, beg, end) 0-based half-open
    for name, length in contigs:
        pos = np.sort(rng.integers(1, length, n if name != "chrB" else n // 10))
        for p in pos:
            p = int(p)
            k = rng.random()
            if k < 0.8:
                ref, alt, info = "ACGT"[int(rng.integers(0, 4))], "T", "."
                end = p - 1 + 1
            elif k < 0.93:
                ref = "A" * int(rng.integers(2, 60))
                alt, info, end = "A", "AC=1", p - 1 + len(ref)
            else:
                span = int(rng.integers(100, 200_000))
                ref, alt = "N", "<DEL>"
                info = ("END=%d;SVTYPE=DEL" if rng.random() < 0.5 else "SVTYPE=DEL;END=%d") % (p + span)
                end = p + span
            lines.append("%s\t%d\tid%d\t%s\t%s\t.\t.\t%s\n" % (name, p, len(lines), ref, alt, info))  # (the ID makes every line unique)
            recs.append((name, p - 1, end))
    return HEADER + "".join(lines), lines, recs


def parse_index(path, csi):
    raw = gzip.decompress(open(path, "rb").read())
    at = 0

    def get(fmt):
        nonlocal at
        v = struct.unpack_from("<" + fmt, raw, at)
        at += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]
    magic = raw[:4]
    at = 4
    if csi:
        assert magic == b"CSI\1"
        min_shift, depth, l_aux = get("iii")
        fmt, col_seq, col_beg, col_end, meta, skip, l_nm = get("7i")
        assert l_aux == 28 + l_nm
    else:
        assert magic == b"TBI\1"
        min_shift, depth = 14, 5
        n_ref = get("i")
        fmt, col_seq, col_beg, col_end, meta, skip, l_nm = get("7i")
    assert (fmt, col_seq, col_beg, col_end, meta, skip) == (2, 1, 2, 0, ord("#"), 0)
    names = raw[at:at + l_nm].split(b"\0")[:-1]
    at += l_nm
    if csi:
        n_ref = get("i")
    assert n_ref == len(names)
    refs = []
    for _ in range(n_ref):
        bins, n_bin = {}, get("i")
        for _ in range(n_bin):
            b = get("I")
            lo = get("Q") if csi else 0
            chunks = [get("QQ") for _ in range(get("i"))]
            bins[b] = (lo, chunks)
        linear = [get("Q") for _ in range(get("i"))] if not csi else None
        refs.append((bins, linear))
    assert get("Q") == 0 and at == len(raw)
    return dict(names=[n.decode() for n in names], refs=refs, min_shift=min_shift, depth=depth)


def reg2bins(beg, end, min_shift, depth):
    """the bins that may hold records overlapping [beg, end) (CSI specification)"""
    out, end = [], end - 1
    s, t = min_shift + depth * 3, 0
    for l in range(depth + 1):
        b, e = t + (beg >> s), t + (end >> s)
        out += range(b, e + 1)
        s -= 3
        t += 1 << (l * 3)
    return out


def records_between(data, member_starts, v0, v1):
    """the text between two virtual offsets of the uncompressed stream `data` (member_starts: file offset -> data offset)"""
    def pos(v):
        return member_starts[v >> 16] + (v & 0xFFFF)
    return data[pos(v0):pos(v1)]


def write_bgzf(path, text, rng):
    """BGZF members of varying sizes; returns file offset of a member -> offset of its data in the text"""
    data = text.encode()
    out, starts, at = bytearray(), {}, 0
    while at < len(data):
        n = int(rng.integers(200, 65000))
        starts[len(out)] = at
        out += bw.bgzf(data[at:at + n], block=65536)[:-28]
        at += n
    starts[len(out)] = len(data)
    out += bw.bgzf(b"")
    open(path, "wb").write(bytes(out))
    return starts, data


@pytest.mark.parametrize("csi", [False, True])
def test_region_queries_through_the_index(tmp_path, csi):
    rng = np.random.default_rng(5 + csi)
    text, lines, recs = make_vcf(rng)
    path = str(tmp_path / "v.vcf.gz")
    starts, data = write_bgzf(path, text, rng)
    gtx.tabix_build(path, 14 if csi else 0)
    idx = parse_index(path + (".csi" if csi else ".tbi"), csi)
    assert idx["names"] == ["chrA", "chrB", "chr10"]
    meta_bin = ((1 << (3 * (idx["depth"] + 1))) - 1) // 7 + 1
    for tid, name in enumerate(idx["names"]):
        bins, linear = idx["refs"][tid]
        mine = [r for r in recs if r[0] == name]
        assert bins[meta_bin][1][1][0] == len(mine)  # the pseudo-bin counts the contig's records
        for _ in range(60):
            b = int(rng.integers(0, max(r[2] for r in mine)))
            e = b + int(rng.choice([1, 100, 20_000, 2_000_000]))
            want = [k for k, r in enumerate(recs) if r[0] == name and r[1] < e and r[2] > b]
            # the specification's query: candidate bins -> chunks (not below the linear index' offset) -> the records inside
            min_off = 0
            if linear is not None and (b >> 14) < len(linear):
                min_off = linear[b >> 14]
            got = set()
            for bn in reg2bins(b, e, idx["min_shift"], idx["depth"]):
                if bn not in bins or bn == meta_bin:
                    continue
                for c0, c1 in bins[bn][1]:
                    if c1 <= min_off:
                        continue
                    for ln in records_between(data, starts, c0, c1).decode().splitlines(True):
                        k = lines.index(ln)
                        if recs[k][0] == name and recs[k][1] < e and recs[k][2] > b:
                            got.add(k)
            assert sorted(got) == want
            # and the library's own look-up: no overlapping record lies in front of the offset it gives
            v = gtx.tabix_start(path, name, b, e)
            if want:
                assert v is not None
                first = data.index(lines[want[0]].encode())
                assert starts[v >> 16] + (v & 0xFFFF) <= first
                at = starts[v >> 16] + (v & 0xFFFF)
                assert at == 0 or data[at - 1:at] == b"\n"  # a line start
    assert gtx.tabix_start(path, "chrZ", 0, 1000) is None


def test_graph_from_an_indexed_vcf(tmp_path):
    """gtx_graph_from_files through the index == through a scan of the plain file, and what lies in front of the region is not
    even read: a BGZF member there is damaged after the index was made -- a scan ends at it, the indexed read never sees it"""
    rng = np.random.default_rng(11)
    ref = "".join(rng.choice(list("ACGT"), 400_000))
    fa = str(tmp_path / "r.fa")
    open(fa, "w").write(">chr1\n" + "\n".join(ref[i:i + 60] for i in range(0, len(ref), 60)) + "\n")
    lines = []
    for p in sorted(set(int(x) for x in rng.integers(20_000, 380_000, 3000))):
        alt = "ACGT"[("ACGT".index(ref[p - 1]) + 1) % 4]
        lines.append("chr1\t%d\t.\t%s\t%s\t.\t.\t.\n" % (p, ref[p - 1], alt))
    plain = str(tmp_path / "clean.vcf")
    open(plain, "w").write(HEADER + "".join(lines))
    vz = str(tmp_path / "v.vcf.gz")
    data = (HEADER + "".join(lines)).encode()
    out, at = bytearray(), 0
    member_at = []
    while at < len(data):  # small members: several lie in front of the region
        member_at.append(len(out))
        out += bw.bgzf(data[at:at + 3000], block=65536)[:-28]
        at += 3000
    out += bw.bgzf(b"")
    open(vz, "wb").write(bytes(out))
    region = "chr1:150001-250000"
    want = gtx.graph_from_files(fa, plain, region)
    assert len(want[0]["var_order"]) > 300
    for min_shift, ext in ((0, ".tbi"), (14, ".csi")):
        open(vz, "wb").write(bytes(out))
        gtx.tabix_build(vz, min_shift)
        got = gtx.graph_from_files(fa, vz, region)
        assert got[1] == want[1] and all(np.array_equal(got[0][k], want[0][k]) for k in want[0])
        # a region the index knows nothing of: the reference-only graph
        assert len(gtx.graph_from_files(fa, vz, "chr1:1-15000")[0]["var_order"]) == 0
        # damage the second member (records around position 25 000)
        broken = bytearray(out)
        for k in range(member_at[1] + 30, member_at[1] + 60):
            broken[k] ^= 0x5A
        open(vz, "wb").write(bytes(broken))
        got = gtx.graph_from_files(fa, vz, region)
        assert all(np.array_equal(got[0][k], want[0][k]) for k in want[0])
        os.remove(vz + ext)
        scanned = gtx.graph_from_files(fa, vz, region)  # without the index the scan ends at the damage: no record of the region is reached
        assert len(scanned[0]["var_order"]) == 0


def test_files_that_cannot_be_indexed(tmp_path):
    rng = np.random.default_rng(3)
    text, lines, _ = make_vcf(rng, 200)
    gz = str(tmp_path / "plain.vcf.gz")
    open(gz, "wb").write(gzip.compress(text.encode()))
    with pytest.raises(gtx.GtxError):
        gtx.tabix_build(gz)  # gzip, not BGZF
    un = str(tmp_path / "unsorted.vcf.gz")
    write_bgzf(un, HEADER + lines[5] + lines[2], rng)
    with pytest.raises(gtx.GtxError):
        gtx.tabix_build(un)
    sp = str(tmp_path / "split.vcf.gz")
    write_bgzf(sp, HEADER + lines[0] + "chrB\t5\t.\tA\tC\t.\t.\t.\n" + lines[1], rng)
    with pytest.raises(gtx.GtxError):
        gtx.tabix_build(sp)  # a contig in two blocks
    with pytest.raises(gtx.GtxError):
        gtx.tabix_build(str(tmp_path / "missing.vcf.gz"))
    with pytest.raises(gtx.GtxError):
        gtx.tabix_start(gz, "chrA", 0, 10)  # no index beside it
