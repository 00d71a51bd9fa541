"""BAM ingest: gtx_reads_* (graphtyper_amd/csrc/gtx_bam.cpp) against BAM files written by
tests/bam_writer.py -- the record stream must come out in the reference's order (records of one position by length and
packed bases inside a file, k-way merge over the files) with the reference's per-record facts (sample / read-group index
from @RG and the RG tag, get_score_diff with its parsing quirks, cigar ends, packed bases verbatim), a region must return
the records that overlap it, and the stream must drive the whole path: BAM -> gtx_stream -> align -> score -> calls == oracle."""
import numpy as np
import pytest

import bam_writer as bw
import harness
import scenarios
from graphtyper_amd import lib as gtx
from graphtyper_amd import synth
from oracle_lib import Oracle

REFS = [("chrA", 5000), ("chrB", 9000)]


def _random_files(tmp_path, seed, n_files=3):
    rng = np.random.default_rng(seed)
    files, paths, headers = [], [], []
    for f in range(n_files):
        rgs = [("rg%d_%d" % (f, k), "samp%d" % (f if k < 2 else 10 + f)) for k in range(int(rng.integers(1, 4)))] if f else []
        header = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in REFS) + \
            "".join("@RG\tID:%s\tPL:x\tSM:%s\tLB:l\n" % rg for rg in rgs)
        recs = []
        for tid in (0, 1):
            pos = np.sort(rng.integers(0, 600, size=int(rng.integers(30, 60))))  # many ties
            for p in pos:
                L = int(rng.choice([100, 150, 150, 150, 151]))
                codes = rng.choice([1, 2, 4, 8], size=L).astype(np.uint8)
                if rng.random() < 0.3 and recs and recs[-1]["tid"] == tid:
                    codes = recs[-1]["codes"].copy()  # duplicates / same start
                    L = len(codes)
                clip = int(rng.integers(0, 12)) if rng.random() < 0.3 else 0
                cigar = ([("S", clip)] if clip else []) + [("M", L - clip - 5), ("D", 3), ("M", 5)]
                aux = []
                if rng.random() < 0.2:
                    aux.append(("XB", "B", ("s", [1, 2, 3])))  # the score parser stops at an array: AS behind it is not seen
                aux.append(("NM", "C", int(rng.integers(0, 5))))
                if rng.random() < 0.9:
                    aux.append(("AS", str(rng.choice(["c", "C", "s", "S", "i", "I"])), int(rng.integers(0, 120))))
                if rng.random() < 0.7:
                    aux.append(("XS", "C", int(rng.integers(0, 120))))
                aux.append(("MD", "Z", "100A49"))
                rg = None
                if rgs:
                    rg = int(rng.integers(0, len(rgs)))
                    aux.append(("RG", "Z", rgs[rg][0]))
                recs.append(dict(tid=tid, pos=int(p), codes=codes, flag=int(rng.choice([0, 16, 99, 147, 1024])), mapq=int(rng.integers(0, 61)),
                                 cigar=cigar, mtid=tid, mpos=int(p) + 200, tlen=350, aux=aux, rg=rg, name="r%d_%d_%d" % (f, tid, len(recs))))
        path = str(tmp_path / ("s%d.x.bam" % f))
        bw.write_bam(path, REFS, header, [bw.record(r["name"], r["flag"], r["tid"], r["pos"], r["mapq"], r["cigar"], r["mtid"], r["mpos"], r["tlen"],
                                                  r["codes"], r["aux"]) for r in recs])
        files.append(recs)
        paths.append(path)
        headers.append(rgs)
    return files, paths, headers


def _expected(files, headers, keep=lambda r: True):
    """[(record dict, sample index, read-group index)] in the reference's order"""
    samples, rg_off, out_meta = [], 0, []
    for f, rgs in enumerate(headers):
        names = []
        for _, sm in rgs:
            if sm not in names:
                names.append(sm)
        if not names:
            names = ["s%d" % f]  # from the file name up to its first '.'
        out_meta.append((len(samples), rg_off, [names.index(sm) for _, sm in rgs]))
        samples += names
        rg_off += max(1, len(rgs))
    filtered = [[r for r in recs if keep(r)] for recs in files]
    out = []
    for f, i in bw.merged_order(filtered):
        r = filtered[f][i]
        s_off, r_off, rg2s = out_meta[f]
        several = len(headers[f]) > 1
        out.append((r, s_off + (rg2s[r["rg"]] if several else 0), r_off + (r["rg"] if several else 0)))
    return out, samples, rg_off


def _check(reads, want):
    got = []
    while True:
        recs, seq = reads.next(37, seq_stride=80)
        if len(recs) == 0:
            break
        got += list(zip(recs, seq))
    assert len(got) == len(want)
    names = {}
    for (g, gs), (r, sample, rg) in zip(got, want):
        L = len(r["codes"])
        assert (int(g["tid"]), int(g["pos"]), int(g["l_qseq"])) == (r["tid"], r["pos"], L)
        assert bytes(gs[:(L + 1) // 2]) == bw.pack_seq(r["codes"]) and not gs[(L + 1) // 2:].any()
        assert (int(g["flag"]), int(g["mapq"]), int(g["mtid"]), int(g["mpos"]), int(g["isize"])) == (r["flag"], r["mapq"], r["mtid"], r["mpos"], r["tlen"])
        assert int(g["score_diff"]) == bw.score_diff(r["aux"]), r["aux"]
        assert (int(g["sample"]), int(g["rg"])) == (sample, rg)
        enc = lambda c: (c[1] << 4) | "MIDNSHP=X".index(c[0])
        assert (int(g["n_cigar"]), int(g["cigar_front"]), int(g["cigar_back"])) == (len(r["cigar"]), enc(r["cigar"][0]), enc(r["cigar"][-1]))
        assert names.setdefault(int(g["name_id"]), r["name"]) == r["name"]
    assert len(names) == len(set(r["name"] for r, _, _ in want))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_record_stream_of_several_bam_files(tmp_path, seed):
    files, paths, headers = _random_files(tmp_path, seed)
    want, samples, n_rg = _expected(files, headers)
    reads = gtx.Reads(paths)
    assert reads.samples == samples and reads.n_read_groups == n_rg
    _check(reads, want)
    assert any(bw.score_diff(r["aux"]) == 0 and any(a[0] == "AS" for a in r["aux"]) and any(a[1] == "B" for a in r["aux"]) for r, _, _ in want)


def test_region_through_the_bai_index(tmp_path):
    """with a .bai beside the file the scan starts where the index says: a member of garbage between the header and the
    records is never read (without the index the same file fails), the records are the overlapping ones, and a region the
    index knows nothing about is empty"""
    rng = np.random.default_rng(5)
    refs = [("chrA", 400000), ("chrB", 900000)]
    recs = []
    for tid in (0, 1):
        for p in np.sort(rng.integers(0, refs[tid][1] - 400, size=4000)):
            L = 150
            codes = rng.choice([1, 2, 4, 8], size=L).astype(np.uint8)
            recs.append(dict(tid=tid, pos=int(p), codes=codes, flag=0, mapq=60, cigar=[("M", 100), ("D", int(rng.integers(1, 200))), ("M", 50)], mtid=-1, mpos=-1,
                             tlen=0, aux=[("AS", "C", 100)], rg=None, name="r%d" % len(recs)))
    header = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    span = lambda r: sum(n for op, n in r["cigar"] if op in "MDN=X")
    path = str(tmp_path / "big.bam")
    blobs = [bw.record(r["name"], r["flag"], r["tid"], r["pos"], r["mapq"], r["cigar"], r["mtid"], r["mpos"], r["tlen"], r["codes"], r["aux"]) for r in recs]
    bw.write_bam(path, refs, header, blobs, index=[(r["tid"], r["pos"], r["pos"] + span(r)) for r in recs], poison=True)
    for region, tid, lo, hi in (("chrB:500001-520000", 1, 500000, 520000), ("chrA:1-3000", 0, 0, 3000), ("chrB:880000-900000", 1, 879999, 900000)):
        want, _, _ = _expected([recs], [[]], keep=lambda r: r["tid"] == tid and r["pos"] < hi and r["pos"] + span(r) > lo)
        assert len(want) > 20
        _check(gtx.Reads([path], region=region), want)
    import os
    os.rename(path + ".bai", path + ".hidden")
    with pytest.raises(gtx.GtxError):  # no index: the scan from the head runs into the garbage
        gtx.Reads([path], region="chrB:500001-520000")
    # an index without any bin for the region: nothing to read (and nothing is scanned)
    bw.write_bam(path, refs, header, blobs[:4000], index=[(r["tid"], r["pos"], r["pos"] + span(r)) for r in recs[:4000]], poison=True)
    recs_b, _ = gtx.Reads([path], region="chrB:1000-2000").next(10)
    assert len(recs_b) == 0


@pytest.mark.parametrize("geometry", [(14, 5), (12, 6), (16, 3)])
def test_region_through_a_csi_index(tmp_path, geometry):
    """the same through a .csi (htslib's index with a free bin geometry, BGZF-compressed): the scan starts where the index says
    -- the member of garbage behind the header is never read -- and yields the overlapping records"""
    rng = np.random.default_rng(6)
    refs = [("chrA", 400000), ("chrB", 900000)]
    recs = []
    for tid in (0, 1):
        for p in np.sort(rng.integers(0, refs[tid][1] - 400, size=1500)):
            codes = rng.choice([1, 2, 4, 8], size=150).astype(np.uint8)
            recs.append(dict(tid=tid, pos=int(p), codes=codes, flag=0, mapq=60, cigar=[("M", 100), ("D", int(rng.integers(1, 200))), ("M", 50)], mtid=-1, mpos=-1,
                             tlen=0, aux=[("AS", "C", 100)], rg=None, name="r%d" % len(recs)))
    header = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    span = lambda r: sum(n for op, n in r["cigar"] if op in "MDN=X")
    path = str(tmp_path / "csi.bam")
    blobs = [bw.record(r["name"], r["flag"], r["tid"], r["pos"], r["mapq"], r["cigar"], r["mtid"], r["mpos"], r["tlen"], r["codes"], r["aux"]) for r in recs]
    bw.write_bam(path, refs, header, blobs, index=[(r["tid"], r["pos"], r["pos"] + span(r)) for r in recs], poison=True, csi=geometry)
    for region, tid, lo, hi in (("chrB:500001-520000", 1, 500000, 520000), ("chrA:1-9000", 0, 0, 9000), ("chrB:860000-900000", 1, 859999, 900000)):
        want, _, _ = _expected([recs], [[]], keep=lambda r: r["tid"] == tid and r["pos"] < hi and r["pos"] + span(r) > lo)
        assert len(want) > 5
        _check(gtx.Reads([path], region=region), want)
    import os
    os.rename(path + ".csi", path + ".hidden")
    with pytest.raises(gtx.GtxError):  # no index: the scan from the head runs into the garbage
        gtx.Reads([path], region="chrB:500001-520000")


def test_region_returns_the_overlapping_records(tmp_path):
    files, paths, headers = _random_files(tmp_path, 7)
    lo, hi = 150, 300  # 1-based inclusive region chrB:150-300 = 0-based [149, 300)

    def overlaps(r):
        span = sum(n for op, n in r["cigar"] if op in "MDN=X")
        return r["tid"] == 1 and r["pos"] < hi and r["pos"] + span > lo - 1
    want, _, _ = _expected(files, headers, keep=overlaps)
    assert 10 < len(want) < sum(len(f) for f in files)
    _check(gtx.Reads(paths, region="chrB:%d-%d" % (lo, hi)), want)
    with pytest.raises(gtx.GtxError):
        gtx.Reads(paths, region="chrZ:1-10")
    with pytest.raises(gtx.GtxError):
        gtx.Reads([str(tmp_path / "missing.bam")])
    (tmp_path / "not.bam").write_bytes(bw.bgzf(b"CRAM....not a bam"))
    with pytest.raises(gtx.GtxError) as e:
        gtx.Reads([str(tmp_path / "not.bam")])
    assert e.value.status == 4


def bam_to_calls_case(Backend, tmp_path):
    """two samples in two BAM files (paired reads of a SNP graph) -> gtx_reads -> gtx_stream -> align -> score == oracle"""
    rb = 310000
    ref, recs, codes, rec = scenarios.paired_case("snp100", n_ref=12000, n_pairs=600, region_begin=rb, n_samples=2)
    refs = [("chr7", rb + 20000)]
    per_file = {0: [], 1: []}
    for i in range(len(rec)):
        r = rec[i]
        flag = int(r["flag"])
        aux = [("AS", "C", 140), ("XS", "C", 140 - int(r["score_diff"]))] if int(r["score_diff"]) else [("AS", "C", 100), ("XS", "C", 100)]
        per_file[int(r["sample"])].append(bw.record("q%d" % int(r["name_id"]), flag, int(r["tid"]), int(r["pos"]), int(r["mapq"]), [("M", codes.shape[1])],
                                                    int(r["mtid"]), 0, int(r["isize"]), codes[i], aux))
    paths = []
    for s in (0, 1):
        paths.append(str(tmp_path / ("SAMP%d.bam" % s)))
        bw.write_bam(paths[-1], refs, "@HD\tVN:1.6\n@SQ\tSN:chr7\tLN:%d\n@RG\tID:a%d\tSM:person%d\n" % (refs[0][1], s, s), per_file[s])
    reads = gtx.Reads(paths, region="chr7")
    assert reads.samples == ["person0", "person1"] and reads.n_read_groups == 2
    o = Oracle(ref, recs, region_begin=rb)
    b = Backend(gtx.graph_from_records(ref, recs, region_begin=rb))
    st = gtx.Stream(b.ctx.params, reads.n_read_groups)
    og = o.genotyper(2, reads.n_read_groups)
    a_seq, a_meta, items = [], [], []
    n_total = 0
    while True:
        srec, sseq = reads.next(500, seq_stride=80)
        if len(srec) == 0:
            break
        n_total += len(srec)
        # the oracle gets the same stream (bases decoded from the packed rows)
        L = int(srec["l_qseq"][0])
        unpacked = np.stack([(sseq[:, :(L + 1) // 2] >> 4), (sseq[:, :(L + 1) // 2] & 15)], axis=2).reshape(len(srec), -1)[:, :L]
        og.push(list(unpacked), flags=srec["flag"], tid=srec["tid"], mtid=srec["mtid"], pos=srec["pos"], isize=srec["isize"], mapq=srec["mapq"],
                score_diff=srec["score_diff"], name=srec["name_id"], sample=srec["sample"], rg=srec["rg"])
        p = st.push(srec, sseq)
        a_seq.append(p[0]); a_meta.append(p[1]); items.append(p[2])
    assert n_total == len(rec)
    a_seq, a_meta, items = np.concatenate(a_seq), np.concatenate(a_meta), np.concatenate(items)
    assert st.counts() == og.counts()
    assert (a_meta["pos"] >= 0).all()  # the position hint travels from the BAM record
    records = b.align(a_seq, a_meta)
    assert b.hinted_done() > len(a_meta) // 2
    acc = b.score(items, records, 2)
    got, want = harness.canonical_scores(b.ctx, acc), og.scores()
    assert len(got) == len(want) and np.array_equal(got, want)
    phred, calls = b.calls(acc, 2)
    assert np.array_equal(harness.canonical_calls(b.ctx, phred, calls, 2), og.calls())
    text = b.ctx.vcf_records("chr7", reads.samples, acc.gt_cov, acc.stat_u64, acc.stat_u32, phred, calls)
    assert text == og.vcf_records("chr7", reads.samples) and text.count(b"\n") == b.ctx.n_hap + 1
    return text


def test_bam_files_to_vcf_text(tmp_path):
    text = bam_to_calls_case(harness.EmuBackend, tmp_path)
    assert b"person0\tperson1" in text.split(b"\n")[0] and b"\t0/1:" in text


def test_damaged_bam_files_are_refused_not_crashed_on(tmp_path):
    """a few seeds of tests/fuzz_bam.py: damaged records, headers, BGZF members and truncated files end in GTX_ERR_* (or in
    the records that are still readable), never in a crash of the calling process"""
    import fuzz_bam
    assert fuzz_bam.run(0, 24, tmp=str(tmp_path)) == 0


@pytest.mark.parametrize("team", ["0", "3"])
def test_readers_on_several_host_threads_share_the_inflate_team(tmp_path, monkeypatch, team):
    """the BGZF members of all open readers are inflated by one team of worker threads (or by the readers themselves,
    GTX_BGZF_THREADS=0): four readers on four host threads, each over several files, must each see the reference's order"""
    import threading
    monkeypatch.setenv("GTX_BGZF_THREADS", team)
    files, paths, headers = _random_files(tmp_path, 11)
    want, samples, n_rg = _expected(files, headers)
    failures = []

    def one():
        try:
            for _ in range(3):
                reads = gtx.Reads(paths)
                _check(reads, want)
                reads.close()
        except Exception as e:  # (an assertion in a thread would otherwise be lost)
            failures.append(repr(e))
    team_of_hosts = [threading.Thread(target=one) for _ in range(4)]
    for t in team_of_hosts:
        t.start()
    for t in team_of_hosts:
        t.join()
    assert not failures, failures[0]


def test_array_field_with_a_hostile_count_ends_the_aux_walk(tmp_path):
    """a 'B' aux field whose element count runs far past the record (count 0xFFFFFFF8, subtype c: a 32-bit cursor would
    wrap back onto the same field and never leave it) in a file with two read groups -- the only case in which the reader
    looks for the RG tag.  The record counts as one without a read group, which such a file does not allow (the reference
    exits there, hts_reader.cpp:354-387): the reader must come back with that error instead of hanging."""
    import os
    import struct
    import subprocess
    import sys
    refs = [("chrA", 5000)]
    header = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chrA\tLN:5000\n@RG\tID:a\tSM:s1\n@RG\tID:b\tSM:s2\n"
    codes = np.full(100, 1, np.uint8)
    good = bw.record("r0", 0, 0, 100, 60, [("M", 100)], -1, -1, 0, codes, [("RG", "Z", "b")])
    body = bw.record("r1", 0, 0, 200, 60, [("M", 100)], -1, -1, 0, codes, [])[4:]
    body += b"XBBc" + struct.pack("<I", 0xFFFFFFF8) + b"\x01\x02\x03"
    bad = struct.pack("<i", len(body)) + body
    path = str(tmp_path / "hostile.bam")
    bw.write_bam(path, refs, header, [good, bad])
    child = ("import sys; sys.path.insert(0, %r)\nfrom graphtyper_amd import lib as gtx\nr = gtx.Reads([%r])\nn = 0\n"
             "while True:\n    recs, seq = r.next(64)\n    if len(recs) == 0: break\n    n += len(recs)\nprint('records', n)\n") % (
                 os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path)
    out = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=60)
    assert "without RG tag" in out.stderr, (out.stdout, out.stderr[-300:])
