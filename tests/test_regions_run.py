"""gtx_regions_run -- the loop of genotype_regions (src/utilities/genotype.cpp:735-738) inside the library: every region's text is
the text the six calls give when made one after the other, whatever the number of threads per stage; a job that ran into a
capacity limit fails, the others are not touched."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from graphtyper_amd import lib as gtx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_structs_of_the_binding_are_the_header_s(tmp_path):
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "gtx.h"\nint main(void) { printf("%zu %zu %zu %zu %zu\\n", sizeof(gtx_region_job), '
                   'sizeof(gtx_regions_stats), offsetof(gtx_region_job, d_items), offsetof(gtx_region_job, text), offsetof(gtx_regions_stats, records_failed)); return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got == [C.sizeof(gtx.RegionJob), C.sizeof(gtx.RegionsStats), gtx.RegionJob.d_items.offset, gtx.RegionJob.text.offset,
                   gtx.RegionsStats.records_failed.offset]


def test_refuses_to_run_without_a_device():
    jobs = gtx.RegionJobs([dict(reference="ACGT" * 100, region_begin=1000, records=[(1100, "A", ["C"], "")], d_planes=0, plane_stride=80, d_meta=0, n_reads=0,
                                d_items=0, n_items=0)])
    with pytest.raises(gtx.GtxError) as e:
        jobs.run(["S"], device=-1)
    assert e.value.status == 2  # GTX_ERR_NO_DEVICE
    with pytest.raises(gtx.GtxError) as e:  # reads promised, no rows
        gtx.RegionJobs([dict(reference="ACGT" * 100, region_begin=1000, records=[], d_planes=0, plane_stride=80, d_meta=0, n_reads=5, d_items=0, n_items=0)]).run(["S"], device=0)
    assert e.value.status == 1  # GTX_ERR_ARG


def _regions(torch, device, n_regions, region_len, n_reads, n_samples, snp_every=200, seed=0):
    import bench
    from graphtyper_amd import synth
    ref = synth.make_reference(n_regions * region_len, seed=31 + seed)
    out = []
    for r in range(n_regions):
        rb = bench.REGION_BEGIN + r * region_len
        sub = np.ascontiguousarray(ref[r * region_len:(r + 1) * region_len])
        if isinstance(snp_every, list):
            recs, clusters = synth.make_snp_records(sub, snp_every[r], seed=100 + r, region_begin=rb), False
        else:
            clusters = False
            recs = synth.make_snp_records(sub, snp_every if r % 3 != 2 else 55, seed=100 + r, region_begin=rb, first=20 + 7 * r)
        d_seq, d_pos = bench.make_reads_on_device(torch, sub, recs, n_reads, seed=900 + r, device=device, REGION_LEN=region_len, region_begin=rb, err_rate=0.005)
        meta = np.zeros(n_reads, gtx.READ_META)
        meta["l_qseq"] = bench.READ_LEN
        meta["flag"] = gtx.FLAG_FORWARD_ONLY
        meta["pos"] = d_pos.cpu().numpy().astype(np.int32)
        items = np.zeros(n_reads, gtx.SCORE_ITEM)
        items["first"]["align_index"] = np.arange(n_reads, dtype=np.uint32)
        items["first"]["mapq"] = 60
        items["first"]["flag"] = gtx.FLAG_FORWARD_ONLY
        items["first"]["pos"] = meta["pos"]
        items["second"]["align_index"] = gtx.INVALID_ID
        items["sample"] = np.random.default_rng(r).integers(0, n_samples, size=n_reads).astype(np.uint32)
        out.append(dict(rb=rb, ref_str=synth.bases_to_str(sub), recs=recs, add_all=clusters, d_seq=d_seq,
                        d_planes=torch.empty((n_reads, 80), dtype=torch.uint8, device=device),
                        d_meta=torch.from_numpy(meta.view(np.uint8).reshape(n_reads, -1).copy()).to(device),
                        d_items=torch.from_numpy(items.view(np.uint8).reshape(n_reads, -1).copy()).to(device)))
    return out


def _one_by_one(torch, device, regions, names, n_reads):
    """the six calls per region from here, as bench.py's regions leg makes them"""
    import bench
    L = gtx.lib()
    texts = []
    for q in regions:
        c = gtx.Context(gtx.graph_from_records(q["ref_str"], q["recs"], region_begin=q["rb"], add_all_variants=q["add_all"]), device=device.index or 0)
        gtx.check(L.gtx_reads_to_planes(c.h, q["d_seq"].data_ptr(), 80, n_reads, q["d_planes"].data_ptr(), 80, None))
        d_rec = torch.zeros(n_reads * 2 * bench.REC_WORDS, dtype=torch.int32, device=device)
        d_fl = torch.zeros(n_reads * 2, dtype=torch.uint8, device=device)
        buf = gtx.ScoreBuffers()
        gtx.check(L.gtx_scores_alloc(c.h, len(names), 1 << 16, C.byref(buf), None))
        d_phred = torch.empty(max(len(names) * c.total_tri, 1), dtype=torch.uint8, device=device)
        d_calls = torch.empty(max(len(names) * c.n_hap, 1) * gtx.SAMPLE_CALL.itemsize, dtype=torch.uint8, device=device)
        gtx.check(L.gtx_align_batch_planes(c.h, q["d_planes"].data_ptr(), 80, q["d_meta"].data_ptr(), n_reads, d_rec.data_ptr(), bench.REC_WORDS, d_fl.data_ptr(), None))
        gtx.check(L.gtx_score_batch_flags(c.h, q["d_items"].data_ptr(), n_reads, d_rec.data_ptr(), bench.REC_WORDS, d_fl.data_ptr(), C.byref(buf), None))
        gtx.check(L.gtx_calls_batch(c.h, C.byref(buf), d_phred.data_ptr(), d_calls.data_ptr(), None))
        torch.cuda.synchronize()
        nh, ta, ns = c.n_hap, c.total_allele, len(names)
        texts.append(c.vcf_records("chr20", names, gtx.download(buf.d_gt_cov, np.uint32, ns * ta), gtx.download(buf.d_stat_u64, np.uint64, nh + 2 * ta),
                                   gtx.download(buf.d_stat_u32, np.uint32, nh + 6 * ta), d_phred.cpu().numpy()[:ns * c.total_tri],
                                   d_calls.cpu().numpy().view(gtx.SAMPLE_CALL)[:ns * nh]))
        L.gtx_scores_free(c.h, C.byref(buf))
        c.close()
    return texts


@pytest.mark.gpu
def test_every_region_s_text_is_the_one_the_six_calls_give():
    import torch
    import bench
    device = torch.device("cuda", 0)
    n_reads, names = 40000, ["SAMP%02d" % i for i in range(6)]
    regions = _regions(torch, device, 9, 20000, n_reads, len(names))
    want = _one_by_one(torch, device, regions, names, n_reads)
    assert all(t.count(b"\n") > 20 for t in want) and len(set(want)) == len(want)
    jobs = gtx.RegionJobs([dict(reference=q["ref_str"], region_begin=q["rb"], records=q["recs"], add_all_variants=q["add_all"], d_planes=q["d_planes"].data_ptr(),
                                plane_stride=80, d_meta=q["d_meta"].data_ptr(), n_reads=n_reads, d_items=q["d_items"].data_ptr(), n_items=n_reads) for q in regions])
    for shape in [(1, 1, 1), (4, 2, 3), (3, 3, 2), (9, 4, 4)]:
        for _ in range(3):
            got, st = jobs.run(names, contig="chr20", rec_words=bench.REC_WORDS, builders=shape[0], device_threads=shape[1], text_threads=shape[2])
            assert got == want, shape
            assert jobs.status == [0] * len(regions) and st["records_failed"] == 0 and st["device_s"] > 0 and st["vcf_text_s"] > 0
            assert (st["n_builders"], st["n_device_threads"], st["n_text_threads"]) == shape
    # the window of gtx_vcf_request: sites outside are not written
    j = jobs.jobs[0]
    lo = regions[0]["rb"] + 5000
    j.vcf_begin, j.vcf_end = lo, lo + 5000
    got, _ = jobs.run(names, contig="chr20", rec_words=bench.REC_WORDS)
    kept = [l for l in got[0].split(b"\n")[1:-1]]
    assert 0 < len(kept) < want[0].count(b"\n") - 1 and all(lo <= int(l.split(b"\t")[1]) <= lo + 5000 for l in kept)
    assert got[1:] == want[1:]


@pytest.mark.gpu
def test_a_job_beyond_a_capacity_limit_fails_and_the_others_do_not():
    """connections beyond conn_cap are connections the region's phase flags would lack: that job has no text.  Regions 0 and 2 have
    a SNP every 3 kb (no read over two sites: nothing to log), region 1 one every 120 bp (reads over two sites 120 bp apart: beyond
    the window of the dense counters, every such read a log entry)"""
    import torch
    import bench
    device = torch.device("cuda", 0)
    n_reads, names = 20000, ["A", "B"]
    regions = _regions(torch, device, 3, 20000, n_reads, len(names), snp_every=[3000, 120, 3000], seed=5)
    want = _one_by_one(torch, device, regions, names, n_reads)
    jobs = gtx.RegionJobs([dict(reference=r["ref_str"], region_begin=r["rb"], records=r["recs"], add_all_variants=r["add_all"], d_planes=r["d_planes"].data_ptr(),
                                plane_stride=80, d_meta=r["d_meta"].data_ptr(), n_reads=n_reads, d_items=r["d_items"].data_ptr(), n_items=n_reads) for r in regions])
    with pytest.raises(gtx.GtxError) as e:
        jobs.run(names, contig="chr20", rec_words=bench.REC_WORDS, conn_cap=16)
    assert e.value.status == 5 and "connections beyond the log" in str(e.value)  # GTX_ERR_CAPACITY
    assert jobs.status == [0, 5, 0]
    got, st = jobs.run(names, contig="chr20", rec_words=bench.REC_WORDS, conn_cap=1 << 20)  # with room for them: every job has its text
    assert got == want and st["connections_dropped"] == 0


@pytest.mark.gpu
def test_a_failed_record_of_one_region_is_not_counted_against_the_next():
    """A device thread recycles its record slots from region to region, and the reverse slot of a GTX_FLAG_FORWARD_ONLY read is
    never written (gtx.h, gtx_records_failed): region 0 has reads in both orientations that are longer than the passes take -- a
    table-overflow status in both of their slots, the job fails -- and region 1's forward-only reads at the same indices must not
    inherit the status of the reverse slots (advisor, round 5: every later region of the thread failed with GTX_ERR_CAPACITY)."""
    import torch
    import bench
    device = torch.device("cuda", 0)
    n_reads, names = 20000, ["A", "B"]
    regions = _regions(torch, device, 3, 20000, n_reads, len(names), snp_every=[500, 500, 500], seed=9)
    want = _one_by_one(torch, device, regions, names, n_reads)
    meta = regions[0]["d_meta"].cpu().numpy().view(gtx.READ_META).reshape(-1).copy()
    meta["flag"][[3, 700, 19999]] = 0
    meta["l_qseq"][[3, 700, 19999]] = 300
    regions[0]["d_meta"] = torch.from_numpy(meta.view(np.uint8).reshape(n_reads, -1).copy()).to(device)
    jobs = gtx.RegionJobs([dict(reference=r["ref_str"], region_begin=r["rb"], records=r["recs"], add_all_variants=r["add_all"], d_planes=r["d_planes"].data_ptr(),
                                plane_stride=80, d_meta=r["d_meta"].data_ptr(), n_reads=n_reads, d_items=r["d_items"].data_ptr(), n_items=n_reads) for r in regions])
    with pytest.raises(gtx.GtxError) as e:
        jobs.run(names, contig="chr20", rec_words=bench.REC_WORDS, builders=1, device_threads=1, text_threads=1)
    assert e.value.status == 5 and "6 records with a table-overflow status" in str(e.value)
    assert jobs.status == [5, 0, 0]
    assert jobs.texts[0] is None and jobs.texts[1:] == want[1:]
