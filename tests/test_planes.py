"""Reads as bit planes (include/gtx.h): the host repack (gtx_pack_planes, gtx_stream_set_planes) against a numpy restatement
of the layout; on the GPU the device repack against the host's and gtx_align_batch_planes against gtx_align_batch."""
import ctypes as C

import numpy as np
import pytest

import harness
import scenarios
from graphtyper_amd import lib as gtx


@pytest.fixture(scope="module", autouse=True)
def _built():
    gtx.build()


@pytest.mark.parametrize("length,stride,plane_stride", [(150, 80, 80), (150, 75, 80), (151, 76, 80), (250, 128, 128), (63, 32, 32), (100, 50, 64),
                                                        (150, 80, 96)])
def test_host_pack_is_the_plane_layout(length, stride, plane_stride):
    rng = np.random.default_rng(length + stride)
    codes = rng.integers(0, 16, size=(257, length)).astype(np.uint8)
    seq = gtx.pack_nibbles(codes, stride=stride)
    got = gtx.pack_planes(seq, plane_stride)
    want = gtx.planes_reference(codes, plane_stride)
    assert got.shape == want.shape and np.array_equal(got, want)


def test_stream_push_writes_plane_rows():
    ref, recs, codes, rec = scenarios.paired_case("snp100", n_ref=20000, n_pairs=300, region_begin=5000)
    params = gtx.Params(75, 0, 0, 0, 0, 3840, 0, 0)
    a = gtx.Stream(params, 1)
    b = gtx.Stream(params, 1)
    b.set_planes(80)
    seq = gtx.pack_nibbles(codes)
    a_seq, a_meta, a_items = a.push(rec, seq)
    b_seq, b_meta, b_items = b.push(rec, seq)
    assert np.array_equal(a_meta, b_meta) and np.array_equal(a_items, b_items) and len(a_seq) > 100
    assert np.array_equal(b_seq, gtx.pack_planes(a_seq, 80))
    with pytest.raises(gtx.GtxError):
        b.set_planes(72)


@pytest.mark.gpu
def test_device_repack_and_plane_entry_point():
    import torch
    ref, recs, codes, pos = scenarios.synthetic_case("snp100", n_ref=60000, n_reads=20000, region_begin=1000000, err=0.01, n_rate=0.003)
    rng = np.random.default_rng(4)
    codes = codes.copy()
    amb = rng.random(codes.shape) < 0.003  # a few IUPAC sets besides N
    codes[amb] = rng.integers(0, 16, size=int(amb.sum())).astype(np.uint8)
    b = harness.GpuBackend(gtx.graph_from_records(ref, recs, region_begin=1000000))
    L = gtx.lib()
    seq = gtx.pack_nibbles(codes)
    meta = harness.read_meta(np.full(len(codes), 150), pos=pos)
    n = len(codes)
    want = b.align(seq, meta).copy()
    done_nibbles = b.hinted_done()
    d_seq, d_meta = b._dev(seq), b._dev(meta)
    d_planes = torch.zeros(n * 80, dtype=torch.uint8, device="cuda:0")
    gtx.check(L.gtx_reads_to_planes(b.ctx.h, d_seq.data_ptr(), 80, n, d_planes.data_ptr(), 80, None))
    torch.cuda.synchronize()
    host = gtx.pack_planes(seq, 80)
    assert np.array_equal(d_planes.cpu().numpy().reshape(n, 80), host)
    # an odd pitch of the nibble rows (byte-wise path of the repack kernel)
    seq75 = np.ascontiguousarray(seq[:, :75])
    d75 = b._dev(seq75)
    d_planes2 = torch.zeros(n * 80, dtype=torch.uint8, device="cuda:0")
    gtx.check(L.gtx_reads_to_planes(b.ctx.h, d75.data_ptr(), 75, n, d_planes2.data_ptr(), 80, None))
    torch.cuda.synchronize()
    assert np.array_equal(d_planes2.cpu().numpy().reshape(n, 80), gtx.pack_planes(seq75, 80))
    # the plane entry point: same records, same side array, same share of the position-hinted pass
    d_rec = torch.zeros(n * 2 * harness.REC_WORDS, dtype=torch.int32, device="cuda:0")
    d_flags = torch.full((2 * n,), 0x55, dtype=torch.uint8, device="cuda:0")
    gtx.check(L.gtx_align_batch_planes(b.ctx.h, d_planes.data_ptr(), 80, d_meta.data_ptr(), n, d_rec.data_ptr(), harness.REC_WORDS,
                                       d_flags.data_ptr(), None))
    torch.cuda.synchronize()
    got = d_rec.cpu().numpy().view(np.uint32)
    assert np.array_equal(got, want)
    assert b.hinted_done() == done_nibbles and done_nibbles > n // 2
    assert np.array_equal(d_flags.cpu().numpy(), (got.reshape(2 * n, -1)[:, 1] >> 31).astype(np.uint8))
    # rows of 96 bytes (six groups): the hinted pass stages them row by row
    host96 = gtx.pack_planes(seq, 96)
    d96 = b._dev(host96)
    d_rec.zero_()
    gtx.check(L.gtx_align_batch_planes(b.ctx.h, d96.data_ptr(), 96, d_meta.data_ptr(), n, d_rec.data_ptr(), harness.REC_WORDS, None, None))
    torch.cuda.synchronize()
    assert np.array_equal(d_rec.cpu().numpy().view(np.uint32), want)
    with pytest.raises(gtx.GtxError):
        gtx.check(L.gtx_align_batch_planes(b.ctx.h, d96.data_ptr(), 90, d_meta.data_ptr(), n, d_rec.data_ptr(), harness.REC_WORDS, None, None))


def test_item_words_host():
    """gtx_item_words: the read's index for an item of one forward-only read, GTX_ITEM_WORD_FULL for everything else"""
    items = np.zeros(6, gtx.SCORE_ITEM)
    items["first"]["align_index"] = [5, 6, 7, 8, 9, 0xFFFFFFFF]
    items["second"]["align_index"] = [gtx.INVALID_ID, 11, gtx.INVALID_ID, gtx.INVALID_ID, gtx.INVALID_ID, gtx.INVALID_ID]
    items["first"]["flag"] = [gtx.FLAG_FORWARD_ONLY, gtx.FLAG_FORWARD_ONLY, 0, gtx.FLAG_FORWARD_ONLY | 16, gtx.FLAG_FORWARD_ONLY, gtx.FLAG_FORWARD_ONLY]
    items["kind"] = [0, 0, 0, 0, 1, 0]
    assert gtx.item_words(items).tolist() == [5, 0xFFFFFFFF, 0xFFFFFFFF, 8, 0xFFFFFFFF, 0xFFFFFFFF]
    assert len(gtx.item_words(items[:0])) == 0
