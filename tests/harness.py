"""Shared test plumbing: run the alignment / scoring kernels either on the GPU through libgtx's C ABI (GpuBackend,
used by the `-m gpu` parity tests) or through the host emulation of the kernel sources (EmuBackend, tests/emu, a
debugging aid for containers without a GPU), and put results into the oracle's canonical forms for exact comparison."""
import ctypes as C
import os
import subprocess

import numpy as np

from graphtyper_amd import lib as gtx

HERE = os.path.dirname(os.path.abspath(__file__))
REC_WORDS = 64
FLAG_PAIRED, FLAG_REVERSED, FLAG_MATE_REVERSED, FLAG_FIRST, FLAG_SECOND = 1, 16, 32, 64, 128


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Accumulators:
    """host copy of the score accumulators (layout: include/gtx.h, gtx_score_buffers)"""

    def __init__(self, ctx, n_samples, conn_cap=1 << 20, near=True):
        self.n_samples = n_samples
        self.conn_cap = conn_cap
        self.log_score = np.zeros(n_samples * ctx.total_tri, np.uint32)
        self.gt_cov = np.zeros(n_samples * ctx.total_allele, np.uint32)
        self.hap_u32 = np.zeros(n_samples * ctx.n_hap * 4, np.uint32)
        self.stat_u64 = np.zeros(ctx.n_hap + 2 * ctx.total_allele, np.uint64)
        self.stat_u32 = np.zeros(ctx.n_hap + 6 * ctx.total_allele, np.uint32)
        self.conn_log = np.zeros(conn_cap * 6, np.uint32)
        self.conn_count = np.zeros(2, np.uint32)
        # dense counters of the connections between near haplotypes; without them every connection goes to the log
        self.conn_near = np.zeros(n_samples * ctx.total_near, np.uint32) if near else None
        # SV calling: the reference-depth track (difference array, one word more than positions per sample)
        self.ref_depth_len = ctx.ref_depth_len
        self.ref_depth = np.zeros(n_samples * (ctx.ref_depth_len + 1), np.uint32) if (near and ctx.params.is_sv_graph) else None

    def arrays(self):
        a = [self.log_score, self.gt_cov, self.hap_u32, self.stat_u64, self.stat_u32, self.conn_log, self.conn_count]
        return a + ([self.conn_near] if self.conn_near is not None else []) + ([self.ref_depth] if self.ref_depth is not None else [])

    def buffers(self, pointers):
        """gtx_score_buffers over `pointers` (one per array of arrays(), host or device)"""
        return gtx.ScoreBuffers(self.n_samples, *pointers[:7], self.conn_cap, pointers[7] if self.conn_near is not None else None,
                                pointers[8] if self.ref_depth is not None else None, self.ref_depth_len if self.ref_depth is not None else 0)

    def depths(self):
        """the finalised reference-depth track [n_samples, ref_depth_len] (gtx_ref_depth_finalize on a copy)"""
        d = self.ref_depth.copy()
        sat = C.c_uint64()
        gtx.check(gtx.lib().gtx_ref_depth_finalize(_p(d), self.n_samples, self.ref_depth_len, C.byref(sat)))
        return d.reshape(self.n_samples, self.ref_depth_len + 1)[:, :self.ref_depth_len]


class EmuBackend:
    name = "emu"

    def __init__(self, graph, **params):
        so = os.environ.get("GTX_EMU_LIB") or os.path.join(HERE, "emu", "libgtx_emu.so")
        subprocess.check_call(["make", "-C", os.path.join(HERE, "emu"), "-s"])
        self.L = C.CDLL(so)
        self.L.emu_new.restype = C.c_void_p
        self.ctx = gtx.Context(graph, device=-1, **params)  # host inspection context (layout tables, index dump)
        err = C.create_string_buffer(256)
        self.h = self.L.emu_new(C.byref(self.ctx.view), C.byref(self.ctx.params), err, 256)
        if not self.h:
            raise RuntimeError(err.value.decode())

    def __del__(self):
        if getattr(self, "h", None):
            self.L.emu_free(C.c_void_p(self.h))
            self.h = None

    def align(self, seq, meta, rec_words=REC_WORDS):
        seq = np.ascontiguousarray(seq, np.uint8)
        meta = np.ascontiguousarray(meta, gtx.READ_META)
        n = len(meta)
        rec = np.zeros(n * 2 * rec_words, np.uint32)
        self.L.emu_align(C.c_void_p(self.h), _p(seq), C.c_uint32(seq.shape[1]), _p(meta), C.c_uint32(n), _p(rec), C.c_uint32(rec_words))
        return rec

    def rewind_big_records(self):
        self.L.emu_big_records_rewind(C.c_void_p(self.h))

    def hinted_done(self):
        """forward tasks the position-hinted pass finished in the last align call"""
        self.L.emu_hinted_done.restype = C.c_uint64
        return int(self.L.emu_hinted_done(C.c_void_p(self.h)))

    def exact_pass_tasks(self):
        """tasks of the last align call that went through the exact pass (a small part of the slab, a large part, the whole slab, still refused)"""
        self.L.emu_exact_pass_tasks.restype = C.c_uint64
        return tuple(int(self.L.emu_exact_pass_tasks(C.c_void_p(self.h), k)) for k in range(4))

    def big_records(self):
        ptr, cap = C.POINTER(C.c_uint32)(), C.c_uint64()
        self.L.emu_big_records(C.c_void_p(self.h), C.byref(ptr), C.byref(cap))
        self.L.emu_second_pass_tasks.restype = C.c_uint64
        tasks = int(self.L.emu_second_pass_tasks(C.c_void_p(self.h)))
        return np.ctypeslib.as_array(ptr, shape=(int(cap.value),)).copy(), tasks

    def score(self, items, records, n_samples=1, rec_words=REC_WORDS, near=True):
        items = np.ascontiguousarray(items, gtx.SCORE_ITEM)
        acc = Accumulators(self.ctx, n_samples, near=near)
        buf = acc.buffers([_p(a) for a in acc.arrays()])
        errors = self.L.emu_score(C.c_void_p(self.h), _p(items), C.c_uint32(len(items)), _p(records), C.c_uint32(rec_words), C.byref(buf))
        assert errors == 0
        return acc

    def score_replay(self, items, records, acc, rec_words=REC_WORDS):
        """gtx_scores_replay on the host arrays of `acc` (in place); returns the number of cells replayed"""
        items = np.ascontiguousarray(items, gtx.SCORE_ITEM)
        buf = acc.buffers([_p(a) for a in acc.arrays()])
        self.L.emu_score_replay.restype = C.c_long
        n = self.L.emu_score_replay(C.c_void_p(self.h), _p(items), C.c_uint32(len(items)), _p(records), C.c_uint32(rec_words), C.byref(buf))
        assert n >= 0
        return int(n)

    def score_replay_log(self, items, records, acc, item_base=0, rec_words=REC_WORDS):
        """gtx_scores_replay_log on host arrays: this rank's entries (gtx.REPLAY_ENTRY) for the cells at the guard of `acc`"""
        items = np.ascontiguousarray(items, gtx.SCORE_ITEM)
        buf = acc.buffers([_p(a) for a in acc.arrays()])
        out = np.zeros(1 << 20, gtx.REPLAY_ENTRY)
        self.L.emu_score_replay_log.restype = C.c_long
        n = self.L.emu_score_replay_log(C.c_void_p(self.h), _p(items), C.c_uint32(len(items)), _p(records), C.c_uint32(rec_words), C.byref(buf),
                                        C.c_uint32(item_base), _p(out), C.c_long(len(out)))
        assert n >= 0
        return out[:n].copy()

    def score_replay_apply(self, acc, entries):
        """gtx_scores_replay_apply on the host arrays of `acc` (in place); returns the number of cells replayed"""
        entries = np.ascontiguousarray(entries, gtx.REPLAY_ENTRY)
        buf = acc.buffers([_p(a) for a in acc.arrays()])
        self.L.emu_score_replay_apply.restype = C.c_long
        return int(self.L.emu_score_replay_apply(C.c_void_p(self.h), C.byref(buf), _p(entries), C.c_long(len(entries))))

    def calls(self, acc, n_samples):
        """gtx_calls_batch contract on the host: (phred [n_samples * total_tri] u8, SAMPLE_CALL [n_samples * n_hap])"""
        buf = acc.buffers([_p(a) for a in acc.arrays()])
        phred = np.zeros(n_samples * self.ctx.total_tri, np.uint8)
        calls = np.zeros(n_samples * self.ctx.n_hap, gtx.SAMPLE_CALL)
        self.L.emu_calls(C.c_void_p(self.h), C.byref(buf), _p(phred), _p(calls))
        return phred, calls


class GpuBackend:
    """libgtx's C ABI on cuda:0; torch only owns the device buffers"""
    name = "gpu"

    def __init__(self, graph, **params):
        import torch
        self.torch = torch
        assert torch.cuda.is_available(), "GpuBackend needs a GPU"
        self.ctx = gtx.Context(graph, device=0, **params)
        self.ctx.pass_times()  # arms the per-launch timing (hinted_done reads the task counts that come with it)

    def _dev(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to("cuda:0")

    def align(self, seq, meta, rec_words=REC_WORDS):
        torch = self.torch
        seq = np.ascontiguousarray(seq, np.uint8)
        meta = np.ascontiguousarray(meta, gtx.READ_META)
        n = len(meta)
        d_seq, d_meta = self._dev(seq), self._dev(meta)
        d_rec = torch.zeros(max(n, 1) * 2 * rec_words, dtype=torch.int32, device="cuda:0")
        self.compact = None
        if os.environ.get("HARNESS_COMPACT") == "1" and n and seq.shape[1] <= 128:
            # the dense records of the position-hinted pass (gtx_align_batch_planes_compact): the plane rows are made on the device, the
            # records come back as gtx_align_batch would have left them (lib.merge_compact); score() reads them where they are
            stride = (seq.shape[1] + 15) // 16 * 16
            d_planes = torch.zeros(n * stride, dtype=torch.uint8, device="cuda:0")
            gtx.check(gtx.lib().gtx_reads_to_planes(self.ctx.h, d_seq.data_ptr(), seq.shape[1], n, d_planes.data_ptr(), stride, None))
            d_comp = torch.full((n * gtx.COMPACT_WORDS,), -1, dtype=torch.int32, device="cuda:0")
            d_fl = torch.zeros(2 * n, dtype=torch.uint8, device="cuda:0")
            gtx.check(gtx.lib().gtx_align_batch_planes_compact(self.ctx.h, d_planes.data_ptr(), stride, d_meta.data_ptr(), n, d_rec.data_ptr(), rec_words,
                                                               d_comp.data_ptr(), d_fl.data_ptr(), None, None, None, None))
            torch.cuda.synchronize()
            raw, comp, fl = d_rec.cpu().numpy().view(np.uint32), d_comp.cpu().numpy().view(np.uint32), d_fl.cpu().numpy()
            compact_tasks = np.nonzero(fl[0::2] & gtx.TASK_COMPACT)[0]
            assert not raw.reshape(2 * n, rec_words)[2 * compact_tasks].any(), "a compact record's slot was written"
            merged = gtx.merge_compact(raw, comp, fl, n, rec_words)
            self.compact = dict(merged=merged, raw=raw.copy(), comp=comp, fl=fl, n=n, share=len(compact_tasks) / n,
                                d_planes=d_planes, d_meta=d_meta, stride=stride)
            self.d_rec = d_rec
            return merged
        gtx.check(gtx.lib().gtx_align_batch(self.ctx.h, d_seq.data_ptr(), seq.shape[1], d_meta.data_ptr(), n, d_rec.data_ptr(),
                                            rec_words, None))
        torch.cuda.synchronize()
        self.d_rec = d_rec
        return d_rec.cpu().numpy().view(np.uint32)[:n * 2 * rec_words]

    def big_records(self):
        return self.ctx.big_records()

    def exact_pass_tasks(self):
        return self.ctx.exact_pass_tasks()

    def hinted_done(self):
        """forward tasks the position-hinted pass finished in the last align call (needs armed timing)"""
        return self.ctx.kernel_times()[0][2]

    def rewind_big_records(self):
        self.ctx.rewind_big_records()

    def score(self, items, records, n_samples=1, rec_words=REC_WORDS, near=True):
        torch = self.torch
        items = np.ascontiguousarray(items, gtx.SCORE_ITEM)
        acc = Accumulators(self.ctx, n_samples, near=near)
        d_items = self._dev(items)
        devs = [self._dev(a) for a in acc.arrays()]
        buf = acc.buffers([d.data_ptr() for d in devs])
        cp = getattr(self, "compact", None)
        if cp is not None and records is cp["merged"]:  # (the records of the compact call above: scored where the device left them)
            d_rec, d_comp, d_fl = self._dev(cp["raw"]), self._dev(cp["comp"]), self._dev(cp["fl"])
            if os.environ.get("HARNESS_TRIAGED") == "1":
                # gtx_align_batch_planes_triaged + gtx_score_batch_queued: the same reads aligned once more with the items at hand -- the
                # scorer's first stage runs behind the alignment, on a second stream --, the same records, the second stage alone
                n = cp["n"]
                words = np.zeros(len(items), np.uint32)
                gtx.check(gtx.lib().gtx_item_words(items.ctypes.data_as(C.c_void_p), len(items), words.ctypes.data_as(C.c_void_p)))
                use_words = os.environ.get("HARNESS_TRIAGED_WORDS", "1") == "1"
                d_words = self._dev(words)
                d_rec2 = torch.zeros(max(n, 1) * 2 * rec_words, dtype=torch.int32, device="cuda:0")
                d_comp2 = torch.full((n * gtx.COMPACT_WORDS,), -1, dtype=torch.int32, device="cuda:0")
                d_fl2 = torch.zeros(2 * n, dtype=torch.uint8, device="cuda:0")
                d_work = torch.full((len(items) + gtx.WORK_HEADER_WORDS,), -1, dtype=torch.int32, device="cuda:0")
                front, done = torch.cuda.Event(), torch.cuda.Event()
                tail = torch.cuda.Stream()
                front.record()
                done.record()
                torch.cuda.synchronize()
                gtx.check(gtx.lib().gtx_align_batch_planes_triaged(self.ctx.h, cp["d_planes"].data_ptr(), cp["stride"], cp["d_meta"].data_ptr(), n, d_rec2.data_ptr(),
                                                                   rec_words, d_comp2.data_ptr(), d_fl2.data_ptr(), d_items.data_ptr(),
                                                                   d_words.data_ptr() if use_words else None, len(items),
                                                                   gtx.TRIAGE_ITEMS_ARE_READS if os.environ.get("HARNESS_TRIAGED_READS") == "1" else 0, d_work.data_ptr(), None,
                                                                   C.c_void_p(front.cuda_event), C.c_void_p(tail.cuda_stream), C.c_void_p(done.cuda_event)))
                torch.cuda.current_stream().wait_event(done)
                gtx.check(gtx.lib().gtx_score_batch_queued(self.ctx.h, d_items.data_ptr(), len(items), d_rec2.data_ptr(), rec_words, d_comp2.data_ptr(),
                                                           d_fl2.data_ptr(), d_work.data_ptr(), C.byref(buf), None))
                torch.cuda.synchronize()
                fl2 = d_fl2.cpu().numpy()
                assert np.array_equal(fl2, cp["fl"]), "the side array of the second alignment differs"
                merged2 = gtx.merge_compact(d_rec2.cpu().numpy().view(np.uint32), d_comp2.cpu().numpy().view(np.uint32), fl2, n, rec_words)
                assert np.array_equal(merged2, cp["merged"]), "the records of the second alignment differ"
                work = d_work.cpu().numpy().view(np.uint32)
                queued = work[gtx.WORK_HEADER_WORDS:gtx.WORK_HEADER_WORDS + int(work[0])]
                assert int(work[0]) <= len(items) and len(np.unique(queued)) == len(queued) and (queued < len(items)).all()
                self.triaged = dict(queued=np.sort(queued), words=words)
            else:
                gtx.check(gtx.lib().gtx_score_batch_compact(self.ctx.h, d_items.data_ptr(), None, len(items), d_rec.data_ptr(), rec_words,
                                                            d_comp.data_ptr(), d_fl.data_ptr(), C.byref(buf), None))
        else:
            d_rec = self._dev(np.ascontiguousarray(records, np.uint32))
            gtx.check(gtx.lib().gtx_score_batch(self.ctx.h, d_items.data_ptr(), len(items), d_rec.data_ptr(), rec_words, C.byref(buf),
                                                None))
        torch.cuda.synchronize()
        assert self.ctx.error_count() == 0
        for host, dev in zip(acc.arrays(), devs):
            host[...] = dev.cpu().numpy().view(host.dtype)
        return acc

    def score_replay(self, items, records, acc, rec_words=REC_WORDS):
        torch = self.torch
        items = np.ascontiguousarray(items, gtx.SCORE_ITEM)
        d_items = self._dev(items)
        devs = [self._dev(a) for a in acc.arrays()]
        buf = acc.buffers([d.data_ptr() for d in devs])
        n, bad = C.c_uint64(), C.c_uint64()
        cp = getattr(self, "compact", None)
        if cp is not None and records is cp["merged"]:
            d_rec, d_comp, d_fl = self._dev(cp["raw"]), self._dev(cp["comp"]), self._dev(cp["fl"])
            gtx.check(gtx.lib().gtx_scores_replay_compact(self.ctx.h, d_items.data_ptr(), len(items), d_rec.data_ptr(), rec_words, d_comp.data_ptr(),
                                                          d_fl.data_ptr(), C.byref(buf), None, C.byref(n), C.byref(bad)))
        else:
            d_rec = self._dev(np.ascontiguousarray(records, np.uint32))
            gtx.check(gtx.lib().gtx_scores_replay(self.ctx.h, d_items.data_ptr(), len(items), d_rec.data_ptr(), rec_words, C.byref(buf), None,
                                                  C.byref(n), C.byref(bad)))
        torch.cuda.synchronize()
        assert bad.value == 0 and self.ctx.error_count() == 0
        for host, dev in zip(acc.arrays(), devs):
            host[...] = dev.cpu().numpy().view(host.dtype)
        return int(n.value)

    def score_replay_log(self, items, records, acc, item_base=0, rec_words=REC_WORDS):
        items = np.ascontiguousarray(items, gtx.SCORE_ITEM)
        d_items = self._dev(items)
        d_rec = self._dev(np.ascontiguousarray(records, np.uint32))
        devs = [self._dev(a) for a in acc.arrays()]
        buf = acc.buffers([d.data_ptr() for d in devs])
        out = np.zeros(1 << 20, gtx.REPLAY_ENTRY)
        n, bad = C.c_uint64(), C.c_uint64()
        gtx.check(gtx.lib().gtx_scores_replay_log(self.ctx.h, d_items.data_ptr(), len(items), d_rec.data_ptr(), rec_words, None, None, C.byref(buf),
                                                  item_base, None, out.ctypes.data_as(C.c_void_p), len(out), C.byref(n), C.byref(bad)))
        assert bad.value == 0
        return out[:n.value].copy()

    def score_replay_apply(self, acc, entries):
        entries = np.ascontiguousarray(entries, gtx.REPLAY_ENTRY)
        devs = [self._dev(a) for a in acc.arrays()]
        buf = acc.buffers([d.data_ptr() for d in devs])
        n = C.c_uint64()
        gtx.check(gtx.lib().gtx_scores_replay_apply(self.ctx.h, C.byref(buf), entries.ctypes.data_as(C.c_void_p), len(entries), None, C.byref(n)))
        self.torch.cuda.synchronize()
        for host, dev in zip(acc.arrays(), devs):
            host[...] = dev.cpu().numpy().view(host.dtype)
        return int(n.value)

    def calls(self, acc, n_samples):
        torch = self.torch
        devs = [self._dev(a) for a in acc.arrays()]
        buf = acc.buffers([d.data_ptr() for d in devs])
        d_phred = torch.zeros(max(n_samples * self.ctx.total_tri, 1), dtype=torch.uint8, device="cuda:0")
        d_calls = torch.zeros(max(n_samples * self.ctx.n_hap, 1) * gtx.SAMPLE_CALL.itemsize, dtype=torch.uint8, device="cuda:0")
        gtx.check(gtx.lib().gtx_calls_batch(self.ctx.h, C.byref(buf), d_phred.data_ptr(), d_calls.data_ptr(), None))
        torch.cuda.synchronize()
        return (d_phred.cpu().numpy()[:n_samples * self.ctx.total_tri],
                d_calls.cpu().numpy().view(gtx.SAMPLE_CALL)[:n_samples * self.ctx.n_hap])


def read_meta(lengths, flags=None, tid=None, mtid=None, isize=None, pos=None):
    """pos: the position hint of every read (0-based contig position of read base 0), None = unknown (-1)"""
    n = len(lengths)
    m = np.zeros(n, gtx.READ_META)
    m["l_qseq"] = lengths
    m["pos"] = -1 if pos is None else pos
    if flags is not None:
        m["flag"] = flags
    if tid is not None:
        m["tid"] = tid
    if mtid is not None:
        m["mtid"] = mtid
    if isize is not None:
        m["isize"] = isize
    return m


def pack_ragged(reads):
    """list of code arrays of different lengths -> packed [n, stride] + lengths"""
    L = max(len(r) for r in reads)
    codes = np.zeros((len(reads), L), np.uint8)
    for i, r in enumerate(reads):
        codes[i, :len(r)] = r
    return gtx.pack_nibbles(codes), np.array([len(r) for r in reads], np.uint16)


def strip(paths_pairs):
    """drop the fields only the product reports so that results compare with the oracle's structure"""
    out = []
    for pair in paths_pairs:
        out.append(tuple(dict(longest=g["longest"], paths=g["paths"]) for g in pair))
    return out


def canonical_scores(ctx, acc):
    """accumulators + connection log -> the oracle's score word stream (oracle/gto_capi.cpp: gto_scores_dump)"""
    L = gtx.lib()
    nsat = C.c_uint64()
    gtx.check(L.gtx_scores_finalize(_p(acc.log_score), len(acc.log_score), _p(acc.gt_cov), len(acc.gt_cov), _p(acc.hap_u32),
                                    len(acc.hap_u32) // 4, C.byref(nsat)))
    assert nsat.value == 0, "a (haplotype,sample) reached the sequential saturation guard (gtx_scores_replay was not run)"
    assert acc.conn_count[1] == 0, "connection log overflow"
    conn = {}
    log = acc.conn_log[:6 * int(acc.conn_count[0])].reshape(-1, 6)
    for s, h1, b1, h2, b2, c in log:
        d = conn.setdefault((int(s), int(h1), int(b1)), {})
        v = d.setdefault(int(h2), np.zeros(int(ctx.hap_cnum[h2]), np.int64))
        v[int(b2)] += int(c)
    if acc.conn_near is not None:  # dense counters of the near pairs (layout: include/gtx.h, d_conn_near)
        near = acc.conn_near.reshape(acc.n_samples, ctx.total_near) if ctx.total_near else np.zeros((acc.n_samples, 0), np.uint32)
        for s in range(acc.n_samples):
            for h1 in np.nonzero(ctx.near_last > np.arange(ctx.n_hap))[0]:
                h1, last = int(h1), int(ctx.near_last[h1])
                first = int(ctx.allele_off[h1 + 1])
                width = int(ctx.allele_off[last]) + int(ctx.hap_cnum[last]) - first
                block = near[s, int(ctx.near_off[h1]):int(ctx.near_off[h1]) + int(ctx.hap_cnum[h1]) * width].reshape(-1, width)
                if not block.any():
                    continue
                for b1 in range(int(ctx.hap_cnum[h1])):
                    for h2 in range(h1 + 1, last + 1):
                        cell = block[b1, int(ctx.allele_off[h2]) - first:int(ctx.allele_off[h2]) - first + int(ctx.hap_cnum[h2])]
                        if cell.any():
                            d = conn.setdefault((s, h1, b1), {})
                            v = d.setdefault(h2, np.zeros(int(ctx.hap_cnum[h2]), np.int64))
                            v += cell.astype(np.int64)
    out = []
    nh = ctx.n_hap

    def put64(x):
        out.extend([int(x) & 0xFFFFFFFF, int(x) >> 32])

    for h in range(nh):
        cnum, aoff, toff = int(ctx.hap_cnum[h]), int(ctx.allele_off[h]), int(ctx.tri_off[h])
        out.extend([int(ctx.hap_order[h]), cnum, int(acc.stat_u32[h])])
        put64(acc.stat_u64[h])
        for a in range(cnum):
            put64(acc.stat_u64[nh + 2 * (aoff + a)])
            put64(acc.stat_u64[nh + 2 * (aoff + a) + 1])
            out.extend(int(x) for x in acc.stat_u32[nh + 6 * (aoff + a): nh + 6 * (aoff + a) + 6])
        for s in range(acc.n_samples):
            out.extend(int(x) for x in acc.hap_u32[(s * nh + h) * 4:(s * nh + h) * 4 + 4])
            out.extend(int(x) for x in acc.gt_cov[s * ctx.total_allele + aoff: s * ctx.total_allele + aoff + cnum])
            tri = cnum * (cnum + 1) // 2
            out.extend(int(x) for x in acc.log_score[s * ctx.total_tri + toff: s * ctx.total_tri + toff + tri])
            for a in range(cnum):
                d = conn.get((s, h, a), {})
                out.append(len(d))
                for h2 in sorted(d):
                    out.append(h2)
                    out.extend(int(min(x, 0xFFFF)) for x in d[h2])
    return np.array(out, np.uint32)


def canonical_calls(ctx, phred, calls, n_samples):
    """(phred, SAMPLE_CALL) -> the oracle's call word stream (oracle/gto_capi.cpp: gto_calls_dump)"""
    out = []
    for h in range(ctx.n_hap):
        cnum = int(ctx.hap_cnum[h])
        n_tri = cnum * (cnum + 1) // 2
        for s in range(n_samples):
            c = calls[s * ctx.n_hap + h]
            out += [int(c["gt_first"]), int(c["gt_second"]), int(c["gq"]), int(c["ref_total_depth"]), int(c["alt_total_depth"]),
                    int(c["ambiguous_depth"]), int(c["alt_proper_pair_depth"]), n_tri]
            o = s * ctx.total_tri + int(ctx.tri_off[h])
            out += phred[o:o + n_tri].tolist()
    return np.array(out, np.uint32)
