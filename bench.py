#!/usr/bin/env python3
"""bench.py -- reads/s of the alignment + genotype-scoring hot path on MI355X.

A "step" is one pass of the hot path (gtx_align_batch + gtx_score_batch + gtx_calls_batch through libgtx's C ABI) over one batch of
synthetic reads that is already resident in HBM.  Workload at N=1 = BASELINE.json configs[1]: 1 sample, 10 M
synthetic 150 bp reads, one 1 Mb region (chr20:1000001-2000000), SNP-only graph.  With --gpus N every rank gets its own
10 M reads of the same region (weak scaling; graph + index replicated per GPU) and the per-sample score vectors are
summed with one RCCL all-reduce per step.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALGO_BYTES_PER_READ = 3296  # B(L) = 86 + 796*n_k + (L - 31*n_k) at L=150, n_k=4
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
REC_WORDS = 64
REGION_BEGIN = 1000000      # chr20:1000001-2000000
REGION_LEN = 1000000
READ_LEN = 150


def make_reads_on_device(torch, ref_bases, records, n, seed, device, REGION_LEN=REGION_LEN, err_rate=0.005, n_rate=0.001):
    """diploid sample: haplotype 0 = reference, haplotype 1 = reference with a random half of the SNPs; 0.5 % substitution
    errors, 0.1 % N; position sorted; returns packed nibbles [n, 80] (uint8) and read start positions"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ref = torch.from_numpy(ref_bases).to(device)
    hap1 = ref.clone()
    pos = torch.tensor([p - REGION_BEGIN for p, _, _, _ in records], device=device, dtype=torch.long)
    alt = torch.tensor(["ACGT".index(a[0]) for _, _, a, _ in records], device=device, dtype=torch.uint8)
    take = torch.rand(len(records), generator=g, device=device) < 0.5
    hap1[pos[take]] = alt[take]
    haps = torch.stack([ref, hap1])
    out_seq = torch.empty((n, 80), dtype=torch.uint8, device=device)
    out_pos = torch.empty(n, dtype=torch.int64, device=device)
    code_of = torch.tensor([1, 2, 4, 8], dtype=torch.uint8, device=device)
    chunk = 1 << 20
    starts_all = torch.randint(0, REGION_LEN - READ_LEN, (n,), generator=g, device=device)
    starts_all, _ = torch.sort(starts_all)
    ar = torch.arange(READ_LEN, device=device)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        st = starts_all[a:b]
        which = torch.randint(0, 2, (b - a,), generator=g, device=device)
        bases = haps[which[:, None], st[:, None] + ar[None, :]]
        err = torch.rand((b - a, READ_LEN), generator=g, device=device) < err_rate
        shift = torch.randint(1, 4, (b - a, READ_LEN), generator=g, device=device, dtype=torch.uint8)
        bases = torch.where(err, (bases + shift) % 4, bases)
        codes = code_of[bases.long()]
        nmask = torch.rand((b - a, READ_LEN), generator=g, device=device) < n_rate
        codes = torch.where(nmask, torch.full_like(codes, 15), codes)
        out_seq[a:b, :75] = (codes[:, 0::2] << 4) | codes[:, 1::2]
        out_seq[a:b, 75:] = 0
        out_pos[a:b] = st + REGION_BEGIN
    return out_seq, out_pos


def unpack_nibbles(packed, length):
    codes = np.empty((packed.shape[0], length + (length & 1)), np.uint8)
    nb = (length + 1) // 2
    codes[:, 0::2] = packed[:, :nb] >> 4
    codes[:, 1::2] = packed[:, :nb] & 15
    return codes[:, :length]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step")
    ap.add_argument("--snp-every", type=int, default=1000)
    ap.add_argument("--err", type=float, default=0.005, help="experiments only; the reported workload uses 0.005")
    ap.add_argument("--nrate", type=float, default=0.001, help="experiments only; the reported workload uses 0.001")
    ap.add_argument("--region-len", type=int, default=REGION_LEN, help="experiments only; the reported workload is 1 Mb")
    ap.add_argument("--cpu-sample", type=int, default=300_000, help="reads timed through the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from graphtyper_amd import lib as gtx
    from graphtyper_amd import synth
    from graphtyper_amd.dist import reduce_scores

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libgtx has no CPU path")
    if not os.path.exists(gtx.LIB_PATH):
        raise SystemExit("libgtx.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)

    # ---- graph + index (replicated on every GPU) ----
    ref = synth.make_reference(args.region_len, seed=42)
    records = synth.make_snp_records(ref, args.snp_every, seed=7, region_begin=REGION_BEGIN)
    ref_str = synth.bases_to_str(ref)
    t0 = time.time()
    ctx = gtx.Context(gtx.graph_from_records(ref_str, records, region_begin=REGION_BEGIN), device=local_rank)
    t_ctx = time.time() - t0
    n_keys, n_labels = ctx.index_stats()

    # ---- reads, resident in HBM before the timed region ----
    n = args.reads
    d_seq, d_pos = make_reads_on_device(torch, ref, records, n, seed=1234 + rank, device=device, REGION_LEN=args.region_len, err_rate=args.err, n_rate=args.nrate)
    meta = np.zeros(1, gtx.READ_META)
    meta["l_qseq"] = READ_LEN
    d_meta = torch.from_numpy(np.repeat(meta, n).view(np.uint8).reshape(n, 16).copy()).to(device)
    items = np.zeros(n, gtx.SCORE_ITEM)
    items["first"]["align_index"] = np.arange(n, dtype=np.uint32)
    items["first"]["mapq"] = 60
    items["first"]["flag"] = gtx.FLAG_FORWARD_ONLY  # unpaired reads are aligned forward only (what gtx_stream_push sets)
    items["first"]["pos"] = d_pos.cpu().numpy().astype(np.int32)
    items["second"]["align_index"] = gtx.INVALID_ID
    d_items = torch.from_numpy(items.view(np.uint8).reshape(n, gtx.SCORE_ITEM.itemsize).copy()).to(device)
    d_rec = torch.empty(n * 2 * REC_WORDS, dtype=torch.int32, device=device)
    n_samples = 1
    nh = ctx.n_hap
    conn_cap = 1 << 24
    acc = dict(log_score=torch.zeros(n_samples * ctx.total_tri, dtype=torch.int32, device=device),
               gt_cov=torch.zeros(n_samples * ctx.total_allele, dtype=torch.int32, device=device),
               hap_u32=torch.zeros(n_samples * nh * 4, dtype=torch.int32, device=device),
               stat_u64=torch.zeros(nh + 2 * ctx.total_allele, dtype=torch.int64, device=device),
               stat_u32=torch.zeros(nh + 6 * ctx.total_allele, dtype=torch.int32, device=device),
               conn_log=torch.zeros(conn_cap * 6, dtype=torch.int32, device=device),
               conn_count=torch.zeros(2, dtype=torch.int32, device=device),
               conn_near=torch.zeros(max(n_samples * ctx.total_near, 1), dtype=torch.int32, device=device))
    buf = gtx.ScoreBuffers(n_samples, acc["log_score"].data_ptr(), acc["gt_cov"].data_ptr(), acc["hap_u32"].data_ptr(),
                           acc["stat_u64"].data_ptr(), acc["stat_u32"].data_ptr(), acc["conn_log"].data_ptr(),
                           acc["conn_count"].data_ptr(), conn_cap, acc["conn_near"].data_ptr())
    L = gtx.lib()
    stream = torch.cuda.Stream(device=device)
    sp = C.c_void_p(stream.cuda_stream)
    align_ms = []
    d_phred = torch.zeros(max(ctx.total_tri, 1), dtype=torch.uint8, device=device)
    d_calls = torch.zeros(max(ctx.n_hap, 1) * gtx.SAMPLE_CALL.itemsize, dtype=torch.uint8, device=device)

    def step(timed):
        with torch.cuda.stream(stream):
            for name, t in acc.items():
                if name != "conn_log":  # (the log's content is defined by conn_count)
                    t.zero_()
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            gtx.check(L.gtx_align_batch(ctx.h, d_seq.data_ptr(), 80, d_meta.data_ptr(), n, d_rec.data_ptr(), REC_WORDS, sp))
            e1.record(stream)
            gtx.check(L.gtx_score_batch(ctx.h, d_items.data_ptr(), n, d_rec.data_ptr(), REC_WORDS, C.byref(buf), sp))
            if world > 1:
                reduce_scores(dist, [acc["log_score"], acc["gt_cov"], acc["hap_u32"], acc["stat_u64"], acc["stat_u32"], acc["conn_near"]])
            # genotype calls (PL, GT, GQ, depths) from the summed accumulators
            gtx.check(L.gtx_calls_batch(ctx.h, C.byref(buf), d_phred.data_ptr(), d_calls.data_ptr(), sp))
        return (e0, e1)

    ctx.pass_times()  # arms the per-pass HIP events inside gtx_align_batch
    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = [step(True) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    align_ms = [a.elapsed_time(b) for a, b in evs]
    pass_ms, n_pass2 = ctx.pass_times()  # last step: express / general / HBM-table kernels
    t_max = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    dt = float(t_max.item())

    # sanity on the results of the last step: every record must be a result, not an overflow
    rec_head = d_rec.view(n * 2, REC_WORDS)[:, 0]
    n_overflow = int((((rec_head >> 16) & gtx.ST_ERROR_MASK) != 0).sum().item())
    calls = d_calls.cpu().numpy().view(gtx.SAMPLE_CALL)[:ctx.n_hap]
    n_nonref_calls = int((calls["gt_second"] > 0).sum())
    n_aligned = int(((rec_head[0::2] & 0xFFFF) > 0).sum().item())
    errors = ctx.error_count()
    conn_logged, conn_dropped = (int(x) for x in acc["conn_count"].cpu().numpy())

    prof = ctx.profile()
    if rank == 0 and prof[15] > 0:
        names = ["load read", "keys + exact probes", "exact labels", "chain exact", "hamming lookup", "chain hamming",
                 "walk starts", "walk ends", "filters", "record"]
        tot = float(prof[:10].sum())
        sys.stderr.write("phase cycles per read-orientation (profiling build), %d tasks:\n" % prof[15])
        for k, nm in enumerate(names):
            sys.stderr.write("  %-22s %10.0f  %5.1f%%\n" % (nm, prof[k] / float(prof[15]), 100.0 * prof[k] / tot))
        sys.stderr.write("  fast-seeded tasks      %9.1f%%\n" % (100.0 * prof[14] / float(prof[15])))

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = 1000.0 * dt / args.steps
    value = world * n * args.steps / dt
    align_avg_ms = float(np.mean(align_ms))
    # dominant kernel: the express pass (every read-orientation task goes through it)
    express_ms = pass_ms[0] if pass_ms[0] > 0 else align_avg_ms
    # units of that launch: the tasks it completes (what it hands to the general pass is not counted for it)
    n_express = n - n_pass2 if pass_ms[0] > 0 else n
    achieved = ALGO_BYTES_PER_READ * n_express / (express_ms * 1e-3) / 1e9
    traffic = None
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tf):
        try:
            traffic = json.load(open(tf)).get("align_kernel_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "aligned+genotyped reads/sec over 1 Mb graph region; VCF bit-identical",
        "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64/u32 integer (2-bit k-mer keys, byte compares, u32 atomics)", "data": "synthetic",
        "config": {"workload": "cfg2: 1 sample, %d synthetic %d bp reads per GPU, chr20:1000001-2000000 (1 Mb), SNP-only graph "
                               "(1 SNP / %d bp), unpaired, 0.5%% substitutions, 0.1%% N" % (n, READ_LEN, args.snp_every),
                   "reads_per_gpu": n, "index_keys": n_keys, "index_labels": n_labels, "haplotypes": ctx.n_hap,
                   "ctx_create_s": round(t_ctx, 3), "reads_aligned": n_aligned, "reads_overflowed": n_overflow, "nonref_genotype_calls": n_nonref_calls,
                   "score_items_refused": errors, "connections_logged": conn_logged, "connections_dropped": conn_dropped, "parallelism": "reads sharded over %d GPU(s), graph+index replicated" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "kernel": "gtx_align_express4_kernel", "kernel_ms": express_ms,
                     "algorithmic_bytes_per_read": ALGO_BYTES_PER_READ,
                     "align_passes_ms": {"express": pass_ms[0], "general": pass_ms[1], "hbm_tables": pass_ms[2],
                                         "all_three_avg": align_avg_ms, "tasks_handed_to_general": n_pass2, "tasks_completed_by_express": n_express}},
    }
    if world == 1 and not args.no_cpu_baseline:
        from oracle_lib import Oracle
        m = min(args.cpu_sample, n)
        sample = unpack_nibbles(d_seq[:m].cpu().numpy(), READ_LEN)
        spos = d_pos[:m].cpu().numpy()
        oracle = Oracle(ref_str, records, region_begin=REGION_BEGIN)
        g = oracle.genotyper(1, 1)
        t0 = time.perf_counter()
        g.push(list(sample), pos=spos)
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": m / cdt, "unit": "reads/s", "cores": 1, "kind": "port",
                               "sample": "first %d reads of the same workload through oracle/ (C++ restatement), 1 thread, %.1f s" % (m, cdt)}
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
