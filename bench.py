#!/usr/bin/env python3
"""bench.py -- reads/s of the alignment + genotype-scoring hot path on MI355X.

A "step" is one pass of the hot path (gtx_align_batch + gtx_score_batch [+ gtx_scores_reduce] + gtx_calls_batch through
libgtx's C ABI) over one batch of synthetic reads that is already resident in HBM.  Workload at N=1 = BASELINE.json
configs[1] ("cfg2"): 1 sample, 10 M synthetic 150 bp reads, one 1 Mb region (chr20:1000001-2000000), SNP-only graph.
With --gpus N > 1 the workload is BASELINE.json configs[3] ("cfg4"): the reads of 1000 samples over the same region, sharded
by read over the ranks (graph + index replicated per GPU), every rank accumulating into the block of all 1000 samples, and the
ranks' blocks summed with one RCCL all-reduce group per step (gtx_scores_reduce).  --scaling weak (default): --reads per GPU
(8 GPUs x 10 M = cfg4's 80 M reads); --scaling strong: --total-reads (80 M) split N ways.

Launch: `python bench.py --gpus N` starts N ranks itself (torch.distributed.run on 127.0.0.1); under a launcher that
already set WORLD_SIZE (the driver's `python -m torch.distributed.run ... bench.py --gpus N`) it is one of the ranks and
WORLD_SIZE must equal --gpus.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` and `cpu_baseline` objects; at N=1 the
line also carries `config.extra`: the cfg3-like workload (30 samples, merged multi-allelic SNP+indel graph) measured in
the same run.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# Steps in flight need their streams on different hardware queues.  The runtime folds a process's streams onto GPU_MAX_HW_QUEUES of
# them (4 by default), in the order they are made; with four, the stream of the position-hinted pass and the stream of the queues
# behind it now and then share one (the library's scratch makes side streams of its own as calls overlap), and the schedule runs at
# the one-at-a-time rate (with 2 queues always: 1.18 against 0.79 ms per step).  Eight leave room.  (A host application sets the
# same variable before it initialises HIP; an explicit setting of the caller's wins.)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ALGO_BYTES_PER_READ = 3296  # B(L) = 86 + 796*n_k + (L - 31*n_k) at L=150, n_k=4
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
EXCHANGE_STREAM = os.environ.get("GTX_BENCH_EXCHANGE_STREAM", "1") != "0"  # N > 1, staggered schedule: the exchange and the calls on a stream of their own
# Timing is sampled: ONE step in four carries the events that time it -- the pair around an align call here, the library's events
# around every launch of the call (GTX_TIME_EVERY, read by libgtx when it is loaded; gtx_ctx_kernel_times gives the means over the
# timed calls).  An event is a packet the stream carries between two kernels: with every call timed the cfg2 step took 0.712 ms,
# with one in four 0.700, with none 0.696 (round 6, one box, A/B).
os.environ.setdefault("GTX_TIME_EVERY", "4")
EV_EVERY = max(1, int(os.environ.get("GTX_BENCH_EV_EVERY", os.environ["GTX_TIME_EVERY"])))  # staggered schedule: steps per (start, end) event pair around an align call
USE_TASK_FLAGS = os.environ.get("GTX_BENCH_FLAGS", "1") != "0"  # the dense side array of the records (A/B switch)
USE_ITEM_WORDS = os.environ.get("GTX_BENCH_ITEM_WORDS", "1") != "0"  # gtx_score_batch_words (0: gtx_score_batch_flags; A/B)
# dense records of the position-hinted pass (gtx_align_batch_planes_compact / gtx_score_batch_compact; 0: every record in its slot; A/B)
USE_COMPACT = os.environ.get("GTX_BENCH_COMPACT", "1") != "0" and USE_TASK_FLAGS and os.environ.get("GTX_BENCH_PLANES", "1") != "0"
# gtx_align_batch_planes_triaged + gtx_score_batch_queued: the scorer's first stage behind the alignment's short queues, on their stream (0: in front of the scoring; A/B)
USE_TRIAGED = os.environ.get("GTX_BENCH_TRIAGED", "1") != "0" and USE_COMPACT and USE_ITEM_WORDS
# (every item of the bench's read sets is one unpaired read, item i = read i -- add_reads --: the reads' bits instead of side bytes and item words; 0: A/B)
TRIAGE_FLAGS = 1 if os.environ.get("GTX_BENCH_TRIAGE_READS", "1") != "0" else 0  # GTX_TRIAGE_ITEMS_ARE_READS
PLANE_INPUT = os.environ.get("GTX_BENCH_PLANES", "1") != "0"    # reads resident as plane rows (0: BAM nibble rows, repacked inside every call)
REC_WORDS = int(os.environ.get("GTX_BENCH_REC_WORDS", "64"))  # uint32 words of a record slot (rec_words of gtx_align_batch; A/B switch)
REGION_BEGIN = 1000000      # chr20:1000001-2000000
REGION_LEN = 1000000
READ_LEN = 150
METRIC = "aligned+genotyped reads/sec over 1 Mb graph region; VCF bit-identical"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step (weak scaling; --scaling strong: see --total-reads)")
    ap.add_argument("--samples", type=int, default=0,
                    help="samples the reads belong to (0 = the config's: 1 at --gpus 1 = cfg2, 1000 at --gpus N > 1 = cfg4)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = --reads per GPU (8 GPUs x 10 M = the 80 M reads of cfg4), strong = --total-reads split N ways")
    ap.add_argument("--total-reads", type=int, default=80_000_000, help="--scaling strong: reads of the whole job (cfg4: 1000 samples x 80 k)")
    ap.add_argument("--snp-every", type=int, default=1000)
    ap.add_argument("--err", type=float, default=0.005, help="experiments only; the reported workload uses 0.005")
    ap.add_argument("--nrate", type=float, default=0.001, help="experiments only; the reported workload uses 0.001")
    ap.add_argument("--region-len", type=int, default=REGION_LEN, help="experiments only; the reported workload is 1 Mb")
    ap.add_argument("--cpu-sample", type=int, default=300_000, help="reads timed through the CPU oracle on one core (rank 0, N=1)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the all-cores CPU baseline (0 = all host hardware threads, max 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the cfg3-like extra workload")
    ap.add_argument("--no-reduce-check", action="store_true", help="N > 1: skip the check of the summed block against one GPU (rank 0 redoes every rank's reads)")
    ap.add_argument("--lanes", type=int, default=4, help="steps in flight (each with its own records and accumulators); 1 = one after the other.  "
                    "Staggered schedule with four and more lanes: the last lane is the host's slack (Workload.steps_staggered)")
    ap.add_argument("--schedule", choices=["staggered", "lanes"], default="staggered",
                    help="staggered = one stream carries the position-hinted pass of step k and then the scoring of step k-lanes+1, a second "
                         "one the short queues behind every position-hinted pass; lanes = step k entirely on stream k mod lanes")
    ap.add_argument("--read-sets", type=int, default=2, help="resident read sets the steps take in turn (1 = the same reads every step)")
    ap.add_argument("--extra-reads", type=int, default=2_000_000)
    ap.add_argument("--no-hint", action="store_true", help="experiments only: withhold the BAM position from the alignment")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the launch (nccl = RCCL)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch logic only (ranks, process group, one packed all-reduce, max-over-ranks timing): no GPU work, "
                         "the line carries value null; used by the CPU test of the N>1 launch")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------
# launch
# ------------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    """not under a launcher and --gpus N > 1: become the launcher (one process per GPU, rendezvous on 127.0.0.1)"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def init_ranks(args):
    """(rank, local_rank, world, dist | None).  The world size comes from the launcher and has to be what --gpus asked for."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if world == 1:
        return rank, local_rank, world, None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(args.backend)
    if dist.get_world_size() != args.gpus:
        raise SystemExit("bench.py: process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
    return rank, local_rank, world, dist


def max_over_ranks(dist, seconds, device):
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_shape(args, world):
    """(samples, reads of this job per rank list) of the workload: cfg2 at one GPU, cfg4 -- 1000 samples, reads sharded over the
    ranks -- at several"""
    from graphtyper_amd.dist import shard_bounds
    samples = args.samples or (1 if world == 1 else 1000)
    if world > 1 and args.scaling == "strong":
        per_rank = [hi - lo for lo, hi in (shard_bounds(args.total_reads, world, r) for r in range(world))]
    else:
        per_rank = [args.reads] * world
    return samples, per_rank


def dry_run(args):
    """the N>1 control flow of main() without device work: same launch, same job shape (samples, reads per rank), same barrier
    + max-over-ranks timing, and the exchange step of cfg4 -- the packed accumulator block of `samples` samples over the 1 000
    biallelic sites of the cfg2 / cfg4 graph summed over the ranks as one int64 and one int32 tensor (what gtx_scores_reduce
    does with RCCL); one line from rank 0"""
    import torch
    rank, local_rank, world, dist = init_ranks(args)
    samples, per_rank = job_shape(args, world)
    n_hap, tri, alle = 1000, 3000, 2000  # (1 SNP / kb over 1 Mb: gtx_score_layout of that graph)
    n64 = n_hap + 2 * alle
    n32 = samples * (tri + alle + 4 * n_hap) + n_hap + 6 * alle
    t64 = (torch.arange(n64, dtype=torch.int64) % 1009) * (rank + 1)
    t32 = ((torch.arange(n32, dtype=torch.int64) % 65521) * (rank + 1)).to(torch.int32)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    reduce_s = 0.0
    for _ in range(args.steps):
        a64, a32 = t64.clone(), t32.clone()
        r0 = time.perf_counter()
        if dist is not None:
            dist.all_reduce(a64, op=dist.ReduceOp.SUM)
            dist.all_reduce(a32, op=dist.ReduceOp.SUM)
        reduce_s += time.perf_counter() - r0
    if dist is not None:
        dist.barrier()
    local = time.perf_counter() - t0
    dt = max_over_ranks(dist, local, "cpu")
    factor = world * (world + 1) // 2
    ok = bool((a64 == (torch.arange(n64, dtype=torch.int64) % 1009) * factor).all()) and \
        bool((a32.to(torch.int64) == (torch.arange(n32, dtype=torch.int64) % 65521) * factor).all())
    per = gather_floats(dist, 1000.0 * local / max(args.steps, 1), "cpu")
    n_gpus = dist.get_world_size() if dist is not None else 1
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "reads/s", "n_gpus": n_gpus, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": 1000.0 * dt / max(args.steps, 1), "higher_is_better": True,
                          "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dry_run": True, "reduce_ok": ok,
                          "config": {"workload": "none (launch logic only)", "backend": args.backend, "samples": samples, "reads_per_rank": per_rank,
                                     "reduced_bytes_per_step": 8 * n64 + 4 * n32, "reduce_ms": 1000.0 * reduce_s / max(args.steps, 1),
                                     "per_rank_ms_per_step": per}}))
    if dist is not None:
        dist.destroy_process_group()
    return 0 if ok else 1


def gather_floats(dist, x, device):
    """[x of rank 0, x of rank 1, ...] on every rank"""
    import torch
    if dist is None:
        return [float(x)]
    t = torch.tensor([x], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


# ------------------------------------------------------------------------------------------------------------------
# synthetic input
# ------------------------------------------------------------------------------------------------------------------
def make_reads_on_device(torch, ref_bases, records, n, seed, device, REGION_LEN=REGION_LEN, err_rate=0.005, n_rate=0.001, region_begin=REGION_BEGIN):
    """diploid sample: haplotype 0 = reference, haplotype 1 = reference with a random half of the SNPs; 0.5 % substitution
    errors, 0.1 % N; position sorted; returns packed nibbles [n, 80] (uint8) and read start positions"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ref = torch.from_numpy(ref_bases).to(device)
    hap1 = ref.clone()
    pos = torch.tensor([p - region_begin for p, _, _, _ in records], device=device, dtype=torch.long)
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
        out_pos[a:b] = st + region_begin
    return out_seq, out_pos


CFG2_READ_SEED = 1234  # read set k of rank r is drawn with seed CFG2_READ_SEED + r + 7919 k


def cfg2_graph_inputs(synth, region_len=REGION_LEN, snp_every=1000):
    """reference bases, SNP records and the reference as a string of the cfg2 graph (seed 42 / 7)"""
    ref = synth.make_reference(region_len, seed=42)
    records = synth.make_snp_records(ref, snp_every, seed=7, region_begin=REGION_BEGIN)
    return ref, records, synth.bases_to_str(ref)


def unpack_nibbles(packed, length):
    codes = np.empty((packed.shape[0], length + (length & 1)), np.uint8)
    nb = (length + 1) // 2
    codes[:, 0::2] = packed[:, :nb] >> 4
    codes[:, 1::2] = packed[:, :nb] & 15
    return codes[:, :length]


class DevView:
    """a device pointer as something torch.as_tensor understands"""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


# ------------------------------------------------------------------------------------------------------------------
# one workload through the C ABI
# ------------------------------------------------------------------------------------------------------------------
class Workload:
    """resident inputs + accumulators of one (graph, reads) pair; step() = align + score [+ reduce] + calls"""

    def __init__(self, torch, gtx, ctx, device, d_seq, d_pos, n_samples, samples=None, hint=True, conn_cap=1 << 24, lanes=1, read_len=READ_LEN):
        """d_seq: [n, stride] BAM nibble rows on the device; they are repacked ONCE into plane rows (gtx_reads_to_planes, the
        layout the kernels read -- what gtx_stream_push writes on the host side) and only those stay resident"""
        self.torch, self.gtx, self.ctx, self.device = torch, gtx, ctx, device
        self.L = gtx.lib()
        n = int(d_seq.shape[0])
        self.n, self.n_samples = n, n_samples
        self.stride = (int(d_seq.shape[1]) + 15) // 16 * 16  # pitch of the plane rows
        self.hint, self.samples, self.read_len = hint, samples, read_len
        self.rewind = False  # step(): rewind the context's big-record arena first (one step at a time only: the arena is the context's)
        # run(): may the calibration settle for whole steps in flight on streams of their own?  The legs may.  The main workload keeps
        # the staggered schedule: there a launch of the position-hinted pass shares the chip with the short queues of its neighbours
        # only and its own time is what the roofline prices (0.40 ms against 0.36 alone); with whole steps in flight three of those
        # launches overlap EACH OTHER, a launch takes 0.93 ms from its first workgroup to its last, and a per-launch roofline says
        # nothing about the kernel any more -- the faster schedule is reported beside the line (config.extra.whole_steps_in_flight).
        self.allow_whole_steps = False
        self.whole_lanes = None  # the most steps in flight when WHOLE steps are (None: as many as there are lanes; cfg2: three beat four)
        self.sets = []   # resident read sets (bases, meta, score items); step k works on set k mod len(sets)
        self.words = {}  # items' data_ptr -> their compact form (gtx_score_batch_words), or None
        self.steps_done = 0
        self.add_reads(d_seq, d_pos)
        self.align_fn = self.L.gtx_align_batch_planes if PLANE_INPUT else self.L.gtx_align_batch_flags
        self.d_rec = torch.zeros(n * 2 * REC_WORDS, dtype=torch.int32, device=device)
        # dense side array of the records (one byte per task): gtx_align_batch_flags / gtx_score_batch_flags
        self.d_flags = torch.zeros(n * 2, dtype=torch.uint8, device=device) if USE_TASK_FLAGS else None
        self.buf = gtx.ScoreBuffers()
        reduced = C.c_uint64()
        gtx.check(self.L.gtx_scores_alloc(ctx.h, n_samples, conn_cap, C.byref(self.buf), C.byref(reduced)))
        self.reduced_bytes = int(reduced.value)
        self.d_phred = torch.zeros(max(n_samples * ctx.total_tri, 1), dtype=torch.uint8, device=device)
        self.d_calls = torch.zeros(max(n_samples * ctx.n_hap, 1) * gtx.SAMPLE_CALL.itemsize, dtype=torch.uint8, device=device)
        # (GTX_BENCH_PRIO, an A/B switch: "H" gives the stream of the position-hinted passes the higher HIP priority, "T" the tails' --
        #  with GTX_BENCH_SCORE_ON=S the scoring then runs beside them on a stream of the default priority and takes what they leave)
        self.prio = os.environ.get("GTX_BENCH_PRIO", "")
        self.stream = torch.cuda.Stream(device=device, priority=-1 if "H" in self.prio else 0)
        self.sp = C.c_void_p(self.stream.cuda_stream)
        # further lanes (--lanes): a step is still align -> score -> calls in stream order, but step k runs on lane k mod
        # lanes with that lane's stream, records and accumulators, so that the short queues at the end of one step (express,
        # general, scoring: latency-bound, the chip mostly idle) run beside the position-hinted pass of the next
        self.d_compact = torch.zeros(n * 8, dtype=torch.int32, device=device) if USE_COMPACT else None
        # (the work queue of the scorer's second stage, filled behind the alignment: gtx_align_batch_planes_triaged)
        self.d_work = torch.zeros(n + gtx.WORK_HEADER_WORDS, dtype=torch.int32, device=device) if USE_TRIAGED else None
        self.lanes = [dict(stream=self.stream, sp=self.sp, buf=self.buf, d_rec=self.d_rec, d_flags=self.d_flags, d_phred=self.d_phred,
                           d_calls=self.d_calls, d_compact=self.d_compact, d_work=self.d_work)]
        for _ in range(1, max(1, lanes)):
            lane = dict(stream=torch.cuda.Stream(device=device), buf=gtx.ScoreBuffers(),
                        d_rec=torch.zeros(n * 2 * REC_WORDS, dtype=torch.int32, device=device),
                        d_flags=torch.zeros(n * 2, dtype=torch.uint8, device=device) if USE_TASK_FLAGS else None,
                        d_phred=torch.zeros_like(self.d_phred), d_calls=torch.zeros_like(self.d_calls),
                        d_compact=torch.zeros(n * 8, dtype=torch.int32, device=device) if USE_COMPACT else None,
                        d_work=torch.zeros(n + gtx.WORK_HEADER_WORDS, dtype=torch.int32, device=device) if USE_TRIAGED else None)
            lane["sp"] = C.c_void_p(lane["stream"].cuda_stream)
            gtx.check(self.L.gtx_scores_alloc(ctx.h, n_samples, conn_cap, C.byref(lane["buf"]), C.byref(reduced)))
            self.lanes.append(lane)
        # (staggered schedule: the short queues behind every position-hinted pass; GTX_BENCH_TAILS=2: those of consecutive steps on
        #  two streams -- measured the same)
        self.tail_streams = [torch.cuda.Stream(device=device, priority=-1 if "T" in self.prio else 0) for _ in range(max(1, int(os.environ.get("GTX_BENCH_TAILS", "1"))))]
        for ln in self.lanes:
            ln["front"], ln["aligned"], ln["scored"] = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
            for ev in (ln["front"], ln["aligned"], ln["scored"]):
                ev.record(self.stream)  # (creates the HIP event: gtx_align_batch_planes_staged takes its handle)
        self.staggered = False
        self.reduce_events = []  # HIP events around the exchange step of every step since the last run()
        self.comm = None       # ncclComm_t made through gtx_comm_init_rank
        self.dist = None       # fallback: torch.distributed on views of the packed block
        self.reduce_kind = None

    def fresh_streams(self):
        """new HIP streams (and events) for every lane and for the tails: where the runtime puts a stream on the hardware queues is
        decided when it is made"""
        torch = self.torch
        torch.cuda.synchronize()
        for k, ln in enumerate(self.lanes):
            ln["stream"] = torch.cuda.Stream(device=self.device, priority=-1 if (k == 0 and "H" in self.prio) else 0)
            ln["sp"] = C.c_void_p(ln["stream"].cuda_stream)
        self.stream, self.sp = self.lanes[0]["stream"], self.lanes[0]["sp"]
        self.tail_streams = [torch.cuda.Stream(device=self.device, priority=-1 if "T" in self.prio else 0) for _ in self.tail_streams]
        for ln in self.lanes:
            ln["front"], ln["aligned"], ln["scored"] = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
            for ev in (ln["front"], ln["aligned"], ln["scored"]):
                ev.record(self.stream)
        torch.cuda.synchronize()

    def add_reads(self, d_seq, d_pos):
        """another resident read set of the same size: the steps take the sets in turn, so that no step finds its own
        reads (or their records) in a cache"""
        torch, gtx, n = self.torch, self.gtx, self.n
        assert int(d_seq.shape[0]) == n and (int(d_seq.shape[1]) + 15) // 16 * 16 == self.stride
        if PLANE_INPUT:
            d_planes = torch.empty((n, self.stride), dtype=torch.uint8, device=self.device)
            gtx.check(self.L.gtx_reads_to_planes(self.ctx.h, d_seq.data_ptr(), int(d_seq.shape[1]), n, d_planes.data_ptr(), self.stride, None))
            torch.cuda.synchronize()
            d_seq = d_planes
        pos_host = d_pos.cpu().numpy().astype(np.int32)
        meta = np.zeros(n, gtx.READ_META)
        meta["l_qseq"] = self.read_len
        meta["flag"] = gtx.FLAG_FORWARD_ONLY  # unpaired reads: no reverse orientation, and nobody reads its (empty) record
        meta["pos"] = pos_host if self.hint else -1
        d_meta = torch.from_numpy(meta.view(np.uint8).reshape(n, gtx.READ_META.itemsize).copy()).to(self.device)
        items = np.zeros(n, gtx.SCORE_ITEM)
        items["first"]["align_index"] = np.arange(n, dtype=np.uint32)
        items["first"]["mapq"] = 60
        items["first"]["flag"] = gtx.FLAG_FORWARD_ONLY  # unpaired reads are aligned forward only (what gtx_stream_push sets)
        items["first"]["pos"] = pos_host
        items["second"]["align_index"] = gtx.INVALID_ID
        if self.samples is not None:
            items["sample"] = self.samples
        d_items = torch.from_numpy(items.view(np.uint8).reshape(n, gtx.SCORE_ITEM.itemsize).copy()).to(self.device)
        # (the items' compact form for the scorer's first stage -- gtx_item_words: 4 bytes per item beside the item's 40)
        d_words = torch.from_numpy(gtx.item_words(items).view(np.uint8).copy()).to(self.device) if USE_TASK_FLAGS and USE_ITEM_WORDS else None
        self.sets.append((d_seq, d_meta, d_items))
        self.words[int(d_items.data_ptr())] = d_words

    def close(self):
        if self.comm is not None:
            self.L.gtx_comm_destroy(self.comm)
            self.comm = None
        for lane in self.lanes:
            self.L.gtx_scores_free(self.ctx.h, C.byref(lane["buf"]))

    def setup_reduce(self, dist, rank, world, local_rank):
        """the exchange step: an RCCL communicator for gtx_scores_reduce (id from rank 0 through the process group); if that
        cannot be had on every rank, the same sums through torch.distributed (also RCCL) on views of the packed block"""
        torch, gtx, L = self.torch, self.gtx, self.L
        ident = torch.zeros(128, dtype=torch.uint8, device=self.device)
        ok = 1
        if rank == 0:
            raw = (C.c_uint8 * 128)()
            ok = 1 if L.gtx_comm_unique_id(raw) == 0 else 0
            ident = torch.tensor(list(raw), dtype=torch.uint8, device=self.device)
        flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            dist.broadcast(ident, src=0)
            raw = (C.c_uint8 * 128)(*ident.cpu().tolist())
            comm = C.c_void_p()
            ok = 1 if L.gtx_comm_init_rank(raw, world, rank, local_rank, C.byref(comm)) == 0 else 0
            flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                self.comm, self.reduce_kind = comm, "gtx_scores_reduce: one RCCL group (u64 + u32 all-reduce) on the packed block"
                return
            if ok:
                L.gtx_comm_destroy(comm)
        sys.stderr.write("[bench] rank %d: gtx_scores_reduce unavailable (%s): falling back to torch.distributed\n" %
                         (rank, L.gtx_last_error().decode()))
        n64 = self.ctx.n_hap + 2 * self.ctx.total_allele
        n32 = (self.reduced_bytes - 8 * n64) // 4
        for ln in self.lanes:
            ln["t64"] = torch.as_tensor(DevView(ln["buf"].d_stat_u64, n64, "<i8"), device=self.device)
            ln["t32"] = torch.as_tensor(DevView(ln["buf"].d_log_score, n32, "<i4"), device=self.device)
        self.dist, self.reduce_kind = dist, "torch.distributed all_reduce x2 on the packed block"

    def step(self, lane=0, zero=True):
        """one step on `lane`; zero=False: the accumulators keep what they hold (several read sets into one block)"""
        gtx, L, ctx = self.gtx, self.L, self.ctx
        torch = self.torch
        ln = self.lanes[lane]
        stream, sp, buf, d_rec, d_flags = ln["stream"], ln["sp"], ln["buf"], ln["d_rec"], ln["d_flags"]
        d_seq, d_meta, d_items = self.sets[self.steps_done % len(self.sets)]
        self.steps_done += 1
        with torch.cuda.stream(stream):
            if self.rewind:  # (a step is a region's worth of records: the arena of the records longer than a slot starts over)
                gtx.check(L.gtx_ctx_big_records_rewind(ctx.h, sp))
            if zero:
                gtx.check(L.gtx_scores_zero(ctx.h, C.byref(buf), sp))
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            fl = d_flags.data_ptr() if d_flags is not None else None
            w = self.words.get(int(d_items.data_ptr()))
            triaged = ln["d_work"] is not None and w is not None
            if triaged:
                gtx.check(L.gtx_align_batch_planes_triaged(ctx.h, d_seq.data_ptr(), self.stride, d_meta.data_ptr(), self.n, d_rec.data_ptr(), REC_WORDS,
                                                           ln["d_compact"].data_ptr(), fl, d_items.data_ptr(), w.data_ptr(), self.n, TRIAGE_FLAGS, ln["d_work"].data_ptr(),
                                                           sp, None, None, None))
            elif ln["d_compact"] is not None:
                gtx.check(L.gtx_align_batch_planes_compact(ctx.h, d_seq.data_ptr(), self.stride, d_meta.data_ptr(), self.n, d_rec.data_ptr(), REC_WORDS,
                                                           ln["d_compact"].data_ptr(), fl, sp, None, None, None))
            else:
                gtx.check(self.align_fn(ctx.h, d_seq.data_ptr(), self.stride, d_meta.data_ptr(), self.n, d_rec.data_ptr(), REC_WORDS, fl, sp))
            e1.record(stream)
            self._score_call(d_items, d_rec, fl, buf, sp, ln["d_compact"], ln["d_work"] if triaged else None)
            if self.comm is not None or self.dist is not None:
                assert lane == 0  # (one communicator: the exchange steps of two streams must not interleave)
                self._reduce(ln, stream, sp)
            # genotype calls (PL, GT, GQ, depths) from the summed accumulators
            gtx.check(L.gtx_calls_batch(ctx.h, C.byref(buf), ln["d_phred"].data_ptr(), ln["d_calls"].data_ptr(), sp))
        return e0, e1

    def _score_call(self, d_items, d_rec, fl, buf, sp, d_compact=None, d_work=None):
        w = self.words.get(int(d_items.data_ptr()))
        if d_work is not None:  # (the first stage ran behind the alignment: gtx_align_batch_planes_triaged)
            self.gtx.check(self.L.gtx_score_batch_queued(self.ctx.h, d_items.data_ptr(), self.n, d_rec.data_ptr(), REC_WORDS, d_compact.data_ptr(), fl,
                                                         d_work.data_ptr(), C.byref(buf), sp))
        elif d_compact is not None:
            self.gtx.check(self.L.gtx_score_batch_compact(self.ctx.h, d_items.data_ptr(), w.data_ptr() if w is not None else None, self.n, d_rec.data_ptr(),
                                                          REC_WORDS, d_compact.data_ptr(), fl, C.byref(buf), sp))
        elif w is not None and fl is not None:
            self.gtx.check(self.L.gtx_score_batch_words(self.ctx.h, d_items.data_ptr(), w.data_ptr(), self.n, d_rec.data_ptr(), REC_WORDS, fl, C.byref(buf), sp))
        else:
            self.gtx.check(self.L.gtx_score_batch_flags(self.ctx.h, d_items.data_ptr(), self.n, d_rec.data_ptr(), REC_WORDS, fl, C.byref(buf), sp))

    def _score(self, ln, stream, after):
        """score + calls of the step whose records lane `ln` holds, on `stream`, behind the events `after`"""
        gtx, L, ctx, torch = self.gtx, self.L, self.ctx, self.torch
        sp = C.c_void_p(stream.cuda_stream)
        with torch.cuda.stream(stream):
            for ev in after:
                stream.wait_event(ev)
            fl = ln["d_flags"].data_ptr() if ln["d_flags"] is not None else None
            exchange = self.comm is not None or self.dist is not None
            if exchange and EXCHANGE_STREAM:
                stream.wait_event(ln["scored"])  # (the lane's block was last read by the calls of its step before, on the exchange's stream)
            if not ln.pop("zeroed", False):  # (steps_staggered has zeroed the block on the tail stream, in front of the step's short queues)
                gtx.check(L.gtx_scores_zero(ctx.h, C.byref(ln["buf"]), sp))  # (on the tail stream BEHIND the step's short queues: 0.737 against 0.712 ms per step, round 6's first session)
            self._score_call(ln["items"], ln["d_rec"], fl, ln["buf"], sp, ln["d_compact"], ln["d_work"] if ln.get("triaged") else None)
            if exchange and EXCHANGE_STREAM:
                # N > 1: the sum over the ranks and the calls from the summed block go to a stream of their own -- the exchange is
                # 36 MB over xGMI (cfg4: 1000 samples), link-bound and all but idle on the CUs, and on the stream that carries the
                # position-hinted passes it held the next one back for its whole length.  One communicator, one stream: the
                # exchange steps of the steps keep their order.  (GTX_BENCH_EXCHANGE_STREAM=0: on the scoring's stream, as before.)
                if not hasattr(self, "exchange_stream"):
                    self.exchange_stream = torch.cuda.Stream(device=self.device)
                    self.exchange_ready = [torch.cuda.Event() for _ in self.lanes]
                R = self.exchange_stream
                ready = self.exchange_ready[self.lanes.index(ln)]
                ready.record(stream)
                with torch.cuda.stream(R):
                    R.wait_event(ready)
                    spR = C.c_void_p(R.cuda_stream)
                    self._reduce(ln, R, spR)
                    gtx.check(L.gtx_calls_batch(ctx.h, C.byref(ln["buf"]), ln["d_phred"].data_ptr(), ln["d_calls"].data_ptr(), spR))
                    ln["scored"].record(R)
                return
            self._reduce(ln, stream, sp)
            gtx.check(L.gtx_calls_batch(ctx.h, C.byref(ln["buf"]), ln["d_phred"].data_ptr(), ln["d_calls"].data_ptr(), sp))
            ln["scored"].record(stream)

    def _reduce(self, ln, stream, sp):
        """the exchange step of a step whose accumulators lane `ln` holds (N > 1), on `stream` (inside `with torch.cuda.stream`)"""
        if self.comm is None and self.dist is None:
            return
        torch = self.torch
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record(stream)
        if self.comm is not None:
            self.gtx.check(self.L.gtx_scores_reduce(self.ctx.h, C.byref(ln["buf"]), self.comm, sp))
        else:
            self.dist.all_reduce(ln["t64"], op=self.dist.ReduceOp.SUM)
            self.dist.all_reduce(ln["t32"], op=self.dist.ReduceOp.SUM)
        r1.record(stream)
        self.reduce_events.append((r0, r1))

    def steps_staggered(self, steps):
        """`steps` steps, three in flight on two streams.  Stream H carries what fills the chip, one kernel after the other:
        the position-hinted pass of step k, then the scoring of step k-2.  Stream T carries the short queues behind every
        position-hinted pass (gtx_align_batch_planes_staged: express, general, HBM tables -- 0.35 % of the reads,
        latency-bound, most CUs idle), which drain beside H's kernels.  Every step is still align -> score -> calls on its own
        records and accumulators; what is staggered is which step's work the chip sees when.  Returns the (start, end)
        events of every align call (start on H, end on T)."""
        gtx, L, ctx, torch = self.gtx, self.L, self.ctx, self.torch
        H = self.lanes[0]["stream"]
        spH = C.c_void_p(H.cuda_stream)
        evs, flight = [], []
        n_l = len(self.lanes)
        # How far the scoring of a step trails its position-hinted pass: n_l - 1 steps with three lanes (rounds 3-5).  The host
        # paces itself on the lane it is about to reuse (below), and with depth = lanes that lane's scoring is the LAST thing queued
        # on H.  Round 6 asked whether the 45 us between `calls` and the next position-hinted pass (kernel trace) were the host
        # waking up: with four lanes and the same depth of three -- the lane the host waits for scored a whole step earlier -- the
        # step is the same (0.729 against 0.728 ms): the gap was the counter reset in front of the pass and the packets around it
        # (gone since: CallScratch::d_counter_sets), not the host.  Three lanes stay the default.  (GTX_BENCH_DEPTH: A/B)
        depth = int(os.environ.get("GTX_BENCH_DEPTH", "0")) or (n_l - 1 if n_l >= 4 else n_l)
        depth = max(1, min(depth, n_l))
        pace = os.environ.get("GTX_BENCH_PACE", "1") != "0"
        score_on = os.environ.get("GTX_BENCH_SCORE_ON", "H")
        zero_on_tail = os.environ.get("GTX_BENCH_ZERO_ON", "H") == "T" and pace  # (A/B switch, round 6: "T" zeroes the lane's block on the tail stream in front of the step -- 0.571 against 0.569 ms, no gain: "H", in front of the scoring, stays)
        if score_on == "S" and not hasattr(self, "score_stream"):
            self.score_stream = torch.cuda.Stream(device=self.device)
        for _ in range(steps):
            ln = self.lanes[self.steps_done % n_l]
            if pace:
                # The host stays at most n_l steps ahead of the device: the lane's last scoring is through before its next step is
                # queued.  Unpaced, the host queues ten steps in a millisecond, every call finds all earlier ones still in flight
                # and the library makes a new scratch for it (queues for 10 M reads: hipMalloc in the middle of the timed region) --
                # now and then the same schedule ran at half the speed.  (1.5 % slower than unpaced when that does not happen.)
                ln["scored"].synchronize()
            T = self.tail_streams[self.steps_done % len(self.tail_streams)]
            spT = C.c_void_p(T.cuda_stream)
            d_seq, d_meta, d_items = self.sets[self.steps_done % len(self.sets)]
            self.steps_done += 1
            if zero_on_tail:
                # the lane's accumulators are zeroed on the tail stream IN FRONT of the step's short queues (the host has waited for the
                # lane's last calls above; the step's `aligned` event lies behind it): a fill and the packets around it off the stream
                # that carries the position-hinted passes and the scoring
                with torch.cuda.stream(T):
                    gtx.check(L.gtx_scores_zero(ctx.h, C.byref(ln["buf"]), spT))
                ln["zeroed"] = True
            with torch.cuda.stream(H):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                sample = self.steps_done % EV_EVERY == 0  # (the wall time of an align call from one step in EV_EVERY: an event is a packet between two kernels)
                if sample:
                    e0.record(H)
                fl = ln["d_flags"].data_ptr() if ln["d_flags"] is not None else None
                w = self.words.get(int(d_items.data_ptr()))
                ln["triaged"] = ln["d_work"] is not None and w is not None
                if ln["triaged"]:
                    gtx.check(L.gtx_align_batch_planes_triaged(ctx.h, d_seq.data_ptr(), self.stride, d_meta.data_ptr(), self.n, ln["d_rec"].data_ptr(),
                                                               REC_WORDS, ln["d_compact"].data_ptr(), fl, d_items.data_ptr(), w.data_ptr(), self.n, TRIAGE_FLAGS,
                                                               ln["d_work"].data_ptr(), spH, C.c_void_p(ln["front"].cuda_event), spT,
                                                               C.c_void_p(ln["aligned"].cuda_event)))
                elif ln["d_compact"] is not None:
                    gtx.check(L.gtx_align_batch_planes_compact(ctx.h, d_seq.data_ptr(), self.stride, d_meta.data_ptr(), self.n, ln["d_rec"].data_ptr(),
                                                               REC_WORDS, ln["d_compact"].data_ptr(), fl, spH, C.c_void_p(ln["front"].cuda_event), spT,
                                                               C.c_void_p(ln["aligned"].cuda_event)))
                else:
                    gtx.check(L.gtx_align_batch_planes_staged(ctx.h, d_seq.data_ptr(), self.stride, d_meta.data_ptr(), self.n, ln["d_rec"].data_ptr(),
                                                              REC_WORDS, fl, spH, C.c_void_p(ln["front"].cuda_event), spT,
                                                              C.c_void_p(ln["aligned"].cuda_event)))
            if sample:
                with torch.cuda.stream(T):
                    e1.record(T)
                evs.append((e0, e1))
            ln["items"] = d_items
            if score_on == "T":  # (experiment: the step's scoring and calls behind its own queues, on the tail stream)
                self._score(ln, T, [ln["aligned"]])
                continue
            if score_on == "S":  # (experiment: on a stream of its own, behind the step's last alignment pass)
                self._score(ln, self.score_stream, [ln["aligned"]])
                continue
            flight.append(ln)
            if len(flight) == depth:  # (the oldest step in flight: its queues had the last depth - 1 position-hinted passes to drain)
                old = flight.pop(0)
                self._score(old, H, [old["aligned"]])
        for old in flight:
            self._score(old, H, [old["aligned"]])
        return evs

    def run(self, steps, warmup, dist):
        """W untimed steps, then exactly K timed steps bracketed by barrier + synchronize; returns (seconds, align ms list)"""
        torch = self.torch
        self.ctx.pass_times()  # arms the per-pass HIP events inside gtx_align_batch
        exchange = self.comm is not None or self.dist is not None
        stag = self.staggered and len(self.lanes) >= 2 and PLANE_INPUT
        # (with an exchange step in every step -- N > 1 -- only the staggered schedule keeps steps in flight: its scoring, and with
        #  it the exchange, is all on one stream; one communicator takes one collective at a time)
        n_lanes = len(self.lanes) if (stag or not exchange) else 1
        # (setup, not steps of the run: the first calls that are in flight together allocate their scratch inside the library)
        self.calibration = None
        if stag and os.environ.get("GTX_BENCH_CALIBRATE", "1") == "0":
            # (tools/profile.sh: under a kernel trace the calibration's other schedules -- whole steps in flight: three launches of the
            #  position-hinted pass side by side, a millisecond each -- would be in the trace's averages)
            self.steps_staggered(2 * n_lanes)
            self.steps_done -= 2 * n_lanes
            torch.cuda.synchronize()
            self.calibration = {"skipped": "GTX_BENCH_CALIBRATE=0: the staggered schedule as it is"}
        elif stag:
            self.steps_staggered(2 * n_lanes)  # (twice round: every lane's scratch and the context's exact-pass slabs exist afterwards)
            self.steps_done -= 2 * n_lanes
            torch.cuda.synchronize()
            # Setup as well: which schedule runs faster here -- a few steps each way, the slower rank decides for all.  (Steps in
            # flight depend on how the runtime folds streams onto hardware queues and on what else the host does; one step at
            # a time does not.)
            def timed(fn, k):
                torch.cuda.synchronize()
                t = time.perf_counter()
                fn(k)
                torch.cuda.synchronize()
                self.steps_done -= k
                return (time.perf_counter() - t) / k
            t_stag = min(timed(self.steps_staggered, 2 * n_lanes) for _ in range(3))
            t_one = min(timed(lambda k: [self.step(0) for _ in range(k)], 4) for _ in range(3))
            # (Round 5: one process in a dozen measured the staggered schedule at 2.7 ms per step, three times the one-at-a-time rate,
            #  in all three tries -- a state of that process' streams, not of the box: the run before it on the same box had 0.75.
            #  New streams get new places on the hardware queues: the schedule is given two more chances on fresh ones before the
            #  run settles for one step at a time.)
            # (Round 6: one process in ten still ran it at the one-at-a-time rate AFTER winning the comparison by a hair -- 0.7645 against
            #  0.771 ms, where a schedule whose streams overlap takes 0.59: the headline of that process was 12.97 G reads/s instead of
            #  16.9.  "Faster" is not the test: the schedule has to be clearly faster than one step at a time -- at least a tenth -- or its
            #  streams share a hardware queue; up to six sets of fresh streams are tried, which leaves one process in a million.)
            retries = 0
            while t_stag > 0.9 * t_one and retries < 6 and dist is None:
                retries += 1
                self.fresh_streams()
                self.steps_staggered(2 * n_lanes)
                self.steps_done -= 2 * n_lanes
                t_stag = min(timed(self.steps_staggered, 2 * n_lanes) for _ in range(3))
            # (round 5: and whole steps in flight, each on its lane's stream -- on graphs whose time is behind the position-hinted pass the
            #  tails of three steps overlap better that way: cfg3 1.50 ms per step against 1.68 staggered; on cfg2 it is the slower one)
            t_lanes = float("inf")
            if not exchange:
                n_whole = min(n_lanes, self.whole_lanes or n_lanes)

                def whole_steps(k):
                    for i in range(k):
                        self.step(i % n_whole)
                whole_steps(n_lanes)
                self.steps_done -= n_lanes
                t_lanes = min(timed(whole_steps, 2 * n_lanes) for _ in range(3))
                # (the same remedy as above where this schedule is allowed and did not come out clearly ahead: on one box in three
                #  it ran cfg3 at 1.69 ms per step instead of 1.45 -- where the runtime puts three streams on the hardware queues)
                lane_retries = 0
                while self.allow_whole_steps and t_lanes > 0.93 * min(t_stag, t_one) and lane_retries < 2 and dist is None:
                    lane_retries += 1
                    self.fresh_streams()
                    whole_steps(2 * n_lanes)
                    self.steps_done -= 2 * n_lanes
                    t_again = min(timed(whole_steps, 2 * n_lanes) for _ in range(3))
                    if t_again < t_lanes:
                        t_lanes = t_again
                    else:  # (no better: the staggered figure above belongs to the streams before -- measure it on these)
                        self.steps_staggered(2 * n_lanes)
                        self.steps_done -= 2 * n_lanes
                        t_stag = min(timed(self.steps_staggered, 2 * n_lanes) for _ in range(3))
                retries += lane_retries
            both = torch.tensor([t_stag, t_one], dtype=torch.float64, device=self.device)
            if dist is not None:
                dist.all_reduce(both, op=dist.ReduceOp.MAX)
            t_stag, t_one = (float(x) for x in both.cpu())
            self.calibration = {"staggered_ms_per_step": 1000.0 * t_stag, "one_at_a_time_ms_per_step": 1000.0 * t_one, "fresh_stream_retries": retries}
            if t_lanes != float("inf"):
                self.calibration["whole_steps_on_streams_of_their_own_ms_per_step"] = 1000.0 * t_lanes
            # (... and a process whose staggered schedule still does not overlap after that takes whole steps in flight when THEY do)
            misfired = t_stag > 0.9 * t_one and t_lanes < 0.95 * min(t_stag, t_one)
            if (self.allow_whole_steps or misfired) and t_lanes < t_stag and t_lanes < t_one:
                stag = False
                self.staggered = False
                if misfired:
                    self.calibration["staggered_schedule_did_not_overlap"] = "whole steps in flight on streams of their own run the timed region"
            elif t_stag > t_one:
                stag, n_lanes = False, 1
                self.staggered = False
        if stag:
            if warmup:
                self.steps_staggered(warmup)
        else:
            n_lanes = min(n_lanes, self.whole_lanes or n_lanes)
            for lane in range(1, n_lanes):
                self.step(lane)
                self.steps_done -= 1
            for k in range(warmup):
                self.step(k % n_lanes)
        torch.cuda.synchronize()
        self.ctx.pass_times()  # (a query: the pass times asked for behind the timed steps are the mean over exactly those)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        self.reduce_events = []
        self.used_lanes = n_lanes
        t0 = time.perf_counter()
        evs = self.steps_staggered(steps) if stag else [self.step(k % n_lanes) for k in range(steps)]
        torch.cuda.synchronize()
        self.local_s = time.perf_counter() - t0  # this rank's own time for its K steps (before waiting for the others)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        self.reduce_ms = float(np.mean([a.elapsed_time(b) for a, b in self.reduce_events])) if self.reduce_events else 0.0
        return max_over_ranks(dist, dt, self.device), [a.elapsed_time(b) for a, b in evs]

    def replay_across_ranks(self, dist, item_base):
        """N > 1, behind a step whose block holds the sums over the ranks: the cells at the saturation guard of explain_to_score
        (haplotype.cpp:560) are replayed in the stream order of ALL ranks' items -- every rank logs its own calls on them
        (gtx_scores_replay_log), the logs are gathered, every rank replays them into its block (gtx_scores_replay_apply).
        Returns (cells replayed, entries of all ranks)."""
        gtx, L, ctx, torch = self.gtx, self.L, self.ctx, self.torch
        ln = self.lanes[0]
        d_seq, d_meta, d_items = self.sets[(self.steps_done - 1) % len(self.sets)]
        cap = 1 << 22
        out = np.zeros(cap, gtx.REPLAY_ENTRY)
        n, bad = C.c_uint64(), C.c_uint64()
        comp = ln["d_compact"].data_ptr() if ln["d_compact"] is not None else None
        fl = ln["d_flags"].data_ptr() if ln["d_compact"] is not None else None
        gtx.check(L.gtx_scores_replay_log(ctx.h, d_items.data_ptr(), self.n, ln["d_rec"].data_ptr(), REC_WORDS, comp, fl, C.byref(ln["buf"]), item_base,
                                          ln["sp"], out.ctypes.data_as(C.c_void_p), cap, C.byref(n), C.byref(bad)))
        logs = [None] * dist.get_world_size()
        dist.all_gather_object(logs, out[:n.value].tobytes())
        entries = np.ascontiguousarray(np.concatenate([np.frombuffer(x, gtx.REPLAY_ENTRY) for x in logs]))
        done = C.c_uint64()
        gtx.check(L.gtx_scores_replay_apply(ctx.h, C.byref(ln["buf"]), entries.ctypes.data_as(C.c_void_p), len(entries), ln["sp"], C.byref(done)))
        return int(done.value), len(entries)

    def sample_names(self):
        return ["SAMP%04d" % i for i in range(self.n_samples)]

    def vcf_text(self):
        """gtx_vcf_records over the results of the last step (host side): the region's VCF records as bytes, and the calls"""
        gtx, ctx = self.gtx, self.ctx
        self.torch.cuda.synchronize()
        calls = self.d_calls.cpu().numpy().view(gtx.SAMPLE_CALL)[:self.n_samples * ctx.n_hap]
        nh, ta = ctx.n_hap, ctx.total_allele
        text = ctx.vcf_records("chr20", self.sample_names(),
                               gtx.download(self.buf.d_gt_cov, np.uint32, self.n_samples * ta), gtx.download(self.buf.d_stat_u64, np.uint64, nh + 2 * ta),
                               gtx.download(self.buf.d_stat_u32, np.uint32, nh + 6 * ta), self.d_phred.cpu().numpy()[:self.n_samples * ctx.total_tri], calls)
        return text, calls

    def vcf_final_text(self):
        """gtx_vcf_records_final over the results of the last step: the records of the file genotype() ends with (vcf_merge_and_break with
        the break-down; on a SNP graph both of the reference's modes write the same)"""
        gtx, ctx = self.gtx, self.ctx
        self.torch.cuda.synchronize()
        calls = self.d_calls.cpu().numpy().view(gtx.SAMPLE_CALL)[:self.n_samples * ctx.n_hap]
        nh, ta = ctx.n_hap, ctx.total_allele
        return ctx.vcf_records_final("chr20", self.sample_names(),
                                     gtx.download(self.buf.d_gt_cov, np.uint32, self.n_samples * ta), gtx.download(self.buf.d_stat_u64, np.uint64, nh + 2 * ta),
                                     gtx.download(self.buf.d_stat_u32, np.uint32, nh + 6 * ta), self.d_phred.cpu().numpy()[:self.n_samples * ctx.total_tri], calls,
                                     no_variant_overlapping=True)

    def calls_checksum(self, read_set=0):
        """One more (untimed) step over resident read set `read_set`, then the SHA-256 of the VCF text gtx_vcf_records writes
        from its results -- every site's GT, AD, DP, GQ, PL and INFO statistics of every sample.  tests/test_gpu_full_size.py
        pushes the same reads (same seed) through the CPU oracle, all of them, and must arrive at the same digest."""
        import hashlib
        self.steps_done = read_set
        self.step()
        text, _ = self.vcf_text()
        final = self.vcf_final_text()
        return {"vcf_sha256": hashlib.sha256(text).hexdigest(), "vcf_bytes": len(text), "read_set": read_set,
                "final_vcf_sha256": hashlib.sha256(final).hexdigest(), "final_vcf_bytes": len(final), "final_vcf_records": final.count(b"\n") - 1,
                "what": "sha256 of the region's VCF records (gtx_vcf_records, column line first) after one step over read set %d; final_*: of the "
                        "records of the file genotype() ends with (gtx_vcf_records_final: vcf_merge_and_break with the break-down)" % read_set}

    def block_digest(self, lane=0):
        """SHA-256 of the packed accumulator block of `lane` (gtx_scores_alloc: [stat_u64 | the u32 sections], what
        gtx_scores_reduce sums over the ranks)"""
        import hashlib
        self.torch.cuda.synchronize()
        raw = self.gtx.download(self.lanes[lane]["buf"].d_stat_u64, np.uint8, self.reduced_bytes)
        return hashlib.sha256(raw.tobytes()).hexdigest()

    def result_facts(self):
        """sanity on the results of the last step: every record must be a result, not an overflow"""
        gtx, ctx = self.gtx, self.ctx
        rec_head = self.d_rec.view(self.n * 2, REC_WORDS)[:, 0].clone()
        compact_records = 0
        if self.d_compact is not None:  # (the records the position-hinted pass left in the dense array: their headers are there)
            is_compact = (self.d_flags[0::2] & 2) != 0
            rec_head[0::2] = self.torch.where(is_compact, self.d_compact.view(self.n, 8)[:, 0], rec_head[0::2])
            compact_records = int(is_compact.sum().item())
        cc = gtx.download(self.buf.d_conn_count, np.uint32, 2)
        # VCF text of the region from the last step's results (host side, outside the timed region)
        t0 = time.perf_counter()
        text, calls = self.vcf_text()
        self.vcf = {"records": text.count(b"\n") - 1, "bytes": len(text), "pass": text.count(b"\tPASS\t"), "host_ms": round((time.perf_counter() - t0) * 1e3, 2)}
        # cells at the sequential saturation guard of explain_to_score (haplotype.cpp:560): a single process would replay them
        # (gtx_scores_replay); with the reads sharded over ranks the call order is spread over the ranks, so the bench only
        # makes sure there is none (cfg4: 12x per sample; the guard stands at ~8 000 reads over one site in one sample)
        hap = self.torch.as_tensor(DevView(self.buf.d_hap_u32, self.n_samples * ctx.n_hap * 4, "<i4"), device=self.device)
        at_guard = int((hap.view(-1, 4)[:, 0] >= 0xFFFF - 8).sum().item())
        if at_guard and self.n_samples > 1:
            sys.stderr.write("[bench] %d (haplotype, sample) cells reached the saturation guard: they need gtx_scores_replay (one process) or "
                             "gtx_scores_replay_log / _apply (ranks) before the calls are the reference's\n" % at_guard)
        return {"reads_aligned": int(((rec_head[0::2] & 0xFFFF) > 0).sum().item()), "vcf_text": self.vcf, "cells_at_saturation_guard": at_guard,
                "reads_overflowed": int((((rec_head >> 16) & gtx.ST_ERROR_MASK) != 0).sum().item()),
                "reads_overflowed_by_kind": {name: int((((rec_head >> 16) & bit) != 0).sum().item())
                                             for name, bit in (("labels", 1), ("paths", 2), ("walk", 4), ("record_arena_full", 8))},
                "records_in_the_arena": int((((rec_head >> 16) & gtx.ST_EXTERNAL) != 0).sum().item()), "records_in_the_dense_array": compact_records,
                "nonref_genotype_calls": int((calls["gt_second"] > 0).sum()), "score_items_refused": ctx.error_count(),
                "connections_logged": int(cc[0]), "connections_dropped": int(cc[1])}


def pinned_digest(checksum, n_reads, args):
    """the committed digest of the benchmark's own result (tests/golden/cfg2_vcf_digest.json: the oracle's VCF text over all
    reads of read set 0) against this run's: the line says itself whether the records it timed are the right ones"""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "cfg2_vcf_digest.json")
    try:
        pin = json.load(open(path))
    except (OSError, ValueError) as e:
        return {"matches_pinned": None, "pinned": "unreadable: %r" % (e,)}
    same_workload = (n_reads == pin["reads"] and args.snp_every == 1000 and args.err == 0.005 and args.nrate == 0.001 and
                     args.region_len == REGION_LEN and CFG2_READ_SEED == pin["reads_seed"])
    if not same_workload:
        return {"matches_pinned": None, "pinned": "tests/golden/cfg2_vcf_digest.json is for the default workload (this run's differs)"}
    return {"matches_pinned": checksum["vcf_sha256"] == pin["vcf_sha256"] and checksum["vcf_bytes"] == pin["vcf_bytes"],
            "pinned": "tests/golden/cfg2_vcf_digest.json"}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, ref_str, records, sample, spos):
    """the oracle (oracle/, the CPU restatement of the reference: kind "port") on the host cores: (i) one core, the literal
    semantics of a one-sample run (the reference cannot use more threads than samples, src/main.cpp:410-414); (ii) all
    cores, one pseudo-sample per thread over one shared graph + index, every thread the same number of reads as (i)
    cycled out of the sample (packing done before the clock starts; the C++ calls release the GIL)"""
    from oracle_lib import Oracle, pack_reads
    oracle = Oracle(ref_str, records, region_begin=REGION_BEGIN)
    m = len(sample)
    packed = pack_reads(list(sample))
    pos64 = np.ascontiguousarray(spos, np.int64)
    g = oracle.genotyper(1, 1)
    t0 = time.perf_counter()
    g.push(None, pos=pos64, packed=packed)
    one = time.perf_counter() - t0
    out = {"value": m / one, "unit": "reads/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(), "host_cores": os.cpu_count(),
           "sample": "first %d reads of the same workload through oracle/ (C++ restatement), 1 thread, %.1f s" % (m, one)}
    threads = args.cpu_threads or min(os.cpu_count() or 1, 256)
    if threads > 1:
        # every hardware thread its own pseudo-sample of m/8 reads (bounded: the whole leg stays around 10-20 s)
        per = max(m // 8, 1000)
        packed_t = pack_reads(list(sample[:per]))
        pos_t = pos64[:per]
        genos = [oracle.genotyper(1, 1) for _ in range(threads)]
        team = [threading.Thread(target=lambda k=k: genos[k].push(None, pos=pos_t, packed=packed_t)) for k in range(threads)]
        t0 = time.perf_counter()
        for th in team:
            th.start()
        for th in team:
            th.join()
        many = time.perf_counter() - t0
        out["all_cores"] = {"value": threads * per / many, "unit": "reads/s", "cores": threads,
                            "sample": "%d pseudo-samples of the first %d of those reads, one thread each over one shared graph + index, %.1f s" %
                                      (threads, per, many)}
    return out


def extra_pcie_fed(w, torch, gtx, steps=3, chunks=4):
    """The same workload with its inputs in (pinned) HOST memory: per step every read's bases, meta record and score item
    cross PCIe (140 B per read) in `chunks` parts on a copy stream while the previous part is aligned and scored on the
    compute stream (two staging buffers).  What an end-to-end pipeline on this box can reach when it feeds the library from
    the host; `value` of the bench line is the resident-input rate."""
    L, ctx = w.L, w.ctx
    d_seq, d_meta, d_items = w.sets[0]
    n = w.n
    c = n // chunks
    if c == 0:
        return None
    host = [t.cpu().pin_memory() for t in (d_seq, d_meta, d_items)]
    stage = [[torch.empty((c,) + tuple(t.shape[1:]), dtype=t.dtype, device=w.device) for t in (d_seq, d_meta, d_items)] for _ in range(2)]
    copy_stream = torch.cuda.Stream(device=w.device)
    ready = [torch.cuda.Event() for _ in range(2)]
    free = [torch.cuda.Event() for _ in range(2)]
    sp = w.sp
    fl = w.d_flags.data_ptr() if w.d_flags is not None else None

    def one_step():
        with torch.cuda.stream(w.stream):
            gtx.check(L.gtx_scores_zero(ctx.h, C.byref(w.buf), sp))
        for k in range(chunks):
            b = k & 1
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(free[b])
                for dst, src in zip(stage[b], host):
                    dst.copy_(src[k * c:(k + 1) * c], non_blocking=True)
                ready[b].record(copy_stream)
            with torch.cuda.stream(w.stream):
                w.stream.wait_event(ready[b])
                s_seq, s_meta, s_items = stage[b]
                gtx.check(w.align_fn(ctx.h, s_seq.data_ptr(), w.stride, s_meta.data_ptr(), c,
                                                  w.d_rec.data_ptr() + 4 * 2 * REC_WORDS * k * c, REC_WORDS, (fl + 2 * k * c) if fl else None, sp))
                gtx.check(L.gtx_score_batch_flags(ctx.h, s_items.data_ptr(), c, w.d_rec.data_ptr(), REC_WORDS, fl, C.byref(w.buf), sp))
                free[b].record(w.stream)
        with torch.cuda.stream(w.stream):
            gtx.check(L.gtx_calls_batch(ctx.h, C.byref(w.buf), w.d_phred.data_ptr(), w.d_calls.data_ptr(), sp))

    w.steps_done = 0  # (set 0 with resident inputs: what the host-fed steps have to reproduce)
    w.step()
    torch.cuda.synchronize()
    want_calls, want_phred = w.d_calls.clone(), w.d_phred.clone()
    for b in range(2):
        free[b].record(w.stream)
    one_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_read = sum(int(t.shape[1]) * t.element_size() if t.dim() > 1 else t.element_size() for t in host)
    done = chunks * c * steps
    same = bool(torch.equal(want_calls, w.d_calls)) and bool(torch.equal(want_phred, w.d_phred)) and chunks * c == n
    return {"reads_per_s": done / dt, "ms_per_step": 1000.0 * dt / steps, "host_bytes_per_read": per_read, "calls_equal_resident_run": same,
            "pcie_gbs": done * per_read / dt / 1e9, "chunks": chunks, "steps": steps,
            "note": "inputs in pinned host memory, copied per step on a second stream under the previous part's kernels"}


def extra_pipeline(args, torch, gtx, synth, device, ctx, ref, records, n=16_000_000, chunk=65536, threads=None):
    """BAM files -> VCF text, wall clock (never `value`): the reads of one sample as T position-sliced BAM files (what a
    region split of one indexed BAM gives; written before the clock starts), T host threads -- the reference's worker threads,
    src/typer/caller.cpp:399-436 -- each running gtx_reads_next (BGZF inflate on the library's team, record parse) ->
    gtx_stream_push (flag filter, duplicate reuse, plane rows) -> pinned staging -> H2D -> gtx_align_batch_planes ->
    gtx_score_batch_flags on its own stream against ONE context and ONE accumulator block; then gtx_calls_batch and
    gtx_vcf_records.  Reports reads/s over the whole leg and where the host threads' time went."""
    import tempfile
    import threading
    L = gtx.lib()
    threads = max(1, min(int(os.environ.get("GTX_BENCH_PIPE_THREADS", "16")) if threads is None else threads, (os.cpu_count() or 2) // 2))
    d_seq, d_pos = make_reads_on_device(torch, ref, records, n, seed=777, device=device, REGION_LEN=args.region_len, err_rate=args.err, n_rate=args.nrate)
    codes = unpack_nibbles(d_seq.cpu().numpy(), READ_LEN)
    pos = d_pos.cpu().numpy()
    tmp = tempfile.mkdtemp(prefix="gtx_pipeline_")
    cuts = [n * k // threads for k in range(threads + 1)]
    paths = []
    t0 = time.perf_counter()
    for k in range(threads):
        paths.append(os.path.join(tmp, "slice%02d.bam" % k))
        synth.write_fixed_bam(paths[-1], "chr20", 64444167, "SAMP0000", codes[cuts[k]:cuts[k + 1]], pos[cuts[k]:cuts[k + 1]])
    t_write = time.perf_counter() - t0
    bam_bytes = sum(os.path.getsize(q) for q in paths)
    # resident run over the same reads: what the pipeline has to reproduce
    w = Workload(torch, gtx, ctx, device, d_seq, d_pos, 1, hint=True, conn_cap=1 << 20)
    w.step()
    want_text, _ = w.vcf_text()
    w.close()
    del w, d_seq
    buf = gtx.ScoreBuffers()
    gtx.check(L.gtx_scores_alloc(ctx.h, 1, 1 << 20, C.byref(buf), None))
    stage_s = [dict(decode=0.0, push=0.0, enqueue=0.0, records=0, tasks=0) for _ in range(threads)]
    errors = []

    # what a host keeps for the life of the process, made before the clock starts: per thread a stream, pinned staging buffers and
    # their device copies, and the record slots of its file (a duplicate read's item names its predecessor's task, batches ago)
    kit = []
    for k in range(threads):
        pin = [torch.empty((chunk, 80), dtype=torch.uint8).pin_memory(), torch.empty((chunk, gtx.READ_META.itemsize), dtype=torch.uint8).pin_memory(),
               torch.empty((chunk, gtx.SCORE_ITEM.itemsize), dtype=torch.uint8).pin_memory()]
        mine = cuts[k + 1] - cuts[k]
        kit.append(dict(stream=torch.cuda.Stream(device=device), pin=pin, dev=[torch.empty_like(t, device=device) for t in pin],
                        d_rec=torch.zeros(max(mine, 1) * 2 * REC_WORDS, dtype=torch.int32, device=device),
                        d_fl=torch.zeros(max(mine, 1) * 2, dtype=torch.uint8, device=device), done=torch.cuda.Event()))

    def worker(k):
        try:
            st_ = stage_s[k]
            stream, pin, dev, d_rec, d_fl, done = (kit[k][x] for x in ("stream", "pin", "dev", "d_rec", "d_fl", "done"))
            sp = C.c_void_p(stream.cuda_stream)
            push = gtx.Stream(ctx.params, 1)
            push.set_planes(80)
            reads = gtx.Reads([paths[k]])
            first = True
            while True:
                t = time.perf_counter()
                recs, seq = reads.next(chunk)
                st_["decode"] += time.perf_counter() - t
                if len(recs) == 0:
                    break
                t = time.perf_counter()
                a_seq, a_meta, items = push.push(recs, seq)
                st_["push"] += time.perf_counter() - t
                t = time.perf_counter()
                na, ni, at = len(a_meta), len(items), st_["tasks"]  # (the stream numbers its tasks over the whole file: this batch's start at `at`)
                if not first:
                    done.synchronize()  # the staging buffers are free again
                first = False
                pin[0][:na].numpy()[...] = a_seq
                pin[1][:na].numpy()[...] = a_meta.view(np.uint8).reshape(na, -1)
                pin[2][:ni].numpy()[...] = items.view(np.uint8).reshape(ni, -1)
                with torch.cuda.stream(stream):
                    for d, h_, m in zip(dev, pin, (na, na, ni)):
                        d[:m].copy_(h_[:m], non_blocking=True)
                    gtx.check(L.gtx_align_batch_planes(ctx.h, dev[0].data_ptr(), 80, dev[1].data_ptr(), na, d_rec.data_ptr() + 4 * 2 * REC_WORDS * at, REC_WORDS,
                                                       d_fl.data_ptr() + 2 * at, sp))
                    gtx.check(L.gtx_score_batch_flags(ctx.h, dev[2].data_ptr(), ni, d_rec.data_ptr(), REC_WORDS, d_fl.data_ptr(), C.byref(buf), sp))
                    done.record(stream)
                st_["enqueue"] += time.perf_counter() - t
                st_["records"] += len(recs)
                st_["tasks"] += na
            reads.close()
            stream.synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    team = [threading.Thread(target=worker, args=(k,)) for k in range(threads)]
    for t in team:
        t.start()
    for t in team:
        t.join()
    torch.cuda.synchronize()
    t_reads = time.perf_counter() - t0
    if errors:
        L.gtx_scores_free(ctx.h, C.byref(buf))
        return {"error": errors[0]}
    def calls_and_text(b):
        d_phred = torch.zeros(max(ctx.total_tri, 1), dtype=torch.uint8, device=device)
        d_calls = torch.zeros(max(ctx.n_hap, 1) * gtx.SAMPLE_CALL.itemsize, dtype=torch.uint8, device=device)
        gtx.check(L.gtx_calls_batch(ctx.h, C.byref(b), d_phred.data_ptr(), d_calls.data_ptr(), None))
        torch.cuda.synchronize()
        nh, ta = ctx.n_hap, ctx.total_allele
        calls = d_calls.cpu().numpy().view(gtx.SAMPLE_CALL)[:nh]
        return ctx.vcf_records("chr20", ["SAMP0000"], gtx.download(b.d_gt_cov, np.uint32, ta), gtx.download(b.d_stat_u64, np.uint64, nh + 2 * ta),
                               gtx.download(b.d_stat_u32, np.uint32, nh + 6 * ta), d_phred.cpu().numpy()[:ctx.total_tri], calls)

    text = calls_and_text(buf)
    wall = time.perf_counter() - t0
    L.gtx_scores_free(ctx.h, C.byref(buf))
    # the same files through the library's own host loop (gtx_pipeline_run: the threads, staging and launches are the library's)
    del kit
    native = None
    try:
        if threads > 32:  # (beyond that every new stream's first call allocates its scratch at once: a storm of device allocations, not a loop's rate)
            raise OverflowError
        buf2 = gtx.ScoreBuffers()
        gtx.check(L.gtx_scores_alloc(ctx.h, 1, 1 << 20, C.byref(buf2), None))
        t1 = time.perf_counter()
        st = gtx.pipeline_run(ctx, paths, threads, buf2, REC_WORDS, max(cuts[k + 1] - cuts[k] for k in range(threads)), chunk)
        t_run = time.perf_counter() - t1
        text2 = calls_and_text(buf2)
        t_all = time.perf_counter() - t1
        L.gtx_scores_free(ctx.h, C.byref(buf2))
        native = {"what": "gtx_pipeline_run over the same files (host threads, pinned staging two sets deep, copies and launches inside the library), then "
                          "gtx_calls_batch + gtx_vcf_records; reads_per_s: from the moment every thread has its buffers to the VCF text (the clock of "
                          "the leg above), reads_per_s_whole_call: opening the files and allocating included",
                  "reads_per_s": n / max(st["loop_s"] + (t_all - t_run), 1e-9), "reads_per_s_whole_call": n / max(t_all, 1e-9),
                  "vcf_equals_resident_run": bool(text2 == want_text), **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()}}
    except OverflowError:
        native = {"skipped": "not run with more than 32 host threads (measured: 15-25 M reads/s at 64, DESIGN.md section 0 item 8)"}
    except Exception as e:  # noqa: BLE001
        native = {"error": repr(e)}
    for q in paths:
        os.remove(q)
    os.rmdir(tmp)
    tot = {k: sum(s_[k] for s_ in stage_s) for k in ("decode", "push", "enqueue")}
    return {"what": "BAM files -> gtx_reads_next -> gtx_stream_push (plane rows) -> pinned staging -> H2D -> align + score (one stream per host thread, "
                    "one context, one accumulator block) -> gtx_calls_batch -> gtx_vcf_records; wall clock from opening the files to the VCF text "
                    "(streams, pinned staging buffers and record slots exist before it starts)",
            "reads": int(sum(s_["records"] for s_ in stage_s)), "reads_per_s": n / wall, "wall_s": wall, "host_threads": threads,
            "bgzf_inflate_team": os.environ.get("GTX_BGZF_THREADS", "library default (up to 16)"), "bam_files": threads, "bam_bytes": bam_bytes,
            "read_loop_s": t_reads, "calls_and_vcf_text_s": wall - t_reads,
            "host_thread_seconds": {k: round(v, 3) for k, v in tot.items()},
            "slowest_thread_s": {k: round(max(s_[k] for s_ in stage_s), 3) for k in ("decode", "push", "enqueue")},
            "records_per_s_per_thread": {"decode": n / max(tot["decode"], 1e-9), "push": n / max(tot["push"], 1e-9)},
            "stage_bound_reads_per_s": n / max(max(s_["decode"] + s_["push"] + s_["enqueue"] for s_ in stage_s), 1e-9), "native_loop": native,
            "vcf_equals_resident_run": bool(text == want_text), "vcf_bytes": len(text), "bam_write_s_before_the_clock": round(t_write, 2),
            **({} if text == want_text else {"vcf_first_differences": [(a[:200], b[:200]) for a, b in zip(text.split("\n"), want_text.split("\n")) if a != b][:3],
                                              "vcf_lines_differing": sum(a != b for a, b in zip(text.split("\n"), want_text.split("\n")))})}


def extra_shrink(args, torch, gtx, synth, device, ref, records, n=2_000_000):
    """The read pre-filter in front of the ingest (gtx_bam_shrink = the reference's bamshrink; host only): one sample's BAM file
    of `n` reads over the region -> the filtered BAM, one host thread doing the filter's work, wall clock.  Once with the reference's depth cap (at this
    depth it drops most reads before they are written) and once without it (every read is trimmed, re-tagged and written)."""
    import tempfile
    d_seq, d_pos = make_reads_on_device(torch, ref, records, n, seed=4242, device=device, REGION_LEN=args.region_len, err_rate=args.err, n_rate=args.nrate)
    codes = unpack_nibbles(d_seq.cpu().numpy(), READ_LEN)
    pos = d_pos.cpu().numpy()
    tmp = tempfile.mkdtemp(prefix="gtx_shrink_")
    src, dst = os.path.join(tmp, "in.bam"), os.path.join(tmp, "out.bam")
    synth.write_fixed_bam(src, "chr20", 64444167, "SAMP0000", codes, pos)
    out = {"what": "gtx_bam_shrink of one BAM file (%d reads of %d bases over %d bp) on one host thread (the inflate team ahead of it, the output's BGZF members deflated on up to "
                   "eight threads): pair / single-read filters, trimming, tag rewrite, depth bins, deflate level 1" % (n, READ_LEN, args.region_len), "bam_bytes_in": os.path.getsize(src)}
    for name, kw in (("with_depth_cap", dict(avg_cov_by_readlen=0.3)), ("without_depth_cap", dict(no_filter_on_coverage=1))):
        t0 = time.perf_counter()
        st = gtx.bam_shrink(src, [("chr20", REGION_BEGIN, REGION_BEGIN + args.region_len - 1)], dst, gtx.shrink_params(**kw))
        dt = time.perf_counter() - t0
        out[name] = {"seconds": round(dt, 3), "records_per_s": st["records_read"] / max(dt, 1e-9), "bam_bytes_out": os.path.getsize(dst), **st}
    for q in (src, dst):
        os.remove(q)
    os.rmdir(tmp)
    return out


def extra_regions(args, torch, gtx, synth, device, ref, n_regions=20, region_len=50000, n_samples=30, depth=30):
    """What `graphtyper genotype` does per region (the reference genotypes 50 kb regions, src/main.cpp:684), on the clock from the
    variant records to the VCF text: 20 consecutive 50 kb regions, 30 samples at 30x (300 k reads per region, resident in HBM as
    plane rows before the clock starts).  Per region: gtx_graph_build -> gtx_ctx_create (index build) -> gtx_align_batch_planes
    -> gtx_score_batch_flags -> gtx_calls_batch -> download -> gtx_vcf_records.  Once one region after the other, once with
    the next region's graph + context built on a second host thread while the current region's reads run, once with two
    builder threads ahead and the VCF text of the region before on a thread of its own, and once by gtx_regions_run -- the same
    six calls per region on the library's own stage threads (the variant records handed over as gtx_record arrays)."""
    import threading
    L = gtx.lib()
    n = depth * n_samples * region_len // READ_LEN
    regions = []
    for r in range(n_regions):
        rb = REGION_BEGIN + r * region_len
        sub = np.ascontiguousarray(ref[r * region_len:(r + 1) * region_len])
        recs = synth.make_snp_records(sub, args.snp_every, seed=100 + r, region_begin=rb)
        d_seq, d_pos = make_reads_on_device(torch, sub, recs, n, seed=900 + r, device=device, REGION_LEN=region_len, region_begin=rb)
        d_planes = torch.empty((n, 80), dtype=torch.uint8, device=device)
        meta = np.zeros(n, gtx.READ_META)
        meta["l_qseq"] = READ_LEN
        meta["flag"] = gtx.FLAG_FORWARD_ONLY
        meta["pos"] = d_pos.cpu().numpy().astype(np.int32)
        items = np.zeros(n, gtx.SCORE_ITEM)
        items["first"]["align_index"] = np.arange(n, dtype=np.uint32)
        items["first"]["mapq"] = 60
        items["first"]["flag"] = gtx.FLAG_FORWARD_ONLY
        items["first"]["pos"] = meta["pos"]
        items["second"]["align_index"] = gtx.INVALID_ID
        items["sample"] = np.random.default_rng(r).integers(0, n_samples, size=n).astype(np.uint32)
        regions.append(dict(rb=rb, ref_str=synth.bases_to_str(sub), recs=recs, d_seq=d_seq, d_planes=d_planes,
                            d_meta=torch.from_numpy(meta.view(np.uint8).reshape(n, -1).copy()).to(device),
                            d_items=torch.from_numpy(items.view(np.uint8).reshape(n, -1).copy()).to(device)))
    d_rec = torch.zeros(n * 2 * REC_WORDS, dtype=torch.int32, device=device)
    d_fl = torch.zeros(n * 2, dtype=torch.uint8, device=device)
    names = ["SAMP%04d" % i for i in range(n_samples)]
    first = True

    def build(reg):
        t0 = time.perf_counter()
        g = gtx.graph_from_records(reg["ref_str"], reg["recs"], region_begin=reg["rb"])
        t1 = time.perf_counter()
        c = gtx.Context(g, device=device.index or 0)
        return c, t1 - t0, time.perf_counter() - t1

    def genotype(reg, c, t):
        nonlocal first
        if first:  # (the repack is part of staging, not of the region's clock: done for every region before the first timed run)
            for q in regions:
                gtx.check(L.gtx_reads_to_planes(c.h, q["d_seq"].data_ptr(), 80, n, q["d_planes"].data_ptr(), 80, None))
            torch.cuda.synchronize()
            first = False
        t0 = time.perf_counter()
        buf = gtx.ScoreBuffers()
        gtx.check(L.gtx_scores_alloc(c.h, n_samples, 1 << 16, C.byref(buf), None))
        d_phred = torch.empty(max(n_samples * c.total_tri, 1), dtype=torch.uint8, device=device)
        d_calls = torch.empty(max(n_samples * c.n_hap, 1) * gtx.SAMPLE_CALL.itemsize, dtype=torch.uint8, device=device)
        gtx.check(L.gtx_align_batch_planes(c.h, reg["d_planes"].data_ptr(), 80, reg["d_meta"].data_ptr(), n, d_rec.data_ptr(), REC_WORDS, d_fl.data_ptr(), None))
        gtx.check(L.gtx_score_batch_flags(c.h, reg["d_items"].data_ptr(), n, d_rec.data_ptr(), REC_WORDS, d_fl.data_ptr(), C.byref(buf), None))
        gtx.check(L.gtx_calls_batch(c.h, C.byref(buf), d_phred.data_ptr(), d_calls.data_ptr(), None))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nh, ta = c.n_hap, c.total_allele
        text = c.vcf_records("chr20", names, gtx.download(buf.d_gt_cov, np.uint32, n_samples * ta), gtx.download(buf.d_stat_u64, np.uint64, nh + 2 * ta),
                             gtx.download(buf.d_stat_u32, np.uint32, nh + 6 * ta), d_phred.cpu().numpy()[:n_samples * c.total_tri],
                             d_calls.cpu().numpy().view(gtx.SAMPLE_CALL)[:n_samples * nh])
        L.gtx_scores_free(c.h, C.byref(buf))
        c.close()
        t["gpu_step"] += t1 - t0
        t["vcf_text"] += time.perf_counter() - t1
        return text

    def run(overlap):
        t = dict(graph_build=0.0, ctx_create=0.0, gpu_step=0.0, vcf_text=0.0)
        texts = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if not overlap:
            for reg in regions:
                c, tg, tc = build(reg)
                t["graph_build"] += tg
                t["ctx_create"] += tc
                texts.append(genotype(reg, c, t))
        else:
            nxt = [None]

            def prefetch(reg):
                nxt[0] = build(reg)
            th = threading.Thread(target=prefetch, args=(regions[0],))
            th.start()
            for k, reg in enumerate(regions):
                th.join()
                c, tg, tc = nxt[0]
                t["graph_build"] += tg
                t["ctx_create"] += tc
                if k + 1 < len(regions):
                    th = threading.Thread(target=prefetch, args=(regions[k + 1],))
                    th.start()
                texts.append(genotype(reg, c, t))
        wall = time.perf_counter() - t0
        return wall, t, texts

    def run_three_stages(builders=2):
        """contexts built ahead by `builders` host threads (region k by thread k mod builders, handed over in order), the device
        step on this thread, the VCF text of the region before on a fourth thread"""
        import queue
        t = dict(graph_build=0.0, ctx_create=0.0, gpu_step=0.0, vcf_text=0.0)
        built = [queue.Queue(maxsize=2) for _ in range(builders)]
        to_text = queue.Queue(maxsize=4)
        texts = [None] * len(regions)
        errors = []

        def builder(b):
            try:
                for k in range(b, len(regions), builders):
                    built[b].put(build(regions[k]))
            except BaseException as e:  # noqa: BLE001
                errors.append(e)
                built[b].put(None)

        def texter():
            try:
                while True:
                    job = to_text.get()
                    if job is None:
                        return
                    k, c, arrays = job
                    t0 = time.perf_counter()
                    texts[k] = c.vcf_records("chr20", names, *arrays)
                    c.close()
                    t["vcf_text"] += time.perf_counter() - t0
            except BaseException as e:  # noqa: BLE001
                errors.append(e)

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        threads = [threading.Thread(target=builder, args=(b,)) for b in range(builders)] + [threading.Thread(target=texter)]
        for th in threads:
            th.start()
        for k, reg in enumerate(regions):
            got = built[k % builders].get()
            if got is None:
                break
            c, tg, tc = got
            t["graph_build"] += tg
            t["ctx_create"] += tc
            t1 = time.perf_counter()
            buf = gtx.ScoreBuffers()
            gtx.check(L.gtx_scores_alloc(c.h, n_samples, 1 << 16, C.byref(buf), None))
            d_phred = torch.empty(max(n_samples * c.total_tri, 1), dtype=torch.uint8, device=device)
            d_calls = torch.empty(max(n_samples * c.n_hap, 1) * gtx.SAMPLE_CALL.itemsize, dtype=torch.uint8, device=device)
            gtx.check(L.gtx_align_batch_planes(c.h, reg["d_planes"].data_ptr(), 80, reg["d_meta"].data_ptr(), n, d_rec.data_ptr(), REC_WORDS, d_fl.data_ptr(), None))
            gtx.check(L.gtx_score_batch_flags(c.h, reg["d_items"].data_ptr(), n, d_rec.data_ptr(), REC_WORDS, d_fl.data_ptr(), C.byref(buf), None))
            gtx.check(L.gtx_calls_batch(c.h, C.byref(buf), d_phred.data_ptr(), d_calls.data_ptr(), None))
            torch.cuda.synchronize()
            nh, ta = c.n_hap, c.total_allele
            arrays = (gtx.download(buf.d_gt_cov, np.uint32, n_samples * ta), gtx.download(buf.d_stat_u64, np.uint64, nh + 2 * ta),
                      gtx.download(buf.d_stat_u32, np.uint32, nh + 6 * ta), d_phred.cpu().numpy()[:n_samples * c.total_tri],
                      d_calls.cpu().numpy().view(gtx.SAMPLE_CALL)[:n_samples * nh])
            L.gtx_scores_free(c.h, C.byref(buf))
            t["gpu_step"] += time.perf_counter() - t1
            to_text.put((k, c, arrays))
        to_text.put(None)
        for th in threads:
            th.join()
        if errors:
            raise errors[0]
        return time.perf_counter() - t0, t, texts

    def run_in_the_library(builders, device_threads, text_threads, times=1):
        """the same six calls per region made by gtx_regions_run's own threads (no interpreter between the stages); times > 1: the
        20 regions that many times over in one call (what a chromosome's 1 200 regions are to the stages: the pipeline stays full)"""
        if first:
            run(False)
        jobs = gtx.RegionJobs([dict(reference=q["ref_str"], region_begin=q["rb"], records=q["recs"], d_planes=q["d_planes"].data_ptr(), plane_stride=80,
                                    d_meta=q["d_meta"].data_ptr(), n_reads=n, d_items=q["d_items"].data_ptr(), n_items=n) for q in regions * times])
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            texts_l, st = jobs.run(names, contig="chr20", device=device.index or 0, rec_words=REC_WORDS, builders=builders, device_threads=device_threads,
                                   text_threads=text_threads)
            wall = time.perf_counter() - t0
            if best is None or wall < best[0]:
                best = (wall, st, texts_l)
        return best

    run(False)  # (warm-up: module load, first scratch, allocator)
    wall_seq, t_seq, texts = run(False)
    wall_two, t_two, texts2 = run(True)
    wall_three, t_three, texts3 = run_three_stages()
    wall_ovl = min(wall_two, wall_three)  # (what is reported is the fastest of the overlapped forms)
    in_lib = {}
    texts4 = texts
    shapes = [(4, 2, 3), (6, 2, 4), (8, 3, 4), (10, 3, 2), (2, 1, 1)]  # (builders, device threads, text threads; the text stage is the cheap one since round 5)
    if os.environ.get("GTX_REGIONS_THREADS"):
        shapes = [tuple(int(x) for x in os.environ["GTX_REGIONS_THREADS"].split(","))]
    for shape in shapes:
        wall_l, st_l, tx = run_in_the_library(*shape)
        in_lib["%d builders, %d device threads, %d text threads" % shape] = {
            "regions_per_s": n_regions / wall_l, "wall_s": wall_l, "stage_s": {k: round(v, 4) for k, v in st_l.items() if k.endswith("_s") and k != "wall_s"},
            "same_text": bool(tx == texts)}
        if wall_l < wall_ovl:
            wall_ovl, texts4 = wall_l, tx
    best_shape = max(shapes, key=lambda sh: in_lib["%d builders, %d device threads, %d text threads" % sh]["regions_per_s"])
    wall_many, st_many, tx_many = run_in_the_library(*best_shape, times=10)
    in_lib["the 20 regions ten times over in one call, %d builders, %d device threads, %d text threads" % best_shape] = {
        "regions_per_s": 10 * n_regions / wall_many, "wall_s": wall_many, "stage_s": {k: round(v, 4) for k, v in st_many.items() if k.endswith("_s") and k != "wall_s"},
        "same_text": bool(tx_many == texts * 10)}
    return {"what": "%d consecutive %d bp regions, %d samples at %dx (%d reads per region, resident as plane rows): variant records -> gtx_graph_build -> "
                    "gtx_ctx_create -> align + score + calls -> VCF text, wall clock" % (n_regions, region_len, n_samples, depth, n),
            "regions_per_s": n_regions / wall_ovl, "reads_per_s": n_regions * n / wall_ovl, "wall_s": wall_ovl,
            "one_after_the_other": {"regions_per_s": n_regions / wall_seq, "wall_s": wall_seq, "stage_s": {k: round(v, 4) for k, v in t_seq.items()}},
            "next_region_built_on_a_second_host_thread": {"regions_per_s": n_regions / wall_two, "wall_s": wall_two, "stage_s": {k: round(v, 4) for k, v in t_two.items()}},
            "contexts_two_ahead_and_vcf_text_on_its_own_thread": {"regions_per_s": n_regions / wall_three, "wall_s": wall_three, "stage_s": {k: round(v, 4) for k, v in t_three.items()}},
            "inside_the_library_gtx_regions_run": in_lib,
            "vcf_bytes": sum(len(x) for x in texts), "same_text_both_ways": bool(texts == texts2 and texts == texts3 and texts == texts4),
            "ms_per_region": {k: round(1e3 * v / n_regions, 3) for k, v in t_seq.items()}}


def extra_cfg3(args, torch, gtx, synth, device, ref, kind="cfg3"):
    """BASELINE configs[2] in the same run: 30 samples joint over the same region, SNP+indel graph built with add_all_variants,
    reads with indels drawn on the host.  kind "cfg3": the graph SURVEY.md 8(d) specifies -- a site every 100 bp, a tenth of
    them 1-10 bp indels with a SNP within 10 bp that merges with them into a multi-allelic site; kind "clusters": the
    stress graph of rounds 1-2 -- clusters of three sites (SNP, SNP, 1-6 bp indel) every 150 bp, every read over a merged
    site of up to 8 alleles"""
    recs = synth.make_cfg3_records(ref, 100, seed=17, region_begin=REGION_BEGIN) if kind == "cfg3" else \
        synth.make_cluster_records(ref, 150, seed=8, region_begin=REGION_BEGIN)
    what = ("cfg3: 30 samples, %d reads, 1 Mb, a site every 100 bp, 10 %% of them 1-10 bp indels with a SNP within 10 bp (SURVEY 8(d)), merged by "
            "add_all_variants, max %d alleles per site" if kind == "cfg3" else
            "cfg3 stress graph: 30 samples, %d reads, 1 Mb, clusters (SNP, SNP, indel) every 150 bp merged into multi-allelic sites "
            "(add_all_variants), max %d alleles per site")
    return extra_workload(args, torch, gtx, synth, device, ref, recs, 30, True, what)


def extra_long_reads(args, torch, gtx, synth, device, ref):
    """The main workload's graph (SNP every 1 kb, one sample) with reads of 250 bases (2 x 250 libraries): rows of 128 bytes, the
    eight-k-mer build of the position-hinted pass (gtx_align_hinted_long_kernel); the express pass takes five k-mers, so what
    that pass declines is a task of the general pass."""
    recs = synth.make_snp_records(ref, 1000, seed=7, region_begin=REGION_BEGIN)
    return extra_workload(args, torch, gtx, synth, device, ref, recs, 1, False,
                          "long reads: 1 sample, %d reads of 250 bases, 1 Mb, SNP every 1 kb, max %d alleles per site", read_len=250)


def extra_repeats(args, torch, gtx, synth, device, ref):
    """The main workload's shape (one sample, SNP every 1 kb) on a reference that is NOT i.i.d.: homopolymer runs, short tandem
    repeats, satellite arrays and near-duplicate segments planted into it (synth.plant_repeats) -- what a real chromosome has.
    There the per-position proofs of the position-hinted pass fail more often, one k-mer has hundreds of places, and the
    chains of a read exceed every fixed table: the share of each pass, the tasks that reach the exact pass and
    reads_overflowed (must be 0) are the figures."""
    ref = ref.copy()
    spots = synth.plant_repeats(ref, seed=21)
    recs = synth.make_snp_records(ref, 1000, seed=7, region_begin=REGION_BEGIN)
    covered = sum(s[2] for s in spots)
    what = ("repeats: 1 sample, %%d reads, 1 Mb with %d planted repeats (%d homopolymer runs of 20-300 bp, %d di-/trinucleotide repeats, %d "
            "arrays of a 171-bp unit, %d near-duplicate 300-bp segments: %.1f %%%% of the region), SNP every 1 kb, max %%d alleles per site" %
            (len(spots), sum(s[0] == "homopolymer" for s in spots), sum(s[0] == "tandem" for s in spots), sum(s[0] == "array" for s in spots),
             sum(s[0] == "near-duplicate" for s in spots), 100.0 * covered / len(ref)))
    # (steps in flight, each on a stream of its own: the HBM-table and exact passes are chains of round trips on a few thousand
    #  wavefronts, and several steps' worth of them overlap -- 20 ms per step against 30 one at a time; 3 / 4 / 5 / 6 steps in flight =
    #  98.0 / 96.5 / 100.3 / 101.0 M reads/s here, 260 / 261 / 266 / 266 M on the genome-like leg: six since round 6's third session.  The arena of the long records is
    #  the context's: with steps in flight it is sized for all the steps of the leg (1 GB) and not started over between them -- a host's
    #  regions in flight are contexts of their own, gtx_regions_run.  GTX_BENCH_REPEATS_LANES=1: one step at a time, the arena
    #  started over per step.)
    lanes = int(os.environ.get("GTX_BENCH_REPEATS_LANES", "6"))
    return extra_workload(args, torch, gtx, synth, device, ref, recs, 1, False, what, lanes=lanes, big_record_words=(1 << 27) if lanes == 1 else (1 << 28),
                          schedule="lanes" if lanes > 1 else None)


def extra_genome_like(args, torch, gtx, synth, device):
    """The main workload's shape (one sample, SNP every 1 kb, 150-bp reads) on a reference and reads that look like a mapped human
    sample instead of SURVEY 8(d)'s friendliest case: synth.make_genome_like_reference (order-5 Markov background, 45 % interspersed
    repeat copies at 5-20 % divergence, 3 % STRs, segmental duplications) and synth.make_mapped_reads (0.5 % substitutions, 0.1 % N,
    0.05 % indel errors, 3 % soft-clipped reads, 2 % wrong or shifted hints).  The figures: who finishes a read, reads/s,
    reads_overflowed (must be 0).  The records do not depend on the hints (tests: four kinds of hints everywhere), so a wrong hint
    costs a trip through the lookup passes, never a result."""
    ref, stats = synth.make_genome_like_reference(REGION_LEN, seed=1999)
    recs = synth.make_snp_records(ref, 1000, seed=7, region_begin=REGION_BEGIN)
    n = args.extra_reads
    made = {}

    def make(seed):
        codes, hint, st = synth.make_mapped_reads(ref, recs, n, seed=seed, region_begin=REGION_BEGIN)
        made[seed] = st
        return codes, hint
    what = ("genome-like: 1 sample, %%d reads, 1 Mb of an order-5 Markov background with %.0f %%%% interspersed repeat copies (%d families, 5-20 %%%% "
            "diverged), %.1f %%%% STRs, %.1f %%%% segmental duplications; SNP every 1 kb, max %%d alleles per site; reads with 0.5 %%%% substitutions, 0.1 %%%% N, "
            "0.05 %%%% indel errors, 3 %%%% soft-clipped, 2 %%%% wrong / shifted position hints" %
            (100.0 * stats["interspersed"], stats["families"], 100.0 * stats["str"], 100.0 * stats["segdup"]))
    lanes = int(os.environ.get("GTX_BENCH_REPEATS_LANES", "6"))  # (as in extra_repeats: steps in flight, each on a stream of its own)
    out = extra_workload(args, torch, gtx, synth, device, ref, recs, 1, False, what, lanes=lanes, big_record_words=(1 << 27) if lanes == 1 else (1 << 28),
                         make_reads=make, schedule="lanes" if lanes > 1 else None)
    out["reads_made"] = made.get(5)
    out["reference"] = stats
    if not args.no_cpu_baseline:  # the oracle on a sample of THESE reads, one host core: what the device rate of this leg stands beside
        try:
            from oracle_lib import Oracle, pack_reads
            codes, hint, _ = synth.make_mapped_reads(ref, recs, 20000, seed=5, region_begin=REGION_BEGIN)
            order = np.argsort(hint, kind="stable")
            g = Oracle(synth.bases_to_str(ref), recs, region_begin=REGION_BEGIN).genotyper(1, 1)
            packed = pack_reads(list(codes[order]))
            t0 = time.perf_counter()
            g.push(None, pos=np.ascontiguousarray(hint[order], np.int64), packed=packed)
            one = time.perf_counter() - t0
            out["cpu_oracle"] = {"reads_per_s": 20000 / one, "cores": 1, "sample": "20 000 reads of this leg's kind through oracle/, 1 thread, %.1f s" % one}
        except Exception as e:
            out["cpu_oracle"] = {"error": repr(e)}
    return out


def extra_cfg5(args, torch, gtx, synth, device, n_pairs_per_sv=160, background_pairs=12000, n_samples=100, steps=10, tile=8):
    """BASELINE configs[4] as far as one GPU goes: `genotype_sv`, 100 samples over a 1 Mb SV-augmented graph -- 100 <DEL> of
    50-5 000 bp and 50 <INS> with 152-bp breakpoint alleles, from FASTA + VCF through gtx_graph_from_files -- FR pairs over every
    breakpoint plus background, the SV stream logic on the host (record filter, coverage filter, leftovers) before the clock,
    then per step align + score (with the reference-depth track) + calls on the device; afterwards the SV post-processing of
    the calls and the VCF text on the host (gtx_vcf_records: BREAKPOINT / COVERAGE / AGGREGATED records).  The read set is the
    one of tests/test_gpu_configs.py::test_cfg5_sv_graph (72 k records: its generator is a Python loop per pair); the timed steps
    run `tile` copies of it side by side in one batch (the same reads, `tile` times the depth), the VCF comes from one copy."""
    import tempfile
    L = gtx.lib()
    t0 = time.time()
    seqs, lines, codes, rec = synth.make_sv_case(n_ref=REGION_LEN, n_del=100, n_ins=50, n_samples=n_samples, pairs_per_sv=n_pairs_per_sv,
                                                 background_pairs=background_pairs)
    t_make = time.time() - t0
    tmp = tempfile.mkdtemp(prefix="gtx_cfg5_")
    fa, vcf = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "in.vcf")
    with open(fa, "w") as f:
        for name, sq in seqs.items():
            f.write(">%s\n" % name)
            f.write("\n".join(sq[i:i + 60] for i in range(0, len(sq), 60)) + "\n")
    with open(vcf, "w") as f:
        f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n" + "\n".join(lines) + "\n")
    t0 = time.time()
    graph, _, sv_table = gtx.graph_from_files(fa, vcf, "chrS", is_sv_graph=True, with_sv_table=True)
    t_graph = time.time() - t0
    os.remove(fa)
    os.remove(vcf)
    for extra in (fa + ".fai",):
        if os.path.exists(extra):
            os.remove(extra)
    os.rmdir(tmp)
    t0 = time.time()
    ctx = gtx.Context(graph, device=0, is_sv_graph=True, big_record_words=1 << 26)  # (the arena, 256 MB: room for what the steps that are in flight together put there; with 2 GB the same step takes 2.85 ms instead of 2.15)
    t_ctx = time.time() - t0
    st = gtx.Stream(ctx.params, 1)
    st.set_coverage([0.5] * n_samples)
    t0 = time.time()
    a_seq, a_meta, items = st.push(rec, gtx.pack_nibbles(codes))
    left = st.finish()
    t_stream = time.time() - t0
    items = np.concatenate([items, left])
    n_align, n_items = len(a_meta), len(items)
    tiled = []
    for k in range(tile):
        it = items.copy()
        for side in ("first", "second"):
            ai = it[side]["align_index"]
            it[side]["align_index"] = np.where(ai != gtx.INVALID_ID, ai + np.uint32(k * n_align), ai)
        tiled.append(it)
    tiled = np.concatenate(tiled)
    d_seq = torch.from_numpy(np.ascontiguousarray(a_seq)).to(device).repeat(tile, 1)
    d_meta = torch.from_numpy(a_meta.view(np.uint8).reshape(n_align, -1).copy()).to(device).repeat(tile, 1)
    d_items = torch.from_numpy(tiled.view(np.uint8).reshape(len(tiled), -1).copy()).to(device)
    d_rec = torch.zeros(tile * n_align * 2 * REC_WORDS, dtype=torch.int32, device=device)
    buf = gtx.ScoreBuffers()
    reduced = C.c_uint64()
    gtx.check(L.gtx_scores_alloc(ctx.h, n_samples, 1 << 22, C.byref(buf), C.byref(reduced)))
    d_phred = torch.zeros(max(n_samples * ctx.total_tri, 1), dtype=torch.uint8, device=device)
    d_calls = torch.zeros(max(n_samples * ctx.n_hap, 1) * gtx.SAMPLE_CALL.itemsize, dtype=torch.uint8, device=device)
    stream = torch.cuda.Stream(device=device)
    sp = C.c_void_p(stream.cuda_stream)

    def step(copies):
        with torch.cuda.stream(stream):
            gtx.check(L.gtx_ctx_big_records_rewind(ctx.h, sp))
            gtx.check(L.gtx_scores_zero(ctx.h, C.byref(buf), sp))
            gtx.check(L.gtx_align_batch(ctx.h, d_seq.data_ptr(), int(d_seq.shape[1]), d_meta.data_ptr(), copies * n_align, d_rec.data_ptr(), REC_WORDS, sp))
            gtx.check(L.gtx_score_batch(ctx.h, d_items.data_ptr(), copies * n_items, d_rec.data_ptr(), REC_WORDS, C.byref(buf), sp))
            gtx.check(L.gtx_calls_batch(ctx.h, C.byref(buf), d_phred.data_ptr(), d_calls.data_ptr(), sp))

    for _ in range(2):
        step(tile)
    torch.cuda.synchronize()
    ctx.pass_times()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(tile)
    torch.cuda.synchronize()
    dt_one = time.perf_counter() - t0
    kt = ctx.kernel_times()
    calls_one = d_calls.cpu().numpy().copy()
    # Four steps in flight, each with records, accumulator block and stream of its own (round 5: the SV graph's reads spend their
    # time in the general pass and behind it, chains of round trips that leave most of the chip idle).  The arena is the context's:
    # it is started over once, in front of the steps, and holds what all of them put there.
    n_lanes = int(os.environ.get("GTX_BENCH_CFG5_LANES", "4"))  # (3: 1.55-1.59 ms per step, 4: 1.43, 6: 1.56)
    lanes = [dict(stream=stream, sp=sp, d_rec=d_rec, buf=buf, d_phred=d_phred, d_calls=d_calls)]
    for _ in range(1, n_lanes):
        b2 = gtx.ScoreBuffers()
        gtx.check(L.gtx_scores_alloc(ctx.h, n_samples, 1 << 22, C.byref(b2), None))
        s2 = torch.cuda.Stream(device=device)
        lanes.append(dict(stream=s2, sp=C.c_void_p(s2.cuda_stream), d_rec=torch.zeros_like(d_rec), buf=b2, d_phred=torch.zeros_like(d_phred),
                          d_calls=torch.zeros_like(d_calls)))

    def lane_step(ln):
        with torch.cuda.stream(ln["stream"]):
            gtx.check(L.gtx_scores_zero(ctx.h, C.byref(ln["buf"]), ln["sp"]))
            gtx.check(L.gtx_align_batch(ctx.h, d_seq.data_ptr(), int(d_seq.shape[1]), d_meta.data_ptr(), tile * n_align, ln["d_rec"].data_ptr(), REC_WORDS, ln["sp"]))
            gtx.check(L.gtx_score_batch(ctx.h, d_items.data_ptr(), tile * n_items, ln["d_rec"].data_ptr(), REC_WORDS, C.byref(ln["buf"]), ln["sp"]))
            gtx.check(L.gtx_calls_batch(ctx.h, C.byref(ln["buf"]), ln["d_phred"].data_ptr(), ln["d_calls"].data_ptr(), ln["sp"]))

    def flight(k):
        torch.cuda.synchronize()
        gtx.check(L.gtx_ctx_big_records_rewind(ctx.h, None))
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(k):
            lane_step(lanes[i % n_lanes])
        torch.cuda.synchronize()
        return time.perf_counter() - t

    flight(2 * n_lanes)
    dt_flight = min(flight(steps) for _ in range(2))
    failed = C.c_uint64()
    same_calls = True
    for ln in lanes:
        gtx.check(L.gtx_records_failed(ctx.h, ln["d_rec"].data_ptr(), REC_WORDS, tile * n_align, None, C.byref(failed)))
        same_calls = same_calls and failed.value == 0 and bool(np.array_equal(ln["d_calls"].cpu().numpy(), calls_one))
    for ln in lanes[1:]:
        L.gtx_scores_free(ctx.h, C.byref(ln["buf"]))
    in_flight = same_calls and dt_flight < dt_one  # (what is reported: the faster way, when it leaves the same calls)
    dt = dt_flight if in_flight else dt_one
    step(1)  # (one copy of the reads: what the VCF is written from)
    torch.cuda.synchronize()
    # the calls' post-processing and the VCF text (host)
    t0 = time.perf_counter()
    nh, ta = ctx.n_hap, ctx.total_allele
    depth = gtx.download(buf.d_ref_depth, np.uint32, n_samples * (ctx.ref_depth_len + 1))
    sat = C.c_uint64()
    gtx.check(L.gtx_ref_depth_finalize(depth.ctypes.data_as(C.c_void_p), n_samples, ctx.ref_depth_len, C.byref(sat)))
    calls = d_calls.cpu().numpy().view(gtx.SAMPLE_CALL)[:n_samples * nh]
    text = ctx.vcf_records("chrS", ["SAMP%03d" % i for i in range(n_samples)], gtx.download(buf.d_gt_cov, np.uint32, n_samples * ta),
                           gtx.download(buf.d_stat_u64, np.uint64, nh + 2 * ta), gtx.download(buf.d_stat_u32, np.uint32, nh + 6 * ta),
                           d_phred.cpu().numpy()[:n_samples * ctx.total_tri], calls, sv_table=sv_table, ref_depth=depth)
    t_vcf = time.perf_counter() - t0
    rec_head = d_rec.view(tile * n_align * 2, REC_WORDS)[:n_align * 2, 0]
    out = {"workload": "cfg5 on one GPU: genotype_sv, %d samples, %d x %d records per step (%d aligned reads, %d score items per copy), 1 Mb SV graph from files "
                       "(100 <DEL> 50-5000 bp, 50 <INS> with breakpoint alleles: %d sites, %d SV table entries)" %
                       (n_samples, tile, len(rec), n_align, n_items, nh, sv_table.count("\n")),
           "reads_per_s": tile * len(rec) * steps / dt, "ms_per_step": 1000.0 * dt / steps, "steps": steps,
           "schedule": ("%d steps in flight, each on a stream of its own" % n_lanes) if in_flight else "one step at a time",
           "one_at_a_time_ms_per_step": 1000.0 * dt_one / steps, "in_flight_ms_per_step": 1000.0 * dt_flight / steps,
           "steps_in_flight_leave_the_same_calls": same_calls,
           "align_kernels": {k[0]: {"ms": k[1], "tasks_completed": k[2]} for k in kt},
           "host_before_the_clock_s": {"make_reads": round(t_make, 2), "graph_from_files": round(t_graph, 3), "ctx_create": round(t_ctx, 3), "stream_logic": round(t_stream, 3)},
           "vcf": {"host_ms": round(1000.0 * t_vcf, 1), "bytes": len(text), "records": text.count(b"\n") - 1,
                   "breakpoint": text.count(b":BREAKPOINT"), "coverage": text.count(b":COVERAGE>"), "aggregated": text.count(b":AGGREGATED>"),
                   "sha256": __import__("hashlib").sha256(text).hexdigest()},
           "reads_overflowed": int((((rec_head >> 16) & gtx.ST_ERROR_MASK) != 0).sum().item()), "score_items_refused": ctx.error_count(),
           "ref_depth_positions_saturated": int(sat.value)}
    L.gtx_scores_free(ctx.h, C.byref(buf))
    ctx.close()
    return out


def extra_workload(args, torch, gtx, synth, device, ref, recs, n_samples, add_all, what, lanes=None, big_record_words=0, read_len=READ_LEN, make_reads=None,
                   schedule=None):
    """make_reads (optional): seed -> (codes, pos) instead of synth.make_reads' clean reads; schedule "lanes": whole steps in flight, each
    on a stream of its own (for inputs whose time is in the passes behind the position-hinted one: chains of round trips that leave
    most of the chip idle -- three steps' worth of them overlap)"""
    n = args.extra_reads
    if make_reads is None:
        make_reads = lambda seed: synth.make_reads(ref, recs, n, read_len=read_len, seed=seed, region_begin=REGION_BEGIN)
    lanes = args.lanes if lanes is None else lanes
    t0 = time.time()
    graph = gtx.graph_from_records(synth.bases_to_str(ref), recs, region_begin=REGION_BEGIN, add_all_variants=add_all)
    t_graph = time.time() - t0
    t0 = time.time()
    ctx = gtx.Context(graph, device=0, big_record_words=big_record_words)
    t_ctx = time.time() - t0
    codes, pos = make_reads(5)
    order = np.argsort(pos, kind="stable")
    codes, pos = codes[order], pos[order]
    d_seq = torch.from_numpy(gtx.pack_nibbles(codes)).to(device)
    samples = np.random.default_rng(3).integers(0, n_samples, size=n).astype(np.uint32)
    w = Workload(torch, gtx, ctx, device, d_seq, torch.from_numpy(pos), n_samples, samples=samples if n_samples > 1 else None,
                 hint=not args.no_hint, lanes=lanes, read_len=read_len)
    w.rewind = lanes == 1 and big_record_words != 0
    w.allow_whole_steps = True
    w.staggered = (schedule or args.schedule) == "staggered" and len(w.lanes) >= 2  # (the same schedule as the main workload, chosen the same way)
    if args.read_sets > 1:  # a second set of reads: the steps alternate
        codes2, pos2 = make_reads(6)
        order2 = np.argsort(pos2, kind="stable")
        w.add_reads(torch.from_numpy(gtx.pack_nibbles(codes2[order2])).to(device), torch.from_numpy(pos2[order2]))
    steps = 9 if len(w.lanes) > 1 else 4
    dt, _ = w.run(steps, 2, None)
    ms, handed = ctx.pass_times()
    facts = w.result_facts()
    prof = ctx.profile()
    if prof[15] > 0:  # GTX_LIB=libgtx_prof.so
        names = ["load read", "keys + exact probes", "exact labels", "chain exact", "hamming lookup", "chain hamming",
                 "walk starts", "walk ends", "filters", "record"]
        tot = float(prof[:10].sum())
        sys.stderr.write("cfg3-like: phase cycles per task of the general pass (profiling build), %d tasks:\n" % prof[15])
        for k, nm in enumerate(names):
            sys.stderr.write("  %-22s %10.0f  %5.1f%%\n" % (nm, prof[k] / float(prof[15]), 100.0 * prof[k] / tot))
        sys.stderr.write("  longest task %d cycles (task %d); tasks over 100 k / 200 k / 400 k cycles: %d / %d / %d (all launches of the run)\n" %
                         (int(prof[10]) >> 32, int(prof[10]) & 0xFFFFFFFF, prof[11], prof[12], prof[13]))
        sys.stderr.write("  handed to the HBM-table pass for: labels %d, paths %d, walk %d, record %d (a task may name several)\n" %
                         (int(prof[29]) & 0xFFFFFFFF, int(prof[29]) >> 32, int(prof[30]) & 0xFFFFFFFF, int(prof[30]) >> 32))
    if prof[28] > 0:
        sys.stderr.write("cfg3-like: scoring kernel, cycles per visit of a workgroup's first lane (profiling build), %d visits: item fetch %.0f, scoring %.0f, flush %.0f\n" %
                         (prof[28], prof[25] / float(prof[28]), prof[26] / float(prof[28]), prof[27] / float(prof[28])))
    if prof[31] > 0:
        names = ["unpack reads", "keys", "index lookups", "half-key entries", "seeding verdict", "run + walk geometry", "compares",
                 "indel-tail compares", "verdict + record"]
        tot = float(prof[16:25].sum())
        sys.stderr.write("cfg3-like: phase cycles per group of four reads of the express pass (profiling build), %d groups:\n" % prof[31])
        for k, nm in enumerate(names):
            sys.stderr.write("  %-22s %10.0f  %5.1f%%\n" % (nm, prof[16 + k] / float(prof[31]), 100.0 * prof[16 + k] / tot))
    exact = ctx.exact_pass_tasks()
    one_at_a_time_ms = None
    if schedule == "lanes" and w.used_lanes > 1:  # (beside it: the same steps one at a time, the arena started over first)
        torch.cuda.synchronize()
        gtx.check(gtx.lib().gtx_ctx_big_records_rewind(ctx.h, None))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            w.step(0)
        torch.cuda.synchronize()
        one_at_a_time_ms = 1000.0 * (time.perf_counter() - t0) / 3
    w.close()
    what = what % (n, int(ctx.hap_cnum.max()))
    kt = ctx.kernel_times()
    out = {"workload": what, "align_kernels": {k[0]: {"ms": k[1], "tasks_completed": k[2]} for k in kt},
           "reads_per_s": n * steps / dt, "ms_per_step": 1000.0 * dt / steps, "steps": steps,
           "schedule": ("staggered, %d steps in flight" % w.used_lanes) if (w.staggered and w.used_lanes > 1) else
                       ("%d steps in flight, each on a stream of its own" % w.used_lanes) if w.used_lanes > 1 else "one step at a time",
           "one_at_a_time_ms_per_step": one_at_a_time_ms,
           "calibration": w.calibration, "resident_read_sets": len(w.sets), "sites": int(ctx.n_hap), "ctx_create_s": round(t_ctx, 3),
           "graph_build_s": round(t_graph, 3),
           "align_passes_ms": {"express": ms[0], "general": ms[1], "hbm_tables": ms[2]},
           "pass_shares": {"tasks": n, "handed_to_general": handed, "share_general": handed / float(n),
                           "position_hinted_share": kt[0][2] / float(n)},
           "exact_pass": {"tasks_with_a_small_part_of_the_slab": exact[0], "tasks_with_a_large_part": exact[1], "tasks_with_the_whole_slab": exact[2],
                          "tasks_refused": exact[3]}}
    out.update(facts)
    ctx.close()
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, argv)
    if args.dry_run:
        return dry_run(args)

    import torch
    from graphtyper_amd import lib as gtx
    from graphtyper_amd import synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libgtx has no CPU path")
    if not os.path.exists(gtx.LIB_PATH):
        raise SystemExit("libgtx.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    rank, local_rank, world, dist = init_ranks(args)
    if os.environ.get("GTX_BENCH_SHARE_DEVICE"):  # test switch: every rank on device 0 (the N > 1 control flow on a one-GPU box, --backend gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    # ---- graph + index (replicated on every GPU) ----
    ref, records, ref_str = cfg2_graph_inputs(synth, args.region_len, args.snp_every)
    t0 = time.time()
    ctx = gtx.Context(gtx.graph_from_records(ref_str, records, region_begin=REGION_BEGIN), device=local_rank)
    t_ctx = time.time() - t0
    t0 = time.time()
    ctx2 = gtx.Context(gtx.graph_from_records(ref_str, records, region_begin=REGION_BEGIN), device=local_rank)
    t_ctx_warm = time.time() - t0  # (the first context of a process also pays the HIP module load)
    ctx2.close()
    n_keys, n_labels = ctx.index_stats()

    # ---- reads, resident in HBM before the timed region ----
    # (one GPU: cfg2, 1 sample.  Several: cfg4 -- the reads of 1000 samples, sharded by read over the ranks, every rank holds
    #  the accumulators of all samples and the ranks' blocks are summed once per step)
    n_samples, per_rank = job_shape(args, world)
    n = per_rank[rank]

    def sample_ids(k, r=rank, count=None):
        return None if n_samples == 1 else np.random.default_rng(4242 + 7919 * k + r).integers(0, n_samples, size=n if count is None else count).astype(np.uint32)

    d_seq, d_pos = make_reads_on_device(torch, ref, records, n, seed=CFG2_READ_SEED + rank, device=device, REGION_LEN=args.region_len,
                                        err_rate=args.err, n_rate=args.nrate)
    w = Workload(torch, gtx, ctx, device, d_seq, d_pos, n_samples, samples=sample_ids(0), hint=not args.no_hint, lanes=args.lanes)
    for k in range(1, max(args.read_sets, 1)):  # the steps alternate between resident read sets (different reads, same size)
        w.samples = sample_ids(k)
        w.add_reads(*make_reads_on_device(torch, ref, records, n, seed=CFG2_READ_SEED + rank + 7919 * k, device=device, REGION_LEN=args.region_len,
                                          err_rate=args.err, n_rate=args.nrate))
    if dist is not None:
        w.setup_reduce(dist, rank, world, local_rank)
    w.staggered = args.schedule == "staggered" and len(w.lanes) >= 2
    dt, align_ms = w.run(args.steps, args.warmup, dist)
    pass_ms, n_pass2 = ctx.pass_times()  # last step: express / general / HBM-table kernels
    kern = ctx.kernel_times() if hasattr(ctx, "kernel_times") else None
    step_alone_ms = None
    if len(w.lanes) > 1 and dist is None:  # (N = 1 only: with an exchange step every rank would have to take part)
        # (behind the timed region and the queries above: one step at a time on one stream -- a step's latency, where the
        #  timed region measures the throughput of steps in flight on several streams)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            w.step(0)
        torch.cuda.synchronize()
        step_alone_ms = 1000.0 * (time.perf_counter() - t0) / 4
    facts = w.result_facts()
    n_gpus = dist.get_world_size() if dist is not None else 1
    per_rank_ms = gather_floats(dist, 1000.0 * w.local_s / args.steps, device)
    reduce_ms = gather_floats(dist, w.reduce_ms, device)

    # ---- N > 1: the exchange checks itself.  One more step over read set 0 on every rank leaves the SUM of the ranks' blocks in
    # every rank's block; rank 0 then makes the reads of every rank again (their seeds are known), runs them one after the other
    # into ONE block without an exchange, and the two blocks must be the same bytes: the first run on real xGMI says itself
    # whether gtx_scores_reduce summed what one GPU computes.
    reduce_check = None
    if dist is not None and not args.no_reduce_check:
        w.steps_done = 0
        w.step(0)
        summed = w.block_digest(0)
        # ... and the same read set through the schedule the timed steps ran: the exchange there is on a stream of its own
        # (Workload._score), which the step above does not touch
        summed_staggered = None
        if w.staggered and len(w.lanes) >= 2:
            w.steps_done = 0
            w.steps_staggered(1)
            torch.cuda.synchronize()
            w.steps_done = 0
            summed_staggered = w.block_digest(0)
        # (the one order-dependent step of the scoring, across ranks: nothing to do at cfg4's 12x per sample, but it is the path
        #  a deeper job takes -- tests/test_saturation.py, tests/test_dist_gloo.py hold it to the oracle)
        replayed = None
        try:
            hap = torch.as_tensor(DevView(w.buf.d_hap_u32, n_samples * ctx.n_hap * 4, "<i4"), device=device)
            if int((hap.view(-1, 4)[:, 0] >= 0xFFFF - 8).sum().item()):
                cells, entries = w.replay_across_ranks(dist, int(sum(per_rank[:rank])))
                replayed = {"cells": cells, "log_entries_of_all_ranks": entries}
        except Exception as e:  # (a diagnostic leg: it must not take the line with it)
            replayed = {"error": str(e)}
        if rank == 0:
            import hashlib
            t0 = time.perf_counter()
            n64 = ctx.n_hap + 2 * ctx.total_allele
            sum64, sum32 = None, None
            for r in range(world):
                d_seq_r, d_pos_r = make_reads_on_device(torch, ref, records, per_rank[r], seed=CFG2_READ_SEED + r, device=device, REGION_LEN=args.region_len,
                                                        err_rate=args.err, n_rate=args.nrate)
                wc = Workload(torch, gtx, ctx, device, d_seq_r, d_pos_r, n_samples, samples=sample_ids(0, r, per_rank[r]), hint=not args.no_hint, lanes=1)
                wc.step(0)
                torch.cuda.synchronize()
                raw = gtx.download(wc.buf.d_stat_u64, np.uint8, wc.reduced_bytes)
                p64, p32 = raw[:8 * n64].view(np.uint64), raw[8 * n64:].view(np.uint32)
                sum64 = p64.copy() if sum64 is None else sum64 + p64  # (wrap-around like the device's adds)
                sum32 = p32.copy() if sum32 is None else sum32 + p32
                wc.close()
                del wc, d_seq_r, d_pos_r
            alone = hashlib.sha256(sum64.tobytes() + sum32.tobytes()).hexdigest()
            reduce_check = {"what": "block of read set 0 summed over the ranks (gtx_scores_reduce) against the same %d read sets run one after the other "
                                    "on rank 0's GPU and added on the host, no exchange" % world,
                            "summed_sha256": summed, "one_gpu_sha256": alone, "equal": summed == alone and summed_staggered in (None, alone),
                            "summed_in_the_timed_schedule_sha256": summed_staggered, "saturation_guard_replay_across_ranks": replayed, "seconds": round(time.perf_counter() - t0, 2)}
            if summed != alone or summed_staggered not in (None, alone):
                sys.stderr.write("[bench] THE SUMMED BLOCK DIFFERS from the one-GPU block of the same reads\n")
        dist.barrier()

    prof = ctx.profile()
    if rank == 0 and prof[15] > 0:  # GTX_LIB=libgtx_prof.so: shader cycles per phase of the general algorithm, per task that ran it
        names = ["load read", "keys + exact probes", "exact labels", "chain exact", "hamming lookup", "chain hamming",
                 "walk starts", "walk ends", "filters", "record"]
        tot = float(prof[:10].sum())
        sys.stderr.write("phase cycles per task of the general pass (profiling build), %d tasks:\n" % prof[15])
        for k, nm in enumerate(names):
            sys.stderr.write("  %-22s %10.0f  %5.1f%%\n" % (nm, prof[k] / float(prof[15]), 100.0 * prof[k] / tot))
        sys.stderr.write("  longest task %d cycles (task %d = read %d); tasks over 100 k / 200 k / 400 k cycles: %d / %d / %d (all launches of the run)\n" %
                         (int(prof[10]) >> 32, int(prof[10]) & 0xFFFFFFFF, (int(prof[10]) & 0xFFFFFFFF) >> 1, prof[11], prof[12], prof[13]))
    if rank == 0 and prof[28] > 0:
        sys.stderr.write("scoring kernel, cycles per visit of a workgroup's first lane (profiling build), %d visits: item fetch %.0f, scoring %.0f, flush %.0f\n" %
                         (prof[28], prof[25] / float(prof[28]), prof[26] / float(prof[28]), prof[27] / float(prof[28])))
    if rank == 0 and prof[31] > 0:
        names = ["unpack reads", "keys", "index lookups", "half-key entries", "seeding verdict", "run + walk geometry", "compares",
                 "indel-tail compares", "verdict + record"]
        tot = float(prof[16:25].sum())
        sys.stderr.write("phase cycles per group of four reads of the express pass (profiling build), %d groups:\n" % prof[31])
        for k, nm in enumerate(names):
            sys.stderr.write("  %-22s %10.0f  %5.1f%%\n" % (nm, prof[16 + k] / float(prof[31]), 100.0 * prof[16 + k] / tot))
    if rank != 0:
        w.close()
        if dist is not None:
            dist.destroy_process_group()
        return 0

    ms_per_step = 1000.0 * dt / args.steps
    value = sum(per_rank) * args.steps / dt
    align_avg_ms = float(np.mean(align_ms)) if len(align_ms) else 0.0
    # dominant kernel of the step and the units it completes (what it hands on is not counted for it)
    roof = dominant_kernel(pass_ms, n_pass2, n, align_avg_ms, kern)
    traffic = None
    import glob
    tfs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))  # the latest round's PMC passes
    tf = tfs[-1] if tfs else ""
    if tf:
        try:
            tj = json.load(open(tf))
            tk = tj.get("kernels", {}).get(roof["kernel"])
            if tk and tk.get("hbm_bytes_per_launch") is not None and tj.get("reads_per_launch"):
                traffic = tk["hbm_bytes_per_launch"] * float(n) / float(tj["reads_per_launch"])  # per launch of THIS run
        except Exception:
            traffic = None
    roof["traffic"] = traffic
    # (bench.py cannot run rocprofv3 on itself: the figure is the PMC measurement of tools/profile.sh on this workload, scaled
    #  to this run's read count -- not a measurement of this run)
    roof["traffic_source"] = ("profiles/" + os.path.basename(tf) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh, FETCH doubled for gfx950; "
                              "scaled by reads)") if traffic is not None else None
    if n_gpus == 1 and n_samples == 1:
        what = "cfg2: 1 sample, %d synthetic %d bp reads per GPU" % (n, READ_LEN)
    else:
        what = ("cfg4: %d samples, %d synthetic %d bp reads in all, sharded by read over %d GPU(s) (%s; %s scaling), every rank holds the "
                "accumulators of all samples, one packed sum over the ranks per step" %
                (n_samples, sum(per_rank), READ_LEN, n_gpus, "%d per GPU" % n if len(set(per_rank)) == 1 else "%s per GPU" % per_rank, args.scaling))
    cfg = {"workload": what + ", chr20:1000001-2000000 (1 Mb), SNP-only graph (1 SNP / %d bp), unpaired, 0.5%% substitutions, 0.1%% N; "
                       "result = SampleCall (GT, PL, GQ, depths) per site and sample; gtx_vcf_records writes the region's VCF records from them on "
                       "the host after the timed steps (config.vcf_text, config.calls_checksum: the digest the full-size GPU test reproduces "
                       "from the oracle over all reads)" % args.snp_every,
           "samples": n_samples, "reads_per_rank": per_rank,
           "reads_per_gpu": n, "index_keys": n_keys, "index_labels": n_labels, "haplotypes": ctx.n_hap,
           "ctx_create_s": round(t_ctx_warm, 3), "ctx_create_first_s": round(t_ctx, 3),
           "position_hint": not args.no_hint, "task_flags_side_array": USE_TASK_FLAGS, "dense_records": USE_COMPACT,
           "read_layout": "bit planes (gtx_align_batch_planes; repacked once from BAM nibbles by gtx_reads_to_planes before the timed region)" if PLANE_INPUT else "BAM nibbles (gtx_align_batch_flags repacks them inside every call)", "resident_read_sets": len(w.sets),
           "streams": {"steps_in_flight": w.used_lanes, "schedule": ("staggered" if w.staggered else "lanes") if w.used_lanes > 1 else "one step at a time",
                       "calibration": w.calibration,
                       "step_alone_ms": step_alone_ms,
                       "note": "every step is align -> score -> calls on its own records and accumulators. staggered: stream H carries, one "
                               "kernel after the other, the position-hinted pass of step k and the scoring of step k-2; stream T the short "
                               "express / general queues behind every position-hinted pass (gtx_align_batch_planes_staged: 0.35 % of the "
                               "reads, latency-bound, most CUs idle), which drain beside H's kernels. ms_per_step = timed wall / steps (the "
                               "last steps' scoring is inside the timed region); step_alone_ms = one step at a time on one stream"},
           "parallelism": "reads sharded over %d GPU(s), graph+index replicated" % n_gpus,
           "reduce": w.reduce_kind, "reduced_bytes_per_step": w.reduced_bytes if n_gpus > 1 else 0,
           "reduce_ms": max(reduce_ms) if n_gpus > 1 else 0.0, "reduce_ms_per_rank": reduce_ms, "per_rank_ms_per_step": per_rank_ms,
           "reduce_check": reduce_check}
    cfg.update(facts)
    if n_gpus == 1 and n_samples == 1:
        try:
            cfg["calls_checksum"] = w.calls_checksum(0)
            cfg["calls_checksum"]["reads_seed"] = CFG2_READ_SEED
            cfg["calls_checksum"].update(pinned_digest(cfg["calls_checksum"], n, args))
            cfg["calls_checksum"]["checked_by"] = ("tests/test_gpu_full_size.py::test_cfg2_every_read_against_the_oracle: all reads of this set through "
                                                   "oracle/ on the host cores -> accumulators, SampleCalls and VCF bytes equal")
        except Exception as e:  # the extra field must never cost the main line
            cfg["calls_checksum"] = {"error": repr(e)}
    out = {"metric": METRIC, "value": value, "unit": "reads/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling if n_gpus > 1 else "weak", "vs_baseline": None,
           "dtype": "u64/u32 integer (2-bit k-mer keys, byte compares, u32 atomics)", "data": "synthetic", "config": cfg,
           "roofline": roof}
    if n_gpus == 1 and not args.no_cpu_baseline:
        m = min(args.cpu_sample, n)
        out["cpu_baseline"] = cpu_baseline(args, ref_str, records, unpack_nibbles(d_seq[:m].cpu().numpy(), READ_LEN), d_pos[:m].cpu().numpy())
    else:
        out["cpu_baseline"] = None
    if n_gpus == 1 and n_samples == 1 and not args.no_extra:
        try:  # the same schedule over 500 steps: a timed window of 15 ms cannot tell a 2 % effect from noise
            calib, stag = w.calibration, w.staggered
            dt_long, _ = w.run(500, 2, None)
            cfg.setdefault("extra", {})["long_run"] = {"steps": 500, "ms_per_step": 1000.0 * dt_long / 500, "reads_per_s": n * 500 / dt_long,
                                                       "schedule": ("staggered, %d steps in flight" % w.used_lanes) if (w.staggered and w.used_lanes > 1) else "one step at a time"}
            w.calibration, w.staggered = calib, stag
        except Exception as e:
            cfg.setdefault("extra", {})["long_run"] = {"error": repr(e)}
        try:  # the same workload with WHOLE steps in flight, each on its lane's stream (see Workload.allow_whole_steps)
            if len(w.lanes) > 1:
                calib, stag = w.calibration, w.staggered
                w.staggered = False
                w.whole_lanes = 3  # (four whole steps in flight are four launches of the position-hinted pass side by side: 0.57 ms per step against 0.52)
                dt_w, _ = w.run(300, 3, None)
                w.whole_lanes = None
                cfg.setdefault("extra", {})["whole_steps_in_flight"] = {
                    "steps": 300, "ms_per_step": 1000.0 * dt_w / 300, "reads_per_s": n * 300 / dt_w,
                    "schedule": "%d steps in flight, each on a stream of its own" % w.used_lanes,
                    "position_hinted_pass_ms_per_launch": (ctx.kernel_times() or [(None, None)])[0][1]}
                w.calibration, w.staggered = calib, stag
        except Exception as e:
            cfg.setdefault("extra", {})["whole_steps_in_flight"] = {"error": repr(e)}
        try:  # BASELINE configs[3] as far as one GPU goes: the reads of 1000 samples, every rank's share of cfg4
            w4 = Workload(torch, gtx, ctx, device, d_seq, d_pos, 1000, samples=np.random.default_rng(4242).integers(0, 1000, size=n).astype(np.uint32),
                          hint=not args.no_hint, lanes=args.lanes)
            w4.staggered = args.schedule == "staggered" and len(w4.lanes) >= 2
            w4.allow_whole_steps = True  # (a leg: no roofline is priced on it)
            dt4, _ = w4.run(20, 2, None)
            k4 = ctx.kernel_times()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                w4.step(0)
            torch.cuda.synchronize()
            alone4 = 1000.0 * (time.perf_counter() - t0) / 4
            f4 = w4.result_facts()
            cfg.setdefault("extra", {})["cfg4_one_gpu"] = {
                "workload": "cfg4's share of one GPU: %d reads of 1000 samples (a sample per read at random), the accumulator block of all 1000 samples "
                            "(%d bytes), no exchange" % (n, w4.reduced_bytes),
                "reads_per_s": n * 20 / dt4, "ms_per_step": 1000.0 * dt4 / 20, "steps": 20, "step_alone_ms": alone4,
                "schedule": ("staggered, %d steps in flight" % w4.used_lanes) if (w4.staggered and w4.used_lanes > 1) else
                            ("%d steps in flight, each on a stream of its own" % w4.used_lanes) if w4.used_lanes > 1 else "one step at a time",
                "calibration": w4.calibration,
                "align_kernels": {k[0]: {"ms": k[1], "tasks_completed": k[2]} for k in k4},
                "reads_overflowed": f4["reads_overflowed"], "score_items_refused": f4["score_items_refused"],
                "cells_at_saturation_guard": f4["cells_at_saturation_guard"], "nonref_genotype_calls": f4["nonref_genotype_calls"]}
            w4.close()
            del w4
        except Exception as e:
            cfg.setdefault("extra", {})["cfg4_one_gpu"] = {"error": repr(e)}
    if n_gpus == 1 and not args.no_extra:
        try:
            cfg.setdefault("extra", {})["pcie_fed"] = extra_pcie_fed(w, torch, gtx)
        except Exception as e:  # the extra line must never cost the main one
            cfg.setdefault("extra", {})["pcie_fed"] = {"error": repr(e)}
    w.close()
    if n_gpus == 1 and n_samples == 1 and not args.no_extra:
        try:
            cfg.setdefault("extra", {})["pipeline"] = extra_pipeline(args, torch, gtx, synth, device, ctx, ref, records)
            # the same leg with four times the host threads (and reads): where the host side stops scaling
            wide = extra_pipeline(args, torch, gtx, synth, device, ctx, ref, records, n=16_000_000, threads=64)
            cfg["extra"]["pipeline_64_threads"] = {k: wide.get(k) for k in ("reads", "reads_per_s", "wall_s", "host_threads", "bam_files", "host_thread_seconds",
                                                                              "slowest_thread_s", "records_per_s_per_thread", "stage_bound_reads_per_s",
                                                                              "vcf_equals_resident_run", "bam_write_s_before_the_clock", "native_loop", "error")}
        except Exception as e:  # the extra line must never cost the main one
            cfg.setdefault("extra", {})["pipeline"] = {"error": repr(e)}
        try:
            cfg.setdefault("extra", {})["shrink"] = extra_shrink(args, torch, gtx, synth, device, ref, records)
        except Exception as e:
            cfg.setdefault("extra", {})["shrink"] = {"error": repr(e)}
    if n_gpus == 1 and n_samples == 1 and not args.no_extra:
        try:
            cfg.setdefault("extra", {})["regions"] = extra_regions(args, torch, gtx, synth, device, ref)
        except Exception as e:  # the extra line must never cost the main one
            cfg.setdefault("extra", {})["regions"] = {"error": repr(e)}
    if n_gpus == 1 and not args.no_extra:
        del d_seq, w
        torch.cuda.empty_cache()
        try:
            cfg.setdefault("extra", {})["cfg3"] = extra_cfg3(args, torch, gtx, synth, device, ref, "cfg3")
            cfg.setdefault("extra", {})["cfg3_clusters"] = extra_cfg3(args, torch, gtx, synth, device, ref, "clusters")
        except Exception as e:  # the extra line must never cost the main one
            cfg.setdefault("extra", {})["cfg3"] = {"error": repr(e)}
        try:
            cfg.setdefault("extra", {})["long_reads"] = extra_long_reads(args, torch, gtx, synth, device, ref)
        except Exception as e:
            cfg.setdefault("extra", {})["long_reads"] = {"error": repr(e)}
        try:
            cfg.setdefault("extra", {})["repeats"] = extra_repeats(args, torch, gtx, synth, device, ref)
        except Exception as e:
            cfg.setdefault("extra", {})["repeats"] = {"error": repr(e)}
        try:
            cfg.setdefault("extra", {})["genome_like"] = extra_genome_like(args, torch, gtx, synth, device)
        except Exception as e:
            cfg.setdefault("extra", {})["genome_like"] = {"error": repr(e)}
        try:
            cfg.setdefault("extra", {})["cfg5"] = extra_cfg5(args, torch, gtx, synth, device)
        except Exception as e:
            cfg.setdefault("extra", {})["cfg5"] = {"error": repr(e)}
    emit_line(out)
    if dist is not None:
        dist.destroy_process_group()
    return 0


def _leg_summary(leg):
    """one extra leg in a few numbers (the whole leg is in the full record)"""
    if not isinstance(leg, dict):
        return leg
    if leg.get("error"):
        return {"error": str(leg["error"])[:120]}
    keep = ("reads_per_s", "ms_per_step", "regions_per_s", "step_alone_ms", "one_at_a_time_ms_per_step", "pcie_gbs", "reads_overflowed")
    s = {k: (round(leg[k], 4) if isinstance(leg[k], float) and leg[k] < 1e6 else (float("%.5g" % leg[k]) if isinstance(leg[k], float) else leg[k]))
         for k in keep if leg.get(k) is not None}
    if isinstance(leg.get("with_depth_cap"), dict):
        s["records_per_s"] = leg["with_depth_cap"].get("records_per_s")
    return s


def compact_line(out):
    """The ONE line of stdout: the contract's keys, `roofline`, `cpu_baseline` and a config of a few hundred bytes.  The driver
    keeps the last 8 KB of stdout only and parsed nothing of round 5's 24 KB line (BENCH_r05.json: "parsed": null), so every
    extra leg is a summary here and the whole record goes to gpurun_out/bench_full.json (`full_record`; a copy of the round's
    last one is committed as profiles/rNN_bench_full.json)."""
    cfg, roof, cpu = out["config"], dict(out["roofline"]), out.get("cpu_baseline")
    r = {k: roof.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "units_per_launch",
                                  "algorithmic_bytes_per_read", "algorithmic_bytes_what", "longest_kernel")}
    if roof.get("traffic") and roof.get("units_per_launch"):
        r["traffic_bytes_per_read"] = round(roof["traffic"] / roof["units_per_launch"], 1)
    if isinstance(roof.get("align_kernels"), dict):
        r["align_kernels_ms"] = {k: round(v["ms"], 4) if isinstance(v, dict) else v for k, v in roof["align_kernels"].items()}
    c = None
    if cpu:
        c = {k: cpu.get(k) for k in ("value", "unit", "cores", "kind", "sample", "cpu_model")}
        if isinstance(cpu.get("all_cores"), dict):
            c["all_cores"] = {k: cpu["all_cores"].get(k) for k in ("value", "cores")}
    st = cfg.get("streams") or {}
    small = {"workload": cfg["workload"].split(";")[0],
             "samples": cfg.get("samples"), "reads_per_gpu": cfg.get("reads_per_gpu"), "haplotypes": cfg.get("haplotypes"),
             "index_keys": cfg.get("index_keys"), "parallelism": cfg.get("parallelism"),
             "schedule": "%s, %s steps in flight" % (st.get("schedule"), st.get("steps_in_flight")), "step_alone_ms": st.get("step_alone_ms"),
             "reduce": cfg.get("reduce"), "reduce_ms": cfg.get("reduce_ms"), "reduced_bytes_per_step": cfg.get("reduced_bytes_per_step"),
             "reads_aligned": cfg.get("reads_aligned"), "reads_overflowed": cfg.get("reads_overflowed"),
             "score_items_refused": cfg.get("score_items_refused"), "cells_at_saturation_guard": cfg.get("cells_at_saturation_guard")}
    rc = cfg.get("reduce_check")
    if isinstance(rc, dict):
        small["reduce_check"] = {"equal": rc.get("equal"), "saturation_guard_replay_across_ranks": rc.get("saturation_guard_replay_across_ranks")}
    cs = cfg.get("calls_checksum")
    if isinstance(cs, dict):
        small["vcf"] = {k: cs.get(k) for k in ("vcf_sha256", "final_vcf_sha256", "matches_pinned", "error") if cs.get(k) is not None}
    if isinstance(cfg.get("extra"), dict):
        small["extra"] = {k: _leg_summary(v) for k, v in cfg["extra"].items()}
    small["full_record"] = "gpurun_out/bench_full.json (every leg whole; the round's last: profiles/r06_bench_full.json)"
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data")}
    line["config"], line["roofline"], line["cpu_baseline"] = small, r, c
    return line


def emit_line(out):
    full = json.dumps(out)
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "bench_full.json"), "w") as f:
            f.write(full + "\n")
    except OSError:
        pass
    # (NOT to stderr: the driver's record of a run is a bounded tail of stdout AND stderr together -- round 4's 17 KB line was
    #  parsed, round 5's 24 KB line was not -- so nothing long may follow the line either)
    line = full if os.environ.get("GTX_BENCH_FULL_LINE") else json.dumps(compact_line(out))
    sys.stdout.write(line + "\n")
    sys.stdout.flush()


# Algorithmic bytes per forward task of each alignment kernel: what ITS algorithm has to move.
# The global-lookup passes are priced as SURVEY.md 8(d) prices the reference (388 key slots probed per read-orientation).
# The position-hinted pass proves the outcome of those probes from per-position flags and never issues them: 20 B meta +
# 80 B bases (plane row) + 84 B reference planes + 48 B position flags + 4 B filter word + 24 B record = 260 B (round 2's 268 B
# minus the 8-byte header of the reverse record, which GTX_FLAG_FORWARD_ONLY reads no longer get).
# Bytes per read the position-hinted pass is priced with.  COMPULSORY: what has to cross the HBM interface for one read whatever the
# kernel does -- 20 meta + 80 plane row + 24 record + 2 (its byte of the side array, its share of the queue words): the roofline's
# `achieved`.  WITH_TABLES: the same plus what the kernel reads of the per-position tables (84 reference planes + 48 flag words + 4
# filter word) -- bytes ten neighbouring reads share and the caches serve; rounds 2-4 priced the roofline with it (kept beside the
# other, labelled).
HINTED_COMPULSORY_BYTES, HINTED_BYTES_WITH_TABLES = 126, 260
KERNEL_BYTES = {"gtx_align_hinted_kernel": HINTED_COMPULSORY_BYTES, "gtx_align_express4_kernel": ALGO_BYTES_PER_READ, "gtx_align_kernel": ALGO_BYTES_PER_READ,
                "gtx_align_big_kernel": ALGO_BYTES_PER_READ}


def dominant_kernel(pass_ms, n_pass2, n, align_avg_ms, kern):
    """roofline object for the dominant kernel = the one that moves most of the step's algorithmic bytes (at cfg2 the
    position-hinted pass: 97 % of the reads); when another launch takes longer (the general pass over the last 0.7 % of the
    reads is a latency chain of about the same duration) it is named in `longest_kernel` and listed in `align_kernels` with
    its own bytes, duration and rate.  Durations are HIP events recorded inside gtx_align_batch on the launch stream around
    each launch (gtx_ctx_kernel_times)."""
    longest = None
    if kern:  # [(name, ms, units completed)]
        name, ms, units = max(kern, key=lambda k: KERNEL_BYTES[k[0]] * k[2])
        longest = max(kern, key=lambda k: k[1])[0]
        passes = {k[0]: {"ms": k[1], "tasks_completed": k[2], "algorithmic_bytes_per_task": KERNEL_BYTES[k[0]],
                         "achieved_gbs": (KERNEL_BYTES[k[0]] * k[2] / (k[1] * 1e-3) / 1e9) if k[1] > 0 else 0.0} for k in kern}
    else:
        name, ms, units = "gtx_align_express4_kernel", (pass_ms[0] if pass_ms[0] > 0 else align_avg_ms), (n - n_pass2 if pass_ms[0] > 0 else n)
        passes = {"express": pass_ms[0], "general": pass_ms[1], "hbm_tables": pass_ms[2], "tasks_handed_to_general": n_pass2}
    per = KERNEL_BYTES.get(name, ALGO_BYTES_PER_READ)
    achieved = per * units / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    total_ms = sum(k[1] for k in kern) if kern else align_avg_ms
    # the whole alignment step priced as the reference's algorithm (every read at SURVEY's 3 296 B): how fast a
    # probe-per-key implementation would have to move data to keep up -- continuity with round 1, NOT a hardware fraction
    ref_equiv = ALGO_BYTES_PER_READ * n / (total_ms * 1e-3) / 1e9 if total_ms > 0 else 0.0
    with_tables = None
    if name == "gtx_align_hinted_kernel" and ms > 0:
        gbs = HINTED_BYTES_WITH_TABLES * units / (ms * 1e-3) / 1e9
        with_tables = {"bytes_per_read": HINTED_BYTES_WITH_TABLES, "gbs": gbs, "frac": gbs / HBM_PEAK_GBS,
                       "note": "the pricing of rounds 2-4: the compulsory bytes plus the per-position tables the kernel reads (shared by neighbouring reads, "
                               "served by the caches) -- not a fraction of HBM bandwidth"}
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "kernel": name, "kernel_ms": ms, "units_per_launch": units, "algorithmic_bytes_per_read": per,
            "algorithmic_bytes_what": "compulsory: 20 meta + 80 plane row + 24 record + 2 side array / queue words" if name == "gtx_align_hinted_kernel" else "SURVEY 8(d)",
            "priced_with_the_shared_tables": with_tables, "longest_kernel": longest,
            "align_kernels": passes, "align_all_kernels_ms": total_ms, "align_wall_ms": align_avg_ms,
            "reference_algorithm_equivalent": {"bytes_per_read": ALGO_BYTES_PER_READ, "gbs": ref_equiv,
                                               "note": "SURVEY 8(d) pricing (388 probes per read) of all reads over the sum of the alignment kernels; "
                                                       "exceeds the HBM peak because the position-hinted pass does not issue those probes"}}


if __name__ == "__main__":
    sys.exit(main())
