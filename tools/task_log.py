#!/usr/bin/env python
"""The general pass' tasks one by one (profiling build): GTX_LIB=libgtx_prof.so python tools/task_log.py [--reads N] [--kind cfg2|cfg3]
Runs a few steps of the cfg2 (or cfg3) workload one at a time, reads gtx_ctx_profile_log behind the last one and writes
gpurun_out/task_log_<kind>.npz: the log (task, workgroup, start, cycles, hardware id, record head, phase cycles), and the plane
rows + meta of the slowest tasks' reads so that the host emulation can replay them."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GTX_LIB", "libgtx_prof.so")
import bench  # noqa: E402

PHASES = ["load read", "keys+exact probes", "exact labels", "chain exact", "hamming lookup", "chain hamming", "walk starts", "walk ends", "filters", "record"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--kind", default="cfg2")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--keep", type=int, default=400, help="slowest tasks whose reads are saved")
    a = ap.parse_args()
    import torch
    from graphtyper_amd import lib as gtx
    from graphtyper_amd import synth
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if a.kind == "cfg2":
        ref, records, ref_str = bench.cfg2_graph_inputs(synth, bench.REGION_LEN, 1000)
        ctx = gtx.Context(gtx.graph_from_records(ref_str, records, region_begin=bench.REGION_BEGIN), device=0)
        d_seq, d_pos = bench.make_reads_on_device(torch, ref, records, a.reads, seed=bench.CFG2_READ_SEED, device=device)
        w = bench.Workload(torch, gtx, ctx, device, d_seq, d_pos, 1, lanes=1)
    else:
        raise SystemExit("kind %s: not wired" % a.kind)
    for _ in range(a.steps - 1):
        w.step(0)
    torch.cuda.synchronize()
    ctx.profile_log()  # (empties the log)
    ctx.pass_times()
    w.step(0)
    torch.cuda.synchronize()
    log = ctx.profile_log()
    kern = ctx.kernel_times()
    print("kernel times / tasks:", kern)
    n = len(log)
    cyc = log[:, 3].astype(np.int64)
    print("%d tasks logged; cycles: mean %.0f median %.0f p90 %.0f p99 %.0f max %d" %
          (n, cyc.mean(), np.median(cyc), np.percentile(cyc, 90), np.percentile(cyc, 99), cyc.max()))
    t0 = log[:, 2].astype(np.int64)
    span = (t0 + cyc).max() - t0.min()
    print("first start to last end: %d cycles; sum of task cycles %d; workgroups with a task %d" % (span, cyc.sum(), len(np.unique(log[:, 1]))))
    ph = log[:, 6:16].astype(np.int64)
    tot = ph.sum(0)
    for k, nm in enumerate(PHASES):
        print("  %-20s mean %8.0f  %5.1f%%   (slowest 5%% of tasks: mean %8.0f)" %
              (nm, ph[:, k].mean(), 100.0 * tot[k] / max(tot.sum(), 1), ph[np.argsort(cyc)[-max(n // 20, 1):], k].mean()))
    # per workgroup: its tasks in order, when its last one ended
    order = np.argsort(cyc)[::-1]
    print("slowest tasks: task cycles start-offset workgroup rec0 rec1 | phases")
    for i in order[:25]:
        print("  %9d %8d %9d %5d %08x %08x | %s" % (log[i, 0], cyc[i], t0[i] - t0.min(), log[i, 1], int(log[i, 5]) & 0xFFFFFFFF, int(log[i, 5]) >> 32,
                                                  " ".join("%6d" % x for x in ph[i])))
    ends = {}
    for i in range(n):
        b = int(log[i, 1])
        ends[b] = max(ends.get(b, 0), int(t0[i] + cyc[i] - t0.min()))
    e = np.array(sorted(ends.values()))
    print("workgroup end offsets: p10 %d median %d p90 %d p99 %d max %d" % tuple(np.percentile(e, q) for q in (10, 50, 90, 99, 100)))
    first = {}
    for i in np.argsort(t0):
        first.setdefault(int(log[i, 1]), i)
    fi = np.array(list(first.values()))
    rest = np.setdiff1d(np.arange(n), fi)
    print("a workgroup's first task: mean %.0f cycles (start offset mean %.0f, max %d); later tasks: mean %.0f" %
          (cyc[fi].mean(), (t0[fi] - t0.min()).mean(), (t0[fi] - t0.min()).max(), cyc[rest].mean() if len(rest) else 0))
    keep = order[:a.keep]
    reads = (log[keep, 0] >> 1).astype(np.int64)
    d_seq_p, d_meta, _ = w.sets[0 if len(w.sets) == 1 else (w.steps_done - 1) % len(w.sets)]
    idx = torch.from_numpy(reads).to(device)
    rows = d_seq_p[idx].cpu().numpy()
    meta = d_meta[idx].cpu().numpy()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "task_log_%s.npz" % a.kind), log=log, rows=rows, meta=meta, keep=keep, stride=w.stride)
    print("saved gpurun_out/task_log_%s.npz" % a.kind)


if __name__ == "__main__":
    main()
