#!/usr/bin/env python3
"""Randomised long-running parity sweep on the CPU (not collected by pytest): the kernel sources through the host
emulation against the oracle over seeds x graph shapes x error / N rates x read lengths x region offsets, for both builds
of pass 1; with --stream also the scoring path (gtx_stream -> align -> score -> calls -> phase flags).
    python tests/stress_emu.py 100 110            # alignment, seeds 100..109
    python tests/stress_emu.py --stream 300 310
    python tests/stress_emu.py --gpu 700 704      # the same cases through the C ABI on the device (dense records; --stream: the
                                                  # scorer's first stage behind the alignment for every second case)"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import harness  # noqa: E402
import scenarios  # noqa: E402
from graphtyper_amd import lib as gtx  # noqa: E402
from oracle_lib import Oracle  # noqa: E402
from test_emu_parity import check_align, run_stream  # noqa: E402


BACKEND = harness.EmuBackend


def align_seed(seed):
    rng = np.random.default_rng(seed)
    n = 0
    for kind in ["snp1k", "snp25", "indel", "cluster", "snp100", "snp7", "rows", "cfg3", "satellite", "repeat", "neardup"]:
        err = float(rng.choice([0.0, 0.005, 0.03]))
        n_rate = float(rng.choice([0.0, 0.001, 0.01]))
        read_len = int(rng.choice([100, 125, 150, 151, 187, 200, 250, 256]))  # (> 160: the eight-k-mer build of pass 0)
        rb = int(rng.choice([0, 1000, 1000000]))
        ref, recs, codes, pos = scenarios.synthetic_case(kind, n_ref=50000, n_reads=600 if kind in ("satellite", "repeat") else 1500, region_begin=rb,
                                                       err=err, n_rate=n_rate, seed=seed, read_len=read_len)
        aav = kind in ("cluster", "cfg3")
        g = gtx.graph_from_records(ref, recs, region_begin=rb, add_all_variants=aav)
        o = Oracle(ref, recs, region_begin=rb, add_all_variants=aav)
        # paired flags: unpaired, discordant (both orientations), proper pair (forward only); ragged lengths
        flags = rng.choice([0, 1 | 64, 1 | 2 | 32 | 64], size=len(codes)).astype(np.uint16)
        isize = rng.integers(-2000, 2000, size=len(codes))
        if rng.random() < 0.5:  # IUPAC sets (2-, 3-, 4-base) on 1 % of the bases: the true base widened, or an unrelated set
            codes = codes.copy()
            amb = rng.random(codes.shape) < 0.01
            extra = rng.integers(1, 16, size=codes.shape).astype(np.uint8)
            codes[amb] = np.where(rng.random(int(amb.sum())) < 0.8, codes[amb] | extra[amb], extra[amb])
        reads = [c[:int(n)] for c, n in zip(codes, rng.integers(max(50, read_len - 60), read_len + 1, size=len(codes)))]
        for mode in ["lean", "wide"]:
            os.environ["GTX_EXPRESS4"] = mode
            os.environ["GTX_HINT_BUILD"] = "dense" if mode == "wide" else "lean"  # (both builds of pass 0 on every graph shape)
            try:
                # position hints: right for most reads, a few bases off or absent for the others; check_align also runs
                # the batch without hints, with shifted hints and with other reads' hints (records must not depend on them)
                hint = np.where(rng.random(len(pos)) < 0.9, pos, np.where(rng.random(len(pos)) < 0.5, pos + rng.integers(-3, 4, size=len(pos)), -1))
                check_align(BACKEND(g), o, reads, flags=flags, isize=isize, pos=hint)
            except AssertionError:
                print("FAIL", dict(seed=seed, kind=kind, mode=mode, err=err, n_rate=n_rate, read_len=read_len, region_begin=rb), flush=True)
                raise
            n += 1
    return n


def stream_seed(seed):
    rng = np.random.default_rng(seed)
    n = 0
    for kind in ["snp100", "snp25", "indel"]:
        ns = int(rng.choice([1, 2, 5]))
        rb = int(rng.choice([0, 310000]))
        ref, recs, codes, rec = scenarios.paired_case(kind, n_ref=30000, n_pairs=700, region_begin=rb, n_samples=ns, seed=seed,
                                                     discordant_frac=float(rng.choice([0.0, 0.1, 0.3])), dup_frac=0.05, lowq_frac=0.1)
        o = Oracle(ref, recs, region_begin=rb)
        for mode in ["lean", "wide"]:
            os.environ["GTX_EXPRESS4"] = mode
            os.environ["GTX_HINT_BUILD"] = "dense" if mode == "wide" else "lean"
            try:
                if BACKEND is not harness.EmuBackend:
                    os.environ["HARNESS_TRIAGED"] = "1" if (n + seed) % 2 else "0"
                    os.environ["HARNESS_TRIAGED_WORDS"] = "1" if seed % 2 else "0"
                run_stream(BACKEND(gtx.graph_from_records(ref, recs, region_begin=rb)), o, codes, rec, n_samples=ns)
            except AssertionError as e:
                if str(e) == "":  # run_stream's "not vacuous" check on a small case: not a parity failure
                    continue
                print("FAIL", dict(seed=seed, kind=kind, mode=mode, n_samples=ns, region_begin=rb), flush=True)
                raise
            n += 1
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("first", type=int)
    ap.add_argument("last", type=int)
    ap.add_argument("--stream", action="store_true")
    ap.add_argument("--gpu", action="store_true", help="through the C ABI on cuda:0 instead of the host emulation")
    a = ap.parse_args()
    if a.gpu:
        global BACKEND
        BACKEND = harness.GpuBackend
        os.environ["HARNESS_COMPACT"] = "1"
        os.environ["HARNESS_SHORT_READS"] = "1"
    else:
        gtx.build()
    t0, n = time.time(), 0
    for seed in range(a.first, a.last):
        n += stream_seed(seed) if a.stream else align_seed(seed)
        print("seed %d ok, %d cases, %.0f s" % (seed, n, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
