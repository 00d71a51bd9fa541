#!/usr/bin/env python3
"""Fuzz of gtx_graph_from_files (graphtyper_amd/csrc/gtx_files.cpp + gtx_graph.cpp), not collected by pytest
(tests/test_constructor_vectors.py runs a few seeds): the reference's own test VCF / FASTA index with damaged bytes, fields
and lines, regions on every contig, SV and merged-variant modes.  Each case runs in a subprocess; anything but a clean exit with
a graph or a GTX_ERR_* is a finding.
    python tests/fuzz_files.py 0 2000"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")

CHILD = r'''
import sys
sys.path.insert(0, sys.argv[6])
from graphtyper_amd import lib as gtx
try:
    g, span = gtx.graph_from_files(sys.argv[1], sys.argv[2], sys.argv[3], add_all_variants=sys.argv[4] == "1", is_sv_graph=sys.argv[5] == "1")
    print("ok", len(g["ref_order"]), span)
except Exception as e:
    print("err", str(e)[:80])
'''


def run(seed0, seed1, tmp="/tmp"):
    vcf = open(os.path.join(GOLD, "index_test.vcf"), "rb").read()
    fai = open(os.path.join(GOLD, "index_test.fa.fai"), "rb").read()
    fa = open(os.path.join(GOLD, "index_test.fa"), "rb").read()
    contigs = [l.split(b"\t")[0].decode() for l in fai.splitlines() if l]
    bad = 0
    for seed in range(seed0, seed1):
        rng = np.random.default_rng(seed)
        d = os.path.join(tmp, "gtx_fuzz_files_%d_%d" % (os.getpid(), seed))
        os.makedirs(d, exist_ok=True)
        v, f, x = bytearray(vcf), bytearray(fa), bytearray(fai)
        kind = seed % 5
        if kind == 0:    # random bytes in the VCF
            for _ in range(int(rng.integers(1, 8))):
                v[int(rng.integers(0, len(v)))] = int(rng.integers(0, 256))
        elif kind == 1:  # field-level damage: drop / duplicate / empty a column, huge numbers, long alleles
            lines = bytes(v).split(b"\n")
            for _ in range(int(rng.integers(1, 4))):
                k = int(rng.integers(2, len(lines)))
                cols = lines[k].split(b"\t")
                if len(cols) < 2:
                    continue
                c = int(rng.integers(0, len(cols)))
                what = int(rng.integers(0, 6))
                if what == 0:
                    del cols[c]
                elif what == 1:
                    cols[c] = b""
                elif what == 2:
                    cols[c] = b"99999999999999999999"
                elif what == 3:
                    cols[c] = b"-" + cols[c]
                elif what == 4:
                    cols[c] = cols[c] * int(rng.integers(2, 400))
                else:
                    cols.insert(c, cols[c])
                lines[k] = b"\t".join(cols)
            v = bytearray(b"\n".join(lines))
        elif kind == 2:  # truncation
            v = v[:int(rng.integers(0, len(v)))]
        elif kind == 3:  # damaged FASTA index
            for _ in range(int(rng.integers(1, 4))):
                x[int(rng.integers(0, len(x)))] = int(rng.choice([ord("0"), ord("9"), 9, 10, ord("-"), ord("x")]))
        else:            # damaged / truncated FASTA
            if rng.random() < 0.5:
                f = f[:int(rng.integers(0, len(f)))]
            else:
                for _ in range(int(rng.integers(1, 6))):
                    f[int(rng.integers(0, len(f)))] = int(rng.integers(0, 256))
        open(os.path.join(d, "t.fa"), "wb").write(bytes(f))
        open(os.path.join(d, "t.fa.fai"), "wb").write(bytes(x))
        open(os.path.join(d, "t.vcf"), "wb").write(bytes(v))
        c = contigs[int(rng.integers(0, len(contigs)))]
        region = c if rng.random() < 0.5 else "%s:%d-%d" % (c, int(rng.integers(0, 50)), int(rng.integers(1, 400)))
        p = subprocess.run([sys.executable, "-c", CHILD, os.path.join(d, "t.fa"), os.path.join(d, "t.vcf"), region,
                            str(int(rng.integers(0, 2))), str(int(rng.integers(0, 2))), os.path.dirname(HERE)],
                           capture_output=True, text=True, timeout=120)
        if p.returncode != 0 or not (p.stdout.startswith("ok") or p.stdout.startswith("err")):
            bad += 1
            print("FINDING seed", seed, "kind", kind, "region", region, "rc", p.returncode, p.stdout[:100], p.stderr[-300:], flush=True)
        for n in ("t.fa", "t.fa.fai", "t.vcf"):
            os.remove(os.path.join(d, n))
        os.rmdir(d)
    return bad


if __name__ == "__main__":
    a, b = int(sys.argv[1]), int(sys.argv[2])
    print("done", a, b, "findings", run(a, b))
