#!/usr/bin/env python3
"""Fuzz of gtx_reads_* and gtx_bam_shrink (graphtyper_amd/csrc/gtx_bam.cpp, gtx_shrink.inl), not collected by pytest (tests/test_bam_ingest.py runs a few seeds of
it): corrupted BAM payloads (re-compressed, so the BGZF layer is intact and the record parser sees the damage), corrupted
headers, truncated payloads / files, corrupted BGZF bytes.  Each case runs in a subprocess; anything but a clean exit with
records or a GTX_ERR_* is a finding.
    python tests/fuzz_bam.py 0 2000"""
import os, subprocess, sys, struct
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import bam_writer as bw

CHILD = r'''
import sys
sys.path.insert(0, sys.argv[3])
from graphtyper_amd import lib as gtx
try:
    r = gtx.Reads([sys.argv[1]], region=(sys.argv[2] if sys.argv[2] else None))
    n = 0
    while True:
        recs, seq = r.next(64)
        if len(recs) == 0: break
        n += len(recs)
    r.close()
    print("ok", n)
except Exception as e:
    print("err", str(e)[:80])
try:  # the pre-filter reads whole records (names, qualities, tags) from the same file
    st = gtx.bam_shrink(sys.argv[1], [("chrA", 100, 2000)], sys.argv[1] + ".out", gtx.shrink_params(min_read_len=20, min_num_matching=10))
    print("ok shrink", st["records_written"])
except Exception as e:
    print("err shrink", str(e)[:80])
'''

def payload(rng, n=60):
    refs = [("chrA", 5000), ("chrB", 9000)]
    header = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs) + "@RG\tID:a\tSM:s1\n"
    head = b"BAM\1" + struct.pack("<i", len(header)) + header.encode() + struct.pack("<i", len(refs))
    for name, length in refs:
        head += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", length)
    recs = b""
    pos = np.sort(rng.integers(0, 3000, size=n))
    for k, p in enumerate(pos):
        L = int(rng.choice([100, 150, 151]))
        codes = rng.choice([1, 2, 4, 8], size=L).astype(np.uint8)
        aux = [("NM", "C", 1), ("AS", "C", 100), ("XS", "C", 50), ("RG", "Z", "a"), ("XB", "B", ("s", [1, 2, 3]))]
        recs += bw.record("r%d" % k, 0, 0, int(p), 60, [("S", 3), ("M", L - 3)], 0, int(p) + 100, 300, codes, aux)
    return head, recs

def run(seed0, seed1, tmp="/tmp"):
    bad = 0
    for seed in range(seed0, seed1):
        rng = np.random.default_rng(seed)
        head, recs = payload(rng)
        kind = seed % 4
        raw = bytearray(head + recs)
        if kind == 0:      # flip bytes in the records
            for _ in range(int(rng.integers(1, 6))):
                raw[int(rng.integers(len(head), len(raw)))] = int(rng.integers(0, 256))
            data = bw.bgzf(bytes(raw), block=int(rng.choice([300, 5000, 60000])))
        elif kind == 1:    # flip bytes in the header
            for _ in range(int(rng.integers(1, 4))):
                raw[int(rng.integers(0, len(head)))] = int(rng.integers(0, 256))
            data = bw.bgzf(bytes(raw))
        elif kind == 2:    # truncate the payload / the file
            cut = int(rng.integers(1, len(raw)))
            data = bw.bgzf(bytes(raw[:cut]))
            if rng.random() < 0.5:
                data = data[:int(rng.integers(1, len(data)))]
        else:              # corrupt the BGZF stream itself
            data = bytearray(bw.bgzf(bytes(raw), block=int(rng.choice([300, 5000, 60000]))))
            for _ in range(int(rng.integers(1, 6))):
                data[int(rng.integers(0, len(data)))] = int(rng.integers(0, 256))
            data = bytes(data)
        path = os.path.join(tmp, "gtx_fuzz_%d_%d.bam" % (os.getpid(), seed))
        open(path, "wb").write(data)
        region = "chrA:100-2000" if seed % 3 == 0 else ""
        p = subprocess.run([sys.executable, "-c", CHILD, path, region, os.path.dirname(HERE)], capture_output=True, text=True, timeout=120)
        if p.returncode != 0 or not (p.stdout.startswith("ok") or p.stdout.startswith("err")) or "shrink" not in p.stdout:
            bad += 1
            print("FINDING seed", seed, "kind", kind, "rc", p.returncode, p.stdout[:100], p.stderr[-300:])
        os.remove(path)
        if os.path.exists(path + ".out"):
            os.remove(path + ".out")
    return bad


if __name__ == "__main__":
    a, b = int(sys.argv[1]), int(sys.argv[2])
    print("done", a, b, "findings", run(a, b))
