#!/usr/bin/env python3
"""Who leaves the position-hinted pass and why (host emulation of the kernel sources, no GPU): a cfg2-like sample
(SNP every 1 kb, 0.5 % substitutions, 0.1 % N) or another scenario kind; prints, per decline note of pass 0, how many reads
carry it and which pass finished them.    python tools/decline_notes.py [kind] [n_reads]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness  # noqa: E402
import scenarios  # noqa: E402
from graphtyper_amd import lib as gtx  # noqa: E402

NOTES = {1: "exact k-mer, place not provably simple", 3: "one substitution, other half shared", 4: "filter says the half may occur",
         5: "ambiguous base(s), other half shared", 6: ">= 2 substitutions, a clean half is shared", 7: "> 1 ambiguous base (not one half)",
         8: "tail leaves the node (not one SNP site)", 9: "no usable hint / length", 10: "run selection (equal runs, parallel chain)",
         11: "head walk leaves the node", 12: "ambiguous base + substitution(s) not provable", 13: "site clash / record too long", 15: "no note"}

kind = sys.argv[1] if len(sys.argv) > 1 else "snp1k"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
add_all = kind in ("cluster", "cfg3")
ref, recs, codes, pos = scenarios.synthetic_case(kind, n_ref=400000, n_reads=n, region_begin=1000000)
b = harness.EmuBackend(gtx.graph_from_records(ref, recs, region_begin=1000000, add_all_variants=add_all))
seq, lens = harness.pack_ragged(list(codes))
b.align(seq, harness.read_meta(lens, pos=pos))
pass_of, why = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
b.L.emu_pass_of(C.c_void_p(b.h), pass_of.ctypes.data_as(C.c_void_p), why.ctypes.data_as(C.c_void_p), C.c_uint32(n))
n_amb = (codes == 15).sum(1)
print("%s: %d reads; finished by hinted %d, express %d, general %d, HBM tables %d" % (kind, n, *(int((pass_of == k).sum()) for k in range(4))))
print("%-50s %8s %8s %8s %8s   %s" % ("decline note of pass 0", "reads", "express", "general", "hbm", "reads with an N"))
for code in sorted(set(why[pass_of > 0].tolist())):
    m = (why == code) & (pass_of > 0)
    print("%-50s %8d %8d %8d %8d   %d" % (NOTES.get(code, str(code)), m.sum(), (m & (pass_of == 1)).sum(), (m & (pass_of == 2)).sum(),
                                         (m & (pass_of == 3)).sum(), (m & (n_amb > 0)).sum()))
