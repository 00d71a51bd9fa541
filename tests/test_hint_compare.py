"""The compare of the position-hinted pass (graphtyper_amd/csrc/hinted.hpp): the bit-plane form the kernel runs must give
the counters of the nibble-word form it replaced -- substitutions / ambiguous bases per k-mer and per k-mer half, sets that
miss the reference base, mismatches by count_mismatches' rule up to every k-mer boundary -- for every read length, every
phase of the position, IUPAC codes, '=' and N in the read and N in the reference."""
import ctypes as C

import numpy as np

import harness  # noqa: F401  (builds the emulation library)
from graphtyper_amd import lib as gtx


def test_plane_compare_equals_nibble_compare():
    b = harness.EmuBackend(gtx.graph_from_records("ACGT" * 40, []))
    L_ = b.L
    rng = np.random.default_rng(11)
    out = (C.c_uint32 * 14)()
    n_cases = 0
    for trial in range(3000):
        n_base = 1200
        base = rng.choice([1, 2, 4, 8], size=n_base).astype(np.uint8)
        base[rng.random(n_base) < 0.01] = 15  # N in the reference
        L = int(rng.integers(63, 161))
        idx = int(rng.integers(0, n_base - 200))
        read = base[idx:idx + L].copy()
        mode = trial % 4
        rate = [0.0, 0.01, 0.05, 0.3][mode]
        e = rng.random(L) < rate
        read[e] = rng.choice([1, 2, 4, 8], size=int(e.sum()))
        a = rng.random(L) < [0.0, 0.004, 0.03, 0.2][mode]
        read[a] = rng.integers(0, 16, size=int(a.sum()))  # any code: '=', IUPAC sets, N
        row = np.zeros(80, np.uint8)
        packed = gtx.pack_nibbles(read[None, :])[0]
        row[:len(packed)] = packed
        if mode == 3:
            row[len(packed):] = rng.integers(0, 256, size=80 - len(packed))  # whatever lies behind the read must not matter
        L_.emu_hint_compare(base.ctypes.data_as(C.c_void_p), C.c_uint32(n_base), C.c_uint32(idx), row.ctypes.data_as(C.c_void_p), C.c_uint32(L), out)
        got = list(out)
        assert got[:7] == got[7:], (trial, L, idx, [hex(x) for x in got])
        n_cases += got[6] != 0
    assert n_cases > 1500  # (not only reads without a difference)
