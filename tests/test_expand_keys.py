"""to_uint64_vec for k-mers with ambiguous bases (src/utilities/type_conversions.cpp:207-266): the general pass makes the list a
key per lane (align_core.inl: expand_keys_lanes); the sequential statement of the reference's loop (expand_keys, which the
oracle's to_uint64_vec is pinned beside in test_oracle_pinned.py) stays as its checker."""
import ctypes as C
import os

import numpy as np

from oracle_lib import Oracle  # noqa: F401  (makes sure the test tree is importable the usual way)

HERE = os.path.dirname(os.path.abspath(__file__))


def _emu():
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(HERE, "emu"), "-s"])
    return C.CDLL(os.path.join(HERE, "emu", "libgtx_emu.so"))


def _both(L, codes):
    ks, kl = np.zeros(400, np.uint64), np.zeros(400, np.uint64)
    ns, nl = C.c_uint32(), C.c_uint32()
    ok = L.emu_expand_keys(codes.ctypes.data_as(C.c_void_p), ks.ctypes.data_as(C.c_void_p), kl.ctypes.data_as(C.c_void_p), C.byref(ns), C.byref(nl))
    return ok, ns.value, nl.value, ks, kl


def test_key_per_lane_equals_the_sequential_list():
    L = _emu()
    rng = np.random.default_rng(77)
    plain = np.array([1, 2, 4, 8], np.uint8)
    seen = {"gave_up": 0, "beyond": 0, "over_97_on_last_base": 0, "lists": 0}
    for it in range(20000):
        codes = plain[rng.integers(0, 4, 32)]
        m = int(rng.integers(0, 9))
        where = rng.choice(32, size=m, replace=False)
        if it % 7 == 0 and m:  # (the one place a list may outgrow 97 keys: the k-mer's last base)
            where[0] = 31
        for t in where:
            codes[t] = rng.integers(0, 16) if rng.random() < 0.7 else 15
        ok, ns, nl, ks, kl = _both(L, codes)
        assert ok == 1, (codes.tolist(), ns, nl)
        seen["lists"] += 1
        seen["gave_up"] += ns == 0
        seen["beyond"] += ns == 0xFFFFFFFF
        seen["over_97_on_last_base"] += 97 < ns < 0xFFFFFFFF
    assert min(seen.values()) > 0, seen


def test_known_lists():
    L = _emu()
    codes = np.full(32, 1, np.uint8)  # A x 32
    ok, ns, nl, ks, _ = _both(L, codes)
    assert ok == 1 and ns == 1 and ks[0] == 0
    codes[31] = 15  # N on the last base: T in place, then A, C, G (plane form: bit 31 of the low / high word)
    ok, ns, nl, ks, _ = _both(L, codes)
    lo, hi = 1 << 31, 1 << 63
    assert ok == 1 and ns == 4 and [int(x) for x in ks[:4]] == [lo | hi, 0, lo, hi]
