"""The index sweep listed in two ways (gtx_host.cpp: enumerate_kmers): every 32-mer by the host, or only those that walk
through a site with the in-node runs left to the device.  The runs, expanded the way the device kernels expand them
(gtx_index_dev.hip: k_emit_runs, k_place_listed), must give the same list in the same order -- the order inside a key is
the order of PHIndex's label lists (indexer.cpp:246-291).  Host only; the device's expansion is checked by the index
comparison of tests/test_gpu_index.py."""
import ctypes as C

import numpy as np
import pytest

import harness
import scenarios
from graphtyper_amd import lib as gtx, synth


def check(ref, recs, **kw):
    b = harness.EmuBackend(gtx.graph_from_records(ref, recs, region_begin=1000000, **kw))
    b.L.emu_enumeration_check.restype = C.c_long
    listed, runs = C.c_uint64(), C.c_uint64()
    n = b.L.emu_enumeration_check(C.c_void_p(b.h), C.byref(listed), C.byref(runs))
    assert n >= 0, "the two listings differ at k-mer %d" % (-n - 1)
    return n, listed.value, runs.value


@pytest.mark.parametrize("kind", ["snp1k", "snp100", "snp25", "indel", "cluster", "cfg3", "snp7"])
def test_runs_give_the_sweep(kind):
    ref, recs, _, _ = scenarios.synthetic_case(kind, n_ref=60000, n_reads=1)
    n, listed, runs = check(ref, recs, add_all_variants=kind in ("cluster", "cfg3"))
    assert n > 0
    if kind == "snp1k":  # nearly everything is the device's: 31 positions behind each site and the site itself stay
        assert runs >= 59 and listed < n // 8


def test_nodes_with_other_characters_stay_with_the_host():
    base = synth.make_reference(30000, seed=5)
    ref = np.frombuffer(synth.bases_to_str(base).encode(), np.uint8).copy()
    ref[1234] = ord("N")          # inside a long node
    ref[15000:15040] = ord("N")   # a stretch
    ref[20499] = ord("N")         # next to a site (sites at 500 + 1000 k)
    recs = synth.make_snp_records(base, 1000, seed=7, region_begin=1000000)
    n, listed, runs = check(ref.tobytes().decode(), recs)
    assert 0 < runs < 30 and n > 0


def test_short_nodes_and_no_sites():
    base = synth.make_reference(2000, seed=6)
    n, listed, runs = check(synth.bases_to_str(base), [])
    assert runs == 1 and listed == 0 and n == 2000 - 31  # one node: the first 31 windows reach past its start, nothing listed
    recs = synth.make_snp_records(base, 20, seed=8, region_begin=1000000)  # nodes of 19 bases: no run at all
    n, listed, runs = check(synth.bases_to_str(base), recs)
    assert runs == 0 and listed == n
