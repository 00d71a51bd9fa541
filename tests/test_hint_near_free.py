"""HINT_NEAR_FREE (gtx_flat.hpp; round 6): bit 31 of a position's second flag word says that the half-key filters hold NONE of the
2 x 16 x 3 sixteen-mers that are one substitution away from the first / last 16 bases of the reference k-mer at the position -- the
position-hinted pass then does not probe the filter for a half of a read's k-mer that differs from the reference's in exactly one
base (hint_kmer, hinted.hpp).  The flag must be exactly the probes' outcome, or the pass would decide differently with it than
without: here the filter's hash is restated in numpy and every position's bit is held to the 96 probes, on the linear reference
and inside the allele windows, for a SNP graph and for the cfg3 graph."""
import numpy as np
import pytest

import scenarios
from graphtyper_amd import lib as gtx

M32 = np.uint64(0xFFFFFFFF)


def _hash(w0, w1):
    """hint_filter_hash (gtx_flat.hpp) on arrays of uint64 holding 32-bit values"""
    h = (w1 * np.uint64(0x9E3779B1)) & M32
    h ^= h >> np.uint64(15)
    h = (h + w0) & M32
    h = (h * np.uint64(0x85EBCA77)) & M32
    h ^= h >> np.uint64(16)
    h2 = (h * np.uint64(0x9E3779B1)) & M32
    return h, h2


def _present(filt, log2_words, w0, w1):
    h, h2 = _hash(w0, w1)
    word = h >> np.uint64(32 - log2_words)
    one = np.uint64(1)
    mask = (one << (h2 >> np.uint64(27))) | (one << ((h2 >> np.uint64(22)) & np.uint64(31))) | (one << ((h2 >> np.uint64(17)) & np.uint64(31))) | \
           (one << ((h2 >> np.uint64(12)) & np.uint64(31)))
    return (filt[word].astype(np.uint64) & mask) == mask


@pytest.mark.parametrize("kind", ["snp100", "cfg3"])
def test_the_bit_is_the_outcome_of_the_96_probes(kind):
    ref, recs, _, _ = scenarios.synthetic_case(kind, n_ref=30000, n_reads=10, region_begin=1000000)
    c = gtx.Context(gtx.graph_from_records(ref, recs, region_begin=1000000, add_all_variants=kind == "cfg3"), device=-1)
    flags = c.hint_table(0).reshape(-1, 2)
    planes = c.hint_table(1).reshape(-1, 4).astype(np.uint64)
    filt = [c.hint_table(3), c.hint_table(4)]
    log2_words = int(np.log2(len(filt[0])))
    assert len(filt[0]) == 1 << log2_words and len(filt[1]) == len(filt[0])
    n = len(flags)
    # the BAM code of every table position from the planes (position 32 q + j at bit j of group q's four words)
    pos = np.arange(n, dtype=np.uint64)
    g, j = (pos >> np.uint64(5)).astype(np.int64), pos & np.uint64(31)
    g = np.minimum(g, len(planes) - 1)
    code = sum(((planes[g, b] >> j) & np.uint64(1)) << np.uint64(b) for b in range(4))
    two = np.full(n, 255, np.uint64)
    for cd, t in ((1, 0), (2, 1), (4, 2), (8, 3)):
        two[code == cd] = t
    bad = (two == 255).astype(np.int64)
    run_bad = np.concatenate([[0], np.cumsum(bad)])
    valid = np.zeros(n, bool)
    valid[:n - 31] = (run_bad[32:] - run_bad[:-32])[:n - 31] == 0
    near_free = np.ones(n, bool)
    for side in range(2):
        # the half's two planes: base k of the half at bit k
        w0 = np.zeros(n, np.uint64)
        w1 = np.zeros(n, np.uint64)
        for k in range(16):
            t = np.roll(two, -(16 * side + k))
            w0 |= (t & np.uint64(1)) << np.uint64(k)
            w1 |= ((t >> np.uint64(1)) & np.uint64(1)) << np.uint64(k)
        for k in range(16):
            for d in (1, 2, 3):
                v0 = w0 ^ (np.uint64(d & 1) << np.uint64(k))
                v1 = w1 ^ (np.uint64(d >> 1) << np.uint64(k))
                near_free &= ~_present(filt[side], log2_words, np.where(valid, v0, 0), np.where(valid, v1, 0))
    near_free &= valid
    got = (flags[:, 1] >> 31).astype(bool)
    differ = np.nonzero(got != near_free)[0]
    assert not len(differ[differ < len(ref) - 31]), "linear reference: %s" % differ[:10]
    # (a window position whose k-mer does not reach into the allele copies the flags of the linear reference's position -- the same
    #  32 bases wherever the window holds them all; at a window's end, where its planes stop, the copied bit describes the linear
    #  reference's k-mer and no read placed on the window has a k-mer there: 160 bases behind the allele, reads of up to 160)
    wrong = np.nonzero((got != near_free) & valid)[0]
    assert not len(wrong), "a k-mer whose bit is not the probes' outcome: %s" % wrong[:10]
    lin = slice(0, len(ref) - 31)
    # (an i.i.d. reference: chance hits of the filter and the other alleles of the sites -- a site every 100 bases clears the bit of
    #  the 32 places whose k-mer lies over it -- are all that clears it)
    assert got[lin].mean() > 0.5
    if kind == "cfg3":
        assert got[len(ref):].sum() > 0  # (bits inside the allele windows as well)
