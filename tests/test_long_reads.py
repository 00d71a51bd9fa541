"""Reads of 161 to 256 bases (2 x 250): the eight-k-mer build of the position-hinted pass (hinted_long.hpp), chosen by the rows'
stride.  Same contract as everywhere: every record equals the oracle's with correct, missing, shifted and foreign hints; the
pass finishes most reads of a sparse graph itself.  CPU: the kernel sources through the host emulation."""
import numpy as np
import pytest

import harness
import scenarios
from graphtyper_amd import lib as gtx
from oracle_lib import Oracle
from test_emu_parity import check_align, run_stream


def long_read_case(Backend, kind, read_len, n_reads, min_share, seed=0, err=None):
    kw = {} if err is None else dict(err=err)
    ref, recs, codes, pos = scenarios.synthetic_case(kind, n_ref=60000, n_reads=n_reads, region_begin=1000000, read_len=read_len, seed=seed, **kw)
    aav = kind in ("cluster", "cfg3")
    o = Oracle(ref, recs, region_begin=1000000, add_all_variants=aav)
    b = Backend(gtx.graph_from_records(ref, recs, region_begin=1000000, add_all_variants=aav))
    rng = np.random.default_rng(seed + 5)
    reads = [c[:int(n)] for c, n in zip(codes, rng.integers(161, read_len + 1, size=len(codes)))]  # ragged: 161 .. read_len
    reads[0] = codes[0][:read_len]
    check_align(b, o, reads, pos=pos)
    assert check_align.hinted_done >= min_share * len(reads), (kind, read_len, check_align.hinted_done)
    return b, o, codes, pos


@pytest.mark.parametrize("kind,read_len,min_share", [("snp1k", 250, 0.9), ("snp1k", 256, 0.9), ("snp100", 250, 0.6), ("snp25", 200, 0.0),
                                                     ("indel", 250, 0.2), ("cfg3", 250, 0.2)])
def test_long_reads_through_pass_0(kind, read_len, min_share):
    long_read_case(harness.EmuBackend, kind, read_len, 1500, min_share)


def test_long_reads_with_many_errors_and_ns():
    """3 % substitutions and 1 % N: holes, runs of one length, twin paths, tails that fail -- whatever the eight-k-mer build decides
    has to be the oracle's result"""
    ref, recs, codes, pos = scenarios.synthetic_case("snp100", n_ref=50000, n_reads=2500, region_begin=7000, read_len=250, err=0.03, n_rate=0.01, seed=9)
    o = Oracle(ref, recs, region_begin=7000)
    b = harness.EmuBackend(gtx.graph_from_records(ref, recs, region_begin=7000))
    check_align(b, o, list(codes), pos=pos)
    assert check_align.hinted_done > 100


def test_long_reads_score_like_the_oracle():
    """the stream through scoring and calls with 250-base reads"""
    ref, recs, codes, pos = scenarios.synthetic_case("snp100", n_ref=40000, n_reads=3000, region_begin=1000000, read_len=250, seed=4)
    o = Oracle(ref, recs, region_begin=1000000)
    b = harness.EmuBackend(gtx.graph_from_records(ref, recs, region_begin=1000000))
    rec = scenarios.stream_records(len(codes), pos, sample=np.arange(len(codes)) % 3, l_qseq=250)
    order = np.argsort(pos, kind="stable")
    run_stream(b, o, codes[order], rec[order], n_samples=3)
