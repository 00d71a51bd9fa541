"""Alignment on a reference and reads that look like a mapped human sample (synth.make_genome_like_reference: order-5 Markov
background, interspersed repeat families at 5-20 % divergence, STRs, segmental duplications; synth.make_mapped_reads: indel errors,
soft clips, wrong and shifted hints) -- bench.py's `genome_like` leg at test size: every record equals the oracle's GenotypePaths
with the mapper's hints, without hints, with shifted and with foreign hints; no table of any pass may overflow."""
import numpy as np
import pytest

import harness
from graphtyper_amd import lib as gtx
from graphtyper_amd import synth
from oracle_lib import Oracle
from test_emu_parity import check_align


def genome_like_case(Backend, n_ref=120000, n_reads=2500, seed=3):
    rb = 1000000
    ref, stats = synth.make_genome_like_reference(n_ref, seed=seed, segdups=2, segdup_len=(2000, 5000))
    assert stats["interspersed"] > 0.4 and stats["str"] > 0.02
    recs = synth.make_snp_records(ref, 1000, seed=seed + 1, region_begin=rb)
    codes, hint, made = synth.make_mapped_reads(ref, recs, n_reads, seed=seed + 2, region_begin=rb, indel_err=0.002, clip_frac=0.08, bad_hint_frac=0.05)
    assert made["indel_reads"] > 50 and made["clipped"] > 50 and made["bad_hints"] > 50
    order = np.argsort(hint, kind="stable")
    ref_s = synth.bases_to_str(ref)
    o = Oracle(ref_s, recs, region_begin=rb)
    b = Backend(gtx.graph_from_records(ref_s, recs, region_begin=rb))
    check_align(b, o, list(codes[order]), pos=hint[order], allow_overflow=False)
    return check_align.hinted_done / float(n_reads)


def test_genome_like_reads_on_the_emulation():
    genome_like_case(harness.EmuBackend)


@pytest.mark.gpu
def test_genome_like_reads_on_the_device():
    import torch  # (before libgtx is loaded: one HIP runtime in the process, torch's)
    assert torch.cuda.is_available()
    share = genome_like_case(harness.GpuBackend, n_ref=400000, n_reads=20000)
    assert 0.2 < share < 0.99, share
