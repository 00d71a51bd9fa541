"""The dense build of the position-hinted pass (hinted.hpp: k-mers over two sites, walks at the read's end over sites with
alleles of any length, allele windows): what the index build lists, that both builds of the pass write the same records, and
that the dense one finishes what it is there for.  CPU: the kernel sources through the host emulation against the oracle."""
import os

import numpy as np
import pytest

import harness
import scenarios
from graphtyper_amd import lib as gtx
from oracle_lib import Oracle
from test_emu_parity import check_align

WIN_BEFORE, WIN_STRIDE = 160, 384


def graph_of(kind, n_ref, n_reads, region_begin=1000000, seed=0):
    ref, recs, codes, pos = scenarios.synthetic_case(kind, n_ref=n_ref, n_reads=n_reads, region_begin=region_begin, seed=seed)
    aav = kind in ("cluster", "cfg3")
    g = gtx.graph_from_records(ref, recs, region_begin=region_begin, add_all_variants=aav)
    return ref, recs, codes, pos, g, aav


def test_windows_are_listed_for_the_sites_that_need_them():
    """an alternative allele gets a window when its site is not a SNP standing alone: an allele of another length than one
    base, or another site within a k-mer's reach; a graph of lone SNPs has none"""
    ref, recs, codes, pos, g, _ = graph_of("cfg3", 60000, 10)
    c = gtx.Context(g, device=-1)
    win = c.hint_table(5).reshape(-1, 8)
    site_win = c.hint_table(6)
    assert len(win) > 0 and len(site_win) == len(g["ref_order"])
    n_flags = len(c.hint_table(0)) // 2
    n_main = int(g["ref_order"][-1] + g["ref_len"][-1] - g["ref_order"][0])
    win_base = (n_main // 64 + 5) * 64
    assert n_flags == win_base + len(win) * WIN_STRIDE + 256
    fv, nv, vlen = g["ref_first_var"], g["ref_nvar"], g["var_len"]
    ro, rl = g["ref_order"], g["ref_len"]
    seen = 0
    for r in range(len(ro) - 1):
        first, count = int(site_win[r]) & 0xFFFFFF, int(site_win[r]) >> 24
        lens = [int(vlen[fv[r] + a]) for a in range(nv[r])]
        lone_snp = all(x == 1 for x in lens) and nv[r] <= 4 and not (r > 0 and nv[r - 1] and rl[r] < 31) and not (r + 2 < len(ro) and nv[r + 1] and rl[r + 1] < 31)
        if lone_snp or nv[r] < 2:
            assert count == 0
            continue
        assert count == sum(1 for a in range(1, nv[r]) if 1 <= lens[a] <= 64) and first == seen
        for k in range(count):
            site, allele, len_a, len_0, site_order = (int(x) for x in win[first + k][:5])
            assert site == r and allele >= 1 and len_a == lens[allele] and len_0 == lens[0] and site_order == ro[r] + rl[r]
        seen += count
    assert seen == len(win)
    # the planes of a window: the linear reference, the allele, the linear reference behind the site's reference allele
    planes = c.hint_table(1)
    base_at = lambda p: sum(((int(planes[4 * (p >> 5) + b]) >> (p & 31)) & 1) << b for b in range(4))
    code = {"A": 1, "C": 2, "G": 4, "T": 8}
    dna = bytes(g["dna"]) if not isinstance(g["dna"], (bytes, str)) else g["dna"]
    dna = dna.decode() if isinstance(dna, bytes) else dna
    for w in (0, len(win) // 2, len(win) - 1):
        site, allele, len_a, len_0, site_order = (int(x) for x in win[w][:5])
        q0 = win_base + w * WIN_STRIDE
        for k in range(len_a):
            assert base_at(q0 + WIN_BEFORE + k) == code[dna[int(g["var_dna_off"][fv[site] + allele]) + k]]
        for k in range(1, 40):  # in front of the site and behind it: what the linear reference has there
            m = site_order - int(ro[0]) - k
            if m >= 0:
                assert base_at(q0 + WIN_BEFORE - k) == base_at(m)
            m = site_order - int(ro[0]) + len_0 + k - 1
            if m < n_main:
                assert base_at(q0 + WIN_BEFORE + len_a + k - 1) == base_at(m)
    snps = gtx.graph_from_records(*graph_of("snp1k", 60000, 10)[0:2], region_begin=1000000)
    assert len(gtx.Context(snps, device=-1).hint_table(5)) == 0


@pytest.mark.parametrize("kind", ["cfg3", "cluster", "indel", "snp25", "snp7"])
def test_both_builds_of_pass_0_write_the_same_records(kind, monkeypatch):
    """lean and dense build on every graph shape: the records are the same words, only who finishes a read differs"""
    ref, recs, codes, pos, g, aav = graph_of(kind, 50000, 2500, seed=3)
    seq, lens = harness.pack_ragged(list(codes))
    meta = harness.read_meta(lens, pos=pos)
    out, done = {}, {}
    for build in ("lean", "dense"):
        monkeypatch.setenv("GTX_HINT_BUILD", build)
        b = harness.EmuBackend(g)
        out[build] = b.align(seq, meta).reshape(2 * len(lens), -1)
        done[build] = b.hinted_done()
    external = ((out["lean"][:, 0] >> 16) & gtx.ST_EXTERNAL) != 0  # (arena offsets differ from call to call)
    differ = out["lean"] != out["dense"]
    differ[external, 2:] = False
    assert not differ.any(), np.nonzero(differ.any(1))[0][:5]
    assert done["dense"] >= done["lean"]


def test_dense_build_finishes_the_reads_of_a_cfg3_graph(monkeypatch):
    """SURVEY 8(d)'s cfg3 graph (a site every 100 bp, a tenth of them short indels with a SNP close by): the dense build of pass 0
    finishes 95 % of the reads -- reads that carry an indel allele through its window -- and every record equals the oracle's
    (with correct, missing, shifted and foreign hints); the lean build stays below 90 %"""
    ref, recs, codes, pos, g, aav = graph_of("cfg3", 120000, 6000, seed=1)
    o = Oracle(ref, recs, region_begin=1000000, add_all_variants=True)
    monkeypatch.setenv("GTX_HINT_BUILD", "dense")
    check_align(harness.EmuBackend(g), o, list(codes), pos=pos)
    dense = check_align.hinted_done
    monkeypatch.setenv("GTX_HINT_BUILD", "lean")
    b = harness.EmuBackend(g)
    seq, lens = harness.pack_ragged(list(codes))
    b.align(seq, harness.read_meta(lens, pos=pos))
    assert dense >= 0.95 * len(codes) and b.hinted_done() < 0.90 * len(codes), (dense, b.hinted_done())


def test_windows_can_be_switched_off(monkeypatch):
    """GTX_HINT_WINDOWS=0: no windows are listed, the tables end with the linear reference, the records stay the same"""
    ref, recs, codes, pos, g, aav = graph_of("cfg3", 50000, 1500, seed=5)
    seq, lens = harness.pack_ragged(list(codes))
    meta = harness.read_meta(lens, pos=pos)
    monkeypatch.setenv("GTX_HINT_BUILD", "dense")
    b = harness.EmuBackend(g)
    with_windows, done = b.align(seq, meta).copy(), b.hinted_done()
    monkeypatch.setenv("GTX_HINT_WINDOWS", "0")
    b0 = harness.EmuBackend(g)
    assert len(b0.ctx.hint_table(5)) == 0
    without = b0.align(seq, meta)
    assert b0.hinted_done() < done
    a, c = with_windows.reshape(2 * len(lens), -1), without.reshape(2 * len(lens), -1)
    external = ((a[:, 0] >> 16) & gtx.ST_EXTERNAL) != 0
    differ = a != c
    differ[external, 2:] = False
    assert not differ.any()
