"""Sites with more than 64 alleles (the reference allows MAX_NUMBER_OF_HAPLOTYPES = 2560, constants.hpp.in:23):
a 100-allele site and a merged cluster whose allele product passes 2559 (VarRecord::merge_one_path, graph.cpp:119-124).
The front passes keep 64-bit allele sets and hand on every task that meets an allele number >= 64; gtx_align_wide_kernel
(allele sets of GTX_WIDE_MASK_WORDS words, records with GTX_REC_WIDE) and gtx_score_wide_kernel finish them.  Here through
the emulation, in tests/test_gpu_configs.py on the device."""
import ctypes as C

import numpy as np

import harness
import scenarios
from graphtyper_amd import lib as gtx
from oracle_lib import Oracle
from test_emu_parity import check_align, run_stream


def wide_sites_case(Backend):
    rb = 20000
    ref, recs, codes, pos, _ = scenarios.wide_site_case(region_begin=rb)
    o = Oracle(ref, recs, region_begin=rb, add_all_variants=True)
    g = gtx.graph_from_records(ref, recs, region_begin=rb, add_all_variants=True)
    cnum = np.sort(g["ref_nvar"])
    assert cnum[-2] == 100 and cnum[-1] > 1000, cnum[-4:]
    b = Backend(g)
    assert b.ctx.total_tri > 500000  # the genotype triangle of the merged site alone
    check_align(b, o, list(codes), pos=pos)
    seq, lens = harness.pack_ragged(list(codes))
    rec_words = b.align(seq, harness.read_meta(lens, pos=pos))
    heads = rec_words.reshape(-1, harness.REC_WORDS)
    wide = (heads[:, 1] & gtx.REC_WIDE) != 0
    assert wide.sum() > 100 and (((heads[wide, 0] >> 16) & gtx.ST_EXTERNAL) != 0).all()
    big, tasks = b.big_records()
    parsed = gtx.parse_records(rec_words, len(codes), harness.REC_WORDS, b.ctx.hap_order, big)
    high = sum(1 for r in parsed for p in r[0]["paths"] for _, alleles in p["vars"] if alleles and max(alleles) >= 64)
    assert high > 100, "no path names an allele beyond the 64-bit sets"
    assert any(len(alleles) > 64 for r in parsed for p in r[0]["paths"] for _, alleles in p["vars"]), "no allele set with more than 64 members"
    rec = scenarios.stream_records(len(codes), pos, sample=np.arange(len(codes)) % 2)
    run_stream(b, o, codes, rec, n_samples=2)
    return b


def test_sites_with_more_than_64_alleles():
    b = wide_sites_case(harness.EmuBackend)
    b.L.emu_wide_pass_tasks.restype = C.c_uint64
    assert b.L.emu_wide_pass_tasks(C.c_void_p(b.h)) > 100
