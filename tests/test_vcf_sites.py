"""The sites one genotyping iteration hands to the next (vcf_merge_and_filter, src/typer/vcf_operations.cpp:
278-478): gtx_vcf_sites against the oracle's restatement (oracle/gto_vcf.hpp: sites) -- inside run_stream for every stream
scenario of the suite -- and here what closes the loop of genotype() (src/utilities/genotype.cpp:520-575): iteration 1's sites
text -> records with GT_ID / GT_ANTI_HAPLOTYPE -> the graph and index of iteration 2 (add_all_variants) -> the same reads aligned,
scored, called and written again, product and oracle side by side at every stage; plus what follows from the reference's text
alone (numbering, order of the tags, who is an anti allele of whom)."""
import numpy as np
import pytest

import harness
import scenarios
from graphtyper_amd import lib as gtx
from oracle_lib import Oracle
from test_emu_parity import run_stream


def parse_sites(text):
    lines = text.decode().split("\n")
    assert lines[0] == "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO" and lines[-1] == ""
    out = []
    for l in lines[1:-1]:
        f = l.split("\t")
        assert len(f) == 8 and f[5] == "0" and f[6] == "." and "," not in f[4]
        info = dict(kv.split("=", 1) for kv in f[7].split(";"))
        assert list(info) == sorted(info) and "GT_ID" in info and set(info) <= {"GT_ID", "GT_ANTI_HAPLOTYPE", "GT_HAPLOTYPE"}
        out.append(dict(chrom=f[0], pos=int(f[1]), id=f[2], ref=f[3], alt=f[4], info=info, info_text=f[7]))
    return out


def as_records(sites):
    """the records construct_graph reads back: (pos0, ref, [alt], INFO)"""
    return [(s["pos"] - 1, s["ref"], [s["alt"]], s["info_text"]) for s in sites]


def first_iteration(kind, Backend, rb=310000):
    if kind == "snp25":
        ref, recs, codes, rec = scenarios.paired_case(kind, n_ref=16000, n_pairs=2600, region_begin=rb, n_samples=3)
        kw, n_samples = {}, 3
    else:
        ref, recs, codes, pos = scenarios.synthetic_case(kind, n_ref=9000, n_reads=5000, region_begin=rb, seed=5)
        kw = dict(add_all_variants=True) if kind == "cluster" else {}
        order = np.argsort(pos, kind="stable")
        rec = scenarios.stream_records(len(codes), pos, sample=np.arange(len(codes)) % 3)[order]
        codes, n_samples = codes[order], 3
    o = Oracle(ref, recs, region_begin=rb, **kw)
    b = Backend(gtx.graph_from_records(ref, recs, region_begin=rb, **kw))
    run_stream(b, o, codes, rec, n_samples=n_samples)  # (compares the sites text with the oracle's)
    return ref, recs, codes, rec, n_samples, b, run_stream.sites


def check_sites(sites, ctx):
    """what vcf_operations.cpp:311-470 implies for the text whatever the reads were"""
    ids = [int(s["info"]["GT_ID"]) for s in sites]
    assert ids == sorted(ids) and len(set(ids)) == len(ids)  # alleles are numbered in file order, from 1, dropped ones counted
    assert ids[0] >= 1
    pos = [s["pos"] for s in sites]
    assert pos == sorted(pos)
    by_id = {int(s["info"]["GT_ID"]): s for s in sites}
    for s in sites:
        me = int(s["info"]["GT_ID"])
        assert s["id"].startswith("%s:%d:" % (s["chrom"], s["pos"]))
        for key in ("GT_ANTI_HAPLOTYPE", "GT_HAPLOTYPE"):
            for other in (int(x) for x in s["info"].get(key, "").split(",") if x):
                assert other != me
                if key == "GT_ANTI_HAPLOTYPE" and other in by_id and by_id[other]["pos"] == s["pos"] and by_id[other]["ref"] == s["ref"]:
                    assert other > me  # the later kept alleles of the site itself come first, ascending


@pytest.mark.parametrize("kind", ["snp25", "indel", "cluster"])
def test_second_iteration_from_the_sites_of_the_first(kind, tmp_path):
    rb = 310000
    ref, recs, codes, rec, n_samples, b, text = first_iteration(kind, harness.EmuBackend, rb)
    sites = parse_sites(text)
    check_sites(sites, b.ctx)
    n_alts = b.ctx.total_allele - b.ctx.n_hap  # alternative alleles of the graph's sites (merged sites: the combinations)
    assert 0 < len(sites) <= n_alts
    assert int(sites[-1]["info"]["GT_ID"]) <= n_alts
    if kind == "snp25":  # (paired reads over sites 25 bp apart, three samples with their own haplotypes: flags both ways)
        assert any("GT_HAPLOTYPE" in s["info"] for s in sites) and any("GT_ANTI_HAPLOTYPE" in s["info"] for s in sites)
    if kind == "cluster":  # merged sites: the kept alleles of one site exclude each other
        assert any("GT_ANTI_HAPLOTYPE" in s["info"] for s in sites)
    # ---- iteration 2: the graph of the kept alleles (genotype.cpp:524-526: add_all_variants), same reads
    recs2 = as_records(sites)
    o2 = Oracle(ref, recs2, region_begin=rb, add_all_variants=True)
    g2 = gtx.graph_from_records(ref, recs2, region_begin=rb, add_all_variants=True)
    b2 = harness.EmuBackend(g2)
    run_stream(b2, o2, codes, rec, n_samples=n_samples)
    sites2 = parse_sites(run_stream.sites)
    check_sites(sites2, b2.ctx)
    assert len(sites2) <= len(sites)
    # ---- the same graph through the files genotype() writes: header (genotypes dropped) + records, a FASTA beside it
    fa, vcf = tmp_path / "ref.fa", tmp_path / "final.vcf"
    fa.write_text(">chrT\n" + "N" * rb + ref + "\n")
    head = gtx.vcf_header("20260928", "2.7.7", [("chrT", rb + len(ref))], [], drop_genotypes=True)
    vcf.write_bytes(head + text[text.index(b"\n") + 1:])
    gf, span = gtx.graph_from_files(fa, vcf, "chrT:%d-%d" % (rb + 1, rb + len(ref)), add_all_variants=True)
    compared = 0
    for name in g2:
        if isinstance(g2[name], np.ndarray):
            assert g2[name].shape == gf[name].shape and np.array_equal(g2[name], gf[name]), name
            compared += 1
    assert compared >= 8


def test_sites_of_an_unobserved_region_are_empty():
    """no read, no allele anybody's reads reached: generate_infos drops every alternative allele (variant.cpp:1048-1052)"""
    ref, recs, codes, rec = scenarios.paired_case("snp100", n_ref=3000, n_pairs=10, region_begin=1000)
    c = gtx.Context(gtx.graph_from_records(ref, recs, region_begin=1000), device=-1)
    z = lambda n, t: np.zeros(max(1, n), t)
    text = c.vcf_sites("chr1", 0, z(0, np.uint32), z(c.n_hap + 2 * c.total_allele, np.uint64), z(c.n_hap + 6 * c.total_allele, np.uint32),
                       z(0, np.uint8), z(0, gtx.SAMPLE_CALL), np.zeros((0, 5), np.int64))
    assert text == b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    sv = gtx.Context(gtx.graph_from_records(ref, recs, region_begin=1000), device=-1, is_sv_graph=True)
    with pytest.raises(gtx.GtxError):
        sv.vcf_sites("chr1", 0, z(0, np.uint32), z(c.n_hap + 2 * c.total_allele, np.uint64), z(c.n_hap + 6 * c.total_allele, np.uint32),
                     z(0, np.uint8), z(0, gtx.SAMPLE_CALL), np.zeros((0, 5), np.int64))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["snp25", "cluster"])
def test_second_iteration_on_the_device(kind):
    """the same loop through the C ABI on the GPU: iteration 1's sites (== the oracle's) make iteration 2's graph and index, built on
    the device; its alignments, scores, calls, VCF text and sites equal the oracle's of that graph"""
    import torch  # (before libgtx is loaded: one HIP runtime in the process, torch's)
    assert torch.cuda.is_available()
    rb = 310000
    ref, recs, codes, rec, n_samples, b, text = first_iteration(kind, harness.GpuBackend, rb)
    sites = parse_sites(text)
    check_sites(sites, b.ctx)
    recs2 = as_records(sites)
    o2 = Oracle(ref, recs2, region_begin=rb, add_all_variants=True)
    b2 = harness.GpuBackend(gtx.graph_from_records(ref, recs2, region_begin=rb, add_all_variants=True))
    run_stream(b2, o2, codes, rec, n_samples=n_samples)
    assert 0 < len(parse_sites(run_stream.sites)) <= len(sites)
