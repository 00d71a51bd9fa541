"""Loads the reference's own test data files (copied verbatim as data into tests/golden/)."""
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def read_fasta(path=os.path.join(GOLDEN, "index_test.fa")):
    seqs, name = {}, None
    for line in open(path):
        line = line.strip()
        if line.startswith(">"):
            name = line[1:].split()[0]
            seqs[name] = ""
        elif name:
            seqs[name] += line
    return seqs


def read_vcf(path=os.path.join(GOLDEN, "index_test.vcf")):
    """-> {chrom: [(pos0, ref, [alts], info)]}; symbolic (<...>) alleles are kept as text."""
    recs = {}
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        f = line.rstrip("\n").split("\t")
        recs.setdefault(f[0], []).append((int(f[1]) - 1, f[3], f[4].split(","), f[7]))
    return recs


def contig(chrom):
    """(reference sequence, records) of one index_test contig"""
    return read_fasta()[chrom], read_vcf().get(chrom, [])


EXTRA_SEQUENCE_LENGTH = 152  # constructor.cpp:1437


def sv_contig(chrom):
    """(reference sequence, records) of an index_test contig whose VCF lines carry symbolic SV alleles: the records the
    reference's constructor synthesises from them, restated here for the ORACLE's input (the product has its own
    implementation in gtx_files.cpp).  Only deletions: add_sv_deletion, constructor.cpp:478-514 -- ref = the base at the
    position, alt = that base + SEQ / SVINSSEQ + the reference behind the deleted stretch, up to 153 characters, + the SV
    tag of append_sv_tag_to_node (constructor.cpp:155-161); INFO parsing and the size defaults: constructor.cpp:1282-1349"""
    ref, recs = contig(chrom)
    out, n_sv = [], 0
    for pos0, _ref_allele, alts, info in recs:
        assert len(alts) == 1 and alts[0].startswith("<DEL"), "only SV deletions are restated here"
        kv = dict(f.split("=", 1) for f in info.split(";") if "=" in f)
        assert kv["SVTYPE"] in ("DEL", "DEL:ME:ALU")
        seq = kv.get("SEQ", "")
        ins_seq = kv.get("SVINSSEQ", "")
        size, length = int(kv.get("SVSIZE", 0)), abs(int(kv.get("SVLEN", 0)))
        if length == 0:
            length = size or len(seq) or len(ins_seq)
        if size == 0:
            size = length
        base = ref[pos0]
        alt = base
        if seq and seq[0] != ".":
            alt += seq
        elif ins_seq and ins_seq[0] != ".":
            alt += ins_seq
        if len(alt) < EXTRA_SEQUENCE_LENGTH + 1:
            begin = pos0 + len(seq) + size + 1
            alt += ref[begin:begin + EXTRA_SEQUENCE_LENGTH + 1 - len(alt)]  # (seqan's readRegion clips at the contig's end)
        alt += "<SV:%07d>" % n_sv
        n_sv += 1
        out.append((pos0, base, [alt], "SV=1"))
    return ref, out
