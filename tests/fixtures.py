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
