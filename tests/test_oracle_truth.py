"""The oracle against ground truth that does not come from the reference's code: reads cut from a known haplotype of a
SNP-only graph must come back as ONE path over exactly the bases they were cut from, without mismatches, carrying at
every SNP site the allele of that haplotype.  (The reference has no active tests for seed chaining and graph walks --
test/typer/test_genotype_path.cpp is commented out -- so this is the independent check the oracle gets for them; the
coordinates are the reference's: 1-based contig positions, test/index/test_index.cpp:183-190.)"""
import numpy as np
import pytest

from graphtyper_amd import synth
from oracle_lib import Oracle

CODE = np.array([1, 2, 4, 8], np.uint8)


@pytest.mark.parametrize("every,read_len", [(1000, 150), (100, 150), (37, 150), (100, 101), (37, 63)])
def test_error_free_reads_come_back_as_their_haplotype(every, read_len):
    n_ref, region_begin, n_reads = 30000, 700000, 300
    rng = np.random.default_rng(every + read_len)
    ref = synth.make_reference(n_ref, seed=every)
    recs = synth.make_snp_records(ref, every, seed=3, region_begin=region_begin)
    pos = np.array([p - region_begin for p, _, _, _ in recs])
    alt = np.array(["ACGT".index(a[0]) for _, _, a, _ in recs], np.uint8)
    take = rng.random(len(recs)) < 0.5
    hap1 = ref.copy()
    hap1[pos[take]] = alt[take]
    haps = [ref, hap1]
    which = rng.integers(0, 2, size=n_reads)
    start = rng.integers(1, n_ref - read_len, size=n_reads)  # (position 0 would be contig position region_begin: 0-based edge)
    reads = [CODE[haps[h][s:s + read_len]] for h, s in zip(which, start)]
    o = Oracle(synth.bases_to_str(ref), recs, region_begin=region_begin)
    got = o.align(reads)
    for i, (fwd, _rev) in enumerate(got):
        assert len(fwd["paths"]) == 1, "read %d: %r" % (i, fwd)
        p = fwd["paths"][0]
        first = region_begin + int(start[i]) + 1  # 1-based contig position of the read's first base
        assert (p["start"], p["end"], p["rs"], p["re"], p["mm"]) == (first, first + read_len - 1, 0, read_len - 1, 0), (i, p)
        assert fwd["longest"] == read_len
        inside = np.nonzero((pos >= start[i]) & (pos < start[i] + read_len))[0]
        want = {int(region_begin + pos[k] + 1): (1 if (which[i] == 1 and take[k]) else 0) for k in inside}
        seen = {order: nums for order, nums in p["vars"]}
        assert set(seen) == set(want), (i, seen, want)
        for order, allele in want.items():
            assert seen[order] == (allele,), (i, order, seen[order], allele)
