"""ctypes binding of the CPU oracle (oracle/libgto.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
INVALID_ID = 0xFFFFFFFF
SPECIAL_START = 0xD0000000

_CODE = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def encode(seq):
    """ASCII IUPAC -> 4-bit BAM/IUPAC codes (A=1 C=2 G=4 T=8 N=15)."""
    return np.frombuffer(bytes(_CODE.get(c, 15) for c in seq), dtype=np.uint8).copy()


def to_uint64(kmer):
    d = 0
    for c in kmer:
        d = (d << 2) | "ACGT".index(c)
    return d


def to_dna_str(d, k=32):
    return "".join("ACGT"[(d >> (2 * i)) & 3] for i in range(k - 1, -1, -1))


def build():
    if os.environ.get("GTO_LIB"):  # (tests/oracle_mutants/run_audit.py: a deliberately broken oracle, to see the tests notice)
        return os.environ["GTO_LIB"]
    src = [os.path.join(ORACLE_DIR, f) for f in ("gto.hpp", "gto_capi.cpp", "gto_vcf.hpp", "gto_sv.hpp", "gto_discovery.hpp", "gto_shrink.hpp")]
    so = os.path.join(ORACLE_DIR, "libgto.so")
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src if os.path.exists(s)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.gto_last_error.restype = C.c_char_p
        L.gto_new.restype = C.c_void_p
        L.gto_new.argtypes = [C.c_char_p, C.c_long, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int]
        L.gto_free.argtypes = [C.c_void_p]
        L.gto_genotyper_new.restype = C.c_void_p
        L.gto_genotyper_new.argtypes = [C.c_void_p, C.c_long, C.c_long]
        L.gto_genotyper_free.argtypes = [C.c_void_p]
        for f in ("gto_index_num_keys", "gto_index_num_labels", "gto_index_get", "gto_query_read", "gto_align",
                  "gto_scores_dump", "gto_to_uint64_vec", "gto_get_num_kmers", "gto_ith_kmer_offset", "gto_all_ref",
                  "gto_genotyper_num_haplotypes"):
            getattr(L, f).restype = C.c_long
        L.gto_get_num_kmers.argtypes = [C.c_long]
        L.gto_ith_kmer_offset.argtypes = [C.c_long, C.c_long]
        L.gto_path_merge_two_ref_labels.argtypes = [C.c_uint32] * 8 + [C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def records_text(records):
    """records: iterable of (pos0, ref, [alts], info)"""
    return "\n".join("%d %s %s %s" % (p, r, ",".join(a), i or ".") for p, r, a, i in records)


def pack_reads(reads):
    """list of uint8 code arrays -> (concatenated codes, offsets[n+1])"""
    offs = np.zeros(len(reads) + 1, dtype=np.uint32)
    for i, r in enumerate(reads):
        offs[i + 1] = offs[i] + len(r)
    codes = np.concatenate(reads).astype(np.uint8) if reads else np.zeros(0, np.uint8)
    return np.ascontiguousarray(codes), offs


class Oracle:
    def __init__(self, reference, records, region_begin=0, is_sv_graph=False, hq_reads=False, force_both=False,
                 max_index_labels=75, add_all_variants=False, extend_prefix=False):
        L = lib()
        self.reference, self.region_begin = reference, region_begin  # (what the VCF writers read through the graph: normalisation)
        self.h = L.gto_new(reference.encode(), region_begin, records_text(records).encode(), int(is_sv_graph),
                           int(hq_reads), int(force_both), max_index_labels, int(add_all_variants), int(extend_prefix))
        if not self.h:
            raise RuntimeError(L.gto_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None):
            lib().gto_free(C.c_void_p(self.h))
            self.h = None

    # ---- graph
    def graph(self):
        L = lib()
        cnt = (C.c_long * 5)()
        L.gto_graph_counts(C.c_void_p(self.h), cnt)
        nr, nv, ns, nd, ne = [int(x) for x in cnt]
        g = dict(ref_order=np.zeros(nr, np.uint32), ref_len=np.zeros(nr, np.uint32), ref_nvar=np.zeros(nr, np.uint32),
                 ref_first_var=np.zeros(nr, np.uint32), var_order=np.zeros(nv, np.uint32), var_len=np.zeros(nv, np.uint32),
                 var_out_ref=np.zeros(nv, np.uint32), dna=np.zeros(nd, np.uint8), ref_reach_poses=np.zeros(ns, np.uint32),
                 actual_poses=np.zeros(ns, np.uint32), events=np.zeros(2 * nv + ne, np.int64))
        L.gto_graph_dump(C.c_void_p(self.h), _p(g["ref_order"]), _p(g["ref_len"]), _p(g["ref_nvar"]), _p(g["ref_first_var"]),
                         _p(g["var_order"]), _p(g["var_len"]), _p(g["var_out_ref"]), _p(g["dna"]), _p(g["ref_reach_poses"]),
                         _p(g["actual_poses"]), _p(g["events"]))
        return g

    def all_ref(self):
        L = lib()
        n = L.gto_all_ref(C.c_void_p(self.h), None, C.c_long(0))
        buf = C.create_string_buffer(n)
        L.gto_all_ref(C.c_void_p(self.h), buf, C.c_long(n))
        return buf.raw.decode()

    # ---- index
    def index_get(self, key):
        L = lib()
        if isinstance(key, str):
            key = to_uint64(key)
        out = np.zeros(3 * 4096, np.uint32)
        n = L.gto_index_get(C.c_void_p(self.h), C.c_uint64(key), _p(out), C.c_long(4096))
        return [tuple(int(x) for x in out[3 * i:3 * i + 3]) for i in range(n)]

    def index_check(self):
        return bool(lib().gto_index_check(C.c_void_p(self.h)))

    def index_dump(self):
        L = lib()
        nk = L.gto_index_num_keys(C.c_void_p(self.h))
        nl = L.gto_index_num_labels(C.c_void_p(self.h))
        keys = np.zeros(nk, np.uint64)
        counts = np.zeros(nk, np.uint32)
        labels = np.zeros(3 * nl, np.uint32)
        L.gto_index_dump(C.c_void_p(self.h), _p(keys), _p(counts), _p(labels))
        return keys, counts, labels.reshape(-1, 3)

    def query_read(self, codes):
        L = lib()
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        n = L.gto_query_read(C.c_void_p(self.h), _p(codes), C.c_long(len(codes)), None, C.c_long(0))
        out = np.zeros(n, np.uint32)
        L.gto_query_read(C.c_void_p(self.h), _p(codes), C.c_long(len(codes)), _p(out), C.c_long(n))
        return out

    # ---- align
    def align(self, reads, flags=None, tid=None, mtid=None, isize=None):
        L = lib()
        codes, offs = pack_reads(reads)
        n = len(reads)
        flags = None if flags is None else np.ascontiguousarray(flags, np.uint16)
        tid = None if tid is None else np.ascontiguousarray(tid, np.int32)
        mtid = None if mtid is None else np.ascontiguousarray(mtid, np.int32)
        isize = None if isize is None else np.ascontiguousarray(isize, np.int64)
        cap = 1 << 16
        while True:
            out = np.zeros(cap, np.uint32)
            w = L.gto_align(C.c_void_p(self.h), C.c_long(n), _p(codes), _p(offs), _p(flags), _p(tid), _p(mtid), _p(isize), _p(out),
                            C.c_long(cap))
            if w < 0:
                raise RuntimeError(L.gto_last_error().decode())
            if w <= cap:
                return parse_path_stream(out[:w], n)
            cap = int(w)

    def genotyper(self, n_samples=1, n_rg=1):
        return OracleGenotyper(self, n_samples, n_rg)


def parse_path_stream(s, n_reads):
    """-> list over reads of (fwd, rev); each = dict(longest=, paths=[dict(start,end,rs,re,mm,vars=[(order,(nums...))])])"""
    out = []
    i = 0
    for _ in range(n_reads):
        pair = []
        for _o in range(2):
            npaths, longest = int(s[i]), int(s[i + 1])
            i += 2
            paths = []
            for _p_ in range(npaths):
                st, en, rs, re_, mm, nv = [int(x) for x in s[i:i + 6]]
                i += 6
                vs = []
                for _v in range(nv):
                    order, cnt = int(s[i]), int(s[i + 1])
                    i += 2
                    vs.append((order, tuple(int(x) for x in s[i:i + cnt])))
                    i += cnt
                paths.append(dict(start=st, end=en, rs=rs, re=re_, mm=mm, vars=vs))
            pair.append(dict(longest=longest, paths=paths))
        out.append(tuple(pair))
    assert i == len(s)
    return out


class OracleGenotyper:
    def __init__(self, oracle, n_samples, n_rg):
        self.o = oracle
        self.g = lib().gto_genotyper_new(C.c_void_p(oracle.h), n_samples, n_rg)

    def __del__(self):
        if getattr(self, "g", None):
            lib().gto_genotyper_free(C.c_void_p(self.g))
            self.g = None

    def push(self, reads, flags=None, tid=None, mtid=None, pos=None, isize=None, mapq=None, score_diff=None, name=None,
             sample=None, rg=None, mpos=None, n_cigar=None, cigar_front=None, cigar_back=None, packed=None):
        """packed = (codes, offsets) as pack_reads returns them: skips the Python-side packing (timed runs)"""
        L = lib()
        codes, offs = pack_reads(reads) if packed is None else packed
        if packed is not None:
            reads = range(len(offs) - 1)

        def arr(a, t):
            return None if a is None else np.ascontiguousarray(a, t)

        a = [arr(flags, np.uint16), arr(tid, np.int32), arr(mtid, np.int32), arr(pos, np.int64), arr(isize, np.int64),
             arr(mapq, np.uint8), arr(score_diff, np.uint8), arr(name, np.uint64), arr(sample, np.int32), arr(rg, np.int32),
             arr(mpos, np.int64), arr(n_cigar, np.uint32), arr(cigar_front, np.uint32), arr(cigar_back, np.uint32)]
        rc = L.gto_genotyper_push_ex(C.c_void_p(self.g), C.c_long(len(reads)), _p(codes), _p(offs), *[_p(x) for x in a])
        if rc != 0:
            raise RuntimeError(L.gto_last_error().decode())

    def set_coverage(self, avg_cov_by_readlen, no_filter_on_coverage=False):
        a = np.ascontiguousarray(avg_cov_by_readlen, np.float64)
        lib().gto_genotyper_set_coverage(C.c_void_p(self.g), _p(a), C.c_long(len(a)), C.c_int(int(no_filter_on_coverage)))

    def finish(self):
        lib().gto_genotyper_finish(C.c_void_p(self.g))

    def scores(self):
        L = lib()
        n = L.gto_scores_dump(C.c_void_p(self.g), None, C.c_long(0))
        out = np.zeros(n, np.uint32)
        L.gto_scores_dump(C.c_void_p(self.g), _p(out), C.c_long(n))
        return out

    def calls(self):
        """word stream of the per haplotype / sample SampleCalls (gto_calls_dump)"""
        L = lib()
        L.gto_calls_dump.restype = C.c_long
        n = L.gto_calls_dump(C.c_void_p(self.g), None, C.c_long(0))
        out = np.zeros(max(n, 1), np.uint32)
        L.gto_calls_dump(C.c_void_p(self.g), _p(out), C.c_long(n))
        return out[:n]

    def phase_flags(self):
        """rows (hap1, allele1, hap2, allele2, flags) of the `ph` map (gto_phase_flags)"""
        L = lib()
        L.gto_phase_flags.restype = C.c_long
        n = L.gto_phase_flags(C.c_void_p(self.g), None, C.c_long(0))
        out = np.zeros((max(n, 1), 5), np.int32)
        L.gto_phase_flags(C.c_void_p(self.g), _p(out), C.c_long(n))
        return out[:n].astype(np.int64)

    def reference_depth(self, sample):
        """ReferenceDepth::depths[sample] (SV calling only; empty otherwise)"""
        L = lib()
        L.gto_reference_depth.restype = C.c_long
        n = L.gto_reference_depth(C.c_void_p(self.g), C.c_long(sample), None, C.c_long(0))
        out = np.zeros(max(n, 1), np.uint16)
        L.gto_reference_depth(C.c_void_p(self.g), C.c_long(sample), _p(out), C.c_long(n))
        return out[:n]

    def vcf_records(self, contig, sample_names, region_begin=0, region_end=0xFFFFFFFF, filter_zero_qual=False, suffix_id=None):
        """oracle/gto_vcf.hpp: the VCF records (column line first) of the genotyper's variant sites as bytes"""
        L = lib()
        L.gto_vcf_records.restype = C.c_long
        args = (C.c_void_p(self.g), contig.encode(), "\n".join(sample_names).encode(), C.c_uint32(region_begin), C.c_uint32(region_end),
                C.c_int(int(filter_zero_qual)), suffix_id.encode() if suffix_id else None)
        n = L.gto_vcf_records(*args, None, C.c_long(0))
        buf = C.create_string_buffer(n + 1)
        L.gto_vcf_records(*args, buf, C.c_long(n))
        return buf.raw[:n]

    def vcf_records_final(self, contig, sample_names, reference, first_pos, region_begin=0, region_end=0xFFFFFFFF, filter_zero_qual=False,
                          no_variant_overlapping=False):
        """oracle/gto_sv.hpp records_final(): vcf_merge_and_break with the variants broken down -- the file genotype() writes"""
        L = lib()
        L.gto_vcf_records_final.restype = C.c_long
        args = (C.c_void_p(self.g), contig.encode(), "\n".join(sample_names).encode(), C.c_uint32(region_begin), C.c_uint32(region_end),
                C.c_int(int(filter_zero_qual)), reference.encode(), C.c_uint32(first_pos), C.c_int(int(no_variant_overlapping)))
        n = L.gto_vcf_records_final(*args, None, C.c_long(0))
        if n < 0:
            raise RuntimeError(L.gto_last_error().decode())
        buf = C.create_string_buffer(n + 1)
        L.gto_vcf_records_final(*args, buf, C.c_long(n))
        return buf.raw[:n]

    def vcf_sites(self, contig):
        """oracle/gto_vcf.hpp sites(): vcf_merge_and_filter's records (column line first) with the genotyper's own `ph`"""
        L = lib()
        L.gto_vcf_sites.restype = C.c_long
        n = L.gto_vcf_sites(C.c_void_p(self.g), contig.encode(), None, C.c_long(0))
        buf = C.create_string_buffer(n + 1)
        L.gto_vcf_sites(C.c_void_p(self.g), contig.encode(), buf, C.c_long(n))
        return buf.raw[:n]

    def vcf_records_sv(self, contig, sample_names, sv_table, reference, first_pos, region_begin=0, region_end=0xFFFFFFFF):
        """oracle/gto_sv.hpp: the VCF records of an SV graph's calls (reformat_sv_vcf_records + the merge of genotype_sv) as bytes;
        reference / first_pos: the region's reference sequence and the 1-based position of its first base"""
        L = lib()
        L.gto_vcf_records_sv.restype = C.c_long
        args = (C.c_void_p(self.g), contig.encode(), "\n".join(sample_names).encode(), C.c_uint32(region_begin), C.c_uint32(region_end),
                sv_table.encode(), reference.encode(), C.c_uint32(first_pos))
        n = L.gto_vcf_records_sv(*args, None, C.c_long(0))
        if n < 0:
            raise RuntimeError(L.gto_last_error().decode())
        buf = C.create_string_buffer(n + 1)
        L.gto_vcf_records_sv(*args, buf, C.c_long(n))
        return buf.raw[:n]

    def merge(self, other):
        """self += other (Genotyper::merge_from: sums of the accumulated state; refused at the saturation guard)"""
        L = lib()
        L.gto_genotyper_merge.argtypes = [C.c_void_p, C.c_void_p]
        if L.gto_genotyper_merge(C.c_void_p(self.g), C.c_void_p(other.g)) != 0:
            raise RuntimeError(L.gto_last_error().decode())

    def counts(self):
        c = (C.c_long * 3)()
        lib().gto_genotyper_counts(C.c_void_p(self.g), c)
        return dict(records=int(c[0]), duplicated=int(c[1]), parked=int(c[2]))


def sharded_genotyper(oracle, codes, pos, n_samples=1, samples=None, threads=None, mapq=None):
    """All reads of a large UNPAIRED, position-sorted read set through the oracle on several host threads: contiguous
    shards, one Genotyper each (the C++ calls release the GIL), summed into the first (Genotyper::merge_from).  codes:
    [n, L] uint8 BAM codes.  A shard boundary only loses the reuse of the previous record's paths for an exact duplicate,
    which is an optimisation of the reference, not part of the result."""
    import threading
    n, L = codes.shape
    threads = max(1, min(threads or (os.cpu_count() or 1), 256, (n + 9999) // 10000))
    cuts = [n * k // threads for k in range(threads + 1)]
    genos = [oracle.genotyper(n_samples, 1) for _ in range(threads)]
    errors = []

    def work(k):
        a, b = cuts[k], cuts[k + 1]
        try:
            flat = np.ascontiguousarray(codes[a:b]).reshape(-1)
            offs = (np.arange(b - a + 1, dtype=np.uint64) * L).astype(np.uint32)
            genos[k].push(None, pos=np.ascontiguousarray(pos[a:b], np.int64), packed=(flat, offs),
                          sample=None if samples is None else samples[a:b], mapq=None if mapq is None else mapq[a:b])
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    team = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    for t in team:
        t.start()
    for t in team:
        t.join()
    if errors:
        raise errors[0]
    for g in genos[1:]:
        genos[0].merge(g)
    return genos[0], threads
