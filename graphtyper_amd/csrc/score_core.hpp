// score_core.hpp -- per read / per mate pair work of the scoring kernel (one thread per score item).
//
// Restates, over the alignment records in HBM and integer atomics into flat accumulators:
//   update_unpaired_read_paths / update_paths / get_better_paths      src/typer/alignment.cpp:365-620
//   compare_pair_of_genotype_paths (single and pair)                   src/typer/genotype_paths.cpp:943-1169
//   are_genotype_paths_good, push_to_haplotype_scores, update_haplotype_scores_geno  src/typer/vcf_writer.cpp:28-250,503-676
//   Haplotype::add_coverage / *_to_stats / explain_to_score / coverage_to_gts        src/graph/haplotype.cpp:180-585
// Every effect of a read on shared state is an integer addition, so the order in which items are scored does not
// matter below the saturation guard of explain_to_score (checked by gtx_scores_finalize).
#pragma once
#include "graph_dev.hpp"

namespace gtx
{
constexpr uint16_t F_PAIRED = 1, F_PROPER_PAIR = 2, F_UNMAPPED = 4, F_SEQ_REVERSED = 16, F_FIRST_IN_PAIR = 64, F_MAPQ_BAD = 4096;
constexpr uint32_t NO_COVERAGE = 0xFFFFu, MULTI_ALT_COVERAGE = 0xFFFEu, MULTI_REF_COVERAGE = 0xFFFDu; // haplotype.hpp:86-88
#ifndef GTX_SCORE_MAX_HAPS
#define GTX_SCORE_MAX_HAPS 8
#endif
constexpr uint32_t SCORE_MAX_HAPS = GTX_SCORE_MAX_HAPS; // distinct variant sites one read can touch in the main scoring pass (per-thread tables)
constexpr uint32_t SCORE_MAX_HAPS_BIG = 1024; // ... in the second pass (tables in HBM)
constexpr uint32_t SCORE_MAX_HAPS_WIDE = 128; // ... in the second pass of a graph with a site of more than 64 alleles (wide sets)

struct ScoreParams
{
  uint32_t is_sv_graph, hq_reads, is_segment_calling, pad;
};

struct RecPath
{
  uint32_t start, end, rs, re, mism, nvar;
  uint32_t const * vars; // nvar * (site, mask words): 2 mask words, GTX_WIDE_MASK_WORDS in a record with GTX_REC_WIDE
  uint32_t stride;       // words per site
  GTX_DEV uint32_t const * mask(uint32_t k) const { return vars + stride * k + 1; }
  GTX_DEV uint32_t site(uint32_t k) const { return vars[stride * k]; }
};

struct Geno // one GenotypePaths as seen by the scorer
{
  uint32_t const * rec;
  uint32_t const * body; // path words: behind the header, or in the big-record arena (GTX_ST_EXTERNAL)
  uint32_t n_paths, longest, read_len;
  uint32_t flags, mapq, score_diff;
  bool proper_pair; // ml_insert_size != INSERT_SIZE_WHEN_NOT_PROPER_PAIR
  bool has_var;     // some path carries a variant site (else the read cannot add anything to the accumulators)
  bool wide;        // GTX_REC_WIDE: allele sets of GTX_WIDE_MASK_WORDS words
};

template <class Acc>
GTX_DEV Geno geno_of(uint32_t const * records, uint32_t rec_words, Acc const & acc, uint32_t align_index, uint32_t orient)
{
  Geno g;
  g.rec = records + (static_cast<uint64_t>(align_index) * 2 + orient) * rec_words;
  if (orient == 0 && acc.compact && (acc.compact_flags[2ull * align_index] & GTX_TASK_COMPACT))
    g.rec = acc.compact + static_cast<uint64_t>(align_index) * GTX_COMPACT_WORDS; // (a record of at most that many words: never external)
  if (g.rec == acc.staged_from)
    g.rec = acc.staged_copy; // (the scoring kernel has fetched this record in one go: every word the parser looks at is in the copy)
  g.body = ((g.rec[0] >> 16) & GTX_ST_EXTERNAL) ? acc.big_records + g.rec[2] : g.rec + 2;
  g.n_paths = g.rec[0] & 0xFFFFu;
  g.longest = g.rec[1] & 0xFFFFu;
  g.read_len = (g.rec[1] >> 16) & 0x3FFFu;
  g.has_var = (g.rec[1] & GTX_REC_HAS_VARIANTS) != 0;
  g.wide = (g.rec[1] & GTX_REC_WIDE) != 0;
  g.flags = 0;
  g.mapq = 255;
  g.score_diff = 0;
  g.proper_pair = false;
  return g;
}

// the reverse orientation of a read that was aligned forward only: an empty result, nothing is fetched
GTX_DEV Geno empty_orientation(Geno const & fwd)
{
  Geno g = fwd;
  g.n_paths = 0;
  g.longest = 0;
  g.has_var = false;
  g.wide = false;
  return g;
}

GTX_DEV uint32_t const * path_at(uint32_t const * w, RecPath & p, bool wide = false) // returns the position behind the path
{
  p.start = w[0];
  p.end = w[1];
  p.rs = w[2] & 0xFFFFu;
  p.re = w[2] >> 16;
  p.mism = w[3] & 0xFFFFu;
  p.nvar = w[3] >> 16;
  p.vars = w + 4;
  p.stride = wide ? 1u + GTX_WIDE_MASK_WORDS : 3u;
  return w + 4 + p.stride * p.nvar;
}

GTX_DEV uint32_t first_mismatches(Geno const & g) // paths[0].mismatches
{
  return g.body[3] & 0xFFFFu;
}

GTX_DEV uint32_t alternative_call_count(Geno const & g) // genotype_paths.cpp:1040-1053
{
  uint32_t c = 0;
  uint32_t const * w = g.body;
  for (uint32_t i = 0; i < g.n_paths; ++i)
  {
    RecPath p;
    w = path_at(w, p, g.wide);
    for (uint32_t k = 0; k < p.nvar; ++k)
      c += (p.mask(k)[0] & 1u) == 0u;
  }
  return c;
}

// genotype_paths.cpp:943-974
GTX_DEV int compare_single(Geno const & g1, Geno const & g2)
{
  uint32_t const t1 = g1.longest, t2 = g2.longest, MIN = 94;
  if (t1 > t2 && t1 > MIN)
    return 1;
  if (t2 > t1 && t2 > MIN)
    return 2;
  if (t1 == t2 && t1 > MIN)
    return first_mismatches(g2) < first_mismatches(g1) ? 2 : 1;
  return 0;
}

// genotype_paths.cpp:976-1169
GTX_DEV int compare_pairs(Geno const & a1, Geno const & a2, Geno const & b1, Geno const & b2)
{
  uint32_t const T11 = a1.n_paths ? a1.longest : 0, T12 = a2.n_paths ? a2.longest : 0;
  uint32_t const T21 = b1.n_paths ? b1.longest : 0, T22 = b2.n_paths ? b2.longest : 0;
  uint32_t const M1 = T11 > T12 ? T11 : T12, M2 = T21 > T22 ? T21 : T22;
  uint32_t const P1 = a1.read_len, P2 = a2.read_len, MIN = 94;
  bool const perfect1 = T11 >= P1 && T12 >= P2, perfect2 = T21 >= P1 && T22 >= P2;
  if (perfect1 || perfect2)
  {
    if (perfect1 && perfect2)
    {
      uint32_t const mm1 = first_mismatches(a1) + first_mismatches(a2), mm2 = first_mismatches(b1) + first_mismatches(b2);
      if (mm1 != mm2)
        return mm1 < mm2 ? 1 : 2;
      uint32_t const n1 = a1.n_paths + a2.n_paths, n2 = b1.n_paths + b2.n_paths;
      if (n1 != n2)
        return n1 < n2 ? 1 : 2;
      return alternative_call_count(a1) + alternative_call_count(a2) >= alternative_call_count(b1) + alternative_call_count(b2) ? 1 : 2;
    }
    return perfect1 ? 1 : 2;
  }
  if (M2 >= MIN && M2 > M1)
    return 2;
  if (M1 >= MIN && M1 > M2)
    return 1;
  if (M1 >= MIN && M2 >= MIN)
  {
    uint32_t mm1 = 10, mm2 = 10;
    if (T11 == M1 && first_mismatches(a1) < mm1)
      mm1 = first_mismatches(a1);
    if (T12 == M1 && first_mismatches(a2) < mm1)
      mm1 = first_mismatches(a2);
    if (T21 == M2 && first_mismatches(b1) < mm2)
      mm2 = first_mismatches(b1);
    if (T22 == M2 && first_mismatches(b2) < mm2)
      mm2 = first_mismatches(b2);
    if (mm1 != mm2)
      return mm1 < mm2 ? 1 : 2;
    uint32_t const mn1 = T11 < T12 ? T11 : T12, mn2 = T21 < T22 ? T21 : T22;
    if (mn1 < mn2)
      return 1;
    if (mn2 < mn1)
      return 2;
    return 0;
  }
  if (M2 == 0u && T11 >= 63u && T12 >= 63u)
    return 1;
  if (M1 == 0u && T21 >= 63u && T22 >= 63u)
    return 2;
  return 1;
}

// vcf_writer.cpp:28-60
GTX_DEV bool geno_is_good(GraphView const & g, ScoreParams const & par, Geno const & ge, bool & fully, bool & unique)
{
  fully = true;
  unique = true;
  if (ge.n_paths == 0)
    return false;
  uint32_t const * w = ge.body;
  RecPath p0;
  uint32_t r0s = 0, r0e = 0;
  for (uint32_t i = 0; i < ge.n_paths; ++i)
  {
    RecPath p;
    w = path_at(w, p, ge.wide);
    if (p.re - p.rs + 1u != ge.read_len)
      fully = false;
    uint32_t const rs_ = g_ref_reach_pos(g, p.start), re_ = g_ref_reach_pos(g, p.end);
    if (i == 0)
    {
      p0 = p;
      r0s = rs_;
      r0e = re_;
    }
    else if (r0s != rs_ && r0e != re_)
      unique = false; // all_paths_unique (genotype_paths.cpp:219-231)
  }
  uint32_t const size0 = p0.re - p0.rs + 1u;
  if (!fully && (!unique || size0 < 63))
    return false;
  double const ratio = static_cast<double>(p0.mism) / static_cast<double>(size0);
  if (ratio > 0.05)
    return false;
  if (!fully && ratio > 0.025)
    return false;
  if (par.is_sv_graph && (!fully || size0 < 90 || ratio > 0.03))
    return false;
  if (par.hq_reads && (!fully || size0 < 90 || ratio > 0.035))
    return false;
  return true;
}

// Haplotype::explains (std::bitset<MAX_NUMBER_OF_HAPLOTYPES>, haplotype.hpp) as NW 64-bit words: one word where every
// site has at most 64 alleles, GTX_WIDE_MASK_WORDS / 2 in the scoring pass of graphs with wider sites
template <uint32_t NW>
struct AlleleSet
{
  uint64_t w[NW];
  static constexpr uint32_t WORDS32 = 2 * NW;
  GTX_DEV void clear()
  {
    for (uint32_t i = 0; i < NW; ++i)
      w[i] = 0;
  }
  GTX_DEV void add_words(uint32_t const * m, uint32_t n32) // n32 <= WORDS32, even
  {
    for (uint32_t i = 0; i < n32 / 2; ++i)
      w[i] |= (static_cast<uint64_t>(m[2 * i + 1]) << 32) | m[2 * i];
  }
  GTX_DEV uint32_t count() const
  {
    uint32_t c = 0;
    for (uint32_t i = 0; i < NW; ++i)
      c += static_cast<uint32_t>(__builtin_popcountll(w[i]));
    return c;
  }
  GTX_DEV bool test(uint32_t a) const { return a < 64 * NW && ((w[a >> 6] >> (a & 63u)) & 1ull); }
  template <class F>
  GTX_DEV void for_each(F && f) const // alleles ascending
  {
    for (uint32_t i = 0; i < NW; ++i)
      for (uint64_t m = w[i]; m; m &= m - 1)
        f(64 * i + static_cast<uint32_t>(__builtin_ctzll(m)));
  }
};

template <uint32_t NW_>
struct RecentHapT // one entry of `recent_ids` + the haplotype's transient explains/coverage (vcf_writer.cpp:519-585); 16 bytes where NW = 1
{
  static constexpr uint32_t NW = NW_;
  uint32_t site;
  uint16_t coverage; // an allele number (< 64 x NW) or one of NO_COVERAGE / MULTI_ALT_COVERAGE / MULTI_REF_COVERAGE
  uint16_t overlapping;
  AlleleSet<NW_> explains;
};
using RecentHap = RecentHapT<1>;
using RecentHapWide = RecentHapT<GTX_WIDE_MASK_WORDS / 2>;

// Where the entries of one read live.  PtrTable: an array (per-thread arrays of the scoring kernel, the HBM tables of the second
// scoring pass, host vectors of the emulation).  StridedTable: entry j of THIS lane at base[j * STRIDE] -- tables in LDS, entry by
// entry across the lanes of a workgroup.  (Measured for the scoring kernel, round 5: 8 entries x 64 lanes x 16 B of LDS instead of
// the per-thread arrays in scratch memory -- cfg3 0.743 ms against 0.674, cfg2 0.232 against 0.185: the LDS costs a wavefront per
// SIMD, and the scratch accesses were not what the kernel waits for.  The kernel keeps its arrays.)
template <class RH>
struct PtrTable
{
  using entry = RH;
  RH * p;
  GTX_DEV RH & operator[](uint32_t j) const { return p[j]; }
  GTX_DEV PtrTable after(uint32_t n) const { return PtrTable{p + n}; }
};
template <class RH, uint32_t STRIDE>
struct StridedTable
{
  using entry = RH;
  RH * p;
  GTX_DEV RH & operator[](uint32_t j) const { return p[j * STRIDE]; }
  GTX_DEV StridedTable after(uint32_t n) const { return StridedTable{p + n * STRIDE}; }
};

GTX_DEV uint32_t add_coverage(uint32_t coverage, uint32_t c) // haplotype.cpp:180-227
{
  if (coverage == NO_COVERAGE)
    return c;
  if (coverage == MULTI_ALT_COVERAGE)
    return c == 0 ? MULTI_REF_COVERAGE : coverage;
  if (coverage == MULTI_REF_COVERAGE)
    return coverage;
  if (coverage != c)
    return (coverage == 0 || c == 0) ? MULTI_REF_COVERAGE : MULTI_ALT_COVERAGE;
  return coverage;
}

struct ScoreAcc // device pointers, see gtx_score_buffers in include/gtx.h
{
  uint32_t n_samples;
  uint32_t conn_cap;
  uint32_t * log_score;
  uint32_t * gt_cov;
  uint32_t * hap_u32;
  unsigned long long * stat_u64;
  uint32_t * stat_u32;
  uint32_t * conn_log;
  uint32_t * conn_count;
  uint32_t * conn_near; // NULL: every connection is logged
  uint32_t const * big_records; // the context's arena for records longer than rec_words
  // a record the caller has copied (whole, in one round trip) to faster memory: geno_of reads the copy (per thread; NULL: none)
  uint32_t const * staged_from = nullptr;
  uint32_t const * staged_copy = nullptr;
  unsigned long long combine_base = 0; // the lowest address of the accumulators above (the scoring kernel's LDS table keys counters by their distance from it)
  // dense records of the position-hinted pass (gtx_align_batch_planes_compact): GTX_COMPACT_WORDS words per read; a task whose
  // byte of the side array carries GTX_TASK_COMPACT has its record there, not in its slot
  uint32_t const * compact = nullptr;
  uint8_t const * compact_flags = nullptr;
  uint32_t * ref_depth = nullptr; // SV calling: [n_samples][ref_depth_len + 1] difference array of ReferenceDepth (NULL: not kept)
  uint32_t ref_depth_len = 0;
  // replay mode (gtx_scores_replay): nothing is added to the accumulators; instead every explain_to_score call on a marked
  // (haplotype, sample) cell is logged with its epsilon and explain set, keyed by the item that caused it
  uint32_t const * replay_cells = nullptr; // bitmap over cells sample * n_hap + hap (NULL: normal scoring)
  struct ReplayEntry * replay_log = nullptr;
  uint32_t * replay_count = nullptr; // [0] entries wanted, capacity replay_cap
  uint32_t replay_cap = 0;
  uint32_t replay_item = 0; // index of the item being scored (set per thread)
};

// one call of Haplotype::explain_to_score (haplotype.cpp:462-585) on a cell whose max_log_score reached the sequential
// guard at :560 -- the only place where the order of the reads matters
struct ReplayEntry
{
  uint32_t item, cell;  // score item (its position in the region's item sequence = the reference's call order), cell
  uint32_t order_eps;   // epsilon_exponent | which read of the item << 8
  uint32_t mask_lo, mask_hi; // explains (sites of at most 64 alleles)
  uint32_t pad;
};

// ReferenceDepth::add_genotype_paths (src/graph/reference_depth.cpp:109-201): every accepted read adds one to the depth of
// its sample over the reference span of its path -- or, with several paths, over the union of their spans (each trimmed
// by 4 positions at both ends when it is 50 or longer).  Kept as a difference array (+1 at the first position, -1 behind
// the last one: two atomics per span instead of one per position); gtx_ref_depth_finalize turns it into depths.
template <class W>
GTX_DEV void add_ref_depth(GraphView const & g, ScoreAcc const & acc, Geno const & ge, uint32_t sample)
{
  if (!acc.ref_depth || ge.n_paths == 0)
    return;
  long const offset = g.first_order, size = acc.ref_depth_len;
  uint32_t * depth = acc.ref_depth + static_cast<uint64_t>(sample) * (acc.ref_depth_len + 1u);
  auto span_of = [&](RecPath const & p, long & a, long & b)
  {
    a = static_cast<long>(g_ref_reach_pos(g, p.start)) - static_cast<long>(p.rs);
    b = static_cast<long>(g_ref_reach_pos(g, p.end)) + (static_cast<long>(ge.read_len) - 1 - static_cast<long>(p.re));
  };
  auto to_index = [&](long a, long b, long & i0, long & i1) // start_pos_to_index / end_pos_to_index (:220-229), clipped to the array
  {
    i0 = a < offset ? 0 : a - offset;
    i1 = b > offset + size ? size : b + 1 - offset;
    if (i1 > size)
      i1 = size;
    return i0 < size && i1 > i0;
  };
  RecPath p;
  uint32_t const * w = path_at(ge.body, p, ge.wide);
  if (p.re - p.rs + 1 < 63)
    return;
  long a, b, i0, i1;
  if (ge.n_paths == 1)
  {
    span_of(p, a, b);
    if (to_index(a, b, i0, i1))
    {
      W::atomic_add_u32(depth + i0, 1u);
      W::atomic_add_u32(depth + i1, 0xFFFFFFFFu);
    }
    return;
  }
  // union of the spans, swept from the left without sorting them: the next piece starts at the smallest start behind what
  // is covered so far and runs while spans touch or overlap it
  long covered = -1; // last index covered so far
  for (;;)
  {
    long best0 = -1, best1 = -1;
    w = ge.body;
    for (uint32_t k = 0; k < ge.n_paths; ++k)
    {
      w = path_at(w, p, ge.wide);
      span_of(p, a, b);
      if (b - a >= 50)
      {
        a += 4;
        b -= 4;
      }
      if (b < offset || !to_index(a, b, i0, i1))
        continue;
      if (i0 <= covered)
        i0 = covered + 1;
      if (i0 >= i1)
        continue;
      if (best0 < 0 || i0 < best0 || (i0 == best0 && i1 > best1))
      {
        best0 = i0;
        best1 = i1;
      }
    }
    if (best0 < 0)
      break;
    // extend the piece by every span that starts inside it or right behind it
    for (bool grown = true; grown;)
    {
      grown = false;
      w = ge.body;
      for (uint32_t k = 0; k < ge.n_paths; ++k)
      {
        w = path_at(w, p, ge.wide);
        span_of(p, a, b);
        if (b - a >= 50)
        {
          a += 4;
          b -= 4;
        }
        if (b < offset || !to_index(a, b, i0, i1))
          continue;
        if (i0 <= best1 && i1 > best1)
        {
          best1 = i1;
          grown = true;
        }
      }
    }
    W::atomic_add_u32(depth + best0, 1u);
    W::atomic_add_u32(depth + best1, 0xFFFFFFFFu);
    covered = best1 - 1;
  }
}

template <class W>
GTX_DEV void emit_conn(GraphView const & g, ScoreAcc const & acc, uint32_t sample, uint32_t h1, uint32_t b1, uint32_t h2, uint32_t b2,
                       uint32_t count)
{
  if (count == 0)
    return;
  if (acc.conn_near && h2 > h1 && h2 <= g.near_last[h1])
  {
    // a pair genotyping reads (hts_parallel_reader.cpp:800-801): dense counters, no log entry
    uint64_t const first = g.allele_off[h1 + 1];
    uint64_t const width = g.allele_off[g.near_last[h1]] + g.ref_nvar[g.near_last[h1]] - first;
    W::atomic_add_u32(acc.conn_near + sample * g.total_near + g.near_off[h1] + b1 * width + (g.allele_off[h2] - first) + b2, count);
    return;
  }
  uint32_t const slot = W::atomic_claim_u32(acc.conn_count);
  if (slot >= acc.conn_cap)
  {
    W::atomic_add_u32(acc.conn_count + 1, 1u);
    return;
  }
  uint32_t * e = acc.conn_log + static_cast<uint64_t>(slot) * 6;
  e[0] = sample;
  e[1] = h1;
  e[2] = b1;
  e[3] = h2;
  e[4] = b2;
  e[5] = count;
}

// push_to_haplotype_scores (vcf_writer.cpp:503-676), first half: the sites the read's paths touch with their explain
// masks and coverage, ascending site (the reference's std::map order).  No effect on shared state.  Returns the number
// of entries, or 0xFFFFFFFF when the read touches more than `cap` sites.
template <class Tab>
GTX_DEV uint32_t collect_recent(GraphView const & g, Geno const & ge, Tab recent, uint32_t cap)
{
  using RecentHap = typename Tab::entry;
  constexpr uint32_t NW = RecentHap::NW;
  uint32_t n = 0;
  uint32_t const * w = ge.body;
  uint32_t const mw = ge.wide ? GTX_WIDE_MASK_WORDS : 2u;
  if (mw > AlleleSet<NW>::WORDS32)
    return 0xFFFFFFFFu; // a record with wide allele sets: for the pass whose tables hold them
  for (uint32_t i = 0; i < ge.n_paths; ++i)
  {
    RecPath p;
    w = path_at(w, p, ge.wide);
    int64_t const s_reach = g_ref_reach_pos(g, p.start), e_reach = g_ref_reach_pos(g, p.end);
    for (uint32_t k = 0; k < p.nvar; ++k)
    {
      uint32_t const site = p.site(k);
      uint32_t const * const mask = p.mask(k);
      uint32_t members = 0, lowest = 0;
      for (uint32_t x = mw; x-- > 0;)
        if (mask[x])
        {
          members += static_cast<uint32_t>(__builtin_popcount(mask[x]));
          lowest = 32 * x + static_cast<uint32_t>(__builtin_ctz(mask[x]));
        }
      if (members == 0)
        continue;
      int64_t const order = site_order(g, site);
      bool const overlapping = s_reach + 3 <= order && e_reach - 3 > order;
      uint32_t j = 0;
      for (; j < n; ++j)
        if (recent[j].site == site)
          break;
      if (j == n)
      {
        if (n >= cap)
          return 0xFFFFFFFFu;
        RecentHap & fresh = recent[n++];
        fresh.site = site;
        fresh.coverage = static_cast<uint16_t>(NO_COVERAGE);
        fresh.explains.clear();
        fresh.overlapping = 0;
      }
      RecentHap & rh = recent[j];
      rh.overlapping = static_cast<uint16_t>(rh.overlapping || overlapping);
      rh.explains.add_words(mask, mw);
      uint32_t cov = rh.coverage;
      if (members == 1)
        cov = add_coverage(cov, lowest);
      else
      {
        cov = add_coverage(cov, 1);
        cov = add_coverage(cov, (mask[0] & 1u) ? 0u : 2u);
      }
      rh.coverage = static_cast<uint16_t>(cov);
    }
  }
  // std::map order: ascending haplotype index
  for (uint32_t a = 1; a < n; ++a)
  {
    RecentHap const x = recent[a];
    uint32_t b = a;
    while (b > 0 && recent[b - 1].site > x.site)
    {
      recent[b] = recent[b - 1];
      --b;
    }
    recent[b] = x;
  }
  return n;
}

// push_to_haplotype_scores, second half: everything the read adds to the accumulators
// epsilon_exponent of explain_to_score (haplotype.cpp:471-501; no low-quality-base penalty in genotype-only mode)
GTX_DEV uint32_t explain_epsilon(Geno const & ge, bool fully, bool unique, bool overlapping)
{
  long e = 12;
  e -= static_cast<long>(first_mismatches(ge));
  if (!unique)
    e -= 3;
  if (ge.flags & F_MAPQ_BAD)
    e -= 2;
  if (!fully)
    e -= 3;
  if (!overlapping)
    e -= 1;
  return static_cast<uint32_t>((e > 8 ? e : 8) - 4);
}

template <class W, class Tab>
GTX_DEV void apply_recent(GraphView const & g, ScoreAcc const & acc, Geno const & ge, bool fully, bool unique, uint32_t sample,
                          Tab recent, uint32_t n, uint32_t order = 0)
{
  using RecentHap = typename Tab::entry;
  if (acc.replay_cells)
  {
    for (uint32_t a = 0; a < n; ++a)
    {
      uint32_t const cell = sample * g.n_hap + recent[a].site;
      if (!((acc.replay_cells[cell >> 5] >> (cell & 31u)) & 1u))
        continue;
      uint32_t const slot = W::atomic_claim_u32(acc.replay_count);
      if (slot < acc.replay_cap)
        acc.replay_log[slot] = ReplayEntry{acc.replay_item, cell, explain_epsilon(ge, fully, unique, recent[a].overlapping) | (order << 8),
                                           static_cast<uint32_t>(recent[a].explains.w[0]), static_cast<uint32_t>(recent[a].explains.w[0] >> 32), 0u};
    }
    return;
  }
  uint32_t const clipped_bp = ge.read_len - ge.longest;
  uint32_t const mismatches = first_mismatches(ge);
  // connections between the sites of this read (vcf_writer.cpp:587-636)
  for (uint32_t a = 0; a < n; ++a)
  {
    uint32_t const n1 = recent[a].explains.count();
    if (n1 == 0 || n1 > 64)
      continue;
    for (uint32_t b = a + 1; b < n; ++b)
    {
      uint32_t const n2 = recent[b].explains.count();
      if (n2 == 0 || n2 > 64)
        continue;
      uint32_t const weight = n1 * n2;
      uint32_t const repeat = weight >= 3 ? 6 / weight : 1;
      if (repeat == 0)
        continue;
      recent[a].explains.for_each([&](uint32_t b1) {
        recent[b].explains.for_each([&](uint32_t b2) { emit_conn<W>(g, acc, sample, recent[a].site, b1, recent[b].site, b2, repeat); });
      });
    }
  }
  // move the explanations to statistics, likelihood and depth (vcf_writer.cpp:638-673)
  uint64_t const nh = g.n_hap;
  for (uint32_t a = 0; a < n; ++a)
  {
    RecentHap const rh = recent[a];
    uint32_t const h = rh.site;
    // (the site's three table entries in one round trip: a load behind an atomic is not moved in front of it by the compiler)
    uint32_t const cnum = g.ref_nvar[h];
    uint64_t const aoff = g.allele_off[h];
    uint64_t const toff = g.tri_off[h];
    uint32_t const cov = rh.coverage;
    bool const unique_allele = cov < MULTI_REF_COVERAGE;
    // clipped_reads_to_stats (haplotype.cpp:229-244)
    if (clipped_bp != 0)
    {
      if (cov != NO_COVERAGE)
        W::atomic_add_u32(acc.stat_u32 + h, 1u);
      if (unique_allele)
        W::atomic_add_u64(acc.stat_u64 + nh + 2 * (aoff + cov) + 0, (static_cast<uint64_t>(clipped_bp) * 1000u) / ge.read_len);
    }
    // mapq_to_stats (:246-261)
    if (ge.mapq != 255)
    {
      uint64_t const sq = static_cast<uint64_t>(ge.mapq) * ge.mapq;
      if (cov != NO_COVERAGE)
        W::atomic_add_u64(acc.stat_u64 + h, sq);
      if (unique_allele)
        W::atomic_add_u64(acc.stat_u64 + nh + 2 * (aoff + cov) + 1, sq);
    }
    if (unique_allele)
    {
      uint32_t * s32 = acc.stat_u32 + nh + 6 * (aoff + cov);
      // strand_to_stats (:263-287)
      bool const fwd = (ge.flags & F_SEQ_REVERSED) == 0, first = (ge.flags & F_FIRST_IN_PAIR) != 0;
      W::atomic_add_u32(s32 + (fwd ? (first ? 2 : 4) : (first ? 3 : 5)), 1u);
      // mismatches_to_stats (:289-300), uint8_t argument
      uint32_t const mm8 = mismatches & 0xFFu;
      if (mm8 != 0)
        W::atomic_add_u32(s32 + 1, (mm8 * 1000u) / ge.read_len);
      // score_diff_to_stats (:302-311)
      if (ge.score_diff != 0)
        W::atomic_add_u32(s32 + 0, ge.score_diff);
    }
    // explain_to_score (:462-585)
    uint32_t const eps = explain_epsilon(ge, fully, unique, rh.overlapping);
    uint32_t * cell = acc.hap_u32 + (static_cast<uint64_t>(sample) * nh + h) * 4;
    W::atomic_add_u32(cell + 0, eps);
    uint32_t * ls = acc.log_score + static_cast<uint64_t>(sample) * g.total_tri + toff;
    uint32_t idx = 0;
    for (uint32_t y = 0; y < cnum; ++y)
    {
      bool const ey = rh.explains.test(y);
      for (uint32_t x = 0; x <= y; ++x, ++idx)
      {
        bool const ex = rh.explains.test(x);
        if (ex && ey)
          W::atomic_add_u32(ls + idx, eps);
        else if (ex || ey)
          W::atomic_add_u32(ls + idx, eps - 1);
      }
    }
    // coverage_to_gts (:315-361)
    if (cov == MULTI_REF_COVERAGE)
      W::atomic_add_u32(cell + 1, 1u);
    else if (cov == MULTI_ALT_COVERAGE)
    {
      W::atomic_add_u32(cell + 1, 1u);
      W::atomic_add_u32(cell + 2, 1u);
      if (ge.proper_pair)
        W::atomic_add_u32(cell + 3, 1u);
    }
    else if (cov != NO_COVERAGE)
    {
      W::atomic_add_u32(acc.gt_cov + static_cast<uint64_t>(sample) * g.total_allele + aoff + cov, 1u);
      if (cov > 0 && ge.proper_pair)
        W::atomic_add_u32(cell + 3, 1u);
    }
  }
}

// one call of genotype_only() that reaches the writer (hts_parallel_reader.cpp:283-337)
// Triage: true when no orientation of the item's read(s) carries a variant site -- whatever the orientation / pair
// selection decides, nothing can be added to the accumulators (the same early exits are inside score_item).  Costs one
// record header per read whose reverse orientation was not aligned.
// task_flags (may be NULL): the dense side array of gtx_align_batch_flags -- one byte per (read, orientation) instead of one
// cache line per record header.
GTX_DEV bool item_is_trivial(gtx_score_item const & it, uint32_t const * records, uint32_t rec_words, bool keeps_depth = false,
                             uint8_t const * task_flags = nullptr)
{
  if (keeps_depth && it.second.align_index != INVALID)
    return false; // (SV calling: every selected pair counts for the reference depth, with or without variant sites)
  gtx_rec_meta const * ms[2] = {&it.first, &it.second};
  for (int r = 0; r < 2; ++r)
  {
    gtx_rec_meta const & m = *ms[r];
    if (r == 1 && m.align_index == INVALID)
      break;
    if (task_flags)
    {
      uint8_t const * f = task_flags + 2ull * m.align_index;
      if ((f[0] & GTX_TASK_HAS_VARIANTS) || (!(m.flag & GTX_FLAG_FORWARD_ONLY) && (f[1] & GTX_TASK_HAS_VARIANTS)))
        return false;
      continue;
    }
    uint32_t const * rec = records + static_cast<uint64_t>(m.align_index) * 2 * rec_words;
    if (rec[1] & GTX_REC_HAS_VARIANTS)
      return false;
    if (!(m.flag & GTX_FLAG_FORWARD_ONLY) && (rec[rec_words + 1] & GTX_REC_HAS_VARIANTS))
      return false;
  }
  return true;
}

// r1 / r2: tables of `cap` entries each.  Returns false, with nothing added to the accumulators, when a read of the item
// touches more than `cap` variant sites (the caller then redoes the item with larger tables).
// shared: r2 is not looked at -- the second read's entries follow the first read's in r1, `cap` is the room for both.
template <class W, class Tab>
GTX_DEV bool score_item(GraphView const & g, ScoreParams const & par, gtx_score_item const & it, uint32_t const * records,
                        uint32_t rec_words, ScoreAcc const & acc, Tab r1, Tab r2, uint32_t cap, bool shared = false)
{
  if (it.second.align_index == INVALID)
  {
    // update_unpaired_read_paths (alignment.cpp:365-455).  clipped_count() returns 0/1, so IS_CLIPPED is never set.
    gtx_rec_meta const & m = it.first;
    uint32_t const mflag = m.flag & 0x7FFFu; // (without GTX_FLAG_FORWARD_ONLY)
#ifdef GTX_PROF_SCORE // (a one-off profiling build: cycles of the parts of a single read's item, lane 0 of a workgroup, in the express pass' slots)
    unsigned long long const ps0 = W::clock();
#define GTX_PS(slot, since) if ((threadIdx.x & 63u) == 0) atomicAdd(g.prof + (slot), W::clock() - (since))
#else
#define GTX_PS(slot, since)
#endif
    Geno fwd = geno_of(records, rec_words, acc, m.align_index, 0);
    Geno rev = (m.flag & GTX_FLAG_FORWARD_ONLY) ? empty_orientation(fwd) : geno_of(records, rec_words, acc, m.align_index, 1);
    if (!fwd.has_var && !rev.has_var)
      return true; // whichever orientation wins, it touches no variant site: nothing to add
    GTX_PS(16, ps0 + (fwd.n_paths & 0u));
#ifdef GTX_PROF_SCORE
    unsigned long long const ps1 = W::clock();
#endif
    int const which = compare_single(fwd, rev);
    if (which == 0)
      return true;
    Geno & ge = which == 1 ? fwd : rev;
    ge.flags = (which == 1 ? mflag : (mflag ^ F_SEQ_REVERSED)) & ~static_cast<uint32_t>(F_PROPER_PAIR) & 0xFFFFu;
    ge.mapq = m.mapq;
    if (m.mapq < 25)
      ge.flags |= F_MAPQ_BAD;
    ge.score_diff = m.score_diff;
    if (par.is_segment_calling)
      return true;
    bool fully, unique;
    bool const good = geno_is_good(g, par, ge, fully, unique);
    GTX_PS(17, ps1 + (good ? 0u : 0u));
    if (good)
    {
#ifdef GTX_PROF_SCORE
      unsigned long long const ps2 = W::clock();
#endif
      uint32_t const n = collect_recent(g, ge, r1, cap);
      if (n == 0xFFFFFFFFu)
        return false;
      GTX_PS(18, ps2 + (n & 0u));
#ifdef GTX_PROF_SCORE
      unsigned long long const ps3 = W::clock();
#endif
      apply_recent<W>(g, acc, ge, fully, unique, it.sample, r1, n);
      GTX_PS(19, ps3);
#ifdef GTX_PROF_SCORE
      if ((threadIdx.x & 63u) == 0)
        atomicAdd(g.prof + 31, 1ull);
#endif
    }
    return true;
  }
  // update_paths for both records (alignment.cpp:482-545): only the forward-orientation geno gets IS_MAPQ_BAD
  Geno q[4];
  gtx_rec_meta const * ms[2] = {&it.first, &it.second};
  for (int r = 0; r < 2; ++r)
  {
    gtx_rec_meta const & m = *ms[r];
    Geno & f = q[2 * r];
    Geno & v = q[2 * r + 1];
    uint32_t const mflag = m.flag & 0x7FFFu; // (without GTX_FLAG_FORWARD_ONLY)
    f = geno_of(records, rec_words, acc, m.align_index, 0);
    v = (m.flag & GTX_FLAG_FORWARD_ONLY) ? empty_orientation(f) : geno_of(records, rec_words, acc, m.align_index, 1);
    f.flags = (mflag & ~static_cast<uint32_t>(F_PROPER_PAIR)) & 0xFFFFu;
    if (m.mapq < 25)
      f.flags |= F_MAPQ_BAD;
    v.flags = ((mflag ^ F_SEQ_REVERSED) & ~static_cast<uint32_t>(F_PROPER_PAIR)) & 0xFFFFu;
    f.mapq = v.mapq = m.mapq;
    f.score_diff = v.score_diff = m.score_diff;
    f.proper_pair = v.proper_pair = true; // ml_insert_size = |isize|, never INSERT_SIZE_WHEN_NOT_PROPER_PAIR for int32 isize
  }
  if (!acc.ref_depth && !q[0].has_var && !q[1].has_var && !q[2].has_var && !q[3].has_var)
    return true; // no orientation of either mate touches a variant site: nothing to add
  // get_better_paths (alignment.cpp:557-620)
  int arr[4] = {-1, -1, -1, -1};
  for (int k = 0; k < 4; ++k)
    arr[((q[k].flags & F_FIRST_IN_PAIR) != 0) + 2 * ((q[k].flags & F_SEQ_REVERSED) == 0)] = k;
  if (arr[0] < 0 || arr[1] < 0 || arr[2] < 0 || arr[3] < 0)
    return true;
  int const which = compare_pairs(q[arr[3]], q[arr[0]], q[arr[1]], q[arr[2]]);
  if (which == 0)
    return true;
  Geno & first = which == 1 ? q[arr[3]] : q[arr[1]];
  Geno & second = which == 1 ? q[arr[0]] : q[arr[2]];
  first.flags |= F_PROPER_PAIR;
  second.flags |= F_PROPER_PAIR;
  // SV calling: the reads of a selected pair count for the reference depth (hts_parallel_reader.cpp:324-329), a leftover
  // read alone (:739-741)
  if (!acc.replay_cells)
  {
    add_ref_depth<W>(g, acc, first, it.sample);
    if (!(it.kind & GTX_ITEM_LEFTOVER))
      add_ref_depth<W>(g, acc, second, it.sample);
  }
  // update_haplotype_scores_geno, pair overload (vcf_writer.cpp:143-250)
  bool f1, u1, f2, u2;
  bool const good1 = geno_is_good(g, par, first, f1, u1), good2 = geno_is_good(g, par, second, f2, u2);
  if (it.kind & GTX_ITEM_LEFTOVER)
  {
    // a read whose mate never came (hts_parallel_reader.cpp:733-744): the single-read overload on better_paths.first
    if (!par.is_segment_calling && good1)
    {
      uint32_t const n = collect_recent(g, first, r1, cap);
      if (n == 0xFFFFFFFFu)
        return false;
      apply_recent<W>(g, acc, first, f1, u1, it.sample, r1, n);
    }
    return true;
  }
  if (par.is_segment_calling && (!good1 || !good2))
    return true;
  uint32_t n1 = 0, n2 = 0;
  if (good1)
    n1 = collect_recent(g, first, r1, cap);
  if (n1 == 0xFFFFFFFFu)
    return false;
  if (shared)
    r2 = r1.after(n1);
  if (good2)
    n2 = collect_recent(g, second, r2, shared ? cap - n1 : cap);
  if (n2 == 0xFFFFFFFFu)
    return false;
  if (good1)
    apply_recent<W>(g, acc, first, f1, u1, it.sample, r1, n1);
  if (good2)
    apply_recent<W>(g, acc, second, f2, u2, it.sample, r2, n2, 1);
  if (acc.replay_cells)
    return true; // (replay mode: the connections between the mates were counted by the scoring pass)
  // cross links between the two mates' sites: every (site, allele) key of one mate gets one count towards every key
  // of the other mate that lies on a later site (vcf_writer.cpp:186-227)
  for (uint32_t a = 0; a < n1; ++a)
  {
    uint32_t const c1 = r1[a].explains.count();
    if (c1 == 0 || c1 > 64)
      continue;
    for (uint32_t b = 0; b < n2; ++b)
    {
      uint32_t const c2 = r2[b].explains.count();
      if (c2 == 0 || c2 > 64 || r1[a].site == r2[b].site)
        continue;
      bool const fwd = r2[b].site > r1[a].site;
      r1[a].explains.for_each([&](uint32_t b1) {
        r2[b].explains.for_each([&](uint32_t b2) {
          if (fwd)
            emit_conn<W>(g, acc, it.sample, r1[a].site, b1, r2[b].site, b2, 1);
          else
            emit_conn<W>(g, acc, it.sample, r2[b].site, b2, r1[a].site, b1, 1);
        });
      });
    }
  }
  return true;
}

// the same over two arrays of `cap` entries each
template <class W, uint32_t NW>
GTX_DEV bool score_item(GraphView const & g, ScoreParams const & par, gtx_score_item const & it, uint32_t const * records,
                        uint32_t rec_words, ScoreAcc const & acc, RecentHapT<NW> * r1, RecentHapT<NW> * r2, uint32_t cap)
{
  return score_item<W>(g, par, it, records, rec_words, acc, PtrTable<RecentHapT<NW>>{r1}, PtrTable<RecentHapT<NW>>{r2}, cap, false);
}

// Genotype call of one (sample, haplotype) cell from the accumulators: get_haplotype_phred (src/typer/vcf.cpp:47-82) and
// the SampleCall built from it (src/typer/sample_call.cpp:34-131: constructor, get_gt_call, get_gq); the saturating
// stores of the reference's u8 / u16 counters (haplotype.cpp:19-44) are applied on the fly.
GTX_DEV void call_cell(GraphView const & g, uint64_t cell, uint32_t const * log_score, uint32_t const * gt_cov, uint32_t const * hap_u32,
                       uint8_t * phred, gtx_sample_call * calls)
{
  uint32_t const s = static_cast<uint32_t>(cell / g.n_hap), h = static_cast<uint32_t>(cell % g.n_hap);
  uint32_t const cnum = g.ref_nvar[h], n_tri = cnum * (cnum + 1) / 2;
  uint32_t const * ls = log_score + static_cast<uint64_t>(s) * g.total_tri + g.tri_off[h];
  uint8_t * ph = phred + static_cast<uint64_t>(s) * g.total_tri + g.tri_off[h];
  uint32_t mx = 0, mn = 0xFFFFFFFFu;
  for (uint32_t i = 0; i < n_tri; ++i)
  {
    uint32_t const v = ls[i];
    mx = v > mx ? v : mx;
    mn = v < mn ? v : mn;
  }
  // PL = llround((max - score) * 10 log10(2)), 255 when that is >= 255, all 0 when every genotype scores the same
  double const LOG10_HALF_times_10 = 3.01029995663981195213738894724493026768189881462108541;
  uint32_t gt_x = 0, gt_y = 0, zeros = 0, next_lowest = 255;
  bool have_gt = false;
  uint32_t i = 0;
  for (uint32_t y = 0; y < cnum; ++y)
    for (uint32_t x = 0; x <= y; ++x, ++i)
    {
      uint32_t p = 0;
      if (mx != mn)
      {
        long long const score = llround(static_cast<double>(mx - ls[i]) * LOG10_HALF_times_10);
        p = score < 255 ? static_cast<uint32_t>(score) : 255u;
      }
      ph[i] = static_cast<uint8_t>(p);
      if (p == 0)
      {
        ++zeros;
        if (!have_gt)
        {
          have_gt = true;
          gt_x = x;
          gt_y = y;
        }
      }
      else if (p < next_lowest)
        next_lowest = p;
    }
  uint32_t const * cov = gt_cov + static_cast<uint64_t>(s) * g.total_allele + g.allele_off[h];
  uint32_t const * cu = hap_u32 + cell * 4;
  auto sat8 = [](uint32_t v) { return v > 0xFFu ? 0xFFu : v; };
  auto sat16 = [](uint32_t v) { return v > 0xFFFFu ? 0xFFFFu : v; };
  uint32_t const ambiguous = sat8(cu[1]), ambiguous_alt = sat8(cu[2]), alt_pp = sat8(cu[3]);
  uint32_t alt_depth = ambiguous;
  for (uint32_t a = 1; a < cnum; ++a)
    alt_depth += sat16(cov[a]);
  gtx_sample_call c;
  c.gt_first = static_cast<uint16_t>(gt_x);
  c.gt_second = static_cast<uint16_t>(gt_y);
  c.ref_total_depth = static_cast<uint16_t>(sat16(sat16(cov[0]) + ambiguous - ambiguous_alt));
  c.alt_total_depth = static_cast<uint16_t>(sat16(alt_depth));
  c.gq = static_cast<uint8_t>(zeros > 1 ? 0u : next_lowest);
  c.ambiguous_depth = static_cast<uint8_t>(ambiguous);
  c.alt_proper_pair_depth = static_cast<uint8_t>(alt_pp);
  c.reserved = 0;
  calls[cell] = c;
}

} // namespace gtx
