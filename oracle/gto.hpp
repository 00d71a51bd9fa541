// gto.hpp -- CPU ORACLE (test infrastructure, NOT product code).
//
// A plain C++17 restatement of graphtyper's read -> pangenome-graph alignment
// + genotype-scoring hot path, written from the behaviour of the reference
// sources (cited as file:line relative to /root/reference).  It exists only so
// that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can
// check / time the HIP product against it.  Nothing under graphtyper_amd/
// may include, link or call this file.
//
// Pinning status: the k-mer packing, IUPAC expansion, Hamming-1 neighbours,
// get_num_kmers, index content (chr1-4, chr9, chr10 of the reference's
// index_test fixture), Path merge of two reference labels and the haplotype
// count are pinned against the reference's own known-answer tests
// (tests/test_oracle_pinned.py).  The reference itself cannot be built in this
// image (seqan, htslib, parallel-hashmap, cereal, paw submodules are empty and
// the build generates constants.hpp), and its tests for align_read /
// iterative_dfs / explain_to_score are disabled upstream, so for seed
// chaining, graph walks, filters and scoring this oracle is "parity unpinned":
// it follows the reference text line by line but has no golden vector.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <numeric>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

namespace gto
{
// include/graphtyper/constants.hpp.in:20-53
constexpr uint32_t K = 32;
constexpr uint32_t INVALID_ID = 0xFFFFFFFFu;
constexpr uint32_t MAX_NUMBER_OF_HAPLOTYPES = 2560u;
constexpr uint32_t SPECIAL_START = 0xD0000000u;
constexpr uint32_t MAX_UNIQUE_KMER_POSITIONS = 512;
constexpr uint32_t MAX_SEED_NUMBER_ALLOWING_MISMATCHES = 64;
constexpr uint32_t MAX_SEED_NUMBER_FOR_WALKING = 256;
constexpr uint32_t MAX_NUM_LOCATIONS_PER_PATH = 256;
constexpr long EPSILON_0_EXPONENT = 12;
constexpr int32_t INSERT_SIZE_WHEN_NOT_PROPER_PAIR = 0x7FFFFFFF;
constexpr uint16_t IS_PAIRED = 1, IS_PROPER_PAIR = 2, IS_UNMAPPED = 4, IS_SEQ_REVERSED = 16,
                   IS_MATE_SEQ_REVERSED = 32, IS_FIRST_IN_PAIR = 64, IS_MAPQ_BAD = 4096, IS_CLIPPED = 8192;

// include/graphtyper/utilities/options.hpp:34,77,82,87,89,90
struct Params
{
  long max_index_labels = 75;
  bool is_sv_graph = false;
  bool hq_reads = false;
  bool force_align_both_orientations = false;
  bool is_segment_calling = false;
  int sam_flag_filter = 3840;
};

// ---------------------------------------------------------------------------
// k-mer packing  (src/utilities/type_conversions.cpp)
// ---------------------------------------------------------------------------

// type_conversions.cpp:19-41 ; aborts on anything but ACGT
inline uint64_t base_to_u64(char c)
{
  switch (c)
  {
  case 'A': return 0;
  case 'C': return 1;
  case 'G': return 2;
  case 'T': return 3;
  default: throw std::runtime_error(std::string("invalid base ") + c);
  }
}

// type_conversions.cpp:75-87
inline uint64_t to_uint64(std::string const & s, int i = 0)
{
  uint64_t d = 0;
  for (int const j = i + 32; i < j; ++i)
    d = (d << 2) + base_to_u64(s[i]);
  return d;
}

// type_conversions.cpp:322-349
inline std::string to_dna_str(uint64_t d, int k = 32)
{
  std::string out;
  while (k > 0)
  {
    --k;
    out.push_back("ACGT"[(d >> (2 * k)) & 3u]);
  }
  return out;
}

// IUPAC 4-bit code of a character: bit0 A, bit1 C, bit2 G, bit3 T (equals BAM's
// seq_nt16 encoding).  Unknown characters map to N (15).
inline uint8_t iupac_code(char c)
{
  static const char tbl[] = "=ACMGRSVTWYHKDBN";
  for (int i = 1; i < 16; ++i)
    if (tbl[i] == c)
      return static_cast<uint8_t>(i);
  return 15;
}

inline char iupac_char(uint8_t code)
{
  // seqan's Iupac alphabet prints value 0 as 'U'; BAM code 0 ('=') is converted to N before it
  // gets here (hts_parallel_reader.cpp:226-243 assigns the *character* '=' to a seqan Iupac).
  static const char tbl[] = "NACMGRSVTWYHKDBN";
  return tbl[code & 15];
}

// type_conversions.cpp:207-266 ; s holds 4-bit IUPAC codes
inline std::vector<uint64_t> to_uint64_vec(std::vector<uint8_t> const & s, std::size_t i)
{
  std::vector<uint64_t> uints(1, 0u);
  for (std::size_t const j = i + 32; i < j; ++i)
  {
    std::size_t const origin_size = uints.size();
    if (origin_size > 97)
      return {};
    unsigned const code = s[i] & 15u;
    for (std::size_t u = 0; u < origin_size; ++u)
    {
      if (code == 15u || code == 0u)
      {
        uints.push_back(uints[u] * 4 + 0);
        uints.push_back(uints[u] * 4 + 1);
        uints.push_back(uints[u] * 4 + 2);
        uints[u] = (uints[u] << 2) + 3;
      }
      else
      {
        int left = __builtin_popcount(code);
        for (unsigned b = 0; b < 4; ++b)
        {
          if (!(code & (1u << b)))
            continue;
          if (left == 1)
            uints[u] = uints[u] * 4 + b;
          else
            uints.push_back(uints[u] * 4 + b);
          --left;
        }
      }
    }
  }
  return uints;
}

// type_conversions.cpp:272-288
inline std::array<uint64_t, 96> hamming1_keys(uint64_t key)
{
  std::array<uint64_t, 96> h;
  for (unsigned bb = 0; bb < 32; ++bb)
    for (uint64_t m = 1; m <= 3; ++m)
      h[bb * 3 + (m - 1)] = (m << (bb * 2)) ^ key;
  return h;
}

// type_conversions.cpp:351-389
inline std::array<uint64_t, 3> mismatches_of_last_base(uint64_t d)
{
  std::array<uint64_t, 3> r;
  int n = 0;
  for (uint64_t b = 0; b < 4; ++b)
    if (b != (d & 3u))
      r[n++] = (d & ~3ull) | b;
  return r;
}

inline std::array<uint64_t, 3> mismatches_of_first_base(uint64_t d)
{
  std::array<uint64_t, 3> r;
  int n = 0;
  for (uint64_t b = 0; b < 4; ++b)
    if (b != (d >> 62))
      r[n++] = (d & 0x3FFFFFFFFFFFFFFFull) | (b << 62);
  return r;
}

// kmer_help_functions.cpp:10-17
inline std::size_t get_num_kmers(std::size_t len)
{
  return len < K ? 0 : 1 + (len - K) / (K - 1);
}

// kmer_help_functions.cpp:19-30 : offset of the i-th (centred) k-mer
inline std::size_t get_ith_kmer_offset(std::size_t len, std::size_t i)
{
  return (len - K) % (K - 1) / 2 + (K - 1) * i;
}

// ---------------------------------------------------------------------------
// graph model (include/graphtyper/graph/{label,node,graph}.hpp)
// ---------------------------------------------------------------------------
struct KmerLabel // include/graphtyper/index/kmer_label.hpp:13-41
{
  uint32_t start_index = 0, end_index = 0, variant_id = INVALID_ID;
  KmerLabel() = default;
  KmerLabel(uint32_t s, uint32_t e, uint32_t v = INVALID_ID) : start_index(s), end_index(e), variant_id(v) {}
};

struct Label // label.hpp:17-41, label.cpp:35-38
{
  uint32_t order = 0;
  std::string dna;
  uint32_t variant_num = 0;
  uint32_t reach() const { return order + static_cast<uint32_t>(dna.size()) - 1; }
};

struct RefNode
{
  Label label;
  std::vector<uint32_t> out_var_ids;
};

struct VarNode
{
  Label label;
  uint32_t out_ref_id = 0;
  std::unordered_set<long> events, anti_events;
};

struct AltAllele
{
  std::string seq;
  std::unordered_set<long> events, anti_events;
};

struct VarRecord // include/graphtyper/graph/var_record.hpp
{
  uint32_t pos = 0; // 0-based
  std::string ref;
  std::unordered_set<long> ref_events, ref_anti_events;
  std::vector<AltAllele> alts;
  bool is_sv = false;

  void clear() // var_record.cpp:154-167
  {
    pos = 0;
    ref.clear();
    alts.clear();
    ref_events.clear();
    ref_anti_events.clear();
    is_sv = false;
  }

  bool is_snp_or_snps() const // var_record.cpp:416-419
  {
    for (auto const & a : alts)
      if (a.seq.size() != ref.size())
        return false;
    return true;
  }

  bool is_any_seq_larger_than(long val) const // var_record.cpp:408-414
  {
    if (static_cast<long>(ref.size()) > val)
      return true;
    for (auto const & a : alts)
      if (static_cast<long>(a.seq.size()) > val)
        return true;
    return false;
  }

  void add_suffix(std::string const & suffix) // var_record.cpp:373-379
  {
    for (auto & a : alts)
      a.seq += suffix;
    ref += suffix;
  }

  // helpers of var_record.cpp:24-109
  void absorb_ref_events(VarRecord const & o)
  {
    ref_events.insert(o.ref_events.begin(), o.ref_events.end());
    ref_anti_events.insert(o.ref_anti_events.begin(), o.ref_anti_events.end());
  }
  static void absorb_ref_events(AltAllele & a, VarRecord const & o)
  {
    a.events.insert(o.ref_events.begin(), o.ref_events.end());
    a.anti_events.insert(o.ref_anti_events.begin(), o.ref_anti_events.end());
  }
  void insert_prior_sequence(VarRecord const & previous) // :31-47
  {
    std::string const prefix = previous.ref.substr(0, pos - previous.pos);
    ref = prefix + ref;
    for (auto & a : alts)
      a.seq = prefix + a.seq;
    pos = previous.pos;
  }
  static void extend_record(VarRecord & current, VarRecord const & previous) // :49-64
  {
    std::string const tail = previous.ref.substr(current.ref.size());
    for (auto & a : current.alts)
      a.seq += tail;
    current.ref += tail;
  }
  static void extend_smaller_record(VarRecord & current, VarRecord & previous) // :66-79
  {
    if (current.ref.size() < previous.ref.size())
      extend_record(current, previous);
    else if (current.ref.size() > previous.ref.size())
      extend_record(previous, current);
  }
  void move_alts(std::vector<AltAllele> && from) // :81-107 (only alts already there before the call count as duplicates)
  {
    std::size_t const original = alts.size();
    for (auto & pa : from)
    {
      bool unique = true;
      for (std::size_t a = 0; a < original; ++a)
        if (alts[a].seq == pa.seq)
        {
          unique = false;
          break;
        }
      if (unique)
        alts.push_back(std::move(pa));
    }
  }
  static AltAllele make_alt(AltAllele const & prev, AltAllele const & curr, long jump) // alt.cpp:60-96
  {
    AltAllele n(prev);
    n.seq += curr.seq.substr(jump);
    n.events.insert(curr.events.begin(), curr.events.end());
    n.anti_events.insert(curr.anti_events.begin(), curr.anti_events.end());
    return n;
  }
  static bool is_ok_to_merge_alts(AltAllele const & prev, AltAllele const & curr) // alt.cpp:98-141
  {
    for (long e : curr.events)
      if (e >= 0 && prev.anti_events.count(e))
        return false;
    return true;
  }

  // GenomicRegion::add_reference_to_record_if_they_have_a_matching_prefix (src/graph/genomic_region.cpp:17-67, 236-256),
  // applied to every VCF record by the constructor (src/graph/constructor.cpp:1740-1744) before the graph is built:
  // while the reference allele is a prefix of (or has as prefix) an alt, or two alts are prefix related, one more
  // reference base is appended to all alleles
  void extend_while_prefix_related(std::string const & reference, long region_begin)
  {
    if (is_sv)
      return;
    auto prefix_match = [](std::string const & a, std::string const & b)
    {
      std::size_t const n = std::min(a.size(), b.size());
      return a.compare(0, n, b, 0, n) == 0;
    };
    auto related = [&]()
    {
      for (auto const & a : alts)
        if (prefix_match(ref, a.seq))
          return true;
      for (std::size_t i = 0; i + 1 < alts.size(); ++i)
        for (std::size_t j = i + 1; j < alts.size(); ++j)
          if (prefix_match(alts[i].seq, alts[j].seq))
          {
            if (alts[i].seq == alts[j].seq)
              throw std::runtime_error("duplicated alt alleles");
            return true;
          }
      return false;
    };
    std::size_t at = static_cast<std::size_t>(static_cast<long>(pos) - region_begin) + ref.size();
    while (at < reference.size() && reference[at] != 'N' && related())
    {
      ref.push_back(reference[at]);
      for (auto & a : alts)
        a.seq.push_back(reference[at]);
      ++at;
    }
  }

  void merge_one_path(VarRecord && prev) // var_record.cpp:179-200
  {
    if (prev.pos < pos)
      insert_prior_sequence(prev);
    extend_smaller_record(*this, prev);
    absorb_ref_events(prev);
    for (auto & a : alts)
      absorb_ref_events(a, prev);
    move_alts(std::move(prev.alts));
  }

  void merge(VarRecord && prev, long EXTRA_SUFFIX) // var_record.cpp:270-371
  {
    long const jump_size = static_cast<long>(pos) - static_cast<long>(prev.pos);
    long const oref_size = static_cast<long>(ref.size());
    if (jump_size > 0)
      insert_prior_sequence(prev);
    long const oref_size_pre = static_cast<long>(ref.size());
    extend_smaller_record(*this, prev);
    long const extension_size = static_cast<long>(ref.size()) - oref_size_pre;
    std::vector<AltAllele> fresh;
    for (auto const & prev_alt : prev.alts)
    {
      if (static_cast<long>(prev_alt.seq.size()) <= oref_size)
        continue;
      long const offset = static_cast<long>(ref.size()) - static_cast<long>(prev_alt.seq.size());
      if (jump_size - offset < 0)
        continue;
      long suffix_matches = 0;
      long const smaller = static_cast<long>(std::min(ref.size(), prev_alt.seq.size()));
      for (long k = 0; k < smaller; ++k)
      {
        if (ref[ref.size() - 1 - k] != prev_alt.seq[prev_alt.seq.size() - 1 - k])
          break;
        ++suffix_matches;
      }
      if (suffix_matches >= extension_size + EXTRA_SUFFIX)
      {
        AltAllele prefix_alt(prev_alt);
        prefix_alt.seq = prev_alt.seq.substr(0, jump_size - offset);
        for (auto const & curr_alt : alts)
          if (is_ok_to_merge_alts(prefix_alt, curr_alt))
            fresh.push_back(make_alt(prefix_alt, curr_alt, jump_size));
      }
    }
    absorb_ref_events(prev);
    for (auto & a : alts)
      absorb_ref_events(a, prev);
    prev.alts.erase(std::remove_if(prev.alts.begin(), prev.alts.end(),
                                   [this](AltAllele const & pa)
                                   {
                                     for (long ae : pa.anti_events)
                                       if (ref_events.count(ae))
                                         return true;
                                     return false;
                                   }),
                    prev.alts.end());
    move_alts(std::move(prev.alts));
    move_alts(std::move(fresh));
  }

  void merge_all(VarRecord && prev) // var_record.cpp:202-268
  {
    if (prev.pos + prev.ref.size() != pos)
    {
      merge(std::move(prev), 0);
      return;
    }
    std::vector<AltAllele> fresh;
    for (auto const & prev_alt : prev.alts)
    {
      for (auto const & curr_alt : alts)
        if (is_ok_to_merge_alts(prev_alt, curr_alt))
          fresh.push_back(make_alt(prev_alt, curr_alt, 0));
      AltAllele n(prev_alt);
      n.seq += ref;
      n.events.insert(ref_events.begin(), ref_events.end());
      n.anti_events.insert(ref_anti_events.begin(), ref_anti_events.end());
      fresh.push_back(std::move(n));
    }
    for (auto & a : alts)
    {
      a.seq = prev.ref + a.seq;
      absorb_ref_events(a, prev);
    }
    pos = prev.pos;
    ref = prev.ref + ref;
    absorb_ref_events(prev);
    move_alts(std::move(fresh));
    alts.erase(std::remove_if(alts.begin(), alts.end(),
                              [](AltAllele const & a)
                              {
                                for (long ae : a.anti_events)
                                  if (a.events.count(ae))
                                    return true;
                                return false;
                              }),
               alts.end());
  }

  // var_record.cpp:381-406
  std::size_t common_suffix_size() const
  {
    if (ref.empty())
      return 0;
    for (auto const & a : alts)
      if (a.seq.empty())
        return 0;
    long n = 0;
    while (n < static_cast<long>(ref.size()) - 1)
    {
      char const c = ref[ref.size() - 1 - n];
      bool ok = true;
      for (auto const & a : alts)
        if (!(n < static_cast<long>(a.seq.size()) - 1 && a.seq[a.seq.size() - 1 - n] == c))
          ok = false;
      if (!ok)
        break;
      ++n;
    }
    return static_cast<std::size_t>(n);
  }
};

struct Location // include/graphtyper/graph/location.hpp
{
  char node_type = 'U';
  uint32_t node_index = 0, node_order = 0, offset = 0;
};

struct Path;

struct Graph
{
  bool is_sv_graph = false;
  bool is_segment_calling = false;
  bool add_all_variants = false; // Options::add_all_variants (include/graphtyper/utilities/options.hpp:77)
  long region_begin = 0, region_end = 0xFFFFFFFFl;
  std::vector<RefNode> ref_nodes;
  std::vector<VarNode> var_nodes;
  std::unordered_map<uint32_t, std::vector<uint32_t>> ref_reach_to_special_pos;
  std::vector<uint32_t> ref_reach_poses, actual_poses;

  // graph.cpp:41-339
  std::size_t reference_size = 0; // Graph::reference.size()
  void add_genomic_region(std::string const & reference, std::vector<VarRecord> records)
  {
    reference_size = reference.size();
    for (auto & r : records) // graph.cpp:48-59
      r.alts.erase(std::remove_if(r.alts.begin(), r.alts.end(),
                                  [](AltAllele const & a) { return a.seq.empty() || a.seq.find('N') != std::string::npos; }),
                   r.alts.end());
    records.erase(std::remove_if(records.begin(), records.end(), // graph.cpp:61-71
                                 [this](VarRecord const & r)
                                 {
                                   return r.ref.find('N') != std::string::npos || r.ref.find('*') != std::string::npos ||
                                          r.alts.empty() || static_cast<long>(r.pos) < region_begin;
                                 }),
                  records.end());
    for (std::size_t v = 0; v < records.size(); ++v) // graph.cpp:73-80
      if (static_cast<long>(records[v].pos) >= region_end)
      {
        records.resize(v);
        break;
      }
    long const n_rec = static_cast<long>(records.size());
    if (add_all_variants) // graph.cpp:81-167
    {
      long constexpr MAX_VAR_MERGE_DIST = 10, MAX_INDEL_MERGE_DIST = 2;
      for (long i = 0; i < n_rec; ++i)
        while (i + 1 < n_rec)
        {
          VarRecord & curr = records[i];
          VarRecord & next = records[i + 1];
          long const curr_end = static_cast<long>(curr.pos + curr.ref.size());
          if (static_cast<long>(next.pos) > curr_end + MAX_VAR_MERGE_DIST)
            break;
          if ((!curr.is_snp_or_snps() || !next.is_snp_or_snps()) && static_cast<long>(next.pos) > curr_end + MAX_INDEL_MERGE_DIST)
            break;
          if (static_cast<long>(next.pos) >= curr_end && (curr.alts.size() > 42 || next.alts.size() > 42 ||
                                                          curr.is_any_seq_larger_than(20) || next.is_any_seq_larger_than(20)))
            break;
          if ((curr.alts.size() + 1) * (next.alts.size() + 1) >= (MAX_NUMBER_OF_HAPLOTYPES - 1))
            next.merge_one_path(std::move(curr));
          else
          {
            if (static_cast<long>(next.pos) > curr_end)
            {
              long const a = std::min<long>(curr_end - region_begin, static_cast<long>(reference.size()));
              long const b = std::min<long>(static_cast<long>(next.pos) - region_begin, static_cast<long>(reference.size()));
              curr.add_suffix(reference.substr(a, b - a));
            }
            next.merge_all(std::move(curr));
          }
          if (next.alts.size() >= MAX_NUMBER_OF_HAPLOTYPES - 1)
            next.alts.resize(MAX_NUMBER_OF_HAPLOTYPES - 1);
          curr.clear();
          ++i;
        }
    }
    else // graph.cpp:169-240
    {
      for (long i = 0; i < n_rec; ++i)
        while (i + 1 < n_rec && records[i + 1].pos < records[i].pos + records[i].ref.size())
        {
          VarRecord & curr = records[i];
          VarRecord & next = records[i + 1];
          if (is_sv_graph && (curr.is_sv || next.is_sv))
          {
            if (curr.is_sv && next.is_sv)
              next.merge_one_path(std::move(curr));
            else if (curr.is_sv)
              next = std::move(curr); // the small variant overlapping an SV breakpoint is dropped
            // else: the previous (small) variant is dropped
          }
          else if (curr.alts.size() > 100 || (next.pos - curr.pos) < 4)
            next.merge_one_path(std::move(curr));
          else
            next.merge(std::move(curr), 4);
          curr.clear();
          ++i;
        }
    }
    for (auto & r : records) // graph.cpp:243-251
      r.alts.erase(std::remove_if(r.alts.begin(), r.alts.end(), [&](AltAllele const & a) { return a.seq == r.ref; }),
                   r.alts.end());
    records.erase(std::remove_if(records.begin(), records.end(), [](VarRecord const & r) { return r.alts.empty(); }),
                  records.end());
    for (auto & r : records) // graph.cpp:259-266
      if (r.alts.size() >= MAX_NUMBER_OF_HAPLOTYPES - 1)
        r.alts.resize(MAX_NUMBER_OF_HAPLOTYPES - 2);
    for (auto & r : records) // graph.cpp:268-285
    {
      std::size_t const n = r.common_suffix_size();
      if (n > 0)
      {
        r.ref.erase(r.ref.size() - n);
        for (auto & a : r.alts)
          a.seq.erase(a.seq.size() - n);
      }
    }
    for (auto & r : records) // graph.cpp:290-293, alt.cpp:46-49
      std::sort(r.alts.begin(), r.alts.end(), [](AltAllele const & a, AltAllele const & b) { return a.seq < b.seq; });
    for (auto & r : records) // graph.cpp:295-301
    {
      add_reference(r.pos, static_cast<unsigned>(r.alts.size()) + 1u, reference);
      add_variants(r);
    }
    add_reference(static_cast<uint32_t>(reference.size() + region_begin), 0, reference);
  }

  // graph.cpp:584-625
  void add_reference(unsigned end_pos, unsigned num_var, std::string const & reference)
  {
    if (end_pos > reference.size() + region_begin)
      end_pos = static_cast<unsigned>(reference.size() + region_begin);
    unsigned start_pos = static_cast<unsigned>(region_begin);
    if (!var_nodes.empty())
    {
      Label const & prev = var_nodes.at(ref_nodes.back().out_var_ids.at(0)).label;
      start_pos = prev.order - 1 + static_cast<unsigned>(prev.dna.size());
    }
    end_pos = std::max(start_pos, end_pos);
    long const a = std::min<long>(static_cast<long>(start_pos) - region_begin, static_cast<long>(reference.size()));
    long const b = std::min<long>(static_cast<long>(end_pos) - region_begin, static_cast<long>(reference.size()));
    RefNode rn;
    rn.label.order = start_pos + 1;
    rn.label.dna = reference.substr(a, b - a);
    rn.label.variant_num = 0;
    for (unsigned i = 0; i < num_var; ++i)
      rn.out_var_ids.push_back(i + static_cast<unsigned>(var_nodes.size()));
    ref_nodes.push_back(std::move(rn));
  }

  // graph.cpp:548-582
  void add_variants(VarRecord const & r)
  {
    VarNode ref_allele;
    ref_allele.label = Label{r.pos + 1, r.ref, 0};
    ref_allele.out_ref_id = static_cast<uint32_t>(ref_nodes.size());
    ref_allele.events = r.ref_events;
    ref_allele.anti_events = r.ref_anti_events;
    var_nodes.push_back(std::move(ref_allele));
    for (std::size_t i = 0; i < r.alts.size(); ++i)
    {
      VarNode alt;
      alt.label = Label{r.pos + 1, r.alts[i].seq, static_cast<uint32_t>(i + 1)};
      alt.out_ref_id = static_cast<uint32_t>(ref_nodes.size());
      alt.events = r.alts[i].events;
      alt.anti_events = r.alts[i].anti_events;
      var_nodes.push_back(std::move(alt));
    }
  }

  // graph.cpp:384-407 + 1759-1773
  void create_special_positions()
  {
    ref_reach_to_special_pos.clear();
    ref_reach_poses.clear();
    actual_poses.clear();
    for (std::size_t r = 0; r + 1 < ref_nodes.size(); ++r)
    {
      auto const & out = ref_nodes[r].out_var_ids;
      if (out.size() <= 1)
        continue;
      uint32_t const ref_reach = var_nodes[out[0]].label.reach();
      uint32_t max_reach = var_nodes[out[1]].label.reach();
      for (std::size_t i = 2; i < out.size(); ++i)
        max_reach = std::max(max_reach, var_nodes[out[i]].label.reach());
      for (uint32_t reach = ref_reach + 1; reach <= max_reach; ++reach)
      {
        ref_reach_poses.push_back(ref_reach);
        actual_poses.push_back(reach);
        ref_reach_to_special_pos[ref_reach].push_back(SPECIAL_START + static_cast<uint32_t>(ref_reach_poses.size()) - 1);
      }
    }
  }

  // graph.cpp:1775-1803
  uint32_t get_special_pos(uint32_t pos, uint32_t ref_reach) const
  {
    return ref_reach_to_special_pos.at(ref_reach).at(pos - ref_reach - 1);
  }
  bool is_special_pos(uint32_t pos) const { return pos >= SPECIAL_START && (pos - SPECIAL_START) < ref_reach_poses.size(); }
  uint32_t get_ref_reach_pos(uint32_t pos) const { return is_special_pos(pos) ? ref_reach_poses.at(pos - SPECIAL_START) : pos; }
  uint32_t get_actual_pos(uint32_t pos) const { return is_special_pos(pos) ? actual_poses.at(pos - SPECIAL_START) : pos; }

  // graph.cpp:341-351
  uint16_t get_variant_num(uint32_t v) const
  {
    return static_cast<uint16_t>(v - ref_nodes[var_nodes[v].out_ref_id - 1].out_var_ids[0]);
  }
  uint32_t get_variant_order(uint32_t v) const { return var_nodes[v].label.order; }

  // reference reach of the site a var node belongs to (the expression repeated at graph.cpp:1212,1276,...)
  uint32_t site_ref_reach(uint32_t v) const
  {
    return var_nodes[ref_nodes[var_nodes[v].out_ref_id - 1].out_var_ids[0]].label.reach();
  }

  // graph.cpp:353-376
  std::string get_all_ref() const
  {
    std::string out;
    if (ref_nodes.empty())
      return out;
    std::size_t r = 0, v = 0;
    while (!ref_nodes[r].out_var_ids.empty())
    {
      out += ref_nodes[r].label.dna;
      out += var_nodes[v].label.dna;
      v += ref_nodes[r].out_var_ids.size();
      ++r;
    }
    out += ref_nodes[r].label.dna;
    return out;
  }

  std::vector<Location> get_locations_of_a_position(uint32_t pos, Path const & path) const;
  std::vector<Location> get_locations_of_an_actual_position(uint32_t pos, Path const & path, bool is_special) const;
  std::vector<KmerLabel> get_labels_forward(Location const & s, std::string const & read, uint32_t & max_mismatches) const;
  std::vector<KmerLabel> get_labels_backward(Location const & e, std::string const & read, uint32_t & max_mismatches) const;
  std::vector<KmerLabel> iterative_dfs(std::vector<Location> const & starts, std::vector<Location> const & ends,
                                       std::string const & subread, uint32_t & max_mismatches) const;
};

// ---------------------------------------------------------------------------
// index  (src/index/*.cpp)
// ---------------------------------------------------------------------------
struct PHIndex // include/graphtyper/index/ph_index.hpp:14-36 ; std::unordered_map stands in for phmap (find/insert only)
{
  std::unordered_map<uint64_t, std::vector<KmerLabel>> hamming0;
  long max_index_labels = 75;

  std::vector<KmerLabel> get(uint64_t key) const // ph_index.cpp:24-32
  {
    auto it = hamming0.find(key);
    return it == hamming0.end() ? std::vector<KmerLabel>() : it->second;
  }

  // ph_index.cpp:34-64 (one list) and :66-107 (list of lists) share this rule
  std::vector<KmerLabel> get(std::vector<uint64_t> const & keys) const
  {
    std::vector<std::vector<KmerLabel> const *> results;
    long num_results = 0;
    long const NUM_KEYS = static_cast<long>(keys.size());
    for (long j = 0; j < NUM_KEYS; ++j)
    {
      auto it = hamming0.find(keys[j]);
      if (it == hamming0.end())
        continue;
      num_results += static_cast<long>(it->second.size());
      if (NUM_KEYS > 1 && num_results > max_index_labels)
      {
        results.clear();
        break;
      }
      results.push_back(&it->second);
    }
    std::vector<KmerLabel> labels;
    for (auto const * r : results)
      labels.insert(labels.end(), r->begin(), r->end());
    return labels;
  }

  std::vector<std::vector<KmerLabel>> multi_get(std::vector<std::vector<uint64_t>> const & keys) const
  {
    std::vector<std::vector<KmerLabel>> out;
    for (auto const & k : keys)
      out.push_back(get(k));
    return out;
  }

  // ph_index.cpp:145-237 : every 5th all-ACGT reference 32-mer must be found
  bool check(Graph const & graph) const
  {
    std::string const ref = graph.get_all_ref();
    if (ref.size() < K)
      return true;
    if (std::all_of(ref.begin(), ref.end(), [](char c) { return c == 'N'; }))
      return true;
    bool ok = true;
    for (std::size_t s = 0; ref.size() - (s + K) >= 5 && s + K <= ref.size(); s += 5)
    {
      std::string const kmer = ref.substr(s, K);
      if (kmer.find_first_not_of("ACGT") != std::string::npos)
        continue;
      if (get(to_uint64(kmer)).empty())
        ok = false;
    }
    return ok;
  }
};

struct IndexEntry // include/graphtyper/index/index_entry.hpp:18-36 (`valid` is never set on reachable paths)
{
  uint64_t dna = 0;
  uint32_t start_index = 0;
  std::set<uint32_t> variant_id;
  uint32_t total_var_num = 1, total_var_count = 0;
  std::unordered_set<long> events, anti_events;
  void add_to_dna(char b) { dna = (dna << 2) + base_to_u64(b); } // index_entry.cpp:20-54
};

using EntrySublist = std::deque<IndexEntry>;
using EntryList = std::deque<EntrySublist>;

inline bool is_acgt(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

// indexer.cpp:26-80
inline void index_reference_label(PHIndex & idx, EntryList & mers, Label const & label)
{
  for (std::size_t d = 0; d < label.dna.size(); ++d)
  {
    char const b = label.dna[d];
    if (!is_acgt(b))
    {
      mers.clear();
      continue;
    }
    for (auto & sub : mers)
      for (auto & e : sub)
        e.add_to_dna(b);
    IndexEntry fresh;
    fresh.start_index = label.order + static_cast<uint32_t>(d);
    fresh.add_to_dna(b);
    mers.push_front(EntrySublist(1, fresh));
    if (mers.size() >= K)
    {
      uint32_t const end = label.order + static_cast<uint32_t>(d);
      for (auto const & e : mers.back())
      {
        if (e.variant_id.empty())
          idx.hamming0[e.dna].push_back(KmerLabel(e.start_index, end));
        else
          for (uint32_t v : e.variant_id)
            idx.hamming0[e.dna].push_back(KmerLabel(e.start_index, end, v));
      }
      mers.pop_back();
    }
  }
}

// indexer.cpp:82-178
inline void insert_variant_label(Graph const & graph, PHIndex & idx, EntryList & mers, uint32_t v, bool is_reference,
                                 unsigned var_count, uint32_t ref_reach)
{
  VarNode const & var = graph.var_nodes[v];
  Label const & label = var.label;
  for (std::size_t d = 0; d < label.dna.size(); ++d)
  {
    char const b = label.dna[d];
    if (!is_acgt(b))
    {
      mers.clear();
      continue;
    }
    for (auto & sub : mers)
    {
      for (auto it = sub.begin(); it != sub.end();)
      {
        bool ok = true;
        for (long a : it->anti_events)
          if (var.events.count(a))
          {
            ok = false;
            break;
          }
        if (!ok)
        {
          it = sub.erase(it);
          continue;
        }
        it->add_to_dna(b);
        it->events.insert(var.events.begin(), var.events.end());
        it->anti_events.insert(var.anti_events.begin(), var.anti_events.end());
        it->variant_id.insert(v);
        ++it;
      }
    }
    uint32_t pos = label.order + static_cast<uint32_t>(d);
    if (pos > ref_reach)
      pos = graph.get_special_pos(pos, ref_reach);
    IndexEntry fresh;
    fresh.start_index = pos;
    fresh.variant_id.insert(v);
    fresh.total_var_num = var_count;
    fresh.total_var_count = is_reference ? 0u : 1u;
    fresh.add_to_dna(b);
    fresh.events = var.events;
    fresh.anti_events = var.anti_events;
    mers.push_front(EntrySublist(1, fresh));
    if (mers.size() >= K)
    {
      for (auto const & e : mers.back())
        for (uint32_t id : e.variant_id)
          idx.hamming0[e.dna].push_back(KmerLabel(e.start_index, pos, id));
      mers.pop_back();
    }
  }
}

// indexer.cpp:180-196
inline void append_list(EntryList & mers, EntryList && list)
{
  if (mers.size() < list.size())
    mers.resize(list.size());
  auto m = mers.begin();
  for (auto l = list.begin(); l != list.end(); ++l, ++m)
    for (auto & e : *l)
      m->push_back(std::move(e));
}

// indexer.cpp:13-20 + 198-211
inline void remove_large_variants_from_list(EntryList & list, unsigned var_count)
{
  for (auto & sub : list)
  {
    for (auto & e : sub)
    {
      e.total_var_num *= var_count;
      ++e.total_var_count;
    }
    sub.erase(std::remove_if(sub.begin(), sub.end(),
                             [](IndexEntry const & e)
                             { return e.total_var_count > 1 && (e.total_var_num > 181u || e.total_var_count > 4u); }),
              sub.end());
  }
}

// indexer.cpp:213-244
inline void index_variant(Graph const & graph, PHIndex & idx, EntryList & mers, unsigned var_count, uint32_t v)
{
  EntryList clean_list(mers);
  uint32_t const ref_reach = graph.var_nodes[v].label.reach();
  insert_variant_label(graph, idx, mers, v, true, 1, ref_reach);
  remove_large_variants_from_list(clean_list, var_count);
  unsigned const var_num = var_count;
  while (var_count > 2)
  {
    --var_count;
    ++v;
    EntryList new_list(clean_list);
    insert_variant_label(graph, idx, new_list, v, false, var_num, ref_reach);
    append_list(mers, std::move(new_list));
  }
  ++v;
  insert_variant_label(graph, idx, clean_list, v, false, var_num, ref_reach);
  append_list(mers, std::move(clean_list));
}

// indexer.cpp:246-291
inline PHIndex index_graph(Graph const & graph, long max_index_labels = 75)
{
  PHIndex idx;
  idx.max_index_labels = max_index_labels;
  EntryList mers;
  if (graph.ref_nodes.empty())
    return idx;
  for (std::size_t r = 0; r + 1 < graph.ref_nodes.size(); ++r)
  {
    index_reference_label(idx, mers, graph.ref_nodes[r].label);
    if (!graph.ref_nodes[r].out_var_ids.empty())
      index_variant(graph, idx, mers, static_cast<unsigned>(graph.ref_nodes[r].out_var_ids.size()),
                    graph.ref_nodes[r].out_var_ids[0]);
  }
  index_reference_label(idx, mers, graph.ref_nodes.back().label);
  return idx;
}

// kmer_help_functions.cpp:53-63
inline std::vector<std::vector<KmerLabel>> query_index(std::vector<uint8_t> const & read, PHIndex const & idx)
{
  std::vector<std::vector<uint64_t>> keys;
  for (std::size_t i = 0, n = get_num_kmers(read.size()); i < n; ++i)
    keys.push_back(to_uint64_vec(read, (K - 1) * i));
  return idx.multi_get(keys);
}

// kmer_help_functions.cpp:97-119
inline std::vector<std::vector<KmerLabel>> query_index_hamming1(std::vector<uint8_t> const & read, PHIndex const & idx)
{
  std::vector<std::vector<uint64_t>> keys;
  std::size_t const n = get_num_kmers(read.size());
  for (std::size_t i = 0; i < n; ++i)
    keys.push_back(to_uint64_vec(read, (K - 1) * i));
  for (std::size_t i = 0; i < n; ++i)
  {
    if (keys[i].size() != 1)
      continue;
    auto const h = hamming1_keys(keys[i][0]);
    keys[i].assign(h.begin(), h.end());
  }
  return idx.multi_get(keys);
}

// ---------------------------------------------------------------------------
// Path  (include/graphtyper/typer/path.hpp, src/typer/path.cpp)
// ---------------------------------------------------------------------------
struct Path
{
  uint32_t start = 0, end = 0;
  uint16_t read_start_index = 0, read_end_index = 0;
  std::vector<uint32_t> var_order;
  std::vector<std::set<uint16_t>> nums; // phmap::flat_hash_set in the reference; only membership/size/insert are used
  uint16_t mismatches = 0;

  Path() = default;
  Path(Graph const & g, KmerLabel const & l, uint16_t rs, uint16_t re, uint16_t mm) // path.cpp:13-36
    : start(l.start_index), end(l.end_index), read_start_index(rs), read_end_index(re), mismatches(mm)
  {
    if (l.variant_id != INVALID_ID)
    {
      var_order.push_back(g.get_variant_order(l.variant_id));
      nums.push_back({g.get_variant_num(l.variant_id)});
    }
  }

  // path.cpp:38-82 : everything from p2, allele sets intersected with p1; returns early (half merged) on an empty set
  Path(Path const & p1, Path const & p2) : Path(p2)
  {
    for (std::size_t i = 0; i < p1.var_order.size(); ++i)
    {
      bool found = false;
      for (std::size_t j = 0; j < var_order.size(); ++j)
      {
        if (p1.var_order[i] != var_order[j])
          continue;
        for (auto it = nums[j].begin(); it != nums[j].end();)
          it = p1.nums[i].count(*it) ? std::next(it) : nums[j].erase(it);
        if (nums[j].empty())
          return;
        found = true;
        break;
      }
      if (!found)
      {
        var_order.push_back(p1.var_order[i]);
        nums.push_back(p1.nums[i]);
      }
    }
    read_start_index = p1.read_start_index;
    start = p1.start;
    mismatches = static_cast<uint16_t>(mismatches + p1.mismatches);
  }

  void merge_with_current(Graph const & g, KmerLabel const & l) // path.cpp:105-129
  {
    if (l.variant_id == INVALID_ID)
      return;
    uint32_t const order = g.get_variant_order(l.variant_id);
    uint16_t const num = g.get_variant_num(l.variant_id);
    for (std::size_t i = 0; i < var_order.size(); ++i)
      if (var_order[i] == order)
      {
        nums[i].insert(num);
        return;
      }
    var_order.push_back(order);
    nums.push_back({num});
  }

  uint32_t size() const { return read_end_index - read_start_index + 1u; } // path.cpp:165-169
  bool is_reference() const // path.cpp:176-185
  {
    for (auto const & n : nums)
      if (!n.count(0))
        return false;
    return true;
  }
  bool is_purely_reference() const
  {
    for (auto const & n : nums)
      if (!n.count(0) || n.size() > 1)
        return false;
    return true;
  }
  bool is_empty() const { return start == end; } // path.cpp:198-201
};

// ---------------------------------------------------------------------------
// graph walks  (src/graph/graph.cpp, include/graphtyper/graph/graph_utils.hpp)
// ---------------------------------------------------------------------------

// graph_utils.hpp:7-37 with read_offset = dna_index = 0
inline uint32_t count_mismatches(std::string const & read, std::string const & dna, uint32_t max_mismatches)
{
  uint32_t mm = 0;
  for (std::size_t i = 0; i < dna.size() && i < read.size(); ++i)
  {
    if (dna[i] == '>' || dna[i] == '<')
      return max_mismatches + 1;
    if (dna[i] != read[i] && read[i] != 'N' && dna[i] != 'N')
      if (++mm > max_mismatches)
        return mm;
  }
  return mm;
}

// graph_utils.hpp:39-69
inline uint32_t count_mismatches_backward(std::string const & read, std::string const & dna, uint32_t max_mismatches)
{
  uint32_t mm = 0;
  for (std::size_t i = 0; i < dna.size() && i < read.size(); ++i)
  {
    char const g = dna[dna.size() - 1 - i], r = read[read.size() - 1 - i];
    if (g == '>' || g == '<')
      return max_mismatches + 1;
    if (g != r && r != 'N' && g != 'N')
      if (++mm > max_mismatches)
        return mm;
  }
  return mm;
}

// graph.cpp:931-1029
inline std::vector<Location> Graph::get_locations_of_an_actual_position(uint32_t pos, Path const & path, bool is_special) const
{
  std::vector<Location> locs;
  if (pos < ref_nodes[0].label.order)
    return locs;
  if (ref_nodes.size() == 1)
  {
    locs.push_back({'R', 0, ref_nodes[0].label.order, pos - ref_nodes[0].label.order});
    return locs;
  }
  for (uint32_t r = 1; r <= ref_nodes.size(); ++r)
  {
    if (r < ref_nodes.size() && ref_nodes[r].label.order <= pos)
      continue;
    int rr = static_cast<int>(r) - 1;
    if (pos < ref_nodes[rr].label.order + ref_nodes[rr].label.dna.size())
    {
      if (!is_special)
      {
        locs.push_back({'R', static_cast<uint32_t>(rr), ref_nodes[rr].label.order, pos - ref_nodes[rr].label.order});
        break;
      }
      --rr;
    }
    long const PADDING = (is_sv_graph || is_segment_calling) ? 1000000 : 1000;
    while (rr >= 0 && static_cast<long>(ref_nodes[rr].label.reach()) + PADDING > static_cast<long>(pos))
    {
      for (int i = 0; i < static_cast<int>(ref_nodes[rr].out_var_ids.size()); ++i)
      {
        uint32_t const v = ref_nodes[rr].out_var_ids[i];
        Label const & l = var_nodes[v].label;
        if (pos >= l.order && pos <= l.reach())
        {
          auto it = std::find(path.var_order.begin(), path.var_order.end(), l.order);
          if (it == path.var_order.end())
            continue;
          long const j = it - path.var_order.begin();
          if (path.is_empty() || (j < static_cast<long>(path.nums.size()) && path.nums[j].count(static_cast<uint16_t>(i))))
            locs.push_back({'V', v, l.order, pos - l.order});
        }
      }
      --rr;
    }
    break;
  }
  return locs;
}

// graph.cpp:1154-1185
inline std::vector<Location> Graph::get_locations_of_a_position(uint32_t pos, Path const & path) const
{
  bool const special = is_special_pos(pos);
  if (special)
    pos = actual_poses.at(pos - SPECIAL_START);
  return get_locations_of_an_actual_position(pos, path, special);
}

// graph.cpp:1187-1439.  All position arithmetic is modulo 2^32 as in the reference.
inline std::vector<KmerLabel> Graph::get_labels_forward(Location const & s, std::string const & read,
                                                        uint32_t & max_mismatches) const
{
  std::vector<KmerLabel> labels;
  std::vector<std::string> seqs(1);
  std::vector<std::vector<uint32_t>> var_ids(1);
  std::vector<uint32_t> end_pos(1, 0u);
  std::vector<uint32_t> vars;
  uint32_t const L = static_cast<uint32_t>(read.size());
  auto special_end = [this](uint32_t e, uint32_t v)
  {
    uint32_t const rr = site_ref_reach(v);
    return e > rr ? get_special_pos(e, rr) : e;
  };

  if (s.node_type == 'V')
  {
    VarNode const & var = var_nodes[s.node_index];
    var_ids[0].push_back(s.node_index);
    seqs[0] = var.label.dna.substr(s.offset);
    if (seqs[0].size() >= L)
      end_pos[0] = special_end(var.label.reach() - (static_cast<uint32_t>(seqs[0].size()) - L), s.node_index);
    else
    {
      RefNode const & ref = ref_nodes[var.out_ref_id];
      vars = ref.out_var_ids;
      seqs[0] += ref.label.dna;
      end_pos[0] = ref.label.reach() - (static_cast<uint32_t>(seqs[0].size()) - L);
    }
  }
  else
  {
    RefNode const & ref = ref_nodes[s.node_index];
    vars = ref.out_var_ids;
    seqs[0] = ref.label.dna.substr(s.offset);
    end_pos[0] = ref.label.reach() - (static_cast<uint32_t>(seqs[0].size()) - L);
  }

  if (!vars.empty() && seqs[0].size() < L)
  {
    uint32_t r = var_nodes[vars[0]].out_ref_id;
    bool all_long = false;
    while (!all_long && seqs.size() < 128 && !vars.empty())
    {
      all_long = true;
      RefNode const & ref = ref_nodes[r];
      std::size_t original_size = seqs.size();
      for (std::size_t j = 0; j < original_size; ++j)
      {
        if (seqs[j].size() >= L)
          continue;
        for (std::size_t i = 0; i + 1 < vars.size(); ++i)
        {
          VarNode const & var = var_nodes[vars[i]];
          std::string nseq = seqs[j] + var.label.dna;
          bool const enough = nseq.size() >= L;
          if (!enough)
            nseq += ref.label.dna;
          if (count_mismatches(read, nseq, max_mismatches) <= max_mismatches)
          {
            std::vector<uint32_t> ids(var_ids[j]);
            ids.push_back(vars[i]);
            var_ids.push_back(std::move(ids));
            if (nseq.size() < L)
              all_long = false;
            if (enough)
              end_pos.push_back(special_end(var.label.reach() - (static_cast<uint32_t>(nseq.size()) - L), vars[i]));
            else
              end_pos.push_back(ref.label.reach() - (static_cast<uint32_t>(nseq.size()) - L));
            seqs.push_back(std::move(nseq));
          }
        }
        VarNode const & var = var_nodes[vars.back()];
        seqs[j] += var.label.dna;
        bool const enough = seqs[j].size() >= L;
        if (!enough)
          seqs[j] += ref.label.dna;
        if (count_mismatches(read, seqs[j], max_mismatches) <= max_mismatches)
        {
          var_ids[j].push_back(vars.back());
          if (seqs[j].size() < L)
            all_long = false;
          if (enough)
            end_pos[j] = special_end(var.label.reach() - (static_cast<uint32_t>(seqs[j].size()) - L), vars.back());
          else
            end_pos[j] = ref.label.reach() - (static_cast<uint32_t>(seqs[j].size()) - L);
        }
        else
        {
          seqs.erase(seqs.begin() + j);
          var_ids.erase(var_ids.begin() + j);
          end_pos.erase(end_pos.begin() + j);
          --original_size;
          --j;
        }
      }
      if (all_long)
        break;
      vars = ref_nodes[r].out_var_ids;
      ++r;
    }
  }

  std::vector<std::size_t> best;
  for (std::size_t j = 0; j < seqs.size(); ++j)
  {
    if (seqs[j].size() < L)
      continue;
    uint32_t const mm = count_mismatches(read, seqs[j], max_mismatches);
    if (mm > max_mismatches)
      continue;
    if (mm < max_mismatches)
    {
      max_mismatches = mm;
      best.clear();
    }
    best.push_back(j);
  }
  for (std::size_t j : best)
  {
    uint32_t start_pos = s.node_order + s.offset;
    if (s.node_type == 'V')
      start_pos = special_end(start_pos, s.node_index);
    if (var_ids[j].empty())
      labels.push_back(KmerLabel(start_pos, end_pos[j]));
    else
      for (uint32_t v : var_ids[j])
        labels.push_back(KmerLabel(start_pos, end_pos[j], v));
  }
  return labels;
}

// graph.cpp:1441-1701
inline std::vector<KmerLabel> Graph::get_labels_backward(Location const & e, std::string const & read,
                                                         uint32_t & max_mismatches) const
{
  std::vector<KmerLabel> labels;
  std::vector<std::string> seqs(1);
  std::vector<std::vector<uint32_t>> var_ids(1);
  std::vector<uint32_t> start_pos(1, 0u);
  std::vector<uint32_t> vars;
  uint32_t const L = static_cast<uint32_t>(read.size());
  auto special_of = [this](uint32_t p, uint32_t v)
  {
    uint32_t const rr = site_ref_reach(v);
    return p > rr ? get_special_pos(p, rr) : p;
  };

  if (e.node_type == 'V')
  {
    VarNode const & var = var_nodes[e.node_index];
    var_ids[0].push_back(e.node_index);
    seqs[0] = var.label.dna.substr(0, e.offset + 1);
    if (seqs[0].size() >= L)
      start_pos[0] = special_of(var.label.order + (static_cast<uint32_t>(seqs[0].size()) - L), e.node_index);
    else
    {
      uint32_t const r = var.out_ref_id - 1;
      RefNode const & ref = ref_nodes[r];
      seqs[0] = ref.label.dna + seqs[0];
      start_pos[0] = ref.label.order + (static_cast<uint32_t>(seqs[0].size()) - L);
      if (r != 0)
        vars = ref_nodes[r - 1].out_var_ids;
    }
  }
  else
  {
    RefNode const & ref = ref_nodes[e.node_index];
    if (e.node_index != 0)
      vars = ref_nodes[e.node_index - 1].out_var_ids;
    seqs[0] = ref.label.dna.substr(0, e.offset + 1);
    start_pos[0] = ref.label.order + (static_cast<uint32_t>(seqs[0].size()) - L);
  }

  if (!vars.empty() && seqs[0].size() < L)
  {
    uint32_t r = var_nodes[vars[0]].out_ref_id - 1;
    bool all_long = false;
    while (!all_long && seqs.size() < 128 && !vars.empty())
    {
      all_long = true;
      RefNode const & ref = ref_nodes[r];
      std::size_t original_size = seqs.size();
      for (std::size_t j = 0; j < original_size; ++j)
      {
        if (seqs[j].size() >= L)
          continue;
        for (std::size_t i = 0; i + 1 < vars.size(); ++i)
        {
          VarNode const & var = var_nodes[vars[i]];
          std::string nseq = var.label.dna + seqs[j];
          bool const enough = nseq.size() >= L;
          if (!enough)
            nseq = ref.label.dna + nseq;
          if (count_mismatches_backward(read, nseq, max_mismatches) <= max_mismatches)
          {
            std::vector<uint32_t> ids(var_ids[j]);
            ids.push_back(vars[i]);
            var_ids.push_back(std::move(ids));
            if (nseq.size() < L)
              all_long = false;
            if (enough)
              start_pos.push_back(special_of(var.label.order + (static_cast<uint32_t>(nseq.size()) - L), vars[i]));
            else
              start_pos.push_back(ref.label.order + (static_cast<uint32_t>(nseq.size()) - L));
            seqs.push_back(std::move(nseq));
          }
        }
        VarNode const & var = var_nodes[vars.back()];
        seqs[j] = var.label.dna + seqs[j];
        bool const enough = seqs[j].size() >= L;
        if (!enough)
          seqs[j] = ref.label.dna + seqs[j];
        if (count_mismatches_backward(read, seqs[j], max_mismatches) <= max_mismatches)
        {
          var_ids[j].push_back(vars.back());
          if (seqs[j].size() < L)
            all_long = false;
          if (enough)
            start_pos[j] = special_of(var.label.order + (static_cast<uint32_t>(seqs[j].size()) - L), vars.back());
          else
            start_pos[j] = ref.label.order + (static_cast<uint32_t>(seqs[j].size()) - L);
        }
        else
        {
          seqs.erase(seqs.begin() + j);
          var_ids.erase(var_ids.begin() + j);
          start_pos.erase(start_pos.begin() + j);
          --original_size;
          --j;
        }
      }
      if (all_long)
        break;
      if (r == 0)
      {
        vars.clear();
        break;
      }
      --r;
      vars = ref_nodes[r].out_var_ids;
    }
  }

  std::vector<std::size_t> best;
  for (std::size_t j = 0; j < seqs.size(); ++j)
  {
    if (seqs[j].size() < L)
      continue;
    uint32_t const mm = count_mismatches_backward(read, seqs[j], max_mismatches);
    if (mm < max_mismatches)
    {
      max_mismatches = mm;
      best.clear();
      best.push_back(j);
    }
    else if (mm == max_mismatches)
      best.push_back(j);
  }
  for (std::size_t j : best)
  {
    uint32_t end_pos = e.node_order + e.offset;
    if (e.node_type == 'V')
      end_pos = special_of(end_pos, e.node_index);
    if (var_ids[j].empty())
      labels.push_back(KmerLabel(start_pos[j], end_pos));
    else
      for (uint32_t v : var_ids[j])
        labels.push_back(KmerLabel(start_pos[j], end_pos, v));
  }
  return labels;
}

// graph.cpp:1703-1754
inline std::vector<KmerLabel> Graph::iterative_dfs(std::vector<Location> const & starts, std::vector<Location> const & ends,
                                                   std::string const & subread, uint32_t & max_mismatches) const
{
  std::vector<KmerLabel> labels;
  if (starts.size() > 1024 || ends.size() > 1024)
    return labels;
  auto add_if_better = [&](std::vector<KmerLabel> && nl, uint32_t mm)
  {
    if (nl.empty())
      return;
    if (mm < max_mismatches)
    {
      max_mismatches = mm;
      labels = std::move(nl);
    }
    else if (mm == max_mismatches)
      labels.insert(labels.end(), nl.begin(), nl.end());
  };
  if (starts.size() == 1 && starts[0].node_type == 'U')
    for (auto const & e : ends)
    {
      uint32_t mm = max_mismatches;
      auto nl = get_labels_backward(e, subread, mm);
      add_if_better(std::move(nl), mm);
    }
  else
    for (auto const & s : starts)
    {
      uint32_t mm = max_mismatches;
      auto nl = get_labels_forward(s, subread, mm);
      add_if_better(std::move(nl), mm);
    }
  return labels;
}

// ---------------------------------------------------------------------------
// GenotypePaths  (src/typer/genotype_paths.cpp)
// ---------------------------------------------------------------------------
struct GenotypePaths
{
  std::string read2;
  std::vector<Path> paths;
  uint16_t read_length = 0;
  uint16_t flags = 0;
  uint32_t longest_path_length = 0;
  uint32_t original_pos = 0;
  uint8_t score_diff = 0;
  uint8_t mapq = 255;
  int32_t ml_insert_size = INSERT_SIZE_WHEN_NOT_PROPER_PAIR;

  GenotypePaths() = default;
  GenotypePaths(uint16_t f, std::size_t len) : read_length(static_cast<uint16_t>(len)), flags(f) {}

  // genotype_paths.cpp:32-66
  static std::vector<Path> nonduplicated(Graph const & g, std::vector<KmerLabel> const & ll, uint32_t rs, uint32_t re, uint16_t mm)
  {
    std::vector<Path> out;
    for (auto const & l : ll)
    {
      bool merged = false;
      for (auto & p : out)
        if (l.start_index == p.start && l.end_index == p.end)
        {
          p.merge_with_current(g, l);
          merged = true;
          break;
        }
      if (!merged)
        out.push_back(Path(g, l, static_cast<uint16_t>(rs), static_cast<uint16_t>(re), mm));
    }
    return out;
  }

  // genotype_paths.cpp:294-352
  void add_next_kmer_labels(Graph const & g, std::vector<KmerLabel> const & ll, uint32_t rs, uint32_t re, int mm)
  {
    std::vector<Path> const pp = nonduplicated(g, ll, rs, re, static_cast<uint16_t>(mm));
    std::size_t const original_size = paths.size();
    std::vector<uint8_t> matched(pp.size(), 0);
    for (std::size_t i = 0; i < original_size; ++i)
    {
      if (paths[i].read_end_index != rs)
        continue;
      bool once = false;
      Path const original = paths[i];
      for (std::size_t j = 0; j < pp.size(); ++j)
      {
        if (!(original.end == pp[j].start && original.read_end_index == pp[j].read_start_index))
          continue;
        Path np(original, pp[j]);
        if (np.start != original.start || np.read_start_index != original.read_start_index)
          continue;
        matched[j] = 1;
        if (once)
          paths.push_back(std::move(np));
        else
        {
          longest_path_length = std::max(np.size(), longest_path_length);
          paths[i] = std::move(np);
          once = true;
        }
      }
    }
    for (std::size_t j = 0; j < pp.size(); ++j)
      if (!matched[j])
      {
        longest_path_length = std::max(pp[j].size(), longest_path_length);
        paths.push_back(pp[j]);
      }
  }

  // genotype_paths.cpp:233-292
  void add_prev_kmer_labels(Graph const & g, std::vector<KmerLabel> const & ll, uint32_t rs, uint32_t re, int mm)
  {
    std::vector<Path> const pp = nonduplicated(g, ll, rs, re, static_cast<uint16_t>(mm));
    std::size_t const original_size = paths.size();
    std::vector<uint8_t> matched(pp.size(), 0);
    for (std::size_t i = 0; i < original_size; ++i)
    {
      if (paths[i].read_start_index != re)
        continue;
      bool once = false;
      Path const original(paths[i]);
      for (std::size_t j = 0; j < pp.size(); ++j)
      {
        if (!(pp[j].end == original.start && pp[j].read_end_index == original.read_start_index))
          continue;
        Path np(pp[j], original);
        if (np.read_start_index != pp[j].read_start_index)
          continue;
        matched[j] = 1;
        if (once)
          paths.push_back(std::move(np));
        else
        {
          longest_path_length = std::max(np.size(), longest_path_length);
          paths[i] = std::move(np);
          once = true;
        }
      }
    }
    for (std::size_t j = 0; j < pp.size(); ++j)
      if (!matched[j])
      {
        longest_path_length = std::max(pp[j].size(), longest_path_length);
        paths.push_back(pp[j]);
      }
  }

  void remove_short_paths() // genotype_paths.cpp:824-834
  {
    if (longest_path_length <= 1)
      return;
    paths.erase(std::remove_if(paths.begin(), paths.end(), [&](Path const & p) { return p.size() < longest_path_length; }),
                paths.end());
  }

  void update_longest_path_size() // genotype_paths.cpp:858-864
  {
    longest_path_length = 0;
    for (auto const & p : paths)
      longest_path_length = std::max(p.size(), longest_path_length);
  }

  void remove_paths_with_too_many_mismatches() // genotype_paths.cpp:360-380
  {
    if (paths.empty())
      return;
    uint16_t mn = 10;
    for (auto const & p : paths)
      mn = std::min(p.mismatches, mn);
    paths.erase(std::remove_if(paths.begin(), paths.end(), [mn](Path const & p) { return p.mismatches > mn; }), paths.end());
  }

  bool all_paths_unique(Graph const & g) const // genotype_paths.cpp:219-231
  {
    for (std::size_t i = 1; i < paths.size(); ++i)
      if (g.get_ref_reach_pos(paths[0].start) != g.get_ref_reach_pos(paths[i].start) &&
          g.get_ref_reach_pos(paths[0].end) != g.get_ref_reach_pos(paths[i].end))
        return false;
    return true;
  }

  bool all_paths_fully_aligned() const // genotype_paths.cpp:836-845
  {
    for (auto const & p : paths)
      if (p.size() != read_length)
        return false;
    return true;
  }

  void remove_non_ref_paths_when_read_matches_ref(Graph const & g) // genotype_paths.cpp:460-474
  {
    if (all_paths_unique(g))
      return;
    if (std::any_of(paths.begin(), paths.end(), [](Path const & p) { return p.is_reference(); }))
      paths.erase(std::remove_if(paths.begin(), paths.end(), [](Path const & p) { return !p.is_reference(); }), paths.end());
  }

  void remove_fully_special_paths(Graph const & g) // genotype_paths.cpp:476-481
  {
    paths.erase(std::remove_if(paths.begin(), paths.end(),
                               [&](Path const & p) { return g.get_ref_reach_pos(p.start) == g.get_ref_reach_pos(p.end); }),
                paths.end());
  }

  void remove_support_from_read_ends(Graph const & g) // genotype_paths.cpp:382-432
  {
    long constexpr MIN_OFFSET = 4;
    for (Path & p : paths)
    {
      if (p.var_order.empty())
        continue;
      if (!g.is_special_pos(p.start) && !g.is_special_pos(p.end))
        continue;
      auto mm = std::minmax_element(p.var_order.begin(), p.var_order.end());
      if (g.is_special_pos(p.end) && static_cast<long>(g.get_actual_pos(p.end)) <= static_cast<long>(*mm.second) + MIN_OFFSET)
        p.nums[mm.second - p.var_order.begin()].clear();
      if (g.is_special_pos(p.start))
      {
        bool ambiguous = true;
        if (g.is_special_pos(p.start + static_cast<uint32_t>(MIN_OFFSET)))
          ambiguous = g.get_ref_reach_pos(p.start) != g.get_ref_reach_pos(p.start + static_cast<uint32_t>(MIN_OFFSET));
        if (ambiguous)
          p.nums[mm.first - p.var_order.begin()].clear();
      }
    }
  }

  // genotype_paths.cpp:483-553 ; seq holds IUPAC characters
  void walk_read_ends(std::string const & seq, int maximum_mismatches, Graph const & g)
  {
    if (paths.empty() || paths[0].size() == seq.size())
      return;
    if (paths.size() > MAX_SEED_NUMBER_FOR_WALKING)
      return;
    if (paths.size() > MAX_SEED_NUMBER_ALLOWING_MISMATCHES)
      maximum_mismatches = 0;
    std::size_t best_mismatches = 7;
    std::vector<uint32_t> best_idx;
    std::vector<std::vector<KmerLabel>> best_labels;
    for (auto & path : paths)
    {
      if (path.read_end_index == seq.size() - 1)
        continue;
      std::vector<Location> s_locs = g.get_locations_of_a_position(path.end, path);
      if (s_locs.empty() || s_locs.size() > MAX_NUM_LOCATIONS_PER_PATH)
        continue;
      std::string const kmer = seq.substr(path.read_end_index);
      std::vector<Location> e_locs(1);
      uint32_t mm = maximum_mismatches < 0 ? static_cast<uint32_t>(std::min(2 + kmer.size() / 11, best_mismatches))
                                           : static_cast<uint32_t>(maximum_mismatches);
      std::vector<KmerLabel> nl = g.iterative_dfs(s_locs, e_locs, kmer, mm);
      if (nl.empty())
        continue;
      if (mm < best_mismatches)
      {
        best_labels.clear();
        best_idx.clear();
        best_mismatches = mm;
      }
      if (mm == best_mismatches)
      {
        best_labels.push_back(std::move(nl));
        best_idx.push_back(path.read_end_index);
      }
    }
    for (std::size_t i = 0; i < best_labels.size(); ++i)
      add_next_kmer_labels(g, best_labels[i], best_idx[i], static_cast<uint32_t>(seq.size()) - 1, static_cast<int>(best_mismatches));
  }

  // genotype_paths.cpp:555-621
  void walk_read_starts(std::string const & seq, int maximum_mismatches, Graph const & g)
  {
    if (paths.empty() || paths[0].size() == seq.size())
      return;
    if (paths.size() > MAX_SEED_NUMBER_FOR_WALKING)
      return;
    if (paths.size() > MAX_SEED_NUMBER_ALLOWING_MISMATCHES)
      maximum_mismatches = 0;
    std::size_t best_mismatches = 7;
    std::vector<uint32_t> best_idx;
    std::vector<std::vector<KmerLabel>> best_labels;
    for (auto & path : paths)
    {
      if (path.read_start_index == 0)
        continue;
      std::string const kmer = seq.substr(0, path.read_start_index + 1u);
      std::vector<Location> e_locs = g.get_locations_of_a_position(path.start, path);
      if (e_locs.empty() || e_locs.size() > MAX_NUM_LOCATIONS_PER_PATH)
        continue;
      std::vector<Location> s_locs(1);
      uint32_t mm = maximum_mismatches < 0 ? static_cast<uint32_t>(std::min(2 + kmer.size() / 11, best_mismatches))
                                           : static_cast<uint32_t>(maximum_mismatches);
      std::vector<KmerLabel> nl = g.iterative_dfs(s_locs, e_locs, kmer, mm);
      if (nl.empty())
        continue;
      if (mm < best_mismatches)
      {
        best_labels.clear();
        best_idx.clear();
        best_mismatches = mm;
      }
      if (mm == best_mismatches)
      {
        best_labels.push_back(std::move(nl));
        best_idx.push_back(path.read_start_index);
      }
    }
    for (std::size_t i = 0; i < best_labels.size(); ++i)
      add_prev_kmer_labels(g, best_labels[i], 0, best_idx[i], static_cast<int>(best_mismatches));
  }

  bool is_proper_pair() const { return ml_insert_size != INSERT_SIZE_WHEN_NOT_PROPER_PAIR; }
};

// src/typer/alignment.cpp:23-103 ; read = 4-bit IUPAC codes
inline void find_genotype_paths_of_one_of_the_sequences(std::vector<uint8_t> const & read, GenotypePaths & geno,
                                                        PHIndex const & idx, Graph const & g)
{
  auto const r0 = query_index(read, idx);
  auto const r1 = query_index_hamming1(read, idx);
  bool any_small = false;
  for (auto const & l : r0)
    if (l.size() < MAX_UNIQUE_KMER_POSITIONS)
    {
      any_small = true;
      break;
    }
  if (!any_small)
    return;
  uint32_t rs = 0;
  for (std::size_t i = 0; i < r0.size(); ++i)
  {
    geno.add_next_kmer_labels(g, r0[i], rs, rs + (K - 1), 0);
    geno.add_next_kmer_labels(g, r1[i], rs, rs + (K - 1), 1);
    rs += K - 1;
  }
  std::string seq;
  for (uint8_t c : read)
    seq.push_back(iupac_char(c));
  geno.remove_short_paths();
  geno.walk_read_starts(seq, -1, g);
  geno.walk_read_ends(seq, -1, g);
  geno.update_longest_path_size();
  geno.remove_short_paths();
  geno.remove_paths_with_too_many_mismatches();
  if (g.is_sv_graph)
    geno.remove_fully_special_paths(g);
  geno.remove_non_ref_paths_when_read_matches_ref(g);
  geno.update_longest_path_size();
  geno.remove_short_paths();
  if (g.is_sv_graph)
    geno.remove_support_from_read_ends(g);
  geno.read2 = seq;
}

// reverse complement of IUPAC codes: complementing swaps A<->T and C<->G, i.e. reverses the 4 bits
inline std::vector<uint8_t> reverse_complement(std::vector<uint8_t> const & s)
{
  std::vector<uint8_t> r(s.size());
  for (std::size_t i = 0; i < s.size(); ++i)
  {
    unsigned const c = s[s.size() - 1 - i] & 15u;
    r[i] = static_cast<uint8_t>(((c & 1u) << 3) | ((c & 2u) << 1) | ((c & 4u) >> 1) | ((c & 8u) >> 3));
  }
  return r;
}

// the part of bam1_t the path looks at (src/typer/alignment.cpp:331-363, 365-545)
struct ReadRecord
{
  uint16_t flag = 0;
  int32_t tid = 0, mtid = 0;
  int64_t pos = 0, isize = 0;
  uint8_t mapq = 60;
  uint8_t score_diff = 0;   // value get_score_diff() (alignment.cpp:140-325) would derive from the AS/XS aux tags
  std::vector<uint8_t> seq; // 4-bit codes, BAM code 0 already mapped to 15
  std::string name;
  int sample = 0;
  int rg = 0;
  // fields only the SV-mode record filter looks at (hts_parallel_reader.cpp:528-568)
  int64_t mpos = 0;
  uint32_t n_cigar = 0;
  uint32_t cigar_front = 0, cigar_back = 0; // raw BAM cigar words (op | len << 4) of the first and last operation
};

// alignment.cpp:331-363
inline std::pair<GenotypePaths, GenotypePaths> align_read(ReadRecord const & rec, PHIndex const & idx, Graph const & g,
                                                          Params const & par)
{
  std::pair<GenotypePaths, GenotypePaths> gp(GenotypePaths(rec.flag, rec.seq.size()), GenotypePaths(rec.flag, rec.seq.size()));
  if (rec.seq.size() < 2 * K - 1)
    return gp;
  std::vector<uint8_t> const rseq = reverse_complement(rec.seq);
  bool const one_orientation =
    (rec.flag & IS_PAIRED) == 0u || (rec.tid == rec.mtid && rec.isize > -1200 && rec.isize < 1200 &&
                                     (((rec.flag & IS_SEQ_REVERSED) != 0u) != ((rec.flag & IS_MATE_SEQ_REVERSED) != 0u)));
  find_genotype_paths_of_one_of_the_sequences(rec.seq, gp.first, idx, g);
  if (!one_orientation || par.force_align_both_orientations)
    find_genotype_paths_of_one_of_the_sequences(rseq, gp.second, idx, g);
  return gp;
}

// genotype_paths.cpp:943-974
inline int compare_pair_of_genotype_paths(GenotypePaths const & g1, GenotypePaths const & g2)
{
  std::size_t const t1 = g1.longest_path_length, t2 = g2.longest_path_length, MIN = 94;
  if (t1 > t2 && t1 > MIN)
    return 1;
  if (t2 > t1 && t2 > MIN)
    return 2;
  if (t1 == t2 && t1 > MIN)
    return g2.paths[0].mismatches < g1.paths[0].mismatches ? 2 : 1;
  return 0;
}

// genotype_paths.cpp:976-1169
inline int compare_pair_of_genotype_paths(std::pair<GenotypePaths *, GenotypePaths *> const & a,
                                          std::pair<GenotypePaths *, GenotypePaths *> const & b)
{
  auto const & a1 = *a.first;
  auto const & a2 = *a.second;
  auto const & b1 = *b.first;
  auto const & b2 = *b.second;
  std::size_t const T11 = a1.paths.empty() ? 0 : a1.longest_path_length;
  std::size_t const T12 = a2.paths.empty() ? 0 : a2.longest_path_length;
  std::size_t const T21 = b1.paths.empty() ? 0 : b1.longest_path_length;
  std::size_t const T22 = b2.paths.empty() ? 0 : b2.longest_path_length;
  std::size_t const M1 = std::max(T11, T12), M2 = std::max(T21, T22);
  std::size_t const P1 = a1.read_length, P2 = a2.read_length, MIN = 94;
  bool const perfect1 = T11 >= P1 && T12 >= P2, perfect2 = T21 >= P1 && T22 >= P2;
  if (perfect1 || perfect2)
  {
    if (perfect1 && perfect2)
    {
      std::size_t const mm1 = a1.paths[0].mismatches + a2.paths[0].mismatches;
      std::size_t const mm2 = b1.paths[0].mismatches + b2.paths[0].mismatches;
      if (mm1 != mm2)
        return mm1 < mm2 ? 1 : 2;
      std::size_t const n1 = a1.paths.size() + a2.paths.size(), n2 = b1.paths.size() + b2.paths.size();
      if (n1 != n2)
        return n1 < n2 ? 1 : 2;
      auto alt_calls = [](std::vector<Path> const & ps)
      {
        std::size_t c = 0;
        for (auto const & p : ps)
          for (auto const & n : p.nums)
            c += n.count(0) == 0;
        return c;
      };
      return alt_calls(a1.paths) + alt_calls(a2.paths) >= alt_calls(b1.paths) + alt_calls(b2.paths) ? 1 : 2;
    }
    return perfect1 ? 1 : 2;
  }
  if (M2 >= MIN && M2 > M1)
    return 2;
  if (M1 >= MIN && M1 > M2)
    return 1;
  if (M1 >= MIN && M2 >= MIN)
  {
    uint16_t mm1 = 10, mm2 = 10;
    if (T11 == M1)
      mm1 = std::min(mm1, a1.paths[0].mismatches);
    if (T12 == M1)
      mm1 = std::min(mm1, a2.paths[0].mismatches);
    if (T21 == M2)
      mm2 = std::min(mm2, b1.paths[0].mismatches);
    if (T22 == M2)
      mm2 = std::min(mm2, b2.paths[0].mismatches);
    if (mm1 != mm2)
      return mm1 < mm2 ? 1 : 2;
    if (std::min(T11, T12) < std::min(T21, T22))
      return 1;
    if (std::min(T21, T22) < std::min(T11, T12))
      return 2;
    return 0;
  }
  if (M2 == 0u && T11 >= 63u && T12 >= 63u)
    return 1;
  if (M1 == 0u && T21 >= 63u && T22 >= 63u)
    return 2;
  return 1;
}

// alignment.cpp:365-455.  clipped_count() (:105-138) returns 0/1, so the `> 3` test never sets IS_CLIPPED.
inline GenotypePaths * update_unpaired_read_paths(std::pair<GenotypePaths, GenotypePaths> & gp, ReadRecord const & rec)
{
  int const which = compare_pair_of_genotype_paths(gp.first, gp.second);
  if (which == 0)
    return nullptr;
  GenotypePaths & geno = which == 1 ? gp.first : gp.second;
  geno.flags = static_cast<uint16_t>((which == 1 ? rec.flag : (rec.flag ^ IS_SEQ_REVERSED)) & ~IS_PROPER_PAIR);
  geno.mapq = rec.mapq;
  if (!(rec.flag & IS_UNMAPPED))
    geno.original_pos = static_cast<uint32_t>(rec.pos);
  if (rec.mapq < 25)
    geno.flags |= IS_MAPQ_BAD;
  geno.score_diff = rec.score_diff;
  return &geno;
}

// alignment.cpp:482-545
inline void update_paths(std::pair<GenotypePaths, GenotypePaths> & gp, ReadRecord const & rec)
{
  GenotypePaths & g1 = gp.first;
  GenotypePaths & g2 = gp.second;
  g1.flags = static_cast<uint16_t>(rec.flag & ~IS_PROPER_PAIR);
  g1.mapq = rec.mapq;
  g1.ml_insert_size = static_cast<int32_t>(std::abs(rec.isize));
  if (!(rec.flag & IS_UNMAPPED))
  {
    g1.original_pos = static_cast<uint32_t>(rec.pos);
    g2.original_pos = g1.original_pos;
  }
  if (rec.mapq < 25)
    g1.flags |= IS_MAPQ_BAD;
  g1.score_diff = rec.score_diff;
  g2.score_diff = rec.score_diff;
  g2.flags = static_cast<uint16_t>((rec.flag ^ IS_SEQ_REVERSED) & ~IS_PROPER_PAIR);
  g2.mapq = g1.mapq;
  g2.ml_insert_size = g1.ml_insert_size;
}

// alignment.cpp:557-620
inline std::pair<GenotypePaths *, GenotypePaths *> get_better_paths(std::pair<GenotypePaths, GenotypePaths> & p1,
                                                                    std::pair<GenotypePaths, GenotypePaths> & p2)
{
  std::array<GenotypePaths *, 4> arr = {nullptr, nullptr, nullptr, nullptr};
  auto idx = [](uint16_t f) { return ((f & IS_FIRST_IN_PAIR) != 0) + 2 * ((f & IS_SEQ_REVERSED) == 0); };
  arr[idx(p1.first.flags)] = &p1.first;
  arr[idx(p1.second.flags)] = &p1.second;
  arr[idx(p2.first.flags)] = &p2.first;
  arr[idx(p2.second.flags)] = &p2.second;
  std::pair<GenotypePaths *, GenotypePaths *> none(nullptr, nullptr);
  if (!arr[0] || !arr[1] || !arr[2] || !arr[3])
    return none;
  std::pair<GenotypePaths *, GenotypePaths *> a(arr[3], arr[0]), b(arr[1], arr[2]);
  switch (compare_pair_of_genotype_paths(a, b))
  {
  case 1:
    a.first->flags |= IS_PROPER_PAIR;
    a.second->flags |= IS_PROPER_PAIR;
    return a;
  case 2:
    b.first->flags |= IS_PROPER_PAIR;
    b.second->flags |= IS_PROPER_PAIR;
    return b;
  default:
    return none;
  }
}

// ---------------------------------------------------------------------------
// scoring sink  (src/graph/haplotype.cpp, src/typer/vcf_writer.cpp)
// ---------------------------------------------------------------------------
struct HapSample // include/graphtyper/graph/haplotype.hpp:31-80
{
  std::vector<uint16_t> log_score, gt_coverage;
  uint16_t max_log_score = 0;
  std::vector<std::map<uint16_t, std::vector<uint16_t>>> connections;
  uint8_t ambiguous_depth = 0, ambiguous_depth_alt = 0, alt_proper_pair_depth = 0;
};

struct PerAlleleStats // include/graphtyper/typer/var_stats.hpp:15-22 (the fields the path writes)
{
  uint64_t clipped_bp = 0, mapq_squared = 0;
  uint32_t score_diff = 0, mismatches = 0;
};

struct ReadStrand // include/graphtyper/graph/read_strand.hpp:14-20
{
  uint32_t r1_forward = 0, r1_reverse = 0, r2_forward = 0, r2_reverse = 0;
};

struct Haplotype
{
  static constexpr uint16_t NO_COVERAGE = 0xFFFFu, MULTI_ALT_COVERAGE = 0xFFFEu, MULTI_REF_COVERAGE = 0xFFFDu;
  uint32_t id = 0;  // gt.id   (= variant order)
  uint16_t num = 0; // gt.num  (= allele count)
  uint32_t first_variant_node = 0;
  std::vector<HapSample> hap_samples;
  std::vector<PerAlleleStats> per_allele;
  std::vector<ReadStrand> read_strand;
  uint32_t clipped_reads = 0;
  uint64_t mapq_squared = 0;
  uint16_t coverage = NO_COVERAGE;
  std::set<uint64_t> explains;

  void clear_and_resize_samples(std::size_t n) // haplotype.cpp:122-147
  {
    hap_samples.clear();
    for (std::size_t i = 0; i < n; ++i)
    {
      HapSample s;
      s.log_score.assign(static_cast<std::size_t>(num) * (num + 1) / 2, 0);
      s.connections.resize(num);
      s.gt_coverage.assign(num, 0);
      hap_samples.push_back(std::move(s));
    }
  }

  void add_coverage(uint16_t c) // haplotype.cpp:180-227
  {
    if (coverage == NO_COVERAGE)
      coverage = c;
    else if (coverage == MULTI_ALT_COVERAGE)
    {
      if (c == 0)
        coverage = MULTI_REF_COVERAGE;
    }
    else if (coverage == MULTI_REF_COVERAGE)
    {
    }
    else if (coverage != c)
      coverage = (coverage == 0 || c == 0) ? MULTI_REF_COVERAGE : MULTI_ALT_COVERAGE;
  }

  void clipped_reads_to_stats(int clipped_bp, int read_length) // haplotype.cpp:229-244
  {
    if (clipped_bp == 0)
      return;
    long const scaled = (clipped_bp * 1000l) / read_length;
    if (coverage != NO_COVERAGE)
      ++clipped_reads;
    if (coverage < MULTI_REF_COVERAGE)
      per_allele[coverage].clipped_bp += scaled;
  }

  void mapq_to_stats(uint8_t mapq) // haplotype.cpp:246-261
  {
    if (mapq == 255)
      return;
    uint64_t const sq = static_cast<uint64_t>(mapq) * mapq;
    if (coverage != NO_COVERAGE)
      mapq_squared += sq;
    if (coverage < MULTI_REF_COVERAGE)
      per_allele[coverage].mapq_squared += sq;
  }

  void strand_to_stats(uint16_t flags) // haplotype.cpp:263-287
  {
    if (coverage >= MULTI_REF_COVERAGE)
      return;
    bool const fwd = (flags & IS_SEQ_REVERSED) == 0, first = (flags & IS_FIRST_IN_PAIR) != 0;
    ReadStrand & rs = read_strand[coverage];
    ++(fwd ? (first ? rs.r1_forward : rs.r2_forward) : (first ? rs.r1_reverse : rs.r2_reverse));
  }

  void mismatches_to_stats(uint8_t mismatches, int read_length) // haplotype.cpp:289-300
  {
    if (mismatches == 0)
      return;
    long const scaled = (mismatches * 1000l) / read_length;
    if (coverage < MULTI_REF_COVERAGE)
      per_allele[coverage].mismatches += static_cast<uint32_t>(scaled);
  }

  void score_diff_to_stats(uint8_t sd) // haplotype.cpp:302-311
  {
    if (sd == 0)
      return;
    if (coverage < MULTI_REF_COVERAGE)
      per_allele[coverage].score_diff += sd;
  }

  void coverage_to_gts(std::size_t pn, bool proper_pair) // haplotype.cpp:315-361 + 19-44
  {
    HapSample & s = hap_samples[pn];
    auto inc8 = [](uint8_t & x)
    {
      if (x < 0xFFu)
        ++x;
    };
    if (coverage == NO_COVERAGE)
      return;
    if (coverage == MULTI_REF_COVERAGE)
      inc8(s.ambiguous_depth);
    else if (coverage == MULTI_ALT_COVERAGE)
    {
      inc8(s.ambiguous_depth);
      inc8(s.ambiguous_depth_alt);
      if (proper_pair)
        inc8(s.alt_proper_pair_depth);
    }
    else
    {
      if (s.gt_coverage[coverage] < 0xFFFFu)
        ++s.gt_coverage[coverage];
      if (coverage > 0 && proper_pair)
        inc8(s.alt_proper_pair_depth);
    }
  }

  // haplotype.cpp:462-585
  void explain_to_score(std::size_t pn, bool non_unique_paths, uint16_t flags, bool fully_aligned, bool is_read_overlapping,
                        bool is_low_qual, std::size_t mismatches)
  {
    long e = EPSILON_0_EXPONENT;
    e -= static_cast<long>(mismatches);
    if (non_unique_paths)
      e -= 3;
    if (flags & IS_MAPQ_BAD)
      e -= 2;
    if (!fully_aligned)
      e -= 3;
    if (!is_read_overlapping)
      e -= 1;
    if (is_low_qual)
      e -= 2;
    uint16_t const eps = static_cast<uint16_t>(std::max(e, 8l) - 4);
    HapSample & s = hap_samples[pn];
    if (s.max_log_score < (0xFFFFul - eps))
    {
      s.max_log_score = static_cast<uint16_t>(s.max_log_score + eps);
      int i = 0;
      for (std::size_t y = 0; y < num; ++y)
      {
        bool const ey = explains.count(y) == 1;
        for (std::size_t x = 0; x <= y; ++x, ++i)
        {
          bool const ex = explains.count(x) == 1;
          if (ex && ey)
            s.log_score[i] = static_cast<uint16_t>(s.log_score[i] + eps);
          else if (ex || ey)
            s.log_score[i] = static_cast<uint16_t>(s.log_score[i] + eps - 1);
        }
      }
    }
  }
};

using ConnKey = std::pair<uint16_t, uint16_t>;
using ConnMap = std::map<ConnKey, std::vector<ConnKey>>;

struct VcfWriter // src/typer/vcf_writer.cpp
{
  Graph const * graph = nullptr;
  Params par;
  std::vector<Haplotype> haplotypes;
  std::unordered_map<uint32_t, uint32_t> id2hap;

  VcfWriter(Graph const & g, Params const & p, std::size_t n_samples) : graph(&g), par(p) // vcf_writer.cpp:66-86, graph.cpp:680-704
  {
    uint32_t v = 0;
    if (!g.var_nodes.empty())
      for (std::size_t r = 0; r + 1 < g.ref_nodes.size(); ++r)
      {
        Haplotype h;
        h.id = g.var_nodes[v].label.order;
        h.num = static_cast<uint16_t>(g.ref_nodes[r].out_var_ids.size());
        h.first_variant_node = v;
        h.per_allele.resize(h.num);
        h.read_strand.resize(h.num);
        haplotypes.push_back(std::move(h));
        v += static_cast<uint32_t>(g.ref_nodes[r].out_var_ids.size());
      }
    for (std::size_t i = 0; i < haplotypes.size(); ++i)
    {
      haplotypes[i].clear_and_resize_samples(n_samples);
      id2hap[haplotypes[i].id] = static_cast<uint32_t>(i);
    }
  }

  bool are_genotype_paths_good(GenotypePaths const & geno) const // vcf_writer.cpp:28-60
  {
    if (geno.paths.empty())
      return false;
    bool const fully = geno.all_paths_fully_aligned();
    if (!fully && (!geno.all_paths_unique(*graph) || geno.paths[0].size() < 63))
      return false;
    double const ratio = static_cast<double>(geno.paths[0].mismatches) / static_cast<double>(geno.paths[0].size());
    if (ratio > 0.05)
      return false;
    if (!fully && ratio > 0.025)
      return false;
    if (graph->is_sv_graph && (!fully || geno.paths[0].size() < 90 || ratio > 0.03))
      return false;
    if (par.hq_reads && (!fully || geno.paths[0].size() < 90 || ratio > 0.035))
      return false;
    return true;
  }

  // vcf_writer.cpp:503-676 (qual2 is empty in release genotype-only mode, so has_low_quality_snp stays false)
  ConnMap push_to_haplotype_scores(GenotypePaths & geno, long pn)
  {
    int const clipped_bp = geno.read_length - static_cast<int>(geno.longest_path_length);
    bool const fully_aligned = clipped_bp == 0;
    bool const non_unique = !geno.all_paths_unique(*graph);
    std::size_t const mismatches = geno.paths[0].mismatches;
    std::map<uint32_t, bool> recent_ids;
    ConnMap new_connections;
    for (auto const & p : geno.paths)
      for (std::size_t i = 0; i < p.var_order.size(); ++i)
      {
        uint32_t const hap_id = id2hap.at(p.var_order[i]);
        if (p.nums[i].empty())
          continue;
        Haplotype & hap = haplotypes[hap_id];
        auto const & num = p.nums[i];
        long constexpr MIN_OFFSET = 3;
        bool const overlapping = static_cast<long>(graph->get_ref_reach_pos(p.start)) + MIN_OFFSET <= static_cast<long>(p.var_order[i]) &&
                                 static_cast<long>(graph->get_ref_reach_pos(p.end)) - MIN_OFFSET > static_cast<long>(p.var_order[i]);
        recent_ids[hap_id] |= overlapping;
        hap.explains.insert(num.begin(), num.end());
        if (num.size() == 1)
          hap.add_coverage(*num.begin());
        else
        {
          hap.add_coverage(1);
          hap.add_coverage(num.count(0) == 1 ? 0 : 2);
        }
      }
    for (auto it = recent_ids.begin(); it != recent_ids.end(); ++it)
    {
      Haplotype & h1 = haplotypes[it->first];
      long const n1 = static_cast<long>(h1.explains.size());
      if (n1 == 0 || n1 > 64)
        continue;
      for (uint64_t b1 : h1.explains)
      {
        auto & conn = new_connections[{static_cast<uint16_t>(it->first), static_cast<uint16_t>(b1)}];
        for (auto it2 = std::next(it); it2 != recent_ids.end(); ++it2)
        {
          Haplotype & h2 = haplotypes[it2->first];
          long const n2 = static_cast<long>(h2.explains.size());
          if (n2 == 0 || n2 > 64)
            continue;
          long const weight = n1 * n2;
          long const repeat = weight >= 3 ? 6 / weight : 1;
          for (uint64_t b2 : h2.explains)
            for (long r = 0; r < repeat; ++r)
              conn.push_back({static_cast<uint16_t>(it2->first), static_cast<uint16_t>(b2)});
        }
      }
    }
    for (auto it = recent_ids.begin(); it != recent_ids.end(); ++it)
    {
      Haplotype & h = haplotypes[it->first];
      h.clipped_reads_to_stats(clipped_bp, geno.read_length);
      h.mapq_to_stats(geno.mapq);
      h.strand_to_stats(geno.flags);
      h.mismatches_to_stats(static_cast<uint8_t>(mismatches), geno.read_length);
      h.score_diff_to_stats(geno.score_diff);
      h.explain_to_score(static_cast<std::size_t>(pn), non_unique, geno.flags, fully_aligned, it->second, false, mismatches);
      h.coverage_to_gts(static_cast<std::size_t>(pn), geno.is_proper_pair());
      h.coverage = Haplotype::NO_COVERAGE;
      h.explains.clear();
    }
    return new_connections;
  }

  void commit_connections(ConnMap const & merged, long pn) // vcf_writer.cpp:120-139, 229-249
  {
    for (auto const & kv : merged)
    {
      auto & conn = haplotypes[kv.first.first].hap_samples[pn].connections[kv.first.second];
      for (auto const & t : kv.second)
      {
        auto ins = conn.insert({t.first, std::vector<uint16_t>(haplotypes[t.first].num)});
        ++ins.first->second[t.second];
      }
    }
  }

  void update_haplotype_scores_geno(GenotypePaths & geno, long pn) // vcf_writer.cpp:88-141
  {
    if (par.is_segment_calling)
      return;
    if (are_genotype_paths_good(geno))
      commit_connections(push_to_haplotype_scores(geno, pn), pn);
  }

  void update_haplotype_scores_geno(std::pair<GenotypePaths *, GenotypePaths *> & gp, long pn) // vcf_writer.cpp:143-250
  {
    bool const good1 = are_genotype_paths_good(*gp.first), good2 = are_genotype_paths_good(*gp.second);
    if (par.is_segment_calling && (!good1 || !good2))
      return;
    ConnMap con1, con2, merged;
    if (good1)
      con1 = push_to_haplotype_scores(*gp.first, pn);
    if (good2)
      con2 = push_to_haplotype_scores(*gp.second, pn);
    for (auto const & kv1 : con1)
    {
      auto & tgt = merged.insert(kv1).first->second;
      for (auto const & kv2 : con2)
        if (kv2.first.first > kv1.first.first)
          tgt.push_back(kv2.first);
    }
    for (auto const & kv2 : con2)
    {
      auto ins = merged.insert(kv2);
      if (!ins.second)
        ins.first->second.insert(ins.first->second.end(), kv2.second.begin(), kv2.second.end());
      for (auto const & kv1 : con1)
        if (kv1.first.first > kv2.first.first)
          ins.first->second.push_back(kv1.first);
    }
    commit_connections(merged, pn);
  }
};

// ---------------------------------------------------------------------------
// ReferenceDepth (src/graph/reference_depth.cpp:17-28, 109-201, 220-229): per sample the number of accepted reads
// over every reference position; SV calling only (hts_parallel_reader.cpp:505-508)
// ---------------------------------------------------------------------------
struct ReferenceDepth
{
  Graph const * graph = nullptr;
  uint32_t reference_offset = 0;
  std::vector<std::vector<uint16_t>> depths;

  void init(Graph const & g, long sample_count)
  {
    graph = &g;
    reference_offset = g.ref_nodes.empty() ? 0 : g.ref_nodes[0].label.order;
    depths.assign(static_cast<std::size_t>(sample_count), std::vector<uint16_t>(g.reference_size, 0));
  }
  long start_pos_to_index(long start_pos) const { return start_pos < reference_offset ? 0 : start_pos - reference_offset; }
  long end_pos_to_index(long end_pos, long depth_size) const
  {
    return end_pos > reference_offset + depth_size ? depth_size : end_pos + 1 - reference_offset;
  }
  void add_genotype_paths(GenotypePaths const & geno, long sample_index) // :109-201
  {
    if (depths.empty() || sample_index >= static_cast<long>(depths.size()))
      return;
    if (geno.paths.empty() || geno.paths[0].size() < 63)
      return;
    auto & depth = depths[sample_index];
    long const size = static_cast<long>(depth.size());
    if (geno.paths.size() == 1)
    {
      Path const & path = geno.paths[0];
      long const start_pos = static_cast<long>(graph->get_ref_reach_pos(path.start)) - path.read_start_index;
      long const end_pos = static_cast<long>(graph->get_ref_reach_pos(path.end)) + (geno.read_length - 1 - path.read_end_index);
      long const start_index = start_pos_to_index(start_pos), end_index = end_pos_to_index(end_pos, size);
      if (start_index < size)
        for (long i = start_index; i < end_index && i < size; ++i) // (an end in front of the start: nothing -- the reference's
          ++depth[i];                                              //  iterator loop would run away there)
      return;
    }
    std::unordered_set<long> local_depth;
    for (Path const & path : geno.paths)
    {
      long start_pos = static_cast<long>(graph->get_ref_reach_pos(path.start)) - path.read_start_index;
      long end_pos = static_cast<long>(graph->get_ref_reach_pos(path.end)) + (geno.read_length - 1 - path.read_end_index);
      if (end_pos - start_pos >= 50)
      {
        start_pos += 4;
        end_pos -= 4;
      }
      if (end_pos < reference_offset)
        continue;
      long const start_index = start_pos_to_index(start_pos), end_index = end_pos_to_index(end_pos, size);
      if (start_index < size)
        for (long i = start_index; i < end_index && i < size; ++i)
          local_depth.insert(i);
    }
    for (long i : local_depth)
      if (depth[i] < 0xFFFFul)
        ++depth[i];
  }
  void merge_from(ReferenceDepth const & o) // (test infrastructure, see Genotyper::merge_from; clamped like gtx_ref_depth_finalize)
  {
    for (std::size_t s = 0; s < depths.size() && s < o.depths.size(); ++s)
      for (std::size_t i = 0; i < depths[s].size(); ++i)
        depths[s][i] = static_cast<uint16_t>(std::min<uint32_t>(0xFFFFu, static_cast<uint32_t>(depths[s][i]) + o.depths[s][i]));
  }
};

// ---------------------------------------------------------------------------
// per-record driver  (src/utilities/hts_parallel_reader.cpp:245-338, 655-708) for non-SV graphs
// ---------------------------------------------------------------------------
struct Genotyper
{
  Graph const & graph;
  PHIndex const & index;
  Params par;
  VcfWriter writer;
  std::vector<std::unordered_map<std::string, std::pair<GenotypePaths, GenotypePaths>>> maps; // one per read group
  std::pair<GenotypePaths, GenotypePaths> prev_paths;
  ReadRecord prev;
  bool have_prev = false;
  long num_records = 0, num_duplicated = 0;

  Genotyper(Graph const & g, PHIndex const & i, Params const & p, std::size_t n_samples, std::size_t n_rg)
    : graph(g), index(i), par(p), writer(g, p, n_samples), maps(n_rg)
  {
    if (g.is_sv_graph) // hts_parallel_reader.cpp:505-508
      reference_depth.init(g, static_cast<long>(n_samples));
  }
  ReferenceDepth reference_depth;

  static bool equal_pos_seq(ReadRecord const & a, ReadRecord const & b) // include/graphtyper/utilities/hts_utils.hpp:110-128
  {
    return a.tid == b.tid && a.pos == b.pos && a.seq == b.seq;
  }

  void genotype_only(ReadRecord const & rec, bool update_prev) // hts_parallel_reader.cpp:245-338
  {
    if (update_prev)
      prev_paths = align_read(rec, index, graph, par);
    std::pair<GenotypePaths, GenotypePaths> gp(prev_paths);
    auto & map = maps.at(rec.rg);
    auto it = map.find(rec.name);
    if (it == map.end())
    {
      if (rec.flag & IS_PAIRED)
      {
        update_paths(gp, rec);
        map[rec.name] = std::move(gp);
        parked_sample[rec.name] = rec.sample;
      }
      else if (GenotypePaths * sel = update_unpaired_read_paths(gp, rec))
        writer.update_haplotype_scores_geno(*sel, rec.sample);
      return;
    }
    update_paths(gp, rec);
    if ((gp.first.flags & IS_FIRST_IN_PAIR) == (it->second.first.flags & IS_FIRST_IN_PAIR))
      throw std::runtime_error("gto: two reads named " + rec.name + " have the same IS_FIRST_IN_PAIR");
    auto better = get_better_paths(it->second, gp);
    if (better.first)
    {
      if (graph.is_sv_graph) // hts_parallel_reader.cpp:324-329
      {
        reference_depth.add_genotype_paths(*better.first, rec.sample);
        reference_depth.add_genotype_paths(*better.second, rec.sample);
      }
      writer.update_haplotype_scores_geno(better, rec.sample);
    }
    map.erase(it);
  }

  // SV calling only: hts_parallel_reader.cpp:528-568
  static bool is_good_read(ReadRecord const & r)
  {
    if ((r.flag & IS_UNMAPPED) != 0u)
      return false;
    static constexpr char CIGAR_MAP[16] = {'M', 'I', 'D', 'N', 'S', 'H', 'P', '=', 'X', 'B', '*', '*', '*', '*', '*', '*'};
    bool const is_mate_far_away = r.tid != r.mtid || std::abs(r.pos - r.mpos) > 200000;
    if (r.mapq <= 15 && is_mate_far_away)
      return false;
    if (r.n_cigar >= 2)
    {
      char const front = CIGAR_MAP[r.cigar_front & 15], back = CIGAR_MAP[r.cigar_back & 15];
      uint32_t const count_front = r.cigar_front >> 4, count_back = r.cigar_back >> 4;
      bool const is_one_clipped = (front == 'S' && count_front >= 12) || (back == 'S' && count_back >= 12);
      bool const are_both_clipped = front == 'S' && back == 'S';
      if (are_both_clipped || (r.mapq <= 15 && is_one_clipped))
        return false;
    }
    return true;
  }

  // (extreme) coverage filter of SV calling: hts_parallel_reader.cpp:594-633
  std::vector<double> avg_cov_by_readlen; // per sample; empty = filter has nothing to go by
  bool no_filter_on_coverage = false;
  std::vector<std::vector<uint16_t>> bin_counts;
  long first_pos = 0;

  bool update_bin_count(ReadRecord const & rec)
  {
    if (!(graph.is_sv_graph && !no_filter_on_coverage))
      return true;
    long const sample_i = rec.sample;
    if (sample_i >= static_cast<long>(avg_cov_by_readlen.size()) || avg_cov_by_readlen[sample_i] <= 0.0)
      return true;
    uint16_t const max_bin_count = static_cast<uint16_t>(std::min(65535l, static_cast<long>(avg_cov_by_readlen[sample_i] * 50.0 * 3.0 + 0.5)));
    if (static_cast<long>(bin_counts.size()) <= sample_i)
      bin_counts.resize(sample_i + 1);
    auto & sample_bin_counts = bin_counts[sample_i];
    long const bin = (rec.pos - first_pos) / 50l;
    if (bin >= static_cast<long>(sample_bin_counts.size()))
    {
      sample_bin_counts.resize(bin + 1, 0u);
      ++sample_bin_counts[bin];
      return true;
    }
    if (sample_bin_counts[bin] > max_bin_count)
      return false;
    ++sample_bin_counts[bin];
    return true;
  }

  void push(ReadRecord const & rec) // hts_parallel_reader.cpp:570-708 (first record and loop body)
  {
    if ((rec.flag & par.sam_flag_filter) != 0 || (graph.is_sv_graph && !is_good_read(rec)))
      return;
    if (!have_prev) // the first record that passes the filters
    {
      first_pos = rec.pos;
      update_bin_count(rec);
      ++num_records;
      genotype_only(rec, true);
      prev = rec;
      have_prev = true;
      return;
    }
    ++num_records;
    if (equal_pos_seq(prev, rec))
    {
      update_bin_count(rec);
      ++num_duplicated;
      genotype_only(rec, false);
      return;
    }
    if (!update_bin_count(rec))
    {
      --num_records; // skipped
      return;
    }
    genotype_only(rec, true);
    prev = rec;
  }

  // SV calling only: reads whose mate never showed up are scored on their own (hts_parallel_reader.cpp:717-772).
  // `sample_of_rg`: the reference derives the sample from the file / read group; the parked read's own sample is that.
  void finish()
  {
    if (graph.is_sv_graph)
      for (auto & map : maps)
        for (auto & kv : map)
        {
          std::pair<GenotypePaths, GenotypePaths> copy(kv.second);
          copy.first.flags ^= (IS_FIRST_IN_PAIR | IS_SEQ_REVERSED);
          copy.second.flags ^= (IS_FIRST_IN_PAIR | IS_SEQ_REVERSED);
          auto better = get_better_paths(kv.second, copy);
          if (better.first)
          {
            reference_depth.add_genotype_paths(*better.first, parked_sample.at(kv.first)); // hts_parallel_reader.cpp:739-741
            writer.update_haplotype_scores_geno(*better.first, parked_sample.at(kv.first));
          }
        }
    for (auto & map : maps)
      map.clear();
    parked_sample.clear();
  }
  std::unordered_map<std::string, int> parked_sample;

  // Test infrastructure, no counterpart in the reference: adds the accumulated state of another Genotyper over the same
  // graph and samples into this one, so that a large read set can be pushed through several Genotypers on several host
  // threads (tests/test_gpu_full_size.py: all 10 M reads of BASELINE cfg2).  Every per-read effect on a haplotype is an
  // addition (explain_to_score, coverage_to_gts, *_to_stats, commit_connections), so the sum equals one sequential pass as
  // long as no counter reaches its saturation point: the u8 / u16 depth counters saturate (haplotype.cpp:19-44) -- summed and
  // clamped the same way here, which is order-free -- but explain_to_score's guard (haplotype.cpp:560) is sequential, so a
  // cell whose summed max_log_score comes within 8 of 0xFFFF is refused.  Reads parked for their mates are not merged
  // (shards must hold both mates).
  void merge_from(Genotyper const & o)
  {
    if (o.writer.haplotypes.size() != writer.haplotypes.size())
      throw std::runtime_error("gto: merge of genotypers over different graphs");
    for (auto const & m : o.maps)
      if (!m.empty())
        throw std::runtime_error("gto: merge of a genotyper with parked mates");
    for (std::size_t h = 0; h < writer.haplotypes.size(); ++h)
    {
      Haplotype & a = writer.haplotypes[h];
      Haplotype const & b = o.writer.haplotypes[h];
      if (a.num != b.num || a.hap_samples.size() != b.hap_samples.size())
        throw std::runtime_error("gto: merge of genotypers with different samples");
      a.clipped_reads += b.clipped_reads;
      a.mapq_squared += b.mapq_squared;
      for (std::size_t k = 0; k < a.per_allele.size(); ++k)
      {
        a.per_allele[k].clipped_bp += b.per_allele[k].clipped_bp;
        a.per_allele[k].mapq_squared += b.per_allele[k].mapq_squared;
        a.per_allele[k].score_diff += b.per_allele[k].score_diff;
        a.per_allele[k].mismatches += b.per_allele[k].mismatches;
        a.read_strand[k].r1_forward += b.read_strand[k].r1_forward;
        a.read_strand[k].r1_reverse += b.read_strand[k].r1_reverse;
        a.read_strand[k].r2_forward += b.read_strand[k].r2_forward;
        a.read_strand[k].r2_reverse += b.read_strand[k].r2_reverse;
      }
      for (std::size_t s = 0; s < a.hap_samples.size(); ++s)
      {
        HapSample & x = a.hap_samples[s];
        HapSample const & y = b.hap_samples[s];
        uint32_t const mx = static_cast<uint32_t>(x.max_log_score) + y.max_log_score;
        if (mx >= 0xFFFFu - 8u)
          throw std::runtime_error("gto: merge would pass the saturation guard of explain_to_score");
        x.max_log_score = static_cast<uint16_t>(mx);
        for (std::size_t i = 0; i < x.log_score.size(); ++i)
          x.log_score[i] = static_cast<uint16_t>(x.log_score[i] + y.log_score[i]); // (<= max_log_score: no wrap)
        for (std::size_t i = 0; i < x.gt_coverage.size(); ++i)
          x.gt_coverage[i] = static_cast<uint16_t>(std::min<uint32_t>(0xFFFFu, static_cast<uint32_t>(x.gt_coverage[i]) + y.gt_coverage[i]));
        auto add8 = [](uint8_t & u, uint8_t v) { u = static_cast<uint8_t>(std::min<uint32_t>(0xFFu, static_cast<uint32_t>(u) + v)); };
        add8(x.ambiguous_depth, y.ambiguous_depth);
        add8(x.ambiguous_depth_alt, y.ambiguous_depth_alt);
        add8(x.alt_proper_pair_depth, y.alt_proper_pair_depth);
        for (std::size_t al = 0; al < x.connections.size(); ++al)
          for (auto const & kv : y.connections[al])
          {
            auto ins = x.connections[al].insert({kv.first, std::vector<uint16_t>(kv.second.size())});
            for (std::size_t i = 0; i < kv.second.size(); ++i)
              ins.first->second[i] = static_cast<uint16_t>(ins.first->second[i] + kv.second[i]); // (wraps like the reference's ++)
          }
      }
    }
    num_records += o.num_records;
    num_duplicated += o.num_duplicated;
    if (graph.is_sv_graph)
      reference_depth.merge_from(o.reference_depth);
  }

  // Vcf::add_haplotype (src/typer/vcf.cpp:1507-1530): per haplotype and sample the SampleCall the reference builds --
  // get_haplotype_phred (vcf.cpp:47-82), SampleCall constructor / get_gt_call / get_gq (sample_call.cpp:34-131).
  struct Call
  {
    std::vector<uint8_t> phred;
    uint16_t gt_first = 0, gt_second = 0, ref_total_depth = 0, alt_total_depth = 0;
    uint8_t gq = 0, ambiguous_depth = 0, alt_proper_pair_depth = 0;
  };
  std::vector<std::vector<Call>> sample_calls() const // [haplotype][sample]
  {
    std::vector<std::vector<Call>> out;
    for (auto const & hap : writer.haplotypes)
    {
      std::vector<Call> row;
      for (auto const & hs : hap.hap_samples)
      {
        Call c;
        long const num = static_cast<long>(hs.log_score.size());
        uint16_t const max_log_score = *std::max_element(hs.log_score.begin(), hs.log_score.end());
        bool const all_equal = std::find_if(hs.log_score.begin(), hs.log_score.end(), [max_log_score](uint16_t v) { return v != max_log_score; }) == hs.log_score.end();
        if (all_equal)
          c.phred.assign(num, 0u);
        else
        {
          c.phred.assign(num, 255u);
          for (long i = 0; i < num; ++i)
          {
            double const LOG10_HALF_times_10 = 3.01029995663981195213738894724493026768189881462108541;
            long const score = std::llround((max_log_score - hs.log_score[i]) * LOG10_HALF_times_10);
            if (score < 255u)
              c.phred[i] = static_cast<uint8_t>(score);
          }
        }
        // SampleCall::SampleCall
        uint32_t const ref_depth = hs.gt_coverage[0] + hs.ambiguous_depth - hs.ambiguous_depth_alt;
        c.ref_total_depth = static_cast<uint16_t>(std::min<uint32_t>(0xFFFFu, ref_depth));
        uint32_t const alt_depth = std::accumulate(hs.gt_coverage.begin() + 1, hs.gt_coverage.end(), 0u) + hs.ambiguous_depth;
        c.alt_total_depth = static_cast<uint16_t>(std::min<uint32_t>(0xFFFFu, alt_depth));
        c.ambiguous_depth = hs.ambiguous_depth;
        c.alt_proper_pair_depth = hs.alt_proper_pair_depth;
        // get_gt_call
        {
          std::size_t i = 0;
          bool found = false;
          for (std::size_t y = 0; y < hs.gt_coverage.size() && !found; ++y)
            for (std::size_t x = 0; x <= y; ++x, ++i)
              if (c.phred[i] == 0)
              {
                c.gt_first = static_cast<uint16_t>(x);
                c.gt_second = static_cast<uint16_t>(y);
                found = true;
                break;
              }
        }
        // get_gq
        {
          bool seen_zero = false, two_zeros = false;
          uint8_t next_lowest_phred = 255;
          for (auto const p : c.phred)
          {
            if (p == 0)
            {
              if (!seen_zero)
                seen_zero = true;
              else
              {
                two_zeros = true;
                break;
              }
            }
            else if (p < next_lowest_phred)
              next_lowest_phred = p;
          }
          c.gq = two_zeros ? 0 : next_lowest_phred;
        }
        row.push_back(std::move(c));
      }
      out.push_back(std::move(row));
    }
    return out;
  }

  // hts_parallel_reader.cpp:782-904: phasing flags between alt alleles of sites less than 100 bp apart, from the
  // per-sample allele depths and allele-pair connection counts.  Keys are (haplotype index, allele) as uint16_t; an outer
  // key is created as soon as a connection exists, whether or not a flag is set under it.
  using PhKey = std::pair<uint16_t, uint16_t>;
  std::map<PhKey, std::map<PhKey, int8_t>> phase_flags() const
  {
    constexpr int8_t IS_ANY_HAP_SUPPORT = 1, IS_ANY_ANTI_HAP_SUPPORT = 2; // include/graphtyper/constants.hpp.in:56-57
    std::map<PhKey, std::map<PhKey, int8_t>> ph;
    auto const & haps = writer.haplotypes;
    for (long ps1 = 0; ps1 < static_cast<long>(haps.size()) - 1l; ++ps1)
    {
      auto const & hap1 = haps[ps1];
      long const order1 = hap1.id;
      for (long ps2 = ps1 + 1l; ps2 < static_cast<long>(haps.size()); ++ps2)
      {
        auto const & hap2 = haps[ps2];
        long const order2 = hap2.id;
        if (order2 >= order1 + 100)
          break;
        for (long s = 0; s < static_cast<long>(hap1.hap_samples.size()); ++s)
        {
          auto const & hs1 = hap1.hap_samples[s];
          auto const & hs2 = hap2.hap_samples[s];
          double const total1 = std::accumulate(hs1.gt_coverage.begin(), hs1.gt_coverage.end(), 0.0);
          double const total2 = std::accumulate(hs2.gt_coverage.begin(), hs2.gt_coverage.end(), 0.0);
          for (long cov1 = 1; cov1 < static_cast<long>(hap1.num); ++cov1)
          {
            auto const & conn = hs1.connections[cov1];
            auto find_it = conn.find(static_cast<uint16_t>(ps2));
            if (find_it == conn.end())
              continue;
            bool const is_clearly_seen1 = hs1.gt_coverage[cov1] >= 4 || static_cast<double>(hs1.gt_coverage[cov1]) / total1 >= 0.28;
            bool const is_not_seen1 = hs1.gt_coverage[cov1] <= 2 || static_cast<double>(hs1.gt_coverage[cov1]) / total1 < 0.22;
            auto & row = ph[{static_cast<uint16_t>(ps1), static_cast<uint16_t>(cov1)}];
            std::vector<uint16_t> const & support_vec = find_it->second;
            long const total_support = std::accumulate(support_vec.begin(), support_vec.end(), 0l);
            for (long cov2 = 1; cov2 < static_cast<long>(support_vec.size()); ++cov2)
            {
              double const support = static_cast<double>(support_vec[cov2]);
              int8_t is_good = 0;
              bool const is_clearly_seen2 = hs2.gt_coverage[cov2] >= 4 || static_cast<double>(hs2.gt_coverage[cov2]) / total2 >= 0.28;
              bool const is_not_seen2 = hs2.gt_coverage[cov2] <= 2 || static_cast<double>(hs2.gt_coverage[cov2]) / total2 < 0.22;
              if (is_not_seen1 && is_not_seen2)
                continue;
              if ((is_not_seen1 && is_clearly_seen2) || (is_not_seen2 && is_clearly_seen1))
                is_good = IS_ANY_ANTI_HAP_SUPPORT;
              else
              {
                if (total_support <= 2)
                  continue;
                if (is_clearly_seen1 && is_clearly_seen2 && support / static_cast<double>(total_support) > 0.78)
                  is_good = IS_ANY_HAP_SUPPORT;
                else if (support / static_cast<double>(total_support) < 0.22)
                  is_good = IS_ANY_ANTI_HAP_SUPPORT;
                else
                  continue;
              }
              row[{static_cast<uint16_t>(ps2), static_cast<uint16_t>(cov2)}] |= is_good;
            }
          }
        }
      }
    }
    return ph;
  }
};

} // namespace gto
