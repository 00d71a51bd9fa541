// gtx_vcf.cpp -- the VCF records of a genotyped region as text, host side.
//
// One record per variant site, what the reference writes for the Variant it makes of a haplotype:
//   Vcf::add_haplotype (src/typer/vcf.cpp:1507-1611) -> Variant::scan_calls (src/typer/variant.cpp:230-428) ->
//   Variant::generate_infos (variant.cpp:430-1096; VarStats::write_stats src/typer/var_stats.cpp:53-141) ->
//   Vcf::write_record (vcf.cpp:767-1149, PL / GQ through include/graphtyper/typer/binned_pl.hpp).
// Input: the per (haplotype, sample) calls of gtx_calls_batch and the accumulators of gtx_score_batch, as host copies.
// Here a site is a row of flat arrays, its INFO field a sorted list of (key, text) built once; numbers are printed with
// printf's %g, which is what the reference's string streams do at the precision they set.
// The calls of an SV graph take another road (sv_graph_records below: reformat_sv_vcf_records and what genotype_sv's merge does).
// Not built: variant break-down and the merge of pools (vcf_operations.cpp).
#include "gtx_ctx.hpp"

#include <algorithm>
#include <charconv>
#include <zlib.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

namespace
{
// PL / GQ binning: the first PL of every bin and the value the bin is written as (binned_pl.hpp as a step function)
const uint16_t BIN_FIRST[] = {0, 1, 3, 5, 8, 11, 14, 18, 23, 28, 33, 38, 45, 55, 68, 80, 113, 138, 175, 228};
const uint16_t BIN_VALUE[] = {0, 1, 3, 6, 9, 12, 15, 20, 25, 30, 35, 40, 50, 60, 75, 99, 125, 150, 200, 255};

struct Binned
{
  uint16_t v[256];
  Binned()
  {
    unsigned b = 0;
    for (unsigned pl = 0; pl < 256; ++pl)
    {
      while (b + 1 < sizeof(BIN_FIRST) / sizeof(BIN_FIRST[0]) && pl >= BIN_FIRST[b + 1])
        ++b;
      v[pl] = BIN_VALUE[b];
    }
  }
};
const Binned BINNED;

void put_u(std::string & s, uint64_t v) // (most of a VCF line is small integers: digits by hand, not through printf)
{
  char b[24];
  int n = 24;
  do
  {
    b[--n] = static_cast<char>('0' + v % 10);
    v /= 10;
  } while (v != 0);
  s.append(b + n, static_cast<size_t>(24 - n));
}

// ostream << double at that precision, default float format (= printf's %.*g).  Most of what a record holds is a whole number
// below 10^precision, which that format writes as its digits; the rest goes through std::to_chars, which is specified to give
// printf's characters (compared with snprintf over 1.8 x 10^8 values -- every a / b up to 3 000, random bit patterns, the
// neighbours of the places where the notation changes -- when it replaced it: a record's text was a third snprintf)
void put_g(std::string & s, double v, int precision)
{
  static double const below[] = {1, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9};
  if (!std::isfinite(v) || precision < 1 || precision > 9)
  {
    char b[40];
    int const n = std::snprintf(b, sizeof(b), "%.*g", precision, v);
    s.append(b, static_cast<size_t>(n));
    return;
  }
  if (!std::signbit(v) && v < below[precision])
  {
    uint64_t const u = static_cast<uint64_t>(v);
    if (static_cast<double>(u) == v)
    {
      put_u(s, u);
      return;
    }
  }
  char b[48];
  auto const r = std::to_chars(b, b + sizeof(b), v, std::chars_format::general, precision);
  s.append(b, static_cast<size_t>(r.ptr - b));
}

// digits behind p; returns the place behind them (the sample columns of a record are written into room made beforehand)
inline char * put_u_at(char * p, uint32_t v)
{
  if (v < 10)
  {
    *p++ = static_cast<char>('0' + v);
    return p;
  }
  if (v < 100)
  {
    *p++ = static_cast<char>('0' + v / 10);
    *p++ = static_cast<char>('0' + v % 10);
    return p;
  }
  char b[12];
  int n = 12;
  do
  {
    b[--n] = static_cast<char>('0' + v % 10);
    v /= 10;
  } while (v != 0);
  std::memcpy(p, b + n, static_cast<size_t>(12 - n));
  return p + (12 - n);
}

struct AlleleStats // VarStatsPerAllele + ReadStrand (var_stats.hpp:15-33, read_strand.hpp)
{
  uint64_t clipped_bp = 0, mapq_squared = 0, qd_qual = 0, qd_depth = 0, total_depth = 0;
  uint32_t score_diff = 0, mismatches = 0, ac = 0, pass_ac = 0, n_ref_ref = 0, n_ref_alt = 0, n_alt_alt = 0;
  uint32_t r1f = 0, r1r = 0, r2f = 0, r2r = 0;
  uint32_t het_first = 0, het_second = 0, hom_first = 0, hom_second = 0;
  uint16_t max_alt_support = 0;
  double max_alt_support_ratio = 0.0;
};

// one sample's call at one site, as views into the caller's arrays
struct CallView
{
  const uint8_t * phred;
  const uint32_t * cov; // unclamped sums
  const gtx_sample_call * c;
  uint32_t cnum, n_tri;
  uint32_t coverage(uint32_t a) const { return std::min<uint32_t>(cov[a], 0xFFFFu); } // gtx_scores_finalize
  uint32_t unique_depth() const
  {
    uint32_t d = 0;
    for (uint32_t a = 0; a < cnum; ++a)
      d += coverage(a);
    return d;
  }
  bool any_pl() const
  {
    for (uint32_t i = 0; i < n_tri; ++i)
      if (phred[i])
        return true;
    return false;
  }
  // SampleCall::get_lowest_phred_not_with (sample_call.cpp:133-157): the smallest PL among genotypes without `allele`
  uint8_t lowest_phred_without(uint32_t allele) const
  {
    uint8_t m = 255;
    uint32_t i = 0;
    for (uint32_t y = 0; y < cnum; ++y)
      for (uint32_t x = 0; x <= y; ++x, ++i)
        if (x != allele && y != allele && phred[i] < m)
          m = phred[i];
    return m;
  }
};

double excess_het_p(int obs_hets, int obs_hom1, int obs_hom2) // p_hwe_excess_het, src/utilities/snp_hwe.cpp:19-111
{
  if (obs_hets == 0 && (obs_hom1 == 0 || obs_hom2 == 0))
    return 1.0;
  int const homc = std::max(obs_hom1, obs_hom2), homr = std::min(obs_hom1, obs_hom2);
  int const rare = 2 * homr + obs_hets, n = obs_hets + homc + homr;
  std::vector<double> p(static_cast<size_t>(rare) + 1, 0.0);
  int mid = static_cast<int>(static_cast<double>(rare) * (static_cast<double>(2 * n - rare) / static_cast<double>(2 * n)));
  if ((rare & 1) ^ (mid & 1))
    ++mid;
  p[mid] = 1.0;
  double sum = 1.0;
  {
    int hr = (rare - mid) / 2, hc = n - mid - hr;
    for (int h = mid; h > 1; h -= 2, ++hr, ++hc)
    {
      p[h - 2] = p[h] * h * (h - 1.0) / (4.0 * (hr + 1.0) * (hc + 1.0));
      sum += p[h - 2];
    }
  }
  {
    int hr = (rare - mid) / 2, hc = n - mid - hr;
    for (int h = mid; h <= rare - 2; h += 2, --hr, --hc)
    {
      p[h + 2] = p[h] * 4.0 * hr * hc / ((h + 2.0) * (h + 1.0));
      sum += p[h + 2];
    }
  }
  for (auto & v : p)
    v /= sum;
  double hi = 0.0;
  for (int i = obs_hets; i <= rare; ++i)
    hi += p[i];
  return hi > 1.0 ? 1.0 : hi;
}

// the two logistic models of include/graphtyper/typer/logistic_constants.hpp
double model_logf(double abhom, double cr_by_seqdepth, double mq, double pass_ratio, double gt_yield, double qd, long abhet_bin, long sbalt_bin)
{
  static const double W_ABHET[11] = {-6.03446, -6.03446, -1.35948, -0.84956, -0.28956, 0.0, -1.05013, -1.35024, -1.34475, -3.74512, -3.74512};
  static const double W_SBALT[11] = {-0.32486, -0.32486, -0.25342, -0.32696, 0.02442, 0.0, -0.33522, -0.41332, -0.74043, -1.60844, -1.60844};
  double const pwr = -29.28908 + abhom * 23.12909 + cr_by_seqdepth * -10.22658 + mq * 0.01024 + pass_ratio * 0.85320 + gt_yield * 4.91178 +
                     qd * 0.23215 + W_ABHET[abhet_bin] + W_SBALT[sbalt_bin];
  return 1.0 / (1.0 + std::max(0.0, std::exp(-pwr)));
}

double model_aa_score(double abhom, double sb, double mm, long sd, double qd, double cr, long mq)
{
  static const double W_ABHOM[5] = {0.0, 1.304140117, 1.681221065, 2.214801195, 3.930106559};
  int const bin = abhom <= 0.85 ? 0 : abhom <= 0.94 ? 1 : abhom <= 0.98 ? 2 : abhom <= 0.99 ? 3 : 4;
  mq = std::min(mq, 60l);
  double const pwr = -6.347426707 + W_ABHOM[bin] + sb * -0.25233400 + mm * -0.04129973 + sd * 0.014572295 + qd * 0.065221319 + cr * -0.01934834 +
                     mq * 0.055973424;
  return 1.0 / (1.0 + std::exp(-pwr));
}

struct Info // the reference's std::map<std::string, std::string> of a record: keys (string literals here) in the order they are made
{
  std::vector<std::pair<char const *, std::string>> kv;
  Info() { kv.reserve(48); }
  std::string & operator[](char const * k)
  {
    kv.emplace_back(k, std::string());
    return kv.back().second;
  }
  std::string const * find(char const * k) const
  {
    for (auto const & e : kv)
      if (std::strcmp(e.first, k) == 0)
        return &e.second;
    return nullptr;
  }
  // The places of kv in the map's order (std::string's: bytes as unsigned).  One record after the other makes the same keys in the
  // same order, so the order found for the record before is checked (the literals' addresses) and used again; sorting the pairs
  // themselves was a fifth of a record's time.
  std::vector<uint16_t> const & sorted() const
  {
    static thread_local std::vector<char const *> keys;
    static thread_local std::vector<uint16_t> order;
    bool same = keys.size() == kv.size();
    for (size_t i = 0; same && i < kv.size(); ++i)
      same = keys[i] == kv[i].first;
    if (!same)
    {
      keys.resize(kv.size());
      order.resize(kv.size());
      for (size_t i = 0; i < kv.size(); ++i)
      {
        keys[i] = kv[i].first;
        order[i] = static_cast<uint16_t>(i);
      }
      std::stable_sort(order.begin(), order.end(), [&](uint16_t a, uint16_t b) { return std::strcmp(keys[a], keys[b]) < 0; });
    }
    return order;
  }
};

// Variant::determine_variant_type (variant.cpp:1430-1520) for the alleles of a graph without SV alleles
const char * variant_type(std::vector<std::pair<const char *, uint32_t>> const & seqs)
{
  size_t longer = 0;
  for (auto const & s : seqs)
    longer += s.second > 1;
  if (longer == 0)
    return "SG";
  if (seqs.size() - longer == 1)
    return "IG";
  if (seqs.size() - longer == 2 && seqs.back().second == 1 && seqs.back().first[0] == '*')
    return "IG";
  return "XG";
}
} // namespace


// ---------------------------------------------------------------------------------------------------------------
// The calls of an SV graph (cfg5, `genotype_sv`).  What the reference does between the haplotypes and the VCF text there:
//   the pool's writer   src/utilities/hts_parallel_reader.cpp:984-1020   Vcf::add_haplotype per site -> reformat_sv_vcf_records
//                                                                        -> sort by (position, alleles) -> stats.clear()
//   reformat            src/graph/sv.cpp:117-655                         a site with SV alleles becomes one bi-allelic record per
//                                                                        SV allele (make_bi_allelic_call, sample_call.cpp:189-253),
//                                                                        moved to the SV's own position, REF "N", ALT <TYPE:SVSIZE=n:MODEL>;
//                                                                        deletions (and the second breakpoint of a duplication)
//                                                                        also get a record from the read depth inside / beside them
//                                                                        (make_call_based_on_coverage, :255-385) and an AGGREGATED
//                                                                        record that takes, per sample, the more confident of the
//                                                                        two calls; the second breakpoint of an insertion / inversion
//                                                                        is aggregated with the first
//   the merge           src/utilities/genotype_sv.cpp:125 -> vcf_merge_and_break(force_no_break_down), vcf_operations.cpp:480-700:
//                                                                        normalize -> generate_infos -> a record whose every
//                                                                        alternative allele is uncalled is dropped
//   the writer          src/typer/vcf.cpp:1161-1275                      order, duplicates, ".<n>" behind the ID of a record that
//                                                                        shares position and type with the one in front of it
// Built as a small pipeline over one struct (SvRecord); the per-site statistics of the alignment (VarStats) play no part here:
// the reference clears them before it writes.  Not built: a site that mixes SV and non-SV alleles (find_variant_sequences,
// variant.cpp:1880-2240), breakend alleles that start with a tag: GTX_ERR_UNSUPPORTED.
// ---------------------------------------------------------------------------------------------------------------
namespace
{
enum SvKind { K_NOT_SV, K_DEL, K_DEL_ALU, K_DUP, K_INS, K_INS_ALU, K_INV, K_BND, K_OTHER };

struct SvEntry // one line of gtx_graph_sv_table
{
  SvKind kind = K_NOT_SV;
  long begin = 0, length = 0, size = 0, end = 0, n_clusters = 0, num_merged_svs = -1, or_start = -1, or_end = -1, related = -1;
  std::string model, old_id, inv, seq, hom_seq, ins_seq, ins_left, ins_right, original_alt;
  char const * type_name() const
  {
    static char const * const N[] = {"SV", "DEL", "DEL:ME:ALU", "DUP", "INS", "INS:ME:ALU", "INV", "BND", "SV"};
    return N[kind];
  }
  std::string allele() const // SV::get_allele (sv.cpp:51-64)
  {
    std::string a = "<";
    a += type_name();
    a += ":SVSIZE=";
    if (size > 0)
      put_u(a, static_cast<uint64_t>(size));
    else
    {
      put_u(a, ins_left.size() + ins_right.size());
      a += '+';
    }
    a += '>';
    return a;
  }
};

bool parse_sv_table(char const * text, std::vector<SvEntry> & out, std::string & why)
{
  static char const * const KINDS[] = {"NOT_SV", "DEL", "DEL_ALU", "DUP", "INS", "INS_ALU", "INV", "BND", "OTHER"};
  for (char const * p = text ? text : ""; *p;)
  {
    char const * eol = std::strchr(p, '\n');
    std::string const line(p, eol ? static_cast<size_t>(eol - p) : std::strlen(p));
    p = eol ? eol + 1 : p + line.size();
    if (line.empty())
      continue;
    std::vector<std::string> f;
    for (size_t i = 0; i <= line.size();)
    {
      size_t const e = std::min(line.find('\t', i), line.size());
      f.push_back(line.substr(i, e - i));
      if (f.back() == ".")
        f.back().clear();
      i = e + 1;
    }
    if (f.size() != 20)
    {
      why = "a line of the SV table has " + std::to_string(f.size()) + " fields (20 expected)";
      return false;
    }
    SvEntry sv;
    int k = -1;
    for (int i = 0; i < 9; ++i)
      if (f[0] == KINDS[i])
        k = i;
    if (k < 0)
    {
      why = "unknown SV type '" + f[0] + "' in the SV table";
      return false;
    }
    sv.kind = static_cast<SvKind>(k);
    long * const nums[] = {&sv.begin, &sv.length, &sv.size, &sv.end, &sv.n_clusters, &sv.num_merged_svs, &sv.or_start, &sv.or_end, &sv.related};
    for (int i = 0; i < 9; ++i)
      *nums[i] = std::atol(f[2 + i].c_str());
    sv.model = f[11];
    sv.old_id = f[12];
    sv.inv = f[13];
    sv.seq = f[14];
    sv.hom_seq = f[15];
    sv.ins_seq = f[16];
    sv.ins_left = f[17];
    sv.ins_right = f[18];
    sv.original_alt = f[19];
    out.push_back(std::move(sv));
  }
  return true;
}

struct SvCall // SampleCall (include/graphtyper/typer/sample_call.hpp)
{
  std::vector<uint8_t> pl;
  std::vector<uint16_t> ad;
  uint16_t ref_total = 0, alt_total = 0;
  uint8_t ambiguous = 0, alt_proper_pair = 0;
  int filter = -1; // -1: not judged yet (SampleCall::check_filter judges by GQ the first time it is asked)

  uint32_t unique_depth() const { return std::accumulate(ad.begin(), ad.end(), 0u); }
  void genotype(uint32_t & a, uint32_t & b) const // the first genotype with PL 0, (0, 0) when there is none
  {
    size_t i = 0;
    for (uint32_t y = 0; y < ad.size(); ++y)
      for (uint32_t x = 0; x <= y; ++x, ++i)
        if (i < pl.size() && pl[i] == 0)
        {
          a = x;
          b = y;
          return;
        }
    a = b = 0;
  }
  long gq() const // the second smallest PL; 0 when two genotypes share PL 0
  {
    bool zero = false;
    long next = 255;
    for (uint8_t p : pl)
    {
      if (p == 0)
      {
        if (zero)
          return 0;
        zero = true;
      }
      else if (p < next)
        next = p;
    }
    return next;
  }
  int judged()
  {
    if (filter < 0)
    {
      long const q = gq();
      filter = q >= 30 ? 0 : q >= 20 ? 1 : q >= 10 ? 2 : 3;
    }
    return filter;
  }
  uint8_t lowest_pl_without(uint32_t allele) const
  {
    uint8_t m = 255;
    size_t i = 0;
    for (uint32_t y = 0; y < ad.size(); ++y)
      for (uint32_t x = 0; x <= y; ++x, ++i)
        if (x != allele && y != allele && pl[i] < m)
          m = pl[i];
    return m;
  }
  void set_pl3(uint64_t g00, uint64_t g01, uint64_t g11)
  {
    pl.assign(3, 0);
    pl[0] = static_cast<uint8_t>(std::min<uint64_t>(255, g00));
    pl[1] = static_cast<uint8_t>(std::min<uint64_t>(255, g01));
    pl[2] = static_cast<uint8_t>(std::min<uint64_t>(255, g11));
  }
};

struct SvRecord // Variant, as far as this path uses it
{
  uint32_t pos = 0;
  std::vector<std::string> alleles;
  std::vector<SvCall> calls;
  std::map<std::string, std::string> info;

  bool has_sv_allele() const // Variant::is_sv (variant.cpp:1098-1118)
  {
    for (size_t a = 1; a < alleles.size(); ++a)
      if (alleles[a].size() >= 5 && (alleles[a][0] == '<' || (alleles[a].size() > 100 && alleles[a].find('<') != std::string::npos)))
        return true;
    return false;
  }
  std::string type_code() const // Variant::determine_variant_type (variant.cpp:1430-1520)
  {
    enum { NONE, DEL, DUP, INS, BND, OTHER } sv = NONE;
    size_t longer = 0;
    for (auto const & s : alleles)
    {
      if (s.size() <= 1)
        continue;
      if (s.size() > 4 && s[0] == '<')
      {
        std::string const t = s.substr(1, 3);
        sv = (t == "DEL" && (sv == NONE || sv == DEL)) ? DEL : (t == "DUP" && (sv == NONE || sv == DUP)) ? DUP : (t == "INS" && (sv == NONE || sv == INS)) ? INS : OTHER;
      }
      else if (s.find_first_of("[]") != std::string::npos)
        sv = (sv == NONE || sv == BND) ? BND : OTHER;
      else
        ++longer;
    }
    switch (sv)
    {
    case DEL: return "DG";
    case DUP: return "UG";
    case INS: return "FG";
    case BND: return "OG";
    case OTHER: return "TG";
    default: break;
    }
    if (longer == 0)
      return "SG";
    if (alleles.size() - longer == 1 || (alleles.size() - longer == 2 && alleles.back() == "*"))
      return "IG";
    return "XG";
  }
  uint64_t qual() const
  {
    uint64_t q = 0;
    for (auto const & c : calls)
      if (!c.pl.empty())
        q += c.pl[0];
    return q;
  }
};

// make_bi_allelic_call (sample_call.cpp:189-253): the call of one site reduced to reference against alternative allele `aa`
SvCall reduce_to_two_alleles(SvCall const & full, size_t aa)
{
  if (full.ad.size() == 2)
    return full;
  SvCall c;
  c.ambiguous = full.ambiguous;
  c.ref_total = full.ref_total;
  c.alt_total = full.alt_total;
  c.alt_proper_pair = full.alt_proper_pair;
  int const ref_cov = full.ad[0];
  // what was ambiguous between alternative alleles only is not ambiguous any more
  int amb_alt = std::min<int>(c.ambiguous, ref_cov + c.ambiguous - c.ref_total);
  c.ambiguous = static_cast<uint8_t>(c.ambiguous - amb_alt);
  int alt_cov = c.alt_total - c.ambiguous;
  for (size_t a = 1; a < full.ad.size(); ++a)
    if (a != aa + 1)
    {
      alt_cov -= full.ad[a];
      c.alt_total = static_cast<uint16_t>(std::max(0, static_cast<int>(c.alt_total) - static_cast<int>(full.ad[a])));
      c.alt_proper_pair = static_cast<uint8_t>(std::max(0, static_cast<int>(c.alt_proper_pair) - static_cast<int>(full.ad[a])));
    }
  c.ad = {static_cast<uint16_t>(ref_cov), static_cast<uint16_t>(std::max(alt_cov, 0))};
  int const not_proper = c.ad[1] > c.alt_proper_pair ? c.ad[1] - c.alt_proper_pair : 0, proper = c.ad[1] - not_proper;
  uint64_t const g00 = 24ull * proper + 12ull * not_proper, g01 = 3ull * (static_cast<uint64_t>(c.ad[0]) + c.ad[1]), g11 = 24ull * c.ad[0];
  uint64_t const low = std::min(g00, std::min(g01, g11));
  c.set_pl3(g00 - low, g01 - low, g11 - low);
  return c;
}

// make_call_based_on_coverage (sample_call.cpp:255-385): a deletion (duplication, inversion) judged by the read depth at up to
// 101 points inside it against 101 points within a kilobase on either side; `depth`: one sample's finalised track, position
// `first` at index 0
bool call_from_depth(SvEntry const & sv, uint32_t const * depth, uint32_t n_depth, uint32_t first, SvCall & call)
{
  auto at = [&](long pos) -> long
  {
    if (n_depth == 0)
      return 0;
    long const i = pos < static_cast<long>(first) ? 0 : pos - static_cast<long>(first);
    return static_cast<long>(std::min<uint32_t>(depth[std::min<long>(i, static_cast<long>(n_depth) - 1)], 0xFFFFu));
  };
  long const begin = sv.begin, end = begin + std::min<long>(sv.size, 190000);
  long constexpr POINTS = 101, STEP = 20;
  std::vector<long> inside, outside;
  {
    long const span = end - begin - 2 * STEP;
    long n = std::min(POINTS, span);
    if (n % 2 == 0)
      --n;
    for (long i = 1; i <= n; ++i)
      inside.push_back(at((i * span) / (n + 1) + begin + STEP));
  }
  for (long i = 1; i <= POINTS / 2 + 1; ++i)
    outside.push_back(at(std::max(begin - i * STEP, 0l)));
  if (sv.size < 190000)
    for (long i = 1; i <= POINTS / 2; ++i)
      outside.push_back(at(std::max(end + i * STEP, 0l)));
  if (inside.empty() || outside.empty())
    return false; // (an SV of 40 bases or fewer: the reference takes the median of nothing)
  auto median = [](std::vector<long> & v)
  {
    std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
    return v[v.size() / 2];
  };
  long const m_out = median(outside), m_in = median(inside);
  auto u16 = [](long v) { return static_cast<uint16_t>(std::max(0l, std::min(0xFFFFl, v))); };
  if (sv.kind == K_DEL || sv.kind == K_DEL_ALU)
    call.ad = {u16(m_in), u16(m_out - m_in)};
  else
  {
    double const centre = static_cast<double>(m_out + m_in) / 2.0;
    long const gain = m_in - m_out;
    if (gain <= 0)
      call.ad = {u16(std::lround(centre)), 0};
    else if (gain >= 2 * m_in)
      call.ad = {0, u16(std::lround(centre))};
    else if (m_out == 0)
      // (no depth at all on the flanks: the reference divides by it, sample_call.cpp:338 -- (1 - inf) * centre, rounded and clamped,
      //  is 0 on the machines it runs on; said here instead of left to lround(-inf))
      call.ad = {0, u16(static_cast<long>(centre))};
    else
    {
      uint16_t const r = u16(std::lround((1.0 - static_cast<double>(gain) / static_cast<double>(m_out)) * centre));
      call.ad = {r, u16(static_cast<long>(centre - r))};
    }
  }
  uint64_t g00 = 12ull * call.ad[1], g01 = 3ull * (static_cast<uint64_t>(call.ad[0]) + call.ad[1]), g11 = 12ull * call.ad[0];
  uint64_t const low = std::min(g00, std::min(g01, g11));
  g00 -= low;
  g01 -= low;
  g11 -= low;
  // (the model is trusted less for short SVs and more for long ones)
  uint64_t const num = sv.size <= 100 ? 2 : sv.size > 1000 ? 3 : 1, den = sv.size <= 100 ? 3 : sv.size > 1000 ? 2 : 1;
  call.set_pl3(g00 * num / den, g01 * num / den, g11 * num / den);
  return true;
}

// Variant::scan_calls + generate_infos (variant.cpp:230-1096) as they run in an SV graph once the alignment statistics have been
// cleared: the INFO fields that are functions of the calls.  Returns, per alternative allele, whether any sample was called with it.
std::vector<char> summarise(SvRecord & r)
{
  size_t const n_alleles = r.alleles.size();
  struct PerAllele
  {
    uint64_t qd_qual = 0, qd_depth = 0;
    uint32_t ac = 0, pass_ac = 0, hom_ref = 0, het = 0, hom_alt = 0;
    uint16_t max_support = 0;
    double max_support_ratio = 0.0;
  };
  std::vector<PerAllele> al(n_alleles);
  uint32_t genotyped = 0, passed = 0;
  uint64_t seqdepth = 0, het_a = 0, het_b = 0, hom_a = 0, hom_b = 0;
  uint8_t max_alt_pp = 0;
  long qd_qual = 0, qd_depth = 0;
  for (SvCall & c : r.calls)
  {
    uint32_t g1, g2;
    c.genotype(g1, g2);
    uint32_t const unique = c.unique_depth();
    if (!c.pl.empty() && c.pl[0] > 0)
    {
      long const alt_depth = std::min<long>(10, static_cast<long>(unique - c.ad[0] + c.ambiguous));
      if (alt_depth > 0)
      {
        qd_qual += std::min<long>(25 * alt_depth, c.pl[0]);
        qd_depth += alt_depth;
      }
      for (int k = 0; k < 2; ++k)
      {
        uint32_t const a = k ? g2 : g1;
        if ((k == 0 && a == 0) || (k == 1 && g1 == g2))
          continue;
        long const d = std::min<long>(10, static_cast<long>(c.ad[a]) + c.ambiguous);
        if (d > 0)
        {
          al[a].qd_qual += static_cast<uint64_t>(std::min<long>(25 * d, c.lowest_pl_without(a)));
          al[a].qd_depth += static_cast<uint64_t>(d);
        }
      }
    }
    max_alt_pp = std::max(max_alt_pp, c.alt_proper_pair);
    for (uint32_t a = 1; a < n_alleles; ++a)
    {
      al[a].max_support = std::max(al[a].max_support, c.ad[a]);
      if (unique > 0)
        al[a].max_support_ratio = std::max(al[a].max_support_ratio, static_cast<double>(c.ad[a]) / static_cast<double>(unique));
      if (g1 == a || g2 == a)
        ++(g1 == g2 ? al[a].hom_alt : al[a].het);
      else
        ++al[a].hom_ref;
    }
    int const verdict = c.judged();
    genotyped += std::any_of(c.pl.begin(), c.pl.end(), [](uint8_t p) { return p != 0; });
    passed += verdict == 0;
    if (g1 != g2)
    {
      het_a += c.ad[g1];
      het_b += c.ad[g2];
    }
    else
    {
      hom_a += c.ad[g1];
      hom_b += unique - c.ad[g1];
    }
    seqdepth += unique + c.ambiguous;
    ++al[g1].ac;
    ++al[g2].ac;
    if (verdict == 0)
    {
      ++al[g1].pass_ac;
      ++al[g2].pass_ac;
    }
  }
  auto & info = r.info;
  auto list = [&](char const * key, auto && value)
  {
    std::string s;
    for (size_t a = 1; a < n_alleles; ++a)
    {
      if (a > 1)
        s += ',';
      value(s, al[a]);
    }
    info[key] = s;
  };
  info["RefLen"] = std::to_string(r.alleles[0].size());
  {
    auto const e = info.find("END"); // END is never in front of POS
    if (e != info.end())
      e->second = std::to_string(std::max(std::strtol(e->second.c_str(), nullptr, 10), static_cast<long>(r.pos)));
  }
  list("MaxAAS", [](std::string & s, PerAllele const & p) { put_u(s, p.max_support); });
  list("MaxAASR", [](std::string & s, PerAllele const & p) { put_g(s, p.max_support_ratio, 4); });
  list("NHomRef", [](std::string & s, PerAllele const & p) { put_u(s, p.hom_ref); });
  list("NHet", [](std::string & s, PerAllele const & p) { put_u(s, p.het); });
  list("NHomAlt", [](std::string & s, PerAllele const & p) { put_u(s, p.hom_alt); });
  list("PexcessHet", [](std::string & s, PerAllele const & p) { put_g(s, excess_het_p(static_cast<int>(p.het), static_cast<int>(p.hom_ref), static_cast<int>(p.hom_alt)), 6); });
  if (r.has_sv_allele())
    info["MaxAltPP"] = std::to_string(static_cast<unsigned>(max_alt_pp));
  list("AC", [](std::string & s, PerAllele const & p) { put_u(s, p.ac); });
  info["AN"] = std::to_string(2ull * genotyped);
  list("AF", [&](std::string & s, PerAllele const & p)
       {
         if (genotyped > 0)
           put_g(s, static_cast<double>(p.ac) / static_cast<double>(2 * genotyped), 4);
         else
           s += "0.0";
       });
  list("PASS_AC", [](std::string & s, PerAllele const & p) { put_u(s, p.pass_ac); });
  info["PASS_AN"] = std::to_string(2ull * passed);
  if (genotyped > 0)
  {
    std::string s;
    put_g(s, static_cast<double>(passed) / static_cast<double>(genotyped), 4);
    info["PASS_ratio"] = s;
  }
  info["SeqDepth"] = std::to_string(seqdepth);
  auto balance = [&](char const * key, uint64_t num, uint64_t other)
  {
    std::string s;
    uint32_t const total = static_cast<uint32_t>(num + other);
    if (total > 0)
      put_g(s, static_cast<double>(static_cast<uint32_t>(num)) / static_cast<double>(total), 4);
    else
      s = "-1";
    info[key] = s;
  };
  balance("ABHet", het_b, het_a);
  balance("ABHom", hom_a, hom_b);
  info["VarType"] = r.type_code();
  {
    std::string s;
    put_g(s, qd_depth == 0 ? 0.0 : static_cast<double>(qd_qual) / static_cast<double>(qd_depth), 4);
    info["QD"] = s;
  }
  // (what generate_infos computes next -- SB, MQ, the per-allele balances -- it erases again in an SV graph: variant.cpp:861-884)
  std::vector<char> called(n_alleles - 1);
  for (size_t a = 1; a < n_alleles; ++a)
    called[a - 1] = al[a].ac > 0;
  return called;
}

// make_variant_with_combined_calls (sv.cpp:232-308): per sample the more confident of two calls, and a verdict on their agreement
SvRecord aggregate(SvRecord const & first, SvRecord const & second)
{
  SvRecord out = first;
  for (size_t i = 0; i < first.calls.size(); ++i)
  {
    SvCall const & a = first.calls[i];
    SvCall const & b = second.calls[i];
    SvCall & c = out.calls[i];
    uint32_t a1, a2, b1, b2;
    a.genotype(a1, a2);
    b.genotype(b1, b2);
    long const gq_a = a.gq(), gq_b = b.gq();
    if (gq_b > gq_a)
      c = b;
    long const high = gq_b, low = gq_a; // (sv.cpp:246-261: "max_gq" is the second call's GQ and "min_gq" the first call's, whichever is larger)
    if (a.filter > 0 && b.filter > 0)
      c.filter = 3;
    else if (a.filter > 0)
      c.filter = a.filter;
    else if (b.filter > 0)
      c.filter = b.filter;
    else if (a.unique_depth() >= 10 && b.unique_depth() >= 10)
    {
      uint32_t c1, c2;
      c.genotype(c1, c2);
      size_t const idx = c1 + static_cast<size_t>(c2 + 1) * c2 / 2;
      bool const agree = c1 == a1 && c2 == a2 && c1 == b1 && c2 == b2;
      c.filter = (agree && low > 10) ? 0 : (high > 40 && static_cast<int>(a.pl[idx]) + static_cast<int>(b.pl[idx]) <= 20) ? 0 : high > 30 ? 1 : 2;
    }
    else
      c.filter = 3;
  }
  summarise(out);
  return out;
}

// left-alignment of a record whose alleles start with the same base (Variant::normalize, variant.cpp:1256-1315); returns how
// far it moved.  `ref` is the region's reference from position `first` on.
template <class Rec>
long left_align(Rec & r, std::string const & ref, uint32_t first)
{
  auto & al = r.alleles;
  if (al.size() < 2)
    return 0;
  for (size_t i = 0; i < al.size(); ++i)
    if (al[i].empty() || al[i][0] != al[0][0] || (i > 0 && al[i] == al[0]))
      return 0;
  auto strip_suffix = [&]()
  {
    while (al[0].size() > 1)
    {
      for (size_t a = 1; a < al.size(); ++a)
        if (al[a].size() <= 1 || al[a].back() != al[0].back())
          return;
      for (auto & s : al)
        s.pop_back();
    }
  };
  auto same_last = [&]()
  {
    for (size_t a = 1; a < al.size(); ++a)
      if (al[a].back() != al[0].back())
        return false;
    return true;
  };
  strip_suffix();
  long moved = 0;
  while (same_last())
  {
    // the base in front of the record (add_base_in_front): inside the region's reference and one of A, C, G, T
    if (r.pos <= first || r.pos - 1 - first >= ref.size())
      break;
    char const b = ref[r.pos - 1 - first];
    if (b != 'A' && b != 'C' && b != 'G' && b != 'T')
      break;
    for (auto & s : al)
      if (s != "*")
        s.insert(s.begin(), b);
    --r.pos;
    ++moved;
    strip_suffix();
  }
  while (al[0].size() > 1) // remove_common_prefix, not keeping a match
  {
    bool all = true;
    for (size_t a = 1; a < al.size(); ++a)
      all = all && al[a].size() > 1 && al[a][0] == al[0][0];
    if (!all)
      break;
    ++r.pos;
    for (auto & s : al)
      s.erase(s.begin());
  }
  return moved;
}

int sv_graph_records(gtx_ctx const * c, gtx_vcf_request const * rq, std::string & text)
{
  gtx::HostGraph const & g = c->graph;
  uint32_t const nh = g.n_hap, ns = rq->n_samples;
  std::vector<SvEntry> svs;
  std::string why;
  if (!parse_sv_table(rq->sv_table, svs, why))
  {
    gtx::g_last_error = "gtx_vcf_records: " + why;
    return GTX_ERR_ARG;
  }
  uint32_t const first_pos = g.ref_order.empty() ? 1u : g.ref_order.front();
  if (rq->ref_depth == nullptr && ns > 0)
  {
    gtx::g_last_error = "gtx_vcf_records: the calls of an SV graph need the reference-depth track (gtx_vcf_request::ref_depth)";
    return GTX_ERR_ARG;
  }
  // ---- one record per site: Vcf::add_haplotype
  std::vector<SvRecord> sites(nh);
  for (uint32_t h = 0; h < nh; ++h)
  {
    uint32_t const cnum = g.ref_nvar[h], v0 = g.ref_first_var[h], n_tri = cnum * (cnum + 1) / 2;
    SvRecord & r = sites[h];
    r.pos = g.var_order[v0];
    for (uint32_t a = 0; a < cnum; ++a)
      r.alleles.emplace_back(g.dna.data() + g.var_dna[v0 + a], g.var_len[v0 + a]);
    r.calls.resize(ns);
    for (uint32_t s = 0; s < ns; ++s)
    {
      SvCall & call = r.calls[s];
      const uint8_t * pl = rq->phred + static_cast<uint64_t>(s) * g.total_tri + g.tri_off[h];
      const uint32_t * cov = rq->gt_cov + static_cast<uint64_t>(s) * g.total_allele + g.allele_off[h];
      gtx_sample_call const & sc = rq->calls[static_cast<uint64_t>(s) * nh + h];
      call.pl.assign(pl, pl + n_tri);
      for (uint32_t a = 0; a < cnum; ++a)
        call.ad.push_back(static_cast<uint16_t>(std::min<uint32_t>(cov[a], 0xFFFFu)));
      call.ref_total = sc.ref_total_depth;
      call.alt_total = sc.alt_total_depth;
      call.ambiguous = sc.ambiguous_depth;
      call.alt_proper_pair = sc.alt_proper_pair_depth;
    }
  }
  // ---- reformat: the sites that carry SV alleles become the SVs' own records
  std::vector<SvRecord> made, untouched;
  std::map<long, size_t> first_breakpoint; // SV id -> the record (in `made`) of the breakpoint that names it as its partner
  auto finish = [&](SvRecord && r, SvEntry const & sv, std::string const & model) // add_sv_to_new_vars_vector (sv.cpp:310-391)
  {
    if (sv.kind == K_BND)
      r.alleles[1] = sv.original_alt;
    else if (!model.empty())
    {
      r.alleles[1].back() = ':';
      r.alleles[1] += model;
      r.alleles[1] += '>';
    }
    auto & info = r.info;
    info["SVTYPE"] = sv.type_name();
    info["END"] = std::to_string(std::max(sv.begin, sv.end));
    if (sv.length != 0)
    {
      info["SVSIZE"] = std::to_string(sv.size);
      info["SVLEN"] = std::to_string(sv.length);
    }
    if (!model.empty())
      info["SVMODEL"] = model;
    if (sv.or_start != -1)
    {
      info["ORSTART"] = std::to_string(sv.or_start);
      info["OREND"] = std::to_string(sv.or_end);
    }
    auto put = [&](char const * key, std::string const & v)
    {
      if (!v.empty())
        info[key] = v;
    };
    put("SEQ", sv.seq);
    if (sv.n_clusters > 0)
      info["NCLUSTERS"] = std::to_string(sv.n_clusters);
    if (sv.num_merged_svs >= 0)
      info["NUM_MERGED_SVS"] = std::to_string(sv.num_merged_svs);
    if (sv.old_id != ".")
      put("OLD_VARIANT_ID", sv.old_id);
    put("HOMSEQ", sv.hom_seq);
    put("SVINSSEQ", sv.ins_seq);
    put("LEFT_SVINSSEQ", sv.ins_left);
    put("RIGHT_SVINSSEQ", sv.ins_right);
    if (sv.kind == K_INV && !sv.inv.empty())
    {
      if (sv.inv == "INV3" || sv.inv == "BOTH")
        info["INV3"] = "";
      if (sv.inv == "INV5" || sv.inv == "BOTH")
        info["INV5"] = "";
    }
    made.push_back(std::move(r));
  };
  for (uint32_t h = 0; h < nh; ++h)
  {
    SvRecord const & site = sites[h];
    std::vector<long> id(site.alleles.size() - 1, -1);
    bool any = false, plain = false;
    for (size_t a = 1; a < site.alleles.size(); ++a)
    {
      size_t const at = site.alleles[a].find('<');
      if (at != std::string::npos && site.alleles[a].size() - at > 11) // "<SV:nnnnnnn>"
      {
        id[a - 1] = std::atol(site.alleles[a].substr(at + 4, 7).c_str());
        if (id[a - 1] < 0 || id[a - 1] >= static_cast<long>(svs.size()))
        {
          gtx::g_last_error = "gtx_vcf_records: an allele names SV " + std::to_string(id[a - 1]) + ", the SV table has " + std::to_string(svs.size());
          return GTX_ERR_ARG;
        }
        any = true;
      }
      else
        plain = true;
    }
    if (!any)
    {
      untouched.push_back(site);
      continue;
    }
    if (plain)
    {
      gtx::g_last_error = "gtx_vcf_records: a site with SV and non-SV alleles is not supported";
      return GTX_ERR_UNSUPPORTED;
    }
    for (size_t aa = 0; aa < id.size(); ++aa)
    {
      SvEntry const & sv = svs[static_cast<size_t>(id[aa])];
      SvRecord r; // make_new_sv_var (sv.cpp:176-230) + add_sv_variant (:393-510)
      r.pos = static_cast<uint32_t>(sv.begin);
      r.alleles = {site.alleles[0], site.alleles[aa + 1]};
      r.info = site.info;
      for (SvCall const & call : site.calls)
        r.calls.push_back(reduce_to_two_alleles(call, aa));
      if (sv.n_clusters > 0)
        r.info["NCLUSTERS"] = std::to_string(sv.n_clusters);
      if (sv.num_merged_svs > 0)
        r.info["NUM_MERGED_SVS"] = std::to_string(sv.num_merged_svs);
      r.info["SV_ID"] = std::to_string(id[aa]);
      if (sv.related >= 0)
        r.info["RELATED_SV_ID"] = std::to_string(sv.related);
      if (sv.kind != K_BND)
        r.alleles = {"N", sv.allele()};
      else if (r.alleles[1].size() > 1 && r.alleles[1][1] == '<')
      {
        gtx::g_last_error = "gtx_vcf_records: breakend alleles that start with a tag are not supported";
        return GTX_ERR_UNSUPPORTED;
      }
      if (sv.kind == K_DUP && (sv.model == "BREAKPOINT1" || sv.model == "BREAKPOINT2")) // (a duplication's breakpoint: a third of the reads show it)
        for (SvCall & call : r.calls)
        {
          uint64_t const g00 = 25ull * call.ad[1];
          uint64_t const g01 = static_cast<uint64_t>(0.499999999 + 4.77121255 * static_cast<double>(call.ad[1]) + 1.76091259 * static_cast<double>(call.ad[0]));
          uint64_t const g11 = 3ull * (static_cast<uint64_t>(call.ad[0]) + call.ad[1]);
          uint64_t const low = std::min(g00, std::min(g01, g11));
          call.set_pl3(g00 - low, g01 - low, g11 - low);
        }
      auto const partner = first_breakpoint.find(id[aa]);
      bool const second = partner != first_breakpoint.end();
      if ((sv.kind == K_INS || sv.kind == K_INV) && second)
      {
        SvRecord const other = made[partner->second];
        finish(aggregate(r, other), sv, "AGGREGATED");
      }
      auto depth_record = [&](SvRecord & by_depth) -> bool
      {
        by_depth = r;
        for (uint32_t s = 0; s < ns; ++s)
        {
          SvCall call;
          if (!call_from_depth(sv, rq->ref_depth + static_cast<uint64_t>(s) * (rq->ref_depth_len + 1u), rq->ref_depth_len, first_pos, call))
            return false;
          by_depth.calls[s] = call;
        }
        return true;
      };
      if (g.is_sv_graph && (sv.kind == K_DEL || sv.kind == K_DEL_ALU || (sv.kind == K_DUP && second)))
      {
        SvRecord by_depth;
        if (!depth_record(by_depth))
        {
          gtx::g_last_error = "gtx_vcf_records: SV " + std::to_string(id[aa]) + " is too short for the coverage model (40 bases or fewer)";
          return GTX_ERR_UNSUPPORTED;
        }
        SvRecord both = aggregate(r, by_depth);
        if (sv.kind == K_DUP)
        {
          SvRecord const other = made[partner->second];
          both = aggregate(both, other);
        }
        finish(std::move(both), sv, "AGGREGATED");
        finish(std::move(by_depth), sv, "COVERAGE");
      }
      if (sv.related != -1)
        first_breakpoint[sv.related] = made.size();
      finish(std::move(r), sv, sv.model);
    }
  }
  for (SvRecord & r : untouched)
    made.push_back(std::move(r));
  std::stable_sort(made.begin(), made.end(), [](SvRecord const & a, SvRecord const & b) { return a.pos < b.pos || (a.pos == b.pos && a.alleles < b.alleles); });
  // ---- the merge: left-align, summarise, drop what nobody was called with
  std::string ref;
  for (size_t n = 0; n < g.ref_order.size(); ++n)
  {
    ref.append(g.dna.data() + g.ref_dna[n], g.ref_len[n]);
    if (g.ref_nvar[n] > 0)
      ref.append(g.dna.data() + g.var_dna[g.ref_first_var[n]], g.var_len[g.ref_first_var[n]]);
  }
  std::vector<SvRecord> kept;
  for (SvRecord & r : made)
  {
    if (left_align(r, ref, first_pos) > 200)
      continue;
    std::vector<char> const called = summarise(r);
    if (std::none_of(called.begin(), called.end(), [](char x) { return x != 0; }))
      continue;
    kept.push_back(std::move(r));
  }
  // ---- the writer's order: position, then deletions before insertions before equal lengths, then the alleles
  std::vector<size_t> order(kept.size());
  std::iota(order.begin(), order.end(), size_t(0));
  auto shape = [](SvRecord const & r) { return static_cast<int>(r.alleles[0].size() > r.alleles[1].size()) + 2 * static_cast<int>(r.alleles[0].size() == r.alleles[1].size()); };
  std::sort(order.begin(), order.end(), [&](size_t i, size_t j)
  {
    SvRecord const & a = kept[i];
    SvRecord const & b = kept[j];
    if (a.pos != b.pos)
      return a.pos < b.pos;
    if (shape(a) != shape(b))
      return shape(a) < shape(b);
    return a.alleles < b.alleles || (a.alleles == b.alleles && a.info.size() > b.info.size());
  });
  long dup = -1;
  for (size_t k = 0; k < order.size(); ++k)
  {
    SvRecord const & r = kept[order[k]];
    std::string suffix;
    if (k > 0)
    {
      SvRecord const & prev = kept[order[k - 1]];
      if (r.pos > rq->region_end)
        break;
      if (r.pos < rq->region_begin)
        continue;
      if (r.pos == prev.pos && r.alleles == prev.alleles)
        continue;
      if (r.pos == prev.pos && r.type_code() == prev.type_code())
        suffix = "." + std::to_string(++dup);
      else
        dup = -1;
    }
    else if (r.pos < rq->region_begin || r.pos > rq->region_end)
      continue;
    // ---- Vcf::write_record (vcf.cpp:767-1149)
    if ((ns > 0 && r.alleles.size() > 80) || std::accumulate(r.alleles.begin(), r.alleles.end(), size_t(0), [](size_t n, std::string const & s) { return n + s.size(); }) > 16000)
      continue;
    uint64_t const qual = r.qual();
    if (rq->filter_zero_qual && qual == 0)
      continue;
    bool const is_sv = r.has_sv_allele();
    text += rq->contig;
    text += '\t';
    put_u(text, r.pos);
    text += '\t';
    text += rq->contig;
    text += ':';
    put_u(text, r.pos);
    text += ':';
    text += r.type_code();
    if (rq->variant_suffix_id && rq->variant_suffix_id[0])
    {
      text += '[';
      text += rq->variant_suffix_id;
      text += ']';
    }
    text += suffix;
    for (size_t a = 0; a < r.alleles.size(); ++a)
    {
      text += a < 2 ? '\t' : ',';
      text += r.alleles[a];
    }
    text += '\t';
    put_u(text, qual);
    text += '\t';
    if (ns == 0)
      text += '.';
    else
    {
      size_t const before = text.size();
      auto fail = [&](char const * name)
      {
        if (text.size() != before)
          text += ';';
        text += name;
      };
      auto num = [&](char const * k) { return std::stod(r.info.at(k)); };
      auto has = [&](char const * k) { return r.info.count(k) == 1; };
      if (is_sv)
      {
        if (has("QD") && num("QD") < 6.0)
          fail("LowQD");
        if (qual < 10)
          fail("LowQUAL");
        if (has("AN") && has("PASS_AC") && has("PASS_ratio") && std::stoi(r.info.at("AN")) >= 100 && (r.info.at("PASS_AC") == "0" || num("PASS_ratio") < 0.01))
          fail("LowPratio");
      }
      else
      {
        if (has("ABHet") && r.info.at("ABHet") != "-1" && num("ABHet") < 0.175)
          fail("LowABHet");
        if (has("ABHom") && r.info.at("ABHom") != "-1" && num("ABHom") < 0.85)
          fail("LowABHom");
        if (has("AN") && std::stoi(r.info.at("AN")) >= 6 && has("QD") && num("QD") < 6.0)
          fail("LowQD");
        if (qual < 10)
          fail("LowQUAL");
        if (has("AN") && has("PASS_ratio") && std::stoi(r.info.at("AN")) >= 500 && num("PASS_ratio") < 0.05)
          fail("LowPratio");
      }
      if (text.size() == before)
        text += "PASS";
    }
    text += '\t';
    if (r.info.empty())
      text += '.';
    bool first_kv = true;
    for (auto const & kv : r.info)
    {
      if (!first_kv)
        text += ';';
      first_kv = false;
      text += kv.first;
      if (!kv.second.empty())
      {
        text += '=';
        text += kv.second;
      }
    }
    if (ns)
    {
      text += is_sv ? "\tGT:FT:AD:MD:DP:RA:PP:GQ:PL" : "\tGT:AD:MD:DP:GQ:PL";
      for (SvCall const & call0 : r.calls)
      {
        SvCall call = call0;
        text += '\t';
        if (std::none_of(call.pl.begin(), call.pl.end(), [](uint8_t p) { return p != 0; }))
          text += "./.";
        else
        {
          uint32_t g1, g2;
          call.genotype(g1, g2);
          put_u(text, g1);
          text += '/';
          put_u(text, g2);
        }
        long const gq = call.gq();
        if (is_sv)
        {
          int const verdict = call.judged();
          text += verdict == 0 ? ":PASS" : ":FAIL";
          if (verdict != 0)
            put_u(text, static_cast<uint64_t>(verdict));
        }
        for (size_t a = 0; a < call.ad.size(); ++a)
        {
          text += a ? ',' : ':';
          put_u(text, call.ad[a]);
        }
        text += ':';
        put_u(text, call.ambiguous);
        text += ':';
        put_u(text, call.unique_depth() + call.ambiguous);
        if (is_sv)
        {
          text += ':';
          put_u(text, call.ref_total);
          text += ',';
          put_u(text, call.alt_total);
          text += ':';
          put_u(text, call.alt_proper_pair);
        }
        text += ':';
        put_u(text, std::min<uint16_t>(99, BINNED.v[gq]));
        for (size_t i = 0; i < call.pl.size(); ++i)
        {
          text += i ? ',' : ':';
          put_u(text, BINNED.v[call.pl[i]]);
        }
      }
    }
    text += '\n';
  }
  return GTX_OK;
}
} // namespace

namespace
{
// The records of the sites [h_begin, h_end) (Vcf::add_haplotype -> Variant::scan_calls / generate_infos -> Vcf::write_record).
// With `good`: nothing is written; (*good)[h] receives generate_infos' verdict on the site's alternative alleles (is_good_alt,
// variant.cpp:1040-1070) -- what vcf_merge_and_filter keeps of a site (gtx_vcf_sites below).
int write_sites(gtx_ctx const * c, gtx_vcf_request const * rq, uint32_t h_begin, uint32_t h_end, std::string & text, std::string & error,
                std::vector<std::vector<int8_t>> * good);
} // namespace

extern "C" int gtx_vcf_records(const gtx_ctx * c, const gtx_vcf_request * rq, char * out, uint64_t cap, uint64_t * len)
{
  if (!c || !rq || !len || (cap && !out) || !rq->contig || !rq->gt_cov || !rq->stat_u64 || !rq->stat_u32 || !rq->phred || !rq->calls ||
      (rq->n_samples && !rq->sample_names))
    return GTX_ERR_ARG;
  gtx::HostGraph const & g = c->graph;
  uint32_t const nh = g.n_hap, ns = rq->n_samples;
  std::string text;
  text.reserve(static_cast<size_t>(nh) * (400 + 24 * static_cast<size_t>(ns)));
  text += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO";
  if (ns)
  {
    text += "\tFORMAT";
    for (uint32_t s = 0; s < ns; ++s)
    {
      text += '\t';
      text += rq->sample_names[s];
    }
  }
  text += '\n';
  if (g.is_sv_graph) // the calls of an SV graph go through the SV post-processing (sv_graph_records above)
  {
    int const rc = sv_graph_records(c, rq, text);
    if (rc != GTX_OK)
      return rc;
    *len = text.size();
    if (out && cap)
      std::memcpy(out, text.data(), static_cast<size_t>(std::min<uint64_t>(cap, text.size())));
    return GTX_OK;
  }
  // The records of the sites are independent of each other: the sites are cut into contiguous ranges for a team of host
  // threads (every record walks the sample-major arrays of all samples -- cache misses, not arithmetic), every range writes
  // its own text, the texts are joined in order.
  unsigned T = 1;
  if (static_cast<uint64_t>(nh) * (ns + 1) >= 200000) // (small jobs: a team costs more to start than it saves)
  {
    T = std::min(std::max(std::thread::hardware_concurrency(), 1u), 32u);
    if (char const * e = std::getenv("GTX_HOST_THREADS"))
      T = static_cast<unsigned>(std::max(1, std::atoi(e)));
    T = std::min<unsigned>(T, std::max<uint32_t>(nh / 16, 1));
  }
  if (T <= 1)
  {
    std::string error;
    int const rc = write_sites(c, rq, 0, nh, text, error, nullptr);
    if (rc != GTX_OK)
    {
      gtx::g_last_error = error;
      return rc;
    }
  }
  else
  {
    std::vector<std::string> part(T), error(T);
    std::vector<int> status(T, GTX_OK);
    std::vector<std::thread> team;
    for (unsigned t = 0; t < T; ++t)
      team.emplace_back([&, t] {
        uint32_t const b = static_cast<uint32_t>(static_cast<uint64_t>(nh) * t / T), e = static_cast<uint32_t>(static_cast<uint64_t>(nh) * (t + 1) / T);
        part[t].reserve(static_cast<size_t>(e - b) * (400 + 24 * static_cast<size_t>(ns)));
        try
        {
          status[t] = write_sites(c, rq, b, e, part[t], error[t], nullptr);
        }
        catch (...)
        {
          status[t] = GTX_ERR_ARG;
          error[t] = "gtx_vcf_records: out of memory";
        }
      });
    for (auto & th : team)
      th.join();
    for (unsigned t = 0; t < T; ++t)
      if (status[t] != GTX_OK)
      {
        gtx::g_last_error = error[t];
        return status[t];
      }
    for (unsigned t = 0; t < T; ++t)
      text += part[t];
  }
  *len = text.size();
  if (out && cap)
    std::memcpy(out, text.data(), static_cast<size_t>(std::min<uint64_t>(cap, text.size())));
  return GTX_OK;
}

namespace
{
// One variant site as the record writer sees it: a site of the graph over the caller's arrays (fill_site), or what the break-down
// made of one (gtx_vcf_records_final: alleles, calls and read statistics of its own).
struct Site
{
  uint32_t pos = 0, cnum = 0, n_tri = 0;
  std::vector<std::pair<const char *, uint32_t>> seqs;
  std::vector<AlleleStats> al; // the alleles' READ statistics (what the calls add is made by emit_site)
  uint64_t hap_mapq_squared = 0;
  uint32_t clipped_reads = 0;
  std::vector<CallView> calls;
  // a derived site's own storage (seqs / calls point into it)
  std::vector<std::string> alleles;
  std::vector<uint8_t> own_phred;
  std::vector<uint32_t> own_cov;
  std::vector<gtx_sample_call> own_calls;
};

void fill_site(gtx_ctx const * c, gtx_vcf_request const * rq, uint32_t h, Site & st)
{
  gtx::HostGraph const & g = c->graph;
  uint32_t const nh = g.n_hap, ns = rq->n_samples;
  uint32_t const cnum = g.ref_nvar[h], v0 = g.ref_first_var[h], n_tri = cnum * (cnum + 1) / 2;
  uint64_t const aoff = g.allele_off[h], toff = g.tri_off[h];
  st.pos = g.var_order[v0]; // Variant::Variant(Genotype): the site's position on its contig
  st.cnum = cnum;
  st.n_tri = n_tri;
  st.seqs.clear();
  for (uint32_t a = 0; a < cnum; ++a)
    st.seqs.emplace_back(g.dna.data() + g.var_dna[v0 + a], g.var_len[v0 + a]);
  // ---- VarStats of the haplotype
  st.al.assign(cnum, AlleleStats());
  st.hap_mapq_squared = rq->stat_u64[h];
  st.clipped_reads = rq->stat_u32[h];
  for (uint32_t a = 0; a < cnum; ++a)
  {
    const uint64_t * s64 = rq->stat_u64 + nh + 2 * (aoff + a);
    const uint32_t * s32 = rq->stat_u32 + nh + 6 * (aoff + a);
    st.al[a].clipped_bp = s64[0];
    st.al[a].mapq_squared = s64[1];
    st.al[a].score_diff = s32[0];
    st.al[a].mismatches = s32[1];
    st.al[a].r1f = s32[2];
    st.al[a].r1r = s32[3];
    st.al[a].r2f = s32[4];
    st.al[a].r2r = s32[5];
  }
  st.calls.resize(ns);
  for (uint32_t s = 0; s < ns; ++s)
  {
    CallView & cv = st.calls[s];
    cv.phred = rq->phred + static_cast<uint64_t>(s) * g.total_tri + toff;
    cv.cov = rq->gt_cov + static_cast<uint64_t>(s) * g.total_allele + aoff;
    cv.c = rq->calls + static_cast<uint64_t>(s) * nh + h;
    cv.cnum = cnum;
    cv.n_tri = n_tri;
  }
}

// Variant::scan_calls + generate_infos + Vcf::write_record for one site.  good (may be NULL): receives generate_infos' verdict on
// the alternative alleles (is_good_alt, variant.cpp:1040-1070); text_too = false: nothing else is done.  region_filter: the
// region test of Vcf::write_records.  id_suffix: what write_records puts behind the ID of a record that shares position and type
// with the one in front of it.
int emit_site(gtx_vcf_request const * rq, Site const & st, bool region_filter, std::string const & id_suffix, std::string & text, std::string & error,
              std::vector<int8_t> * good, bool text_too)
{
  uint32_t const ns = rq->n_samples, cnum = st.cnum, n_tri = st.n_tri, pos = st.pos;
  std::string const contig = rq->contig;
  auto const & seqs = st.seqs;
  auto const & calls = st.calls;
  std::vector<AlleleStats> al = st.al;
  uint64_t const hap_mapq_squared = st.hap_mapq_squared;
  uint32_t const clipped_reads = st.clipped_reads;
  size_t total_len = 0;
  for (auto const & sq : seqs)
    total_len += sq.second;
  std::vector<double> qd_alt, aa_score;
  {
    uint32_t n_genotyped = 0, n_passed = 0;
    uint64_t seqdepth = 0, qual = 0;
    uint32_t het_first = 0, het_second = 0, hom_first = 0, hom_second = 0;
    long qd_total_qual = 0, qd_total_depth = 0; // Variant::get_qual_by_depth (variant.cpp:1535-1559)
    for (uint32_t s = 0; s < ns; ++s)
    {
      CallView const & cv = calls[s];
      uint32_t const g1 = cv.c->gt_first, g2 = cv.c->gt_second, amb = cv.c->ambiguous_depth;
      if (g1 >= cnum || g2 >= cnum)
      {
        error = "gtx_vcf_records: a call names an allele the site does not have";
        return GTX_ERR_ARG;
      }
      qual += cv.phred[0];
      uint32_t const unique = cv.unique_depth();
      if (cv.phred[0] > 0) // not a homozygous reference call: QD of the variant and of the called alleles
      {
        long const alt_depth = std::min<long>(10, static_cast<long>(unique - cv.coverage(0) + amb));
        if (alt_depth > 0)
        {
          qd_total_qual += std::min<long>(25 * alt_depth, cv.phred[0]);
          qd_total_depth += alt_depth;
        }
        for (int k = 0; k < 2; ++k)
        {
          uint32_t const a = k ? g2 : g1;
          if ((k == 0 && a == 0) || (k == 1 && g1 == g2))
            continue;
          long const depth = std::min<long>(10, static_cast<long>(cv.coverage(a) + amb));
          if (depth > 0)
          {
            al[a].qd_qual += static_cast<uint64_t>(std::min<long>(25 * depth, cv.lowest_phred_without(a)));
            al[a].qd_depth += static_cast<uint64_t>(depth);
          }
        }
      }
      for (uint32_t a = 1; a < cnum; ++a)
      {
        AlleleStats & p = al[a];
        p.max_alt_support = std::max<uint16_t>(p.max_alt_support, static_cast<uint16_t>(cv.coverage(a)));
        if (unique > 0)
          p.max_alt_support_ratio = std::max(p.max_alt_support_ratio, static_cast<double>(cv.coverage(a)) / static_cast<double>(unique));
        if (g1 == a || g2 == a)
          ++(g1 == g2 ? p.n_alt_alt : p.n_ref_alt);
        else
          ++p.n_ref_ref;
        p.total_depth += cv.coverage(a);
      }
      bool const pass = cv.c->gq >= 30; // SampleCall::check_filter == 0
      n_genotyped += cv.any_pl();
      n_passed += pass;
      if (g1 != g2)
      {
        het_first += cv.coverage(g1);
        het_second += cv.coverage(g2);
        al[g1].het_first += cv.coverage(g1);
        al[g1].het_second += unique - cv.coverage(g1);
        al[g2].het_first += cv.coverage(g2);
        al[g2].het_second += unique - cv.coverage(g2);
      }
      else
      {
        hom_first += cv.coverage(g1);
        hom_second += unique - cv.coverage(g1);
        al[g1].hom_first += cv.coverage(g1);
        al[g1].hom_second += unique - cv.coverage(g1);
      }
      seqdepth += unique + amb;
      ++al[g1].ac;
      ++al[g2].ac;
      if (pass)
      {
        ++al[g1].pass_ac;
        ++al[g2].pass_ac;
      }
    }
    if (good)
    {
      // generate_infos' last step (variant.cpp:1040-1070): an alternative allele nobody's reads reached is dropped, the others are
      // held to QD per allele and to their best support in any one sample -- stricter on sites of 71 / 131 alleles and more
      std::vector<int8_t> & out = *good;
      out.assign(cnum - 1, 0);
      for (uint32_t a = 1; a < cnum; ++a)
      {
        AlleleStats const & p = al[a];
        if (p.total_depth == 0)
          continue;
        double const q = p.qd_depth > 0 ? static_cast<double>(p.qd_qual) / static_cast<double>(p.qd_depth) : 0.0;
        out[a - 1] = static_cast<int8_t>(q >= 1.0 && p.max_alt_support >= 2 && (cnum < 71 || (q >= 1.5 && p.max_alt_support_ratio >= 0.2)) &&
                                         (cnum < 131 || (q >= 2.0 && p.max_alt_support_ratio >= 0.225)));
      }
      if (!text_too)
        return GTX_OK;
    }
    // ---- what is skipped (vcf.cpp:775-830, 1226-1258)
    if (region_filter && (pos < rq->region_begin || pos > rq->region_end))
      return GTX_OK;
    if ((ns > 0 && cnum > 80) || total_len > 16000)
      return GTX_OK;
    if (rq->filter_zero_qual && qual == 0)
      return GTX_OK;
    // ---- INFO (Variant::generate_infos)
    Info info;
    auto list_u = [&](char const * key, auto && get, uint32_t first)
    {
      std::string & s = info[key];
      for (uint32_t a = first; a < cnum; ++a)
      {
        if (a > first)
          s += ',';
        put_u(s, get(al[a]));
      }
    };
    auto ratio_or_minus1 = [&](std::string & s, uint64_t num, uint64_t den, double * keep)
    {
      if (den > 0)
      {
        double const v = static_cast<double>(num) / static_cast<double>(den);
        if (keep)
          *keep = v;
        put_g(s, v, 4);
      }
      else
        s += "-1";
    };
    put_u(info["CR"], clipped_reads);
    put_u(info["MQsquared"], hap_mapq_squared);
    list_u("SBF", [](AlleleStats const & p) { return static_cast<uint64_t>(p.r1f + p.r2f); }, 0);
    list_u("SBR", [](AlleleStats const & p) { return static_cast<uint64_t>(p.r1r + p.r2r); }, 0);
    list_u("SBF1", [](AlleleStats const & p) { return static_cast<uint64_t>(p.r1f); }, 0);
    list_u("SBF2", [](AlleleStats const & p) { return static_cast<uint64_t>(p.r2f); }, 0);
    list_u("SBR1", [](AlleleStats const & p) { return static_cast<uint64_t>(p.r1r); }, 0);
    list_u("SBR2", [](AlleleStats const & p) { return static_cast<uint64_t>(p.r2r); }, 0);
    list_u("CRal", [](AlleleStats const & p) { return p.clipped_bp; }, 0);
    list_u("MQSal", [](AlleleStats const & p) { return p.mapq_squared; }, 0);
    list_u("SDal", [](AlleleStats const & p) { return static_cast<uint64_t>(p.score_diff); }, 0);
    list_u("MMal", [](AlleleStats const & p) { return static_cast<uint64_t>(p.mismatches); }, 0);
    put_u(info["RefLen"], seqs[0].second);
    list_u("MaxAAS", [](AlleleStats const & p) { return static_cast<uint64_t>(p.max_alt_support); }, 1);
    {
      std::string & s = info["MaxAASR"];
      for (uint32_t a = 1; a < cnum; ++a)
      {
        if (a > 1)
          s += ',';
        put_g(s, al[a].max_alt_support_ratio, 4);
      }
    }
    list_u("NHomRef", [](AlleleStats const & p) { return static_cast<uint64_t>(p.n_ref_ref); }, 1);
    list_u("NHet", [](AlleleStats const & p) { return static_cast<uint64_t>(p.n_ref_alt); }, 1);
    list_u("NHomAlt", [](AlleleStats const & p) { return static_cast<uint64_t>(p.n_alt_alt); }, 1);
    {
      std::string & s = info["PexcessHet"];
      for (uint32_t a = 1; a < cnum; ++a)
      {
        if (a > 1)
          s += ',';
        put_g(s, excess_het_p(static_cast<int>(al[a].n_ref_alt), static_cast<int>(al[a].n_ref_ref), static_cast<int>(al[a].n_alt_alt)), 6);
      }
    }
    list_u("AC", [](AlleleStats const & p) { return static_cast<uint64_t>(p.ac); }, 1);
    put_u(info["AN"], 2ull * n_genotyped);
    {
      std::string & s = info["AF"];
      for (uint32_t a = 1; a < cnum; ++a)
      {
        if (a > 1)
          s += ',';
        if (n_genotyped > 0)
          put_g(s, static_cast<double>(al[a].ac) / static_cast<double>(2 * n_genotyped), 4);
        else
          s += "0.0";
      }
    }
    list_u("PASS_AC", [](AlleleStats const & p) { return static_cast<uint64_t>(p.pass_ac); }, 1);
    put_u(info["PASS_AN"], 2ull * n_passed);
    double pass_ratio = 0.0;
    if (n_genotyped > 0)
    {
      pass_ratio = static_cast<double>(n_passed) / static_cast<double>(n_genotyped);
      put_g(info["PASS_ratio"], pass_ratio, 4);
    }
    put_u(info["SeqDepth"], seqdepth);
    double ab_het = 0.5, ab_hom = 0.985, sb_alt = 0.0;
    ratio_or_minus1(info["ABHet"], het_second, static_cast<uint32_t>(het_first + het_second), &ab_het);
    ratio_or_minus1(info["ABHom"], hom_first, static_cast<uint32_t>(hom_first + hom_second), &ab_hom);
    {
      uint64_t f = 0, r = 0, fa = 0, ra = 0;
      for (uint32_t a = 0; a < cnum; ++a)
      {
        f += al[a].r1f + al[a].r2f;
        r += al[a].r1r + al[a].r2r;
        if (a)
        {
          fa += al[a].r1f + al[a].r2f;
          ra += al[a].r1r + al[a].r2r;
        }
      }
      // the reference sums the per-allele lists back into 32-bit numbers
      ratio_or_minus1(info["SB"], static_cast<uint32_t>(f), static_cast<uint64_t>(static_cast<uint32_t>(f + r)), nullptr);
      ratio_or_minus1(info["SBAlt"], static_cast<uint32_t>(fa), static_cast<uint64_t>(static_cast<uint32_t>(fa + ra)), &sb_alt);
    }
    {
      std::string & het = info["ABHetMulti"];
      for (uint32_t a = 0; a < cnum; ++a)
      {
        if (a)
          het += ',';
        ratio_or_minus1(het, al[a].het_second, static_cast<uint64_t>(static_cast<uint32_t>(al[a].het_first + al[a].het_second)), nullptr);
      }
      std::string & hom = info["ABHomMulti"];
      for (uint32_t a = 0; a < cnum; ++a)
      {
        if (a)
          hom += ',';
        ratio_or_minus1(hom, al[a].hom_first, static_cast<uint64_t>(static_cast<uint32_t>(al[a].hom_first + al[a].hom_second)), nullptr);
      }
    }
    char const * const type = variant_type(seqs);
    info["VarType"] = type;
    double const qd = qd_total_depth == 0 ? 0.0 : static_cast<double>(qd_total_qual) / static_cast<double>(qd_total_depth);
    put_g(info["QD"], qd, 4);
    qd_alt.assign(cnum - 1, 0.0);
    {
      std::string & s = info["QDalt"];
      for (uint32_t a = 1; a < cnum; ++a)
      {
        if (al[a].qd_depth > 0)
          qd_alt[a - 1] = static_cast<double>(al[a].qd_qual) / static_cast<double>(al[a].qd_depth);
        if (a > 1)
          s += ',';
        put_g(s, qd_alt[a - 1], 4);
      }
    }
    long mq = 60;
    if (seqdepth > 0)
    {
      mq = std::lround(std::sqrt(static_cast<double>(hap_mapq_squared) / static_cast<double>(seqdepth)));
      info["MQ"] = std::to_string(mq);
    }
    else
      info["MQ"] = "0";
    {
      std::string &sd = info["SDalt"], &mm = info["MMalt"], &cr = info["CRalt"], &mqa = info["MQalt"];
      for (uint32_t a = 1; a < cnum; ++a)
      {
        if (a > 1)
        {
          sd += ',';
          mm += ',';
          cr += ',';
          mqa += ',';
        }
        if (al[a].total_depth > 0)
        {
          double const d = static_cast<double>(al[a].total_depth);
          put_g(sd, static_cast<double>(al[a].score_diff) / d, 6);
          put_g(mm, static_cast<double>(al[a].mismatches) / d / 10.0, 6);
          put_g(cr, static_cast<double>(al[a].clipped_bp) / d / 10.0, 6);
          mqa += std::to_string(std::lround(std::sqrt(static_cast<double>(al[a].mapq_squared) / d)));
        }
        else
        {
          sd += "0.0";
          mm += "0.0";
          cr += "0.0";
          mqa += "0";
        }
      }
    }
    aa_score.assign(cnum - 1, 0.0);
    {
      std::string & s = info["AAScore"];
      for (uint32_t a = 1; a < cnum; ++a)
      {
        AlleleStats const & p = al[a];
        double const q = qd_alt[a - 1];
        if (p.total_depth > 0 && q > 0.1 && p.max_alt_support >= 2 && p.max_alt_support_ratio >= 0.15)
        {
          double const depth = static_cast<double>(p.total_depth);
          double const sb = std::fabs(2.0 * ((static_cast<double>(static_cast<uint64_t>(p.r1r) + p.r2r) / depth) - 0.5));
          double const mm = static_cast<double>(p.mismatches) / depth / 10.0;
          long const sd = std::lround(static_cast<double>(p.score_diff) / depth);
          double const cr = static_cast<double>(p.clipped_bp) / depth / 10.0;
          long const mqa = std::lround(std::sqrt(static_cast<double>(p.mapq_squared) / depth));
          double score = model_aa_score(ab_hom, sb, mm, sd, q, cr, mqa);
          if (mm > 1.5)
            score *= std::max(0.5, 1.0 - ((mm - 1.5) / 20.0));
          if (cr + mm > 2.5)
            score *= std::max(0.5, 1.0 - ((cr + mm - 2.5) / 40.0));
          aa_score[a - 1] = score;
        }
        if (a > 1)
          s += ',';
        put_g(s, aa_score[a - 1], 4);
      }
    }
    {
      long const abhet_bin = static_cast<long>(ab_het * 10.0 + 0.00001), sbalt_bin = static_cast<long>(sb_alt * 10.0 + 0.00001);
      double const cr_by_seqdepth = static_cast<double>(static_cast<long>(clipped_reads)) / static_cast<double>(seqdepth);
      double const gt_yield = static_cast<double>(n_genotyped) / static_cast<double>(ns);
      put_g(info["LOGF"], model_logf(ab_hom, cr_by_seqdepth, static_cast<double>(mq), pass_ratio, gt_yield, qd, abhet_bin, sbalt_bin), 4);
    }
    std::vector<uint16_t> const & info_order = info.sorted();
    // ---- the record (Vcf::write_record)
    text += contig;
    text += '\t';
    put_u(text, pos);
    text += '\t';
    text += contig;
    text += ':';
    put_u(text, pos);
    text += ':';
    text += type;
    if (rq->variant_suffix_id && rq->variant_suffix_id[0])
    {
      text += '[';
      text += rq->variant_suffix_id;
      text += ']';
    }
    text += id_suffix;
    for (uint32_t a = 0; a < cnum; ++a)
    {
      text += a < 2 ? '\t' : ',';
      text.append(seqs[a].first, seqs[a].second);
    }
    text += '\t';
    put_u(text, qual);
    text += '\t';
    if (ns == 0)
      text += '.';
    else
    {
      // the thresholds are applied to the numbers as printed (the reference parses its own INFO text)
      size_t const before = text.size();
      auto fail = [&](char const * name)
      {
        if (text.size() != before)
          text += ';';
        text += name;
      };
      auto as_double = [&](char const * k) { return std::stod(*info.find(k)); };
      long const an = 2l * n_genotyped;
      if (*info.find("ABHet") != "-1" && as_double("ABHet") < 0.175)
        fail("LowABHet");
      if (*info.find("ABHom") != "-1" && as_double("ABHom") < 0.85)
        fail("LowABHom");
      if (an >= 6 && as_double("QD") < 6.0)
        fail("LowQD");
      if (an >= 6)
      {
        bool good = false;
        std::string const & s = *info.find("AAScore");
        for (size_t i = 0; i < s.size();)
        {
          size_t const e = std::min(s.find(',', i), s.size());
          good = good || std::stod(s.substr(i, e - i)) > 0.15;
          i = e + 1;
        }
        if (!good)
          fail("LowAAScore");
      }
      if (qual < 10)
        fail("LowQUAL");
      if (an >= 500 && info.find("PASS_ratio") && as_double("PASS_ratio") < 0.05)
        fail("LowPratio");
      if (text.size() == before)
        text += "PASS";
    }
    text += '\t';
    for (size_t i = 0; i < info_order.size(); ++i)
    {
      auto const & e = info.kv[info_order[i]];
      if (i)
        text += ';';
      text += e.first;
      if (!e.second.empty())
      {
        text += '=';
        text += e.second;
      }
    }
    if (ns)
    {
      text += "\tGT:AD:MD:DP:GQ:PL";
      // (the columns are most of a record's bytes: written into room made for the longest they can be -- a tab, two alleles and a
      //  slash, a depth per allele, three numbers, a value per genotype, each behind a separator -- and cut to what was written)
      size_t const room = 64 + 11 * static_cast<size_t>(cnum) + 4 * static_cast<size_t>(n_tri);
      size_t const at = text.size();
      text.resize(at + room * ns + 1);
      char * p = &text[at];
      for (uint32_t s = 0; s < ns; ++s)
      {
        CallView const & cv = calls[s];
        *p++ = '\t';
        if (!cv.any_pl())
        {
          *p++ = '.';
          *p++ = '/';
          *p++ = '.';
        }
        else
        {
          p = put_u_at(p, cv.c->gt_first);
          *p++ = '/';
          p = put_u_at(p, cv.c->gt_second);
        }
        uint32_t unique = 0;
        for (uint32_t a = 0; a < cnum; ++a)
        {
          *p++ = a ? ',' : ':';
          uint32_t const d = cv.coverage(a);
          unique += d;
          p = put_u_at(p, d);
        }
        *p++ = ':';
        p = put_u_at(p, cv.c->ambiguous_depth);
        *p++ = ':';
        p = put_u_at(p, unique + cv.c->ambiguous_depth);
        *p++ = ':';
        p = put_u_at(p, std::min<uint16_t>(99, BINNED.v[cv.c->gq]));
        for (uint32_t i = 0; i < n_tri; ++i)
        {
          *p++ = i ? ',' : ':';
          p = put_u_at(p, BINNED.v[cv.phred[i]]);
        }
      }
      text.resize(static_cast<size_t>(p - text.data()));
    }
    text += '\n';
  }
  return GTX_OK;
}

int write_sites(gtx_ctx const * c, gtx_vcf_request const * rq, uint32_t h_begin, uint32_t h_end, std::string & text, std::string & error,
                std::vector<std::vector<int8_t>> * good)
{
  Site st;
  for (uint32_t h = h_begin; h < h_end; ++h)
  {
    fill_site(c, rq, h, st);
    int const rc = emit_site(rq, st, true, std::string(), text, error, good ? &(*good)[h] : nullptr, good == nullptr);
    if (rc != GTX_OK)
      return rc;
  }
  return GTX_OK;
}
} // namespace

// ---- the final VCF of a small-variant graph: vcf_merge_and_break with the break-down (src/typer/vcf_operations.cpp:480-732 with
// force_no_break_down = false -- the file genotype() ends with, src/utilities/genotype.cpp:577-604).
namespace
{
// SampleCall::get_gt_call / get_gq (sample_call.cpp:70-131) of a call made here (what gtx_calls_batch does for the graph's own sites)
void recall(gtx_sample_call & c, uint8_t const * phred, uint32_t cnum)
{
  c.gt_first = c.gt_second = 0;
  bool found = false, seen_zero = false, two_zeros = false;
  uint8_t next_lowest = 255;
  uint32_t i = 0;
  for (uint32_t y = 0; y < cnum; ++y)
    for (uint32_t x = 0; x <= y; ++x, ++i)
    {
      if (phred[i] == 0)
      {
        if (!found)
        {
          c.gt_first = static_cast<uint16_t>(x);
          c.gt_second = static_cast<uint16_t>(y);
          found = true;
        }
        if (seen_zero)
          two_zeros = true;
        seen_zero = true;
      }
      else if (phred[i] < next_lowest)
        next_lowest = phred[i];
    }
  c.gq = two_zeros ? 0 : next_lowest;
}

struct Broken // a site the break-down made, with its place for left_align
{
  Site site;
  std::vector<std::string> & alleles;
  uint32_t & pos;
  explicit Broken(Site && s) : site(std::move(s)), alleles(site.alleles), pos(site.pos) {}
  Broken(Broken && o) noexcept : site(std::move(o.site)), alleles(site.alleles), pos(site.pos) {}
  Broken & operator=(Broken && o) noexcept
  {
    site = std::move(o.site);
    return *this;
  }
  void rebind() // (after the alleles changed or the site moved: the writer's views of the site's own alleles and calls)
  {
    site.seqs.clear();
    for (std::string const & a : site.alleles)
      site.seqs.emplace_back(a.data(), static_cast<uint32_t>(a.size()));
    if (!site.own_calls.empty())
    {
      uint32_t const ns = static_cast<uint32_t>(site.own_calls.size());
      site.calls.resize(ns);
      for (uint32_t s = 0; s < ns; ++s)
      {
        CallView & cv = site.calls[s];
        cv.phred = site.own_phred.data() + static_cast<size_t>(s) * site.n_tri;
        cv.cov = site.own_cov.data() + static_cast<size_t>(s) * site.cnum;
        cv.c = &site.own_calls[s];
        cv.cnum = site.cnum;
        cv.n_tri = site.n_tri;
      }
    }
  }
};

// break_multi_snps (variant.cpp:1996-2111): the alleles of `st` -- all of one length, as strings in `alleles` -- position by position
void break_multi_snps(gtx_vcf_request const * rq, Site const & st, std::vector<std::string> const & alleles, uint32_t pos, std::vector<Broken> & out)
{
  uint32_t const ns = rq->n_samples, cnum = st.cnum;
  std::vector<int> ac(cnum, 0);
  for (uint32_t s = 0; s < ns; ++s)
  {
    ac[st.calls[s].c->gt_first]++;
    ac[st.calls[s].c->gt_second]++;
  }
  for (size_t j = 0; j < alleles[0].size(); ++j)
  {
    std::string new_seqs(1, alleles[0][j]);
    std::vector<uint16_t> old_to_new(1, 0);
    for (uint32_t k = 1; k < cnum; ++k)
    {
      if (ac[k] == 0)
      {
        old_to_new.push_back(0);
        continue;
      }
      size_t const at = new_seqs.find(alleles[k][j]);
      if (at == std::string::npos)
      {
        old_to_new.push_back(static_cast<uint16_t>(new_seqs.size()));
        new_seqs.push_back(alleles[k][j]);
      }
      else
        old_to_new.push_back(static_cast<uint16_t>(at));
    }
    if (new_seqs.size() == 1)
      continue; // no SNP at this position
    Site nv;
    uint32_t const n_new = static_cast<uint32_t>(new_seqs.size()), new_tri = n_new * (n_new + 1) / 2;
    nv.pos = pos + static_cast<uint32_t>(j);
    nv.cnum = n_new;
    nv.n_tri = new_tri;
    for (char ch : new_seqs)
      nv.alleles.emplace_back(1, ch);
    // update_per_allele_stats (variant.cpp:34-82): the read statistics of the old alleles under the new ones
    nv.al.assign(n_new, AlleleStats());
    nv.hap_mapq_squared = st.hap_mapq_squared;
    nv.clipped_reads = st.clipped_reads;
    for (uint32_t y = 0; y < cnum; ++y)
    {
      AlleleStats const & o = st.al[y];
      AlleleStats & n = nv.al[old_to_new[y]];
      n.clipped_bp += o.clipped_bp;
      n.mapq_squared += o.mapq_squared;
      n.score_diff += o.score_diff;
      n.mismatches += o.mismatches;
      n.r1f += o.r1f;
      n.r2f += o.r2f;
      n.r1r += o.r1r;
      n.r2r += o.r2r;
    }
    nv.own_phred.assign(static_cast<size_t>(ns) * new_tri, 255u);
    nv.own_cov.assign(static_cast<size_t>(ns) * n_new, 0u);
    nv.own_calls.resize(ns);
    for (uint32_t s = 0; s < ns; ++s)
    {
      CallView const & cv = st.calls[s];
      uint8_t * ph = nv.own_phred.data() + static_cast<size_t>(s) * new_tri;
      uint32_t * cov = nv.own_cov.data() + static_cast<size_t>(s) * n_new;
      uint32_t i = 0;
      for (uint32_t y = 0; y < cnum; ++y)
      {
        for (uint32_t x = 0; x <= y; ++x, ++i)
        {
          uint32_t ny = old_to_new[y], nx = old_to_new[x];
          if (nx > ny)
            std::swap(nx, ny);
          uint32_t const ni = nx + (ny + 1) * ny / 2;
          ph[ni] = std::min(ph[ni], cv.phred[i]);
        }
        uint32_t const ny = old_to_new[y];
        cov[ny] = cov[ny] + cv.coverage(y) < 0xFFFFu ? cov[ny] + cv.coverage(y) : 0xFFFFu;
      }
      gtx_sample_call & nc = nv.own_calls[s];
      nc = *cv.c; // (ambiguous depth, the total depths, the proper-pair depth: copied)
      recall(nc, ph, n_new);
    }
    out.emplace_back(std::move(nv));
  }
}
} // namespace

extern "C" int gtx_vcf_records_final(const gtx_ctx * c, const gtx_vcf_request * rq, int no_variant_overlapping, int no_filter_bad_alts, char * out,
                                     uint64_t cap, uint64_t * len)
{
  if (!c || !rq || !len || (cap && !out) || !rq->contig || !rq->gt_cov || !rq->stat_u64 || !rq->stat_u32 || !rq->phred || !rq->calls ||
      (rq->n_samples && !rq->sample_names))
    return GTX_ERR_ARG;
  gtx::HostGraph const & g = c->graph;
  if (g.is_sv_graph)
  {
    gtx::g_last_error = "gtx_vcf_records_final: an SV graph's calls go through gtx_vcf_records (the SV post-processing breaks nothing down)";
    return GTX_ERR_UNSUPPORTED;
  }
  uint32_t const nh = g.n_hap, ns = rq->n_samples;
  std::string text = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO";
  if (ns)
  {
    text += "\tFORMAT";
    for (uint32_t s = 0; s < ns; ++s)
    {
      text += '\t';
      text += rq->sample_names[s];
    }
  }
  text += '\n';
  // the region's reference (add_base_in_front, Variant::normalize read it through the graph: graph.cpp get_generated_reference_genome)
  uint32_t const first_pos = g.ref_order.empty() ? 1u : g.ref_order.front();
  std::string ref;
  for (size_t n = 0; n < g.ref_order.size(); ++n)
  {
    ref.append(g.dna.data() + g.ref_dna[n], g.ref_len[n]);
    if (g.ref_nvar[n] > 0)
      ref.append(g.dna.data() + g.var_dna[g.ref_first_var[n]], g.var_len[g.ref_first_var[n]]);
  }
  std::string error;
  std::vector<Broken> broken; // broken_vars of the reference's loop
  auto shape = [](Site const & s) { return static_cast<int>(s.seqs[0].second > s.seqs[1].second) + 2 * static_cast<int>(s.seqs[0].second == s.seqs[1].second); };
  auto seqs_less = [](Site const & a, Site const & b) { return a.alleles < b.alleles; };
  // Vcf::write_records (vcf.cpp:1161-1275) over what `broken` holds, positions [lo, hi]
  auto write_records = [&](uint32_t lo, uint32_t hi) -> int
  {
    if (broken.empty())
      return GTX_OK;
    std::vector<size_t> order(broken.size());
    std::iota(order.begin(), order.end(), size_t(0));
    std::sort(order.begin(), order.end(), [&](size_t i, size_t j)
    {
      Site const & a = broken[i].site;
      Site const & b = broken[j].site;
      if (a.pos != b.pos)
        return a.pos < b.pos;
      if (shape(a) != shape(b))
        return shape(a) < shape(b);
      return seqs_less(a, b); // (equal alleles: the reference prefers the record with more INFO keys -- every record here has the same)
    });
    long dup = -1;
    for (size_t k = 0; k < order.size(); ++k)
    {
      Site const & cur = broken[order[k]].site;
      std::string suffix;
      if (k > 0)
      {
        Site const & prev = broken[order[k - 1]].site;
        if (cur.pos > hi)
          break;
        if (cur.pos < lo)
          continue;
        if (cur.pos == prev.pos && cur.alleles == prev.alleles)
          continue;
        if (cur.pos == prev.pos && std::strcmp(variant_type(cur.seqs), variant_type(prev.seqs)) == 0)
          suffix = "." + std::to_string(++dup);
        else
          dup = -1;
      }
      else if (cur.pos < lo || cur.pos > hi)
        continue;
      int const rc = emit_site(rq, cur, false, suffix, text, error, nullptr, true);
      if (rc != GTX_OK)
        return rc;
    }
    return GTX_OK;
  };
  Site st;
  std::vector<int8_t> good;
  std::string scratch;
  for (uint32_t h = 0; h < nh; ++h)
  {
    fill_site(c, rq, h, st);
    for (uint32_t s = 0; s < ns; ++s)
      if (st.calls[s].c->gt_first >= st.cnum || st.calls[s].c->gt_second >= st.cnum)
      {
        gtx::g_last_error = "gtx_vcf_records_final: a call names an allele the site does not have";
        return GTX_ERR_ARG;
      }
    // ---- break_down_variant (variant.cpp:1652-1713; no --no_decompose, no --is_all_biallelic)
    std::vector<Broken> made;
    std::vector<std::string> alleles;
    for (auto const & sq : st.seqs)
      alleles.emplace_back(sq.first, sq.second);
    bool const all_same_size = std::all_of(alleles.begin() + 1, alleles.end(), [&](std::string const & a) { return a.size() == alleles[0].size(); });
    if (all_same_size)
    {
      uint32_t pos = st.pos;
      bool matching = true; // Variant::is_with_matching_first_bases
      for (size_t a = 1; a < alleles.size(); ++a)
        matching = matching && alleles[a][0] == alleles[0][0];
      if (!matching && pos > first_pos && pos - 1 - first_pos < ref.size()) // add_base_in_front(true): the base in front, N unless A C G T
      {
        char b = ref[pos - 1 - first_pos];
        if (b != 'A' && b != 'C' && b != 'G' && b != 'T')
          b = 'N';
        for (std::string & a : alleles)
          if (a != "*")
            a.insert(a.begin(), b);
        --pos;
      }
      break_multi_snps(rq, st, alleles, pos, made);
    }
    else if (no_variant_overlapping)
    {
      Site whole = st; // (views into the caller's arrays stay views; the alleles become its own: left_align may change them)
      whole.alleles = alleles;
      made.emplace_back(std::move(whole));
    }
    else
    {
      // (break_down_skyr, variant.cpp:2113-2190: an alternative allele nobody is called with is handed to paw::Skyr as the
      //  reference allele, :2137-2155 -- when that is all of them there is nothing to find and the site leaves no record: the one
      //  case that does not hang on the absent library)
      bool carried = false;
      for (uint32_t s = 0; s < ns && !carried; ++s)
        carried = st.calls[s].c->gt_first != 0 || st.calls[s].c->gt_second != 0;
      if (!carried)
        continue;
      gtx::g_last_error = "gtx_vcf_records_final: the site at " + std::to_string(st.pos) + " has alleles of different lengths: the reference breaks it "
                          "down with paw::Skyr (variant.cpp:2113-2190), whose source its tree does not hold -- no_variant_overlapping writes such sites whole";
      return GTX_ERR_UNSUPPORTED;
    }
    // ---- normalize, judge, keep (vcf_operations.cpp:622-663)
    bool any = false;
    for (Broken & b : made)
    {
      b.rebind();
      if (left_align(b, ref, first_pos) > 200)
        continue;
      b.rebind();
      good.clear();
      scratch.clear();
      int const rc = emit_site(rq, b.site, false, std::string(), scratch, error, &good, false);
      if (rc != GTX_OK)
      {
        gtx::g_last_error = error;
        return rc;
      }
      if (!no_filter_bad_alts && std::all_of(good.begin(), good.end(), [](int8_t x) { return x == 0; }))
        continue;
      broken.emplace_back(std::move(b));
      broken.back().rebind();
      any = true;
    }
    if (!any)
      continue;
    // ---- written in windows as the loop goes (:668-716)
    long const W = 700;
    long min_pos = broken[0].site.pos, max_pos = broken[0].site.pos;
    for (Broken const & b : broken)
    {
      min_pos = std::min<long>(min_pos, b.site.pos);
      max_pos = std::max<long>(max_pos, b.site.pos);
    }
    if (min_pos + 2 * W < max_pos)
    {
      long const reg_end = std::min<long>(rq->region_end, max_pos - W);
      if (reg_end >= static_cast<long>(rq->region_begin))
      {
        int const rc = write_records(rq->region_begin, static_cast<uint32_t>(reg_end));
        if (rc != GTX_OK)
        {
          gtx::g_last_error = error;
          return rc;
        }
        std::vector<Broken> rest;
        for (Broken & b : broken)
          if (static_cast<long>(b.site.pos) > reg_end)
            rest.emplace_back(std::move(b));
        broken = std::move(rest);
        for (Broken & b : broken)
          b.rebind();
      }
    }
  }
  {
    int const rc = write_records(rq->region_begin, rq->region_end);
    if (rc != GTX_OK)
    {
      gtx::g_last_error = error;
      return rc;
    }
  }
  *len = text.size();
  if (out && cap)
    std::memcpy(out, text.data(), static_cast<size_t>(std::min<uint64_t>(cap, text.size())));
  return GTX_OK;
}

// ---- the sites a genotyping iteration hands to the next one: vcf_merge_and_filter (src/typer/vcf_operations.cpp:278-478).
// The reference reads the pools' variants back (their calls already scanned into the statistics and cleared,
// hts_parallel_reader.cpp:938-962), adds the pools' statistics, lets generate_infos judge the alternative alleles and writes
// every allele it keeps as a bi-allelic record of its own: no samples, QUAL 0, FILTER ".", and in INFO the allele's number over
// the whole file (GT_ID: alleles are counted from 1 in file order, kept or not), the alleles it cannot share a haplotype with
// (GT_ANTI_HAPLOTYPE: the later kept alleles of its own site, then what `ph` says) and those it was always seen with
// (GT_HAPLOTYPE) -- what the next iteration's graph construction turns into events (constructor.cpp:1540-1588).  Here the
// statistics of all pools are one accumulator block (sums are sums), so the "merge" is the block itself.
extern "C" int gtx_vcf_sites(const gtx_ctx * c, const gtx_vcf_request * rq, const gtx_phase_entry * ph, uint64_t n_ph, char * out, uint64_t cap,
                             uint64_t * len)
{
  if (!c || !rq || !len || (cap && !out) || !rq->contig || !rq->gt_cov || !rq->stat_u64 || !rq->stat_u32 || !rq->phred || !rq->calls || (n_ph && !ph))
    return GTX_ERR_ARG;
  gtx::HostGraph const & g = c->graph;
  if (g.is_sv_graph) // (the reference writes no haplotype sites from an SV graph: hts_parallel_reader.cpp:941)
  {
    gtx::g_last_error = "gtx_vcf_sites: not for SV graphs";
    return GTX_ERR_UNSUPPORTED;
  }
  uint32_t const nh = g.n_hap;
  std::vector<std::vector<int8_t>> good(nh);
  {
    std::string none, error;
    int const rc = write_sites(c, rq, 0, nh, none, error, &good);
    if (rc != GTX_OK)
    {
      gtx::g_last_error = error;
      return rc;
    }
  }
  // hap_id2var_id: alleles in front of a site's first alternative one
  std::vector<uint64_t> first_id(nh + 1, 0);
  for (uint32_t h = 0; h < nh; ++h)
    first_id[h + 1] = first_id[h] + (g.ref_nvar[h] - 1u);
  // ph rows come in the map's order (hap1, allele1, hap2, allele2): the rows of one outer key are one run
  auto const key_less = [](gtx_phase_entry const & e, std::pair<uint16_t, uint16_t> const & k)
  { return e.hap1 < k.first || (e.hap1 == k.first && e.allele1 < k.second); };
  std::string text = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n";
  std::string const contig = rq->contig;
  std::vector<std::pair<const char *, uint32_t>> seqs(2);
  std::string anti, hap;
  for (uint32_t h = 0; h < nh; ++h)
  {
    uint32_t const cnum = g.ref_nvar[h], v0 = g.ref_first_var[h];
    uint32_t const pos = g.var_order[v0];
    std::vector<int8_t> const & ok = good[h];
    for (uint32_t a = 0; a + 1 < cnum; ++a)
    {
      uint64_t const var_id = first_id[h] + a + 1;
      if (!ok[a])
        continue;
      anti.clear();
      hap.clear();
      for (uint32_t a2 = a + 1; a2 + 1 < cnum; ++a2)
        if (ok[a2])
        {
          if (!anti.empty())
            anti += ',';
          put_u(anti, var_id + a2 - a);
        }
      // (the map's keys are 16-bit: std::make_pair<uint16_t, uint16_t>(var.hap_id, a + 1))
      std::pair<uint16_t, uint16_t> const key(static_cast<uint16_t>(h), static_cast<uint16_t>(a + 1));
      for (gtx_phase_entry const * e = std::lower_bound(ph, ph + n_ph, key, key_less); e != ph + n_ph && e->hap1 == key.first && e->allele1 == key.second; ++e)
      {
        if (e->hap2 == 0xFFFFu && e->allele2 == 0xFFFFu) // (an outer key without a flag under it)
          continue;
        if (e->allele2 == 0 || (e->flags != 1 && e->flags != 2))
          continue;
        if (e->hap2 >= nh)
        {
          gtx::g_last_error = "gtx_vcf_sites: a phase row names a haplotype the graph does not have";
          return GTX_ERR_ARG;
        }
        std::string & to = e->flags == 1 ? hap : anti;
        if (!to.empty())
          to += ',';
        put_u(to, first_id[e->hap2] + e->allele2);
      }
      seqs[0] = {g.dna.data() + g.var_dna[v0], g.var_len[v0]};
      seqs[1] = {g.dna.data() + g.var_dna[v0 + a + 1], g.var_len[v0 + a + 1]};
      if (static_cast<size_t>(seqs[0].second) + seqs[1].second > 16000) // (write_record, vcf.cpp:790-805)
        continue;
      text += contig;
      text += '\t';
      put_u(text, pos);
      text += '\t';
      text += contig;
      text += ':';
      put_u(text, pos);
      text += ':';
      text += variant_type(seqs);
      text += '\t';
      text.append(seqs[0].first, seqs[0].second);
      text += '\t';
      text.append(seqs[1].first, seqs[1].second);
      text += "\t0\t.\t";
      if (!anti.empty())
      {
        text += "GT_ANTI_HAPLOTYPE=";
        text += anti;
        text += ';';
      }
      if (!hap.empty())
      {
        text += "GT_HAPLOTYPE=";
        text += hap;
        text += ';';
      }
      text += "GT_ID=";
      put_u(text, var_id);
      text += '\n';
    }
  }
  *len = text.size();
  if (out && cap)
    std::memcpy(out, text.data(), static_cast<size_t>(std::min<uint64_t>(cap, text.size())));
  return GTX_OK;
}

// ---- the description lines of the header: Vcf::write_header (src/typer/vcf.cpp:526-760).  The texts are the file format
// (tests/golden/vcf_header_definitions.txt holds the reference's own output for them; tests/test_vcf_text.py compares).
namespace
{
struct HeaderLine
{
  char const * kind;   // INFO, FORMAT, FILTER
  char const * id;
  char const * number; // "" for FILTER lines
  char const * type;
  char const * description;
};
HeaderLine const HEADER_LINES[] = {
  {"INFO", "AAScore", "A", "Float",
   "Alternative allele confidence score in range [0.0,1.0]. The score is determined by a logistic regression model which was trained on GIAB truth data using other INFOs metrics as covariates."},
  {"INFO", "ABHet", "1", "Float",
   "Allele Balance for heterozygouscalls (read count of call2/(call1+call2)) where the called genotype is call1/call2. -1 if no heterozygous calls."},
  {"INFO", "ABHom", "1", "Float",
   "Allele Balance for homozygous calls(read count of A/(A+O)) where A is the called allele and O is anything else. -1 if no homozygous calls."},
  {"INFO", "ABHetMulti", "R", "Float",
   "List of Allele Balance values for heterozygous calls (alt/(ref+alt)). -1 if not available."},
  {"INFO", "ABHomMulti", "R", "Float",
   "List of Allele Balance values for homozygous calls (A/(A+0)) where A is the called allele and O is anything else. -1 if not available."},
  {"INFO", "AC", "A", "Integer",
   "Number of alternate alleles in called genotypes."},
  {"INFO", "AF", "A", "Float",
   "Allele frequency."},
  {"INFO", "AN", "1", "Integer",
   "Number of alleles in called genotypes."},
  {"INFO", "CR", "1", "Integer",
   "Number of clipped reads in the graph alignment."},
  {"INFO", "CRal", ".", "String",
   "Number of clipped bp per allele."},
  {"INFO", "CRalt", "A", "Float",
   "Percent of clipped reads per allele."},
  {"INFO", "END", "1", "Integer",
   "End position of an SV."},
  {"INFO", "FEATURE", "1", "String",
   "Gene feature."},
  {"INFO", "GT_ANTI_HAPLOTYPE", ".", "String",
   "Haplotype string with downstream variants  with no (or very low) evidence of being in the same haplotype. Used internally by Graphtyper."},
  {"INFO", "GT_HAPLOTYPE", ".", "String",
   "Haplotype string with downstream variants  with high evidence of being always in the same haplotype. Used internally by Graphtyper."},
  {"INFO", "GT_ID", ".", "String",
   "ID for variant. Used internally by Graphtyper."},
  {"INFO", "HOMSEQ", ".", "String",
   "Sequence of base pair identical homology at event breakpoints."},
  {"INFO", "INV3", "0", "Flag",
   "Inversion breakends open 3' of reported location"},
  {"INFO", "INV5", "0", "Flag",
   "Inversion breakends open 5' of reported location"},
  {"INFO", "LEFT_SVINSSEQ", ".", "String",
   "Known left side of insertion for an insertion of unknown length."},
  {"INFO", "LOGF", "1", "Float",
   "Output from logistic regression model."},
  {"INFO", "MaxAAS", "A", "Integer",
   "Maximum alternative allele support per alt. allele."},
  {"INFO", "MaxAASR", "A", "Float",
   "Maximum alternative allele support ratio per alt. allele."},
  {"INFO", "MaxAltPP", "1", "Integer",
   "Maximum number of proper pairs support the alternative allele."},
  {"INFO", "MMal", ".", "String",
   "Scaled mismatch count per allele."},
  {"INFO", "MMalt", "A", "Float",
   "Mismatch percent per alternative allele."},
  {"INFO", "MQ", "1", "Integer",
   "Root-mean-square mapping quality."},
  {"INFO", "MQalt", "A", "Integer",
   "Mapping qualities per alternative allele."},
  {"INFO", "MQSal", ".", "String",
   "Sum of squared mapping qualities per allele."},
  {"INFO", "MQsquared", ".", "String",
   "Sum of squared mapping qualities. Used to calculate MQ."},
  {"INFO", "NCLUSTERS", "1", "Integer",
   "Number of SV candidates in cluster."},
  {"INFO", "NGT", "3", "Integer",
   "Number of REF/REF, REF/ALT and ALT/ALTgenotypes, respectively."},
  {"INFO", "NHet", "A", "Integer",
   "Number of heterozygous genotype calls."},
  {"INFO", "NHomRef", "A", "Integer",
   "Number of homozygous reference genotype calls."},
  {"INFO", "NHomAlt", "A", "Integer",
   "Number of homozygous alternative genotype calls."},
  {"INFO", "NUM_MERGED_SVS", "1", "Integer",
   "Number of SVs merged."},
  {"INFO", "OLD_VARIANT_ID", "1", "String",
   "Variant ID from a VCF (SVs only)."},
  {"INFO", "ORSTART", "1", "Integer",
   "Start coordinate of sequence origin."},
  {"INFO", "OREND", "1", "Integer",
   "End coordinate of sequence origin."},
  {"INFO", "QD", "1", "Float",
   "QUAL divided by NonReferenceSeqDepth."},
  {"INFO", "QDalt", "A", "Float",
   "Simplified QD calculated separately for each allele against all other alleles."},
  {"INFO", "PASS_AC", "A", "Integer",
   "Number of alternate alleles in called genotyped that have FT = PASS."},
  {"INFO", "PASS_AN", "1", "Integer",
   "Number of genotype calls that haveFT = PASS."},
  {"INFO", "PASS_ratio", "1", "Float",
   "Ratio of genotype calls that haveFT = PASS."},
  {"INFO", "PexcessHet", "A", "Float",
   "Pval of excess heterozygous calls."},
  {"INFO", "RefLen", "1", "Integer",
   "Length of the reference allele."},
  {"INFO", "RELATED_SV_ID", "1", "Integer",
   "GraphTyper ID of a related SV."},
  {"INFO", "RIGHT_SVINSSEQ", ".", "String",
   "Known right side of insertion for an insertion of unknown length."},
  {"INFO", "SB", "1", "Float",
   "Strand bias (F/(F+R)) where F and R are forward and reverse strands, respectively. -1 if not available."},
  {"INFO", "SBAlt", "1", "Float",
   "Strand bias of alternative alleles only. -1 if not available."},
  {"INFO", "SBF", "R", "Integer",
   "Number of forward stranded reads per allele."},
  {"INFO", "SBF1", "R", "Integer",
   "Number of first forward stranded reads per allele."},
  {"INFO", "SBF2", "R", "Integer",
   "Number of second forward stranded reads per allele."},
  {"INFO", "SBR", "R", "Integer",
   "Number of reverse stranded reads per allele."},
  {"INFO", "SBR1", "R", "Integer",
   "Number of first reverse stranded reads per allele."},
  {"INFO", "SBR2", "R", "Integer",
   "Number of second reverse stranded reads per allele."},
  {"INFO", "SDal", ".", "String",
   "Score difference of AS and XS tags per allele."},
  {"INFO", "SDalt", "A", "Float",
   "Avergae score difference of AS and XS tags per alternative allele."},
  {"INFO", "SEQ", "1", "String",
   "Inserted sequence at variant site."},
  {"INFO", "SeqDepth", "1", "Integer",
   "Total accumulated sequencing depth over all the samples."},
  {"INFO", "SV_ID", "1", "Integer",
   "GraphTyper's ID on SV."},
  {"INFO", "SVINSSEQ", ".", "String",
   "Sequence of insertion."},
  {"INFO", "SVLEN", "1", "Integer",
   "Length of structural variant in bp. Negative lengths indicate a deletion."},
  {"INFO", "SVMODEL", "1", "String",
   "Model used for SV genotyping."},
  {"INFO", "SVSIZE", "1", "Integer",
   "Size of structural variant in bp. Always 50 or more."},
  {"INFO", "SVTYPE", "1", "String",
   "Type of structural variant."},
  {"INFO", "VarType", "1", "String",
   "First letter is program identifier,the second letter is variant type."},
  {"FORMAT", "GT", "1", "String",
   "GenoType call. ./. is called if there is no coverage at the variant site."},
  {"FORMAT", "FT", "1", "String",
   "Filter. PASS or FAILN where N is a number."},
  {"FORMAT", "AD", "R", "Integer",
   "Allelic depths for the ref and alt alleles in the order listed."},
  {"FORMAT", "MD", "1", "Integer",
   "Read depth of multiple alleles."},
  {"FORMAT", "DP", "1", "Integer",
   "Approximate read depth."},
  {"FORMAT", "RA", "2", "Integer",
   "Total read depth of the reference allele and all alternative alleles, including reads that support more than one allele."},
  {"FORMAT", "PP", "1", "Integer",
   "Number of reads that support non-reference haplotype that are proper pairs."},
  {"FORMAT", "GQ", "1", "Integer",
   "Genotype Quality."},
  {"FORMAT", "PL", "G", "Integer",
   "PHRED-scaled genotype likelihoods."},
  {"FILTER", "PASS", "", "",
   "All filters passed"},
  {"FILTER", "LowAAScore", "", "",
   "Alternative alleles have a low score."},
  {"FILTER", "LowABHet", "", "",
   "Allele balance of heterozygous carriers is below 17.5%."},
  {"FILTER", "LowABHom", "", "",
   "Allele balance of homozygous carriers is below 90%."},
  {"FILTER", "LowQD", "", "",
   "QD (quality by depth) is below 6.0."},
  {"FILTER", "LowQUAL", "", "",
   "QUAL score is less than 10."},
  {"FILTER", "LowPratio", "", "",
   "Ratio of PASSed calls was too low."},
};
} // namespace

extern "C" int gtx_vcf_header(const gtx_vcf_header_request * rq, char * out, uint64_t cap, uint64_t * len)
{
  if (!rq || !len || (cap && !out) || (rq->n_contigs && (!rq->contig_names || !rq->contig_lengths)) || (rq->n_samples && !rq->sample_names))
  {
    gtx::g_last_error = "gtx_vcf_header: bad argument";
    return GTX_ERR_ARG;
  }
  std::string s = "##fileformat=VCFv4.2\n##fileDate=";
  s += rq->file_date ? rq->file_date : "";
  s += "\n##source=Graphtyper\n##graphtyperVersion=";
  s += rq->version ? rq->version : "";
  if (rq->dirty)
    s += "-dirty";
  s += "\n##graphtyperGitBranch=";
  s += rq->git_branch ? rq->git_branch : "";
  s += "\n##graphtyperSHA1=";
  s += rq->git_sha1 ? rq->git_sha1 : "";
  s += "\n";
  for (uint32_t i = 0; i < rq->n_contigs; ++i)
    s += std::string("##contig=<ID=") + rq->contig_names[i] + ",length=" + std::to_string(rq->contig_lengths[i]) + ">\n";
  for (HeaderLine const & h : HEADER_LINES)
  {
    s += std::string("##") + h.kind + "=<ID=" + h.id;
    if (h.number[0])
      s += std::string(",Number=") + h.number + ",Type=" + h.type;
    s += std::string(",Description=\"") + h.description + "\">\n";
  }
  s += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO";
  if (!rq->drop_genotypes && rq->n_samples)
  {
    s += "\tFORMAT";
    for (uint32_t i = 0; i < rq->n_samples; ++i)
      s += std::string("\t") + rq->sample_names[i];
  }
  s += "\n";
  *len = s.size();
  if (out)
    std::memcpy(out, s.data(), std::min<uint64_t>(cap, s.size()));
  return GTX_OK;
}

// ---- BGZF: the members htslib's bgzf_write makes (SAM spec 4.1): gzip members with the BC extra field, at most 0xff00 bytes of
// input each, and the 28-byte empty member at the end of a file (what the reference's bgzf_stream writes its VCF through,
// include/graphtyper/utilities/bgzf_stream.hpp).
extern "C" int gtx_bgzf_compress(const void * in, uint64_t in_len, int level, int with_eof, void * out, uint64_t cap, uint64_t * out_len)
{
  if (!out_len || (in_len && !in) || (cap && !out))
  {
    gtx::g_last_error = "gtx_bgzf_compress: bad argument";
    return GTX_ERR_ARG;
  }
  std::string res;
  uint8_t const * p = static_cast<uint8_t const *>(in);
  auto member = [&](uint8_t const * data, uint32_t n, std::string & res) -> bool
  {
    std::vector<uint8_t> buf(compressBound(n) + 64);
    z_stream zs{};
    if (deflateInit2(&zs, level < 0 ? Z_DEFAULT_COMPRESSION : std::min(level, 9), Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK)
      return false;
    zs.next_in = const_cast<Bytef *>(data);
    zs.avail_in = n;
    zs.next_out = buf.data();
    zs.avail_out = static_cast<uInt>(buf.size());
    int const rc = deflate(&zs, Z_FINISH);
    uint32_t const clen = static_cast<uint32_t>(zs.total_out);
    deflateEnd(&zs);
    if (rc != Z_STREAM_END || clen + 26u > 0x10000u)
      return false;
    uint8_t head[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 0, 0};
    uint16_t const bsize = static_cast<uint16_t>(clen + 25u);
    head[16] = static_cast<uint8_t>(bsize & 255u);
    head[17] = static_cast<uint8_t>(bsize >> 8);
    res.append(reinterpret_cast<char const *>(head), 18);
    res.append(reinterpret_cast<char const *>(buf.data()), clen);
    uint32_t const tail[2] = {static_cast<uint32_t>(crc32(crc32(0L, Z_NULL, 0), data, n)), n};
    res.append(reinterpret_cast<char const *>(tail), 8);
    return true;
  };
  // members are independent: beyond a megabyte of input they are made on a few threads, each a run of consecutive members,
  // and put together in order (the bytes are those of one thread)
  uint64_t const n_members = (in_len + 0xff00u - 1) / 0xff00u;
  unsigned const n_threads = n_members < 16 ? 1u : static_cast<unsigned>(std::min<uint64_t>(std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency() / 2)), n_members / 8));
  std::vector<std::string> parts(std::max(1u, n_threads));
  std::vector<char> failed(parts.size(), 0);
  auto run = [&](unsigned k)
  {
    uint64_t const m0 = n_members * k / parts.size(), m1 = n_members * (k + 1) / parts.size();
    for (uint64_t m = m0; m < m1 && !failed[k]; ++m)
    {
      uint64_t const at = m * 0xff00u;
      if (!member(p + at, static_cast<uint32_t>(std::min<uint64_t>(0xff00u, in_len - at)), parts[k]))
        failed[k] = 1;
    }
  };
  if (parts.size() == 1)
    run(0);
  else
  {
    std::vector<std::thread> team;
    for (unsigned k = 1; k < parts.size(); ++k)
      team.emplace_back(run, k);
    run(0);
    for (auto & t : team)
      t.join();
  }
  for (size_t k = 0; k < parts.size(); ++k)
  {
    if (failed[k])
    {
      gtx::g_last_error = "gtx_bgzf_compress: deflate failed";
      return GTX_ERR_IO;
    }
    res += parts[k];
  }
  if (with_eof)
  {
    // the end-of-file marker is a fixed member (SAM spec 4.1.2), whatever the level: deflating nothing at level 0 gives a stored
    // block and a member of 31 bytes, which htslib's bgzf_check_EOF does not take for the marker
    static unsigned char const EOF_MEMBER[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    res.append(reinterpret_cast<char const *>(EOF_MEMBER), sizeof EOF_MEMBER);
  }
  *out_len = res.size();
  if (res.size() > cap)
    return out ? GTX_ERR_CAPACITY : GTX_OK;
  std::memcpy(out, res.data(), res.size());
  return GTX_OK;
}
