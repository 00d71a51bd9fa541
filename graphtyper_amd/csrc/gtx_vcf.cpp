// gtx_vcf.cpp -- the VCF records of a genotyped region as text, host side.
//
// One record per variant site, what the reference writes for the Variant it makes of a haplotype:
//   Vcf::add_haplotype (src/typer/vcf.cpp:1507-1611) -> Variant::scan_calls (src/typer/variant.cpp:230-428) ->
//   Variant::generate_infos (variant.cpp:430-1096; VarStats::write_stats src/typer/var_stats.cpp:53-141) ->
//   Vcf::write_record (vcf.cpp:767-1149, PL / GQ through include/graphtyper/typer/binned_pl.hpp).
// Input: the per (haplotype, sample) calls of gtx_calls_batch and the accumulators of gtx_score_batch, as host copies.
// Here a site is a row of flat arrays, its INFO field a sorted list of (key, text) built once; numbers are printed with
// printf's %g, which is what the reference's string streams do at the precision they set.
// Not built: SV post-processing (reformat_sv_vcf_records) -- SV graphs are refused --, variant break-down and the merge of
// pools (vcf_operations.cpp), the description lines of the header.
#include "gtx_ctx.hpp"

#include <algorithm>
#include <zlib.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
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

void put_g(std::string & s, double v, int precision) // ostream << double at that precision, default float format
{
  char b[40];
  int const n = std::snprintf(b, sizeof(b), "%.*g", precision, v);
  s.append(b, static_cast<size_t>(n));
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

struct Info
{
  std::vector<std::pair<std::string, std::string>> kv;
  std::string & operator[](char const * k)
  {
    kv.emplace_back(k, std::string());
    return kv.back().second;
  }
  std::string const * find(char const * k) const
  {
    for (auto const & e : kv)
      if (e.first == k)
        return &e.second;
    return nullptr;
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

extern "C" int gtx_vcf_records(const gtx_ctx * c, const gtx_vcf_request * rq, char * out, uint64_t cap, uint64_t * len)
{
  if (!c || !rq || !len || (cap && !out) || !rq->contig || !rq->gt_cov || !rq->stat_u64 || !rq->stat_u32 || !rq->phred || !rq->calls ||
      (rq->n_samples && !rq->sample_names))
    return GTX_ERR_ARG;
  gtx::HostGraph const & g = c->graph;
  if (g.is_sv_graph)
  {
    gtx::g_last_error = "gtx_vcf_records: the SV post-processing of the calls (reformat_sv_vcf_records) is not built";
    return GTX_ERR_UNSUPPORTED;
  }
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
  std::string const contig = rq->contig;
  // The records of the sites are independent of each other: the sites are cut into contiguous ranges for a team of host
  // threads (every record walks the sample-major arrays of all samples -- cache misses, not arithmetic), every range writes
  // its own text, the texts are joined in order.
  auto write_sites = [&](uint32_t h_begin, uint32_t h_end, std::string & text, std::string & error) -> int
  {
  std::vector<AlleleStats> al;
  std::vector<std::pair<const char *, uint32_t>> seqs;
  std::vector<CallView> calls(ns);
  std::vector<double> qd_alt, aa_score;
  for (uint32_t h = h_begin; h < h_end; ++h)
  {
    uint32_t const cnum = g.ref_nvar[h], v0 = g.ref_first_var[h], n_tri = cnum * (cnum + 1) / 2;
    uint64_t const aoff = g.allele_off[h], toff = g.tri_off[h];
    uint32_t const pos = g.var_order[v0]; // Variant::Variant(Genotype): the site's position on its contig
    seqs.clear();
    size_t total_len = 0;
    for (uint32_t a = 0; a < cnum; ++a)
    {
      seqs.emplace_back(g.dna.data() + g.var_dna[v0 + a], g.var_len[v0 + a]);
      total_len += g.var_len[v0 + a];
    }
    // ---- VarStats of the haplotype + Variant::scan_calls over the samples
    al.assign(cnum, AlleleStats());
    uint64_t const hap_mapq_squared = rq->stat_u64[h];
    uint32_t const clipped_reads = rq->stat_u32[h];
    for (uint32_t a = 0; a < cnum; ++a)
    {
      const uint64_t * s64 = rq->stat_u64 + nh + 2 * (aoff + a);
      const uint32_t * s32 = rq->stat_u32 + nh + 6 * (aoff + a);
      al[a].clipped_bp = s64[0];
      al[a].mapq_squared = s64[1];
      al[a].score_diff = s32[0];
      al[a].mismatches = s32[1];
      al[a].r1f = s32[2];
      al[a].r1r = s32[3];
      al[a].r2f = s32[4];
      al[a].r2r = s32[5];
    }
    uint32_t n_genotyped = 0, n_passed = 0;
    uint64_t seqdepth = 0, qual = 0;
    uint32_t het_first = 0, het_second = 0, hom_first = 0, hom_second = 0;
    long qd_total_qual = 0, qd_total_depth = 0; // Variant::get_qual_by_depth (variant.cpp:1535-1559)
    for (uint32_t s = 0; s < ns; ++s)
    {
      CallView & cv = calls[s];
      cv.phred = rq->phred + static_cast<uint64_t>(s) * g.total_tri + toff;
      cv.cov = rq->gt_cov + static_cast<uint64_t>(s) * g.total_allele + aoff;
      cv.c = rq->calls + static_cast<uint64_t>(s) * nh + h;
      cv.cnum = cnum;
      cv.n_tri = n_tri;
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
    // ---- what is skipped (vcf.cpp:775-830, 1226-1258)
    if (pos < rq->region_begin || pos > rq->region_end)
      continue;
    if ((ns > 0 && cnum > 80) || total_len > 16000)
      continue;
    if (rq->filter_zero_qual && qual == 0)
      continue;
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
    std::sort(info.kv.begin(), info.kv.end(), [](auto const & a, auto const & b) { return a.first < b.first; });
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
    for (size_t i = 0; i < info.kv.size(); ++i)
    {
      if (i)
        text += ';';
      text += info.kv[i].first;
      if (!info.kv[i].second.empty())
      {
        text += '=';
        text += info.kv[i].second;
      }
    }
    if (ns)
    {
      text += "\tGT:AD:MD:DP:GQ:PL";
      for (uint32_t s = 0; s < ns; ++s)
      {
        CallView const & cv = calls[s];
        text += '\t';
        if (!cv.any_pl())
          text += "./.";
        else
        {
          put_u(text, cv.c->gt_first);
          text += '/';
          put_u(text, cv.c->gt_second);
        }
        for (uint32_t a = 0; a < cnum; ++a)
        {
          text += a ? ',' : ':';
          put_u(text, cv.coverage(a));
        }
        text += ':';
        put_u(text, cv.c->ambiguous_depth);
        text += ':';
        put_u(text, cv.unique_depth() + cv.c->ambiguous_depth);
        text += ':';
        put_u(text, std::min<uint16_t>(99, BINNED.v[cv.c->gq]));
        for (uint32_t i = 0; i < n_tri; ++i)
        {
          text += i ? ',' : ':';
          put_u(text, BINNED.v[cv.phred[i]]);
        }
      }
    }
    text += '\n';
  }
  return GTX_OK;
  };
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
    int const rc = write_sites(0, nh, text, error);
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
          status[t] = write_sites(b, e, part[t], error[t]);
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
  auto member = [&](uint8_t const * data, uint32_t n) -> bool
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
  for (uint64_t at = 0; at < in_len; at += 0xff00u)
    if (!member(p + at, static_cast<uint32_t>(std::min<uint64_t>(0xff00u, in_len - at))))
    {
      gtx::g_last_error = "gtx_bgzf_compress: deflate failed";
      return GTX_ERR_IO;
    }
  if (with_eof && !member(p, 0))
    return GTX_ERR_IO;
  *out_len = res.size();
  if (res.size() > cap)
    return out ? GTX_ERR_CAPACITY : GTX_OK;
  std::memcpy(out, res.data(), res.size());
  return GTX_OK;
}
