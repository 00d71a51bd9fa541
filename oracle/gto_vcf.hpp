// gto_vcf.hpp -- TEST INFRASTRUCTURE (oracle): the text of the VCF records the reference writes for the variant sites of a
// genotyped region -- a CPU restatement, function by function, of
//   Vcf::add_haplotype            src/typer/vcf.cpp:1507-1530, 1597-1611   (haplotype -> Variant with one SampleCall per sample)
//   SampleCall                    src/typer/sample_call.cpp:34-170
//   Variant::scan_calls           src/typer/variant.cpp:230-428
//   VarStats::write_stats         src/typer/var_stats.cpp:53-141, 425-452
//   Variant::generate_infos       src/typer/variant.cpp:430-1096           (genotype calling: no segment / long-read calling)
//   determine_variant_type / get_qual / get_qual_by_depth[_per_alt_allele]   variant.cpp:1430-1576
//   get_logf / get_aa_score       include/graphtyper/typer/logistic_constants.hpp
//   p_hwe_excess_het              src/utilities/snp_hwe.cpp:19-111
//   Vcf::write_record             src/typer/vcf.cpp:767-1149, binned_pl: include/graphtyper/typer/binned_pl.hpp
//   Vcf::write_records            src/typer/vcf.cpp:1161-1275 (sites of one region are unique and sorted: the region filter)
// with the std containers and string streams the reference uses.  Parity unpinned: the reference's tests hold no VCF text
// (test/typer/test_vcf.cpp checks sample names and allele sequences only).  Not restated: the SV post-processing
//   vcf_merge_and_filter          src/typer/vcf_operations.cpp:278-478 (sites(): the alleles the next iteration keeps, with their tags)
// (reformat_sv_vcf_records: gto_sv.hpp), variant break-down (vcf_operations.cpp:480-), the header's description lines.
// Only tests/ may use this file (through oracle/libgto.so).
#pragma once
#include "gto.hpp"

#include <cmath>
#include <map>
#include <sstream>

namespace gto
{
namespace vcf
{
// include/graphtyper/typer/binned_pl.hpp as runs: (number of PL values, the value they are written as)
inline uint16_t binned_pl(unsigned pl)
{
  static const uint16_t runs[][2] = {{1, 0},  {2, 1},  {2, 3},  {3, 6},  {3, 9},   {3, 12},  {4, 15},  {5, 20},  {5, 25},  {5, 30},
                                     {5, 35}, {7, 40}, {10, 50}, {13, 60}, {12, 75}, {33, 99}, {25, 125}, {37, 150}, {53, 200}, {28, 255}};
  unsigned first = 0;
  for (auto const & r : runs)
  {
    if (pl < first + r[0])
      return r[1];
    first += r[0];
  }
  return 255;
}

struct SampleCall // sample_call.hpp / sample_call.cpp:34-170
{
  std::vector<uint8_t> phred;
  std::vector<uint16_t> coverage;
  uint16_t ref_total_depth = 0, alt_total_depth = 0;
  uint8_t ambiguous_depth = 0, alt_proper_pair_depth = 0;
  mutable int8_t filter = -1;

  uint32_t get_depth() const { return std::accumulate(coverage.begin(), coverage.end(), static_cast<uint32_t>(ambiguous_depth)); }
  uint32_t get_unique_depth() const { return std::accumulate(coverage.begin(), coverage.end(), 0u); }
  uint32_t get_alt_depth() const { return std::accumulate(coverage.begin() + 1, coverage.end(), static_cast<uint32_t>(ambiguous_depth)); }
  std::pair<uint16_t, uint16_t> get_gt_call() const
  {
    if (phred.empty())
      return {0, 0};
    std::size_t i = 0;
    for (std::size_t y = 0; y < coverage.size(); ++y)
      for (std::size_t x = 0; x <= y; ++x, ++i)
        if (phred[i] == 0)
          return {static_cast<uint16_t>(x), static_cast<uint16_t>(y)};
    return {0, 0};
  }
  uint8_t get_gq() const
  {
    bool seen_zero = false;
    uint8_t next_lowest = 255;
    for (auto const p : phred)
    {
      if (p == 0)
      {
        if (!seen_zero)
          seen_zero = true;
        else
          return 0;
      }
      else if (p < next_lowest)
        next_lowest = p;
    }
    return next_lowest;
  }
  uint8_t get_lowest_phred_not_with(uint16_t allele) const
  {
    long i = 0;
    uint8_t min_phred = 255;
    for (long y = 0; y < static_cast<long>(coverage.size()); ++y)
    {
      if (y == allele)
      {
        i += y + 1;
        continue;
      }
      for (long x = 0; x <= y; ++x, ++i)
      {
        if (x == allele)
          continue;
        if (phred[i] < min_phred)
          min_phred = phred[i];
      }
    }
    return min_phred;
  }
  long check_filter(long gq) const
  {
    if (filter < 0)
      filter = gq >= 30 ? 0 : gq >= 20 ? 1 : gq >= 10 ? 2 : 3;
    return filter;
  }
};

struct PerAllele // var_stats.hpp:15-33
{
  uint64_t clipped_bp = 0, mapq_squared = 0;
  uint32_t score_diff = 0, mismatches = 0;
  uint64_t qd_qual = 0, qd_depth = 0, total_depth = 0;
  uint32_t ac = 0, pass_ac = 0, n_ref_ref = 0, n_ref_alt = 0, n_alt_alt = 0;
  uint16_t maximum_alt_support = 0;
  double maximum_alt_support_ratio = 0;
  std::pair<uint32_t, uint32_t> het_multi_allele_depth{0, 0}, hom_multi_allele_depth{0, 0};
};

struct VarStats // var_stats.hpp:56-84
{
  std::vector<PerAllele> per_allele;
  std::vector<ReadStrand> read_strand;
  uint32_t clipped_reads = 0;
  uint64_t mapq_squared = 0;
  uint32_t n_genotyped = 0, n_calls = 0, n_passed_calls = 0;
  uint8_t n_max_alt_proper_pairs = 0;
  uint64_t seqdepth = 0;
  std::pair<uint32_t, uint32_t> het_allele_depth{0, 0}, hom_allele_depth{0, 0};

  void write_stats(std::map<std::string, std::string> & infos) const // var_stats.cpp:53-141
  {
    long const n = static_cast<long>(per_allele.size());
    if (n <= 1)
      return;
    infos["CR"] = std::to_string(clipped_reads);
    infos["MQsquared"] = std::to_string(mapq_squared);
    {
      std::ostringstream f, r, f1, f2, r1, r2;
      for (long i = 0; i < n; ++i)
      {
        auto const & s = read_strand[i];
        char const * sep = i ? "," : "";
        f << sep << (s.r1_forward + s.r2_forward);
        r << sep << (s.r1_reverse + s.r2_reverse);
        f1 << sep << s.r1_forward;
        f2 << sep << s.r2_forward;
        r1 << sep << s.r1_reverse;
        r2 << sep << s.r2_reverse;
      }
      infos["SBF"] = f.str();
      infos["SBR"] = r.str();
      infos["SBF1"] = f1.str();
      infos["SBF2"] = f2.str();
      infos["SBR1"] = r1.str();
      infos["SBR2"] = r2.str();
    }
    {
      std::ostringstream cr, mq, sd, mm;
      for (long i = 0; i < n; ++i)
      {
        auto const & a = per_allele[i];
        char const * sep = i ? "," : "";
        cr << sep << a.clipped_bp;
        mq << sep << a.mapq_squared;
        sd << sep << a.score_diff;
        mm << sep << a.mismatches;
      }
      infos["CRal"] = cr.str();
      infos["MQSal"] = mq.str();
      infos["SDal"] = sd.str();
      infos["MMal"] = mm.str();
    }
  }
};

// var_stats.cpp:425-452: the per-allele lists are read back from their text
inline long accumulated(std::map<std::string, std::string> const & infos, std::string const & key, bool alt_only)
{
  auto it = infos.find(key);
  if (it == infos.end())
    return 0;
  std::vector<long> nums;
  std::stringstream ss(it->second);
  std::string tok;
  while (std::getline(ss, tok, ','))
    nums.push_back(std::strtoll(tok.c_str(), nullptr, 10));
  if (nums.empty())
    return 0;
  return std::accumulate(nums.begin() + (alt_only ? 1 : 0), nums.end(), 0l);
}

inline double p_hwe_excess_het(int obs_hets, int obs_hom1, int obs_hom2) // snp_hwe.cpp:19-111
{
  if (obs_hets == 0 && (obs_hom1 == 0 || obs_hom2 == 0))
    return 1.0;
  int const obs_homc = obs_hom1 < obs_hom2 ? obs_hom2 : obs_hom1;
  int const obs_homr = obs_hom1 < obs_hom2 ? obs_hom1 : obs_hom2;
  int const rare_copies = 2 * obs_homr + obs_hets;
  int const genotypes = obs_hets + obs_homc + obs_homr;
  std::vector<double> het_probs(rare_copies + 1, 0.0);
  double const mid_double = static_cast<double>(rare_copies) * (static_cast<double>(2 * genotypes - rare_copies) / static_cast<double>(2 * genotypes));
  int mid = static_cast<int>(mid_double);
  if ((rare_copies & 1) ^ (mid & 1))
    mid++;
  int curr_hets = mid;
  int curr_homr = (rare_copies - mid) / 2;
  int curr_homc = genotypes - curr_hets - curr_homr;
  het_probs[mid] = 1.0;
  double sum = het_probs[mid];
  for (; curr_hets > 1; curr_hets -= 2)
  {
    het_probs[curr_hets - 2] = het_probs[curr_hets] * curr_hets * (curr_hets - 1.0) / (4.0 * (curr_homr + 1.0) * (curr_homc + 1.0));
    sum += het_probs[curr_hets - 2];
    curr_homr++;
    curr_homc++;
  }
  curr_hets = mid;
  curr_homr = (rare_copies - mid) / 2;
  curr_homc = genotypes - curr_hets - curr_homr;
  for (curr_hets = mid; curr_hets <= rare_copies - 2; curr_hets += 2)
  {
    het_probs[curr_hets + 2] = het_probs[curr_hets] * 4.0 * curr_homr * curr_homc / ((curr_hets + 2.0) * (curr_hets + 1.0));
    sum += het_probs[curr_hets + 2];
    curr_homr--;
    curr_homc--;
  }
  for (int i = 0; i <= rare_copies; i++)
    het_probs[i] /= sum;
  double p_hi = 0.0;
  for (int i = obs_hets; i <= rare_copies; i++)
    p_hi += het_probs[i];
  return p_hi > 1.0 ? 1.0 : p_hi;
}

inline double get_logf(double abhom, double cr_by_seqdepth, double mq, double pass_ratio, double gt_yield, double qd, long ab_het_bin, long sbalt_bin)
{ // logistic_constants.hpp:8-50
  static const double ABHet[11] = {-6.03446, -6.03446, -1.35948, -0.84956, -0.28956, 0.0, -1.05013, -1.35024, -1.34475, -3.74512, -3.74512};
  static const double SBAlt[11] = {-0.32486, -0.32486, -0.25342, -0.32696, 0.02442, 0.0, -0.33522, -0.41332, -0.74043, -1.60844, -1.60844};
  double const pwr = -29.28908 + abhom * 23.12909 + cr_by_seqdepth * -10.22658 + mq * 0.01024 + pass_ratio * 0.85320 + gt_yield * 4.91178 +
                     qd * 0.23215 + ABHet[ab_het_bin] + SBAlt[sbalt_bin];
  double const e = std::max(0.0, std::exp(-pwr));
  return 1.0 / (1.0 + e);
}

inline double get_aa_score(double abhom, double sb, double mm, long sd, double qd, double cr, long mq) // logistic_constants.hpp:52-92
{
  static const double ABHom[5] = {0.0, 1.304140117, 1.681221065, 2.214801195, 3.930106559};
  long bin = 4;
  if (abhom <= 0.85)
    bin = 0;
  else if (abhom <= 0.94)
    bin = 1;
  else if (abhom <= 0.98)
    bin = 2;
  else if (abhom <= 0.99)
    bin = 3;
  if (mq > 60)
    mq = 60;
  double const pwr = -6.347426707 + ABHom[bin] + sb * -0.25233400 + mm * -0.04129973 + sd * 0.014572295 + qd * 0.065221319 + cr * -0.01934834 +
                     mq * 0.055973424;
  return 1.0 / (1.0 + std::exp(-pwr));
}

struct Variant
{
  uint32_t abs_pos = 0; // one contig: its 1-based position
  std::vector<std::string> seqs;
  std::vector<SampleCall> calls;
  VarStats stats;
  std::map<std::string, std::string> infos;
  std::string suffix_id;
  int32_t hap_id = -1; // variant.hpp:30; Vcf::add_haplotype sets it to the haplotype's index (vcf.cpp:1509-1510)

  bool is_sv() const // variant.cpp:1098-1118
  {
    for (std::size_t s = 1; s < seqs.size(); ++s)
    {
      auto const & seq = seqs[s];
      if (seq.size() < 5)
        continue;
      if (seq[0] == '<' || (seq.size() > 100 && seq.find('<') != std::string::npos))
        return true;
    }
    return false;
  }

  std::string determine_variant_type() const // variant.cpp:1430-1520
  {
    enum { NOT_SV, DEL, DUP, INS, BND, OTHER } sv = NOT_SV;
    std::size_t num_non_ones = 0;
    for (auto const & s : seqs)
      if (s.size() > 1)
      {
        if (s.size() > 4 && s[0] == '<')
        {
          std::string const type(s.begin() + 1, s.begin() + 4);
          if (type == "DEL" && (sv == NOT_SV || sv == DEL))
            sv = DEL;
          else if (type == "DUP" && (sv == NOT_SV || sv == DUP))
            sv = DUP;
          else if (type == "INS" && (sv == NOT_SV || sv == INS))
            sv = INS;
          else
            sv = OTHER;
        }
        else if (s.find_first_of("[]") != std::string::npos)
          sv = (sv == NOT_SV || sv == BND) ? BND : OTHER;
        else
          ++num_non_ones;
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
    if (num_non_ones == 0)
      return "SG";
    if (seqs.size() - num_non_ones == 1)
      return "IG";
    if (seqs.size() - num_non_ones == 2 && seqs.back().size() == 1 && seqs.back()[0] == '*')
      return "IG";
    return "XG";
  }

  uint64_t get_qual() const // variant.cpp:1522-1533
  {
    uint64_t q = 0;
    for (auto const & c : calls)
      if (!c.phred.empty())
        q += c.phred[0];
    return q;
  }

  double get_qual_by_depth() const // variant.cpp:1535-1559
  {
    long total_qual = 0, total_depth = 0;
    for (auto const & c : calls)
      if (!c.phred.empty() && c.phred[0] > 0)
      {
        long const depth = std::min(10l, static_cast<long>(c.get_alt_depth()));
        if (depth > 0)
        {
          total_qual += std::min(25l * depth, static_cast<long>(c.phred[0]));
          total_depth += depth;
        }
      }
    return total_depth == 0 ? 0.0 : static_cast<double>(total_qual) / static_cast<double>(total_depth);
  }

  void scan_calls() // variant.cpp:230-428
  {
    if (stats.seqdepth > 0 || stats.n_calls > 0)
      return;
    if (stats.per_allele.empty())
    {
      stats.per_allele.resize(seqs.size());
      stats.read_strand.resize(seqs.size());
    }
    long const num_alts = static_cast<long>(seqs.size()) - 1;
    stats.n_calls += static_cast<uint32_t>(calls.size());
    for (auto const & sc : calls)
    {
      if (!sc.phred.empty() && sc.phred[0] > 0)
      {
        auto const gt = sc.get_gt_call();
        auto const & cov = sc.coverage;
        if (gt.first > 0)
        {
          auto & pa = stats.per_allele[gt.first];
          long const depth = std::min(10l, static_cast<long>(cov[gt.first] + sc.ambiguous_depth));
          if (depth > 0)
          {
            pa.qd_qual += std::min(25l * depth, static_cast<long>(sc.get_lowest_phred_not_with(gt.first)));
            pa.qd_depth += depth;
          }
        }
        if (gt.first != gt.second)
        {
          auto & pa = stats.per_allele[gt.second];
          long const depth = std::min(10l, static_cast<long>(cov[gt.second] + sc.ambiguous_depth));
          if (depth > 0)
          {
            pa.qd_qual += std::min(25l * depth, static_cast<long>(sc.get_lowest_phred_not_with(gt.second)));
            pa.qd_depth += depth;
          }
        }
      }
      stats.n_max_alt_proper_pairs = std::max(stats.n_max_alt_proper_pairs, sc.alt_proper_pair_depth);
      uint32_t const total_depth = std::accumulate(sc.coverage.cbegin(), sc.coverage.cend(), 0u);
      auto const call = sc.get_gt_call();
      for (long c = 0; c < num_alts; ++c)
      {
        auto & pa = stats.per_allele[c + 1];
        pa.maximum_alt_support = std::max(pa.maximum_alt_support, sc.coverage[c + 1]);
        if (total_depth > 0)
        {
          double const ratio = static_cast<double>(sc.coverage[c + 1]) / static_cast<double>(total_depth);
          pa.maximum_alt_support_ratio = std::max(pa.maximum_alt_support_ratio, ratio);
        }
        if (call.first == (c + 1) || call.second == (c + 1))
        {
          if (call.second == call.first)
            ++pa.n_alt_alt;
          else
            ++pa.n_ref_alt;
        }
        else
          ++pa.n_ref_ref;
      }
      long const gq = sc.get_gq();
      long const filter = sc.check_filter(gq);
      if (std::find_if(sc.phred.begin(), sc.phred.end(), [](uint8_t pl) { return pl != 0; }) != sc.phred.end())
        ++stats.n_genotyped;
      if (filter == 0)
        ++stats.n_passed_calls;
      if (call.first != call.second)
      {
        stats.het_allele_depth.first += sc.coverage[call.first];
        stats.het_allele_depth.second += sc.coverage[call.second];
      }
      else
      {
        stats.hom_allele_depth.first += sc.coverage[call.first];
        stats.hom_allele_depth.second += std::accumulate(sc.coverage.cbegin(), sc.coverage.cend(), 0ull) - sc.coverage[call.first];
      }
      {
        uint32_t const call_depth = sc.get_unique_depth();
        if (call.first != call.second)
        {
          auto const c1 = call.first, c2 = call.second;
          stats.per_allele[c1].het_multi_allele_depth.first += sc.coverage[c1];
          stats.per_allele[c1].het_multi_allele_depth.second += call_depth - sc.coverage[c1];
          stats.per_allele[c2].het_multi_allele_depth.first += sc.coverage[c2];
          stats.per_allele[c2].het_multi_allele_depth.second += call_depth - sc.coverage[c2];
        }
        else
        {
          auto const c = call.first;
          stats.per_allele[c].hom_multi_allele_depth.first += sc.coverage[c];
          stats.per_allele[c].hom_multi_allele_depth.second += call_depth - sc.coverage[c];
        }
      }
      if (!sc.coverage.empty())
      {
        stats.seqdepth += sc.get_depth();
        for (long c = 1; c < static_cast<long>(sc.coverage.size()); ++c)
          stats.per_allele[c].total_depth += sc.coverage[c];
      }
      ++stats.per_allele[call.first].ac;
      ++stats.per_allele[call.second].ac;
      if (filter == 0)
      {
        ++stats.per_allele[call.first].pass_ac;
        ++stats.per_allele[call.second].pass_ac;
      }
    }
  }

  // returns is_good_alt (one flag per alternative allele)
  std::vector<int8_t> generate_infos(bool is_sv_graph) // variant.cpp:430-1096
  {
    long const num_seqs = static_cast<long>(seqs.size());
    long const num_alts = num_seqs - 1;
    std::vector<int8_t> is_good_alt(static_cast<std::size_t>(num_alts), 1);
    bool const is_stats = !stats.per_allele.empty();
    if (is_stats)
    {
      scan_calls();
      stats.write_stats(infos);
    }
    else
    {
      stats.per_allele.resize(num_seqs);
      stats.read_strand.resize(num_seqs);
      scan_calls();
    }
    infos["RefLen"] = std::to_string(seqs[0].size());
    {
      auto const end_it = infos.find("END"); // variant.cpp:467-481: END is never in front of POS (one contig: abs_pos is the position)
      if (end_it != infos.end())
        end_it->second = std::to_string(std::max(std::strtol(end_it->second.c_str(), nullptr, 10), static_cast<long>(abs_pos)));
    }
    auto const & pa = stats.per_allele;
    {
      std::stringstream ss;
      ss << static_cast<uint16_t>(pa[1].maximum_alt_support);
      for (long e = 2; e < num_seqs; ++e)
        ss << ',' << static_cast<uint16_t>(pa[e].maximum_alt_support);
      infos["MaxAAS"] = ss.str();
    }
    {
      std::stringstream ss;
      ss.precision(4);
      ss << pa[1].maximum_alt_support_ratio;
      for (long e = 2; e < num_seqs; ++e)
        ss << ',' << pa[e].maximum_alt_support_ratio;
      infos["MaxAASR"] = ss.str();
    }
    {
      std::stringstream rr, ra, aa, ex;
      for (long e = 1; e < num_seqs; ++e)
      {
        char const * sep = e > 1 ? "," : "";
        rr << sep << pa[e].n_ref_ref;
        ra << sep << pa[e].n_ref_alt;
        aa << sep << pa[e].n_alt_alt;
        ex << sep << p_hwe_excess_het(pa[e].n_ref_alt, pa[e].n_ref_ref, pa[e].n_alt_alt);
      }
      infos["NHomRef"] = rr.str();
      infos["NHet"] = ra.str();
      infos["NHomAlt"] = aa.str();
      infos["PexcessHet"] = ex.str();
    }
    if (is_sv())
      infos["MaxAltPP"] = std::to_string(static_cast<uint16_t>(stats.n_max_alt_proper_pairs));
    {
      std::ostringstream ss;
      ss << pa[1].ac;
      for (long e = 2; e < num_seqs; ++e)
        ss << ',' << pa[e].ac;
      infos["AC"] = ss.str();
    }
    infos["AN"] = std::to_string(2 * stats.n_genotyped);
    {
      std::ostringstream ss;
      ss.precision(4);
      for (long e = 1; e < num_seqs; ++e)
      {
        if (e > 1)
          ss << ',';
        if (stats.n_genotyped > 0)
          ss << static_cast<double>(static_cast<double>(pa[e].ac) / static_cast<double>(2 * stats.n_genotyped));
        else
          ss << "0.0";
      }
      infos["AF"] = ss.str();
    }
    {
      std::ostringstream ss;
      ss << pa[1].pass_ac;
      for (long e = 2; e < num_seqs; ++e)
        ss << ',' << pa[e].pass_ac;
      infos["PASS_AC"] = ss.str();
    }
    infos["PASS_AN"] = std::to_string(2 * stats.n_passed_calls);
    double info_pass_ratio = 0.0;
    if (stats.n_genotyped > 0)
    {
      info_pass_ratio = static_cast<double>(stats.n_passed_calls) / static_cast<double>(stats.n_genotyped);
      std::stringstream ss;
      ss.precision(4);
      ss << info_pass_ratio;
      infos["PASS_ratio"] = ss.str();
    }
    infos["SeqDepth"] = std::to_string(stats.seqdepth);
    double info_ab_het = 0.5;
    {
      std::stringstream ss;
      ss.precision(4);
      uint32_t const total = stats.het_allele_depth.first + stats.het_allele_depth.second;
      if (total > 0)
      {
        info_ab_het = static_cast<double>(stats.het_allele_depth.second) / static_cast<double>(total);
        ss << info_ab_het;
      }
      else
        ss << "-1";
      infos["ABHet"] = ss.str();
    }
    double info_abhom = 0.985;
    {
      std::stringstream ss;
      ss.precision(4);
      uint32_t const total = stats.hom_allele_depth.first + stats.hom_allele_depth.second;
      if (total > 0)
      {
        info_abhom = static_cast<double>(stats.hom_allele_depth.first) / static_cast<double>(total);
        ss << info_abhom;
      }
      else
        ss << "-1";
      infos["ABHom"] = ss.str();
    }
    {
      uint32_t const f = static_cast<uint32_t>(accumulated(infos, "SBF", false)), r = static_cast<uint32_t>(accumulated(infos, "SBR", false));
      std::stringstream ss;
      ss.precision(4);
      if (f + r == 0)
        ss << "-1";
      else
        ss << (static_cast<double>(f) / static_cast<double>(f + r));
      infos["SB"] = ss.str();
    }
    double info_sbalt = 0.0;
    {
      uint32_t const f = static_cast<uint32_t>(accumulated(infos, "SBF", true)), r = static_cast<uint32_t>(accumulated(infos, "SBR", true));
      std::stringstream ss;
      ss.precision(4);
      if (f + r == 0)
        ss << "-1";
      else
      {
        info_sbalt = static_cast<double>(f) / static_cast<double>(f + r);
        ss << info_sbalt;
      }
      infos["SBAlt"] = ss.str();
    }
    {
      std::stringstream het, hom;
      het.precision(4);
      hom.precision(4);
      for (std::size_t i = 0; i < pa.size(); ++i)
      {
        if (i > 0)
        {
          het << ",";
          hom << ",";
        }
        auto const & h = pa[i].het_multi_allele_depth;
        if (h.first + h.second > 0)
          het << (static_cast<double>(h.second) / static_cast<double>(h.first + h.second));
        else
          het << "-1";
        auto const & o = pa[i].hom_multi_allele_depth;
        if (o.first + o.second > 0)
          hom << (static_cast<double>(o.first) / static_cast<double>(o.first + o.second));
        else
          hom << "-1";
      }
      infos["ABHetMulti"] = het.str();
      infos["ABHomMulti"] = hom.str();
    }
    infos["VarType"] = determine_variant_type();
    double info_qd = 0.0;
    {
      std::stringstream ss;
      ss.precision(4);
      info_qd = get_qual_by_depth();
      ss << info_qd;
      infos["QD"] = ss.str();
    }
    std::vector<double> qd_alt(num_alts, 0.0); // variant.cpp:1561-1576
    for (long s = 0; s < num_alts; ++s)
      if (pa[s + 1].qd_depth > 0)
        qd_alt[s] = static_cast<double>(pa[s + 1].qd_qual) / static_cast<double>(pa[s + 1].qd_depth);
    {
      std::stringstream ss;
      ss.precision(4);
      ss << qd_alt[0];
      for (long q = 1; q < num_alts; ++q)
        ss << ',' << qd_alt[q];
      infos["QDalt"] = ss.str();
    }
    long info_mq = 60;
    if (stats.seqdepth > 0)
    {
      double const mapq = std::sqrt(static_cast<double>(stats.mapq_squared) / static_cast<double>(stats.seqdepth));
      info_mq = std::lround(mapq);
      infos["MQ"] = std::to_string(info_mq);
    }
    else
      infos["MQ"] = "0";
    if (is_sv_graph) // variant.cpp:861-884
    {
      for (char const * k : {"ABHetMulti", "ABHomMulti", "CR", "QDalt", "MQ", "MQsquared", "SB", "SBAlt", "SBF", "SBR", "SBF1", "SBF2", "SBR1", "SBR2"})
        infos.erase(k);
      for (long a = 1; a < num_seqs; ++a)
        is_good_alt[a - 1] = static_cast<int8_t>(pa[a].ac > 0);
      return is_good_alt;
    }
    if (!is_stats)
      return is_good_alt;
    {
      std::ostringstream sd, mm, cr, mq;
      for (long s = 1; s < num_seqs; ++s)
      {
        auto const & a = pa[s];
        if (s > 1)
        {
          sd << ',';
          mm << ',';
          cr << ',';
          mq << ',';
        }
        if (a.total_depth > 0)
        {
          double const d = static_cast<double>(a.total_depth);
          sd << (static_cast<double>(a.score_diff) / d);
          mm << (static_cast<double>(a.mismatches) / d / 10.0);
          cr << (static_cast<double>(a.clipped_bp) / d / 10.0);
          mq << std::lround(std::sqrt(static_cast<double>(a.mapq_squared) / d));
        }
        else
        {
          sd << "0.0";
          mm << "0.0";
          cr << "0.0";
          mq << "0";
        }
      }
      infos["SDalt"] = sd.str();
      infos["MMalt"] = mm.str();
      infos["CRalt"] = cr.str();
      infos["MQalt"] = mq.str();
    }
    std::vector<double> aa_score(num_alts);
    for (long s = 0; s < num_alts; ++s)
    {
      auto const & a = pa[s + 1];
      double const qd = qd_alt[s];
      if (a.total_depth > 0 && qd > 0.1 && a.maximum_alt_support >= 2 && a.maximum_alt_support_ratio >= 0.15)
      {
        double const depth = static_cast<double>(a.total_depth);
        uint64_t const reverse = static_cast<uint64_t>(stats.read_strand[s + 1].r1_reverse) + stats.read_strand[s + 1].r2_reverse; // read_strand.hpp get_reverse_count
        double const sb0 = 2.0 * ((static_cast<double>(reverse) / depth) - 0.5);
        double const sb = sb0 >= 0.0 ? sb0 : -sb0;
        double const mm = static_cast<double>(a.mismatches) / depth / 10.0;
        long const sd = std::lround(static_cast<double>(a.score_diff) / depth);
        double const cr = static_cast<double>(a.clipped_bp) / depth / 10.0;
        long const mq = std::lround(std::sqrt(static_cast<double>(a.mapq_squared) / depth));
        double score = get_aa_score(info_abhom, sb, mm, sd, qd, cr, mq);
        if (mm > 1.5)
        {
          double m = 1.0 - ((mm - 1.5) / 20.0);
          m = m <= 0.5 ? 0.5 : m;
          score *= m;
        }
        if ((cr + mm) > 2.5)
        {
          double m = 1.0 - ((cr + mm - 2.5) / 40.0);
          m = m <= 0.5 ? 0.5 : m;
          score *= m;
        }
        aa_score[s] = score;
      }
      else
        aa_score[s] = 0.0;
    }
    {
      std::ostringstream ss;
      ss.precision(4);
      ss << aa_score[0];
      for (long s = 1; s < num_alts; ++s)
        ss << "," << aa_score[s];
      infos["AAScore"] = ss.str();
    }
    {
      long const info_cr = infos.count("CR") > 0 ? std::stol(infos.at("CR")) : 0;
      long const ab_het_bin = static_cast<long>(info_ab_het * 10.0 + 0.00001);
      long const sbalt_bin = static_cast<long>(info_sbalt * 10.0 + 0.00001);
      double const cr_by_seqdepth = static_cast<double>(info_cr) / static_cast<double>(stats.seqdepth);
      double const gt_yield = static_cast<double>(stats.n_genotyped) / static_cast<double>(stats.n_calls);
      double const logf = get_logf(info_abhom, cr_by_seqdepth, static_cast<double>(info_mq), info_pass_ratio, gt_yield, info_qd, ab_het_bin, sbalt_bin);
      std::ostringstream ss;
      ss.precision(4);
      ss << logf;
      infos["LOGF"] = ss.str();
    }
    for (long a = 0; a < num_alts; ++a) // variant.cpp:1036-1063
    {
      auto const & per_al = pa[a + 1];
      if (per_al.total_depth == 0)
      {
        is_good_alt[a] = 0;
        continue;
      }
      double const qd = qd_alt[a];
      is_good_alt[a] = static_cast<int8_t>(qd >= 1.0 && per_al.maximum_alt_support >= 2 &&
                                           (seqs.size() < 71 || (qd >= 1.5 && per_al.maximum_alt_support_ratio >= 0.2)) &&
                                           (seqs.size() < 131 || (qd >= 2.0 && per_al.maximum_alt_support_ratio >= 0.225)));
    }
    return is_good_alt;
  }
};

struct WriteOptions
{
  std::string contig;
  std::vector<std::string> sample_names;
  uint32_t region_begin = 0, region_end = 0xFFFFFFFFu; // 1-based, inclusive (vcf.cpp:1277-1295)
  bool filter_zero_qual = false;
  std::string variant_suffix_id;
};

// `suffix`: what Vcf::write_records puts behind the ID of a record that shares position and type with the one in front of it
inline void write_record(std::ostream & out, Variant const & var, WriteOptions const & o, bool is_sv_graph, std::string const & suffix = std::string()) // vcf.cpp:767-1149
{
  (void)is_sv_graph;
  if (!var.calls.empty() && var.seqs.size() > 80)
    return;
  {
    std::size_t total = 0;
    for (auto const & s : var.seqs)
    {
      total += s.size();
      if (total > 16000)
        return;
    }
  }
  uint64_t const qual = var.get_qual();
  if (o.filter_zero_qual && qual == 0)
    return;
  bool const is_sv = var.is_sv();
  out << o.contig << '\t' << var.abs_pos << '\t' << o.contig << ':' << var.abs_pos << ':' << var.determine_variant_type();
  if (!var.suffix_id.empty())
    out << "[" << var.suffix_id << "]";
  out << suffix;
  out << '\t' << var.seqs[0] << '\t' << var.seqs[1];
  for (std::size_t a = 2; a < var.seqs.size(); ++a)
    out << ',' << var.seqs[a];
  out << "\t" << std::to_string(qual) << "\t";
  auto const & infos = var.infos;
  if (o.sample_names.empty())
    out << ".\t";
  else if (is_sv)
  {
    bool pass = true;
    if (infos.count("QD") == 1 && std::stod(infos.at("QD")) < 6.0)
    {
      out << "LowQD";
      pass = false;
    }
    if (qual < 10)
    {
      out << (pass ? "" : ";") << "LowQUAL";
      pass = false;
    }
    if (infos.count("AN") == 1 && infos.count("PASS_AC") == 1 && infos.count("PASS_ratio") == 1 &&
        (std::stoi(infos.at("AN")) >= 100 && (infos.at("PASS_AC") == "0" || std::stod(infos.at("PASS_ratio")) < 0.01)))
    {
      out << (pass ? "" : ";") << "LowPratio";
      pass = false;
    }
    if (pass)
      out << "PASS";
    out << "\t";
  }
  else
  {
    bool pass = true;
    if (infos.count("ABHet") == 1 && infos.at("ABHet") != "-1" && std::stod(infos.at("ABHet")) < 0.175)
    {
      out << "LowABHet";
      pass = false;
    }
    if (infos.count("ABHom") == 1 && infos.at("ABHom") != "-1" && std::stod(infos.at("ABHom")) < 0.85)
    {
      out << (pass ? "" : ";") << "LowABHom";
      pass = false;
    }
    if (infos.count("AN") == 1 && std::stoi(infos.at("AN")) >= 6 && infos.count("QD") == 1 && std::stod(infos.at("QD")) < 6.0)
    {
      out << (pass ? "" : ";") << "LowQD";
      pass = false;
    }
    auto aa = infos.find("AAScore");
    if (infos.count("AN") == 1 && std::stoi(infos.at("AN")) >= 6 && aa != infos.end())
    {
      std::stringstream ss(aa->second);
      bool good = false;
      for (double num = 0.0; ss >> num;)
      {
        if (num > 0.15)
          good = true;
        if (ss.peek() == ',')
          ss.ignore();
      }
      if (!good)
      {
        out << (pass ? "" : ";") << "LowAAScore";
        pass = false;
      }
    }
    if (qual < 10)
    {
      out << (pass ? "" : ";") << "LowQUAL";
      pass = false;
    }
    if (infos.count("AN") == 1 && infos.count("PASS_ratio") == 1 && (std::stoi(infos.at("AN")) >= 500 && std::stod(infos.at("PASS_ratio")) < 0.05))
    {
      out << (pass ? "" : ";") << "LowPratio";
      pass = false;
    }
    if (pass)
      out << "PASS";
    out << "\t";
  }
  if (infos.empty())
    out << ".";
  else
  {
    bool first = true;
    for (auto const & kv : infos)
    {
      if (!first)
        out << ';';
      first = false;
      out << kv.first;
      if (!kv.second.empty())
        out << '=' << kv.second;
    }
  }
  if (!var.calls.empty())
  {
    out << (is_sv ? "\tGT:FT:AD:MD:DP:RA:PP:GQ:PL" : "\tGT:AD:MD:DP:GQ:PL");
    for (auto const & call : var.calls)
    {
      if (std::find_if(call.phred.begin(), call.phred.end(), [](uint8_t pl) { return pl != 0; }) == call.phred.end())
        out << "\t./.";
      else
      {
        auto const gt = call.get_gt_call();
        out << "\t" << gt.first << "/" << gt.second;
      }
      long const gq = call.get_gq();
      if (is_sv)
      {
        long const filter = call.check_filter(gq);
        if (filter == 0)
          out << ":PASS";
        else
          out << ":FAIL" << filter;
      }
      out << ":" << call.coverage[0];
      for (std::size_t a = 1; a < call.coverage.size(); ++a)
        out << "," << call.coverage[a];
      out << ":" << static_cast<uint16_t>(call.ambiguous_depth);
      out << ":" << call.get_depth();
      if (is_sv)
      {
        out << ":" << call.ref_total_depth << "," << call.alt_total_depth;
        out << ':' << static_cast<std::size_t>(call.alt_proper_pair_depth);
      }
      out << ':' << std::min(static_cast<uint16_t>(99), binned_pl(static_cast<unsigned>(gq)));
      out << ':' << binned_pl(call.phred[0]);
      for (std::size_t p = 1; p < call.phred.size(); ++p)
        out << ',' << binned_pl(call.phred[p]);
    }
  }
  out << '\n';
}

// the records of every variant site of the genotyper's graph (Vcf::add_haplotype per haplotype, generate_infos, the region
// filter of write_records), after a column line
inline void write_column_line(std::ostream & out, WriteOptions const & o)
{
  out << "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO";
  if (!o.sample_names.empty())
  {
    out << "\tFORMAT";
    for (auto const & s : o.sample_names)
      out << '\t' << s;
  }
  out << '\n';
}

// Vcf::add_haplotype (vcf.cpp:1507-1611) for every haplotype of the genotyper: one Variant per variant site
inline std::vector<Variant> haplotype_variants(Genotyper const & g, WriteOptions const & o)
{
  std::vector<Variant> out;
  auto const calls = g.sample_calls();
  for (std::size_t h = 0; h < g.writer.haplotypes.size(); ++h)
  {
    Haplotype const & hap = g.writer.haplotypes[h];
    Variant var;
    var.abs_pos = hap.id; // Variant::Variant(Genotype): variant.cpp:136-141
    for (uint16_t a = 0; a < hap.num; ++a) // Graph::get_all_sequences_of_a_genotype: graph.cpp:818-840
      var.seqs.push_back(g.graph.var_nodes[hap.first_variant_node + a].label.dna);
    var.stats.per_allele.resize(hap.num);
    var.stats.read_strand = hap.read_strand;
    var.stats.clipped_reads = hap.clipped_reads;
    var.stats.mapq_squared = hap.mapq_squared;
    for (uint16_t a = 0; a < hap.num; ++a)
    {
      var.stats.per_allele[a].clipped_bp = hap.per_allele[a].clipped_bp;
      var.stats.per_allele[a].mapq_squared = hap.per_allele[a].mapq_squared;
      var.stats.per_allele[a].score_diff = hap.per_allele[a].score_diff;
      var.stats.per_allele[a].mismatches = hap.per_allele[a].mismatches;
    }
    for (std::size_t s = 0; s < hap.hap_samples.size(); ++s)
    {
      SampleCall sc;
      sc.phred = calls[h][s].phred;
      sc.coverage = hap.hap_samples[s].gt_coverage;
      sc.ref_total_depth = calls[h][s].ref_total_depth;
      sc.alt_total_depth = calls[h][s].alt_total_depth;
      sc.ambiguous_depth = hap.hap_samples[s].ambiguous_depth;
      sc.alt_proper_pair_depth = hap.hap_samples[s].alt_proper_pair_depth;
      var.calls.push_back(std::move(sc));
    }
    var.suffix_id = o.variant_suffix_id;
    out.push_back(std::move(var));
  }
  return out;
}

// vcf_merge_and_filter (src/typer/vcf_operations.cpp:278-478) over the result of one calling pass: the sites-only records an
// iteration hands to the next graph construction.  What the reference reads back from its pools are the variants of
// parallel_reader_genotype_only's Vcf when no calls file is written (hts_parallel_reader.cpp:939-962): add_haplotype for every
// haplotype (hap_id = its index), scan_calls, calls cleared.  Further pools only add their statistics (:366-374); the Genotyper
// here holds all samples, which is the same sums.  `ph`: the phasing flags of call() (caller.cpp:439-479).
inline std::string sites(Genotyper const & g, std::string const & contig, std::map<std::pair<uint16_t, uint16_t>, std::map<std::pair<uint16_t, uint16_t>, int8_t>> const & ph)
{
  constexpr int8_t IS_ANY_HAP_SUPPORT = 1, IS_ANY_ANTI_HAP_SUPPORT = 2; // include/graphtyper/constants.hpp.in:56-57
  WriteOptions pool_options;
  std::vector<Variant> variants = haplotype_variants(g, pool_options);
  for (std::size_t h = 0; h < variants.size(); ++h)
  {
    variants[h].hap_id = static_cast<int32_t>(h); // hts_parallel_reader.cpp:952
    variants[h].suffix_id.clear();
    variants[h].scan_calls(); // :957-961
    variants[h].calls.clear();
  }
  WriteOptions o; // vcf.sample_names.clear(): only variant sites (:337)
  o.contig = contig;
  std::ostringstream out;
  write_column_line(out, o); // write_header(true): the column line ends at INFO
  long var_id = 0;
  std::unordered_map<int32_t, long> hap_id2var_id; // :311-320
  for (Variant const & var : variants)
  {
    hap_id2var_id.emplace(var.hap_id, var_id);
    var_id += static_cast<long>(var.seqs.size()) - 1l;
  }
  var_id = 0;
  for (Variant & var : variants) // :340-470
  {
    std::vector<int8_t> const is_good_alt = var.generate_infos(g.graph.is_sv_graph);
    for (long a = 0; a < static_cast<long>(var.seqs.size()) - 1l; ++a)
    {
      ++var_id;
      if (is_good_alt[static_cast<std::size_t>(a)] == 0)
        continue;
      Variant new_var;
      new_var.abs_pos = var.abs_pos;
      new_var.seqs.push_back(var.seqs[0]);
      new_var.seqs.push_back(var.seqs[static_cast<std::size_t>(a) + 1]);
      new_var.infos["GT_ID"] = std::to_string(var_id);
      std::ostringstream ss_anti, ss_hap;
      bool is_anti_empty = true, is_hap_empty = true;
      for (long a2 = a + 1; a2 < static_cast<long>(var.seqs.size()) - 1l; ++a2) // other alleles of this variant are anti alleles
      {
        if (is_good_alt[static_cast<std::size_t>(a2)] == 0)
          continue;
        if (!is_anti_empty)
          ss_anti << ",";
        ss_anti << (var_id + a2 - a);
        is_anti_empty = false;
      }
      auto find_it = ph.find(std::make_pair(static_cast<uint16_t>(var.hap_id), static_cast<uint16_t>(a + 1)));
      if (find_it != ph.end())
        for (auto const & other : find_it->second)
        {
          int32_t const other_hap_id = other.first.first, other_allele = other.first.second;
          if (other_allele == 0)
            continue;
          int8_t const flags = other.second;
          if (flags != IS_ANY_HAP_SUPPORT && flags != IS_ANY_ANTI_HAP_SUPPORT)
            continue;
          long const var_id_other = hap_id2var_id.at(other_hap_id) + other_allele;
          if (flags == IS_ANY_HAP_SUPPORT)
          {
            if (!is_hap_empty)
              ss_hap << ",";
            ss_hap << var_id_other;
            is_hap_empty = false;
          }
          else
          {
            if (!is_anti_empty)
              ss_anti << ",";
            ss_anti << var_id_other;
            is_anti_empty = false;
          }
        }
      if (!is_anti_empty)
        new_var.infos["GT_ANTI_HAPLOTYPE"] = ss_anti.str();
      if (!is_hap_empty)
        new_var.infos["GT_HAPLOTYPE"] = ss_hap.str();
      write_record(out, new_var, o, g.graph.is_sv_graph); // ("", no FILTER_ZERO_QUAL, genotypes dropped)
    }
  }
  return out.str();
}

inline std::string records(Genotyper const & g, WriteOptions const & o)
{
  std::ostringstream out;
  write_column_line(out, o);
  for (Variant & var : haplotype_variants(g, o))
  {
    var.generate_infos(g.graph.is_sv_graph);
    if (var.abs_pos < o.region_begin || var.abs_pos > o.region_end)
      continue;
    write_record(out, var, o, g.graph.is_sv_graph);
  }
  return out.str();
}
} // namespace vcf
} // namespace gto
