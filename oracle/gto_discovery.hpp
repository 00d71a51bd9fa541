// gto_discovery.hpp -- CPU oracle (test infrastructure only, see gto.hpp) for the first slice of variant discovery:
// the per-sample first pass over the reads of a region, run_first_pass (src/typer/caller.cpp:488-1186) up to and including its
// two support filters -- SNP / indel events read off the CIGARs with their EventSupport, the phase counts between the events
// of a read, the coverage difference arrays, has_good_support for SNPs, the good-support / realignment-support classes of
// indels.  What follows in the reference (haplotypes of the surviving events, realignment through paw::pairwise_alignment
// -- a dependency absent from the reference tree --, the second pass) is not restated.
// PARITY UNPINNED: the reference holds no test or vector for discovery (test/typer has none); this follows the text line by line.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <string>
#include <vector>

namespace gto
{
namespace disc
{
// include/graphtyper/typer/event.hpp:30-73, src/typer/event.cpp:185-207
struct Event
{
  uint32_t pos = 0;
  char type = 0;
  std::vector<char> sequence;
  bool operator<(Event const & b) const
  {
    int const order_a = (type == 'D') + 2 * (type == 'X'), order_b = (b.type == 'D') + 2 * (b.type == 'X');
    return pos < b.pos || (pos == b.pos && order_a < order_b) || (pos == b.pos && order_a == order_b && sequence < b.sequence);
  }
};

// event.hpp:75-113
struct EventSupport
{
  uint16_t hq_count = 0, lq_count = 0, proper_pairs = 0, first_in_pairs = 0, sequence_reversed = 0, clipped = 0;
  uint8_t max_mapq = 0, max_distance = 0;
  int32_t uniq_pos1 = -1, uniq_pos2 = -1, uniq_pos3 = -1;
  std::map<Event, uint16_t> phase;
  uint16_t multi_count = 0, anti_count = 0, span = 1;
  bool has_realignment_support = false, has_indel_good_support = false;
  uint32_t max_log_qual = 0;

  int get_raw_support() const { return hq_count + lq_count; }
  double corrected_support() const { return static_cast<double>(hq_count) + static_cast<double>(lq_count) / 2.0; }
  // event.cpp:226-256 with the defaults of Options (options.hpp:46-49: filter_on_proper_pairs, filter_on_read_bias,
  // filter_on_strand_bias = true, no_filter_on_begin_pos = false)
  bool has_good_support(long cov) const
  {
    if (cov < 1)
      cov = 1;
    int const raw_support = get_raw_support();
    double const ratio = static_cast<double>(raw_support) / static_cast<double>(cov);
    bool const is_very_promising = uniq_pos3 != -1 && ((hq_count >= 8 && ratio >= 0.35) || (hq_count >= 7 && ratio >= 0.40)) && proper_pairs >= 6;
    bool const is_promising =
      uniq_pos3 != -1 && ((hq_count >= 7 && ratio >= 0.20) || (hq_count >= 6 && ratio >= 0.30) || (hq_count >= 5 && ratio >= 0.40)) && proper_pairs >= 4;
    return uniq_pos2 != -1 && proper_pairs >= 2 && hq_count >= 3 && (is_promising || (first_in_pairs > 0 && first_in_pairs < raw_support)) &&
           (is_very_promising || (is_promising && sequence_reversed > 0 && sequence_reversed < raw_support) ||
            (sequence_reversed > 1 && sequence_reversed < (raw_support - 1))) &&
           (clipped <= 1 || (clipped + 5) <= raw_support) && (max_distance >= 10 || (is_promising && hq_count >= 10)) &&
           corrected_support() >= 3.9 && (ratio > 0.26 || is_promising);
  }
};

inline uint32_t get_log_qual_double(double count, double anti_count, double eps) // event.cpp:102-113
{
  double const gt00 = count * eps, gt01 = count + anti_count, gt11 = anti_count * eps, gt_alt = std::min(gt01, gt11);
  return gt00 > gt_alt ? static_cast<uint32_t>(gt00 - gt_alt + 0.5) : 0u;
}

struct Read // the fields of bam1_t the pass reads
{
  int32_t pos;
  uint16_t flag;
  uint8_t mapq;
  std::vector<uint32_t> cigar; // raw BAM words
  std::string sequence;        // seq_nt16_str of the packed bases
  std::vector<uint8_t> qual;
};

struct FirstPass
{
  static constexpr uint16_t IS_PROPER_PAIR = 2, IS_SEQ_REVERSED = 16, IS_FIRST_IN_PAIR = 64;
  using Events = std::map<Event, EventSupport>;
  std::vector<Events> buckets;
  std::vector<uint32_t> cov_up, cov_down;

  static bool is_clipped(Read const & r) // caller.cpp:167-196
  {
    if (r.cigar.empty())
      return false;
    if ((r.cigar.front() & 15) == 4 && (r.cigar.front() >> 4) >= 1)
      return true;
    return (r.cigar.back() & 15) == 4 && (r.cigar.back() >> 4) >= 1;
  }

  Events::iterator add_snp(Event && e, long region_begin, long BUCKET_SIZE) // bucket.cpp:164-182
  {
    long const b = (e.pos - region_begin) / BUCKET_SIZE;
    if (b >= static_cast<long>(buckets.size()))
      buckets.resize(b + 1);
    return buckets[b].insert({std::move(e), EventSupport()}).first;
  }

  Events::iterator add_indel(Event && e, long region_begin, long BUCKET_SIZE, std::string const & reference, long ref_offset) // bucket.cpp:75-162
  {
    long const REF_SIZE = reference.size();
    long const b = (e.pos - region_begin) / BUCKET_SIZE;
    if (b >= static_cast<long>(buckets.size()))
      buckets.resize(b + 1);
    auto it_pair = buckets[b].insert({std::move(e), EventSupport()});
    if (it_pair.second)
    {
      Event const & ne = it_pair.first->first;
      long span = 0;
      long const count = static_cast<long>(ne.sequence.size());
      if (ne.type == 'I')
      {
        while (span < count)
        {
          if ((ref_offset + span) >= REF_SIZE || ne.sequence[span] != reference[ref_offset + span])
            break;
          ++span;
        }
        if (span == count)
          while ((ref_offset + span) < REF_SIZE)
          {
            if (reference[ref_offset + span - count] != reference[ref_offset + span])
              break;
            ++span;
          }
      }
      else
        while ((ref_offset + span) < REF_SIZE)
        {
          // (the reference reads one position past the end here when the repeat runs to the end of the sequence; a '\0' there
          //  ends the loop in this restatement)
          char const behind = ref_offset + span + count < REF_SIZE ? reference[ref_offset + span + count] : '\0';
          if (reference[ref_offset + span] != behind)
            break;
          ++span;
        }
      if ((span + 1) >= std::numeric_limits<uint16_t>::max())
        span = std::numeric_limits<uint16_t>::max() - 1;
      it_pair.first->second.span = static_cast<uint16_t>(span + 1);
    }
    return it_pair.first;
  }

  // caller.cpp:488-1186
  void run(std::vector<Read> const & reads, std::string const & reference, long region_begin, long BUCKET_SIZE)
  {
    long const REF_SIZE = reference.size();
    cov_up.assign(REF_SIZE, 0);
    cov_down.assign(REF_SIZE, 0);
    buckets.clear();
    static char const CIGAR_MAP[] = "MIDNSHP=XB******";
    for (Read const & r : reads)
    {
      if (r.cigar.empty() || r.pos < region_begin) // :517-526
        continue;
      long read_offset = 0, ref_offset = static_cast<long>(r.pos) - region_begin;
      long const bucket_index = ref_offset / BUCKET_SIZE;
      if (bucket_index >= static_cast<long>(buckets.size()))
        buckets.resize(bucket_index + 1);
      if (ref_offset >= REF_SIZE)
        break; // :551-561
      std::vector<Events::iterator> cigar_events;
      bool const is_read_clipped = is_clipped(r);
      long const l_qseq = static_cast<long>(r.sequence.size());
      auto common = [&](EventSupport & s)
      {
        if (r.mapq != 255 && r.mapq > s.max_mapq)
          s.max_mapq = r.mapq;
        s.proper_pairs += ((r.flag & IS_PROPER_PAIR) != 0);
        s.sequence_reversed += ((r.flag & IS_SEQ_REVERSED) != 0);
        s.clipped += is_read_clipped;
      };
      for (uint32_t const word : r.cigar)
      {
        long const cigar_count = word >> 4;
        char const op = CIGAR_MAP[word & 15];
        if (ref_offset >= REF_SIZE)
          break;
        switch (op)
        {
        case 'M':
        case '=':
        case 'X':
          for (long k = 0; k < cigar_count; ++k)
          {
            long const ref_pos = ref_offset + k;
            if (ref_pos >= REF_SIZE)
              break;
            char const ref = reference[ref_pos];
            long const read_pos = read_offset + k;
            if (read_pos >= l_qseq)
              break;
            char const rb = r.sequence[read_pos];
            if (rb == ref || (ref != 'A' && ref != 'C' && ref != 'G' && ref != 'T') || (rb != 'A' && rb != 'C' && rb != 'G' && rb != 'T'))
              continue;
            Event e;
            e.pos = static_cast<uint32_t>(ref_pos + region_begin);
            e.type = 'X';
            e.sequence = {rb};
            auto it = add_snp(std::move(e), region_begin, BUCKET_SIZE);
            EventSupport & s = it->second;
            if (r.qual[read_pos] >= 25)
              ++s.hq_count;
            else
              ++s.lq_count;
            common(s);
            s.first_in_pairs += ((r.flag & IS_FIRST_IN_PAIR) != 0); // (SNPs only, :656)
            if (s.uniq_pos1 == -1)
              s.uniq_pos1 = r.pos;
            else if (s.uniq_pos2 == -1)
            {
              if (s.uniq_pos1 != r.pos)
                s.uniq_pos2 = r.pos;
            }
            else if (s.uniq_pos3 == -1 && s.uniq_pos2 != r.pos)
              s.uniq_pos3 = r.pos;
            long const max_distance = std::min(read_pos, l_qseq - 1 - read_pos);
            if (max_distance > s.max_distance)
              s.max_distance = static_cast<uint8_t>(max_distance);
            cigar_events.push_back(it);
          }
          read_offset += cigar_count;
          ref_offset += cigar_count;
          break;
        case 'I':
        {
          long const b = std::min(read_offset, l_qseq), e_ = std::min(read_offset + cigar_count, l_qseq);
          if (b == e_)
            break; // (:698-699: leaves read_offset as it is)
          std::string const ins = r.sequence.substr(b, e_ - b);
          if (std::all_of(ins.begin(), ins.end(), [](char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }))
          {
            Event e;
            e.pos = static_cast<uint32_t>(region_begin + ref_offset);
            e.type = 'I';
            e.sequence.assign(ins.begin(), ins.end());
            auto it = add_indel(std::move(e), region_begin, BUCKET_SIZE, reference, ref_offset);
            ++it->second.hq_count;
            common(it->second);
            cigar_events.push_back(it);
          }
          read_offset += cigar_count;
          break;
        }
        case 'D':
        {
          if (ref_offset + cigar_count >= REF_SIZE)
          {
            ref_offset += cigar_count;
            break;
          }
          Event e;
          e.pos = static_cast<uint32_t>(region_begin + ref_offset);
          e.type = 'D';
          e.sequence.assign(reference.begin() + ref_offset, reference.begin() + ref_offset + cigar_count);
          if (std::all_of(e.sequence.begin(), e.sequence.end(), [](char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }))
          {
            auto it = add_indel(std::move(e), region_begin, BUCKET_SIZE, reference, ref_offset);
            ++it->second.hq_count;
            common(it->second);
            cigar_events.push_back(it);
          }
          ref_offset += cigar_count;
          break;
        }
        case 'S':
          read_offset += cigar_count;
          break;
        default:
          break;
        }
      }
      int constexpr HIGH_EVENT_COUNT = 12, VHIGH_EVENT_COUNT = 18; // :777-806
      if (static_cast<int>(cigar_events.size()) >= HIGH_EVENT_COUNT)
        for (auto & ev : cigar_events)
        {
          EventSupport & info = ev->second;
          if (static_cast<int>(cigar_events.size()) >= VHIGH_EVENT_COUNT)
          {
            if (info.hq_count > 0)
              --info.hq_count;
            else if (info.lq_count > 0)
              --info.lq_count;
          }
          else if (info.hq_count > 0)
          {
            --info.hq_count;
            ++info.lq_count;
          }
        }
      if (static_cast<int>(cigar_events.size()) < VHIGH_EVENT_COUNT) // :808-822
        for (long e2 = 1; e2 < static_cast<long>(cigar_events.size()); ++e2)
          for (long prev = 0; prev < e2; ++prev)
            ++cigar_events[prev]->second.phase.insert({cigar_events[e2]->first, 0}).first->second;
      long const pos_end = region_begin + std::min(ref_offset, REF_SIZE - 1); // :824-834
      ++cov_up[r.pos - region_begin];
      ++cov_down[pos_end - region_begin];
    }
    if ((static_cast<long>(buckets.size()) - 1l) * BUCKET_SIZE >= REF_SIZE) // :868-874
      buckets.resize(((REF_SIZE - 1) / BUCKET_SIZE) + 1);
    long const NUM_BUCKETS = buckets.size();
    auto update_coverage = [&](long & cov, long const pos, long const b)
    {
      long offset = pos + 1;
      if (offset > b * BUCKET_SIZE)
      {
        offset = b * BUCKET_SIZE;
        while (offset <= pos)
        {
          cov += static_cast<long>(cov_up[offset]) - static_cast<long>(cov_down[offset]);
          ++offset;
        }
      }
    };
    {
      long depth = 0; // SNPs with low support (:897-985)
      for (long b = 0; b < NUM_BUCKETS; ++b)
      {
        for (auto it = buckets[b].begin(); it != buckets[b].end();)
        {
          if (it->first.type != 'X')
          {
            ++it;
            continue;
          }
          long cov = depth;
          update_coverage(cov, std::max(0l, static_cast<long>(it->first.pos) - region_begin), b);
          if (it->second.has_good_support(cov))
            ++it;
          else
            it = buckets[b].erase(it);
        }
        if (b * BUCKET_SIZE >= REF_SIZE)
          break;
        for (long offset = b * BUCKET_SIZE, end = std::min(REF_SIZE, (b + 1) * BUCKET_SIZE); offset < end; ++offset)
          depth += static_cast<long>(cov_up[offset]) - static_cast<long>(cov_down[offset]);
      }
    }
    long depth = 0; // indels: good support, realignment support, or gone (:990-1186)
    for (long b = 0; b < NUM_BUCKETS; ++b)
    {
      for (auto it = buckets[b].begin(); it != buckets[b].end();)
      {
        Event const & indel = it->first;
        if (indel.type == 'X')
        {
          ++it;
          continue;
        }
        EventSupport & info = it->second;
        long const naive_pad = static_cast<long>(4.0 + static_cast<double>(indel.sequence.size()) / 3.0);
        long const naive_begin = std::max(0l, static_cast<long>(indel.pos) - naive_pad - region_begin);
        long const naive_end = std::min(REF_SIZE, static_cast<long>(indel.pos) + info.span + naive_pad - region_begin);
        double const correction = indel.type == 'I' ? static_cast<double>(indel.sequence.size() / 2.0 + 8.0) / 8.0
                                                    : static_cast<double>(indel.sequence.size() / 3.0 + 10.0) / 10.0;
        double const count = correction * (info.hq_count + info.lq_count);
        long cov = depth, offset = naive_begin;
        if (offset <= b * BUCKET_SIZE)
          while (offset < b * BUCKET_SIZE)
          {
            cov -= static_cast<long>(cov_up[offset]) - static_cast<long>(cov_down[offset]);
            ++offset;
          }
        else
        {
          offset = b * BUCKET_SIZE;
          while (offset < naive_begin)
          {
            cov += static_cast<long>(cov_up[offset]) - static_cast<long>(cov_down[offset]);
            ++offset;
          }
        }
        while (offset <= naive_end)
        {
          cov -= offset < REF_SIZE ? static_cast<long>(cov_down[offset]) : 0l; // (the reference reads cov_down[REF_SIZE] when the interval ends at the region's end)
          ++offset;
        }
        double const corrected_cov = std::max(static_cast<double>(cov), count), anti_count_d = corrected_cov - count;
        uint32_t const log_qual = get_log_qual_double(count, anti_count_d, 10.0);
        if (info.hq_count >= 6 && count >= 8.0 && log_qual >= 60 && info.sequence_reversed > 0 && info.sequence_reversed < info.hq_count &&
            info.proper_pairs >= 3 && info.max_mapq >= 20 && (info.clipped == 0 || (info.clipped + 3) <= info.hq_count))
        {
          info.has_indel_good_support = true;
          info.has_realignment_support = true;
          info.max_log_qual = log_qual;
          ++it;
        }
        else if (count >= 3.0 && log_qual > 0 && info.proper_pairs >= 1 && (info.hq_count >= 5 || info.max_mapq >= 25) && info.max_mapq >= 10 &&
                 info.clipped < info.hq_count)
        {
          info.has_realignment_support = true;
          info.max_log_qual = log_qual;
          ++it;
        }
        else
          it = buckets[b].erase(it);
      }
      if (b * BUCKET_SIZE >= REF_SIZE)
        break;
      for (long offset = b * BUCKET_SIZE, end = std::min(REF_SIZE, (b + 1) * BUCKET_SIZE); offset < end; ++offset)
        depth += static_cast<long>(cov_up[offset]) - static_cast<long>(cov_down[offset]);
    }
  }

  // canonical word stream of what is left: per event pos, type, length, its characters, the support fields, the phase entries
  static void put_event(std::vector<uint32_t> & s, Event const & e)
  {
    s.push_back(e.pos);
    s.push_back(static_cast<uint32_t>(e.type));
    s.push_back(static_cast<uint32_t>(e.sequence.size()));
    for (char c : e.sequence)
      s.push_back(static_cast<uint32_t>(static_cast<unsigned char>(c)));
  }
  std::vector<uint32_t> dump() const
  {
    std::vector<uint32_t> s;
    for (auto const & bucket : buckets)
      for (auto const & kv : bucket)
      {
        put_event(s, kv.first);
        EventSupport const & i = kv.second;
        for (uint32_t v : {uint32_t(i.hq_count), uint32_t(i.lq_count), uint32_t(i.proper_pairs), uint32_t(i.first_in_pairs), uint32_t(i.sequence_reversed),
                           uint32_t(i.clipped), uint32_t(i.max_mapq), uint32_t(i.max_distance), uint32_t(i.uniq_pos1), uint32_t(i.uniq_pos2),
                           uint32_t(i.uniq_pos3), uint32_t(i.span), uint32_t(i.has_realignment_support), uint32_t(i.has_indel_good_support),
                           i.max_log_qual, uint32_t(i.phase.size())})
          s.push_back(v);
        for (auto const & ph : i.phase)
        {
          put_event(s, ph.first);
          s.push_back(ph.second);
        }
      }
    return s;
  }
};
} // namespace disc
} // namespace gto
