// gto_discovery.hpp -- CPU oracle (test infrastructure only, see gto.hpp) for the first slice of variant discovery:
// the per-sample first pass over the reads of a region, run_first_pass (src/typer/caller.cpp:488-1186) up to and including its
// two support filters -- SNP / indel events read off the CIGARs with their EventSupport, the phase counts between the events
// of a read, the coverage difference arrays, has_good_support for SNPs, the good-support / realignment-support classes of
// indels; then the end of the pass (:1186-1365: which events of a sample travel together -- "ever" and "always" -- from the phase
// counts and the coverage), merge_haplotypes2 (:64-165) and the union of the files' indels (streamlined_discovery, :2853-2903).
// What follows in the reference (realignment through paw::pairwise_alignment -- a dependency absent from the reference tree --,
// the second pass) is not restated.
// PARITY UNPINNED: the reference holds no test or vector for discovery (test/typer has none); this follows the text line by line.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <set>
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
  int32_t max_log_qual_file_i = 0; // (event.hpp: which file had the best support of an indel)

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
  // what the product's device kernel hands its host stage, made here so that the host stage can be tested without a device
  // (include/gtx.h: gtx_disc_event, gtx_disc_read_out)
  struct RawEvent
  {
    uint32_t read, pos, seq;
    uint16_t len;
    uint8_t type, hq;
    uint16_t max_distance, reserved;
  };
  struct ReadOut
  {
    uint32_t first_event, n_events;
    int32_t pos_end;
    uint32_t state; // 0 skipped, 1 counted, 2 the pass ends here
  };
  std::vector<RawEvent> raw_events;
  std::vector<ReadOut> read_outs;
  long file_i = 0;
  struct Thap // HaplotypeInfo (caller.cpp:45-52); ordered sets: only membership matters
  {
    std::set<Event> ever_together, always_together;
  };
  std::map<Event, Thap> sample_haplotypes;

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
    raw_events.clear();
    read_outs.assign(reads.size(), ReadOut{0, 0, 0, 0});
    for (Read const & r : reads)
    {
      size_t const read_index = static_cast<size_t>(&r - reads.data());
      if (r.cigar.empty() || r.pos < region_begin) // :517-526
        continue;
      long read_offset = 0, ref_offset = static_cast<long>(r.pos) - region_begin;
      long const bucket_index = ref_offset / BUCKET_SIZE;
      if (bucket_index >= static_cast<long>(buckets.size()))
        buckets.resize(bucket_index + 1);
      if (ref_offset >= REF_SIZE)
      {
        read_outs[read_index].state = 2;
        break; // :551-561
      }
      read_outs[read_index].state = 1;
      read_outs[read_index].first_event = static_cast<uint32_t>(raw_events.size());
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
            raw_events.push_back(RawEvent{static_cast<uint32_t>(read_index), static_cast<uint32_t>(ref_pos + region_begin), static_cast<uint32_t>(static_cast<unsigned char>(rb)), 1,
                                          'X', static_cast<uint8_t>(r.qual[read_pos] >= 25), static_cast<uint16_t>(max_distance), 0});
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
            raw_events.push_back(RawEvent{static_cast<uint32_t>(read_index), static_cast<uint32_t>(region_begin + ref_offset), static_cast<uint32_t>(b),
                                          static_cast<uint16_t>(e_ - b), 'I', 1, 0, 0});
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
            raw_events.push_back(RawEvent{static_cast<uint32_t>(read_index), static_cast<uint32_t>(region_begin + ref_offset), static_cast<uint32_t>(ref_offset),
                                          static_cast<uint16_t>(cigar_count), 'D', 1, 0, 0});
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
      read_outs[read_index].n_events = static_cast<uint32_t>(raw_events.size()) - read_outs[read_index].first_event;
      read_outs[read_index].pos_end = static_cast<int32_t>(pos_end - region_begin);
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
          info.max_log_qual_file_i = static_cast<int32_t>(file_i);
          ++it;
        }
        else if (count >= 3.0 && log_qual > 0 && info.proper_pairs >= 1 && (info.hq_count >= 5 || info.max_mapq >= 25) && info.max_mapq >= 10 &&
                 info.clipped < info.hq_count)
        {
          info.has_realignment_support = true;
          info.max_log_qual = log_qual;
          info.max_log_qual_file_i = static_cast<int32_t>(file_i);
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

  // The end of run_first_pass (caller.cpp:1186-1365): for every event left, which later events within two buckets it is seen
  // with ("ever": in enough of the reads that cover both; "always": those at most 10 positions on), judged from its phase counts
  // and the coverage between the two; the SNPs then leave the buckets (they live on in the haplotype map).
  void run_haplotypes(std::string const & reference, long region_begin, long BUCKET_SIZE)
  {
    long const REF_SIZE = reference.size(), NUM_BUCKETS = buckets.size();
    uint16_t const IS_ANY_HAP_SUPPORT = 1, IS_ANY_ANTI_HAP_SUPPORT = 2; // constants.hpp.in:56-57
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
    sample_haplotypes.clear();
    long depth = 0;
    for (long b = 0; b < NUM_BUCKETS; ++b)
    {
      Events & bucket = buckets[b];
      for (auto event_it = bucket.begin(); event_it != bucket.end();)
      {
        Event const & event = event_it->first;
        EventSupport const & info = event_it->second;
        long const begin = std::max(0l, static_cast<long>(event.pos) - region_begin);
        long cov = depth;
        Thap & hap = sample_haplotypes.insert({event, Thap()}).first->second;
        update_coverage(cov, begin, b);
        double support_ratio = static_cast<double>(info.get_raw_support()) / static_cast<double>(cov);
        if (support_ratio < 0.3)
          support_ratio = 0.3;
        auto is_good_support = [&](long local_cov, long local_offset, Event const & other) -> uint16_t // :1216-1268
        {
          auto const find_it = info.phase.find(other);
          bool const is_indel = event.type != 'X' || other.type != 'X';
          if (is_indel)
            return (find_it == info.phase.end() || find_it->second == 0) ? IS_ANY_ANTI_HAP_SUPPORT : (IS_ANY_HAP_SUPPORT | IS_ANY_ANTI_HAP_SUPPORT);
          long const end = std::max(0l, static_cast<long>(other.pos) - region_begin);
          while (local_offset <= end)
          {
            local_cov -= local_offset < REF_SIZE ? static_cast<long>(cov_down[local_offset]) : 0l; // (behind the region: nothing ends there)
            ++local_offset;
          }
          if (local_cov <= 2)
            return 0;
          double const support = find_it == info.phase.end() ? 0.0 : find_it->second;
          if ((support / static_cast<double>(local_cov) / support_ratio) < 0.22)
            return IS_ANY_ANTI_HAP_SUPPORT;
          if ((support / static_cast<double>(local_cov) / support_ratio) > 0.78)
            return IS_ANY_HAP_SUPPORT;
          return IS_ANY_ANTI_HAP_SUPPORT | IS_ANY_HAP_SUPPORT;
        };
        for (auto it2 = std::next(event_it); it2 != bucket.end(); ++it2) // this bucket (:1271-1294)
        {
          Event const & other = it2->first;
          if (other.pos == event.pos && other.type == event.type)
            continue;
          if ((is_good_support(cov, begin + 1, other) & IS_ANY_HAP_SUPPORT) != 0)
          {
            hap.ever_together.insert(other);
            if (other.pos <= event.pos + 10)
              hap.always_together.insert(other);
          }
        }
        if (b + 1 < NUM_BUCKETS) // the next one (:1297-1318)
          for (auto const & kv : buckets[b + 1])
            if ((is_good_support(cov, begin + 1, kv.first) & IS_ANY_HAP_SUPPORT) != 0)
            {
              hap.ever_together.insert(kv.first);
              if (kv.first.pos <= event.pos + 10)
                hap.always_together.insert(kv.first);
            }
        if (b + 2 < NUM_BUCKETS) // and the one behind it, up to two bucket sizes away (:1321-1342)
          for (auto const & kv : buckets[b + 2])
          {
            if (kv.first.pos >= event.pos + 2 * BUCKET_SIZE)
              break;
            if ((is_good_support(cov, begin + 1, kv.first) & IS_ANY_HAP_SUPPORT) != 0)
              hap.ever_together.insert(kv.first);
          }
        if (event_it->first.type == 'X') // :1345-1348
          event_it = bucket.erase(event_it);
        else
          ++event_it;
      }
      if (b * BUCKET_SIZE >= REF_SIZE)
        break;
      for (long offset = b * BUCKET_SIZE, end = std::min(REF_SIZE, (b + 1) * BUCKET_SIZE); offset < end; ++offset)
        depth += static_cast<long>(cov_up[offset]) - static_cast<long>(cov_down[offset]);
    }
  }

  // what a file leaves behind: its indels (Tindel_events) and its haplotype map
  struct Result
  {
    Events indels;
    std::map<Event, Thap> haplotypes;
  };
  Result result() const
  {
    Result r;
    for (auto const & bucket : buckets)
      for (auto const & kv : bucket)
        r.indels.insert(kv);
    r.haplotypes = sample_haplotypes;
    return r;
  }
  // merge_haplotypes2 (caller.cpp:64-165) and the union of the indels (streamlined_discovery, :2853-2903)
  static void merge(Result & into, Result & from)
  {
    if (into.haplotypes.empty())
      into.haplotypes = std::move(from.haplotypes);
    else
      for (auto & kv : from.haplotypes)
      {
        auto ins = into.haplotypes.insert(kv);
        Thap & mine = ins.first->second;
        if (ins.second)
        {
          for (auto it = mine.always_together.begin(); it != mine.always_together.end();)
            it = into.haplotypes.count(*it) > 0 ? mine.always_together.erase(it) : std::next(it);
        }
        else
        {
          mine.ever_together.insert(kv.second.ever_together.begin(), kv.second.ever_together.end());
          std::set<Event> both;
          for (Event const & e : kv.second.always_together)
            if (mine.always_together.count(e) > 0)
              both.insert(e);
          mine.always_together = std::move(both);
        }
      }
    from.haplotypes.clear();
    for (auto & kv : from.indels)
    {
      auto ins = into.indels.insert(kv);
      if (!ins.second)
      {
        EventSupport & old_info = ins.first->second;
        old_info.has_indel_good_support |= kv.second.has_indel_good_support;
        if (kv.second.max_log_qual > old_info.max_log_qual)
        {
          old_info.max_log_qual = kv.second.max_log_qual;
          old_info.max_log_qual_file_i = kv.second.max_log_qual_file_i;
        }
      }
    }
    from.indels.clear();
  }
  static std::vector<uint32_t> dump(Result const & r)
  {
    std::vector<uint32_t> s;
    s.push_back(static_cast<uint32_t>(r.indels.size()));
    for (auto const & kv : r.indels)
      put_support(s, kv.first, kv.second, true);
    s.push_back(static_cast<uint32_t>(r.haplotypes.size()));
    for (auto const & kv : r.haplotypes)
    {
      put_event(s, kv.first);
      for (std::set<Event> const * set : {&kv.second.ever_together, &kv.second.always_together})
      {
        s.push_back(static_cast<uint32_t>(set->size()));
        for (Event const & e : *set)
          put_event(s, e);
      }
    }
    return s;
  }
  static void put_support(std::vector<uint32_t> & s, Event const & e, EventSupport const & i, bool with_file)
  {
    put_event(s, e);
    for (uint32_t v : {uint32_t(i.hq_count), uint32_t(i.lq_count), uint32_t(i.proper_pairs), uint32_t(i.first_in_pairs), uint32_t(i.sequence_reversed),
                       uint32_t(i.clipped), uint32_t(i.max_mapq), uint32_t(i.max_distance), uint32_t(i.uniq_pos1), uint32_t(i.uniq_pos2),
                       uint32_t(i.uniq_pos3), uint32_t(i.span), uint32_t(i.has_realignment_support), uint32_t(i.has_indel_good_support),
                       i.max_log_qual})
      s.push_back(v);
    if (with_file)
      s.push_back(static_cast<uint32_t>(i.max_log_qual_file_i));
    s.push_back(static_cast<uint32_t>(i.phase.size()));
    for (auto const & ph : i.phase)
    {
      put_event(s, ph.first);
      s.push_back(ph.second);
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
