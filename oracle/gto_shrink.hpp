// TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's read pre-filter "bamshrink"
// (/root/reference/src/utilities/bamshrink.cpp:64-1045, options include/graphtyper/utilities/bamshrink.hpp:7-27 and
// include/graphtyper/utilities/options.hpp:63-69,90).  Only tests/ may use it; the product is graphtyper_amd/csrc/gtx_shrink.cpp.
//
// Parity unpinned: the reference has no test of bamshrink; its record I/O lives in two absent submodules (seqan fork
// hannespetur/seqan, htslib), so this restatement works on decoded BAM records (SAM spec 4.2) and states the places where the
// reference reads outside a container (undefined there) with the value chosen here:
//   * findNum2Clip (:567-603) indexes the CIGAR at its length when the forward start lies behind the reverse read: "not a D" here;
//   * the write loops (:886-904, :1020-1041) index bin_counts with the bin of a begin position that removeNsAtEnds moved behind the
//     last counted bin: a count of 0 here;
//   * process_tags (:102-308) walks the aux area without looking at its end: a field that runs past it ends the walk here.
// Bases stay BAM 4-bit codes (N = 15; the reference goes through IUPAC letters), qualities stay raw phred bytes (the reference
// adds 33 and compares against 33 + x).  The iteration order of `read_first` (std::unordered_map with the hash of :321-336) is
// taken from the same container with the same hash: equal under the same standard library.
// Names here <- the reference's (bamshrink.cpp): counter_name <- decimal_to_read_name_string :34-61; strip_hard_clips <- removeHardClipped
// :64-76; quals_of_20_and_more <- countHighBaseQuality :78-81; two_level_quals <- binarizeQual :83-87; both_ends_clipped / an_end_clipped
// <- is_clipped_both_ends / is_one_end_clipped :89-99; scores_say_keep <- process_tags :102-308; make_single <- makeUnpaired :345-356;
// matching_bases <- countMatchingBases :358-369; cigar_shorten_back / cigar_shorten_front <- resetCigarStringEnd / resetCigarStringBegin
// :388-482; strip_soft_clips <- removeSoftClipped :484-521; strip_n_ends <- removeNsAtEnds :523-584; clip_and_shift <- findNum2Clip
// :567-604; trim_adapters <- removeAdapters :606-665; shrink_interval <- qualityFilterSlice2 :667-1045 (single_ok / mate_ok <- its
// filter_unpaired / filter_paired, keep_single / finish_mate <- post_process_unpaired / post_process_paired, ready <- read_set, waiting <-
// read_first, bins <- bin_counts, bin_cap <- max_bin_sum, origin <- first_pos, counter <- read_num).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

namespace gto_shrink
{
struct Options // bamshrink.hpp:7-27
{
  int maxFragLen = 1000;
  int minNumMatching = 55;
  bool is_filtering_mapq0 = true;
  bool no_filter_on_coverage = false;
  int minReadLen = 75;
  int minReadLenMapQ0 = 94;
  int minUnpairedReadLen = 94;
  long as_filter_threshold = 40;
  double avgCovByReadLen = 0.30000001;
  long SUPER_HI_DEPTH = 2;
  int sam_flag_filter = 3840;     // options.hpp:90
  bool change_read_names = true;  // CHANGE_READ_NAMES of a release build (:24-28)
};

struct CigarElement
{
  char op;
  uint32_t len;
};

struct Record // seqan::BamAlignmentRecord as far as bamshrink touches it
{
  std::string name;
  uint32_t flag = 0;
  int32_t ref_id = -1, pos = -1;
  uint8_t mapq = 0;
  std::vector<CigarElement> cigar;
  int32_t mate_ref_id = -1, mate_pos = -1, tlen = 0;
  std::vector<uint8_t> seq;  // 4-bit codes, one per base
  std::vector<uint8_t> qual; // raw phred
  std::string aux;          // raw aux bytes
  bool operator<(Record const & b) const { return pos < b.pos; } // :312-316
};

inline bool flag_multiple(Record const & r) { return r.flag & 1u; }
inline bool flag_unmapped(Record const & r) { return r.flag & 4u; }
inline bool flag_next_unmapped(Record const & r) { return r.flag & 8u; }
inline bool flag_rc(Record const & r) { return r.flag & 16u; }
inline bool flag_next_rc(Record const & r) { return r.flag & 32u; }

// :34-61
inline char digit_char(long in)
{
  if (in >= 31)
    ++in;
  return static_cast<char>('!' + in);
}

inline std::string counter_name(long in)
{
  long const DIGITS = 93;
  std::string str;
  while (in >= DIGITS)
  {
    long const rem = in % DIGITS;
    in = in / DIGITS;
    str.push_back(digit_char(rem));
  }
  str.push_back(digit_char(in));
  return str;
}

// :64-76
inline void strip_hard_clips(std::vector<CigarElement> & cigar)
{
  long n_cigar = static_cast<long>(cigar.size());
  if (n_cigar >= 1 && cigar[0].op == 'H')
  {
    cigar.erase(cigar.begin());
    --n_cigar;
  }
  if (n_cigar >= 2 && cigar[n_cigar - 1].op == 'H')
    cigar.pop_back();
}

// :78-87
inline long quals_of_20_and_more(std::vector<uint8_t> const & qual)
{
  return std::count_if(qual.begin(), qual.end(), [](uint8_t q) { return q >= 20; });
}

inline void two_level_quals(std::vector<uint8_t> & qual)
{
  for (auto & q : qual)
    q = q >= 24 ? ('?' - 33) : (',' - 33);
}

// :89-99
inline bool both_ends_clipped(std::vector<CigarElement> const & cigar, long const min_clip = 15)
{
  return cigar.size() >= 1 && cigar.front().op == 'S' && cigar.back().op == 'S' &&
         static_cast<long>(cigar.front().len + cigar.back().len) >= min_clip;
}

inline bool an_end_clipped(std::vector<CigarElement> const & cigar, long const min_clip = 0)
{
  return cigar.size() == 0 || (cigar.front().op == 'S' && static_cast<long>(cigar.front().len) >= min_clip) ||
         (cigar.back().op == 'S' && static_cast<long>(cigar.back().len) >= min_clip);
}

// :102-308; true: the alignment is good.  kept: RG and the AS / XS / WS fields.
inline bool scores_say_keep(Record const & record, std::string & kept, Options const & opts)
{
  size_t i = 0;
  int64_t as = -1, xs = -1, ws = -1;
  std::string const & aux = record.aux;
  size_t const tags_length = aux.size();
  while (i < tags_length)
  {
    size_t const begin_it = i;
    size_t end_it = begin_it;
    i += 3;
    if (i > tags_length)
      break; // (outside the area)
    char const type = aux[i - 1];
    bool tag_as = false, tag_xs = false, tag_ws = false;
    if (aux[i - 2] == 'S')
    {
      if (aux[i - 3] == 'A')
        tag_as = true;
      else if (aux[i - 3] == 'X')
        tag_xs = true;
      else if (aux[i - 3] == 'W')
        tag_ws = true;
    }
    auto set_score = [&](int64_t score)
    {
      if (tag_as)
        as = score;
      else if (tag_xs)
        xs = score;
      else if (tag_ws)
        ws = score;
    };
    bool outside = false;
    auto number = [&](auto zero)
    {
      decltype(zero) num = 0;
      if (i + sizeof(num) > tags_length)
      {
        outside = true;
        return;
      }
      std::memcpy(&num, aux.data() + i, sizeof(num));
      set_score(static_cast<int64_t>(num));
      i += sizeof(num);
      end_it = i;
    };
    switch (type)
    {
    case 'A': ++i; break;
    case 'Z':
    {
      bool const is_rg = aux[i - 3] == 'R' && aux[i - 2] == 'G';
      while (i < tags_length && aux[i] != '\0' && aux[i] != '\n')
        ++i;
      ++i;
      if (i > tags_length)
      {
        outside = true;
        break;
      }
      if (is_rg)
      {
        end_it = i;
        kept.append(aux, begin_it, end_it - begin_it);
      }
      break;
    }
    case 'c': number(int8_t()); break;
    case 'C': number(uint8_t()); break;
    case 's': number(int16_t()); break;
    case 'S': number(uint16_t()); break;
    case 'i': number(int32_t()); break;
    case 'I': number(uint32_t()); break;
    case 'f':
      if (i + 4 > tags_length)
        outside = true;
      else
      {
        i += 4;
        end_it = i;
      }
      break;
    default: i = tags_length; break; // unknown type: the walk stops
    }
    if (outside)
      break;
    if (tag_as || tag_xs || tag_ws)
      kept.append(aux, begin_it, end_it - begin_it);
  }
  if (as != -1 && ws == -1)
    ws = as;
  if (ws != -1 && xs != -1 && (!flag_multiple(record) || flag_next_unmapped(record)))
  {
    if (ws <= xs + 5)
      return false;
    long matches = 0, indels = 0;
    for (auto const & c : record.cigar)
    {
      if (c.op == 'M')
        matches += c.len;
      else if (c.op == 'D' || c.op == 'I')
        indels += c.len + 2;
    }
    if (std::max(ws, as) + opts.as_filter_threshold <= matches - indels)
      return false;
  }
  return true;
}

struct NameHash // :321-336 (the sum is formed in 32 bits: 0x9e3779b9 is an unsigned int)
{
  std::size_t operator()(std::string const & s) const
  {
    std::size_t seed = 42;
    for (char c : s)
      seed ^= c + 0x9e3779b9 + (static_cast<unsigned>(c) << 6) + (c >> 2); // (`c << 6` there: the same bits for a negative char, without the undefined shift)
    return seed;
  }
};

// :345-356
inline void make_single(Record & record)
{
  record.mate_pos = -1;
  record.mate_ref_id = -1;
  record.flag &= ~8u;
  record.flag &= ~2u;
  record.flag &= ~1u;
  record.flag &= ~32u;
}

// :358-369
inline long matching_bases(std::vector<CigarElement> const & cg)
{
  long n = 0;
  for (auto const & c : cg)
    if (c.op == 'M')
      n += c.len;
  return n;
}

// :388-420
inline void cigar_shorten_back(std::vector<CigarElement> & cg, unsigned n_gone)
{
  if (cg.empty())
    return;
  if (cg.back().op == 'D')
  {
    cg.pop_back();
    if (cg.empty())
      return;
  }
  auto & tail = cg.back();
  if (tail.len > n_gone)
    tail.len -= n_gone;
  else if (tail.len == n_gone)
  {
    cg.pop_back();
    if (!cg.empty() && cg.back().op == 'D')
      cg.pop_back();
  }
  else
  {
    unsigned const rest = n_gone - tail.len;
    cg.pop_back();
    cigar_shorten_back(cg, rest);
  }
}

// :423-482: the number of reference bases taken off the front
inline unsigned cigar_shorten_front(std::vector<CigarElement> & cg, unsigned n_gone)
{
  if (cg.empty())
    return 0;
  unsigned ref_gone = 0;
  if (cg[0].op == 'D')
  {
    ref_gone = cg[0].len;
    cg.erase(cg.begin());
    if (cg.empty())
      return ref_gone;
  }
  if (cg[0].len > n_gone)
  {
    cg[0].len -= n_gone;
    if (cg[0].op == 'M')
      ref_gone += n_gone;
  }
  else if (cg[0].len == n_gone)
  {
    if (cg[0].op == 'M')
      ref_gone += cg[0].len;
    cg.erase(cg.begin());
    if (cg.empty())
      return ref_gone;
    if (cg[0].op == 'D')
    {
      ref_gone += cg[0].len;
      cg.erase(cg.begin());
    }
  }
  else
  {
    if (cg[0].op == 'M')
      ref_gone += cg[0].len;
    unsigned const rest = n_gone - cg[0].len;
    cg.erase(cg.begin());
    if (cg.empty())
      return ref_gone;
    return ref_gone + cigar_shorten_front(cg, rest);
  }
  return ref_gone;
}

inline bool long_enough(Record const & record, Options const & opts)
{
  return !(static_cast<long>(record.seq.size()) < opts.minReadLen ||
           (record.mapq < 25 && static_cast<long>(record.seq.size()) < opts.minReadLenMapQ0));
}

template <class V>
void erase_range(V & v, size_t a, size_t b)
{
  a = std::min(a, v.size());
  b = std::min(b, v.size());
  if (a < b)
    v.erase(v.begin() + static_cast<long>(a), v.begin() + static_cast<long>(b));
}

// :484-521
inline bool strip_soft_clips(Record & record, Options const & opts)
{
  long n_cigar = static_cast<long>(record.cigar.size());
  if (n_cigar >= 1)
  {
    if (record.cigar[0].op == 'S')
    {
      uint32_t const count = record.cigar[0].len;
      erase_range(record.seq, 0, count);
      erase_range(record.qual, 0, count);
      record.cigar.erase(record.cigar.begin());
      --n_cigar;
    }
    if (n_cigar >= 2)
    {
      auto const last_el = record.cigar[n_cigar - 1];
      if (last_el.op == 'S')
      {
        long const n_bases = static_cast<long>(record.seq.size());
        long const left = std::max(0l, n_bases - static_cast<long>(last_el.len));
        record.seq.resize(left);
        record.qual.resize(left);
        record.cigar.pop_back();
      }
    }
  }
  return long_enough(record, opts);
}

// :523-584
inline bool strip_n_ends(Record & record, Options const & opts)
{
  int n_ns = 0;
  auto is_n = [&](long idx) { return idx >= 0 && idx < static_cast<long>(record.seq.size()) && record.seq[idx] == 15; };
  if (is_n(0))
  {
    ++n_ns;
    int idx = 1;
    while (is_n(idx) && idx < static_cast<long>(record.seq.size()) - 1)
    {
      ++n_ns;
      ++idx;
    }
    erase_range(record.seq, 0, n_ns);
    erase_range(record.qual, 0, n_ns);
    if (!flag_unmapped(record))
    {
      unsigned const shift = cigar_shorten_front(record.cigar, n_ns);
      record.pos += shift;
    }
  }
  if (!long_enough(record, opts))
    return false;
  n_ns = 0;
  if (is_n(static_cast<long>(record.seq.size()) - 1))
  {
    ++n_ns;
    int idx = static_cast<int>(record.seq.size()) - 2;
    while (is_n(idx) && idx > 0)
    {
      ++n_ns;
      --idx;
    }
    erase_range(record.seq, record.seq.size() - n_ns, record.seq.size());
    erase_range(record.qual, record.qual.size() - std::min<size_t>(n_ns, record.qual.size()), record.qual.size());
    if (!flag_unmapped(record))
      cigar_shorten_back(record.cigar, n_ns);
  }
  return long_enough(record, opts);
}

// :567-604: (bases to clip off the reverse read's front, positions to shift it by)
inline std::pair<int, int> clip_and_shift(Record const & rev, int fwd_start)
{
  int n_clip = 0, n_shift = 0;
  unsigned ci = 0;
  long rev_at = rev.pos;
  unsigned n = 0;
  auto const & cigar = rev.cigar;
  if (!cigar.empty() && cigar[0].op == 'S')
  {
    n_clip = cigar[0].len;
    ++ci;
  }
  while (ci < cigar.size())
  {
    char const o = cigar[ci].op;
    n = 0;
    while (rev_at < fwd_start && n < cigar[ci].len)
    {
      if (o != 'D')
        ++n_clip;
      if (o != 'I')
        ++rev_at;
      ++n;
    }
    if (rev_at == fwd_start)
      break;
    ++ci;
  }
  if (ci < cigar.size() && cigar[ci].op == 'D')
    n_shift = static_cast<int>(cigar[ci].len) - static_cast<int>(n);
  return {n_clip, n_shift};
}

// :606-665
inline bool trim_adapters(Record & fwd, Record & rev, Options const & opts)
{
  if (strip_soft_clips(fwd, opts) && strip_soft_clips(rev, opts))
    return false;
  int const diff = fwd.pos - rev.pos;
  if (diff < 0)
    return true;
  auto const cs = clip_and_shift(rev, fwd.pos);
  int const index = cs.first, shift = cs.second;
  erase_range(rev.seq, 0, index);
  erase_range(rev.qual, 0, index);
  cigar_shorten_front(rev.cigar, index);
  if (fwd.seq.size() > rev.seq.size() && index > 0)
  {
    int const over = static_cast<int>(fwd.seq.size() - rev.seq.size());
    erase_range(fwd.seq, rev.seq.size(), fwd.seq.size());
    erase_range(fwd.qual, rev.qual.size(), fwd.qual.size());
    cigar_shorten_back(fwd.cigar, over);
  }
  rev.pos = fwd.pos;
  if (shift > 0)
    rev.pos += shift;
  fwd.mate_pos = rev.pos;
  return long_enough(fwd, opts);
}

inline void reverse_complement(Record & record) // seqan::reverseComplement on IUPAC: the 4-bit code with its bits mirrored
{
  std::reverse(record.seq.begin(), record.seq.end());
  for (auto & c : record.seq)
    c = static_cast<uint8_t>(((c & 1u) << 3) | ((c & 2u) << 1) | ((c & 4u) >> 1) | ((c & 8u) >> 3));
}

// qualityFilterSlice2 (:667-1045) over the records `readRegion` hands over, in file order.  `in` holds the records of the
// whole file; the region of :681-699 (records of contig `rid` that overlap [begin, end), htslib's iterator: an alignment
// without reference bases counts as one position long) is cut here.  interval: 0-based first and last position (i2, i3).
inline void shrink_interval(Options const & opts, std::vector<Record> const & in, int32_t rid, int interval_begin, int interval_end,
                            long & counter, bool const is_single_contig, std::vector<Record> & out)
{
  int const begin = std::max(interval_begin - (opts.maxFragLen - 100), 0);
  long const end = static_cast<long>(interval_end) + (opts.maxFragLen - 100);
  std::multiset<Record> ready;
  std::unordered_map<std::string, Record, NameHash> waiting;
  long origin = -1;
  std::vector<uint32_t> bins;
  long const bin_cap = opts.no_filter_on_coverage ? (std::numeric_limits<int>::max() / 10)
                                                      : static_cast<long>(opts.avgCovByReadLen * 50.0 * 2.5);
  long const max_frag = opts.maxFragLen;
  auto count_at = [&](long bin) -> long { return bin >= 0 && bin < static_cast<long>(bins.size()) ? bins[bin] : 0; };

  auto single_ok = [&](Record const & rec) -> bool // :716-734
  {
    if (static_cast<long>(rec.pos + static_cast<long>(rec.seq.size())) < static_cast<long>(interval_begin) || rec.pos > interval_end)
      return false;
    if (rec.mapq < 40 || static_cast<long>(rec.seq.size()) < opts.minUnpairedReadLen || an_end_clipped(rec.cigar, 12) ||
        both_ends_clipped(rec.cigar, 5) || matching_bases(rec.cigar) < (opts.minNumMatching + 5) ||
        quals_of_20_and_more(rec.qual) < static_cast<long>(rec.seq.size()) / 4l)
      return false;
    return true;
  };

  auto mate_ok = [&](Record const & rec) -> bool // :736-776
  {
    if (opts.is_filtering_mapq0 && rec.mapq <= 1)
      return false;
    long const len = static_cast<long>(rec.seq.size());
    if (rec.pos + len < static_cast<long>(interval_begin) && static_cast<long>(rec.pos) + rec.tlen < static_cast<long>(interval_begin))
      return false;
    if (static_cast<long>(rec.pos) > static_cast<long>(interval_end) &&
        static_cast<long>(rec.pos) + rec.tlen - len > static_cast<long>(interval_end))
      return false;
    if (flag_unmapped(rec))
      return true;
    if (len < opts.minReadLen || (rec.mapq < 55 && both_ends_clipped(rec.cigar, 12)) ||
        (rec.mapq < 5 && an_end_clipped(rec.cigar, len / 4)) || both_ends_clipped(rec.cigar, len / 3) ||
        matching_bases(rec.cigar) < opts.minNumMatching || quals_of_20_and_more(rec.qual) <= len / 10l)
      return false;
    return true;
  };

  auto keep_single = [&](Record && rec) -> void // :778-813
  {
    std::string kept;
    if (!scores_say_keep(rec, kept, opts))
      return;
    if (!strip_n_ends(rec, opts))
      return;
    rec.aux = std::move(kept);
    long const bin = (rec.pos - origin) / 50l;
    if (bin >= static_cast<long>(bins.size()))
      bins.resize(bin + 1, 0u);
    else if (bins[bin] >= (bin_cap / 3l))
    {
      ++bins[bin];
      return;
    }
    two_level_quals(rec.qual);
    strip_hard_clips(rec.cigar);
    if (opts.change_read_names)
    {
      rec.name = counter_name(counter);
      ++counter;
    }
    ++bins[bin];
    ready.insert(std::move(rec));
  };

  auto finish_mate = [&](Record & rec, long const num) -> bool // :815-840
  {
    std::string kept;
    if (!scores_say_keep(rec, kept, opts))
      return false;
    if (!strip_n_ends(rec, opts))
      return false;
    rec.aux = std::move(kept);
    two_level_quals(rec.qual);
    strip_hard_clips(rec.cigar);
    if (opts.change_read_names)
      rec.name = counter_name(num);
    return true;
  };

  auto write_unless_too_deep = [&](Record const & rec) // :888-903, :1022-1040
  {
    long const bin1 = (rec.pos - origin) / 50l;
    long const bin2 = (rec.mate_pos - origin) / 50l;
    if (count_at(bin1) < (opts.SUPER_HI_DEPTH * bin_cap) || (flag_multiple(rec) && count_at(bin2) < (opts.SUPER_HI_DEPTH * bin_cap)))
      out.push_back(rec);
  };

  for (Record const & from_file : in)
  {
    // readRegion: what the iterator over [begin, end) returns
    {
      if (from_file.ref_id != rid)
        continue;
      long span = 0;
      for (auto const & c : from_file.cigar)
        if (c.op == 'M' || c.op == 'D' || c.op == 'N' || c.op == '=' || c.op == 'X')
          span += c.len;
      long const end_pos = static_cast<long>(from_file.pos) + (span > 0 && !flag_unmapped(from_file) ? span : 1);
      if (end_pos <= begin || from_file.pos >= end)
        continue;
    }
    Record record = from_file;
    // :849-853
    if ((record.flag & static_cast<uint32_t>(opts.sam_flag_filter)) != 0 || (record.tlen != 0 && std::abs(record.tlen) < opts.minReadLen))
      continue;
    if (origin < 0) // :855-861
    {
      if (record.pos < 0)
        continue;
      origin = record.pos;
    }
    // :866-907
    if (ready.size() > 0 && (record.pos > (max_frag + ready.begin()->pos + 600)))
    {
      for (auto it = waiting.begin(); it != waiting.end();)
      {
        if (static_cast<long>(record.pos) > (max_frag + it->second.pos + 400))
        {
          make_single(it->second);
          if (single_ok(it->second))
            keep_single(std::move(it->second));
          it = waiting.erase(it);
        }
        else
          ++it;
      }
      auto it = ready.begin();
      while (it != ready.end() && (record.pos > (max_frag + it->pos + 400)))
      {
        write_unless_too_deep(*it);
        ++it;
      }
      ready.erase(ready.begin(), it);
    }
    // :909-922
    if (is_single_contig)
    {
      if (record.mate_ref_id == record.ref_id)
      {
        record.ref_id = 0;
        record.mate_ref_id = 0;
      }
      else
      {
        record.ref_id = 0;
        record.mate_ref_id = 1;
      }
    }
    // :924-929
    if ((flag_unmapped(record) || flag_next_unmapped(record)) && flag_rc(record) == flag_next_rc(record))
    {
      reverse_complement(record);
      std::reverse(record.qual.begin(), record.qual.end());
      record.flag ^= 16u;
    }
    // :931-937
    if (record.ref_id != record.mate_ref_id || flag_rc(record) == flag_next_rc(record) || std::abs(record.tlen) > max_frag ||
        (record.tlen > 0 && flag_rc(record)) || (record.tlen < 0 && !flag_rc(record)))
      make_single(record);
    if (!flag_multiple(record)) // :939-946
    {
      if (single_ok(record))
        keep_single(std::move(record));
      continue;
    }
    if (!mate_ok(record)) // :949-950
      continue;
    auto mate = waiting.find(record.name);
    if (mate == waiting.end()) // :954-963
    {
      if (record.mate_pos >= record.pos)
      {
        std::string const key = record.name;
        waiting[key] = std::move(record);
      }
      continue;
    }
    long const bin1 = (record.pos - origin) / 50l;
    long const bin2 = (mate->second.pos - origin) / 50l;
    {
      long const max_bin = std::max(bin1, bin2);
      if (max_bin >= static_cast<long>(bins.size()))
        bins.resize(max_bin + 1, 0u);
    }
    ++bins[bin1];
    ++bins[bin2];
    if (bins[bin1] < bin_cap) // :979-1016
    {
      if (bins[bin2] < bin_cap)
      {
        bool is_ok;
        if (record.tlen == 0 ||
            std::abs(record.tlen) > static_cast<long>(std::max(record.seq.size(), mate->second.seq.size())))
          is_ok = true;
        else if (flag_rc(record))
          is_ok = trim_adapters(mate->second, record, opts);
        else
          is_ok = trim_adapters(record, mate->second, opts);
        if (is_ok && finish_mate(record, counter) && finish_mate(mate->second, counter))
        {
          if ((!flag_unmapped(record) && !flag_unmapped(mate->second)) || (flag_unmapped(record) && single_ok(mate->second)) ||
              (flag_unmapped(mate->second) && single_ok(record)))
          {
            ++counter;
            ready.insert(std::move(record));
            ready.insert(std::move(mate->second));
          }
        }
      }
      else if (bins[bin1] < (bin_cap / 3))
      {
        make_single(record);
        if (single_ok(record))
          keep_single(std::move(record));
      }
    }
    else if (bins[bin2] < (bin_cap / 3))
    {
      make_single(mate->second);
      if (single_ok(mate->second))
        keep_single(std::move(mate->second));
    }
    waiting.erase(mate);
  }
  // :1006-1017 leftovers become unpaired
  for (auto && rec : waiting)
  {
    make_single(rec.second);
    if (single_ok(rec.second))
      keep_single(std::move(rec.second));
  }
  waiting.clear();
  for (auto const & rec : ready) // :1020-1041
    write_unless_too_deep(rec);
}

// the header text of the one-interval case (:1304-1335): @HD, @RG and the interval's @SQ line
inline std::string single_contig_header(std::string const & text, std::string const & chrom)
{
  std::string const sq = "@SQ\tSN:" + chrom + "\t";
  std::string out;
  size_t t = 0;
  while (t <= text.size())
  {
    size_t const nl = std::min(text.find('\n', t), text.size());
    size_t const line_size = nl - t;
    if (line_size > 4 && (text.compare(t, 4, "@HD\t") == 0 || text.compare(t, 4, "@RG\t") == 0 ||
                          (line_size > sq.size() && text.compare(t, sq.size(), sq) == 0)))
      out += text.substr(t, line_size) + '\n';
    t += line_size + 1;
  }
  return out;
}
// --- BAM record stream <-> Record (SAM spec 4.2; the reference does this through seqan / htslib) ---------------------------
inline bool decode_records(uint8_t const * p, size_t len, std::vector<Record> & out)
{
  size_t at = 0;
  while (at + 4 <= len)
  {
    int32_t block;
    std::memcpy(&block, p + at, 4);
    at += 4;
    if (block < 32 || at + static_cast<size_t>(block) > len)
      return false;
    uint8_t const * b = p + at;
    at += static_cast<size_t>(block);
    Record r;
    int32_t l_seq;
    uint16_t n_cigar, flag;
    std::memcpy(&r.ref_id, b, 4);
    std::memcpy(&r.pos, b + 4, 4);
    uint8_t const l_read_name = b[8];
    r.mapq = b[9];
    std::memcpy(&n_cigar, b + 12, 2);
    std::memcpy(&flag, b + 14, 2);
    r.flag = flag;
    std::memcpy(&l_seq, b + 16, 4);
    std::memcpy(&r.mate_ref_id, b + 20, 4);
    std::memcpy(&r.mate_pos, b + 24, 4);
    std::memcpy(&r.tlen, b + 28, 4);
    size_t const o_cigar = 32 + l_read_name, o_seq = o_cigar + 4ull * n_cigar, o_qual = o_seq + (static_cast<size_t>(l_seq) + 1) / 2,
                 o_aux = o_qual + static_cast<size_t>(l_seq);
    if (l_seq < 0 || o_aux > static_cast<size_t>(block))
      return false;
    r.name.assign(reinterpret_cast<char const *>(b + 32), l_read_name ? l_read_name - 1u : 0u);
    for (unsigned c = 0; c < n_cigar; ++c)
    {
      uint32_t w;
      std::memcpy(&w, b + o_cigar + 4ull * c, 4);
      r.cigar.push_back(CigarElement{"MIDNSHP=X???????"[w & 15u], w >> 4});
    }
    r.seq.resize(static_cast<size_t>(l_seq));
    for (int32_t i = 0; i < l_seq; ++i)
      r.seq[i] = (b[o_seq + i / 2] >> ((i & 1) ? 0 : 4)) & 15u;
    r.qual.assign(b + o_qual, b + o_aux);
    r.aux.assign(reinterpret_cast<char const *>(b + o_aux), static_cast<size_t>(block) - o_aux);
    out.push_back(std::move(r));
  }
  return at == len;
}

inline uint16_t reg2bin(int64_t beg, int64_t end) // SAM spec 5.3
{
  --end;
  if (beg >> 14 == end >> 14) return static_cast<uint16_t>(((1 << 15) - 1) / 7 + (beg >> 14));
  if (beg >> 17 == end >> 17) return static_cast<uint16_t>(((1 << 12) - 1) / 7 + (beg >> 17));
  if (beg >> 20 == end >> 20) return static_cast<uint16_t>(((1 << 9) - 1) / 7 + (beg >> 20));
  if (beg >> 23 == end >> 23) return static_cast<uint16_t>(((1 << 6) - 1) / 7 + (beg >> 23));
  if (beg >> 26 == end >> 26) return static_cast<uint16_t>(((1 << 3) - 1) / 7 + (beg >> 26));
  return 0;
}

inline void encode_record(Record const & r, std::vector<uint8_t> & out)
{
  long span = 0;
  for (auto const & c : r.cigar)
    if (c.op == 'M' || c.op == 'D' || c.op == 'N' || c.op == '=' || c.op == 'X')
      span += c.len;
  int64_t const end = static_cast<int64_t>(r.pos) + ((r.flag & 4u) || span == 0 ? 1 : span);
  uint16_t const bin = reg2bin(r.pos < 0 ? -1 : r.pos, r.pos < 0 ? 0 : end);
  uint8_t const l_read_name = static_cast<uint8_t>(r.name.size() + 1);
  int32_t const l_seq = static_cast<int32_t>(r.seq.size());
  int32_t const block = static_cast<int32_t>(32 + l_read_name + 4 * r.cigar.size() + (r.seq.size() + 1) / 2 + r.seq.size() + r.aux.size());
  size_t const at = out.size();
  out.resize(at + 4 + static_cast<size_t>(block));
  uint8_t * b = out.data() + at;
  std::memcpy(b, &block, 4);
  b += 4;
  std::memcpy(b, &r.ref_id, 4);
  std::memcpy(b + 4, &r.pos, 4);
  b[8] = l_read_name;
  b[9] = r.mapq;
  std::memcpy(b + 10, &bin, 2);
  uint16_t const n_cigar = static_cast<uint16_t>(r.cigar.size()), flag = static_cast<uint16_t>(r.flag);
  std::memcpy(b + 12, &n_cigar, 2);
  std::memcpy(b + 14, &flag, 2);
  std::memcpy(b + 16, &l_seq, 4);
  std::memcpy(b + 20, &r.mate_ref_id, 4);
  std::memcpy(b + 24, &r.mate_pos, 4);
  std::memcpy(b + 28, &r.tlen, 4);
  std::memcpy(b + 32, r.name.c_str(), l_read_name);
  uint8_t * q = b + 32 + l_read_name;
  for (auto const & c : r.cigar)
  {
    char const * ops = "MIDNSHP=X";
    char const * f = std::strchr(ops, c.op);
    uint32_t const w = (c.len << 4) | static_cast<uint32_t>(f ? f - ops : 15);
    std::memcpy(q, &w, 4);
    q += 4;
  }
  std::memset(q, 0, (r.seq.size() + 1) / 2);
  for (size_t i = 0; i < r.seq.size(); ++i)
    q[i / 2] |= static_cast<uint8_t>(r.seq[i] << ((i & 1) ? 0 : 4));
  q += (r.seq.size() + 1) / 2;
  for (size_t i = 0; i < r.seq.size(); ++i)
    q[i] = i < r.qual.size() ? r.qual[i] : 0xFF;
  q += r.seq.size();
  std::memcpy(q, r.aux.data(), r.aux.size());
}
} // namespace gto_shrink
