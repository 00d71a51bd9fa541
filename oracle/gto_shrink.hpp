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
  char operation;
  uint32_t count;
};

struct Record // seqan::BamAlignmentRecord as far as bamshrink touches it
{
  std::string qName;
  uint32_t flag = 0;
  int32_t rID = -1, beginPos = -1;
  uint8_t mapQ = 0;
  std::vector<CigarElement> cigar;
  int32_t rNextId = -1, pNext = -1, tLen = 0;
  std::vector<uint8_t> seq;  // 4-bit codes, one per base
  std::vector<uint8_t> qual; // raw phred
  std::string tags;          // raw aux bytes
  bool operator<(Record const & b) const { return beginPos < b.beginPos; } // :312-316
};

inline bool flag_multiple(Record const & r) { return r.flag & 1u; }
inline bool flag_unmapped(Record const & r) { return r.flag & 4u; }
inline bool flag_next_unmapped(Record const & r) { return r.flag & 8u; }
inline bool flag_rc(Record const & r) { return r.flag & 16u; }
inline bool flag_next_rc(Record const & r) { return r.flag & 32u; }

// :34-61
inline char long_to_ascii(long in)
{
  if (in >= 31)
    ++in;
  return static_cast<char>('!' + in);
}

inline std::string decimal_to_read_name_string(long in)
{
  long const CHAR_SET_SIZE = 93;
  std::string str;
  while (in >= CHAR_SET_SIZE)
  {
    long const rem = in % CHAR_SET_SIZE;
    in = in / CHAR_SET_SIZE;
    str.push_back(long_to_ascii(rem));
  }
  str.push_back(long_to_ascii(in));
  return str;
}

// :64-76
inline void removeHardClipped(std::vector<CigarElement> & cigar)
{
  long n_cigar = static_cast<long>(cigar.size());
  if (n_cigar >= 1 && cigar[0].operation == 'H')
  {
    cigar.erase(cigar.begin());
    --n_cigar;
  }
  if (n_cigar >= 2 && cigar[n_cigar - 1].operation == 'H')
    cigar.pop_back();
}

// :78-87
inline long countHighBaseQuality(std::vector<uint8_t> const & qual)
{
  return std::count_if(qual.begin(), qual.end(), [](uint8_t q) { return q >= 20; });
}

inline void binarizeQual(std::vector<uint8_t> & qual)
{
  for (auto & q : qual)
    q = q >= 24 ? ('?' - 33) : (',' - 33);
}

// :89-99
inline bool is_clipped_both_ends(std::vector<CigarElement> const & cigar, long const min_clip = 15)
{
  return cigar.size() >= 1 && cigar.front().operation == 'S' && cigar.back().operation == 'S' &&
         static_cast<long>(cigar.front().count + cigar.back().count) >= min_clip;
}

inline bool is_one_end_clipped(std::vector<CigarElement> const & cigar, long const min_clip = 0)
{
  return cigar.size() == 0 || (cigar.front().operation == 'S' && static_cast<long>(cigar.front().count) >= min_clip) ||
         (cigar.back().operation == 'S' && static_cast<long>(cigar.back().count) >= min_clip);
}

// :102-308; true: the alignment is good.  new_tags: RG and the AS / XS / WS fields.
inline bool process_tags(Record const & record, std::string & new_tags, Options const & opts)
{
  size_t i = 0;
  int64_t as = -1, xs = -1, ws = -1;
  std::string const & tags = record.tags;
  size_t const tags_length = tags.size();
  while (i < tags_length)
  {
    size_t const begin_it = i;
    size_t end_it = begin_it;
    i += 3;
    if (i > tags_length)
      break; // (outside the area)
    char const type = tags[i - 1];
    bool is_as = false, is_xs = false, is_ws = false;
    if (tags[i - 2] == 'S')
    {
      if (tags[i - 3] == 'A')
        is_as = true;
      else if (tags[i - 3] == 'X')
        is_xs = true;
      else if (tags[i - 3] == 'W')
        is_ws = true;
    }
    auto set_alignment_score = [&](int64_t score)
    {
      if (is_as)
        as = score;
      else if (is_xs)
        xs = score;
      else if (is_ws)
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
      std::memcpy(&num, tags.data() + i, sizeof(num));
      set_alignment_score(static_cast<int64_t>(num));
      i += sizeof(num);
      end_it = i;
    };
    switch (type)
    {
    case 'A': ++i; break;
    case 'Z':
    {
      bool const is_rg = tags[i - 3] == 'R' && tags[i - 2] == 'G';
      while (i < tags_length && tags[i] != '\0' && tags[i] != '\n')
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
        new_tags.append(tags, begin_it, end_it - begin_it);
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
    if (is_as || is_xs || is_ws)
      new_tags.append(tags, begin_it, end_it - begin_it);
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
      if (c.operation == 'M')
        matches += c.count;
      else if (c.operation == 'D' || c.operation == 'I')
        indels += c.count + 2;
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
inline void makeUnpaired(Record & record)
{
  record.pNext = -1;
  record.rNextId = -1;
  record.flag &= ~8u;
  record.flag &= ~2u;
  record.flag &= ~1u;
  record.flag &= ~32u;
}

// :358-369
inline long countMatchingBases(std::vector<CigarElement> const & cigarString)
{
  long n = 0;
  for (auto const & c : cigarString)
    if (c.operation == 'M')
      n += c.count;
  return n;
}

// :388-420
inline void resetCigarStringEnd(std::vector<CigarElement> & cigarString, unsigned nRemoved)
{
  if (cigarString.empty())
    return;
  if (cigarString.back().operation == 'D')
  {
    cigarString.pop_back();
    if (cigarString.empty())
      return;
  }
  auto & cigar_end = cigarString.back();
  if (cigar_end.count > nRemoved)
    cigar_end.count -= nRemoved;
  else if (cigar_end.count == nRemoved)
  {
    cigarString.pop_back();
    if (!cigarString.empty() && cigarString.back().operation == 'D')
      cigarString.pop_back();
  }
  else
  {
    unsigned const nLeft = nRemoved - cigar_end.count;
    cigarString.pop_back();
    resetCigarStringEnd(cigarString, nLeft);
  }
}

// :423-482: the number of reference bases taken off the front
inline unsigned resetCigarStringBegin(std::vector<CigarElement> & cigarString, unsigned nRemoved)
{
  if (cigarString.empty())
    return 0;
  unsigned removed = 0;
  if (cigarString[0].operation == 'D')
  {
    removed = cigarString[0].count;
    cigarString.erase(cigarString.begin());
    if (cigarString.empty())
      return removed;
  }
  if (cigarString[0].count > nRemoved)
  {
    cigarString[0].count -= nRemoved;
    if (cigarString[0].operation == 'M')
      removed += nRemoved;
  }
  else if (cigarString[0].count == nRemoved)
  {
    if (cigarString[0].operation == 'M')
      removed += cigarString[0].count;
    cigarString.erase(cigarString.begin());
    if (cigarString.empty())
      return removed;
    if (cigarString[0].operation == 'D')
    {
      removed += cigarString[0].count;
      cigarString.erase(cigarString.begin());
    }
  }
  else
  {
    if (cigarString[0].operation == 'M')
      removed += cigarString[0].count;
    unsigned const nLeft = nRemoved - cigarString[0].count;
    cigarString.erase(cigarString.begin());
    if (cigarString.empty())
      return removed;
    return removed + resetCigarStringBegin(cigarString, nLeft);
  }
  return removed;
}

inline bool long_enough(Record const & record, Options const & opts)
{
  return !(static_cast<long>(record.seq.size()) < opts.minReadLen ||
           (record.mapQ < 25 && static_cast<long>(record.seq.size()) < opts.minReadLenMapQ0));
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
inline bool removeSoftClipped(Record & record, Options const & opts)
{
  long n_cigar = static_cast<long>(record.cigar.size());
  if (n_cigar >= 1)
  {
    if (record.cigar[0].operation == 'S')
    {
      uint32_t const count = record.cigar[0].count;
      erase_range(record.seq, 0, count);
      erase_range(record.qual, 0, count);
      record.cigar.erase(record.cigar.begin());
      --n_cigar;
    }
    if (n_cigar >= 2)
    {
      auto const last_cigar = record.cigar[n_cigar - 1];
      if (last_cigar.operation == 'S')
      {
        long const sequence_length = static_cast<long>(record.seq.size());
        long const left = std::max(0l, sequence_length - static_cast<long>(last_cigar.count));
        record.seq.resize(left);
        record.qual.resize(left);
        record.cigar.pop_back();
      }
    }
  }
  return long_enough(record, opts);
}

// :523-584
inline bool removeNsAtEnds(Record & record, Options const & opts)
{
  int nOfNs = 0;
  auto is_n = [&](long idx) { return idx >= 0 && idx < static_cast<long>(record.seq.size()) && record.seq[idx] == 15; };
  if (is_n(0))
  {
    ++nOfNs;
    int idx = 1;
    while (is_n(idx) && idx < static_cast<long>(record.seq.size()) - 1)
    {
      ++nOfNs;
      ++idx;
    }
    erase_range(record.seq, 0, nOfNs);
    erase_range(record.qual, 0, nOfNs);
    if (!flag_unmapped(record))
    {
      unsigned const shift = resetCigarStringBegin(record.cigar, nOfNs);
      record.beginPos += shift;
    }
  }
  if (!long_enough(record, opts))
    return false;
  nOfNs = 0;
  if (is_n(static_cast<long>(record.seq.size()) - 1))
  {
    ++nOfNs;
    int idx = static_cast<int>(record.seq.size()) - 2;
    while (is_n(idx) && idx > 0)
    {
      ++nOfNs;
      --idx;
    }
    erase_range(record.seq, record.seq.size() - nOfNs, record.seq.size());
    erase_range(record.qual, record.qual.size() - std::min<size_t>(nOfNs, record.qual.size()), record.qual.size());
    if (!flag_unmapped(record))
      resetCigarStringEnd(record.cigar, nOfNs);
  }
  return long_enough(record, opts);
}

// :567-604: (bases to clip off the reverse read's front, positions to shift it by)
inline std::pair<int, int> findNum2Clip(Record const & recordReverse, int forwardStartPos)
{
  int num2clip = 0, num2shift = 0;
  unsigned cigarIndex = 0;
  long reverseStartPos = recordReverse.beginPos;
  unsigned n = 0;
  auto const & cigar = recordReverse.cigar;
  if (!cigar.empty() && cigar[0].operation == 'S')
  {
    num2clip = cigar[0].count;
    ++cigarIndex;
  }
  while (cigarIndex < cigar.size())
  {
    char const cigarOperation = cigar[cigarIndex].operation;
    n = 0;
    while (reverseStartPos < forwardStartPos && n < cigar[cigarIndex].count)
    {
      if (cigarOperation != 'D')
        ++num2clip;
      if (cigarOperation != 'I')
        ++reverseStartPos;
      ++n;
    }
    if (reverseStartPos == forwardStartPos)
      break;
    ++cigarIndex;
  }
  if (cigarIndex < cigar.size() && cigar[cigarIndex].operation == 'D')
    num2shift = static_cast<int>(cigar[cigarIndex].count) - static_cast<int>(n);
  return {num2clip, num2shift};
}

// :606-665
inline bool removeAdapters(Record & recordForward, Record & recordReverse, Options const & opts)
{
  if (removeSoftClipped(recordForward, opts) && removeSoftClipped(recordReverse, opts))
    return false;
  int const startPosDiff = recordForward.beginPos - recordReverse.beginPos;
  if (startPosDiff < 0)
    return true;
  auto const clipAndShift = findNum2Clip(recordReverse, recordForward.beginPos);
  int const index = clipAndShift.first, shift = clipAndShift.second;
  erase_range(recordReverse.seq, 0, index);
  erase_range(recordReverse.qual, 0, index);
  resetCigarStringBegin(recordReverse.cigar, index);
  if (recordForward.seq.size() > recordReverse.seq.size() && index > 0)
  {
    int const forwardClip = static_cast<int>(recordForward.seq.size() - recordReverse.seq.size());
    erase_range(recordForward.seq, recordReverse.seq.size(), recordForward.seq.size());
    erase_range(recordForward.qual, recordReverse.qual.size(), recordForward.qual.size());
    resetCigarStringEnd(recordForward.cigar, forwardClip);
  }
  recordReverse.beginPos = recordForward.beginPos;
  if (shift > 0)
    recordReverse.beginPos += shift;
  recordForward.pNext = recordReverse.beginPos;
  return long_enough(recordForward, opts);
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
                            long & read_num, bool const is_single_contig, std::vector<Record> & out)
{
  int const begin = std::max(interval_begin - (opts.maxFragLen - 100), 0);
  long const end = static_cast<long>(interval_end) + (opts.maxFragLen - 100);
  std::multiset<Record> read_set;
  std::unordered_map<std::string, Record, NameHash> read_first;
  long first_pos = -1;
  std::vector<uint32_t> bin_counts;
  long const max_bin_sum = opts.no_filter_on_coverage ? (std::numeric_limits<int>::max() / 10)
                                                      : static_cast<long>(opts.avgCovByReadLen * 50.0 * 2.5);
  long const max_fragment_length = opts.maxFragLen;
  auto count_at = [&](long bin) -> long { return bin >= 0 && bin < static_cast<long>(bin_counts.size()) ? bin_counts[bin] : 0; };

  auto filter_unpaired = [&](Record const & rec) -> bool // :716-734
  {
    if (static_cast<long>(rec.beginPos + static_cast<long>(rec.seq.size())) < static_cast<long>(interval_begin) || rec.beginPos > interval_end)
      return false;
    if (rec.mapQ < 40 || static_cast<long>(rec.seq.size()) < opts.minUnpairedReadLen || is_one_end_clipped(rec.cigar, 12) ||
        is_clipped_both_ends(rec.cigar, 5) || countMatchingBases(rec.cigar) < (opts.minNumMatching + 5) ||
        countHighBaseQuality(rec.qual) < static_cast<long>(rec.seq.size()) / 4l)
      return false;
    return true;
  };

  auto filter_paired = [&](Record const & rec) -> bool // :736-776
  {
    if (opts.is_filtering_mapq0 && rec.mapQ <= 1)
      return false;
    long const len = static_cast<long>(rec.seq.size());
    if (rec.beginPos + len < static_cast<long>(interval_begin) && static_cast<long>(rec.beginPos) + rec.tLen < static_cast<long>(interval_begin))
      return false;
    if (static_cast<long>(rec.beginPos) > static_cast<long>(interval_end) &&
        static_cast<long>(rec.beginPos) + rec.tLen - len > static_cast<long>(interval_end))
      return false;
    if (flag_unmapped(rec))
      return true;
    if (len < opts.minReadLen || (rec.mapQ < 55 && is_clipped_both_ends(rec.cigar, 12)) ||
        (rec.mapQ < 5 && is_one_end_clipped(rec.cigar, len / 4)) || is_clipped_both_ends(rec.cigar, len / 3) ||
        countMatchingBases(rec.cigar) < opts.minNumMatching || countHighBaseQuality(rec.qual) <= len / 10l)
      return false;
    return true;
  };

  auto post_process_unpaired = [&](Record && rec) -> void // :778-813
  {
    std::string new_tags;
    if (!process_tags(rec, new_tags, opts))
      return;
    if (!removeNsAtEnds(rec, opts))
      return;
    rec.tags = std::move(new_tags);
    long const bin = (rec.beginPos - first_pos) / 50l;
    if (bin >= static_cast<long>(bin_counts.size()))
      bin_counts.resize(bin + 1, 0u);
    else if (bin_counts[bin] >= (max_bin_sum / 3l))
    {
      ++bin_counts[bin];
      return;
    }
    binarizeQual(rec.qual);
    removeHardClipped(rec.cigar);
    if (opts.change_read_names)
    {
      rec.qName = decimal_to_read_name_string(read_num);
      ++read_num;
    }
    ++bin_counts[bin];
    read_set.insert(std::move(rec));
  };

  auto post_process_paired = [&](Record & rec, long const num) -> bool // :815-840
  {
    std::string new_tags;
    if (!process_tags(rec, new_tags, opts))
      return false;
    if (!removeNsAtEnds(rec, opts))
      return false;
    rec.tags = std::move(new_tags);
    binarizeQual(rec.qual);
    removeHardClipped(rec.cigar);
    if (opts.change_read_names)
      rec.qName = decimal_to_read_name_string(num);
    return true;
  };

  auto write_if_not_too_deep = [&](Record const & rec) // :888-903, :1022-1040
  {
    long const bin1 = (rec.beginPos - first_pos) / 50l;
    long const bin2 = (rec.pNext - first_pos) / 50l;
    if (count_at(bin1) < (opts.SUPER_HI_DEPTH * max_bin_sum) || (flag_multiple(rec) && count_at(bin2) < (opts.SUPER_HI_DEPTH * max_bin_sum)))
      out.push_back(rec);
  };

  for (Record const & from_file : in)
  {
    // readRegion: what the iterator over [begin, end) returns
    {
      if (from_file.rID != rid)
        continue;
      long span = 0;
      for (auto const & c : from_file.cigar)
        if (c.operation == 'M' || c.operation == 'D' || c.operation == 'N' || c.operation == '=' || c.operation == 'X')
          span += c.count;
      long const end_pos = static_cast<long>(from_file.beginPos) + (span > 0 && !flag_unmapped(from_file) ? span : 1);
      if (end_pos <= begin || from_file.beginPos >= end)
        continue;
    }
    Record record = from_file;
    // :849-853
    if ((record.flag & static_cast<uint32_t>(opts.sam_flag_filter)) != 0 || (record.tLen != 0 && std::abs(record.tLen) < opts.minReadLen))
      continue;
    if (first_pos < 0) // :855-861
    {
      if (record.beginPos < 0)
        continue;
      first_pos = record.beginPos;
    }
    // :866-907
    if (read_set.size() > 0 && (record.beginPos > (max_fragment_length + read_set.begin()->beginPos + 600)))
    {
      for (auto it = read_first.begin(); it != read_first.end();)
      {
        if (static_cast<long>(record.beginPos) > (max_fragment_length + it->second.beginPos + 400))
        {
          makeUnpaired(it->second);
          if (filter_unpaired(it->second))
            post_process_unpaired(std::move(it->second));
          it = read_first.erase(it);
        }
        else
          ++it;
      }
      auto it = read_set.begin();
      while (it != read_set.end() && (record.beginPos > (max_fragment_length + it->beginPos + 400)))
      {
        write_if_not_too_deep(*it);
        ++it;
      }
      read_set.erase(read_set.begin(), it);
    }
    // :909-922
    if (is_single_contig)
    {
      if (record.rNextId == record.rID)
      {
        record.rID = 0;
        record.rNextId = 0;
      }
      else
      {
        record.rID = 0;
        record.rNextId = 1;
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
    if (record.rID != record.rNextId || flag_rc(record) == flag_next_rc(record) || std::abs(record.tLen) > max_fragment_length ||
        (record.tLen > 0 && flag_rc(record)) || (record.tLen < 0 && !flag_rc(record)))
      makeUnpaired(record);
    if (!flag_multiple(record)) // :939-946
    {
      if (filter_unpaired(record))
        post_process_unpaired(std::move(record));
      continue;
    }
    if (!filter_paired(record)) // :949-950
      continue;
    auto find_it = read_first.find(record.qName);
    if (find_it == read_first.end()) // :954-963
    {
      if (record.pNext >= record.beginPos)
      {
        std::string const key = record.qName;
        read_first[key] = std::move(record);
      }
      continue;
    }
    long const bin1 = (record.beginPos - first_pos) / 50l;
    long const bin2 = (find_it->second.beginPos - first_pos) / 50l;
    {
      long const max_bin = std::max(bin1, bin2);
      if (max_bin >= static_cast<long>(bin_counts.size()))
        bin_counts.resize(max_bin + 1, 0u);
    }
    ++bin_counts[bin1];
    ++bin_counts[bin2];
    if (bin_counts[bin1] < max_bin_sum) // :979-1016
    {
      if (bin_counts[bin2] < max_bin_sum)
      {
        bool is_ok;
        if (record.tLen == 0 ||
            std::abs(record.tLen) > static_cast<long>(std::max(record.seq.size(), find_it->second.seq.size())))
          is_ok = true;
        else if (flag_rc(record))
          is_ok = removeAdapters(find_it->second, record, opts);
        else
          is_ok = removeAdapters(record, find_it->second, opts);
        if (is_ok && post_process_paired(record, read_num) && post_process_paired(find_it->second, read_num))
        {
          if ((!flag_unmapped(record) && !flag_unmapped(find_it->second)) || (flag_unmapped(record) && filter_unpaired(find_it->second)) ||
              (flag_unmapped(find_it->second) && filter_unpaired(record)))
          {
            ++read_num;
            read_set.insert(std::move(record));
            read_set.insert(std::move(find_it->second));
          }
        }
      }
      else if (bin_counts[bin1] < (max_bin_sum / 3))
      {
        makeUnpaired(record);
        if (filter_unpaired(record))
          post_process_unpaired(std::move(record));
      }
    }
    else if (bin_counts[bin2] < (max_bin_sum / 3))
    {
      makeUnpaired(find_it->second);
      if (filter_unpaired(find_it->second))
        post_process_unpaired(std::move(find_it->second));
    }
    read_first.erase(find_it);
  }
  // :1006-1017 leftovers become unpaired
  for (auto && rec : read_first)
  {
    makeUnpaired(rec.second);
    if (filter_unpaired(rec.second))
      post_process_unpaired(std::move(rec.second));
  }
  read_first.clear();
  for (auto const & rec : read_set) // :1020-1041
    write_if_not_too_deep(rec);
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
    std::memcpy(&r.rID, b, 4);
    std::memcpy(&r.beginPos, b + 4, 4);
    uint8_t const l_read_name = b[8];
    r.mapQ = b[9];
    std::memcpy(&n_cigar, b + 12, 2);
    std::memcpy(&flag, b + 14, 2);
    r.flag = flag;
    std::memcpy(&l_seq, b + 16, 4);
    std::memcpy(&r.rNextId, b + 20, 4);
    std::memcpy(&r.pNext, b + 24, 4);
    std::memcpy(&r.tLen, b + 28, 4);
    size_t const o_cigar = 32 + l_read_name, o_seq = o_cigar + 4ull * n_cigar, o_qual = o_seq + (static_cast<size_t>(l_seq) + 1) / 2,
                 o_aux = o_qual + static_cast<size_t>(l_seq);
    if (l_seq < 0 || o_aux > static_cast<size_t>(block))
      return false;
    r.qName.assign(reinterpret_cast<char const *>(b + 32), l_read_name ? l_read_name - 1u : 0u);
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
    r.tags.assign(reinterpret_cast<char const *>(b + o_aux), static_cast<size_t>(block) - o_aux);
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
    if (c.operation == 'M' || c.operation == 'D' || c.operation == 'N' || c.operation == '=' || c.operation == 'X')
      span += c.count;
  int64_t const end = static_cast<int64_t>(r.beginPos) + ((r.flag & 4u) || span == 0 ? 1 : span);
  uint16_t const bin = reg2bin(r.beginPos < 0 ? -1 : r.beginPos, r.beginPos < 0 ? 0 : end);
  uint8_t const l_read_name = static_cast<uint8_t>(r.qName.size() + 1);
  int32_t const l_seq = static_cast<int32_t>(r.seq.size());
  int32_t const block = static_cast<int32_t>(32 + l_read_name + 4 * r.cigar.size() + (r.seq.size() + 1) / 2 + r.seq.size() + r.tags.size());
  size_t const at = out.size();
  out.resize(at + 4 + static_cast<size_t>(block));
  uint8_t * b = out.data() + at;
  std::memcpy(b, &block, 4);
  b += 4;
  std::memcpy(b, &r.rID, 4);
  std::memcpy(b + 4, &r.beginPos, 4);
  b[8] = l_read_name;
  b[9] = r.mapQ;
  std::memcpy(b + 10, &bin, 2);
  uint16_t const n_cigar = static_cast<uint16_t>(r.cigar.size()), flag = static_cast<uint16_t>(r.flag);
  std::memcpy(b + 12, &n_cigar, 2);
  std::memcpy(b + 14, &flag, 2);
  std::memcpy(b + 16, &l_seq, 4);
  std::memcpy(b + 20, &r.rNextId, 4);
  std::memcpy(b + 24, &r.pNext, 4);
  std::memcpy(b + 28, &r.tLen, 4);
  std::memcpy(b + 32, r.qName.c_str(), l_read_name);
  uint8_t * q = b + 32 + l_read_name;
  for (auto const & c : r.cigar)
  {
    char const * ops = "MIDNSHP=X";
    char const * f = std::strchr(ops, c.operation);
    uint32_t const w = (c.count << 4) | static_cast<uint32_t>(f ? f - ops : 15);
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
  std::memcpy(q, r.tags.data(), r.tags.size());
}
} // namespace gto_shrink
