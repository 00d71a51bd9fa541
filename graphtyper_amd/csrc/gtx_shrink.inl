// gtx_shrink.inl -- the read pre-filter in front of the path (part of gtx_bam.cpp: shares its BGZF reader and index look-ups).
//
// gtx_bam_shrink replaces gyper::bamshrink / bamshrink_multi (/root/reference/src/utilities/bamshrink.cpp:1248-1371), whose
// work is qualityFilterSlice2 (:667-1045): of the records around an interval keep the pairs and single reads that pass the
// mapping-quality / clipping / matching-bases / base-quality filters (:716-776), cut adapters off pairs whose fragment is
// shorter than a read (:606-665), drop reads by their AS / XS / WS tags (:102-308), cut Ns off the ends (:523-584), cap the
// depth per 50-bp bin (:703-711, :789-801, :966-1016), keep only the RG / AS / XS / WS tags, two-level qualities (:83-87) and
// short read names (:34-61), and write the survivors sorted by begin position (:886-907, :1020-1041).
//
// One pass over the file: a record is decoded once into a Read whose bases and qualities are never moved -- trimming narrows
// [lo, hi) -- and leaves as a BAM record appended to the output block.  The two orders the result depends on are kept: pairs
// waiting for their mate live in a hash map with the reference's name hash (:321-336; its iteration order decides which
// stale mate is counted first), reads waiting to be written in a multimap keyed by begin position (equal keys stay in insertion
// order, like the reference's multiset :312-316).
namespace shrink
{
struct Limits // bamshrink::Options (include/graphtyper/utilities/bamshrink.hpp:7-27)
{
  long max_frag = 1000, min_matching = 55, min_len = 75, min_len_low_mapq = 94, min_len_unpaired = 94, as_threshold = 40;
  bool drop_mapq0 = true, keep_coverage = false, rename = true;
  long bin_cap = 37, deep_factor = 2; // max_bin_sum (:710-711), SUPER_HI_DEPTH
  uint32_t flag_filter = 3840;
};

enum : uint32_t { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_EQ = 7, OP_X = 8 };
enum : uint16_t { F_PAIRED = 1, F_PROPER = 2, F_UNMAPPED = 4, F_MATE_UNMAPPED = 8, F_REVERSE = 16, F_MATE_REVERSE = 32 };

struct Read
{
  int32_t tid = -1, pos = -1, mtid = -1, mpos = -1, tlen = 0;
  uint16_t flag = 0;
  uint8_t mapq = 0;
  std::string name;
  std::vector<uint32_t> cigar; // len << 4 | op
  std::vector<uint8_t> bases, quals; // one 4-bit code / one phred byte per base of the record as read
  uint32_t lo = 0, hi = 0;          // what is left of them
  std::vector<uint8_t> aux;

  long len() const { return static_cast<long>(hi - lo); }
  bool is(uint16_t f) const { return (flag & f) != 0; }
  uint32_t op(size_t i) const { return cigar[i] & 15u; }
  uint32_t cnt(size_t i) const { return cigar[i] >> 4; }
  void cut_front(long n) { lo = static_cast<uint32_t>(std::min<long>(hi, static_cast<long>(lo) + std::max(0l, n))); }
  void cut_back(long n) { hi = static_cast<uint32_t>(std::max<long>(lo, static_cast<long>(hi) - std::max(0l, n))); }
};

inline long matching_bases(Read const & r) // countMatchingBases :358-369
{
  long n = 0;
  for (size_t i = 0; i < r.cigar.size(); ++i)
    n += r.op(i) == OP_M ? r.cnt(i) : 0;
  return n;
}

inline long bases_at_least(Read const & r, uint8_t q) // countHighBaseQuality :78-81
{
  long n = 0;
  for (uint32_t i = r.lo; i < r.hi; ++i)
    n += r.quals[i] >= q;
  return n;
}

inline bool clipped_at_both_ends(Read const & r, long min_clip) // :89-93
{
  return !r.cigar.empty() && r.op(0) == OP_S && r.op(r.cigar.size() - 1) == OP_S &&
         static_cast<long>(r.cnt(0) + r.cnt(r.cigar.size() - 1)) >= min_clip;
}

inline bool clipped_at_one_end(Read const & r, long min_clip) // :95-99
{
  if (r.cigar.empty())
    return true;
  size_t const last = r.cigar.size() - 1;
  return (r.op(0) == OP_S && static_cast<long>(r.cnt(0)) >= min_clip) || (r.op(last) == OP_S && static_cast<long>(r.cnt(last)) >= min_clip);
}

inline void single(Read & r) // makeUnpaired :345-356
{
  r.mpos = -1;
  r.mtid = -1;
  r.flag &= static_cast<uint16_t>(~(F_PAIRED | F_PROPER | F_MATE_UNMAPPED | F_MATE_REVERSE));
}

inline bool long_enough(Read const & r, Limits const & lim) // the test that ends :484-521, :523-584, :606-665
{
  return r.len() >= lim.min_len && !(r.mapq < 25 && r.len() < lim.min_len_low_mapq);
}

// resetCigarStringBegin (:423-482): n read bases leave the front; returns the reference positions that went with them.  A
// deletion that ends up in front goes as well (also in front of the first element looked at, whatever n is).
inline uint32_t cigar_drop_front(std::vector<uint32_t> & cg, uint32_t n)
{
  uint32_t ref = 0;
  size_t at = 0;
  auto op = [&](size_t i) { return cg[i] & 15u; };
  auto cnt = [&](size_t i) { return cg[i] >> 4; };
  for (;;)
  {
    if (at == cg.size())
      break;
    if (op(at) == OP_D)
    {
      ref += cnt(at);
      if (++at == cg.size())
        break;
    }
    uint32_t const c = cnt(at);
    bool const m = op(at) == OP_M;
    if (c > n)
    {
      cg[at] = ((c - n) << 4) | op(at);
      ref += m ? n : 0;
      break;
    }
    ref += m ? c : 0;
    ++at;
    if (c == n)
    {
      if (at < cg.size() && op(at) == OP_D)
        ref += cnt(at++);
      break;
    }
    n -= c;
  }
  cg.erase(cg.begin(), cg.begin() + static_cast<long>(at));
  return ref;
}

// resetCigarStringEnd (:388-420)
inline void cigar_drop_back(std::vector<uint32_t> & cg, uint32_t n)
{
  for (;;)
  {
    if (cg.empty())
      return;
    if ((cg.back() & 15u) == OP_D)
    {
      cg.pop_back();
      if (cg.empty())
        return;
    }
    uint32_t const c = cg.back() >> 4;
    if (c > n)
    {
      cg.back() = ((c - n) << 4) | (cg.back() & 15u);
      return;
    }
    cg.pop_back();
    if (c == n)
    {
      if (!cg.empty() && (cg.back() & 15u) == OP_D)
        cg.pop_back();
      return;
    }
    n -= c;
  }
}

// removeSoftClipped (:484-521)
inline bool drop_soft_clips(Read & r, Limits const & lim)
{
  if (!r.cigar.empty())
  {
    if (r.op(0) == OP_S)
    {
      r.cut_front(r.cnt(0));
      r.cigar.erase(r.cigar.begin());
    }
    if (r.cigar.size() >= 2 && r.op(r.cigar.size() - 1) == OP_S)
    {
      r.cut_back(r.cnt(r.cigar.size() - 1));
      r.cigar.pop_back();
    }
  }
  return long_enough(r, lim);
}

// removeNsAtEnds (:523-584): runs of N at either end go (never the whole read), with their part of the CIGAR when the read is mapped
inline bool drop_n_ends(Read & r, Limits const & lim)
{
  auto n_at = [&](uint32_t i) { return r.bases[i] == 15; };
  if (r.len() > 0 && n_at(r.lo))
  {
    uint32_t k = 1;
    while (r.lo + k + 1 < r.hi && n_at(r.lo + k))
      ++k;
    r.cut_front(k);
    if (!r.is(F_UNMAPPED))
      r.pos += static_cast<int32_t>(cigar_drop_front(r.cigar, k));
  }
  if (!long_enough(r, lim))
    return false;
  if (r.len() > 0 && n_at(r.hi - 1))
  {
    long k = 1;
    while (static_cast<long>(r.hi) - 1 - k > static_cast<long>(r.lo) && n_at(static_cast<uint32_t>(r.hi - 1 - k)))
      ++k;
    r.cut_back(k);
    if (!r.is(F_UNMAPPED))
      cigar_drop_back(r.cigar, static_cast<uint32_t>(k));
  }
  return long_enough(r, lim);
}

// process_tags (:102-308): false = the alignment scores say "drop"; kept: the RG field and the AS / XS / WS fields
inline bool keep_by_tags(Read const & r, Limits const & lim, std::vector<uint8_t> & kept)
{
  uint8_t const * a = r.aux.data();
  size_t const n = r.aux.size();
  int64_t score[3] = {-1, -1, -1}; // AS, XS, WS
  size_t i = 0;
  while (i + 3 <= n)
  {
    size_t const field = i;
    char const t0 = static_cast<char>(a[i]), t1 = static_cast<char>(a[i + 1]), type = static_cast<char>(a[i + 2]);
    i += 3;
    int const which = t1 != 'S' ? -1 : t0 == 'A' ? 0 : t0 == 'X' ? 1 : t0 == 'W' ? 2 : -1;
    size_t width = 0;
    int64_t value = 0;
    bool number = true;
    switch (type)
    {
    case 'c': width = 1; if (i + 1 <= n) value = static_cast<int8_t>(a[i]); break;
    case 'C': width = 1; if (i + 1 <= n) value = a[i]; break;
    case 's': width = 2; if (i + 2 <= n) { int16_t v; std::memcpy(&v, a + i, 2); value = v; } break;
    case 'S': width = 2; if (i + 2 <= n) { uint16_t v; std::memcpy(&v, a + i, 2); value = v; } break;
    case 'i': width = 4; if (i + 4 <= n) { int32_t v; std::memcpy(&v, a + i, 4); value = v; } break;
    case 'I': width = 4; if (i + 4 <= n) { uint32_t v; std::memcpy(&v, a + i, 4); value = v; } break;
    default: number = false; break;
    }
    if (number)
    {
      if (i + width > n)
        break; // (a field that runs out of the area)
      if (which >= 0)
      {
        score[which] = value;
        kept.insert(kept.end(), a + field, a + i + width);
      }
      i += width;
      continue;
    }
    if (type == 'A')
      ++i; // (a score tag of this type or of the next two keeps nothing)
    else if (type == 'f')
    {
      if (i + 4 > n)
        break;
      i += 4;
      if (which >= 0)
        kept.insert(kept.end(), a + field, a + i);
    }
    else if (type == 'Z')
    {
      while (i < n && a[i] != '\0' && a[i] != '\n')
        ++i;
      if (++i > n)
        break;
      if (t0 == 'R' && t1 == 'G')
        kept.insert(kept.end(), a + field, a + i);
    }
    else
      break; // any other type ends the walk
  }
  int64_t const as = score[0], xs = score[1];
  int64_t const ws = score[2] == -1 ? as : score[2];
  if (ws == -1 || xs == -1 || (r.is(F_PAIRED) && !r.is(F_MATE_UNMAPPED)))
    return true;
  if (ws <= xs + 5)
    return false;
  long matches = 0, indels = 0;
  for (size_t c = 0; c < r.cigar.size(); ++c)
  {
    if (r.op(c) == OP_M)
      matches += r.cnt(c);
    else if (r.op(c) == OP_D || r.op(c) == OP_I)
      indels += r.cnt(c) + 2;
  }
  return std::max(ws, as) + lim.as_threshold > matches - indels;
}

// removeAdapters (:606-665) with findNum2Clip (:567-604): the pair's fragment is not longer than a read.  false = drop the pair.
inline bool cut_adapters(Read & fwd, Read & rev, Limits const & lim)
{
  if (drop_soft_clips(fwd, lim) && drop_soft_clips(rev, lim))
    return false;
  if (fwd.pos < rev.pos)
    return true;
  // walk the reverse read's alignment up to the forward read's start
  long clip = 0, shift = 0, at = rev.pos;
  size_t c = 0;
  uint32_t used = 0;
  if (!rev.cigar.empty() && rev.op(0) == OP_S)
  {
    clip = rev.cnt(0);
    c = 1;
  }
  for (; c < rev.cigar.size(); ++c)
  {
    uint32_t const o = rev.op(c);
    for (used = 0; at < fwd.pos && used < rev.cnt(c); ++used)
    {
      clip += o != OP_D;
      at += o != OP_I;
    }
    if (at == fwd.pos)
      break;
  }
  if (c < rev.cigar.size() && rev.op(c) == OP_D)
    shift = static_cast<long>(rev.cnt(c)) - static_cast<long>(used);
  rev.cut_front(clip);
  cigar_drop_front(rev.cigar, static_cast<uint32_t>(clip));
  if (fwd.len() > rev.len() && clip > 0)
  {
    long const over = fwd.len() - rev.len();
    fwd.cut_back(over);
    cigar_drop_back(fwd.cigar, static_cast<uint32_t>(over));
  }
  rev.pos = fwd.pos + static_cast<int32_t>(std::max(0l, shift));
  fwd.mpos = rev.pos;
  return long_enough(fwd, lim);
}

inline std::string short_name(long n) // decimal_to_read_name_string (:34-61): base 93 over [!-?A-~], lowest digit first
{
  std::string s;
  do
  {
    long const d = n % 93;
    s.push_back(static_cast<char>('!' + d + (d >= 31)));
    n /= 93;
  } while (n > 0);
  return s;
}

struct NameHash // :321-336 (32-bit sum per character, xor-ed into 42)
{
  size_t operator()(std::string const & s) const
  {
    size_t h = 42;
    for (char c : s)
      h ^= static_cast<uint32_t>(c) + 0x9e3779b9u + (static_cast<uint32_t>(c) << 6) + static_cast<uint32_t>(c >> 2); // (c is signed: bytes above 127 extend)
    return h;
  }
};

inline uint16_t bam_bin(int64_t beg, int64_t end) // SAM spec 5.3
{
  --end;
  for (int shift = 14, first = 4681; shift <= 26; first = (first - 1) / 8, shift += 3)
    if (beg >> shift == end >> shift)
      return static_cast<uint16_t>(first + (beg >> shift));
  return 0;
}

// One interval's worth of state: qualityFilterSlice2's locals.
class Slice
{
public:
  Slice(Limits const & lim, int32_t first, int32_t last, bool one_contig, long & read_num, std::vector<uint8_t> & sink, gtx_shrink_stats & st) :
    lim_(lim), first_(first), last_(last), one_contig_(one_contig), read_num_(read_num), sink_(sink), st_(st)
  {
  }

  void take(Read && r) // the body of the loop :847-1004
  {
    if ((r.flag & lim_.flag_filter) != 0 || (r.tlen != 0 && std::labs(r.tlen) < lim_.min_len))
      return;
    if (origin_ < 0)
    {
      if (r.pos < 0)
        return;
      origin_ = r.pos;
    }
    if (!ready_.empty() && r.pos > lim_.max_frag + ready_.begin()->first + 600)
      release(r.pos);
    if (one_contig_) // the output header has this contig only (:909-922)
    {
      r.mtid = r.mtid == r.tid ? 0 : 1;
      r.tid = 0;
    }
    if ((r.is(F_UNMAPPED) || r.is(F_MATE_UNMAPPED)) && r.is(F_REVERSE) == r.is(F_MATE_REVERSE)) // :924-929
    {
      std::reverse(r.bases.begin(), r.bases.end());
      for (auto & b : r.bases)
        b = static_cast<uint8_t>(((b & 1u) << 3) | ((b & 2u) << 1) | ((b & 4u) >> 1) | ((b & 8u) >> 3));
      std::reverse(r.quals.begin(), r.quals.end());
      r.flag ^= F_REVERSE;
    }
    if (r.tid != r.mtid || r.is(F_REVERSE) == r.is(F_MATE_REVERSE) || std::labs(r.tlen) > lim_.max_frag || (r.tlen > 0 && r.is(F_REVERSE)) ||
        (r.tlen < 0 && !r.is(F_REVERSE))) // :931-937
      single(r);
    if (!r.is(F_PAIRED))
    {
      if (good_single(r))
        keep_single(std::move(r));
      return;
    }
    if (!good_mate(r))
      return;
    auto mate = waiting_.find(r.name);
    if (mate == waiting_.end())
    {
      if (r.mpos >= r.pos) // (a mate in front of this read is not going to come)
      {
        std::string const key = r.name;
        waiting_[key] = std::move(r);
      }
      return;
    }
    Read & m = mate->second;
    long const b1 = bin_of(r.pos), b2 = bin_of(m.pos);
    if (std::max(b1, b2) >= static_cast<long>(depth_.size()))
      depth_.resize(static_cast<size_t>(std::max(b1, b2)) + 1, 0u);
    ++depth_[b1];
    ++depth_[b2];
    if (depth_[b1] >= lim_.bin_cap) // :979-1016
    {
      if (depth_[b2] < lim_.bin_cap / 3)
      {
        single(m);
        if (good_single(m))
          keep_single(std::move(m));
      }
    }
    else if (depth_[b2] >= lim_.bin_cap)
    {
      if (depth_[b1] < lim_.bin_cap / 3)
      {
        single(r);
        if (good_single(r))
          keep_single(std::move(r));
      }
    }
    else
    {
      bool ok = true;
      if (r.tlen != 0 && std::labs(r.tlen) <= std::max(r.len(), m.len()))
        ok = r.is(F_REVERSE) ? cut_adapters(m, r, lim_) : cut_adapters(r, m, lim_);
      if (ok && finish_mate(r) && finish_mate(m) &&
          ((!r.is(F_UNMAPPED) && !m.is(F_UNMAPPED)) || (r.is(F_UNMAPPED) && good_single(m)) || (m.is(F_UNMAPPED) && good_single(r))))
      {
        ++read_num_;
        ++st_.pairs_kept;
        int32_t const p1 = r.pos, p2 = m.pos;
        ready_.emplace(p1, std::move(r));
        ready_.emplace(p2, std::move(m));
      }
    }
    waiting_.erase(mate);
  }

  void finish() // :1006-1041
  {
    for (auto & w : waiting_)
    {
      single(w.second);
      if (good_single(w.second))
        keep_single(std::move(w.second));
    }
    waiting_.clear();
    for (auto const & r : ready_)
      write(r.second);
    ready_.clear();
  }

private:
  long bin_of(long pos) const { return (pos - origin_) / 50; }
  long depth_at(long bin) const { return bin >= 0 && bin < static_cast<long>(depth_.size()) ? static_cast<long>(depth_[bin]) : 0; }

  bool good_single(Read const & r) const // filter_unpaired :716-734
  {
    if (r.pos + r.len() < first_ || r.pos > last_)
      return false;
    return !(r.mapq < 40 || r.len() < lim_.min_len_unpaired || clipped_at_one_end(r, 12) || clipped_at_both_ends(r, 5) ||
             matching_bases(r) < lim_.min_matching + 5 || bases_at_least(r, 20) < r.len() / 4);
  }

  bool good_mate(Read const & r) const // filter_paired :736-776
  {
    if (lim_.drop_mapq0 && r.mapq <= 1)
      return false;
    long const pos = r.pos, len = r.len();
    if (pos + len < first_ && pos + r.tlen < first_)
      return false;
    if (pos > last_ && pos + r.tlen - len > last_)
      return false;
    if (r.is(F_UNMAPPED)) // (its mate may be mapped)
      return true;
    return !(len < lim_.min_len || (r.mapq < 55 && clipped_at_both_ends(r, 12)) || (r.mapq < 5 && clipped_at_one_end(r, len / 4)) ||
             clipped_at_both_ends(r, len / 3) || matching_bases(r) < lim_.min_matching || bases_at_least(r, 20) <= len / 10);
  }

  static void two_levels(Read & r) // binarizeQual :83-87, removeHardClipped :64-76
  {
    for (uint32_t i = r.lo; i < r.hi; ++i)
      r.quals[i] = r.quals[i] >= 24 ? 30 : 11;
    if (!r.cigar.empty() && (r.cigar.front() & 15u) == OP_H)
      r.cigar.erase(r.cigar.begin());
    if (r.cigar.size() >= 2 && (r.cigar.back() & 15u) == OP_H)
      r.cigar.pop_back();
  }

  void keep_single(Read && r) // post_process_unpaired :778-813
  {
    std::vector<uint8_t> kept;
    if (!keep_by_tags(r, lim_, kept) || !drop_n_ends(r, lim_))
      return;
    r.aux.swap(kept);
    long const bin = bin_of(r.pos);
    if (bin >= static_cast<long>(depth_.size()))
      depth_.resize(static_cast<size_t>(bin) + 1, 0u);
    else if (depth_[bin] >= lim_.bin_cap / 3)
    {
      ++depth_[bin];
      ++st_.dropped_by_depth;
      return;
    }
    two_levels(r);
    if (lim_.rename)
      r.name = short_name(read_num_++);
    ++depth_[bin];
    ++st_.singles_kept;
    int32_t const p = r.pos;
    ready_.emplace(p, std::move(r));
  }

  bool finish_mate(Read & r) // post_process_paired :815-840
  {
    std::vector<uint8_t> kept;
    if (!keep_by_tags(r, lim_, kept) || !drop_n_ends(r, lim_))
      return false;
    r.aux.swap(kept);
    two_levels(r);
    if (lim_.rename)
      r.name = short_name(read_num_);
    return true;
  }

  // :866-907: first the mates that waited too long (they may join `ready_`), then everything far enough behind `pos`
  void release(long pos)
  {
    for (auto it = waiting_.begin(); it != waiting_.end();)
    {
      if (pos > lim_.max_frag + it->second.pos + 400)
      {
        single(it->second);
        if (good_single(it->second))
          keep_single(std::move(it->second));
        it = waiting_.erase(it);
      }
      else
        ++it;
    }
    auto it = ready_.begin();
    for (; it != ready_.end() && pos > lim_.max_frag + it->first + 400; ++it)
      write(it->second);
    ready_.erase(ready_.begin(), it);
  }

  void write(Read const & r) // the depth test of :888-903 and the record itself
  {
    long const deep = lim_.deep_factor * lim_.bin_cap;
    if (!(depth_at(bin_of(r.pos)) < deep || (r.is(F_PAIRED) && depth_at(bin_of(r.mpos)) < deep)))
    {
      ++st_.dropped_by_depth;
      return;
    }
    long span = 0;
    for (size_t c = 0; c < r.cigar.size(); ++c)
      if (r.op(c) == OP_M || r.op(c) == OP_D || r.op(c) == OP_N || r.op(c) == OP_EQ || r.op(c) == OP_X)
        span += r.cnt(c);
    uint16_t const bin = r.pos < 0 ? bam_bin(-1, 0) : bam_bin(r.pos, static_cast<int64_t>(r.pos) + (r.is(F_UNMAPPED) || span == 0 ? 1 : span));
    uint32_t const l_seq = r.hi - r.lo;
    uint8_t const l_name = static_cast<uint8_t>(r.name.size() + 1);
    int32_t const block = static_cast<int32_t>(32 + l_name + 4 * r.cigar.size() + (l_seq + 1) / 2 + l_seq + r.aux.size());
    size_t const at = sink_.size();
    sink_.resize(at + 4 + static_cast<size_t>(block));
    uint8_t * b = sink_.data() + at;
    auto put = [&](auto v)
    {
      std::memcpy(b, &v, sizeof(v));
      b += sizeof(v);
    };
    put(block);
    put(r.tid);
    put(r.pos);
    put(l_name);
    put(r.mapq);
    put(bin);
    put(static_cast<uint16_t>(r.cigar.size()));
    put(r.flag);
    put(static_cast<int32_t>(l_seq));
    put(r.mtid);
    put(r.mpos);
    put(r.tlen);
    std::memcpy(b, r.name.c_str(), l_name);
    b += l_name;
    if (!r.cigar.empty())
      std::memcpy(b, r.cigar.data(), 4 * r.cigar.size());
    b += 4 * r.cigar.size();
    for (uint32_t i = 0; i < l_seq; i += 2)
      *b++ = static_cast<uint8_t>((r.bases[r.lo + i] << 4) | (i + 1 < l_seq ? r.bases[r.lo + i + 1] : 0));
    std::memcpy(b, r.quals.data() + r.lo, l_seq);
    b += l_seq;
    if (!r.aux.empty())
      std::memcpy(b, r.aux.data(), r.aux.size());
    ++st_.records_written;
  }

  Limits const & lim_;
  long const first_, last_; // the interval, 0-based, both inside
  bool const one_contig_;
  long & read_num_;
  std::vector<uint8_t> & sink_;
  gtx_shrink_stats & st_;
  long origin_ = -1; // first_pos
  std::vector<uint32_t> depth_; // bin_counts: reads per 50 positions from origin_
  std::unordered_map<std::string, Read, NameHash> waiting_; // read_first
  std::multimap<int32_t, Read> ready_;                      // read_set
};

// the next record of `fp` as a Read; 0 = end of the file, -1 = damaged
inline int next_read(Bgzf & fp, std::vector<uint8_t> & buf, Read & r)
{
  int32_t block = 0;
  long const got = fp.read(&block, 4);
  if (got == 0)
    return 0;
  if (got != 4 || block < 32)
    return -1;
  buf.resize(static_cast<size_t>(block));
  if (fp.read(buf.data(), buf.size()) != static_cast<long>(buf.size()))
    return -1;
  uint8_t const * p = buf.data();
  int32_t l_seq;
  uint16_t n_cigar;
  std::memcpy(&r.tid, p, 4);
  std::memcpy(&r.pos, p + 4, 4);
  uint8_t const l_name = p[8];
  r.mapq = p[9];
  std::memcpy(&n_cigar, p + 12, 2);
  std::memcpy(&r.flag, p + 14, 2);
  std::memcpy(&l_seq, p + 16, 4);
  std::memcpy(&r.mtid, p + 20, 4);
  std::memcpy(&r.mpos, p + 24, 4);
  std::memcpy(&r.tlen, p + 28, 4);
  size_t const o_cigar = 32 + static_cast<size_t>(l_name), o_seq = o_cigar + 4ull * n_cigar, o_qual = o_seq + (static_cast<size_t>(std::max(l_seq, 0)) + 1) / 2,
               o_aux = o_qual + static_cast<size_t>(std::max(l_seq, 0));
  if (l_seq < 0 || o_aux > buf.size())
    return -1;
  r.name.assign(reinterpret_cast<char const *>(p + 32), l_name ? l_name - 1u : 0u);
  r.name.resize(std::strlen(r.name.c_str()));
  r.cigar.resize(n_cigar);
  if (n_cigar)
    std::memcpy(r.cigar.data(), p + o_cigar, 4ull * n_cigar);
  r.bases.resize(static_cast<size_t>(l_seq));
  for (int32_t i = 0; i < l_seq; ++i)
    r.bases[i] = (p[o_seq + i / 2] >> ((i & 1) ? 0 : 4)) & 15u;
  r.quals.assign(p + o_qual, p + o_aux);
  r.lo = 0;
  r.hi = static_cast<uint32_t>(l_seq);
  r.aux.assign(p + o_aux, p + buf.size());
  return 1;
}

// the header text of a one-interval run (:1304-1335): the @HD and @RG lines and the interval's own @SQ line
inline std::string one_contig_text(std::string const & text, std::string const & chrom)
{
  std::string const sq = "@SQ\tSN:" + chrom + "\t";
  std::string out;
  for (size_t at = 0; at < text.size();)
  {
    size_t const nl = std::min(text.find('\n', at), text.size());
    std::string const line = text.substr(at, nl - at);
    at = nl + 1;
    if (line.size() > 4 && (line.rfind("@HD\t", 0) == 0 || line.rfind("@RG\t", 0) == 0 || (line.size() > sq.size() && line.rfind(sq, 0) == 0)))
      out += line + '\n';
  }
  return out;
}

struct Header
{
  std::string text;
  std::vector<std::pair<std::string, int32_t>> refs;
};

inline bool read_header(Bgzf & fp, Header & h, std::string & err, std::string const & path)
{
  char magic[4];
  int32_t l_text = 0, n_ref = 0;
  auto rd = [&](void * d, size_t n) { return fp.read(d, n) == static_cast<long>(n); };
  if (!rd(magic, 4) || std::memcmp(magic, "BAM\1", 4) != 0 || !rd(&l_text, 4) || l_text < 0)
  {
    err = path + " is not a BAM file (CRAM is not read)";
    return false;
  }
  h.text.assign(static_cast<size_t>(l_text), '\0');
  if ((l_text && !rd(&h.text[0], static_cast<size_t>(l_text))) || !rd(&n_ref, 4) || n_ref < 0)
  {
    err = path + ": truncated header";
    return false;
  }
  h.text.resize(std::strlen(h.text.c_str()));
  h.refs.clear();
  for (int32_t i = 0; i < n_ref; ++i)
  {
    int32_t l_name = 0, l_ref = 0;
    std::string name;
    if (!rd(&l_name, 4) || l_name <= 0 || l_name > (1 << 20) || (name.resize(static_cast<size_t>(l_name)), !rd(&name[0], static_cast<size_t>(l_name))) || !rd(&l_ref, 4))
    {
      err = path + ": truncated header";
      return false;
    }
    name.resize(std::strlen(name.c_str()));
    h.refs.emplace_back(name, l_ref);
  }
  return true;
}

inline void append_header(Header const & h, std::vector<uint8_t> & out)
{
  auto put = [&](void const * p, size_t n) { out.insert(out.end(), static_cast<uint8_t const *>(p), static_cast<uint8_t const *>(p) + n); };
  int32_t const l_text = static_cast<int32_t>(h.text.size()), n_ref = static_cast<int32_t>(h.refs.size());
  put("BAM\1", 4);
  put(&l_text, 4);
  put(h.text.data(), h.text.size());
  put(&n_ref, 4);
  for (auto const & r : h.refs)
  {
    int32_t const l_name = static_cast<int32_t>(r.first.size() + 1);
    put(&l_name, 4);
    put(r.first.c_str(), r.first.size() + 1);
    put(&r.second, 4);
  }
}
} // namespace shrink
