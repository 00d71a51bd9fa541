// gtx_bam.cpp -- BAM ingest in front of gtx_stream_push, host side, zlib only.
//
// What the reference does with htslib before a record reaches genotype_only():
//   HtsReader::open                      src/utilities/hts_reader.cpp:17-124   header, @RG -> (read group, sample) tables, region
//   HtsReader::get_next_read_in_order    hts_reader.cpp:166-303                records of one position sorted by (l_qseq, packed bases)
//   HtsParallelReader::open/read_record  src/utilities/hts_parallel_reader.cpp:66-136   k-way merge of the files by
//                                        (tid, pos, l_qseq, packed bases)       include/graphtyper/utilities/hts_utils.hpp:48-108
//   HtsReader::get_sample_and_rg_index   hts_reader.cpp:354-387                RG tag -> read group / sample index
//   get_score_diff                       src/typer/alignment.cpp:140-325       AS - XS from the aux fields, with its parsing quirks
// Here: BGZF members are inflated member by member (zlib, raw deflate; ahead of the reader by a team of worker threads,
// see Bgzf below); a BAM record is parsed in place into a
// gtx_stream_record + its packed bases (copied verbatim: the kernels read BAM nibbles).  Equal keys keep file order, then
// position in the file (the reference's std::sort / heap leave the order of exact duplicates unspecified; their results do
// not depend on it).  A region starts from the .bai or .csi when there is one (else the file is scanned from its head).  Not read:
// CRAM (needs htslib's codecs).
#include "gtx_ctx.hpp"
#include "gtx_inflate.hpp"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace
{
// BGZF: a series of gzip members of at most 64 KB, each with its compressed size in a "BC" extra field (SAM spec 4.1); a
// virtual offset = (file offset of a member) << 16 | offset in its data.  Members are inflated one at a time (raw
// deflate), which is what makes seeking by virtual offset possible -- and what makes them independent: inflating is nine
// tenths of the time of reading a BAM file, so a reader keeps up to RING members in flight (refilled by halves).  The calling thread reads
// the compressed members ahead (sequential file reads), a small team of worker threads shared by all open readers
// inflates them, and the caller takes them in file order; a member nobody has started on when the caller needs it is
// inflated by the caller itself, so a reader is never slower than without the team (many readers on many host threads
// each still get their own core).  The team lives while a reader is open (GTX_BGZF_THREADS sizes it, 0 = none).
struct InflateJob
{
  std::vector<uint8_t> comp, data;
  long clen = 0;
  std::atomic<int> state{2}; // 0 queued, 1 being inflated, 2 done
  bool ok = false;
  std::mutex m;
  std::condition_variable cv;
};

void inflate_member(InflateJob & j)
{
  // the library's own decoder (gtx_inflate.hpp: built for whole members of known size, 1.5-1.9 x zlib's rate); what it refuses --
  // a damaged member, or a code whose tables do not fit its fixed ones -- gets zlib's verdict.  GTX_INFLATE=zlib: zlib only.
  static bool const own = !(std::getenv("GTX_INFLATE") && std::strcmp(std::getenv("GTX_INFLATE"), "zlib") == 0);
  // (GTX_BGZF_CRC=0: the member's CRC32 is not compared -- htslib compares it, and it is what holds the decoder here to the file)
  static bool const check_crc = !(std::getenv("GTX_BGZF_CRC") && std::getenv("GTX_BGZF_CRC")[0] == '0');
  j.ok = own && gtx::inflate_raw(j.comp.data(), static_cast<size_t>(j.clen), j.data.data(), j.data.size());
  uint32_t want = 0;
  std::memcpy(&want, j.comp.data() + j.clen, 4);
  if (j.ok && check_crc && gtx::crc32_of(j.data.data(), j.data.size()) != want)
    j.ok = false; // (zlib gets the member; a member that is damaged stays damaged)
  z_stream z{};
  if (!j.ok && inflateInit2(&z, -15) == Z_OK)
  {
    z.next_in = j.comp.data();
    z.avail_in = static_cast<uInt>(j.clen);
    z.next_out = j.data.data();
    z.avail_out = static_cast<uInt>(j.data.size());
    int const rc = inflate(&z, Z_FINISH);
    inflateEnd(&z);
    j.ok = rc == Z_STREAM_END && z.avail_out == 0 && (!check_crc || gtx::crc32_of(j.data.data(), j.data.size()) == want);
  }
  {
    // (notified under the lock: the reader may free the job as soon as it sees it done, and it sees that only after
    // this thread has let go of the mutex -- the last thing it touches)
    std::lock_guard<std::mutex> lock(j.m);
    j.state.store(2);
    j.cv.notify_all();
  }
}

class InflateTeam
{
public:
  static void acquire()
  {
    std::lock_guard<std::mutex> lock(gate());
    if (users()++ == 0)
    {
      unsigned n = std::min(std::max(std::thread::hardware_concurrency(), 1u), 16u);
      if (char const * e = std::getenv("GTX_BGZF_THREADS"))
        n = static_cast<unsigned>(std::max(0, std::atoi(e)));
      self() = new InflateTeam(n);
    }
  }
  static void release()
  {
    InflateTeam * gone = nullptr;
    {
      std::lock_guard<std::mutex> lock(gate());
      if (--users() == 0)
      {
        gone = self();
        self() = nullptr;
      }
    }
    delete gone;
  }
  // hands queued jobs to the team (no team: they stay queued and their reader inflates them when it gets there).  The
  // caller wakes at most ONE sleeping worker, and only when nobody is looking at the queue already: waking a thread costs the
  // caller a system call -- a third of a millisecond where the host is a virtual machine and the worker's core is halted,
  // as long as inflating the member takes -- so workers wake each other (run()) and linger a little before they sleep.
  static void submit(InflateJob * const * jobs, size_t n)
  {
    InflateTeam * t = self();
    if (!t || t->workers_.empty() || n == 0)
      return;
    bool wake;
    {
      std::lock_guard<std::mutex> lock(t->m_);
      t->queue_.insert(t->queue_.end(), jobs, jobs + n);
      t->pending_.store(t->queue_.size(), std::memory_order_release);
      wake = t->sleepers_ > 0 && t->lingering_.load(std::memory_order_acquire) == 0;
    }
    if (wake)
      t->cv_.notify_one();
  }
  // forgets the jobs of a reader that goes away (none of them is running any more: the reader has waited for those)
  static void forget(InflateJob const * first, InflateJob const * last)
  {
    InflateTeam * t = self();
    if (!t)
      return;
    std::lock_guard<std::mutex> lock(t->m_);
    t->queue_.erase(std::remove_if(t->queue_.begin(), t->queue_.end(), [&](InflateJob * j) { return j >= first && j < last; }), t->queue_.end());
    t->pending_.store(t->queue_.size(), std::memory_order_release);
  }

private:
  explicit InflateTeam(unsigned n)
  {
    if (char const * e = std::getenv("GTX_BGZF_LINGER_US"))
      linger_us_ = std::max(0, std::atoi(e));
    for (unsigned i = 0; i < n; ++i)
      workers_.emplace_back([this] { run(); });
  }
  ~InflateTeam()
  {
    {
      std::lock_guard<std::mutex> lock(m_);
      stop_ = true;
      stop_flag_.store(true);
    }
    cv_.notify_all();
    for (auto & w : workers_)
      w.join();
  }
  void run()
  {
    for (;;)
    {
      InflateJob * j = nullptr;
      bool wake_next = false;
      {
        std::unique_lock<std::mutex> lock(m_);
        if (queue_.empty() && !stop_)
        {
          // nothing to do: look at the queue for a little while without sleeping (two workers at most do; a reader hands
          // over its next members within that time when it is reading at all), then sleep
          if (linger_us_ > 0 && lingering_.load(std::memory_order_relaxed) < 2)
          {
            lingering_.fetch_add(1, std::memory_order_acq_rel);
            lock.unlock();
            auto const until = std::chrono::steady_clock::now() + std::chrono::microseconds(linger_us_);
            while (pending_.load(std::memory_order_acquire) == 0 && !stop_flag_.load(std::memory_order_relaxed) &&
                   std::chrono::steady_clock::now() < until)
              std::this_thread::yield();
            lock.lock();
            lingering_.fetch_sub(1, std::memory_order_acq_rel);
          }
          if (queue_.empty() && !stop_)
          {
            ++sleepers_;
            cv_.wait(lock, [this] { return stop_ || !queue_.empty(); });
            --sleepers_;
          }
        }
        if (stop_)
          return;
        j = queue_.front();
        queue_.pop_front();
        pending_.store(queue_.size(), std::memory_order_release);
        wake_next = !queue_.empty() && sleepers_ > 0; // more than this worker can take at once: the next worker is woken from here
        int expect = 0;
        if (!j->state.compare_exchange_strong(expect, 1)) // (its reader got there first)
          j = nullptr;
      }
      if (wake_next)
        cv_.notify_one();
      if (j)
        inflate_member(*j);
    }
  }
  static std::mutex & gate()
  {
    static std::mutex m;
    return m;
  }
  static int & users()
  {
    static int n = 0;
    return n;
  }
  static InflateTeam *& self()
  {
    static InflateTeam * t = nullptr;
    return t;
  }
  int linger_us_ = 300;                  // GTX_BGZF_LINGER_US
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<InflateJob *> queue_;
  std::vector<std::thread> workers_;
  bool stop_ = false;
  int sleepers_ = 0;                     // workers inside cv_.wait (under m_)
  std::atomic<int> lingering_{0};        // workers polling pending_ before they sleep
  std::atomic<size_t> pending_{0};       // queue_.size() for those
  std::atomic<bool> stop_flag_{false};
};

class Bgzf
{
public:
  ~Bgzf() { close(); }
  bool open(std::string const & path)
  {
    fp_ = std::fopen(path.c_str(), "rb");
    if (fp_)
    {
      InflateTeam::acquire();
      ring_.reset(new InflateJob[RING]);
    }
    return fp_ != nullptr;
  }
  void close()
  {
    if (fp_)
    {
      drain();
      ring_.reset();
      InflateTeam::release();
      std::fclose(fp_);
    }
    fp_ = nullptr;
  }
  bool is_open() const { return fp_ != nullptr; }
  // reads n bytes; returns the number read (short at the end of the file), -1 on a malformed member
  long read(void * dst, size_t n)
  {
    size_t done = 0;
    while (done < n)
    {
      if (at_ == data_.size() && !next_block())
        return bad_ ? -1 : static_cast<long>(done);
      size_t const take = std::min(n - done, data_.size() - at_);
      std::memcpy(static_cast<uint8_t *>(dst) + done, data_.data() + at_, take);
      at_ += take;
      done += take;
    }
    return static_cast<long>(done);
  }
  bool seek(uint64_t voffset)
  {
    drain();
    if (std::fseek(fp_, static_cast<long>(voffset >> 16), SEEK_SET) != 0)
      return false;
    data_.clear();
    at_ = 0;
    if ((voffset & 0xFFFFu) == 0)
      return true;
    if (!next_block() || (voffset & 0xFFFFu) > data_.size())
      return false;
    at_ = voffset & 0xFFFFu;
    return true;
  }

private:
  static constexpr unsigned RING = 32; // members in flight per reader (2 MB of data at most)
  enum Ahead { MORE, END, BROKEN };

  // the next member of the file into job j (compressed bytes only); END at the end of the file, BROKEN on a malformed member
  Ahead read_member(InflateJob & j)
  {
    for (;;)
    {
      uint8_t h[18];
      size_t const got = std::fread(h, 1, 18, fp_);
      if (got == 0)
        return END;
      if (got != 18 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4))
        return BROKEN;
      unsigned const xlen = h[10] | (h[11] << 8);
      // the BC field is the first extra field in every writer there is; look through the extra fields anyway
      std::vector<uint8_t> extra(xlen);
      std::memcpy(extra.data(), h + 12, std::min<size_t>(6, xlen));
      if (xlen > 6 && std::fread(extra.data() + 6, 1, xlen - 6, fp_) != xlen - 6)
        return BROKEN;
      long bsize = -1;
      for (unsigned i = 0; i + 4 <= xlen;)
      {
        unsigned const slen = extra[i + 2] | (extra[i + 3] << 8);
        if (extra[i] == 'B' && extra[i + 1] == 'C' && slen == 2 && i + 6 <= xlen)
          bsize = extra[i + 4] | (extra[i + 5] << 8);
        i += 4 + slen;
      }
      if (bsize < 0)
        return BROKEN;
      long const clen = bsize + 1 - 12 - static_cast<long>(xlen) - 8; // compressed data between header and CRC32 / ISIZE
      if (clen < 0)
        return BROKEN;
      j.comp.resize(static_cast<size_t>(clen) + 8);
      if (std::fread(j.comp.data(), 1, j.comp.size(), fp_) != j.comp.size())
        return BROKEN;
      uint32_t isize;
      std::memcpy(&isize, j.comp.data() + clen + 4, 4);
      if (isize > 65536)
        return BROKEN;
      if (isize == 0)
        continue; // (the end-of-file marker, or an empty member)
      j.clen = clen;
      j.data.resize(isize);
      return MORE;
    }
  }
  // reads members ahead until the ring is full or the file ends / breaks (which is reported when the caller gets there)
  void fill()
  {
    if (tail_ - head_ > RING / 2) // (refilled by halves: the members go to the team in one hand-over)
      return;
    InflateJob * fresh[RING];
    size_t n = 0;
    while (ahead_ == MORE && tail_ - head_ < RING)
    {
      InflateJob & j = ring_[tail_ % RING];
      Ahead const a = read_member(j);
      if (a != MORE)
      {
        ahead_ = a;
        break;
      }
      j.state.store(0);
      ++tail_;
      fresh[n++] = &j;
    }
    InflateTeam::submit(fresh, n);
  }
  bool next_block()
  {
    fill();
    if (head_ == tail_)
    {
      if (ahead_ == BROKEN)
        return fail();
      return false; // end of the file
    }
    InflateJob & j = ring_[head_ % RING];
    int expect = 0;
    if (j.state.compare_exchange_strong(expect, 1))
      inflate_member(j); // (nobody has started on it: this thread does)
    else
    {
      std::unique_lock<std::mutex> lock(j.m);
      j.cv.wait(lock, [&] { return j.state.load() == 2; });
    }
    ++head_;
    if (!j.ok)
      return fail();
    data_.swap(j.data);
    at_ = 0;
    return true;
  }
  // nothing of this reader is in flight or queued afterwards
  void drain()
  {
    if (!ring_)
      return;
    for (; head_ < tail_; ++head_)
    {
      InflateJob & j = ring_[head_ % RING];
      int expect = 0;
      if (j.state.compare_exchange_strong(expect, 2))
        continue; // (never started)
      std::unique_lock<std::mutex> lock(j.m);
      j.cv.wait(lock, [&] { return j.state.load() == 2; });
    }
    InflateTeam::forget(ring_.get(), ring_.get() + RING);
    head_ = tail_ = 0;
    ahead_ = MORE;
  }
  bool fail()
  {
    bad_ = true;
    return false;
  }
  std::FILE * fp_ = nullptr;
  std::unique_ptr<InflateJob[]> ring_;
  uint64_t head_ = 0, tail_ = 0; // members taken / read ahead
  Ahead ahead_ = MORE;
  std::vector<uint8_t> data_;
  size_t at_ = 0;
  bool bad_ = false;
};

// Where to start reading for a region: the .bai beside the BAM (SAM spec 5.2).  The smallest chunk start among the bins that
// can hold an overlapping record, not below the linear index' offset for the region's first 16 kb window: no overlapping
// record of a sorted file starts in front of it.  (Reading goes on sequentially from there -- records of other bins in
// between are filtered like any other -- and stops behind the region.)  false: no usable index.
bool bai_start(std::string const & bam_path, int32_t tid, int64_t begin, int64_t end, bool & any, uint64_t & voffset)
{
  std::FILE * fp = std::fopen((bam_path + ".bai").c_str(), "rb");
  if (!fp && bam_path.size() > 4)
    fp = std::fopen((bam_path.substr(0, bam_path.size() - 4) + ".bai").c_str(), "rb");
  if (!fp)
    return false;
  auto rd = [&](void * d, size_t n) { return std::fread(d, 1, n, fp) == n; };
  char magic[4];
  int32_t n_ref = 0;
  bool ok = rd(magic, 4) && std::memcmp(magic, "BAI\1", 4) == 0 && rd(&n_ref, 4) && tid >= 0 && tid < n_ref;
  any = false;
  voffset = UINT64_MAX;
  int64_t const last = std::min<int64_t>(end, (1ll << 29)) - 1;
  for (int32_t r = 0; ok && r <= tid; ++r)
  {
    int32_t n_bin = 0;
    ok = rd(&n_bin, 4);
    uint64_t best = UINT64_MAX;
    for (int32_t b = 0; ok && b < n_bin; ++b)
    {
      uint32_t bin = 0;
      int32_t n_chunk = 0;
      ok = rd(&bin, 4) && rd(&n_chunk, 4) && n_chunk >= 0;
      for (int32_t c = 0; ok && c < n_chunk; ++c)
      {
        uint64_t cb = 0, ce = 0;
        ok = rd(&cb, 8) && rd(&ce, 8);
        if (!ok || r != tid || bin == 37450) // (37450: the pseudo-bin with the mapped / unmapped counts)
          continue;
        // does the bin overlap [begin, last]?  level l holds bins of 2^(29 - 3 l) bases from offset ((8^l - 1) / 7)
        bool overlaps = false;
        for (int l = 0, first = 0; l <= 5; first += 1 << (3 * l), ++l)
        {
          int const shift = 29 - 3 * l;
          if (bin >= static_cast<uint32_t>(first) && bin < static_cast<uint32_t>(first + (1 << (3 * l))))
          {
            int64_t const k = bin - first;
            overlaps = k >= (begin >> shift) && k <= (last >> shift);
          }
        }
        if (overlaps && cb < best)
          best = cb;
      }
    }
    int32_t n_intv = 0;
    ok = ok && rd(&n_intv, 4) && n_intv >= 0;
    uint64_t linear = 0;
    for (int32_t i = 0; ok && i < n_intv; ++i)
    {
      uint64_t io = 0;
      ok = rd(&io, 8);
      if (ok && r == tid && i == (begin >> 14))
        linear = io;
    }
    if (ok && r == tid && best != UINT64_MAX)
    {
      any = true;
      voffset = std::max(best, linear);
    }
  }
  std::fclose(fp);
  return ok;
}

// The same from a .csi index (<bam>.csi; htslib's coordinate-sorted index with a free bin geometry: min_shift, depth; the
// file is BGZF-compressed).  Level l of the bins holds 8^l bins of 2^(min_shift + 3 (depth - l)) bases from bin number
// (8^l - 1) / 7; a bin carries `loffset`, the offset of the first record that overlaps it -- the lower bound the .bai
// takes from its linear index comes from the smallest bin around the region's first base here.  false: no usable index.
bool csi_start(std::string const & bam_path, int32_t tid, int64_t begin, int64_t end, bool & any, uint64_t & voffset)
{
  Bgzf z;
  if (!z.open(bam_path + ".csi") && !(bam_path.size() > 4 && z.open(bam_path.substr(0, bam_path.size() - 4) + ".csi")))
    return false;
  auto rd = [&](void * d, size_t n) { return z.read(d, n) == static_cast<long>(n); };
  char magic[4];
  int32_t min_shift = 0, depth = 0, l_aux = 0, n_ref = 0;
  bool ok = rd(magic, 4) && std::memcmp(magic, "CSI\1", 4) == 0 && rd(&min_shift, 4) && rd(&depth, 4) && rd(&l_aux, 4) && min_shift >= 0 &&
            min_shift <= 32 && depth >= 0 && depth <= 10 && l_aux >= 0 && l_aux < (1 << 24);
  if (ok && l_aux)
  {
    std::vector<char> aux(static_cast<size_t>(l_aux));
    ok = rd(aux.data(), aux.size());
  }
  ok = ok && rd(&n_ref, 4) && tid >= 0 && tid < n_ref;
  any = false;
  voffset = UINT64_MAX;
  int64_t const max_pos = 1ll << std::min(62, min_shift + 3 * depth);
  int64_t const last = std::min<int64_t>(end, max_pos) - 1;
  uint64_t const meta_bin = ((1ull << (3 * (depth + 1))) - 1) / 7 + 1; // the pseudo-bin with the mapped / unmapped counts
  for (int32_t r = 0; ok && r <= tid; ++r)
  {
    int32_t n_bin = 0;
    ok = rd(&n_bin, 4) && n_bin >= 0;
    uint64_t best = UINT64_MAX, lower = 0;
    int lower_level = -1;
    for (int32_t b = 0; ok && b < n_bin; ++b)
    {
      uint32_t bin = 0;
      uint64_t loffset = 0;
      int32_t n_chunk = 0;
      ok = rd(&bin, 4) && rd(&loffset, 8) && rd(&n_chunk, 4) && n_chunk >= 0;
      bool overlaps = false;
      if (ok && r == tid && bin != meta_bin)
      {
        uint64_t first = 0;
        for (int l = 0; l <= depth; first += 1ull << (3 * l), ++l)
        {
          int const shift = min_shift + 3 * (depth - l);
          if (bin >= first && bin < first + (1ull << (3 * l)))
          {
            int64_t const k = static_cast<int64_t>(bin - first);
            overlaps = k >= (begin >> shift) && k <= (last >> shift);
            if (k == (begin >> shift) && l > lower_level) // the smallest bin around the region's first base
            {
              lower_level = l;
              lower = loffset;
            }
          }
        }
      }
      for (int32_t c = 0; ok && c < n_chunk; ++c)
      {
        uint64_t cb = 0, ce = 0;
        ok = rd(&cb, 8) && rd(&ce, 8);
        if (ok && overlaps && cb < best)
          best = cb;
      }
    }
    if (ok && r == tid && best != UINT64_MAX)
    {
      any = true;
      voffset = std::max(best, lower);
    }
  }
  return ok;
}

struct Rec // one BAM record as the merge needs it
{
  gtx_stream_record r{};
  std::vector<uint8_t> seq; // (l_qseq + 1) / 2 bytes
  int64_t end_pos = 0;      // first reference position behind the alignment
};

// gt_pos_seq_same_pos / gt_pos_seq (hts_utils.hpp:48-108) as "a comes before b"
bool seq_before(Rec const & a, Rec const & b)
{
  if (a.r.l_qseq != b.r.l_qseq)
    return a.r.l_qseq < b.r.l_qseq;
  return a.seq < b.seq; // bytewise, equal lengths
}

bool record_before(Rec const & a, Rec const & b)
{
  if (a.r.tid != b.r.tid)
    return a.r.tid < b.r.tid;
  if (a.r.pos != b.r.pos)
    return a.r.pos < b.r.pos;
  return seq_before(a, b);
}

uint64_t name_hash(char const * s, size_t n) // identity of a read name: 64-bit FNV-1a
{
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i)
    h = (h ^ static_cast<uint8_t>(s[i])) * 1099511628211ull;
  return h;
}

struct File
{
  Bgzf fp;
  std::string path;
  std::vector<std::string> ref_names;
  std::vector<std::string> samples;          // of this file, in header order
  std::map<std::string, uint32_t> rg2index;  // read group id -> index in this file
  std::vector<uint32_t> rg2sample;           // -> sample index in this file
  uint32_t sample_offset = 0, rg_offset = 0;
  // region (by scanning): records of contig `want_tid` that overlap [begin, end)
  int32_t want_tid = -2;
  int64_t begin = 0, end = INT64_MAX;
  bool eof = false;
  bool indexed = false;        // the scan started from the .bai
  Rec ahead;                   // the first record of the next position
  bool have_ahead = false;
  std::deque<Rec> same_pos;    // the records of the current position, in order
  std::vector<uint8_t> buf;

  bool read_exact(void * dst, unsigned n)
  {
    return fp.read(dst, n) == static_cast<long>(n);
  }

  uint32_t num_rg() const { return std::max<uint32_t>(1, static_cast<uint32_t>(rg2sample.size())); }

  // alignment.cpp:140-325
  static uint8_t score_diff(uint8_t const * it, uint32_t l_aux)
  {
    uint32_t i = 0;
    int64_t as = -1, xs = -1;
    auto load = [&](auto tag, bool is_as, bool is_xs)
    {
      decltype(tag) num = 0;
      if (i + sizeof(num) > l_aux) // (a truncated aux area: the reference reads on into htslib's padding; here the field is not there)
      {
        i = l_aux;
        return;
      }
      std::memcpy(&num, it + i, sizeof(num));
      if (is_as)
        as = num;
      else if (is_xs)
        xs = num;
      i += sizeof(num);
    };
    while (i < l_aux)
    {
      i += 3;
      if (i > l_aux)
        break;
      char const type = static_cast<char>(it[i - 1]);
      bool const is_s = it[i - 2] == 'S', is_as = is_s && it[i - 3] == 'A', is_xs = is_s && it[i - 3] == 'X';
      switch (type)
      {
      case 'A': ++i; break;
      case 'Z':
        while (i < l_aux && it[i] != '\0' && it[i] != '\n')
          ++i;
        ++i;
        break;
      case 'c': load(int8_t(), is_as, is_xs); break;
      case 'C': load(uint8_t(), is_as, is_xs); break;
      case 's': load(int16_t(), is_as, is_xs); break;
      case 'S': load(uint16_t(), is_as, is_xs); break;
      case 'i': load(int32_t(), is_as, is_xs); break;
      case 'I': load(uint32_t(), is_as, is_xs); break;
      case 'f': i += 4; break;
      default: i = l_aux; break; // unknown tag type: the reference stops here
      }
    }
    if (as == -1 || as < xs)
      return 0;
    if (xs == -1)
      xs = 0;
    int64_t const diff = as - xs;
    return diff < 255 ? static_cast<uint8_t>(diff) : 255;
  }

  // bam_aux_get(rec, "RG"): htslib walks the fields by their types
  static bool find_rg(uint8_t const * aux, uint32_t l_aux, std::string & out)
  {
    uint64_t i = 0; // (64 bits: a hostile 'B' count must not wrap the cursor back into the area)
    while (i + 3 <= l_aux)
    {
      char const t0 = static_cast<char>(aux[i]), t1 = static_cast<char>(aux[i + 1]), type = static_cast<char>(aux[i + 2]);
      i += 3;
      uint64_t size = 0;
      switch (type)
      {
      case 'A': case 'c': case 'C': size = 1; break;
      case 's': case 'S': size = 2; break;
      case 'i': case 'I': case 'f': size = 4; break;
      case 'd': size = 8; break;
      case 'Z': case 'H':
      {
        uint64_t j = i;
        while (j < l_aux && aux[j] != '\0')
          ++j;
        if (t0 == 'R' && t1 == 'G' && type == 'Z')
        {
          out.assign(reinterpret_cast<char const *>(aux + i), j - i);
          return true;
        }
        size = j - i + 1;
        break;
      }
      case 'B':
      {
        if (i + 5 > l_aux)
          return false;
        char const sub = static_cast<char>(aux[i]);
        uint32_t n;
        std::memcpy(&n, aux + i + 1, 4);
        uint64_t const w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
        size = 5 + static_cast<uint64_t>(n) * w;
        break;
      }
      default: return false;
      }
      if (i + size > l_aux) // a field that runs past the aux area: malformed record, no read group
        return false;
      i += size;
    }
    return false;
  }

  // next record of the file that lies in the region; false at the end of the file (or behind the region)
  bool read_one(Rec & out, std::string & err)
  {
    for (;;)
    {
      if (eof)
        return false;
      int32_t block = 0;
      long const got = fp.read(&block, 4);
      if (got == 0)
      {
        eof = true;
        return false;
      }
      if (got != 4 || block < 32)
      {
        err = path + ": truncated BAM record";
        eof = true;
        return false;
      }
      buf.resize(static_cast<size_t>(block));
      if (!read_exact(buf.data(), static_cast<unsigned>(block)))
      {
        err = path + ": truncated BAM record";
        eof = true;
        return false;
      }
      uint8_t const * p = buf.data();
      int32_t tid, pos, l_seq, mtid, mpos, tlen;
      uint8_t l_read_name, mapq;
      uint16_t n_cigar, flag;
      std::memcpy(&tid, p, 4);
      std::memcpy(&pos, p + 4, 4);
      l_read_name = p[8];
      mapq = p[9];
      std::memcpy(&n_cigar, p + 12, 2);
      std::memcpy(&flag, p + 14, 2);
      std::memcpy(&l_seq, p + 16, 4);
      std::memcpy(&mtid, p + 20, 4);
      std::memcpy(&mpos, p + 24, 4);
      std::memcpy(&tlen, p + 28, 4);
      size_t const o_name = 32, o_cigar = o_name + l_read_name, o_seq = o_cigar + 4ull * n_cigar,
                   o_qual = o_seq + (static_cast<size_t>(l_seq) + 1) / 2, o_aux = o_qual + static_cast<size_t>(l_seq);
      if (l_seq < 0 || o_aux > buf.size())
      {
        err = path + ": malformed BAM record";
        eof = true;
        return false;
      }
      // reference span of the alignment (bam_endpos: M, D, N, =, X consume the reference; at least one position)
      int64_t span = 0;
      uint32_t first = 0, last = 0;
      for (uint32_t c = 0; c < n_cigar; ++c)
      {
        uint32_t w;
        std::memcpy(&w, p + o_cigar + 4ull * c, 4);
        if (c == 0)
          first = w;
        last = w;
        uint32_t const op = w & 15u;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8)
          span += w >> 4;
      }
      int64_t const end_pos = static_cast<int64_t>(pos) + (span > 0 ? span : 1);
      if (want_tid != -2)
      {
        if (tid != want_tid || end_pos <= begin)
        {
          if (tid > want_tid && tid >= 0 && want_tid >= 0)
          {
            eof = true; // sorted file: behind the contig
            return false;
          }
          continue;
        }
        if (pos >= end)
        {
          eof = true;
          return false;
        }
      }
      if (l_seq > 0xFFFF)
      {
        err = path + ": a read of more than 65535 bases";
        eof = true;
        return false;
      }
      out.r = gtx_stream_record{};
      out.r.flag = flag;
      out.r.mapq = mapq;
      out.r.tid = tid;
      out.r.mtid = mtid;
      out.r.pos = pos;
      out.r.isize = tlen;
      out.r.l_qseq = static_cast<uint16_t>(l_seq);
      out.r.mpos = mpos;
      out.r.n_cigar = n_cigar;
      out.r.cigar_front = first;
      out.r.cigar_back = last;
      out.r.name_id = name_hash(reinterpret_cast<char const *>(p + o_name), l_read_name ? l_read_name - 1u : 0u);
      uint32_t const l_aux = static_cast<uint32_t>(buf.size() - o_aux);
      out.r.score_diff = score_diff(p + o_aux, l_aux);
      uint32_t rg = 0, sample = 0;
      if (rg2sample.size() > 1) // hts_reader.cpp:354-387
      {
        std::string id;
        if (!find_rg(p + o_aux, l_aux, id))
        {
          err = path + ": a record without RG tag in a file with several read groups";
          eof = true;
          return false;
        }
        auto it = rg2index.find(id);
        if (it == rg2index.end())
        {
          err = path + ": unknown read group " + id;
          eof = true;
          return false;
        }
        rg = it->second;
        sample = rg2sample[rg];
      }
      out.r.rg = static_cast<uint16_t>(rg + rg_offset);
      out.r.sample = sample + sample_offset;
      out.seq.assign(p + o_seq, p + o_qual);
      out.end_pos = end_pos;
      return true;
    }
  }

  // HtsReader::get_next_read_in_order: the records of one position (the reference compares core.pos only), sorted
  bool next(Rec & out, std::string & err)
  {
    if (same_pos.empty())
    {
      if (!have_ahead)
        have_ahead = read_one(ahead, err);
      if (!have_ahead)
        return false;
      int32_t const pos = ahead.r.pos;
      std::vector<Rec> group;
      group.push_back(std::move(ahead));
      have_ahead = false;
      Rec r;
      while (read_one(r, err))
      {
        if (r.r.pos != pos)
        {
          ahead = std::move(r);
          have_ahead = true;
          break;
        }
        group.push_back(std::move(r));
        r = Rec();
      }
      std::stable_sort(group.begin(), group.end(), seq_before);
      for (auto & g : group)
        same_pos.push_back(std::move(g));
    }
    out = std::move(same_pos.front());
    same_pos.pop_front();
    return true;
  }
};

#include "gtx_shrink.inl"
} // namespace

struct gtx_reads
{
  std::vector<std::unique_ptr<File>> files;
  std::vector<std::string> samples;
  uint32_t n_rg = 0;
  // merge front: one record per file that still has some
  std::vector<std::pair<Rec, uint32_t>> front;
  std::string error;
};

namespace
{
int fail(gtx_reads * r, std::string const & msg, int status)
{
  gtx::g_last_error = msg;
  delete r;
  return status;
}
} // namespace

extern "C" int gtx_reads_open(const char * const * bam_paths, uint32_t n_paths, const char * region, gtx_reads ** out)
{
  if (!bam_paths || n_paths == 0 || !out)
    return GTX_ERR_ARG;
  *out = nullptr;
  auto * r = new gtx_reads();
  // "chr", "chr:begin", "chr:begin-end" (1-based, inclusive like a samtools region); "" / "." / NULL: everything
  std::string contig;
  int64_t begin = 0, end = INT64_MAX;
  bool const whole = !region || std::strlen(region) <= 1;
  if (!whole)
  {
    std::string const s(region);
    size_t const colon = s.rfind(':');
    contig = s.substr(0, colon);
    if (colon != std::string::npos)
    {
      std::string rest = s.substr(colon + 1);
      rest.erase(std::remove(rest.begin(), rest.end(), ','), rest.end());
      size_t const dash = rest.find('-');
      begin = std::max<int64_t>(0, std::atoll(rest.substr(0, dash).c_str()) - 1);
      if (dash != std::string::npos && dash + 1 < rest.size())
        end = std::atoll(rest.substr(dash + 1).c_str());
    }
  }
  for (uint32_t f = 0; f < n_paths; ++f)
  {
    auto file = std::make_unique<File>();
    file->path = bam_paths[f] ? bam_paths[f] : "";
    if (!file->fp.open(file->path))
      return fail(r, "could not open " + file->path, GTX_ERR_IO);
    char magic[4];
    int32_t l_text = 0, n_ref = 0;
    if (!file->read_exact(magic, 4) || std::memcmp(magic, "BAM\1", 4) != 0 || !file->read_exact(&l_text, 4) || l_text < 0)
    {
      return fail(r, file->path + " is not a BAM file (CRAM is not read)", GTX_ERR_UNSUPPORTED);
    }
    std::string text(static_cast<size_t>(l_text), '\0');
    if ((l_text && !file->read_exact(&text[0], static_cast<unsigned>(l_text))) || !file->read_exact(&n_ref, 4) || n_ref < 0)
    {
      return fail(r, file->path + ": truncated header", GTX_ERR_IO);
    }
    for (int32_t i = 0; i < n_ref; ++i)
    {
      int32_t l_name = 0, l_ref = 0;
      std::string name;
      if (!file->read_exact(&l_name, 4) || l_name <= 0 || (name.resize(static_cast<size_t>(l_name)), !file->read_exact(&name[0], static_cast<unsigned>(l_name))) ||
          !file->read_exact(&l_ref, 4))
      {
        return fail(r, file->path + ": truncated header", GTX_ERR_IO);
      }
      name.resize(std::strlen(name.c_str()));
      file->ref_names.push_back(name);
    }
    // @RG lines -> read groups and samples (hts_reader.cpp:31-80: first "\tID:", last "\tSM:")
    size_t at = 0;
    while (at < text.size())
    {
      size_t const nl = std::min(text.find('\n', at), text.size());
      std::string const line = text.substr(at, nl - at);
      at = nl + 1;
      if (line.rfind("@RG", 0) != 0)
        continue;
      size_t const pid = line.find("\tID:"), psm = line.rfind("\tSM:");
      if (pid == std::string::npos || psm == std::string::npos)
      {
        return fail(r, file->path + ": an @RG line without ID or SM", GTX_ERR_ARG);
      }
      size_t const eid = std::min(line.find('\t', pid + 1), line.size()), esm = std::min(line.find('\t', psm + 1), line.size());
      std::string const id = line.substr(pid + 4, eid - pid - 4), sm = line.substr(psm + 4, esm - psm - 4);
      file->rg2index[id] = static_cast<uint32_t>(file->rg2sample.size());
      auto it = std::find(file->samples.begin(), file->samples.end(), sm);
      file->rg2sample.push_back(static_cast<uint32_t>(it - file->samples.begin()));
      if (it == file->samples.end())
        file->samples.push_back(sm);
    }
    if (file->samples.empty()) // the file name up to its first '.' (hts_reader.cpp:83-91)
    {
      std::string s = file->path.substr(file->path.rfind('/') + 1);
      if (s.find('.') != std::string::npos)
        s = s.substr(0, s.find('.'));
      file->samples.push_back(s);
    }
    if (!whole)
    {
      auto it = std::find(file->ref_names.begin(), file->ref_names.end(), contig);
      if (it == file->ref_names.end())
      {
        return fail(r, file->path + ": no contig " + contig, GTX_ERR_ARG);
      }
      file->want_tid = static_cast<int32_t>(it - file->ref_names.begin());
      file->begin = begin;
      file->end = end;
      // with a .bai the scan starts at the first place an overlapping record can be, else behind the header
      bool any = false;
      uint64_t voffset = 0;
      if (bai_start(file->path, file->want_tid, begin, end, any, voffset) || csi_start(file->path, file->want_tid, begin, end, any, voffset))
      {
        file->indexed = true;
        if (!any)
          file->eof = true; // the index knows of no record there
        else if (!file->fp.seek(voffset))
          return fail(r, file->path + ": the index points outside the file", GTX_ERR_IO);
      }
    }
    file->sample_offset = static_cast<uint32_t>(r->samples.size());
    file->rg_offset = r->n_rg;
    r->samples.insert(r->samples.end(), file->samples.begin(), file->samples.end());
    r->n_rg += file->num_rg();
    r->files.push_back(std::move(file));
  }
  for (uint32_t f = 0; f < r->files.size(); ++f)
  {
    Rec rec;
    if (r->files[f]->next(rec, r->error))
      r->front.emplace_back(std::move(rec), f);
    if (!r->error.empty())
    {
      std::string const e = r->error;
      return fail(r, e, GTX_ERR_IO);
    }
  }
  *out = r;
  return GTX_OK;
}

extern "C" int gtx_reads_info(const gtx_reads * r, uint32_t * n_samples, uint32_t * n_read_groups)
{
  if (!r)
    return GTX_ERR_ARG;
  if (n_samples)
    *n_samples = static_cast<uint32_t>(r->samples.size());
  if (n_read_groups)
    *n_read_groups = r->n_rg;
  return GTX_OK;
}

extern "C" const char * gtx_reads_sample_name(const gtx_reads * r, uint32_t i)
{
  return r && i < r->samples.size() ? r->samples[i].c_str() : nullptr;
}

extern "C" int gtx_reads_next(gtx_reads * r, gtx_stream_record * recs, uint8_t * seq, uint32_t seq_stride, uint32_t cap, uint32_t * n)
{
  if (!r || !recs || !seq || !n)
    return GTX_ERR_ARG;
  *n = 0;
  while (*n < cap && !r->front.empty())
  {
    // the smallest record; equal keys: the file that was opened first
    size_t best = 0;
    for (size_t i = 1; i < r->front.size(); ++i)
      if (record_before(r->front[i].first, r->front[best].first))
        best = i;
    Rec & rec = r->front[best].first;
    if (rec.seq.size() > seq_stride)
    {
      gtx::g_last_error = "gtx_reads_next: a read does not fit in seq_stride";
      return GTX_ERR_ARG;
    }
    recs[*n] = rec.r;
    uint8_t * row = seq + static_cast<size_t>(*n) * seq_stride;
    std::memcpy(row, rec.seq.data(), rec.seq.size());
    std::memset(row + rec.seq.size(), 0, seq_stride - rec.seq.size());
    ++*n;
    uint32_t const f = r->front[best].second;
    Rec next;
    if (r->files[f]->next(next, r->error))
      r->front[best].first = std::move(next);
    else
      r->front.erase(r->front.begin() + static_cast<long>(best));
    if (!r->error.empty())
    {
      gtx::g_last_error = r->error;
      return GTX_ERR_IO;
    }
  }
  return GTX_OK;
}

extern "C" void gtx_reads_close(gtx_reads * r)
{
  if (!r)
    return;
  delete r;
}

// ---- the pre-filter (gtx_shrink.inl) ------------------------------------------------------------------------------------

extern "C" void gtx_shrink_params_default(gtx_shrink_params * p)
{
  if (!p)
    return;
  *p = gtx_shrink_params{};
  p->max_frag_len = 1000;        // options.hpp:63-69
  p->min_num_matching = 55;
  p->filter_mapq0 = 1;
  p->no_filter_on_coverage = 0;  // options.hpp:50
  p->min_read_len = 75;
  p->min_read_len_low_mapq = 94;
  p->min_unpaired_read_len = 94;
  p->sam_flag_filter = 3840;     // options.hpp:90
  p->as_filter_threshold = 40;
  p->avg_cov_by_readlen = 0.0;   // unknown
  p->change_read_names = 1;      // (release builds of the reference, bamshrink.cpp:24-28)
  p->compress_level = 1;         // "wb1" (bamshrink.cpp:1263)
}

extern "C" int gtx_bam_shrink(const char * bam_in, const char * const * chroms, const int32_t * begins, const int32_t * ends, uint32_t n_intervals,
                              const gtx_shrink_params * params, const char * bam_out, gtx_shrink_stats * stats)
{
  gtx_shrink_stats st{};
  if (stats)
    *stats = st;
  if (!bam_in || !chroms || !begins || !ends || n_intervals == 0 || !bam_out)
  {
    gtx::g_last_error = "gtx_bam_shrink: bad argument (at least one interval is needed, bamshrink.cpp:1296-1300)";
    return GTX_ERR_ARG;
  }
  gtx_shrink_params par;
  if (params)
    par = *params;
  else
    gtx_shrink_params_default(&par);
  shrink::Limits lim;
  lim.max_frag = par.max_frag_len;
  lim.min_matching = par.min_num_matching;
  lim.min_len = par.min_read_len;
  lim.min_len_low_mapq = par.min_read_len_low_mapq;
  lim.min_len_unpaired = par.min_unpaired_read_len;
  lim.as_threshold = par.as_filter_threshold;
  lim.drop_mapq0 = par.filter_mapq0 != 0;
  lim.rename = par.change_read_names != 0;
  lim.flag_filter = static_cast<uint32_t>(par.sam_flag_filter);
  // bamshrink.cpp:1268-1271 and :710-711: without a coverage the default one caps the bins and nothing counts as "super high"
  double const cov = par.avg_cov_by_readlen > 0.0 ? par.avg_cov_by_readlen : 0.30000001;
  lim.deep_factor = par.avg_cov_by_readlen > 0.0 ? 2 : 1000;
  lim.bin_cap = par.no_filter_on_coverage ? (std::numeric_limits<int>::max() / 10) : static_cast<long>(cov * 50.0 * 2.5);

  std::string const path(bam_in);
  shrink::Header head;
  std::string err;
  {
    Bgzf fp;
    if (!fp.open(path))
    {
      gtx::g_last_error = "could not open " + path;
      return GTX_ERR_IO;
    }
    if (!shrink::read_header(fp, head, err, path))
    {
      gtx::g_last_error = err;
      return GTX_ERR_UNSUPPORTED;
    }
  }
  std::vector<int32_t> tids(n_intervals);
  for (uint32_t i = 0; i < n_intervals; ++i)
  {
    std::string const chrom = chroms[i] ? chroms[i] : "";
    auto it = std::find_if(head.refs.begin(), head.refs.end(), [&](auto const & r) { return r.first == chrom; });
    if (it == head.refs.end() || begins[i] < 0 || ends[i] < begins[i])
    {
      gtx::g_last_error = path + ": no contig " + chrom + " (or an interval that ends in front of its begin)";
      return GTX_ERR_ARG;
    }
    tids[i] = static_cast<int32_t>(it - head.refs.begin());
  }
  bool const one_contig = n_intervals == 1;
  shrink::Header out_head = head;
  if (one_contig) // only this contig stays in the header (bamshrink.cpp:1304-1335)
  {
    out_head.text = shrink::one_contig_text(head.text, head.refs[static_cast<size_t>(tids[0])].first);
    out_head.refs.assign(1, head.refs[static_cast<size_t>(tids[0])]);
  }
  std::FILE * out = std::fopen(bam_out, "wb");
  if (!out)
  {
    gtx::g_last_error = std::string("could not create ") + bam_out;
    return GTX_ERR_IO;
  }
  std::vector<uint8_t> sink, packed;
  auto flush = [&](bool last) -> bool
  {
    // whole 0xff00-byte members while more is coming; the rest stays in the sink
    size_t const take = last ? sink.size() : sink.size() / 0xff00u * 0xff00u;
    if (take == 0 && !last)
      return true;
    packed.resize(take + take / 8 + (take / 0xff00u + 2) * 64);
    uint64_t n = 0;
    if (gtx_bgzf_compress(sink.data(), take, par.compress_level, last ? 1 : 0, packed.data(), packed.size(), &n) != GTX_OK)
      return false;
    sink.erase(sink.begin(), sink.begin() + static_cast<long>(take));
    return std::fwrite(packed.data(), 1, n, out) == n;
  };
  shrink::append_header(out_head, sink);
  long read_num = 0;
  int status = GTX_OK;
  std::vector<uint8_t> buf;
  for (uint32_t i = 0; i < n_intervals && status == GTX_OK; ++i)
  {
    // the records the reference asks its index for (bamshrink.cpp:681-699): those that overlap [first - pad, last + pad)
    int64_t const pad = lim.max_frag - 100;
    int64_t const from = std::max<int64_t>(static_cast<int64_t>(begins[i]) - pad, 0), to = static_cast<int64_t>(ends[i]) + pad;
    Bgzf fp;
    shrink::Header again;
    if (!fp.open(path) || !shrink::read_header(fp, again, err, path))
    {
      gtx::g_last_error = "could not read " + path;
      status = GTX_ERR_IO;
      break;
    }
    bool any = true;
    uint64_t voffset = 0;
    if (bai_start(path, tids[i], from, to, any, voffset) || csi_start(path, tids[i], from, to, any, voffset))
    {
      if (!any)
        continue;
      if (!fp.seek(voffset))
      {
        gtx::g_last_error = path + ": the index points outside the file";
        status = GTX_ERR_IO;
        break;
      }
    }
    shrink::Slice slice(lim, begins[i], ends[i], one_contig, read_num, sink, st);
    for (;;)
    {
      shrink::Read r;
      int const got = shrink::next_read(fp, buf, r);
      if (got == 0)
        break;
      if (got < 0)
      {
        gtx::g_last_error = path + ": damaged BAM record";
        status = GTX_ERR_IO;
        break;
      }
      if (r.tid != tids[i])
      {
        if (r.tid > tids[i] || r.tid < 0)
          break; // sorted file: behind the contig
        continue;
      }
      if (r.pos >= to)
        break;
      int64_t span = 0;
      for (size_t c = 0; c < r.cigar.size(); ++c)
        if (r.op(c) == shrink::OP_M || r.op(c) == shrink::OP_D || r.op(c) == shrink::OP_N || r.op(c) == shrink::OP_EQ || r.op(c) == shrink::OP_X)
          span += r.cnt(c);
      if (static_cast<int64_t>(r.pos) + (span > 0 && !r.is(shrink::F_UNMAPPED) ? span : 1) <= from)
        continue;
      ++st.records_read;
      slice.take(std::move(r));
      if (sink.size() > (8u << 20) && !flush(false))
      {
        gtx::g_last_error = std::string("could not write ") + bam_out;
        status = GTX_ERR_IO;
        break;
      }
    }
    if (status == GTX_OK)
      slice.finish();
  }
  if (status == GTX_OK && !flush(true))
  {
    gtx::g_last_error = std::string("could not write ") + bam_out;
    status = GTX_ERR_IO;
  }
  if (std::fclose(out) != 0 && status == GTX_OK)
  {
    gtx::g_last_error = std::string("could not write ") + bam_out;
    status = GTX_ERR_IO;
  }
  if (status != GTX_OK)
    std::remove(bam_out);
  else if (stats)
    *stats = st;
  return status;
}

extern "C" int gtx_inflate_raw(const void * in, uint64_t in_len, void * out, uint64_t out_len)
{
  if ((in_len && !in) || (out_len && !out))
    return GTX_ERR_ARG;
  std::vector<uint8_t> padded(in_len + 8, 0); // (the decoder loads 8 bytes at a time: a BGZF member has its CRC32 and ISIZE there)
  if (in_len)
    std::memcpy(padded.data(), in, in_len);
  uint8_t nothing = 0;
  if (gtx::inflate_raw(padded.data(), in_len, out_len ? static_cast<uint8_t *>(out) : &nothing, out_len))
    return GTX_OK;
  gtx::g_last_error = "gtx_inflate_raw: not a DEFLATE stream of the given size";
  return GTX_ERR_IO;
}

// bamshrink_multi (bamshrink.cpp:1352-1371) with readIntervals (:1047-1130): the intervals of a file -- lines of "contig first
// last", 1-based, sorted -- where a neighbour that begins within 2 * max_frag_len of the one before is one interval with it
// (otherwise the output could not stay sorted); then the filter over all of them into one file with the whole header.
extern "C" int gtx_bam_shrink_multi(const char * bam_in, const char * interval_file, const gtx_shrink_params * params, const char * bam_out,
                                    gtx_shrink_stats * stats)
{
  if (!bam_in || !interval_file || !bam_out)
  {
    gtx::g_last_error = "gtx_bam_shrink_multi: bad argument";
    return GTX_ERR_ARG;
  }
  gtx_shrink_params par;
  if (params)
    par = *params;
  else
    gtx_shrink_params_default(&par);
  std::FILE * fp = std::fopen(interval_file, "r");
  if (!fp)
  {
    gtx::g_last_error = std::string("Unable to locate interval file at: ") + interval_file;
    return GTX_ERR_IO;
  }
  std::vector<std::string> names;
  std::vector<int32_t> firsts, lasts;
  char contig[1024];
  long a = 0, b = 0;
  int status = GTX_OK;
  size_t n_lines = 0;
  while (std::fscanf(fp, "%1023s %ld %ld", contig, &a, &b) == 3)
  {
    // (:1076-1086: a line other than the first counts only when something follows its last number -- the stream is asked for its
    //  end before the interval is used, so the last line of a file that does not end in a newline is left out)
    int const behind = std::fgetc(fp);
    if (behind == EOF && n_lines > 0)
      break;
    if (behind != EOF)
      std::ungetc(behind, fp);
    ++n_lines;
    int32_t const first = static_cast<int32_t>(a - 1), last = static_cast<int32_t>(b - 1);
    if (!names.empty() && names.back() == contig)
    {
      if (first < firsts.back())
      {
        gtx::g_last_error = "The input intervals are not sorted.";
        status = GTX_ERR_ARG;
        break;
      }
      if (static_cast<long>(first) - lasts.back() <= 2l * par.max_frag_len)
      {
        lasts.back() = last; // (the reference takes the later interval's end, also when it is the smaller one)
        continue;
      }
    }
    names.emplace_back(contig);
    firsts.push_back(first);
    lasts.push_back(last);
  }
  std::fclose(fp);
  if (status != GTX_OK)
    return status;
  if (names.empty())
  {
    gtx::g_last_error = std::string("The interval file \"") + interval_file + "\" contained no intervals!";
    return GTX_ERR_ARG;
  }
  std::vector<char const *> chroms;
  for (auto const & n : names)
    chroms.push_back(n.c_str());
  return gtx_bam_shrink(bam_in, chroms.data(), firsts.data(), lasts.data(), static_cast<uint32_t>(names.size()), &par, bam_out, stats);
}
