// gtx_discover.hip -- first slice of variant discovery: the per-sample first pass over the reads of a
// region, run_first_pass (src/typer/caller.cpp:488-1186).
//
// Device: gtx_disc_events_kernel, one read per lane.  A read's CIGAR is walked against the region's reference held as bit
// planes (the layout of the alignment kernels' reads, graph_dev.hpp): an M block of 32 bases is four XORs and two one-hot
// tests, a mismatch of two unambiguous bases is a set bit, and every set bit is one SNP event; I and D operations give indel
// events when their bases are all A/C/G/T (one-hot over the range).  Events leave in the read's CIGAR order: every lane
// counts first, a wavefront claims one contiguous piece of the output with one atomic, the lanes write behind each other.
// Host: gtx_disc_first_pass keeps what is order-dependent in the reference -- which read sees an event first (span, the
// three distinct start positions), the correction for reads with 12 and more events, the phase counts between the events
// of a read -- by going over the reads in stream order, then applies the two support filters over the coverage arrays.
// gtx_disc_first_pass_haplotypes takes the pass to its end (the sample's haplotype map, :1186-1365), gtx_disc_merge puts the files'
// results together (merge_haplotypes2 :64-165, the union of the indels :2853-2903).
// (Next: the per-event sums as a device sort + segmented reduction; only the reads with >= 12 events need the order.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <limits>
#include <map>
#include <set>
#include <iterator>
#include <memory>
#include <string>
#include <vector>

#include "../../include/gtx.h"
#include "graph_dev.hpp"
#include "gtx_devmem.hpp"

namespace gtx
{
extern thread_local std::string g_last_error;
}

struct gtx_disc
{
  int device = -1;
  int64_t region_begin = 0;
  std::string reference; // region's bases as given (upper case letters)
  uint32_t * d_refp = nullptr;
  uint32_t ref_groups = 0;
};

namespace
{
using namespace gtx;

// 32 codes from bit offset `o` of a plane array of `groups` groups (zeros behind its end), plane b
__device__ inline uint32_t plane_bits(uint32_t const * planes, uint32_t groups, uint32_t o, uint32_t b)
{
  uint32_t const g = o >> 5, s = o & 31u;
  uint32_t const lo = g < groups ? planes[4 * g + b] : 0u, hi = g + 1 < groups ? planes[4 * (g + 1) + b] : 0u;
  return s == 0 ? lo : (lo >> s) | (hi << (32 - s));
}

struct Bits32
{
  uint32_t p0, p1, p2, p3;
  __device__ uint32_t onehot() const
  {
    uint32_t const odd = p0 ^ p1 ^ p2 ^ p3, three = (p0 & p1 & (p2 | p3)) | (p2 & p3 & (p0 | p1));
    return odd & ~three;
  }
};

__device__ inline Bits32 load32(uint32_t const * planes, uint32_t groups, uint32_t o)
{
  return Bits32{plane_bits(planes, groups, o, 0), plane_bits(planes, groups, o, 1), plane_bits(planes, groups, o, 2), plane_bits(planes, groups, o, 3)};
}

// all of the `n` bases from offset o are A / C / G / T
__device__ inline bool all_acgt(uint32_t const * planes, uint32_t groups, uint32_t o, uint32_t n)
{
  for (uint32_t k = 0; k < n; k += 32)
  {
    uint32_t const m = n - k >= 32 ? 0xFFFFFFFFu : (1u << (n - k)) - 1u;
    if ((load32(planes, groups, o + k).onehot() & m) != m)
      return false;
  }
  return true;
}

// One walk over a read's CIGAR (caller.cpp:583-775).  EMIT = false counts the events, EMIT = true writes them to out[0..).
// Returns the number of events; pos_end = region-relative end of the alignment (min(ref_offset, REF_SIZE - 1)).
template <bool EMIT>
__device__ uint32_t walk(uint32_t const * refp, uint32_t ref_groups, long REF_SIZE, long region_begin, uint32_t const * row, uint32_t row_groups,
                         uint8_t const * qual, gtx_disc_read const & r, uint32_t const * cigar, uint32_t read_index, gtx_disc_event * out, long & pos_end)
{
  uint32_t n = 0;
  long read_offset = 0, ref_offset = static_cast<long>(r.pos) - region_begin;
  long const l_qseq = r.l_qseq;
  auto put = [&](uint32_t pos, uint8_t type, uint16_t len, uint32_t seq, uint8_t hq, uint16_t dist)
  {
    if (EMIT)
      out[n] = gtx_disc_event{read_index, pos, seq, len, type, hq, dist, 0};
    ++n;
  };
  for (uint32_t i = 0; i < r.n_cigar; ++i)
  {
    uint32_t const word = cigar[i];
    long const count = word >> 4;
    uint32_t const op = word & 15u;
    if (ref_offset >= REF_SIZE)
      break;
    if (op == 0 || op == 7 || op == 8) // M = X
    {
      long const span = std::min<long>(count, std::min(REF_SIZE - ref_offset, std::max<long>(l_qseq - read_offset, 0)));
      for (long k = 0; k < span; k += 32)
      {
        uint32_t const m = span - k >= 32 ? 0xFFFFFFFFu : (1u << (span - k)) - 1u;
        Bits32 const a = load32(row, row_groups, static_cast<uint32_t>(read_offset + k)), g = load32(refp, ref_groups, static_cast<uint32_t>(ref_offset + k));
        uint32_t diff = ((a.p0 ^ g.p0) | (a.p1 ^ g.p1) | (a.p2 ^ g.p2) | (a.p3 ^ g.p3)) & a.onehot() & g.onehot() & m;
        while (diff)
        {
          uint32_t const j = static_cast<uint32_t>(__builtin_ctz(diff));
          diff &= diff - 1u;
          long const read_pos = read_offset + k + j;
          uint32_t const code = ((a.p0 >> j) & 1u) | (((a.p1 >> j) & 1u) << 1) | (((a.p2 >> j) & 1u) << 2) | (((a.p3 >> j) & 1u) << 3);
          char const base = code == 1 ? 'A' : code == 2 ? 'C' : code == 4 ? 'G' : 'T';
          long const dist = std::min(read_pos, l_qseq - 1 - read_pos);
          put(static_cast<uint32_t>(ref_offset + k + j + region_begin), 'X', 1, static_cast<uint32_t>(base), EMIT && qual[read_pos] >= 25 ? 1 : 0,
              static_cast<uint16_t>(std::min<long>(dist, 0xFFFF)));
        }
      }
      read_offset += count;
      ref_offset += count;
    }
    else if (op == 1) // I
    {
      long const b = std::min(read_offset, l_qseq), e = std::min(read_offset + count, l_qseq);
      if (b == e)
        continue; // (caller.cpp:698-699: the read offset stays)
      if (all_acgt(row, row_groups, static_cast<uint32_t>(b), static_cast<uint32_t>(e - b)))
        put(static_cast<uint32_t>(region_begin + ref_offset), 'I', static_cast<uint16_t>(e - b), static_cast<uint32_t>(b), 1, 0);
      read_offset += count;
    }
    else if (op == 2) // D
    {
      if (ref_offset + count < REF_SIZE && all_acgt(refp, ref_groups, static_cast<uint32_t>(ref_offset), static_cast<uint32_t>(count)))
        put(static_cast<uint32_t>(region_begin + ref_offset), 'D', static_cast<uint16_t>(std::min<long>(count, 0xFFFF)), static_cast<uint32_t>(ref_offset), 1, 0);
      ref_offset += count;
    }
    else if (op == 4) // S
      read_offset += count;
  }
  pos_end = std::min(ref_offset, REF_SIZE - 1);
  return n;
}

__global__ __launch_bounds__(256) void gtx_disc_events_kernel(uint32_t const * __restrict__ refp, uint32_t ref_groups, long REF_SIZE, long region_begin,
                                                              uint8_t const * __restrict__ rows, uint32_t plane_stride, uint8_t const * __restrict__ qual,
                                                              uint32_t qual_stride, gtx_disc_read const * __restrict__ reads,
                                                              uint32_t const * __restrict__ cigar, uint32_t n_reads, gtx_disc_event * __restrict__ events,
                                                              uint32_t event_cap, uint32_t * counts, gtx_disc_read_out * __restrict__ read_out)
{
  uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
  bool const live = i < n_reads;
  gtx_disc_read r{};
  uint32_t n = 0, state = GTX_DISC_SKIPPED;
  long pos_end = 0;
  uint32_t const * row = nullptr;
  if (live)
  {
    r = reads[i];
    row = reinterpret_cast<uint32_t const *>(rows + static_cast<uint64_t>(i) * plane_stride);
    // caller.cpp:517-561: reads without a cigar or in front of the region are passed over; a read that starts at or behind the
    // region's end ends the pass
    if (r.n_cigar != 0 && r.pos >= region_begin)
    {
      if (static_cast<long>(r.pos) - region_begin >= REF_SIZE)
        state = GTX_DISC_END;
      else
      {
        state = GTX_DISC_COUNTED;
        n = walk<false>(refp, ref_groups, REF_SIZE, region_begin, row, plane_stride / PLANE_GROUP_BYTES, qual + static_cast<uint64_t>(i) * qual_stride, r,
                        cigar + r.cigar_off, i, nullptr, pos_end);
      }
    }
  }
  // one contiguous piece of the output per wavefront
  uint32_t x = n;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
  {
    uint32_t const y = __shfl_up(x, d);
    if (lane >= static_cast<uint32_t>(d))
      x += y;
  }
  uint32_t const total = __shfl(x, 63);
  uint32_t base = 0;
  if (lane == 0 && total)
    base = atomicAdd(counts, total);
  base = __shfl(base, 0);
  uint32_t const first = base + x - n;
  if (live)
  {
    bool const fits = static_cast<uint64_t>(first) + n <= event_cap;
    if (n && fits)
      (void)walk<true>(refp, ref_groups, REF_SIZE, region_begin, row, plane_stride / PLANE_GROUP_BYTES, qual + static_cast<uint64_t>(i) * qual_stride, r,
                       cigar + r.cigar_off, i, events + first, pos_end);
    if (n && !fits)
      atomicAdd(counts + 1, n);
    read_out[i] = gtx_disc_read_out{first, n, static_cast<int32_t>(pos_end), state};
  }
}
} // namespace

extern "C" int gtx_disc_create(const char * reference, uint64_t reference_len, int64_t region_begin, int device, gtx_disc ** out)
{
  if (!reference || !out || reference_len == 0 || reference_len > 0x7FFFFFFFull)
  {
    g_last_error = "gtx_disc_create: bad argument";
    return GTX_ERR_ARG;
  }
  *out = nullptr;
  if (device == -1) // like gtx_ctx_create's -1: an object for the host stages only (the bookkeeping over events a device made elsewhere);
  {                 // gtx_disc_events_batch refuses it -- there is no CPU path for the walk over the CIGARs
    auto h = std::make_unique<gtx_disc>();
    h->device = -1;
    h->region_begin = region_begin;
    h->reference.assign(reference, reference_len);
    *out = h.release();
    return GTX_OK;
  }
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev)
  {
    g_last_error = "gtx_disc_create: no such HIP device (libgtx has no CPU path)";
    return GTX_ERR_NO_DEVICE;
  }
  auto d = std::make_unique<gtx_disc>();
  d->device = device;
  d->region_begin = region_begin;
  d->reference.assign(reference, reference_len);
  // the region as bit planes of BAM codes (anything but A / C / G / T: N)
  d->ref_groups = static_cast<uint32_t>((reference_len + 31) / 32) + 2;
  std::vector<uint32_t> planes(static_cast<size_t>(d->ref_groups) * 4, 0);
  for (uint64_t i = 0; i < reference_len; ++i)
  {
    char const c = reference[i];
    uint32_t const code = c == 'A' ? 1u : c == 'C' ? 2u : c == 'G' ? 4u : c == 'T' ? 8u : 15u;
    for (uint32_t b = 0; b < 4; ++b)
      planes[4 * (i >> 5) + b] |= ((code >> b) & 1u) << (i & 31u);
  }
  void * p = nullptr;
  if (hipSetDevice(device) != hipSuccess || gtx::dev_malloc(&p, planes.size() * 4) != hipSuccess ||
      hipMemcpy(p, planes.data(), planes.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
  {
    if (p)
      (void)gtx::dev_free(p);
    g_last_error = "gtx_disc_create: upload of the reference failed";
    return GTX_ERR_HIP;
  }
  d->d_refp = static_cast<uint32_t *>(p);
  *out = d.release();
  return GTX_OK;
}

extern "C" void gtx_disc_destroy(gtx_disc * d)
{
  if (!d)
    return;
  if (d->d_refp)
  {
    (void)hipSetDevice(d->device);
    (void)hipDeviceSynchronize();
    (void)gtx::dev_free(d->d_refp);
  }
  delete d;
}

extern "C" int gtx_disc_events_batch(gtx_disc * d, const uint8_t * d_planes, uint32_t plane_stride, const uint8_t * d_qual, uint32_t qual_stride,
                                     const gtx_disc_read * d_reads, const uint32_t * d_cigar, uint32_t n_reads, gtx_disc_event * d_events,
                                     uint32_t event_cap, uint32_t * d_counts, gtx_disc_read_out * d_read_out, void * stream)
{
  if (!d || plane_stride == 0 || (plane_stride % PLANE_GROUP_BYTES) != 0 || (reinterpret_cast<uintptr_t>(d_planes) & 3u) != 0 ||
      (n_reads != 0 && (!d_planes || !d_qual || !d_reads || !d_cigar || !d_counts || !d_read_out || (event_cap && !d_events))))
  {
    g_last_error = "gtx_disc_events_batch: bad argument";
    return GTX_ERR_ARG;
  }
  if (d->device < 0)
  {
    g_last_error = "gtx_disc_events_batch: the object was created without a device (libgtx has no CPU path)";
    return GTX_ERR_NO_DEVICE;
  }
  if (n_reads == 0)
    return GTX_OK;
  if (hipSetDevice(d->device) != hipSuccess)
    return GTX_ERR_HIP;
  hipLaunchKernelGGL(gtx_disc_events_kernel, dim3((n_reads + 255u) / 256u), dim3(256), 0, static_cast<hipStream_t>(stream), d->d_refp, d->ref_groups,
                     static_cast<long>(d->reference.size()), static_cast<long>(d->region_begin), d_planes, plane_stride, d_qual, qual_stride, d_reads,
                     d_cigar, n_reads, d_events, event_cap, d_counts, d_read_out);
  if (hipGetLastError() != hipSuccess)
  {
    g_last_error = "gtx_disc_events_kernel launch failed";
    return GTX_ERR_HIP;
  }
  return GTX_OK;
}

// ---- host: the order-dependent bookkeeping and the filters --------------------------------------------------------------
namespace
{
struct Ev // Event (include/graphtyper/typer/event.hpp:30-73) with its ordering (src/typer/event.cpp:198-207)
{
  uint32_t pos;
  uint8_t type;
  std::string seq;
  bool operator<(Ev const & o) const
  {
    int const a = (type == 'D') + 2 * (type == 'X'), b = (o.type == 'D') + 2 * (o.type == 'X');
    if (pos != o.pos)
      return pos < o.pos;
    if (a != b)
      return a < b;
    return seq < o.seq;
  }
};

struct Support // EventSupport (event.hpp:75-113): what the first pass fills
{
  uint32_t hq = 0, lq = 0, proper = 0, first = 0, reversed = 0, clipped = 0;
  uint8_t max_mapq = 0, max_distance = 0;
  int32_t u1 = -1, u2 = -1, u3 = -1;
  uint16_t span = 1;
  bool realign = false, good = false;
  uint32_t max_log_qual = 0;
  int32_t file_i = 0; // max_log_qual_file_i: the file whose reads gave max_log_qual
  std::map<Ev, uint16_t> phase;
};

struct PassState // what run_first_pass has when its two filters are through
{
  std::vector<std::map<Ev, Support>> buckets;
  std::vector<uint32_t> up, down; // cov_up / cov_down
  long REF = 0, B = 0, begin = 0;
};

uint16_t wrap16(uint32_t v) { return static_cast<uint16_t>(v); } // (the reference's counters are uint16_t and wrap)

bool good_snp(Support const & s, long cov) // EventSupport::has_good_support with the default Options (event.cpp:226-256)
{
  cov = std::max(cov, 1l);
  int const hq = wrap16(s.hq), raw = wrap16(s.hq) + wrap16(s.lq), pp = wrap16(s.proper), fip = wrap16(s.first), rev = wrap16(s.reversed), cl = wrap16(s.clipped);
  double const ratio = static_cast<double>(raw) / static_cast<double>(cov);
  bool const very = s.u3 != -1 && ((hq >= 8 && ratio >= 0.35) || (hq >= 7 && ratio >= 0.40)) && pp >= 6;
  bool const prom = s.u3 != -1 && ((hq >= 7 && ratio >= 0.20) || (hq >= 6 && ratio >= 0.30) || (hq >= 5 && ratio >= 0.40)) && pp >= 4;
  return s.u2 != -1 && pp >= 2 && hq >= 3 && (prom || (fip > 0 && fip < raw)) && (very || (prom && rev > 0 && rev < raw) || (rev > 1 && rev < raw - 1)) &&
         (cl <= 1 || cl + 5 <= raw) && (s.max_distance >= 10 || (prom && hq >= 10)) && (hq + (raw - hq) / 2.0) >= 3.9 && (ratio > 0.26 || prom);
}
} // namespace

// run_first_pass up to and including its two support filters (caller.cpp:488-1186) from the device's events
static int first_pass_state(const gtx_disc * d, const gtx_disc_read * reads, const uint32_t * cigar, const gtx_disc_read_out * read_out, uint32_t n_reads,
                            const gtx_disc_event * events, uint64_t n_events, const uint8_t * seq, uint32_t seq_stride, uint32_t bucket_size,
                            int32_t file_index, PassState & state)
{
  std::string const & ref = d->reference;
  long const REF = static_cast<long>(ref.size()), B = bucket_size, begin = d->region_begin;
  state.REF = REF;
  state.B = B;
  state.begin = begin;
  std::vector<std::map<Ev, Support>> & buckets = state.buckets;
  state.up.assign(REF, 0);
  state.down.assign(REF, 0);
  std::vector<uint32_t> &up = state.up, &down = state.down;
  static char const NT16[] = "=ACMGRSVTWYHKDBN";
  auto bucket_of = [&](uint32_t pos) -> std::map<Ev, Support> &
  {
    size_t const b = static_cast<size_t>((static_cast<long>(pos) - begin) / B);
    if (b >= buckets.size())
      buckets.resize(b + 1);
    return buckets[b];
  };
  std::vector<std::map<Ev, Support>::iterator> mine;
  for (uint32_t i = 0; i < n_reads; ++i)
  {
    gtx_disc_read const & r = reads[i];
    gtx_disc_read_out const & ro = read_out[i];
    if (ro.state == GTX_DISC_SKIPPED)
      continue;
    size_t const start_bucket = static_cast<size_t>((static_cast<long>(r.pos) - begin) / B);
    if (start_bucket >= buckets.size())
      buckets.resize(start_bucket + 1); // (caller.cpp:546-548: before the end-of-region test)
    if (ro.state == GTX_DISC_END)
      break;
    if (static_cast<uint64_t>(ro.first_event) + ro.n_events > n_events)
    {
      g_last_error = "gtx_disc_first_pass: a read's events lie behind the event buffer (it overflowed: event_cap too small)";
      return GTX_ERR_CAPACITY;
    }
    uint32_t const front = cigar[r.cigar_off], back = cigar[r.cigar_off + r.n_cigar - 1];
    bool const clipped = ((front & 15u) == 4 && (front >> 4) >= 1) || ((back & 15u) == 4 && (back >> 4) >= 1); // is_clipped (caller.cpp:167-196)
    mine.clear();
    for (uint32_t k = 0; k < ro.n_events; ++k)
    {
      gtx_disc_event const & e = events[ro.first_event + k];
      Ev ev{e.pos, e.type, {}};
      long const ref_offset = static_cast<long>(e.pos) - begin;
      if (e.type == 'X')
        ev.seq.assign(1, static_cast<char>(e.seq));
      else if (e.type == 'I')
      {
        uint8_t const * row = seq + static_cast<uint64_t>(i) * seq_stride;
        for (uint32_t j = 0; j < e.len; ++j)
        {
          uint32_t const at = e.seq + j;
          ev.seq.push_back(NT16[(row[at >> 1] >> ((~at & 1u) << 2)) & 15u]);
        }
      }
      else
        ev.seq = ref.substr(e.seq, e.len);
      auto ins = bucket_of(e.pos).insert({std::move(ev), Support()});
      Support & s = ins.first->second;
      if (ins.second && e.type != 'X') // span of a new indel (bucket.cpp:100-160)
      {
        std::string const & q = ins.first->first.seq;
        long span = 0, count = static_cast<long>(q.size());
        if (e.type == 'I')
        {
          while (span < count && ref_offset + span < REF && q[span] == ref[ref_offset + span])
            ++span;
          if (span == count)
            while (ref_offset + span < REF && ref[ref_offset + span - count] == ref[ref_offset + span])
              ++span;
        }
        else
          while (ref_offset + span < REF && ref_offset + span + count < REF && ref[ref_offset + span] == ref[ref_offset + span + count])
            ++span;
        s.span = static_cast<uint16_t>(std::min<long>(span, std::numeric_limits<uint16_t>::max() - 1) + 1); // (bucket.cpp:128-131, 156-159)
      }
      if (e.type == 'X')
      {
        if (e.hq)
          ++s.hq;
        else
          ++s.lq;
        s.first += (r.flag & 64u) != 0;
        if (s.u1 == -1)
          s.u1 = r.pos;
        else if (s.u2 == -1)
        {
          if (s.u1 != r.pos)
            s.u2 = r.pos;
        }
        else if (s.u3 == -1 && s.u2 != r.pos)
          s.u3 = r.pos;
        if (static_cast<long>(e.max_distance) > static_cast<long>(s.max_distance))
          s.max_distance = static_cast<uint8_t>(e.max_distance);
      }
      else
        ++s.hq;
      if (r.mapq != 255 && r.mapq > s.max_mapq)
        s.max_mapq = r.mapq;
      s.proper += (r.flag & 2u) != 0;
      s.reversed += (r.flag & 16u) != 0;
      s.clipped += clipped;
      mine.push_back(ins.first);
    }
    // reads with many events (caller.cpp:777-822)
    if (mine.size() >= 12)
      for (auto & it : mine)
      {
        Support & s = it->second;
        if (mine.size() >= 18)
        {
          if (wrap16(s.hq) > 0)
            --s.hq;
          else if (wrap16(s.lq) > 0)
            --s.lq;
        }
        else if (wrap16(s.hq) > 0)
        {
          --s.hq;
          ++s.lq;
        }
      }
    if (mine.size() < 18)
      for (size_t b2 = 1; b2 < mine.size(); ++b2)
        for (size_t a = 0; a < b2; ++a)
          ++mine[a]->second.phase.insert({mine[b2]->first, 0}).first->second;
    ++up[static_cast<long>(r.pos) - begin];
    ++down[ro.pos_end];
  }
  if ((static_cast<long>(buckets.size()) - 1) * B >= REF)
    buckets.resize((REF - 1) / B + 1);
  long const NB = static_cast<long>(buckets.size());
  auto delta = [&](long o) { return static_cast<long>(up[o]) - static_cast<long>(down[o]); };
  // SNPs with low support (caller.cpp:897-985)
  {
    long depth = 0;
    for (long b = 0; b < NB; ++b)
    {
      for (auto it = buckets[b].begin(); it != buckets[b].end();)
      {
        if (it->first.type != 'X')
        {
          ++it;
          continue;
        }
        long cov = depth;
        long const at = std::max(0l, static_cast<long>(it->first.pos) - begin);
        if (at + 1 > b * B)
          for (long o = b * B; o <= at; ++o)
            cov += delta(o);
        if (good_snp(it->second, cov))
          ++it;
        else
          it = buckets[b].erase(it);
      }
      if (b * B >= REF)
        break;
      for (long o = b * B, e = std::min(REF, (b + 1) * B); o < e; ++o)
        depth += delta(o);
    }
  }
  // indels: good support, worth a realignment, or dropped (caller.cpp:990-1186)
  long depth = 0;
  for (long b = 0; b < NB; ++b)
  {
    for (auto it = buckets[b].begin(); it != buckets[b].end();)
    {
      if (it->first.type == 'X')
      {
        ++it;
        continue;
      }
      Support & s = it->second;
      double const len = static_cast<double>(it->first.seq.size());
      long const pad = static_cast<long>(4.0 + len / 3.0), pos = static_cast<long>(it->first.pos);
      long const lo = std::max(0l, pos - pad - begin), hi = std::min(REF, pos + s.span + pad - begin);
      double const count = (it->first.type == 'I' ? (len / 2.0 + 8.0) / 8.0 : (len / 3.0 + 10.0) / 10.0) * (wrap16(s.hq) + wrap16(s.lq));
      long cov = depth, o = lo;
      if (o <= b * B)
        for (; o < b * B; ++o)
          cov -= delta(o);
      else
        for (o = b * B; o < lo; ++o)
          cov += delta(o);
      for (; o <= hi; ++o)
        cov -= o < REF ? static_cast<long>(down[o]) : 0l;
      double const corrected = std::max(static_cast<double>(cov), count), anti = corrected - count;
      double const gt00 = count * 10.0, gt_alt = std::min(count + anti, anti * 10.0); // get_log_qual_double (event.cpp:102-113)
      uint32_t const log_qual = gt00 > gt_alt ? static_cast<uint32_t>(gt00 - gt_alt + 0.5) : 0u;
      int const hq = wrap16(s.hq), rev = wrap16(s.reversed), pp = wrap16(s.proper), cl = wrap16(s.clipped);
      if (hq >= 6 && count >= 8.0 && log_qual >= 60 && rev > 0 && rev < hq && pp >= 3 && s.max_mapq >= 20 && (cl == 0 || cl + 3 <= hq))
      {
        s.good = s.realign = true;
        s.max_log_qual = log_qual;
        s.file_i = file_index;
        ++it;
      }
      else if (count >= 3.0 && log_qual > 0 && pp >= 1 && (hq >= 5 || s.max_mapq >= 25) && s.max_mapq >= 10 && cl < hq)
      {
        s.realign = true;
        s.max_log_qual = log_qual;
        s.file_i = file_index;
        ++it;
      }
      else
        it = buckets[b].erase(it);
    }
    if (b * B >= REF)
      break;
    for (long o = b * B, e = std::min(REF, (b + 1) * B); o < e; ++o)
      depth += delta(o);
  }
  return GTX_OK;
}

namespace
{
void put_ev(std::vector<uint32_t> & w, Ev const & e)
{
  w.push_back(e.pos);
  w.push_back(e.type);
  w.push_back(static_cast<uint32_t>(e.seq.size()));
  for (char c : e.seq)
    w.push_back(static_cast<uint32_t>(static_cast<unsigned char>(c)));
}

// an event with its support: the fields, (the file of the best support,) the phase entries
void put_support(std::vector<uint32_t> & w, Ev const & e, Support const & s, bool with_file)
{
  put_ev(w, e);
  for (uint32_t v : {uint32_t(wrap16(s.hq)), uint32_t(wrap16(s.lq)), uint32_t(wrap16(s.proper)), uint32_t(wrap16(s.first)), uint32_t(wrap16(s.reversed)),
                     uint32_t(wrap16(s.clipped)), uint32_t(s.max_mapq), uint32_t(s.max_distance), uint32_t(s.u1), uint32_t(s.u2), uint32_t(s.u3),
                     uint32_t(s.span), uint32_t(s.realign), uint32_t(s.good), s.max_log_qual})
    w.push_back(v);
  if (with_file)
    w.push_back(static_cast<uint32_t>(s.file_i));
  w.push_back(static_cast<uint32_t>(s.phase.size()));
  for (auto const & ph : s.phase)
  {
    put_ev(w, ph.first);
    w.push_back(ph.second);
  }
}

int hand_over(std::vector<uint32_t> const & w, uint32_t * out, uint64_t cap, uint64_t * n_words)
{
  *n_words = w.size();
  if (w.size() <= cap && !w.empty())
    std::memcpy(out, w.data(), w.size() * 4);
  return w.size() <= cap ? GTX_OK : GTX_ERR_CAPACITY;
}

// HaplotypeInfo (caller.cpp:45-52): the events an event is seen with -- in some sample, in every sample that has it
struct Together
{
  std::set<Ev> ever, always;
};

struct FileResult // what a file (or several, merged) leaves behind: Tindel_events and the haplotype map
{
  std::map<Ev, Support> indels;
  std::map<Ev, Together> haplotypes;
};

void put_result(std::vector<uint32_t> & w, FileResult const & r)
{
  w.push_back(static_cast<uint32_t>(r.indels.size()));
  for (auto const & kv : r.indels)
    put_support(w, kv.first, kv.second, true);
  w.push_back(static_cast<uint32_t>(r.haplotypes.size()));
  for (auto const & kv : r.haplotypes)
  {
    put_ev(w, kv.first);
    for (std::set<Ev> const * set : {&kv.second.ever, &kv.second.always})
    {
      w.push_back(static_cast<uint32_t>(set->size()));
      for (Ev const & e : *set)
        put_ev(w, e);
    }
  }
}

bool read_result(uint32_t const * w, uint64_t n, FileResult & r)
{
  uint64_t at = 0;
  bool ok = true;
  auto word = [&]() -> uint32_t
  {
    if (at >= n)
    {
      ok = false;
      return 0;
    }
    return w[at++];
  };
  auto event = [&]()
  {
    Ev e;
    e.pos = word();
    e.type = static_cast<uint8_t>(word());
    uint32_t const len = word();
    if (!ok || len > n - at)
    {
      ok = false;
      return e;
    }
    for (uint32_t k = 0; k < len; ++k)
      e.seq.push_back(static_cast<char>(w[at + k]));
    at += len;
    return e;
  };
  if (n == 0)
    return true;
  for (uint32_t i = 0, m = word(); ok && i < m; ++i)
  {
    Ev e = event();
    Support s;
    s.hq = word(); s.lq = word(); s.proper = word(); s.first = word(); s.reversed = word(); s.clipped = word();
    s.max_mapq = static_cast<uint8_t>(word()); s.max_distance = static_cast<uint8_t>(word());
    s.u1 = static_cast<int32_t>(word()); s.u2 = static_cast<int32_t>(word()); s.u3 = static_cast<int32_t>(word());
    s.span = static_cast<uint16_t>(word()); s.realign = word() != 0; s.good = word() != 0; s.max_log_qual = word();
    s.file_i = static_cast<int32_t>(word());
    for (uint32_t k = 0, np = word(); ok && k < np; ++k)
    {
      Ev pe = event();
      s.phase[pe] = static_cast<uint16_t>(word());
    }
    r.indels.insert({std::move(e), std::move(s)});
  }
  for (uint32_t i = 0, m = word(); ok && i < m; ++i)
  {
    Ev e = event();
    Together t;
    for (std::set<Ev> * set : {&t.ever, &t.always})
      for (uint32_t k = 0, ns = word(); ok && k < ns; ++k)
        set->insert(event());
    r.haplotypes.insert({std::move(e), std::move(t)});
  }
  return ok && at == n;
}
} // namespace

extern "C" int gtx_disc_first_pass(const gtx_disc * d, const gtx_disc_read * reads, const uint32_t * cigar, const gtx_disc_read_out * read_out,
                                   uint32_t n_reads, const gtx_disc_event * events, uint64_t n_events, const uint8_t * seq, uint32_t seq_stride,
                                   uint32_t bucket_size, uint32_t * out, uint64_t cap, uint64_t * n_words)
{
  if (!d || !n_words || bucket_size == 0 || (n_reads && (!reads || !cigar || !read_out || !seq)) || (n_events && !events) || (cap && !out))
  {
    g_last_error = "gtx_disc_first_pass: bad argument";
    return GTX_ERR_ARG;
  }
  PassState st;
  int const rc = first_pass_state(d, reads, cigar, read_out, n_reads, events, n_events, seq, seq_stride, bucket_size, 0, st);
  if (rc != GTX_OK)
    return rc;
  // the surviving events as a word stream: pos, type, length, characters, the support fields, the phase entries
  std::vector<uint32_t> w;
  for (auto const & bucket : st.buckets)
    for (auto const & kv : bucket)
      put_support(w, kv.first, kv.second, false);
  return hand_over(w, out, cap, n_words);
}

// run_first_pass to its end (caller.cpp:1186-1365): for every event that is left, which later events within two buckets it
// travels with -- "ever": in enough of the reads that cover both (by its phase counts and the coverage between the two; any
// shared read when one of them is an indel), "always": those of them at most ten positions on -- the sample's haplotype map;
// the SNPs then leave the buckets.  Output: the file's result (put_result: indels with their support, the haplotype map).
extern "C" int gtx_disc_first_pass_haplotypes(const gtx_disc * d, const gtx_disc_read * reads, const uint32_t * cigar, const gtx_disc_read_out * read_out,
                                              uint32_t n_reads, const gtx_disc_event * events, uint64_t n_events, const uint8_t * seq, uint32_t seq_stride,
                                              uint32_t bucket_size, int32_t file_index, uint32_t * out, uint64_t cap, uint64_t * n_words)
{
  if (!d || !n_words || bucket_size == 0 || (n_reads && (!reads || !cigar || !read_out || !seq)) || (n_events && !events) || (cap && !out))
  {
    g_last_error = "gtx_disc_first_pass_haplotypes: bad argument";
    return GTX_ERR_ARG;
  }
  PassState st;
  int const rc = first_pass_state(d, reads, cigar, read_out, n_reads, events, n_events, seq, seq_stride, bucket_size, file_index, st);
  if (rc != GTX_OK)
    return rc;
  long const REF = st.REF, B = st.B, begin = st.begin, NB = static_cast<long>(st.buckets.size());
  auto delta = [&](long o) { return static_cast<long>(st.up[o]) - static_cast<long>(st.down[o]); };
  FileResult res;
  long depth = 0;
  for (long b = 0; b < NB; ++b)
  {
    auto & bucket = st.buckets[b];
    for (auto it = bucket.begin(); it != bucket.end();)
    {
      Ev const & ev = it->first;
      Support const & info = it->second;
      long const at = std::max(0l, static_cast<long>(ev.pos) - begin);
      long cov = depth;
      if (at + 1 > b * B)
        for (long o = b * B; o <= at; ++o)
          cov += delta(o);
      Together & tg = res.haplotypes.insert({ev, Together()}).first->second;
      double ratio = static_cast<double>(wrap16(info.hq) + wrap16(info.lq)) / static_cast<double>(cov);
      if (ratio < 0.3)
        ratio = 0.3;
      // 1: seen together, 2: seen apart (caller.cpp:1216-1268; 0: too little coverage to say)
      auto judge = [&](Ev const & other) -> int
      {
        auto const ph = info.phase.find(other);
        if (ev.type != 'X' || other.type != 'X')
          return (ph == info.phase.end() || ph->second == 0) ? 2 : 3;
        long local = cov;
        for (long o = at + 1, last = std::max(0l, static_cast<long>(other.pos) - begin); o <= last; ++o)
          local -= o < REF ? static_cast<long>(st.down[o]) : 0l;
        if (local <= 2)
          return 0;
        double const support = ph == info.phase.end() ? 0.0 : ph->second;
        if ((support / static_cast<double>(local) / ratio) < 0.22)
          return 2;
        if ((support / static_cast<double>(local) / ratio) > 0.78)
          return 1;
        return 3;
      };
      auto look = [&](Ev const & other, bool may_be_always)
      {
        if (judge(other) & 1)
        {
          tg.ever.insert(other);
          if (may_be_always && other.pos <= ev.pos + 10)
            tg.always.insert(other);
        }
      };
      for (auto it2 = std::next(it); it2 != bucket.end(); ++it2)
        if (!(it2->first.pos == ev.pos && it2->first.type == ev.type)) // (alleles of one place exclude each other)
          look(it2->first, true);
      if (b + 1 < NB)
        for (auto const & kv : st.buckets[b + 1])
          look(kv.first, true);
      if (b + 2 < NB)
        for (auto const & kv : st.buckets[b + 2])
        {
          if (kv.first.pos >= ev.pos + 2 * B)
            break;
          look(kv.first, false);
        }
      if (ev.type == 'X')
        it = bucket.erase(it);
      else
        ++it;
    }
    if (b * B >= REF)
      break;
    for (long o = b * B, e = std::min(REF, (b + 1) * B); o < e; ++o)
      depth += delta(o);
  }
  for (auto & bucket : st.buckets)
    for (auto & kv : bucket)
      res.indels.insert(kv);
  std::vector<uint32_t> w;
  put_result(w, res);
  return hand_over(w, out, cap, n_words);
}

// The results of two files (or of files merged before) as one: merge_haplotypes2 (caller.cpp:64-165) -- an event new to `into`
// keeps of its "always" set what `into` has never seen; one known to both has the union of the "ever" sets and the
// intersection of the "always" sets -- and the union of the indels (streamlined_discovery, :2853-2903: good support from any
// file, the best max_log_qual with its file).  `into` may be empty.  Files are merged in their order.
extern "C" int gtx_disc_merge(const uint32_t * into, uint64_t n_into, const uint32_t * from, uint64_t n_from, uint32_t * out, uint64_t cap,
                              uint64_t * n_words)
{
  if (!n_words || (n_into && !into) || (n_from && !from) || (cap && !out))
  {
    g_last_error = "gtx_disc_merge: bad argument";
    return GTX_ERR_ARG;
  }
  FileResult a, b;
  if (!read_result(into, n_into, a) || !read_result(from, n_from, b))
  {
    g_last_error = "gtx_disc_merge: not a result of gtx_disc_first_pass_haplotypes / gtx_disc_merge";
    return GTX_ERR_ARG;
  }
  if (a.haplotypes.empty())
    a.haplotypes = std::move(b.haplotypes);
  else
    for (auto & kv : b.haplotypes)
    {
      auto ins = a.haplotypes.insert(kv);
      Together & mine = ins.first->second;
      if (ins.second)
      {
        for (auto it = mine.always.begin(); it != mine.always.end();)
          it = a.haplotypes.count(*it) ? mine.always.erase(it) : std::next(it);
        continue;
      }
      mine.ever.insert(kv.second.ever.begin(), kv.second.ever.end());
      std::set<Ev> both;
      std::set_intersection(mine.always.begin(), mine.always.end(), kv.second.always.begin(), kv.second.always.end(), std::inserter(both, both.begin()));
      mine.always.swap(both);
    }
  for (auto & kv : b.indels)
  {
    auto ins = a.indels.insert(kv);
    if (ins.second)
      continue;
    Support & old = ins.first->second;
    old.good = old.good || kv.second.good;
    if (kv.second.max_log_qual > old.max_log_qual)
    {
      old.max_log_qual = kv.second.max_log_qual;
      old.file_i = kv.second.file_i;
    }
  }
  std::vector<uint32_t> w;
  put_result(w, a);
  return hand_over(w, out, cap, n_words);
}
