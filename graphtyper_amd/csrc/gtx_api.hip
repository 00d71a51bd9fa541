// gtx_api.hip -- the HIP kernels (gfx950) and the device half of the C ABI declared in include/gtx.h.
//
// align kernel : one wavefront (= one 64-thread workgroup, so the workgroup barrier is the wave barrier and every
//                wave owns its LDS workspace) per (read, orientation) task; persistent grid-stride over tasks.
// score kernel : one thread per score item (an unpaired read or a mate pair); integer atomics into flat accumulators.
#include <hip/hip_runtime.h>
#include <chrono>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "gtx_ctx.hpp"

#include <map>
#include "gtx_devmem.hpp"
#include "wave_hip.hpp"
#include "align_core.hpp"
#include "gtx_hbm_passes.hpp"
#include "score_core.hpp"
#include "score_replay.hpp"

namespace gtx
{
// profile counters (GraphView::prof): 32 sums, then -- profiling build -- the count and the entries of the general pass' task log
#ifdef GTX_PROF
constexpr uint64_t PROF_LOG_ENTRIES = 1u << 16, PROF_LOG_ENTRY = 16;
#else
constexpr uint64_t PROF_LOG_ENTRIES = 0, PROF_LOG_ENTRY = 16;
#endif
constexpr uint64_t PROF_WORDS = 40 + PROF_LOG_ENTRIES * PROF_LOG_ENTRY;
#ifndef GTX_BIG_BLOCKS_PER_CU
#define GTX_BIG_BLOCKS_PER_CU 8u // workgroups (one wavefront, one workspace of 0.7 MB in HBM each) of the HBM-table pass per CU: 1.4 GB per call scratch
#endif
#ifndef GTX_EXACT_PARTS_MAX
#define GTX_EXACT_PARTS_MAX 1024u // the most parts the exact pass' first launch cuts its slab into
#endif

// Scoring adds small integers to per-(haplotype, sample) counters, and the reads of a workgroup -- neighbours in a
// position-sorted stream -- hit the same few counters: 1 500 reads deep, every counter of a site would take thousands of
// same-address atomics at the L2.  The workgroup therefore sums into an LDS table keyed by the counter's address first and
// sends one atomic per (counter, workgroup visit) to memory.  Sums are order-free, so the accumulators come out the same.
#ifndef GTX_SCORE_THREADS
#define GTX_SCORE_THREADS 64 // threads of a gtx_score_kernel workgroup: one wavefront goes wherever a slot is beside another batch's queues (cfg2, steps in flight: 256 / 128 / 64 threads = 0.798 / 0.788 / 0.783 ms per step; alone the same)
#endif
#ifdef GTX_SCORE_ADDS_CALLED // (A/B build: one body for all of an item's adds, a call each)
#define GTX_SCORE_ADD_INLINE __attribute__((noinline))
#else
#define GTX_SCORE_ADD_INLINE inline
#endif
#ifndef GTX_WAVE_GROUP_MIN
#define GTX_WAVE_GROUP_MIN 4
#endif
#ifndef GTX_SCORE_TABLE_LOG2
#define GTX_SCORE_TABLE_LOG2 8
#endif
#ifndef GTX_SCORE_CHUNK_MAX
#define GTX_SCORE_CHUNK_MAX 64 // items of one workgroup between two flushes of its table; at most 1 024
#endif
// What one add may bring to a 32-bit sum of the table: between two flushes a slot sees at most GTX_SCORE_CHUNK_MAX items, an item
// adds to one counter at most SCORE_ADDS_PER_ITEM times (two reads, a read once or twice -- the bound has room for eight), so sums
// of addends below this cannot wrap; a larger addend (a wavefront group's sum of squares, say) goes straight to memory.
constexpr uint32_t SCORE_ADDS_PER_ITEM = 8;
static_assert(GTX_SCORE_CHUNK_MAX >= 1 && GTX_SCORE_CHUNK_MAX <= 1024, "GTX_SCORE_CHUNK_MAX: 1 .. 1 024");
constexpr unsigned long long SCORE_COMBINE_ADDEND_LIMIT = (1ull << 32) / (static_cast<unsigned long long>(GTX_SCORE_CHUNK_MAX) * SCORE_ADDS_PER_ITEM);
// The table: 32-bit keys -- the counter's word offset from the lowest accumulator address (gtx_scores_alloc puts them in one block),
// bit 31 = the counter is 64 bits wide -- and 32-bit sums, 8 KB for 1 024 entries.  A workgroup scores a CONTIGUOUS run of the work
// queue (neighbours in the stream: the same two or three sites, the same samples over and over) and flushes once at its end:
// round 4 flushed after every 64 items, where 30 samples leave two items per counter.
struct ScoreCombiner
{
  static constexpr uint32_t LOG2N = GTX_SCORE_TABLE_LOG2, N = 1u << LOG2N, PROBES = 8, EMPTY = 0xFFFFFFFFu;
  uint32_t key[N];
  uint32_t val[N];
  unsigned long long base; // the lowest accumulator address
};

struct WaveHipCombine : WaveHip
{
  static __device__ inline ScoreCombiner & table()
  {
    __shared__ ScoreCombiner t;
    return t;
  }
  static __device__ inline bool combine(unsigned long long address, bool is64, unsigned long long v)
  {
    ScoreCombiner & t = table();
    unsigned long long const d = address - t.base;
    if ((d >> 33) != 0ull || v >= SCORE_COMBINE_ADDEND_LIMIT) // (not within 8 GB above the base, or an addend the 32-bit sum has no room for: straight to memory)
      return false;
    uint32_t const key = static_cast<uint32_t>(d >> 2) | (is64 ? 0x80000000u : 0u);
    uint32_t h = (key * 0x9E3779B1u) >> (32u - ScoreCombiner::LOG2N);
    for (uint32_t probe = 0; probe < ScoreCombiner::PROBES; ++probe)
    {
      uint32_t const old = atomicCAS(&t.key[h], ScoreCombiner::EMPTY, key);
      if (old == ScoreCombiner::EMPTY || old == key)
      {
        atomicAdd(&t.val[h], static_cast<uint32_t>(v));
        return true;
      }
      h = (h + 1u) & (ScoreCombiner::N - 1);
    }
    return false; // crowded (many samples in one run): straight to memory
  }
  // The lanes that are here together with the same (counter, addend) -- neighbours in the stream see the same site with the
  // same alleles and the same epsilon -- send ONE add of addend x lanes through their first lane: 64 same-address LDS
  // atomics take the LDS unit 64 turns each (a third of the kernel's time per CU before this), the ballots take none.
  // Returns true for the lane that has to add *sum (false: another lane does it).
  static __device__ inline bool wave_group(unsigned long long key, unsigned long long v, unsigned long long & sum)
  {
#ifdef GTX_NO_WAVE_GROUP // (A/B build: every lane for itself)
    sum = v;
    return true;
#else
    // (groups are formed while they are worth it: one sample deep, a wavefront's lanes make two or three large ones; thirty samples
    //  wide, nearly every lane has a counter of its own, and forming 64 groups of one -- a round of readfirstlanes and a ballot
    //  each, for every one of an item's fifteen adds -- was HALF of the kernel on cfg3.  The first group of fewer than
    //  GTX_WAVE_GROUP_MIN lanes is the last one: the lanes left over add for themselves)
    uint32_t const lane = threadIdx.x & 63u;
    for (;;)
    {
      uint32_t const k_lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(key)), k_hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(key >> 32));
      uint32_t const v_lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v)), v_hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
      bool const same = key == ((static_cast<unsigned long long>(k_hi) << 32) | k_lo) && v == ((static_cast<unsigned long long>(v_hi) << 32) | v_lo);
      unsigned long long const group = __ballot(same);
      uint32_t const members = static_cast<uint32_t>(__builtin_popcountll(group));
      if (same)
      {
        sum = v * static_cast<unsigned long long>(members);
        return lane == static_cast<uint32_t>(__builtin_ctzll(group));
      }
      if (members < GTX_WAVE_GROUP_MIN)
        break;
    }
    sum = v;
    return true;
#endif
  }
#if defined(GTX_EXPERIMENT) && defined(GTX_X_SCORE_NO_ADDS) // (experiment builds only, wrong results: what the kernel takes without its adds)
  static __device__ inline void atomic_add_u32(uint32_t *, uint32_t) {}
  static __device__ inline void atomic_add_u64(unsigned long long *, unsigned long long) {}
#else
  // (inlined, the fifteen adds of a site times the four places an item's entries are applied make the kernel 131 KB of code, twice
  //  the instruction cache two CUs share; as ONE body and a call each -- GTX_SCORE_ADDS_CALLED -- it is 49 KB and SLOWER: cfg3
  //  0.687 ms against 0.634, cfg2 0.160 against 0.146.  The instruction cache is not what the kernel waits for.)
  static __device__ GTX_SCORE_ADD_INLINE void add_to(unsigned long long address, bool is64, unsigned long long v)
  {
    unsigned long long sum;
    if (wave_group(address, v, sum) && !combine(address, is64, sum))
    {
      if (is64)
        atomicAdd(reinterpret_cast<unsigned long long *>(address), sum);
      else
        atomicAdd(reinterpret_cast<uint32_t *>(address), static_cast<uint32_t>(sum));
    }
  }
  static __device__ inline void atomic_add_u32(uint32_t * p, uint32_t v) { add_to(reinterpret_cast<unsigned long long>(p), false, v); }
  static __device__ inline void atomic_add_u64(unsigned long long * p, unsigned long long v) { add_to(reinterpret_cast<unsigned long long>(p), true, v); }
#endif
  // all threads of the workgroup
  static __device__ inline void clear(unsigned long long base)
  {
    ScoreCombiner & t = table();
    for (uint32_t i = threadIdx.x; i < ScoreCombiner::N; i += blockDim.x)
    {
      t.key[i] = ScoreCombiner::EMPTY;
      t.val[i] = 0;
    }
    if (threadIdx.x == 0)
      t.base = base;
    __syncthreads();
  }
  static __device__ inline void flush()
  {
    ScoreCombiner & t = table();
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < ScoreCombiner::N; i += blockDim.x)
    {
      uint32_t const k = t.key[i];
      if (k != ScoreCombiner::EMPTY)
      {
        uint32_t const v = t.val[i];
        unsigned long long const address = t.base + (static_cast<unsigned long long>(k & 0x7FFFFFFFu) << 2);
#if defined(GTX_EXPERIMENT) && defined(GTX_X_SCORE_NO_FLUSH) // (experiment builds only, wrong results: the table is summed into, nothing reaches memory)
        if (v == 0xFFFFFFFFu && address == 1ull)
#endif
        if (k & 0x80000000u)
          atomicAdd(reinterpret_cast<unsigned long long *>(address), static_cast<unsigned long long>(v));
        else
          atomicAdd(reinterpret_cast<uint32_t *>(address), v);
        t.key[i] = ScoreCombiner::EMPTY;
        t.val[i] = 0;
      }
    }
    __syncthreads();
  }
};

// The bulk traffic of the position-hinted pass -- every read's bases and meta record in, every record out, each touched
// once -- is marked non-temporal (the `nt` bit of the global load / store): it streams through the caches without pushing
// out what the passes behind it live on (their code, the index and graph tables), which they would otherwise find cold at
// every launch.  GTX_NO_NT at build time: plain accesses (A/B).
#ifdef GTX_NO_NT
template <class T>
static __device__ inline T stream_load(T const * p) { return *p; }
template <class T>
static __device__ inline void stream_store(T * p, T const & v) { *p = v; }
#else
static __device__ inline uint32_t stream_load(uint32_t const * p) { return __builtin_nontemporal_load(p); }
static __device__ inline uint4_t stream_load(uint4_t const * p)
{
  typedef uint32_t v4 __attribute__((ext_vector_type(4)));
  v4 const v = __builtin_nontemporal_load(reinterpret_cast<v4 const *>(p));
  return uint4_t{v.x, v.y, v.z, v.w};
}
static __device__ inline void stream_store(uint4_t * p, uint4_t const & v)
{
  typedef uint32_t v4 __attribute__((ext_vector_type(4)));
  v4 const x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, reinterpret_cast<v4 *>(p));
}
#endif

#ifndef GTX_TASK_CHUNK
#define GTX_TASK_CHUNK 64
#endif
constexpr uint32_t TASK_CHUNK = GTX_TASK_CHUNK; // reads a wave claims per visit to the task counter

// Pass 1 (express): every (read, orientation) task.  The simple reads -- one label per k-mer, abutting, at most one
// variant, tail inside a reference node -- are finished here with a workspace of 2 KB, i.e. at full occupancy (the
// kernel is bound by the latency of its dependent memory round trips, resident waves are throughput).  Every other
// task is queued for pass 2.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6))) void gtx_align_express_kernel(GraphView g, IndexView ix, uint8_t const * __restrict__ seq,
                                                               uint32_t seq_stride, gtx_read_meta const * __restrict__ meta,
                                                               uint32_t n_reads, uint32_t * __restrict__ records,
                                                               uint32_t rec_words, uint32_t force_both, uint32_t * task_counter,
                                                               uint32_t * __restrict__ queue, uint32_t * queue_count,
                                                               uint32_t queue_all)
{
  __shared__ SeedWorkspace ws;
  // The packed bases of the next read are fetched while the current one is processed (one round trip off the chain)
  // and handed over through LDS.
  __shared__ uint32_t seq_words[AlignCfg::MAX_READ / 8];
  // Tasks for pass 2 are collected per chunk and appended with one atomic: a counter takes ~90 M atomics/s, one per task
  // would set the pace of this kernel.
  __shared__ uint32_t pending[2 * TASK_CHUNK];
  bool const prefetch = ((seq_stride | static_cast<uint32_t>(reinterpret_cast<uintptr_t>(seq))) & 3u) == 0u && seq_stride <= AlignCfg::MAX_READ / 2;
  uint32_t const lane = threadIdx.x & 63u;
#ifdef GTX_PROF
  if (threadIdx.x < 16)
    ws.prof_acc[threadIdx.x] = 0;
  WaveHip::lds_sync();
#endif
  // Reads are claimed dynamically (one atomic per TASK_CHUNK reads): the grid is sized to what is resident at once and
  // reads differ in cost, a static split leaves CUs idle.
  for (;;)
  {
    uint32_t const base = wave_claim(task_counter, TASK_CHUNK);
    if (base >= n_reads)
      break;
    uint32_t const end = base + TASK_CHUNK < n_reads ? base + TASK_CHUNK : n_reads;
    uint32_t next_word = 0, n_pending = 0;
    if (prefetch && 4 * lane < seq_stride)
      next_word = reinterpret_cast<uint32_t const *>(seq + static_cast<uint64_t>(base) * seq_stride)[lane];
    gtx_read_meta next_meta = meta[base];
    for (uint32_t read = base; read < end; ++read)
    {
      gtx_read_meta const m = next_meta;
      if (read + 1 < end)
        next_meta = meta[read + 1]; // (needed one read later: off the latency chain)
      uint32_t const len = m.l_qseq;
      uint8_t const * read_seq = seq + static_cast<uint64_t>(read) * seq_stride;
      if (prefetch)
      {
        if (4 * lane < seq_stride)
          seq_words[lane] = next_word;
        WaveHip::lds_sync();
        if (read + 1 < end && 4 * lane < seq_stride)
          next_word = reinterpret_cast<uint32_t const *>(read_seq + seq_stride)[lane];
        read_seq = reinterpret_cast<uint8_t const *>(seq_words);
      }
      // align_read (alignment.cpp:331-363): reads shorter than 2K-1 stay unaligned; the reverse orientation is only
      // computed for reads that are not part of a concordant pair
      bool const too_short = len < 2 * K - 1, too_long = len > AlignCfg::MAX_READ;
      bool const rev = needs_reverse(m, force_both != 0);
      for (uint32_t orient = 0; orient < 2; ++orient)
      {
        uint32_t * rec = records + (static_cast<uint64_t>(read) * 2 + orient) * rec_words;
        if (too_short || too_long || (orient == 1 && !rev))
        {
          if (lane == 0)
          {
            rec[0] = too_long ? (static_cast<uint32_t>(GTX_ST_RECORD_OVERFLOW) << 16) : 0u;
            rec[1] = len << 16;
          }
          continue;
        }
        if (queue_all || !express_one<WaveHip>(g, ix, ws, read_seq, len, orient == 1, rec, rec_words))
        {
          if (lane == 0)
            pending[n_pending] = read * 2 + orient;
          ++n_pending;
        }
      }
    }
    if (n_pending)
    {
      WaveHip::lds_sync();
      uint32_t const at = wave_claim(queue_count, n_pending); // (the queue has room for every task)
      for (uint32_t k = lane; k < n_pending; k += 64)
        queue[at + k] = pending[k];
      WaveHip::lds_sync();
    }
  }
#ifdef GTX_PROF
  WaveHip::lds_sync();
  if (threadIdx.x < 16)
    atomicAdd(g.prof + threadIdx.x, ws.prof_acc[threadIdx.x]);
#endif
}

// Pass 1, four reads per wavefront (express4.inl): the default.  Group gi of 16 lanes takes read `first + gi` of the
// chunk; the forward task of a read is finished here or queued, a reverse-orientation task (discordant pairs,
// force_align_both_orientations) always goes to pass 2.
template <class E4>
__device__ __forceinline__ void express4_pass(GraphView const & g, IndexView const & ix, uint8_t const * __restrict__ seq, uint32_t seq_stride,
                                              gtx_read_meta const * __restrict__ meta, uint32_t n_reads, uint32_t * __restrict__ records,
                                              uint32_t rec_words, uint32_t force_both, uint32_t * task_counter,
                                              uint32_t * __restrict__ queue, uint32_t * queue_count, uint32_t queue_all)
{
  __shared__ Express4Workspace<E4> ws;
  __shared__ uint8_t pending[2 * TASK_CHUNK]; // tasks handed on, as offsets from the chunk's first task
  static_assert(2 * TASK_CHUNK <= 256, "pending[] holds task offsets as bytes");
  uint32_t const lane = threadIdx.x & 63u;
  for (;;)
  {
    uint32_t const base = wave_claim(task_counter, TASK_CHUNK);
    if (base >= n_reads)
      break;
    uint32_t const end = base + TASK_CHUNK < n_reads ? base + TASK_CHUNK : n_reads;
    uint32_t n_pending = 0;
    for (uint32_t first = base; first < end; first += 4)
    {
      uint32_t const n_valid = end - first < 4 ? end - first : 4;
      // reverse orientations: queued when align_read asks for them, else an empty record
      uint32_t const gi = lane >> 4;
      bool rev = false;
      if (gi < n_valid)
      {
        gtx_read_meta const m = meta[first + gi];
        uint32_t const len = m.l_qseq;
        rev = needs_reverse(m, force_both != 0) && len >= 2 * K - 1 && len <= AlignCfg::MAX_READ;
        if (!rev && (lane & 15u) == 0)
        {
          uint32_t * rec = records + (static_cast<uint64_t>(first + gi) * 2 + 1) * rec_words;
          rec[0] = len > AlignCfg::MAX_READ ? (static_cast<uint32_t>(GTX_ST_RECORD_OVERFLOW) << 16) : 0u;
          rec[1] = len << 16;
        }
      }
      unsigned long long const REV = __ballot(rev);
      uint32_t const fwd_mask = express4<WaveHip, E4>(g, ix, ws, seq, seq_stride, meta, first, n_valid, records, rec_words, queue_all != 0);
      for (uint32_t k = 0; k < n_valid; ++k)
      {
        if ((fwd_mask >> k) & 1u)
        {
          if (lane == 0)
            pending[n_pending] = static_cast<uint8_t>((first + k - base) * 2);
          ++n_pending;
        }
        if ((REV >> (16 * k)) & 1ull)
        {
          if (lane == 0)
            pending[n_pending] = static_cast<uint8_t>((first + k - base) * 2 + 1);
          ++n_pending;
        }
      }
    }
    if (n_pending)
    {
      WaveHip::lds_sync();
      uint32_t const at = wave_claim(queue_count, n_pending); // (the queue has room for every task)
      for (uint32_t k = lane; k < n_pending; k += 64)
        queue[at + k] = base * 2 + pending[k];
      WaveHip::lds_sync();
    }
  }
}

#define GTX_EXPRESS4_ARGS                                                                                                          \
  GraphView g, IndexView ix, uint8_t const *__restrict__ seq, uint32_t seq_stride, gtx_read_meta const *__restrict__ meta,         \
    uint32_t n_reads, uint32_t *__restrict__ records, uint32_t rec_words, uint32_t force_both, uint32_t *task_counter,             \
    uint32_t *__restrict__ queue, uint32_t *queue_count, uint32_t queue_all

// the lean build: graphs whose variant sites lie far apart (ctx_upload picks by the mean distance between sites)
#ifndef GTX_LEAN_WAVES
#define GTX_LEAN_WAVES 4
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GTX_LEAN_WAVES))) void gtx_align_express4_kernel(GTX_EXPRESS4_ARGS)
{
  express4_pass<Express4Lean>(g, ix, seq, seq_stride, meta, n_reads, records, rec_words, force_both, task_counter, queue, queue_count,
                              queue_all);
}

// the wide build: dense graphs (k-mers over several sites, tails over several SNPs)
#ifndef GTX_WIDE_WAVES
#define GTX_WIDE_WAVES 4 // resident waves per SIMD the register budget is set for
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GTX_WIDE_WAVES))) void gtx_align_express4_wide_kernel(GTX_EXPRESS4_ARGS)
{
  express4_pass<Express4Wide>(g, ix, seq, seq_stride, meta, n_reads, records, rec_words, force_both, task_counter, queue, queue_count,
                              queue_all);
}

// Pass 0 (hinted.hpp): ONE READ PER LANE.  Every read of the batch comes through here first: the reverse-orientation
// task is settled (empty record, or queued for pass 2 when align_read asks for it), reads outside the length limits get
// their empty records, and the forward task is finished from the position hint where the flags of that place prove the
// global lookups -- else the read is queued for pass 1 (express4 over the queue).  One atomic per wavefront and queue.
// (profiling build only, GTX_HINT=x: a timing experiment that lets the results land in the first 1 024 record slots, i.e. stay in
//  the L2 -- it destroys the results, so the release library has no such switch)
#ifdef GTX_PROF
#define GTX_HINT_REC_SLOT(read) ((decline_all & 2u) ? ((read) & 1023u) : (read))
#else
#define GTX_HINT_REC_SLOT(read) (read)
#endif
template <uint32_t WAVES, bool DENSE, uint32_t NK = AlignCfg::KC, bool META_LDS = true>
__device__ __forceinline__ void hinted_pass(GraphView const & g, IndexView const & ix, uint8_t const * __restrict__ seq, uint32_t seq_stride,
                                            gtx_read_meta const * __restrict__ meta, uint32_t n_reads, uint32_t * __restrict__ records,
                                            uint32_t rec_words, uint32_t force_both, uint32_t * __restrict__ queue1, uint32_t * __restrict__ queue2,
                                            unsigned long long * queue_counts /* queue 2's fill count (low word), queue 1's (high word) */,
                                            uint32_t decline_all, uint8_t * __restrict__ task_flags, uint32_t * __restrict__ compact,
                                            unsigned long long * __restrict__ var_mask)
{
  // The 64 reads of a wavefront lie side by side in memory: their bases (80 B each: five 16-byte groups of four plane words,
  // graph_dev.hpp) and their meta records (20 B each) are fetched with coalesced loads -- 1 KB and 256 B per instruction
  // instead of 64 scattered lines -- and handed to the lanes through LDS (row pitch 80 B = 20 banks: 16-byte reads of 16
  // neighbouring lanes hit all 64 banks once).
  constexpr uint32_t ROW_BYTES = HintGeom<NK>::ROW_BYTES, ROW_VEC = ROW_BYTES / 16, META_WORDS = sizeof(gtx_read_meta) / 4;
  // (rows of 128 bytes -- the eight-k-mer build -- lie one vector apart: 32-word rows would put every lane's word on two banks)
  constexpr uint32_t ROW_PITCH = NK == AlignCfg::KC ? ROW_VEC : ROW_VEC + 1;
  __shared__ uint4_t s_seq[WAVES][64 * ROW_PITCH];
  // (META_LDS = false, the lean build: every lane fetches its own meta record -- twenty bytes at a stride of twenty, the
  //  wavefront's 1 280 bytes still ten whole lines -- and the workgroup's 5 KB of LDS for them are not taken: 20.5 KB per
  //  workgroup are SEVEN workgroups per CU where 25.6 were six, and the kernel's 67 registers allow seven wavefronts per SIMD)
  __shared__ uint32_t s_meta[META_LDS ? WAVES : 1][META_LDS ? 64 * META_WORDS : 1];
  // (the queue appends' few words live in the rows the wavefronts are through with by then -- wavefront w's two counts in the
  //  first words of its own rows, the workgroup's two bases behind wavefront 0's counts: with no LDS besides the rows a workgroup
  //  of the lean build takes 20 KB exactly and eight of them fit a CU)
  auto s_count = [&](uint32_t q, uint32_t w) -> uint32_t & { return reinterpret_cast<uint32_t *>(&s_seq[w][0])[q]; };
  auto s_base = [&](uint32_t q) -> uint32_t & { return reinterpret_cast<uint32_t *>(&s_seq[0][0])[2 + q]; };
  static_assert(sizeof(gtx_read_meta) % 4 == 0, "meta records are staged word-wise");
  uint32_t const lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t const wave_first = blockIdx.x * blockDim.x + wave * 64u;
  uint32_t const read = wave_first + lane;
  bool const full = wave_first + 64u <= n_reads; // (uniform per wavefront)
  bool const staged = full && seq_stride == ROW_BYTES && (reinterpret_cast<uintptr_t>(seq) & 15u) == 0;
  uint32_t own_meta[META_WORDS] = {};
  if constexpr (META_LDS)
  {
    if (full)
    {
      uint32_t const * src = reinterpret_cast<uint32_t const *>(meta + wave_first);
#pragma unroll
      for (uint32_t it = 0; it < META_WORDS; ++it)
        s_meta[wave][it * 64 + lane] = stream_load(src + it * 64 + lane);
    }
  }
  else if (read < n_reads)
  {
    uint32_t const * src = reinterpret_cast<uint32_t const *>(meta + read);
#pragma unroll
    for (uint32_t it = 0; it < META_WORDS; ++it)
      own_meta[it] = stream_load(src + it);
  }
  if (staged)
  {
    uint4_t const * src = reinterpret_cast<uint4_t const *>(seq + static_cast<uint64_t>(wave_first) * ROW_BYTES);
#pragma unroll
    for (uint32_t it = 0; it < ROW_VEC; ++it)
    {
      uint32_t const v = it * 64 + lane; // (vector v of the wavefront's 64 rows: row v / ROW_VEC, part v % ROW_VEC)
      s_seq[wave][ROW_PITCH == ROW_VEC ? v : (v / ROW_VEC) * ROW_PITCH + v % ROW_VEC] = stream_load(src + v);
    }
  }
  else if (read < n_reads)
  {
    // any other layout (another stride, an unaligned buffer, the last wavefront of a batch): every lane brings its own
    // row, byte-wise as far as the stride goes, zeros behind it
    uint8_t const * src = seq + static_cast<uint64_t>(read) * seq_stride;
    uint8_t * dst = reinterpret_cast<uint8_t *>(&s_seq[wave][lane * ROW_PITCH]);
#pragma unroll 1
    for (uint32_t k = 0; k < ROW_BYTES; ++k)
      dst[k] = k < seq_stride ? src[k] : static_cast<uint8_t>(0);
  }
  WaveHip::lds_sync();
  bool fwd = false, fwd2 = false, rev = false, staged_rec = false; // forward task to the express pass / straight to the general pass, reverse task
  uint32_t fwd_flag = 0;
  // (dense records: whole wavefronts only -- the block of 64 compact records is written as two stores over whole lines)
  bool const compact_wave = compact != nullptr && task_flags != nullptr && full;
  static_assert(HINT_STAGE_WORDS == 16 && HINT_STAGE_WORDS * 4 <= ROW_BYTES, "a staged record is four 16-byte parts inside the read's row");
  if (read < n_reads)
  {
    gtx_read_meta m;
    if constexpr (!META_LDS)
    {
      uint32_t * mw = reinterpret_cast<uint32_t *>(&m);
#pragma unroll
      for (uint32_t k = 0; k < META_WORDS; ++k)
        mw[k] = own_meta[k];
    }
    else if (full)
    {
      uint32_t * mw = reinterpret_cast<uint32_t *>(&m);
#pragma unroll
      for (uint32_t k = 0; k < META_WORDS; ++k)
        mw[k] = s_meta[wave][lane * META_WORDS + k];
    }
    else
      m = meta[read];
    uint32_t const len = m.l_qseq;
    // (the slot's address is made where it is needed -- the rare branches below, a record that is not staged -- behind a barrier
    //  the optimiser cannot move it over: kept from here it costs two registers across the whole read, and with 80 that was six
    //  spilled ones and 2 % of the kernel)
    auto rec_at = [&]() -> uint32_t *
    {
      uint32_t r = GTX_HINT_REC_SLOT(read);
      GTX_PIN(r);
      return records + static_cast<uint64_t>(r) * 2 * rec_words;
    };
    bool const outside = len < 2 * K - 1 || len > AlignCfg::MAX_READ; // align_read (alignment.cpp:331-363): stay unaligned
    rev = !outside && needs_reverse(m, force_both != 0);
    // (GTX_FLAG_FORWARD_ONLY in the read's own flag word: the caller will never look at the reverse record of a read whose
    //  reverse orientation is not aligned -- gtx_stream_push promises that for its items -- so its empty header, a cache line
    //  visit per read for eight bytes, is not written)
    if (!rev && (m.flag & GTX_FLAG_FORWARD_ONLY) == 0)
    {
      uint32_t const h0 = len > AlignCfg::MAX_READ ? (static_cast<uint32_t>(GTX_ST_RECORD_OVERFLOW) << 16) : 0u;
      uint32_t * const rec = rec_at();
      if ((rec_words & 1u) == 0 && (reinterpret_cast<uintptr_t>(records) & 7u) == 0)
        *reinterpret_cast<uint2_t *>(rec + rec_words) = uint2_t{h0, len << 16}; // (one store instruction, not two)
      else
      {
        rec[rec_words] = h0;
        rec[rec_words + 1] = len << 16;
      }
    }
    if (outside)
    {
      uint32_t * const rec = rec_at();
      rec[0] = len > AlignCfg::MAX_READ ? (static_cast<uint32_t>(GTX_ST_RECORD_OVERFLOW) << 16) : 0u;
      rec[1] = len << 16;
    }
    else if ((decline_all & 1u) != 0)
      fwd = true;
    else
    {
      // The record goes to the lane's own row in LDS first (the bases are not needed any more once hinted_one writes) and
      // from there to memory four lanes per record: a wavefront's 64 record slots are 64 different cache lines, and what
      // the store path charges is line visits per instruction -- 16 per instruction this way instead of 64.
      uint32_t * row = reinterpret_cast<uint32_t *>(&s_seq[wave][lane * ROW_PITCH]);
      bool const can_stage = (rec_words & 3u) == 0 && (reinterpret_cast<uintptr_t>(records) & 15u) == 0; // (16-byte stores into the slots)
      uint32_t where;
      if constexpr (NK == AlignCfg::KC)
        where = hinted_one<DENSE>(g, ix, row, ROW_BYTES, m, rec_at, rec_words, can_stage ? row : nullptr);
      else
        where = hinted_long_one<NK>(g, ix, row, ROW_BYTES, m, rec_at(), rec_words, can_stage ? row : nullptr);
      fwd = where == 0;
      fwd2 = where == HINT_TO_GENERAL;
      staged_rec = where == 2;
      fwd_flag = (where == 0 || where == HINT_TO_GENERAL) ? 0u : ((where == 2 ? row[1] : rec_at()[1]) >> 31);
      // (dense records, gtx_align_batch_planes_compact: a staged record of at most GTX_COMPACT_WORDS words -- no path, or one
      //  without a variant site -- leaves through the wavefront's block of the compact array below, not through its slot)
      if (compact_wave && staged_rec && (row[0] & 0xFFFFu) <= 1u && fwd_flag == 0)
      {
        staged_rec = false;
        fwd_flag = GTX_TASK_COMPACT;
      }
    }
    // (decline_all & 4, SV graphs: what this pass declines goes straight to the general pass.  The express pass finishes next to
    //  none of the reads of an SV graph -- breakpoint alleles are not its cases: cfg5, 624 of 184 k -- but looked every one of them
    //  up first: 0.66 ms of a 2.1 ms step.)
    if ((decline_all & 4u) != 0 && fwd)
    {
      fwd = false;
      fwd2 = true;
    }
    // the dense side array (gtx_align_batch_flags): what this pass settles -- the forward task it finished, the reverse
    // task that is not aligned at all; the queued tasks get theirs from gtx_task_flags_fixup_kernel behind the last pass
#ifndef GTX_X_NO_FLAG_STORE /* (experiment build: the kernel without its side bytes) */
    if (task_flags)
    {
      // (both bytes of a read in ONE store where this pass settles both -- nearly every read: a wavefront's 64 pairs are one whole
      //  line in one instruction.  As two byte stores, each of which covers every second byte of the line, the side array cost the
      //  pass 5 % of its time for 2 of its 165 bytes per read: experiment build without them, round 6)
      bool const set_fwd = !fwd && !fwd2, set_rev = !rev;
      if (set_fwd && set_rev && (reinterpret_cast<uintptr_t>(task_flags) & 1u) == 0)
        *reinterpret_cast<uint16_t *>(task_flags + 2ull * read) = static_cast<uint16_t>(fwd_flag & 0xFFu);
      else
      {
        if (set_fwd)
          task_flags[2ull * read] = static_cast<uint8_t>(fwd_flag);
        if (set_rev)
          task_flags[2ull * read + 1] = 0;
      }
    }
#endif
  }
  if (var_mask)
  {
    // (gtx_align_batch_planes_triaged: which of the wavefront's reads left this pass with a forward record that carries a variant
    //  site, one word -- what the scorer's first stage otherwise gathers from ten million side bytes)
    unsigned long long const V = __ballot(read < n_reads && !fwd && !fwd2 && (fwd_flag & GTX_TASK_HAS_VARIANTS) != 0u);
    if (lane == 0 && wave_first < n_reads)
      var_mask[wave_first >> 6] = V;
  }
  if (compact_wave)
  {
    // The wavefront's 64 compact records are 2 KB side by side: lane l of store k carries bytes [16 (64 k + l), + 16) of the block --
    // the half (l & 1) of the record of read 32 k + l / 2, out of that read's row in LDS.  Every read's eight words leave, whatever
    // they are (a record that went to its slot, a declined read's bases): whole lines, no partial write; GTX_TASK_COMPACT says
    // which ones are records.
    WaveHip::lds_sync();
    uint4_t * dst = reinterpret_cast<uint4_t *>(compact + static_cast<uint64_t>(wave_first) * GTX_COMPACT_WORDS);
#pragma unroll
    for (uint32_t k = 0; k < 2; ++k)
    {
      uint32_t const r = 32u * k + (lane >> 1);
#ifndef GTX_X_NO_COMPACT_STORE /* (experiment build: the kernel without its dense records) */
      stream_store(dst + 64u * k + lane, s_seq[wave][r * ROW_PITCH + (lane & 1u)]);
#endif
    }
  }
  {
    unsigned long long const S = __ballot(staged_rec);
    if (S != 0)
    {
      // records of HINT_STAGE_WORDS words: lanes 4r .. 4r+3 of a round carry record 16 * round + r, 16 bytes each
      uint32_t const part = lane & 3u;
#pragma unroll
      for (uint32_t round = 0; round < 4; ++round)
      {
        uint32_t const r = round * 16u + (lane >> 2);
        // (All four 16-byte parts of the staged record leave, 64 bytes for a record of 24: storing only the parts the record
        //  reaches into -- round 4, WRITE_SIZE 66 -> 34 B per read -- made this kernel SLOWER, 0.44 -> 0.565 ms: a store that
        //  covers half of a 64-byte piece of a line presumably becomes a read-modify-write at the memory side.  Fewer bytes need a
        //  denser record layout, not narrower stores.)
        if ((S >> r) & 1ull)
        {
          uint4_t const v = s_seq[wave][r * ROW_PITCH + part];
          uint32_t const rd = wave_first + r;
          uint32_t * dst = records + static_cast<uint64_t>(GTX_HINT_REC_SLOT(rd)) * 2 * rec_words;
#ifndef GTX_X_NO_REC_STORE /* (experiment build: the kernel without its record stores) */
          stream_store(reinterpret_cast<uint4_t *>(dst + 4 * part), v);
#endif
        }
      }
    }
  }
  // Queue appends: ONE atomic per workgroup for BOTH queues (a device counter takes ~100 M returning atomics a second; one
  // per wavefront -- 156 k per 10 M reads -- set the pace of this kernel, and so did a second counter per workgroup once a
  // quarter of the workgroups had something for each queue).  The two fill counts are neighbouring words -- queue 2's low,
  // queue 1's high -- and take one 64-bit add.  Every wavefront posts its counts, the first thread claims room for the
  // workgroup, every wavefront writes at its offset.
  unsigned long long const F = __ballot(fwd), R = __ballot(rev), F2 = __ballot(fwd2);
  if (lane == 0) // (a wavefront is through with its own rows here, and wavefront 0 with its own when it writes the bases)
  {
    s_count(0, wave) = static_cast<uint32_t>(__builtin_popcountll(F));
    s_count(1, wave) = static_cast<uint32_t>(__builtin_popcountll(R) + __builtin_popcountll(F2));
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    uint32_t total1 = 0, total2 = 0;
    for (uint32_t w = 0; w < WAVES; ++w)
    {
      total1 += s_count(0, w);
      total2 += s_count(1, w);
    }
    unsigned long long base = 0;
    if (total1 | total2)
      base = atomicAdd(queue_counts, (static_cast<unsigned long long>(total1) << 32) | total2);
    s_base(0) = static_cast<uint32_t>(base >> 32);
    s_base(1) = static_cast<uint32_t>(base);
  }
  __syncthreads();
  uint32_t off1 = s_base(0), off2 = s_base(1);
  for (uint32_t w = 0; w < WAVES; ++w)
    if (w < wave)
    {
      off1 += s_count(0, w);
      off2 += s_count(1, w);
    }
  if (fwd)
    queue1[off1 + static_cast<uint32_t>(__builtin_popcountll(F & ((1ull << lane) - 1ull)))] = read;
  if (rev)
    queue2[off2 + static_cast<uint32_t>(__builtin_popcountll(R & ((1ull << lane) - 1ull)))] = read * 2 + 1;
  if (fwd2)
    queue2[off2 + static_cast<uint32_t>(__builtin_popcountll(R) + __builtin_popcountll(F2 & ((1ull << lane) - 1ull)))] = read * 2;
}

#define GTX_HINTED_ARGS                                                                                                            \
  GraphView g, IndexView ix, uint8_t const *__restrict__ seq, uint32_t seq_stride, gtx_read_meta const *__restrict__ meta,         \
    uint32_t n_reads, uint32_t *__restrict__ records, uint32_t rec_words, uint32_t force_both, uint32_t *__restrict__ queue1,      \
    uint32_t *__restrict__ queue2, unsigned long long *queue_counts, uint32_t decline_all, uint8_t *__restrict__ task_flags,       \
    uint32_t *__restrict__ compact, unsigned long long *span, unsigned long long *__restrict__ var_mask
// (span, timed calls only: [0] = the largest ~(wall clock) a workgroup saw at its start, [1] = the largest wall clock at an end -- the
//  launch's own time from its first workgroup's start to its last one's end, which is what rocprofv3 reports for it.  HIP events
//  around the launch measure that only while the launch does not wait for room: with whole steps in flight on streams of their
//  own the interval between the events was 0.98 ms for a launch that runs 0.45.
//  Workgroup 0 gives the start, one workgroup in sixteen of the last 1 024 the end -- workgroups start and finish in the order of
//  their numbers to within a few microseconds, and 78 000 atomics on one word, one per workgroup's start and end, took 0.6 ms.)
#define GTX_HINTED_PASS(W, ...)                                                                                                    \
  if (span && threadIdx.x == 0 && blockIdx.x == 0)                                                                                 \
    atomicMax(span, ~static_cast<unsigned long long>(wall_clock64()));                                                             \
  hinted_pass<W, __VA_ARGS__>(g, ix, seq, seq_stride, meta, n_reads, records, rec_words, force_both, queue1, queue2, queue_counts, decline_all, task_flags, compact, var_mask); \
  if (span && threadIdx.x == 0 && blockIdx.x + 1024u >= gridDim.x && ((blockIdx.x & 15u) == 15u || blockIdx.x + 1u == gridDim.x))   \
    atomicMax(span + 1, static_cast<unsigned long long>(wall_clock64()))

#ifndef GTX_HINT_VGPRS
#define GTX_HINT_VGPRS 72
#endif
#ifndef GTX_HINT_WAVES
// Wavefronts per workgroup of the position-hinted pass.  A workgroup's LDS (6.4 KB per wavefront) is held until its last
// wavefront is through: with 16-wave workgroups one fits a CU and every workgroup's start (nothing to overlap the first
// loads with) and end are paid with idle SIMDs; several small workgroups per CU overlap them (cfg2: 16 waves 0.845 ms,
// 8 waves 0.77-0.78, 4 waves with 80 registers -- six resident workgroups -- 0.755).
// (Round 6, third session -- the dense and the eight-k-mer builds in ONE-wave workgroups: on the graphs that get them whole steps are
//  in flight, and a workgroup of four wavefronts with 25 KB of LDS is the last to be placed among the one-wave workgroups of the other
//  steps' express, general and scoring kernels -- cfg3 1.27-1.31 -> 1.35-1.38 G reads/s, 2 x 250 reads 6.06-6.23 -> 6.33-6.47 in three /
//  two A/B rounds on one box, the cluster graph the same.  The lean build: GTX_HINT_WAVES_LEAN, below.)
#define GTX_HINT_WAVES 1
#endif
// (80 registers: six wavefronts per SIMD = three of these workgroups per CU, which their LDS also allows; the compiler
// takes 83-85 when left alone -- allocated as 88: five wavefronts, two workgroups)
#ifndef GTX_HINT_PER_EU
#define GTX_HINT_PER_EU 7
#endif
// (round 6, third session: the lean build in workgroups of TWO wavefronts -- fourteen of 10 KB per CU.  A/B on one box, the cfg2 step
//  in flight: 8 / 4 / 2 / 1 wavefronts = 0.657 / 0.590 / 0.582 / 0.605 ms, the pass itself 0.46 / 0.407 / 0.394 / 0.329: the smaller the
//  workgroup the sooner a retiring one's room is taken again -- and with one-wave workgroups it is always this pass that takes it:
//  the short queues beside it get no wavefront in (express 0.18 -> 0.55 ms, a higher priority for their stream changes nothing) and
//  the scoring waits for them.)
#ifndef GTX_HINT_WAVES_LEAN
#define GTX_HINT_WAVES_LEAN 2
#endif
__global__ __launch_bounds__(64 * GTX_HINT_WAVES_LEAN) __attribute__((amdgpu_num_vgpr(GTX_HINT_VGPRS), amdgpu_waves_per_eu(GTX_HINT_PER_EU, GTX_HINT_PER_EU))) void gtx_align_hinted_kernel(GTX_HINTED_ARGS)
{
  GTX_HINTED_PASS(GTX_HINT_WAVES_LEAN, false, AlignCfg::KC, GTX_HINT_PER_EU <= 6);
}

#ifndef GTX_HINT_DENSE_WAVES
#define GTX_HINT_DENSE_WAVES 4
#endif
// The dense build (hinted.hpp: k-mers over two sites, ...) for the graphs that get the wide express pass: what it saves is
// a trip through that pass, fifty times the cost of a read here, so its own registers and occupancy matter less.
__global__ __launch_bounds__(64 * GTX_HINT_WAVES) __attribute__((amdgpu_waves_per_eu(GTX_HINT_DENSE_WAVES, GTX_HINT_DENSE_WAVES))) void gtx_align_hinted_dense_kernel(GTX_HINTED_ARGS)
{
  GTX_HINTED_PASS(GTX_HINT_WAVES, true);
}

// The build for reads of up to 256 bases (hinted_long.hpp: eight k-mers, rows of 128 bytes, the lean build's proofs): chosen when
// the rows are longer than 80 bytes.
__global__ __launch_bounds__(64 * GTX_HINT_WAVES) __attribute__((amdgpu_waves_per_eu(3, 3))) void gtx_align_hinted_long_kernel(GTX_HINTED_ARGS)
{
  GTX_HINTED_PASS(GTX_HINT_WAVES, false, 8);
}

// Pass 1 behind pass 0: express4 over the queue of forward tasks the position-hinted pass declined (four reads per
// wavefront as above, the reads of a group come from the queue instead of lying side by side).
constexpr uint32_t EXPRESS_GROUPS_PER_WAVE = 4, GENERAL_TASKS_PER_WAVE = 3; // what a wavefront of a short queue should find to do
template <class E4>
__device__ __forceinline__ void express4_queue_pass(GraphView const & g, IndexView const & ix, uint8_t const * __restrict__ seq,
                                                    uint32_t seq_stride, gtx_read_meta const * __restrict__ meta,
                                                    uint32_t * __restrict__ records, uint32_t rec_words, uint32_t * task_counter,
                                                    uint32_t const * __restrict__ queue1, uint32_t const * queue1_count,
                                                    uint32_t * __restrict__ queue2, uint32_t * queue2_count, uint32_t * handed_on,
                                                    uint32_t queue_all)
{
  __shared__ Express4Workspace<E4> ws;
  __shared__ uint32_t pending[TASK_CHUNK];
  uint32_t const lane = threadIdx.x & 63u;
  uint32_t const n = queue1_count[0];
  // The unit of work is a group of four reads.  Three quarters of a wavefront's even share are its own without asking
  // (group w, w + G, w + 2 G, ... of a grid of G wavefronts: a single device counter takes ~100 M returning atomics a
  // second, and thousands of wavefronts asking at the kernel's start was a tenth of a millisecond); the rest is claimed one
  // group at a time, which is what evens out the end of the pass.
  // (short queues only: on a long one -- dense graphs: a hundred groups per wavefront, of very different cost -- a fixed
  //  share costs more in imbalance than the atomics do; the cfg3-like workload lost a tenth)
  // A short queue is done by a part of the grid -- four groups per wavefront -- and the other wavefronts leave at once: every
  // wavefront's first group costs five times a later one (instruction fetch, first-touch translation), and the more of
  // them start at once on a CU the more each pays (cfg2, 5.2 k groups: 16 / 8 / 4 wavefronts per CU = 0.194 / 0.129 /
  // 0.127 ms).
  uint32_t const n_groups = (n + 3u) / 4u, G_all = gridDim.x;
  uint32_t const per_wave_goal = (queue_all >> 8) ? (queue_all >> 8) : EXPRESS_GROUPS_PER_WAVE; // (bits 8..: A/B override)
  uint32_t const G_want = (n_groups + per_wave_goal - 1u) / per_wave_goal;
  uint32_t const G = G_want >= G_all ? G_all : G_want > 0u ? G_want : 1u;
  if (blockIdx.x >= G)
    return;
  uint32_t const own = n_groups / G <= 8u ? (n_groups / G) * 3u / 4u : 0u;
#ifdef GTX_PROF
  if (threadIdx.x < 16)
    ws.prof_acc[threadIdx.x] = 0;
  WaveHip::lds_sync();
#endif
  uint32_t n_pending = 0;
  auto flush = [&]()
  {
    if (n_pending)
    {
      WaveHip::lds_sync();
      uint32_t const at = wave_claim(queue2_count, n_pending); // (the queue has room for every task)
      if (lane == 0)
        atomicAdd(handed_on, n_pending);
      for (uint32_t k = lane; k < n_pending; k += 64)
        queue2[at + k] = pending[k];
      WaveHip::lds_sync();
      n_pending = 0;
    }
  };
  // (groups per visit to the counter: one on a short queue, up to sixteen -- the 64 reads of TASK_CHUNK -- on a long one,
  //  where one atomic per group is 400 k atomics on one address: four milliseconds on the cfg3-like workload)
  uint32_t const per_visit = n_groups / G >= 128u ? 16u : n_groups / G >= 32u ? 4u : 1u;
  for (uint32_t i = 0, grp = 0, grp_end = 0;; ++i, ++grp)
  {
    if (i < own)
      grp = blockIdx.x + i * G;
    else if (i == own || grp == grp_end)
    {
      grp = own * G + wave_claim(task_counter, per_visit);
      grp_end = grp + per_visit;
    }
    if (grp >= n_groups)
      break;
    uint32_t const first = 4u * grp;
    uint32_t const n_valid = n - first < 4 ? n - first : 4;
    uint32_t const fwd_mask = express4<WaveHip, E4>(g, ix, ws, seq, seq_stride, meta, 0, n_valid, records, rec_words, (queue_all & 1u) != 0, queue1 + first);
    for (uint32_t k = 0; k < n_valid; ++k)
      if ((fwd_mask >> k) & 1u)
      {
        if (lane == 0)
          pending[n_pending] = queue1[first + k] * 2;
        ++n_pending;
      }
    if (n_pending + 4u > TASK_CHUNK)
      flush();
  }
  flush();
#ifdef GTX_PROF
  WaveHip::lds_sync();
  if (threadIdx.x < 16)
    atomicAdd(g.prof + 16 + threadIdx.x, ws.prof_acc[threadIdx.x]); // (second half of the counters: the express pass, per group of four reads)
#endif
}

#define GTX_EXPRESS4Q_ARGS                                                                                                         \
  GraphView g, IndexView ix, uint8_t const *__restrict__ seq, uint32_t seq_stride, gtx_read_meta const *__restrict__ meta,         \
    uint32_t *__restrict__ records, uint32_t rec_words, uint32_t *task_counter, uint32_t const *__restrict__ queue1,               \
    uint32_t const *queue1_count, uint32_t *__restrict__ queue2, uint32_t *queue2_count, uint32_t *handed_on, uint32_t queue_all

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GTX_LEAN_WAVES))) void gtx_align_express4q_kernel(GTX_EXPRESS4Q_ARGS)
{
  express4_queue_pass<Express4Lean>(g, ix, seq, seq_stride, meta, records, rec_words, task_counter, queue1, queue1_count, queue2,
                                    queue2_count, handed_on, queue_all);
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GTX_WIDE_WAVES))) void gtx_align_express4q_wide_kernel(GTX_EXPRESS4Q_ARGS)
{
  express4_queue_pass<Express4Wide>(g, ix, seq, seq_stride, meta, records, rec_words, task_counter, queue1, queue1_count, queue2,
                                    queue2_count, handed_on, queue_all);
}

// Pass 2 (general): the queued tasks through the full algorithm over LDS tables.  A task that exceeds them goes on to
// pass 3 (gtx_align_big_kernel).
#ifndef GTX_GENERAL_WAVES
#define GTX_GENERAL_WAVES 5 // resident waves per SIMD the register budget of the general pass is set for
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GTX_GENERAL_WAVES))) void gtx_align_kernel(GraphView g, IndexView ix, uint8_t const * __restrict__ seq,
                                                       uint32_t seq_stride, gtx_read_meta const * __restrict__ meta,
                                                       uint32_t * __restrict__ records, uint32_t rec_words,
                                                       uint32_t const * __restrict__ queue, uint32_t const * queue_count,
                                                       uint32_t * task_counter, uint32_t * __restrict__ big_tasks,
                                                       uint32_t big_task_cap, uint32_t * big_state, uint32_t force_big, uint32_t task_base,
                                                       uint32_t claim, uint32_t * forward_done)
{
  __shared__ AlignWorkspace ws;
#ifdef GTX_PROF
  if (threadIdx.x < 16)
    ws.prof_acc[threadIdx.x] = 0;
  WaveHip::lds_sync();
#endif
  uint32_t const queued = queue_count[0];
  // (a short queue is done by a part of the grid, three tasks per wavefront, as in the express pass: cfg2, 7.6 k tasks:
  //  20 / 12 / 8 / 4 wavefronts per CU = 0.38 / 0.33 / 0.33 / 0.40 ms)
  uint32_t const per_wave_goal = (force_big >> 8) ? (force_big >> 8) : GENERAL_TASKS_PER_WAVE; // (bits 8..: A/B override)
  uint32_t const G_all = gridDim.x, G_want = (queued + per_wave_goal - 1u) / per_wave_goal;
  uint32_t const G = G_want >= G_all ? G_all : G_want > 0u ? G_want : 1u;
  if (blockIdx.x >= G)
    return;
  // tasks per visit to the counter (one atomic per task: the counter, not the work, sets the pace of a long queue; large
  // claims: the last claims of a short queue decide when the pass ends -- a wavefront that draws a second claim of four
  // ~70 us tasks finishes 0.3 ms after one that does not).  `claim` = 0: sized to the queue -- four tasks per visit while
  // every wavefront gets 32 and more, two from twelve on, else one ...
  uint32_t const per_wave = queued / G;
  uint32_t const CLAIM = claim ? claim : per_wave >= 32u ? 4u : per_wave >= 12u ? 2u : 1u;
  // ... and three quarters of a wavefront's even share are its own without asking (task w, w + G, w + 2 G, ... of a grid
  // of G wavefronts): thousands of wavefronts asking one counter at the kernel's start cost as much as the tasks
  // (short queues only: a long one -- dense graphs -- holds tasks of very different cost, and a fixed share costs more in
  //  imbalance than the atomics do)
  uint32_t const own = (claim || per_wave > 8u) ? 0u : per_wave * 3u / 4u;
  uint32_t n_forward = 0; // forward tasks this wavefront did (statistics: one add per wavefront at the end)
  for (uint32_t i = 0, t = 0, t_end = 0;; ++i, ++t)
  {
    if (i < own)
      t = blockIdx.x + i * G;
    else if (i == own || t == t_end)
    {
      t = own * G + wave_claim(task_counter, CLAIM);
      if (t >= queued)
        break;
      t_end = t + CLAIM < queued ? t + CLAIM : queued;
    }
    uint32_t const task = WaveHip::uni(queue[t]), read = task >> 1, orient = task & 1u;
    n_forward += orient == 0 ? 1u : 0u;
    uint32_t const len = WaveHip::uni(static_cast<uint32_t>(meta[read].l_qseq));
    uint32_t * rec = records + static_cast<uint64_t>(task) * rec_words;
#ifdef GTX_PROF
    unsigned long long const task_t0 = clock64();
    unsigned long long phase0 = 0;
    WaveHip::lds_sync();
    if (threadIdx.x < 10)
      phase0 = ws.prof_acc[threadIdx.x];
    WaveHip::lds_sync();
#endif
    uint32_t const st = align_one<WaveHip>(g, ix, ws, seq + static_cast<uint64_t>(read) * seq_stride, len, orient == 1, rec, rec_words,
                                           /*try_fast=*/false);
#ifdef GTX_PROF
    // (profiling build: how long the slowest tasks are -- a short queue ends with its longest task: [10] = the longest task's
    //  cycles << 32 | its task id, [11] / [12] / [13] = tasks of more than 100 k / 200 k / 400 k cycles, [14] = this kernel's
    //  first-to-last cycle span is left to the host's events)
    if ((threadIdx.x & 63u) == 0)
    {
      unsigned long long const dt = clock64() - task_t0;
      atomicMax(g.prof + 10, (dt << 32) | task);
      if (dt > 100000ull)
        atomicAdd(g.prof + 11, 1ull);
      if (dt > 200000ull)
        atomicAdd(g.prof + 12, 1ull);
      if (dt > 400000ull)
        atomicAdd(g.prof + 13, 1ull);
    }
    // (the task log: one entry of PROF_LOG_ENTRY words per task of this pass -- task, workgroup, start, cycles, the hardware
    //  id of the wavefront's place, the records' first words, then the cycles of phases 0..9; tools/task_log.py reads it)
    {
      unsigned long long slot = 0;
      if ((threadIdx.x & 63u) == 0)
        slot = atomicAdd(g.prof + 32, 1ull);
      slot = WaveHip::uni(static_cast<uint64_t>(slot));
      if (slot < PROF_LOG_ENTRIES)
      {
        unsigned long long * e = g.prof + 40 + slot * PROF_LOG_ENTRY;
        if ((threadIdx.x & 63u) == 0)
        {
          e[0] = task;
          e[1] = blockIdx.x;
          e[2] = task_t0;
          e[3] = clock64() - task_t0;
          e[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11))) << 32);
          e[5] = (static_cast<unsigned long long>(rec[1]) << 32) | rec[0];
        }
        if (threadIdx.x < 10)
          e[6 + threadIdx.x] = ws.prof_acc[threadIdx.x] - phase0;
      }
    }
#endif
    // a table of this pass overflowed: queue the task for the next pass (gtx_align_big_kernel)
    if (big_tasks && ((st & (GTX_ST_LABEL_OVERFLOW | GTX_ST_PATH_OVERFLOW | GTX_ST_DFS_OVERFLOW | GTX_ST_RECORD_OVERFLOW)) || (force_big & 1u)) &&
        (threadIdx.x & 63u) == 0)
    {
      uint32_t const slot = atomicAdd(big_state, 1u);
      if (slot < big_task_cap)
        big_tasks[slot] = task_base + task; // (the HBM-table pass runs once over the whole batch)
      else
        atomicAdd(big_state + 3, 1u);
#ifdef GTX_PROF // (profiling build: why the tasks leave this pass -- [29] = labels | paths << 32, [30] = walk | record << 32)
      atomicAdd(g.prof + 29, ((st & GTX_ST_LABEL_OVERFLOW) ? 1ull : 0ull) | ((st & GTX_ST_PATH_OVERFLOW) ? 1ull << 32 : 0ull));
      atomicAdd(g.prof + 30, ((st & GTX_ST_DFS_OVERFLOW) ? 1ull : 0ull) | ((st & GTX_ST_RECORD_OVERFLOW) ? 1ull << 32 : 0ull));
#endif
    }
  }
  if (n_forward && (threadIdx.x & 63u) == 0)
    atomicAdd(forward_done, n_forward);
#ifdef GTX_PROF
  WaveHip::lds_sync();
  if (threadIdx.x < 16)
    atomicAdd(g.prof + threadIdx.x, ws.prof_acc[threadIdx.x]);
#endif
}

// BAM nibble rows -> plane rows (graph_dev.hpp), one thread per (read, group of 32 bases): the one-off repack of callers that
// hold bam_get_seq bytes on the device (gtx_reads_to_planes), and what gtx_align_batch does with its nibble rows before the
// alignment kernels -- which read planes only -- run.
__global__ __launch_bounds__(256) void gtx_planes_kernel(uint8_t const * __restrict__ seq, uint32_t seq_stride, uint32_t n_reads,
                                                         uint32_t * __restrict__ planes, uint32_t groups)
{
  uint64_t const t = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= static_cast<uint64_t>(n_reads) * groups)
    return;
  uint32_t const read = static_cast<uint32_t>(t / groups), grp = static_cast<uint32_t>(t % groups);
  uint8_t const * row = seq + static_cast<uint64_t>(read) * seq_stride;
  uint32_t w[4];
  if ((seq_stride & 15u) == 0 && (reinterpret_cast<uintptr_t>(seq) & 15u) == 0 && 16u * (grp + 1) <= seq_stride)
  {
    uint4_t const v = reinterpret_cast<uint4_t const *>(row)[grp];
    w[0] = v.x;
    w[1] = v.y;
    w[2] = v.z;
    w[3] = v.w;
  }
  else
    for (uint32_t k = 0; k < 4; ++k)
    {
      uint32_t v = 0;
      for (uint32_t b = 0; b < 4; ++b)
      {
        uint32_t const at = 16u * grp + 4u * k + b;
        v |= (at < seq_stride ? static_cast<uint32_t>(row[at]) : 0u) << (8 * b);
      }
      w[k] = v;
    }
  uint32_t o[4];
  planes_from_nibble_words(w, o);
  reinterpret_cast<uint4_t *>(planes)[t] = uint4_t{o[0], o[1], o[2], o[3]};
}

// Scoring, stage 1 (triage): one thread per item reads the record header(s) and decides whether the item can add
// anything; 85 % of the cfg2 items cannot and end here.  No per-thread tables, so this kernel runs at full occupancy.
// The others are appended to a work queue, one atomic per wavefront.
// The dense side array of gtx_align_batch_flags for the tasks a queue names (queue1: reads whose forward task the
// position-hinted pass declined; queue2: the tasks of the general pass): one header read per queued task, behind the last pass.
// What the passes behind the general one were sent in this batch, written straight into the context's pinned words for the next
// batch's launch sizes ([0]: tasks that left the general pass, [2]: tasks that reached the exact pass).  A kernel of one wavefront and
// not a copy: the runtime's copy of four bytes is a kernel with workgroups of 1 024 threads, and on the stream of the short queues --
// beside the position-hinted pass of the next batch -- such a workgroup waited for that pass' end (190 us in the trace; the 512-thread
// copy this replaces took 9-14 us).
__global__ __launch_bounds__(64) void gtx_seen_kernel(uint32_t * __restrict__ seen, uint32_t const * __restrict__ big_state, uint32_t const * __restrict__ exact_state)
{
  if (threadIdx.x == 0)
  {
    __hip_atomic_store(seen + 0, big_state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (exact_state)
      __hip_atomic_store(seen + 2, exact_state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ __launch_bounds__(256) void gtx_task_flags_fixup_kernel(uint32_t const * __restrict__ records, uint32_t rec_words,
                                                                   uint8_t * __restrict__ task_flags, uint32_t const * __restrict__ queue1,
                                                                   uint32_t const * queue1_count, uint32_t const * __restrict__ queue2,
                                                                   uint32_t const * queue2_count, unsigned long long * __restrict__ var_mask)
{
  uint32_t const n1 = queue1 ? queue1_count[0] : 0u, n2 = queue2_count[0];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2; i += gridDim.x * blockDim.x)
  {
    uint32_t const task = i < n1 ? 2u * queue1[i] : queue2[i - n1];
    uint8_t const f = static_cast<uint8_t>(records[static_cast<uint64_t>(task) * rec_words + 1] >> 31);
    task_flags[task] = f;
    // (the position-hinted pass left the read's bit clear; a forward task of its queue is also a forward task of the general
    //  pass' queue when it went there: the same bit twice)
    if (var_mask && (task & 1u) == 0u && f != 0)
      atomicOr(var_mask + (task >> 7), 1ull << ((task >> 1) & 63u));
  }
}

// ... the bits of every read from the side array (batches aligned without the position-hinted pass)
__global__ __launch_bounds__(256) void gtx_var_masks_kernel(uint8_t const * __restrict__ task_flags, uint32_t n_reads, unsigned long long * __restrict__ var_mask)
{
  uint32_t const read = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long const V = __ballot(read < n_reads && (task_flags[2ull * (read < n_reads ? read : 0u)] & GTX_TASK_HAS_VARIANTS) != 0u);
  if ((threadIdx.x & 63u) == 0 && (read & ~63u) < n_reads)
    var_mask[read >> 6] = V;
}

// The scorer's first stage where item i is read i (gtx_align_batch_planes_triaged, GTX_TRIAGE_ITEMS_ARE_READS): the set bits of the
// reads' words as item numbers, one queue append per workgroup (256 words: 16 384 reads).
__global__ __launch_bounds__(256) void gtx_mask_triage_kernel(unsigned long long const * __restrict__ var_mask, uint32_t n_words,
                                                              uint32_t * __restrict__ work_queue, uint32_t * work_count)
{
  __shared__ uint32_t s_wave[4], s_base;
  uint32_t const t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  unsigned long long m = t < n_words ? var_mask[t] : 0ull;
  uint32_t const mine = static_cast<uint32_t>(__builtin_popcountll(m));
  uint32_t incl = mine; // inclusive prefix over the wavefront
  for (uint32_t d = 1; d < 64; d <<= 1)
  {
    uint32_t const up = __shfl_up(incl, d);
    incl += lane >= d ? up : 0u;
  }
  if (lane == 63)
    s_wave[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    uint32_t const total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    s_base = total ? atomicAdd(work_count, total) : 0u;
  }
  __syncthreads();
  uint32_t at = s_base + incl - mine;
  for (uint32_t w = 0; w < wave; ++w)
    at += s_wave[w];
  while (m != 0)
  {
    work_queue[at++] = 64u * t + static_cast<uint32_t>(__builtin_ctzll(m));
    m &= m - 1ull;
  }
}

// ... and for every task of a batch (batches aligned without the position-hinted pass)
__global__ __launch_bounds__(256) void gtx_task_flags_all_kernel(uint32_t const * __restrict__ records, uint32_t rec_words,
                                                                 uint8_t * __restrict__ task_flags, uint32_t n_tasks)
{
  uint32_t const task = blockIdx.x * blockDim.x + threadIdx.x;
  if (task < n_tasks)
    task_flags[task] = static_cast<uint8_t>(records[static_cast<uint64_t>(task) * rec_words + 1] >> 31);
}

// One-wave workgroups, sixteen items per thread (their loads are in flight together; one queue append per 1 024 items).
// A host that keeps several batches in flight (gtx_align_batch_planes_staged) runs this kernel beside the resident one-wave
// workgroups of another batch's express / general pass: a 1 024-thread workgroup then waits for sixteen free wave slots on
// ONE CU -- 0.38 ms instead of 0.09 in a kernel trace -- where a one-wave workgroup
// goes wherever a slot is.  Alone the small form costs 0.03 ms per 10 M items (GTX_TRIAGE_THREADS=1024 at build time: A/B).
#ifndef GTX_TRIAGE_THREADS
#define GTX_TRIAGE_THREADS 64
#endif
constexpr uint32_t TRIAGE_THREADS = GTX_TRIAGE_THREADS, TRIAGE_PER_THREAD = 1024 / GTX_TRIAGE_THREADS;
// records that are not results: a table-overflow status nobody took away (GTX_ST_ERROR_MASK) -- one count per workgroup visit
__global__ __launch_bounds__(256) void gtx_records_failed_kernel(uint32_t const * __restrict__ records, uint32_t rec_words, uint64_t n_slots,
                                                                 unsigned long long * __restrict__ count)
{
  unsigned long long mine = 0;
  for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < n_slots; t += static_cast<uint64_t>(gridDim.x) * blockDim.x)
    mine += ((records[t * rec_words] >> 16) & GTX_ST_ERROR_MASK) != 0u ? 1ull : 0ull;
  for (int d = 32; d >= 1; d >>= 1)
    mine += __shfl_xor(mine, d);
  if ((threadIdx.x & 63u) == 0 && mine)
    atomicAdd(count, mine);
}

// the header word of every slot back to zero (gtx_regions_run recycles its slots from region to region: the reverse slot of a
// GTX_FLAG_FORWARD_ONLY read is never written by a call and would keep what an earlier region's read left there)
__global__ __launch_bounds__(256) void gtx_records_clear_kernel(uint32_t * __restrict__ records, uint32_t rec_words, uint64_t n_slots)
{
  for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < n_slots; t += static_cast<uint64_t>(gridDim.x) * blockDim.x)
    records[t * rec_words] = 0u;
}

__global__ __launch_bounds__(TRIAGE_THREADS) void gtx_score_triage_kernel(gtx_score_item const * __restrict__ items, uint32_t n_items,
                                                                          uint32_t const * __restrict__ records, uint32_t rec_words,
                                                                          uint32_t * __restrict__ work_queue, uint32_t * work_count,
                                                                          uint32_t keeps_depth, uint8_t const * __restrict__ task_flags,
                                                                          uint32_t const * __restrict__ item_words, uint32_t * __restrict__ zero_next)
{
  // (the state words of the NEXT scoring call on this scratch -- CallScratch::d_score_state, two sets used in turn: no memset in
  //  front of this launch)
  if (zero_next && blockIdx.x == 0 && threadIdx.x < 4)
    zero_next[threadIdx.x] = 0u;
  // one queue append per WORKGROUP (a device counter takes a few hundred million returning atomics a second: one per
  // wavefront -- 156 k per 10 M items -- set the pace of this kernel)
  __shared__ uint32_t s_count[TRIAGE_PER_THREAD][TRIAGE_THREADS / 64], s_base;
  uint32_t const lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t const first = blockIdx.x * (TRIAGE_THREADS * TRIAGE_PER_THREAD);
  bool work[TRIAGE_PER_THREAD];
  unsigned long long mask[TRIAGE_PER_THREAD];
  // Three rounds, each round's loads independent of one another: the item words, then the side bytes they name, then -- only for
  // items that are not one forward-only read, none in most batches -- the items themselves.  (Written as one loop -- word, branch
  // on it, side byte -- the sixteen pairs of loads of a thread were sixteen pairs of round trips one after the other: 52 us for
  // 10 M items, a kernel that moves 60 MB; round 6.)
  uint32_t word[TRIAGE_PER_THREAD];
#pragma unroll
  for (uint32_t k = 0; k < TRIAGE_PER_THREAD; ++k)
  {
    uint32_t const i = first + k * TRIAGE_THREADS + threadIdx.x;
    // (gtx_score_batch_words: the read of an item of one forward-only read is in the compact array -- 4 bytes instead of 40)
    word[k] = item_words && i < n_items ? item_words[i] : GTX_ITEM_WORD_FULL;
  }
  bool any_full = false;
  if (task_flags)
  {
    uint8_t side[TRIAGE_PER_THREAD];
#pragma unroll
    for (uint32_t k = 0; k < TRIAGE_PER_THREAD; ++k)
      side[k] = task_flags[word[k] != GTX_ITEM_WORD_FULL ? 2ull * word[k] : 0ull]; // (a full item's byte is not looked at: any address that can be read)
#pragma unroll
    for (uint32_t k = 0; k < TRIAGE_PER_THREAD; ++k)
    {
      work[k] = word[k] != GTX_ITEM_WORD_FULL && (side[k] & GTX_TASK_HAS_VARIANTS) != 0;
      any_full = any_full || (word[k] == GTX_ITEM_WORD_FULL && first + k * TRIAGE_THREADS + threadIdx.x < n_items);
    }
  }
  else
  {
#pragma unroll
    for (uint32_t k = 0; k < TRIAGE_PER_THREAD; ++k)
    {
      work[k] = false;
      any_full = any_full || first + k * TRIAGE_THREADS + threadIdx.x < n_items; // (without the side array no item has a word)
    }
  }
  if (__ballot(any_full) != 0ull)
  {
    // (one copy of the item's test, the word read again: this loop is not unrolled -- sixteen copies cost 38 registers)
    uint32_t full_work = 0;
#pragma unroll 1
    for (uint32_t k = 0; k < TRIAGE_PER_THREAD; ++k)
    {
      uint32_t const i = first + k * TRIAGE_THREADS + threadIdx.x;
      if (i < n_items && (!task_flags || !item_words || item_words[i] == GTX_ITEM_WORD_FULL))
        full_work |= static_cast<uint32_t>(!item_is_trivial(items[i], records, rec_words, keeps_depth != 0, task_flags)) << k;
    }
#pragma unroll
    for (uint32_t k = 0; k < TRIAGE_PER_THREAD; ++k)
      work[k] = work[k] || ((full_work >> k) & 1u) != 0u;
  }
#pragma unroll
  for (uint32_t k = 0; k < TRIAGE_PER_THREAD; ++k)
  {
    mask[k] = __ballot(work[k]);
    if (lane == 0)
      s_count[k][wave] = static_cast<uint32_t>(__builtin_popcountll(mask[k]));
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    uint32_t total = 0;
    for (uint32_t k = 0; k < TRIAGE_PER_THREAD; ++k)
      for (uint32_t w = 0; w < TRIAGE_THREADS / 64; ++w)
        total += s_count[k][w];
    s_base = total ? atomicAdd(work_count, total) : 0u;
  }
  __syncthreads();
  uint32_t at = s_base;
#pragma unroll
  for (uint32_t k = 0; k < TRIAGE_PER_THREAD; ++k)
    for (uint32_t w = 0; w < TRIAGE_THREADS / 64; ++w)
    {
      if (w == wave && work[k])
        work_queue[at + static_cast<uint32_t>(__builtin_popcountll(mask[k] & ((1ull << lane) - 1ull)))] = first + k * TRIAGE_THREADS + threadIdx.x;
      at += s_count[k][w];
    }
}

// Scoring, stage 2: the items of the work queue, one thread each (orientation / pair selection, path checks, atomics).
#ifdef GTX_SCORE_WAVES // A/B builds: wavefronts per SIMD the compiler sizes the registers for
#define GTX_SCORE_ATTR __attribute__((amdgpu_waves_per_eu(GTX_SCORE_WAVES, GTX_SCORE_WAVES)))
#else
#define GTX_SCORE_ATTR
#endif
// one item of the scoring kernels' visits (lane = item): the record staged through LDS, score_item, the second pass' queue
constexpr uint32_t SCORE_STAGE_WORDS = 16, SCORE_STAGE_PITCH = SCORE_STAGE_WORDS + 1; // (an odd pitch: a word of every lane's row in a bank of its own)
__device__ __forceinline__ void score_visit(GraphView const & g, ScoreParams const & par, gtx_score_item const & it, uint32_t i, uint32_t const * __restrict__ records,
                                            uint32_t rec_words, ScoreAcc const & acc, uint32_t * s_stage, uint32_t * error_flag, uint32_t * __restrict__ big_queue,
                                            uint32_t big_queue_cap, uint32_t * big_state)
{
  // The record of the item's (first) read, forward orientation, fetched in ONE round trip and parsed from LDS: the parser
  // walks it word by word, every look a dependent load (PMC, cfg3: 68 vector loads per visit, each waited for -- the
  // kernel's time is their latencies in a row).  Records that are longer than the copy, in the arena or wide stay where they are.
  ScoreAcc mine = acc;
#ifndef GTX_NO_SCORE_STAGING
  if ((rec_words & 3u) == 0u)
  {
    uint32_t const ai = it.first.align_index;
    bool const dense = acc.compact && (acc.compact_flags[2ull * ai] & GTX_TASK_COMPACT);
    uint32_t const * const src = dense ? acc.compact + static_cast<uint64_t>(ai) * GTX_COMPACT_WORDS : records + static_cast<uint64_t>(ai) * 2 * rec_words;
    uint32_t const have = dense ? GTX_COMPACT_WORDS : (rec_words < SCORE_STAGE_WORDS ? rec_words : SCORE_STAGE_WORDS);
    uint4_t const * const q = reinterpret_cast<uint4_t const *>(src);
    uint4_t x[SCORE_STAGE_WORDS / 4];
#pragma unroll
    for (uint32_t k = 0; k < SCORE_STAGE_WORDS / 4; ++k)
      x[k] = 4 * k < have ? q[k] : uint4_t{0, 0, 0, 0};
    uint32_t const n_paths = x[0].x & 0xFFFFu, nvar = x[1].y >> 16;
    bool const whole = ((x[0].x >> 16) & GTX_ST_EXTERNAL) == 0u && (x[0].y & GTX_REC_WIDE) == 0u &&
                       (n_paths == 0 || (n_paths == 1 && 6u + 3u * nvar <= have));
    if (whole)
    {
      uint32_t * const row = s_stage + threadIdx.x * SCORE_STAGE_PITCH;
#pragma unroll
      for (uint32_t k = 0; k < SCORE_STAGE_WORDS / 4; ++k)
      {
        row[4 * k + 0] = x[k].x;
        row[4 * k + 1] = x[k].y;
        row[4 * k + 2] = x[k].z;
        row[4 * k + 3] = x[k].w;
      }
      mine.staged_from = src;
      mine.staged_copy = row;
    }
  }
#endif
  RecentHap r1[SCORE_MAX_HAPS], r2[SCORE_MAX_HAPS];
  if (!score_item<WaveHipCombine>(g, par, it, records, rec_words, mine, r1, r2, SCORE_MAX_HAPS))
  {
    // a read of this item touches more variant sites than the tables above hold (long results of the alignment's
    // last pass): nothing was added yet, queue the item for gtx_score_big_kernel
    uint32_t const slot = big_queue ? atomicAdd(big_state, 1u) : big_queue_cap;
    if (slot < big_queue_cap)
      big_queue[slot] = i;
    else
      atomicAdd(error_flag, 1u);
  }
}

__global__ __launch_bounds__(GTX_SCORE_THREADS) GTX_SCORE_ATTR void gtx_score_kernel(GraphView g, ScoreParams par, gtx_score_item const * __restrict__ items,
                                                        uint32_t const * __restrict__ work_queue, uint32_t const * work_count,
                                                        uint32_t const * __restrict__ records, uint32_t rec_words, ScoreAcc acc,
                                                        uint32_t * error_flag, uint32_t * __restrict__ big_queue,
                                                        uint32_t big_queue_cap, uint32_t * big_state)
{
  __shared__ uint32_t s_stage[GTX_SCORE_THREADS * SCORE_STAGE_PITCH];
  uint32_t const n_work = work_count[0];
  WaveHipCombine::clear(acc.combine_base);
  // a workgroup's share of the queue in one piece (every thread of the workgroup makes the same visits: the table is flushed
  // behind the piece; queues of more than GTX_SCORE_CHUNK_MAX items per workgroup are taken in several pieces)
  uint32_t per = (n_work + gridDim.x - 1u) / gridDim.x;
  per = (per + blockDim.x - 1u) / blockDim.x * blockDim.x;
  per = per > GTX_SCORE_CHUNK_MAX ? GTX_SCORE_CHUNK_MAX : per;
  for (uint32_t piece = blockIdx.x * per; piece < n_work; piece += gridDim.x * per)
  {
  uint32_t const piece_end = piece + per < n_work ? piece + per : n_work;
  for (uint32_t first = piece; first < piece_end; first += blockDim.x)
  {
    uint32_t const w = first + threadIdx.x;
#ifdef GTX_PROF // (profiling build: cycles of a visit's three parts, lane 0 of every workgroup: [25] item fetch, [26] scoring, [27] flush, [28] visits)
    unsigned long long const t0 = clock64();
    unsigned long long t1 = t0, t2 = t0;
#endif
    if (w < piece_end)
    {
      uint32_t const i = work_queue[w];
      gtx_score_item const it = items[i];
#ifdef GTX_PROF
      t1 = clock64() + (it.sample & 0u); // (behind the loads)
#endif
      score_visit(g, par, it, i, records, rec_words, acc, s_stage, error_flag, big_queue, big_queue_cap, big_state);
#ifdef GTX_PROF
      t2 = clock64();
#endif
    }
#ifdef GTX_PROF
    if (threadIdx.x == 0 && w < piece_end)
    {
      atomicAdd(g.prof + 25, t1 - t0);
      atomicAdd(g.prof + 26, t2 - t1);
      atomicAdd(g.prof + 28, 1ull);
    }
#endif
  }
#ifdef GTX_PROF
  unsigned long long const t3 = clock64();
#endif
  WaveHipCombine::flush();
#ifdef GTX_PROF
  if (threadIdx.x == 0)
    atomicAdd(g.prof + 27, clock64() - t3);
#endif
  }
}

// (Round 6, the same question again with the 4-byte item words: a one-wave workgroup walking groups of 64 neighbouring items -- group
//  b, b + G, ... --, four or sixteen groups' item words and side bytes in flight together, scoring the groups with work where they
//  lie: 0.744 / 0.758 ms per cfg2 step against 0.729 with the two launches.  The scoring code's registers leave five wavefronts per
//  SIMD to do the streaming the triage kernel does with eight of a fifth the size.)
// (Both stages in one launch -- a workgroup takes 1 024 items, keeps those with work in an LDS list and scores the list: no
//  work queue through memory, one launch less -- measured slower, 1.36 ms per cfg2 step against 1.27: the triage streams
//  400 MB and wants every wave slot, the scoring code's registers leave it 5 of 8.  Two kernels stay.)

template <class RH, uint32_t CAP>
GTX_DEV void score_big_pass(GraphView const & g, ScoreParams const & par, gtx_score_item const * __restrict__ items,
                            uint32_t const * __restrict__ records, uint32_t rec_words, ScoreAcc const & acc, uint32_t * error_flag,
                            uint32_t const * __restrict__ big_queue, uint32_t big_queue_cap, uint32_t const * big_state, RH * tables)
{
  uint32_t const queued = big_state[0] < big_queue_cap ? big_state[0] : big_queue_cap;
  uint32_t const tid = blockIdx.x * blockDim.x + threadIdx.x, n_threads = gridDim.x * blockDim.x;
  RH * r1 = tables + static_cast<uint64_t>(tid) * 2 * CAP;
  for (uint32_t q = tid; q < queued; q += n_threads)
    if (!score_item<WaveHip>(g, par, items[big_queue[q]], records, rec_words, acc, r1, r1 + CAP, CAP))
      atomicAdd(error_flag, 1u);
}

__global__ __launch_bounds__(64) void gtx_score_big_kernel(GraphView g, ScoreParams par, gtx_score_item const * __restrict__ items,
                                                           uint32_t const * __restrict__ records, uint32_t rec_words, ScoreAcc acc,
                                                           uint32_t * error_flag, uint32_t const * __restrict__ big_queue,
                                                           uint32_t big_queue_cap, uint32_t const * big_state, RecentHap * tables,
                                                           uint32_t * __restrict__ zero_next)
{
  if (zero_next && blockIdx.x == 0 && threadIdx.x < 4) // (gtx_score_batch_queued: no first stage in the call to do this)
    zero_next[threadIdx.x] = 0u;
  score_big_pass<RecentHap, SCORE_MAX_HAPS_BIG>(g, par, items, records, rec_words, acc, error_flag, big_queue, big_queue_cap, big_state, tables);
}

// the same for graphs with a site of more than 64 alleles: explain sets of GTX_WIDE_MASK_WORDS words (records with
// GTX_REC_WIDE are refused by the passes in front)
__global__ __launch_bounds__(64) void gtx_score_wide_kernel(GraphView g, ScoreParams par, gtx_score_item const * __restrict__ items,
                                                            uint32_t const * __restrict__ records, uint32_t rec_words, ScoreAcc acc,
                                                            uint32_t * error_flag, uint32_t const * __restrict__ big_queue,
                                                            uint32_t big_queue_cap, uint32_t const * big_state, RecentHapWide * tables,
                                                            uint32_t * __restrict__ zero_next)
{
  if (zero_next && blockIdx.x == 0 && threadIdx.x < 4)
    zero_next[threadIdx.x] = 0u;
  score_big_pass<RecentHapWide, SCORE_MAX_HAPS_WIDE>(g, par, items, records, rec_words, acc, error_flag, big_queue, big_queue_cap, big_state,
                                                     tables);
}

// gtx_scores_replay: every item again, in replay mode (score_core.hpp: ScoreAcc::replay_*) -- nothing is added, the
// explain_to_score calls on the marked cells are logged.  Rare (a cell reaches the guard at ~8 000x), so one launch with the
// HBM tables serves every item.
template <class RH, uint32_t CAP>
GTX_DEV void score_replay_pass(GraphView const & g, ScoreParams const & par, gtx_score_item const * __restrict__ items, uint32_t n_items,
                               uint32_t const * __restrict__ records, uint32_t rec_words, ScoreAcc acc, uint32_t * error_flag, RH * tables)
{
  uint32_t const tid = blockIdx.x * blockDim.x + threadIdx.x, n_threads = gridDim.x * blockDim.x;
  RH * r1 = tables + static_cast<uint64_t>(tid) * 2 * CAP;
  for (uint32_t i = tid; i < n_items; i += n_threads)
  {
    acc.replay_item = i;
    if (!score_item<WaveHip>(g, par, items[i], records, rec_words, acc, r1, r1 + CAP, CAP))
      atomicAdd(error_flag, 1u);
  }
}

__global__ __launch_bounds__(64) void gtx_score_replay_kernel(GraphView g, ScoreParams par, gtx_score_item const * __restrict__ items,
                                                              uint32_t n_items, uint32_t const * __restrict__ records, uint32_t rec_words,
                                                              ScoreAcc acc, uint32_t * error_flag, RecentHap * tables)
{
  score_replay_pass<RecentHap, SCORE_MAX_HAPS_BIG>(g, par, items, n_items, records, rec_words, acc, error_flag, tables);
}

__global__ __launch_bounds__(64) void gtx_score_replay_wide_kernel(GraphView g, ScoreParams par, gtx_score_item const * __restrict__ items,
                                                                   uint32_t n_items, uint32_t const * __restrict__ records,
                                                                   uint32_t rec_words, ScoreAcc acc, uint32_t * error_flag,
                                                                   RecentHapWide * tables)
{
  score_replay_pass<RecentHapWide, SCORE_MAX_HAPS_WIDE>(g, par, items, n_items, records, rec_words, acc, error_flag, tables);
}

// One thread per (sample, haplotype): call_cell in score_core.hpp.
__global__ __launch_bounds__(256) void gtx_calls_kernel(GraphView g, uint32_t n_samples, uint32_t const * __restrict__ log_score,
                                                        uint32_t const * __restrict__ gt_cov, uint32_t const * __restrict__ hap_u32,
                                                        uint8_t * __restrict__ phred, gtx_sample_call * __restrict__ calls)
{
  uint64_t const cell = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (cell < static_cast<uint64_t>(n_samples) * g.n_hap)
    call_cell(g, cell, log_score, gt_cov, hap_u32, phred, calls);
}

// ---------------------------------------------------------------------------------------------------------------
thread_local std::string g_last_error;

static bool hip_ok(hipError_t e, char const * what)
{
  if (e == hipSuccess)
    return true;
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  return false;
}

template <class T>
static bool upload(std::vector<void *> & owned, T const *& dst, T const * src, size_t n, char const * what)
{
  void * p = nullptr;
  size_t const bytes = (n ? n : 1) * sizeof(T);
  if (!hip_ok(gtx::dev_malloc(&p, bytes), what))
    return false;
  owned.push_back(p);
  if (n && !hip_ok(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice), what))
    return false;
  dst = static_cast<T const *>(p);
  return true;
}

template <class T>
static bool dev_alloc(T *& dst, size_t n, char const * what, bool zero = false)
{
  void * p = nullptr;
  if (!hip_ok(gtx::dev_malloc(&p, (n ? n : 1) * sizeof(T)), what))
    return false;
  if (zero && !hip_ok(gtx::dev_zero(p, (n ? n : 1) * sizeof(T)), what))
  {
    (void)gtx::dev_free(p);
    return false;
  }
  dst = static_cast<T *>(p);
  return true;
}

// ---- per-call scratch (gtx_ctx.hpp) ----------------------------------------------------------------------------
static void scratch_free(CallScratch & s)
{
  // (d_big_state lies behind d_counters in one allocation: one reset for both)
  void * ptrs[] = {s.d_counter_sets, s.d_queue1, s.d_queue, s.d_big_tasks, s.d_big_ws, s.d_score_state, s.d_score_queue,
                   s.d_score_tables, s.d_score_work, s.d_var_masks, s.d_wide_tasks, s.d_wide_ws, s.d_planes, s.d_exact_tasks};
  for (void * p : ptrs)
    if (p)
      (void)gtx::dev_free(p);
  for (auto & slot : s.time_ring)
    for (auto & row : slot)
      for (auto & e : row)
        if (e)
          (void)hipEventDestroy(static_cast<hipEvent_t>(e));
  for (auto & e : s.sync_events)
    if (e)
      (void)hipEventDestroy(static_cast<hipEvent_t>(e));
  if (s.side_stream)
    (void)hipStreamDestroy(static_cast<hipStream_t>(s.side_stream));
  if (s.done)
    (void)hipEventDestroy(static_cast<hipEvent_t>(s.done));
  if (s.h_span)
    (void)hipHostFree(s.h_span);
  s = CallScratch();
}

// d_counters and the pointers into it name set k of the scratch's two (gtx_ctx.hpp)
static void counter_set_select(CallScratch & s, uint32_t k)
{
  s.counter_set = k;
  s.d_counters = s.d_counter_sets + static_cast<size_t>(k) * CallScratch::COUNTER_PITCH;
  s.d_span = reinterpret_cast<unsigned long long *>(s.d_counters + 8 * CallScratch::MAX_PARTS + 48); // (its pinned home is made by the first timed call)
  s.d_big_state = s.has_big ? s.d_counters + 8 * CallScratch::MAX_PARTS : nullptr;
  s.d_wide_state = s.has_wide ? s.d_big_state + 8 : nullptr;
  s.d_exact_state = s.has_big ? s.d_big_state + 16 : nullptr;
}

static std::unique_ptr<CallScratch> scratch_new(gtx_ctx & c)
{
  auto s = std::make_unique<CallScratch>();
  hipEvent_t ev;
  bool ok = hip_ok(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "scratch event");
  if (ok)
    s->done = ev;
  static_assert(CallScratch::COUNTER_WORDS == 8 * CallScratch::MAX_PARTS + 48 + 4 && CallScratch::COUNTER_WORDS <= CallScratch::COUNTER_PITCH, "counter sets");
  ok = ok && dev_alloc(s->d_counter_sets, 2 * CallScratch::COUNTER_PITCH, "task counters + second-pass state", true); // (two sets: gtx_ctx.hpp)
  s->has_big = !c.params.no_second_pass;
  s->has_wide = s->has_big && c.has_wide_sites;
  if (ok)
    counter_set_select(*s, 0);
  if (ok && !c.params.no_second_pass)
  {
    void * ws = nullptr;
    // (workspaces for a small batch -- one workgroup per CU; a large batch grows them to c.big_blocks: align_planes)
    s->big_blocks = std::min<uint32_t>(c.big_blocks, static_cast<uint32_t>(c.n_cu > 0 ? c.n_cu : 256));
    ok = ok && hip_ok(gtx::dev_malloc(&ws, static_cast<size_t>(s->big_blocks) * sizeof(big::AlignWorkspace)), "second-pass workspaces");
    s->d_big_ws = ws;
    if (ok && c.has_wide_sites)
    {
      ok = ok && dev_alloc(s->d_wide_tasks, CallScratch::WIDE_TASK_CAP, "wide-site pass queue");
      void * wws = nullptr;
      ok = ok && hip_ok(gtx::dev_malloc(&wws, static_cast<size_t>(CallScratch::WIDE_BLOCKS) * sizeof(wide::AlignWorkspace)), "wide-site pass workspaces");
      s->d_wide_ws = wws;
    }
    // the exact pass: two queues (what did not fit the tables above; what did not fit a part of the slab) and the slab
    ok = ok && dev_alloc(s->d_exact_tasks, 3 * static_cast<size_t>(CallScratch::EXACT_TASK_CAP), "exact pass queues");
    ok = ok && dev_alloc(s->d_score_state, 8, "second-pass score state", true); // (two sets of four words; [2]: the work queue's count)
    ok = ok && dev_alloc(s->d_score_queue, gtx_ctx::SCORE_QUEUE_CAP, "second-pass score queue");
    if (c.has_wide_sites)
    {
      RecentHapWide * tables = nullptr;
      ok = ok && dev_alloc(tables, static_cast<size_t>(gtx_ctx::SCORE_BIG_THREADS) * 2 * SCORE_MAX_HAPS_WIDE, "wide-site score tables");
      s->d_score_tables = tables;
    }
    else
    {
      RecentHap * tables = nullptr;
      ok = ok && dev_alloc(tables, static_cast<size_t>(gtx_ctx::SCORE_BIG_THREADS) * 2 * SCORE_MAX_HAPS_BIG, "second-pass score tables");
      s->d_score_tables = tables;
    }
  }
  if (!ok)
  {
    scratch_free(*s);
    return nullptr;
  }
  return s;
}

// A scratch nobody is inside of: the one this stream used last (stream order separates the calls), else one whose last
// launch has completed, else a new one.
static CallScratch * scratch_acquire(gtx_ctx & c, hipStream_t stream)
{
  std::unique_lock<std::mutex> lock(c.pool_mutex);
  CallScratch * pick = nullptr;
  for (auto & s : c.pool)
    if (!s->busy && s->used && s->last_stream == stream)
      pick = s.get();
  if (!pick)
    for (auto & s : c.pool)
      if (!pick && !s->busy && (!s->used || hipEventQuery(static_cast<hipEvent_t>(s->done)) == hipSuccess))
        pick = s.get();
  if (!pick)
  {
    // A host that queues calls on one stream faster than the device does them would get a new scratch -- queues for a whole
    // batch, hipMalloc in the middle of its stream of work -- for every call it is ahead: beyond a handful in flight from
    // this stream the call waits for the oldest of them instead.  (Calls of other streams -- a host thread per BAM pool -- have
    // their own.)
    size_t mine = 0;
    for (auto & s : c.pool)
      if (!s->busy && s->submit_stream == stream)
      {
        ++mine;
        if (!pick || s->use_seq < pick->use_seq)
          pick = s.get();
      }
    if (mine < gtx_ctx::MAX_SCRATCH_IN_FLIGHT)
      pick = nullptr;
    if (pick)
    {
      pick->busy = true; // (ours from here on; the wait itself is outside the lock)
      lock.unlock();
      (void)hipEventSynchronize(static_cast<hipEvent_t>(pick->done));
      return pick;
    }
  }
  if (!pick)
  {
    auto s = scratch_new(c);
    if (!s)
      return nullptr;
    pick = s.get();
    c.pool.push_back(std::move(s));
  }
  pick->busy = true;
  return pick;
}

static void scratch_release(gtx_ctx & c, CallScratch * s, hipStream_t stream, bool was_align, hipStream_t submit_stream)
{
  s->submit_stream = submit_stream;
  (void)hipEventRecord(static_cast<hipEvent_t>(s->done), stream);
  std::lock_guard<std::mutex> lock(c.pool_mutex);
  s->used = true;
  s->last_stream = stream;
  s->use_seq = ++c.scratch_uses;
  s->busy = false;
  if (was_align)
    c.last_align = s;
}

template <class T>
static bool grow(T *& p, uint64_t & cap, uint64_t want, char const * what)
{
  if (want <= cap)
    return true;
  if (p && (!hip_ok(hipDeviceSynchronize(), what) || !hip_ok(gtx::dev_free(p), what))) // (earlier launches may still use the old buffer)
    return false;
  p = nullptr;
  cap = 0;
  if (!dev_alloc(p, want, what))
    return false;
  cap = want;
  return true;
}

// GraphView::pos_info / pos_back / pos_node from the node tables (gtx_host.cpp: flatten_graph has the host's form)
__global__ __launch_bounds__(256) void gtx_pos_tables_kernel(GraphView g, uint32_t n, uint32_t * __restrict__ pos_info, uint8_t * __restrict__ pos_back,
                                                             uint32_t * __restrict__ pos_node)
{
  uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  uint32_t const order = g.first_order + i;
  uint32_t lo = 0, hi = g.n_ref; // the last reference node that starts at or in front of the position
  while (hi - lo > 1)
  {
    uint32_t const mid = (lo + hi) >> 1;
    if (g.ref_order[mid] <= order)
      lo = mid;
    else
      hi = mid;
  }
  uint32_t const d = order - g.ref_order[lo], len = g.ref_len[lo];
  bool const inside = d < len;
  uint32_t const room = len - d;
  pos_info[i] = inside ? ((g.ref_dna[lo] + d) << 8) | (room < 255 ? room : 255) : INVALID;
  pos_back[i] = static_cast<uint8_t>(inside ? (d < 255 ? d : 255) : 0);
  pos_node[i] = inside ? lo : INVALID;
}

struct DeviceFacts
{
  bool have_prop = false;
  int n_cu = 0;
  uint32_t wall_clock_khz = 100000u;
  int align_blocks_per_cu = 0, express_blocks_per_cu = 0, express4_blocks_per_cu = 0, express4_wide_blocks_per_cu = 0, score_blocks_per_cu = 0;
};

// asked of the runtime the first time a context is made on the device (which is current: ctx_upload has set it)
static DeviceFacts const & device_facts(int device)
{
  static std::mutex m;
  static std::map<int, DeviceFacts> known;
  std::lock_guard<std::mutex> lock(m);
  auto it = known.find(device);
  if (it != known.end())
    return it->second;
  DeviceFacts f;
  hipDeviceProp_t prop;
  f.have_prop = hipGetDeviceProperties(&prop, device) == hipSuccess;
  if (f.have_prop)
    f.n_cu = prop.multiProcessorCount;
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0)
    f.wall_clock_khz = static_cast<uint32_t>(khz);
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gtx_align_kernel, 64, 0) == hipSuccess && per_cu > 0)
    f.align_blocks_per_cu = per_cu;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gtx_align_express_kernel, 64, 0) == hipSuccess && per_cu > 0)
    f.express_blocks_per_cu = per_cu;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gtx_align_express4_kernel, 64, 0) == hipSuccess && per_cu > 0)
    f.express4_blocks_per_cu = per_cu;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gtx_align_express4_wide_kernel, 64, 0) == hipSuccess && per_cu > 0)
    f.express4_wide_blocks_per_cu = per_cu;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gtx_score_kernel, GTX_SCORE_THREADS, 0) == hipSuccess && per_cu > 0)
    f.score_blocks_per_cu = per_cu;
  return known.emplace(device, f).first->second;
}

int ctx_upload(gtx_ctx & c, int device)
{
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
  {
    g_last_error = "no HIP device visible (libgtx has no CPU path)";
    return GTX_ERR_NO_DEVICE;
  }
  if (device >= n_dev)
  {
    g_last_error = "device index out of range";
    return GTX_ERR_NO_DEVICE;
  }
  if (!hip_ok(hipSetDevice(device), "hipSetDevice"))
    return GTX_ERR_HIP;
  c.device = device;
  bool const timing = std::getenv("GTX_TIMING") != nullptr; // stage times on stderr
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](char const * what)
  {
    auto const now = std::chrono::steady_clock::now();
    if (timing)
      std::fprintf(stderr, "[gtx]   upload: %-22s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  HostGraph const & h = c.graph;
  GraphView v = h.view();
  bool ok = true;
  // the node tables travel as ONE block: one allocation, one copy on the context's build stream (they were 19 of each, and a
  // region's context is made in well under a millisecond: the calls were a third of it)
  hipStream_t const bs = gtx::tls_build_stream;
  std::vector<uint8_t> & stage = c.upload_stage; // (lives until the stream has been waited for: gtx_ctx_create drops it)
  size_t cursor = 0;
  auto room = [&](size_t bytes) { size_t const at = cursor; cursor += (std::max<size_t>(bytes, 1) + 255) / 256 * 256; return at; };
  struct Piece { size_t at; void const * src; size_t bytes; };
  std::vector<Piece> pieces;
  auto place = [&](void const * src, size_t bytes) { pieces.push_back({room(bytes), src, bytes}); return pieces.back().at; };
  size_t const at_ref_order = place(h.ref_order.data(), h.ref_order.size() * sizeof(h.ref_order[0]));
  size_t const at_ref_len = place(h.ref_len.data(), h.ref_len.size() * sizeof(h.ref_len[0]));
  size_t const at_ref_dna = place(h.ref_dna.data(), h.ref_dna.size() * sizeof(h.ref_dna[0]));
  size_t const at_ref_nvar = place(h.ref_nvar.data(), h.ref_nvar.size() * sizeof(h.ref_nvar[0]));
  size_t const at_ref_first_var = place(h.ref_first_var.data(), h.ref_first_var.size() * sizeof(h.ref_first_var[0]));
  size_t const at_var_order = place(h.var_order.data(), h.var_order.size() * sizeof(h.var_order[0]));
  size_t const at_var_len = place(h.var_len.data(), h.var_len.size() * sizeof(h.var_len[0]));
  size_t const at_var_dna = place(h.var_dna.data(), h.var_dna.size() * sizeof(h.var_dna[0]));
  size_t const at_var_out_ref = place(h.var_out_ref.data(), h.var_out_ref.size() * sizeof(h.var_out_ref[0]));
  size_t const at_site_ref_reach = place(h.site_ref_reach.data(), h.site_ref_reach.size() * sizeof(h.site_ref_reach[0]));
  size_t const at_site_special_base = place(h.site_special_base.data(), h.site_special_base.size() * sizeof(h.site_special_base[0]));
  size_t const at_special_ref_reach = place(h.special_ref_reach.data(), h.special_ref_reach.size() * sizeof(h.special_ref_reach[0]));
  size_t const at_special_actual = place(h.special_actual.data(), h.special_actual.size() * sizeof(h.special_actual[0]));
  size_t const at_pos_bucket = place(h.pos_bucket.data(), h.pos_bucket.size() * sizeof(h.pos_bucket[0]));
  size_t const at_codes = place(h.codes.data(), h.codes.size() * sizeof(h.codes[0]));
  size_t const at_tri_off = place(h.tri_off.data(), h.tri_off.size() * sizeof(h.tri_off[0]));
  size_t const at_allele_off = place(h.allele_off.data(), h.allele_off.size() * sizeof(h.allele_off[0]));
  size_t const at_near_last = place(h.near_last.data(), h.near_last.size() * sizeof(h.near_last[0]));
  size_t const at_near_off = place(h.near_off.data(), h.near_off.size() * sizeof(h.near_off[0]));
  stage.assign(cursor, 0);
  for (Piece const & pc : pieces)
    if (pc.bytes)
      std::memcpy(stage.data() + pc.at, pc.src, pc.bytes);
  void * block = nullptr;
  ok = hip_ok(gtx::dev_malloc(&block, cursor), "graph tables");
  if (ok)
  {
    c.dev_allocs.push_back(block);
    ok = hip_ok(hipMemcpyAsync(block, stage.data(), cursor, hipMemcpyHostToDevice, bs), "graph tables");
    auto at = [&](auto const *& dst, size_t off) { dst = reinterpret_cast<std::remove_reference_t<decltype(dst)>>(static_cast<uint8_t *>(block) + off); };
    at(v.ref_order, at_ref_order);
    at(v.ref_len, at_ref_len);
    at(v.ref_dna, at_ref_dna);
    at(v.ref_nvar, at_ref_nvar);
    at(v.ref_first_var, at_ref_first_var);
    at(v.var_order, at_var_order);
    at(v.var_len, at_var_len);
    at(v.var_dna, at_var_dna);
    at(v.var_out_ref, at_var_out_ref);
    at(v.site_ref_reach, at_site_ref_reach);
    at(v.site_special_base, at_site_special_base);
    at(v.special_ref_reach, at_special_ref_reach);
    at(v.special_actual, at_special_actual);
    at(v.pos_bucket, at_pos_bucket);
    at(v.dna, at_codes);
    at(v.tri_off, at_tri_off);
    at(v.allele_off, at_allele_off);
    at(v.near_last, at_near_last);
    at(v.near_off, at_near_off);
  }
  if (ok && h.pos_table_len != 0)
  {
    // position -> where its base is, how far its reference node goes on and back, which node it is: made here from the
    // node tables (flatten_graph fills the host's copies for contexts without a device only)
    uint32_t const n = h.pos_table_len;
    void *pi = nullptr, *pb = nullptr, *pn = nullptr;
    ok = hip_ok(gtx::dev_malloc(&pi, n * sizeof(uint32_t)), "pos_info");
    if (ok)
      c.dev_allocs.push_back(pi);
    ok = ok && hip_ok(gtx::dev_malloc(&pb, n), "pos_back");
    if (ok)
      c.dev_allocs.push_back(pb);
    ok = ok && hip_ok(gtx::dev_malloc(&pn, n * sizeof(uint32_t)), "pos_node");
    if (ok)
    {
      c.dev_allocs.push_back(pn);
      v.pos_info = static_cast<uint32_t *>(pi);
      v.pos_back = static_cast<uint8_t *>(pb);
      v.pos_node = static_cast<uint32_t *>(pn);
      v.n_pos_info = n;
      hipLaunchKernelGGL(gtx_pos_tables_kernel, dim3((n + 255u) / 256u), dim3(256), 0, bs, v, n, static_cast<uint32_t *>(pi),
                         static_cast<uint8_t *>(pb), static_cast<uint32_t *>(pn));
      ok = hip_ok(hipGetLastError(), "gtx_pos_tables_kernel launch");
    }
  }
  lap("graph tables");
  void * pf = nullptr;
  ok = ok && hip_ok(gtx::dev_malloc(&pf, PROF_WORDS * sizeof(unsigned long long)), "profile counters");
  if (ok)
  {
    c.dev_allocs.push_back(pf);
    v.prof = static_cast<unsigned long long *>(pf);
    ok = hip_ok(gtx::dev_zero(pf, PROF_WORDS * sizeof(unsigned long long)), "profile counters");
  }
  void * ef = nullptr;
  ok = ok && hip_ok(gtx::dev_malloc(&ef, sizeof(uint32_t)), "error flag");
  if (ok)
  {
    c.dev_allocs.push_back(ef);
    c.d_error_flag = static_cast<uint32_t *>(ef);
    ok = hip_ok(gtx::dev_zero(ef, sizeof(uint32_t)), "error flag");
  }
  lap("counters");
  // (what the runtime says about the device and the kernels does not change between contexts: asked once per device -- these
  //  calls take the runtime's lock, which the builder threads of gtx_regions_run share with every launch)
  DeviceFacts const & facts = device_facts(device);
  struct { int multiProcessorCount; } prop{facts.n_cu};
  bool const have_prop = facts.have_prop;
  lap("device properties");
  if (have_prop)
    c.n_cu = prop.multiProcessorCount;
  c.wall_clock_khz = facts.wall_clock_khz;
  if (ok && !c.params.no_second_pass)
  {
    // HBM-table pass: one workspace per workgroup (one workgroup per CU is plenty for the few queued reads), arena
    // (GTX_BIG_BLOCKS_PER_CU: A/B switch.  One workgroup per CU was "plenty for the few queued reads" of an i.i.d. reference; on a
    //  reference with repeats tens of thousands of tasks come this far, every one a chain of dependent round trips to HBM, and the
    //  pass' throughput is the number of tasks in flight)
    {
      char const * bb = std::getenv("GTX_BIG_BLOCKS_PER_CU");
      uint32_t const per_cu = bb && std::atoi(bb) > 0 ? static_cast<uint32_t>(std::min(std::atoi(bb), 16)) : GTX_BIG_BLOCKS_PER_CU;
      c.big_blocks = (have_prop ? static_cast<uint32_t>(prop.multiProcessorCount) : 256u) * per_cu;
    }
    c.big_record_words = c.params.big_record_words ? c.params.big_record_words : (16ull << 20);
    // exact pass: the slab of a call in flight, and the walk candidates one task of this graph can have alive
    {
      uint32_t widest = 0;
      for (uint32_t n : c.graph.ref_nvar)
        widest = std::max(widest, n);
      c.exact_cand_cap = exact::exact_cand_cap(widest);
      char const * xm = std::getenv("GTX_EXACT_PASS_MB");
      uint64_t mb = c.params.exact_pass_mb ? c.params.exact_pass_mb : (xm && std::atol(xm) > 0) ? static_cast<uint64_t>(std::atol(xm)) : (c.has_wide_sites ? 4096u : 2048u);
      c.exact_slab_bytes = mb << 20;
      c.exact_mb_given = c.params.exact_pass_mb != 0 || (xm && std::atol(xm) > 0);
      // at most 1 024 parts, none smaller than 2 MB (32 MB where allele sets are wide); the kernel makes as many as there are tasks:
      // the tasks that come this far come in bulk -- every read over one long repeat -- and what one task takes is milliseconds
      // of dependent round trips
      char const * xq = std::getenv("GTX_EXACT_PARTS_MAX"); // (A/B switch: the most parts of the first launch)
      uint64_t const most_parts = xq && std::atol(xq) > 0 ? static_cast<uint64_t>(std::min<long>(std::atol(xq), 4096)) : GTX_EXACT_PARTS_MAX;
      c.exact_parts = static_cast<uint32_t>(std::min<uint64_t>(most_parts, std::max<uint64_t>(1u, mb / (c.has_wide_sites ? 32u : 2u))));
      if (char const * xp = std::getenv("GTX_EXACT_PARTS")) // (tests: smaller parts, so that tasks reach the launch with the whole slab)
        if (std::atol(xp) > 0)
        {
          c.exact_parts = static_cast<uint32_t>(std::min<long>(std::atol(xp), 1024));
          c.exact_fixed_parts = true;
        }
    }
    void * p = nullptr;
    ok = ok && hip_ok(gtx::dev_malloc(&p, c.big_record_words * sizeof(uint32_t)), "big-record arena");
    if (ok)
    {
      c.dev_allocs.push_back(p);
      c.d_big_records = static_cast<uint32_t *>(p);
    }
    if (ok && (c.h_big_seen = static_cast<uint32_t *>(gtx::pinned_slot_get())) != nullptr)
      c.h_big_seen[0] = c.h_big_seen[2] = 0xFFFFFFFFu; // ([0]: tasks that left the general pass in the last batch, [2]: tasks that reached the exact pass)
    else
      c.h_big_seen = nullptr; // (without it the pass is launched whole)
    ok = ok && hip_ok(gtx::dev_malloc(&p, sizeof(unsigned long long)), "arena cursor");
    if (ok)
    {
      c.dev_allocs.push_back(p);
      c.d_arena_cursor = static_cast<unsigned long long *>(p);
      ok = hip_ok(gtx::dev_zero(p, sizeof(unsigned long long)), "arena cursor");
    }
  }
  if (!ok)
    return GTX_ERR_HIP;
  lap("second-pass arena");
  c.dev_graph = v;
  if (facts.align_blocks_per_cu > 0)
    c.align_blocks_per_cu = facts.align_blocks_per_cu;
  if (facts.express_blocks_per_cu > 0)
    c.express_blocks_per_cu = facts.express_blocks_per_cu;
  if (facts.express4_blocks_per_cu > 0)
    c.express4_blocks_per_cu = facts.express4_blocks_per_cu;
  if (facts.express4_wide_blocks_per_cu > 0)
    c.express4_wide_blocks_per_cu = facts.express4_wide_blocks_per_cu;
  // (the scoring grid is what is resident, no more: 1.28 ms per cfg2 step against 1.32 with 8 workgroups per CU)
  if (facts.score_blocks_per_cu > 0)
    c.score_blocks_per_cu = facts.score_blocks_per_cu;
  lap("occupancy queries");
  // the first scratch now, so that the first call does not pay for it
  auto s = scratch_new(c);
  if (!s)
    return GTX_ERR_HIP;
  lap("first scratch");
  c.pool.push_back(std::move(s));
  return GTX_OK;
}

void ctx_release_device(gtx_ctx & c)
{
  if (c.device >= 0)
  {
    (void)hipSetDevice(c.device);
    if (!c.quiet) // (gtx_regions_run has waited for the one stream that used the context: no reason to wait for the other regions')
      (void)hipDeviceSynchronize();
  }
  if (c.h_big_seen)
  {
    gtx::pinned_slot_put(c.h_big_seen);
    c.h_big_seen = nullptr;
  }
  for (auto & s : c.pool)
    scratch_free(*s);
  c.pool.clear();
  c.last_align = nullptr;
  for (int k = 0; k < c.n_exact_slots; ++k)
  {
    if (c.exact_slot[k].idle)
      (void)hipEventDestroy(static_cast<hipEvent_t>(c.exact_slot[k].idle));
    c.exact_slot[k] = gtx_ctx::ExactSlot(); // (the slabs are among dev_allocs)
  }
  c.n_exact_slots = 0;
  for (void * st : c.pipeline_streams_all)
    (void)hipStreamDestroy(static_cast<hipStream_t>(st));
  c.pipeline_streams_all.clear();
  c.pipeline_streams_idle.clear();
  for (void * p : c.dev_allocs)
    (void)gtx::dev_free(p);
  c.dev_allocs.clear();
}

} // namespace gtx

using namespace gtx;

namespace
{
// RAII: the scratch goes back to the pool when the entry point returns, however it returns
struct ScratchHold
{
  gtx_ctx & c;
  CallScratch * s;
  hipStream_t stream; // the stream the call's last launch is on (the entry points that end elsewhere than they began set it)
  bool align;
  hipStream_t submit = stream; // the stream the call was made on
  ~ScratchHold()
  {
    if (s)
      scratch_release(c, s, stream, align, submit);
  }
};
} // namespace

extern "C" int gtx_align_batch(gtx_ctx * c, const uint8_t * d_seq, uint32_t seq_stride, const gtx_read_meta * d_meta,
                               uint32_t n_reads, uint32_t * d_records, uint32_t rec_words, void * stream)
{
  return gtx_align_batch_flags(c, d_seq, seq_stride, d_meta, n_reads, d_records, rec_words, nullptr, stream);
}

static int launch_planes_kernel(const uint8_t * d_seq, uint32_t seq_stride, uint32_t n_reads, uint8_t * d_planes, uint32_t plane_stride,
                                hipStream_t st)
{
  uint32_t const groups = plane_stride / PLANE_GROUP_BYTES;
  uint64_t const threads = static_cast<uint64_t>(n_reads) * groups;
  if (threads == 0)
    return GTX_OK;
  if (threads > 0xFFFFFFFFull * 256ull)
  {
    g_last_error = "gtx_reads_to_planes: batch too large";
    return GTX_ERR_ARG;
  }
  hipLaunchKernelGGL(gtx_planes_kernel, dim3(static_cast<uint32_t>((threads + 255u) / 256u)), dim3(256), 0, st, d_seq, seq_stride, n_reads,
                     reinterpret_cast<uint32_t *>(d_planes), groups);
  return hip_ok(hipGetLastError(), "gtx_planes_kernel launch") ? GTX_OK : GTX_ERR_HIP;
}

extern "C" int gtx_reads_to_planes(gtx_ctx * c, const uint8_t * d_seq, uint32_t seq_stride, uint32_t n_reads, uint8_t * d_planes,
                                   uint32_t plane_stride, void * stream)
{
  if (!c || plane_stride == 0 || (plane_stride % PLANE_GROUP_BYTES) != 0 || (reinterpret_cast<uintptr_t>(d_planes) & 15u) != 0 ||
      (n_reads != 0 && (!d_seq || !d_planes || seq_stride == 0)))
  {
    g_last_error = "gtx_reads_to_planes: bad argument (plane rows are 16-byte groups at a 16-byte aligned address)";
    return GTX_ERR_ARG;
  }
  if (c->device < 0)
  {
    g_last_error = "context was created without a device (libgtx has no CPU path)";
    return GTX_ERR_NO_DEVICE;
  }
  if (!hip_ok(hipSetDevice(c->device), "hipSetDevice"))
    return GTX_ERR_HIP;
  return launch_planes_kernel(d_seq, seq_stride, n_reads, d_planes, plane_stride, static_cast<hipStream_t>(stream));
}

// the slab a call's exact launches use (gtx_ctx::exact_slot); c.exact_mutex is held by the caller.  *wait: the call's stream has
// to wait for the slab's event first (every slab is busy)
// The slab a call of `n_reads` reads gets.  A large batch (a whole region's reads of a sample: HBM_SMALL_BATCH and more) gets the
// context's slab -- 2 GB by default, up to 1 024 tasks of a repeat side by side --, a small one (a 50 kb region, a chunk of a host
// thread's stream) a quarter of it: a context that lives for a millisecond must not pay for 2 GB, and the tasks of a small batch
// that come this far are few.  (gtx_params::exact_pass_mb given: always that.)  A slot is as large as the first call that made it
// needed; a larger call takes an idle slot that is large enough, makes one, or makes do with the one it has to wait for -- the
// launches are told the bytes they really have.
static uint64_t exact_slab_for(gtx_ctx const & c, uint32_t n_reads)
{
  if (c.exact_mb_given || n_reads >= gtx_ctx::HBM_SMALL_BATCH)
    return c.exact_slab_bytes;
  return std::max<uint64_t>(c.exact_slab_bytes / 4, 64ull << 20);
}

static gtx_ctx::ExactSlot const * exact_slot_for_call(gtx_ctx & c, uint64_t want, bool * wait)
{
  *wait = false;
  gtx_ctx::ExactSlot * idle_small = nullptr;
  for (int k = 0; k < c.n_exact_slots; ++k)
    if (hipEventQuery(static_cast<hipEvent_t>(c.exact_slot[k].idle)) == hipSuccess)
    {
      if (c.exact_slot[k].bytes >= want)
        return &c.exact_slot[k];
      idle_small = &c.exact_slot[k];
    }
  if (c.n_exact_slots < gtx_ctx::EXACT_SLOTS)
  {
    void * p = nullptr;
    hipEvent_t ev = nullptr;
    bool ok = hip_ok(gtx::dev_malloc(&p, want), "exact pass slab");
    if (ok)
      c.dev_allocs.push_back(p);
    ok = ok && hip_ok(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "exact pass event");
    if (ok)
    {
      c.exact_slot[c.n_exact_slots].slab = static_cast<uint8_t *>(p);
      c.exact_slot[c.n_exact_slots].bytes = want;
      c.exact_slot[c.n_exact_slots].idle = ev;
      return &c.exact_slot[c.n_exact_slots++];
    }
    if (c.n_exact_slots == 0)
      return nullptr;
  }
  if (idle_small)
    return idle_small; // (smaller than asked for, but free now)
  *wait = true;
  c.next_exact_slot = (c.next_exact_slot + 1) % c.n_exact_slots;
  return &c.exact_slot[c.next_exact_slot];
}

// gtx_align_batch_planes_triaged: the scorer's first stage behind the call's last alignment launch, on that launch's stream
struct TriageRequest
{
  gtx_score_item const * d_items;
  uint32_t const * d_item_words;
  uint32_t n_items;
  uint32_t * d_work; // [0] = count, [GTX_WORK_HEADER_WORDS ...] = the items with work
  bool items_are_reads; // GTX_TRIAGE_ITEMS_ARE_READS: item i is read i, alone and aligned forward only
};
static int align_planes(gtx_ctx * c, CallScratch * s, const uint8_t * d_seq, uint32_t seq_stride, const gtx_read_meta * d_meta, uint32_t n_reads,
                        uint32_t * d_records, uint32_t rec_words, uint8_t * d_task_flags, hipStream_t st, hipEvent_t front_event = nullptr,
                        hipStream_t tail_stream = nullptr, hipEvent_t done_event = nullptr, hipStream_t * last_stream = nullptr,
                        uint32_t * d_compact = nullptr, TriageRequest const * triage = nullptr);

// BAM nibble rows: repacked into plane rows in the call's scratch, then the same kernels
extern "C" int gtx_align_batch_flags(gtx_ctx * c, const uint8_t * d_seq, uint32_t seq_stride, const gtx_read_meta * d_meta,
                                     uint32_t n_reads, uint32_t * d_records, uint32_t rec_words, uint8_t * d_task_flags, void * stream)
{
  if (!c || rec_words < 8 || (n_reads != 0 && (!d_seq || !d_meta || !d_records || seq_stride == 0)))
  {
    g_last_error = "gtx_align_batch: bad argument";
    return GTX_ERR_ARG;
  }
  if (c->device < 0)
  {
    g_last_error = "context was created without a device (libgtx has no CPU path)";
    return GTX_ERR_NO_DEVICE;
  }
  if (n_reads == 0) // an empty batch is valid (and its buffers may be NULL)
    return GTX_OK;
  hipStream_t const st = static_cast<hipStream_t>(stream);
  if (!hip_ok(hipSetDevice(c->device), "hipSetDevice"))
    return GTX_ERR_HIP;
  ScratchHold hold{*c, scratch_acquire(*c, st), st, true};
  CallScratch * s = hold.s;
  if (!s)
    return GTX_ERR_HIP;
  uint32_t const plane_stride = (seq_stride + PLANE_GROUP_BYTES - 1u) / PLANE_GROUP_BYTES * PLANE_GROUP_BYTES;
  if (!grow(s->d_planes, s->planes_cap, static_cast<uint64_t>(n_reads) * plane_stride, "plane rows"))
    return GTX_ERR_HIP;
  if (int const rc = launch_planes_kernel(d_seq, seq_stride, n_reads, s->d_planes, plane_stride, st))
    return rc;
  return align_planes(c, s, s->d_planes, plane_stride, d_meta, n_reads, d_records, rec_words, d_task_flags, st);
}

extern "C" int gtx_align_batch_planes(gtx_ctx * c, const uint8_t * d_planes, uint32_t plane_stride, const gtx_read_meta * d_meta,
                                      uint32_t n_reads, uint32_t * d_records, uint32_t rec_words, uint8_t * d_task_flags, void * stream)
{
  return gtx_align_batch_planes_staged(c, d_planes, plane_stride, d_meta, n_reads, d_records, rec_words, d_task_flags, stream, nullptr, nullptr, nullptr);
}

static int align_batch_planes_staged(gtx_ctx * c, const uint8_t * d_planes, uint32_t plane_stride, const gtx_read_meta * d_meta, uint32_t n_reads,
                                     uint32_t * d_records, uint32_t rec_words, uint8_t * d_task_flags, void * stream, void * front_event,
                                     void * tail_stream, void * done_event, uint32_t * d_compact, TriageRequest const * triage = nullptr);

extern "C" int gtx_align_batch_planes_staged(gtx_ctx * c, const uint8_t * d_planes, uint32_t plane_stride, const gtx_read_meta * d_meta,
                                             uint32_t n_reads, uint32_t * d_records, uint32_t rec_words, uint8_t * d_task_flags, void * stream,
                                             void * front_event, void * tail_stream, void * done_event)
{
  return align_batch_planes_staged(c, d_planes, plane_stride, d_meta, n_reads, d_records, rec_words, d_task_flags, stream, front_event, tail_stream,
                                   done_event, nullptr);
}

extern "C" int gtx_align_batch_planes_compact(gtx_ctx * c, const uint8_t * d_planes, uint32_t plane_stride, const gtx_read_meta * d_meta,
                                              uint32_t n_reads, uint32_t * d_records, uint32_t rec_words, uint32_t * d_compact, uint8_t * d_task_flags,
                                              void * stream, void * front_event, void * tail_stream, void * done_event)
{
  if (n_reads != 0 && (!d_compact || !d_task_flags || (reinterpret_cast<uintptr_t>(d_compact) & 15u) != 0))
  {
    g_last_error = "gtx_align_batch_planes_compact: needs d_task_flags and a 16-byte aligned d_compact";
    return GTX_ERR_ARG;
  }
  return align_batch_planes_staged(c, d_planes, plane_stride, d_meta, n_reads, d_records, rec_words, d_task_flags, stream, front_event, tail_stream,
                                   done_event, d_compact);
}

extern "C" int gtx_align_batch_planes_triaged(gtx_ctx * c, const uint8_t * d_planes, uint32_t plane_stride, const gtx_read_meta * d_meta,
                                              uint32_t n_reads, uint32_t * d_records, uint32_t rec_words, uint32_t * d_compact, uint8_t * d_task_flags,
                                              const gtx_score_item * d_items, const uint32_t * d_item_words, uint32_t n_items, uint32_t triage_flags,
                                              uint32_t * d_work, void * stream, void * front_event, void * tail_stream, void * done_event)
{
  bool const are_reads = (triage_flags & GTX_TRIAGE_ITEMS_ARE_READS) != 0;
  if (!d_task_flags || (!d_items && !are_reads) || !d_work || (reinterpret_cast<uintptr_t>(d_work) & 15u) != 0 ||
      (d_compact && (reinterpret_cast<uintptr_t>(d_compact) & 15u) != 0) || (are_reads && n_items != n_reads) || (triage_flags & ~GTX_TRIAGE_ITEMS_ARE_READS) != 0)
  {
    g_last_error = "gtx_align_batch_planes_triaged: needs d_task_flags, the items (or GTX_TRIAGE_ITEMS_ARE_READS with n_items == n_reads) and a 16-byte "
                   "aligned d_work (and d_compact, when given, 16-byte aligned)";
    return GTX_ERR_ARG;
  }
  if (are_reads && c && c->params.is_sv_graph)
  {
    // (an SV graph's scorer also visits the reads without a variant site -- the reference depth -- which the reads' bits do not name)
    g_last_error = "gtx_align_batch_planes_triaged: GTX_TRIAGE_ITEMS_ARE_READS is not for SV graphs (every read adds to the reference depth)";
    return GTX_ERR_UNSUPPORTED;
  }
  TriageRequest const t{d_items, d_item_words, n_items, d_work, are_reads};
  return align_batch_planes_staged(c, d_planes, plane_stride, d_meta, n_reads, d_records, rec_words, d_task_flags, stream, front_event, tail_stream,
                                   done_event, d_compact, &t);
}

static int align_batch_planes_staged(gtx_ctx * c, const uint8_t * d_planes, uint32_t plane_stride, const gtx_read_meta * d_meta, uint32_t n_reads,
                                     uint32_t * d_records, uint32_t rec_words, uint8_t * d_task_flags, void * stream, void * front_event,
                                     void * tail_stream, void * done_event, uint32_t * d_compact, TriageRequest const * triage)
{
  if (!c || rec_words < 8 || plane_stride == 0 || (plane_stride % PLANE_GROUP_BYTES) != 0 || (reinterpret_cast<uintptr_t>(d_planes) & 15u) != 0 ||
      (n_reads != 0 && (!d_planes || !d_meta || !d_records)))
  {
    g_last_error = "gtx_align_batch_planes: bad argument (plane rows are 16-byte groups at a 16-byte aligned address)";
    return GTX_ERR_ARG;
  }
  if (c->device < 0)
  {
    g_last_error = "context was created without a device (libgtx has no CPU path)";
    return GTX_ERR_NO_DEVICE;
  }
  hipStream_t const st = static_cast<hipStream_t>(stream);
  if (tail_stream && !front_event)
  {
    g_last_error = "gtx_align_batch_planes_staged: a tail stream needs the front event (it is what the tail stream waits for)";
    return GTX_ERR_ARG;
  }
  if (n_reads == 0)
  {
    if ((front_event || done_event || triage) && !hip_ok(hipSetDevice(c->device), "hipSetDevice"))
      return GTX_ERR_HIP;
    if (front_event && !hip_ok(hipEventRecord(static_cast<hipEvent_t>(front_event), st), "front event"))
      return GTX_ERR_HIP;
    if (triage && !hip_ok(hipMemsetAsync(triage->d_work, 0, GTX_WORK_HEADER_WORDS * sizeof(uint32_t), st), "work queue reset")) // (no read: no item has work)
      return GTX_ERR_HIP;
    if (done_event && !hip_ok(hipEventRecord(static_cast<hipEvent_t>(done_event), st), "done event"))
      return GTX_ERR_HIP;
    return GTX_OK;
  }
  if (!hip_ok(hipSetDevice(c->device), "hipSetDevice"))
    return GTX_ERR_HIP;
  ScratchHold hold{*c, scratch_acquire(*c, st), st, true};
  if (!hold.s)
    return GTX_ERR_HIP;
  // (with a tail stream the call ends there: the scratch is free when THAT stream is through -- and is not handed to the
  //  next call on `stream` by stream order, which would reset queues the tail still reads)
  hipStream_t last = st;
  int const rc = align_planes(c, hold.s, d_planes, plane_stride, d_meta, n_reads, d_records, rec_words, d_task_flags, st,
                              static_cast<hipEvent_t>(front_event), std::getenv("GTX_PARTS") ? nullptr : static_cast<hipStream_t>(tail_stream),
                              static_cast<hipEvent_t>(done_event), &last, d_compact, triage);
  hold.stream = last; // (the stream the call's last launch is on)
  return rc;
}

// the passes over plane rows (d_seq / seq_stride: the plane rows and their pitch)
static int align_planes(gtx_ctx * c, CallScratch * s, const uint8_t * d_seq, uint32_t seq_stride, const gtx_read_meta * d_meta, uint32_t n_reads,
                        uint32_t * d_records, uint32_t rec_words, uint8_t * d_task_flags, hipStream_t st, hipEvent_t front_event, hipStream_t tail_stream,
                        hipEvent_t done_event, hipStream_t * last_stream, uint32_t * d_compact, TriageRequest const * triage)
{
  // (the pass counters and, behind them, the state of the HBM-table and wide-site passes: the set the last call left zeroed --
  //  CallScratch::d_counter_sets; after a call that failed on its way the set is zeroed here, as every call did before round 6)
  {
    uint32_t const use = s->counter_set ^ 1u;
    if (!s->spare_set_clean &&
        !hip_ok(hipMemsetAsync(s->d_counter_sets + static_cast<size_t>(use) * CallScratch::COUNTER_PITCH, 0, CallScratch::COUNTER_PITCH * sizeof(uint32_t), st), "task counter reset"))
      return GTX_ERR_HIP;
    counter_set_select(*s, use);
    s->spare_set_clean = false;
  }
  // queues: room for every task (a graph on which no read is simple sends them all)
  if (!grow(s->d_queue, s->queue_cap, 2ull * n_reads, "pass-2 queue") || !grow(s->d_queue1, s->queue1_cap, n_reads, "pass-1 queue"))
    return GTX_ERR_HIP;
  bool const second_pass = s->d_big_state != nullptr;
  if (second_pass)
  {
    // the queue holds every task of a small batch and 8 Mi tasks of a large one (tasks beyond it keep their status bit)
    uint64_t const want = std::min<uint64_t>(2ull * n_reads, 8ull << 20);
    uint64_t cap = s->big_task_cap;
    if (!grow(s->d_big_tasks, cap, want, "second-pass queue"))
      return GTX_ERR_HIP;
    s->big_task_cap = static_cast<uint32_t>(cap);
  }
  // tasks a wave of the general pass claims per visit to the counter
  char const * gc = std::getenv("GTX_GENERAL_CLAIM");
  uint32_t const general_claim = gc && std::atoi(gc) > 0 ? static_cast<uint32_t>(std::atoi(gc)) : 0u; // (0: the kernel sizes its claims to its queue)
  // test switch: 1 = every task goes through all passes (the last one decides), 2 = every task is done by pass 2
  char const * fb = std::getenv("GTX_FORCE_SECOND_PASS");
  uint32_t const force = fb ? static_cast<uint32_t>(std::atoi(fb)) : 0u;
  uint32_t const n_cu = static_cast<uint32_t>(c->n_cu > 0 ? c->n_cu : 256);
  bool timed = false;
  uint32_t epoch = 0;
  {
    std::lock_guard<std::mutex> lock(c->pool_mutex);
    timed = c->timing_armed;
    if (timed && c->epoch_queried) // the first timed call behind a query: a new epoch
    {
      ++c->time_epoch;
      c->epoch_queried = false;
    }
    epoch = c->time_epoch;
    // (GTX_TIME_EVERY=n: one call in n is timed -- a timed call brackets its launches with events, packets the streams carry
    //  between the kernels; the means of gtx_ctx_kernel_times are over the timed calls)
    static uint32_t const every = [] { char const * e = std::getenv("GTX_TIME_EVERY"); int const v = e ? std::atoi(e) : 1; return static_cast<uint32_t>(v > 0 ? v : 1); }();
    if (timed && every > 1 && (c->timed_seq++ % every) != 0)
      timed = false;
  }
  if (timed && s->ring_epoch != epoch)
  {
    s->ring_epoch = epoch;
    s->ring_used = 0;
  }
  uint32_t const force_both = static_cast<uint32_t>(c->params.force_align_both_orientations != 0);
  char const * e4 = std::getenv("GTX_EXPRESS4"); // A/B switch: 0 = one read per wavefront in pass 1
  char const * eh = std::getenv("GTX_HINT");     // A/B switch: 0 = no position-hinted pass
  bool const four = !(e4 && e4[0] == '0');
  bool const hinted = four && !(eh && eh[0] == '0');
  // GTX_EXPRESS4=lean / wide force a build (tests); else by the graph's density
  bool const wide = e4 && e4[0] == 'w' ? true : e4 && e4[0] == 'l' ? false : c->express4_wide;
  // A batch can be cut into parts whose general passes run on a second stream beside the front passes of the next part
  // (GTX_PARTS=n).  Measured on cfg2 it does not pay: the general pass is bound by instruction issue and wave slots, not
  // by memory latency -- ~6 000 wave instructions per task, one task per wavefront, a full grid holds all of a CU's LDS --
  // so its parts neither shrink with their share of the tasks (0.7 ms for a quarter of them against 1.0 ms for all) nor
  // leave room for the other stream: 3.1 / 3.6 / 4.7 / 6.6 ms per step with 1 / 2 / 4 / 8 parts.  Default: one part.
  // (Round 3, same finding with the short queues behind the position-hinted pass: pass 1 on a second stream beside a first
  //  launch of pass 2 over what pass 0 sent it, a second launch of pass 2 behind both -- 1.53 ms per step against 1.36 ms
  //  one after the other; side by side pass 1 took 0.41 ms instead of 0.19 ms and the two launches of pass 2 0.74 ms instead
  //  of 0.38 ms.  Reading every lookup table once in front of pass 1 (to spare it the cold misses) did not move it either:
  //  the cost of a short queue on this chip is the launch itself, per wavefront, not the tables' first touch.)
  char const * ep = std::getenv("GTX_PARTS");
  uint32_t parts = 1u;
  if (ep)
    parts = static_cast<uint32_t>(std::min<long>(std::max<long>(std::atol(ep), 1), CallScratch::MAX_PARTS));
  // (gtx_align_batch_planes_staged with a done event: the HBM-table pass and what follows it on a stream of the scratch's own)
  // (GTX_OWN_END=1: the HBM-table pass and what follows it on a stream of the scratch's own behind the general pass -- it has no
  //  task on most batches but wants 196 registers per wavefront to be placed, 0.14 ms of waiting beside a full chip that the
  //  tail stream need not share.  Measured: 1.12 ms per cfg2 step against 0.85 -- the runtime folds more streams than it has
  //  hardware queues onto the same ones, and the schedule loses its overlap.  Off; the done event is recorded on the tail
  //  stream.)
  char const * eo = std::getenv("GTX_OWN_END");
  bool const own_end = done_event != nullptr && tail_stream != nullptr && tail_stream != st && parts == 1 && eo && eo[0] == '1';
  if ((parts > 1 || own_end) && !s->side_stream)
  {
    hipStream_t side;
    if (!hip_ok(hipStreamCreateWithFlags(&side, hipStreamNonBlocking), "side stream"))
      return GTX_ERR_HIP;
    s->side_stream = side;
    for (auto & e : s->sync_events)
    {
      hipEvent_t ev;
      if (!hip_ok(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "sync events"))
        return GTX_ERR_HIP;
      e = ev;
    }
  }
  // (calls beyond the ring are not timed: a query empties it)
  if (timed && s->ring_used >= CallScratch::TIME_RING)
    timed = false;
  uint32_t const slot = timed ? s->ring_used : 0u;
  if (timed && !s->h_span) // (pinned: only a host that asks for kernel times pays for it; without it pass 0's time is the interval between its events)
  {
    if (hipHostMalloc(reinterpret_cast<void **>(&s->h_span), CallScratch::TIME_RING * 2 * sizeof(unsigned long long)) != hipSuccess)
      s->h_span = nullptr;
  }
  if (timed && s->h_span)
    s->h_span[2 * slot] = s->h_span[2 * slot + 1] = 0ull;
  if (timed && !s->time_ring[slot][0][0])
    for (uint32_t p = 0; p < (parts > 1 ? CallScratch::MAX_PARTS : 1u); ++p)
      for (auto & e : s->time_ring[slot][p])
      {
        hipEvent_t ev;
        if (!hip_ok(hipEventCreate(&ev), "pass events"))
          return GTX_ERR_HIP;
        e = ev;
      }
  if (timed && parts > 1 && !s->time_ring[slot][1][0]) // (a slot made for a one-part call)
    for (uint32_t p = 1; p < CallScratch::MAX_PARTS; ++p)
      for (auto & e : s->time_ring[slot][p])
      {
        hipEvent_t ev;
        if (!hip_ok(hipEventCreate(&ev), "pass events"))
          return GTX_ERR_HIP;
        e = ev;
      }
  s->timed = false;
  hipStream_t sg = parts > 1 ? static_cast<hipStream_t>(s->side_stream) : st; // stream of the general / HBM-table passes
  hipStream_t s1 = st;                                                         // stream of the express pass
  // gtx_align_batch_planes_staged with a tail stream: everything behind the front event goes there
  auto front_done = [&]()
  {
    if (!front_event)
      return;
    (void)hipEventRecord(front_event, st);
    if (tail_stream && tail_stream != st && parts == 1)
    {
      (void)hipStreamWaitEvent(tail_stream, front_event, 0);
      s1 = sg = tail_stream;
      // (from here on the tail stream holds launches that use the scratch: an error return below must leave the caller's
      //  ScratchHold recording `done` on THAT stream, else the next call on the front stream would reset counters and queues the
      //  tail stream may still be reading)
      if (last_stream)
        *last_stream = tail_stream;
    }
  };
  // (gtx_align_batch_planes_triaged where item i is read i: a bit per read, written by the position-hinted pass, completed behind
  //  the last pass)
  unsigned long long * var_masks = nullptr;
  if (triage && triage->items_are_reads)
  {
    if (!grow(s->d_var_masks, s->var_mask_cap, (static_cast<uint64_t>(n_reads) + 63u) / 64u, "variant-site bits of the reads"))
      return GTX_ERR_HIP;
    var_masks = s->d_var_masks;
  }
  auto mark = [&](uint32_t part, int k, hipStream_t on)
  {
    if (timed)
      (void)hipEventRecord(static_cast<hipEvent_t>(s->time_ring[slot][part][k]), on);
  };
  if (parts > 1)
  {
    // (the side stream starts behind everything the caller's stream holds so far: the resets above, the caller's uploads)
    (void)hipEventRecord(static_cast<hipEvent_t>(s->sync_events[CallScratch::MAX_PARTS]), st);
    (void)hipStreamWaitEvent(sg, static_cast<hipEvent_t>(s->sync_events[CallScratch::MAX_PARTS]), 0);
  }
  // (whole kilo-reads per part, at least one: n_reads < parts must not give a step of 0, an inexact division must not give
  //  a part more than asked for -- the counters, events and queues are sized for MAX_PARTS)
  uint32_t const step = parts == 1 ? n_reads : std::max<uint32_t>(1024u, static_cast<uint32_t>(((static_cast<uint64_t>(n_reads) + parts - 1u) / parts + 1023u) / 1024u * 1024u));
  uint32_t used_parts = 0;
  for (uint32_t first = 0; first < n_reads; first += step, ++used_parts)
  {
    uint32_t const n = std::min(step, n_reads - first), part = used_parts;
    uint32_t * counters = s->d_counters + 8 * part;
    uint8_t const * seq = d_seq + static_cast<uint64_t>(first) * seq_stride;
    gtx_read_meta const * meta = d_meta + first;
    uint32_t * records = d_records + static_cast<uint64_t>(first) * 2 * rec_words;
    uint32_t * queue1 = s->d_queue1 + first;
    uint32_t * queue2 = s->d_queue + 2ull * first;
    // grids: as many single-wave workgroups as are resident at once; they pull work from shared counters
    uint64_t const chunks = (static_cast<uint64_t>(n) + TASK_CHUNK - 1) / TASK_CHUNK;
    uint32_t const blocks1 = static_cast<uint32_t>(std::min<uint64_t>(chunks, static_cast<uint64_t>(n_cu) * c->express_blocks_per_cu));
    char const * egw = std::getenv("GTX_GENERAL_PER_WAVE"); // (A/B switch: tasks a wavefront of a short queue should find)
    uint32_t const general_goal = egw && std::atoi(egw) > 0 ? static_cast<uint32_t>(std::atoi(egw)) : 0u;
    // (GTX_GENERAL_GRID=<wavefronts per CU>: A/B switch; default: as many as are resident)
    char const * eg = std::getenv("GTX_GENERAL_GRID");
    // (a long queue -- the cfg3 graph: 48 k tasks -- is not done faster by more than 12 wavefronts per CU: 4 / 8 / 12 / 20 per CU =
    //  1.51 / 0.91 / 0.70 / 0.76 ms, the longest task 0.5 M cycles with 4 per CU and 3.4 M with 20; what they queue for is per CU)
    uint32_t const general_per_cu = eg && std::atoi(eg) > 0 ? static_cast<uint32_t>(std::atoi(eg)) : std::min<uint32_t>(static_cast<uint32_t>(c->align_blocks_per_cu), 12u);
    uint32_t const blocks2 = static_cast<uint32_t>(std::min<uint64_t>(2ull * n, static_cast<uint64_t>(n_cu) * general_per_cu));
    uint32_t const blocks4 = static_cast<uint32_t>(std::min<uint64_t>(
      chunks, static_cast<uint64_t>(n_cu) * (wide ? c->express4_wide_blocks_per_cu : c->express4_blocks_per_cu)));
    mark(part, 0, st);
    if (hinted)
    {
      // pass 0: one read per lane from the position hint; what it declines is queued for pass 1.  (GTX_HINT=decline: the
      // pass runs but declines everything -- a test of the queue plumbing)
      static bool const sv_express = std::getenv("GTX_SV_EXPRESS") && std::getenv("GTX_SV_EXPRESS")[0] == '1'; // (A/B switch: the express pass on SV graphs as well)
      bool const sv_skips_express = c->params.is_sv_graph != 0 && !sv_express && force == 0;
      char const * hb = std::getenv("GTX_HINT_BUILD"); // (test switch: lean | dense build of pass 0; default: dense beside the wide express pass)
      bool const hint_dense = hb && hb[0] == 'd' ? true : hb && hb[0] == 'l' ? false : c->express4_wide;
      bool const hint_long = seq_stride > HintGeom<AlignCfg::KC>::ROW_BYTES; // (rows for reads of more than 160 bases: the eight-k-mer build)
      uint32_t const hint_threads = 64u * ((hint_long || hint_dense) ? GTX_HINT_WAVES : GTX_HINT_WAVES_LEAN);
      hipLaunchKernelGGL(hint_long ? gtx_align_hinted_long_kernel : hint_dense ? gtx_align_hinted_dense_kernel : gtx_align_hinted_kernel,
                         dim3((n + hint_threads - 1u) / hint_threads), dim3(hint_threads), 0, st, c->dev_graph, c->dev_index, seq, seq_stride,
                         meta, n, records, rec_words, force_both, queue1, queue2, reinterpret_cast<unsigned long long *>(counters + 2),
                         static_cast<uint32_t>(force != 0 || (eh && eh[0] == 'd')) | (sv_skips_express ? 4u : 0u)
#ifdef GTX_PROF
                           | (eh && eh[0] == 'x' ? 2u : 0u)
#endif
                           ,
                         d_task_flags ? d_task_flags + 2ull * first : static_cast<uint8_t *>(nullptr),
                         d_compact ? d_compact + static_cast<uint64_t>(first) * GTX_COMPACT_WORDS : static_cast<uint32_t *>(nullptr),
                         timed && s->h_span ? s->d_span : static_cast<unsigned long long *>(nullptr),
                         var_masks ? var_masks + first / 64u : static_cast<unsigned long long *>(nullptr));
      if (!hip_ok(hipGetLastError(), "gtx_align_hinted_kernel launch"))
        return GTX_ERR_HIP;
      mark(part, 1, st);
      // (gtx_align_batch_planes_staged: from here on the call is short queues -- the caller's other streams may come in;
      //  GTX_STAGED_FRONT=express: the express pass stays on the caller's stream as well and the front event is recorded behind it)
      static bool const front_with_express = std::getenv("GTX_STAGED_FRONT") && std::getenv("GTX_STAGED_FRONT")[0] == 'e';
      if (first + step >= n_reads && !front_with_express)
        front_done();
      // (the queue's length is known on the device only: the grid is what can be resident, or one wavefront per group of four
      //  reads of a small batch; the kernel sizes its claims to the queue)
      char const * ew = std::getenv("GTX_EXPRESS_PER_WAVE"); // (A/B switch: groups a wavefront of a short queue should find)
      uint32_t const express_goal = ew && std::atoi(ew) > 0 ? static_cast<uint32_t>(std::atoi(ew)) : 0u;
      char const * eq = std::getenv("GTX_EXPRESS_GRID"); // (A/B switch: wavefronts per CU of the express pass)
      uint32_t const express_per_cu = eq && std::atoi(eq) > 0 ? static_cast<uint32_t>(std::atoi(eq))
                                                               : static_cast<uint32_t>(wide ? c->express4_wide_blocks_per_cu : c->express4_blocks_per_cu);
      uint32_t const blocks4q = static_cast<uint32_t>(std::min<uint64_t>((static_cast<uint64_t>(n) + 3u) / 4u, static_cast<uint64_t>(n_cu) * express_per_cu));
      hipLaunchKernelGGL(wide ? gtx_align_express4q_wide_kernel : gtx_align_express4q_kernel, dim3(blocks4q), dim3(64), 0, s1, c->dev_graph,
                         c->dev_index, seq, seq_stride, meta, records, rec_words, counters, queue1, counters + 3, queue2, counters + 2,
                         counters + 4, static_cast<uint32_t>(force != 0) | (express_goal << 8));
      if (first + step >= n_reads && front_with_express)
        front_done();
      // (the call's launches of the position-hinted pass have added to the span: home with it -- on the stream of the short queues,
      //  behind the express launch: on the caller's stream the copy sat between the pass and whatever the caller queues behind it)
      if (timed && s->h_span && first + step >= n_reads)
        (void)hipMemcpyAsync(s->h_span + 2 * slot, s->d_span, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s1);
    }
    else
    {
      mark(part, 1, st);
      if (first + step >= n_reads) // (no position-hinted pass: the event marks the call's start)
        front_done();
      if (four)
        hipLaunchKernelGGL(wide ? gtx_align_express4_wide_kernel : gtx_align_express4_kernel, dim3(blocks4), dim3(64), 0, s1, c->dev_graph,
                           c->dev_index, seq, seq_stride, meta, n, records, rec_words, force_both, counters, queue2, counters + 2,
                           static_cast<uint32_t>(force != 0));
      else
        hipLaunchKernelGGL(gtx_align_express_kernel, dim3(blocks1), dim3(64), 0, s1, c->dev_graph, c->dev_index, seq, seq_stride, meta, n,
                           records, rec_words, force_both, counters, queue2, counters + 2, static_cast<uint32_t>(force != 0));
    }
    if (!hip_ok(hipGetLastError(), "express kernel launch"))
      return GTX_ERR_HIP;
    mark(part, 2, s1);
    if (parts > 1)
    {
      (void)hipEventRecord(static_cast<hipEvent_t>(s->sync_events[part]), st);
      (void)hipStreamWaitEvent(sg, static_cast<hipEvent_t>(s->sync_events[part]), 0);
    }
    mark(part, 3, sg);
    hipLaunchKernelGGL(gtx_align_kernel, dim3(blocks2), dim3(64), 0, sg, c->dev_graph, c->dev_index, seq, seq_stride, meta, records, rec_words,
                       queue2, counters + 2, counters + 1, second_pass ? s->d_big_tasks : nullptr, s->big_task_cap, s->d_big_state,
                       static_cast<uint32_t>(force == 1) | (general_goal << 8), 2u * first, general_claim, counters + 5);
    if (!hip_ok(hipGetLastError(), "gtx_align_kernel launch"))
      return GTX_ERR_HIP;
    mark(part, 4, sg);
  }
  if (own_end && sg == tail_stream)
  {
    // The HBM-table pass has no task on most batches but wants 196 registers per wavefront to be placed at all: beside a
    // full chip that is 0.14 ms of waiting, which the tail stream -- the next batch's queues are behind it -- need not share.
    (void)hipEventRecord(static_cast<hipEvent_t>(s->sync_events[0]), sg);
    sg = static_cast<hipStream_t>(s->side_stream);
    (void)hipStreamWaitEvent(sg, static_cast<hipEvent_t>(s->sync_events[0]), 0);
  }
#ifdef GTX_EXPERIMENT
  // (experiment builds only -- never the product: what a step would take if the passes that find empty queues cost nothing)
  static bool const skip_hbm_passes = std::getenv("GTX_SKIP_HBM_PASSES") != nullptr;
#else
  constexpr bool skip_hbm_passes = false;
#endif
  if (second_pass && !skip_hbm_passes)
  {
    HbmPassArgs a;
    a.g = c->dev_graph;
    a.ix = c->dev_index;
    a.seq = d_seq;
    a.seq_stride = seq_stride;
    a.meta = d_meta;
    a.records = d_records;
    a.rec_words = rec_words;
    a.big_tasks = s->d_big_tasks;
    a.big_task_cap = s->big_task_cap;
    a.big_state = s->d_big_state;
    // (a large batch: the HBM-table pass with all its workgroups -- the workspaces grow once, the scratch is this call's)
    if (n_reads >= gtx_ctx::HBM_SMALL_BATCH && s->big_blocks < c->big_blocks)
    {
      void * ws = nullptr;
      if (!hip_ok(gtx::dev_malloc(&ws, static_cast<size_t>(c->big_blocks) * big_workspace_bytes()), "second-pass workspaces"))
        return GTX_ERR_HIP;
      // (the scratch came back to this stream in stream order only -- scratch_acquire does not wait for the host: an earlier,
      //  small call's HBM-table pass may still be running on the old block, and a block goes back to the cache only when no
      //  kernel can still use it, gtx_devmem.hpp: another context's thread could be handed it for another stream.  Once per
      //  scratch, at its first large batch.)
      if (s->d_big_ws && s->used && s->done)
        (void)hipEventSynchronize(static_cast<hipEvent_t>(s->done));
      (void)gtx::dev_free(s->d_big_ws);
      s->d_big_ws = ws;
      s->big_blocks = c->big_blocks;
    }
    a.big_blocks = n_reads >= gtx_ctx::HBM_SMALL_BATCH ? s->big_blocks : std::min<uint32_t>(s->big_blocks, static_cast<uint32_t>(c->n_cu > 0 ? c->n_cu : 256));
    if (c->h_big_seen)
    {
      // (tasks are claimed one by one from the queue: any number of workgroups does them all -- fewer only take longer when the
      //  guess is too low, and the next call knows better)
      uint32_t const seen = *static_cast<uint32_t volatile *>(c->h_big_seen);
      static bool const adaptive = !(std::getenv("GTX_BIG_GRID_ADAPTIVE") && std::getenv("GTX_BIG_GRID_ADAPTIVE")[0] == '0'); // (A/B switch)
      if (adaptive && seen != 0xFFFFFFFFu)
        a.big_blocks = static_cast<uint32_t>(std::min<uint64_t>(a.big_blocks, 2ull * seen + 32u));
    }
    a.big_ws = s->d_big_ws;
    a.wide_tasks = s->d_wide_tasks;
    a.wide_state = s->d_wide_state;
    a.wide_ws = s->d_wide_ws;
    a.exact_tasks = s->d_exact_tasks;
    a.exact_state = s->d_exact_state;
    a.exact_slab = nullptr;
    a.exact_slab_bytes = c->exact_slab_bytes;
    a.exact_cand_cap = c->exact_cand_cap;
    a.exact_part_cand_cap = std::min<uint32_t>(c->exact_cand_cap, CallScratch::EXACT_PART_CANDIDATES);
    a.exact_parts = c->exact_parts;
    a.exact_fixed_parts = c->exact_fixed_parts;
    a.wide_sites = c->has_wide_sites;
    a.arena = c->d_big_records;
    a.arena_words = c->big_record_words;
    a.arena_cursor = c->d_arena_cursor;
    char const * what = launch_hbm_passes(a, sg);

    if (!what)
    {
      // the exact launches, with one of the context's slabs: chosen, waited for if need be, used and marked busy again in one
      // critical section (the next call's wait has to see this call's record)
      std::lock_guard<std::mutex> lock(c->exact_mutex);
      bool wait = false;
      gtx_ctx::ExactSlot const * slot = exact_slot_for_call(*c, exact_slab_for(*c, n_reads), &wait);
      if (!slot)
        return GTX_ERR_HIP;
      if (wait)
        (void)hipStreamWaitEvent(sg, static_cast<hipEvent_t>(slot->idle), 0);
      a.exact_slab = slot->slab;
      a.exact_slab_bytes = slot->bytes;
      if (!c->exact_fixed_parts) // (as many parts as the slab has room for: none smaller than 2 MB, 32 MB where allele sets are wide)
        a.exact_parts = static_cast<uint32_t>(std::min<uint64_t>(c->exact_parts, std::max<uint64_t>(1u, (slot->bytes >> 20) / (c->has_wide_sites ? 32u : 2u))));
      // (the launches' grids by what the batch before sent this way -- tasks are claimed one by one, any number of workgroups does
      //  them all: a workgroup of the pass wants 32 KB of LDS, and beside the position-hinted pass of the next batch a thousand of them
      //  waited for that pass' end to find an empty queue -- 150-200 us on the stream of the short queues, the scoring behind them)
      {
        static bool const adaptive = !(std::getenv("GTX_BIG_GRID_ADAPTIVE") && std::getenv("GTX_BIG_GRID_ADAPTIVE")[0] == '0');
        uint32_t const seen_exact = c->h_big_seen ? static_cast<uint32_t volatile *>(c->h_big_seen)[2] : 0xFFFFFFFFu;
        a.exact_grid_limit = (!adaptive || c->exact_fixed_parts || seen_exact == 0xFFFFFFFFu) ? 0u : 2u * seen_exact + 4u;
      }
      what = launch_exact_passes(a, sg);
      if (!what && c->h_big_seen)
        hipLaunchKernelGGL(gtx_seen_kernel, dim3(1), dim3(64), 0, sg, c->h_big_seen, s->d_big_state, s->d_exact_state);
      (void)hipEventRecord(static_cast<hipEvent_t>(slot->idle), sg);
    }
    if (what)
    {
      (void)hip_ok(hipErrorLaunchFailure, what);
      return GTX_ERR_HIP;
    }
  }
  if (d_task_flags)
  {
    // the dense side array for what the position-hinted pass did not settle (its queue and the general pass' queue, per
    // part), or -- without that pass -- for every task
    uint32_t part = 0;
    for (uint32_t first = 0; first < n_reads; first += step, ++part)
    {
      uint32_t const n = std::min(step, n_reads - first);
      uint32_t const * counters = s->d_counters + 8 * part;
      uint32_t const * rec_part = d_records + static_cast<uint64_t>(first) * 2 * rec_words;
      uint8_t * flags_part = d_task_flags + 2ull * first;
      if (hinted)
        // (one-wave workgroups: behind the short queues this launch stands beside the position-hinted pass of the NEXT batch, whose
        //  workgroups of two wavefronts take every pair of slots a retiring one frees -- a workgroup of four wavefronts waited for that
        //  pass' end, 130-180 us in the trace of the staggered schedule, and the scoring of this batch with it)
        hipLaunchKernelGGL(gtx_task_flags_fixup_kernel, dim3(n_cu * 16u), dim3(64), 0, sg, rec_part, rec_words, flags_part, s->d_queue1 + first,
                           counters + 3, s->d_queue + 2ull * first, counters + 2,
                           var_masks ? var_masks + first / 64u : static_cast<unsigned long long *>(nullptr));
      else
      {
        hipLaunchKernelGGL(gtx_task_flags_all_kernel, dim3((2u * n + 255u) / 256u), dim3(256), 0, sg, rec_part, rec_words, flags_part, 2u * n);
        if (var_masks)
          hipLaunchKernelGGL(gtx_var_masks_kernel, dim3((n + 255u) / 256u), dim3(256), 0, sg, flags_part, n, var_masks + first / 64u);
      }
      if (!hip_ok(hipGetLastError(), "task flags launch"))
        return GTX_ERR_HIP;
    }
  }
  if (triage)
  {
    // The scorer's first stage (which items' reads carry a variant site: the side array, complete behind the launch above, and
    // the items' words) HERE, behind the short queues on their stream, instead of in front of the scoring on the stream that
    // carries the position-hinted passes: 37 us of a 650 us step there, nothing here -- the queues' stream is idle half of the time.
    if (!hip_ok(hipMemsetAsync(triage->d_work, 0, GTX_WORK_HEADER_WORDS * sizeof(uint32_t), sg), "work queue reset"))
      return GTX_ERR_HIP;
    if (var_masks)
    {
      uint32_t const n_words = (n_reads + 63u) / 64u;
      hipLaunchKernelGGL(gtx_mask_triage_kernel, dim3((n_words + 255u) / 256u), dim3(256), 0, sg, var_masks, n_words, triage->d_work + GTX_WORK_HEADER_WORDS,
                         triage->d_work);
      if (!hip_ok(hipGetLastError(), "gtx_mask_triage_kernel launch"))
        return GTX_ERR_HIP;
    }
    else if (triage->n_items)
    {
      hipLaunchKernelGGL(gtx_score_triage_kernel, dim3((triage->n_items + TRIAGE_THREADS * TRIAGE_PER_THREAD - 1) / (TRIAGE_THREADS * TRIAGE_PER_THREAD)),
                         dim3(TRIAGE_THREADS), 0, sg, triage->d_items, triage->n_items, d_records, rec_words, triage->d_work + GTX_WORK_HEADER_WORDS, triage->d_work,
                         static_cast<uint32_t>(c->params.is_sv_graph != 0), d_task_flags, triage->d_item_words, static_cast<uint32_t *>(nullptr));
      if (!hip_ok(hipGetLastError(), "gtx_score_triage_kernel launch (behind the alignment)"))
        return GTX_ERR_HIP;
    }
  }
  mark(0, 5, sg);
  if (done_event)
    (void)hipEventRecord(done_event, sg);
  // the other set of counters, zeroed for the next call behind this call's last launch (the scratch is handed on in the order of
  // that stream, or when the event recorded behind this is through: scratch_release)
  s->spare_set_clean = hipMemsetAsync(s->d_counter_sets + static_cast<size_t>(s->counter_set ^ 1u) * CallScratch::COUNTER_PITCH, 0,
                                      CallScratch::COUNTER_PITCH * sizeof(uint32_t), sg) == hipSuccess;
  if (last_stream)
    *last_stream = sg;
  if (parts > 1)
  {
    // the caller's stream goes on when the side stream is through
    (void)hipEventRecord(static_cast<hipEvent_t>(s->sync_events[CallScratch::MAX_PARTS]), sg);
    (void)hipStreamWaitEvent(st, static_cast<hipEvent_t>(s->sync_events[CallScratch::MAX_PARTS]), 0);
  }
  s->timed = timed;
  s->timed_reads = n_reads;
  if (timed)
  {
    s->ring_parts[slot] = used_parts;
    ++s->ring_used;
  }
  return GTX_OK;
}

// (ms[4], tasks[4]): position-hinted pass, express pass, general pass, HBM-table pass.  ms: the mean over the timed
// gtx_align_batch calls between the last two queries -- or since the last one --, of every stream's scratch (a host with
// several calls in flight asks once, behind them); tasks: of the last of them.
static int kernel_times(gtx_ctx * c, float * ms, uint32_t * tasks)
{
  for (int k = 0; k < 4; ++k)
  {
    ms[k] = 0.0f;
    tasks[k] = 0;
  }
  if (c->device < 0)
    return GTX_ERR_NO_DEVICE;
  CallScratch * s = nullptr;
  std::vector<CallScratch *> all;
  {
    std::lock_guard<std::mutex> lock(c->pool_mutex);
    if (!c->timing_armed) // first call: the calls from now on are timed
    {
      c->timing_armed = true;
      return GTX_OK;
    }
    s = c->last_align;
    for (auto & u : c->pool)
      if (u->ring_epoch == c->time_epoch)
        all.push_back(u.get());
    c->epoch_queried = true;
  }
  // ("nothing was timed" is decided by the ring's slots below, not by the last call: beyond TIME_RING calls of an epoch the last
  //  one is untimed while up to TIME_RING earlier ones sit in the ring.  The query is meant to come behind the calls in flight:
  //  ring_used / ring_parts of a scratch another host thread is inside of are read here without its lock.)
  if (!s)
    return GTX_OK;
  if (!hip_ok(hipSetDevice(c->device), "hipSetDevice"))
    return GTX_ERR_HIP;
  double sum[4] = {0, 0, 0, 0};
  uint32_t calls = 0;
  for (CallScratch * u : all)
  {
    uint32_t const used = std::min(u->ring_used, CallScratch::TIME_RING);
    for (uint32_t slot = 0; slot < used; ++slot)
    {
      uint32_t const n_parts = u->ring_parts[slot];
      if (n_parts == 0 || !hip_ok(hipEventSynchronize(static_cast<hipEvent_t>(u->time_ring[slot][0][5])), "pass events"))
        continue;
      for (uint32_t p = 0; p < n_parts; ++p)
      {
        float d = 0.0f;
        auto ev = [&](int k) { return static_cast<hipEvent_t>(u->time_ring[slot][p][k]); };
        if (hipEventElapsedTime(&d, ev(0), ev(1)) == hipSuccess)
        {
          // (the position-hinted pass by its own clock where the call brought it home: gtx_align_hinted_kernel's span)
          unsigned long long const t0 = u->h_span ? ~u->h_span[2 * slot] : 0ull, t1 = u->h_span ? u->h_span[2 * slot + 1] : 0ull;
          if (p == 0 && u->h_span && u->h_span[2 * slot + 1] != 0ull && t1 > t0 && c->wall_clock_khz > 0)
            d = static_cast<float>(static_cast<double>(t1 - t0) / static_cast<double>(c->wall_clock_khz));
          else if (p != 0 && u->h_span && u->h_span[2 * slot + 1] != 0ull)
            d = 0.0f; // (the span covers all parts of the call)
          sum[0] += d;
        }
        if (hipEventElapsedTime(&d, ev(1), ev(2)) == hipSuccess)
          sum[1] += d;
        if (hipEventElapsedTime(&d, ev(3), ev(4)) == hipSuccess)
          sum[2] += d;
      }
      float d = 0.0f;
      if (hipEventElapsedTime(&d, static_cast<hipEvent_t>(u->time_ring[slot][n_parts - 1][4]), static_cast<hipEvent_t>(u->time_ring[slot][0][5])) == hipSuccess)
        sum[3] += d;
      ++calls;
    }
  }
  if (calls == 0)
    return GTX_OK; // nothing was timed yet
  for (int k = 0; k < 4; ++k)
    ms[k] = static_cast<float>(sum[k] / calls);
  // (the last call's counters: when that call was beyond the ring its stream may still be busy -- wait for the scratch's own event)
  if (s->done && s->used)
    (void)hipEventSynchronize(static_cast<hipEvent_t>(s->done));
  uint32_t cnt[8 * CallScratch::MAX_PARTS] = {}, big[4] = {0, 0, 0, 0};
  (void)hipMemcpy(cnt, s->d_counters, sizeof(cnt), hipMemcpyDeviceToHost);
  if (s->d_big_state)
    (void)hipMemcpy(big, s->d_big_state, sizeof(big), hipMemcpyDeviceToHost);
  uint32_t queued2 = 0, queued1 = 0, handed = 0, direct = 0;
  for (uint32_t p = 0; p < CallScratch::MAX_PARTS; ++p)
  {
    queued2 += cnt[8 * p + 2];
    queued1 += cnt[8 * p + 3];
    handed += cnt[8 * p + 4];
    direct += cnt[8 * p + 5] - std::min(cnt[8 * p + 5], cnt[8 * p + 4]); // forward tasks of the general pass that did not come through pass 1
  }
  // forward tasks only: reverse-orientation tasks all go to the general pass
  uint32_t const hbm = std::min<uint32_t>(big[0], s->big_task_cap);
  bool const hinted = ms[0] > 0.0f;
  tasks[0] = hinted ? s->timed_reads - queued1 - direct : 0; // (direct: forward tasks the position-hinted pass sent straight to the general pass)
  tasks[1] = hinted ? queued1 - handed : s->timed_reads - std::min(queued2, s->timed_reads);
  tasks[2] = queued2 - std::min(hbm, queued2);
  tasks[3] = hbm;
  return GTX_OK;
}

extern "C" int gtx_ctx_kernel_times(gtx_ctx * c, float * ms, uint32_t * tasks)
{
  if (!c || !ms || !tasks)
    return GTX_ERR_ARG;
  return kernel_times(c, ms, tasks);
}

extern "C" int gtx_ctx_pass_times(gtx_ctx * c, float * ms, uint32_t * queued)
{
  if (!c || !ms)
    return GTX_ERR_ARG;
  float m4[4];
  uint32_t t4[4];
  int const rc = kernel_times(c, m4, t4);
  ms[0] = m4[0] + m4[1]; // everything in front of the general pass
  ms[1] = m4[2];
  ms[2] = m4[3];
  if (queued)
    *queued = t4[2] + t4[3];
  return rc;
}

extern "C" int gtx_score_batch(gtx_ctx * c, const gtx_score_item * d_items, uint32_t n_items, const uint32_t * d_records,
                               uint32_t rec_words, const gtx_score_buffers * acc, void * stream)
{
  return gtx_score_batch_flags(c, d_items, n_items, d_records, rec_words, nullptr, acc, stream);
}

static int score_batch(gtx_ctx * c, const gtx_score_item * d_items, const uint32_t * d_item_words, uint32_t n_items, const uint32_t * d_records,
                       uint32_t rec_words, const uint8_t * d_task_flags, const gtx_score_buffers * acc, void * stream, const uint32_t * d_compact = nullptr,
                       const uint32_t * d_work = nullptr);

extern "C" int gtx_score_batch_flags(gtx_ctx * c, const gtx_score_item * d_items, uint32_t n_items, const uint32_t * d_records,
                                     uint32_t rec_words, const uint8_t * d_task_flags, const gtx_score_buffers * acc, void * stream)
{
  return score_batch(c, d_items, nullptr, n_items, d_records, rec_words, d_task_flags, acc, stream);
}

extern "C" int gtx_score_batch_words(gtx_ctx * c, const gtx_score_item * d_items, const uint32_t * d_item_words, uint32_t n_items,
                                     const uint32_t * d_records, uint32_t rec_words, const uint8_t * d_task_flags, const gtx_score_buffers * acc,
                                     void * stream)
{
  if (d_item_words && !d_task_flags)
  {
    g_last_error = "gtx_score_batch_words: the compact item array needs the task-flag array (gtx_align_batch_flags / _planes)";
    return GTX_ERR_ARG;
  }
  return score_batch(c, d_items, d_item_words, n_items, d_records, rec_words, d_task_flags, acc, stream);
}

extern "C" int gtx_score_batch_compact(gtx_ctx * c, const gtx_score_item * d_items, const uint32_t * d_item_words, uint32_t n_items,
                                       const uint32_t * d_records, uint32_t rec_words, const uint32_t * d_compact, const uint8_t * d_task_flags,
                                       const gtx_score_buffers * acc, void * stream)
{
  if (!d_compact || !d_task_flags)
  {
    g_last_error = "gtx_score_batch_compact: needs the compact records and the side array gtx_align_batch_planes_compact filled";
    return GTX_ERR_ARG;
  }
  return score_batch(c, d_items, d_item_words, n_items, d_records, rec_words, d_task_flags, acc, stream, d_compact);
}

// stage 2 alone, over the work queue gtx_align_batch_planes_triaged left
extern "C" int gtx_score_batch_queued(gtx_ctx * c, const gtx_score_item * d_items, uint32_t n_items, const uint32_t * d_records, uint32_t rec_words,
                                      const uint32_t * d_compact, const uint8_t * d_task_flags, const uint32_t * d_work,
                                      const gtx_score_buffers * acc, void * stream)
{
  if (!d_work || !d_task_flags || (reinterpret_cast<uintptr_t>(d_work) & 15u) != 0)
  {
    g_last_error = "gtx_score_batch_queued: needs the work queue and the side array of gtx_align_batch_planes_triaged";
    return GTX_ERR_ARG;
  }
  return score_batch(c, d_items, nullptr, n_items, d_records, rec_words, d_task_flags, acc, stream, d_compact, d_work);
}

// gtx_score_batch_words' compact form of the items (host)
extern "C" int gtx_item_words(const gtx_score_item * items, uint32_t n_items, uint32_t * words)
{
  if (n_items && (!items || !words))
    return GTX_ERR_ARG;
  for (uint32_t i = 0; i < n_items; ++i)
  {
    gtx_score_item const & it = items[i];
    bool const one = it.second.align_index == GTX_INVALID_ID && it.kind == 0 && (it.first.flag & GTX_FLAG_FORWARD_ONLY) != 0 &&
                     it.first.align_index != GTX_ITEM_WORD_FULL;
    words[i] = one ? it.first.align_index : GTX_ITEM_WORD_FULL;
  }
  return GTX_OK;
}

static int score_batch(gtx_ctx * c, const gtx_score_item * d_items, const uint32_t * d_item_words, uint32_t n_items, const uint32_t * d_records,
                       uint32_t rec_words, const uint8_t * d_task_flags, const gtx_score_buffers * acc, void * stream, const uint32_t * d_compact,
                       const uint32_t * d_work)
{
  if (!c || !d_items || !d_records || !acc || !acc->d_log_score || !acc->d_gt_cov || !acc->d_hap_u32 || !acc->d_stat_u64 ||
      !acc->d_stat_u32 || !acc->d_conn_log || !acc->d_conn_count)
  {
    g_last_error = "gtx_score_batch: bad argument";
    return GTX_ERR_ARG;
  }
  if (c->device < 0)
  {
    g_last_error = "context was created without a device (libgtx has no CPU path)";
    return GTX_ERR_NO_DEVICE;
  }
  if (n_items == 0)
    return GTX_OK;
  hipStream_t const st = static_cast<hipStream_t>(stream);
  if (!hip_ok(hipSetDevice(c->device), "hipSetDevice"))
    return GTX_ERR_HIP;
  ScratchHold hold{*c, scratch_acquire(*c, st), st, false};
  CallScratch * s = hold.s;
  if (!s)
    return GTX_ERR_HIP;
  ScoreAcc a;
  a.n_samples = acc->n_samples;
  a.conn_cap = acc->conn_cap;
  a.log_score = acc->d_log_score;
  a.gt_cov = acc->d_gt_cov;
  a.hap_u32 = acc->d_hap_u32;
  a.stat_u64 = reinterpret_cast<unsigned long long *>(acc->d_stat_u64);
  a.stat_u32 = acc->d_stat_u32;
  a.conn_log = acc->d_conn_log;
  a.conn_count = acc->d_conn_count;
  a.conn_near = acc->d_conn_near;
  a.big_records = c->d_big_records;
  {
    // (the scoring kernel's table names a counter by its distance from the lowest of these: score_core.hpp)
    unsigned long long base = reinterpret_cast<unsigned long long>(a.log_score);
    for (void const * p : {static_cast<void const *>(a.gt_cov), static_cast<void const *>(a.hap_u32), static_cast<void const *>(a.stat_u64),
                           static_cast<void const *>(a.stat_u32), static_cast<void const *>(a.conn_near)})
      if (p)
        base = std::min(base, reinterpret_cast<unsigned long long>(p));
    a.combine_base = base & ~7ull;
  }
  a.compact = d_compact;
  a.compact_flags = d_compact ? d_task_flags : nullptr;
  a.ref_depth = c->params.is_sv_graph ? acc->d_ref_depth : nullptr;
  a.ref_depth_len = acc->ref_depth_len;
  if (a.ref_depth && a.ref_depth_len != c->graph.ref_order.back() + c->graph.ref_len.back() - c->graph.ref_order.front())
  {
    g_last_error = "gtx_score_batch: ref_depth_len is not the length of the region's reference (gtx_score_layout::ref_depth_len)";
    return GTX_ERR_ARG;
  }
  ScoreParams par{static_cast<uint32_t>(c->params.is_sv_graph != 0), static_cast<uint32_t>(c->params.hq_reads != 0),
                  static_cast<uint32_t>(c->params.is_segment_calling != 0), 0};
  uint32_t const blocks = (n_items + GTX_SCORE_THREADS - 1u) / GTX_SCORE_THREADS;
  bool const second_pass = s->d_score_state != nullptr;
  // the state of the second pass and the work queue's count: the set of four words the last call's triage kernel zeroed (zeroed
  // here after a call whose triage launch failed)
  uint32_t * score_state = nullptr, * score_state_next = nullptr;
  if (second_pass)
  {
    uint32_t const use = s->score_set ^ 1u;
    score_state = s->d_score_state + 4u * use;
    score_state_next = s->d_score_state + 4u * (use ^ 1u);
    if (!s->score_spare_clean && !hip_ok(hipMemsetAsync(score_state, 0, 4 * sizeof(uint32_t), st), "second-pass state + work count reset"))
      return GTX_ERR_HIP;
    s->score_set = use;
    s->score_spare_clean = false;
  }
  // work queue of stage 2: room for every item ([0] = count, [1..] = item indices)
  uint64_t cap = s->score_work_cap ? static_cast<uint64_t>(s->score_work_cap) + 1 : 0;
  if (!grow(s->d_score_work, cap, static_cast<uint64_t>(n_items) + 1, "score work queue"))
    return GTX_ERR_HIP;
  s->score_work_cap = static_cast<uint32_t>(cap - 1);
  uint32_t const * work_count = second_pass ? score_state + 2 : s->d_score_work;
  uint32_t const * work_queue = s->d_score_work + 1;
  if (d_work)
  {
    // (gtx_score_batch_queued: the first stage ran behind the alignment -- the queue is the caller's; the second pass' state for
    //  the next call on this scratch is zeroed by the launch that ends this call, below)
    work_count = d_work;
    work_queue = d_work + GTX_WORK_HEADER_WORDS;
  }
  else
  {
    if (!second_pass && !hip_ok(hipMemsetAsync(s->d_score_work, 0, sizeof(uint32_t), st), "score work queue reset"))
      return GTX_ERR_HIP;
    hipLaunchKernelGGL(gtx_score_triage_kernel, dim3((n_items + TRIAGE_THREADS * TRIAGE_PER_THREAD - 1) / (TRIAGE_THREADS * TRIAGE_PER_THREAD)), dim3(TRIAGE_THREADS), 0, st, d_items, n_items, d_records, rec_words, s->d_score_work + 1,
                       second_pass ? score_state + 2 : s->d_score_work, static_cast<uint32_t>(a.ref_depth != nullptr), d_task_flags, d_item_words, score_state_next);
    if (!hip_ok(hipGetLastError(), "gtx_score_triage_kernel launch"))
      return GTX_ERR_HIP;
    s->score_spare_clean = second_pass;
  }
  uint32_t const work_blocks = std::min<uint32_t>(blocks, static_cast<uint32_t>(c->n_cu > 0 ? c->n_cu : 256) * c->score_blocks_per_cu);
  hipLaunchKernelGGL(gtx_score_kernel, dim3(work_blocks), dim3(GTX_SCORE_THREADS), 0, st, c->dev_graph, par, d_items, work_queue, work_count,
                     d_records, rec_words, a, c->d_error_flag, second_pass ? s->d_score_queue : nullptr,
                     second_pass ? gtx_ctx::SCORE_QUEUE_CAP : 0u, score_state);
  if (!hip_ok(hipGetLastError(), "gtx_score_kernel launch"))
    return GTX_ERR_HIP;
  if (second_pass)
  {
    uint32_t * const zero_next = d_work ? score_state_next : nullptr; // (the set the next call on this scratch uses: nobody reads it in this call)
    if (c->has_wide_sites)
      hipLaunchKernelGGL(gtx_score_wide_kernel, dim3(gtx_ctx::SCORE_BIG_THREADS / 64), dim3(64), 0, st, c->dev_graph, par, d_items, d_records,
                         rec_words, a, c->d_error_flag, s->d_score_queue, gtx_ctx::SCORE_QUEUE_CAP, score_state,
                         static_cast<RecentHapWide *>(s->d_score_tables), zero_next);
    else
      hipLaunchKernelGGL(gtx_score_big_kernel, dim3(gtx_ctx::SCORE_BIG_THREADS / 64), dim3(64), 0, st, c->dev_graph, par, d_items, d_records,
                         rec_words, a, c->d_error_flag, s->d_score_queue, gtx_ctx::SCORE_QUEUE_CAP, score_state,
                         static_cast<RecentHap *>(s->d_score_tables), zero_next);
    if (!hip_ok(hipGetLastError(), "gtx_score_big_kernel launch"))
      return GTX_ERR_HIP;
    if (d_work)
      s->score_spare_clean = true;
  }
  return GTX_OK;
}

// Sequential replay of the cells that reached the saturation guard of explain_to_score (score_replay.hpp).
static int scores_replay(gtx_ctx * c, const gtx_score_item * d_items, uint32_t n_items, const uint32_t * d_records, uint32_t rec_words,
                         const gtx_score_buffers * acc, void * stream, uint64_t * n_replayed, uint64_t * n_unsupported, const uint32_t * d_compact,
                         const uint8_t * d_task_flags);

extern "C" int gtx_scores_replay(gtx_ctx * c, const gtx_score_item * d_items, uint32_t n_items, const uint32_t * d_records,
                                 uint32_t rec_words, const gtx_score_buffers * acc, void * stream, uint64_t * n_replayed,
                                 uint64_t * n_unsupported)
{
  return scores_replay(c, d_items, n_items, d_records, rec_words, acc, stream, n_replayed, n_unsupported, nullptr, nullptr);
}

extern "C" int gtx_scores_replay_compact(gtx_ctx * c, const gtx_score_item * d_items, uint32_t n_items, const uint32_t * d_records,
                                         uint32_t rec_words, const uint32_t * d_compact, const uint8_t * d_task_flags,
                                         const gtx_score_buffers * acc, void * stream, uint64_t * n_replayed, uint64_t * n_unsupported)
{
  if (!d_compact || !d_task_flags)
  {
    g_last_error = "gtx_scores_replay_compact: needs the compact records and their side array";
    return GTX_ERR_ARG;
  }
  return scores_replay(c, d_items, n_items, d_records, rec_words, acc, stream, n_replayed, n_unsupported, d_compact, d_task_flags);
}

// the calls of explain_to_score on the cells of `acc` that stand at the guard, logged by a pass over the items (any order)
static int replay_collect(gtx_ctx * c, const gtx_score_item * d_items, uint32_t n_items, const uint32_t * d_records, uint32_t rec_words,
                          const gtx_score_buffers * acc, hipStream_t st, const uint32_t * d_compact, const uint8_t * d_task_flags,
                          std::vector<ReplayEntry> & log, uint64_t & unsupported)
{
  log.clear();
  unsupported = 0;
  if (!hip_ok(hipSetDevice(c->device), "hipSetDevice") || !hip_ok(hipStreamSynchronize(st), "stream synchronize"))
    return GTX_ERR_HIP;
  HostGraph const & g = c->graph;
  uint64_t const n_cells = static_cast<uint64_t>(acc->n_samples) * g.n_hap;
  std::vector<uint32_t> cells(4 * n_cells);
  if (!hip_ok(hipMemcpy(cells.data(), acc->d_hap_u32, cells.size() * sizeof(uint32_t), hipMemcpyDeviceToHost), "cells"))
    return GTX_ERR_HIP;
  std::vector<uint32_t> marked((n_cells + 31) / 32, 0u);
  uint64_t n_marked = 0;
  for (uint64_t cell = 0; cell < n_cells; ++cell)
  {
    uint32_t const m = cells[4 * cell];
    if ((m & GTX_CELL_REPLAYED) || m < SATURATION_GUARD)
      continue;
    if (g.ref_nvar[cell % g.n_hap] > 64) // (the log keeps 64-bit explain sets)
    {
      ++unsupported;
      continue;
    }
    marked[cell >> 5] |= 1u << (cell & 31u);
    ++n_marked;
  }
  if (n_marked == 0 || n_items == 0)
    return GTX_OK;
  ScratchHold hold{*c, scratch_acquire(*c, st), st, false};
  CallScratch * s = hold.s;
  if (!s || !s->d_score_tables)
  {
    g_last_error = "gtx_scores_replay: the context has no second scoring pass (gtx_params::no_second_pass)";
    return s ? GTX_ERR_UNSUPPORTED : GTX_ERR_HIP;
  }
  uint32_t * d_marked = nullptr;
  uint32_t * d_count = nullptr;
  ReplayEntry * d_log = nullptr;
  uint32_t cap = 1u << 20;
  bool ok = dev_alloc(d_marked, marked.size(), "replay bitmap") && dev_alloc(d_count, 1, "replay count") &&
            hip_ok(hipMemcpy(d_marked, marked.data(), marked.size() * sizeof(uint32_t), hipMemcpyHostToDevice), "replay bitmap");
  for (int attempt = 0; ok && attempt < 2; ++attempt) // (a second launch when the log was too small)
  {
    ok = dev_alloc(d_log, cap, "replay log") && hip_ok(gtx::dev_zero(d_count, sizeof(uint32_t)), "replay count");
    if (!ok)
      break;
    ScoreAcc a;
    a.n_samples = acc->n_samples;
    a.conn_cap = 0;
    a.log_score = acc->d_log_score;
    a.gt_cov = acc->d_gt_cov;
    a.hap_u32 = acc->d_hap_u32;
    a.stat_u64 = reinterpret_cast<unsigned long long *>(acc->d_stat_u64);
    a.stat_u32 = acc->d_stat_u32;
    a.conn_log = acc->d_conn_log;
    a.conn_count = acc->d_conn_count;
    a.conn_near = acc->d_conn_near;
    a.big_records = c->d_big_records;
    a.compact = d_compact;
    a.compact_flags = d_compact ? d_task_flags : nullptr;
    a.replay_cells = d_marked;
    a.replay_log = d_log;
    a.replay_count = d_count;
    a.replay_cap = cap;
    ScoreParams par{static_cast<uint32_t>(c->params.is_sv_graph != 0), static_cast<uint32_t>(c->params.hq_reads != 0),
                    static_cast<uint32_t>(c->params.is_segment_calling != 0), 0};
    if (c->has_wide_sites)
      hipLaunchKernelGGL(gtx_score_replay_wide_kernel, dim3(gtx_ctx::SCORE_BIG_THREADS / 64), dim3(64), 0, st, c->dev_graph, par, d_items, n_items,
                         d_records, rec_words, a, c->d_error_flag, static_cast<RecentHapWide *>(s->d_score_tables));
    else
      hipLaunchKernelGGL(gtx_score_replay_kernel, dim3(gtx_ctx::SCORE_BIG_THREADS / 64), dim3(64), 0, st, c->dev_graph, par, d_items, n_items,
                         d_records, rec_words, a, c->d_error_flag, static_cast<RecentHap *>(s->d_score_tables));
    uint32_t wanted = 0;
    ok = hip_ok(hipGetLastError(), "gtx_score_replay_kernel launch") && hip_ok(hipStreamSynchronize(st), "replay") &&
         hip_ok(hipMemcpy(&wanted, d_count, sizeof(wanted), hipMemcpyDeviceToHost), "replay count");
    if (ok && wanted <= cap)
    {
      log.resize(wanted);
      ok = wanted == 0 || hip_ok(hipMemcpy(log.data(), d_log, static_cast<size_t>(wanted) * sizeof(ReplayEntry), hipMemcpyDeviceToHost), "replay log");
      break;
    }
    (void)gtx::dev_free(d_log);
    d_log = nullptr;
    cap = wanted;
    if (attempt == 1)
    {
      g_last_error = "gtx_scores_replay: the log kept growing";
      ok = false;
    }
  }
  for (void * p : {static_cast<void *>(d_marked), static_cast<void *>(d_count), static_cast<void *>(d_log)})
    if (p)
      (void)gtx::dev_free(p);
  return ok ? GTX_OK : GTX_ERR_HIP;
}

// the logged calls replayed one by one in call order (score_replay.hpp), the exact rows stored back into `acc`
static int replay_store(gtx_ctx * c, const gtx_score_buffers * acc, std::vector<ReplayEntry> & log, uint64_t * n_replayed)
{
  HostGraph const & g = c->graph;
  uint64_t const n_cells = static_cast<uint64_t>(acc->n_samples) * g.n_hap;
  for (ReplayEntry const & e : log)
    if (e.cell >= n_cells || g.ref_nvar[e.cell % g.n_hap] > 64)
    {
      g_last_error = "gtx_scores_replay_apply: an entry names a cell the block does not have";
      return GTX_ERR_ARG;
    }
  std::vector<ReplayedCell> const done = replay_cells(g, log);
  bool ok = true;
  for (ReplayedCell const & rc : done)
  {
    uint32_t const h = rc.cell % g.n_hap, sample = rc.cell / g.n_hap;
    uint32_t const head = rc.max_log_score | GTX_CELL_REPLAYED;
    ok = ok && hip_ok(hipMemcpy(acc->d_hap_u32 + 4ull * rc.cell, &head, sizeof(head), hipMemcpyHostToDevice), "replayed cell") &&
         hip_ok(hipMemcpy(acc->d_log_score + static_cast<uint64_t>(sample) * g.total_tri + g.tri_off[h], rc.log_score.data(),
                          rc.log_score.size() * sizeof(uint32_t), hipMemcpyHostToDevice),
                "replayed scores");
  }
  if (n_replayed)
    *n_replayed = done.size();
  return ok ? GTX_OK : GTX_ERR_HIP;
}

static int scores_replay(gtx_ctx * c, const gtx_score_item * d_items, uint32_t n_items, const uint32_t * d_records, uint32_t rec_words,
                         const gtx_score_buffers * acc, void * stream, uint64_t * n_replayed, uint64_t * n_unsupported, const uint32_t * d_compact,
                         const uint8_t * d_task_flags)
{
  if (!c || !acc || !acc->d_log_score || !acc->d_hap_u32 || (n_items && (!d_items || !d_records)))
  {
    g_last_error = "gtx_scores_replay: bad argument";
    return GTX_ERR_ARG;
  }
  if (c->device < 0)
  {
    g_last_error = "context was created without a device (libgtx has no CPU path)";
    return GTX_ERR_NO_DEVICE;
  }
  if (n_replayed)
    *n_replayed = 0;
  if (n_unsupported)
    *n_unsupported = 0;
  std::vector<ReplayEntry> log;
  uint64_t unsupported = 0;
  int rc = replay_collect(c, d_items, n_items, d_records, rec_words, acc, static_cast<hipStream_t>(stream), d_compact, d_task_flags, log, unsupported);
  if (n_unsupported)
    *n_unsupported = unsupported;
  if (rc != GTX_OK || log.empty())
    return rc;
  return replay_store(c, acc, log, n_replayed);
}

// ---- the same in two halves for reads sharded over ranks: every rank logs what ITS items did to the cells that stand
// at the guard in the block summed over all ranks, the hosts exchange the logs, and whoever makes the calls replays them all.
static_assert(sizeof(gtx_replay_entry) == sizeof(ReplayEntry), "gtx_replay_entry is ReplayEntry");
extern "C" int gtx_scores_replay_log(gtx_ctx * c, const gtx_score_item * d_items, uint32_t n_items, const uint32_t * d_records, uint32_t rec_words,
                                     const uint32_t * d_compact, const uint8_t * d_task_flags, const gtx_score_buffers * acc, uint32_t item_base,
                                     void * stream, gtx_replay_entry * out, uint64_t cap, uint64_t * n, uint64_t * n_unsupported)
{
  if (!c || !n || !acc || !acc->d_log_score || !acc->d_hap_u32 || (n_items && (!d_items || !d_records)) || (cap && !out) || (d_compact && !d_task_flags))
  {
    g_last_error = "gtx_scores_replay_log: bad argument";
    return GTX_ERR_ARG;
  }
  *n = 0;
  if (n_unsupported)
    *n_unsupported = 0;
  if (c->device < 0)
  {
    g_last_error = "context was created without a device (libgtx has no CPU path)";
    return GTX_ERR_NO_DEVICE;
  }
  std::vector<ReplayEntry> log;
  uint64_t unsupported = 0;
  int const rc = replay_collect(c, d_items, n_items, d_records, rec_words, acc, static_cast<hipStream_t>(stream), d_compact, d_task_flags, log, unsupported);
  if (n_unsupported)
    *n_unsupported = unsupported;
  if (rc != GTX_OK)
    return rc;
  *n = log.size();
  if (log.size() > cap)
  {
    g_last_error = "gtx_scores_replay_log: " + std::to_string(log.size()) + " entries, room for " + std::to_string(cap);
    return GTX_ERR_CAPACITY;
  }
  for (ReplayEntry & e : log)
    e.item += item_base; // (the item's place in the region's sequence over all ranks)
  if (!log.empty())
    std::memcpy(out, log.data(), log.size() * sizeof(ReplayEntry));
  return GTX_OK;
}

extern "C" int gtx_scores_replay_apply(gtx_ctx * c, const gtx_score_buffers * acc, const gtx_replay_entry * entries, uint64_t n_entries, void * stream,
                                       uint64_t * n_replayed)
{
  if (!c || !acc || !acc->d_log_score || !acc->d_hap_u32 || (n_entries && !entries))
  {
    g_last_error = "gtx_scores_replay_apply: bad argument";
    return GTX_ERR_ARG;
  }
  if (n_replayed)
    *n_replayed = 0;
  if (c->device < 0)
  {
    g_last_error = "context was created without a device (libgtx has no CPU path)";
    return GTX_ERR_NO_DEVICE;
  }
  if (n_entries == 0)
    return GTX_OK;
  if (!hip_ok(hipSetDevice(c->device), "hipSetDevice") || !hip_ok(hipStreamSynchronize(static_cast<hipStream_t>(stream)), "stream synchronize"))
    return GTX_ERR_HIP;
  std::vector<ReplayEntry> log(n_entries);
  std::memcpy(log.data(), entries, n_entries * sizeof(ReplayEntry));
  return replay_store(c, acc, log, n_replayed);
}

extern "C" int gtx_ctx_big_records(gtx_ctx * c, const uint32_t ** d_words, uint64_t * capacity_words, uint64_t * used_words,
                                   uint64_t * tasks)
{
  if (!c || !d_words || !capacity_words)
    return GTX_ERR_ARG;
  *d_words = c->d_big_records;
  *capacity_words = c->d_big_records ? c->big_record_words : 0;
  if (used_words)
    *used_words = 0;
  if (tasks)
    *tasks = 0;
  if ((used_words || tasks) && c->d_arena_cursor)
  {
    if (!hip_ok(hipSetDevice(c->device), "hipSetDevice"))
      return GTX_ERR_HIP;
    unsigned long long used = 0;
    if (!hip_ok(hipMemcpy(&used, c->d_arena_cursor, sizeof(used), hipMemcpyDeviceToHost), "arena cursor"))
      return GTX_ERR_HIP;
    if (used_words)
      *used_words = std::min<uint64_t>(used, c->big_record_words);
    CallScratch * s = nullptr;
    {
      std::lock_guard<std::mutex> lock(c->pool_mutex);
      s = c->last_align;
    }
    if (tasks && s && s->d_big_state)
    {
      uint32_t queued = 0;
      if (!hip_ok(hipMemcpy(&queued, s->d_big_state, sizeof(queued), hipMemcpyDeviceToHost), "second-pass state"))
        return GTX_ERR_HIP;
      *tasks = std::min<uint32_t>(queued, s->big_task_cap);
    }
  }
  return GTX_OK;
}

extern "C" int gtx_ctx_exact_pass_tasks(gtx_ctx * c, uint64_t * out)
{
  if (!c || !out)
    return GTX_ERR_ARG;
  out[0] = out[1] = out[2] = out[3] = 0;
  if (c->device < 0)
    return GTX_ERR_NO_DEVICE;
  CallScratch * s = nullptr;
  {
    std::lock_guard<std::mutex> lock(c->pool_mutex);
    s = c->last_align;
  }
  if (!s || !s->d_exact_state)
    return GTX_OK;
  if (!hip_ok(hipSetDevice(c->device), "hipSetDevice"))
    return GTX_ERR_HIP;
  uint32_t st[32]; // 8 words per launch (small parts, large parts, whole slab), then what the last handed on (nothing takes it: the count is what is left)
  if (!hip_ok(hipDeviceSynchronize(), "exact-pass state") || !hip_ok(hipMemcpy(st, s->d_exact_state, sizeof(st), hipMemcpyDeviceToHost), "exact-pass state"))
    return GTX_ERR_HIP;
  out[0] = st[0];
  out[1] = st[8];
  out[2] = st[16];
  out[3] = st[24] + st[3] + st[11] + st[19]; // (+ tasks a full queue dropped)
  return GTX_OK;
}

extern "C" int gtx_ctx_big_records_rewind(gtx_ctx * c, void * stream)
{
  if (!c)
    return GTX_ERR_ARG;
  if (c->d_arena_cursor &&
      !hip_ok(hipMemsetAsync(c->d_arena_cursor, 0, sizeof(unsigned long long), static_cast<hipStream_t>(stream)), "arena rewind"))
    return GTX_ERR_HIP;
  return GTX_OK;
}

extern "C" int gtx_calls_batch(gtx_ctx * c, const gtx_score_buffers * acc, uint8_t * d_phred, gtx_sample_call * d_calls, void * stream)
{
  if (!c || !acc || !d_phred || !d_calls || !acc->d_log_score || !acc->d_gt_cov || !acc->d_hap_u32)
  {
    g_last_error = "gtx_calls_batch: bad argument";
    return GTX_ERR_ARG;
  }
  if (c->device < 0)
  {
    g_last_error = "context was created without a device (libgtx has no CPU path)";
    return GTX_ERR_NO_DEVICE;
  }
  uint64_t const cells = static_cast<uint64_t>(acc->n_samples) * c->dev_graph.n_hap;
  if (cells == 0)
    return GTX_OK;
  if (!hip_ok(hipSetDevice(c->device), "hipSetDevice"))
    return GTX_ERR_HIP;
  hipLaunchKernelGGL(gtx_calls_kernel, dim3(static_cast<uint32_t>((cells + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     c->dev_graph, acc->n_samples, acc->d_log_score, acc->d_gt_cov, acc->d_hap_u32, d_phred, d_calls);
  if (!hip_ok(hipGetLastError(), "gtx_calls_kernel launch"))
    return GTX_ERR_HIP;
  return GTX_OK;
}

extern "C" int gtx_ctx_error_count(gtx_ctx * c, uint32_t * out)
{
  if (!c || !out)
    return GTX_ERR_ARG;
  *out = 0;
  if (c->device < 0)
    return GTX_OK;
  if (!hip_ok(hipMemcpy(out, c->d_error_flag, sizeof(uint32_t), hipMemcpyDeviceToHost), "error flag"))
    return GTX_ERR_HIP;
  return GTX_OK;
}

// How many of the 2 * n_reads record slots of gtx_align_batch hold a table-overflow status instead of a result (a record arena
// that was full, an exact-pass slab too small, a read beyond the supported length): every one of them is a read the
// accumulators lack.  Synchronises with `stream`.
extern "C" int gtx_records_failed(gtx_ctx * c, const uint32_t * d_records, uint32_t rec_words, uint64_t n_reads, void * stream, uint64_t * out)
{
  if (!c || !out || rec_words < 8 || (n_reads && !d_records))
    return GTX_ERR_ARG;
  *out = 0;
  if (c->device < 0)
  {
    g_last_error = "context was created without a device (libgtx has no CPU path)";
    return GTX_ERR_NO_DEVICE;
  }
  if (n_reads == 0)
    return GTX_OK;
  if (!hip_ok(hipSetDevice(c->device), "hipSetDevice"))
    return GTX_ERR_HIP;
  hipStream_t const st = static_cast<hipStream_t>(stream);
  void * d_count = nullptr;
  if (!hip_ok(gtx::dev_malloc(&d_count, sizeof(unsigned long long)), "failed-record counter"))
    return GTX_ERR_HIP;
  unsigned long long n = 0;
  bool ok = hip_ok(hipMemsetAsync(d_count, 0, sizeof n, st), "failed-record counter");
  if (ok)
  {
    uint64_t const slots = 2ull * n_reads;
    uint32_t const blocks = static_cast<uint32_t>(std::min<uint64_t>((slots + 255u) / 256u, 4096u));
    hipLaunchKernelGGL(gtx_records_failed_kernel, dim3(blocks), dim3(256), 0, st, d_records, rec_words, slots, static_cast<unsigned long long *>(d_count));
    ok = hip_ok(hipGetLastError(), "gtx_records_failed_kernel launch") &&
         hip_ok(hipMemcpyAsync(&n, d_count, sizeof n, hipMemcpyDeviceToHost, st), "failed-record counter") &&
         hip_ok(hipStreamSynchronize(st), "failed-record counter");
  }
  (void)gtx::dev_free(d_count);
  if (!ok)
    return GTX_ERR_HIP;
  *out = n;
  return GTX_OK;
}

int gtx::records_clear_enqueue(gtx_ctx * c, uint32_t * d_records, uint32_t rec_words, uint64_t n_reads, void * stream)
{
  if (!c || rec_words < 8 || (n_reads && !d_records) || c->device < 0)
    return GTX_ERR_ARG;
  if (n_reads == 0)
    return GTX_OK;
  uint64_t const slots = 2ull * n_reads;
  uint32_t const blocks = static_cast<uint32_t>(std::min<uint64_t>((slots + 255u) / 256u, 4096u));
  hipLaunchKernelGGL(gtx_records_clear_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), d_records, rec_words, slots);
  return hip_ok(hipGetLastError(), "gtx_records_clear_kernel launch") ? GTX_OK : GTX_ERR_HIP;
}

int gtx::records_failed_enqueue(gtx_ctx * c, const uint32_t * d_records, uint32_t rec_words, uint64_t n_reads, void * stream, unsigned long long * d_count)
{
  if (!c || !d_count || rec_words < 8 || (n_reads && !d_records) || c->device < 0)
    return GTX_ERR_ARG;
  hipStream_t const st = static_cast<hipStream_t>(stream);
  if (!hip_ok(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), st), "failed-record counter"))
    return GTX_ERR_HIP;
  if (n_reads == 0)
    return GTX_OK;
  uint64_t const slots = 2ull * n_reads;
  uint32_t const blocks = static_cast<uint32_t>(std::min<uint64_t>((slots + 255u) / 256u, 4096u));
  hipLaunchKernelGGL(gtx_records_failed_kernel, dim3(blocks), dim3(256), 0, st, d_records, rec_words, slots, d_count);
  return hip_ok(hipGetLastError(), "gtx_records_failed_kernel launch") ? GTX_OK : GTX_ERR_HIP;
}

// the general pass' task log of the profiling build (no entry in the normal build): up to `cap` entries of 16 words into `out`,
// their number into *n; the log is emptied
extern "C" int gtx_ctx_profile_log(gtx_ctx * c, uint64_t * out, uint64_t cap, uint64_t * n)
{
  if (!c || !n || (cap && !out))
    return GTX_ERR_ARG;
  *n = 0;
  if (c->device < 0 || !c->dev_graph.prof || PROF_LOG_ENTRIES == 0)
    return GTX_OK;
  unsigned long long count = 0;
  if (!hip_ok(hipSetDevice(c->device), "hipSetDevice") || !hip_ok(hipDeviceSynchronize(), "profile log") ||
      !hip_ok(hipMemcpy(&count, c->dev_graph.prof + 32, sizeof count, hipMemcpyDeviceToHost), "profile log"))
    return GTX_ERR_HIP;
  uint64_t const have = std::min<uint64_t>(std::min<uint64_t>(count, PROF_LOG_ENTRIES), cap);
  if (have && !hip_ok(hipMemcpy(out, c->dev_graph.prof + 40, have * PROF_LOG_ENTRY * sizeof(uint64_t), hipMemcpyDeviceToHost), "profile log"))
    return GTX_ERR_HIP;
  if (!hip_ok(gtx::dev_zero(c->dev_graph.prof + 32, sizeof(unsigned long long)), "profile log"))
    return GTX_ERR_HIP;
  *n = have;
  return GTX_OK;
}

// phase cycle counters of the profiling build (all zero in the normal build); out[32]
extern "C" int gtx_ctx_profile(gtx_ctx * c, uint64_t * out)
{
  if (!c || !out)
    return GTX_ERR_ARG;
  std::memset(out, 0, 32 * sizeof(uint64_t));
  if (c->device < 0 || !c->dev_graph.prof)
    return GTX_OK;
  if (!hip_ok(hipMemcpy(out, c->dev_graph.prof, 32 * sizeof(uint64_t), hipMemcpyDeviceToHost), "profile counters"))
    return GTX_ERR_HIP;
  return GTX_OK;
}
