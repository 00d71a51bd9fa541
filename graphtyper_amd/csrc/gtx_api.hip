// gtx_api.hip -- the HIP kernels (gfx950) and the device half of the C ABI declared in include/gtx.h.
//
// align kernel : one wavefront (= one 64-thread workgroup, so the workgroup barrier is the wave barrier and every
//                wave owns its LDS workspace) per (read, orientation) task; persistent grid-stride over tasks.
// score kernel : one thread per score item (an unpaired read or a mate pair); integer atomics into flat accumulators.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "gtx_ctx.hpp"
#include "score_core.hpp"

namespace gtx
{
struct WaveHip
{
  template <class T>
  struct PerLane
  {
    T v;
    __device__ inline T & operator[](uint32_t) { return v; }
    __device__ inline T const & operator[](uint32_t) const { return v; }
  };
  template <class F>
  static __device__ inline void lanes(F && f)
  {
    f(threadIdx.x & 63u);
  }
  static __device__ inline bool leader() { return (threadIdx.x & 63u) == 0; }
  // value known to be equal on all lanes -> scalar register
  static __device__ inline uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
  static __device__ inline int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
  static __device__ inline bool uni(bool v) { return __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v)) != 0; }
  static __device__ inline uint64_t uni(uint64_t v)
  {
    uint32_t const lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
    uint32_t const hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
    return (static_cast<uint64_t>(hi) << 32) | lo;
  }
  // Orders the leader's LDS writes before the other lanes' reads.  All 64 lanes belong to one wavefront whose LDS
  // instructions are issued and serviced in program order, so no hardware wait is needed: the wavefront-scope fences
  // only stop the compiler from moving or merging LDS accesses across this point (GTX_HARD_SYNC=1 at build time
  // falls back to a real workgroup barrier for A/B checks).
  static __device__ inline void lds_sync()
  {
#ifdef GTX_HARD_SYNC
    __syncthreads();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
  }
  static __device__ inline uint64_t ballot(PerLane<bool> const & p) { return __ballot(p.v); }
  static __device__ inline uint32_t sum(PerLane<uint32_t> const & p)
  {
    uint32_t x = p.v;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
      x += __shfl_xor(x, d);
    return x;
  }
  static __device__ inline void excl_scan(PerLane<uint32_t> const & in, PerLane<uint32_t> & out, uint32_t & total)
  {
    uint32_t x = in.v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
      uint32_t const y = __shfl_up(x, d);
      if ((threadIdx.x & 63u) >= static_cast<uint32_t>(d))
        x += y;
    }
    total = __shfl(x, 63);
    out.v = x - in.v;
  }
  static __device__ inline unsigned long long clock() { return clock64(); }
  static __device__ inline uint32_t atomic_add_u32(uint32_t * p, uint32_t v) { return atomicAdd(p, v); }
  static __device__ inline void atomic_add_u64(unsigned long long * p, unsigned long long v) { atomicAdd(p, v); }
};

#ifndef GTX_TASK_CHUNK
#define GTX_TASK_CHUNK 16
#endif
constexpr uint32_t TASK_CHUNK = GTX_TASK_CHUNK; // reads a wave claims per visit to the task counter

__global__ __launch_bounds__(64) void gtx_align_kernel(GraphView g, IndexView ix, uint8_t const * __restrict__ seq,
                                                       uint32_t seq_stride, gtx_read_meta const * __restrict__ meta,
                                                       uint32_t n_reads, uint32_t * __restrict__ records, uint32_t rec_words,
                                                       uint32_t force_both, uint32_t * task_counter)
{
  __shared__ AlignWorkspace ws;
  __shared__ uint32_t task_base;
#ifdef GTX_PAD_LDS // occupancy experiment: waste LDS to lower the number of resident waves
  __shared__ uint32_t lds_pad[GTX_PAD_LDS / 4];
  if (n_reads == 0xFFFFFFFFu)
    lds_pad[threadIdx.x] = 1;
#endif
  // Reads are claimed dynamically (one atomic per TASK_CHUNK reads): the grid is sized to what is resident at once and
  // reads differ in cost (mismatches, ambiguous bases, the optional reverse orientation), a static split leaves CUs idle.
  for (;;)
  {
    if ((threadIdx.x & 63u) == 0)
      task_base = atomicAdd(task_counter, TASK_CHUNK);
    __syncthreads();
    uint32_t const base = task_base;
    __syncthreads();
    if (base >= n_reads)
      break;
    uint32_t const end = base + TASK_CHUNK < n_reads ? base + TASK_CHUNK : n_reads;
    for (uint32_t read = base; read < end; ++read)
    {
      gtx_read_meta const m = meta[read];
      uint32_t const len = m.l_qseq;
      // align_read (alignment.cpp:331-363): reads shorter than 2K-1 stay unaligned; the reverse orientation is only
      // computed for reads that are not part of a concordant pair
      bool const too_short = len < 2 * K - 1, too_long = len > AlignCfg::MAX_READ;
      bool const rev = needs_reverse(m, force_both != 0);
      for (uint32_t orient = 0; orient < 2; ++orient)
      {
        uint32_t * rec = records + (static_cast<uint64_t>(read) * 2 + orient) * rec_words;
        if (too_short || too_long || (orient == 1 && !rev))
        {
          if ((threadIdx.x & 63u) == 0)
          {
            rec[0] = too_long ? (static_cast<uint32_t>(GTX_ST_RECORD_OVERFLOW) << 16) : 0u;
            rec[1] = len << 16;
          }
          continue;
        }
        align_one<WaveHip>(g, ix, ws, seq + static_cast<uint64_t>(read) * seq_stride, len, orient == 1, rec, rec_words);
      }
    }
  }
}

__global__ __launch_bounds__(256) void gtx_score_kernel(GraphView g, ScoreParams par, gtx_score_item const * __restrict__ items,
                                                        uint32_t n_items, uint32_t const * __restrict__ records,
                                                        uint32_t rec_words, ScoreAcc acc, uint32_t * error_flag)
{
  uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_items)
    return;
  gtx_score_item const it = items[i];
  score_item<WaveHip>(g, par, it, records, rec_words, acc, error_flag);
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
  if (!hip_ok(hipMalloc(&p, bytes), what))
    return false;
  owned.push_back(p);
  if (n && !hip_ok(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice), what))
    return false;
  dst = static_cast<T const *>(p);
  return true;
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
  HostGraph const & h = c.graph;
  GraphView v = h.view();
  bool ok = true;
  ok = ok && upload(c.dev_allocs, v.ref_order, h.ref_order.data(), h.ref_order.size(), "ref_order");
  ok = ok && upload(c.dev_allocs, v.ref_len, h.ref_len.data(), h.ref_len.size(), "ref_len");
  ok = ok && upload(c.dev_allocs, v.ref_dna, h.ref_dna.data(), h.ref_dna.size(), "ref_dna");
  ok = ok && upload(c.dev_allocs, v.ref_nvar, h.ref_nvar.data(), h.ref_nvar.size(), "ref_nvar");
  ok = ok && upload(c.dev_allocs, v.ref_first_var, h.ref_first_var.data(), h.ref_first_var.size(), "ref_first_var");
  ok = ok && upload(c.dev_allocs, v.var_order, h.var_order.data(), h.var_order.size(), "var_order");
  ok = ok && upload(c.dev_allocs, v.var_len, h.var_len.data(), h.var_len.size(), "var_len");
  ok = ok && upload(c.dev_allocs, v.var_dna, h.var_dna.data(), h.var_dna.size(), "var_dna");
  ok = ok && upload(c.dev_allocs, v.var_out_ref, h.var_out_ref.data(), h.var_out_ref.size(), "var_out_ref");
  ok = ok && upload(c.dev_allocs, v.site_ref_reach, h.site_ref_reach.data(), h.site_ref_reach.size(), "site_ref_reach");
  ok = ok && upload(c.dev_allocs, v.site_special_base, h.site_special_base.data(), h.site_special_base.size(), "site_special_base");
  ok = ok && upload(c.dev_allocs, v.special_ref_reach, h.special_ref_reach.data(), h.special_ref_reach.size(), "special_ref_reach");
  ok = ok && upload(c.dev_allocs, v.special_actual, h.special_actual.data(), h.special_actual.size(), "special_actual");
  ok = ok && upload(c.dev_allocs, v.pos_bucket, h.pos_bucket.data(), h.pos_bucket.size(), "pos_bucket");
  ok = ok && upload(c.dev_allocs, v.dna, h.codes.data(), h.codes.size(), "dna codes");
  ok = ok && upload(c.dev_allocs, v.tri_off, h.tri_off.data(), h.tri_off.size(), "tri_off");
  ok = ok && upload(c.dev_allocs, v.allele_off, h.allele_off.data(), h.allele_off.size(), "allele_off");
  IndexView ix{};
  ix.log2_cap = c.index.log2_cap;
  ix.max_index_labels = static_cast<uint32_t>(c.params.max_index_labels);
  ok = ok && upload(c.dev_allocs, ix.slots, c.index.slots.data(), c.index.slots.size(), "index slots");
  ok = ok && upload(c.dev_allocs, ix.labels, c.index.dev_labels.data(), c.index.dev_labels.size(), "index labels");
  ix.h_log2_cap = c.index.h_log2_cap;
  ix.half_bucket_cap = HALF_BUCKET_CAP;
  if (char const * e = std::getenv("GTX_HALF_BUCKET_CAP")) // A/B switch for benchmarking: 0 = probe the 96 neighbours directly
    ix.half_bucket_cap = static_cast<uint32_t>(std::min<long>(std::max<long>(std::atol(e), 0), HALF_BUCKET_CAP));
  ok = ok && upload(c.dev_allocs, ix.hslots, c.index.hslots.data(), c.index.hslots.size(), "half-key slots");
  ok = ok && upload(c.dev_allocs, ix.hlist, c.index.hlist.data(), c.index.hlist.size(), "half-key buckets");
  void * pf = nullptr;
  ok = ok && hip_ok(hipMalloc(&pf, 32 * sizeof(unsigned long long)), "profile counters");
  if (ok)
  {
    c.dev_allocs.push_back(pf);
    v.prof = static_cast<unsigned long long *>(pf);
    ok = hip_ok(hipMemset(pf, 0, 32 * sizeof(unsigned long long)), "profile counters");
  }
  void * tc = nullptr;
  ok = ok && hip_ok(hipMalloc(&tc, gtx_ctx::N_TASK_COUNTERS * sizeof(uint32_t)), "task counters");
  if (ok)
  {
    c.dev_allocs.push_back(tc);
    c.d_task_counters = static_cast<uint32_t *>(tc);
  }
  void * ef = nullptr;
  ok = ok && hip_ok(hipMalloc(&ef, sizeof(uint32_t)), "error flag");
  if (ok)
  {
    c.dev_allocs.push_back(ef);
    c.d_error_flag = static_cast<uint32_t *>(ef);
    ok = hip_ok(hipMemset(ef, 0, sizeof(uint32_t)), "error flag");
  }
  if (!ok)
    return GTX_ERR_HIP;
  c.dev_graph = v;
  c.dev_index = ix;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess)
    c.n_cu = prop.multiProcessorCount;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gtx_align_kernel, 64, 0) == hipSuccess && per_cu > 0)
    c.align_blocks_per_cu = per_cu;
  return GTX_OK;
}

void ctx_release_device(gtx_ctx & c)
{
  if (c.device >= 0)
    (void)hipSetDevice(c.device);
  for (void * p : c.dev_allocs)
    (void)hipFree(p);
  c.dev_allocs.clear();
}

} // namespace gtx

using namespace gtx;

extern "C" int gtx_align_batch(gtx_ctx * c, const uint8_t * d_seq, uint32_t seq_stride, const gtx_read_meta * d_meta,
                               uint32_t n_reads, uint32_t * d_records, uint32_t rec_words, void * stream)
{
  if (!c || !d_seq || !d_meta || !d_records || rec_words < 8)
  {
    g_last_error = "gtx_align_batch: bad argument";
    return GTX_ERR_ARG;
  }
  if (c->device < 0)
  {
    g_last_error = "context was created without a device (libgtx has no CPU path)";
    return GTX_ERR_NO_DEVICE;
  }
  if (n_reads == 0)
    return GTX_OK;
  // as many single-wave workgroups as are resident at once (LDS bound); they pull reads from a shared counter
  uint32_t const max_blocks = static_cast<uint32_t>(c->n_cu > 0 ? c->n_cu : 256) * static_cast<uint32_t>(c->align_blocks_per_cu);
  uint64_t const chunks = (static_cast<uint64_t>(n_reads) + TASK_CHUNK - 1) / TASK_CHUNK;
  uint32_t const blocks = static_cast<uint32_t>(chunks < max_blocks ? chunks : max_blocks);
  uint32_t * counter = c->d_task_counters + (c->launch_seq.fetch_add(1) % gtx_ctx::N_TASK_COUNTERS);
  if (!hip_ok(hipMemsetAsync(counter, 0, sizeof(uint32_t), static_cast<hipStream_t>(stream)), "task counter reset"))
    return GTX_ERR_HIP;
  hipLaunchKernelGGL(gtx_align_kernel, dim3(blocks), dim3(64), 0, static_cast<hipStream_t>(stream), c->dev_graph, c->dev_index,
                     d_seq, seq_stride, d_meta, n_reads, d_records, rec_words,
                     static_cast<uint32_t>(c->params.force_align_both_orientations != 0), counter);
  if (!hip_ok(hipGetLastError(), "gtx_align_kernel launch"))
    return GTX_ERR_HIP;
  return GTX_OK;
}

extern "C" int gtx_score_batch(gtx_ctx * c, const gtx_score_item * d_items, uint32_t n_items, const uint32_t * d_records,
                               uint32_t rec_words, const gtx_score_buffers * acc, void * stream)
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
  ScoreParams par{static_cast<uint32_t>(c->params.is_sv_graph != 0), static_cast<uint32_t>(c->params.hq_reads != 0),
                  static_cast<uint32_t>(c->params.is_segment_calling != 0), 0};
  uint32_t const blocks = (n_items + 255u) / 256u;
  hipLaunchKernelGGL(gtx_score_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), c->dev_graph, par, d_items,
                     n_items, d_records, rec_words, a, c->d_error_flag);
  if (!hip_ok(hipGetLastError(), "gtx_score_kernel launch"))
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
