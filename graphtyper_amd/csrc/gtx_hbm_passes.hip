// gtx_hbm_passes.hip -- the alignment passes behind the general one: the same algorithm (align_core.inl) over tables that
// live in HBM.  gtx_align_big_kernel (512 paths, 2 048 labels per k-mer), gtx_align_wide_kernel (allele sets of
// GTX_WIDE_MASK_WORDS words) and the exact pass, gtx_align_exact(_wide)_kernel, whose tables have no fixed size.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "gtx_ctx.hpp"
#include "wave_hip.hpp"
#include "align_core.hpp"
#include "gtx_hbm_passes.hpp"

namespace gtx
{
// Second pass over the queued (read, orientation) tasks: same algorithm instantiated over tables large enough for what
// the reference's own limits admit; one workspace in HBM per workgroup.  The queue is usually empty or tiny.
// A graph with a site of more than 64 alleles has a further pass of the same shape behind it (NS = wide: allele sets of
// GTX_WIDE_MASK_WORDS words, a larger table of walk candidates: one round of a walk branches into every allele of a site):
// a task that met an allele number >= 64 or overflowed a table is handed on to it (next_tasks / next_state).
// (registers for four wavefronts per SIMD -- 128 instead of the 193 the compiler takes when left alone: the pass has no task on
//  most batches, but every one of its workgroups has to be PLACED before it can see that, and beside another batch's kernels
//  a wavefront of 196 registers waited 0.14 ms for room; what it spills only matters on the rare batch that needs the pass)
#ifndef GTX_HBM_PASS_WAVES
#define GTX_HBM_PASS_WAVES 4
#endif
#define GTX_HBM_PASS_ATTR __attribute__((amdgpu_waves_per_eu(GTX_HBM_PASS_WAVES, GTX_HBM_PASS_WAVES)))
#define GTX_HBM_PASS_KERNEL(NAME, NS)                                                                                              \
  __global__ __launch_bounds__(64) GTX_HBM_PASS_ATTR void NAME(GraphView g, IndexView ix, uint8_t const * __restrict__ seq, uint32_t seq_stride, \
                                             gtx_read_meta const * __restrict__ meta, uint32_t * __restrict__ records,             \
                                             uint32_t rec_words, uint32_t const * __restrict__ big_tasks, uint32_t big_task_cap,   \
                                             uint32_t * big_state, NS::AlignWorkspace * workspaces, uint32_t * __restrict__ arena, \
                                             unsigned long long arena_words, unsigned long long * arena_cursor,                    \
                                             uint32_t * __restrict__ next_tasks, uint32_t next_cap, uint32_t * next_state)         \
  {                                                                                                                                \
    if (big_state[0] == 0) /* (nearly every batch: nothing reached this pass -- leave before the workspace is touched) */          \
      return;                                                                                                                      \
    NS::AlignWorkspace & ws = workspaces[blockIdx.x];                                                                              \
    /* the dense start / end tables of the chaining's searches live in LDS (align_core.inl: PpKeyTables) */                       \
    __shared__ uint32_t s_pp_keys[2][NS::AlignCfg::MAXPP];                                                                         \
    __shared__ uint64_t s_pp_bits[NS::AlignCfg::MAXPP / 64 + 1];                                                                   \
    ws.pp_start = s_pp_keys[0];                                                                                                    \
    ws.pp_end = s_pp_keys[1];                                                                                                      \
    ws.bits_pp = s_pp_bits;                                                                                                        \
    WaveHipMem::mem_sync();                                                                                                        \
    GTX_HBM_PASS_BODY(NS)                                                                                                          \
  }

#define GTX_HBM_PASS_BODY(NS)                                                                                                      \
    GTX_HBM_PASS_PROF_INIT                                                                                                         \
    uint32_t const queued = big_state[0] < big_task_cap ? big_state[0] : big_task_cap;                                             \
    for (;;)                                                                                                                       \
    {                                                                                                                              \
      uint32_t const t = wave_claim(big_state + 1, 1u);                                                                            \
      if (t >= queued)                                                                                                             \
        break;                                                                                                                     \
      uint32_t const task = WaveHip::uni(big_tasks[t]), read = task >> 1, orient = task & 1u;                                      \
      uint32_t const len = WaveHip::uni(static_cast<uint32_t>(meta[read].l_qseq));                                                 \
      uint32_t * rec = records + static_cast<uint64_t>(task) * rec_words;                                                          \
      uint32_t np = 0, longest = 0, ext = 0;                                                                                       \
      uint32_t status =                                                                                                            \
        NS::align_paths<WaveHipMem>(g, ix, ws, seq + static_cast<uint64_t>(read) * seq_stride, len, orient == 1, np, longest);     \
      status = WaveHip::uni(status);                                                                                               \
      np = WaveHip::uni(np);                                                                                                       \
      longest = WaveHip::uni(longest);                                                                                             \
      if (status && next_tasks && (threadIdx.x & 63u) == 0) /* an allele >= 64, or a table the next pass has larger */            \
      {                                                                                                                            \
        uint32_t const slot = atomicAdd(next_state, 1u);                                                                           \
        if (slot < next_cap)                                                                                                       \
          next_tasks[slot] = task;                                                                                                 \
        else                                                                                                                       \
          atomicAdd(next_state + 3, 1u);                                                                                           \
      }                                                                                                                            \
      status &= ~GTX_ST_WIDE_ALLELE;                                                                                               \
      uint32_t * body = rec + 2;                                                                                                   \
      unsigned long long off = 0;                                                                                                  \
      if (status)                                                                                                                  \
        np = 0;                                                                                                                    \
      else                                                                                                                         \
      {                                                                                                                            \
        uint32_t const size = WaveHip::uni(NS::record_size<WaveHipMem>(NS::Here{}, ws, np));                                       \
        if (size > rec_words)                                                                                                      \
        {                                                                                                                          \
          /* long record: room in the arena; the slot keeps the header and the offset */                                           \
          off = wave_claim64(arena_cursor, size - 2);                                                                              \
          if (off + (size - 2) > arena_words || off + (size - 2) > 0xFFFFFFFFull)                                                  \
          {                                                                                                                        \
            status = GTX_ST_RECORD_OVERFLOW;                                                                                       \
            np = 0;                                                                                                                \
          }                                                                                                                        \
          else                                                                                                                     \
          {                                                                                                                        \
            body = arena + off;                                                                                                    \
            ext = GTX_ST_EXTERNAL;                                                                                                 \
          }                                                                                                                        \
        }                                                                                                                          \
      }                                                                                                                            \
      uint32_t const has_var = NS::write_record_body<WaveHipMem>(NS::Here{}, ws, np, body);                                        \
      if ((threadIdx.x & 63u) == 0)                                                                                                \
      {                                                                                                                            \
        rec[0] = np | ((status | ext) << 16);                                                                                      \
        rec[1] = (np == 0 ? 0 : longest) | (len << 16) | (np == 0 ? 0u : has_var);                                                 \
        if (ext)                                                                                                                   \
          rec[2] = static_cast<uint32_t>(off);                                                                                     \
      }                                                                                                                            \
      WaveHipMem::mem_sync();                                                                                                      \
    }

// The exact pass (align_core.hpp: namespace exact): the same body over a workspace whose tables are cut out of the
// workgroup's part of a slab of HBM at run time.  Launched twice: EXACT_PARTS workgroups with a part each, then one workgroup
// with the whole slab for what did not fit a part (its queue is the first launch's next_tasks).
// (at most one workgroup of this pass per CU, usually none with work: no register diet -- all the registers a wavefront can name)
#ifdef GTX_EXACT_PASS_WAVES // (A/B builds: registers for that many wavefronts per SIMD)
#define GTX_EXACT_PASS_ATTR __attribute__((amdgpu_waves_per_eu(GTX_EXACT_PASS_WAVES, GTX_EXACT_PASS_WAVES)))
#else
#define GTX_EXACT_PASS_ATTR __attribute__((amdgpu_waves_per_eu(1, 2)))
#endif
#ifndef GTX_EXACT_LDS_KEYS
#define GTX_EXACT_LDS_KEYS 4096
#endif
constexpr uint32_t EXACT_LDS_KEYS = GTX_EXACT_LDS_KEYS; // paths whose dense start / end tables fit the exact pass' 32 KB of LDS
#define GTX_EXACT_PASS_KERNEL(NAME, NS, ATTR)                                                                                      \
  __global__ __launch_bounds__(64) ATTR void NAME(GraphView g, IndexView ix, uint8_t const * __restrict__ seq, uint32_t seq_stride,\
                                             gtx_read_meta const * __restrict__ meta, uint32_t * __restrict__ records,             \
                                             uint32_t rec_words, uint32_t const * __restrict__ big_tasks, uint32_t big_task_cap,   \
                                             uint32_t * big_state, uint8_t * slab, unsigned long long part_bytes /* the slab's */,   \
                                             unsigned long long min_part_bytes, uint32_t fixed_parts, uint32_t cand_cap, uint32_t cap_v, \
                                             uint32_t * __restrict__ arena,                                                        \
                                             unsigned long long arena_words, unsigned long long * arena_cursor,                    \
                                             uint32_t * __restrict__ next_tasks, uint32_t next_cap, uint32_t * next_state, uint32_t lds_keys)\
  {                                                                                                                                \
    if (big_state[0] == 0) /* (nearly every batch: nothing reached this pass) */                                                   \
      return;                                                                                                                      \
    /* The slab is cut into as many parts as there are tasks, at most one per workgroup of the grid and none smaller than      */ \
    /* min_part_bytes: a handful of tasks get large parts, thousands -- every read over one long repeat -- get small ones and    */ \
    /* all of the grid (a task is milliseconds of dependent round trips: the tasks in flight are the pass' throughput).          */ \
    {                                                                                                                              \
      unsigned long long const q = big_state[0] < big_task_cap ? big_state[0] : big_task_cap;                                      \
      unsigned long long parts = (q < gridDim.x && !fixed_parts) ? q : gridDim.x; /* (fixed_parts: a test switch, GTX_EXACT_PARTS) */ \
      if (min_part_bytes && parts * min_part_bytes > part_bytes)                                                                   \
        parts = part_bytes / min_part_bytes ? part_bytes / min_part_bytes : 1ull;                                                  \
      if (blockIdx.x >= parts)                                                                                                     \
        return;                                                                                                                    \
      part_bytes = (part_bytes / parts) & ~255ull;                                                                                 \
    }                                                                                                                              \
    NS::AlignWorkspace & ws = *reinterpret_cast<NS::AlignWorkspace *>(slab + static_cast<unsigned long long>(blockIdx.x) * part_bytes); \
    if (!NS::exact_setup<WaveHipMem>(&ws, part_bytes, cand_cap, cap_v))                                                            \
    {                                                                                                                              \
      /* a part that cannot hold one path: everything goes on to the launch that has the whole slab (or keeps its status) */      \
      if (blockIdx.x == 0 && (threadIdx.x & 63u) == 0 && next_tasks)                                                               \
      {                                                                                                                            \
        uint32_t const q = big_state[0] < big_task_cap ? big_state[0] : big_task_cap;                                              \
        for (uint32_t t = 0; t < q; ++t)                                                                                           \
        {                                                                                                                          \
          uint32_t const slot = atomicAdd(next_state, 1u);                                                                         \
          if (slot < next_cap)                                                                                                     \
            next_tasks[slot] = big_tasks[t];                                                                                       \
          else                                                                                                                     \
            atomicAdd(next_state + 3, 1u);                                                                                         \
        }                                                                                                                          \
      }                                                                                                                            \
      return;                                                                                                                      \
    }                                                                                                                              \
    /* (the dense start / end tables in LDS when the part's paths fit there, else where exact_setup put them in the slab) */       \
    extern __shared__ uint32_t s_pp_keys[]; /* 2 x lds_keys words: the launch sizes them */                                        \
    if (ws.cap_p <= lds_keys)                                                                                                      \
    {                                                                                                                              \
      ws.pp_start = s_pp_keys;                                                                                                     \
      ws.pp_end = s_pp_keys + lds_keys;                                                                                            \
    }                                                                                                                              \
    WaveHipMem::mem_sync();                                                                                                        \
    GTX_HBM_PASS_BODY(NS)                                                                                                          \
  }

#ifdef GTX_PROF
#define GTX_HBM_PASS_PROF_INIT                                                                                                     \
  if (threadIdx.x < 16)                                                                                                            \
    ws.prof_acc[threadIdx.x] = 0; /* (second-pass cycles are not added to the report) */                                           \
  WaveHipMem::mem_sync();
#else
#define GTX_HBM_PASS_PROF_INIT
#endif
GTX_HBM_PASS_KERNEL(gtx_align_big_kernel, big)
GTX_HBM_PASS_KERNEL(gtx_align_wide_kernel, wide)
GTX_EXACT_PASS_KERNEL(gtx_align_exact_kernel, exact, GTX_EXACT_PASS_ATTR)
GTX_EXACT_PASS_KERNEL(gtx_align_exact_wide_kernel, exactw, GTX_EXACT_PASS_ATTR)
// ... and a build of it that can be PLACED beside a full chip -- registers for four wavefronts per SIMD (128, what the HBM-table
// pass takes; the pass' own 186 need three wavefronts of the position-hinted pass to retire on one SIMD and nobody to take their
// room), launched with 2 KB of LDS instead of 32 (a CU that holds fourteen workgroups of that pass has 20 KB left) -- for the batch
// whose predecessor sent nothing this way: its four workgroups find the empty queue in 5 us instead of waiting 100-200 us for the
// position-hinted pass of the next batch to end (round 6, kernel trace of the staggered schedule).  A task that does come is done, slowly.
GTX_EXACT_PASS_KERNEL(gtx_align_exact_light_kernel, exact, __attribute__((amdgpu_waves_per_eu(4, 4))))
GTX_EXACT_PASS_KERNEL(gtx_align_exact_wide_light_kernel, exactw, __attribute__((amdgpu_waves_per_eu(4, 4))))


uint64_t big_workspace_bytes() { return sizeof(big::AlignWorkspace); }
uint64_t wide_workspace_bytes() { return sizeof(wide::AlignWorkspace); }

char const * launch_hbm_passes(HbmPassArgs const & a, hipStream_t stream)
{
  unsigned long long const arena_words = a.arena_words;
  bool const wide_pass = a.wide_tasks != nullptr;
  hipLaunchKernelGGL(gtx_align_big_kernel, dim3(a.big_blocks), dim3(64), 0, stream, a.g, a.ix, a.seq, a.seq_stride, a.meta, a.records, a.rec_words,
                     a.big_tasks, a.big_task_cap, a.big_state, static_cast<big::AlignWorkspace *>(a.big_ws), a.arena, arena_words, a.arena_cursor,
                     wide_pass ? a.wide_tasks : a.exact_tasks, wide_pass ? CallScratch::WIDE_TASK_CAP : CallScratch::EXACT_TASK_CAP,
                     wide_pass ? a.wide_state : a.exact_state);
  if (hipGetLastError() != hipSuccess)
    return "gtx_align_big_kernel launch";
  if (wide_pass) // graphs with a site of more than 64 alleles: the tasks that met an allele number >= 64
  {
    hipLaunchKernelGGL(gtx_align_wide_kernel, dim3(CallScratch::WIDE_BLOCKS), dim3(64), 0, stream, a.g, a.ix, a.seq, a.seq_stride, a.meta, a.records,
                       a.rec_words, a.wide_tasks, CallScratch::WIDE_TASK_CAP, a.wide_state, static_cast<wide::AlignWorkspace *>(a.wide_ws), a.arena,
                       arena_words, a.arena_cursor, a.exact_tasks, CallScratch::EXACT_TASK_CAP, a.exact_state);
    if (hipGetLastError() != hipSuccess)
      return "gtx_align_wide_kernel launch";
  }
  return nullptr;
}

char const * launch_exact_passes(HbmPassArgs const & a, hipStream_t stream)
{
  unsigned long long const arena_words = a.arena_words;
  // the exact pass: what exceeded the tables above, with a small part of the slab per task (up to exact_parts of them side by
  // side), then what did not fit with a large part (up to EXACT_LARGE_PARTS), then -- one workgroup -- with all of it.  Nearly
  // always all three find an empty queue and leave at once.
  // (a small part's paths have room for EXACT_PART_SITES variant sites and its walks for exact_part_cand_cap candidates, a large
  //  part's for EXACT_LARGE_SITES, the whole slab for the proven bounds: a site per read base, 128 live sequences times the alleles
  //  of the graph's widest site)
  uint32_t * const q1 = a.exact_tasks, * const q2 = q1 + CallScratch::EXACT_TASK_CAP, * const q3 = q2 + CallScratch::EXACT_TASK_CAP;
  uint32_t * const st1 = a.exact_state, * const st2 = st1 + 8, * const st3 = st2 + 8, * const st4 = st3 + 8;
  // (nothing expected -- the batch before sent no task here: the build that can be placed beside a full chip, four workgroups a launch)
  bool const light = a.exact_grid_limit != 0 && a.exact_grid_limit <= 4u;
  auto kernel = light ? (a.wide_sites ? gtx_align_exact_wide_light_kernel : gtx_align_exact_light_kernel)
                      : (a.wide_sites ? gtx_align_exact_wide_kernel : gtx_align_exact_kernel);
  uint32_t const lds_keys = light ? 256u : EXACT_LDS_KEYS;
  size_t const lds_bytes = 2u * lds_keys * sizeof(uint32_t);
  unsigned long long const slab_bytes = a.exact_slab_bytes, min_part = slab_bytes / a.exact_parts;
  // (the second launch: parts of at least 16 MB -- 32 of them in a 512 MB slab, 128 in the 2 GB slab of a large batch: on a reference
  //  with repeats two hundred tasks come this far, and 32 parts took them in six rounds of a millisecond each)
  static uint32_t const large_parts_env = []
  {
    char const * e = std::getenv("GTX_EXACT_LARGE_PARTS"); // (A/B switch)
    return e && std::atoi(e) > 0 ? static_cast<uint32_t>(std::min(std::atoi(e), 1024)) : 0u;
  }();
  // (graphs with sites of more than 64 alleles keep 32: a walk's candidate table alone can be tens of megabytes there)
  uint32_t const large_parts = a.exact_fixed_parts ? 1u
                               : large_parts_env ? large_parts_env
                               : a.wide_sites    ? CallScratch::EXACT_LARGE_PARTS
                                                 : static_cast<uint32_t>(std::min<unsigned long long>(256ull, std::max<unsigned long long>(CallScratch::EXACT_LARGE_PARTS, slab_bytes >> 24)));
  uint32_t const grid1 = a.exact_grid_limit ? std::min<uint32_t>(a.exact_parts, a.exact_grid_limit) : a.exact_parts;
  uint32_t const grid2 = a.exact_grid_limit ? std::min<uint32_t>(large_parts, a.exact_grid_limit) : large_parts;
  hipLaunchKernelGGL(kernel, dim3(grid1), dim3(64), lds_bytes, stream, a.g, a.ix, a.seq, a.seq_stride, a.meta, a.records, a.rec_words, q1,
                     CallScratch::EXACT_TASK_CAP, st1, a.exact_slab, slab_bytes, min_part, a.exact_fixed_parts ? 1u : 0u, a.exact_part_cand_cap,
                     CallScratch::EXACT_PART_SITES, a.arena, arena_words, a.arena_cursor, q2, CallScratch::EXACT_TASK_CAP, st2, lds_keys);
  hipLaunchKernelGGL(kernel, dim3(grid2), dim3(64), lds_bytes, stream, a.g, a.ix, a.seq, a.seq_stride, a.meta, a.records, a.rec_words, q2,
                     CallScratch::EXACT_TASK_CAP, st2, a.exact_slab, slab_bytes, slab_bytes / large_parts, 0u, a.exact_cand_cap,
                     CallScratch::EXACT_LARGE_SITES, a.arena, arena_words, a.arena_cursor, q3, CallScratch::EXACT_TASK_CAP, st3, lds_keys);
  // (what even the whole slab cannot hold is counted in st4: a queue of capacity 0)
  hipLaunchKernelGGL(kernel, dim3(1), dim3(64), lds_bytes, stream, a.g, a.ix, a.seq, a.seq_stride, a.meta, a.records, a.rec_words, q3, CallScratch::EXACT_TASK_CAP,
                     st3, a.exact_slab, slab_bytes, 0ull, 0u, a.exact_cand_cap, exact::AlignCfg::MAXV, a.arena, arena_words, a.arena_cursor, q3, 0u, st4, lds_keys);
  if (hipGetLastError() != hipSuccess)
    return "gtx_align_exact_kernel launch";
  return nullptr;
}
} // namespace gtx
