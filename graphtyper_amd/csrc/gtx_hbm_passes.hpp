// gtx_hbm_passes.hpp -- the alignment passes behind the general one (gtx_hbm_passes.hip): tables in HBM (512 paths), wide allele
// sets (graphs with a site of more than 64 alleles), and the exact pass whose tables are cut out of a slab at run time.
// A translation unit of their own: five instantiations of align_core.inl compile side by side with gtx_api.hip's kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "graph_dev.hpp"

namespace gtx
{
struct HbmPassArgs
{
  GraphView g;
  IndexView ix;
  uint8_t const * seq;
  uint32_t seq_stride;
  gtx_read_meta const * meta;
  uint32_t * records;
  uint32_t rec_words;
  // HBM-table pass
  uint32_t * big_tasks;
  uint32_t big_task_cap;
  uint32_t * big_state;
  uint32_t big_blocks;
  void * big_ws; // big_blocks x big::AlignWorkspace
  // wide-site pass (null: the graph has no site of more than 64 alleles)
  uint32_t * wide_tasks;
  uint32_t * wide_state;
  void * wide_ws; // CallScratch::WIDE_BLOCKS x wide::AlignWorkspace
  // exact pass
  uint32_t * exact_tasks; // three queues of CallScratch::EXACT_TASK_CAP
  uint32_t * exact_state; // 4 x 8 words
  uint8_t * exact_slab;
  uint64_t exact_slab_bytes;
  uint32_t exact_cand_cap;      // walk candidates of a task that has the whole slab (the proven bound)
  uint32_t exact_part_cand_cap; // ... of a task that has a part of it
  uint32_t exact_parts;         // workgroups of the first launch
  bool exact_fixed_parts;       // (test switch) always that many parts, however few tasks there are
  uint32_t exact_grid_limit = 0; // the most workgroups of an exact launch (0: no limit) -- what the batch before sent this way, and a few
  bool wide_sites;
  // big-record arena
  uint32_t * arena;
  uint64_t arena_words;
  unsigned long long * arena_cursor;
};

// Launches the passes on `stream`; returns null, or the name of the launch that failed.
// the HBM-table pass (and the wide-site pass of a graph that has one), then -- launch_exact_passes, with a.exact_slab chosen by
// the caller -- the three launches of the exact pass; both return NULL or the name of the launch that failed
char const * launch_hbm_passes(HbmPassArgs const & a, hipStream_t stream);
char const * launch_exact_passes(HbmPassArgs const & a, hipStream_t stream);

// bytes of one workspace of the HBM-table / wide-site pass
uint64_t big_workspace_bytes();
uint64_t wide_workspace_bytes();
} // namespace gtx
