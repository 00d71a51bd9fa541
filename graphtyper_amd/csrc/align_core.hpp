// align_core.hpp -- per-read work of the alignment kernel: one wavefront per (read, orientation).
//
// What the reference does per read (src/typer/alignment.cpp:23-103, find_genotype_paths_of_one_of_the_sequences) with
// heap containers, restated over fixed tables in LDS.
//
// Execution style ("wave-uniform + lane lambdas"): all 64 lanes run the same control flow on the same values (state
// lives in LDS, scalars are replicated), LDS writes are done by the leader lane only, and the data-parallel pieces --
// read unpacking, 2-bit key assembly, the 97 index probes per k-mer, stable hit compaction, character comparison of a
// read against graph sequence, table copies -- are expressed as lambdas over the lane index plus wave primitives
// (ballot, exclusive scan).  A policy type W supplies those primitives: WaveHip (gtx_api.hip) maps them to the
// hardware; tests/emu supplies a sequential stand-in so the very same source can be debugged without a GPU.
#pragma once
#include <type_traits>

#include "graph_dev.hpp"

namespace gtx
{
// main pass: tables sized for LDS (a read that exceeds one gets a status bit and goes to the second pass)
struct AlignCfg
{
  // Sized so that a workspace is at most 8 KB: 20 single-wave workgroups per CU (160 KB LDS).  The kernel is bound by
  // the latency of its dependent memory round trips, so resident waves are throughput; what does not fit is exact work
  // for the second pass, not a loss.
  static constexpr uint32_t MAX_READ = 256;  // bases
  static constexpr uint32_t MAX_KMERS = 8;   // get_num_kmers(MAX_READ)
  static constexpr uint32_t LBL_CAP = 40;    // labels of one k-mer list (multi-key lists are cut at 75 by the reference)
  static constexpr uint32_t MAXP = 16;       // live paths
  static constexpr uint32_t MAXPP = 12;      // paths made from one label list
  static constexpr uint32_t MAXV = 8;        // variant sites per path
  static constexpr uint32_t CAND_CAP = 24;   // sequences alive in one graph walk
  static constexpr uint32_t MAXIDS = 6;      // variant nodes on one walked sequence
  static constexpr uint32_t LOC_CAP = 16;    // graph locations of one path end
  static constexpr uint32_t WL_CAP = 32;     // labels kept by walk_read_ends/starts
  static constexpr uint32_t WLISTS = 8;      // label lists kept by walk_read_ends/starts
  static constexpr uint32_t KEY_CAP = 192;   // keys of a multi-key list (to_uint64_vec can return up to 4*97)
  static constexpr uint32_t KC = 5;          // k-mers whose index lookups are issued together up front (reads <= 187 bp)
  static constexpr uint32_t HE_CAP = 4;      // half-key bucket entries fetched up front per (k-mer, side)
  static constexpr uint32_t XL_CAP = 2;      // exact labels fetched up front per k-mer
  static constexpr uint32_t MW = 2;          // 32-bit words of an allele set: alleles 0..63 (a label beyond is an overflow)
  static constexpr bool DYN = false;         // tables are arrays of the sizes above (true: pointers into a slab of HBM, sizes at run time)
};

#include "align_core.inl"
#include "express4.inl"
} // namespace gtx
#include "hinted.hpp"
#include "hinted_long.hpp"
namespace gtx
{

// second pass: the same code over large tables that live in HBM (one workspace per workgroup)
namespace big
{
struct AlignCfg
{
  static constexpr uint32_t MAX_READ = 256;
  static constexpr uint32_t MAX_KMERS = 8;
  static constexpr uint32_t LBL_CAP = 2048;  // a single exact key may have thousands of occurrences in repeats
  static constexpr uint32_t MAXP = 512;      // (the reference stops walking above 256 seeds)
  static constexpr uint32_t MAXPP = 512;
  static constexpr uint32_t MAXV = 40;
  static constexpr uint32_t CAND_CAP = 512;  // the reference stops branching at 128 but one round may overshoot
  static constexpr uint32_t MAXIDS = 40;
  static constexpr uint32_t LOC_CAP = 256;   // MAX_NUM_LOCATIONS_PER_PATH
  static constexpr uint32_t WL_CAP = 2048;
  static constexpr uint32_t WLISTS = 256;
  static constexpr uint32_t KEY_CAP = 388;
  static constexpr uint32_t KC = 5;
  static constexpr uint32_t HE_CAP = 4;
  static constexpr uint32_t XL_CAP = 4;
  static constexpr uint32_t MW = 2;
  static constexpr bool DYN = false;
};
#include "align_core.inl"
} // namespace big

// a further HBM-table pass for graphs that have a site with more than 64 alleles (merged clusters: up to
// MAX_NUMBER_OF_HAPLOTYPES = 2560, include/graphtyper/constants.hpp.in:23): allele sets of GTX_WIDE_MASK_WORDS words.  The
// passes in front hand on every task that meets an allele number >= 64 (GTX_ST_WIDE_ALLELE).
namespace wide
{
struct AlignCfg // (a path with its allele sets is 5 KB here: fewer of them than in big::, what exceeds them keeps its status)
{
  static constexpr uint32_t MAX_READ = 256;
  static constexpr uint32_t MAX_KMERS = 8;
  static constexpr uint32_t LBL_CAP = 2048;
  static constexpr uint32_t MAXP = 128;
  static constexpr uint32_t MAXPP = 128;
  static constexpr uint32_t MAXV = 16;
  static constexpr uint32_t CAND_CAP = 8192; // one round of a walk over a site branches into every allele within the mismatch budget
  static constexpr uint32_t MAXIDS = 24;
  static constexpr uint32_t LOC_CAP = 256;
  static constexpr uint32_t WL_CAP = 2048;
  static constexpr uint32_t WLISTS = 256;
  static constexpr uint32_t KEY_CAP = 388;
  static constexpr uint32_t KC = 5;
  static constexpr uint32_t HE_CAP = 4;
  static constexpr uint32_t XL_CAP = 4;
  static constexpr uint32_t MW = GTX_WIDE_MASK_WORDS;
  static constexpr bool DYN = false;
};
#include "align_core.inl"
} // namespace wide

// The last pass: EXACT.  The reference keeps its paths, labels and walk candidates in heap containers and has no limit on
// any of them (genotype_paths.cpp:294-352: a read inside a 280-bp homopolymer chains 249 x 249 labels); every pass above
// refuses what exceeds its tables and hands the task on, and this one must not refuse.  Its tables are pointers into a
// slab of HBM that belongs to the workgroup for the duration of a task, cut to sizes computed at run time from the slab's
// size (exact_layout, below): a first launch gives each of a few workgroups a part of the scratch's slab, a second launch
// gives ONE workgroup the whole slab for what still did not fit -- so the only refusal left is a task that needs more
// than the slab the caller configured (gtx_params::exact_pass_mb / GTX_EXACT_PASS_MB; 288 GB of HBM are there to be used).
// The sizes that stay constants are proven bounds, not budgets: a path crosses at most one variant site per read base
// (every allele holds at least one base), a walked sequence one variant node per base, walks are not done for more than
// MAX_SEED_NUMBER_FOR_WALKING paths, a multi-key list is cut at max_index_labels.
namespace exact
{
struct AlignCfg
{
  static constexpr uint32_t MAX_READ = 256;
  static constexpr uint32_t MAX_KMERS = 8;
  static constexpr uint32_t LBL_CAP = 1, MAXP = 1, MAXPP = 1, CAND_CAP = 1, WL_CAP = 1; // (run-time sizes: DynTables)
  static constexpr uint32_t MAXV = MAX_READ;    // variant sites of a path
  static constexpr uint32_t MAXIDS = MAX_READ;  // variant nodes of a walked sequence
  static constexpr uint32_t LOC_CAP = 256;      // MAX_NUM_LOCATIONS_PER_PATH (beyond it the reference skips the path)
  static constexpr uint32_t WLISTS = 256;       // MAX_SEED_NUMBER_FOR_WALKING
  static constexpr uint32_t KEY_CAP = 388;      // 4 * 97 (type_conversions.cpp:207-266)
  static constexpr uint32_t KC = 5;
  static constexpr uint32_t HE_CAP = 4;
  static constexpr uint32_t XL_CAP = 4;
  static constexpr uint32_t MW = 2;
  static constexpr bool DYN = true;
};
#include "align_core.inl"
} // namespace exact

namespace exactw // ... for graphs that have a site of more than 64 alleles
{
struct AlignCfg
{
  static constexpr uint32_t MAX_READ = 256;
  static constexpr uint32_t MAX_KMERS = 8;
  static constexpr uint32_t LBL_CAP = 1, MAXP = 1, MAXPP = 1, CAND_CAP = 1, WL_CAP = 1;
  static constexpr uint32_t MAXV = MAX_READ;
  static constexpr uint32_t MAXIDS = MAX_READ;
  static constexpr uint32_t LOC_CAP = 256;
  static constexpr uint32_t WLISTS = 256;
  static constexpr uint32_t KEY_CAP = 388;
  static constexpr uint32_t KC = 5;
  static constexpr uint32_t HE_CAP = 4;
  static constexpr uint32_t XL_CAP = 4;
  static constexpr uint32_t MW = GTX_WIDE_MASK_WORDS;
  static constexpr bool DYN = true;
};
#include "align_core.inl"
} // namespace exactw

} // namespace gtx
