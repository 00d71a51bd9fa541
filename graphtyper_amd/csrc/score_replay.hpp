// score_replay.hpp -- host side of gtx_scores_replay: the sequential part of Haplotype::explain_to_score.
//
// Every effect of a read on the accumulators is an integer addition, except one: explain_to_score stops adding to a
// (haplotype, sample) cell once its max_log_score has come within epsilon of 0xFFFF (src/graph/haplotype.cpp:560; depth of
// roughly 8 000 reads on one site in one sample).  Which reads are dropped then depends on the order of the calls, and a
// read with a small epsilon may still be accepted after one with a larger epsilon was refused.  The scoring kernels add
// without the guard; for the cells whose sum reached the guard the calls are logged in a second pass over the items
// (ScoreAcc::replay_*), sorted back into call order here and replayed one by one exactly as the reference does.
#pragma once
#include "gtx_flat.hpp"
#include "score_core.hpp"

#include <algorithm>
#include <vector>

namespace gtx
{
constexpr uint32_t GTX_CELL_REPLAYED = 0x80000000u; // flag in hap_u32[cell][0]: the cell holds its replayed (exact) value
constexpr uint32_t SATURATION_GUARD = 0xFFFFu - 8u; // below this sum no call was ever refused (epsilon <= 8)

struct ReplayedCell
{
  uint32_t cell;
  uint32_t max_log_score;
  std::vector<uint32_t> log_score; // the cell's genotype triangle
};

// log: the entries of the marked cells in any order.  n_hap, ref_nvar: the graph's sites.
inline std::vector<ReplayedCell> replay_cells(HostGraph const & g, std::vector<ReplayEntry> & log)
{
  std::sort(log.begin(), log.end(), [](ReplayEntry const & a, ReplayEntry const & b) {
    if (a.cell != b.cell)
      return a.cell < b.cell;
    if (a.item != b.item)
      return a.item < b.item;
    return (a.order_eps >> 8) < (b.order_eps >> 8);
  });
  std::vector<ReplayedCell> out;
  for (size_t i = 0; i < log.size();)
  {
    uint32_t const cell = log[i].cell, h = cell % g.n_hap, cnum = g.ref_nvar[h];
    ReplayedCell rc;
    rc.cell = cell;
    rc.max_log_score = 0;
    rc.log_score.assign(static_cast<size_t>(cnum) * (cnum + 1) / 2, 0u);
    for (; i < log.size() && log[i].cell == cell; ++i)
    {
      uint32_t const eps = log[i].order_eps & 0xFFu;
      uint64_t const explains = (static_cast<uint64_t>(log[i].mask_hi) << 32) | log[i].mask_lo;
      if (!(rc.max_log_score < 0xFFFFu - eps)) // haplotype.cpp:560
        continue;
      rc.max_log_score += eps;
      size_t idx = 0;
      for (uint32_t y = 0; y < cnum; ++y)
      {
        bool const ey = (explains >> y) & 1ull;
        for (uint32_t x = 0; x <= y; ++x, ++idx)
        {
          bool const ex = (explains >> x) & 1ull;
          if (ex && ey)
            rc.log_score[idx] += eps;
          else if (ex || ey)
            rc.log_score[idx] += eps - 1;
        }
      }
    }
    out.push_back(std::move(rc));
  }
  return out;
}
} // namespace gtx
