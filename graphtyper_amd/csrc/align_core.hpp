// align_core.hpp -- per-read work of the alignment kernel: one wavefront per (read, orientation).
//
// What the reference does per read (src/typer/alignment.cpp:23-103, find_genotype_paths_of_one_of_the_sequences) with
// heap containers, restated over fixed tables in LDS:
//   * k-mer extraction + index probes are wave-parallel (97 keys per k-mer spread over the 64 lanes, stable
//     prefix-sum compaction of the hits so label order equals the reference's key order / bucket order);
//   * seed chaining, graph walks and the path filters are short, data dependent and branchy: lane 0 runs them on
//     the LDS tables while the other lanes wait (round 1; see DESIGN.md for what moves to all lanes next).
//
// The code is written against a `W` (wave) policy so that tests/emu can run the very same source on host threads
// (64 threads + barriers standing in for one wavefront).  The product instantiates it with WaveHip only.
#pragma once
#include <cstdint>

#include "gtx_flat.hpp"

#if defined(__HIPCC__)
#define GTX_DEV __device__ inline
#else
#define GTX_DEV inline
#endif

namespace gtx
{
// include/graphtyper/constants.hpp.in:43-48
constexpr uint32_t MAX_UNIQUE_KMER_POSITIONS = 512;
constexpr uint32_t MAX_SEED_NUMBER_ALLOWING_MISMATCHES = 64;
constexpr uint32_t MAX_SEED_NUMBER_FOR_WALKING = 256;
constexpr uint32_t MAX_NUM_LOCATIONS_PER_PATH = 256;

struct AlignCfg
{
  static constexpr uint32_t MAX_READ = 256;  // bases
  static constexpr uint32_t MAX_KMERS = 8;   // get_num_kmers(MAX_READ)
  static constexpr uint32_t LBL_CAP = 128;   // labels of one k-mer list
  static constexpr uint32_t MAXP = 32;       // live paths
  static constexpr uint32_t MAXPP = 32;      // paths made from one label list
  static constexpr uint32_t MAXV = 8;        // variant sites per path
  static constexpr uint32_t CAND_CAP = 32;   // sequences alive in one graph walk
  static constexpr uint32_t MAXIDS = 8;      // variant nodes on one walked sequence
  static constexpr uint32_t LOC_CAP = 32;    // graph locations of one path end
  static constexpr uint32_t WL_CAP = 64;     // labels kept by walk_read_ends/starts
  static constexpr uint32_t WLISTS = 16;     // label lists kept by walk_read_ends/starts
  static constexpr uint32_t KEY_CAP = 388;   // to_uint64_vec can return up to 4*97 keys
};

struct PVar
{
  uint32_t site;
  uint32_t mlo, mhi; // allele set (Path::nums[i]) as a 64-bit mask
};

struct DPath // gyper::Path (include/graphtyper/typer/path.hpp:18-79)
{
  uint32_t start, end;
  uint16_t rs, re; // read_start_index, read_end_index
  uint16_t mism, nvar;
  PVar v[AlignCfg::MAXV];
};

struct Loc // gyper::Location (include/graphtyper/graph/location.hpp)
{
  uint32_t type; // 0 = 'U', 1 = 'R', 2 = 'V'
  uint32_t node, order, offset;
};

struct Cand // one element of var_and_refs / var_ids / end_pos in Graph::get_labels_forward (graph.cpp:1192-1196)
{
  uint32_t len;  // var_and_refs[j].size()
  uint32_t mism; // mismatches of its first min(len, L) characters against the sub-read; max+1 once it is dead
  uint32_t pos;  // end_pos[j] (forward) or start_pos[j] (backward)
  uint32_t nids;
  uint32_t ids[AlignCfg::MAXIDS];
};

struct AlignWorkspace // lives in LDS, one per wavefront
{
  uint8_t rd[AlignCfg::MAX_READ]; // read as 4-bit IUPAC codes, orientation applied
  DevLabel lbl[AlignCfg::LBL_CAP];
  DPath paths[AlignCfg::MAXP];
  DPath pp[AlignCfg::MAXPP];
  union
  {
    uint64_t keybuf[AlignCfg::KEY_CAP];
    Cand cand[AlignCfg::CAND_CAP];
  } u;
  Loc locs[AlignCfg::LOC_CAP];
  DevLabel wl[AlignCfg::WL_CAP];
  uint32_t wl_off[AlignCfg::WLISTS + 1];
  uint32_t wl_idx[AlignCfg::WLISTS];
  DevLabel dfs_out[AlignCfg::WL_CAP]; // labels of the current iterative_dfs call
  // per k-mer exact-probe results
  uint64_t key0[AlignCfg::MAX_KMERS];
  uint32_t nkeys0[AlignCfg::MAX_KMERS];
  uint32_t off0[AlignCfg::MAX_KMERS];
  uint32_t cnt0[AlignCfg::MAX_KMERS];
  uint32_t n_paths, longest, status, n_lbl, n_keys, read_len, n_wl, n_wlists, n_dfs, scratch;
};

GTX_DEV uint64_t pv_mask(PVar const & v)
{
  return (static_cast<uint64_t>(v.mhi) << 32) | v.mlo;
}

GTX_DEV void pv_set(PVar & v, uint64_t m)
{
  v.mlo = static_cast<uint32_t>(m);
  v.mhi = static_cast<uint32_t>(m >> 32);
}

GTX_DEV uint32_t path_size(DPath const & p)
{
  return static_cast<uint32_t>(p.re) - static_cast<uint32_t>(p.rs) + 1u;
}

GTX_DEV char code_to_char(uint32_t c)
{
  // seq_nt16_str (htslib) assigned to a seqan Iupac: '=' is not an IUPAC letter and becomes N
  // (src/utilities/hts_parallel_reader.cpp:226-243)
  return "NACMGRSVTWYHKDBN"[c & 15u];
}

// ---------------------------------------------------------------------------------------------------------------
// graph helpers
// ---------------------------------------------------------------------------------------------------------------
GTX_DEV bool g_is_special(GraphView const & g, uint32_t pos)
{
  return pos >= SPECIAL_START && (pos - SPECIAL_START) < g.n_special; // graph.cpp:1784-1787
}

GTX_DEV uint32_t g_ref_reach_pos(GraphView const & g, uint32_t pos)
{
  return g_is_special(g, pos) ? g.special_ref_reach[pos - SPECIAL_START] : pos; // graph.cpp:1789-1795
}

GTX_DEV uint32_t g_actual_pos(GraphView const & g, uint32_t pos)
{
  return g_is_special(g, pos) ? g.special_actual[pos - SPECIAL_START] : pos; // graph.cpp:1797-1803
}

// Graph::get_special_pos for a position inside an allele of `site` (graph.cpp:1775-1782); identity up to the
// reference allele's reach.
GTX_DEV uint32_t g_special_of(GraphView const & g, uint32_t site, uint32_t pos)
{
  uint32_t const rr = g.site_ref_reach[site];
  return pos > rr ? SPECIAL_START + g.site_special_base[site] + (pos - rr - 1) : pos;
}

// last reference node whose order is <= pos (the `rr` of graph.cpp:950-955); pos >= first_order required
GTX_DEV uint32_t g_ref_node_at(GraphView const & g, uint32_t pos)
{
  uint32_t b = (pos - g.first_order) >> POS_BUCKET_SHIFT;
  if (b >= g.n_bucket)
    b = g.n_bucket - 1;
  uint32_t r = g.pos_bucket[b];
  while (r + 1 < g.n_ref && g.ref_order[r + 1] <= pos)
    ++r;
  return r;
}

// Graph::get_locations_of_a_position (graph.cpp:1154-1185 -> 931-1029).  The reference scans reference nodes
// backwards and looks every variant node up in path.var_order; here the (few) sites of the path are visited in
// descending order instead, which yields the same locations in the same order.
GTX_DEV uint32_t get_locations(GraphView const & g, uint32_t pos, DPath const & path, Loc * locs, uint32_t cap, uint32_t & status)
{
  bool const special = g_is_special(g, pos);
  if (special)
    pos = g.special_actual[pos - SPECIAL_START];
  uint32_t n = 0;
  if (pos < g.first_order)
    return 0;
  if (g.n_ref == 1)
  {
    locs[0] = Loc{1, 0, g.ref_order[0], pos - g.ref_order[0]};
    return 1;
  }
  int64_t rr = g_ref_node_at(g, pos);
  if (pos < g.ref_order[rr] + g.ref_len[rr])
  {
    if (!special)
    {
      locs[0] = Loc{1, static_cast<uint32_t>(rr), g.ref_order[rr], pos - g.ref_order[rr]};
      return 1;
    }
    --rr;
  }
  // sites rr' <= rr with reach(rr') + PADDING > pos, descending; only sites the path carries can contribute
  bool const path_empty = path.start == path.end;
  int64_t bound = rr + 1;
  for (;;)
  {
    // next site of the path below `bound` (largest first); a site listed twice counts once (std::find -> first)
    int64_t best = -1;
    uint32_t best_j = 0;
    for (uint32_t j = 0; j < path.nvar; ++j)
    {
      int64_t const s = path.v[j].site;
      if (s < bound && s > best)
      {
        best = s;
        best_j = j;
      }
      else if (s == best && j < best_j)
        best_j = j;
    }
    if (best < 0)
      break;
    bound = best;
    uint32_t const site = static_cast<uint32_t>(best);
    int64_t const reach = static_cast<int64_t>(g.ref_order[site]) + g.ref_len[site] - 1;
    if (!(reach + static_cast<int64_t>(g.padding) > static_cast<int64_t>(pos)))
      break; // the reference stops its backward scan here; lower sites reach even less far
    uint32_t const fv = g.ref_first_var[site], nv = g.ref_nvar[site];
    uint64_t const mask = pv_mask(path.v[best_j]);
    for (uint32_t i = 0; i < nv; ++i)
    {
      uint32_t const v = fv + i;
      if (pos >= g.var_order[v] && pos <= g.var_order[v] + g.var_len[v] - 1)
        if (path_empty || ((mask >> i) & 1ull))
        {
          if (n >= cap)
          {
            status |= GTX_ST_DFS_OVERFLOW;
            return n;
          }
          locs[n++] = Loc{2, v, g.var_order[v], pos - g.var_order[v]};
        }
    }
  }
  return n;
}

// ---------------------------------------------------------------------------------------------------------------
// graph walks: Graph::get_labels_forward / get_labels_backward (graph.cpp:1187-1439 / 1441-1701)
// Sequences are never materialised: a candidate keeps its length and the mismatches of its already compared prefix
// (count_mismatches restarts from 0 every time in the reference, which is the same sum).
// ---------------------------------------------------------------------------------------------------------------
struct SubRead
{
  uint8_t const * rd; // codes of the whole read
  uint32_t begin;     // first base of the sub-read
  uint32_t len;       // L
};

// compares graph characters dna[0..n) with sub-read characters starting at read offset `at` (forward), stops at L.
// Returns the mismatch count capped at max+1 (= dead); '<' or '>' in the compared range kills (graph_utils.hpp:7-37).
GTX_DEV uint32_t cmp_forward(SubRead const & sr, uint32_t at, char const * dna, uint32_t n, uint32_t mism, uint32_t maxmm)
{
  for (uint32_t i = 0; i < n && at + i < sr.len; ++i)
  {
    if (mism > maxmm)
      return maxmm + 1;
    char const gc = dna[i];
    if (gc == '>' || gc == '<')
      return maxmm + 1;
    char const rc = code_to_char(sr.rd[sr.begin + at + i]);
    if (gc != rc && rc != 'N' && gc != 'N')
      ++mism;
  }
  return mism > maxmm ? maxmm + 1 : mism;
}

// backward flavour: the candidate already covers the last `at` characters of the sub-read; dna[0..n) is prepended,
// i.e. dna[n-1] aligns with sub-read character L-1-at (graph_utils.hpp:39-69).
GTX_DEV uint32_t cmp_backward(SubRead const & sr, uint32_t at, char const * dna, uint32_t n, uint32_t mism, uint32_t maxmm)
{
  for (uint32_t i = 0; i < n && at + i < sr.len; ++i)
  {
    if (mism > maxmm)
      return maxmm + 1;
    char const gc = dna[n - 1 - i];
    if (gc == '>' || gc == '<')
      return maxmm + 1;
    char const rc = code_to_char(sr.rd[sr.begin + sr.len - 1 - at - i]);
    if (gc != rc && rc != 'N' && gc != 'N')
      ++mism;
  }
  return mism > maxmm ? maxmm + 1 : mism;
}

GTX_DEV void cand_erase(Cand * c, uint32_t & n, uint32_t j)
{
  for (uint32_t k = j; k + 1 < n; ++k)
    c[k] = c[k + 1];
  --n;
}

// appends the labels of one start location; returns false on table overflow
GTX_DEV bool labels_forward(GraphView const & g, Loc const & s, SubRead const & sr, uint32_t & max_mismatches, Cand * cand,
                            DevLabel * out, uint32_t & n_out, uint32_t out_cap, uint32_t & status)
{
  uint32_t const L = sr.len;
  uint32_t const maxmm = max_mismatches;
  uint32_t n = 1;
  Cand & c0 = cand[0];
  c0.nids = 0;
  uint32_t site = INVALID; // site whose alleles come next (`vars`), INVALID = none
  if (s.type == 2)
  {
    uint32_t const v = s.node;
    c0.ids[c0.nids++] = v;
    uint32_t const vlen = g.var_len[v] - s.offset;
    c0.mism = cmp_forward(sr, 0, g.dna + g.var_dna[v] + s.offset, vlen, 0, maxmm);
    c0.len = vlen;
    uint32_t const vsite = g.var_out_ref[v] - 1;
    if (c0.len >= L)
      c0.pos = g_special_of(g, vsite, (g.var_order[v] + g.var_len[v] - 1) - (c0.len - L));
    else
    {
      uint32_t const r = g.var_out_ref[v];
      c0.mism = cmp_forward(sr, c0.len, g.dna + g.ref_dna[r], g.ref_len[r], c0.mism, maxmm);
      c0.len += g.ref_len[r];
      c0.pos = (g.ref_order[r] + g.ref_len[r] - 1) - (c0.len - L);
      if (g.ref_nvar[r] > 0)
        site = r;
    }
  }
  else
  {
    uint32_t const r = s.node;
    uint32_t const rl = g.ref_len[r] - s.offset;
    c0.mism = cmp_forward(sr, 0, g.dna + g.ref_dna[r] + s.offset, rl, 0, maxmm);
    c0.len = rl;
    c0.pos = (g.ref_order[r] + g.ref_len[r] - 1) - (c0.len - L);
    if (g.ref_nvar[r] > 0)
      site = r;
  }

  if (site != INVALID && cand[0].len < L)
  {
    bool all_long = false;
    while (!all_long && n < 128 && site != INVALID)
    {
      all_long = true;
      uint32_t const r = site + 1; // reference node behind the site
      uint32_t const fv = g.ref_first_var[site], nv = g.ref_nvar[site];
      char const * rdna = g.dna + g.ref_dna[r];
      uint32_t const rlen = g.ref_len[r];
      uint32_t const rreach = g.ref_order[r] + rlen - 1;
      uint32_t original = n;
      for (uint32_t j = 0; j < original; ++j)
      {
        if (cand[j].len >= L)
          continue;
        for (uint32_t i = 0; i + 1 < nv; ++i)
        {
          uint32_t const v = fv + i;
          uint32_t len = cand[j].len;
          uint32_t mm = cmp_forward(sr, len, g.dna + g.var_dna[v], g.var_len[v], cand[j].mism, maxmm);
          len += g.var_len[v];
          bool const enough = len >= L;
          if (!enough)
          {
            mm = cmp_forward(sr, len, rdna, rlen, mm, maxmm);
            len += rlen;
          }
          if (mm <= maxmm)
          {
            if (n >= AlignCfg::CAND_CAP || cand[j].nids >= AlignCfg::MAXIDS)
            {
              status |= GTX_ST_DFS_OVERFLOW;
              return false;
            }
            Cand & nc = cand[n++];
            nc = cand[j];
            nc.ids[nc.nids++] = v;
            nc.len = len;
            nc.mism = mm;
            if (len < L)
              all_long = false;
            nc.pos = enough ? g_special_of(g, site, (g.var_order[v] + g.var_len[v] - 1) - (len - L)) : rreach - (len - L);
          }
        }
        uint32_t const v = fv + nv - 1;
        Cand & c = cand[j];
        c.mism = cmp_forward(sr, c.len, g.dna + g.var_dna[v], g.var_len[v], c.mism, maxmm);
        c.len += g.var_len[v];
        bool const enough = c.len >= L;
        if (!enough)
        {
          c.mism = cmp_forward(sr, c.len, rdna, rlen, c.mism, maxmm);
          c.len += rlen;
        }
        if (c.mism <= maxmm)
        {
          if (c.nids >= AlignCfg::MAXIDS)
          {
            status |= GTX_ST_DFS_OVERFLOW;
            return false;
          }
          c.ids[c.nids++] = v;
          if (c.len < L)
            all_long = false;
          c.pos = enough ? g_special_of(g, site, (g.var_order[v] + g.var_len[v] - 1) - (c.len - L)) : rreach - (c.len - L);
        }
        else
        {
          cand_erase(cand, n, j);
          --original;
          --j;
        }
      }
      if (all_long)
        break;
      site = g.ref_nvar[r] > 0 ? r : INVALID;
    }
  }

  // keep the sequences with the fewest mismatches (graph.cpp:1375-1402); `<` tightens the budget and restarts
  uint32_t first_out = n_out;
  for (uint32_t j = 0; j < n; ++j)
  {
    if (cand[j].len < L)
      continue;
    uint32_t const mm = cand[j].mism;
    if (mm > max_mismatches)
      continue;
    if (mm < max_mismatches)
    {
      max_mismatches = mm;
      n_out = first_out;
    }
    uint32_t start_pos = s.order + s.offset;
    if (s.type == 2)
      start_pos = g_special_of(g, g.var_out_ref[s.node] - 1, start_pos);
    uint32_t const nl = cand[j].nids == 0 ? 1 : cand[j].nids;
    if (n_out + nl > out_cap)
    {
      status |= GTX_ST_DFS_OVERFLOW;
      return false;
    }
    if (cand[j].nids == 0)
      out[n_out++] = DevLabel{start_pos, cand[j].pos, INVALID, 0};
    else
      for (uint32_t k = 0; k < cand[j].nids; ++k)
      {
        uint32_t const v = cand[j].ids[k];
        uint32_t const vs = g.var_out_ref[v] - 1;
        out[n_out++] = DevLabel{start_pos, cand[j].pos, vs, v - g.ref_first_var[vs]};
      }
  }
  return true;
}

GTX_DEV bool labels_backward(GraphView const & g, Loc const & e, SubRead const & sr, uint32_t & max_mismatches, Cand * cand,
                             DevLabel * out, uint32_t & n_out, uint32_t out_cap, uint32_t & status)
{
  uint32_t const L = sr.len;
  uint32_t const maxmm = max_mismatches;
  uint32_t n = 1;
  Cand & c0 = cand[0];
  c0.nids = 0;
  uint32_t site = INVALID; // site whose alleles are prepended next
  if (e.type == 2)
  {
    uint32_t const v = e.node;
    c0.ids[c0.nids++] = v;
    uint32_t const vlen = e.offset + 1;
    c0.mism = cmp_backward(sr, 0, g.dna + g.var_dna[v], vlen, 0, maxmm);
    c0.len = vlen;
    uint32_t const vsite = g.var_out_ref[v] - 1;
    if (c0.len >= L)
      c0.pos = g_special_of(g, vsite, g.var_order[v] + (c0.len - L));
    else
    {
      uint32_t const r = vsite;
      c0.mism = cmp_backward(sr, c0.len, g.dna + g.ref_dna[r], g.ref_len[r], c0.mism, maxmm);
      c0.len += g.ref_len[r];
      c0.pos = g.ref_order[r] + (c0.len - L);
      if (r != 0)
        site = r - 1;
    }
  }
  else
  {
    uint32_t const r = e.node;
    if (r != 0)
      site = r - 1;
    uint32_t const rl = e.offset + 1;
    c0.mism = cmp_backward(sr, 0, g.dna + g.ref_dna[r], rl, 0, maxmm);
    c0.len = rl;
    c0.pos = g.ref_order[r] + (c0.len - L);
  }

  if (site != INVALID && cand[0].len < L)
  {
    bool all_long = false;
    while (!all_long && n < 128 && site != INVALID)
    {
      all_long = true;
      uint32_t const r = site; // reference node in front of the site
      uint32_t const fv = g.ref_first_var[site], nv = g.ref_nvar[site];
      char const * rdna = g.dna + g.ref_dna[r];
      uint32_t const rlen = g.ref_len[r];
      uint32_t original = n;
      for (uint32_t j = 0; j < original; ++j)
      {
        if (cand[j].len >= L)
          continue;
        for (uint32_t i = 0; i + 1 < nv; ++i)
        {
          uint32_t const v = fv + i;
          uint32_t len = cand[j].len;
          uint32_t mm = cmp_backward(sr, len, g.dna + g.var_dna[v], g.var_len[v], cand[j].mism, maxmm);
          len += g.var_len[v];
          bool const enough = len >= L;
          if (!enough)
          {
            mm = cmp_backward(sr, len, rdna, rlen, mm, maxmm);
            len += rlen;
          }
          if (mm <= maxmm)
          {
            if (n >= AlignCfg::CAND_CAP || cand[j].nids >= AlignCfg::MAXIDS)
            {
              status |= GTX_ST_DFS_OVERFLOW;
              return false;
            }
            Cand & nc = cand[n++];
            nc = cand[j];
            nc.ids[nc.nids++] = v;
            nc.len = len;
            nc.mism = mm;
            if (len < L)
              all_long = false;
            nc.pos = enough ? g_special_of(g, site, g.var_order[v] + (len - L)) : g.ref_order[r] + (len - L);
          }
        }
        uint32_t const v = fv + nv - 1;
        Cand & c = cand[j];
        c.mism = cmp_backward(sr, c.len, g.dna + g.var_dna[v], g.var_len[v], c.mism, maxmm);
        c.len += g.var_len[v];
        bool const enough = c.len >= L;
        if (!enough)
        {
          c.mism = cmp_backward(sr, c.len, rdna, rlen, c.mism, maxmm);
          c.len += rlen;
        }
        if (c.mism <= maxmm)
        {
          if (c.nids >= AlignCfg::MAXIDS)
          {
            status |= GTX_ST_DFS_OVERFLOW;
            return false;
          }
          c.ids[c.nids++] = v;
          if (c.len < L)
            all_long = false;
          c.pos = enough ? g_special_of(g, site, g.var_order[v] + (c.len - L)) : g.ref_order[r] + (c.len - L);
        }
        else
        {
          cand_erase(cand, n, j);
          --original;
          --j;
        }
      }
      if (all_long)
        break;
      if (r == 0)
        break;
      site = r - 1;
    }
  }

  uint32_t first_out = n_out;
  for (uint32_t j = 0; j < n; ++j)
  {
    if (cand[j].len < L)
      continue;
    uint32_t const mm = cand[j].mism;
    if (mm > max_mismatches)
      continue;
    if (mm < max_mismatches)
    {
      max_mismatches = mm;
      n_out = first_out;
    }
    uint32_t end_pos = e.order + e.offset;
    if (e.type == 2)
      end_pos = g_special_of(g, g.var_out_ref[e.node] - 1, end_pos);
    uint32_t const nl = cand[j].nids == 0 ? 1 : cand[j].nids;
    if (n_out + nl > out_cap)
    {
      status |= GTX_ST_DFS_OVERFLOW;
      return false;
    }
    if (cand[j].nids == 0)
      out[n_out++] = DevLabel{cand[j].pos, end_pos, INVALID, 0};
    else
      for (uint32_t k = 0; k < cand[j].nids; ++k)
      {
        uint32_t const v = cand[j].ids[k];
        uint32_t const vs = g.var_out_ref[v] - 1;
        out[n_out++] = DevLabel{cand[j].pos, end_pos, vs, v - g.ref_first_var[vs]};
      }
  }
  return true;
}

// Graph::iterative_dfs (graph.cpp:1703-1754): labels of all locations that tie the fewest mismatches
GTX_DEV uint32_t iterative_dfs(GraphView const & g, AlignWorkspace & ws, uint32_t n_locs, bool backward, SubRead const & sr,
                               uint32_t & max_mismatches)
{
  uint32_t n_out = 0;
  if (n_locs > 1024)
    return 0;
  for (uint32_t k = 0; k < n_locs; ++k)
  {
    uint32_t mm = max_mismatches;
    uint32_t const before = n_out;
    uint32_t after = n_out;
    bool ok = backward ? labels_backward(g, ws.locs[k], sr, mm, ws.u.cand, ws.dfs_out, after, AlignCfg::WL_CAP, ws.status)
                       : labels_forward(g, ws.locs[k], sr, mm, ws.u.cand, ws.dfs_out, after, AlignCfg::WL_CAP, ws.status);
    if (!ok)
      return 0;
    if (after == before)
      continue; // no labels from this location
    if (mm < max_mismatches)
    {
      max_mismatches = mm;
      // labels = new_labels
      uint32_t const cnt = after - before;
      for (uint32_t i = 0; i < cnt; ++i)
        ws.dfs_out[i] = ws.dfs_out[before + i];
      n_out = cnt;
    }
    else if (mm == max_mismatches)
      n_out = after;
    // mm > max_mismatches cannot happen: the walk never returns labels above its budget
  }
  return n_out;
}

// ---------------------------------------------------------------------------------------------------------------
// seed chaining: GenotypePaths::add_next_kmer_labels / add_prev_kmer_labels (genotype_paths.cpp:294-352 / 233-292)
// ---------------------------------------------------------------------------------------------------------------

// find_all_nonduplicated_paths (genotype_paths.cpp:32-66) + Path::merge_with_current (path.cpp:105-129)
GTX_DEV uint32_t make_pp(AlignWorkspace & ws, DevLabel const * ll, uint32_t n, uint32_t rs, uint32_t re, uint32_t mism)
{
  uint32_t npp = 0;
  for (uint32_t i = 0; i < n; ++i)
  {
    DevLabel const l = ll[i];
    uint32_t d = 0;
    for (; d < npp; ++d)
      if (ws.pp[d].start == l.start && ws.pp[d].end == l.end)
        break;
    if (d == npp)
    {
      if (npp >= AlignCfg::MAXPP)
      {
        ws.status |= GTX_ST_PATH_OVERFLOW;
        return npp;
      }
      DPath & p = ws.pp[npp++];
      p.start = l.start;
      p.end = l.end;
      p.rs = static_cast<uint16_t>(rs);
      p.re = static_cast<uint16_t>(re);
      p.mism = static_cast<uint16_t>(mism);
      p.nvar = 0;
      if (l.site != INVALID)
      {
        p.v[0].site = l.site;
        pv_set(p.v[0], 1ull << l.allele);
        p.nvar = 1;
      }
      continue;
    }
    if (l.site == INVALID)
      continue;
    DPath & p = ws.pp[d];
    uint32_t k = 0;
    for (; k < p.nvar; ++k)
      if (p.v[k].site == l.site)
        break;
    if (k < p.nvar)
      pv_set(p.v[k], pv_mask(p.v[k]) | (1ull << l.allele));
    else
    {
      if (p.nvar >= AlignCfg::MAXV)
      {
        ws.status |= GTX_ST_PATH_OVERFLOW;
        return npp;
      }
      p.v[p.nvar].site = l.site;
      pv_set(p.v[p.nvar], 1ull << l.allele);
      ++p.nvar;
    }
  }
  return npp;
}

// Path::Path(p1, p2) (path.cpp:38-82): everything from p2, allele sets of shared sites intersected with p1's, p1's
// other sites appended, start/read_start_index taken from p1.  false <=> the reference returns early on an empty
// intersection (its caller then discards the half merged object).
GTX_DEV bool merge_paths(DPath const & p1, DPath const & p2, DPath & np, uint32_t & status)
{
  np = p2;
  for (uint32_t i = 0; i < p1.nvar; ++i)
  {
    uint32_t j = 0;
    for (; j < np.nvar; ++j)
      if (np.v[j].site == p1.v[i].site)
        break;
    if (j < np.nvar)
    {
      uint64_t const m = pv_mask(np.v[j]) & pv_mask(p1.v[i]);
      pv_set(np.v[j], m);
      if (m == 0)
        return false;
    }
    else
    {
      if (np.nvar >= AlignCfg::MAXV)
      {
        status |= GTX_ST_PATH_OVERFLOW;
        return false;
      }
      np.v[np.nvar++] = p1.v[i];
    }
  }
  np.rs = p1.rs;
  np.start = p1.start;
  np.mism = static_cast<uint16_t>(np.mism + p1.mism);
  return true;
}

GTX_DEV void push_path(AlignWorkspace & ws, DPath const & p)
{
  if (ws.n_paths >= AlignCfg::MAXP)
  {
    ws.status |= GTX_ST_PATH_OVERFLOW;
    return;
  }
  ws.paths[ws.n_paths++] = p;
}

GTX_DEV void add_kmer_labels(AlignWorkspace & ws, DevLabel const * ll, uint32_t n, uint32_t rs, uint32_t re, uint32_t mism,
                             bool prev)
{
  uint32_t const npp = make_pp(ws, ll, n, rs, re, mism);
  if (ws.status)
    return;
  uint32_t const original_size = ws.n_paths;
  uint64_t matched = 0;
  for (uint32_t i = 0; i < original_size; ++i)
  {
    if (prev ? (ws.paths[i].rs != re) : (ws.paths[i].re != rs))
      continue;
    bool once = false;
    DPath const original = ws.paths[i];
    for (uint32_t j = 0; j < npp; ++j)
    {
      DPath np;
      bool ok;
      if (prev)
      {
        if (!(ws.pp[j].end == original.start))
          continue;
        ok = merge_paths(ws.pp[j], original, np, ws.status);
      }
      else
      {
        if (!(original.end == ws.pp[j].start))
          continue;
        ok = merge_paths(original, ws.pp[j], np, ws.status);
      }
      if (ws.status)
        return;
      if (!ok)
        continue;
      matched |= 1ull << j;
      if (once)
        push_path(ws, np);
      else
      {
        uint32_t const sz = path_size(np);
        if (sz > ws.longest)
          ws.longest = sz;
        ws.paths[i] = np;
        once = true;
      }
    }
  }
  for (uint32_t j = 0; j < npp; ++j)
    if (!((matched >> j) & 1ull))
    {
      uint32_t const sz = path_size(ws.pp[j]);
      if (sz > ws.longest)
        ws.longest = sz;
      push_path(ws, ws.pp[j]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// path filters (genotype_paths.cpp)
// ---------------------------------------------------------------------------------------------------------------
GTX_DEV void remove_short_paths(AlignWorkspace & ws) // :824-834
{
  if (ws.longest <= 1)
    return;
  uint32_t k = 0;
  for (uint32_t i = 0; i < ws.n_paths; ++i)
    if (!(path_size(ws.paths[i]) < ws.longest))
    {
      if (k != i)
        ws.paths[k] = ws.paths[i];
      ++k;
    }
  ws.n_paths = k;
}

GTX_DEV void update_longest(AlignWorkspace & ws) // :858-864
{
  uint32_t m = 0;
  for (uint32_t i = 0; i < ws.n_paths; ++i)
  {
    uint32_t const s = path_size(ws.paths[i]);
    if (s > m)
      m = s;
  }
  ws.longest = m;
}

GTX_DEV void remove_paths_with_too_many_mismatches(AlignWorkspace & ws) // :360-380
{
  if (ws.n_paths == 0)
    return;
  uint32_t mn = 10;
  for (uint32_t i = 0; i < ws.n_paths; ++i)
    if (ws.paths[i].mism < mn)
      mn = ws.paths[i].mism;
  uint32_t k = 0;
  for (uint32_t i = 0; i < ws.n_paths; ++i)
    if (!(ws.paths[i].mism > mn))
    {
      if (k != i)
        ws.paths[k] = ws.paths[i];
      ++k;
    }
  ws.n_paths = k;
}

GTX_DEV bool all_paths_unique(GraphView const & g, DPath const * paths, uint32_t n) // :219-231
{
  for (uint32_t i = 1; i < n; ++i)
    if (g_ref_reach_pos(g, paths[0].start) != g_ref_reach_pos(g, paths[i].start) &&
        g_ref_reach_pos(g, paths[0].end) != g_ref_reach_pos(g, paths[i].end))
      return false;
  return true;
}

GTX_DEV bool path_is_reference(DPath const & p) // path.cpp:176-185
{
  for (uint32_t k = 0; k < p.nvar; ++k)
    if (!(pv_mask(p.v[k]) & 1ull))
      return false;
  return true;
}

GTX_DEV void remove_non_ref_paths_when_read_matches_ref(GraphView const & g, AlignWorkspace & ws) // :460-474
{
  if (all_paths_unique(g, ws.paths, ws.n_paths))
    return;
  bool any = false;
  for (uint32_t i = 0; i < ws.n_paths; ++i)
    if (path_is_reference(ws.paths[i]))
      any = true;
  if (!any)
    return;
  uint32_t k = 0;
  for (uint32_t i = 0; i < ws.n_paths; ++i)
    if (path_is_reference(ws.paths[i]))
    {
      if (k != i)
        ws.paths[k] = ws.paths[i];
      ++k;
    }
  ws.n_paths = k;
}

GTX_DEV void remove_fully_special_paths(GraphView const & g, AlignWorkspace & ws) // :476-481
{
  uint32_t k = 0;
  for (uint32_t i = 0; i < ws.n_paths; ++i)
    if (g_ref_reach_pos(g, ws.paths[i].start) != g_ref_reach_pos(g, ws.paths[i].end))
    {
      if (k != i)
        ws.paths[k] = ws.paths[i];
      ++k;
    }
  ws.n_paths = k;
}

GTX_DEV uint32_t site_order(GraphView const & g, uint32_t site)
{
  return g.ref_order[site] + g.ref_len[site]; // order of the site's variant nodes
}

GTX_DEV void remove_support_from_read_ends(GraphView const & g, AlignWorkspace & ws) // :382-432
{
  for (uint32_t i = 0; i < ws.n_paths; ++i)
  {
    DPath & p = ws.paths[i];
    if (p.nvar == 0)
      continue;
    bool const ss = g_is_special(g, p.start), es = g_is_special(g, p.end);
    if (!ss && !es)
      continue;
    // std::minmax_element: first smallest, last largest
    uint32_t imin = 0, imax = 0;
    for (uint32_t k = 1; k < p.nvar; ++k)
    {
      if (site_order(g, p.v[k].site) < site_order(g, p.v[imin].site))
        imin = k;
      if (!(site_order(g, p.v[k].site) < site_order(g, p.v[imax].site)))
        imax = k;
    }
    if (es && static_cast<int64_t>(g_actual_pos(g, p.end)) <= static_cast<int64_t>(site_order(g, p.v[imax].site)) + 4)
      pv_set(p.v[imax], 0);
    if (ss)
    {
      bool ambiguous = true;
      if (g_is_special(g, p.start + 4u))
        ambiguous = g_ref_reach_pos(g, p.start) != g_ref_reach_pos(g, p.start + 4u);
      if (ambiguous)
        pv_set(p.v[imin], 0);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// GenotypePaths::walk_read_ends / walk_read_starts (genotype_paths.cpp:483-553 / 555-621)
// ---------------------------------------------------------------------------------------------------------------
GTX_DEV void walk_read(GraphView const & g, AlignWorkspace & ws, bool starts)
{
  uint32_t const L = ws.read_len;
  if (ws.n_paths == 0 || path_size(ws.paths[0]) == L)
    return;
  if (ws.n_paths > MAX_SEED_NUMBER_FOR_WALKING)
    return;
  int maximum_mismatches = -1;
  if (ws.n_paths > MAX_SEED_NUMBER_ALLOWING_MISMATCHES)
    maximum_mismatches = 0;
  uint32_t best = 7;
  ws.n_wl = 0;
  ws.n_wlists = 0;
  ws.wl_off[0] = 0;
  for (uint32_t i = 0; i < ws.n_paths; ++i)
  {
    DPath const & path = ws.paths[i];
    SubRead sr;
    sr.rd = ws.rd;
    uint32_t n_locs;
    if (starts)
    {
      if (path.rs == 0)
        continue;
      sr.begin = 0;
      sr.len = path.rs + 1u;
      n_locs = get_locations(g, path.start, path, ws.locs, AlignCfg::LOC_CAP, ws.status);
    }
    else
    {
      if (path.re == L - 1)
        continue;
      n_locs = get_locations(g, path.end, path, ws.locs, AlignCfg::LOC_CAP, ws.status);
      sr.begin = path.re;
      sr.len = L - path.re;
    }
    if (ws.status)
      return;
    if (n_locs == 0 || n_locs > MAX_NUM_LOCATIONS_PER_PATH)
      continue;
    uint32_t mm;
    if (maximum_mismatches < 0)
    {
      uint32_t const budget = 2 + sr.len / 11;
      mm = budget < best ? budget : best;
    }
    else
      mm = static_cast<uint32_t>(maximum_mismatches);
    uint32_t const nl = iterative_dfs(g, ws, n_locs, starts, sr, mm);
    if (ws.status)
      return;
    if (nl == 0)
      continue;
    if (mm < best)
    {
      ws.n_wl = 0;
      ws.n_wlists = 0;
      best = mm;
    }
    if (mm == best)
    {
      if (ws.n_wlists >= AlignCfg::WLISTS || ws.n_wl + nl > AlignCfg::WL_CAP)
      {
        ws.status |= GTX_ST_DFS_OVERFLOW;
        return;
      }
      for (uint32_t k = 0; k < nl; ++k)
        ws.wl[ws.n_wl + k] = ws.dfs_out[k];
      ws.wl_idx[ws.n_wlists] = starts ? path.rs : path.re;
      ws.n_wl += nl;
      ++ws.n_wlists;
      ws.wl_off[ws.n_wlists] = ws.n_wl;
    }
  }
  for (uint32_t k = 0; k < ws.n_wlists; ++k)
  {
    DevLabel const * ll = ws.wl + ws.wl_off[k];
    uint32_t const n = ws.wl_off[k + 1] - ws.wl_off[k];
    if (starts)
      add_kmer_labels(ws, ll, n, 0, ws.wl_idx[k], best, true);
    else
      add_kmer_labels(ws, ll, n, ws.wl_idx[k], L - 1, best, false);
    if (ws.status)
      return;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k-mer keys (src/utilities/type_conversions.cpp:207-288) and index probes (src/index/ph_index.cpp:66-107)
// ---------------------------------------------------------------------------------------------------------------

// to_uint64_vec for a k-mer with ambiguous bases; serial, rare.  Returns the number of keys (0 = gave up, > 97 partial keys)
GTX_DEV uint32_t expand_keys(uint8_t const * rd, uint32_t at, uint64_t * keys)
{
  uint32_t n = 1;
  keys[0] = 0;
  for (uint32_t i = at; i < at + K; ++i)
  {
    uint32_t const origin = n;
    if (origin > 97)
      return 0;
    uint32_t const code = rd[i] & 15u;
    for (uint32_t u = 0; u < origin; ++u)
    {
      if (code == 15u || code == 0u)
      {
        keys[n++] = keys[u] * 4 + 0;
        keys[n++] = keys[u] * 4 + 1;
        keys[n++] = keys[u] * 4 + 2;
        keys[u] = (keys[u] << 2) + 3;
      }
      else
      {
        int left = __builtin_popcount(code);
        for (uint32_t b = 0; b < 4; ++b)
        {
          if (!(code & (1u << b)))
            continue;
          if (left == 1)
            keys[u] = keys[u] * 4 + b;
          else
            keys[n++] = keys[u] * 4 + b;
          --left;
        }
      }
    }
  }
  return n;
}

// PHIndex lookup of one key: (offset, count) of its labels, count 0 when absent
GTX_DEV void index_find(IndexView const & ix, uint64_t key, uint32_t & off, uint32_t & cnt)
{
  uint64_t const mask = (1ull << ix.log2_cap) - 1;
  uint64_t h = hash_key(key, ix.log2_cap);
  for (;;)
  {
    IndexSlot const s = ix.slots[h];
    if (s.cnt == 0)
    {
      off = 0;
      cnt = 0;
      return;
    }
    if (s.key == key)
    {
      off = s.off;
      cnt = s.cnt;
      return;
    }
    h = (h + 1) & mask;
  }
}

GTX_DEV uint64_t spread_bits(uint32_t x) // bit i -> bit 2i
{
  uint64_t v = x;
  v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
  v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
  v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
  v = (v | (v << 2)) & 0x3333333333333333ull;
  v = (v | (v << 1)) & 0x5555555555555555ull;
  return v;
}

GTX_DEV uint32_t reverse_bits32(uint32_t x)
{
  x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
  x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
  x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
  x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
  return (x >> 16) | (x << 16);
}

// Wave-parallel probe of a key list (`nkeys` keys: either keybuf[0..nkeys) or the 96 Hamming-1 neighbours of `base`
// generated on the fly) with the multi_get rule: a list of more than one key whose hits total more than
// max_index_labels yields nothing.  Labels land in ws.lbl in key order, bucket order inside a key.
template <class W>
GTX_DEV void probe_list(IndexView const & ix, AlignWorkspace & ws, bool hamming, uint64_t base, uint32_t nkeys)
{
  uint32_t const lane = W::lane();
  constexpr uint32_t ROUNDS = (AlignCfg::KEY_CAP + 63) / 64;
  uint32_t off[ROUNDS], cnt[ROUNDS], pre[ROUNDS];
  uint32_t total = 0;
  uint32_t const rounds = (nkeys + 63) / 64;
  for (uint32_t t = 0; t < ROUNDS; ++t)
  {
    if (t >= rounds)
      break;
    uint32_t const j = t * 64 + lane;
    off[t] = 0;
    cnt[t] = 0;
    if (j < nkeys)
    {
      uint64_t key;
      if (hamming)
        key = base ^ (static_cast<uint64_t>(j % 3 + 1) << (2 * (j / 3))); // type_conversions.cpp:272-288
      else
        key = ws.u.keybuf[j];
      index_find(ix, key, off[t], cnt[t]);
    }
    uint32_t round_total;
    pre[t] = total + W::excl_scan(cnt[t], round_total);
    total += round_total;
  }
  if (nkeys > 1 && total > ix.max_index_labels)
    total = 0; // ph_index.cpp:84-89
  if (total > AlignCfg::LBL_CAP)
  {
    if (lane == 0)
      ws.status |= GTX_ST_LABEL_OVERFLOW;
    total = 0;
  }
  if (total != 0)
    for (uint32_t t = 0; t < ROUNDS; ++t)
    {
      if (t >= rounds)
        break;
      for (uint32_t k = 0; k < cnt[t]; ++k)
        ws.lbl[pre[t] + k] = ix.labels[off[t] + k];
    }
  if (lane == 0)
    ws.n_lbl = total;
  W::sync();
}

// ---------------------------------------------------------------------------------------------------------------
// one (read, orientation): find_genotype_paths_of_one_of_the_sequences (alignment.cpp:23-103)
// ---------------------------------------------------------------------------------------------------------------
template <class W>
GTX_DEV void align_one(GraphView const & g, IndexView const & ix, AlignWorkspace & ws, uint8_t const * seq4, uint32_t len,
                       bool reverse, uint32_t * rec, uint32_t rec_words)
{
  uint32_t const lane = W::lane();
  // -- load the read: BAM nibbles -> one code per byte; the reverse orientation is the reverse complement, and
  //    complementing an IUPAC code is reversing its 4 bits (A<->T, C<->G)
  for (uint32_t i = lane; i < len; i += 64)
  {
    uint32_t const src = reverse ? (len - 1 - i) : i;
    uint32_t c = (seq4[src >> 1] >> ((~src & 1u) << 2)) & 15u;
    if (c == 0)
      c = 15;
    if (reverse)
      c = ((c & 1u) << 3) | ((c & 2u) << 1) | ((c & 4u) >> 1) | ((c & 8u) >> 3);
    ws.rd[i] = static_cast<uint8_t>(c);
  }
  if (lane == 0)
  {
    ws.n_paths = 0;
    ws.longest = 0;
    ws.status = 0;
    ws.read_len = len;
  }
  W::sync();

  uint32_t const n_k = len < K ? 0 : 1 + (len - K) / (K - 1); // kmer_help_functions.cpp:10-17
  // -- exact keys of every k-mer.  Unambiguous k-mer: lanes 0..31 each hold one base, two ballots give the low/high
  //    bit planes, interleaving them gives the key (first base in the top bits, type_conversions.cpp:75-87).
  for (uint32_t i = 0; i < n_k; ++i)
  {
    uint32_t const c = lane < K ? ws.rd[(K - 1) * i + lane] : 1u;
    bool const single = (c & (c - 1u)) == 0u && c != 0u;
    uint64_t const amb = W::ballot(!single);
    uint32_t const two = (c == 2u) ? 1u : (c == 4u) ? 2u : (c == 8u) ? 3u : 0u;
    uint32_t const b0 = static_cast<uint32_t>(W::ballot(lane < K && (two & 1u)));
    uint32_t const b1 = static_cast<uint32_t>(W::ballot(lane < K && (two & 2u)));
    if (amb == 0)
    {
      uint64_t const key = spread_bits(reverse_bits32(b0)) | (spread_bits(reverse_bits32(b1)) << 1);
      uint32_t off, cnt;
      index_find(ix, key, off, cnt);
      if (lane == 0)
      {
        ws.key0[i] = key;
        ws.nkeys0[i] = 1;
        ws.off0[i] = off;
        ws.cnt0[i] = cnt;
      }
    }
    else if (lane == 0)
    {
      ws.nkeys0[i] = 2; // "not a single key"; the list is generated when the k-mer is processed
      ws.cnt0[i] = 0;
    }
  }
  W::sync();
  // -- stop if every k-mer is extremely common (alignment.cpp:35-49); only single-key lists can reach 512 labels
  bool all_common = n_k > 0;
  for (uint32_t i = 0; i < n_k; ++i)
    if (!(ws.nkeys0[i] == 1 && ws.cnt0[i] >= MAX_UNIQUE_KMER_POSITIONS))
      all_common = false;

  if (!all_common && n_k > 0)
  {
    for (uint32_t i = 0; i < n_k; ++i)
    {
      uint32_t const rs = (K - 1) * i, re = rs + (K - 1);
      bool const single = ws.nkeys0[i] == 1;
      if (single)
      {
        // exact list: one key, never cut (ph_index.cpp:84)
        uint32_t const cnt = ws.cnt0[i], off = ws.off0[i];
        uint32_t n = cnt;
        if (cnt > AlignCfg::LBL_CAP)
        {
          if (lane == 0)
            ws.status |= GTX_ST_LABEL_OVERFLOW;
          n = 0;
        }
        for (uint32_t k = lane; k < n; k += 64)
          ws.lbl[k] = ix.labels[off + k];
        if (lane == 0)
          ws.n_lbl = n;
        W::sync();
      }
      else
      {
        if (lane == 0)
          ws.n_keys = expand_keys(ws.rd, rs, ws.u.keybuf);
        W::sync();
        probe_list<W>(ix, ws, false, 0, ws.n_keys);
      }
      if (lane == 0 && !ws.status)
        add_kmer_labels(ws, ws.lbl, ws.n_lbl, rs, re, 0, false);
      W::sync();
      // Hamming-1 list: the 96 neighbours of a unique exact key, else the exact list again
      // (kmer_help_functions.cpp:97-119 keeps multi-key lists as they are)
      if (single)
        probe_list<W>(ix, ws, true, ws.key0[i], 96);
      // (multi-key list: ws.lbl still holds exactly what multi_get returns for it)
      if (lane == 0 && !ws.status)
        add_kmer_labels(ws, ws.lbl, ws.n_lbl, rs, re, 1, false);
      W::sync();
    }
    if (lane == 0 && !ws.status)
    {
      remove_short_paths(ws);
      walk_read(g, ws, true);
      if (!ws.status)
        walk_read(g, ws, false);
      if (!ws.status)
      {
        update_longest(ws);
        remove_short_paths(ws);
        remove_paths_with_too_many_mismatches(ws);
        if (g.is_sv_graph)
          remove_fully_special_paths(g, ws);
        remove_non_ref_paths_when_read_matches_ref(g, ws);
        update_longest(ws);
        remove_short_paths(ws);
        if (g.is_sv_graph)
          remove_support_from_read_ends(g, ws);
      }
    }
    W::sync();
  }

  // -- result record (layout: include/gtx.h, gtx_align_batch)
  if (lane == 0)
  {
    uint32_t status = ws.status;
    uint32_t np = status ? 0 : ws.n_paths;
    uint32_t w = 2;
    for (uint32_t i = 0; i < np; ++i)
    {
      DPath const & p = ws.paths[i];
      if (w + 4 + 3 * p.nvar > rec_words)
      {
        status |= GTX_ST_RECORD_OVERFLOW;
        np = 0;
        break;
      }
      rec[w++] = p.start;
      rec[w++] = p.end;
      rec[w++] = static_cast<uint32_t>(p.rs) | (static_cast<uint32_t>(p.re) << 16);
      rec[w++] = static_cast<uint32_t>(p.mism) | (static_cast<uint32_t>(p.nvar) << 16);
      for (uint32_t k = 0; k < p.nvar; ++k)
      {
        rec[w++] = p.v[k].site;
        rec[w++] = p.v[k].mlo;
        rec[w++] = p.v[k].mhi;
      }
    }
    rec[0] = np | (status << 16);
    rec[1] = ((status || np == 0) ? 0 : ws.longest) | (len << 16);
  }
  W::sync();
}

// align_read (alignment.cpp:331-363): which orientations a record gets
GTX_DEV bool needs_reverse(gtx_read_meta const & m, bool force_both)
{
  bool const one = (m.flag & 1u) == 0u || (m.tid == m.mtid && m.isize > -1200 && m.isize < 1200 &&
                                           (((m.flag & 16u) != 0u) != ((m.flag & 32u) != 0u)));
  return !one || force_both;
}

} // namespace gtx
