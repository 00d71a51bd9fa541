// gtx_host.cpp -- host side of libgtx: graph flattening, special positions, k-mer index construction.
//
// Index construction here is NOT the reference's forward sweep with a deque of partial k-mers
// (src/index/indexer.cpp:26-291).  It enumerates, for every end position in sweep order, the K-base walks that end
// there by walking the graph BACKWARDS, branching over the alleles of each site in ascending order.  That visits
// walks in exactly the order the reference's sublists hold them (later site = more significant), so
// bucket order -- which decides Path order downstream -- is identical, while every end position is independent
// (no sweep state), which is what a later GPU build of the index needs.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>

#include "gtx_flat.hpp"
#include "index_build.hpp"

namespace gtx
{
GraphView HostGraph::view() const
{
  GraphView v{};
  v.n_ref = static_cast<uint32_t>(ref_order.size());
  v.n_var = static_cast<uint32_t>(var_order.size());
  v.n_special = static_cast<uint32_t>(special_actual.size());
  v.n_bucket = static_cast<uint32_t>(pos_bucket.size());
  v.first_order = ref_order.empty() ? 0 : ref_order[0];
  v.padding = padding;
  v.is_sv_graph = is_sv_graph;
  v.ref_order = ref_order.data();
  v.ref_len = ref_len.data();
  v.ref_dna = ref_dna.data();
  v.ref_nvar = ref_nvar.data();
  v.ref_first_var = ref_first_var.data();
  v.var_order = var_order.data();
  v.var_len = var_len.data();
  v.var_dna = var_dna.data();
  v.var_out_ref = var_out_ref.data();
  v.site_ref_reach = site_ref_reach.data();
  v.site_special_base = site_special_base.data();
  v.special_ref_reach = special_ref_reach.data();
  v.special_actual = special_actual.data();
  v.pos_bucket = pos_bucket.data();
  v.pos_info = pos_info.empty() ? nullptr : pos_info.data();
  v.pos_back = pos_back.empty() ? nullptr : pos_back.data();
  v.pos_node = pos_node.empty() ? nullptr : pos_node.data();
  v.n_pos_info = static_cast<uint32_t>(pos_info.size());
  v.dna = codes.data();
  v.tri_off = tri_off.data();
  v.allele_off = allele_off.data();
  v.total_tri = total_tri;
  v.total_allele = total_allele;
  v.near_last = near_last.data();
  v.near_off = near_off.data();
  v.total_near = total_near;
  v.n_hap = n_hap;
  return v;
}

std::string flatten_graph(gtx_graph_view const & g, gtx_params const & par, HostGraph & out, bool host_positions)
{
  if (g.n_ref == 0)
    return "graph has no reference node";
  if (!g.ref_order || !g.ref_len || !g.ref_dna_off || !g.ref_nvar || !g.ref_first_var || !g.dna)
    return "NULL reference-node table";
  if (g.n_var && (!g.var_order || !g.var_len || !g.var_dna_off || !g.var_out_ref))
    return "NULL variant-node table";
  uint32_t const R = g.n_ref, V = g.n_var;
  out = HostGraph();
  out.is_sv_graph = par.is_sv_graph != 0;
  out.padding = (par.is_sv_graph || par.is_segment_calling) ? 1000000u : 1000u; // graph.cpp:973
  out.ref_order.assign(g.ref_order, g.ref_order + R);
  out.ref_len.assign(g.ref_len, g.ref_len + R);
  out.ref_nvar.assign(g.ref_nvar, g.ref_nvar + R);
  out.ref_first_var.assign(g.ref_first_var, g.ref_first_var + R);
  out.var_order.assign(g.var_order, g.var_order + V);
  out.var_len.assign(g.var_len, g.var_len + V);
  out.var_out_ref.assign(g.var_out_ref, g.var_out_ref + V);
  // repack the sequence arena: ref nodes then var nodes, each contiguous
  out.ref_dna.resize(R);
  out.var_dna.resize(V);
  for (uint32_t r = 0; r < R; ++r)
  {
    if (static_cast<uint64_t>(g.ref_dna_off[r]) + g.ref_len[r] > g.dna_len)
      return "reference node sequence outside the arena";
    out.ref_dna[r] = static_cast<uint32_t>(out.dna.size());
    out.dna.append(g.dna + g.ref_dna_off[r], g.ref_len[r]);
  }
  for (uint32_t v = 0; v < V; ++v)
  {
    if (static_cast<uint64_t>(g.var_dna_off[v]) + g.var_len[v] > g.dna_len)
      return "variant node sequence outside the arena";
    if (g.var_len[v] == 0)
      return "variant node with empty sequence";
    out.var_dna[v] = static_cast<uint32_t>(out.dna.size());
    out.dna.append(g.dna + g.var_dna_off[v], g.var_len[v]);
  }
  // comparison codes: IUPAC letters keep their BAM 4-bit code, '<' / '>' kill a walk, anything else matches nothing
  out.codes.resize(out.dna.size());
  {
    uint8_t code_of[256];
    std::memset(code_of, 0x40, sizeof(code_of));
    char const * tbl = "=ACMGRSVTWYHKDBN";
    for (int k = 1; k < 16; ++k)
      code_of[static_cast<uint8_t>(tbl[k])] = static_cast<uint8_t>(k);
    code_of[static_cast<uint8_t>('<')] = code_of[static_cast<uint8_t>('>')] = 0x80;
    for (size_t i = 0; i < out.dna.size(); ++i)
      out.codes[i] = static_cast<char>(code_of[static_cast<uint8_t>(out.dna[i])]);
  }
  // structure: ref r -> vars [first, first+nvar) -> ref r+1
  uint32_t next_var = 0;
  for (uint32_t r = 0; r < R; ++r)
  {
    uint32_t const n = out.ref_nvar[r];
    if (r + 1 == R)
    {
      if (n != 0)
        return "last reference node has outgoing variants";
      break;
    }
    if (n == 0)
      return "inner reference node without a variant site";
    if (n > MAX_ALLELES)
      return "unsupported: a site has more alleles than the reference's MAX_NUMBER_OF_HAPLOTYPES (2560)";
    if (out.ref_first_var[r] != next_var)
      return "variant nodes are not laid out site by site";
    if (next_var + n > V)
      return "variant node index out of range";
    for (uint32_t i = 0; i < n; ++i)
    {
      if (out.var_out_ref[next_var + i] != r + 1)
        return "variant node does not lead to the next reference node";
      if (out.var_order[next_var + i] != out.ref_order[r] + out.ref_len[r])
        return "variant node order does not follow its reference node";
    }
    if (out.ref_order[r + 1] != out.var_order[next_var] + out.var_len[next_var])
      return "reference node order does not follow the reference allele";
    next_var += n;
  }
  if (next_var != V)
    return "dangling variant nodes";
  // events
  if (g.event_off && g.event_val)
  {
    out.event_off.assign(g.event_off, g.event_off + 2 * V + 1);
    out.event_val.assign(g.event_val, g.event_val + out.event_off.back());
    if (out.event_off.back() == 0)
    {
      out.event_off.clear();
      out.event_val.clear();
    }
  }
  // special positions: Graph::create_special_positions (graph.cpp:384-407) + add_special_pos (:1759-1773)
  out.site_ref_reach.assign(R, 0);
  out.site_special_base.assign(R, INVALID);
  for (uint32_t r = 0; r + 1 < R; ++r)
  {
    uint32_t const fv = out.ref_first_var[r], n = out.ref_nvar[r];
    uint32_t const ref_reach = out.var_order[fv] + out.var_len[fv] - 1;
    out.site_ref_reach[r] = ref_reach;
    if (n <= 1)
      continue;
    uint32_t max_reach = 0;
    for (uint32_t i = 1; i < n; ++i)
      max_reach = std::max(max_reach, out.var_order[fv + i] + out.var_len[fv + i] - 1);
    if (max_reach > ref_reach)
      out.site_special_base[r] = static_cast<uint32_t>(out.special_actual.size());
    for (uint32_t reach = ref_reach + 1; reach <= max_reach; ++reach)
    {
      out.special_ref_reach.push_back(ref_reach);
      out.special_actual.push_back(reach);
    }
  }
  // position -> reference node buckets
  uint32_t const first = out.ref_order[0];
  uint32_t const last = out.ref_order[R - 1] + out.ref_len[R - 1];
  uint32_t const nb = ((last - first) >> POS_BUCKET_SHIFT) + 2;
  out.pos_bucket.assign(nb, 0);
  uint32_t rr = 0;
  for (uint32_t b = 0; b < nb; ++b)
  {
    uint32_t const p = first + (b << POS_BUCKET_SHIFT);
    while (rr + 1 < R && out.ref_order[rr + 1] <= p)
      ++rr;
    out.pos_bucket[b] = rr;
  }
  // position -> where its base is and how far its reference node goes on (the tail compare of the simple reads)
  out.pos_info.clear();
  out.pos_back.clear();
  out.pos_node.clear();
  out.pos_table_len = out.codes.size() < (1u << 24) ? last - first : 0u;
  if (out.pos_table_len != 0 && host_positions)
  {
    out.pos_info.assign(last - first, INVALID);
    out.pos_back.assign(last - first, 0);
    out.pos_node.assign(last - first, INVALID);
    for (uint32_t r = 0; r < R; ++r)
      for (uint32_t d = 0; d < out.ref_len[r]; ++d)
      {
        uint32_t const room = out.ref_len[r] - d;
        out.pos_info[out.ref_order[r] - first + d] = ((out.ref_dna[r] + d) << 8) | (room < 255 ? room : 255);
        out.pos_back[out.ref_order[r] - first + d] = static_cast<uint8_t>(d < 255 ? d : 255);
        out.pos_node[out.ref_order[r] - first + d] = r;
      }
  }
  // haplotype h <-> site h (Graph::get_all_haplotypes, graph.cpp:680-704); accumulator offsets
  out.n_hap = V == 0 ? 0 : R - 1;
  out.tri_off.assign(R, 0);
  out.allele_off.assign(R, 0);
  for (uint32_t r = 0; r + 1 < R; ++r)
  {
    uint64_t const c = out.ref_nvar[r];
    out.tri_off[r] = out.total_tri;
    out.allele_off[r] = out.total_allele;
    out.total_tri += c * (c + 1) / 2;
    out.total_allele += c;
  }
  // windows of the near-pair connection counters: the haplotypes whose order is < 100 above this one's
  out.near_last.assign(R, 0);
  out.near_off.assign(R, 0);
  for (uint32_t r = 0; r + 1 < R; ++r)
  {
    uint32_t last = r;
    uint64_t const order = out.var_order[out.ref_first_var[r]];
    while (last + 2 < R && out.var_order[out.ref_first_var[last + 1]] < order + 100)
      ++last;
    out.near_last[r] = last;
    out.near_off[r] = out.total_near;
    out.total_near += static_cast<uint64_t>(out.ref_nvar[r]) * (out.allele_off[last] + out.ref_nvar[last] - out.allele_off[r] - out.ref_nvar[r]);
  }
  return "";
}

namespace
{
// The first and the last 31 bases of every node as 2-bit codes (a walk takes a node's bases from its end backwards, or --
// its first piece -- from a position within 31 bases of a node's start back to that start: both are one shift and mask of
// these instead of a loop over the bases).  sfx: the last base in bits 0-1, the one before it in bits 2-3, ...; pfx: base 0
// in the highest of its 2 m bits (m = min(31, length)); *_ok: how many bases from that end are A/C/G/T.
struct NodePacks
{
  std::vector<uint64_t> ref_sfx, ref_pfx, var_sfx, var_pfx;
  std::vector<uint8_t> ref_sfx_ok, ref_pfx_ok, var_sfx_ok, var_pfx_ok;
  static void pack(char const * dna, uint32_t len, uint64_t & sfx, uint8_t & sfx_ok, uint64_t & pfx, uint8_t & pfx_ok)
  {
    auto code = [](char c) -> int { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; };
    uint32_t const m = len < K - 1 ? len : K - 1;
    sfx = pfx = 0;
    sfx_ok = pfx_ok = 0;
    bool ok = true;
    for (uint32_t i = 0; i < m; ++i)
    {
      int const c = code(dna[len - 1 - i]);
      ok = ok && c >= 0;
      sfx |= static_cast<uint64_t>(c < 0 ? 0 : c) << (2 * i);
      sfx_ok = static_cast<uint8_t>(sfx_ok + (ok ? 1 : 0));
    }
    ok = true;
    for (uint32_t j = 0; j < m; ++j)
    {
      int const c = code(dna[j]);
      ok = ok && c >= 0;
      pfx |= static_cast<uint64_t>(c < 0 ? 0 : c) << (2 * (m - 1 - j));
      pfx_ok = static_cast<uint8_t>(pfx_ok + (ok ? 1 : 0));
    }
  }
  explicit NodePacks(HostGraph const & g)
  {
    size_t const R = g.ref_order.size(), V = g.var_order.size();
    ref_sfx.resize(R); ref_pfx.resize(R); ref_sfx_ok.resize(R); ref_pfx_ok.resize(R);
    var_sfx.resize(V); var_pfx.resize(V); var_sfx_ok.resize(V); var_pfx_ok.resize(V);
    for (size_t r = 0; r < R; ++r)
      pack(g.dna.data() + g.ref_dna[r], g.ref_len[r], ref_sfx[r], ref_sfx_ok[r], ref_pfx[r], ref_pfx_ok[r]);
    for (size_t v = 0; v < V; ++v)
      pack(g.dna.data() + g.var_dna[v], g.var_len[v], var_sfx[v], var_sfx_ok[v], var_pfx[v], var_pfx_ok[v]);
  }
};

struct Walker
{
  HostGraph const & g;
  std::vector<Emit> & out;
  bool has_events;
  NodePacks const & packs;
  // walk state (backwards): var nodes visited, how many bases of each were used
  uint32_t vars[K];
  uint32_t used[K];
  uint32_t n_vars = 0;
  uint32_t end_label = 0;

  static int code(char c)
  {
    switch (c)
    {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return -1;
    }
  }

  uint32_t special_of(uint32_t site, uint32_t pos) const // Graph::get_special_pos (graph.cpp:1775-1782)
  {
    uint32_t const rr = g.site_ref_reach[site];
    return pos > rr ? SPECIAL_START + g.site_special_base[site] + (pos - rr - 1) : pos;
  }

  // the pruning rules of the forward sweep, evaluated on a finished walk (see file header / DESIGN.md)
  bool walk_is_kept() const
  {
    // entry_has_too_many_nonrefs (indexer.cpp:13-20): counted over the non-reference alleles of the walk
    uint32_t cnt = 0;
    uint64_t num = 1;
    for (uint32_t i = 0; i < n_vars; ++i)
    {
      uint32_t const v = vars[i];
      uint32_t const site = g.var_out_ref[v] - 1;
      if (v != g.ref_first_var[site])
      {
        ++cnt;
        num = std::min<uint64_t>(num * g.ref_nvar[site], 1u << 30);
      }
    }
    if (cnt > 1 && (num > 181 || cnt > 4))
      return false;
    if (!has_events)
      return true;
    // anti events (indexer.cpp:114-140): walking forward, a node is refused when one of its events is among the anti
    // events gathered so far; a node's own anti events count from its second base on.
    // (the anti events so far are ranges of event_val, one per node behind: no list is built -- this runs once per k-mer)
    for (uint32_t i = n_vars; i-- > 0;) // vars[] is in backward order
    {
      uint32_t const v = vars[i];
      int64_t const * ev = g.event_val.data() + g.event_off[2 * v];
      uint32_t const n_ev = g.event_off[2 * v + 1] - g.event_off[2 * v];
      int64_t const * an = g.event_val.data() + g.event_off[2 * v + 1];
      uint32_t const n_an = g.event_off[2 * v + 2] - g.event_off[2 * v + 1];
      if (n_ev == 0)
        continue;
      for (uint32_t j = n_vars; j-- > i + 1;) // the nodes in front of this one on the walk
      {
        uint32_t const u = vars[j];
        int64_t const * au = g.event_val.data() + g.event_off[2 * u + 1];
        int64_t const * au_end = g.event_val.data() + g.event_off[2 * u + 2];
        for (uint32_t e = 0; e < n_ev; ++e)
          if (std::find(au, au_end, ev[e]) != au_end)
            return false;
      }
      if (used[i] >= 2)
        for (uint32_t e = 0; e < n_ev; ++e)
          if (std::find(an, an + n_an, ev[e]) != an + n_an)
            return false;
    }
    return true;
  }

  void finish(uint64_t key, uint32_t start_label)
  {
    if (!walk_is_kept())
      return;
    if (n_vars == 0)
    {
      out.push_back({key, {start_label, end_label, INVALID}});
      return;
    }
    // IndexEntry::variant_id is a std::set: ascending ids.  vars[] is descending already (backward walk).
    for (uint32_t i = n_vars; i-- > 0;)
      out.push_back({key, {start_label, end_label, vars[i]}});
  }

  // take bases [.., off] of a node backwards; `have` bases collected so far in `key` (placed from the low end up).  A walk
  // enters a node at its last base, or starts -- have = 0 -- within 31 bases of a node's start: one shift and mask of the
  // node's packed ends (NodePacks); a start deeper inside a long allele walks base by base.
  void back_ref(uint32_t r, int64_t off, uint32_t have, uint64_t key)
  {
    uint32_t const len = g.ref_len[r];
    uint32_t const n_here = static_cast<uint32_t>(off) + 1u, need = K - have, n = n_here < need ? n_here : need;
    uint32_t const m = len < K - 1 ? len : K - 1;
    if (static_cast<uint32_t>(off) + 1u == len && n <= m)
    {
      if (packs.ref_sfx_ok[r] < n)
        return;
      key |= (packs.ref_sfx[r] & ((1ull << (2 * n)) - 1ull)) << (2 * have);
    }
    else if (have == 0 && n_here <= m)
    {
      if (packs.ref_pfx_ok[r] < n_here)
        return;
      key = packs.ref_pfx[r] >> (2 * (m - n_here)); // (n = n_here: fewer than K bases to the node's start)
    }
    else
    {
      char const * dna = g.dna.data() + g.ref_dna[r];
      for (uint32_t i = 0; i < n; ++i)
      {
        int const c = code(dna[off - i]);
        if (c < 0)
          return;
        key |= static_cast<uint64_t>(c) << (2 * (have + i));
      }
    }
    have += n;
    if (have == K)
    {
      finish(key, g.ref_order[r] + static_cast<uint32_t>(off) - (n - 1u));
      return;
    }
    if (r == 0)
      return;
    uint32_t const site = r - 1, fv = g.ref_first_var[site], nv = g.ref_nvar[site];
    for (uint32_t a = 0; a < nv; ++a)
      back_var(fv + a, static_cast<int64_t>(g.var_len[fv + a]) - 1, have, key);
  }

  void back_var(uint32_t v, int64_t off, uint32_t have, uint64_t key)
  {
    uint32_t const site = g.var_out_ref[v] - 1, len = g.var_len[v];
    uint32_t const slot = n_vars++;
    vars[slot] = v;
    uint32_t const n_here = static_cast<uint32_t>(off) + 1u, need = K - have, n = n_here < need ? n_here : need;
    uint32_t const m = len < K - 1 ? len : K - 1;
    bool ok = true;
    if (static_cast<uint32_t>(off) + 1u == len && n <= m)
    {
      ok = packs.var_sfx_ok[v] >= n;
      key |= (packs.var_sfx[v] & ((1ull << (2 * n)) - 1ull)) << (2 * have);
    }
    else if (have == 0 && n_here <= m)
    {
      ok = packs.var_pfx_ok[v] >= n_here;
      key = packs.var_pfx[v] >> (2 * (m - n_here));
    }
    else
    {
      char const * dna = g.dna.data() + g.var_dna[v];
      for (uint32_t i = 0; i < n && ok; ++i)
      {
        int const c = code(dna[off - i]);
        ok = c >= 0;
        key |= static_cast<uint64_t>(c < 0 ? 0 : c) << (2 * (have + i));
      }
    }
    if (ok)
    {
      used[slot] = n;
      have += n;
      if (have == K)
        finish(key, special_of(site, g.var_order[v] + static_cast<uint32_t>(off) - (n - 1u)));
      else
        back_ref(site, static_cast<int64_t>(g.ref_len[site]) - 1, have, key);
    }
    --n_vars;
  }
};

} // namespace

// ---- a team of host threads for the index build (GTX_HOST_THREADS overrides; the build is ~1 M keys per Mb of graph)
static unsigned host_threads()
{
  if (char const * e = std::getenv("GTX_HOST_THREADS"))
    return static_cast<unsigned>(std::max(1, std::atoi(e)));
  unsigned const hw = std::thread::hardware_concurrency();
  return std::max(1u, std::min(hw ? hw : 1u, 8u)); // (measured on a 256-thread host: 1 / 4 / 8 / 16 / 32 threads = 0.25 / 0.21 / 0.17 / 0.23 / 0.42 s:
                                                   //  the stages are short and memory-bound, a larger team costs more to start than it saves)
}

// runs fn(t, begin, end) over T contiguous slices of [0, n)
template <class F>
static void parallel_slices(std::size_t n, unsigned T, F fn)
{
  if (T <= 1 || n < 4096)
  {
    fn(0u, std::size_t(0), n);
    return;
  }
  std::vector<std::thread> team;
  team.reserve(T);
  for (unsigned t = 0; t < T; ++t)
    team.emplace_back([=, &fn] { fn(t, n * t / T, n * (t + 1) / T); });
  for (auto & th : team)
    th.join();
}

// Stable LSD radix sort of (key, payload) pairs on the low `key_bits` bits of the key, 16 bits per pass; every pass is
// done by the team: per-slice histograms, offsets in (digit, slice) order -- which keeps it stable --, parallel scatter.
static void radix_sort_pairs(std::vector<std::pair<uint64_t, uint32_t>> & v, unsigned key_bits)
{
  std::size_t const n = v.size();
  unsigned const T = n < (1u << 16) ? 1u : host_threads();
  std::vector<std::pair<uint64_t, uint32_t>> tmp(n);
  std::vector<uint32_t> count(static_cast<std::size_t>(T) << 16);
  for (unsigned shift = 0; shift < key_bits; shift += 16)
  {
    std::fill(count.begin(), count.end(), 0u);
    parallel_slices(n, T, [&](unsigned t, std::size_t b, std::size_t e) {
      uint32_t * c = count.data() + (static_cast<std::size_t>(t) << 16);
      for (std::size_t i = b; i < e; ++i)
        ++c[(v[i].first >> shift) & 0xFFFFu];
    });
    uint32_t sum = 0;
    for (uint32_t d = 0; d < (1u << 16); ++d)
      for (unsigned t = 0; t < T; ++t)
      {
        uint32_t & c = count[(static_cast<std::size_t>(t) << 16) + d];
        uint32_t const k = c;
        c = sum;
        sum += k;
      }
    parallel_slices(n, T, [&](unsigned t, std::size_t b, std::size_t e) {
      uint32_t * c = count.data() + (static_cast<std::size_t>(t) << 16);
      for (std::size_t i = b; i < e; ++i)
        tmp[c[(v[i].first >> shift) & 0xFFFFu]++] = v[i];
    });
    v.swap(tmp);
  }
}

static void bucket_insert(std::vector<IndexSlot> & slots, uint32_t log2_buckets, IndexSlot const & s)
{
  uint64_t const mask = (1ull << log2_buckets) - 1;
  for (uint64_t b = hash_key(s.key, log2_buckets);; b = (b + 1) & mask)
    for (uint32_t k = 0; k < BUCKET_SLOTS; ++k)
      if (slots[b * BUCKET_SLOTS + k].cnt == 0)
      {
        slots[b * BUCKET_SLOTS + k] = s;
        return;
      }
}

// Inserts in bucket order: the table is written front to back instead of at a million random places (which bucket a
// spilled slot lands in depends on the order, what a lookup finds does not).  The team splits the sorted items at bucket
// boundaries; a thread may only write buckets of its own range, an item that would spill past it waits for the
// sequential sweep at the end (linear probing without deletions: any insertion order gives a table lookups can read).
static void bucket_insert_all(std::vector<IndexSlot> & slots, uint32_t log2_buckets, std::vector<IndexSlot> const & items)
{
  std::size_t const n = items.size();
  std::vector<std::pair<uint64_t, uint32_t>> order(n);
  unsigned const T = n < (1u << 16) ? 1u : host_threads();
  parallel_slices(n, T, [&](unsigned, std::size_t b, std::size_t e) {
    for (std::size_t i = b; i < e; ++i)
      order[i] = {hash_key(items[i].key, log2_buckets), static_cast<uint32_t>(i)};
  });
  radix_sort_pairs(order, log2_buckets);
  if (T <= 1)
  {
    for (auto const & o : order)
      bucket_insert(slots, log2_buckets, items[o.second]);
    return;
  }
  // slice boundaries moved forward to the next change of bucket
  std::vector<std::size_t> cut(T + 1);
  for (unsigned t = 0; t <= T; ++t)
  {
    std::size_t c = n * t / T;
    while (c > 0 && c < n && order[c].first == order[c - 1].first)
      ++c;
    cut[t] = c;
  }
  std::vector<std::vector<uint32_t>> late(T);
  std::vector<std::thread> team;
  for (unsigned t = 0; t < T; ++t)
    team.emplace_back([&, t] {
      std::size_t const b = cut[t], e = cut[t + 1];
      if (b >= e)
        return;
      uint64_t const limit = e < n ? order[e].first : (1ull << log2_buckets); // first bucket of the next slice
      for (std::size_t i = b; i < e; ++i)
      {
        IndexSlot const & s = items[order[i].second];
        bool placed = false;
        for (uint64_t bk = order[i].first; bk < limit && !placed; ++bk)
          for (uint32_t k = 0; k < BUCKET_SLOTS && !placed; ++k)
            if (slots[bk * BUCKET_SLOTS + k].cnt == 0)
            {
              slots[bk * BUCKET_SLOTS + k] = s;
              placed = true;
            }
        if (!placed)
          late[t].push_back(order[i].second);
      }
    });
  for (auto & th : team)
    th.join();
  for (auto const & l : late)
    for (uint32_t i : l)
      bucket_insert(slots, log2_buckets, items[i]);
}

IndexView HostIndex::view(uint32_t max_index_labels, uint32_t half_bucket_cap) const
{
  IndexView ix{};
  ix.slots = slots.data();
  ix.labels = dev_labels.data();
  ix.log2_cap = log2_cap;
  ix.max_index_labels = max_index_labels;
  ix.hslots = hslots.data();
  ix.hlist = hlist.data();
  ix.h_log2_cap = h_log2_cap;
  ix.half_bucket_cap = half_bucket_cap;
  ix.refp = refp.data();
  ix.pos_flags = pos_flags.data();
  ix.tail_info = tail_info.data();
  ix.filt[0] = filt[0].data();
  ix.filt[1] = filt[1].data();
  ix.hint_first = hint_first;
  ix.n_hint = n_hint;
  ix.filt_log2 = filt_log2;
  ix.win = win.data();
  ix.site_win = site_win.data();
  ix.win_base = win_base;
  ix.n_win = static_cast<uint32_t>(win.size());
  return ix;
}

// Tables of the position-hinted pass (IndexView::refp ..., hinted.hpp).  For every reference position: what a read
// k-mer that equals -- or is one substitution / one ambiguous base away from -- the reference 32-mer of that place
// would get from the global lookups, decided here once from the finished index.
void hint_graph_tables(HostGraph const & g, HintGraphTables & t)
{
  uint32_t const R = static_cast<uint32_t>(g.ref_order.size());
  t = HintGraphTables();
  t.hint_first = g.ref_order.empty() ? 0 : g.ref_order[0] - 1; // order = 1-based contig position
  if (R == 0 || R - 1 >= HINT_NO_SITE)
    return;
  // linear reference = reference nodes and allele 0 of every site, in order
  uint32_t const first = g.ref_order[0], last = g.ref_order[R - 1] + g.ref_len[R - 1];
  uint32_t const n = last - first;
  t.n = n;
  t.base.assign(n, 15); // nibble codes: an IUPAC letter's 4-bit code (A=1 C=2 G=4 T=8 ... N=15), 0 for anything else (hint_plane_code)
  t.room.assign(n, 0);  // bases to the end / from the start of the reference node (capped), 0 outside reference nodes
  t.back.assign(n, 0);
  auto nib = [](char c) -> uint8_t { return c == 'A' ? 1 : c == 'C' ? 2 : c == 'G' ? 4 : c == 'T' ? 8 : 15; };
  for (uint32_t r = 0; r < R; ++r)
  {
    uint32_t const at = g.ref_order[r] - first;
    for (uint32_t d = 0; d < g.ref_len[r]; ++d)
    {
      t.base[at + d] = static_cast<uint8_t>(hint_plane_code(static_cast<uint8_t>(g.codes[g.ref_dna[r] + d])));
      uint32_t const left = g.ref_len[r] - d;
      t.room[at + d] = static_cast<uint8_t>(left < 255 ? left : 255);
      t.back[at + d] = static_cast<uint8_t>(d < 255 ? d : 255);
    }
    if (r + 1 < R && g.ref_nvar[r] != 0)
    {
      uint32_t const v = g.ref_first_var[r], vat = g.var_order[v] - first;
      for (uint32_t d = 0; d < g.var_len[v]; ++d)
        t.base[vat + d] = static_cast<uint8_t>(hint_plane_code(static_cast<uint8_t>(g.codes[g.var_dna[v] + d])));
    }
  }
  // the site behind every reference node, as a walk at the read's end may cross it (tail_info)
  t.tail_info.assign(n, uint2_t{0, 0});
  for (uint32_t r = 0; r + 1 < R && !g.is_sv_graph; ++r)
  {
    uint32_t const fv = g.ref_first_var[r], nv = g.ref_nvar[r];
    bool snp = nv >= 2 && nv <= 4;
    uint32_t codes = 0;
    for (uint32_t a = 0; a < nv && snp; ++a)
    {
      uint8_t const c = g.var_len[fv + a] == 1 ? nib(g.dna[g.var_dna[fv + a]]) : 15;
      snp = c != 15; // (A, C, G or T)
      codes |= static_cast<uint32_t>(c) << (4 * a);
    }
    uint32_t const next_len = g.ref_len[r + 1] < 255 ? g.ref_len[r + 1] : 255;
    uint2_t const ti = !snp ? uint2_t{HINT_TAIL_NODE, r} : uint2_t{HINT_TAIL_NODE | HINT_TAIL_OK | (nv << HINT_TAIL_NALL_SHIFT) | (next_len << HINT_TAIL_NEXT_SHIFT) | (codes << HINT_TAIL_CODES_SHIFT), r};
    uint32_t const at = g.ref_order[r] - first;
    for (uint32_t d = 0; d < g.ref_len[r]; ++d)
      t.tail_info[at + d] = ti;
  }
  t.refp.assign(4 * (static_cast<std::size_t>(n) / 32 + 8), 0); // (padded: the kernel loads 6 groups from any position without a bounds test)
  for (uint32_t i = 0; i < n; ++i)
    for (uint32_t b = 0; b < 4; ++b)
      t.refp[4 * (i >> 5) + b] |= ((static_cast<uint32_t>(t.base[i]) >> b) & 1u) << (i & 31u);
}

// the allele windows of a graph (IndexView::win): one per alternative allele of the sites hint_site_wants_windows() names, in
// site and allele order, as many as `cap` admits (GTX_HINT_WINDOWS, default 16 384: 6.3 M table positions)
void hint_list_windows(HostGraph const & g, std::vector<HintWindow> & win, std::vector<uint32_t> & site_win)
{
  win.clear();
  site_win.assign(g.ref_order.size(), 0u);
  char const * e = std::getenv("GTX_HINT_WINDOWS");
  uint64_t const cap = e ? static_cast<uint64_t>(std::max(0l, std::atol(e))) : 16384u;
  GraphView const gv = g.view();
  uint32_t const R = static_cast<uint32_t>(g.ref_order.size());
  if (R == 0 || R - 1 >= HINT_NO_SITE)
    return;
  for (uint32_t r = 0; r + 1 < R; ++r)
  {
    if (!hint_site_wants_windows(gv, r))
      continue;
    uint32_t const first = static_cast<uint32_t>(win.size());
    uint32_t count = 0;
    for (uint32_t a = 1; a < g.ref_nvar[r]; ++a)
      if (hint_allele_gets_window(gv, r, a) && win.size() < cap)
      {
        win.push_back(HintWindow{r, a, g.var_len[g.ref_first_var[r] + a], g.var_len[g.ref_first_var[r]], g.ref_order[r] + g.ref_len[r], {0, 0, 0}});
        ++count;
      }
    site_win[r] = count ? first | (count << 24) : 0u;
  }
}

// Tables of the position-hinted pass (IndexView::refp ..., hinted.hpp).  For every reference position: what a read
// k-mer that equals -- or is one substitution / one ambiguous base away from -- the reference 32-mer of that place
// would get from the global lookups, decided here once from the finished index.
static void build_hints(HostGraph const & g, HostIndex & out)
{
  HintGraphTables gt;
  hint_graph_tables(g, gt);
  out.hint_first = gt.hint_first;
  out.n_hint = gt.n;
  out.filt_log2 = 0;
  if (gt.n == 0)
  {
    out.refp.assign(32, 0);
    out.pos_flags.assign(1, uint2_t{0, 0});
    out.tail_info.assign(1, uint2_t{0, 0});
    out.filt[0].assign(1, 0);
    out.filt[1].assign(1, 0);
    return;
  }
  uint32_t const n = gt.n;
  // the allele windows continue the per-position arrays behind win_base
  hint_list_windows(g, out.win, out.site_win);
  uint32_t const n_win = static_cast<uint32_t>(out.win.size());
  out.win_base = hint_win_base(n);
  uint64_t const total = hint_total_positions(n, n_win);
  GraphView const gv0 = g.view();
  if (n_win)
  {
    std::vector<uint8_t> mbase = gt.base, mroom = gt.room, mback = gt.back;
    std::vector<uint2_t> mtail = gt.tail_info;
    gt.base.assign(total, 15);
    gt.room.assign(total, 0);
    gt.back.assign(total, 0);
    gt.tail_info.assign(total, uint2_t{0, 0});
    std::copy(mbase.begin(), mbase.end(), gt.base.begin());
    std::copy(mroom.begin(), mroom.end(), gt.room.begin());
    std::copy(mback.begin(), mback.end(), gt.back.begin());
    std::copy(mtail.begin(), mtail.end(), gt.tail_info.begin());
    parallel_slices(n_win, host_threads(), [&](unsigned, std::size_t b, std::size_t e) {
      for (std::size_t w = b; w < e; ++w)
        for (uint32_t local = 0; local < HINT_WIN_STRIDE; ++local)
        {
          std::size_t const p = out.win_base + w * HINT_WIN_STRIDE + local;
          hint_window_cell(gv0, out.win[w], local, mbase.data(), mroom.data(), mback.data(), mtail.data(), n, gt.base[p], gt.room[p], gt.back[p], gt.tail_info[p]);
        }
    });
    // (planes: the linear reference as before, nothing up to win_base, every window position its base -- 15 where it has none)
    gt.refp.assign(4 * (total / 32 + 8), 0);
    uint64_t const win_end = out.win_base + static_cast<uint64_t>(n_win) * HINT_WIN_STRIDE;
    for (uint64_t i = 0; i < win_end; ++i)
      if (i < n || i >= out.win_base)
        for (uint32_t b = 0; b < 4; ++b)
          gt.refp[4 * (i >> 5) + b] |= ((static_cast<uint32_t>(gt.base[i]) >> b) & 1u) << (i & 31u);
  }
  std::vector<uint8_t> const & base = gt.base;
  std::vector<uint8_t> const & room = gt.room;
  std::vector<uint8_t> const & back = gt.back;
  out.refp = std::move(gt.refp);
  out.tail_info = std::move(gt.tail_info);
  // per key: the keys that share its first / last 16 bases (groups), then the neighbour verdict (index_build.hpp)
  std::size_t const nk = out.keys.size();
  std::vector<uint32_t> lbegin(nk), lsize(nk), rorder(nk), rbegin(nk), rsize(nk), nb(nk, 0);
  std::vector<uint8_t> nb_same(nk, 1);
  for (std::size_t k = 0; k < nk;) // keys ascending: equal first 16 bases are neighbours in the array
  {
    std::size_t e = k + 1;
    while (e < nk && (out.keys[e] >> 32) == (out.keys[k] >> 32))
      ++e;
    for (std::size_t m = k; m < e; ++m)
    {
      lbegin[m] = static_cast<uint32_t>(k);
      lsize[m] = static_cast<uint32_t>(e - k);
    }
    k = e;
  }
  {
    std::vector<std::pair<uint64_t, uint32_t>> order(nk);
    for (std::size_t k = 0; k < nk; ++k)
      order[k] = {out.keys[k] & 0xFFFFFFFFull, static_cast<uint32_t>(k)};
    radix_sort_pairs(order, 32);
    for (std::size_t k = 0; k < nk;)
    {
      std::size_t e = k + 1;
      while (e < nk && order[e].first == order[k].first)
        ++e;
      for (std::size_t m = k; m < e; ++m)
      {
        rorder[m] = order[m].second;
        rbegin[order[m].second] = static_cast<uint32_t>(k);
        rsize[order[m].second] = static_cast<uint32_t>(e - k);
      }
      k = e;
    }
  }
  HintKeys const t{out.keys.data(), out.key_off.data(), out.dev_labels.data(), static_cast<uint32_t>(nk), lbegin.data(), lsize.data(),
                   rorder.data(), rbegin.data(), rsize.data()};
  parallel_slices(nk, host_threads(), [&](unsigned, std::size_t b, std::size_t e) {
    for (std::size_t k = b; k < e; ++k)
    {
      uint32_t same = 1, known = 0;
      hint_judge_key(t, static_cast<uint32_t>(k), nb[k], same, known);
      nb_same[k] = static_cast<uint8_t>(same | (known << 1)); // (bit 1: SLOT_NB_KNOWN of the key's slot, set below)
    }
  });
  // the exact table's slots of the keys whose neighbours are known (gtx_flat.hpp: SLOT_NB_KNOWN; GTX_NB_KNOWN=0: a test
  // switch that leaves the bit clear everywhere, so that the kernels verify the neighbours themselves)
  char const * nbk = std::getenv("GTX_NB_KNOWN");
  if (!out.slots.empty() && !(nbk && nbk[0] == '0'))
    parallel_slices(nk, host_threads(), [&](unsigned, std::size_t b, std::size_t e) {
      uint64_t const mask = (1ull << out.log2_cap) - 1;
      for (std::size_t k = b; k < e; ++k)
        if ((nb_same[k] & 2u) != 0 && nb[k] != 0)
        {
          uint64_t const key = plane_key(out.keys[k]);
          bool done = false;
          for (uint64_t bk = hash_key(key, out.log2_cap); !done; bk = (bk + 1) & mask)
            for (uint32_t j = 0; j < BUCKET_SLOTS && !done; ++j)
            {
              IndexSlot & sl = out.slots[bk * BUCKET_SLOTS + j];
              if (sl.cnt == 0)
                done = true; // (cannot happen: every key is in the table)
              else if (sl.key == key)
              {
                sl.off |= SLOT_NB_KNOWN;
                done = true;
              }
            }
        }
    });
  // filters over the halves of every indexed key (nibble form, as the kernel hashes them): 32 bits per key and side
  uint32_t const fl = hint_filter_log2_words(nk);
  out.filt_log2 = fl;
  out.filt[0].assign(1ull << fl, 0);
  out.filt[1].assign(1ull << fl, 0);
  for (std::size_t k = 0; k < nk; ++k)
    for (uint32_t side = 0; side < 2; ++side)
    {
      uint32_t w0, w1, word, mask;
      hint_half_planes(static_cast<uint32_t>(side == 0 ? out.keys[k] >> 32 : out.keys[k]), w0, w1);
      hint_filter_slot(w0, w1, fl, word, mask);
      out.filt[side][word] |= mask;
    }
  // per position (the flags look at the filters: HINT_NEAR_FREE)
  HintKeys tf = t;
  tf.filt0 = out.filt[0].data();
  tf.filt1 = out.filt[1].data();
  tf.filt_log2 = fl;
  GraphView const gv = g.view();
  out.pos_flags.assign(total, uint2_t{0, 0}); // (every position of the linear reference and of the windows is written below; the padding stays 0)
  parallel_slices(n, host_threads(), [&](unsigned, std::size_t b, std::size_t e) {
    for (std::size_t p = b; p < e; ++p)
      out.pos_flags[p] = hint_position_flags(gv, tf, nb.data(), nb_same.data(), base.data(), room.data(), back.data(), n, static_cast<uint32_t>(p));
  });
  parallel_slices(n_win, host_threads(), [&](unsigned, std::size_t b, std::size_t e) {
    for (std::size_t w = b; w < e; ++w)
      for (uint32_t local = 0; local < HINT_WIN_STRIDE; ++local)
      {
        uint32_t const p = out.win_base + static_cast<uint32_t>(w) * HINT_WIN_STRIDE + local;
        out.pos_flags[p] = hint_window_flags(gv, tf, nb.data(), nb_same.data(), base.data(), room.data(), back.data(), static_cast<uint32_t>(total),
                                             out.pos_flags.data(), n, out.win[w], local, p);
      }
  });
}

void build_index(HostGraph const & g, HostIndex & out)
{
  std::vector<Emit> em;
  enumerate_kmers(g, em);
  build_tables_host(g, em, out);
}

// index_graph's sweep (indexer.cpp:246-291): every end position is independent, so the reference nodes (with the site
// behind each) are cut into contiguous ranges for the host team; concatenated in order the ranges give the sweep's order.
void enumerate_kmers(HostGraph const & g, std::vector<Emit> & em) { enumerate_kmers(g, em, nullptr); }

// With `runs`: the k-mers that lie inside ONE reference node of plain A/C/G/T -- one per position, all but the first 31
// of a node -- are not listed but left to the device as runs (node, count); the list keeps everything that walks through a
// site.  runs[i].host_before / dev_before place both kinds in the sweep's order (gtx_index_dev.hip, k_emit_runs).
void enumerate_kmers(HostGraph const & g, std::vector<Emit> & em, std::vector<EmitRun> * runs)
{
  uint32_t const R = static_cast<uint32_t>(g.ref_order.size());
  // (the one stage of a device context's build that is still on the host; its slices are long and independent, so it takes
  // a larger team than the other stages: 8 / 16 / 32 / 64 threads = 62 / 42 / 28 / 22 ms on the merged-cluster graph)
  // (with runs the host's share is the positions within 31 bases behind a site and the sites' own: a small team does unless
  //  the graph is mostly sites)
  auto weight = [&](uint32_t r) -> uint64_t
  {
    if (!runs)
      return g.ref_len[r] + 8;
    uint64_t w = std::min<uint32_t>(g.ref_len[r], K - 1) + g.ref_len[r] / 64 + 1; // (+ the scan for other characters)
    if (r + 1 < R)
      for (uint32_t a = 0; a < g.ref_nvar[r]; ++a)
        w += g.var_len[g.ref_first_var[r] + a];
    return w;
  };
  uint64_t total = 0;
  for (uint32_t r = 0; r < R; ++r)
    total += weight(r);
  unsigned T = R < 64 ? 1u : host_threads();
  if (T > 1 && !std::getenv("GTX_HOST_THREADS"))
  {
    if (runs && total < 400000)
      T = std::min(T, 4u);
    else
      T = std::max(T, std::min(std::thread::hardware_concurrency(), 64u));
  }
  NodePacks const packs(g);
  std::vector<std::vector<Emit>> part(T);
  std::vector<std::vector<EmitRun>> part_runs(T);
  // ranges of equal weight
  std::vector<uint32_t> cut(T + 1, R);
  {
    uint64_t run = 0;
    unsigned t = 1;
    cut[0] = 0;
    for (uint32_t r = 0; r < R && t < T; ++r)
    {
      run += weight(r);
      if (run * T >= total * t)
        cut[t++] = r + 1;
    }
  }
  auto work = [&](unsigned t)
  {
    std::vector<Emit> & out = part[t];
    uint32_t const r0 = cut[t], r1 = cut[t + 1];
    if (r0 >= r1)
      return;
    uint64_t bases = 0;
    for (uint32_t r = r0; r < r1; ++r)
      bases += runs ? weight(r) : g.ref_len[r];
    out.reserve(bases + bases / 4 + 64);
    Walker w{g, out, !g.event_off.empty(), packs};
    for (uint32_t r = r0; r < r1; ++r)
    {
      // fast path inside a reference node: a rolling 2-bit window while the k-mer stays within this node
      char const * dna = g.dna.data() + g.ref_dna[r];
      uint32_t len = g.ref_len[r];
      if (runs && len >= K)
      {
        // (A/C/G/T only?  On the comparison codes -- 1 2 4 8 for those four -- eight bases at a time: no high nibble, no zero
        //  byte, eight bits in all)
        bool plain = true;
        char const * cd = g.codes.data() + g.ref_dna[r];
        uint32_t d = 0;
        for (; d + 8 <= len && plain; d += 8)
        {
          uint64_t w;
          std::memcpy(&w, cd + d, 8);
          plain = (w & 0xF0F0F0F0F0F0F0F0ull) == 0 && ((w - 0x0101010101010101ull) & ~w & 0x8080808080808080ull) == 0 &&
                  __builtin_popcountll(w) == 8;
        }
        for (; d < len && plain; ++d)
          plain = Walker::code(dna[d]) >= 0;
        if (plain)
        {
          len = K - 1; // the positions whose window reaches back past the node's start, here; the others are the device's
          part_runs[t].push_back(EmitRun{r, static_cast<uint32_t>(out.size()), 0u, g.ref_len[r] - (K - 1)});
        }
      }
      uint64_t roll = 0;
      uint32_t valid = 0;
      // (a run's place in the list: behind the first 31 positions' k-mers -- set below)
      EmitRun * const my_run = runs && len != g.ref_len[r] ? &part_runs[t].back() : nullptr;
      for (uint32_t d = 0; d < len; ++d)
      {
        int const c = Walker::code(dna[d]);
        if (c < 0)
        {
          valid = 0;
          continue;
        }
        roll = (roll << 2) | static_cast<uint64_t>(c);
        ++valid;
        if (valid >= K)
          out.push_back({roll, {g.ref_order[r] + d - (K - 1), g.ref_order[r] + d, INVALID}});
        else if (valid == d + 1) // window reaches back past the node start: enumerate through the previous site(s)
        {
          w.n_vars = 0;
          w.end_label = g.ref_order[r] + d;
          w.back_ref(r, d, 0, 0);
        }
      }
      if (my_run)
        my_run->host_before = static_cast<uint32_t>(out.size());
      if (r + 1 == R || g.ref_nvar[r] == 0)
        continue;
      uint32_t const fv = g.ref_first_var[r];
      for (uint32_t a = 0; a < g.ref_nvar[r]; ++a)
      {
        uint32_t const v = fv + a;
        for (uint32_t d = 0; d < g.var_len[v]; ++d)
        {
          w.n_vars = 0;
          w.end_label = w.special_of(r, g.var_order[v] + d);
          w.back_var(v, d, 0, 0);
        }
      }
    }
  };
  if (T == 1)
    work(0);
  else
  {
    std::vector<std::thread> team;
    for (unsigned t = 0; t < T; ++t)
      team.emplace_back(work, t);
    for (auto & th : team)
      th.join();
  }
  std::size_t listed = 0;
  for (auto const & p : part)
    listed += p.size();
  em.clear();
  em.reserve(listed);
  for (auto const & p : part)
    em.insert(em.end(), p.begin(), p.end());
  if (runs)
  {
    runs->clear();
    uint32_t host_base = 0, dev = 0;
    for (unsigned t = 0; t < T; ++t)
    {
      for (EmitRun e : part_runs[t])
      {
        e.host_before += host_base;
        e.dev_before = dev;
        dev += e.count;
        runs->push_back(e);
      }
      host_base += static_cast<uint32_t>(part[t].size());
    }
  }
}

void build_tables_host(HostGraph const & g, std::vector<Emit> const & em, HostIndex & out)
{
  bool const timing = std::getenv("GTX_TIMING") != nullptr; // stage times on stderr
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](char const * what)
  {
    auto const now = std::chrono::steady_clock::now();
    if (timing)
      std::fprintf(stderr, "[gtx] build_index %-28s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  out = HostIndex();
  // group by key keeping emission order inside a key
  std::vector<std::pair<uint64_t, uint32_t>> perm(em.size());
  for (std::size_t i = 0; i < em.size(); ++i)
    perm[i] = {em[i].key, static_cast<uint32_t>(i)};
  radix_sort_pairs(perm, 64);
  out.labels.reserve(em.size());
  for (std::size_t i = 0; i < perm.size(); ++i)
  {
    Emit const & e = em[perm[i].second];
    if (i == 0 || e.key != out.keys.back())
    {
      out.keys.push_back(e.key);
      out.key_off.push_back(static_cast<uint32_t>(out.labels.size()));
    }
    out.labels.push_back(e.label);
  }
  out.key_off.push_back(static_cast<uint32_t>(out.labels.size()));
  lap("group by key");
  out.dev_labels.resize(out.labels.size());
  parallel_slices(out.labels.size(), host_threads(), [&](unsigned, std::size_t b, std::size_t e) {
    for (std::size_t i = b; i < e; ++i)
    {
      gtx_label const & l = out.labels[i];
      DevLabel d{l.start_index, l.end_index, INVALID, 0};
      if (l.variant_id != INVALID)
      {
        d.site = g.var_out_ref[l.variant_id] - 1;
        d.allele = l.variant_id - g.ref_first_var[d.site];
      }
      out.dev_labels[i] = d;
    }
  });
  // device form: (plane key, label offset, label count) of every key, then the exact table (own thread) and the two
  // half-key tables
  std::vector<HalfEntry> all(out.keys.size());
  parallel_slices(out.keys.size(), host_threads(), [&](unsigned, std::size_t b, std::size_t e) {
    for (std::size_t k = b; k < e; ++k)
      all[k] = HalfEntry{plane_key(out.keys[k]), out.key_off[k], out.key_off[k + 1] - out.key_off[k]};
  });
  lap("plane keys");
  uint32_t log2_cap = 2;
  while ((static_cast<uint64_t>(BUCKET_SLOTS) << log2_cap) < 2 * out.keys.size() + 1)
    ++log2_cap;
  out.log2_cap = log2_cap;
  unsigned const T = host_threads();
  {
    out.slots.assign(static_cast<uint64_t>(BUCKET_SLOTS) << log2_cap, IndexSlot{0, 0, 0, {0, 0, 0, 0}});
    std::vector<IndexSlot> items(all.size());
    parallel_slices(all.size(), T, [&](unsigned, std::size_t b, std::size_t e) {
      for (std::size_t k = b; k < e; ++k)
      {
        items[k] = IndexSlot{all[k].key, all[k].off, all[k].cnt, {0, 0, 0, 0}};
        if (all[k].cnt == 1) // inline copy of the one label
        {
          DevLabel const & d = out.dev_labels[all[k].off];
          items[k].p[0] = d.start;
          items[k].p[1] = d.end;
          items[k].p[2] = d.site;
          items[k].p[3] = d.allele;
        }
      }
    });
    bucket_insert_all(out.slots, log2_cap, items);
  }
  lap("exact table");
  // half-key buckets (plane-form keys: the 16 first bases are bits 0..15 of both words, the 16 last bases bits 16..31)
  {
    size_t const n = out.keys.size();
    auto half_of = [](uint64_t pk, int side) -> uint64_t
    {
      uint32_t const lo = static_cast<uint32_t>(pk), hi = static_cast<uint32_t>(pk >> 32);
      return side == 0 ? ((lo & 0xFFFFu) | ((hi & 0xFFFFu) << 16)) : ((lo >> 16) | (hi & 0xFFFF0000u));
    };
    out.hlist.resize(2 * n);
    uint32_t hl = 2;
    while ((static_cast<uint64_t>(BUCKET_SLOTS) << hl) < 4 * n + 1)
      ++hl;
    out.h_log2_cap = hl;
    out.hslots.assign(static_cast<uint64_t>(BUCKET_SLOTS) << hl, IndexSlot{0, 0, 0, {0, 0, 0, 0}});
    std::vector<IndexSlot> side_items[2];
    auto build_side = [&](int side)
    {
      // group the keys by this half (order of the groups and inside a group does not matter to the kernel, which
      // orders its candidates by neighbour number): out.keys ascending is already grouped by the first 16 bases; for
      // the last 16 bases a stable radix sort on the low 32 bits of the 2-bit key
      std::vector<std::pair<uint64_t, uint32_t>> order(n);
      parallel_slices(n, T, [&](unsigned, std::size_t b, std::size_t e) {
        for (size_t k = b; k < e; ++k)
          order[k] = {out.keys[k], static_cast<uint32_t>(k)};
      });
      if (side == 1)
        radix_sort_pairs(order, 32);
      size_t const base = side * n;
      parallel_slices(n, T, [&](unsigned, std::size_t b, std::size_t e) {
        for (size_t k = b; k < e; ++k)
          out.hlist[base + k] = all[order[k].second];
      });
      std::vector<IndexSlot> & items = side_items[side];
      items.reserve(n);
      size_t k = 0;
      while (k < n)
      {
        uint64_t const half = half_of(out.hlist[base + k].key, side);
        size_t e = k + 1;
        while (e < n && half_of(out.hlist[base + e].key, side) == half)
          ++e;
        IndexSlot s{half | (static_cast<uint64_t>(side) << 32), static_cast<uint32_t>(base + k), static_cast<uint32_t>(e - k),
                    {0, 0, 0, 0}};
        if (e - k == 1) // inline copy of the one bucket entry
        {
          HalfEntry const & he = out.hlist[base + k];
          s.p[0] = static_cast<uint32_t>(he.key);
          s.p[1] = static_cast<uint32_t>(he.key >> 32);
          s.p[2] = he.off;
          s.p[3] = he.cnt;
        }
        items.push_back(s);
        k = e;
      }
    };
    build_side(0);
    build_side(1);
    std::vector<IndexSlot> half_items(std::move(side_items[0]));
    half_items.insert(half_items.end(), side_items[1].begin(), side_items[1].end());
    lap("  half lists");
    bucket_insert_all(out.hslots, hl, half_items);
  }
  lap("half-key tables");
  build_hints(g, out);
  lap("position hints");
}

} // namespace gtx
