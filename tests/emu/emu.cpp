// emu.cpp -- TEST HARNESS ONLY: runs the kernel sources (graphtyper_amd/csrc/align_core.hpp, score_core.hpp) on the
// host, with a sequential stand-in for the wavefront (lane lambdas are looped over the 64 lanes).  It exists so that kernel logic
// can be debugged in a container without a GPU; it is never built into, linked against or loaded by libgtx.so, and
// nothing outside tests/ uses it.  Parity claims are made by the `-m gpu` tests through the C ABI only.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../graphtyper_amd/csrc/gtx_flat.hpp"
#include "../../graphtyper_amd/csrc/align_core.hpp"
#include "../../graphtyper_amd/csrc/score_core.hpp"
#include "../../graphtyper_amd/csrc/score_replay.hpp"

namespace
{
// Sequential stand-in for one wavefront: lane lambdas run for lane 0..63 one after the other, per-lane values are
// arrays of 64, the wave-uniform parts of the kernel source run once.
uint64_t g_notes[16];  // why a task left the express pass (W::note), test diagnostics only
uint64_t g_hnotes[16]; // ... and the position-hinted pass (hint_note)
uint32_t g_last_hnote;  // the last non-zero note of the current hinted_one call: why THIS read was declined

} // namespace
namespace gtx
{
void hint_note(uint32_t k)
{
  ++g_hnotes[k & 15u];
  if (k)
    g_last_hnote = k;
}
} // namespace gtx
namespace
{
struct WaveEmu
{
  static void note(uint32_t k) { ++g_notes[k & 15u]; }
  template <class T>
  struct PerLane
  {
    T v[64];
    T & operator[](uint32_t l) { return v[l]; }
    T const & operator[](uint32_t l) const { return v[l]; }
  };
  static uint32_t from_lane(PerLane<uint32_t> const & p, uint32_t lane) { return p.v[lane]; }
  template <class F>
  static void lanes(F && f)
  {
    for (uint32_t l = 0; l < 64; ++l)
      f(l);
  }
  static bool leader() { return true; }
  template <class T>
  static T uni(T v)
  {
    return v;
  }
  static void lds_sync() {}
  static uint64_t ballot(PerLane<bool> const & p)
  {
    uint64_t m = 0;
    for (uint32_t l = 0; l < 64; ++l)
      m |= static_cast<uint64_t>(p.v[l] ? 1u : 0u) << l;
    return m;
  }
  static uint32_t sum(PerLane<uint32_t> const & p)
  {
    uint32_t s = 0;
    for (uint32_t l = 0; l < 64; ++l)
      s += p.v[l];
    return s;
  }
  static void excl_scan(PerLane<uint32_t> const & in, PerLane<uint32_t> & out, uint32_t & total)
  {
    uint32_t s = 0;
    for (uint32_t l = 0; l < 64; ++l)
    {
      out.v[l] = s;
      s += in.v[l];
    }
    total = s;
  }
  static uint32_t max(PerLane<uint32_t> const & p)
  {
    uint32_t m = 0;
    for (uint32_t l = 0; l < 64; ++l)
      m = p.v[l] > m ? p.v[l] : m;
    return m;
  }
  static void atomic_or_u64(uint64_t * p, uint64_t v) { *p |= v; }
  static unsigned long long clock() { return 0; }
  static void atomic_add_u32(uint32_t * p, uint32_t v) { __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
  static uint32_t atomic_claim_u32(uint32_t * p) { return __atomic_fetch_add(p, 1u, __ATOMIC_RELAXED); }
  static void atomic_add_u64(unsigned long long * p, unsigned long long v) { __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
};

struct Emu
{
  gtx_params params{};
  gtx::HostGraph graph;
  gtx::HostIndex index;
  std::vector<uint32_t> arena; // big-record arena (see gtx_align_batch)
  uint64_t arena_used = 0;
  uint64_t second_pass_tasks = 0; // tasks that reached the last pass (HBM tables)
  uint64_t wide_pass_tasks = 0;   // tasks that went on to the pass with wide allele sets
  uint64_t exact_pass_tasks[4] = {0, 0, 0, 0}; // tasks that reached the exact pass (a small part of the slab / a large part / the whole slab / still refused)
  std::vector<uint8_t> exact_slab;
  uint64_t general_tasks = 0;     // tasks pass 1 handed to pass 2
  uint64_t hinted_done = 0;       // forward tasks the position-hinted pass finished
  // diagnostics of the last emu_align (tools/decline_notes.py): per read the pass that finished its forward task (0 hinted,
  // 1 express, 2 general, 3 HBM tables) and the note pass 0 left when it declined the read (0: none / no note)
  std::vector<uint8_t> pass_of, hint_decline;
};

} // namespace

extern "C"
{
  void * emu_new(const gtx_graph_view * g, const gtx_params * p, char * err, int err_cap)
  {
    auto e = std::make_unique<Emu>();
    e->params = *p;
    std::string const msg = gtx::flatten_graph(*g, *p, e->graph);
    if (!msg.empty())
    {
      std::snprintf(err, err_cap, "%s", msg.c_str());
      return nullptr;
    }
    gtx::build_index(e->graph, e->index);
    return e.release();
  }

  void emu_free(void * p) { delete static_cast<Emu *>(p); }

  // the two statements of to_uint64_vec for a k-mer of 32 codes: the sequential one (expand_keys, the reference's loop) and the
  // key-per-lane one the general pass runs (expand_keys_lanes).  keys_*: room for 4 * 97 keys each; returns 1 when count and
  // keys agree (the counts are returned through n_seq / n_lanes: 0 = gave up, 0xFFFFFFFF = beyond the pass' key buffer)
  int emu_expand_keys(const uint8_t * codes32, uint64_t * keys_seq, uint64_t * keys_lanes, uint32_t * n_seq, uint32_t * n_lanes)
  {
    auto ws = std::make_unique<gtx::AlignWorkspace>();
    uint32_t amb = 0;
    uint64_t base = 0;
    for (uint32_t t = 0; t < 32; ++t)
    {
      uint32_t const c = codes32[t] & 15u;
      ws->rd[t] = static_cast<uint8_t>(c);
      bool const single = (c & (c - 1u)) == 0u && c != 0u;
      if (!single)
        amb |= 1u << t;
      uint64_t const two = c == 2u ? 1u : c == 4u ? 2u : c == 8u ? 3u : 0u;
      base |= ((two & 1u) << t) | ((two >> 1) << (32u + t));
    }
    *n_seq = gtx::expand_keys(ws->rd, 0, keys_seq, gtx::AlignCfg::KEY_CAP);
    *n_lanes = gtx::expand_keys_lanes<WaveEmu>(*ws, 0, amb, base);
    uint32_t const n = *n_lanes == 0xFFFFFFFFu ? 0u : *n_lanes;
    std::memcpy(keys_lanes, ws->u.keybuf, n * sizeof(uint64_t));
    if (*n_seq != *n_lanes)
      return 0;
    return std::memcmp(keys_seq, keys_lanes, n * sizeof(uint64_t)) == 0 ? 1 : 0;
  }

  // milliseconds of one listing of the sweep (tools: where a context's host time goes)
  double emu_time_listing(void * p, int with_runs)
  {
    gtx::HostGraph const & g = static_cast<Emu *>(p)->graph;
    std::vector<gtx::Emit> em;
    std::vector<gtx::EmitRun> runs;
    auto const t0 = std::chrono::steady_clock::now();
    gtx::enumerate_kmers(g, em, with_runs ? &runs : nullptr);
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }

  // The sweep listed in two ways (gtx_host.cpp: enumerate_kmers): every k-mer by the host, and with the in-node runs left to
  // the device -- expanded here the way k_emit_runs / k_place_listed (gtx_index_dev.hip) do it.  Returns the number of
  // k-mers, or -(1 + index of the first difference); n_listed / n_runs: what the host listed in the second form.
  long emu_enumeration_check(void * p, uint64_t * n_listed, uint64_t * n_runs)
  {
    gtx::HostGraph const & g = static_cast<Emu *>(p)->graph;
    std::vector<gtx::Emit> full, listed;
    std::vector<gtx::EmitRun> runs;
    gtx::enumerate_kmers(g, full);
    gtx::enumerate_kmers(g, listed, &runs);
    *n_listed = listed.size();
    *n_runs = runs.size();
    uint64_t const n_run = runs.empty() ? 0 : static_cast<uint64_t>(runs.back().dev_before) + runs.back().count;
    std::vector<gtx::Emit> merged(listed.size() + n_run);
    std::vector<uint8_t> written(merged.size(), 0);
    size_t u = 0;
    for (uint64_t t = 0; t < n_run; ++t)
    {
      while (u + 1 < runs.size() && runs[u + 1].dev_before <= t)
        ++u;
      gtx::EmitRun const & r = runs[u];
      uint32_t const k = static_cast<uint32_t>(t) - r.dev_before;
      uint64_t key = 0;
      for (uint32_t i = 0; i < gtx::K; ++i)
      {
        int const code = g.codes[g.ref_dna[r.node] + k + i]; // (1 2 4 8 = A C G T)
        key = (key << 2) | static_cast<uint64_t>(__builtin_ctz(static_cast<unsigned>(code)));
      }
      size_t const at = r.host_before + t;
      if (at >= merged.size() || written[at])
        return -1;
      merged[at] = gtx::Emit{key, gtx_label{g.ref_order[r.node] + k, g.ref_order[r.node] + k + (gtx::K - 1), gtx::INVALID}};
      written[at] = 1;
    }
    size_t lo = 0;
    for (size_t j = 0; j < listed.size(); ++j)
    {
      while (lo < runs.size() && runs[lo].host_before <= j)
        ++lo;
      size_t const at = j + (lo ? runs[lo - 1].dev_before + runs[lo - 1].count : 0u);
      if (at >= merged.size() || written[at])
        return -1;
      merged[at] = listed[j];
      written[at] = 1;
    }
    if (merged.size() != full.size())
      return -1;
    for (size_t i = 0; i < full.size(); ++i)
      if (merged[i].key != full[i].key || merged[i].label.start_index != full[i].label.start_index ||
          merged[i].label.end_index != full[i].label.end_index || merged[i].label.variant_id != full[i].label.variant_id)
        return -static_cast<long>(i) - 1;
    return static_cast<long>(full.size());
  }

  // same contract as gtx_align_batch, host pointers: BAM nibble rows, repacked into plane rows first (what the library does
  // with them on the device); the kernel sources below read plane rows only
  int emu_align(void * p, const uint8_t * nibble_rows, uint32_t nibble_stride, const gtx_read_meta * meta, uint32_t n_reads,
                uint32_t * records, uint32_t rec_words)
  {
    using namespace gtx;
    Emu & e = *static_cast<Emu *>(p);
    uint32_t const seq_stride = (nibble_stride + PLANE_GROUP_BYTES - 1u) / PLANE_GROUP_BYTES * PLANE_GROUP_BYTES;
    std::vector<uint32_t> plane_rows(static_cast<size_t>(n_reads) * (seq_stride / 4) + 4);
    for (uint32_t r = 0; r < n_reads; ++r)
      planes_from_nibbles(nibble_rows + static_cast<uint64_t>(r) * nibble_stride, nibble_stride, plane_rows.data() + static_cast<size_t>(r) * (seq_stride / 4),
                          seq_stride / PLANE_GROUP_BYTES);
    uint8_t const * seq = reinterpret_cast<uint8_t const *>(plane_rows.data());
    GraphView const g = e.graph.view();
    IndexView ix = e.index.view(static_cast<uint32_t>(e.params.max_index_labels), HALF_BUCKET_CAP);
    if (char const * cap = std::getenv("GTX_HALF_BUCKET_CAP"))
      ix.half_bucket_cap = static_cast<uint32_t>(std::atol(cap));
    auto ws = std::make_unique<AlignWorkspace>();
    auto big_ws = std::make_unique<big::AlignWorkspace>();
    std::vector<uint32_t> big_keys(2 * big::AlignCfg::MAXPP, 0xABABABABu), wide_keys(2 * wide::AlignCfg::MAXPP, 0xABABABABu);
    std::vector<uint64_t> big_bits(big::AlignCfg::MAXPP / 64 + 1, 0xABABABABABABABABull), wide_bits(wide::AlignCfg::MAXPP / 64 + 1, 0xABABABABABABABABull);
    bool const second_pass = !e.params.no_second_pass;
    char const * fe = std::getenv("GTX_FORCE_SECOND_PASS"); // 1: every task through all passes, 2: every task done by pass 2
    int const force = fe ? std::atoi(fe) : 0;
    bool const force_big = force == 1;
    auto seed_ws = std::make_unique<SeedWorkspace>();
    e.general_tasks = 0;
    char const * fl = std::getenv("GTX_EMU_FILL"); // what uninitialised workspace memory looks like
    int const fill = fl ? std::atoi(fl) : 0xAB;
    if (e.arena.empty()) // grows until emu_big_records_rewind, like the device arena
      e.arena.assign(e.params.big_record_words ? e.params.big_record_words : (1u << 20), 0xABABABABu);
    e.second_pass_tasks = 0;
    bool const force_both = e.params.force_align_both_orientations != 0;
    char const * e4 = std::getenv("GTX_EXPRESS4"); // 0: one read per wavefront in pass 1
    bool const four = !(e4 && e4[0] == '0');
    // lean / wide build of pass 1: forced by GTX_EXPRESS4=lean|wide, else by the graph's density (as gtx_align_batch does)
    bool const wide = e4 && e4[0] == 'w' ? true : e4 && e4[0] == 'l' ? false : express4_prefers_wide(e.graph, e.index);
    // ... and of pass 0 (GTX_HINT_BUILD=lean|dense; default: the dense build on the graphs that get the wide express pass)
    char const * hb = std::getenv("GTX_HINT_BUILD");
    bool const hint_dense = hb && hb[0] == 'd' ? true : hb && hb[0] == 'l' ? false : express4_prefers_wide(e.graph, e.index);
    auto e4_ws = std::make_unique<Express4Workspace<Express4Lean>>();
    auto e4_wide_ws = std::make_unique<Express4Workspace<Express4Wide>>();
    bool has_wide_sites = false;
    for (uint32_t n : e.graph.ref_nvar)
      has_wide_sites = has_wide_sites || n > 64;
    auto wide_ws = has_wide_sites ? std::make_unique<wide::AlignWorkspace>() : nullptr;
    e.wide_pass_tasks = 0;
    uint32_t widest_site = 0;
    for (uint32_t n : e.graph.ref_nvar)
      widest_site = std::max(widest_site, n);
    // the exact pass' slab (gtx_api.hip: exact_slab_mb; here one slab, used by one task at a time)
    constexpr uint32_t EXACT_PART_SITES = 24, EXACT_PART_CANDIDATES = 8256, EXACT_LARGE_PARTS = 32, EXACT_LARGE_SITES = 64; // (gtx_ctx.hpp: CallScratch)
    char const * xm = std::getenv("GTX_EXACT_PASS_MB");
    std::vector<uint8_t> & exact_slab = e.exact_slab;
    if (exact_slab.empty())
      exact_slab.resize(static_cast<size_t>(e.params.exact_pass_mb ? e.params.exact_pass_mb : xm && std::atol(xm) > 0 ? std::atol(xm) : (has_wide_sites ? 1024 : 512)) << 20);
    uint64_t exact_parts = std::min<uint64_t>(256u, std::max<uint64_t>(1u, (exact_slab.size() >> 20) / (has_wide_sites ? 32u : 2u))); // (the smallest part: as on the device with a full queue)
    if (char const * xp = std::getenv("GTX_EXACT_PARTS"))
      if (std::atol(xp) > 0)
        exact_parts = static_cast<uint64_t>(std::min<long>(std::atol(xp), 1024));
    e.exact_pass_tasks[0] = e.exact_pass_tasks[1] = e.exact_pass_tasks[2] = e.exact_pass_tasks[3] = 0;
    // one task through an HBM-table pass (the body of GTX_HBM_PASS_KERNEL in gtx_api.hip); returns the pass' status
    auto hbm_pass = [&](auto &, auto && align, auto && size_of, auto && write_body, uint32_t * rec, uint32_t len) -> uint32_t
    {
      uint32_t np = 0, longest = 0, ext = 0;
      uint32_t const raw = align(np, longest);
      uint32_t status = raw & ~GTX_ST_WIDE_ALLELE;
      uint32_t * body = rec + 2;
      uint64_t off = 0;
      if (status)
        np = 0;
      else
      {
        uint32_t const size = size_of(np);
        if (size > rec_words)
        {
          off = e.arena_used;
          if (off + (size - 2) > e.arena.size())
          {
            status = GTX_ST_RECORD_OVERFLOW;
            np = 0;
          }
          else
          {
            e.arena_used += size - 2;
            body = e.arena.data() + off;
            ext = GTX_ST_EXTERNAL;
          }
        }
      }
      uint32_t const has_var = write_body(np, body);
      rec[0] = np | ((status | ext) << 16);
      rec[1] = (np == 0 ? 0 : longest) | (len << 16) | (np == 0 ? 0u : has_var);
      if (ext)
        rec[2] = static_cast<uint32_t>(off);
      return raw;
    };
    // passes 2 and 3 for one task
    auto general = [&](uint32_t t)
    {
      uint32_t const read = t >> 1, orient = t & 1u;
      uint32_t * rec = records + static_cast<uint64_t>(t) * rec_words;
      uint32_t const len = meta[read].l_qseq;
      // pass 2 (gtx_align_kernel)
      ++e.general_tasks;
      std::memset(static_cast<void *>(ws.get()), fill, sizeof(AlignWorkspace));
      uint32_t const st = align_one<WaveEmu>(g, ix, *ws, seq + static_cast<uint64_t>(read) * seq_stride, len, orient == 1, rec, rec_words,
                                             /*try_fast=*/false);
      if (!second_pass || !(st || force_big))
        return;
      // second pass (gtx_align_big_kernel), and for a graph with a site of more than 64 alleles the pass behind it
      // (gtx_align_wide_kernel) for the tasks that met an allele number >= 64
      ++e.second_pass_tasks;
      std::memset(static_cast<void *>(big_ws.get()), fill, sizeof(big::AlignWorkspace));
      big_ws->pp_start = big_keys.data(); // (the kernel points them at LDS)
      big_ws->pp_end = big_keys.data() + big::AlignCfg::MAXPP;
      big_ws->bits_pp = big_bits.data();
      uint32_t const bst = hbm_pass(
        *big_ws, [&](uint32_t & np, uint32_t & longest)
        { return big::align_paths<WaveEmu>(g, ix, *big_ws, seq + static_cast<uint64_t>(read) * seq_stride, len, orient == 1, np, longest); },
        [&](uint32_t np) { return big::record_size<WaveEmu>(big::Here{}, *big_ws, np); },
        [&](uint32_t np, uint32_t * body) { return big::write_record_body<WaveEmu>(big::Here{}, *big_ws, np, body); }, rec, len);
      uint32_t last = bst;
      if (bst && has_wide_sites)
      {
        ++e.wide_pass_tasks;
        std::memset(static_cast<void *>(wide_ws.get()), fill, sizeof(wide::AlignWorkspace));
        wide_ws->pp_start = wide_keys.data();
        wide_ws->pp_end = wide_keys.data() + wide::AlignCfg::MAXPP;
        wide_ws->bits_pp = wide_bits.data();
        last = hbm_pass(
          *wide_ws, [&](uint32_t & np, uint32_t & longest)
          { return wide::align_paths<WaveEmu>(g, ix, *wide_ws, seq + static_cast<uint64_t>(read) * seq_stride, len, orient == 1, np, longest); },
          [&](uint32_t np) { return wide::record_size<WaveEmu>(wide::Here{}, *wide_ws, np); },
          [&](uint32_t np, uint32_t * body) { return wide::write_record_body<WaveEmu>(wide::Here{}, *wide_ws, np, body); }, rec, len);
      }
      // the exact pass (gtx_align_exact_kernel): first with a part of the slab, then with all of it
      constexpr uint32_t TABLES = GTX_ST_LABEL_OVERFLOW | GTX_ST_PATH_OVERFLOW | GTX_ST_DFS_OVERFLOW;
      bool const fixed_parts = std::getenv("GTX_EXACT_PARTS") != nullptr; // (the test switch: no large parts either)
      for (uint32_t level = 0; level < 3 && (last & TABLES); ++level)
      {
        ++e.exact_pass_tasks[level];
        uint64_t const bytes = level == 0 ? ((exact_slab.size() / exact_parts) & ~255ull) : (level == 1 && !fixed_parts) ? ((exact_slab.size() / EXACT_LARGE_PARTS) & ~255ull) : exact_slab.size();
        uint32_t const cap_v = level == 0 ? EXACT_PART_SITES : level == 1 ? EXACT_LARGE_SITES : exact::AlignCfg::MAXV;
        std::memset(exact_slab.data(), fill, 65536);
        if (has_wide_sites)
        {
          auto * xws = reinterpret_cast<exactw::AlignWorkspace *>(exact_slab.data());
          if (!exactw::exact_setup<WaveEmu>(xws, bytes, level == 0 ? std::min(exactw::exact_cand_cap(widest_site), EXACT_PART_CANDIDATES) : exactw::exact_cand_cap(widest_site), cap_v))
            continue;
          last = hbm_pass(
            *xws, [&](uint32_t & np, uint32_t & longest)
            { return exactw::align_paths<WaveEmu>(g, ix, *xws, seq + static_cast<uint64_t>(read) * seq_stride, len, orient == 1, np, longest); },
            [&](uint32_t np) { return exactw::record_size<WaveEmu>(exactw::Here{}, *xws, np); },
            [&](uint32_t np, uint32_t * body) { return exactw::write_record_body<WaveEmu>(exactw::Here{}, *xws, np, body); }, rec, len);
        }
        else
        {
          auto * xws = reinterpret_cast<exact::AlignWorkspace *>(exact_slab.data());
          if (!exact::exact_setup<WaveEmu>(xws, bytes, level == 0 ? std::min(exact::exact_cand_cap(widest_site), EXACT_PART_CANDIDATES) : exact::exact_cand_cap(widest_site), cap_v))
            continue;
          last = hbm_pass(
            *xws, [&](uint32_t & np, uint32_t & longest)
            { return exact::align_paths<WaveEmu>(g, ix, *xws, seq + static_cast<uint64_t>(read) * seq_stride, len, orient == 1, np, longest); },
            [&](uint32_t np) { return exact::record_size<WaveEmu>(exact::Here{}, *xws, np); },
            [&](uint32_t np, uint32_t * body) { return exact::write_record_body<WaveEmu>(exact::Here{}, *xws, np, body); }, rec, len);
        }
      }
      if (last & TABLES)
        ++e.exact_pass_tasks[3];
    };
    auto empty_record = [&](uint32_t t, uint32_t len)
    {
      uint32_t * rec = records + static_cast<uint64_t>(t) * rec_words;
      rec[0] = len > AlignCfg::MAX_READ ? (static_cast<uint32_t>(GTX_ST_RECORD_OVERFLOW) << 16) : 0u;
      rec[1] = len << 16;
    };
    char const * eh = std::getenv("GTX_HINT"); // 0: no position-hinted pass, d(ecline): the pass declines every task
    if (four && !(eh && eh[0] == '0'))
    {
      // pass 0, one read per lane from the position hint (gtx_align_hinted_kernel), then pass 1 over its queue
      // (gtx_align_express4q_kernel), as gtx_align_batch launches them
      std::vector<uint32_t> queue1, direct;
      e.hinted_done = 0;
      e.pass_of.assign(n_reads, 0);
      e.hint_decline.assign(n_reads, 0);
      for (uint32_t read = 0; read < n_reads; ++read)
      {
        g_last_hnote = 0;
        gtx_read_meta const m = meta[read];
        uint32_t const len = m.l_qseq;
        bool const outside = len < 2 * K - 1 || len > AlignCfg::MAX_READ;
        bool const rev = !outside && needs_reverse(m, force_both);
        if (!rev && (m.flag & GTX_FLAG_FORWARD_ONLY) == 0) // (the caller never looks at that record: gtx.h, GTX_FLAG_FORWARD_ONLY)
          empty_record(read * 2 + 1, len);
        if (outside)
          empty_record(read * 2, len);
        else
        {
          uint32_t const * row = reinterpret_cast<uint32_t const *>(seq + static_cast<uint64_t>(read) * seq_stride);
          uint32_t * slot = records + static_cast<uint64_t>(read) * 2 * rec_words;
          // (rows longer than 80 bytes: the eight-k-mer build, as gtx_align_batch chooses gtx_align_hinted_long_kernel)
          uint32_t const where = (force != 0 || (eh && eh[0] == 'd')) ? 0u
                                 : seq_stride > HintGeom<AlignCfg::KC>::ROW_BYTES ? hinted_long_one<8>(g, ix, row, seq_stride, m, slot, rec_words)
                                 : hint_dense                                      ? hinted_one<true>(g, ix, row, seq_stride, m, slot, rec_words)
                                                                                   : hinted_one<false>(g, ix, row, seq_stride, m, slot, rec_words);
          if (where == 0)
          {
            queue1.push_back(read);
            e.pass_of[read] = 1;
            e.hint_decline[read] = static_cast<uint8_t>(g_last_hnote ? g_last_hnote : 15u);
          }
          else if (where == HINT_TO_GENERAL) // (declined, and the express pass would decline as well: straight to the general pass)
          {
            direct.push_back(read);
            e.hint_decline[read] = static_cast<uint8_t>(g_last_hnote ? g_last_hnote : 15u);
          }
          else
            ++e.hinted_done;
        }
      }
      for (uint32_t read : direct)
      {
        uint64_t const before = e.second_pass_tasks;
        general(read * 2);
        e.pass_of[read] = e.second_pass_tasks != before ? 3 : 2;
      }
      for (uint32_t read = 0; read < n_reads; ++read) // (the reverse tasks pass 0 queued for pass 2)
      {
        uint32_t const len = meta[read].l_qseq;
        if (len >= 2 * K - 1 && len <= AlignCfg::MAX_READ && needs_reverse(meta[read], force_both))
          general(read * 2 + 1);
      }
      for (uint32_t first = 0; first < queue1.size(); first += 4)
      {
        uint32_t const n_valid = queue1.size() - first < 4 ? static_cast<uint32_t>(queue1.size() - first) : 4;
        uint32_t mask;
        if (wide)
        {
          std::memset(static_cast<void *>(e4_wide_ws.get()), fill, sizeof(Express4Workspace<Express4Wide>));
          mask = express4<WaveEmu, Express4Wide>(g, ix, *e4_wide_ws, seq, seq_stride, meta, 0, n_valid, records, rec_words, force != 0, queue1.data() + first);
        }
        else
        {
          std::memset(static_cast<void *>(e4_ws.get()), fill, sizeof(Express4Workspace<Express4Lean>));
          mask = express4<WaveEmu, Express4Lean>(g, ix, *e4_ws, seq, seq_stride, meta, 0, n_valid, records, rec_words, force != 0, queue1.data() + first);
        }
        for (uint32_t k = 0; k < n_valid; ++k)
          if ((mask >> k) & 1u)
          {
            uint64_t const before = e.second_pass_tasks;
            general(queue1[first + k] * 2);
            e.pass_of[queue1[first + k]] = e.second_pass_tasks != before ? 3 : 2;
          }
      }
      return 0;
    }
    if (four)
    {
      // pass 1, four reads per wavefront (gtx_align_express4_kernel)
      for (uint32_t first = 0; first < n_reads; first += 4)
      {
        uint32_t const n_valid = n_reads - first < 4 ? n_reads - first : 4;
        uint32_t mask;
        if (wide)
        {
          std::memset(static_cast<void *>(e4_wide_ws.get()), fill, sizeof(Express4Workspace<Express4Wide>));
          mask = express4<WaveEmu, Express4Wide>(g, ix, *e4_wide_ws, seq, seq_stride, meta, first, n_valid, records, rec_words, force != 0);
        }
        else
        {
          std::memset(static_cast<void *>(e4_ws.get()), fill, sizeof(Express4Workspace<Express4Lean>));
          mask = express4<WaveEmu, Express4Lean>(g, ix, *e4_ws, seq, seq_stride, meta, first, n_valid, records, rec_words, force != 0);
        }
        for (uint32_t k = 0; k < n_valid; ++k)
        {
          uint32_t const read = first + k, len = meta[read].l_qseq;
          if ((mask >> k) & 1u)
            general(read * 2);
          bool const rev = needs_reverse(meta[read], force_both) && len >= 2 * K - 1 && len <= AlignCfg::MAX_READ;
          if (rev)
            general(read * 2 + 1);
          else
            empty_record(read * 2 + 1, len);
        }
      }
      return 0;
    }
    for (uint32_t t = 0; t < 2 * n_reads; ++t)
    {
      uint32_t const read = t >> 1, orient = t & 1u;
      gtx_read_meta const m = meta[read];
      uint32_t * rec = records + static_cast<uint64_t>(t) * rec_words;
      uint32_t const len = m.l_qseq;
      bool const skip = len < 2 * K - 1 || len > AlignCfg::MAX_READ || (orient == 1 && !needs_reverse(m, force_both));
      if (skip)
      {
        empty_record(t, len);
        continue;
      }
      // pass 1 (gtx_align_express_kernel)
      std::memset(static_cast<void *>(seed_ws.get()), fill, sizeof(SeedWorkspace)); // LDS is not zeroed between reads
      if (!force && express_one<WaveEmu>(g, ix, *seed_ws, seq + static_cast<uint64_t>(read) * seq_stride, len, orient == 1, rec, rec_words))
        continue;
      general(t);
    }
    return 0;
  }

  int emu_big_records(void * p, const uint32_t ** words, uint64_t * capacity_words)
  {
    Emu & e = *static_cast<Emu *>(p);
    *words = e.arena.data();
    *capacity_words = e.arena_used;
    return 0;
  }

  void emu_big_records_rewind(void * p) { static_cast<Emu *>(p)->arena_used = 0; }

  void emu_notes(uint64_t * out, int reset)
  {
    for (int i = 0; i < 16; ++i)
    {
      out[i] = g_notes[i];
      if (reset)
        g_notes[i] = 0;
    }
  }

  void emu_hint_notes(uint64_t * out, int reset)
  {
    for (int i = 0; i < 16; ++i)
    {
      out[i] = g_hnotes[i];
      if (reset)
        g_hnotes[i] = 0;
    }
  }

  uint64_t emu_hinted_done(void * p) { return static_cast<Emu *>(p)->hinted_done; }
  uint64_t emu_exact_pass_tasks(void * p, int level) { return static_cast<Emu *>(p)->exact_pass_tasks[level & 3]; }

  // per read of the last emu_align with the position-hinted pass: finishing pass of the forward task, pass 0's decline note
  void emu_pass_of(void * p, uint8_t * pass_of, uint8_t * hint_decline, uint32_t n)
  {
    Emu & e = *static_cast<Emu *>(p);
    for (uint32_t i = 0; i < n && i < e.pass_of.size(); ++i)
    {
      pass_of[i] = e.pass_of[i];
      hint_decline[i] = e.hint_decline[i];
    }
  }

  uint64_t emu_general_tasks(void * p) { return static_cast<Emu *>(p)->general_tasks; }

  uint64_t emu_second_pass_tasks(void * p) { return static_cast<Emu *>(p)->second_pass_tasks; }
  uint64_t emu_wide_pass_tasks(void * p) { return static_cast<Emu *>(p)->wide_pass_tasks; }

  // same contract as gtx_calls_batch, host pointers
  int emu_calls(void * p, const gtx_score_buffers * acc, uint8_t * phred, gtx_sample_call * calls)
  {
    using namespace gtx;
    Emu & e = *static_cast<Emu *>(p);
    GraphView const g = e.graph.view();
    uint64_t const cells = static_cast<uint64_t>(acc->n_samples) * g.n_hap;
    for (uint64_t cell = 0; cell < cells; ++cell)
      call_cell(g, cell, acc->d_log_score, acc->d_gt_cov, acc->d_hap_u32, phred, calls);
    return 0;
  }

  // the position-hint tables of the index (IndexView::pos_flags): n = positions; flags may be NULL
  uint32_t emu_hint_flags(void * p, uint32_t * flags, uint32_t cap)
  {
    Emu & e = *static_cast<Emu *>(p);
    for (uint32_t i = 0; flags && i < e.index.n_hint && i < cap; ++i)
      flags[i] = e.index.pos_flags[i].x;
    return e.index.n_hint;
  }

  int emu_workspace_bytes(int big) { return static_cast<int>(big ? sizeof(gtx::big::AlignWorkspace) : sizeof(gtx::AlignWorkspace));
  }

  // same contract as gtx_score_batch, host pointers; returns the number of refused items
  int emu_score(void * p, const gtx_score_item * items, uint32_t n_items, const uint32_t * records, uint32_t rec_words,
                const gtx_score_buffers * acc)
  {
    using namespace gtx;
    Emu & e = *static_cast<Emu *>(p);
    GraphView const g = e.graph.view();
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
    a.big_records = e.arena.data();
    a.ref_depth = e.params.is_sv_graph ? acc->d_ref_depth : nullptr;
    a.ref_depth_len = acc->ref_depth_len;
    ScoreParams par{static_cast<uint32_t>(e.params.is_sv_graph != 0), static_cast<uint32_t>(e.params.hq_reads != 0),
                    static_cast<uint32_t>(e.params.is_segment_calling != 0), 0};
    uint32_t errors = 0;
    std::vector<RecentHap> small(2 * SCORE_MAX_HAPS), large(2 * SCORE_MAX_HAPS_BIG);
    bool has_wide_sites = false;
    for (uint32_t n : e.graph.ref_nvar)
      has_wide_sites = has_wide_sites || n > 64;
    std::vector<RecentHapWide> wide_tables(has_wide_sites ? 2 * SCORE_MAX_HAPS_WIDE : 0);
    for (uint32_t i = 0; i < n_items; ++i)
      if (item_is_trivial(items[i], records, rec_words, a.ref_depth != nullptr)) // stage 1 (gtx_score_triage_kernel)
        continue;
      else if (!score_item<WaveEmu>(g, par, items[i], records, rec_words, a, small.data(), small.data() + SCORE_MAX_HAPS, SCORE_MAX_HAPS))
      {
        // second scoring pass (gtx_score_big_kernel; gtx_score_wide_kernel for a graph with a site of more than 64 alleles)
        if (e.params.no_second_pass)
          ++errors;
        else if (has_wide_sites)
          errors += !score_item<WaveEmu>(g, par, items[i], records, rec_words, a, wide_tables.data(), wide_tables.data() + SCORE_MAX_HAPS_WIDE,
                                         SCORE_MAX_HAPS_WIDE);
        else if (!score_item<WaveEmu>(g, par, items[i], records, rec_words, a, large.data(), large.data() + SCORE_MAX_HAPS_BIG, SCORE_MAX_HAPS_BIG))
          ++errors;
      }
    return static_cast<int>(errors);
  }

  // The compare of the position-hinted pass in its two forms (hinted.hpp): nibble words (the definition) and bit planes
  // (what the kernel runs).  base: reference nibble codes [n_base], idx: position of the read's first base, row: the read's
  // packed bases (>= 80 bytes), L bases.  out: k[0..4], upto, more of the nibble form, then of the plane form.
  void emu_hint_compare(const uint8_t * base, uint32_t n_base, uint32_t idx, const uint8_t * row, uint32_t L, uint32_t * out)
  {
    using namespace gtx;
    std::vector<uint32_t> ref4(n_base / 8 + 32, 0), refp(4 * (n_base / 32 + 8), 0);
    for (uint32_t i = 0; i < n_base; ++i)
    {
      ref4[i >> 3] |= static_cast<uint32_t>(base[i]) << (28 - 4 * (i & 7u));
      for (uint32_t b = 0; b < 4; ++b)
        refp[4 * (i >> 5) + b] |= ((static_cast<uint32_t>(base[i]) >> b) & 1u) << (i & 31u);
    }
    uint32_t words[20], planes[20];
    std::memcpy(words, row, 80);
    planes_from_nibbles(row, 80u, planes, 5u);
    HintCounts a{}, b{};
    hint_compare_nibbles(words, 80u, ref4.data() + (idx >> 3), 4 * (idx & 7u), L, a);
    hint_compare(planes, 80u, refp.data() + 4 * (idx >> 5), idx & 31u, L, b);
    for (int i = 0; i < 5; ++i)
    {
      out[i] = a.k[i];
      out[7 + i] = b.k[i];
    }
    out[5] = a.upto;
    out[6] = a.more;
    out[12] = b.upto;
    out[13] = b.more;
  }

  // gtx_scores_replay_log over host arrays: marks the cells at the guard of `acc` (the block summed over all ranks), logs their
  // explain_to_score calls by `items` with the kernel source in replay mode.  Returns the number of entries (<= cap), -1 / -2 on a
  // table or log overflow.
  long emu_score_replay_log(void * p, const gtx_score_item * items, uint32_t n_items, const uint32_t * records, uint32_t rec_words,
                            const gtx_score_buffers * acc, uint32_t item_base, gtx_replay_entry * out, long cap)
  {
    using namespace gtx;
    Emu & e = *static_cast<Emu *>(p);
    GraphView const g = e.graph.view();
    uint64_t const n_cells = static_cast<uint64_t>(acc->n_samples) * e.graph.n_hap;
    std::vector<uint32_t> marked((n_cells + 31) / 32, 0u);
    uint64_t n_marked = 0;
    for (uint64_t cell = 0; cell < n_cells; ++cell)
    {
      uint32_t const m = acc->d_hap_u32[4 * cell];
      if ((m & GTX_CELL_REPLAYED) || m < SATURATION_GUARD || e.graph.ref_nvar[cell % e.graph.n_hap] > 64)
        continue;
      marked[cell >> 5] |= 1u << (cell & 31u);
      ++n_marked;
    }
    if (n_marked == 0)
      return 0;
    std::vector<ReplayEntry> log(1u << 22);
    uint32_t count = 0;
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
    a.big_records = e.arena.data();
    a.replay_cells = marked.data();
    a.replay_log = log.data();
    a.replay_count = &count;
    a.replay_cap = static_cast<uint32_t>(log.size());
    ScoreParams par{static_cast<uint32_t>(e.params.is_sv_graph != 0), static_cast<uint32_t>(e.params.hq_reads != 0),
                    static_cast<uint32_t>(e.params.is_segment_calling != 0), 0};
    std::vector<RecentHap> large(2 * SCORE_MAX_HAPS_BIG);
    for (uint32_t i = 0; i < n_items; ++i)
    {
      a.replay_item = i;
      if (!score_item<WaveEmu>(g, par, items[i], records, rec_words, a, large.data(), large.data() + SCORE_MAX_HAPS_BIG, SCORE_MAX_HAPS_BIG))
        return -1;
    }
    if (count > log.size() || static_cast<long>(count) > cap)
      return -2;
    static_assert(sizeof(gtx_replay_entry) == sizeof(ReplayEntry), "gtx_replay_entry is ReplayEntry");
    for (uint32_t k = 0; k < count; ++k)
    {
      log[k].item += item_base;
      std::memcpy(out + k, &log[k], sizeof(ReplayEntry));
    }
    return static_cast<long>(count);
  }

  // gtx_scores_replay_apply over host arrays: the entries of all ranks replayed in call order with the library's host code
  // (score_replay.hpp), the exact rows stored into `acc`.  Returns the cells replayed.
  long emu_score_replay_apply(void * p, const gtx_score_buffers * acc, const gtx_replay_entry * entries, long n)
  {
    using namespace gtx;
    Emu & e = *static_cast<Emu *>(p);
    std::vector<ReplayEntry> log(static_cast<size_t>(n));
    if (n)
      std::memcpy(log.data(), entries, static_cast<size_t>(n) * sizeof(ReplayEntry));
    std::vector<ReplayedCell> const done = replay_cells(e.graph, log);
    for (ReplayedCell const & rc : done)
    {
      uint32_t const h = rc.cell % e.graph.n_hap, sample = rc.cell / e.graph.n_hap;
      acc->d_hap_u32[4ull * rc.cell] = rc.max_log_score | GTX_CELL_REPLAYED;
      std::copy(rc.log_score.begin(), rc.log_score.end(), acc->d_log_score + static_cast<uint64_t>(sample) * e.graph.total_tri + e.graph.tri_off[h]);
    }
    return static_cast<long>(done.size());
  }

  // gtx_scores_replay over host arrays: one process -- its own log, replayed.  Returns the cells replayed.
  long emu_score_replay(void * p, const gtx_score_item * items, uint32_t n_items, const uint32_t * records, uint32_t rec_words,
                        const gtx_score_buffers * acc)
  {
    std::vector<gtx_replay_entry> log(1u << 22);
    long const n = emu_score_replay_log(p, items, n_items, records, rec_words, acc, 0, log.data(), static_cast<long>(log.size()));
    if (n <= 0)
      return n;
    return emu_score_replay_apply(p, acc, log.data(), n);
  }
}
