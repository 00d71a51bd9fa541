// gtx_capi.cpp -- host half of the C ABI (include/gtx.h): context life cycle, inspection, finalisation and the
// per-record stream logic.  No compute on reads happens here.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <exception>
#include <memory>
#include <map>
#include <tuple>
#include <string>
#include <thread>
#include <unordered_map>

#include "gtx_ctx.hpp"
#include "gtx_devmem.hpp"
#include "graph_dev.hpp"
#if defined(__x86_64__)
#include <immintrin.h>
#endif

using namespace gtx;

extern "C"
{
  const char * gtx_strerror(int status)
  {
    switch (status)
    {
    case GTX_OK: return "ok";
    case GTX_ERR_ARG: return "bad argument";
    case GTX_ERR_NO_DEVICE: return "no HIP device (libgtx has no CPU path)";
    case GTX_ERR_HIP: return "HIP runtime error";
    case GTX_ERR_UNSUPPORTED: return "graph outside the supported envelope";
    case GTX_ERR_CAPACITY: return "caller buffer too small";
    case GTX_ERR_GRAPH: return "malformed graph view";
    case GTX_ERR_IO: return "file could not be read";
    default: return "unknown status";
    }
  }

  const char * gtx_last_error(void) { return g_last_error.c_str(); }

  int gtx_ctx_create(const gtx_graph_view * graph, const gtx_params * params, int device, gtx_ctx ** out)
  {
    if (!graph || !params || !out)
    {
      g_last_error = "gtx_ctx_create: NULL argument";
      return GTX_ERR_ARG;
    }
    *out = nullptr;
    bool const timing = std::getenv("GTX_TIMING") != nullptr; // stage times on stderr
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](char const * what)
    {
      auto const now = std::chrono::steady_clock::now();
      if (timing)
        std::fprintf(stderr, "[gtx] ctx_create %-24s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
      t_last = now;
    };
    auto c = std::make_unique<gtx_ctx>();
    c->params = *params;
    std::string const err = flatten_graph(*graph, *params, c->graph, device < 0); // (a device makes its position tables itself)
    lap("flatten the graph (host)");
    if (!err.empty())
    {
      g_last_error = err;
      return err.rfind("unsupported", 0) == 0 ? GTX_ERR_UNSUPPORTED : GTX_ERR_GRAPH;
    }
    for (uint32_t n : c->graph.ref_nvar)
      c->has_wide_sites = c->has_wide_sites || n > 64;
    char const * hb = std::getenv("GTX_INDEX_BUILD"); // A/B switch: "host" builds the tables on the host and uploads them
    if (device < 0)
    {
      build_index(c->graph, c->index);
      c->n_keys = static_cast<uint32_t>(c->index.keys.size());
      c->n_labels = static_cast<uint32_t>(c->index.labels.size());
      c->index_downloaded = true;
    }
    else
    {
      // the host lists the 32-mers of the sweep that walk through a site (index_graph's sweep); the ones inside a reference
      // node -- one per position -- and everything else, grouping, hash tables, hint tables, are made on the device
      // (gtx_index_dev.hip).  GTX_INDEX_RUNS=0: the host lists every k-mer (A/B, tests).
      std::vector<Emit> em;
      std::vector<EmitRun> runs;
      char const * er = std::getenv("GTX_INDEX_RUNS");
      bool const with_runs = !(er && er[0] == '0');
      enumerate_kmers(c->graph, em, with_runs ? &runs : nullptr);
      lap("list the k-mers through sites (host)");
      if (em.size() + (runs.empty() ? 0ull : static_cast<uint64_t>(runs.back().dev_before) + runs.back().count) >= SLOT_OFF_MASK) // (label offsets are 31 bits in the index slots: gtx_flat.hpp, SLOT_NB_KNOWN)
      {
        gtx::g_last_error = "gtx_ctx_create: more than 2^31 k-mer labels in one region";
        return GTX_ERR_UNSUPPORTED;
      }
      // (everything the device does for this context is ordered on a stream of this call's own: gtx_devmem.hpp)
      (void)hipSetDevice(device); // (a bad index is ctx_upload's to report)
      gtx::BuildStreamScope build_stream;
      int rc = ctx_upload(*c, device);
      lap("graph upload + scratch");
      if (rc == GTX_OK)
        rc = build_index_device(*c, em, runs);
      lap("index build (device)");
      if (rc != GTX_OK)
      {
        ctx_release_device(*c);
        return rc;
      }
      std::vector<uint8_t>().swap(c->upload_stage); // (the index build ended with a wait for the stream: the copy has been made)
      (void)hb;
    }
    *out = c.release();
    return GTX_OK;
  }

  void gtx_ctx_destroy(gtx_ctx * c)
  {
    if (!c)
      return;
    ctx_release_device(*c);
    delete c;
  }

  int gtx_ctx_special_positions(const gtx_ctx * c, uint32_t * n_special, uint32_t * ref_reach_poses, uint32_t * actual_poses,
                                uint32_t cap)
  {
    if (!c || !n_special)
      return GTX_ERR_ARG;
    uint32_t const n = static_cast<uint32_t>(c->graph.special_actual.size());
    *n_special = n;
    if (ref_reach_poses || actual_poses)
    {
      if (cap < n)
        return GTX_ERR_CAPACITY;
      if (ref_reach_poses)
        std::memcpy(ref_reach_poses, c->graph.special_ref_reach.data(), n * sizeof(uint32_t));
      if (actual_poses)
        std::memcpy(actual_poses, c->graph.special_actual.data(), n * sizeof(uint32_t));
    }
    return GTX_OK;
  }

  int gtx_ctx_score_layout(const gtx_ctx * c, gtx_score_layout * out)
  {
    if (!c || !out)
      return GTX_ERR_ARG;
    out->n_hap = c->graph.n_hap;
    out->total_tri = c->graph.total_tri;
    out->total_allele = c->graph.total_allele;
    out->total_near = c->graph.total_near;
    out->ref_depth_len = c->graph.ref_order.empty() ? 0u : c->graph.ref_order.back() + c->graph.ref_len.back() - c->graph.ref_order.front();
    return GTX_OK;
  }

  int gtx_ref_depth_finalize(uint32_t * ref_depth, uint32_t n_samples, uint32_t ref_depth_len, uint64_t * n_saturated)
  {
    if (!ref_depth && n_samples != 0)
      return GTX_ERR_ARG;
    uint64_t sat = 0;
    for (uint32_t s = 0; s < n_samples; ++s)
    {
      uint32_t * row = ref_depth + static_cast<uint64_t>(s) * (ref_depth_len + 1u);
      uint32_t run = 0;
      for (uint32_t i = 0; i < ref_depth_len; ++i)
      {
        run += row[i]; // (modulo 2^32: the -1 entries are 0xFFFFFFFF)
        row[i] = run < 0xFFFFu ? run : 0xFFFFu;
        sat += run >= 0xFFFFu ? 1u : 0u;
      }
      row[ref_depth_len] = 0;
    }
    if (n_saturated)
      *n_saturated = sat;
    return GTX_OK;
  }

  int gtx_ctx_near_pairs(const gtx_ctx * c, uint32_t * near_last, uint64_t * near_off)
  {
    if (!c)
      return GTX_ERR_ARG;
    HostGraph const & g = c->graph;
    for (uint32_t h = 0; h < g.n_hap; ++h)
    {
      if (near_last)
        near_last[h] = g.near_last[h];
      if (near_off)
        near_off[h] = g.near_off[h];
    }
    return GTX_OK;
  }

  int gtx_ctx_haplotypes(const gtx_ctx * c, uint32_t * hap_order, uint32_t * hap_cnum, uint64_t * tri_off, uint64_t * allele_off)
  {
    if (!c)
      return GTX_ERR_ARG;
    HostGraph const & g = c->graph;
    for (uint32_t h = 0; h < g.n_hap; ++h)
    {
      if (hap_order)
        hap_order[h] = g.var_order[g.ref_first_var[h]];
      if (hap_cnum)
        hap_cnum[h] = g.ref_nvar[h];
      if (tri_off)
        tri_off[h] = g.tri_off[h];
      if (allele_off)
        allele_off[h] = g.allele_off[h];
    }
    return GTX_OK;
  }

  int gtx_index_stats(const gtx_ctx * c, uint64_t * n_keys, uint64_t * n_labels)
  {
    if (!c)
      return GTX_ERR_ARG;
    if (n_keys)
      *n_keys = c->n_keys;
    if (n_labels)
      *n_labels = c->n_labels;
    return GTX_OK;
  }

  // (a device-built index keeps its keys and labels on the device: they are fetched when somebody looks)
  static int inspectable(const gtx_ctx * c)
  {
    return c->index_downloaded.load(std::memory_order_acquire) ? GTX_OK : download_index(*const_cast<gtx_ctx *>(c));
  }

  int gtx_index_get(const gtx_ctx * c, uint64_t key, gtx_label * out, uint32_t cap, uint32_t * n)
  {
    if (!c || !n)
      return GTX_ERR_ARG;
    *n = 0;
    if (int const rc = inspectable(c))
      return rc;
    HostIndex const & ix = c->index;
    auto it = std::lower_bound(ix.keys.begin(), ix.keys.end(), key); // PHIndex::get(key)
    if (it == ix.keys.end() || *it != key)
      return GTX_OK;
    std::size_t const k = static_cast<std::size_t>(it - ix.keys.begin());
    *n = ix.key_off[k + 1] - ix.key_off[k];
    if (out)
    {
      if (cap < *n)
        return GTX_ERR_CAPACITY;
      std::memcpy(out, ix.labels.data() + ix.key_off[k], sizeof(gtx_label) * *n);
    }
    return GTX_OK;
  }

  int gtx_index_dump(const gtx_ctx * c, uint64_t * keys, uint32_t * counts, gtx_label * labels)
  {
    if (!c)
      return GTX_ERR_ARG;
    if (int const rc = inspectable(c))
      return rc;
    HostIndex const & ix = c->index;
    for (size_t k = 0; k < ix.keys.size(); ++k)
    {
      if (keys)
        keys[k] = ix.keys[k];
      if (counts)
        counts[k] = ix.key_off[k + 1] - ix.key_off[k];
    }
    if (labels && !ix.labels.empty())
      std::memcpy(labels, ix.labels.data(), sizeof(gtx_label) * ix.labels.size());
    return GTX_OK;
  }

  int gtx_ctx_hint_table(const gtx_ctx * c, int which, void * out, uint64_t cap_bytes, uint64_t * bytes)
  {
    if (!c || !bytes || which < 0 || which > 6)
      return GTX_ERR_ARG;
    if (c->device >= 0)
      return download_hint_table(*c, which, out, cap_bytes, bytes);
    HostIndex const & ix = c->index;
    void const * src = nullptr;
    uint64_t n = 0;
    switch (which)
    {
    case 0: src = ix.pos_flags.data(); n = ix.pos_flags.size() * sizeof(uint2_t); break;
    case 1: src = ix.refp.data(); n = ix.refp.size() * sizeof(uint32_t); break;
    case 2: src = ix.tail_info.data(); n = ix.tail_info.size() * sizeof(uint2_t); break;
    case 3:
    case 4: src = ix.filt[which - 3].data(); n = ix.filt[which - 3].size() * sizeof(uint32_t); break;
    case 5: src = ix.win.data(); n = ix.win.size() * sizeof(HintWindow); break;
    default: src = ix.site_win.data(); n = ix.win.empty() ? 0 : ix.site_win.size() * sizeof(uint32_t); break;
    }
    *bytes = n;
    if (!out)
      return GTX_OK;
    if (cap_bytes < n)
      return GTX_ERR_CAPACITY;
    std::memcpy(out, src, n);
    return GTX_OK;
  }

  // haplotype.cpp:19-44 (saturating u8 / u16 counters) and :560 (sequential guard of explain_to_score)
  int gtx_scores_finalize(uint32_t * log_score, uint64_t n_log, uint32_t * gt_cov, uint64_t n_cov, uint32_t * hap_u32,
                          uint64_t n_hap_cells, uint64_t * n_saturated)
  {
    if (!gt_cov || !hap_u32 || !n_saturated)
      return GTX_ERR_ARG;
    (void)log_score;
    (void)n_log;
    uint64_t sat = 0;
    for (uint64_t i = 0; i < n_cov; ++i)
      if (gt_cov[i] > 0xFFFFu)
        gt_cov[i] = 0xFFFFu;
    for (uint64_t i = 0; i < n_hap_cells; ++i)
    {
      uint32_t * cell = hap_u32 + 4 * i;
      // a read is only added while max_log_score < 0xFFFF - epsilon, epsilon in [4,8]: below 0xFFFF - 8 no read was refused
      if (cell[0] & 0x80000000u) // replayed in call order by gtx_scores_replay: the exact value
        cell[0] &= 0x7FFFFFFFu;
      else if (cell[0] >= 0xFFFFu - 8u)
        ++sat;
      for (int k = 1; k < 4; ++k)
        if (cell[k] > 0xFFu)
          cell[k] = 0xFFu;
    }
    *n_saturated = sat;
    return GTX_OK;
  }

  // hts_parallel_reader.cpp:782-904.  Connection counts are uint16_t cells incremented without a guard in the reference
  // (vcf_writer.cpp:136), i.e. sums modulo 65536.
  int gtx_phase_flags(const gtx_ctx * c, uint32_t n_samples, const uint32_t * gt_cov, const uint32_t * conn_log, uint64_t n_conn,
                      const uint32_t * conn_near, gtx_phase_entry * out, uint64_t cap, uint64_t * n_out)
  {
    if (!c || !gt_cov || (n_conn && !conn_log) || !n_out || (cap && !out))
      return GTX_ERR_ARG;
    gtx::HostGraph const & g = c->graph;
    uint64_t const n_hap = g.n_hap;
    // connections grouped by (sample, hap1, allele1, hap2): support vector over hap2's alleles
    struct Key
    {
      uint32_t s, h1, a1, h2;
      bool operator<(Key const & o) const { return std::tie(h1, h2, s, a1) < std::tie(o.h1, o.h2, o.s, o.a1); }
    };
    std::map<Key, std::vector<uint16_t>> conn;
    for (uint64_t i = 0; i < n_conn; ++i)
    {
      uint32_t const * e = conn_log + 6 * i;
      if (e[0] >= n_samples || e[1] >= n_hap || e[3] >= n_hap || e[2] >= g.ref_nvar[e[1]] || e[4] >= g.ref_nvar[e[3]])
      {
        gtx::g_last_error = "gtx_phase_flags: connection entry out of range";
        return GTX_ERR_ARG;
      }
      auto & v = conn[Key{e[0], e[1], e[2], e[3]}];
      if (v.empty())
        v.assign(g.ref_nvar[e[3]], 0);
      v[e[4]] = static_cast<uint16_t>(v[e[4]] + e[5]);
    }
    if (conn_near) // the dense counters of near pairs: a row exists as soon as one of its cells was counted
      for (uint32_t s = 0; s < n_samples; ++s)
        for (uint32_t h1 = 0; h1 < n_hap; ++h1)
        {
          uint32_t const last = g.near_last[h1];
          if (last == h1)
            continue;
          uint64_t const first = g.allele_off[h1 + 1], width = g.allele_off[last] + g.ref_nvar[last] - first;
          for (uint32_t a1 = 0; a1 < g.ref_nvar[h1]; ++a1)
          {
            uint32_t const * row = conn_near + s * g.total_near + g.near_off[h1] + a1 * width;
            for (uint32_t h2 = h1 + 1; h2 <= last; ++h2)
            {
              uint32_t const * cell = row + (g.allele_off[h2] - first);
              bool any = false;
              for (uint32_t a2 = 0; a2 < g.ref_nvar[h2]; ++a2)
                any = any || cell[a2] != 0;
              if (!any)
                continue;
              auto & v = conn[Key{s, h1, a1, h2}];
              if (v.empty())
                v.assign(g.ref_nvar[h2], 0);
              for (uint32_t a2 = 0; a2 < g.ref_nvar[h2]; ++a2)
                v[a2] = static_cast<uint16_t>(v[a2] + cell[a2]);
            }
          }
        }
    constexpr int8_t HAP = 1, ANTI = 2; // IS_ANY_HAP_SUPPORT, IS_ANY_ANTI_HAP_SUPPORT (constants.hpp.in:56-57)
    using PhKey = std::pair<uint16_t, uint16_t>;
    std::map<PhKey, std::map<PhKey, int8_t>> ph;
    auto cov = [&](uint64_t s, uint64_t h, uint64_t a) { return gt_cov[s * g.total_allele + g.allele_off[h] + a]; };
    auto total = [&](uint64_t s, uint64_t h)
    {
      double t = 0.0;
      for (uint32_t a = 0; a < g.ref_nvar[h]; ++a)
        t += cov(s, h, a);
      return t;
    };
    auto it = conn.begin();
    while (it != conn.end())
    {
      // one (hap1, hap2) group; the reference only looks at pairs whose variant orders are < 100 apart (:800-801)
      uint32_t const h1 = it->first.h1, h2 = it->first.h2;
      auto group_end = it;
      while (group_end != conn.end() && group_end->first.h1 == h1 && group_end->first.h2 == h2)
        ++group_end;
      // Haplotype::gt.id = order of the site's variant nodes (graph.cpp:680-704)
      bool const near = h2 > h1 && static_cast<long>(g.var_order[g.ref_first_var[h2]]) < static_cast<long>(g.var_order[g.ref_first_var[h1]]) + 100;
      if (near)
        for (auto e = it; e != group_end; ++e)
        {
          uint32_t const s = e->first.s, cov1 = e->first.a1;
          if (cov1 == 0)
            continue; // skip reference (:812)
          double const total1 = total(s, h1), total2 = total(s, h2);
          uint32_t const c1 = cov(s, h1, cov1);
          bool const clearly1 = c1 >= 4 || static_cast<double>(c1) / total1 >= 0.28;
          bool const not1 = c1 <= 2 || static_cast<double>(c1) / total1 < 0.22;
          auto & row = ph[{static_cast<uint16_t>(h1), static_cast<uint16_t>(cov1)}];
          std::vector<uint16_t> const & support_vec = e->second;
          long total_support = 0;
          for (uint16_t v : support_vec)
            total_support += v;
          for (uint32_t cov2 = 1; cov2 < support_vec.size(); ++cov2)
          {
            double const support = static_cast<double>(support_vec[cov2]);
            uint32_t const c2 = cov(s, h2, cov2);
            bool const clearly2 = c2 >= 4 || static_cast<double>(c2) / total2 >= 0.28;
            bool const not2 = c2 <= 2 || static_cast<double>(c2) / total2 < 0.22;
            int8_t flag;
            if (not1 && not2)
              continue;
            if ((not1 && clearly2) || (not2 && clearly1))
              flag = ANTI;
            else
            {
              if (total_support <= 2)
                continue;
              if (clearly1 && clearly2 && support / static_cast<double>(total_support) > 0.78)
                flag = HAP;
              else if (support / static_cast<double>(total_support) < 0.22)
                flag = ANTI;
              else
                continue;
            }
            row[{static_cast<uint16_t>(h2), static_cast<uint16_t>(cov2)}] |= flag;
          }
        }
      it = group_end;
    }
    uint64_t n = 0;
    auto put = [&](PhKey a, PhKey b, int8_t f)
    {
      if (n < cap)
        out[n] = gtx_phase_entry{a.first, a.second, b.first, b.second, f, 0};
      ++n;
    };
    for (auto const & r : ph)
    {
      if (r.second.empty())
        put(r.first, PhKey{0xFFFF, 0xFFFF}, 0);
      for (auto const & e : r.second)
        put(r.first, e.first, e.second);
    }
    *n_out = n;
    if (n > cap)
    {
      gtx::g_last_error = "gtx_phase_flags: output buffer too small";
      return GTX_ERR_CAPACITY;
    }
    return GTX_OK;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// stream: the per-record control flow of parallel_reader_genotype_only / genotype_only for non-SV graphs
// (src/utilities/hts_parallel_reader.cpp:570-708, 245-338)
// ---------------------------------------------------------------------------------------------------------------
// ---- BAM nibble rows -> plane rows on the host (graph_dev.hpp: the layout the alignment kernels read).  With BMI2 one PEXT
// gathers a plane's 16 bits of 16 bases; the pairs come out swapped (a byte holds the even base in its HIGH nibble).
namespace
{
#if defined(__x86_64__)
__attribute__((target("bmi2"))) void planes_row_bmi2(uint8_t const * nib, uint32_t nib_bytes, uint32_t * out, uint32_t groups)
{
  for (uint32_t g = 0; g < groups; ++g)
  {
    uint32_t o[4] = {0, 0, 0, 0};
    for (uint32_t h = 0; h < 2; ++h)
    {
      uint32_t const at = 16u * g + 8u * h;
      uint64_t x = 0;
      if (at + 8u <= nib_bytes)
        std::memcpy(&x, nib + at, 8);
      else if (at < nib_bytes)
        std::memcpy(&x, nib + at, nib_bytes - at);
      for (uint32_t b = 0; b < 4; ++b)
      {
        uint32_t const r = static_cast<uint32_t>(_pext_u64(x, 0x1111111111111111ull << b));
        o[b] |= (((r & 0x5555u) << 1) | ((r >> 1) & 0x5555u)) << (16 * h);
      }
    }
    out[4 * g + 0] = o[0];
    out[4 * g + 1] = o[1];
    out[4 * g + 2] = o[2];
    out[4 * g + 3] = o[3];
  }
}
#define GTX_HAVE_BMI2_PATH 1
#endif

void planes_row(uint8_t const * nib, uint32_t nib_bytes, uint32_t * out, uint32_t groups)
{
#ifdef GTX_HAVE_BMI2_PATH
  static bool const bmi2 = __builtin_cpu_supports("bmi2");
  if (bmi2)
  {
    planes_row_bmi2(nib, nib_bytes, out, groups);
    return;
  }
#endif
  gtx::planes_from_nibbles(nib, nib_bytes, out, groups);
}
} // namespace

extern "C" int gtx_pack_planes(const uint8_t * seq, uint32_t seq_stride, uint32_t n, uint8_t * planes, uint32_t plane_stride)
{
  if ((n != 0 && (!seq || !planes)) || seq_stride == 0 || plane_stride == 0 || (plane_stride % gtx::PLANE_GROUP_BYTES) != 0 ||
      (reinterpret_cast<uintptr_t>(planes) & 3u) != 0)
  {
    gtx::g_last_error = "gtx_pack_planes: bad argument (plane rows are 16-byte groups of 32-bit words)";
    return GTX_ERR_ARG;
  }
  for (uint32_t i = 0; i < n; ++i)
    planes_row(seq + static_cast<uint64_t>(i) * seq_stride, seq_stride, reinterpret_cast<uint32_t *>(planes + static_cast<uint64_t>(i) * plane_stride),
               plane_stride / gtx::PLANE_GROUP_BYTES);
  return GTX_OK;
}

struct gtx_stream
{
  gtx_params params{};
  uint32_t plane_stride = 0; // != 0: gtx_stream_push writes the alignment tasks' bases as plane rows of this pitch
  struct Parked
  {
    gtx_rec_meta meta;
    uint32_t sample;
  };
  std::vector<std::unordered_map<uint64_t, Parked>> parked; // per read group: read name -> parked mate
  // SV calling: coverage filter state (hts_parallel_reader.cpp:594-633)
  std::vector<double> avg_cov_by_readlen;
  std::vector<std::vector<uint16_t>> bin_counts;
  long first_pos = 0;

  bool update_bin_count(gtx_stream_record const & r)
  {
    if (!params.is_sv_graph || avg_cov_by_readlen.empty())
      return true;
    if (r.sample >= avg_cov_by_readlen.size() || avg_cov_by_readlen[r.sample] <= 0.0)
      return true;
    uint16_t const max_bin_count = static_cast<uint16_t>(std::min(65535l, static_cast<long>(avg_cov_by_readlen[r.sample] * 50.0 * 3.0 + 0.5)));
    if (bin_counts.size() <= r.sample)
      bin_counts.resize(r.sample + 1);
    auto & bins = bin_counts[r.sample];
    long const bin = (static_cast<long>(r.pos) - first_pos) / 50l;
    if (bin >= static_cast<long>(bins.size()))
    {
      bins.resize(bin + 1, 0u);
      ++bins[bin];
      return true;
    }
    if (bin < 0) // cannot happen in a position-sorted stream
      return true;
    if (bins[bin] > max_bin_count)
      return false;
    ++bins[bin];
    return true;
  }

  // hts_parallel_reader.cpp:528-568
  static bool is_good_read(gtx_stream_record const & r)
  {
    if (r.flag & 4u) // IS_UNMAPPED
      return false;
    bool const is_mate_far_away = r.tid != r.mtid || std::labs(static_cast<long>(r.pos) - r.mpos) > 200000;
    if (r.mapq <= 15 && is_mate_far_away)
      return false;
    if (r.n_cigar >= 2)
    {
      constexpr uint32_t SOFT_CLIP = 4; // BAM_CSOFT_CLIP, 'S'
      bool const front_s = (r.cigar_front & 15u) == SOFT_CLIP, back_s = (r.cigar_back & 15u) == SOFT_CLIP;
      bool const is_one_clipped = (front_s && (r.cigar_front >> 4) >= 12) || (back_s && (r.cigar_back >> 4) >= 12);
      if ((front_s && back_s) || (r.mapq <= 15 && is_one_clipped))
        return false;
    }
    return true;
  }
  bool have_prev = false;
  int32_t prev_tid = 0, prev_pos = 0;
  std::vector<uint8_t> prev_seq;
  uint32_t prev_len = 0;
  uint32_t prev_align_index = 0;
  bool prev_forward_only = false; // the previous alignment task had no reverse orientation
  uint32_t next_align_index = 0;
  uint64_t n_records = 0, n_duplicated = 0;
};

extern "C"
{
  int gtx_stream_create(const gtx_params * params, uint32_t n_read_groups, gtx_stream ** out)
  {
    if (!params || !out || n_read_groups == 0)
      return GTX_ERR_ARG;
    auto s = std::make_unique<gtx_stream>();
    s->params = *params;
    s->parked.resize(n_read_groups);
    *out = s.release();
    return GTX_OK;
  }

  void gtx_stream_destroy(gtx_stream * s) { delete s; }

  int gtx_stream_push(gtx_stream * s, const gtx_stream_record * recs, const uint8_t * seq, uint32_t seq_stride, uint32_t n,
                      uint8_t * align_seq, gtx_read_meta * align_meta, uint32_t align_cap, uint32_t * n_align,
                      gtx_score_item * items, uint32_t item_cap, uint32_t * n_items)
  {
    if (!s || !recs || !seq || !align_seq || !align_meta || !n_align || !items || !n_items)
      return GTX_ERR_ARG;
    // everything that can fail is checked before the stream's state is touched: a failing call consumes nothing
    if (n > align_cap || n > item_cap)
    {
      g_last_error = "gtx_stream_push: align_cap and item_cap have to hold one entry per pushed record";
      return GTX_ERR_CAPACITY;
    }
    for (uint32_t i = 0; i < n; ++i)
    {
      if (recs[i].rg >= s->parked.size())
      {
        g_last_error = "gtx_stream_push: read group index out of range";
        return GTX_ERR_ARG;
      }
      if (recs[i].l_qseq > GTX_MAX_READ)
      {
        g_last_error = "gtx_stream_push: a read of " + std::to_string(recs[i].l_qseq) + " bases (the kernels align up to 256)";
        return GTX_ERR_UNSUPPORTED;
      }
      if ((static_cast<uint32_t>(recs[i].l_qseq) + 1u) / 2u > seq_stride || (s->plane_stride && recs[i].l_qseq > 2u * s->plane_stride))
      {
        g_last_error = "gtx_stream_push: a record is longer than seq_stride (or than the plane rows of gtx_stream_set_planes)";
        return GTX_ERR_ARG;
      }
    }
    uint32_t na = 0, ni = 0;
    for (uint32_t i = 0; i < n; ++i)
    {
      gtx_stream_record const & r = recs[i];
      if ((r.flag & s->params.sam_flag_filter) != 0 || (s->params.is_sv_graph && !gtx_stream::is_good_read(r))) // :658-663
        continue;
      if (r.rg >= s->parked.size())
      {
        g_last_error = "gtx_stream_push: read group index out of range";
        return GTX_ERR_ARG;
      }
      ++s->n_records;
      uint8_t const * rseq = seq + static_cast<uint64_t>(i) * seq_stride;
      uint32_t const nbytes = (static_cast<uint32_t>(r.l_qseq) + 1u) / 2u;
      if (nbytes > seq_stride)
        return GTX_ERR_ARG;
      // equal_pos_seq (include/graphtyper/utilities/hts_utils.hpp:110-128): same tid, pos, length and packed bytes
      bool const dup = s->have_prev && r.tid == s->prev_tid && r.pos == s->prev_pos && r.l_qseq == s->prev_len &&
                       std::memcmp(rseq, s->prev_seq.data(), nbytes) == 0;
      uint32_t align_index;
      if (!s->have_prev)
        s->first_pos = r.pos; // the first record that passes the filters anchors the coverage bins (:594)
      if (dup)
      {
        (void)s->update_bin_count(r);
        ++s->n_duplicated;
        align_index = s->prev_align_index; // prev_paths are reused as they are (hts_parallel_reader.cpp:666-684)
      }
      else
      {
        if (!s->update_bin_count(r) && s->have_prev) // too many reads in this bin: the record is skipped (:685-690)
        {
          --s->n_records;
          continue;
        }
        if (na >= align_cap)
          return GTX_ERR_CAPACITY;
        if (s->plane_stride)
          planes_row(rseq, nbytes, reinterpret_cast<uint32_t *>(align_seq + static_cast<uint64_t>(na) * s->plane_stride),
                     s->plane_stride / gtx::PLANE_GROUP_BYTES);
        else
          std::memcpy(align_seq + static_cast<uint64_t>(na) * seq_stride, rseq, nbytes);
        // position hint of the alignment: where read base 0 lies when the mapper was right (leading soft clip removed)
        int32_t const clip = (r.n_cigar != 0 && (r.cigar_front & 15u) == 4u) ? static_cast<int32_t>(r.cigar_front >> 4) : 0;
        // align_read (alignment.cpp:341-352): forward only for unpaired reads and concordant pairs
        bool const one_orientation = (r.flag & 1u) == 0u || (r.tid == r.mtid && r.isize > -1200 && r.isize < 1200 &&
                                                              (((r.flag & 16u) != 0u) != ((r.flag & 32u) != 0u)));
        s->prev_forward_only = one_orientation && !s->params.force_align_both_orientations;
        // (every item made of this task carries GTX_FLAG_FORWARD_ONLY then: the task's reverse record is never read, and the
        //  same bit in the read's flag word lets the alignment skip writing its empty header)
        align_meta[na] = gtx_read_meta{r.l_qseq, static_cast<uint16_t>(r.flag | (s->prev_forward_only ? GTX_FLAG_FORWARD_ONLY : 0u)), r.tid, r.mtid,
                                       r.isize, r.pos - clip};
        align_index = s->next_align_index++;
        ++na;
        s->have_prev = true;
        s->prev_tid = r.tid;
        s->prev_pos = r.pos;
        s->prev_len = r.l_qseq;
        s->prev_seq.assign(rseq, rseq + nbytes);
        s->prev_align_index = align_index;
      }
      gtx_rec_meta const me{align_index, static_cast<uint16_t>(r.flag | (s->prev_forward_only ? GTX_FLAG_FORWARD_ONLY : 0u)), r.mapq,
                            r.score_diff, r.pos, r.isize};
      auto & map = s->parked[r.rg];
      auto it = map.find(r.name_id);
      if (it == map.end())
      {
        if (r.flag & 1u) // IS_PAIRED: wait for the mate (hts_parallel_reader.cpp:283-290)
        {
          map.emplace(r.name_id, gtx_stream::Parked{me, r.sample});
          continue;
        }
        if (ni >= item_cap)
          return GTX_ERR_CAPACITY;
        gtx_score_item item{};
        item.first = me;
        item.second.align_index = GTX_INVALID_ID;
        item.sample = r.sample;
        items[ni++] = item;
        continue;
      }
      if ((it->second.meta.flag & 64u) == (r.flag & 64u)) // both mates claim the same IS_FIRST_IN_PAIR: the reference exits (:306-315)
      {
        g_last_error = "gtx_stream_push: two reads with one name have the same IS_FIRST_IN_PAIR";
        return GTX_ERR_ARG;
      }
      if (ni >= item_cap)
        return GTX_ERR_CAPACITY;
      gtx_score_item item{};
      item.first = it->second.meta;
      item.second = me;
      item.sample = r.sample;
      items[ni++] = item;
      map.erase(it);
    }
    *n_align = na;
    *n_items = ni;
    return GTX_OK;
  }

  int gtx_stream_set_planes(gtx_stream * s, uint32_t plane_stride)
  {
    if (!s || (plane_stride % gtx::PLANE_GROUP_BYTES) != 0)
    {
      g_last_error = "gtx_stream_set_planes: the pitch of plane rows is a multiple of 16 bytes";
      return GTX_ERR_ARG;
    }
    s->plane_stride = plane_stride;
    return GTX_OK;
  }

  int gtx_stream_set_coverage(gtx_stream * s, const double * avg_cov_by_readlen, uint32_t n_samples)
  {
    if (!s || (n_samples && !avg_cov_by_readlen))
      return GTX_ERR_ARG;
    s->avg_cov_by_readlen.assign(avg_cov_by_readlen, avg_cov_by_readlen + n_samples);
    return GTX_OK;
  }

  int gtx_stream_finish(gtx_stream * s, gtx_score_item * items, uint32_t item_cap, uint32_t * n_items)
  {
    if (!s || !n_items || (item_cap && !items))
      return GTX_ERR_ARG;
    uint32_t ni = 0;
    if (s->params.is_sv_graph)
      for (auto const & map : s->parked)
        for (auto const & kv : map)
        {
          if (ni >= item_cap)
            return GTX_ERR_CAPACITY; // nothing was forgotten: call again with room for gtx_stream_counts' parked reads
          gtx_score_item item{};
          item.first = kv.second.meta;
          item.second = kv.second.meta;
          item.second.flag ^= (64u | 16u); // IS_FIRST_IN_PAIR | IS_SEQ_REVERSED (:729-730)
          item.sample = kv.second.sample;
          item.kind = GTX_ITEM_LEFTOVER;
          items[ni++] = item;
        }
    for (auto & map : s->parked)
      map.clear();
    *n_items = ni;
    return GTX_OK;
  }

  int gtx_stream_counts(const gtx_stream * s, uint64_t * n_records, uint64_t * n_duplicated, uint64_t * n_parked)
  {
    if (!s)
      return GTX_ERR_ARG;
    if (n_records)
      *n_records = s->n_records;
    if (n_duplicated)
      *n_duplicated = s->n_duplicated;
    if (n_parked)
    {
      uint64_t p = 0;
      for (auto const & m : s->parked)
        p += m.size();
      *n_parked = p;
    }
    return GTX_OK;
  }
}
