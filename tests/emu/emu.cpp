// emu.cpp -- TEST HARNESS ONLY: runs the kernel sources (graphtyper_amd/csrc/align_core.hpp, score_core.hpp) on the
// host, with 64 OS threads + barriers standing in for the 64 lanes of one wavefront.  It exists so that kernel logic
// can be debugged in a container without a GPU; it is never built into, linked against or loaded by libgtx.so, and
// nothing outside tests/ uses it.  Parity claims are made by the `-m gpu` tests through the C ABI only.
#include <pthread.h>

#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../graphtyper_amd/csrc/gtx_flat.hpp"
#include "../../graphtyper_amd/csrc/score_core.hpp"

namespace
{
struct EmuWaveCtx
{
  pthread_barrier_t bar;
  uint64_t slots[64];
};

thread_local uint32_t t_lane = 0;
thread_local EmuWaveCtx * t_ctx = nullptr;

struct WaveEmu
{
  static uint32_t lane() { return t_lane; }
  static void sync()
  {
    if (t_ctx)
      pthread_barrier_wait(&t_ctx->bar);
  }
  static uint64_t ballot(bool p)
  {
    t_ctx->slots[t_lane] = p ? 1u : 0u;
    sync();
    uint64_t m = 0;
    for (int i = 0; i < 64; ++i)
      m |= t_ctx->slots[i] << i;
    sync();
    return m;
  }
  static uint32_t excl_scan(uint32_t v, uint32_t & total)
  {
    t_ctx->slots[t_lane] = v;
    sync();
    uint32_t pre = 0, tot = 0;
    for (uint32_t i = 0; i < 64; ++i)
    {
      if (i < t_lane)
        pre += static_cast<uint32_t>(t_ctx->slots[i]);
      tot += static_cast<uint32_t>(t_ctx->slots[i]);
    }
    sync();
    total = tot;
    return pre;
  }
  static uint32_t atomic_add_u32(uint32_t * p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
  static void atomic_add_u64(unsigned long long * p, unsigned long long v) { __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
};

struct Emu
{
  gtx_params params{};
  gtx::HostGraph graph;
  gtx::HostIndex index;
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

  // same contract as gtx_align_batch, host pointers
  int emu_align(void * p, const uint8_t * seq, uint32_t seq_stride, const gtx_read_meta * meta, uint32_t n_reads,
                uint32_t * records, uint32_t rec_words)
  {
    using namespace gtx;
    Emu & e = *static_cast<Emu *>(p);
    GraphView const g = e.graph.view();
    IndexView ix{e.index.slots.data(), e.index.dev_labels.data(), e.index.log2_cap, static_cast<uint32_t>(e.params.max_index_labels)};
    auto ws = std::make_unique<AlignWorkspace>();
    EmuWaveCtx wctx;
    pthread_barrier_init(&wctx.bar, nullptr, 64);
    bool const force_both = e.params.force_align_both_orientations != 0;
    auto body = [&](uint32_t lane)
    {
      t_lane = lane;
      t_ctx = &wctx;
      for (uint32_t t = 0; t < 2 * n_reads; ++t)
      {
        uint32_t const read = t >> 1, orient = t & 1u;
        gtx_read_meta const m = meta[read];
        uint32_t * rec = records + static_cast<uint64_t>(t) * rec_words;
        uint32_t const len = m.l_qseq;
        bool const skip = len < 2 * K - 1 || len > AlignCfg::MAX_READ || (orient == 1 && !needs_reverse(m, force_both));
        if (skip)
        {
          if (lane == 0)
          {
            rec[0] = len > AlignCfg::MAX_READ ? (static_cast<uint32_t>(GTX_ST_RECORD_OVERFLOW) << 16) : 0u;
            rec[1] = len << 16;
          }
          continue;
        }
        align_one<WaveEmu>(g, ix, *ws, seq + static_cast<uint64_t>(read) * seq_stride, len, orient == 1, rec, rec_words);
      }
    };
    std::vector<std::thread> th;
    for (uint32_t l = 1; l < 64; ++l)
      th.emplace_back(body, l);
    body(0);
    for (auto & t : th)
      t.join();
    pthread_barrier_destroy(&wctx.bar);
    t_ctx = nullptr;
    return 0;
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
    ScoreParams par{static_cast<uint32_t>(e.params.is_sv_graph != 0), static_cast<uint32_t>(e.params.hq_reads != 0),
                    static_cast<uint32_t>(e.params.is_segment_calling != 0), 0};
    uint32_t errors = 0;
    t_ctx = nullptr;
    for (uint32_t i = 0; i < n_items; ++i)
      score_item<WaveEmu>(g, par, items[i], records, rec_words, a, &errors);
    return static_cast<int>(errors);
  }
}
