// gtx_ctx.hpp -- the opaque context behind gtx_ctx* (host copies + device copies of graph and index)
#pragma once
#include <atomic>
#include <string>
#include <vector>

#include "gtx_flat.hpp"

struct gtx_ctx
{
  gtx_params params{};
  gtx::HostGraph graph;
  gtx::HostIndex index;
  int device = -1; // -1: inspection-only context (no device entry point works)
  int n_cu = 0;
  std::vector<void *> dev_allocs;
  gtx::GraphView dev_graph{};
  gtx::IndexView dev_index{};
  uint32_t * d_error_flag = nullptr;
  static constexpr unsigned N_TASK_COUNTERS = 64; // one per launch in flight (launches may overlap on different streams)
  uint32_t * d_task_counters = nullptr;
  std::atomic<unsigned> launch_seq{0};
  int align_blocks_per_cu = 8, express_blocks_per_cu = 16, express4_blocks_per_cu = 8;
  int express4_wide_blocks_per_cu = 8;
  bool express4_wide = false; // pass 1 runs gtx_align_express4_wide_kernel (express4_prefers_wide, gtx_flat.hpp)
  uint32_t * d_queue = nullptr; // tasks pass 1 hands to pass 2 (grow-only)
  uint64_t queue_cap = 0;
  void * pass_events[4] = {nullptr, nullptr, nullptr, nullptr}; // hipEvent_t around the three passes (gtx_ctx_pass_times)
  // second pass (reads that overflowed the LDS-sized tables): task list, HBM workspaces, arena for long records
  uint32_t * d_big_tasks = nullptr;  // (read * 2 + orientation) of every queued task
  uint32_t big_task_cap = 0;
  uint32_t * d_big_state = nullptr;  // [0] tasks queued, [1] claim cursor, [2] arena cursor (words), [3] tasks dropped (list full)
  void * d_big_ws = nullptr;         // big_blocks workspaces
  uint32_t big_blocks = 0;
  uint32_t * d_big_records = nullptr;
  uint64_t big_record_words = 0;
  // second scoring pass (items whose reads touch more variant sites than the main pass' tables hold)
  static constexpr uint32_t SCORE_QUEUE_CAP = 1u << 20, SCORE_BIG_THREADS = 1024;
  uint32_t * d_score_state = nullptr; // [0] items queued
  uint32_t * d_score_queue = nullptr;
  void * d_score_tables = nullptr;
  uint32_t * d_score_work = nullptr; // [0] number of items the triage kernel found worth scoring, [1..] their indices (grow-only)
  uint32_t score_work_cap = 0;
};

namespace gtx
{
extern thread_local std::string g_last_error;
int ctx_upload(gtx_ctx & c, int device);
void ctx_release_device(gtx_ctx & c);
} // namespace gtx
