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
  int align_blocks_per_cu = 8;
};

namespace gtx
{
extern thread_local std::string g_last_error;
int ctx_upload(gtx_ctx & c, int device);
void ctx_release_device(gtx_ctx & c);
} // namespace gtx
