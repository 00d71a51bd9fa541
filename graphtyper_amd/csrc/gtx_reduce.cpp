// gtx_reduce.cpp -- multi-GPU exchange step of the path: one packed allocation for the score
// accumulators of a region and one RCCL all-reduce group over it.
//
// The reference has no distributed layer; its per-thread VcfWriters meet in per-pool files that are merged on the host
// (src/typer/caller.cpp:439-482, src/typer/vcf_operations.cpp:366-374).  Every per-read effect on the accumulators is an
// integer addition, so with reads sharded over the GPUs of a node the only exchange is a sum of the counters.
//
// RCCL is bound at run time (dlopen): a process that already carries an RCCL (PyTorch ships its own librccl.so) keeps
// using that one, a single-GPU host program needs none, and libgtx.so has no link-time dependency on it.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>

#include "gtx_ctx.hpp"
#include "gtx_devmem.hpp"

using namespace gtx;

namespace
{
struct Rccl
{
  void * handle = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclGroupStart) group_start = nullptr;
  decltype(&ncclGroupEnd) group_end = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  std::string why;
};

Rccl & rccl()
{
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // an RCCL that is already part of the process first (RTLD_NOLOAD), then the loader's search path, then ROCm's
    char const * names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (char const * n : names)
      if (!r.handle)
        r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    for (char const * n : names)
      if (!r.handle)
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle)
    {
      r.why = "librccl.so not found";
      return;
    }
    auto sym = [&](char const * name) -> void *
    {
      void * p = dlsym(r.handle, name);
      if (!p && r.why.empty())
        r.why = std::string("librccl.so lacks ") + name;
      return p;
    };
    r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(sym("ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(sym("ncclCommInitRank"));
    r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(sym("ncclCommDestroy"));
    r.all_reduce = reinterpret_cast<decltype(r.all_reduce)>(sym("ncclAllReduce"));
    r.group_start = reinterpret_cast<decltype(r.group_start)>(sym("ncclGroupStart"));
    r.group_end = reinterpret_cast<decltype(r.group_end)>(sym("ncclGroupEnd"));
    r.error_string = reinterpret_cast<decltype(r.error_string)>(sym("ncclGetErrorString"));
  });
  return r;
}

bool rccl_ready()
{
  Rccl & r = rccl();
  if (r.handle && r.why.empty())
    return true;
  g_last_error = "RCCL unavailable: " + r.why;
  return false;
}

bool nccl_ok(ncclResult_t e, char const * what)
{
  if (e == ncclSuccess)
    return true;
  Rccl & r = rccl();
  g_last_error = std::string(what) + ": " + (r.error_string ? r.error_string(e) : "RCCL error");
  return false;
}

// Sizes (in elements) of the accumulator sections for this graph and sample count
struct Sections
{
  uint64_t stat_u64, log_score, gt_cov, hap_u32, stat_u32, conn_near, ref_depth;
  uint32_t ref_depth_len;
  uint64_t u32_total() const { return log_score + gt_cov + hap_u32 + stat_u32 + conn_near + ref_depth; }
};

Sections sections_of(gtx_ctx const & c, uint32_t n_samples)
{
  HostGraph const & g = c.graph;
  Sections s;
  s.stat_u64 = g.n_hap + 2 * g.total_allele;
  s.log_score = static_cast<uint64_t>(n_samples) * g.total_tri;
  s.gt_cov = static_cast<uint64_t>(n_samples) * g.total_allele;
  s.hap_u32 = static_cast<uint64_t>(n_samples) * g.n_hap * 4;
  s.stat_u32 = g.n_hap + 6 * g.total_allele;
  s.conn_near = static_cast<uint64_t>(n_samples) * g.total_near;
  // SV calling keeps the reference-depth track (a difference array: one word more than positions)
  s.ref_depth_len = g.ref_order.empty() ? 0u : g.ref_order.back() + g.ref_len.back() - g.ref_order.front();
  s.ref_depth = c.params.is_sv_graph ? static_cast<uint64_t>(n_samples) * (s.ref_depth_len + 1u) : 0u;
  return s;
}

// the buffers are one block laid out as gtx_scores_alloc does it
bool is_packed(Sections const & s, gtx_score_buffers const & b)
{
  uint32_t const * u32 = reinterpret_cast<uint32_t const *>(b.d_stat_u64 + s.stat_u64);
  return b.d_log_score == u32 && b.d_gt_cov == u32 + s.log_score && b.d_hap_u32 == b.d_gt_cov + s.gt_cov &&
         b.d_stat_u32 == b.d_hap_u32 + s.hap_u32 && b.d_conn_near == b.d_stat_u32 + s.stat_u32 &&
         (s.ref_depth == 0 ? true : b.d_ref_depth == b.d_conn_near + s.conn_near);
}
} // namespace

// (on_stream: the block is zeroed on `stream` and the call does not wait -- for a caller whose first use of it is on that stream)
static int scores_alloc_impl(gtx_ctx * c, uint32_t n_samples, uint32_t conn_cap, gtx_score_buffers * out, uint64_t * reduced_bytes, void * stream, bool on_stream)
{
  {
    if (!c || !out || n_samples == 0)
    {
      g_last_error = "gtx_scores_alloc: bad argument";
      return GTX_ERR_ARG;
    }
    if (c->device < 0)
    {
      g_last_error = "context was created without a device (libgtx has no CPU path)";
      return GTX_ERR_NO_DEVICE;
    }
    Sections const s = sections_of(*c, n_samples);
    // [u64 stats][u32: log_score, gt_cov, hap_u32, stat_u32, conn_near] = what is summed over ranks; then the
    // rank-local connection log (count words first)
    uint64_t const reduced = s.stat_u64 * 8 + s.u32_total() * 4;
    uint64_t const bytes = reduced + 2 * 4 + static_cast<uint64_t>(conn_cap) * 6 * 4;
    void * p = nullptr;
    if (hipSetDevice(c->device) != hipSuccess || gtx::dev_malloc(&p, bytes ? bytes : 8) != hipSuccess ||
        (on_stream ? hipMemsetAsync(p, 0, bytes, static_cast<hipStream_t>(stream)) : gtx::dev_zero(p, bytes)) != hipSuccess)
    {
      if (p)
        (void)gtx::dev_free(p);
      g_last_error = "gtx_scores_alloc: hipMalloc of " + std::to_string(bytes) + " bytes failed";
      return GTX_ERR_HIP;
    }
    std::memset(out, 0, sizeof(*out));
    out->n_samples = n_samples;
    out->d_stat_u64 = static_cast<uint64_t *>(p);
    uint32_t * u32 = reinterpret_cast<uint32_t *>(out->d_stat_u64 + s.stat_u64);
    out->d_log_score = u32;
    out->d_gt_cov = out->d_log_score + s.log_score;
    out->d_hap_u32 = out->d_gt_cov + s.gt_cov;
    out->d_stat_u32 = out->d_hap_u32 + s.hap_u32;
    out->d_conn_near = out->d_stat_u32 + s.stat_u32;
    out->d_ref_depth = s.ref_depth ? out->d_conn_near + s.conn_near : nullptr;
    out->ref_depth_len = s.ref_depth ? s.ref_depth_len : 0u;
    out->d_conn_count = out->d_conn_near + s.conn_near + s.ref_depth;
    out->d_conn_log = out->d_conn_count + 2;
    out->conn_cap = conn_cap;
    if (reduced_bytes)
      *reduced_bytes = reduced;
    return GTX_OK;
  }
}

int gtx::scores_alloc_on(gtx_ctx * c, uint32_t n_samples, uint32_t conn_cap, gtx_score_buffers * out, uint64_t * reduced_bytes, void * stream)
{
  return scores_alloc_impl(c, n_samples, conn_cap, out, reduced_bytes, stream, true);
}

extern "C"
{
  int gtx_scores_alloc(gtx_ctx * c, uint32_t n_samples, uint32_t conn_cap, gtx_score_buffers * out, uint64_t * reduced_bytes)
  {
    return scores_alloc_impl(c, n_samples, conn_cap, out, reduced_bytes, nullptr, false);
  }

  int gtx_scores_zero(gtx_ctx * c, const gtx_score_buffers * b, void * stream)
  {
    if (!c || !b || !b->d_stat_u64)
      return GTX_ERR_ARG;
    Sections const s = sections_of(*c, b->n_samples);
    if (!is_packed(s, *b) || b->d_conn_count != b->d_stat_u32 + s.stat_u32 + s.conn_near + s.ref_depth)
    {
      g_last_error = "gtx_scores_zero: buffers were not made by gtx_scores_alloc";
      return GTX_ERR_ARG;
    }
    // (the log's content is defined by its count words -- so a few bytes of its first entry may be zeroed with them: the size is
    //  rounded up to sixteen bytes where the log has an entry's room, and the runtime makes ONE fill kernel of it instead of an
    //  aligned one and a tail; the block itself is aligned by the allocator)
    uint64_t bytes = s.stat_u64 * 8 + s.u32_total() * 4 + 2 * 4;
    if (b->conn_cap >= 1)
      bytes = (bytes + 15u) & ~static_cast<uint64_t>(15u);
    if (hipMemsetAsync(b->d_stat_u64, 0, bytes, static_cast<hipStream_t>(stream)) != hipSuccess)
    {
      g_last_error = "gtx_scores_zero: hipMemsetAsync failed";
      return GTX_ERR_HIP;
    }
    return GTX_OK;
  }

  int gtx_scores_free(gtx_ctx * c, gtx_score_buffers * b)
  {
    if (!c || !b)
      return GTX_ERR_ARG;
    if (b->d_stat_u64 && (hipDeviceSynchronize() != hipSuccess || gtx::dev_free(b->d_stat_u64) != hipSuccess)) // (the block is handed out again: nothing may still use it)
    {
      g_last_error = "gtx_scores_free: hipFree failed";
      return GTX_ERR_HIP;
    }
    std::memset(b, 0, sizeof(*b));
    return GTX_OK;
  }

  int gtx_comm_unique_id(void * id128)
  {
    if (!id128)
      return GTX_ERR_ARG;
    if (!rccl_ready())
      return GTX_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (!nccl_ok(rccl().get_unique_id(&id), "ncclGetUniqueId"))
      return GTX_ERR_HIP;
    static_assert(sizeof(id) == GTX_COMM_ID_BYTES, "ncclUniqueId size");
    std::memcpy(id128, &id, sizeof(id));
    return GTX_OK;
  }

  int gtx_comm_init_rank(const void * id128, int n_ranks, int rank, int device, void ** comm)
  {
    if (!id128 || !comm || n_ranks <= 0 || rank < 0 || rank >= n_ranks)
      return GTX_ERR_ARG;
    *comm = nullptr;
    if (!rccl_ready())
      return GTX_ERR_UNSUPPORTED;
    if (hipSetDevice(device) != hipSuccess)
    {
      g_last_error = "gtx_comm_init_rank: hipSetDevice failed";
      return GTX_ERR_NO_DEVICE;
    }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t cm = nullptr;
    if (!nccl_ok(rccl().comm_init_rank(&cm, n_ranks, id, rank), "ncclCommInitRank"))
      return GTX_ERR_HIP;
    *comm = cm;
    return GTX_OK;
  }

  int gtx_comm_destroy(void * comm)
  {
    if (!comm)
      return GTX_OK;
    if (!rccl_ready())
      return GTX_ERR_UNSUPPORTED;
    return nccl_ok(rccl().comm_destroy(static_cast<ncclComm_t>(comm)), "ncclCommDestroy") ? GTX_OK : GTX_ERR_HIP;
  }

  int gtx_scores_reduce(gtx_ctx * c, const gtx_score_buffers * b, void * rccl_comm, void * stream)
  {
    if (!c || !b || !rccl_comm || !b->d_stat_u64 || !b->d_log_score || !b->d_gt_cov || !b->d_hap_u32 || !b->d_stat_u32)
    {
      g_last_error = "gtx_scores_reduce: bad argument";
      return GTX_ERR_ARG;
    }
    if (c->device < 0)
    {
      g_last_error = "context was created without a device (libgtx has no CPU path)";
      return GTX_ERR_NO_DEVICE;
    }
    if (!rccl_ready())
      return GTX_ERR_UNSUPPORTED;
    Rccl & r = rccl();
    ncclComm_t const cm = static_cast<ncclComm_t>(rccl_comm);
    hipStream_t const st = static_cast<hipStream_t>(stream);
    Sections const s = sections_of(*c, b->n_samples);
    // One group = one fused launch.  u64 sums cannot travel as pairs of u32 (carries), so the block is two operations:
    // the u64 statistics and everything else.
    bool ok = nccl_ok(r.group_start(), "ncclGroupStart");
    auto sum = [&](void * p, uint64_t n, ncclDataType_t t)
    {
      if (ok && n != 0)
        ok = nccl_ok(r.all_reduce(p, p, n, t, ncclSum, cm, st), "ncclAllReduce");
    };
    sum(b->d_stat_u64, s.stat_u64, ncclUint64);
    if (is_packed(s, *b))
      sum(b->d_log_score, s.u32_total(), ncclUint32);
    else
    {
      if (b->d_ref_depth)
        sum(b->d_ref_depth, static_cast<uint64_t>(b->n_samples) * (b->ref_depth_len + 1u), ncclUint32);
      sum(b->d_log_score, s.log_score, ncclUint32);
      sum(b->d_gt_cov, s.gt_cov, ncclUint32);
      sum(b->d_hap_u32, s.hap_u32, ncclUint32);
      sum(b->d_stat_u32, s.stat_u32, ncclUint32);
      if (b->d_conn_near)
        sum(b->d_conn_near, s.conn_near, ncclUint32);
    }
    bool const ended = nccl_ok(r.group_end(), "ncclGroupEnd");
    return ok && ended ? GTX_OK : GTX_ERR_HIP;
  }
}
