// gtx_regions.cpp -- region after region, inside the library.
//
// gtx_regions_run replaces the loop of genotype_regions (src/utilities/genotype.cpp:735-738: "Genotype regions serially", every
// region a call of genotype(), :406-604) for the last iteration of a region -- the one whose graph is made of the variant records
// the iterations before have agreed on:
//   records -> gtx_graph_build -> gtx_ctx_create (flatten + index; builder threads, several regions ahead)
//           -> gtx_align_batch_planes -> gtx_score_batch_flags -> gtx_calls_batch -> accumulators to the host (device threads,
//              a stream each)
//           -> gtx_vcf_records (text threads)
// The reference runs the regions one after the other with its threads inside a region; a region's device work here is a third of
// a millisecond and what the host does around it three times that, so the stages of DIFFERENT regions overlap instead: each
// region is a job that goes through the three stages on whichever thread of the stage is free.  Every job's text depends on that
// job's inputs only (tests/test_regions_run.py: the same bytes as the stages called one by one).
#include "gtx_ctx.hpp"
#include "gtx_devmem.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace
{
double seconds_since(std::chrono::steady_clock::time_point t0)
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// what a job carries from one stage to the next
struct Built
{
  uint32_t job = 0;
  gtx_ctx * ctx = nullptr;
};

struct Scored
{
  uint32_t job = 0;
  gtx_ctx * ctx = nullptr;
  std::vector<uint32_t> gt_cov, stat_u32;
  std::vector<uint64_t> stat_u64;
  std::vector<uint8_t> phred;
  std::vector<gtx_sample_call> calls;
};

// a bounded queue between two stages; close() after the producers are done lets the consumers run dry and leave
template <class T>
class Channel
{
  std::mutex m;
  std::condition_variable not_empty, not_full;
  std::deque<T> q;
  size_t cap;
  bool closed = false;

public:
  explicit Channel(size_t cap_) : cap(cap_) {}
  void put(T && v)
  {
    std::unique_lock<std::mutex> lock(m);
    not_full.wait(lock, [&] { return q.size() < cap; });
    q.push_back(std::move(v));
    not_empty.notify_one();
  }
  bool get(T & out)
  {
    std::unique_lock<std::mutex> lock(m);
    not_empty.wait(lock, [&] { return !q.empty() || closed; });
    if (q.empty())
      return false;
    out = std::move(q.front());
    q.pop_front();
    not_full.notify_one();
    return true;
  }
  void close()
  {
    std::lock_guard<std::mutex> lock(m);
    closed = true;
    not_empty.notify_all();
  }
};

// a stage thread ran out of host memory (or anything else was thrown) in the middle of a job: the job fails with a message that
// needs no allocation to speak of, the thread goes on with the next one -- nothing may cross the C boundary or end the process
void fail_thrown(gtx_region_job & j, std::mutex & m, std::string & first_error, int & first_status) noexcept
{
  j.status = GTX_ERR_CAPACITY;
  try
  {
    std::lock_guard<std::mutex> lock(m);
    if (first_status == GTX_OK)
    {
      first_status = GTX_ERR_CAPACITY;
      first_error.assign("gtx_regions_run: a job failed for lack of host memory");
    }
  }
  catch (...)
  {
    first_status = GTX_ERR_CAPACITY;
  }
}

void fail(gtx_region_job & j, int status, std::string const & what, std::mutex & m, std::string & first_error, int & first_status)
{
  j.status = status;
  std::lock_guard<std::mutex> lock(m);
  if (first_status == GTX_OK)
  {
    first_status = status;
    first_error = what;
  }
}
} // namespace

extern "C" int gtx_regions_run(gtx_region_job * jobs, uint32_t n_jobs, const gtx_params * params, int device, const char * contig,
                               const char * const * sample_names, uint32_t n_samples, uint32_t rec_words, uint32_t conn_cap,
                               uint32_t n_builders, uint32_t n_device_threads, uint32_t n_text_threads, gtx_regions_stats * stats)
{
  using gtx::g_last_error;
  if ((!jobs && n_jobs) || !params || !contig || n_samples == 0 || !sample_names || rec_words < 8)
  {
    g_last_error = "gtx_regions_run: bad argument";
    return GTX_ERR_ARG;
  }
  if (device < 0)
  {
    g_last_error = "gtx_regions_run: no device given (libgtx has no CPU path)";
    return GTX_ERR_NO_DEVICE;
  }
  auto const t_all = std::chrono::steady_clock::now();
  n_builders = std::max(1u, std::min(n_builders ? n_builders : 4u, std::max(1u, n_jobs)));
  n_device_threads = std::max(1u, std::min(n_device_threads ? n_device_threads : 2u, std::max(1u, n_jobs)));
  n_text_threads = std::max(1u, std::min(n_text_threads ? n_text_threads : 3u, std::max(1u, n_jobs)));
  uint64_t max_reads = 0;
  for (uint32_t k = 0; k < n_jobs; ++k)
  {
    gtx_region_job & j = jobs[k];
    j.text = nullptr;
    j.text_len = 0;
    j.status = GTX_OK;
    if (!j.reference || (j.n_records && !j.records) || (j.n_reads && (!j.d_planes || !j.d_meta)) || (j.n_items && !j.d_items) || j.n_reads > 0x7FFFFFFFull ||
        j.n_items > 0x7FFFFFFFull)
    {
      g_last_error = "gtx_regions_run: job " + std::to_string(k) + " lacks an input";
      return GTX_ERR_ARG;
    }
    max_reads = std::max(max_reads, j.n_reads);
  }
  std::mutex err_m;
  std::string first_error;
  int first_status = GTX_OK;
  std::atomic<uint32_t> next_job{0};
  Channel<Built> built(2 * n_builders);
  Channel<Scored> scored(2 * n_text_threads);
  std::mutex stat_m;
  gtx_regions_stats s{};

  auto builder = [&]
  {
    double t_graph = 0, t_ctx = 0;
    for (;;)
    {
      uint32_t const k = next_job.fetch_add(1);
      if (k >= n_jobs)
        break;
      gtx_region_job & j = jobs[k];
      gtx_ctx * c = nullptr;
      try
      {
      auto t0 = std::chrono::steady_clock::now();
      gtx_graph * g = nullptr;
      int rc = gtx_graph_build(j.reference, j.reference_len, j.region_begin, j.region_end, j.records, j.n_records, j.add_all_variants, params->is_sv_graph, 0, &g);
      gtx_graph_view view{};
      if (rc == GTX_OK)
        rc = gtx_graph_get_view(g, &view);
      t_graph += seconds_since(t0);
      t0 = std::chrono::steady_clock::now();
      if (rc == GTX_OK)
        rc = gtx_ctx_create(&view, params, device, &c);
      if (g)
        gtx_graph_destroy(g); // (the context holds its own flat copy)
      t_ctx += seconds_since(t0);
      if (rc != GTX_OK)
      {
        fail(j, rc, "gtx_regions_run: job " + std::to_string(k) + ": " + gtx_last_error(), err_m, first_error, first_status);
        continue;
      }
      Built b;
      b.job = k;
      b.ctx = c;
      built.put(std::move(b));
      c = nullptr; // (the device stage's from here on)
      }
      catch (...)
      {
        if (c)
          gtx_ctx_destroy(c);
        fail_thrown(j, err_m, first_error, first_status);
      }
    }
    std::lock_guard<std::mutex> lock(stat_m);
    s.graph_build_s += t_graph;
    s.ctx_create_s += t_ctx;
  };

  auto device_worker = [&]
  {
    double t_dev = 0;
    uint64_t failed_total = 0, refused_total = 0, dropped_total = 0;
    hipStream_t st = nullptr;
    void *d_rec = nullptr, *d_fl = nullptr;
    size_t const rec_bytes = static_cast<size_t>(std::max<uint64_t>(max_reads, 1)) * 2 * rec_words * 4, fl_bytes = static_cast<size_t>(std::max<uint64_t>(max_reads, 1)) * 2;
    void * d_failed_v = nullptr;
    uint8_t * pinned = nullptr;
    size_t pinned_cap = 0;
    // (the record slots are recycled from region to region: a call writes the record and the flag byte of every task it is
    //  given and a region's items name that region's tasks only -- but the reverse slot of a GTX_FLAG_FORWARD_ONLY read is no task,
    //  and the count of failed records looks at every slot: the header words are zeroed on the stream in front of every region,
    //  so that a table-overflow status one region's read left there is not counted against the regions behind it)
    bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
              gtx::dev_malloc(&d_rec, rec_bytes) == hipSuccess && gtx::dev_malloc(&d_fl, fl_bytes) == hipSuccess &&
              gtx::dev_malloc(&d_failed_v, 8) == hipSuccess && hipMemsetAsync(d_rec, 0, rec_bytes, st) == hipSuccess &&
              hipMemsetAsync(d_fl, 0, fl_bytes, st) == hipSuccess;
    unsigned long long * const d_failed = static_cast<unsigned long long *>(d_failed_v);
    Built b;
    while (built.get(b))
    {
      gtx_region_job & j = jobs[b.job];
      gtx_ctx * c = b.ctx;
      void *d_phred = nullptr, *d_calls = nullptr;
      gtx_score_buffers acc{};
      bool blocks_held = true; // (d_phred, d_calls and the accumulator block are this iteration's until they went back to the cache)
      try
      {
      auto const t0 = std::chrono::steady_clock::now();
      Scored out;
      out.job = b.job;
      out.ctx = c;
      int rc = ok ? GTX_OK : GTX_ERR_HIP;
      std::string what = ok ? "" : "gtx_regions_run: a device thread could not get its stream / record slots";
      gtx_score_layout lay{};
      if (rc == GTX_OK)
        rc = gtx_ctx_score_layout(c, &lay);
      uint64_t const n_phred = static_cast<uint64_t>(n_samples) * lay.total_tri, n_calls = static_cast<uint64_t>(n_samples) * lay.n_hap;
      // Everything of a region is queued on the thread's stream and waited for ONCE: the accumulator block zeroed on the stream, the
      // three launches' work, the count of records that are a table-overflow status, and the results -- into one pinned buffer (a copy
      // into pageable memory waits for the device before it returns: six of them, the overflow count's own wait and the error flag's
      // were eight round trips per region; a device thread spent 1.2 ms on a region whose kernels take 0.3).
      if (rc == GTX_OK && (rc = gtx::scores_alloc_on(c, n_samples, conn_cap, &acc, nullptr, st)) != GTX_OK)
        what = gtx_last_error();
      if (rc == GTX_OK && (gtx::dev_malloc(&d_phred, std::max<uint64_t>(n_phred, 1)) != hipSuccess ||
                           gtx::dev_malloc(&d_calls, std::max<uint64_t>(n_calls, 1) * sizeof(gtx_sample_call)) != hipSuccess))
      {
        rc = GTX_ERR_HIP;
        what = "gtx_regions_run: device memory for the calls";
      }
      if (rc == GTX_OK && j.n_reads && hipMemsetAsync(d_fl, 0, static_cast<size_t>(j.n_reads) * 2, st) != hipSuccess) // (the side bytes: two per read, all of them looked at)
      {
        rc = GTX_ERR_HIP;
        what = "gtx_regions_run: hipMemsetAsync";
      }
      if (rc == GTX_OK && j.n_reads && (rc = gtx::records_clear_enqueue(c, static_cast<uint32_t *>(d_rec), rec_words, j.n_reads, st)) != GTX_OK)
        what = gtx_last_error();
      if (rc == GTX_OK && j.n_reads &&
          (rc = gtx_align_batch_planes(c, j.d_planes, j.plane_stride, j.d_meta, static_cast<uint32_t>(j.n_reads), static_cast<uint32_t *>(d_rec), rec_words,
                                       static_cast<uint8_t *>(d_fl), st)) != GTX_OK)
        what = gtx_last_error();
      if (rc == GTX_OK && j.n_items &&
          (rc = gtx_score_batch_flags(c, j.d_items, static_cast<uint32_t>(j.n_items), static_cast<uint32_t const *>(d_rec), rec_words, static_cast<uint8_t const *>(d_fl), &acc,
                                      st)) != GTX_OK)
        what = gtx_last_error();
      if (rc == GTX_OK && (rc = gtx_calls_batch(c, &acc, static_cast<uint8_t *>(d_phred), static_cast<gtx_sample_call *>(d_calls), st)) != GTX_OK)
        what = gtx_last_error();
      if (rc == GTX_OK && j.n_reads && (rc = gtx::records_failed_enqueue(c, static_cast<uint32_t const *>(d_rec), rec_words, j.n_reads, st, d_failed)) != GTX_OK)
        what = gtx_last_error();
      if (rc == GTX_OK)
      {
        size_t const b_cov = static_cast<size_t>(n_samples) * lay.total_allele * 4, b_u64 = (static_cast<size_t>(lay.n_hap) + 2ull * lay.total_allele) * 8,
                     b_u32 = (static_cast<size_t>(lay.n_hap) + 6ull * lay.total_allele) * 4, b_phred = n_phred, b_calls = n_calls * sizeof(gtx_sample_call);
        auto up = [](size_t x) { return (x + 63) & ~static_cast<size_t>(63); };
        size_t const o_cov = 64, o_u64 = o_cov + up(b_cov), o_u32 = o_u64 + up(b_u64), o_phred = o_u32 + up(b_u32), o_calls = o_phred + up(b_phred),
                     total = o_calls + up(b_calls);
        if (total > pinned_cap)
        {
          if (pinned)
            (void)hipHostFree(pinned);
          pinned = nullptr;
          pinned_cap = 0;
          if (hipHostMalloc(reinterpret_cast<void **>(&pinned), 2 * total) == hipSuccess)
            pinned_cap = 2 * total;
        }
        // ([0..1] connections appended / dropped, [2] the context's error flag, [4..5] the failed-record count)
        auto down = [&](size_t at, void const * src, size_t bytes) { return bytes == 0 || hipMemcpyAsync(pinned + at, src, bytes, hipMemcpyDeviceToHost, st) == hipSuccess; };
        if (!pinned || !down(0, acc.d_conn_count, 8) || !down(8, c->d_error_flag, 4) || !(j.n_reads == 0 || down(16, d_failed, 8)) || !down(o_cov, acc.d_gt_cov, b_cov) ||
            !down(o_u64, acc.d_stat_u64, b_u64) || !down(o_u32, acc.d_stat_u32, b_u32) || !down(o_phred, d_phred, b_phred) || !down(o_calls, d_calls, b_calls) ||
            hipStreamSynchronize(st) != hipSuccess)
        {
          rc = GTX_ERR_HIP;
          what = "gtx_regions_run: device to host copy";
        }
        else
        {
          out.gt_cov.resize(b_cov / 4);
          out.stat_u64.resize(b_u64 / 8);
          out.stat_u32.resize(b_u32 / 4);
          out.phred.resize(b_phred);
          out.calls.resize(n_calls);
          std::memcpy(out.gt_cov.data(), pinned + o_cov, b_cov);
          std::memcpy(out.stat_u64.data(), pinned + o_u64, b_u64);
          std::memcpy(out.stat_u32.data(), pinned + o_u32, b_u32);
          std::memcpy(out.phred.data(), pinned + o_phred, b_phred);
          std::memcpy(out.calls.data(), pinned + o_calls, b_calls);
          // what a capacity limit dropped makes the text a wrong result: the job fails instead
          uint32_t conn[2], refused = 0;
          uint64_t failed = 0;
          std::memcpy(conn, pinned, 8);
          std::memcpy(&refused, pinned + 8, 4);
          if (j.n_reads)
            std::memcpy(&failed, pinned + 16, 8);
          if (failed || refused || conn[1])
          {
            rc = GTX_ERR_CAPACITY;
            what = "gtx_regions_run: job " + std::to_string(b.job) + " is incomplete -- " + std::to_string(failed) + " records with a table-overflow status, " +
                   std::to_string(refused) + " score items refused, " + std::to_string(conn[1]) + " connections beyond the log";
            failed_total += failed;
            refused_total += refused;
            dropped_total += conn[1];
          }
        }
      }
      if (st)
        (void)hipStreamSynchronize(st); // (nothing of this region may still run when its blocks go back to the cache)
      (void)gtx::dev_free(d_phred);
      (void)gtx::dev_free(d_calls);
      (void)gtx::dev_free(acc.d_stat_u64);
      blocks_held = false;
      t_dev += seconds_since(t0);
      if (rc != GTX_OK)
      {
        gtx_ctx_destroy(c);
        c = nullptr;
        fail(j, rc, what, err_m, first_error, first_status);
        continue;
      }
      scored.put(std::move(out));
      c = nullptr; // (the text stage's from here on)
      }
      catch (...)
      {
        if (st)
          (void)hipStreamSynchronize(st);
        if (blocks_held)
        {
          (void)gtx::dev_free(d_phred);
          (void)gtx::dev_free(d_calls);
          (void)gtx::dev_free(acc.d_stat_u64);
        }
        if (c)
          gtx_ctx_destroy(c);
        fail_thrown(j, err_m, first_error, first_status);
      }
    }
    if (st)
    {
      (void)hipStreamSynchronize(st);
      (void)hipStreamDestroy(st);
    }
    (void)gtx::dev_free(d_rec);
    (void)gtx::dev_free(d_fl);
    (void)gtx::dev_free(d_failed_v);
    if (pinned)
      (void)hipHostFree(pinned);
    std::lock_guard<std::mutex> lock(stat_m);
    s.device_s += t_dev;
    s.records_failed += failed_total;
    s.score_items_refused += refused_total;
    s.connections_dropped += dropped_total;
  };

  auto texter = [&]
  {
    double t_text = 0;
    (void)hipSetDevice(device); // (gtx_ctx_destroy gives the context's device memory back)
    Scored sc;
    while (scored.get(sc))
    {
      gtx_region_job & j = jobs[sc.job];
      char * text = nullptr;
      bool ctx_held = true;
      try
      {
      auto const t0 = std::chrono::steady_clock::now();
      gtx_vcf_request rq{};
      rq.contig = contig;
      rq.sample_names = sample_names;
      rq.n_samples = n_samples;
      rq.region_begin = j.vcf_begin;
      rq.region_end = j.vcf_end;
      rq.filter_zero_qual = j.filter_zero_qual;
      rq.gt_cov = sc.gt_cov.data();
      rq.stat_u64 = sc.stat_u64.data();
      rq.stat_u32 = sc.stat_u32.data();
      rq.phred = sc.phred.data();
      rq.calls = sc.calls.data();
      gtx_score_layout lay{};
      (void)gtx_ctx_score_layout(sc.ctx, &lay);
      uint64_t cap = 4096 + static_cast<uint64_t>(lay.n_hap) * (600 + 40ull * n_samples), len = 0;
      text = static_cast<char *>(std::malloc(cap));
      int rc = text ? gtx_vcf_records(sc.ctx, &rq, text, cap, &len) : GTX_ERR_CAPACITY;
      if (rc == GTX_OK && len > cap) // (the estimate was short: once more with what it takes)
      {
        std::free(text);
        cap = len;
        text = static_cast<char *>(std::malloc(cap));
        rc = text ? gtx_vcf_records(sc.ctx, &rq, text, cap, &len) : GTX_ERR_CAPACITY;
      }
      sc.ctx->quiet = true; // (its device thread waited for the stream the region ran on)
      gtx_ctx_destroy(sc.ctx);
      ctx_held = false;
      t_text += seconds_since(t0);
      if (rc != GTX_OK)
      {
        bool const had_text = text != nullptr;
        std::free(text);
        text = nullptr;
        fail(j, rc, "gtx_regions_run: job " + std::to_string(sc.job) + ": " + (had_text ? gtx_last_error() : "out of memory"), err_m, first_error, first_status);
        continue;
      }
      j.text = text;
      j.text_len = len;
      }
      catch (...)
      {
        std::free(text);
        if (ctx_held)
        {
          sc.ctx->quiet = true;
          gtx_ctx_destroy(sc.ctx);
        }
        fail_thrown(j, err_m, first_error, first_status);
      }
    }
    std::lock_guard<std::mutex> lock(stat_m);
    s.vcf_text_s += t_text;
  };

  // (a stage runs with the threads that could be started; a stage without any thread ends the call -- the stages before it are
  //  not started, so nobody waits for room in a queue nobody empties)
  std::vector<std::thread> builders, devs, texters;
  auto start = [](std::vector<std::thread> & v, uint32_t n, auto & body)
  {
    try
    {
      v.reserve(n);
      for (uint32_t t = 0; t < n; ++t)
        v.emplace_back(body);
    }
    catch (...)
    {
    }
    return !v.empty();
  };
  bool const started = start(texters, n_text_threads, texter) && start(devs, n_device_threads, device_worker) && start(builders, n_builders, builder);
  if (!started)
  {
    for (uint32_t k = 0; k < n_jobs; ++k)
      jobs[k].status = GTX_ERR_CAPACITY;
    first_status = GTX_ERR_CAPACITY;
    first_error = "gtx_regions_run: the stage threads could not be started";
  }
  for (auto & t : builders)
    t.join();
  built.close();
  for (auto & t : devs)
    t.join();
  scored.close();
  for (auto & t : texters)
    t.join();
  s.n_builders = n_builders;
  s.n_device_threads = n_device_threads;
  s.n_text_threads = n_text_threads;
  s.wall_s = seconds_since(t_all);
  if (stats)
    *stats = s;
  if (first_status != GTX_OK)
    g_last_error = first_error;
  return first_status;
}

extern "C" void gtx_regions_free(gtx_region_job * jobs, uint32_t n_jobs)
{
  for (uint32_t k = 0; jobs && k < n_jobs; ++k)
  {
    std::free(jobs[k].text);
    jobs[k].text = nullptr;
    jobs[k].text_len = 0;
  }
}
