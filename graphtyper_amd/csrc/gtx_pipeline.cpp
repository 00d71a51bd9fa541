// gtx_pipeline.cpp -- the host loop around the path, inside the library.
//
// gtx_pipeline_run replaces the reference's worker threads over BAM pools (Caller: src/typer/caller.cpp:399-436, each running
// parallel_reader_genotype_only, src/utilities/hts_parallel_reader.cpp:245-338: read a record, filter it, align it, score it):
// T host threads, each with its own group of BAM files, run
//   gtx_reads_next (BGZF members inflated by the library's team, records in the reference's merged order)
//   -> gtx_stream_push (flag filter, duplicate reuse, mate parking; reads leave as plane rows)
//   -> pinned staging -> H2D -> gtx_align_batch_planes -> gtx_score_batch_flags
// on a stream of their own, against ONE context and ONE accumulator block (every per-read effect is an integer addition:
// the block does not depend on which thread scored a read).  Two staging sets per thread: while the device works on one
// batch the thread decodes the next.  What a host language has to add is what follows the loop: gtx_calls_batch,
// gtx_vcf_records.
#include "gtx_ctx.hpp"
#include "gtx_devmem.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
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

struct Worker
{
  gtx_reads * reads = nullptr;
  uint32_t n_samples = 0, n_rg = 0;
  std::vector<uint32_t> sample_of; // the group's sample -> the run's
  bool renumber = false;
  std::vector<std::string> paths;
  double decode = 0, push = 0, enqueue = 0;
  uint64_t records = 0, tasks = 0, items = 0, failed = 0;
  int status = GTX_OK;
  std::string error;
};
} // namespace

extern "C" int gtx_pipeline_run(gtx_ctx * c, const char * const * bam_paths, uint32_t n_paths, uint32_t n_threads, const char * region, uint32_t chunk,
                                uint32_t rec_words, uint64_t record_slots_per_thread, const gtx_score_buffers * acc, gtx_pipeline_stats * stats)
{
  using gtx::g_last_error;
  if (!c || !bam_paths || n_paths == 0 || n_threads == 0 || chunk == 0 || rec_words < 8 || record_slots_per_thread == 0 || !acc)
  {
    g_last_error = "gtx_pipeline_run: bad argument";
    return GTX_ERR_ARG;
  }
  if (c->device < 0)
  {
    g_last_error = "context was created without a device (libgtx has no CPU path)";
    return GTX_ERR_NO_DEVICE;
  }
  auto const t_all = std::chrono::steady_clock::now();
  n_threads = std::min(n_threads, n_paths);
  std::vector<Worker> team(n_threads);
  for (uint32_t f = 0; f < n_paths; ++f)
    team[f % n_threads].paths.push_back(bam_paths[f] ? bam_paths[f] : "");
  // the files are opened first (one thread each): the samples' numbers need every group's names
  {
    std::vector<std::thread> openers;
    for (Worker & w : team)
      openers.emplace_back([&w, region]
      {
        std::vector<char const *> p;
        for (auto const & s : w.paths)
          p.push_back(s.c_str());
        w.status = gtx_reads_open(p.data(), static_cast<uint32_t>(p.size()), region, &w.reads);
        if (w.status != GTX_OK)
          w.error = gtx_last_error();
        else
          gtx_reads_info(w.reads, &w.n_samples, &w.n_rg);
      });
    for (auto & t : openers)
      t.join();
  }
  int status = GTX_OK;
  uint32_t samples = 0;
  std::vector<std::string> names;
  for (Worker & w : team)
  {
    if (w.status != GTX_OK && status == GTX_OK)
    {
      status = w.status;
      g_last_error = w.error;
    }
    // samples are numbered by name in the order the groups bring them (position-sliced files of one sample are one sample)
    for (uint32_t i = 0; w.reads && i < w.n_samples; ++i)
    {
      char const * nm = gtx_reads_sample_name(w.reads, i);
      std::string const name = nm ? nm : "";
      auto it = std::find(names.begin(), names.end(), name);
      w.sample_of.push_back(static_cast<uint32_t>(it - names.begin()));
      if (it == names.end())
        names.push_back(name);
      w.renumber = w.renumber || w.sample_of.back() != i;
    }
    samples = static_cast<uint32_t>(names.size());
  }
  if (status == GTX_OK && samples > acc->n_samples)
  {
    status = GTX_ERR_ARG;
    g_last_error = "gtx_pipeline_run: the files hold " + std::to_string(samples) + " samples, the accumulator block " + std::to_string(acc->n_samples);
  }
  // (the context's and the block's counters of refused work before the run: what the run adds is the run's)
  uint32_t refused_before = 0, conn_dropped_before = 0;
  if (status == GTX_OK)
  {
    uint32_t conn[2] = {0, 0};
    (void)gtx_ctx_error_count(c, &refused_before);
    if (acc->d_conn_count && hipSetDevice(c->device) == hipSuccess && hipMemcpy(conn, acc->d_conn_count, sizeof conn, hipMemcpyDeviceToHost) == hipSuccess)
      conn_dropped_before = conn[1];
  }
  std::atomic<uint32_t> ready{0}, go{0};
  auto run = [&](Worker & w)
  {
    auto fail = [&](int st, std::string const & what)
    {
      w.status = st;
      w.error = what;
    };
    if (hipSetDevice(c->device) != hipSuccess)
    {
      fail(GTX_ERR_HIP, "hipSetDevice");
      ready.fetch_add(1); // (the others wait for every thread to be counted)
      return;
    }
    hipStream_t st = nullptr;
    hipEvent_t done[2] = {nullptr, nullptr};
    uint8_t * pin_seq[2] = {nullptr, nullptr};
    gtx_read_meta * pin_meta[2] = {nullptr, nullptr};
    gtx_score_item * pin_items[2] = {nullptr, nullptr};
    void *dev_seq[2] = {nullptr, nullptr}, *dev_meta[2] = {nullptr, nullptr}, *dev_items[2] = {nullptr, nullptr}, *d_rec = nullptr, *d_fl = nullptr;
    gtx_stream * push = nullptr;
    uint32_t const stride = 80; // plane rows of reads of up to 160 bases
    {
      std::lock_guard<std::mutex> lock(c->pipeline_mutex);
      if (!c->pipeline_streams_idle.empty())
      {
        st = static_cast<hipStream_t>(c->pipeline_streams_idle.back());
        c->pipeline_streams_idle.pop_back();
      }
    }
    bool ok = st != nullptr;
    if (!ok && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess)
    {
      ok = true;
      std::lock_guard<std::mutex> lock(c->pipeline_mutex);
      c->pipeline_streams_all.push_back(st);
    }
    for (int b = 0; b < 2 && ok; ++b)
      ok = hipEventCreateWithFlags(&done[b], hipEventDisableTiming) == hipSuccess &&
           hipHostMalloc(reinterpret_cast<void **>(&pin_seq[b]), static_cast<size_t>(chunk) * stride) == hipSuccess &&
           hipHostMalloc(reinterpret_cast<void **>(&pin_meta[b]), static_cast<size_t>(chunk) * sizeof(gtx_read_meta)) == hipSuccess &&
           hipHostMalloc(reinterpret_cast<void **>(&pin_items[b]), static_cast<size_t>(chunk) * sizeof(gtx_score_item)) == hipSuccess &&
           gtx::dev_malloc(&dev_seq[b], static_cast<size_t>(chunk) * stride) == hipSuccess &&
           gtx::dev_malloc(&dev_meta[b], static_cast<size_t>(chunk) * sizeof(gtx_read_meta)) == hipSuccess &&
           gtx::dev_malloc(&dev_items[b], static_cast<size_t>(chunk) * sizeof(gtx_score_item)) == hipSuccess;
    size_t const rec_bytes = static_cast<size_t>(record_slots_per_thread) * 2 * rec_words * 4, fl_bytes = static_cast<size_t>(record_slots_per_thread) * 2;
    // (the record slots are large and live for one run: straight from the driver, not through the library's cache of freed blocks --
    //  sixteen of them would fill it and turn every later context's small allocations into driver calls)
    ok = ok && hipMalloc(&d_rec, rec_bytes) == hipSuccess && hipMalloc(&d_fl, fl_bytes) == hipSuccess &&
         hipMemsetAsync(d_rec, 0, rec_bytes, st) == hipSuccess && hipMemsetAsync(d_fl, 0, fl_bytes, st) == hipSuccess;
    if (!ok)
      fail(GTX_ERR_HIP, "gtx_pipeline_run: could not allocate a thread's staging buffers / record slots");
    if (ok && gtx_stream_create(&c->params, std::max(1u, w.n_rg), &push) != GTX_OK)
    {
      ok = false;
      fail(GTX_ERR_ARG, gtx_last_error());
    }
    if (ok)
      gtx_stream_set_planes(push, stride);
    std::vector<gtx_stream_record> recs(chunk);
    std::vector<uint8_t> seq(static_cast<size_t>(chunk) * stride);
    if (st)
      (void)hipStreamSynchronize(st);
    // every thread has its buffers: the loop's clock starts when the last one gets here
    ready.fetch_add(1);
    while (go.load(std::memory_order_acquire) == 0)
      std::this_thread::yield();
    bool used[2] = {false, false};
    uint64_t at = 0; // tasks of this thread so far: the stream numbers them over all its records
    auto submit = [&](int b, uint32_t na, uint32_t ni) -> bool
    {
      if ((na && (hipMemcpyAsync(dev_seq[b], pin_seq[b], static_cast<size_t>(na) * stride, hipMemcpyHostToDevice, st) != hipSuccess ||
                  hipMemcpyAsync(dev_meta[b], pin_meta[b], static_cast<size_t>(na) * sizeof(gtx_read_meta), hipMemcpyHostToDevice, st) != hipSuccess)) ||
          (ni && hipMemcpyAsync(dev_items[b], pin_items[b], static_cast<size_t>(ni) * sizeof(gtx_score_item), hipMemcpyHostToDevice, st) != hipSuccess))
      {
        fail(GTX_ERR_HIP, "gtx_pipeline_run: host to device copy");
        return false;
      }
      int rc = GTX_OK;
      if (na)
        rc = gtx_align_batch_planes(c, static_cast<uint8_t const *>(dev_seq[b]), stride, static_cast<gtx_read_meta const *>(dev_meta[b]), na,
                                    static_cast<uint32_t *>(d_rec) + at * 2 * rec_words, rec_words, static_cast<uint8_t *>(d_fl) + at * 2, st);
      if (rc == GTX_OK && ni)
        rc = gtx_score_batch_flags(c, static_cast<gtx_score_item const *>(dev_items[b]), ni, static_cast<uint32_t const *>(d_rec), rec_words,
                                   static_cast<uint8_t const *>(d_fl), acc, st);
      if (rc != GTX_OK)
      {
        fail(rc, gtx_last_error());
        return false;
      }
      (void)hipEventRecord(done[b], st);
      used[b] = true;
      at += na;
      w.tasks += na;
      w.items += ni;
      return true;
    };
    for (int b = 0; ok; b ^= 1)
    {
      auto t0 = std::chrono::steady_clock::now();
      uint32_t n = 0;
      int rc = gtx_reads_next(w.reads, recs.data(), seq.data(), stride, chunk, &n);
      w.decode += seconds_since(t0);
      if (rc != GTX_OK)
      {
        fail(rc, gtx_last_error());
        break;
      }
      if (n == 0)
        break;
      if (w.renumber)
        for (uint32_t i = 0; i < n; ++i)
          recs[i].sample = w.sample_of[recs[i].sample];
      t0 = std::chrono::steady_clock::now();
      if (used[b])
        (void)hipEventSynchronize(done[b]); // the staging set is free again
      uint32_t na = 0, ni = 0;
      rc = gtx_stream_push(push, recs.data(), seq.data(), stride, n, pin_seq[b], pin_meta[b], chunk, &na, pin_items[b], chunk, &ni);
      w.push += seconds_since(t0);
      if (rc != GTX_OK)
      {
        fail(rc, gtx_last_error());
        break;
      }
      if (at + na > record_slots_per_thread)
      {
        fail(GTX_ERR_CAPACITY, "gtx_pipeline_run: more reads to align in a thread's files than record_slots_per_thread");
        break;
      }
      t0 = std::chrono::steady_clock::now();
      if (!submit(b, na, ni))
        break;
      w.enqueue += seconds_since(t0);
      w.records += n;
    }
    if (w.status == GTX_OK && push)
    {
      // the end of the stream: reads still waiting for their mate (SV calling scores them on their own)
      uint64_t n_rec = 0, n_dup = 0, n_parked = 0;
      gtx_stream_counts(push, &n_rec, &n_dup, &n_parked);
      std::vector<gtx_score_item> left(std::max<uint64_t>(n_parked, 1));
      uint32_t ni = 0;
      int const rc_fin = gtx_stream_finish(push, left.data(), static_cast<uint32_t>(left.size()), &ni);
      if (rc_fin != GTX_OK) // (parked mates left unscored are a wrong result, not a detail)
        w.status = rc_fin;
      else if (ni)
      {
        if (used[0] && hipEventSynchronize(done[0]) != hipSuccess)
          w.status = GTX_ERR_HIP;
        for (uint32_t o = 0; o < ni && w.status == GTX_OK; o += chunk)
        {
          uint32_t const m = std::min(chunk, ni - o);
          std::memcpy(pin_items[0], left.data() + o, static_cast<size_t>(m) * sizeof(gtx_score_item));
          if (submit(0, 0, m) && hipEventSynchronize(done[0]) != hipSuccess)
            w.status = GTX_ERR_HIP;
        }
      }
    }
    if (st && hipStreamSynchronize(st) != hipSuccess && w.status == GTX_OK)
      w.status = GTX_ERR_HIP;
    // (records that are a table-overflow status instead of a result: reads the accumulators lack)
    if (w.status == GTX_OK && at && d_rec)
    {
      uint64_t failed = 0;
      int const rc = gtx_records_failed(c, static_cast<uint32_t const *>(d_rec), rec_words, at, st, &failed);
      if (rc != GTX_OK)
        fail(rc, gtx_last_error());
      w.failed = failed;
    }
    if (push)
      gtx_stream_destroy(push);
    for (int b = 0; b < 2; ++b)
    {
      if (done[b])
        (void)hipEventDestroy(done[b]);
      (void)hipHostFree(pin_seq[b]);
      (void)hipHostFree(pin_meta[b]);
      (void)hipHostFree(pin_items[b]);
      (void)gtx::dev_free(dev_seq[b]);
      (void)gtx::dev_free(dev_meta[b]);
      (void)gtx::dev_free(dev_items[b]);
    }
    (void)hipFree(d_rec);
    (void)hipFree(d_fl);
    if (st) // (kept for the context's life: see gtx_ctx::pipeline_streams_all)
    {
      std::lock_guard<std::mutex> lock(c->pipeline_mutex);
      c->pipeline_streams_idle.push_back(st);
    }
  };
  double t_loop = 0;
  if (status == GTX_OK)
  {
    std::vector<std::thread> threads;
    for (Worker & w : team)
      threads.emplace_back([&run, &w] { run(w); });
    while (ready.load() < n_threads)
      std::this_thread::yield();
    auto const t0 = std::chrono::steady_clock::now();
    go.store(1, std::memory_order_release);
    for (auto & t : threads)
      t.join();
    t_loop = seconds_since(t0);
  }
  gtx_pipeline_stats s{};
  for (Worker & w : team)
  {
    if (w.reads)
      gtx_reads_close(w.reads);
    if (w.status != GTX_OK && status == GTX_OK)
    {
      status = w.status;
      g_last_error = w.error;
    }
    s.records_failed += w.failed;
    s.records += w.records;
    s.tasks += w.tasks;
    s.items += w.items;
    s.decode_s += w.decode;
    s.push_s += w.push;
    s.enqueue_s += w.enqueue;
    s.slowest_thread_s = std::max(s.slowest_thread_s, w.decode + w.push + w.enqueue);
  }
  // What a capacity limit of the library dropped makes the block a wrong result, not a smaller one: records with a table-overflow
  // status, score items over more sites than the scorer's table, far-pair connections beyond the log.  (acc is then undefined.)
  if (status == GTX_OK && c->device >= 0)
  {
    uint32_t refused = 0, conn[2] = {0, 0};
    (void)gtx_ctx_error_count(c, &refused);
    if (acc->d_conn_count)
      (void)hipMemcpy(conn, acc->d_conn_count, sizeof conn, hipMemcpyDeviceToHost);
    s.score_items_refused = refused >= refused_before ? refused - refused_before : refused;
    s.connections_dropped = conn[1] >= conn_dropped_before ? conn[1] - conn_dropped_before : conn[1];
    if (s.records_failed || s.score_items_refused || s.connections_dropped)
    {
      status = GTX_ERR_CAPACITY;
      g_last_error = "gtx_pipeline_run: the result is incomplete -- " + std::to_string(s.records_failed) + " records with a table-overflow status, " +
                     std::to_string(s.score_items_refused) + " score items refused, " + std::to_string(s.connections_dropped) +
                     " connections beyond the log (gtx_params.exact_pass_mb / big_record_words, gtx_scores_alloc's conn_cap)";
    }
  }
  s.n_samples = samples;
  s.n_threads = n_threads;
  s.loop_s = t_loop;
  s.wall_s = seconds_since(t_all);
  if (stats)
    *stats = s;
  return status;
}
