// gtx_ctx.hpp -- the opaque context behind gtx_ctx* (host copies + device copies of graph and index)
#pragma once
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "gtx_flat.hpp"

namespace gtx
{
// Everything a gtx_align_batch / gtx_score_batch call writes besides the caller's buffers: queues between the passes,
// device counters, workspaces of the HBM-table passes, timing events.  A call owns one CallScratch from entry until its
// last launch is enqueued; the scratch is handed to the next call on the same stream at once (stream order keeps them
// apart) and to a call on another stream once its `done` event has completed.  That is what makes the entry points
// re-entrant: the reference calls align_read / update_haplotype_scores_geno from `jobs` threads at the same time
// (src/typer/caller.cpp:399-436) against one immutable index + graph.
struct CallScratch
{
  bool busy = false;         // a host thread is inside an entry point with it (guarded by gtx_ctx::pool_mutex)
  void * last_stream = nullptr;
  void * submit_stream = nullptr; // the stream the last call that used it was made on
  bool used = false;         // `done` has been recorded at least once
  uint64_t use_seq = 0;      // gtx_ctx::scratch_uses at its last release (the smallest: the one whose work was queued longest ago)
  void * done = nullptr;     // hipEvent_t recorded behind the last launch that uses this scratch
  // per part of the batch, 8 words: [0] read / queue-1 claim counter of pass 1, [1] task counter of pass 2, [2] tasks queued
  // for pass 2, [3] forward tasks the position-hinted pass handed to pass 1, [4] forward tasks pass 1 handed to pass 2,
  // [5] forward tasks pass 2 did (those beyond [4] came straight from the position-hinted pass); [2] and [3] are one 64-bit
  // word for that pass' single add per workgroup
  uint32_t * d_counters = nullptr;
  // TWO sets of them (round 6): a call counts in the set the call before it did not use and zeroes that one, behind its last
  // launch on its last stream, for the call after it -- the reset used to be a hipMemsetAsync in FRONT of the position-hinted pass
  // on the caller's stream: a fill kernel and the gaps around it, 25 us of an idle chip per step of 0.72 ms.  d_counters and
  // the pointers into the set (d_big_state, d_wide_state, d_exact_state, d_span) name the set of the last call.
  static constexpr uint32_t COUNTER_WORDS = 8 * 8 + 48 + 4, COUNTER_PITCH = 128; // (8 words x MAX_PARTS; big, wide, exact x 3: 8 words each; span: 2 x 64 bits)
  uint32_t * d_counter_sets = nullptr; // [2][COUNTER_PITCH]
  uint32_t counter_set = 0;            // the set of the last call
  bool spare_set_clean = true;         // the other one is zero (false after a call that failed on its way)
  bool has_big = false, has_wide = false;
  uint8_t * d_planes = nullptr;  // plane rows of a batch that came as BAM nibbles (gtx_align_batch; grow-only)
  uint64_t planes_cap = 0;
  uint32_t * d_queue1 = nullptr; // reads whose forward task the position-hinted pass declined (grow-only)
  uint64_t queue1_cap = 0;
  uint32_t * d_queue = nullptr;  // (read * 2 + orientation) tasks for pass 2 (grow-only)
  uint64_t queue_cap = 0;
  static constexpr uint32_t MAX_PARTS = 8; // parts a large batch is cut into (their general passes overlap the next part)
  void * side_stream = nullptr;            // hipStream_t of the general / HBM-table passes when a batch has several parts
  void * sync_events[MAX_PARTS + 1] = {};  // hipEvent_t (no timing): part p's express pass done; [MAX_PARTS]: fork / join
  // Pass times of the calls since the last gtx_ctx_kernel_times (a ring: a host that keeps several calls in flight asks
  // once behind them).  Per call and part 6 events: around hinted, express (caller's stream), general (side stream);
  // [0][5] = the call's end.  Created at a slot's first use.
  static constexpr uint32_t TIME_RING = 32;
  void * time_ring[TIME_RING][MAX_PARTS][6] = {};
  uint32_t ring_parts[TIME_RING] = {}; // parts of the call in the slot
  uint32_t ring_used = 0;              // calls recorded in epoch ring_epoch (the first TIME_RING of them are kept)
  uint32_t ring_epoch = 0;             // gtx_ctx::time_epoch of the slots above
  bool timed = false;                  // the last call was timed (its slot: ring_used - 1)
  unsigned long long * d_span = nullptr; // behind d_counters: the position-hinted pass' own clock (gtx_api.hip: GTX_HINTED_PASS), reset with the counters
  unsigned long long * h_span = nullptr; // pinned, [TIME_RING][2]: the spans of the timed calls
  uint32_t timed_reads = 0;
  // HBM-table pass (reads that overflowed the LDS-sized tables)
  uint32_t * d_big_tasks = nullptr;
  uint32_t big_task_cap = 0;
  uint32_t * d_big_state = nullptr; // [0] tasks queued, [1] claim cursor, [3] tasks dropped (list full)
  void * d_big_ws = nullptr;
  uint32_t big_blocks = 0;   // workspaces d_big_ws holds (grown to gtx_ctx::big_blocks by the first large batch)
  // wide-site pass (graphs with a site of more than 64 alleles only): tasks that met an allele number >= 64
  uint32_t * d_wide_tasks = nullptr;
  uint32_t * d_wide_state = nullptr; // inside d_big_state's allocation (same layout)
  void * d_wide_ws = nullptr;
  static constexpr uint32_t WIDE_TASK_CAP = 1u << 20, WIDE_BLOCKS = 64;
  // exact pass (align_core.hpp: namespace exact): tasks that exceeded the tables of the passes above; tables cut out of a slab
  // at run time -- gtx_ctx::exact_parts workgroups with a part of it each, then one workgroup with all of it
  uint32_t * d_exact_tasks = nullptr; // three queues of EXACT_TASK_CAP (small parts, large parts, the whole slab)
  uint32_t * d_exact_state = nullptr; // inside d_big_state's allocation: 8 words per launch (same layout) + 8 for what is left
  static constexpr uint32_t EXACT_TASK_CAP = 1u << 20;
  static constexpr uint32_t EXACT_LARGE_PARTS = 32, EXACT_LARGE_SITES = 64; // the launch between the small parts and the whole slab
  static constexpr uint32_t EXACT_PART_SITES = 24;       // variant sites a path has room for while a task has a small part of the slab
  static constexpr uint32_t EXACT_PART_CANDIDATES = 8256; // ... and walk candidates (128 live sequences x 64 alleles + a round's slack)
  // second scoring pass (items whose reads touch more variant sites than the main pass' tables hold)
  uint32_t * d_score_state = nullptr; // two sets of 4 words, used in turn ([0] items queued, [2] the work queue's count): the triage kernel of a call zeroes the other set
  uint32_t score_set = 0;
  bool score_spare_clean = true;
  uint32_t * d_score_queue = nullptr;
  void * d_score_tables = nullptr;
  uint32_t * d_score_work = nullptr; // [0] number of items the triage kernel found worth scoring, [1..] their indices (grow-only)
  uint32_t score_work_cap = 0;
  // gtx_align_batch_planes_triaged, items that are the batch's reads: a bit per read (a word per wavefront of the position-hinted
  // pass) -- the forward record carries a variant site (grow-only)
  unsigned long long * d_var_masks = nullptr;
  uint64_t var_mask_cap = 0;
};
} // namespace gtx

struct gtx_ctx
{
  gtx_params params{};
  gtx::HostGraph graph;
  gtx::HostIndex index;
  int device = -1; // -1: inspection-only context (no device entry point works)
  int n_cu = 0;
  uint32_t wall_clock_khz = 100000; // the rate of wall_clock64() (hipDeviceAttributeWallClockRate)
  std::vector<void *> dev_allocs;
  bool quiet = false; // set by a caller that knows every launch on this context has completed (ctx_release_device then does not wait for the device)
  std::vector<uint8_t> upload_stage; // host source of the graph tables' one asynchronous copy (ctx_upload); empty once the context is made
  gtx::GraphView dev_graph{};
  gtx::IndexView dev_index{};
  uint32_t * d_error_flag = nullptr;
  int align_blocks_per_cu = 8, express_blocks_per_cu = 16, express4_blocks_per_cu = 8;
  int express4_wide_blocks_per_cu = 8;
  uint32_t score_blocks_per_cu = 8;  // resident 256-thread workgroups of gtx_score_kernel per CU
  bool express4_wide = false; // pass 1 runs gtx_align_express4_wide_kernel (express4_prefers_wide, gtx_flat.hpp)
  uint32_t big_blocks = 0;       // workgroups (= workspaces) of the HBM-table pass for a large batch; a small one gets n_cu (HBM_SMALL_BATCH)
  // What the HBM-table pass of the call before found in its queue, copied to pinned host memory behind that pass (asynchronously:
  // the value read here may be a call or two old): the pass is launched with as many workgroups as that many tasks can use
  // instead of all of them -- nearly every batch queues nothing, and 2 048 wavefronts of 128 registers that only look at an empty
  // queue waited a quarter of a millisecond for room beside the other stream's kernels.  0xFFFFFFFF: nothing seen yet.
  uint32_t * h_big_seen = nullptr;
  static constexpr uint32_t HBM_SMALL_BATCH = 1u << 20; // reads: below this a call takes the small configuration of the passes behind the general one
  bool exact_mb_given = false;   // gtx_params::exact_pass_mb / GTX_EXACT_PASS_MB: every call gets that slab
  bool has_wide_sites = false; // some site has more than 64 alleles: the wide-site passes (alignment, scoring) exist
  // slab of the exact alignment pass (gtx_params::exact_pass_mb / GTX_EXACT_PASS_MB).  The context owns up to EXACT_SLOTS of them,
  // each with an event behind the exact launches of the last call that used it.  A call takes the first slab whose event is
  // through, makes a new one while there are fewer than EXACT_SLOTS, else lets its stream wait for one's event, in turn; it
  // launches on its own stream and records the event again (all under exact_mutex, so that the next call's wait sees this
  // call's record).  A caller that runs one batch at a time uses one slab, three batches in flight three, sixty-four host
  // threads four -- not half a gigabyte per call in flight for a pass that nearly always finds its queue empty.
  uint64_t exact_slab_bytes = 0;
  struct ExactSlot
  {
    uint8_t * slab = nullptr;
    uint64_t bytes = 0;    // (a slot made for a small batch is a quarter of the size: see exact_slab_for)
    void * idle = nullptr; // hipEvent_t behind the slab's last exact launches
  };
  static constexpr int EXACT_SLOTS = 4;
  ExactSlot exact_slot[EXACT_SLOTS];
  int n_exact_slots = 0, next_exact_slot = 0;
  std::mutex exact_mutex;
  uint32_t exact_cand_cap = 0;   // walk candidates a task of that pass can have alive: exact_cand_cap(widest site of the graph)
  uint32_t exact_parts = 0;      // workgroups (= the most parts of the slab) of the pass' first launch
  bool exact_fixed_parts = false; // GTX_EXACT_PARTS (tests): always that many parts
  static constexpr uint32_t SCORE_QUEUE_CAP = 1u << 20, SCORE_BIG_THREADS = 1024;
  // arena for records longer than a record slot: shared by all calls (it only grows; the cursor is a device counter)
  uint32_t * d_big_records = nullptr;
  uint64_t big_record_words = 0;
  unsigned long long * d_arena_cursor = nullptr;
  // streams gtx_pipeline_run made for its host threads: they live as long as the context (scratches and exact-pass slots keep
  // events that were recorded on them -- a destroyed stream behind such an event is an error at the next query), idle ones
  // are taken again by the next run
  std::mutex pipeline_mutex;
  std::vector<void *> pipeline_streams_idle, pipeline_streams_all; // hipStream_t
  // pool of per-call scratch (see CallScratch)
  std::mutex pool_mutex;
  std::vector<std::unique_ptr<gtx::CallScratch>> pool;
  gtx::CallScratch * last_align = nullptr; // scratch of the most recent gtx_align_batch (pass times, second-pass task count)
  uint64_t scratch_uses = 0;
  static constexpr size_t MAX_SCRATCH_IN_FLIGHT = 4; // calls in flight from ONE stream before the next waits for the oldest (gtx_api.hip: scratch_acquire)
  bool timing_armed = false;
  // the timed calls between two queries are one epoch: the first timed call behind a query opens the next one (a scratch
  // drops the slots of an older epoch when it records again; a query reads the current epoch only, as often as it likes)
  uint32_t time_epoch = 0;
  bool epoch_queried = false;
  uint32_t timed_seq = 0;  // calls since the context was made, counted for GTX_TIME_EVERY
  // index facts of a device-built index (gtx_index_dev.hip); its keys / labels in reference order stay on the device and
  // are downloaded into `index` when an inspection entry point asks for them
  uint32_t n_keys = 0, n_labels = 0;
  uint64_t * d_keys = nullptr;
  uint32_t * d_key_off = nullptr;
  gtx_label * d_labels_sorted = nullptr;
  // the index tables the global-lookup passes probe at random (exact slots, half-key slots and buckets, labels): device
  // ranges for gtx_warm_kernel (gtx_api.hip)
  struct Range
  {
    void const * p;
    uint64_t bytes;
  };
  std::vector<Range> lookup_tables;
  std::atomic<bool> index_downloaded{false}; // set (release) by download_index under index_mutex; readers check it first (acquire)
  std::mutex index_mutex;
};

namespace gtx
{
extern thread_local std::string g_last_error;
// gtx_scores_alloc with the block zeroed on `stream` (no wait: for a caller whose first use of the block is on that stream)
int scores_alloc_on(gtx_ctx * c, uint32_t n_samples, uint32_t conn_cap, gtx_score_buffers * out, uint64_t * reduced_bytes, void * stream);
// zeroes the header word of the 2 * n_reads record slots on `stream` (slots recycled from one region to the next)
int records_clear_enqueue(gtx_ctx * c, uint32_t * d_records, uint32_t rec_words, uint64_t n_reads, void * stream);
// gtx_records_failed without its wait: zeroes *d_count and queues the count on `stream`
int records_failed_enqueue(gtx_ctx * c, const uint32_t * d_records, uint32_t rec_words, uint64_t n_reads, void * stream, unsigned long long * d_count);
int ctx_upload(gtx_ctx & c, int device); // graph tables + per-call scratch (no index)
void ctx_release_device(gtx_ctx & c);
int build_index_device(gtx_ctx & c, std::vector<Emit> const & em, std::vector<EmitRun> const & runs); // gtx_index_dev.hip
int download_index(gtx_ctx & c);
int download_hint_table(gtx_ctx const & c, int which, void * out, uint64_t cap_bytes, uint64_t * bytes); // gtx_index_dev.hip
} // namespace gtx
