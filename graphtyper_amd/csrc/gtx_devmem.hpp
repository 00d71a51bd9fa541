// gtx_devmem.hpp -- device memory of the library through a process-wide cache.
//
// The reference genotypes a chromosome region by region (50 kb each, src/main.cpp:684): a context -- graph tables, index,
// scratch, the HBM-table workspaces -- lives for milliseconds, and hipMalloc / hipFree (each hipFree also waits for the
// device) of its ~50 allocations were most of what creating one cost.  Freed blocks are kept per device in size classes
// and handed out again; nothing is returned to the driver before the cache holds more than its limit (GTX_DEVICE_CACHE_MB,
// default 32768) or gtx_device_cache_release() is called.  A block goes back to the cache only when no kernel can still use
// it: callers free after the work on it is known to be done (context destruction and the index build synchronise first).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace gtx
{
hipError_t dev_malloc(void ** p, size_t bytes); // on the current device
hipError_t dev_free(void * p);                  // back to the cache (NULL is fine)
// Zeroes device memory and returns when it IS zero.  (hipMemset on device memory returns before the fill has run, and the fill
// runs on the null stream: a kernel on a non-blocking stream -- every stream PyTorch makes -- is not ordered behind it, so a
// counter block "zeroed" by a plain hipMemset can be cleared in the middle of the first call that counts in it.)
// The stream of the context that is being made on this thread (gtx_ctx_create: uploads, the index build's kernels, the zeroing of
// what it allocates), nullptr outside of one: the work of making a context is ordered on a stream of its own, so that host
// threads that make contexts side by side (gtx_regions_run's builders) neither queue behind each other on the null stream nor
// wait for the regions that are running on other streams.
extern thread_local hipStream_t tls_build_stream;
inline hipError_t dev_zero(void * p, size_t bytes)
{
  hipError_t const e = hipMemsetAsync(p, 0, bytes, tls_build_stream);
  return e != hipSuccess ? e : hipStreamSynchronize(tls_build_stream);
}
// the same without the wait: for memory whose first user is a later launch on tls_build_stream
inline hipError_t dev_zero_async(void * p, size_t bytes) { return hipMemsetAsync(p, 0, bytes, tls_build_stream); }
// A non-blocking stream of the current device from a process-wide pool becomes this thread's tls_build_stream for the scope's
// life (GTX_BUILD_STREAM=0: the null stream, as before round 5 -- A/B).  The scope's end waits for the stream.
struct BuildStreamScope
{
  hipStream_t stream = nullptr, before = nullptr;
  int device = -1;
  BuildStreamScope();
  ~BuildStreamScope();
  BuildStreamScope(BuildStreamScope const &) = delete;
  BuildStreamScope & operator=(BuildStreamScope const &) = delete;
};
void dev_cache_release();                       // hipFree everything the cache holds
// A 64-byte slot of pinned host memory (a word the device writes and a later call reads, per context): cut from pages that are
// pinned once and kept for the life of the process -- hipHostMalloc / hipHostFree per context were most of what making and
// destroying a small region's context cost its host thread, and the free waits for the device.  nullptr when nothing can be pinned.
void * pinned_slot_get();
void pinned_slot_put(void * p);
} // namespace gtx
