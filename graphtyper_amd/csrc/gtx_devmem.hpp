// gtx_devmem.hpp -- device memory of the library through a process-wide cache.
//
// The reference genotypes a chromosome region by region (50 kb each, src/main.cpp:684): a context -- graph tables, index,
// scratch, the HBM-table workspaces -- lives for milliseconds, and hipMalloc / hipFree (each hipFree also waits for the
// device) of its ~50 allocations were most of what creating one cost.  Freed blocks are kept per device in size classes
// and handed out again; nothing is returned to the driver before the cache holds more than its limit (GTX_DEVICE_CACHE_MB,
// default 8192) or gtx_device_cache_release() is called.  A block goes back to the cache only when no kernel can still use
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
inline hipError_t dev_zero(void * p, size_t bytes)
{
  hipError_t const e = hipMemsetAsync(p, 0, bytes, nullptr);
  return e != hipSuccess ? e : hipStreamSynchronize(nullptr);
}
void dev_cache_release();                       // hipFree everything the cache holds
} // namespace gtx
