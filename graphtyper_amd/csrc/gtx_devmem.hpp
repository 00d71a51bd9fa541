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
void dev_cache_release();                       // hipFree everything the cache holds
} // namespace gtx
