// gtx_devmem.cpp -- see gtx_devmem.hpp
#include "gtx_devmem.hpp"

#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace gtx
{
namespace
{
struct DevCache
{
  std::mutex m;
  struct Block
  {
    int device;
    size_t bytes;
  };
  std::unordered_map<void *, Block> live;                         // every block handed out or cached
  std::map<std::pair<int, size_t>, std::vector<void *>> free_by; // (device, class size) -> cached blocks
  size_t cached_bytes = 0;
  size_t limit = 0;
};

DevCache & cache()
{
  static DevCache * c = new DevCache; // (never destroyed: the runtime may be gone before static destructors run)
  return *c;
}

// size classes: powers of two up to 1 MiB, then multiples of 1 MiB up to 64 MiB, then multiples of 16 MiB
size_t class_of(size_t bytes)
{
  if (bytes <= 256)
    return 256;
  if (bytes <= (1u << 20))
  {
    size_t c = 256;
    while (c < bytes)
      c <<= 1;
    return c;
  }
  size_t const step = bytes <= (64ull << 20) ? (1ull << 20) : (16ull << 20);
  return (bytes + step - 1) / step * step;
}
} // namespace

hipError_t dev_malloc(void ** p, size_t bytes)
{
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess)
    return e;
  size_t const cls = class_of(bytes);
  DevCache & c = cache();
  {
    std::lock_guard<std::mutex> lock(c.m);
    if (c.limit == 0)
    {
      char const * env = std::getenv("GTX_DEVICE_CACHE_MB");
      long long const mb = env ? std::atoll(env) : 32768; // (32 GB of 288: a context in flight holds 1.4 GB of workspaces per call scratch and 2 GB slabs) // (a negative value keeps nothing: it must not become a huge unsigned limit)
      c.limit = static_cast<size_t>(mb < 0 ? 0 : mb) << 20;
      if (c.limit == 0)
        c.limit = 1; // (0 MB: nothing is kept)
    }
    auto it = c.free_by.find({dev, cls});
    if (it != c.free_by.end() && !it->second.empty())
    {
      *p = it->second.back();
      it->second.pop_back();
      c.cached_bytes -= cls;
      return hipSuccess;
    }
  }
  e = hipMalloc(p, cls);
  if (e != hipSuccess)
  {
    // the cache may hold what the driver lacks: give it back and try once more
    dev_cache_release();
    e = hipMalloc(p, cls);
    if (e != hipSuccess)
      return e;
  }
  std::lock_guard<std::mutex> lock(c.m);
  c.live[*p] = {dev, cls};
  return hipSuccess;
}

hipError_t dev_free(void * p)
{
  if (!p)
    return hipSuccess;
  DevCache & c = cache();
  {
    std::lock_guard<std::mutex> lock(c.m);
    auto it = c.live.find(p);
    if (it == c.live.end())
      return hipFree(p); // not ours
    if (c.cached_bytes + it->second.bytes <= c.limit)
    {
      c.free_by[{it->second.device, it->second.bytes}].push_back(p);
      c.cached_bytes += it->second.bytes;
      return hipSuccess;
    }
    c.live.erase(it);
  }
  return hipFree(p);
}

thread_local hipStream_t tls_build_stream = nullptr;

namespace
{
struct StreamPool
{
  std::mutex m;
  std::map<int, std::vector<hipStream_t>> idle; // per device; the streams live as long as the process
};
StreamPool & stream_pool()
{
  static StreamPool * p = new StreamPool;
  return *p;
}
} // namespace

BuildStreamScope::BuildStreamScope()
{
  before = tls_build_stream;
  char const * off = std::getenv("GTX_BUILD_STREAM");
  if ((off && off[0] == '0') || hipGetDevice(&device) != hipSuccess)
  {
    device = -1;
    return;
  }
  {
    StreamPool & sp = stream_pool();
    std::lock_guard<std::mutex> lock(sp.m);
    auto & v = sp.idle[device];
    if (!v.empty())
    {
      stream = v.back();
      v.pop_back();
    }
  }
  if (!stream && hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess)
  {
    stream = nullptr;
    device = -1;
    return;
  }
  tls_build_stream = stream;
}

BuildStreamScope::~BuildStreamScope()
{
  tls_build_stream = before;
  if (!stream)
    return;
  (void)hipStreamSynchronize(stream);
  StreamPool & sp = stream_pool();
  std::lock_guard<std::mutex> lock(sp.m);
  sp.idle[device].push_back(stream);
}

void dev_cache_release()
{
  DevCache & c = cache();
  std::vector<void *> all;
  {
    std::lock_guard<std::mutex> lock(c.m);
    for (auto & kv : c.free_by)
    {
      for (void * p : kv.second)
      {
        all.push_back(p);
        c.live.erase(p);
      }
      kv.second.clear();
    }
    c.cached_bytes = 0;
  }
  for (void * p : all)
    (void)hipFree(p);
}
namespace
{
std::mutex g_pinned_m;
std::vector<void *> g_pinned_free;
} // namespace

void * pinned_slot_get()
{
  std::lock_guard<std::mutex> lock(g_pinned_m);
  if (g_pinned_free.empty())
  {
    size_t constexpr PAGE = 4096, SLOT = 64;
    void * page = nullptr;
    if (hipHostMalloc(&page, PAGE) != hipSuccess || !page)
      return nullptr;
    for (size_t k = 0; k < PAGE / SLOT; ++k)
      g_pinned_free.push_back(static_cast<char *>(page) + k * SLOT);
  }
  void * p = g_pinned_free.back();
  g_pinned_free.pop_back();
  return p;
}

void pinned_slot_put(void * p)
{
  if (!p)
    return;
  std::lock_guard<std::mutex> lock(g_pinned_m);
  g_pinned_free.push_back(p);
}
} // namespace gtx

extern "C" void gtx_device_cache_release(void) { gtx::dev_cache_release(); }
