// wave_hip.hpp -- the hardware side of the wave policy the kernel sources are written against (graph_dev.hpp: "wave-uniform +
// lane lambdas"): WaveHip for workspaces in LDS, WaveHipMem for workspaces in HBM, and the wave-level claims on device counters.
// Device code only; included by the translation units that hold kernels (gtx_api.hip, gtx_hbm_passes.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace gtx
{
struct WaveHip
{
  template <class T>
  struct PerLane
  {
    T v;
    __device__ inline T & operator[](uint32_t) { return v; }
    __device__ inline T const & operator[](uint32_t) const { return v; }
  };
  template <class F>
  static __device__ inline void lanes(F && f)
  {
    f(threadIdx.x & 63u);
  }
  static __device__ inline bool leader() { return (threadIdx.x & 63u) == 0; }
  // the value lane `lane` (wave-uniform) holds
  static __device__ inline uint32_t from_lane(PerLane<uint32_t> const & p, uint32_t lane) { return __builtin_amdgcn_readlane(p.v, lane); }
  // value known to be equal on all lanes -> scalar register
  static __device__ inline uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
  static __device__ inline int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
  static __device__ inline bool uni(bool v) { return __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v)) != 0; }
  static __device__ inline uint64_t uni(uint64_t v)
  {
    uint32_t const lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
    uint32_t const hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
    return (static_cast<uint64_t>(hi) << 32) | lo;
  }
  // Orders the leader's LDS writes before the other lanes' reads.  All 64 lanes belong to one wavefront whose LDS
  // instructions are issued and serviced in program order, so no hardware wait is needed: the wavefront-scope fences
  // only stop the compiler from moving or merging LDS accesses across this point (GTX_HARD_SYNC=1 at build time
  // falls back to a real workgroup barrier for A/B checks).
  static __device__ inline void lds_sync()
  {
#ifdef GTX_HARD_SYNC
    __syncthreads();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
  }
  static __device__ inline uint64_t ballot(PerLane<bool> const & p) { return __ballot(p.v); }
  static __device__ inline uint32_t sum(PerLane<uint32_t> const & p)
  {
    uint32_t x = p.v;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
      x += __shfl_xor(x, d);
    return x;
  }
  static __device__ inline void excl_scan(PerLane<uint32_t> const & in, PerLane<uint32_t> & out, uint32_t & total)
  {
    uint32_t x = in.v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
      uint32_t const y = __shfl_up(x, d);
      if ((threadIdx.x & 63u) >= static_cast<uint32_t>(d))
        x += y;
    }
    total = __shfl(x, 63);
    out.v = x - in.v;
  }
  static __device__ inline uint32_t max(PerLane<uint32_t> const & p)
  {
    uint32_t x = p.v;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
    {
      uint32_t const y = __shfl_xor(x, d);
      x = y > x ? y : x;
    }
    return x;
  }
  static __device__ inline void atomic_or_u64(uint64_t * p, uint64_t v) { atomicOr(reinterpret_cast<unsigned long long *>(p), static_cast<unsigned long long>(v)); }
  static __device__ inline unsigned long long clock() { return clock64(); }
  static __device__ inline void atomic_add_u32(uint32_t * p, uint32_t v) { atomicAdd(p, v); }
  static __device__ inline void atomic_add_u64(unsigned long long * p, unsigned long long v) { atomicAdd(p, v); }
  // next free slot of a log every lane appends to (fetch-and-increment).  The lanes that are here together ask with one
  // atomic: a single device counter sustains ~90 M returning atomics a second, a dense graph wants more log entries.
  static __device__ inline uint32_t atomic_claim_u32(uint32_t * p)
  {
    unsigned long long const here = __ballot(1);
    uint32_t const lane = threadIdx.x & 63u, leader = static_cast<uint32_t>(__builtin_ctzll(here));
    uint32_t base = 0;
    if (lane == leader)
      base = atomicAdd(p, static_cast<uint32_t>(__builtin_popcountll(here)));
    base = __shfl(base, static_cast<int>(leader));
    return base + static_cast<uint32_t>(__builtin_popcountll(here & ((1ull << lane) - 1ull)));
  }
};

// Second pass: the workspace is in global memory, so the leader's stores must have completed (vmcnt) before the other
// lanes load them: workgroup-scope release/acquire fences are exactly that wait (one CU, one L1: no cache maintenance).
// The wave barrier between them is the convergence point that keeps the compiler from letting lanes run ahead of the
// leader (a workgroup barrier would be dropped for a one-wave workgroup and leave nothing to anchor on).
struct WaveHipMem : WaveHip
{
  static __device__ inline void mem_sync()
  {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  static __device__ inline void lds_sync() { mem_sync(); }
};

// The leader takes `n` units from a device counter; every lane gets the old value (readfirstlane is the convergence
// point: no lane continues before the leader's atomic has returned).
static __device__ inline uint32_t wave_claim(uint32_t * counter, uint32_t n)
{
  uint32_t v = 0;
  if ((threadIdx.x & 63u) == 0)
    v = atomicAdd(counter, n);
  return __builtin_amdgcn_readfirstlane(v);
}

static __device__ inline unsigned long long wave_claim64(unsigned long long * counter, unsigned long long n)
{
  unsigned long long v = 0;
  if ((threadIdx.x & 63u) == 0)
    v = atomicAdd(counter, n);
  return WaveHip::uni(static_cast<uint64_t>(v));
}

} // namespace gtx
