// Issue rate of wave64 vector ALU instructions on gfx950, by kind: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
// Each thread runs ITER rounds of 16 independent chains of one instruction kind; with W wavefronts per SIMD resident the SIMD's issue
// rate is (instructions per wavefront x W) / cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define ITER 4096
template <int KIND>
__global__ __launch_bounds__(64) void k(uint32_t * out, uint32_t seed, unsigned long long * cyc)
{
  uint32_t a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
    a[i] = seed * (i + 1) + threadIdx.x;
  uint32_t const b = seed ^ 0x9E3779B9u, c = seed + 7;
  unsigned long long const t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < ITER; ++it)
  {
#pragma unroll
    for (int i = 0; i < 16; ++i)
    {
      if (KIND == 0) a[i] = a[i] ^ b;                                         // v_xor_b32
      if (KIND == 1) a[i] = a[i] + b;                                         // v_add_u32
      if (KIND == 2) a[i] = __builtin_popcount(a[i]) + b;                     // v_bcnt_u32_b32 (adds its second operand)
      if (KIND == 3) a[i] = __builtin_amdgcn_alignbit(a[i], b, c & 31);       // v_alignbit_b32
      if (KIND == 4) a[i] = (a[i] & b) | c;                                   // v_and_or_b32
      if (KIND == 5) a[i] = a[i] * b;                                         // v_mul_lo_u32
      if (KIND == 6) a[i] = (a[i] >> (c & 31)) & 0xFFFFu;                     // v_bfe_u32
      if (KIND == 7) a[i] = a[i] > b ? a[i] - c : a[i] + c;                   // compare + select
      if (KIND == 8) { float f = __uint_as_float(a[i]); f = f * 1.0001f + 0.5f; a[i] = __float_as_uint(f); } // v_fma_f32
    }
  }
  unsigned long long const t1 = __builtin_readcyclecounter();
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i)
    x ^= a[i];
  out[blockIdx.x * 64 + threadIdx.x] = x;
  if (threadIdx.x == 0)
    atomicMax(cyc, t1 - t0);
}
template <int KIND>
void run(char const * name, int waves_per_simd)
{
  int const blocks = 256 * 4 * waves_per_simd;
  uint32_t * out;
  unsigned long long * cyc;
  hipMalloc(&out, blocks * 64 * 4);
  hipMalloc(&cyc, 8);
  hipMemset(cyc, 0, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, 12345u, cyc);
  hipDeviceSynchronize();
  hipMemset(cyc, 0, 8);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, 12345u, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  double const insts = 16.0 * ITER;
  // s_memtime / readcyclecounter ticks at a constant 100 MHz on gfx9: use the event time and the nominal clock instead
  printf("%-18s W=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD  (= %.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ms,
         ms * 1e6 / (insts * waves_per_simd), ms * 1e6 / (insts * waves_per_simd) * 2.4);
  hipFree(out);
  hipFree(cyc);
}
int main()
{
  for (int w : {1, 2, 4, 8})
  {
    run<0>("v_xor_b32", w);
    run<1>("v_add_u32", w);
    run<2>("v_bcnt_u32_b32", w);
    run<3>("v_alignbit_b32", w);
    run<4>("v_and_or_b32", w);
    run<5>("v_mul_lo_u32", w);
    run<6>("shift+and (bfe)", w);
    run<7>("cmp+cndmask+add/sub", w);
    run<8>("v_fma_f32", w);
  }
  return 0;
}
