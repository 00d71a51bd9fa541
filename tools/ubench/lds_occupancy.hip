// How many 256-thread workgroups with 20 480 bytes of LDS each does a CU of gfx950 hold at once (160 KB per CU: eight, if nothing is
// set aside)?  Every workgroup spins for a fixed time; a launch of 256 x k workgroups takes one spin when k fit a CU, two when they do not.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BYTES>
__global__ __launch_bounds__(256) void spin(unsigned long long ticks, unsigned * out)
{
  __shared__ unsigned s[BYTES / 4];
  s[threadIdx.x] = threadIdx.x;
  __syncthreads();
  unsigned long long const t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks)
    ;
  if (threadIdx.x == 0)
    out[blockIdx.x] = s[(blockIdx.x * 7) % (BYTES / 4)];
}
template <int BYTES>
void run(int per_cu)
{
  unsigned * out;
  hipMalloc(&out, 256 * 16 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(spin<BYTES>, dim3(256), dim3(256), 0, 0, 1000ull, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(spin<BYTES>, dim3(256 * per_cu), dim3(256), 0, 0, 10000ull /* 100 us at 100 MHz */, out);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin<BYTES>, 256, 0);
  printf("LDS %6d B per workgroup, %d workgroups per CU launched: %.3f ms (occupancy API: %d per CU)\n", BYTES, per_cu, ms, occ);
  hipFree(out);
}
int main()
{
  for (int k : {6, 7, 8, 9})
    run<20480>(k);
  for (int k : {7, 8})
    run<20224>(k);
  for (int k : {7, 8})
    run<16384>(k);
  return 0;
}
