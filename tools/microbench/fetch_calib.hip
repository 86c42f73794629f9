// Calibration of rocprofv3's FETCH_SIZE / TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES for the access pattern of a BVH
// node visit: every lane gathers one random 48-byte record (three dwordx4 loads) from a table far larger than L2 +
// Infinity Cache, so every record is fetched from HBM.  MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide
// coalesced streams (where it reports half the bytes) and says other widths are uncalibrated: this is that check.
//   hipcc -O3 --offload-arch=gfx950 -o fetch_calib fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib            (and the TCP / TCC counters in a second run)
// Prints what one launch must have fetched if memory is read in 64-B or in 128-B pieces.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(64) void gather48(const uint4* __restrict__ tab, uint32_t n, int iters, uint32_t* out) {
  uint32_t h = (blockIdx.x * 64u + threadIdx.x) * 2654435761u + 12345u;
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
    h = h * 1664525u + 1013904223u;
    const uint32_t idx = (uint32_t)(((uint64_t)(h >> 1) * n) >> 31);  // independent of the data: the loads are not a dependent chain
    const uint4* p = tab + (size_t)idx * 3;
    const uint4 a = p[0], b = p[1], c = p[2];
    acc += a.x ^ b.y ^ c.z;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// the same lanes streaming the table in order, 16 B per lane: the pattern the guide's factor of two was measured on
__global__ __launch_bounds__(256) void stream16(const uint4* __restrict__ tab, size_t nvec, uint32_t* out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) acc += tab[i].x;
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atoll(argv[1]) : 40000000u;  // 48-B records: 1.92 GB
  const int iters = 64, grid = 256 * 32;
  uint4* tab;
  uint32_t* out;
  CK(hipMalloc(&tab, (size_t)n * 48));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(tab, 1, (size_t)n * 48));
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(gather48, dim3(grid), dim3(64), 0, 0, tab, n, iters, out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double visits = (double)grid * 64 * iters;
  // a 48-B record at a multiple of 48 B crosses a 64-B boundary in 2 of 4 positions and a 128-B boundary in 2 of 8
  printf("gather48: %.0f records of 48 B from a %.2f GB table in %.3f ms: %.1f GB at 64-B granularity, %.1f GB at 128-B, %.1f GB of records; "
         "%.0f lane-loads\n", visits, (double)n * 48 / 1e9, ms, visits * 1.5 * 64 / 1e9, visits * 1.25 * 128 / 1e9, visits * 48 / 1e9, visits * 3);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(stream16, dim3(256 * 8), dim3(256), 0, 0, tab, (size_t)n * 3, out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("stream16: %.2f GB read in order in %.3f ms (%.0f GB/s)\n", (double)n * 48 / 1e9, ms, (double)n * 48 / 1e6 / ms);
  return 0;
}
