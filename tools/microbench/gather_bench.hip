// Microbenchmark: dependent random gathers of 64-B records (4 x dwordx4 per lane) from an L2-resident table,
// one wave per workgroup, W waves per CU — the memory-side pattern of a BVH4 node visit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int LOADS, int VALU, int ACTIVE, int STRIDE16>
__global__ __launch_bounds__(64) void gather(const uint4* __restrict__ tab, uint32_t n, int iters, uint32_t* out) {
  const int lane = threadIdx.x;
  uint32_t idx = (blockIdx.x * 64u + lane) * 2654435761u % n;
  float acc = 0.f;
  uint32_t h = idx;
  if (lane < ACTIVE) {
    for (int it = 0; it < iters; it++) {
      const uint4* p = tab + (size_t)idx * STRIDE16;
      uint4 q[4];
#pragma unroll
      for (int l = 0; l < LOADS; l++) q[l] = p[l];
      uint32_t x = 0;
#pragma unroll
      for (int l = 0; l < LOADS; l++) x ^= q[l].x + q[l].y * 3u + q[l].z * 5u + q[l].w * 7u;
      float f = __uint_as_float((x & 0x007fffffu) | 0x3f800000u);
#pragma unroll
      for (int v = 0; v < VALU; v++) f = fmaf(f, 1.0000001f, 0.25f) ;
      acc += f;
      h = h * 1664525u + 1013904223u + x;
      idx = (h >> 4) % n;
    }
  }
  if (acc == 123.f) out[0] = h;
}

template <int LOADS, int VALU, int ACTIVE, int STRIDE16>
void run(const uint4* tab, uint32_t n, int wavesPerCU, uint32_t* out, const char* name) {
  const int iters = 2000;
  const int grid = 256 * wavesPerCU;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((gather<LOADS, VALU, ACTIVE, STRIDE16>), dim3(grid), dim3(64), 0, 0, tab, n, 200, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((gather<LOADS, VALU, ACTIVE, STRIDE16>), dim3(grid), dim3(64), 0, 0, tab, n, iters, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  double visits = (double)grid * ACTIVE * iters;
  printf("%-46s waves/CU %2d  %.3f ms  %.1f G visits/s  (%.2f lane-loads/clk/CU at 2.4 GHz)\n", name, wavesPerCU, ms, visits / ms * 1e-6,
         visits * LOADS / (ms * 1e-3) / 256 / 2.4e9);
}

int main(int argc, char** argv) {
  uint32_t n = argc > 1 ? atoi(argv[1]) : 68135;
  std::vector<uint32_t> h((size_t)n * 20);
  for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)rand() * 2654435761u + (uint32_t)i;
  uint4* tab; uint32_t* out;
  CK(hipMalloc(&tab, h.size() * 4)); CK(hipMalloc(&out, 64));
  CK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  printf("table: %u records\n", n);
  for (int w : {8, 16, 20, 32}) {
    run<4, 0, 64, 4>(tab, n, w, out, "4 loads, no valu, 64 lanes");
    run<4, 0, 32, 4>(tab, n, w, out, "4 loads, no valu, 32 lanes");
    run<3, 0, 64, 3>(tab, n, w, out, "3 loads (48-B records), no valu, 64 lanes");
    run<3, 0, 32, 3>(tab, n, w, out, "3 loads (48-B records), no valu, 32 lanes");
    run<2, 0, 64, 4>(tab, n, w, out, "2 loads, no valu, 64 lanes");
    run<1, 0, 64, 4>(tab, n, w, out, "1 load, no valu, 64 lanes");
    run<4, 100, 64, 4>(tab, n, w, out, "4 loads, 100 fma, 64 lanes");
    run<4, 100, 32, 4>(tab, n, w, out, "4 loads, 100 fma, 32 lanes");
    run<3, 100, 32, 3>(tab, n, w, out, "3 loads, 100 fma, 32 lanes");
    run<0, 100, 64, 4>(tab, n, w, out, "0 loads, 100 fma, 64 lanes");
  }
  return 0;
}
