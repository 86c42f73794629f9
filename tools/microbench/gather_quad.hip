// Quad-cooperative 64-B record gather: the 4 lanes of a quad fetch the 4 x 16-B pieces of ONE record per instruction
// (contiguous 64 B per quad), 4 instructions serve the quad's 4 rays; transposition through LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int S> __device__ __forceinline__ uint32_t quadBcast(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, S | (S << 2) | (S << 4) | (S << 6), 0xf, 0xf, false);
}

// MODE 0: baseline per-lane 4 loads.  1: quad loads to registers, no transposition (TA rate only; wrong data use).
// 2: quad loads by LDS-DMA + ds_read_b128 x4.  3: quad loads to registers + ds_write_b128 + ds_read_b128.
template <int MODE, int VALU, int HALF>
__global__ __launch_bounds__(64) void gather(const uint4* __restrict__ tab, uint32_t n, int iters, uint32_t* out) {
  __shared__ uint4 s_x[4 * 65];
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)tab, 0, (int)(n * 64u), 0x00020000);
  const bool active = HALF ? ((((threadIdx.x * 2654435761u) >> 7) & 1u) != 0u) : true;
  const int lane = threadIdx.x;
  uint32_t idx = (blockIdx.x * 64u + lane) * 2654435761u % n;
  float acc = 0.f;
  uint32_t h = idx;
  for (int it = 0; it < iters; it++) {
    uint4 q[4];
    if (MODE == 4) {
      // buffer_load ... lds with a 32-bit per-lane offset: own record's byte offset broadcast inside the quad by DPP
      const uint32_t own = active ? idx * 64u : 0xffffff00u;  // inactive ray: out of range, dropped by the range check
      const uint32_t pieceOff = (uint32_t)(lane & 3) * 16u;
#define QB(S) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)own, S | (S << 2) | (S << 4) | (S << 6), 0xf, 0xf, false) + pieceOff)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(s_x + 0), 16, QB(0), 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(s_x + 65), 16, QB(1), 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(s_x + 130), 16, QB(2), 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(s_x + 195), 16, QB(3), 0, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const uint4* r = s_x + (lane & 3) * 65 + (lane >> 2) * 4;
      q[0] = q[1] = q[2] = q[3] = make_uint4(0, 0, 0, 0);
      if (active) {
#pragma unroll
        for (int l = 0; l < 4; l++) q[l] = r[l];
      }
    } else if (MODE == 0) {
      const uint4* p = tab + (size_t)idx * 4;
#pragma unroll
      for (int l = 0; l < 4; l++) q[l] = p[l];
    } else {
      const uint32_t i0 = quadBcast<0>(idx), i1 = quadBcast<1>(idx), i2 = quadBcast<2>(idx), i3 = quadBcast<3>(idx);
      const int piece = lane & 3;
      const uint4* p0 = tab + (size_t)i0 * 4 + piece;
      const uint4* p1 = tab + (size_t)i1 * 4 + piece;
      const uint4* p2 = tab + (size_t)i2 * 4 + piece;
      const uint4* p3 = tab + (size_t)i3 * 4 + piece;
      if (MODE == 1) {
        q[0] = *p0; q[1] = *p1; q[2] = *p2; q[3] = *p3;
      } else if (MODE == 2) {
        __builtin_amdgcn_global_load_lds(p0, s_x + 0, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(p1, s_x + 64, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(p2, s_x + 128, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(p3, s_x + 192, 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // record of ray `lane` = region (lane & 3), quad (lane >> 2): 4 consecutive uint4
        const uint4* r = s_x + (lane & 3) * 64 + (lane >> 2) * 4;
#pragma unroll
        for (int l = 0; l < 4; l++) q[l] = r[l];
        __builtin_amdgcn_s_barrier();
      } else {
        const uint4 a = *p0, b = *p1, c = *p2, d = *p3;
        s_x[lane] = a; s_x[64 + lane] = b; s_x[128 + lane] = c; s_x[192 + lane] = d;
        __builtin_amdgcn_s_barrier();
        const uint4* r = s_x + (lane & 3) * 64 + (lane >> 2) * 4;
#pragma unroll
        for (int l = 0; l < 4; l++) q[l] = r[l];
        __builtin_amdgcn_s_barrier();
      }
    }
    uint32_t x = 0;
#pragma unroll
    for (int l = 0; l < 4; l++) x ^= q[l].x + q[l].y * 3u + q[l].z * 5u + q[l].w * 7u;
    float f = __uint_as_float((x & 0x007fffffu) | 0x3f800000u);
#pragma unroll
    for (int v = 0; v < VALU; v++) f = fmaf(f, 1.0000001f, 0.25f);
    acc += f;
    h = h * 1664525u + 1013904223u + x;
    idx = (h >> 4) % n;
  }
  if (acc == 123.f) out[0] = h;
  if (MODE >= 2 && iters < 0) out[1] = s_x[lane].x;
}

template <int MODE, int VALU, int HALF = 0>
void run(const uint4* tab, uint32_t n, int wavesPerCU, uint32_t* out, const char* name) {
  const int iters = 2000;
  const int grid = 256 * wavesPerCU;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((gather<MODE, VALU, HALF>), dim3(grid), dim3(64), 0, 0, tab, n, 200, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((gather<MODE, VALU, HALF>), dim3(grid), dim3(64), 0, 0, tab, n, iters, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  double visits = (double)grid * 64 * iters;
  printf("%-52s waves/CU %2d  %.3f ms  %.1f G visits/s\n", name, wavesPerCU, ms, visits / ms * 1e-6);
}

int main(int argc, char** argv) {
  uint32_t n = argc > 1 ? atoi(argv[1]) : 68135;
  std::vector<uint32_t> h((size_t)n * 16);
  for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)rand() * 2654435761u + (uint32_t)i;
  uint4* tab; uint32_t* out;
  CK(hipMalloc(&tab, h.size() * 4)); CK(hipMalloc(&out, 64));
  CK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  printf("table: %u records\n", n);
  for (int w : {8, 12, 16, 20}) {
    run<0, 0>(tab, n, w, out, "per-lane 4 loads");
    run<1, 0>(tab, n, w, out, "quad loads, registers, no transposition");
    run<2, 0>(tab, n, w, out, "quad loads by LDS-DMA + 4 ds_read_b128");
    run<3, 0>(tab, n, w, out, "quad loads, registers, ds_write + ds_read");
    run<4, 0>(tab, n, w, out, "quad buffer LDS-DMA (dpp offsets, padded regions)");
    run<4, 100>(tab, n, w, out, "quad buffer LDS-DMA, 100 fma");
    run<4, 100, 1>(tab, n, w, out, "quad buffer LDS-DMA, 100 fma, half of the rays active (x2 for per-ray rate)");
    run<0, 100>(tab, n, w, out, "per-lane 4 loads, 100 fma");
    run<2, 100>(tab, n, w, out, "quad LDS-DMA, 100 fma");
    run<3, 100>(tab, n, w, out, "quad reg + LDS transposition, 100 fma");
  }
  return 0;
}
