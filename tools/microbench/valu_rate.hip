// VALU issue-rate microbenchmark: cycles per wave64 instruction per SIMD, by opcode, at W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int OP>
__global__ __launch_bounds__(64) void k(float* out, int iters, float seed, unsigned useed) {
  float a = seed + threadIdx.x, b = seed * 2.f, c = seed * 3.f, d = seed * 5.f;
  unsigned u = useed + threadIdx.x, v = useed * 3u, w = useed * 7u, x = useed * 11u;
  for (int i = 0; i < iters; i++) {
    if (OP == 0) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    if (OP == 1) { REP64(asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if (OP == 2) { REP64(asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a) : "v"(u));) }
    if (OP == 3) { REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b));) }
    if (OP == 4) { REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(u) : "v"(v));) }
    if (OP == 5) { REP64(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if (OP == 6) { REP64(asm volatile("v_cmp_le_f32 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc");) }
    if (OP == 7) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double*)&a) : "v"(*(double*)&b), "v"(*(double*)&c));) }
    if (OP == 8) { REP8(REP8(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(a), "+v"(d) : "v"(b), "v"(c));)) }  // two independent chains
    if (OP == 9) { REP64(asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    if (OP == 10) { REP64(asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u) : "v"(v));) }
    if (OP == 12) { REP64(asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a) : "v"(b), "v"(c));) }
    if (OP == 13) { REP64(asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[4:5]" : "+v"(a) : "v"(b) : "s4", "s5");) }
    if (OP == 14) { REP64(asm volatile("v_cmp_le_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b), "v"(c) : "vcc");) }
    if (OP == 15) { REP64(asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u) : "v"(v), "v"(w));) }
    if (OP == 16) { REP64(asm volatile("v_min_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if (OP == 17) { REP64(asm volatile("v_cmp_le_f32 s[6:7], %1, %2\n v_cndmask_b32_e64 %0, %0, %1, s[6:7]" : "+v"(a) : "v"(b), "v"(c) : "s6", "s7");) }
    if (OP == 18) { REP64(asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    if (OP == 19) { REP64(asm volatile("v_and_b32 %0, %0, %1" : "+v"(u) : "v"(v));) }
    if (OP == 20) { REP64(asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(b));) }
    if (OP == 21) { REP64(asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if (OP == 22) { REP64(asm volatile("v_cvt_f32_ubyte0 %0, %1 \n v_fma_f32 %0, %0, %2, %3" : "=&v"(a) : "v"(u), "v"(b), "v"(c));) }
    if (OP == 11) { REP64(asm volatile("v_rcp_f32 %0, %0" : "+v"(a));) }
  }
  if (a == 12345.f || u == 77u || d == 3.f) out[0] = a + u + w + x + d;
}

template <int OP>
void run(float* out, const char* name, int perInstr = 1) {
  for (int w : {1, 4, 8}) {
    const int iters = 2000, grid = 256 * 4 * w;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(64), 0, 0, out, 100, 1.0f, 3u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(64), 0, 0, out, iters, 1.0f, 3u);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double instrPerSimd = (double)w * iters * 64 * perInstr;
    printf("%-28s waves/SIMD %d  %.3f ms  %.2f ns per wave-instr per SIMD (%.2f clk at 2.4 GHz)\n", name, w, ms, ms * 1e6 / instrPerSimd, ms * 1e6 / instrPerSimd * 2.4);
  }
}

int main() {
  float* out; CK(hipMalloc(&out, 64));
  run<0>(out, "v_fma_f32 (dependent)");
  run<8>(out, "v_fma_f32 (2 chains)", 2);
  run<1>(out, "v_max_f32");
  run<9>(out, "v_max3_f32");
  run<5>(out, "v_mul_f32");
  run<2>(out, "v_cvt_f32_ubyte1");
  run<3>(out, "v_cndmask_b32");
  run<6>(out, "v_cmp_le_f32");
  run<4>(out, "v_add_u32");
  run<10>(out, "v_lshl_add_u32");
  run<7>(out, "v_pk_fma_f32");
  run<11>(out, "v_rcp_f32");
  run<12>(out, "v_cndmask e32 vcc, independent");
  run<13>(out, "v_cndmask e64 sgpr mask");
  run<14>(out, "v_cmp vcc + v_cndmask vcc", 2);
  run<17>(out, "v_cmp sgpr + v_cndmask e64", 2);
  run<15>(out, "v_bfi_b32");
  run<16>(out, "v_min_f32");
  run<18>(out, "v_med3_f32");
  run<19>(out, "v_and_b32");
  run<20>(out, "v_mov_b32");
  run<21>(out, "v_sub_f32");
  run<22>(out, "cvt_ubyte + fma", 2);
  return 0;
}
