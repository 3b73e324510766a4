// Probe: do i8 MFMAs of one wave and VALU work of ANOTHER wave on the same SIMD overlap?
// 512-thread workgroups, one per CU: waves 0-3 and 4-7 land pairwise on SIMDs 0-3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE>   // bit0: waves 0-3 run MFMAs, bit1: waves 4-7 run VALU, bit2: waves 4-7 run MFMA too, bit3: waves 0-3 also VALU
__global__ __launch_bounds__(512) void k(int iters, int *out) {
  const int w = threadIdx.x >> 6;
  v16i c0 = {0}, c1 = {0}, c2 = {0};
  v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, (int)blockIdx.x};
  int x0 = threadIdx.x, x1 = 3, x2 = 5, x3 = 7, x4 = 11, x5 = 13;
  const bool do_m = (w < 4) ? (MODE & 1) : (MODE & 4);
  const bool do_v = (w < 4) ? (MODE & 8) : (MODE & 2);
  for (int i = 0; i < iters; i++) {
    if (do_m) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
      }
    }
    if (do_v) {
#pragma unroll
      for (int j = 0; j < 16; j++) {   // 96 dependent-free-ish VALU ops
        x0 = x0 * 3 + x1; x1 = (x1 << 1) ^ x2; x2 = x2 + x3; x3 = x3 ^ (x4 >> 1); x4 = x4 + x5; x5 = x5 ^ x0;
      }
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + x0 + x1 + x2 + x3 + x4 + x5;
}

template <int MODE> float run(int iters, int *d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, iters, d);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, iters, d);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  int *d; hipMalloc(&d, 256 * 512 * 4);
  const int it = 20000;
  printf("MFMA only (waves 0-3)            : %.3f ms\n", run<1>(it, d));
  printf("VALU only (waves 4-7)            : %.3f ms\n", run<2>(it, d));
  printf("MFMA w0-3 + VALU w4-7 (same SIMD): %.3f ms\n", run<3>(it, d));
  printf("MFMA on both wave sets           : %.3f ms\n", run<5>(it, d));
  printf("VALU on both wave sets           : %.3f ms\n", run<10>(it, d));
  printf("each wave MFMA then VALU (serial): %.3f ms\n", run<15>(it, d));
  return 0;
}
