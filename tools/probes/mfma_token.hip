// Probe: two waves per SIMD, each alternating an MFMA phase (12 i8 MFMAs) and a VALU phase (96 ops) — free-running
// against an "MFMA token" per SIMD (LDS flag: only one wave of the pair is in its MFMA phase at a time).
// 512-thread workgroups, one per CU: waves w and w+4 share SIMD w.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE, int NM, int NV>   // MODE 0: free-running, 1: token, 2: token + the holder runs at priority 3
__global__ __launch_bounds__(512) void k(int iters, int *out) {
  __shared__ volatile int tok[4];
  const int w = threadIdx.x >> 6, simd = w & 3, me = w >> 2;
  if (threadIdx.x < 4) tok[threadIdx.x] = 0;
  __syncthreads();
  v16i c0 = {0}, c1 = {0}, c2 = {0};
  v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, (int)blockIdx.x};
  int x0 = threadIdx.x, x1 = 3, x2 = 5, x3 = 7, x4 = 11, x5 = 13;
  for (int i = 0; i < iters; i++) {
    if (MODE >= 1) { while (tok[simd] != me) __builtin_amdgcn_s_sleep(1); }
    if (MODE == 2) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int j = 0; j < NM / 3; j++) {
      c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
    }
    if (MODE == 2) __builtin_amdgcn_s_setprio(0);
    if (MODE >= 1) { asm volatile("" ::: "memory"); if ((threadIdx.x & 63) == 0) tok[simd] = me ^ 1; }
#pragma unroll
    for (int j = 0; j < NV / 6; j++) {
      x0 = x0 * 3 + x1; x1 = (x1 << 1) ^ x2; x2 = x2 + x3; x3 = x3 ^ (x4 >> 1); x4 = x4 + x5; x5 = x5 ^ x0;
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + x0 + x1 + x2 + x3 + x4 + x5;
}

template <int MODE, int NM, int NV> float run(int iters, int *d) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NM, NV>), dim3(256), dim3(512), 0, 0, iters, d);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, NM, NV>), dim3(256), dim3(512), 0, 0, iters, d);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  int *d; (void)hipMalloc(&d, 256 * 512 * 4);
  const int it = 4000;
  printf("phases of 12 MFMA / 96 VALU  : free %.3f ms, token %.3f ms, token+prio %.3f ms\n", run<0, 12, 96>(it * 2, d), run<1, 12, 96>(it * 2, d), run<2, 12, 96>(it * 2, d));
  printf("phases of 27 MFMA / 174 VALU : free %.3f ms, token %.3f ms, token+prio %.3f ms\n", run<0, 27, 174>(it, d), run<1, 27, 174>(it, d), run<2, 27, 174>(it, d));
  return 0;
}
