// Probe: W waves per SIMD, each alternating an MFMA phase (NM i8 32x32x32 MFMAs, operands optionally re-read from LDS
// by ds_read_b128 like the K1 K loop) and a VALU phase (NV integer ops). One workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NT, int NM, int NV, bool LDSOPS>
__global__ __launch_bounds__(NT) void k(int iters, int *out) {
  __shared__ v4i buf[2048];
  for (int i = threadIdx.x; i < 2048; i += NT) buf[i] = v4i{i, 2, 3, 4};
  __syncthreads();
  v16i c0 = {0}, c1 = {0}, c2 = {0};
  v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, (int)blockIdx.x};
  int x0 = threadIdx.x, x1 = 3, x2 = 5, x3 = 7, x4 = 11, x5 = 13;
  const int l = threadIdx.x & 63;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < NM / 3; j++) {
      if (LDSOPS) { a = buf[(l + 64 * j + i) & 2047]; b = buf[(l + 64 * j + 1024 + i) & 2047]; }
      c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NV / 6; j++) {
      x0 = x0 * 3 + x1; x1 = (x1 << 1) ^ x2; x2 = x2 + x3; x3 = x3 ^ (x4 >> 1); x4 = x4 + x5; x5 = x5 ^ x0;
    }
    x0 += c0[i & 15];   // the VALU phase depends on the MFMA results like an epilogue
  }
  out[blockIdx.x * NT + threadIdx.x] = c0[0] + c1[1] + c2[2] + x0 + x1 + x2 + x3 + x4 + x5;
}

template <int NT, int NM, int NV, bool LDSOPS> float run(int iters, int *d) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NT, NM, NV, LDSOPS>), dim3(256), dim3(NT), 0, 0, iters, d);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NT, NM, NV, LDSOPS>), dim3(256), dim3(NT), 0, 0, iters, d);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  int *d; (void)hipMalloc(&d, 256 * 1024 * 4);
  const int it = 4000;
  // ns per wave-iteration on a SIMD = ms * 1e6 / (it * waves per SIMD)
  printf("27 MFMA + 174 VALU per iteration, ns per wave-iteration per SIMD (register operands):\n");
  printf("  1 wave/SIMD %.0f   2 waves %.0f   4 waves %.0f\n", run<256, 27, 174, false>(it, d) * 1e6 / it, run<512, 27, 174, false>(it, d) * 1e6 / it / 2, run<1024, 27, 174, false>(it, d) * 1e6 / it / 4);
  printf("same with 2 ds_read_b128 per 3 MFMAs:\n");
  printf("  1 wave/SIMD %.0f   2 waves %.0f   4 waves %.0f\n", run<256, 27, 174, true>(it, d) * 1e6 / it, run<512, 27, 174, true>(it, d) * 1e6 / it / 2, run<1024, 27, 174, true>(it, d) * 1e6 / it / 4);
  printf("MFMA only (27, register operands): 1 wave %.0f  4 waves %.0f ;  VALU only (174): 1 wave %.0f  4 waves %.0f\n", run<256, 27, 0, false>(it, d) * 1e6 / it, run<1024, 27, 0, false>(it, d) * 1e6 / it / 4, run<256, 0, 174, false>(it, d) * 1e6 / it, run<1024, 0, 174, false>(it, d) * 1e6 / it / 4);
  return 0;
}
