// Probe: what an LDS operand read costs beside MFMAs. 4 waves per SIMD (1024-thread workgroup per CU), each wave
// alternating a phase of 27 i8 32x32x32 MFMAs and a phase of 174 integer VALU ops; NR ds_read_b128 (or b64) per
// iteration feed the MFMAs, issued PF groups ahead of their use.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NT, int NR, bool B64, bool VALU>
__global__ __launch_bounds__(NT) void k(int iters, int *out) {
  __shared__ v4i buf[2048];
  for (int i = threadIdx.x; i < 2048; i += NT) buf[i] = v4i{i, 2, 3, 4};
  __syncthreads();
  v16i c0 = {0}, c1 = {0}, c2 = {0};
  int x0 = threadIdx.x, x1 = 3, x2 = 5, x3 = 7, x4 = 11, x5 = 13;
  const int l = threadIdx.x & 63;
  v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, (int)blockIdx.x};
  for (int i = 0; i < iters; i++) {
    v4i an = a, bn = b;
#pragma unroll
    for (int j = 0; j < 9; j++) {
      // operands of the NEXT group (one group ahead), NR reads spread over the 9 groups
      constexpr int per = NR / 9;   // 0, 1, 2 or 4 reads per group
      if (per >= 1) { if (B64) { const v2i t = reinterpret_cast<const v2i *>(buf)[(2 * l + 128 * j + i) & 4095]; an.x = t.x; an.y = t.y; } else an = buf[(l + 64 * j + i) & 2047]; }
      if (per >= 2) { if (B64) { const v2i t = reinterpret_cast<const v2i *>(buf)[(2 * l + 128 * j + 2048 + i) & 4095]; bn.x = t.x; bn.y = t.y; } else bn = buf[(l + 64 * j + 1024 + i) & 2047]; }
      if (per >= 4) { const v4i e = buf[(l + 64 * j + 512 + i) & 2047], f = buf[(l + 64 * j + 1536 + i) & 2047]; an.z ^= e.z; bn.w ^= f.w; an.w ^= e.x; bn.z ^= f.y; }
      c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
      a = an; b = bn;
    }
    if (VALU) {
#pragma unroll
      for (int j = 0; j < 29; j++) { x0 = x0 * 3 + x1; x1 = (x1 << 1) ^ x2; x2 = x2 + x3; x3 = x3 ^ (x4 >> 1); x4 = x4 + x5; x5 = x5 ^ x0; }
      x0 += c0[i & 15];
    }
  }
  out[blockIdx.x * NT + threadIdx.x] = c0[0] + c1[1] + c2[2] + x0 + x1 + x2 + x3 + x4 + x5;
}
template <int NT, int NR, bool B64, bool VALU> float run(int iters, int *d) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NT, NR, B64, VALU>), dim3(256), dim3(NT), 0, 0, iters, d);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NT, NR, B64, VALU>), dim3(256), dim3(NT), 0, 0, iters, d);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms * 1e6f / iters / (NT / 256);
}
int main() {
  int *d; (void)hipMalloc(&d, 256 * 1024 * 4);
  const int it = 4000;
  printf("ns per wave-iteration per SIMD (27 MFMA + 174 VALU), 4 waves/SIMD, reads one group ahead:\n");
  printf("  ds_read_b128 per iteration:  0: %.0f   9: %.0f   18: %.0f   36: %.0f\n", run<1024, 0, false, true>(it, d), run<1024, 9, false, true>(it, d), run<1024, 18, false, true>(it, d), run<1024, 36, false, true>(it, d));
  printf("  ds_read_b64  per iteration:  9: %.0f   18: %.0f\n", run<1024, 9, true, true>(it, d), run<1024, 18, true, true>(it, d));
  printf("  no VALU phase, b128 reads:   0: %.0f   18: %.0f   36: %.0f\n", run<1024, 0, false, false>(it, d), run<1024, 18, false, false>(it, d), run<1024, 36, false, false>(it, d));
  printf("  2 waves/SIMD, b128 reads:    0: %.0f   18: %.0f   36: %.0f\n", run<512, 0, false, true>(it, d), run<512, 18, false, true>(it, d), run<512, 36, false, true>(it, d));
  return 0;
}
