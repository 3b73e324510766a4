// Probe: issue rate of the three fp64 instructions the exact int16 FIR is made of (v_mul_f64, v_add_f64, v_trunc_f64),
// alone and as the FIR's mul/add/trunc chain, 8 independent chains per lane, 4 and 8 waves per SIMD; and the clock held.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o _bin/dp_rates dp_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k(int iters, double *out, unsigned long long *clk) {
  double a[8], x = 1.0000001 + threadIdx.x * 1e-9, al = 0.999999;
  for (int i = 0; i < 8; i++) a[i] = 100.0 + i + threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (MODE == 0) a[i] = __dmul_rn(a[i], al);
        else if (MODE == 1) a[i] = __dadd_rn(a[i], x);
        else if (MODE == 2) { double t; asm volatile("v_trunc_f64 %0, %1" : "=v"(t) : "v"(a[i])); a[i] = t; }
        else a[i] = __builtin_trunc(__dadd_rn(a[i], __dmul_rn(al, a[(i + 3) & 7])));   // (the product is not loop invariant)
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0; for (int i = 0; i < 8; i++) s += a[i];
  out[blockIdx.x * NT + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int MODE, int NT> void run(const char *name, double *d, unsigned long long *c) {
  const int it = 20000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NT>), dim3(256), dim3(NT), 0, 0, it, d, c);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, NT>), dim3(256), dim3(NT), 0, 0, it, d, c);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; (void)hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
  const double insts = (double)it * 64 * (MODE == 3 ? 3 : 1) * (NT / 64) / 4;   // wave-instructions per SIMD
  const double ghz = (double)h[0] / (double)h[1] / 10.0;
  printf("  %-26s %2d waves/SIMD: %.3f ms, %.2f cycles per wave-instruction per SIMD at %.2f GHz\n", name, NT / 256, ms, ms * 1e-3 * ghz * 1e9 / insts, ghz);
}
int main() {
  double *d; unsigned long long *c; (void)hipMalloc(&d, 256 * 2048 * 8); (void)hipMalloc(&c, 16);
  run<0, 1024>("v_mul_f64", d, c); run<1, 1024>("v_add_f64", d, c); run<2, 1024>("v_trunc_f64", d, c); run<3, 1024>("mul+add+trunc (FIR tap)", d, c);
  run<3, 2048 / 2>("mul+add+trunc", d, c);
  return 0;
}
