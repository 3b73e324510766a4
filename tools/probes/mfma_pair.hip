// Probe: what would TWO sample blocks per wave per K step buy the K1 hot kernel (one tap-fragment operand feeding two
// MFMAs)? Per wave-iteration and slice: 28 i8 32x32x32 MFMAs into 3 accumulators + 168 integer vector instructions that
// consume those accumulators (real dependency, like the kernel's epilogue).
//   MODE 0: the kernel's structure — one slice per iteration, A and B operands from LDS (46 ds_read_b128 per slice)
//   MODE 1: two slices per iteration sharing the A reads (32 reads per slice), 6 accumulators
//   MODE 2: two slices per iteration, A operands resident in 56 registers (18 reads per slice)
// One workgroup of NT threads per CU (NT / 256 waves per SIMD). Random operands. ns per SLICE per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o _bin/mfma_pair mfma_pair.hip && ./_bin/mfma_pair
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define VALU6(P, r) do { x0 = x0 * 3 + P##0[r]; x1 = (x1 << 1) ^ P##1[r]; x2 = x2 + (P##2[r] >> 3); x3 = x3 ^ (x4 >> 1); x4 = x4 + x5; x5 = x5 ^ x0; } while (0)

template <int NT, int MODE, int NVG>
__global__ __launch_bounds__(NT) void k(int iters, const v4i *src, int *out, unsigned long long *clk) {
  __shared__ v4i buf[4096];
  for (int i = threadIdx.x; i < 4096; i += NT) buf[i] = src[i & 2047];
  __syncthreads();
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const v4i *ta = buf + l, *tb = buf + 1024 + (w & 7) * 64 + l;   // tap fragments (shared), the wave's planes
  int x0 = threadIdx.x, x1 = 3, x2 = 5, x3 = 7, x4 = 11, x5 = 13;
  v4i Al[9], Ah[5];
  if (MODE == 2) {
#pragma unroll
    for (int s = 0; s < 9; s++) Al[s] = ta[64 * s];
#pragma unroll
    for (int s = 0; s < 5; s++) Ah[s] = ta[64 * (9 + s)];
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
    asm volatile("" ::: "memory");
    v16i a0 = {0}, a1 = {0}, a2 = {0}, b0 = {0}, b1 = {0}, b2 = {0};
#pragma unroll
    for (int s = 0; s < 9; s++) {
      const bool hi = s >= 2 && s < 7;
      v4i al, ah;
      if (MODE == 2) { al = Al[s]; ah = Ah[hi ? s - 2 : 0]; }
      else { al = ta[64 * s]; ah = al; if (hi) ah = ta[64 * (9 + s - 2)]; }
      const v4i uh = tb[512 * 0 + 16 * s], ul = tb[512 * 1 + 16 * s];
      a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(al, uh, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(al, ul, a2, 0, 0, 0);
      if (hi) {
        a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ah, uh, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ah, ul, a1, 0, 0, 0);
      }
      if (MODE >= 1) {
        const v4i vh = tb[512 * 2 + 16 * s], vl = tb[512 * 3 + 16 * s];
        b1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(al, vh, b1, 0, 0, 0);
        b2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(al, vl, b2, 0, 0, 0);
        if (hi) {
          b0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ah, vh, b0, 0, 0, 0);
          b1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ah, vl, b1, 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" :: "v"(a0), "v"(a1), "v"(a2));   // (the accumulators are results even when no vector work reads them)
    if (MODE >= 1) asm volatile("" :: "v"(b0), "v"(b1), "v"(b2));
#pragma unroll
    for (int g = 0; g < NVG; g++) VALU6(a, g & 15);
    if (MODE >= 1) {
#pragma unroll
      for (int g = 0; g < NVG; g++) VALU6(b, g & 15);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
  out[blockIdx.x * NT + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5;
}

static unsigned long long *g_clk; static double g_ghz;
template <int NT, int MODE, int NVG> float run(int iters, const v4i *src, int *d) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; w++) hipLaunchKernelGGL((k<NT, MODE, NVG>), dim3(256), dim3(NT), 0, 0, iters, src, d, g_clk);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NT, MODE, NVG>), dim3(256), dim3(NT), 0, 0, iters, src, d, g_clk);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  unsigned long long h[2]; (void)hipMemcpy(h, g_clk, 16, hipMemcpyDeviceToHost);
  g_ghz = (double)h[0] / (double)h[1] / 10.0;
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  int *d; (void)hipMalloc(&d, 256 * 1024 * 4);
  v4i *src; (void)hipMalloc(&src, 2048 * 16); (void)hipMalloc(&g_clk, 16);
  { static int h[2048 * 4]; srand(7); for (int i = 0; i < 2048 * 4; i++) h[i] = rand() ^ (rand() << 16); (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice); }
  const int it = 20000;
#define CELL(NT, MODE, NVG) do { const double t = run<NT, MODE, NVG>(it, src, d) * 1e6 / it / (NT / 256) / (MODE ? 2 : 1); printf("  %d wave%s %5.0f ns (%.2f GHz)", NT / 256, NT > 256 ? "s" : " ", t, g_ghz); } while (0)
#define ROW(MODE, NVG, name) do { printf("%-34s", name); CELL(256, MODE, NVG); CELL(512, MODE, NVG); if (MODE == 0) { CELL(768, MODE, NVG); CELL(1024, MODE, NVG); } printf("\n"); } while (0)
  printf("ns per slice (28 MFMA + 168 vector instructions) per SIMD, waves per SIMD:\n");
  ROW(0, 28, "one slice, A+B from LDS");
  ROW(1, 28, "two slices share A reads");
  ROW(2, 28, "two slices, A in registers");
  printf("28 MFMA alone:\n");
  ROW(0, 0, "one slice, A+B from LDS");
  ROW(1, 0, "two slices share A reads");
  ROW(2, 0, "two slices, A in registers");
  return 0;
}
