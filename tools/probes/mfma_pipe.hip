// Probe: does an IN-WAVE software pipeline (the MFMA chain of slice i+1 interleaved, instruction by instruction, with
// the vector epilogue of slice i) beat the K1 hot kernel's structure (each wave alternates a matrix phase and a vector
// phase, 4 waves per SIMD overlap by chance)? Per wave-iteration: NM i8 32x32x32 MFMAs into 3 accumulators (operands
// re-read from LDS by ds_read_b128 like the K1 K loop, 32 reads per 28 MFMAs) and NV integer vector instructions that
// consume the PREVIOUS iteration's accumulators. One workgroup of W*4 waves per CU. Random operands (the clock the chip
// holds depends on the data).
//   hipcc -O3 --offload-arch=gfx950 -o mfma_pipe.bin mfma_pipe.hip && ./mfma_pipe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// six independent integer chains fed by 2 accumulator registers: 6 vector instructions
#define VALU6(P, r) do { x0 = x0 * 3 + P##0[r]; x1 = (x1 << 1) ^ P##1[r]; x2 = x2 + (P##2[r] >> 3); x3 = x3 ^ (x4 >> 1); x4 = x4 + x5; x5 = x5 ^ x0; } while (0)

template <int NT, int NM, int NVG, int MODE>   // NVG groups of 6 vector instructions; MODE 0 = alternate phases, 1 = interleave
__global__ __launch_bounds__(NT) void k(int iters, const v4i *src, int *out, unsigned long long *clk) {
  __shared__ v4i buf[2048];
  for (int i = threadIdx.x; i < 2048; i += NT) buf[i] = src[i];
  __syncthreads();
  v16i c0 = {0}, c1 = {0}, c2 = {0}, p0 = {0}, p1 = {0}, p2 = {0};
  int x0 = threadIdx.x, x1 = 3, x2 = 5, x3 = 7, x4 = 11, x5 = 13;
  const int l = threadIdx.x & 63;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  auto body = [&](auto par_, int i) __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    constexpr int PAR = decltype(par_)::value;
    v16i &a0 = PAR ? p0 : c0, &a1 = PAR ? p1 : c1, &a2 = PAR ? p2 : c2;   // written by this iteration's MFMAs
    v16i &q0 = PAR ? c0 : p0, &q1 = PAR ? c1 : p1, &q2 = PAR ? c2 : p2;   // read by this iteration's vector work
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < NM / 4; j++) {
        const v4i oa = buf[l + 64 * ((4 * j) & 31)], ob = buf[l + 64 * ((4 * j + 1) & 31)];
        const v4i oa2 = buf[l + 64 * ((4 * j + 2) & 31)], ob2 = buf[l + 64 * ((4 * j + 3) & 31)];
        a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(oa, ob, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(oa, ob2, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(oa2, ob, a2, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(oa2, ob2, a1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < NVG; g++) VALU6(q, g & 15);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      // per MFMA: NVG*6/NM vector instructions behind it, operand reads one quad ahead
      constexpr int GP = NM > 0 ? (NVG + NM - 1) / NM : 0;   // groups of 6 per MFMA (1 for 28 MFMAs / 28 groups)
      v4i oa = buf[l], ob = buf[l + 64], oa2 = buf[l + 128], ob2 = buf[l + 192];
      int g = 0;
#pragma unroll
      for (int j = 0; j < NM / 4; j++) {
        v4i na = oa, nb = ob, na2 = oa2, nb2 = ob2;
        if (j + 1 < NM / 4) {
          na = buf[l + 64 * ((4 * j + 4) & 31)]; nb = buf[l + 64 * ((4 * j + 5) & 31)];
          na2 = buf[l + 64 * ((4 * j + 6) & 31)]; nb2 = buf[l + 64 * ((4 * j + 7) & 31)];
        }
        a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(oa, ob, a0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < GP; t++) if (g < NVG) { VALU6(q, g & 15); g++; }
        __builtin_amdgcn_sched_barrier(0);
        a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(oa, ob2, a1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < GP; t++) if (g < NVG) { VALU6(q, g & 15); g++; }
        __builtin_amdgcn_sched_barrier(0);
        a2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(oa2, ob, a2, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < GP; t++) if (g < NVG) { VALU6(q, g & 15); g++; }
        __builtin_amdgcn_sched_barrier(0);
        a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(oa2, ob2, a1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < GP; t++) if (g < NVG) { VALU6(q, g & 15); g++; }
        __builtin_amdgcn_sched_barrier(0);
        oa = na; ob = nb; oa2 = na2; ob2 = nb2;
      }
    }
  };
  for (int i = 0; i < iters; i += 2) {
    body(std::integral_constant<int, 0>{}, i);
    body(std::integral_constant<int, 1>{}, i + 1);
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
  out[blockIdx.x * NT + threadIdx.x] = c0[0] + c1[1] + c2[2] + p0[3] + p1[4] + p2[5] + x0 + x1 + x2 + x3 + x4 + x5;
}

static unsigned long long *g_clk; static double g_ghz;
template <int NT, int NM, int NVG, int MODE> float run(int iters, const v4i *src, int *d) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; w++) hipLaunchKernelGGL((k<NT, NM, NVG, MODE>), dim3(256), dim3(NT), 0, 0, iters, src, d, g_clk);   // warm up: the clock settles
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NT, NM, NVG, MODE>), dim3(256), dim3(NT), 0, 0, iters, src, d, g_clk);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  unsigned long long h[2]; (void)hipMemcpy(h, g_clk, 16, hipMemcpyDeviceToHost);
  g_ghz = (double)h[0] / (double)h[1] / 10.0;   // shader cycles per 100 MHz tick: the clock this kernel held
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  int *d; (void)hipMalloc(&d, 256 * 1024 * 4);
  v4i *src; (void)hipMalloc(&src, 2048 * 16); (void)hipMalloc(&g_clk, 16);
  { int h[2048 * 4]; srand(7); for (int i = 0; i < 2048 * 4; i++) h[i] = rand() ^ (rand() << 16); (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice); }
  const int it = 20000;
#define ROW(NM, NVG, MODE) do { const double a1 = run<256, NM, NVG, MODE>(it, src, d) * 1e6 / it, g1 = g_ghz, a2 = run<512, NM, NVG, MODE>(it, src, d) * 1e6 / it / 2, g2 = g_ghz, \
    a3 = run<768, NM, NVG, MODE>(it, src, d) * 1e6 / it / 3, g3 = g_ghz; \
    printf("  %-12s 1 wave/SIMD %5.0f ns (%.2f GHz)   2 waves %5.0f (%.2f)   3 waves %5.0f (%.2f)\n", MODE ? "interleaved" : "alternating", a1, g1, a2, g2, a3, g3); } while (0)
  printf("ns per wave-iteration per SIMD; 28 MFMA (32 LDS operand reads) + 168 vector instructions (K1 at 127 taps):\n");
  ROW(28, 28, 0); ROW(28, 28, 1);
  printf("8 MFMA + 168 vector instructions (K1 at 16 taps):\n");
  ROW(8, 28, 0); ROW(8, 28, 1);
  printf("28 MFMA + 112 vector instructions:\n");
  ROW(28, 19, 0); ROW(28, 19, 1);
  printf("28 MFMA alone:\n");
  ROW(28, 0, 0);
  printf("168 vector instructions alone:\n");
  ROW(0, 28, 0);
  return 0;
}
