// Probe: what does a kernel boundary cost on one stream, and does it depend on what the kernel wrote? Each kernel keeps
// every resident wave busy for a fixed time (s_memrealtime, 100 MHz) and optionally writes `mb` MB of output in that time;
// per-launch wall time of back-to-back launches minus the busy time = the boundary.
//   hipcc -O3 --offload-arch=gfx950 -o _bin/gap gap.hip && ./_bin/gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>

template <int MODE>   // 0: no stores, 1: plain stores, 2: nontemporal stores
__global__ __launch_bounds__(256) void k(uint4 *out, size_t n16, unsigned busy_ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nt = (size_t)gridDim.x * 256;
  if (MODE) {
    const uint4 v = make_uint4((unsigned)tid, 1, 2, 3);
    for (size_t i = tid; i < n16; i += nt) {
      if (MODE == 2) { typedef unsigned v4u __attribute__((ext_vector_type(4))); __builtin_nontemporal_store(v4u{v.x, v.y, v.z, v.w}, reinterpret_cast<v4u *>(out + i)); } else out[i] = v;
    }
  }
  while (__builtin_amdgcn_s_memrealtime() - t0 < busy_ticks) __builtin_amdgcn_s_sleep(1);
}

template <int MODE> double run(uint4 *out, size_t bytes, unsigned ticks, int launches) {
  for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 0, 0, out, bytes / 16, ticks);
  (void)hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < launches; i++) hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 0, 0, out, bytes / 16, ticks);
  (void)hipDeviceSynchronize();
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / launches;
}

int main() {
  uint4 *out; (void)hipMalloc(&out, 64u << 20);
  const int L = 3000;
  for (unsigned us : {20u, 80u}) {
    const unsigned ticks = us * 100;
    printf("busy %3u us:  no stores %.2f us per launch   plain 17 MB %.2f   nontemporal 17 MB %.2f   plain 1 MB %.2f   plain 64 MB %.2f\n", us,
           run<0>(out, 0, ticks, L), run<1>(out, 17u << 20, ticks, L), run<2>(out, 17u << 20, ticks, L), run<1>(out, 1u << 20, ticks, L), run<1>(out, 64u << 20, ticks, L));
  }
  return 0;
}
