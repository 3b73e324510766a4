// Probe: ENERGY per operation under the socket's power limit. The K1 kernels run at the power cap (bench.py's
// roofline.power_w ~ 1335 W of 1400), so the time of a launch is its energy / the cap: what counts is joules per MAC, per
// vector instruction, per LDS read — not cycles. Each loop below runs alone on the whole chip (4 waves per SIMD, operands in
// registers unless it says LDS, random data) for ~1.5 s while a host thread reads amdgpu's hwmon files; reported: rate,
// shader clock, socket power, and power above the idle loop per operation.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_energy.bin mfma_energy.hip -lpthread && ./mfma_energy.bin
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <dirent.h>
#include <unistd.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v16i __attribute__((ext_vector_type(16)));

enum { M_IDLE = 0, M_I8_32, M_I8_16, M_I8_32_ZERO, M_I8_32_LDS, M_VALU, M_LDS128, M_I8_32_3ACC, M_I8_16_3ACC };

template <int MODE>
__global__ __launch_bounds__(256, 4) void k(long iters, const v4i *src, int *out) {
  __shared__ v4i buf[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) buf[i] = src[i];
  __syncthreads();
  const int l = threadIdx.x & 63;
  v4i a = src[threadIdx.x], b = src[256 + threadIdx.x], a2 = src[512 + threadIdx.x], b2 = src[768 + threadIdx.x];
  if (MODE == M_I8_32_ZERO) { a = v4i{0, 0, 0, 0}; b = a; a2 = a; b2 = a; }
  v16i c0 = {0}, c1 = {0}, c2 = {0};
  v4i d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
  int x0 = threadIdx.x, x1 = 3, x2 = 5, x3 = 7, x4 = 11, x5 = 13, x6 = 17, x7 = 19;
  for (long it = 0; it < iters; it++) {
    if (MODE == M_I8_32 || MODE == M_I8_32_ZERO) {
#pragma unroll
      for (int j = 0; j < 8; j++) {   // 16 MFMAs, two independent accumulators
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2, b2, c1, 0, 0, 0);
      }
    } else if (MODE == M_I8_32_3ACC) {   // K1's pattern: 4 MFMAs into 3 accumulators per step
#pragma unroll
      for (int j = 0; j < 4; j++) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b2, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2, b, c2, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2, b2, c0, 0, 0, 0);
      }
    } else if (MODE == M_I8_16 || MODE == M_I8_16_3ACC) {
#pragma unroll
      for (int j = 0; j < 8; j++) {   // 32 MFMAs of half the duration: the same MACs as 16 of the 32x32x32 shape
        d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, b2, d1, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b2, d2, 0, 0, 0);
        d3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, b, d3, 0, 0, 0);
      }
    } else if (MODE == M_I8_32_LDS) {
#pragma unroll
      for (int j = 0; j < 8; j++) {   // every operand read from LDS like K1's K loop (2 reads per MFMA here; K1: 46 per 28)
        const v4i oa = buf[l + 64 * ((4 * j) & 31)], ob = buf[l + 64 * ((4 * j + 1) & 31)];
        const v4i oa2 = buf[l + 64 * ((4 * j + 2) & 31)], ob2 = buf[l + 64 * ((4 * j + 3) & 31)];
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(oa, ob, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(oa2, ob2, c1, 0, 0, 0);
      }
    } else if (MODE == M_VALU) {
#pragma unroll
      for (int j = 0; j < 16; j++) {   // 128 integer vector instructions, 8 independent chains
        x0 = x0 * 3 + x4; x1 = (x1 << 1) ^ x5; x2 = x2 + (x6 >> 3); x3 = x3 ^ (x7 >> 1); x4 = x4 + x0; x5 = x5 ^ x1; x6 = x6 * 5 + x2; x7 = x7 + x3;
      }
    } else if (MODE == M_LDS128) {
#pragma unroll
      for (int j = 0; j < 16; j++) {   // 16 ds_read_b128 consumed by one xor each
        const v4i o = buf[(l + 64 * j + (x0 & 1)) & 2047];
        x0 ^= o.x; x1 ^= o.y; x2 ^= o.z; x3 ^= o.w;
      }
    } else {
      x0 += (int)it;   // idle loop: scalar bookkeeping only
      asm volatile("s_sleep 8");
    }
  }
  int s = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
  for (int r = 0; r < 16; r++) s ^= c0[r] ^ c1[r] ^ c2[r];
  s ^= d0.x ^ d1.y ^ d2.z ^ d3.w;
  if (s == 0x12345678) out[0] = s;
}

static std::string g_power, g_sclk;
static void find_hwmon(const char *pci) {
  for (int c = 0; c < 128; c++) {
    char dev[256]; snprintf(dev, sizeof dev, "/sys/class/drm/card%d/device", c);
    char real[512]; ssize_t n = readlink(dev, real, sizeof real - 1);
    if (n <= 0) continue; real[n] = 0;
    if (pci[0] && !strstr(real, pci)) continue;
    char hm[300]; snprintf(hm, sizeof hm, "%s/hwmon", dev);
    DIR *d = opendir(hm); if (!d) continue;
    while (dirent *e = readdir(d)) {
      if (strncmp(e->d_name, "hwmon", 5)) continue;
      g_power = std::string(hm) + "/" + e->d_name + "/power1_input";
      g_sclk = std::string(hm) + "/" + e->d_name + "/freq1_input";
    }
    closedir(d);
    if (!g_power.empty()) return;
  }
}
static double read_num(const std::string &p) { FILE *f = fopen(p.c_str(), "r"); if (!f) return 0; double v = 0; if (fscanf(f, "%lf", &v) != 1) v = 0; fclose(f); return v; }

template <int MODE>
static void run(const char *name, double ops_per_iter, const char *unit, const v4i *src, int *out, double idle_w) {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int grid = 4 * pr.multiProcessorCount;
  long iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // calibrate to ~1.5 s
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, iters, src, out); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, iters, src, out); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  iters = (long)(iters * 1500.0 / (ms > 0.01 ? ms : 0.01));
  std::atomic<bool> stop(false); std::vector<double> pw, ck;
  std::thread t([&] { std::this_thread::sleep_for(std::chrono::milliseconds(300));
                      while (!stop) { pw.push_back(read_num(g_power) / 1e6); ck.push_back(read_num(g_sclk) / 1e6); std::this_thread::sleep_for(std::chrono::milliseconds(20)); } });
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, iters, src, out); hipEventRecord(e1); hipEventSynchronize(e1);
  stop = true; t.join();
  hipEventElapsedTime(&ms, e0, e1);
  double P = 0, F = 0; for (double v : pw) P += v; for (double v : ck) F += v;
  if (!pw.empty()) { P /= pw.size(); F /= ck.size(); }
  const double waves = (double)grid * 4, total = ops_per_iter * iters * waves, rate = total / (ms * 1e-3);
  printf("%-34s %8.1f ms  %9.3f G%s/s  sclk %6.0f MHz  power %6.0f W", name, ms, rate / 1e9, unit, F, P);
  if (idle_w > 0 && ops_per_iter > 0) printf("  (P - idle) / rate = %8.3f nJ per %s", (P - idle_w) / rate * 1e9, unit);
  printf("\n");
  fflush(stdout);
}

int main() {
  char pci[64] = ""; hipDeviceGetPCIBusId(pci, sizeof pci, 0);
  for (char *p = pci; *p; p++) *p = (char)tolower(*p);
  find_hwmon(pci);
  if (g_power.empty()) find_hwmon("");
  printf("device %s  power file %s\n", pci, g_power.c_str());
  std::vector<v4i> h(2048);
  srand(7);
  for (auto &v : h) v = v4i{rand() ^ (rand() << 16), rand() ^ (rand() << 16), rand() ^ (rand() << 16), rand() ^ (rand() << 16)};
  v4i *src; int *out; hipMalloc(&src, 2048 * sizeof(v4i)); hipMalloc(&out, 64);
  hipMemcpy(src, h.data(), 2048 * sizeof(v4i), hipMemcpyHostToDevice);
  std::this_thread::sleep_for(std::chrono::milliseconds(500));
  const double idle_dev = read_num(g_power) / 1e6;
  printf("socket power before any launch: %.0f W\n", idle_dev);
  run<M_IDLE>("idle loop (s_sleep)", 0, "-", src, out, 0);
  const double idle = idle_dev;
  // one 32x32x32 i8 MFMA per wave = 32768 MACs
  run<M_I8_32>("i8 32x32x32, 2 accumulators", 16 * 32768.0, "MAC", src, out, idle);
  run<M_I8_32_3ACC>("i8 32x32x32, K1's 4-into-3 pattern", 16 * 32768.0, "MAC", src, out, idle);
  run<M_I8_32_ZERO>("i8 32x32x32, all-zero operands", 16 * 32768.0, "MAC", src, out, idle);
  run<M_I8_16>("i8 16x16x64, 4 accumulators", 32 * 16384.0, "MAC", src, out, idle);
  run<M_I8_32_LDS>("i8 32x32x32, operands from LDS", 16 * 32768.0, "MAC", src, out, idle);
  run<M_VALU>("integer vector instructions", 128 * 64.0, "lane-op", src, out, idle);
  run<M_LDS128>("ds_read_b128 (+1 xor each)", 16 * 1024.0, "B", src, out, idle);
  return 0;
}
