// Probe: operand/result lane maps of v_mfma_i32_32x32x32_i8 on gfx950, with exact integer data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void k(const v4i *a, const v4i *b, v16i *d) {
  const int l = threadIdx.x;
  v16i c = {0};
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[l], b[l], c, 0, 0, 0);
  d[l] = c;
}

int main() {
  std::vector<signed char> A(32 * 32), B(32 * 32);
  srand(1);
  for (auto &v : A) v = (signed char)(rand() % 255 - 127);
  for (auto &v : B) v = (signed char)(rand() % 255 - 127);
  std::vector<int> ref(32 * 32);
  for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { int s = 0; for (int kk = 0; kk < 32; kk++) s += (int)A[i * 32 + kk] * (int)B[kk * 32 + j]; ref[i * 32 + j] = s; }
  for (int variant = 0; variant < 2; variant++) {
    std::vector<signed char> fa(64 * 16), fb(64 * 16);
    for (int l = 0; l < 64; l++) for (int j = 0; j < 16; j++) {
      const int r = l & 31, h = l >> 5;
      const int kk = variant == 0 ? 16 * h + j : (j < 8 ? 8 * h + j : 16 + 8 * h + (j - 8));
      fa[l * 16 + j] = A[r * 32 + kk];
      fb[l * 16 + j] = B[kk * 32 + r];
    }
    void *da, *db, *dd;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 64 * 64);
    hipMemcpy(da, fa.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, fb.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (const v4i *)da, (const v4i *)db, (v16i *)dd);
    std::vector<int> out(64 * 16);
    hipMemcpy(out.data(), dd, 64 * 64, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int reg = 0; reg < 16; reg++) {
      const int col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
      if (out[l * 16 + reg] != ref[row * 32 + col]) bad++;
    }
    printf("variant %d (k = %s): %d mismatches of 1024\n", variant, variant == 0 ? "16h+j" : "8h+j | 16+8h+j", bad);
  }
  return 0;
}
