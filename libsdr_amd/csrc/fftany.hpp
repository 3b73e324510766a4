// fftany.hpp — a batched DFT plan for ANY length, complex<float> or complex<double>, built ONCE (tables resident in
// device memory) and executed many times: the plan behind sdrhip_fft_plan_* (FFTPlan<Scalar>, which the reference builds in
// its constructor: src/fftplan_fftw3.hh:14-36,82-104) and behind the FFT filter for block sizes whose transform does not
// fit one workgroup's LDS or has a prime factor above 13 (FilterNode(size_t block_size), src/filternode.hh:236-245 —
// FFTW plans any n, src/fftplan_fftw3.hh:34-36).
//
// Four forms, chosen at build():
//   LDS       n made of 2, 3, 5, 7, 11, 13 and n <= one workgroup's LDS: fftgen::c2c_kernel (one workgroup per transform)
//   FOURSTEP  such an n beyond the LDS, n = n1 x n2 with both parts in LDS: two strided passes through a temporary
//   CHIRP     n with a larger prime factor, chirp length M = 2^k >= 2n - 1 in LDS: fftgen::bluestein_kernel
//   CHIRPBIG  ... M beyond the LDS (or an n no FOURSTEP split serves): the same chirp transform written as global passes
//             around an inner plan of M points (chirp-in, forward, spectrum product, backward, chirp-out)
// A backward transform is the forward one between two conjugations (done in the kernels' loads and stores), so one set
// of tables serves both directions. Everything is asynchronous on the context's stream; scratch is owned by the plan and
// grown (with a stream synchronisation) when a larger batch arrives. No BASELINE figure rides on these forms — the tuned
// power-of-two kernels of fftconv.hip serve those; what matters here is that every size exists and is right.
#pragma once
#include <memory>

#include "fftgen.hpp"

namespace sdrhip {
namespace fftany {

using fftgen::GT;
using fftgen::gmul;
using fftgen::mk;

// one transform per workgroup in LDS, batch along grid x (and y beyond 65535); CONJ: backward = conj(forward(conj(x)))
template <class T2>
__global__ __launch_bounds__(GT) void lds_c2c_kernel(const fftgen::GenDev<T2> p, const int *perm, int conj, long batch, const T2 *in, T2 *out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T2 *xl = reinterpret_cast<T2 *>(smem_raw);
  const int tid = threadIdx.x, L = p.L;
  const long b = (long)blockIdx.x + (long)blockIdx.y * gridDim.x;
  if (b >= batch) return;
  const T2 *src = in + b * L;
  T2 *dst = out + b * L;
  for (int i = tid; i < L; i += GT) { T2 v = src[i]; if (conj) v.y = -v.y; xl[i] = v; }
  __syncthreads();
  fftgen::forward_dif(xl, p, tid);
  for (int i = tid; i < L; i += GT) { T2 v = xl[i]; if (conj) v.y = -v.y; dst[perm[i]] = v; }   // position i holds frequency perm[i]
}

// one pass of the four-step plan (fftgen.hpp explains the algebra), batched: transform t of batch item z reads
// in[z n + t ibs + i is] and writes out[z n + t obs + k os]; TW: times W_n^(t k) = wa[a] wb[b], t k = a n2 + b.
template <class T2, bool TW>
__global__ __launch_bounds__(GT) void fourstep_pass_kernel(const fftgen::GenDev<T2> p, const int *perm, int conj, long n, const T2 *in, long is, long ibs,
                                                            T2 *out, long os, long obs, const T2 *wa, const T2 *wb, int n2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T2 *xl = reinterpret_cast<T2 *>(smem_raw);
  const int tid = threadIdx.x, L = p.L;
  const long t = blockIdx.x, z = blockIdx.y;
  const T2 *src = in + z * n + t * ibs;
  T2 *dst = out + z * n + t * obs;
  // pass 1 (TW) conjugates what it reads, pass 2 what it writes: conj(F(conj x)) with the twiddles in between untouched
  for (int i = tid; i < L; i += GT) { T2 v = src[(long)i * is]; if (conj && TW) v.y = -v.y; xl[i] = v; }
  __syncthreads();
  fftgen::forward_dif(xl, p, tid);
  for (int i = tid; i < L; i += GT) {
    const int k = perm[i];
    T2 v = xl[i];
    if (TW) {
      const long m = t * (long)k, a_ = m / n2, b_ = m - a_ * n2;
      v = gmul(v, gmul(wa[a_], wb[b_]));
    } else if (conj) {
      v.y = -v.y;
    }
    dst[(long)k * os] = v;
  }
}

// Bluestein in LDS, batched, either direction (fftgen::bluestein_kernel is the forward-only one-shot form)
template <class T2>
__global__ __launch_bounds__(GT) void chirp_lds_kernel(const fftgen::GenDev<T2> p, int n, int conj, long batch, const T2 *w, const T2 *bspec, const T2 *in, T2 *out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T2 *xl = reinterpret_cast<T2 *>(smem_raw);
  const int tid = threadIdx.x, M = p.L;
  const long b = (long)blockIdx.x + (long)blockIdx.y * gridDim.x;
  if (b >= batch) return;
  const T2 *src = in + b * n;
  T2 *dst = out + b * n;
  for (int i = tid; i < M; i += GT) {
    T2 v = mk<T2>(0, 0);
    if (i < n) { v = src[i]; if (conj) v.y = -v.y; v = gmul(v, w[i]); }
    xl[i] = v;
  }
  __syncthreads();
  fftgen::forward_dif(xl, p, tid);
  for (int i = tid; i < M; i += GT) xl[i] = gmul(xl[i], bspec[i]);   // (stored in the forward transform's output order, pre-scaled by 1 / M)
  __syncthreads();
  fftgen::inverse_dit(xl, p, tid);
  for (int i = tid; i < n; i += GT) { T2 v = gmul(xl[i], w[i]); if (conj) v.y = -v.y; dst[i] = v; }
}

// the global passes of the chirp transform around an inner plan of M points
template <class T2>
__global__ void chirp_in_kernel(int n, long M, int conj, const T2 *w, const T2 *in, T2 *y) {   // y[b][i] = x[b][i] w[i], zero beyond n
  const long b = blockIdx.y;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long)gridDim.x * blockDim.x) {
    T2 v = mk<T2>(0, 0);
    if (i < n) { v = in[b * n + i]; if (conj) v.y = -v.y; v = gmul(v, w[i]); }
    y[b * M + i] = v;
  }
}
template <class T2>
__global__ void spectrum_mul_kernel(long M, const T2 *spec, T2 *y) {   // y[b][k] *= spec[k] (natural order)
  const long b = blockIdx.y;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long)gridDim.x * blockDim.x) y[b * M + i] = gmul(y[b * M + i], spec[i]);
}
template <class T2>
__global__ void chirp_out_kernel(int n, long M, int conj, const T2 *w, const T2 *y, T2 *out) {
  const long b = blockIdx.y;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    T2 v = gmul(y[b * M + i], w[i]);
    if (conj) v.y = -v.y;
    out[b * n + i] = v;
  }
}

template <class K>
inline void allow_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024)
    SDRHIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

template <class T2>
struct AnyFft {
  typedef typename fftgen::Real<T2>::type R;
  enum Kind { LDS, FOURSTEP, CHIRP, CHIRPBIG };
  sdrhip_ctx *ctx = nullptr;
  long n = 0, M = 0, n1 = 0, n2 = 0;
  Kind kind = LDS;
  fftgen::GenPlan<T2> p1, p2;
  DevBuf<T2> wa, wb;         // FOURSTEP: W_n1^a and W_n^b
  DevBuf<T2> w, bspec;       // CHIRP*: the chirp exp(-pi i j^2 / n) and the spectrum of its conjugate (pre-scaled by 1 / M)
  std::unique_ptr< AnyFft<T2> > inner;   // CHIRPBIG: the M-point plan
  DevBuf<T2> tmp;            // FOURSTEP: batch x n; CHIRPBIG: 2 x batch x M
  long tmp_batch = 0;

  static long max_lds_points() { return (long)(128 * 1024 / sizeof(T2)); }
  const char *kind_name() const { return kind == LDS ? "lds" : kind == FOURSTEP ? "four-step" : kind == CHIRP ? "chirp" : "chirp over four-step"; }

  // the largest plannable divisor of n whose cofactor is plannable too (0: none)
  static long split(long n) {
    const long maxL = max_lds_points();
    std::vector<int> rx;
    if (!fftgen::GenPlan<T2>::factor_long(n)) return 0;
    for (long d = maxL; d >= 2; d--) {
      if (n % d || n / d > maxL) continue;
      if (fftgen::GenPlan<T2>::factor((int)d, rx, nullptr) && fftgen::GenPlan<T2>::factor((int)(n / d), rx, nullptr)) return d;
    }
    return 0;
  }

  void build(sdrhip_ctx *ctx_, long n_) {
    ctx = ctx_; n = n_;
    SDRHIP_REQUIRE(n >= 1 && n <= (1L << 27), SDRHIP_E_UNSUPPORTED, "FFT size %ld outside [1, 2^27]", n);
    const long maxL = max_lds_points();
    std::vector<int> rx;
    const bool smooth = fftgen::GenPlan<T2>::factor_long(n);
    if (smooth && n <= maxL && fftgen::GenPlan<T2>::factor((int)n, rx, nullptr)) {
      kind = LDS;
      p1.build(ctx, (int)n, (int)maxL);
      allow_lds(lds_c2c_kernel<T2>, p1.lds_bytes());
      return;
    }
    if (smooth && (n1 = split(n)) != 0) {
      kind = FOURSTEP; n2 = n / n1;
      p1.build(ctx, (int)n1, (int)maxL); p2.build(ctx, (int)n2, (int)maxL);
      const long double PI2 = 2.0L * 3.14159265358979323846264338327950288L;
      std::vector<T2> ha(n1), hb(n2);   // W_n^(a n2) = W_n1^a and W_n^b
      for (long a = 0; a < n1; a++) { const long double ang = -PI2 * (long double)a / (long double)n1; ha[a].x = (R)cosl(ang); ha[a].y = (R)sinl(ang); }
      for (long b = 0; b < n2; b++) { const long double ang = -PI2 * (long double)b / (long double)n; hb[b].x = (R)cosl(ang); hb[b].y = (R)sinl(ang); }
      wa.alloc(n1); wa.upload(ha.data(), n1, ctx->stream);
      wb.alloc(n2); wb.upload(hb.data(), n2, ctx->stream);
      allow_lds(fourstep_pass_kernel<T2, true>, p1.lds_bytes());
      allow_lds(fourstep_pass_kernel<T2, false>, p2.lds_bytes());
      return;
    }
    // a prime factor above 13 (or no split): Bluestein over M = 2^k >= 2n - 1
    M = 1; while (M < 2 * n - 1) M <<= 1;
    SDRHIP_REQUIRE(M <= maxL * maxL, SDRHIP_E_UNSUPPORTED, "FFT size %ld needs a chirp transform of %ld points: beyond the four-step plan (%ld)", n, M, maxL * maxL);
    const long double PI = 3.14159265358979323846264338327950288L;
    std::vector<T2> hw(n);
    std::vector< std::complex<double> > b(M, std::complex<double>(0, 0));
    for (long j = 0; j < n; j++) {
      const long double ang = -PI * (long double)((j * j) % (2 * n)) / (long double)n;   // (j^2 mod 2n: the phase stays exact)
      hw[j].x = (R)cosl(ang); hw[j].y = (R)sinl(ang);
      const std::complex<double> cw((double)cosl(ang), -(double)sinl(ang));   // conj(w[j])
      b[j] = cw;
      if (j) b[M - j] = cw;
    }
    host_fft_pow2(b);
    w.alloc(n); w.upload(hw.data(), n, ctx->stream);
    std::vector<T2> bs(M);
    if (M <= maxL) {
      kind = CHIRP;
      p1.build(ctx, (int)M, (int)maxL);
      for (long pos = 0; pos < M; pos++) { const std::complex<double> v = b[p1.perm[pos]] / (double)M; bs[pos].x = (R)v.real(); bs[pos].y = (R)v.imag(); }
      allow_lds(chirp_lds_kernel<T2>, p1.lds_bytes());
    } else {
      kind = CHIRPBIG;
      for (long k = 0; k < M; k++) { const std::complex<double> v = b[k] / (double)M; bs[k].x = (R)v.real(); bs[k].y = (R)v.imag(); }
      inner.reset(new AnyFft<T2>());
      inner->build(ctx, M);
    }
    bspec.alloc(M); bspec.upload(bs.data(), M, ctx->stream);
  }

  // iterative radix-2 in double on the host (the chirp's spectrum, one-off; M is a power of two)
  static void host_fft_pow2(std::vector< std::complex<double> > &a) {
    const size_t L = a.size();
    for (size_t i = 1, j = 0; i < L; i++) {
      size_t bit = L >> 1;
      for (; j & bit; bit >>= 1) j ^= bit;
      j ^= bit;
      if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= L; len <<= 1)
      for (size_t k = 0; k < len / 2; k++) {
        const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)len;
        const std::complex<double> wk((double)cosl(ang), (double)sinl(ang));
        for (size_t s = 0; s < L; s += len) {
          const std::complex<double> u = a[s + k], t = wk * a[s + k + len / 2];
          a[s + k] = u + t; a[s + k + len / 2] = u - t;
        }
      }
  }

  void reserve(long batch) {
    if (kind == CHIRPBIG) inner->reserve(std::min<long>(batch, 32768));
    if ((kind != FOURSTEP && kind != CHIRPBIG) || batch <= tmp_batch) return;
    SDRHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));   // (launches in flight still use the old scratch)
    tmp.alloc((size_t)batch * (kind == FOURSTEP ? (size_t)n : 2 * (size_t)M));
    tmp_batch = batch;
  }

  // batch transforms of n points, contiguous; sign -1 forward, +1 backward (unnormalised); in == out is allowed
  void exec(int sign, long batch, const T2 *in, T2 *out) {
    if (batch <= 0) return;
    ctx->use();
    reserve(batch);
    const int conj = sign > 0 ? 1 : 0;
    hipStream_t st = ctx->stream;
    auto grid2 = [](long b) { const long gx = b < 32768 ? b : 32768; return dim3((unsigned)gx, (unsigned)((b + gx - 1) / gx)); };
    switch (kind) {
      case LDS:
        hipLaunchKernelGGL(lds_c2c_kernel<T2>, grid2(batch), dim3(GT), p1.lds_bytes(), st, p1.dev, p1.perm_d.p, conj, batch, in, out);
        break;
      case FOURSTEP:
        for (long z0 = 0; z0 < batch; z0 += 32768) {
          const long zb = std::min<long>(32768, batch - z0);
          // pass 1: column j2 (stride n2) -> A[k1][j2] = tmp[k1 n2 + j2], twiddled; pass 2: row k1 of A -> X[k1 + n1 k2]
          hipLaunchKernelGGL((fourstep_pass_kernel<T2, true>), dim3((unsigned)n2, (unsigned)zb), dim3(GT), p1.lds_bytes(), st, p1.dev, p1.perm_d.p, conj, n,
                             in + z0 * n, n2, 1L, tmp.p + z0 * n, n2, 1L, wa.p, wb.p, (int)n2);
          hipLaunchKernelGGL((fourstep_pass_kernel<T2, false>), dim3((unsigned)n1, (unsigned)zb), dim3(GT), p2.lds_bytes(), st, p2.dev, p2.perm_d.p, conj, n,
                             tmp.p + z0 * n, 1L, n2, out + z0 * n, n1, 1L, wa.p, wb.p, (int)n2);
        }
        break;
      case CHIRP:
        hipLaunchKernelGGL(chirp_lds_kernel<T2>, grid2(batch), dim3(GT), p1.lds_bytes(), st, p1.dev, (int)n, conj, batch, w.p, bspec.p, in, out);
        break;
      case CHIRPBIG:
        for (long z0 = 0; z0 < batch; z0 += 32768) {
          const long zb = std::min<long>(32768, batch - z0);
          T2 *y = tmp.p + 2 * z0 * M, *y2 = y + zb * M;
          const unsigned gx = (unsigned)std::min<long>((M + 255) / 256, 4096);
          hipLaunchKernelGGL(chirp_in_kernel<T2>, dim3(gx, (unsigned)zb), dim3(256), 0, st, (int)n, M, conj, w.p, in + z0 * n, y);
          inner->exec(-1, zb, y, y2);
          hipLaunchKernelGGL(spectrum_mul_kernel<T2>, dim3(gx, (unsigned)zb), dim3(256), 0, st, M, bspec.p, y2);
          inner->exec(+1, zb, y2, y);
          hipLaunchKernelGGL(chirp_out_kernel<T2>, dim3(gx, (unsigned)zb), dim3(256), 0, st, (int)n, M, conj, w.p, y, out + z0 * n);
        }
        break;
    }
    SDRHIP_CHECK_HIP(hipGetLastError());
  }
};

}  // namespace fftany
}  // namespace sdrhip
