// fftgen.hpp — the general in-LDS FFT: any size whose prime factors are 2, 3, 5, 7, 11 or 13, complex<float> or complex<double>.
//
// Replaces (reference, file:line): FFTPlan<float> / FFTPlan<double> / FFT::exec for ANY buffer size
// (src/fftplan_fftw3.hh:12-142 plans whatever `in.size()` is; src/fftplan.hh:14-36) and with them FilterSink / FilterSource
// for any block size and either Scalar (src/filternode.hh:38-47,81-88,164-181,230-245: `FilterNode(size_t block_size=1024)`,
// `template <class Scalar>`).
//
// The tuned kernels of fftconv.hip serve the power-of-two complex<float> plans (every BASELINE configuration); this file
// is the same algebra written for generality: a forward decimation-in-frequency transform (natural order in,
// digit-reversed order out) as a list of passes of radix r in {4, 2, 3, 5, 7, 11, 13}, the mirrored decimation-in-time
// inverse, the spectrum stored in the forward transform's output order so that no reordering pass runs. One workgroup owns
// one transform in LDS. Butterflies of radix 3 and above are evaluated as the r x r DFT they are (r^2 complex products with
// the roots taken from the plan's own table) — these sizes carry no BASELINE figure; what matters is that they exist and
// are right (<= 1e-5 of numpy for float, <= 1e-12 for double: tests/test_gpu_parity.py::test_fft_any_size*).
#pragma once
#include "sdrhip_internal.hpp"

namespace sdrhip {
namespace fftgen {

constexpr int GT = 256;         // lanes per workgroup: one wave per SIMD, so that a radix-13 butterfly in double (150 registers) does not spill
constexpr int GMAX_PASS = 24;

template <class T2> struct Real;
template <> struct Real<float2> { typedef float type; };
template <> struct Real<double2> { typedef double type; };

template <class T2> __device__ __forceinline__ T2 mk(typename Real<T2>::type x, typename Real<T2>::type y) { T2 r; r.x = x; r.y = y; return r; }
template <class T2> __device__ __forceinline__ T2 gadd(T2 a, T2 b) { return mk<T2>(a.x + b.x, a.y + b.y); }
template <class T2> __device__ __forceinline__ T2 gsub(T2 a, T2 b) { return mk<T2>(a.x - b.x, a.y - b.y); }
template <class T2> __device__ __forceinline__ T2 gmul(T2 a, T2 b) { return mk<T2>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
template <class T2> __device__ __forceinline__ T2 gmulc(T2 a, T2 b) { return mk<T2>(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a * conj(b)

template <class T2>
struct GenDev {
  int L, npass;
  int radix[GMAX_PASS];   // forward pass order (DIF); the inverse walks it backwards
  const T2 *W;            // W[t] = exp(-2 pi i t / L), t < L
};

// r-point DFT of v[0..R) in place, X[m] = sum_k v[k] w^(k m), w = exp(SIGN 2 pi i / R) = W[L / R] or its conjugate
template <class T2, int R, int SIGN>
__device__ __forceinline__ void dft_small(T2 *v, const GenDev<T2> &p) {
  if (R == 2) {
    const T2 a = v[0], b = v[1];
    v[0] = gadd(a, b); v[1] = gsub(a, b);
  } else if (R == 4) {
    const T2 t0 = gadd(v[0], v[2]), t1 = gsub(v[0], v[2]), t2 = gadd(v[1], v[3]), d = gsub(v[1], v[3]);
    // X1 = t1 + (SIGN i) d, X3 = t1 - (SIGN i) d
    const T2 id = SIGN < 0 ? mk<T2>(d.y, -d.x) : mk<T2>(-d.y, d.x);
    v[0] = gadd(t0, t2); v[2] = gsub(t0, t2); v[1] = gadd(t1, id); v[3] = gsub(t1, id);
  } else {
    T2 root[R];
    const int st = p.L / R;
#pragma unroll
    for (int q = 0; q < R; q++) { root[q] = p.W[q * st]; if (SIGN > 0) root[q].y = -root[q].y; }
    T2 X[R];
#pragma unroll
    for (int m = 0; m < R; m++) {
      T2 acc = v[0];
#pragma unroll
      for (int k = 1; k < R; k++) acc = gadd(acc, gmul(v[k], root[(k * m) % R]));
      X[m] = acc;
    }
#pragma unroll
    for (int m = 0; m < R; m++) v[m] = X[m];
  }
}

// one DIF pass of radix R over a transform of length L whose current sub-length is n (s = n / R, tw = L / n)
template <class T2, int R>
__device__ __forceinline__ void pass_dif(T2 *x, const GenDev<T2> &p, int n, int tid) {
  const int s = n / R, tw = p.L / n;
  for (int b = tid; b < p.L / R; b += GT) {
    const int blk = b / s, j = b - blk * s, base = blk * n + j;
    T2 v[R];
#pragma unroll
    for (int k = 0; k < R; k++) v[k] = x[base + k * s];
    dft_small<T2, R, -1>(v, p);
    x[base] = v[0];
#pragma unroll
    for (int m = 1; m < R; m++) x[base + m * s] = s > 1 ? gmul(v[m], p.W[j * tw * m]) : v[m];   // j tw m < s tw R = L
  }
}
template <class T2, int R>
__device__ __forceinline__ void pass_dit(T2 *x, const GenDev<T2> &p, int s, int tid) {   // s = the sub-length before this pass
  const int n = s * R, tw = p.L / n;
  for (int b = tid; b < p.L / R; b += GT) {
    const int blk = b / s, j = b - blk * s, base = blk * n + j;
    T2 v[R];
    v[0] = x[base];
#pragma unroll
    for (int k = 1; k < R; k++) v[k] = s > 1 ? gmulc(x[base + k * s], p.W[j * tw * k]) : x[base + k * s];
    dft_small<T2, R, 1>(v, p);
#pragma unroll
    for (int m = 0; m < R; m++) x[base + m * s] = v[m];
  }
}

#define SDRHIP_GEN_RADIX_SWITCH(r_, CALL)                                                              \
  switch (r_) {                                                                                        \
    case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break; case 5: CALL(5); break;    \
    case 7: CALL(7); break; case 11: CALL(11); break; default: CALL(13); break;                        \
  }

// forward, decimation in frequency: natural order in, digit-reversed order out (position -> frequency: the plan's perm)
template <class T2>
__device__ void forward_dif(T2 *x, const GenDev<T2> &p, int tid) {
  int n = p.L;
  for (int pass = 0; pass < p.npass; pass++) {
    const int r = p.radix[pass];
#define SDRHIP_GEN_CALL(R_) pass_dif<T2, R_>(x, p, n, tid)
    SDRHIP_GEN_RADIX_SWITCH(r, SDRHIP_GEN_CALL)
#undef SDRHIP_GEN_CALL
    __syncthreads();
    n /= r;
  }
}
// backward (unnormalised), decimation in time: digit-reversed order in, natural order out
template <class T2>
__device__ void inverse_dit(T2 *x, const GenDev<T2> &p, int tid) {
  int s = 1;
  for (int pass = p.npass - 1; pass >= 0; pass--) {
    const int r = p.radix[pass];
#define SDRHIP_GEN_CALL(R_) pass_dit<T2, R_>(x, p, s, tid)
    SDRHIP_GEN_RADIX_SWITCH(r, SDRHIP_GEN_CALL)
#undef SDRHIP_GEN_CALL
    __syncthreads();
    s *= r;
  }
}

// plain batched DFT (FFT::exec / FFTPlan): one workgroup per transform
template <class T2>
__global__ __launch_bounds__(GT) void c2c_kernel(const GenDev<T2> p, const int *perm, int sign, const T2 *in, T2 *out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T2 *xl = reinterpret_cast<T2 *>(smem_raw);
  const int tid = threadIdx.x, L = p.L;
  const T2 *src = in + (long)blockIdx.x * L;
  T2 *dst = out + (long)blockIdx.x * L;
  if (sign < 0) {
    for (int i = tid; i < L; i += GT) xl[i] = src[i];
    __syncthreads();
    forward_dif(xl, p, tid);
    for (int i = tid; i < L; i += GT) dst[perm[i]] = xl[i];   // position i holds frequency perm[i]
  } else {
    for (int i = tid; i < L; i += GT) xl[i] = src[perm[i]];
    __syncthreads();
    inverse_dit(xl, p, tid);
    for (int i = tid; i < L; i += GT) dst[i] = xl[i];
  }
}

// One pass of the FOUR-STEP plan for transforms longer than one workgroup's LDS holds, n = n1 * n2:
//   X[k1 + n1 k2] = sum_j2 W_n^(j2 k1) ( sum_j1 x[j1 n2 + j2] W_n1^(j1 k1) ) W_n2^(j2 k2)
// pass 1: n2 transforms of n1 points over the stride-n2 columns, times the twiddle W_n^(j2 k1) = Wa[a] Wb[b] with
// j2 k1 = a n2 + b (two small tables instead of n roots), written transposed as A[k1][j2]; pass 2: n1 transforms of n2
// points over A's rows, written with stride n1. This kernel is either pass: transform t reads in[t ibs + i is], writes
// out[t obs + k os]; TW: multiply output k of transform t by the twiddle (t k) first. (No BASELINE figure: generality.)
template <class T2, bool TW>
__global__ __launch_bounds__(GT) void strided_c2c_kernel(const GenDev<T2> p, const int *perm, int sign, const T2 *in, long is, long ibs,
                                                          T2 *out, long os, long obs, const T2 *wa, const T2 *wb, int n2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T2 *xl = reinterpret_cast<T2 *>(smem_raw);
  const int tid = threadIdx.x, L = p.L;
  const long t = blockIdx.x;
  const T2 *src = in + t * ibs;
  T2 *dst = out + t * obs;
  auto put = [&](int k, T2 v) {
    if (TW) {   // W_n^(sign t k): t k = a n2 + b, a < n1 (k < n1, t < n2), b < n2
      const long m = t * (long)k, a_ = m / n2, b_ = m - a_ * n2;
      T2 w = gmul(wa[a_], wb[b_]);
      if (sign > 0) w.y = -w.y;
      v = gmul(v, w);
    }
    dst[(long)k * os] = v;
  };
  if (sign < 0) {
    for (int i = tid; i < L; i += GT) xl[i] = src[(long)i * is];
    __syncthreads();
    forward_dif(xl, p, tid);
    for (int i = tid; i < L; i += GT) put(perm[i], xl[i]);   // position i holds frequency perm[i]
  } else {
    for (int i = tid; i < L; i += GT) xl[i] = src[(long)perm[i] * is];
    __syncthreads();
    inverse_dit(xl, p, tid);
    for (int i = tid; i < L; i += GT) put(i, xl[i]);
  }
}

// Bluestein's chirp transform for lengths with a prime factor above 13 (the reference plans them like any other size):
//   X[k] = w[k] * sum_j (x[j] w[j]) conj(w)[k - j],   w[j] = exp(sign * pi * i * j^2 / n)
// i.e. one circular convolution of length M >= 2n - 1 (a power of two) between the chirped input and the conjugate chirp,
// evaluated with the plan's own in-LDS FFT: forward DIF, product with the chirp's spectrum (made on the host, stored in the
// forward transform's output order, pre-scaled by 1 / M), inverse DIT, and the output chirp. One workgroup per transform.
template <class T2>
__global__ __launch_bounds__(GT) void bluestein_kernel(const GenDev<T2> p, int n, const T2 *w, const T2 *bspec, const T2 *in, T2 *out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T2 *xl = reinterpret_cast<T2 *>(smem_raw);
  const int tid = threadIdx.x, M = p.L;
  const T2 *src = in + (long)blockIdx.x * n;
  T2 *dst = out + (long)blockIdx.x * n;
  for (int i = tid; i < M; i += GT) xl[i] = i < n ? gmul(src[i], w[i]) : mk<T2>(0, 0);
  __syncthreads();
  forward_dif(xl, p, tid);
  for (int i = tid; i < M; i += GT) xl[i] = gmul(xl[i], bspec[i]);
  __syncthreads();
  inverse_dit(xl, p, tid);
  for (int i = tid; i < n; i += GT) dst[i] = gmul(xl[i], w[i]);
}

// FFT convolution by overlap-save, as fftconv.hip's fftconv_kernel: block b transforms the L samples ending at its last
// output and keeps the last `hop` results. Several bands share the forward transform when two LDS images fit (`two`);
// otherwise the block is transformed once per band.
template <class T2>
struct GenConvArgs {
  GenDev<T2> fft;
  const T2 *in; long in_stride;
  const T2 *hist; int HH;            // HH = L - hop samples preceding the call
  T2 *hist_new;                      // the channel's last block also writes the history of the next call
  const T2 *Kp;                      // spectra (band b at Kp + b*L), forward-output order, pre-scaled by 1/L
  T2 *out; long out_stride;
  int N, hop, nb; long out_band;
  int two;                           // a second LDS image holds the product (bands reuse the forward transform)
};

template <class T2>
__global__ __launch_bounds__(GT) void conv_kernel(const GenConvArgs<T2> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T2 *xl = reinterpret_cast<T2 *>(smem_raw);
  const int c = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x, L = a.fft.L;
  const int first = blk * a.hop - a.HH;   // call-relative index of xl[0]
  // (U loads in flight per lane: a plain `for (i = tid; ...) xl[i] = load` waits for every load before the next is issued — a
  // dozen dependent trips to memory per block; round 5 found that in the four-step passes and here)
  constexpr int U = 8;
  auto load = [&]() {
    for (int i0 = tid; i0 < L; i0 += GT * U) {
      T2 v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int i = min(i0 + u * GT, L - 1), rel = first + i, h = a.HH + rel;
        const T2 *ptr = rel >= 0 ? a.in + (long)c * a.in_stride + min(rel, a.N - 1) : a.hist + (long)c * a.HH + max(h, 0);
        v[u] = *ptr;
        if (rel >= a.N || h < 0) v[u] = mk<T2>(0, 0);
      }
#pragma unroll
      for (int u = 0; u < U; u++) if (i0 + u * GT < L) xl[i0 + u * GT] = v[u];
    }
    __syncthreads();
    forward_dif(xl, a.fft, tid);
  };
  T2 *xw = a.two ? xl + L : xl;
  if (a.two) load();
  for (int band = 0; band < a.nb; band++) {
    if (!a.two) load();
    for (int i0 = tid; i0 < L; i0 += GT * U) {
      T2 kq[U];
#pragma unroll
      for (int u = 0; u < U; u++) kq[u] = a.Kp[(long)band * L + min(i0 + u * GT, L - 1)];
#pragma unroll
      for (int u = 0; u < U; u++) if (i0 + u * GT < L) xw[i0 + u * GT] = gmul(xl[i0 + u * GT], kq[u]);
    }
    __syncthreads();
    inverse_dit(xw, a.fft, tid);
    const int o0 = blk * a.hop;
    for (int i = tid; i < a.hop; i += GT) {
      const int o = o0 + i;
      if (o < a.N) a.out[(long)band * a.out_band + (long)c * a.out_stride + o] = xw[a.HH + i];
    }
    __syncthreads();
  }
  if (a.hist_new != nullptr && blk == (int)gridDim.x - 1) {
    for (int k = tid; k < a.HH; k += GT) {
      const long qq = (long)a.N + k;
      a.hist_new[(long)c * a.HH + k] = qq < a.HH ? a.hist[(long)c * a.HH + qq] : a.in[(long)c * a.in_stride + (qq - a.HH)];
    }
  }
}

// host side of a plan: radix list, root table (made in long double), output permutation
template <class T2>
struct GenPlan {
  int L = 0;
  GenDev<T2> dev{};
  DevBuf<T2> W;
  DevBuf<int> perm_d;
  std::vector<int> perm;   // position -> frequency index after the forward DIF

  static bool factor_long(long n) {   // made of 2, 3, 5, 7, 11, 13 only?
    const long primes[] = {2, 3, 5, 7, 11, 13};
    for (long q : primes) while (n > 1 && n % q == 0) n /= q;
    return n == 1;
  }
  static bool factor(int L, std::vector<int> &radix, int *bad) {
    radix.clear();
    int n = L;
    const int primes[] = {13, 11, 7, 5, 3};
    for (int q : primes) while (n % q == 0) { radix.push_back(q); n /= q; }
    while (n % 4 == 0) { radix.push_back(4); n /= 4; }
    if (n % 2 == 0) { radix.push_back(2); n /= 2; }
    if (bad) { *bad = n; for (int q = 17; q * q <= n; q += 2) if (n % q == 0) { *bad = q; break; } }   // (the smallest prime factor left)
    return n == 1 && (int)radix.size() <= GMAX_PASS;
  }
  void build(sdrhip_ctx *ctx, int L_, int max_L) {
    std::vector<int> rx;
    int bad = 1;
    SDRHIP_REQUIRE(L_ >= 1 && L_ <= max_L, SDRHIP_E_UNSUPPORTED, "FFT size %d outside [1,%d] (one transform lives in one workgroup's LDS)", L_, max_L);
    SDRHIP_REQUIRE(factor(L_, rx, &bad), SDRHIP_E_UNSUPPORTED,
                   "FFT size %d has the prime factor %d: the device plans sizes made of 2, 3, 5, 7, 11 and 13", L_, bad);
    L = L_; dev.L = L; dev.npass = (int)rx.size();
    for (int q = 0; q < dev.npass; q++) dev.radix[q] = rx[q];
    typedef typename Real<T2>::type R;
    std::vector<T2> w(L);
    for (int t = 0; t < L; t++) {
      const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)t / (long double)L;
      w[t].x = (R)cosl(ang); w[t].y = (R)sinl(ang);
    }
    W.alloc(L); W.upload(w.data(), L, ctx->stream);
    dev.W = W.p;
    perm.resize(L);
    for (int pos = 0; pos < L; pos++) {
      int rem = pos, n = L, k = 0, mult = 1;
      for (int ps = 0; ps < dev.npass; ps++) {
        const int r = dev.radix[ps], s = n / r, m = rem / s;
        rem -= m * s; k += m * mult; mult *= r; n = s;
      }
      perm[pos] = k;
    }
    perm_d.alloc(L); perm_d.upload(perm.data(), L, ctx->stream);
  }
  size_t lds_bytes() const { return (size_t)L * sizeof(T2); }
};

// host DFT in double for any length (the spectrum of the zero-padded taps, one-off at create): mixed radix by the
// smallest prime factor, a direct sum for prime lengths
inline void host_dft(std::vector< std::complex<double> > &a, int sign) {
  const size_t n = a.size();
  if (n <= 1) return;
  size_t r = 0;
  for (size_t q = 2; q * q <= n && !r; q++) if (n % q == 0) r = q;
  const long double PI2 = 2.0L * 3.14159265358979323846264338327950288L;
  if (!r && n > 64) {   // a large prime: Bluestein's chirp transform over radix-2 transforms of M >= 2n - 1 points, all in double
    size_t M = 1; while (M < 2 * n - 1) M <<= 1;
    auto fft2 = [](std::vector< std::complex<long double> > &v, int sg) {
      const size_t L = v.size();
      for (size_t i = 1, j = 0; i < L; i++) { size_t bit = L >> 1; for (; j & bit; bit >>= 1) j ^= bit; j ^= bit; if (i < j) std::swap(v[i], v[j]); }
      for (size_t len = 2; len <= L; len <<= 1)
        for (size_t k = 0; k < len / 2; k++) {
          const long double ang = sg * 2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)len;
          const std::complex<long double> wk(cosl(ang), sinl(ang));
          for (size_t q = 0; q < L; q += len) { const std::complex<long double> u = v[q + k], t = wk * v[q + k + len / 2]; v[q + k] = u + t; v[q + k + len / 2] = u - t; }
        }
    };
    std::vector< std::complex<long double> > w(n), A(M, std::complex<long double>(0, 0)), B(M, std::complex<long double>(0, 0));
    for (size_t j = 0; j < n; j++) {
      const long double ang = sign * 3.14159265358979323846264338327950288L * (long double)((j * j) % (2 * n)) / (long double)n;   // (j^2 mod 2n: exact phase)
      w[j] = std::complex<long double>(cosl(ang), sinl(ang));
      A[j] = std::complex<long double>(a[j].real(), a[j].imag()) * w[j];
      B[j] = std::conj(w[j]);
      if (j) B[M - j] = std::conj(w[j]);
    }
    fft2(A, -1); fft2(B, -1);
    for (size_t k = 0; k < M; k++) A[k] *= B[k];
    fft2(A, +1);
    for (size_t k = 0; k < n; k++) { const std::complex<long double> v = A[k] / (long double)M * w[k]; a[k] = std::complex<double>((double)v.real(), (double)v.imag()); }
    return;
  }
  if (!r) {   // a small prime: direct
    std::vector< std::complex<double> > o(n);
    for (size_t m = 0; m < n; m++) {
      std::complex<long double> acc(0, 0);
      for (size_t k = 0; k < n; k++) {
        const long double ang = sign * PI2 * (long double)((k * m) % n) / (long double)n;
        acc += std::complex<long double>(a[k].real(), a[k].imag()) * std::complex<long double>(cosl(ang), sinl(ang));
      }
      o[m] = std::complex<double>((double)acc.real(), (double)acc.imag());
    }
    a = o;
    return;
  }
  const size_t s = n / r;   // n = r * s: r interleaved sub-sequences of length s (decimation in time)
  std::vector< std::vector< std::complex<double> > > sub(r, std::vector< std::complex<double> >(s));
  for (size_t q = 0; q < r; q++) { for (size_t i = 0; i < s; i++) sub[q][i] = a[i * r + q]; host_dft(sub[q], sign); }
  for (size_t m = 0; m < n; m++) {
    std::complex<long double> acc(0, 0);
    for (size_t q = 0; q < r; q++) {
      const long double ang = sign * PI2 * (long double)((q * m) % n) / (long double)n;
      const std::complex<double> v = sub[q][m % s];
      acc += std::complex<long double>(v.real(), v.imag()) * std::complex<long double>(cosl(ang), sinl(ang));
    }
    a[m] = std::complex<double>((double)acc.real(), (double)acc.imag());
  }
}

}  // namespace fftgen
}  // namespace sdrhip
