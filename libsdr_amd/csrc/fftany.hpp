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
  constexpr int U = 8;   // (loads in flight per lane, see fourstep_tile_kernel)
  for (int i0 = tid; i0 < L; i0 += GT * U) {
    T2 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = src[min(i0 + u * GT, L - 1)];
#pragma unroll
    for (int u = 0; u < U; u++) if (i0 + u * GT < L) { if (conj) v[u].y = -v[u].y; xl[i0 + u * GT] = v[u]; }
  }
  __syncthreads();
  fftgen::forward_dif(xl, p, tid);
  for (int i0 = tid; i0 < L; i0 += GT * U) {   // position i holds frequency perm[i]
    int pk[U];
#pragma unroll
    for (int u = 0; u < U; u++) pk[u] = perm[min(i0 + u * GT, L - 1)];
#pragma unroll
    for (int u = 0; u < U; u++) if (i0 + u * GT < L) { T2 v = xl[i0 + u * GT]; if (conj) v.y = -v.y; dst[pk[u]] = v; }
  }
}

// The general plan's passes for a TEAM of `nt` lanes (nt divides GT): a workgroup then runs GT / nt transforms side by side,
// each in its own LDS array, all teams in step (the barriers are the workgroup's). Same algebra as fftgen::pass_dif.
template <class T2, int R>
__device__ __forceinline__ void team_pass_dif(T2 *x, const fftgen::GenDev<T2> &p0, const T2 *W, int n, int lt, int nt) {
  fftgen::GenDev<T2> p = p0; p.W = W;   // (the roots from the workgroup's LDS copy)
  const int s = n / R, tw = p.L / n;
  const float inv_s = 1.0f / (float)s;   // (b < 2^14: floor((b + 0.5) / s) in float is exact — an integer division costs 30 instructions)
  for (int b = lt; b < p.L / R; b += nt) {
    const int blk = (int)(((float)b + 0.5f) * inv_s), j = b - blk * s, base = blk * n + j;
    T2 v[R];
#pragma unroll
    for (int k = 0; k < R; k++) v[k] = x[base + k * s];
    fftgen::dft_small<T2, R, -1>(v, p);
    x[base] = v[0];
#pragma unroll
    for (int m = 1; m < R; m++) x[base + m * s] = s > 1 ? gmul(v[m], p.W[j * tw * m]) : v[m];
  }
}
template <class T2>
__device__ void team_forward_dif(T2 *x, const fftgen::GenDev<T2> &p, const T2 *W, int lt, int nt) {
  int n = p.L;
  for (int pass = 0; pass < p.npass; pass++) {
    const int r = p.radix[pass];
#define SDRHIP_GEN_CALL(R_) team_pass_dif<T2, R_>(x, p, W, n, lt, nt)
    SDRHIP_GEN_RADIX_SWITCH(r, SDRHIP_GEN_CALL)
#undef SDRHIP_GEN_CALL
    __syncthreads();
    n /= r;
  }
}

// One pass of the four-step plan (fftgen.hpp explains the algebra), batched and TILED: a workgroup takes TC neighbouring
// transforms — pass 1 (COLS): the columns j2 = t TC ... of the n1 x n2 image, each n1 points at stride n2, so that every row
// of the tile is TC contiguous elements; results, twiddled by W_n^(j2 k1) = wa[a] wb[b] (j2 k1 = a n2 + b), go to
// A[k1][j2] — again TC contiguous elements per k1. Pass 2 (!COLS): the rows k1 = t TC ... of A (n2 contiguous points each);
// X[k1 + n1 k2] is TC contiguous elements per k2. `len` = points per transform, `cnt` = transforms per batch item.
// Backward = conj(forward(conj x)): pass 1 conjugates what it reads, pass 2 what it writes.
// What the overlap-save filter folds into the transform's passes (BigConv, fftconv.hip): the block GATHER (history | input, zero
// beyond the call) into forward pass 1's loads, the spectrum PRODUCT (all bands) into forward pass 2's stores, and the SCATTER
// of a block's last `hop` results into inverse pass 2's stores — 7.5 instead of 13.5 array passes per block.
// Batch item z of the forward transform = block z % nblk of channel z / nblk; of the inverse: band z / (cg nblk) first.
template <class T2>
struct ConvFuse {
  const T2 *in; long in_stride; const T2 *hist; int HH, N, hop, nblk;   // FUSE_GATHER
  const T2 *Kp; int nb; long band_elems;                                // FUSE_MUL: Y[b * band_elems + z n + idx] = X[idx] Kp[b n + idx]
  T2 *out; long out_stride, out_band; int cg;                           // FUSE_SCATTER
};
enum { FUSE_NONE = 0, FUSE_GATHER = 1, FUSE_MUL = 2, FUSE_SCATTER = 3 };

template <class T2, bool COLS, int FUSE = FUSE_NONE>
__global__ __launch_bounds__(GT) void fourstep_tile_kernel(const fftgen::GenDev<T2> p, const int *perm, int conj, long n, int len, int cnt, int TC,
                                                            const T2 *in, T2 *out, const T2 *wa, const T2 *wb, int n1, int n2, const ConvFuse<T2> f, int wl_on) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T2 *xl = reinterpret_cast<T2 *>(smem_raw);
  const int tid = threadIdx.x, nt = GT / TC, team = tid / nt, lt = tid - team * nt;
  const long z = blockIdx.y;
  const int t0 = blockIdx.x * TC;   // first transform of the tile
  const T2 *src = in + z * n;
  T2 *dst = out + z * n;
  const int LP = len + 1;           // (one pad element per array: the tile's loads walk the arrays at the same index)
  T2 *wl = xl + (size_t)TC * LP;    // the plan's roots exp(-2 pi i t / len) and its output permutation, once per workgroup (where they fit beside the tile)
  int *pl_ = reinterpret_cast<int *>(wl + len);
  if (wl_on) for (int i = tid; i < len; i += GT) { wl[i] = p.W[i]; pl_[i] = perm[i]; }
  const T2 *Wr = wl_on ? wl : p.W;
  const int *permL = wl_on ? pl_ : perm;
  // load: element i of transform t0 + c sits at  COLS: i n2 + (t0 + c)   rows: (t0 + c) n2 + i
  const long gc = FUSE == FUSE_GATHER ? z / f.nblk : 0, gfirst = FUSE == FUSE_GATHER ? (z - gc * f.nblk) * (long)f.hop - f.HH : 0;
  const int tsh = 31 - __clz(TC);   // (TC is a power of two)
  const float inv_len = 1.0f / (float)len;
  // (U loads in flight per lane: written as "issue U loads, then U LDS stores" — a plain loop waits for every load before the
  // next one is issued, and 16 dependent trips to memory per workgroup were most of this kernel's time)
  constexpr int U = 8;
  const int total = len * TC;
  for (int e0 = tid; e0 < total; e0 += GT * U) {   // (e < 2^18: the float quotient below is exact)
    T2 v[U];
    int li[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int e = min(e0 + u * GT, total - 1);
      int i, c;
      if (COLS) { i = e >> tsh; c = e & (TC - 1); } else { c = (int)(((float)e + 0.5f) * inv_len); i = e - c * len; }
      li[u] = c * LP + i;
      const bool ok = t0 + c < cnt;
      const int cc = ok ? c : 0;   // (clamped address, value dropped below)
      if (FUSE == FUSE_GATHER) {   // (COLS) the block's sample (i n2 + col) straight from the call's input / the history
        const long rel = gfirst + (long)i * n2 + (t0 + cc);
        const long h = f.HH + rel;
        const T2 *ptr = rel >= 0 ? f.in + gc * f.in_stride + (rel < f.N ? rel : (long)f.N - 1) : f.hist + gc * f.HH + (h >= 0 ? h : 0);
        v[u] = *ptr;
        if (!ok || rel >= f.N || h < 0) v[u] = mk<T2>(0, 0);
      } else {
        v[u] = COLS ? src[(long)i * n2 + (t0 + cc)] : src[(long)(t0 + cc) * n2 + i];
        if (!ok) v[u] = mk<T2>(0, 0);
      }
      if (conj && COLS) v[u].y = -v[u].y;
    }
#pragma unroll
    for (int u = 0; u < U; u++) if (e0 + u * GT < total) xl[li[u]] = v[u];
  }
  __syncthreads();
  team_forward_dif(xl + team * LP, p, Wr, lt, nt);
  // store: frequency k of transform t0 + c goes to  COLS: k n2 + (t0 + c) [twiddled]   rows: (t0 + c) + n1 k
  const float inv_n2 = 1.0f / (float)n2;
  for (int e0 = tid; e0 < total; e0 += GT * U) {   // (likewise: the permutation and twiddle lookups of U elements first, then the stores)
    T2 v[U], w1[U], w2[U];
    int kk[U], cc_[U];
    bool okk[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int e = min(e0 + u * GT, total - 1);
      const int pos = e >> tsh, c = e & (TC - 1);
      okk[u] = e0 + u * GT < total && t0 + c < cnt;
      cc_[u] = c;
      kk[u] = permL[pos];
      v[u] = xl[c * LP + pos];
      if (COLS) {
        // m = a n2 + b (m < n <= 2^27): a float estimate of the quotient, off by at most two, then exact remainder steps
        const unsigned m = okk[u] ? (unsigned)(t0 + c) * (unsigned)kk[u] : 0u;
        unsigned a_ = (unsigned)((float)m * inv_n2);
        int r_ = (int)(m - a_ * (unsigned)n2);
        if (r_ < 0) { a_--; r_ += n2; }
        if (r_ < 0) { a_--; r_ += n2; }
        if (r_ >= n2) { a_++; r_ -= n2; }
        if (r_ >= n2) { a_++; r_ -= n2; }
        w1[u] = wa[a_]; w2[u] = wb[r_];
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (!okk[u]) continue;
      const int c = cc_[u], k = kk[u];
      T2 val = v[u];
      if (COLS) {
        val = gmul(val, gmul(w1[u], w2[u]));
        dst[(long)k * n2 + (t0 + c)] = val;
      } else {
        if (conj) val.y = -val.y;
        const long idx = (long)(t0 + c) + (long)n1 * k;   // natural order
        if (FUSE == FUSE_MUL) {
          for (int b = 0; b < f.nb; b++) out[(long)b * f.band_elems + z * n + idx] = gmul(val, f.Kp[(long)b * n + idx]);
        } else if (FUSE == FUSE_SCATTER) {
          if (idx >= f.HH) {
            const long per = (long)f.cg * f.nblk, band = z / per, r = z - band * per, ch = r / f.nblk, blk = r - ch * f.nblk;
            const long o = blk * (long)f.hop + (idx - f.HH);
            if (o < f.N) f.out[band * f.out_band + ch * f.out_stride + o] = val;
          }
        } else {
          dst[idx] = val;
        }
      }
    }
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
  constexpr int U = 8;   // (loads in flight per lane, see fourstep_tile_kernel)
  for (int i0 = tid; i0 < M; i0 += GT * U) {
    T2 v[U], wv[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const int i = min(i0 + u * GT, n - 1); v[u] = src[i]; wv[u] = w[i]; }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = i0 + u * GT;
      if (i < M) { T2 t = v[u]; if (conj) t.y = -t.y; xl[i] = i < n ? gmul(t, wv[u]) : mk<T2>(0, 0); }
    }
  }
  __syncthreads();
  fftgen::forward_dif(xl, p, tid);
  for (int i0 = tid; i0 < M; i0 += GT * U) {   // (the spectrum: stored in the forward transform's output order, pre-scaled by 1 / M)
    T2 bq[U];
#pragma unroll
    for (int u = 0; u < U; u++) bq[u] = bspec[min(i0 + u * GT, M - 1)];
#pragma unroll
    for (int u = 0; u < U; u++) if (i0 + u * GT < M) xl[i0 + u * GT] = gmul(xl[i0 + u * GT], bq[u]);
  }
  __syncthreads();
  fftgen::inverse_dit(xl, p, tid);
  for (int i0 = tid; i0 < n; i0 += GT * U) {
    T2 wv[U];
#pragma unroll
    for (int u = 0; u < U; u++) wv[u] = w[min(i0 + u * GT, n - 1)];
#pragma unroll
    for (int u = 0; u < U; u++) if (i0 + u * GT < n) { T2 v = gmul(xl[i0 + u * GT], wv[u]); if (conj) v.y = -v.y; dst[i0 + u * GT] = v; }
  }
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

// (a kernel's dynamic-LDS limit: allow_lds_max, sdrhip_internal.hpp — once per kernel and device, to the hardware's maximum)
template <class K>
inline void allow_lds(K kernel, size_t bytes) { allow_lds_max(kernel, bytes); }

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

  // n = n1 x n2 with both parts plannable in LDS, as BALANCED as the factors allow (two passes of sqrt(n)-point transforms:
  // a lopsided split would run one pass as tens of thousands of 2-point transforms); 0: none
  static long split(long n) {
    const long maxL = max_lds_points();
    std::vector<int> rx;
    if (!fftgen::GenPlan<T2>::factor_long(n)) return 0;
    long best = 0;
    for (long d = 2; d * d <= n; d++) {   // d <= n / d: the larger part is n / d
      if (n % d || n / d > maxL) continue;
      if (fftgen::GenPlan<T2>::factor((int)d, rx, nullptr) && fftgen::GenPlan<T2>::factor((int)(n / d), rx, nullptr)) best = d;
    }
    return best ? n / best : 0;   // n1 = the larger part (columns of n1 points in pass 1), n2 = n / n1 <= n1
  }
  int tc1 = 1, tc2 = 1;   // transforms per workgroup in pass 1 / pass 2
  // a pass's LDS: the tile's arrays, and the roots beside them where that still fits a workgroup
  static bool roots_fit(int tc, long len) { return ((size_t)tc * (len + 1) + len) * sizeof(T2) + (size_t)len * 4 <= 150 * 1024; }
  static size_t pass_lds(int tc, long len) { return ((size_t)tc * (len + 1) + (roots_fit(tc, len) ? len : 0)) * sizeof(T2) + (roots_fit(tc, len) ? (size_t)len * 4 : 0); }
  static int tile_count(long len) {   // a power of two <= 16, <= GT / 16 lanes per team, arrays within 96 KB
    int tc = 16;
    while (tc > 1 && ((size_t)tc * (len + 1) + len) * sizeof(T2) > 96 * 1024) tc >>= 1;   // (tc = 1, len 16384: 128 KB, the roots stay in global memory)
    return tc;
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
      tc1 = tile_count(n1); tc2 = tile_count(n2);
      allow_lds((fourstep_tile_kernel<T2, true>), pass_lds(tc1, n1));
      allow_lds((fourstep_tile_kernel<T2, false>), pass_lds(tc2, n2));
      allow_lds((fourstep_tile_kernel<T2, true, FUSE_GATHER>), pass_lds(tc1, n1));
      allow_lds((fourstep_tile_kernel<T2, false, FUSE_MUL>), pass_lds(tc2, n2));
      allow_lds((fourstep_tile_kernel<T2, false, FUSE_SCATTER>), pass_lds(tc2, n2));
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

  // (FOURSTEP plans) the filter's two transforms with gather / product / scatter folded in (ConvFuse). conv_forward: `batch`
  // blocks gathered from f.in / f.hist, transformed, times the f.nb spectra -> Y (f.nb x batch x n, band-major);
  // conv_inverse: the f.nb x batch spectra in Y transformed back, each block's last hop results scattered to f.out.
  // The plan's scratch must hold f.nb x batch transforms (reserve).
  bool fuses() const { return kind == FOURSTEP; }
  void conv_forward(const ConvFuse<T2> &f, long batch, T2 *Y) {
    ctx->use();
    reserve(batch);
    hipStream_t st = ctx->stream;
    for (long z0 = 0; z0 < batch; z0 += 32768) {   // (chunk starts are multiples of 32768 blocks; the gather needs z = absolute block number)
      const long zb = std::min<long>(32768, batch - z0);
      SDRHIP_REQUIRE(z0 == 0, SDRHIP_E_UNSUPPORTED, "more than 32768 blocks in one pass");
      hipLaunchKernelGGL((fourstep_tile_kernel<T2, true, FUSE_GATHER>), dim3((unsigned)((n2 + tc1 - 1) / tc1), (unsigned)zb), dim3(GT), pass_lds(tc1, n1), st,
                         p1.dev, p1.perm_d.p, 0, n, (int)n1, (int)n2, tc1, (const T2 *)nullptr, tmp.p, wa.p, wb.p, (int)n1, (int)n2, f, (int)roots_fit(tc1, n1));
      hipLaunchKernelGGL((fourstep_tile_kernel<T2, false, FUSE_MUL>), dim3((unsigned)((n1 + tc2 - 1) / tc2), (unsigned)zb), dim3(GT), pass_lds(tc2, n2), st,
                         p2.dev, p2.perm_d.p, 0, n, (int)n2, (int)n1, tc2, tmp.p, Y, wa.p, wb.p, (int)n1, (int)n2, f, (int)roots_fit(tc2, n2));
    }
    SDRHIP_CHECK_HIP(hipGetLastError());
  }
  void conv_inverse(const ConvFuse<T2> &f, long batch_all, T2 *Y) {
    ctx->use();
    reserve(batch_all);
    hipStream_t st = ctx->stream;
    SDRHIP_REQUIRE(batch_all <= 32768, SDRHIP_E_UNSUPPORTED, "more than 32768 blocks in one pass");
    hipLaunchKernelGGL((fourstep_tile_kernel<T2, true>), dim3((unsigned)((n2 + tc1 - 1) / tc1), (unsigned)batch_all), dim3(GT), pass_lds(tc1, n1), st,
                       p1.dev, p1.perm_d.p, 1, n, (int)n1, (int)n2, tc1, Y, tmp.p, wa.p, wb.p, (int)n1, (int)n2, ConvFuse<T2>{}, (int)roots_fit(tc1, n1));
    hipLaunchKernelGGL((fourstep_tile_kernel<T2, false, FUSE_SCATTER>), dim3((unsigned)((n1 + tc2 - 1) / tc2), (unsigned)batch_all), dim3(GT), pass_lds(tc2, n2), st,
                       p2.dev, p2.perm_d.p, 1, n, (int)n2, (int)n1, tc2, tmp.p, (T2 *)nullptr, wa.p, wb.p, (int)n1, (int)n2, f, (int)roots_fit(tc2, n2));
    SDRHIP_CHECK_HIP(hipGetLastError());
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
          hipLaunchKernelGGL((fourstep_tile_kernel<T2, true>), dim3((unsigned)((n2 + tc1 - 1) / tc1), (unsigned)zb), dim3(GT), pass_lds(tc1, n1), st,
                             p1.dev, p1.perm_d.p, conj, n, (int)n1, (int)n2, tc1, in + z0 * n, tmp.p + z0 * n, wa.p, wb.p, (int)n1, (int)n2, ConvFuse<T2>{}, (int)roots_fit(tc1, n1));
          hipLaunchKernelGGL((fourstep_tile_kernel<T2, false>), dim3((unsigned)((n1 + tc2 - 1) / tc2), (unsigned)zb), dim3(GT), pass_lds(tc2, n2), st,
                             p2.dev, p2.perm_d.p, conj, n, (int)n2, (int)n1, tc2, tmp.p + z0 * n, out + z0 * n, wa.p, wb.p, (int)n1, (int)n2, ConvFuse<T2>{}, (int)roots_fit(tc2, n2));
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
