// fir.hip — K2: FIRFilter<complex<int16_t>> (bit-exact) and K3: FIRFilter<complex<float>>
// (tolerance path, optional SubSample<complex<float>> folded in) with demodulator epilogues.
//
// Replaces (reference, file:line):
//   FIRFilter<Scalar,Coeffs>::_process          src/firfilter.hh:231-247
//   operator*(double, complex<int16>/<float>)    src/operators.hh:16-18,24-26
//   SubSample<complex<float>>::_process          src/subsample.hh:92-101   (K3, decim > 1)
//   FMDemod / AMDemod / USBDemod                 src/demod.hh:242-254, :73-76, :156-161
//
// K2 semantics are ORDER DEPENDENT: `out += alpha[j]*x` converts back to int16
// after every tap, so per component   acc <- trunc_toward_zero( fl64(acc + fl64(alpha[j]*x)) )
// for j = 0..order-1 in that order. One lane owns R consecutive outputs and walks the taps
// sequentially in fp64 with contraction off (v_mul_f64, v_add_f64, v_trunc_f64); parallelism is
// across outputs and channels only. acc stays an integer-valued double; the int16 wrap of the
// reference can only trigger when sum|alpha| > 1, which selects the slower WRAP variant.
//
// K3 may reassociate (<= 1e-5 rel): taps are convolved with the 1/D box on the host in double
// (beta = alpha (*) box_D / D) and the filter is evaluated only at the decimated instants.
#include "fm_phi.hpp"
#include "sdrhip_internal.hpp"
#include <cstdlib>

using namespace sdrhip;

namespace {

constexpr int TPB = 256;

// ---------------------------------------------------------------------------------------------
// shared epilogue helpers (int16)
// ---------------------------------------------------------------------------------------------
// trunc(num/den) for |num| <= 4096*den, 0 < den < 2^16 (the only divisions fast_atan2 makes): float
// estimate (|q| <= 4096, error < 1) + one exact remainder correction, instead of the generic 32-bit sequence
__device__ __forceinline__ int div_small(int num, int den) {
  const unsigned nu = (unsigned)(num < 0 ? -num : num), de = (unsigned)den;
  unsigned q = (unsigned)((float)nu * __builtin_amdgcn_rcpf((float)de));   // v_rcp_f32: 1 ulp, |q| <= 4096
  int r = (int)(nu - __umul24(q, de));
  if (r < 0) { q -= 1; r += (int)de; }
  if (r >= (int)de) q += 1;
  return num < 0 ? -(int)q : (int)q;
}
__device__ __forceinline__ short am_i16(int re, int im) {
  const int m = (int)((unsigned)(re * re) + (unsigned)(im * im));
  return (short)(int)sqrt((double)m);
}
__device__ __forceinline__ short usb_i16(int re, int im) { return (short)((re + im) / 2); }

// =============================================================================================
// K2: exact complex<int16> FIR
// =============================================================================================
// R2 consecutive outputs per lane: 8 for batches that fill the chip (the circular window's conversions amortise over twice
// the taps: +3 % over R2 = 4), 4 for small ones (twice the workgroups, half the serial work per lane: latency)
constexpr int R2MAX = 8;
constexpr int U2 = 4;              // taps per unrolled chunk (the order is zero-padded at the front to whole window periods R2 + U2)

struct Fir16Args {
  const uint32_t *in; long in_stride;
  const uint32_t *hist_old; uint32_t *hist_new; int HH;   // HH = OP-1 samples before the call
  const short *fm_old; short *fm_new;
  const double *alpha;   // OP taps, zero padded at the front
  int OP, N, ovl, OT;    // OT = outputs emitted per tile = T2 - ovl
  void *out; long out_stride; int epilogue;
};

__device__ __forceinline__ uint32_t load_x16(const Fir16Args &a, int c, int rel) {
  if (rel >= 0) return rel < a.N ? a.in[(long)c * a.in_stride + rel] : 0u;
  const int h = a.HH + rel;
  return h >= 0 ? a.hist_old[(long)c * a.HH + h] : 0u;
}

template <bool WRAP>
__device__ __forceinline__ double tap_step(double acc, double alpha, double x) {
  // fl64(alpha*x), fl64(acc + .), trunc toward zero — no FMA (x86-64 reference has none)
  double t = __builtin_trunc(__dadd_rn(acc, __dmul_rn(alpha, x)));
  if (WRAP) t = (double)(short)(int)t;   // int16 <- int32 <- double, as gcc/x86-64 converts
  return t;
}

template <bool WRAP, int R2>
__global__ __launch_bounds__(TPB) void fir_cs16_exact_kernel(const Fir16Args a) {
  constexpr int T2 = TPB * R2;       // outputs computed per tile
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int XS = T2 + a.OP + 8;
  uint32_t *xs = smem;            // xs[i] = x[tb-(OP-1)+i]
  uint32_t *ybuf = smem + XS;     // T2 packed cs16 results

  int c, tile;
  xcd_unit_order(tile, c);   // (neighbouring tiles of a channel share OP - 1 input samples: same XCD, same L2)
  const int tid = threadIdx.x;
  const int tb = tile * a.OT - a.ovl;            // index of the tile's first computed output
  const int outs_here = min(T2, a.N - tb);

  {
    const int first = tb - (a.OP - 1);
    const int need = min(XS, outs_here + a.OP + 8);
    for (int i = tid; i < need; i += TPB) xs[i] = load_x16(a, c, first + i);
  }
  __syncthreads();

  if (R2 * tid < outs_here) {
    double are[R2], aim[R2];
#pragma unroll
    for (int r = 0; r < R2; r++) { are[r] = 0.0; aim[r] = 0.0; }
    // CIRCULAR window of converted samples: slot (m mod W) holds x[R2*tid + m]; a chunk of U2 taps starting at tap i0
    // needs m = i0 .. i0 + R2 + U2 - 2 and loads m = i0 + R2 .. i0 + R2 + U2 - 1 while the slots of m < i0 are dead.
    // With W = R2 + U2 and the tap loop unrolled over W / U2 chunks every slot index is a compile-time constant: no
    // register moves (the sliding window's 2 * R2 moves per chunk were a sixth of the loop's vector instructions).
    constexpr int W = R2 + U2;
    static_assert(W % U2 == 0 && R2 % U2 == 0, "the circular window's period is a whole number of chunks");
    double wre[W], wim[W];
    const uint32_t *px = xs + R2 * tid;
#pragma unroll
    for (int k = 0; k < R2; k++) {
      const uint32_t v = px[k];
      wre[k] = (double)(short)(v & 0xffffu); wim[k] = (double)(short)(v >> 16);
    }
    const double *__restrict__ al = a.alpha;
    for (int i0 = 0; i0 < a.OP; i0 += W) {   // (OP is a multiple of W: the order is zero-padded at the front)
#pragma unroll
      for (int ch = 0; ch < W / U2; ch++) {   // chunk ch of this period: taps i0 + ch * U2 ..
        const int ib = i0 + ch * U2;
#pragma unroll
        for (int k = 0; k < U2; k++) {
          const uint32_t v = px[ib + R2 + k];
          wre[(ch * U2 + R2 + k) % W] = (double)(short)(v & 0xffffu); wim[(ch * U2 + R2 + k) % W] = (double)(short)(v >> 16);
        }
#pragma unroll
        for (int u = 0; u < U2; u++) {
          const double al_j = al[ib + u];     // wave-uniform -> scalar load
#pragma unroll
          for (int r = 0; r < R2; r++) {
            are[r] = tap_step<WRAP>(are[r], al_j, wre[(ch * U2 + u + r) % W]);
            aim[r] = tap_step<WRAP>(aim[r], al_j, wim[(ch * U2 + u + r) % W]);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R2; r++) {
      const int yr = (int)are[r], yi = (int)aim[r];
      ybuf[R2 * tid + r] = ((uint32_t)(uint16_t)yr) | ((uint32_t)(uint16_t)yi << 16);
    }
  }
  __syncthreads();

  for (int l = a.ovl + tid; l < outs_here; l += TPB) {
    const int j = tb + l;
    if (j < 0 || j >= a.N) continue;
    const uint32_t y = ybuf[l];
    const int yr = (short)(y & 0xffffu), yi = (short)(y >> 16);
    if (a.epilogue == SDRHIP_EPI_NONE) {
      reinterpret_cast<uint32_t *>(a.out)[(long)c * a.out_stride + j] = y;
    } else {
      short o;
      if (a.epilogue == SDRHIP_EPI_AM) o = am_i16(yr, yi);
      else if (a.epilogue == SDRHIP_EPI_USB) o = usb_i16(yr, yi);
      else {
        const int phi = fm_phi(yr, yi);
        if (j == 0) o = (short)yr;
        else {
          int prev;
          if (j == 1) prev = a.fm_old[c];
          else { const uint32_t yp = ybuf[l - 1]; prev = fm_phi((short)(yp & 0xffffu), (short)(yp >> 16)); }
          o = (short)(prev - phi);
        }
        if (j == a.N - 1 && a.N >= 2) a.fm_new[c] = (short)phi;
      }
      reinterpret_cast<short *>(a.out)[(long)c * a.out_stride + j] = o;
    }
  }

  if (tile == (int)gridDim.x - 1) {
    for (int k = tid; k < a.HH; k += TPB) {
      const long qq = (long)a.N + k;
      a.hist_new[(long)c * a.HH + k] =
          qq < a.HH ? a.hist_old[(long)c * a.HH + qq] : a.in[(long)c * a.in_stride + (qq - a.HH)];
    }
  }
}

// =============================================================================================
// K3: complex<float> FIR, decimation folded (outputs at absolute input indices g*D + D-1)
// =============================================================================================
constexpr int T3 = TPB;   // one output per lane and tile pass

struct Fir32Args {
  const float2 *in; long in_stride;
  const float2 *hist_old; float2 *hist_new; int HH;   // HH = M-1
  const float *beta;     // M folded taps (float), beta[M-1] meets the newest sample
  int M, D, N;
  int first_rel;   // call-relative index of the newest sample of the first output of this call
  int n_out;
  void *out; long out_stride; int epilogue;
  // register-tiled kernel
  const float *betap;      // beta with (R-1)*D zeros on either side
  unsigned rd_magic;       // ceil(2^32 / (R*D)): i / (R*D) = umulhi(i, rd_magic) for the tile's i
  // fused frequency shift (float baseband, fbb_f32.hip): x[n] * exp(-2 pi i fc n / fs), n = absolute sample index
  // The phase of sample i of a tile (absolute index n = n_tile + i, i = t + TPB*k) splits into three exact factors:
  // exp(-iw n) = P[tile] * E[t] * W[k]. E (per lane) and W (lane-uniform) are tables made once from float64 values,
  // P[tile] = exp(-iw n_tile) is evaluated in float64 per call by a tiny kernel and multiplies the OUTPUT (the filter is
  // linear), so staging a sample costs 8 float32 operations and no lane evaluates sincos or carries a float64 phasor.
  int shift_on;
  const float2 *etab, *wtab, *ptab;
  const float2 *etab2;              // E2[t] = the phasor of sample 2t (staging two samples per lane and load)
  float2 parg[32]; int p_in_args;   // calls of up to 32 tiles carry their tile factors in the arguments (host float64): no table kernel
  int roll;                         // the call's last tile also rolls the FIR history forward (hist_new <- concat(hist_old, in))
  int tiles, tpw;                   // pipelined kernel: tiles of the call, consecutive tiles per workgroup
};

__device__ __forceinline__ float2 load_x32(const Fir32Args &a, int c, int rel) {
  if (rel >= 0) return rel < a.N ? a.in[(long)c * a.in_stride + rel] : make_float2(0.f, 0.f);
  const int h = a.HH + rel;
  return h >= 0 ? a.hist_old[(long)c * a.HH + h] : make_float2(0.f, 0.f);
}

__global__ __launch_bounds__(TPB) void fir_cf32_kernel(const Fir32Args a) {
  extern __shared__ __attribute__((aligned(16))) float2 smemf[];
  float2 *xs = smemf;   // xs[i] = x[base-(M-1)+i], base = newest sample of the tile's first output
  const int c = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
  const int j0 = tile * T3;
  const int outs_here = min(T3, a.n_out - j0);
  const int base = a.first_rel + j0 * a.D;
  if (outs_here > 0) {
    const int need = (outs_here - 1) * a.D + a.M;
    for (int i = tid; i < need; i += TPB) xs[i] = load_x32(a, c, base - (a.M - 1) + i);
  }
  __syncthreads();
  if (tid < outs_here) {
    const float2 *px = xs + tid * a.D;
    const float *__restrict__ b = a.beta;
    float sr0 = 0.f, si0 = 0.f, sr1 = 0.f, si1 = 0.f;
    int m = 0;
    for (; m + 1 < a.M; m += 2) {
      const float2 x0 = px[m], x1 = px[m + 1];
      const float b0 = b[m], b1 = b[m + 1];
      sr0 = __builtin_fmaf(b0, x0.x, sr0); si0 = __builtin_fmaf(b0, x0.y, si0);
      sr1 = __builtin_fmaf(b1, x1.x, sr1); si1 = __builtin_fmaf(b1, x1.y, si1);
    }
    if (m < a.M) { const float2 x0 = px[m]; const float b0 = b[m]; sr0 = __builtin_fmaf(b0, x0.x, sr0); si0 = __builtin_fmaf(b0, x0.y, si0); }
    const float yr = sr0 + sr1, yi = si0 + si1;
    const int j = j0 + tid;
    if (a.epilogue == SDRHIP_EPI_NONE) reinterpret_cast<float2 *>(a.out)[(long)c * a.out_stride + j] = make_float2(yr, yi);
    else if (a.epilogue == SDRHIP_EPI_AM) reinterpret_cast<float *>(a.out)[(long)c * a.out_stride + j] = sqrtf(yr * yr + yi * yi);
    else reinterpret_cast<float *>(a.out)[(long)c * a.out_stride + j] = (yr + yi) / 2;
  }
}

// Register-tiled form: a lane owns R consecutive outputs (R*D input samples apart from its neighbour's) and walks
// its window once — every sample it reads from LDS feeds R accumulator pairs (the taps are lane-uniform: scalar
// loads from the zero-padded copy of beta), 2R FMAs per 8-byte LDS read instead of 2. One pad element per R*D
// samples makes the lane stride R*D+1 elements, i.e. an odd number of 8-byte bank pairs: conflict-free.
// The float baseband's frequency shift (fbb_f32.hip) is applied while staging: each lane evaluates the phasor of
// its first sample in float64 (the oracle's closed form) and advances it by a constant float64 rotation.
// DC > 0: the decimation is the compile-time constant DC — the R tap streams of a chunk of U steps are then ONE
// window of U + (R-1)*DC consecutive taps with compile-time indices (wide scalar loads, no per-step tap loads)
template <int R, int DC>
__global__ __launch_bounds__(TPB) void fir_cf32_rt_kernel(const Fir32Args a) {
  extern __shared__ __attribute__((aligned(16))) float2 smemf[];
  float2 *xs = smemf;
  int c, tile;
  xcd_unit_order(tile, c);   // (neighbouring tiles of a channel share M - 1 input samples: same XCD, same L2)
  const int tid = threadIdx.x;
  const int D = DC > 0 ? DC : a.D;
  const int RD = R * D;
  // pad elements per R*D samples: 1 keeps the lanes' stride an odd number of 8-byte bank pairs; the compile-time-D path takes
  // 2 — the stride (R*D + 2) * 8 bytes is then a multiple of 16, so a lane reads its window by ds_read_b128 (two samples
  // per LDS instruction), and 36-dword strides still spread the 16 lanes of a b128 group over all banks
  constexpr int PD = DC > 0 ? 2 : 1;
  const int j0 = tile * (TPB * R);
  const int outs_here = min(TPB * R, a.n_out - j0);
  const int w0 = a.first_rel + j0 * D - (a.M - 1);   // call-relative index of xs[0]
  const int need = (outs_here - 1) * D + a.M;
  // staging, 16 elements per lane in flight: an interior tile (no history, no end of call) issues 16 unconditional
  // coalesced loads before the first wait; per-element bounds logic would make every load a branch with its own wait
  const bool interior = (w0 >= 0) && (w0 + need <= a.N);
  const float2 *src = a.in + (long)c * a.in_stride + w0;
  float2 E = make_float2(1.f, 0.f);
  if (a.shift_on) E = a.etab[tid];
  // (lane-uniform row phasors through the constant address space: scalar loads, see the pipelined kernel)
  typedef float v2f __attribute__((ext_vector_type(2)));
  const __attribute__((address_space(4))) v2f *wtab_c = reinterpret_cast<const __attribute__((address_space(4))) v2f *>((uintptr_t)a.wtab);
  // staging by rows of TPB consecutive samples. The first FR rows are complete in every full tile whatever M is, so an
  // interior full tile (no history, no end of call) issues FR unconditional coalesced loads before its first wait;
  // the remaining rows (and every row of a border tile) go four at a time through clamped, branch-free loads that
  // are masked afterwards — per-element bounds branches would make every load its own round trip.
  auto put = [&](float2 v, int row, int i) {
    if (a.shift_on) {
      const v2f W_ = wtab_c[row];   // lane-uniform
      const float2 W = make_float2(W_.x, W_.y);
      const float zr = E.x * W.x - E.y * W.y, zi = E.x * W.y + E.y * W.x;
      v = make_float2(v.x * zr - v.y * zi, v.x * zi + v.y * zr);
    }
    const int q = DC > 0 ? i / (R * (DC > 0 ? DC : 1)) : (int)__umulhi((unsigned)i, a.rd_magic);
    xs[i + PD * q] = v;
  };
  constexpr int FR = DC > 0 ? ((TPB * R - 1) * (DC > 0 ? DC : 1) + 1) / TPB : 0;
  int row0 = 0;
  if (FR > 1 && PD == 2 && interior && outs_here == TPB * R && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
    // 16-byte aligned interior full tile: two samples per lane and load (dwordx4), two per LDS store (b128) — half the
    // vector-memory and LDS instructions of the staging pass. Sample 2t + 512 kk carries E2[t] * W[2 kk], its neighbour
    // that times the one-sample rotation.
    float4 v[FR / 2 > 0 ? FR / 2 : 1];
#pragma unroll
    for (int kk = 0; kk < FR / 2; kk++) v[kk] = *reinterpret_cast<const float4 *>(src + 2 * tid + 2 * TPB * kk);
    float2 E2 = make_float2(1.f, 0.f), w1 = make_float2(1.f, 0.f);
    if (a.shift_on) { E2 = a.etab2[tid]; w1 = a.etab[1]; }
#pragma unroll
    for (int kk = 0; kk < FR / 2; kk++) {
      float4 o = v[kk];
      if (a.shift_on) {
        const v2f W_ = wtab_c[2 * kk];   // lane-uniform
        const float2 W = make_float2(W_.x, W_.y);
        const float z0r = E2.x * W.x - E2.y * W.y, z0i = E2.x * W.y + E2.y * W.x;
        const float z1r = z0r * w1.x - z0i * w1.y, z1i = z0r * w1.y + z0i * w1.x;
        o = make_float4(v[kk].x * z0r - v[kk].y * z0i, v[kk].x * z0i + v[kk].y * z0r, v[kk].z * z1r - v[kk].w * z1i, v[kk].z * z1i + v[kk].w * z1r);
      }
      const int i = 2 * tid + 2 * TPB * kk;
      *reinterpret_cast<float4 *>(xs + i + PD * (i / (R * (DC > 0 ? DC : 1)))) = o;
    }
    row0 = 2 * (FR / 2);   // (an odd last full row goes through the clamped loads below with the partial ones)
  } else if (FR > 0 && interior && outs_here == TPB * R) {
    float2 v[FR > 0 ? FR : 1];
#pragma unroll
    for (int k = 0; k < FR; k++) v[k] = src[tid + k * TPB];
#pragma unroll
    for (int k = 0; k < FR; k++) put(v[k], k, tid + k * TPB);
    row0 = FR;
  }
  {
    const float2 *hrow = a.hist_old + (long)c * a.HH, *irow = a.in + (long)c * a.in_stride;
    for (int row = row0; row * TPB < need; row += 4) {
      float2 v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int rel = w0 + tid + (row + k) * TPB, hh = a.HH + rel;
        const float2 *p = rel >= 0 ? irow + min(rel, a.N - 1) : hrow + max(hh, 0);
        v[k] = *p;
        if (rel >= a.N || hh < 0) v[k] = make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int i = tid + (row + k) * TPB;
        if (i < need) put(v[k], row + k, i);
      }
    }
  }
  // a lane whose last outputs fall beyond the call still walks its whole window against zero taps: what it reads
  // behind the staged samples must be finite (0 * NaN would poison its valid accumulators)
  for (int i = need + tid; i < need + (R - 1) * D + 16; i += TPB) {
    const int q = DC > 0 ? i / (R * (DC > 0 ? DC : 1)) : (int)__umulhi((unsigned)i, a.rd_magic);
    xs[i + PD * q] = make_float2(0.f, 0.f);
  }
  __syncthreads();
  if (R * tid < outs_here) {
    float sr[R], si[R];
#pragma unroll
    for (int r = 0; r < R; r++) { sr[r] = 0.f; si[r] = 0.f; }
    const float2 *px = xs + tid * (RD + PD);
    const float *__restrict__ bp = a.betap + (R - 1) * D;   // bp[t] = beta[t], zero for t in [-(R-1)D, 0) and [M, M+(R-1)D)
    const int steps = a.M + (R - 1) * D;
    if (DC > 0) {
      constexpr int U = 16, PADZ = (R - 1) * (DC > 0 ? DC : 1), WN = U + PADZ;   // (R*DC is a multiple of U for the instantiated DC)
      const float2 *pu = px;
      for (int u0 = 0; u0 < steps; u0 += U) {
        float tw[WN];   // taps u0-PADZ .. u0+U-1: lane-uniform, one window for all R streams
#pragma unroll
        for (int k = 0; k < WN; k++) tw[k] = bp[u0 - PADZ + k];
        float2 x[U];
#pragma unroll
        for (int uu = 0; uu < U; uu += 2) {   // (chunks start 16-byte aligned: see PD)
          const float4 v = *reinterpret_cast<const float4 *>(pu + uu);
          x[uu] = make_float2(v.x, v.y); x[uu + 1] = make_float2(v.z, v.w);
        }
#pragma unroll
        for (int uu = 0; uu < U; uu++) {
#pragma unroll
          for (int r = 0; r < R; r++) {
            const float b = tw[uu - r * (DC > 0 ? DC : 1) + PADZ];
            sr[r] = __builtin_fmaf(b, x[uu].x, sr[r]); si[r] = __builtin_fmaf(b, x[uu].y, si[r]);
          }
        }
        pu += U;
        if (((u0 + U) % (R * (DC > 0 ? DC : 1))) == 0) pu += PD;   // the pad elements after every R*D samples
      }
    } else {
    constexpr int U = 8;   // steps per chunk
    int udiv = 0, umod = 0;
    for (int u0 = 0; u0 < steps; u0 += U) {   // (betap carries 16 zeros behind the padded taps; a sample read past the
      float tb[U][R];                          //  window meets only those)
      float2 x[U];
#pragma unroll
      for (int uu = 0; uu < U; uu++) {
#pragma unroll
        for (int r = 0; r < R; r++) tb[uu][r] = bp[u0 + uu - r * D];   // lane-uniform
        x[uu] = px[u0 + uu + udiv];
        if (++umod == RD) { umod = 0; udiv++; }
      }
#pragma unroll
      for (int uu = 0; uu < U; uu++) {
#pragma unroll
        for (int r = 0; r < R; r++) {
          sr[r] = __builtin_fmaf(tb[uu][r], x[uu].x, sr[r]); si[r] = __builtin_fmaf(tb[uu][r], x[uu].y, si[r]);
        }
      }
    }
    }
    if (a.shift_on) {   // the tile's phase factor (see Fir32Args)
      const float2 P = a.p_in_args ? a.parg[tile] : a.ptab[tile];
#pragma unroll
      for (int r = 0; r < R; r++) { const float yr = sr[r] * P.x - si[r] * P.y, yi = sr[r] * P.y + si[r] * P.x; sr[r] = yr; si[r] = yi; }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int j = j0 + R * tid + r;
      if (j < a.n_out) {
        if (a.epilogue == SDRHIP_EPI_NONE) reinterpret_cast<float2 *>(a.out)[(long)c * a.out_stride + j] = make_float2(sr[r], si[r]);
        else if (a.epilogue == SDRHIP_EPI_AM) reinterpret_cast<float *>(a.out)[(long)c * a.out_stride + j] = sqrtf(sr[r] * sr[r] + si[r] * si[r]);
        else reinterpret_cast<float *>(a.out)[(long)c * a.out_stride + j] = (sr[r] + si[r]) / 2;
      }
    }
  }
  if (a.roll && tile == (int)gridDim.x - 1) {   // (hist_old is only read, by this launch's border tiles; hist_new only written here)
    for (int k = tid; k < a.HH; k += TPB) {
      const long qq = (long)a.N + k;
      a.hist_new[(long)c * a.HH + k] = qq < a.HH ? a.hist_old[(long)c * a.HH + qq] : a.in[(long)c * a.in_stride + (qq - a.HH)];
    }
  }
}

// The same register-tiled kernel as a SOFTWARE PIPELINE over `tpw` consecutive tiles of a channel (compile-time
// decimation): the samples of tile t+1 are fetched into registers before tile t is computed out of LDS, and written to
// LDS (shift applied) after it — one workgroup overlaps its own memory phase with its own arithmetic instead of
// relying on the other workgroups of the CU being in the other phase (a tile is 34 KB: only 4 fit). Border tiles
// (history, end of the call, unaligned rows) are staged by the clamped loads of the plain kernel, without prefetch.
template <int R, int DC>
__global__ __launch_bounds__(TPB, R == 4 ? 2 : 4) void fir_cf32_pipe_kernel(const Fir32Args a) {
  extern __shared__ __attribute__((aligned(16))) float2 smemf[];
  float2 *xs = smemf;
  const int c = blockIdx.y, tid = threadIdx.x;
  constexpr int D = DC, RD = R * D, PD = 2;
  constexpr int FR = ((TPB * R - 1) * D + 1) / TPB, NV4 = FR / 2, NX = 4;   // rows fetched as float4 pairs / as single rows
  const int t_begin = blockIdx.x * a.tpw, t_end = min(t_begin + a.tpw, a.tiles);
  const float2 *irow = a.in + (long)c * a.in_stride, *hrow = a.hist_old + (long)c * a.HH;
  // The lane-uniform tables (taps, row and tile phasors) are read through the CONSTANT address space: scalar loads. Left
  // as global pointers they become vector loads as soon as the loop holds a store (the scalar cache is not coherent with
  // vector stores and nothing tells the compiler that `out` does not alias them) — and the wait for the first tap load
  // would wait for the whole prefetch in front of it (vector memory returns in order).
  typedef const __attribute__((address_space(4))) float *cfp;
  typedef float v2f __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(4))) v2f *cf2p;
  const cfp betap_c = reinterpret_cast<cfp>((uintptr_t)a.betap);
  const cf2p wtab_c = reinterpret_cast<cf2p>((uintptr_t)a.wtab), ptab_c = reinterpret_cast<cf2p>((uintptr_t)a.ptab);
  float2 E = make_float2(1.f, 0.f), E2 = make_float2(1.f, 0.f), w1 = make_float2(1.f, 0.f);
  if (a.shift_on) { E = a.etab[tid]; E2 = a.etab2[tid]; w1 = a.etab[1]; }
  auto put = [&](float2 v, int row, int i) {
    if (a.shift_on) {
      const v2f W_ = wtab_c[row];   // lane-uniform
      const float2 W = make_float2(W_.x, W_.y);
      const float zr = E.x * W.x - E.y * W.y, zi = E.x * W.y + E.y * W.x;
      v = make_float2(v.x * zr - v.y * zi, v.x * zi + v.y * zr);
    }
    xs[i + PD * (i / RD)] = v;
  };
  auto w0_of = [&](int t) { return a.first_rel + t * (TPB * R) * D - (a.M - 1); };   // call-relative index of xs[0]
  auto outs_of = [&](int t) { return min(TPB * R, a.n_out - t * (TPB * R)); };
  auto fast = [&](int t) {   // a full interior tile with 16-byte aligned rows, short enough for the prefetch registers
    const int w0 = w0_of(t), need = (TPB * R - 1) * D + a.M;
    return outs_of(t) == TPB * R && w0 >= 0 && w0 + need <= a.N && need <= (2 * NV4 + NX) * TPB &&
           ((reinterpret_cast<uintptr_t>(irow + w0) & 15) == 0);
  };
  float4 v4[NV4];
  float2 vx[NX];
  auto fetch = [&](int t) {
    const float2 *src = irow + w0_of(t);
    const int need = (TPB * R - 1) * D + a.M;
#pragma unroll
    for (int kk = 0; kk < NV4; kk++) v4[kk] = *reinterpret_cast<const float4 *>(src + 2 * tid + 2 * TPB * kk);
#pragma unroll
    for (int k = 0; k < NX; k++) vx[k] = src[min(tid + (2 * NV4 + k) * TPB, need - 1)];
  };
  auto zero_tail = [&](int need) {   // what a lane reads behind the staged samples must be finite (see the plain kernel)
    for (int i = need + tid; i < need + (R - 1) * D + 16; i += TPB) xs[i + PD * (i / RD)] = make_float2(0.f, 0.f);
  };
  auto commit = [&](int t) {
    const int need = (TPB * R - 1) * D + a.M;
#pragma unroll
    for (int kk = 0; kk < NV4; kk++) {
      float4 o = v4[kk];
      if (a.shift_on) {
        const v2f W_ = wtab_c[2 * kk];   // lane-uniform
        const float2 W = make_float2(W_.x, W_.y);
        const float z0r = E2.x * W.x - E2.y * W.y, z0i = E2.x * W.y + E2.y * W.x;
        const float z1r = z0r * w1.x - z0i * w1.y, z1i = z0r * w1.y + z0i * w1.x;
        o = make_float4(v4[kk].x * z0r - v4[kk].y * z0i, v4[kk].x * z0i + v4[kk].y * z0r, v4[kk].z * z1r - v4[kk].w * z1i, v4[kk].z * z1i + v4[kk].w * z1r);
      }
      const int i = 2 * tid + 2 * TPB * kk;
      *reinterpret_cast<float4 *>(xs + i + PD * (i / RD)) = o;
    }
#pragma unroll
    for (int k = 0; k < NX; k++) {
      const int i = tid + (2 * NV4 + k) * TPB;
      if (i < need) put(vx[k], 2 * NV4 + k, i);
    }
    zero_tail(need);
  };
  auto stage_plain = [&](int t) {
    const int w0 = w0_of(t), outs_here = outs_of(t), need = (outs_here - 1) * D + a.M;
    for (int row = 0; row * TPB < need; row += 4) {
      float2 v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int rel = w0 + tid + (row + k) * TPB, hh = a.HH + rel;
        const float2 *p = rel >= 0 ? irow + min(rel, a.N - 1) : hrow + max(hh, 0);
        v[k] = *p;
        if (rel >= a.N || hh < 0) v[k] = make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int i = tid + (row + k) * TPB;
        if (i < need) put(v[k], row + k, i);
      }
    }
    zero_tail(need);
  };
  auto compute = [&](int t) {
    const int j0 = t * (TPB * R), outs_here = outs_of(t);
    if (R * tid >= outs_here) return;
    float sr[R], si[R];
#pragma unroll
    for (int r = 0; r < R; r++) { sr[r] = 0.f; si[r] = 0.f; }
    const cfp bp = betap_c + (R - 1) * D;
    const int steps = a.M + (R - 1) * D;
    constexpr int U = 16, PADZ = (R - 1) * D, WN = U + PADZ;
    const float2 *pu = xs + tid * (RD + PD);
    for (int u0 = 0; u0 < steps; u0 += U) {
      float tw[WN];
#pragma unroll
      for (int k = 0; k < WN; k++) tw[k] = bp[u0 - PADZ + k];
      float2 x[U];
#pragma unroll
      for (int uu = 0; uu < U; uu += 2) {
        const float4 v = *reinterpret_cast<const float4 *>(pu + uu);
        x[uu] = make_float2(v.x, v.y); x[uu + 1] = make_float2(v.z, v.w);
      }
#pragma unroll
      for (int uu = 0; uu < U; uu++) {
#pragma unroll
        for (int r = 0; r < R; r++) {
          const float b = tw[uu - r * D + PADZ];
          sr[r] = __builtin_fmaf(b, x[uu].x, sr[r]); si[r] = __builtin_fmaf(b, x[uu].y, si[r]);
        }
      }
      pu += U;
      if (((u0 + U) % RD) == 0) pu += PD;
    }
    if (a.shift_on) {
      float2 P = a.parg[t & 31];
      if (!a.p_in_args) { const v2f P_ = ptab_c[t]; P = make_float2(P_.x, P_.y); }
#pragma unroll
      for (int r = 0; r < R; r++) { const float yr = sr[r] * P.x - si[r] * P.y, yi = sr[r] * P.y + si[r] * P.x; sr[r] = yr; si[r] = yi; }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int j = j0 + R * tid + r;
      if (j < a.n_out) {
        if (a.epilogue == SDRHIP_EPI_NONE) reinterpret_cast<float2 *>(a.out)[(long)c * a.out_stride + j] = make_float2(sr[r], si[r]);
        else if (a.epilogue == SDRHIP_EPI_AM) reinterpret_cast<float *>(a.out)[(long)c * a.out_stride + j] = sqrtf(sr[r] * sr[r] + si[r] * si[r]);
        else reinterpret_cast<float *>(a.out)[(long)c * a.out_stride + j] = (sr[r] + si[r]) / 2;
      }
    }
  };

  if (t_begin < t_end) {
    if (fast(t_begin)) { fetch(t_begin); commit(t_begin); } else stage_plain(t_begin);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();
    // (tuning variant -DK3_PRIO=1: the short phases between the barriers — issuing the next tile's loads, staging it into LDS — at
    // priority 3, the long multiply-add loop at 0, so that a workgroup in a memory phase is not queued behind the arithmetic of the
    // other three on its CU. Measured +-0 on the float baseband, three interleaved runs: 0.1359 / 0.1362 against 0.1360 / 0.1357 /
    // 0.1355 ms — what helped the one-workgroup-per-CU FFT kernel does nothing where four workgroups already interleave. Off.)
#ifndef K3_PRIO
#define K3_PRIO 0
#endif
    for (int t = t_begin; t < t_end; t++) {
      const bool more = t + 1 < t_end, nf = more && fast(t + 1);
      if (K3_PRIO) __builtin_amdgcn_s_setprio(3);
      if (nf) fetch(t + 1);   // in flight while this tile is computed
      if (K3_PRIO) __builtin_amdgcn_s_setprio(0);
      compute(t);
      if (more) {
        if (K3_PRIO) __builtin_amdgcn_s_setprio(3);
        __syncthreads();   // every lane is done with tile t's samples
        if (nf) commit(t + 1); else stage_plain(t + 1);
        // nothing of this iteration stays in flight across the back edge: the compiler's wait-count model merges the
        // border path's loads into the loop head otherwise and waits for the prefetch right after issuing it
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        __syncthreads();
      }
    }
  }
  if (a.roll && t_end == a.tiles) {   // (hist_old is only read, by this launch's border tiles; hist_new only written here)
    for (int k = tid; k < a.HH; k += TPB) {
      const long qq = (long)a.N + k;
      a.hist_new[(long)c * a.HH + k] = qq < a.HH ? a.hist_old[(long)c * a.HH + qq] : a.in[(long)c * a.in_stride + (qq - a.HH)];
    }
  }
}

// per-tile phase factors of the fused frequency shift: P[t] = exp(-2 pi i frac(fc (n_first + t * step) / fs)), float64
__global__ void tile_phasor_kernel(float2 *ptab, int tiles, long long n_first, long long step, double fc, double fs) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= tiles) return;
  const double n = (double)(n_first + (long long)t * step);
  const double turns = fmod(__ddiv_rn(__dmul_rn(fc, n), fs), 1.0);
  double sn, cs;
  sincos(__dmul_rn(-2.0 * M_PI, turns), &sn, &cs);
  ptab[t] = make_float2((float)cs, (float)sn);
}

// history roll for K3 (separate tiny kernel: a call may produce zero outputs, i.e. zero tiles)
__global__ void hist_roll_cf32(const float2 *in, long in_stride, const float2 *hist_old, float2 *hist_new, int HH, int N) {
  const int c = blockIdx.y;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < HH; k += gridDim.x * blockDim.x) {
    const long qq = (long)N + k;
    hist_new[(long)c * HH + k] = qq < HH ? hist_old[(long)c * HH + qq] : in[(long)c * in_stride + (qq - HH)];
  }
}

}  // namespace

struct sdrhip_fir {
  sdrhip_ctx *ctx = nullptr;
  int kind = 0, order = 0, D = 1, C = 1, epi = 0;
  size_t max_in = 0, max_out = 0;
  uint64_t n0 = 0;
  int par = 0, par_fm = 0;
  // K2
  int OP = 0, HH = 0, ovl = 0, R2 = 4;   // R2: outputs per lane of the exact kernel (chosen at create)
  bool wrap = false;
  DevBuf<double> alpha;
  DevBuf<uint32_t> hist16[2];
  DevBuf<short> fm[2];
  // K3
  int M = 0, R = 1;
  bool pipe = true;   // D = 8: several tiles per workgroup, software-pipelined (SDRHIP_FIR_PIPE=0: one tile per workgroup, tuning/tests)
  DevBuf<float> beta, betap;
  size_t lds3 = 0;
  // fused frequency shift (set by the float baseband)
  // complex<float>, no decimation, no demodulator: the filter runs as overlap-save FFT convolution behind this handle (see create)
  sdrhip_fftconv *fftc = nullptr;
  bool shift_on = false; double fc = 0, fs = 1;
  long long phase0 = 0;   // absolute sample index at which the fused shift's phasor was last (re)started (set_shift)
  DevBuf<float2> etab, etab2, wtab, ptab;   // phase tables of the fused shift (see Fir32Args)
  DevBuf<float2> hist32[2];
  // staging
  DevBuf<uint8_t> stage_in, stage_out;

  size_t in_elem() const { return kind == SDRHIP_FIR_CS16_EXACT ? 4 : 8; }
  size_t out_elem() const {
    if (kind == SDRHIP_FIR_CS16_EXACT) return epi == SDRHIP_EPI_NONE ? 4 : 2;
    return epi == SDRHIP_EPI_NONE ? 8 : 4;
  }
  // SubSample emits after every D-th input counted from the last reset (src/subsample.hh:94-99)
  size_t out_count(size_t N) const { return (size_t)((n0 + N) / (uint64_t)D - n0 / (uint64_t)D); }

  // K3 at D = 8: consecutive tiles per workgroup of the pipelined kernel (1: the plain kernel, one tile per workgroup) —
  // as many as leave >= 8 workgroups per CU
  int pipe_tpw(int tiles) const {
    if (!(D == 8 && (R == 2 || R == 4) && pipe)) return 1;
    int tpw = 16;
    while (tpw > 1 && ceil_div((size_t)tiles, (size_t)tpw) * (size_t)C < 8 * (size_t)ctx->prop.multiProcessorCount) tpw >>= 1;
    if (const char *e = getenv("SDRHIP_FIR_TPW")) tpw = std::max(1, atoi(e));   // tuning hook
    return tpw;
  }
  const char *kernel_name(size_t N) const {
    if (kind == SDRHIP_FIR_CS16_EXACT) return "fir_cs16_exact_kernel";
    const int tiles = (int)ceil_div(out_count(N), (size_t)TPB * R);
    return pipe_tpw(tiles) > 1 ? "fir_cf32_pipe_kernel" : "fir_cf32_rt_kernel";
  }

  void launch(const void *in_dev, size_t N, size_t in_stride, void *out_dev, size_t out_stride, size_t *n_out) {
    ctx->use();
    if (N == 0) { if (n_out) *n_out = 0; return; }
    const size_t no = out_count(N);
    SDRHIP_REQUIRE(out_stride >= no, SDRHIP_E_SIZE, "out_stride %zu < outputs %zu", out_stride, no);
    if (kind == SDRHIP_FIR_CS16_EXACT) {
      Fir16Args a;
      a.in = (const uint32_t *)in_dev; a.in_stride = (long)in_stride;
      a.hist_old = hist16[par].p; a.hist_new = hist16[par ^ 1].p; a.HH = HH;
      a.fm_old = fm[par_fm].p; a.fm_new = fm[par_fm ^ 1].p;
      const int T2 = TPB * R2;
      a.alpha = alpha.p; a.OP = OP; a.N = (int)N; a.ovl = ovl; a.OT = T2 - ovl;
      a.out = out_dev; a.out_stride = (long)out_stride; a.epilogue = epi;
      const int tiles = (int)ceil_div(N, (size_t)a.OT);
      const size_t lds = ((size_t)T2 + OP + 8 + T2) * 4;
      dim3 grid(tiles, C), block(TPB);
      if (R2 == 8) {
        if (wrap) hipLaunchKernelGGL((fir_cs16_exact_kernel<true, 8>), grid, block, lds, ctx->stream, a);
        else hipLaunchKernelGGL((fir_cs16_exact_kernel<false, 8>), grid, block, lds, ctx->stream, a);
      } else {
        if (wrap) hipLaunchKernelGGL((fir_cs16_exact_kernel<true, 4>), grid, block, lds, ctx->stream, a);
        else hipLaunchKernelGGL((fir_cs16_exact_kernel<false, 4>), grid, block, lds, ctx->stream, a);
      }
      SDRHIP_CHECK_HIP(hipGetLastError());
      if (epi == SDRHIP_EPI_FM && N >= 2) par_fm ^= 1;
    } else {
      Fir32Args a;
      a.in = (const float2 *)in_dev; a.in_stride = (long)in_stride;
      a.hist_old = hist32[par].p; a.hist_new = hist32[par ^ 1].p; a.HH = M - 1;
      a.beta = beta.p; a.M = M; a.D = D; a.N = (int)N;
      // first output of this call: group g = n0/D completes at absolute index g*D + D-1 >= n0
      const uint64_t g = n0 / (uint64_t)D;
      a.first_rel = (int)((int64_t)(g * D + D - 1) - (int64_t)n0);
      a.n_out = (int)no;
      a.out = out_dev; a.out_stride = (long)out_stride; a.epilogue = epi;
      a.betap = betap.p; a.rd_magic = (unsigned)((0x100000000ull + (uint64_t)(R * D) - 1) / (uint64_t)(R * D));
      a.shift_on = shift_on ? 1 : 0; a.etab = etab.p; a.wtab = wtab.p; a.ptab = ptab.p; a.etab2 = etab2.p;
      if (no) {
        const int tiles = (int)ceil_div(no, (size_t)TPB * R);
        a.p_in_args = 0; a.roll = M > 1 ? 1 : 0;
        if (shift_on) {   // sample 0 of tile t is absolute index n0 + first_rel - (M-1) + t * TPB*R*D
          const long long n_first = (long long)n0 - phase0 + a.first_rel - (M - 1), step = (long long)TPB * R * D;
          if (tiles <= 32) {
            a.p_in_args = 1;
            for (int t = 0; t < tiles; t++) {
              const double ph = -2.0 * M_PI * std::fmod(fc * (double)(n_first + t * step) / fs, 1.0);
              a.parg[t] = make_float2((float)std::cos(ph), (float)std::sin(ph));
            }
          } else {
            if (ptab.n < (size_t)tiles) ptab.alloc((size_t)tiles + 64);
            a.ptab = ptab.p;
            hipLaunchKernelGGL(tile_phasor_kernel, dim3((unsigned)ceil_div((size_t)tiles, (size_t)64)), dim3(64), 0, ctx->stream, ptab.p, tiles,
                               n_first, step, fc, fs);
          }
        }
        dim3 grid(tiles, C), block(TPB);
        // D = 8: the pipelined form, `tpw` consecutive tiles per workgroup while that leaves >= 8 workgroups per CU
        const int tpw = pipe_tpw(tiles);
        a.tiles = tiles; a.tpw = tpw;
        dim3 gridp((unsigned)ceil_div((size_t)tiles, (size_t)tpw), C);
        if (R == 4 && D == 8 && tpw > 1) hipLaunchKernelGGL((fir_cf32_pipe_kernel<4, 8>), gridp, block, lds3, ctx->stream, a);
        else if (R == 2 && D == 8 && tpw > 1) hipLaunchKernelGGL((fir_cf32_pipe_kernel<2, 8>), gridp, block, lds3, ctx->stream, a);
        else if (R == 4 && D == 8) hipLaunchKernelGGL((fir_cf32_rt_kernel<4, 8>), grid, block, lds3, ctx->stream, a);
        else if (R == 2 && D == 8) hipLaunchKernelGGL((fir_cf32_rt_kernel<2, 8>), grid, block, lds3, ctx->stream, a);
        else if (R == 4) hipLaunchKernelGGL((fir_cf32_rt_kernel<4, 0>), grid, block, lds3, ctx->stream, a);
        else if (R == 2) hipLaunchKernelGGL((fir_cf32_rt_kernel<2, 0>), grid, block, lds3, ctx->stream, a);
        else hipLaunchKernelGGL((fir_cf32_rt_kernel<1, 0>), grid, block, lds3, ctx->stream, a);
        SDRHIP_CHECK_HIP(hipGetLastError());
      }
      if (M > 1 && !no) {   // (a call that completes no output launches no tile: the history still rolls)
        dim3 grid((unsigned)ceil_div((size_t)(M - 1), (size_t)256), C);
        hipLaunchKernelGGL(hist_roll_cf32, grid, dim3(256), 0, ctx->stream, a.in, a.in_stride, a.hist_old, a.hist_new, M - 1, (int)N);
        SDRHIP_CHECK_HIP(hipGetLastError());
      }
    }
    par ^= 1;
    n0 += N;
    if (n_out) *n_out = no;
  }
};

namespace sdrhip {
// the coefficient tables of a plan from `order` doubles (create, and set_taps between calls: the uploads are ordered on the
// context's stream behind the launches already enqueued; buffers are allocated by create)
void fir_load_taps(sdrhip_fir *h, const double *alpha) {
  const int order = h->order, decim = h->D;
  if (h->kind == SDRHIP_FIR_CS16_EXACT) {
    std::vector<double> ap(h->OP, 0.0);
    double P = 0, Q = 0;   // sum of the positive / |negative| taps
    for (int i = 0; i < order; i++) {
      ap[h->OP - order + i] = alpha[i];
      if (alpha[i] >= 0) P += alpha[i]; else Q -= alpha[i];
    }
    // Truncation toward zero never grows a partial sum, so |acc| <= sum |alpha_k x_k| with
    // x in [-32768, 32767]. The int16 wrap of the reference can only trigger if a partial sum
    // can reach +32768 or go below -32768; otherwise acc stays in range and the wrap is skipped.
    const bool pos_ok = 32767.0 * P + 32768.0 * Q < 32768.0 - 1e-6;
    const bool neg_ok = 32768.0 * P + 32767.0 * Q < 32769.0 - 1e-6;
    h->wrap = !(pos_ok && neg_ok) || !(P == P && Q == Q);
    h->alpha.upload(ap.data(), h->OP, h->ctx->stream);
    return;
  }
  // beta[m] = (1/D) * sum_k alpha[m-k], k in [0,D): FIR followed by the D-sample box average
  std::vector<float> b(h->M);
  for (int m = 0; m < h->M; m++) {
    double s = 0;
    for (int k = 0; k < decim; k++) { const int i = m - k; if (i >= 0 && i < order) s += alpha[i]; }
    b[m] = (float)(s / decim);
  }
  h->beta.upload(b.data(), h->M, h->ctx->stream);
  const int padz = (h->R - 1) * decim;
  std::vector<float> bpad((size_t)h->M + 2 * padz + 16, 0.f);
  for (int m = 0; m < h->M; m++) bpad[padz + m] = b[m];
  h->betap.upload(bpad.data(), bpad.size(), h->ctx->stream);
}

// FIRFilter<complex<float>> is a tolerance path (<= 1e-5: the reference itself sits 4e-7 from the exact convolution,
// src/firfilter.hh:231-247 with src/operators.hh:16-18), and without decimation the time-domain kernel pays `order` multiply-adds
// per sample where an overlap-save FFT convolution pays a constant: 1024 channels x 65536 samples, ms per call, time domain /
// FFT — 8 taps 0.35 / 0.25, 127 taps 1.50 / 0.24, 1023 taps 10.4 / 0.36, 4097 taps 60.2 / 0.46
// (tools/probes/fir_cf32_vs_fft.py). So such a plan IS an overlap-save plan on the tuned FFT kernels (fftconv.hip): h[k] =
// alpha[order - 1 - k], transform 2048 points up to 512 taps, 4096 up to 2048, 16384 beyond. SDRHIP_FIR_TIME_DOMAIN=1 keeps the
// time-domain kernel (tests); decimating plans (the float baseband: the folded, shift-fused kernel) and fused demodulators keep it.
static std::vector<float> fir_fft_taps(const double *alpha, int order) {
  std::vector<float> t(2 * (size_t)order, 0.f);
  for (int k = 0; k < order; k++) t[2 * k] = (float)alpha[order - 1 - k];
  return t;
}

// (re)starts the fused shift's phasor at the CURRENT sample: exp(-2 pi i fc (n - n_now) / fs) from the next call on — what
// FreqShiftBase::setFrequencyShift does to its LUT counter (src/freqshift.hh:78-87). FIR history (raw input samples),
// decimator phase and sample counter go on.
void fir_set_shift(sdrhip_fir *h, double fc, double fs) {
  h->shift_on = true; h->fc = fc; h->fs = fs; h->phase0 = (long long)h->n0;
  h->ctx->use();
  auto ph = [&](double t) {   // exp(-2 pi i frac(fc t / fs)), the oracle's closed form, rounded to float once
    const double a = -2.0 * M_PI * std::fmod(fc * t / fs, 1.0);
    return make_float2((float)std::cos(a), (float)std::sin(a));
  };
  std::vector<float2> e(TPB), e2(TPB), w(256);
  for (int t = 0; t < TPB; t++) { e[t] = ph((double)t); e2[t] = ph((double)(2 * t)); }
  for (int k = 0; k < 256; k++) w[k] = ph((double)TPB * k);
  if (!h->etab.p) { h->etab.alloc(TPB); h->etab2.alloc(TPB); h->wtab.alloc(256); }   // (a retune rewrites the tables in stream order)
  h->etab.upload(e.data(), TPB, h->ctx->stream);
  h->etab2.upload(e2.data(), TPB, h->ctx->stream);
  h->wtab.upload(w.data(), 256, h->ctx->stream);
}
}  // namespace sdrhip

extern "C" {

}  // extern "C"

namespace sdrhip {
// allow_fft = false: the time-domain kernel whatever the plan (the float baseband fuses its frequency shift into that kernel's staging)
int fir_create_impl(sdrhip_ctx *ctx, int kind, const double *alpha, int order, int decim, int channels,
                    size_t max_in, int epilogue, bool allow_fft, sdrhip_fir **out) {
  return guarded([&] {
    SDRHIP_REQUIRE(ctx && alpha && out, SDRHIP_E_INVALID, "NULL argument");
    *out = nullptr;
    SDRHIP_REQUIRE(kind == SDRHIP_FIR_CS16_EXACT || kind == SDRHIP_FIR_CF32, SDRHIP_E_INVALID, "bad kind %d", kind);
    SDRHIP_REQUIRE(order >= 1 && order <= 8192, SDRHIP_E_UNSUPPORTED, "order %d outside [1,8192]", order);
    SDRHIP_REQUIRE(decim >= 1, SDRHIP_E_INVALID, "decim %d < 1", decim);
    SDRHIP_REQUIRE(channels >= 1 && channels <= 65535, SDRHIP_E_INVALID, "channels %d outside [1,65535]", channels);
    SDRHIP_REQUIRE(max_in >= 1 && max_in < (size_t(1) << 30), SDRHIP_E_SIZE, "max_in %zu outside [1,2^30)", max_in);
    SDRHIP_REQUIRE(epilogue >= SDRHIP_EPI_NONE && epilogue <= SDRHIP_EPI_USB, SDRHIP_E_INVALID, "bad epilogue %d", epilogue);
    if (kind == SDRHIP_FIR_CS16_EXACT)
      SDRHIP_REQUIRE(decim == 1, SDRHIP_E_UNSUPPORTED, "the exact int16 FIR does not decimate (chain a SubSample)");
    else
      SDRHIP_REQUIRE(epilogue != SDRHIP_EPI_FM, SDRHIP_E_UNSUPPORTED,
                     "FMDemod<float> does not exist in the reference (fast_atan2 has no float form)");
    ctx->use();
    sdrhip_fir *h = new sdrhip_fir;
    try {
      h->ctx = ctx; h->kind = kind; h->order = order; h->D = decim; h->C = channels; h->epi = epilogue;
      h->max_in = max_in; h->max_out = max_in / decim + 1;
      if (kind == SDRHIP_FIR_CS16_EXACT) {
        // outputs per lane: 8 when full-size calls still leave >= 8 workgroups per CU, else 4
        h->R2 = (ceil_div(max_in, (size_t)(TPB * 8)) * (size_t)channels >= 8 * (size_t)ctx->prop.multiProcessorCount) ? 8 : 4;
        h->OP = (int)ceil_div((size_t)order, (size_t)(h->R2 + U2)) * (h->R2 + U2);   // whole periods of the kernel's circular window
        h->HH = h->OP - 1;
        h->ovl = epilogue == SDRHIP_EPI_FM ? 1 : 0;
        SDRHIP_REQUIRE(((size_t)TPB * R2MAX + h->OP + 8 + TPB * R2MAX) * 4 <= 64 * 1024, SDRHIP_E_UNSUPPORTED, "order %d exceeds the LDS tile", order);
        h->alpha.alloc(h->OP);
        fir_load_taps(h, alpha);
        for (int p = 0; p < 2; p++) {
          h->hist16[p].alloc((size_t)channels * h->HH); h->hist16[p].zero(ctx->stream);
          h->fm[p].alloc(channels); h->fm[p].zero(ctx->stream);
        }
      } else if (allow_fft && decim == 1 && epilogue == SDRHIP_EPI_NONE && !(getenv("SDRHIP_FIR_TIME_DOMAIN") && getenv("SDRHIP_FIR_TIME_DOMAIN")[0] != '0') &&
                 // measured crossover (tools/probes/fir_cf32_small.py, profiles/r18_fir_cf32_small.txt): a block transform costs a call
                 // 7 us however little it filters — up to 32 taps on plans of at most 2^18 samples per call the time-domain kernel
                 // is the faster one (4.6 - 6.8 us); everywhere else the FFT plan wins (127 taps: 7 against 14 us on ONE channel)
                 (order > 32 || (size_t)channels * max_in > ((size_t)1 << 18) || getenv("SDRHIP_FIR_FFT_ALWAYS"))) {
        const std::vector<float> t = fir_fft_taps(alpha, order);
        const int rc = sdrhip_fftconv_create(ctx, SDRHIP_FFTCONV_OLS, ols_fft_size(order, (size_t)channels, max_in, ctx->prop.multiProcessorCount), t.data(), order, channels, max_in, &h->fftc);
        if (rc != SDRHIP_OK) throw Failure{rc};
      } else {
        // beta[m] = (1/D) * sum_k alpha[m-k], k in [0,D): FIR followed by the D-sample box average
        h->M = order + decim - 1;
        auto tile_bytes = [&](int R_) { const size_t need = ((size_t)TPB * R_ - 1) * decim + h->M; return (need + (size_t)(R_ - 1) * decim + 16 + (decim == 8 ? 2 : 1) * ((need + (size_t)(R_ - 1) * decim + 16) / ((size_t)R_ * decim) + 2)) * 8; };
        h->R = 4;
        if (const char *e = getenv("SDRHIP_FIR_R")) h->R = std::max(1, std::min(4, atoi(e)));   // tuning hook
        if (const char *e = getenv("SDRHIP_FIR_PIPE")) h->pipe = e[0] != '0';
        // as many outputs per lane as keep the tile under 40 KB (4 workgroups = 16 waves per CU): more waves in
        // flight beat more reuse per LDS read — D = 8: R = 2 measured 25 % faster than R = 4 (2 workgroups per CU)
        // (R*D stays >= 2: the pad-per-R*D-samples layout needs an even lane stride)
        while (h->R > 1 && tile_bytes(h->R) > 40 * 1024 && (h->R / 2) * decim >= 2 && !getenv("SDRHIP_FIR_RFORCE")) h->R >>= 1;   // (RFORCE: tuning, keep SDRHIP_FIR_R)
        SDRHIP_REQUIRE(tile_bytes(h->R) <= 144 * 1024, SDRHIP_E_UNSUPPORTED, "order %d with decim %d exceeds the LDS tile", order, decim);
        h->lds3 = tile_bytes(h->R);
        if (h->lds3 > 64 * 1024) {
          const void *fns[7] = {(const void *)fir_cf32_rt_kernel<4, 8>, (const void *)fir_cf32_rt_kernel<4, 0>, (const void *)fir_cf32_rt_kernel<2, 8>,
                                (const void *)fir_cf32_rt_kernel<2, 0>, (const void *)fir_cf32_rt_kernel<1, 0>,
                                (const void *)fir_cf32_pipe_kernel<4, 8>, (const void *)fir_cf32_pipe_kernel<2, 8>};
          for (int k = 0; k < 7; k++) allow_lds_max(fns[k], h->lds3);   // (once per kernel and device, to the hardware's maximum)
        }
        h->beta.alloc(h->M);
        h->betap.alloc((size_t)h->M + 2 * (size_t)(h->R - 1) * decim + 16);
        fir_load_taps(h, alpha);
        for (int p = 0; p < 2; p++) {
          h->hist32[p].alloc((size_t)channels * std::max(1, h->M - 1)); h->hist32[p].zero(ctx->stream);
        }
      }
      SDRHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    } catch (...) { if (h->fftc) (void)sdrhip_fftconv_destroy(h->fftc); delete h; throw; }
    *out = h;
  });
}
}  // namespace sdrhip

extern "C" {

int sdrhip_fir_create(sdrhip_ctx *ctx, int kind, const double *alpha, int order, int decim, int channels,
                      size_t max_in, int epilogue, sdrhip_fir **out) {
  return fir_create_impl(ctx, kind, alpha, order, decim, channels, max_in, epilogue, true, out);
}

int sdrhip_fir_kernel_names(sdrhip_fir *h, size_t n_in, char *buf, size_t len) {
  return guarded([&] {
    SDRHIP_REQUIRE(h && buf && len, SDRHIP_E_INVALID, "NULL argument");
    if (h->fftc) { snprintf(buf, len, "fftconv_fused_kernel"); return; }
    snprintf(buf, len, "%s", h->kernel_name(n_in ? n_in : h->max_in));
  });
}

int sdrhip_fir_out_count(sdrhip_fir *h, size_t n_in, size_t *n_out) {
  return guarded([&] {
    SDRHIP_REQUIRE(h && n_out, SDRHIP_E_INVALID, "NULL argument");
    *n_out = h->out_count(n_in);
  });
}

int sdrhip_fir_process_dev(sdrhip_fir *h, const void *in_dev, size_t n_in, size_t in_stride, void *out_dev,
                           size_t out_stride, size_t *n_out) {
  return guarded([&] {
    Range roctx_range("sdrhip_fir_process_dev");
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    SDRHIP_REQUIRE(n_in <= h->max_in, SDRHIP_E_SIZE, "n_in %zu > max_in %zu", n_in, h->max_in);
    if (n_in == 0) { if (n_out) *n_out = 0; return; }
    SDRHIP_REQUIRE(in_dev && out_dev, SDRHIP_E_INVALID, "NULL buffer");
    if (h->fftc) {
      const int rc = sdrhip_fftconv_process_dev(h->fftc, static_cast<const float *>(in_dev), n_in, in_stride, static_cast<float *>(out_dev), out_stride);
      if (rc != SDRHIP_OK) throw Failure{rc};
      h->n0 += n_in;
      if (n_out) *n_out = n_in;
      return;
    }
    if (in_stride == 0) in_stride = n_in;
    SDRHIP_REQUIRE(in_stride >= n_in, SDRHIP_E_SIZE, "in_stride %zu < n_in %zu", in_stride, n_in);
    if (out_stride == 0) out_stride = h->out_count(n_in);
    {
      const size_t ie = h->kind == SDRHIP_FIR_CS16_EXACT ? 4 : 8;
      const size_t oe = h->kind == SDRHIP_FIR_CS16_EXACT ? (h->epi == SDRHIP_EPI_NONE ? 4 : 2) : (h->epi == SDRHIP_EPI_NONE ? 8 : 4);
      require_disjoint(in_dev, in_stride, n_in, ie, out_dev, out_stride, h->out_count(n_in), oe, (size_t)h->C);
    }
    h->launch(in_dev, n_in, in_stride, out_dev, out_stride, n_out);
  });
}

int sdrhip_fir_process(sdrhip_fir *h, const void *in_host, size_t n_in, size_t in_stride, void *out_host,
                       size_t out_stride, size_t *n_out) {
  return guarded([&] {
    Range roctx_range("sdrhip_fir_process");
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    SDRHIP_REQUIRE(n_in <= h->max_in, SDRHIP_E_SIZE, "n_in %zu > max_in %zu", n_in, h->max_in);
    if (n_in == 0) { if (n_out) *n_out = 0; return; }
    SDRHIP_REQUIRE(in_host && out_host, SDRHIP_E_INVALID, "NULL buffer");
    if (h->fftc) {
      const int rc = sdrhip_fftconv_process(h->fftc, static_cast<const float *>(in_host), n_in, in_stride, static_cast<float *>(out_host), out_stride);
      if (rc != SDRHIP_OK) throw Failure{rc};
      h->n0 += n_in;
      if (n_out) *n_out = n_in;
      return;
    }
    h->ctx->use();
    if (in_stride == 0) in_stride = n_in;
    const size_t no = h->out_count(n_in);
    if (out_stride == 0) out_stride = no;
    SDRHIP_REQUIRE(out_stride >= no, SDRHIP_E_SIZE, "out_stride %zu < outputs %zu", out_stride, no);
    const size_t ib = h->in_elem(), ob = h->out_elem();
    if (!h->stage_in.p) {
      h->stage_in.alloc((size_t)h->C * h->max_in * ib);
      h->stage_out.alloc((size_t)h->C * h->max_out * ob);
    }
    copy_h2d_rows(h->ctx, h->stage_in.p, n_in * ib, in_host, in_stride * ib, n_in * ib, h->C);
    size_t produced = 0;
    h->launch(h->stage_in.p, n_in, n_in, h->stage_out.p, h->max_out, &produced);
    copy_d2h_rows(h->ctx, out_host, out_stride * ob, h->stage_out.p, h->max_out * ob, produced * ob, h->C);
    SDRHIP_CHECK_HIP(hipStreamSynchronize(h->ctx->stream));
    if (n_out) *n_out = produced;
  });
}

int sdrhip_fir_set_taps(sdrhip_fir *h, const double *alpha) {
  return guarded([&] {
    SDRHIP_REQUIRE(h && alpha, SDRHIP_E_INVALID, "NULL argument");
    h->ctx->use();
    if (h->fftc) {   // (the overlap-save history is input samples: the stream goes on under the new taps, as the ring does)
      const std::vector<float> t = fir_fft_taps(alpha, h->order);
      const int rc = sdrhip_fftconv_set_kernel(h->fftc, 0, t.data());
      if (rc != SDRHIP_OK) throw Failure{rc};
      return;
    }
    fir_load_taps(h, alpha);
  });
}

int sdrhip_fir_reset(sdrhip_fir *h) {
  return guarded([&] {
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    h->ctx->use();
    h->n0 = 0; h->phase0 = 0;
    if (h->fftc) { const int rc = sdrhip_fftconv_reset(h->fftc); if (rc != SDRHIP_OK) throw Failure{rc}; return; }
    for (int p = 0; p < 2; p++) {
      h->hist16[p].zero(h->ctx->stream); h->hist32[p].zero(h->ctx->stream); h->fm[p].zero(h->ctx->stream);
    }
  });
}

int sdrhip_fir_destroy(sdrhip_fir *h) {
  return guarded([&] {
    if (!h) return;
    h->ctx->use();
    (void)hipStreamSynchronize(h->ctx->stream);
    if (h->fftc) (void)sdrhip_fftconv_destroy(h->fftc);
    delete h;
  });
}

}  // extern "C"
