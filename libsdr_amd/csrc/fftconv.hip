// fftconv.hip — K7: FFT convolution filter with a hand-written in-LDS FFT (no rocFFT/hipFFT).
//
// Replaces (reference, file:line):
//   FilterSink<float>::process      src/filternode.hh:81-88   (zero-pad + forward FFT, FFTW3f)
//   FilterSource<float>::process    src/filternode.hh:164-181 (spectrum multiply, inverse FFT, /2N, overlap-add)
//   FFTPlan<float> / FFT::exec      src/fftplan.hh:14-36, src/fftplan_fftw3.hh:79-142
//
// The reference's overlap-ADD with block N, FFT size 2N and an N-tap kernel is the causal linear
// convolution y[n] = sum_{k<N} h[k] x[n-k]. We evaluate that same convolution by
// overlap-SAVE: block b transforms the L samples ending at its last output and keeps the last
// `hop` results, so blocks (and channels) are independent — no tail carried between workgroups,
// any call length is accepted, and BASELINE config 4's (L=16384, M=4097, hop 12288) mode is the
// same kernel with a different hop. One workgroup (1024 lanes) owns one block: L complex floats
// live in LDS (128 KiB at L=16384), the forward transform is an in-place radix-4 DIF (result in
// digit-reversed order), the kernel spectrum is stored pre-permuted and pre-scaled by 1/L, and
// the inverse is the mirrored in-place DIT, so no reordering pass is ever executed.
#include "sdrhip_internal.hpp"

#include <cmath>
#include <complex>
#include <memory>

#include "fftgen.hpp"
#include "fftany.hpp"

using namespace sdrhip;

// float tolerance path (<= 1e-5 relative): fused multiply-adds are welcome here (the library is built with
// -ffp-contract=off for the bit-exact fp64 FIR)
#pragma clang fp contract(fast)

namespace {

constexpr int FT = 1024;       // lanes per workgroup
constexpr int MAX_PASS = 16;

struct FftDev {
  int L, npass;
  int radix[MAX_PASS];   // forward pass order (DIF); the inverse walks it backwards
  const float2 *W;       // W[t] = exp(-2 pi i t / L)
  const float2 *T;       // per radix-16 pass q: T[toff[q] + (k-1)*s_q + j] = W^(j tw_q k), k = 1..15 (coalesced along j)
  int toff[MAX_PASS];
};

// Complex arithmetic on register PAIRS: one packed instruction handles (re, im) together, and the op_sel / neg modifiers
// of the packed forms (which 32-bit half of each source feeds the low and the high result, negated or not) fold the
// multiplications by +-i into the add that follows. Written as instructions: left to the compiler's SLP vectoriser the
// same arithmetic came out as 3 packed instructions per complex product (each computing a half that is thrown away) and
// one v_mov_b32 per 4 arithmetic instructions to re-pair halves — 1 924 vector instructions per wave and block of the
// 16384-point filter where the kernel is bound by vector issue (r08's counters: 61 % busy, no other unit above 30 %).
typedef float v2f __attribute__((ext_vector_type(2)));
#define PK2(name_, text_)                                                                                             \
  __device__ __forceinline__ float2 name_(float2 a, float2 b) {                                                        \
    v2f d;                                                                                                             \
    asm(text_ : "=v"(d) : "v"(__builtin_bit_cast(v2f, a)), "v"(__builtin_bit_cast(v2f, b)));                           \
    return __builtin_bit_cast(float2, d);                                                                              \
  }
PK2(cadd, "v_pk_add_f32 %0, %1, %2")
PK2(csub, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]")
PK2(cadd_mi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")   // a + (-i) b = (a.x + b.y, a.y - b.x)
PK2(cadd_pi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")   // a + (+i) b = (a.x - b.y, a.y + b.x)
PK2(pk_mul_xx, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]")              // (a.x b.x, a.y b.x)
#undef PK2
// a * b = (a.x b.x - a.y b.y, a.y b.x + a.x b.y) and a * conj(b): a packed multiply and a packed fused multiply-add
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  const float2 t = pk_mul_xx(a, b);
  v2f d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"
      : "=v"(d) : "v"(__builtin_bit_cast(v2f, a)), "v"(__builtin_bit_cast(v2f, b)), "v"(__builtin_bit_cast(v2f, t)));
  return __builtin_bit_cast(float2, d);
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {
  const float2 t = pk_mul_xx(a, b);
  v2f d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"
      : "=v"(d) : "v"(__builtin_bit_cast(v2f, a)), "v"(__builtin_bit_cast(v2f, b)), "v"(__builtin_bit_cast(v2f, t)));
  return __builtin_bit_cast(float2, d);
}
// the same with a wave-uniform constant factor in a scalar register pair
template <bool CONJ>
__device__ __forceinline__ float2 cmul_k(float2 a, float wr, float wi) {
  const v2f w = {wr, wi};
  v2f t, d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(__builtin_bit_cast(v2f, a)), "s"(w));
  if (CONJ) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(d) : "v"(__builtin_bit_cast(v2f, a)), "s"(w), "v"(t));
  else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(__builtin_bit_cast(v2f, a)), "s"(w), "v"(t));
  return __builtin_bit_cast(float2, d);
}
__device__ __forceinline__ float2 mul_mi(float2 a) { return cadd_mi(make_float2(0.f, 0.f), a); }   // a * (-i)
__device__ __forceinline__ float2 mul_pi(float2 a) { return cadd_pi(make_float2(0.f, 0.f), a); }   // a * (+i)
// the radix-4 butterfly every pass is made of: (x0..x3) -> (X0..X3), X_m = sum_k x_k exp(SIGN 2 pi i k m / 4); 8 packed adds
template <int SIGN>
__device__ __forceinline__ void bfly4(float2 x0, float2 x1, float2 x2, float2 x3, float2 &X0, float2 &X1, float2 &X2, float2 &X3) {
  const float2 t0 = cadd(x0, x2), t1 = csub(x0, x2), t2 = cadd(x1, x3), d = csub(x1, x3);
  X0 = cadd(t0, t2); X2 = csub(t0, t2);
  if (SIGN < 0) { X1 = cadd_mi(t1, d); X3 = cadd_pi(t1, d); }   // t1 -+ i d
  else { X1 = cadd_pi(t1, d); X3 = cadd_mi(t1, d); }
}

// LDS index of element i: 4 pad elements after every 64, so that the stride-4 radix-16 pass (lanes 64 elements
// = 512 B apart) spreads over the banks instead of hitting one
__device__ __forceinline__ int PAD(int i) { return i + ((i >> 6) << 2); }

// in-place 16-point DFT, v[m] <- sum_k v[k] exp(SIGN 2 pi i k m / 16), as 4 x 4 (k = a + 4b, m = c + 4d)
template <int SIGN>
__device__ __forceinline__ void dft16(float2 *v) {
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
  // step 1: for each a, the 4-point DFT over b of (v[a], v[a+4], v[a+8], v[a+12]) -> Y_a[c] kept at v[a + 4c]
#pragma unroll
  for (int a = 0; a < 4; a++) bfly4<SIGN>(v[a], v[a + 4], v[a + 8], v[a + 12], v[a], v[a + 4], v[a + 8], v[a + 12]);
  // step 2: Z_a[c] = W16^(a c) Y_a[c]   (W16 = exp(SIGN 2 pi i / 16)); a c in {1,2,3,2,4,6,3,6,9}: the factor is
  // (wr, SIGN wi), i.e. the constant (wr, wi) or its conjugate
  constexpr bool CJ = SIGN < 0;
  v[1 + 4] = cmul_k<CJ>(v[1 + 4], C1, S1);           // a=1,c=1: W^1 = (cos pi/8, SIGN sin pi/8)
  v[1 + 8] = cmul_k<CJ>(v[1 + 8], H, H);             // a=1,c=2: W^2
  v[1 + 12] = cmul_k<CJ>(v[1 + 12], S1, C1);         // a=1,c=3: W^3
  v[2 + 4] = cmul_k<CJ>(v[2 + 4], H, H);             // a=2,c=1: W^2
  v[2 + 8] = SIGN < 0 ? mul_mi(v[2 + 8]) : mul_pi(v[2 + 8]);   // a=2,c=2: W^4 = SIGN i
  v[2 + 12] = cmul_k<CJ>(v[2 + 12], -H, H);          // a=2,c=3: W^6
  v[3 + 4] = cmul_k<CJ>(v[3 + 4], S1, C1);           // a=3,c=1: W^3
  v[3 + 8] = cmul_k<CJ>(v[3 + 8], -H, H);            // a=3,c=2: W^6
  v[3 + 12] = cmul_k<CJ>(v[3 + 12], -C1, -S1);       // a=3,c=3: W^9 = -W^1
  // step 3: for each c, the 4-point DFT over a of Z_a[c] (at v[a + 4c]) -> X[c + 4d]
  float2 o[16];
#pragma unroll
  for (int c = 0; c < 4; c++) bfly4<SIGN>(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3], o[c], o[c + 4], o[c + 8], o[c + 12]);
#pragma unroll
  for (int m = 0; m < 16; m++) v[m] = o[m];
}

// The last two passes of a plan that ends in radix 4 (stride 2) and radix 2 (stride 1) — 2048 = 16 x 16 x 4 x 2 — work
// inside groups of 8 CONSECUTIVE elements: one lane runs both on a group held in registers (same butterflies, same
// twiddles W8^(j m), same element order as the two LDS passes they replace, so the spectrum layout is unchanged).
// forward (decimation in frequency): radix 4 over (e[j], e[j+2], e[j+4], e[j+6]), j = 0, 1, then radix 2 over the pairs
__device__ __forceinline__ void dif8_fwd(float2 *e) {
  constexpr float H = 0.70710678118654752f;
  float2 a0, a1, a2, a3, b0, b1, b2, b3;
  bfly4<-1>(e[0], e[2], e[4], e[6], a0, a1, a2, a3);
  bfly4<-1>(e[1], e[3], e[5], e[7], b0, b1, b2, b3);
  b1 = cmul_k<true>(b1, H, H);      // W8^1 = (H, -H)
  b2 = mul_mi(b2);                  // W8^2 = -i
  b3 = cmul_k<true>(b3, -H, H);     // W8^3 = (-H, -H)
  e[0] = cadd(a0, b0); e[1] = csub(a0, b0); e[2] = cadd(a1, b1); e[3] = csub(a1, b1);
  e[4] = cadd(a2, b2); e[5] = csub(a2, b2); e[6] = cadd(a3, b3); e[7] = csub(a3, b3);
}
// backward (decimation in time): radix 2 over the pairs, then radix 4 with the conjugate twiddles
__device__ __forceinline__ void dit8_inv(float2 *e) {
  constexpr float H = 0.70710678118654752f;
  const float2 a0 = cadd(e[0], e[1]), b0 = csub(e[0], e[1]), a1 = cadd(e[2], e[3]), b1 = csub(e[2], e[3]);
  const float2 a2 = cadd(e[4], e[5]), b2 = csub(e[4], e[5]), a3 = cadd(e[6], e[7]), b3 = csub(e[6], e[7]);
  bfly4<1>(a0, a1, a2, a3, e[0], e[2], e[4], e[6]);
  bfly4<1>(b0, cmul_k<false>(b1, H, H), mul_pi(b2), cmul_k<false>(b3, -H, H), e[1], e[3], e[5], e[7]);
}

// twiddles W^(j tw k), k = 1..15, of radix-16 pass q for butterfly column j: straight from the pass's own table
// (lanes walk consecutive j: 15 coalesced loads that hit L1/L2). Deriving them from 4 table entries with 11 complex
// products cost a quarter of the kernel (ablation: 0.80 -> 0.61 ms).
__device__ __forceinline__ void twiddles16(const FftDev &p, int q, int s, int j, float2 *w) {
  const float2 *t = p.T + p.toff[q] + j;
#ifndef K7_TW_MIN
#define K7_TW_MIN 4
#endif
#ifndef K7_TW_LOADS
#define K7_TW_LOADS 2
#endif
#ifndef K7_TW_ALL
  if (s >= K7_TW_MIN) {   // the big first / last pass: its table (15 s entries) does not stay in L1 — 4 loads and 11 products (2 packed
                    // instructions each) instead of 15 loads through L2
#if K7_TW_LOADS == 1
    const float2 w1 = t[0], w2 = cmul(w1, w1), w4 = cmul(w2, w2), w8 = cmul(w4, w4);
#elif K7_TW_LOADS == 2
    const float2 w1 = t[0], w2 = cmul(w1, w1), w4 = t[3 * s], w8 = cmul(w4, w4);
#else
    const float2 w1 = t[0], w2 = t[s], w4 = t[3 * s], w8 = t[7 * s];
#endif
    w[1] = w1; w[2] = w2; w[4] = w4; w[8] = w8;
    w[3] = cmul(w1, w2); w[5] = cmul(w4, w1); w[6] = cmul(w4, w2); w[7] = cmul(w4, w[3]);
    w[9] = cmul(w8, w1); w[10] = cmul(w8, w2); w[11] = cmul(w8, w[3]); w[12] = cmul(w8, w4);
    w[13] = cmul(w8, w[5]); w[14] = cmul(w8, w[6]); w[15] = cmul(w8, w[7]);
    return;
  }
#endif
#pragma unroll
  for (int k = 1; k < 16; k++) w[k] = t[(k - 1) * s];
}

// the 15 twiddles of a radix-16 pass from the two seeds w1 = W^(j tw), w4 = W^(4 j tw) the 2-load form reads (already in registers),
// applied as they are made, v[k] *= w^k (CONJ: the conjugates): 8 twiddles live instead of 15 (the pipelined form holds the
// next block's inputs in 32 registers through the passes that use this)
template <bool CONJ>
__device__ __forceinline__ void twiddle_apply_seeded(float2 *v, float2 w1, float2 w4) {
  auto mul = [](float2 x, float2 w) { return CONJ ? cmulc(x, w) : cmul(x, w); };
  const float2 w2 = cmul(w1, w1), w3 = cmul(w1, w2);
  v[1] = mul(v[1], w1); v[2] = mul(v[2], w2); v[3] = mul(v[3], w3); v[4] = mul(v[4], w4);
  const float2 w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3), w8 = cmul(w4, w4);
  v[5] = mul(v[5], w5); v[6] = mul(v[6], w6); v[7] = mul(v[7], w7); v[8] = mul(v[8], w8);
  v[9] = mul(v[9], cmul(w8, w1)); v[10] = mul(v[10], cmul(w8, w2)); v[11] = mul(v[11], cmul(w8, w3)); v[12] = mul(v[12], cmul(w8, w4));
  v[13] = mul(v[13], cmul(w8, w5)); v[14] = mul(v[14], cmul(w8, w6)); v[15] = mul(v[15], cmul(w8, w7));
}

// forward, decimation in frequency: natural order in, digit-reversed order out
__device__ void fft_forward_dif(float2 *x, const FftDev &p, int tid) {
  int n = p.L;
  for (int pass = 0; pass < p.npass; pass++) {
    const int r = p.radix[pass], s = n / r, tw = p.L / n;
    if (r == 16) {
      for (int b = tid; b < p.L / 16; b += FT) {
        const int j = b & (s - 1), base = (b / s) * n + j;
        float2 v[16], w[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = x[PAD(base + k * s)];
        dft16<-1>(v);
        if (s > 1) {
          twiddles16(p, pass, s, j, w);
#pragma unroll
          for (int k = 1; k < 16; k++) v[k] = cmul(v[k], w[k]);
        }
#pragma unroll
        for (int k = 0; k < 16; k++) x[PAD(base + k * s)] = v[k];
      }
    } else if (r == 4) {
      for (int b = tid; b < p.L / 4; b += FT) {
        const int j = b & (s - 1), base = (b / s) * n + j;
        const float2 a0 = x[PAD(base)], a1 = x[PAD(base + s)], a2 = x[PAD(base + 2 * s)], a3 = x[PAD(base + 3 * s)];
        float2 X0, X1, X2, X3;
        bfly4<-1>(a0, a1, a2, a3, X0, X1, X2, X3);
        x[PAD(base)] = X0;
        x[PAD(base + s)] = cmul(X1, p.W[j * tw]);
        x[PAD(base + 2 * s)] = cmul(X2, p.W[2 * j * tw]);
        x[PAD(base + 3 * s)] = cmul(X3, p.W[3 * j * tw]);
      }
    } else {   // radix 2
      for (int b = tid; b < p.L / 2; b += FT) {
        const int j = b & (s - 1), base = (b / s) * n + j;
        const float2 a0 = x[PAD(base)], a1 = x[PAD(base + s)];
        x[PAD(base)] = cadd(a0, a1);
        x[PAD(base + s)] = cmul(csub(a0, a1), p.W[j * tw]);
      }
    }
    __syncthreads();
    n = s;
  }
}

// backward (unnormalised), decimation in time: digit-reversed order in, natural order out
__device__ void fft_inverse_dit(float2 *x, const FftDev &p, int tid) {
  int n = 1;
  for (int pass = p.npass - 1; pass >= 0; pass--) {
    const int r = p.radix[pass], s = n;
    n *= r;
    const int tw = p.L / n;
    if (r == 16) {
      for (int b = tid; b < p.L / 16; b += FT) {
        const int j = b & (s - 1), base = (b / s) * n + j;
        float2 v[16], w[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = x[PAD(base + k * s)];
        if (s > 1) {
          twiddles16(p, pass, s, j, w);
#pragma unroll
          for (int k = 1; k < 16; k++) v[k] = cmulc(v[k], w[k]);
        }
        dft16<1>(v);
#pragma unroll
        for (int k = 0; k < 16; k++) x[PAD(base + k * s)] = v[k];
      }
    } else if (r == 4) {
      for (int b = tid; b < p.L / 4; b += FT) {
        const int j = b & (s - 1), base = (b / s) * n + j;
        const float2 a0 = x[PAD(base)];
        const float2 a1 = cmulc(x[PAD(base + s)], p.W[j * tw]);
        const float2 a2 = cmulc(x[PAD(base + 2 * s)], p.W[2 * j * tw]);
        const float2 a3 = cmulc(x[PAD(base + 3 * s)], p.W[3 * j * tw]);
        float2 X0, X1, X2, X3;
        bfly4<1>(a0, a1, a2, a3, X0, X1, X2, X3);
        x[PAD(base)] = X0; x[PAD(base + s)] = X1; x[PAD(base + 2 * s)] = X2; x[PAD(base + 3 * s)] = X3;
      }
    } else {
      for (int b = tid; b < p.L / 2; b += FT) {
        const int j = b & (s - 1), base = (b / s) * n + j;
        const float2 a0 = x[PAD(base)], a1 = cmulc(x[PAD(base + s)], p.W[j * tw]);
        x[PAD(base)] = cadd(a0, a1);
        x[PAD(base + s)] = csub(a0, a1);
      }
    }
    __syncthreads();
  }
}

struct ConvArgs {
  FftDev fft;
  const float2 *in; long in_stride;
  const float2 *hist; int HH;        // HH = L - hop: the samples of a block in front of the ones it keeps
  int HL, delay;                     // history rows hold the HL samples preceding the call (HH + delay); the block's window starts `delay`
                                     // samples earlier in the stream (partitioned convolution: the later tap partitions, fftconv_fused_kernel ACC)
  float2 *hist_new;                  // fused kernel: the channel's last block also writes the history of the next call (NULL: not this launch)
  const float2 *Kp;                  // spectra (band b at Kp + b*L), digit-reversed order, pre-scaled by 1/L
  float2 *out; long out_stride;
  int N, hop;
  int nb; long out_band;             // filter bank: bands sharing ONE forward transform; band b's rows start at out + b*out_band
  int lds_elems;                     // padded elements of one LDS image
  int nblk, nchan;                   // (PIPE form) blocks per channel and channels: the launch's grid no longer says
  float2 *dump;                      // (PIPE form) 128 bytes per wave of the launch where the masked lanes' stores go
#ifdef K7_STAMPS
  unsigned long long *stamps;        // diagnostic build: 16 words per workgroup (8 phase totals in shader clocks, turns, start, end)
#endif
};

__global__ __launch_bounds__(FT) void fftconv_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float2 xl[];
  const int c = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
  const int L = a.fft.L;
  const int first = blk * a.hop - a.HH - a.delay;   // call-relative index of xl[0]
  for (int i = tid; i < L; i += FT) {
    const int rel = first + i;
    float2 v = make_float2(0.f, 0.f);
    if (rel >= 0) { if (rel < a.N) v = a.in[(long)c * a.in_stride + rel]; }
    else { const int h = a.HL + rel; if (h >= 0) v = a.hist[(long)c * a.HL + h]; }
    xl[PAD(i)] = v;
  }
  __syncthreads();
  fft_forward_dif(xl, a.fft, tid);
  float2 *xw = a.nb > 1 ? xl + a.lds_elems : xl;
  for (int band = 0; band < a.nb; band++) {
    for (int i = tid; i < L; i += FT) xw[PAD(i)] = cmul(xl[PAD(i)], a.Kp[(long)band * L + i]);
    __syncthreads();
    fft_inverse_dit(xw, a.fft, tid);
    const int o0 = blk * a.hop;
    for (int i = tid; i < a.hop; i += FT) {
      const int o = o0 + i;
      if (o < a.N) a.out[(long)band * a.out_band + (long)c * a.out_stride + o] = xw[PAD(a.HH + i)];
    }
    __syncthreads();
  }
}

// value of the neighbouring lane (lane ^ 1): DPP quad_perm [1,0,3,2]
__device__ __forceinline__ float lane_xor1(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));
}

// Fused form of the same convolution (plans whose first pass is radix 16 and that have at least two passes, i.e.
// every L >= 32): the first forward pass takes its 16 inputs per butterfly straight from global memory (history /
// call input / zeros) and the last inverse pass stores its outputs straight to global memory — lanes walk
// consecutive j, so both are coalesced — and the last forward pass, the spectrum product and the first inverse pass
// (same element groups: stride 1) are one LDS round trip. 6 LDS reads + 6 writes of the block instead of 11 + 11.
__device__ __forceinline__ float2 conv_fetch(const ConvArgs &a, int c, int rel) {
  if (rel >= 0) return rel < a.N ? a.in[(long)c * a.in_stride + rel] : make_float2(0.f, 0.f);
  const int h = a.HL + rel;
  return h >= 0 ? a.hist[(long)c * a.HL + h] : make_float2(0.f, 0.f);
}

// LG > 0: the plan (L = 2^LG: radix-16 passes, then 4 and/or 2) is a compile-time constant — strides, pad offsets and
// the per-pass butterfly maps fold into immediates and shifts (instantiated for the BASELINE size 16384); LG = 0: the
// plan is read from the arguments
constexpr int plan_npass(int lg) { return lg / 4 + ((lg % 4) >= 2 ? 1 : 0) + ((lg % 4) & 1); }
constexpr int plan_radix(int lg, int pass) { return pass < lg / 4 ? 16 : ((lg % 4) >= 2 && pass == lg / 4) ? 4 : 2; }
// BANK: several bands behind one forward transform (forward image kept, one work image); false = the single-band
// kernel exactly as before (in place, no band loop — the 1024-lane workgroup sits at the 128-VGPR cap and any extra
// live value spills: the runtime band loop alone cost 36 %)
// NT: lanes per workgroup — L / 16 (one radix-16 butterfly per lane and pass), at least one wave, at most 1024: a 2048-point
// filter-bank block on 1024 lanes kept 7 of 8 lanes idle in the radix-16 passes and, one workgroup at a time per CU pair of
// images, nothing overlapped its latencies
// ACC: the block's kept samples are ADDED to what the output rows hold (partitioned convolution: a filter of more taps than one
// 16384-point block can carry runs as two passes over the same input — tap partition 0, then partition 1 on the window 8192
// samples earlier, accumulated: y = h0 (*) x + z^-8192 (h1 (*) x))
// PIPE (16384 points, one band): a persistent workgroup per CU walks its blocks in a loop and the NEXT block's 16 inputs
// per lane are in flight (32 registers) while the last two inverse passes of the current one run. Nothing else can overlap
// the global phases of this kernel — one image fills the CU's LDS, so no second workgroup is resident — and no extra LDS is
// needed: butterfly j of the last inverse pass READS the image elements j + 1024 k and butterfly j of the next block's pass 0
// WRITES the same elements, both on lane j, so the hand-over between blocks is lane-private (no barrier; the block's stores
// drain while the next pass 0 runs). vmcnt counts in issue order: every other global load of the tail (the twiddles of the
// last two inverse passes) is issued BEFORE the prefetch, or waiting for it would wait for the prefetch as well. And the wait
// counts the compiler inserts are only exact along straight-line code — where two paths with different numbers of outstanding
// loads or stores join it waits for everything (first version: `s_waitcnt vmcnt(0)` at the top of inverse pass 1, the prefetch
// waited for on the spot; measured +-0) — so from the prefetch to the next pass 0 every path issues the SAME loads and stores:
// edge blocks (history, ragged end, no next block) select a pointer per lane (a harmless one for what is not there) instead of
// branching around the load, mask at consumption, and masked stores go to a dump line instead of being skipped; the history
// roll runs behind the walk.
// PIPE = 4: 16-byte loads / stores by lane pairs (everything even: hop, history, strides, N, 16-byte aligned rows), 2: 8-byte
// ones (any alignment). SKIP: the first SKIP stores of a lane are in front of the kept samples for every lane
// (HH >= SKIP x 2048 resp. 1024) and do not exist.
template <int LG, bool BANK, int NT, bool ACC = false, int PIPE = 0, int SKIP = 0>
__global__ __launch_bounds__(NT) void fftconv_fused_kernel(const ConvArgs a) {
  static_assert(!PIPE || (LG == 14 && !BANK && NT == 1024 && !ACC), "the pipelined form exists for the 16384-point single-band plan");
  constexpr bool PV = PIPE == 4;
  constexpr int FT = NT;   // (shadows the file-wide workgroup size)
  extern __shared__ __attribute__((aligned(16))) float2 xl[];
  const FftDev &p = a.fft;
  // TAIL8 (the 2048-point plan 16 x 16 x 4 x 2, the filter-bank node's default block): the radix-4 (stride 2) and radix-2
  // (stride 1) passes run in REGISTERS on groups of 8 consecutive elements (dif8_fwd / dit8_inv), two groups per lane — one
  // LDS round trip and one barrier less per transform, and the two passes whose 16-byte-strided accesses were 4-way bank
  // conflicts (r12's counters: 55 % of this kernel's LDS cycles) are gone. The image is padded 2 elements per 32 instead
  // of 4 per 64: a lane's 64-byte group then starts 16 (t >> 2) + 64 t bytes in, so the 16 lanes of one ds_read_b128 /
  // ds_write_b128 cover all 16 slots of a bank row, and the radix-16 stride-8 pass (8 lanes per 64 contiguous bytes,
  // neighbouring octets 1088 bytes apart) stays conflict-free as before.
#ifdef K7_NO_TAIL8   // (A/B: the four-pass LDS form)
  constexpr bool TAIL8 = false;
#else
  constexpr bool TAIL8 = LG == 11 && NT == 128;
#endif
  auto P = [](int i) { return TAIL8 ? i + ((i >> 5) << 1) : PAD(i); };
  const int L = LG ? (1 << LG) : p.L, np = LG ? plan_npass(LG) : p.npass;
  const int np_lds = TAIL8 ? np - 1 : np;   // passes that go through LDS (TAIL8: the last two are one register step)
  auto radix_at = [&](int q) { return LG ? plan_radix(LG, q) : p.radix[q]; };
  // (one workgroup per (channel, block). A persistent grid — the resident workgroups walking the units in a loop —
  // measured 4.5 % SLOWER: a 16384-point block fills the CU's LDS, so either way one workgroup runs per CU, but the
  // dispatcher starts the next workgroup's waves while the last one's stores drain, and the loop's closing barrier does not.
  // Round 3 tried it again WITH a register prefetch of the next block's input issued before the last inverse pass: 512
  // lanes (registers to spare) 0.535 ms, 1024 lanes (prefetch once the twiddles are dead, 114 registers) 0.524 ms, against
  // 0.508 ms for this kernel on the same box)
  // XCD-aware (channel, block) assignment (xcd_unit_order): consecutive blocks of a channel overlap by the filter's
  // history — a quarter of a 16384-point block with 4097 taps, half of a filter-bank block
  int c = 0, blk = 0;
  const int tid0 = threadIdx.x;
  // (every phase starts from an opaque copy of the lane index: the lane's LDS and table addresses of all passes are
  // otherwise computed up front and, kept live across the phases, spill — 49 to 93 registers in the run-time-plan kernels)
  // (the compile-time 16384-point plan fits without: there the hoisted addresses are worth 5 %)
  // (PIPE: opaque as well — inside the walk's loop everything derived from the lane index is loop-invariant, gets hoisted in front of
  // the loop and spilled there: 79 registers' worth, reloaded one by one through scratch, i.e. through vmcnt)
  auto lane = [&]() { int t = tid0; if (LG == 0 || PIPE) { asm volatile("" : "+v"(t)); __builtin_assume(t >= 0 && t < NT); } return t; };
  // WAVE_LOCAL: the passes between the first forward and the last inverse pass need no workgroup barrier (see below). 16384 points
  // on 1024 lanes (a segment = one wave). -DK7_WL_MID=1: also 8192 points on 512 lanes and 4096 on 256 (a segment = 32 / 16 lanes of
  // ONE wave; the radix-2 butterflies of the 8192-point middle pass re-mapped to the lane group's own segment) — parity green, time
  // +-0 on FilterNode(2048) / (4096) / (1000) (0.114 / 0.096 / 0.087 ms either way: two to eight workgroups per CU cover each other's
  // barriers), so they keep their barrier per pass
#ifndef K7_WL_MID
#define K7_WL_MID 0
#endif
  constexpr bool WAVE_LOCAL = !BANK && ((LG == 14 && NT == 1024) || (K7_WL_MID && ((LG == 13 && NT == 512) || (LG == 12 && NT == 256))));
  // (K7_PRIO, 16384 points) between two barriers the waves of a SIMD run at their own pace and the arbiter prefers the oldest: wave 0
  // arrives at the next barrier 12 600 clocks of a 32 300-clock turn before the last one (stamps), and the last one runs alone, its
  // LDS latencies uncovered. Priority by progress — the further along, the lower — keeps the four together.
// sites: forward passes 1, 2, the middle pass, inverse pass 2 (and behind its butterfly), inverse pass 1 (and in front of its
// butterfly), the last pass (and behind its butterfly), pass 0 (and behind its butterfly); K7_PRIO picks the level table
#ifndef K7_PRIO
#define K7_PRIO 1   // (A/B on one box, 4097 taps: table 1 0.3868 ms, 3 0.3903, 2 0.3944, none 0.4135; -DK7_PRIO=0: none)
#endif
#if K7_PRIO
  enum { S_F1, S_F2, S_MID, S_I2, S_I2B, S_I1, S_I1B, S_LAST, S_LASTB, S_P0, S_P0B };
#if K7_PRIO == 1
#define K7_PRIO_TAB {3, 2, 2, 1, 1, 0, 0, 3, 3, 1, 1}
#elif K7_PRIO == 2
#define K7_PRIO_TAB {3, 3, 2, 1, 1, 0, 0, 3, 2, 1, 0}
#else
#define K7_PRIO_TAB {3, 3, 3, 2, 1, 1, 0, 3, 2, 1, 0}
#endif
#define K7_SETPRIO(site_) do { if (WAVE_LOCAL) { constexpr int tab_[] = K7_PRIO_TAB; __builtin_amdgcn_s_setprio(tab_[site_]); } } while (0)
#else
#define K7_SETPRIO(site_) do { } while (0)
#endif
#ifdef K7_STAMPS   // diagnostic build: wave 0 of every workgroup sums the shader clocks between its phase boundaries
  const bool st_on = PIPE && __builtin_amdgcn_readfirstlane(tid0) == 0;
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime(), st_turns = 0;
  const unsigned long long st_t0 = __builtin_amdgcn_s_memrealtime();
#define K7_STAMP(ph_, dep_) do { if (st_on) { asm volatile("" :: "v"(dep_) : "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    asm volatile("" ::: "memory"); st_acc[ph_] += now_ - st_t; st_t = now_; } } while (0)
#else
#define K7_STAMP(ph_, dep_) do { } while (0)
#endif
  // (PIPE) the workgroup's walk: with channels and workgroups in multiples of 8, XCD x = workgroup & 7 (round-robin dispatch) owns
  // channels x, x + 8, ... and its workgroups take that list's (channel, block) units turn by turn — as xcd_unit_order, the two
  // readers of a block overlap run on one XCD at about the same time
  const bool pipe_xcd = PIPE && (a.nchan & 7) == 0 && (gridDim.x & 7u) == 0;
  int uk = PIPE ? (pipe_xcd ? (int)(blockIdx.x >> 3) : (int)blockIdx.x) : 0;
  const int uk_step = PIPE ? (pipe_xcd ? (int)(gridDim.x >> 3) : (int)gridDim.x) : 1;
  const int uk_end = PIPE ? (pipe_xcd ? (a.nchan >> 3) * a.nblk : a.nchan * a.nblk) : 1;
  auto unit_of = [&](int k, int &ublk, int &uc) {
    const int q = k / a.nblk;
    ublk = k - q * a.nblk; uc = pipe_xcd ? (int)(blockIdx.x & 7u) + 8 * q : q;
  };
  float4 pfq[PIPE ? 8 : 1];   // the prefetched inputs of the lane's pass-0 butterfly: PV the raw lane-pair loads, else (element 2m, element 2m + 1)
  int pf_state = 0;           // 0: nothing fetched (no next block), 1: an interior block, 2: an edge block (masks at consumption)
  float2 pf_tw[4];            // twiddle seeds (w1, w4) of inverse pass 1 and of the last inverse pass, loaded ahead of the prefetch
  auto prefetch = [&](int k, bool have) {
    int ublk, uc;
    unit_of(k, ublk, uc);
    const int f = ublk * a.hop - a.HH - a.delay;
    const bool interior = have && f >= 0 && f + L <= a.N;
    pf_state = have ? (interior ? 1 : 2) : 0;
    const float2 *in_c = a.in + (long)uc * a.in_stride, *hist_c = a.hist + (long)uc * a.HL, *safe = a.Kp;
    const int j = lane(), s = L / 16, odd = j & 1, je = j & ~1;
    // where element e of the block lives (edge blocks): the call's input, the history rows, or nowhere (zeros in front of the
    // history / behind the call's end: `safe`, masked at consumption)
    auto where = [&](int e) {
      const int rel = f + e, h = a.HL + rel;
      const float2 *q = rel >= 0 ? (rel < a.N ? in_c + rel : safe) : (h >= 0 ? hist_c + h : safe);
      return have ? q : safe;
    };
    // the branch computes ADDRESSES only (no load on either side: the two sides of a branch are laid out one after the other
    // behind a flag, and the path that runs neither — impossible, but in the flow graph — would set every later wait to zero)
    constexpr int NP = PV ? 8 : 16;
    const float2 *ptr[NP];
    if (interior) {
      const float2 *src = in_c + f + (PV ? je : j);
#pragma unroll
      for (int n = 0; n < NP; n++) ptr[n] = src + (PV ? 2 * n + odd : n) * s;
    } else {
#pragma unroll
      for (int n = 0; n < NP; n++) ptr[n] = where((PV ? je + (2 * n + odd) * s : j + n * s));
    }
#pragma unroll
    for (int m = 0; m < 8; m++) {
      if (PV) pfq[m] = *reinterpret_cast<const float4 *>(ptr[m]);
      else { const float2 x = *ptr[2 * m], y = *ptr[2 * m + 1]; pfq[m] = make_float4(x.x, x.y, y.x, y.y); }
    }
  };
  // (PIPE) the lane's LDS addresses of the four access patterns (stride 1024: pass 0 and the last pass; stride 64; stride 4; the
  // middle pass's consecutive quadruples), computed ONCE in front of the walk: with the opaque lane index every pass of every turn
  // recomputed them (1 427 vector instructions per wave and block against the one-block kernel's 1 278), and these four — unlike
  // everything the compiler hoisted by itself — fit the registers that are free. Element base + off is lp[off + pad(off)] wherever
  // (base mod 64) + (off mod 64) < 64, which holds for all four.
  float2 *lp_s1024 = xl, *lp_s64 = xl, *lp_s4 = xl, *lp_mid = xl;
  if (PIPE) {
    lp_s1024 = xl + P(tid0); lp_s64 = xl + P((tid0 >> 6) * 1024 + (tid0 & 63)); lp_s4 = xl + P((tid0 >> 2) * 64 + (tid0 & 3));
    lp_mid = xl + P(4 * ((tid0 >> 6) * 256 + (tid0 & 63)));
  }
  auto X = [&](float2 *lp, int base, int off) -> float2 & { return PIPE ? lp[off + ((off >> 6) << 2)] : xl[P(base + off)]; };
  // (PIPE, tuning variant -DK7_LDS_TW=1) the twiddles of the stride-64 and stride-4 passes in LDS behind the image (15 x 64 + 15 x 4
  // entries, 8 KB of the 21 KB the image leaves), copied once per workgroup from the plan's tables: 15 ds_read_b64 per pass instead
  // of 2 global loads + 11 complex products. Measured +-0 (0.3655 against 0.3636 ms, profiles/r18_k7_pipe_ab.txt) and no longer
  // bit-identical to the one-block kernel (table entries against products): off.
#ifndef K7_LDS_TW
#define K7_LDS_TW 0
#endif
  constexpr bool LTW = PIPE && K7_LDS_TW;
  float2 *ltw64 = xl, *ltw4 = xl;
  if (LTW) {
    float2 *tab = xl + (L + (L >> 6) * 4);
    if (tid0 < 15 * 64) tab[tid0] = p.T[p.toff[1] + tid0];
    if (tid0 < 15 * 4) tab[15 * 64 + tid0] = p.T[p.toff[2] + tid0];
    ltw64 = tab + (tid0 & 63); ltw4 = tab + 15 * 64 + (tid0 & 3);   // (visible behind the barrier that closes the first pass 0)
  }
  // (PIPE) pass 0 of the block whose inputs the prefetch brought: registers -> LDS. Runs at the END of a turn (and once in front
  // of the loop), so that the prefetched registers are written and read inside one turn: carried around the loop's back edge the
  // register allocator moved two of the 32 to other registers there — a copy of a register a load is still writing, i.e. a full wait
  // (the two copies carry different marker comments: identical, the compiler merges them back into ONE at the loop's head)
  auto pass0_pipe = [&](int first, bool in_loop) __attribute__((always_inline)) {
    if (in_loop) asm volatile("; pass 0 of the next block (end of a turn)"); else asm volatile("; pass 0 of the first block");
    K7_SETPRIO(S_P0);
    const int j = lane(), s = L / 16;
    float2 v[16];
    if (PV) {
      const int odd = j & 1;
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const float4 q = pfq[m];
        const float gx = odd ? q.x : q.z, gy = odd ? q.y : q.w;
        const float rx = lane_xor1(gx), ry = lane_xor1(gy);
        v[2 * m] = odd ? make_float2(rx, ry) : make_float2(q.x, q.y);
        v[2 * m + 1] = odd ? make_float2(q.z, q.w) : make_float2(rx, ry);
      }
    } else {
#pragma unroll
      for (int m = 0; m < 8; m++) { v[2 * m] = make_float2(pfq[m].x, pfq[m].y); v[2 * m + 1] = make_float2(pfq[m].z, pfq[m].w); }
    }
    K7_STAMP(6, v[15].x + v[14].y);
    if (pf_state == 2) {   // an edge block: what lies in front of the history or behind the call's end is zero
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int rel = first + j + k * s;
        if (rel >= a.N || rel < -a.HL) v[k] = make_float2(0.f, 0.f);
      }
    }
    dft16<-1>(v);
    K7_SETPRIO(S_P0B);
    twiddle_apply_seeded<false>(v, pf_tw[2], pf_tw[3]);
#pragma unroll
    for (int k = 0; k < 16; k++) X(lp_s1024, j, k * s) = v[k];
    __syncthreads();
    K7_STAMP(7, j);
  };
  if (PIPE) {
    // the seeds of the stride-L/16 passes (the last inverse pass and pass 0 use the same table row, conjugated or not) stay in
    // registers for the whole walk
    const float2 *t0 = p.T + p.toff[0] + lane();
    pf_tw[2] = t0[0]; pf_tw[3] = t0[3 * (L / 16)];
    if (uk < uk_end) {
      prefetch(uk, true);
      unit_of(uk, blk, c);
      pass0_pipe(blk * a.hop - a.HH - a.delay, false);
    }
  } else {
#ifdef FFTCONV_NO_XCD   // (tuning: launch-order assignment)
    c = blockIdx.y; blk = blockIdx.x;
#else
    xcd_unit_order(blk, c);
#endif
  }
  // After the first radix-16 pass the transform splits into 16 independent segments of L/16 points, and with one
  // butterfly per lane (NT = L/16) the butterflies of a segment belong to consecutive lanes: for L = 16384 a segment is
  // exactly one WAVE's 64 lanes x 16 points. The passes between the first forward and the last inverse pass then touch
  // only data the same wave wrote — LDS operations of one wave execute in order, so they need no workgroup barrier: 2
  // barriers per block instead of 7 (r08's counters: 44 % of the wave cycles parked at them, all 16 waves in lock step).
  auto pass_sync = [&]() { if (WAVE_LOCAL) asm volatile("" ::: "memory"); else __syncthreads(); };
  // radix-4 butterfly q (0..3) of this lane: with WAVE_LOCAL the 256 butterflies of the wave's own segment
  auto bfly4_index = [&](int tid, int q) { return WAVE_LOCAL ? ((tid >> 6) * 256 + (tid & 63) + 64 * q) : (tid + q * FT); };
  // radix-2 butterfly q (0..7) of this lane (8192 points on 512 lanes): with WAVE_LOCAL the L / 32 butterflies of the segment the
  // lane's 32-lane group owns (in lane order a lane's eight butterflies lie in eight different segments)
  auto bfly2_index = [&](int tid, int q) { return ((tid / (NT / 16)) * (L / 32) + (tid % (NT / 16)) + (NT / 16) * q); };
  for (; uk < uk_end; uk += uk_step) {   // (one turn unless PIPE)
  if (PIPE) unit_of(uk, blk, c);
  const int first = blk * a.hop - a.HH - a.delay;   // call-relative index of element 0
  // ---- forward pass 0 (radix 16, stride L/16): global -> registers -> LDS ----
  if (PIPE) {
    // (pass 0 of this block ran at the end of the previous turn — or in front of the loop)
  } else {
    const int tid = lane();
    const int s = L / 16;
    // interior, 16-byte aligned block: a lane pair loads 16 bytes per lane (elements j&~1, (j&~1)+1 of every other k)
    // and swaps halves — 8 dwordx4 loads per lane instead of 16 dwordx2 (the per-CU load/store issue rate, not
    // HBM, bounded these phases)
    const float2 *src = a.in + (long)c * a.in_stride + first;
    const bool vec_in = first >= 0 && first + L <= a.N && ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && (s & 1) == 0 && s >= FT;
    for (int j = tid; j < s; j += FT) {
      float2 v[16], w[16];
#ifdef K7_PROBE_NOLOAD   // (ceiling probe, results wrong: pass 0 makes its inputs up — what the block costs with its global loads hidden completely)
      if (a.N > 0) {
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = make_float2((float)(j + k), (float)(tid - k));
      } else
#endif
      if (vec_in) {
        const int odd = j & 1, je = j & ~1;
#pragma unroll
        for (int m = 0; m < 8; m++) {
          const float4 q = *reinterpret_cast<const float4 *>(src + je + (2 * m + odd) * s);   // (E[k], O[k]), k = 2m + odd
          // even lane keeps E[2m] = q.xy and needs E[2m+1] = the odd lane's q.xy; odd lane keeps O[2m+1] = q.zw and
          // needs O[2m] = the even lane's q.zw
          const float gx = odd ? q.x : q.z, gy = odd ? q.y : q.w;
          const float rx = lane_xor1(gx), ry = lane_xor1(gy);
          v[2 * m] = odd ? make_float2(rx, ry) : make_float2(q.x, q.y);
          v[2 * m + 1] = odd ? make_float2(q.z, q.w) : make_float2(rx, ry);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = conv_fetch(a, c, first + j + k * s);
      }
      dft16<-1>(v);
      twiddles16(p, 0, s, j, w);
#pragma unroll
      for (int k = 1; k < 16; k++) v[k] = cmul(v[k], w[k]);
#pragma unroll
      for (int k = 0; k < 16; k++) xl[P(j + k * s)] = v[k];
    }
    __syncthreads();
  }
  // ---- forward passes 1 .. np-2 in LDS ----
  int n = L / 16;
  auto fwd_pass = [&](int pass) __attribute__((always_inline)) {
    if (pass == 1) K7_SETPRIO(S_F1); else K7_SETPRIO(S_F2);
    const int tid = lane();
    const int r = radix_at(pass), s = n / r, tw = L / n;
    if (r == 16) {
      for (int b = tid; b < L / 16; b += FT) {
        const int j = b & (s - 1), base = (b / s) * n + j;
        float2 v[16], w[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = X(s == 64 ? lp_s64 : lp_s4, base, k * s);
        dft16<-1>(v);
        if (LTW) {
#pragma unroll
          for (int k = 1; k < 16; k++) w[k] = (s == 64 ? ltw64 : ltw4)[(k - 1) * s];
        } else twiddles16(p, pass, s, j, w);
#pragma unroll
        for (int k = 1; k < 16; k++) v[k] = cmul(v[k], w[k]);
#pragma unroll
        for (int k = 0; k < 16; k++) X(s == 64 ? lp_s64 : lp_s4, base, k * s) = v[k];
      }
    } else {   // radix 4 (a radix-2 pass can only be the last one)
      for (int b = tid; b < L / 4; b += FT) {   // (not reached by the 16384-point plan: its radix-4 pass is the last one)
        const int j = b & (s - 1), base = (b / s) * n + j;
        const float2 a0 = xl[P(base)], a1 = xl[P(base + s)], a2 = xl[P(base + 2 * s)], a3 = xl[P(base + 3 * s)];
        float2 X0, X1, X2, X3;
        bfly4<-1>(a0, a1, a2, a3, X0, X1, X2, X3);
        xl[P(base)] = X0;
        xl[P(base + s)] = cmul(X1, p.W[j * tw]);
        xl[P(base + 2 * s)] = cmul(X2, p.W[2 * j * tw]);
        xl[P(base + 3 * s)] = cmul(X3, p.W[3 * j * tw]);
      }
    }
    pass_sync();
    n = s;
  };
  // (PIPE: written out — inside the walk's loop the compiler no longer unrolls the pass loops of the compile-time plan by itself, and
  // a rolled loop reads strides, radices and the twiddle form at run time)
  if constexpr (PIPE) { fwd_pass(1); fwd_pass(2); }
  else for (int pass = 1; pass + 1 < np_lds; pass++) fwd_pass(pass);
  // ---- filter bank (BANK): one forward transform per input block for all bands, as FilterSink feeds every FilterSource
  // from one FFT (reference src/filternode.hh:81-88,257-270). The workgroup has L / 16 lanes, so the last forward pass
  // leaves exactly 16 spectrum values per lane: they stay in REGISTERS across the bands (32 of them), every band
  // multiplies them by its own spectrum, runs the first inverse pass on them and writes the ONE LDS image the rest of
  // its inverse transform works in — the second image a bank used to need halved the workgroups per CU. ----
  float2 *xw = xl;
  const int nb = BANK ? a.nb : 1;
  float2 fwd[16];
  // (TAIL8) the lane's two groups of 8 consecutive elements: group g at elements 8 g .. 8 g + 7, four 16-byte pieces
  auto load8 = [&](const float2 *img, int g, float2 *e) {
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const float4 q = *reinterpret_cast<const float4 *>(img + P(8 * g + 2 * m));
      e[2 * m] = make_float2(q.x, q.y); e[2 * m + 1] = make_float2(q.z, q.w);
    }
  };
  auto store8 = [&](float2 *img, int g, const float2 *e) {
#pragma unroll
    for (int m = 0; m < 4; m++)
      *reinterpret_cast<float4 *>(img + P(8 * g + 2 * m)) = make_float4(e[2 * m].x, e[2 * m].y, e[2 * m + 1].x, e[2 * m + 1].y);
  };
  // which group a lane owns: lane order. (-DK7_GROUP_PERM: lane bits (0, 1, 2, 3) -> group bits (0, 2, 3, 1), which makes
  // the ds_write_b128 groups of 8 lanes conflict-free — in lane order they land on 4 of the 8 slots of a 128-byte row, two
  // by two: every conflict cycle the TAIL8 kernel has left, counter = model = 256 per wave and block — at the price of one
  // extra cycle on each ds_read_b128. Counters: SQ_LDS_BANK_CONFLICT 33.5 M -> 4.2 M of 180 M / 151 M LDS cycles; time,
  // three interleaved runs on one box: 1.040 -> 1.115 ms. The reads sit on the dependent path, the stores do not (a
  // ds_write_b128 costs its 13-cycle register transfer either way): the layout with FEWER conflicts is 7 % slower. Off.)
#ifdef K7_GROUP_PERM
  auto group_of = [](int t) { return (t & ~0xE) | (((t >> 1) & 1) << 2) | (((t >> 2) & 1) << 3) | (((t >> 3) & 1) << 1); };
#else
  auto group_of = [](int t) { return t; };
#endif
  if (BANK && TAIL8) {
    const int tid = group_of(lane());
    load8(xl, tid, fwd); load8(xl, tid + FT, fwd + 8);
    dif8_fwd(fwd); dif8_fwd(fwd + 8);
    __syncthreads();   // every lane holds its part of the spectrum: the image is free
  } else if (BANK) {
    const int tid = lane();
    const int r = radix_at(np - 1);
    if (r == 16) {
#pragma unroll
      for (int k = 0; k < 16; k++) fwd[k] = xl[P(16 * tid + k)];
      dft16<-1>(fwd);
    } else if (r == 4) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int b = tid + q * FT;
        bfly4<-1>(xl[P(4 * b)], xl[P(4 * b + 1)], xl[P(4 * b + 2)], xl[P(4 * b + 3)], fwd[4 * q], fwd[4 * q + 1], fwd[4 * q + 2], fwd[4 * q + 3]);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int b = tid + q * FT;
        const float2 a0 = xl[P(2 * b)], a1 = xl[P(2 * b + 1)];
        fwd[2 * q] = cadd(a0, a1); fwd[2 * q + 1] = csub(a0, a1);
      }
    }
    __syncthreads();   // every lane holds its part of the spectrum: the image is free
  }
  for (int band = 0; band < nb; band++) {
  const float2 *kp = BANK ? a.Kp + (long)band * L : a.Kp;
  float2 *outb = BANK ? a.out + (long)band * a.out_band : a.out;
  // ---- last forward pass (stride 1, no twiddles) x spectrum x first inverse pass ----
  if (TAIL8) {
    const int tid = group_of(lane());
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int g = tid + h * FT;
      float2 v[8];
      if (BANK) {
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = fwd[8 * h + k];
      } else {
        load8(xl, g, v);
        dif8_fwd(v);
      }
      const float4 *kq = reinterpret_cast<const float4 *>(kp + 8 * g);   // (the spectrum rows are 16-byte aligned: hipMalloc + 8 L bytes per band)
#pragma unroll
      for (int m = 0; m < 4; m++) {
        const float4 q = kq[m];
        v[2 * m] = cmul(v[2 * m], make_float2(q.x, q.y)); v[2 * m + 1] = cmul(v[2 * m + 1], make_float2(q.z, q.w));
      }
      dit8_inv(v);
      store8(xw, g, v);   // (in place: the group is this lane's own)
    }
    __syncthreads();
  } else if (BANK) {
    const int tid = lane();
    const int r = radix_at(np - 1);
    if (r == 16) {
      float2 v[16];
#pragma unroll
      for (int k = 0; k < 16; k++) v[k] = cmul(fwd[k], kp[16 * tid + k]);
      dft16<1>(v);
#pragma unroll
      for (int k = 0; k < 16; k++) xw[P(16 * tid + k)] = v[k];
    } else if (r == 4) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int b = tid + q * FT;
        float2 Z0, Z1, Z2, Z3;
        bfly4<1>(cmul(fwd[4 * q], kp[4 * b]), cmul(fwd[4 * q + 1], kp[4 * b + 1]), cmul(fwd[4 * q + 2], kp[4 * b + 2]), cmul(fwd[4 * q + 3], kp[4 * b + 3]), Z0, Z1, Z2, Z3);
        xw[P(4 * b)] = Z0; xw[P(4 * b + 1)] = Z1; xw[P(4 * b + 2)] = Z2; xw[P(4 * b + 3)] = Z3;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int b = tid + q * FT;
        const float2 y0 = cmul(fwd[2 * q], kp[2 * b]), y1 = cmul(fwd[2 * q + 1], kp[2 * b + 1]);
        xw[P(2 * b)] = cadd(y0, y1); xw[P(2 * b + 1)] = csub(y0, y1);
      }
    }
    __syncthreads();
  } else {
    K7_SETPRIO(S_MID);
    const int tid = lane();
    const float2 *xs = xl;
    const int r = radix_at(np - 1);
    if (r == 16) {
      for (int b = tid; b < L / 16; b += FT) {
        float2 v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = xs[P(16 * b + k)];
        dft16<-1>(v);
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = cmul(v[k], kp[16 * b + k]);
        dft16<1>(v);
#pragma unroll
        for (int k = 0; k < 16; k++) xw[P(16 * b + k)] = v[k];
      }
    } else if (r == 4) {
      for (int q = 0; q * FT + (WAVE_LOCAL ? 0 : tid) < L / 4; q++) {   // (small plans: fewer butterflies than lanes)
        const int b = bfly4_index(tid, q);
        float2 a0, a1, a2, a3;
        if (PIPE) { a0 = X(lp_mid, 0, 256 * q); a1 = X(lp_mid, 0, 256 * q + 1); a2 = X(lp_mid, 0, 256 * q + 2); a3 = X(lp_mid, 0, 256 * q + 3); }
        else { a0 = xs[P(4 * b)]; a1 = xs[P(4 * b + 1)]; a2 = xs[P(4 * b + 2)]; a3 = xs[P(4 * b + 3)]; }
        float2 X0, X1, X2, X3, Z0, Z1, Z2, Z3;
        bfly4<-1>(a0, a1, a2, a3, X0, X1, X2, X3);
        bfly4<1>(cmul(X0, kp[4 * b]), cmul(X1, kp[4 * b + 1]), cmul(X2, kp[4 * b + 2]), cmul(X3, kp[4 * b + 3]), Z0, Z1, Z2, Z3);
        if (PIPE) { X(lp_mid, 0, 256 * q) = Z0; X(lp_mid, 0, 256 * q + 1) = Z1; X(lp_mid, 0, 256 * q + 2) = Z2; X(lp_mid, 0, 256 * q + 3) = Z3; }
        else { xw[P(4 * b)] = Z0; xw[P(4 * b + 1)] = Z1; xw[P(4 * b + 2)] = Z2; xw[P(4 * b + 3)] = Z3; }
      }
    } else if (WAVE_LOCAL) {
#pragma unroll
      for (int q = 0; q < (L / 2) / NT; q++) {
        const int b = bfly2_index(tid, q);
        const float2 a0 = xs[P(2 * b)], a1 = xs[P(2 * b + 1)];
        const float2 y0 = cmul(cadd(a0, a1), kp[2 * b]), y1 = cmul(csub(a0, a1), kp[2 * b + 1]);
        xw[P(2 * b)] = cadd(y0, y1); xw[P(2 * b + 1)] = csub(y0, y1);
      }
    } else {
      for (int b = tid; b < L / 2; b += FT) {
        const float2 a0 = xs[P(2 * b)], a1 = xs[P(2 * b + 1)];
        const float2 y0 = cmul(cadd(a0, a1), kp[2 * b]), y1 = cmul(csub(a0, a1), kp[2 * b + 1]);
        xw[P(2 * b)] = cadd(y0, y1); xw[P(2 * b + 1)] = csub(y0, y1);
      }
    }
    pass_sync();
  }
  // ---- inverse passes np-2 .. 1 in LDS ----
  n = TAIL8 ? 8 : radix_at(np - 1);
  auto inv_pass = [&](int pass) __attribute__((always_inline)) {
    if (pass == 2) K7_SETPRIO(S_I2); else K7_SETPRIO(S_I1);
    const int tid = lane();
    const int r = radix_at(pass), s = n;
    n *= r;
    const int tw = L / n;
    if (PIPE && pass == 1) {   // the tail's own global loads first, then the next block's inputs (vmcnt is in issue order)
      if (!LTW) {
        const float2 *t1 = p.T + p.toff[1] + (tid & (s - 1));
        pf_tw[0] = t1[0]; pf_tw[1] = t1[3 * s];
      }
      K7_STAMP(0, tid);
      __builtin_amdgcn_sched_barrier(0);
      prefetch(uk + uk_step < uk_end ? uk + uk_step : uk, uk + uk_step < uk_end);
      __builtin_amdgcn_sched_barrier(0);
      K7_STAMP(1, tid);
    }
    if (r == 16) {
      for (int b = tid; b < L / 16; b += FT) {
        const int j = b & (s - 1), base = (b / s) * n + j;
        float2 v[16], w[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = X(s == 64 ? lp_s64 : lp_s4, base, k * s);   // (xw is xl: one band)
        if (LTW) {
#pragma unroll
          for (int k = 1; k < 16; k++) v[k] = cmulc(v[k], (s == 64 ? ltw64 : ltw4)[(k - 1) * s]);
        } else if (PIPE && pass == 1) twiddle_apply_seeded<true>(v, pf_tw[0], pf_tw[1]);
        else {
          twiddles16(p, pass, s, j, w);
#pragma unroll
          for (int k = 1; k < 16; k++) v[k] = cmulc(v[k], w[k]);
        }
        if (pass == 1) K7_SETPRIO(S_I1B);
        dft16<1>(v);
        if (pass == 2) K7_SETPRIO(S_I2B);
#pragma unroll
        for (int k = 0; k < 16; k++) { if (PIPE) X(s == 64 ? lp_s64 : lp_s4, base, k * s) = v[k]; else xw[P(base + k * s)] = v[k]; }
      }
    } else {
      for (int b = tid; b < L / 4; b += FT) {
        const int j = b & (s - 1), base = (b / s) * n + j;
        const float2 a0 = xw[P(base)];
        const float2 a1 = cmulc(xw[P(base + s)], p.W[j * tw]);
        const float2 a2 = cmulc(xw[P(base + 2 * s)], p.W[2 * j * tw]);
        const float2 a3 = cmulc(xw[P(base + 3 * s)], p.W[3 * j * tw]);
        float2 X0, X1, X2, X3;
        bfly4<1>(a0, a1, a2, a3, X0, X1, X2, X3);
        xw[P(base)] = X0; xw[P(base + s)] = X1; xw[P(base + 2 * s)] = X2; xw[P(base + 3 * s)] = X3;
      }
    }
    if (pass > 1) pass_sync();
    else {   // (the last inverse pass crosses the segments again)
#ifdef K7_STAMPS
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      K7_STAMP(2, tid);
      __syncthreads();
      K7_SETPRIO(S_LAST);
      K7_STAMP(3, tid);
    }
  };
  if constexpr (PIPE) { inv_pass(2); inv_pass(1); }
  else for (int pass = np_lds - 2; pass >= 1; pass--) inv_pass(pass);
  // ---- last inverse pass (radix 16, stride L/16): LDS -> registers -> global (only the hop kept samples) ----
  {
    const int tid = lane();
    const int s = L / 16, o0 = blk * a.hop;
    for (int j = tid; j < s; j += FT) {
      float2 v[16], w[16];
#pragma unroll
      for (int k = 0; k < 16; k++) v[k] = PIPE ? X(lp_s1024, j, k * s) : xw[P(j + k * s)];
      if (PIPE) twiddle_apply_seeded<true>(v, pf_tw[2], pf_tw[3]);
      else {
        twiddles16(p, 0, s, j, w);
#pragma unroll
        for (int k = 1; k < 16; k++) v[k] = cmulc(v[k], w[k]);
      }
      dft16<1>(v);
      K7_SETPRIO(S_LASTB);
      float2 *dst = outb + (long)c * a.out_stride + o0 - a.HH;   // element i of the block goes to dst[i] (i >= HH)
      K7_STAMP(4, v[0].x + v[15].y);
      if (PIPE) {   // every store is issued; lanes in front of the kept samples or behind the call's end write the wave's dump line
        float2 *dump = a.dump + ((long)blockIdx.x * (NT / 64) + (tid >> 6)) * 16;
        const int odd = j & 1, je = j & ~1;
        constexpr int NS = PV ? 8 : 16;
        float2 *to_[NS];
        const bool whole = SKIP > 0 && a.HH == SKIP * (PV ? 2 : 1) * s && o0 + (L - a.HH) <= a.N;   // stores SKIP ... are all kept samples
        if (whole) {
#pragma unroll
          for (int n = SKIP; n < NS; n++) to_[n] = dst + (PV ? je + (2 * n + odd) * s : j + n * s);
        } else {
#pragma unroll
          for (int n = SKIP; n < NS; n++) {
            const int i = PV ? je + (2 * n + odd) * s : j + n * s;
            to_[n] = (i >= a.HH && o0 - a.HH + i < a.N) ? dst + i : dump;
          }
        }
        if (PV) {
#pragma unroll
          for (int m = SKIP; m < 8; m++) {
            // (values, not array slots: the select between two elements of v otherwise becomes a load from a selected ADDRESS and the
            // whole array moves to scratch — whose loads and stores count in vmcnt)
            float ex = v[2 * m].x, ey = v[2 * m].y, ox = v[2 * m + 1].x, oy = v[2 * m + 1].y;
            asm("" : "+v"(ex), "+v"(ey), "+v"(ox), "+v"(oy));
            const float rx = lane_xor1(odd ? ex : ox), ry = lane_xor1(odd ? ey : oy);
            const float4 q = odd ? make_float4(rx, ry, ox, oy) : make_float4(ex, ey, rx, ry);
            *reinterpret_cast<float4 *>(to_[m]) = q;
          }
        } else {
#pragma unroll
          for (int k = SKIP; k < 16; k++) *to_[k] = v[k];
        }
        K7_STAMP(5, tid);
#ifdef K7_STAMPS
        st_turns++;
#endif
        continue;
      }
      const bool vec_out = ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) && (s & 1) == 0 && (a.HH & 1) == 0 && s >= FT &&
                           o0 + (L - a.HH) <= a.N;
      if (vec_out) {   // lane pairs swap halves and store 16 bytes per lane: 8 dwordx4 stores instead of 16 dwordx2
        const int odd = j & 1, je = j & ~1;
#pragma unroll
        for (int m = 0; m < 8; m++) {
          // even lane stores (E[2m], O[2m]), odd lane stores (E[2m+1], O[2m+1])
          const float2 give = odd ? v[2 * m] : v[2 * m + 1];
          const float rx = lane_xor1(give.x), ry = lane_xor1(give.y);
          const float4 q = odd ? make_float4(rx, ry, v[2 * m + 1].x, v[2 * m + 1].y) : make_float4(v[2 * m].x, v[2 * m].y, rx, ry);
          const int i = je + (2 * m + odd) * s;
#ifdef K7_PROBE_NOSTORE   // (ceiling probe, results missing: the stores sit behind a condition that is never true at run time)
          if (i >= a.HH && a.N < 0) {
#else
          if (i >= a.HH) {
#endif
            if (ACC) { const float4 t = *reinterpret_cast<const float4 *>(dst + i); *reinterpret_cast<float4 *>(dst + i) = make_float4(q.x + t.x, q.y + t.y, q.z + t.z, q.w + t.w); }
            else *reinterpret_cast<float4 *>(dst + i) = q;
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < 16; k++) {
          const int i = j + k * s - a.HH, o = o0 + i;
          if (i >= 0 && o < a.N) {
            float2 *po = outb + (long)c * a.out_stride + o;
            if (ACC) { const float2 t = *po; *po = make_float2(v[k].x + t.x, v[k].y + t.y); }
            else *po = v[k];
          }
        }
      }
    }
  }
  if (BANK && band + 1 < nb) __syncthreads();   // the work image is read out before the next band's middle pass overwrites it
  }
  // the channel's last block rolls the overlap history forward (hist_new <- the last HH samples of concat(hist, in); `hist`
  // is only read, by this launch's first blocks): no separate launch
  if (!PIPE && a.hist_new != nullptr && blk == (int)gridDim.x - 1) {   // (PIPE: behind the walk — loads and stores on one path only)
    for (int k = tid0; k < a.HL; k += FT) {
      const long qq = (long)a.N + k;
      a.hist_new[(long)c * a.HL + k] = qq < a.HL ? a.hist[(long)c * a.HL + qq] : a.in[(long)c * a.in_stride + (qq - a.HL)];
    }
  }
  if (PIPE && uk + uk_step < uk_end) {
    int nblk_, nc_;
    unit_of(uk + uk_step, nblk_, nc_);
    pass0_pipe(nblk_ * a.hop - a.HH - a.delay, true);
  }
  }   // (the PIPE walk)
  // (PIPE) the history roll BEHIND the walk (inside it, its loads and stores on one path only would set the loop's waits to zero):
  // every workgroup copies the tails of its share of the channels
  if (PIPE && a.hist_new != nullptr) {
    for (int cc = blockIdx.x; cc < a.nchan; cc += gridDim.x)
      for (int k = tid0; k < a.HL; k += FT) {
        const long qq = (long)a.N + k;
        a.hist_new[(long)cc * a.HL + k] = qq < a.HL ? a.hist[(long)cc * a.HL + qq] : a.in[(long)cc * a.in_stride + (qq - a.HL)];
      }
  }
#ifdef K7_STAMPS
  if (PIPE && tid0 == 0) {
    for (int q = 0; q < 8; q++) a.stamps[blockIdx.x * 16 + q] = st_acc[q];
    a.stamps[blockIdx.x * 16 + 8] = st_turns; a.stamps[blockIdx.x * 16 + 9] = st_t0; a.stamps[blockIdx.x * 16 + 10] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

__global__ void hist_roll_kernel(const float2 *in, long in_stride, const float2 *hist_old, float2 *hist_new, int HH, int N) {
  const int c = blockIdx.y;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < HH; k += gridDim.x * blockDim.x) {
    const long qq = (long)N + k;
    hist_new[(long)c * HH + k] = qq < HH ? hist_old[(long)c * HH + qq] : in[(long)c * in_stride + (qq - HH)];
  }
}

// test entry: plain batched DFT through the same passes
__global__ __launch_bounds__(FT) void fft_c2c_kernel(const FftDev p, const int *perm, int sign, const float2 *in, float2 *out) {
  extern __shared__ __attribute__((aligned(16))) float2 xl[];
  const int tid = threadIdx.x, L = p.L;
  const float2 *src = in + (long)blockIdx.x * L;
  float2 *dst = out + (long)blockIdx.x * L;
  if (sign < 0) {
    for (int i = tid; i < L; i += FT) xl[PAD(i)] = src[i];
    __syncthreads();
    fft_forward_dif(xl, p, tid);
    for (int i = tid; i < L; i += FT) dst[perm[i]] = xl[PAD(i)];   // position i holds frequency perm[i]
  } else {
    for (int i = tid; i < L; i += FT) xl[PAD(i)] = src[perm[i]];
    __syncthreads();
    fft_inverse_dit(xl, p, tid);
    for (int i = tid; i < L; i += FT) dst[i] = xl[PAD(i)];
  }
}

// FFTPlan<double> (reference src/fftplan_fftw3.hh:12-76: fftw_plan_dft_1d on complex<double>, unnormalised either way):
// one workgroup per transform, the block in LDS as double2, bit-reversed load + in-place radix-2 decimation in time,
// twiddles exp(-2 pi i k / L) from a table made on the host in long double. No BASELINE configuration runs it (the
// filter bank is the float plan); it is there so that a caller of FFT::exec<double> finds it — L <= 8192 (128 KB of LDS).
__global__ __launch_bounds__(FT) void fft_c2c_f64_kernel(int L, int lg, const double2 *W, int sign, const double2 *in, double2 *out) {
  extern __shared__ __attribute__((aligned(16))) double2 xd[];
  const int tid = threadIdx.x;
  const double2 *src = in + (long)blockIdx.x * L;
  double2 *dst = out + (long)blockIdx.x * L;
  for (int i = tid; i < L; i += FT) xd[__builtin_bitreverse32((unsigned)i) >> (32 - lg)] = src[i];
  __syncthreads();
  for (int half = 1, st = L / 2; half < L; half <<= 1, st >>= 1) {
    for (int b = tid; b < L / 2; b += FT) {
      const int j = b & (half - 1), base = ((b - j) << 1) + j;
      double2 w = W[j * st];
      if (sign > 0) w.y = -w.y;
      const double2 u = xd[base], v = xd[base + half];
      const double tr = __dsub_rn(__dmul_rn(v.x, w.x), __dmul_rn(v.y, w.y)), ti = __dadd_rn(__dmul_rn(v.x, w.y), __dmul_rn(v.y, w.x));
      xd[base] = make_double2(u.x + tr, u.y + ti);
      xd[base + half] = make_double2(u.x - tr, u.y - ti);
    }
    __syncthreads();
  }
  for (int i = tid; i < L; i += FT) dst[i] = xd[i];
}

struct FftPlan {
  int L = 0;
  FftDev dev{};
  DevBuf<float2> W, T;
  DevBuf<int> perm_d;
  std::vector<int> perm;   // position -> frequency index after the forward DIF

  void build(sdrhip_ctx *ctx, int L_) {
    SDRHIP_REQUIRE(L_ >= 4 && L_ <= 16384 && (L_ & (L_ - 1)) == 0, SDRHIP_E_UNSUPPORTED,
                   "FFT size %d: need a power of two in [4,16384]", L_);
    L = L_;
    int lg = 0; while ((1 << lg) < L) lg++;
    dev.L = L; dev.npass = 0;
    // radix-16 passes (one LDS round trip per 4 bits) first, then what is left of log2 L
    int left = lg;
    while (left >= 4) { dev.radix[dev.npass++] = 16; left -= 4; }
    if (left >= 2) { dev.radix[dev.npass++] = 4; left -= 2; }
    if (left) dev.radix[dev.npass++] = 2;
    std::vector<float2> w(L);
    for (int t = 0; t < L; t++) {
      const double ang = -2.0 * M_PI * (double)t / (double)L;
      w[t] = make_float2((float)std::cos(ang), (float)std::sin(ang));
    }
    W.alloc(L); W.upload(w.data(), L, ctx->stream);
    dev.W = W.p;
    {   // per-pass twiddle tables of the radix-16 passes (the last pass, stride 1, has none)
      std::vector<float2> tt;
      int n = L;
      for (int q = 0; q < dev.npass; q++) {
        const int r = dev.radix[q], s = n / r, tw = L / n;
        dev.toff[q] = (int)tt.size();
        if (r == 16 && s > 1)
          for (int k = 1; k < 16; k++)
            for (int j = 0; j < s; j++) {
              const double ang = -2.0 * M_PI * (double)(((long)j * tw * k) % L) / (double)L;
              tt.push_back(make_float2((float)std::cos(ang), (float)std::sin(ang)));
            }
        n = s;
      }
      if (tt.empty()) tt.push_back(make_float2(1.f, 0.f));
      T.alloc(tt.size()); T.upload(tt.data(), tt.size(), ctx->stream);
      dev.T = T.p;
    }
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
  size_t lds_bytes() const { return (size_t)(L + (L >> 6) * 4) * sizeof(float2); }
};

inline bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

template <class K>
void allow_big_lds(K kernel, size_t bytes) { allow_lds_max(kernel, bytes); }   // (once per kernel and device, to the hardware's maximum)

// The general plan (fftgen.hpp) behind the same handle: any FFT size made of the factors 2 ... 13 in complex<float>, and
// every size in complex<double> (FilterNode<double>, FFTPlan<double>). Same overlap-save evaluation, same state rules.
// what the handle holds for every plan other than the tuned power-of-two complex<float> one
struct ConvAny {
  bool f64 = false;
  virtual ~ConvAny() {}
  virtual void load_kernel(int band, const void *kernel) = 0;
  virtual void process_dev(const void *in_dev, size_t n_in, size_t in_stride, void *out_dev, size_t out_stride) = 0;
  virtual void process(const void *in_host, size_t n_in, size_t in_stride, void *out_host, size_t out_stride) = 0;
  virtual void reset() = 0;
  virtual const char *kernel_names() const = 0;
};

template <class T2>
struct GenConv : ConvAny {
  typedef typename fftgen::Real<T2>::type R;
  sdrhip_ctx *ctx = nullptr;
  int mode = 0, C = 1, B = 1, hop = 0, HH = 0, par = 0, n_taps = 0;
  size_t max_in = 0;
  fftgen::GenPlan<T2> plan;
  DevBuf<T2> Kp, hist[2], stage_in, stage_out;
  static constexpr size_t kMaxLds = 160 * 1024;

  void create(sdrhip_ctx *ctx_, int mode_, int L, const R *kernels, int n_taps_, int n_bands, int channels, size_t max_in_) {
    ctx = ctx_; mode = mode_; C = channels; B = n_bands; max_in = max_in_; f64 = sizeof(R) == 8;
    plan.build(ctx, L, (int)(128 * 1024 / sizeof(T2)));
    if (mode == SDRHIP_FFTCONV_OLA) {
      SDRHIP_REQUIRE(L % 2 == 0, SDRHIP_E_INVALID, "overlap-add mode: fft_size %d must be 2N", L);
      hop = L / 2; n_taps = L / 2;
    } else {
      SDRHIP_REQUIRE(n_taps_ >= 1 && n_taps_ <= L, SDRHIP_E_INVALID, "n_taps %d outside [1,%d]", n_taps_, L);
      hop = L - n_taps_ + 1; n_taps = n_taps_;
    }
    HH = L - hop;
    Kp.alloc((size_t)L * B);
    const size_t per_band = mode == SDRHIP_FFTCONV_OLA ? (size_t)2 * L : (size_t)2 * n_taps;   // reals per band in `kernels`
    for (int b = 0; b < B; b++) load_kernel(b, kernels + (size_t)b * per_band);
    for (int p = 0; p < 2; p++) { hist[p].alloc((size_t)C * std::max(1, HH)); hist[p].zero(ctx->stream); }
    SDRHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }
  void load_kernel(int band, const R *kernel) {
    const int L = plan.L;
    std::vector< std::complex<double> > spec(L, std::complex<double>(0, 0));
    if (mode == SDRHIP_FFTCONV_OLA) {
      for (int i = 0; i < L; i++) spec[i] = std::complex<double>(kernel[2 * i], kernel[2 * i + 1]);
    } else {
      for (int i = 0; i < n_taps; i++) spec[i] = std::complex<double>(kernel[2 * i], kernel[2 * i + 1]);
      fftgen::host_dft(spec, -1);
    }
    std::vector<T2> kp(L);
    for (int pos = 0; pos < L; pos++) {
      const std::complex<double> v = spec[plan.perm[pos]] / (double)L;
      kp[pos].x = (R)v.real(); kp[pos].y = (R)v.imag();
    }
    SDRHIP_CHECK_HIP(hipMemcpyAsync(Kp.p + (size_t)band * L, kp.data(), (size_t)L * sizeof(T2), hipMemcpyHostToDevice, ctx->stream));
    SDRHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }
  void launch(const T2 *in_dev, size_t N, size_t in_stride, T2 *out_dev, size_t out_stride, size_t out_band) {
    ctx->use();
    if (N == 0) return;
    fftgen::GenConvArgs<T2> a;
    a.fft = plan.dev; a.in = in_dev; a.in_stride = (long)in_stride; a.hist = hist[par].p; a.HH = HH;
    a.hist_new = HH > 0 ? hist[par ^ 1].p : nullptr;
    a.Kp = Kp.p; a.out = out_dev; a.out_stride = (long)out_stride; a.N = (int)N; a.hop = hop; a.nb = B; a.out_band = (long)out_band;
    a.two = (B > 1 && 2 * plan.lds_bytes() <= kMaxLds) ? 1 : 0;
    const size_t lds = plan.lds_bytes() * (a.two ? 2 : 1);
    allow_big_lds(fftgen::conv_kernel<T2>, lds);
    hipLaunchKernelGGL(fftgen::conv_kernel<T2>, dim3((unsigned)ceil_div(N, (size_t)hop), C), dim3(fftgen::GT), lds, ctx->stream, a);
    SDRHIP_CHECK_HIP(hipGetLastError());
    if (HH > 0) par ^= 1;
  }
  void process_dev(const R *in_dev, size_t n_in, size_t in_stride, R *out_dev, size_t out_stride) {
    SDRHIP_REQUIRE(n_in <= max_in, SDRHIP_E_SIZE, "n_in %zu > max_in %zu", n_in, max_in);
    if (n_in == 0) return;
    SDRHIP_REQUIRE(in_dev && out_dev, SDRHIP_E_INVALID, "NULL buffer");
    if (in_stride == 0) in_stride = n_in;
    if (out_stride == 0) out_stride = n_in;
    SDRHIP_REQUIRE(in_stride >= n_in && out_stride >= n_in, SDRHIP_E_SIZE, "stride smaller than n_in");
    require_disjoint(in_dev, in_stride, n_in, sizeof(T2), out_dev, out_stride, n_in, sizeof(T2), (size_t)C, (size_t)C * B);
    launch(reinterpret_cast<const T2 *>(in_dev), n_in, in_stride, reinterpret_cast<T2 *>(out_dev), out_stride, (size_t)C * out_stride);
  }
  void process(const R *in_host, size_t n_in, size_t in_stride, R *out_host, size_t out_stride) {
    SDRHIP_REQUIRE(n_in <= max_in, SDRHIP_E_SIZE, "n_in %zu > max_in %zu", n_in, max_in);
    if (n_in == 0) return;
    SDRHIP_REQUIRE(in_host && out_host, SDRHIP_E_INVALID, "NULL buffer");
    ctx->use();
    if (in_stride == 0) in_stride = n_in;
    if (out_stride == 0) out_stride = n_in;
    SDRHIP_REQUIRE(in_stride >= n_in && out_stride >= n_in, SDRHIP_E_SIZE, "stride smaller than n_in");
    if (!stage_in.p) { stage_in.alloc((size_t)C * max_in); stage_out.alloc((size_t)B * C * max_in); }
    const size_t eb = sizeof(T2);
    copy_h2d_rows(ctx, stage_in.p, n_in * eb, in_host, in_stride * eb, n_in * eb, C);
    launch(stage_in.p, n_in, n_in, stage_out.p, n_in, (size_t)C * n_in);
    copy_d2h_rows(ctx, out_host, out_stride * eb, stage_out.p, n_in * eb, n_in * eb, (size_t)B * C);
    SDRHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }
  void reset() override { ctx->use(); for (int p = 0; p < 2; p++) hist[p].zero(ctx->stream); }
  void load_kernel(int band, const void *kernel) override { load_kernel(band, static_cast<const R *>(kernel)); }
  void process_dev(const void *in_dev, size_t n_in, size_t in_stride, void *out_dev, size_t out_stride) override {
    process_dev(static_cast<const R *>(in_dev), n_in, in_stride, static_cast<R *>(out_dev), out_stride); }
  void process(const void *in_host, size_t n_in, size_t in_stride, void *out_host, size_t out_stride) override {
    process(static_cast<const R *>(in_host), n_in, in_stride, static_cast<R *>(out_host), out_stride); }
  const char *kernel_names() const override { return "conv_kernel"; }
};

// ---- the FFT filter for transforms that do NOT fit one workgroup's LDS, or whose size has a prime factor above 13 ------
// (FilterNode<float>(16384), (12000), (1009), FilterNode<double>(8192), ...: the reference plans any 2 x block_size,
// src/filternode.hh:236-245, src/fftplan_fftw3.hh:34-36). Same overlap-save evaluation and state rules as above, as passes
// over device memory around one AnyFft plan (fftany.hpp: four-step beyond the LDS, Bluestein's chirp transform for large
// primes): gather the blocks (history | input, zero beyond the call) -> forward transforms -> per band: spectrum product,
// backward transforms, scatter of each block's last `hop` results. Channels go through in groups that keep the two block
// images under kScratchBytes. Generality, not a BASELINE figure: every pass streams through HBM.
template <class T2>
__global__ void big_gather_kernel(const T2 *in, long in_stride, const T2 *hist, int HH, int N, int hop, long L, int nblk, T2 *X) {
  const long blk = blockIdx.y, c = blockIdx.z;
  const long first = blk * hop - HH;   // call-relative index of the block's first sample
  T2 *dst = X + (c * nblk + blk) * L;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (long)gridDim.x * blockDim.x) {
    const long rel = first + i;
    T2 v = fftgen::mk<T2>(0, 0);
    if (rel >= 0) { if (rel < N) v = in[c * in_stride + rel]; }
    else { const long h = HH + rel; if (h >= 0) v = hist[c * HH + h]; }
    dst[i] = v;
  }
}
template <class T2>
__global__ void big_mul_kernel(long L, const T2 *spec, const T2 *X, T2 *Y) {   // Y[b][k] = X[b][k] spec[k] (natural order, spec pre-scaled by 1 / L)
  const long b = blockIdx.y;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (long)gridDim.x * blockDim.x) Y[b * L + i] = fftgen::gmul(X[b * L + i], spec[i]);
}
template <class T2>
__global__ void big_scatter_kernel(const T2 *Y, long L, int HH, int hop, int N, int nblk, T2 *out, long out_stride) {
  const long blk = blockIdx.y, c = blockIdx.z;
  const T2 *src = Y + (c * nblk + blk) * L + HH;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < hop; i += (long)gridDim.x * blockDim.x) {
    const long o = blk * hop + i;
    if (o < N) out[c * out_stride + o] = src[i];
  }
}
template <class T2>
__global__ void big_hist_kernel(const T2 *in, long in_stride, const T2 *hist, T2 *hist_new, int HH, int N) {
  const long c = blockIdx.y;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < HH; k += (long)gridDim.x * blockDim.x) {
    const long qq = (long)N + k;
    hist_new[c * HH + k] = qq < HH ? hist[c * HH + qq] : in[c * in_stride + (qq - HH)];
  }
}

template <class T2>
struct BigConv : ConvAny {
  typedef typename fftgen::Real<T2>::type R;
  sdrhip_ctx *ctx = nullptr;
  int mode = 0, C = 1, B = 1, hop = 0, HH = 0, par = 0, n_taps = 0;
  long L = 0;
  size_t max_in = 0;
  fftany::AnyFft<T2> fft;
  DevBuf<T2> Kp, hist[2], X, Y, stage_in, stage_out;
  int group = 1;   // channels per pass
  static constexpr size_t kScratchBytes = (size_t)1 << 30;

  void create(sdrhip_ctx *ctx_, int mode_, int L_, const R *kernels, int n_taps_, int n_bands, int channels, size_t max_in_) {
    ctx = ctx_; mode = mode_; C = channels; B = n_bands; max_in = max_in_; L = L_; f64 = sizeof(R) == 8;
    SDRHIP_REQUIRE(L >= 2, SDRHIP_E_INVALID, "fft_size %ld", L);
    if (mode == SDRHIP_FFTCONV_OLA) {
      SDRHIP_REQUIRE(L % 2 == 0, SDRHIP_E_INVALID, "overlap-add mode: fft_size %ld must be 2N", L);
      hop = (int)(L / 2); n_taps = (int)(L / 2);
    } else {
      SDRHIP_REQUIRE(n_taps_ >= 1 && n_taps_ <= L, SDRHIP_E_INVALID, "n_taps %d outside [1,%ld]", n_taps_, L);
      hop = (int)(L - n_taps_ + 1); n_taps = n_taps_;
    }
    HH = (int)(L - hop);
    fft.build(ctx, L);
    const size_t nblk = ceil_div(max_in, (size_t)hop);
    if (fft.fuses()) {
      // four-step plans: gather, spectrum product and scatter ride in the transform's own passes (fftany.hpp, ConvFuse) —
      // the scratch is the bands' spectra (B x blocks) plus the plan's own temporary of the same size
      group = (int)std::max<size_t>(1, std::min<size_t>((size_t)C, kScratchBytes / (2 * (size_t)B * nblk * (size_t)L * sizeof(T2))));
      group = (int)std::min<size_t>((size_t)group, std::max<size_t>(1, 32768 / ((size_t)B * nblk)));   // (one launch's grid.y)
      SDRHIP_REQUIRE((size_t)B * nblk <= 32768, SDRHIP_E_UNSUPPORTED, "%zu blocks x %d bands per call exceed one pass: lower max_in", nblk, B);
      Y.alloc((size_t)B * group * nblk * L);
      fft.reserve((long)((size_t)B * group * nblk));
    } else {
      // (the gather / scatter kernels take one block per grid.y entry: a small awkward size with a long max_in must not run past it)
      SDRHIP_REQUIRE(nblk <= 65535, SDRHIP_E_UNSUPPORTED, "%zu blocks per call exceed one launch's grid: lower max_in", nblk);
      group = (int)std::max<size_t>(1, std::min<size_t>((size_t)C, kScratchBytes / (2 * nblk * (size_t)L * sizeof(T2))));
      group = std::min(group, 65535);
      X.alloc((size_t)group * nblk * L); Y.alloc((size_t)group * nblk * L);
      fft.reserve((long)((size_t)group * nblk));
    }
    Kp.alloc((size_t)L * B);
    const size_t per_band = mode == SDRHIP_FFTCONV_OLA ? (size_t)2 * L : (size_t)2 * n_taps;
    for (int b = 0; b < B; b++) load_kernel_t(b, kernels + (size_t)b * per_band);
    for (int p = 0; p < 2; p++) { hist[p].alloc((size_t)C * std::max(1, HH)); hist[p].zero(ctx->stream); }
    SDRHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }
  void load_kernel_t(int band, const R *kernel) {
    std::vector< std::complex<double> > spec((size_t)L, std::complex<double>(0, 0));
    if (mode == SDRHIP_FFTCONV_OLA) {
      for (long i = 0; i < L; i++) spec[i] = std::complex<double>(kernel[2 * i], kernel[2 * i + 1]);
    } else {
      for (int i = 0; i < n_taps; i++) spec[i] = std::complex<double>(kernel[2 * i], kernel[2 * i + 1]);
      fftgen::host_dft(spec, -1);
    }
    std::vector<T2> kp((size_t)L);
    for (long k = 0; k < L; k++) { const std::complex<double> v = spec[k] / (double)L; kp[k].x = (R)v.real(); kp[k].y = (R)v.imag(); }
    SDRHIP_CHECK_HIP(hipMemcpyAsync(Kp.p + (size_t)band * L, kp.data(), (size_t)L * sizeof(T2), hipMemcpyHostToDevice, ctx->stream));
    SDRHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }
  void load_kernel(int band, const void *kernel) override { load_kernel_t(band, static_cast<const R *>(kernel)); }
  void launch(const T2 *in_dev, size_t N, size_t in_stride, T2 *out_dev, size_t out_stride, size_t out_band) {
    ctx->use();
    if (N == 0) return;
    hipStream_t st = ctx->stream;
    const int nblk = (int)ceil_div(N, (size_t)hop);
    const unsigned gx = (unsigned)std::min<long>((L + 255) / 256, 1024);
    for (int c0 = 0; c0 < C; c0 += group) {
      const int cg = std::min(group, C - c0);
      const long batch = (long)cg * nblk;
      if (fft.fuses()) {
        fftany::ConvFuse<T2> f{};
        f.in = in_dev + (size_t)c0 * in_stride; f.in_stride = (long)in_stride; f.hist = hist[par].p + (size_t)c0 * HH; f.HH = HH; f.N = (int)N; f.hop = hop; f.nblk = nblk;
        f.Kp = Kp.p; f.nb = B; f.band_elems = batch * L;
        f.out = out_dev + (size_t)c0 * out_stride; f.out_stride = (long)out_stride; f.out_band = (long)out_band; f.cg = cg;
        fft.conv_forward(f, batch, Y.p);
        fft.conv_inverse(f, (long)B * batch, Y.p);
        continue;
      }
      hipLaunchKernelGGL(big_gather_kernel<T2>, dim3(gx, nblk, cg), dim3(256), 0, st, in_dev + (size_t)c0 * in_stride, (long)in_stride,
                         hist[par].p + (size_t)c0 * HH, HH, (int)N, hop, L, nblk, X.p);
      fft.exec(-1, batch, X.p, X.p);
      for (int band = 0; band < B; band++) {
        for (long z0 = 0; z0 < batch; z0 += 32768) {
          const long zb = std::min<long>(32768, batch - z0);
          hipLaunchKernelGGL(big_mul_kernel<T2>, dim3(gx, (unsigned)zb), dim3(256), 0, st, L, Kp.p + (size_t)band * L, X.p + z0 * L, Y.p + z0 * L);
        }
        fft.exec(+1, batch, Y.p, Y.p);
        hipLaunchKernelGGL(big_scatter_kernel<T2>, dim3((unsigned)std::min<long>((hop + 255) / 256, 1024), nblk, cg), dim3(256), 0, st, Y.p, L, HH, hop, (int)N, nblk,
                           out_dev + (size_t)band * out_band + (size_t)c0 * out_stride, (long)out_stride);
      }
    }
    if (HH > 0) {
      hipLaunchKernelGGL(big_hist_kernel<T2>, dim3((unsigned)std::min<long>((HH + 255) / 256, 1024), C), dim3(256), 0, st, in_dev, (long)in_stride,
                         hist[par].p, hist[par ^ 1].p, HH, (int)N);
      par ^= 1;
    }
    SDRHIP_CHECK_HIP(hipGetLastError());
  }
  void process_dev(const void *in_dev, size_t n_in, size_t in_stride, void *out_dev, size_t out_stride) override {
    SDRHIP_REQUIRE(n_in <= max_in, SDRHIP_E_SIZE, "n_in %zu > max_in %zu", n_in, max_in);
    if (n_in == 0) return;
    SDRHIP_REQUIRE(in_dev && out_dev, SDRHIP_E_INVALID, "NULL buffer");
    if (in_stride == 0) in_stride = n_in;
    if (out_stride == 0) out_stride = n_in;
    SDRHIP_REQUIRE(in_stride >= n_in && out_stride >= n_in, SDRHIP_E_SIZE, "stride smaller than n_in");
    require_disjoint(in_dev, in_stride, n_in, sizeof(T2), out_dev, out_stride, n_in, sizeof(T2), (size_t)C, (size_t)C * B);
    launch(static_cast<const T2 *>(in_dev), n_in, in_stride, static_cast<T2 *>(out_dev), out_stride, (size_t)C * out_stride);
  }
  void process(const void *in_host, size_t n_in, size_t in_stride, void *out_host, size_t out_stride) override {
    SDRHIP_REQUIRE(n_in <= max_in, SDRHIP_E_SIZE, "n_in %zu > max_in %zu", n_in, max_in);
    if (n_in == 0) return;
    SDRHIP_REQUIRE(in_host && out_host, SDRHIP_E_INVALID, "NULL buffer");
    ctx->use();
    if (in_stride == 0) in_stride = n_in;
    if (out_stride == 0) out_stride = n_in;
    SDRHIP_REQUIRE(in_stride >= n_in && out_stride >= n_in, SDRHIP_E_SIZE, "stride smaller than n_in");
    if (!stage_in.p) { stage_in.alloc((size_t)C * max_in); stage_out.alloc((size_t)B * C * max_in); }
    const size_t eb = sizeof(T2);
    copy_h2d_rows(ctx, stage_in.p, n_in * eb, in_host, in_stride * eb, n_in * eb, C);
    launch(stage_in.p, n_in, n_in, stage_out.p, n_in, (size_t)C * n_in);
    copy_d2h_rows(ctx, out_host, out_stride * eb, stage_out.p, n_in * eb, n_in * eb, (size_t)B * C);
    SDRHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }
  void reset() override { ctx->use(); for (int p = 0; p < 2; p++) hist[p].zero(ctx->stream); }
  const char *kernel_names() const override { return fft.fuses() ? "fourstep_tile_kernel x4 (gather, product, scatter fused)" : "big_gather_kernel,fft passes,big_mul_kernel,big_scatter_kernel"; }
};

}  // namespace

struct sdrhip_fftconv {
  sdrhip_ctx *ctx = nullptr;
  std::unique_ptr<ConvAny> any;   // every plan but the tuned power-of-two complex<float> one: GenConv (one transform in one workgroup's
                                  // LDS, factors 2 ... 13) or BigConv (any size: four-step / chirp passes), float or double
  int mode = 0, C = 1, hop = 0, HH = 0, par = 0;
  int ola_L = 0;        // an overlap-add plan (2N-point spectra) that runs as overlap-save on another transform: the caller's 2N (set_kernel converts)
  int B = 1;            // bands of the bank (spectra sharing one forward transform)
  int n_taps = 0;
  // Partitioned overlap-save (12290 ... 16384 taps: FilterNode<float>(N) with 12289 < N <= 16384 — no single 16384-point block can
  // carry them and a 32768-point block does not fit a workgroup's LDS): the taps in `parts` partitions of `part_taps`, every block
  // convolved with each partition on the tuned 16384-point kernel — partition p on the window p * part_taps samples earlier,
  // accumulated into the output. The history rows hold HL = HH + (parts - 1) * part_taps samples.
  int parts = 1, part_taps = 0, HL = 0;
  size_t max_in = 0;
  FftPlan plan;
  DevBuf<float2> Kp;
  DevBuf<float2> hist[2];
  DevBuf<float2> stage_in, stage_out;
  int stamps_grid = 0;   // (diagnostic builds)
  DevBuf<float2> dump;   // (the pipelined 16384-point form) 128 bytes per wave of its grid for the stores of masked lanes

  // spectrum of one band -> the device layout (digit-reversed position order, pre-scaled by 1/L)
  void load_kernel(int band, const float *kernel) {
    for (int part = 0; part < parts; part++) load_kernel_part(band, part, kernel);
  }
  void load_kernel_part(int band, int part, const float *kernel) {
    const int L = plan.L;
    std::vector< std::complex<double> > spec(L);
    if (mode == SDRHIP_FFTCONV_OLA) {
      // kernel = the FilterSource spectrum (2N points); its time-domain support is N taps
      for (int i = 0; i < L; i++) spec[i] = std::complex<double>(kernel[2 * i], kernel[2 * i + 1]);
    } else {
      // host DFT in double of the zero-padded taps (one-off); partitioned plans: of partition `part`'s taps
      std::vector< std::complex<double> > a(L, std::complex<double>(0, 0));
      const int t0 = parts > 1 ? part * part_taps : 0, t1 = parts > 1 ? std::min(n_taps, t0 + part_taps) : n_taps;
      for (int i = t0; i < t1; i++) a[i - t0] = std::complex<double>(kernel[2 * i], kernel[2 * i + 1]);
      for (size_t i = 1, j = 0; i < (size_t)L; i++) {
        size_t bit = (size_t)L >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
      }
      for (size_t len = 2; len <= (size_t)L; len <<= 1)
        for (size_t k = 0; k < len / 2; k++) {
          const double ang = -2.0 * M_PI * (double)k / (double)len;
          const std::complex<double> w(std::cos(ang), std::sin(ang));
          for (size_t s = 0; s < (size_t)L; s += len) {
            const std::complex<double> u = a[s + k], t = w * a[s + k + len / 2];
            a[s + k] = u + t; a[s + k + len / 2] = u - t;
          }
        }
      spec = a;
    }
    std::vector<float2> kp(L);
    for (int pos = 0; pos < L; pos++) {
      const std::complex<double> v = spec[plan.perm[pos]] / (double)L;
      kp[pos] = make_float2((float)v.real(), (float)v.imag());
    }
    SDRHIP_CHECK_HIP(hipMemcpyAsync(Kp.p + ((size_t)band * parts + part) * L, kp.data(), (size_t)L * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
    SDRHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }

  // workgroups of the pipelined 16384-point form: one per CU (SDRHIP_K7_PIPE_GRID=n: tests walk small calls with few workgroups;
  // 0: the one-block-per-workgroup kernel, the A/B hook). Read per launch: a call costs microseconds, getenv nanoseconds
  static constexpr int kPipeGridMax = 1024;   // (the dump lines are allocated with the handle: nothing is allocated in a launch)
  int pipe_grid() const {
    const char *e = getenv("SDRHIP_K7_PIPE_GRID");
    return std::min(e ? atoi(e) : ctx->prop.multiProcessorCount, kPipeGridMax);
  }

  // bands a launch can serve from one forward transform: the bank kernel keeps the spectrum in registers, 16 values per
  // lane of an L/16-lane workgroup (L = 1024 .. 8192; 16384 points sit at the register cap of a 1024-lane workgroup)
  int bands_per_launch() const { return (plan.L >= 1024 && plan.L <= 8192) ? B : 1; }

  // out_band: elements between band b's and band b+1's rows
  void launch(const float2 *in_dev, size_t N, size_t in_stride, float2 *out_dev, size_t out_stride, size_t out_band) {
    ctx->use();
    if (N == 0) return;
    const int bpl = bands_per_launch();
    bool rolled = false;
    for (int b0 = 0; b0 < B; b0 += bpl) {   // (a plan too large for two LDS images transforms the input once per band)
    for (int part = 0; part < parts; part++) {
    ConvArgs a;
    a.nblk = 0; a.nchan = 0; a.dump = nullptr;
    a.nb = std::min(bpl, B - b0); a.out_band = (long)out_band; a.lds_elems = (int)(plan.lds_bytes() / sizeof(float2));
    const size_t lds = plan.lds_bytes();
    a.fft = plan.dev; a.in = in_dev; a.in_stride = (long)in_stride;
    a.hist = hist[par].p; a.HH = HH; a.HL = HL; a.delay = part * part_taps; a.Kp = Kp.p + ((size_t)b0 * parts + part) * plan.L;
    a.hist_new = nullptr;
    a.out = out_dev + (size_t)b0 * out_band; a.out_stride = (long)out_stride; a.N = (int)N; a.hop = hop;
    const int blocks = (int)ceil_div(N, (size_t)hop);
    auto fused = [&](auto kernel, int nt) {
      if (HL > 0 && b0 + bpl >= B && part + 1 == parts) { a.hist_new = hist[par ^ 1].p; rolled = true; }   // the call's last launch
      allow_big_lds(kernel, lds);
      hipLaunchKernelGGL(kernel, dim3(blocks, C), dim3(nt), lds, ctx->stream, a);
    };
    const bool fusable = plan.dev.npass >= 2 && plan.dev.radix[0] == 16;
    int nt = plan.L / 16 >= 1024 ? 1024 : plan.L / 16 >= 512 ? 512 : plan.L / 16 >= 256 ? 256 : plan.L / 16 >= 128 ? 128 : 64;
    { const char *e = getenv("SDRHIP_K7_NT"); if (e && a.nb == 1) nt = atoi(e); }   // tuning hook (the bank kernel needs L / 16 lanes)
#define SDRHIP_FUSED(BANK_) do { switch (nt) { \
      case 1024: fused(fftconv_fused_kernel<0, BANK_, 1024>, 1024); break; \
      case 512: fused(fftconv_fused_kernel<0, BANK_, 512>, 512); break; \
      case 256: fused(fftconv_fused_kernel<0, BANK_, 256>, 256); break; \
      case 128: fused(fftconv_fused_kernel<0, BANK_, 128>, 128); break; \
      default: fused(fftconv_fused_kernel<0, BANK_, 64>, 64); break; } } while (0)
    // compile-time plans (strides, pad offsets and butterfly maps fold into immediates and shifts; the run-time-plan
    // kernel divides by the pass stride per butterfly): 16384 points, and 2048 / 4096 / 8192 with L / 16 lanes
    auto plan_is = [&](int lg) {
      if (plan.L != (1 << lg) || plan.dev.npass != plan_npass(lg)) return false;
      for (int q = 0; q < plan.dev.npass; q++) if (plan.dev.radix[q] != plan_radix(lg, q)) return false;
      return true;
    };
    const bool ct = getenv("SDRHIP_K7_RUNTIME_PLAN") == nullptr;   // (tuning / tests: the run-time-plan kernel for every size)
    if (plan.L == 16384 && part > 0) {
      fused(fftconv_fused_kernel<14, false, 1024, true>, 1024);   // (a later tap partition: accumulated)
    } else if (plan.L == 16384 && pipe_grid() > 0 && (long)blocks * C > pipe_grid()) {
      // the pipelined form: one persistent workgroup per CU (more units than CUs: otherwise there is no next block to fetch)
      if (HL > 0 && b0 + bpl >= B && part + 1 == parts) { a.hist_new = hist[par ^ 1].p; rolled = true; }   // the call's last launch
      const int grid = pipe_grid();
      a.nblk = blocks; a.nchan = C;
      a.dump = dump.p;
#ifdef K7_STAMPS
      a.stamps = reinterpret_cast<unsigned long long *>(dump.p + (size_t)grid * 16 * 16); stamps_grid = grid;
#endif
      auto even = [](size_t v) { return (v & 1) == 0; };
      const bool pv = even(hop) && even(HH) && even(HL) && even(a.delay) && even(N) && even(in_stride) && even(out_stride) &&
                      (reinterpret_cast<uintptr_t>(in_dev) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 &&
                      getenv("SDRHIP_K7_PIPE_X2") == nullptr;   // (A/B hook: the 8-byte form everywhere)
      const size_t lds_p = lds + (15 * 64 + 15 * 4) * sizeof(float2);   // (+ room for the K7_LDS_TW variant's tables)
      auto go = [&](auto kernel) {
        allow_big_lds(kernel, lds_p);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(1024), lds_p, ctx->stream, a);
      };
      // (the two BASELINE shapes get their stores counted at compile time: 4097 taps keep 12288 of 16384, the reference mode's
      // 8192 taps 8192)
      if (pv && HH == 4096) go(fftconv_fused_kernel<14, false, 1024, false, 4, 2>);
      else if (pv && HH == 8192) go(fftconv_fused_kernel<14, false, 1024, false, 4, 4>);
      else if (pv) go(fftconv_fused_kernel<14, false, 1024, false, 4, 0>);
      else if (HH == 4096) go(fftconv_fused_kernel<14, false, 1024, false, 2, 4>);
      else if (HH == 8192) go(fftconv_fused_kernel<14, false, 1024, false, 2, 8>);
      else go(fftconv_fused_kernel<14, false, 1024, false, 2, 0>);
    } else if (plan.L == 16384) {   // (never a bank: two images of 16384 points do not fit the LDS)
      fused(fftconv_fused_kernel<14, false, 1024>, 1024);   // (512 / 256 lanes measured 0.78x / 0.59x)
    } else if (ct && fusable && nt == 128 && plan_is(11)) {
      if (a.nb > 1) fused(fftconv_fused_kernel<11, true, 128>, 128); else fused(fftconv_fused_kernel<11, false, 128>, 128);
    } else if (ct && fusable && nt == 256 && plan_is(12)) {
      if (a.nb > 1) fused(fftconv_fused_kernel<12, true, 256>, 256); else fused(fftconv_fused_kernel<12, false, 256>, 256);
    } else if (ct && fusable && nt == 512 && plan_is(13)) {
      if (a.nb > 1) fused(fftconv_fused_kernel<13, true, 512>, 512); else fused(fftconv_fused_kernel<13, false, 512>, 512);
    } else if (fusable && a.nb > 1) {
      SDRHIP_FUSED(true);
    } else if (fusable) {
      SDRHIP_FUSED(false);
#undef SDRHIP_FUSED
    } else {
      allow_big_lds(fftconv_kernel, lds);
      hipLaunchKernelGGL(fftconv_kernel, dim3(blocks, C), dim3(FT), lds, ctx->stream, a);
    }
    }
    }
    SDRHIP_CHECK_HIP(hipGetLastError());
    if (HL > 0 && rolled) par ^= 1;
    else if (HL > 0) {
      hipLaunchKernelGGL(hist_roll_kernel, dim3((unsigned)ceil_div((size_t)HL, (size_t)256), C), dim3(256), 0, ctx->stream,
                         in_dev, (long)in_stride, hist[par].p, hist[par ^ 1].p, HL, (int)N);
      SDRHIP_CHECK_HIP(hipGetLastError());
      par ^= 1;
    }
  }
};

// ---- FFTPlan<float|double>: planned once, executed many times (reference src/fftplan_fftw3.hh:14-36,82-104: the
// plan is made in the constructor, operator() only executes) -------------------------------------------------------
struct sdrhip_fft_plan {
  sdrhip_ctx *ctx = nullptr;
  int dtype = SDRHIP_T_CF32, n = 0;
  std::unique_ptr<FftPlan> p32;                            // complex<float>, a power of two in [4, 16384]: the tuned radix-16 kernel
  DevBuf<double2> W64; int lg64 = 0;                       // complex<double>, a power of two in [2, 8192]
  std::unique_ptr< fftany::AnyFft<float2> > a32;           // any other size (fftany.hpp)
  std::unique_ptr< fftany::AnyFft<double2> > a64;
  DevBuf<char> stage_in, stage_out;                        // exec() on host buffers
  size_t elem() const { return dtype == SDRHIP_T_CF64 ? 16 : 8; }
  const char *form() const {
    return p32 ? "radix-16 lds" : W64.p ? "radix-2 lds (double)" : a32 ? a32->kind_name() : a64 ? a64->kind_name() : "?";
  }
  void build(sdrhip_ctx *ctx_, int dtype_, int n_) {
    ctx = ctx_; dtype = dtype_; n = n_;
    SDRHIP_REQUIRE(n >= 1, SDRHIP_E_INVALID, "FFT size %d", n);
    ctx->use();
    if (dtype == SDRHIP_T_CF32) {
      if (is_pow2(n) && n >= 4 && n <= 16384) {
        p32.reset(new FftPlan()); p32->build(ctx, n);
        allow_big_lds(fft_c2c_kernel, p32->lds_bytes());
      } else { a32.reset(new fftany::AnyFft<float2>()); a32->build(ctx, n); }
    } else {
      if (is_pow2(n) && n >= 2 && n <= 8192) {
        while ((1 << lg64) < n) lg64++;
        std::vector<double2> w(n / 2);
        for (int k = 0; k < n / 2; k++) {
          const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)n;
          w[k] = make_double2((double)cosl(ang), (double)sinl(ang));
        }
        W64.alloc(n / 2); W64.upload(w.data(), n / 2, ctx->stream);
        allow_big_lds(fft_c2c_f64_kernel, (size_t)n * sizeof(double2));
      } else { a64.reset(new fftany::AnyFft<double2>()); a64->build(ctx, n); }
    }
  }
  // asynchronous on the context's stream; in == out allowed
  void exec_dev(int sign, int batch, const void *in_dev, void *out_dev) {
    ctx->use();
    if (p32) {
      hipLaunchKernelGGL(fft_c2c_kernel, dim3(batch), dim3(FT), p32->lds_bytes(), ctx->stream, p32->dev, p32->perm_d.p, sign,
                         reinterpret_cast<const float2 *>(in_dev), reinterpret_cast<float2 *>(out_dev));
    } else if (W64.p) {
      hipLaunchKernelGGL(fft_c2c_f64_kernel, dim3(batch), dim3(FT), (size_t)n * sizeof(double2), ctx->stream, n, lg64, W64.p, sign,
                         reinterpret_cast<const double2 *>(in_dev), reinterpret_cast<double2 *>(out_dev));
    } else if (a32) {
      a32->exec(sign, batch, reinterpret_cast<const float2 *>(in_dev), reinterpret_cast<float2 *>(out_dev));
    } else {
      a64->exec(sign, batch, reinterpret_cast<const double2 *>(in_dev), reinterpret_cast<double2 *>(out_dev));
    }
    SDRHIP_CHECK_HIP(hipGetLastError());
  }
};

namespace {
// the one-shot entry points (sdrhip_fft_c2c / _f64 / sdrhip_fft_exec) keep their plans in the context: (dtype, n) -> plan,
// freed with the context
sdrhip_fft_plan *cached_plan(sdrhip_ctx *ctx, int dtype, int n) {
  const long long key = ((long long)dtype << 40) | (long long)(unsigned)n;
  auto it = ctx->cache.find(key);
  if (it != ctx->cache.end()) return static_cast<sdrhip_fft_plan *>(it->second.get());
  std::unique_ptr<sdrhip_fft_plan> p(new sdrhip_fft_plan());
  p->build(ctx, dtype, n);
  if (ctx->cache.size() >= 64) ctx->cache.clear();   // (a sweep over many sizes: start over rather than hoard tables)
  sdrhip_fft_plan *raw = p.release();
  ctx->cache[key] = std::shared_ptr<void>(raw, [](void *q) { delete static_cast<sdrhip_fft_plan *>(q); });
  return raw;
}
}  // namespace

namespace {
// FilterNode's block size is the granularity of ITS buffers, not a property of the result: an overlap-add filter with a
// 2N-point spectrum of support N is the N-tap convolution y[n] = sum_k h[k] x[n - k], whatever transform evaluates it. A block
// size whose 2N-point transform is awkward (not a power of two: generic radix passes, four-step or chirp plans) is therefore run
// as OVERLAP-SAVE with the same N taps on the power-of-two transform that costs least per output — the tuned kernels in
// complex<float>. The taps are the first N points of the inverse DFT of the spectrum (host, double, one-off per kernel).
template <class R>
std::vector<R> ola_spectrum_to_taps(const R *K, int L) {
  std::vector< std::complex<double> > spec(L);
  for (int i = 0; i < L; i++) spec[i] = std::complex<double>(K[2 * i], K[2 * i + 1]);
  fftgen::host_dft(spec, +1);
  std::vector<R> taps((size_t)L);   // N = L / 2 complex taps
  for (int i = 0; i < L / 2; i++) { taps[2 * i] = (R)(spec[i].real() / L); taps[2 * i + 1] = (R)(spec[i].imag() / L); }
  return taps;
}
// the power of two (4 ... max_l) on which an N-tap overlap-save filter costs the fewest butterfly operations per output, provided
// a block keeps at least a quarter of its points; 0: none (N too long for max_l)
int overlap_save_fft_size(int N, int max_l) {
  int best = 0;
  double best_cost = 0;
  for (int l = 4, lg = 2; l <= max_l; l <<= 1, lg++) {
    const int hop = l - N + 1;
    if (hop < 1 || 4 * hop < l) continue;
    const double cost = (double)l * lg / hop;
    if (!best || cost < best_cost) { best = l; best_cost = cost; }
  }
  return best;
}

// GenConv where one transform fits one workgroup's LDS and has no prime factor above 13, BigConv for every other size
template <class T2, class R>
ConvAny *make_any_conv(sdrhip_ctx *ctx, int mode, int fft_size, const R *kernels, int n_taps, int n_bands, int channels, size_t max_in) {
  std::vector<int> rx;
  const long maxL = (long)(128 * 1024 / sizeof(T2));
  if (fft_size >= 1 && fft_size <= maxL && fftgen::GenPlan<T2>::factor(fft_size, rx, nullptr)) {
    std::unique_ptr< GenConv<T2> > g(new GenConv<T2>());
    g->create(ctx, mode, fft_size, kernels, n_taps, n_bands, channels, max_in);
    return g.release();
  }
  std::unique_ptr< BigConv<T2> > b(new BigConv<T2>());
  b->create(ctx, mode, fft_size, kernels, n_taps, n_bands, channels, max_in);
  return b.release();
}
}  // namespace

extern "C" {

int sdrhip_fftconv_create_bank(sdrhip_ctx *ctx, int mode, int fft_size, const float *kernels, int n_taps, int n_bands,
                               int channels, size_t max_in, sdrhip_fftconv **out) {
  return guarded([&] {
    SDRHIP_REQUIRE(ctx && kernels && out, SDRHIP_E_INVALID, "NULL argument");
    *out = nullptr;
    SDRHIP_REQUIRE(mode == SDRHIP_FFTCONV_OLA || mode == SDRHIP_FFTCONV_OLS, SDRHIP_E_INVALID, "bad mode %d", mode);
    SDRHIP_REQUIRE(channels >= 1 && channels <= 65535, SDRHIP_E_INVALID, "channels %d outside [1,65535]", channels);
    SDRHIP_REQUIRE(n_bands >= 1 && n_bands <= 256, SDRHIP_E_INVALID, "n_bands %d outside [1,256]", n_bands);
    SDRHIP_REQUIRE(max_in >= 1 && max_in < (size_t(1) << 30), SDRHIP_E_SIZE, "max_in %zu outside [1,2^30)", max_in);
    ctx->use();
    sdrhip_fftconv *h = new sdrhip_fftconv;
    try {
      h->ctx = ctx; h->mode = mode; h->C = channels; h->max_in = max_in; h->B = n_bands;
      std::vector<float> taps_all;
      // (a single band on a power of two as well: half of every 2N-point block is overlap, a longer block keeps up to 7/8 —
      // N = 256: 13.9 -> 50.4 % of the roofline, 512: 28 -> 43 %, 2048: 29.5 -> 33 %, 4096: 21.6 -> 28 % at 256 channels. Not the
      // 2048-point plan (its last passes run in registers: 35.6 % against 33.8 % remapped) and not the banks (one forward transform
      // per block for all bands, tuned at these sizes: 32 % against 26 %). profiles/r17_bigconv_time.txt)
      const bool awkward = !is_pow2(fft_size) || fft_size < 4 || fft_size > 16384;
      // (round 6: the 2048-point plan too where the call has blocks enough for the pipelined 16384-point kernel — FilterNode(1024) on
      // 256 channels x 65536: 0.093 -> 0.078 ms)
      const bool keep2048 = fft_size == 2048 && ols_fft_size(1024, (size_t)channels, max_in, ctx->prop.multiProcessorCount) != 16384;
      if (mode == SDRHIP_FFTCONV_OLA && (awkward || (n_bands == 1 && !keep2048)) && fft_size % 2 == 0 && !getenv("SDRHIP_FFTCONV_LITERAL")) {
        // the same N taps by overlap-save on the best power of two (ola_spectrum_to_taps); SDRHIP_FFTCONV_LITERAL=1
        // keeps the 2N-point transform (tests of the general plans)
        // (complex<float>: the tuned kernels' own ranking, ols_fft_size — the operation count alone would pick 8192 points for
        // 1000 taps: 0.39 ms against 0.34 on 16384 and 0.35 on 4096)
        // (12290 ... 16384 taps, one band: the same 16384-point kernel in two tap partitions — sdrhip_fftconv::parts)
        // (banks keep the ranking measured for them: one forward transform serves all bands up to 8192 points, not at 16384)
        const int N = fft_size / 2, lp = N <= 12289 ? (n_bands == 1 ? ols_fft_size(N, (size_t)channels, max_in, ctx->prop.multiProcessorCount)
                                                                     : N <= 512 ? 2048 : N <= 2048 ? 4096 : 16384)
                                           : (N <= 16384 && n_bands == 1 && !getenv("SDRHIP_FFTCONV_NO_PARTS")) ? 16384 : 0;
        if (lp && N > 12289) { h->parts = 2; h->part_taps = 8192; }
        if (lp && lp != fft_size) {
          for (int b = 0; b < n_bands; b++) {
            const std::vector<float> t = ola_spectrum_to_taps(kernels + (size_t)b * 2 * fft_size, fft_size);
            taps_all.insert(taps_all.end(), t.begin(), t.begin() + 2 * N);
          }
          h->ola_L = fft_size;
          mode = SDRHIP_FFTCONV_OLS; h->mode = mode; fft_size = lp; n_taps = N; kernels = taps_all.data();
        }
      }
      if (!is_pow2(fft_size) || fft_size < 4 || fft_size > 16384) {   // (FilterNode(size_t block_size) takes any block size: src/filternode.hh:235-245)
        h->any.reset(make_any_conv<float2, float>(ctx, mode, fft_size, kernels, n_taps, n_bands, channels, max_in));
        *out = h;
        return;
      }
      h->plan.build(ctx, fft_size);
      const int L = fft_size;
      if (mode == SDRHIP_FFTCONV_OLA) { h->hop = L / 2; h->n_taps = L / 2; }
      else if (h->parts > 1) { h->hop = L - h->part_taps; h->n_taps = n_taps; }   // (a partition's taps: part_taps <= L - hop + 1)
      else {
        SDRHIP_REQUIRE(n_taps >= 1 && n_taps <= L, SDRHIP_E_INVALID, "n_taps %d outside [1,%d]", n_taps, L);
        h->hop = L - n_taps + 1; h->n_taps = n_taps;
        // an even hop (one more sample of history than the taps need) keeps every block of an aligned call 16-byte aligned: the
        // lane-pair loads and stores of the first and last pass apply, and the pipelined form's 16-byte variant. The reference
        // mode's 8192 taps on 16384 points: hop 8193 -> 8192, the same 8 blocks per 65536 samples. (SDRHIP_FFTCONV_ODD_HOP=1
        // keeps L - n_taps + 1: tests of the unaligned paths.)
        if ((h->hop & 1) && h->hop > 1 && !getenv("SDRHIP_FFTCONV_ODD_HOP")) h->hop -= 1;
      }
      h->HH = L - h->hop;
      h->HL = h->HH + (h->parts - 1) * h->part_taps;
      h->Kp.alloc((size_t)L * n_bands * h->parts);
      if (L == 16384) h->dump.alloc((size_t)sdrhip_fftconv::kPipeGridMax * (16 * 16 + 16));   // (+ 16 words per workgroup: diagnostic builds' stamps)
      const size_t per_band = mode == SDRHIP_FFTCONV_OLA ? (size_t)2 * L : (size_t)2 * n_taps;   // floats per band in `kernels`
      for (int b = 0; b < n_bands; b++) h->load_kernel(b, kernels + (size_t)b * per_band);
      for (int p = 0; p < 2; p++) { h->hist[p].alloc((size_t)channels * std::max(1, h->HL)); h->hist[p].zero(ctx->stream); }
      SDRHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    } catch (...) { delete h; throw; }
    *out = h;
  });
}

int sdrhip_fftconv_create(sdrhip_ctx *ctx, int mode, int fft_size, const float *kernel, int n_taps, int channels,
                          size_t max_in, sdrhip_fftconv **out) {
  return sdrhip_fftconv_create_bank(ctx, mode, fft_size, kernel, n_taps, 1, channels, max_in, out);
}

int sdrhip_fftconv_bands(sdrhip_fftconv *h, int *n_bands) {
  return guarded([&] {
    SDRHIP_REQUIRE(h && n_bands, SDRHIP_E_INVALID, "NULL argument");
    *n_bands = h->B;
  });
}

int sdrhip_fftconv_set_kernel(sdrhip_fftconv *h, int band, const float *kernel) {
  return guarded([&] {
    SDRHIP_REQUIRE(h && kernel, SDRHIP_E_INVALID, "NULL argument");
    SDRHIP_REQUIRE(band >= 0 && band < h->B, SDRHIP_E_INVALID, "band %d outside [0,%d)", band, h->B);
    SDRHIP_REQUIRE(!(h->any && h->any->f64), SDRHIP_E_INVALID, "a complex<double> plan: use sdrhip_fftconv_f64_set_kernel");
    h->ctx->use();
    SDRHIP_CHECK_HIP(hipStreamSynchronize(h->ctx->stream));   // launches in flight still read the old spectrum
    std::vector<float> taps;
    if (h->ola_L) { taps = ola_spectrum_to_taps(kernel, h->ola_L); kernel = taps.data(); }   // (the plan runs as overlap-save: see create)
    if (h->any) { h->any->load_kernel(band, kernel); return; }
    h->load_kernel(band, kernel);
  });
}

int sdrhip_fftconv_process_dev(sdrhip_fftconv *h, const float *in_dev, size_t n_in, size_t in_stride,
                               float *out_dev, size_t out_stride) {
  return guarded([&] {
    Range roctx_range("sdrhip_fftconv_process_dev");
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    SDRHIP_REQUIRE(!(h->any && h->any->f64), SDRHIP_E_INVALID, "a complex<double> plan: use sdrhip_fftconv_f64_process_dev");
    if (h->any) { h->any->process_dev(in_dev, n_in, in_stride, out_dev, out_stride); return; }
    SDRHIP_REQUIRE(n_in <= h->max_in, SDRHIP_E_SIZE, "n_in %zu > max_in %zu", n_in, h->max_in);
    if (n_in == 0) return;
    SDRHIP_REQUIRE(in_dev && out_dev, SDRHIP_E_INVALID, "NULL buffer");
    if (in_stride == 0) in_stride = n_in;
    if (out_stride == 0) out_stride = n_in;
    SDRHIP_REQUIRE(in_stride >= n_in && out_stride >= n_in, SDRHIP_E_SIZE, "stride smaller than n_in");
    require_disjoint(in_dev, in_stride, n_in, 8, out_dev, out_stride, n_in, 8, (size_t)h->C, (size_t)h->C * h->B);
    h->launch(reinterpret_cast<const float2 *>(in_dev), n_in, in_stride, reinterpret_cast<float2 *>(out_dev), out_stride,
              (size_t)h->C * out_stride);
  });
}

int sdrhip_fftconv_process(sdrhip_fftconv *h, const float *in_host, size_t n_in, size_t in_stride, float *out_host,
                           size_t out_stride) {
  return guarded([&] {
    Range roctx_range("sdrhip_fftconv_process");
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    SDRHIP_REQUIRE(!(h->any && h->any->f64), SDRHIP_E_INVALID, "a complex<double> plan: use sdrhip_fftconv_f64_process");
    if (h->any) { h->any->process(in_host, n_in, in_stride, out_host, out_stride); return; }
    SDRHIP_REQUIRE(n_in <= h->max_in, SDRHIP_E_SIZE, "n_in %zu > max_in %zu", n_in, h->max_in);
    if (n_in == 0) return;
    SDRHIP_REQUIRE(in_host && out_host, SDRHIP_E_INVALID, "NULL buffer");
    h->ctx->use();
    if (in_stride == 0) in_stride = n_in;
    if (out_stride == 0) out_stride = n_in;
    SDRHIP_REQUIRE(in_stride >= n_in && out_stride >= n_in, SDRHIP_E_SIZE, "stride smaller than n_in");
    if (!h->stage_in.p) { h->stage_in.alloc((size_t)h->C * h->max_in); h->stage_out.alloc((size_t)h->B * h->C * h->max_in); }
    copy_h2d_rows(h->ctx, h->stage_in.p, n_in * 8, in_host, in_stride * 8, n_in * 8, h->C);
    h->launch(h->stage_in.p, n_in, n_in, h->stage_out.p, n_in, (size_t)h->C * n_in);
    copy_d2h_rows(h->ctx, out_host, out_stride * 8, h->stage_out.p, n_in * 8, n_in * 8, (size_t)h->B * h->C);
    SDRHIP_CHECK_HIP(hipStreamSynchronize(h->ctx->stream));
  });
}

int sdrhip_fftconv_reset(sdrhip_fftconv *h) {
  return guarded([&] {
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    h->ctx->use();
    if (h->any) { h->any->reset(); return; }
    for (int p = 0; p < 2; p++) h->hist[p].zero(h->ctx->stream);
  });
}

int sdrhip_fftconv_destroy(sdrhip_fftconv *h) {
  return guarded([&] {
    if (!h) return;
    h->ctx->use();
    (void)hipStreamSynchronize(h->ctx->stream);
    delete h;
  });
}

// ---- FilterNode<double> (reference src/filternode.hh:230-232: the filter classes are templates over Scalar) ----
int sdrhip_fftconv_f64_create_bank(sdrhip_ctx *ctx, int mode, int fft_size, const double *kernels, int n_taps, int n_bands,
                                   int channels, size_t max_in, sdrhip_fftconv **out) {
  return guarded([&] {
    SDRHIP_REQUIRE(ctx && kernels && out, SDRHIP_E_INVALID, "NULL argument");
    *out = nullptr;
    SDRHIP_REQUIRE(mode == SDRHIP_FFTCONV_OLA || mode == SDRHIP_FFTCONV_OLS, SDRHIP_E_INVALID, "bad mode %d", mode);
    SDRHIP_REQUIRE(channels >= 1 && channels <= 65535, SDRHIP_E_INVALID, "channels %d outside [1,65535]", channels);
    SDRHIP_REQUIRE(n_bands >= 1 && n_bands <= 256, SDRHIP_E_INVALID, "n_bands %d outside [1,256]", n_bands);
    SDRHIP_REQUIRE(max_in >= 1 && max_in < (size_t(1) << 30), SDRHIP_E_SIZE, "max_in %zu outside [1,2^30)", max_in);
    ctx->use();
    sdrhip_fftconv *h = new sdrhip_fftconv;
    try {
      h->ctx = ctx; h->mode = mode; h->C = channels; h->max_in = max_in; h->B = n_bands;
      std::vector<double> taps_all;
      if (mode == SDRHIP_FFTCONV_OLA && !(is_pow2(fft_size) && fft_size <= 8192) && fft_size % 2 == 0 && fft_size >= 4 && !getenv("SDRHIP_FFTCONV_LITERAL")) {
        const int N = fft_size / 2, lp = overlap_save_fft_size(N, 8192);   // (as the complex<float> plans: the general in-LDS plan's power-of-two passes)
        if (lp) {
          for (int b = 0; b < n_bands; b++) {
            const std::vector<double> t = ola_spectrum_to_taps(kernels + (size_t)b * 2 * fft_size, fft_size);
            taps_all.insert(taps_all.end(), t.begin(), t.begin() + 2 * N);
          }
          h->ola_L = fft_size;
          mode = SDRHIP_FFTCONV_OLS; h->mode = mode; fft_size = lp; n_taps = N; kernels = taps_all.data();
        }
      }
      h->any.reset(make_any_conv<double2, double>(ctx, mode, fft_size, kernels, n_taps, n_bands, channels, max_in));
    } catch (...) { delete h; throw; }
    *out = h;
  });
}

int sdrhip_fftconv_f64_set_kernel(sdrhip_fftconv *h, int band, const double *kernel) {
  return guarded([&] {
    SDRHIP_REQUIRE(h && kernel && h->any && h->any->f64, SDRHIP_E_INVALID, "not a complex<double> plan");
    SDRHIP_REQUIRE(band >= 0 && band < h->B, SDRHIP_E_INVALID, "band %d outside [0,%d)", band, h->B);
    h->ctx->use();
    SDRHIP_CHECK_HIP(hipStreamSynchronize(h->ctx->stream));
    std::vector<double> taps;
    if (h->ola_L) { taps = ola_spectrum_to_taps(kernel, h->ola_L); kernel = taps.data(); }
    h->any->load_kernel(band, kernel);
  });
}

int sdrhip_fftconv_f64_process(sdrhip_fftconv *h, const double *in_host, size_t n_in, size_t in_stride, double *out_host, size_t out_stride) {
  return guarded([&] {
    Range roctx_range("sdrhip_fftconv_f64_process");
    SDRHIP_REQUIRE(h && h->any && h->any->f64, SDRHIP_E_INVALID, "not a complex<double> plan");
    h->any->process(in_host, n_in, in_stride, out_host, out_stride);
  });
}

int sdrhip_fftconv_f64_process_dev(sdrhip_fftconv *h, const double *in_dev, size_t n_in, size_t in_stride, double *out_dev, size_t out_stride) {
  return guarded([&] {
    Range roctx_range("sdrhip_fftconv_f64_process_dev");
    SDRHIP_REQUIRE(h && h->any && h->any->f64, SDRHIP_E_INVALID, "not a complex<double> plan");
    h->any->process_dev(in_dev, n_in, in_stride, out_dev, out_stride);
  });
}


int sdrhip_fft_plan_create(sdrhip_ctx *ctx, int dtype, int n, sdrhip_fft_plan **out) {
  return guarded([&] {
    SDRHIP_REQUIRE(ctx && out, SDRHIP_E_INVALID, "NULL argument");
    *out = nullptr;
    SDRHIP_REQUIRE(dtype == SDRHIP_T_CF32 || dtype == SDRHIP_T_CF64, SDRHIP_E_INVALID, "FFT dtype %d: complex<float> or complex<double>", dtype);
    SDRHIP_REQUIRE(n >= 1, SDRHIP_E_INVALID, "FFT size %d", n);
    std::unique_ptr<sdrhip_fft_plan> p(new sdrhip_fft_plan());
    p->build(ctx, dtype, n);
    *out = p.release();
  });
}

int sdrhip_fft_plan_form(sdrhip_fft_plan *p, const char **name) {
  return guarded([&] {
    SDRHIP_REQUIRE(p && name, SDRHIP_E_INVALID, "NULL argument");
    *name = p->form();
  });
}

int sdrhip_fft_plan_exec_dev(sdrhip_fft_plan *p, int sign, int batch, const void *in_dev, void *out_dev) {
  return guarded([&] {
    SDRHIP_REQUIRE(p && in_dev && out_dev && batch >= 1 && (sign == 1 || sign == -1), SDRHIP_E_INVALID, "bad argument");
    p->exec_dev(sign, batch, in_dev, out_dev);
  });
}

int sdrhip_fft_plan_exec(sdrhip_fft_plan *p, int sign, const void *in_host, void *out_host) {
  return guarded([&] {
    SDRHIP_REQUIRE(p && in_host && out_host && (sign == 1 || sign == -1), SDRHIP_E_INVALID, "bad argument");
    p->ctx->use();
    const size_t bytes = (size_t)p->n * p->elem();
    if (!p->stage_in.p) { p->stage_in.alloc(bytes); p->stage_out.alloc(bytes); }
    SDRHIP_CHECK_HIP(hipMemcpyAsync(p->stage_in.p, in_host, bytes, hipMemcpyHostToDevice, p->ctx->stream));
    p->exec_dev(sign, 1, p->stage_in.p, p->stage_out.p);
    SDRHIP_CHECK_HIP(hipMemcpyAsync(out_host, p->stage_out.p, bytes, hipMemcpyDeviceToHost, p->ctx->stream));
    SDRHIP_CHECK_HIP(hipStreamSynchronize(p->ctx->stream));
  });
}

int sdrhip_fft_plan_destroy(sdrhip_fft_plan *p) {
  return guarded([&] {
    if (!p) return;
    p->ctx->use();
    (void)hipStreamSynchronize(p->ctx->stream);
    delete p;
  });
}

int sdrhip_fft_c2c_f64(sdrhip_ctx *ctx, int n, int sign, int batch, const double *in_dev, double *out_dev) {
  return guarded([&] {
    SDRHIP_REQUIRE(ctx && in_dev && out_dev && batch >= 1 && n >= 1 && (sign == 1 || sign == -1), SDRHIP_E_INVALID, "bad argument");
    cached_plan(ctx, SDRHIP_T_CF64, n)->exec_dev(sign, batch, in_dev, out_dev);
    SDRHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  });
}

int sdrhip_fft_exec(sdrhip_ctx *ctx, int dtype, int n, int sign, const void *in_host, void *out_host) {
  return guarded([&] {
    SDRHIP_REQUIRE(ctx && in_host && out_host, SDRHIP_E_INVALID, "NULL argument");
    SDRHIP_REQUIRE(dtype == SDRHIP_T_CF32 || dtype == SDRHIP_T_CF64, SDRHIP_E_INVALID, "FFT dtype %d: complex<float> or complex<double>", dtype);
    SDRHIP_REQUIRE(n >= 1 && (sign == 1 || sign == -1), SDRHIP_E_INVALID, "FFT size %d, sign %d", n, sign);
    const int rc = sdrhip_fft_plan_exec(cached_plan(ctx, dtype, n), sign, in_host, out_host);
    if (rc != SDRHIP_OK) throw Failure{rc};   // (the message is already set)
  });
}

int sdrhip_fft_c2c(sdrhip_ctx *ctx, int n, int sign, int batch, const float *in_dev, float *out_dev) {
  return guarded([&] {
    SDRHIP_REQUIRE(ctx && in_dev && out_dev && batch >= 1 && n >= 1 && (sign == 1 || sign == -1), SDRHIP_E_INVALID, "bad argument");
    cached_plan(ctx, SDRHIP_T_CF32, n)->exec_dev(sign, batch, in_dev, out_dev);
    SDRHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  });
}

}  // extern "C"

#ifdef K7_STAMPS
extern "C" int sdrhip_debug_k7_stamps(sdrhip_fftconv *h, unsigned long long *out, int words) {   // diagnostic builds only (tools/build_variant_k7.sh)
  if (!h || !h->dump.p || !h->stamps_grid) return -3;
  (void)hipStreamSynchronize(h->ctx->stream);
  return hipMemcpy(out, h->dump.p + (size_t)h->stamps_grid * 16 * 16, (size_t)std::min(words, h->stamps_grid * 16) * 8, hipMemcpyDeviceToHost) == hipSuccess ? h->stamps_grid : -3;
}
#endif
