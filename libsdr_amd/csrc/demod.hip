// demod.hip — K4/K5 stand-alone demodulators and K6 SubSample: HBM-bound elementwise kernels.
//
// Replaces (reference, file:line):
//   FMDemod<int16_t,int16_t>::_process + fast_atan2   src/demod.hh:242-254, src/math.hh:31-40
//   AMDemod<Scalar>::process                          src/demod.hh:65-81
//   USBDemod<Scalar>::_process                        src/demod.hh:156-161
//   SubSample<Scalar>::_process                       src/subsample.hh:92-101
// Each lane handles 4 consecutive samples: one 16-byte (cs16) or two 16-byte (cf32) loads, one
// 8/16-byte store; grid-stride over the (channel, sample) plane.
#include "fm_phi.hpp"
#include "sdrhip_internal.hpp"

using namespace sdrhip;

namespace {

constexpr int TPB = 256;

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
__device__ __forceinline__ int lo16(uint32_t v) { return (short)(v & 0xffffu); }
__device__ __forceinline__ int hi16(uint32_t v) { return (short)(v >> 16); }

struct DemodArgs {
  const void *in; long in_stride; void *out; long out_stride;
  int N, kind, fm0;
  const short *fm_old; short *fm_new;
};

// cs16 -> int16. A lane owns SP consecutive samples (SP = 4 or 8: 16-byte loads, 8-byte stores): FM needs the angle of the
// sample before its first one, so a lane evaluates SP + 1 angles for SP outputs — the stand-alone FM demodulator is bound
// by that arithmetic (32 vector instructions per angle), not by HBM: 8 samples per lane make it 9/8 instead of 5/4.
template <int SP>
__global__ __launch_bounds__(TPB) void demod_cs16_kernel(const DemodArgs a) {
  const int c = blockIdx.y;
  const uint32_t *in = reinterpret_cast<const uint32_t *>(a.in) + (long)c * a.in_stride;
  short *out = reinterpret_cast<short *>(a.out) + (long)c * a.out_stride;
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(in) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 7) == 0);
  for (int i0 = SP * (blockIdx.x * TPB + threadIdx.x); i0 < a.N; i0 += SP * gridDim.x * TPB) {
    uint32_t x[SP];
    const int cnt = min(SP, a.N - i0);
    if (vec_ok && cnt == SP) {
#pragma unroll
      for (int q = 0; q < SP / 4; q++) {
        const uint4 v = *reinterpret_cast<const uint4 *>(in + i0 + 4 * q);
        x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < SP; k++) x[k] = k < cnt ? in[i0 + k] : 0u;
    }
    short o[SP];
    if (a.kind == SDRHIP_EPI_AM) {
#pragma unroll
      for (int k = 0; k < SP; k++) o[k] = am_i16(lo16(x[k]), hi16(x[k]));
    } else if (a.kind == SDRHIP_EPI_USB) {
#pragma unroll
      for (int k = 0; k < SP; k++) o[k] = usb_i16(lo16(x[k]), hi16(x[k]));
    } else {
      // out[i] = phi[i-1] - phi[i] (i >= 2); out[1] = last - phi[1]; out[0]: never written by FMDemod
      int prev = 0;
      if (i0 >= 2) { const uint32_t p = in[i0 - 1]; prev = fm_phi(lo16(p), hi16(p)); }
#pragma unroll
      for (int k = 0; k < SP; k++) {
        const int i = i0 + k;
        const int phi = fm_phi(lo16(x[k]), hi16(x[k]));
        if (i == 0) o[k] = (short)lo16(x[k]);
        else if (i == 1) o[k] = (short)((int)a.fm_old[c] - phi);
        else o[k] = (short)(prev - phi);
        prev = phi;
        if (i == a.N - 1 && a.N >= 2) a.fm_new[c] = (short)phi;
      }
    }
    const bool skip0 = (a.kind == SDRHIP_EPI_FM) && !a.fm0 && i0 == 0;
    if (vec_ok && cnt == SP && !skip0) {
#pragma unroll
      for (int q = 0; q < SP / 4; q++) {
        uint2 pk;
        pk.x = (uint32_t)(uint16_t)o[4 * q] | ((uint32_t)(uint16_t)o[4 * q + 1] << 16);
        pk.y = (uint32_t)(uint16_t)o[4 * q + 2] | ((uint32_t)(uint16_t)o[4 * q + 3] << 16);
        *reinterpret_cast<uint2 *>(out + i0 + 4 * q) = pk;
      }
    } else {
      for (int k = 0; k < cnt; k++) if (!(skip0 && k == 0)) out[i0 + k] = o[k];
    }
  }
}

// complex<int8_t> -> int16: FMDemod<int8_t,int16_t> (reference src/demod.hh:242-254 with fast_atan2<int8_t,int16_t>,
// src/math.hh:12-21 — the formula of the int16 form on int8 inputs). In place an output element (2 B) covers exactly its
// input sample (2 B): out[0] of a call, which FMDemod never writes, is the two bytes of in[0].
__global__ __launch_bounds__(TPB) void demod_cs8_fm_kernel(const DemodArgs a) {
  const int c = blockIdx.y;
  const uint16_t *in = reinterpret_cast<const uint16_t *>(a.in) + (long)c * a.in_stride;
  short *out = reinterpret_cast<short *>(a.out) + (long)c * a.out_stride;
  for (int i = blockIdx.x * TPB + threadIdx.x; i < a.N; i += gridDim.x * TPB) {
    const uint32_t x = in[i];
    const int phi = fm_phi((int)(signed char)(x & 0xffu), (int)(signed char)(x >> 8));
    short o;
    if (i == 0) o = (short)x;
    else if (i == 1) o = (short)((int)a.fm_old[c] - phi);
    else { const uint32_t p = in[i - 1]; o = (short)(fm_phi((int)(signed char)(p & 0xffu), (int)(signed char)(p >> 8)) - phi); }
    if (i == a.N - 1 && a.N >= 2) a.fm_new[c] = (short)phi;
    if (!(i == 0 && !a.fm0)) out[i] = o;
  }
}

// cf32 -> float (AM, USB)
__global__ __launch_bounds__(TPB) void demod_cf32_kernel(const DemodArgs a) {
  const int c = blockIdx.y;
  const float2 *in = reinterpret_cast<const float2 *>(a.in) + (long)c * a.in_stride;
  float *out = reinterpret_cast<float *>(a.out) + (long)c * a.out_stride;
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(in) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  for (int i0 = 4 * (blockIdx.x * TPB + threadIdx.x); i0 < a.N; i0 += 4 * gridDim.x * TPB) {
    float2 x[4];
    const int cnt = min(4, a.N - i0);
    if (vec_ok && cnt == 4) {
      const float4 v0 = *reinterpret_cast<const float4 *>(in + i0), v1 = *reinterpret_cast<const float4 *>(in + i0 + 2);
      x[0] = make_float2(v0.x, v0.y); x[1] = make_float2(v0.z, v0.w);
      x[2] = make_float2(v1.x, v1.y); x[3] = make_float2(v1.z, v1.w);
    } else {
      for (int k = 0; k < 4; k++) x[k] = k < cnt ? in[i0 + k] : make_float2(0.f, 0.f);
    }
    float o[4];
    for (int k = 0; k < 4; k++)
      o[k] = (a.kind == SDRHIP_EPI_AM) ? sqrtf(x[k].x * x[k].x + x[k].y * x[k].y) : (x[k].x + x[k].y) / 2;
    if (vec_ok && cnt == 4) *reinterpret_cast<float4 *>(out + i0) = make_float4(o[0], o[1], o[2], o[3]);
    else for (int k = 0; k < cnt; k++) out[i0 + k] = o[k];
  }
}

// ---------------------------------------------------------------------------------------------
// K6 SubSample: groups of n inputs counted from the last reset; the open group's partial sum is
// carried between calls.
// ---------------------------------------------------------------------------------------------
struct SubArgs {
  const void *in; long in_stride; void *out; long out_stride;
  int N, n;
  int first_rel;   // call-relative index of the first sample of the first group touched (<= 0)
  int n_groups, n_out;
  const void *acc_old; void *acc_new;
};

__device__ __forceinline__ int cdiv_int(int s, int n) {   // (s*n)/(n*n), wrapping, truncating
  const int d = (int)((unsigned)n * (unsigned)n);
  const int r = (int)((unsigned)s * (unsigned)n);
  if (d == 0) return 0;
  if (r == (int)0x80000000 && d == -1) return r;
  return r / d;
}

__global__ __launch_bounds__(TPB) void subsample_cs16_kernel(const SubArgs a) {
  const int c = blockIdx.y;
  const uint32_t *in = reinterpret_cast<const uint32_t *>(a.in) + (long)c * a.in_stride;
  uint32_t *out = reinterpret_cast<uint32_t *>(a.out) + (long)c * a.out_stride;
  const int2 *acc_old = reinterpret_cast<const int2 *>(a.acc_old);
  int2 *acc_new = reinterpret_cast<int2 *>(a.acc_new);
  for (int g = blockIdx.x * TPB + threadIdx.x; g < a.n_groups; g += gridDim.x * TPB) {
    const int lo = max(0, a.first_rel + g * a.n), hi = min(a.N, a.first_rel + (g + 1) * a.n);
    int sr = 0, si = 0;
    if (g == 0) { const int2 cy = acc_old[c]; sr = cy.x; si = cy.y; }
    for (int i = lo; i < hi; i++) {
      const uint32_t v = in[i];
      sr = (int)((unsigned)sr + (unsigned)lo16(v)); si = (int)((unsigned)si + (unsigned)hi16(v));
    }
    const bool emits = g < a.n_out;
    if (emits) {
      const int yr = cdiv_int(sr, a.n), yi = cdiv_int(si, a.n);
      out[g] = (uint32_t)(uint16_t)yr | ((uint32_t)(uint16_t)yi << 16);
    }
    if (g == a.n_groups - 1) acc_new[c] = emits ? make_int2(0, 0) : make_int2(sr, si);
  }
}

// n == 8, the call starts on a group boundary, whole groups only, 16-byte aligned rows: a lane loads 4 samples (16 bytes,
// consecutive lanes consecutive quads: every wave-load is 1 KB contiguous — the general kernel's lanes walk 32-byte
// strides with 4-byte loads), lane pairs add their halves of a group through DPP (int32 sums are order-free) and the
// even lane stores the average.
__global__ __launch_bounds__(TPB) void subsample8_cs16_kernel(const SubArgs a) {
  const int c = blockIdx.y;
  const uint32_t *in = reinterpret_cast<const uint32_t *>(a.in) + (long)c * a.in_stride;
  uint32_t *out = reinterpret_cast<uint32_t *>(a.out) + (long)c * a.out_stride;
  const int quads = 2 * a.n_out;
  for (int q = blockIdx.x * TPB + threadIdx.x; q < quads; q += gridDim.x * TPB) {   // (quads is even: lane pairs stay together)
    const uint4 v = *reinterpret_cast<const uint4 *>(in + 4 * q);
    const int sr = lo16(v.x) + lo16(v.y) + lo16(v.z) + lo16(v.w), si = hi16(v.x) + hi16(v.y) + hi16(v.z) + hi16(v.w);
    const int pr = __builtin_amdgcn_update_dpp(0, sr, 0xB1, 0xf, 0xf, true), pi = __builtin_amdgcn_update_dpp(0, si, 0xB1, 0xf, 0xf, true);
    if (!(q & 1)) {
      const int yr = cdiv_int(sr + pr, 8), yi = cdiv_int(si + pi, 8);
      out[q >> 1] = (uint32_t)(uint16_t)yr | ((uint32_t)(uint16_t)yi << 16);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<int2 *>(a.acc_new)[c] = make_int2(0, 0);   // (no open group)
}

__global__ __launch_bounds__(TPB) void subsample_cf32_kernel(const SubArgs a) {
  const int c = blockIdx.y;
  const float2 *in = reinterpret_cast<const float2 *>(a.in) + (long)c * a.in_stride;
  float2 *out = reinterpret_cast<float2 *>(a.out) + (long)c * a.out_stride;
  const float2 *acc_old = reinterpret_cast<const float2 *>(a.acc_old);
  float2 *acc_new = reinterpret_cast<float2 *>(a.acc_new);
  const float d = (float)a.n;
  for (int g = blockIdx.x * TPB + threadIdx.x; g < a.n_groups; g += gridDim.x * TPB) {
    const int lo = max(0, a.first_rel + g * a.n), hi = min(a.N, a.first_rel + (g + 1) * a.n);
    float sr = 0.f, si = 0.f;
    if (g == 0) { const float2 cy = acc_old[c]; sr = cy.x; si = cy.y; }
    for (int i = lo; i < hi; i++) { const float2 v = in[i]; sr += v.x; si += v.y; }   // same order as the reference
    const bool emits = g < a.n_out;
    if (emits) out[g] = make_float2(sr / d, si / d);
    if (g == a.n_groups - 1) acc_new[c] = emits ? make_float2(0.f, 0.f) : make_float2(sr, si);
  }
}


// ---------------------------------------------------------------------------------------------
// FMDeemph<int16_t> (src/demod.hh:342-351): avg += (x - avg +/- alpha/2) / alpha, all in int with an int16
// wrap of the difference and of the average — a nonlinear recursion, sequential per channel. Four kernels:
//   deemph_i16_copy_kernel   alpha = 1: the update is avg = x
//   deemph_i16_seq_kernel    one lane walks one channel's row in registers
//   deemph_i16_spec_kernel   long rows, alpha <= 32: P lanes per channel from guessed segment states, checked and repaired
//   deemph_i16_kernel        rounds 1-2 (kept as a parity form, SDRHIP_DEEMPH_TILED): a 64-lane workgroup moves 64 channels x 128
//                            samples at a time through LDS so that global loads and stores stay coalesced along the channel rows
// ---------------------------------------------------------------------------------------------
constexpr int DE_CH = 64, DE_N = 128, DE_LD = DE_N + 2;

struct DeemphArgs {
  const short *in; long in_stride; short *out; long out_stride;
  int N, C, alpha;
  short *avg;   // one per channel, updated in place (each channel has exactly one lane)
  unsigned magic;   // floor(2^32 / alpha) + 1: trunc(m / alpha) = umulhi(m, magic) for m < 2^16 (alpha < 2^15)
};

// The recursion as a latency chain, nothing else: one lane walks one channel's row in registers — 16-byte loads, DE_PF chunks
// of 8 samples in flight ahead of the arithmetic (a lane's loads are its own row: uncoalesced, but a buffer of demodulated
// audio is a megabyte), the division by the node's constant alpha as one multiply-high, 16-byte stores. The LDS-tiled
// kernel below (coalesced, but 2-byte loads by the 64 lanes that also do the arithmetic, an LDS round trip and a float
// reciprocal per step) took 172 us for 1024 channels x 524 samples — twice the baseband kernel in front of it in the
// reference's sdr_fm chain — where this one takes the chain's latency.
constexpr int DE_PF = 8;
// One step in plain C++ (the rows' first and last samples, off the 16-byte chunks; the chunks run deemph_chunk below)
__device__ __forceinline__ int deemph_step(int x, int &avg, int half, unsigned magic) {
  const int d = (int)(short)(x - avg);                 // the int16 wrap of the difference
  const int s = (d - 1) >> 31;                         // -1 for d <= 0 (the reference subtracts alpha / 2 then), else 0
  const int n = d + half + (s & (-2 * half));          // d > 0 ? d + half : d - half
  const int sn = n >> 31;
  const unsigned m = (unsigned)((n ^ sn) - sn);        // |n| < 2^16
  const int q = (int)__umulhi(m, magic);
  avg = (int)(short)(avg + ((q ^ sn) - sn));           // trunc(n / alpha); ... and the int16 wrap of the average
  return avg;
}
// Eight steps on a 16-byte chunk, 5.5 vector instructions per step (a step of deemph_step compiles to 14, and a lone wave's
// chain is bound by its SIMD's issue port: 4 cycles per instruction whatever the number of active lanes):
//   d  = int16(x - avg)                  one SDWA subtract: 16-bit halves read sign-extended, the result's low half sign-extended
//   sg = sign(d) = med3(d, -1, 1)        (d = 0: the reference subtracts alpha / 2 and divides -alpha / 2 to 0 — as 0 * anything)
//   m  = |d| = d * sg                    24-bit multiply
//   q  = (m + alpha / 2) / alpha         the high half of m * magic + (alpha / 2) * magic: one 64-bit multiply-add
//   avg += q * sg                        24-bit multiply-add; avg's upper half is never wrapped — only its low 16 bits are read
template <int HI>
__device__ __forceinline__ void deemph_half(uint32_t w, int &avg, unsigned magic, unsigned long long hm) {
  int d;
  if (HI) asm("v_sub_u32_sdwa %0, sext(%1), sext(%2) dst_sel:WORD_0 dst_unused:UNUSED_SEXT src0_sel:WORD_1 src1_sel:WORD_0" : "=v"(d) : "v"(w), "v"(avg));
  else asm("v_sub_u32_sdwa %0, sext(%1), sext(%2) dst_sel:WORD_0 dst_unused:UNUSED_SEXT src0_sel:WORD_0 src1_sel:WORD_0" : "=v"(d) : "v"(w), "v"(avg));
  int sg;
  asm("v_med3_i32 %0, %1, -1, 1" : "=v"(sg) : "v"(d));   // (left to itself the compiler builds two compare + select pairs here,
  const unsigned m = (unsigned)__mul24(d, sg);            //  and a 64-bit add, a multiply-high and a multiply-add for the line below)
  unsigned long long r;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(m), "s"(magic), "v"(hm) : "vcc");
  avg = __mul24((int)(unsigned)(r >> 32), sg) + avg;
}
__device__ __forceinline__ void deemph_chunk(uint32_t (&w)[4], int &avg, int half, unsigned magic) {
  const unsigned long long hm = (unsigned long long)(unsigned)half * magic;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    deemph_half<0>(w[j], avg, magic, hm);
    const int y0 = avg;
    deemph_half<1>(w[j], avg, magic, hm);
    w[j] = __builtin_amdgcn_perm((uint32_t)avg, (uint32_t)y0, 0x05040100u);   // the low halves of (y0, avg)
  }
  avg = (int)(short)avg;
}
__global__ __launch_bounds__(64) void deemph_i16_seq_kernel(const DeemphArgs a) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= a.C) return;
  const short *in = a.in + (long)c * a.in_stride;
  short *out = a.out + (long)c * a.out_stride;
  int avg = (int)a.avg[c];
  const int half = a.alpha / 2;
  const unsigned magic = a.magic;
  // the row's samples before its first 16-byte boundary (0 .. 7, a lane's own count), one by one
  const int head = min(a.N, (int)(((16u - (unsigned)(reinterpret_cast<uintptr_t>(in) & 15u)) & 15u) >> 1));
  {
    short hd[7];
#pragma unroll
    for (int j = 0; j < 7; j++) hd[j] = in[min(j, max(head - 1, 0))];   // (all loads first: one latency)
#pragma unroll
    for (int j = 0; j < 7; j++) if (j < head) out[j] = (short)deemph_step((int)hd[j], avg, half, magic);
  }
  const int nch = (a.N - head) >> 3;   // whole 16-byte chunks of 8 samples
  const uint4 *in4 = reinterpret_cast<const uint4 *>(in + head);
  short *o8 = out + head;
  const bool out16 = (reinterpret_cast<uintptr_t>(o8) & 15u) == 0;   // (else eight 2-byte stores per chunk)
  uint4 cur[DE_PF];
  if (nch > 0) {
#pragma unroll
    for (int k = 0; k < DE_PF; k++) cur[k] = in4[min(k, nch - 1)];   // (clamped: a short row repeats its last chunk, unused)
  }
  for (int g = 0; g < nch; g += DE_PF) {
    uint4 nxt[DE_PF];
#pragma unroll
    for (int k = 0; k < DE_PF; k++) nxt[k] = in4[min(g + DE_PF + k, nch - 1)];   // the next group, in flight during this one's arithmetic
#pragma unroll
    for (int k = 0; k < DE_PF; k++) {
      if (g + k < nch) {
        uint32_t w[4] = {cur[k].x, cur[k].y, cur[k].z, cur[k].w};
        deemph_chunk(w, avg, half, magic);
        if (out16) reinterpret_cast<uint4 *>(o8)[g + k] = make_uint4(w[0], w[1], w[2], w[3]);
        else {
#pragma unroll
          for (int j = 0; j < 4; j++) { o8[8 * (g + k) + 2 * j] = (short)(w[j] & 0xffffu); o8[8 * (g + k) + 2 * j + 1] = (short)(w[j] >> 16); }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < DE_PF; k++) cur[k] = nxt[k];
  }
  const int done = head + 8 * nch, rem = a.N - done;   // the last samples (< 8): all loads first, then the chain
  if (rem > 0) {
    short tl[7];
#pragma unroll
    for (int j = 0; j < 7; j++) tl[j] = in[done + min(j, rem - 1)];
#pragma unroll
    for (int j = 0; j < 7; j++) if (j < rem) out[done + j] = (short)deemph_step((int)tl[j], avg, half, magic);
  }
  a.avg[c] = (short)avg;
}

// The same chain, cut in time. The recursion cannot be split exactly ahead of time — but it forgets: two runs over the same
// samples from different averages never cross and close in (the update is monotone in the average, |difference| never
// grows), and once they agree they agree for good. So P lanes share a channel: lane p owns the p-th segment of the row and
// first runs over the `wc` chunks in front of it from a guessed average (the sample there), which leaves it — almost
// always — with the exact state at its segment's start. "Almost" is then removed: after the pass, every lane compares the
// state it started from with the final state of the lane before it (lane 0 started from the node's true state); a lane that
// guessed wrong runs its segment again from the right state, until its average meets the value it had stored at the same
// index (from there on the stored row is right) or the segment ends (then its final state changes and the lane after it
// checks again). The result is the sequential recursion's, bit for bit, whatever the data; what the data decides is the
// time: (n / P + 8 wc) steps when the guesses hold, at worst (constant rows sit inside the rounding dead zone and never
// meet) one sequential pass on top.
struct DeemphSpecArgs { DeemphArgs d; int lgP, Lc, wc; };
__device__ __forceinline__ void deemph_store(short *o8, bool out16, int id, const uint32_t (&w)[4]) {
  if (out16) reinterpret_cast<uint4 *>(o8)[id] = make_uint4(w[0], w[1], w[2], w[3]);
  else {
#pragma unroll
    for (int j = 0; j < 4; j++) { o8[8 * id + 2 * j] = (short)(w[j] & 0xffffu); o8[8 * id + 2 * j + 1] = (short)(w[j] >> 16); }
  }
}
__global__ __launch_bounds__(256) void deemph_i16_spec_kernel(const DeemphSpecArgs sa) {
  __shared__ int redo_from[128], redo_state[128];   // per channel of the workgroup (at most 128: P >= 2): where a given-up row resumes
  const DeemphArgs &a = sa.d;
  const int P = 1 << sa.lgP, Lc = sa.Lc;
  const int lin = blockIdx.x * 256 + threadIdx.x;
  const bool live = (lin >> sa.lgP) < a.C;
  const int c = min(lin >> sa.lgP, a.C - 1), p = lin & (P - 1);   // (lanes beyond the last channel shadow it, without stores)
  const short *in = a.in + (long)c * a.in_stride;
  short *out = a.out + (long)c * a.out_stride;
  const int half = a.alpha / 2;
  const unsigned magic = a.magic;
  const int head = min(a.N, (int)(((16u - (unsigned)(reinterpret_cast<uintptr_t>(in) & 15u)) & 15u) >> 1));
  const int nch = (a.N - head) >> 3;
  const uint4 *in4 = reinterpret_cast<const uint4 *>(in + head);
  short *o8 = out + head;
  const bool out16 = (reinterpret_cast<uintptr_t>(o8) & 15u) == 0;
  const int c0 = p * Lc;                          // the lane's first chunk; its segment: [c0, min(c0 + Lc, nch))
  int avg;
  if (p == 0) {                                   // the channel's first lane: the true state, and the samples before the first chunk
    avg = (int)a.avg[c];
    short hd[7];
#pragma unroll
    for (int j = 0; j < 7; j++) hd[j] = in[min(j, max(head - 1, 0))];
#pragma unroll
    for (int j = 0; j < 7; j++) if (j < head) { const int y = deemph_step((int)hd[j], avg, half, magic); if (live) out[j] = (short)y; }
  } else avg = (int)(in + head)[8 * min(max(c0 - sa.wc, 0), max(nch - 1, 0))];   // the guess: the sample the run-in starts at
  int s = avg, f;
  uint4 cur[DE_PF];
#pragma unroll
  for (int k = 0; k < DE_PF; k++) cur[k] = in4[min(max(c0 - sa.wc + k, 0), max(nch - 1, 0))];
  for (int g = -sa.wc; g < Lc; g += DE_PF) {      // (wc is a multiple of DE_PF: the segment starts at a group boundary)
    if (g == 0) s = avg;
    uint4 nxt[DE_PF];
#pragma unroll
    for (int k = 0; k < DE_PF; k++) nxt[k] = in4[min(max(c0 + g + DE_PF + k, 0), max(nch - 1, 0))];
#pragma unroll
    for (int k = 0; k < DE_PF; k++) {
      const int id = c0 + g + k;
      if (id >= 0 && id < nch && g + k < Lc && (p > 0 || g >= 0)) {
        uint32_t w[4] = {cur[k].x, cur[k].y, cur[k].z, cur[k].w};
        deemph_chunk(w, avg, half, magic);
        if (g >= 0 && live) deemph_store(o8, out16, id, w);
      }
    }
#pragma unroll
    for (int k = 0; k < DE_PF; k++) cur[k] = nxt[k];
  }
  f = avg;
  // the check: P - 1 rounds at most (a wrong state can travel one lane per round)
  bool given_up = false;
  for (int it = 1; it < P; it++) {
    const int pf = __shfl_up(f, 1, 64);
    const bool need = !given_up && p > 0 && pf != s;
    if (!__any(need)) break;
    bool unmet = false;
    if (need) {
      s = pf;
      int av = pf;
      bool met = false;
      const int hi = min(c0 + Lc, nch);
      // (loads a group ahead, as in the pass above: a row that never meets — a constant one — costs the sequential chain, not
      // the chain plus a load latency per chunk)
      uint4 cu[DE_PF];
      short ol[DE_PF];
#pragma unroll
      for (int k = 0; k < DE_PF; k++) { const int id = min(c0 + k, nch - 1); cu[k] = in4[id]; ol[k] = o8[8 * id + 7]; }
      for (int g = c0; g < hi && !met; g += DE_PF) {
        uint4 nx[DE_PF];
        short no[DE_PF];
#pragma unroll
        for (int k = 0; k < DE_PF; k++) { const int id = min(g + DE_PF + k, nch - 1); nx[k] = in4[id]; no[k] = o8[8 * id + 7]; }
#pragma unroll
        for (int k = 0; k < DE_PF; k++) {
          if (g + k < hi && !met) {
            uint32_t w[4] = {cu[k].x, cu[k].y, cu[k].z, cu[k].w};
            deemph_chunk(w, av, half, magic);
            if (live) deemph_store(o8, out16, g + k, w);
            met = av == (int)ol[k];   // the value stored here by the run from the wrong state: from now on the two are one
          }
        }
#pragma unroll
        for (int k = 0; k < DE_PF; k++) { cu[k] = nx[k]; ol[k] = no[k]; }
      }
      if (c0 >= nch) f = pf;   // an empty segment (P * Lc > nch): its end state IS its start state, exact now — nothing to repair
      else if (!met) { f = av; unmet = true; }
    }
    // a lane that ran to its segment's end without meeting (a run-in AND a segment of 16 alpha samples each were not enough):
    // this row does not forget — a constant stretch: every run stops inside the rounding dead zone, each on its own value —
    // and the rounds would hand the right state on one lane at a time, each with its load latencies. After round `it` the
    // lanes 0 .. it are exact (each was compared with an exact predecessor): the row is walked from lane it's end, alone.
    const unsigned long long um = __ballot(unmet) >> (threadIdx.x & 63 & ~(P - 1));
    if (!given_up && (um & ((P == 64) ? ~0ull : ((1ull << P) - 1ull))) != 0) {
      given_up = true;
      if (p == it) { redo_from[threadIdx.x >> sa.lgP] = (it + 1) * Lc; redo_state[threadIdx.x >> sa.lgP] = f; }
    }
  }
  if (p == P - 1 && !given_up) {                  // the last samples (< 8) and the state
    redo_from[threadIdx.x >> sa.lgP] = -1;
    avg = f;
    const int done = head + 8 * nch, rem = a.N - done;
    if (rem > 0) {
      short tl[7];
#pragma unroll
      for (int j = 0; j < 7; j++) tl[j] = in[done + min(j, rem - 1)];
#pragma unroll
      for (int j = 0; j < 7; j++) if (j < rem) { const int y = deemph_step((int)tl[j], avg, half, magic); if (live) out[done + j] = (short)y; }
    }
    if (live) a.avg[c] = (short)avg;
  }
  // The rows that were given up, one per LANE of the workgroup's first wave: a wave's chain keeps its SIMD's issue port busy
  // whatever the number of active lanes (about 14 instructions of 4 cycles per step), so the walkers are packed — left in
  // their own waves (one or two active lanes each, several waves to a SIMD) a constant batch took twice the one-lane kernel's
  // time; like this it takes that time plus the first pass.
  const int teams = 256 >> sa.lgP;
  __syncthreads();
  bool any_redo = false;
  for (int t = 0; t < teams; t++) any_redo |= redo_from[t] >= 0;
  if (any_redo) {   // (workgroup-uniform) every lane's stores to the rows are done before a row is written again — by another wave
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();
  }
  if (threadIdx.x < teams && redo_from[threadIdx.x] >= 0 && blockIdx.x * teams + threadIdx.x < a.C) {
    const int cw = blockIdx.x * teams + threadIdx.x;
    const short *inw = a.in + (long)cw * a.in_stride;
    short *outw = a.out + (long)cw * a.out_stride;
    const int hw = min(a.N, (int)(((16u - (unsigned)(reinterpret_cast<uintptr_t>(inw) & 15u)) & 15u) >> 1));
    const int nw = (a.N - hw) >> 3;
    const uint4 *i4 = reinterpret_cast<const uint4 *>(inw + hw);
    short *ow = outw + hw;
    const bool o16 = (reinterpret_cast<uintptr_t>(ow) & 15u) == 0;
    int av = redo_state[threadIdx.x];
    const int from = redo_from[threadIdx.x];
    uint4 cu[DE_PF];
#pragma unroll
    for (int k = 0; k < DE_PF; k++) cu[k] = i4[min(from + k, nw - 1)];
    for (int g = from; g < nw; g += DE_PF) {
      uint4 nx[DE_PF];
#pragma unroll
      for (int k = 0; k < DE_PF; k++) nx[k] = i4[min(g + DE_PF + k, nw - 1)];
#pragma unroll
      for (int k = 0; k < DE_PF; k++) {
        if (g + k < nw) {
          uint32_t w[4] = {cu[k].x, cu[k].y, cu[k].z, cu[k].w};
          deemph_chunk(w, av, half, magic);
          deemph_store(ow, o16, g + k, w);
        }
      }
#pragma unroll
      for (int k = 0; k < DE_PF; k++) cu[k] = nx[k];
    }
    const int done = hw + 8 * nw, rem = a.N - done;
    if (rem > 0) {
      short tl[7];
#pragma unroll
      for (int j = 0; j < 7; j++) tl[j] = inw[done + min(j, rem - 1)];
#pragma unroll
      for (int j = 0; j < 7; j++) if (j < rem) outw[done + j] = (short)deemph_step((int)tl[j], av, half, magic);
    }
    a.avg[cw] = (short)av;
  }
}

// alpha = 1 (sample rates below about 11 kS/s: src/demod.hh:305-306 rounds 1 / (1 - exp(-1 / (Fs 75 us))) to 1): the
// update is avg = x exactly — a copy, and the last sample as the state
__global__ __launch_bounds__(256) void deemph_i16_copy_kernel(const DeemphArgs a) {
  const int c = blockIdx.y;
  const short *in = a.in + (long)c * a.in_stride;
  short *out = a.out + (long)c * a.out_stride;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.N; i += gridDim.x * 256) out[i] = in[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) a.avg[c] = in[a.N - 1];
}

__global__ __launch_bounds__(DE_CH) void deemph_i16_kernel(const DeemphArgs a) {
  __shared__ short tile[DE_CH * DE_LD];
  const int c0 = blockIdx.x * DE_CH, lane = threadIdx.x;
  const int nch = min(DE_CH, a.C - c0);
  int avg = lane < nch ? (int)a.avg[c0 + lane] : 0;
  const int half = a.alpha / 2;
  for (int n0 = 0; n0 < a.N; n0 += DE_N) {
    const int cnt = min(DE_N, a.N - n0);
    for (int idx = lane; idx < nch * DE_N; idx += DE_CH) {
      const int ch = idx / DE_N, i = idx % DE_N;
      if (i < cnt) tile[ch * DE_LD + i] = a.in[(long)(c0 + ch) * a.in_stride + n0 + i];
    }
    __syncthreads();
    if (lane < nch) {
      short *row = tile + lane * DE_LD;
      for (int i = 0; i < cnt; i++) {
        const int diff = (short)((int)row[i] - avg);
        const int q = div_small(diff > 0 ? diff + half : diff - half, a.alpha);
        avg = (short)(avg + q);
        row[i] = (short)avg;
      }
    }
    __syncthreads();
    for (int idx = lane; idx < nch * DE_N; idx += DE_CH) {
      const int ch = idx / DE_N, i = idx % DE_N;
      if (i < cnt) a.out[(long)(c0 + ch) * a.out_stride + n0 + i] = tile[ch * DE_LD + i];
    }
    __syncthreads();
  }
  if (lane < nch) a.avg[c0 + lane] = (short)avg;
}

}  // namespace

struct sdrhip_demod {
  sdrhip_ctx *ctx = nullptr;
  int kind = 0, dtype = 0, C = 1, fm0 = 1, par_fm = 0;
  size_t max_in = 0;
  DevBuf<short> fm[2];
  DevBuf<uint8_t> stage_in, stage_out;
  size_t in_elem() const { return dtype == SDRHIP_T_CS16 ? 4 : dtype == SDRHIP_T_CS8 ? 2 : 8; }
  size_t out_elem() const { return dtype == SDRHIP_T_CF32 ? 4 : 2; }
  void launch(const void *in_dev, size_t N, size_t in_stride, void *out_dev, size_t out_stride) {
    ctx->use();
    if (N == 0) return;   // FMDemod::process returns without sending on an empty buffer (src/demod.hh:231)
    DemodArgs a;
    a.in = in_dev; a.in_stride = (long)in_stride; a.out = out_dev; a.out_stride = (long)out_stride;
    a.N = (int)N; a.kind = kind; a.fm0 = fm0;
    a.fm_old = fm[par_fm].p; a.fm_new = fm[par_fm ^ 1].p;
    const bool fm8 = dtype == SDRHIP_T_CS16 && kind == SDRHIP_EPI_FM && N >= (size_t)8 * TPB;   // (AM / USB have no angle to repeat)
    const bool fm16 = fm8 && N >= (size_t)16 * TPB;   // (17 angles per 16 outputs: +3 % over 8 per lane)
    const unsigned bx = (unsigned)std::min<size_t>(ceil_div(N, (size_t)(fm16 ? 16 : fm8 ? 8 : 4) * TPB), 4096);
    dim3 grid(bx, C), block(TPB);
    if (fm16) hipLaunchKernelGGL(demod_cs16_kernel<16>, grid, block, 0, ctx->stream, a);
    else if (fm8) hipLaunchKernelGGL(demod_cs16_kernel<8>, grid, block, 0, ctx->stream, a);
    else if (dtype == SDRHIP_T_CS16) hipLaunchKernelGGL(demod_cs16_kernel<4>, grid, block, 0, ctx->stream, a);
    else if (dtype == SDRHIP_T_CS8) hipLaunchKernelGGL(demod_cs8_fm_kernel, grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL(demod_cf32_kernel, grid, block, 0, ctx->stream, a);
    SDRHIP_CHECK_HIP(hipGetLastError());
    if (kind == SDRHIP_EPI_FM && N >= 2) par_fm ^= 1;
  }
};

struct sdrhip_subsample {
  sdrhip_ctx *ctx = nullptr;
  int dtype = 0, C = 1, par = 0;
  size_t n = 1, max_in = 0, max_out = 0;
  uint64_t n0 = 0;
  DevBuf<uint8_t> acc[2];
  DevBuf<uint8_t> stage_in, stage_out;
  size_t elem() const { return dtype == SDRHIP_T_CS16 ? 4 : 8; }
  size_t out_count(size_t N) const { return (size_t)((n0 + N) / n - n0 / n); }
  void launch(const void *in_dev, size_t N, size_t in_stride, void *out_dev, size_t out_stride, size_t *n_out) {
    ctx->use();
    if (N == 0) { if (n_out) *n_out = 0; return; }
    const size_t no = out_count(N);
    SDRHIP_REQUIRE(out_stride >= no, SDRHIP_E_SIZE, "out_stride %zu < outputs %zu", out_stride, no);
    SubArgs a;
    a.in = in_dev; a.in_stride = (long)in_stride; a.out = out_dev; a.out_stride = (long)out_stride;
    a.N = (int)N; a.n = (int)n;
    const uint64_t g0 = n0 / n, g1 = (n0 + N - 1) / n;
    a.first_rel = (int)((int64_t)(g0 * n) - (int64_t)n0);
    a.n_groups = (int)(g1 - g0 + 1); a.n_out = (int)no;
    a.acc_old = acc[par].p; a.acc_new = acc[par ^ 1].p;
    const unsigned bx = (unsigned)std::min<size_t>(ceil_div((size_t)a.n_groups, (size_t)TPB), 4096);
    dim3 grid(bx, C), block(TPB);
    const bool fast8 = dtype == SDRHIP_T_CS16 && n == 8 && a.first_rel == 0 && N % 8 == 0 && in_stride % 4 == 0 &&
                       (reinterpret_cast<uintptr_t>(in_dev) & 15) == 0;   // whole groups, no carry in or out
    if (fast8) {
      const unsigned bq = (unsigned)std::min<size_t>(ceil_div((size_t)2 * no, (size_t)TPB), 8192);
      hipLaunchKernelGGL(subsample8_cs16_kernel, dim3(bq, C), block, 0, ctx->stream, a);
    } else if (dtype == SDRHIP_T_CS16) hipLaunchKernelGGL(subsample_cs16_kernel, grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL(subsample_cf32_kernel, grid, block, 0, ctx->stream, a);
    SDRHIP_CHECK_HIP(hipGetLastError());
    par ^= 1; n0 += N;
    if (n_out) *n_out = no;
  }
};


struct sdrhip_deemph {
  sdrhip_ctx *ctx = nullptr;
  int alpha = 1, C = 1;
  size_t max_in = 0;
  DevBuf<short> avg;
  DevBuf<short> stage_in, stage_out;
  // which kernel a call of N samples per channel runs: 0 copy (alpha = 1), 1 one lane per channel, 2 P = 2^lgP lanes per
  // channel (long rows of a filter that forgets fast: the run-in is 16 alpha samples — measured on noise-like rows, two
  // runs meet within about 12 alpha — in whole groups of 64; P is the largest of 32 … 4 whose segments are at least as
  // long as the run-in), 3 the LDS-tiled kernel of rounds 1-2 (SDRHIP_DEEMPH_TILED, tests)
  // test / timing hooks, read once at create: SDRHIP_DEEMPH_TILED; SDRHIP_DEEMPH_SPEC (0 = the one-lane kernel, else P;
  // -1: not set); SDRHIP_DEEMPH_WC (the run-in in groups of 64 samples; 0: every guess is checked cold; -1: not set)
  bool env_tiled = false;
  int env_spec = -1, env_wc = -1;
  void read_env() {
    env_tiled = getenv("SDRHIP_DEEMPH_TILED") != nullptr;
    if (const char *e = getenv("SDRHIP_DEEMPH_SPEC")) env_spec = std::max(0, atoi(e));
    if (const char *e = getenv("SDRHIP_DEEMPH_WC")) env_wc = std::max(0, atoi(e));
  }
  int plan(size_t N, int *lgP_out, int *wc_out) const {
    if (env_tiled) return 3;
    if (alpha == 1) return 0;
    int lgP = 0;
    const int wc = (env_wc >= 0 ? env_wc : (int)ceil_div((size_t)16 * alpha, (size_t)64)) * DE_PF, n8 = (int)(N / 8);
    if (alpha <= 32 && env_spec != 0)
      for (int l = 5; l >= 2 && !lgP; l--) if ((n8 + (1 << l) - 1) >> l >= wc) lgP = l;
    if (env_spec > 1 && n8 >= env_spec) { lgP = 0; while ((2 << lgP) <= env_spec && lgP < 6) lgP++; }
    if (lgP_out) *lgP_out = lgP;
    if (wc_out) *wc_out = wc;
    return lgP ? 2 : 1;
  }
  void launch(const short *in_dev, size_t N, size_t in_stride, short *out_dev, size_t out_stride) {
    ctx->use();
    if (N == 0) return;
    DeemphArgs a;
    a.in = in_dev; a.in_stride = (long)in_stride; a.out = out_dev; a.out_stride = (long)out_stride;
    a.N = (int)N; a.C = C; a.alpha = alpha; a.avg = avg.p;
    a.magic = (unsigned)((1ull << 32) / (unsigned)alpha) + 1u;   // (alpha = 1 never divides)
    int lgP = 0, wc = 0;
    const int k = plan(N, &lgP, &wc);
    if (k == 0)
      hipLaunchKernelGGL(deemph_i16_copy_kernel, dim3((unsigned)std::min<size_t>(ceil_div(N, (size_t)256), 64), C), dim3(256), 0, ctx->stream, a);
    else if (k == 2) {
      DeemphSpecArgs sa;
      sa.d = a; sa.lgP = lgP; sa.Lc = ((int)(N / 8) + (1 << lgP) - 1) >> lgP; sa.wc = wc;
      hipLaunchKernelGGL(deemph_i16_spec_kernel, dim3((unsigned)ceil_div((size_t)C << lgP, (size_t)256)), dim3(256), 0, ctx->stream, sa);
    } else if (k == 1)
      hipLaunchKernelGGL(deemph_i16_seq_kernel, dim3((unsigned)ceil_div((size_t)C, (size_t)64)), dim3(64), 0, ctx->stream, a);
    else
      hipLaunchKernelGGL(deemph_i16_kernel, dim3((unsigned)ceil_div((size_t)C, (size_t)DE_CH)), dim3(DE_CH), 0, ctx->stream, a);
    SDRHIP_CHECK_HIP(hipGetLastError());
  }
};

extern "C" {

int sdrhip_demod_create(sdrhip_ctx *ctx, int kind, int dtype, int channels, size_t max_in, int inplace_fm0,
                        sdrhip_demod **out) {
  return guarded([&] {
    SDRHIP_REQUIRE(ctx && out, SDRHIP_E_INVALID, "NULL argument");
    *out = nullptr;
    SDRHIP_REQUIRE(kind == SDRHIP_EPI_FM || kind == SDRHIP_EPI_AM || kind == SDRHIP_EPI_USB, SDRHIP_E_INVALID, "bad kind %d", kind);
    SDRHIP_REQUIRE(dtype == SDRHIP_T_CS16 || dtype == SDRHIP_T_CF32 || dtype == SDRHIP_T_CS8, SDRHIP_E_INVALID, "bad dtype %d", dtype);
    SDRHIP_REQUIRE(!(kind == SDRHIP_EPI_FM && dtype == SDRHIP_T_CF32), SDRHIP_E_UNSUPPORTED,
                   "FMDemod<float> does not exist in the reference (fast_atan2 has no float form)");
    SDRHIP_REQUIRE(!(dtype == SDRHIP_T_CS8 && kind != SDRHIP_EPI_FM), SDRHIP_E_UNSUPPORTED,
                   "complex<int8_t> input: only FMDemod<int8_t,int16_t> is implemented");
    SDRHIP_REQUIRE(channels >= 1 && channels <= 65535, SDRHIP_E_INVALID, "channels %d outside [1,65535]", channels);
    SDRHIP_REQUIRE(max_in >= 1 && max_in < (size_t(1) << 30), SDRHIP_E_SIZE, "max_in %zu outside [1,2^30)", max_in);
    ctx->use();
    sdrhip_demod *h = new sdrhip_demod;
    try {
      h->ctx = ctx; h->kind = kind; h->dtype = dtype; h->C = channels; h->max_in = max_in; h->fm0 = inplace_fm0 ? 1 : 0;
      for (int p = 0; p < 2; p++) { h->fm[p].alloc(channels); h->fm[p].zero(ctx->stream); }
      SDRHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    } catch (...) { delete h; throw; }
    *out = h;
  });
}

int sdrhip_demod_process_dev(sdrhip_demod *h, const void *in_dev, size_t n, size_t in_stride, void *out_dev,
                             size_t out_stride) {
  return guarded([&] {
    Range roctx_range("sdrhip_demod_process_dev");
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    SDRHIP_REQUIRE(n <= h->max_in, SDRHIP_E_SIZE, "n %zu > max_in %zu", n, h->max_in);
    if (n == 0) return;
    SDRHIP_REQUIRE(in_dev && out_dev, SDRHIP_E_INVALID, "NULL buffer");
    if (in_stride == 0) in_stride = n;
    if (out_stride == 0) out_stride = n;
    SDRHIP_REQUIRE(in_stride >= n && out_stride >= n, SDRHIP_E_SIZE, "stride smaller than n");
    require_disjoint(in_dev, in_stride, n, h->in_elem(), out_dev, out_stride, n, h->out_elem(), (size_t)h->C);
    h->launch(in_dev, n, in_stride, out_dev, out_stride);
  });
}

int sdrhip_demod_process(sdrhip_demod *h, const void *in_host, size_t n, size_t in_stride, void *out_host,
                         size_t out_stride) {
  return guarded([&] {
    Range roctx_range("sdrhip_demod_process");
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    SDRHIP_REQUIRE(n <= h->max_in, SDRHIP_E_SIZE, "n %zu > max_in %zu", n, h->max_in);
    if (n == 0) return;
    SDRHIP_REQUIRE(in_host && out_host, SDRHIP_E_INVALID, "NULL buffer");
    h->ctx->use();
    if (in_stride == 0) in_stride = n;
    if (out_stride == 0) out_stride = n;
    SDRHIP_REQUIRE(in_stride >= n && out_stride >= n, SDRHIP_E_SIZE, "stride smaller than n");
    const size_t ib = h->in_elem(), ob = h->out_elem();
    if (!h->stage_in.p) { h->stage_in.alloc((size_t)h->C * h->max_in * ib); h->stage_out.alloc((size_t)h->C * h->max_in * ob); }
    copy_h2d_rows(h->ctx, h->stage_in.p, n * ib, in_host, in_stride * ib, n * ib, h->C);
    if (h->kind == SDRHIP_EPI_FM && !h->fm0)   // index 0 is left as the caller's buffer had it
      copy_h2d_rows(h->ctx, h->stage_out.p, n * ob, out_host, out_stride * ob, ob, h->C);
    h->launch(h->stage_in.p, n, n, h->stage_out.p, n);
    copy_d2h_rows(h->ctx, out_host, out_stride * ob, h->stage_out.p, n * ob, n * ob, h->C);
    SDRHIP_CHECK_HIP(hipStreamSynchronize(h->ctx->stream));
  });
}

int sdrhip_demod_reset(sdrhip_demod *h) {
  return guarded([&] {
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    h->ctx->use();
    for (int p = 0; p < 2; p++) h->fm[p].zero(h->ctx->stream);
  });
}

int sdrhip_demod_destroy(sdrhip_demod *h) {
  return guarded([&] {
    if (!h) return;
    h->ctx->use();
    (void)hipStreamSynchronize(h->ctx->stream);
    delete h;
  });
}

int sdrhip_deemph_i16_create(sdrhip_ctx *ctx, int alpha, int channels, size_t max_in, sdrhip_deemph **out) {
  return guarded([&] {
    SDRHIP_REQUIRE(ctx && out, SDRHIP_E_INVALID, "NULL argument");
    *out = nullptr;
    SDRHIP_REQUIRE(alpha >= 1 && alpha <= 32767, SDRHIP_E_INVALID, "alpha %d outside [1,32767]", alpha);
    SDRHIP_REQUIRE(channels >= 1 && channels <= (1 << 20), SDRHIP_E_INVALID, "channels %d out of range", channels);
    SDRHIP_REQUIRE(max_in >= 1 && max_in < (size_t(1) << 30), SDRHIP_E_SIZE, "max_in %zu outside [1,2^30)", max_in);
    ctx->use();
    sdrhip_deemph *h = new sdrhip_deemph;
    try {
      h->ctx = ctx; h->alpha = alpha; h->C = channels; h->max_in = max_in; h->read_env();
      h->avg.alloc(channels); h->avg.zero(ctx->stream);
      SDRHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    } catch (...) { delete h; throw; }
    *out = h;
  });
}

int sdrhip_deemph_i16_process_dev(sdrhip_deemph *h, const int16_t *in_dev, size_t n, size_t in_stride, int16_t *out_dev,
                                  size_t out_stride) {
  return guarded([&] {
    Range roctx_range("sdrhip_deemph_i16_process_dev");
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    SDRHIP_REQUIRE(n <= h->max_in, SDRHIP_E_SIZE, "n %zu > max_in %zu", n, h->max_in);
    if (n == 0) return;
    SDRHIP_REQUIRE(in_dev && out_dev, SDRHIP_E_INVALID, "NULL buffer");
    if (in_stride == 0) in_stride = n;
    if (out_stride == 0) out_stride = n;
    SDRHIP_REQUIRE(in_stride >= n && out_stride >= n, SDRHIP_E_SIZE, "stride smaller than n");
    require_disjoint(in_dev, in_stride, n, 2, out_dev, out_stride, n, 2, (size_t)h->C);
    h->launch(in_dev, n, in_stride, out_dev, out_stride);
  });
}

int sdrhip_deemph_i16_process(sdrhip_deemph *h, const int16_t *in_host, size_t n, size_t in_stride, int16_t *out_host,
                              size_t out_stride) {
  return guarded([&] {
    Range roctx_range("sdrhip_deemph_i16_process");
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    SDRHIP_REQUIRE(n <= h->max_in, SDRHIP_E_SIZE, "n %zu > max_in %zu", n, h->max_in);
    if (n == 0) return;
    SDRHIP_REQUIRE(in_host && out_host, SDRHIP_E_INVALID, "NULL buffer");
    h->ctx->use();
    if (in_stride == 0) in_stride = n;
    if (out_stride == 0) out_stride = n;
    SDRHIP_REQUIRE(in_stride >= n && out_stride >= n, SDRHIP_E_SIZE, "stride smaller than n");
    if (!h->stage_in.p) { h->stage_in.alloc((size_t)h->C * h->max_in); h->stage_out.alloc((size_t)h->C * h->max_in); }
    copy_h2d_rows(h->ctx, h->stage_in.p, n * 2, in_host, in_stride * 2, n * 2, h->C);
    h->launch(h->stage_in.p, n, n, h->stage_out.p, n);
    copy_d2h_rows(h->ctx, out_host, out_stride * 2, h->stage_out.p, n * 2, n * 2, h->C);
    SDRHIP_CHECK_HIP(hipStreamSynchronize(h->ctx->stream));
  });
}

int sdrhip_deemph_i16_kernel_names(sdrhip_deemph *h, size_t n, char *buf, size_t len) {
  return guarded([&] {
    SDRHIP_REQUIRE(h && buf && len, SDRHIP_E_INVALID, "NULL argument");
    static const char *const nm[4] = {"deemph_i16_copy_kernel", "deemph_i16_seq_kernel", "deemph_i16_spec_kernel", "deemph_i16_kernel"};
    snprintf(buf, len, "%s", nm[h->plan(n ? n : h->max_in, nullptr, nullptr)]);
  });
}

int sdrhip_deemph_i16_reset(sdrhip_deemph *h) {
  return guarded([&] {
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    h->ctx->use();
    h->avg.zero(h->ctx->stream);
  });
}

int sdrhip_deemph_i16_destroy(sdrhip_deemph *h) {
  return guarded([&] {
    if (!h) return;
    h->ctx->use();
    (void)hipStreamSynchronize(h->ctx->stream);
    delete h;
  });
}

int sdrhip_subsample_create(sdrhip_ctx *ctx, int dtype, size_t n, int channels, size_t max_in, sdrhip_subsample **out) {
  return guarded([&] {
    SDRHIP_REQUIRE(ctx && out, SDRHIP_E_INVALID, "NULL argument");
    *out = nullptr;
    SDRHIP_REQUIRE(dtype == SDRHIP_T_CS16 || dtype == SDRHIP_T_CF32, SDRHIP_E_INVALID, "bad dtype %d", dtype);
    SDRHIP_REQUIRE(n >= 1 && n <= 46340, SDRHIP_E_UNSUPPORTED, "n %zu outside [1,46340]", n);
    SDRHIP_REQUIRE(channels >= 1 && channels <= 65535, SDRHIP_E_INVALID, "channels %d outside [1,65535]", channels);
    SDRHIP_REQUIRE(max_in >= 1 && max_in < (size_t(1) << 30), SDRHIP_E_SIZE, "max_in %zu outside [1,2^30)", max_in);
    ctx->use();
    sdrhip_subsample *h = new sdrhip_subsample;
    try {
      h->ctx = ctx; h->dtype = dtype; h->n = n; h->C = channels; h->max_in = max_in; h->max_out = max_in / n + 1;
      for (int p = 0; p < 2; p++) { h->acc[p].alloc((size_t)channels * 8); h->acc[p].zero(ctx->stream); }
      SDRHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    } catch (...) { delete h; throw; }
    *out = h;
  });
}

int sdrhip_subsample_out_count(sdrhip_subsample *h, size_t n_in, size_t *n_out) {
  return guarded([&] {
    SDRHIP_REQUIRE(h && n_out, SDRHIP_E_INVALID, "NULL argument");
    *n_out = h->out_count(n_in);
  });
}

int sdrhip_subsample_process_dev(sdrhip_subsample *h, const void *in_dev, size_t n_in, size_t in_stride,
                                 void *out_dev, size_t out_stride, size_t *n_out) {
  return guarded([&] {
    Range roctx_range("sdrhip_subsample_process_dev");
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    SDRHIP_REQUIRE(n_in <= h->max_in, SDRHIP_E_SIZE, "n_in %zu > max_in %zu", n_in, h->max_in);
    if (n_in == 0) { if (n_out) *n_out = 0; return; }
    SDRHIP_REQUIRE(in_dev && out_dev, SDRHIP_E_INVALID, "NULL buffer");
    if (in_stride == 0) in_stride = n_in;
    SDRHIP_REQUIRE(in_stride >= n_in, SDRHIP_E_SIZE, "in_stride %zu < n_in %zu", in_stride, n_in);
    if (out_stride == 0) out_stride = h->out_count(n_in);
    require_disjoint(in_dev, in_stride, n_in, h->elem(), out_dev, out_stride, h->out_count(n_in), h->elem(), (size_t)h->C);
    h->launch(in_dev, n_in, in_stride, out_dev, out_stride, n_out);
  });
}

int sdrhip_subsample_process(sdrhip_subsample *h, const void *in_host, size_t n_in, size_t in_stride,
                             void *out_host, size_t out_stride, size_t *n_out) {
  return guarded([&] {
    Range roctx_range("sdrhip_subsample_process");
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    SDRHIP_REQUIRE(n_in <= h->max_in, SDRHIP_E_SIZE, "n_in %zu > max_in %zu", n_in, h->max_in);
    if (n_in == 0) { if (n_out) *n_out = 0; return; }
    SDRHIP_REQUIRE(in_host && out_host, SDRHIP_E_INVALID, "NULL buffer");
    h->ctx->use();
    if (in_stride == 0) in_stride = n_in;
    const size_t no = h->out_count(n_in);
    if (out_stride == 0) out_stride = no;
    SDRHIP_REQUIRE(out_stride >= no, SDRHIP_E_SIZE, "out_stride %zu < outputs %zu", out_stride, no);
    const size_t eb = h->elem();
    if (!h->stage_in.p) { h->stage_in.alloc((size_t)h->C * h->max_in * eb); h->stage_out.alloc((size_t)h->C * h->max_out * eb); }
    copy_h2d_rows(h->ctx, h->stage_in.p, n_in * eb, in_host, in_stride * eb, n_in * eb, h->C);
    size_t produced = 0;
    h->launch(h->stage_in.p, n_in, n_in, h->stage_out.p, h->max_out, &produced);
    copy_d2h_rows(h->ctx, out_host, out_stride * eb, h->stage_out.p, h->max_out * eb, produced * eb, h->C);
    SDRHIP_CHECK_HIP(hipStreamSynchronize(h->ctx->stream));
    if (n_out) *n_out = produced;
  });
}

int sdrhip_subsample_reset(sdrhip_subsample *h) {
  return guarded([&] {
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    h->ctx->use();
    h->n0 = 0;
    for (int p = 0; p < 2; p++) h->acc[p].zero(h->ctx->stream);
  });
}

int sdrhip_subsample_destroy(sdrhip_subsample *h) {
  return guarded([&] {
    if (!h) return;
    h->ctx->use();
    (void)hipStreamSynchronize(h->ctx->stream);
    delete h;
  });
}

}  // extern "C"
