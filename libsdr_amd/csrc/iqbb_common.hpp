// iqbb_common.hpp — K1's shared device helpers: argument block, sample loads, the LUT rotation, the lane-owned
// decimation-group epilogue (group_sum / group_finish) that the general MFMA kernels (iqbb_i16.hip) and the cold phase
// of the hot kernels (iqbb_hot.hpp) both use. Reference arithmetic: src/baseband.hh:198-236, src/freqshift.hh:58-74,
// src/demod.hh:73-76,156-161,242-254, src/math.hh:31-40, src/autocast.hh:187-194.
#pragma once
#include "fm_phi.hpp"
#include "sdrhip_internal.hpp"

#include <cstdlib>

using namespace sdrhip;

namespace sdrhip {

constexpr int TPB = 256;       // threads per workgroup (4 waves)
constexpr int R = 8;           // consecutive input samples per lane
constexpr int TI = TPB * R;    // input samples per tile
constexpr int TAPC = 8;        // taps per unrolled chunk (order is zero-padded at the front to a multiple)
constexpr int MAX_ORDER = 2048;

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct IqbbArgs {
  const uint32_t *in; long in_stride;            // cs16 packed as one dword per sample (or cu8: one ushort, in_cu8)
  int in_cu8;                                    // input is complex<uint8> and AutoCast<cs16> is applied on load
  int in_real;                                   // real-input BaseBand<int16_t>: one int16 per sample, taps are raw (Kr, Ki) int32
  int i8;                                        // IQBaseBand<int8_t> (VALU kernel only): complex<int8> in and out, the frequency shift in int16 (see rotate)
  const uint32_t *hist_old; uint32_t *hist_new;  // C x HH samples preceding the call
  const int2 *acc_old; int2 *acc_new;            // partial box sum of the open group
  const short *fm_old; short *fm_new;            // FMDemod::_last_value
  const uint2 *taps;                             // OP x {pack(Kr,-Ki), pack(Ki,Kr)}
  const int2 *lut; uint32_t inc; int negative;
  int OP, HH, D, N;
  uint32_t n0_lo;   // absolute index of the call's first sample, low 32 bits (LUT phase)
  int base0_rel;    // index (relative to the call start) of the first sample of the first group
  int n_groups;     // groups touched by this call
  int n_out;        // groups that complete in this call (always the first n_out of them)
  int extra0;       // absolute sample 0 joins group 0 (src/baseband.hh:200,212: D+1 first window)
  int CG, OG, ovl;  // groups computed / emitted per tile; FM recomputes one leading group
  int CGr;          // CG rounded up to 4: ybuf[CGr] is followed by the FM angle cache [CGr]
  void *out; long out_stride; int epilogue;
  const v4i *tapfrag; int cre, cim;   // MFMA paths: tap fragments, 128*sum(a) per component
  unsigned ah_mask;   // bit s: the high-byte tap fragments of K step s are not all zero (small outer taps: |a| < 128)
  int lpg;          // path 3: lanes that share one box window
  int tiles, tpw;   // tiles per channel in this call; consecutive tiles walked by one workgroup (MFMA paths)
  int bt_hi;        // hot kernel's cold phase: beside tile 0, the tiles bt_hi .. tiles-1 hold cold slices
  int fix_lo, fix_hi;   // any-D forms with FM, channel-resident units: the slices [fix_lo, fix_hi) get their first angle difference
                        // completed at the END of the hot kernel (iqbb_hot.hpp); empty: by iqbb_fm_fixup_kernel behind it
};

// Arguments of the hot kernels (iqbb_hot.hpp): the persistent grid's work split; everything the cold phase needs
// beyond them comes from the IqbbArgs block passed beside it.
// the hot kernel's fifth "epilogue" (internal; sdrhip.h's are 0 ... 3): no division, no demodulator — partial box sums for
// iqbb_bigd_finish_kernel (decimations above 512: a group spans slices)
constexpr int HOT_EPI_PARTIAL = 4;

struct HotArgs {
  const void *in; long in_stride;       // cs16: one dword per sample; cu8: one ushort (stride in samples)
  void *out; long out_stride;
  const v4i *tapfrag; const int2 *lut;
  uint32_t inc, n0_lo; int negative;
  int base0_rel, OG, ovl;               // as IqbbArgs
  int t_lo, t_hi, tpw;                  // tiles [t_lo, t_hi); a work unit = tpw consecutive ones of a channel
  int G, U, dq, dr;                     // units per channel, units in all, (virtual workgroups) / G and % G (persistent grid)
  int N, n_out;                         // samples per channel in this call, groups emitted (slice_is_hot)
  int C;                                // channels (cold phase)
  int cre, cim;
  int D, GS, lpg_sh; float inv_d;       // any-D form: decimation, whole groups per slice (512 / D), log2 of the lanes per group team, (1 / D)(1 - 2^-20)
  short *philast; int philast_stride;   // any-D form with FM: slice (tile, w) leaves the angle of its last group in philast[c * stride + 4 * tile + w]
  int tiles_h;                          // any-D form: tiles of the call (4 slices of GS groups each)
  // any-D forms with FM whose units are NOT whole channels (few channels): neighbouring slices complete the angle difference
  // between them by a handshake through device memory instead of a second launch (iqbb_hot.hpp, hs_exchange). Entry
  // {seq << 32 | phi}: hs[c * hs_stride + sid + 1] = the angle of slice sid's LAST group, hs[(C + c) * hs_stride + sid + 1] =
  // the angle of its FIRST group; hs_seq: this call's number (an entry of another call never matches). nullptr: off.
  long long *hs; int hs_stride, hs_seq;
  // large-decimation form (HOT_EPI_PARTIAL): slices of 512 samples from the call's first sample on, whatever the groups; slice
  // sid of channel c leaves the sums of its (at most three) stretches between group boundaries in part[c * part_stride + 3 sid + k]
  int2 *part; int part_stride;
  int Dreal, base_real;                 // the plan's decimation, and the call-relative index its groups are counted from (boundaries: base_real + j Dreal, j >= 1)
  int fin_groups, fin_out, fin_epi;     // ... whole channels as units: the groups (touched / completed in this call) the workgroup finishes itself as its last step, the plan's demodulator; fin_groups = 0: iqbb_bigd_finish_kernel does
  // multi-buffer calls with FM at decimation 8 (sdrhip_iqbb_i16_process_dev_multi): the buffers' first outputs are the groups
  // mb_q1, mb_q1 + mb_p, ... <= mb_qlast of the long call (mb_p = 0: none). Where such a group and the one behind it lie in ONE hot
  // slice (lanes 1 ... 62), the slice writes FMDemod's per-buffer values itself (iqbb_hot.hpp, stageF); the host leaves the other
  // boundaries to iqbb_fm_multi_fixup_kernel. mb_magic = floor(2^32 / mb_p).
  int mb_q1, mb_p, mb_qlast; unsigned mb_magic;
  unsigned long long *stamps;           // diagnostic builds (-DK1_STAMPS) only
};

}  // namespace sdrhip

namespace {

// AutoCast< complex<int16_t> > on a complex<uint8_t> sample (reference src/autocast.hh:62,187-194): each byte is
// read as int8 and becomes (int16(b) - 127) << 8, i.e. low byte 0 and high byte (b + 129) mod 256
__device__ __forceinline__ uint32_t cast_cu8(uint32_t u16) {
  return (((u16 & 0xffu) + 129u) & 0xffu) << 8 | ((((u16 >> 8) & 0xffu) + 129u) & 0xffu) << 24;
}
__device__ __forceinline__ uint32_t raw_x(const IqbbArgs &a, int c, long rel) {   // 0 <= rel < N
  if (a.in_cu8) return cast_cu8(reinterpret_cast<const uint16_t *>(a.in)[(long)c * a.in_stride + rel]);
  if (a.in_real) return (uint32_t)(int)reinterpret_cast<const short *>(a.in)[(long)c * a.in_stride + rel];   // sign-extended
  if (a.i8) {   // complex<int8_t>: both bytes sign-extended to the packed (re, im) int16 pair the FIR works on
    const uint32_t u = reinterpret_cast<const uint16_t *>(a.in)[(long)c * a.in_stride + rel];
    return ((uint32_t)(int)(signed char)(u & 0xffu) & 0xffffu) | ((uint32_t)(int)(signed char)(u >> 8) << 16);
  }
  return a.in[(long)c * a.in_stride + rel];
}
__device__ __forceinline__ uint32_t load_x(const IqbbArgs &a, int c, int rel) {
  if (rel >= 0) return rel < a.N ? raw_x(a, c, rel) : 0u;
  const int h = a.HH + rel;
  return h >= 0 ? a.hist_old[(long)c * a.HH + h] : 0u;
}

__device__ __forceinline__ int dot2(uint32_t x, uint32_t k, int acc) {
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, x), __builtin_bit_cast(s16x2, k), acc, false);
}

__device__ __forceinline__ int mulw(int a, int b) { return (int)((unsigned)a * (unsigned)b); }

// FreqShiftBase<int16_t>::applyFrequencyShift at absolute index n (low 32 bits suffice).
// Full-rate 24-bit multiplies: only the low 15 bits of n*inc matter; |LUT| < 2^23 (checked at create,
// the reference's is <= 2^16) and r = S>>14 lies in [-2^17, 2^17), so v_mul_i32_i24's low 32 bits equal
// the reference's wrapping 32-bit products.
__device__ __forceinline__ int2 rotate(const IqbbArgs &a, const int2 *lut_s, int2 r, uint32_t n_lo) {
  if (a.i8) {
    // FreqShiftBase<int8_t> computes in complex<int16_t> (src/freqshift.hh:18-22, src/traits.hh:58-73): the FIR value
    // is converted (wrapped) to int16 at the call, the LUT is 2^8 * exp(..), the product wraps to int16 and is shifted
    // by Traits<int8_t>::shift = 8 (src/freqshift.hh:58-74, src/traits.cc:11)
    r.x = (short)r.x; r.y = (short)r.y;
    if (a.inc == 0) return r;
    uint32_t idx = (__umul24(n_lo & 32767u, a.inc & 32767u) & 32767u) >> 8;
    if (a.negative) idx = 127u - idx;
    const int2 L = lut_s[idx];
    int2 v;
    v.x = (int)(short)(mulw(L.x, r.x) - mulw(L.y, r.y)) >> 8;
    v.y = (int)(short)(mulw(L.x, r.y) + mulw(L.y, r.x)) >> 8;
    return v;
  }
  if (a.inc == 0) return r;
  uint32_t idx = (__umul24(n_lo & 32767u, a.inc & 32767u) & 32767u) >> 8;
  if (a.negative) idx = 127u - idx;
  const int2 L = lut_s[idx];
  int2 v;
  v.x = (int)((unsigned)__mul24(L.x, r.x) - (unsigned)__mul24(L.y, r.y)) >> 16;
  v.y = (int)((unsigned)__mul24(L.x, r.y) + (unsigned)__mul24(L.y, r.x)) >> 16;
  return v;
}

// libstdc++ complex<int32>::operator/=(complex<int32>(D,0)): (a*D)/(D*D), wrapping, truncating
__device__ __forceinline__ int box_div(int s, int D) {
  const int n = mulw(D, D);
  const int r = mulw(s, D);
  if (n == 0) return 0;
  if (r == (int)0x80000000 && n == -1) return r;
  return r / n;
}

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
  const int m = (int)((unsigned)mulw(re, re) + (unsigned)mulw(im, im));
  return (short)(int)sqrt((double)m);
}

__device__ __forceinline__ short usb_i16(int re, int im) { return (short)((re + im) / 2); }

// Decimations above 256 on the hot structure (iqbb_hot.hpp, PART): the hot kernel left, per slice of 512 samples, the sums
// of its stretches between group boundaries; one lane per group adds the stretches that are its own (a group spans
// D / 512 slices), the carry of the open group (src/baseband.hh:212-217: the window sum lives across buffers), divides
// (libstdc++'s wrapping complex division by (D, 0): box_div) and demodulates, with the call-border rules of the other
// kernels: the stream's sample 0 belongs to group 0, FMDemod's outputs 0 and 1 of a buffer, the states for the next call.
struct BigdArgs {
  const int2 *part; int part_stride;
  int D, base0_rel, N, n_groups, n_out, epi, C;
  const int2 *acc_old; int2 *acc_new;
  const short *fm_old; short *fm_new;
  void *out; long out_stride;
};
__device__ __forceinline__ void bigd_finish_group(const BigdArgs &a, int c, int q) {
  const int2 *pc = a.part + (long)c * a.part_stride;
  auto group_sum = [&](int g) {
    // the group's samples inside the call: [lo, hi) — group 0 takes everything in front of its first boundary
    const long lo = g == 0 ? 0 : (long)a.base0_rel + (long)g * a.D, hi = min((long)a.base0_rel + (long)(g + 1) * a.D, (long)a.N);
    int2 s = make_int2(0, 0);
    if (g == 0) s = a.acc_old[c];
    for (long sl = lo >> 9; sl <= (hi - 1) >> 9; sl++) {
      const long x0 = sl << 9;   // the slice's first sample: in group gf, the slice's stretch 0
      const int gf = x0 < a.base0_rel ? 0 : (int)((x0 - a.base0_rel) / a.D);
      const int2 v = pc[3 * sl + (g - gf)];
      s.x = (int)((unsigned)s.x + (unsigned)v.x); s.y = (int)((unsigned)s.y + (unsigned)v.y);
    }
    return s;
  };
  const int2 s = group_sum(q);
  const bool emits = q < a.n_out;
  if (q == a.n_groups - 1) a.acc_new[c] = emits ? make_int2(0, 0) : s;
  if (!emits) return;
  const int yr = (short)box_div(s.x, a.D), yi = (short)box_div(s.y, a.D);
  if (a.epi == SDRHIP_EPI_NONE) {
    reinterpret_cast<uint32_t *>(a.out)[(long)c * a.out_stride + q] = ((uint32_t)(uint16_t)yr) | ((uint32_t)(uint16_t)yi << 16);
  } else if (a.epi == SDRHIP_EPI_AM) {
    reinterpret_cast<short *>(a.out)[(long)c * a.out_stride + q] = am_i16(yr, yi);
  } else if (a.epi == SDRHIP_EPI_USB) {
    reinterpret_cast<short *>(a.out)[(long)c * a.out_stride + q] = usb_i16(yr, yi);
  } else {
    const int phi = fm_phi(yr, yi);
    short o;
    if (q == 0) o = (short)yr;                              // index 0 is never written by FMDemod (in place)
    else if (q == 1) o = (short)((int)a.fm_old[c] - phi);   // y[0] is never looked at: the previous call's last angle
    else {                                                  // (the group before: summed again — a group is a handful of loads)
      const int2 sp = group_sum(q - 1);
      o = (short)(fm_phi((short)box_div(sp.x, a.D), (short)box_div(sp.y, a.D)) - phi);
    }
    reinterpret_cast<short *>(a.out)[(long)c * a.out_stride + q] = o;
    if (q == a.n_out - 1 && a.n_out >= 2) a.fm_new[c] = (short)phi;
  }
}

// one decimation group is complete (or left open at the end of the call): carry, first-sample quirk,
// truncating division, state
__device__ __forceinline__ void finalize_group(const IqbbArgs &a, int c, const int2 *lut_s, uint32_t *ybuf, int ql, int q, int2 s, int D) {
  if (q == 0) {
    const int2 carry = a.acc_old[c];
    s.x = (int)((unsigned)s.x + (unsigned)carry.x);
    s.y = (int)((unsigned)s.y + (unsigned)carry.y);
    if (a.extra0) {   // absolute sample 0: one slow FIR evaluation per channel and stream start
      int er = 0, ei = 0;
      for (int i = 0; i < a.OP; i++) {
        const uint32_t x = load_x(a, c, -(a.OP - 1) + i);
        const uint2 k = a.taps[i];
        er = dot2(x, k.x, er); ei = dot2(x, k.y, ei);
      }
      const int2 v = rotate(a, lut_s, make_int2(er >> 14, ei >> 14), a.n0_lo);
      s.x = (int)((unsigned)s.x + (unsigned)v.x);
      s.y = (int)((unsigned)s.y + (unsigned)v.y);
    }
  }
  const bool emits = q < a.n_out;
  if (emits) {
    int yr = (short)box_div(s.x, D), yi = (short)box_div(s.y, D);
    if (a.i8) { yr = (signed char)yr; yi = (signed char)yi; }   // the int8 node's output type (kept sign-extended in ybuf)
    ybuf[ql] = ((uint32_t)(uint16_t)yr) | ((uint32_t)(uint16_t)yi << 16);
    if (a.epilogue == SDRHIP_EPI_FM) reinterpret_cast<int *>(ybuf + a.CGr)[ql] = fm_phi(yr, yi);   // angle cache
  }
  if (q == a.n_groups - 1) a.acc_new[c] = emits ? make_int2(0, 0) : s;
}

// store / demodulate the tile's outputs (ybuf complete), and let the channel's last tile roll the history
__device__ __forceinline__ void epilogue_and_roll(const IqbbArgs &a, int c, int tile, int tid, int q0, int groups_here,
                                                  const uint32_t *ybuf) {
  for (int ql = a.ovl + tid; ql < groups_here; ql += TPB) {
    const int j = q0 + ql;   // output index within this call
    if (j >= a.n_out) continue;
    const uint32_t y = ybuf[ql];
    const int yr = (short)(y & 0xffffu), yi = (short)(y >> 16);
    if (a.epilogue == SDRHIP_EPI_NONE) {
      if (a.i8) reinterpret_cast<uint16_t *>(a.out)[(long)c * a.out_stride + j] = (uint16_t)((yr & 0xff) | ((yi & 0xff) << 8));   // complex<int8_t>
      else reinterpret_cast<uint32_t *>(a.out)[(long)c * a.out_stride + j] = y;
    } else {
      short o;
      if (a.epilogue == SDRHIP_EPI_AM) o = am_i16(yr, yi);
      else if (a.epilogue == SDRHIP_EPI_USB) o = usb_i16(yr, yi);
      else {
        const int *phib = reinterpret_cast<const int *>(ybuf + a.CGr);
        const int phi = phib[ql];
        if (j == 0) o = a.i8 ? (short)((yr & 0xff) | ((yi & 0xff) << 8))   // FMDemod<int8_t,int16_t> in place: out[0] = the 2 bytes of in[0]
                             : (short)yr;             // index 0 is never written by FMDemod (in place)
        else o = (short)((j == 1 ? (int)a.fm_old[c] : phib[ql - 1]) - phi);   // y[0] is never looked at: the
                                                                              // previous call's last angle
        if (j == a.n_out - 1 && a.n_out >= 2) a.fm_new[c] = (short)phi;
      }
      reinterpret_cast<short *>(a.out)[(long)c * a.out_stride + j] = o;
    }
  }
  if (tile == a.tiles - 1) {
    for (int k = tid; k < a.HH; k += TPB) {
      const long qq = (long)a.N + k;   // index into concat(hist_old, in)
      a.hist_new[(long)c * a.HH + k] =
          qq < a.HH ? a.hist_old[(long)c * a.HH + qq] : raw_x(a, c, qq - a.HH);
    }
  }
}

// REAL = the real-input BaseBand<int16_t> (src/baseband.hh:425-460): the staged dwords are sign-extended real
// samples, a tap is a raw (Kr, Ki) int32 pair (Q16, up to 17 bits) and one v_mad_i32_i24 per component replaces
// the dot2 (its low 32 bits equal the reference's wrapping int32 product for |K| < 2^23); >>16 instead of >>14.
template <bool FAST8, bool REAL>
__global__ __launch_bounds__(TPB) void iqbb_i16_kernel(const IqbbArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int XS = TI + a.OP + 8;
  uint32_t *xs = smem;                                  // staged samples, x[tb-(OP-1) ...]
  int2 *lut_s = reinterpret_cast<int2 *>(smem + XS);    // 128 entries
  uint32_t *ybuf = smem + XS + 256;                     // CG packed cs16 results
  int2 *vbuf = reinterpret_cast<int2 *>(ybuf + 2 * a.CGr);  // generic path only: TI entries

  const int c = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
  const int q0 = tile * a.OG - a.ovl;    // first group (relative to the call's first group) of this tile
  const int tb = a.base0_rel + q0 * a.D; // call-relative index of the tile's first sample
  const int groups_here = min(a.CG, a.n_groups - q0);

  // ---- stage samples (history / input / zeros) and the LUT into LDS ---------------------------
  {
    const int first = tb - (a.OP - 1);
    const int need = min(XS, groups_here * a.D + a.OP + 8);
    for (int i = tid; i < need; i += TPB) xs[i] = load_x(a, c, first + i);
    if (tid < 128) lut_s[tid] = a.lut[tid];
  }
  __syncthreads();

  // ---- FIR at 8 consecutive samples per lane ---------------------------------------------------
  int2 gsum = make_int2(0, 0);
  if (R * tid < groups_here * a.D) {
    int sre[R], sim[R];
#pragma unroll
    for (int r = 0; r < R; r++) { sre[r] = 0; sim[r] = 0; }
    const uint4 *win = reinterpret_cast<const uint4 *>(xs + R * tid);
    uint32_t w[16];
    {
      const uint4 p0 = win[0], p1 = win[1];
      w[0] = p0.x; w[1] = p0.y; w[2] = p0.z; w[3] = p0.w;
      w[4] = p1.x; w[5] = p1.y; w[6] = p1.z; w[7] = p1.w;
    }
    const uint2 *__restrict__ tp = a.taps;
    for (int i0 = 0; i0 < a.OP; i0 += TAPC) {
      const uint4 p2 = win[i0 / 4 + 2], p3 = win[i0 / 4 + 3];
      w[8] = p2.x; w[9] = p2.y; w[10] = p2.z; w[11] = p2.w;
      w[12] = p3.x; w[13] = p3.y; w[14] = p3.z; w[15] = p3.w;
#pragma unroll
      for (int u = 0; u < TAPC; u++) {
        const uint2 k = tp[i0 + u];   // wave-uniform -> scalar loads
#pragma unroll
        for (int r = 0; r < R; r++) {
          if (REAL) {
            sre[r] = (int)((unsigned)__mul24((int)k.x, (int)w[u + r]) + (unsigned)sre[r]);
            sim[r] = (int)((unsigned)__mul24((int)k.y, (int)w[u + r]) + (unsigned)sim[r]);
          } else {
            sre[r] = dot2(w[u + r], k.x, sre[r]);
            sim[r] = dot2(w[u + r], k.y, sim[r]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 8; u++) w[u] = w[u + 8];
    }
    // ---- >>14, rotate, mask samples outside this call -----------------------------------------
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int rel = tb + R * tid + r;
      constexpr int FSH = REAL ? 16 : 14;   // Traits<int16_t>::shift vs the literal 14 of IQBaseBand (:235, :459)
      int2 v = rotate(a, lut_s, make_int2(sre[r] >> FSH, sim[r] >> FSH), a.n0_lo + (uint32_t)rel);
      const bool valid = (rel >= 0) && (rel < a.N);
      if (!valid) v = make_int2(0, 0);
      if (FAST8) {
        gsum.x = (int)((unsigned)gsum.x + (unsigned)v.x);
        gsum.y = (int)((unsigned)gsum.y + (unsigned)v.y);
      } else {
        vbuf[R * tid + r] = v;
      }
    }
  }
  if (!FAST8) __syncthreads();

  // ---- box average per group -----------------------------------------------------------------------
  for (int ql = tid; ql < groups_here; ql += TPB) {
    const int q = q0 + ql;
    if (q < 0) continue;                      // tile 0's overlap slot precedes the call
    int2 s;
    if (FAST8) {
      s = gsum;                               // lane == group
    } else {
      s = make_int2(0, 0);
      for (int k = 0; k < a.D; k++) {
        const int2 v = vbuf[ql * a.D + k];
        s.x = (int)((unsigned)s.x + (unsigned)v.x);
        s.y = (int)((unsigned)s.y + (unsigned)v.y);
      }
    }
    finalize_group(a, c, lut_s, ybuf, ql, q, s, a.D);
  }
  __syncthreads();
  epilogue_and_roll(a, c, tile, tid, q0, groups_here, ybuf);
}

// =================================================================================================
// MFMA formulation (D == 8): the FIR as a block-Toeplitz int8 GEMM on the matrix cores.
//
//   Dmat[m = (t, comp)][n = block] = sum_k TapT[m][k] * U[k][n]
// A block is 16 consecutive samples of the channel, a wave owns 32 consecutive blocks (512 samples);
// U[k][n] is element k of the block's window in the interleaved (re,im) int16 element stream and
// TapT[m][k] = a_comp[k - 2t] the Toeplitz matrix of the interleaved tap vectors (re: Kr,-Ki ...;
// im: Ki,Kr ...). int16 x int16 products are made exact on v_mfma_i32_32x32x32_i8 by byte planes:
//   u = 256*uh + ul' + 128 (uh = u>>8, ul' = (u&255)-128),   a = 256*ah + al (al in [-128,127])
//   S = 65536*sum(ah*uh) + 256*sum(ah*ul' + al*uh) + sum(al*ul') + 128*sum(a)      (mod 2^32)
// i.e. 4 MFMAs per 32-deep K step into 3 accumulators; int32 ring arithmetic makes the recombination
// bit-exact. The tap fragments are wave-invariant and live in LDS ([S][2][64] x 16 B, fetched once per
// workgroup), the sample planes are staged once per tile into LDS and read as conflict-free 16-byte rows.
// Result layout (32x32 C/D map): lane (n = l&31, h = l>>5), register r -> comp = r&1,
// t = ((r&3)>>1) + 4*(r>>2) + 2h: a lane holds (re,im) pairs of 8 samples of its block, 4 per decimation
// group; the other 4 sit in lane l^32.
// =================================================================================================
constexpr int MF_BLK = 16;    // samples per block (one column)

// full-rate 24-bit integer multiplies as instructions: the compiler keeps explicit sign-extension code around
// v_mad_i32_i24 operands it cannot prove to be 24-bit, which these operands (LUT entries, FIR results) are
__device__ __forceinline__ int mul24a(int x, int y) { int d; asm("v_mul_i32_i24 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; }
__device__ __forceinline__ int mad24a(int x, int y, int z) { int d; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z)); return d; }

__device__ __forceinline__ unsigned mulu24a(unsigned x, unsigned y) { unsigned d; asm("v_mul_u32_u24 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; }
__device__ __forceinline__ int sub32(int x, int y) { return (int)((unsigned)x - (unsigned)y); }
// acc + (x >> 16): SDWA picks the sign-extended high half of x, so the shift of the rotation and the box-sum add are one instruction
__device__ __forceinline__ int add_hi16(int x, int acc) {
  int d;
  asm("v_add_u32_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(d) : "v"(x), "v"(acc));
  return d;
}
// (short)trunc(s / 8) for |s| < 2^29
__device__ __forceinline__ int div8_i16(int s) {
  const int t = (int)((unsigned)s + __builtin_amdgcn_ubfe((unsigned)s, 29, 3));
  return __builtin_amdgcn_sbfe(t, 3, 16);
}
// Lane (n, h) of a wave (n = l & 31, h = l >> 5) owns group 2n + h: the value of the previous group, 2n + h - 1, sits in
// lane (n, 0) for h = 1 and in lane (n - 1, 1) for h = 0 (lane 0 gets lane 63's: its own group is the wave's overlap slot).
__device__ __forceinline__ int prev_group_value(int v, int h) {
  // v_permlane32_swap: lanes 32..63 of the first operand <-> lanes 0..31 of the second; with both = v the first
  // result carries the lower half's values in both halves, the second the upper half's
  const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  const int hi_shr = __builtin_amdgcn_update_dpp(0, (int)sw[1], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  return h ? (int)sw[0] : hi_shr;
}

// Path 1's per-lane epilogue core: the lane's 8 samples (accumulator registers 2j / 2j+1 = re / im of sample j, planes hh,
// mid, ll) -> recombine -> >>14 -> rotate by LUT[idx(n)] -> sum of the (product >> 16) = the decimation window's sum.
// The rotation table sits at LDS offset 0 (dynamic LDS starts there: the kernel has no static LDS), so a table
// read's address is the byte offset itself.
// TBL: the table's layout at LDS offset 0 — 0: 128 x {Lx, Ly} in table order (a negative shift reads entry 127 - i);
// 1: the same stored reversed for a negative shift; 2: 16-byte entries {Lx, Ly, -Ly, 0}, stored reversed (hot kernel)
template <bool ROT, bool CU8, bool EDGE, int TBL = 0, int FSH = 14, bool I8 = false>   // FSH: the FIR's right shift (16: real-input BaseBand); I8: IQBaseBand<int8_t> on one byte plane
__device__ __forceinline__ int2 group_sum(const IqbbArgs &a, const v16i &acc_hh, const v16i &acc_mid, const v16i &acc_ll, int rel0) {
  typedef int v2i __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) const v2i lds_v2i;
  int2 L[8];
  if (ROT) {   // 8 independent table reads in flight while the accumulators are recombined
    // phase counter of the lane's first sample; only its low 15 bits matter, so a 24-bit multiply is exact enough
    const uint32_t cnt0 = mulu24a(a.n0_lo + (uint32_t)rel0, a.inc);
    const uint32_t negx = (TBL == 0 && a.negative) ? (127u << 3) : 0u;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t cj = cnt0 + (uint32_t)j * a.inc;   // j * inc: wave-uniform
      const uint32_t off = TBL == 2 ? ((cj >> 4) & (127u << 4)) : ((cj >> 5) & (127u << 3)) ^ negx;
      const v2i e = *reinterpret_cast<lds_v2i *>((uintptr_t)off);
      L[j] = make_int2(e.x, e.y);
    }
  }
  int2 sum = make_int2(0, 0);
#pragma unroll
  for (int j = 0; j < 8; j++) {
    // two v_lshl_add_u32 per component; the empty asm keeps the compiler from re-associating into 2 shifts + add3.
    // (The first level must stay compiler-generated code: it reads MFMA results, and only the compiler pads the
    // MFMA -> VALU read hazard; an asm v_lshl_add_u32 there reads stale accumulators.)
    unsigned tre = ((unsigned)acc_hh[2 * j] << 8) + (unsigned)acc_mid[2 * j];
    unsigned tim = ((unsigned)acc_hh[2 * j + 1] << 8) + (unsigned)acc_mid[2 * j + 1];
    int rr, ri;
    if (I8) {    // S = t exactly; wrapped to int16 after the shift (src/baseband.hh:206)
      rr = (short)((int)tre >> 14); ri = (short)((int)tim >> 14);
    } else if (CU8) {   // S = t << 8 exactly
      rr = (int)(tre << 8) >> FSH; ri = (int)(tim << 8) >> FSH;
    } else {
      asm("" : "+v"(tre)); asm("" : "+v"(tim));
      rr = (int)((tre << 8) + (unsigned)acc_ll[2 * j]) >> FSH; ri = (int)((tim << 8) + (unsigned)acc_ll[2 * j + 1]) >> FSH;
    }
    if (EDGE) { const int rel = rel0 + j; if (rel < 0 || rel >= a.N) { rr = 0; ri = 0; } }   // outside the call: r = 0 -> v = 0
    if (ROT) {
      const int x = sub32(mul24a(L[j].x, rr), mul24a(L[j].y, ri));
      const int y = mad24a(L[j].x, ri, mul24a(L[j].y, rr));
      if (I8) { sum.x += (int)(short)x >> 8; sum.y += (int)(short)y >> 8; }   // (the product wraps to int16, Traits<int8_t>::shift = 8)
      else { sum.x = add_hi16(x, sum.x); sum.y = add_hi16(y, sum.y); }   // += (x >> 16): one SDWA add each
    } else {
      sum.x = (int)((unsigned)sum.x + (unsigned)rr); sum.y = (int)((unsigned)sum.y + (unsigned)ri);
    }
  }
  return sum;
}

// per byte (b + 129) mod 256: the high byte AutoCast< complex<int16_t> > gives a complex<uint8_t> component
__device__ __forceinline__ uint32_t add129_bytes(uint32_t x) {
  const uint32_t y = x ^ 0x80808080u;   // + 128
  return ((y & 0x7f7f7f7fu) + 0x01010101u) ^ (y & 0x80808080u);   // + 1 without carries between bytes
}

// CU8: the input is complex<uint8_t> (SDRHIP_IN_CU8). After AutoCast every sample is 256 * uh exactly, so the low
// byte plane and both of its products vanish: S = 65536 * sum(ah*uh) + 256 * sum(al*uh) — two MFMAs per K step into
// two accumulators, one plane to stage, read and keep in LDS, 2 bytes per sample from HBM; 5 waves per SIMD fit.

// Path 1's group epilogue: lane (n, h) holds the window sum of group glw = 2n + h of its wave; carry / first-sample
// quirk for the call's first group, truncating division by 8, state for the next call, demodulator, store.
__device__ __forceinline__ void group_finish(const IqbbArgs &a, const int2 *lut_s, int c, int n, int h, int gw, int q0,
                                             int groups_here, int2 sum) {
  const int glw = 2 * n + h;
  const int ql = gw + glw, q = q0 + ql;   // q = output index within the call when the group completes
  const bool live = (ql < groups_here) && (q >= 0);
  if (q0 + gw <= 0 && live && q == 0) {   // (scalar test first: only the wave that holds the call's first group)
    const int2 carry = a.acc_old[c];
    sum.x = (int)((unsigned)sum.x + (unsigned)carry.x);
    sum.y = (int)((unsigned)sum.y + (unsigned)carry.y);
    if (a.extra0) {   // absolute sample 0: one slow FIR evaluation per channel and stream start
      int er = 0, ei = 0;
      for (int i = 0; i < a.OP; i++) {
        const uint32_t x = load_x(a, c, -(a.OP - 1) + i);
        const uint2 k = a.taps[i];
        er = dot2(x, k.x, er); ei = dot2(x, k.y, ei);
      }
      const int2 v = rotate(a, lut_s, make_int2(er >> 14, ei >> 14), a.n0_lo);
      sum.x = (int)((unsigned)sum.x + (unsigned)v.x);
      sum.y = (int)((unsigned)sum.y + (unsigned)v.y);
    }
  }
  const bool own = live && (glw >= a.ovl);            // the FM overlap slot belongs to the previous wave / tile
  const bool emits = live && (q < a.n_out);
  // libstdc++'s (s*8)/(8*8) (src/baseband.hh:214): |s| <= 9 * 2^17 here (a window of 16-bit rotated values, or of
  // 18-bit FIR values when there is no shift), so nothing wraps and it is trunc(s / 8): bias 7 for negative sums
  // (bits 31..29 of s), arithmetic shift, and the int16 wrap of the assignment in the same bit-field extract
  int yr = div8_i16(sum.x), yi = div8_i16(sum.y);
  if (a.i8) { yr = (signed char)yr; yi = (signed char)yi; }   // (IQBaseBand<int8_t>: the node's output type)
  if (own && q == a.n_groups - 1) a.acc_new[c] = emits ? make_int2(0, 0) : sum;
  if (a.epilogue == SDRHIP_EPI_NONE && a.i8) {
    if (own && emits) reinterpret_cast<uint16_t *>(a.out)[(long)c * a.out_stride + q] = (uint16_t)((yr & 0xff) | ((yi & 0xff) << 8));
  } else if (a.epilogue == SDRHIP_EPI_NONE) {
    if (own && emits) reinterpret_cast<uint32_t *>(a.out)[(long)c * a.out_stride + q] = ((uint32_t)(uint16_t)yr) | ((uint32_t)(uint16_t)yi << 16);
  } else {
    short o;
    if (a.epilogue == SDRHIP_EPI_AM) o = am_i16(yr, yi);
    else if (a.epilogue == SDRHIP_EPI_USB) o = usb_i16(yr, yi);
    else {
      const int phi = fm_phi(yr, yi);
      // previous group's angle: lane (n,0) for h=1, lane (n-1,1) for h=0 — one v_permlane32_swap (both halves'
      // values in both halves) and one wave_shr:1 DPP move, no LDS round trip
      const int prev = prev_group_value(phi, h);
      if (q == 0) o = a.i8 ? (short)((yr & 0xff) | ((yi & 0xff) << 8)) : (short)yr;   // index 0 is never written by FMDemod (in place; int8 chain: the 2 bytes of in[0])
      else o = (short)((q == 1 ? (int)a.fm_old[c] : prev) - phi);   // y[0] is never looked at
      if (own && emits && q == a.n_out - 1 && a.n_out >= 2) a.fm_new[c] = (short)phi;
    }
    if (own && emits) reinterpret_cast<short *>(a.out)[(long)c * a.out_stride + q] = o;
  }

}

}  // namespace
