// iqbb_i16.hip — K1: IQBaseBand<int16_t> (Q14 complex FIR at every input sample -> >>14 -> LUT
// rotate -> >>16 -> box average /D) with the FM / AM / USB demodulator fused as epilogue.
//
// Replaces (reference, file:line):
//   IQBaseBand<int16_t>::_process / _filter_ring      src/baseband.hh:198-236
//   FreqShiftBase<int16_t>::applyFrequencyShift        src/freqshift.hh:58-74
//   FMDemod<int16_t>::_process + fast_atan2            src/demod.hh:242-254, src/math.hh:31-40
//   AMDemod<int16_t>::process, USBDemod<int16_t>       src/demod.hh:73-76, :156-161
//
// Formulation: everything is a closed form of the ABSOLUTE sample index n
// since the last reset, so tiles of one channel are independent:
//   S[n]   = sum_i K[i] * x[n-(order-1)+i]          complex int32, wrapping (exact, associative)
//   r[n]   = S[n] >> 14
//   v[n]   = (LUT[idx(n)] * r[n]) >> 16,  idx = ((n*inc) mod 32768) >> 8   (127-idx if negative)
//   y[g]   = trunc( sum_{n in group g} v[n] / D ),  group g = { gD+1 .. (g+1)D }  (+ n=0 in group 0)
// The FIR is the hot loop: per input sample 2*order v_dot2_i32_i16 (packed (re,im) int16 sample
// against taps packed (Kr,-Ki) and (Ki,Kr)); each lane owns 8 consecutive samples and slides a
// 16-sample register window over an LDS-staged tile; taps arrive through the scalar cache.
#include "sdrhip_internal.hpp"

using namespace sdrhip;

namespace {

constexpr int TPB = 256;       // threads per workgroup (4 waves)
constexpr int R = 8;           // consecutive input samples per lane
constexpr int TI = TPB * R;    // input samples per tile
constexpr int TAPC = 8;        // taps per unrolled chunk (order is zero-padded at the front to a multiple)
constexpr int MAX_ORDER = 2048;

typedef short s16x2 __attribute__((ext_vector_type(2)));

struct IqbbArgs {
  const uint32_t *in; long in_stride;            // cs16 packed as one dword per sample
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
  void *out; long out_stride; int epilogue;
};

__device__ __forceinline__ uint32_t load_x(const IqbbArgs &a, int c, int rel) {
  if (rel >= 0) return rel < a.N ? a.in[(long)c * a.in_stride + rel] : 0u;
  const int h = a.HH + rel;
  return h >= 0 ? a.hist_old[(long)c * a.HH + h] : 0u;
}

__device__ __forceinline__ int dot2(uint32_t x, uint32_t k, int acc) {
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, x), __builtin_bit_cast(s16x2, k), acc, false);
}

__device__ __forceinline__ int mulw(int a, int b) { return (int)((unsigned)a * (unsigned)b); }

// FreqShiftBase<int16_t>::applyFrequencyShift at absolute index n (low 32 bits suffice)
__device__ __forceinline__ int2 rotate(const IqbbArgs &a, const int2 *lut_s, int2 r, uint32_t n_lo) {
  if (a.inc == 0) return r;
  uint32_t idx = ((n_lo * a.inc) & 32767u) >> 8;
  if (a.negative) idx = 127u - idx;
  const int2 L = lut_s[idx];
  int2 v;
  v.x = (int)((unsigned)mulw(L.x, r.x) - (unsigned)mulw(L.y, r.y)) >> 16;
  v.y = (int)((unsigned)mulw(L.x, r.y) + (unsigned)mulw(L.y, r.x)) >> 16;
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

// fast_atan2<int16_t,int16_t>(a, b) / 2   (src/math.hh:31-40, src/demod.hh:246)
__device__ __forceinline__ int fm_phi(int a, int b) {
  if (a == 0 && b == 0) return 0;
  const int aabs = a >= 0 ? a : -a;
  int angle;
  if (b >= 0) angle = 4096 - 4096 * (b - aabs) / (b + aabs);
  else angle = 12288 - 4096 * (b + aabs) / (aabs - b);
  const short at = (short)(a >= 0 ? angle : -angle);
  return (int)at / 2;
}

__device__ __forceinline__ short am_i16(int re, int im) {
  const int m = (int)((unsigned)mulw(re, re) + (unsigned)mulw(im, im));
  return (short)(int)sqrt((double)m);
}

__device__ __forceinline__ short usb_i16(int re, int im) { return (short)((re + im) / 2); }

template <bool FAST8>
__global__ __launch_bounds__(TPB) void iqbb_i16_kernel(const IqbbArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int XS = TI + a.OP + 8;
  uint32_t *xs = smem;                                  // staged samples, x[tb-(OP-1) ...]
  int2 *lut_s = reinterpret_cast<int2 *>(smem + XS);    // 128 entries
  uint32_t *ybuf = smem + XS + 256;                     // CG packed cs16 results
  int2 *vbuf = reinterpret_cast<int2 *>(ybuf + ((a.CG + 3) & ~3));  // generic path only: TI entries

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
          sre[r] = dot2(w[u + r], k.x, sre[r]);
          sim[r] = dot2(w[u + r], k.y, sim[r]);
        }
      }
#pragma unroll
      for (int u = 0; u < 8; u++) w[u] = w[u + 8];
    }
    // ---- >>14, rotate, mask samples outside this call -----------------------------------------
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int rel = tb + R * tid + r;
      int2 v = rotate(a, lut_s, make_int2(sre[r] >> 14, sim[r] >> 14), a.n0_lo + (uint32_t)rel);
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
      const int yr = box_div(s.x, a.D), yi = box_div(s.y, a.D);
      ybuf[ql] = ((uint32_t)(uint16_t)yr) | ((uint32_t)(uint16_t)yi << 16);
    }
    if (q == a.n_groups - 1) a.acc_new[c] = emits ? make_int2(0, 0) : s;
  }
  __syncthreads();

  // ---- epilogue: store / demodulate --------------------------------------------------------------
  for (int ql = a.ovl + tid; ql < groups_here; ql += TPB) {
    const int j = q0 + ql;   // output index within this call
    if (j >= a.n_out) continue;
    const uint32_t y = ybuf[ql];
    const int yr = (short)(y & 0xffffu), yi = (short)(y >> 16);
    if (a.epilogue == SDRHIP_EPI_NONE) {
      reinterpret_cast<uint32_t *>(a.out)[(long)c * a.out_stride + j] = y;
    } else {
      short o;
      if (a.epilogue == SDRHIP_EPI_AM) o = am_i16(yr, yi);
      else if (a.epilogue == SDRHIP_EPI_USB) o = usb_i16(yr, yi);
      else {
        const int phi = fm_phi(yr, yi);
        if (j == 0) o = (short)yr;             // index 0 is never written by FMDemod (in place)
        else {
          int prev;
          if (j == 1) prev = a.fm_old[c];      // y[0] is never looked at; last angle of the previous call
          else { const uint32_t yp = ybuf[ql - 1]; prev = fm_phi((short)(yp & 0xffffu), (short)(yp >> 16)); }
          o = (short)(prev - phi);
        }
        if (j == a.n_out - 1 && a.n_out >= 2) a.fm_new[c] = (short)phi;
      }
      reinterpret_cast<short *>(a.out)[(long)c * a.out_stride + j] = o;
    }
  }

  // ---- the last tile of a channel also rolls the FIR history forward ---------------------------
  if (tile == (int)gridDim.x - 1) {
    for (int k = tid; k < a.HH; k += TPB) {
      const long qq = (long)a.N + k;   // index into concat(hist_old, in)
      a.hist_new[(long)c * a.HH + k] =
          qq < a.HH ? a.hist_old[(long)c * a.HH + qq] : a.in[(long)c * a.in_stride + (qq - a.HH)];
    }
  }
}

}  // namespace

struct sdrhip_iqbb_i16 {
  sdrhip_ctx *ctx = nullptr;
  int order = 0, OP = 0, HH = 0, D = 1, C = 1, epi = 0, negative = 0;
  uint32_t inc = 0;
  size_t max_in = 0;
  uint64_t n0 = 0;
  int par = 0, par_fm = 0;
  int CG = 0, OG = 0, ovl = 0;
  bool fast8 = false;
  size_t lds_bytes = 0;
  DevBuf<uint2> taps;
  DevBuf<int2> lut;
  DevBuf<uint32_t> hist[2];
  DevBuf<int2> acc[2];
  DevBuf<short> fm[2];
  DevBuf<uint32_t> stage_in;
  DevBuf<uint32_t> stage_out;
  size_t max_out = 0;

  struct Geometry { uint64_t g_first; int n_groups, n_out, base0_rel, extra0; };
  Geometry geometry(size_t N) const {
    Geometry g{};
    const uint64_t D64 = (uint64_t)D, shift1 = D > 1 ? 1 : 0;
    auto group_of = [&](uint64_t n) -> uint64_t { return n < shift1 ? 0 : (n - shift1) / D64; };
    const uint64_t gf = group_of(n0), gl = group_of(n0 + N - 1);
    const uint64_t last_end = (gl + 1) * D64 - 1 + shift1;
    g.g_first = gf;
    g.n_groups = (int)(gl - gf + 1);
    g.n_out = g.n_groups - (last_end <= n0 + N - 1 ? 0 : 1);
    g.base0_rel = (int)((int64_t)(gf * D64 + shift1) - (int64_t)n0);
    g.extra0 = (n0 == 0 && shift1) ? 1 : 0;
    return g;
  }
  size_t out_elem_bytes() const { return epi == SDRHIP_EPI_NONE ? 4 : 2; }

  void launch(const uint32_t *in_dev, size_t N, size_t in_stride, void *out_dev, size_t out_stride, size_t *n_out) {
    ctx->use();
    if (N == 0) { if (n_out) *n_out = 0; return; }   // empty buffer: nothing moves (src/baseband.hh:200)
    const Geometry g = geometry(N);
    SDRHIP_REQUIRE(out_stride >= (size_t)g.n_out, SDRHIP_E_SIZE, "out_stride %zu < outputs %d", out_stride, g.n_out);
    IqbbArgs a;
    a.in = in_dev; a.in_stride = (long)in_stride;
    a.hist_old = hist[par].p; a.hist_new = hist[par ^ 1].p; a.HH = HH;
    a.acc_old = acc[par].p; a.acc_new = acc[par ^ 1].p;
    const bool fm_flip = (epi == SDRHIP_EPI_FM && g.n_out >= 2);
    a.fm_old = fm[par_fm].p; a.fm_new = fm[par_fm ^ 1].p;
    a.taps = taps.p; a.lut = lut.p; a.inc = inc; a.negative = negative;
    a.OP = OP; a.D = D; a.N = (int)N; a.n0_lo = (uint32_t)n0;
    a.base0_rel = g.base0_rel; a.n_groups = g.n_groups; a.n_out = g.n_out; a.extra0 = g.extra0;
    a.CG = CG; a.OG = OG; a.ovl = ovl;
    a.out = out_dev; a.out_stride = (long)out_stride; a.epilogue = epi;
    const int tiles = (int)ceil_div((size_t)g.n_groups, (size_t)OG);
    dim3 grid(tiles, C), block(TPB);
    if (fast8) hipLaunchKernelGGL(iqbb_i16_kernel<true>, grid, block, lds_bytes, ctx->stream, a);
    else hipLaunchKernelGGL(iqbb_i16_kernel<false>, grid, block, lds_bytes, ctx->stream, a);
    SDRHIP_CHECK_HIP(hipGetLastError());
    par ^= 1;
    if (fm_flip) par_fm ^= 1;
    n0 += N;
    if (n_out) *n_out = (size_t)g.n_out;
  }
};

extern "C" {

int sdrhip_iqbb_i16_create(sdrhip_ctx *ctx, const int32_t *taps, int order, const int32_t *lut, uint32_t lut_inc,
                           int negative, int decim, int channels, size_t max_in, int epilogue,
                           sdrhip_iqbb_i16 **out) {
  return guarded([&] {
    SDRHIP_REQUIRE(ctx && taps && lut && out, SDRHIP_E_INVALID, "NULL argument");
    *out = nullptr;
    SDRHIP_REQUIRE(order >= 1 && order <= MAX_ORDER, SDRHIP_E_UNSUPPORTED, "order %d outside [1,%d]", order, MAX_ORDER);
    SDRHIP_REQUIRE(decim >= 1, SDRHIP_E_INVALID, "decim %d < 1", decim);
    SDRHIP_REQUIRE(channels >= 1 && channels <= 65535, SDRHIP_E_INVALID, "channels %d outside [1,65535]", channels);
    SDRHIP_REQUIRE(max_in >= 1 && max_in < (size_t(1) << 30), SDRHIP_E_SIZE, "max_in %zu outside [1,2^30)", max_in);
    SDRHIP_REQUIRE(epilogue >= SDRHIP_EPI_NONE && epilogue <= SDRHIP_EPI_USB, SDRHIP_E_INVALID, "bad epilogue %d", epilogue);
    const int ovl = epilogue == SDRHIP_EPI_FM ? 1 : 0;
    const int CG = TI / decim;
    SDRHIP_REQUIRE(CG - ovl >= 1, SDRHIP_E_UNSUPPORTED, "decim %d too large (max %d)", decim, TI / (1 + ovl));
    for (int i = 0; i < 2 * order; i++)
      SDRHIP_REQUIRE(taps[i] >= -32767 && taps[i] <= 32767, SDRHIP_E_UNSUPPORTED,
                     "tap %d = %d does not fit the packed int16 path", i / 2, taps[i]);
    ctx->use();
    sdrhip_iqbb_i16 *h = new sdrhip_iqbb_i16;
    try {
      h->ctx = ctx; h->order = order; h->D = decim; h->C = channels; h->epi = epilogue;
      h->negative = negative ? 1 : 0; h->inc = lut_inc; h->max_in = max_in;
      h->OP = (int)ceil_div((size_t)order, (size_t)TAPC) * TAPC;
      h->HH = h->OP;   // one more than the FIR needs: reset(keep_history) must see the whole ring
      h->CG = CG; h->ovl = ovl; h->OG = CG - ovl;
      h->fast8 = (decim == R);
      const size_t XS = TI + h->OP + 8;
      h->lds_bytes = (XS + 256 + ((CG + 3) & ~3)) * 4 + (h->fast8 ? 0 : (size_t)TI * 8);
      SDRHIP_REQUIRE(h->lds_bytes <= 64 * 1024, SDRHIP_E_UNSUPPORTED, "LDS budget exceeded (%zu B)", h->lds_bytes);
      // taps: zero-padded at the FRONT (older samples) so that the newest sample still meets K[order-1]
      std::vector<uint2> tp(h->OP, make_uint2(0, 0));
      const int pad = h->OP - order;
      for (int i = 0; i < order; i++) {
        const int kr = taps[2 * i], ki = taps[2 * i + 1];
        tp[pad + i].x = ((uint32_t)(uint16_t)(int16_t)kr) | ((uint32_t)(uint16_t)(int16_t)(-ki) << 16);
        tp[pad + i].y = ((uint32_t)(uint16_t)(int16_t)ki) | ((uint32_t)(uint16_t)(int16_t)kr << 16);
      }
      h->taps.alloc(h->OP); h->taps.upload(tp.data(), h->OP, ctx->stream);
      h->lut.alloc(128); h->lut.upload(reinterpret_cast<const int2 *>(lut), 128, ctx->stream);
      for (int p = 0; p < 2; p++) {
        h->hist[p].alloc((size_t)channels * h->HH); h->hist[p].zero(ctx->stream);
        h->acc[p].alloc(channels); h->acc[p].zero(ctx->stream);
        h->fm[p].alloc(channels); h->fm[p].zero(ctx->stream);
      }
      h->max_out = max_in / decim + 2;
      SDRHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    } catch (...) { delete h; throw; }
    *out = h;
  });
}

int sdrhip_iqbb_i16_out_count(sdrhip_iqbb_i16 *h, size_t n_in, size_t *n_out) {
  return guarded([&] {
    SDRHIP_REQUIRE(h && n_out, SDRHIP_E_INVALID, "NULL argument");
    *n_out = n_in ? (size_t)h->geometry(n_in).n_out : 0;
  });
}

int sdrhip_iqbb_i16_process_dev(sdrhip_iqbb_i16 *h, const int16_t *in_dev, size_t n_in, size_t in_stride,
                                void *out_dev, size_t out_stride, size_t *n_out) {
  return guarded([&] {
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    SDRHIP_REQUIRE(n_in <= h->max_in, SDRHIP_E_SIZE, "n_in %zu > max_in %zu", n_in, h->max_in);
    if (n_in == 0) { if (n_out) *n_out = 0; return; }
    SDRHIP_REQUIRE(in_dev && out_dev, SDRHIP_E_INVALID, "NULL buffer");
    if (in_stride == 0) in_stride = n_in;
    SDRHIP_REQUIRE(in_stride >= n_in, SDRHIP_E_SIZE, "in_stride %zu < n_in %zu", in_stride, n_in);
    if (out_stride == 0) out_stride = (size_t)h->geometry(n_in).n_out;
    h->launch(reinterpret_cast<const uint32_t *>(in_dev), n_in, in_stride, out_dev, out_stride, n_out);
  });
}

int sdrhip_iqbb_i16_process(sdrhip_iqbb_i16 *h, const int16_t *in_host, size_t n_in, size_t in_stride,
                            void *out_host, size_t out_stride, size_t *n_out) {
  return guarded([&] {
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    SDRHIP_REQUIRE(n_in <= h->max_in, SDRHIP_E_SIZE, "n_in %zu > max_in %zu", n_in, h->max_in);
    if (n_in == 0) { if (n_out) *n_out = 0; return; }
    SDRHIP_REQUIRE(in_host && out_host, SDRHIP_E_INVALID, "NULL buffer");
    h->ctx->use();
    if (in_stride == 0) in_stride = n_in;
    const size_t no = (size_t)h->geometry(n_in).n_out;
    if (out_stride == 0) out_stride = no;
    SDRHIP_REQUIRE(out_stride >= no, SDRHIP_E_SIZE, "out_stride %zu < outputs %zu", out_stride, no);
    if (!h->stage_in.p) {
      h->stage_in.alloc((size_t)h->C * h->max_in);
      h->stage_out.alloc((size_t)h->C * h->max_out);
    }
    copy_h2d_rows(h->ctx, h->stage_in.p, n_in * 4, in_host, in_stride * 4, n_in * 4, h->C);
    const size_t eb = h->out_elem_bytes();
    size_t produced = 0;
    h->launch(h->stage_in.p, n_in, n_in, h->stage_out.p, h->max_out * 4 / eb, &produced);
    copy_d2h_rows(h->ctx, out_host, out_stride * eb, h->stage_out.p, h->max_out * 4, produced * eb, h->C);
    SDRHIP_CHECK_HIP(hipStreamSynchronize(h->ctx->stream));
    if (n_out) *n_out = produced;
  });
}

int sdrhip_iqbb_i16_reset(sdrhip_iqbb_i16 *h, int keep_history) {
  return guarded([&] {
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    h->ctx->use();
    hipStream_t st = h->ctx->stream;
    for (int p = 0; p < 2; p++) { h->acc[p].zero(st); h->fm[p].zero(st); }
    if (!keep_history) {
      for (int p = 0; p < 2; p++) h->hist[p].zero(st);
    } else if (h->n0 % (uint64_t)h->order != 0) {
      // IQBaseBand::_reconfigure resets _ring_offset but leaves the ring contents where they are
      // (src/baseband.hh:175-177), so the node afterwards reads the old ring ROTATED: with
      // P = (samples so far) mod order the apparent history, oldest first, is ring[1..order-1],
      // ring[i] = t[order-P+i] (i < P) or t[i-P] (i >= P), t = the last `order` samples in time order.
      const int order = h->order, HH = h->HH, P = (int)(h->n0 % (uint64_t)order);
      std::vector<uint32_t> old((size_t)h->C * HH), neu((size_t)h->C * HH, 0u);
      SDRHIP_CHECK_HIP(hipMemcpyAsync(old.data(), h->hist[h->par].p, old.size() * 4, hipMemcpyDeviceToHost, st));
      SDRHIP_CHECK_HIP(hipStreamSynchronize(st));
      for (int c = 0; c < h->C; c++) {
        const uint32_t *t = old.data() + (size_t)c * HH + (HH - order);
        uint32_t *d = neu.data() + (size_t)c * HH + (HH - (order - 1));
        for (int k = 0; k + 1 < order; k++) { const int i = k + 1; d[k] = i < P ? t[order - P + i] : t[i - P]; }
      }
      SDRHIP_CHECK_HIP(hipMemcpyAsync(h->hist[h->par].p, neu.data(), neu.size() * 4, hipMemcpyHostToDevice, st));
      SDRHIP_CHECK_HIP(hipStreamSynchronize(st));
    }
    h->n0 = 0;
  });
}

int sdrhip_iqbb_i16_destroy(sdrhip_iqbb_i16 *h) {
  return guarded([&] {
    if (!h) return;
    h->ctx->use();
    (void)hipStreamSynchronize(h->ctx->stream);
    delete h;
  });
}

}  // extern "C"
