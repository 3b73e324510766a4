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
#include "iqbb_common.hpp"
#include "iqbb_hot.hpp"

namespace {

template <int S, bool ROT, bool CU8>
__global__ __launch_bounds__(TPB, CU8 ? 5 : 4) void iqbb_i16_mfma_kernel(const IqbbArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int PLW = (2 * (TI + a.OP) + 64 + 31) / 32 * 8;    // dwords per byte plane
  // LDS: rotation table at offset 0 (a table read then needs no base add) | two plane pairs (double buffer:
  // tile i+1 is written while slower waves still read tile i; CU8: two single planes) | tap fragments [S][2][64]
  constexpr int NPL = CU8 ? 1 : 2;   // byte planes per buffer
  int2 *lut_s = reinterpret_cast<int2 *>(smem);
  uint32_t *planes = smem + 256;
  v4i *taps_s = reinterpret_cast<v4i *>(smem + 256 + 2 * NPL * PLW);

  const int c = blockIdx.y, tid = threadIdx.x;
  const int w = tid >> 6, l = tid & 63, n = l & 31, h = l >> 5;
  // tap fragments are wave-invariant: fetched once per workgroup into LDS and read per K step (keeping them in
  // 72 VGPRs halves the occupancy and measured slower)
  for (int i = tid; i < S * 2 * 64; i += TPB) taps_s[i] = a.tapfrag[i];
  if (tid < 128) lut_s[tid] = a.lut[tid];

  // Software pipeline over the tiles this workgroup walks: the global loads of tile i+1 are issued into
  // registers before the MFMA/epilogue work of tile i and written to the other LDS plane pair after it.
  constexpr int NQ = (TI + 16 * (S - 1) + 1 + 2 + 4 * TPB - 1) / (4 * TPB);   // sample quads per lane and tile (OP = 16(S-1)+1)
  struct __attribute__((packed, aligned(4))) Quad { uint32_t v[4]; };   // 16-byte load from a 4-byte aligned address
  Quad px[NQ];
  auto fetch = [&](int tile_) {
#ifdef K1_ABL_NOFETCH   // tuning ablation (results wrong): no global loads
    for (int k = 0; k < NQ; k++) for (int j = 0; j < 4; j++) px[k].v[j] = tile_ + k + j;
    return;
#endif
    const int q0_ = tile_ * a.OG - a.ovl;
    const int first = a.base0_rel + q0_ * 8 - (a.OP - 1);
    const int quads = (min(a.CG, a.n_groups - q0_) * 8 + a.OP + 4) / 4;
    const bool interior = first >= 0 && first + 4 * quads <= a.N;   // no history, no end of call
    if (CU8) {   // px[k].v[0..1] = the 8 high-plane bytes of 4 samples
      struct __attribute__((packed, aligned(2))) Oct { uint32_t v[2]; };   // 8-byte load from a 2-byte aligned address
      const uint16_t *src = reinterpret_cast<const uint16_t *>(a.in) + (long)c * a.in_stride + first;
#pragma unroll
      for (int k = 0; k < NQ; k++) {
        const int p = tid + k * TPB;
        if (p < quads) {
          if (interior) {
            const Oct o = *reinterpret_cast<const Oct *>(src + 4 * p);
            px[k].v[0] = add129_bytes(o.v[0]); px[k].v[1] = add129_bytes(o.v[1]);
          } else {
            const uint32_t x0 = load_x(a, c, first + 4 * p), x1 = load_x(a, c, first + 4 * p + 1);
            const uint32_t x2 = load_x(a, c, first + 4 * p + 2), x3 = load_x(a, c, first + 4 * p + 3);
            px[k].v[0] = __builtin_amdgcn_perm(x1, x0, 0x07050301u); px[k].v[1] = __builtin_amdgcn_perm(x3, x2, 0x07050301u);
          }
        }
      }
    } else if (interior && !a.in_cu8) {
      const uint32_t *src = a.in + (long)c * a.in_stride + first;
#pragma unroll
      for (int k = 0; k < NQ; k++) {
        const int p = tid + k * TPB;
        if (p < quads) px[k] = *reinterpret_cast<const Quad *>(src + 4 * p);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NQ; k++) {
        const int p = tid + k * TPB;
        if (p < quads) {
#pragma unroll
          for (int j = 0; j < 4; j++) px[k].v[j] = load_x(a, c, first + 4 * p + j);
        }
      }
    }
  };
  const int tile_end = min((int)(blockIdx.x + 1) * a.tpw, a.tiles);
  int tile = blockIdx.x * a.tpw;
  if (tile < tile_end) fetch(tile);
  const int OGw = 64 - a.ovl;   // groups a wave emits; with FM its first group only supplies the previous angle
  for (int it = 0; tile < tile_end; tile++, it++) {
    const int q0 = tile * a.OG - a.ovl;
    const int tb = a.base0_rel + q0 * 8;
    const int groups_here = min(a.CG, a.n_groups - q0);
    uint32_t *lo = planes + (it & 1) * NPL * PLW, *hi = lo + (NPL - 1) * PLW;

    // ---- stage: four samples -> 8 bytes of the low plane (offset to signed) and 8 of the high plane ----
    {
      const int quads = (groups_here * 8 + a.OP + 4) / 4;
#pragma unroll
      for (int k = 0; k < NQ; k++) {
        const int p = tid + k * TPB;
        if (p < quads) {
          // 16-byte chunks (8 samples) are de-interleaved by parity (even chunks in the first half of the
          // plane, odd in the second): a lane's K steps then walk consecutive chunks and the 16 lanes a
          // ds_read_b128 services together cover one contiguous 256-byte bank row instead of every other slot
          const int d = (((p >> 1) & 1) * (PLW >> 1)) + ((p >> 2) << 2) + ((p & 1) << 1);
          if (CU8) {
            *reinterpret_cast<uint2 *>(hi + d) = make_uint2(px[k].v[0], px[k].v[1]);
          } else {
            uint2 l2, h2;
            l2.x = __builtin_amdgcn_perm(px[k].v[1], px[k].v[0], 0x06040200u) ^ 0x80808080u;
            l2.y = __builtin_amdgcn_perm(px[k].v[3], px[k].v[2], 0x06040200u) ^ 0x80808080u;
            h2.x = __builtin_amdgcn_perm(px[k].v[1], px[k].v[0], 0x07050301u);
            h2.y = __builtin_amdgcn_perm(px[k].v[3], px[k].v[2], 0x07050301u);
            *reinterpret_cast<uint2 *>(lo + d) = l2;
            *reinterpret_cast<uint2 *>(hi + d) = h2;
          }
        }
      }
    }
    __syncthreads();   // the only barrier per tile: planes[it&1] complete; planes[(it+1)&1] were last read before it
    // The next tile's samples are only pulled towards L2 here — one dword per 64-byte line into a scratch
    // register — and loaded into registers after the K loop, so that 12 VGPRs of prefetch are not live across it.
    // The compiler does not know the asm is a load: `touch` stays tied to it until the explicit wait below.
    // The touch runs K1_TOUCH_AHEAD tiles ahead (default 2): a tile period is longer than an HBM round trip under
    // load, so the register loads after the K loop find their lines in L2 — with one tile of lead they still waited
    // on HBM, and a workgroup has only that one tile of loads in flight (measured: the kernel without any arithmetic
    // took 104 of the 160 us).
    uint32_t touch = 0;
    auto touch_tile = [&](int tile_) {
      if (tile_ >= tile_end) return;
      const int q0_ = tile_ * a.OG - a.ovl;
      constexpr int LINE = CU8 ? 32 : 16;   // samples per 64-byte line
      const long first = (long)a.base0_rel + (long)q0_ * 8 - (a.OP - 1) + (long)LINE * tid;   // one line per lane
#ifndef K1_ABL_NOFETCH
      if ((CU8 || !a.in_cu8) && first >= 0 && first < (long)a.N && LINE * tid < TI + a.OP + LINE)
#else
      if (false)
#endif
      {
        const void *pa = CU8 ? (const void *)((reinterpret_cast<const uint16_t *>(a.in) + (long)c * a.in_stride + first))
                             : (const void *)(a.in + (long)c * a.in_stride + first);
        pa = (const void *)((uintptr_t)pa & ~(uintptr_t)3);
        asm volatile("global_load_dword %0, %1, off" : "+v"(touch) : "v"(pa) : "memory");
      }
    };
#ifndef K1_TOUCH_AHEAD
#define K1_TOUCH_AHEAD 2
#endif
    if (it == 0) {
#pragma unroll
      for (int d = 1; d < K1_TOUCH_AHEAD; d++) touch_tile(tile + d);
    }
    touch_tile(tile + K1_TOUCH_AHEAD);
    const bool wave_has_work = (w * OGw + a.ovl < groups_here);
    if (!wave_has_work) {   // (a wave without groups in a ragged last tile)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(touch) : : "memory");
      if (tile + 1 < tile_end) fetch(tile + 1);
    }

    const int gw = w * OGw;   // this wave's first group within the tile
    if (gw + a.ovl < groups_here) {   // wave-uniform: the wave has at least one group of its own
      v16i acc_hh = {0}, acc_mid = {0}, acc_ll = {0};
      if (!CU8) {
#pragma unroll
        for (int r = 0; r < 16; r++) acc_ll[r] = (r & 1) ? a.cim : a.cre;   // + 128*sum(a) rides in as C
      }
      // chunk (16 B = 8 samples of one plane) gw + 2(n+s) + h of the tile, in the parity-split layout
      const int coff = ((gw + h) & 1) * (2 * PLW) + 16 * (((gw + h) >> 1) + n);
      const char *pl = reinterpret_cast<const char *>(lo) + coff;
      const char *ph = reinterpret_cast<const char *>(hi) + coff;
#ifdef K1_ABL_NOKLOOP   // tuning ablation (results wrong): no LDS operand reads, no MFMAs
      acc_mid[0] = *reinterpret_cast<const int *>(ph); acc_ll[1] = *reinterpret_cast<const int *>(pl);
#else
#pragma unroll
      for (int s = 0; s < S; s++) {
        // the outer taps of a windowed sinc are small: where every tap a K step touches has a zero high byte, its
        // two high-plane products are skipped (scalar branch on a mask made at create time; 4 of 9 steps for the
        // 127-tap north-star filter)
        const bool has_ah = (a.ah_mask >> s) & 1;
        const v4i uh = *reinterpret_cast<const v4i *>(ph + 16 * s);
        const v4i Al = taps_s[(2 * s + 1) * 64 + l];
#ifdef K1_ABL_NOMFMA    // tuning ablation (results wrong): the LDS operand reads stay, the matrix instructions go
        acc_mid[s] += uh.x ^ uh.y ^ uh.z ^ uh.w ^ Al.x ^ Al.y ^ Al.z ^ Al.w;
        { const v4i ul = *reinterpret_cast<const v4i *>(pl + 16 * s); acc_ll[s] += ul.x ^ ul.y ^ ul.z ^ ul.w; }
        if (has_ah) { const v4i Ah = taps_s[(2 * s) * 64 + l]; acc_hh[s] += Ah.x ^ Ah.y ^ Ah.z ^ Ah.w; }
        continue;
#endif
        // (the unconditional low-plane products come first: step 0 writes acc_mid with C = 0, so that only acc_hh
        // needs an explicit zero — a conditional first write makes the compiler materialise zeros on the other path)
        acc_mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al, uh, acc_mid, 0, 0, 0);
        if (CU8) {
          if (has_ah) acc_hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(taps_s[(2 * s) * 64 + l], uh, acc_hh, 0, 0, 0);
        } else {
          const v4i ul = *reinterpret_cast<const v4i *>(pl + 16 * s);
          acc_ll = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al, ul, acc_ll, 0, 0, 0);
          if (has_ah) {
            const v4i Ah = taps_s[(2 * s) * 64 + l];
            acc_hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah, uh, acc_hh, 0, 0, 0);
            acc_mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah, ul, acc_mid, 0, 0, 0);
          }
        }
      }
#endif
      // ---- epilogue, all in this lane: the tap fragments' rows are permuted at create time so that lane (n, h)
      // holds the WHOLE decimation group glw = 2n + h of the wave — sample j = 0..7 of it in accumulator registers
      // 2j (re) and 2j+1 (im): recombine -> >>14 -> rotate -> the box sum takes the products' high halves ----
      const int rel0 = tb + 8 * gw + MF_BLK * n + 8 * h;   // call-relative index of the lane's first sample
      const bool edge = (tb < 0) || (tb + groups_here * 8 > a.N);   // tile touches the call's borders (scalar)
      int2 sum;
      // (two copies of the sample loop behind a scalar branch: written as one loop with `if (edge)` inside, the
      // compiler if-converts the border test into 8 compares + selects per sample on every tile)
#ifdef K1_ABL_NOEPI   // tuning ablation (results wrong): no recombination / rotation / window sum
      sum = make_int2(0, 0);
#pragma unroll
      for (int r = 0; r < 16; r += 2) { sum.x += acc_hh[r] ^ acc_mid[r] ^ acc_ll[r]; sum.y += acc_hh[r + 1] ^ acc_mid[r + 1] ^ acc_ll[r + 1]; }
#else
      if (edge) sum = group_sum<ROT, CU8, true>(a, acc_hh, acc_mid, acc_ll, rel0);
      else sum = group_sum<ROT, CU8, false>(a, acc_hh, acc_mid, acc_ll, rel0);
#endif
      // the accumulators are dead now: the loads ride through the rest of the epilogue (the touch landed long ago)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(touch) : : "memory");
      if (tile + 1 < tile_end) fetch(tile + 1);

      group_finish(a, lut_s, c, n, h, gw, q0, groups_here, sum);
    }
    // the channel's last tile rolls the FIR history forward
    if (tile == a.tiles - 1) {
      for (int k = tid; k < a.HH; k += TPB) {
        const long qq = (long)a.N + k;   // index into concat(hist_old, in)
        a.hist_new[(long)c * a.HH + k] =
            qq < a.HH ? a.hist_old[(long)c * a.HH + qq] : raw_x(a, c, qq - a.HH);
      }
    }
  }
}


// =================================================================================================
// Path 1, complex<int16> input: the same matrix part and epilogue as iqbb_i16_mfma_kernel, fed by LDS-DMA.
//
// Measured on the register-staged kernel above (ablation builds, one box): without any arithmetic it still takes
// 104 us of its 160 us, without any global load 105 us — a workgroup has ONE tile of loads in flight, issued after
// its K loop and waited for before the next staging pass, so HBM latency sits on the critical path of every tile
// and the 12 prefetch registers cannot be doubled at 127 VGPRs. Here the next tile's raw samples travel global ->
// LDS by `global_load_lds_dwordx4` (no VGPR destination, nothing to wait for until the tile is needed), issued
// BEFORE the K loop of the current tile, so a tile has a whole K loop + epilogue to land:
//   loop:  barrier B1 (raw[i] landed: every wave's vmcnt(0); planes free: every wave is past K loop i-1)
//          raw[i] -> byte planes (ds_read_b128, 4 v_perm, 2 ds_write_b64 per 4 samples)
//          barrier B2 (planes[i] complete; raw free)
//          DMA raw[i+1]  (asynchronous)         K loop i            epilogue i
// LDS: table | one plane pair | raw tile | tap fragments = 37.6 KB -> 4 workgroups per CU as before.
// Tiles that touch the call's borders (history, zeros, end of input) fill the raw buffer by ordinary loads.
// =================================================================================================
template <int S, bool ROT>
__device__ __forceinline__ void iqbb_i16_mfma_dma_body(const IqbbArgs &a, const int bx, const int c) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  constexpr int OPc = 16 * (S - 1) + 1;
  constexpr int PLW = (2 * (TI + OPc) + 64 + 31) / 32 * 8;    // dwords per byte plane
  constexpr int QUADS = (TI + OPc + 4) / 4;                   // 16-byte pieces (4 samples) of a tile's window
  constexpr int RAWQ = (QUADS + 63) / 64 * 64;                // raw buffer in pieces: whole wave-instructions
  constexpr int NQ = (QUADS + TPB - 1) / TPB;
  int2 *lut_s = reinterpret_cast<int2 *>(smem);
  uint32_t *lo = smem + 256, *hi = lo + PLW;
  uint4 *raw = reinterpret_cast<uint4 *>(smem + 256 + 2 * PLW);
  v4i *taps_s = reinterpret_cast<v4i *>(smem + 256 + 2 * PLW + 4 * RAWQ);

  const int tid = threadIdx.x;
  const int w = tid >> 6, l = tid & 63, n = l & 31, h = l >> 5;
  for (int i = tid; i < S * 2 * 64; i += TPB) taps_s[i] = a.tapfrag[i];
  if (tid < 128) lut_s[tid] = a.lut[tid];

  const int tile_end = min((bx + 1) * a.tpw, a.tiles);
  int tile = bx * a.tpw;
  auto next_tile = [&](int t) { return t + 1; };
  const uint32_t *row = a.in + (long)c * a.in_stride;

  // raw[p] = samples first + 4p .. first + 4p + 3 of the tile's window (first = tile start - (OP - 1))
  auto stage_raw = [&](int tile_) {
    const int q0_ = tile_ * a.OG - a.ovl;
    const int first = a.base0_rel + q0_ * 8 - (a.OP - 1);
    const bool interior = first >= 0 && first + 4 * QUADS <= a.N;   // no history, no end of call (scalar)
    if (interior) {
#ifndef K1_ABL_NOFETCH
#pragma unroll
      for (int k = 0; k < NQ; k++) {
        const int p = tid + k * TPB;
        if (p < QUADS)   // destination: wave-uniform base + 16 * lane
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(row + first + 4 * p),
                                           (__attribute__((address_space(3))) void *)(raw + (p - l)), 16, 0, 0);
      }
#endif
    } else {
      // border tile: history / input / zeros per sample, branch-free — every load is issued unconditionally from a
      // clamped address and masked afterwards, so that all of a lane's 12 loads are in flight together (per-sample
      // branches made each load its own round trip: 28 us for the 2 border tiles of 1024 channels)
      const uint32_t *hrow = a.hist_old + (long)c * a.HH;
      uint32_t v[NQ][4];
#pragma unroll
      for (int k = 0; k < NQ; k++) {
        const int p = min(tid + k * TPB, QUADS - 1);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int rel = first + 4 * p + j, hh = a.HH + rel;
          const uint32_t *src = rel >= 0 ? row + min(rel, a.N - 1) : hrow + max(hh, 0);
          v[k][j] = *src;
        }
      }
#pragma unroll
      for (int k = 0; k < NQ; k++) {
        const int p = tid + k * TPB;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int rel = first + 4 * min(p, QUADS - 1) + j;
          if (rel >= a.N || a.HH + rel < 0) v[k][j] = 0u;
        }
        if (p < QUADS) raw[p] = make_uint4(v[k][0], v[k][1], v[k][2], v[k][3]);
      }
    }
  };
  if (tile < tile_end) stage_raw(tile);
  const int OGw = 64 - a.ovl;   // groups a wave emits; with FM its first group only supplies the previous angle
  for (; tile < tile_end; tile = next_tile(tile)) {
    const int q0 = tile * a.OG - a.ovl;
    const int tb = a.base0_rel + q0 * 8;
    const int groups_here = min(a.CG, a.n_groups - q0);

    __syncthreads();   // B1: the tile's raw samples have landed (vmcnt(0) of every wave) and the planes are free
    // ---- four samples -> 8 bytes of the low plane (offset to signed) and 8 of the high plane ----
#pragma unroll
    for (int k = 0; k < NQ; k++) {
      const int p = tid + k * TPB;
      if (p < QUADS) {
        const uint4 x = raw[p];
        // 16-byte chunks (8 samples) are de-interleaved by parity (even chunks in the first half of the plane, odd in
        // the second): a lane's K steps then walk consecutive chunks and the 16 lanes a ds_read_b128 services together
        // cover one contiguous 256-byte bank row instead of every other slot
        const int d = (((p >> 1) & 1) * (PLW >> 1)) + ((p >> 2) << 2) + ((p & 1) << 1);
        uint2 l2, h2;
        l2.x = __builtin_amdgcn_perm(x.y, x.x, 0x06040200u) ^ 0x80808080u;
        l2.y = __builtin_amdgcn_perm(x.w, x.z, 0x06040200u) ^ 0x80808080u;
        h2.x = __builtin_amdgcn_perm(x.y, x.x, 0x07050301u);
        h2.y = __builtin_amdgcn_perm(x.w, x.z, 0x07050301u);
        *reinterpret_cast<uint2 *>(lo + d) = l2;
        *reinterpret_cast<uint2 *>(hi + d) = h2;
      }
    }
    __syncthreads();   // B2: planes complete, raw buffer free
    if (next_tile(tile) < tile_end) stage_raw(next_tile(tile));   // in flight during this tile's K loop and epilogue

    const int gw = w * OGw;   // this wave's first group within the tile
    if (gw + a.ovl < groups_here) {   // wave-uniform: the wave has at least one group of its own
      v16i acc_hh = {0}, acc_mid = {0}, acc_ll = {0};
#pragma unroll
      for (int r = 0; r < 16; r++) acc_ll[r] = (r & 1) ? a.cim : a.cre;   // + 128*sum(a) rides in as C
      // chunk (16 B = 8 samples of one plane) gw + 2(n+s) + h of the tile, in the parity-split layout
      const int coff = ((gw + h) & 1) * (2 * PLW) + 16 * (((gw + h) >> 1) + n);
      const char *pl = reinterpret_cast<const char *>(lo) + coff;
      const char *ph = reinterpret_cast<const char *>(hi) + coff;
#ifdef K1_ABL_NOKLOOP   // tuning ablation (results wrong): no LDS operand reads, no MFMAs
      acc_mid[0] = *reinterpret_cast<const int *>(ph); acc_ll[1] = *reinterpret_cast<const int *>(pl);
#else
#pragma unroll
      for (int s = 0; s < S; s++) {
        // the outer taps of a windowed sinc are small: where every tap a K step touches has a zero high byte, its
        // two high-plane products are skipped (scalar branch on a mask made at create time; 4 of 9 steps for the
        // 127-tap north-star filter)
        const bool has_ah = (a.ah_mask >> s) & 1;
        const v4i uh = *reinterpret_cast<const v4i *>(ph + 16 * s);
        const v4i Al = taps_s[(2 * s + 1) * 64 + l];
        const v4i ul = *reinterpret_cast<const v4i *>(pl + 16 * s);
        // (the unconditional low-plane products come first: step 0 writes acc_mid with C = 0, so that only acc_hh
        // needs an explicit zero)
        acc_mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al, uh, acc_mid, 0, 0, 0);
        acc_ll = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al, ul, acc_ll, 0, 0, 0);
        if (has_ah) {
          const v4i Ah = taps_s[(2 * s) * 64 + l];
          acc_hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah, uh, acc_hh, 0, 0, 0);
          acc_mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah, ul, acc_mid, 0, 0, 0);
        }
      }
#endif
      // ---- epilogue, all in this lane (see iqbb_i16_mfma_kernel): lane (n, h) holds the whole group 2n + h ----
      const int rel0 = tb + 8 * gw + MF_BLK * n + 8 * h;   // call-relative index of the lane's first sample
      const bool edge = (tb < 0) || (tb + groups_here * 8 > a.N);   // tile touches the call's borders (scalar)
      int2 sum;
#ifdef K1_ABL_NOEPI   // tuning ablation (results wrong): no recombination / rotation / window sum
      sum = make_int2(0, 0);
#pragma unroll
      for (int r = 0; r < 16; r += 2) { sum.x += acc_hh[r] ^ acc_mid[r] ^ acc_ll[r]; sum.y += acc_hh[r + 1] ^ acc_mid[r + 1] ^ acc_ll[r + 1]; }
      (void)rel0; (void)edge;
#else
      if (edge) sum = group_sum<ROT, false, true>(a, acc_hh, acc_mid, acc_ll, rel0);
      else sum = group_sum<ROT, false, false>(a, acc_hh, acc_mid, acc_ll, rel0);
#endif
      group_finish(a, lut_s, c, n, h, gw, q0, groups_here, sum);
    }
    // the channel's last tile rolls the FIR history forward
    if (tile == a.tiles - 1) {
      for (int k = tid; k < a.HH; k += TPB) {
        const long qq = (long)a.N + k;   // index into concat(hist_old, in)
        a.hist_new[(long)c * a.HH + k] =
            qq < a.HH ? a.hist_old[(long)c * a.HH + qq] : raw_x(a, c, qq - a.HH);
      }
    }
  }
}
template <int S, bool ROT>
__global__ __launch_bounds__(TPB, 4) void iqbb_i16_mfma_dma_kernel(const IqbbArgs a) {
  iqbb_i16_mfma_dma_body<S, ROT>(a, (int)blockIdx.x, (int)blockIdx.y);
}

// =================================================================================================
// Path 4: the real-input BaseBand<int16_t> (src/baseband.hh:425-460), D = 8, on the matrix cores.
//
// The element stream is the real sample stream itself (one int16 per sample), a tap is a complex Q16 value, so
//   Dmat[m = (t, comp)][n = block] = sum_k TapT[m][k] * U[k][n],   TapT[m][k] = K_comp[k - t]
// with the same 32 x 32 tiles (16 samples x 2 components per block, 32 blocks per wave), the same byte-plane
// products and the same lane-owned decimation groups as path 1 — only the window of a block advances by 16 elements
// instead of 32 (a K step's operand is the 16-byte chunk n + 2s + h of a plane: consecutive lanes, consecutive
// chunks, conflict-free without the parity split), a 127-tap filter needs 5 K steps instead of 9, the FIR's shift
// is Traits<int16_t>::shift = 16, and the taps must fit two byte planes (|K| < 2^15: a Q16 tap reaches that only for
// filters wider than half the band; those stay on the VALU kernel).
// Wave-autonomous like the hot kernel: a wave stages its own 512 + 32S - 16 sample window (its start is 8-sample
// aligned only: a private copy keeps every operand read 16-byte aligned) into private byte planes — no workgroup
// barrier in the loop; the next tile's window is loaded into registers before the K loop and written after the
// epilogue. LDS: table 1 KB | tap fragments [S][2][64] x 16 B | per wave 2 planes of 512 + 32S bytes.
// =================================================================================================
template <int S, bool ROT>
__global__ __launch_bounds__(TPB, 4) void bb_real_mfma_kernel(const IqbbArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  constexpr int WINB = 512 + 32 * S;   // bytes per plane = samples in a wave's window (OP - 1 = 32S - 16 of them the halo, 16 spare)
  constexpr int NPC = WINB / 8;        // 8-sample pieces (one 16-byte global load each)
  int2 *lut_s = reinterpret_cast<int2 *>(smem);
  v4i *taps_s = reinterpret_cast<v4i *>(smem + 256);
  const int c = blockIdx.y, tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, n = l & 31, h = l >> 5;
  char *lo = reinterpret_cast<char *>(smem + 256 + S * 2 * 64 * 4) + w * 2 * WINB, *hi = lo + WINB;
  for (int i = tid; i < S * 2 * 64; i += TPB) taps_s[i] = a.tapfrag[i];
  if (tid < 128) lut_s[tid] = a.lut[tid];

  const int OGw = 64 - a.ovl, gw = w * OGw;
  const short *row = reinterpret_cast<const short *>(a.in) + (long)c * a.in_stride;
  struct __attribute__((packed, aligned(2))) Oct { uint32_t v[4]; };   // 16-byte load from a 2-byte aligned address
  uint32_t px[2][4];
  auto fetch = [&](int tile_) {
    const int ws = a.base0_rel + (tile_ * a.OG - a.ovl + gw) * 8 - (a.OP - 1);   // the window's first sample
    const bool interior = ws >= 0 && ws + WINB <= a.N;                            // (scalar)
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int p = l + 64 * k;
      if (p < NPC) {
        if (interior) {
          const Oct o = *reinterpret_cast<const Oct *>(row + ws + 8 * p);
#pragma unroll
          for (int j = 0; j < 4; j++) px[k][j] = o.v[j];
        } else {   // history / zeros beyond the call: per sample (load_x sign-extends a real sample into a dword)
#pragma unroll
          for (int j = 0; j < 4; j++)
            px[k][j] = (load_x(a, c, ws + 8 * p + 2 * j) & 0xffffu) | (load_x(a, c, ws + 8 * p + 2 * j + 1) << 16);
        }
      }
    }
  };
  auto split = [&]() {   // 8 samples -> 8 bytes of the low plane (offset to signed) and 8 of the high plane
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int p = l + 64 * k;
      if (p < NPC) {
        uint2 l2, h2;
        l2.x = __builtin_amdgcn_perm(px[k][1], px[k][0], 0x06040200u) ^ 0x80808080u;
        l2.y = __builtin_amdgcn_perm(px[k][3], px[k][2], 0x06040200u) ^ 0x80808080u;
        h2.x = __builtin_amdgcn_perm(px[k][1], px[k][0], 0x07050301u);
        h2.y = __builtin_amdgcn_perm(px[k][3], px[k][2], 0x07050301u);
        *reinterpret_cast<uint2 *>(lo + 8 * p) = l2;
        *reinterpret_cast<uint2 *>(hi + 8 * p) = h2;
      }
    }
  };
  const int tile_end = min((int)(blockIdx.x + 1) * a.tpw, a.tiles);
  int tile = blockIdx.x * a.tpw;
  auto has_work = [&](int tile_) { return tile_ < tile_end && gw + a.ovl < min(a.CG, a.n_groups - (tile_ * a.OG - a.ovl)); };
  if (has_work(tile)) fetch(tile);
  __syncthreads();   // tap fragments and table in place (the only workgroup barrier)
  for (; tile < tile_end; tile++) {
    const int q0 = tile * a.OG - a.ovl, tb = a.base0_rel + q0 * 8;
    const int groups_here = min(a.CG, a.n_groups - q0);
    if (gw + a.ovl < groups_here) {   // wave-uniform: the wave has at least one group of its own
      split();
      asm volatile("" ::: "memory");   // (one wave's LDS operations execute in order: the reads below see these writes)
      if (has_work(tile + 1)) fetch(tile + 1);   // in flight during the K loop and the epilogue
      v16i acc_hh = {0}, acc_mid = {0}, acc_ll = {0};
#pragma unroll
      for (int r = 0; r < 16; r++) acc_ll[r] = (r & 1) ? a.cim : a.cre;   // + 128*sum(K) rides in as C
      const char *pl = lo + 16 * (n + h), *ph = hi + 16 * (n + h);
#pragma unroll
      for (int s = 0; s < S; s++) {
        const bool has_ah = (a.ah_mask >> s) & 1;   // K steps whose taps all have a zero high byte skip two products
        const v4i uh = *reinterpret_cast<const v4i *>(ph + 32 * s), ul = *reinterpret_cast<const v4i *>(pl + 32 * s);
        const v4i Al = taps_s[(2 * s + 1) * 64 + l];
        acc_mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al, uh, acc_mid, 0, 0, 0);
        acc_ll = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al, ul, acc_ll, 0, 0, 0);
        if (has_ah) {
          const v4i Ah = taps_s[(2 * s) * 64 + l];
          acc_hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah, uh, acc_hh, 0, 0, 0);
          acc_mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah, ul, acc_mid, 0, 0, 0);
        }
      }
      const int rel0 = tb + 8 * gw + MF_BLK * n + 8 * h;   // call-relative index of the lane's first sample
      const bool edge = (tb < 0) || (tb + groups_here * 8 > a.N);
      int2 sum;
      if (edge) sum = group_sum<ROT, false, true, 0, 16>(a, acc_hh, acc_mid, acc_ll, rel0);
      else sum = group_sum<ROT, false, false, 0, 16>(a, acc_hh, acc_mid, acc_ll, rel0);
      group_finish(a, lut_s, c, n, h, gw, q0, groups_here, sum);
      asm volatile("" ::: "memory");
    }
    if (tile == a.tiles - 1) {   // the channel's last tile rolls the FIR history forward
      for (int k = tid; k < a.HH; k += TPB) {
        const long qq = (long)a.N + k;   // index into concat(hist_old, in)
        a.hist_new[(long)c * a.HH + k] = qq < a.HH ? a.hist_old[(long)c * a.HH + qq] : raw_x(a, c, qq - a.HH);
      }
    }
  }
}


// =================================================================================================
// Path 3: the 32x32x32 formulation for ANY decimation D. The matrix part is path 1's (every input sample's FIR
// value is needed whatever D is), but the box windows no longer line up with lanes: the rotated samples go to LDS
// (vbuf, one pad entry per 16 so that the lanes' 128-byte stride spreads over the banks) and the windows are summed
// from there by the grouping code of the VALU kernel (finalize_group / epilogue_and_roll), one tile per workgroup.
// =================================================================================================
__device__ __forceinline__ int PADV(int p) { return p + (p >> 4); }

template <int S, bool ROT, bool CU8>   // CU8: complex<uint8> input, one byte plane (see path 1)
__global__ __launch_bounds__(TPB, 3) void iqbb_i16_mfmag_kernel(const IqbbArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int PLW = (2 * (TI + a.OP) + 64 + 31) / 32 * 8;    // dwords per byte plane
  // LDS: rotation table | one plane pair | tap fragments [S][2][64] | vbuf (TI + TI/16 int2) | ybuf (+ FM angle cache)
  int2 *lut_s = reinterpret_cast<int2 *>(smem);
  uint32_t *lo = smem + 256, *hi = lo + PLW;
  v4i *taps_s = reinterpret_cast<v4i *>(smem + 256 + 2 * PLW);
  int2 *vbuf = reinterpret_cast<int2 *>(smem + 256 + 2 * PLW + S * 2 * 64 * 4);
  uint32_t *ybuf = reinterpret_cast<uint32_t *>(vbuf + TI + TI / 16 + 2);

  const int c = blockIdx.y, tid = threadIdx.x;
  const int w = tid >> 6, l = tid & 63, n = l & 31, h = l >> 5;
  for (int i = tid; i < S * 2 * 64; i += TPB) taps_s[i] = a.tapfrag[i];
  if (tid < 128) lut_s[tid] = a.lut[tid];

  // a workgroup walks `tpw` consecutive tiles; the next tile's samples are loaded into registers while this one is
  // computed (16-byte loads, all in flight), so that HBM latency is not paid once per tile
  constexpr int NQ = (TI + 16 * (S - 1) + 1 + 2 + 4 * TPB - 1) / (4 * TPB);
  struct __attribute__((packed, aligned(4))) Quad { uint32_t v[4]; };
  const int quads = (TI + a.OP + 4) / 4;
  Quad px[NQ];
  auto fetch = [&](int tile_) {
    const int first = a.base0_rel + (tile_ * a.OG - a.ovl) * a.D - (a.OP - 1);
    const bool interior = first >= 0 && first + 4 * quads <= a.N;
    if (CU8) {   // px[k].v[0..1] = the 8 high-plane bytes of 4 samples
      struct __attribute__((packed, aligned(2))) Oct { uint32_t v[2]; };
      const uint16_t *src8 = reinterpret_cast<const uint16_t *>(a.in) + (long)c * a.in_stride + first;
#pragma unroll
      for (int k = 0; k < NQ; k++) {
        const int p = tid + k * TPB;
        if (p < quads) {
          if (interior) {
            const Oct o = *reinterpret_cast<const Oct *>(src8 + 4 * p);
            px[k].v[0] = add129_bytes(o.v[0]); px[k].v[1] = add129_bytes(o.v[1]);
          } else {
            const uint32_t x0 = load_x(a, c, first + 4 * p), x1 = load_x(a, c, first + 4 * p + 1);
            const uint32_t x2 = load_x(a, c, first + 4 * p + 2), x3 = load_x(a, c, first + 4 * p + 3);
            px[k].v[0] = __builtin_amdgcn_perm(x1, x0, 0x07050301u); px[k].v[1] = __builtin_amdgcn_perm(x3, x2, 0x07050301u);
          }
        }
      }
      return;
    }
    const uint32_t *src = a.in + (long)c * a.in_stride + first;
#pragma unroll
    for (int k = 0; k < NQ; k++) {
      const int p = tid + k * TPB;
      if (p < quads) {
        if (interior && !a.in_cu8) px[k] = *reinterpret_cast<const Quad *>(src + 4 * p);
        else {
#pragma unroll
          for (int j = 0; j < 4; j++) px[k].v[j] = load_x(a, c, first + 4 * p + j);
        }
      }
    }
  };
  const int tile_end = min((int)(blockIdx.x + 1) * a.tpw, a.tiles);
  int tile = blockIdx.x * a.tpw;
  if (tile < tile_end) fetch(tile);
  for (; tile < tile_end; tile++) {
  const int q0 = tile * a.OG - a.ovl;    // first group (relative to the call's first group) of this tile
  const int tb = a.base0_rel + q0 * a.D; // call-relative index of the tile's first sample
  const int groups_here = min(a.CG, a.n_groups - q0);

  // ---- stage the tile's TI + OP samples as byte planes (parity-split 16-byte chunks, as path 1) ----
  {
#pragma unroll
    for (int k = 0; k < NQ; k++) {
      const int p = tid + k * TPB;
      if (p < quads) {
        const int d = (((p >> 1) & 1) * (PLW >> 1)) + ((p >> 2) << 2) + ((p & 1) << 1);
        if (CU8) {
          *reinterpret_cast<uint2 *>(hi + d) = make_uint2(px[k].v[0], px[k].v[1]);
        } else {
          uint2 l2, h2;
          l2.x = __builtin_amdgcn_perm(px[k].v[1], px[k].v[0], 0x06040200u) ^ 0x80808080u;
          l2.y = __builtin_amdgcn_perm(px[k].v[3], px[k].v[2], 0x06040200u) ^ 0x80808080u;
          h2.x = __builtin_amdgcn_perm(px[k].v[1], px[k].v[0], 0x07050301u);
          h2.y = __builtin_amdgcn_perm(px[k].v[3], px[k].v[2], 0x07050301u);
          *reinterpret_cast<uint2 *>(lo + d) = l2;
          *reinterpret_cast<uint2 *>(hi + d) = h2;
        }
      }
    }
  }
  __syncthreads();
  if (tile + 1 < tile_end) fetch(tile + 1);   // in flight during the matrix work and the window sums below

  // ---- the FIR at this wave's 512 samples (blocks 32w .. 32w+31 of the tile) ----
  {
    v16i acc_hh = {0}, acc_mid = {0}, acc_ll = {0};
    if (!CU8) {
#pragma unroll
      for (int r = 0; r < 16; r++) acc_ll[r] = (r & 1) ? a.cim : a.cre;   // + 128*sum(a) rides in as C
    }
    const int cw = 64 * w;   // the wave's first chunk (8 samples) within the tile
    const int coff = ((cw + h) & 1) * (2 * PLW) + 16 * (((cw + h) >> 1) + n);
    const char *pl = reinterpret_cast<const char *>(lo) + coff;
    const char *ph = reinterpret_cast<const char *>(hi) + coff;
#pragma unroll
    for (int s = 0; s < S; s++) {
      const bool has_ah = (a.ah_mask >> s) & 1;
      const v4i uh = *reinterpret_cast<const v4i *>(ph + 16 * s);
      const v4i Al = taps_s[(2 * s + 1) * 64 + l];
      acc_mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al, uh, acc_mid, 0, 0, 0);
      if (CU8) {
        if (has_ah) acc_hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(taps_s[(2 * s) * 64 + l], uh, acc_hh, 0, 0, 0);
      } else {
        const v4i ul = *reinterpret_cast<const v4i *>(pl + 16 * s);
        acc_ll = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al, ul, acc_ll, 0, 0, 0);
        if (has_ah) {
          const v4i Ah = taps_s[(2 * s) * 64 + l];
          acc_hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah, uh, acc_hh, 0, 0, 0);
          acc_mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah, ul, acc_mid, 0, 0, 0);
        }
      }
    }
    // recombine, >>14, rotate, mask samples outside the call, park in vbuf
    const int pos0 = 512 * w + MF_BLK * n + 2 * h;   // sample k = 2q+tt of this lane sits at pos0 + 4q + tt of the tile
    uint32_t cnt0 = 0;
    if (ROT) cnt0 = (a.n0_lo + (uint32_t)(tb + pos0)) * a.inc;
    const uint32_t negx = a.negative ? (127u << 3) : 0u;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int rr = 4 * (k >> 1) + 2 * (k & 1), dk = 4 * (k >> 1) + (k & 1);
      unsigned tre = ((unsigned)acc_hh[rr] << 8) + (unsigned)acc_mid[rr];
      unsigned tim = ((unsigned)acc_hh[rr + 1] << 8) + (unsigned)acc_mid[rr + 1];
      int2 v;
      if (CU8) {   // S = t << 8 exactly
        v = make_int2((int)(tre << 8) >> 14, (int)(tim << 8) >> 14);
      } else {
        asm("" : "+v"(tre)); asm("" : "+v"(tim));
        const unsigned sre = (tre << 8) + (unsigned)acc_ll[rr], sim = (tim << 8) + (unsigned)acc_ll[rr + 1];
        v = make_int2((int)sre >> 14, (int)sim >> 14);
      }
      if (ROT) {
        const uint32_t off = (((cnt0 + (uint32_t)dk * a.inc) >> 5) & (127u << 3)) ^ negx;
        const int2 L = *reinterpret_cast<const int2 *>(reinterpret_cast<const char *>(lut_s) + off);
        const int2 r = v;
        v.x = mad24a(L.x, r.x, -mul24a(L.y, r.y)) >> 16;
        v.y = mad24a(L.x, r.y, mul24a(L.y, r.x)) >> 16;
      }
      const int rel = tb + pos0 + dk;
      if (rel < 0 || rel >= a.N) v = make_int2(0, 0);
      vbuf[PADV(pos0 + dk)] = v;
    }
  }
  __syncthreads();

  // ---- box sums per group, truncating division, state (as the VALU kernel) ----
  // `lpg` lanes (a power of two <= 64, about D/8) share one window: strided partial sums, then a butterfly over them
  {
    const int lpg = a.lpg, sub = tid & (lpg - 1), gpp = TPB / lpg;
    for (int g0 = 0; g0 < groups_here; g0 += gpp) {
      const int ql = g0 + tid / lpg, q = q0 + ql;
      int2 s = make_int2(0, 0);
      if (ql < groups_here) {
        for (int k = sub; k < a.D; k += lpg) {
          const int2 v = vbuf[PADV(ql * a.D + k)];
          s.x = (int)((unsigned)s.x + (unsigned)v.x);
          s.y = (int)((unsigned)s.y + (unsigned)v.y);
        }
      }
      for (int m = lpg >> 1; m >= 1; m >>= 1) {
        s.x = (int)((unsigned)s.x + (unsigned)__shfl_xor(s.x, m));
        s.y = (int)((unsigned)s.y + (unsigned)__shfl_xor(s.y, m));
      }
      if (sub == 0 && ql < groups_here && q >= 0)   // (tile 0's overlap slot precedes the call)
        finalize_group(a, c, lut_s, ybuf, ql, q, s, a.D);
    }
  }
  __syncthreads();
  epilogue_and_roll(a, c, tile, tid, q0, groups_here, ybuf);
  }   // (the next tile's planes / vbuf / ybuf writes are each separated from this tile's reads by one of its barriers)
}

// The any-D hot form with FM stores the first output of slice s as -phi and leaves the last angle of every slice in
// philast; this adds philast[s - 1] for the slices fix_lo <= s < fix_hi (first output fix_gs * s). After the hot kernel.
// (one slice per lane, blockIdx.y = channel)
__global__ void iqbb_fm_fixup_kernel(short *__restrict__ out, long out_stride, const short *__restrict__ philast, int philast_stride, int fix_lo, int fix_hi, int fix_gs, int C) {
  // one slice per lane (a wave per channel walking its 512 slices in 8 dependent trips took 5 us for half a megabyte)
  const int c = blockIdx.y, sl = fix_lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (sl >= fix_hi) return;
  short *row = out + (long)c * out_stride;
  row[sl * fix_gs] = (short)(row[sl * fix_gs] + philast[(long)c * philast_stride + sl - 1]);
}

// sdrhip_iqbb_i16_process_dev_multi with the FM epilogue: B reference-sized buffers went through ONE launch as one long call,
// whose FM outputs are the differences of consecutive angles throughout. FMDemod (src/demod.hh:242-254) starts every buffer
// anew: index 0 of a buffer's output is never written (in place it keeps the real part of the baseband's first value) and
// index 1 is the PREVIOUS BUFFER's last angle minus the angle of element 1 — element 0's angle is never looked at. With
// q = the long call's output index of a buffer's first element and L the long call's outputs (mod 2^16):
//   out[q + 1] = phi(q - 1) - phi(q + 1) = L[q] + L[q + 1],     out[q] = Re(baseband output q),
// the latter recomputed from the input the way the reference does: FIR at the group's D samples, >>14 (real input: >>16),
// LUT rotation, box sum, truncating division. One workgroup per (boundary, channel): TL lanes share a sample's taps.
struct MultiFix { int nb; int q[63]; };
__global__ __launch_bounds__(256) void iqbb_fm_multi_fixup_kernel(const IqbbArgs a, const MultiFix f) {
  __shared__ int2 part[256];
  const int c = blockIdx.y, q = f.q[blockIdx.x], tid = threadIdx.x, D = a.D;
  int TL = 1;   // (bounded: left open-ended the compiler derived a zero stride for the final sum and dropped the block behind it)
  while (TL < 256 && 2 * TL * min(D, 256) <= 256) TL *= 2;
  const int SB = 256 / TL, il = tid / TL, ts = tid - il * TL;   // samples per pass, this thread's sample and tap slice
  const int first = a.base0_rel + q * D;                       // the group's first sample, call-relative (q >= 2: no carry, not the stream's first window)
  int2 vsum = make_int2(0, 0);
  for (int p0 = 0; p0 < D; p0 += SB) {
    const int i = p0 + il, rel = first + i;
    int er = 0, ei = 0;
    if (i < D)
      for (int k = ts; k < a.OP; k += TL) {
        const uint32_t x = load_x(a, c, rel - (a.OP - 1) + k);
        const uint2 kk = a.taps[k];
        if (a.in_real) { er = (int)((unsigned)er + (unsigned)mulw((int)kk.x, (int)x)); ei = (int)((unsigned)ei + (unsigned)mulw((int)kk.y, (int)x)); }
        else { er = dot2(x, kk.x, er); ei = dot2(x, kk.y, ei); }
      }
    part[tid] = make_int2(er, ei);
    __syncthreads();
    if (ts == 0 && i < D) {
      for (int t = 1; t < TL; t++) { er = (int)((unsigned)er + (unsigned)part[tid + t].x); ei = (int)((unsigned)ei + (unsigned)part[tid + t].y); }
      const int sh = a.in_real ? 16 : 14;
      const int2 v = rotate(a, a.lut, make_int2(er >> sh, ei >> sh), a.n0_lo + (uint32_t)rel);
      vsum.x = (int)((unsigned)vsum.x + (unsigned)v.x); vsum.y = (int)((unsigned)vsum.y + (unsigned)v.y);
    }
    __syncthreads();
  }
  part[tid] = vsum;
  __syncthreads();
  if (tid == 0) {
    int sx = 0;
    for (int t = 0; t < 256; t += TL) sx = (int)((unsigned)sx + (unsigned)part[t].x);
    short *row = reinterpret_cast<short *>(a.out) + (long)c * a.out_stride;
    const short l0 = row[q], l1 = row[q + 1];
    row[q + 1] = (short)(l0 + l1);
    row[q] = (short)box_div(sx, D);
  }
}

// Decimations above 256 on the hot structure (iqbb_hot.hpp, PART), the stand-alone finishing launch: one lane per group
// (bigd_finish_group, iqbb_common.hpp). Where whole channels are the hot kernel's units it runs the same function as its
// workgroups' last step instead.
__global__ __launch_bounds__(256) void iqbb_bigd_finish_kernel(const BigdArgs a) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q < a.n_groups) bigd_finish_group(a, (int)blockIdx.y, q);
}

}  // namespace

namespace {
// the hot kernels live in one translation unit per filter-length class (iqbb_hot_s*.hip)
void launch_hot(int S, int in, int range, bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b) {
  if (in == HOT_REAL) { hot_launch_real(S, range, rot, epi, hl, ha, b); return; }
  if (in == HOT_CS8) { hot_launch_cs8(S, range, rot, epi, hl, ha, b); return; }
  const bool cu8 = in == HOT_CU8;
  switch (S) {
    case 2: hot_launch_s2(in, range, rot, epi, hl, ha, b); break;
    case 3: hot_launch_s3(in, range, rot, epi, hl, ha, b); break;
    case 5: hot_launch_s5(in, range, rot, epi, hl, ha, b); break;
    case 9: if (cu8) hot_launch_s9_cu8(range, rot, epi, hl, ha, b); else hot_launch_s9_cs16(range, rot, epi, hl, ha, b); break;
    case 17: if (cu8) hot_launch_s17_cu8(range, rot, epi, hl, ha, b); else hot_launch_s17_cs16(range, rot, epi, hl, ha, b); break;
    default: if (cu8) hot_launch_s33_cu8(range, rot, epi, hl, ha, b); else hot_launch_s33_cs16(range, rot, epi, hl, ha, b); break;
  }
}
}  // namespace

struct sdrhip_iqbb_i16 {
  sdrhip_ctx *ctx = nullptr;
  int order = 0, OP = 0, HH = 0, D = 1, C = 1, epi = 0, negative = 0;
  uint32_t inc = 0;
  size_t max_in = 0;
  uint64_t n0 = 0;
  uint64_t phase0 = 0;   // absolute sample index at which the LUT phase counter was last restarted (set_shift)
  // the reference's _ring_offset at sample n0 is (ring_off0 + n0 - ring_n0) mod order: n0 mod order unless setOrder changed
  // the ring's length mid-stream (adopt_state with SDRHIP_KEEP_COUNTERS); what a later _reconfigure rotates the ring by
  uint64_t ring_n0 = 0;
  int ring_off0 = 0;
  int ring_offset() const { return (int)(((uint64_t)ring_off0 + (n0 - ring_n0)) % (uint64_t)order); }
  int par = 0, par_fm = 0;
  int CG = 0, OG = 0, ovl = 0;
  bool fast8 = false;
  bool use_hot = true;   // path 1, calls of >= 3 tiles: the hot kernel (SDRHIP_IQBB_HOT=0: the general kernels only, tuning/tests)
  int env_tpw = 0, env_wgpcu = 0;   // tuning hooks SDRHIP_IQBB_TPW / SDRHIP_IQBB_WGPCU, read once at create (0: not set)
  int env_fm_resident = -1;         // SDRHIP_IQBB_FM_RESIDENT=0|1: never / always complete the any-D forms' FM outputs inside the hot kernel (-1: by the channel count)
  int hot_range = -1;    // which compile-time high-plane K-step range of the hot kernel covers ah_mask (-1: none)
  int in_cu8 = 0, real = 0, i8 = 0;   // input kinds: complex<uint8> with AutoCast, real int16 (BaseBand), complex<int8> (IQBaseBand<int8_t>)
  int path = 0, S = 0, cre = 0, cim = 0;   // path 1 = int8-MFMA formulation with S K-steps
  uint64_t ah_mask = 0;   // bit s: the high-byte tap fragments of K step s are not all zero (up to 33 steps)
  DevBuf<v4i> tapfrag;
  DevBuf<v4i> tapfrag_hot;   // path 3: the any-D hot forms' fragments (rows permuted as path 1's)
  DevBuf<v4i> tapfrag_rot;   // paths 1 and 3: the hot forms' fragments for complex<uint8> input (bytes rotated inside every dword)
  size_t lds_bytes = 0;
  DevBuf<uint2> taps;
  DevBuf<int2> lut;
  DevBuf<uint32_t> hist[2];
  DevBuf<int2> acc[2];
  DevBuf<short> fm[2];
  DevBuf<uint32_t> stage_in;
  DevBuf<uint32_t> stage_out;
  size_t max_out = 0;
#ifdef K1_STAMPS
  DevBuf<unsigned long long> k1_stamps;   // diagnostic builds: per-wave phase totals of the hot kernel
#endif
  DevBuf<short> philast;   // any-D hot form with FM: the last angle of every slice (HotArgs::philast)
  DevBuf<long long> hs;    // ... where the units are not whole channels: the neighbouring slices' handshake entries (HotArgs::hs)
  int hs_stride = 0, hs_seq = 0;
  int env_fm_handshake = -1;   // SDRHIP_IQBB_FM_HANDSHAKE=0|1: the fix-up launch / the in-kernel handshake for such calls (-1 = 0: the launch — measured
                               // equal at 1 channel and faster from 16 channels on, profiles/r17_ab_fm_handshake.txt, r17_fm_latency.txt)
  DevBuf<int2> part;       // decimations above 256: three partial box sums per slice of the longest call (HotArgs::part)
  // Decimations 257 ... 512 run either form: the any-D form's one group per slice uses D of a slice's 512 samples (÷257: half
  // of the matrix work is thrown away), the large-decimation form all of them plus a 5 us launch — measured crossover at
  // D = 470 (21 taps, complex<uint8>, FM: ÷257 0.120 -> 0.080 ms per step, ÷300 0.105 -> 0.078, ÷400 0.085 -> 0.077, ÷480 0.075 / 0.076,
  // ÷512 0.072 / 0.075). SDRHIP_IQBB_BIGD_MIN=n (tests, A/B): exactly the decimations >= n take the large-decimation form.
  int bigd_min = 257, bigd_skip_lo = 465;   // (default: 257 ... 464 and 513 ...)
  bool bigd_always = false;                 // every call, however short, through the large-decimation form (its cold path serves any slice)


  // (re)loads the tap-dependent device data: packed taps (VALU kernel, the slow first-sample evaluation), the
  // Toeplitz byte-plane fragments, their constant term and high-plane step mask (MFMA paths). create and retap.
  void load_taps(const int32_t *taps) {
    ah_mask = 0; hot_range = -1;
    // taps: zero-padded at the FRONT (older samples) so that the newest sample still meets K[order-1]
    std::vector<uint2> tp(OP, make_uint2(0, 0));
    const int pad = OP - order;
    for (int i = 0; i < order; i++) {
      const int kr = taps[2 * i], ki = taps[2 * i + 1];
      if (real) { tp[pad + i].x = (uint32_t)kr; tp[pad + i].y = (uint32_t)ki; continue; }
      tp[pad + i].x = ((uint32_t)(uint16_t)(int16_t)kr) | ((uint32_t)(uint16_t)(int16_t)(-ki) << 16);
      tp[pad + i].y = ((uint32_t)(uint16_t)(int16_t)ki) | ((uint32_t)(uint16_t)(int16_t)kr << 16);
    }
    if (path == 4) {
      // real input: TapT[m = (t, comp)][k] = K_comp[k - t] over the real sample stream (window of a block = OP - 1 + 16
      // elements, OP = 32S - 15), rows permuted as for path 1; lane (m = l&31, hh = l>>5), byte j of K step s <-> k = 32s+16hh+j
      // (OPm: the window the matrix part covers; the plan's OP is the same at decimation 8 and the VALU kernel's at the others)
      const int OPm = 32 * S - 15, padm = OPm - order;
      std::vector<int> kre(OPm, 0), kim(OPm, 0);
      unsigned sre = 0, sim = 0;
      for (int i = 0; i < order; i++) { kre[padm + i] = taps[2 * i]; kim[padm + i] = taps[2 * i + 1]; sre += (unsigned)taps[2 * i]; sim += (unsigned)taps[2 * i + 1]; }
      cre = (int)(128u * sre); cim = (int)(128u * sim);
      std::vector<int8_t> frag((size_t)S * 2 * 64 * 16, 0);
      for (int st = 0; st < S; st++)
        for (int l = 0; l < 64; l++)
          for (int j = 0; j < 16; j++) {
            const int m = l & 31, hh = l >> 5;
            const int hC = (m >> 2) & 1, r = (m & 3) + 4 * (m >> 3);
            const int t = 8 * hC + (r >> 1), comp = r & 1;
            const int idx = 32 * st + 16 * hh + j - t;
            const int v = (idx >= 0 && idx < OPm) ? (comp ? kim[idx] : kre[idx]) : 0;
            const int al = ((v + 128) & 255) - 128, ah = (v - al) >> 8;
            frag[(((size_t)(2 * st) * 64 + l) * 16) + j] = (int8_t)ah;
            if (ah != 0) ah_mask |= (uint64_t)1 << st;
            frag[(((size_t)(2 * st + 1) * 64 + l) * 16) + j] = (int8_t)al;
          }
      {   // the hot kernel's compile-time high-plane range, as for path 1
        int nr = 0;
        const HotRange *rg = hot_ranges(S, &nr);
        for (int r = 0; r < nr && hot_range < 0; r++)
          if ((ah_mask & ~((((uint64_t)1 << rg[r].NH) - 1u) << rg[r].S0)) == 0) hot_range = r;
      }
      if (!tapfrag.p) tapfrag.alloc((size_t)S * 2 * 64);
      tapfrag.upload(reinterpret_cast<const v4i *>(frag.data()), (size_t)S * 2 * 64, ctx->stream);
    } else if (path >= 1) {   // (a path 3 plan that fell back to the VALU kernel above has path 0 by now)
      // interleaved tap vectors a_comp[2i+c] and their Toeplitz fragments, TapT[m][k] = a_comp[k-2t], m = 2t+comp:
      // 32x32x32: lane (m = l&31, hh = l>>5), byte j of K-step s <-> k = 32s+16hh+j
      // (OPm: the window the matrix part covers; the plan's OP is the same up to 257 taps, the VALU kernel's chunked length beyond)
      const int OPm = 16 * (S - 1) + 1, padm = OPm - order;
      std::vector<int> are(2 * OPm, 0), aim(2 * OPm, 0);
      for (int i = 0; i < order; i++) {
        const int kr = taps[2 * i], ki = taps[2 * i + 1];
        are[2 * (padm + i)] = kr; are[2 * (padm + i) + 1] = -ki;
        aim[2 * (padm + i)] = ki; aim[2 * (padm + i) + 1] = kr;
      }
      unsigned sre = 0, sim = 0;
      for (int k = 0; k < 2 * OPm; k++) { sre += (unsigned)are[k]; sim += (unsigned)aim[k]; }
      cre = (int)(128u * sre); cim = (int)(128u * sim);
      std::vector<int8_t> frag((size_t)S * 2 * 64 * 16, 0);
      auto build = [&](bool permuted) {
        for (int st = 0; st < S; st++)
          for (int l = 0; l < 64; l++)
            for (int j = 0; j < 16; j++) {
              const int m = l & 31, hh = l >> 5;
              int t = m >> 1, comp = m & 1;
              if (permuted) {   // row permutation: the 32x32 C/D map gives lane half hC = (m>>2)&1 the rows m with register
                                // r = (m&3) + 4*(m>>3); row m carries sample t = 8*hC + (r>>1), component r&1, so that
                                // a lane ends up with 8 consecutive samples (decimation 8: one whole group)
                const int hC = (m >> 2) & 1, r = (m & 3) + 4 * (m >> 3);
                t = 8 * hC + (r >> 1); comp = r & 1;
              }
              const int idx = 32 * st + 16 * hh + j - 2 * t;
              const int v = (idx >= 0 && idx < 2 * OPm) ? (comp ? aim[idx] : are[idx]) : 0;
              const int al = ((v + 128) & 255) - 128, ah = (v - al) >> 8;
              frag[(((size_t)(2 * st) * 64 + l) * 16) + j] = (int8_t)ah;
              if (ah != 0) ah_mask |= (uint64_t)1 << st;
              frag[(((size_t)(2 * st + 1) * 64 + l) * 16) + j] = (int8_t)al;
            }
      };
      // complex<uint8> input: the hot kernels leave the sample plane's bytes rotated by one inside every dword (add129_rot,
      // iqbb_hot.hpp) — their tap fragments follow (byte p of a dword = element (p + 1) mod 4 of that dword's four K elements)
      auto upload_rot = [&]() {
        std::vector<int8_t> r(frag.size());
        for (size_t i = 0; i < frag.size(); i++) r[i] = frag[(i & ~(size_t)3) | ((i + 1) & 3)];
        if (!tapfrag_rot.p) tapfrag_rot.alloc((size_t)S * 2 * 64);
        tapfrag_rot.upload(reinterpret_cast<const v4i *>(r.data()), (size_t)S * 2 * 64, ctx->stream);
      };
      if (path == 3) {   // the hot kernel's any-D forms read a permuted set of their own (iqbb_hot.hpp); the general kernel of
                         // the plan (short calls) keeps the natural row order
        build(true);
        if (!tapfrag_hot.p) tapfrag_hot.alloc((size_t)S * 2 * 64);
        tapfrag_hot.upload(reinterpret_cast<const v4i *>(frag.data()), (size_t)S * 2 * 64, ctx->stream);
        upload_rot();
      }
      build(path == 1);
      if (path == 1) upload_rot();
      if (path == 1 || path == 3) {   // smallest centred range [S0, S0+NH) of the hot kernel that covers the mask (path 3: its any-D form)
        int nr = 0;
        const HotRange *rg = hot_ranges(S, &nr);
        for (int r = 0; r < nr && hot_range < 0; r++)
          if ((ah_mask & ~((((uint64_t)1 << rg[r].NH) - 1u) << rg[r].S0)) == 0) hot_range = r;
      }
      if (!tapfrag.p) tapfrag.alloc((size_t)S * 2 * 64);
      tapfrag.upload(reinterpret_cast<const v4i *>(frag.data()), (size_t)S * 2 * 64, ctx->stream);
    }
    if (!this->taps.p) this->taps.alloc(OP);
    this->taps.upload(tp.data(), OP, ctx->stream);
  }

  struct Geometry { uint64_t g_first; int n_groups, n_out, base0_rel, extra0; };
  Geometry geometry(size_t N) const {
    Geometry g{};
    // IQBaseBand closes its first window after D+1 samples (:200,:212); the real BaseBand after D (:431-438)
    const uint64_t D64 = (uint64_t)D, shift1 = (D > 1 && !real) ? 1 : 0;
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
  size_t out_elem_bytes() const { return epi == SDRHIP_EPI_NONE ? (i8 ? 2 : 4) : 2; }
  size_t in_elem_bytes() const { return (in_cu8 || real || i8) ? 2 : 4; }

  // Path 1's hot kernel (iqbb_hot.hpp): one launch for the whole call — a persistent grid over the wave slices that
  // touch no border of the call, then the same workgroups' share of the cold slices (tile 0 and the tiles from t_hi on).
  // false: the call is too short to have a tile of hot slices; the general kernel runs it.
  bool launch_hot_call(IqbbArgs &a, const Geometry &g, const uint32_t *in_dev, size_t N, size_t in_stride, void *out_dev,
                       size_t out_stride, int tiles) {
    const int kind = hot_kind(), halo = hot_halo(S, kind), win = hot_win(S, kind);
    auto host_hot = [&](int t, int w) { return slice_is_hot(halo, win, g.base0_rel, OG, ovl, (int)N, g.n_out, t, w); };
    long t = tiles - 1;   // the last tile always holds cold slices (history roll, state)
    while (t >= 2 && !(host_hot((int)t - 1, 0) && host_hot((int)t - 1, 1) && host_hot((int)t - 1, 2) && host_hot((int)t - 1, 3))) t--;
    if (t < 2) return false;
    int nr = 0;
    const HotRange *rg = hot_ranges(S, &nr);
    const int NW = rg[hot_range].NW;
    HotArgs ha{};
    ha.in = in_dev; ha.in_stride = (long)in_stride; ha.out = out_dev; ha.out_stride = (long)out_stride;
    ha.tapfrag = in_cu8 ? tapfrag_rot.p : tapfrag.p; ha.lut = lut.p; ha.inc = inc; ha.n0_lo = (uint32_t)(n0 - phase0); ha.negative = negative;
    ha.base0_rel = g.base0_rel; ha.OG = OG; ha.ovl = ovl; ha.t_lo = 0; ha.t_hi = tiles; ha.cre = cre; ha.cim = cim;
    ha.N = (int)N; ha.n_out = g.n_out; ha.C = C; ha.stamps = nullptr;
    ha.D = 8; ha.GS = 64; ha.lpg_sh = 0; ha.inv_d = 0.125f; ha.philast = nullptr; ha.philast_stride = 0; ha.part = nullptr; ha.fin_groups = 0;   // (the any-D form's fields)
    // multi-buffer call with FM (launch_multi): the hot slices write the buffers' first two outputs themselves where the boundary
    // group and the one behind it are stored lanes of ONE hot slice; the others stay on the list for the fix-up launch
    ha.mb_p = 0; ha.mb_q1 = 0; ha.mb_qlast = 0; ha.mb_magic = 0;
    if (!mb_q.empty() && epi == SDRHIP_EPI_FM && !i8 && !hot_pair(kind, rg[hot_range].NW, false)) {
      const int per = 64 - ovl;   // stored groups per slice
      long P = mb_q.size() >= 2 ? (long)mb_q[1] - mb_q[0] : (long)1 << 30;
      bool uniform = P >= 128;
      for (size_t j = 2; j < mb_q.size(); j++) uniform = uniform && (long)mb_q[j] - mb_q[j - 1] == P;
      if (uniform) {
        ha.mb_q1 = mb_q[0]; ha.mb_p = (int)P; ha.mb_qlast = mb_q.back(); ha.mb_magic = (unsigned)((((unsigned long long)1) << 32) / (unsigned long long)P);
        std::vector<int> left;
        for (int q : mb_q) {
          const int tl = q / OG, w = (q % OG) / per, lane = (q % OG) % per + ovl;   // the slice that STORES group q, and its lane there
          if (!(lane <= 62 && host_hot(tl, w))) left.push_back(q);
        }
        mb_q.swap(left);   // what the fix-up launch still has to do
      }
    }
#ifdef K1_STAMPS
    if (!k1_stamps.p) { k1_stamps.alloc(32768 * 16); k1_stamps.zero(ctx->stream); }
    ha.stamps = k1_stamps.p;
#endif
    // persistent grid of 4 virtual (4-wave) workgroups per CU = 4 waves per SIMD; a real workgroup is NW / 4 of them.
    // Units of at most 4 tiles so that the static split leaves a short tail.
    const int nvwg = wgpcu() * ctx->prop.multiProcessorCount;   // virtual (4-wave) workgroups = waves per SIMD (SDRHIP_IQBB_WGPCU: tuning hook, builds with -DK1_MINWAVES=5)
    int htpw = 4; while (htpw > 1 && (size_t)ceil_div((size_t)tiles, (size_t)htpw) * C < 4 * (size_t)nvwg) htpw >>= 1;
    if (env_tpw) htpw = env_tpw;   // tuning hook
    ha.tpw = htpw;
    ha.G = (int)ceil_div((size_t)tiles, (size_t)htpw); ha.U = ha.G * C;
    const int vper = NW / 4;
    const int grid = (int)ceil_div((size_t)std::min(nvwg, std::max(ha.U, C)), (size_t)vper);   // (every channel's cold slices need a taker too)
    const int gx = grid * vper;   // virtual workgroups
    ha.dq = gx / ha.G; ha.dr = gx % ha.G;
    a.bt_hi = (int)t; a.tpw = 1;
    HotLaunch hl{(unsigned)grid, ctx->stream};
    launch_hot(S, kind, hot_range, inc != 0, epi, hl, ha, a);
    return true;
  }

  // Path 3's long calls: the hot kernel's any-D form (iqbb_hot.hpp, DG) — hot slices and, at the end of each workgroup,
  // the call's cold slices (history, carries, the stream's first sample, state for the next call); with FM a second,
  // tiny launch adds the previous slice's last angle to every slice's first output.
  // false: not this plan / call (the general kernel runs it).
  // Decimations from 257 on (bigd_min, bigd_skip_lo): the hot kernel's large-decimation form + iqbb_bigd_finish_kernel (iqbb_hot.hpp, PART)
  bool bigd_plan() const {
    return path == 3 && use_hot && hot_range >= 0 && S <= 33 && !i8 && !real && D >= bigd_min && D >= 257 && !(D >= bigd_skip_lo && D <= 512) && part.p != nullptr;
  }
  bool launch_bigd_call(const IqbbArgs &a0, const Geometry &g, const uint32_t *in_dev, size_t N, size_t in_stride, void *out_dev, size_t out_stride) {
    const int kind = in_cu8 ? HOT_CU8 : HOT_CS16, halo = hot_halo(S, kind), win = hot_win(S, kind);
    // the kernel's geometry: slices of 512 samples from the call's first sample on = "decimation 512", one pseudo-group per slice
    const int nsl = (int)ceil_div(N, (size_t)512), tiles_h = (int)ceil_div((size_t)nsl, (size_t)4);
    auto slice_hot = [&](int sl) { return slice_is_hot(halo, win, 0, 4, 0, (int)N, nsl, sl >> 2, sl & 3, 512, 1); };
    int s_lo = 0, s_hi = 4 * tiles_h;
    while (s_lo < s_hi && !slice_hot(s_lo)) s_lo++;
    while (s_hi > s_lo && !slice_hot(s_hi - 1)) s_hi--;
    if (s_hi - s_lo < 16 && !bigd_always) return false;   // (short calls: the general kernel, where the plan has one)
    SDRHIP_REQUIRE(part.n >= (size_t)C * 12 * tiles_h, SDRHIP_E_SIZE, "part holds %zu entries, the call needs %zu", part.n, (size_t)C * 12 * tiles_h);
    IqbbArgs a = a0;   // (the cold phase walks the PSEUDO groups; the real geometry goes to the finishing kernel)
    a.base0_rel = 0; a.n_groups = nsl; a.n_out = nsl; a.extra0 = 0; a.D = 512; a.fix_lo = a.fix_hi = 0;
    HotArgs ha{};
    ha.in = in_dev; ha.in_stride = (long)in_stride; ha.out = out_dev; ha.out_stride = (long)out_stride;
    ha.tapfrag = in_cu8 ? tapfrag_rot.p : tapfrag_hot.p; ha.lut = lut.p; ha.inc = inc; ha.n0_lo = (uint32_t)(n0 - phase0); ha.negative = negative;
    ha.base0_rel = 0; ha.OG = 4; ha.ovl = 0; ha.t_lo = s_lo >> 2; ha.t_hi = (s_hi + 3) >> 2; ha.cre = cre; ha.cim = cim;
    ha.N = (int)N; ha.n_out = nsl; ha.C = C; ha.stamps = nullptr;
    ha.D = 512; ha.GS = 1; ha.tiles_h = tiles_h; ha.lpg_sh = 6; ha.inv_d = 0.f;
    ha.philast = nullptr; ha.philast_stride = 0;
    ha.part = part.p; ha.part_stride = 12 * tiles_h; ha.Dreal = D; ha.base_real = g.base0_rel;
    int cnt = 0;
    const HotRange *ranges = hot_ranges(S, &cnt);
    const int NW = ranges[std::min(hot_range, cnt - 1)].NW, vper = NW / 4;
    const int nvwg = wgpcu() * ctx->prop.multiProcessorCount;
    int htpw = 4; while (htpw > 1 && (size_t)ceil_div((size_t)tiles_h, (size_t)htpw) * C < 4 * (size_t)nvwg) htpw >>= 1;
    // whole channels as units where they deal evenly over the grid (as the FM fix-up of the any-D form): a workgroup then
    // finishes the groups of its channels itself and the second launch is not needed
    const bool resident = channel_units();
    if (resident) htpw = tiles_h;
    else if (env_tpw) htpw = env_tpw;   // tuning hook
    ha.fin_groups = resident ? g.n_groups : 0; ha.fin_out = g.n_out; ha.fin_epi = epi;
    ha.tpw = htpw;
    ha.G = (int)ceil_div((size_t)tiles_h, (size_t)htpw); ha.U = ha.G * C;
    const int grid = (int)ceil_div((size_t)std::max(1, std::min(nvwg, std::max(ha.U, C))), (size_t)vper);
    const int gx = grid * vper;
    ha.dq = gx / ha.G; ha.dr = gx % ha.G;
    HotLaunch hl{(unsigned)grid, ctx->stream};
    hot_launch_anyd(S, kind, hot_range, inc != 0, HOT_EPI_PARTIAL, hl, ha, a);
    if (resident) return true;
    BigdArgs f;
    f.part = part.p; f.part_stride = 12 * tiles_h;
    f.D = D; f.base0_rel = g.base0_rel; f.N = (int)N; f.n_groups = g.n_groups; f.n_out = g.n_out; f.epi = epi; f.C = C;
    f.acc_old = a0.acc_old; f.acc_new = a0.acc_new; f.fm_old = a0.fm_old; f.fm_new = a0.fm_new;
    f.out = out_dev; f.out_stride = (long)out_stride;
    hipLaunchKernelGGL(iqbb_bigd_finish_kernel, dim3((unsigned)ceil_div((size_t)g.n_groups, (size_t)256), (unsigned)C), dim3(256), 0, ctx->stream, f);
    return true;
  }
  // whole channels as the persistent grid's units (any-D forms with FM: the slices' first angle differences, large-decimation
  // form: the groups, finished inside the hot kernel instead of by a second launch): where they deal evenly over the grid
  // virtual (4-wave) workgroups per CU of the persistent grids: 4 = four waves per SIMD; the 33-step class (orders 258 ... 513) runs
  // one 8-wave workgroup per CU (SDRHIP_IQBB_WGPCU: tuning hook)
  int wgpcu() const { return env_wgpcu ? env_wgpcu : S >= 33 ? 2 : 4; }
  bool long_filter() const { return (path == 1 || path == 3) && (S >= 33 || i8); }   // no general MFMA kernel (orders 258 ... 513; the int8 chain): short calls run the VALU kernel
  bool channel_units() const {
    if (env_fm_resident >= 0) return env_fm_resident != 0;   // tuning / test hook (SDRHIP_IQBB_FM_RESIDENT=0|1)
    const size_t nvwg = (size_t)wgpcu() * (size_t)ctx->prop.multiProcessorCount, rounds = ceil_div((size_t)C, nvwg);
    return (size_t)C * 100 >= rounds * nvwg * 97;
  }
  bool real_anyd() const { return path == 4 && D != R; }   // real input at a decimation other than 8: the any-D forms of the hot kernel
  int hot_kind() const { return real ? HOT_REAL : i8 ? HOT_CS8 : in_cu8 ? HOT_CU8 : HOT_CS16; }
  bool anyd_plan() const {
    if (!(((path == 3 && !real) || real_anyd()) && use_hot && hot_range >= 0 && S <= 33)) return false;
    if (D >= 9 && D <= 512) return true;
    if (i8) return false;   // (the int8 chain: decimation 8 and 9 ... 512 on the matrix cores)
    // decimations 1 ... 7: the small-decimation form, where its sample arrays fit a workgroup's LDS (iqbb_hot.hpp, SD, hot_sd_nw)
    return D >= 1 && D <= 7 && hot_launch_sd(S, hot_kind(), hot_range, inc != 0, epi, HotLaunch{0, nullptr}, HotArgs{}, IqbbArgs{}, true) != 0;
  }
  bool launch_anyd_call(IqbbArgs &a, const Geometry &g, const uint32_t *in_dev, size_t N, size_t in_stride, void *out_dev,
                        size_t out_stride) {
    const int kind = hot_kind(), halo = hot_halo(S, kind), win = hot_win(S, kind);
    const int GS = 512 / D, OGh = 4 * GS;   // (no recomputed overlap group: FM's first angles come through philast)
    const int tiles_h = (int)ceil_div((size_t)g.n_groups, (size_t)OGh);
    // the hot slices of a channel are ONE range of slice numbers s = 4 * tile + w (slice_is_hot is monotone in s): every
    // wave finds its own tile range inside [t_lo, t_hi) in the kernel
    auto slice_hot = [&](int sl) { return slice_is_hot(halo, win, g.base0_rel, OGh, 0, (int)N, g.n_out, sl >> 2, sl & 3, D, GS); };
    int s_lo = 0, s_hi = 4 * tiles_h;
    while (s_lo < s_hi && !slice_hot(s_lo)) s_lo++;
    while (s_hi > s_lo && !slice_hot(s_hi - 1)) s_hi--;
    if (s_hi - s_lo < 16) return false;
    const int t_lo = s_lo >> 2, t_hi = (s_hi + 3) >> 2;
    HotArgs ha{};
    ha.in = in_dev; ha.in_stride = (long)in_stride; ha.out = out_dev; ha.out_stride = (long)out_stride;
    ha.tapfrag = real ? tapfrag.p : in_cu8 ? tapfrag_rot.p : tapfrag_hot.p;   // (path 4's only set is the permuted one)
    ha.lut = lut.p; ha.inc = inc; ha.n0_lo = (uint32_t)(n0 - phase0); ha.negative = negative;
    ha.base0_rel = g.base0_rel; ha.OG = OGh; ha.ovl = 0; ha.t_lo = t_lo; ha.t_hi = t_hi; ha.cre = cre; ha.cim = cim;
    ha.N = (int)N; ha.n_out = g.n_out; ha.C = C; ha.stamps = nullptr;
    ha.D = D; ha.GS = GS; ha.tiles_h = tiles_h;
    { int lpg = 1; while (2 * lpg <= 64 && 2 * lpg * GS <= 64) lpg *= 2; int sh = 0; while ((1 << sh) < lpg) sh++; ha.lpg_sh = sh; }
    ha.inv_d = (float)((1.0 / D) * (1.0 - 1.0 / 1048576.0));
    ha.philast = nullptr; ha.philast_stride = 4 * tiles_h; ha.part = nullptr; ha.fin_groups = 0;
    if (epi == SDRHIP_EPI_FM) {   // (sized at create for max_in: no allocation, i.e. no device-wide synchronisation, on the call path)
      SDRHIP_REQUIRE(philast.n >= (size_t)C * 4 * tiles_h, SDRHIP_E_SIZE, "philast holds %zu entries, the call needs %zu", philast.n, (size_t)C * 4 * tiles_h);
      ha.philast = philast.p;
    }
#ifdef K1_STAMPS
    if (!k1_stamps.p) { k1_stamps.alloc(32768 * 16); k1_stamps.zero(ctx->stream); }
    ha.stamps = k1_stamps.p;
#endif
    // (17 K steps: 8- or 16-wave workgroups = 2 or 4 virtual ones sharing the tap fragments, as the /8 kernel of that class)
    int cnt = 0;
    const HotRange *ranges = hot_ranges(S, &cnt);
    int NW = ranges[std::min(hot_range, cnt - 1)].NW;
    if (D < 8) NW = hot_launch_sd(S, kind, hot_range, inc != 0, epi, HotLaunch{0, nullptr}, HotArgs{}, IqbbArgs{}, true);   // (the small-decimation form picks its own: hot_sd_nw)
    const int vper = NW / 4;
    const int nvwg = wgpcu() * ctx->prop.multiProcessorCount;   // (SDRHIP_IQBB_WGPCU: tuning hook — waves per SIMD)
    int htpw = 4; while (htpw > 1 && (size_t)ceil_div((size_t)tiles_h, (size_t)htpw) * C < 4 * (size_t)nvwg) htpw >>= 1;
    // FM: the slices whose first output is neither out[0] nor out[1] (their own rules) and is emitted lack the angle of the
    // slice before them. Where whole channels deal evenly over the persistent grid (within 3 %: 1024 or 8192 channels on 1024
    // workgroups) a unit is a channel — one workgroup then finishes every slice of a channel and completes those outputs
    // itself, behind a barrier at its end; otherwise (few channels: units of 4 tiles keep the grid full) a second launch does.
    const int fix_lo = GS == 1 ? 2 : 1, fix_hi = epi == SDRHIP_EPI_FM ? (int)ceil_div((size_t)g.n_out, (size_t)GS) : 0;
    const bool resident = fix_hi > fix_lo && channel_units();
    if (resident) htpw = tiles_h;
    else if (env_tpw) htpw = env_tpw;   // tuning hook
    a.fix_lo = resident ? fix_lo : 0; a.fix_hi = resident ? fix_hi : 0;
    // ... and where they do not (few channels, or a count that leaves the grid uneven): the owners of neighbouring slices
    // can complete the first output between them by a handshake through device memory (iqbb_hot.hpp, hs_exchange) — ONE
    // launch, in builds with -DK1_FM_HANDSHAKE under SDRHIP_IQBB_FM_HANDSHAKE=1. NOT in the shipped build: the write-through stores and the wait in front of the loads
    // cost each wave two memory round trips per unit, and the second, tiny launch (iqbb_fm_fixup_kernel) it replaces costs
    // nothing that can be measured — per buffer, host to host, on ONE channel: 42.2 us against 41.4 us (sdr_fm's plan),
    // device-resident 11.7 against 11.2 us; at 128 channels 27 against 21 us (profiles/r17_*).
#ifdef K1_FM_HANDSHAKE
    const bool handshake = fix_hi > fix_lo && !resident && env_fm_handshake == 1 && hs.p != nullptr && 4 * tiles_h + 2 <= hs_stride;
#else
    const bool handshake = false;   // (the kernels are built without it: see iqbb_hot.hpp, hs_exchange)
#endif
    if (handshake) {
      if (++hs_seq <= 0) hs_seq = 1;
      ha.hs = hs.p; ha.hs_stride = hs_stride; ha.hs_seq = hs_seq;
    }
    ha.tpw = htpw;
    ha.G = (int)ceil_div((size_t)tiles_h, (size_t)htpw); ha.U = ha.G * C;
    const int grid = (int)ceil_div((size_t)std::max(1, std::min(nvwg, std::max(ha.U, C))), (size_t)vper);   // (every channel's cold slices need a taker too)
    const int gx = grid * vper;   // virtual workgroups
    ha.dq = gx / ha.G; ha.dr = gx % ha.G;
    HotLaunch hl{(unsigned)grid, ctx->stream};
    if (D < 8) (void)hot_launch_sd(S, kind, hot_range, inc != 0, epi, hl, ha, a, false);
    else hot_launch_anyd(S, kind, hot_range, inc != 0, epi, hl, ha, a);
    if (!resident && !handshake && fix_hi > fix_lo)
      hipLaunchKernelGGL(iqbb_fm_fixup_kernel, dim3((unsigned)ceil_div((size_t)(fix_hi - fix_lo), (size_t)256), (unsigned)C), dim3(256), 0, ctx->stream,
                         reinterpret_cast<short *>(out_dev), (long)out_stride, philast.p, 4 * tiles_h, fix_lo, fix_hi, GS, C);
    return true;
  }

  void launch(const uint32_t *in_dev, size_t N, size_t in_stride, void *out_dev, size_t out_stride, size_t *n_out) {
    ctx->use();
    if (N == 0) { if (n_out) *n_out = 0; return; }   // empty buffer: nothing moves (src/baseband.hh:200)
    const Geometry g = geometry(N);
    SDRHIP_REQUIRE(out_stride >= (size_t)g.n_out, SDRHIP_E_SIZE, "out_stride %zu < outputs %d", out_stride, g.n_out);
    IqbbArgs a;
    a.in = in_dev; a.in_stride = (long)in_stride; a.in_cu8 = in_cu8; a.in_real = real; a.i8 = i8;
    a.hist_old = hist[par].p; a.hist_new = hist[par ^ 1].p; a.HH = HH;
    a.acc_old = acc[par].p; a.acc_new = acc[par ^ 1].p;
    const bool fm_flip = (epi == SDRHIP_EPI_FM && g.n_out >= 2);
    a.fm_old = fm[par_fm].p; a.fm_new = fm[par_fm ^ 1].p;
    a.taps = taps.p; a.lut = lut.p; a.inc = inc; a.negative = negative;
    a.OP = OP; a.D = D; a.N = (int)N; a.n0_lo = (uint32_t)(n0 - phase0);   // (the kernels use n0_lo for the LUT phase only)
    a.base0_rel = g.base0_rel; a.n_groups = g.n_groups; a.n_out = g.n_out; a.extra0 = g.extra0;
    a.CG = CG; a.OG = OG; a.ovl = ovl; a.CGr = (CG + 3) & ~3;
    a.out = out_dev; a.out_stride = (long)out_stride; a.epilogue = epi;
    a.tapfrag = tapfrag.p; a.cre = cre; a.cim = cim; a.ah_mask = (unsigned)ah_mask;   // (the general kernels: at most 17 steps)
#ifdef SDRHIP_AH_FULL
    a.ah_mask = ~0u;   // tuning: never skip
#endif
    const int tiles = (int)ceil_div((size_t)g.n_groups, (size_t)OG);
    // MFMA path: one workgroup walks `tpw` consecutive tiles so that the tap fragments are fetched once;
    // keep >= ~8 workgroups per CU in flight for balance
    int tpw = 1;
    const bool mf8 = (path == 1 && !long_filter()) || (path == 4 && D == R);   // the lane-owned-group kernels (decimation 8) with a general form
    if (mf8) { tpw = 8; while (tpw > 1 && (size_t)ceil_div((size_t)tiles, (size_t)tpw) * C < 2048) tpw >>= 1; }
    if (env_tpw && mf8) tpw = env_tpw;   // tuning hook
    a.tiles = tiles; a.tpw = tpw; a.bt_hi = 0; a.fix_lo = a.fix_hi = 0;
    a.lpg = 1; while (a.lpg < 64 && a.lpg * 8 < D) a.lpg <<= 1;
    dim3 grid((unsigned)ceil_div((size_t)tiles, (size_t)tpw), C), block(TPB);
    if (path == 4 && D == R && use_hot && hot_range >= 0 && tiles >= 3 && launch_hot_call(a, g, in_dev, N, in_stride, out_dev, out_stride, tiles)) {
      // (real int16 input: the hot kernel took the whole call)
    } else if (real_anyd() && anyd_plan() && launch_anyd_call(a, g, in_dev, N, in_stride, out_dev, out_stride)) {
      // (real int16 input at any other decimation: the hot kernel's any-D / small-decimation form took the whole call)
    } else if (real_anyd()) {   // ... its short calls: the VALU kernel
      hipLaunchKernelGGL((iqbb_i16_kernel<false, true>), grid, block, lds_bytes, ctx->stream, a);
    } else if (path == 4) {
#define SDRHIP_MFR(S_) do { if (inc != 0) hipLaunchKernelGGL((bb_real_mfma_kernel<S_, true>), grid, block, lds_bytes, ctx->stream, a); \
                             else hipLaunchKernelGGL((bb_real_mfma_kernel<S_, false>), grid, block, lds_bytes, ctx->stream, a); } while (0)
      switch (S) {
        case 3: SDRHIP_MFR(3); break;
        case 5: SDRHIP_MFR(5); break;
        default: SDRHIP_MFR(9); break;
      }
#undef SDRHIP_MFR
    } else if (long_filter() && path == 1 && use_hot && hot_range >= 0 && tiles >= 3 && launch_hot_call(a, g, in_dev, N, in_stride, out_dev, out_stride, tiles)) {
      // (orders 258 ... 513 at decimation 8: the hot kernel's 33-step class took the whole call)
    } else if (long_filter() && path == 3 && bigd_plan() && launch_bigd_call(a, g, in_dev, N, in_stride, out_dev, out_stride)) {
    } else if (long_filter() && path == 3 && anyd_plan() && launch_anyd_call(a, g, in_dev, N, in_stride, out_dev, out_stride)) {
    } else if (long_filter()) {   // ... their short calls: the VALU kernel (the class has no general matrix kernel)
      if (fast8) hipLaunchKernelGGL((iqbb_i16_kernel<true, false>), grid, block, lds_bytes, ctx->stream, a);
      else hipLaunchKernelGGL((iqbb_i16_kernel<false, false>), grid, block, lds_bytes, ctx->stream, a);
    } else if (path == 3 && bigd_plan() && launch_bigd_call(a, g, in_dev, N, in_stride, out_dev, out_stride)) {
      // (decimations above 256: partial sums by the hot kernel, groups finished in its last step or by a second, small launch)
    } else if (path == 3 && anyd_plan() && launch_anyd_call(a, g, in_dev, N, in_stride, out_dev, out_stride)) {
      // (the hot kernel's any-D form took the whole call)
    } else if (path == 3) {
      int tpw3 = 8; while (tpw3 > 1 && (size_t)ceil_div((size_t)tiles, (size_t)tpw3) * C < 2048) tpw3 >>= 1;
      if (env_tpw) tpw3 = env_tpw;   // tuning hook
      a.tpw = tpw3;
      dim3 grid3((unsigned)ceil_div((size_t)tiles, (size_t)tpw3), C);
#define SDRHIP_MFG(S_) do { if (in_cu8 && inc != 0) hipLaunchKernelGGL((iqbb_i16_mfmag_kernel<S_, true, true>), grid3, block, lds_bytes, ctx->stream, a); \
                             else if (in_cu8) hipLaunchKernelGGL((iqbb_i16_mfmag_kernel<S_, false, true>), grid3, block, lds_bytes, ctx->stream, a); \
                             else if (inc != 0) hipLaunchKernelGGL((iqbb_i16_mfmag_kernel<S_, true, false>), grid3, block, lds_bytes, ctx->stream, a); \
                             else hipLaunchKernelGGL((iqbb_i16_mfmag_kernel<S_, false, false>), grid3, block, lds_bytes, ctx->stream, a); } while (0)
      if (lds_bytes > 64 * 1024) {   // (17 K steps, small decimations; once per kernel and device: allow_lds_max)
        allow_lds_max(&iqbb_i16_mfmag_kernel<17, true, true>, lds_bytes); allow_lds_max(&iqbb_i16_mfmag_kernel<17, false, true>, lds_bytes);
        allow_lds_max(&iqbb_i16_mfmag_kernel<17, true, false>, lds_bytes); allow_lds_max(&iqbb_i16_mfmag_kernel<17, false, false>, lds_bytes);
      }
      switch (S) {
        case 2: SDRHIP_MFG(2); break;
        case 3: SDRHIP_MFG(3); break;
        case 5: SDRHIP_MFG(5); break;
        case 9: SDRHIP_MFG(9); break;
        default: SDRHIP_MFG(17); break;
      }
#undef SDRHIP_MFG
    } else if (path == 1 && use_hot && hot_range >= 0 && tiles >= 3 && launch_hot_call(a, g, in_dev, N, in_stride, out_dev, out_stride, tiles)) {
      // (complex<int16> or complex<uint8> input, any filter length of path 1: the hot kernel took the whole call)
    } else if (path == 1 && !in_cu8) {
      // complex<int16> input, calls too short for the hot kernel (or SDRHIP_IQBB_HOT=0): the general kernel, raw tiles
      // by LDS-DMA (LDS: table | one plane pair | raw tile | tap fragments)
      const size_t PLWd = (2 * (size_t)(TI + OP) + 64 + 31) / 32 * 8, quads = (TI + OP + 4) / 4;
      const size_t ldsd = (256 + 2 * PLWd + 4 * ((quads + 63) / 64 * 64)) * 4 + (size_t)S * 2 * 64 * 16;
#define SDRHIP_MFD(S_) do { if (inc != 0) hipLaunchKernelGGL((iqbb_i16_mfma_dma_kernel<S_, true>), grid, block, ldsd, ctx->stream, a); \
                             else hipLaunchKernelGGL((iqbb_i16_mfma_dma_kernel<S_, false>), grid, block, ldsd, ctx->stream, a); } while (0)
      switch (S) {
        case 2: SDRHIP_MFD(2); break;
        case 3: SDRHIP_MFD(3); break;
        case 5: SDRHIP_MFD(5); break;
        case 9: SDRHIP_MFD(9); break;
        default: SDRHIP_MFD(17); break;
      }
#undef SDRHIP_MFD
    } else if (path == 1) {
      // complex<uint8> input, calls too short for the hot kernel: the one-plane general kernel (its LDS: two single planes
      // instead of two pairs)
      const size_t lds1 = lds_bytes - 2 * (((2 * (size_t)(TI + OP) + 64 + 31) / 32 * 8) * 4);
#define SDRHIP_MF(S_) do { if (inc != 0) hipLaunchKernelGGL((iqbb_i16_mfma_kernel<S_, true, true>), grid, block, lds1, ctx->stream, a); \
                            else hipLaunchKernelGGL((iqbb_i16_mfma_kernel<S_, false, true>), grid, block, lds1, ctx->stream, a); } while (0)
      switch (S) {
        case 2: SDRHIP_MF(2); break;
        case 3: SDRHIP_MF(3); break;
        case 5: SDRHIP_MF(5); break;
        case 9: SDRHIP_MF(9); break;
        default: SDRHIP_MF(17); break;
      }
#undef SDRHIP_MF
    } else if (real) {
      if (fast8) hipLaunchKernelGGL((iqbb_i16_kernel<true, true>), grid, block, lds_bytes, ctx->stream, a);
      else hipLaunchKernelGGL((iqbb_i16_kernel<false, true>), grid, block, lds_bytes, ctx->stream, a);
    } else if (fast8) hipLaunchKernelGGL((iqbb_i16_kernel<true, false>), grid, block, lds_bytes, ctx->stream, a);
    else hipLaunchKernelGGL((iqbb_i16_kernel<false, false>), grid, block, lds_bytes, ctx->stream, a);
    SDRHIP_CHECK_HIP(hipGetLastError());
    par ^= 1;
    if (fm_flip) par_fm ^= 1;
    n0 += N;
    if (n_out) *n_out = (size_t)g.n_out;
    last_args = a;
  }
  IqbbArgs last_args{};   // the argument block of the last launch (launch_multi's fix-up reads the same call)
  int last_multi_left = -1;   // buffer boundaries the last one-launch multi call left to iqbb_fm_multi_fixup_kernel (-1: none made yet)
  std::vector<int> mb_q;  // launch_multi -> launch: the long call's output indices of the buffers' first elements; on return: those no
                          // kernel of the call has dealt with (the hot /8 kernel takes the ones inside its hot slices, launch_hot_call)

  // B consecutive buffers of nb samples per channel in one launch (sdrhip_iqbb_i16_process_dev_multi). IQBaseBand's own
  // state runs on across buffers, so the baseband part IS one long call; the buffer boundaries are only visible to the FM
  // demodulator (iqbb_fm_multi_fixup_kernel). Where a buffer would emit fewer than two values (FMDemod then leaves its
  // angle untouched), with the int8 chain, or beyond 64 buffers: one launch per buffer, same results.
  void launch_multi(const uint32_t *in_dev, size_t B, size_t nb, size_t in_stride, void *out_dev, size_t out_stride, size_t *counts, size_t *total) {
    std::vector<size_t> qs(B + 1, 0);   // outputs in front of buffer j
    for (size_t j = 1; j <= B; j++) qs[j] = (size_t)geometry(j * nb).n_out;
    for (size_t j = 0; j < B && counts; j++) counts[j] = qs[j + 1] - qs[j];
    if (total) *total = qs[B];
    bool one_launch = B >= 2 && !i8 && B <= 64;
    if (epi == SDRHIP_EPI_FM) for (size_t j = 0; j < B; j++) if (qs[j + 1] - qs[j] < 2) one_launch = false;
    if (!one_launch) {
      const size_t ib = in_elem_bytes(), ob = out_elem_bytes();
      for (size_t j = 0; j < B; j++)
        launch(reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(in_dev) + j * nb * ib), nb, in_stride,
               reinterpret_cast<char *>(out_dev) + qs[j] * ob, out_stride, nullptr);
      return;
    }
    mb_q.clear();
    if (epi == SDRHIP_EPI_FM) for (size_t j = 1; j < B; j++) mb_q.push_back((int)qs[j]);
    try { launch(in_dev, B * nb, in_stride, out_dev, out_stride, nullptr); } catch (...) { mb_q.clear(); throw; }
    std::vector<int> left;
    left.swap(mb_q);   // (the boundaries no hot slice wrote: border slices, a boundary on a slice's last lane, plans of other kernels)
    last_multi_left = (int)left.size();
    if (left.empty()) return;
    MultiFix f{};
    f.nb = (int)left.size();
    for (size_t j = 0; j < left.size(); j++) f.q[j] = left[j];
    hipLaunchKernelGGL(iqbb_fm_multi_fixup_kernel, dim3((unsigned)left.size(), (unsigned)C), dim3(256), 0, ctx->stream, last_args, f);
    SDRHIP_CHECK_HIP(hipGetLastError());
  }
};

namespace {

int create_baseband(sdrhip_ctx *ctx, const int32_t *taps, int order, const int32_t *lut, uint32_t lut_inc,
                    int negative, int decim, int channels, size_t max_in, int epilogue, bool real,
                    sdrhip_iqbb_i16 **out, bool i8 = false) {
  return guarded([&] {
    SDRHIP_REQUIRE(ctx && taps && lut && out, SDRHIP_E_INVALID, "NULL argument");
    *out = nullptr;
    SDRHIP_REQUIRE(order >= 1 && order <= MAX_ORDER, SDRHIP_E_UNSUPPORTED, "order %d outside [1,%d]", order, MAX_ORDER);
    SDRHIP_REQUIRE(decim >= 1, SDRHIP_E_INVALID, "decim %d < 1", decim);
    SDRHIP_REQUIRE(channels >= 1 && channels <= 65535, SDRHIP_E_INVALID, "channels %d outside [1,65535]", channels);
    SDRHIP_REQUIRE(max_in >= 1 && max_in < (size_t(1) << 30), SDRHIP_E_SIZE, "max_in %zu outside [1,2^30)", max_in);
    SDRHIP_REQUIRE(epilogue >= SDRHIP_EPI_NONE && epilogue <= SDRHIP_EPI_USB, SDRHIP_E_INVALID, "bad epilogue %d", epilogue);
    const int ovl = epilogue == SDRHIP_EPI_FM ? 1 : 0;
    // Decimations beyond the general kernels' tile (TI / (1 + ovl): 2048, with FM 1024) exist as the hot kernel's large-
    // decimation form only, which then takes every call, however short (its cold path serves any slice): complex<int16> /
    // complex<uint8> input, up to 257 taps whose high bytes fit int8, up to 32768 (D * D must not wrap: box_div).
    const bool beyond = TI / decim - ovl < 1;
    const int CG = beyond ? 1 + ovl : TI / decim;   // (beyond: a placeholder — no general kernel ever runs such a plan)
    SDRHIP_REQUIRE(!beyond || (!real && !i8 && order <= 257 && decim <= 32768), SDRHIP_E_UNSUPPORTED,
                   "decim %d too large (max %d; complex<int16> / complex<uint8> plans of up to 257 taps: 32768)", decim, TI / (1 + ovl));
    for (int i = 0; i < 2 * order; i++) {
      if (real) SDRHIP_REQUIRE(taps[i] > -(1 << 23) && taps[i] < (1 << 23), SDRHIP_E_UNSUPPORTED,
                               "tap %d = %d exceeds 24 bits", i / 2, taps[i]);
      else SDRHIP_REQUIRE(taps[i] >= -32767 && taps[i] <= 32767, SDRHIP_E_UNSUPPORTED,
                          "tap %d = %d does not fit the packed int16 path", i / 2, taps[i]);
    }
    for (int i = 0; i < 256; i++)
      SDRHIP_REQUIRE(lut[i] > -(1 << 23) && lut[i] < (1 << 23), SDRHIP_E_UNSUPPORTED, "LUT entry %d = %d exceeds 24 bits", i / 2, lut[i]);
    ctx->use();
    sdrhip_iqbb_i16 *h = new sdrhip_iqbb_i16;
    try {
      h->ctx = ctx; h->order = order; h->D = decim; h->C = channels; h->epi = epilogue;
      h->negative = negative ? 1 : 0; h->inc = lut_inc; h->max_in = max_in; h->real = real ? 1 : 0; h->i8 = i8 ? 1 : 0;
      SDRHIP_REQUIRE(!i8 || epilogue == SDRHIP_EPI_NONE || epilogue == SDRHIP_EPI_FM, SDRHIP_E_UNSUPPORTED,
                     "the int8 chain is IQBaseBand<int8_t> (-> FMDemod<int8_t,int16_t>): epilogue NONE or FM");
      // path: the int8-MFMA formulations need D == 8, order <= 257 (32x32x32) / 153 (16x16x64) and tap high bytes that fit int8
      { const char *d = getenv("SDRHIP_IQBB_HOT"); if (d && d[0] == '0') h->use_hot = false; }
      // (orders 258 ... 513: the hot kernel's 33-step class only — no general matrix kernel; SDRHIP_IQBB_HOT=0 leaves them the VALU kernel)
      const int mfma_max_order = h->use_hot ? 513 : 257;
      // (IQBaseBand<int8_t>: hot forms only, up to 129 taps, decimation 8 and 9 ... 512)
      const bool i8_hot = i8 && h->use_hot && order <= 129;
      bool mfma_ok = !real && (!i8 || i8_hot) && (decim == R) && (order <= mfma_max_order);
      auto high_byte = [](int v) { const int al = ((v + 128) & 255) - 128; return (v - al) >> 8; };
      for (int i = 0; i < 2 * order && mfma_ok; i++)   // both v and -v are packed (Kr, -Ki / Ki, Kr)
        if (high_byte(taps[i]) > 127 || high_byte(-taps[i]) > 127) mfma_ok = false;
      { const char *e = getenv("SDRHIP_IQBB_TPW"); if (e) h->env_tpw = std::max(1, atoi(e)); }
      { const char *e = getenv("SDRHIP_IQBB_WGPCU"); if (e) h->env_wgpcu = std::max(1, atoi(e)); }
      { const char *e = getenv("SDRHIP_IQBB_FM_RESIDENT"); if (e) h->env_fm_resident = atoi(e) != 0; }
      { const char *e = getenv("SDRHIP_IQBB_FM_HANDSHAKE"); if (e) h->env_fm_handshake = atoi(e) != 0; }
      const char *force = getenv("SDRHIP_IQBB_PATH");   // "valu": test hook (the VALU kernel for every plan)
      if (force && !strcmp(force, "valu")) mfma_ok = false;
      h->path = mfma_ok ? 1 : 0;
      // path 3: the same matrix part for any decimation (measured ahead of the VALU kernel at every order tried, 9 ... 257 taps)
      bool mfmag_ok = !real && (!i8 || (i8_hot && decim >= 9 && decim <= 512)) && decim != R && order <= mfma_max_order && (order <= 257 || decim >= 9);   // (the 33-step class has no small-decimation form)
      for (int i = 0; i < 2 * order && mfmag_ok; i++)
        if (high_byte(taps[i]) > 127 || high_byte(-taps[i]) > 127) mfmag_ok = false;
      if (force && !strcmp(force, "valu")) mfmag_ok = false;
      if (h->path == 0 && mfmag_ok) h->path = 3;
      SDRHIP_REQUIRE(!beyond || (h->path == 3 && h->use_hot), SDRHIP_E_UNSUPPORTED,
                     "decim %d too large (max %d) for this plan: beyond it only the hot kernel's large-decimation form exists (tap high bytes within int8, SDRHIP_IQBB_HOT / SDRHIP_IQBB_PATH unset)",
                     decim, TI / (1 + ovl));
      h->bigd_always = beyond;
      // path 4: real input on the matrix cores — D == 8, at most 9 K steps, taps that fit two byte planes
      // ... at decimation 8 the lane-owned-group kernel; at any other up to 512 the hot kernel's any-D forms for the long calls
      // and the VALU kernel for the short ones (src/baseband.hh:305-529: any sub_sample)
      if (real && !i8 && (decim == R || (decim <= 512 && h->use_hot)) && order <= 273 && !(force && !strcmp(force, "valu"))) {
        bool fits = true;
        for (int i = 0; i < 2 * order && fits; i++) if (high_byte(taps[i]) > 127 || high_byte(-taps[i]) > 127) fits = false;   // (the rule set_taps applies)
        if (fits) h->path = 4;
      }
      if (h->path == 4) {
        h->S = order <= 81 ? 3 : order <= 145 ? 5 : 9;   // the hot kernel's filter-length classes (K steps of 32 real samples)
        h->OP = decim == R ? 32 * h->S - 15 : (int)ceil_div((size_t)order, (size_t)TAPC) * TAPC;   // (other decimations: the VALU kernel's tap chunks)
      } else if (h->path == 1 || h->path == 3) {
        h->S = order <= 17 ? 2 : order <= 33 ? 3 : order <= 65 ? 5 : order <= 129 ? 9 : order <= 257 ? 17 : 33;
        h->OP = (h->S >= 33 || i8) ? (int)ceil_div((size_t)order, (size_t)TAPC) * TAPC : 16 * (h->S - 1) + 1;   // (33 steps, int8 chain: the VALU kernel's tap chunks — it runs the short calls)
      } else {
        h->OP = (int)ceil_div((size_t)order, (size_t)TAPC) * TAPC;
      }
      h->HH = h->OP;   // one more than the FIR needs: reset(keep_history) must see the whole ring
      if (h->path == 4) h->HH = std::max(h->OP, 32 * h->S - 15);   // (the hot kernel's window reaches 32 S - 16 samples back)
      if (h->long_filter()) h->HH = std::max(h->OP, 16 * (h->S - 1) + 1);   // (... 16 (S - 1) back)
      h->CG = CG; h->ovl = ovl; h->OG = CG - ovl;
      if (h->path == 1 || (h->path == 4 && decim == R)) { h->OG = 4 * (64 - ovl); h->CG = h->OG + ovl; }   // every wave recomputes its own FM overlap group
      h->fast8 = (decim == R);
      if ((h->path == 4 && decim != R) || h->long_filter()) {   // (the VALU kernel's: it runs this plan's short calls)
        const size_t XS = TI + h->OP + 8;
        h->lds_bytes = (XS + 256 + 2 * (((size_t)h->CG + 3) & ~(size_t)3)) * 4 + (h->fast8 ? 0 : (size_t)TI * 8);
      } else if (h->path == 4) {
        h->lds_bytes = 1024 + (size_t)h->S * 2 * 64 * 16 + 4 * 2 * (size_t)(512 + 32 * h->S);
      } else if (h->path == 3) {
        const size_t PLW = (2 * (size_t)(TI + h->OP) + 64 + 31) / 32 * 8;
        h->lds_bytes = (2 * PLW + 256) * 4 + (size_t)h->S * 2 * 64 * 16 + (size_t)(TI + TI / 16 + 2) * 8 + 2 * (size_t)((CG + 3) & ~3) * 4;
        // (17 K steps at decimations up to 5 need 65 ... 79 KB: two workgroups per CU of gfx950's 160 KB — the launch raises the
        // kernel's dynamic-LDS limit; beyond that: back to the VALU kernel)
        if (h->lds_bytes > 80 * 1024) {
          h->path = 0; h->OP = (int)ceil_div((size_t)order, (size_t)TAPC) * TAPC; h->HH = h->OP;
          const size_t XS = TI + h->OP + 8;
          h->lds_bytes = (XS + 256 + 2 * ((CG + 3) & ~3)) * 4 + (h->fast8 ? 0 : (size_t)TI * 8);
        }
      } else if (h->path == 1) {
        const size_t PLW = (2 * (size_t)(TI + h->OP) + 64 + 31) / 32 * 8;
        h->lds_bytes = (4 * PLW + 256) * 4 + (size_t)h->S * 2 * 64 * 16;
      } else {
        const size_t XS = TI + h->OP + 8;
        h->lds_bytes = (XS + 256 + 2 * ((CG + 3) & ~3)) * 4 + (h->fast8 ? 0 : (size_t)TI * 8);
      }
      SDRHIP_REQUIRE(h->lds_bytes <= (h->path == 3 ? 80 : 64) * 1024, SDRHIP_E_UNSUPPORTED, "LDS budget exceeded (%zu B)", h->lds_bytes);
      h->load_taps(taps);
      h->lut.alloc(128); h->lut.upload(reinterpret_cast<const int2 *>(lut), 128, ctx->stream);
      for (int p = 0; p < 2; p++) {
        h->hist[p].alloc((size_t)channels * h->HH); h->hist[p].zero(ctx->stream);
        h->acc[p].alloc(channels); h->acc[p].zero(ctx->stream);
        h->fm[p].alloc(channels); h->fm[p].zero(ctx->stream);
      }
      h->max_out = max_in / decim + 2;
      if (epilogue == SDRHIP_EPI_FM && (h->path == 3 || h->real_anyd()) && decim >= 1 && decim <= 512) {   // any-D hot forms: one angle per slice of the longest call (launch_anyd_call)
        const size_t GS = 512 / (size_t)decim, tiles_h = ceil_div(max_in / (size_t)decim + 2, 4 * GS);
        h->philast.alloc((size_t)channels * 4 * tiles_h + 1024);
#ifdef K1_FM_HANDSHAKE
        h->hs_stride = (int)(4 * tiles_h + 2);
        h->hs.alloc((size_t)2 * channels * h->hs_stride); h->hs.zero(ctx->stream);   // (call numbers start at 1: a zeroed entry matches none)
#endif
      }
      { const char *e = getenv("SDRHIP_IQBB_BIGD_MIN"); if (e) { h->bigd_min = std::max(257, atoi(e)); h->bigd_skip_lo = 513; } }   // tuning / test hook
      { const char *e = getenv("SDRHIP_IQBB_BIGD_ALWAYS"); if (e && atoi(e) != 0 && decim >= 257) { h->bigd_always = true; h->bigd_skip_lo = 513; } }   // test hook: short calls too
      if (beyond) { h->bigd_min = 257; h->bigd_skip_lo = 513; }   // (such a plan has no other kernel, whatever the hooks say)
      if (h->path == 3 && decim >= 257 && decim >= h->bigd_min)   // (launch_bigd_call: 3 sums per slice of 512 samples)
        h->part.alloc((size_t)channels * 12 * ceil_div(ceil_div(max_in, (size_t)512), (size_t)4) + 64);
      SDRHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    } catch (...) { delete h; throw; }
    *out = h;
  });
}

}  // namespace

// What IQBaseBand::_reconfigure leaves of the FIR ring (src/baseband.hh:175-177: _ring_offset = 0, the ring's contents
// stay where they lie), as the history rows (HH_dst entries per channel, oldest first, the newest at the end) of a plan
// that starts counting at zero: with P = (samples so far) mod order the node afterwards reads the old ring ROTATED —
// the apparent history, oldest first, is ring[1..order-1], ring[i] = t[order-P+i] (i < P) or t[i-P] (i >= P),
// t = the last `order` samples in time order (the tail of the source plan's rows).
static std::vector<uint32_t> reconfigured_ring(const sdrhip_iqbb_i16 *src, int HH_dst) {
  const int order = src->order, HH = src->HH, P = src->ring_offset();
  hipStream_t st = src->ctx->stream;
  std::vector<uint32_t> old((size_t)src->C * HH), neu((size_t)src->C * HH_dst, 0u);
  SDRHIP_CHECK_HIP(hipMemcpyAsync(old.data(), src->hist[src->par].p, old.size() * 4, hipMemcpyDeviceToHost, st));
  SDRHIP_CHECK_HIP(hipStreamSynchronize(st));
  for (int c = 0; c < src->C; c++) {
    const uint32_t *t = old.data() + (size_t)c * HH + (HH - order);
    uint32_t *d = neu.data() + (size_t)c * HH_dst + (HH_dst - (order - 1));
    for (int k = 0; k + 1 < order; k++) { const int i = k + 1; d[k] = i < P ? t[order - P + i] : t[i - P]; }
  }
  return neu;
}

extern "C" {

int sdrhip_iqbb_i16_create(sdrhip_ctx *ctx, const int32_t *taps, int order, const int32_t *lut, uint32_t lut_inc,
                           int negative, int decim, int channels, size_t max_in, int epilogue,
                           sdrhip_iqbb_i16 **out) {
  return create_baseband(ctx, taps, order, lut, lut_inc, negative, decim, channels, max_in, epilogue, false, out);
}

int sdrhip_bb_i16_create(sdrhip_ctx *ctx, const int32_t *taps, int order, const int32_t *lut, uint32_t lut_inc,
                         int negative, int decim, int channels, size_t max_in, int epilogue,
                         sdrhip_iqbb_i16 **out) {
  return create_baseband(ctx, taps, order, lut, lut_inc, negative, decim, channels, max_in, epilogue, true, out);
}

int sdrhip_iqbb_i8_create(sdrhip_ctx *ctx, const int32_t *taps, int order, const int32_t *lut, uint32_t lut_inc,
                          int negative, int decim, int channels, size_t max_in, int epilogue,
                          sdrhip_iqbb_i16 **out) {
  return create_baseband(ctx, taps, order, lut, lut_inc, negative, decim, channels, max_in, epilogue, false, out, true);
}

int sdrhip_iqbb_i16_path(sdrhip_iqbb_i16 *h, int *path) {
  return guarded([&] {
    SDRHIP_REQUIRE(h && path, SDRHIP_E_INVALID, "NULL argument");
    *path = h->path;
  });
}

int sdrhip_iqbb_i16_plan_info(sdrhip_iqbb_i16 *h, int *info, int n) {
  return guarded([&] {
    SDRHIP_REQUIRE(h && info && n >= 8, SDRHIP_E_INVALID, "info must hold 8 ints");
    int cnt = 0;
    const HotRange *rg = hot_ranges(h->S, &cnt);
    const bool hot = h->hot_range >= 0 && h->hot_range < cnt;
    info[0] = h->path; info[1] = h->S; info[2] = hot ? rg[h->hot_range].S0 : 0; info[3] = hot ? rg[h->hot_range].NH : 0;
    info[4] = hot ? rg[h->hot_range].NW : 4; info[5] = h->hot_kind(); info[6] = h->OP; info[7] = h->HH;
    if (n >= 9) info[8] = h->last_multi_left;
  });
}

int sdrhip_iqbb_i16_kernel_names(sdrhip_iqbb_i16 *h, char *buf, size_t len) {
  return guarded([&] {
    SDRHIP_REQUIRE(h && buf && len, SDRHIP_E_INVALID, "NULL argument");
    const char *nm = "iqbb_i16_kernel";
    if (h->real_anyd() && !h->anyd_plan()) nm = "iqbb_i16_kernel";
    else if (h->real_anyd()) {   // (calls of a few tiles: the VALU kernel)
      nm = h->D < 8 ? "iqbb_hot_sd_kernel" : "iqbb_hot_anyd_kernel";
      if (h->epi == SDRHIP_EPI_FM && !h->channel_units()) nm = h->D < 8 ? "iqbb_hot_sd_kernel,iqbb_fm_fixup_kernel" : "iqbb_hot_anyd_kernel,iqbb_fm_fixup_kernel";
    }
    else if (h->path == 4 && h->use_hot && h->hot_range >= 0) nm = "iqbb_hot_kernel";   // (calls of < 3 tiles: the general kernel)
    else if (h->path == 4) nm = "bb_real_mfma_kernel";
    else if (h->path == 3 && h->bigd_plan()) nm = h->channel_units() ? "iqbb_hot_anyd_kernel" : "iqbb_hot_anyd_kernel,iqbb_bigd_finish_kernel";   // (calls of a few tiles: the general kernel)
    else if (h->path == 3 && h->anyd_plan()) {   // (calls of a few tiles: the general kernel "iqbb_i16_mfmag_kernel")
      nm = h->D < 8 ? "iqbb_hot_sd_kernel" : "iqbb_hot_anyd_kernel";
      if (h->epi == SDRHIP_EPI_FM && !h->channel_units() && (h->env_fm_handshake != 1 || !h->hs.p))
        nm = h->D < 8 ? "iqbb_hot_sd_kernel,iqbb_fm_fixup_kernel" : "iqbb_hot_anyd_kernel,iqbb_fm_fixup_kernel";
    }
    else if (h->path == 3) nm = "iqbb_i16_mfmag_kernel";
    else if (h->path == 1 && h->use_hot && h->hot_range >= 0) nm = "iqbb_hot_kernel";   // (calls of < 3 tiles: the general kernel)
    else if (h->path == 1 && h->in_cu8) nm = "iqbb_i16_mfma_kernel";
    else if (h->path == 1) nm = "iqbb_i16_mfma_dma_kernel";
    snprintf(buf, len, "%s", nm);
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
    Range roctx_range("sdrhip_iqbb_i16_process_dev");
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    SDRHIP_REQUIRE(n_in <= h->max_in, SDRHIP_E_SIZE, "n_in %zu > max_in %zu", n_in, h->max_in);
    if (n_in == 0) { if (n_out) *n_out = 0; return; }
    SDRHIP_REQUIRE(in_dev && out_dev, SDRHIP_E_INVALID, "NULL buffer");
    if (in_stride == 0) in_stride = n_in;
    SDRHIP_REQUIRE(in_stride >= n_in, SDRHIP_E_SIZE, "in_stride %zu < n_in %zu", in_stride, n_in);
    if (out_stride == 0) out_stride = (size_t)h->geometry(n_in).n_out;
    require_disjoint(in_dev, in_stride, n_in, h->in_elem_bytes(), out_dev, out_stride, (size_t)h->geometry(n_in).n_out,
                     h->out_elem_bytes(), (size_t)h->C);
    h->launch(reinterpret_cast<const uint32_t *>(in_dev), n_in, in_stride, out_dev, out_stride, n_out);
  });
}

int sdrhip_iqbb_i16_process_dev_multi(sdrhip_iqbb_i16 *h, const int16_t *in_dev, size_t n_buffers, size_t n_per_buffer, size_t in_stride,
                                      void *out_dev, size_t out_stride, size_t *n_out_per_buffer, size_t *n_out_total) {
  return guarded([&] {
    Range roctx_range("sdrhip_iqbb_i16_process_dev_multi");
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    const size_t n_in = n_buffers * n_per_buffer;
    SDRHIP_REQUIRE(n_buffers == 0 || n_in / n_buffers == n_per_buffer, SDRHIP_E_SIZE, "n_buffers * n_per_buffer overflows");
    SDRHIP_REQUIRE(n_in <= h->max_in, SDRHIP_E_SIZE, "n_buffers * n_per_buffer = %zu > max_in %zu", n_in, h->max_in);
    if (n_in == 0) {
      for (size_t j = 0; j < n_buffers && n_out_per_buffer; j++) n_out_per_buffer[j] = 0;
      if (n_out_total) *n_out_total = 0;
      return;
    }
    SDRHIP_REQUIRE(in_dev && out_dev, SDRHIP_E_INVALID, "NULL buffer");
    if (in_stride == 0) in_stride = n_in;
    SDRHIP_REQUIRE(in_stride >= n_in, SDRHIP_E_SIZE, "in_stride %zu < n_buffers * n_per_buffer %zu", in_stride, n_in);
    const size_t no = (size_t)h->geometry(n_in).n_out;
    if (out_stride == 0) out_stride = no;
    SDRHIP_REQUIRE(out_stride >= no, SDRHIP_E_SIZE, "out_stride %zu < outputs %zu", out_stride, no);
    require_disjoint(in_dev, in_stride, n_in, h->in_elem_bytes(), out_dev, out_stride, no, h->out_elem_bytes(), (size_t)h->C);
    h->ctx->use();
    h->launch_multi(reinterpret_cast<const uint32_t *>(in_dev), n_buffers, n_per_buffer, in_stride, out_dev, out_stride, n_out_per_buffer, n_out_total);
  });
}

int sdrhip_iqbb_i16_process(sdrhip_iqbb_i16 *h, const int16_t *in_host, size_t n_in, size_t in_stride,
                            void *out_host, size_t out_stride, size_t *n_out) {
  return guarded([&] {
    Range roctx_range("sdrhip_iqbb_i16_process");
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
    const size_t ib = h->in_elem_bytes();
    copy_h2d_rows(h->ctx, h->stage_in.p, n_in * ib, in_host, in_stride * ib, n_in * ib, h->C);
    const size_t eb = h->out_elem_bytes();
    size_t produced = 0;
    h->launch(h->stage_in.p, n_in, n_in, h->stage_out.p, h->max_out * 4 / eb, &produced);
    copy_d2h_rows(h->ctx, out_host, out_stride * eb, h->stage_out.p, h->max_out * 4, produced * eb, h->C);
    SDRHIP_CHECK_HIP(hipStreamSynchronize(h->ctx->stream));
    if (n_out) *n_out = produced;
  });
}

int sdrhip_iqbb_i16_set_input_format(sdrhip_iqbb_i16 *h, int format) {
  return guarded([&] {
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    SDRHIP_REQUIRE(format == SDRHIP_IN_CS16 || format == SDRHIP_IN_CU8, SDRHIP_E_INVALID, "bad input format %d", format);
    SDRHIP_REQUIRE(!h->real && !h->i8, SDRHIP_E_INVALID, "the real-input and the int8 baseband take their own sample type only");
    SDRHIP_REQUIRE(h->n0 == 0, SDRHIP_E_INVALID, "the input format can only change before the first buffer / after a reset");
    h->in_cu8 = format == SDRHIP_IN_CU8;
  });
}

int sdrhip_iqbb_i16_reset(sdrhip_iqbb_i16 *h, int keep_history) {
  return guarded([&] {
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    h->ctx->use();
    hipStream_t st = h->ctx->stream;
    // bit 0 of keep_history: the FIR ring survives (rotated, above); bit 1: so does the fused FMDemod's last angle
    // — IQBaseBand::_reconfigure does not touch the FMDemod node behind it, whose config() (and with it the reset of
    // _last_value, src/demod.hh:210) only runs when the Config the baseband propagates CHANGES (src/node.cc:98-105)
    const bool keep_fm = (keep_history & 2) != 0;
    keep_history &= 1;
    for (int p = 0; p < 2; p++) { h->acc[p].zero(st); if (!keep_fm) h->fm[p].zero(st); }
    if (!keep_history) {
      for (int p = 0; p < 2; p++) h->hist[p].zero(st);
    } else if (h->ring_offset() != 0) {
      const std::vector<uint32_t> neu = reconfigured_ring(h, h->HH);
      SDRHIP_CHECK_HIP(hipMemcpyAsync(h->hist[h->par].p, neu.data(), neu.size() * 4, hipMemcpyHostToDevice, st));
      SDRHIP_CHECK_HIP(hipStreamSynchronize(st));
    }
    h->n0 = 0; h->phase0 = 0; h->ring_n0 = 0; h->ring_off0 = 0;
  });
}

int sdrhip_iqbb_i16_adopt_state(sdrhip_iqbb_i16 *h, sdrhip_iqbb_i16 *from, int what) {
  return guarded([&] {
    SDRHIP_REQUIRE(h && from && h != from, SDRHIP_E_INVALID, "two distinct handles are needed");
    SDRHIP_REQUIRE(h->n0 == 0, SDRHIP_E_INVALID, "the adopting plan must be fresh (no buffer processed since create / reset)");
    SDRHIP_REQUIRE(h->C == from->C && h->real == from->real && h->i8 == from->i8 && h->in_cu8 == from->in_cu8, SDRHIP_E_INVALID,
                   "plans differ in channels (%d / %d) or sample kind", h->C, from->C);
    SDRHIP_REQUIRE(h->ctx->device == from->ctx->device, SDRHIP_E_INVALID, "plans live on different devices");
    const bool ring = (what & SDRHIP_KEEP_RING) != 0, fmk = (what & SDRHIP_KEEP_FM) != 0, stream = (what & SDRHIP_KEEP_COUNTERS) != 0;
    SDRHIP_REQUIRE(!ring || h->order == from->order, SDRHIP_E_INVALID, "SDRHIP_KEEP_RING needs equal orders (%d / %d)", h->order, from->order);
    SDRHIP_REQUIRE(!stream || h->D == from->D, SDRHIP_E_INVALID, "SDRHIP_KEEP_COUNTERS needs equal decimations (%d / %d)", h->D, from->D);
    SDRHIP_REQUIRE(!fmk || (h->epi == SDRHIP_EPI_FM && from->epi == SDRHIP_EPI_FM), SDRHIP_E_INVALID,
                   "SDRHIP_KEEP_FM needs the FM epilogue on both plans (%d / %d)", h->epi, from->epi);
    h->ctx->use();
    hipStream_t st = h->ctx->stream;
    SDRHIP_CHECK_HIP(hipStreamSynchronize(from->ctx->stream));   // (the source plan's last launch wrote the state read here)
    const size_t C = (size_t)h->C;
    if (ring && stream) {
      // the stream goes on: the last `order` samples in time order, as they lie at the end of the source's rows
      const int order = h->order;
      SDRHIP_CHECK_HIP(hipMemcpy2DAsync(h->hist[h->par].p + (h->HH - order), (size_t)h->HH * 4, from->hist[from->par].p + (from->HH - order),
                                        (size_t)from->HH * 4, (size_t)order * 4, C, hipMemcpyDeviceToDevice, st));
    } else if (ring) {
      const std::vector<uint32_t> neu = reconfigured_ring(from, h->HH);
      SDRHIP_CHECK_HIP(hipMemcpyAsync(h->hist[h->par].p, neu.data(), neu.size() * 4, hipMemcpyHostToDevice, st));
      SDRHIP_CHECK_HIP(hipStreamSynchronize(st));
    }
    if (fmk) SDRHIP_CHECK_HIP(hipMemcpyAsync(h->fm[h->par_fm].p, from->fm[from->par_fm].p, C * sizeof(short), hipMemcpyDeviceToDevice, st));
    if (stream) {   // decimator window (position and partial sum), sample counter and LUT phase go on
      SDRHIP_CHECK_HIP(hipMemcpyAsync(h->acc[h->par].p, from->acc[from->par].p, C * sizeof(int2), hipMemcpyDeviceToDevice, st));
      h->n0 = from->n0; h->phase0 = from->phase0;
      // _ring_offset goes on from where it stands; a position at or beyond a SHORTER new ring is an out-of-bounds write
      // in the reference (undefined): here, as in the oracle's set_order, the offset restarts at 0
      h->ring_n0 = h->n0; h->ring_off0 = from->ring_offset() < h->order ? from->ring_offset() : 0;
    }
    SDRHIP_CHECK_HIP(hipStreamSynchronize(st));
  });
}

int sdrhip_iqbb_i16_set_taps(sdrhip_iqbb_i16 *h, const int32_t *taps) {
  return guarded([&] {
    SDRHIP_REQUIRE(h && taps, SDRHIP_E_INVALID, "NULL argument");
    auto high_byte = [](int v) { const int al = ((v + 128) & 255) - 128; return (v - al) >> 8; };
    for (int i = 0; i < 2 * h->order; i++) {
      if (h->real) SDRHIP_REQUIRE(taps[i] > -(1 << 23) && taps[i] < (1 << 23), SDRHIP_E_UNSUPPORTED, "tap %d = %d exceeds 24 bits", i / 2, taps[i]);
      else SDRHIP_REQUIRE(taps[i] >= -32767 && taps[i] <= 32767, SDRHIP_E_UNSUPPORTED, "tap %d = %d does not fit the packed int16 path", i / 2, taps[i]);
      if (h->path >= 1)
        SDRHIP_REQUIRE(high_byte(taps[i]) <= 127 && high_byte(-taps[i]) <= 127, SDRHIP_E_UNSUPPORTED,
                       "tap %d = %d does not fit the plan's int8 byte planes: create a new plan", i / 2, taps[i]);
    }
    h->ctx->use();
    h->load_taps(taps);   // stream-ordered after the launches already enqueued
  });
}

int sdrhip_iqbb_i16_set_shift(sdrhip_iqbb_i16 *h, uint32_t lut_inc, int negative) {
  return guarded([&] {
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    h->inc = lut_inc; h->negative = negative ? 1 : 0;
    h->phase0 = h->n0;   // _lut_count = 0 (src/freqshift.hh:86): the phase is a closed form of (n - phase0)
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

#ifdef K1_STAMPS
extern "C" int sdrhip_debug_k1_stamps(sdrhip_iqbb_i16 *h, unsigned long long *out, int words) {   // diagnostic builds only (tools/build_variant.sh)
  if (!h || !h->k1_stamps.p) return -3;
  (void)hipStreamSynchronize(h->ctx->stream);
  return hipMemcpy(out, h->k1_stamps.p, (size_t)words * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
#endif
