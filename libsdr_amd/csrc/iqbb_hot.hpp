// iqbb_hot.hpp — K1's HOT kernel: IQBaseBand<int16_t> (D = 8) as the int8-MFMA block-Toeplitz GEMM, one launch per
// call, for EVERY filter length path 1 serves (S = 2, 3, 5, 9, 17, 33 K steps: orders up to 17 / 33 / 65 / 129 / 257 / 513) and
// for complex<int16> and complex<uint8> (AutoCast fused) input.
//
// Replaces (reference, file:line): IQBaseBand<int16_t>::_process / _filter_ring  src/baseband.hh:198-236,
// FreqShiftBase<int16_t>::applyFrequencyShift  src/freqshift.hh:58-74, FM/AM/USB demodulators  src/demod.hh:73-76,
// 156-161,242-254 (+ src/math.hh:31-40), AutoCast<complex<int16>> on complex<uint8>  src/autocast.hh:62,187-194.
//
// Structure (algebra: iqbb_i16.hip, "MFMA formulation"):
//   * A wave slice = 512 consecutive samples of one channel (64 decimation groups; lane (n, h) ends up owning group
//     2n + h). A slice is HOT when nothing about it touches the call's borders (slice_is_hot): its window of
//     512 + 16(S-1) samples lies inside the input, none of its groups is the call's first or last emitted one.
//   * Hot loop, wave-autonomous, no workgroup barrier: the wave's raw window comes in by LDS-DMA
//     (global_load_lds_dwordx4) into one of TWO wave-private window buffers, is split IN PLACE into the byte planes
//     the MFMA operands are read from (cs16: low plane | high plane; cu8: the one high plane, (b + 129) mod 256), and
//     only this wave reads them. While slice i is split, multiplied and finished in buffer i&1, the DMA of slice i+1
//     lands in the other buffer: it is issued at the top of iteration i and has a whole slice period to arrive
//     (round 2's single raw area could only be refilled after the split, half a period before it was needed).
//   * The K-step range that needs the taps' high byte plane is a compile-time parameter [S0, S0 + NH): the K loop is
//     straight-line code with the operands of step s+1 in flight while step s's MFMAs issue.
//   * Persistent grid of (virtual) 4-wave workgroups, static equal-length work units, rotating wave priority.
//     A real workgroup is NW waves (4, 8 or 16): NW/4 virtual workgroups sharing one LDS copy of the tap fragments and
//     the rotation table — long filters' fragments (17-34 KB) would otherwise cap the CU at 2 waves per SIMD.
//   * Cold phase, same launch: every virtual workgroup, done with its units, takes the channels vb, vb + vgx, ...:
//     wave w computes slice w of tile 0 and of the tiles from bt_hi on where that slice is cold — window by clamped
//     ordinary loads (history / input / zeros), the same LDS-resident tap fragments and table, the general epilogue
//     (group_sum<EDGE> + group_finish: edge masks, carry, first-sample quirk, state) — and rolls the FIR history.
//
// Forms of the same body (template flags; windows, K loop, grid and cold phase are shared):
//   D = 8             iqbb_hot_kernel       lane (n, h) owns one group of a slice; FM recomputes one overlap group per slice
//   D = 9 ... 512     iqbb_hot_anyd_kernel  DG: a slice holds 512 / D whole groups, summed out of in-register prefix sums and
//                                           parked; FM's first angle difference of a slice completed afterwards (in the
//                                           kernel's last step where whole channels are its units, else by a tiny launch)
//   D = 2 ... 7       iqbb_hot_sd_kernel    SD: 73 ... 256 groups per slice out of an LDS array, finished per slice
//   D = 257 ... 32768 iqbb_hot_anyd_kernel  PART: slices of 512 samples whatever the groups; a slice leaves the sums of its
//                                           stretches between group boundaries, one lane per group finishes (same choice)
#pragma once
#include <atomic>
#include "iqbb_common.hpp"

#include <type_traits>

namespace sdrhip {

// ---- geometry shared by host and device ----------------------------------------------------------------------
// input kinds: complex<int16> (one dword per sample), complex<uint8> (AutoCast fused), real int16 (BaseBand<int16_t>: the
// element stream is the sample stream itself — a block advances by 16 elements instead of 32, a tap row by 1 instead of 2)
// ... and complex<int8> (IQBaseBand<int8_t>, src/sdr.hh:225-240's chain: the sample IS one signed byte plane — no conversion at all)
enum { HOT_CS16 = 0, HOT_CU8 = 1, HOT_REAL = 2, HOT_CS8 = 3 };
constexpr bool hot_one_plane(int in) { return in == HOT_CU8 || in == HOT_CS8; }
constexpr int hot_halo(int S, int in) { return in == HOT_REAL ? 32 * S - 16 : 16 * (S - 1); }   // samples before the slice's first one
constexpr int hot_win(int S, int in) { return 512 + hot_halo(S, in) + (in == HOT_REAL ? 16 : 0); }   // samples in a wave window (real: whole 16-byte pieces)
// bytes per byte plane: complex kinds 2 per sample (+ one chunk pair: both parity halves 16-byte aligned), real 1 per sample
constexpr int hot_plb(int S, int in) { return in == HOT_REAL ? hot_win(S, in) : 2 * hot_win(S, in) + 32; }
constexpr int hot_bufb(int S, int in) { return hot_one_plane(in) ? hot_plb(S, in) : 2 * hot_plb(S, in); }   // one window buffer
// -DK1_PAIR (tuning variant, D = 8, complex<int16>, 4-wave workgroups; build with -DK1_MINWAVES=2, run with
// SDRHIP_IQBB_WGPCU=2): a wave works on TWO slices at a time — consecutive tiles of its unit — so that one read of a tap
// fragment feeds the MFMAs of both (6 accumulators, 4 window buffers per wave, 2 waves per SIMD)
#ifdef K1_PAIR
constexpr bool hot_pair(int in, int NW, bool dg) { return !dg && in == HOT_CS16 && NW == 4; }
#else
constexpr bool hot_pair(int, int, bool) { return false; }
#endif
constexpr int hot_lds_bytes(int S, int NH, int in, int NW, bool wide, bool pair = false) {
  return (wide ? 4096 : 1024) + (S + NH) * 1024 + NW * (pair ? 4 : 2) * hot_bufb(S, in);
}
// 4 waves per SIMD: 160 KB / workgroups per CU. (S = 33, orders 258 ... 513: 33 ... 66 KB of tap fragments and 1024-sample windows —
// ONE 8-wave workgroup per CU, two waves per SIMD: its 70 ... 132 MFMAs per slice keep the matrix pipe busy behind one
// wave's vector phase)
constexpr int hot_lds_cap(int NW, bool pair = false, int S = 0) { return S >= 33 ? 163840 : NW == 4 ? (pair ? 81920 : 40960) : NW == 8 ? 81920 : 163840; }
// 4 KB rotation table ({Lx, Ly, -Ly, 0} x 256: one SDWA shift makes the address, no subtraction) while it fits
#ifdef K1_NARROW_LUT   // (tuning: the 1 KB {Lx, Ly} table everywhere — half the rotation's LDS bytes, one subtraction more per sample)
constexpr bool hot_wide(int, int, int, int, int = 0, bool = false) { return false; }
#else
constexpr bool hot_wide(int S, int NH, int in, int NW, int extra = 0, bool pair = false) { return hot_lds_bytes(S, NH, in, NW, true, pair) + extra <= hot_lds_cap(NW, pair, S); }
#endif
// the any-D form's additions to a workgroup's LDS: parked group sums, and the rotated samples where the window buffer is too small
// (rot = false, plans without a shift: the unrotated values have 18 bits — two arrays of dwords instead of one of int16 pairs)
// (the team sums stay in registers: only the parked group sums, 512 bytes per wave)
constexpr int hot_anyd_extra(int, int, bool = true, int NW = 4) { return NW * 512; }

// the small-decimation form (2 <= D <= 7, SD): the wave's 512 rotated samples go through a 2 KB per-wave LDS array (the
// slice's own window buffer when that is large enough; two arrays without a shift: the unrotated values have 18 bits)
constexpr int hot_sd_extra(int S, int in, bool rot, int NW = 4) { return (hot_bufb(S, in) >= 2048 ? 0 : NW * 2048) + (rot ? 0 : NW * 2048); }
constexpr bool hot_sd_fits(int S, int NH, int in, bool rot, int NW = 4) { return hot_lds_bytes(S, NH, in, NW, false) + hot_sd_extra(S, in, rot, NW) <= hot_lds_cap(NW); }
// ... and the workgroup it runs in: the class's own size (HotRange::NW) or, where the arrays do not fit beside the tap
// fragments (9 K steps without a shift; 17 K steps without one), the next one that shares a copy of the fragments among
// twice the waves — 0: none fits
constexpr int hot_sd_nw(int S, int NH, int in, bool rot, int nw0) {
  for (int nw = nw0; nw <= 16; nw *= 2) if (hot_sd_fits(S, NH, in, rot, nw)) return nw;
  return 0;
}

struct HotRange { int S0, NH, NW; };
// per S: centred high-plane ranges, narrowest first; the last one covers every step
constexpr HotRange hot_ranges_2[] = {{0, 2, 4}};
constexpr HotRange hot_ranges_3[] = {{1, 2, 4}, {0, 3, 4}};
constexpr HotRange hot_ranges_5[] = {{1, 3, 4}, {0, 5, 4}};
constexpr HotRange hot_ranges_9[] = {{3, 3, 4}, {2, 5, 4}, {1, 7, 4}, {0, 9, 4}};
constexpr HotRange hot_ranges_17[] = {{6, 5, 8}, {4, 9, 8}, {0, 17, 16}};
// (33 steps: the taps sit at the END of the 513-tap window — zero-padded at the front — so the steps that need the high plane
// move with the order: 14 ... 18 at 513 taps, 17 ... 22 at 400, 20 ... 25 at 300, 22 ... 26 at 258)
constexpr HotRange hot_ranges_33[] = {{12, 9, 8}, {16, 9, 8}, {18, 9, 8}, {8, 17, 8}, {0, 33, 8}};
inline const HotRange *hot_ranges(int S, int *count) {
  switch (S) {
    case 2: *count = 1; return hot_ranges_2;
    case 3: *count = 2; return hot_ranges_3;
    case 5: *count = 2; return hot_ranges_5;
    case 9: *count = 4; return hot_ranges_9;
    case 17: *count = 3; return hot_ranges_17;
    case 33: *count = 5; return hot_ranges_33;
    default: *count = 0; return nullptr;
  }
}

struct HotLaunch { unsigned grid; hipStream_t stream; };
// one function per translation unit (iqbb_hot_s*.hip): `range` indexes hot_ranges(S)
void hot_launch_s2(int in, int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &);
void hot_launch_s3(int in, int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &);
void hot_launch_s5(int in, int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &);
void hot_launch_s9_cs16(int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &);
void hot_launch_s9_cu8(int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &);
void hot_launch_s17_cs16(int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &);
void hot_launch_s17_cu8(int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &);
void hot_launch_s33_cs16(int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &);   // (orders 258 ... 513)
void hot_launch_s33_cu8(int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &);
void hot_launch_anyd33_cs16(int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &);
void hot_launch_anyd33_cu8(int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &);
void hot_launch_cs8(int S, int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &);        // complex<int8>: S = 2, 3, 5 or 9; no demodulator or FM
void hot_launch_anyd_cs8(int S, int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &);
void hot_launch_real(int S, int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &);   // S = 3, 5 or 9
// real input at any other decimation: the any-D form (9 ... 512) and the small-decimation form (1 ... 7; 0: its LDS does not fit)
void hot_launch_real_anyd(int S, int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &);
void hot_launch_real_anyd9(int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &);
int hot_launch_real_sd(int S, int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &, bool dry_run);
int hot_launch_real_sd9(int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &, bool dry_run);
void hot_launch_anyd(int S, int in, int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &);   // S = 2, 3, 5, 9 or 17, cs16 / cu8
void hot_launch_anyd9(int in, int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &);
void hot_launch_anyd17_cs16(int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &);   // (orders 130 ... 257: 8- and 16-wave workgroups, as the /8 kernel's)
void hot_launch_anyd17_cu8(int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &);
// decimations 2 ... 7 (S = 2, 3, 5, 9 or 17, cs16 / cu8): returns the waves per workgroup the plan runs in — 0: its LDS fits none (the general kernel runs it)
int hot_launch_sd(int S, int in, int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &, bool dry_run);   // (returns the waves per workgroup, 0: no form fits)
int hot_launch_sd9(int in, int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &, bool dry_run);
int hot_launch_sd17_cs16(int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &, bool dry_run);
int hot_launch_sd17_cu8(int range, bool rot, int epi, const HotLaunch &, const HotArgs &, const IqbbArgs &, bool dry_run);

}  // namespace sdrhip

namespace {

// AutoCast's high byte (b + 129) mod 256 of the four bytes of a dword in THREE instructions: a packed 16-bit add of 0x8100
// reaches the upper byte of each half and cannot carry into anything, the dword is rotated by one byte, the same add again
// reaches the other two. The result holds the bytes (b1', b2', b3', b0'): rotated by one inside the dword — the sample plane's
// K order inside every dword, which the complex<uint8> plans' tap fragments follow on the host (frag_rot, iqbb_i16.hip). The
// carry-free SWAR form it replaces (add129_bytes) took five; the plane conversion was 40 of the any-D kernel's 250 vector
// instructions per slice.
__device__ __forceinline__ uint32_t ror8(uint32_t x) { return __builtin_amdgcn_alignbit(x, x, 8); }
__device__ __forceinline__ uint32_t add129_rot(uint32_t x) {
  unsigned k = 0x81008100u;
  asm("" : "+s"(k));   // (VOP3P takes no literal: one scalar register)
  uint32_t t;
  asm("v_pk_add_u16 %0, %1, %2" : "=v"(t) : "v"(x), "s"(k));
  t = ror8(t);
  asm("v_pk_add_u16 %0, %1, %2" : "=v"(t) : "v"(t), "s"(k));
  return t;
}

// A wave slice (tile t, wave w of the tile's 4 waves: 64 groups, the first of them FM's overlap slot) is "hot" when
// nothing about it touches the call's borders: its window lies inside the input, none of its groups is the call's first
// (carry, the D+1 first window, FM's out[0] / out[1] rules) nor its last emitted one (that group hands the demodulator's
// angle and the window carry to the next call — state only the cold phase's epilogue writes; found by
// test_one_launch_kernel_random_long_calls), and all of them complete and are emitted.
// D: decimation, GS: whole groups in a slice's 512 samples (8 and 64 for the lane-owned-group kernels).
__host__ __device__ __forceinline__ bool slice_is_hot(int halo, int win, int base0_rel, int OG, int ovl, int N, int n_out, int tile, int w,
                                                      int D = 8, int GS = 64) {
  const int qf = tile * OG - ovl + w * (GS - ovl);   // the slice's first group (output index within the call)
  const int ws = base0_rel + qf * D - halo;          // its window's first sample (N < 2^30: no overflow)
  // (the call's groups 0 and 1 are cold: carry and D+1 window; FMDemod's out[1] takes the previous CALL's last angle)
  return ws >= 0 && ws + win <= N && qf >= (GS == 1 ? 2 : 1) && qf + GS - 1 < n_out - 1;
}

// DG: ANY decimation 9 <= D <= 512 (the reference's own receivers decimate by 62 and 125, examples/sdr_rec.cc:68,
// examples/sdr_fm.cc:40). The matrix part, the windows and the grid are the same; a slice's 512 samples hold GS = 512 / D
// whole groups (the slices of a wave advance by GS * D samples, so every slice starts on a group), the rotated samples are
// combined in registers (stageE: prefix sums, one team of lanes per group). The tap fragments are a set of their own with
// the /8 kernel's row permutation (a lane ends up with 8 consecutive samples; the plan's general kernel for short calls
// keeps its natural-order set). The cold phase is the same code with the samples outside the call masked and the team
// leaders applying the border rules themselves (cold_finish_gen).
// SD (with DG): decimations 2 ... 7 — the windows, the grid and the tap fragments are the any-D form's; a slice holds
// GS = 512 / D groups (73 ... 256: more than lanes), so the wave's rotated samples go through a 2 KB LDS array in stream
// order and lane l sums and FINISHES the groups l, l + 64, l + 128, l + 192 per slice (consecutive lanes, consecutive
// outputs: coalesced stores), the /8 kernel's per-slice finish instead of the parked one.
template <int S, int S0, int NH, bool ROT, int EPI, int IN, int NW, bool DG = false, bool SD = false>
__device__ __forceinline__ void iqbb_hot_body(const HotArgs &a, const IqbbArgs &b_kernarg) {
  // `b` serves the cold phase only. Left to itself the compiler loads the whole block at kernel entry and keeps it in
  // scalar registers ACROSS the hot loop — in the any-D forms that pushed the loop's own scalars into spill lanes (27 to
  // 49 v_readlane per slice in kernels that are bound by vector issue). It is copied out of the kernarg segment behind
  // the hot loop instead, through a pointer the compiler cannot see through (K1_LATE_B=0: the old behaviour, A/B).
#ifndef K1_LATE_B
#define K1_LATE_B 1
#endif
#ifndef K1_LATE_A
#define K1_LATE_A 0
#endif
  // Only the any-D forms do this: in the /8 kernels nothing was spilled to begin with, and there the late copies cost
  // vector registers instead (16 of them spilled to scratch in the USB and no-demodulator kernels of the 9-step class:
  // +11 % time, +28 % HBM traffic on the 127-tap USB workload when it was tried for all kernels).
  constexpr bool LATE_B = K1_LATE_B && DG, LATE_A = K1_LATE_A && DG;
  IqbbArgs b_late;
  const IqbbArgs &b = LATE_B ? b_late : b_kernarg;
  HotArgs a_late;                                        // (K1_LATE_A: likewise the cold phase's copy of `a`; off — it left 8 to
  const HotArgs &ac = LATE_A ? a_late : a;               //  24 bytes of scratch in a dozen any-D kernels for -0.8 % in the others)
  // CU8: ONE byte plane per sample (complex<uint8> after AutoCast: the high plane; complex<int8>: the sample itself) — two MFMAs
  // per K step, no low-plane accumulator. CS8 marks what differs for IQBaseBand<int8_t>: no plane conversion, the FIR value is
  // (hh << 8) + mid wrapped to int16 after the shift (src/baseband.hh:206), the rotation wraps to int16 and shifts by 8
  // (src/freqshift.hh:58-74, Traits<int8_t>::shift), outputs are int8.
  constexpr bool CS8 = IN == HOT_CS8, CU8 = IN == HOT_CU8 || CS8, REAL = IN == HOT_REAL;
  static_assert(!CS8 || (!SD && EPI != HOT_EPI_PARTIAL && (EPI == SDRHIP_EPI_NONE || EPI == SDRHIP_EPI_FM)), "int8 chain: /8 and any-D forms, no demodulator or FM");
  // PART (with DG): decimations above 512 — a group no longer fits a slice. The kernel runs the any-D form's geometry of
  // "decimation 512" from the call's first sample on (one pseudo-group per slice, cold slices where the window leaves the
  // call) and a slice, instead of finishing groups, leaves the sums of its stretches between the REAL group boundaries
  // (at most two inside 512 samples) in global memory; iqbb_bigd_finish_kernel adds each group's stretches, the carry and
  // the border rules, divides and demodulates.
  constexpr bool PART = EPI == HOT_EPI_PARTIAL;
  static_assert(!PART || (DG && !SD), "partial sums: a variant of the any-D form");
  static_assert(!SD || DG, "the small-decimation form is a variant of the any-D form");

  const int DD = DG ? a.D : 8, GS = DG ? a.GS : 64;   // decimation, whole groups per slice
  static_assert(S >= 2 && S0 >= 0 && NH >= 1 && S0 + NH <= S, "high-plane range inside the K loop");
  static_assert(!REAL || NW == 4, "real input: 4-wave workgroups");
  static_assert(NW == 4 || NW == 8 || NW == 16, "4-wave virtual workgroups");
  constexpr int HALO = hot_halo(S, IN), WIN = hot_win(S, IN), PLB = hot_plb(S, IN), HALF = PLB / 2, BUFB = hot_bufb(S, IN);
  constexpr int NPIECE = (IN == HOT_CS16 ? 4 * WIN : 2 * WIN) / 16;   // 16-byte pieces of the raw window (cs16: 4 samples, cu8 / real: 8)
  constexpr int KSB = REAL ? 32 : 16;   // plane bytes a K step advances by (real: a block's window moves 16 elements per column, 32 per step)
  constexpr int FSH = REAL ? 16 : 14;   // the FIR's right shift (Traits<int16_t>::shift for the real node, the literal 14 of IQBaseBand)
  constexpr int NDMA = (NPIECE + 63) / 64;                     // DMA wave-instructions per window
  constexpr int LASTL = NPIECE - 64 * (NDMA - 1);              // lanes of the last one
  static_assert(NDMA <= 4, "immediate offsets 0 / 1024 / 2048 / 3072");
  constexpr bool PAIR = hot_pair(IN, NW, DG);
  constexpr bool WIDE = SD ? false : hot_wide(S, NH, IN, NW, DG ? hot_anyd_extra(S, IN, ROT, NW) : 0, PAIR);   // (SD: the narrow table, the LDS goes to the sample arrays)
  static_assert(!SD || hot_sd_fits(S, NH, IN, ROT, NW), "small-decimation form: LDS budget");
  constexpr int NBUF = PAIR ? 4 : 2;   // window buffers per wave
  static_assert(hot_lds_bytes(S, NH, IN, NW, WIDE, PAIR) <= hot_lds_cap(NW, PAIR, S), "LDS budget for 4 waves per SIMD");
  constexpr int TBLW = WIDE ? 1024 : 256;   // dwords
  constexpr int TPBH = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  // Rotation table at LDS offset 0. WIDE: 256 entries of 16 bytes {Lx, Ly, -Ly, 0}, the 128-entry table twice — a table
  // address is then byte 1 of the phase counter << 4 with no mask, and the complex product needs no subtraction.
  // Otherwise 128 entries {Lx, Ly}. A negative shift reads the table backwards: it is stored reversed.
  int2 *lut_s = reinterpret_cast<int2 *>(smem);
  v4i *taps_s = reinterpret_cast<v4i *>(smem + TBLW);
  // (the wave index is wave-uniform but the compiler cannot know: through readfirstlane the slice bookkeeping runs on
  // the scalar unit)
  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63, n = l & 31, h = l >> 5;
  const int wv = w & 3;                                                   // slice of a tile this wave computes
  const int bx = (int)blockIdx.x * (NW / 4) + (w >> 2), gx = (int)gridDim.x * (NW / 4);   // virtual workgroup, virtual grid
  // tap fragments in LDS: the S low-plane fragments, then only the NH high-plane fragments of steps [S0, S0 + NH)
  char *wbase = reinterpret_cast<char *>(smem + TBLW + (S + NH) * 64 * 4) + w * (NBUF * BUFB);
  char *pendb = reinterpret_cast<char *>(smem + TBLW + (S + NH) * 64 * 4) + NW * (NBUF * BUFB) + w * 512;    // (DG) the wave's parked group sums (64 x int2)
  char *gscb = reinterpret_cast<char *>(smem + TBLW + (S + NH) * 64 * 4) + NW * (NBUF * BUFB) + (SD ? 0 : NW * 512) + w * 2048;   // (DG) the wave's rotated samples (when its window buffer is too small; SD: no parked sums in front)
  char *gscb2 = gscb + NW * 2048;   // (DG, no shift, small window buffers) the second array
#ifdef K1_ABL_ASAME   // (tuning ablation, results wrong: every step reads the SAME fragment values — K1_ABL_AREG's operand data with the reads kept)
  for (int i = tid; i < S * 64; i += TPBH) taps_s[i] = a.tapfrag[64 + (i & 63)];
  for (int i = tid; i < NH * 64; i += TPBH) taps_s[S * 64 + i] = a.tapfrag[(2 * S0) * 64 + (i & 63)];
#else
  for (int i = tid; i < S * 64; i += TPBH) taps_s[i] = a.tapfrag[(2 * (i >> 6) + 1) * 64 + (i & 63)];
  for (int i = tid; i < NH * 64; i += TPBH) taps_s[S * 64 + i] = a.tapfrag[(2 * (S0 + (i >> 6))) * 64 + (i & 63)];
#endif
  if (tid < 256) {
    const int2 e = a.lut[(tid & 127) ^ (a.negative ? 127 : 0)];
    if (WIDE) reinterpret_cast<v4i *>(smem)[tid] = v4i{e.x, e.y, -e.y, 0};
    else if (tid < 128) lut_s[tid] = e;
  }

  const int OGw = GS - a.ovl, gw = wv * OGw;
  // PERSISTENT grid: gx virtual workgroups (4 per CU) stay resident and walk the work units u = bx, bx + gx, ...;
  // unit u = (channel u / G, tile group u % G of `tpw` consecutive tiles). All units are the same length, so the static
  // assignment balances (one workgroup per unit left ~3 of 4 workgroups per CU resident: the unsynchronised waves of a
  // workgroup finish up to 2x apart and its LDS stays allocated until the slowest is done).
  // For a given slice index wv the hot tiles of a channel are ONE range [hl, hh) — slice_is_hot's conditions are
  // monotone in the tile — the same for every channel: found once per wave, so that the loop below tests tile numbers
  // instead of evaluating the conditions per slice.
  int hl = a.t_lo, hh = a.t_hi;
  while (hl < hh && !slice_is_hot(HALO, WIN, a.base0_rel, a.OG, a.ovl, a.N, a.n_out, hl, wv, DD, GS)) hl++;
  while (hh > hl && !slice_is_hot(HALO, WIN, a.base0_rel, a.OG, a.ovl, a.N, a.n_out, hh - 1, wv, DD, GS)) hh--;
  int u = bx;
  int c = u / a.G, g = u - c * a.G;   // (one division per wave, at start; afterwards (c, g) advance by (dq, dr))
  int tile = max(a.t_lo + g * a.tpw, hl), tend = min(a.t_lo + g * a.tpw + a.tpw, hh);
  // advance (u, c, g) to the next unit that holds a hot tile of this wave's slice (units wholly outside [hl, hh) — the
  // first or last unit of a channel, at most — are skipped)
  auto next_unit = [&](int &u_, int &c_, int &g_, int &tile_, int &tend_) {
    do {
      u_ += gx; c_ += a.dq; g_ += a.dr;
      if (g_ >= a.G) { g_ -= a.G; c_++; }
      tile_ = max(a.t_lo + g_ * a.tpw, hl); tend_ = min(a.t_lo + g_ * a.tpw + a.tpw, hh);
    } while (u_ < a.U && tile_ >= tend_);
  };
  if (u < a.U && tile >= tend) next_unit(u, c, g, tile, tend);
  // Per channel: scalar bases of the wave's input window and output row; per slice they advance by whole tiles.
  // The window of tile t starts at sample base0_rel + (t * OG - ovl + gw) * 8 - HALO of the channel's row. The lane's
  // global pointer: scalar base + the lane's 32-bit byte offset, ONE 64-bit vector add per slice; the DMA pieces differ
  // by the instruction's immediate offset, which applies to the global AND the LDS address alike.
  const uint32_t lane_byte = 16u * (uint32_t)l;
  constexpr int SB = IN == HOT_CS16 ? 4 : 2;                   // bytes per input sample
  constexpr int OB = (EPI == SDRHIP_EPI_NONE && !CS8) ? 4 : 2;   // bytes per output element (int8 chain: complex<int8>)
  const int tile_in_bytes = a.OG * DD * SB, tile_out_bytes = a.OG * OB;
  const uint32_t tile_cnt = (uint32_t)(a.OG * DD) * a.inc;      // LUT phase counter advance per tile
  const uint32_t cnt0 = (a.n0_lo + (uint32_t)(a.base0_rel + (gw - a.ovl) * DD)) * a.inc;   // ... of the wave's first sample in tile 0
  auto chan_src = [&](int c_) { return reinterpret_cast<const char *>(a.in) + ((long)c_ * a.in_stride + (a.base0_rel + (gw - a.ovl) * DD - HALO)) * SB; };
  auto chan_out = [&](int c_) { return reinterpret_cast<char *>(a.out) + ((long)c_ * a.out_stride + (gw - a.ovl)) * OB; };
  const char *srcb = chan_src(c);
  char *outb = chan_out(c);
  // The DMA is issued from inline asm, not through __builtin_amdgcn_global_load_lds: the compiler's wait-count pass
  // puts an `s_waitcnt vmcnt(0)` in front of every LDS access that MAY alias an LDS-DMA destination it knows of, and
  // it cannot tell the two window buffers apart — the prefetch would be waited for right after it was issued. What the
  // compiler does not see it does not wait for; the waits for these loads are the explicit counted ones below (an
  // untracked load only ever makes a compiler-generated vmcnt wait longer, never shorter: VMEM retires in order).
  // LDS address = M0 + instruction offset + 16 * lane; SALU write of M0 -> LDS-DMA needs one wait state.
  auto dma_issue = [&](const char *src, char *buf) {
#ifndef K1_ABL_NOFETCH
    const unsigned ldsb = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)buf);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    if (NDMA == 1) {
      if (l < LASTL) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(ldsb) : "memory", "m0");
    } else if (NDMA == 2) {
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(ldsb) : "memory", "m0");
      if (l < LASTL) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:1024" :: "v"(src), "s"(ldsb) : "memory", "m0");
    } else if (NDMA == 3) {
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024" :: "v"(src), "s"(ldsb) : "memory", "m0");
      if (l < LASTL) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:2048" :: "v"(src), "s"(ldsb) : "memory", "m0");
    } else {
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\tglobal_load_lds_dwordx4 %0, off offset:2048"
                   :: "v"(src), "s"(ldsb) : "memory", "m0");
      if (l < LASTL) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:3072" :: "v"(src), "s"(ldsb) : "memory", "m0");
    }
#pragma clang diagnostic pop
#endif
  };
  if (u < a.U) {
    dma_issue(srcb + (long)tile * tile_in_bytes + lane_byte, wbase);
    if (PAIR && tile + 1 < tend) dma_issue(srcb + (long)(tile + 1) * tile_in_bytes + lane_byte, wbase + BUFB);
  }
  __syncthreads();   // tap fragments and table in place (the only workgroup barrier)

  // plane byte offsets of this lane's pieces. cs16: piece p = 4 samples = half of chunk j = p >> 1 (a chunk = 8 samples =
  // 16 bytes of a plane); cu8: piece p = chunk p. Chunks are de-interleaved by parity (even chunks in the first half of
  // the plane, odd in the second): a lane's K steps walk consecutive chunks and the 16 lanes a ds_read_b128 services
  // together cover one contiguous 256-byte bank row.
  int dofs[NDMA];
#pragma unroll
  for (int k = 0; k < NDMA; k++) {
    const int p = l + 64 * k;
    if (REAL) dofs[k] = 8 * p;   // linear planes: consecutive lanes read consecutive 16-byte chunks as it is
    else if (CU8) dofs[k] = (p & 1) * HALF + (p >> 1) * 16;
    else { const int j = p >> 1; dofs[k] = (j & 1) * HALF + (j >> 1) * 16 + (p & 1) * 8; }
  }
  const int coff = REAL ? 16 * (n + h) : h * HALF + 16 * n;   // chunk 2(n + s) + h of the wave's window (real: chunk n + 2s + h)
  // phase counters of the lane's samples 0 and 1 relative to the wave's first sample, as a 16-bit pair (only bits 8..14
  // of a counter pick the table entry); the wave's part is scalar and joins per slice in one v_pk_add_u16 per sample pair
  const uint32_t lane_cnt = (uint32_t)(MF_BLK * n + 8 * h) * a.inc;   // (every form: the lane's 8 samples are 16n + 8h + {0 ... 7} of the slice)
  const uint32_t lane_pair = (lane_cnt & 0xffffu) | ((lane_cnt + a.inc) << 16);

  // + 128*sum(a) rides into the low-plane accumulator as the first MFMA's C operand: a 16-register block that must
  // stay in vector registers (left to itself the compiler keeps it in 16 scalar registers, spills them, and pays 16
  // v_readlane + 16 v_mov per slice)
  v16i cinit;
#pragma unroll
  for (int r = 0; r < 16; r++) cinit[r] = (r & 1) ? a.cim : a.cre;
  asm volatile("" : "+v"(cinit));

#ifdef K1_STAMPS   // diagnostic build: shader-clock stamps at the phase boundaries, summed per phase per wave
  unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime();
  const unsigned long long st_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned st_tiles = 0;
#define K1_STAMP(i_) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i_] += t_ - st_t; st_t = t_; } while (0)
#else
#define K1_STAMP(i_) do { } while (0)
#endif
#ifndef K1_PRIO_ROT
#define K1_PRIO_ROT 1
#endif
#ifndef K1_PRIO_SHIFT
#define K1_PRIO_SHIFT 11
#endif
  unsigned prio_it = 0;
  const unsigned prio_slot = __builtin_amdgcn_s_getreg((3 << 11) | 4) & 3u;   // HW_REG_HW_ID[3:0]: the wave's slot in its SIMD

  // ---- the four stages of a slice -----------------------------------------------------------------------------------
  // P: raw window -> byte planes, in place: every read of the wave is issued before its first write, and the LDS
  // executes one wave's instructions in order
  auto stageP = [&](char *cb) __attribute__((always_inline)) {
    uint4 x[NDMA];
#pragma unroll
    for (int k = 0; k < NDMA; k++)
      if (k < NDMA - 1 || l < LASTL) x[k] = *reinterpret_cast<const uint4 *>(cb + 16 * (l + 64 * k));
    asm volatile("" ::: "memory");
#ifndef K1_ABL_NOCONV
#pragma unroll
    for (int k = 0; k < NDMA; k++) {
      if (k < NDMA - 1 || l < LASTL) {
        if (CS8) {   // (the bytes are the plane: only their place changes)
          *reinterpret_cast<uint4 *>(cb + dofs[k]) = make_uint4(x[k].x, x[k].y, x[k].z, x[k].w);   // (component-wise: the whole-struct copy went through scratch)
        } else if (CU8) {   // AutoCast: high byte (b + 129) mod 256, low byte 0 (no low plane at all)
          *reinterpret_cast<uint4 *>(cb + dofs[k]) = make_uint4(add129_rot(x[k].x), add129_rot(x[k].y), add129_rot(x[k].z), add129_rot(x[k].w));
        } else {
          uint2 l2, h2;
          l2.x = __builtin_amdgcn_perm(x[k].y, x[k].x, 0x06040200u) ^ 0x80808080u;
          l2.y = __builtin_amdgcn_perm(x[k].w, x[k].z, 0x06040200u) ^ 0x80808080u;
          h2.x = __builtin_amdgcn_perm(x[k].y, x[k].x, 0x07050301u);
          h2.y = __builtin_amdgcn_perm(x[k].w, x[k].z, 0x07050301u);
          *reinterpret_cast<uint2 *>(cb + dofs[k]) = l2;
          *reinterpret_cast<uint2 *>(cb + PLB + dofs[k]) = h2;
        }
      }
    }
#endif
  };
  // K: the K loop out of the planes in `cb`: operands of step s+1 in flight while the MFMAs of step s issue
#ifndef K1_AREG
#define K1_AREG 0
#endif
  // (tuning variant K1_AREG=1: the tap fragments live in 4(S + NH) registers — build with -DK1_MINWAVES=3, run with
  // SDRHIP_IQBB_WGPCU=3)
  v4i AlR[K1_AREG ? S : 1], AhR[K1_AREG ? NH : 1];
  if (K1_AREG) {
#pragma unroll
    for (int s = 0; s < S; s++) AlR[s] = taps_s[s * 64 + l];
#pragma unroll
    for (int s = 0; s < NH; s++) AhR[s] = taps_s[(S + s) * 64 + l];
  }
  // (K1_ARES = n: the HIGH-plane tap fragments of the first n high steps stay in registers for the whole kernel — the headline
  // kernel uses 114 of the 128 registers four waves per SIMD leave it; a fragment read is 1 KB per wave and slice, and LDS
  // reads are a fifth of the energy of a launch that runs at the power limit. Not for the AM epilogue: 126 registers.)
#ifndef K1_ARES
#define K1_ARES 0
#endif
  constexpr int NRES = (K1_AREG || CU8 || REAL || EPI == SDRHIP_EPI_AM || NW != 4) ? 0 : (K1_ARES < NH ? K1_ARES : NH);
  v4i AhRes[NRES > 0 ? NRES : 1];
#pragma unroll
  for (int s = 0; s < NRES; s++) AhRes[s] = taps_s[(S + s) * 64 + l];
  struct KOps { v4i uh, ul, Al, Ah; };
  auto stageK_begin = [&](const char *cb, KOps &o) __attribute__((always_inline)) {
    const char *pl = cb + coff, *ph = cb + (CU8 ? 0 : PLB) + coff;
    o.uh = *reinterpret_cast<const v4i *>(ph); o.ul = o.uh;
    if (!CU8) o.ul = *reinterpret_cast<const v4i *>(pl);
    if (K1_AREG) { o.Al = AlR[0]; o.Ah = AhR[0]; }
    else {
      o.Al = taps_s[l]; o.Ah = o.Al;
      if (S0 == 0) o.Ah = NRES > 0 ? AhRes[0] : taps_s[S * 64 + l];
    }
  };
  auto stageK = [&](const char *cb, KOps &o, v16i &acc_hh, v16i &acc_mid, v16i &acc_ll) __attribute__((always_inline)) {
    constexpr int SA = 0, SB = S;
    const char *pl = cb + coff, *ph = cb + (CU8 ? 0 : PLB) + coff;
#ifdef K1_ABL_NOKLOOP
    if (SA == 0) acc_mid[0] = o.uh.x ^ o.ul.x ^ o.Al.x ^ o.Ah.x;
#else
#pragma unroll
    for (int s = SA; s < SB; s++) {
#ifdef K1_BSHIFT
      // (the two-lane reads below are invisible to the compiler's wait counts: this step's operands — issued a whole step
      // of MFMAs ago — are waited for here, BEFORE the next step's reads are issued)
      if (!REAL && s > SA) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o.uh), "+v"(o.ul) :: "memory");
#endif
      KOps nx = o;
      if (s + 1 < S) {
#ifdef K1_BSHIFT
        // (-DK1_BSHIFT) The sample-plane operand of step s + 1 IS the operand of step s one lane to the left inside each
        // half of the wave (lane (n, h) reads piece n + s of half h): the next operand comes by four DPP moves per plane
        // (wave_shl:1) and only the lanes (31, h) read their fresh piece from LDS — 32 bytes per plane and step instead of
        // 1 KB. An LDS read costs energy by the byte (tools/probes/mfma_energy.hip: one wave-wide ds_read_b128 is half an
        // MFMA's worth), and the kernel runs at the power limit.
        if (!REAL) {
#if K1_BSHIFT == 2
          // (round 5's second cut: the moves have no tied old operand — bound_ctrl zero-fills the lanes without a source, which
          // the two-lane reads overwrite anyway — and the reads run under an exec mask set INSIDE the asm statement: no
          // branch, no memory clobber that would pin the tap-fragment loads)
          auto shl1 = [](v4i x) __attribute__((always_inline)) {
            v4i r;
#pragma unroll
            for (int q = 0; q < 4; q++) r[q] = __builtin_amdgcn_mov_dpp(x[q], 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
            return r;
          };
          nx.uh = shl1(o.uh);
          if (!CU8) nx.ul = shl1(o.ul);
          const unsigned ah_ = (unsigned)(uintptr_t)(ph + KSB * (s + 1)), al_ = (unsigned)(uintptr_t)(pl + KSB * (s + 1));
          const unsigned long long m2 = 0x8000000080000000ull;   // lanes 31 and 63
          if (!CU8)
            asm volatile("s_mov_b64 exec, %4\n\tds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_mov_b64 exec, -1"
                         : "+v"(nx.uh), "+v"(nx.ul) : "v"(ah_), "v"(al_), "s"(m2));
          else
            asm volatile("s_mov_b64 exec, %2\n\tds_read_b128 %0, %1\n\ts_mov_b64 exec, -1" : "+v"(nx.uh) : "v"(ah_), "s"(m2));
#else
          auto shl1 = [](v4i x) __attribute__((always_inline)) {
            v4i r;
#pragma unroll
            for (int q = 0; q < 4; q++) r[q] = __builtin_amdgcn_update_dpp(x[q], x[q], 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
            return r;
          };
          nx.uh = shl1(o.uh);
          if (!CU8) nx.ul = shl1(o.ul);
          if (n == 31) {   // (lanes 31 and 63; asm: the compiler must not turn the branch into a wave-wide read and a select)
            const unsigned ah_ = (unsigned)(uintptr_t)(ph + KSB * (s + 1));
            asm volatile("ds_read_b128 %0, %1" : "+v"(nx.uh) : "v"(ah_) : "memory");
            if (!CU8) {
              const unsigned al_ = (unsigned)(uintptr_t)(pl + KSB * (s + 1));
              asm volatile("ds_read_b128 %0, %1" : "+v"(nx.ul) : "v"(al_) : "memory");
            }
          }
#endif
        } else
#endif
        {
#if defined(K1_ABL_BSAME_NOREAD)   // (tuning ablation, results wrong: step 0's sample planes serve every step, no reads)
#elif defined(K1_ABL_BSAME_READ)   // (... the same operand values with the reads KEPT: their difference prices the reads alone)
        { const v4i t0 = *reinterpret_cast<const v4i *>(ph + KSB * (s + 1)), t1 = *reinterpret_cast<const v4i *>(pl + KSB * (s + 1));
          asm volatile("" :: "v"(t0), "v"(t1)); }
#else
        nx.uh = *reinterpret_cast<const v4i *>(ph + KSB * (s + 1));
        if (!CU8) nx.ul = *reinterpret_cast<const v4i *>(pl + KSB * (s + 1));
#endif
        }
#ifndef K1_ABL_AREG   // (tuning ablation, results wrong: the tap fragments of step 0 serve every step — what resident fragments would save)
        if (K1_AREG) {
          nx.Al = AlR[s + 1];
          if (s + 1 >= S0 && s + 1 < S0 + NH) nx.Ah = AhR[s + 1 - S0];
        } else {
          nx.Al = taps_s[(s + 1) * 64 + l];
          if (s + 1 >= S0 && s + 1 < S0 + NH) nx.Ah = (s + 1 - S0 < NRES) ? AhRes[s + 1 - S0 < NRES ? s + 1 - S0 : 0] : taps_s[(S + s + 1 - S0) * 64 + l];
        }
#endif
      }
#ifndef K1_MFMA_ORDER
#define K1_MFMA_ORDER 1
#endif
#if K1_MFMA_ORDER == 1
      // A step's four MFMAs in an order in which ONE operand stays between neighbours — (Al,uh) (Ah,uh) (Ah,ul) (Al,ul): two
      // changes of the tap operand and one of the sample plane inside a step, where (Al,uh) (Al,ul) (Ah,uh) (Ah,ul) changed
      // the sample plane every time. The kernel runs at the socket's power limit and an int8 MFMA's energy follows what its
      // multipliers' inputs switch: -1.4 % per launch, FM and USB alike (tools/abk1.py, 9 interleaved rounds, bands that
      // do not overlap: profiles/r17_ab_mfma_order.txt; -DK1_MFMA_ORDER=0: the old order). The asm statements pin the order.
      acc_mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Al, o.uh, acc_mid, 0, 0, 0);
      asm volatile("" : "+v"(acc_mid));
      if (s >= S0 && s < S0 + NH) {
        acc_hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Ah, o.uh, acc_hh, 0, 0, 0);
        asm volatile("" : "+v"(acc_hh));
        if (!CU8) { acc_mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Ah, o.ul, acc_mid, 0, 0, 0); asm volatile("" : "+v"(acc_mid)); }
      }
      if (!CU8) { acc_ll = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Al, o.ul, acc_ll, 0, 0, 0); asm volatile("" : "+v"(acc_ll)); }
#elif K1_MFMA_ORDER == 2
      // (tuning variant: alternate the step's direction — even steps (Al,uh) (Ah,uh) (Ah,ul) (Al,ul), odd steps the reverse —
      // so that the tap operand Al also carries over the step boundary's neighbours as little changed as the algebra allows)
      if ((s & 1) == 0) {
        acc_mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Al, o.uh, acc_mid, 0, 0, 0); asm volatile("" : "+v"(acc_mid));
        if (s >= S0 && s < S0 + NH) {
          acc_hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Ah, o.uh, acc_hh, 0, 0, 0); asm volatile("" : "+v"(acc_hh));
          if (!CU8) { acc_mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Ah, o.ul, acc_mid, 0, 0, 0); asm volatile("" : "+v"(acc_mid)); }
        }
        if (!CU8) { acc_ll = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Al, o.ul, acc_ll, 0, 0, 0); asm volatile("" : "+v"(acc_ll)); }
      } else {
        if (!CU8) { acc_ll = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Al, o.ul, acc_ll, 0, 0, 0); asm volatile("" : "+v"(acc_ll)); }
        if (s >= S0 && s < S0 + NH) {
          if (!CU8) { acc_mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Ah, o.ul, acc_mid, 0, 0, 0); asm volatile("" : "+v"(acc_mid)); }
          acc_hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Ah, o.uh, acc_hh, 0, 0, 0); asm volatile("" : "+v"(acc_hh));
        }
        acc_mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Al, o.uh, acc_mid, 0, 0, 0); asm volatile("" : "+v"(acc_mid));
      }
#else
      acc_mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Al, o.uh, acc_mid, 0, 0, 0);
      if (!CU8) acc_ll = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Al, o.ul, acc_ll, 0, 0, 0);
      if (s >= S0 && s < S0 + NH) {
        acc_hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Ah, o.uh, acc_hh, 0, 0, 0);
        if (!CU8) acc_mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Ah, o.ul, acc_mid, 0, 0, 0);
      }
#endif
      o = nx;
    }
#endif
  };
  // (PAIR) the K loop over TWO windows: the tap fragments of a step are read once and feed the MFMAs of both slices
  struct KOps2 { v4i uhA, ulA, uhB, ulB, Al, Ah; };
  auto stageK2 = [&](const char *cbA, const char *cbB, v16i &hhA, v16i &midA, v16i &llA, v16i &hhB, v16i &midB, v16i &llB) __attribute__((always_inline)) {
    const char *plA = cbA + coff, *phA = cbA + PLB + coff, *plB = cbB + coff, *phB = cbB + PLB + coff;
    KOps2 o;
    o.uhA = *reinterpret_cast<const v4i *>(phA); o.ulA = *reinterpret_cast<const v4i *>(plA);
    o.uhB = *reinterpret_cast<const v4i *>(phB); o.ulB = *reinterpret_cast<const v4i *>(plB);
    o.Al = taps_s[l]; o.Ah = o.Al;
    if (S0 == 0) o.Ah = taps_s[S * 64 + l];
#pragma unroll
    for (int s = 0; s < S; s++) {
      KOps2 nx = o;
      if (s + 1 < S) {
        nx.uhA = *reinterpret_cast<const v4i *>(phA + KSB * (s + 1)); nx.ulA = *reinterpret_cast<const v4i *>(plA + KSB * (s + 1));
        nx.uhB = *reinterpret_cast<const v4i *>(phB + KSB * (s + 1)); nx.ulB = *reinterpret_cast<const v4i *>(plB + KSB * (s + 1));
        nx.Al = taps_s[(s + 1) * 64 + l];
        if (s + 1 >= S0 && s + 1 < S0 + NH) nx.Ah = taps_s[(S + s + 1 - S0) * 64 + l];
      }
      midA = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Al, o.uhA, midA, 0, 0, 0);
      midB = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Al, o.uhB, midB, 0, 0, 0);
      llA = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Al, o.ulA, llA, 0, 0, 0);
      llB = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Al, o.ulB, llB, 0, 0, 0);
      if (s >= S0 && s < S0 + NH) {
        hhA = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Ah, o.uhA, hhA, 0, 0, 0);
        hhB = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Ah, o.uhB, hhB, 0, 0, 0);
        midA = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Ah, o.ulA, midA, 0, 0, 0);
        midB = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.Ah, o.ulB, midB, 0, 0, 0);
      }
      o = nx;
    }
  };
  // E: lane (n, h) holds the 8 consecutive samples 16n + 8h + {0 ... 7} of the wave's slice — block Lb = 2n + h — in EVERY form
  // (the tap fragments' rows are permuted for it on the host: complex and real input alike): recombine the byte-plane
  // accumulators, >>14, rotate by LUT[idx(n)]. D = 8: the block IS the lane's group — window sum of the products' high halves.
  // wave_cnt: LUT phase counter of the wave's first sample (scalar)
  // (edge_: the cold slices of the any-D form — samples outside the call contribute nothing; erel0: call-relative index of the lane's first sample)
  // (DG) the group sums without an LDS round trip, out of PREFIX sums over the wave's 512 samples: D >= 9 puts at most ONE
  // group boundary inside a block, tm_sp samples into it (8: none) — a constant of the lane, like the lanes a team leader
  // fetches its group's two prefix sums from (tm_a0 / tm_a1: ds_bpermute addresses). With a frequency shift the rotated
  // samples are 16-bit: packed in pairs, the block's total and the part in front of its boundary are four v_dot2_i32_i16
  // each — against (1, 1), and against the lane's 0 / 1 masks tm_m (sample 2jj in the low half, 2jj + 1 in the high half).
  // Where the slice's last group ends with the slice (GS * D = 512) block 63 has no boundary inside and publishes the
  // wave's total instead: the last team's leader fetches it like any other prefix.
  int tm_sp = 8, tm_a0 = 0, tm_a1 = 0;
  unsigned tm_m[4] = {0u, 0u, 0u, 0u};
  bool tm_end = false;
  if (DG) {
    const int Lb = 2 * n + h;
    const int g = (8 * Lb + DD - 1) / DD, pos = g * DD;
    tm_sp = (g <= GS && pos < 8 * Lb + 8) ? pos - 8 * Lb : 8;
    const int kk = min(l >> a.lpg_sh, GS - 1);
    const int b0 = (kk * DD) >> 3, b1 = ((kk + 1) * DD) >> 3, b1c = min(b1, 63);
    tm_a0 = 4 * ((b0 >> 1) + 32 * (b0 & 1));
    tm_a1 = 4 * ((b1c >> 1) + 32 * (b1c & 1));
    tm_end = b1 >= 64;
    const int spm = tm_sp < 8 ? tm_sp : (Lb == 63 && GS * DD == 512) ? 8 : 0;   // samples of the block that count for the published prefix
#pragma unroll
    for (int jj = 0; jj < 4; jj++) tm_m[jj] = (2 * jj < spm ? 1u : 0u) | (2 * jj + 1 < spm ? 0x10000u : 0u);
  }
  (void)tm_sp; (void)tm_a0; (void)tm_a1; (void)tm_end; (void)tm_m;
  auto stageE = [&](auto edge_, const v16i &acc_hh, const v16i &acc_mid, const v16i &acc_ll, uint32_t wave_cnt, char *escr, int erel0,
                    int2 *pdst = nullptr, int ob0 = 0) __attribute__((always_inline)) {   // (PART: where the slice's partial sums go, its first group boundary)
    constexpr bool EDGE = decltype(edge_)::value;
    int L[8][3];
#ifdef K1_ABL_NOEPI
    if (false) {
#else
    if (ROT) {
#endif
      typedef int v2i __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int jj = 0; jj < 4; jj++) {
        const uint32_t wc = (wave_cnt + 2u * jj * a.inc) & 0xffffu, wpair = wc | (wc << 16);
        uint32_t pr, o0, o1;
        asm("v_pk_add_u16 %0, %1, %2" : "=v"(pr) : "v"(lane_pair), "s"(wpair));
        if (WIDE) {
          asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(o0) : "s"(4), "v"(pr));
          asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(o1) : "s"(4), "v"(pr));
          const v4i e0 = *reinterpret_cast<__attribute__((address_space(3))) const v4i *>((uintptr_t)o0);
          const v4i e1 = *reinterpret_cast<__attribute__((address_space(3))) const v4i *>((uintptr_t)o1);
          L[2 * jj][0] = e0.x; L[2 * jj][1] = e0.y; L[2 * jj][2] = e0.z;
          L[2 * jj + 1][0] = e1.x; L[2 * jj + 1][1] = e1.y; L[2 * jj + 1][2] = e1.z;
        } else {
          o0 = (pr >> 5) & (127u << 3); o1 = (pr >> 21) & (127u << 3);
          const v2i e0 = *reinterpret_cast<__attribute__((address_space(3))) const v2i *>((uintptr_t)o0);
          const v2i e1 = *reinterpret_cast<__attribute__((address_space(3))) const v2i *>((uintptr_t)o1);
          L[2 * jj][0] = e0.x; L[2 * jj][1] = e0.y; L[2 * jj + 1][0] = e1.x; L[2 * jj + 1][1] = e1.y;
        }
      }
    }
    int2 sum = make_int2(0, 0);
    int vx[8], vy[8];   // (DG) the lane's samples after the rotation (32-bit products: the sample is the high half), or (no shift) the FIR values
#ifdef K1_ABL_NOEPI
#pragma unroll
    for (int r = 0; r < 16; r += 2) { sum.x += acc_hh[r] ^ acc_mid[r] ^ acc_ll[r]; sum.y += acc_hh[r + 1] ^ acc_mid[r + 1] ^ acc_ll[r + 1]; }
#pragma unroll
    for (int j = 0; j < 8; j++) { vx[j] = sum.x; vy[j] = sum.y; }
#else
#pragma unroll
    for (int j = 0; j < 8; j++) {
      unsigned tre = ((unsigned)acc_hh[2 * j] << 8) + (unsigned)acc_mid[2 * j];   // (compiler code: it pads the MFMA -> VALU hazard)
      unsigned tim = ((unsigned)acc_hh[2 * j + 1] << 8) + (unsigned)acc_mid[2 * j + 1];
      int rr, ri;
      if (CS8) {   // S = t exactly; the int16 the reference's complex<int32> -> complex<int16> conversion leaves of S >> 14
        asm("v_bfe_i32 %0, %1, 14, 16" : "=v"(rr) : "v"(tre));
        asm("v_bfe_i32 %0, %1, 14, 16" : "=v"(ri) : "v"(tim));
      } else if (CU8) {   // S = t << 8 exactly (mod 2^32): S >> 14 = bits 6 .. 23 of t, sign-extended — one bit-field extract
        // (as an instruction: the builtin came out as a shift pair — 16 vector instructions more per slice in kernels
        // that are bound by vector issue)
        asm("v_bfe_i32 %0, %1, 6, 18" : "=v"(rr) : "v"(tre));
        asm("v_bfe_i32 %0, %1, 6, 18" : "=v"(ri) : "v"(tim));
      } else {
        asm("" : "+v"(tre)); asm("" : "+v"(tim));   // no re-association into 2 shifts + add3
        rr = (int)((tre << 8) + (unsigned)acc_ll[2 * j]) >> FSH; ri = (int)((tim << 8) + (unsigned)acc_ll[2 * j + 1]) >> FSH;
      }
      if (EDGE) {   // (register pair j is sample j of the lane's run)
        const int rel = erel0 + j;
        if (rel < 0 || rel >= a.N) { rr = 0; ri = 0; }
      }
      if (ROT) {
        int x = WIDE ? mad24a(L[j][0], rr, mul24a(L[j][2], ri)) : sub32(mul24a(L[j][0], rr), mul24a(L[j][1], ri));
        int y = mad24a(L[j][0], ri, mul24a(L[j][1], rr));
        if (CS8) {   // the product wraps to int16 and is shifted by Traits<int8_t>::shift = 8: bits 8 .. 15, sign-extended
          asm("v_bfe_i32 %0, %0, 8, 8" : "+v"(x));
          asm("v_bfe_i32 %0, %0, 8, 8" : "+v"(y));
        }
        if (DG) { vx[j] = x; vy[j] = y; }
        else if (CS8) { sum.x += x; sum.y += y; }
        else { sum.x = add_hi16(x, sum.x); sum.y = add_hi16(y, sum.y); }
      } else if (DG) {
        vx[j] = rr; vy[j] = ri;
      } else {
        sum.x = (int)((unsigned)sum.x + (unsigned)rr); sum.y = (int)((unsigned)sum.y + (unsigned)ri);
      }
    }
#endif
    if (SD) {
      // the wave's 512 rotated samples in stream order: lane (n, h) holds 16n + 8h + {0 ... 7}, the wave writes 64 contiguous
      // 32-byte pieces ({x >> 16, y >> 16} as two int16 per sample; without a shift the 18-bit parts in two arrays)
      unsigned *gsc = reinterpret_cast<unsigned *>(BUFB >= 2048 ? escr : gscb);
      unsigned *gsy = reinterpret_cast<unsigned *>(BUFB >= 2048 ? gscb : gscb2);   // (no shift) the imaginary parts' array
      unsigned q8[8];
#pragma unroll
      for (int j = 0; j < 8; j++) q8[j] = ROT ? __builtin_amdgcn_perm((unsigned)vy[j], (unsigned)vx[j], 0x07060302u) : (unsigned)vx[j];
      *reinterpret_cast<uint4 *>(gsc + 8 * (2 * n + h)) = make_uint4(q8[0], q8[1], q8[2], q8[3]);
      *reinterpret_cast<uint4 *>(gsc + 8 * (2 * n + h) + 4) = make_uint4(q8[4], q8[5], q8[6], q8[7]);
      if (!ROT) {
        *reinterpret_cast<uint4 *>(gsy + 8 * (2 * n + h)) = make_uint4((unsigned)vy[0], (unsigned)vy[1], (unsigned)vy[2], (unsigned)vy[3]);
        *reinterpret_cast<uint4 *>(gsy + 8 * (2 * n + h) + 4) = make_uint4((unsigned)vy[4], (unsigned)vy[5], (unsigned)vy[6], (unsigned)vy[7]);
      }
      asm volatile("" ::: "memory");   // (the group sums below read them: same wave, in order)
      return sum;
    }
#ifdef K1_ABL_NOTEAM   // (timing only: no team sums)
    if (DG) { for (int j = 0; j < 8; j++) { sum.x ^= vx[j]; sum.y ^= vy[j]; } return sum; }
#endif
    if (DG) {
      // Group sums as differences of PREFIX sums over the wave's 512 samples, all in registers. Per block: its total tX / tY
      // and the part in front of its group boundary aX / aY (0 where it has none). The totals of a lane pair's 16 samples are
      // scanned over n (the same scan in both halves of the wave: four row shifts and one row broadcast); a lane publishes the
      // prefix sum AT its boundary, and the team leader of group k fetches the two that bound its group (ds_bpermute: the LDS
      // crossbar, no LDS memory). History: through LDS with strided reads per team 28 - 41 % of the kernel's time (round 3);
      // running sums + a select tree over them 70 vector instructions per slice (round 4); this form 30.
      int tX, tY, aX, aY;
      int px[4], py[4];   // (with a shift) the samples packed in pairs: {sample 2jj, sample 2jj + 1} as two int16
      if (ROT) {
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
          px[jj] = (int)__builtin_amdgcn_perm((unsigned)vx[2 * jj + 1], (unsigned)vx[2 * jj], CS8 ? 0x05040100u : 0x07060302u);   // (int8 chain: the values themselves)
          py[jj] = (int)__builtin_amdgcn_perm((unsigned)vy[2 * jj + 1], (unsigned)vy[2 * jj], CS8 ? 0x05040100u : 0x07060302u);
        }
        auto dsum = [&](const int *p4, unsigned m0, unsigned m1, unsigned m2, unsigned m3) __attribute__((always_inline)) {
          int acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, p4[0]), __builtin_bit_cast(s16x2, m0), 0, false);
          acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, p4[1]), __builtin_bit_cast(s16x2, m1), acc, false);
          acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, p4[2]), __builtin_bit_cast(s16x2, m2), acc, false);
          return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, p4[3]), __builtin_bit_cast(s16x2, m3), acc, false);
        };
        unsigned ones = 0x00010001u;
        asm("" : "+s"(ones));   // (one scalar register: VOP3P takes no literal)
        tX = dsum(px, ones, ones, ones, ones); tY = dsum(py, ones, ones, ones, ones);
        if (!PART) { aX = dsum(px, tm_m[0], tm_m[1], tm_m[2], tm_m[3]); aY = dsum(py, tm_m[0], tm_m[1], tm_m[2], tm_m[3]); }
        else { aX = 0; aY = 0; }
      } else {
        // (no shift: 18-bit values) running sums inside the block, the one in front of the boundary by a three-level select
        // tree on the bits of the lane-constant index
        int rx[8], ry[8];
        rx[0] = vx[0]; ry[0] = vy[0];
#pragma unroll
        for (int j = 1; j < 8; j++) { rx[j] = (int)((unsigned)rx[j - 1] + (unsigned)vx[j]); ry[j] = (int)((unsigned)ry[j - 1] + (unsigned)vy[j]); }
        const int ti = tm_sp - 1;
        const bool tb0 = (ti & 1) != 0, tb1 = (ti & 2) != 0, tb2 = (ti & 4) != 0, tok = ti >= 0 && ti <= 6;
        auto pick7 = [&](const int *v) __attribute__((always_inline)) {
          const int t0 = tb0 ? v[1] : v[0], t1 = tb0 ? v[3] : v[2], t2 = tb0 ? v[5] : v[4], t3 = v[6];
          const int u0 = tb1 ? t1 : t0, u1 = tb1 ? t3 : t2;
          return tok ? (tb2 ? u1 : u0) : 0;
        };
        tX = rx[7]; tY = ry[7];
        aX = pick7(rx); aY = pick7(ry);
        if (PART) {   // (the boundaries are the wave's, not the lane's: the running sums stay for prefix_at)
#pragma unroll
          for (int j = 0; j < 8; j++) { vx[j] = rx[j]; vy[j] = ry[j]; }
        }
      }
      const auto tx = __builtin_amdgcn_permlane32_swap((unsigned)tX, (unsigned)tX, false, false);   // {block (n, 0)'s total, block (n, 1)'s} in both halves
      const auto ty = __builtin_amdgcn_permlane32_swap((unsigned)tY, (unsigned)tY, false, false);
      const int cx = (int)(tx[0] + tx[1]), cy = (int)(ty[0] + ty[1]);
      int ix = cx, iy = cy;   // inclusive scan over n: inside the 16-lane rows (row_shr: s, zero beyond the row), then the row before (row_bcast: 15 into rows 1 and 3)
      ix += __builtin_amdgcn_update_dpp(0, ix, 0x111, 0xf, 0xf, true); iy += __builtin_amdgcn_update_dpp(0, iy, 0x111, 0xf, 0xf, true);
      ix += __builtin_amdgcn_update_dpp(0, ix, 0x112, 0xf, 0xf, true); iy += __builtin_amdgcn_update_dpp(0, iy, 0x112, 0xf, 0xf, true);
      ix += __builtin_amdgcn_update_dpp(0, ix, 0x114, 0xf, 0xf, true); iy += __builtin_amdgcn_update_dpp(0, iy, 0x114, 0xf, 0xf, true);
      ix += __builtin_amdgcn_update_dpp(0, ix, 0x118, 0xf, 0xf, true); iy += __builtin_amdgcn_update_dpp(0, iy, 0x118, 0xf, 0xf, true);
      ix += __builtin_amdgcn_update_dpp(0, ix, 0x142 /* row_bcast:15 */, 0xa, 0xf, false); iy += __builtin_amdgcn_update_dpp(0, iy, 0x142, 0xa, 0xf, false);
      // the prefix sum in front of the block
      const int bpx = ix - (h ? (int)tx[1] : cx), bpy = iy - (h ? (int)ty[1] : cy);
      if (PART) {
        // the sums in front of the (wave-uniform) boundaries ob0 and ob0 + Dreal, where they fall inside the slice: the
        // prefix in front of the boundary's block + the block's samples in front of it, read from the lane that owns it
        const int totx = __builtin_amdgcn_readlane(ix, 31), toty = __builtin_amdgcn_readlane(iy, 31);
        auto prefix_at = [&](int p, int &ox, int &oy) __attribute__((always_inline)) {   // 0 < p < 512 (scalar)
          const int j = p & 7, lb = p >> 3, ln = (lb >> 1) + 32 * (lb & 1);
          int wx = bpx, wy = bpy;
          if (ROT) {   // (scalar masks: samples 2jj, 2jj + 1 in front of position j)
            unsigned m[4];
#pragma unroll
            for (int jj = 0; jj < 4; jj++) m[jj] = (2 * jj < j ? 1u : 0u) | (2 * jj + 1 < j ? 0x10000u : 0u);
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
              wx = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, px[jj]), __builtin_bit_cast(s16x2, m[jj]), wx, false);
              wy = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, py[jj]), __builtin_bit_cast(s16x2, m[jj]), wy, false);
            }
          } else {
#pragma unroll
            for (int t = 1; t < 8; t++) if (j == t) { wx += vx[t - 1]; wy += vy[t - 1]; }   // (scalar conditions; vx / vy: the running sums)
          }
          ox = __builtin_amdgcn_readlane(wx, ln); oy = __builtin_amdgcn_readlane(wy, ln);
        };
        int p0x = totx, p0y = toty, p1x = totx, p1y = toty;
        if (ob0 < 512) prefix_at(ob0, p0x, p0y);
        if (ob0 + a.Dreal < 512) prefix_at(ob0 + a.Dreal, p1x, p1y); else { p1x = totx; p1y = toty; }
        if (ob0 >= 512) { p1x = totx; p1y = toty; }
        if (l == 0) {
          pdst[0] = make_int2(p0x, p0y);
          pdst[1] = make_int2(p1x - p0x, p1y - p0y);
          pdst[2] = make_int2(totx - p1x, toty - p1y);
        }
        return sum;
      }
      const int sbx = bpx + aX, sby = bpy + aY;
      const int s0x = __builtin_amdgcn_ds_bpermute(tm_a0, sbx), s0y = __builtin_amdgcn_ds_bpermute(tm_a0, sby);
      int s1x = __builtin_amdgcn_ds_bpermute(tm_a1, sbx), s1y = __builtin_amdgcn_ds_bpermute(tm_a1, sby);
      if (!ROT) {   // (the select tree has no entry for a whole block: the wave's total by a broadcast)
        const int totx = __builtin_amdgcn_readlane(ix, 31), toty = __builtin_amdgcn_readlane(iy, 31);
        if (tm_end) { s1x = totx; s1y = toty; }
      }
      sum = make_int2(s1x - s0x, s1y - s0y);   // (whole in the team's first lane)
    }
    return sum;
  };
  // F (DG): the team's first lane owns group k = l >> lpg_sh of the slice: truncating division by D (|sum| <= D * 2^15; up to
  // D = 180 libstdc++'s (s * D) / (D * D) of src/baseband.hh:214 cannot wrap, it is trunc(s / D): a float estimate
  // biased down + one exact remainder step), demodulator, store; FM: the previous group's angle from the team before
  auto div_d = [&](int v) __attribute__((always_inline)) {
    // (s * D can wrap from D = 182 on — unrotated 18-bit values: from 128 on, and past 2^24 the float estimate is no longer
    // exact enough: the reference's wrapping arithmetic then, a generic division, once per group)
    if (DD > (ROT ? 180 : 90)) return (int)(short)box_div(v, DD);
    const unsigned m = (unsigned)max(v, -v);
    unsigned q = (unsigned)((float)m * a.inv_d);          // inv_d = (1 / D)(1 - 2^-20): q or q - 1 (m < 2^23: the product is good to 2^-7)
    const unsigned r = m - q * (unsigned)DD;
    q += (r >= (unsigned)DD) ? 1u : 0u;
    return (int)(short)(v < 0 ? -(int)q : (int)q);        // (the int16 wrap of the assignment)
  };
  // A slice yields only GS group sums (4 at decimation 125) in its team leaders, and finishing them — two divisions, the
  // demodulator (28 slots for FM alone) — would cost every slice a fifth of its vector instructions for a handful of
  // lanes. The leaders PARK their sums in a 64-entry per-wave LDS array instead, slice after slice of the unit (a unit's
  // tiles are consecutive), and one pass finishes up to 64 groups, one per lane, when the unit ends or the array is full.
  // (HS) FM in the any-D forms where the units are not whole channels: a slice's first output is the difference between the
  // last angle of the slice BEFORE it — another wave's, possibly on another XCD — and its own first angle. Both owners post
  // their angle in device memory (agent-scope stores: written through the XCD's L2), wait for the stores, then look for the
  // other's entry (agent-scope loads): whoever finds it — at least one of the two does, both may — writes the output
  // (again through the L2; the two values are the same). The entry carries the call's number, so nothing of an older call
  // is mistaken for this one's. A lane passes first / last for the group it holds; sid: its slice, phi: its angle.
  // Compiled in only under -DK1_FM_HANDSHAKE (tools/build_variant.sh hs "-DK1_FM_HANDSHAKE"; run with SDRHIP_IQBB_FM_HANDSHAKE=1):
  // it measured no faster than the fix-up launch at any channel count, and its mere presence — a kernel-uniform branch at
  // the four emission sites — cost the any-D FM kernels 2-4 % (profiles/r17_ab_fm_handshake.txt, r17_ab_nohs.txt).
#ifdef K1_FM_HANDSHAKE
#define HS_ON(A_) ((A_).hs != nullptr)
#else
#define HS_ON(A_) false
#endif
  auto hs_exchange = [&](const HotArgs &A, bool first, bool last, int cc, int sid, int phi) __attribute__((always_inline)) {
    long long *plast = A.hs + (long)cc * A.hs_stride + sid + 1;              // slice sid's last angle
    long long *pfirst = A.hs + ((long)A.C + cc) * A.hs_stride + sid + 1;     // slice sid's first angle
    const long long mine = ((long long)A.hs_seq << 32) | (long long)(unsigned)phi;
    if (last) __hip_atomic_store(plast, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (first) __hip_atomic_store(pfirst, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the stores are acknowledged from beyond the L2 before the loads go out)
    short *row = reinterpret_cast<short *>(A.out) + (long)cc * A.out_stride;
    if (first) {
      const long long v = __hip_atomic_load(plast - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((int)(v >> 32) == A.hs_seq) __hip_atomic_store(row + (long)sid * GS, (short)((int)(unsigned)v - phi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (last) {
      const long long v = __hip_atomic_load(pfirst + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((int)(v >> 32) == A.hs_seq) __hip_atomic_store(row + (long)(sid + 1) * GS, (short)(phi - (int)(unsigned)v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  int2 *pend = reinterpret_cast<int2 *>(pendb);
  int npend = 0, ptile0 = 0;          // (scalar) groups parked, tile of the first parked slice
  // the lane's place in a flush: parked entry l is group k_f of the j_f-th parked slice
  const int j_f = l / GS, k_f = l - j_f * GS;
  auto park = [&](int2 sum) __attribute__((always_inline)) {
    const int lsh = a.lpg_sh, k = l >> lsh;
    if ((l & ((1 << lsh) - 1)) == 0 && k < GS) pend[npend + k] = sum;
    npend += GS;
    asm volatile("" ::: "memory");
  };
  auto flush = [&](int c_) __attribute__((always_inline)) {
    const int2 sum = pend[min(l, 63)];
    asm volatile("" ::: "memory");
    const bool live = l < npend;
    int yr = div_d(sum.x), yi = div_d(sum.y);
    if (CS8) { yr = (signed char)yr; yi = (signed char)yi; }   // (the int8 node's output type)
    // group k_f of tile ptile0 + j_f: output index (ptile0 + j_f) * OG + gw + k_f of the channel's row (outb points at gw)
    char *orow = outb + ((long)(ptile0 + j_f) * a.OG + k_f) * OB;
    if (EPI == SDRHIP_EPI_NONE && CS8) {
      if (live) *reinterpret_cast<uint16_t *>(orow) = (uint16_t)((yr & 0xff) | ((yi & 0xff) << 8));
    } else if (EPI == SDRHIP_EPI_NONE) {
      if (live) *reinterpret_cast<uint32_t *>(orow) = ((uint32_t)(uint16_t)yr) | ((uint32_t)(uint16_t)yi << 16);
    } else if (EPI == SDRHIP_EPI_AM) {
      const short o = am_i16(yr, yi);
      if (live) *reinterpret_cast<short *>(orow) = o;
    } else if (EPI == SDRHIP_EPI_USB) {
      const short o = usb_i16(yr, yi);
      if (live) *reinterpret_cast<short *>(orow) = o;
    } else {
      // every group of a slice is emitted; its first one as -phi: a tiny launch behind this one (iqbb_fm_fixup_kernel) adds
      // the previous slice's last angle, which that slice's last group leaves in philast
      const int phi = fm_phi(yr, yi);
      const int prev = __builtin_amdgcn_update_dpp(0, phi, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);   // entry l - 1: the group before, within a slice
      if (HS_ON(a)) {   // (kernel-uniform) few channels: the slices' first outputs by the neighbours' handshake
        if (live && k_f > 0) *reinterpret_cast<short *>(orow) = (short)(prev - phi);
        hs_exchange(a, live && k_f == 0, live && k_f == GS - 1, c_, 4 * (ptile0 + j_f) + wv, phi);
      } else {
        if (live) *reinterpret_cast<short *>(orow) = (short)((k_f > 0 ? prev : 0) - phi);
        if (live && k_f == GS - 1) a.philast[(long)c_ * a.philast_stride + 4 * (ptile0 + j_f) + wv] = (short)phi;
      }
    }
    npend = 0;
  };
  // (SD) group k = l + 64 i of the slice: the sum of its D consecutive samples out of the LDS arrays stageE left
  auto sd_group_sum = [&](const char *escr, int k) __attribute__((always_inline)) {
    const unsigned *gsc = reinterpret_cast<const unsigned *>(BUFB >= 2048 ? escr : gscb);
    const unsigned *gsy = reinterpret_cast<const unsigned *>(BUFB >= 2048 ? gscb : gscb2);
    const int kk = min(k, GS - 1);   // (lanes beyond the slice's last group read its samples and drop the result)
    int sx = 0, sy = 0;
    auto add1 = [&](unsigned v, unsigned vy) __attribute__((always_inline)) {
      if (ROT) {
        asm("v_add_u32_sdwa %0, sext(%1), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "+v"(sx) : "v"(v));
        asm("v_add_u32_sdwa %0, sext(%1), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(sy) : "v"(v));
      } else {
        sx = (int)((unsigned)sx + v); sy = (int)((unsigned)sy + vy);
      }
    };
    // (wave-uniform branches) an even D reads its group in 16- or 8-byte pieces — consecutive lanes, contiguous or
    // 24-byte-strided addresses: conflict-free, where dword reads at a stride of 2, 4 or 6 dwords are 2- to 4-way bank
    // conflicts (r13b's counters: half of the ÷4 kernel's LDS cycles); an odd stride is conflict-free as it is
    if (DD == 4) {
      const uint4 v = *reinterpret_cast<const uint4 *>(gsc + 4 * kk);
      uint4 vy = make_uint4(0, 0, 0, 0);
      if (!ROT) vy = *reinterpret_cast<const uint4 *>(gsy + 4 * kk);
      add1(v.x, vy.x); add1(v.y, vy.y); add1(v.z, vy.z); add1(v.w, vy.w);
    } else if ((DD & 1) == 0) {
      for (int t = 0; t < DD; t += 2) {   // (scalar trip count)
        const uint2 v = *reinterpret_cast<const uint2 *>(gsc + kk * DD + t);
        uint2 vy = make_uint2(0, 0);
        if (!ROT) vy = *reinterpret_cast<const uint2 *>(gsy + kk * DD + t);
        add1(v.x, vy.x); add1(v.y, vy.y);
      }
    } else {
      for (int t = 0; t < DD; t++) add1(gsc[kk * DD + t], ROT ? 0u : gsy[kk * DD + t]);   // (scalar trip count)
    }
    return make_int2(sx, sy);
  };
  // (SD) a HOT slice finished: every group whole, emitted, none of them the call's first or last. FM: the slice's first
  // group goes out as -phi (iqbb_fm_fixup_kernel adds the previous slice's last angle), its last one leaves phi in philast.
  auto sd_finish_hot = [&](const char *escr, char *orow, int c_, int tile_) __attribute__((always_inline)) {
    int carry_phi = 0;   // (scalar) the angle of group 64 i - 1
    for (int i = 0; 64 * i < GS; i++) {   // (scalar trip count: 2 ... 4)
      const int k = l + 64 * i;
      const bool live = k < GS;
      const int2 sum = sd_group_sum(escr, k);
      const int yr = div_d(sum.x), yi = div_d(sum.y);
      if (EPI == SDRHIP_EPI_NONE) {
        if (live) reinterpret_cast<uint32_t *>(orow)[k] = ((uint32_t)(uint16_t)yr) | ((uint32_t)(uint16_t)yi << 16);
      } else if (EPI == SDRHIP_EPI_AM) {
        const short o = am_i16(yr, yi);
        if (live) reinterpret_cast<short *>(orow)[k] = o;
      } else if (EPI == SDRHIP_EPI_USB) {
        const short o = usb_i16(yr, yi);
        if (live) reinterpret_cast<short *>(orow)[k] = o;
      } else {
        const int phi = fm_phi(yr, yi);
        int prev = __builtin_amdgcn_update_dpp(0, phi, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);   // lane l - 1: group k - 1
        if (l == 0) prev = carry_phi;   // (i = 0: the slice's first group, -phi for now)
        if (HS_ON(a)) {   // (kernel-uniform)
          if (live && k > 0) reinterpret_cast<short *>(orow)[k] = (short)(prev - phi);
          if (i == 0 || 64 * (i + 1) >= GS) hs_exchange(a, live && k == 0, live && k == GS - 1, c_, 4 * tile_ + wv, phi);   // (scalar: the chunks that hold the first / last group)
        } else {
          if (live) reinterpret_cast<short *>(orow)[k] = (short)(prev - phi);
          if (live && k == GS - 1) a.philast[(long)c_ * a.philast_stride + 4 * tile_ + wv] = (short)phi;
        }
        carry_phi = __builtin_amdgcn_readlane(phi, 63);
      }
    }
  };
  // (SD) a COLD slice finished: the border rules of cold_finish_gen, per group. sid: the slice's number 4 * tile + wv.
  auto sd_finish_cold = [&](const char *escr, int cc, int sid) __attribute__((always_inline)) {
    int carry_phi = 0;
    for (int i = 0; 64 * i < GS; i++) {
      const int k = l + 64 * i, q = sid * GS + k;   // q: the group's output index within the call
      const bool lead = k < GS && q < b.n_groups;
      int2 sum = sd_group_sum(escr, k);
      if (lead && q == 0) {
        const int2 carry = b.acc_old[cc];
        sum.x = (int)((unsigned)sum.x + (unsigned)carry.x);
        sum.y = (int)((unsigned)sum.y + (unsigned)carry.y);
        if (b.extra0) {   // absolute sample 0: one slow FIR evaluation per channel and stream start
          int er = 0, ei = 0;
          for (int ii = 0; ii < b.OP; ii++) {
            const uint32_t x = load_x(b, cc, -(b.OP - 1) + ii);
            const uint2 kk = b.taps[ii];
            er = dot2(x, kk.x, er); ei = dot2(x, kk.y, ei);
          }
          const int2 v = rotate(b, b.lut, make_int2(er >> 14, ei >> 14), b.n0_lo);
          sum.x = (int)((unsigned)sum.x + (unsigned)v.x);
          sum.y = (int)((unsigned)sum.y + (unsigned)v.y);
        }
      }
      const bool emits = lead && q < b.n_out;
      if (lead && q == b.n_groups - 1) b.acc_new[cc] = emits ? make_int2(0, 0) : sum;
      const int yr = div_d(sum.x), yi = div_d(sum.y);
      if (EPI == SDRHIP_EPI_NONE) {
        if (emits) reinterpret_cast<uint32_t *>(ac.out)[(long)cc * ac.out_stride + q] = ((uint32_t)(uint16_t)yr) | ((uint32_t)(uint16_t)yi << 16);
      } else if (EPI == SDRHIP_EPI_AM) {
        const short o = am_i16(yr, yi);
        if (emits) reinterpret_cast<short *>(ac.out)[(long)cc * ac.out_stride + q] = o;
      } else if (EPI == SDRHIP_EPI_USB) {
        const short o = usb_i16(yr, yi);
        if (emits) reinterpret_cast<short *>(ac.out)[(long)cc * ac.out_stride + q] = o;
      } else {
        const int phi = fm_phi(yr, yi);
        int prev = __builtin_amdgcn_update_dpp(0, phi, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        if (l == 0) prev = carry_phi;
        short o;
        if (q == 0) o = (short)yr;                                   // index 0 is never written by FMDemod (in place)
        else if (q == 1) o = (short)((int)b.fm_old[cc] - phi);       // y[0] is never looked at: the previous call's last angle
        else o = (short)(prev - phi);                                // (a slice's first group: prev = 0, the fix-up launch adds philast)
        if (HS_ON(ac)) {   // (kernel-uniform)
          const bool hfirst = emits && k == 0 && q >= 2;   // (outputs 0 and 1 have their own rules above)
          if (emits && !hfirst) reinterpret_cast<short *>(ac.out)[(long)cc * ac.out_stride + q] = o;
          if (i == 0 || 64 * (i + 1) >= GS) hs_exchange(ac, hfirst, emits && k == GS - 1, cc, sid, phi);
        } else {
          if (emits) reinterpret_cast<short *>(ac.out)[(long)cc * ac.out_stride + q] = o;
          if (emits && k == GS - 1) ac.philast[(long)cc * ac.philast_stride + sid] = (short)phi;
        }
        if (emits && q == b.n_out - 1 && b.n_out >= 2) b.fm_new[cc] = (short)phi;
        carry_phi = __builtin_amdgcn_readlane(phi, 63);
      }
    }
  };
  // F: truncating division by 8, demodulator, store. orow: (scalar) the wave's first group of this slice; lanes below
  // glw_lo store nothing (FM: group 0 only supplies the previous angle)
  // (mb_bl, scalar, FM only: a buffer boundary of a multi-buffer call lies at lane mb_bl of this slice, 1 <= mb_bl <= 62 — that
  // group is a buffer's index 0: FMDemod never writes it, in place it holds the real part; the group behind it takes the
  // PREVIOUS buffer's last angle, two lanes back, src/demod.hh:242-254. 0: none)
  auto stageF = [&](int2 sum, char *orow, int glw_lo, int mb_bl = 0) __attribute__((always_inline)) {
    const int glw = 2 * n + h;
    // libstdc++'s (s*8)/(8*8) (src/baseband.hh:214): |s| <= 9 * 2^17 (a window of 16-bit rotated values, or of 18-bit FIR
    // values when there is no shift), so nothing wraps and it is trunc(s / 8) + the int16 wrap of the assignment
    int yr = div8_i16(sum.x), yi = div8_i16(sum.y);
    if (CS8) { yr = (signed char)yr; yi = (signed char)yi; }   // (the int8 node's output type)
    if (EPI == SDRHIP_EPI_NONE && CS8) {
      if (glw >= glw_lo) reinterpret_cast<uint16_t *>(orow)[glw] = (uint16_t)((yr & 0xff) | ((yi & 0xff) << 8));
    } else if (EPI == SDRHIP_EPI_NONE) {
      if (glw >= glw_lo) reinterpret_cast<uint32_t *>(orow)[glw] = ((uint32_t)(uint16_t)yr) | ((uint32_t)(uint16_t)yi << 16);
    } else if (EPI == SDRHIP_EPI_AM) {
      const short o = am_i16(yr, yi);
      if (glw >= glw_lo) reinterpret_cast<short *>(orow)[glw] = o;
    } else if (EPI == SDRHIP_EPI_USB) {
      const short o = usb_i16(yr, yi);
      if (glw >= glw_lo) reinterpret_cast<short *>(orow)[glw] = o;
    } else {
      const int phi = fm_phi(yr, yi);
      const int prev = prev_group_value(phi, h);
      int o = prev - phi;
      if (!DG && !CS8 && mb_bl != 0) {   // (wave-uniform branch: one slice in 130 of a multi-buffer call)
        const int prev2 = prev_group_value(prev, h);
        if (glw == mb_bl) o = yr;
        else if (glw == mb_bl + 1) o = prev2 - phi;
      }
      if (glw >= glw_lo) reinterpret_cast<short *>(orow)[glw] = (short)o;
    }
  };
  constexpr int GLW0 = EPI == SDRHIP_EPI_FM ? 1 : 0;
  // the lane of this slice's buffer boundary (multi-buffer calls; scalar arithmetic: first stored group qs = tile * OG - ovl + gw + 1,
  // boundaries at mb_q1 + j * mb_p): the smallest k >= 0 with (qs + k - mb_q1) % mb_p == 0, taken if the boundary AND the group
  // behind it are stored lanes of this slice (k <= 61)
  auto boundary_lane = [&](int tile_) __attribute__((always_inline)) {
    if (!(EPI == SDRHIP_EPI_FM && !DG && !CS8) || a.mb_p == 0) return 0;
    const int qs = tile_ * a.OG - a.ovl + gw + 1, lo = qs - a.mb_q1;
    int k;
    if (lo <= 0) k = -lo;
    else {
      const unsigned qq = (unsigned)(((unsigned long long)(unsigned)lo * a.mb_magic) >> 32);
      unsigned r = (unsigned)lo - qq * (unsigned)a.mb_p;
      if (r >= (unsigned)a.mb_p) r -= (unsigned)a.mb_p;
      k = r ? a.mb_p - (int)r : 0;
    }
    return (k <= 61 && qs + k <= a.mb_qlast) ? k + 1 : 0;
  };
  // Fairness: the SIMD arbitrates its waves by priority, then AGE, and in a persistent grid the ages never change —
  // the oldest wave of a SIMD ran at full speed and was done after 64 us, the youngest starved and finished alone
  // at 120 us (s_memrealtime stamps). The priority rotates over the SIMD's four wave slots, one step per slice.
  unsigned prio_time = 0;
  auto rotate_priority = [&]() __attribute__((always_inline)) {
    if (K1_PRIO_ROT == 1 || K1_PRIO_ROT == 4) {   // (s_setprio takes an immediate: a two-level branch tree, 5-6 scalar instructions executed)
      // (K1_PRIO_ROT=4, tuning variant: the rotation follows the shader clock — one step per 2048 cycles, the same for every wave
      // of a SIMD — instead of the wave's own slice count: waves that drift apart in slice count can meet at EQUAL priority, the
      // tie goes to the older one and the drift feeds itself. The counter is read one slice ahead: its latency is hidden.)
      unsigned pv = prio_it++ + prio_slot;
      if (K1_PRIO_ROT == 4) { pv = prio_time + prio_slot; prio_time = (unsigned)(__builtin_amdgcn_s_memtime() >> K1_PRIO_SHIFT); }
#ifdef K1_PRIO_BIAS   // (tuning variant: every K1_PRIO_BIAS-th slice the YOUNGER slots of a SIMD take the top priorities, whatever the rotation says)
      if ((prio_it % K1_PRIO_BIAS) == 0) pv = prio_slot;
#endif
      asm volatile("s_bitcmp1_b32 %0, 1\n\ts_cbranch_scc1 2f\n\ts_bitcmp1_b32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_setprio 0\n\ts_branch 4f\n"
                   "1:\n\ts_setprio 1\n\ts_branch 4f\n"
                   "2:\n\ts_bitcmp1_b32 %0, 0\n\ts_cbranch_scc1 3f\n\ts_setprio 2\n\ts_branch 4f\n"
                   "3:\n\ts_setprio 3\n"
                   "4:" :: "s"(pv) : "scc");
    }
  };
  auto wait_dma = [&](bool newer_in_flight) __attribute__((always_inline)) {   // all VMEM older than the NDMA youngest instructions (or everything)
    if (__builtin_expect(newer_in_flight, 1)) {
      if (NDMA == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (NDMA == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if (NDMA == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };

  {
    // ---- one slice after the other; 4 waves per SIMD overlap one wave's matrix phase with the others' vector phases ----
    // (window buffer PAR is compile-time: the buffer's offset folds into the LDS instructions' immediates)
    auto slice = [&](auto par_) __attribute__((always_inline)) {
      constexpr int PAR = decltype(par_)::value;
      char *cb = wbase + PAR * BUFB, *nb = wbase + (1 - PAR) * BUFB;
      // the step after this one: next tile of the unit, or the first hot tile of the workgroup's next unit
      int nu = u, nc = c, ng = g, ntile = tile + 1, ntend = tend;
      const char *nsrcb = srcb;
      char *noutb = outb;
      if (ntile >= tend) {
        next_unit(nu, nc, ng, ntile, ntend);
        nsrcb = chan_src(nc); noutb = chan_out(nc);
      }
      const bool more = nu < a.U;
      rotate_priority();
      // the next slice's window starts its journey into the other buffer (its planes were last read in the previous
      // slice's K loop); then wait for this slice's own window: everything older than those NDMA instructions
      K1_STAMP(0);
      if (__builtin_expect(more, 1)) dma_issue(nsrcb + (long)ntile * tile_in_bytes + lane_byte, nb);
      wait_dma(more);
      K1_STAMP(1);
      stageP(cb);
      // (the plane reads below see these writes: same wave, in order. The empty asm statements keep the COMPILER from
      // moving LDS accesses across: no barrier or fence instruction separates them.)
      K1_STAMP(2);
      asm volatile("" ::: "memory");
      v16i acc_hh = {0}, acc_mid = {0}, acc_ll = {0};
      if (!CU8) acc_ll = cinit;
      KOps ops;
      if (K1_PRIO_ROT == 2) asm volatile("s_setprio 3");   // (tuning variants: the matrix phase first / last)
      if (K1_PRIO_ROT == 3) asm volatile("s_setprio 0");
      stageK_begin(cb, ops);
      stageK(cb, ops, acc_hh, acc_mid, acc_ll);
      if (K1_PRIO_ROT == 2) asm volatile("s_setprio 0");
      if (K1_PRIO_ROT == 3) asm volatile("s_setprio 3");
#ifdef K1_STAMPS
      asm volatile("s_nop 0" : "+v"(acc_hh), "+v"(acc_mid), "+v"(acc_ll));   // the accumulators are complete before the stamp
#endif
      K1_STAMP(3);
      int2 *pdst = nullptr;
      int ob0 = 0;
      if (PART) {   // (scalar) the slice's number, its first sample, its first group boundary
        const int sid = 4 * tile + wv, t0 = a.base0_rel + sid * 512 - a.base_real;
        ob0 = t0 < 0 ? a.Dreal - t0 : a.Dreal - (int)((unsigned)t0 % (unsigned)a.Dreal);
        pdst = a.part + (long)c * a.part_stride + 3 * sid;
      }
      int2 sum = stageE(std::false_type{}, acc_hh, acc_mid, acc_ll, cnt0 + (uint32_t)tile * tile_cnt, cb, 0, pdst, ob0);
#ifdef K1_STAMPS
      asm volatile("" : "+v"(sum.x), "+v"(sum.y));
#endif
      K1_STAMP(4);
      if (SD) {
        sd_finish_hot(cb, outb + (long)tile * tile_out_bytes, c, tile);
      } else if (PART) {
        // (the slice's partial sums are out: stageE)
      } else if (DG) {
        if (npend == 0) ptile0 = tile;
        park(sum);
        if (tile + 1 >= tend || npend + GS > 64) flush(c);   // the unit ends here (its tiles were consecutive), or the array is full
      } else {
        stageF(sum, outb + (long)tile * tile_out_bytes, GLW0, boundary_lane(tile));
      }
      K1_STAMP(5);
#ifdef K1_STAMPS
      st_tiles++;
#endif
      u = nu; c = nc; g = ng; tile = ntile; tend = ntend; srcb = nsrcb; outb = noutb;
    };
    // (PAIR) one step = the unit's next TWO tiles (the last step of a unit with an odd tile count: one; its second window
    // then holds stale samples and its results are dropped). Buffers 2 PAR, 2 PAR + 1 hold this step's windows, the other
    // two receive the next step's.
    auto slice2 = [&](auto par_) __attribute__((always_inline)) {
      constexpr int PAR = decltype(par_)::value;
      char *cbA = wbase + (2 * PAR) * BUFB, *cbB = cbA + BUFB, *nbA = wbase + (2 * (1 - PAR)) * BUFB, *nbB = nbA + BUFB;
      const bool hasB = tile + 1 < tend;
      int nu = u, nc = c, ng = g, ntile = tile + (hasB ? 2 : 1), ntend = tend;
      const char *nsrcb = srcb;
      char *noutb = outb;
      if (ntile >= tend) {
        next_unit(nu, nc, ng, ntile, ntend);
        nsrcb = chan_src(nc); noutb = chan_out(nc);
      }
      const bool more = nu < a.U, nhasB = more && ntile + 1 < ntend;
      rotate_priority();
      if (__builtin_expect(more, 1)) {
        dma_issue(nsrcb + (long)ntile * tile_in_bytes + lane_byte, nbA);
        if (nhasB) dma_issue(nsrcb + (long)(ntile + 1) * tile_in_bytes + lane_byte, nbB);
      }
      // everything older than the DMA instructions just issued (VMEM retires in order)
      static_assert(!PAIR || NDMA == 3, "pair variant: three DMA instructions per window");
      if (nhasB) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (more) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stageP(cbA);
      if (hasB) stageP(cbB);
      asm volatile("" ::: "memory");
      v16i hhA = {0}, midA = {0}, llA = cinit, hhB = {0}, midB = {0}, llB = cinit;
      stageK2(cbA, cbB, hhA, midA, llA, hhB, midB, llB);
      {
        const int2 sum = stageE(std::false_type{}, hhA, midA, llA, cnt0 + (uint32_t)tile * tile_cnt, cbA, 0);
        stageF(sum, outb + (long)tile * tile_out_bytes, GLW0);
      }
      if (hasB) {
        const int2 sum = stageE(std::false_type{}, hhB, midB, llB, cnt0 + (uint32_t)(tile + 1) * tile_cnt, cbB, 0);
        stageF(sum, outb + (long)(tile + 1) * tile_out_bytes, GLW0);
      }
      u = nu; c = nc; g = ng; tile = ntile; tend = ntend; srcb = nsrcb; outb = noutb;
    };
    if (PAIR) {
      while (u < a.U) {
        slice2(std::integral_constant<int, 0>{});
        if (!(u < a.U)) break;
        slice2(std::integral_constant<int, 1>{});
      }
    } else {
      while (u < a.U) {
        slice(std::integral_constant<int, 0>{});
        if (!(u < a.U)) break;
        slice(std::integral_constant<int, 1>{});
      }
    }
  }

  // (DG) a cold slice's groups, finished by the team leaders themselves: the carry into the call's first group and the
  // stream's D+1 first window, the open last group and the demodulator's angle for the next call, FMDemod's first two
  // outputs of a buffer (group_finish is the decimation-8 form of the same rules). sid: the slice's number 4 * tile + wv.
  auto cold_finish_gen = [&](int2 sum, int cc, int sid) __attribute__((always_inline)) {
    const int lsh = ac.lpg_sh, k = l >> lsh, q = sid * GS + k;   // q: the group's output index within the call
    const bool lead = (l & ((1 << lsh) - 1)) == 0 && k < GS && q < b.n_groups;
    if (lead && q == 0) {
      const int2 carry = b.acc_old[cc];
      sum.x = (int)((unsigned)sum.x + (unsigned)carry.x);
      sum.y = (int)((unsigned)sum.y + (unsigned)carry.y);
      if (b.extra0) {   // absolute sample 0: one slow FIR evaluation per channel and stream start
        int er = 0, ei = 0;
        for (int i = 0; i < b.OP; i++) {
          const uint32_t x = load_x(b, cc, -(b.OP - 1) + i);
          const uint2 kk = b.taps[i];
          er = dot2(x, kk.x, er); ei = dot2(x, kk.y, ei);
        }
        const int2 v = rotate(b, b.lut, make_int2(er >> 14, ei >> 14), b.n0_lo);
        sum.x = (int)((unsigned)sum.x + (unsigned)v.x);
        sum.y = (int)((unsigned)sum.y + (unsigned)v.y);
      }
    }
    const bool emits = lead && q < b.n_out;
    if (lead && q == b.n_groups - 1) b.acc_new[cc] = emits ? make_int2(0, 0) : sum;
    int yr = div_d(sum.x), yi = div_d(sum.y);
    if (CS8) { yr = (signed char)yr; yi = (signed char)yi; }   // (the int8 node's output type)
    if (EPI == SDRHIP_EPI_NONE && CS8) {
      if (emits) reinterpret_cast<uint16_t *>(ac.out)[(long)cc * ac.out_stride + q] = (uint16_t)((yr & 0xff) | ((yi & 0xff) << 8));
    } else if (EPI == SDRHIP_EPI_NONE) {
      if (emits) reinterpret_cast<uint32_t *>(ac.out)[(long)cc * ac.out_stride + q] = ((uint32_t)(uint16_t)yr) | ((uint32_t)(uint16_t)yi << 16);
    } else if (EPI == SDRHIP_EPI_AM) {
      const short o = am_i16(yr, yi);
      if (emits) reinterpret_cast<short *>(ac.out)[(long)cc * ac.out_stride + q] = o;
    } else if (EPI == SDRHIP_EPI_USB) {
      const short o = usb_i16(yr, yi);
      if (emits) reinterpret_cast<short *>(ac.out)[(long)cc * ac.out_stride + q] = o;
    } else {
      const int phi = fm_phi(yr, yi);
      const int prev = __builtin_amdgcn_ds_bpermute(4 * (((k - 1) << lsh) & 63), phi);   // the leader of team k - 1
      short o;
      if (q == 0) o = CS8 ? (short)((yr & 0xff) | ((yi & 0xff) << 8))   // FMDemod<int8_t,int16_t> in place: out[0] = the 2 bytes of in[0]
                          : (short)yr;                             // index 0 is never written by FMDemod (in place)
      else if (q == 1) o = (short)((int)b.fm_old[cc] - phi);       // y[0] is never looked at: the previous call's last angle
      else o = (short)((k > 0 ? prev : 0) - phi);                  // (a slice's first group: the fix-up launch adds philast)
      if (HS_ON(ac)) {   // (kernel-uniform)
        const bool hfirst = emits && k == 0 && q >= 2;   // (outputs 0 and 1 have their own rules above)
        if (emits && !hfirst) reinterpret_cast<short *>(ac.out)[(long)cc * ac.out_stride + q] = o;
        hs_exchange(ac, hfirst, emits && k == GS - 1, cc, sid, phi);
      } else {
        if (emits) reinterpret_cast<short *>(ac.out)[(long)cc * ac.out_stride + q] = o;
        if (emits && k == GS - 1) ac.philast[(long)cc * ac.philast_stride + sid] = (short)phi;
      }
      if (emits && q == b.n_out - 1 && b.n_out >= 2) b.fm_new[cc] = (short)phi;
    }
  };

#if defined(__HIP_DEVICE_COMPILE__)
  if (LATE_B || LATE_A) {
    // the blocks' places in the kernarg segment: the explicit arguments of every kernel that calls this body
    // (`(const HotArgs a, const IqbbArgs b)`), laid out in declaration order at their natural alignment
    typedef const uint32_t __attribute__((address_space(4))) *KernargP;   // (the kernarg segment is constant memory: scalar loads)
    constexpr size_t B_OFF = (sizeof(HotArgs) + alignof(IqbbArgs) - 1) / alignof(IqbbArgs) * alignof(IqbbArgs);
    const char __attribute__((address_space(4))) *kp = (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp) :: "memory");
    static_assert(sizeof(IqbbArgs) % 4 == 0 && sizeof(HotArgs) % 4 == 0, "copied by dwords");
    uint32_t *dst = reinterpret_cast<uint32_t *>(&b_late);
#pragma unroll
    for (int i = 0; i < (LATE_B ? (int)(sizeof(IqbbArgs) / 4) : 0); i++) dst[i] = ((KernargP)(kp + B_OFF))[i];
    uint32_t *dsta = reinterpret_cast<uint32_t *>(&a_late);
#pragma unroll
    for (int i = 0; i < (LATE_A ? (int)(sizeof(HotArgs) / 4) : 0); i++) dsta[i] = ((KernargP)kp)[i];
  }
#endif
  // ---- the call's COLD slices --------------------------------------------------------------------------------
  // The slices slice_is_hot() rejects — history in the window, the call's first group, incomplete or unemitted
  // groups, the end of the input — are a few per channel (2 of 130 on the headline workload).
  {
    char *cb = wbase;
#ifdef K1_ABL_NOCOLD   // tuning ablation (results wrong): no cold phase
    for (int cc = ac.C; cc < ac.C; cc += gx) {
#else
    for (int cc = bx; cc < ac.C; cc += gx) {
#endif
      // (DG: every tile of the call in turn; the wave's own hot tiles [hl, hh) and the slices behind the call's last group are skipped)
      for (int t = 0; t < (DG ? ac.tiles_h : b.tiles); t = DG ? t + 1 : (t == 0 ? max(b.bt_hi, 1) : t + 1)) {
        const int q0 = t * ac.OG - ac.ovl, groups_here = DG ? 0 : min(b.CG, b.n_groups - q0);
        if (DG) { if ((t >= hl && t < hh) || (q0 + gw) >= b.n_groups) continue; }
        else if (slice_is_hot(HALO, WIN, ac.base0_rel, ac.OG, ac.ovl, ac.N, ac.n_out, t, wv) || gw + ac.ovl >= groups_here) continue;
        // the wave's window by ordinary loads: history / input / zeros per sample, every load issued from a clamped
        // address and masked afterwards (all in flight together)
        const int first = ac.base0_rel + (q0 + gw) * DD - HALO;
        const uint32_t *hrow = b.hist_old + (long)cc * b.HH;
        if (REAL) {   // 8 real samples per piece: input int16, or the low half of a history dword
          const uint16_t *row = reinterpret_cast<const uint16_t *>(ac.in) + (long)cc * ac.in_stride;
          uint32_t v[NDMA][8];
#pragma unroll
          for (int k = 0; k < NDMA; k++) {
            const int pp = min(l + 64 * k, NPIECE - 1);
#pragma unroll
            for (int j = 0; j < 8; j++) v[k][j] = row[max(min(first + 8 * pp + j, ac.N - 1), 0)];
          }
#pragma unroll
          for (int k = 0; k < NDMA; k++) {
            const int pp = min(l + 64 * k, NPIECE - 1);
#pragma unroll
            for (int j = 0; j < 8; j++) {
              const int rel = first + 8 * pp + j;
              if (rel >= ac.N) v[k][j] = 0u;
              if (first < 0) {   // (wave-uniform: only the call's first slices reach into the history)
                const uint32_t xh = hrow[max(b.HH + rel, 0)];
                if (rel < 0) v[k][j] = (b.HH + rel >= 0) ? (xh & 0xffffu) : 0u;
              }
            }
            if (k < NDMA - 1 || l < LASTL) {
              const uint32_t x0 = v[k][0] | (v[k][1] << 16), x1 = v[k][2] | (v[k][3] << 16), x2 = v[k][4] | (v[k][5] << 16), x3 = v[k][6] | (v[k][7] << 16);
              uint2 l2, h2;
              l2.x = __builtin_amdgcn_perm(x1, x0, 0x06040200u) ^ 0x80808080u;
              l2.y = __builtin_amdgcn_perm(x3, x2, 0x06040200u) ^ 0x80808080u;
              h2.x = __builtin_amdgcn_perm(x1, x0, 0x07050301u);
              h2.y = __builtin_amdgcn_perm(x3, x2, 0x07050301u);
              *reinterpret_cast<uint2 *>(cb + dofs[k]) = l2;
              *reinterpret_cast<uint2 *>(cb + PLB + dofs[k]) = h2;
            }
          }
        } else if (CU8) {
          const uint16_t *row = reinterpret_cast<const uint16_t *>(ac.in) + (long)cc * ac.in_stride;
          uint32_t v[NDMA][8];
#pragma unroll
          for (int k = 0; k < NDMA; k++) {
            const int pp = min(l + 64 * k, NPIECE - 1);
#pragma unroll
            for (int j = 0; j < 8; j++) v[k][j] = row[max(min(first + 8 * pp + j, ac.N - 1), 0)];
          }
#pragma unroll
          for (int k = 0; k < NDMA; k++) {
            const int pp = min(l + 64 * k, NPIECE - 1);
#pragma unroll
            for (int j = 0; j < 8; j++) {
              const int rel = first + 8 * pp + j;
              // the sample's two high-plane bytes: AutoCast of the input bytes, or bytes 1 and 3 of a history dword
              // (int8 chain: the input bytes themselves, or bytes 0 and 2 of a history dword — the sign-extended pair)
              uint32_t hb = CS8 ? v[k][j] : ((v[k][j] + 0x81u) & 0xffu) | ((v[k][j] + 0x8100u) & 0xff00u);
              if (rel >= ac.N) hb = 0u;
              if (first < 0) {   // (wave-uniform: only the call's first slices reach into the history)
                const uint32_t xh = hrow[max(b.HH + rel, 0)];
                if (rel < 0) hb = (b.HH + rel >= 0) ? (CS8 ? ((xh & 0xffu) | ((xh >> 8) & 0xff00u)) : (((xh >> 8) & 0xffu) | ((xh >> 16) & 0xff00u))) : 0u;
              }
              v[k][j] = hb;
            }
            auto rotb = [](uint32_t d) __attribute__((always_inline)) { return CS8 ? d : ror8(d); };   // (complex<uint8>: the hot loop's byte order inside a dword, add129_rot)
            if (k < NDMA - 1 || l < LASTL)
              *reinterpret_cast<uint4 *>(cb + dofs[k]) = make_uint4(rotb(v[k][0] | (v[k][1] << 16)), rotb(v[k][2] | (v[k][3] << 16)),
                                                                    rotb(v[k][4] | (v[k][5] << 16)), rotb(v[k][6] | (v[k][7] << 16)));
          }
        } else {
          const uint32_t *row = reinterpret_cast<const uint32_t *>(ac.in) + (long)cc * ac.in_stride;
          uint32_t v[NDMA][4];
#pragma unroll
          for (int k = 0; k < NDMA; k++) {
            const int pp = min(l + 64 * k, NPIECE - 1);
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const int rel = first + 4 * pp + j, hh = b.HH + rel;
              const uint32_t *src = rel >= 0 ? row + min(rel, ac.N - 1) : hrow + max(hh, 0);
              v[k][j] = *src;
            }
          }
#pragma unroll
          for (int k = 0; k < NDMA; k++) {
            const int pp = min(l + 64 * k, NPIECE - 1);
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const int rel = first + 4 * pp + j;
              if (rel >= ac.N || b.HH + rel < 0) v[k][j] = 0u;
            }
            if (k < NDMA - 1 || l < LASTL) {
              uint2 l2, h2;
              l2.x = __builtin_amdgcn_perm(v[k][1], v[k][0], 0x06040200u) ^ 0x80808080u;
              l2.y = __builtin_amdgcn_perm(v[k][3], v[k][2], 0x06040200u) ^ 0x80808080u;
              h2.x = __builtin_amdgcn_perm(v[k][1], v[k][0], 0x07050301u);
              h2.y = __builtin_amdgcn_perm(v[k][3], v[k][2], 0x07050301u);
              *reinterpret_cast<uint2 *>(cb + dofs[k]) = l2;
              *reinterpret_cast<uint2 *>(cb + PLB + dofs[k]) = h2;
            }
          }
        }
        asm volatile("" ::: "memory");   // (one wave's LDS operations execute in order: the reads below see these writes)
        v16i acc_hh = {0}, acc_mid = {0}, acc_ll = {0};
        if (!CU8) {
#pragma unroll
          for (int r = 0; r < 16; r++) acc_ll[r] = (r & 1) ? ac.cim : ac.cre;
        }
        const char *pl = cb + coff, *ph = cb + (CU8 ? 0 : PLB) + coff;
#pragma unroll
        for (int s_ = 0; s_ < S; s_++) {
          const v4i uh = *reinterpret_cast<const v4i *>(ph + KSB * s_);
          const v4i Al = taps_s[s_ * 64 + l];
          acc_mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al, uh, acc_mid, 0, 0, 0);
          if (!CU8) {
            const v4i ul = *reinterpret_cast<const v4i *>(pl + KSB * s_);
            acc_ll = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al, ul, acc_ll, 0, 0, 0);
            if (s_ >= S0 && s_ < S0 + NH) {
              const v4i Ah = taps_s[(S + s_ - S0) * 64 + l];
              acc_hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah, uh, acc_hh, 0, 0, 0);
              acc_mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah, ul, acc_mid, 0, 0, 0);
            }
          } else if (s_ >= S0 && s_ < S0 + NH) {
            acc_hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(taps_s[(S + s_ - S0) * 64 + l], uh, acc_hh, 0, 0, 0);
          }
        }
        if (DG) {
          const int s0 = ac.base0_rel + (q0 + gw) * DD;   // the slice's first sample, call-relative
          int2 *pdst = nullptr;
          int ob0 = 0;
          if (PART) {
            const int t0 = s0 - ac.base_real;
            ob0 = t0 < 0 ? ac.Dreal - t0 : ac.Dreal - (int)((unsigned)t0 % (unsigned)ac.Dreal);
            pdst = ac.part + (long)cc * ac.part_stride + 3 * (4 * t + wv);
          }
          const int2 sum = stageE(std::true_type{}, acc_hh, acc_mid, acc_ll, (ac.n0_lo + (uint32_t)s0) * ac.inc, cb, s0 + MF_BLK * n + 8 * h, pdst, ob0);
          if (SD) sd_finish_cold(cb, cc, 4 * t + wv);
          else if (!PART) cold_finish_gen(sum, cc, 4 * t + wv);
        } else {
          const int tb = ac.base0_rel + q0 * 8, rel0 = tb + 8 * gw + MF_BLK * n + 8 * h;
          const int2 sum = group_sum<ROT, CU8, true, WIDE ? 2 : 1, FSH, CS8>(b, acc_hh, acc_mid, acc_ll, rel0);
          group_finish(b, b.lut, cc, n, h, gw, q0, groups_here, sum);   // (its one table user, the stream's first sample, reads global memory)
        }
        asm volatile("" ::: "memory");
      }
      for (int k = tid & 255; k < b.HH; k += 256) {   // the FIR history for the next call (this virtual workgroup's channel)
        const long qq = (long)ac.N + k;   // index into concat(hist_old, in)
        b.hist_new[(long)cc * b.HH + k] = qq < b.HH ? b.hist_old[(long)cc * b.HH + qq] : raw_x(b, cc, qq - b.HH);
      }
    }
  }
  // FM in the any-D forms: a slice's first output went out as -phi (the slice before it, another wave's, holds the angle it
  // is a difference with, and leaves it in philast). When the units are whole channels (G = 1: the host chooses that where
  // the channels fill the grid evenly) every slice of the channels this workgroup walked — hot or cold — was finished by
  // one of ITS four waves: the missing angles are all in place behind one workgroup barrier, and the fix-up is the
  // workgroup's last step instead of a second launch (5 us: a launch's floor) behind this one.
  if (DG && EPI == SDRHIP_EPI_FM) {
    if (b.fix_hi > b.fix_lo) {   // (kernel-uniform)
      // (workgroup scope is all this needs — the four waves sit on one CU and share its L1, the barrier's release / acquire
      // pair orders their stores and loads; a device-scope fence here writes the XCD's whole L2 back, once per workgroup:
      // measured +110 us per launch)
      __syncthreads();
      for (int cc = bx; cc < ac.C; cc += gx) {
        short *row = reinterpret_cast<short *>(ac.out) + (long)cc * ac.out_stride;
        short *pl = ac.philast + (long)cc * ac.philast_stride;
        for (int sl = b.fix_lo + (tid & 255); sl < b.fix_hi; sl += 256) {
          short *o = row + (long)sl * GS;
          *o = (short)(*o + pl[sl - 1]);
        }
      }
    }
  }
  // (PART, whole channels as units) likewise the groups of the large-decimation form: every slice's partial sums of this
  // workgroup's channels are its own waves' — the finishing launch's work, one lane per group, behind the same barrier
  if (PART) {
    if (ac.fin_groups > 0) {   // (kernel-uniform)
      __syncthreads();
      BigdArgs f;
      f.part = ac.part; f.part_stride = ac.part_stride;
      f.D = ac.Dreal; f.base0_rel = ac.base_real; f.N = ac.N; f.n_groups = ac.fin_groups; f.n_out = ac.fin_out; f.epi = ac.fin_epi; f.C = ac.C;
      f.acc_old = b.acc_old; f.acc_new = b.acc_new; f.fm_old = b.fm_old; f.fm_new = b.fm_new;
      f.out = ac.out; f.out_stride = ac.out_stride;
      for (int cc = bx; cc < ac.C; cc += gx)
        for (int q = tid & 255; q < f.n_groups; q += 256) bigd_finish_group(f, cc, q);
    }
  }
#ifdef K1_STAMPS
  if (l == 0 && a.stamps) {   // 16 words per wave: 6 phase totals, -, -, slices, HW_ID, first and last realtime stamp
    unsigned long long *o = a.stamps + (size_t)((((unsigned)bx * 4 + wv) & 32767u) * 16);
    for (int i = 0; i < 6; i++) o[i] = st_acc[i];
    o[8] = st_tiles; o[9] = __builtin_amdgcn_s_getreg((15 << 11) | 4);
    o[10] = st_r0; o[11] = __builtin_amdgcn_s_memrealtime(); o[12] = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // XCC_ID[3:0]
  }
#endif
}

#ifndef K1_MINWAVES
#define K1_MINWAVES 4
#endif
// One launch per call: the hot grid, then each virtual workgroup's share of the cold slices.
template <int S, int S0, int NH, bool ROT, int EPI, int IN, int NW>
__global__ __launch_bounds__(64 * NW, K1_MINWAVES) void iqbb_hot_kernel(const HotArgs a, const IqbbArgs b) {
  iqbb_hot_body<S, S0, NH, ROT, EPI, IN, NW>(a, b);
}
template <int S, int S0, int NH, bool ROT, int EPI, int IN, int NW = 4>
__global__ __launch_bounds__(64 * NW, K1_MINWAVES) void iqbb_hot_anyd_kernel(const HotArgs a, const IqbbArgs b) {
  iqbb_hot_body<S, S0, NH, ROT, EPI, IN, NW, true>(a, b);
}

template <int S, int S0, int NH, bool ROT, int EPI, int IN, int NW = 4>
__global__ __launch_bounds__(64 * NW, K1_MINWAVES) void iqbb_hot_sd_kernel(const HotArgs a, const IqbbArgs b) {
  iqbb_hot_body<S, S0, NH, ROT, EPI, IN, NW, true, true>(a, b);
}

// HIP function attributes are PER DEVICE, and one process may drive several (sdrhip_comm_create: one context per
// device): a kernel's dynamic-LDS limit is raised once per device id (bit d of `mask`; the launch runs on the device the
// context made current). The setter is idempotent, so two threads racing on one device only set it twice.
template <class F>
inline void once_per_device(std::atomic<uint64_t> &mask, F &&set_attributes) {
  int d = 0;
  SDRHIP_CHECK_HIP(hipGetDevice(&d));
  const uint64_t bit = 1ull << (d & 63);
  if (mask.load(std::memory_order_acquire) & bit) return;
  set_attributes();
  mask.fetch_or(bit, std::memory_order_release);
}

// returns the waves per workgroup the plan runs in (hot_sd_nw; the host sizes the grid by it) — 0: its LDS fits none (nothing
// launched); dry_run: only answer
template <int S, int S0, int NH, int IN, int NW0 = 4>
int hot_launch_sd_one(bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b, bool dry_run) {
  const dim3 grid(hl.grid, 1);
#define SDRHIP_SD(R_, E_) hipLaunchKernelGGL((iqbb_hot_sd_kernel<S, S0, NH, R_, E_, IN, NWX>), grid, dim3(64 * NWX), lds, hl.stream, ha, b)
#define SDRHIP_SD_ATTR(R_, E_) SDRHIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&iqbb_hot_sd_kernel<S, S0, NH, R_, E_, IN, NWX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
#define SDRHIP_SD_E(R_) do { if (dry_run) return NWX; \
    const size_t lds = (size_t)hot_lds_bytes(S, NH, IN, NWX, false) + hot_sd_extra(S, IN, R_, NWX); \
    if (lds > 64 * 1024) { static std::atomic<uint64_t> attr_set{0}; once_per_device(attr_set, [&] { SDRHIP_SD_ATTR(R_, SDRHIP_EPI_FM); SDRHIP_SD_ATTR(R_, SDRHIP_EPI_AM); SDRHIP_SD_ATTR(R_, SDRHIP_EPI_USB); SDRHIP_SD_ATTR(R_, SDRHIP_EPI_NONE); }); } \
    switch (epi) { \
    case SDRHIP_EPI_FM: SDRHIP_SD(R_, SDRHIP_EPI_FM); break; \
    case SDRHIP_EPI_AM: SDRHIP_SD(R_, SDRHIP_EPI_AM); break; \
    case SDRHIP_EPI_USB: SDRHIP_SD(R_, SDRHIP_EPI_USB); break; \
    default: SDRHIP_SD(R_, SDRHIP_EPI_NONE); break; } return NWX; } while (0)
  // (real input: 4-wave workgroups only — a plan whose arrays need a larger one runs the VALU kernel)
  if (rot) { constexpr int NWX = hot_sd_nw(S, NH, IN, true, NW0); if constexpr (NWX > 0 && (IN != HOT_REAL || NWX == 4)) SDRHIP_SD_E(true); }
  else { constexpr int NWX = hot_sd_nw(S, NH, IN, false, NW0); if constexpr (NWX > 0 && (IN != HOT_REAL || NWX == 4)) SDRHIP_SD_E(false); }
#undef SDRHIP_SD_E
#undef SDRHIP_SD_ATTR
#undef SDRHIP_SD
  return 0;
}

template <int S, int S0, int NH, int IN, int NW>
void hot_launch_one(bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b) {
  constexpr bool PAIR = hot_pair(IN, NW, false);
  const size_t lds = (size_t)hot_lds_bytes(S, NH, IN, NW, hot_wide(S, NH, IN, NW, 0, PAIR), PAIR);
  if constexpr (IN != HOT_CS8) if (lds > 64 * 1024) {   // (the pair variant's four window buffers per wave; long filters)
    static std::atomic<uint64_t> attr_set{0};
    once_per_device(attr_set, [&] {
#define SDRHIP_HOT_ATTR(R_, E_) SDRHIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&iqbb_hot_kernel<S, S0, NH, R_, E_, IN, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
    SDRHIP_HOT_ATTR(true, SDRHIP_EPI_FM); SDRHIP_HOT_ATTR(true, SDRHIP_EPI_AM); SDRHIP_HOT_ATTR(true, SDRHIP_EPI_USB); SDRHIP_HOT_ATTR(true, SDRHIP_EPI_NONE);
    SDRHIP_HOT_ATTR(false, SDRHIP_EPI_FM); SDRHIP_HOT_ATTR(false, SDRHIP_EPI_AM); SDRHIP_HOT_ATTR(false, SDRHIP_EPI_USB); SDRHIP_HOT_ATTR(false, SDRHIP_EPI_NONE);
#undef SDRHIP_HOT_ATTR
    });
  }
  const dim3 grid(hl.grid, 1), block(64 * NW);
#define SDRHIP_HOT(R_, E_) hipLaunchKernelGGL((iqbb_hot_kernel<S, S0, NH, R_, E_, IN, NW>), grid, block, lds, hl.stream, ha, b)
#define SDRHIP_HOT_E(R_) do { switch (epi) { \
    case SDRHIP_EPI_FM: SDRHIP_HOT(R_, SDRHIP_EPI_FM); break; \
    case SDRHIP_EPI_AM: if constexpr (IN != HOT_CS8) SDRHIP_HOT(R_, SDRHIP_EPI_AM); break; \
    case SDRHIP_EPI_USB: if constexpr (IN != HOT_CS8) SDRHIP_HOT(R_, SDRHIP_EPI_USB); break; \
    default: SDRHIP_HOT(R_, SDRHIP_EPI_NONE); break; } } while (0)
  if (rot) SDRHIP_HOT_E(true); else SDRHIP_HOT_E(false);
#undef SDRHIP_HOT_E
#undef SDRHIP_HOT
}

template <int S, int S0, int NH, int IN, int NW = 4>
void hot_launch_anyd_one(bool rot, int epi, const HotLaunch &hl, const HotArgs &ha, const IqbbArgs &b) {
  const int extra = hot_anyd_extra(S, IN, rot, NW);
  const size_t lds = (size_t)hot_lds_bytes(S, NH, IN, NW, hot_wide(S, NH, IN, NW, extra)) + extra;
  static_assert(hot_lds_bytes(S, NH, IN, NW, false) + hot_anyd_extra(S, IN, false, NW) <= 163840, "any-D form: a workgroup's LDS");
  if constexpr (IN != HOT_CS8) if (NW > 4) {   // (beyond 64 KB of dynamic LDS: once per DEVICE and kernel)
    static std::atomic<uint64_t> attr_set{0};
    once_per_device(attr_set, [] {
#define SDRHIP_ANYD_ATTR(R_, E_) SDRHIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&iqbb_hot_anyd_kernel<S, S0, NH, R_, E_, IN, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, hot_lds_cap(NW, false, S) + hot_anyd_extra(S, IN, false, NW) > 163840 ? 163840 : hot_lds_cap(NW, false, S) + hot_anyd_extra(S, IN, false, NW)))
      SDRHIP_ANYD_ATTR(true, SDRHIP_EPI_NONE); SDRHIP_ANYD_ATTR(true, SDRHIP_EPI_FM); SDRHIP_ANYD_ATTR(true, SDRHIP_EPI_AM); SDRHIP_ANYD_ATTR(true, SDRHIP_EPI_USB);
      SDRHIP_ANYD_ATTR(false, SDRHIP_EPI_NONE); SDRHIP_ANYD_ATTR(false, SDRHIP_EPI_FM); SDRHIP_ANYD_ATTR(false, SDRHIP_EPI_AM); SDRHIP_ANYD_ATTR(false, SDRHIP_EPI_USB);
      SDRHIP_ANYD_ATTR(true, HOT_EPI_PARTIAL); SDRHIP_ANYD_ATTR(false, HOT_EPI_PARTIAL);
#undef SDRHIP_ANYD_ATTR
    });
  }
  const dim3 grid(hl.grid, 1), block(64 * NW);
#define SDRHIP_ANYD(R_, E_) hipLaunchKernelGGL((iqbb_hot_anyd_kernel<S, S0, NH, R_, E_, IN, NW>), grid, block, lds, hl.stream, ha, b)
#define SDRHIP_ANYD_E(R_) do { switch (epi) { \
    case SDRHIP_EPI_FM: SDRHIP_ANYD(R_, SDRHIP_EPI_FM); break; \
    case SDRHIP_EPI_AM: if constexpr (IN != HOT_CS8) SDRHIP_ANYD(R_, SDRHIP_EPI_AM); break; \
    case SDRHIP_EPI_USB: if constexpr (IN != HOT_CS8) SDRHIP_ANYD(R_, SDRHIP_EPI_USB); break; \
    case HOT_EPI_PARTIAL: if constexpr (IN != HOT_REAL && IN != HOT_CS8) SDRHIP_ANYD(R_, HOT_EPI_PARTIAL); break; \
    default: SDRHIP_ANYD(R_, SDRHIP_EPI_NONE); break; } } while (0)
  if (rot) SDRHIP_ANYD_E(true); else SDRHIP_ANYD_E(false);
#undef SDRHIP_ANYD_E
#undef SDRHIP_ANYD
}

}  // namespace
