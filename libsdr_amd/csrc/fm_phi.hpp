// fast_atan2<int16_t,int16_t>(a, b) / 2 — the angle FMDemod differences (reference src/math.hh:31-40, src/demod.hh:246),
// for a, b in the int16 range (the int8 chain calls it with int8 values: same formula, src/math.hh:12-29 differs only
// in the argument type).
//
//   b >= 0: angle = pi/4 - pi/4 * (b - |a|) / (b + |a|);   b < 0: angle = 3pi/4 - pi/4 * (b + |a|) / (|a| - b)
//   with pi/4 = 4096, integer (truncating) division; a < 0 negates; (0, 0) gives 0; the demodulator halves (trunc).
//
// Both branches divide by den = |a| + |b| and their numerators are +-(|b| - |a|) * 4096, so there is ONE unsigned division
// q = floor(4096 * ||b| - |a|| / den), q <= 4096, whose sign is sign(|b| - |a|) for b >= 0 and the opposite for b < 0.
// The division: a float estimate biased DOWN by 2^-20 (v_rcp_f32 is good to 1 ulp, the two multiplies to half an ulp each,
// the conversions of |d| <= 2^16 and den <= 2^17 are exact: the estimate lies in (x - 0.005, x], so its floor is q or q - 1)
// and one exact remainder test. 32 vector instructions; the generic form (signed numerator, two-sided correction, separate
// selects per branch) took 44.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ int fm_phi(int a, int b) {
  const unsigned aabs = (unsigned)max(a, -a), babs = (unsigned)max(b, -b);
  const unsigned den = aabs + babs, den1 = max(den, 1u);
  unsigned dabs;   // ||b| - |a||
  asm("v_sad_u32 %0, %1, %2, 0" : "=v"(dabs) : "v"(babs), "v"(aabs));
  const float est = ((float)dabs * __builtin_amdgcn_rcpf((float)den1)) * 4095.99609375f;   // 4096 (1 - 2^-20)
  unsigned q = (unsigned)est;
  const unsigned r = (dabs << 12) - __umul24(q, den1);   // in [0, 2 den)
  q += (r >= den1);
  // sign of the quotient term: (|b| < |a|) xor (b < 0)
  const int sx = ((int)(babs - aabs) ^ b) >> 31;           // -1: the term is negative
  const int nterm = sx - ((int)q ^ sx);                    // -(+-q)
  int angle = nterm + ((b >> 31) & 8192) + 4096;           // base 4096 (b >= 0) or 12288, minus the term
  if (den == 0) angle = 0;
  const int sa = a >> 31;
  return ((angle >> 1) ^ sa) - sa;                         // a < 0 negates; angle >= 0, so trunc(-angle / 2) = -(angle >> 1)
}
