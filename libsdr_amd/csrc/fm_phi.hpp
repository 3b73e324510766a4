// fast_atan2<int16_t,int16_t>(a, b) / 2 — the angle FMDemod differences (reference src/math.hh:31-40, src/demod.hh:246),
// for a, b in the int16 range (the int8 chain calls it with int8 values: same formula, src/math.hh:12-29 differs only
// in the argument type).
//
//   b >= 0: angle = pi/4 - pi/4 * (b - |a|) / (b + |a|);   b < 0: angle = 3pi/4 - pi/4 * (b + |a|) / (|a| - b)
//   with pi/4 = 4096, integer (truncating) division; a < 0 negates; (0, 0) gives 0; the demodulator halves (trunc).
//
// Both branches divide by den = |a| + |b| and their numerators are +-(|b| - |a|) * 4096, so there is ONE unsigned division
// q = floor(4096 * ||b| - |a|| / den), q <= 4096, and angle = (b >= 0 ? 4096 : 12288) - sigma * q with
// sigma = sign(|b| - |a|) * sign(b) (sign(0) = +1 for b).
//
// Round 3: the whole evaluation runs on FLOAT registers holding exact integers (everything stays below 2^24, the one
// larger product sits inside an fma whose exact result is small): |.| and negation are operand modifiers, the sign
// transfers are one v_bfi each, the remainder of the division is one fma. 28 issue slots per angle (the integer form:
// 37; the generic two-branch form of round 1: 44).
//   division: a float estimate biased DOWN by 2^-20 (v_rcp_f32 is good to 1 ulp, the two multiplies to half an ulp each:
//   the estimate lies in (x - 0.005, x], so its floor is q or q - 1), then the exact remainder r = 4096 d - q' den by fma
//   (r is an integer in [0, 2 den), den < 2^17: representable, so the single rounding returns it exactly) and one test.
#pragma once
#include <hip/hip_runtime.h>

// the angle as a float holding an exact integer in [-8192, 8192]
__device__ __forceinline__ float fm_phi_f(int a, int b) {
  const float af = (float)a, bf = (float)b;                 // exact (|.| <= 2^15)
  const float den = __builtin_fabsf(af) + __builtin_fabsf(bf);   // <= 2^16, exact
  const float d = __builtin_fabsf(bf) - __builtin_fabsf(af);     // exact
  const float den1 = __builtin_fmaxf(den, 1.0f);
  const float est = (__builtin_fabsf(d) * __builtin_amdgcn_rcpf(den1)) * 4095.99609375f;   // 4096 (1 - 2^-20)
  float q = __builtin_floorf(est);
  const float r = __builtin_fmaf(-q, den1, __builtin_fabsf(d) * 4096.0f);   // exact, in [0, 2 den)
  q += (r >= den1) ? 1.0f : 0.0f;
  // angle = 8192 - copysign(4096, b) - copysign(q, d xor b): base 4096 (b >= 0) or 12288, term's sign sign(d) * sign(b)
  const float sq = __builtin_copysignf(q, __builtin_bit_cast(float, __builtin_bit_cast(unsigned, d) ^ __builtin_bit_cast(unsigned, bf)));
  float angle = 8192.0f - (__builtin_copysignf(4096.0f, bf) + sq);
  if (den == 0.0f) angle = 0.0f;
  const float h = __builtin_floorf(angle * 0.5f);           // angle >= 0: trunc(angle / 2); a < 0 negates: trunc(-angle / 2) = -h
  return __builtin_copysignf(h, af);
}

__device__ __forceinline__ int fm_phi(int a, int b) { return (int)fm_phi_f(a, b); }
