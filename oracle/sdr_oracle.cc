// sdr_oracle.cc — CPU restatement of the libsdr streaming-DSP hot path.
//
// TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg as the checker / baseline; never by the product path.  Written from the behavioural
// description in SURVEY.md §8a (each function cites the reference file:line it restates); it
// shares no text with the reference.  Pinned against golden vectors cut from the compiled
// reference (tests/golden/, oracle/ref_driver.cc) by tests/test_oracle_golden.py.
// The FFT-convolution part is PARITY UNPINNED at the FFTW3 boundary (see sdr_oracle.h).
//
// Build: g++ -O3 -fPIC -ffp-contract=off -shared (oracle/Makefile) — no -march, no fast-math,
// mirroring the reference's release flags (CMakeLists.txt:51-54).
#include "sdr_oracle.h"

#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <chrono>

namespace {

// two's-complement wrapping int32 arithmetic (what the x86-64 reference binary does for the
// signed-overflow corners SURVEY §7 lists); written on uint32 so the oracle itself has no UB.
inline int32_t addw(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
inline int32_t subw(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
inline int32_t mulw(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
inline int32_t divt(int32_t a, int32_t b) {           // C++ '/' : truncates toward zero
  if (b == 0) return 0;                                // reference would trap (D*D wrapped to 0)
  if (a == INT32_MIN && b == -1) return INT32_MIN;
  return a / b;
}
inline int16_t wrap16(int32_t v) { return (int16_t)(uint16_t)(uint32_t)v; }
// double -> int16 as gcc/x86-64 does it: cvttsd2si to int32, keep the low 16 bits
inline int16_t d2i16(double d) { return wrap16((int32_t)d); }

struct C32 { int32_t re, im; };

// libstdc++ generic complex<int>::operator/=(complex<int>(s, 0))  
inline C32 cdiv_int(C32 a, int32_t s) {
  int32_t r = addw(mulw(a.re, s), mulw(a.im, 0));
  int32_t n = addw(mulw(s, s), 0);
  C32 o;
  o.im = divt(subw(mulw(a.im, s), mulw(a.re, 0)), n);
  o.re = divt(r, n);
  return o;
}

}  // namespace

extern "C" {

// ===============================================================================================
// designers
// ===============================================================================================

// src/baseband.hh:239-262. Ff, Fs, width are held as int32_t members there (:266-272), so the
// doubles are truncated before use.
void orc_iqbb_design(double Ff_, double width_, double Fs_, int order, int32_t *taps) {
  const int32_t Ff = (int32_t)Ff_, Fs = (int32_t)Fs_, width = (int32_t)width_;
  std::vector< std::complex<double> > a(order);
  const double w = (M_PI * width) / (Fs);
  const double M = double(order) / 2.;
  double norm = 0;
  for (size_t i = 0; i < (size_t)order; i++) {
    double s;
    if ((size_t)order == 2 * i) s = 4 * (w / M_PI);
    else s = std::sin(w * (i - M)) / (w * (i - M));
    a[i] = s;
    a[i] *= std::exp(std::complex<double>(0.0, (-2 * M_PI * Ff * i) / Fs));
    a[i] *= (0.42 - 0.5 * cos((2 * M_PI * i) / order) + 0.08 * cos((4 * M_PI * i) / order));
    norm += std::abs(a[i]);
  }
  for (int i = 0; i < order; i++) {
    std::complex<double> k = (double(1 << 14) * a[i]) / norm;
    taps[2 * i] = (int32_t)k.real();      // complex<int32> = complex<double> : trunc per component
    taps[2 * i + 1] = (int32_t)k.imag();
  }
}

// src/baseband.hh:159-162 (_Fs is the int32 member; _sub_sample is size_t)
int orc_iqbb_decim(double Fs_, int sub_sample, double oFs) {
  const int32_t Fs = (int32_t)Fs_;
  size_t sub = (size_t)sub_sample;
  if (oFs > 0) {
    sub = (size_t)(Fs / oFs);
    if (sub < 1) sub = 1;
  }
  return (int)sub;
}

// src/freqshift.hh:31-35, Traits<int16_t>::shift = 16 (src/traits.cc:22)
void orc_freqshift_lut_i16(int32_t *lut) {
  const size_t L = 128;
  for (size_t i = 0; i < L; i++) {
    std::complex<double> v = double(1 << 16) * std::exp(std::complex<double>(0, -(2 * M_PI * i) / L));
    lut[2 * i] = (int32_t)v.real();
    lut[2 * i + 1] = (int32_t)v.imag();
  }
}

// src/freqshift.hh:78-87
uint32_t orc_freqshift_inc(double F, double Fs) {
  const size_t L = 128;
  size_t inc = (size_t)((L * (1 << 8) * std::abs(F)) / Fs);
  return (uint32_t)inc;
}

// src/firfilter.hh:16-32
void orc_fir_lowpass_design(int N_, double Fu, double Fs, double *alpha) {
  const size_t N = (size_t)N_;
  const double w = 2 * M_PI * Fu / Fs;
  const double M = double(N) / 2;
  double norm = 0;
  for (size_t i = 0; i < N; i++) {
    if (N == 2 * i) alpha[i] = 4 * w / M_PI;
    else alpha[i] = std::sin(w * (i - M)) / (w * (i - M));
    alpha[i] *= (0.42 - 0.5 * cos((2 * M_PI * i) / N) + 0.08 * cos((4 * M_PI * i) / N));
    norm += std::abs(alpha[i]);
  }
  for (size_t i = 0; i < N; i++) alpha[i] /= norm;
}

// src/filternode.hh:18-28 (sinc_flt_kernel<float>) and :186-196 (band clamp). The modulation
// phase is rounded to float BEFORE exp; N/2 is integer division.
void orc_fftfilt_design_h(int N, double fmin_, double fmax_, double Fs, float *h) {
  const double fmin = std::max(fmin_, -Fs / 2);
  const double fmax = std::min(fmax_, Fs / 2);
  const double bw = fmax - fmin;
  const double Fc = fmin + bw / 2;
  for (int i = 0; i < N; i++) {
    std::complex<float> v;
    if ((N / 2) == i) v = M_PI * (bw / Fs);
    else v = std::sin(M_PI * (bw / Fs) * (i - N / 2)) / (i - N / 2);
    v *= std::exp(std::complex<float>(0.0, (2 * M_PI * Fc * i) / Fs));
    v *= (0.42 - 0.5 * cos((2 * M_PI * i) / N) + 0.08 * cos((4 * M_PI * i) / N));
    h[2 * i] = v.real();
    h[2 * i + 1] = v.imag();
  }
}

// DFT in double of any length n (the reference hands every size to FFTW): decimation in frequency over the prime
// factors of n, smallest first, each stage's butterflies evaluated as the small DFTs they are; the digit-reversed result
// is sorted out at the end. Powers of two run the same code (factor 2 throughout).
void orc_dft_f64(int n, int sign, const double *in, double *out) {
  typedef std::complex<double> cd;
  std::vector<cd> a(n), t(n);
  for (int i = 0; i < n; i++) a[i] = cd(in[2 * i], in[2 * i + 1]);
  std::vector<int> fac;
  for (int m = n, q = 2; m > 1;) { if (m % q == 0) { fac.push_back(q); m /= q; } else q++; }
  int len = n;                                   // current sub-transform length
  for (size_t f = 0; f < fac.size(); f++) {
    const int r = fac[f], s = len / r;
    std::vector<cd> wr(r);
    for (int q = 0; q < r; q++) { const double ang = sign * 2.0 * M_PI * q / r; wr[q] = cd(std::cos(ang), std::sin(ang)); }
    for (int base = 0; base < n; base += len)
      for (int j = 0; j < s; j++) {
        for (int m = 0; m < r; m++) {
          cd acc(0, 0);
          for (int k = 0; k < r; k++) acc += a[base + j + k * s] * wr[(k * m) % r];
          const double ang = sign * 2.0 * M_PI * (double)((long)j * m) / (double)len;
          t[base + j + m * s] = acc * cd(std::cos(ang), std::sin(ang));
        }
      }
    a.swap(t);
    len = s;
  }
  // position -> frequency: position = sum_f d_f * (len after stage f), frequency = sum_f d_f * (product of the radices before f)
  for (int pos = 0; pos < n; pos++) {
    int rem = pos, l = n, k = 0, mult = 1;
    for (size_t f = 0; f < fac.size(); f++) { const int s = l / fac[f], d = rem / s; rem -= d * s; k += d * mult; mult *= fac[f]; l = s; }
    out[2 * k] = a[pos].real(); out[2 * k + 1] = a[pos].imag();
  }
}

// sinc_flt_kernel<double> (src/filternode.hh:16-28) with FilterSource<double>::_updateFilter's band clamp (:186-196)
void orc_fftfilt_design_h_f64(int N, double fmin_, double fmax_, double Fs, double *h) {
  const double fmin = std::max(fmin_, -Fs / 2);
  const double fmax = std::min(fmax_, Fs / 2);
  const double bw = fmax - fmin;
  const double Fc = fmin + bw / 2;
  for (int i = 0; i < N; i++) {
    std::complex<double> v;
    if ((N / 2) == i) v = M_PI * (bw / Fs);
    else v = std::sin(M_PI * (bw / Fs) * (i - N / 2)) / (i - N / 2);
    v *= std::exp(std::complex<double>(0.0, (2 * M_PI * Fc * i) / Fs));
    v *= (0.42 - 0.5 * cos((2 * M_PI * i) / N) + 0.08 * cos((4 * M_PI * i) / N));
    h[2 * i] = v.real();
    h[2 * i + 1] = v.imag();
  }
}

// FilterSource<double>::_updateFilter (:197-202) and ::process (:164-181) in double: K = DFT_2N([h, 0]) / ||K||_2;
// one block: out = last + IDFT(DFT([x, 0]) K) / 2N, last = second half. `last` holds N complex doubles of state.
void orc_fftfilt_design_K_f64(int N, const double *h, double *K) {
  const int L = 2 * N;
  std::vector<double> in(2 * L, 0.0);
  for (int i = 0; i < 2 * N; i++) in[i] = h[i];
  orc_dft_f64(L, -1, in.data(), K);
  double nrm2 = 0;
  for (int i = 0; i < L; i++) nrm2 += K[2 * i] * K[2 * i] + K[2 * i + 1] * K[2 * i + 1];
  const double d = std::sqrt(nrm2);
  for (int i = 0; i < 2 * L; i++) K[i] = K[i] / d;
}
void orc_fftfilt_process_f64(int N, const double *K, double *last, const double *in, double *out) {
  const int L = 2 * N;
  std::vector<double> a(2 * L, 0.0), b(2 * L);
  for (int i = 0; i < 2 * N; i++) a[i] = in[i];
  orc_dft_f64(L, -1, a.data(), b.data());
  for (int i = 0; i < L; i++) {
    const double re = b[2 * i] * K[2 * i] - b[2 * i + 1] * K[2 * i + 1], im = b[2 * i] * K[2 * i + 1] + b[2 * i + 1] * K[2 * i];
    a[2 * i] = re; a[2 * i + 1] = im;
  }
  orc_dft_f64(L, +1, a.data(), b.data());
  for (int i = 0; i < 2 * N; i++) { out[i] = last[i] + b[i] / (double)L; last[i] = b[2 * N + i] / (double)L; }
}

// src/filternode.hh:197-202: K = DFT_2N([h, 0]) stored as float, then K /= norm2(K) where
// norm2 accumulates real(conj(k)*k) (float product) into a double (src/buffer.hh:182-188) and
// the division is complex<float> /= complex<float>(float(norm2), 0) = componentwise.
void orc_fftfilt_design_K(int N, const float *h, float *K) {
  const int L = 2 * N;
  std::vector<double> in(2 * L, 0.0), out(2 * L);
  for (int i = 0; i < 2 * N; i++) in[i] = h[i];
  orc_dft_f64(L, -1, in.data(), out.data());
  double nrm2 = 0;
  for (int i = 0; i < L; i++) {
    K[2 * i] = (float)out[2 * i]; K[2 * i + 1] = (float)out[2 * i + 1];
    float p = K[2 * i] * K[2 * i] + K[2 * i + 1] * K[2 * i + 1];
    nrm2 += p;
  }
  const float d = (float)std::sqrt(nrm2);
  for (int i = 0; i < 2 * L; i++) K[i] = K[i] / d;
}

// ===============================================================================================
// IQSigGen (src/siggen.hh:116-131): scale = 1 for every IQSigGen<T>;
// t accumulates dt in double; each sine is added with a conversion through the sample type.
// ===============================================================================================
struct SigGen { double dt, t; std::vector<double> f, a, p; };

void *orc_iqsiggen_create(double Fs) { SigGen *g = new SigGen; g->dt = 1. / Fs; g->t = 0; return g; }
void orc_iqsiggen_add_sine(void *gp, double f, double a, double phi) {
  SigGen *g = (SigGen *)gp; g->f.push_back(f); g->a.push_back(a); g->p.push_back(phi);
}
void orc_iqsiggen_destroy(void *gp) { delete (SigGen *)gp; }

void orc_iqsiggen_next_cs16(void *gp, size_t n, int16_t *out) {
  SigGen *g = (SigGen *)gp;
  const double scale = 1, ns = double(g->f.size());
  for (size_t i = 0; i < n; i++) {
    int16_t re = 0, im = 0;
    for (size_t s = 0; s < g->f.size(); s++) {
      std::complex<double> v =
          (scale * (g->a[s] * std::exp(std::complex<double>(0, 2 * M_PI * g->f[s] * g->t + g->p[s])))) / ns;
      re = d2i16((double)re + v.real());    // complex<int16> += complex<double>
      im = d2i16((double)im + v.imag());
    }
    out[2 * i] = re; out[2 * i + 1] = im;
    g->t += g->dt;
  }
}

void orc_iqsiggen_next_cf32(void *gp, size_t n, float *out) {
  SigGen *g = (SigGen *)gp;
  const double scale = 1, ns = double(g->f.size());
  for (size_t i = 0; i < n; i++) {
    float re = 0, im = 0;
    for (size_t s = 0; s < g->f.size(); s++) {
      std::complex<double> v =
          (scale * (g->a[s] * std::exp(std::complex<double>(0, 2 * M_PI * g->f[s] * g->t + g->p[s])))) / ns;
      re = (float)((double)re + v.real());  // complex<float> += complex<double> adds in double
      im = (float)((double)im + v.imag());
    }
    out[2 * i] = re; out[2 * i + 1] = im;
    g->t += g->dt;
  }
}

// ===============================================================================================
// The int8 chain of the documentation example (src/sdr.hh:225-240): IQBaseBand<int8_t> -> FMDemod<int8_t,int16_t>.
// IQBaseBand's own compute type is int32 for every Scalar (src/baseband.hh:28-31): ring, kernel, FIR sum, running
// window sum and its division are exactly the int16 node's. What differs is the frequency shift it inherits from
// FreqShiftBase<int8_t>, whose compute type is Traits<int8_t>::SScalar = int16 (src/freqshift.hh:18-22, src/traits.hh:
// 58-73): the FIR value is CONVERTED to complex<int16_t> when it is passed to applyFrequencyShift (wrap mod 2^16, also
// when the shift is zero), the LUT holds 2^8 * exp(..) as int16, the complex product is stored to complex<int16_t>
// (wrap) and then shifted by Traits<int8_t>::shift = 8 (src/freqshift.hh:58-74, src/traits.cc:11). The output is the
// window average wrapped to int8.
// ===============================================================================================
static inline int32_t wrap8(int32_t v) { return (int32_t)(int8_t)(uint8_t)(uint32_t)v; }

void orc_freqshift_lut_i8(int32_t *lut) {   // FreqShiftBase<int8_t> ctor (src/freqshift.hh:31-35): 2^8 * exp(-2 pi i k/128) -> int16
  for (size_t i = 0; i < 128; i++) {
    std::complex<double> v = double(1 << 8) * std::exp(std::complex<double>(0, -(2 * M_PI * i) / 128));
    lut[2 * i] = (int32_t)(int16_t)v.real();
    lut[2 * i + 1] = (int32_t)(int16_t)v.imag();
  }
}

struct IQBB8 {
  int order, decim, negative; uint32_t inc;
  std::vector<C32> k, ring, lut;   // (the LUT's int16 values kept in int32 fields)
  size_t off, count, lut_count; C32 last;
};

void *orc_iqbb_i8_create(const int32_t *taps, int order, const int32_t *lut, uint32_t inc, int negative, int decim) {
  IQBB8 *s = new IQBB8;
  s->order = order; s->decim = decim; s->negative = negative; s->inc = inc;
  s->k.resize(order); s->ring.assign(order, C32{0, 0}); s->lut.resize(128);
  for (int i = 0; i < order; i++) { s->k[i].re = taps[2 * i]; s->k[i].im = taps[2 * i + 1]; }
  for (int i = 0; i < 128; i++) { s->lut[i].re = wrap16(lut[2 * i]); s->lut[i].im = wrap16(lut[2 * i + 1]); }
  s->off = 0; s->count = 0; s->lut_count = 0; s->last = C32{0, 0};
  return s;
}
void orc_iqbb_i8_destroy(void *h) { delete (IQBB8 *)h; }

size_t orc_iqbb_i8_process(void *h, const int8_t *in, size_t n, int8_t *out) {
  IQBB8 *s = (IQBB8 *)h;
  const size_t order = (size_t)s->order, D = (size_t)s->decim;
  size_t j = 0;
  for (size_t i = 0; i < n; i++) {
    s->ring[s->off] = C32{in[2 * i], in[2 * i + 1]};
    C32 acc = {0, 0};
    size_t idx = s->off + 1; if (idx == order) idx = 0;
    for (size_t t = 0; t < order; t++, idx++) {
      if (idx == order) idx = 0;
      const C32 a = s->k[t], b = s->ring[idx];
      acc.re = addw(acc.re, subw(mulw(a.re, b.re), mulw(a.im, b.im)));
      acc.im = addw(acc.im, addw(mulw(a.re, b.im), mulw(a.im, b.re)));
    }
    C32 v = {wrap16(acc.re >> 14), wrap16(acc.im >> 14)};   // complex<int32> -> complex<int16> at the call (:206)
    if (s->inc != 0) {
      size_t li = s->lut_count >> 8;
      if (s->negative) li = 128 - li - 1;
      const C32 L = s->lut[li];
      const int32_t pre = wrap16(subw(mulw(L.re, v.re), mulw(L.im, v.im)));   // complex<int16> * complex<int16>
      const int32_t pim = wrap16(addw(mulw(L.re, v.im), mulw(L.im, v.re)));
      v.re = pre >> 8; v.im = pim >> 8;
      s->lut_count += s->inc;
      while (s->lut_count >= (128u << 8)) s->lut_count -= (128u << 8);
    }
    s->last.re = addw(s->last.re, v.re); s->last.im = addw(s->last.im, v.im);
    s->off++; if (s->off == order) s->off = 0;
    if (D == s->count) {
      C32 q = cdiv_int(s->last, (int32_t)D);
      out[2 * j] = (int8_t)wrap8(q.re); out[2 * j + 1] = (int8_t)wrap8(q.im);
      s->last = C32{0, 0}; s->count = 0; j++;
    } else if (D == 1) {
      out[2 * j] = (int8_t)wrap8(s->last.re); out[2 * j + 1] = (int8_t)wrap8(s->last.im);
      s->last = C32{0, 0}; s->count = 0; j++;
    }
    s->count++;
  }
  return j;
}

// FMDemod<int8_t,int16_t>::_process (src/demod.hh:242-254) with fast_atan2<int8_t,int16_t> (src/math.hh:12-21: the
// same formula as the int16 form); in place, out[i] (2 bytes) covers exactly sample i (2 bytes)
void orc_fm_i8(const int8_t *in, size_t n, int16_t *out, int16_t *last) {
  for (size_t i = 1; i < n; i++) {
    const int16_t phi = (int16_t)(orc_fast_atan2_i16(in[2 * i], in[2 * i + 1]) / 2);
    out[i] = wrap16((int32_t)*last - (int32_t)phi);
    *last = phi;
  }
}

// ===============================================================================================
// IQBaseBand<int16_t>::_process (src/baseband.hh:198-223), _filter_ring (:226-236),
// FreqShiftBase::applyFrequencyShift (src/freqshift.hh:58-74)
// ===============================================================================================
struct IQBB {
  int order, decim, negative; uint32_t inc;
  std::vector<C32> k, ring, lut;
  size_t off, count, lut_count; C32 last;
};

void *orc_iqbb_i16_create(const int32_t *taps, int order, const int32_t *lut, uint32_t inc,
                          int negative, int decim) {
  IQBB *s = new IQBB;
  s->order = order; s->decim = decim; s->negative = negative; s->inc = inc;
  s->k.resize(order); s->ring.assign(order, C32{0, 0}); s->lut.resize(128);
  for (int i = 0; i < order; i++) { s->k[i].re = taps[2 * i]; s->k[i].im = taps[2 * i + 1]; }
  for (int i = 0; i < 128; i++) { s->lut[i].re = lut[2 * i]; s->lut[i].im = lut[2 * i + 1]; }
  s->off = 0; s->count = 0; s->lut_count = 0; s->last = C32{0, 0};
  return s;
}
void orc_iqbb_i16_destroy(void *h) { delete (IQBB *)h; }
// setFilterFrequency / setFilterWidth (src/baseband.hh:92-104): _update_filter_kernel() only — the kernel is swapped,
// ring, ring offset, decimator state and LUT phase go on as they are
void orc_iqbb_i16_set_taps(void *h, const int32_t *taps) {
  IQBB *s = (IQBB *)h;
  for (int i = 0; i < s->order; i++) { s->k[i].re = taps[2 * i]; s->k[i].im = taps[2 * i + 1]; }
}
// setCenterFrequency (src/baseband.hh:84-86) -> setFrequencyShift -> _update_lut_incr (src/freqshift.hh:52-54,78-87):
// new increment and sign, the LUT phase counter restarts at 0; nothing else changes
void orc_iqbb_i16_set_shift(void *h, uint32_t inc, int negative) {
  IQBB *s = (IQBB *)h; s->inc = inc; s->negative = negative; s->lut_count = 0;
}
void orc_iqbb_i16_reset(void *h) {   // _reconfigure (:175-177) + setSampleRate -> _update_lut_incr
  IQBB *s = (IQBB *)h; s->off = 0; s->count = 0; s->lut_count = 0; s->last = C32{0, 0};
}
// setSubsample / setOutputSampleRate (src/baseband.hh:106-112): the new _sub_sample, then _reconfigure (call
// orc_iqbb_i16_reset next); the ring is not touched
void orc_iqbb_i16_set_decim(void *h, int decim) { ((IQBB *)h)->decim = decim; }
// setOrder (src/baseband.hh:69-79): _kernel and _ring are REALLOCATED (the new ring is uninitialised memory in the
// reference; zeros here — outputs are defined again once `order` samples have passed), _update_filter_kernel();
// _ring_offset, _sample_count, _last and the LUT phase are not touched. (A ring offset at or beyond the new order is an
// out-of-bounds write in the reference — undefined; here the offset restarts at 0.)
void orc_iqbb_i16_set_order(void *h, const int32_t *taps, int order) {
  IQBB *s = (IQBB *)h;
  s->order = order; s->k.resize(order); s->ring.assign(order, C32{0, 0});
  for (int i = 0; i < order; i++) { s->k[i].re = taps[2 * i]; s->k[i].im = taps[2 * i + 1]; }
  if (s->off >= (size_t)order) s->off = 0;
}

// Test-bench helper (no counterpart in the reference): put the decimator and the LUT phase where they stand when the NEXT
// sample is absolute index `abs_index` of a stream, for an index right behind an emission (abs_index = g*D + 1, g >= 1:
// after the D+1 first window every window is D samples, emitted at indices D, 2D, ... — _process :200-214). The ring is
// left as it is: prime it with the `order` samples in front of abs_index first. Returns 0, or -1 for an index that is
// not such a boundary.
int orc_iqbb_i16_seek(void *h, uint64_t abs_index) {
  IQBB *s = (IQBB *)h;
  const uint64_t D = (uint64_t)s->decim;
  if (D < 2 || abs_index < D + 1 || (abs_index - 1) % D != 0) return -1;
  s->count = 1; s->last = C32{0, 0};
  s->lut_count = (size_t)((abs_index % 32768u) * (uint64_t)(s->inc % 32768u) % 32768u);   // lut_count = abs_index * inc mod 128 * 256
  return 0;
}

size_t orc_iqbb_i16_process(void *h, const int16_t *in, size_t n, int16_t *out) {
  IQBB *s = (IQBB *)h;
  const size_t order = (size_t)s->order, D = (size_t)s->decim;
  size_t j = 0;
  for (size_t i = 0; i < n; i++) {
    s->ring[s->off] = C32{in[2 * i], in[2 * i + 1]};
    // FIR over the ring, oldest sample first; k[order-1] meets the newest sample
    C32 acc = {0, 0};
    size_t idx = s->off + 1; if (idx == order) idx = 0;
    for (size_t t = 0; t < order; t++, idx++) {
      if (idx == order) idx = 0;
      const C32 a = s->k[t], b = s->ring[idx];
      acc.re = addw(acc.re, subw(mulw(a.re, b.re), mulw(a.im, b.im)));
      acc.im = addw(acc.im, addw(mulw(a.re, b.im), mulw(a.im, b.re)));
    }
    C32 v = {acc.re >> 14, acc.im >> 14};
    // frequency shift by LUT (skipped entirely when the increment is zero)
    if (s->inc != 0) {
      size_t li = s->lut_count >> 8;
      if (s->negative) li = 128 - li - 1;
      const C32 L = s->lut[li];
      C32 p;
      p.re = subw(mulw(L.re, v.re), mulw(L.im, v.im));
      p.im = addw(mulw(L.re, v.im), mulw(L.im, v.re));
      v.re = p.re >> 16; v.im = p.im >> 16;
      s->lut_count += s->inc;
      while (s->lut_count >= (128u << 8)) s->lut_count -= (128u << 8);
    }
    s->last.re = addw(s->last.re, v.re); s->last.im = addw(s->last.im, v.im);
    s->off++; if (s->off == order) s->off = 0;
    if (D == s->count) {
      C32 q = cdiv_int(s->last, (int32_t)D);
      out[2 * j] = wrap16(q.re); out[2 * j + 1] = wrap16(q.im);
      s->last = C32{0, 0}; s->count = 0; j++;
    } else if (D == 1) {
      out[2 * j] = wrap16(s->last.re); out[2 * j + 1] = wrap16(s->last.im);
      s->last = C32{0, 0}; s->count = 0; j++;
    }
    s->count++;   // the for-header increment (:200) runs after the body, also after a reset
  }
  return j;
}

// ===============================================================================================
// FIRFilter<Scalar,...>::_process (src/firfilter.hh:231-247) with the mixed-type operators of
// src/operators.hh:16-18,24-26: one conversion back to Scalar per tap.
// ===============================================================================================
struct FIR { int order; std::vector<double> a; std::vector<double> rr, ri; size_t off; };

void *orc_fir_create(const double *alpha, int order) {
  FIR *s = new FIR; s->order = order; s->a.assign(alpha, alpha + order);
  s->rr.assign(order, 0.0); s->ri.assign(order, 0.0); s->off = 0; return s;
}
void orc_fir_destroy(void *h) { delete (FIR *)h; }
void orc_fir_reset(void *h) {
  FIR *s = (FIR *)h; std::fill(s->rr.begin(), s->rr.end(), 0.0); std::fill(s->ri.begin(), s->ri.end(), 0.0); s->off = 0;
}

void orc_fir_cs16_process(void *h, const int16_t *in, size_t n, int16_t *out) {
  FIR *s = (FIR *)h; const size_t order = (size_t)s->order;
  for (size_t i = 0; i < n; i++) {
    s->rr[s->off] = (double)in[2 * i]; s->ri[s->off] = (double)in[2 * i + 1];   // int16 -> double is exact
    s->off++; if (s->off == order) s->off = 0;
    int16_t re = 0, im = 0;
    size_t idx = s->off;
    for (size_t j = 0; j < order; j++, idx++) {
      if (idx == order) idx = 0;
      re = d2i16((double)re + s->a[j] * s->rr[idx]);
      im = d2i16((double)im + s->a[j] * s->ri[idx]);
    }
    out[2 * i] = re; out[2 * i + 1] = im;
  }
}

void orc_fir_cf32_process(void *h, const float *in, size_t n, float *out) {
  FIR *s = (FIR *)h; const size_t order = (size_t)s->order;
  for (size_t i = 0; i < n; i++) {
    s->rr[s->off] = (double)in[2 * i]; s->ri[s->off] = (double)in[2 * i + 1];
    s->off++; if (s->off == order) s->off = 0;
    float re = 0, im = 0;
    size_t idx = s->off;
    for (size_t j = 0; j < order; j++, idx++) {
      if (idx == order) idx = 0;
      re = (float)((double)re + s->a[j] * s->rr[idx]);
      im = (float)((double)im + s->a[j] * s->ri[idx]);
    }
    out[2 * i] = re; out[2 * i + 1] = im;
  }
}

// ===============================================================================================
// demodulators
// ===============================================================================================

// src/math.hh:31-40
int16_t orc_fast_atan2_i16(int16_t a, int16_t b) {
  const int32_t pi4 = (1 << 12), pi34 = 3 * (1 << 12);
  if (a == 0 && b == 0) return 0;
  const int32_t aabs = (a >= 0) ? a : -(int32_t)a;
  int32_t angle;
  if (b >= 0) angle = pi4 - pi4 * (b - aabs) / (b + aabs);
  else angle = pi34 - pi4 * (b + aabs) / (aabs - b);
  return (int16_t)((a >= 0) ? angle : -angle);
}

// src/demod.hh:242-254
void orc_fm_i16(const int16_t *in, size_t n, int16_t *out, int16_t *last) {
  for (size_t i = 1; i < n; i++) {
    const int16_t re = in[2 * i], im = in[2 * i + 1];   // read before out[i] may overwrite in[i/2]
    const int16_t phi = (int16_t)(orc_fast_atan2_i16(re, im) / 2);
    out[i] = wrap16((int32_t)*last - (int32_t)phi);
    *last = phi;
  }
}

// src/demod.hh:73-76: int products/sum in int, sqrt in double, truncation to int16
void orc_am_i16(const int16_t *in, size_t n, int16_t *out) {
  for (size_t i = 0; i < n; i++) {
    const int32_t re = in[2 * i], im = in[2 * i + 1];
    const int32_t m = addw(mulw(re, re), mulw(im, im));
    out[i] = d2i16(std::sqrt((double)m));
  }
}
void orc_am_f32(const float *in, size_t n, float *out) {
  for (size_t i = 0; i < n; i++) { const float re = in[2 * i], im = in[2 * i + 1]; out[i] = std::sqrt(re * re + im * im); }
}
// src/demod.hh:156-161
void orc_usb_i16(const int16_t *in, size_t n, int16_t *out) {
  for (size_t i = 0; i < n; i++) out[i] = wrap16(((int32_t)in[2 * i] + (int32_t)in[2 * i + 1]) / 2);
}
void orc_usb_f32(const float *in, size_t n, float *out) {
  for (size_t i = 0; i < n; i++) out[i] = (in[2 * i] + in[2 * i + 1]) / 2;
}

// ===============================================================================================
// SubSample (src/subsample.hh:92-101)
// ===============================================================================================
struct Sub { size_t n, left; C32 acc; float fre, fim; };
void *orc_subsample_create(size_t n) { Sub *s = new Sub; s->n = n; s->left = 0; s->acc = C32{0, 0}; s->fre = s->fim = 0; return s; }
void orc_subsample_destroy(void *h) { delete (Sub *)h; }

size_t orc_subsample_cs16_process(void *h, const int16_t *in, size_t n, int16_t *out) {
  Sub *s = (Sub *)h; size_t j = 0;
  for (size_t i = 0; i < n; i++) {
    s->acc.re = addw(s->acc.re, in[2 * i]); s->acc.im = addw(s->acc.im, in[2 * i + 1]); s->left++;
    if (s->n <= s->left) {
      C32 q = cdiv_int(s->acc, (int32_t)s->n);
      out[2 * j] = wrap16(q.re); out[2 * j + 1] = wrap16(q.im);
      j++; s->acc = C32{0, 0}; s->left = 0;
    }
  }
  return j;
}
size_t orc_subsample_cf32_process(void *h, const float *in, size_t n, float *out) {
  Sub *s = (Sub *)h; size_t j = 0;
  const float d = (float)s->n;   // complex<float>(size_t) ; division by (d, 0) is componentwise
  for (size_t i = 0; i < n; i++) {
    s->fre += in[2 * i]; s->fim += in[2 * i + 1]; s->left++;
    if (s->n <= s->left) {
      out[2 * j] = s->fre / d; out[2 * j + 1] = s->fim / d;
      j++; s->fre = s->fim = 0; s->left = 0;
    }
  }
  return j;
}

// ===============================================================================================
// FFT convolution: FilterSink::process (src/filternode.hh:81-88) + FilterSource::process
// (:164-181).  PARITY UNPINNED at the FFTW boundary: DFTs in double, rounded to float where the
// reference holds complex<float> buffers.
// ===============================================================================================
struct FFTFilt { int N; std::vector<float> K, last; };
void *orc_fftfilt_create(int N, const float *K) {
  FFTFilt *s = new FFTFilt; s->N = N; s->K.assign(K, K + 4 * N); s->last.assign(2 * N, 0.f); return s;
}
void orc_fftfilt_destroy(void *h) { delete (FFTFilt *)h; }

void orc_fftfilt_process(void *h, const float *in, float *out) {
  FFTFilt *s = (FFTFilt *)h; const int N = s->N, L = 2 * N;
  std::vector<double> a(2 * L, 0.0), b(2 * L);
  for (int i = 0; i < 2 * N; i++) a[i] = in[i];
  orc_dft_f64(L, -1, a.data(), b.data());
  for (int i = 0; i < L; i++) {
    const float xr = (float)b[2 * i], xi = (float)b[2 * i + 1];   // FFTW3f output is float
    const float kr = s->K[2 * i], ki = s->K[2 * i + 1];
    a[2 * i] = (double)(xr * kr - xi * ki);                        // complex<float> product
    a[2 * i + 1] = (double)(xr * ki + xi * kr);
  }
  orc_dft_f64(L, +1, a.data(), b.data());
  const float sc = (float)L;
  for (int i = 0; i < N; i++) {
    for (int c = 0; c < 2; c++) {
      const float y0 = (float)b[2 * i + c], y1 = (float)b[2 * (i + N) + c];
      out[2 * i + c] = s->last[2 * i + c] + y0 / sc;
      s->last[2 * i + c] = y1 / sc;
    }
  }
}

// build-defined float frequency shift — PARITY UNPINNED, no reference node.
void orc_freqshift_cf32(const float *in, size_t n, uint64_t n0, double Fc, double Fs, float *out) {
  for (size_t i = 0; i < n; i++) {
    const double ph = -2.0 * M_PI * std::fmod(Fc * (double)(n0 + i) / Fs, 1.0);
    const double c = std::cos(ph), s = std::sin(ph);
    const double xr = in[2 * i], xi = in[2 * i + 1];
    out[2 * i] = (float)(xr * c - xi * s);
    out[2 * i + 1] = (float)(xr * s + xi * c);
  }
}

// ===============================================================================================
// "next" rows: AutoCast cu8 -> cs16 (src/autocast.hh:187-194) and FMDeemph<int16_t> (src/demod.hh:305-351)
// ===============================================================================================
void orc_autocast_cu8_cs16(const uint8_t *in, size_t n_bytes, int16_t *out) {
  for (size_t i = 0; i < n_bytes; i++) {
    const int8_t b = (int8_t)in[i];                       // the reference reads the unsigned bytes through an int8_t*
    out[i] = wrap16((int32_t)((uint32_t)((int32_t)b - 127) << 8));
  }
}

int orc_fmdeemph_alpha(double sample_rate) {
  return (int)round(1.0 / ((1.0 - exp(-1.0 / (sample_rate * 75e-6)))));
}

void orc_fmdeemph_i16(const int16_t *in, size_t n, int alpha, int16_t *avg, int16_t *out) {
  int16_t a = *avg;
  for (size_t i = 0; i < n; i++) {
    const int16_t diff = wrap16((int32_t)in[i] - (int32_t)a);
    if (diff > 0) a = wrap16((int32_t)a + ((int32_t)diff + alpha / 2) / alpha);
    else a = wrap16((int32_t)a + ((int32_t)diff - alpha / 2) / alpha);
    out[i] = a;
  }
  *avg = a;
}

// ===============================================================================================
// "next" row 3: BaseBand<int16_t>, the REAL-input variant (src/baseband.hh:305-529)
//   design  _update_filter_kernel :464-491  (Fs, Ff, width are doubles here; centre tap of an even order is 1;
//           +Ff modulation; Blackman on (i+1)/(order+2); Q16 = Traits<int16_t>::shift, src/traits.cc:22)
//   stream  _process :425-445, _filter_ring :448-460  (ring of int32 holding the real samples; complex<int32>
//           taps x real sample, wrapping; >>16; LUT rotation as IQBaseBand; D-sample windows from sample 0)
// ===============================================================================================
void orc_bb_design(double Ff, double width, double Fs, int order, int32_t *taps) {
  std::vector< std::complex<double> > a(order);
  const double w = (2 * M_PI * width) / (2 * Fs);
  const double M = double(order) / 2;
  double norm = 0;
  for (size_t i = 0; i < (size_t)order; i++) {
    if ((size_t)order == (2 * i)) a[i] = 1;
    else a[i] = std::sin(w * (i - M)) / (w * (i - M));
  }
  for (size_t i = 0; i < (size_t)order; i++) {
    a[i] = a[i] * std::exp(std::complex<double>(0, (2 * M_PI * Ff * i) / Fs));
    a[i] *= (0.42 - 0.5 * cos((2 * M_PI * (i + 1)) / (order + 2)) + 0.08 * cos((4 * M_PI * (i + 1)) / (order + 2)));
    norm += std::abs(a[i]);
  }
  for (int i = 0; i < order; i++) {
    std::complex<double> k = (double(1 << 16) * a[i]) / norm;
    taps[2 * i] = (int32_t)k.real();
    taps[2 * i + 1] = (int32_t)k.imag();
  }
}

struct RBB {
  int order, decim, negative; uint32_t inc;
  std::vector<C32> k, lut; std::vector<int32_t> ring;
  size_t off, count, lut_count; C32 last;
};

void *orc_bb_i16_create(const int32_t *taps, int order, const int32_t *lut, uint32_t inc, int negative, int decim) {
  RBB *s = new RBB;
  s->order = order; s->decim = decim; s->negative = negative; s->inc = inc;
  s->k.resize(order); s->ring.assign(order, 0); s->lut.resize(128);
  for (int i = 0; i < order; i++) { s->k[i].re = taps[2 * i]; s->k[i].im = taps[2 * i + 1]; }
  for (int i = 0; i < 128; i++) { s->lut[i].re = lut[2 * i]; s->lut[i].im = lut[2 * i + 1]; }
  s->off = 0; s->count = 0; s->lut_count = 0; s->last = C32{0, 0};
  return s;
}
void orc_bb_i16_destroy(void *h) { delete (RBB *)h; }
void orc_bb_i16_reset(void *h) {   // config() :391-393 + setSampleRate -> _update_lut_incr; the ring keeps its content
  RBB *s = (RBB *)h; s->off = 0; s->count = 0; s->lut_count = 0; s->last = C32{0, 0};
}
// FreqShiftBase::setFrequencyShift (src/freqshift.hh:52-54 -> _update_lut_incr :78-87): new increment and sign, the LUT
// phase counter restarts at 0; ring, decimator and kernel go on
void orc_bb_i16_set_shift(void *h, uint32_t inc, int negative) {
  RBB *s = (RBB *)h; s->inc = inc; s->negative = negative; s->lut_count = 0;
}

// test-bench helper (as orc_iqbb_i16_seek): decimator and LUT phase as they stand in front of absolute sample abs_index = g*D
// (right behind an emission: the real node closes its first window after D samples); the ring is kept
int orc_bb_i16_seek(void *h, uint64_t abs_index) {
  RBB *s = (RBB *)h;
  const uint64_t D = (uint64_t)s->decim;
  if (D < 1 || abs_index % D != 0) return -1;
  s->count = 0; s->last = C32{0, 0};
  s->lut_count = (size_t)((abs_index % 32768u) * (uint64_t)(s->inc % 32768u) % 32768u);
  return 0;
}

size_t orc_bb_i16_process(void *h, const int16_t *in, size_t n, int16_t *out) {
  RBB *s = (RBB *)h;
  const size_t order = (size_t)s->order, D = (size_t)s->decim;
  size_t j = 0;
  for (size_t i = 0; i < n; i++) {
    s->ring[s->off] = in[i];
    C32 acc = {0, 0};
    size_t idx = s->off + 1; if (idx == order) idx = 0;
    for (size_t t = 0; t < order; t++, idx++) {
      if (idx == order) idx = 0;
      // complex<int32> * int32 (libstdc++ operator*(complex, scalar): both parts times the scalar)
      acc.re = addw(acc.re, mulw(s->k[t].re, s->ring[idx]));
      acc.im = addw(acc.im, mulw(s->k[t].im, s->ring[idx]));
    }
    C32 v = {acc.re >> 16, acc.im >> 16};
    if (s->inc != 0) {
      size_t li = s->lut_count >> 8;
      if (s->negative) li = 128 - li - 1;
      const C32 L = s->lut[li];
      C32 p;
      p.re = subw(mulw(L.re, v.re), mulw(L.im, v.im));
      p.im = addw(mulw(L.re, v.im), mulw(L.im, v.re));
      v.re = p.re >> 16; v.im = p.im >> 16;
      s->lut_count += s->inc;
      while (s->lut_count >= (128u << 8)) s->lut_count -= (128u << 8);
    }
    s->last.re = addw(s->last.re, v.re); s->last.im = addw(s->last.im, v.im);
    s->count++;
    s->off++; if (s->off == order) s->off = 0;
    if (D == s->count) {
      C32 q = cdiv_int(s->last, (int32_t)D);
      out[2 * j] = wrap16(q.re); out[2 * j + 1] = wrap16(q.im);
      s->last = C32{0, 0}; s->count = 0; j++;
    }
  }
  return j;
}

// ===============================================================================================
// cpu_baseline helper (kind "port"): IQBaseBand -> FMDemod in place, one thread
// ===============================================================================================
double orc_bench_iqbb_fm(const int32_t *taps, int order, const int32_t *lut, uint32_t inc, int negative,
                         int decim, const int16_t *in, size_t n, size_t nbuf, long *checksum) {
  void *bb = orc_iqbb_i16_create(taps, order, lut, inc, negative, decim);
  std::vector<int16_t> y(2 * (n / (decim > 0 ? decim : 1) + 2));
  int16_t last = 0; long cs = 0;
  auto t0 = std::chrono::steady_clock::now();
  for (size_t b = 0; b < nbuf; b++) {
    size_t m = orc_iqbb_i16_process(bb, in, n, y.data());
    if (m) { orc_fm_i16(y.data(), m, y.data(), &last); cs += y[m - 1]; }
  }
  auto t1 = std::chrono::steady_clock::now();
  orc_iqbb_i16_destroy(bb);
  if (checksum) *checksum = cs;
  return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
