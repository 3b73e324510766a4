// design.hh — host-side designers of the taps / LUT / FFT kernels the GPU nodes take as inputs.
//
// The kernels never design anything on the device: a 1-ulp libm difference may flip a truncation
// (SURVEY §7 "Tap / LUT / kernel provenance"), so the numbers are produced here, on the host, with
// the same operation order as the reference designers, and pinned against golden vectors
// (tests/test_abi.py; tests/cpp/test_gpu_nodes.cc --host-only).  Reference formulas restated (file:line):
//   IQBaseBand::_update_filter_kernel   src/baseband.hh:239-262  (Ff, Fs, width held as int32: :266-272)
//   BaseBand::_update_filter_kernel     src/baseband.hh:464-491  (real-input node, Q16)
//   FreqShiftBase ctor / _update_lut_incr   src/freqshift.hh:26-36, :78-87
//   FIRLowPassCoeffs::coeffs            src/firfilter.hh:16-32
//   sinc_flt_kernel / FilterSource::_updateFilter   src/filternode.hh:18-28, :186-203
// The reference's HighPass/BandPass/BandStop designers are broken and are
// deliberately not reproduced.
#ifndef SDR_GPU_DESIGN_HH
#define SDR_GPU_DESIGN_HH

#include <cmath>
#include <complex>
#include <cstdint>
#include <cstddef>
#include <vector>

namespace sdr {
namespace gpu {
namespace design {

static const size_t kLutSize = 128;

/** Q14 complex band-pass taps of IQBaseBand<int16_t>: order x (re, im). */
inline void iqbbTaps(double filterFreq, double width, double sampleRate, size_t order, int32_t *taps) {
  // the reference node keeps these three in int32_t members
  const int32_t Ff = int32_t(filterFreq), Fs = int32_t(sampleRate), W = int32_t(width);
  std::vector< std::complex<double> > c(order);
  const double w = (M_PI * W) / (Fs);
  const double mid = double(order) / 2.;
  double l1 = 0;
  for (size_t i = 0; i < order; i++) {
    const double arg = w * (i - mid);
    const double sinc = (order == 2 * i) ? 4 * (w / M_PI) : std::sin(arg) / arg;
    const std::complex<double> mod = std::exp(std::complex<double>(0.0, (-2 * M_PI * Ff * i) / Fs));
    const double win = (0.42 - 0.5 * cos((2 * M_PI * i) / order) + 0.08 * cos((4 * M_PI * i) / order));
    c[i] = std::complex<double>(sinc * mod.real(), sinc * mod.imag());
    c[i] = std::complex<double>(c[i].real() * win, c[i].imag() * win);
    l1 += std::abs(c[i]);
  }
  const double q = double(1 << 14);
  for (size_t i = 0; i < order; i++) {
    taps[2 * i] = int32_t((q * c[i].real()) / l1);
    taps[2 * i + 1] = int32_t((q * c[i].imag()) / l1);
  }
}

/** Q16 complex band-pass taps of the real-input BaseBand<int16_t> (reference src/baseband.hh:464-491). */
inline void bbTaps(double filterFreq, double width, double sampleRate, size_t order, int32_t *taps) {
  std::vector< std::complex<double> > c(order);
  const double w = (2 * M_PI * width) / (2 * sampleRate);
  const double mid = double(order) / 2;
  double l1 = 0;
  for (size_t i = 0; i < order; i++) {
    // note: the centre tap of an even order is 1, not the sinc limit times anything
    if (order == (2 * i)) c[i] = 1;
    else c[i] = std::sin(w * (i - mid)) / (w * (i - mid));
  }
  for (size_t i = 0; i < order; i++) {
    c[i] = c[i] * std::exp(std::complex<double>(0, (2 * M_PI * filterFreq * i) / sampleRate));
    c[i] *= (0.42 - 0.5 * cos((2 * M_PI * (i + 1)) / (order + 2)) + 0.08 * cos((4 * M_PI * (i + 1)) / (order + 2)));
    l1 += std::abs(c[i]);
  }
  for (size_t i = 0; i < order; i++) {
    const std::complex<double> k = (double(1 << 16) * c[i]) / l1;
    taps[2 * i] = int32_t(k.real());
    taps[2 * i + 1] = int32_t(k.imag());
  }
}

/** Decimation of IQBaseBand: explicit, or floor(Fs/oFs) (at least 1) when an output rate is given. */
inline size_t iqbbDecimation(double sampleRate, size_t subSample, double outRate) {
  if (outRate > 0) {
    size_t d = size_t(int32_t(sampleRate) / outRate);
    return d < 1 ? 1 : d;
  }
  return subSample;
}

/** Rotation LUT of FreqShiftBase<int16_t>: 128 x (re, im) = trunc(2^16 exp(-2 pi i k/128)). */
inline void freqShiftLutI16(int32_t *lut) {
  for (size_t k = 0; k < kLutSize; k++) {
    const std::complex<double> e = std::exp(std::complex<double>(0, -(2 * M_PI * k) / kLutSize));
    const double s = double(1 << 16);
    lut[2 * k] = int32_t(s * e.real());
    lut[2 * k + 1] = int32_t(s * e.imag());
  }
}

/** Rotation LUT of FreqShiftBase<int8_t> (Traits<int8_t>::shift = 8, compute type int16): trunc(2^8 exp(-2 pi i k/128)). */
inline void freqShiftLutI8(int32_t *lut) {
  for (size_t k = 0; k < kLutSize; k++) {
    const std::complex<double> e = std::exp(std::complex<double>(0, -(2 * M_PI * k) / kLutSize));
    const double s = double(1 << 8);
    lut[2 * k] = int32_t(int16_t(s * e.real()));
    lut[2 * k + 1] = int32_t(int16_t(s * e.imag()));
  }
}

/** Phase increment per sample in 1/256 LUT steps. */
inline uint32_t freqShiftIncrement(double shift, double sampleRate) {
  return uint32_t(size_t((kLutSize * (1 << 8) * std::abs(shift)) / sampleRate));
}

/** Windowed-sinc low-pass, L1-normalised (FIRLowPass(order, Fc): Fl = 0, Fu = Fc). */
inline void firLowPass(size_t order, double upperFreq, double sampleRate, double *alpha) {
  const double w = 2 * M_PI * upperFreq / sampleRate;
  const double mid = double(order) / 2;
  double l1 = 0;
  for (size_t i = 0; i < order; i++) {
    const double arg = w * (i - mid);
    double v = (order == 2 * i) ? 4 * w / M_PI : std::sin(arg) / arg;
    v *= (0.42 - 0.5 * cos((2 * M_PI * i) / order) + 0.08 * cos((4 * M_PI * i) / order));
    alpha[i] = v;
    l1 += std::abs(v);
  }
  for (size_t i = 0; i < order; i++) alpha[i] /= l1;
}

/** FMDeemph filter constant (reference src/demod.hh:305-306). */
inline int fmDeemphAlpha(double sampleRate) {
  return int(round(1.0 / ((1.0 - exp(-1.0 / (sampleRate * 75e-6))))));
}

/** Time-domain kernel of the FFT filter, N complex Scalars: sinc_flt_kernel<Scalar> + FilterSource::_updateFilter's band
 * clamp (reference src/filternode.hh:18-28,186-196). The value is a complex<Scalar> from the first assignment on, so for
 * float the modulation phase is rounded to float before the exponential (a double phase is 3e-5 off). */
template <class Scalar>
inline void fftFilterKernel(int N, double fmin, double fmax, double sampleRate, Scalar *h) {
  const double lo = std::max(fmin, -sampleRate / 2), hi = std::min(fmax, sampleRate / 2);
  const double bw = hi - lo, fc = lo + bw / 2;
  const int c = N / 2;
  for (int i = 0; i < N; i++) {
    std::complex<Scalar> v;
    if (c == i) v = M_PI * (bw / sampleRate);
    else v = std::sin(M_PI * (bw / sampleRate) * (i - c)) / (i - c);
    v *= std::exp(std::complex<Scalar>(0.0, (2 * M_PI * fc * i) / sampleRate));
    v *= (0.42 - 0.5 * cos((2 * M_PI * i) / N) + 0.08 * cos((4 * M_PI * i) / N));
    h[2 * i] = v.real();
    h[2 * i + 1] = v.imag();
  }
}

/** DFT in double of any length (host only; sign -1 = forward): Cooley-Tukey over the smallest prime factor, the factor's
 * own small DFTs written out as sums; a prime length is a plain sum. */
inline void dft(std::vector< std::complex<double> > &a, int sign) {
  const size_t n = a.size();
  if (n < 2) return;
  size_t p = n;
  for (size_t q = 2; q * q <= n; q++) if (n % q == 0) { p = q; break; }
  if (p == n && n > 64) {   // a large prime (FilterNode(1009): 2018 = 2 x 1009): the chirp transform over radix-2 transforms
    size_t M = 1; while (M < 2 * n - 1) M <<= 1;
    struct R2 { static void run(std::vector< std::complex<long double> > &v, int sg) {
      const size_t L = v.size();
      for (size_t i = 1, j = 0; i < L; i++) { size_t bit = L >> 1; for (; j & bit; bit >>= 1) j ^= bit; j ^= bit; if (i < j) std::swap(v[i], v[j]); }
      for (size_t len = 2; len <= L; len <<= 1)
        for (size_t k = 0; k < len / 2; k++) {
          const long double ang = sg * 2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)len;
          const std::complex<long double> wk(cosl(ang), sinl(ang));
          for (size_t q = 0; q < L; q += len) { const std::complex<long double> u = v[q + k], t = wk * v[q + k + len / 2]; v[q + k] = u + t; v[q + k + len / 2] = u - t; }
        } } };
    std::vector< std::complex<long double> > w(n), A(M), B(M);
    for (size_t j = 0; j < n; j++) {
      const long double ang = sign * 3.14159265358979323846264338327950288L * (long double)((j * j) % (2 * n)) / (long double)n;
      w[j] = std::complex<long double>(cosl(ang), sinl(ang));
      A[j] = std::complex<long double>(a[j].real(), a[j].imag()) * w[j];
      B[j] = std::conj(w[j]);
      if (j) B[M - j] = std::conj(w[j]);
    }
    R2::run(A, -1); R2::run(B, -1);
    for (size_t k = 0; k < M; k++) A[k] *= B[k];
    R2::run(A, +1);
    for (size_t k = 0; k < n; k++) { const std::complex<long double> v = A[k] / (long double)M * w[k]; a[k] = std::complex<double>((double)v.real(), (double)v.imag()); }
    return;
  }
  const size_t m = n / p;
  std::vector< std::vector< std::complex<double> > > part(p);
  if (m > 1)
    for (size_t r = 0; r < p; r++) {
      part[r].resize(m);
      for (size_t i = 0; i < m; i++) part[r][i] = a[i * p + r];
      dft(part[r], sign);
    }
  std::vector< std::complex<double> > out(n);
  for (size_t k = 0; k < n; k++) {
    std::complex<double> acc(0, 0);
    for (size_t r = 0; r < p; r++) {
      const double ang = sign * 2.0 * M_PI * double((r * k) % n) / double(n);
      acc += (m > 1 ? part[r][k % m] : a[r]) * std::complex<double>(std::cos(ang), std::sin(ang));
    }
    out[k] = acc;
  }
  a.swap(out);
}

/** Spectrum the FilterSource multiplies with: K = DFT_2N([h, 0]) / ||K||_2, 2N complex Scalars
 * (src/filternode.hh:197-202; the norm accumulates real(conj(k) k) in double, src/buffer.hh:182-188).
 * The reference computes this DFT with FFTW3 (un-vendored): any correct DFT is within rounding of it. */
template <class Scalar>
inline void fftFilterSpectrum(int N, const Scalar *h, Scalar *K) {
  std::vector< std::complex<double> > a(2 * size_t(N));
  for (int i = 0; i < N; i++) a[i] = std::complex<double>(h[2 * i], h[2 * i + 1]);
  dft(a, -1);
  double e = 0;
  for (size_t i = 0; i < a.size(); i++) {
    K[2 * i] = Scalar(a[i].real());
    K[2 * i + 1] = Scalar(a[i].imag());
    e += K[2 * i] * K[2 * i] + K[2 * i + 1] * K[2 * i + 1];
  }
  const Scalar d = Scalar(std::sqrt(e));
  for (size_t i = 0; i < 2 * a.size(); i++) K[i] /= d;
}

}  // namespace design
}  // namespace gpu
}  // namespace sdr

#endif
