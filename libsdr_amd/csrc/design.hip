// design.hip — exports the host-side designers of include/sdr/gpu/design.hh through the C ABI so
// that every binding (the C++ nodes, the ctypes tests, bench.py) gets bit-identical taps.
#include "sdrhip_internal.hpp"
#include "sdr/gpu/design.hh"

using namespace sdrhip;
namespace dz = sdr::gpu::design;

extern "C" {

int sdrhip_design_iqbb_taps(double filter_freq, double width, double sample_rate, int order, int32_t *taps) {
  return guarded([&] {
    SDRHIP_REQUIRE(taps && order >= 1 && sample_rate != 0, SDRHIP_E_INVALID, "bad argument");
    dz::iqbbTaps(filter_freq, width, sample_rate, (size_t)order, taps);
  });
}

int sdrhip_design_bb_taps(double filter_freq, double width, double sample_rate, int order, int32_t *taps) {
  return guarded([&] {
    SDRHIP_REQUIRE(taps && order >= 1 && sample_rate != 0, SDRHIP_E_INVALID, "bad argument");
    dz::bbTaps(filter_freq, width, sample_rate, (size_t)order, taps);
  });
}

int sdrhip_design_iqbb_decim(double sample_rate, int sub_sample, double out_rate, int *decim) {
  return guarded([&] {
    SDRHIP_REQUIRE(decim && sub_sample >= 1, SDRHIP_E_INVALID, "bad argument");
    *decim = (int)dz::iqbbDecimation(sample_rate, (size_t)sub_sample, out_rate);
  });
}

int sdrhip_design_freqshift_lut_i16(int32_t *lut) {
  return guarded([&] {
    SDRHIP_REQUIRE(lut, SDRHIP_E_INVALID, "lut is NULL");
    dz::freqShiftLutI16(lut);
  });
}

int sdrhip_design_freqshift_lut_i8(int32_t *lut) {
  return guarded([&] {
    SDRHIP_REQUIRE(lut, SDRHIP_E_INVALID, "lut is NULL");
    dz::freqShiftLutI8(lut);
  });
}

int sdrhip_design_freqshift_inc(double shift, double sample_rate, uint32_t *inc) {
  return guarded([&] {
    SDRHIP_REQUIRE(inc && sample_rate != 0, SDRHIP_E_INVALID, "bad argument");
    *inc = dz::freqShiftIncrement(shift, sample_rate);
  });
}

int sdrhip_design_fir_lowpass(int order, double upper_freq, double sample_rate, double *alpha) {
  return guarded([&] {
    SDRHIP_REQUIRE(alpha && order >= 1 && sample_rate != 0, SDRHIP_E_INVALID, "bad argument");
    dz::firLowPass((size_t)order, upper_freq, sample_rate, alpha);
  });
}

int sdrhip_design_fmdeemph_alpha(double sample_rate, int *alpha) {
  return guarded([&] {
    SDRHIP_REQUIRE(alpha && sample_rate > 0, SDRHIP_E_INVALID, "bad argument");
    *alpha = dz::fmDeemphAlpha(sample_rate);
  });
}

int sdrhip_design_fftfilt_kernel(int n, double fmin, double fmax, double sample_rate, float *h) {
  return guarded([&] {
    SDRHIP_REQUIRE(h && n >= 2 && sample_rate != 0, SDRHIP_E_INVALID, "bad argument");
    dz::fftFilterKernel(n, fmin, fmax, sample_rate, h);
  });
}

int sdrhip_design_fftfilt_spectrum(int n, const float *h, float *spectrum) {
  return guarded([&] {
    SDRHIP_REQUIRE(h && spectrum && n >= 1, SDRHIP_E_INVALID, "bad argument");
    dz::fftFilterSpectrum(n, h, spectrum);
  });
}

int sdrhip_design_fftfilt_kernel_f64(int n, double fmin, double fmax, double sample_rate, double *h) {
  return guarded([&] {
    SDRHIP_REQUIRE(h && n >= 2 && sample_rate != 0, SDRHIP_E_INVALID, "bad argument");
    dz::fftFilterKernel(n, fmin, fmax, sample_rate, h);
  });
}

int sdrhip_design_fftfilt_spectrum_f64(int n, const double *h, double *spectrum) {
  return guarded([&] {
    SDRHIP_REQUIRE(h && spectrum && n >= 1, SDRHIP_E_INVALID, "bad argument");
    dz::fftFilterSpectrum(n, h, spectrum);
  });
}

}  // extern "C"
