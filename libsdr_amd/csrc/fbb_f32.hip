// fbb_f32.hip — float baseband for BASELINE config 2 (build-defined):
//   y = SubSample_D( FIRLowPass_cf32( x[n] * exp(-2 pi i Fc n / Fs) ) )
// The reference has NO float baseband: IQBaseBand<float> does not compile (src/baseband.hh:29,205)
// and FreqShift<float> truncates samples to int16 (src/utils.hh:497-504, src/operators.hh:40-42;
// SURVEY fact 6). The reference-pinned sub-steps are the cf32 FIR (src/firfilter.hh:231-247) and
// SubSample<cf32> (src/subsample.hh:92-101), both provided by fir.hip (K3); the frequency shift is
// defined by this build: the phasor is a closed form of the absolute sample index in float64
// (oracle orc_freqshift_cf32); it is applied inside the FIR kernel's staging loop (fir.hip,
// fir_cf32_rt_kernel: closed form per lane and tile, constant float64 rotation in between), so the
// whole chain is ONE launch with 8 B read + 1 B written per input sample. PARITY UNPINNED for the shift.
#include "sdrhip_internal.hpp"

using namespace sdrhip;

struct sdrhip_fbb_f32 {
  sdrhip_ctx *ctx = nullptr;
  sdrhip_fir *fir = nullptr;
  double fc = 0, fs = 1;
  int C = 1;
  size_t max_in = 0;
  unsigned long long n0 = 0;
  DevBuf<float2> stage_in, stage_out;
  size_t max_out = 0;
};

extern "C" {

int sdrhip_fbb_f32_create(sdrhip_ctx *ctx, double Fc, double Fs, const double *alpha, int order, int decim,
                          int channels, size_t max_in, sdrhip_fbb_f32 **out) {
  return guarded([&] {
    SDRHIP_REQUIRE(ctx && alpha && out, SDRHIP_E_INVALID, "NULL argument");
    *out = nullptr;
    SDRHIP_REQUIRE(Fs > 0, SDRHIP_E_INVALID, "sample rate must be positive");
    ctx->use();
    sdrhip_fbb_f32 *h = new sdrhip_fbb_f32;
    try {
      h->ctx = ctx; h->fc = Fc; h->fs = Fs; h->C = channels; h->max_in = max_in;
      int rc = fir_create_impl(ctx, SDRHIP_FIR_CF32, alpha, order, decim, channels, max_in, SDRHIP_EPI_NONE, false, &h->fir);   // (the shift rides in the time-domain kernel's staging)
      if (rc != SDRHIP_OK) throw Failure{rc};
      fir_set_shift(h->fir, Fc, Fs);   // the shift rides in the FIR's staging: one kernel, no intermediate buffer
      h->max_out = max_in / decim + 1;
    } catch (...) { if (h->fir) sdrhip_fir_destroy(h->fir); delete h; throw; }
    *out = h;
  });
}

int sdrhip_fbb_f32_out_count(sdrhip_fbb_f32 *h, size_t n_in, size_t *n_out) {
  if (!h) { set_error("handle is NULL"); return SDRHIP_E_INVALID; }
  return sdrhip_fir_out_count(h->fir, n_in, n_out);
}

int sdrhip_fbb_f32_kernel_names(sdrhip_fbb_f32 *h, size_t n_in, char *buf, size_t len) {
  if (!h) { set_error("handle is NULL"); return SDRHIP_E_INVALID; }
  return sdrhip_fir_kernel_names(h->fir, n_in, buf, len);
}

int sdrhip_fbb_f32_process_dev(sdrhip_fbb_f32 *h, const float *in_dev, size_t n_in, size_t in_stride, float *out_dev,
                               size_t out_stride, size_t *n_out) {
  return guarded([&] {
    Range roctx_range("sdrhip_fbb_f32_process_dev");
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    SDRHIP_REQUIRE(n_in <= h->max_in, SDRHIP_E_SIZE, "n_in %zu > max_in %zu", n_in, h->max_in);
    if (n_in == 0) { if (n_out) *n_out = 0; return; }
    SDRHIP_REQUIRE(in_dev && out_dev, SDRHIP_E_INVALID, "NULL buffer");
    h->ctx->use();
    if (in_stride == 0) in_stride = n_in;
    int rc = sdrhip_fir_process_dev(h->fir, in_dev, n_in, in_stride, out_dev, out_stride, n_out);
    if (rc != SDRHIP_OK) throw Failure{rc};
    h->n0 += n_in;
  });
}

int sdrhip_fbb_f32_process(sdrhip_fbb_f32 *h, const float *in_host, size_t n_in, size_t in_stride, float *out_host,
                           size_t out_stride, size_t *n_out) {
  return guarded([&] {
    Range roctx_range("sdrhip_fbb_f32_process");
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    SDRHIP_REQUIRE(n_in <= h->max_in, SDRHIP_E_SIZE, "n_in %zu > max_in %zu", n_in, h->max_in);
    if (n_in == 0) { if (n_out) *n_out = 0; return; }
    SDRHIP_REQUIRE(in_host && out_host, SDRHIP_E_INVALID, "NULL buffer");
    h->ctx->use();
    if (in_stride == 0) in_stride = n_in;
    size_t no = 0;
    int rc = sdrhip_fir_out_count(h->fir, n_in, &no);
    if (rc != SDRHIP_OK) throw Failure{rc};
    if (out_stride == 0) out_stride = no;
    SDRHIP_REQUIRE(out_stride >= no, SDRHIP_E_SIZE, "out_stride %zu < outputs %zu", out_stride, no);
    if (!h->stage_in.p) { h->stage_in.alloc((size_t)h->C * h->max_in); h->stage_out.alloc((size_t)h->C * h->max_out); }
    copy_h2d_rows(h->ctx, h->stage_in.p, n_in * 8, in_host, in_stride * 8, n_in * 8, h->C);
    size_t produced = 0;
    rc = sdrhip_fbb_f32_process_dev(h, reinterpret_cast<const float *>(h->stage_in.p), n_in, n_in,
                                    reinterpret_cast<float *>(h->stage_out.p), h->max_out, &produced);
    if (rc != SDRHIP_OK) throw Failure{rc};
    copy_d2h_rows(h->ctx, out_host, out_stride * 8, h->stage_out.p, h->max_out * 8, produced * 8, h->C);
    SDRHIP_CHECK_HIP(hipStreamSynchronize(h->ctx->stream));
    if (n_out) *n_out = produced;
  });
}

int sdrhip_fbb_f32_set_taps(sdrhip_fbb_f32 *h, const double *alpha) {
  return guarded([&] {
    SDRHIP_REQUIRE(h && alpha, SDRHIP_E_INVALID, "NULL argument");
    const int rc = sdrhip_fir_set_taps(h->fir, alpha);
    if (rc != SDRHIP_OK) throw Failure{rc};
  });
}

int sdrhip_fbb_f32_set_shift(sdrhip_fbb_f32 *h, double Fc) {
  return guarded([&] {
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    h->fc = Fc;
    fir_set_shift(h->fir, Fc, h->fs);   // the phasor restarts at the current sample; history and decimator go on
  });
}

int sdrhip_fbb_f32_reset(sdrhip_fbb_f32 *h) {
  return guarded([&] {
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    h->n0 = 0;
    int rc = sdrhip_fir_reset(h->fir);
    if (rc != SDRHIP_OK) throw Failure{rc};
  });
}

int sdrhip_fbb_f32_destroy(sdrhip_fbb_f32 *h) {
  return guarded([&] {
    if (!h) return;
    h->ctx->use();
    (void)hipStreamSynchronize(h->ctx->stream);
    if (h->fir) sdrhip_fir_destroy(h->fir);
    delete h;
  });
}

}  // extern "C"
