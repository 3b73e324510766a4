// fbb_f32.hip — float baseband for BASELINE config 2 (build-defined):
//   y = SubSample_D( FIRLowPass_cf32( x[n] * exp(-2 pi i Fc n / Fs) ) )
// The reference has NO float baseband: IQBaseBand<float> does not compile (src/baseband.hh:29,205)
// and FreqShift<float> truncates samples to int16 (src/utils.hh:497-504, src/operators.hh:40-42;
// SURVEY fact 6). The reference-pinned sub-steps are the cf32 FIR (src/firfilter.hh:231-247) and
// SubSample<cf32> (src/subsample.hh:92-101), both provided by fir.hip (K3); the frequency shift is
// defined here: the phasor is a closed form of the absolute sample index evaluated in float64
// (no recurrence drift), the product is rounded to float once. PARITY UNPINNED for the shift.
#include "sdrhip_internal.hpp"

using namespace sdrhip;

namespace {

constexpr int TPB = 256;

__global__ __launch_bounds__(TPB) void freqshift_cf32_kernel(const float2 *in, long in_stride, float2 *out, long out_stride,
                                                            int N, unsigned long long n0, double fc, double fs) {
  const int c = blockIdx.y;
  for (int i = blockIdx.x * TPB + threadIdx.x; i < N; i += gridDim.x * TPB) {
    const float2 x = in[(long)c * in_stride + i];
    // same expression as the float64 closed form it is checked against (oracle orc_freqshift_cf32)
    const double turns = fmod(__ddiv_rn(__dmul_rn(fc, (double)(n0 + (unsigned long long)i)), fs), 1.0);
    const double ph = __dmul_rn(-2.0 * M_PI, turns);
    double s, co;
    sincos(ph, &s, &co);
    const double xr = x.x, xi = x.y;
    out[(long)c * out_stride + i] = make_float2((float)__dsub_rn(__dmul_rn(xr, co), __dmul_rn(xi, s)),
                                                (float)__dadd_rn(__dmul_rn(xr, s), __dmul_rn(xi, co)));
  }
}

}  // namespace

struct sdrhip_fbb_f32 {
  sdrhip_ctx *ctx = nullptr;
  sdrhip_fir *fir = nullptr;
  double fc = 0, fs = 1;
  int C = 1;
  size_t max_in = 0;
  unsigned long long n0 = 0;
  DevBuf<float2> shifted;
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
      int rc = sdrhip_fir_create(ctx, SDRHIP_FIR_CF32, alpha, order, decim, channels, max_in, SDRHIP_EPI_NONE, &h->fir);
      if (rc != SDRHIP_OK) throw Failure{rc};
      h->shifted.alloc((size_t)channels * max_in);
      h->max_out = max_in / decim + 1;
    } catch (...) { if (h->fir) sdrhip_fir_destroy(h->fir); delete h; throw; }
    *out = h;
  });
}

int sdrhip_fbb_f32_out_count(sdrhip_fbb_f32 *h, size_t n_in, size_t *n_out) {
  if (!h) { set_error("handle is NULL"); return SDRHIP_E_INVALID; }
  return sdrhip_fir_out_count(h->fir, n_in, n_out);
}

int sdrhip_fbb_f32_process_dev(sdrhip_fbb_f32 *h, const float *in_dev, size_t n_in, size_t in_stride, float *out_dev,
                               size_t out_stride, size_t *n_out) {
  return guarded([&] {
    SDRHIP_REQUIRE(h, SDRHIP_E_INVALID, "handle is NULL");
    SDRHIP_REQUIRE(n_in <= h->max_in, SDRHIP_E_SIZE, "n_in %zu > max_in %zu", n_in, h->max_in);
    if (n_in == 0) { if (n_out) *n_out = 0; return; }
    SDRHIP_REQUIRE(in_dev && out_dev, SDRHIP_E_INVALID, "NULL buffer");
    h->ctx->use();
    if (in_stride == 0) in_stride = n_in;
    const unsigned bx = (unsigned)std::min<size_t>(ceil_div(n_in, (size_t)TPB), 4096);
    hipLaunchKernelGGL(freqshift_cf32_kernel, dim3(bx, h->C), dim3(TPB), 0, h->ctx->stream,
                       reinterpret_cast<const float2 *>(in_dev), (long)in_stride, h->shifted.p, (long)n_in, (int)n_in,
                       h->n0, h->fc, h->fs);
    SDRHIP_CHECK_HIP(hipGetLastError());
    int rc = sdrhip_fir_process_dev(h->fir, h->shifted.p, n_in, n_in, out_dev, out_stride, n_out);
    if (rc != SDRHIP_OK) throw Failure{rc};
    h->n0 += n_in;
  });
}

int sdrhip_fbb_f32_process(sdrhip_fbb_f32 *h, const float *in_host, size_t n_in, size_t in_stride, float *out_host,
                           size_t out_stride, size_t *n_out) {
  return guarded([&] {
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
