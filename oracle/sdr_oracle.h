/* sdr_oracle.h — CPU restatement of the libsdr hot path (TEST INFRASTRUCTURE ONLY).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * there only as the checker / baseline.  The product (libsdr_amd/, include/) never links it.
 *
 * Pinning: every function below is checked against golden vectors cut from the compiled,
 * unmodified reference (oracle/ref_driver.cc -> tests/golden/, see tests/test_oracle_golden.py).
 * EXCEPTION: the FFT-convolution functions (orc_fftfilt_*): the reference delegates the DFT to
 * FFTW3 (src/fftplan_fftw3.hh:34-36,64), an un-vendored, un-pinned, un-installed dependency, so
 * that path is "PARITY UNPINNED at the FFTW boundary"; only its time-domain kernel design
 * (sinc_flt_kernel, src/filternode.hh:18-28) is pinned by fixture g7_*.
 *
 * All interleaved complex data: (re, im) pairs.  cs16 = 2 x int16, cf32 = 2 x float.
 */
#ifndef SDR_ORACLE_H
#define SDR_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- host-side designers ------------------------------------------------------------------ */
/* src/baseband.hh:239-262 (+ :266-272 int32 storage of Fc/Ff/Fs/width). taps: order x (re,im) */
void orc_iqbb_design(double Ff, double width, double Fs, int order, int32_t *taps);
/* src/baseband.hh:159-162 */
int orc_iqbb_decim(double Fs, int sub_sample, double oFs);
/* src/freqshift.hh:31-35 with Traits<int16_t>::shift == 16 (src/traits.cc:22). lut: 128 x (re,im) */
void orc_freqshift_lut_i16(int32_t *lut);
/* src/freqshift.hh:78-87 */
uint32_t orc_freqshift_inc(double F, double Fs);
/* src/firfilter.hh:16-32 */
void orc_fir_lowpass_design(int N, double Fu, double Fs, double *alpha);
/* src/filternode.hh:18-28,186-196: h[0..N) complex float (time domain) */
void orc_fftfilt_design_h(int N, double fmin, double fmax, double Fs, float *h);
/* src/filternode.hh:197-202: K = DFT_2N(h zero-padded) / ||K||_2 ; K: 2N x (re,im) float */
void orc_fftfilt_design_K(int N, const float *h, float *K);

/* ---- generators (src/siggen.hh:116-131) ---------------------------------------------------- */
void *orc_iqsiggen_create(double Fs);
void orc_iqsiggen_add_sine(void *g, double f, double a, double phi);
void orc_iqsiggen_next_cs16(void *g, size_t n, int16_t *out);
void orc_iqsiggen_next_cf32(void *g, size_t n, float *out);
void orc_iqsiggen_destroy(void *g);

/* ---- IQBaseBand<int16_t> (src/baseband.hh:198-236, src/freqshift.hh:58-74) ---------------- */
void *orc_iqbb_i16_create(const int32_t *taps, int order, const int32_t *lut, uint32_t lut_inc,
                          int negative, int decim);
/* returns number of complex outputs; in/out may alias (reference runs in place) */
size_t orc_iqbb_i16_process(void *h, const int16_t *in, size_t n, int16_t *out);
/* IQBaseBand<int8_t> -> FMDemod<int8_t,int16_t> (the documentation example's chain, src/sdr.hh:225-240) */
void orc_freqshift_lut_i8(int32_t *lut);
void *orc_iqbb_i8_create(const int32_t *taps, int order, const int32_t *lut, uint32_t lut_inc, int negative, int decim);
size_t orc_iqbb_i8_process(void *h, const int8_t *in, size_t n, int8_t *out);
void orc_iqbb_i8_destroy(void *h);
void orc_fm_i8(const int8_t *in, size_t n, int16_t *out, int16_t *last);
void orc_iqbb_i16_reset(void *h);   /* what _reconfigure does: counters, NOT the ring */
void orc_iqbb_i16_set_decim(void *h, int decim);   /* setSubsample: the new decimation (follow with reset = _reconfigure) */
void orc_iqbb_i16_set_order(void *h, const int32_t *taps, int order);   /* setOrder: new kernel + new (zeroed) ring, counters go on */
void orc_iqbb_i16_set_taps(void *h, const int32_t *taps);               /* setFilterFrequency / setFilterWidth: kernel only */
void orc_iqbb_i16_set_shift(void *h, uint32_t lut_inc, int negative);   /* setCenterFrequency: increment, sign, LUT phase = 0 */
void orc_iqbb_i16_destroy(void *h);

/* ---- FIRFilter<complex<int16>> / <complex<float>> (src/firfilter.hh:231-247) --------------- */
void *orc_fir_create(const double *alpha, int order);
void orc_fir_cs16_process(void *h, const int16_t *in, size_t n, int16_t *out);
void orc_fir_cf32_process(void *h, const float *in, size_t n, float *out);
void orc_fir_reset(void *h);        /* ring zeroed as in config() (:193-195) */
void orc_fir_destroy(void *h);

/* ---- demodulators (src/demod.hh, src/math.hh) ---------------------------------------------- */
int16_t orc_fast_atan2_i16(int16_t a, int16_t b);            /* src/math.hh:31-40 */
/* FMDemod<int16,int16>::_process (src/demod.hh:242-254): writes out[1..n-1], never out[0];
 * *last is _last_value (carried across buffers). in/out may alias (in-place layout). */
void orc_fm_i16(const int16_t *in, size_t n, int16_t *out, int16_t *last);
void orc_am_i16(const int16_t *in, size_t n, int16_t *out);  /* src/demod.hh:73-76 */
void orc_am_f32(const float *in, size_t n, float *out);
void orc_usb_i16(const int16_t *in, size_t n, int16_t *out); /* src/demod.hh:156-161 */
void orc_usb_f32(const float *in, size_t n, float *out);

/* ---- SubSample (src/subsample.hh:92-101) --------------------------------------------------- */
void *orc_subsample_create(size_t n);
size_t orc_subsample_cs16_process(void *h, const int16_t *in, size_t n, int16_t *out);
size_t orc_subsample_cf32_process(void *h, const float *in, size_t n, float *out);
void orc_subsample_destroy(void *h);

/* ---- FFT convolution: FilterSink + FilterSource (src/filternode.hh:81-88,164-181) ---------- */
/* PARITY UNPINNED at the FFTW boundary (see header). DFTs are evaluated in double and rounded
 * to float where FFTW3f would store floats. */
void *orc_fftfilt_create(int N, const float *K);
void orc_fftfilt_process(void *h, const float *in /*N*/, float *out /*N*/);
void orc_fftfilt_destroy(void *h);
/* plain double-precision DFT helper (sign = -1 forward, +1 backward, unnormalised), n = 2^k */
void orc_dft_f64(int n, int sign, const double *in, double *out);   /* any n */
/* FilterSource<double> (the filter classes are templates over Scalar): designer, spectrum, one overlap-add block */
void orc_fftfilt_design_h_f64(int N, double fmin, double fmax, double Fs, double *h);
void orc_fftfilt_design_K_f64(int N, const double *h, double *K);
void orc_fftfilt_process_f64(int N, const double *K, double *last /*N cf64 state*/, const double *in /*N*/, double *out /*N*/);

/* ---- build-defined float baseband (NO reference node exists) ----------------- */
/* y = x[n] * exp(-2*pi*i*Fc*n/Fs), phasor in float64 closed form per absolute index, result
 * rounded to float.  PARITY UNPINNED (there is nothing in the reference to pin it to). */
void orc_freqshift_cf32(const float *in, size_t n, uint64_t n0, double Fc, double Fs, float *out);

/* ---- "next" rows ------------------------------------------------------------------ */
/* AutoCast< complex<int16> > fed complex<uint8> (src/autocast.hh:62,187-194): every byte is read as int8,
 * (int16(b) - 127) << 8 wrapped to int16. n_bytes = 2 x samples. */
void orc_autocast_cu8_cs16(const uint8_t *in, size_t n_bytes, int16_t *out);
/* "next" row 3: BaseBand<int16_t> real-input variant (src/baseband.hh:305-529) */
void orc_bb_design(double Ff, double width, double Fs, int order, int32_t *taps);
void *orc_bb_i16_create(const int32_t *taps, int order, const int32_t *lut, uint32_t lut_inc, int negative, int decim);
size_t orc_bb_i16_process(void *h, const int16_t *in /* n real */, size_t n, int16_t *out /* cs16 */);
void orc_bb_i16_reset(void *h);
int orc_bb_i16_seek(void *h, uint64_t abs_index);   /* test bench: decimator / LUT phase in front of sample g*D; ring kept */
void orc_bb_i16_set_shift(void *h, uint32_t lut_inc, int negative);   /* setFrequencyShift: increment, sign, LUT phase = 0 */
void orc_bb_i16_destroy(void *h);

/* FMDeemph<int16_t> (src/demod.hh:305-306 alpha, :342-351 recursion); *avg is the node's _avg */
int orc_fmdeemph_alpha(double sample_rate);
void orc_fmdeemph_i16(const int16_t *in, size_t n, int alpha, int16_t *avg, int16_t *out);

/* ---- throughput helper for bench.py cpu_baseline (kind "port") ------------------------------ */
/* runs IQBaseBand(127,/8)->FM on `nbuf` buffers of `n` samples; returns seconds */
double orc_bench_iqbb_fm(const int32_t *taps, int order, const int32_t *lut, uint32_t inc, int negative,
                         int decim, const int16_t *in, size_t n, size_t nbuf, long *checksum);

#ifdef __cplusplus
}
#endif
#endif
