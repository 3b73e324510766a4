/* sdrhip.h — C ABI of the MI355X-native libsdr hot path (libsdrhip.so).
 *
 * The reference (hmatuschek/libsdr) has no FFI: its hot path sits behind the C++ virtual node
 * interface sdr::Sink<T>::config()/process() + sdr::Source::send() (reference src/node.hh:174-258).
 * This header is what a binding for that interface calls; include/sdr/gpu/ *.hh holds the
 * header-only C++ nodes (same class names, same config/ownership/allow_overwrite rules) that sit
 * on top of it, and INTEGRATION.md shows how they drop into an existing libsdr graph.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types. Every function returns an int:
 *     0 = SDRHIP_OK, negative = SDRHIP_E_*. Nothing throws. sdrhip_last_error() gives the text.
 *   - A context owns one HIP stream (its own, or one adopted from the caller); all handles created
 *     from it enqueue on that stream and are single-threaded, like a libsdr node (one Queue
 *     worker calls process(), reference src/queue.cc:95-106).
 *   - Complex samples are interleaved (re, im): cs16 = 2 x int16 (4 B), cf32 = 2 x float (8 B).
 *   - Batched ("channel bank") layout is channel-major: channel c starts at base + c*stride
 *     samples; every channel receives the same number of samples per call, because buffer
 *     boundaries are part of the numerical contract (FMDemod skips index 0 of every buffer,
 *     reference src/demod.hh:245; IQBaseBand's first window is D+1 long, src/baseband.hh:200).
 *   - *_process()     : host pointers in/out (H2D + kernel + D2H, synchronous on return); the output may
 *                       alias the input (the reference nodes run in place when allow_overwrite), because
 *                       the data is staged through separate device buffers.
 *     *_process_dev() : device pointers, asynchronous on the context stream. The output range must NOT
 *                       overlap the input range: the kernels are tile-parallel and would race silently;
 *                       an overlap is rejected with SDRHIP_E_INVALID.
 *   - Taps / LUT / FFT kernels are INPUTS, designed on the host (include/sdr/gpu/design.hh
 *     restates the reference designers); a 1-ulp libm difference must not change results.
 *   - There is no CPU fallback: without a HIP device every create() fails with SDRHIP_E_NODEVICE.
 */
#ifndef SDRHIP_H
#define SDRHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDRHIP_VERSION 100 /* 0.1.0 */

enum {
  SDRHIP_OK = 0,
  SDRHIP_E_INVALID = -1,     /* bad argument */
  SDRHIP_E_NODEVICE = -2,    /* no HIP device / device index out of range */
  SDRHIP_E_HIP = -3,         /* a HIP runtime call failed (text in sdrhip_last_error) */
  SDRHIP_E_NOMEM = -4,       /* device or host allocation failed */
  SDRHIP_E_UNSUPPORTED = -5, /* parameter outside what the kernels implement */
  SDRHIP_E_SIZE = -6         /* buffer larger than the plan's max_in / stride too small */
};

/* demodulator fused behind a filter node (reference src/demod.hh) */
enum {
  SDRHIP_EPI_NONE = 0, /* complex output, same sample type as the input */
  SDRHIP_EPI_FM = 1,   /* FMDemod<int16_t>   (src/demod.hh:242-254, src/math.hh:31-40), in-place
                          convention: out[0] of every call = in[0].real() */
  SDRHIP_EPI_AM = 2,   /* AMDemod<Scalar>    (src/demod.hh:73-76) */
  SDRHIP_EPI_USB = 3   /* USBDemod<Scalar>   (src/demod.hh:156-161) */
};

typedef struct sdrhip_ctx sdrhip_ctx;
typedef struct sdrhip_timer sdrhip_timer;
typedef struct sdrhip_iqbb_i16 sdrhip_iqbb_i16;
typedef struct sdrhip_fir sdrhip_fir;
typedef struct sdrhip_demod sdrhip_demod;
typedef struct sdrhip_subsample sdrhip_subsample;
typedef struct sdrhip_fftconv sdrhip_fftconv;
typedef struct sdrhip_fbb_f32 sdrhip_fbb_f32;

/* ---- library / context ------------------------------------------------------------------- */
int sdrhip_version(void);
const char *sdrhip_strerror(int code);
/* text of the last failure on this thread (ctx may be NULL for create-time failures) */
const char *sdrhip_last_error(void);
int sdrhip_device_count(int *count);

/* stream: a hipStream_t to adopt (e.g. torch.cuda.current_stream().cuda_stream) or NULL to create
 * a private non-blocking stream. */
int sdrhip_ctx_create(int device, void *stream, sdrhip_ctx **out);
int sdrhip_ctx_destroy(sdrhip_ctx *ctx);
int sdrhip_ctx_synchronize(sdrhip_ctx *ctx);
int sdrhip_ctx_device_name(sdrhip_ctx *ctx, char *buf, size_t len);

/* device memory helpers (tests and the C++ nodes; bench.py hands over torch tensors instead) */
int sdrhip_malloc(sdrhip_ctx *ctx, size_t bytes, void **dptr);
int sdrhip_free(sdrhip_ctx *ctx, void *dptr);
int sdrhip_memcpy_h2d(sdrhip_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes); /* sync */
int sdrhip_memcpy_d2h(sdrhip_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes); /* sync */
int sdrhip_memset(sdrhip_ctx *ctx, void *dst_dev, int value, size_t bytes);                /* async */

/* HIP-event timer on the context stream (bench.py's roofline leg) */
int sdrhip_timer_create(sdrhip_ctx *ctx, sdrhip_timer **out);
int sdrhip_timer_start(sdrhip_timer *t);
int sdrhip_timer_stop(sdrhip_timer *t);
int sdrhip_timer_elapsed_ms(sdrhip_timer *t, float *ms); /* synchronises on the stop event */
int sdrhip_timer_destroy(sdrhip_timer *t);

/* Stream-read microbenchmark (measurement aid for bench.py, SURVEY §8d "stream-read microbenchmark on the box"):
 * reads `bytes` of device memory `iters` times with 16-byte loads and reports GB/s. */
int sdrhip_bench_stream_read(sdrhip_ctx *ctx, const void *dev, size_t bytes, int iters, double *gb_per_s);

/* ---- host-side designers (no device work; same code as include/sdr/gpu/design.hh) ----------- */
/* IQBaseBand::_update_filter_kernel (reference src/baseband.hh:239-262): order x (re,im) Q14 */
int sdrhip_design_iqbb_taps(double filter_freq, double width, double sample_rate, int order, int32_t *taps);
/* BaseBand<int16_t>::_update_filter_kernel, the REAL-input node (src/baseband.hh:464-491): order x (re,im) Q16;
 * differs from the IQ designer: double-valued Fs/Ff/width, +Ff modulation, Blackman on (i+1)/(order+2),
 * centre tap of an even order = 1 */
int sdrhip_design_bb_taps(double filter_freq, double width, double sample_rate, int order, int32_t *taps);
/* IQBaseBand::_reconfigure (src/baseband.hh:159-162): explicit D, or floor(Fs/out_rate) >= 1 */
int sdrhip_design_iqbb_decim(double sample_rate, int sub_sample, double out_rate, int *decim);
/* FreqShiftBase<int16_t> ctor (src/freqshift.hh:31-35): 128 x (re,im) */
int sdrhip_design_freqshift_lut_i16(int32_t *lut);
/* FreqShiftBase::_update_lut_incr (src/freqshift.hh:78-87) */
int sdrhip_design_freqshift_inc(double shift, double sample_rate, uint32_t *inc);
/* FIRLowPassCoeffs::coeffs (src/firfilter.hh:16-32) */
int sdrhip_design_fir_lowpass(int order, double upper_freq, double sample_rate, double *alpha);
/* sinc_flt_kernel<float> (src/filternode.hh:18-28,186-196): N x (re,im) float */
int sdrhip_design_fftfilt_kernel(int n, double fmin, double fmax, double sample_rate, float *h);
/* FilterSource::_updateFilter (src/filternode.hh:197-202): 2N x (re,im) float spectrum */
int sdrhip_design_fftfilt_spectrum(int n, const float *h, float *spectrum);
/* the same for FilterSource<double> (the filter classes are templates over Scalar, src/filternode.hh:16-17,104-105) */
int sdrhip_design_fftfilt_kernel_f64(int n, double fmin, double fmax, double sample_rate, double *h);
int sdrhip_design_fftfilt_spectrum_f64(int n, const double *h, double *spectrum);

/* ---- K1: IQBaseBand<int16_t> (+ fused demodulator) ---------------------------------------- */
/* Replaces IQBaseBand<int16_t>::_process/_filter_ring (reference src/baseband.hh:198-236) and
 * FreqShiftBase<int16_t>::applyFrequencyShift (src/freqshift.hh:58-74), optionally followed by
 * FMDemod/AMDemod/USBDemod<int16_t> run in place on its output (src/demod.hh).
 *   taps     order x (re,im) int32, Q14, |component| <= 32767  (src/baseband.hh:239-262)
 *   lut      128 x (re,im) int32                               (src/freqshift.hh:31-35)
 *   lut_inc  phase increment in 1/256 LUT steps, 0 = no shift  (src/freqshift.hh:78-87)
 *   negative 1 if the frequency shift is negative              (src/freqshift.hh:65)
 *   decim    box-average decimation D >= 1                     (src/baseband.hh:212-219)
 * Output per channel and call: sdrhip_iqbb_i16_out_count() samples — cs16 for EPI_NONE, int16
 * otherwise. State (FIR history, decimator phase and partial sum, LUT phase, FM last angle)
 * persists across calls exactly as the reference node's members do. */
int sdrhip_iqbb_i16_create(sdrhip_ctx *ctx, const int32_t *taps, int order, const int32_t *lut,
                           uint32_t lut_inc, int negative, int decim, int channels, size_t max_in,
                           int epilogue, sdrhip_iqbb_i16 **out);
/* which kernel formulation the plan selected: 0 = VALU v_dot2c_i32_i16 (any decim/order), 1 = int8-MFMA
 * block-Toeplitz GEMM on 32x32x32 tiles (decim 8, order <= 513; LDS-DMA fed, see kernel_names), 3 = the same matrix
 * part for any other decimation (order <= 257; 258 ... 513 from decimation 9 on), 4 = the real-input node on the matrix
 * cores (any decimation up to 512, order <= 273); all bit-exact. Orders 258 ... 513 (33 K steps), the int8 chain
 * (IQBaseBand<int8_t>, up to 129 taps, decimation 8 and 9 ... 512) and the real-input node at a decimation other than 8
 * exist as forms of the hot kernel only: their calls too short to hold a hot tile — and, with SDRHIP_IQBB_HOT=0, the whole
 * plan — run the VALU kernel. (2 was round 1's 16x16x64 shape, removed in round 4.) SDRHIP_IQBB_PATH=valu at create time
 * selects the VALU kernel for every plan (tests). */
int sdrhip_iqbb_i16_path(sdrhip_iqbb_i16 *h, int *path);
/* Names of the kernels a call of this plan launches, dominant one first, comma separated (measurement aid: what to
 * look for in a rocprofv3 kernel trace). Path 1 (decim 8, order <= 513, complex<int16>, complex<uint8> or complex<int8> input) runs
 * ONE launch per call, "iqbb_hot_kernel": a persistent grid over the wave slices that touch no border of the call,
 * whose workgroups finish with the call's first and last slices, the state and the history roll (calls of fewer than
 * 3 tiles, about 6000 samples, run the general kernel "iqbb_i16_mfma_dma_kernel" / "iqbb_i16_mfma_kernel" instead).
 * Path 3 plans of up to 257 taps and 9 <= decim <= 512, shifted or not (the reference's receivers: 16 taps / 83 or / 20
 * without a shift, 21 taps / 125 with one; examples/sdr_rec.cc:42-68, examples/sdr_fm.cc:40) run "iqbb_hot_anyd_kernel" on long calls — the same persistent
 * structure, cold slices included (with FM the slices' first outputs are completed at the end of that kernel where the
 * channels deal evenly over its grid, else by a second, tiny launch "iqbb_fm_fixup_kernel"); decimations 2 ... 7 run
 * "iqbb_hot_sd_kernel" (the small-decimation form: a slice's 73 ... 256 groups summed out of an LDS array; in workgroups
 * of 4, 8 or 16 waves by what fits beside the tap fragments); decimations 257 ... 464 and 513 ... 32768 run the same hot
 * kernel emitting partial box sums per 512-sample slice + "iqbb_bigd_finish_kernel" (a group spans slices; beyond 2048,
 * with FM 1024, there is no general kernel: every call takes that form, and plans the hot kernel cannot serve — real or
 * int8 input, SDRHIP_IQBB_HOT=0 — are SDRHIP_E_UNSUPPORTED); other path 3 plans and short calls the general kernel "iqbb_i16_mfmag_kernel".
 * The real-input node (path 4) runs "iqbb_hot_kernel" at decimation 8 and the same any-D / small-decimation forms at every other
 * decimation up to 512 (round 6: the real-input tile hands every lane 8 consecutive samples, as the any-D forms' permuted
 * complex tile does — one epilogue for both).
 * Tuning / test variables, all read at create time: SDRHIP_IQBB_HOT=0 (general kernels only), SDRHIP_IQBB_TPW (tiles per
 * work unit), SDRHIP_IQBB_WGPCU (workgroups per CU of the persistent grid), SDRHIP_IQBB_FM_RESIDENT=0|1 (FM at a
 * decimation other than 8: never / always whole channels as work units, i.e. the slices' first angle differences by a
 * second launch / inside the hot kernel; unset: by whether the channels deal evenly over the grid), SDRHIP_IQBB_BIGD_MIN=n
 * (exactly the decimations >= n, n >= 257, take the large-decimation form). None changes results. */
int sdrhip_iqbb_i16_kernel_names(sdrhip_iqbb_i16 *h, char *buf, size_t len);
/* The plan the handle runs, for measurement tools (bench.py prices the matrix work with it): info[0] = path (0 VALU kernel,
 * 1 int8-MFMA at decimation 8, 3 int8-MFMA at any decimation, 4 real input on the matrix cores), info[1] = K steps S of 32
 * plane bytes per 512-sample slice, info[2], info[3] = first step and number of steps that also multiply the taps' high byte
 * plane, info[4] = waves per workgroup, info[5] = input kind (0 complex<int16>, 1 complex<uint8>, 2 real int16), info[6] =
 * padded filter length, info[7] = history samples kept per channel. n >= 8; with n >= 9 also info[8] = the buffer boundaries the
 * last one-launch sdrhip_iqbb_i16_process_dev_multi call left to its fix-up launch (0: the hot kernel wrote them all itself; -1: no
 * such call yet). */
int sdrhip_iqbb_i16_plan_info(sdrhip_iqbb_i16 *h, int *info, int n);
/* outputs the next call of n_in samples will produce (does not advance the state) */
int sdrhip_iqbb_i16_out_count(sdrhip_iqbb_i16 *h, size_t n_in, size_t *n_out);
/* in: channels x n_in cs16 (row stride in_stride samples); out: channels rows of out_stride
 * elements. Strides are in elements of the respective type; 0 means "tightly packed". */
int sdrhip_iqbb_i16_process(sdrhip_iqbb_i16 *h, const int16_t *in_host, size_t n_in, size_t in_stride,
                            void *out_host, size_t out_stride, size_t *n_out);
int sdrhip_iqbb_i16_process_dev(sdrhip_iqbb_i16 *h, const int16_t *in_dev, size_t n_in, size_t in_stride,
                                void *out_dev, size_t out_stride, size_t *n_out);
/* n_buffers consecutive reference-sized buffers of n_per_buffer samples per channel (row c of `in` holds them back to back,
 * n_buffers * n_per_buffer <= max_in) in ONE launch, with the buffer boundaries kept: the outputs are exactly those of
 * n_buffers successive sdrhip_iqbb_i16_process_dev calls written one behind the other into row c of `out` — the baseband's
 * state runs on across buffers anyway (src/baseband.hh:198-219); the fused FMDemod starts every buffer anew (index 0 of a
 * buffer's output is the in-place value, index 1 takes the previous buffer's last angle: src/demod.hh:242-254, SURVEY §7
 * "the batched API must carry buffer boundaries"). One launch amortises the grid's ramp, the call's border slices and the
 * state hand-over over n_buffers buffers. FM at decimation 8: the hot kernel writes the per-buffer values itself wherever a
 * boundary and the output behind it fall into one hot slice (equal buffers: always, unless the boundary sits on a slice's last
 * lane); the other boundaries, and every other plan's, are patched by a second, tiny launch (iqbb_fm_multi_fixup_kernel). n_out_per_buffer (n_buffers entries) and n_out_total may be NULL. */
int sdrhip_iqbb_i16_process_dev_multi(sdrhip_iqbb_i16 *h, const int16_t *in_dev, size_t n_buffers, size_t n_per_buffer, size_t in_stride,
                                      void *out_dev, size_t out_stride, size_t *n_out_per_buffer, size_t *n_out_total);
/* Input sample format ("next" row). SDRHIP_IN_CU8: the buffers hold complex<uint8_t> (2 B per
 * sample, e.g. RTL-SDR) and AutoCast< complex<int16_t> > (reference src/autocast.hh:62,187-194: each byte read
 * as int8, (int16(b)-127)<<8) is applied while loading, as examples/sdr_fm.cc:49-50 chains the two nodes.
 * Strides stay in samples. Allowed before the first buffer or right after a reset. */
enum { SDRHIP_IN_CS16 = 0, SDRHIP_IN_CU8 = 1 };
int sdrhip_iqbb_i16_set_input_format(sdrhip_iqbb_i16 *h, int format);
/* "next" row: BaseBand<int16_t>, the real-input node (reference src/baseband.hh:305-529:
 * _process :425-445, _filter_ring :448-460). Same handle type and the same process / out_count / reset /
 * destroy calls as above, with these differences: the input rows hold n_in REAL int16 samples (2 B per sample,
 * strides in samples); taps are Q16 (sdrhip_design_bb_taps; any |component| < 2^23); the FIR result is shifted by
 * Traits<int16_t>::shift = 16; decimation windows are the D samples {gD .. gD+D-1} from the first sample on
 * (no D+1 first window). Output: cs16 (or the demodulated int16 with an epilogue).
 * Kernels: for decim == 8, order <= 273 and taps within two byte planes (|component| < 2^15 - 128: every filter but
 * very short ones) the int8-MFMA formulation over the real sample stream (iqbb_hot_kernel's real-input instantiation;
 * calls shorter than 3 tiles: bb_real_mfma_kernel), else the VALU kernel; all bit-exact,
 * sdrhip_iqbb_i16_kernel_names says which one a plan runs. */
int sdrhip_bb_i16_create(sdrhip_ctx *ctx, const int32_t *taps, int order, const int32_t *lut,
                         uint32_t lut_inc, int negative, int decim, int channels, size_t max_in,
                         int epilogue, sdrhip_iqbb_i16 **out);
/* "next" row: IQBaseBand<int8_t>, the baseband of the reference's documentation example
 * (src/sdr.hh:225-240: IQBaseBand<int8_t> -> FMDemod<int8_t,int16_t>). Same handle type and calls. Input rows hold
 * complex<int8_t> (2 B per sample). The filter, window sum and division are the int16 node's (the class computes in
 * int32 for every Scalar, src/baseband.hh:28-31); the frequency shift it inherits from FreqShiftBase<int8_t> computes
 * in complex<int16_t>: the FIR value wraps to int16 at the call, `lut` is sdrhip_design_freqshift_lut_i8 (2^8 scale),
 * the product wraps to int16 and is shifted by 8 (src/freqshift.hh:18-22,58-74, src/traits.cc:11). Output:
 * complex<int8_t> (2 B; EPI_NONE) or, with EPI_FM, FMDemod<int8_t,int16_t>'s int16 run in place (out[0] of a call =
 * the two bytes of its first complex<int8_t> output). VALU kernel. */
int sdrhip_design_freqshift_lut_i8(int32_t *lut);
int sdrhip_iqbb_i8_create(sdrhip_ctx *ctx, const int32_t *taps, int order, const int32_t *lut, uint32_t lut_inc,
                          int negative, int decim, int channels, size_t max_in, int epilogue, sdrhip_iqbb_i16 **out);
/* Mid-stream retuning with the reference's semantics (src/baseband.hh:82-112), for plans of either create call:
 *   set_taps   setFilterFrequency / setFilterWidth -> _update_filter_kernel(): only the kernel changes (same order);
 *              FIR history, decimator phase and partial sum, LUT phase and FM angle go on as they are.
 *              SDRHIP_E_UNSUPPORTED when the new taps cannot run on the plan's kernel formulation (create a new plan).
 *   set_shift  setCenterFrequency -> setFrequencyShift -> _update_lut_incr() (src/freqshift.hh:52-54,78-87): new
 *              increment and sign, and the LUT phase counter restarts at 0 with the next sample; nothing else changes.
 * setSubsample / setOutputSampleRate / config() run _reconfigure: set_taps + set_shift + reset(keep_history = 1). */
int sdrhip_iqbb_i16_set_taps(sdrhip_iqbb_i16 *h, const int32_t *taps);
int sdrhip_iqbb_i16_set_shift(sdrhip_iqbb_i16 *h, uint32_t lut_inc, int negative);
/* keep_history = 1: what IQBaseBand::_reconfigure does (counters and phases reset, FIR ring kept,
 * src/baseband.hh:175-177); 0: a freshly constructed node (ring zeroed, :41-43). With a fused FMDemod: | 2 also
 * keeps the demodulator's last angle — the FMDemod node behind a reconfigured baseband is only reset when the Config
 * it receives changes (src/node.cc:98-105, src/demod.hh:210); 1 alone resets it (a changed output Config). */
int sdrhip_iqbb_i16_reset(sdrhip_iqbb_i16 *h, int keep_history);
/* Streaming state carried from one plan into a FRESH one (same channels, same sample kind, same device) — what the
 * reference node keeps when a setter changes what a device plan is made for (decimation, buffer size, order, fused
 * demodulator), so that a new plan is no visible event:
 *   SDRHIP_KEEP_RING      the FIR ring (equal orders). Alone: as IQBaseBand::_reconfigure leaves it — setSubsample /
 *                         setOutputSampleRate / config(), src/baseband.hh:106-112,115-132,156-194: counters reset, ring
 *                         contents kept where they lie, i.e. read ROTATED afterwards (as reset(keep_history = 1)).
 *   SDRHIP_KEEP_FM        the fused FMDemod's last angle (the node behind the baseband is reset only when the Config it
 *                         receives changes, src/node.cc:98-105).
 *   SDRHIP_KEEP_COUNTERS  the stream goes on (equal decimations): absolute sample index, open decimator window and its
 *                         partial sum, LUT phase. With KEEP_RING the history is copied in time order (no rotation). This
 *                         is IQBaseBand::setOrder (src/baseband.hh:69-79: new kernel, NEW ring, nothing else touched)
 *                         without KEEP_RING — the reference's new ring is uninitialised memory, a fresh plan's is zeros.
 * `from` is only read; destroy it afterwards. */
enum { SDRHIP_KEEP_RING = 1, SDRHIP_KEEP_FM = 2, SDRHIP_KEEP_COUNTERS = 4 };
int sdrhip_iqbb_i16_adopt_state(sdrhip_iqbb_i16 *h, sdrhip_iqbb_i16 *from, int what);
int sdrhip_iqbb_i16_destroy(sdrhip_iqbb_i16 *h);

/* ---- K2/K3: FIRFilter<complex<int16_t>> (exact) and FIRFilter<complex<float>> ------------- */
enum {
  SDRHIP_FIR_CS16_EXACT = 0, /* fp64, one truncation per tap, taps walked in order: bit-exact
                                (reference src/firfilter.hh:237-243, src/operators.hh:24-26) */
  SDRHIP_FIR_CF32 = 1        /* complex<float>; tolerance path (<= 1e-5 rel), may reassociate */
};
/* alpha: `order` doubles (src/firfilter.hh:16-32). decim: SubSample<Scalar>(decim) fused behind
 * the filter (src/subsample.hh:92-101), 1 = none. epilogue as above (FM only for CS16). */
int sdrhip_fir_create(sdrhip_ctx *ctx, int kind, const double *alpha, int order, int decim, int channels,
                      size_t max_in, int epilogue, sdrhip_fir **out);
int sdrhip_fir_out_count(sdrhip_fir *h, size_t n_in, size_t *n_out);
/* The kernel a call of n_in samples per channel runs (0: the plan's max_in), for profiles and tests:
 * "fir_cs16_exact_kernel"; complex<float>: "fir_cf32_pipe_kernel" (decimation 8 and enough tiles to give every
 * workgroup several: consecutive tiles software-pipelined in one workgroup) or "fir_cf32_rt_kernel". */
int sdrhip_fir_kernel_names(sdrhip_fir *h, size_t n_in, char *buf, size_t len);
int sdrhip_fir_process(sdrhip_fir *h, const void *in_host, size_t n_in, size_t in_stride, void *out_host,
                       size_t out_stride, size_t *n_out);
int sdrhip_fir_process_dev(sdrhip_fir *h, const void *in_dev, size_t n_in, size_t in_stride, void *out_dev,
                           size_t out_stride, size_t *n_out);
/* A complex<float> plan with decim = 1 and no epilogue (FIRLowPass<complex<float>>) runs as overlap-save FFT convolution on the
 * tuned FFT kernels behind this handle (transform size as for sdrhip_fftconv_create's converted plans, see there) — a tolerance
 * path either way (<= 1e-5), 6x faster at 127 taps and 145x at 4097 than
 * `order` multiply-adds per sample; SDRHIP_FIR_TIME_DOMAIN=1 in the environment keeps the time-domain kernel (tests). Below the
 * measured crossover — up to 32 taps on plans of at most 2^18 samples per call (channels x max_in) — the time-domain kernel stays
 * (a block transform costs a call 7 us however little it filters). On an FFT-backed plan sdrhip_fir_set_taps SYNCHRONISES the
 * context's stream (the new kernel spectrum is transformed and swapped in before it returns); on the others it is stream-ordered.
 * New coefficients for the SAME order between calls: FIRFilter::setLowerFreq / setUpperFreq (FIRLowPass::setFreq) only
 * recompute _alpha — the ring, and with it the stream, goes on (reference src/firfilter.hh:155-170,287). `alpha`: order doubles. */
int sdrhip_fir_set_taps(sdrhip_fir *h, const double *alpha);
int sdrhip_fir_reset(sdrhip_fir *h); /* ring zeroed, as FIRFilter::config does (:193-195) */
int sdrhip_fir_destroy(sdrhip_fir *h);

/* ---- K4/K5: stand-alone demodulators ------------------------------------------------------ */
enum { SDRHIP_T_CS16 = 0, SDRHIP_T_CF32 = 1, SDRHIP_T_CS8 = 2 /* complex<int8_t>: FMDemod<int8_t,int16_t> only */,
       SDRHIP_T_CF64 = 3 /* complex<double>: sdrhip_fft_exec only */ };
/* kind = SDRHIP_EPI_FM|AM|USB, dtype = SDRHIP_T_*; FM exists for cs16 only (the reference's
 * fast_atan2 has no float form, src/math.hh:9-40). inplace_fm0: 1 -> out[0] = in[0].real() per call
 * (in-place chain), 0 -> out[0] left untouched (the reference leaves it uninitialised). */
int sdrhip_demod_create(sdrhip_ctx *ctx, int kind, int dtype, int channels, size_t max_in, int inplace_fm0,
                        sdrhip_demod **out);
int sdrhip_demod_process(sdrhip_demod *h, const void *in_host, size_t n, size_t in_stride, void *out_host,
                         size_t out_stride);
int sdrhip_demod_process_dev(sdrhip_demod *h, const void *in_dev, size_t n, size_t in_stride, void *out_dev,
                             size_t out_stride);
int sdrhip_demod_reset(sdrhip_demod *h);
int sdrhip_demod_destroy(sdrhip_demod *h);

/* ---- FMDeemph<int16_t> ("next" row; reference src/demod.hh:272-362) ------------ */
/* avg += (x - avg +/- alpha/2) / alpha per sample, a nonlinear integer recursion: one lane per channel.
 * alpha = round(1/(1-exp(-1/(Fs*75us)))) (src/demod.hh:305-306) from sdrhip_design_fmdeemph_alpha. */
typedef struct sdrhip_deemph sdrhip_deemph;
int sdrhip_design_fmdeemph_alpha(double sample_rate, int *alpha);
int sdrhip_deemph_i16_create(sdrhip_ctx *ctx, int alpha, int channels, size_t max_in, sdrhip_deemph **out);
int sdrhip_deemph_i16_process(sdrhip_deemph *h, const int16_t *in_host, size_t n, size_t in_stride, int16_t *out_host,
                              size_t out_stride);
int sdrhip_deemph_i16_process_dev(sdrhip_deemph *h, const int16_t *in_dev, size_t n, size_t in_stride, int16_t *out_dev,
                                  size_t out_stride);
/* The kernel a call of n samples per channel runs (0: the plan's max_in): "deemph_i16_copy_kernel" (alpha = 1: the update is
 * avg = x), "deemph_i16_seq_kernel" (one lane walks one channel), "deemph_i16_spec_kernel" (alpha <= 32 and rows of at least
 * 64 alpha samples: 4 … 32 lanes per channel start their segments from guessed states that the kernel checks and repairs —
 * the same bits as the sequential recursion, whatever the data). */
int sdrhip_deemph_i16_kernel_names(sdrhip_deemph *h, size_t n, char *buf, size_t len);
int sdrhip_deemph_i16_reset(sdrhip_deemph *h);
int sdrhip_deemph_i16_destroy(sdrhip_deemph *h);

/* ---- K6: SubSample<complex<int16_t>|complex<float>> (src/subsample.hh:92-101) ------------- */
int sdrhip_subsample_create(sdrhip_ctx *ctx, int dtype, size_t n, int channels, size_t max_in,
                            sdrhip_subsample **out);
int sdrhip_subsample_out_count(sdrhip_subsample *h, size_t n_in, size_t *n_out);
int sdrhip_subsample_process(sdrhip_subsample *h, const void *in_host, size_t n_in, size_t in_stride,
                             void *out_host, size_t out_stride, size_t *n_out);
int sdrhip_subsample_process_dev(sdrhip_subsample *h, const void *in_dev, size_t n_in, size_t in_stride,
                                 void *out_dev, size_t out_stride, size_t *n_out);
int sdrhip_subsample_reset(sdrhip_subsample *h);
int sdrhip_subsample_destroy(sdrhip_subsample *h);

/* ---- K7: FFT convolution (FilterSink + FilterSource, src/filternode.hh:81-88,164-181) ----- */
enum {
  SDRHIP_FFTCONV_OLA = 0, /* reference mode: block N, FFT 2N, kernel = 2N-point spectrum K,
                             out = last + IFFT(FFT([x,0]) * K)/2N  (overlap-add) */
  SDRHIP_FFTCONV_OLS = 1  /* overlap-save with M real/complex taps, FFT size L, hop L-M+1 (rounded down to even: one more
                             sample of history keeps every block of an aligned call 16-byte aligned)
                             (BASELINE config 4: L=16384, M=4097); plain causal convolution */
};
/* OLA: fft_size = 2N, kernel = 2N cf32 spectrum (already normalised), n_taps ignored. The result is the
 *      reference's overlap-add stream y = IDFT(DFT([x,0]) * K)/2N summed over blocks, i.e. the causal
 *      convolution with the N-tap kernel behind K; it is evaluated by overlap-save underneath, so a call
 *      may carry ANY number of samples (the reference's FilterSink insists on exactly N per buffer,
 *      src/filternode.hh:69-74 — a multiple of N gives its blocks exactly).
 * OLS: kernel = n_taps cf32 time-domain taps; any n_in. */
int sdrhip_fftconv_create(sdrhip_ctx *ctx, int mode, int fft_size, const float *kernel, int n_taps,
                          int channels, size_t max_in, sdrhip_fftconv **out);
/* Filter bank (FilterNode, reference src/filternode.hh:232-284): n_bands kernels behind ONE forward transform per
 * input block, as FilterSink's spectrum feeds every FilterSource (:81-88,257-270). `kernels` holds the bands' kernels
 * back to back (OLA: 2N cf32 each; OLS: n_taps cf32 each). process / process_dev then write n_bands x channels rows:
 * band b's channel c at row b*channels + c (row stride out_stride). Plans whose FFT does not fit the CU's LDS twice
 * (fft_size 16384) transform the input once per band instead. set_kernel swaps one band's kernel between calls
 * (FilterSource::setFreq -> _updateFilter, :132-139,186-203) and keeps the stream going; the convolution is evaluated by
 * overlap-save, so the block after the swap is the NEW kernel applied to the input history, where the reference's
 * overlap-add adds the OLD kernel's tail to the new kernel's head — a one-block transient, stated here, not hidden. */
int sdrhip_fftconv_create_bank(sdrhip_ctx *ctx, int mode, int fft_size, const float *kernels, int n_taps, int n_bands,
                               int channels, size_t max_in, sdrhip_fftconv **out);
int sdrhip_fftconv_bands(sdrhip_fftconv *h, int *n_bands);
int sdrhip_fftconv_set_kernel(sdrhip_fftconv *h, int band, const float *kernel);
int sdrhip_fftconv_process(sdrhip_fftconv *h, const float *in_host, size_t n_in, size_t in_stride,
                           float *out_host, size_t out_stride);
int sdrhip_fftconv_process_dev(sdrhip_fftconv *h, const float *in_dev, size_t n_in, size_t in_stride,
                               float *out_dev, size_t out_stride);
int sdrhip_fftconv_reset(sdrhip_fftconv *h);
int sdrhip_fftconv_destroy(sdrhip_fftconv *h);
/* FFT sizes: the reference plans ANY size (FilterNode(size_t block_size = 1024), src/filternode.hh:235-245;
 * fftw_plan_dft_1d(in.size(), ...), src/fftplan_fftw3.hh:34-36), and so do these entry points, the filter included:
 *   - powers of two from 4 to 16384 in complex<float>: the tuned radix-16 kernels (every BASELINE configuration);
 *   - any other fft_size made of the prime factors 2 ... 13 that fits one workgroup's LDS (16384 points in float, 8192 in
 *     double): the general in-LDS mixed-radix plan (csrc/fftgen.hpp), one launch per call;
 *   - every other size (csrc/fftany.hpp): longer transforms by the four-step plan n = n1 x n2 through device memory, sizes
 *     with a prime factor above 13 by Bluestein's chirp transform over the next power of two >= 2n - 1 (in LDS, or over a
 *     four-step plan beyond it); the filter then runs as passes over device memory (gather blocks, forward transforms,
 *     per band: spectrum product, backward transforms, scatter) — FilterNode<float>(16384), (12000), (1009),
 *     FilterNode<double>(8192) ... Limit: 2^27 points (2^25 with a large prime factor).
 * An OLA plan whose fft_size = 2N is not one of the tuned powers of two does not have to transform 2N points at all: its
 * result is the N-tap convolution whatever evaluates it, so it runs as overlap-save with the same N taps (the first N points
 * of the spectrum's inverse DFT, taken on the host in double) on the power of two that costs least per output — provided N
 * leaves such a block a quarter of its points (N <= 12289; 6144 in double). FilterNode<float>(1000), (1009), (12000) run the
 * tuned kernels that way — and so does a single band on a power of two other than 2048 (half of every 2N-point block is
 * overlap, a longer block keeps up to 7/8 of its points); filter banks keep their own transform, and so does the 2048-point
 * plan unless the call has blocks enough for the 16384-point kernel's pipelined form (4 per CU). The transform size follows the
 * taps and the blocks a call has (channels x max_in): 2048 points up to 768 taps; from there 16384 points wherever that form can
 * walk, otherwise 4096 points up to 2048 taps and 16384 beyond (measured: profiles/r19_fft_rank.txt). Results agree within the
 * tolerance whatever the size; two plans of different channels / max_in may therefore differ in the last bits.
 * SDRHIP_FFTCONV_LITERAL=1 in the environment keeps the 2N-point transform everywhere (tests).
 *
 * FilterNode<double> (the filter classes are templates over Scalar, src/filternode.hh:30-32,102-104,230-232): the same
 * plan on complex<double> buffers; kernels / spectra are doubles (sdrhip_design_fftfilt_*_f64). bands / reset / destroy
 * are the calls above; a handle made here refuses the complex<float> entry points and vice versa. */
int sdrhip_fftconv_f64_create_bank(sdrhip_ctx *ctx, int mode, int fft_size, const double *kernels, int n_taps, int n_bands,
                                   int channels, size_t max_in, sdrhip_fftconv **out);
int sdrhip_fftconv_f64_set_kernel(sdrhip_fftconv *h, int band, const double *kernel);
int sdrhip_fftconv_f64_process(sdrhip_fftconv *h, const double *in_host, size_t n_in, size_t in_stride,
                               double *out_host, size_t out_stride);
int sdrhip_fftconv_f64_process_dev(sdrhip_fftconv *h, const double *in_dev, size_t n_in, size_t in_stride,
                                   double *out_dev, size_t out_stride);
/* plain batched DFT of the library's own FFT (FFTPlan<float>): sign -1 forward / +1 backward, unnormalised; any n (see
 * above) */
int sdrhip_fft_c2c(sdrhip_ctx *ctx, int n, int sign, int batch, const float *in_dev, float *out_dev);
/* FFTPlan<double> (reference src/fftplan_fftw3.hh:12-76): the same on complex<double> (in LDS, double arithmetic, roots
 * from a host table made in long double). */
int sdrhip_fft_c2c_f64(sdrhip_ctx *ctx, int n, int sign, int batch, const double *in_dev, double *out_dev);
/* FFT::exec / FFTPlan<Scalar>::operator() on HOST buffers (reference src/fftplan.hh:22-36): one transform of n points,
 * dtype SDRHIP_T_CF32 or SDRHIP_T_CF64, sign -1 = FFT::FORWARD, +1 = FFT::BACKWARD, unnormalised like FFTW. in == out
 * (the in-place plan) is allowed. */
int sdrhip_fft_exec(sdrhip_ctx *ctx, int dtype, int n, int sign, const void *in_host, void *out_host);
/* FFTPlan<Scalar> as the reference builds it: PLANNED ONCE in the constructor (fftw_plan_dft_1d, src/fftplan_fftw3.hh:34-36,
 * 52-54,102-104,120-122), executed by operator() (fftw_execute, :59,127), destroyed with the object (:64,132). The plan
 * owns its tables (roots, permutation, chirp, twiddles) and scratch on the context's device. `form` names what was
 * planned: "radix-16 lds", "radix-2 lds (double)", "lds", "four-step", "chirp", "chirp over four-step". exec_dev:
 * `batch` contiguous transforms of n points in device memory, asynchronous on the context's stream, in == out allowed;
 * exec: one transform on host buffers (FFTPlan::operator()), returns when out_host is written. The one-shot calls above
 * keep such plans in the context, keyed by (dtype, n). */
typedef struct sdrhip_fft_plan sdrhip_fft_plan;
int sdrhip_fft_plan_create(sdrhip_ctx *ctx, int dtype, int n, sdrhip_fft_plan **out);
int sdrhip_fft_plan_form(sdrhip_fft_plan *p, const char **name);
int sdrhip_fft_plan_exec_dev(sdrhip_fft_plan *p, int sign, int batch, const void *in_dev, void *out_dev);
int sdrhip_fft_plan_exec(sdrhip_fft_plan *p, int sign, const void *in_host, void *out_host);
int sdrhip_fft_plan_destroy(sdrhip_fft_plan *p);

/* ---- float baseband (BASELINE config 2; build-defined) ---------------------- */
/* y = SubSample_D( FIR_cf32( x[n] * exp(-2*pi*i*Fc*n/Fs) ) ); the reference has no float
 * baseband (IQBaseBand<float> does not compile, FreqShift<float> is wrong: SURVEY fact 6). */
int sdrhip_fbb_f32_create(sdrhip_ctx *ctx, double Fc, double Fs, const double *alpha, int order, int decim,
                          int channels, size_t max_in, sdrhip_fbb_f32 **out);
int sdrhip_fbb_f32_out_count(sdrhip_fbb_f32 *h, size_t n_in, size_t *n_out);
int sdrhip_fbb_f32_kernel_names(sdrhip_fbb_f32 *h, size_t n_in, char *buf, size_t len);   /* as sdrhip_fir_kernel_names */
int sdrhip_fbb_f32_process(sdrhip_fbb_f32 *h, const float *in_host, size_t n_in, size_t in_stride,
                           float *out_host, size_t out_stride, size_t *n_out);
int sdrhip_fbb_f32_process_dev(sdrhip_fbb_f32 *h, const float *in_dev, size_t n_in, size_t in_stride,
                               float *out_dev, size_t out_stride, size_t *n_out);
/* Setters that keep the stream going (same order, decimation and buffer size: no new plan). set_taps: new low-pass
 * coefficients (`order` doubles) — what IQBaseBand::setFilterWidth / setFilterFrequency do to the kernel, the ring kept
 * (reference src/baseband.hh:88-101). set_shift: a new centre frequency; the phasor restarts at the current sample,
 * exp(-2 pi i Fc (n - n_now) / Fs), as setCenterFrequency -> _update_lut_incr restarts the LUT counter (src/baseband.hh:82-86,
 * src/freqshift.hh:78-87); FIR history (raw input samples), decimator phase and sample counter go on. */
int sdrhip_fbb_f32_set_taps(sdrhip_fbb_f32 *h, const double *alpha);
int sdrhip_fbb_f32_set_shift(sdrhip_fbb_f32 *h, double Fc);
int sdrhip_fbb_f32_reset(sdrhip_fbb_f32 *h);
int sdrhip_fbb_f32_destroy(sdrhip_fbb_f32 *h);

/* ---- one process, several GPUs (BASELINE config 5 from the C++ side) ----------------------------- */
/* The reference composes many inputs through one port per input (Combine::sink(i), reference src/combine.hh:66-150);
 * here the independent channels of such a bank are split into contiguous blocks, one per device. The data path has
 * no collective; the two exchanges are a broadcast of the read-only design (taps / LUT / FFT kernel, KBs, at config
 * time) and one gather of the demodulated output per step. `devices[r]` is the HIP device of rank r. Collectives
 * are RCCL over xGMI (librccl is opened with dlopen at the first create that needs it) when the ranks sit on
 * distinct devices; ranks that all share one device (single-GPU boxes: tests) use device-to-device copies on that
 * device instead — RCCL refuses two ranks on one device. Every call is asynchronous on the ranks' streams and ordered
 * on them in both directions under either transport: work enqueued afterwards on any rank's stream (the next step
 * overwriting a send buffer, the root reusing its broadcast buffer) runs after the transfer has read what it needs —
 * no sdrhip_comm_synchronize is required between steps (tests/cpp/test_gpu_nodes.cc: three pipelined gathers). */
typedef struct sdrhip_comm sdrhip_comm;
int sdrhip_comm_create(const int *devices, int nranks, sdrhip_comm **out);
int sdrhip_comm_size(sdrhip_comm *c, int *nranks);
/* rank r's context (its own stream on devices[r]); borrowed, lives as long as the comm: create the rank's plans on it */
int sdrhip_comm_ctx(sdrhip_comm *c, int rank, sdrhip_ctx **ctx);
int sdrhip_comm_transport(sdrhip_comm *c, const char **name); /* "rccl" or "same-device copies" */
/* bufs_dev[root] -> bufs_dev[r] for every rank r (bytes each) */
int sdrhip_comm_broadcast(sdrhip_comm *c, void *const *bufs_dev, size_t bytes, int root);
/* rank r's bytes[r] bytes at send_dev[r] -> recv_dev (on the root's device), concatenated in rank order */
int sdrhip_comm_gather(sdrhip_comm *c, const void *const *send_dev, const size_t *bytes, void *recv_dev, int root);
/* The same gather OVERLAPPED with the ranks' next kernels (double-buffered outputs, as the reference's nodes keep sending
 * while downstream still reads the previous buffer): the transfer runs on streams the comm owns, behind what the ranks'
 * streams have enqueued so far, and nothing waits for it until sdrhip_comm_gather_wait(c, slot) enqueues — on every rank's
 * stream — a wait for the transfer begun under `slot` (0 ... 3; a slot never begun is no wait). Between begin and wait the
 * caller neither overwrites send_dev[r] nor reads recv_dev. Typical step k: wait(k & 1); launch the kernels writing output
 * buffer k & 1; begin(k & 1, ...). */
int sdrhip_comm_gather_begin(sdrhip_comm *c, int slot, const void *const *send_dev, const size_t *bytes, void *recv_dev, int root);
int sdrhip_comm_gather_wait(sdrhip_comm *c, int slot);
int sdrhip_comm_synchronize(sdrhip_comm *c);   /* every rank's stream, and the comm's own transfer streams */
int sdrhip_comm_destroy(sdrhip_comm *c);

/* ---- pinned host memory, asynchronous copies (staging of the many-channel nodes) --------------------------------- */
/* The *_process entry points take pageable host memory and let the runtime stage it. A node that owns its staging
 * buffers (gpu::ChannelBank) allocates them pinned, so that H2D / D2H run as plain DMA on the rank's stream and the
 * devices of a multi-GPU bank copy and compute at the same time. host_register pins memory the caller owns. */
int sdrhip_host_alloc(size_t bytes, void **hptr);
int sdrhip_host_free(void *hptr);
int sdrhip_host_register(void *hptr, size_t bytes);
int sdrhip_host_unregister(void *hptr);
int sdrhip_memcpy_h2d_async(sdrhip_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int sdrhip_memcpy_d2h_async(sdrhip_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int sdrhip_memcpy2d_d2h_async(sdrhip_ctx *ctx, void *dst_host, size_t dst_pitch, const void *src_dev, size_t src_pitch,
                              size_t row_bytes, size_t rows);

#ifdef __cplusplus
}
#endif
#endif /* SDRHIP_H */
