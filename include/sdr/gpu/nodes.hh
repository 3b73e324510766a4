// nodes.hh — MI355X nodes behind libsdr's node API (header-only; link with -lsdrhip).
//
// Every class here is an ordinary sdr::Sink<T> + sdr::Source with the reference node's name,
// constructor arguments, config()/process() behaviour, ownership and error rules, so it can replace
// the CPU node in an existing graph:
//   sdr::gpu::IQBaseBand<int16_t>           <->  sdr::IQBaseBand<int16_t>            reference src/baseband.hh:22-293
//   sdr::gpu::FIRLowPass<complex<int16|float>>  <->  sdr::FIRLowPass<...>            src/firfilter.hh:117-289
//   sdr::gpu::FMDemod<int16_t>, AMDemod<S>, USBDemod<S>  <->  same names             src/demod.hh:18-264
//   sdr::gpu::SubSample<complex<...>>       <->  sdr::SubSample<...>                 src/subsample.hh:16-116
//   sdr::gpu::FilterNode<float|double>      <->  sdr::FilterNode<Scalar>             src/filternode.hh:230-284
//   sdr::gpu::FFT, FFTPlan<float|double>    <->  sdr::FFT, sdr::FFTPlan<...>        src/fftplan.hh, src/fftplan_fftw3.hh
//   sdr::gpu::ChannelBank<int16_t>          many IQBaseBand(+demod) channels in ONE batched kernel launch;
//                                           sink(c)/source(c) per channel like Combine::sink(i) (src/combine.hh:66-150)
// Rules reproduced: config() returns silently while the upstream Config is incomplete and throws
// ConfigError on a type mismatch; the output buffer is owned by the node, allocated in config(),
// reused only when isUnused(), otherwise the input is dropped; a node writes into its input only
// when allow_overwrite; views (out.head(n)) are sent, never copies; process() never throws (device
// errors are logged at LOG_ERROR and the buffer is dropped).
//
// The header compiles against this repository's core (include/sdr/node.hh) or, when the
// reference's own node.hh was included first, against the reference core unchanged.
#ifndef SDR_GPU_NODES_HH
#define SDR_GPU_NODES_HH

#if !defined(__SDR_NODE_HH__) && !defined(SDR_CORE_NODE_HH)
#include "../node.hh"
#include "../logger.hh"
#endif

#include <algorithm>
#include <complex>
#include <cstring>
#include <list>
#include <map>
#include <vector>

#include "../../sdrhip.h"
#include "design.hh"

namespace sdr {
namespace gpu {

typedef std::complex<int16_t> cs16;
typedef std::complex<float> cf32;

/** One shared device context per HIP device. Throws ConfigError when there is no GPU (no CPU fallback). */
class Device {
public:
  static sdrhip_ctx *get(int device = 0) {
    static std::map<int, sdrhip_ctx *> ctxs;
    std::map<int, sdrhip_ctx *>::iterator it = ctxs.find(device);
    if (it != ctxs.end()) return it->second;
    sdrhip_ctx *c = 0;
    const int rc = sdrhip_ctx_create(device, 0, &c);
    if (rc != SDRHIP_OK) {
      ConfigError err;
      err << "sdr::gpu: can not open HIP device " << device << ": " << sdrhip_strerror(rc) << " (" << sdrhip_last_error() << ")";
      throw err;
    }
    ctxs[device] = c;
    return c;
  }
};

namespace detail {
inline void configCheck(int rc, const char *what) {
  if (rc == SDRHIP_OK) return;
  ConfigError err;
  err << "Can not configure " << what << ": " << sdrhip_strerror(rc) << " (" << sdrhip_last_error() << ")";
  throw err;
}
inline bool processOk(int rc, const char *what) {
  if (rc == SDRHIP_OK) return true;
  LogMessage msg(LOG_ERROR);
  msg << what << ": drop buffer: " << sdrhip_strerror(rc) << " (" << sdrhip_last_error() << ")";
  Logger::get().log(msg);
  return false;
}
template <class T> struct TypeTag;
template <> struct TypeTag<cs16> { enum { dtype = SDRHIP_T_CS16, fir = SDRHIP_FIR_CS16_EXACT }; typedef int16_t Real; };
template <> struct TypeTag<cf32> { enum { dtype = SDRHIP_T_CF32, fir = SDRHIP_FIR_CF32 }; typedef float Real; };
template <> struct TypeTag< std::complex<int8_t> > { enum { dtype = SDRHIP_T_CS8 }; typedef int8_t Real; };
template <class S> struct RealTag;
template <> struct RealTag<int16_t> { enum { dtype = SDRHIP_T_CS16 }; };
template <> struct RealTag<float> { enum { dtype = SDRHIP_T_CF32 }; };
}  // namespace detail

// =================================================================================================
// IQBaseBand<int16_t>
// =================================================================================================
namespace detail {
/** IQBaseBand on the int16 kernels; SIn = int16_t (complex<int16_t> in) or uint8_t (complex<uint8_t> in, with
 * AutoCast< complex<int16_t> > fused into the load: the cast -> baseband pair of examples/sdr_fm.cc:49-50). */
/** Sample types per input scalar: int16_t and uint8_t (AutoCast fused) produce complex<int16_t>; int8_t is
 * IQBaseBand<int8_t>, which produces complex<int8_t> (reference src/baseband.hh:22-31, src/sdr.hh:225-240). */
template <class S> struct BbIo { typedef cs16 COut; enum { int8 = 0, cu8 = 0 }; };
template <> struct BbIo<uint8_t> { typedef cs16 COut; enum { int8 = 0, cu8 = 1 }; };
template <> struct BbIo<int8_t> { typedef std::complex<int8_t> COut; enum { int8 = 1, cu8 = 0 }; };

template <class SIn>
class IQBB16 : public Sink< std::complex<SIn> >, public Source {
public:
  typedef std::complex<SIn> CIn;
  typedef typename BbIo<SIn>::COut COut;
  enum { kInt8 = BbIo<SIn>::int8, kCu8 = BbIo<SIn>::cu8 };
  IQBB16(double Fc, double width, size_t order, size_t sub_sample, double oFs = 0.0, int device = 0)
    : _Fc(Fc), _Ff(Fc), _shift(Fc), _Fs(0), _width(width), _order(std::max(size_t(1), order)), _sub_sample(sub_sample),
      _oFs(oFs), _sourceBs(0), _epilogue(SDRHIP_EPI_NONE), _device(device), _plan(0) {}
  IQBB16(double Fc, double Ff, double width, size_t order, size_t sub_sample, double oFs = 0.0, int device = 0)
    : _Fc(Fc), _Ff(Ff), _shift(Fc), _Fs(0), _width(width), _order(std::max(size_t(1), order)), _sub_sample(sub_sample),
      _oFs(oFs), _sourceBs(0), _epilogue(SDRHIP_EPI_NONE), _device(device), _plan(0) {}
  virtual ~IQBB16() {
    if (_plan) sdrhip_iqbb_i16_destroy(_plan);
    _buffer.unref();
  }

  /** Extension: fuse FMDemod / AMDemod / USBDemod<int16_t> (run in place on the node's output, as
   * `baseband.connect(&demod, true)` does in examples/sdr_fm.cc:51) into the same kernel launch.
   * The node then is a source of int16_t. Call before connecting / configuring. */
  void setDemod(int epilogue) { _epilogue = epilogue; if (_Fs) _reconfigure(); }

  inline size_t order() const { return _order; }
  /** As the reference (src/baseband.hh:69-79): a new kernel and a NEW ring; the decimator window, the sample counter, the
   * LUT phase and the Config go on untouched. (The reference's new ring is uninitialised memory until `order` samples
   * have passed; here it is zeros.) */
  void setOrder(size_t o) {
    _order = std::max(size_t(1), o);
    if (_plan) _newPlan(SDRHIP_KEEP_FM | SDRHIP_KEEP_COUNTERS);
  }
  inline double centerFrequency() const { return _Fc; }
  /** As the reference (src/baseband.hh:84-86 -> src/freqshift.hh:52-54,78-87): new LUT increment and sign, the LUT
   * phase restarts; filter history, decimator state and the kernel go on unchanged. */
  void setCenterFrequency(double Fc) {
    _Fc = int32_t(Fc); _shift = _Fc;
    if (_plan) configCheck(sdrhip_iqbb_i16_set_shift(_plan, design::freqShiftIncrement(_shift, double(_Fs)), 0 > _shift), "IQBaseBand");
  }
  inline double filterFrequency() const { return _Ff; }
  /** As the reference (:92-104): only the filter kernel is recomputed, all streaming state goes on. */
  void setFilterFrequency(double Ff) { _Ff = int32_t(Ff); _retap(); }
  inline double filterWidth() const { return _width; }
  void setFilterWidth(double width) { _width = int32_t(width); _retap(); }
  size_t subSample() const { return _sub_sample; }
  void setSubsample(size_t sub_sample) { _sub_sample = std::max(size_t(1), sub_sample); if (_Fs) _reconfigure(); }
  void setOutputSampleRate(double Fs) { _oFs = Fs; if (_Fs) _reconfigure(); }

  virtual void config(const Config &src_cfg) {
    if (!src_cfg.hasType() || !src_cfg.hasSampleRate() || !src_cfg.hasBufferSize()) return;
    if (Config::typeId<CIn>() != src_cfg.type()) {
      ConfigError err;
      err << "Can not configure IQBaseBand: Invalid type " << src_cfg.type() << ", expected " << Config::typeId<CIn>();
      throw err;
    }
    _Fs = int32_t(src_cfg.sampleRate());
    _sourceBs = src_cfg.bufferSize();
    _reconfigure();
  }

  virtual void process(const Buffer<CIn> &buffer, bool allow_overwrite) {
    if (!_plan) return;
    if (allow_overwrite && sizeof(CIn) == sizeof(COut)) _process(buffer, Buffer<COut>(buffer));   // in place needs equal sample sizes
    else if (_buffer.isUnused()) _process(buffer, _buffer);
    // else: output buffer still in use downstream -> the input is dropped (src/baseband.hh:141-150)
  }

protected:
  /** _update_filter_kernel() on a configured node: swap the kernel, keep every bit of streaming state. */
  void _retap() {
    if (!_plan) return;
    std::vector<int32_t> taps(2 * _order);
    design::iqbbTaps(_Ff, _width, _Fs, _order, taps.data());
    const int rc = sdrhip_iqbb_i16_set_taps(_plan, taps.data());
    // the new kernel does not fit the plan's formulation (tap bytes): a new plan that takes the whole stream state over
    if (rc == SDRHIP_E_UNSUPPORTED) _newPlan(SDRHIP_KEEP_RING | SDRHIP_KEEP_FM | SDRHIP_KEEP_COUNTERS);
    else configCheck(rc, "IQBaseBand");
  }

  /** A device plan for the node's present parameters; `carry` (SDRHIP_KEEP_*) names the streaming state it takes over
   * from the plan it replaces (sdrhip_iqbb_i16_adopt_state), so that a new plan is not an event of its own. */
  void _newPlan(int carry) {
    const size_t D = std::max(size_t(1), _sub_sample);
    std::vector<int32_t> taps(2 * _order), lut(2 * design::kLutSize);
    design::iqbbTaps(_Ff, _width, _Fs, _order, taps.data());
    if (kInt8) design::freqShiftLutI8(lut.data()); else design::freqShiftLutI16(lut.data());
    const uint32_t inc = design::freqShiftIncrement(_shift, double(_Fs));
    sdrhip_iqbb_i16 *neu = 0;
    if (kInt8) configCheck(sdrhip_iqbb_i8_create(Device::get(_device), taps.data(), int(_order), lut.data(), inc, 0 > _shift,
                                                 int(D), 1, _sourceBs, _epilogue, &neu), "IQBaseBand");
    else configCheck(sdrhip_iqbb_i16_create(Device::get(_device), taps.data(), int(_order), lut.data(), inc, 0 > _shift,
                                            int(D), 1, _sourceBs, _epilogue, &neu), "IQBaseBand");
    int rc = kCu8 ? sdrhip_iqbb_i16_set_input_format(neu, SDRHIP_IN_CU8) : SDRHIP_OK;
    if (rc == SDRHIP_OK && _plan) {
      if (_planOrder != _order) carry &= ~SDRHIP_KEEP_RING;   // (a new order is a new ring, src/baseband.hh:75-76)
      if (_planEpi != SDRHIP_EPI_FM || _epilogue != SDRHIP_EPI_FM) carry &= ~SDRHIP_KEEP_FM;
      rc = sdrhip_iqbb_i16_adopt_state(neu, _plan, carry);
    }
    if (rc != SDRHIP_OK) { sdrhip_iqbb_i16_destroy(neu); configCheck(rc, "IQBaseBand"); }
    if (_plan) sdrhip_iqbb_i16_destroy(_plan);
    _plan = neu;
    _planOrder = _order; _planD = D; _planBs = _sourceBs; _planEpi = _epilogue;
  }

  /** IQBaseBand::_reconfigure (src/baseband.hh:156-194): kernel and LUT increment recomputed, counters and phases reset,
   * the FIR ring's CONTENTS kept where they lie (read rotated afterwards) — on the plan in place
   * (sdrhip_iqbb_i16_reset(keep_history = 1)) or, when the geometry changed (decimation, buffer size, demodulator),
   * carried into the new device plan. */
  void _reconfigure() {
    const size_t D = design::iqbbDecimation(_Fs, _sub_sample, _oFs);
    _sub_sample = D;
    size_t buffer_size = _sourceBs / D;
    if (_sourceBs % D) buffer_size += 1;
    const double oRate = double(size_t(_Fs) / D);   // the reference divides int32 by size_t (src/baseband.hh:192-193)
    const Config out_cfg(_epilogue == SDRHIP_EPI_NONE ? Config::typeId<COut>() : Config::typeId<int16_t>(), oRate, buffer_size, 1);
    // (a fused demodulator is reconfigured — FM's last angle zeroed — only if the Config we propagate changes,
    // src/node.cc:98-105, src/demod.hh:210)
    const bool same_cfg = (out_cfg == this->_config);
    bool reuse = _plan && _planOrder == _order && _planD == D && _planBs == _sourceBs && _planEpi == _epilogue;
    if (reuse) {
      std::vector<int32_t> taps(2 * _order);
      design::iqbbTaps(_Ff, _width, _Fs, _order, taps.data());
      const int rc = sdrhip_iqbb_i16_set_taps(_plan, taps.data());
      if (rc == SDRHIP_E_UNSUPPORTED) reuse = false;
      else {
        configCheck(rc, "IQBaseBand");
        configCheck(sdrhip_iqbb_i16_set_shift(_plan, design::freqShiftIncrement(_shift, double(_Fs)), 0 > _shift), "IQBaseBand");
        configCheck(sdrhip_iqbb_i16_reset(_plan, same_cfg ? 3 : 1), "IQBaseBand");
      }
    }
    if (!reuse) _newPlan(SDRHIP_KEEP_RING | (same_cfg ? SDRHIP_KEEP_FM : 0));
    _buffer.unref();
    _buffer = Buffer<COut>(buffer_size);

    LogMessage msg(LOG_DEBUG);
    msg << "Configured gpu::IQBaseBand node:" << std::endl << " sample-rate " << _Fs << "Hz" << std::endl
        << " center freq " << _Fc << "Hz" << std::endl << " width " << _width << "Hz" << std::endl
        << " in buffer size " << _sourceBs << std::endl << " sub-sample by " << D << std::endl
        << " out buffer size " << buffer_size;
    Logger::get().log(msg);

    this->setConfig(out_cfg);
  }

  void _process(const Buffer<CIn> &in, const Buffer<COut> &out) {
    size_t n = 0;
    // (out_stride in output elements: the demodulators' int16 fits sizeof(COut) / 2 times into a complex output sample)
    if (!processOk(sdrhip_iqbb_i16_process(_plan, reinterpret_cast<const int16_t *>(in.data()), in.size(), 0,
                                                   out.data(), out.size() * (_epilogue == SDRHIP_EPI_NONE ? 1 : sizeof(COut) / 2), &n),
                           "gpu::IQBaseBand"))
      return;
    if (_epilogue == SDRHIP_EPI_NONE) this->send(out.head(n), true);
    else if (_epilogue == SDRHIP_EPI_FM) { if (n) this->send(Buffer<int16_t>(out).head(n), false); }   // FMDemod: no send when empty
    else this->send(Buffer<int16_t>(out).head(n), _epilogue == SDRHIP_EPI_AM);
  }

  int32_t _Fc, _Ff;
  double _shift;
  int32_t _Fs, _width;
  size_t _order, _sub_sample;
  double _oFs;
  size_t _sourceBs;
  int _epilogue, _device;
  sdrhip_iqbb_i16 *_plan;
  size_t _planOrder = 0, _planD = 0, _planBs = 0;   // geometry the device plan was made for
  int _planEpi = 0;
  Buffer<COut> _buffer;
};
}  // namespace detail

template <class Scalar> class IQBaseBand;
/** Drop-in for sdr::IQBaseBand<int16_t>. */
template <>
class IQBaseBand<int16_t> : public detail::IQBB16<int16_t> {
public:
  IQBaseBand(double Fc, double width, size_t order, size_t sub_sample, double oFs = 0.0, int device = 0)
    : detail::IQBB16<int16_t>(Fc, width, order, sub_sample, oFs, device) {}
  IQBaseBand(double Fc, double Ff, double width, size_t order, size_t sub_sample, double oFs = 0.0, int device = 0)
    : detail::IQBB16<int16_t>(Fc, Ff, width, order, sub_sample, oFs, device) {}
};
/** Drop-in for sdr::IQBaseBand<int8_t>, the baseband of the reference's documentation example (src/sdr.hh:225-240):
 * sinks and sources complex<int8_t>. */
template <>
class IQBaseBand<int8_t> : public detail::IQBB16<int8_t> {
public:
  IQBaseBand(double Fc, double width, size_t order, size_t sub_sample, double oFs = 0.0, int device = 0)
    : detail::IQBB16<int8_t>(Fc, width, order, sub_sample, oFs, device) {}
  IQBaseBand(double Fc, double Ff, double width, size_t order, size_t sub_sample, double oFs = 0.0, int device = 0)
    : detail::IQBB16<int8_t>(Fc, Ff, width, order, sub_sample, oFs, device) {}
};
/** AutoCast< complex<int16_t> > + IQBaseBand<int16_t> in one node: sinks complex<uint8_t> (RTL-SDR bytes). */
template <>
class IQBaseBand<uint8_t> : public detail::IQBB16<uint8_t> {
public:
  IQBaseBand(double Fc, double width, size_t order, size_t sub_sample, double oFs = 0.0, int device = 0)
    : detail::IQBB16<uint8_t>(Fc, width, order, sub_sample, oFs, device) {}
  IQBaseBand(double Fc, double Ff, double width, size_t order, size_t sub_sample, double oFs = 0.0, int device = 0)
    : detail::IQBB16<uint8_t>(Fc, Ff, width, order, sub_sample, oFs, device) {}
};

/** IQBaseBand<float> — BASELINE config 2's node. The reference has NO working float baseband (IQBaseBand<float> does not
 * compile: its compute type is hard-wired to int32, src/baseband.hh:28-31,205; FreqShift<float> truncates every sample to
 * int16), so this node is BUILD-DEFINED behind the reference's constructor signatures
 * (src/baseband.hh:33-37) and its config / ownership / drop rules (:115-151):
 *     y = SubSample_D( FIRLowPass_cf32( x[n] * exp(-2*pi*i*Fc*n/Fs) ) ),   low-pass cut-off = width / 2
 * — the band of `width` Hz around Fc moved to 0 Hz, filtered with FIRLowPassCoeffs (src/firfilter.hh:16-32: the
 * reference's own cf32 arithmetic, pinned) and box-averaged like SubSample<complex<float>> (src/subsample.hh:92-101:
 * pinned); the phasor is the float64 closed form of the absolute sample index (parity unpinned by necessity, checked
 * against that closed form, <= 1e-5). A filter frequency other than Fc would need complex taps this definition does not
 * have: it is refused with a ConfigError rather than silently filtered around the wrong centre. D = Fs / oFs when an
 * output rate is given, as the reference node (:159-162). Never in place (the output is D times shorter and the node
 * owns it); the input is dropped while the output buffer is still referenced downstream (:141-150). */
template <>
class IQBaseBand<float> : public Sink<cf32>, public Source {
public:
  IQBaseBand(double Fc, double width, size_t order, size_t sub_sample, double oFs = 0.0, int device = 0)
    : _Fc(Fc), _Ff(Fc), _width(width), _Fs(0), _order(std::max(size_t(1), order)), _sub_sample(std::max(size_t(1), sub_sample)),
      _oFs(oFs), _sourceBs(0), _device(device), _plan(0) {}
  IQBaseBand(double Fc, double Ff, double width, size_t order, size_t sub_sample, double oFs = 0.0, int device = 0)
    : _Fc(Fc), _Ff(Ff), _width(width), _Fs(0), _order(std::max(size_t(1), order)), _sub_sample(std::max(size_t(1), sub_sample)),
      _oFs(oFs), _sourceBs(0), _device(device), _plan(0) {}
  virtual ~IQBaseBand() {
    if (_plan) sdrhip_fbb_f32_destroy(_plan);
    _buffer.unref();
  }
  inline size_t order() const { return _order; }
  /** A new order is a new plan (filter history, decimator and phasor restart: the reference's setOrder reallocates its ring
   * too, src/baseband.hh:69-79); the same order changes nothing. */
  void setOrder(size_t o) { o = std::max(size_t(1), o); if (o == _order) return; _order = o; if (_Fs) _reconfigure(); }
  inline double centerFrequency() const { return _Fc; }
  /** The plan and its streaming state are KEPT (src/baseband.hh:82-86: setCenterFrequency only updates the LUT increment):
   * the phasor restarts at the current sample with the new frequency, filter history and decimator go on.
   * The build-defined float baseband filters AROUND its centre frequency (class comment): the centre takes the filter
   * frequency with it, so that a retune is one call and never passes through an unsupported (Fc, Ff) pair. */
  void setCenterFrequency(double Fc) {
    _Fc = Fc; _Ff = Fc;
    if (_plan) detail::configCheck(sdrhip_fbb_f32_set_shift(_plan, _Fc), "IQBaseBand<float>");
    else if (_Fs) _reconfigure();
  }
  inline double filterFrequency() const { return _Ff; }
  /** Checked BEFORE anything changes: on a configured node a filter frequency other than the centre frequency throws
   * ConfigError and leaves the node as it was (same values, same plan). */
  void setFilterFrequency(double Ff) {
    if (_Fs && Ff != _Fc) {
      ConfigError err;
      err << "Can not set filter frequency " << Ff << "Hz on IQBaseBand<float>: it differs from the center frequency " << _Fc
          << "Hz (the build-defined float baseband low-pass filters the shifted band: sdr/gpu/nodes.hh); the node is unchanged";
      throw err;
    }
    _Ff = Ff;   // (equal to the centre frequency: nothing to update)
    if (_Fs && !_plan) _reconfigure();
  }
  inline double filterWidth() const { return _width; }
  /** New coefficients on the SAME plan (src/baseband.hh:95-101: setFilterWidth only recomputes the kernel; the ring stays). */
  void setFilterWidth(double width) {
    _width = width;
    if (_plan) {
      std::vector<double> alpha(_order);
      design::firLowPass(_order, _width / 2, _Fs, alpha.data());
      detail::configCheck(sdrhip_fbb_f32_set_taps(_plan, alpha.data()), "IQBaseBand<float>");
    } else if (_Fs) _reconfigure();
  }
  size_t subSample() const { return _sub_sample; }
  /** A decimation that does not change keeps plan and state; a new one is a new plan (the reference runs _reconfigure:
   * counters reset, src/baseband.hh:106-112). */
  void setSubsample(size_t sub_sample) {
    sub_sample = std::max(size_t(1), sub_sample);
    const bool same = _plan && _oFs <= 0 && sub_sample == _sub_sample;
    _sub_sample = sub_sample;
    if (_Fs && !same) _reconfigure();
  }
  void setOutputSampleRate(double Fs) {
    const bool same = _plan && Fs > 0 && design::iqbbDecimation(_Fs, _sub_sample, Fs) == _sub_sample;
    _oFs = Fs;
    if (_Fs && !same) _reconfigure();
  }

  virtual void config(const Config &src_cfg) {
    if (!src_cfg.hasType() || !src_cfg.hasSampleRate() || !src_cfg.hasBufferSize()) return;
    if (Config::typeId<cf32>() != src_cfg.type()) {
      ConfigError err;
      err << "Can not configure IQBaseBand: Invalid type " << src_cfg.type() << ", expected " << Config::typeId<cf32>();
      throw err;
    }
    _Fs = src_cfg.sampleRate();
    _sourceBs = src_cfg.bufferSize();
    _reconfigure();
  }

  virtual void process(const Buffer<cf32> &buffer, bool allow_overwrite) {
    (void)allow_overwrite;
    if (!_plan) return;
    if (!_buffer.isUnused()) return;   // output still in use downstream: the input is dropped (src/baseband.hh:141-150)
    size_t n = 0;
    if (!detail::processOk(sdrhip_fbb_f32_process(_plan, reinterpret_cast<const float *>(buffer.data()), buffer.size(), 0,
                                                  reinterpret_cast<float *>(_buffer.data()), _buffer.size(), &n), "gpu::IQBaseBand<float>"))
      return;
    this->send(_buffer.head(n), true);
  }

protected:
  void _reconfigure() {
    if (_Ff != _Fc) {
      ConfigError err;
      err << "Can not configure IQBaseBand<float>: filter frequency " << _Ff << "Hz differs from the center frequency " << _Fc
          << "Hz (the build-defined float baseband low-pass filters the shifted band: sdr/gpu/nodes.hh)";
      throw err;
    }
    const size_t D = design::iqbbDecimation(_Fs, _sub_sample, _oFs);
    _sub_sample = D;
    std::vector<double> alpha(_order);
    design::firLowPass(_order, _width / 2, _Fs, alpha.data());
    if (_plan) { sdrhip_fbb_f32_destroy(_plan); _plan = 0; }
    detail::configCheck(sdrhip_fbb_f32_create(Device::get(_device), _Fc, _Fs, alpha.data(), int(_order), int(D), 1, _sourceBs, &_plan),
                        "IQBaseBand<float>");
    size_t buffer_size = _sourceBs / D + 1;   // (a call may emit one output more than Bs / D: the decimator's phase carries over)
    _buffer.unref();
    _buffer = Buffer<cf32>(buffer_size);
    LogMessage msg(LOG_DEBUG);
    msg << "Configured gpu::IQBaseBand<float> node:" << std::endl << " sample-rate " << _Fs << "Hz" << std::endl
        << " center freq " << _Fc << "Hz" << std::endl << " width " << _width << "Hz" << std::endl
        << " in buffer size " << _sourceBs << std::endl << " sub-sample by " << D << std::endl
        << " out buffer size " << buffer_size;
    Logger::get().log(msg);
    this->setConfig(Config(Config::typeId<cf32>(), _Fs / double(D), buffer_size, 1));
  }

  double _Fc, _Ff, _width, _Fs;
  size_t _order, _sub_sample;
  double _oFs;
  size_t _sourceBs;
  int _device;
  sdrhip_fbb_f32 *_plan;
  Buffer<cf32> _buffer;
};

// =================================================================================================
// BaseBand<int16_t>: the REAL-input baseband node (reference src/baseband.hh:305-529)
// =================================================================================================
template <class Scalar> class BaseBand;
/** Drop-in for sdr::BaseBand<int16_t>: sinks int16_t, sources complex<int16_t> at Fs/sub_sample. Never works in
 * place (the reference's process() ignores allow_overwrite, :408-419); drops the input while the output buffer is
 * still in use downstream. */
template <>
class BaseBand<int16_t> : public Sink<int16_t>, public Source {
public:
  BaseBand(double Fc, double width, size_t order, size_t sub_sample)
    : _shift(Fc), _Ff(Fc), _width(width), _Fs(0), _order(std::max(size_t(1), order)), _sub_sample(sub_sample), _sourceBs(0),
      _epilogue(SDRHIP_EPI_NONE), _device(0), _plan(0) {}
  BaseBand(double Fc, double Ff, double width, size_t order, size_t sub_sample, int device = 0)
    : _shift(Fc), _Ff(Ff), _width(width), _Fs(0), _order(std::max(size_t(1), order)), _sub_sample(sub_sample), _sourceBs(0),
      _epilogue(SDRHIP_EPI_NONE), _device(device), _plan(0) {}
  virtual ~BaseBand() {
    if (_plan) sdrhip_iqbb_i16_destroy(_plan);
    _buffer.unref();
  }
  /** Extension, as IQBaseBand::setDemod: fuse FM / AM / USB<int16_t> into the launch. */
  void setDemod(int epilogue) { _epilogue = epilogue; if (_Fs) _reconfigure(); }
  inline double sampleRate() const { return _Fs; }
  inline double frequencyShift() const { return _shift; }
  /** FreqShiftBase::setFrequencyShift (src/freqshift.hh:52-54,78-87): new increment and sign, the LUT phase restarts;
   * ring, decimator and kernel go on. */
  void setFrequencyShift(double F) {
    _shift = F;
    if (_plan) detail::configCheck(sdrhip_iqbb_i16_set_shift(_plan, design::freqShiftIncrement(_shift, _Fs), 0 > _shift), "BaseBand");
  }
  /** BaseBand::setSampleRate (src/baseband.hh:397-401): the LUT increment and the kernel are recomputed, nothing is
   * propagated (the reference marks that as a bug of its own, :400). */
  void setSampleRate(double Fs) {
    _Fs = Fs;
    if (!_plan) return;
    detail::configCheck(sdrhip_iqbb_i16_set_shift(_plan, design::freqShiftIncrement(_shift, _Fs), 0 > _shift), "BaseBand");
    std::vector<int32_t> taps(2 * _order);
    design::bbTaps(_Ff, _width, _Fs, _order, taps.data());
    const int rc = sdrhip_iqbb_i16_set_taps(_plan, taps.data());
    if (rc == SDRHIP_E_UNSUPPORTED) _newPlan(SDRHIP_KEEP_RING | SDRHIP_KEEP_FM | SDRHIP_KEEP_COUNTERS);
    else detail::configCheck(rc, "BaseBand");
  }

  virtual void config(const Config &src_cfg) {
    if (!src_cfg.hasType() || !src_cfg.hasSampleRate() || !src_cfg.hasBufferSize()) return;
    if (Config::typeId<int16_t>() != src_cfg.type()) {
      ConfigError err;
      err << "Can not configure BaseBand: Invalid type " << src_cfg.type() << ", expected " << Config::typeId<int16_t>();
      throw err;
    }
    _Fs = src_cfg.sampleRate();
    _sourceBs = src_cfg.bufferSize();
    _reconfigure();
  }

  virtual void process(const Buffer<int16_t> &buffer, bool allow_overwrite) {
    (void)allow_overwrite;
    if (!_plan || !_buffer.isUnused()) return;
    size_t n = 0;
    if (!detail::processOk(sdrhip_iqbb_i16_process(_plan, reinterpret_cast<const int16_t *>(buffer.data()), buffer.size(), 0, _buffer.data(),
                                                   _buffer.size() * (_epilogue == SDRHIP_EPI_NONE ? 1 : 2), &n), "gpu::BaseBand"))
      return;
    if (_epilogue == SDRHIP_EPI_NONE) this->send(_buffer.head(n), true);
    else if (_epilogue == SDRHIP_EPI_FM) { if (n) this->send(Buffer<int16_t>(_buffer).head(n), false); }
    else this->send(Buffer<int16_t>(_buffer).head(n), _epilogue == SDRHIP_EPI_AM);
  }

protected:
  void _newPlan(int carry) {
    std::vector<int32_t> taps(2 * _order), lut(2 * design::kLutSize);
    design::bbTaps(_Ff, _width, _Fs, _order, taps.data());
    design::freqShiftLutI16(lut.data());
    sdrhip_iqbb_i16 *neu = 0;
    detail::configCheck(sdrhip_bb_i16_create(Device::get(_device), taps.data(), int(_order), lut.data(), design::freqShiftIncrement(_shift, _Fs),
                                             0 > _shift, int(_sub_sample), 1, _sourceBs, _epilogue, &neu), "BaseBand");
    if (_plan) {
      if (_planEpi != SDRHIP_EPI_FM || _epilogue != SDRHIP_EPI_FM) carry &= ~SDRHIP_KEEP_FM;
      const int rc = sdrhip_iqbb_i16_adopt_state(neu, _plan, carry);
      if (rc != SDRHIP_OK) { sdrhip_iqbb_i16_destroy(neu); detail::configCheck(rc, "BaseBand"); }
      sdrhip_iqbb_i16_destroy(_plan);
    }
    _plan = neu; _planBs = _sourceBs; _planEpi = _epilogue;
  }

  /** BaseBand::config (src/baseband.hh:357-395): sample rate (LUT increment, kernel), output buffer, counters reset —
   * the ring's contents stay where they lie, as in IQBaseBand::_reconfigure. */
  void _reconfigure() {
    size_t buffer_size = _sourceBs / _sub_sample;
    if (_sourceBs % _sub_sample) buffer_size += 1;
    const Config out_cfg(_epilogue == SDRHIP_EPI_NONE ? Config::typeId<cs16>() : Config::typeId<int16_t>(), _Fs / _sub_sample, buffer_size, 1);
    const bool same_cfg = (out_cfg == this->_config);
    bool reuse = _plan && _planBs == _sourceBs && _planEpi == _epilogue;
    if (reuse) {
      std::vector<int32_t> taps(2 * _order);
      design::bbTaps(_Ff, _width, _Fs, _order, taps.data());
      const int rc = sdrhip_iqbb_i16_set_taps(_plan, taps.data());
      if (rc == SDRHIP_E_UNSUPPORTED) reuse = false;
      else {
        detail::configCheck(rc, "BaseBand");
        detail::configCheck(sdrhip_iqbb_i16_set_shift(_plan, design::freqShiftIncrement(_shift, _Fs), 0 > _shift), "BaseBand");
        detail::configCheck(sdrhip_iqbb_i16_reset(_plan, same_cfg ? 3 : 1), "BaseBand");
      }
    }
    if (!reuse) _newPlan(SDRHIP_KEEP_RING | (same_cfg ? SDRHIP_KEEP_FM : 0));
    _buffer.unref();
    _buffer = Buffer<cs16>(buffer_size);
    this->setConfig(out_cfg);
  }

  size_t _planBs = 0;
  int _planEpi = 0;
  double _shift, _Ff, _width, _Fs;
  size_t _order, _sub_sample, _sourceBs;
  int _epilogue, _device;
  sdrhip_iqbb_i16 *_plan;
  Buffer<cs16> _buffer;
};

// =================================================================================================
// FIRLowPass<complex<int16_t>> (bit-exact) / FIRLowPass<complex<float>>
// =================================================================================================
template <class Scalar>
class FIRLowPass : public Sink<Scalar>, public Source {
public:
  FIRLowPass(size_t order, double Fc, int device = 0)
    : _enabled(true), _order(std::max(size_t(1), order)), _Fu(Fc), _Fs(0), _bs(0), _device(device), _plan(0) {}
  virtual ~FIRLowPass() {
    if (_plan) sdrhip_fir_destroy(_plan);
    _buffer.unref();
  }
  inline bool enabled() const { return _enabled; }
  inline void enable(bool enable) { _enabled = enable; }
  inline size_t order() const { return _order; }
  virtual void setOrder(size_t order) { order = std::max(size_t(1), order); if (order == _order) return; _order = order; if (_Fs) _plan_(); }
  inline double freq() const { return _Fu; }
  /** FIRFilter::setUpperFreq (src/firfilter.hh:165-170,287): only the coefficients change; the ring — the stream — goes on. */
  inline void setFreq(double freq) {
    _Fu = freq;
    if (!_Fs) return;
    if (!_plan) { _plan_(); return; }
    std::vector<double> alpha(_order);
    design::firLowPass(_order, _Fu, _Fs, alpha.data());
    detail::configCheck(sdrhip_fir_set_taps(_plan, alpha.data()), "FIRLowPass");
  }

  virtual void config(const Config &src_cfg) {
    if (!src_cfg.hasType() || !src_cfg.hasSampleRate() || !src_cfg.hasBufferSize()) return;
    if (Config::typeId<Scalar>() != src_cfg.type()) {
      ConfigError err;
      err << "Can not configure FIRLowPass: Invalid type " << src_cfg.type() << ", expected " << Config::typeId<Scalar>();
      throw err;
    }
    _Fs = src_cfg.sampleRate();
    _bs = src_cfg.bufferSize();
    _plan_();   // a fresh plan = zeroed ring, as FIRFilter::config does (src/firfilter.hh:193-195)
    if (!_buffer.isEmpty()) _buffer.unref();
    _buffer = Buffer<Scalar>(_bs);
    this->setConfig(Config(src_cfg.type(), src_cfg.sampleRate(), src_cfg.bufferSize(), 1));
  }

  virtual void process(const Buffer<Scalar> &buffer, bool allow_overwrite) {
    if (!_enabled) { this->send(buffer, allow_overwrite); return; }
    if (!_plan) return;
    if (allow_overwrite) _process(buffer, buffer);
    else if (_buffer.isUnused()) _process(buffer, _buffer);
  }

protected:
  void _plan_() {
    std::vector<double> alpha(_order);
    design::firLowPass(_order, _Fu, _Fs, alpha.data());
    if (_plan) { sdrhip_fir_destroy(_plan); _plan = 0; }
    detail::configCheck(sdrhip_fir_create(Device::get(_device), detail::TypeTag<Scalar>::fir, alpha.data(), int(_order), 1, 1,
                                          _bs, SDRHIP_EPI_NONE, &_plan), "FIRLowPass");
  }
  void _process(const Buffer<Scalar> &in, const Buffer<Scalar> &out) {
    size_t n = 0;
    if (!detail::processOk(sdrhip_fir_process(_plan, in.data(), in.size(), 0, out.data(), out.size(), &n), "gpu::FIRLowPass")) return;
    this->send(out.head(in.size()), true);
  }
  bool _enabled;
  size_t _order;
  double _Fu, _Fs;
  size_t _bs;
  int _device;
  sdrhip_fir *_plan;
  Buffer<Scalar> _buffer;
};

// =================================================================================================
// demodulators
// =================================================================================================
namespace detail {
template <class InC, class OutR>
class DemodBase : public Sink<InC>, public Source {
public:
  DemodBase(int kind, const char *name, int device) : _kind(kind), _name(name), _device(device), _plan(0), _can_overwrite(true) {}
  virtual ~DemodBase() {
    if (_plan) sdrhip_demod_destroy(_plan);
    _buffer.unref();
  }
  virtual void config(const Config &src_cfg) {
    if (!src_cfg.hasType() || !src_cfg.hasBufferSize()) return;
    if (Config::typeId<InC>() != src_cfg.type()) {
      ConfigError err;
      err << "Can not configure " << _name << ": Invalid type " << src_cfg.type() << ", expected " << Config::typeId<InC>();
      throw err;
    }
    if (_plan) { sdrhip_demod_destroy(_plan); _plan = 0; }
    configCheck(sdrhip_demod_create(Device::get(_device), _kind, TypeTag<InC>::dtype, 1, src_cfg.bufferSize(), 0, &_plan), _name);
    if (!_buffer.isEmpty()) _buffer.unref();
    _buffer = Buffer<OutR>(src_cfg.bufferSize());
    this->setConfig(Config(Config::typeId<OutR>(), src_cfg.sampleRate(), src_cfg.bufferSize(),
                           _kind == SDRHIP_EPI_AM ? src_cfg.numBuffers() : 1));
  }
  virtual void process(const Buffer<InC> &buffer, bool allow_overwrite) {
    if (!_plan) return;
    if (_kind == SDRHIP_EPI_FM && 0 == buffer.size()) return;            // src/demod.hh:231
    Buffer<OutR> out = (allow_overwrite && _can_overwrite) ? Buffer<OutR>(buffer) : _buffer;
    // in place the output aliases the input bytes, so FM's untouched out[0] is in[0].real()
    if (!processOk(sdrhip_demod_process(_plan, buffer.data(), buffer.size(), 0, out.data(), 0), _name)) return;
    this->send(out.head(buffer.size()), _kind == SDRHIP_EPI_AM);        // AM sends with allow_overwrite=true (:80)
  }

protected:
  int _kind;
  const char *_name;
  int _device;
  sdrhip_demod *_plan;
  bool _can_overwrite;
  Buffer<OutR> _buffer;
};
}  // namespace detail

template <class iScalar, class oScalar = iScalar> class FMDemod;
template <>
class FMDemod<int16_t, int16_t> : public detail::DemodBase<cs16, int16_t> {
public:
  explicit FMDemod(int device = 0) : detail::DemodBase<cs16, int16_t>(SDRHIP_EPI_FM, "FMDemod", device) {}
};
/** FMDemod<int8_t,int16_t> (reference src/demod.hh:173-262 with src/math.hh:12-21): complex<int8_t> in, int16_t out,
 * in place when allowed (an output element covers exactly its input sample). */
template <>
class FMDemod<int8_t, int16_t> : public detail::DemodBase<std::complex<int8_t>, int16_t> {
public:
  explicit FMDemod(int device = 0) : detail::DemodBase<std::complex<int8_t>, int16_t>(SDRHIP_EPI_FM, "FMDemod", device) {}
};
template <class Scalar>
class AMDemod : public detail::DemodBase<std::complex<Scalar>, Scalar> {
public:
  explicit AMDemod(int device = 0) : detail::DemodBase<std::complex<Scalar>, Scalar>(SDRHIP_EPI_AM, "AMDemod", device) {}
};
template <class Scalar>
class USBDemod : public detail::DemodBase<std::complex<Scalar>, Scalar> {
public:
  explicit USBDemod(int device = 0) : detail::DemodBase<std::complex<Scalar>, Scalar>(SDRHIP_EPI_USB, "USBDemod", device) {}
};

/** FMDeemph<int16_t> (reference src/demod.hh:272-362). */
template <class Scalar> class FMDeemph;
template <>
class FMDeemph<int16_t> : public Sink<int16_t>, public Source {
public:
  explicit FMDeemph(bool enabled = true, int device = 0) : _enabled(enabled), _device(device), _plan(0) {}
  virtual ~FMDeemph() {
    if (_plan) sdrhip_deemph_i16_destroy(_plan);
    _buffer.unref();
  }
  inline bool isEnabled() const { return _enabled; }
  inline void enable(bool enabled) { _enabled = enabled; }
  virtual void config(const Config &src_cfg) {
    if (!src_cfg.hasType() || !src_cfg.hasSampleRate() || !src_cfg.hasBufferSize()) return;
    if (Config::typeId<int16_t>() != src_cfg.type()) {
      ConfigError err;
      err << "Can not configure FMDeemph: Invalid type " << src_cfg.type() << ", expected " << Config::typeId<int16_t>();
      throw err;
    }
    if (_plan) { sdrhip_deemph_i16_destroy(_plan); _plan = 0; }   // a fresh plan: average reset to 0 (:308)
    detail::configCheck(sdrhip_deemph_i16_create(Device::get(_device), design::fmDeemphAlpha(src_cfg.sampleRate()), 1,
                                                 src_cfg.bufferSize(), &_plan), "FMDeemph");
    _buffer.unref();
    _buffer = Buffer<int16_t>(src_cfg.bufferSize());
    this->setConfig(Config(src_cfg.type(), src_cfg.sampleRate(), src_cfg.bufferSize(), 1));
  }
  virtual void process(const Buffer<int16_t> &buffer, bool allow_overwrite) {
    if (!_enabled) { this->send(buffer, allow_overwrite); return; }
    if (!_plan) return;
    const Buffer<int16_t> &out = allow_overwrite ? buffer : _buffer;
    if (!detail::processOk(sdrhip_deemph_i16_process(_plan, reinterpret_cast<const int16_t *>(buffer.data()), buffer.size(), 0,
                                                     reinterpret_cast<int16_t *>(out.data()), 0), "gpu::FMDeemph")) return;
    if (allow_overwrite) this->send(buffer, allow_overwrite);
    else this->send(_buffer.head(buffer.size()), false);
  }

protected:
  bool _enabled;
  int _device;
  sdrhip_deemph *_plan;
  Buffer<int16_t> _buffer;
};

// =================================================================================================
// SubSample<complex<int16_t>|complex<float>>
// =================================================================================================
template <class Scalar>
class SubSample : public Sink<Scalar>, public Source {
public:
  explicit SubSample(size_t n, int device = 0) : _n(n), _oFs(0), _device(device), _plan(0) {}
  explicit SubSample(double Fs, int device = 0) : _n(1), _oFs(Fs), _device(device), _plan(0) {}
  virtual ~SubSample() {
    if (_plan) sdrhip_subsample_destroy(_plan);
    _buffer.unref();
  }
  virtual void config(const Config &src_cfg) {
    if (!src_cfg.hasType() || !src_cfg.hasBufferSize()) return;
    if (Config::typeId<Scalar>() != src_cfg.type()) {
      ConfigError err;
      err << "Can not configure SubSample node: Invalid buffer type " << src_cfg.type() << ", expected " << Config::typeId<Scalar>();
      throw err;
    }
    if (_oFs > 0) _n = size_t(std::max(1.0, src_cfg.sampleRate() / _oFs));
    size_t out_size = src_cfg.bufferSize() / _n;
    if (src_cfg.bufferSize() % _n) out_size += 1;
    if (_plan) { sdrhip_subsample_destroy(_plan); _plan = 0; }
    detail::configCheck(sdrhip_subsample_create(Device::get(_device), detail::TypeTag<Scalar>::dtype, _n, 1, src_cfg.bufferSize(), &_plan),
                        "SubSample");
    _buffer.unref();
    _buffer = Buffer<Scalar>(out_size);
    this->setConfig(Config(src_cfg.type(), src_cfg.sampleRate() / _n, out_size, 1));
  }
  virtual void process(const Buffer<Scalar> &buffer, bool allow_overwrite) {
    if (!_plan) return;
    if (allow_overwrite) _process(buffer, buffer);
    else if (_buffer.isUnused()) _process(buffer, _buffer);
  }

protected:
  void _process(const Buffer<Scalar> &in, const Buffer<Scalar> &out) {
    size_t n = 0;
    if (!detail::processOk(sdrhip_subsample_process(_plan, in.data(), in.size(), 0, out.data(), out.size(), &n), "gpu::SubSample")) return;
    this->send(out.head(n), true);
  }
  size_t _n;
  double _oFs;
  int _device;
  sdrhip_subsample *_plan;
  Buffer<Scalar> _buffer;
};

// =================================================================================================
// FilterNode<float|double>: FFT filter bank (one forward transform's worth of input, several band filters)
// =================================================================================================
namespace detail {
/** The complex<float> / complex<double> entry points of the FFT filter behind one set of names. */
template <class Scalar> struct FftConvApi;
template <> struct FftConvApi<float> {
  static int create(sdrhip_ctx *c, int fft, const float *K, int bands, size_t max_in, sdrhip_fftconv **out) {
    return sdrhip_fftconv_create_bank(c, SDRHIP_FFTCONV_OLA, fft, K, 0, bands, 1, max_in, out); }
  static int setKernel(sdrhip_fftconv *h, int band, const float *K) { return sdrhip_fftconv_set_kernel(h, band, K); }
  static int process(sdrhip_fftconv *h, const float *in, size_t n, float *out) { return sdrhip_fftconv_process(h, in, n, 0, out, 0); }
};
template <> struct FftConvApi<double> {
  static int create(sdrhip_ctx *c, int fft, const double *K, int bands, size_t max_in, sdrhip_fftconv **out) {
    return sdrhip_fftconv_f64_create_bank(c, SDRHIP_FFTCONV_OLA, fft, K, 0, bands, 1, max_in, out); }
  static int setKernel(sdrhip_fftconv *h, int band, const double *K) { return sdrhip_fftconv_f64_set_kernel(h, band, K); }
  static int process(sdrhip_fftconv *h, const double *in, size_t n, double *out) { return sdrhip_fftconv_f64_process(h, in, n, 0, out, 0); }
};
}  // namespace detail

/** Drop-in for sdr::FilterNode<Scalar>, Scalar = float or double (reference src/filternode.hh:230-284), ANY block size
 * (:235: `FilterNode(size_t block_size=1024)`; FFTW plans any 2 x block_size): one launch per buffer where the transform
 * fits a workgroup's LDS and is made of the factors 2 ... 13, passes over device memory around a four-step / chirp
 * plan otherwise (csrc/fftany.hpp). */
template <class Scalar>
class FilterNode {
public:
  typedef std::complex<Scalar> CScalar;
  /** One band of the bank: a Source of complex<float> buffers (role of FilterSource, src/filternode.hh:105-227). */
  class Band : public Source {
  public:
    Band(FilterNode *p, size_t index, double fmin, double fmax) : _p(p), _index(index), _fmin(fmin), _fmax(fmax) {}
    virtual ~Band() { _buffer.unref(); }
    /** FilterSource::setFreq (:132-139): only this band's kernel is recomputed; the overlap history goes on. */
    void setFreq(double fmin, double fmax) {
      if (fmax < fmin) std::swap(fmin, fmax);
      _fmin = fmin; _fmax = fmax;
      _p->_bandChanged(_index);
    }
    double fmin() const { return _fmin; }
    double fmax() const { return _fmax; }

  protected:
    friend class FilterNode;
    FilterNode *_p;
    size_t _index;
    double _fmin, _fmax;
    Buffer<CScalar> _buffer;
  };

  explicit FilterNode(size_t block_size = 1024, int device = 0) : _block(block_size), _device(device), _plan(0), _sink(this) {}
  virtual ~FilterNode() {
    if (_plan) sdrhip_fftconv_destroy(_plan);
    for (size_t b = 0; b < _bands.size(); b++) delete _bands[b];
  }

  /** The input of the bank. Unlike the reference (whose BufferNode crashes: SURVEY fact 7) any buffer size is accepted. */
  Sink<CScalar> *sink() { return &_sink; }
  /** Adds a band [fmin, fmax]; the returned Source emits the filtered stream. Adding a band to a configured bank makes
   * a new device plan (the bands' overlap history restarts). */
  Band *addFilter(double fmin, double fmax) {
    if (fmax < fmin) std::swap(fmin, fmax);
    _bands.push_back(new Band(this, _bands.size(), fmin, fmax));
    if (_cfg.hasSampleRate()) _configure(_cfg);
    return _bands.back();
  }

protected:
  void _kernelOf(const Band *b, Scalar *K) const {   // sinc_flt_kernel + FilterSource::_updateFilter (:18-28,186-203)
    std::vector<Scalar> h(2 * _block);
    design::fftFilterKernel(int(_block), b->_fmin, b->_fmax, _cfg.sampleRate(), h.data());
    design::fftFilterSpectrum(int(_block), h.data(), K);
  }
  /** ONE device plan for all bands: one upload and one forward FFT per input block feed every band
   * (FilterSink -> FilterSource fan-out, src/filternode.hh:81-88,257-270). */
  void _configure(const Config &cfg) {
    _cfg = cfg;
    if (_plan) { sdrhip_fftconv_destroy(_plan); _plan = 0; }
    if (_bands.empty()) return;
    std::vector<Scalar> K(4 * _block * _bands.size());
    for (size_t b = 0; b < _bands.size(); b++) _kernelOf(_bands[b], K.data() + b * 4 * _block);
    detail::configCheck(detail::FftConvApi<Scalar>::create(Device::get(_device), int(2 * _block), K.data(), int(_bands.size()),
                                                           cfg.bufferSize(), &_plan), "FFT filter");
    _stage.resize(2 * cfg.bufferSize() * _bands.size());
    for (size_t b = 0; b < _bands.size(); b++) {
      _bands[b]->_buffer.unref();
      _bands[b]->_buffer = Buffer<CScalar>(cfg.bufferSize());
      _bands[b]->setConfig(Config(Config::typeId<CScalar>(), cfg.sampleRate(), cfg.bufferSize(), 1));
    }
  }
  void _bandChanged(size_t index) {
    if (!_plan) return;
    std::vector<Scalar> K(4 * _block);
    _kernelOf(_bands[index], K.data());
    detail::configCheck(detail::FftConvApi<Scalar>::setKernel(_plan, int(index), K.data()), "FFT filter");
  }
  void _run(const Buffer<CScalar> &in) {
    if (!_plan || in.size() * 2 * _bands.size() > _stage.size()) return;
    // a band whose output buffer is still referenced downstream drops this block (src/baseband.hh:141-150 rule); the
    // bank still runs, so that every band's overlap history stays aligned with the input
    if (!detail::processOk(detail::FftConvApi<Scalar>::process(_plan, reinterpret_cast<const Scalar *>(in.data()), in.size(), _stage.data()),
                           "gpu::FilterNode")) return;
    for (size_t b = 0; b < _bands.size(); b++) {
      Band *bd = _bands[b];
      if (!bd->_buffer.isUnused()) continue;
      memcpy(bd->_buffer.data(), _stage.data() + b * 2 * in.size(), in.size() * sizeof(CScalar));
      bd->send(bd->_buffer.head(in.size()), false);
    }
  }

  class In : public Sink<CScalar> {
  public:
    explicit In(FilterNode *p) : _p(p) {}
    virtual void config(const Config &src_cfg) {
      if (Config::Type_UNDEFINED == src_cfg.type() || 0 == src_cfg.sampleRate() || 0 == src_cfg.bufferSize()) return;
      if (Config::typeId<CScalar>() != src_cfg.type()) {
        ConfigError err;
        err << "Can not configure filter-sink: Invalid type " << src_cfg.type() << ", expected " << Config::typeId<CScalar>();
        throw err;
      }
      _p->_configure(src_cfg);
    }
    virtual void process(const Buffer<CScalar> &buffer, bool) { _p->_run(buffer); }
    FilterNode *_p;
  };
  friend class Band;
  size_t _block;
  int _device;
  Config _cfg;
  sdrhip_fftconv *_plan;
  In _sink;
  std::vector<Band *> _bands;
  std::vector<Scalar> _stage;
};

// =================================================================================================
// ChannelBank<int16_t>: C independent IQBaseBand<int16_t>(+demod) channels, one batched launch per device
// =================================================================================================
template <class Scalar> class ChannelBank;

/** Many independent channels behind one port per channel — sink(c) / source(c), the reference's own pattern for
 * multi-input nodes (Combine::sink(i), src/combine.hh:66-150). The channels are split into contiguous blocks over
 * the given devices (one rank each, sdrhip_comm_*): every device filters its block in one batched launch, no data
 * path collective; the demodulated rows are gathered on rank 0's device (RCCL over xGMI when the devices differ)
 * and leave in one device-to-host copy — BASELINE config 5's shape, driven from C++ in one process. The staging
 * buffers are pinned (registered), so each device's copies are plain DMA on its own stream and run beside the
 * other devices' kernels. */
template <>
class ChannelBank<int16_t> {
public:
  class Out : public Source {
  public:
    void configure(const Config &c) { this->setConfig(c); }
    void emit(const RawBuffer &b, bool aw) { this->send(b, aw); }
  };

  /** All channels share the band-select parameters (taps / LUT are read-only data, designed once on the host). */
  ChannelBank(size_t channels, double Fc, double Ff, double width, size_t order, size_t sub_sample, int epilogue = SDRHIP_EPI_NONE,
              int device = 0)
    : _C(channels), _Fc(Fc), _Ff(Ff), _width(width), _order(std::max(size_t(1), order)), _D(sub_sample), _epilogue(epilogue),
      _devices(1, device), _comm(0), _gather(0), _bs(0), _have(0), _ins(channels, In(this)), _outs(channels), _pending(channels, false) {
    for (size_t c = 0; c < _C; c++) _ins[c]._index = c;
  }
  /** The same bank over several devices: rank r (device devices[r]) owns the r-th contiguous block of channels. */
  ChannelBank(size_t channels, double Fc, double Ff, double width, size_t order, size_t sub_sample, int epilogue,
              const std::vector<int> &devices)
    : _C(channels), _Fc(Fc), _Ff(Ff), _width(width), _order(std::max(size_t(1), order)), _D(sub_sample), _epilogue(epilogue),
      _devices(devices.empty() ? std::vector<int>(1, 0) : devices), _comm(0), _gather(0), _bs(0), _have(0), _ins(channels, In(this)),
      _outs(channels), _pending(channels, false) {
    for (size_t c = 0; c < _C; c++) _ins[c]._index = c;
  }
  virtual ~ChannelBank() {
    _release();
    _stageOut.unref();
    _stageIn.unref();
  }
  Sink<cs16> *sink(size_t c) { return &_ins[c]; }
  Source *source(size_t c) { return &_outs[c]; }
  size_t channels() const { return _C; }
  size_t ranks() const { return _devices.size(); }
  /** "rccl" or "same-device copies" once configured (sdrhip_comm_transport). */
  const char *transport() const { const char *n = ""; if (_comm) sdrhip_comm_transport(_comm, &n); return n; }

protected:
  class In : public Sink<cs16> {
  public:
    explicit In(ChannelBank *p) : _p(p), _index(0) {}
    virtual void config(const Config &cfg) { _p->_config(cfg); }
    virtual void process(const Buffer<cs16> &b, bool) { _p->_deliver(_index, b); }
    ChannelBank *_p;
    size_t _index;
  };
  struct Rank {
    sdrhip_ctx *ctx; sdrhip_iqbb_i16 *plan; size_t c0, c1; void *din, *dout;
    Rank() : ctx(0), plan(0), c0(0), c1(0), din(0), dout(0) {}
  };

  void _release() {
    if (_comm) sdrhip_comm_synchronize(_comm);
    for (size_t r = 0; r < _ranks.size(); r++) {
      if (_ranks[r].plan) sdrhip_iqbb_i16_destroy(_ranks[r].plan);
      if (_ranks[r].din) sdrhip_free(_ranks[r].ctx, _ranks[r].din);
      if (_ranks[r].dout) sdrhip_free(_ranks[r].ctx, _ranks[r].dout);
    }
    if (_gather) sdrhip_free(_ranks[0].ctx, _gather);
    _gather = 0;
    _ranks.clear();
    if (!_stageIn.isEmpty()) sdrhip_host_unregister(_stageIn.data());
    if (!_stageOut.isEmpty()) sdrhip_host_unregister(_stageOut.data());
    if (_comm) { sdrhip_comm_destroy(_comm); _comm = 0; }
  }

  void _config(const Config &cfg) {
    if (!cfg.hasType() || !cfg.hasSampleRate() || !cfg.hasBufferSize()) return;
    if (Config::typeId<cs16>() != cfg.type()) {
      ConfigError err;
      err << "Can not configure ChannelBank: Invalid type " << cfg.type() << ", expected " << Config::typeId<cs16>();
      throw err;
    }
    if (_comm && cfg == _cfg) return;   // every channel's source pushes the same Config
    _cfg = cfg;
    _bs = cfg.bufferSize();
    const int32_t Fs = int32_t(cfg.sampleRate());
    std::vector<int32_t> taps(2 * _order), lut(2 * design::kLutSize);
    design::iqbbTaps(_Ff, _width, Fs, _order, taps.data());
    design::freqShiftLutI16(lut.data());
    _release();
    _stageIn.unref(); _stageOut.unref();
    _outStride = _bs / _D + 2;
    _stageIn = Buffer<cs16>(_C * _bs);
    _stageOut = Buffer<cs16>(_C * _outStride);
    detail::configCheck(sdrhip_host_register(_stageIn.data(), _C * _bs * sizeof(cs16)), "ChannelBank");
    detail::configCheck(sdrhip_host_register(_stageOut.data(), _C * _outStride * sizeof(cs16)), "ChannelBank");
    const size_t R = std::min(_devices.size(), _C);
    detail::configCheck(sdrhip_comm_create(_devices.data(), int(R), &_comm), "ChannelBank");
    _ranks.assign(R, Rank());
    for (size_t r = 0; r < R; r++) {   // contiguous blocks, sizes differ by at most one
      Rank &k = _ranks[r];
      k.c0 = r * (_C / R) + std::min(r, _C % R);
      k.c1 = k.c0 + _C / R + (r < _C % R ? 1 : 0);
      detail::configCheck(sdrhip_comm_ctx(_comm, int(r), &k.ctx), "ChannelBank");
      detail::configCheck(sdrhip_iqbb_i16_create(k.ctx, taps.data(), int(_order), lut.data(), design::freqShiftIncrement(_Fc, double(Fs)),
                                                 0 > _Fc, int(_D), int(k.c1 - k.c0), _bs, _epilogue, &k.plan), "ChannelBank");
      detail::configCheck(sdrhip_malloc(k.ctx, (k.c1 - k.c0) * _bs * sizeof(cs16), &k.din), "ChannelBank");
      detail::configCheck(sdrhip_malloc(k.ctx, (k.c1 - k.c0) * _outStride * sizeof(cs16), &k.dout), "ChannelBank");
    }
    if (R > 1) detail::configCheck(sdrhip_malloc(_ranks[0].ctx, _C * _outStride * sizeof(cs16), &_gather), "ChannelBank");
    std::fill(_pending.begin(), _pending.end(), false);
    _chunkHave.assign((_C + kChunk - 1) / kChunk, 0);
    _copyOk = true;
    _have = 0;
    const double oRate = double(size_t(Fs) / _D);
    for (size_t c = 0; c < _C; c++)
      _outs[c].configure(Config(_epilogue == SDRHIP_EPI_NONE ? Config::typeId<cs16>() : Config::typeId<int16_t>(), oRate, _outStride, 1));
  }

  /** Collects one buffer per channel (all of the same length: buffer boundaries are part of the
   * numerical contract), then launches once per device for the whole bank. */
  void _deliver(size_t c, const Buffer<cs16> &b) {
    if (!_comm || b.size() > _bs) return;
    if (_have == 0) { _len = b.size(); _copyOk = true; }
    if (_pending[c] || b.size() != _len) {
      LogMessage msg(LOG_WARNING);
      msg << "gpu::ChannelBank: channel " << c << " delivered out of step; buffer dropped";
      Logger::get().log(msg);
      return;
    }
    memcpy(_stageIn.data() + c * _bs * sizeof(cs16), b.data(), b.size() * sizeof(cs16));
    _pending[c] = true;
    // The round's input travels while the round is still being delivered: the channels come one buffer at a time from the
    // caller's thread (a 256 KB memcpy each into the pinned staging area: 10 ms for 1024 channels on one core), so every
    // completed chunk of kChunk channels starts its H2D copy at once and the copies hide behind the memcpys of the chunks
    // that follow (measured through examples/bench_graph.cc: profiles/r18_host_path.txt)
    const size_t k = c / kChunk;
    if (++_chunkHave[k] == std::min<size_t>(kChunk, _C - k * kChunk)) _copyChunk(k);
    if (++_have < _C) return;
    _have = 0;
    std::fill(_pending.begin(), _pending.end(), false);
    std::fill(_chunkHave.begin(), _chunkHave.end(), 0);
    // the per-channel outputs are views of _stageOut: while a consumer (e.g. a queued edge) still holds one of the
    // last round, this round is dropped, as every node drops its input while its output buffer is in use
    // (src/baseband.hh:141-150)
    if (!_stageOut.isUnused()) {
      LogMessage msg(LOG_WARNING);
      msg << "gpu::ChannelBank: output of the last round still in use downstream; round dropped";
      Logger::get().log(msg);
      return;
    }
    size_t n = 0;
    const size_t per = _epilogue == SDRHIP_EPI_NONE ? 1 : 2;   // int16 elements fit twice into a cs16 row
    const size_t R = _ranks.size(), rowB = _outStride * sizeof(cs16);
    std::vector<const void *> send(R); std::vector<size_t> bytes(R);
    bool ok = _copyOk;
    for (size_t r = 0; r < R && ok; r++) {   // every rank: its batched launch behind its chunks' H2D copies (same stream), all asynchronous
      Rank &k = _ranks[r];
      ok = detail::processOk(sdrhip_iqbb_i16_process_dev(k.plan, reinterpret_cast<const int16_t *>(k.din), _len, _bs, k.dout,
                                                         _outStride * per, &n), "gpu::ChannelBank");
      send[r] = k.dout; bytes[r] = (k.c1 - k.c0) * rowB;
    }
    if (ok && R > 1)   // rows gathered on rank 0's device in channel order (RCCL over xGMI), then one copy to the host
      ok = detail::processOk(sdrhip_comm_gather(_comm, send.data(), bytes.data(), _gather, 0), "gpu::ChannelBank") &&
           detail::processOk(sdrhip_memcpy_d2h_async(_ranks[0].ctx, _stageOut.data(), _gather, _C * rowB), "gpu::ChannelBank");
    else if (ok)
      ok = detail::processOk(sdrhip_memcpy_d2h_async(_ranks[0].ctx, _stageOut.data(), _ranks[0].dout, _C * rowB), "gpu::ChannelBank");
    if (!detail::processOk(sdrhip_comm_synchronize(_comm), "gpu::ChannelBank") || !ok) return;
    for (size_t ch = 0; ch < _C; ch++) {
      if (_epilogue == SDRHIP_EPI_NONE) _outs[ch].emit(_stageOut.sub(ch * _outStride, n), false);
      else if (!(_epilogue == SDRHIP_EPI_FM && n == 0))
        _outs[ch].emit(Buffer<int16_t>(_stageOut).sub(ch * _outStride * 2, n), false);
    }
  }

  /** H2D copy of the channels [k * kChunk, (k + 1) * kChunk) of the round being collected: each rank's part on that rank's stream. */
  void _copyChunk(size_t k) {
    const size_t a = k * kChunk, b = std::min<size_t>(_C, a + kChunk);
    for (size_t r = 0; r < _ranks.size(); r++) {
      Rank &rk = _ranks[r];
      const size_t lo = std::max(a, rk.c0), hi = std::min(b, rk.c1);
      if (lo >= hi) continue;
      if (!detail::processOk(sdrhip_memcpy_h2d_async(rk.ctx, static_cast<char *>(rk.din) + (lo - rk.c0) * _bs * sizeof(cs16),
                                                     _stageIn.data() + lo * _bs * sizeof(cs16), (hi - lo) * _bs * sizeof(cs16)), "gpu::ChannelBank"))
        _copyOk = false;
    }
  }

  enum { kChunk = 32 };   // channels per H2D chunk (an enumerator: no out-of-class definition needed when bound to a reference)
  std::vector<size_t> _chunkHave;
  bool _copyOk = true;
  size_t _C;
  double _Fc, _Ff, _width;
  size_t _order, _D;
  int _epilogue;
  std::vector<int> _devices;
  sdrhip_comm *_comm;
  std::vector<Rank> _ranks;
  void *_gather;
  Config _cfg;
  size_t _bs, _have, _len, _outStride;
  std::vector<In> _ins;
  std::vector<Out> _outs;
  std::vector<bool> _pending;
  Buffer<cs16> _stageIn, _stageOut;
};

// =================================================================================================
// FFT::exec / FFTPlan<float|double> on host buffers (reference src/fftplan.hh:14-36, src/fftplan_fftw3.hh:12-142)
// =================================================================================================
/** The reference's FFT module: the same Direction enum, exec<Scalar>(in, out, dir) and exec<Scalar>(inplace, dir). */
template <class Scalar> class FFTPlan;

class FFT {
public:
  typedef enum { FORWARD, BACKWARD } Direction;
  template <class Scalar>
  static void exec(const Buffer< std::complex<Scalar> > &in, const Buffer< std::complex<Scalar> > &out, FFT::Direction dir, int device = 0) {
    FFTPlan<Scalar> plan(in, out, dir, device); plan();
  }
  template <class Scalar>
  static void exec(const Buffer< std::complex<Scalar> > &inplace, FFT::Direction dir, int device = 0) {
    FFTPlan<Scalar> plan(inplace, dir, device); plan();
  }
};

/** FFTPlan<float> and FFTPlan<double>: same constructors and error texts as the FFTW-backed reference classes; the
 * transform is the library's own (unnormalised either way, like FFTW), for ANY size. As in the reference the plan is made
 * ONCE, in the constructor (src/fftplan_fftw3.hh:34-36,52-54: fftw_plan_dft_1d) — a size the device cannot plan is a
 * ConfigError there — and operator() only executes it (:59); the destructor frees it (:64). */
template <class Scalar>
class FFTPlan {
public:
  FFTPlan(const Buffer< std::complex<Scalar> > &in, const Buffer< std::complex<Scalar> > &out, FFT::Direction dir, int device = 0)
    : _in(in), _out(out), _sign(dir == FFT::BACKWARD ? 1 : -1), _device(device), _plan(0) {
    if (in.size() != out.size()) {
      ConfigError err;
      err << "Can not construct FFT plan: input & output buffers are of different size!";
      throw err;
    }
    if (in.isEmpty() || out.isEmpty()) {
      ConfigError err;
      err << "Can not construct FFT plan: input or output buffer is empty!";
      throw err;
    }
    _make();
  }
  FFTPlan(const Buffer< std::complex<Scalar> > &inplace, FFT::Direction dir, int device = 0)
    : _in(inplace), _out(inplace), _sign(dir == FFT::BACKWARD ? 1 : -1), _device(device), _plan(0) {
    if (inplace.isEmpty()) {
      ConfigError err;
      err << "Can not construct FFT plan: Buffer is empty!";
      throw err;
    }
    _make();
  }
  virtual ~FFTPlan() { if (_plan) sdrhip_fft_plan_destroy(_plan); }
  /** Performs the transformation. */
  void operator() () {
    detail::configCheck(sdrhip_fft_plan_exec(_plan, _sign, _in.data(), _out.data()), "FFT plan");
  }
  /** Which device plan serves this size ("radix-16 lds", "lds", "four-step", "chirp", ...). */
  const char *form() const { const char *s = ""; sdrhip_fft_plan_form(_plan, &s); return s; }

protected:
  static int _dtype() { return sizeof(Scalar) == 8 ? SDRHIP_T_CF64 : SDRHIP_T_CF32; }
  void _make() {
    if (_in.size() > (size_t(1) << 27)) {
      ConfigError err;
      err << "Can not construct FFT plan: " << _in.size() << " points exceed the device plans (2^27)";
      throw err;
    }
    detail::configCheck(sdrhip_fft_plan_create(Device::get(_device), _dtype(), int(_in.size()), &_plan), "FFT plan");
  }
  Buffer< std::complex<Scalar> > _in, _out;
  int _sign, _device;
  sdrhip_fft_plan *_plan;

private:
  FFTPlan(const FFTPlan &);              // (owns a device plan)
  FFTPlan &operator=(const FFTPlan &);
};

}  // namespace gpu
}  // namespace sdr

#endif
