// siggen.hh — IQSigGen<Scalar>: sum-of-sines test source of the API-compatible sdr:: core (own code).
// Behaviour restated from the reference (src/siggen.hh:90-157) and pinned by golden vectors
// (tests: the C++ harness compares its output with tests/golden/g1_iq_cs16.bin):
//   every sine contributes scale*A*exp(i(2 pi f t + phi))/nsines, ADDED THROUGH the sample type (so
//   integer types truncate after each sine), t advances by 1/Fs accumulated in double, and scale is
//   1 for every IQSigGen<T> (amplitudes are in sample units).
#ifndef SDR_CORE_SIGGEN_HH
#define SDR_CORE_SIGGEN_HH

#include <cmath>
#include <vector>

#include "node.hh"

namespace sdr {

template <class Scalar>
class IQSigGen : public Source {
public:
  IQSigGen(double samplerate, size_t buffersize, double tmax = -1)
    : Source(), _dt(1. / samplerate), _t(0), _tMax(tmax), _bufferSize(buffersize), _buffer(buffersize) {
    this->setConfig(Config(Config::typeId< std::complex<Scalar> >(), samplerate, buffersize, 1));
  }
  virtual ~IQSigGen() { _buffer.unref(); }

  void addSine(double freq, double ampl = 1, double phase = 0) { _tones.push_back(Tone{freq, ampl, phase}); }

  /** Produces and sends the next buffer; hook it to Queue::addIdle(). Stops the queue after tmax. */
  void next() {
    if (_tMax > 0 && _t >= _tMax) { Queue::get().stop(); return; }
    const double n = double(_tones.size()), scale = 1;
    for (size_t i = 0; i < _bufferSize; i++) {
      Scalar re = 0, im = 0;
      for (size_t s = 0; s < _tones.size(); s++) {
        const Tone &k = _tones[s];
        const std::complex<double> v = (scale * (k.a * std::exp(std::complex<double>(0, 2 * M_PI * k.f * _t + k.p)))) / n;
        re = narrow(double(re) + v.real());
        im = narrow(double(im) + v.imag());
      }
      _buffer[i] = std::complex<Scalar>(re, im);
      _t += _dt;
    }
    this->send(_buffer);
  }

protected:
  struct Tone { double f, a, p; };
  static inline Scalar narrow(double d) { return Scalar(d); }
  double _dt, _t, _tMax;
  std::vector<Tone> _tones;
  size_t _bufferSize;
  Buffer< std::complex<Scalar> > _buffer;
};

template <> inline int16_t IQSigGen<int16_t>::narrow(double d) { return int16_t(int32_t(d)); }

}  // namespace sdr
#endif
