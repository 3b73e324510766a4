// node.hh — Config / Sink / Source of the API-compatible sdr:: core (own code).
//
// This is the drop-in boundary of the hot path: a node is a Sink<T> and/or a Source.
// Rules kept from the reference (src/node.hh:35-258, src/node.cc):
//   - Source::connect(sink, direct) records the edge and immediately pushes the current Config to
//     the sink; setConfig() re-propagates only when the Config changed;
//   - sinks are kept in a std::map keyed by pointer, so delivery order is pointer order;
//   - send(buffer, allow_overwrite): a sink may overwrite the buffer only if the sender allows it
//     AND it is the only sink; direct edges are called synchronously, others go through the Queue;
//   - Sink<T>::handleBuffer re-types the raw view and calls process().
#ifndef SDR_CORE_NODE_HH
#define SDR_CORE_NODE_HH

#include <complex>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <stdint.h>
#include <thread>

#include "buffer.hh"
#include "exception.hh"
#include "queue.hh"

namespace sdr {

class Config {
public:
  typedef enum {
    Type_UNDEFINED = 0,
    Type_u8, Type_s8, Type_u16, Type_s16, Type_f32, Type_f64,
    Type_cu8, Type_cs8, Type_cu16, Type_cs16, Type_cf32, Type_cf64
  } Type;

  Config() : _type(Type_UNDEFINED), _sampleRate(0), _bufferSize(0), _numBuffers(0) {}
  Config(Type type, double sampleRate, size_t bufferSize, size_t numBuffers)
    : _type(type), _sampleRate(sampleRate), _bufferSize(bufferSize), _numBuffers(numBuffers) {}

  bool operator==(const Config &o) const {
    return o._type == _type && o._sampleRate == _sampleRate && o._bufferSize == _bufferSize && o._numBuffers == _numBuffers;
  }

  // a field still at its zero value has not been decided upstream yet (nodes wait for the fields they need)
  bool hasType() const { return _type != Type_UNDEFINED; }
  bool hasSampleRate() const { return _sampleRate != 0.0; }
  bool hasBufferSize() const { return _bufferSize > 0; }
  bool hasNumBuffers() const { return _numBuffers > 0; }

  Type type() const { return _type; }
  double sampleRate() const { return _sampleRate; }
  size_t bufferSize() const { return _bufferSize; }
  size_t numBuffers() const { return _numBuffers; }

  void setType(Type t) { _type = t; }
  void setSampleRate(double hz) { _sampleRate = hz; }
  void setBufferSize(size_t samples) { _bufferSize = samples; }
  void setNumBuffers(size_t count) { _numBuffers = count; }

  template <typename T> static inline Type typeId();

protected:
  Type _type;
  double _sampleRate;
  size_t _bufferSize, _numBuffers;
};

template <> inline Config::Type Config::typeId<uint8_t>() { return Type_u8; }
template <> inline Config::Type Config::typeId<int8_t>() { return Type_s8; }
template <> inline Config::Type Config::typeId<uint16_t>() { return Type_u16; }
template <> inline Config::Type Config::typeId<int16_t>() { return Type_s16; }
template <> inline Config::Type Config::typeId<float>() { return Type_f32; }
template <> inline Config::Type Config::typeId<double>() { return Type_f64; }
template <> inline Config::Type Config::typeId< std::complex<uint8_t> >() { return Type_cu8; }
template <> inline Config::Type Config::typeId< std::complex<int8_t> >() { return Type_cs8; }
template <> inline Config::Type Config::typeId< std::complex<uint16_t> >() { return Type_cu16; }
template <> inline Config::Type Config::typeId< std::complex<int16_t> >() { return Type_cs16; }
template <> inline Config::Type Config::typeId< std::complex<float> >() { return Type_cf32; }
template <> inline Config::Type Config::typeId< std::complex<double> >() { return Type_cf64; }

inline const char *typeName(Config::Type type) {
  static const char *names[] = {"UNDEFINED", "uint8", "int8", "uint16", "int16", "float", "double", "complex uint8",
                                "complex int8", "complex uint16", "complex int16", "complex float", "complex double"};
  return (int(type) >= 0 && int(type) <= int(Config::Type_cf64)) ? names[int(type)] : "unknown";
}

inline std::ostream &operator<<(std::ostream &stream, Config::Type type) {
  stream << typeName(type) << " (" << (int)type << ")";
  return stream;
}

class SinkBase {
public:
  SinkBase() {}
  virtual ~SinkBase() {}
  virtual void handleBuffer(const RawBuffer &buffer, bool allow_overwrite) = 0;
  virtual void config(const Config &src_cfg) = 0;
};

namespace detail {
inline void deliver(SinkBase *sink, const RawBuffer &buffer, bool allow_overwrite) { sink->handleBuffer(buffer, allow_overwrite); }
}

template <class Scalar>
class Sink : public SinkBase {
public:
  Sink() : SinkBase() {}
  virtual ~Sink() {}
  virtual void process(const Buffer<Scalar> &buffer, bool allow_overwrite) = 0;
  virtual void handleBuffer(const RawBuffer &buffer, bool allow_overwrite) {
    this->process(Buffer<Scalar>(buffer), allow_overwrite);
  }
};

class Source {
public:
  Source() {}
  virtual ~Source() {}

  virtual void send(const RawBuffer &buffer, bool allow_overwrite = false) {
    const bool exclusive = allow_overwrite && (1 == _sinks.size());
    for (std::map<SinkBase *, bool>::iterator it = _sinks.begin(); it != _sinks.end(); ++it) {
      if (it->second) it->first->handleBuffer(buffer, exclusive);
      else Queue::get().send(buffer, it->first, exclusive);
    }
  }

  void connect(SinkBase *sink, bool direct = false) {
    _sinks[sink] = direct;
    sink->config(_config);
  }
  void disconnect(SinkBase *sink) { _sinks.erase(sink); }

  virtual void setConfig(const Config &config) {
    if (config == _config) return;
    _config = config;
    propagateConfig(_config);
  }

  virtual double sampleRate() const { return _config.sampleRate(); }
  virtual Config::Type type() const { return _config.type(); }

  template <class T>
  void addEOS(T *instance, void (T::*function)()) { _eos.emplace_back(new Delegate<T>(instance, function)); }

protected:
  void signalEOS() { for (auto &d : _eos) (*d)(); }
  void propagateConfig(const Config &) {
    for (std::map<SinkBase *, bool>::iterator it = _sinks.begin(); it != _sinks.end(); ++it) it->first->config(_config);
  }

  Config _config;
  std::map<SinkBase *, bool> _sinks;
  std::list< std::shared_ptr<DelegateInterface> > _eos;
};

/** A source that blocks in next() (device / file input); optionally runs in its own thread. */
class BlockingSource : public Source {
public:
  BlockingSource(bool parallel = false, bool connect_idle = true, bool stop_queue_on_eos = false)
    : Source(), _is_active(false), _is_parallel(parallel) {
    if (!parallel && connect_idle) Queue::get().addIdle(this, &BlockingSource::idleCallback);
    if (stop_queue_on_eos) this->addEOS(&Queue::get(), &Queue::stop);
  }
  virtual ~BlockingSource() {
    Queue::get().remIdle(this);
    if (isActive()) stop();
    if (_thread.joinable()) _thread.join();
  }
  virtual void next() = 0;
  inline bool isActive() const { return _is_active; }
  virtual void start() {
    if (_is_active) return;
    _is_active = true;   // (the reference leaves this to subclasses: SURVEY Appendix A)
    if (_is_parallel) _thread = std::thread([this] { while (_is_active && Queue::get().isRunning()) this->next(); });
  }
  virtual void stop() {
    if (!_is_active) return;
    _is_active = false;
    if (_is_parallel && _thread.joinable()) _thread.join();
  }

protected:
  void idleCallback() { if (_is_active && Queue::get().isRunning()) this->next(); }
  volatile bool _is_active;
  bool _is_parallel;
  std::thread _thread;
};

/** Pass-through node. */
class Proxy : public SinkBase, public Source {
public:
  Proxy() {}
  virtual ~Proxy() {}
  virtual void config(const Config &src_cfg) { this->setConfig(src_cfg); }
  virtual void handleBuffer(const RawBuffer &buffer, bool) { this->send(buffer); }
};

}  // namespace sdr
#endif
