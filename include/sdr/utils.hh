// utils.hh — small helper sinks of the API-compatible sdr:: core (own code).
#ifndef SDR_CORE_UTILS_HH
#define SDR_CORE_UTILS_HH

#include <cstring>
#include <vector>

#include "node.hh"

namespace sdr {

/** Keeps a copy of the last buffer received (role of the reference's DebugStore, src/utils.hh:799-841). */
template <class Scalar>
class DebugStore : public Sink<Scalar> {
public:
  DebugStore() : Sink<Scalar>() {}
  virtual ~DebugStore() { _buffer.unref(); }
  virtual void config(const Config &src_cfg) {
    if (!src_cfg.hasType() || !src_cfg.hasBufferSize()) return;
    if (Config::typeId<Scalar>() != src_cfg.type()) {
      ConfigError err;
      err << "Can not configure DebugStore node: Invalid input type " << src_cfg.type() << ", expected " << Config::typeId<Scalar>();
      throw err;
    }
    _buffer.unref();
    _buffer = Buffer<Scalar>(src_cfg.bufferSize());
  }
  virtual void process(const Buffer<Scalar> &buffer, bool) {
    const size_t n = std::min(buffer.size(), _buffer.size());
    memcpy(_buffer.ptr(), buffer.data(), n * sizeof(Scalar));
    _view = _buffer.head(n);
  }
  inline const Buffer<Scalar> &buffer() const { return _view; }
  inline void clear() { _view = Buffer<Scalar>(); }

protected:
  Buffer<Scalar> _buffer, _view;
};

/** Appends every received buffer to a vector (test harness). */
template <class Scalar>
class Recorder : public Sink<Scalar> {
public:
  std::vector<Scalar> data;
  std::vector<size_t> lens;
  virtual void config(const Config &) {}
  virtual void process(const Buffer<Scalar> &b, bool) {
    lens.push_back(b.size());
    for (size_t i = 0; i < b.size(); i++) data.push_back(b[i]);
  }
};

}  // namespace sdr
#endif
