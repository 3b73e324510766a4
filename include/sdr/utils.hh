// utils.hh — the capture sink the tests and examples record through (own code). The reference's helper nodes
// (src/utils.hh: DebugStore, Scale, AGC, ...) are out of scope and have no counterpart here.
#ifndef SDR_CORE_UTILS_HH
#define SDR_CORE_UTILS_HH

#include <vector>

#include "node.hh"

namespace sdr {

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
