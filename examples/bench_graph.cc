// bench_graph.cc — the host path THROUGH THE DROP-IN GRAPH, measured from C++: C sources -> sdr::gpu::ChannelBank<int16_t>
// (IQBaseBand<int16>(127 taps, /8) -> FMDemod fused, one batched launch per round) -> C counting sinks, wired with the
// sdr:: core's own Source / Sink / connect() exactly as a reference graph is (src/node.cc:66-84: a direct edge hands
// the buffer to the sink in the caller's thread; Combine-style per-channel ports, src/combine.hh:66-150).
//
// What a round costs here is what a host that hands over HOST buffers pays: C x memcpy into the bank's pinned staging
// area, one H2D copy, the kernel, one D2H copy, C x send() to the sinks. It is the PCIe-inclusive figure (never bench.py's
// `value`, which is device-resident). The signal is synthesised once by IQSigGen (the reference's generator evaluates two
// complex exponentials per sample: replaying its buffer keeps the generator out of the measurement).
//
//   bench_graph [rounds [samples_per_buffer]]      prints one line per channel count (1, 64, 1024) and a JSON summary
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "sdr/sdr.hh"
#include "sdr/gpu/nodes.hh"

using namespace sdr;
typedef std::complex<int16_t> cs16;

// a source that sends views of one pre-generated buffer (what a driver thread handing over its DMA buffer does)
class Replay : public Source {
public:
  void configure(double Fs, size_t n) { this->setConfig(Config(Config::typeId<cs16>(), Fs, n, 1)); }
  void feed(cs16 *p, size_t n) { Buffer<cs16> view(p, n); this->send(view, false); }
};

// catches IQSigGen's buffer once
class Grab : public Sink<cs16> {
public:
  std::vector<cs16> data;
  virtual void config(const Config &) {}
  virtual void process(const Buffer<cs16> &b, bool) { data.assign(&b[0], &b[0] + b.size()); }
};

class Counter : public Sink<int16_t> {
public:
  size_t samples = 0, buffers = 0; long sum = 0;
  virtual void config(const Config &) {}
  virtual void process(const Buffer<int16_t> &b, bool) { samples += b.size(); buffers++; if (b.size() > 1) sum += b[b.size() - 1]; }
};

int main(int argc, char **argv) {
  const size_t rounds_arg = argc > 1 ? (size_t)atol(argv[1]) : 0, N = argc > 2 ? (size_t)atol(argv[2]) : 65536;
  const double Fs = 2.4e6;
  IQSigGen<int16_t> gen(Fs, N);
  gen.addSine(100e3, 8000, 0.0); gen.addSine(-300e3, 6000, 0.3);
  Grab grab; gen.connect(&grab, true);
  gen.next();
  if (grab.data.size() != N) { std::fprintf(stderr, "bench_graph: the generator delivered %zu samples\n", grab.data.size()); return 2; }
  const size_t Cs[3] = {1, 64, 1024};
  double msps[3] = {0, 0, 0}, ms[3] = {0, 0, 0};
  for (int k = 0; k < 3; k++) {
    const size_t C = Cs[k];
    gpu::ChannelBank<int16_t> bank(C, 100e3, 100e3, 50e3, 127, 8, SDRHIP_EPI_FM);
    std::vector<Replay> src(C); std::vector<Counter> cnt(C);
    try {
      for (size_t c = 0; c < C; c++) {
        src[c].connect(bank.sink(c), true); bank.source(c)->connect(&cnt[c], true);
        src[c].configure(Fs, N);
      }
    } catch (std::exception &e) { std::fprintf(stderr, "bench_graph: %s\n", e.what()); return 3; }
    const size_t rounds = rounds_arg ? rounds_arg : (C == 1 ? 400 : C == 64 ? 60 : 12);
    for (int w = 0; w < 3; w++) for (size_t c = 0; c < C; c++) src[c].feed(grab.data.data(), N);   // warm-up (plans, pinned pages)
    const size_t before = cnt[0].buffers;
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t r = 0; r < rounds; r++) for (size_t c = 0; c < C; c++) src[c].feed(grab.data.data(), N);
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (cnt[0].buffers - before != rounds || cnt[C - 1].samples == 0) {
      std::fprintf(stderr, "bench_graph: C=%zu delivered %zu of %zu rounds (no MI355X? sdrhip: %s)\n", C, cnt[0].buffers - before, rounds, sdrhip_last_error());
      return 4;
    }
    ms[k] = sec / rounds * 1e3; msps[k] = double(C) * N * rounds / sec / 1e6;
    std::printf("ChannelBank<int16>(127,/8)->FM through the graph: C=%4zu  %8.3f ms per round  %9.1f MS/s  (%zu rounds of %zu samples per channel; "
                "real time: %.1f ms of signal per buffer)\n", C, ms[k], msps[k], rounds, N, N / Fs * 1e3);
  }
  std::printf("{\"bench_graph\": {\"samples_per_buffer\": %zu, \"c1_ms\": %.4f, \"c1_msps\": %.1f, \"c64_ms\": %.4f, \"c64_msps\": %.1f, "
              "\"c1024_ms\": %.4f, \"c1024_msps\": %.1f}}\n", N, ms[0], msps[0], ms[1], msps[1], ms[2], msps[2]);
  return 0;
}
