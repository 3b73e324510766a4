// dropin_ref.cc — drop-in proof: the UNMODIFIED reference runtime (its Source/Sink/Buffer/Queue,
// IQSigGen, FMDemod, compiled from /root/reference/src by oracle/Makefile) drives this repository's
// sdr::gpu nodes, and the result is compared with the all-reference CPU graph.
//   reference IQSigGen -> sdr::gpu::IQBaseBand<int16_t> -> reference FMDemod<int16_t>   (direct edges)
//   reference IQSigGen -> [Queue] -> sdr::gpu::FIRLowPass<cs16> -> sdr::gpu::FMDemod     (config-1 plumbing)
// Test infrastructure (built only where /root/reference exists; the binary travels to the GPU box).
#include "sdr.hh"                 // the reference's umbrella header
#include "sdr/gpu/nodes.hh"       // our nodes, compiled against the reference core

#include <cstdio>
#include <cstring>
#include <vector>

using namespace sdr;
typedef std::complex<int16_t> cs16;

template <class T>
class Rec : public Sink<T> {
public:
  std::vector<T> data;
  virtual void config(const Config &) {}
  virtual void process(const Buffer<T> &b, bool) { for (size_t i = 0; i < b.size(); i++) data.push_back(b[i]); }
};

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); failures++; } } while (0)

int main() {
  const double Fs = 2.4e6; const size_t N = 4096, NB = 4;
  {
    IQSigGen<int16_t> g1(Fs, N), g2(Fs, N);
    g1.addSine(100e3, 8000, 0); g1.addSine(-300e3, 6000, 0.3); g2.addSine(100e3, 8000, 0); g2.addSine(-300e3, 6000, 0.3);
    sdr::IQBaseBand<int16_t> cpu_bb(100e3, 100e3, 50e3, 127, 8); sdr::FMDemod<int16_t> cpu_fm; Rec<int16_t> cpu_out;
    sdr::gpu::IQBaseBand<int16_t> gpu_bb(100e3, 100e3, 50e3, 127, 8); sdr::FMDemod<int16_t> ref_fm; Rec<int16_t> gpu_out;
    g1.connect(&cpu_bb, true); cpu_bb.connect(&cpu_fm, true); cpu_fm.connect(&cpu_out, true);
    g2.connect(&gpu_bb, true); gpu_bb.connect(&ref_fm, true); ref_fm.connect(&gpu_out, true);
    for (size_t b = 0; b < NB; b++) { g1.next(); g2.next(); }
    CHECK(cpu_out.data.size() == 2047 && cpu_out.data == gpu_out.data);
    CHECK(gpu_bb.Source::sampleRate() == cpu_bb.Source::sampleRate());
  }
  {
    const size_t M = 8192, K = 5;
    IQSigGen<int16_t> g1(Fs, M), g2(Fs, M, K * M / Fs - 0.5 / Fs);
    g1.addSine(100e3, 8000, 0); g1.addSine(-300e3, 6000, 0.3); g2.addSine(100e3, 8000, 0); g2.addSine(-300e3, 6000, 0.3);
    sdr::FIRLowPass<cs16> cpu_fir(127, 100e3); sdr::FMDemod<int16_t> cpu_fm; Rec<int16_t> cpu_out;
    g1.connect(&cpu_fir, true); cpu_fir.connect(&cpu_fm, true); cpu_fm.connect(&cpu_out, true);
    for (size_t b = 0; b < K; b++) g1.next();
    sdr::gpu::FIRLowPass<cs16> gpu_fir(127, 100e3); sdr::gpu::FMDemod<int16_t> gpu_fm; Rec<int16_t> gpu_out;
    g2.connect(&gpu_fir, false); gpu_fir.connect(&gpu_fm, true); gpu_fm.connect(&gpu_out, true);
    Queue::get().addIdle(&g2, &IQSigGen<int16_t>::next);
    Queue::get().start(); Queue::get().wait();
    CHECK(cpu_out.data.size() == K * M && cpu_out.data == gpu_out.data);
  }
  {   // mid-stream retuning: the reference node and the GPU node get the same setter calls between buffers
    const size_t N2 = 4096;
    IQSigGen<int16_t> g1(Fs, N2), g2(Fs, N2);
    g1.addSine(100e3, 8000, 0); g1.addSine(-300e3, 6000, 0.3); g2.addSine(100e3, 8000, 0); g2.addSine(-300e3, 6000, 0.3);
    sdr::IQBaseBand<int16_t> cpu_bb(100e3, 100e3, 50e3, 127, 8); sdr::FMDemod<int16_t> cpu_fm; Rec<int16_t> cpu_out;
    sdr::gpu::IQBaseBand<int16_t> gpu_bb(100e3, 100e3, 50e3, 127, 8); sdr::FMDemod<int16_t> ref_fm; Rec<int16_t> gpu_out;
    g1.connect(&cpu_bb, true); cpu_bb.connect(&cpu_fm, true); cpu_fm.connect(&cpu_out, true);
    g2.connect(&gpu_bb, true); gpu_bb.connect(&ref_fm, true); ref_fm.connect(&gpu_out, true);
    g1.next(); g2.next();
    cpu_bb.setCenterFrequency(-150e3); gpu_bb.setCenterFrequency(-150e3);
    g1.next(); g2.next();
    cpu_bb.setFilterFrequency(-150e3); gpu_bb.setFilterFrequency(-150e3);
    cpu_bb.setFilterWidth(30e3); gpu_bb.setFilterWidth(30e3);
    g1.next(); g2.next();
    cpu_bb.setSubsample(8); gpu_bb.setSubsample(8);            // _reconfigure, same geometry: ring kept
    g1.next(); g2.next();
    cpu_bb.setSubsample(4); gpu_bb.setSubsample(4);            // _reconfigure with a new decimation: new output Config, new device plan
    g1.next(); g2.next();
    cpu_bb.setOutputSampleRate(100e3); gpu_bb.setOutputSampleRate(100e3);   // ... and ÷24
    g1.next(); g2.next();
    cpu_bb.setCenterFrequency(80e3); gpu_bb.setCenterFrequency(80e3);
    cpu_bb.setSubsample(8); gpu_bb.setSubsample(8);            // (the output rate set above still rules: ÷24 again, same Config)
    g1.next(); g2.next();
    // every output must match: the new device plans take the ring over as _reconfigure leaves it (rotated), the
    // reference FMDemod behind either node is reset by the same Config changes
    CHECK(cpu_out.data.size() > 3300 && cpu_out.data.size() == gpu_out.data.size());
    size_t bad = cpu_out.data.size();
    for (size_t i = 0; i < cpu_out.data.size() && i < gpu_out.data.size(); i++) if (cpu_out.data[i] != gpu_out.data[i]) { bad = i; break; }
    if (bad != cpu_out.data.size()) std::printf("first mismatch at output %zu: reference %d, gpu %d\n", bad, cpu_out.data[bad], gpu_out.data[bad]);
    CHECK(bad == cpu_out.data.size());
    CHECK(gpu_bb.subSample() == cpu_bb.subSample() && gpu_bb.Source::sampleRate() == cpu_bb.Source::sampleRate());
  }
  {   // setOrder mid-stream (src/baseband.hh:69-79: new kernel, new ring — uninitialised in the reference —, nothing else):
      // once `order` samples have passed the two nodes agree again, i.e. decimator, counters and LUT phase went on
    const size_t N2 = 4096;
    IQSigGen<int16_t> g1(Fs, N2), g2(Fs, N2);
    g1.addSine(100e3, 8000, 0); g1.addSine(-300e3, 6000, 0.3); g2.addSine(100e3, 8000, 0); g2.addSine(-300e3, 6000, 0.3);
    sdr::IQBaseBand<int16_t> cpu_bb(100e3, 100e3, 50e3, 127, 8); Rec<cs16> cpu_out;
    sdr::gpu::IQBaseBand<int16_t> gpu_bb(100e3, 100e3, 50e3, 127, 8); Rec<cs16> gpu_out;
    g1.connect(&cpu_bb, true); cpu_bb.connect(&cpu_out, true);
    g2.connect(&gpu_bb, true); gpu_bb.connect(&gpu_out, true);
    g1.next(); g2.next();
    const size_t before = cpu_out.data.size();
    cpu_bb.setOrder(161); gpu_bb.setOrder(161);
    g1.next(); g2.next(); g1.next(); g2.next();
    CHECK(cpu_out.data.size() == gpu_out.data.size() && cpu_out.data.size() == before + 1024);
    const size_t skip = (161 + 8) / 8 + 1;   // windows that may still see the reference's uninitialised ring
    size_t bad = cpu_out.data.size();
    for (size_t i = 0; i < cpu_out.data.size() && i < gpu_out.data.size(); i++) {
      if (i >= before && i < before + skip) continue;
      if (cpu_out.data[i] != gpu_out.data[i]) { bad = i; break; }
    }
    if (bad != cpu_out.data.size()) std::printf("setOrder: first mismatch at output %zu\n", bad);
    CHECK(bad == cpu_out.data.size());
  }
  std::printf("%s (%d failures)\n", failures ? "FAILED" : "OK", failures);
  return failures ? 1 : 0;
}
