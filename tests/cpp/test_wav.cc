// test_wav.cc — WavSink / WavSource of this repository's core (include/sdr/wavfile.hh) against files written
// and read by the reference's own nodes (tests/golden/g11_*, cut by oracle/ref_driver.cc). CPU only.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "sdr/exception.hh"
#include "sdr/logger.hh"
#include "sdr/buffer.hh"
#include "sdr/queue.hh"
#include "sdr/node.hh"
#include "sdr/siggen.hh"
#include "sdr/utils.hh"
#include "sdr/wavfile.hh"

using namespace sdr;
typedef std::complex<int16_t> cs16;
typedef std::complex<uint8_t> cu8;

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); failures++; } } while (0)

static std::string g_golden = "tests/golden", g_tmp = "/tmp";
static std::vector<uint8_t> slurp(const std::string &path) {
  std::vector<uint8_t> v; FILE *f = fopen(path.c_str(), "rb"); if (!f) return v;
  uint8_t tmp[4096]; size_t n; while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) v.insert(v.end(), tmp, tmp + n);
  fclose(f); return v;
}
static void spit(const std::string &path, const std::vector<uint8_t> &v) {
  FILE *f = fopen(path.c_str(), "wb"); fwrite(v.data(), 1, v.size(), f); fclose(f);
}

template <class T> struct Feeder : public Source {
  void cfg(double Fs, size_t bs) { setConfig(Config(Config::typeId<T>(), Fs, bs, 1)); }
  void feed(T *p, size_t n) { Buffer<T> b(p, n); send(b, false); }
};

// files written here are byte-identical to the reference's
static void testSinkBytes() {
  const std::string out = g_tmp + "/sdr_test_wav_out.wav";
  { IQSigGen<int16_t> gen(2.4e6, 1000); gen.addSine(100e3, 8000, 0.0); gen.addSine(-300e3, 6000, 0.3);
    Recorder<cs16> r; gen.connect(&r, true); gen.next();
    Feeder<cs16> src; src.cfg(2.4e6, 600); WavSink<cs16> sink(out); src.connect(&sink, true);
    src.feed(&r.data[0], 600); src.feed(&r.data[600], 400); sink.close();
    CHECK(slurp(out) == slurp(g_golden + "/g11_wav_cs16.bin")); }
  { std::vector<int16_t> y(777); for (size_t i = 0; i < y.size(); i++) y[i] = (int16_t)(12000 * std::sin(2 * M_PI * i / 50.0));
    { Feeder<int16_t> src; src.cfg(22050, 777); WavSink<int16_t> sink(out); src.connect(&sink, true); src.feed(&y[0], 777); }
    CHECK(slurp(out) == slurp(g_golden + "/g11_wav_s16.bin")); }
  { std::vector<uint8_t> raw = slurp(g_golden + "/g9_iq_cu8.bin");
    { Feeder<cu8> src; src.cfg(1e6, 4096); WavSink<cu8> sink(out); src.connect(&sink, true);
      for (size_t b = 0; b < 3; b++) src.feed(reinterpret_cast<cu8 *>(&raw[b * 8192]), 4096); }
    CHECK(slurp(out) == slurp(g_golden + "/g11_wav_cu8.bin")); }
  // type rules (src/wavfile.hh:56-58, :69-74)
  bool thrown = false;
  try { WavSink<float> bad(out); } catch (ConfigError &) { thrown = true; }
  CHECK(thrown);
  thrown = false;
  try { Feeder<cs16> src; src.cfg(1e3, 16); WavSink<int16_t> sink(out); src.connect(&sink, true); } catch (ConfigError &) { thrown = true; }
  CHECK(thrown);
  remove(out.c_str());
}

struct Flag { int n; Flag() : n(0) {} void hit() { n++; } };

// the reference's files read back: type, rate, buffer lengths, EOS (facts recorded in the manifest by the reference reader)
static void testSourceReads() {
  const std::string in = g_tmp + "/sdr_test_wav_in.wav";
  std::vector<uint8_t> raw = slurp(g_golden + "/g9_iq_cu8.bin");
  spit(in, slurp(g_golden + "/g11_wav_cu8.bin"));
  { WavSource rd(in, 5000); Recorder<cu8> cap; rd.connect(&cap, true);
    Flag eos; rd.addEOS(&eos, &Flag::hit);
    CHECK(rd.isOpen() && !rd.isReal() && rd.type() == Config::Type_cu8 && rd.sampleRate() == 1e6 && rd.frameCount() == 12288);
    for (int k = 0; k < 5; k++) rd.next();
    CHECK(cap.lens.size() == 3 && cap.lens[0] == 5000 && cap.lens[1] == 5000 && cap.lens[2] == 2288 && eos.n == 2 && !rd.isOpen());
    CHECK(cap.data.size() * 2 == raw.size() && 0 == memcmp(cap.data.data(), raw.data(), raw.size())); }
  spit(in, slurp(g_golden + "/g11_wav_s16.bin"));
  { WavSource rd(in, 1024); Recorder<int16_t> cap; rd.connect(&cap, true);
    CHECK(rd.isReal() && rd.type() == Config::Type_s16 && rd.sampleRate() == 22050.0);
    for (int k = 0; k < 2; k++) rd.next();
    CHECK(cap.data.size() == 777 && cap.data[1] == (int16_t)(12000 * std::sin(2 * M_PI * 1 / 50.0))); }
  spit(in, slurp(g_golden + "/g11_wav_cs16.bin"));
  { WavSource rd(1000); rd.open(in); Recorder<cs16> cap; rd.connect(&cap, true);
    CHECK(rd.type() == Config::Type_cs16 && rd.sampleRate() == 2.4e6);
    rd.next(); CHECK(cap.lens.size() == 1 && cap.lens[0] == 1000); }
  // an extra chunk between `fmt ` and `data` is skipped; junk is refused
  { std::vector<uint8_t> f = slurp(g_golden + "/g11_wav_s16.bin"), g;
    const uint8_t list[12] = {'L', 'I', 'S', 'T', 4, 0, 0, 0, 'a', 'b', 'c', 'd'};
    for (size_t i = 0; i < f.size(); i++) { if (i == 36) for (int k = 0; k < 12; k++) g.push_back(list[k]); g.push_back(f[i]); }
    spit(in, g);
    WavSource rd(in, 1024); Recorder<int16_t> cap; rd.connect(&cap, true); rd.next();
    CHECK(cap.lens.size() == 1 && cap.lens[0] == 777); }
  { std::vector<uint8_t> junk(100, 7); spit(in, junk);
    bool thrown = false; try { WavSource rd(in, 16); } catch (RuntimeError &) { thrown = true; } CHECK(thrown); }
  { WavSource rd("/nonexistent/file.wav", 16); CHECK(!rd.isOpen()); }
  remove(in.c_str());
}

// WavSource as a Queue idle handler with stop-on-EOS: the arrangement of the reference examples (sdr_wavplay style)
static void testQueueDriven() {
  const std::string in = g_tmp + "/sdr_test_wav_q.wav";
  spit(in, slurp(g_golden + "/g11_wav_cu8.bin"));
  WavSource rd(in, 4096); Recorder<cu8> cap; rd.connect(&cap);   // queued edge
  Queue &q = Queue::get();
  q.addIdle(&rd, &WavSource::next);
  rd.addEOS(&q, &Queue::stop);
  q.start(); q.wait();
  q.remIdle(&rd);
  CHECK(cap.data.size() == 12288 && cap.lens.size() == 3);
  remove(in.c_str());
}

int main(int argc, char **argv) {
  if (argc > 1) g_golden = argv[1];
  if (argc > 2) g_tmp = argv[2];
  try {
    testSinkBytes();
    testSourceReads();
    testQueueDriven();
  } catch (std::exception &e) {
    std::printf("FAIL: exception: %s\n", e.what());
    return 2;
  }
  std::printf("%s (%d failures)\n", failures ? "FAILED" : "OK", failures);
  return failures ? 1 : 0;
}
