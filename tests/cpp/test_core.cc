// test_core.cc — CPU-only checks of the API-compatible sdr:: core (include/sdr/*.hh): the buffer
// reference-count contract, Config propagation, the allow_overwrite rule, the Queue worker and the
// IQSigGen restatement (against tests/golden/g1_iq_cs16.bin). Mirrors what the reference's own
// unit tests pin for these classes (test/buffertest.cc:10-122) plus the rules SURVEY §8b lists.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <vector>

#include "sdr/exception.hh"
#include "sdr/logger.hh"
#include "sdr/buffer.hh"
#include "sdr/queue.hh"
#include "sdr/node.hh"
#include "sdr/siggen.hh"
#include "sdr/utils.hh"

using namespace sdr;
typedef std::complex<int16_t> cs16;

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); failures++; } } while (0)

struct Probe : public Sink<cs16> {
  Config last; int configs = 0; std::vector<bool> aw; size_t samples = 0;
  virtual void config(const Config &c) { last = c; configs++; }
  virtual void process(const Buffer<cs16> &b, bool allow) { aw.push_back(allow); samples += b.size(); }
};
struct Feeder : public Source {
  void cfg(const Config &c) { setConfig(c); }
};
struct Thrower : public Sink<cs16> {
  virtual void config(const Config &c) {
    if (c.hasType() && c.type() != Config::typeId<cs16>()) { ConfigError e; e << "bad type " << c.type(); throw e; }
  }
  virtual void process(const Buffer<cs16> &, bool) {}
};

static void testBuffer() {
  Buffer<int8_t> a(3);
  CHECK(a.refCount() == 1 && a.isUnused());
  { Buffer<int8_t> b(a); CHECK(a.refCount() == 1 && b.refCount() == 1); }   // copies do not ref
  { Buffer<int8_t> b; b = a; CHECK(a.refCount() == 1); }
  a.ref(); CHECK(a.refCount() == 2 && !a.isUnused());
  a.unref(); CHECK(a.refCount() == 1);
  Buffer<int8_t> r(4);
  for (int i = 0; i < 4; i++) r[i] = int8_t(i);
  Buffer< std::complex<int8_t> > c(r);                                      // re-typed view: bytes / sizeof(T)
  CHECK(c.size() == 2 && c[0].real() == 0 && c[0].imag() == 1 && c[1].real() == 2 && c[1].imag() == 3);
  Buffer<int8_t> h = r.head(2), t = r.tail(1), s = r.sub(1, 2);
  CHECK(h.size() == 2 && t.size() == 1 && t[0] == 3 && s[0] == 1 && s[1] == 2 && r.sub(3, 2).isEmpty());
  CHECK(h.refCount() == 1);                                                  // views share the counter
  int16_t raw[4] = {1, 2, 3, 4};
  Buffer<int16_t> w(raw, 4);
  CHECK(w.refCount() == 0 && w.isUnused() && w[2] == 3);                     // foreign memory: uncounted
  w.unref(); CHECK(!w.isEmpty());
  r.unref(); CHECK(r.isEmpty());
  a.unref(); CHECK(a.isEmpty());
  BufferSet<float> set(2, 16);
  Buffer<float> b1 = set.getBuffer(), b2 = set.getBuffer(); CHECK(!set.hasBuffer());
  b1.ref(); b1.unref(); CHECK(set.hasBuffer());                              // a consumer's ref/unref recycles it
  Buffer<float> b3 = set.getBuffer(), b4 = set.getBuffer();                  // the pool grows instead of crashing
  CHECK(!b3.isEmpty() && !b4.isEmpty()); (void)b2;
}

static void testConfigAndOverwriteRule() {
  Feeder src; Probe p1, p2;
  src.connect(&p1, true);
  CHECK(p1.configs == 1 && !p1.last.hasType());                              // connect pushes the (empty) config at once
  src.cfg(Config(Config::Type_cs16, 2.4e6, 64, 1));
  CHECK(p1.configs == 2 && p1.last.sampleRate() == 2.4e6 && src.sampleRate() == 2.4e6 && src.type() == Config::Type_cs16);
  src.cfg(Config(Config::Type_cs16, 2.4e6, 64, 1));
  CHECK(p1.configs == 2);                                                    // unchanged config is not re-propagated
  Buffer<cs16> b(64);
  src.send(b, true);  CHECK(p1.aw.back() == true);                           // single direct sink + sender allows
  src.send(b, false); CHECK(p1.aw.back() == false);
  src.connect(&p2, true);
  src.send(b, true);  CHECK(p1.aw.back() == false && p2.aw.back() == false); // two sinks: nobody may overwrite
  src.disconnect(&p2);
  src.send(b, true);  CHECK(p1.aw.back() == true);
  Feeder f2; Thrower th; f2.cfg(Config(Config::Type_f32, 1e3, 8, 1));
  bool thrown = false;
  try { f2.connect(&th, true); } catch (ConfigError &e) { thrown = std::string(e.what()).find("bad type") != std::string::npos; }
  CHECK(thrown);
  CHECK(Config::typeId< std::complex<float> >() == Config::Type_cf32 && std::string(typeName(Config::Type_cs16)) == "complex int16");
  b.unref();
}

static void testSigGenGolden(const char *golden_dir) {
  std::string path = std::string(golden_dir) + "/g1_iq_cs16.bin";
  std::ifstream f(path.c_str(), std::ios::binary);
  std::vector<int16_t> ref(4 * 4096 * 2);
  f.read((char *)ref.data(), ref.size() * 2);
  CHECK(f.gcount() == std::streamsize(ref.size() * 2));
  IQSigGen<int16_t> gen(2.4e6, 4096);
  gen.addSine(100e3, 8000, 0.0); gen.addSine(-300e3, 6000, 0.3);
  Recorder<cs16> rec; gen.connect(&rec, true);
  for (int i = 0; i < 4; i++) gen.next();
  CHECK(rec.data.size() == 4 * 4096);
  size_t bad = 0;
  for (size_t i = 0; i < rec.data.size(); i++) if (rec.data[i].real() != ref[2 * i] || rec.data[i].imag() != ref[2 * i + 1]) bad++;
  CHECK(bad == 0);
}

static void testQueue() {
  // config-1 plumbing: generator on the idle signal, queued edge to the sink, tmax stops the queue
  IQSigGen<int16_t> gen(2.4e6, 1024, 10 * 1024 / 2.4e6);
  gen.addSine(100e3, 8000, 0.0);
  Probe sink; gen.connect(&sink, false);
  int started = 0, stopped = 0;
  struct Cnt { int *p; void hit() { (*p)++; } } cs{&started}, ce{&stopped};
  Queue::get().addStart(&cs, &Cnt::hit); Queue::get().addStop(&ce, &Cnt::hit);
  Queue::get().addIdle(&gen, &IQSigGen<int16_t>::next);
  Queue::get().start(); Queue::get().wait();
  CHECK(Queue::get().isStopped() && started == 1 && stopped == 1);
  CHECK(sink.samples >= 10 * 1024 && sink.samples <= 11 * 1024);
  CHECK(!sink.aw.empty() && sink.aw.back() == false);
  Queue::get().remIdle(&gen); Queue::get().remStart(&cs); Queue::get().remStop(&ce);
}

int main(int argc, char **argv) {
  testBuffer();
  testConfigAndOverwriteRule();
  testSigGenGolden(argc > 1 ? argv[1] : "tests/golden");
  testQueue();
  std::printf("%s (%d failures)\n", failures ? "FAILED" : "OK", failures);
  return failures ? 1 : 0;
}
