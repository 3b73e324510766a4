// ref_driver — drives the UNMODIFIED reference nodes (compiled from /root/reference/src where
// they lie; see oracle/Makefile) to (1) cut golden vectors for tests/golden/ and (2) time the
// reference CPU path as bench.py's `cpu_baseline` (kind "reference").
//
// TEST INFRASTRUCTURE ONLY. Nothing in the product path (libsdr_amd/, include/) may link,
// call or execute this program. It is our own code: it only *includes* the reference headers at
// build time; no reference source text lives in this repository.
//
//   ref_driver golden <outdir>          write fixtures + <outdir>/manifest.json
//   ref_driver bench <chain> <nbuf>     time a chain over <nbuf> 65536-sample buffers (1 thread)
//
// Wiring follows examples/sdr_fm.cc:49-53 (source -> node -> demod, direct edges); the capture
// sink plays the role of DebugStore (src/utils.hh:799-841) but keeps every buffer.
#include "sdr.hh"
#include "filternode.hh"   // only for sinc_flt_kernel<> (a free template; needs no FFTW)

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <sstream>
#include <chrono>

using namespace sdr;
typedef std::complex<int16_t> cs16;
typedef std::complex<float> cf32;

// ---------------------------------------------------------------------------------------------
// capture sink / chunk feeder
// ---------------------------------------------------------------------------------------------
template <class T>
class Capture : public Sink<T> {
public:
  std::vector<T> data;
  std::vector<size_t> lens;
  bool keep;
  size_t total;
  Capture() : keep(true), total(0) {}
  virtual void config(const Config &) {}
  virtual void process(const Buffer<T> &b, bool) {
    lens.push_back(b.size());
    total += b.size();
    if (keep) { for (size_t i = 0; i < b.size(); i++) data.push_back(b[i]); }
  }
};

template <class T>
class Feeder : public Source {
public:
  void configure(double Fs, size_t maxlen) {
    this->setConfig(Config(Config::typeId<T>(), Fs, maxlen, 1));
  }
  void feed(T *p, size_t n) {
    Buffer<T> view(p, n);
    this->send(view, false);
  }
};

// exposes the protected design products of IQBaseBand (kernel, LUT, increment, decimation)
class BBProbe : public IQBaseBand<int16_t> {
public:
  BBProbe(double Fc, double Ff, double width, size_t order, size_t sub, double oFs = 0.0)
    : IQBaseBand<int16_t>(Fc, Ff, width, order, sub, oFs) {}
  std::vector<int32_t> taps() const {
    std::vector<int32_t> k;
    for (size_t i = 0; i < _order; i++) { k.push_back(_kernel[i].real()); k.push_back(_kernel[i].imag()); }
    return k;
  }
  std::vector<int32_t> lut() const {
    std::vector<int32_t> l;
    for (size_t i = 0; i < _lut_size; i++) { l.push_back(_lut[i].real()); l.push_back(_lut[i].imag()); }
    return l;
  }
  size_t lutInc() const { return _lut_inc; }
  bool negative() const { return 0 > _freq_shift; }
  size_t decim() const { return _sub_sample; }
};


// same for the real-input BaseBand<int16_t> (src/baseband.hh:305-529)
class RBProbe : public BaseBand<int16_t> {
public:
  RBProbe(double Fc, double Ff, double width, size_t order, size_t sub) : BaseBand<int16_t>(Fc, Ff, width, order, sub) {}
  std::vector<int32_t> taps() const {
    std::vector<int32_t> k;
    for (size_t i = 0; i < _order; i++) { k.push_back(_kernel[i].real()); k.push_back(_kernel[i].imag()); }
    return k;
  }
  size_t lutInc() const { return FreqShiftBase<int16_t>::_lut_inc; }
  bool negative() const { return 0 > FreqShiftBase<int16_t>::_freq_shift; }
};

// ---------------------------------------------------------------------------------------------
// manifest writer
// ---------------------------------------------------------------------------------------------
static std::string g_out;
static std::ostringstream g_manifest;
static bool g_first = true;

template <class T>
static void dump(const std::string &name, const char *dtype, const std::vector<T> &v,
                 const std::string &extra_json = "") {
  std::string path = g_out + "/" + name + ".bin";
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) { perror(path.c_str()); exit(2); }
  if (v.size()) fwrite(&v[0], sizeof(T), v.size(), f);
  fclose(f);
  if (!g_first) g_manifest << ",\n";
  g_first = false;
  g_manifest << "  \"" << name << "\": {\"file\": \"" << name << ".bin\", \"dtype\": \"" << dtype
             << "\", \"count\": " << v.size();
  if (extra_json.size()) g_manifest << ", " << extra_json;
  g_manifest << "}";
}

static std::string lens_json(const char *key, const std::vector<size_t> &l) {
  std::ostringstream s;
  s << "\"" << key << "\": [";
  for (size_t i = 0; i < l.size(); i++) { if (i) s << ", "; s << l[i]; }
  s << "]";
  return s.str();
}

template <class T> static std::vector<int16_t> flat16(const std::vector<T> &v);
template <> std::vector<int16_t> flat16<cs16>(const std::vector<cs16> &v) {
  std::vector<int16_t> o; for (size_t i = 0; i < v.size(); i++) { o.push_back(v[i].real()); o.push_back(v[i].imag()); } return o;
}
static std::vector<float> flatf(const std::vector<cf32> &v) {
  std::vector<float> o; for (size_t i = 0; i < v.size(); i++) { o.push_back(v[i].real()); o.push_back(v[i].imag()); } return o;
}

// ---------------------------------------------------------------------------------------------
// input generation with the reference's own generator (src/siggen.hh:90-157)
// ---------------------------------------------------------------------------------------------
struct Tone { double f, a, p; };

template <class S>
static std::vector< std::complex<S> > siggen(double Fs, size_t bs, size_t nbuf, const std::vector<Tone> &tones) {
  IQSigGen<S> gen(Fs, bs);
  for (size_t i = 0; i < tones.size(); i++) gen.addSine(tones[i].f, tones[i].a, tones[i].p);
  Capture< std::complex<S> > cap;
  gen.connect(&cap, true);
  for (size_t b = 0; b < nbuf; b++) gen.next();
  return cap.data;
}

static std::vector<Tone> two_tone_i16() {
  std::vector<Tone> t; Tone a = {100e3, 8000, 0.0}, b = {-300e3, 6000, 0.3}; t.push_back(a); t.push_back(b); return t;
}
static std::vector<Tone> two_tone_f32() {
  std::vector<Tone> t; Tone a = {100e3, 0.5, 0.0}, b = {-300e3, 0.3, 0.3}; t.push_back(a); t.push_back(b); return t;
}

// feed `x` to `head` in the given chunk lengths (cycled); returns number of chunks fed
template <class T>
static void feed_chunks(Feeder<T> &src, std::vector<T> &x, const std::vector<size_t> &chunks,
                        std::vector<size_t> &used) {
  size_t off = 0, c = 0;
  while (off < x.size()) {
    size_t n = std::min(chunks[c % chunks.size()], x.size() - off);
    src.feed(&x[off], n);  // n may be 0
    used.push_back(n);
    off += n; c++;
    if (c > 100000) break;
  }
}

// ---------------------------------------------------------------------------------------------
// golden cases
// ---------------------------------------------------------------------------------------------
enum Demod { D_NONE, D_FM, D_AM, D_USB };

static void case_iqbb(const std::string &name, std::vector<cs16> x, double Fs, double Fc, double Ff,
                      double width, size_t order, size_t sub, double oFs,
                      const std::vector<size_t> &chunks, Demod demod, bool dump_design) {
  size_t maxlen = 0; for (size_t i = 0; i < chunks.size(); i++) maxlen = std::max(maxlen, chunks[i]);
  Feeder<cs16> src; src.configure(Fs, maxlen);
  BBProbe bb(Fc, Ff, width, order, sub, oFs);
  src.connect(&bb, true);
  std::vector<size_t> used;
  std::ostringstream par;
  par << "\"Fs\": " << Fs << ", \"Fc\": " << Fc << ", \"Ff\": " << Ff << ", \"width\": " << width
      << ", \"order\": " << order << ", \"sub\": " << sub << ", \"oFs\": " << oFs
      << ", \"decim\": " << bb.decim() << ", \"lut_inc\": " << bb.lutInc()
      << ", \"negative\": " << (bb.negative() ? 1 : 0);
  if (dump_design) {
    dump(name + "_taps", "i32", bb.taps(), par.str());
    dump(name + "_lut", "i32", bb.lut());
  }
  if (demod == D_NONE) {
    Capture<cs16> cap; bb.connect(&cap, true);
    feed_chunks(src, x, chunks, used);
    dump(name + "_out", "cs16", flat16(cap.data),
         par.str() + ", " + lens_json("in_lens", used) + ", " + lens_json("out_lens", cap.lens));
  } else if (demod == D_FM) {
    FMDemod<int16_t> fm; Capture<int16_t> cap; bb.connect(&fm, true); fm.connect(&cap, true);
    feed_chunks(src, x, chunks, used);
    dump(name + "_fm", "i16", cap.data,
         par.str() + ", " + lens_json("in_lens", used) + ", " + lens_json("out_lens", cap.lens));
  } else if (demod == D_AM) {
    AMDemod<int16_t> am; Capture<int16_t> cap; bb.connect(&am, true); am.connect(&cap, true);
    feed_chunks(src, x, chunks, used);
    dump(name + "_am", "i16", cap.data,
         par.str() + ", " + lens_json("in_lens", used) + ", " + lens_json("out_lens", cap.lens));
  } else {
    USBDemod<int16_t> usb; Capture<int16_t> cap; bb.connect(&usb, true); usb.connect(&cap, true);
    feed_chunks(src, x, chunks, used);
    dump(name + "_usb", "i16", cap.data,
         par.str() + ", " + lens_json("in_lens", used) + ", " + lens_json("out_lens", cap.lens));
  }
}

static void case_fir_cs16(const std::string &name, std::vector<cs16> x, double Fs, size_t order,
                          double Fcut, const std::vector<size_t> &chunks, bool with_fm) {
  size_t maxlen = 0; for (size_t i = 0; i < chunks.size(); i++) maxlen = std::max(maxlen, chunks[i]);
  Feeder<cs16> src; src.configure(Fs, maxlen);
  FIRLowPass<cs16> fir(order, Fcut);
  src.connect(&fir, true);
  std::vector<size_t> used;
  std::ostringstream par; par << "\"Fs\": " << Fs << ", \"order\": " << order << ", \"Fcut\": " << Fcut;
  if (!with_fm) {
    Capture<cs16> cap; fir.connect(&cap, true);
    feed_chunks(src, x, chunks, used);
    dump(name + "_out", "cs16", flat16(cap.data), par.str() + ", " + lens_json("in_lens", used));
  } else {
    FMDemod<int16_t> fm; Capture<int16_t> cap; fir.connect(&fm, true); fm.connect(&cap, true);
    feed_chunks(src, x, chunks, used);
    dump(name + "_fm", "i16", cap.data, par.str() + ", " + lens_json("in_lens", used)
         + ", " + lens_json("out_lens", cap.lens));
  }
}

static void golden() {
  const double Fs = 2.4e6;
  std::vector<size_t> c4096(1, 4096);

  // G1 — inputs
  std::vector<cs16> x16 = siggen<int16_t>(Fs, 4096, 4, two_tone_i16());
  dump("g1_iq_cs16", "cs16", flat16(x16), "\"Fs\": 2400000, \"bufsize\": 4096, \"nbuf\": 4, "
       "\"tones\": [[100000, 8000, 0.0], [-300000, 6000, 0.3]]");
  std::vector<cf32> xf = siggen<float>(Fs, 4096, 3, two_tone_f32());
  dump("g1_iq_cf32", "cf32", flatf(xf), "\"Fs\": 2400000, \"bufsize\": 4096, \"nbuf\": 3, "
       "\"tones\": [[100000, 0.5, 0.0], [-300000, 0.3, 0.3]]");
  // single tone inputs used by SURVEY Appendix B known answers
  { std::vector<Tone> t; Tone a = {100e3, 8000, 0}; t.push_back(a);
    dump("g1_iq_cs16_tone_p100k", "cs16", flat16(siggen<int16_t>(Fs, 4096, 2, t)),
         "\"Fs\": 2400000, \"bufsize\": 4096, \"nbuf\": 2, \"tones\": [[100000, 8000, 0.0]]"); }
  { std::vector<Tone> t; Tone a = {-100e3, 8000, 0}; t.push_back(a);
    dump("g1_iq_cs16_tone_m100k", "cs16", flat16(siggen<int16_t>(Fs, 4096, 1, t)),
         "\"Fs\": 2400000, \"bufsize\": 4096, \"nbuf\": 1, \"tones\": [[-100000, 8000, 0.0]]"); }

  // G2/G3/G4 — IQBaseBand<int16>(Fc=Ff=100k, width 50k, order 127, /8) and demods in place
  case_iqbb("g3_iqbb127d8", x16, Fs, 100e3, 100e3, 50e3, 127, 8, 0.0, c4096, D_NONE, true);
  case_iqbb("g4_iqbb127d8", x16, Fs, 100e3, 100e3, 50e3, 127, 8, 0.0, c4096, D_FM, false);
  case_iqbb("g4_iqbb127d8", x16, Fs, 100e3, 100e3, 50e3, 127, 8, 0.0, c4096, D_AM, false);
  case_iqbb("g4_iqbb127d8", x16, Fs, 100e3, 100e3, 50e3, 127, 8, 0.0, c4096, D_USB, false);

  // G2 — FIRLowPass designs
  { size_t Ns[3] = {127, 255, 4097};
    for (int k = 0; k < 3; k++) {
      std::vector<double> a(Ns[k], 0.0);
      FIRLowPassCoeffs::coeffs(a, 0, 100e3, Fs);
      std::ostringstream nm; nm << "g2_firlp_alpha" << Ns[k];
      dump(nm.str(), "f64", a, "\"Fs\": 2400000, \"Fcut\": 100000"); } }

  // G5 — FIRLowPass<cs16> 127 / 255, plain and -> FMDemod in place
  case_fir_cs16("g5_fir127", x16, Fs, 127, 100e3, c4096, false);
  case_fir_cs16("g5_fir255", x16, Fs, 255, 100e3, c4096, false);
  case_fir_cs16("g5_fir127", x16, Fs, 127, 100e3, c4096, true);
  case_fir_cs16("g5_fir255", x16, Fs, 255, 100e3, c4096, true);
  { // Appendix B: 2nd generator buffer of the +100k tone into a fresh FIR
    std::vector<Tone> t; Tone a = {100e3, 8000, 0}; t.push_back(a);
    std::vector<cs16> two = siggen<int16_t>(Fs, 4096, 2, t);
    std::vector<cs16> second(two.begin() + 4096, two.end());
    case_fir_cs16("g5_fir127_tone_buf2", second, Fs, 127, 100e3, c4096, false);
    case_fir_cs16("g5_fir127_tone_buf2", second, Fs, 127, 100e3, c4096, true); }

  // G6 — cf32: FIRLowPass<cf32>(127) -> SubSample(8); AM/USB float on the FIR output
  { Feeder<cf32> src; src.configure(Fs, 4096);
    FIRLowPass<cf32> fir(127, 100e3); src.connect(&fir, true);
    Capture<cf32> cap; fir.connect(&cap, true);
    for (size_t b = 0; b < 3; b++) src.feed(&xf[b * 4096], 4096);
    dump("g6_fir127_cf32_out", "cf32", flatf(cap.data), "\"order\": 127, \"Fcut\": 100000, \"bufsize\": 4096");
    std::vector<cf32> firout = cap.data;
    { Feeder<cf32> s2; s2.configure(Fs, 4096); SubSample<cf32> sub(size_t(8)); s2.connect(&sub, true);
      Capture<cf32> c2; sub.connect(&c2, true);
      for (size_t b = 0; b < 3; b++) s2.feed(&firout[b * 4096], 4096);
      dump("g6_fir127_cf32_sub8", "cf32", flatf(c2.data), lens_json("out_lens", c2.lens)); }
    { Feeder<cf32> s2; s2.configure(Fs, 4096); SubSample<cf32> sub(size_t(3)); s2.connect(&sub, true);
      Capture<cf32> c2; sub.connect(&c2, true);
      for (size_t b = 0; b < 3; b++) s2.feed(&firout[b * 4096], 4096);
      dump("g6_fir127_cf32_sub3", "cf32", flatf(c2.data), lens_json("out_lens", c2.lens)); }
    { Feeder<cf32> s2; s2.configure(Fs, 4096); AMDemod<float> am; s2.connect(&am, true);
      Capture<float> c2; am.connect(&c2, true);
      for (size_t b = 0; b < 3; b++) s2.feed(&firout[b * 4096], 4096);
      dump("g6_fir127_cf32_am", "f32", c2.data); }
    { Feeder<cf32> s2; s2.configure(Fs, 4096); USBDemod<float> usb; s2.connect(&usb, true);
      Capture<float> c2; usb.connect(&c2, true);
      for (size_t b = 0; b < 3; b++) s2.feed(&firout[b * 4096], 4096);
      dump("g6_fir127_cf32_usb", "f32", c2.data); }
    { // FIRLowPass<cf32>(4097) on the first 2 buffers (config-4 comparison oracle)
      Feeder<cf32> s3; s3.configure(Fs, 4096); FIRLowPass<cf32> f2(4097, 100e3); s3.connect(&f2, true);
      Capture<cf32> c3; f2.connect(&c3, true);
      for (size_t b = 0; b < 3; b++) s3.feed(&xf[b * 4096], 4096);
      dump("g6_fir4097_cf32_out", "cf32", flatf(c3.data), "\"order\": 4097, \"Fcut\": 100000, \"bufsize\": 4096"); }
  }
  // SubSample<cs16> n=8 and n=3 over the raw int16 input
  { size_t ns[2] = {8, 3};
    for (int k = 0; k < 2; k++) {
      Feeder<cs16> s2; s2.configure(Fs, 4096); SubSample<cs16> sub(ns[k]); s2.connect(&sub, true);
      Capture<cs16> c2; sub.connect(&c2, true);
      for (size_t b = 0; b < 4; b++) s2.feed(&x16[b * 4096], 4096);
      std::ostringstream nm; nm << "g6_subsample_cs16_n" << ns[k];
      dump(nm.str(), "cs16", flat16(c2.data), lens_json("out_lens", c2.lens)); } }
  // standalone int16 demods over the raw input (out of place AM/USB are fully defined)
  { Feeder<cs16> s2; s2.configure(Fs, 4096); AMDemod<int16_t> am; s2.connect(&am, true);
    Capture<int16_t> c2; am.connect(&c2, true);
    for (size_t b = 0; b < 4; b++) s2.feed(&x16[b * 4096], 4096);
    dump("g4_raw_am", "i16", c2.data); }
  { Feeder<cs16> s2; s2.configure(Fs, 4096); USBDemod<int16_t> usb; s2.connect(&usb, true);
    Capture<int16_t> c2; usb.connect(&c2, true);
    for (size_t b = 0; b < 4; b++) s2.feed(&x16[b * 4096], 4096);
    dump("g4_raw_usb", "i16", c2.data); }
  { // FMDemod out of place over the raw input: index 0 of each buffer is uninitialised memory in
    // the reference (src/demod.hh:208,245) -> the test masks it; we zero it here for determinism.
    Feeder<cs16> s2; s2.configure(Fs, 4096); FMDemod<int16_t> fm; s2.connect(&fm, true);
    Capture<int16_t> c2; fm.connect(&c2, true);
    for (size_t b = 0; b < 4; b++) s2.feed(&x16[b * 4096], 4096);
    for (size_t b = 0; b < 4; b++) c2.data[b * 4096] = 0;
    dump("g4_raw_fm_masked0", "i16", c2.data); }

  // G7 — FFT-filter time-domain kernel h (float), the libm/float-phase sensitive part
  { int Ns[2] = {1024, 8192};
    for (int k = 0; k < 2; k++) {
      int N = Ns[k]; double fmin = 50e3, fmax = 150e3; double bw = fmax - fmin, Fc = fmin + bw / 2;
      std::vector<cf32> h;
      for (int i = 0; i < N; i++) h.push_back(sinc_flt_kernel<float>(i, N, Fc, bw, Fs));
      std::ostringstream nm; nm << "g7_fftfilt_h" << N;
      dump(nm.str(), "cf32", flatf(h), "\"Fs\": 2400000, \"fmin\": 50000, \"fmax\": 150000"); } }

  // G8 — edge cases
  { std::vector<cs16> xm = siggen<int16_t>(Fs, 4096, 1, std::vector<Tone>(1, Tone{-100e3, 8000, 0}));
    case_iqbb("g8_neg_o16_d1", xm, Fs, -100e3, -100e3, 50e3, 16, 1, 0.0, c4096, D_NONE, true); }
  case_iqbb("g8_o21_d3", x16, Fs, 100e3, 100e3, 12.5e3, 21, 3, 0.0, c4096, D_NONE, true);
  case_iqbb("g8_o33_d5", x16, Fs, -300e3, -300e3, 50e3, 33, 5, 0.0, c4096, D_NONE, true);
  case_iqbb("g8_o33_d5", x16, Fs, -300e3, -300e3, 50e3, 33, 5, 0.0, c4096, D_FM, false);
  case_iqbb("g8_o16_d4_even", x16, Fs, 100e3, 120e3, 80e3, 16, 4, 0.0, c4096, D_NONE, true);
  case_iqbb("g8_o255_d8", x16, Fs, 100e3, 100e3, 50e3, 255, 8, 0.0, c4096, D_NONE, true);
  case_iqbb("g8_noshift_o21_d8", x16, Fs, 0.0, 100e3, 50e3, 21, 8, 0.0, c4096, D_NONE, true);
  case_iqbb("g8_ofs_d300", x16, Fs, 100e3, 100e3, 12.5e3, 21, 1, 8000.0, c4096, D_NONE, true);
  { // irregular chunking incl. zero-output and zero-length calls
    size_t cl[] = {1, 7, 8, 9, 0, 3, 4096, 5, 1000, 17, 2, 2, 2, 2048, 4000, 1};
    std::vector<size_t> chunks(cl, cl + sizeof(cl) / sizeof(cl[0]));
    case_iqbb("g8_irregular", x16, Fs, 100e3, 100e3, 50e3, 127, 8, 0.0, chunks, D_NONE, false);
    case_iqbb("g8_irregular", x16, Fs, 100e3, 100e3, 50e3, 127, 8, 0.0, chunks, D_FM, false);
    case_iqbb("g8_irregular", x16, Fs, 100e3, 100e3, 50e3, 127, 8, 0.0, chunks, D_USB, false);
    case_fir_cs16("g8_irregular_fir127", x16, Fs, 127, 100e3, chunks, false);
    case_fir_cs16("g8_irregular_fir127", x16, Fs, 127, 100e3, chunks, true); }
  { // fast_atan2<int16,int16> known answers (src/math.hh:31-40), pairs (a, b) -> angle
    int16_t ab[][2] = {{0,0},{1,0},{0,1},{-1,0},{0,-1},{1,1},{-1,1},{-1,-1},{1,-1},{32767,32767},
      {-32768,-32768},{-32768,32767},{32767,-32768},{1000,3},{3,1000},{-3,-1000},{12345,-6789},
      {-32768,0},{0,-32768},{32767,1},{-1,32767},{7,-7},{-20000,15000}};
    std::vector<int16_t> v;
    for (size_t i = 0; i < sizeof(ab) / sizeof(ab[0]); i++) {
      v.push_back(ab[i][0]); v.push_back(ab[i][1]);
      v.push_back(fast_atan2<int16_t, int16_t>(ab[i][0], ab[i][1])); }
    dump("g8_fast_atan2_triples", "i16", v); }
  { // full-scale / near-overflow input: amplitudes at 2^14 (IQSigGen's intended scale)
    std::vector<Tone> t; Tone a = {90e3, 16384, 0.7}, b = {110e3, 16383, 2.0}; t.push_back(a); t.push_back(b);
    std::vector<cs16> xl = siggen<int16_t>(Fs, 4096, 2, t);
    dump("g8_iq_cs16_loud", "cs16", flat16(xl), "\"Fs\": 2400000, \"bufsize\": 4096, \"nbuf\": 2");
    case_iqbb("g8_loud_iqbb127d8", xl, Fs, 100e3, 100e3, 50e3, 127, 8, 0.0, c4096, D_FM, false);
    case_iqbb("g8_loud_iqbb127d8", xl, Fs, 100e3, 100e3, 50e3, 127, 8, 0.0, c4096, D_AM, false);
    case_fir_cs16("g8_loud_fir127", xl, Fs, 127, 100e3, c4096, false); }
}


// ---------------------------------------------------------------------------------------------
// "next" rows: AutoCast cu8 -> cs16 in front of the path, FMDeemph behind it
//   cu8 source -> AutoCast<cs16> -> IQBaseBand<int16> -> FMDemod -> FMDeemph     (examples/sdr_fm.cc:38-53)
// ---------------------------------------------------------------------------------------------
static void golden_next() {
  typedef std::complex<uint8_t> cu8;
  const double Fs = 1e6; const size_t N = 4096, NB = 3;
  // synthetic RTL-SDR style bytes: two tones around 127 plus a deterministic dither
  std::vector<cu8> x(N * NB);
  uint32_t lcg = 12345u;
  for (size_t i = 0; i < x.size(); i++) {
    lcg = lcg * 1664525u + 1013904223u;
    const int d0 = (int)((lcg >> 24) & 3) - 1, d1 = (int)((lcg >> 16) & 3) - 1;
    const double a = 2 * M_PI * (100e3 + 3e3 * std::sin(2 * M_PI * 1e3 * i / Fs) / 1.0) * i / Fs, b = 2 * M_PI * (-250e3) * i / Fs + 0.3;
    int re = 127 + (int)std::lround(90 * std::cos(a) + 25 * std::cos(b)) + d0;
    int im = 127 + (int)std::lround(90 * std::sin(a) + 25 * std::sin(b)) + d1;
    re = std::min(255, std::max(0, re)); im = std::min(255, std::max(0, im));
    x[i] = cu8((uint8_t)re, (uint8_t)im);
  }
  { std::vector<uint8_t> flat; for (size_t i = 0; i < x.size(); i++) { flat.push_back(x[i].real()); flat.push_back(x[i].imag()); }
    // a few extreme bytes pin the int8 reinterpretation of AutoCast (src/autocast.hh:187-194)
    uint8_t ext[] = {0, 127, 128, 200, 255, 1, 126, 129};
    for (int k = 0; k < 8; k++) flat[k] = ext[k];
    for (size_t i = 0; i < 4; i++) x[i] = cu8(flat[2 * i], flat[2 * i + 1]);
    dump("g9_iq_cu8", "u8", flat, "\"Fs\": 1000000, \"bufsize\": 4096, \"nbuf\": 3"); }
  // AutoCast alone
  { Feeder<cu8> src; src.configure(Fs, N); AutoCast<cs16> cast; src.connect(&cast, true);
    Capture<cs16> cap; cast.connect(&cap, true);
    for (size_t b = 0; b < NB; b++) src.feed(&x[b * N], N);
    dump("g9_autocast_cs16", "cs16", flat16(cap.data)); }
  // the sdr_fm chain: orders 21 (the example's) and 127, /8; FM in place; de-emphasis at 125 kS/s
  size_t orders[2] = {21, 127};
  for (int k = 0; k < 2; k++) {
    Feeder<cu8> src; src.configure(Fs, N); AutoCast<cs16> cast; BBProbe bb(100e3, 100e3, 50e3, orders[k], 8);
    FMDemod<int16_t> fm; FMDeemph<int16_t> de; Capture<int16_t> capfm, capde;
    src.connect(&cast, true); cast.connect(&bb, true); bb.connect(&fm, true); fm.connect(&capfm, true); fm.connect(&de, true); de.connect(&capde, true);
    for (size_t b = 0; b < NB; b++) src.feed(&x[b * N], N);
    std::ostringstream nm; nm << "g9_cu8_iqbb" << orders[k] << "d8";
    std::ostringstream par; par << "\"Fs\": " << Fs << ", \"Fc\": 100000, \"Ff\": 100000, \"width\": 50000, \"order\": " << orders[k]
        << ", \"decim\": " << bb.decim() << ", \"lut_inc\": " << bb.lutInc() << ", \"negative\": 0";
    dump(nm.str() + "_taps", "i32", bb.taps(), par.str());
    // bb sends with allow_overwrite to its only sink, so FMDemod runs in place: index 0 = y[0].real()
    dump(nm.str() + "_fm", "i16", capfm.data, lens_json("out_lens", capfm.lens));
    dump(nm.str() + "_fm_deemph", "i16", capde.data, lens_json("out_lens", capde.lens));
  }
  // FMDeemph alone on a known int16 stream (the masked FM output above would carry garbage at index 0)
  { std::vector<int16_t> y(3 * 512);
    uint32_t l2 = 777u;
    for (size_t i = 0; i < y.size(); i++) { l2 = l2 * 1664525u + 1013904223u; y[i] = (int16_t)(2000 * std::sin(2 * M_PI * i / 97.0) + (int)((l2 >> 20) & 255) - 128); }
    y[5] = 32767; y[6] = -32768; y[7] = 0;
    dump("g9_deemph_in", "i16", y);
    double rates[2] = {125e3, 48e3};
    for (int k = 0; k < 2; k++) {
      Feeder<int16_t> src; src.setConfig(Config(Config::Type_s16, rates[k], 512, 1));
      FMDeemph<int16_t> de; src.connect(&de, true); Capture<int16_t> cap; de.connect(&cap, true);
      for (int b = 0; b < 3; b++) src.feed(&y[b * 512], 512);
      std::ostringstream nm; nm << "g9_deemph_out_" << (int)rates[k];
      dump(nm.str(), "i16", cap.data); } }
}


// ---------------------------------------------------------------------------------------------
// "next" row 3: the real-input BaseBand<int16_t>  (src/baseband.hh:305-529)
// ---------------------------------------------------------------------------------------------
static void case_bb_real(const std::string &name, std::vector<int16_t> x, double Fs, double Fc, double Ff, double width,
                         size_t order, size_t sub, const std::vector<size_t> &chunks) {
  size_t maxlen = 0; for (size_t i = 0; i < chunks.size(); i++) maxlen = std::max(maxlen, chunks[i]);
  Feeder<int16_t> src; src.configure(Fs, maxlen);
  RBProbe bb(Fc, Ff, width, order, sub);
  src.connect(&bb, true);
  Capture<cs16> cap; bb.connect(&cap, true);
  std::vector<size_t> used;
  feed_chunks(src, x, chunks, used);
  std::ostringstream par;
  par << "\"Fs\": " << Fs << ", \"Fc\": " << Fc << ", \"Ff\": " << Ff << ", \"width\": " << width
      << ", \"order\": " << order << ", \"decim\": " << sub << ", \"lut_inc\": " << bb.lutInc()
      << ", \"negative\": " << (bb.negative() ? 1 : 0);
  dump(name + "_taps", "i32", bb.taps(), par.str());
  dump(name + "_out", "cs16", flat16(cap.data),
       par.str() + ", " + lens_json("in_lens", used) + ", " + lens_json("out_lens", cap.lens));
}

static void golden_real() {
  const double Fs = 1e6; const size_t N = 4096, NB = 3;
  std::vector<int16_t> x(N * NB), loud(N * NB);
  uint32_t lcg = 4242u;
  for (size_t i = 0; i < x.size(); i++) {
    lcg = lcg * 1664525u + 1013904223u;
    const double v = 9000 * std::cos(2 * M_PI * 100e3 * i / Fs) + 5000 * std::cos(2 * M_PI * 230e3 * i / Fs + 0.4)
                   + 2000 * std::sin(2 * M_PI * 101.5e3 * i / Fs);
    x[i] = (int16_t)((int)v + (int)((lcg >> 22) & 63) - 32);
    loud[i] = (int16_t)(((i / 3) & 1) ? 32767 : -32768);   // full-scale square wave
  }
  x[0] = 32767; x[1] = -32768; x[2] = 0;
  dump("g10_real_in", "i16", x, "\"Fs\": 1000000, \"bufsize\": 4096, \"nbuf\": 3");
  dump("g10_real_loud_in", "i16", loud);
  std::vector<size_t> c4096(1, 4096), ragged, c1(1, 1000);
  size_t r[] = {1000, 7, 0, 4096, 333, 1, 2048}; ragged.assign(r, r + 7);
  case_bb_real("g10_bb21d8", x, Fs, 100e3, 100e3, 50e3, 21, 8, c4096);
  case_bb_real("g10_bb127d8_neg_ragged", x, Fs, -100e3, 100e3, 50e3, 127, 8, ragged);
  case_bb_real("g10_bb64d5", x, Fs, 100e3, 100e3, 80e3, 64, 5, c4096);      // even order: the centre tap is 1 (:470)
  case_bb_real("g10_bb16d1_noshift", x, Fs, 0, 230e3, 50e3, 16, 1, c1);
  case_bb_real("g10_bb1d3", x, Fs, 230e3, 230e3, 50e3, 1, 3, ragged);       // order 1: the single tap is 65536 (17 bits)
  case_bb_real("g10_bb127d8_loud", loud, Fs, 100e3, 100e3, 300e3, 127, 8, c4096);

  // G16 — the real-input node retuned and reconfigured MID-STREAM: setFrequencyShift (FreqShiftBase, src/freqshift.hh:52-54,
  // 78-87: new increment and sign, LUT phase restarts, nothing else) and a new source Config with another buffer size
  // (BaseBand::config, src/baseband.hh:357-395: setSampleRate -> LUT increment + kernel; _last, _sample_count and _ring_offset
  // reset; the ring's contents kept where they lie)
  { Feeder<int16_t> src; src.configure(Fs, 4096);
    RBProbe bb(100e3, 100e3, 50e3, 127, 8);
    src.connect(&bb, true);
    Capture<cs16> cap; bb.connect(&cap, true);
    std::vector<size_t> used;
    size_t off = 0;
    auto feed = [&](size_t n) { src.feed(&x[off], n); used.push_back(n); off += n; };
    feed(4096); feed(1000);
    bb.setFrequencyShift(-150e3);
    feed(3000);
    src.configure(Fs, 2048);
    feed(2048); feed(2048);
    const std::string ev = "\"Fs\": 1000000, \"Fc\": 100000, \"Ff\": 100000, \"width\": 50000, \"order\": 127, \"decim\": 8, \"events\": [[\"feed\", 4096], [\"feed\", 1000], "
        "[\"shift\", -150000], [\"feed\", 3000], [\"bufsize\", 2048], [\"feed\", 2048], [\"feed\", 2048]]";
    dump("g16_bb_real_retune_out", "cs16", flat16(cap.data), ev + ", " + lens_json("in_lens", used) + ", " + lens_json("out_lens", cap.lens)); }
}


// ---------------------------------------------------------------------------------------------
// "next" row 4: WAV files as the reference writes / reads them (src/wavfile.hh, src/wavfile.cc)
// and the sdr_rec NFM chain (examples/sdr_rec.cc:46-48,66-97) run file to file
// ---------------------------------------------------------------------------------------------
static std::vector<uint8_t> file_bytes(const std::string &path) {
  std::vector<uint8_t> v; FILE *f = fopen(path.c_str(), "rb"); if (!f) return v;
  uint8_t tmp[4096]; size_t n; while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) v.insert(v.end(), tmp, tmp + n);
  fclose(f); return v;
}

static void golden_wav() {
  typedef std::complex<uint8_t> cu8;
  const std::string tmp = g_out + "/_tmp.wav", tmp2 = g_out + "/_tmp2.wav";
  // (1) files written by the reference WavSink
  { std::vector<cs16> x = siggen<int16_t>(2.4e6, 1000, 1, two_tone_i16());
    { Feeder<cs16> src; src.configure(2.4e6, 600); WavSink<cs16> sink(tmp); src.connect(&sink, true);
      src.feed(&x[0], 600); src.feed(&x[600], 400); sink.close(); }
    dump("g11_wav_cs16", "u8", file_bytes(tmp), "\"Fs\": 2400000, \"frames\": 1000, \"in_lens\": [600, 400]"); }
  { std::vector<int16_t> y(777); for (size_t i = 0; i < y.size(); i++) y[i] = (int16_t)(12000 * std::sin(2 * M_PI * i / 50.0));
    { Feeder<int16_t> src; src.configure(22050, 777); WavSink<int16_t> sink(tmp); src.connect(&sink, true);
      src.feed(&y[0], 777); }   // closed by the destructor
    dump("g11_wav_s16", "u8", file_bytes(tmp), "\"Fs\": 22050, \"frames\": 777"); }
  // (2) a cu8 recording (the g9 bytes) and what the reference WavSource makes of it
  { std::vector<uint8_t> raw = file_bytes(g_out + "/g9_iq_cu8.bin");
    { Feeder<cu8> src; src.configure(1e6, 4096); WavSink<cu8> sink(tmp); src.connect(&sink, true);
      for (size_t b = 0; b < 3; b++) src.feed(reinterpret_cast<cu8 *>(&raw[b * 8192]), 4096); }
    dump("g11_wav_cu8", "u8", file_bytes(tmp), "\"Fs\": 1000000, \"frames\": 12288");
    WavSource rd(tmp, 5000); Capture<cu8> cap; rd.connect(&cap, true);
    int eos = 0; struct Flag { int *p; void hit() { (*p)++; } } flag = {&eos}; rd.addEOS(&flag, &Flag::hit);
    for (int k = 0; k < 5; k++) rd.next();
    std::vector<uint8_t> got; for (size_t i = 0; i < cap.data.size(); i++) { got.push_back(cap.data[i].real()); got.push_back(cap.data[i].imag()); }
    std::ostringstream ex; ex << lens_json("out_lens", cap.lens) << ", \"eos\": " << eos << ", \"type\": " << (int)rd.type()
                              << ", \"Fs\": " << rd.sampleRate() << ", \"same_as_input\": " << (got == raw ? 1 : 0);
    dump("g11_wavsource_cu8_readback", "u8", std::vector<uint8_t>(), ex.str());   // facts only; the data is g9_iq_cu8
    // (3) the NFM chain of sdr_rec, file to file, direct edges
    { WavSource src(tmp, 4096); AutoCast<cs16> cast; IQBaseBand<int16_t> bb(0, 0, 12.5e3, 16, 1, 12e3);
      FMDemod<int16_t> fm; FMDeemph<int16_t> de; WavSink<int16_t> sink(tmp2);
      src.connect(&cast, true); cast.connect(&bb, true); bb.connect(&fm, true); fm.connect(&de, true); de.connect(&sink, true);
      for (int k = 0; k < 4; k++) src.next(); }
    dump("g11_chain_nfm_wav", "u8", file_bytes(tmp2));
    { WavSource src(tmp, 4096); AutoCast<cs16> cast; IQBaseBand<int16_t> bb(0, 1500, 3e3, 16, 1, 12e3);
      USBDemod<int16_t> usb; WavSink<int16_t> sink(tmp2);
      src.connect(&cast, true); cast.connect(&bb, true); bb.connect(&usb, true); usb.connect(&sink, true);
      for (int k = 0; k < 4; k++) src.next(); }
    dump("g11_chain_usb_wav", "u8", file_bytes(tmp2)); }
  remove(tmp.c_str()); remove(tmp2.c_str());

  // G13 — the int8 chain of the reference's documentation example (src/sdr.hh:225-240): IQBaseBand<int8_t> ->
  // FMDemod<int8_t,int16_t>. Its compute type is int16 (Traits<int8_t>::SScalar), so the Q14 kernel products wrap and
  // what comes out is a few noisy bits — parity is what is asked, not usefulness. Inputs: full-range pseudo-random
  // complex<int8> (an LCG written out here) so that every wrap is exercised.
  const double Fs = 2.4e6;
  {
    typedef std::complex<int8_t> cs8;
    std::vector<cs8> x8(3 * 4096);
    uint32_t lcg = 12345u;
    for (size_t i = 0; i < x8.size(); i++) {
      lcg = lcg * 1664525u + 1013904223u; const int8_t re = (int8_t)(lcg >> 24);
      lcg = lcg * 1664525u + 1013904223u; const int8_t im = (int8_t)(lcg >> 24);
      x8[i] = cs8(re, im);
    }
    { std::vector<int8_t> f; for (size_t i = 0; i < x8.size(); i++) { f.push_back(x8[i].real()); f.push_back(x8[i].imag()); }
      dump("g13_iq_cs8", "i8", f, "\"Fs\": 2400000, \"bufsize\": 4096, \"nbuf\": 3"); }
    struct P8 { const char *name; double Fc, Ff, width; size_t order, sub; double oFs; };
    const P8 cases[3] = {{"g13_i8_o21_d8", 100e3, 100e3, 50e3, 21, 8, 0.0}, {"g13_i8_doc_o16", 0.0, 0.0, 100e3, 16, 0, 100e3},
                         {"g13_i8_neg_o33_d5", -300e3, -300e3, 50e3, 33, 5, 0.0}};
    for (int k = 0; k < 3; k++) for (int fm = 0; fm < 2; fm++) {
      const P8 &c = cases[k];
      Feeder<cs8> src; src.configure(Fs, 4096);
      IQBaseBand<int8_t> bb(c.Fc, c.Ff, c.width, c.order, c.sub, c.oFs);
      src.connect(&bb, true);
      std::ostringstream par;
      par << "\"Fs\": " << Fs << ", \"Fc\": " << c.Fc << ", \"Ff\": " << c.Ff << ", \"width\": " << c.width << ", \"order\": " << c.order
          << ", \"sub\": " << c.sub << ", \"oFs\": " << c.oFs << ", \"decim\": " << bb.subSample();
      std::vector<size_t> used;
      const size_t chunks[4] = {4096, 1000, 3096, 4096};
      size_t off = 0;
      if (!fm) {
        Capture<cs8> cap; bb.connect(&cap, true);
        for (int q = 0; q < 4; q++) { src.feed(&x8[off], chunks[q]); used.push_back(chunks[q]); off += chunks[q]; }
        std::vector<int8_t> f; for (size_t i = 0; i < cap.data.size(); i++) { f.push_back(cap.data[i].real()); f.push_back(cap.data[i].imag()); }
        dump(std::string(c.name) + "_out", "i8", f, par.str() + ", " + lens_json("in_lens", used) + ", " + lens_json("out_lens", cap.lens));
      } else {
        FMDemod<int8_t, int16_t> dem; Capture<int16_t> cap; bb.connect(&dem, true); dem.connect(&cap, true);
        for (int q = 0; q < 4; q++) { src.feed(&x8[off], chunks[q]); used.push_back(chunks[q]); off += chunks[q]; }
        dump(std::string(c.name) + "_fm", "i16", cap.data, par.str() + ", " + lens_json("in_lens", used) + ", " + lens_json("out_lens", cap.lens));
      }
    }
  }

  // G12 — the reference node retuned MID-STREAM (src/baseband.hh:82-112): setCenterFrequency only restarts the LUT
  // phase with the new increment (src/freqshift.hh:52-54,78-87), setFilterFrequency / setFilterWidth only swap the
  // kernel, setSubsample runs _reconfigure (counters reset, ring contents kept where they lie, :156-177).
  std::vector<cs16> x16 = siggen<int16_t>(Fs, 4096, 4, two_tone_i16());   // = g1_iq_cs16
  for (int fm = 0; fm < 2; fm++) {
    Feeder<cs16> src; src.configure(Fs, 4096);
    BBProbe bb(100e3, 100e3, 50e3, 127, 8);
    src.connect(&bb, true);
    Capture<cs16> cap; Capture<int16_t> capf; FMDemod<int16_t> dem;
    if (fm) { bb.connect(&dem, true); dem.connect(&capf, true); } else bb.connect(&cap, true);
    std::vector<size_t> used;
    size_t off = 0;
    auto feed = [&](size_t n) { src.feed(&x16[off], n); used.push_back(n); off += n; };
    feed(4096); feed(3000);
    bb.setCenterFrequency(-150e3);
    feed(2000);
    bb.setFilterFrequency(-150e3); bb.setFilterWidth(30e3);
    feed(3192);
    bb.setSubsample(8);
    feed(4096);
    const std::string ev = "\"Fs\": 2400000, \"order\": 127, \"decim\": 8, \"events\": [[\"feed\", 4096], [\"feed\", 3000], "
        "[\"center\", -150000], [\"feed\", 2000], [\"filter\", -150000, 30000], [\"feed\", 3192], [\"reconfigure\"], [\"feed\", 4096]]";
    if (fm) dump("g12_retune_fm", "i16", capf.data, ev + ", " + lens_json("in_lens", used) + ", " + lens_json("out_lens", capf.lens));
    else dump("g12_retune_out", "cs16", flat16(cap.data), ev + ", " + lens_json("in_lens", used) + ", " + lens_json("out_lens", cap.lens));
  }

  // G14 — the node's GEOMETRY changed mid-stream: setSubsample / setOutputSampleRate (src/baseband.hh:106-112) and a new
  // source Config (another buffer size, :115-132) all run _reconfigure (:156-194: counters and LUT phase reset, the
  // ring's contents kept where they lie); setOrder (:69-79) swaps kernel and ring and touches nothing else — its new
  // ring is UNINITIALISED memory, so the outputs whose windows still see it are not defined: the first `undefined_head`
  // outputs of the feed behind it are written as zeros here and skipped by the tests (the order only grows: a smaller
  // one can leave _ring_offset beyond the new ring).
  for (int fm = 0; fm < 2; fm++) {
    Feeder<cs16> src; src.configure(Fs, 4096);
    BBProbe bb(100e3, 100e3, 50e3, 127, 8);
    src.connect(&bb, true);
    Capture<cs16> cap; Capture<int16_t> capf; FMDemod<int16_t> dem;
    if (fm) { bb.connect(&dem, true); dem.connect(&capf, true); } else bb.connect(&cap, true);
    std::vector<size_t> used;
    size_t off = 0;
    auto feed = [&](size_t n) { src.feed(&x16[off], n); used.push_back(n); off += n; };
    feed(4096); feed(3000);
    bb.setSubsample(4);
    feed(2000);
    bb.setOutputSampleRate(100e3);            // D = trunc(2.4e6 / 100e3) = 24
    feed(3192);
    src.configure(Fs, 2048);                  // a new source Config: config() -> _reconfigure
    feed(2048);
    bb.setOrder(161);
    const size_t before = fm ? capf.data.size() : cap.data.size();
    feed(2048);
    const size_t undefined_head = (161 + 24 + 23) / 24 + 2;   // windows that may still see the new ring's old contents (+1: FM looks one back)
    std::ostringstream ev;
    ev << "\"Fs\": 2400000, \"order\": 127, \"decim\": 8, \"undefined_head\": " << undefined_head << ", \"events\": [[\"feed\", 4096], [\"feed\", 3000], "
          "[\"subsample\", 4], [\"feed\", 2000], [\"orate\", 100000], [\"feed\", 3192], [\"bufsize\", 2048], [\"feed\", 2048], "
          "[\"order\", 161], [\"feed\", 2048]]";
    if (fm) {
      for (size_t i = before; i < before + undefined_head && i < capf.data.size(); i++) capf.data[i] = 0;
      dump("g14_regeom_fm", "i16", capf.data, ev.str() + ", " + lens_json("in_lens", used) + ", " + lens_json("out_lens", capf.lens));
    } else {
      for (size_t i = before; i < before + undefined_head && i < cap.data.size(); i++) cap.data[i] = cs16(0, 0);
      dump("g14_regeom_out", "cs16", flat16(cap.data), ev.str() + ", " + lens_json("in_lens", used) + ", " + lens_json("out_lens", cap.lens));
    }
  }

  // G15 — the FFT filter's kernel designer for a block size that is not a power of two and for Scalar = double
  // (sinc_flt_kernel<Scalar>, src/filternode.hh:16-28: FilterNode(size_t block_size), template <class Scalar>)
  { const double fmin = -350e3, fmax = -250e3, bw = fmax - fmin, Fc = fmin + bw / 2;
    std::vector<cf32> h;
    for (int i = 0; i < 1000; i++) h.push_back(sinc_flt_kernel<float>(i, 1000, Fc, bw, Fs));
    dump("g15_fftfilt_h1000", "cf32", flatf(h), "\"Fs\": 2400000, \"fmin\": -350000, \"fmax\": -250000");
    const int Ns[2] = {1000, 1024};
    for (int k = 0; k < 2; k++) {
      std::vector<double> hd;
      for (int i = 0; i < Ns[k]; i++) { const std::complex<double> v = sinc_flt_kernel<double>(i, Ns[k], Fc, bw, Fs); hd.push_back(v.real()); hd.push_back(v.imag()); }
      std::ostringstream nm; nm << "g15_fftfilt_h" << Ns[k] << "_f64";
      dump(nm.str(), "f64", hd, "\"Fs\": 2400000, \"fmin\": -350000, \"fmax\": -250000");
    }
  }
}

// G17 — FIRLowPass::setFreq between buffers (FIRFilter::setUpperFreq, src/firfilter.hh:165-170,287): only _alpha is
// recomputed, the ring goes on. 4 buffers of G1's inputs, cut-off 100 kHz -> 40 kHz after the second buffer.
static void golden_setters() {
  const double Fs = 2.4e6; const size_t N = 4096;
  std::vector<cs16> x16 = siggen<int16_t>(Fs, N, 4, two_tone_i16());
  std::vector<cf32> xf = siggen<float>(Fs, N, 3, two_tone_f32());
  { Feeder<cs16> src; src.configure(Fs, N); FIRLowPass<cs16> fir(127, 100e3); Capture<cs16> cap;
    src.connect(&fir, true); fir.connect(&cap, true);
    for (size_t b = 0; b < 4; b++) { if (b == 2) fir.setFreq(40e3); src.feed(&x16[b * N], N); }
    dump("g17_fir127_setfreq_cs16", "cs16", flat16(cap.data), "\"Fs\": 2400000, \"order\": 127, \"freq\": [100000, 40000], \"switch_after_buffers\": 2, \"bufsize\": 4096"); }
  { Feeder<cf32> src; src.configure(Fs, N); FIRLowPass<cf32> fir(127, 100e3); Capture<cf32> cap;
    src.connect(&fir, true); fir.connect(&cap, true);
    for (size_t b = 0; b < 3; b++) { if (b == 1) fir.setFreq(40e3); src.feed(&xf[b * N], N); }
    dump("g17_fir127_setfreq_cf32", "cf32", flatf(cap.data), "\"Fs\": 2400000, \"order\": 127, \"freq\": [100000, 40000], \"switch_after_buffers\": 1, \"bufsize\": 4096"); }
}

// ---------------------------------------------------------------------------------------------
// timing of the reference CPU path (bench.py cpu_baseline kind "reference")
// ---------------------------------------------------------------------------------------------
static int bench(const std::string &chain, size_t nbuf, size_t N) {
  const double Fs = 2.4e6;
  std::vector<cs16> x = siggen<int16_t>(Fs, N, 1, two_tone_i16());
  std::vector<cf32> xf;
  if (chain == "fir_cf32_sub8" || chain == "fir_cf32_4097") xf = siggen<float>(Fs, N, 1, two_tone_f32());
  std::vector<cs16> work(N);
  size_t total_out = 0; long checksum = 0;
  std::chrono::steady_clock::time_point t0, t1;

#define RUN_I16(HEAD, TAILCAP, CT)                                                  \
  { Feeder<cs16> src; src.configure(Fs, N); src.connect(&(HEAD), true);             \
    Capture<CT> cap; cap.keep = false; (TAILCAP).connect(&cap, true);               \
    for (size_t b = 0; b < 2; b++) { src.feed(&x[0], N); }                          \
    t0 = std::chrono::steady_clock::now();                                          \
    for (size_t b = 0; b < nbuf; b++) { src.feed(&x[0], N); }                       \
    t1 = std::chrono::steady_clock::now(); total_out = cap.total; }

  if (chain == "iqbb_fm") {
    IQBaseBand<int16_t> bb(100e3, 100e3, 50e3, 127, 8); FMDemod<int16_t> fm; bb.connect(&fm, true);
    RUN_I16(bb, fm, int16_t)
  } else if (chain == "iqbb_usb") {
    IQBaseBand<int16_t> bb(100e3, 100e3, 50e3, 127, 8); USBDemod<int16_t> d; bb.connect(&d, true);
    RUN_I16(bb, d, int16_t)
  } else if (chain == "fir127_fm") {
    FIRLowPass<cs16> fir(127, 100e3); FMDemod<int16_t> fm; fir.connect(&fm, true);
    RUN_I16(fir, fm, int16_t)
  } else if (chain == "fir127_fm_queue") {
    // BASELINE config 1 as written: the reference's IQSigGen driven by the Queue's idle signal, its buffers travelling through the
    // Queue (a queued edge) into FIRLowPass<cs16>(127) -> FMDemod (direct edges) — examples/sdr_fm.cc:49-53's plumbing with the
    // signal generator in the RTL source's place (src/queue.cc:83-125, src/siggen.hh:116-133). The generator's own synthesis
    // (two complex exponentials per sample in double) is part of what the Queue thread does, as in the reference's examples.
    IQSigGen<int16_t> gen(Fs, N, double(nbuf * N) / Fs - 0.5 / Fs);
    gen.addSine(100e3, 0.5, 0.0); gen.addSine(-300e3, 0.4, 0.3);
    FIRLowPass<cs16> fir(127, 100e3); FMDemod<int16_t> fm; Capture<int16_t> cap; cap.keep = false;
    gen.connect(&fir); fir.connect(&fm, true); fm.connect(&cap, true);
    Queue::get().addIdle(&gen, &IQSigGen<int16_t>::next);
    t0 = std::chrono::steady_clock::now();
    Queue::get().start(); Queue::get().wait();
    t1 = std::chrono::steady_clock::now(); total_out = cap.total;
    nbuf = total_out / N;
  } else if (chain == "fir255_fm") {
    FIRLowPass<cs16> fir(255, 100e3); FMDemod<int16_t> fm; fir.connect(&fm, true);
    RUN_I16(fir, fm, int16_t)
  } else if (chain == "fir_cf32_sub8") {
    Feeder<cf32> src; src.configure(Fs, N);
    FIRLowPass<cf32> fir(127, 100e3); SubSample<cf32> sub(size_t(8));
    src.connect(&fir, true); fir.connect(&sub, true);
    Capture<cf32> cap; cap.keep = false; sub.connect(&cap, true);
    for (size_t b = 0; b < 2; b++) src.feed(&xf[0], N);
    t0 = std::chrono::steady_clock::now();
    for (size_t b = 0; b < nbuf; b++) src.feed(&xf[0], N);
    t1 = std::chrono::steady_clock::now(); total_out = cap.total;
  } else if (chain == "fir_cf32_4097") {
    // BASELINE config 4 (ii)'s time-domain side: FIRLowPass<complex<float>>(4097 taps) (src/firfilter.hh:231-247)
    Feeder<cf32> src; src.configure(Fs, N);
    FIRLowPass<cf32> fir(4097, 100e3);
    src.connect(&fir, true);
    Capture<cf32> cap; cap.keep = false; fir.connect(&cap, true);
    src.feed(&xf[0], N);
    t0 = std::chrono::steady_clock::now();
    for (size_t b = 0; b < nbuf; b++) src.feed(&xf[0], N);
    t1 = std::chrono::steady_clock::now(); total_out = cap.total;
  } else if (chain == "sdr_fm_cu8") {
    // the reference's own FM receiver plan (examples/sdr_fm.cc:38-53 without the audio device): complex<uint8> at 1 MS/s
    // -> AutoCast<cs16> -> IQBaseBand<int16>(100 kHz, 12.5 kHz wide, 21 taps, to 8 kS/s = /125) -> FMDemod
    typedef std::complex<uint8_t> cu8;
    std::vector<cu8> xb(N);
    for (size_t i = 0; i < N; i++) xb[i] = cu8((uint8_t)((x[i].real() >> 6) + 127), (uint8_t)((x[i].imag() >> 6) + 127));
    Feeder<cu8> src; src.configure(1e6, N); AutoCast<cs16> cast;
    IQBaseBand<int16_t> bb(100e3, 12.5e3, 21, 1, 8000.0); bb.setCenterFrequency(100e3); bb.setFilterFrequency(100e3);
    FMDemod<int16_t> fm; Capture<int16_t> cap; cap.keep = false;
    src.connect(&cast, true); cast.connect(&bb, true); bb.connect(&fm, true); fm.connect(&cap, true);
    for (size_t b = 0; b < 2; b++) src.feed(&xb[0], N);
    t0 = std::chrono::steady_clock::now();
    for (size_t b = 0; b < nbuf; b++) src.feed(&xb[0], N);
    t1 = std::chrono::steady_clock::now(); total_out = cap.total;
  } else {
    fprintf(stderr, "unknown chain %s\n", chain.c_str()); return 2;
  }
  double sec = std::chrono::duration<double>(t1 - t0).count();
  printf("{\"chain\": \"%s\", \"buffers\": %zu, \"samples\": %zu, \"seconds\": %.6f, "
         "\"msps\": %.4f, \"outputs\": %zu, \"threads\": 1, \"kind\": \"reference\"}\n",
         chain.c_str(), nbuf, nbuf * N, sec, nbuf * N / sec / 1e6, total_out + (size_t)(checksum & 0));
  return 0;
}

int main(int argc, char **argv) {
  if (argc >= 3 && std::string(argv[1]) == "golden") {
    g_out = argv[2];
    g_manifest << "{\n";
    golden();
    golden_next();
    golden_real();
    golden_wav();
    golden_setters();
    g_manifest << "\n}\n";
    std::string mp = g_out + "/manifest.json";
    FILE *f = fopen(mp.c_str(), "w"); fputs(g_manifest.str().c_str(), f); fclose(f);
    return 0;
  }
  if (argc >= 4 && std::string(argv[1]) == "bench") {
    return bench(argv[2], (size_t)atol(argv[3]), argc >= 5 ? (size_t)atol(argv[4]) : (size_t)65536);
  }
  fprintf(stderr, "usage: ref_driver golden <outdir> | bench <chain> <nbuf> [samples per buffer]\n");
  return 1;
}
