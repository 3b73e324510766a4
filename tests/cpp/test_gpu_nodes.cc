// test_gpu_nodes.cc — the MI355X nodes (include/sdr/gpu/nodes.hh) wired into real graphs with this
// repository's sdr:: core, compared bit-exactly with the CPU oracle (oracle/sdr_oracle.h; test side
// only). Graph shapes follow examples/sdr_fm.cc:49-53 and SURVEY §3.2/§3.3. Needs an MI355X.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <iostream>
#include <vector>

#include "sdr/sdr.hh"
#include "sdr_oracle.h"

using namespace sdr;
typedef std::complex<int16_t> cs16;
typedef std::complex<float> cf32;

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); failures++; } } while (0)

static const double FS = 2.4e6;

struct Feeder : public Source {
  void cfg(Config::Type t, size_t bs) { setConfig(Config(t, FS, bs, 1)); }
  template <class T> void feed(T *p, size_t n) { Buffer<T> b(p, n); send(b, false); }
};

static std::vector<cs16> tones(size_t n, double f1 = 100e3) {
  IQSigGen<int16_t> gen(FS, n); gen.addSine(f1, 8000, 0.0); gen.addSine(-300e3, 6000, 0.3);
  Recorder<cs16> r; gen.connect(&r, true); gen.next();
  return r.data;
}

// IQSigGen -> gpu::IQBaseBand -> gpu::FMDemod (direct edges, FM in place over the baseband's buffer)
static void testBasebandFmChain() {
  const size_t N = 4096, NB = 4;
  IQSigGen<int16_t> gen(FS, N); gen.addSine(100e3, 8000, 0.0); gen.addSine(-300e3, 6000, 0.3);
  gpu::IQBaseBand<int16_t> bb(100e3, 100e3, 50e3, 127, 8);
  gpu::FMDemod<int16_t> fm;
  Recorder<cs16> raw; Recorder<int16_t> out;
  gen.connect(&raw, true); gen.connect(&bb, true); bb.connect(&fm, true); fm.connect(&out, true);
  for (size_t b = 0; b < NB; b++) gen.next();
  CHECK(out.lens.size() == NB && out.lens[0] == 511 && out.lens[1] == 512);
  CHECK(bb.Source::sampleRate() == 300000.0 && fm.type() == Config::Type_s16);
  // oracle
  std::vector<int32_t> taps(2 * 127), lut(256);
  orc_iqbb_design(100e3, 50e3, FS, 127, taps.data()); orc_freqshift_lut_i16(lut.data());
  void *o = orc_iqbb_i16_create(taps.data(), 127, lut.data(), orc_freqshift_inc(100e3, FS), 0, 8);
  std::vector<int16_t> ref; int16_t last = 0;
  for (size_t b = 0; b < NB; b++) {
    std::vector<int16_t> y(2 * 520);
    size_t n = orc_iqbb_i16_process(o, (const int16_t *)&raw.data[b * N], N, y.data());
    std::vector<int16_t> f(n); f[0] = y[0];
    orc_fm_i16(y.data(), n, f.data(), &last);
    ref.insert(ref.end(), f.begin(), f.end());
  }
  orc_iqbb_i16_destroy(o);
  CHECK(ref.size() == out.data.size() && 0 == memcmp(ref.data(), out.data.data(), ref.size() * 2));
  // fused variant: one kernel launch per buffer, same numbers
  IQSigGen<int16_t> gen2(FS, N); gen2.addSine(100e3, 8000, 0.0); gen2.addSine(-300e3, 6000, 0.3);
  gpu::IQBaseBand<int16_t> fused(100e3, 100e3, 50e3, 127, 8); fused.setDemod(SDRHIP_EPI_FM);
  Recorder<int16_t> out2; gen2.connect(&fused, true); fused.connect(&out2, true);
  for (size_t b = 0; b < NB; b++) gen2.next();
  CHECK(out2.data == out.data);
}

// config 1 on the Queue: generator idle-driven, queued edge into the FIR, direct edge FIR -> FM
static void testFirFmOnQueue() {
  const size_t N = 8192, NB = 6;
  IQSigGen<int16_t> gen(FS, N, NB * N / FS - 0.5 / FS); gen.addSine(100e3, 8000, 0.0); gen.addSine(-300e3, 6000, 0.3);
  gpu::FIRLowPass<cs16> fir(127, 100e3);
  gpu::FMDemod<int16_t> fm;
  Recorder<int16_t> out;
  gen.connect(&fir, false); fir.connect(&fm, true); fm.connect(&out, true);
  Queue::get().addIdle(&gen, &IQSigGen<int16_t>::next);
  Queue::get().start(); Queue::get().wait();
  Queue::get().remIdle(&gen);
  CHECK(out.data.size() == NB * N);
  IQSigGen<int16_t> g2(FS, N); g2.addSine(100e3, 8000, 0.0); g2.addSine(-300e3, 6000, 0.3);
  Recorder<cs16> raw; g2.connect(&raw, true);
  std::vector<double> a(127); orc_fir_lowpass_design(127, 100e3, FS, a.data());
  void *f = orc_fir_create(a.data(), 127);
  std::vector<int16_t> ref; int16_t last = 0;
  for (size_t b = 0; b < NB; b++) {
    g2.next();
    std::vector<int16_t> y(2 * N), o(N);
    orc_fir_cs16_process(f, (const int16_t *)&raw.data[b * N], N, y.data());
    o[0] = y[0]; orc_fm_i16(y.data(), N, o.data(), &last);
    ref.insert(ref.end(), o.begin(), o.end());
  }
  orc_fir_destroy(f);
  CHECK(ref.size() == out.data.size() && 0 == memcmp(ref.data(), out.data.data(), ref.size() * 2));
}

// FIRLowPass::setFreq between buffers (FIRFilter::setUpperFreq, src/firfilter.hh:165-170,287): only the coefficients change,
// the ring goes on. The oracle has no setter: a fresh filter with the new coefficients primed with the `order` samples in
// front of the switch continues the reference's stream (pinned to the compiled reference: tests/golden g17).
static void testFirSetFreqKeepsTheRing() {
  const size_t N = 4096;
  IQSigGen<int16_t> gen(FS, N); gen.addSine(100e3, 8000, 0.0); gen.addSine(-300e3, 6000, 0.3);
  Recorder<cs16> raw, out; gpu::FIRLowPass<cs16> fir(127, 100e3);
  gen.connect(&raw, true); gen.connect(&fir, true); fir.connect(&out, true);
  gen.next(); gen.next();
  fir.setFreq(40e3);
  CHECK(fir.freq() == 40e3);
  gen.next(); gen.next();
  CHECK(out.data.size() == 4 * N);
  std::vector<double> a(127), b(127);
  orc_fir_lowpass_design(127, 100e3, FS, a.data()); orc_fir_lowpass_design(127, 40e3, FS, b.data());
  void *f1 = orc_fir_create(a.data(), 127), *f2 = orc_fir_create(b.data(), 127);
  std::vector<int16_t> y1(2 * 2 * N), y2(2 * 2 * N), scratch(2 * 127);
  orc_fir_cs16_process(f1, (const int16_t *)&raw.data[0], 2 * N, y1.data());
  orc_fir_cs16_process(f2, (const int16_t *)&raw.data[2 * N - 127], 127, scratch.data());
  orc_fir_cs16_process(f2, (const int16_t *)&raw.data[2 * N], 2 * N, y2.data());
  orc_fir_destroy(f1); orc_fir_destroy(f2);
  CHECK(0 == memcmp(y1.data(), &out.data[0], 2 * N * 4) && 0 == memcmp(y2.data(), &out.data[2 * N], 2 * N * 4));
}

// ownership rules: in place when allowed, own buffer otherwise, drop while the own buffer is referenced
static void testOwnership() {
  std::vector<cs16> x = tones(4096);
  Feeder src; src.cfg(Config::Type_cs16, 4096);
  gpu::IQBaseBand<int16_t> bb(100e3, 100e3, 50e3, 21, 8);
  struct Hold : public Sink<cs16> {
    RawBuffer kept; bool keep = false; size_t calls = 0; const char *last_ptr = 0; bool aw = false;
    virtual void config(const Config &) {}
    virtual void process(const Buffer<cs16> &b, bool a) { calls++; last_ptr = b.data(); aw = a; if (keep) { kept = b; kept.ref(); } }
  } hold;
  src.connect(&bb, true); bb.connect(&hold, true);
  src.feed(x.data(), 4096);
  CHECK(hold.calls == 1 && hold.aw && hold.last_ptr != (const char *)x.data());   // own buffer, downstream may overwrite
  hold.keep = true; src.feed(x.data(), 4096); CHECK(hold.calls == 2);
  src.feed(x.data(), 4096); CHECK(hold.calls == 2);                               // still referenced -> dropped
  hold.kept.unref(); hold.keep = false;
  src.feed(x.data(), 4096); CHECK(hold.calls == 3);
  Buffer<cs16> mine(4096); memcpy(mine.data(), x.data(), 4096 * 4);
  bb.handleBuffer(mine, true);                                                     // allow_overwrite -> in place
  CHECK(hold.calls == 4 && hold.last_ptr == mine.data());
  mine.unref();
  // wrong type -> ConfigError, incomplete config -> silent
  Feeder bad; bad.cfg(Config::Type_cf32, 64);
  bool thrown = false;
  try { bad.connect(&bb, true); } catch (ConfigError &e) { thrown = std::string(e.what()).find("Invalid type") != std::string::npos; }
  CHECK(thrown);
  Feeder empty; gpu::FMDemod<int16_t> fm; empty.connect(&fm, true);
}

// 16 channels through the ChannelBank = 16 independent single nodes
static void testChannelBank() {
  const size_t C = 16, N = 4096;
  gpu::ChannelBank<int16_t> bank(C, 100e3, 100e3, 50e3, 127, 8, SDRHIP_EPI_USB);
  std::vector<Feeder> src(C); std::vector< Recorder<int16_t> > rec(C);
  std::vector< std::vector<cs16> > x(C);
  for (size_t c = 0; c < C; c++) {
    x[c] = tones(2 * N, 90e3 + 1500.0 * c);
    src[c].cfg(Config::Type_cs16, N);
    src[c].connect(bank.sink(c), true); bank.source(c)->connect(&rec[c], true);
  }
  for (int k = 0; k < 2; k++) for (size_t c = 0; c < C; c++) src[c].feed(&x[c][k * N], N);
  std::vector<int32_t> taps(2 * 127), lut(256);
  orc_iqbb_design(100e3, 50e3, FS, 127, taps.data()); orc_freqshift_lut_i16(lut.data());
  for (size_t c = 0; c < C; c++) {
    void *o = orc_iqbb_i16_create(taps.data(), 127, lut.data(), orc_freqshift_inc(100e3, FS), 0, 8);
    std::vector<int16_t> ref;
    for (int k = 0; k < 2; k++) {
      std::vector<int16_t> y(2 * 520), u(520);
      size_t n = orc_iqbb_i16_process(o, (const int16_t *)&x[c][k * N], N, y.data());
      orc_usb_i16(y.data(), n, u.data()); ref.insert(ref.end(), u.begin(), u.begin() + n);
    }
    orc_iqbb_i16_destroy(o);
    CHECK(ref == rec[c].data);
  }
}

// the bank's outputs are views of one stage buffer: while a consumer still references a view of the last round
// (what a queued edge does until the Queue worker has delivered it), the next round is dropped — the rule every node
// follows for its own output buffer (reference src/baseband.hh:141-150)
static void testChannelBankDropsWhileOutputInUse() {
  const size_t C = 4, N = 4096;
  gpu::ChannelBank<int16_t> bank(C, 100e3, 100e3, 50e3, 21, 8);
  struct Hold : public Sink<cs16> {
    RawBuffer kept; bool keep = false; size_t calls = 0; std::vector<cs16> first;
    virtual void config(const Config &) {}
    virtual void process(const Buffer<cs16> &b, bool) {
      calls++;
      if (keep) { kept = b; kept.ref(); first.assign(b.data() ? reinterpret_cast<const cs16 *>(b.data()) : 0, reinterpret_cast<const cs16 *>(b.data()) + b.size()); }
    }
  };
  std::vector<Hold> hold(C); std::vector<Feeder> src(C);
  std::vector<cs16> x = tones(3 * N);
  for (size_t c = 0; c < C; c++) {
    src[c].cfg(Config::Type_cs16, N);
    src[c].connect(bank.sink(c), true); bank.source(c)->connect(&hold[c], true);
  }
  hold[2].keep = true;
  for (size_t c = 0; c < C; c++) src[c].feed(&x[0], N);
  CHECK(hold[0].calls == 1 && hold[2].calls == 1 && hold[2].first.size() == 511);
  hold[2].keep = false;
  for (size_t c = 0; c < C; c++) src[c].feed(&x[N], N);            // view of round 1 still referenced -> round dropped
  CHECK(hold[0].calls == 1 && hold[3].calls == 1);
  CHECK(0 == memcmp(hold[2].kept.data(), hold[2].first.data(), hold[2].first.size() * sizeof(cs16)));   // not clobbered
  hold[2].kept.unref();
  for (size_t c = 0; c < C; c++) src[c].feed(&x[2 * N], N);
  CHECK(hold[0].calls == 2 && hold[3].calls == 2);
}

// BASELINE config 5 from the C++ side: the bank split over several ranks (both on device 0 here: the GPU boxes of the
// test pool have one device; with distinct devices the same calls go through RCCL) equals the one-device bank
static void testChannelBankMultiRank() {
  const size_t C = 10, N = 4096;
  for (int epi = 0; epi < 2; epi++) {
    const int e = epi ? SDRHIP_EPI_FM : SDRHIP_EPI_USB;
    gpu::ChannelBank<int16_t> one(C, 100e3, 100e3, 50e3, 127, 8, e);
    gpu::ChannelBank<int16_t> three(C, 100e3, 100e3, 50e3, 127, 8, e, std::vector<int>(3, 0));   // blocks of 4, 3, 3 channels
    std::vector<Feeder> s1(C), s3(C); std::vector< Recorder<int16_t> > r1(C), r3(C);
    std::vector< std::vector<cs16> > x(C);
    for (size_t c = 0; c < C; c++) {
      x[c] = tones(3 * N, 80e3 + 2500.0 * c);
      s1[c].cfg(Config::Type_cs16, N); s3[c].cfg(Config::Type_cs16, N);
      s1[c].connect(one.sink(c), true); one.source(c)->connect(&r1[c], true);
      s3[c].connect(three.sink(c), true); three.source(c)->connect(&r3[c], true);
    }
    CHECK(three.ranks() == 3 && std::string(three.transport()) == "same-device copies");
    for (int k = 0; k < 3; k++) for (size_t c = 0; c < C; c++) { s1[c].feed(&x[c][k * N], N); s3[c].feed(&x[c][k * N], N); }
    for (size_t c = 0; c < C; c++) CHECK(r1[c].data.size() == 1535 && r1[c].data == r3[c].data);
  }
}

// the C ABI's comm calls directly: broadcast of a design buffer, gather of ragged blocks; two ranks on device 0, and one
// rank forced through RCCL (SDRHIP_COMM_FORCE_RCCL=1: dlopen, ncclCommInitAll, grouped send/recv on a 1-rank communicator)
static void testCommCalls() {
  for (int rccl = 0; rccl < 2; rccl++) {
    const int nr = rccl ? 1 : 2;
    int devs[2] = {0, 0};
    if (rccl) setenv("SDRHIP_COMM_FORCE_RCCL", "1", 1);
    sdrhip_comm *cm = 0;
    const int rc = sdrhip_comm_create(devs, nr, &cm);
    if (rccl) unsetenv("SDRHIP_COMM_FORCE_RCCL");
    CHECK(rc == SDRHIP_OK);
    if (rc != SDRHIP_OK) { std::printf("  comm_create: %s\n", sdrhip_last_error()); continue; }
    const char *tr = ""; sdrhip_comm_transport(cm, &tr);
    CHECK(std::string(tr) == (rccl ? "rccl" : "same-device copies"));
    std::vector<sdrhip_ctx *> ctx(nr); std::vector<void *> buf(nr), blk(nr);
    std::vector<uint32_t> design(1000); for (size_t i = 0; i < design.size(); i++) design[i] = uint32_t(i * 2654435761u);
    const size_t blkn[2] = {300, 177};
    void *all = 0;
    for (int r = 0; r < nr; r++) {
      sdrhip_comm_ctx(cm, r, &ctx[r]);
      sdrhip_malloc(ctx[r], design.size() * 4, &buf[r]); sdrhip_malloc(ctx[r], blkn[r] * 4, &blk[r]);
      sdrhip_memset(ctx[r], buf[r], 0, design.size() * 4);
      std::vector<uint32_t> v(blkn[r]); for (size_t i = 0; i < v.size(); i++) v[i] = uint32_t(1000 * (r + 1) + i);
      sdrhip_memcpy_h2d(ctx[r], blk[r], v.data(), v.size() * 4);
    }
    sdrhip_malloc(ctx[0], (blkn[0] + blkn[1]) * 4, &all);
    sdrhip_memcpy_h2d(ctx[0], buf[0], design.data(), design.size() * 4);
    CHECK(sdrhip_comm_broadcast(cm, buf.data(), design.size() * 4, 0) == SDRHIP_OK);
    std::vector<size_t> bytes(nr); for (int r = 0; r < nr; r++) bytes[r] = blkn[r] * 4;
    std::vector<const void *> send(blk.begin(), blk.end());
    CHECK(sdrhip_comm_gather(cm, send.data(), bytes.data(), all, 0) == SDRHIP_OK);
    CHECK(sdrhip_comm_synchronize(cm) == SDRHIP_OK);
    for (int r = 0; r < nr; r++) {
      std::vector<uint32_t> got(design.size());
      sdrhip_memcpy_d2h(ctx[r], got.data(), buf[r], got.size() * 4);
      CHECK(got == design);
    }
    size_t tot = 0; for (int r = 0; r < nr; r++) tot += blkn[r];
    std::vector<uint32_t> g(tot); sdrhip_memcpy_d2h(ctx[0], g.data(), all, tot * 4);
    bool okg = true; size_t o = 0;
    for (int r = 0; r < nr; r++) for (size_t i = 0; i < blkn[r]; i++, o++) okg = okg && g[o] == uint32_t(1000 * (r + 1) + i);
    CHECK(okg);
    for (int r = 0; r < nr; r++) { sdrhip_free(ctx[r], buf[r]); sdrhip_free(ctx[r], blk[r]); }
    sdrhip_free(ctx[0], all);
    CHECK(sdrhip_comm_destroy(cm) == SDRHIP_OK);
  }
}

// Pipelined use without sdrhip_comm_synchronize between the steps (sdrhip.h documents the comm calls as asynchronous):
// step k + 1 overwrites every rank's send buffer right after gather k was enqueued, twice over; each gather must still
// deliver the values of ITS step. Same-device transport (2 ranks on device 0) and, where the box has two devices, RCCL
// between device 0 and device 1 — the first time ncclSend/ncclRecv move bytes between two GPUs from this side.
static void testCommPipelinedAndTwoDevices() {
  int ndev = 0; sdrhip_device_count(&ndev);
  for (int two = 0; two < 2; two++) {
    if (two && ndev < 2) { std::printf("  (one device: the two-device RCCL case is skipped)\n"); continue; }
    int devs[2] = {0, two ? 1 : 0};
    sdrhip_comm *cm = 0;
    CHECK(sdrhip_comm_create(devs, 2, &cm) == SDRHIP_OK);
    if (!cm) { std::printf("  comm_create: %s\n", sdrhip_last_error()); continue; }
    const char *tr = ""; sdrhip_comm_transport(cm, &tr);
    CHECK(std::string(tr) == (two ? "rccl" : "same-device copies"));
    const size_t n = 1 << 20;   // 4 MiB per rank and step: long enough for a copy to still run when the next step starts
    sdrhip_ctx *ctx[2]; void *snd[2], *all[3], *des[2];
    for (int r = 0; r < 2; r++) { sdrhip_comm_ctx(cm, r, &ctx[r]); sdrhip_malloc(ctx[r], n * 4, &snd[r]); sdrhip_malloc(ctx[r], 4096, &des[r]); }
    for (int k = 0; k < 3; k++) sdrhip_malloc(ctx[0], 2 * n * 4, &all[k]);
    std::vector<uint32_t> design(1024); for (size_t i = 0; i < design.size(); i++) design[i] = uint32_t(i * 40503u + 7u);
    sdrhip_memcpy_h2d(ctx[0], des[0], design.data(), 4096); sdrhip_memset(ctx[1], des[1], 0, 4096);
    void *dd[2] = {des[0], des[1]};
    CHECK(sdrhip_comm_broadcast(cm, dd, 4096, 0) == SDRHIP_OK);
    sdrhip_memset(ctx[0], des[0], 0x55, 4096);   // the root reuses its buffer at once: the broadcast must have read it first
    const size_t bytes[2] = {n * 4, n * 4};
    const void *send[2] = {snd[0], snd[1]};
    for (int k = 0; k < 3; k++) {   // step k: every rank fills its buffer with the byte 0x10 * (k + 1) + rank, then the gather
      for (int r = 0; r < 2; r++) sdrhip_memset(ctx[r], snd[r], 0x10 * (k + 1) + r, n * 4);
      CHECK(sdrhip_comm_gather(cm, send, bytes, all[k], 0) == SDRHIP_OK);
    }
    CHECK(sdrhip_comm_synchronize(cm) == SDRHIP_OK);
    std::vector<uint32_t> got(1024);
    sdrhip_memcpy_d2h(ctx[1], got.data(), des[1], 4096);
    CHECK(got == design);
    std::vector<uint8_t> g(2 * n * 4);
    for (int k = 0; k < 3; k++) {
      sdrhip_memcpy_d2h(ctx[0], g.data(), all[k], g.size());
      bool ok = true;
      for (int r = 0; r < 2; r++) for (size_t i = 0; i < n * 4; i += 4093) ok = ok && g[r * n * 4 + i] == uint8_t(0x10 * (k + 1) + r);
      ok = ok && g[n * 4 - 1] == uint8_t(0x10 * (k + 1)) && g[2 * n * 4 - 1] == uint8_t(0x10 * (k + 1) + 1);
      CHECK(ok);
    }
    // the overlapped form (sdrhip_comm_gather_begin / _wait): double-buffered send buffers, step k waits for the gather of
    // step k - 2 before it refills buffer k & 1; six steps, every landing zone checked, then the slots drained
    { void *snd2[2][2], *land[6];
      for (int r = 0; r < 2; r++) for (int o = 0; o < 2; o++) sdrhip_malloc(ctx[r], n * 4, &snd2[r][o]);
      for (int k = 0; k < 6; k++) sdrhip_malloc(ctx[0], 2 * n * 4, &land[k]);
      for (int k = 0; k < 6; k++) {
        const int o = k & 1;
        CHECK(sdrhip_comm_gather_wait(cm, o) == SDRHIP_OK);
        for (int r = 0; r < 2; r++) sdrhip_memset(ctx[r], snd2[r][o], 0x20 + 8 * k + r, n * 4);
        const void *sp[2] = {snd2[0][o], snd2[1][o]};
        CHECK(sdrhip_comm_gather_begin(cm, o, sp, bytes, land[k], 0) == SDRHIP_OK);
      }
      CHECK(sdrhip_comm_gather_wait(cm, 0) == SDRHIP_OK && sdrhip_comm_gather_wait(cm, 1) == SDRHIP_OK);
      CHECK(sdrhip_comm_gather_wait(cm, 3) == SDRHIP_OK);   // (a slot never begun: no wait)
      CHECK(sdrhip_comm_gather_begin(cm, 9, send, bytes, land[0], 0) == SDRHIP_E_INVALID);
      CHECK(sdrhip_comm_synchronize(cm) == SDRHIP_OK);
      for (int k = 0; k < 6; k++) {
        sdrhip_memcpy_d2h(ctx[0], g.data(), land[k], g.size());
        bool ok = true;
        for (int r = 0; r < 2; r++) for (size_t i = 0; i < n * 4; i += 4093) ok = ok && g[r * n * 4 + i] == uint8_t(0x20 + 8 * k + r);
        CHECK(ok);
      }
      for (int r = 0; r < 2; r++) for (int o = 0; o < 2; o++) sdrhip_free(ctx[r], snd2[r][o]);
      for (int k = 0; k < 6; k++) sdrhip_free(ctx[0], land[k]); }
    for (int r = 0; r < 2; r++) { sdrhip_free(ctx[r], snd[r]); sdrhip_free(ctx[r], des[r]); }
    for (int k = 0; k < 3; k++) sdrhip_free(ctx[0], all[k]);
    CHECK(sdrhip_comm_destroy(cm) == SDRHIP_OK);
  }
}

// float nodes: FIRLowPass<cf32> -> SubSample<cf32>(8) and the FFT filter bank vs direct convolution
static void testFloatNodes() {
  const size_t N = 4096;
  IQSigGen<float> gen(FS, N); gen.addSine(100e3, 0.5, 0.0); gen.addSine(-300e3, 0.3, 0.3);
  Recorder<cf32> raw; gpu::FIRLowPass<cf32> fir(127, 100e3); gpu::SubSample<cf32> sub(size_t(8)); Recorder<cf32> out;
  gpu::FilterNode<float> bank(1024); Recorder<cf32> band;
  gen.connect(&raw, true); gen.connect(&fir, true); fir.connect(&sub, true); sub.connect(&out, true);
  Recorder<cf32> band2;
  gen.connect(bank.sink(), true); bank.addFilter(50e3, 150e3)->connect(&band, true);
  bank.addFilter(-350e3, -250e3)->connect(&band2, true);   // a second band behind the same forward transform
  for (int b = 0; b < 3; b++) gen.next();
  CHECK(out.data.size() == 3 * N / 8 && band.data.size() == 3 * N);
  std::vector<double> a(127); orc_fir_lowpass_design(127, 100e3, FS, a.data());
  void *f = orc_fir_create(a.data(), 127); void *s = orc_subsample_create(8);
  std::vector<float> y(2 * 3 * N), d(2 * 3 * N / 8);
  orc_fir_cf32_process(f, (const float *)raw.data.data(), 3 * N, y.data());
  orc_subsample_cf32_process(s, y.data(), 3 * N, d.data());
  double err = 0, mx = 0;
  for (size_t i = 0; i < out.data.size(); i++) {
    err = std::max(err, (double)std::abs(out.data[i] - cf32(d[2 * i], d[2 * i + 1]))); mx = std::max(mx, (double)std::abs(cf32(d[2 * i], d[2 * i + 1])));
  }
  CHECK(err / mx <= 1e-5);
  orc_fir_destroy(f); orc_subsample_destroy(s);
  std::vector<float> h(2 * 1024), K(4 * 1024), fo(2 * 1024);
  orc_fftfilt_design_h(1024, 50e3, 150e3, FS, h.data()); orc_fftfilt_design_K(1024, h.data(), K.data());
  void *ff = orc_fftfilt_create(1024, K.data());
  err = 0; mx = 0;
  for (size_t blk = 0; blk < 3 * N / 1024; blk++) {
    orc_fftfilt_process(ff, (const float *)&raw.data[blk * 1024], fo.data());
    for (size_t i = 0; i < 1024; i++) {
      const cf32 r(fo[2 * i], fo[2 * i + 1]);
      err = std::max(err, (double)std::abs(band.data[blk * 1024 + i] - r)); mx = std::max(mx, (double)std::abs(r));
    }
  }
  CHECK(err / mx <= 1e-5);
  orc_fftfilt_destroy(ff);
  orc_fftfilt_design_h(1024, -350e3, -250e3, FS, h.data()); orc_fftfilt_design_K(1024, h.data(), K.data());
  ff = orc_fftfilt_create(1024, K.data());
  err = 0; mx = 0;
  CHECK(band2.data.size() == 3 * N);
  for (size_t blk = 0; blk < 3 * N / 1024 && band2.data.size() == 3 * N; blk++) {
    orc_fftfilt_process(ff, (const float *)&raw.data[blk * 1024], fo.data());
    for (size_t i = 0; i < 1024; i++) {
      const cf32 r(fo[2 * i], fo[2 * i + 1]);
      err = std::max(err, (double)std::abs(band2.data[blk * 1024 + i] - r)); mx = std::max(mx, (double)std::abs(r));
    }
  }
  CHECK(err / mx <= 1e-5);
  orc_fftfilt_destroy(ff);
}

// the whole DSP of examples/sdr_fm.cc:38-53 on the GPU: cu8 -> [AutoCast+IQBaseBand] -> FMDemod -> FMDeemph,
// against the golden vector cut from the reference chain
static std::string g_golden = "tests/golden";
template <class T> static std::vector<T> slurp(const std::string &name) {
  std::vector<T> v; FILE *f = fopen((g_golden + "/" + name).c_str(), "rb");
  if (!f) return v;
  T tmp[1024]; size_t n;
  while ((n = fread(tmp, sizeof(T), 1024, f)) > 0) v.insert(v.end(), tmp, tmp + n);
  fclose(f); return v;
}
static void testSdrFmChainCu8() {
  typedef std::complex<uint8_t> cu8;
  std::vector<uint8_t> raw = slurp<uint8_t>("g9_iq_cu8.bin");
  std::vector<int16_t> ref = slurp<int16_t>("g9_cu8_iqbb21d8_fm_deemph.bin");
  CHECK(raw.size() == 3 * 4096 * 2 && ref.size() == 3 * 512 - 1);
  struct U8Feeder : public Source { void cfg() { setConfig(Config(Config::Type_cu8, 1e6, 4096, 1)); }
                                     void feed(cu8 *p, size_t n) { Buffer<cu8> b(p, n); send(b, false); } } src;
  src.cfg();
  gpu::IQBaseBand<uint8_t> bb(100e3, 100e3, 50e3, 21, 8);
  gpu::FMDemod<int16_t> fm; gpu::FMDeemph<int16_t> de; Recorder<int16_t> out;
  src.connect(&bb, true); bb.connect(&fm, true); fm.connect(&de, true); de.connect(&out, true);
  CHECK(de.sampleRate() == 125000.0);
  for (int b = 0; b < 3; b++) src.feed(reinterpret_cast<cu8 *>(&raw[b * 8192]), 4096);
  CHECK(out.data == ref);
}

// real-input BaseBand<int16_t> node against the golden vector cut from the reference node (order 21, /8)
static void testRealBaseBand() {
  std::vector<int16_t> x = slurp<int16_t>("g10_real_in.bin"), ref = slurp<int16_t>("g10_bb21d8_out.bin");
  CHECK(x.size() == 3 * 4096 && ref.size() == 2 * 3 * 512);
  struct S16Feeder : public Source { void cfg() { setConfig(Config(Config::Type_s16, 1e6, 4096, 1)); }
                                     void feed(int16_t *p, size_t n) { Buffer<int16_t> b(p, n); send(b, false); } } src;
  src.cfg();
  gpu::BaseBand<int16_t> bb(100e3, 100e3, 50e3, 21, 8);
  Recorder<cs16> out;
  src.connect(&bb, true); bb.connect(&out, true);
  CHECK(bb.type() == Config::Type_cs16 && bb.Source::sampleRate() == 125000.0);
  for (int b = 0; b < 3; b++) src.feed(&x[b * 4096], 4096);
  CHECK(out.data.size() * 2 == ref.size() && 0 == memcmp(out.data.data(), ref.data(), ref.size() * 2));
  // a wrong input type is a ConfigError, as in the reference (:371-377)
  struct CFeeder : public Source { void cfg() { setConfig(Config(Config::Type_cs16, 1e6, 4096, 1)); } } bad;
  gpu::BaseBand<int16_t> bb2(100e3, 50e3, 21, 8);
  bool thrown = false;
  try { bad.cfg(); bad.connect(&bb2, true); } catch (ConfigError &) { thrown = true; }
  CHECK(thrown);
}

// the real-input node retuned and reconfigured between buffers (golden g16: setFrequencyShift, then a new source buffer size)
static void testRealBaseBandRetune() {
  std::vector<int16_t> x = slurp<int16_t>("g10_real_in.bin"), ref = slurp<int16_t>("g16_bb_real_retune_out.bin");
  CHECK(x.size() == 3 * 4096 && ref.size() == 2 * 1524);
  struct S16Feeder : public Source { void cfg(size_t bs) { setConfig(Config(Config::Type_s16, 1e6, bs, 1)); }
                                     void feed(int16_t *p, size_t n) { Buffer<int16_t> b(p, n); send(b, false); } } src;
  src.cfg(4096);
  gpu::BaseBand<int16_t> bb(100e3, 100e3, 50e3, 127, 8);
  Recorder<cs16> out;
  src.connect(&bb, true); bb.connect(&out, true);
  size_t off = 0;
  src.feed(&x[off], 4096); off += 4096; src.feed(&x[off], 1000); off += 1000;
  bb.setFrequencyShift(-150e3);
  CHECK(bb.frequencyShift() == -150e3);
  src.feed(&x[off], 3000); off += 3000;
  src.cfg(2048);
  src.feed(&x[off], 2048); off += 2048; src.feed(&x[off], 2048);
  CHECK(out.data.size() * 2 == ref.size() && 0 == memcmp(out.data.data(), ref.data(), ref.size() * 2));
}

// the node retuned between buffers, against the golden vector cut from the reference node doing the same
// (setCenterFrequency: LUT phase restarts only; setFilterFrequency / setFilterWidth: kernel only; setSubsample: _reconfigure)
static void testRetuneMidStream() {
  std::vector<int16_t> xin = slurp<int16_t>("g1_iq_cs16.bin"), ref = slurp<int16_t>("g12_retune_out.bin"), reff = slurp<int16_t>("g12_retune_fm.bin");
  CHECK(xin.size() == 2 * 16384 && ref.size() == 2 * 2046 && reff.size() == 2046);
  for (int fused = 0; fused < 3; fused++) {   // 0: complex out; 1: gpu::FMDemod behind the node; 2: FM fused into the launch
    Feeder src; src.cfg(Config::Type_cs16, 4096);
    gpu::IQBaseBand<int16_t> bb(100e3, 100e3, 50e3, 127, 8);
    gpu::FMDemod<int16_t> fm; Recorder<cs16> out; Recorder<int16_t> outf;
    if (fused == 2) bb.setDemod(SDRHIP_EPI_FM);
    src.connect(&bb, true);
    if (fused == 0) bb.connect(&out, true);
    else if (fused == 1) { bb.connect(&fm, true); fm.connect(&outf, true); }
    else bb.connect(&outf, true);
    cs16 *x = reinterpret_cast<cs16 *>(xin.data());
    size_t off = 0;
    src.feed(x + off, 4096); off += 4096; src.feed(x + off, 3000); off += 3000;
    bb.setCenterFrequency(-150e3);
    src.feed(x + off, 2000); off += 2000;
    bb.setFilterFrequency(-150e3); bb.setFilterWidth(30e3);
    src.feed(x + off, 3192); off += 3192;
    bb.setSubsample(8);
    src.feed(x + off, 4096);
    if (fused == 0) CHECK(out.data.size() * 2 == ref.size() && 0 == memcmp(out.data.data(), ref.data(), ref.size() * 2));
    else CHECK(outf.data == reff);
  }
}

// the node's GEOMETRY changed between buffers, against the golden vector cut from the reference node doing the same:
// setSubsample(4), setOutputSampleRate(100e3) (÷24), a new source buffer size (all _reconfigure: ring kept) and
// setOrder(161) (kernel and ring only). Every change needs a new device plan; nothing but the outputs the reference
// itself leaves undefined behind setOrder (`undefined_head` = 10 of the last feed) may differ.
static void testRegeometryMidStream() {
  std::vector<int16_t> xin = slurp<int16_t>("g1_iq_cs16.bin"), ref = slurp<int16_t>("g14_regeom_out.bin"), reff = slurp<int16_t>("g14_regeom_fm.bin");
  CHECK(xin.size() == 2 * 16384 && ref.size() == 2 * 1687 && reff.size() == 1687);
  const size_t undef_lo = 1687 - 85, undef_hi = undef_lo + 10;
  for (int fused = 0; fused < 3; fused++) {   // 0: complex out; 1: gpu::FMDemod behind the node; 2: FM fused into the launch
    Feeder src; src.cfg(Config::Type_cs16, 4096);
    gpu::IQBaseBand<int16_t> bb(100e3, 100e3, 50e3, 127, 8);
    gpu::FMDemod<int16_t> fm; Recorder<cs16> out; Recorder<int16_t> outf;
    if (fused == 2) bb.setDemod(SDRHIP_EPI_FM);
    src.connect(&bb, true);
    if (fused == 0) bb.connect(&out, true);
    else if (fused == 1) { bb.connect(&fm, true); fm.connect(&outf, true); }
    else bb.connect(&outf, true);
    cs16 *x = reinterpret_cast<cs16 *>(xin.data());
    size_t off = 0;
    src.feed(x + off, 4096); off += 4096; src.feed(x + off, 3000); off += 3000;
    bb.setSubsample(4);
    CHECK(bb.Source::sampleRate() == 600000.0);
    src.feed(x + off, 2000); off += 2000;
    bb.setOutputSampleRate(100e3);
    CHECK(bb.subSample() == 24 && bb.Source::sampleRate() == 100000.0);
    src.feed(x + off, 3192); off += 3192;
    src.cfg(Config::Type_cs16, 2048);
    src.feed(x + off, 2048); off += 2048;
    bb.setOrder(161);
    CHECK(bb.order() == 161);
    src.feed(x + off, 2048);
    bool same = true;
    if (fused == 0) {
      CHECK(out.data.size() * 2 == ref.size());
      for (size_t i = 0; i < out.data.size() && same; i++)
        if ((i < undef_lo || i >= undef_hi) && (out.data[i].real() != ref[2 * i] || out.data[i].imag() != ref[2 * i + 1])) { same = false; std::printf("regeometry: output %zu differs\n", i); }
    } else {
      CHECK(outf.data.size() == reff.size());
      for (size_t i = 0; i < outf.data.size() && i < reff.size() && same; i++)
        if ((i < undef_lo || i >= undef_hi) && outf.data[i] != reff[i]) { same = false; std::printf("regeometry (fm %d): output %zu differs\n", fused, i); }
    }
    CHECK(same);
  }
}

// the documentation example's chain (reference src/sdr.hh:225-240) on the GPU nodes: IQBaseBand<int8_t>(0, 100e3, 16, 0, 100e3)
// -> FMDemod<int8_t,int16_t>, against the golden vector cut from the reference chain
static void testInt8Chain() {
  typedef std::complex<int8_t> cs8;
  std::vector<int8_t> raw = slurp<int8_t>("g13_iq_cs8.bin");
  std::vector<int16_t> ref = slurp<int16_t>("g13_i8_doc_o16_fm.bin");
  CHECK(raw.size() == 2 * 3 * 4096 && ref.size() == 511);
  struct S8Feeder : public Source { void cfg() { setConfig(Config(Config::Type_cs8, FS, 4096, 1)); }
                                    void feed(cs8 *p, size_t n) { Buffer<cs8> b(p, n); send(b, false); } } src;
  src.cfg();
  gpu::IQBaseBand<int8_t> bb(0.0, 100e3, 16, 0, 100e3);
  gpu::FMDemod<int8_t, int16_t> fm; Recorder<int16_t> out;
  src.connect(&bb, true); bb.connect(&fm, true); fm.connect(&out, true);
  CHECK(bb.type() == Config::Type_cs8 && fm.type() == Config::Type_s16 && bb.subSample() == 24);
  cs8 *x = reinterpret_cast<cs8 *>(raw.data());
  const size_t chunks[4] = {4096, 1000, 3096, 4096};
  size_t off = 0;
  for (int k = 0; k < 4; k++) { src.feed(x + off, chunks[k]); off += chunks[k]; }
  CHECK(out.data == ref);
}

// FFT::exec / FFTPlan<float|double> on host buffers against a direct O(n^2) DFT in long double
// BASELINE config 2 through the node API: gpu::IQBaseBand<float> (build-defined: shift -> FIRLowPass<cf32>(order, width/2)
// -> /D) against the oracle chain freqshift (float64 closed form) -> FIR cf32 -> SubSample, <= 1e-5; config / drop rules.
static void testFloatBaseBandNode() {
  const size_t N = 4096, NB = 3;
  IQSigGen<float> gen(FS, N); gen.addSine(100e3, 0.5, 0.0); gen.addSine(-300e3, 0.3, 0.3);
  gpu::IQBaseBand<float> bb(100e3, 100e3, 200e3, 127, 8);
  Recorder<cf32> raw, out;
  gen.connect(&raw, true); gen.connect(&bb, true); bb.connect(&out, true);
  CHECK(bb.Source::sampleRate() == FS / 8 && bb.type() == Config::Type_cf32);
  for (size_t b = 0; b < NB; b++) gen.next();
  CHECK(out.data.size() == NB * N / 8);
  std::vector<double> a(127); orc_fir_lowpass_design(127, 100e3, FS, a.data());
  void *f = orc_fir_create(a.data(), 127); void *s = orc_subsample_create(8);
  std::vector<float> sh(2 * NB * N), y(2 * NB * N), d(2 * NB * N / 8);
  orc_freqshift_cf32((const float *)raw.data.data(), NB * N, 0, 100e3, FS, sh.data());
  orc_fir_cf32_process(f, sh.data(), NB * N, y.data());
  const size_t nd = orc_subsample_cf32_process(s, y.data(), NB * N, d.data());
  orc_fir_destroy(f); orc_subsample_destroy(s);
  CHECK(nd == out.data.size());
  double err = 0, mx = 0;
  for (size_t i = 0; i < std::min(nd, out.data.size()); i++) {
    err = std::max(err, (double)std::abs(out.data[i] - cf32(d[2 * i], d[2 * i + 1]))); mx = std::max(mx, (double)std::abs(cf32(d[2 * i], d[2 * i + 1])));
  }
  CHECK(mx > 0.05 && err <= 1e-5 * mx);
  // a filter that is not centred on the shift frequency is refused, a wrong input type too
  bool threw = false;
  try { gpu::IQBaseBand<float> off(100e3, 90e3, 50e3, 127, 8); off.config(Config(Config::Type_cf32, FS, N, 1)); } catch (ConfigError &) { threw = true; }
  CHECK(threw);
  threw = false;
  try { gpu::IQBaseBand<float> t(100e3, 50e3, 127, 8); t.config(Config(Config::Type_cs16, FS, N, 1)); } catch (ConfigError &) { threw = true; }
  CHECK(threw);
  {   // a setter that would leave an unsupported (Fc, Ff) pair throws BEFORE it changes anything; a retune is one call
    gpu::IQBaseBand<float> t(100e3, 50e3, 127, 8);
    t.config(Config(Config::Type_cf32, FS, N, 1));
    threw = false;
    try { t.setFilterFrequency(80e3); } catch (ConfigError &) { threw = true; }
    CHECK(threw && t.filterFrequency() == 100e3 && t.centerFrequency() == 100e3);
    t.setCenterFrequency(80e3);
    CHECK(t.filterFrequency() == 80e3 && t.centerFrequency() == 80e3);
    t.setFilterFrequency(80e3);   // (equal to the centre: fine)
  }
  {   // setters that change neither order, decimation nor buffer size keep the plan and its stream: the decimator's phase
      // carries across them (5 samples in, then a retune / a new width, then 11 more: 16 samples = 2 outputs in all)
    gpu::IQBaseBand<float> t(100e3, 50e3, 127, 8);
    Feeder src; src.cfg(Config::Type_cf32, 64); Recorder<cf32> rec;
    src.connect(&t, true); t.connect(&rec, true);
    std::vector<cf32> x(64, cf32(0.25f, -0.5f));
    src.feed(x.data(), 5);
    t.setCenterFrequency(80e3); t.setFilterWidth(30e3); t.setSubsample(8); t.setOrder(127);
    src.feed(x.data(), 11);
    CHECK(rec.data.size() == 2);
    t.setSubsample(4);   // a new decimation IS a new plan: counters restart
    src.feed(x.data(), 5);
    CHECK(rec.data.size() == 3 && t.Source::sampleRate() == FS / 4);
  }
  // output rate given instead of the decimation (src/baseband.hh:159-162): 2.4 MS/s -> 300 kS/s = /8
  gpu::IQBaseBand<float> byrate(100e3, 200e3, 127, 1, 300e3);
  byrate.config(Config(Config::Type_cf32, FS, N, 1));
  CHECK(byrate.subSample() == 8 && byrate.Source::sampleRate() == 300e3);
}

template <class Scalar>
static void testFftPlanOf(size_t n, double tol) {
  typedef std::complex<Scalar> CS;
  Buffer<CS> in(n), out(n);
  for (size_t i = 0; i < n; i++) in[i] = CS(Scalar(std::sin(0.37 * i) + 0.25 * std::cos(1.9 * i)), Scalar(std::cos(0.11 * i * i)));
  gpu::FFT::exec<Scalar>(in, out, gpu::FFT::FORWARD);
  double worst = 0, scale = 0;
  for (size_t k = 0; k < n; k += (n > 8192 ? n / 61 : n > 256 ? 37 : 1)) {   // sampled bins for the larger sizes
    std::complex<long double> acc(0, 0);
    for (size_t i = 0; i < n; i++) {
      const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)((i * k) % n) / (long double)n;
      acc += std::complex<long double>(in[i].real(), in[i].imag()) * std::complex<long double>(cosl(ang), sinl(ang));
    }
    worst = std::max(worst, (double)std::abs(acc - std::complex<long double>(out[k].real(), out[k].imag())));
    scale = std::max(scale, (double)std::abs(acc));
  }
  CHECK(worst <= tol * scale);
  gpu::FFTPlan<Scalar> back(out, gpu::FFT::BACKWARD);   // the in-place plan: backward of the spectrum = n * x
  back();
  double werr = 0;
  for (size_t i = 0; i < n; i++) werr = std::max(werr, (double)std::abs(out[i] / Scalar(n) - in[i]));
  CHECK(werr <= 20 * tol);
}
static void testFftPlan() {
  testFftPlanOf<float>(64, 2e-6); testFftPlanOf<float>(4096, 2e-6);
  testFftPlanOf<double>(64, 1e-13); testFftPlanOf<double>(8192, 1e-13);
  // any size, as the reference plans whatever in.size() is (src/fftplan_fftw3.hh:34-36): FilterNode(1000)'s 2000 points,
  // 3000, 5000, and factors up to 13
  testFftPlanOf<float>(2000, 3e-6); testFftPlanOf<float>(3000, 3e-6); testFftPlanOf<float>(5000, 3e-6); testFftPlanOf<float>(1001, 3e-6);
  testFftPlanOf<double>(1000, 1e-13); testFftPlanOf<double>(6006, 1e-13);
  testFftPlanOf<float>(1003, 1e-5); testFftPlanOf<double>(2053, 1e-12);   // a prime factor above 13: the chirp transform
  testFftPlanOf<float>(65536, 3e-6); testFftPlanOf<double>(30000, 1e-12);   // longer than the LDS holds: the four-step plan
  // a chirp transform beyond one workgroup's LDS runs over a four-step plan (round 4 refused these at construction)
  testFftPlanOf<double>(4099, 1e-12); testFftPlanOf<float>(2 * 16411, 1e-5);
  { Buffer< std::complex<float> > a(20014), b(20014); gpu::FFTPlan<float> p(a, b, gpu::FFT::FORWARD);
    CHECK(std::string(p.form()) == "chirp over four-step");
    for (size_t i = 0; i < a.size(); i++) a[i] = std::complex<float>(float(i % 7) - 3.f, float(i % 5) - 2.f);
    p(); const std::complex<float> first = b[1]; p();   // planned once, executed twice
    CHECK(b[1] == first); }
  bool threw = false;
  try { Buffer< std::complex<float> > a(64), b(128); gpu::FFTPlan<float> p(a, b, gpu::FFT::FORWARD); } catch (ConfigError &) { threw = true; }
  CHECK(threw);   // sizes differ (the reference's check)
}

// FilterNode for a block size that is not a power of two and for Scalar = double (src/filternode.hh:230-245), against the
// closed form of the reference's overlap-add filter: y = h (*) x / (sqrt(2N) ||h||_2)
template <class Scalar>
static void testFilterNodeOf(size_t N, double tol) {
  typedef std::complex<Scalar> CS;
  const size_t nblk = 4;
  IQSigGen<Scalar> gen(FS, N); gen.addSine(100e3, 0.5, 0.0); gen.addSine(-300e3, 0.3, 0.3);
  Recorder<CS> raw, band, band2;
  gpu::FilterNode<Scalar> bank(N);
  gen.connect(&raw, true); gen.connect(bank.sink(), true);
  bank.addFilter(50e3, 150e3)->connect(&band, true);
  typename gpu::FilterNode<Scalar>::Band *b2 = bank.addFilter(-350e3, -250e3);
  b2->connect(&band2, true);
  for (size_t b = 0; b < nblk; b++) gen.next();
  CHECK(band.data.size() == nblk * N && band2.data.size() == nblk * N && b2->Source::sampleRate() == FS);
  const double lo[2] = {50e3, -350e3}, hi[2] = {150e3, -250e3};
  for (int k = 0; k < 2; k++) {
    std::vector<Scalar> h(2 * N);
    gpu::design::fftFilterKernel(int(N), lo[k], hi[k], FS, h.data());
    long double e = 0;
    for (size_t i = 0; i < 2 * N; i++) e += (long double)h[i] * h[i];
    const long double sc = 1.0L / (sqrtl((long double)(2 * N)) * sqrtl(e));
    const std::vector<CS> &y = k ? band2.data : band.data;
    double err = 0, mx = 0;
    for (size_t n = 0; n < y.size() && n < raw.data.size(); n += (N > 4096 ? 397 : 7)) {   // sampled outputs
      std::complex<long double> acc(0, 0);
      for (size_t t = 0; t < N && t <= n; t++)
        acc += std::complex<long double>(h[2 * t], h[2 * t + 1]) * std::complex<long double>(raw.data[n - t].real(), raw.data[n - t].imag());
      acc *= sc;
      err = std::max(err, (double)std::abs(acc - std::complex<long double>(y[n].real(), y[n].imag())));
      mx = std::max(mx, (double)std::abs(acc));
    }
    if (!(err <= tol * mx)) std::printf("  FilterNode<%s>(%zu) band %d: error %.3g of %.3g\n", sizeof(Scalar) == 8 ? "double" : "float", N, k, err, mx);
    CHECK(err <= tol * mx);
  }
}
static void testFilterNodeAnySizeAndDouble() {
  testFilterNodeOf<float>(1000, 1e-5);
  testFilterNodeOf<double>(1024, 1e-12);
  testFilterNodeOf<double>(1000, 1e-12);
  // block sizes whose 2N-point transform does not fit one workgroup's LDS (four-step) or has a prime factor above 13 (chirp
  // transform): the reference takes any block_size (src/filternode.hh:236-245, FFTW plans any n)
  testFilterNodeOf<float>(16384, 1e-5);
  testFilterNodeOf<float>(12000, 1e-5);
  testFilterNodeOf<float>(1009, 1e-5);
  testFilterNodeOf<double>(8192, 1e-11);
}

// The HOST half of the nodes (no device needed; tests/test_cpp.py builds this file with -fsanitize=address,undefined
// and runs `--host-only` in the build container): the designers in design.hh against the golden taps / LUT / low-pass,
// and the config() rules every node follows before it ever touches the device (reference src/baseband.hh:115-132,
// src/firfilter.hh:172-207, src/demod.hh:195-226): silent return while the upstream Config is incomplete, ConfigError on
// a type mismatch; with a complete Config either the plan is made (a GPU is present) or a ConfigError says that there
// is no device and no CPU fallback — never a crash, and every node destructs cleanly either way.
template <class Node> static void hostConfigRules(Node &n, Config::Type good, Config::Type bad, const char *what) {
  n.config(Config());                                  // nothing known yet: silent
  n.config(Config(good, 0, 0, 1));                     // type only: still silent
  bool threw = false;
  try { n.config(Config(bad, FS, 4096, 1)); } catch (ConfigError &) { threw = true; }
  if (!threw) std::printf("  %s accepted a wrong input type\n", what);
  CHECK(threw);
  try { n.config(Config(good, FS, 4096, 1)); } catch (ConfigError &e) { (void)e; }   // no device here: a ConfigError, not a crash
}
static void testHostOnly() {
  std::vector<int32_t> taps(254), lut(256);
  gpu::design::iqbbTaps(100e3, 50e3, FS, 127, taps.data());
  gpu::design::freqShiftLutI16(lut.data());
  CHECK(taps == slurp<int32_t>("g3_iqbb127d8_taps.bin") && lut == slurp<int32_t>("g3_iqbb127d8_lut.bin"));
  CHECK(gpu::design::freqShiftIncrement(100e3, FS) == 1365u && gpu::design::iqbbDecimation(FS, 8, 0.0) == 8);
  CHECK(gpu::design::iqbbDecimation(2.4e6, 1, 8000.0) == 300);
  for (int order : {127, 255, 4097}) {
    std::vector<double> a(order);
    gpu::design::firLowPass(order, 100e3, FS, a.data());
    CHECK(a == slurp<double>("g2_firlp_alpha" + std::to_string(order) + ".bin"));
  }
  { gpu::IQBaseBand<int16_t> n(100e3, 100e3, 50e3, 127, 8); hostConfigRules(n, Config::Type_cs16, Config::Type_cf32, "IQBaseBand<int16_t>"); }
  { gpu::IQBaseBand<float> n(100e3, 200e3, 127, 8); hostConfigRules(n, Config::Type_cf32, Config::Type_cs16, "IQBaseBand<float>"); }
  { gpu::IQBaseBand<uint8_t> n(100e3, 100e3, 50e3, 21, 8); hostConfigRules(n, Config::Type_cu8, Config::Type_cs16, "IQBaseBand<uint8_t>"); }
  { gpu::BaseBand<int16_t> n(100e3, 100e3, 50e3, 64, 8); hostConfigRules(n, Config::Type_s16, Config::Type_cs16, "BaseBand<int16_t>"); }
  { gpu::FIRLowPass<cs16> n(127, 100e3); hostConfigRules(n, Config::Type_cs16, Config::Type_s16, "FIRLowPass<cs16>"); }
  { gpu::FIRLowPass<cf32> n(127, 100e3); hostConfigRules(n, Config::Type_cf32, Config::Type_cs16, "FIRLowPass<cf32>"); }
  { gpu::FMDemod<int16_t> n; hostConfigRules(n, Config::Type_cs16, Config::Type_cf32, "FMDemod<int16_t>"); }
  { gpu::AMDemod<int16_t> n; hostConfigRules(n, Config::Type_cs16, Config::Type_cf32, "AMDemod<int16_t>"); }
  { gpu::USBDemod<float> n; hostConfigRules(n, Config::Type_cf32, Config::Type_cs16, "USBDemod<float>"); }
  { gpu::SubSample<cs16> n(size_t(8)); hostConfigRules(n, Config::Type_cs16, Config::Type_cf32, "SubSample<cs16>"); }
  { gpu::FMDeemph<int16_t> n; hostConfigRules(n, Config::Type_s16, Config::Type_cs16, "FMDeemph<int16_t>"); }
  {   // the designers' own DFT (the spectrum of a FilterNode kernel, any 2N): composite, small-prime and large-prime lengths
      // (a large prime runs the chirp transform) against the direct sum, both signs; the f64 golden spectrum of N = 1000
    for (size_t n : {12u, 61u, 2018u, 20014u}) {
      std::vector< std::complex<double> > a(n), b;
      for (size_t i = 0; i < n; i++) a[i] = std::complex<double>(std::sin(0.37 * i) + 0.25, std::cos(0.011 * i * i));
      for (int sign = -1; sign <= 1; sign += 2) {
        b = a; gpu::design::dft(b, sign);
        double worst = 0, scale = 0;
        for (size_t k = 0; k < n; k += (n > 256 ? n / 7 : 1)) {
          std::complex<long double> acc(0, 0);
          for (size_t i = 0; i < n; i++) {
            const long double ang = sign * 2.0L * 3.14159265358979323846264338327950288L * (long double)((i * k) % n) / (long double)n;
            acc += std::complex<long double>(a[i].real(), a[i].imag()) * std::complex<long double>(cosl(ang), sinl(ang));
          }
          worst = std::max(worst, (double)std::abs(acc - std::complex<long double>(b[k].real(), b[k].imag())));
          scale = std::max(scale, (double)std::abs(acc));
        }
        if (!(worst <= 1e-11 * scale)) std::printf("  design::dft(%zu, %d): error %.3g of %.3g\n", n, sign, worst, scale);
        CHECK(worst <= 1e-11 * scale);
      }
    }
  }
  {   // buffers and views as the nodes hand them on (ownership rules of src/buffer.hh:54-104)
    Buffer<cs16> b(64); CHECK(b.isUnused());
    Buffer<cs16> v = b.head(10); b.ref(); CHECK(!b.isUnused() && v.size() == 10); b.unref(); CHECK(b.isUnused());
    b.unref();
  }
}

int main(int argc, char **argv) {
  if (argc > 1 && std::string(argv[1]) == "--host-only") {
    if (argc > 2) g_golden = argv[2];
    Logger::get().addHandler(new StreamLogHandler(std::cerr, LOG_WARNING));
    try { testHostOnly(); } catch (std::exception &e) { std::printf("FAIL: exception: %s\n", e.what()); return 2; }
    std::printf("%s (%d failures)\n", failures ? "FAILED" : "OK", failures);
    return failures ? 1 : 0;
  }
  if (argc > 1) g_golden = argv[1];
  Logger::get().addHandler(new StreamLogHandler(std::cerr, LOG_WARNING));
  try {
    testBasebandFmChain();
    testFirFmOnQueue();
    testOwnership();
    testChannelBank();
    testChannelBankDropsWhileOutputInUse();
    testChannelBankMultiRank();
    testCommCalls();
    testCommPipelinedAndTwoDevices();
    testHostOnly();
    testFloatNodes();
    testFloatBaseBandNode();
    testSdrFmChainCu8();
    testRealBaseBand();
    testRealBaseBandRetune();
    testRetuneMidStream();
    testRegeometryMidStream();
    testInt8Chain();
    testFftPlan();
    testFilterNodeAnySizeAndDouble();
    testFirSetFreqKeepsTheRing();
  } catch (std::exception &e) {
    std::printf("FAIL: exception: %s\n", e.what());
    return 2;
  }
  std::printf("%s (%d failures)\n", failures ? "FAILED" : "OK", failures);
  return failures ? 1 : 0;
}
