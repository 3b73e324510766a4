// sdr_rec_wav — the receiver of the reference's examples/sdr_rec.cc, headless and file to file:
// the RTL-SDR source and the PortAudio sink (hardware) are replaced by WavSource / WavSink, the DSP runs on the
// MI355X nodes of include/sdr/gpu/nodes.hh. Mode table and wiring follow examples/sdr_rec.cc:43-109
// (baseband order 16, output rate 12 kS/s, 48 kS/s for WFM; FM demodulates in place over the baseband's
// buffer and is followed by the de-emphasis; AM / SSB demodulators sit on queued edges).
//
//   sdr_rec_wav INPUT.wav MODE OUTPUT.wav [BUFFER_SIZE]      MODE = WFM | NFM | AM | USB | LSB; default 65536 samples
//   INPUT.wav: 2-channel PCM, 8 bit (complex<uint8_t>, RTL-SDR bytes) or 16 bit (complex<int16_t>)
//
// build: g++ -O2 -std=c++17 -Iinclude examples/sdr_rec_wav.cc -Llibsdr_amd -lsdrhip -lpthread
#include <cstdlib>
#include <iostream>
#include <memory>
#include <string>

#include "sdr/sdr.hh"

using namespace sdr;

template <class SIn>
static int run(WavSource &src, const std::string &mode, const std::string &outFile) {
  double f_center = 0, f_filter = 0, flt_width = 0, out_f_sample = 12e3;
  const int sub_sample = 1;
  if (mode == "WFM") { flt_width = 50e3; out_f_sample = 48e3; }
  else if (mode == "NFM") { flt_width = 12.5e3; }
  else if (mode == "AM") { flt_width = 15e3; }
  else if (mode == "USB") { f_filter = 1500; flt_width = 3e3; }
  else if (mode == "LSB") { f_filter = -1500; flt_width = 3e3; }
  else {
    std::cerr << "Unknown mode '" << mode << "': Possible values are WFM, NFM, AM, USB, LSB." << std::endl;
    return -1;
  }

  gpu::IQBaseBand<SIn> baseband(f_center, f_filter, flt_width, 16, sub_sample, out_f_sample);
  std::unique_ptr< gpu::FMDemod<int16_t> > fm_demod;
  std::unique_ptr< gpu::FMDeemph<int16_t> > fm_deemph;
  std::unique_ptr< gpu::AMDemod<int16_t> > am_demod;
  std::unique_ptr< gpu::USBDemod<int16_t> > usb_demod;
  WavSink<int16_t> wav_sink(outFile);

  src.connect(&baseband);   // queued, as cast -> baseband in the reference
  if (mode == "WFM" || mode == "NFM") {
    fm_demod.reset(new gpu::FMDemod<int16_t>());
    fm_deemph.reset(new gpu::FMDeemph<int16_t>());
    baseband.connect(fm_demod.get(), true);
    fm_demod->connect(fm_deemph.get(), true);
    fm_deemph->connect(&wav_sink);
  } else if (mode == "AM") {
    am_demod.reset(new gpu::AMDemod<int16_t>());
    baseband.connect(am_demod.get());
    am_demod->connect(&wav_sink);
  } else {
    usb_demod.reset(new gpu::USBDemod<int16_t>());
    baseband.connect(usb_demod.get());
    usb_demod->connect(&wav_sink);
  }

  Queue &queue = Queue::get();
  queue.addIdle(&src, &WavSource::next);   // read the next buffer whenever the queue runs dry
  src.addEOS(&queue, &Queue::stop);
  queue.start();
  queue.wait();
  queue.remIdle(&src);
  wav_sink.close();
  std::cerr << "Demodulated " << src.frameCount() << " samples (" << mode << ") into " << outFile << std::endl;
  return 0;
}

int main(int argc, char *argv[]) {
  if (4 > argc) {
    std::cout << "USAGE: sdr_rec_wav INPUT.wav MODE OUTPUT.wav [BUFFER_SIZE]" << std::endl;
    return -1;
  }
  Logger::get().addHandler(new StreamLogHandler(std::cerr, LOG_WARNING));
  try {
    const size_t bufferSize = argc > 4 ? std::max(1L, atol(argv[4])) : 65536;
    WavSource src(argv[1], bufferSize);
    if (!src.isOpen()) { std::cerr << "Can not open " << argv[1] << std::endl; return -1; }
    if (src.type() == Config::Type_cu8) return run<uint8_t>(src, argv[2], argv[3]);
    if (src.type() == Config::Type_cs16) return run<int16_t>(src, argv[2], argv[3]);
    std::cerr << "Input must be a 2-channel (I/Q) recording, 8 or 16 bit." << std::endl;
    return -1;
  } catch (std::exception &e) {
    std::cerr << "Error: " << e.what() << std::endl;
    return -2;
  }
}
