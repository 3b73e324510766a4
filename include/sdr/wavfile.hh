// wavfile.hh — PCM WAV reader / writer nodes with the reference's interface and on-disk behaviour
// (SURVEY §8f row 4). Written from scratch around one Header record that is parsed / emitted as a whole.
//
// Interface mirrored (reference, file:line):
//   WavSink<Scalar>   src/wavfile.hh:15-124   ctor(filename), config(), process(), close()
//   WavSource         src/wavfile.hh:130-171, src/wavfile.cc:9-238   open(), close(), isOpen(), isReal(), next()
// Behaviour kept on purpose (files written here are byte-identical to the reference's, tests/test_wav.py):
//   * the sink reserves 44 bytes and writes the header on close(); its RIFF size field is 36 + 2*frames
//     whatever the sample type (src/wavfile.hh:88), the data size is channels*frames*bytes (:103);
//   * the source accepts PCM (format 1), 1 or 2 channels, 8 or 16 bits, `fmt ` directly behind `WAVE`,
//     skips unknown chunks up to `data`, maps (channels, bits) to u8 / s16 / cu8 / cs16 (src/wavfile.cc:141-148),
//     sends views of ONE internal buffer with allow_overwrite = true and signals EOS on the call after the
//     last frame (:203-208).
#ifndef SDR_WAVFILE_HH
#define SDR_WAVFILE_HH

#include <cstdint>
#include <cstring>
#include <fstream>
#include <string>

#include "node.hh"
#include "logger.hh"

namespace sdr {
namespace wav {

/** The 44-byte canonical PCM header, as numbers. */
struct Header {
  uint16_t channels, bitsPerSample;
  uint32_t sampleRate, riffSize, dataBytes;
  Header() : channels(0), bitsPerSample(0), sampleRate(0), riffSize(0), dataBytes(0) {}
  inline uint16_t frameBytes() const { return uint16_t(channels * (bitsPerSample / 8)); }

  /** Serialises into 44 little-endian bytes (this library only targets little-endian hosts, as the reference). */
  void emit(char *out) const {
    const uint32_t fmtSize = 16, byteRate = uint32_t(channels) * sampleRate * (bitsPerSample / 8);
    const uint16_t pcm = 1, align = frameBytes();
    char *p = out;
    put(p, "RIFF", 4); put(p, &riffSize, 4); put(p, "WAVE", 4);
    put(p, "fmt ", 4); put(p, &fmtSize, 4); put(p, &pcm, 2); put(p, &channels, 2); put(p, &sampleRate, 4);
    put(p, &byteRate, 4); put(p, &align, 2); put(p, &bitsPerSample, 2);
    put(p, "data", 4); put(p, &dataBytes, 4);
  }

private:
  static void put(char *&p, const void *src, size_t n) { std::memcpy(p, src, n); p += n; }
};

/** Maps a WAV layout onto the stream type the reference announces for it. */
inline Config::Type streamType(uint16_t channels, uint16_t bits) {
  if (channels == 1) return bits == 8 ? Config::Type_u8 : Config::Type_s16;
  return bits == 8 ? Config::Type_cu8 : Config::Type_cs16;
}

}  // namespace wav


/** Stores a stream of integer samples as a PCM WAV file. */
template <class Scalar>
class WavSink : public Sink<Scalar> {
public:
  explicit WavSink(const std::string &filename)
    : Sink<Scalar>(), _file(filename.c_str(), std::ios_base::out | std::ios_base::binary), _frames(0) {
    if (!_file.is_open()) {
      ConfigError err;
      err << "Can not open wav file for output: " << filename;
      throw err;
    }
    switch (Config::typeId<Scalar>()) {
      case Config::Type_u8: case Config::Type_s8: _hdr.bitsPerSample = 8; _hdr.channels = 1; break;
      case Config::Type_cu8: case Config::Type_cs8: _hdr.bitsPerSample = 8; _hdr.channels = 2; break;
      case Config::Type_u16: case Config::Type_s16: _hdr.bitsPerSample = 16; _hdr.channels = 1; break;
      case Config::Type_cu16: case Config::Type_cs16: _hdr.bitsPerSample = 16; _hdr.channels = 2; break;
      default: {
        ConfigError err;
        err << "WAV format only allows (real) integer typed data.";
        throw err;
      }
    }
    const char blank[44] = {0};
    _file.write(blank, sizeof(blank));   // room for the header, filled in by close()
  }

  virtual ~WavSink() { if (_file.is_open()) this->close(); }

  virtual void config(const Config &src_cfg) {
    if (!src_cfg.hasType() || !src_cfg.hasSampleRate()) return;
    if (Config::typeId<Scalar>() != src_cfg.type()) {
      ConfigError err;
      err << "Can not configure WavSink: Invalid buffer type " << src_cfg.type() << ", expected " << Config::typeId<Scalar>();
      throw err;
    }
    _hdr.sampleRate = uint32_t(src_cfg.sampleRate());
  }

  /** Completes the header and closes the file. */
  void close() {
    if (!_file.is_open()) return;
    _hdr.riffSize = uint32_t(36u + 2u * _frames);
    _hdr.dataBytes = uint32_t(_hdr.channels) * _frames * (_hdr.bitsPerSample / 8);
    char raw[44];
    _hdr.emit(raw);
    _file.seekp(0);
    _file.write(raw, sizeof(raw));
    _file.close();
  }

  virtual void process(const Buffer<Scalar> &buffer, bool allow_overwrite) {
    (void)allow_overwrite;
    if (!_file.is_open()) return;
    _file.write(buffer.data(), buffer.size() * sizeof(Scalar));
    _frames += uint32_t(buffer.size());
  }

protected:
  std::fstream _file;
  wav::Header _hdr;
  uint32_t _frames;
};


/** Reads a PCM WAV file buffer by buffer; drive it with next() (e.g. as a Queue idle handler). */
class WavSource : public Source {
public:
  explicit WavSource(size_t buffer_size = 1024)
    : Source(), _buffer_size(buffer_size), _frame_count(0), _type(Config::Type_UNDEFINED), _sample_rate(0), _frames_left(0),
      _frame_bytes(0) {}
  WavSource(const std::string &filename, size_t buffer_size = 1024)
    : Source(), _buffer_size(buffer_size), _frame_count(0), _type(Config::Type_UNDEFINED), _sample_rate(0), _frames_left(0),
      _frame_bytes(0) {
    open(filename);
  }
  virtual ~WavSource() {
    _file.close();
    if (!_buffer.isEmpty()) _buffer.unref();
  }

  bool isOpen() const { return _file.is_open(); }
  bool isReal() const { return Config::Type_u8 == _type || Config::Type_s16 == _type; }
  inline size_t frameCount() const { return _frame_count; }   // (extension)

  /** Opens the file, parses the header and announces the stream; a missing file is not an error (isOpen()). */
  void open(const std::string &filename) {
    if (_file.is_open()) _file.close();
    _file.open(filename.c_str(), std::ios_base::in | std::ios_base::binary);
    if (!_file.is_open()) return;

    // RIFF <size> WAVE fmt_ <size>  — 20 bytes, then the PCM description
    char head[20];
    _file.read(head, sizeof(head));
    if (_file.gcount() != std::streamsize(sizeof(head)) || 0 != std::memcmp(head, "RIFF", 4) || 0 != std::memcmp(head + 8, "WAVE", 4)) {
      RuntimeError err;
      err << "File '" << filename << "' is not a WAV file.";
      throw err;
    }
    if (0 != std::memcmp(head + 12, "fmt ", 4)) {
      RuntimeError err;
      err << "'File 'fmt' header missing in file " << filename << "' @" << 12;
      throw err;
    }
    uint32_t fmt_size; std::memcpy(&fmt_size, head + 16, 4);
    struct { uint16_t format, channels; uint32_t rate, byte_rate; uint16_t align, bits; } __attribute__((packed)) f;
    _file.read(reinterpret_cast<char *>(&f), sizeof(f));
    if (1 != f.format) {
      RuntimeError err;
      err << "Unsupported WAV data format: " << f.format << " of file " << filename << ". Expected " << 1;
      throw err;
    }
    if (1 != f.channels && 2 != f.channels) {
      RuntimeError err;
      err << "Unsupported number of chanels: " << f.channels << " of file " << filename << ". Expected 1 or 2.";
      throw err;
    }
    if (16 != f.bits && 8 != f.bits) {
      RuntimeError err;
      err << "Unsupported sample format: " << f.bits << "b of file " << filename << ". Expected 16b or 8b.";
      throw err;
    }
    if (f.align != f.channels * (f.bits / 8)) {
      RuntimeError err;
      err << "Unsupported alignment: " << f.align << "byte of file " << filename << ". Expected " << (f.bits / 8) << "byte.";
      throw err;
    }

    // walk the chunk list behind `fmt ` until `data`
    uint32_t at = 12 + 8 + fmt_size, size = 0;
    for (;;) {
      char tag[8];
      _file.clear();
      _file.seekg(at);
      _file.read(tag, sizeof(tag));
      if (_file.gcount() != std::streamsize(sizeof(tag))) {
        RuntimeError err;
        err << "WAV file '" << filename << "' contains no 'data' chunk.";
        throw err;
      }
      std::memcpy(&size, tag + 4, 4);
      if (0 == std::memcmp(tag, "data", 4)) break;
      at += 8 + size;
    }

    _frame_bytes = size_t(f.channels) * (f.bits / 8);
    _frame_count = size / _frame_bytes;
    _type = wav::streamType(f.channels, f.bits);
    _sample_rate = f.rate;
    _frames_left = _frame_count;

    LogMessage msg(LOG_DEBUG);
    msg << "Configured WavSource:" << std::endl << " file: " << filename << std::endl << " type: " << _type << std::endl
        << " sample-rate: " << _sample_rate << std::endl << " frame-count: " << _frame_count << std::endl
        << " buffer-size: " << _buffer_size;
    Logger::get().log(msg);

    if (!_buffer.isEmpty()) _buffer.unref();
    _buffer = RawBuffer(_buffer_size * _frame_bytes);
    this->setConfig(Config(_type, _sample_rate, _buffer_size, 1));
  }

  void close() { _file.close(); _frames_left = 0; }

  /** Sends the next (up to) buffer_size frames; once nothing is left: closes the file and signals EOS. */
  void next() {
    if (0 == _frames_left) {
      _file.close();
      signalEOS();
      return;
    }
    const size_t n = std::min(_frames_left, _buffer_size);
    _file.read(_buffer.ptr(), n * _frame_bytes);
    _frames_left -= n;
    this->send(RawBuffer(_buffer, 0, n * _frame_bytes), true);
  }

protected:
  std::fstream _file;
  RawBuffer _buffer;
  size_t _buffer_size, _frame_count;
  Config::Type _type;
  double _sample_rate;
  size_t _frames_left, _frame_bytes;
};

}  // namespace sdr
#endif
