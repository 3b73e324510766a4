// exception.hh — error types of the API-compatible sdr:: core (this repository's own code).
// Mirrors the reference's surface (src/exception.hh:10-44): an exception that is also a string
// stream, so call sites read `ConfigError err; err << "..."; throw err;`.
#ifndef SDR_CORE_EXCEPTION_HH
#define SDR_CORE_EXCEPTION_HH

#include <exception>
#include <sstream>
#include <string>

namespace sdr {

class SDRError : public std::exception, public std::stringstream {
public:
  SDRError() {}
  SDRError(const SDRError &o) : std::basic_ios<char>(), std::exception(), std::stringstream() { this->str(o.str()); }
  virtual ~SDRError() throw() {}
  virtual const char *what() const throw() {
    _text = this->str();
    return _text.c_str();
  }

private:
  mutable std::string _text;   // keeps what() valid after the call
};

class ConfigError : public SDRError {
public:
  ConfigError() {}
  ConfigError(const ConfigError &o) : std::basic_ios<char>(), SDRError(o) {}
  virtual ~ConfigError() throw() {}
};

class RuntimeError : public SDRError {
public:
  RuntimeError() {}
  RuntimeError(const RuntimeError &o) : std::basic_ios<char>(), SDRError(o) {}
  virtual ~RuntimeError() throw() {}
};

}  // namespace sdr
#endif
