// logger.hh — singleton logger of the API-compatible sdr:: core (own code; surface of the
// reference's src/logger.hh: LogLevel, LogMessage, LogHandler, StreamLogHandler, Logger::get()).
#ifndef SDR_CORE_LOGGER_HH
#define SDR_CORE_LOGGER_HH

#include <list>
#include <memory>
#include <mutex>
#include <ostream>
#include <sstream>
#include <string>

namespace sdr {

typedef enum { LOG_DEBUG = 0, LOG_INFO, LOG_WARNING, LOG_ERROR } LogLevel;

class LogMessage : public std::stringstream {
public:
  LogMessage(LogLevel level, const std::string &msg = "") : _level(level) { (*this) << msg; }
  LogMessage(const LogMessage &o) : std::basic_ios<char>(), std::stringstream(), _level(o._level) { (*this) << o.str(); }
  virtual ~LogMessage() {}
  LogLevel level() const { return _level; }
  std::string message() const { return this->str(); }

protected:
  LogLevel _level;
};

class LogHandler {
protected:
  LogHandler() {}

public:
  virtual ~LogHandler() {}
  virtual void handle(const LogMessage &msg) = 0;
};

class StreamLogHandler : public LogHandler {
public:
  StreamLogHandler(std::ostream &stream, LogLevel level) : _stream(stream), _level(level) {}
  virtual ~StreamLogHandler() {}
  virtual void handle(const LogMessage &msg) {
    if (msg.level() < _level) return;
    static const char *tag[] = {"DEBUG: ", "INFO: ", "WARN: ", "ERROR: "};
    _stream << tag[msg.level()] << msg.message() << std::endl;
  }

protected:
  std::ostream &_stream;
  LogLevel _level;
};

class Logger {
protected:
  Logger() {}

public:
  virtual ~Logger() {}
  static Logger &get() {
    static Logger instance;
    return instance;
  }
  void log(const LogMessage &message) {
    std::lock_guard<std::mutex> g(_lock);
    for (auto &h : _handler) h->handle(message);
  }
  /** Takes ownership of the handler. */
  void addHandler(LogHandler *handler) {
    std::lock_guard<std::mutex> g(_lock);
    _handler.emplace_back(handler);
  }

protected:
  std::mutex _lock;
  std::list< std::unique_ptr<LogHandler> > _handler;
};

}  // namespace sdr
#endif
