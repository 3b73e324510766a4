// queue.hh — the central message queue of the API-compatible sdr:: core (own code).
//
// Surface and behaviour of the reference's Queue (src/queue.hh:53-216, src/queue.cc): a singleton
// that owns ONE worker thread; Source::send() on a queued edge refs the buffer and enqueues
// (buffer, sink, allow_overwrite); the worker pops, calls sink->handleBuffer() and unrefs; when
// the queue runs dry the idle delegates are signalled (sources produce their next buffer there);
// stop() lets the worker drain what is queued and then fire the stop delegates; an exception on
// the worker thread is logged and stops the queue (src/queue.cc:151-169).
// Implementation differences: std::thread / std::mutex / std::condition_variable instead of raw
// pthreads, an atomic running flag (the reference's plain bool is a data race) and
// delegates owned by unique_ptr.
#ifndef SDR_CORE_QUEUE_HH
#define SDR_CORE_QUEUE_HH

#include <atomic>
#include <condition_variable>
#include <list>
#include <memory>
#include <mutex>
#include <thread>

#include "buffer.hh"
#include "logger.hh"

namespace sdr {

class SinkBase;

class DelegateInterface {
public:
  virtual ~DelegateInterface() {}
  virtual void operator()() = 0;
  virtual void *instance() = 0;
};

template <class T>
class Delegate : public DelegateInterface {
public:
  Delegate(T *instance, void (T::*func)(void)) : _instance(instance), _function(func) {}
  virtual ~Delegate() {}
  virtual void operator()() { (_instance->*_function)(); }
  virtual void *instance() { return _instance; }

protected:
  T *_instance;
  void (T::*_function)(void);
};

namespace detail {
// defined in node.hh (SinkBase is incomplete here)
void deliver(SinkBase *sink, const RawBuffer &buffer, bool allow_overwrite);
}

class Queue {
public:
  /** One queued delivery: what goes where, and whether the receiver may write into it. */
  class Message {
  public:
    Message(const RawBuffer &what, SinkBase *to, bool writable) : _what(what), _to(to), _writable(writable) {}
    RawBuffer &buffer() { return _what; }
    const RawBuffer &buffer() const { return _what; }
    SinkBase *sink() const { return _to; }
    bool allowOverwrite() const { return _writable; }

  private:
    RawBuffer _what;
    SinkBase *_to;
    bool _writable;
  };

protected:
  Queue() : _running(false) {}

public:
  virtual ~Queue() {
    if (_thread.joinable()) { stop(); _thread.join(); }
  }

  static Queue &get() {
    static Queue instance;
    return instance;
  }

  void send(const RawBuffer &buffer, SinkBase *sink, bool allow_overwrite = false) {
    std::lock_guard<std::mutex> g(_lock);
    buffer.ref();
    _queue.push_back(Message(buffer, sink, allow_overwrite));
    _cond.notify_one();
  }

  void start() {
    if (_running.load()) return;
    if (_thread.joinable()) _thread.join();
    _running.store(true);   // set before the thread exists so that isRunning() holds on return
    _thread = std::thread(&Queue::threadMain, this);
  }

  void stop() {
    { std::lock_guard<std::mutex> g(_lock); _running.store(false); }
    _cond.notify_all();
  }

  void wait() {
    if (_thread.joinable()) _thread.join();
    std::lock_guard<std::mutex> g(_lock);
    for (auto &m : _queue) m.buffer().unref();
    _queue.clear();
  }

  bool isStopped() const { return !_running.load(); }
  bool isRunning() const { return _running.load(); }

  template <class T> void addIdle(T *instance, void (T::*function)(void)) { _idle.emplace_back(new Delegate<T>(instance, function)); }
  template <class T> void remIdle(T *instance) { removeFrom(_idle, (void *)instance); }
  template <class T> void addStart(T *instance, void (T::*function)(void)) { _onStart.emplace_back(new Delegate<T>(instance, function)); }
  template <class T> void remStart(T *instance) { removeFrom(_onStart, (void *)instance); }
  template <class T> void addStop(T *instance, void (T::*function)(void)) { _onStop.emplace_back(new Delegate<T>(instance, function)); }
  template <class T> void remStop(T *instance) { removeFrom(_onStop, (void *)instance); }

protected:
  typedef std::list< std::unique_ptr<DelegateInterface> > Delegates;

  static void removeFrom(Delegates &l, void *instance) {
    for (auto it = l.begin(); it != l.end();) {
      if ((*it)->instance() == instance) it = l.erase(it);
      else ++it;
    }
  }
  static void fire(Delegates &l) { for (auto &d : l) (*d)(); }

  bool pop(Message &out) {
    std::lock_guard<std::mutex> g(_lock);
    if (_queue.empty()) return false;
    out = _queue.front();
    _queue.pop_front();
    return true;
  }

  void loop() {
    Logger::get().log(LogMessage(LOG_DEBUG, "Queue started."));
    fire(_onStart);
    Message msg(RawBuffer(), 0, false);
    for (;;) {
      while (pop(msg)) {
        detail::deliver(msg.sink(), msg.buffer(), msg.allowOverwrite());
        msg.buffer().unref();
      }
      if (!_running.load()) {
        std::lock_guard<std::mutex> g(_lock);
        if (_queue.empty()) break;
        continue;
      }
      fire(_idle);   // sources push their next buffer from here
      std::unique_lock<std::mutex> lk(_lock);
      _cond.wait(lk, [this] { return !_queue.empty() || !_running.load(); });
    }
    fire(_onStop);
    LogMessage done(LOG_DEBUG, "Queue stopped.");
    Logger::get().log(done);
  }

  void threadMain() {
    try {
      loop();
    } catch (std::exception &err) {
      LogMessage msg(LOG_ERROR);
      msg << "Caught exception in thread: " << err.what() << " -> Stop thread.";
      Logger::get().log(msg);
    } catch (...) {
      Logger::get().log(LogMessage(LOG_ERROR, "Caught (unknown) exception in thread -> Stop thread."));
    }
    _running.store(false);
  }

  std::atomic<bool> _running;
  std::thread _thread;
  std::mutex _lock;
  std::condition_variable _cond;
  std::list<Message> _queue;
  Delegates _idle, _onStart, _onStop;
};

}  // namespace sdr
#endif
