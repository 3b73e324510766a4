// buffer.hh — reference-counted sample buffers of the API-compatible sdr:: core (own code).
//
// Contract kept from the reference (src/buffer.hh:32-251, src/buffer.cc): a RawBuffer is a VIEW
// (pointer, byte offset, byte length) on shared storage plus a pointer to a shared counter.
// Copying or assigning a buffer does NOT take a reference; ref()/unref() are explicit; a buffer
// is "unused" when its owner holds the only reference (count == 1); storage is released when
// the count reaches zero; wrapping foreign memory (Buffer(T*, n)) has no counter at all.
// Difference: the counter is atomic, because Queue::send() refs on the producer thread while
// the worker thread unrefs (SURVEY §5 lists the reference's plain int as a known race).
#ifndef SDR_CORE_BUFFER_HH
#define SDR_CORE_BUFFER_HH

#include <atomic>
#include <cmath>
#include <complex>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <inttypes.h>
#include <list>
#include <map>
#include <ostream>
#include <vector>

#include "exception.hh"

namespace sdr {

class RawBuffer;

/** Gets notified when only the owner's reference to one of its buffers is left. */
class BufferOwner {
public:
  virtual ~BufferOwner() {}
  virtual void bufferUnused(const RawBuffer &buffer) = 0;
};

class RawBuffer {
public:
  RawBuffer() : _ptr(0), _storage_size(0), _b_offset(0), _b_length(0), _refcount(0), _owner(0) {}

  /** View on memory this library does not manage (never freed, never counted). */
  RawBuffer(char *data, size_t offset, size_t len)
    : _ptr(data), _storage_size(offset + len), _b_offset(offset), _b_length(len), _refcount(0), _owner(0) {}

  /** Allocates N bytes; the creator holds the first reference. */
  explicit RawBuffer(size_t N, BufferOwner *owner = 0)
    : _ptr(0), _storage_size(0), _b_offset(0), _b_length(0), _refcount(0), _owner(owner) {
    // 64-byte alignment: the GPU nodes hand these pointers to hipMemcpy2DAsync
    void *p = 0;
    if (N == 0 || posix_memalign(&p, 64, N) == 0) {
      _ptr = (char *)p;
      _refcount = new (std::nothrow) std::atomic<int>(1);
      if (!_refcount) { free(_ptr); _ptr = 0; return; }
      _storage_size = _b_length = N;
    }
  }

  RawBuffer(const RawBuffer &o)
    : _ptr(o._ptr), _storage_size(o._storage_size), _b_offset(o._b_offset), _b_length(o._b_length),
      _refcount(o._refcount), _owner(o._owner) {}

  /** Sub-view (offset relative to the other view). */
  RawBuffer(const RawBuffer &o, size_t offset, size_t len)
    : _ptr(o._ptr), _storage_size(o._storage_size), _b_offset(o._b_offset + offset), _b_length(len),
      _refcount(o._refcount), _owner(o._owner) {}

  virtual ~RawBuffer() {}

  const RawBuffer &operator=(const RawBuffer &o) {
    _ptr = o._ptr; _storage_size = o._storage_size; _b_offset = o._b_offset; _b_length = o._b_length;
    _refcount = o._refcount; _owner = o._owner;
    return *this;
  }

  inline char *ptr() const { return _ptr; }
  inline char *data() const { return _ptr + _b_offset; }
  inline size_t bytesOffset() const { return _b_offset; }
  inline size_t bytesLen() const { return _b_length; }
  inline size_t storageSize() const { return _storage_size; }
  inline bool isEmpty() const { return 0 == _ptr; }

  void ref() const { if (_refcount) _refcount->fetch_add(1, std::memory_order_relaxed); }

  void unref() {
    if (!_ptr || !_refcount) return;
    const int left = _refcount->fetch_sub(1, std::memory_order_acq_rel) - 1;
    if (left == 1 && _owner) _owner->bufferUnused(*this);
    if (left == 0) {
      free(_ptr);
      delete _refcount;
      _ptr = 0; _refcount = 0;
    }
  }

  inline int refCount() const { return _refcount ? _refcount->load(std::memory_order_acquire) : 0; }
  inline bool isUnused() const { return !_refcount || _refcount->load(std::memory_order_acquire) == 1; }

protected:
  char *_ptr;
  size_t _storage_size, _b_offset, _b_length;
  std::atomic<int> *_refcount;
  BufferOwner *_owner;
};

template <class T>
class Buffer : public RawBuffer {
public:
  Buffer() : RawBuffer(), _size(0) {}
  Buffer(T *data, size_t size) : RawBuffer((char *)data, 0, sizeof(T) * size), _size(size) {}
  explicit Buffer(size_t N, BufferOwner *owner = 0) : RawBuffer(N * sizeof(T), owner), _size(N) {}
  Buffer(const Buffer<T> &o) : RawBuffer(o), _size(o._size) {}
  /** Re-types an untyped view; size = bytes / sizeof(T). */
  explicit Buffer(const RawBuffer &o) : RawBuffer(o), _size(o.bytesLen() / sizeof(T)) {}
  virtual ~Buffer() { _size = 0; }

  const Buffer<T> &operator=(const Buffer<T> o) {
    RawBuffer::operator=(o);
    _size = o._size;
    return *this;
  }
  inline bool operator<(const Buffer<T> &o) const { return this->_ptr < o._ptr; }

  inline size_t size() const { return _size; }

  inline T &operator[](int idx) const {
#ifdef SDR_DEBUG
    if (idx < 0 || size_t(idx) >= _size) {
      RuntimeError err;
      err << "Index " << idx << " out of bounds [0," << _size << ")";
      throw err;
    }
#endif
    return reinterpret_cast<T *>(_ptr + _b_offset)[idx];
  }

  inline double norm2() const {
    double s = 0;
    for (size_t i = 0; i < _size; i++) s += std::real(std::conj((*this)[i]) * (*this)[i]);
    return std::sqrt(s);
  }
  inline double norm() const {
    double s = 0;
    for (size_t i = 0; i < _size; i++) s += std::abs((*this)[i]);
    return s;
  }
  inline double norm(double p) const {
    double s = 0;
    for (size_t i = 0; i < _size; i++) s += std::pow(std::abs((*this)[i]), p);
    return std::pow(s, 1. / p);
  }
  inline Buffer<T> &operator*=(const T &a) { for (size_t i = 0; i < _size; i++) (*this)[i] *= a; return *this; }
  inline Buffer<T> &operator/=(const T &a) { for (size_t i = 0; i < _size; i++) (*this)[i] /= a; return *this; }

  template <class oT> Buffer<oT> as() const { return Buffer<oT>((const RawBuffer &)(*this)); }

  inline Buffer<T> sub(size_t offset, size_t len) const {
    if (offset + len > _size) return Buffer<T>();
    return Buffer<T>(RawBuffer(*this, offset * sizeof(T), len * sizeof(T)));
  }
  inline Buffer<T> head(size_t n) const { return n > _size ? Buffer<T>() : sub(0, n); }
  inline Buffer<T> tail(size_t n) const { return n > _size ? Buffer<T>() : sub(_size - n, n); }

protected:
  size_t _size;
};

template <class Scalar>
std::ostream &operator<<(std::ostream &s, const Buffer<Scalar> &b) {
  s << "[";
  const size_t n = b.size(), show = n > 10 ? 5 : n;
  for (size_t i = 0; i < show; i++) s << (i ? ", " : "") << +b[i];
  if (n > 10) { s << ", ..."; for (size_t i = n - 3; i < n; i++) s << ", " << +b[i]; }
  s << "]";
  return s;
}

/** A pool of equally sized buffers; a buffer returns to the free list when its consumers have all
 * unref()'d it. (The reference's BufferSet never fills its free list on resize() and pops from an
 * empty list — SURVEY fact 7; this one allocates what it promises.) */
template <class Scalar>
class BufferSet : public BufferOwner {
public:
  BufferSet(size_t N, size_t size) : _bufferSize(size) { grow(N); }
  virtual ~BufferSet() {
    for (auto &kv : _buffers) { Buffer<Scalar> b = kv.second; b.unref(); }
  }
  bool hasBuffer() const { return !_free.empty(); }
  Buffer<Scalar> getBuffer() {
    if (_free.empty()) grow(1);
    void *id = _free.front();
    _free.pop_front();
    return _buffers[id];   // not ref'd: it is recycled once a consumer's ref()/unref() pair (e.g. the
                           // Queue's) brings the count back to 1; un-ref'd direct use just grows the pool
  }
  virtual void bufferUnused(const RawBuffer &buffer) { _free.push_back(buffer.ptr()); }
  void resize(size_t numBuffers) { if (numBuffers > _buffers.size()) grow(numBuffers - _buffers.size()); }

protected:
  void grow(size_t n) {
    for (size_t i = 0; i < n; i++) {
      Buffer<Scalar> b(_bufferSize, this);
      _buffers[b.ptr()] = b;
      _free.push_back(b.ptr());
    }
  }
  size_t _bufferSize;
  std::map<void *, Buffer<Scalar> > _buffers;
  std::list<void *> _free;
};

}  // namespace sdr
#endif
