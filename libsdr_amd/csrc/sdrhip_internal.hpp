// sdrhip_internal.hpp — shared plumbing of libsdrhip.so (context, error handling, device buffers).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <utility>
#include <new>
#include <string>
#include <vector>

#include "sdrhip.h"

namespace sdrhip {

void set_error(const char *fmt, ...);

struct Failure {
  int code;
};

#define SDRHIP_FAIL(code_, ...)                 \
  do {                                          \
    ::sdrhip::set_error(__VA_ARGS__);           \
    throw ::sdrhip::Failure{code_};             \
  } while (0)

#define SDRHIP_CHECK_HIP(expr)                                                              \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      int c_ = (e_ == hipErrorOutOfMemory) ? SDRHIP_E_NOMEM                                 \
               : (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? SDRHIP_E_NODEVICE \
                                                                         : SDRHIP_E_HIP;    \
      SDRHIP_FAIL(c_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    }                                                                                       \
  } while (0)

#define SDRHIP_REQUIRE(cond, code_, ...) \
  do {                                   \
    if (!(cond)) SDRHIP_FAIL(code_, __VA_ARGS__); \
  } while (0)

// wraps every extern "C" entry point: C++ failures become error codes, nothing escapes
template <class F>
static inline int guarded(F &&f) {
  try {
    f();
    return SDRHIP_OK;
  } catch (const Failure &e) {
    return e.code;
  } catch (const std::bad_alloc &) {
    set_error("host allocation failed");
    return SDRHIP_E_NOMEM;
  } catch (...) {
    set_error("unexpected C++ exception");
    return SDRHIP_E_INVALID;
  }
}

// Transform size of an overlap-save plan for n_taps complex<float> taps (the tuned kernels' own ranking, measured on the
// round-6 kernels at 16 ... 1024 channels x 65536 samples, tools/probes/fft_rank.py = profiles/r19_fft_rank.txt): the 2048-point
// plan up to 768 taps; beyond, the pipelined 16384-point kernel wherever the call has blocks enough to walk (4 per CU: it then
// beats the 4096-point plan by 3 % at 1024 taps and by 35 % at 2048) and the 4096-point plan up to 2048 taps where not.
inline int ols_fft_size(int n_taps, size_t channels, size_t max_in, int cus) {
  if (n_taps <= 768) return 2048;
  const size_t units16 = channels * ((max_in + (size_t)(16384 - n_taps)) / (size_t)(16384 - n_taps + 1));
  if (n_taps <= 12289 && units16 >= (size_t)4 * (size_t)cus) return 16384;
  return n_taps <= 2048 ? 4096 : 16384;
}

}  // namespace sdrhip

struct sdrhip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipDeviceProp_t prop;
  // plans the one-shot entry points keep between calls (sdrhip_fft_c2c / sdrhip_fft_exec: (dtype, n) -> plan), type-erased;
  // released in sdrhip_ctx_destroy before the stream goes
  std::map<long long, std::shared_ptr<void> > cache;
  void use() const;  // hipSetDevice
};

struct sdrhip_timer {
  sdrhip_ctx *ctx;
  hipEvent_t a, b;
};

namespace sdrhip {

// RAII device allocation bound to a context
template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() {}
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void alloc(size_t count) {
    release();
    n = count;
    if (count) SDRHIP_CHECK_HIP(hipMalloc((void **)&p, count * sizeof(T)));
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void zero(hipStream_t s) {
    if (p) SDRHIP_CHECK_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s));
  }
  void upload(const T *h, size_t count, hipStream_t s) {
    SDRHIP_CHECK_HIP(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s));
    SDRHIP_CHECK_HIP(hipStreamSynchronize(s));
  }
};

// staging for the host-pointer entry points: 2-D copies between a strided host layout and a
// packed device layout
void copy_h2d_rows(const sdrhip_ctx *ctx, void *dst_dev, size_t dst_pitch_b, const void *src_host,
                   size_t src_pitch_b, size_t row_bytes, size_t rows);
void copy_d2h_rows(const sdrhip_ctx *ctx, void *dst_host, size_t dst_pitch_b, const void *src_dev,
                   size_t src_pitch_b, size_t row_bytes, size_t rows);

static inline size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }

// A kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) belongs to the (kernel, device) pair, and one
// template instance serves every plan of its type — cached and persistent plans included. It is therefore raised ONCE per
// kernel and device to all the hardware has (160 KB less the kernel's static LDS), never to one plan's own need: a later
// plan that needs less would otherwise lower the cap under a live plan that launches with more.
template <class K>
inline void allow_lds_max(K kernel, size_t need_bytes) {
  if (need_bytes <= 64 * 1024) return;
  constexpr size_t LDS_PER_CU = 160 * 1024;
  SDRHIP_REQUIRE(need_bytes <= LDS_PER_CU, SDRHIP_E_UNSUPPORTED, "%zu bytes of LDS per workgroup exceed the CU's 160 KB", need_bytes);
  static std::mutex mu;
  static std::set<std::pair<const void *, int>> done;
  int dev = 0;
  SDRHIP_CHECK_HIP(hipGetDevice(&dev));
  const void *fn = reinterpret_cast<const void *>(kernel);
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({fn, dev})) return;
  hipFuncAttributes fa{};
  SDRHIP_CHECK_HIP(hipFuncGetAttributes(&fa, fn));
  SDRHIP_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_PER_CU - fa.sharedSizeBytes)));
  done.insert({fn, dev});
}

// roctx range around one process() call (SURVEY §5 tracing): with SDRHIP_ROCTX=1 in the environment every *_process /
// *_process_dev entry point is bracketed by roctxRangePush / roctxRangePop, so that a `rocprofv3 --marker-trace` run
// shows the node calls beside their kernels. libroctx64 is opened with dlopen the first time a range is asked for;
// without the variable (or the library) a range costs one branch.
struct Range {
  explicit Range(const char *name);
  ~Range();
  bool on;
};

// *_process_dev contract (sdrhip.h): the kernels are tile-parallel (a workgroup reads neighbouring tiles' inputs and
// the last one rolls the history from the input), so an output range that overlaps the input range would race
// silently. rows x row_elems elements of elem bytes at a row stride of `stride` elements.
static inline void require_disjoint(const void *in, size_t in_stride, size_t in_row, size_t in_elem, const void *out,
                                    size_t out_stride, size_t out_row, size_t out_elem, size_t rows, size_t out_rows = 0) {
  if (!out_rows) out_rows = rows;   // (a filter bank writes bands x channels rows from channels rows of input)
  if (!rows || !in_row || !out_row) return;
  const uintptr_t a0 = (uintptr_t)in, a1 = a0 + ((rows - 1) * in_stride + in_row) * in_elem;
  const uintptr_t b0 = (uintptr_t)out, b1 = b0 + ((out_rows - 1) * out_stride + out_row) * out_elem;
  SDRHIP_REQUIRE(a1 <= b0 || b1 <= a0, SDRHIP_E_INVALID,
                 "process_dev: the output range overlaps the input range (in place is only supported by the host-pointer "
                 "*_process entry points, which stage through separate device buffers)");
}

// fir.hip: turn on the frequency shift fused into the cf32 FIR's staging (used by the float baseband, fbb_f32.hip)
void fir_set_shift(sdrhip_fir *h, double fc, double fs);
void fir_load_taps(sdrhip_fir *h, const double *alpha);
int fir_create_impl(sdrhip_ctx *ctx, int kind, const double *alpha, int order, int decim, int channels, size_t max_in, int epilogue,
                    bool allow_fft, sdrhip_fir **out);

#ifdef __HIPCC__
// XCD-aware unit order for grids of (units-per-channel, channels) whose neighbouring units of a channel re-read each
// other's input (FIR history, overlap-save). Workgroups are dealt round-robin over the 8 XCDs by linear id and each XCD
// has its own L2: in launch order (unit fastest) the two readers of an overlap sit on different XCDs and it is fetched
// from HBM twice. With this map XCD x walks channels x, x + 8, ... unit by unit, so neighbours run on one XCD at the same
// time and the second reader hits in L2. Speed only: any placement is correct (MI355X_MICROARCH.md, workgroup dispatch).
__device__ __forceinline__ void xcd_unit_order(int &unit, int &chan) {
  unit = (int)blockIdx.x; chan = (int)blockIdx.y;
  if ((gridDim.y & 7u) == 0) {
    const unsigned long long lin = blockIdx.x + (unsigned long long)gridDim.x * blockIdx.y, k = lin >> 3;
    chan = (int)((unsigned)(lin & 7u) + 8u * (unsigned)(k / gridDim.x)); unit = (int)(k % gridDim.x);
  }
}
#endif

}  // namespace sdrhip
