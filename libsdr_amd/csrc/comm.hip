// comm.hip — one process, several MI355X: per-rank contexts, broadcast of the read-only design and gather of the
// demodulated output over RCCL (SURVEY §8b "sdrhip_comm_*", §8e; BASELINE config 5 from the C++ side), plus the
// pinned-host / asynchronous copy helpers the multi-device ChannelBank stages through.
//
// Channels are independent units: the data path has NO collective (each device filters its contiguous block of
// channels). The two exchanges that exist are a broadcast of a few KB at config time and one gather of the
// demodulated output per step — point-to-point xGMI traffic, no all-reduce.
//
// RCCL is opened with dlopen at the first sdrhip_comm_create that needs it, so libsdrhip.so itself carries no link
// dependency on it (tests/test_abi.py checks `ldd`): a process that only drives one GPU never loads it.
// RCCL refuses two ranks on one device ("Duplicate GPU detected"); ranks that share a device — the single-GPU
// boxes the tests run on — are served by device-to-device copies on that device's streams instead. That is a
// transport choice between DEVICE paths; nothing here touches the CPU.
#include "sdrhip_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only; every entry point is resolved through dlsym

#include <cstdlib>
#include <mutex>
#include <set>

using namespace sdrhip;

namespace {

struct RcclApi {
  void *lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

RcclApi &rccl() {
  static RcclApi api;
  static std::mutex mtx;   // (a failed load throws and may be retried, so no once-flag: the table is filled under the lock)
  std::lock_guard<std::mutex> lock(mtx);
  if (api.lib) return api;   // (set last: only once every symbol below has been found)
  RcclApi a;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names) { a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (a.lib) break; }
  SDRHIP_REQUIRE(a.lib, SDRHIP_E_UNSUPPORTED, "RCCL (librccl.so.1) can not be loaded: %s", dlerror());
#define SDRHIP_SYM(field, sym)                                                                  \
  a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.lib, #sym));                             \
  if (!a.field) { dlclose(a.lib); SDRHIP_FAIL(SDRHIP_E_UNSUPPORTED, "librccl lacks %s", #sym); }
  SDRHIP_SYM(CommInitAll, ncclCommInitAll); SDRHIP_SYM(CommDestroy, ncclCommDestroy); SDRHIP_SYM(Broadcast, ncclBroadcast);
  SDRHIP_SYM(Send, ncclSend); SDRHIP_SYM(Recv, ncclRecv); SDRHIP_SYM(GroupStart, ncclGroupStart);
  SDRHIP_SYM(GroupEnd, ncclGroupEnd); SDRHIP_SYM(GetErrorString, ncclGetErrorString);
#undef SDRHIP_SYM
  api = a;
  return api;
}

#define SDRHIP_CHECK_NCCL(expr)                                                                               \
  do {                                                                                                        \
    ncclResult_t r_ = (expr);                                                                                 \
    if (r_ != ncclSuccess) SDRHIP_FAIL(SDRHIP_E_HIP, "%s failed: %s", #expr, rccl().GetErrorString(r_));      \
  } while (0)

}  // namespace

struct sdrhip_comm {
  std::vector<int> devices;
  std::vector<sdrhip_ctx *> ctx;        // one per rank (its own stream on its device)
  std::vector<ncclComm_t> nccl;         // empty when the ranks share devices
  std::vector<hipEvent_t> ev;           // same-device transport: one event per rank (its stream has produced / consumed a buffer)
  std::vector<hipEvent_t> ev2;          // ... and one more per rank for the opposite direction of the same call
  bool use_rccl = false;
  // gather_begin / gather_wait: the transfers run on streams the comm owns (one per rank, made at the first begin), so that
  // a rank's next kernels do not queue up behind them
  static constexpr int kSlots = 4;
  std::vector<hipStream_t> side;                 // per rank
  std::vector<hipEvent_t> ready;                 // per rank: its stream has produced the send buffer
  std::vector<hipEvent_t> done[kSlots];          // per slot and rank: the transfer issued under the slot has read / written that rank's memory
  bool slot_used[kSlots] = {false, false, false, false};
  void make_side() {
    if (!side.empty()) return;
    // built aside and swapped in only when every stream and event exists: a creation that throws half-way must not leave
    // `side` non-empty with null handles (the next gather_begin would skip this and issue on the null stream)
    const size_t n = ctx.size();
    std::vector<hipStream_t> s(n, nullptr);
    std::vector<hipEvent_t> rd(n, nullptr), dn[kSlots];
    for (int k = 0; k < kSlots; k++) dn[k].assign(n, nullptr);
    try {
      for (size_t r = 0; r < n; r++) {
        ctx[r]->use();
        SDRHIP_CHECK_HIP(hipStreamCreateWithFlags(&s[r], hipStreamNonBlocking));
        SDRHIP_CHECK_HIP(hipEventCreateWithFlags(&rd[r], hipEventDisableTiming));
        for (int k = 0; k < kSlots; k++) SDRHIP_CHECK_HIP(hipEventCreateWithFlags(&dn[k][r], hipEventDisableTiming));
      }
    } catch (...) {
      for (size_t r = 0; r < n; r++) {
        if (s[r]) (void)hipStreamDestroy(s[r]);
        if (rd[r]) (void)hipEventDestroy(rd[r]);
        for (int k = 0; k < kSlots; k++) if (dn[k][r]) (void)hipEventDestroy(dn[k][r]);
      }
      throw;
    }
    side.swap(s); ready.swap(rd);
    for (int k = 0; k < kSlots; k++) done[k].swap(dn[k]);
  }
};

extern "C" {

int sdrhip_comm_create(const int *devices, int nranks, sdrhip_comm **out) {
  return guarded([&] {
    SDRHIP_REQUIRE(devices && out, SDRHIP_E_INVALID, "NULL argument");
    *out = nullptr;
    SDRHIP_REQUIRE(nranks >= 1 && nranks <= 64, SDRHIP_E_INVALID, "nranks %d outside [1,64]", nranks);
    sdrhip_comm *c = new sdrhip_comm;
    try {
      c->devices.assign(devices, devices + nranks);
      for (int r = 0; r < nranks; r++) {
        sdrhip_ctx *x = nullptr;
        const int rc = sdrhip_ctx_create(devices[r], nullptr, &x);
        if (rc != SDRHIP_OK) throw Failure{rc};
        c->ctx.push_back(x);
      }
      const std::set<int> distinct(c->devices.begin(), c->devices.end());
      // one rank needs no transport at all (its gather is a copy on its own stream); SDRHIP_COMM_FORCE_RCCL=1 makes a
      // single rank go through RCCL anyway, which is how the single-GPU test boxes exercise the RCCL code path
      const char *force = getenv("SDRHIP_COMM_FORCE_RCCL");
      c->use_rccl = (int)distinct.size() == nranks && (nranks > 1 || (force && force[0] == '1'));
      if (c->use_rccl) {
        c->nccl.resize(nranks);
        SDRHIP_CHECK_NCCL(rccl().CommInitAll(c->nccl.data(), nranks, c->devices.data()));
      } else {
        SDRHIP_REQUIRE(distinct.size() == 1, SDRHIP_E_UNSUPPORTED,
                       "ranks must sit on distinct devices (RCCL) or all on one device (same-device copies)");
        c->ev.resize(nranks); c->ev2.resize(nranks);
        for (int r = 0; r < nranks; r++) {
          c->ctx[r]->use();
          SDRHIP_CHECK_HIP(hipEventCreateWithFlags(&c->ev[r], hipEventDisableTiming));
          SDRHIP_CHECK_HIP(hipEventCreateWithFlags(&c->ev2[r], hipEventDisableTiming));
        }
      }
    } catch (...) {
      for (sdrhip_ctx *x : c->ctx) (void)sdrhip_ctx_destroy(x);
      delete c;
      throw;
    }
    *out = c;
  });
}

int sdrhip_comm_size(sdrhip_comm *c, int *nranks) {
  return guarded([&] {
    SDRHIP_REQUIRE(c && nranks, SDRHIP_E_INVALID, "NULL argument");
    *nranks = (int)c->ctx.size();
  });
}

int sdrhip_comm_ctx(sdrhip_comm *c, int rank, sdrhip_ctx **ctx) {
  return guarded([&] {
    SDRHIP_REQUIRE(c && ctx, SDRHIP_E_INVALID, "NULL argument");
    SDRHIP_REQUIRE(rank >= 0 && rank < (int)c->ctx.size(), SDRHIP_E_INVALID, "rank %d outside [0,%zu)", rank, c->ctx.size());
    *ctx = c->ctx[rank];
  });
}

int sdrhip_comm_transport(sdrhip_comm *c, const char **name) {
  return guarded([&] {
    SDRHIP_REQUIRE(c && name, SDRHIP_E_INVALID, "NULL argument");
    *name = c->use_rccl ? "rccl" : "same-device copies";
  });
}

int sdrhip_comm_broadcast(sdrhip_comm *c, void *const *bufs_dev, size_t bytes, int root) {
  return guarded([&] {
    SDRHIP_REQUIRE(c && bufs_dev, SDRHIP_E_INVALID, "NULL argument");
    const int n = (int)c->ctx.size();
    SDRHIP_REQUIRE(root >= 0 && root < n, SDRHIP_E_INVALID, "root %d outside [0,%d)", root, n);
    if (!bytes) return;
    for (int r = 0; r < n; r++) SDRHIP_REQUIRE(bufs_dev[r], SDRHIP_E_INVALID, "rank %d: NULL buffer", r);
    if (c->use_rccl) {
      SDRHIP_CHECK_NCCL(rccl().GroupStart());
      for (int r = 0; r < n; r++)
        SDRHIP_CHECK_NCCL(rccl().Broadcast(bufs_dev[root], bufs_dev[r], bytes, ncclUint8, root, c->nccl[r], c->ctx[r]->stream));
      SDRHIP_CHECK_NCCL(rccl().GroupEnd());
    } else {   // every rank copies from the root's buffer once the root's stream has produced it ...
      c->ctx[root]->use();
      SDRHIP_CHECK_HIP(hipEventRecord(c->ev[root], c->ctx[root]->stream));
      for (int r = 0; r < n; r++) {
        if (r == root || bufs_dev[r] == bufs_dev[root]) continue;
        SDRHIP_CHECK_HIP(hipStreamWaitEvent(c->ctx[r]->stream, c->ev[root], 0));
        SDRHIP_CHECK_HIP(hipMemcpyAsync(bufs_dev[r], bufs_dev[root], bytes, hipMemcpyDeviceToDevice, c->ctx[r]->stream));
        // ... and the root's stream may not overwrite its buffer before every copy has read it (the call is
        // asynchronous like the RCCL transport's: whatever the caller enqueues next on any rank's stream is ordered)
        SDRHIP_CHECK_HIP(hipEventRecord(c->ev2[r], c->ctx[r]->stream));
        SDRHIP_CHECK_HIP(hipStreamWaitEvent(c->ctx[root]->stream, c->ev2[r], 0));
      }
    }
  });
}

int sdrhip_comm_gather(sdrhip_comm *c, const void *const *send_dev, const size_t *bytes, void *recv_dev, int root) {
  return guarded([&] {
    SDRHIP_REQUIRE(c && send_dev && bytes && recv_dev, SDRHIP_E_INVALID, "NULL argument");
    const int n = (int)c->ctx.size();
    SDRHIP_REQUIRE(root >= 0 && root < n, SDRHIP_E_INVALID, "root %d outside [0,%d)", root, n);
    std::vector<size_t> off(n + 1, 0);
    for (int r = 0; r < n; r++) { SDRHIP_REQUIRE(!bytes[r] || send_dev[r], SDRHIP_E_INVALID, "rank %d: NULL buffer", r); off[r + 1] = off[r] + bytes[r]; }
    char *dst = static_cast<char *>(recv_dev);
    if (c->use_rccl) {   // grouped point-to-point: every other rank sends its block, the root posts one receive per rank
      c->ctx[root]->use();   // (the root's own block is a copy on its stream: no self send/recv inside the group)
      if (bytes[root]) SDRHIP_CHECK_HIP(hipMemcpyAsync(dst + off[root], send_dev[root], bytes[root], hipMemcpyDeviceToDevice, c->ctx[root]->stream));
      bool any = false;
      for (int r = 0; r < n; r++) any = any || (r != root && bytes[r]);
      if (any) {
        SDRHIP_CHECK_NCCL(rccl().GroupStart());
        for (int r = 0; r < n; r++) {
          if (r == root || !bytes[r]) continue;
          SDRHIP_CHECK_NCCL(rccl().Send(send_dev[r], bytes[r], ncclUint8, root, c->nccl[r], c->ctx[r]->stream));
          SDRHIP_CHECK_NCCL(rccl().Recv(dst + off[r], bytes[r], ncclUint8, r, c->nccl[root], c->ctx[root]->stream));
        }
        SDRHIP_CHECK_NCCL(rccl().GroupEnd());
      }
    } else {   // the root's stream copies each block once the owning rank's stream has produced it ...
      for (int r = 0; r < n; r++) {
        if (!bytes[r]) continue;
        c->ctx[r]->use();
        if (r != root) {
          SDRHIP_CHECK_HIP(hipEventRecord(c->ev[r], c->ctx[r]->stream));
          SDRHIP_CHECK_HIP(hipStreamWaitEvent(c->ctx[root]->stream, c->ev[r], 0));
        }
        SDRHIP_CHECK_HIP(hipMemcpyAsync(dst + off[r], send_dev[r], bytes[r], hipMemcpyDeviceToDevice, c->ctx[root]->stream));
      }
      // ... and no rank's stream may overwrite its send buffer before the root has copied it: with RCCL the send sits on
      // the owning rank's stream; here every rank's stream waits for the root's copies (a caller that enqueues step
      // k + 1 without sdrhip_comm_synchronize would otherwise race the copy of step k)
      c->ctx[root]->use();
      SDRHIP_CHECK_HIP(hipEventRecord(c->ev2[root], c->ctx[root]->stream));
      for (int r = 0; r < n; r++)
        if (r != root && bytes[r]) SDRHIP_CHECK_HIP(hipStreamWaitEvent(c->ctx[r]->stream, c->ev2[root], 0));
    }
  });
}

int sdrhip_comm_gather_begin(sdrhip_comm *c, int slot, const void *const *send_dev, const size_t *bytes, void *recv_dev, int root) {
  return guarded([&] {
    SDRHIP_REQUIRE(c && send_dev && bytes && recv_dev, SDRHIP_E_INVALID, "NULL argument");
    const int n = (int)c->ctx.size();
    SDRHIP_REQUIRE(root >= 0 && root < n, SDRHIP_E_INVALID, "root %d outside [0,%d)", root, n);
    SDRHIP_REQUIRE(slot >= 0 && slot < sdrhip_comm::kSlots, SDRHIP_E_INVALID, "slot %d outside [0,%d)", slot, sdrhip_comm::kSlots);
    std::vector<size_t> off(n + 1, 0);
    for (int r = 0; r < n; r++) { SDRHIP_REQUIRE(!bytes[r] || send_dev[r], SDRHIP_E_INVALID, "rank %d: NULL buffer", r); off[r + 1] = off[r] + bytes[r]; }
    c->make_side();
    char *dst = static_cast<char *>(recv_dev);
    // every rank's side stream starts behind what the rank's own stream has enqueued so far (the kernel that fills send_dev[r])
    for (int r = 0; r < n; r++) {
      c->ctx[r]->use();
      SDRHIP_CHECK_HIP(hipEventRecord(c->ready[r], c->ctx[r]->stream));
      SDRHIP_CHECK_HIP(hipStreamWaitEvent(c->side[r], c->ready[r], 0));
    }
    if (c->use_rccl) {
      c->ctx[root]->use();
      if (bytes[root]) SDRHIP_CHECK_HIP(hipMemcpyAsync(dst + off[root], send_dev[root], bytes[root], hipMemcpyDeviceToDevice, c->side[root]));
      bool any = false;
      for (int r = 0; r < n; r++) any = any || (r != root && bytes[r]);
      if (any) {
        SDRHIP_CHECK_NCCL(rccl().GroupStart());
        for (int r = 0; r < n; r++) {
          if (r == root || !bytes[r]) continue;
          SDRHIP_CHECK_NCCL(rccl().Send(send_dev[r], bytes[r], ncclUint8, root, c->nccl[r], c->side[r]));
          SDRHIP_CHECK_NCCL(rccl().Recv(dst + off[r], bytes[r], ncclUint8, r, c->nccl[root], c->side[root]));
        }
        SDRHIP_CHECK_NCCL(rccl().GroupEnd());
      }
      for (int r = 0; r < n; r++) { c->ctx[r]->use(); SDRHIP_CHECK_HIP(hipEventRecord(c->done[slot][r], c->side[r])); }
    } else {   // one device: the root's side stream copies every block once the owning rank's stream has produced it
      c->ctx[root]->use();
      for (int r = 0; r < n; r++) {
        if (!bytes[r]) continue;
        if (r != root) SDRHIP_CHECK_HIP(hipStreamWaitEvent(c->side[root], c->ready[r], 0));
        SDRHIP_CHECK_HIP(hipMemcpyAsync(dst + off[r], send_dev[r], bytes[r], hipMemcpyDeviceToDevice, c->side[root]));
      }
      for (int r = 0; r < n; r++) SDRHIP_CHECK_HIP(hipEventRecord(c->done[slot][r], c->side[root]));
    }
    c->slot_used[slot] = true;
  });
}

int sdrhip_comm_gather_wait(sdrhip_comm *c, int slot) {
  return guarded([&] {
    SDRHIP_REQUIRE(c, SDRHIP_E_INVALID, "comm is NULL");
    SDRHIP_REQUIRE(slot >= 0 && slot < sdrhip_comm::kSlots, SDRHIP_E_INVALID, "slot %d outside [0,%d)", slot, sdrhip_comm::kSlots);
    if (!c->slot_used[slot]) return;
    const int n = (int)c->ctx.size();
    for (int r = 0; r < n; r++) { c->ctx[r]->use(); SDRHIP_CHECK_HIP(hipStreamWaitEvent(c->ctx[r]->stream, c->done[slot][r], 0)); }
    c->slot_used[slot] = false;
  });
}

int sdrhip_comm_synchronize(sdrhip_comm *c) {
  return guarded([&] {
    SDRHIP_REQUIRE(c, SDRHIP_E_INVALID, "comm is NULL");
    for (size_t r = 0; r < c->ctx.size(); r++) {
      c->ctx[r]->use();
      SDRHIP_CHECK_HIP(hipStreamSynchronize(c->ctx[r]->stream));
      if (!c->side.empty()) SDRHIP_CHECK_HIP(hipStreamSynchronize(c->side[r]));
    }
  });
}

int sdrhip_comm_destroy(sdrhip_comm *c) {
  return guarded([&] {
    if (!c) return;
    for (sdrhip_ctx *x : c->ctx) { x->use(); (void)hipStreamSynchronize(x->stream); }
    for (size_t r = 0; r < c->side.size(); r++) {
      c->ctx[r]->use();
      (void)hipStreamSynchronize(c->side[r]); (void)hipStreamDestroy(c->side[r]); (void)hipEventDestroy(c->ready[r]);
      for (int k = 0; k < sdrhip_comm::kSlots; k++) (void)hipEventDestroy(c->done[k][r]);
    }
    if (c->use_rccl) for (ncclComm_t k : c->nccl) if (k) (void)rccl().CommDestroy(k);
    for (size_t r = 0; r < c->ev.size(); r++) { c->ctx[r]->use(); (void)hipEventDestroy(c->ev[r]); (void)hipEventDestroy(c->ev2[r]); }
    for (sdrhip_ctx *x : c->ctx) (void)sdrhip_ctx_destroy(x);
    delete c;
  });
}

// ---- pinned host memory and asynchronous copies (staging of the many-channel nodes) ------------------------------

int sdrhip_host_alloc(size_t bytes, void **hptr) {
  return guarded([&] {
    SDRHIP_REQUIRE(hptr, SDRHIP_E_INVALID, "hptr is NULL");
    *hptr = nullptr;
    if (bytes) SDRHIP_CHECK_HIP(hipHostMalloc(hptr, bytes, hipHostMallocPortable));
  });
}

int sdrhip_host_free(void *hptr) {
  return guarded([&] { if (hptr) SDRHIP_CHECK_HIP(hipHostFree(hptr)); });
}

int sdrhip_host_register(void *hptr, size_t bytes) {
  return guarded([&] {
    SDRHIP_REQUIRE(hptr && bytes, SDRHIP_E_INVALID, "bad argument");
    SDRHIP_CHECK_HIP(hipHostRegister(hptr, bytes, hipHostRegisterPortable));
  });
}

int sdrhip_host_unregister(void *hptr) {
  return guarded([&] { if (hptr) SDRHIP_CHECK_HIP(hipHostUnregister(hptr)); });
}

int sdrhip_memcpy_h2d_async(sdrhip_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
  return guarded([&] {
    SDRHIP_REQUIRE(ctx, SDRHIP_E_INVALID, "ctx is NULL");
    if (!bytes) return;
    ctx->use();
    SDRHIP_CHECK_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
  });
}

int sdrhip_memcpy_d2h_async(sdrhip_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) {
  return guarded([&] {
    SDRHIP_REQUIRE(ctx, SDRHIP_E_INVALID, "ctx is NULL");
    if (!bytes) return;
    ctx->use();
    SDRHIP_CHECK_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  });
}

int sdrhip_memcpy2d_d2h_async(sdrhip_ctx *ctx, void *dst_host, size_t dst_pitch, const void *src_dev, size_t src_pitch,
                              size_t row_bytes, size_t rows) {
  return guarded([&] {
    SDRHIP_REQUIRE(ctx, SDRHIP_E_INVALID, "ctx is NULL");
    if (!row_bytes || !rows) return;
    ctx->use();
    SDRHIP_CHECK_HIP(hipMemcpy2DAsync(dst_host, dst_pitch, src_dev, src_pitch, row_bytes, rows, hipMemcpyDeviceToHost, ctx->stream));
  });
}

}  // extern "C"
