// Transports for the oversized-trace split (include/sbn.h sbn_comm): the collectives a rank of sbn_split_prover_* calls
// when one trace is proved by several GPUs (BASELINE config[4]; reference workload src/fields/fq12/exp.rs:638-696).
//
//   * RCCL (sbn_rccl_*): one process per GPU, blocks move with ncclSend / ncclRecv inside one group call on the prover's
//     stream -- point-to-point over xGMI, every pair of GPUs has its own link, so the all-to-all of a step uses all seven
//     links of a GPU at once and no ring is involved.  librccl is dlopen()ed: the library carries no link-time dependency
//     and a single-GPU user never loads it.
//   * local (sbn_local_*): the ranks are threads of ONE process (rank r on device r, or all on one device for the parity
//     tests on a one-GPU box).  A receiver pulls its blocks from the senders' staging buffers with stream-ordered copies
//     behind the senders' events; host barriers only line the enqueue calls up, they never wait for the GPU.
//
// Both are stream-ordered as sbn.h asks: nothing here blocks the host on device work except the small host all-gathers.
#include "../../include/sbn.h"
#include "settings.hpp"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only: every entry point is looked up with dlsym
#include <dlfcn.h>
#include <sched.h>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace sbn { extern thread_local std::string g_last_error; int current_device(); }
static int tfail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  sbn::g_last_error = buf;
  return code;
}
#define THIP(expr)                                                                                              \
  do {                                                                                                          \
    hipError_t e_ = (expr);                                                                                     \
    if (e_ != hipSuccess) return tfail(SBN_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// =====================================================================================================================
// RCCL
// =====================================================================================================================
namespace {
struct RcclApi {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;   // optional
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;

int rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.h) return 0;
  const sbn::Settings set = sbn::Settings::from_env_or_default();   // SBN_RCCL_LIB
  const char* names[] = {set.rccl_lib.empty() ? nullptr : set.rccl_lib.c_str(), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* nm : names) if (nm && *nm && (h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!h) return tfail(SBN_ERR_NO_DEVICE, "librccl could not be loaded (set SBN_RCCL_LIB): %s", dlerror());
  RcclApi a; a.h = h;
  bool ok = true;
  auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) ok = false; return p; };
  a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
  a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
  a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
  a.Send = (decltype(a.Send))sym("ncclSend");
  a.Recv = (decltype(a.Recv))sym("ncclRecv");
  a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
  a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
  a.CommAbort = (decltype(a.CommAbort))dlsym(h, "ncclCommAbort");
  if (!ok) { dlclose(h); return tfail(SBN_ERR_NO_DEVICE, "librccl lacks an expected entry point"); }
  g_rccl = a;
  return 0;
}
#define TNCCL(expr)                                                                                                       \
  do {                                                                                                                    \
    ncclResult_t r_ = (expr);                                                                                             \
    if (r_ != ncclSuccess) return tfail(SBN_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
  } while (0)

struct RcclCtx {
  ncclComm_t comm = nullptr;
  uint32_t rank = 0, world = 1;
  int device = 0;
  void *send = nullptr, *recv = nullptr;
  uint64_t send_bytes = 0, recv_bytes = 0;
  double timeout_s = 600;                 // SBN_COMM_TIMEOUT_S at creation
  bool dead = false;                      // a collective timed out or failed: the communicator was aborted, nothing may wait on it again
  hipStream_t hstream = nullptr;          // the host all-gathers' own stream
  void* d_gather = nullptr; size_t gather_bytes = 0;   // [1 + world][bytes] device staging of all_gather_host, grown on demand
  void* h_gather = nullptr; size_t h_gather_bytes = 0; // the same in pinned host memory, OWNED here: a copy that completes after a
                                                       // timeout must not land in a caller's buffer that no longer exists
};
// offsets and lengths of a plan against the staging buffers (a wrong plan must fail here, not inside a collective)
int check_plan(uint32_t world, uint64_t send_bytes, uint64_t recv_bytes, const uint64_t* so, const uint64_t* sl, const uint64_t* ro, const uint64_t* rl) {
  for (uint32_t p = 0; p < world; p++) {
    if (sl[p] && (so[p] > send_bytes || sl[p] > send_bytes - so[p])) return tfail(SBN_ERR_BAD_ARG, "all_to_all: the block for rank %u (offset %llu, %llu bytes) leaves the send buffer of %llu bytes", p, (unsigned long long)so[p], (unsigned long long)sl[p], (unsigned long long)send_bytes);
    if (rl[p] && (ro[p] > recv_bytes || rl[p] > recv_bytes - ro[p])) return tfail(SBN_ERR_BAD_ARG, "all_to_all: the block from rank %u (offset %llu, %llu bytes) leaves the receive buffer of %llu bytes", p, (unsigned long long)ro[p], (unsigned long long)rl[p], (unsigned long long)recv_bytes);
  }
  return 0;
}
void rccl_mark_dead(RcclCtx* c) {
  c->dead = true;
  if (c->comm && g_rccl.CommAbort) { (void)g_rccl.CommAbort(c->comm); c->comm = nullptr; }
}

int rccl_all_to_all(void* vctx, void* vstream, const uint64_t* so, const uint64_t* sl, const uint64_t* ro, const uint64_t* rl) {
  RcclCtx* c = (RcclCtx*)vctx;
  hipStream_t st = (hipStream_t)vstream;
  if (c->dead) return tfail(SBN_ERR_HIP, "all_to_all: this communicator was aborted after an earlier failure");
  if (int rc = check_plan(c->world, c->send_bytes, c->recv_bytes, so, sl, ro, rl)) return rc;
  // One group: every send and receive of the step is posted at once, RCCL runs them concurrently on their own xGMI links.
  // The group is ALWAYS closed: a return between GroupStart and GroupEnd would leave every later RCCL call of this thread
  // (CommDestroy included) nested inside it.
  TNCCL(g_rccl.GroupStart());
  ncclResult_t first = ncclSuccess;
  for (uint32_t p = 0; p < c->world && first == ncclSuccess; p++) {
    if (sl[p]) first = g_rccl.Send((const char*)c->send + so[p], (size_t)sl[p], ncclUint8, (int)p, c->comm, st);
    if (rl[p] && first == ncclSuccess) first = g_rccl.Recv((char*)c->recv + ro[p], (size_t)rl[p], ncclUint8, (int)p, c->comm, st);
  }
  const ncclResult_t end = g_rccl.GroupEnd();
  if (first != ncclSuccess || end != ncclSuccess) {
    const ncclResult_t r = first != ncclSuccess ? first : end;
    rccl_mark_dead(c);
    return tfail(SBN_ERR_HIP, "ncclSend / ncclRecv failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  }
  return 0;
}
int rccl_all_gather_host(void* vctx, const void* send, void* recv, uint64_t bytes) {
  RcclCtx* c = (RcclCtx*)vctx;
  if (bytes == 0) return 0;
  if (c->dead) return tfail(SBN_ERR_HIP, "all_gather_host: this communicator was aborted after an earlier failure");
  THIP(hipSetDevice(c->device));
  const size_t need = (size_t)(1 + c->world) * bytes;
  if (c->gather_bytes < need) {
    if (c->d_gather) (void)hipFree(c->d_gather);
    c->d_gather = nullptr; c->gather_bytes = 0;
    THIP(hipMalloc(&c->d_gather, need));
    c->gather_bytes = need;
  }
  if (c->h_gather_bytes < need) {
    if (c->h_gather) (void)hipHostFree(c->h_gather);
    c->h_gather = nullptr; c->h_gather_bytes = 0;
    THIP(hipHostMalloc(&c->h_gather, need, hipHostMallocDefault));
    c->h_gather_bytes = need;
  }
  char* d_in = (char*)c->d_gather; char* d_out = d_in + bytes;
  char* h_in = (char*)c->h_gather; char* h_out = h_in + bytes;
  memcpy(h_in, send, bytes);
  THIP(hipMemcpyAsync(d_in, h_in, bytes, hipMemcpyHostToDevice, c->hstream));
  { const ncclResult_t r = g_rccl.AllGather(d_in, d_out, (size_t)bytes, ncclUint8, c->comm, c->hstream);
    if (r != ncclSuccess) { rccl_mark_dead(c); return tfail(SBN_ERR_HIP, "ncclAllGather failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"); } }
  THIP(hipMemcpyAsync(h_out, d_out, (size_t)c->world * bytes, hipMemcpyDeviceToHost, c->hstream));
  // a peer that never arrives must fail this rank, not hang it: poll with a deadline (SBN_COMM_TIMEOUT_S, default 600 s).  After
  // a timeout the communicator is ABORTED and marked dead: nothing is left queued behind a collective that will never finish, and
  // sbn_rccl_comm_destroy then frees nothing that the device may still be writing (the process is expected to exit non-zero).
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  for (;;) {
    const hipError_t e = hipStreamQuery(c->hstream);
    if (e == hipSuccess) { memcpy(recv, h_out, (size_t)c->world * bytes); return 0; }
    if (e != hipErrorNotReady) { rccl_mark_dead(c); return tfail(SBN_ERR_HIP, "all_gather_host: %s", hipGetErrorString(e)); }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) {
      rccl_mark_dead(c);
      return tfail(SBN_ERR_HIP, "all_gather_host: no answer from the other ranks within %.0f s (communicator aborted; exit this process)", c->timeout_s);
    }
    if (++spins < 2000) sched_yield(); else std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}
}  // namespace

extern "C" int sbn_rccl_unique_id(uint8_t id_out[128]) {
  if (!id_out) return tfail(SBN_ERR_BAD_ARG, "null argument");
  if (int rc = rccl_load()) return rc;
  static_assert(sizeof(ncclUniqueId) == 128, "sbn.h hands the RCCL unique id around as 128 bytes");
  ncclUniqueId id;
  TNCCL(g_rccl.GetUniqueId(&id));
  memcpy(id_out, &id, 128);
  return SBN_OK;
}
extern "C" int sbn_rccl_comm_create(const uint8_t id[128], uint32_t rank, uint32_t world, uint64_t send_bytes, uint64_t recv_bytes, sbn_comm* out) {
  if (!id || !out || world == 0 || rank >= world) return tfail(SBN_ERR_BAD_ARG, "bad arguments");
  if (int rc = rccl_load()) return rc;
  memset(out, 0, sizeof *out);
  RcclCtx* c = new RcclCtx();
  c->rank = rank; c->world = world; c->send_bytes = send_bytes; c->recv_bytes = recv_bytes;
  c->timeout_s = sbn::Settings::from_env_or_default().comm_timeout_s;
  auto cleanup = [&](int rc) { sbn_comm tmp{}; tmp.ctx = c; tmp.all_to_all = rccl_all_to_all; sbn_rccl_comm_destroy(&tmp); return rc; };
  // the LIBRARY's device of this thread (sbn_set_device / sbn_set_thread_device), not whatever HIP's current device happens to
  // be: staging buffers, the gather stream and ncclCommInitRank must sit on the GPU the prover of this rank will use
  c->device = sbn::current_device();
  if (hipSetDevice(c->device) != hipSuccess) return cleanup(tfail(SBN_ERR_NO_DEVICE, "device %d not available", c->device));
  if (hipMalloc(&c->send, send_bytes ? send_bytes : 8) != hipSuccess || hipMalloc(&c->recv, recv_bytes ? recv_bytes : 8) != hipSuccess)
    return cleanup(tfail(SBN_ERR_HIP, "staging buffers of %llu + %llu bytes do not fit", (unsigned long long)send_bytes, (unsigned long long)recv_bytes));
  if (hipStreamCreate(&c->hstream) != hipSuccess) return cleanup(tfail(SBN_ERR_HIP, "hipStreamCreate failed"));
  ncclUniqueId uid; memcpy(&uid, id, 128);
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, (int)world, uid, (int)rank);
  if (r != ncclSuccess) { c->comm = nullptr; return cleanup(tfail(SBN_ERR_HIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r))); }
  out->struct_size = sizeof(sbn_comm); out->ctx = c; out->rank = rank; out->world = world;
  out->send_buf = c->send; out->recv_buf = c->recv; out->send_bytes = send_bytes; out->recv_bytes = recv_bytes;
  out->all_to_all = rccl_all_to_all; out->all_gather_host = rccl_all_gather_host;
  return SBN_OK;
}
extern "C" void sbn_rccl_comm_destroy(sbn_comm* comm) {
  if (!comm || !comm->ctx || comm->all_to_all != rccl_all_to_all) return;
  RcclCtx* c = (RcclCtx*)comm->ctx;
  (void)hipSetDevice(c->device);
  if (!c->dead) {   // (dead: a collective may still be stuck on the device -- hipFree would wait for it for ever; the memory goes with the process)
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    if (c->hstream) (void)hipStreamDestroy(c->hstream);
    if (c->d_gather) (void)hipFree(c->d_gather);
    if (c->h_gather) (void)hipHostFree(c->h_gather);
    if (c->send) (void)hipFree(c->send);
    if (c->recv) (void)hipFree(c->recv);
  }
  delete c;
  memset(comm, 0, sizeof *comm);
}

// =====================================================================================================================
// local: the ranks are threads of one process
// =====================================================================================================================
struct sbn_local_group {
  uint32_t world = 0;
  std::vector<int> dev;
  std::vector<void*> send, recv;
  std::vector<hipEvent_t> ready, done;            // per rank: "my send blocks are packed" / "I have pulled my blocks"
  std::vector<std::vector<uint64_t>> so, sl;      // per rank: the send plan of the call in flight (read by the receivers)
  std::vector<const void*> hsend; uint64_t hbytes = 0;   // host all-gather
  std::mutex mu; std::condition_variable cv;
  uint32_t waiting = 0; uint64_t generation = 0; bool aborted = false;
  uint64_t send_bytes = 0, recv_bytes = 0;
  double timeout_s = 600;                         // SBN_COMM_TIMEOUT_S at creation: every wait on another rank honours it
  struct RankCtx { sbn_local_group* g; uint32_t rank; };
  std::vector<RankCtx> rctx;
  // All ranks meet; false = the group was aborted (a rank failed) or a rank never came (timeout_s).
  bool barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (aborted) return false;
    const uint64_t gen = generation;
    if (++waiting == world) { waiting = 0; generation++; cv.notify_all(); return true; }
    const bool ok = cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return generation != gen || aborted; });
    if (!ok) { aborted = true; cv.notify_all(); }
    return ok && !aborted;
  }
};
namespace {
using RankCtx = sbn_local_group::RankCtx;

int local_all_to_all_body(RankCtx* rc, hipStream_t st, const uint64_t* so, const uint64_t* sl, const uint64_t* ro, const uint64_t* rl) {
  sbn_local_group* g = rc->g;
  const uint32_t me = rc->rank, R = g->world;
  THIP(hipSetDevice(g->dev[me]));
  if (int prc = check_plan(R, g->send_bytes, g->recv_bytes, so, sl, ro, rl)) return prc;
  g->so[me].assign(so, so + R); g->sl[me].assign(sl, sl + R);
  THIP(hipEventRecord(g->ready[me], st));                     // everything that packed my send blocks precedes this
  if (!g->barrier()) return tfail(SBN_ERR_HIP, "local transport: a rank failed or never arrived");
  for (uint32_t s = 0; s < R; s++) {
    if (!rl[s]) continue;
    if (g->sl[s][me] != rl[s]) return tfail(SBN_ERR_HIP, "local transport: rank %u sends %llu bytes to rank %u, which expects %llu", s, (unsigned long long)g->sl[s][me], me, (unsigned long long)rl[s]);
    if (s != me) THIP(hipStreamWaitEvent(st, g->ready[s], 0));
    const char* src = (const char*)g->send[s] + g->so[s][me];
    char* dst = (char*)g->recv[me] + ro[s];
    if (g->dev[s] == g->dev[me]) THIP(hipMemcpyAsync(dst, src, rl[s], hipMemcpyDeviceToDevice, st));
    else THIP(hipMemcpyPeerAsync(dst, g->dev[me], src, g->dev[s], rl[s], st));   // xGMI peer copy
  }
  THIP(hipEventRecord(g->done[me], st));
  if (!g->barrier()) return tfail(SBN_ERR_HIP, "local transport: a rank failed or never arrived");
  // my send blocks may be overwritten by later work on `st` only after every receiver has pulled them
  for (uint32_t d = 0; d < R; d++) if (d != me && sl[d]) THIP(hipStreamWaitEvent(st, g->done[d], 0));
  return 0;
}
int local_all_to_all(void* vctx, void* vstream, const uint64_t* so, const uint64_t* sl, const uint64_t* ro, const uint64_t* rl) {
  RankCtx* rc = (RankCtx*)vctx;
  const int r = local_all_to_all_body(rc, (hipStream_t)vstream, so, sl, ro, rl);
  if (r) sbn_local_comm_abort(rc->g);   // whatever went wrong on this rank: the others must not wait for it at the next barrier
  return r;
}
int local_all_gather_host(void* vctx, const void* send, void* recv, uint64_t bytes) {
  RankCtx* rc = (RankCtx*)vctx;
  sbn_local_group* g = rc->g;
  g->hsend[rc->rank] = send;
  if (!g->barrier()) return tfail(SBN_ERR_HIP, "local transport: a rank failed or never arrived");
  for (uint32_t s = 0; s < g->world; s++) memcpy((char*)recv + (size_t)s * bytes, g->hsend[s], bytes);
  if (!g->barrier()) return tfail(SBN_ERR_HIP, "local transport: a rank failed or never arrived");   // the senders' buffers stay valid until everybody has read them
  return 0;
}
}  // namespace

extern "C" int sbn_local_comm_create(uint32_t world, const int* devices, uint64_t send_bytes, uint64_t recv_bytes, sbn_comm* comms_out, sbn_local_group** out) {
  if (!comms_out || !out || world == 0 || world > 64) return tfail(SBN_ERR_BAD_ARG, "bad arguments");
  *out = nullptr;
  int cur = 0, ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0 || hipGetDevice(&cur) != hipSuccess) return tfail(SBN_ERR_NO_DEVICE, "no HIP device available");
  sbn_local_group* g = new sbn_local_group();
  g->world = world; g->send_bytes = send_bytes; g->recv_bytes = recv_bytes;
  g->timeout_s = sbn::Settings::from_env_or_default().comm_timeout_s;
  g->dev.resize(world); g->send.assign(world, nullptr); g->recv.assign(world, nullptr); g->ready.assign(world, nullptr); g->done.assign(world, nullptr);
  g->so.resize(world); g->sl.resize(world); g->hsend.assign(world, nullptr); g->rctx.resize(world);
  int rc = 0;
  for (uint32_t r = 0; r < world && !rc; r++) {
    g->dev[r] = devices ? devices[r] : cur;
    if (g->dev[r] < 0 || g->dev[r] >= ndev) { rc = tfail(SBN_ERR_BAD_ARG, "rank %u: device %d not available", r, g->dev[r]); break; }
    if (hipSetDevice(g->dev[r]) != hipSuccess || hipMalloc(&g->send[r], send_bytes ? send_bytes : 8) != hipSuccess || hipMalloc(&g->recv[r], recv_bytes ? recv_bytes : 8) != hipSuccess ||
        hipEventCreateWithFlags(&g->ready[r], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&g->done[r], hipEventDisableTiming) != hipSuccess)
      rc = tfail(SBN_ERR_HIP, "rank %u: staging buffers of %llu + %llu bytes do not fit on device %d", r, (unsigned long long)send_bytes, (unsigned long long)recv_bytes, g->dev[r]);
  }
  if (!rc && devices)   // peer access for the pull copies (ignored if already on, or the same device)
    for (uint32_t a = 0; a < world; a++)
      for (uint32_t b = 0; b < world; b++)
        if (g->dev[a] != g->dev[b]) { (void)hipSetDevice(g->dev[a]); (void)hipDeviceEnablePeerAccess(g->dev[b], 0); (void)hipGetLastError(); }
  (void)hipSetDevice(cur);
  if (rc) { sbn_local_comm_destroy(g); return rc; }
  for (uint32_t r = 0; r < world; r++) {
    g->rctx[r] = {g, r};
    sbn_comm& c = comms_out[r];
    memset(&c, 0, sizeof c);
    c.struct_size = sizeof(sbn_comm); c.ctx = &g->rctx[r]; c.rank = r; c.world = world;
    c.send_buf = g->send[r]; c.recv_buf = g->recv[r]; c.send_bytes = send_bytes; c.recv_bytes = recv_bytes;
    c.all_to_all = local_all_to_all; c.all_gather_host = local_all_gather_host;
  }
  *out = g;
  return SBN_OK;
}
extern "C" void sbn_local_comm_abort(sbn_local_group* g) {
  if (!g) return;
  std::lock_guard<std::mutex> lk(g->mu);
  g->aborted = true;
  g->cv.notify_all();
}
extern "C" void sbn_local_comm_destroy(sbn_local_group* g) {
  if (!g) return;
  int cur = 0; (void)hipGetDevice(&cur);
  for (uint32_t r = 0; r < g->world; r++) {
    (void)hipSetDevice(g->dev[r]);
    if (g->send[r]) (void)hipFree(g->send[r]);
    if (g->recv[r]) (void)hipFree(g->recv[r]);
    if (g->ready[r]) (void)hipEventDestroy(g->ready[r]);
    if (g->done[r]) (void)hipEventDestroy(g->done[r]);
  }
  (void)hipSetDevice(cur);
  delete g;
}

// =====================================================================================================================
// self-test of a transport: a pattern exchange checked on the device
// =====================================================================================================================
namespace {
// byte k of the block rank `src` sends to rank `dst` in round `round`
__host__ __device__ inline uint8_t pattern_byte(uint32_t src, uint32_t dst, uint32_t round, uint64_t k) {
  return (uint8_t)((k * 7 + 131 * src + 31 * dst + 17 * round + (k >> 8)) % 251);
}
__global__ void pattern_fill_kernel(uint8_t* p, uint64_t len, uint32_t src, uint32_t dst, uint32_t round) {
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < len) p[k] = pattern_byte(src, dst, round, k);
}
__global__ void pattern_check_kernel(const uint8_t* p, uint64_t len, uint32_t src, uint32_t dst, uint32_t round, unsigned int* bad) {
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < len && p[k] != pattern_byte(src, dst, round, k)) atomicAdd(bad, 1u);   // bad: this block's own counter
}
}  // namespace

extern "C" int sbn_comm_selftest(const sbn_comm* comm) {
  if (!comm || comm->struct_size != sizeof(sbn_comm) || !comm->all_to_all || !comm->all_gather_host || !comm->send_buf || !comm->recv_buf)
    return tfail(SBN_ERR_BAD_ARG, "bad sbn_comm");
  const uint32_t R = comm->world, me = comm->rank;
  // round 0: uneven blocks, rank s sends (s + 1) * U bytes to every rank (itself included); round 1: one block to everybody
  // (the library's device all-gather) behind the first result; then the host all-gather
  const uint64_t U = 40960, G = 4096;
  const uint64_t need_s = (uint64_t)R * (me + 1) * U, need_r = U * R * (R + 1) / 2 + G * R;
  if (comm->send_bytes < need_s || comm->send_bytes < G || comm->recv_bytes < need_r)
    return tfail(SBN_ERR_BAD_ARG, "selftest needs %llu send / %llu receive bytes of staging", (unsigned long long)need_s, (unsigned long long)need_r);
  hipStream_t st = nullptr; unsigned int* d_bad = nullptr;   // [2][R]: wrong bytes per (round, source rank)
  THIP(hipSetDevice(sbn::current_device()));
  THIP(hipStreamCreate(&st));
  int rc = 0;
  auto blocks = [](uint64_t k) { return dim3((unsigned)((k + 255) / 256)); };
  do {
    if (hipMalloc((void**)&d_bad, 2 * R * sizeof(unsigned int)) != hipSuccess || hipMemsetAsync(d_bad, 0, 2 * R * sizeof(unsigned int), st) != hipSuccess) { rc = tfail(SBN_ERR_HIP, "hipMalloc failed"); break; }
    const uint64_t mine = (me + 1) * U;
    std::vector<uint64_t> so(R), sl(R), ro(R), rl(R);
    uint64_t off = 0;
    for (uint32_t p = 0; p < R; p++) {
      so[p] = p * mine; sl[p] = mine; ro[p] = off; rl[p] = (p + 1) * U; off += rl[p];
      hipLaunchKernelGGL(pattern_fill_kernel, blocks(mine), dim3(256), 0, st, (uint8_t*)comm->send_buf + so[p], mine, me, p, 0u);
    }
    if ((rc = comm->all_to_all(comm->ctx, (void*)st, so.data(), sl.data(), ro.data(), rl.data()))) { rc = tfail(SBN_ERR_HIP, "all_to_all failed (%d): %s", rc, sbn_last_error()); break; }
    for (uint32_t p = 0; p < R; p++)
      hipLaunchKernelGGL(pattern_check_kernel, blocks(rl[p]), dim3(256), 0, st, (const uint8_t*)comm->recv_buf + ro[p], rl[p], p, me, 0u, d_bad + p);
    // all-gather form: the same G bytes to every rank (stream-ordered behind the checks that still read round 0's blocks)
    hipLaunchKernelGGL(pattern_fill_kernel, blocks(G), dim3(256), 0, st, (uint8_t*)comm->send_buf, G, me, 99u, 1u);
    for (uint32_t p = 0; p < R; p++) { so[p] = 0; sl[p] = G; ro[p] = off + p * G; rl[p] = G; }
    if ((rc = comm->all_to_all(comm->ctx, (void*)st, so.data(), sl.data(), ro.data(), rl.data()))) { rc = tfail(SBN_ERR_HIP, "all_to_all (gather form) failed (%d): %s", rc, sbn_last_error()); break; }
    for (uint32_t p = 0; p < R; p++)
      hipLaunchKernelGGL(pattern_check_kernel, blocks(G), dim3(256), 0, st, (const uint8_t*)comm->recv_buf + ro[p], G, p, 99u, 1u, d_bad + R + p);
    std::vector<unsigned int> bad(2 * R, 0);
    if (hipMemcpyAsync(bad.data(), d_bad, 2 * R * sizeof(unsigned int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { rc = tfail(SBN_ERR_HIP, "selftest kernels failed"); break; }
    for (uint32_t k = 0; k < 2 * R && !rc; k++)   // fail fast, naming the receiving rank and the block (= sending rank) that differs
      if (bad[k]) rc = tfail(SBN_ERR_HIP, "transport selftest: on rank %u the block from rank %u of the %s has %u wrong bytes", me, k % R, k < R ? "uneven all-to-all" : "all-gather form", bad[k]);
    if (rc) break;
    std::vector<uint8_t> hs(777), hr((size_t)777 * R);
    for (size_t k = 0; k < hs.size(); k++) hs[k] = pattern_byte(me, 7, 2, k);
    if ((rc = comm->all_gather_host(comm->ctx, hs.data(), hr.data(), hs.size()))) { rc = tfail(SBN_ERR_HIP, "all_gather_host failed (%d): %s", rc, sbn_last_error()); break; }
    for (uint32_t p = 0; p < R && !rc; p++)
      for (size_t k = 0; k < hs.size(); k++)
        if (hr[p * hs.size() + k] != pattern_byte(p, 7, 2, k)) { rc = tfail(SBN_ERR_HIP, "transport selftest: host all-gather block %u is wrong on rank %u", p, me); break; }
  } while (0);
  if (d_bad) (void)hipFree(d_bad);
  (void)hipStreamDestroy(st);
  return rc;
}
