// The exchange step of data-parallel training INSIDE the library (include/tfkaldi_hip.h, "tfk_comm"): the reference's
// only coupling between micro-batches -- G += g, batch_loss += loss, num_frames += n (neuralNetworks/trainer.py:165-169)
// followed by one mean -> clip -> Adam (:174-184) -- as collectives over RCCL / xGMI, launched from C++ on a stream the
// library owns, with no host language in the step.
//
// Protocol (the one tfkaldi_amd/dataparallel.py BucketReducer runs through torch.distributed; that class stays as the
// gloo test double and as the fallback, this file is what runs on RCCL):
//   backward announces gradient buckets (engine hook, tfk_set_bucket_callback) -> adjacent buckets are coalesced until a
//   collective carries >= bucket_bytes -> "sharded": in-place reduce-scatter of the span, Adam on this rank's 1/world of
//   it (tfk_apply_span), in-place all-gather of the updated parameters (mixed precision: of the bf16 weight shadow the
//   forward pass reads, half the bytes; the fp32 masters stay with their owner) -- "allreduce": SUM all-reduce + full Adam.
//   The bias / beta vectors and the scalar + BN tail are always all-reduced.  The next forward pass waits layer by layer
//   for the gather that covers the layer it is about to read (engine hook, tfk_set_layer_callback).
// Streams: collectives run on ONE comm stream per engine, in launch order; the engine stream and the comm stream are
// ordered against each other by events only (gradients ready -> collective -> done -> Adam -> gather -> done -> forward).
//
// Built on the PUBLIC engine ABI only (the same calls a host language would make).  RCCL is bound at run time
// (dlopen "librccl.so.1": the copy the process already holds, e.g. PyTorch's), so the library has no link dependency on it.
// TFK_COMM_LOOPBACK (tests): N engines of ONE process on one GPU form a group; collectives are rendezvous + plain kernels.
// That is how the C++ protocol is exercised at world 2 / 4 / 8 on a single-GPU box (RCCL refuses two ranks per device).
#include "../../include/tfkaldi_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

namespace tfk {
int set_error(int code, const char* msg);  // engine.hip: the library's thread-local error message
}

namespace {

int failx(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return tfk::set_error(code ? code : -1, buf);
}
#define XHIP(expr)                                                                                          \
  do {                                                                                                      \
    hipError_t e_ = (expr);                                                                                 \
    if (e_ != hipSuccess) return failx((int)e_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define XCHK(expr)           \
  do {                       \
    int rc_ = (expr);        \
    if (rc_ != 0) return rc_; \
  } while (0)

// ---- RCCL, bound at run time ----
struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  // (only the bf16 wire format needs these two: bound without complaint, checked where they are used)
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  std::string error;
};
Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // by SONAME: a process that already holds a copy (PyTorch ships its own next to its HIP runtime) gets THAT one
    // (env TFK_RCCL_LIB names another file: a site's own build of RCCL -- or a missing one, to rehearse the failure)
    if (const char* named = getenv("TFK_RCCL_LIB")) {
      r.handle = dlopen(named, RTLD_NOW | RTLD_GLOBAL);
    } else {
      for (const char* name : {"librccl.so.1", "librccl.so"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
      }
    }
    if (!r.handle) {
      const char* why = dlerror();
      r.error = std::string("RCCL could not be loaded: ") + (why ? why : "?");
      return;
    }
#define BIND(field, sym)                                                    \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, sym));     \
  if (!r.field && r.error.empty()) r.error = std::string("librccl lacks ") + sym
    BIND(GetUniqueId, "ncclGetUniqueId");
    BIND(CommInitRank, "ncclCommInitRank");
    BIND(CommDestroy, "ncclCommDestroy");
    BIND(AllReduce, "ncclAllReduce");
    BIND(ReduceScatter, "ncclReduceScatter");
    BIND(AllGather, "ncclAllGather");
    BIND(GetErrorString, "ncclGetErrorString");
    BIND(GroupStart, "ncclGroupStart");
    BIND(GroupEnd, "ncclGroupEnd");
#undef BIND
    r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(r.handle, "ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(r.handle, "ncclRecv"));
  });
  return &r;
}
#define XNCCL(expr)                                                                                              \
  do {                                                                                                           \
    ncclResult_t r_ = (expr);                                                                                    \
    if (r_ != ncclSuccess) return failx((int)r_, "%s failed: %s (%s:%d)", #expr, rccl()->GetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

// ---- the collectives the protocol needs; every one IN PLACE on `world` equal shards ----
struct Backend {
  int rank = 0, world = 1;
  virtual ~Backend() {}
  // buf[0 : world * per_rank) -> this rank's shard [rank * per_rank, +per_rank) holds the SUM over ranks
  virtual int reduce_scatter(float* buf, size_t per_rank, hipStream_t st) = 0;
  // every rank's shard (bytes_per_rank at rank * bytes_per_rank) -> all shards everywhere
  virtual int all_gather(void* buf, size_t bytes_per_rank, hipStream_t st) = 0;
  virtual int all_reduce(float* buf, size_t count, hipStream_t st) = 0;
  // all-to-all of byte blocks: send[q * bytes, +bytes) goes to rank q's recv[rank * bytes, +bytes) for every q != rank (the own
  // block does not travel: recv[rank * bytes, +bytes) is left as it is)
  virtual int exchange_shards(const void* send, void* recv, size_t bytes, hipStream_t st) = 0;
  // all_gather as point-to-point transfers: this rank's shard goes to every peer and every peer's shard arrives in place, all
  // world - 1 links at once (what an all-gather IS on a full mesh of point-to-point links)
  virtual int all_gather_direct(void* buf, size_t bytes_per_rank, hipStream_t st) = 0;
  // host-level: v[0] = min over ranks, v[1] = max over ranks of the value passed in v[0] (blocks the calling thread)
  virtual int min_max(unsigned long long* v, hipStream_t st) = 0;
  // the collectives posted between the two calls (all on ONE stream) may be launched as one operation
  virtual int group_begin() { return 0; }
  virtual int group_end() { return 0; }
  virtual const char* name() const = 0;
};

struct RcclBackend : Backend {
  ncclComm_t comm = nullptr;
  unsigned long long* d_pair = nullptr;
  ~RcclBackend() override {
    if (comm) rccl()->CommDestroy(comm);
    if (d_pair) (void)hipFree(d_pair);
  }
  int reduce_scatter(float* buf, size_t per_rank, hipStream_t st) override {
    XNCCL(rccl()->ReduceScatter(buf, buf + (size_t)rank * per_rank, per_rank, ncclFloat32, ncclSum, comm, st));
    return 0;
  }
  int all_gather(void* buf, size_t bytes_per_rank, hipStream_t st) override {
    char* b = static_cast<char*>(buf);
    XNCCL(rccl()->AllGather(b + (size_t)rank * bytes_per_rank, b, bytes_per_rank, ncclInt8, comm, st));
    return 0;
  }
  int all_reduce(float* buf, size_t count, hipStream_t st) override {
    XNCCL(rccl()->AllReduce(buf, buf, count, ncclFloat32, ncclSum, comm, st));
    return 0;
  }
  int exchange_shards(const void* send, void* recv, size_t bytes, hipStream_t st) override {
    Rccl* r = rccl();
    if (!r->Send || !r->Recv)
      return failx(-1, "this RCCL has no ncclSend / ncclRecv: TFK_DP_ALGO=direct and TFK_DP_WIRE=bf16 are not available");
    const char* s = static_cast<const char*>(send);
    char* d = static_cast<char*>(recv);
    // one group: every rank sends world - 1 blocks and receives world - 1 blocks over its world - 1 links at once -- on
    // point-to-point xGMI this IS the direct reduce-scatter's data movement
    XNCCL(r->GroupStart());
    for (int q = 0; q < world; ++q) {
      if (q == rank) continue;
      ncclResult_t a = r->Send(s + (size_t)q * bytes, bytes, ncclInt8, q, comm, st);
      ncclResult_t b = a == ncclSuccess ? r->Recv(d + (size_t)q * bytes, bytes, ncclInt8, q, comm, st) : a;
      if (b != ncclSuccess) {
        (void)r->GroupEnd();
        return failx((int)b, "ncclSend / ncclRecv failed: %s", r->GetErrorString(b));
      }
    }
    XNCCL(r->GroupEnd());
    return 0;
  }
  int all_gather_direct(void* buf, size_t bytes_per_rank, hipStream_t st) override {
    Rccl* r = rccl();
    if (!r->Send || !r->Recv) return failx(-1, "this RCCL has no ncclSend / ncclRecv: TFK_DP_ALGO=direct is not available");
    char* b = static_cast<char*>(buf);
    XNCCL(r->GroupStart());
    for (int q = 0; q < world; ++q) {
      if (q == rank) continue;
      ncclResult_t a = r->Send(b + (size_t)rank * bytes_per_rank, bytes_per_rank, ncclInt8, q, comm, st);
      ncclResult_t c = a == ncclSuccess ? r->Recv(b + (size_t)q * bytes_per_rank, bytes_per_rank, ncclInt8, q, comm, st) : a;
      if (c != ncclSuccess) {
        (void)r->GroupEnd();
        return failx((int)c, "ncclSend / ncclRecv failed: %s", r->GetErrorString(c));
      }
    }
    XNCCL(r->GroupEnd());
    return 0;
  }
  int min_max(unsigned long long* v, hipStream_t st) override {
    if (!d_pair) XHIP(hipMalloc((void**)&d_pair, 2 * sizeof(unsigned long long)));
    const unsigned long long in[2] = {v[0], v[0]};
    XHIP(hipMemcpyAsync(d_pair, in, sizeof(in), hipMemcpyHostToDevice, st));
    XNCCL(rccl()->AllReduce(d_pair, d_pair, 1, ncclUint64, ncclMin, comm, st));
    XNCCL(rccl()->AllReduce(d_pair + 1, d_pair + 1, 1, ncclUint64, ncclMax, comm, st));
    XHIP(hipMemcpyAsync(v, d_pair, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    XHIP(hipStreamSynchronize(st));
    return 0;
  }
  // (an RCCL call costs its stream 10-13 us of event work whatever it moves -- profiles/r04_dp_trace.txt; a group pays it once)
  int group_begin() override {
    XNCCL(rccl()->GroupStart());
    return 0;
  }
  int group_end() override {
    XNCCL(rccl()->GroupEnd());
    return 0;
  }
  const char* name() const override { return "rccl"; }
};

// ---- loopback: the ranks are threads of one process on one device ----
constexpr int kLoopMaxWorld = 16;
struct SrcList {
  const float* p[kLoopMaxWorld];
};
__global__ void loop_sum_kernel(float* __restrict__ dst, SrcList src, int world, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float s = src.p[0][i];
    for (int q = 1; q < world; ++q) s += src.p[q][i];  // rank order: the order a serial run adds micro-batches in
    dst[i] = s;
  }
}

// ---- bf16 wire format of the gradient reduce-scatter (TFK_DP_WIRE=bf16) ----
// pack: send[i] = bf16(round to nearest even) of g[i], the whole span (world sub-spans of `per` floats back to back)
__global__ void wire_pack_kernel(const float* __restrict__ g, uint16_t* __restrict__ send, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    send[i] = __builtin_bit_cast(uint16_t, (__bf16)g[i]);
}
// sum: shard[i] = sum over ranks q, in rank order, of (q == rank ? this rank's own fp32 value : the bf16 value rank q sent);
// fp32 accumulate on the owner, the owner's own contribution exact (world = 1 is the identity)
__global__ void wire_sum_kernel(float* __restrict__ shard, const uint16_t* __restrict__ recv, int rank, int world, size_t per) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
    float s = rank == 0 ? shard[i] : __builtin_bit_cast(float, (uint32_t)recv[i] << 16);
    for (int q = 1; q < world; ++q)
      s += q == rank ? shard[i] : __builtin_bit_cast(float, (uint32_t)recv[(size_t)q * per + i] << 16);
    shard[i] = s;
  }
}

// ---- the direct reduce-scatter on the fp32 wire (TFK_DP_ALGO=direct) ----
// shard[i] = ((g_0[i] + g_1[i]) + g_2[i]) + ... : the ranks' contributions added in RANK ORDER -- the order a serial run adds its
// micro-batches in, so the owner's sum is that run's G bit for bit (a promise RCCL's own reduce-scatter does not make: its order
// follows its ring).  g_rank is the shard itself, g_q (q != rank) what rank q sent: recv[q * per + i].  per % 4 == 0.
__global__ void direct_sum_kernel(float* __restrict__ shard, const float* __restrict__ recv, int rank, int world, size_t per) {
  const size_t n4 = per / 4;
  float4* out = reinterpret_cast<float4*>(shard);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 own = out[i];
    float4 s = rank == 0 ? own : reinterpret_cast<const float4*>(recv)[i];
    for (int q = 1; q < world; ++q) {
      const float4 v = q == rank ? own : reinterpret_cast<const float4*>(recv + (size_t)q * per)[i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    out[i] = s;
  }
}

}  // namespace

struct tfk_loopback {
  int world = 0;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  unsigned long long generation = 0;
  struct Slot {
    int kind = 0;  // 0 reduce-scatter, 1 all-gather, 2 all-reduce, 3 min / max, 4 all-to-all of byte blocks
    void* buf = nullptr;
    const void* send = nullptr;  // (kind 4: buf = recv)
    size_t count = 0;  // per-rank floats (0), per-rank bytes (1), floats (2)
    hipStream_t st = nullptr;
    hipEvent_t arrive = nullptr, finish = nullptr;
    unsigned long long value = 0;
  };
  std::vector<Slot> slot;
  float* scratch = nullptr;
  size_t scratch_floats = 0;
  hipEvent_t sum_done = nullptr;
  unsigned long long vmin = 0, vmax = 0;
  int error = 0;
  std::string error_text;
};

namespace {

struct LoopBackend : Backend {
  tfk_loopback* g = nullptr;
  hipEvent_t arrive = nullptr, finish = nullptr;
  ~LoopBackend() override {
    if (arrive) (void)hipEventDestroy(arrive);
    if (finish) (void)hipEventDestroy(finish);
  }
  // run by the LAST rank to arrive, with the group's mutex held: enqueue the operation for every rank
  static int run(tfk_loopback* g) {
    const int W = g->world;
    const tfk_loopback::Slot& s0 = g->slot[0];
    for (int r = 1; r < W; ++r)
      if (g->slot[r].kind != s0.kind || g->slot[r].count != s0.count)
        return failx(-1, "loopback group: rank %d posted collective kind %d x %zu, rank 0 kind %d x %zu -- the ranks "
                         "launched different collectives", r, g->slot[r].kind, g->slot[r].count, s0.kind, s0.count);
    if (s0.kind == 3) {
      g->vmin = g->vmax = g->slot[0].value;
      for (int r = 1; r < W; ++r) {
        g->vmin = std::min(g->vmin, g->slot[r].value);
        g->vmax = std::max(g->vmax, g->slot[r].value);
      }
      return 0;
    }
    // one communicator's operations run one after the other whatever streams they were posted on (RCCL orders them itself; the
    // comm posts the tail of a step on the engine stream, the rest on its comm stream): behind the previous operation's end ...
    for (int r = 0; r < W; ++r)
      for (int q = 0; q < W; ++q) XHIP(hipStreamWaitEvent(g->slot[r].st, g->slot[q].finish, 0));
    for (int r = 0; r < W; ++r)  // ... and nobody starts before every rank's operand is ready
      for (int q = 0; q < W; ++q) XHIP(hipStreamWaitEvent(g->slot[r].st, g->slot[q].arrive, 0));
    const size_t n = s0.count;
    if (s0.kind == 0) {
      for (int r = 0; r < W; ++r) {
        SrcList src;
        for (int q = 0; q < W; ++q) src.p[q] = static_cast<const float*>(g->slot[q].buf) + (size_t)r * n;
        float* dst = static_cast<float*>(g->slot[r].buf) + (size_t)r * n;
        const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
        hipLaunchKernelGGL(loop_sum_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, g->slot[r].st, dst, src, W, n);
      }
    } else if (s0.kind == 4) {
      for (int r = 0; r < W; ++r)
        for (int q = 0; q < W; ++q)
          if (q != r)
            XHIP(hipMemcpyAsync(static_cast<char*>(g->slot[r].buf) + (size_t)q * n,
                                static_cast<const char*>(g->slot[q].send) + (size_t)r * n, n, hipMemcpyDeviceToDevice, g->slot[r].st));
    } else if (s0.kind == 1) {
      for (int r = 0; r < W; ++r)
        for (int q = 0; q < W; ++q)
          if (q != r)
            XHIP(hipMemcpyAsync(static_cast<char*>(g->slot[r].buf) + (size_t)q * n,
                                static_cast<const char*>(g->slot[q].buf) + (size_t)q * n, n, hipMemcpyDeviceToDevice,
                                g->slot[r].st));
    } else {
      if (n > g->scratch_floats) {
        if (g->scratch) XHIP(hipFree(g->scratch));  // (synchronises the device: every earlier operation is over)
        g->scratch = nullptr;
        XHIP(hipMalloc((void**)&g->scratch, n * sizeof(float)));
        g->scratch_floats = n;
      }
      SrcList src;
      for (int q = 0; q < W; ++q) src.p[q] = static_cast<const float*>(g->slot[q].buf);
      const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
      hipLaunchKernelGGL(loop_sum_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, g->slot[0].st, g->scratch, src, W, n);
      XHIP(hipEventRecord(g->sum_done, g->slot[0].st));
      for (int r = 0; r < W; ++r) {
        XHIP(hipStreamWaitEvent(g->slot[r].st, g->sum_done, 0));
        XHIP(hipMemcpyAsync(g->slot[r].buf, g->scratch, n * sizeof(float), hipMemcpyDeviceToDevice, g->slot[r].st));
      }
    }
    XHIP(hipGetLastError());
    // ... and nobody's operation is complete before every rank has finished reading its peers' buffers
    for (int r = 0; r < W; ++r) XHIP(hipEventRecord(g->slot[r].finish, g->slot[r].st));
    for (int r = 0; r < W; ++r)
      for (int q = 0; q < W; ++q)
        if (q != r) XHIP(hipStreamWaitEvent(g->slot[r].st, g->slot[q].finish, 0));
    return 0;
  }
  int post(int kind, void* buf, size_t count, hipStream_t st, unsigned long long value = 0, const void* send = nullptr) {
    if (!arrive) {
      XHIP(hipEventCreateWithFlags(&arrive, hipEventDisableTiming));
      XHIP(hipEventCreateWithFlags(&finish, hipEventDisableTiming));
    }
    if (kind != 3) XHIP(hipEventRecord(arrive, st));
    std::unique_lock<std::mutex> lock(g->m);
    if (g->error) return failx(g->error, "loopback group failed earlier: %s", g->error_text.c_str());
    tfk_loopback::Slot& s = g->slot[rank];
    s.kind = kind; s.buf = buf; s.count = count; s.st = st; s.arrive = arrive; s.finish = finish; s.value = value;
    s.send = send;
    const unsigned long long gen = g->generation;
    if (++g->arrived == g->world) {
      const int rc = run(g);
      if (rc) {
        g->error = rc;
        g->error_text = tfk_last_error();
      }
      g->arrived = 0;
      g->generation += 1;
      g->cv.notify_all();
    } else {
      g->cv.wait(lock, [&] { return g->generation != gen; });
    }
    if (g->error) return failx(g->error, "%s", g->error_text.c_str());
    return 0;
  }
  int reduce_scatter(float* buf, size_t per_rank, hipStream_t st) override { return post(0, buf, per_rank, st); }
  int all_gather(void* buf, size_t bytes_per_rank, hipStream_t st) override { return post(1, buf, bytes_per_rank, st); }
  int all_reduce(float* buf, size_t count, hipStream_t st) override { return post(2, buf, count, st); }
  int exchange_shards(const void* send, void* recv, size_t bytes, hipStream_t st) override {
    return post(4, recv, bytes, st, 0, send);
  }
  int all_gather_direct(void* buf, size_t bytes_per_rank, hipStream_t st) override { return post(1, buf, bytes_per_rank, st); }
  int min_max(unsigned long long* v, hipStream_t st) override {
    XCHK(post(3, nullptr, 0, st, v[0]));
    // (the values stay valid until the next operation completes, which needs this rank again)
    v[0] = g->vmin;
    v[1] = g->vmax;
    return 0;
  }
  const char* name() const override { return "loopback"; }
};

struct Span {
  size_t off = 0, n = 0;
  bool rs = false;
  hipEvent_t ready = nullptr, done = nullptr;
  hipEvent_t wait_on = nullptr;  // what the engine stream waits for: `done` of the last piece of the range this span was cut from
  bool waited = false;
};
struct Gather {
  size_t off = 0, n = 0;
  hipEvent_t done = nullptr;
};

}  // namespace

struct tfk_comm {
  tfk_engine* e = nullptr;
  Backend* be = nullptr;
  int mode = TFK_EXCHANGE_SHARDED;
  int device = 0;  // the engine's device: events and the comm stream are created on it, every entry point selects it
  size_t min_floats = 0, min_shard_floats = 1 << 14;
  hipStream_t comm_stream = nullptr, engine_stream = nullptr;
  float *grad = nullptr, *param = nullptr;
  void* shadow = nullptr;  // bf16 weight shadow mirroring the arena (mixed precision), else NULL
  size_t reduce_floats = 0, num_params = 0, vec_off = 0;
  std::vector<std::pair<size_t, size_t>> buckets;  // (offset, floats), index = the engine's bucket number
  int L = 0;

  bool have_range = false;
  size_t lo = 0, hi = 0;
  bool have_hold = false, have_head = false;  // ranges held back for the inline tail (hold_last): the last weight range; [scalars | E]
  size_t hold_lo = 0, hold_hi = 0, head_lo = 0, head_hi = 0;
  std::vector<Span> spans;  // collectives of the current step, launch order
  size_t num_spans = 0;
  size_t waited_upto = 0;       // spans [0, waited_upto) of this step are complete as far as the engine stream is concerned
  std::vector<Gather> pending;  // parameter gathers nobody has waited for yet (ascending offsets = forward order)
  std::vector<hipEvent_t> gather_events;
  size_t gathers_used = 0;
  hipEvent_t ev_adam = nullptr;
  bool masters_stale = false;
  std::vector<std::pair<size_t, size_t>> shard_spans;
  // every piece this comm has sharded since the shards were last assigned: Adam's moments of a piece are current on its owner
  // only (the optimiser state is sharded with the optimiser), so a change of the assignment gathers them first (rehome)
  std::vector<std::pair<size_t, size_t>> owned_pieces;
  float *adam_m = nullptr, *adam_v = nullptr;
  int verify_left = 2;
  bool apply_enqueued = false, apply_sharded = false, apply_via_shadow = false, apply_planes = false;  // between _enqueue and _end
  // TFK_DP_WIRE=bf16: reduce-scattered spans travel as bf16 (2 B per parameter in instead of 4), summed in fp32 by the owner
  bool wire_bf16 = false;
  // TFK_DP_ALGO: how a reduce-scattered span / a parameter gather travels -- RCCL's own collective (its algorithm, its
  // summation order) or DIRECT: grouped ncclSend / ncclRecv of the sub-spans to / from all world - 1 peers at once and, for the
  // reduce-scatter, the owner's sum in rank order (direct_sum_kernel).  `auto`: both are timed at attach on scratch memory and
  // the faster one is kept, per operation (tfk_comm_tune)
  int algo_rs = TFK_ALGO_RCCL, algo_ag = TFK_ALGO_RCCL;
  // TFK_DP_GATHER=planes (emulated fp32, sharded): the sharding unit becomes the weight MATRIX instead of the coalesced span --
  // rank r owns rows [r R / W, (r + 1) R / W) of every matrix of a span; collectives are still launched per span, as one group
  // -- and what the owner's Adam wrote for a matrix, its three-plane twin rows, is gathered (6 B per weight) in place of the fp32
  // parameters (4 B) + a rebuild on every rank.  Matrices whose rows do not divide by 2 W (row pairs are the twin's unit) keep the
  // fp32 gather + rebuild.  The fp32 masters of a plane-gathered matrix stay with their owner (tfk_comm_gather_masters).
  bool gather_planes = false;
  struct TwinOf { void* ptr = nullptr; size_t bytes = 0; int rows = 0; };
  std::vector<TwinOf> twin;  // per layer (0 .. L); bytes == 0: nothing to gather in place of the parameters
  int chosen_by = 0;  // 0 default, 1 environment, 2 tuned, 3 tfk_comm_set_exchange
  double tune_us[4] = {0, 0, 0, 0};  // reduce-scatter rccl / direct, all-gather rccl / direct (max over ranks; 0: not tuned)
  // staging of the direct / bf16 exchanges: 4 B per parameter, EVERY span its own region (at its arena offset) -- spans are
  // launched on two streams (comm stream under backward, engine stream for the tail) and must never share staging memory
  //   fp32 direct: peers' sub-spans arrive at (float*)stage + span offset + q * per
  //   bf16 wire:   packed span at (uint16_t*)stage + span offset, arrivals at (uint16_t*)stage + num_params + span offset + q * per
  void* stage = nullptr;
  // per-phase device times (tfk_comm_timing): pairs of timing events per phase, summed when read
  struct TimedPair { int phase; hipEvent_t a, b; };
  bool timing = false;
  std::vector<hipEvent_t> timing_pool;
  size_t timing_used = 0;
  std::vector<TimedPair> timed;
  long timed_steps = 0;
  int error = 0;  // a failure inside an engine hook (cannot propagate through the hook): raised by the next call
  std::string error_text;
  // what ran in the last completed step (tfk_comm_last_step)
  int last_rs = 0, last_ag = 0, last_ar = 0, cur_rs = 0, cur_ag = 0, cur_ar = 0;
  std::vector<std::pair<size_t, size_t>> last_spans;
  std::vector<int> last_span_rs;
};

namespace {

constexpr size_t kDefaultBucketBytes = (size_t)32 << 20;  // (tfk_comm_create: why)

int new_event(hipEvent_t* ev) {
  XHIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
  return 0;
}

// What a rank owns of a reduce-scattered span is cut per PIECE: the span itself (one piece: rank r owns the r-th of world equal
// sub-spans) or, under gather_planes, each weight matrix of the span (rank r owns the r-th of world equal row blocks of it).
struct Piece {
  size_t off = 0, n = 0;
  int layer = -1;       // the weight matrix this piece is (gather_planes), else -1
  bool planes = false;  // its twin rows are gathered in place of its parameters
};
std::vector<Piece> pieces_of(const tfk_comm* c, size_t off, size_t n) {
  std::vector<Piece> out;
  if (!c->gather_planes) {
    Piece p;
    p.off = off; p.n = n;
    out.push_back(p);
    return out;
  }
  for (int b = 0; b <= c->L; ++b) {  // bucket b = weight matrix of layer L - b
    const size_t bo = c->buckets[b].first, bn = c->buckets[b].second;
    if (bo < off || bo + bn > off + n) continue;
    Piece p;
    p.off = bo; p.n = bn; p.layer = c->L - b;
    const tfk_comm::TwinOf& t = c->twin[p.layer];
    p.planes = t.bytes > 0 && t.rows % (2 * c->be->world) == 0;
    out.push_back(p);
  }
  std::sort(out.begin(), out.end(), [](const Piece& a, const Piece& b) { return a.off < b.off; });
  return out;
}

bool shardable(const tfk_comm* c, size_t lo, size_t hi) {
  const size_t n = hi - lo;
  if (!(c->mode == TFK_EXCHANGE_SHARDED && hi <= c->vec_off && n >= c->min_shard_floats)) return false;
  size_t covered = 0;
  for (const Piece& p : pieces_of(c, lo, n)) {
    if (p.n % (4 * (size_t)c->be->world) != 0) return false;
    covered += p.n;
  }
  return covered == n;  // (a span is a union of whole weight matrices: anything else is all-reduced)
}

// One coalesced range of gradients is ready on the engine stream: its collective(s) go to the comm stream.
// In the sharded mode a range that runs from the weight matrices into the bias / beta vectors, or from the gradient arena into
// the scalar + BN tail, is cut there: only weight matrices are sharded (the vectors are a few thousand values, all-reduced and
// updated on every rank, so that no layer ever waits for THEIR gather).  The pieces of one range share ONE `ready` record on the
// engine stream and ONE `done` record behind the last of them: an event record between two kernels costs the engine stream
// ~6 us of idle time and every record / wait pair on the comm stream ~9 us of latency before the optimiser may start
// (profiles/r04_dp_trace.txt) -- the range launched by tfk_comm_apply, after the last backward kernel, has two or three pieces;
// they also go to RCCL as one group (one launch).
// inline_on_engine: nothing is left to overlap with (the range launched by tfk_comm_apply behind the last backward kernel, the
// evaluation sums): the collectives go to the ENGINE stream itself and no event is needed at all -- the record -> comm stream
// -> record -> engine stream round trip measured ~26 us of idle time in front of the optimiser.  (One communicator on two
// streams is legal for RCCL: it orders its operations itself.)
int ensure_stage(tfk_comm* c) {
  if (c->stage) return 0;
  XHIP(hipMalloc(&c->stage, std::max<size_t>(c->num_params, 1) * sizeof(float)));
  return 0;
}

enum { PH_RS = 0, PH_AR = 1, PH_TAIL_EXPOSED = 2, PH_ADAM = 3, PH_AG = 4, PH_TWINS = 5, PH_GATHER_EXPOSED = 6, PH_COUNT = 7 };
// a pair of timing events around a piece of stream work; nothing at all unless tfk_comm_timing switched it on (an event record
// between two kernels costs the stream ~6 us: profiles/r04_dp_trace.txt -- the timed steps are a diagnostic pass, never `value`)
struct Timed {
  tfk_comm* c;
  hipStream_t st;
  int phase;
  hipEvent_t a = nullptr;
  int rc = 0;
  static hipEvent_t take(tfk_comm* c) {
    if (c->timing_used == c->timing_pool.size()) {
      hipEvent_t ev = nullptr;
      if (hipEventCreate(&ev) != hipSuccess) return nullptr;
      c->timing_pool.push_back(ev);
    }
    return c->timing_pool[c->timing_used++];
  }
  Timed(tfk_comm* c_, int phase_, hipStream_t st_) : c(c_), st(st_), phase(phase_) {
    if (!c->timing || phase < 0) return;
    a = take(c);
    if (!a || hipEventRecord(a, st) != hipSuccess) rc = failx(-1, "timing event could not be recorded");
  }
  int end() {
    if (!c->timing || rc || !a || phase < 0) return rc;
    hipEvent_t b = take(c);
    if (!b || hipEventRecord(b, st) != hipSuccess) return failx(-1, "timing event could not be recorded");
    c->timed.push_back({phase, a, b});
    a = nullptr;
    return 0;
  }
};

struct Group {  // a backend group, closed on every way out: a failed collective must not leave RCCL inside a group
  Backend* be;
  bool open;
  ~Group() { if (open) (void)be->group_end(); }
  int begin() {
    XCHK(be->group_begin());
    open = true;
    return 0;
  }
  int end() {
    open = false;
    return be->group_end();
  }
};

// a parameter gather of `world` equal shards in place, by the algorithm in force
int gather_span(tfk_comm* c, void* buf, size_t bytes_per_rank, hipStream_t st, bool timed = true) {
  Timed t(c, timed ? PH_AG : -1, st);
  if (c->algo_ag == TFK_ALGO_DIRECT) XCHK(c->be->all_gather_direct(buf, bytes_per_rank, st));
  else XCHK(c->be->all_gather(buf, bytes_per_rank, st));
  return t.end();
}

bool inline_tail() {
  static const bool on = !getenv("TFK_DP_INLINE_TAIL") || atoi(getenv("TFK_DP_INLINE_TAIL")) != 0;
  return on;
}
int launch_range(tfk_comm* c, size_t lo, size_t hi, bool inline_on_engine = false) {
  size_t cuts[4] = {lo, 0, 0, 0};
  int pieces = 1;
  if (c->mode == TFK_EXCHANGE_SHARDED)
    for (size_t cut : {c->vec_off, c->num_params})
      if (lo < cut && cut < hi) cuts[pieces++] = cut;
  cuts[pieces] = hi;
  const size_t first = c->num_spans;
  for (int k = 0; k < pieces; ++k) {
    if (c->num_spans == c->spans.size()) {
      Span s;
      XCHK(new_event(&s.ready));
      XCHK(new_event(&s.done));
      c->spans.push_back(s);
    }
    Span& s = c->spans[c->num_spans++];
    s.off = cuts[k]; s.n = cuts[k + 1] - cuts[k]; s.waited = false;
    s.rs = shardable(c, s.off, s.off + s.n);
  }
  hipStream_t st = inline_on_engine ? c->engine_stream : c->comm_stream;
  if (!inline_on_engine) {
    XHIP(hipEventRecord(c->spans[first].ready, c->engine_stream));
    XHIP(hipStreamWaitEvent(c->comm_stream, c->spans[first].ready, 0));
  }
  // reduce-scattered spans that travel as point-to-point transfers (direct algorithm, bf16 wire) first, each on its own: the
  // exchange must have been LAUNCHED, not deferred to the end of a group, when the owner's sum is enqueued behind it.  The
  // pieces of one span (gather_planes: its weight matrices) go out as ONE group -- one launch -- with the sums behind it.
  const bool p2p = c->wire_bf16 || c->algo_rs == TFK_ALGO_DIRECT;
  const int W = c->be->world, R = c->be->rank;
  size_t rest = 0;  // operations of the second pass
  for (size_t k = first; k < c->num_spans; ++k) {
    Span& s = c->spans[k];
    s.wait_on = inline_on_engine ? nullptr : c->spans[c->num_spans - 1].done;
    const std::vector<Piece> pcs = s.rs ? pieces_of(c, s.off, s.n) : std::vector<Piece>();
    if (!(s.rs && p2p)) {
      rest += s.rs ? pcs.size() : 1;
      continue;
    }
    XCHK(ensure_stage(c));
    Timed t(c, PH_RS, st);
    Group g = {c->be, false};
    if (c->wire_bf16) {
      // (the whole span is packed at its arena offset in the first half of the staging buffer; arrivals land at the same offset
      //  in the second half)
      const unsigned blocks = (unsigned)std::min<size_t>((s.n + 255) / 256, 1 << 14);
      hipLaunchKernelGGL(wire_pack_kernel, dim3(blocks), dim3(256), 0, st, c->grad + s.off, static_cast<uint16_t*>(c->stage) + s.off, s.n);
    }
    if (pcs.size() > 1) XCHK(g.begin());
    for (const Piece& p : pcs) {
      const size_t per = p.n / W;
      if (c->wire_bf16)
        XCHK(c->be->exchange_shards(static_cast<uint16_t*>(c->stage) + p.off, static_cast<uint16_t*>(c->stage) + c->num_params + p.off,
                                    per * sizeof(uint16_t), st));
      else
        XCHK(c->be->exchange_shards(c->grad + p.off, static_cast<float*>(c->stage) + p.off, per * sizeof(float), st));
    }
    if (pcs.size() > 1) XCHK(g.end());
    for (const Piece& p : pcs) {
      const size_t per = p.n / W;
      if (c->wire_bf16) {
        const unsigned sb = (unsigned)std::min<size_t>((per + 255) / 256, 1 << 14);
        hipLaunchKernelGGL(wire_sum_kernel, dim3(sb), dim3(256), 0, st, c->grad + p.off + (size_t)R * per,
                           static_cast<uint16_t*>(c->stage) + c->num_params + p.off, R, W, per);
      } else {
        const unsigned sb = (unsigned)std::min<size_t>((per / 4 + 255) / 256, 1 << 13);
        hipLaunchKernelGGL(direct_sum_kernel, dim3(sb ? sb : 1), dim3(256), 0, st, c->grad + p.off + (size_t)R * per,
                           static_cast<float*>(c->stage) + p.off, R, W, per);
      }
    }
    XHIP(hipGetLastError());
    XCHK(t.end());
    c->cur_rs += 1;
  }
  const bool grouped = rest > 1;
  Group group = {c->be, false};
  // (a group launches as one operation: timed as one and booked as all-reduce time -- in practice the vectors + the scalar tail)
  int lone_phase = PH_AR;
  for (size_t k = first; k < c->num_spans; ++k)
    if (c->spans[k].rs && !p2p) lone_phase = PH_RS;
  Timed tg(c, lone_phase, st);
  if (grouped) XCHK(group.begin());
  for (size_t k = first; k < c->num_spans; ++k) {
    Span& s = c->spans[k];
    if (s.rs && p2p) continue;
    if (s.rs) {
      for (const Piece& p : pieces_of(c, s.off, s.n)) XCHK(c->be->reduce_scatter(c->grad + p.off, p.n / W, st));
      c->cur_rs += 1;
    } else {
      XCHK(c->be->all_reduce(c->grad + s.off, s.n, st));
      c->cur_ar += 1;
    }
  }
  if (grouped) XCHK(group.end());
  if (rest) XCHK(tg.end());
  if (!inline_on_engine) XHIP(hipEventRecord(c->spans[c->num_spans - 1].done, c->comm_stream));
  return 0;
}

// env TFK_DP_HOLD_LAST (default on; needs the inline tail): the weight range still coalescing when the step's LAST announcement
// arrives (the bias / beta vectors, behind the last backward kernel) is not flushed to the comm stream -- nothing is left to
// overlap it with, and the `ready` record + comm stream + `done` wait round trip showed as ~6 + ~16 us of idle engine stream
// in front of the optimiser (profiles/r06_dp_trace.txt) -- but held and launched with the vectors on the engine stream itself,
// where RCCL's own collectives go out as ONE group (one launch).  The step's FIRST announcement ([scalars | BN increments])
// joins that group too instead of costing the engine stream an event record in the middle of the backward pass.
bool hold_last() {
  static const bool on = inline_tail() && (!getenv("TFK_DP_HOLD_LAST") || atoi(getenv("TFK_DP_HOLD_LAST")) != 0);
  return on;
}

int flush_range(tfk_comm* c, bool inline_on_engine = false) {
  if (inline_on_engine && (c->have_head || c->have_hold)) {
    // the tail of a step: [scalars + BN increments] held since the first announcement, the weight range held since the last one,
    // the vectors -- on the engine stream, behind the last backward kernel
    const bool p2p = c->mode == TFK_EXCHANGE_SHARDED && (c->wire_bf16 || c->algo_rs == TFK_ALGO_DIRECT);
    // (point-to-point reduce-scatters enqueue the owner's sum behind their exchange and cannot wait for an outer group's end;
    //  under tfk_comm_timing every operation keeps its own launch so that its events bracket it)
    const bool one_launch = !p2p && !c->timing;
    Group g = {c->be, false};
    if (one_launch) XCHK(g.begin());
    if (c->have_head) {
      c->have_head = false;
      XCHK(launch_range(c, c->head_lo, c->head_hi, true));
    }
    if (c->have_hold) {
      c->have_hold = false;
      XCHK(launch_range(c, c->hold_lo, c->hold_hi, true));
    }
    if (c->have_range) {
      c->have_range = false;
      XCHK(launch_range(c, c->lo, c->hi, true));
    }
    return one_launch ? g.end() : 0;
  }
  if (c->have_hold) {  // (a flush in the middle of a step: the held weight range goes first, in announcement order)
    c->have_hold = false;
    XCHK(launch_range(c, c->hold_lo, c->hold_hi, inline_on_engine));
  }
  if (!c->have_range) return 0;
  c->have_range = false;
  return launch_range(c, c->lo, c->hi, inline_on_engine);
}

int announce(tfk_comm* c, int b) {
  if (b < 0 || b >= (int)c->buckets.size()) return failx(-1, "bucket %d out of range", b);
  const size_t off = c->buckets[b].first, n = c->buckets[b].second;
  if (b == c->L + 2 && !c->have_range && hold_last()) {
    // [scalars + BN increments]: always the FIRST announcement of a step.  Its all-reduce used to go to the comm stream at once
    // (a `ready` record between two kernels of the engine stream: ~6 us of idle time) so as to be long done when the optimiser
    // needs the frame count; in the tail's one group launch it costs nothing and needs no event.  (Announced again before the
    // tail went out -- a step the host gave up and started over -- it is still ONE all-reduce: the sums must not be taken twice)
    c->have_head = true;
    c->head_lo = off;
    c->head_hi = off + n;
    return 0;
  }
  if (c->have_range && off + n == c->lo) {
    c->lo = off;
  } else if (c->have_range && off == c->hi) {
    c->hi = off + n;
  } else if (c->have_range && b == c->L + 1 && !c->have_hold && c->hi <= c->vec_off && hold_last()) {
    // the vectors: always the last announcement of a step (engine.hip: backward, tfk_comm_idle) -- the pending weight range waits
    // for tfk_comm_apply_enqueue / tfk_comm_finish_reduce, which follow at once, instead of taking the comm-stream round trip
    c->have_hold = true;
    c->hold_lo = c->lo;
    c->hold_hi = c->hi;
    c->lo = off;
    c->hi = off + n;
  } else {
    XCHK(flush_range(c));
    c->have_range = true;
    c->lo = off;
    c->hi = off + n;
  }
  if (c->hi - c->lo >= c->min_floats) XCHK(flush_range(c));
  return 0;
}

void remember(tfk_comm* c, int rc) {
  if (rc && !c->error) {
    c->error = rc;
    c->error_text = tfk_last_error();
  }
}
int raise_remembered(tfk_comm* c) {
  if (!c->error) return 0;
  const int rc = c->error;
  c->error = 0;
  return failx(rc, "%s", c->error_text.c_str());
}

double g_bucket_us = 0;  // (TFK_DP_HOST_PHASES) host time inside the bucket callback, i.e. the collectives launched under backward
void on_bucket(void* user, int b) {
  tfk_comm* c = static_cast<tfk_comm*>(user);
  static const bool timed = getenv("TFK_DP_HOST_PHASES") != nullptr;
  if (!timed) {
    remember(c, announce(c, b));
    return;
  }
  const auto t0 = std::chrono::steady_clock::now();
  remember(c, announce(c, b));
  g_bucket_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

int drain(tfk_comm* c) {
  for (const Gather& g : c->pending) XHIP(hipStreamWaitEvent(c->engine_stream, g.done, 0));
  c->pending.clear();
  return 0;
}

int wait_layer(tfk_comm* c, int layer) {
  if (c->pending.empty()) return 0;
  if (layer < 0) return drain(c);
  const std::pair<size_t, size_t> want[2] = {c->buckets[c->L - layer], c->buckets[c->L + 1]};
  int last = -1;
  for (size_t i = 0; i < c->pending.size(); ++i)
    for (const auto& w : want)
      if (c->pending[i].off < w.first + w.second && c->pending[i].off + c->pending[i].n > w.first) last = (int)i;
  if (last < 0) return 0;
  // (the gathers run in launch order on one stream: the last one that matters implies the earlier ones)
  Timed exposed(c, PH_GATHER_EXPOSED, c->engine_stream);  // what the forward pass really waits: the gap this wait opens
  XHIP(hipStreamWaitEvent(c->engine_stream, c->pending[last].done, 0));
  XCHK(exposed.end());
  c->pending.erase(c->pending.begin(), c->pending.begin() + last + 1);
  return 0;
}

// Emulated fp32: the contractions read three-plane twins of the weights, and a gather has just replaced the fp32 values of a
// span behind the optimiser's back.  The twins of that span are rebuilt on the stream the gather ran on, right behind it --
// under the forward pass of the layers below (the engine waits for `done` layer by layer) -- instead of all of them, every
// gather awaited, in front of the next pass (what tfk_params_touched amounts to).
int twins_behind_gather(tfk_comm* c, const std::pair<size_t, size_t>& span, hipStream_t st, bool* all_current) {
  int current = 0;
  Timed t(c, PH_TWINS, st);
  XCHK(tfk_twins_from_params(c->e, span.first, span.second, st, &current));
  if (!current) *all_current = false;
  return t.end();
}

// The gathers of one reduce-scattered span, launched as one group: per piece the shard owner's fp32 parameters (or bf16 shadow),
// or -- gather_planes, a matrix whose rows divide -- the twin rows its Adam wrote.  What the contractions read of the fp32-gathered
// pieces is rebuilt behind the group on the same stream.
int gather_pieces(tfk_comm* c, const std::pair<size_t, size_t>& span, hipStream_t st, bool via_shadow, bool* derived_current,
                  bool* planes_any) {
  const std::vector<Piece> pcs = pieces_of(c, span.first, span.second);
  const int W = c->be->world;
  char* target = static_cast<char*>(via_shadow ? c->shadow : (void*)c->param);
  const size_t elem = via_shadow ? 2 : 4;
  {
    Timed t(c, PH_AG, st);
    Group g = {c->be, false};
    if (pcs.size() > 1) XCHK(g.begin());
    for (const Piece& p : pcs) {
      if (p.planes) {
        const tfk_comm::TwinOf& tw = c->twin[p.layer];
        XCHK(gather_span(c, tw.ptr, tw.bytes / W, st, false));
        *planes_any = true;
        const std::pair<size_t, size_t> piece(p.off, p.n);
        if (std::find(c->shard_spans.begin(), c->shard_spans.end(), piece) == c->shard_spans.end()) c->shard_spans.push_back(piece);
      } else {
        XCHK(gather_span(c, target + p.off * elem, p.n / W * elem, st, false));
      }
    }
    if (pcs.size() > 1) XCHK(g.end());
    XCHK(t.end());
  }
  if (!via_shadow)
    for (const Piece& p : pcs)
      if (!p.planes) XCHK(twins_behind_gather(c, {p.off, p.n}, st, derived_current));
  return 0;
}

void on_layer(void* user, int layer) {
  tfk_comm* c = static_cast<tfk_comm*>(user);
  remember(c, wait_layer(c, layer));
}

// The engine stream waits for span s's collective.  The comm stream runs its collectives in launch order (= span index), so a
// wait for span j covers every span before it: one is only issued for a span beyond the last one waited for.  (A stream wait
// whose event fired long ago costs the engine stream nothing measurable; what costs are event RECORDS and round trips through
// the other stream -- profiles/r04_dp_trace.txt.)
int wait_span(tfk_comm* c, Span& s) {
  const size_t index = (size_t)(&s - c->spans.data());
  if (!s.waited && index >= c->waited_upto) {
    if (s.wait_on) XHIP(hipStreamWaitEvent(c->engine_stream, s.wait_on, 0));  // (nullptr: it ran on the engine stream itself)
    if (s.wait_on) c->waited_upto = index + 1;
  }
  s.waited = true;
  return 0;
}

int verify_replicas(tfk_comm* c, bool via_shadow, bool planes) {
  XCHK(drain(c));
  // what every rank must agree on after the gathers: the fp32 parameters -- or, where the masters stay with their owners, what
  // the contractions read (bf16 shadow / every three-plane twin) + the all-reduced vectors
  const int which[2] = {via_shadow ? 1 : planes ? 3 : 0, 2};
  for (int k = 0; k < ((via_shadow || planes) ? 2 : 1); ++k) {
    uint64_t sum = 0;
    XCHK(tfk_param_checksum(c->e, which[k], &sum));
    unsigned long long v[2] = {(unsigned long long)sum, 0};
    XCHK(c->be->min_max(v, c->comm_stream));
    if (v[0] != v[1])
      return failx(-1, "data-parallel replicas diverged after the sharded exchange step: rank %d holds checksum %llu of "
                       "the %s, the ranks' values span %llu .. %llu", c->be->rank, (unsigned long long)sum,
                   which[k] == 0 ? "fp32 parameters" : which[k] == 1 ? "bf16 shadow" : which[k] == 3 ? "three-plane twins"
                                                                                                       : "bias / beta vectors", v[0], v[1]);
  }
  return 0;
}

int attach(tfk_engine* e, Backend* be, int mode, size_t bucket_bytes, tfk_comm** out) {
  tfk_comm* c = new tfk_comm;
  c->e = e;
  c->be = be;
  auto bail = [&](int rc) {
    tfk_comm_destroy(c);
    return rc;
  };
  if (mode != TFK_EXCHANGE_SHARDED && mode != TFK_EXCHANGE_ALLREDUCE) return bail(failx(-1, "unknown exchange mode %d", mode));
  c->mode = mode;
  // default: 32 MiB per collective (kDefaultBucketBytes).  Every collective launched under backward costs the engine stream a
  // fixed ~6-10 us (an event record between two backward kernels, RCCL's own stream work) whatever it carries -- with ONE RCCL rank
  // fewer spans are simply cheaper (BASELINE cfg3: +11.5 % with 24 MiB spans, +6.4 % with 64: profiles/r04_dp_overhead*.txt), and
  // rounds 4-6 kept 64 MiB on that evidence.  With wire time it is the other way round: the next forward pass cannot start a layer
  // before the WHOLE gather covering it has arrived, and what is still coalescing when backward ends is reduce-scattered with
  // nothing left to hide it under -- at cfg2 the 64 MiB rule cuts [W6 .. W2] (83 MB) + [W1, W0] (20 MB), the 32 MiB rule
  // [W6 .. W4] + [W3, W2] + [W1, W0].  dataparallel.exchange_timeline prices both over a range of wire rates and latencies (no
  // multi-GPU node was available to measure them): 32 MiB is ahead at cfg2 in every cell at 2 / 4 / 8 ranks (1-15 %), level at
  // cfg3, and cfg4's 64 MiB matrices are spans of their own either way.  bench.py --gpus N sweeps 16 / 32 / 64 / 128 on the real links.
  if (bucket_bytes == 0) bucket_bytes = kDefaultBucketBytes;
  c->min_floats = std::max<size_t>(1, bucket_bytes / 4);
  if (const char* v = getenv("TFK_DP_MIN_SHARD")) c->min_shard_floats = (size_t)atol(v);
  if (const char* v = getenv("TFK_DP_VERIFY_STEPS")) c->verify_left = atoi(v);
  if (const char* v = getenv("TFK_DP_WIRE")) {
    if (!strcmp(v, "bf16")) c->wire_bf16 = true;
    else if (strcmp(v, "fp32") && strcmp(v, "float32")) return bail(failx(-1, "TFK_DP_WIRE=%s (fp32 | bf16)", v));
  }
  bool tune = false;
  {
    // default: RCCL's own collectives, the code path every RCCL user runs.  `auto` times both at attach and keeps the faster one
    // per operation -- opt-in until a multi-GPU node has run it (no round of this build had one): a tuning pass that went wrong
    // would take the whole job with it, the A/B of bench.py --gpus N only its diagnostics (it runs behind the timed region)
    const char* v = getenv("TFK_DP_ALGO");
    if (!v || !strcmp(v, "rccl")) {
      c->chosen_by = v ? 1 : 0;
    } else if (!strcmp(v, "auto")) {
      // (only where there is something to choose between: more than one rank of a real RCCL group, sharded mode)
      tune = !strcmp(be->name(), "rccl") && be->world > 1 && mode == TFK_EXCHANGE_SHARDED;
    } else if (!strcmp(v, "direct")) {
      c->algo_rs = c->algo_ag = TFK_ALGO_DIRECT;
      c->chosen_by = 1;
    } else {
      return bail(failx(-1, "TFK_DP_ALGO=%s (rccl | direct | auto)", v));
    }
  }
  void* st = nullptr;
  if (tfk_stream(e, &st)) return bail(-1);
  c->engine_stream = (hipStream_t)st;
  void* p = nullptr;
  size_t n = 0;
  if (tfk_reduce_region(e, &p, &n)) return bail(-1);
  c->grad = static_cast<float*>(p);
  c->reduce_floats = n;
  {
    hipPointerAttribute_t attr;
    hipError_t he = hipPointerGetAttributes(&attr, p);
    if (he != hipSuccess) return bail(failx((int)he, "hipPointerGetAttributes(engine state) failed: %s", hipGetErrorString(he)));
    c->device = attr.device;
    he = hipSetDevice(c->device);
    if (he != hipSuccess) return bail(failx((int)he, "hipSetDevice(%d) failed: %s", c->device, hipGetErrorString(he)));
  }
  if (tfk_param_region(e, &p, &n)) return bail(-1);
  c->param = static_cast<float*>(p);
  {
    void *m = nullptr, *v = nullptr;
    size_t nm = 0;
    if (tfk_moment_regions(e, &m, &v, &nm)) return bail(-1);
    c->adam_m = static_cast<float*>(m);
    c->adam_v = static_cast<float*>(v);
  }
  int nb = 0, mirrors = 0;
  if (tfk_num_buckets(e, &nb)) return bail(-1);
  c->L = nb - 3;
  for (int b = 0; b < nb; ++b) {
    size_t off = 0, cnt = 0;
    if (tfk_reduce_bucket(e, b, &off, &cnt)) return bail(-1);
    c->buckets.push_back({off, cnt});
  }
  c->num_params = c->buckets[nb - 1].first;
  c->vec_off = c->buckets[nb - 2].first;
  if (tfk_shadow_region(e, &p, &n, &mirrors)) return bail(-1);
  c->shadow = (mirrors && n && mode == TFK_EXCHANGE_SHARDED) ? p : nullptr;
  c->twin.resize(c->L + 1);
  bool twins = false;
  for (int l = 0; l <= c->L; ++l) {
    if (tfk_twin_region(e, l, &c->twin[l].ptr, &c->twin[l].bytes, &c->twin[l].rows)) return bail(-1);
    twins = twins || c->twin[l].bytes > 0;
  }
  if (const char* v = getenv("TFK_DP_GATHER")) {
    // (an arithmetic without owner-written twins -- exact fp32, mixed precision -- has nothing to gather in place of the
    // parameters / the shadow: the variable is a job-wide setting and is simply not applicable there)
    if (!strcmp(v, "planes")) c->gather_planes = twins && mode == TFK_EXCHANGE_SHARDED;
    else if (strcmp(v, "params")) return bail(failx(-1, "TFK_DP_GATHER=%s (params | planes)", v));
  }
  hipError_t he = hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking);
  if (he != hipSuccess) return bail(failx((int)he, "hipStreamCreate failed: %s", hipGetErrorString(he)));
  if (new_event(&c->ev_adam)) return bail(-1);
  if (tfk_set_bucket_callback(e, on_bucket, c) || tfk_set_layer_callback(e, mode == TFK_EXCHANGE_SHARDED ? on_layer : nullptr, c))
    return bail(-1);
  // staging memory is allocated HERE, never in the middle of a step (hipMalloc synchronises the device)
  if (mode == TFK_EXCHANGE_SHARDED && (c->wire_bf16 || c->algo_rs == TFK_ALGO_DIRECT) && ensure_stage(c)) return bail(-1);
  if (tune) {
    // COLLECTIVE (tfk_comm_create is): every rank times RCCL's own reduce-scatter / all-gather and the direct forms on scratch
    // memory of a span's size; the slowest rank's times decide, identically everywhere.  A failure here is a failure of the
    // communicator itself and is reported as one.
    size_t biggest = 0;
    for (int b = 0; b <= c->L; ++b) biggest = std::max(biggest, c->buckets[b].second);
    const size_t floats = std::max(biggest, std::min(c->min_floats, c->vec_off));
    if (tfk_comm_tune(c, floats, 5)) return bail(-1);
  }
  *out = c;
  return 0;
}

}  // namespace

extern "C" {

int tfk_comm_available(tfk_engine* e, int mode) {
  // everything tfk_comm_create can refuse WITHOUT talking to another rank: RCCL loadable with every symbol bound, a known
  // exchange mode, an engine whose regions and stream can be queried on a device this process can select
  if (!e) return failx(-1, "NULL argument");
  if (mode != TFK_EXCHANGE_SHARDED && mode != TFK_EXCHANGE_ALLREDUCE) return failx(-1, "unknown exchange mode %d", mode);
  Rccl* r = rccl();
  if (!r->error.empty()) return failx(-1, "%s", r->error.c_str());
  void* p = nullptr;
  size_t n = 0;
  int nb = 0;
  XCHK(tfk_stream(e, &p));
  XCHK(tfk_reduce_region(e, &p, &n));
  hipPointerAttribute_t attr;
  XHIP(hipPointerGetAttributes(&attr, p));
  XHIP(hipSetDevice(attr.device));
  XCHK(tfk_param_region(e, &p, &n));
  XCHK(tfk_num_buckets(e, &nb));
  if (nb < 3) return failx(-1, "engine announces %d buckets", nb);
  return 0;
}

int tfk_comm_unique_id(void* id, size_t capacity, size_t* size) {
  if (!id || capacity < sizeof(ncclUniqueId)) return failx(-1, "tfk_comm_unique_id needs %zu bytes", sizeof(ncclUniqueId));
  Rccl* r = rccl();
  if (!r->error.empty()) return failx(-1, "%s", r->error.c_str());
  ncclUniqueId uid;
  XNCCL(r->GetUniqueId(&uid));
  memcpy(id, &uid, sizeof(uid));
  if (size) *size = sizeof(uid);
  return 0;
}

int tfk_comm_create(tfk_engine* e, const void* id, size_t id_size, int rank, int world, int mode, size_t bucket_bytes,
                    tfk_comm** out) {
  if (!e || !id || !out) return failx(-1, "NULL argument");
  if (world < 1 || rank < 0 || rank >= world) return failx(-1, "rank %d of world %d", rank, world);
  if (id_size != sizeof(ncclUniqueId)) return failx(-1, "unique id of %zu bytes, expected %zu", id_size, sizeof(ncclUniqueId));
  Rccl* r = rccl();
  if (!r->error.empty()) return failx(-1, "%s", r->error.c_str());
  {  // the communicator binds to the CURRENT device: the engine's
    void* p = nullptr;
    size_t n = 0;
    XCHK(tfk_reduce_region(e, &p, &n));
    hipPointerAttribute_t attr;
    XHIP(hipPointerGetAttributes(&attr, p));
    XHIP(hipSetDevice(attr.device));
  }
  RcclBackend* be = new RcclBackend;
  be->rank = rank;
  be->world = world;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  const ncclResult_t rc = r->CommInitRank(&be->comm, world, uid, rank);
  if (rc != ncclSuccess) {
    be->comm = nullptr;
    delete be;
    return failx((int)rc, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, r->GetErrorString(rc));
  }
  return attach(e, be, mode, bucket_bytes, out);
}

int tfk_loopback_create(int world, tfk_loopback** out) {
  if (!out || world < 1 || world > kLoopMaxWorld) return failx(-1, "loopback group of %d ranks (1 .. %d)", world, kLoopMaxWorld);
  tfk_loopback* g = new tfk_loopback;
  g->world = world;
  g->slot.resize(world);
  hipError_t he = hipEventCreateWithFlags(&g->sum_done, hipEventDisableTiming);
  if (he != hipSuccess) {
    delete g;
    return failx((int)he, "hipEventCreate failed: %s", hipGetErrorString(he));
  }
  *out = g;
  return 0;
}
int tfk_loopback_destroy(tfk_loopback* g) {
  if (!g) return 0;
  if (g->scratch) (void)hipFree(g->scratch);
  if (g->sum_done) (void)hipEventDestroy(g->sum_done);
  delete g;
  return 0;
}
int tfk_comm_create_loopback(tfk_engine* e, tfk_loopback* group, int rank, int mode, size_t bucket_bytes, tfk_comm** out) {
  if (!e || !group || !out) return failx(-1, "NULL argument");
  if (rank < 0 || rank >= group->world) return failx(-1, "rank %d of a loopback group of %d", rank, group->world);
  LoopBackend* be = new LoopBackend;
  be->rank = rank;
  be->world = group->world;
  be->g = group;
  return attach(e, be, mode, bucket_bytes, out);
}

int tfk_comm_destroy(tfk_comm* c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  if (c->e) {
    (void)tfk_set_bucket_callback(c->e, nullptr, nullptr);
    (void)tfk_set_layer_callback(c->e, nullptr, nullptr);
  }
  if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
  for (Span& s : c->spans) {
    if (s.ready) (void)hipEventDestroy(s.ready);
    if (s.done) (void)hipEventDestroy(s.done);
  }
  for (hipEvent_t ev : c->gather_events) (void)hipEventDestroy(ev);
  if (c->ev_adam) (void)hipEventDestroy(c->ev_adam);
  if (c->stage) (void)hipFree(c->stage);
  for (hipEvent_t ev : c->timing_pool) (void)hipEventDestroy(ev);
  delete c->be;
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  delete c;
  return 0;
}

int tfk_comm_info(tfk_comm* c, int* rank, int* world, int* mode, int* gathers_shadow) {
  if (!c) return failx(-1, "comm is NULL");
  if (rank) *rank = c->be->rank;
  if (world) *world = c->be->world;
  if (mode) *mode = c->mode;
  if (gathers_shadow) *gathers_shadow = c->shadow ? 1 : c->gather_planes ? 2 : 0;
  return 0;
}

int tfk_comm_idle(tfk_comm* c) {
  if (!c) return failx(-1, "comm is NULL");
  XHIP(hipSetDevice(c->device));
  XCHK(raise_remembered(c));
  XCHK(tfk_zero_accumulators(c->e));
  // the order accumulate(TFK_LAST_MICROBATCH) announces them in: scalars + BN increments, the weight matrices from the
  // output layer down, the bias / beta gradients -- every rank must launch the same collectives in the same order
  XCHK(announce(c, c->L + 2));
  for (int b = 0; b <= c->L; ++b) XCHK(announce(c, b));
  XCHK(announce(c, c->L + 1));
  return 0;
}

// env TFK_DP_HOST_PHASES=1 (tools): host microseconds tfk_comm_apply spends per phase, averaged, printed when the comm goes
struct Phases {
  bool on = getenv("TFK_DP_HOST_PHASES") != nullptr;
  double sum[8] = {0};
  long calls = 0;
  std::chrono::steady_clock::time_point t;
  void start() { if (on) t = std::chrono::steady_clock::now(); }
  void mark(int k) {
    if (!on) return;
    const auto n = std::chrono::steady_clock::now();
    sum[k] += std::chrono::duration<double, std::micro>(n - t).count();
    t = n;
  }
  ~Phases() {
    if (on && calls)
      fprintf(stderr, "tfk_comm_apply host us/call: flush %.1f | head wait %.1f | apply_begin %.1f | adam launches %.1f | gathers %.1f | "
              "apply_end (waits for the loss) %.1f | bucket callbacks under backward %.1f   [%ld calls]\n", sum[0] / calls,
              sum[1] / calls, sum[2] / calls, sum[3] / calls, sum[4] / calls, sum[5] / calls, g_bucket_us / calls, calls);
  }
};
Phases g_phases;

// The exchange + optimiser step in two halves: ENQUEUE (everything up to the last parameter gather is on the streams; the host
// has not waited for anything) and END (the host waits for the step's loss).  Between the two the host is free -- the place
// for work that should run while the GPU is busy (the dispenser's prefetch of the next batch): round 4 ran that work BEFORE
// the tail collectives, Adam and the gathers were launched, so a rank with slow I/O held every peer in its collectives.
int tfk_comm_apply_enqueue(tfk_comm* c) {
  if (!c) return failx(-1, "comm is NULL");
  if (c->apply_enqueued) return failx(-1, "tfk_comm_apply_enqueue twice without tfk_comm_apply_end");
  XHIP(hipSetDevice(c->device));
  XCHK(raise_remembered(c));
  g_phases.start();
  // engine-stream time between the last backward kernel and the first optimiser kernel: the tail collectives, the waits for the
  // spans still in flight, tfk_apply_begin -- what the exchange leaves exposed in front of Adam
  Timed tail(c, PH_TAIL_EXPOSED, c->engine_stream);
  XCHK(flush_range(c, inline_tail()));
  g_phases.mark(0);
  const size_t head_off = c->buckets.back().first, head_n = c->buckets.back().second;
  for (size_t i = 0; i < c->num_spans; ++i) {
    Span& s = c->spans[i];
    if (s.off < head_off + head_n && s.off + s.n > head_off) XCHK(wait_span(c, s));
  }
  XCHK(drain(c));  // (gathers of the previous step that no forward pass has consumed: none in a training loop)
  g_phases.mark(1);
  XCHK(tfk_apply_begin(c->e));
  g_phases.mark(2);
  bool via_shadow = c->shadow != nullptr;
  if (via_shadow) {
    int direct = 0;
    XCHK(tfk_apply_writes_shadow(c->e, &direct));
    if (!direct) return failx(-1, "mixed-precision engine with an arena-mirroring shadow that the optimiser does not write");
  }
  const int W = c->be->world, R = c->be->rank;
  // every span is complete by now (the head wait above covered them all, see wait_span); what this rank updates -- its 1/W of
  // a reduce-scattered span, the whole of an all-reduced one -- goes to the optimiser in ascending order with neighbours
  // merged: the all-reduce mode is back to ONE Adam launch over the arena, the sharded mode to one per shard
  std::vector<std::pair<size_t, size_t>> sharded, mine;
  for (size_t i = 0; i < c->num_spans; ++i) {
    Span& s = c->spans[i];
    XCHK(wait_span(c, s));
    if (s.rs) {
      for (const Piece& p : pieces_of(c, s.off, s.n)) {
        const size_t per = p.n / W;
        mine.push_back({p.off + (size_t)R * per, per});
        const std::pair<size_t, size_t> piece(p.off, p.n);
        if (std::find(c->owned_pieces.begin(), c->owned_pieces.end(), piece) == c->owned_pieces.end()) c->owned_pieces.push_back(piece);
      }
      sharded.push_back({s.off, s.n});
    } else {
      mine.push_back({s.off, s.n});  // (spans beyond the parameter arena are clipped by the engine)
    }
  }
  std::sort(mine.begin(), mine.end());
  XCHK(tail.end());
  Timed adam(c, PH_ADAM, c->engine_stream);
  for (size_t i = 0; i < mine.size();) {
    size_t off = mine[i].first, n = mine[i].second, j = i + 1;
    while (j < mine.size() && mine[j].first == off + n) n += mine[j++].second;
    XCHK(tfk_apply_span(c->e, off, n));
    i = j;
  }
  XCHK(adam.end());
  g_phases.mark(3);
  bool planes_any = false;
  if (!sharded.empty()) {
    if (c->gather_planes) {
      int direct = 0;
      XCHK(tfk_apply_writes_shadow(c->e, &direct));
      if (!direct) return failx(-1, "TFK_DP_GATHER=planes and an optimiser step that does not write the three-plane twins");
    } else if (c->masters_stale && !via_shadow) {
      return failx(-1, "sharded fp32 masters and a step that does not write the shadow");
    }
    std::sort(sharded.begin(), sharded.end());  // lowest offsets (layer 0) first: the order the next forward pass reads in
    c->gathers_used = 0;
    size_t first_on_comm = 0;
    bool derived_current = true;  // what the contractions read (fp32 emulated: the three-plane twins) follows every gather
    if (inline_tail()) {
      // the span the next forward pass reads FIRST is gathered on the engine stream itself, right behind the optimiser: that
      // pass could not start before it anyway, and the hop to the comm stream and back is saved; the others follow on the
      // comm stream, under the first layers
      Timed exposed(c, PH_GATHER_EXPOSED, c->engine_stream);  // (nothing runs beside it: the forward pass waits for exactly this)
      XCHK(gather_pieces(c, sharded[0], c->engine_stream, via_shadow, &derived_current, &planes_any));
      XCHK(exposed.end());
      c->cur_ag += 1;
      first_on_comm = 1;
    }
    if (first_on_comm < sharded.size()) {
      XHIP(hipEventRecord(c->ev_adam, c->engine_stream));
      XHIP(hipStreamWaitEvent(c->comm_stream, c->ev_adam, 0));
    }
    for (size_t k = first_on_comm; k < sharded.size(); ++k) {
      const auto& s = sharded[k];
      if (c->gathers_used == c->gather_events.size()) {
        hipEvent_t ev;
        XCHK(new_event(&ev));
        c->gather_events.push_back(ev);
      }
      hipEvent_t done = c->gather_events[c->gathers_used++];
      XCHK(gather_pieces(c, s, c->comm_stream, via_shadow, &derived_current, &planes_any));
      XHIP(hipEventRecord(done, c->comm_stream));
      Gather g;
      g.off = s.first; g.n = s.second; g.done = done;
      c->pending.push_back(g);
      c->cur_ag += 1;
    }
    if (via_shadow) {
      c->masters_stale = true;
      for (const auto& s : sharded)
        if (std::find(c->shard_spans.begin(), c->shard_spans.end(), s) == c->shard_spans.end()) c->shard_spans.push_back(s);
    } else {
      if (planes_any) c->masters_stale = true;  // (gather_pieces noted which matrices)
      if (!derived_current) XCHK(tfk_params_touched(c->e));  // parameters outside this rank's spans change behind the optimiser's back
    }
  }
  c->last_spans.clear();
  c->last_span_rs.clear();
  for (size_t i = 0; i < c->num_spans; ++i) {
    c->last_spans.push_back({c->spans[i].off, c->spans[i].n});
    c->last_span_rs.push_back(c->spans[i].rs ? 1 : 0);
  }
  c->last_rs = c->cur_rs; c->last_ag = c->cur_ag; c->last_ar = c->cur_ar;
  c->cur_rs = c->cur_ag = c->cur_ar = 0;
  c->num_spans = 0;
  c->waited_upto = 0;
  g_phases.mark(4);
  c->apply_enqueued = true;
  c->apply_sharded = !sharded.empty();
  c->apply_via_shadow = via_shadow;
  c->apply_planes = planes_any;
  return 0;
}
int tfk_comm_apply_end(tfk_comm* c, float* average_loss) {
  if (!c) return failx(-1, "comm is NULL");
  if (!c->apply_enqueued) return failx(-1, "tfk_comm_apply_end without tfk_comm_apply_enqueue");
  c->apply_enqueued = false;
  XHIP(hipSetDevice(c->device));
  XCHK(tfk_apply_end(c->e, average_loss));
  g_phases.mark(5);
  g_phases.calls += 1;
  if (c->timing) c->timed_steps += 1;
  if (c->verify_left > 0 && c->apply_sharded) {
    c->verify_left -= 1;
    XCHK(verify_replicas(c, c->apply_via_shadow, c->apply_planes));
  }
  return 0;
}
int tfk_comm_apply(tfk_comm* c, float* average_loss) {
  XCHK(tfk_comm_apply_enqueue(c));
  return tfk_comm_apply_end(c, average_loss);
}

int tfk_comm_finish_reduce(tfk_comm* c) {
  if (!c) return failx(-1, "comm is NULL");
  XHIP(hipSetDevice(c->device));
  XCHK(raise_remembered(c));
  XCHK(flush_range(c, inline_tail()));
  for (size_t i = 0; i < c->num_spans; ++i) XCHK(wait_span(c, c->spans[i]));
  return 0;
}

int tfk_comm_eval_finish(tfk_comm* c, float* average_loss) {
  if (!c) return failx(-1, "comm is NULL");
  XHIP(hipSetDevice(c->device));
  XCHK(raise_remembered(c));
  const size_t off = c->buckets.back().first, n = c->buckets.back().second;
  if (c->num_spans || c->have_range || c->have_hold || c->have_head) return failx(-1, "tfk_comm_eval_finish in the middle of a training step");
  XCHK(launch_range(c, off, off + n, inline_tail()));
  XCHK(wait_span(c, c->spans[0]));
  c->num_spans = 0;
  c->waited_upto = 0;
  c->cur_ar = 0;
  return tfk_eval_finish(c->e, average_loss);
}

int tfk_comm_drain(tfk_comm* c) {
  if (!c) return failx(-1, "comm is NULL");
  XHIP(hipSetDevice(c->device));
  return drain(c);
}

int tfk_comm_masters_stale(tfk_comm* c, int* stale) {
  if (!c || !stale) return failx(-1, "NULL argument");
  *stale = c->masters_stale ? 1 : 0;
  return 0;
}

int tfk_comm_gather_masters(tfk_comm* c) {
  if (!c) return failx(-1, "comm is NULL");
  XHIP(hipSetDevice(c->device));
  XCHK(raise_remembered(c));
  XCHK(drain(c));
  if (!c->masters_stale) return 0;
  std::sort(c->shard_spans.begin(), c->shard_spans.end());
  XHIP(hipEventRecord(c->ev_adam, c->engine_stream));
  XHIP(hipStreamWaitEvent(c->comm_stream, c->ev_adam, 0));
  for (const auto& s : c->shard_spans)
    XCHK(gather_span(c, c->param + s.first, s.second / c->be->world * sizeof(float), c->comm_stream));
  XHIP(hipEventRecord(c->ev_adam, c->comm_stream));
  XHIP(hipStreamWaitEvent(c->engine_stream, c->ev_adam, 0));
  c->masters_stale = false;
  return 0;
}

namespace {
// Everything a rank holds of the optimiser's state becomes whole again: the fp32 masters left with their owners (if any) and
// Adam's moments of every piece sharded so far.  COLLECTIVE.  After it any assignment of shards to ranks is as good as any other.
int rehome(tfk_comm* c) {
  XCHK(tfk_comm_gather_masters(c));
  c->shard_spans.clear();
  if (c->owned_pieces.empty()) return 0;
  std::sort(c->owned_pieces.begin(), c->owned_pieces.end());
  XHIP(hipEventRecord(c->ev_adam, c->engine_stream));
  XHIP(hipStreamWaitEvent(c->comm_stream, c->ev_adam, 0));
  const size_t W = (size_t)c->be->world;
  for (const auto& p : c->owned_pieces) {
    XCHK(gather_span(c, c->adam_m + p.first, p.second / W * sizeof(float), c->comm_stream, false));
    XCHK(gather_span(c, c->adam_v + p.first, p.second / W * sizeof(float), c->comm_stream, false));
  }
  XHIP(hipEventRecord(c->ev_adam, c->comm_stream));
  XHIP(hipStreamWaitEvent(c->engine_stream, c->ev_adam, 0));
  c->owned_pieces.clear();
  return 0;
}
}  // namespace

int tfk_comm_set_exchange(tfk_comm* c, int algo, int wire) {
  if (!c) return failx(-1, "comm is NULL");
  XHIP(hipSetDevice(c->device));
  XCHK(raise_remembered(c));
  if (c->num_spans || c->have_range || c->have_hold || c->have_head || c->apply_enqueued) return failx(-1, "tfk_comm_set_exchange in the middle of a step");
  if (algo != -1 && algo != TFK_ALGO_RCCL && algo != TFK_ALGO_DIRECT) return failx(-1, "exchange algorithm %d", algo);
  if (wire != -1 && wire != TFK_WIRE_FP32 && wire != TFK_WIRE_BF16) return failx(-1, "wire format %d", wire);
  if (algo != -1) {
    c->algo_rs = c->algo_ag = algo;
    c->chosen_by = 3;
  }
  if (wire != -1) c->wire_bf16 = wire == TFK_WIRE_BF16;
  if (c->mode == TFK_EXCHANGE_SHARDED && (c->wire_bf16 || c->algo_rs == TFK_ALGO_DIRECT)) XCHK(ensure_stage(c));
  return 0;
}

int tfk_comm_set_gather(tfk_comm* c, int planes) {
  if (!c) return failx(-1, "comm is NULL");
  XHIP(hipSetDevice(c->device));
  XCHK(raise_remembered(c));
  if (c->num_spans || c->have_range || c->have_hold || c->have_head || c->apply_enqueued) return failx(-1, "tfk_comm_set_gather in the middle of a step");
  if (planes) {
    bool twins = false;
    for (const tfk_comm::TwinOf& t : c->twin) twins = twins || t.bytes > 0;
    if (!twins || c->mode != TFK_EXCHANGE_SHARDED)
      return failx(-1, "nothing to gather in place of the parameters: owner-written three-plane twins exist under the emulated fp32 "
                       "arithmetic and the sharded exchange only");
    XCHK(rehome(c));  // (the sharding unit changes with the switch: nothing may be left with a previous owner)
    c->gather_planes = true;
    return 0;
  }
  // back to fp32 gathers: the masters the plane gathers left with their owners, and Adam's moments, come home first
  // (COLLECTIVE, like this call)
  XCHK(rehome(c));
  c->gather_planes = false;
  return 0;
}

int tfk_comm_set_bucket_bytes(tfk_comm* c, size_t bucket_bytes) {
  if (!c) return failx(-1, "comm is NULL");
  if (c->num_spans || c->have_range || c->have_hold || c->have_head || c->apply_enqueued) return failx(-1, "tfk_comm_set_bucket_bytes in the middle of a step");
  // another span cut is another assignment of shards to ranks: what lives with a shard's owner only -- fp32 masters (mixed
  // precision, plane gathers) and Adam's moments (always) -- comes home first, COLLECTIVE like this call, or the new owner would
  // update stale values
  XCHK(rehome(c));
  if (bucket_bytes == 0) bucket_bytes = kDefaultBucketBytes;
  c->min_floats = std::max<size_t>(1, bucket_bytes / 4);
  return 0;
}

int tfk_comm_get_exchange(tfk_comm* c, int* algo_reduce_scatter, int* algo_all_gather, int* wire, int* chosen_by, double* tune_us) {
  if (!c) return failx(-1, "comm is NULL");
  if (algo_reduce_scatter) *algo_reduce_scatter = c->algo_rs;
  if (algo_all_gather) *algo_all_gather = c->algo_ag;
  if (wire) *wire = c->wire_bf16 ? TFK_WIRE_BF16 : TFK_WIRE_FP32;
  if (chosen_by) *chosen_by = c->chosen_by;
  if (tune_us) memcpy(tune_us, c->tune_us, sizeof(c->tune_us));
  return 0;
}

int tfk_comm_tune(tfk_comm* c, size_t floats, int iters) {
  if (!c) return failx(-1, "comm is NULL");
  XHIP(hipSetDevice(c->device));
  if (c->num_spans || c->have_range || c->have_hold || c->have_head || c->apply_enqueued) return failx(-1, "tfk_comm_tune in the middle of a step");
  const int W = c->be->world, R = c->be->rank;
  const size_t unit = 4 * (size_t)W;
  floats = std::max(unit, floats / unit * unit);
  iters = std::max(1, iters);
  const size_t per = floats / W;
  float *buf = nullptr, *recv = nullptr;
  hipEvent_t a = nullptr, b = nullptr;
  struct Scratch {  // (freed on every way out)
    float **buf, **recv;
    hipEvent_t *a, *b;
    ~Scratch() {
      if (*buf) (void)hipFree(*buf);
      if (*recv) (void)hipFree(*recv);
      if (*a) (void)hipEventDestroy(*a);
      if (*b) (void)hipEventDestroy(*b);
    }
  } scratch = {&buf, &recv, &a, &b};
  XHIP(hipMalloc((void**)&buf, floats * sizeof(float)));
  XHIP(hipMalloc((void**)&recv, floats * sizeof(float)));
  XHIP(hipMemsetAsync(buf, 0, floats * sizeof(float), c->comm_stream));
  XHIP(hipEventCreate(&a));
  XHIP(hipEventCreate(&b));
  hipStream_t st = c->comm_stream;
  auto run = [&](int what) -> int {  // 0 / 1: reduce-scatter rccl / direct, 2 / 3: all-gather rccl / direct
    if (what == 0) return c->be->reduce_scatter(buf, per, st);
    if (what == 1) {
      XCHK(c->be->exchange_shards(buf, recv, per * sizeof(float), st));
      const unsigned sb = (unsigned)std::min<size_t>((per / 4 + 255) / 256, 1 << 13);
      hipLaunchKernelGGL(direct_sum_kernel, dim3(sb ? sb : 1), dim3(256), 0, st, buf + (size_t)R * per, recv, R, W, per);
      XHIP(hipGetLastError());
      return 0;
    }
    if (what == 2) return c->be->all_gather(buf, per * sizeof(float), st);
    return c->be->all_gather_direct(buf, per * sizeof(float), st);
  };
  for (int what = 0; what < 4; ++what) {
    XCHK(run(what));  // (warm-up: connection set-up, first-use kernels)
    XHIP(hipStreamSynchronize(st));
    XHIP(hipEventRecord(a, st));
    for (int i = 0; i < iters; ++i) XCHK(run(what));
    XHIP(hipEventRecord(b, st));
    XHIP(hipStreamSynchronize(st));
    float ms = 0.f;
    XHIP(hipEventElapsedTime(&ms, a, b));
    // the slowest rank's time counts, and every rank must reach the same decision: MAX over the ranks, in integer nanoseconds
    unsigned long long v[2] = {(unsigned long long)(1e6 * ms / iters), 0};
    XCHK(c->be->min_max(v, st));
    c->tune_us[what] = 1e-3 * (double)v[1];
  }
  // RCCL's own collective unless the direct form is clearly faster (3 %: a tie goes to the vendor's code path)
  c->algo_rs = c->tune_us[1] < 0.97 * c->tune_us[0] ? TFK_ALGO_DIRECT : TFK_ALGO_RCCL;
  c->algo_ag = c->tune_us[3] < 0.97 * c->tune_us[2] ? TFK_ALGO_DIRECT : TFK_ALGO_RCCL;
  c->chosen_by = 2;
  if (c->mode == TFK_EXCHANGE_SHARDED && (c->wire_bf16 || c->algo_rs == TFK_ALGO_DIRECT)) XCHK(ensure_stage(c));
  if (R == 0 && getenv("TFK_DP_QUIET") == nullptr)
    fprintf(stderr, "tfkaldi_amd exchange tuned on %zu floats, %d ranks (us, slowest rank): reduce-scatter rccl %.1f / direct %.1f -> %s; "
            "all-gather rccl %.1f / direct %.1f -> %s\n", floats, W, c->tune_us[0], c->tune_us[1],
            c->algo_rs == TFK_ALGO_DIRECT ? "direct" : "rccl", c->tune_us[2], c->tune_us[3],
            c->algo_ag == TFK_ALGO_DIRECT ? "direct" : "rccl");
  return 0;
}

int tfk_comm_timing(tfk_comm* c, int on) {
  if (!c) return failx(-1, "comm is NULL");
  if (c->num_spans || c->have_range || c->have_hold || c->have_head || c->apply_enqueued) return failx(-1, "tfk_comm_timing in the middle of a step");
  c->timing = on != 0;
  c->timed.clear();
  c->timing_used = 0;
  c->timed_steps = 0;
  return 0;
}

int tfk_comm_timing_read(tfk_comm* c, double* ms_per_step, int capacity, long* steps) {
  if (!c || !ms_per_step) return failx(-1, "NULL argument");
  XHIP(hipSetDevice(c->device));
  XHIP(hipStreamSynchronize(c->engine_stream));
  XHIP(hipStreamSynchronize(c->comm_stream));
  double sum[PH_COUNT] = {0};
  for (const tfk_comm::TimedPair& t : c->timed) {
    float ms = 0.f;
    XHIP(hipEventElapsedTime(&ms, t.a, t.b));
    sum[t.phase] += ms;
  }
  const double per = c->timed_steps > 0 ? 1.0 / (double)c->timed_steps : 0.0;
  for (int k = 0; k < capacity; ++k) ms_per_step[k] = k < PH_COUNT ? sum[k] * per : 0.0;
  if (steps) *steps = c->timed_steps;
  return 0;
}

int tfk_comm_last_step(tfk_comm* c, int* reduce_scatters, int* all_gathers, int* all_reduces, size_t* spans, int capacity,
                       int* num_spans) {
  if (!c) return failx(-1, "comm is NULL");
  if (reduce_scatters) *reduce_scatters = c->last_rs;
  if (all_gathers) *all_gathers = c->last_ag;
  if (all_reduces) *all_reduces = c->last_ar;
  if (num_spans) *num_spans = (int)c->last_spans.size();
  if (spans)
    for (int i = 0; i < capacity && i < (int)c->last_spans.size(); ++i) {
      spans[3 * i] = c->last_spans[i].first;
      spans[3 * i + 1] = c->last_spans[i].second;
      spans[3 * i + 2] = (size_t)c->last_span_rs[i];
    }
  return 0;
}

const char* tfk_comm_backend(tfk_comm* c) { return c ? c->be->name() : "?"; }

}  // extern "C"
