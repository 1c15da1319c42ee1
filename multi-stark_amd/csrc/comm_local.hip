// In-process transport for ms_prove_sharded: the ranks are THREADS of one process, each with its own ms_ctx (on one device
// or on several), and the two exchanges of the ms_comm table (include/mstark.h) are device-to-device copies that the
// receiving rank pulls out of the sender's buffers, ordered by HIP events between the ranks' streams. Two uses:
//   * a host that drives all GPUs of a node from one process - the shape of the reference itself, whose prover is one process
//     with a thread pool (Cargo.toml:45) - needs neither RCCL nor a rendezvous: peer copies over xGMI (hipMemcpy between
//     devices with peer access) do the row-range exchange;
//   * the pool's test boxes have ONE GPU and allow few processes on it, and RCCL refuses two ranks on one device: with
//     thread ranks the joint prover runs at world 8 (BASELINE config 3 as specified) inside one test process, through the same
//     callback table, the same stream-ordered mode and the same non-blocking column exchange the RCCL transport offers.
// Protocol of one collective (every rank calls the same sequence of collectives):
//   1. publish what I offer (pointers, strides, sizes) and record `ready` on my stream (behind the caller's stream);
//   2. host barrier; every rank checks that all ranks entered the same collective with the same sizes;
//   3. my stream waits for every peer's `ready`, then pulls its blocks out of the peers' send buffers; record `done`;
//   4. host barrier; my stream waits for every peer's `done` (so a later overwrite of my send buffer is ordered behind
//      the peers' reads); the blocking forms then hand over to the caller's stream (or wait on the host).
// A rank that fails, or never arrives, does not hang the others: barriers time out (MSAMD_LOCAL_TIMEOUT_S, default 120) and
// ms_comm_local_group_abort wakes every waiter with an error.
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mstark.h"
#include "msamd.h"

namespace msamd {
void set_last_error(const char* what);
Ctx* ctx_of(ms_ctx* c);
void ctx_retain(ms_ctx* c);
void ctx_release(ms_ctx* c);
}  // namespace msamd
using namespace msamd;

namespace {
enum OpKind : int { OP_NONE = 0, OP_A2A, OP_COLS, OP_GATHER, OP_SCATTER };
const char* op_name(int k) {
  return k == OP_A2A ? "all_to_all" : k == OP_COLS ? "all_to_all_cols" : k == OP_GATHER ? "all_gather" : k == OP_SCATTER ? "scatter_cols" : "none";
}
struct Offer {
  int kind = OP_NONE;
  const uint8_t* send = nullptr;
  size_t send_peer_stride = 0, send_col_stride = 0;
  size_t ncols = 0, seg = 0;  // a2a: ncols = 1, seg = bytes per peer; gather: seg = bytes
  int root = -1;              // scatter: the rank that hands its matrix out
};
}  // namespace

struct ms_comm_local_group {
  int world = 1;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  bool aborted = false;
  std::string abort_why;
  double timeout_s = 120;
  int members = 0, refs = 1;  // handles created on this group; the group object lives until the last of them is gone
  std::vector<Offer> offers;
  // One pair of events per rank, owned by the GROUP and destroyed with it: a rank that has returned from the last collective
  // may destroy its handle while a slower peer is still ordering its stream behind that rank's `done` event.
  std::vector<hipEvent_t> ready, done;
  std::vector<int> devices;
  std::vector<char> taken;
  ~ms_comm_local_group() {
    for (size_t k = 0; k < ready.size(); k++) {
      if (devices[k] >= 0) (void)hipSetDevice(devices[k]);
      if (ready[k]) (void)hipEventDestroy(ready[k]);
      if (done[k]) (void)hipEventDestroy(done[k]);
    }
  }

  void abort(const std::string& why) {
    std::lock_guard<std::mutex> lk(mu);
    if (!aborted) {
      aborted = true;
      abort_why = why;
    }
    cv.notify_all();
  }
  // all ranks meet here; throws on every rank when one has failed or does not arrive in time
  void barrier(int rank, const char* what) {
    std::unique_lock<std::mutex> lk(mu);
    if (aborted) throw std::runtime_error("local transport aborted: " + abort_why);
    const uint64_t gen = generation;
    if (++arrived == world) {
      arrived = 0;
      generation++;
      cv.notify_all();
      return;
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s);
    while (generation == gen && !aborted) {
      if (cv.wait_until(lk, deadline) == std::cv_status::timeout && generation == gen && !aborted) {
        aborted = true;
        abort_why = std::string("rank ") + std::to_string(rank) + " waited " + std::to_string((int)timeout_s) + " s in " + what + " for " +
                    std::to_string(world - arrived) + " rank(s) that never arrived";
        cv.notify_all();
      }
    }
    if (generation == gen) throw std::runtime_error("local transport aborted: " + abort_why);
  }
};

struct ms_comm_local {
  ms_comm_local_group* g = nullptr;
  ms_ctx* owner = nullptr;
  Ctx* ctx = nullptr;
  hipStream_t stream = nullptr;
  int rank = 0;
  uint64_t bytes_moved = 0;
  ms_comm table;
  hipStream_t peer = nullptr;  // stream-ordered mode (ms_comm.set_stream_ordered): the caller's stream
  hipEvent_t ev_in = nullptr, ev_out = nullptr;

  void begin() {
    if (!peer) return;
    HIP_CHECK(hipEventRecord(ev_in, peer));
    HIP_CHECK(hipStreamWaitEvent(stream, ev_in, 0));
  }
  void complete() {
    if (peer) {
      HIP_CHECK(hipEventRecord(ev_out, stream));
      HIP_CHECK(hipStreamWaitEvent(peer, ev_out, 0));
    } else {
      HIP_CHECK(hipStreamSynchronize(stream));
    }
  }
  // steps 1-4 of the protocol; `recv*` describe where block k (from rank k) lands on this rank
  void collective(const Offer& mine, uint8_t* recv, size_t recv_peer_stride, size_t recv_col_stride, const char* what, bool skip_self = false) {
    const int N = g->world;
    bytes_moved += mine.seg * mine.ncols * (size_t)(skip_self ? N - 1 : N);
    try {
      g->offers[rank] = mine;
      HIP_CHECK(hipEventRecord(g->ready[rank], stream));
      g->barrier(rank, what);
      for (int k = 0; k < N; k++) {
        const Offer& o = g->offers[k];
        if (o.kind != mine.kind || o.seg != mine.seg || o.ncols != mine.ncols || o.root != mine.root)
          throw std::runtime_error(std::string("local transport: rank ") + std::to_string(rank) + " entered " + op_name(mine.kind) + "(" +
                                   std::to_string(mine.ncols) + " x " + std::to_string(mine.seg) + " B) while rank " + std::to_string(k) +
                                   " entered " + op_name(o.kind) + "(" + std::to_string(o.ncols) + " x " + std::to_string(o.seg) + " B)");
      }
      if (mine.kind == OP_SCATTER) {
        // only the root's offer carries data: every other rank pulls its row range out of the root's matrix
        if (rank != mine.root && mine.seg && mine.ncols) {
          const Offer& o = g->offers[mine.root];
          HIP_CHECK(hipStreamWaitEvent(stream, g->ready[mine.root], 0));
          HIP_CHECK(hipMemcpy2DAsync(recv, recv_col_stride, o.send + (size_t)rank * o.send_peer_stride, o.send_col_stride, mine.seg, mine.ncols,
                                     hipMemcpyDeviceToDevice, stream));
        }
      } else if (mine.seg && mine.ncols) {
        for (int k = 0; k < N; k++) {
          const Offer& o = g->offers[k];
          if (k == rank && skip_self) continue;
          if (k != rank) HIP_CHECK(hipStreamWaitEvent(stream, g->ready[k], 0));
          const uint8_t* src = mine.kind == OP_GATHER ? o.send : o.send + (size_t)rank * o.send_peer_stride;
          uint8_t* dst = recv + (size_t)k * recv_peer_stride;
          if (mine.ncols == 1)
            HIP_CHECK(hipMemcpyAsync(dst, src, mine.seg, hipMemcpyDeviceToDevice, stream));
          else
            HIP_CHECK(hipMemcpy2DAsync(dst, recv_col_stride, src, o.send_col_stride, mine.seg, mine.ncols, hipMemcpyDeviceToDevice, stream));
        }
      }
      HIP_CHECK(hipEventRecord(g->done[rank], stream));
      g->barrier(rank, what);
      for (int k = 0; k < N; k++)
        if (k != rank) HIP_CHECK(hipStreamWaitEvent(stream, g->done[k], 0));
    } catch (const std::exception& e) {
      g->abort(e.what());  // the peers are (or will be) waiting for this rank: wake them with the reason
      throw;
    }
  }
};

namespace {
template <class F>
int32_t guarded(ms_comm_local* c, F f) {
  try {
    HIP_CHECK(hipSetDevice(c->ctx->device));
    f();
    return 0;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return -1;
  }
}
Offer a2a_offer(const void* send, size_t send_stride, size_t per_peer) {
  Offer o;
  o.kind = OP_A2A;
  o.send = (const uint8_t*)send;
  o.send_peer_stride = send_stride;
  o.ncols = 1;
  o.seg = per_peer;
  return o;
}
int32_t cb_all_to_all(void* user, const void* send, void* recv, size_t per_peer) {
  ms_comm_local* c = (ms_comm_local*)user;
  return guarded(c, [&] {
    c->begin();
    c->collective(a2a_offer(send, per_peer, per_peer), (uint8_t*)recv, per_peer, 0, "all_to_all");
    c->complete();
  });
}
int32_t cb_all_gather(void* user, const void* send, void* recv, size_t bytes) {
  ms_comm_local* c = (ms_comm_local*)user;
  return guarded(c, [&] {
    Offer o;
    o.kind = OP_GATHER;
    o.send = (const uint8_t*)send;
    o.ncols = 1;
    o.seg = bytes;
    c->begin();
    c->collective(o, (uint8_t*)recv, bytes, 0, "all_gather");
    c->complete();
  });
}
int32_t cb_start(void* user, const void* send, size_t send_stride, void* recv, size_t recv_stride, size_t per_peer) {
  ms_comm_local* c = (ms_comm_local*)user;
  return guarded(c, [&] {
    c->begin();
    c->collective(a2a_offer(send, send_stride, per_peer), (uint8_t*)recv, recv_stride, 0, "all_to_all_start");
  });
}
int32_t cb_cols_start(void* user, const void* send, size_t sps, size_t scs, void* recv, size_t rps, size_t rcs, size_t ncols, size_t seg) {
  ms_comm_local* c = (ms_comm_local*)user;
  return guarded(c, [&] {
    Offer o;
    o.kind = OP_COLS;
    o.send = (const uint8_t*)send;
    o.send_peer_stride = sps;
    o.send_col_stride = scs;
    o.ncols = ncols;
    o.seg = seg;
    c->begin();
    c->collective(o, (uint8_t*)recv, rps, rcs, "all_to_all_cols_start");
  });
}
int32_t cb_cols_start2(void* user, const void* send, size_t sps, size_t scs, void* recv, size_t rps, size_t rcs, size_t ncols, size_t seg,
                       uint32_t flags) {
  ms_comm_local* c = (ms_comm_local*)user;
  return guarded(c, [&] {
    Offer o;
    o.kind = OP_COLS;
    o.send = (const uint8_t*)send;
    o.send_peer_stride = sps;
    o.send_col_stride = scs;
    o.ncols = ncols;
    o.seg = seg;
    c->begin();
    c->collective(o, (uint8_t*)recv, rps, rcs, "all_to_all_cols_start", (flags & MS_COMM_SKIP_SELF) != 0);
  });
}
int32_t cb_scatter(void* user, int32_t root, const void* send, size_t sps, size_t scs, void* recv, size_t rcs, size_t ncols, size_t seg) {
  ms_comm_local* c = (ms_comm_local*)user;
  return guarded(c, [&] {
    if (root < 0 || root >= c->g->world) throw std::runtime_error("scatter_cols_start: root out of range");
    Offer o;
    o.kind = OP_SCATTER;
    o.root = root;
    o.send = (const uint8_t*)send;
    o.send_peer_stride = sps;
    o.send_col_stride = scs;
    o.ncols = ncols;
    o.seg = seg;
    c->begin();
    c->collective(o, (uint8_t*)recv, 0, rcs, "scatter_cols_start", true);
  });
}
int32_t cb_wait(void* user) {
  ms_comm_local* c = (ms_comm_local*)user;
  return guarded(c, [&] { c->complete(); });
}
void cb_abort(void* user, const char* why) {
  ms_comm_local* c = static_cast<ms_comm_local*>(user);
  try {
    c->g->abort(std::string("rank ") + std::to_string(c->rank) + " left the proof: " + (why ? why : "error"));
  } catch (...) {
  }
}
int32_t cb_set_stream_ordered(void* user, void* hip_stream) {
  ms_comm_local* c = (ms_comm_local*)user;
  return guarded(c, [&] {
    if (!hip_stream && c->peer) HIP_CHECK(hipStreamSynchronize(c->stream));
    if (hip_stream && !c->ev_in) {
      HIP_CHECK(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
      HIP_CHECK(hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming));
    }
    c->peer = (hipStream_t)hip_stream;
  });
}
void group_unref(ms_comm_local_group* g) {
  bool last;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    last = --g->refs == 0;
  }
  if (last) delete g;
}
}  // namespace

extern "C" {

int32_t ms_comm_local_group_create(int32_t world, ms_comm_local_group** out) {
  *out = nullptr;
  try {
    if (world < 1 || world > 64) throw std::runtime_error("ms_comm_local_group_create: world out of range");
    ms_comm_local_group* g = new ms_comm_local_group();
    g->world = world;
    g->offers.resize(world);
    g->ready.assign(world, nullptr);
    g->done.assign(world, nullptr);
    g->devices.assign(world, -1);
    g->taken.assign(world, 0);
    if (const char* e = getenv("MSAMD_LOCAL_TIMEOUT_S")) g->timeout_s = std::max(1.0, atof(e));
    *out = g;
    return MS_OK;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return MS_ERR;
  }
}

void ms_comm_local_group_abort(ms_comm_local_group* g) {
  if (g) g->abort("aborted by the host (ms_comm_local_group_abort)");
}

void ms_comm_local_group_destroy(ms_comm_local_group* g) {
  if (g) group_unref(g);
}

int32_t ms_comm_local_create(ms_comm_local_group* g, ms_ctx* ctx, int32_t rank, ms_comm_local** out) {
  *out = nullptr;
  ms_comm_local* c = nullptr;
  try {
    if (!g) throw std::runtime_error("ms_comm_local_create: null group");
    if (rank < 0 || rank >= g->world) throw std::runtime_error("ms_comm_local_create: rank out of range");
    c = new ms_comm_local();
    c->ctx = ctx_of(ctx);
    c->g = g;
    c->rank = rank;
    HIP_CHECK(hipSetDevice(c->ctx->device));
    HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    {
      std::lock_guard<std::mutex> lk(g->mu);
      if (g->taken[rank]) throw std::runtime_error("ms_comm_local_create: this rank of the group is taken");
      if (g->ready[rank] && g->devices[rank] != c->ctx->device) {  // the rank moves to another device: fresh events
        (void)hipSetDevice(g->devices[rank]);
        (void)hipEventDestroy(g->ready[rank]);
        (void)hipEventDestroy(g->done[rank]);
        g->ready[rank] = g->done[rank] = nullptr;
        (void)hipSetDevice(c->ctx->device);
      }
      if (!g->ready[rank]) {
        HIP_CHECK(hipEventCreateWithFlags(&g->ready[rank], hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&g->done[rank], hipEventDisableTiming));
      }
      g->taken[rank] = 1;
      g->devices[rank] = c->ctx->device;
      g->members++;
      g->refs++;
      // ranks on different devices read each other's buffers directly (xGMI): peer access both ways, best effort - without
      // it the copies are staged by the runtime
      for (int k = 0; k < g->world; k++) {
        const int d = g->devices[k];
        if (d < 0 || d == c->ctx->device) continue;
        if (hipDeviceEnablePeerAccess(d, 0) != hipSuccess) (void)hipGetLastError();
        if (hipSetDevice(d) == hipSuccess && hipDeviceEnablePeerAccess(c->ctx->device, 0) != hipSuccess) (void)hipGetLastError();
        (void)hipSetDevice(c->ctx->device);
      }
    }
    memset(&c->table, 0, sizeof(c->table));
    c->table.size = (uint32_t)sizeof(ms_comm);
    c->table.abort = cb_abort;
    c->table.rank = rank;
    c->table.world = g->world;
    c->table.user = c;
    c->table.all_to_all = cb_all_to_all;
    c->table.all_gather = cb_all_gather;
    c->table.all_to_all_start = cb_start;
    c->table.all_to_all_wait = cb_wait;
    c->table.all_to_all_cols_start = cb_cols_start;
    c->table.set_stream_ordered = cb_set_stream_ordered;
    c->table.all_to_all_cols_start2 = cb_cols_start2;
    c->table.scatter_cols_start = cb_scatter;
    c->owner = ctx;
    ctx_retain(ctx);
    *out = c;
    return MS_OK;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    if (c) {
      if (c->stream) (void)hipStreamDestroy(c->stream);
      delete c;
    }
    return MS_ERR;
  }
}

const ms_comm* ms_comm_local_table(ms_comm_local* c) { return c ? &c->table : nullptr; }
uint64_t ms_comm_local_bytes_moved(ms_comm_local* c) { return c ? c->bytes_moved : 0; }

void ms_comm_local_destroy(ms_comm_local* c) {
  if (!c) return;
  (void)hipSetDevice(c->ctx->device);
  (void)hipStreamSynchronize(c->stream);
  ms_comm_local_group* g = c->g;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->taken[c->rank] = 0;  // (the rank's events stay with the group: a peer may still be waiting on them)
    g->members--;
  }
  if (c->ev_in) (void)hipEventDestroy(c->ev_in);
  if (c->ev_out) (void)hipEventDestroy(c->ev_out);
  (void)hipStreamDestroy(c->stream);
  ms_ctx* o = c->owner;
  delete c;
  group_unref(g);
  ctx_release(o);
}

}  // extern "C"
