// Native transport for ms_prove_sharded: the two exchanges of the ms_comm table (include/mstark.h) on RCCL over xGMI,
// with no Python and no torch in the data path, so that the reference-side (Rust) host of INTEGRATION.md can run the
// joint proof of BASELINE config 3 by itself. librccl is loaded at run time (dlopen), like hiprtc: the prover library
// has no link-time dependency on it and single-GPU users never load it.
//   all_to_all  = grouped ncclSend / ncclRecv, one pair per peer (xGMI is point-to-point: one link per pair)
//   all_gather  = ncclAllGather
// Both run on the transport's own stream; the blocking forms return when the data has arrived (the ms_comm contract),
// the start / wait pair lets the prover overlap the exchange of one column group with the transforms of the next.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <mutex>
#include <string>

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
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_api;

// one copy of RCCL per process: if the host has already loaded one (torch ships its own), use that one; otherwise the
// library MSAMD_RCCL_LIB names (and only that one), otherwise the usual names
const RcclApi& rccl() {
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (g_api.lib) return g_api;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  const char* env = getenv("MSAMD_RCCL_LIB");
  if (!h && env && *env) h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
  for (const char* n : names)
    if (!h && !(env && *env)) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    const char* why = dlerror();  // (one call: dlerror() clears the message it returns)
    throw std::runtime_error(std::string("cannot load librccl: ") + (why ? why : "not found"));
  }
  RcclApi a;
  a.lib = h;
#define MS_SYM(field, name)                                                       \
  a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));                  \
  if (!a.field) throw std::runtime_error(std::string("librccl lacks ") + name)
  MS_SYM(GetUniqueId, "ncclGetUniqueId");
  MS_SYM(CommInitRank, "ncclCommInitRank");
  MS_SYM(CommDestroy, "ncclCommDestroy");
  MS_SYM(GroupStart, "ncclGroupStart");
  MS_SYM(GroupEnd, "ncclGroupEnd");
  MS_SYM(Send, "ncclSend");
  MS_SYM(Recv, "ncclRecv");
  MS_SYM(AllGather, "ncclAllGather");
  MS_SYM(GetErrorString, "ncclGetErrorString");
#undef MS_SYM
  a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(dlsym(h, "ncclCommAbort"));
  g_api = a;
  return g_api;
}

void nccl_check(ncclResult_t r, const char* what) {
  if (r != ncclSuccess) throw std::runtime_error(std::string("RCCL ") + what + ": " + rccl().GetErrorString(r));
}
// ncclGroupStart ... ncclGroupEnd that is closed on every path: an exception between the two would otherwise leave the
// thread inside an open group, and every later RCCL call of that thread - the next proof's all_gather - would be queued
// into it and never launched (the peers hang instead of failing). end() reports the group's own result; the destructor
// only closes.
struct GroupGuard {
  bool open = false;
  GroupGuard() {
    nccl_check(rccl().GroupStart(), "ncclGroupStart");
    open = true;
  }
  void end() {
    open = false;
    nccl_check(rccl().GroupEnd(), "ncclGroupEnd");
  }
  ~GroupGuard() {
    if (open) (void)rccl().GroupEnd();
  }
};
// most point-to-point operations in one group (MSAMD_RCCL_GROUP_OPS): the column exchange issues one send and one receive
// per peer and column, and splits its columns over several groups above this count
size_t max_group_ops() {
  static const size_t n = [] {
    const char* e = getenv("MSAMD_RCCL_GROUP_OPS");
    const long v = e ? atol(e) : 0;
    return (size_t)(v > 0 ? v : 128);
  }();
  return n;
}
}  // namespace

struct ms_comm_rccl {
  ms_ctx* owner = nullptr;
  Ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  int rank = 0, world = 1;
  uint64_t bytes_moved = 0;
  ms_comm table;
  // stream-ordered mode (ms_comm.set_stream_ordered): `peer` is the caller's stream; events order the two streams
  hipStream_t peer = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  uint64_t groups_issued = 0;
  bool failed = false;
  // MSAMD_RCCL_SELF=1 (diagnostics, one-GPU boxes): a single rank still creates a communicator and sends its own block to
  // itself through ncclSend / ncclRecv / ncclAllGather instead of copying it, so that every RCCL entry point this file
  // binds - the dlsym'd signatures, the groups, the stream and event ordering around RCCL's kernels - runs on a box where
  // RCCL refuses a second rank on the same device
  bool self_peer = false;
  // An RCCL call of this rank has failed: the peers are waiting for operations that will never be issued. Abort the
  // communicator so that they get an error instead of a hang, and refuse further use of this transport.
  void fail() {
    if (failed) return;
    failed = true;
    peer = nullptr;
    if (comm && rccl().CommAbort) {
      (void)rccl().CommAbort(comm);
      comm = nullptr;
    }
  }
  void usable() const {
    if (failed) throw std::runtime_error("RCCL transport: an earlier exchange failed and the communicator was aborted; create a new transport");
  }
  void begin() {  // the exchange that follows runs behind everything queued on the caller's stream so far
    usable();
    if (!peer) return;
    HIP_CHECK(hipEventRecord(ev_in, peer));
    HIP_CHECK(hipStreamWaitEvent(stream, ev_in, 0));
  }
  void complete() {  // what the caller queues next runs behind every exchange given to this transport so far
    if (peer) {
      HIP_CHECK(hipEventRecord(ev_out, stream));
      HIP_CHECK(hipStreamWaitEvent(peer, ev_out, 0));
    } else {
      HIP_CHECK(hipStreamSynchronize(stream));
    }
  }

  void exchange(const uint8_t* send, size_t send_stride, uint8_t* recv, size_t recv_stride, size_t n) {
    bytes_moved += n * (size_t)world;
    if (n == 0) return;
    if (world > 1 || self_peer) {
      try {
        GroupGuard grp;
        for (int k = 0; k < world; k++) {
          if (k == rank && !self_peer) continue;
          nccl_check(rccl().Send(send + (size_t)k * send_stride, n, ncclUint8, k, comm, stream), "ncclSend");
          nccl_check(rccl().Recv(recv + (size_t)k * recv_stride, n, ncclUint8, k, comm, stream), "ncclRecv");
        }
        grp.end();
      } catch (...) {
        fail();
        throw;
      }
    }
    if (self_peer) return;
    // this rank's own block never leaves the device
    HIP_CHECK(hipMemcpyAsync(recv + (size_t)rank * recv_stride, send + (size_t)rank * send_stride, n, hipMemcpyDeviceToDevice, stream));
  }
  // the exchange read straight out of a column-major matrix (ms_comm.all_to_all_cols_start): one ncclSend / ncclRecv per
  // peer and column in ONE group (4 MB segments at the bench size); this rank's own rows are a strided device copy
  void exchange_cols(const uint8_t* send, size_t sps, size_t scs, uint8_t* recv, size_t rps, size_t rcs, size_t ncols, size_t seg,
                     bool skip_self = false) {
    bytes_moved += seg * ncols * (size_t)(skip_self ? world - 1 : world);
    if (seg == 0 || ncols == 0) return;
    const bool self = self_peer && !skip_self;
    if (world > 1 || self) {
      // every rank cuts the columns into the same groups, and inside a group every send has its receive on the peer: the
      // groups complete one after the other on all ranks. Group size bounded (7 peers x 7 columns x 2 = 98 operations for a
      // quarter of the stage-2 LDE at world 8; wider circuits would otherwise grow the group without limit).
      const size_t per_col = 2 * (size_t)std::max(1, world - 1);
      const size_t cols_per_group = std::max<size_t>(1, max_group_ops() / per_col);
      try {
        for (size_t c0 = 0; c0 < ncols; c0 += cols_per_group) {
          const size_t c1 = std::min(ncols, c0 + cols_per_group);
          GroupGuard grp;
          for (int k = 0; k < world; k++) {
            if (k == rank && !self) continue;
            for (size_t c = c0; c < c1; c++) {
              nccl_check(rccl().Send(send + (size_t)k * sps + c * scs, seg, ncclUint8, k, comm, stream), "ncclSend");
              nccl_check(rccl().Recv(recv + (size_t)k * rps + c * rcs, seg, ncclUint8, k, comm, stream), "ncclRecv");
            }
          }
          grp.end();
          groups_issued++;
        }
      } catch (...) {
        fail();
        throw;
      }
    }
    if (!skip_self && !self)
      HIP_CHECK(hipMemcpy2DAsync(recv + (size_t)rank * rps, rcs, send + (size_t)rank * sps, scs, seg, ncols, hipMemcpyDeviceToDevice, stream));
  }
  // one rank's matrix handed out by row ranges (ms_comm.scatter_cols_start): the root sends ncols segments to every other
  // rank, every other rank receives its ncols segments; groups bounded like the column exchange's
  void scatter_cols(int root, const uint8_t* send, size_t sps, size_t scs, uint8_t* recv, size_t rcs, size_t ncols, size_t seg) {
    if (root < 0 || root >= world) throw std::runtime_error("scatter_cols_start: root out of range");
    bytes_moved += rank == root ? seg * ncols * (size_t)(world - 1) : seg * ncols;
    if (seg == 0 || ncols == 0 || world == 1) return;
    const size_t per_col = rank == root ? (size_t)(world - 1) : 1;
    const size_t cols_per_group = std::max<size_t>(1, max_group_ops() / (size_t)(world - 1));  // the same cut on every rank
    (void)per_col;
    try {
      for (size_t c0 = 0; c0 < ncols; c0 += cols_per_group) {
        const size_t c1 = std::min(ncols, c0 + cols_per_group);
        GroupGuard grp;
        if (rank == root) {
          for (int k = 0; k < world; k++) {
            if (k == rank) continue;
            for (size_t c = c0; c < c1; c++) nccl_check(rccl().Send(send + (size_t)k * sps + c * scs, seg, ncclUint8, k, comm, stream), "ncclSend");
          }
        } else {
          for (size_t c = c0; c < c1; c++) nccl_check(rccl().Recv(recv + c * rcs, seg, ncclUint8, root, comm, stream), "ncclRecv");
        }
        grp.end();
        groups_issued++;
      }
    } catch (...) {
      fail();
      throw;
    }
  }
  void gather(const void* send, void* recv, size_t n) {
    bytes_moved += n * (size_t)world;
    if (n == 0) return;
    if (world > 1 || self_peer) {
      try {
        nccl_check(rccl().AllGather(send, recv, n, ncclUint8, comm, stream), "ncclAllGather");
      } catch (...) {
        fail();
        throw;
      }
    } else
      HIP_CHECK(hipMemcpyAsync(recv, send, n, hipMemcpyDeviceToDevice, stream));
  }
};

namespace {
template <class F>
int32_t guarded(ms_comm_rccl* c, F f) {
  try {
    HIP_CHECK(hipSetDevice(c->ctx->device));
    f();
    return 0;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return -1;
  }
}
int32_t cb_all_to_all(void* user, const void* send, void* recv, size_t per_peer) {
  ms_comm_rccl* c = (ms_comm_rccl*)user;
  return guarded(c, [&] {
    c->begin();
    c->exchange((const uint8_t*)send, per_peer, (uint8_t*)recv, per_peer, per_peer);
    c->complete();
  });
}
int32_t cb_all_gather(void* user, const void* send, void* recv, size_t bytes) {
  ms_comm_rccl* c = (ms_comm_rccl*)user;
  return guarded(c, [&] {
    c->begin();
    c->gather(send, recv, bytes);
    c->complete();
  });
}
int32_t cb_start(void* user, const void* send, size_t send_stride, void* recv, size_t recv_stride, size_t per_peer) {
  ms_comm_rccl* c = (ms_comm_rccl*)user;
  return guarded(c, [&] {
    c->begin();
    c->exchange((const uint8_t*)send, send_stride, (uint8_t*)recv, recv_stride, per_peer);
  });
}
int32_t cb_cols_start(void* user, const void* send, size_t sps, size_t scs, void* recv, size_t rps, size_t rcs, size_t ncols, size_t seg) {
  ms_comm_rccl* c = (ms_comm_rccl*)user;
  return guarded(c, [&] {
    c->begin();
    c->exchange_cols((const uint8_t*)send, sps, scs, (uint8_t*)recv, rps, rcs, ncols, seg);
  });
}
int32_t cb_cols_start2(void* user, const void* send, size_t sps, size_t scs, void* recv, size_t rps, size_t rcs, size_t ncols, size_t seg,
                       uint32_t flags) {
  ms_comm_rccl* c = (ms_comm_rccl*)user;
  return guarded(c, [&] {
    c->begin();
    c->exchange_cols((const uint8_t*)send, sps, scs, (uint8_t*)recv, rps, rcs, ncols, seg, (flags & MS_COMM_SKIP_SELF) != 0);
  });
}
int32_t cb_scatter(void* user, int32_t root, const void* send, size_t sps, size_t scs, void* recv, size_t rcs, size_t ncols, size_t seg) {
  ms_comm_rccl* c = (ms_comm_rccl*)user;
  return guarded(c, [&] {
    c->begin();
    c->scatter_cols(root, (const uint8_t*)send, sps, scs, (uint8_t*)recv, rcs, ncols, seg);
  });
}
int32_t cb_wait(void* user) {
  ms_comm_rccl* c = (ms_comm_rccl*)user;
  return guarded(c, [&] { c->complete(); });
}
void cb_abort(void* user, const char*) {
  try {
    static_cast<ms_comm_rccl*>(user)->fail();  // ncclCommAbort: the peers' pending and future operations return an error
  } catch (...) {
  }
}
int32_t cb_set_stream_ordered(void* user, void* hip_stream) {
  ms_comm_rccl* c = (ms_comm_rccl*)user;
  return guarded(c, [&] {
    if (!hip_stream && c->peer) HIP_CHECK(hipStreamSynchronize(c->stream));  // leaving the mode: nothing of ours stays in flight
    if (hip_stream && !c->ev_in) {
      HIP_CHECK(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
      HIP_CHECK(hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming));
    }
    c->peer = (hipStream_t)hip_stream;
  });
}
}  // namespace

extern "C" {

int32_t ms_comm_rccl_unique_id(uint8_t out[MS_RCCL_UNIQUE_ID_BYTES]) {
  try {
    static_assert(sizeof(ncclUniqueId) == MS_RCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    nccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(out, &id, sizeof(id));
    return MS_OK;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return MS_ERR;
  }
}

int32_t ms_comm_rccl_create(ms_ctx* ctx, const uint8_t unique_id[MS_RCCL_UNIQUE_ID_BYTES], int32_t rank, int32_t world,
                            ms_comm_rccl** out) {
  *out = nullptr;
  ms_comm_rccl* c = nullptr;
  try {
    if (world < 1 || rank < 0 || rank >= world) throw std::runtime_error("ms_comm_rccl_create: rank out of range");
    c = new ms_comm_rccl();
    c->ctx = ctx_of(ctx);
    c->rank = rank;
    c->world = world;
    HIP_CHECK(hipSetDevice(c->ctx->device));
    HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->self_peer = world == 1 && getenv("MSAMD_RCCL_SELF") != nullptr;
    if (world > 1 || c->self_peer) {
      ncclUniqueId id;
      memcpy(&id, unique_id, sizeof(id));
      nccl_check(rccl().CommInitRank(&c->comm, world, id, rank), "ncclCommInitRank");
    }
    memset(&c->table, 0, sizeof(c->table));
    c->table.size = (uint32_t)sizeof(ms_comm);
    c->table.abort = cb_abort;
    c->table.rank = rank;
    c->table.world = world;
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

const ms_comm* ms_comm_rccl_table(ms_comm_rccl* c) { return c ? &c->table : nullptr; }
uint64_t ms_comm_rccl_bytes_moved(ms_comm_rccl* c) { return c ? c->bytes_moved : 0; }

void ms_comm_rccl_destroy(ms_comm_rccl* c) {
  if (!c) return;
  (void)hipSetDevice(c->ctx->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  if (c->ev_in) (void)hipEventDestroy(c->ev_in);
  if (c->ev_out) (void)hipEventDestroy(c->ev_out);
  (void)hipStreamDestroy(c->stream);
  ms_ctx* o = c->owner;
  delete c;
  ctx_release(o);
}

}  // extern "C"
