// comm.hip — the RCCL communicator of a column-sharded solve (one process per GPU, xGMI).
//
// (Round 3: every collective of a solve is issued on the SOLVE's stream — one communicator, one stream, issue
// order = stream order on every rank; `stream` below only serves bsn_comm_allreduce, the stand-alone self-test.)
// north_star: "SNP columns shard naturally across the 8 GPUs of one node, with the SVD's panel
// all-reduced over RCCL/xGMI".  The collectives of bsn_bed_randomsvd run INSIDE the library, on
// device buffers and HIP streams it owns (svd.hip); the host program only has to carry the 128-byte
// RCCL unique id from rank 0 to the other ranks (any channel: MPI, a socket, torch.distributed's
// store) and call bsn_comm_init on every rank.
//
// RCCL is bound at run time (dlopen of librccl.so.1) the first time a communicator is asked for, so
// that single-GPU users — the R package on a workstation — do not need the library at all, and so
// that a host process which already carries an RCCL (PyTorch) shares that one instead of loading a
// second copy.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>

#include "bsn_internal.hpp"

namespace bsn {

struct Rccl {
  void *h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclReduceScatter) ReduceScatter = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;   // optional
};

static Rccl &rccl() {
  static Rccl r;
  if (r.h) return r;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  // BSN_RCCL_LIBRARY: another library with RCCL's entry points (the tests load a shared-memory stand-in
  // that lets two ranks share the one GPU of the test box, tests/native/mock_rccl.cpp)
  if (const char *over = getenv("BSN_RCCL_LIBRARY")) {
    h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
    if (!h) fail("cannot load BSN_RCCL_LIBRARY=%s: %s", over, dlerror());
  }
  for (const char *nm : names) {
    if (h) break;
    h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
  }
  if (!h) fail("cannot load RCCL (librccl.so.1): %s", dlerror());
#define BSN_SYM(name)                                                      \
  r.name = (decltype(r.name))dlsym(h, "nccl" #name);                       \
  if (!r.name) fail("RCCL symbol nccl" #name " not found: %s", dlerror());
  BSN_SYM(GetUniqueId) BSN_SYM(CommInitRank) BSN_SYM(CommDestroy) BSN_SYM(AllReduce)
  BSN_SYM(ReduceScatter) BSN_SYM(AllGather) BSN_SYM(GetErrorString)
#undef BSN_SYM
  r.CommAbort = (decltype(r.CommAbort))dlsym(h, "ncclCommAbort");
  r.h = h;
  return r;
}

#define BSN_NCCL(expr)                                                                      \
  do {                                                                                      \
    ncclResult_t r__ = (expr);                                                              \
    if (r__ != ncclSuccess)                                                                 \
      ::bsn::fail("RCCL error %s at %s:%d (%s)", ::bsn::rccl().GetErrorString(r__), __FILE__, \
                  __LINE__, #expr);                                                         \
  } while (0)

static void check_alive(bsn_comm *c) {
  if (c->aborted.load()) fail("the communicator was aborted (a collective of a sharded solve did not finish in time)");
}
hipEvent_t comm_time_begin(bsn_comm *c, hipStream_t st) {
  if (!c->timing) return nullptr;
  hipEvent_t e = nullptr;
  if (!c->ev_pool.empty()) {
    e = c->ev_pool.back();
    c->ev_pool.pop_back();
  } else {
    BSN_HIP(hipEventCreate(&e));
  }
  BSN_HIP(hipEventRecord(e, st));
  return e;
}
void comm_time_end(bsn_comm *c, hipEvent_t begin, int cls, hipStream_t st) {
  if (!begin) return;
  hipEvent_t e = nullptr;
  if (!c->ev_pool.empty()) {
    e = c->ev_pool.back();
    c->ev_pool.pop_back();
  } else {
    BSN_HIP(hipEventCreate(&e));
  }
  BSN_HIP(hipEventRecord(e, st));
  c->timed.push_back({begin, e, cls});
}
void comm_time_collect(bsn_comm *c, double ms[kCommClasses], int count[kCommClasses]) {
  for (int k = 0; k < kCommClasses; k++) ms[k] = 0, count[k] = 0;
  for (auto &t : c->timed) {
    float f = 0;
    if (hipEventSynchronize(t.b) == hipSuccess && hipEventElapsedTime(&f, t.a, t.b) == hipSuccess) {
      ms[t.cls] += f;
      count[t.cls]++;
    }
    c->ev_pool.push_back(t.a);
    c->ev_pool.push_back(t.b);
  }
  c->timed.clear();
  (void)hipGetLastError();
}
bool comm_abort(bsn_comm *c) {
  if (c->aborted.exchange(1)) return true;
  if (!rccl().CommAbort || !c->comm) return false;
  // (the handle stays in place: the solve's thread may be inside a collective call that holds it; bsn_comm_destroy
  // does not hand an aborted communicator to ncclCommDestroy)
  (void)rccl().CommAbort((ncclComm_t)c->comm);
  return true;
}

void comm_allreduce_sum(bsn_comm *c, double *d_buf, int64_t count, hipStream_t st) {
  check_alive(c);
  hipEvent_t t0 = comm_time_begin(c, st);
  BSN_NCCL(rccl().AllReduce(d_buf, d_buf, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)c->comm, st));
  comm_time_end(c, t0, 2, st);
}
void comm_reduce_scatter_sum(bsn_comm *c, const double *d_send, double *d_recv, int64_t recv_count,
                             hipStream_t st) {
  check_alive(c);
  hipEvent_t t0 = comm_time_begin(c, st);
  BSN_NCCL(rccl().ReduceScatter(d_send, d_recv, (size_t)recv_count, ncclDouble, ncclSum, (ncclComm_t)c->comm, st));
  comm_time_end(c, t0, 0, st);
}
void comm_all_gather(bsn_comm *c, const double *d_send, double *d_recv, int64_t send_count, hipStream_t st) {
  check_alive(c);
  hipEvent_t t0 = comm_time_begin(c, st);
  BSN_NCCL(rccl().AllGather(d_send, d_recv, (size_t)send_count, ncclDouble, (ncclComm_t)c->comm, st));
  comm_time_end(c, t0, send_count >= 4096 ? 1 : 2, st);   // (a basis block / u against the column maxima)
}

}  // namespace bsn

using namespace bsn;

extern "C" {

int bsn_comm_unique_id(uint8_t *id_out) {
  return guarded([&] {
    static_assert(sizeof(ncclUniqueId) == BSN_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    BSN_NCCL(rccl().GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
  });
}

int bsn_comm_init(const uint8_t *id, int rank, int world, bsn_comm **out) {
  return guarded([&] {
    if (world < 1 || rank < 0 || rank >= world) fail("bsn_comm_init: rank %d of %d", rank, world);
    std::unique_ptr<bsn_comm> c(new bsn_comm());
    c->rank = rank;
    c->world = world;
    BSN_HIP(hipGetDevice(&c->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    BSN_NCCL(rccl().CommInitRank(&comm, world, uid, rank));
    c->comm = comm;
    // from here on a failure must give the communicator (and whatever else exists) back
    struct Undo {
      bsn_comm *c;
      ~Undo() {
        if (!c) return;
        if (c->stream) (void)hipStreamDestroy(c->stream);
        if (c->comm) (void)rccl().CommDestroy((ncclComm_t)c->comm);
      }
    } undo{c.get()};
    BSN_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    undo.c = nullptr;
    *out = c.release();
  });
}

int bsn_comm_rank(const bsn_comm *c) { return c->rank; }
int bsn_comm_world(const bsn_comm *c) { return c->world; }

int bsn_comm_allreduce(bsn_comm *c, double *d_buf, int64_t count) {
  return guarded([&] {
    BSN_HIP(hipSetDevice(c->device));
    comm_allreduce_sum(c, d_buf, count, c->stream);
    BSN_HIP(hipStreamSynchronize(c->stream));
  });
}

int bsn_comm_abort(bsn_comm *c) {
  return guarded([&] {
    if (!c) return;
    if (!comm_abort(c)) fail("this RCCL has no ncclCommAbort");
  });
}

int bsn_comm_destroy(bsn_comm *c) {
  return guarded([&] {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm && !c->aborted.load()) (void)rccl().CommDestroy((ncclComm_t)c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
  });
}

}  // extern "C"
