// TEST INFRASTRUCTURE — not part of the product.
//
// A stand-in for librccl.so that lets several ranks share ONE GPU: the test box has a single MI355X and RCCL
// refuses two ranks on one device, so the multi-rank code of libbigsnpr_hip.so (comm.hip: which buffers, which
// counts, which order of ncclReduceScatter / ncclAllReduce / ncclAllGather on which stream) could otherwise
// only run with one rank.  This library exports the seven RCCL entry points comm.hip binds and moves the
// data through POSIX shared memory, synchronously: copy the send buffer to the rank's slot, barrier, reduce /
// gather on the host, barrier, copy the result to the receive buffer.  It is loaded instead of RCCL only when
// the environment names it (BSN_RCCL_LIBRARY); tests/test_gpu_comm.py does that for its two-rank cases.
// What it cannot show is RCCL itself (transport, topology, asynchrony): that is the driver's multi-GPU run.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3,
               ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef void *ncclComm_t;
typedef enum { ncclDouble = 8 } ncclDataType_t;  // RCCL's value of ncclFloat64
typedef enum { ncclSum = 0 } ncclRedOp_t;

}  // extern "C"

namespace {

constexpr size_t kSlot = (size_t)512 << 20;  // bytes per rank (sparse: only touched pages exist)
constexpr int kMaxRanks = 8;
constexpr uint32_t kMagic = 0x4D4F434Bu;

struct Shared {
  std::atomic<uint32_t> ready;
  pthread_barrier_t barrier;
};

struct Comm {
  int rank = 0, world = 1;
  std::atomic<int> abort{0};            // ncclCommAbort
  std::vector<hipStream_t> seen;        // streams that have carried an all-reduce / all-gather of this communicator
  char name[64] = {0};
  uint8_t *base = nullptr;
  size_t bytes = 0;
  Shared *sh() { return (Shared *)base; }
  double *slot(int r) { return (double *)(base + 4096 + (size_t)r * kSlot); }
};

ncclResult_t fail(const char *what) {
  std::fprintf(stderr, "[mock rccl] %s\n", what);
  return ncclInternalError;
}

bool sync_ranks(Comm *c) {
  const int rc = pthread_barrier_wait(&c->sh()->barrier);
  return rc == 0 || rc == PTHREAD_BARRIER_SERIAL_THREAD;
}

// kind 0: all-reduce (count per rank in, count out); 1: reduce-scatter (world * count in, count out);
// 2: all-gather (count in, world * count out)
ncclResult_t collective(int kind, const void *send, void *recv, size_t count, Comm *c, hipStream_t st) {
  const size_t in = kind == 1 ? count * c->world : count;
  const size_t out = kind == 2 ? count * c->world : count;
  static const bool trace = getenv("MOCK_RCCL_TRACE") != nullptr;  // rank 0: one line per collective
  if (trace && c->rank == 0)
    std::fprintf(stderr, "[mock rccl] %s count %zu (%zu bytes in, %zu out) stream %p\n",
                 kind == 0 ? "all-reduce" : kind == 1 ? "reduce-scatter" : "all-gather", count, in * 8, out * 8, (void *)st);
  if (c->abort.load()) return fail("communicator aborted");
  // MOCK_RCCL_STALL=second_stream: a reduce-scatter on a stream that never carried an all-reduce / all-gather of this
  // communicator — the exchange stream of the overlapped product pass (svd.hip) — never completes: the call blocks
  // until ncclCommAbort, like a transport that cannot progress two streams at once.  What the watchdog of a sharded
  // solve and the fall-back of bigsnpr_amd.comm.negotiate are tested against.
  static const char *stall = getenv("MOCK_RCCL_STALL");
  bool known = false;
  for (hipStream_t s : c->seen) known = known || s == st;
  if (kind != 1 && !known) c->seen.push_back(st);
  if (stall && !std::strcmp(stall, "second_stream") && kind == 1 && !known) {
    if (trace) std::fprintf(stderr, "[mock rccl] rank %d: stalling the reduce-scatter on stream %p\n", c->rank, (void *)st);
    while (!c->abort.load()) usleep(2000);
    return fail("collective aborted");
  }
  if (in * 8 > kSlot) return fail("message larger than the mock's slot");
  if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(c->slot(c->rank), send, in * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  if (!sync_ranks(c)) return fail("barrier");
  std::vector<double> res(out);
  if (kind == 0) {
    for (size_t i = 0; i < count; i++) {
      double s = 0;
      for (int r = 0; r < c->world; r++) s += c->slot(r)[i];  // rank order: the same sum on every rank
      res[i] = s;
    }
  } else if (kind == 1) {
    for (size_t i = 0; i < count; i++) {
      double s = 0;
      for (int r = 0; r < c->world; r++) s += c->slot(r)[(size_t)c->rank * count + i];
      res[i] = s;
    }
  } else {
    for (int r = 0; r < c->world; r++) std::memcpy(res.data() + (size_t)r * count, c->slot(r), count * 8);
  }
  if (!sync_ranks(c)) return fail("barrier");  // everybody has read the slots
  if (hipMemcpy(recv, res.data(), out * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

}  // namespace

extern "C" {

const char *ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled HIP error (mock)";
    case ncclInvalidArgument: return "invalid argument (mock)";
    case ncclInvalidUsage: return "invalid usage (mock)";
    default: return "internal error (mock)";
  }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  std::memset(id, 0, sizeof(*id));
  unsigned long long r = 0;
  FILE *f = std::fopen("/dev/urandom", "rb");
  if (f) {
    if (std::fread(&r, sizeof(r), 1, f) != 1) r = 0;
    std::fclose(f);
  }
  std::snprintf(id->internal, sizeof(id->internal), "/bsnmock_%d_%llx", (int)getpid(), r);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  Comm *c = new Comm();
  c->rank = rank;
  c->world = nranks;
  std::strncpy(c->name, id.internal, sizeof(c->name) - 1);
  c->bytes = 4096 + (size_t)nranks * kSlot;
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) return fail("shm_open / ftruncate");
  c->base = (uint8_t *)mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->base == (uint8_t *)MAP_FAILED) return fail("mmap");
  if (rank == 0) {
    pthread_barrierattr_t at;
    pthread_barrierattr_init(&at);
    pthread_barrierattr_setpshared(&at, PTHREAD_PROCESS_SHARED);
    pthread_barrier_init(&c->sh()->barrier, &at, (unsigned)nranks);
    pthread_barrierattr_destroy(&at);
    c->sh()->ready.store(kMagic, std::memory_order_release);
  } else {
    for (int spin = 0; c->sh()->ready.load(std::memory_order_acquire) != kMagic; spin++) {
      if (spin > 60000) return fail("rank 0 never initialised the segment");
      usleep(1000);
    }
  }
  if (!sync_ranks(c)) return fail("barrier");
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t comm) {
  Comm *c = (Comm *)comm;
  if (c) c->abort.store(1);   // (nothing is unmapped: a stalled call of another thread is still reading the handle)
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm *c = (Comm *)comm;
  if (!c) return ncclSuccess;
  sync_ranks(c);
  if (c->rank == 0) shm_unlink(c->name);
  munmap(c->base, c->bytes);
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t st) {
  if (dt != ncclDouble || op != ncclSum) return ncclInvalidArgument;
  return collective(0, send, recv, count, (Comm *)comm, st);
}

ncclResult_t ncclReduceScatter(const void *send, void *recv, size_t recvcount, ncclDataType_t dt, ncclRedOp_t op,
                               ncclComm_t comm, hipStream_t st) {
  if (dt != ncclDouble || op != ncclSum) return ncclInvalidArgument;
  return collective(1, send, recv, recvcount, (Comm *)comm, st);
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t sendcount, ncclDataType_t dt, ncclComm_t comm,
                           hipStream_t st) {
  if (dt != ncclDouble) return ncclInvalidArgument;
  return collective(2, send, recv, sendcount, (Comm *)comm, st);
}

}  // extern "C"
