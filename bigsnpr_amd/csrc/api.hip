// api.hip — the extern "C" surface declared in include/bigsnpr_hip.h.
#include <dlfcn.h>
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <condition_variable>
#include <chrono>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <sched.h>
#include <atomic>
#include <thread>

#include <cstdlib>

#include "bsn_internal.hpp"
#include <cxxabi.h>

namespace bsn {

// ---- roctx (optional) ------------------------------------------------------------------------------------------
namespace {
struct Roctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    if (!getenv("BSN_ROCTX")) return;
    void *h = nullptr;
    for (const char *name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"})
      if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return;
    push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
    pop = (int (*)())dlsym(h, "roctxRangePop");
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
};
Roctx &roctx() {
  static Roctx r;
  return r;
}
}  // namespace
void roctx_push(const char *name) {
  if (roctx().push) roctx().push(name);
}
void roctx_pop() {
  if (roctx().pop) roctx().pop();
}

static thread_local std::string g_err;
void set_error(const char *msg) { g_err = msg ? msg : ""; }
void fail(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  throw Error(buf);
}

void require_gpu() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    fail("no HIP device available: libbigsnpr_hip has no CPU fallback (%s)",
         e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
}

// a NULL index list means the first `len` rows / columns, everywhere in this ABI
static std::vector<int32_t> to_i32(const int64_t *ind, int64_t len, int64_t limit, const char *what) {
  std::vector<int32_t> v((size_t)len);
  if (!ind) {
    if (len > limit)
      fail("Tested %lld < %lld. Subscript out of bounds (%s).", (long long)(len - 1), (long long)limit, what);
    for (int64_t i = 0; i < len; i++) v[(size_t)i] = (int32_t)i;
    return v;
  }
  for (int64_t i = 0; i < len; i++) {
    if (ind[i] < 0 || ind[i] >= limit)
      // bigstatsr vec_int_to_size(): "Tested %s < %s. Subscript out of bounds."
      fail("Tested %lld < %lld. Subscript out of bounds (%s).", (long long)ind[i], (long long)limit,
           what);
    v[(size_t)i] = (int32_t)ind[i];
  }
  return v;
}

// ---- cache of released device blocks (bsn_internal.hpp) ---------------------------------
namespace {
struct DevCache {
  std::mutex mu;
  std::multimap<size_t, void *> free_blocks[16];  // per device, keyed by block size
  size_t held[16] = {0};
};
DevCache &dev_cache() {
  static DevCache *c = new DevCache();  // never destroyed: blocks may be released during process exit
  return *c;
}
// sizes are rounded to 2^k or 1.5 * 2^k (>= 4 KiB) so that a repeated call finds its blocks again
size_t size_class(size_t bytes) {
  size_t c = 4096;
  while (c < bytes) {
    if (c + c / 2 >= bytes) return c + c / 2;
    c *= 2;
  }
  return c;
}
}  // namespace

void *dev_alloc(size_t bytes, size_t *granted, int *device) {
  if (bytes == 0) bytes = 1;
  int dev = 0;
  BSN_HIP(hipGetDevice(&dev));
  *device = dev;
  const bool cached = bytes <= kDevCacheMaxBlock && dev < 16;
  const size_t want = cached ? size_class(bytes) : bytes;
  if (cached) {
    DevCache &c = dev_cache();
    std::lock_guard<std::mutex> lk(c.mu);
    auto it = c.free_blocks[dev].find(want);
    if (it != c.free_blocks[dev].end()) {
      void *p = it->second;
      c.free_blocks[dev].erase(it);
      c.held[dev] -= want;
      *granted = want;
      return p;
    }
  }
  static const bool trace = getenv("BSN_ALLOC_TRACE") != nullptr;  // every real allocation on stderr
  void *p = nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t e = hipMalloc(&p, want);
  if (trace)
    std::fprintf(stderr, "[bsn alloc] hipMalloc %zu bytes (asked %zu): %.2f ms\n", want, bytes,
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  if (e == hipErrorOutOfMemory) {  // give the cached blocks back and try once more
    (void)hipGetLastError();
    dev_cache_flush();
    e = hipMalloc(&p, want);
  }
  if (e != hipSuccess) fail("HIP error %s allocating %zu bytes of device memory", hipGetErrorString(e), want);
  *granted = want;
  return p;
}

void dev_release(void *p, size_t granted, int dev) {
  if (!p) return;
  DevCache &c = dev_cache();
  if (granted <= kDevCacheMaxBlock && dev >= 0 && dev < 16 && size_class(granted) == granted) {
    // hipFree's ordering: nothing on the device still uses the block once it can be handed out again
    int cur = dev;
    (void)hipGetDevice(&cur);
    if (cur != dev) (void)hipSetDevice(dev);
    (void)hipDeviceSynchronize();
    if (cur != dev) (void)hipSetDevice(cur);
    std::lock_guard<std::mutex> lk(c.mu);
    if (c.held[dev] + granted <= kDevCacheMaxTotal) {
      c.free_blocks[dev].emplace(granted, p);
      c.held[dev] += granted;
      return;
    }
  }
  static const bool trace = getenv("BSN_ALLOC_TRACE") != nullptr;
  if (trace) std::fprintf(stderr, "[bsn alloc] hipFree %zu bytes\n", granted);
  (void)hipFree(p);
}

size_t dev_cache_held() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  DevCache &c = dev_cache();
  std::lock_guard<std::mutex> lk(c.mu);
  return c.held[dev];
}

void dev_cache_flush() {
  DevCache &c = dev_cache();
  std::lock_guard<std::mutex> lk(c.mu);
  for (int d = 0; d < 16; d++) {
    for (auto &kv : c.free_blocks[d]) (void)hipFree(kv.second);
    c.free_blocks[d].clear();
    c.held[d] = 0;
  }
}

constexpr size_t kStagePiece = 32u << 20;

void stage_init(bsn_bed *b) {
  for (int i = 0; i < 2; i++) {
    if (!b->h_stage[i]) BSN_HIP(hipHostMalloc((void **)&b->h_stage[i], kStagePiece, hipHostMallocDefault));
    if (!b->ev_stage[i]) BSN_HIP(hipEventCreateWithFlags(&b->ev_stage[i], hipEventDisableTiming));
  }
}

// Helper threads that live as long as the process (created on first use, re-created in a forked child) take their
// share of a host copy of 256 KB or more: the three m-vectors of a one-shot product at BASELINE config 2 — x, centre,
// scale, 1.6 MB each — cost 0.12 ms apiece through one core, a third of the call's time outside its streaming kernel;
// threads created per copy (round 2's way for pieces of 8 MB and more) cost more than they save at this size.  A
// helper that has just finished a share polls for the next one for some tens of microseconds before it goes to
// sleep: the copies of one call follow each other within that time, so only the first pays a wake-up.
namespace {
struct CopyJob {
  void *dst = nullptr;
  const void *src = nullptr;
  size_t len = 0;
};
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  __asm__ __volatile__("yield");
#else
  std::this_thread::yield();
#endif
}
struct CopyPool {
  static constexpr int kMaxHelpers = 6;
  int nh = 2;
  std::mutex mu;
  std::condition_variable cv_go;
  CopyJob job[kMaxHelpers];
  std::atomic<unsigned long long> seq{0};
  std::atomic<int> pending{0}, sleepers{0};
  pid_t owner = 0;
  void start() {
    owner = getpid();
    cpu_set_t set;
    int cpus = (int)std::thread::hardware_concurrency();
    if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = std::min(cpus > 0 ? cpus : 1, CPU_COUNT(&set));
    nh = std::max(1, std::min(kMaxHelpers, cpus / 2 - 1));
    if (const char *e = getenv("BSN_COPY_THREADS")) nh = std::max(1, std::min(kMaxHelpers, atoi(e) - 1));
    for (int h = 0; h < nh; h++)
      std::thread([this, h] {
        unsigned long long seen = 0;
        for (;;) {
          int spins = 0;
          while (seq.load(std::memory_order_acquire) == seen) {
            if (++spins < 4000) {
              cpu_relax();
              continue;
            }
            std::unique_lock<std::mutex> lk(mu);
            sleepers.fetch_add(1);
            cv_go.wait(lk, [&] { return seq.load() != seen; });
            sleepers.fetch_sub(1);
          }
          seen = seq.load(std::memory_order_acquire);   // (the caller waits for every share before the next copy)
          const CopyJob j = job[h];
          if (j.len) std::memcpy(j.dst, j.src, j.len);
          pending.fetch_sub(1, std::memory_order_release);
        }
      }).detach();
  }
  void copy(void *dst, const void *src, size_t len) {
    const size_t part = (len / (size_t)(nh + 1) + 63) & ~(size_t)63;
    for (int h = 0; h < nh; h++) {
      const size_t lo = part * (size_t)(h + 1), hi = h + 1 == nh ? len : std::min(len, lo + part);
      job[h] = lo < hi ? CopyJob{(uint8_t *)dst + lo, (const uint8_t *)src + lo, hi - lo} : CopyJob{};
    }
    pending.store(nh, std::memory_order_relaxed);
    seq.fetch_add(1);
    if (sleepers.load() > 0) {
      { std::lock_guard<std::mutex> lk(mu); }
      cv_go.notify_all();
    }
    std::memcpy(dst, src, std::min(part, len));
    while (pending.load(std::memory_order_acquire) != 0) cpu_relax();
  }
};
CopyPool *g_pool = nullptr;
// one copy at a time goes through the pool (calls on different handles may be concurrent: the others copy on their own
// thread instead of queueing behind it).  A raw pointer, so that the child of a fork can start from a fresh mutex: the
// parent's may have been held by a thread that does not exist in the child.
std::mutex *g_pool_mu = new std::mutex();
void pool_atfork_child() {
  g_pool_mu = new std::mutex();   // (the parent's mutex and pool object are abandoned in the child, not freed)
  g_pool = nullptr;
}
const int g_pool_atfork = pthread_atfork(nullptr, nullptr, pool_atfork_child);
}  // namespace

// host copy between caller memory and a staging buffer
static void host_copy(void *dst, const void *src, size_t len) {
  static const bool pool_off = [] {
    const char *e = getenv("BSN_COPY_THREADS");  // 1: every copy on the calling thread
    return e && atoi(e) <= 1;
  }();
  if (len < (256u << 10) || pool_off) {
    std::memcpy(dst, src, len);
    return;
  }
  std::unique_lock<std::mutex> lk(*g_pool_mu, std::try_to_lock);
  if (!lk.owns_lock()) {   // another thread's copy is in the pool: do not queue behind it
    std::memcpy(dst, src, len);
    return;
  }
  if (!g_pool || g_pool->owner != getpid()) {   // first use, or a forked child (threads do not survive a fork)
    g_pool = new CopyPool();
    g_pool->start();
  }
  g_pool->copy(dst, src, len);
}

// Host -> device through the two pinned staging buffers, ordered on the handle's stream.  Returns as
// soon as the caller's memory has been read (the last pieces may still be on their way: whatever uses
// d_dst is queued behind them on the same stream).
void copy_h2d(bsn_bed *b, void *d_dst, const void *src, size_t bytes, hipStream_t stream) {
  if (!stream) stream = b->stream;
  stage_init(b);
  for (size_t off = 0; off < bytes; off += kStagePiece) {
    const size_t len = std::min(kStagePiece, bytes - off);
    const int s = (int)(b->stage_next++ & 1);
    if (b->stage_busy[s]) BSN_HIP(hipEventSynchronize(b->ev_stage[s]));  // its previous piece has left the buffer
    host_copy(b->h_stage[s], (const uint8_t *)src + off, len);
    BSN_HIP(hipMemcpyAsync((uint8_t *)d_dst + off, b->h_stage[s], len, hipMemcpyHostToDevice, stream));
    BSN_HIP(hipEventRecord(b->ev_stage[s], stream));
    b->stage_busy[s] = true;
  }
}

// the handle's second stream: uploads that may run beside the kernels of the main one (op_cprod's centre / scale)
hipStream_t upload_stream(bsn_bed *b) {
  if (!b->stream_up) {
    BSN_HIP(hipStreamCreateWithFlags(&b->stream_up, hipStreamNonBlocking));
    BSN_HIP(hipEventCreateWithFlags(&b->ev_up, hipEventDisableTiming));
  }
  return b->stream_up;
}

// true when the runtime knows `p` as page-locked host memory (bsn_host_alloc, or registered by the
// caller): the DMA engines can reach it directly, no staging
static bool host_is_pinned(const void *p) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();   // an ordinary pointer is reported as an error: not one
    return false;
  }
  return at.type == hipMemoryTypeHost;
}

// Device -> host; complete when it returns.
void copy_d2h(bsn_bed *b, void *dst, const void *d_src, size_t bytes) {
  if (bytes >= (1u << 20) && host_is_pinned(dst)) {   // page-locked destination: one DMA, no host copy
    BSN_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, b->stream));
    BSN_HIP(hipStreamSynchronize(b->stream));
    return;
  }
  stage_init(b);
  if (b->stream_up)   // an upload on the second stream may still be reading a staging buffer
    for (int i = 0; i < 2; i++)
      if (b->stage_busy[i]) BSN_HIP(hipEventSynchronize(b->ev_stage[i]));
  const size_t npiece = (bytes + kStagePiece - 1) / kStagePiece;
  for (size_t k = 0; k <= npiece; k++) {
    if (k < npiece) {  // launch piece k (stream order puts it behind any upload still reading the buffer)
      const size_t off = k * kStagePiece, len = std::min(kStagePiece, bytes - off);
      BSN_HIP(hipMemcpyAsync(b->h_stage[k & 1], (const uint8_t *)d_src + off, len, hipMemcpyDeviceToHost, b->stream));
      BSN_HIP(hipEventRecord(b->ev_stage[k & 1], b->stream));
    }
    if (k >= 1) {  // drain piece k - 1 while piece k is on its way
      const size_t off = (k - 1) * kStagePiece, len = std::min(kStagePiece, bytes - off);
      BSN_HIP(hipEventSynchronize(b->ev_stage[(k - 1) & 1]));
      host_copy((uint8_t *)dst + off, b->h_stage[(k - 1) & 1], len);
    }
  }
  b->stage_busy[0] = b->stage_busy[1] = false;  // every event recorded so far has completed
}

static void free_bed(bsn_bed *b) { bed_free(b); }
void bed_free(bsn_bed *b) {
  if (!b) return;
  for (int i = 0; i < 2; i++) {
    if (b->h_stage[i]) (void)hipHostFree(b->h_stage[i]);
    if (b->ev_stage[i]) (void)hipEventDestroy(b->ev_stage[i]);
  }
  if (b->smaj_job) {   // a background build of the sample-major copy reads d_img: let it end before the image goes
    try { image_smaj_wait(b); } catch (...) {}
  }
  if (b->d_img) (void)hipFree(b->d_img);
  if (b->map_base) (void)munmap(b->map_base, b->map_len);
  if (b->fd_file >= 0) (void)close(b->fd_file);
  if (b->slab_stage) delete (bsn::FileStage *)b->slab_stage;
  if (b->slab_img) bed_free(b->slab_img);
  if (b->sub) bed_free(b->sub);
  if (b->d_tiled) (void)hipFree(b->d_tiled);
  if (b->d_smaj) (void)hipFree(b->d_smaj);
  if (b->d_lut) (void)hipFree(b->d_lut);
  if (b->h_na_blocks) (void)hipHostFree(b->h_na_blocks);
  if (b->d_na_blocks) (void)hipFree(b->d_na_blocks);
  if (b->ev0) (void)hipEventDestroy(b->ev0);
  if (b->ev1) (void)hipEventDestroy(b->ev1);
  if (b->ev_up) (void)hipEventDestroy(b->ev_up);
  if (b->stream_up) (void)hipStreamDestroy(b->stream_up);
  if (b->stream && !b->stream_borrowed) (void)hipStreamDestroy(b->stream);
  delete b;
}

__global__ void k_fill_f64(double *p, int64_t n, double v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

void fill_op(bsn_op *op, bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
             int64_t m, const double *center, const double *scale, bool defer_scale, bool allow_streamed) {
  if (n <= 0 || m <= 0) fail("'ind.row' and 'ind.col' can't be empty.");
  if (bed->n >= (int64_t)1 << 31 || bed->m >= (int64_t)1 << 31) fail("dimension too large");
  if (!allow_streamed) require_resident(bed, "this function");
  BSN_HIP(hipSetDevice(bed->device));
  op->bed = bed;
  op->n = n;
  op->m = m;
  // rows (NULL: the first n)
  if (!ind_row && n > bed->n)
    fail("Tested %lld < %lld. Subscript out of bounds (ind.row).", (long long)(n - 1), (long long)bed->n);
  bool ident = (n == bed->n);
  if (ind_row) {
    for (int64_t i = 0; ident && i < n; i++) ident = (ind_row[i] == i);
  }
  op->rows_identity = ident;
  if (!ident) {
    auto r = to_i32(ind_row, n, bed->n, "ind.row");
    copy_h2d(bed, op->d_rows.ensure((size_t)n), r.data(), (size_t)n * 4);
  }
  // cols
  bool contig = true;
  int64_t c0 = ind_col ? ind_col[0] : 0;
  if (ind_col)
    for (int64_t j = 0; contig && j < m; j++) contig = (ind_col[j] == c0 + j);
  if (c0 < 0 || c0 >= bed->m || (contig && c0 + m > bed->m))
    fail("Tested %lld < %lld. Subscript out of bounds (ind.col).", (long long)(c0 + m - 1),
         (long long)bed->m);
  op->cols_contig = contig;
  op->col0 = contig ? c0 : 0;
  if (!contig) {
    auto c = to_i32(ind_col, m, bed->m, "ind.col");
    int64_t m_pad = bsn::round_up(m, 64);
    c.resize((size_t)m_pad, c[0]);
    copy_h2d(bed, op->d_cols.ensure((size_t)m_pad), c.data(), (size_t)m_pad * 4);
  }
  // missing-value knowledge (BSN_FORCE_NA_PLANE=1 keeps the general kernels, for A/B tests)
  op->no_na = false;
  if ((int64_t)bed->na_cnt.size() == bed->m && !getenv("BSN_FORCE_NA_PLANE")) {
    bool all0 = true;
    for (int64_t j = 0; all0 && j < m; j++) all0 = bed->na_cnt[(size_t)(ind_col ? ind_col[j] : j)] == 0;
    op->no_na = all0;
  }
  if (defer_scale) {  // written on the device by the first crossproduct pass (matvec.hip)
    op->d_center.ensure((size_t)m);
    op->d_scale.ensure((size_t)m);
    return;
  }
  // centre / scale (defaults 0 / 1, R/bed-mult-vec.R:23-24)
  op->d_center.ensure((size_t)m);
  op->d_scale.ensure((size_t)m);
  if (center) copy_h2d(bed, op->d_center.p, center, (size_t)m * 8);
  else BSN_HIP(hipMemsetAsync(op->d_center.p, 0, (size_t)m * 8, bed->stream));
  if (scale) {
    copy_h2d(bed, op->d_scale.p, scale, (size_t)m * 8);
  } else {
    hipLaunchKernelGGL(k_fill_f64, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, bed->stream, op->d_scale.p, m, 1.0);
    BSN_HIP(hipGetLastError());
  }
}

// counts of the codes 0, 1, 2, missing of every selected variant over the selected rows, 4 x m int32 on
// the device
static void counts_device(bsn_op *op, const int64_t *ind_row, int64_t n, int32_t *d_counts) {
  bsn_bed *bed = op->bed;
  if (op->rows_identity) {
    counts_all_rows(bed, op->cols_contig ? nullptr : op->d_cols.p, op->col0, op->m, d_counts);
  } else {
    std::vector<double> w((size_t)bed->n, 0.0);
    for (int64_t i = 0; i < n; i++) w[(size_t)(ind_row ? ind_row[i] : i)] += 1.0;
    DevBuf<double> d_w;
    copy_h2d(bed, d_w.ensure((size_t)bed->n), w.data(), (size_t)bed->n * 8);
    counts_weighted(op, d_w.p, n, d_counts);
    BSN_HIP(hipStreamSynchronize(bed->stream));  // d_w is released on return
  }
}

// the same into a host 4 x m int32 array
void require_resident(const bsn_bed *b, const char *what) {
  if (b->streamed())
    fail("%s needs the genotype image resident on the device: this handle streams its file (%.1f GB do not fit the device "
         "memory that was free when it was opened); bed_prodVec, bed_cprodVec, counts / colstats / MAF / scaleBinom and the "
         "`[` accessor are served out of core",
         what, (double)b->m * (double)b->pitch / 1e9);
}

// ---- out-of-core handles: the resident slab image --------------------------------------------------------------------
// One image of slab_cols variants lives on the handle (with its page-locked staging buffers) and is filled with one
// slab of the file at a time.  It works on the HANDLE's stream: whatever the caller queues there — the kernels of a
// one-shot entry point, the passes of a solve (svd.hip) — is ordered with the uploads without further ado.
bsn_bed *slab_image(bsn_bed *b) {
  BSN_HIP(hipSetDevice(b->device));
  if (!b->slab_img) {
    std::unique_ptr<bsn_bed, void (*)(bsn_bed *)> im(new bsn_bed(), bed_free);
    image_alloc(im.get(), b->n, b->slab_cols);
    (void)hipStreamDestroy(im->stream);
    im->stream = b->stream;
    im->stream_borrowed = true;
    b->slab_img = im.release();
  }
  if (!b->slab_stage) b->slab_stage = new FileStage();
  return b->slab_img;
}
int64_t slab_count(const bsn_bed *b) { return (b->m + b->slab_cols - 1) / b->slab_cols; }
// slab sl of the file into the slab image (pread into pinned buffers, double-buffered against the DMA, recode, zero
// pad rows); returns its number of variants, *j0 = its first variant
void slab_upload_range(bsn_bed *b, int64_t j0, int64_t cnt) {
  bsn_bed *img = slab_image(b);
  if (j0 < 0 || cnt <= 0 || j0 + cnt > b->m || cnt > b->slab_cols)
    fail("internal: variants %lld .. %lld into a slab image of %lld", (long long)j0, (long long)(j0 + cnt), (long long)b->slab_cols);
  img->m = cnt;
  img->na_cnt.clear();
  img->counts_cache.clear();
  image_from_file(img, b->fd_file, 3 + j0 * b->n_byte, b->n_byte, (FileStage *)b->slab_stage);
}
int64_t slab_upload(bsn_bed *b, int64_t sl, int64_t *j0_out) {
  const int64_t j0 = sl * b->slab_cols, cnt = std::min(b->slab_cols, b->m - j0);
  if (cnt <= 0) fail("internal: slab %lld of %lld", (long long)sl, (long long)slab_count(b));
  slab_upload_range(b, j0, cnt);
  if (j0_out) *j0_out = j0;
  return cnt;
}

// ---- out-of-core handles: the one-shot entry points over slabs of variants --------------------------------------
// The selected variants are grouped by slab (order inside a slab = order in ind_col); `f(slab image, positions in
// ind_col, local variant indices)` runs per non-empty slab on a resident image of that slab, in ascending slab order.
namespace {
struct SlabWalk {
  bsn_bed *bed;
  bsn_bed *img;        // the resident slab image: kept on the handle between calls, like its staging buffers
  FileStage *stage;
  SlabWalk(bsn_bed *b) : bed(b) {
    img = slab_image(b);
    stage = (FileStage *)b->slab_stage;
  }
  template <class F>
  void run(const int64_t *ind_col, int64_t m, F f) {
    const int64_t nslab = (bed->m + bed->slab_cols - 1) / bed->slab_cols;
    std::vector<std::vector<int64_t>> pos((size_t)nslab);
    for (int64_t p = 0; p < m; p++) {
      const int64_t c = ind_col ? ind_col[p] : p;
      if (c < 0 || c >= bed->m) fail("Tested %lld < %lld. Subscript out of bounds (ind.col).", (long long)c, (long long)bed->m);
      pos[(size_t)(c / bed->slab_cols)].push_back(p);
    }
    std::vector<int64_t> local;
    for (int64_t sl = 0; sl < nslab; sl++) {
      const std::vector<int64_t> &P = pos[(size_t)sl];
      if (P.empty()) continue;
      const int64_t j0 = sl * bed->slab_cols;
      slab_upload(bed, sl);
      local.resize(P.size());
      for (size_t k = 0; k < P.size(); k++) local[k] = (ind_col ? ind_col[P[k]] : P[k]) - j0;
      f(img, P, local);
    }
  }
};
}  // namespace

void counts_host(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                 int64_t m, int32_t *res) {
  if (bed->streamed()) {
    SlabWalk W(bed);
    std::vector<int32_t> part;
    W.run(ind_col, m, [&](bsn_bed *sb, const std::vector<int64_t> &P, const std::vector<int64_t> &local) {
      part.resize(4 * P.size());
      counts_host(sb, ind_row, n, local.data(), (int64_t)local.size(), part.data());
      for (size_t k = 0; k < P.size(); k++) std::memcpy(res + 4 * P[k], part.data() + 4 * k, 16);
    });
    return;
  }
  // the counts of this row selection as an earlier call left them (BSN_NO_COUNTS_CACHE=1: always count)
  auto &cc = bed->counts_cache;
  const bool use_cache = n > 0 && m > 0 && !getenv("BSN_NO_COUNTS_CACHE");
  bool rows_all = use_cache && n == bed->n;
  for (int64_t i = 0; rows_all && ind_row && i < n; i++) rows_all = ind_row[i] == i;
  std::vector<int64_t> lead;         // (a NULL list with n < the handle's samples: the leading n)
  const int64_t *rows = ind_row;
  if (use_cache && !rows_all && !ind_row) {
    lead.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) lead[(size_t)i] = i;
    rows = lead.data();
  }
  const bool same_rows = use_cache && (int64_t)cc.cnt.size() == 4 * bed->m &&
                         (rows_all ? cc.rows_all
                                   : (!cc.rows_all && (int64_t)cc.rows.size() == n && std::memcmp(cc.rows.data(), rows, (size_t)n * 8) == 0));
  if (same_rows) {
    bool known = true;
    for (int64_t j = 0; known && j < m; j++) {
      const int64_t c = ind_col ? ind_col[j] : j;
      known = c >= 0 && c < bed->m && cc.cnt[(size_t)4 * c] >= 0;
    }
    if (known) {
      for (int64_t j = 0; j < m; j++) std::memcpy(res + 4 * j, &cc.cnt[(size_t)4 * (ind_col ? ind_col[j] : j)], 16);
      return;
    }
  }
  bsn_op op;
  fill_op(&op, bed, ind_row, n, ind_col, m, nullptr, nullptr, true);
  DevBuf<int32_t> d_counts;
  d_counts.ensure((size_t)4 * m);
  counts_device(&op, ind_row, n, d_counts.p);
  copy_d2h(bed, res, d_counts.p, (size_t)4 * m * 4);
  if (op.rows_identity) {  // remember which variants are complete
    if ((int64_t)bed->na_cnt.size() != bed->m) bed->na_cnt.assign((size_t)bed->m, -1);
    for (int64_t j = 0; j < m; j++) bed->na_cnt[(size_t)(ind_col ? ind_col[j] : j)] = res[4 * j + 3];
  }
  if (use_cache) {
    if (!same_rows) {
      cc.clear();
      cc.rows_all = rows_all;
      if (!rows_all) cc.rows.assign(rows, rows + n);
      cc.cnt.assign((size_t)4 * bed->m, -1);
    }
    for (int64_t j = 0; j < m; j++) std::memcpy(&cc.cnt[(size_t)4 * (ind_col ? ind_col[j] : j)], res + 4 * j, 16);
  }
}

// ---- multLinReg on the device ----------------------------------------------------------------
// X = [U, U^2] (n x 2K) and the column totals sum_i y, sum_i y^2 (one workgroup per column, fixed order)
__global__ __launch_bounds__(1024) void k_mlr_prepare(const double *U, int64_t n, int K, double *X, double *tot) {
  const int k = blockIdx.x;
  double s1 = 0, s2 = 0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const double y = U[i + (int64_t)k * n], yy = y * y;
    X[i + (int64_t)k * n] = y;
    X[i + (int64_t)(K + k) * n] = yy;
    s1 += y;
    s2 += yy;
  }
  for (int off = 32; off > 0; off >>= 1) {
    s1 += __shfl_down(s1, off);
    s2 += __shfl_down(s2, off);
  }
  __shared__ double sm[2][16];
  if ((threadIdx.x & 63) == 0) sm[0][threadIdx.x >> 6] = s1, sm[1][threadIdx.x >> 6] = s2;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; w++) s1 += sm[0][w], s2 += sm[1][w];
    tot[k] = s1;
    tot[K + k] = s2;
  }
}

// src/multLinReg.cpp:36-57 per (variant, column) from the plane sums: P = sum g y over the non-missing,
// Q = sum of y (columns < K) / y^2 (columns >= K) over the MISSING samples, counts = code counts
#pragma clang fp contract(off)
__global__ void k_mlr_final(const double *P, const double *Q, int64_t ld, const int32_t *counts, const double *tot,
                            int64_t m, int K, int64_t n, bool has_q, double *res) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (j >= m) return;
  const int4 c = *(const int4 *)(counts + 4 * j);
  const int nona = (int)(n - c.w);
  const double xSum = (double)c.y + 2.0 * c.z, xxSum = (double)c.y + 4.0 * c.z;
  const double deno_x = xxSum - xSum * xSum / nona;
  const double xySum = P[j + (int64_t)k * ld];
  const double ySum = tot[k] - (has_q ? Q[j + (int64_t)k * ld] : 0.0);
  const double yySum = tot[K + k] - (has_q ? Q[j + (int64_t)(K + k) * ld] : 0.0);
  const double num = xySum - xSum * ySum / nona;
  const double deno_y = yySum - ySum * ySum / nona;
  const double deno = deno_x * deno_y - num * num;
  res[j + (int64_t)k * m] = (deno == 0 || nona < 2) ? __longlong_as_double(0x7ff8000000000000LL)
                                                    : num * sqrt((nona - 2) / deno);
}
#pragma clang fp contract(on)

}  // namespace bsn

using namespace bsn;

extern "C" {

const char *bsn_last_error(void) { return g_err.c_str(); }
int bsn_version(void) { return 100; }

int bsn_device_count(int *count) {
  return guarded([&] {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
  });
}

int bsn_set_device(int device) {
  return guarded([&] {
    require_gpu();
    BSN_HIP(hipSetDevice(device));
  });
}

int bsn_selftest(void) {
  return guarded([&] {
    require_gpu();
    selftest();
  });
}

int bsn_bed_from_host(const uint8_t *payload, int64_t n, int64_t m, int64_t n_byte, bsn_bed **out) {
  return guarded([&] {
    require_gpu();
    if (n_byte < (n + 3) / 4) fail("n or p does not match the dimensions of the file.");
    std::unique_ptr<bsn_bed, void (*)(bsn_bed *)> b(new bsn_bed(), free_bed);
    image_alloc(b.get(), n, m);
    image_from_host(b.get(), payload, n_byte);
    *out = b.release();
  });
}

int bsn_bed_open(const char *path, int64_t n, int64_t m, bsn_bed **out) {
  return guarded([&] {
    // validation order and messages of src/bed-acc-xptr.cpp:14-35 (the reference maps the file;
    // here it is read once into the device image, so the header is checked with a 3-byte read)
    int fd = open(path, O_RDONLY);
    if (fd < 0) fail("Error when mapping file:\n  %s.\n", strerror(errno));
    struct Close {
      int fd;
      ~Close() { close(fd); }
    } closer{fd};
    struct stat st;
    if (fstat(fd, &st) != 0) fail("Error when mapping file:\n  %s.\n", strerror(errno));
    const size_t size = (size_t)st.st_size;
    uint8_t f[3] = {0, 0, 0};
    if (size == 0) fail("Error when mapping file:\n  %s.\n", strerror(EINVAL));  // mmap of an empty file
    const ssize_t got = pread(fd, f, 3, 0);
    if (got < 0) fail("Error when mapping file:\n  %s.\n", strerror(errno));
    if (size < 3 || !(f[0] == 0x6C && f[1] == 0x1B)) fail("File is not a binary PED file.");
    if (f[2] != 0x01) fail("Variant-major is the only mode supported.");
    int64_t n_byte = (n + 3) / 4;
    if (n <= 0 || m <= 0 || (size_t)(3 + n_byte * m) != size)
      fail("n or p does not match the dimensions of the file.");
    require_gpu();
    std::unique_ptr<bsn_bed, void (*)(bsn_bed *)> b(new bsn_bed(), free_bed);
    // does the image fit?  free device memory less 2 GB of working room, or BSN_IMAGE_BUDGET (bytes)
    const int64_t pitch = round_up(n_byte, 256);
    const double image_bytes = (double)(m + 64) * (double)pitch;
    size_t free_b = 0, total_b = 0;
    BSN_HIP(hipMemGetInfo(&free_b, &total_b));
    double budget = (double)(free_b + dev_cache_held()) - 2e9;
    const char *forced = getenv("BSN_IMAGE_BUDGET");
    if (forced) budget = atof(forced);
    // The resident image is tried first whenever it could fit at all: a device that other handles have filled to
    // within the 2 GB of working room must not turn a small file into a streamed handle, which most entry points
    // refuse.  Only an allocation that really fails — or a file beyond the budget — goes out of core.
    bool resident = image_bytes <= budget;
    if (!resident && !forced && image_bytes <= (double)(free_b + dev_cache_held())) resident = true;
    if (resident) {
      bool ok = true;
      try {
        image_alloc(b.get(), n, m);
      } catch (const std::exception &) {
        if (forced || image_bytes <= budget - 8e9) throw;   // (not a matter of room)
        (void)hipGetLastError();
        ok = false;
        b.reset(new bsn_bed());
      }
      if (ok) {
        image_from_file(b.get(), fd, 3, n_byte);
        *out = b.release();
        return;
      }
    }
    // out-of-core: keep the file mapped, walk it in slabs (bsn_internal.hpp)
    void *map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map == MAP_FAILED) fail("Error when mapping file:\n  %s.\n", strerror(errno));
    image_alloc(b.get(), n, 1);   // stream, events, geometry; no variants resident
    (void)hipFree(b->d_img);
    b->d_img = nullptr;
    b->m = m;
    b->map_base = map;
    b->map_len = size;
    b->h_map = (const uint8_t *)map + 3;
    b->fd_file = dup(fd);
    if (b->fd_file < 0) fail("Error when mapping file:\n  %s.\n", strerror(errno));
    // one resident slab image, kept on the handle between calls (bsn_bed_release_workspace frees it): a slab of at most
    // 16 GB — PCIe paces the walk, a larger slab buys nothing and would pin the device memory other handles need
    b->slab_cols = std::max<int64_t>(64, (int64_t)(std::min(std::max(budget, 0.0), 16e9) / (double)pitch) / 64 * 64 - 64);
    if (b->slab_cols > m) b->slab_cols = round_up(m, 64);
    if (getenv("BSN_VERBOSE"))
      std::fprintf(stderr, "[bsn] %s: image of %.1f GB does not fit (%.1f GB available): out-of-core handle, slabs of %lld variants\n",
                   path, image_bytes / 1e9, budget / 1e9, (long long)b->slab_cols);
    *out = b.release();
  });
}

int bsn_bed_is_streamed(const bsn_bed *bed) { return bed->streamed() ? 1 : 0; }

// FBM.code256 -> device image.  The 256 decoded values decide the image kind:
//   all of them in {0, 1, 2, NA}          -> 2-bit image (CODE_012, CODE_IMPUTE_PRED, ...): every entry point
//   on a grid v_off + v_step k, |k| <= 127 -> byte image (CODE_DOSAGE: 0.00 .. 2.00 by 0.01): colstats and
//                                             the products (and what is built on them: PRS, SVD)
//   anything else                          -> refused (there is no exact integer image of it)
static void fbm_open(const uint8_t *bytes, int64_t n, int64_t m, int64_t ld, const double *code256, bsn_bed **out) {
  require_gpu();
  if (ld < n) fail("Incompatibility between dimensions.");
  bool calls = true;
  double vmin = 0, vmax = 0;
  int nval = 0;
  for (int c = 0; c < 256; c++) {
    const double v = code256[c];
    if (std::isnan(v)) continue;
    if (!(v == 0.0 || v == 1.0 || v == 2.0)) calls = false;
    if (nval == 0 || v < vmin) vmin = v;
    if (nval == 0 || v > vmax) vmax = v;
    nval++;
  }
  if (nval == 0) fail("'code256' holds no value");
  uint8_t lut[256];
  std::unique_ptr<bsn_bed, void (*)(bsn_bed *)> b(new bsn_bed(), free_bed);
  if (calls) {
    for (int c = 0; c < 256; c++) lut[c] = std::isnan(code256[c]) ? 3 : (uint8_t)code256[c];
    image_alloc(b.get(), n, m, 2);
    image_from_fbm(b.get(), bytes, ld, lut);
    // FBMs of imputed data have no missing value: one count pass settles it for every later operator
    std::vector<int32_t> cnt((size_t)4 * m);
    counts_host(b.get(), nullptr, n, nullptr, m, cnt.data());
  } else {
    // smallest positive difference between two values as the step, mid-range as the offset; the grid is then
    // SNAPPED to the range (step = range / number of steps): the raw difference of two rounded decimals, e.g.
    // 0.00999999999999979 for CODE_DOSAGE, would bias every decoded value by up to 4e-14
    double step = 0;
    for (int a = 0; a < 256; a++)
      for (int c = 0; c < 256; c++) {
        const double d = code256[a] - code256[c];
        if (d > 1e-12 * (std::fabs(vmax) + std::fabs(vmin) + 1) && (step == 0 || d < step)) step = d;
      }
    double off = vmin;
    bool ok = step > 0 && (vmax - vmin) / step <= 254.5;
    if (ok) {
      const double nsteps = std::round((vmax - vmin) / step);
      step = (vmax - vmin) / nsteps;
      off = vmin + std::round(nsteps / 2.0) * step;
    }
    for (int c = 0; c < 256 && ok; c++) {
      if (std::isnan(code256[c])) {
        lut[c] = 0x80;
        continue;
      }
      // values like 0.07 are not exactly k * 0.01 in binary: the decoded value must reproduce the table entry to
      // a few units in the last place of the table's range
      const double kr = std::round((code256[c] - off) / step);
      // 16 units in the last place of the table's range: off + step * kr carries three roundings and the table was
      // written by other arithmetic (i * 0.01, seq(0, 2, by = 0.01), a decimal parser ...) — one ulp of slack (round 3)
      // sent such tables to the slow generic image for nothing; an entry really off the grid misses by ~step
      if (kr < -127 || kr > 127 ||
          std::fabs(off + step * kr - code256[c]) > 16 * 2.220446049250313e-16 * (std::fabs(vmax) + std::fabs(vmin) + step))
        ok = false;
      lut[c] = (uint8_t)(int8_t)kr;
    }
    if (!ok) {
      // any other table: the FBM's own bytes + the table, served by the fp64 look-up kernels only
      if (getenv("BSN_VERBOSE"))
        std::fprintf(stderr, "[bsn] code256 is neither calls nor a regular grid of <= 255 steps: generic look-up image "
                             "(colstats / prodVec / cprodVec only)\n");
      for (int c = 0; c < 256; c++) lut[c] = (uint8_t)c;
      image_alloc(b.get(), n, m, 8);
      b->generic = true;
      BSN_HIP(hipMalloc((void **)&b->d_lut, 256 * sizeof(double)));
      BSN_HIP(hipMemcpy(b->d_lut, code256, 256 * sizeof(double), hipMemcpyHostToDevice));
      image_from_fbm(b.get(), bytes, ld, lut);
      // missing values: which codes are NaN, counted per variant on the host from the caller's bytes would
      // cost a pass; the look-up kernels propagate NaN on their own, so only "some / none" is recorded
      bool any_nan = false;
      for (int c = 0; c < 256; c++) any_nan = any_nan || std::isnan(code256[c]);
      if (!any_nan) b->na_cnt.assign((size_t)m, 0);
      *out = b.release();
      return;
    }
    image_alloc(b.get(), n, m, 8);
    b->v_off = off;
    b->v_step = step;
    image_from_fbm(b.get(), bytes, ld, lut);
    // which variants are complete (no missing value): the products then skip the missing-value plane
    DevBuf<long long> d_st;
    stats8(b.get(), nullptr, n, nullptr, 0, m, d_st.ensure((size_t)3 * m));
    std::vector<long long> st((size_t)3 * m);
    copy_d2h(b.get(), st.data(), d_st.p, st.size() * 8);
    b->na_cnt.resize((size_t)m);
    for (int64_t j = 0; j < m; j++) b->na_cnt[(size_t)j] = (int32_t)st[(size_t)(3 * j + 2)];
  }
  *out = b.release();
}

int bsn_bed_from_fbm(const uint8_t *bytes, int64_t n, int64_t m, int64_t ld, bsn_bed **out) {
  return guarded([&] {
    double code[256];
    for (int c = 0; c < 256; c++) code[c] = c < 3 ? (double)c : std::numeric_limits<double>::quiet_NaN();
    fbm_open(bytes, n, m, ld, code, out);  // CODE_012, R/bigSNP-class.R:7
  });
}

int bsn_fbm_open(const uint8_t *bytes, int64_t n, int64_t m, int64_t ld, const double *code256, bsn_bed **out) {
  return guarded([&] { fbm_open(bytes, n, m, ld, code256, out); });
}

int bsn_bed_bits(const bsn_bed *bed) { return bed->bits; }

int64_t bsn_bed_na_known(const bsn_bed *bed) {
  if ((int64_t)bed->na_cnt.size() != bed->m) return -1;
  int64_t tot = 0;
  for (int32_t c : bed->na_cnt) {
    if (c < 0) return -1;
    tot += c;
  }
  return tot;
}

int bsn_bed_synthetic(int64_t n, int64_t m, uint32_t seed, uint32_t npop, uint32_t na16,
                      int64_t j_begin, bsn_bed **out) {
  return guarded([&] {
    require_gpu();
    std::unique_ptr<bsn_bed, void (*)(bsn_bed *)> b(new bsn_bed(), free_bed);
    image_alloc(b.get(), n, m);
    image_generate(b.get(), seed, npop ? npop : 1, na16, j_begin);
    if (na16 == 0) b->na_cnt.assign((size_t)m, 0);  // the generator draws no missing value then
    *out = b.release();
  });
}

int bsn_bed_close(bsn_bed *bed) {
  return guarded([&] {
    if (bed) {
      (void)hipSetDevice(bed->device);
      free_bed(bed);
    }
  });
}

int bsn_bed_tile(bsn_bed *bed, int *built) {
  return guarded([&] {
    BSN_HIP(hipSetDevice(bed->device));
    const bool ok = image_tile(bed);
    if (built) *built = ok ? 1 : 0;
  });
}

// Names (demangled, without the argument list: rocprofv3's Kernel_Name up to the parenthesis) of the streaming
// kernels the LAST profiled solve on this handle launched, one "kind=name" per line for the kinds cprod / prod /
// cprod_stats / warm.  Lets a measurement record be tied to the kernels of the running build.
int bsn_bed_streaming_kernels(bsn_bed *bed, char *buf, int64_t len) {
  return guarded([&] {
    if (!buf || len < 1) fail("bsn_bed_streaming_kernels: no buffer");
    buf[0] = 0;
    if (bed->last_solve_on_sub && bed->sub) bed = bed->sub;   // the last solve ran on the compacted copy: its operator lives there
    if (!bed->svd_op) return;
    static const char *kinds[kProfKinds] = {"cprod", "prod", "cprod_stats", "warm", "cprod_wide", "prod_wide"};
    std::string out;
    for (int k = 0; k < kProfKinds; k++) {
      const void *fn = bed->svd_op->prof_kernel[k];
      if (!fn) continue;
      const char *mangled = hipKernelNameRefByPtr(fn, bed->stream);
      if (!mangled) continue;
      int status = 1;
      char *dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
      std::string name = (status == 0 && dem) ? dem : mangled;
      free(dem);
      const size_t par = name.find('(');
      if (par != std::string::npos) name.resize(par);
      out += std::string(kinds[k]) + "=" + name + "\n";
    }
    snprintf(buf, (size_t)len, "%s", out.c_str());
  });
}

int bsn_bed_sample_major(bsn_bed *bed, int *built) {
  return guarded([&] {
    BSN_HIP(hipSetDevice(bed->device));
    const bool ok = image_smaj(bed);
    if (built) *built = ok ? 1 : 0;
  });
}

int bsn_bed_release_workspace(bsn_bed *bed) {
  return guarded([&] {
    BSN_HIP(hipSetDevice(bed->device));
    BSN_HIP(hipStreamSynchronize(bed->stream));
    bed->svd_op.reset();
    bed->svd_ws.reset();
    if (bed->d_tiled) {  // the streaming-layout copy is rebuilt by the next solve if there is room again
      BSN_HIP(hipFree(bed->d_tiled));
      bed->d_tiled = nullptr;
    }
    bed->tiled_tried = false;
    image_smaj_wait(bed);   // (a build in flight is adopted first, then freed with the rest)
    if (bed->d_smaj) {  // likewise the sample-major copy
      BSN_HIP(hipFree(bed->d_smaj));
      bed->d_smaj = nullptr;
    }
    bed->smaj_tried = false;
    bed->smaj_cap = 0;
    if (bed->sub) {     // the compacted sub-image of the last solve over a column list (and everything IT holds)
      bed_free(bed->sub);
      bed->sub = nullptr;
      bed->sub_key = 0;
      bed->sub_cols.clear();
      bed->sub_rows.clear();
      bed->last_solve_on_sub = false;
    }
    if (bed->slab_img) {   // out-of-core handle: the resident slab image and its page-locked staging buffers
      bed_free(bed->slab_img);
      bed->slab_img = nullptr;
    }
    if (bed->slab_stage) {
      delete (bsn::FileStage *)bed->slab_stage;
      bed->slab_stage = nullptr;
    }
    dev_cache_flush();
  });
}

int64_t bsn_bed_nrow(const bsn_bed *bed) { return bed->n; }
int64_t bsn_bed_ncol(const bsn_bed *bed) { return bed->m; }
int64_t bsn_bed_bytes(const bsn_bed *bed) { return bed->pitch * bed->m; }

int bsn_bed_download(bsn_bed *bed, uint8_t *payload_out) {
  return guarded([&] {
    BSN_HIP(hipSetDevice(bed->device));
    if (bed->streamed()) {   // the payload is the mapped file itself
      std::memcpy(payload_out, bed->h_map, (size_t)bed->n_byte * (size_t)bed->m);
      return;
    }
    image_download(bed, payload_out);
  });
}

// ---- operator -------------------------------------------------------------------
int bsn_op_create(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                  int64_t m, const double *center, const double *scale, bsn_op **out) {
  return guarded([&] {
    std::unique_ptr<bsn_op> op(new bsn_op());
    fill_op(op.get(), bed, ind_row, n, ind_col, m, center, scale);
    *out = op.release();
  });
}

int bsn_op_destroy(bsn_op *op) {
  return guarded([&] {
    if (op) {
      (void)hipSetDevice(op->bed->device);
      delete op;
    }
  });
}

int bsn_op_set_slices(bsn_op *op, int slices) {
  return guarded([&] {
    if (slices < 1 || slices > 7) fail("slices must be in 1..7");
    op->slices = slices;
  });
}

int bsn_op_prod(bsn_op *op, const double *d_X, int64_t ldx, int nvec, double *d_Y, int64_t ldy) {
  return guarded([&] {
    BSN_HIP(hipSetDevice(op->bed->device));
    op_prod(op, d_X, ldx, nvec, d_Y, ldy);
  });
}

int bsn_op_cprod(bsn_op *op, const double *d_X, int64_t ldx, int nvec, double *d_Z, int64_t ldz) {
  return guarded([&] {
    BSN_HIP(hipSetDevice(op->bed->device));
    op_cprod(op, d_X, ldx, nvec, d_Z, ldz);
  });
}

int bsn_op_sync(bsn_op *op) {
  return guarded([&] { BSN_HIP(hipStreamSynchronize(op->bed->stream)); });
}

// ---- .Call replacements ------------------------------------------------------------
static void matvec_host(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                        int64_t m, const double *center, const double *scale, const double *x,
                        double *out, bool transpose, bsn_comm *comm = nullptr) {
  if (bed->streamed()) {
    // out of core: A~' x slab by slab is the result slab by slab; A~ x is the sum of the slabs' products (added on the
    // host in ascending slab order: deterministic)
    if (comm) fail("sharded products are not available for an out-of-core handle");
    SlabWalk W(bed);
    std::vector<double> ce, sc, xs, part;
    if (!transpose) std::fill(out, out + n, 0.0);
    W.run(ind_col, m, [&](bsn_bed *sb, const std::vector<int64_t> &P, const std::vector<int64_t> &local) {
      const int64_t k = (int64_t)P.size();
      if (center) { ce.resize((size_t)k); for (int64_t t = 0; t < k; t++) ce[(size_t)t] = center[P[(size_t)t]]; }
      if (scale) { sc.resize((size_t)k); for (int64_t t = 0; t < k; t++) sc[(size_t)t] = scale[P[(size_t)t]]; }
      if (transpose) {
        part.resize((size_t)k);
        matvec_host(sb, ind_row, n, local.data(), k, center ? ce.data() : nullptr, scale ? sc.data() : nullptr, x,
                    part.data(), true);
        BSN_HIP(hipStreamSynchronize(sb->stream));
        for (int64_t t = 0; t < k; t++) out[P[(size_t)t]] = part[(size_t)t];
      } else {
        xs.resize((size_t)k);
        for (int64_t t = 0; t < k; t++) xs[(size_t)t] = x[P[(size_t)t]];
        part.resize((size_t)n);
        matvec_host(sb, ind_row, n, local.data(), k, center ? ce.data() : nullptr, scale ? sc.data() : nullptr, xs.data(),
                    part.data(), false);
        BSN_HIP(hipStreamSynchronize(sb->stream));
        for (int64_t i = 0; i < n; i++) out[i] += part[(size_t)i];
      }
    });
    return;
  }
  bsn_op op;
  // A~' x needs centre / scale only in its finalize kernel: their upload is queued on the second stream and runs
  // beside the streaming kernel (2-bit image; the byte and look-up kernels read them earlier)
  const bool late = transpose && bed->bits == 2 && !bed->generic && (center || scale);
  fill_op(&op, bed, ind_row, n, ind_col, m, center, scale, late);
  if (late) {   // (fill_op has only allocated centre / scale; a default is written here, on the main stream)
    op.late_center = center;
    op.late_scale = scale;
    if (!center) BSN_HIP(hipMemsetAsync(op.d_center.p, 0, (size_t)m * 8, bed->stream));
    if (!scale) {
      hipLaunchKernelGGL(k_fill_f64, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, bed->stream, op.d_scale.p, m, 1.0);
      BSN_HIP(hipGetLastError());
    }
  }
  op.slices = 7;  // 56-bit fixed point: fp64-grade for a single vector, still one MFMA column block
  int64_t nin = transpose ? n : m, nout = transpose ? m : n;
  DevBuf<double> d_in, d_out;
  copy_h2d(bed, d_in.ensure((size_t)nin), x, (size_t)nin * 8);
  d_out.ensure((size_t)nout);
  if (bed->generic) {   // arbitrary decode table: fp64 look-up kernels
    const int32_t *rows = op.rows_identity ? nullptr : op.d_rows.p, *cols = op.cols_contig ? nullptr : op.d_cols.p;
    if (comm) fail("sharded products are not available for a generic decode table");
    if (transpose)
      lut_cprod(bed, rows, n, cols, op.col0, m, center ? op.d_center.p : nullptr, scale ? op.d_scale.p : nullptr,
                d_in.p, d_out.p);
    else
      lut_prod(bed, rows, n, cols, op.col0, m, center ? op.d_center.p : nullptr, scale ? op.d_scale.p : nullptr,
               d_in.p, d_out.p);
    copy_d2h(bed, out, d_out.p, (size_t)nout * 8);
    return;
  }
  if (transpose)
    op_cprod(&op, d_in.p, nin, 1, d_out.p, nout);
  else
    op_prod(&op, d_in.p, nin, 1, d_out.p, nout);
  if (comm) {  // column shards: sum of the partial products over the ranks, on the device
    if (transpose) fail("internal: the crossproduct of a column shard needs no exchange");
    if (comm->device != bed->device) fail("the communicator was created on another device");
    comm_allreduce_sum(comm, d_out.p, nout, bed->stream);
  }
  copy_d2h(bed, out, d_out.p, (size_t)nout * 8);
}

int bsn_bed_prodvec(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                    int64_t m, const double *center, const double *scale, const double *x,
                    double *y) {
  return guarded([&] { matvec_host(bed, ind_row, n, ind_col, m, center, scale, x, y, false); });
}

int bsn_bed_prodvec_sharded(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                            int64_t m, const double *center, const double *scale, const double *x,
                            bsn_comm *comm, double *y) {
  return guarded([&] { matvec_host(bed, ind_row, n, ind_col, m, center, scale, x, y, false, comm); });
}

int bsn_bed_cprodvec(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                     int64_t m, const double *center, const double *scale, const double *x,
                     double *z) {
  return guarded([&] { matvec_host(bed, ind_row, n, ind_col, m, center, scale, x, z, true); });
}

int bsn_bed_col_counts(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                       int64_t m, int32_t *res) {
  return guarded([&] { counts_host(bed, ind_row, n, ind_col, m, res); });
}

int bsn_bed_row_counts(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col, int64_t m,
                       int32_t *res) {
  return guarded([&] {
    bsn_op op;
    fill_op(&op, bed, ind_row, n, ind_col, m, nullptr, nullptr);
    DevBuf<double> d_c;
    d_c.ensure((size_t)3 * n);
    op_row_counts(&op, d_c.p);
    std::vector<double> c((size_t)3 * n);
    copy_d2h(bed, c.data(), d_c.p, (size_t)3 * n * 8);
    for (int64_t i = 0; i < n; i++) {
      const int64_t n2 = (int64_t)c[(size_t)i], n1 = (int64_t)c[(size_t)(n + i)], na = (int64_t)c[(size_t)(2 * n + i)];
      res[4 * i + 0] = (int32_t)(m - n1 - n2 - na);
      res[4 * i + 1] = (int32_t)n1;
      res[4 * i + 2] = (int32_t)n2;
      res[4 * i + 3] = (int32_t)na;
    }
  });
}

int bsn_bed_colstats(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                     int64_t m, double *sumX, double *denoX, int32_t *nb_nona_col,
                     int32_t *n_bad) {
  return guarded([&] {
    // integer-exact restatement of src/bed-fun.cpp:22-38 from the code counts:
    // xSum = n1 + 2 n2, xxSum = n1 + 4 n2 (both exact in fp64), c = n - nNA
    std::vector<int32_t> cnt((size_t)4 * m);
    counts_host(bed, ind_row, n, ind_col, m, cnt.data());
    int32_t bad = 0;
    for (int64_t j = 0; j < m; j++) {
      const int32_t *c = &cnt[(size_t)4 * j];
      double xSum = (double)c[1] + 2.0 * c[2], xxSum = (double)c[1] + 4.0 * c[2];
      int32_t nona = (int32_t)(n - c[3]);
      sumX[j] = xSum;
      denoX[j] = xxSum - xSum * xSum / nona;
      nb_nona_col[j] = nona;
      if (2 * (int64_t)nona < n) bad++;
    }
    if (n_bad) *n_bad = bad;
  });
}

int bsn_snp_colstats(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                     int64_t m, double *sumX, double *denoX) {
  return guarded([&] {
    // src/colstats.cpp:22-32: no NA handling, denominator n.  The accessor decodes a missing code to
    // NA_real (code256 is passed as is, :14), which poisons the sums of its column: NaN here.
    const double qnan = std::numeric_limits<double>::quiet_NaN();
    if (bed->generic) {   // arbitrary decode table: sum v and sum v^2 by look-up, src/colstats.cpp:22-32 in fp64
      bsn_op op;
      fill_op(&op, bed, ind_row, n, ind_col, m, nullptr, nullptr);
      DevBuf<double> d_st;
      lut_colstats(bed, op.rows_identity ? nullptr : op.d_rows.p, n, op.cols_contig ? nullptr : op.d_cols.p, op.col0, m,
                   d_st.ensure((size_t)2 * m));
      std::vector<double> st((size_t)2 * m);
      copy_d2h(bed, st.data(), d_st.p, st.size() * 8);
      for (int64_t j = 0; j < m; j++) {
        sumX[j] = st[(size_t)(2 * j)];
        denoX[j] = st[(size_t)(2 * j + 1)] - st[(size_t)(2 * j)] * st[(size_t)(2 * j)] / n;
      }
      return;
    }
    if (bed->bits == 8) {
      bsn_op op;
      fill_op(&op, bed, ind_row, n, ind_col, m, nullptr, nullptr);
      DevBuf<long long> d_st;
      stats8(bed, op.rows_identity ? nullptr : op.d_rows.p, n, op.cols_contig ? nullptr : op.d_cols.p, op.col0, m,
             d_st.ensure((size_t)3 * m));
      std::vector<long long> st((size_t)3 * m);
      copy_d2h(bed, st.data(), d_st.p, st.size() * 8);
      const double a = bed->v_off, h = bed->v_step;
      for (int64_t j = 0; j < m; j++) {
        const double s1 = (double)st[(size_t)(3 * j)], s2 = (double)st[(size_t)(3 * j + 1)];
        // sum v = n a + h S1;  sum v^2 = n a^2 + 2 a h S1 + h^2 S2   (v = a + h k, exact integer S1, S2)
        const double xSum = (double)n * a + h * s1;
        const double xxSum = (double)n * a * a + 2.0 * a * h * s1 + h * h * s2;
        const bool na = st[(size_t)(3 * j + 2)] > 0;
        sumX[j] = na ? qnan : xSum;
        denoX[j] = na ? qnan : xxSum - xSum * xSum / n;
      }
      return;
    }
    std::vector<int32_t> cnt((size_t)4 * m);
    counts_host(bed, ind_row, n, ind_col, m, cnt.data());
    for (int64_t j = 0; j < m; j++) {
      const int32_t *c = &cnt[(size_t)4 * j];
      double xSum = (double)c[1] + 2.0 * c[2];
      double xxSum = (double)c[1] + 4.0 * c[2];
      sumX[j] = c[3] > 0 ? qnan : xSum;
      denoX[j] = c[3] > 0 ? qnan : xxSum - xSum * xSum / n;
    }
  });
}

static void read_host(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                      int64_t m, const double *center, const double *scale, int32_t na_val,
                      int32_t *out_i, double *out_d) {
  if (n <= 0 || m <= 0) fail("'ind.row' and 'ind.col' can't be empty.");
  BSN_HIP(hipSetDevice(bed->device));
  if (bed->streamed()) {   // out of core: the columns of every slab, placed where ind.col has them
    SlabWalk W(bed);
    std::vector<int32_t> pi;
    std::vector<double> pd, ce, sc;
    W.run(ind_col, m, [&](bsn_bed *sb, const std::vector<int64_t> &P, const std::vector<int64_t> &local) {
      const int64_t k = (int64_t)P.size();
      if (center) { ce.resize((size_t)k); for (int64_t t = 0; t < k; t++) ce[(size_t)t] = center[P[(size_t)t]]; }
      if (scale) { sc.resize((size_t)k); for (int64_t t = 0; t < k; t++) sc[(size_t)t] = scale[P[(size_t)t]]; }
      if (out_i) pi.resize((size_t)(n * k));
      if (out_d) pd.resize((size_t)(n * k));
      read_host(sb, ind_row, n, local.data(), k, center ? ce.data() : nullptr, scale ? sc.data() : nullptr, na_val,
                out_i ? pi.data() : nullptr, out_d ? pd.data() : nullptr);
      BSN_HIP(hipStreamSynchronize(sb->stream));
      for (int64_t t = 0; t < k; t++) {
        if (out_i) std::memcpy(out_i + P[(size_t)t] * n, pi.data() + t * n, (size_t)n * 4);
        if (out_d) std::memcpy(out_d + P[(size_t)t] * n, pd.data() + t * n, (size_t)n * 8);
      }
    });
    return;
  }
  auto r = to_i32(ind_row, n, bed->n, "ind.row");
  auto c = to_i32(ind_col, m, bed->m, "ind.col");
  DevBuf<int32_t> d_r, d_c, d_oi;
  DevBuf<double> d_ce, d_sc, d_od;
  copy_h2d(bed, d_r.ensure((size_t)n), r.data(), (size_t)n * 4);
  copy_h2d(bed, d_c.ensure((size_t)m), c.data(), (size_t)m * 4);
  if (out_d) {
    copy_h2d(bed, d_ce.ensure((size_t)m), center, (size_t)m * 8);
    copy_h2d(bed, d_sc.ensure((size_t)m), scale, (size_t)m * 8);
    d_od.ensure((size_t)n * m);
  } else {
    d_oi.ensure((size_t)n * m);
  }
  read_dense(bed, d_r.p, n, d_c.p, m, d_ce.p, d_sc.p, na_val, d_oi.p, d_od.p);
  BSN_HIP(hipStreamSynchronize(bed->stream));
  if (out_d)
    copy_d2h(bed, out_d, d_od.p, (size_t)n * m * 8);
  else
    copy_d2h(bed, out_i, d_oi.p, (size_t)n * m * 4);
}

int bsn_bed_read(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                 int64_t m, int32_t na_val, int32_t *out) {
  return guarded([&] { read_host(bed, ind_row, n, ind_col, m, nullptr, nullptr, na_val, out, nullptr); });
}

int bsn_bed_read_scaled(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                        int64_t m, const double *center, const double *scale, double *out) {
  return guarded([&] { read_host(bed, ind_row, n, ind_col, m, center, scale, 0, nullptr, out); });
}

// plane sums behind multLinReg (src/multLinReg.cpp:25-44)
int bsn_bed_cprod_planes(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                         int64_t m, const double *X, int64_t K, double *P, double *Q) {
  return guarded([&] {
    if (K <= 0) fail("'U' must have at least one column.");
    bsn_op op;
    fill_op(&op, bed, ind_row, n, ind_col, m, nullptr, nullptr);
    op.slices = 7;
    DevBuf<double> d_X, d_P, d_Q;
    copy_h2d(bed, d_X.ensure((size_t)n * K), X, (size_t)n * K * 8);
    d_P.ensure((size_t)m * K);
    d_Q.ensure((size_t)m * K);
    op_cprod_raw(&op, d_X.p, n, (int)K, d_P.p, d_Q.p, m);
    copy_d2h(bed, P, d_P.p, (size_t)m * K * 8);
    copy_d2h(bed, Q, d_Q.p, (size_t)m * K * 8);
    BSN_HIP(hipStreamSynchronize(bed->stream));
  });
}

// _bigsnpr_prod_and_rowSumsSq (6 args) src/bed-fun.cpp:103-133
int bsn_bed_prod_and_rowsumssq(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                               int64_t m, const double *center, const double *scale, const double *V,
                               int64_t K, double *XV, double *rowSumsSq) {
  return guarded([&] {
    if (K <= 0) fail("'V' must have at least one column.");
    bsn_op op;
    fill_op(&op, bed, ind_row, n, ind_col, m, center, scale);
    op.slices = 7;
    DevBuf<double> d_V, d_XV, d_r;
    copy_h2d(bed, d_V.ensure((size_t)m * K), V, (size_t)m * K * 8);
    d_XV.ensure((size_t)n * K);
    d_r.ensure((size_t)n);
    op_prod(&op, d_V.p, m, (int)K, d_XV.p, n);
    op_row_sums_sq(&op, d_r.p);
    copy_d2h(bed, XV, d_XV.p, (size_t)n * K * 8);
    copy_d2h(bed, rowSumsSq, d_r.p, (size_t)n * 8);
    BSN_HIP(hipStreamSynchronize(bed->stream));
  });
}

// _bigsnpr_multLinReg (5 args) src/multLinReg.cpp:8-86: K t-scores per variant.  The sums over the
// samples are plane sums of the crossproduct kernel on the panels (U, U^2) at 56 bits plus the exact
// genotype counts; the rest is the reference's expressions per (variant, column).  NA_REAL -> NaN.
int bsn_mult_lin_reg(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col, int64_t m,
                     const double *U, int64_t K, double *res) {
  return guarded([&] {
    if (K <= 0) fail("'U' must have at least one column.");
    if (K > 1024) fail("'U' has too many columns.");
    bsn_op op;
    fill_op(&op, bed, ind_row, n, ind_col, m, nullptr, nullptr, true);
    op.slices = 7;
    DevBuf<double> d_U, d_X, d_P, d_Q, d_tot, d_res;
    DevBuf<int32_t> d_counts;
    copy_h2d(bed, d_U.ensure((size_t)n * K), U, (size_t)n * K * 8);
    d_X.ensure((size_t)n * 2 * K);
    d_tot.ensure((size_t)2 * K);
    hipLaunchKernelGGL(k_mlr_prepare, dim3((unsigned)K), dim3(1024), 0, bed->stream, d_U.p, n, (int)K, d_X.p, d_tot.p);
    BSN_HIP(hipGetLastError());
    counts_device(&op, ind_row, n, d_counts.ensure((size_t)4 * m));
    // complete data (known from an earlier counting pass): the sums over the missing samples are zero and the
    // squared columns are not needed at all
    const bool has_q = !op.no_na;
    const int nvec = (int)(has_q ? 2 * K : K);
    d_P.ensure((size_t)m * nvec);
    d_Q.ensure((size_t)m * nvec);
    op_cprod_raw(&op, d_X.p, n, nvec, d_P.p, d_Q.p, m);
    d_res.ensure((size_t)m * K);
    hipLaunchKernelGGL(k_mlr_final, dim3((unsigned)((m + 255) / 256), (unsigned)K), dim3(256), 0, bed->stream, d_P.p,
                       d_Q.p, m, d_counts.p, d_tot.p, m, (int)K, n, has_q, d_res.p);
    BSN_HIP(hipGetLastError());
    copy_d2h(bed, res, d_res.p, (size_t)m * K * 8);
  });
}

// _bigsnpr_prod_and_rowSumsSq2 (6 args) src/project-utils.cpp:12-43, the FBM.code256 twin: the FBM
// accessor has no missing-value handling, so a missing code is NA_real and poisons its whole row of XV
// and its rowSumsSq entry (:33-38) — reproduced with NaN from the per-sample missing counts.
int bsn_snp_prod_and_rowsumssq2(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                                int64_t m, const double *center, const double *scale, const double *V,
                                int64_t K, double *XV, double *rowSumsSq) {
  int rc = bsn_bed_prod_and_rowsumssq(bed, ind_row, n, ind_col, m, center, scale, V, K, XV, rowSumsSq);
  if (rc != 0) return rc;
  return guarded([&] {
    std::vector<int32_t> rcnt((size_t)4 * n);
    if (bsn_bed_row_counts(bed, ind_row, n, ind_col, m, rcnt.data()) != 0) throw Error(bsn_last_error());
    const double qnan = std::numeric_limits<double>::quiet_NaN();
    for (int64_t i = 0; i < n; i++)
      if (rcnt[(size_t)(4 * i + 3)] > 0) {
        for (int64_t k = 0; k < K; k++) XV[i + k * n] = qnan;
        rowSumsSq[i] = qnan;
      }
  });
}

// running sums of snp_grid_PRS: last[:, c] += Y[:, c]; out[:, c * T + t] = last[:, c]
__global__ void k_prs_accum(const double *__restrict__ Y, int64_t n, int64_t C, double *__restrict__ last,
                            double *__restrict__ out, int64_t T, int64_t t) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * C) return;
  const int64_t i = idx % n, c = idx / n;
  const double v = last[idx] + (Y ? Y[idx] : 0.0);
  last[idx] = v;
  out[(c * T + t) * n + i] = v;
}

int bsn_snp_grid_prs(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col, int64_t m,
                     const double *betas, const int32_t *bin, const uint8_t *member, int64_t C, int64_t T,
                     int slices, double *out) {
  return guarded([&] {
    if (n <= 0) fail("'ind.row' can't be empty.");
    if (C <= 0 || T <= 0) fail("the grid is empty");
    require_gpu();
    BSN_HIP(hipSetDevice(bed->device));
    // columns by bin, highest first (R/PRS.R:66-73 accumulates from the highest threshold down)
    std::vector<std::vector<int64_t>> by_bin((size_t)T + 1);
    for (int64_t j = 0; j < m; j++) {
      if (bin[j] < 0 || bin[j] > T) fail("'bin' out of range");
      bool any = false;
      for (int64_t c = 0; c < C && !any; c++) any = member[c * m + j] != 0;
      if (any && bin[j] > 0) by_bin[(size_t)bin[j]].push_back(j);
    }
    DevBuf<double> d_last, d_out, d_V, d_Y;
    BSN_HIP(hipMemsetAsync(d_last.ensure((size_t)n * C), 0, (size_t)n * C * 8, bed->stream));
    d_out.ensure((size_t)n * C * T);
    d_Y.ensure((size_t)n * C);
    std::vector<double> V;
    std::vector<int64_t> cols;
    const unsigned nblk = (unsigned)((n * C + 255) / 256);
    for (int64_t b = T; b >= 1; b--) {
      const std::vector<int64_t> &jj = by_bin[(size_t)b];
      const int64_t mb = (int64_t)jj.size();
      if (mb > 0) {
        cols.resize((size_t)mb);
        V.assign((size_t)mb * C, 0.0);
        for (int64_t k = 0; k < mb; k++) {
          cols[(size_t)k] = ind_col ? ind_col[jj[(size_t)k]] : jj[(size_t)k];
          for (int64_t c = 0; c < C; c++)
            if (member[c * m + jj[(size_t)k]]) V[(size_t)(c * mb + k)] = betas[jj[(size_t)k]];
        }
        bsn_op op;
        fill_op(&op, bed, ind_row, n, cols.data(), mb, nullptr, nullptr);
        op.slices = slices > 0 ? slices : 7;
        copy_h2d(bed, d_V.ensure((size_t)mb * C), V.data(), (size_t)mb * C * 8);
        op_prod(&op, d_V.p, mb, (int)C, d_Y.p, n);
        hipLaunchKernelGGL(k_prs_accum, dim3(nblk), dim3(256), 0, bed->stream, d_Y.p, n, C, d_last.p, d_out.p,
                           T, b - 1);
        BSN_HIP(hipGetLastError());
        BSN_HIP(hipStreamSynchronize(bed->stream));  // `op` and V are reused by the next bin
      } else {
        hipLaunchKernelGGL(k_prs_accum, dim3(nblk), dim3(256), 0, bed->stream, (const double *)nullptr, n, C,
                           d_last.p, d_out.p, T, b - 1);
        BSN_HIP(hipGetLastError());
      }
    }
    copy_d2h(bed, out, d_out.p, (size_t)n * C * T * 8);
    BSN_HIP(hipStreamSynchronize(bed->stream));
  });
}

static void convert_host(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                         int64_t m, uint8_t *out, bool packed) {
  if (n <= 0 || m <= 0) fail("'ind.row' and 'ind.col' can't be empty.");
  BSN_HIP(hipSetDevice(bed->device));
  if (bed->streamed()) {
    // (round 6) out of core: a column of the result depends on ONE variant, so the selected variants are served slab by
    // slab of the file (the reference maps the file and walks it: src/read-plink.cpp:61-80, src/write-plink.cpp:13-52);
    // every slab's columns go to their places in the caller's array — byte-identical to the resident conversion
    auto r = to_i32(ind_row, n, bed->n, "ind.row");
    const size_t col_bytes = packed ? (size_t)((n + 3) / 4) : (size_t)n;
    SlabWalk walk(bed);
    DevBuf<int32_t> d_r, d_c;
    DevBuf<uint8_t> d_o;
    std::vector<uint8_t> h_o;
    copy_h2d(bed, d_r.ensure((size_t)n), r.data(), (size_t)n * 4);
    walk.run(ind_col, m, [&](bsn_bed *img, const std::vector<int64_t> &P, const std::vector<int64_t> &local) {
      const int64_t ml = (int64_t)P.size();
      auto c = to_i32(local.data(), ml, img->m, "ind.col");
      copy_h2d(bed, d_c.ensure((size_t)ml), c.data(), (size_t)ml * 4);
      d_o.ensure(col_bytes * (size_t)ml);
      if (packed) subset_pack(img, d_r.p, n, d_c.p, ml, d_o.p);
      else to_bytes(img, d_r.p, n, d_c.p, ml, d_o.p);
      h_o.resize(col_bytes * (size_t)ml);
      copy_d2h(bed, h_o.data(), d_o.p, h_o.size());
      BSN_HIP(hipStreamSynchronize(bed->stream));
      for (int64_t t = 0; t < ml; t++) std::memcpy(out + (size_t)P[(size_t)t] * col_bytes, h_o.data() + (size_t)t * col_bytes, col_bytes);
    });
    return;
  }
  auto r = to_i32(ind_row, n, bed->n, "ind.row");
  auto c = to_i32(ind_col, m, bed->m, "ind.col");
  DevBuf<int32_t> d_r, d_c;
  DevBuf<uint8_t> d_o;
  copy_h2d(bed, d_r.ensure((size_t)n), r.data(), (size_t)n * 4);
  copy_h2d(bed, d_c.ensure((size_t)m), c.data(), (size_t)m * 4);
  const size_t bytes = packed ? (size_t)((n + 3) / 4) * m : (size_t)n * m;
  d_o.ensure(bytes);
  if (packed)
    subset_pack(bed, d_r.p, n, d_c.p, m, d_o.p);
  else
    to_bytes(bed, d_r.p, n, d_c.p, m, d_o.p);
  copy_d2h(bed, out, d_o.p, bytes);
  BSN_HIP(hipStreamSynchronize(bed->stream));
}

int bsn_bed_to_fbm(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col, int64_t m,
                   uint8_t *out) {
  return guarded([&] { convert_host(bed, ind_row, n, ind_col, m, out, false); });
}

int bsn_bed_readbina(bsn_bed *bed, const uint8_t *tab, uint8_t *out) {
  return guarded([&] {
    if (!tab || !out) fail("readbina: 'tab' and the output must not be NULL");
    BSN_HIP(hipSetDevice(bed->device));
    DevBuf<uint8_t> d_tab, d_o;
    copy_h2d(bed, d_tab.ensure(1024), tab, 1024);
    if (bed->streamed()) {   // (round 6) out of core: slab by slab of the file, each slab's n x cnt bytes to its place
      const int64_t nslab = slab_count(bed);
      for (int64_t sl = 0; sl < nslab; sl++) {
        int64_t j0 = 0;
        const int64_t cnt = slab_upload(bed, sl, &j0);
        bsn_bed *img = slab_image(bed);
        const size_t bytes = (size_t)bed->n * (size_t)cnt;
        readbina_bytes(img, d_tab.p, d_o.ensure(bytes));
        copy_d2h(bed, out + (size_t)j0 * (size_t)bed->n, d_o.p, bytes);
        BSN_HIP(hipStreamSynchronize(bed->stream));
      }
      return;
    }
    const size_t bytes = (size_t)bed->n * (size_t)bed->m;
    readbina_bytes(bed, d_tab.p, d_o.ensure(bytes));
    copy_d2h(bed, out, d_o.p, bytes);
    BSN_HIP(hipStreamSynchronize(bed->stream));
  });
}

int bsn_bed_subset_payload(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                           int64_t m, uint8_t *payload_out) {
  return guarded([&] { convert_host(bed, ind_row, n, ind_col, m, payload_out, true); });
}

// ---- helpers ---------------------------------------------------------------------------
int bsn_malloc(void **d_ptr, int64_t bytes) {
  return guarded([&] {
    require_gpu();
    BSN_HIP(hipMalloc(d_ptr, (size_t)bytes));
  });
}
int bsn_free(void *d_ptr) {
  return guarded([&] { BSN_HIP(hipFree(d_ptr)); });
}
int bsn_host_alloc(void **h_ptr, int64_t bytes) {
  return guarded([&] {
    require_gpu();
    BSN_HIP(hipHostMalloc(h_ptr, (size_t)bytes, hipHostMallocDefault));
  });
}
int bsn_host_free(void *h_ptr) {
  return guarded([&] { BSN_HIP(hipHostFree(h_ptr)); });
}
int bsn_memcpy_h2d(void *d_dst, const void *src, int64_t bytes) {
  return guarded([&] { BSN_HIP(hipMemcpy(d_dst, src, (size_t)bytes, hipMemcpyHostToDevice)); });
}
int bsn_memcpy_d2h(void *dst, const void *d_src, int64_t bytes) {
  return guarded([&] { BSN_HIP(hipMemcpy(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost)); });
}
int bsn_device_sync(void) {
  return guarded([&] { BSN_HIP(hipDeviceSynchronize()); });
}
int bsn_timer_start(bsn_bed *bed) {
  return guarded([&] { BSN_HIP(hipEventRecord(bed->ev0, bed->stream)); });
}
int bsn_timer_stop(bsn_bed *bed, double *ms) {
  return guarded([&] {
    BSN_HIP(hipEventRecord(bed->ev1, bed->stream));
    BSN_HIP(hipEventSynchronize(bed->ev1));
    float t = 0;
    BSN_HIP(hipEventElapsedTime(&t, bed->ev0, bed->ev1));
    *ms = t;
  });
}

}  // extern "C"
