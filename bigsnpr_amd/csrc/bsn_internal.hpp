// Internal declarations shared by the translation units of libbigsnpr_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <functional>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "bigsnpr_hip.h"

namespace bsn {
// Switches that only exist in the PROFILING build (python -m bigsnpr_amd.build --ablation, libbigsnpr_hip_abl.so): experiments
// whose records are under profiles/ (BSN_ZQ_SPLIT, BSN_START_SLICES, BSN_LD_NOFUSE, BSN_LD_NO_SHARED_DECODE,
// BSN_TCROSS_WAVES) beside BSN_TUNE / BSN_KY / BSN_KY_T / BSN_NB3 / BSN_DIGITS.  The product library does not read them
// (README.md lists the switches it does read).
inline const char *abl_getenv(const char *name) {
#ifdef BSN_ABLATION
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// ---- errors ----------------------------------------------------------------
struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};
void set_error(const char *msg);
[[noreturn]] void fail(const char *fmt, ...);
void require_gpu();   // api.hip: fails ("no CPU fallback") without a HIP device

#define BSN_HIP(expr)                                                             \
  do {                                                                            \
    hipError_t e__ = (expr);                                                      \
    if (e__ != hipSuccess)                                                        \
      ::bsn::fail("HIP error %s at %s:%d (%s)", hipGetErrorString(e__), __FILE__, \
                  __LINE__, #expr);                                               \
  } while (0)

// Every extern "C" entry point body is wrapped with this.
template <class F>
int guarded(F &&f) {
  try {
    f();
    return 0;
  } catch (const std::exception &ex) {
    set_error(ex.what());
    (void)hipGetLastError();   // a failed runtime call stays "the last error" of the thread until it is read: the next
    return 1;                  // entry point's launch check must not trip over it
  } catch (...) {
    set_error("unknown C++ exception");
    (void)hipGetLastError();
    return 1;
  }
}

// ---- device buffer ------------------------------------------------------------
// Work buffers come from a small per-device cache of released blocks (api.hip): a one-shot entry such
// as bsn_bed_prodvec needs about ten of them, and hipMalloc / hipFree pairs cost more than its kernel
// on a matrix of a few GB.  A release keeps hipFree's ordering (the device is idle when the block
// changes hands); blocks above kDevCacheMaxBlock bypass the cache.
constexpr size_t kDevCacheMaxBlock = (size_t)256 << 20;
constexpr size_t kDevCacheMaxTotal = (size_t)2 << 30;
void *dev_alloc(size_t bytes, size_t *granted, int *device);
void dev_release(void *p, size_t granted, int device);
void dev_cache_flush();
size_t dev_cache_held();  // bytes of released blocks held for the current device (handed back on demand)

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  size_t granted = 0;  // bytes of the block behind p
  int device = 0;      // and the device it lives on
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) dev_release(p, granted, device);
    p = nullptr;
    n = 0;
    granted = 0;
  }
  // grow-only
  T *ensure(size_t count) {
    if (count > n) {
      release();
      p = (T *)dev_alloc(count * sizeof(T), &granted, &device);
      n = count;
    }
    return p;
  }
};

constexpr int64_t kPitchAlign = 256;  // bytes; one SNP row = pitch bytes, 1024 samples per 256 B

inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// ---- device coding of a genotype -----------------------------------------------
// The .bed file codes a genotype as 00 = 2 copies of A1, 01 = missing, 10 = 1, 11 = 0
// (src/bed-acc.h:22-37).  The HBM image holds the same 2 bits per genotype at the same
// position, RECODED once at upload to   0, 1, 2 = allele count, 3 = missing,
// so that the masked 2-bit field IS the int8 MFMA operand of the genotype plane (no look-up;
// the 3 of a missing value is subtracted through the missing-value plane, matvec.hip) and
// pad samples / pad variants are plain zero bytes.  Every path that hands bytes back to the
// caller (bsn_bed_download, bsn_bed_subset_payload) applies the inverse.
//   high' = ~high, low' = high ^ low   (and back: high = ~high', low = ~high' ^ low')
__host__ __device__ inline uint32_t dev_from_plink(uint32_t w) {
  return (~w & 0xAAAAAAAAu) | ((w ^ (w >> 1)) & 0x55555555u);
}
__host__ __device__ inline uint32_t plink_from_dev(uint32_t w) {
  return (~w & 0xAAAAAAAAu) | (~(w ^ (w >> 1)) & 0x55555555u);
}

}  // namespace bsn

struct bsn_bed;
namespace bsn {
struct SvdWorkspace;  // svd.hip
}

// RCCL communicator of a column-sharded solve (comm.hip); `comm` is an ncclComm_t
constexpr int kCommClasses = 4;   // timing classes of a solve's collectives: reduce-scatter, all-gather of a block, small, wait
struct bsn_comm {
  void *comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;           // bsn_comm_allreduce (stand-alone self-test); a solve's collectives run on the solve's stream
  // set by comm_abort (the watchdog of a sharded solve, bsn_comm_abort): the communicator is gone, every later call fails
  std::atomic<int> aborted{0};
  // HIP-event timing of every collective of a solve (bsn_svd_options::exchange_timing): class 0 = reduce-scatter of the
  // panel (or of a segment of it), 1 = all-gather of a basis block / of u, 2 = the small all-reduces and all-gathers,
  // 3 = time the solve's stream WAITED for the exchange stream (what of the overlapped reduce-scatters stayed exposed)
  bool timing = false;
  struct Timed { hipEvent_t a, b; int cls; };
  std::vector<Timed> timed;
  std::vector<hipEvent_t> ev_pool;
  ~bsn_comm() {
    for (auto &t : timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    for (auto e : ev_pool) (void)hipEventDestroy(e);
  }
};

constexpr int kProfKinds = 6;   // timing classes of the streaming launches of a solve (prof_collect)
struct bsn_op {
  bsn_bed *bed = nullptr;
  int64_t n = 0, m = 0;        // dimensions of the sub-view
  bool rows_identity = true;   // ind_row == 0..n_file-1
  bool cols_contig = true;     // ind_col == col0 .. col0+m-1
  bool no_na = false;          // every selected variant is known to have no missing genotype
  // the kernels that issue the missing-value plane only for K-steps with a missing code (k_cprod / k_prodT NASKIP),
  // set from the handle's measured share of K-steps without one (op_na_blocks, matvec.hip)
  bool na_skip_c = false, na_skip_p = false;
  // centre / scale are not set yet: the next crossproduct pass counts the codes of every selected
  // variant on the side and derives the binomial scaling from them (bsn_bed_randomsvd)
  bool stats_pending = false;
  bsn::DevBuf<int32_t> d_counts;  // 4 x m code counts of that pass
  bsn::DevBuf<int32_t> d_na;      // m: their missing-value column (kept for the handle's completeness record)
  // [0] total number of missing genotypes of that pass, [1] number of variants with > 50 % missing, copied
  // to pinned host memory behind the pass (-1 until they have arrived): lets the solve switch to the complete-data kernels at its next
  // synchronisation point without one of its own
  long long *h_na_total = nullptr;
  bool na_poll = false;
  int64_t col0 = 0;
  int slices = 4;
  int64_t passes = 0;         // streaming launches over the image issued so far
  // one-shot crossproduct (api.hip): centre / scale (host pointers, m doubles each, may be null = default) are only needed
  // by the finalize kernel; the next op_cprod uploads them on the handle's second stream beside its streaming kernel
  const double *late_center = nullptr, *late_scale = nullptr;
  // op_cprod_prequant: the digits of this panel are already in d_q (quantised ahead of the call that uses them)
  const double *preq_X = nullptr;
  int64_t preq_ldx = 0;
  int preq_nvec = 0, preq_S = 0;
  // per-launch HIP-event timing of the streaming kernels (kind 0 = k_cprod, 1 = k_prod, 2 = the k_cprod
  // launch that also counts the codes, first pass of a solve with fused scaling statistics)
  bool profile = false;
  int prof_kind_override = -1;  // >= 0: every launch is filed under this kind (3 = warm-start launches on a subset)
  std::vector<hipEvent_t> ev_begin, ev_end;
  std::vector<int> ev_kind;
  std::vector<char> ev_more;   // 1: a further piece of the previous launch (prof_begin)
  const void *prof_kernel[kProfKinds] = {};  // host stub of the last kernel launched under each kind
  ~bsn_op() {
    for (auto e : ev_begin) (void)hipEventDestroy(e);
    for (auto e : ev_end) (void)hipEventDestroy(e);
    if (h_na_total) (void)hipHostFree(h_na_total);
  }
  bsn::DevBuf<int32_t> d_rows;   // n (gather list) when !rows_identity
  bsn::DevBuf<int32_t> d_cols;   // m_pad (padded by repeating a valid column)
  bsn::DevBuf<double> d_center, d_scale;  // m
  // workspaces (grow-only)
  bsn::DevBuf<double> d_xfull;   // scattered / padded input panel
  bsn::DevBuf<int8_t> d_q;       // quantised panels
  bsn::DevBuf<int32_t> d_acc;    // raw int32 MFMA accumulators
  bsn::DevBuf<double> d_meta;    // per-vector scale, sums
  bsn::DevBuf<double> d_yfull;   // full-length output before the row gather
};

// ---- the handle -------------------------------------------------------------
// HBM layout: variant-major like the file (src/bed-acc.h:71-75): variant j occupies
// bytes [j*pitch, j*pitch + n_byte); sample i sits in bits 2*(i%4) of byte i/4, in the
// device coding above (0/1/2 = allele count, 3 = missing).  pitch = n_byte rounded up to
// 256 B.  Pad samples (pad bits of the last real byte and all pad bytes) are 0 (genotype 0,
// not missing) so that they contribute nothing to any plane product; kernels may therefore
// run over [0, 4*pitch) samples.
struct bsn_bed {
  int64_t n = 0, m = 0, n_byte = 0, pitch = 0;
  // bits = 2: the 2-bit image described above.  bits = 8 (FBM.code256 whose decoded values lie on a
  // grid v = v_off + v_step k, |k| <= 127, e.g. CODE_DOSAGE: 0.00 .. 2.00 by 0.01, R/bigSNP-class.R:13):
  // one int8 k per genotype, -128 = missing, variant j at j * pitch, pitch = n rounded up to 256 B,
  // pad samples 0.  The byte IS the int8 MFMA operand; sums of k are exact integers and the affine map
  // back to values is applied in fp64 at the end.
  int bits = 2;
  double v_off = 0.0, v_step = 1.0;
  // bits = 8 and generic: an FBM.code256 whose decode table is neither calls nor a regular grid (any 256
  // doubles, src/colstats.cpp:13-14).  The image holds the FBM's own bytes and d_lut the table; there is no
  // exact integer image, so only the fp64 look-up kernels of image.hip serve it (snp_colstats, big_prodVec,
  // big_cprodVec: VALU-bound, correct, slow) and every other entry point refuses it by name.
  bool generic = false;
  double *d_lut = nullptr;
  uint8_t *d_img = nullptr;
  // Second copy of a 2-bit image in the STREAMING layout (image.hip, image_tile): tiles of 64 variants x
  // 256 B (1024 samples), tile (vb, sb) at ((vb * pitch / 256) + sb) * 16 KB, variant v of the block at
  // v * 256 inside it.  k_cprod / k_prod read it when the operator covers a 64-aligned contiguous range of
  // variants: what a wave touches per step is then one contiguous 4 - 16 KB run instead of 16 - 64 pieces
  // 100 KB apart.  Built on demand when the device has the room, freed with the handle.
  uint8_t *d_tiled = nullptr;
  bool tiled_tried = false;
  size_t tiled_cap = 0;   // bytes of the allocation behind d_tiled (an in-place re-gather rebuilds the copy inside it)
  // Second copy of a 2-bit image in SAMPLE-MAJOR order (image.hip, image_smaj; round 4): the variants are the contiguous
  // index — four per byte, same device coding, variant 4 k + e in bits 2 e — and the copy is laid out CHUNK-MAJOR: the
  // 128 bytes "variants 512 ch .. 512 ch + 511 of sample i" sit at (ch * rows_smaj + i) * 128, rows_smaj = n rounded up
  // to 256 (pad samples are code 0), pitch_smaj / 128 chunks (m + 512 rounded up to 1024 variants; zeros past variant m).
  // On it the product A~ X contracts over the contiguous index, like the crossproduct does on the variant-major image:
  // k_prodT is k_cprod's shape (no 4 x 4 byte transposes, 16 instead of 128 accumulator registers, four waves per SIMD
  // instead of two), and the 512 samples x 128 B a workgroup reads per chunk are one contiguous 64-KB run — DESIGN.md
  // 3.3b.  Built once per handle by a solve on the two-block kernels when the device has the room (image + 24 GB),
  // freed with the handle / bsn_bed_release_workspace.  BSN_NO_SMAJ=1 forbids it.
  // Out-of-core handle (round 4).  A .bed file whose 2-bit image does not fit the free device memory (or the budget
  // BSN_IMAGE_BUDGET, bytes) is not refused — the reference maps a file of any size (src/bed-acc.h:46,
  // src/bed-acc-xptr.cpp:14-35): the file stays mapped (h_map = its payload), d_img is NULL, and the one-shot entry
  // points (bed_prodVec, bed_cprodVec, column counts / colstats / MAF / scaleBinom) walk it in slabs of `slab_cols`
  // variants, each uploaded into one resident slab image and served by the same kernels: PCIe-bound (~30 GB/s)
  // instead of HBM-bound.  Every other entry point names the reason it needs a resident image.
  const uint8_t *h_map = nullptr;   // first payload byte of the mapped file
  void *map_base = nullptr;
  size_t map_len = 0;
  int64_t slab_cols = 0;
  int fd_file = -1;                 // the open file: slabs are read through the pinned double-buffered upload of image_from_file
  bsn_bed *slab_img = nullptr;      // the resident slab image and its staging buffers, kept between calls (api.hip, SlabWalk)
  void *slab_stage = nullptr;       // bsn::FileStage
  bool streamed() const { return d_img == nullptr && h_map != nullptr; }
  uint8_t *d_smaj = nullptr;
  int64_t pitch_smaj = 0, rows_smaj = 0;
  bool smaj_tried = false;
  size_t smaj_cap = 0;              // bytes of the d_smaj allocation
  // (round 6) the copy being made BESIDE the first solve that wants it (image.hip, image_smaj_start): a helper thread
  // allocates it and queues k_smaj_build on a stream of its own; the passes of the solve take k_prod until an event says
  // the copy is complete (image_smaj_poll) — same integer sums either way.  nullptr: no build in flight.
  void *smaj_job = nullptr;         // bsn::SmajJob
  int64_t cap_m = 0;                // variants the d_img allocation holds (image_alloc)
  // Compacted sub-image (round 5, svd.hip compacted_view): a solve over a NON-contiguous list of variants — every
  // solve of bed_autoSVD / snp_autoSVD, whose ind.col is the clumped set (R/autoSVD.R:296-301) — runs on a gathered
  // copy of the selected variants (and samples, when ind.row is a proper list) instead of through gather lists: the
  // copy is a contiguous image, so the fast kernel family applies to it (buffer loads, its own sample-major copy,
  // k_prodT, three column blocks, the counts riding along the first pass).  One gather pass buys every pass of the
  // solve.  Owned by this handle, keyed by a hash of the lists, re-gathered IN PLACE when the next selection fits the
  // allocation (the rounds of autoSVD only remove variants), freed with the handle / bsn_bed_release_workspace.
  bsn_bed *sub = nullptr;
  uint64_t sub_key = 0;
  std::vector<int64_t> sub_cols, sub_rows;   // the lists behind sub_key (compared on a key match: a 64-bit hash can collide)
  bool last_solve_on_sub = false;            // bsn_bed_streaming_kernels reports the sub-handle's launches then
  int device = 0;
  bool stream_borrowed = false;      // `stream` belongs to another handle (the slab image of an out-of-core handle)
  hipStream_t stream = nullptr;
  hipStream_t stream_up = nullptr;   // second stream (created on first use): uploads beside the kernels of `stream`
  hipEvent_t ev_up = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // number of missing genotypes per variant over ALL samples, -1 = not known yet; filled as a
  // by-product of every full-row count (and at creation for FBM / NA-free synthetic images).
  // An operator whose variants are all known to be complete skips the missing-value plane.
  std::vector<int32_t> na_cnt;
  // code counts per variant as the last count entries returned them, for ONE row selection (round 6: snp_autoSVD asks for the
  // column statistics of the same rows three times — snp_MAF, snp_clumping, the scaling function: R/autoSVD.R:103-127 — and
  // each is a pass over the whole image).  cnt[4 j] < 0: not known.  Dropped wherever na_cnt is (the image changed).
  struct CountsCache {
    bool rows_all = false;
    std::vector<int64_t> rows;      // the selection when it is not all rows in file order
    std::vector<int32_t> cnt;       // 4 per variant of the handle
    void clear() {
      rows_all = false;
      rows.clear();
      cnt.clear();
    }
  } counts_cache;
  // Share of the K-steps of the two streaming products that carry NO missing code (round 5, matvec.hip op_na_blocks):
  // sampled once per image by a small kernel queued on the handle's stream, picked up from pinned memory by a later
  // launch (no synchronisation).  state 0 = not measured, 1 = queued, 2 = known.  [0] crossproduct (16 variants x
  // 64 samples per step), [1] product on the sample-major copy (16 samples x 64 variants).
  int na_blocks_state = 0;
  long long *h_na_blocks = nullptr;   // pinned: free / sampled steps of the two shapes, then the arrival flag
  long long *d_na_blocks = nullptr;
  double na_free[2] = {0.0, 0.0};
  // Workspace of bsn_bed_randomsvd, kept between solves on this handle (basis, panels, quantised
  // operands: ~4 GB at 400K x 1M) instead of being allocated and freed by every solve; released by
  // bsn_bed_release_workspace or with the handle.
  std::unique_ptr<bsn_op> svd_op;
  std::shared_ptr<bsn::SvdWorkspace> svd_ws;
  // two pinned staging buffers for host <-> device transfers of caller (pageable) memory, see copy_h2d
  uint8_t *h_stage[2] = {nullptr, nullptr};
  hipEvent_t ev_stage[2] = {nullptr, nullptr};
  bool stage_busy[2] = {false, false};  // ev_stage[i] marks a transfer that may still read h_stage[i]
  unsigned stage_next = 0;
};


namespace bsn {

// image.hip
void image_alloc(bsn_bed *b, int64_t n, int64_t m, int bits = 2);
void stage_init(bsn_bed *b);   // api.hip: the two page-locked staging buffers of copy_h2d / copy_d2h (made with a large image, else on first use)
void bed_free(bsn_bed *b);  // api.hip: everything a handle owns
// a new handle holding the sub-matrix [ind_row, ind_col] (rows in list order, repeats allowed), same coding
// `reuse`: a handle made by an earlier call whose allocation holds the new sub-matrix (same number of samples, at
// least m variants): gathered into in place, its sample-major copy rebuilt if it has one
bsn_bed *image_gather(bsn_bed *src, const int64_t *ind_row, int64_t n, const int64_t *ind_col, int64_t m,
                      bsn_bed *reuse = nullptr);
bool image_tile(bsn_bed *b);  // true when the streaming-layout copy exists (builds it if memory allows)
bool image_smaj(bsn_bed *b);  // true when the sample-major copy exists (builds it if memory allows; waits for a build in flight)
// Non-blocking forms (round 6): start the build beside whatever the caller queues next — true when the copy exists or is on
// its way —, ask whether it has arrived (cheap: an atomic and at most one hipEventQuery), wait for a build in flight.
bool image_smaj_start(bsn_bed *b);
bool image_smaj_poll(bsn_bed *b);
void image_smaj_wait(bsn_bed *b);
void image_from_host(bsn_bed *b, const uint8_t *payload, int64_t n_byte_src);
// the two page-locked staging buffers (+ their events) of image_from_file; a caller that uploads many pieces (the slabs of an
// out-of-core handle) keeps one set alive instead of paying two hipHostMalloc of 256 MB per piece
struct FileStage {
  uint8_t *pin[2] = {nullptr, nullptr};
  hipEvent_t done[2] = {nullptr, nullptr};
  size_t bytes = 0;
  ~FileStage();
  FileStage() = default;
  FileStage(const FileStage &) = delete;
  FileStage &operator=(const FileStage &) = delete;
};
void image_from_file(bsn_bed *b, int fd, int64_t offset, int64_t n_byte_src, FileStage *stage = nullptr);
// FBM bytes -> device image through a 256-entry byte look-up (lut[byte] = device code 0..3 for a 2-bit
// image, int8 grid index or 0x80 for a byte image); pinned double-buffered upload
void image_from_fbm(bsn_bed *b, const uint8_t *bytes, int64_t ld, const uint8_t *lut);
// byte image: per-variant sums of k, k^2 and number of missing values over all file rows (d_rows ==
// NULL) or over a row list (with multiplicity); out: 3 x m int64 (S1, S2, nNA)
void stats8(bsn_bed *b, const int32_t *d_rows, int64_t n, const int32_t *d_cols, int64_t col0, int64_t m,
            long long *d_out);
// generic decode table (bsn_bed::generic): fp64 look-up kernels.  d_rows / d_cols as above; center / scale may
// be NULL (0 / 1).  lut_colstats: out 2 x m doubles (sum v, sum v^2 over the selected rows);
// lut_cprod: z[j] = (sum_i v_ij x_i - c_j sum_i x_i) / s_j;  lut_prod: y[i] = sum_j (v_ij - c_j) x_j / s_j.
void lut_colstats(bsn_bed *b, const int32_t *d_rows, int64_t n, const int32_t *d_cols, int64_t col0, int64_t m,
                  double *d_out);
void lut_cprod(bsn_bed *b, const int32_t *d_rows, int64_t n, const int32_t *d_cols, int64_t col0, int64_t m,
               const double *d_center, const double *d_scale, const double *d_x, double *d_z);
void lut_prod(bsn_bed *b, const int32_t *d_rows, int64_t n, const int32_t *d_cols, int64_t col0, int64_t m,
              const double *d_center, const double *d_scale, const double *d_x, double *d_y);
void refuse_generic(const bsn_bed *b, const char *what);  // fails with a clear message on a generic image
void image_generate(bsn_bed *b, uint32_t seed, uint32_t npop, uint32_t na16, int64_t j_begin);
void image_download(bsn_bed *b, uint8_t *payload_out);
// counts for variants cols[0..m) (device list or contiguous from col0) over all file rows;
// d_counts: 4 x m int32 column-major (0,1,2,NA)
void counts_all_rows(bsn_bed *b, const int32_t *d_cols, int64_t col0, int64_t m, int32_t *d_counts);
// host result, 4 x m (counts of 0, 1, 2, NA) for arbitrary row / column selections (api.hip)
void counts_host(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col, int64_t m,
                 int32_t *res);
void read_dense(bsn_bed *b, const int32_t *d_rows, int64_t n, const int32_t *d_cols, int64_t m,
                const double *d_center, const double *d_scale, int32_t na_val, int32_t *d_out_i,
                double *d_out_d);

void to_bytes(bsn_bed *b, const int32_t *d_rows, int64_t n, const int32_t *d_cols, int64_t m,
              uint8_t *d_out);
void readbina_bytes(bsn_bed *b, const uint8_t *d_tab, uint8_t *d_out);
void subset_pack(bsn_bed *b, const int32_t *d_rows, int64_t n, const int32_t *d_cols, int64_t m,
                 uint8_t *d_out);

// api.hip: transfers between caller memory (pageable) and the device, staged through the handle's two
// pinned buffers in 32-MB pieces (host copy of piece k+1, on a few threads, overlaps the DMA of piece k);
// synchronous.
// The runtime's own handling of pageable buffers was measured to leave every later stream
// synchronisation of the process with a ~4 ms wake-up latency (tools/gpu/r02_h.sh), which costs a
// solve 10 % — so no hot entry point hands pageable memory to hipMemcpy.
void copy_h2d(bsn_bed *b, void *d_dst, const void *src, size_t bytes, hipStream_t stream = nullptr);  // default: the handle's stream
hipStream_t upload_stream(bsn_bed *b);
void copy_d2h(bsn_bed *b, void *dst, const void *d_src, size_t bytes);

// api.hip: operator over a sub-view; defer_scale leaves centre / scale unset (stats_pending path)
void fill_op(bsn_op *op, bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
             int64_t m, const double *center, const double *scale, bool defer_scale = false,
             bool allow_streamed = false);
// api.hip: the resident slab image of an out-of-core handle (on the handle's stream), the number of slabs, and the
// upload of one slab into it (returns its number of variants)
bsn_bed *slab_image(bsn_bed *b);
int64_t slab_count(const bsn_bed *b);
int64_t slab_upload(bsn_bed *b, int64_t sl, int64_t *j0_out = nullptr);
void slab_upload_range(bsn_bed *b, int64_t j0, int64_t cnt);   // any run of at most slab_cols variants

// comm.hip: RCCL collectives on device buffers of doubles, enqueued on `st`
// (exchange timing) a pair of events around `what` on `st`, filed under class cls; no-ops unless c->timing
hipEvent_t comm_time_begin(bsn_comm *c, hipStream_t st);
void comm_time_end(bsn_comm *c, hipEvent_t begin, int cls, hipStream_t st);
// sums (ms) and counts per class of everything timed since the last call; waits for the events
void comm_time_collect(bsn_comm *c, double ms[kCommClasses], int count[kCommClasses]);
// ncclCommAbort: collectives in flight end, later ones fail; returns false if the RCCL at hand has no such entry point
bool comm_abort(bsn_comm *c);
void comm_allreduce_sum(bsn_comm *c, double *d_buf, int64_t count, hipStream_t st);
void comm_reduce_scatter_sum(bsn_comm *c, const double *d_send, double *d_recv, int64_t recv_count,
                             hipStream_t st);
void comm_all_gather(bsn_comm *c, const double *d_send, double *d_recv, int64_t send_count, hipStream_t st);

// matvec.hip
void op_poll_stats(bsn_op *op);  // after a stream synchronisation: pick up the missing-value total
void op_na_blocks(bsn_op *op, int digit_cols);   // queue / pick up the handle's share of K-steps without a missing code; sets op->na_skip_*
void op_prod(bsn_op *op, const double *d_X, int64_t ldx, int nvec, double *d_Y, int64_t ldy);
// Y = beta Y + A~ X (the slabs of an out-of-core solve add up their products on the device)
void op_prod_acc(bsn_op *op, const double *d_X, int64_t ldx, int nvec, double *d_Y, int64_t ldy, double beta);
// The product pass in SEGMENTS of sample blocks (sharded solve, svd.hip: the reduce-scatter of a finished segment runs
// on a second stream while the next segment computes).  The samples are seen as `pieces` equal pieces (the ranks' sample
// blocks) of `stride` workgroup blocks of 512 samples each; segment s covers blocks [off, off + bs) of EVERY piece, and
// its result goes to d_out in the blocked layout [piece][vector][row of the segment] (what a reduce-scatter exchanges).
// d_rows lists the segment's samples piece by piece (pieces * bs * 512 entries, -1 = padding past the last sample: zero).
// `after(s)` is called once the kernels of segment s are queued.  Returns false — nothing queued — when the pass cannot
// run as ONE k_prodT launch geometry (no sample-major copy, several launches, ...): the caller takes op_prod.
struct ProdSegment {
  int bs = 0, off = 0;
  const int32_t *d_rows = nullptr;
  double *d_out = nullptr;
};
bool op_prod_segments(bsn_op *op, const double *d_X, int64_t ldx, int nvec, int pieces, int stride, int nseg,
                      const ProdSegment *segs, const std::function<void(int)> &after);
void op_cprod(bsn_op *op, const double *d_X, int64_t ldx, int nvec, double *d_Z, int64_t ldz);
// the quantisation half of op_cprod for a panel that fits one launch, queued ahead of time (the SVD driver
// does it while the host still works on the previous step); op_cprod with the same panel then starts with
// its streaming kernel.  Any other product on the operator forgets the digits.
void op_cprod_prequant(bsn_op *op, const double *d_X, int64_t ldx, int nvec);
void op_cprod_raw(bsn_op *op, const double *d_X, int64_t ldx, int nvec, double *d_P, double *d_Q,
                  int64_t ld);  // P = sum_i g0 x, Q = sum_i na x (m x nvec each)
void require_bits(const bsn_bed *b, int bits, const char *what);  // fails with a clear message otherwise
void op_row_sums_sq(bsn_op *op, double *d_out);
void op_row_counts(bsn_op *op, double *d_out);  // n doubles: sum_j A~[i, j]^2
// weighted code counts: d_w = per-file-row integer weights (n_file doubles); out 4 x m
void counts_weighted(bsn_op *op, const double *d_w, int64_t n_sub, int32_t *d_counts);
void selftest();
// event helpers around a streaming launch (no-ops unless op->profile)
// roctx ranges (BSN_ROCTX=1: the roctx library is dlopen'ed on first use; otherwise, or when it is absent, no-ops):
// rocprofv3 --marker-trace --kernel-trace then cuts the timeline of a solve by itself — warm start, every block step's
// phases, the formation of u / v (tools/README.md)
void roctx_push(const char *name);
void roctx_pop();
struct RoctxRange {
  explicit RoctxRange(const char *name) { roctx_push(name); }
  ~RoctxRange() { roctx_pop(); }
  RoctxRange(const RoctxRange &) = delete;
  RoctxRange &operator=(const RoctxRange &) = delete;
};
// api.hip: fails with the reason when `b` is an out-of-core handle (bsn_bed::streamed)
void require_resident(const bsn_bed *b, const char *what);
void prof_begin(bsn_op *op, int kind, bool more = false);
void prof_end(bsn_op *op);
// sums the recorded launches: ms[kind], count[kind]; clears the records
// kind 2: k_cprod carrying the code counts; 3: warm start; 4 / 5: the three-column-block launches of k_cprod / k_prodT
void prof_collect(bsn_op *op, double ms[kProfKinds], int count[kProfKinds]);

}  // namespace bsn
