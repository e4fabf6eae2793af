/*
 * bigsnpr_hip.h — C ABI of libbigsnpr_hip.so: the MI355X (gfx950) implementation of
 * bigsnpr's genotype-matrix hot path.
 *
 * This is the drop-in boundary.  Every entry point replaces one `.Call` target of
 * the reference package (bigsnpr 1.12.21, registration table
 * src/RcppExports.cpp:597-640) or one operator that the reference hands to
 * bigstatsr::big_randomSVD (R/autoSVD.R:216-218).  The reference-side binding (an R
 * `.Call` shim) is shown in INTEGRATION.md and bindings/R/.
 *
 * Conventions
 *  - plain C types only; no R, Rcpp, torch or HIP types in any signature;
 *  - all functions return 0 on success, non-zero on error; bsn_last_error() then
 *    returns a thread-local message whose text matches the reference's
 *    Rcpp::stop() strings where the reference has one;
 *  - indices are 0-based int64 (the R shim subtracts 1 exactly where the reference
 *    does, src/bed-acc.h:64-65); arbitrary order, duplicates allowed;
 *  - `*_host` style pointers are borrowed host buffers (never freed, never kept);
 *    `d_*` pointers are device (HBM) pointers on the handle's device;
 *  - a handle owns the device image of one genotype matrix; it is created on the
 *    current device (bsn_set_device) and all work is issued on its own HIP stream.
 *  - There is NO CPU fallback: without a gfx950 device every compute call fails.
 */
#ifndef BIGSNPR_HIP_H
#define BIGSNPR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bsn_bed bsn_bed; /* packed 2-bit genotype matrix resident in HBM */
typedef struct bsn_op bsn_op;   /* scaled sub-view  A~ = ((G - center)/scale)[ind_row, ind_col] */

/* ---- runtime ------------------------------------------------------------- */
const char *bsn_last_error(void);
int bsn_version(void);
int bsn_device_count(int *count);
int bsn_set_device(int device);
/* verifies the gfx950 instruction-level assumptions the kernels rely on
 * (v_perm_b32 byte order, i8 MFMA operand/accumulator layout) on the device */
int bsn_selftest(void);

/* ---- genotype image (replaces `class bed` + bedXPtr) ----------------------
 * src/bed-acc.h:18-48, src/bed-acc-xptr.cpp:14-53 (_bigsnpr_bedXPtr, 3 args).
 * Error strings: "Error when mapping file:\n  %s.\n", "File is not a binary PED
 * file.", "Variant-major is the only mode supported.", "n or p does not match the
 * dimensions of the file." */
int bsn_bed_open(const char *path, int64_t n, int64_t m, bsn_bed **out);
/* 1 if the handle is OUT OF CORE (round 4): the file's 2-bit image did not fit the free device memory (or the budget
 * BSN_IMAGE_BUDGET, bytes) when bsn_bed_open was called, so the file stays mapped and bsn_bed_prodvec, bsn_bed_cprodvec,
 * bsn_bed_col_counts and bsn_bed_colstats walk it in slabs of variants (PCIe-bound instead of HBM-bound; same kernels,
 * same results).  The reference maps a file of any size (src/bed-acc.h:46); every other entry point refuses such a
 * handle with the reason. */
int bsn_bed_is_streamed(const bsn_bed *bed);
/* same, from a payload already in host memory (n_byte = bytes per variant, >= ceil(n/4)) */
int bsn_bed_from_host(const uint8_t *payload, int64_t n, int64_t m, int64_t n_byte, bsn_bed **out);
/* FBM.code256 (bigstatsr, one byte per genotype, column-major, n_total rows) with the
 * CODE_012 coding (R/bigSNP-class.R:7: 0,1,2, everything else NA) repacked to 2 bits
 * on the device; the FBM twin of bedXPtr for snp_* functions (src/colstats.cpp:13-14). */
int bsn_bed_from_fbm(const uint8_t *bytes, int64_t n, int64_t m, int64_t ld, bsn_bed **out);
/* The same for any FBM.code256: `code256` is the object's 256-entry decode table (R/bigSNP-class.R:7-13).
 * Tables that decode to genotype calls only (0 / 1 / 2 / NA: CODE_012, CODE_IMPUTE_PRED) give the 2-bit
 * image and every entry point of this header.  Tables whose values lie on a regular grid of at most 255
 * steps (CODE_DOSAGE: calls, imputed calls and dosages 0.00 .. 2.00 by 0.01) give a BYTE image — one int8
 * grid index per genotype, exact integer sums, the affine map back to values applied in fp64 — which
 * serves bsn_snp_colstats, bsn_bed_prodvec / bsn_bed_cprodvec (big_prodVec / big_cprodVec and so snp_PRS),
 * bsn_op_*, bsn_bed_randomsvd with explicit centre / scale and, for data without missing values,
 * bsn_cormat / bsn_ld_scores / bsn_clumping_chr(_cached) in the FBM mode; the other entry points fail on it
 * with a message.  Any other table is refused. */
int bsn_fbm_open(const uint8_t *bytes, int64_t n, int64_t m, int64_t ld, const double *code256, bsn_bed **out);
int bsn_bed_bits(const bsn_bed *bed); /* 2 or 8 */
/* total number of missing genotypes of the image if a full count has seen every variant (FBM handles
 * know it from their creation), -1 otherwise */
int64_t bsn_bed_na_known(const bsn_bed *bed);
/* synthetic matrix generated directly in HBM (DESIGN.md "Synthetic inputs");
 * byte-identical to oracle/bsn_oracle.c:orc_fake_bed */
int bsn_bed_synthetic(int64_t n, int64_t m, uint32_t seed, uint32_t npop, uint32_t na16,
                      int64_t j_begin, bsn_bed **out);
int bsn_bed_close(bsn_bed *bed);
int64_t bsn_bed_nrow(const bsn_bed *bed);
int64_t bsn_bed_ncol(const bsn_bed *bed);
int64_t bsn_bed_bytes(const bsn_bed *bed); /* HBM bytes held by the image */
/* copies the image back in .bed payload layout (ceil(n/4) bytes per variant) */
int bsn_bed_download(bsn_bed *bed, uint8_t *payload_out);

/* ---- .Call replacements (host buffers in, host buffers out) --------------- */
/* _bigsnpr_bed_pMatVec4 (7 args) src/bed-prod-vec.cpp:15-54 — R: bed_prodVec.
 * y[n] = A~ x[m].  "Incompatibility between dimensions" is raised by the caller
 * layer (lengths are explicit here). `ncores` of the reference is dropped. */
int bsn_bed_prodvec(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                    int64_t m, const double *center, const double *scale, const double *x,
                    double *y);
/* _bigsnpr_bed_cpMatVec4 (7 args) src/bed-prod-vec.cpp:59-97 — R: bed_cprodVec. z[m] = A~' x[n] */
int bsn_bed_cprodvec(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                     int64_t m, const double *center, const double *scale, const double *x,
                     double *z);

/* bed_prodVec on a column-sharded matrix ("each GPU produces a partial n-vector, one all-reduce"):
 * every rank passes its own shard (handle, ind_col, center / scale and the matching slice of x) and the
 * communicator of bsn_comm_init; y[n] receives the sum over the ranks — the product with the whole matrix — on
 * every rank.  The partial vectors are added on the device over RCCL (fp64, 3.2 MB at 400 000 samples) before
 * the one download.  comm == NULL: plain bsn_bed_prodvec.  The crossproduct needs no counterpart: z = A~' x is
 * sharded like the columns, bsn_bed_cprodvec on every rank already returns that rank's slice. */
struct bsn_comm;
int bsn_bed_prodvec_sharded(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                            int64_t m, const double *center, const double *scale, const double *x,
                            struct bsn_comm *comm, double *y);
/* _bigsnpr_bed_col_counts_cpp (4 args) src/bed-fun.cpp:51-69: res is 4 x m column-major
 * (rows: counts of 0, 1, 2, NA) */
int bsn_bed_col_counts(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                       int64_t m, int32_t *res);
/* _bigsnpr_bed_row_counts_cpp (4 args) src/bed-fun.cpp:72-99 (bed_counts(byrow = TRUE),
 * R/binom-scaling.R:166-178): per SAMPLE counts of 0, 1, 2, NA over the selected variants,
 * res[4 x n] column-major. */
int bsn_bed_row_counts(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col, int64_t m,
                       int32_t *res);
/* _bigsnpr_bed_colstats (4 args) src/bed-fun.cpp:9-46; *n_bad = number of variants with
 * more than 50 % missing values (the reference warns "%d variants have >50%% missing values.") */
int bsn_bed_colstats(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                     int64_t m, double *sumX, double *denoX, int32_t *nb_nona_col, int32_t *n_bad);
/* _bigsnpr_snp_colstats (4 args) src/colstats.cpp:8-35 (no NA handling: NA counts as 3
 * exactly as code256[3] would if it were 3; callers assert no NA, R/utils-assert.R:31) */
int bsn_snp_colstats(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                     int64_t m, double *sumX, double *denoX);
/* _bigsnpr_read_bed (3 args) / _bigsnpr_read_bed_scaled (5 args) src/bed-mat-acc.cpp:8-49:
 * dense n x m column-major; read_bed codes NA as na_val */
int bsn_bed_read(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                 int64_t m, int32_t na_val, int32_t *out);
int bsn_bed_read_scaled(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                        int64_t m, const double *center, const double *scale, double *out);

/* The genotype-touching part of _bigsnpr_multLinReg (5 args, src/multLinReg.cpp:8-86; SURVEY.md
 * §8f-2): for every variant j and column k of X (n x K, e.g. PC scores)
 *   P[j, k] = sum_i g_ij X[i, k] over non-missing genotypes,  Q[j, k] = sum_{i: g_ij missing} X[i, k]
 * (m x K column-major).  With the genotype counts these give xySum, ySum, yySum of the reference;
 * the K t-scores per variant are then a few flops on the host (bigsnpr_amd/pcadapt.py). */
int bsn_bed_cprod_planes(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                         int64_t m, const double *X, int64_t K, double *P, double *Q);

/* _bigsnpr_multLinReg (5 args) src/multLinReg.cpp:8-86, whole: res[m x K] column-major, the t-score of
 * each variant regressed on each column of U (n x K) over its non-missing genotypes; NaN where the
 * reference returns NA_REAL (zero denominator or fewer than two non-missing values). */
int bsn_mult_lin_reg(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col, int64_t m,
                     const double *U, int64_t K, double *res);

/* _bigsnpr_prod_and_rowSumsSq (6 args) src/bed-fun.cpp:103-133 (the kernel of
 * bed_projectSelfPCA, R/bed-projectPCA.R:45-59): XV[n x K] = A~ V[m x K] and
 * rowSumsSq[i] = sum_j A~[i, j]^2, column-major host buffers. */
int bsn_bed_prod_and_rowsumssq(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                               int64_t m, const double *center, const double *scale, const double *V,
                               int64_t K, double *XV, double *rowSumsSq);

/* _bigsnpr_prod_and_rowSumsSq2 (6 args) src/project-utils.cpp:12-43: the same on an FBM.code256 handle;
 * rows that hold a missing code come back NaN (the FBM accessor yields NA_real there). */
int bsn_snp_prod_and_rowsumssq2(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                                int64_t m, const double *center, const double *scale, const double *V,
                                int64_t K, double *XV, double *rowSumsSq);

/* The genotype-touching part of snp_grid_PRS (R/SCT.R:201-262), which in the
 * reference is one snp_PRS call (R/PRS.R:36-76 -> bigstatsr::big_prodVec) per clumping set:
 * scores for C column sets x T thresholds in a single sweep, one n x C GEMM per threshold bin.
 *   ind_col[m], betas[m]  the union of the C sets and its weights
 *   bin[m]                number of (ascending-sorted) thresholds that lpS[j] exceeds, 0..T
 *   member[m x C]         column-major 0/1: column j belongs to set c
 *   out[n x (C*T)]        column c*T + t = sum over j in set c with bin[j] > t of G[, j] betas[j],
 *                         accumulated from the highest threshold down as snp_PRS does
 * slices: fixed-point width of the weight panel (0 -> 7 = 56 bits). */
int bsn_snp_grid_prs(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col, int64_t m,
                     const double *betas, const int32_t *bin, const uint8_t *member, int64_t C, int64_t T,
                     int slices, double *out);

/* ---- .bed <-> FBM.code256 (the data formats either side of the path) ------
 * _bigsnpr_readbina2 (5 args) src/read-plink.cpp:61-80: decoded genotypes of the sub-matrix,
 * one byte each (0, 1, 2, 3 = missing), n x m column-major — the content of the .bk file that
 * snp_readBed / snp_readBed2 create (R/read-plink.R:27-111). */
int bsn_bed_to_fbm(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col, int64_t m,
                   uint8_t *out);
/* _bigsnpr_readbina (3 args) src/read-plink.cpp:13-56 (snp_readBed, R/read-plink.R:54): the WHOLE matrix of the
 * handle as FBM bytes, column-major n x m, every .bed byte decoded through the caller's 4 x 256 raw table
 * (tab[4 * byte + e] = FBM byte of genotype e of that byte; getCode(), R/utils.R:21-31).  The end-of-file flag
 * the reference returns is a property of the file, not of the payload: the R shim derives it from the file size. */
int bsn_bed_readbina(bsn_bed *bed, const uint8_t *tab, uint8_t *out);
/* _bigsnpr_writebina (5 args) src/write-plink.cpp:13-52: the .bed payload (ceil(n/4) bytes per
 * variant, pad bits 0, no magic header) of the sub-matrix [ind_row, ind_col] of the image;
 * snp_writeBed (R/write-plink.R:15-44) writes magic + this + .bim/.fam. */
int bsn_bed_subset_payload(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                           int64_t m, uint8_t *payload_out);

/* ---- device-resident operator (the fun.prod / fun.cprod seam of big_randomSVD,
 *      R/autoSVD.R:216-218, kept on the device so vectors never cross PCIe) ----- */
int bsn_op_create(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                  int64_t m, const double *center, const double *scale, bsn_op **out);
int bsn_op_destroy(bsn_op *op);
/* number of 8-bit slices the fp64 panels are split into (exact int8 MFMA arithmetic on
 * a fixed-point image of the vectors; 4 = 32-bit, 7 = 56-bit).  Default 4. */
int bsn_op_set_slices(bsn_op *op, int slices);
/* Y[n x nvec] = A~ X[m x nvec];  Z[m x nvec] = A~' X[n x nvec]; column-major, leading
 * dimensions ldx / ldy in elements; all pointers are DEVICE pointers */
int bsn_op_prod(bsn_op *op, const double *d_X, int64_t ldx, int nvec, double *d_Y, int64_t ldy);
int bsn_op_cprod(bsn_op *op, const double *d_X, int64_t ldx, int nvec, double *d_Z, int64_t ldz);
int bsn_op_sync(bsn_op *op);

/* ---- partial SVD (replaces bed_randomSVD -> bigstatsr::big_randomSVD -> RSpectra::svds,
 *      R/autoSVD.R:205-219; result fields of class "big_SVD": d, u, v, niter, nops) --------
 * Block Lanczos with full re-orthogonalisation on A~ A~' driven entirely on the device; basis
 * blocks are rounded to the fixed-point grid of the streaming products, which makes the products
 * exact, and the Ritz values come from the pair (Z'Z, Q'Q): d does not depend on `slices`.
 * `center`/`scale` are what fun.scaling returned (bed_scaleBinom, R/binom-scaling.R:133-142).
 * Multi-GPU: every rank calls this on its own column shard (ind_col local to its image),
 * passes m_total and an all-reduce hook that sums a device buffer of `count` doubles over
 * ranks in place (RCCL); u (n x k) is then identical on all ranks and v (m x k) is the
 * rank's own shard.  u / v may be NULL. */
typedef void (*bsn_allreduce_fn)(void *d_buf, int64_t count, void *ctx);

/* ---- multi-GPU: one process per GPU, variants (columns) sharded over the ranks -------------
 * north_star: "SNP columns shard naturally across the 8 GPUs of one node, with the SVD's panel
 * all-reduced over RCCL/xGMI".  A communicator wraps an RCCL (ncclComm_t) communicator created
 * on the CURRENT device; RCCL is loaded on first use (dlopen), single-GPU callers never need it.
 * Rank 0 calls bsn_comm_unique_id and hands the BSN_COMM_ID_BYTES bytes to the other ranks by
 * whatever means the host program has (MPI, a socket, a file); every rank then calls
 * bsn_comm_init (collective).  With `comm` set in bsn_svd_options, bsn_bed_randomsvd runs its
 * exchange inside the library: the n x b partial panel is reduce-scattered by sample blocks,
 * each rank orthogonalises its block of rows (small b x p coefficient all-reduces), and the
 * finished basis block is all-gathered for the next crossproduct pass — all on HIP streams the
 * library owns, the reduce-scatter overlapping the local Gram kernels.  There is no reference
 * counterpart: bigsnpr parallelises over OpenMP threads of one process (src/bed-prod-vec.cpp:29). */
#define BSN_COMM_ID_BYTES 128
typedef struct bsn_comm bsn_comm;
int bsn_comm_unique_id(uint8_t *id_out /* BSN_COMM_ID_BYTES */);
int bsn_comm_init(const uint8_t *id, int rank, int world, bsn_comm **out);
int bsn_comm_rank(const bsn_comm *comm);
int bsn_comm_world(const bsn_comm *comm);
/* sum of a device buffer of doubles over the ranks, in place, blocking (tests, host-side reductions) */
int bsn_comm_allreduce(bsn_comm *comm, double *d_buf, int64_t count);
/* ncclCommAbort: ends the collectives in flight on this communicator and makes every later call fail; the handle must
 * still be given to bsn_comm_destroy.  What the watchdog of a sharded solve calls (bsn_svd_options.exchange_timeout_ms);
 * exported for hosts that run their own. */
int bsn_comm_abort(bsn_comm *comm);
int bsn_comm_destroy(bsn_comm *comm);
typedef struct bsn_svd_options {
  int32_t k;          /* number of singular triplets (R default 10) */
  double tol;         /* relative residual on eigenvalues of A~A~' (R default 1e-4) */
  int32_t block;      /* vectors per pass, 1..16 (0 -> 16 / slices, at most 8: 8 at the default tol) */
  int32_t slices;     /* int8 slices of the fp64 panels, 1..7 (0 -> from tol and block: 2 at tol 1e-4
                         with block 8, 3 with block 5; the Ritz values do not depend on it) */
  int32_t max_basis;  /* cap on the Krylov basis (0 -> automatic) */
  uint32_t seed;      /* start block seed (0 -> 1) */
  int32_t verbose;
  int64_t m_total;    /* total number of columns over all ranks (0 -> m) */
  bsn_allreduce_fn allreduce; /* host-side hook summing a device buffer over ranks (tests); NULL otherwise */
  void *allreduce_ctx;
  int32_t hook_rank, hook_world; /* rank / number of ranks behind the hook (ignored with `comm`) */
  struct bsn_comm *comm; /* RCCL communicator of bsn_comm_init: the panel exchange then runs inside the
                            library on its own stream (NULL: one GPU, or the hook above) */
  /* binom_scaling = 1: fun.scaling is bed_scaleBinom (R/binom-scaling.R:133-142) evaluated INSIDE the
   * solve: the code counts ride along the first crossproduct pass (no separate statistics pass);
   * the `center` / `scale` arguments are ignored and the values used are written to center_out /
   * scale_out (length m each, may be NULL). */
  int32_t binom_scaling;
  double *center_out, *scale_out;
  /* warm start: power iterations of the random start block on the leading 1/16 of the variants before
   * the first full pass (each costs two streaming launches over that subset, 1/8 of a pass together).
   * 0 -> 2 (default since round 5; 1 before), n > 0 -> n, -1 -> none.  Matrices with fewer than 262 144 variants
   * (over all ranks) skip it. */
  int32_t warm_start;
  int32_t warm_denominator; /* the subset is the leading 1 / warm_denominator of the variants (0 -> 16) */
  /* A Krylov basis that fills up (max_basis) before the residuals meet tol is compressed to the k + block
   * best Ritz vectors and the iteration goes on (thick restart; RSpectra::svds behind the reference restarts
   * implicitly).  0 -> at most 100 restarts, n > 0 -> n, < 0 -> none: a full basis ends the solve with return
   * code 2. */
  int32_t max_restarts;
  /* Accuracy of the singular VECTORS (round 5).  The Ritz values do not see the rounding of the panels (exact-product
   * Rayleigh-Ritz), the vectors do: at `slices` digits every vector that converges early keeps a relative residual
   * of 1.2 * 2^(-8 slices) — 1.8e-5 at 16 bits, i.e. angles of 1e-4 .. 3e-4 to the true singular vectors, where the
   * reference's fp64 Lanczos leaves its leading vectors at 1e-7.  vec_floor is the residual floor wanted instead:
   * the early block steps (while the leading half of the k pairs is still far from converged) then run on wider
   * panels — 24 bits for the default — and the late ones, whose rounding enters a converged vector with the small
   * weight of its last components, stay narrow.  0 -> 1e-7 when `slices` is 0 (automatic; 2.5e-7 until round 5), none when the caller
   * fixed `slices`; > 0 -> that floor (digits up to 56 bits); < 0 -> none: every step at `slices` (round 4). */
  double vec_floor;
  /* Sharded solve (comm != NULL), round 5.  exchange_timing = 1: every collective of the solve is bracketed by HIP
   * events on the stream it runs on; bsn_svd_info.exchange_ms reports the sums.  exchange_timeout_ms > 0: a watchdog
   * thread aborts the communicator (ncclCommAbort) when the solve has not returned after that many milliseconds — a
   * collective that never completes (the first contact of the overlapped exchange with a real transport) then ends
   * the call with an error instead of hanging the process; the communicator is dead afterwards (destroy it, make a
   * new one, choose a more conservative exchange: BSN_NO_OVERLAP=1, BSN_NO_SEGMENTS=1 — bigsnpr_amd.comm.negotiate
   * does exactly that on a miniature solve).  0: the environment variable BSN_EXCHANGE_TIMEOUT_MS, else no watchdog. */
  int32_t exchange_timing;
  int32_t exchange_timeout_ms;
} bsn_svd_options;
typedef struct bsn_svd_info {
  int32_t niter;      /* block steps */
  int32_t nops;       /* streaming passes over the genotype image (A~ or A~' applications) */
  int32_t basis;      /* Krylov basis size at exit */
  int32_t converged;
  double max_rel_resid;
  double gpu_ms;      /* HIP-event time of the whole solve on the handle's stream */
  /* per-launch HIP-event time of the two streaming kernels inside this solve */
  double cprod_ms;    /* total over n_cprod launches of k_cprod (A~' panel) */
  double prod_ms;     /* total over n_prod launches of k_prod (A~ panel) */
  int32_t n_cprod, n_prod;
  int32_t block, slices; /* vectors per pass and int8 slices actually used */
  int32_t n_bad;      /* binom_scaling: variants with > 50 % missing values (src/bed-fun.cpp:40-41) */
  int32_t fused_stats;/* 1 if the scaling statistics rode along the first crossproduct pass */
  double cprod_stats_ms; /* the k_cprod launches that also counted the codes (not in cprod_ms) */
  int32_t n_cprod_stats;
  int32_t warm_launches; /* streaming launches of the warm start, each over warm_fraction of the variants
                            (they are counted in nops and in the kernel timings too) */
  double warm_fraction;
  double warm_ms;        /* HIP-event time of those launches (not in cprod_ms / prod_ms) */
  int32_t tiled;         /* 1 if the streaming kernels read the tiled second copy of the image (bsn_bed_tile); 2 if the
                            product passes read the sample-major second copy (two-block solves, round 4) */
  int32_t segmented_passes; /* sharded solve: product passes that ran in segments of sample blocks with their reduce-scatters
                            queued behind each segment (on a second stream unless BSN_NO_OVERLAP=1); round 4 */
  int32_t compact_gathers;  /* sharded solve: basis blocks all-gathered as the 16-bit integers they are rounded to (a quarter
                            of the fp64 volume, same values; BSN_NO_COMPACT_GATHER=1: fp64); round 4 */
  /* precision schedule (vec_floor): widest panels used, block steps that ran wider than `slices`, and the
   * streaming launches with THREE column blocks (48 digit columns: 16 vectors x 24 bits), timed apart from
   * cprod_ms / prod_ms */
  int32_t slices_max, wide_steps;
  double wide_cprod_ms, wide_prod_ms;
  int32_t n_wide_cprod, n_wide_prod;
  double lead_rel_resid;    /* residual estimate of the leading half of the k pairs at exit (max_rel_resid: all k) */
  /* sharded solve: how the product passes exchanged their panel — 0 no exchange, 1 whole pass + one reduce-scatter,
   * 2 segments of sample blocks on the solve's stream, 3 segments with the reduce-scatters on a second stream — and,
   * with exchange_timing, HIP-event time (ms) and number of its collectives by class: [0] reduce-scatter of the
   * panel / of its segments, [1] all-gather of a basis block or of u, [2] small all-reduces / all-gathers, [3] what
   * the solve's stream WAITED for the exchange stream (the part of the overlapped reduce-scatters left exposed) */
  int32_t exchange_mode;
  int32_t n_exchange[4];
  double exchange_ms[4];
  /* 1 when ind_col was not a contiguous range and the solve ran on a compacted copy of the selected variants (kept on
   * the handle, re-gathered in place by the next solve over another list: the rounds of bed_autoSVD); compact_ms =
   * host wall time of making / finding that copy inside this call (0.0x ms when the same list was solved before) */
  int32_t compacted;
  double compact_ms;
  /* 1 when the handle is out of core (bsn_bed_is_streamed): both passes of every block step walked the file in slabs
   * of variants through the resident slab image — PCIe-bound (the whole file per pass), same kernels, same arithmetic */
  int32_t out_of_core;
  /* share of the K-steps (1 024 genotypes each) of the two streaming products that hold NO missing code, sampled
   * once per image ([0] crossproduct, [1] product; 0 while not measured), and na_skip: bit 0 / bit 1 = the
   * crossproduct / the product passes of this solve ended on the kernels that issue the missing-value plane only
   * for steps that have one (used from a share of 0.30: nearly complete or batch-structured data; same integers) */
  double na_free_steps[2];
  int32_t na_skip;
} bsn_svd_info;
/* Returns 0 on success, 1 on error, and 2 when the solve ran to the end of its basis without all
 * k residuals meeting tol (outputs are filled with the best available triplets, bsn_last_error()
 * says how far they are; RSpectra warns in that case, so should the caller). */
int bsn_bed_randomsvd(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                      int64_t m, const double *center, const double *scale,
                      const bsn_svd_options *options, double *d, double *u, double *v,
                      bsn_svd_info *info);
/* Measurement hygiene: the names (as rocprofv3 prints them, without the argument list) of the streaming kernels the
 * last bsn_bed_randomsvd on this handle launched, one "kind=name" line per kind (cprod, prod, cprod_stats, warm).
 * bench.py matches them against profiles/pmc_traffic.json before it quotes that record's HBM traffic. */
int bsn_bed_streaming_kernels(bsn_bed *bed, char *buf, int64_t len);
/* Streaming layout.  The passes of bsn_bed_randomsvd / bsn_bed_prodvec / bsn_bed_cprodvec over a 64-aligned
 * contiguous range of variants run faster (about 8 %) on a second copy of the 2-bit image stored in tiles of
 * 64 variants x 1024 samples (16 KB): every load of a wavefront then lands in one contiguous run instead of 16 -
 * 64 rows a whole variant apart.  bsn_bed_randomsvd builds the copy by itself when the device has the room
 * (image size + 24 GB free; set BSN_NO_TILED=1 to forbid it); this call builds it ahead of time for the
 * one-shot products.  *built = 1 if the copy exists afterwards.  It costs one extra pass (read + write) once,
 * doubles the HBM held by the handle, and is freed by bsn_bed_close.  Results are bit-identical. */
int bsn_bed_tile(bsn_bed *bed, int *built);
/* Sample-major layout (round 4).  A second copy of the 2-bit image with the VARIANTS contiguous (sample i at
 * i * pitch, four variants per byte): on it the product A~ X of 9 .. 16 vectors (two MFMA column blocks) contracts
 * over the contiguous index like the crossproduct does on the variant-major image, and runs as the crossproduct's
 * kernel shape (no byte transposes, 16 accumulator registers instead of 128).  bsn_bed_randomsvd builds it by itself
 * for a solve on 16-vector blocks when the device has the room (image size + 24 GB free; BSN_NO_SMAJ=1 forbids it);
 * this call builds it ahead of time for bsn_op_prod / bsn_bed_prod* on 9 .. 16 vectors.  One extra read + write pass
 * once, doubles the HBM held by the handle, freed by bsn_bed_close / bsn_bed_release_workspace.  Results are
 * bit-identical (same exact integer sums). */
int bsn_bed_sample_major(bsn_bed *bed, int *built);

/* The device workspace of a solve (Krylov basis, panels, quantised operands: about
 * 8 (n + m) (8 k + 4 block) bytes, 4 GB at 400K x 1M, k = 20) stays on the handle for the next
 * solve (snp_autoSVD runs up to six in a row); this frees it early, together with the streaming-layout copy
 * of the image (bsn_bed_tile) and the cached work buffers of the device.  bsn_bed_close frees it too. */
int bsn_bed_release_workspace(bsn_bed *bed);

/* bed_tcrossprodSelf (R/bed-tcrossprodSelf.R:21-52): K[n x n] = A~ A~'; center / scale of length m as
 * returned by fun.scaling.  block_size is the reference's RAM block (R/bed-tcrossprodSelf.R:40-49) and is
 * ignored: the genotypes are decoded inside the fp64-MFMA kernel, nothing dense is materialised.  K is
 * column-major and exactly symmetric (the upper triangle is computed, the lower one mirrored). */
int bsn_bed_tcrossprod(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                       int64_t m, const double *center, const double *scale, int64_t block_size,
                       double *K);

/* ---- windowed LD (replaces corMat, ld_scores, clumping_chr, bed_clumping_chr) -----------
 * The handle may come from bsn_bed_open (.bed) or bsn_bed_from_fbm (FBM.code256 with NA),
 * which is the dispatch of src/corr.cpp:113-125.  `pos` has length m and must be sorted;
 * `size` is already in position units (R multiplies by 1000, R/corr.R:29).  A row list with repeated
 * samples is served from a gathered copy of the selected sub-matrix. */
typedef struct bsn_cor bsn_cor;
/* _bigsnpr_corMat (8 args) src/corr.cpp:102-126.  Two-phase: this call computes everything
 * on the device, fills p_out[m+1] (CSC column pointers exactly as R/corr.R:43-47 builds
 * them) and *nnz_out; bsn_cormat_fetch copies @i (ascending row index, diagonal last in each
 * column) and @x; thr has length n (thr[nona-1], src/corr.cpp:82). */
int bsn_cormat(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col, int64_t m,
               double size, const double *thr, const double *pos, int fill_diag, int32_t *p_out,
               int64_t *nnz_out, bsn_cor **out);
int bsn_cormat_fetch(bsn_cor *cor, int32_t *i_out, double *x_out);
/* *out = 1 when a stored correlation is NaN (a variant without variation among the selected samples): what
 * R/corr.R:53-54 finds with anyNA(corr@x), here noted by the kernel that writes @x instead of a pass over the
 * 12 bytes per pair on the host. */
int bsn_cormat_has_nan(const bsn_cor *cor, int *out);
int bsn_cormat_free(bsn_cor *cor);
/* measurement hook (bench.py --workload ld): figures of the last bsn_cormat / bsn_ld_scores /
 * bsn_clumping_* call of this process: out[0] = variant pairs in the band, out[1] = 64 x 64 tile
 * pairs, out[2] = total HIP-event time (ms) of the pair-statistics launches, out[3] = launches,
 * out[4] = kernel (0: six products, fused epilogue; 1: six products, K split; 2: cross product only) */
int bsn_ld_last_stats(double *out /* 5 */);
/* _bigsnpr_ld_scores (6 args) src/ld-scores.cpp:83-105 */
int bsn_ld_scores(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col, int64_t m,
                  double size, const double *pos, double *out);
/* _bigsnpr_clumping_chr (12 args) src/clumping.cpp:10-91 (mode 0: aux1 = sumX, aux2 = denoX)
 * and _bigsnpr_bed_clumping_chr (12 args) src/clumping-bed.cpp:11-91 (mode 1: aux1 = center,
 * aux2 = scale).  ordInd / rankInd are 0-based; keep[m] receives 0 / 1 (the reference's
 * `keep` FBM, R/clumping.R:116). */
int bsn_clumping_chr(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                     int64_t m, int mode, const double *aux1, const double *aux2,
                     const int32_t *ordInd, const int32_t *rankInd, const double *pos, double size,
                     double thr, int32_t *keep);
/* _bigsnpr_clumping_chr_cached (14 args) src/clumping-cached.cpp:11-107 as driven by the grid
 * loops of snp_grid_clumping (R/SCT.R:100-131): the same greedy clumping for
 * n_grid (size, thr) pairs over one column set.  The reference threads a sparse r2 cache through
 * the calls; here the r2 band is computed once at the largest window and every grid point sweeps
 * it.  keep[g * m + j] receives 0 / 1 for grid point g. */
int bsn_clumping_chr_cached(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                            int64_t m, int mode, const double *aux1, const double *aux2,
                            const int32_t *ordInd, const int32_t *rankInd, const double *pos,
                            int64_t n_grid, const double *sizes, const double *thrs, int32_t *keep);

/* ---- robust statistics of the outlier step of snp_autoSVD / bed_autoSVD (round 5) ----------------------------------
 * R/autoSVD.R:142-148,295-301: bigutilsr::dist_ogk(obj.svd$v) — the orthogonalised Gnanadesikan-Kettenring estimator
 * (Maronna & Zamar 2002; robustbase::covOGK with scaleTau2) — needs, per round, p + p (p - 1) robust scales of vectors as
 * long as the number of variants: two medians and two weighted sums each, the bulk of the function's wall time once its
 * solves run on the GPU.  These entry points compute them for whole batches of columns of a DEVICE matrix (column-major,
 * leading dimension ld >= m; bsn_malloc / bsn_memcpy_h2d): medians by radix select, sums in a fixed order.  The loop
 * around them (p x p eigen-decompositions, hard rejection, Mahalanobis distances) is host code (bigsnpr_amd/autosvd.py,
 * an R shim would keep bigutilsr's own).  Small host vectors in and out. */
/* robustbase::scaleTau2(x, c1, c2, mu.too = TRUE, consistency = TRUE) of every column: mu_out / s_out [ncol] (either may be NULL) */
int bsn_robust_scale_tau2(const double *d_X, int64_t m, int64_t ld, int32_t ncol, double c1, double c2, double *mu_out, double *s_out);
/* the same scale of Z_i + Z_j and Z_i - Z_j for every pair i > j of the p <= 64 columns, pairs in the order
 * (1,0), (2,0), (2,1), (3,0) ...: s_sum_out / s_diff_out [p (p - 1) / 2] */
int bsn_robust_pair_scales(const double *d_Z, int64_t m, int64_t ld, int32_t p, double c1, double c2, double *s_sum_out, double *s_diff_out);
/* medcouple of tukey_mc_up (Brys, Hubert & Struyf 2004): number of pairs (u, l), u from d_up [nu], l from the ASCENDING
 * d_lo [nl] (both positive distances to the median), with (u - l) / (u + l) <= t, -1 < t < 1 — the count the bisection
 * on t evaluates ~50 times per call */
int bsn_robust_mc_count(const double *d_up, int64_t nu, const double *d_lo, int64_t nl, double t, int64_t *count_out);
int bsn_robust_scale_cols(double *d_Z, int64_t m, int64_t ld, int32_t p, const double *div /* [p] */);      /* Z[, j] /= div[j] */
int bsn_robust_rotate(double *d_Z, int64_t m, int64_t ld, int32_t p, const double *E /* p x p, column-major */); /* Z <- Z E */
/* out[i] = sum_j ((Z[i, j] - mu[j]) / sig[j])^2, to the host */
int bsn_robust_wdist(const double *d_Z, int64_t m, int64_t ld, int32_t p, const double *mu, const double *sig, double *out);
/* (round 6) bigutilsr::dist_ogk end to end: squared robust Mahalanobis distances of the m rows of the DEVICE matrix d_U
 * (m x p, p <= 64, not modified) to the host vector dist_out [m] — the OGK rounds above (niter of them), the hard
 * rejection wdist <= median(wdist) * cut_ratio (cut_ratio = qchisq(beta, p) / qchisq(0.5, p), the caller's quantiles),
 * centre and covariance (denominator n_kept - 1) of the kept rows, distances with the pseudo-inverse of that covariance
 * (singular values <= 1e-15 x the largest dropped).  Only p x p matrices visit the host.  n_kept_out may be NULL.
 * R/autoSVD.R:142,295 (`bigutilsr::dist_ogk(obj.svd$v)`) */
int bsn_robust_dist_ogk(const double *d_U, int64_t m, int64_t ld, int32_t p, int32_t niter, double cut_ratio, double c1, double c2,
                        double *dist_out, int64_t *n_kept_out);
/* bigutilsr::rollmean of the HOST vector x [m] with the odd-length weight vector w [len] inside each of the consecutive
 * groups [group_off[g], group_off[g + 1]) (NULL / 0: one group); edge windows are renormalised by the weights they
 * contain.  R/autoSVD.R:143-144,296-297 (per chromosome) */
int bsn_robust_rollmean(const double *x, int64_t m, const double *w, int32_t len, const int64_t *group_off, int32_t ngroups, double *out);
/* ascending sort of a HOST vector of finite doubles in place (device radix sort): the order statistics of tukey_mc_up */
int bsn_robust_sort(double *x, int64_t m);
/* medcouple, the end of the bisection: the kernel values (u - l) / (u + l) in (a, b], -1 < a < b < 1, of the pairs counted
 * by bsn_robust_mc_count, to the host in no particular order; count_out receives their number and nothing is written when
 * it exceeds cap */
int bsn_robust_mc_window(const double *d_up, int64_t nu, const double *d_lo, int64_t nl, double a, double b, int64_t cap, double *out,
                         int64_t *count_out);

/* R/clumping.R:106, R/bed-clumping.R:48 — `ord <- order(S.chr, decreasing = TRUE)` — for every chromosome at once: S [m]
 * holds the statistics of the groups [group_off[g], group_off[g + 1]) one after the other (NULL / 0: one group); ord
 * receives, group by group, the 0-based positions INSIDE the group in decreasing order of S, ties in index order (R's
 * order is stable), rank its inverse (rank[group_off[g] + ord[group_off[g] + t]] = t): the ordInd / rankInd arguments of
 * bsn_clumping_chr.  Two stable device radix sorts (by value, then by group); a NaN is refused (the host path puts
 * them last, as R does). */
int bsn_order_decreasing(const double *S, int64_t m, const int64_t *group_off, int32_t ngroups, int32_t *ord, int32_t *rank);

/* ---- device memory + timing helpers for hosts without a HIP binding -------- */
int bsn_malloc(void **d_ptr, int64_t bytes);
int bsn_free(void *d_ptr);
/* Page-locked host memory.  Every entry point accepts ordinary (pageable) caller memory and stages it
 * through pinned buffers of its own; a result buffer obtained here is written by the DMA engines directly
 * (bsn_bed_randomsvd: u and v, 224 MB at 400K x 1M, k = 20 — no staging copy, no first-touch page faults).
 * The reference returns freshly allocated R vectors (R/autoSVD.R:216-218); an R shim has to copy into
 * those, a host that controls its allocations (the Python mirror) does not. */
int bsn_host_alloc(void **h_ptr, int64_t bytes);
int bsn_host_free(void *h_ptr);
int bsn_memcpy_h2d(void *d_dst, const void *src, int64_t bytes);
int bsn_memcpy_d2h(void *dst, const void *d_src, int64_t bytes);
int bsn_device_sync(void);
/* HIP events on the handle's stream (bench.py roofline timing) */
int bsn_timer_start(bsn_bed *bed);
int bsn_timer_stop(bsn_bed *bed, double *ms);

#ifdef __cplusplus
}
#endif
#endif /* BIGSNPR_HIP_H */
