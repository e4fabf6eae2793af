// tcross.hip — bed_tcrossprodSelf (R/bed-tcrossprodSelf.R:21-52): K = A~ A~' for the scaled, mean-imputed
// sub-matrix A~ (n samples x m variants), n x n doubles.
//
// The reference materialises A~ block by block in RAM (read_bed_scaled, src/bed-mat-acc.cpp:30-49) and
// hands the blocks to BLAS (K += A_b A_b').  Here the 2-bit codes are decoded inside the GEMM kernel: a
// workgroup owns one 128 x 128 tile of the UPPER triangle of K, walks over the variants in chunks of 16,
// decodes its two 128 x 16 operand tiles through a 4-entry table per variant ((g - c_j) / s_j for g = 0, 1, 2
// and 0 for a missing value, exactly the table of bedAccScaled, src/bed-acc.h:95-108) into LDS and feeds
// v_mfma_f64_16x16x4_f64.  A~ never exists in memory: the kernel reads 2 bits per genotype and the fp64 MFMA
// pipe is the bound (78.6 TFLOP/s); only the upper triangle is computed and mirrored.
//
// Few samples give few tiles: the variants are then split into slabs (grid.y), every slab writes its own
// partial K, and a second kernel adds the slabs in slab order — no atomics, the result does not depend on
// the schedule.
#include "bsn_internal.hpp"

namespace bsn {

typedef double v4d __attribute__((ext_vector_type(4)));

// T[4 j + g] = (g - center_j) / scale_j, g = 0, 1, 2; T[4 j + 3] = 0 (missing)
#pragma clang fp contract(off)
__global__ void k_scaled_table(const double *center, const double *scale, int64_t m, double *T) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const double c = center[j], s = scale[j];
  T[4 * j + 0] = (0.0 - c) / s;
  T[4 * j + 1] = (1.0 - c) / s;
  T[4 * j + 2] = (2.0 - c) / s;
  T[4 * j + 3] = 0.0;
}
#pragma clang fp contract(on)

constexpr int kTile = 128;        // samples per operand tile
constexpr int kChunk = 16;        // variants per pipeline stage (4 MFMA k-steps)
constexpr int kLdsRow = kTile + 16;  // doubles per k-row in LDS: 16 of padding keep the two half-waves of a
                                     // ds_read_b64 (k and k + 1) on disjoint banks

// rows: sample index of every selected row (nullptr: 0 .. n-1); cols: variant index of every selected variant
// (nullptr: col0 + j).  pairs[p] = (bi, bj), bi <= bj, tile coordinates.  Slab y handles the variants
// [y * m_slab, min(m, (y + 1) * m_slab)) and writes Kout + y * n * n (Kout itself when gridDim.y == 1).
//
// Decode of a chunk (2 tiles x 16 variants x 128 samples):
//   IDENT (rows 0 .. n-1): the 128 samples of a tile are 32 consecutive bytes of a variant row, so each of the
//   256 threads loads ONE 32-bit word (16 samples of one variant of one tile) and the variant's table once,
//   and writes its 16 values to LDS.  Lane L starts at sample (L >> 1) & 15 of its word and goes round: with
//   the 144-double row pitch the 64 lanes of a wave then hit every bank exactly twice per store.
//   gather (any ind.row): one byte load and one table load per value (thread = one sample of both tiles x 8
//   variants).
template <bool IDENT, int WAVES, int PRIO = 0>
__global__ __launch_bounds__(64 * WAVES, 2) void k_tcross(const uint8_t *__restrict__ img, int64_t pitch,
                                                          const int32_t *__restrict__ rows,
                                                          const int32_t *__restrict__ cols, int64_t col0, int64_t n,
                                                          int64_t m, int64_t m_slab, const double *__restrict__ T,
                                                          const int2 *__restrict__ pairs, double *__restrict__ Kout) {
  // WAVES = 4: 2 x 2 waves of 64 x 64 (16 MFMA tiles, 128 accumulator registers, 2 waves per SIMD);
  // WAVES = 8: 2 x 4 waves of 64 x 32 (8 tiles, 64 registers, 4 waves per SIMD: the decode of one wave and the
  // barrier of its workgroup hide behind the MFMAs of three others)
  constexpr int NT = 64 * WAVES;
  constexpr int WCOLS = WAVES / 2;         // waves along the columns of the tile
  constexpr int TB = (kTile / WCOLS) / 16;  // MFMA tiles per wave along the columns (4 or 2)
  constexpr int SPT = 2 * kChunk * kTile / NT;  // IDENT: samples decoded per thread and chunk (16 or 8)
  constexpr int VPT = kChunk * kTile / NT;      // gather: variants per thread and chunk (8 or 4)
  __shared__ double sA[2][kChunk * kLdsRow];
  __shared__ double sB[2][kChunk * kLdsRow];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r16 = lane & 15, kq = lane >> 4;
  const int wi = wave / WCOLS, wj = wave % WCOLS;
  const int2 pr = pairs[blockIdx.x];
  const int64_t i0 = (int64_t)pr.x * kTile, j0 = (int64_t)pr.y * kTile;
  const int64_t v_lo = (int64_t)blockIdx.y * m_slab;
  const int64_t v_hi = v_lo + m_slab < m ? v_lo + m_slab : m;
  if (v_lo >= v_hi) return;

  // ---- gather decode: this thread fills row `dr` of both operand tiles for VPT of the chunk's 16 variants
  const int dr = tid & (kTile - 1), dg = tid >> 7;
  int64_t offa = 0, offb = 0;
  int sha = 0, shb = 0;
  bool oka = false, okb = false;
  double va[VPT], vb[VPT];
  // ---- IDENT decode: SPT consecutive samples of variant `dq` of tile `dh` (0: rows i0.., 1: rows j0..)
  const int dh = tid / (NT / 2), tt = tid % (NT / 2);
  const int dq = tt / (NT / 32), dw = (tt % (NT / 32)) / (16 / SPT), dpart = tt % (16 / SPT);
  const int64_t rbase = (dh ? j0 : i0) + dw * 16;  // first sample of the thread's 32-bit word
  uint32_t word = 0;
  double t0 = 0, t1 = 0, t2 = 0;
  bool okw = false;
  if constexpr (!IDENT) {
    const int64_t ia = i0 + dr, ib = j0 + dr;
    oka = ia < n;
    okb = ib < n;
    const int64_t sa = (int64_t)rows[oka ? ia : n - 1], sb = (int64_t)rows[okb ? ib : n - 1];
    offa = sa >> 2;
    offb = sb >> 2;
    sha = (int)(sa & 3) * 2;
    shb = (int)(sb & 3) * 2;
  }
  auto fetch = [&](int64_t v0) {
    if constexpr (IDENT) {
      const int64_t v = v0 + dq;
      okw = v < v_hi && rbase < n;
      const int64_t vc = v < v_hi ? v : v_hi - 1;
      const int64_t col = cols ? (int64_t)cols[vc] : col0 + vc;
      // rbase < pitch * 4 always holds for a tile that starts below n (the pitch is padded to 256 B)
      word = *(const uint32_t *)(img + col * pitch + (rbase < n ? rbase >> 2 : 0));
      const double2 t01 = *(const double2 *)(T + 4 * vc);
      t0 = t01.x;
      t1 = t01.y;
      t2 = T[4 * vc + 2];
    } else {
#pragma unroll
      for (int q = 0; q < VPT; q++) {
        const int64_t v = v0 + dg * VPT + q;
        const bool okv = v < v_hi;
        const int64_t vc = okv ? v : v_hi - 1;
        const int64_t col = cols ? (int64_t)cols[vc] : col0 + vc;
        const uint8_t *rowp = img + col * pitch;
        const int ga = (rowp[offa] >> sha) & 3, gb = (rowp[offb] >> shb) & 3;
        const double ta = T[4 * vc + ga], tb = T[4 * vc + gb];
        va[q] = (okv && oka) ? ta : 0.0;
        vb[q] = (okv && okb) ? tb : 0.0;
      }
    }
  };
  auto stash = [&](int buf) {
    if constexpr (IDENT) {
      double *dst = (dh ? sB[buf] : sA[buf]) + dq * kLdsRow + dw * 16 + dpart * SPT;
      const int rot = lane >> 1;  // every store of a wave then hits each 8-B bank exactly twice
#pragma unroll
      for (int e = 0; e < SPT; e++) {
        const int x = (e + rot) & (SPT - 1), sx = dpart * SPT + x;
        const uint32_t code = (word >> (2 * sx)) & 3u;
        double val = code == 0u ? t0 : (code == 1u ? t1 : (code == 2u ? t2 : 0.0));
        if (!okw || rbase + sx >= n) val = 0.0;
        dst[x] = val;
      }
    } else {
#pragma unroll
      for (int q = 0; q < VPT; q++) {
        sA[buf][(dg * VPT + q) * kLdsRow + dr] = va[q];
        sB[buf][(dg * VPT + q) * kLdsRow + dr] = vb[q];
      }
    }
  };

  v4d acc[4][TB];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < TB; b++) acc[a][b] = v4d{0, 0, 0, 0};

  fetch(v_lo);
  stash(0);
  __syncthreads();
  int buf = 0;
  for (int64_t v0 = v_lo; v0 < v_hi; v0 += kChunk, buf ^= 1) {
    const bool more = v0 + kChunk < v_hi;
    if (more) fetch(v0 + kChunk);  // global loads of the next chunk fly under this chunk's MFMAs
#pragma unroll
    for (int ks = 0; ks < kChunk / 4; ks++) {
      const int k = ks * 4 + kq;
      double a[4], b[TB];
#pragma unroll
      for (int t = 0; t < 4; t++) a[t] = sA[buf][k * kLdsRow + wi * 64 + t * 16 + r16];
#pragma unroll
      for (int t = 0; t < TB; t++) b[t] = sB[buf][k * kLdsRow + wj * (16 * TB) + t * 16 + r16];
      if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);   // (experiment, profiling build: BSN_TCROSS_PRIO=1)
#pragma unroll
      for (int ta = 0; ta < 4; ta++)
#pragma unroll
        for (int tb = 0; tb < TB; tb++)
          acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
      if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(0);
    }
    if (more) stash(buf ^ 1);
    __syncthreads();
  }

  // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
  double *Kp = Kout + (int64_t)blockIdx.y * n * n;
  const bool mirror = pr.x != pr.y;
#pragma unroll
  for (int ta = 0; ta < 4; ta++)
#pragma unroll
    for (int tb = 0; tb < TB; tb++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int64_t i = i0 + wi * 64 + ta * 16 + kq + 4 * r, j = j0 + wj * (16 * TB) + tb * 16 + r16;
        if (i < n && j < n) {
          Kp[i + j * n] = acc[ta][tb][r];
          if (mirror) Kp[j + i * n] = acc[ta][tb][r];
        }
      }
}

// K = sum of the slabs, in slab order
__global__ void k_tcross_reduce(const double *__restrict__ part, int64_t nn, int nslab, double *__restrict__ K) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nn) return;
  double s = part[t];
  for (int y = 1; y < nslab; y++) s += part[(int64_t)y * nn + t];
  K[t] = s;
}

}  // namespace bsn

using namespace bsn;

__global__ void k_add_into(double *__restrict__ dst, const double *__restrict__ src, int64_t count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) dst[i] += src[i];
}

// K (n x n, device) = A~ A~' over the selected variants of a RESIDENT image
static void tcross_resident(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col, int64_t m,
                            const double *center, const double *scale, DevBuf<double> &d_K) {
  {
    bsn_op op;
    fill_op(&op, bed, ind_row, n, ind_col, m, center, scale);
    DevBuf<double> d_T, d_part;
    d_T.ensure((size_t)4 * m);
    hipLaunchKernelGGL(k_scaled_table, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, bed->stream, op.d_center.p,
                       op.d_scale.p, m, d_T.p);
    BSN_HIP(hipGetLastError());
    const int64_t nt = (n + kTile - 1) / kTile;
    std::vector<int2> pairs;
    pairs.reserve((size_t)(nt * (nt + 1) / 2));
    for (int64_t bj = 0; bj < nt; bj++)
      for (int64_t bi = 0; bi <= bj; bi++) pairs.push_back(int2{(int)bi, (int)bj});
    DevBuf<int2> d_pairs;
    copy_h2d(bed, d_pairs.ensure(pairs.size()), pairs.data(), pairs.size() * sizeof(int2));
    // slabs of variants when there are too few tiles to fill 256 CUs x 2 workgroups
    int64_t nslab = (int64_t)((1024 + pairs.size() - 1) / pairs.size());
    nslab = std::min<int64_t>(nslab, 32);
    nslab = std::min<int64_t>(nslab, (m + 255) / 256);
    while (nslab > 1 && (double)nslab * (double)n * (double)n * 8.0 > 4e9) nslab--;
    if (nslab < 1) nslab = 1;
    const int64_t m_slab = ((m + nslab - 1) / nslab + kChunk - 1) / kChunk * kChunk;
    nslab = (m + m_slab - 1) / m_slab;
    d_K.ensure((size_t)n * n);
    double *out = d_K.p;
    if (nslab > 1) out = d_part.ensure((size_t)nslab * n * n);
    const dim3 grid((unsigned)pairs.size(), (unsigned)nslab);
    const int32_t *cols = op.cols_contig ? nullptr : op.d_cols.p;
    // BSN_TCROSS_WAVES=4 selects the 4-wave shape (A/B measurements)
    const char *we = abl_getenv("BSN_TCROSS_WAVES");
    const bool w4 = we && atoi(we) == 4;
#define BSN_TCROSS(IDENTV, WV, ROWS)                                                                              \
  hipLaunchKernelGGL((k_tcross<IDENTV, WV>), grid, dim3(64 * WV), 0, bed->stream, bed->d_img, bed->pitch, ROWS, \
                     cols, op.col0, n, m, m_slab, d_T.p, d_pairs.p, out)
    if (op.rows_identity && !w4 && abl_getenv("BSN_TCROSS_PRIO")) {
      hipLaunchKernelGGL((k_tcross<true, 8, 2>), grid, dim3(64 * 8), 0, bed->stream, bed->d_img, bed->pitch, (const int32_t *)nullptr,
                         cols, op.col0, n, m, m_slab, d_T.p, d_pairs.p, out);
    } else if (op.rows_identity) {
      if (w4) BSN_TCROSS(true, 4, nullptr);
      else BSN_TCROSS(true, 8, nullptr);
    } else {
      if (w4) BSN_TCROSS(false, 4, op.d_rows.p);
      else BSN_TCROSS(false, 8, op.d_rows.p);
    }
#undef BSN_TCROSS
    BSN_HIP(hipGetLastError());
    if (nslab > 1) {
      hipLaunchKernelGGL(k_tcross_reduce, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, bed->stream, d_part.p,
                         n * n, (int)nslab, d_K.p);
      BSN_HIP(hipGetLastError());
    }
    BSN_HIP(hipStreamSynchronize(bed->stream));   // (the operator and the work buffers are released on return)
  }
}

extern "C" int bsn_bed_tcrossprod(bsn_bed *bed, const int64_t *ind_row, int64_t n, const int64_t *ind_col,
                                  int64_t m, const double *center, const double *scale, int64_t block_size,
                                  double *K) {
  return guarded([&] {
    (void)block_size;  // the reference's RAM block size: nothing is materialised here
    require_bits(bed, 2, "bed_tcrossprodSelf");
    if ((double)n * (double)n * 8.0 > 64e9) fail("n x n result does not fit: use bed_randomSVD");
    if (n <= 0 || m <= 0) fail("'ind.row' and 'ind.col' can't be empty.");
    DevBuf<double> d_K;
    if (!bed->streamed()) {
      tcross_resident(bed, ind_row, n, ind_col, m, center, scale, d_K);
      copy_d2h(bed, K, d_K.p, (size_t)n * n * 8);
      return;
    }
    // (round 5) out-of-core handle: K is a sum over the variants — the file is walked in slabs, every slab adds the
    // K of its selected variants on the device (R/bed-tcrossprodSelf.R:40-49 walks blocks of columns just so)
    BSN_HIP(hipSetDevice(bed->device));
    const int64_t nslab = slab_count(bed);
    std::vector<std::vector<int64_t>> pos((size_t)nslab);
    for (int64_t t = 0; t < m; t++) {
      const int64_t c = ind_col ? ind_col[t] : t;
      if (c < 0 || c >= bed->m) fail("Tested %lld < %lld. Subscript out of bounds (ind.col).", (long long)c, (long long)bed->m);
      pos[(size_t)(c / bed->slab_cols)].push_back(t);
    }
    DevBuf<double> d_sum;
    d_sum.ensure((size_t)n * n);
    BSN_HIP(hipMemsetAsync(d_sum.p, 0, (size_t)n * n * 8, bed->stream));
    std::vector<int64_t> loc;
    std::vector<double> ce, sc;
    for (int64_t sl = 0; sl < nslab; sl++) {
      const std::vector<int64_t> &P = pos[(size_t)sl];
      if (P.empty()) continue;
      int64_t j0 = 0;
      slab_upload(bed, sl, &j0);
      loc.resize(P.size());
      ce.resize(P.size());
      sc.resize(P.size());
      for (size_t t = 0; t < P.size(); t++) {
        loc[t] = (ind_col ? ind_col[P[t]] : P[t]) - j0;
        ce[t] = center ? center[P[t]] : 0.0;
        sc[t] = scale ? scale[P[t]] : 1.0;
      }
      tcross_resident(slab_image(bed), ind_row, n, loc.data(), (int64_t)loc.size(), ce.data(), sc.data(), d_K);
      hipLaunchKernelGGL(k_add_into, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, bed->stream, d_sum.p, d_K.p, n * n);
      BSN_HIP(hipGetLastError());
    }
    copy_d2h(bed, K, d_sum.p, (size_t)n * n * 8);
  });
}
