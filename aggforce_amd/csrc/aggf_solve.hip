// K2: equality-constrained least squares for all coarse-grained sites at once.
//
// Replaces the per-site qpsolvers/OSQP loop of the reference (qp/qplinear.py:76-86,
// featlinearmap.py:370-381):   min 1/2 x'Px  s.t.  Ax = b_i   for every column b_i.
// With P~ = P/s + A'A (same minimiser on the feasible set, positive definite whenever P
// is positive definite on null(A)):
//     L L' = P~,  Y = L^-1 A',  S = Y'Y = A P~^-1 A',  x = L^-T Y S^-1 b
// followed by one refinement step on the constraint residual.  Everything is fp64 and
// stays on the device: a blocked right-looking Cholesky (64-wide panels: LDS diagonal
// factor + explicit 64x64 inverse, then MFMA GEMMs for the panel and the trailing
// update) and triangular solves expressed as GEMMs with the inverted diagonal blocks.
// All internal matrices are padded to multiples of 64 so that the GEMM kernel needs no
// bounds checks.
#include "aggf_common.h"

namespace aggf {

constexpr int NB = 64;          // Cholesky panel width == GEMM tile edge
constexpr int GK = 16;          // GEMM K chunk
constexpr int GS = GK + 2;      // LDS row stride of GEMM operand tiles (elements)

// C[M x N] = alpha * op(A) op(B) + beta * C  (and the same values to C2 if non-null).
// Row-major; M % 64 == 0, N % 64 == 0, K % 16 == 0.  op(A) is M x K, op(B) is K x N.
// lower_only: skip tiles strictly above the block diagonal (square trailing updates).
// blockIdx.z = batch index z: operand X is offset by (z / inner) * x_o + (z % inner) * x_i elements.
struct GemmBatch {
  int inner;
  int64_t a_o, a_i, b_o, b_i, c_o, c_i;
};

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm64_kernel(int K, double alpha,
                                                        const double* A, int64_t lda,
                                                        const double* B, int64_t ldb,
                                                        double beta, double* C, int64_t ldc,
                                                        double* C2, int64_t ldc2, int lower_only,
                                                        GemmBatch bt) {
  using MF = Mfma<double>;
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (lower_only && bj > bi) return;
  if (gridDim.z > 1) {
    const int zo = blockIdx.z / bt.inner, zi = blockIdx.z % bt.inner;
    A += zo * bt.a_o + zi * bt.a_i;
    B += zo * bt.b_o + zi * bt.b_i;
    C += zo * bt.c_o + zi * bt.c_i;
  }
  __shared__ __attribute__((aligned(16))) double sA[2][64 * GS];
  __shared__ __attribute__((aligned(16))) double sB[2][64 * GS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t i0 = (int64_t)bi * 64, j0 = (int64_t)bj * 64;

  double ra[4], rb[4];
  auto load_stage = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 256 * q;
      if (TA) {  // op(A)[i][k] = A[k][i]: contiguous along i
        const int k = e >> 6, i = e & 63;
        ra[q] = A[(int64_t)(k0 + k) * lda + i0 + i];
      } else {   // A[i][k]: contiguous along k
        const int i = e >> 4, k = e & 15;
        ra[q] = A[(i0 + i) * lda + k0 + k];
      }
      if (TB) {  // op(B)[k][j] = B[j][k]: contiguous along k
        const int j = e >> 4, k = e & 15;
        rb[q] = B[(j0 + j) * ldb + k0 + k];
      } else {   // B[k][j]: contiguous along j
        const int k = e >> 6, j = e & 63;
        rb[q] = B[(int64_t)(k0 + k) * ldb + j0 + j];
      }
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 256 * q;
      if (TA) {
        const int k = e >> 6, i = e & 63;
        sA[buf][i * GS + k] = ra[q];
      } else {
        const int i = e >> 4, k = e & 15;
        sA[buf][i * GS + k] = ra[q];
      }
      if (TB) {
        const int j = e >> 4, k = e & 15;
        sB[buf][j * GS + k] = rb[q];
      } else {
        const int k = e >> 6, j = e & 63;
        sB[buf][j * GS + k] = rb[q];
      }
    }
  };

  f64x4 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] = acc_zero<double>();

  const int offA = (wm * 32 + (lane & 15)) * GS + (lane >> 4);
  const int offB = (wn * 32 + (lane & 15)) * GS + (lane >> 4);
  const int n_stage = K / GK;
  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int s = 0; s < n_stage; ++s) {
    const int cur = s & 1;
    if (s + 1 < n_stage) load_stage((s + 1) * GK);
#pragma unroll
    for (int kk = 0; kk < GK / 4; ++kk) {
      double a[2], b[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) a[m] = sA[cur][offA + 16 * m * GS + 4 * kk];
#pragma unroll
      for (int n = 0; n < 2; ++n) b[n] = sB[cur][offB + 16 * n * GS + 4 * kk];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = MF::mma(a[m], b[n], acc[m][n]);
    }
    if (s + 1 < n_stage) store_stage(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = i0 + wm * 32 + m * 16 + MF::row(lane, r);
        const int64_t col = j0 + wn * 32 + n * 16 + (lane & 15);
        double v = alpha * acc[m][n][r];
        if (beta != 0.0) v += beta * C[row * ldc + col];
        C[row * ldc + col] = v;
        if (C2) C2[row * ldc2 + col] = v;
      }
}

// Factor one 64x64 diagonal block: A_kk = L L' (lower), and Linv = L^-1.
// info[0]: 1-based global index of the first non-positive pivot (0 = ok).
//
// ONE wave, the block in registers: lane i holds row i (64 doubles); the loops are fully unrolled,
// so every register index is static.  Column j: pivot by v_readlane from lane j, the scaled column
// goes through a 64-entry LDS buffer and comes back as uniform-address (broadcast) reads for the
// rank-1 update -- one LDS round trip per column and no multi-wave barrier (the 256-thread LDS
// version spent ~1.2 us per column, 80 us per block, 5.4 ms of a 12 ms solve at n = 4096; reading
// the column with v_readlane instead costs two SGPR hazards per element).  The inverse is a forward
// substitution with lane c holding column c of L^-1 and L[i][k] broadcast from LDS.
__device__ __forceinline__ double lane_bcast(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(64) void potrf_diag_kernel(double* __restrict__ Akk, int64_t lda,
                                                        double* __restrict__ Linv,
                                                        double* __restrict__ info, int pivot_base) {
  __shared__ double a[NB][NB + 1];
  __shared__ double colbuf[2][NB];
  const int lane = threadIdx.x;
  for (int row = 0; row < NB; ++row) a[row][lane] = (lane <= row) ? Akk[(int64_t)row * lda + lane] : 0.0;
  __syncthreads();
  double r[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) r[k] = a[lane][k];
  double rinv = 0.0;  // lane j keeps 1 / L[j][j]
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    double d = lane_bcast(r[j], j);
    if (!(d > 0.0)) {
      if (lane == 0 && info[0] == 0.0) info[0] = (double)(pivot_base + j + 1);
      d = 1.0;
    }
    // 1/sqrt(d): hardware estimate + two Newton steps (the IEEE divide + sqrt sequence is ~4x longer
    // and sits on the 64-step dependency chain)
    double rs = __builtin_amdgcn_rsq(d);
    double e = fma(-d * rs, rs, 1.0);
    rs = fma(rs * e, fma(e, 0.375, 0.5), rs);
    e = fma(-d * rs, rs, 1.0);
    rs = fma(rs * 0.5, e, rs);
    r[j] = (lane == j) ? d * rs : r[j] * rs;  // column j of L (rows above the diagonal: unused values)
    if (lane == j) rinv = rs;
    colbuf[j & 1][lane] = r[j];
    __syncthreads();  // one wave: orders the LDS write before the broadcast reads
#pragma unroll
    for (int k = j + 1; k < NB; ++k)
      r[k] = fma(-r[j], colbuf[j & 1][k], r[k]);  // A[i][k] -= L[i][j] L[k][j]; only rows i >= k are ever read
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NB; ++k) a[lane][k] = (k <= lane) ? r[k] : 0.0;
  __syncthreads();
  for (int row = 0; row < NB; ++row)
    if (lane <= row) Akk[(int64_t)row * lda + lane] = a[row][lane];
  // X = L^-1 by forward substitution; x[k] of lane c is X[k][c] (zero for k < c by construction)
  double x[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    double sum = (lane == i) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < i; ++k) sum = fma(-a[i][k], x[k], sum);
    x[i] = sum * lane_bcast(rinv, i);
    Linv[i * NB + lane] = x[i];
  }
}

// scale[0] = max_i (G[i,i] + l2*diag[i]), or 1 if that is not positive/finite
__global__ __launch_bounds__(256) void max_diag_kernel(const double* __restrict__ G, int n,
                                                       double l2, const double* __restrict__ l2d,
                                                       double* __restrict__ scale) {
  __shared__ double sh[256];
  double m = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double v = G[(int64_t)i * n + i] + l2 * (l2d ? l2d[i] : 1.0);
    if (v > m) m = v;
  }
  sh[threadIdx.x] = m;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + w]);
    __syncthreads();
  }
  if (threadIdx.x == 0) scale[0] = (sh[0] > 0.0 && sh[0] < 1e300) ? sh[0] : 1.0;
}

// Pt (npad x npad) = (G + l2*diag)/s on the n x n block, identity on the padding
__global__ __launch_bounds__(256) void build_pt_kernel(const double* __restrict__ G, int n, int npad,
                                                       double l2, const double* __restrict__ l2d,
                                                       const double* __restrict__ scale,
                                                       double* __restrict__ Pt) {
  const double inv_s = 1.0 / scale[0];
  const int64_t total = (int64_t)npad * npad;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / npad), j = (int)(e - (int64_t)i * npad);
    double v;
    if (i < n && j < n) {
      v = G[(int64_t)i * n + j];
      if (i == j) v += l2 * (l2d ? l2d[i] : 1.0);
      v *= inv_s;
    } else {
      v = (i == j) ? 1.0 : 0.0;
    }
    Pt[e] = v;
  }
}

// dst (rd x cd, zero padded) = src (rs x cs) or its transpose; identity if src == NULL
__global__ __launch_bounds__(256) void pad_copy_kernel(const double* __restrict__ src, int rs, int cs,
                                                       int transpose, double* __restrict__ dst,
                                                       int rd, int cd) {
  const int64_t total = (int64_t)rd * cd;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / cd), j = (int)(e - (int64_t)i * cd);
    double v = 0.0;
    if (!src) {
      v = (i == j && i < rs) ? 1.0 : 0.0;
    } else if (!transpose) {
      if (i < rs && j < cs) v = src[(int64_t)i * cs + j];
    } else {
      if (j < rs && i < cs) v = src[(int64_t)j * cs + i];
    }
    dst[e] = v;
  }
}

// S[i,i] = 1 for padding rows i >= m
__global__ void fix_pad_diag_kernel(double* __restrict__ S, int m, int mpad) {
  const int i = m + blockIdx.x * blockDim.x + threadIdx.x;
  if (i < mpad) S[(int64_t)i * mpad + i] = 1.0;
}

// R -= Bp elementwise on (rows x cols), then out[0] = max |R| over the valid block
__global__ __launch_bounds__(256) void resid_kernel(double* __restrict__ R, const double* __restrict__ Bp,
                                                    int rows, int cols, int ld, double* __restrict__ out) {
  __shared__ double sh[256];
  double m = 0.0;
  const int64_t total = (int64_t)rows * cols;
  for (int64_t e = threadIdx.x; e < total; e += 256) {
    const int i = (int)(e / cols), j = (int)(e - (int64_t)i * cols);
    const double v = R[(int64_t)i * ld + j] - Bp[(int64_t)i * ld + j];
    R[(int64_t)i * ld + j] = v;
    const double av = fabs(v);
    if (av > m || v != v) m = (v != v) ? INFINITY : av;
  }
  sh[threadIdx.x] = m;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + w]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}

__global__ __launch_bounds__(256) void axpy_kernel(double* __restrict__ x, const double* __restrict__ y,
                                                   double a, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x)
    x[e] += a * y[e];
}

// X (nrhs x n) = Xt (npad x rpad) transposed and cropped
__global__ __launch_bounds__(256) void crop_transpose_kernel(const double* __restrict__ Xt, int rpad,
                                                             int n, int nrhs, double* __restrict__ X) {
  const int64_t total = (int64_t)nrhs * n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / n), i = (int)(e - (int64_t)r * n);
    X[e] = Xt[(int64_t)i * rpad + r];
  }
}

// S[i,i] += reg * trace(S[:m,:m]) / m  for i < m  (Tikhonov shift for redundant constraint rows)
__global__ __launch_bounds__(256) void schur_reg_kernel(double* __restrict__ S, int m, int mpad, double reg) {
  __shared__ double sh[256];
  double t = 0.0;
  for (int i = threadIdx.x; i < m; i += 256) t += S[(int64_t)i * mpad + i];
  sh[threadIdx.x] = t;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  const double shift = reg * sh[0] / m;
  for (int i = threadIdx.x; i < m; i += 256) S[(int64_t)i * mpad + i] += shift;
}

__global__ void init_stats_kernel(double* stats) {
  if (threadIdx.x < 4) stats[threadIdx.x] = 0.0;
}
__global__ void copy_scalar_kernel(const double* src, double* dst) { dst[0] = src[0]; }

// ---------------------------------------------------------------------------
struct Ctx {
  hipStream_t stream;
  int rc = AGGF_OK;
};

static inline dim3 flat_grid(int64_t n) {
  int64_t g = ceil_div(n, 256);
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return dim3((unsigned)g);
}

template <bool TA, bool TB>
static void gemm(Ctx& c, int M, int N, int K, double alpha, const double* A, int64_t lda,
                 const double* B, int64_t ldb, double beta, double* C, int64_t ldc,
                 double* C2 = nullptr, int64_t ldc2 = 0, int lower_only = 0, int nbatch = 1,
                 GemmBatch bt = GemmBatch{1, 0, 0, 0, 0, 0, 0}) {
  if (c.rc || M <= 0 || N <= 0 || nbatch <= 0) return;
  hipLaunchKernelGGL((gemm64_kernel<TA, TB>), dim3(N / 64, M / 64, nbatch), dim3(256), 0, c.stream, K, alpha,
                     A, lda, B, ldb, beta, C, ldc, C2, ldc2, lower_only, bt);
  if (hipGetLastError() != hipSuccess) c.rc = fail(AGGF_ERR_HIP, "gemm launch failed");
}

// in-place lower Cholesky of the npad x npad matrix P (ld = npad); Dinv: npad/64 blocks.
// Two-level blocking: 64-wide inner panels (the diagonal block kernel's size) inside 256-wide outer
// panels.  Inner steps update only the columns of their outer panel; the matrix to the right of it
// is updated once per outer panel with K = 256, so the trailing matrix -- the n^3/3 part -- is read
// and written 4x less often than with K = 64 updates (those ran at ~4 TF, bound by the C traffic).
constexpr int OUTER_PANELS = 4;
constexpr int BIG = OUTER_PANELS * NB;  // 256: edge of the inverted diagonal blocks used by the solves

// elements of the "Dinv" workspace of an npad x npad factor: [npad/64 inverses of 64x64 diagonal
// blocks | npad/256 inverses of 256x256 diagonal blocks | scratch for building them]
static size_t dinv_elems(int npad) {
  return (size_t)npad * NB + (size_t)(npad / BIG) * BIG * BIG + (size_t)(npad / BIG + 1) * (BIG / 2) * (BIG / 2);
}
static inline double* dbig_of(double* Dinv, int npad) { return Dinv + (size_t)npad * NB; }
static inline const double* dbig_of(const double* Dinv, int npad) { return Dinv + (size_t)npad * NB; }

// Dbig[ob] (256x256) <- block diagonal of the four 64x64 inverses of outer block ob, zeros elsewhere
__global__ __launch_bounds__(256) void dbig_init_kernel(const double* __restrict__ Dinv, double* __restrict__ Dbig,
                                                        int nbig) {
  const int64_t total = (int64_t)nbig * BIG * BIG;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int ob = (int)(e / (BIG * BIG)), r = (int)(e % (BIG * BIG)), i = r / BIG, j = r % BIG;
    Dbig[e] = (i / NB == j / NB) ? Dinv[((int64_t)(ob * OUTER_PANELS + i / NB) * NB + i % NB) * NB + j % NB] : 0.0;
  }
}

// Inverses of the 256x256 diagonal blocks of L by block doubling,
//   [[A,0],[B,C]]^-1 = [[A^-1,0],[-C^-1 B A^-1, C^-1]],
// 64 -> 128 -> 256, every level two batched GEMMs over all blocks.  With them a triangular solve
// takes npad/256 steps of K = 256 GEMMs instead of npad/64 steps of K = 64 (those were bound by
// the ~128 dependent launches per solve, not by their flops).
static void build_big_inverses(Ctx& c, const double* L, int npad, double* Dinv) {
  const int nbig = npad / BIG;
  if (nbig <= 0 || c.rc) return;
  double* Dbig = dbig_of(Dinv, npad);
  double* T = Dbig + (size_t)nbig * BIG * BIG;
  hipLaunchKernelGGL(dbig_init_kernel, flat_grid((int64_t)nbig * BIG * BIG), dim3(256), 0, c.stream, Dinv, Dbig, nbig);
  if (hipGetLastError() != hipSuccess) { c.rc = fail(AGGF_ERR_HIP, "dbig_init launch failed"); return; }
  const int64_t l_o = (int64_t)BIG * npad + BIG, d_o = (int64_t)BIG * BIG, t_o = (int64_t)(BIG / 2) * (BIG / 2);
  // level 1: pairs of 64-blocks (p = 0, 1) inside every outer block
  {
    const int64_t l_i = (int64_t)2 * NB * npad + 2 * NB, d_i = (int64_t)2 * NB * BIG + 2 * NB;
    // T = B Ainv
    gemm<false, false>(c, NB, NB, NB, 1.0, L + (int64_t)NB * npad, npad, Dbig, BIG, 0.0, T, NB, nullptr, 0, 0,
                       2 * nbig, GemmBatch{2, l_o, l_i, d_o, d_i, t_o, (int64_t)NB * NB});
    // X = -Cinv T
    gemm<false, false>(c, NB, NB, NB, -1.0, Dbig + (int64_t)NB * BIG + NB, BIG, T, NB, 0.0, Dbig + (int64_t)NB * BIG, BIG,
                       nullptr, 0, 0, 2 * nbig, GemmBatch{2, d_o, d_i, t_o, (int64_t)NB * NB, d_o, d_i});
  }
  // level 2: the two 128-blocks of every outer block
  {
    const int H = BIG / 2;
    gemm<false, false>(c, H, H, H, 1.0, L + (int64_t)H * npad, npad, Dbig, BIG, 0.0, T, H, nullptr, 0, 0, nbig,
                       GemmBatch{1, l_o, 0, d_o, 0, t_o, 0});
    gemm<false, false>(c, H, H, H, -1.0, Dbig + (int64_t)H * BIG + H, BIG, T, H, 0.0, Dbig + (int64_t)H * BIG, BIG,
                       nullptr, 0, 0, nbig, GemmBatch{1, d_o, 0, t_o, 0, d_o, 0});
  }
}

static void cholesky(Ctx& c, double* P, int npad, double* Dinv, double* info, int pivot_base) {
  const int nb = npad / NB;
  for (int k0 = 0; k0 < nb && !c.rc; k0 += OUTER_PANELS) {
    const int kend = k0 + OUTER_PANELS < nb ? k0 + OUTER_PANELS : nb;
    for (int k = k0; k < kend && !c.rc; ++k) {
      double* Akk = P + (int64_t)k * NB * npad + (int64_t)k * NB;
      double* Dk = Dinv + (int64_t)k * NB * NB;
      hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(64), 0, c.stream, Akk, (int64_t)npad, Dk,
                         info, pivot_base + k * NB);
      if (hipGetLastError() != hipSuccess) c.rc = fail(AGGF_ERR_HIP, "potrf launch failed");
      const int rem = npad - (k + 1) * NB;
      if (rem <= 0) break;
      double* panel = Akk + (int64_t)NB * npad;  // rows below the diagonal block, same columns
      // panel <- panel * Linv'  (in place: every workgroup reads exactly the rows it writes)
      gemm<false, true>(c, rem, NB, NB, 1.0, panel, npad, Dk, NB, 0.0, panel, npad);
      // remaining columns of this outer panel <- themselves - panel panel'   (lower tiles only)
      const int inner_cols = (kend - 1 - k) * NB;
      if (inner_cols > 0)
        gemm<false, true>(c, rem, inner_cols, NB, -1.0, panel, npad, panel, npad, 1.0, panel + NB, npad, nullptr, 0, 1);
    }
    const int rem2 = npad - kend * NB;
    if (rem2 > 0) {
      // trailing <- trailing - L21 L21' with all columns of the outer panel at once
      const int kw = (kend - k0) * NB;
      const double* L21 = P + (int64_t)kend * NB * npad + (int64_t)k0 * NB;
      double* A22 = P + (int64_t)kend * NB * npad + (int64_t)kend * NB;
      gemm<false, true>(c, rem2, rem2, kw, -1.0, L21, npad, L21, npad, 1.0, A22, npad, nullptr, 0, 1);
    }
  }
  build_big_inverses(c, P, npad, Dinv);
}

// Y = L^-1 Bw ; Bw (npad x w, ld = w) is consumed.  Full 256-row blocks use their inverted diagonal
// block, the (< 256 rows) remainder the 64x64 inverses.
static void solve_lower(Ctx& c, const double* L, int npad, const double* Dinv, double* Bw, double* Y,
                        int w) {
  const int nb = npad / NB, nbig = npad / BIG;
  const double* Dbig = dbig_of(Dinv, npad);
  for (int ob = 0; ob < nbig && !c.rc; ++ob) {
    const int64_t r0 = (int64_t)ob * BIG;
    gemm<false, false>(c, BIG, w, BIG, 1.0, Dbig + (int64_t)ob * BIG * BIG, BIG, Bw + r0 * w, w, 0.0, Y + r0 * w, w);
    const int rem = npad - (int)(r0 + BIG);
    if (rem > 0)
      gemm<false, false>(c, rem, w, BIG, -1.0, L + (r0 + BIG) * npad + r0, npad, Y + r0 * w, w, 1.0,
                         Bw + (r0 + BIG) * w, w);
  }
  for (int k = nbig * OUTER_PANELS; k < nb && !c.rc; ++k) {
    const double* Dk = Dinv + (int64_t)k * NB * NB;
    gemm<false, false>(c, NB, w, NB, 1.0, Dk, NB, Bw + (int64_t)k * NB * w, w, 0.0,
                       Y + (int64_t)k * NB * w, w);
    const int rem = npad - (k + 1) * NB;
    if (rem > 0)
      gemm<false, false>(c, rem, w, NB, -1.0, L + (int64_t)(k + 1) * NB * npad + (int64_t)k * NB, npad,
                         Y + (int64_t)k * NB * w, w, 1.0, Bw + (int64_t)(k + 1) * NB * w, w);
  }
}

// X = L^-T Zw ; Zw (npad x w) is consumed
static void solve_lower_t(Ctx& c, const double* L, int npad, const double* Dinv, double* Zw, double* X,
                          int w) {
  const int nb = npad / NB, nbig = npad / BIG;
  const double* Dbig = dbig_of(Dinv, npad);
  for (int k = nb - 1; k >= nbig * OUTER_PANELS && !c.rc; --k) {
    const double* Dk = Dinv + (int64_t)k * NB * NB;
    gemm<true, false>(c, NB, w, NB, 1.0, Dk, NB, Zw + (int64_t)k * NB * w, w, 0.0,
                      X + (int64_t)k * NB * w, w);
    if (k > 0)  // Zw[0:k] -= L[k, 0:k]' X_k
      gemm<true, false>(c, k * NB, w, NB, -1.0, L + (int64_t)k * NB * npad, npad,
                        X + (int64_t)k * NB * w, w, 1.0, Zw, w);
  }
  for (int ob = nbig - 1; ob >= 0 && !c.rc; --ob) {
    const int64_t r0 = (int64_t)ob * BIG;
    gemm<true, false>(c, BIG, w, BIG, 1.0, Dbig + (int64_t)ob * BIG * BIG, BIG, Zw + r0 * w, w, 0.0, X + r0 * w, w);
    if (r0 > 0)  // Zw[0:r0] -= L[r0:r0+256, 0:r0]' X_ob
      gemm<true, false>(c, (int)r0, w, BIG, -1.0, L + r0 * npad, npad, X + r0 * w, w, 1.0, Zw, w);
  }
}

struct SolveLayout {
  int npad, mpad, rpad;
  size_t off_Pt, off_Dinv, off_Ap, off_Y, off_Bw, off_S, off_DinvS, off_Bp, off_T1, off_T2, off_Lam,
      off_Z, off_Xt, off_X2, off_scal, total;
};

static SolveLayout solve_layout(int n, int m, int nrhs) {
  SolveLayout l;
  l.npad = (int)round_up(n, NB);
  l.mpad = (int)round_up(m, NB);
  l.rpad = (int)round_up(nrhs, NB);
  size_t o = 0;
  auto take = [&](size_t elems) {
    size_t r = o;
    o += (size_t)round_up((int64_t)(elems * sizeof(double)), 256);
    return r;
  };
  const size_t wmax = (size_t)(l.mpad > l.rpad ? l.mpad : l.rpad);
  l.off_Pt = take((size_t)l.npad * l.npad);
  l.off_Dinv = take(dinv_elems(l.npad));
  l.off_Ap = take((size_t)l.mpad * l.npad);
  l.off_Y = take((size_t)l.npad * l.mpad);
  l.off_Bw = take((size_t)(l.npad > l.mpad ? l.npad : l.mpad) * wmax);
  l.off_S = take((size_t)l.mpad * l.mpad);
  l.off_DinvS = take(dinv_elems(l.mpad));
  l.off_Bp = take((size_t)l.mpad * l.rpad);
  l.off_T1 = take((size_t)l.mpad * l.rpad);
  l.off_T2 = take((size_t)l.mpad * l.rpad);
  l.off_Lam = take((size_t)l.mpad * l.rpad);
  l.off_Z = take((size_t)l.npad * l.rpad);
  l.off_Xt = take((size_t)l.npad * l.rpad);
  l.off_X2 = take((size_t)l.npad * l.rpad);
  l.off_scal = take(32);
  l.total = o;
  return l;
}

}  // namespace aggf

using namespace aggf;

extern "C" size_t aggf_eq_qp_workspace_bytes(int32_t n, int32_t m, int32_t nrhs) {
  if (n <= 0 || m <= 0 || nrhs <= 0) return 0;
  return solve_layout(n, m, nrhs).total;
}

extern "C" int aggf_eq_qp_solve(const double* G, int32_t n, double l2, const double* l2_diag,
                                const double* A, int32_t m, const double* B, int32_t nrhs,
                                double schur_reg, int32_t n_refine, double* X, double* stats,
                                void* ws, size_t ws_bytes, void* stream_v) {
  if (!G || !A || !X || !stats || !ws) return fail(AGGF_ERR_ARG, "aggf_eq_qp_solve: NULL pointer");
  if (n <= 0 || m <= 0 || nrhs <= 0) return fail(AGGF_ERR_ARG, "aggf_eq_qp_solve: empty problem");
  if (!B && nrhs != m) return fail(AGGF_ERR_ARG, "aggf_eq_qp_solve: B == NULL needs nrhs == m");
  if (!(l2 >= 0.0)) return fail(AGGF_ERR_ARG, "aggf_eq_qp_solve: l2 must be >= 0");
  if (!(schur_reg >= 0.0) || n_refine < 0 || n_refine > 100)
    return fail(AGGF_ERR_ARG, "aggf_eq_qp_solve: bad schur_reg / n_refine");
  if (((uintptr_t)ws & 255) != 0) return fail(AGGF_ERR_ARG, "aggf_eq_qp_solve: workspace not 256-byte aligned");
  const SolveLayout l = solve_layout(n, m, nrhs);
  if (ws_bytes < l.total) return fail(AGGF_ERR_WORKSPACE, "aggf_eq_qp_solve: workspace too small (%zu < %zu)", ws_bytes, l.total);
  Ctx c;
  c.stream = (hipStream_t)stream_v;
  char* w = (char*)ws;
  auto P = [&](size_t off) { return reinterpret_cast<double*>(w + off); };
  double *Pt = P(l.off_Pt), *Dinv = P(l.off_Dinv), *Ap = P(l.off_Ap), *Y = P(l.off_Y), *Bw = P(l.off_Bw),
         *S = P(l.off_S), *DinvS = P(l.off_DinvS), *Bp = P(l.off_Bp), *T1 = P(l.off_T1), *T2 = P(l.off_T2),
         *Lam = P(l.off_Lam), *Z = P(l.off_Z), *Xt = P(l.off_Xt), *X2 = P(l.off_X2), *scal = P(l.off_scal);
  const int npad = l.npad, mpad = l.mpad, rpad = l.rpad;
  hipStream_t st = c.stream;

  hipLaunchKernelGGL(init_stats_kernel, dim3(1), dim3(64), 0, st, stats);
  hipLaunchKernelGGL(max_diag_kernel, dim3(1), dim3(256), 0, st, G, n, l2, l2_diag, scal);
  hipLaunchKernelGGL(copy_scalar_kernel, dim3(1), dim3(1), 0, st, scal, stats + 3);
  hipLaunchKernelGGL(build_pt_kernel, flat_grid((int64_t)npad * npad), dim3(256), 0, st, G, n, npad, l2, l2_diag, scal, Pt);
  hipLaunchKernelGGL(pad_copy_kernel, flat_grid((int64_t)mpad * npad), dim3(256), 0, st, A, m, n, 0, Ap, mpad, npad);
  hipLaunchKernelGGL(pad_copy_kernel, flat_grid((int64_t)mpad * rpad), dim3(256), 0, st, B, B ? m : (m < nrhs ? m : nrhs), nrhs, 0, Bp, mpad, rpad);
  AGGF_LAUNCH_OK();
  // P~ = P/s + A'A
  gemm<true, false>(c, npad, npad, mpad, 1.0, Ap, npad, Ap, npad, 1.0, Pt, npad);
  cholesky(c, Pt, npad, Dinv, stats, 0);
  // Y = L^-1 A'
  hipLaunchKernelGGL(pad_copy_kernel, flat_grid((int64_t)npad * mpad), dim3(256), 0, st, A, m, n, 1, Bw, npad, mpad);
  solve_lower(c, Pt, npad, Dinv, Bw, Y, mpad);
  // S = Y'Y (identity on the padding), factor it
  gemm<true, false>(c, mpad, mpad, npad, 1.0, Y, mpad, Y, mpad, 0.0, S, mpad);
  if (mpad > m) hipLaunchKernelGGL(fix_pad_diag_kernel, dim3(1), dim3(64), 0, st, S, m, mpad);
  if (schur_reg > 0.0) hipLaunchKernelGGL(schur_reg_kernel, dim3(1), dim3(256), 0, st, S, m, mpad, schur_reg);
  cholesky(c, S, mpad, DinvS, stats, n);
  // Lam = S^-1 Bp ; Xt = L^-T (Y Lam)
  auto schur_solve = [&](const double* rhs, double* out) {
    if (hipMemcpyAsync(T1, rhs, (size_t)mpad * rpad * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess)
      c.rc = fail(AGGF_ERR_HIP, "aggf_eq_qp_solve: device copy failed");
    solve_lower(c, S, mpad, DinvS, T1, T2, rpad);
    solve_lower_t(c, S, mpad, DinvS, T2, out, rpad);
  };
  schur_solve(Bp, Lam);
  gemm<false, false>(c, npad, rpad, mpad, 1.0, Y, mpad, Lam, rpad, 0.0, Z, rpad);
  solve_lower_t(c, Pt, npad, Dinv, Z, Xt, rpad);
  // refinement on the constraint residual R = A Xt - B:  Xt -= P~^-1 A' S^-1 R
  for (int it = 0; it < n_refine; ++it) {
    gemm<false, false>(c, mpad, rpad, npad, 1.0, Ap, npad, Xt, rpad, 0.0, Lam, rpad);
    hipLaunchKernelGGL(resid_kernel, dim3(1), dim3(256), 0, st, Lam, Bp, m, nrhs, rpad,
                       it == 0 ? stats + 2 : scal + 1);
    // padded rows/cols of A Xt - Bp are exact zeros, so the padded residual needs no masking
    schur_solve(Lam, Bw);  // Bw reused as (mpad x rpad) scratch for S^-1 R
    gemm<false, false>(c, npad, rpad, mpad, 1.0, Y, mpad, Bw, rpad, 0.0, Z, rpad);
    solve_lower_t(c, Pt, npad, Dinv, Z, X2, rpad);
    hipLaunchKernelGGL(axpy_kernel, flat_grid((int64_t)npad * rpad), dim3(256), 0, st, Xt, X2, -1.0, (int64_t)npad * rpad);
  }
  gemm<false, false>(c, mpad, rpad, npad, 1.0, Ap, npad, Xt, rpad, 0.0, Lam, rpad);
  hipLaunchKernelGGL(resid_kernel, dim3(1), dim3(256), 0, st, Lam, Bp, m, nrhs, rpad, stats + 1);
  hipLaunchKernelGGL(crop_transpose_kernel, flat_grid((int64_t)nrhs * n), dim3(256), 0, st, Xt, rpad, n, nrhs, X);
  AGGF_LAUNCH_OK();
  return c.rc;
}

// ---- Gram algebra for cross-validation (project_forces_grid_cv with Gram reuse) ---------------
namespace aggf {

// q[i] = sum_a X[i,a] * Y[i,a]; one workgroup per row, fixed summation order
__global__ __launch_bounds__(256) void rowdot_kernel(const double* __restrict__ X, int64_t ldx,
                                                     const double* __restrict__ Y, int64_t ldy, int n,
                                                     double* __restrict__ q) {
  __shared__ double part[256];
  const int i = blockIdx.x, tid = threadIdx.x;
  double s = 0.0;
  for (int a = tid; a < n; a += 256) s += X[(int64_t)i * ldx + a] * Y[(int64_t)i * ldy + a];
  part[tid] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) part[tid] += part[tid + w];
    __syncthreads();
  }
  if (tid == 0) q[i] = part[0];
}

__global__ __launch_bounds__(256) void axpby_kernel(int64_t n, double a, const double* __restrict__ x,
                                                    double b, const double* __restrict__ y,
                                                    double* __restrict__ out) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x)
    out[e] = a * x[e] + b * y[e];
}

}  // namespace aggf

extern "C" size_t aggf_gram_quadform_workspace_bytes(int32_t n, int32_t m) {
  if (n <= 0 || m <= 0) return 0;
  const size_t npad = (size_t)round_up(n, 64), mpad = (size_t)round_up(m, 64);
  return (npad * npad + 2 * mpad * npad) * sizeof(double) + 256;
}

extern "C" int aggf_gram_quadform(const double* G, int32_t n, const double* X, int32_t m, double* q,
                                  void* ws, size_t ws_bytes, void* stream_v) {
  hipStream_t st = (hipStream_t)stream_v;
  if (!G || !X || !q || !ws) return fail(AGGF_ERR_ARG, "aggf_gram_quadform: NULL pointer");
  if (n <= 0 || m <= 0) return fail(AGGF_ERR_ARG, "aggf_gram_quadform: empty problem");
  if (ws_bytes < aggf_gram_quadform_workspace_bytes(n, m))
    return fail(AGGF_ERR_WORKSPACE, "aggf_gram_quadform: workspace too small");
  const int npad = (int)round_up(n, 64), mpad = (int)round_up(m, 64);
  double* Gp = reinterpret_cast<double*>(ws);
  double* Xp = Gp + (size_t)npad * npad;
  double* Y = Xp + (size_t)mpad * npad;
  Ctx c{st};
  hipLaunchKernelGGL(pad_copy_kernel, flat_grid((int64_t)npad * npad), dim3(256), 0, st, G, n, n, 0, Gp, npad, npad);
  hipLaunchKernelGGL(pad_copy_kernel, flat_grid((int64_t)mpad * npad), dim3(256), 0, st, X, m, n, 0, Xp, mpad, npad);
  AGGF_LAUNCH_OK();
  gemm<false, false>(c, mpad, npad, npad, 1.0, Xp, npad, Gp, npad, 0.0, Y, npad);  // Y = X G
  if (c.rc) return c.rc;
  hipLaunchKernelGGL(rowdot_kernel, dim3(m), dim3(256), 0, st, Xp, (int64_t)npad, Y, (int64_t)npad, n, q);
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_daxpby(int64_t n, double a, const double* x, double b, const double* y, double* out,
                           void* stream_v) {
  hipStream_t st = (hipStream_t)stream_v;
  if (!x || !y || !out) return fail(AGGF_ERR_ARG, "aggf_daxpby: NULL pointer");
  if (n <= 0) return AGGF_OK;
  hipLaunchKernelGGL(axpby_kernel, flat_grid(n), dim3(256), 0, st, n, a, x, b, y, out);
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}
