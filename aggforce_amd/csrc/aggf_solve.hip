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

// C[M x N] = alpha * op(A) op(B) + beta * C.
// Row-major; M % 64 == 0, N % 64 == 0, K % 16 == 0; every operand 16-byte aligned with an even leading dimension (the
// solver pads every matrix to multiples of 64).  op(A) is M x K, op(B) is K x N.
// lower_only: skip tiles strictly above the block diagonal and, on diagonal 128-tiles, the 64 x 64 quadrant above it
// (square trailing updates).
// blockIdx.z = problem * per_prob + z: operand X is offset by problem * x_p + (z / inner) * x_o +
// (z % inner) * x_i elements (problem = one of the independent QPs of a batched solve; z = the
// batched GEMMs of one algorithmic step inside a problem).
struct GemmBatch {
  int inner;
  int64_t a_o, a_i, b_o, b_i, c_o, c_i;
  int per_prob = 1;
  int64_t a_p = 0, b_p = 0, c_p = 0;
};

// Two shapes of one kernel.  TM x TN = 128 x 128 with 8 waves (wave tile 64 x 32: 6 operand reads per 8 MFMAs, two
// workgroups per CU) for everything that is at least 128 wide -- the trailing updates, the Schur products, the
// triangular solves with many right-hand sides: the n^3 part; 64 x 64 with 4 waves (wave tile 32 x 32) for the 64-wide
// panel steps.  (Until round 4 there was only the 64 x 64 shape, fed by 8-BYTE global loads: 0.07-0.35 of the fp64 MFMA
// peak.)  Operand tiles travel global -> registers -> LDS with 16-byte loads, one stage (16 k) ahead of the MFMAs;
// the refill and the barrier sit in front of the stage's last MFMA group, whose operands are in registers by then.
// An operand that is contiguous along k is stored with ds_write_b128; one that is contiguous along the tile's
// rows/columns (op = transpose) is transposed by the LDS store.  LDS rows are 18 elements apart: the operand reads
// (lane = (row l & 15, k l >> 4)) touch 32 different bank pairs.
template <bool TA, bool TB, int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(64 * (TM / WM) * (TN / WN), 2) void gemm_tile_kernel(int M, int N, int K, double alpha,
                                                                                 const double* A, int64_t lda,
                                                                                 const double* B, int64_t ldb,
                                                                                 double beta, double* C, int64_t ldc,
                                                                                 int lower_only, GemmBatch bt) {
  AGGF_GATED_BODY_BEGIN
  using MF = Mfma<double>;
  typedef double __attribute__((ext_vector_type(2))) d2;
  constexpr int NWN = TN / WN, NW = (TM / WM) * NWN, THREADS = 64 * NW;
  constexpr int AM = WM / 16, AN = WN / 16;
  constexpr int CA = TM * 8 / THREADS, CB = TN * 8 / THREADS;  // 16-byte chunks per thread and stage
  static_assert(TM * 8 % THREADS == 0 && TN * 8 % THREADS == 0, "chunk split");
  const int bi = blockIdx.y, bj = blockIdx.x;
  const int64_t i0 = (int64_t)bi * TM, j0 = (int64_t)bj * TN;
  if (lower_only && j0 > i0 + (TM - 64)) return;  // (tiles are squares here: strictly above the block diagonal)
  if (gridDim.z > 1) {
    const int prob = blockIdx.z / bt.per_prob, z = blockIdx.z % bt.per_prob;
    const int zo = z / bt.inner, zi = z % bt.inner;
    A += prob * bt.a_p + zo * bt.a_o + zi * bt.a_i;
    B += prob * bt.b_p + zo * bt.b_o + zi * bt.b_i;
    C += prob * bt.c_p + zo * bt.c_o + zi * bt.c_i;
  }
  extern __shared__ __attribute__((aligned(16))) char gemm_smem[];
  double* sA = reinterpret_cast<double*>(gemm_smem);  // [2][TM * GS]
  double* sB = sA + 2 * TM * GS;                      // [2][TN * GS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  const int m_valid = (int)((M - i0) < TM ? (M - i0) : TM), n_valid = (int)((N - j0) < TN ? (N - j0) : TN);

  d2 ra[CA], rb[CB];
  auto load_stage = [&](int k0) {
#pragma unroll
    for (int q = 0; q < CA; ++q) {
      const int e = tid + THREADS * q;
      d2 v = {0.0, 0.0};
      if (TA) {  // op(A)[i][k] = A[k][i]: contiguous along i
        const int k = e / (TM / 2), ic = e % (TM / 2);
        if (2 * ic < m_valid) v = *reinterpret_cast<const d2*>(A + (int64_t)(k0 + k) * lda + i0 + 2 * ic);
      } else {   // A[i][k]: contiguous along k
        const int i = e >> 3, kc = e & 7;
        if (i < m_valid) v = *reinterpret_cast<const d2*>(A + (i0 + i) * lda + k0 + 2 * kc);
      }
      ra[q] = v;
    }
#pragma unroll
    for (int q = 0; q < CB; ++q) {
      const int e = tid + THREADS * q;
      d2 v = {0.0, 0.0};
      if (TB) {  // op(B)[k][j] = B[j][k]: contiguous along k
        const int j = e >> 3, kc = e & 7;
        if (j < n_valid) v = *reinterpret_cast<const d2*>(B + (j0 + j) * ldb + k0 + 2 * kc);
      } else {   // B[k][j]: contiguous along j
        const int k = e / (TN / 2), jc = e % (TN / 2);
        if (2 * jc < n_valid) v = *reinterpret_cast<const d2*>(B + (int64_t)(k0 + k) * ldb + j0 + 2 * jc);
      }
      rb[q] = v;
    }
  };
  auto store_stage = [&](int buf) {
    double* a = sA + buf * TM * GS;
    double* b = sB + buf * TN * GS;
#pragma unroll
    for (int q = 0; q < CA; ++q) {
      const int e = tid + THREADS * q;
      if (TA) {
        const int k = e / (TM / 2), ic = e % (TM / 2);
        a[(2 * ic) * GS + k] = ra[q][0];
        a[(2 * ic + 1) * GS + k] = ra[q][1];
      } else {
        const int i = e >> 3, kc = e & 7;
        *reinterpret_cast<d2*>(a + i * GS + 2 * kc) = ra[q];
      }
    }
#pragma unroll
    for (int q = 0; q < CB; ++q) {
      const int e = tid + THREADS * q;
      if (TB) {
        const int j = e >> 3, kc = e & 7;
        *reinterpret_cast<d2*>(b + j * GS + 2 * kc) = rb[q];
      } else {
        const int k = e / (TN / 2), jc = e % (TN / 2);
        b[(2 * jc) * GS + k] = rb[q][0];
        b[(2 * jc + 1) * GS + k] = rb[q][1];
      }
    }
  };

  f64x4 acc[AM][AN];
#pragma unroll
  for (int m = 0; m < AM; ++m)
#pragma unroll
    for (int n = 0; n < AN; ++n) acc[m][n] = acc_zero<double>();

  // a wave whose tile lies outside the matrix, or in the skipped quadrant of a diagonal tile, still stages and meets the
  // barriers but does not multiply
  const bool live = wm * WM < m_valid && wn * WN < n_valid &&
                    !(lower_only && j0 + wn * WN > i0 + wm * WM + (WM > 64 ? WM : 64) - 1);
  const int offA = (wm * WM + (lane & 15)) * GS + (lane >> 4);
  const int offB = (wn * WN + (lane & 15)) * GS + (lane >> 4);
  const int n_stage = K / GK;
  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int s = 0; s < n_stage; ++s) {
    const int cur = s & 1;
    if (s + 1 < n_stage) load_stage((s + 1) * GK);
    const double* a_s = sA + cur * TM * GS;
    const double* b_s = sB + cur * TN * GS;
#pragma unroll
    for (int kk = 0; kk < GK / 4; ++kk) {
      double a[AM], b[AN];
#pragma unroll
      for (int m = 0; m < AM; ++m) a[m] = a_s[offA + 16 * m * GS + 4 * kk];
#pragma unroll
      for (int n = 0; n < AN; ++n) b[n] = b_s[offB + 16 * n * GS + 4 * kk];
      if (kk == GK / 4 - 1) {
        if (s + 1 < n_stage) store_stage(cur ^ 1);
        __syncthreads();
      }
      if (live) {
#pragma unroll
        for (int m = 0; m < AM; ++m)
#pragma unroll
          for (int n = 0; n < AN; ++n) acc[m][n] = MF::mma(a[m], b[n], acc[m][n]);
      }
    }
  }
  if (!live) return;
#pragma unroll
  for (int m = 0; m < AM; ++m)
#pragma unroll
    for (int n = 0; n < AN; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = i0 + wm * WM + m * 16 + MF::row(lane, r);
        const int64_t col = j0 + wn * WN + n * 16 + (lane & 15);
        double v = alpha * acc[m][n][r];
        if (beta != 0.0) v += beta * C[row * ldc + col];
        C[row * ldc + col] = v;
      }
  AGGF_GATED_BODY_END
}

// Factor one 64x64 diagonal block: A_kk = L L' (lower), and Linv = L^-1.
// info[0]: 1-based global index of the first non-positive pivot (0 = ok).
// (Rounds 1-2 had a one-wave register version -- 35.5 us per block -- and a 256-thread version whose block products
// were scalar FMAs -- 40.3 us: profiles/r04_pruned_variants.patch.)  Here
//  * a 16 x 16 sub-block is factored AND inverted by all 256 threads, thread (i, c) owning entry (i, c) of the block and
//    of the running inverse Z (Z starts as the identity and receives the same row operations): per column one 16-entry
//    column buffer + one 16-entry Z-row buffer in LDS and ONE barrier; every thread recomputes the pivot's 1/sqrt itself;
//  * every block product runs on v_mfma_f64_16x16x4_f64, operands straight from the LDS copies of A / L / X, one
//    16 x 16 output tile per wave and pass.  The second product of an inverse level, X(bi,bi) W, takes W from the
//    accumulator registers of the first: the contraction index is walked in the order the D layout holds it
//    (k = (lane >> 4) + 4 r), so no LDS round trip is needed.
constexpr int PB = 16;
__device__ __forceinline__ double rsqrt_newton(double d) {
  // v_rsq_f64 is good to 5.2e-8; the third-order correction leaves 1.4e-16 (a further Newton step: 1.37e-16)
  const double rs = __builtin_amdgcn_rsq(d);
  const double e = fma(-d * rs, rs, 1.0);
  return fma(rs * e, fma(e, 0.375, 0.5), rs);
}

#ifdef AGGF_POTRF_PROF
// tools/potrf_probe.hip: shader cycles of thread 0 at the phase boundaries of one diagonal block
__device__ unsigned long long aggf_potrf_prof[8];
#define AGGF_PP(i) do { if (threadIdx.x == 0) aggf_potrf_prof[i] = __builtin_readcyclecounter(); } while (0)
// phase sums inside potrf64_factor_invert: [8] sub-blocks, [9] panels, [10] trailing updates, [11] inverse levels
__device__ unsigned long long aggf_potrf_phase[6];
#define AGGF_PH_BEGIN unsigned long long ph_t = __builtin_readcyclecounter(), ph_s[6] = {0, 0, 0, 0, 0, 0}
#define AGGF_PH(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); ph_s[i] += n_ - ph_t; ph_t = n_; } while (0)
#define AGGF_PH_LANDED asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define AGGF_PH_END do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 6; ++i_) aggf_potrf_phase[i_] = ph_s[i_]; } while (0)
#else
#define AGGF_PP(i)
#define AGGF_PH_BEGIN
#define AGGF_PH(i)
#define AGGF_PH_LANDED
#define AGGF_PH_END
#endif
// LDS of the diagonal-block kernel: A / L and L^-1 (64 x 65 each)
constexpr int POTRF3_LDS = 2 * NB * (NB + 1) * (int)sizeof(double);

__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// One wave factors AND inverts the 16 x 16 sub-block at (k0, k0) of the LDS block `a` (lower storage), four columns per
// step, everything in registers:
//   * the sub-block sits in the accumulator layout of v_mfma_f64_16x16x4 as a full SYMMETRIC matrix
//     (lane (c = lane & 15, q = lane >> 4) holds rows q, q+4, q+8, q+12 of column c), Z = the running inverse likewise;
//   * the 4 x 4 diagonal micro-block (10 entries) is read with v_readlane into wave-uniform values, factored
//     (four chained 1/sqrt: v_rsq_f64 + one third-order correction -- 1.4e-16 relative, a second step changes nothing,
//     tools/dp_latency.hip) and inverted (M = L_d^-1) by every lane alike;
//   * rows j0..j0+3 of the symmetric matrix ARE the transposed column block in the B-operand layout, so one MFMA with
//     M as the A operand gives the normalised columns L(:, j0..j0+3) = A(:, j0..) M' -- in the OPERAND layout
//     (lane (row, k)), which is what the rank-4 update C -= L L' needs for both operands: a second MFMA, no LDS, no
//     barrier, no transposition.  The same two MFMAs move Z: rows j0.. of Z are X = M Z(j0.., :), then Z -= L X.
// 1740 cycles per four columns with the exchange through the LDS (520 for the round trip + barrier, 1220 of arithmetic
// issue: every thread repeated the micro-block's factorisation and three forward substitutions) -> see DESIGN.md.
__device__ __forceinline__ void potrf16_wave(double (*a)[NB + 1], double (*x)[NB + 1], int k0,
                                             double* __restrict__ info, int pivot0) {
  using MF = Mfma<double>;
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  f64x4 ac, zc;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = lk + 4 * r;
    ac[r] = row >= li ? a[k0 + row][k0 + li] : a[k0 + li][k0 + row];
    zc[r] = row == li ? 1.0 : 0.0;
  }
  // The two MFMAs that move Z do not feed the next micro-block's pivots.  Issue order per step (pinned with scheduling
  // barriers; the matrix pipe takes one 64-cycle MFMA at a time): v_readlanes, two 1/sqrt chains, [Z -= L X of the
  // previous step], two more chains, M, [L = A M'], [A -= L L'], [X = M Z(j0.., :)] -- the Z products run under the
  // scalar chains instead of standing in front of them.
  double lv_prev = 0.0;
  f64x4 xr = acc_zero<double>();
  const f64x4 zero = acc_zero<double>();
#pragma unroll
  for (int jb = 0; jb < PB / 4; ++jb) {
    const int j0 = 4 * jb;
    // entry (j0+q, j0+p) of the symmetric matrix: register jb of lane (c = j0+p, q)
    const double blk = ac[jb];
    double d00 = readlane_f64(blk, j0 + 0), d10 = readlane_f64(blk, j0 + 16), d20 = readlane_f64(blk, j0 + 32),
           d30 = readlane_f64(blk, j0 + 48), d11 = readlane_f64(blk, j0 + 1 + 16), d21 = readlane_f64(blk, j0 + 1 + 32),
           d31 = readlane_f64(blk, j0 + 1 + 48), d22 = readlane_f64(blk, j0 + 2 + 32), d32 = readlane_f64(blk, j0 + 2 + 48),
           d33 = readlane_f64(blk, j0 + 3 + 48);
    const int piv = pivot0 + j0 + 1;
    if (!(d00 > 0.0)) { if (lane == 0 && info[0] == 0.0) info[0] = (double)(piv + 0); d00 = 1.0; }
    const double rs0 = rsqrt_newton(d00);
    const double l10 = d10 * rs0, l20 = d20 * rs0, l30 = d30 * rs0;
    d11 = fma(-l10, l10, d11); d21 = fma(-l20, l10, d21); d31 = fma(-l30, l10, d31);
    d22 = fma(-l20, l20, d22); d32 = fma(-l30, l20, d32); d33 = fma(-l30, l30, d33);
    if (!(d11 > 0.0)) { if (lane == 0 && info[0] == 0.0) info[0] = (double)(piv + 1); d11 = 1.0; }
    const double rs1 = rsqrt_newton(d11);
    const double l21 = d21 * rs1, l31 = d31 * rs1;
    d22 = fma(-l21, l21, d22); d32 = fma(-l31, l21, d32); d33 = fma(-l31, l31, d33);
    __builtin_amdgcn_sched_barrier(0);
    if (jb > 0) {  // Z of the previous step: xr holds rows j0-4 .. j0-1 of the inverse, lane (c, k): X[j0 - 4 + k][c]
      const double xv = xr[0];
      zc = MF::mma(-lv_prev, xv, zc);
      zc[jb - 1] = xv;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!(d22 > 0.0)) { if (lane == 0 && info[0] == 0.0) info[0] = (double)(piv + 2); d22 = 1.0; }
    const double rs2 = rsqrt_newton(d22);
    const double l32 = d32 * rs2;
    d33 = fma(-l32, l32, d33);
    if (!(d33 > 0.0)) { if (lane == 0 && info[0] == 0.0) info[0] = (double)(piv + 3); d33 = 1.0; }
    const double rs3 = rsqrt_newton(d33);
    // M = L_d^-1 (lower): the identity forward-substituted
    const double m10 = -(l10 * rs0) * rs1;
    const double m21 = -(l21 * rs1) * rs2;
    const double m32 = -(l32 * rs2) * rs3;
    const double m20 = -fma(l21, m10, l20 * rs0) * rs2;
    const double m31 = -fma(l32, m21, l31 * rs1) * rs3;
    const double m30 = -fma(l32, m20, fma(l31, m10, l30 * rs0)) * rs3;
    // A operand: lane (k = li < 4, p = lk) holds M[k][p]
    double am = 0.0;
    am = lane == 0 ? rs0 : am;
    am = lane == 1 ? m10 : am;
    am = lane == 2 ? m20 : am;
    am = lane == 3 ? m30 : am;
    am = lane == 17 ? rs1 : am;
    am = lane == 18 ? m21 : am;
    am = lane == 19 ? m31 : am;
    am = lane == 34 ? rs2 : am;
    am = lane == 35 ? m32 : am;
    am = lane == 51 ? rs3 : am;
    __builtin_amdgcn_sched_barrier(0);
    // L(c, j0 + k) in lane (c, k): zero above the diagonal and for the rows that are final already
    const f64x4 lt = MF::mma(am, blk, zero);
    const double lv = li >= j0 + lk ? lt[0] : 0.0;
    ac = MF::mma(-lv, lv, ac);
    __builtin_amdgcn_sched_barrier(0);
    xr = MF::mma(am, zc[jb], zero);
    a[k0 + li][k0 + j0 + lk] = lv;
    lv_prev = lv;
    __builtin_amdgcn_sched_barrier(0);
  }
  {
    const double xv = xr[0];
    zc = MF::mma(-lv_prev, xv, zc);
    zc[PB / 4 - 1] = xv;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = lk + 4 * r;
    x[k0 + row][k0 + li] = li <= row ? zc[r] : 0.0;
  }
}

// In place, by the 256 threads of a workgroup: a (lower triangle of a 64 x 64 block, zeros above) <- its Cholesky factor
// L, x <- L^-1.  Starts and ends with a barrier.  Four 16-column steps: the diagonal sub-block by one wave
// (potrf16_wave), the panel below and the trailing tiles by all four on MFMA; then the sub-blocks of the inverse
// below the diagonal, level by level.
__device__ __forceinline__ void potrf64_factor_invert(double (*a)[NB + 1], double (*x)[NB + 1],
                                                      double* __restrict__ info, int pivot_base) {
  using MF = Mfma<double>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  __syncthreads();
  AGGF_PH_BEGIN;
  for (int kb = 0; kb < NB / PB; ++kb) {
    const int k0 = kb * PB;
    // 1. diagonal sub-block: factored and inverted by wave 0 alone, in registers (see potrf16_wave)
    if (wave == 0) potrf16_wave(a, x, k0, info, pivot_base + k0);
    __syncthreads();
    AGGF_PH(0);
    const int nt16 = (NB - k0 - PB) / PB;  // 16-row tiles below the sub-block: 3, 2, 1, 0
    // 2. panel below: P = A_panel X_kk'  (wave w takes row tile w); in place -- a wave reads and writes its own rows only
    if (wave < nt16) {
      const int r0 = k0 + PB + wave * PB;
      f64x4 acc = acc_zero<double>();
#pragma unroll
      for (int kk = 0; kk < PB / 4; ++kk)
        acc = MF::mma(a[r0 + li][k0 + kk * 4 + lk], x[k0 + li][k0 + kk * 4 + lk], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) a[r0 + MF::row(lane, r)][k0 + li] = acc[r];
    }
    __syncthreads();
    AGGF_PH(1);
    // 3. trailing block -= P P' on the lower tiles (ti >= tj), dealt to the waves
    {
      int q = 0;
      for (int ti = 0; ti < nt16; ++ti)
        for (int tj = 0; tj <= ti; ++tj, ++q) {
          if ((q & 3) != wave) continue;
          const int r0 = k0 + PB + ti * PB, c0 = k0 + PB + tj * PB;
          f64x4 acc;
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] = a[r0 + MF::row(lane, r)][c0 + li];
#pragma unroll
          for (int kk = 0; kk < PB / 4; ++kk)
            acc = MF::mma(-a[r0 + li][k0 + kk * 4 + lk], a[c0 + li][k0 + kk * 4 + lk], acc);
#pragma unroll
          for (int r = 0; r < 4; ++r) a[r0 + MF::row(lane, r)][c0 + li] = acc[r];
        }
    }
    __syncthreads();
    AGGF_PH(2);
  }
  // sub-blocks of the inverse below the diagonal, by distance d from it:
  //   X(bi,bj) = -X(bi,bi) * sum_{k = bj}^{bi-1} L(bi,k) X(k,bj)        (one block per wave)
  for (int d = 1; d < NB / PB; ++d) {
    const int nblk = NB / PB - d;
    f64x4 out = acc_zero<double>();
    const int bj = wave, bi = wave + d;
    if (wave < nblk) {
      f64x4 w = acc_zero<double>();
      for (int kb = bj; kb < bi; ++kb) {
#pragma unroll
        for (int kk = 0; kk < PB / 4; ++kk)
          w = MF::mma(a[bi * PB + li][kb * PB + kk * 4 + lk], x[kb * PB + kk * 4 + lk][bj * PB + li], w);
      }
      // out = -X(bi,bi) W with the contraction index in D-layout order: k = lk + 4 r
#pragma unroll
      for (int r = 0; r < 4; ++r) out = MF::mma(-x[bi * PB + li][bi * PB + lk + 4 * r], w[r], out);
    }
    __syncthreads();  // (every X(k,bj) read above belongs to an earlier level or the diagonal: no hazard with the writes below)
    if (wave < nblk) {
#pragma unroll
      for (int r = 0; r < 4; ++r) x[bi * PB + MF::row(lane, r)][bj * PB + li] = out[r];
    }
    __syncthreads();
  }
  AGGF_PH(3);
  AGGF_PH_END;
}

__global__ __launch_bounds__(256) void potrf_diag_mfma_kernel(double* __restrict__ Akk, int64_t lda,
                                                             double* __restrict__ Linv,
                                                             double* __restrict__ info, int pivot_base,
                                                             int64_t a_ps, int64_t linv_ps, int64_t info_ps) {
  extern __shared__ __attribute__((aligned(16))) char potrf_smem[];
  double (*a)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(potrf_smem);        // A, then L (lower)
  double (*x)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(potrf_smem) + NB;   // L^-1
  const int tid = threadIdx.x;
  Akk += blockIdx.x * a_ps;  // blockIdx.x = problem of a batched solve
  Linv += blockIdx.x * linv_ps;
  info += blockIdx.x * info_ps;
  AGGF_PP(0);
  {
    // all 16 loads of a thread in flight together (the upper triangle is storage of the same matrix: read and dropped;
    // with the predicate inside the load the compiler waited for each one: 20 k cycles for this loop)
    double tmp[NB * NB / 256];
#pragma unroll
    for (int q = 0; q < NB * NB / 256; ++q) {
      const int e = tid + 256 * q;
      tmp[q] = Akk[(int64_t)(e / NB) * lda + (e % NB)];
    }
#pragma unroll
    for (int q = 0; q < NB * NB / 256; ++q) {
      const int e = tid + 256 * q, r = e / NB, c = e % NB;
      a[r][c] = (c <= r) ? tmp[q] : 0.0;
      x[r][c] = 0.0;
    }
  }
  AGGF_PP(1);
  potrf64_factor_invert(a, x, info, pivot_base);
  AGGF_PP(2);
  for (int e = tid; e < NB * NB; e += 256) {
    const int r = e / NB, c = e - r * NB;
    if (c <= r) Akk[(int64_t)r * lda + c] = a[r][c];
  }
  AGGF_PP(3);
  for (int e = tid; e < NB * NB; e += 256) {
    const int r = e / NB, c = e - r * NB;
    Linv[r * NB + c] = x[r][c];
  }
  AGGF_PP(4);
  AGGF_PP(5);
}

// One 64-column step of the factorisation INSIDE a 256-wide outer panel, in one launch (left-looking): block column k
// of P = rows of the diagonal block and the `nrb` 64-row blocks below it (the extra rows of the right-hand sides ride
// along) first receives the products with the j earlier block columns of its outer panel (K = 64 j <= 192), then the
// diagonal block is factored and inverted, then the rows below are multiplied by the inverse.  Every workgroup forms and
// factors the diagonal block ITSELF (the same 32 KB from the L2 for all of them, ~13 us of redundant work on otherwise
// idle CUs) and then owns row blocks blockIdx.x, blockIdx.x + gridDim.x, ...: the chain diag -> panel -> inner update of
// the right-looking form (three dependent launches, 28 + 13 + 13 us at n = 4096) becomes one launch.  Workgroup 0 stores
// the inverse of the diagonal block (the block itself stays as it was: see below).  blockIdx.y = problem of a batched solve.
constexpr int STEP_LDS = 4 * NB * (NB + 1) * (int)sizeof(double);
__global__ __launch_bounds__(256) void chol_step_kernel(double* __restrict__ P, int64_t ld, int k, int j, int nrb,
                                                       double* __restrict__ Linv, double* __restrict__ info,
                                                       int pivot_base, int64_t p_ps, int64_t linv_ps, int64_t info_ps) {
  AGGF_GATED_BODY_BEGIN
  using MF = Mfma<double>;
  extern __shared__ __attribute__((aligned(16))) char step_smem[];
  double (*a)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(step_smem);            // diagonal block, then its factor
  double (*x)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(step_smem) + NB;       // its inverse
  double (*sk)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(step_smem) + 2 * NB;  // L(k, p): rows of the diagonal block
  double (*sr)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(step_smem) + 3 * NB;  // L(r, p): rows of the row block; then U
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  P += blockIdx.y * p_ps;
  Linv += blockIdx.y * linv_ps;
  info += blockIdx.y * info_ps;
  const int64_t kr = (int64_t)k * NB;          // first row / column of the diagonal block
  const int64_t pc = (int64_t)(k - j) * NB;    // first column of the outer panel
  int rb = blockIdx.x;                         // row block of this workgroup (-1 + ... below): rows kr + 64 (rb + 1) ..
  const bool has_rows = rb < nrb;

  // a 64 x 64 block of P (rows r0.., columns c0..) into an LDS tile, 16 loads per thread in flight
  auto stage = [&](double (*dst)[NB + 1], int64_t r0, int64_t c0) {
    double tmp[NB * NB / 256];
#pragma unroll
    for (int q = 0; q < NB * NB / 256; ++q) {
      const int e = tid + 256 * q;
      tmp[q] = P[(r0 + e / NB) * ld + c0 + (e % NB)];
    }
#pragma unroll
    for (int q = 0; q < NB * NB / 256; ++q) {
      const int e = tid + 256 * q;
      dst[e / NB][e % NB] = tmp[q];
    }
  };
  // wave w owns the 16-row tile w of a 64 x 64 result: four 16 x 16 accumulators (column tiles 0..3)
  auto load_acc = [&](f64x4 (&acc)[4], int64_t r0, int64_t c0) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = P[(r0 + wave * 16 + MF::row(lane, r)) * ld + c0 + t * 16 + li];
  };
  // acc -= rows(lhs tile row `wave`) * rows(rhs)'   over the 64 columns of the staged tiles
  auto minus_abt = [&](f64x4 (&acc)[4], double (*lhs)[NB + 1], double (*rhs)[NB + 1]) {
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) {
      const double av = -lhs[wave * 16 + li][kk * 4 + lk];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = MF::mma(av, rhs[t * 16 + li][kk * 4 + lk], acc[t]);
    }
  };

  f64x4 accD[4], accU[4];
  load_acc(accD, kr, kr);
  if (has_rows) load_acc(accU, kr + (int64_t)(rb + 1) * NB, kr);
  for (int p = 0; p < j; ++p) {
    stage(sk, kr, pc + (int64_t)p * NB);
    if (has_rows) stage(sr, kr + (int64_t)(rb + 1) * NB, pc + (int64_t)p * NB);
    __syncthreads();
    minus_abt(accD, sk, sk);
    if (has_rows) minus_abt(accU, sr, sk);
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wave * 16 + MF::row(lane, r), col = t * 16 + li;
      a[row][col] = col <= row ? accD[t][r] : 0.0;
      x[row][col] = 0.0;
    }
  potrf64_factor_invert(a, x, info, pivot_base);
  // Workgroup 0 stores the inverse.  The FACTOR of the diagonal block is not stored: every workgroup of this launch
  // reads the block from P when it starts, and one that is dispatched late (a busy GPU, a second process on it) would
  // find it half overwritten -- seen once as two ranks of a rehearsal on one GPU disagreeing on the solved map.  Nothing
  // reads the diagonal blocks of the factor afterwards: the solves use their inverses, the updates the blocks below.
  if (blockIdx.x == 0) {
    for (int e = tid; e < NB * NB; e += 256) Linv[e] = x[e / NB][e % NB];
  }
  // rows below: L(r, k) = U X'   (X lower triangular: column tile t needs the contraction index up to 16 (t + 1) only)
  for (bool first = true; rb < nrb; rb += gridDim.x, first = false) {
    const int64_t r0 = kr + (int64_t)(rb + 1) * NB;
    if (!first) {
      load_acc(accU, r0, kr);
      for (int p = 0; p < j; ++p) {
        __syncthreads();
        stage(sk, kr, pc + (int64_t)p * NB);
        stage(sr, r0, pc + (int64_t)p * NB);
        __syncthreads();
        minus_abt(accU, sr, sk);
      }
    }
    __syncthreads();  // sr free (and, first pass, the factor's last barrier passed)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) sr[wave * 16 + MF::row(lane, r)][t * 16 + li] = accU[t][r];
    __syncthreads();
    f64x4 out[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) out[t] = acc_zero<double>();
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) {
      const double uv = sr[wave * 16 + li][kk * 4 + lk];
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (kk < 4 * (t + 1)) out[t] = MF::mma(uv, x[t * 16 + li][kk * 4 + lk], out[t]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) P[(r0 + wave * 16 + MF::row(lane, r)) * ld + kr + t * 16 + li] = out[t][r];
  }
  AGGF_GATED_BODY_END
}

// Helper kernels of the solve.  blockIdx.y = problem of a batched solve; every array argument comes
// with its per-problem stride (`*_ps`, elements).

// scale[0] = max_i (G[i,i] + l2*diag[i]), or 1 if that is not positive/finite
__global__ __launch_bounds__(256) void max_diag_kernel(const double* __restrict__ G, int n, int64_t g_ps,
                                                       double l2, const double* __restrict__ l2d,
                                                       double* __restrict__ scale, int64_t scale_ps) {
  __shared__ double sh[256];
  G += blockIdx.y * g_ps;
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
  if (threadIdx.x == 0) scale[blockIdx.y * scale_ps] = (sh[0] > 0.0 && sh[0] < 1e300) ? sh[0] : 1.0;
}

// Pt (npad x npad) = (G + l2*diag)/s [+ shift] on the n x n block, identity on the padding; `shift` (n x n per
// problem, may be NULL) is a caller-formed A'A of which only the lower triangle is read; `perm` (n per problem, may be
// NULL) reorders the variables
__global__ __launch_bounds__(256) void build_pt_kernel(const double* __restrict__ G, int n, int64_t g_ps, int npad,
                                                       double l2, const double* __restrict__ l2d,
                                                       const double* __restrict__ scale, int64_t scale_ps,
                                                       const double* __restrict__ shift,
                                                       const int32_t* __restrict__ perm,
                                                       double* __restrict__ Pt, int64_t pt_ps) {
  G += blockIdx.y * g_ps;
  if (shift) shift += blockIdx.y * g_ps;
  if (perm) perm += (int64_t)blockIdx.y * n;
  Pt += blockIdx.y * pt_ps;
  const double inv_s = 1.0 / scale[blockIdx.y * scale_ps];
  const int64_t total = (int64_t)npad * npad;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / npad), j = (int)(e - (int64_t)i * npad);
    double v;
    if (i < n && j < n) {
      // variable i of the factorisation = variable perm[i] of the caller (symmetric permutation of G and of the shift,
      // whose lower triangle is all the caller wrote)
      const int gi = perm ? perm[i] : i, gj = perm ? perm[j] : j;
      v = G[(int64_t)gi * n + gj];
      if (i == j) v += l2 * (l2d ? l2d[gi] : 1.0);
      v *= inv_s;
      if (shift && j <= i) v += gi >= gj ? shift[(int64_t)gi * n + gj] : shift[(int64_t)gj * n + gi];
    } else {
      v = (i == j) ? 1.0 : 0.0;
    }
    Pt[e] = v;
  }
}

// dst (rd x cd, zero padded) = src (rs x cs) or its transpose; identity if src == NULL
__global__ __launch_bounds__(256) void pad_copy_kernel(const double* __restrict__ src, int rs, int cs, int64_t src_ps,
                                                       int transpose, double* __restrict__ dst,
                                                       int rd, int cd, int64_t dst_ps,
                                                       const int32_t* __restrict__ colperm = nullptr) {
  if (src) src += blockIdx.y * src_ps;
  dst += blockIdx.y * dst_ps;
  if (colperm) colperm += (int64_t)blockIdx.y * cs;  // column j of the copy = column colperm[j] of src
  const int64_t total = (int64_t)rd * cd;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / cd), j = (int)(e - (int64_t)i * cd);
    double v = 0.0;
    if (!src) {
      v = (i == j && i < rs) ? 1.0 : 0.0;
    } else if (!transpose) {
      if (i < rs && j < cs) v = src[(int64_t)i * cs + (colperm ? colperm[j] : j)];
    } else {
      if (j < rs && i < cs) v = src[(int64_t)j * cs + (colperm ? colperm[i] : i)];
    }
    dst[e] = v;
  }
}

// S[i,i] = 1 for padding rows i >= m
__global__ void fix_pad_diag_kernel(double* __restrict__ S, int m, int mpad, int64_t s_ps) {
  S += blockIdx.y * s_ps;
  const int i = m + blockIdx.x * blockDim.x + threadIdx.x;
  if (i < mpad) S[(int64_t)i * mpad + i] = 1.0;
}

// R -= Bp elementwise on (rows x cols), then out[0] = max(out[0], max |R| over the valid block); out[0] must hold 0
// (or an earlier maximum) on entry: the workgroups of a problem combine through an atomic max on the bit pattern of
// the non-negative double (order-preserving; NaN is reported as +inf).  One workgroup took 104 us per call.
__global__ __launch_bounds__(256) void resid_kernel(double* __restrict__ R, const double* __restrict__ Bp,
                                                    int rows, int cols, int ld, int64_t mat_ps,
                                                    double* __restrict__ out, int64_t out_ps) {
  __shared__ double sh[256];
  R += blockIdx.y * mat_ps;
  Bp += blockIdx.y * mat_ps;
  double m = 0.0;
  const int64_t total = (int64_t)rows * cols;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
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
  if (threadIdx.x == 0)
    atomicMax(reinterpret_cast<unsigned long long*>(out + blockIdx.y * out_ps), (unsigned long long)__double_as_longlong(sh[0]));
}

__global__ void zero_scalar_kernel(double* dst, int64_t dst_ps) { dst[blockIdx.x * dst_ps] = 0.0; }

__global__ __launch_bounds__(256) void axpy_kernel(double* __restrict__ x, const double* __restrict__ y,
                                                   double a, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x)
    x[e] += a * y[e];
}

// X (nrhs x n) = Xt (npad x rpad) transposed and cropped
__global__ __launch_bounds__(256) void crop_transpose_kernel(const double* __restrict__ Xt, int rpad, int64_t xt_ps,
                                                             int n, int nrhs, double* __restrict__ X,
                                                             const int32_t* __restrict__ perm = nullptr) {
  Xt += blockIdx.y * xt_ps;
  X += blockIdx.y * (int64_t)nrhs * n;
  if (perm) perm += (int64_t)blockIdx.y * n;
  const int64_t total = (int64_t)nrhs * n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / n), i = (int)(e - (int64_t)r * n);
    X[(int64_t)r * n + (perm ? perm[i] : i)] = Xt[(int64_t)i * rpad + r];
  }
}

// S[i,i] += reg * trace(S[:m,:m])  for i < m  (Tikhonov shift for redundant constraint rows).  Relative to the TRACE,
// an upper bound of the largest eigenvalue -- not to the mean diagonal entry: with 20 n_cg sampled rows of rank ~30
// the trace sits in a few eigenvalues, reg * trace / m fell to 1e-14 of the largest one, the level of the blocked
// factorisation's own rounding, and an exactly redundant row produced a non-positive pivot (round 3, 100 rows).
__global__ __launch_bounds__(256) void schur_reg_kernel(double* __restrict__ S, int m, int mpad, int64_t s_ps,
                                                        double reg) {
  __shared__ double sh[256];
  S += blockIdx.y * s_ps;
  double t = 0.0;
  for (int i = threadIdx.x; i < m; i += 256) t += S[(int64_t)i * mpad + i];
  sh[threadIdx.x] = t;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  const double shift = reg * sh[0];
  for (int i = threadIdx.x; i < m; i += 256) S[(int64_t)i * mpad + i] += shift;
}

__global__ void init_stats_kernel(double* stats, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) stats[i] = 0.0;
}
__global__ void copy_scalar_kernel(const double* src, int64_t src_ps, double* dst, int64_t dst_ps) {
  dst[blockIdx.x * dst_ps] = src[blockIdx.x * src_ps];
}

// ---------------------------------------------------------------------------
struct Ctx {
  hipStream_t stream;
  int rc = AGGF_OK;
  int nprob = 1;  // independent problems solved side by side (blockIdx.z of the GEMMs, .x/.y of the helpers)
};

// a matrix of every problem of the batch: problem p starts at p + p * ps
struct Mat {
  double* p;
  int64_t ld;
  int64_t ps;
  Mat at(int64_t row, int64_t col) const { return Mat{p + row * ld + col, ld, ps}; }
};

static inline dim3 flat_grid(int64_t n, int nprob = 1) {
  int64_t g = ceil_div(n, 256);
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return dim3((unsigned)g, (unsigned)nprob);
}

template <bool TA, bool TB>
static void gemm(Ctx& c, int M, int N, int K, double alpha, Mat A, Mat B, double beta, Mat C, int lower_only = 0,
                 int nbatch = 1, GemmBatch bt = GemmBatch{1, 0, 0, 0, 0, 0, 0}) {
  if (c.rc || M <= 0 || N <= 0 || nbatch <= 0) return;
  // In place (C == A): sound only while a workgroup reads exactly the rows it writes -- A not transposed and ONE column
  // tile (N <= 64 keeps the 64 x 64 shape: 128-wide tiles are chosen for N >= 128 only), the panel products of the
  // three-launch factorisation.  C == B never is.
  if ((C.p == A.p && (TA || N > NB)) || C.p == B.p) {
    c.rc = fail(AGGF_ERR_ARG, "gemm: in-place product with more than one column tile (workgroups would read what others write)");
    return;
  }
  bt.per_prob = nbatch;
  bt.a_p = A.ps;
  bt.b_p = B.ps;
  bt.c_p = C.ps;
  const int64_t gz = (int64_t)nbatch * c.nprob;
  if (gz > 65535) {
    c.rc = fail(AGGF_ERR_ARG, "batched solve: too many problems for one launch");
    return;
  }
  // the 128 x 128 shape when its grid still fills the chip twice over (2 workgroups per CU): the batched solves of the
  // featurised fit; a single problem (n = 4096: 528 lower tiles at most) keeps the 64 x 64 shape's four times as many
  // workgroups (same box: c3 solve 5.2 ms with 128-tiles throughout against 4.0)
  const int64_t wgs128 = (int64_t)ceil_div(M, 128) * ceil_div(N, 128) * gz / (lower_only ? 2 : 1);
  if (M >= 128 && N >= 128 && wgs128 >= 2 * 2 * (int64_t)device_cu_count()) {
    constexpr size_t lds = (size_t)2 * (128 + 128) * GS * sizeof(double);  // 73.7 KB: two workgroups per CU
    static thread_local PerDeviceOnce once;
    bool& done = *once.flag();
    if (!done) {
      if (hipFuncSetAttribute((const void*)gemm_tile_kernel<TA, TB, 128, 128, 64, 32>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        c.rc = fail(AGGF_ERR_HIP, "gemm LDS attribute failed");
      done = true;
    }
    AGGF_LAUNCH_GATED(384, (gemm_tile_kernel<TA, TB, 128, 128, 64, 32>),
                       dim3((unsigned)ceil_div(N, 128), (unsigned)ceil_div(M, 128), (unsigned)gz), dim3(512), lds, c.stream, M,
                       N, K, alpha, A.p, A.ld, B.p, B.ld, beta, C.p, C.ld, lower_only, bt);
  } else {
    constexpr size_t lds = (size_t)2 * (64 + 64) * GS * sizeof(double);
    AGGF_LAUNCH_GATED(512, (gemm_tile_kernel<TA, TB, 64, 64, 32, 32>), dim3(N / 64, M / 64, (unsigned)gz), dim3(256), lds,
                       c.stream, M, N, K, alpha, A.p, A.ld, B.p, B.ld, beta, C.p, C.ld, lower_only, bt);
  }
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
static inline Mat dbig_of(Mat Dinv, int npad) { return Mat{Dinv.p + (size_t)npad * NB, BIG, Dinv.ps}; }

// Dbig[ob] (256x256) <- block diagonal of the four 64x64 inverses of outer block ob, zeros elsewhere
__global__ __launch_bounds__(256) void dbig_init_kernel(const double* __restrict__ Dinv, double* __restrict__ Dbig,
                                                        int nbig, int64_t d_ps) {
  Dinv += blockIdx.y * d_ps;
  Dbig += blockIdx.y * d_ps;
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
static void build_big_inverses(Ctx& c, Mat L, int npad, Mat Dinv) {
  const int nbig = npad / BIG;
  if (nbig <= 0 || c.rc) return;
  const Mat Dbig = dbig_of(Dinv, npad);
  double* Tp = Dbig.p + (size_t)nbig * BIG * BIG;
  AGGF_LAUNCH(dbig_init_kernel, flat_grid((int64_t)nbig * BIG * BIG, c.nprob), dim3(256), 0, c.stream, Dinv.p,
                     Dbig.p, nbig, Dinv.ps);
  if (hipGetLastError() != hipSuccess) { c.rc = fail(AGGF_ERR_HIP, "dbig_init launch failed"); return; }
  const int64_t l_o = (int64_t)BIG * npad + BIG, d_o = (int64_t)BIG * BIG, t_o = (int64_t)(BIG / 2) * (BIG / 2);
  // level 1: pairs of 64-blocks (p = 0, 1) inside every outer block
  {
    const int64_t l_i = (int64_t)2 * NB * npad + 2 * NB, d_i = (int64_t)2 * NB * BIG + 2 * NB;
    const Mat T{Tp, NB, Dinv.ps};
    // T = B Ainv
    gemm<false, false>(c, NB, NB, NB, 1.0, L.at(NB, 0), Dbig, 0.0, T, 0, 2 * nbig,
                       GemmBatch{2, l_o, l_i, d_o, d_i, t_o, (int64_t)NB * NB});
    // X = -Cinv T
    gemm<false, false>(c, NB, NB, NB, -1.0, Dbig.at(NB, NB), T, 0.0, Dbig.at(NB, 0), 0, 2 * nbig,
                       GemmBatch{2, d_o, d_i, t_o, (int64_t)NB * NB, d_o, d_i});
  }
  // level 2: the two 128-blocks of every outer block
  {
    const int H = BIG / 2;
    const Mat T{Tp, H, Dinv.ps};
    gemm<false, false>(c, H, H, H, 1.0, L.at(H, 0), Dbig, 0.0, T, 0, nbig, GemmBatch{1, l_o, 0, d_o, 0, t_o, 0});
    gemm<false, false>(c, H, H, H, -1.0, Dbig.at(H, H), T, 0.0, Dbig.at(H, 0), 0, nbig,
                       GemmBatch{1, d_o, 0, t_o, 0, d_o, 0});
  }
}

// info: first of the 4 stats doubles of problem 0 (stride 4 between problems)
// extra_rows (a multiple of 64): rows npad .. npad+extra_rows-1 of P (same ld) ride along -- they receive every panel
// and trailing update, i.e. leave as B' L^-T for the B' they held: the forward solve Y = L^-1 B of those right-hand
// sides comes out of the factorisation and needs no launches of its own.
static void cholesky(Ctx& c, Mat P, int npad, Mat Dinv, double* info, int pivot_base, int extra_rows = 0) {
  const int nb = npad / NB;
  static thread_local PerDeviceOnce attr_once3;
  bool& attr_done3 = *attr_once3.flag();
  if (!attr_done3) {
    if (hipFuncSetAttribute((const void*)potrf_diag_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            POTRF3_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)chol_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, STEP_LDS) !=
            hipSuccess)
      c.rc = fail(AGGF_ERR_HIP, "potrf LDS attribute failed");
    attr_done3 = true;
  }
  // workgroups per problem that find a CU of their own (the step kernel takes a CU's LDS).  AGGF_SOLVE_WGS (tests: read
  // per call) caps it, which sends small systems through the row walk and the three-launch form.
  int gx_max = device_cu_count() / c.nprob > 0 ? device_cu_count() / c.nprob : 1;
  if (const char* e = getenv("AGGF_SOLVE_WGS")) {
    const int cap = atoi(e);
    if (cap > 0 && cap < gx_max) gx_max = cap;
  }
  for (int k0 = 0; k0 < nb && !c.rc; k0 += OUTER_PANELS) {
    const int kend = k0 + OUTER_PANELS < nb ? k0 + OUTER_PANELS : nb;
    // One launch per 64-column step (chol_step_kernel, left-looking inside the outer panel) while a workgroup has at
    // most 8 row blocks to walk; the large batched solves keep the three launches per step (diagonal block, panel,
    // right-looking inner update), whose GEMMs spread the rows over the whole chip.  One form per outer panel: the two
    // differ in WHEN a block column receives the inner updates.
    const int nrb0 = (npad - (k0 + 1) * NB + extra_rows) / NB;
    const bool one_launch = ceil_div(nrb0 > 0 ? nrb0 : 1, gx_max) <= 8;
    for (int k = k0; k < kend && !c.rc; ++k) {
      const Mat Akk = P.at((int64_t)k * NB, (int64_t)k * NB);
      const Mat Dk{Dinv.p + (int64_t)k * NB * NB, NB, Dinv.ps};
      const int rem = npad - (k + 1) * NB + extra_rows;
      if (one_launch) {
        const int nrb = rem / NB;
        const int gx = nrb < 1 ? 1 : (nrb < gx_max ? nrb : gx_max);
        AGGF_LAUNCH_GATED(256, chol_step_kernel, dim3((unsigned)gx, (unsigned)c.nprob), dim3(256), STEP_LDS, c.stream, P.p,
                           P.ld, k, k - k0, nrb, Dk.p, info, pivot_base + k * NB, P.ps, Dinv.ps, (int64_t)4);
        if (hipGetLastError() != hipSuccess) c.rc = fail(AGGF_ERR_HIP, "chol_step launch failed");
        continue;
      }
      AGGF_LAUNCH(potrf_diag_mfma_kernel, dim3(c.nprob), dim3(256), POTRF3_LDS, c.stream, Akk.p, P.ld,
                         Dk.p, info, pivot_base + k * NB, P.ps, Dinv.ps, (int64_t)4);
      if (hipGetLastError() != hipSuccess) c.rc = fail(AGGF_ERR_HIP, "potrf launch failed");
      if (rem <= 0) break;
      const Mat panel = Akk.at(NB, 0);  // rows below the diagonal block, same columns
      // panel <- panel * Linv'  (in place: every workgroup reads exactly the rows it writes)
      gemm<false, true>(c, rem, NB, NB, 1.0, panel, Dk, 0.0, panel);
      // remaining columns of this outer panel <- themselves - panel panel'   (lower tiles only)
      const int inner_cols = (kend - 1 - k) * NB;
      if (inner_cols > 0) gemm<false, true>(c, rem, inner_cols, NB, -1.0, panel, panel, 1.0, panel.at(0, NB), 1);
    }
    const int rem2 = npad - kend * NB;
    if (rem2 > 0) {
      // trailing <- trailing - L21 L21' with all columns of the outer panel at once (extra rows: all their tiles
      // lie below the block diagonal)
      const int kw = (kend - k0) * NB;
      const Mat L21 = P.at((int64_t)kend * NB, (int64_t)k0 * NB);
      gemm<false, true>(c, rem2 + extra_rows, rem2, kw, -1.0, L21, L21, 1.0, P.at((int64_t)kend * NB, (int64_t)kend * NB), 1);
    }
  }
  build_big_inverses(c, P, npad, Dinv);
}

// Y = L^-1 Bw ; Bw (npad x w, ld = w) is consumed.  Full 256-row blocks use their inverted diagonal
// block, the (< 256 rows) remainder the 64x64 inverses.
// first_big > 0: rows of Bw above block first_big are zero, and so are Y's -- they are neither read nor written.
static void solve_lower(Ctx& c, Mat L, int npad, Mat Dinv, Mat Bw, Mat Y, int w, int first_big = 0) {
  const int nb = npad / NB, nbig = npad / BIG;
  const Mat Dbig = dbig_of(Dinv, npad);
  for (int ob = first_big; ob < nbig && !c.rc; ++ob) {
    const int64_t r0 = (int64_t)ob * BIG;
    gemm<false, false>(c, BIG, w, BIG, 1.0, Dbig.at(r0, 0), Bw.at(r0, 0), 0.0, Y.at(r0, 0));
    const int rem = npad - (int)(r0 + BIG);
    if (rem > 0) gemm<false, false>(c, rem, w, BIG, -1.0, L.at(r0 + BIG, r0), Y.at(r0, 0), 1.0, Bw.at(r0 + BIG, 0));
  }
  for (int k = nbig * OUTER_PANELS; k < nb && !c.rc; ++k) {
    const Mat Dk{Dinv.p + (int64_t)k * NB * NB, NB, Dinv.ps};
    const int64_t r0 = (int64_t)k * NB;
    gemm<false, false>(c, NB, w, NB, 1.0, Dk, Bw.at(r0, 0), 0.0, Y.at(r0, 0));
    const int rem = npad - (k + 1) * NB;
    if (rem > 0) gemm<false, false>(c, rem, w, NB, -1.0, L.at(r0 + NB, r0), Y.at(r0, 0), 1.0, Bw.at(r0 + NB, 0));
  }
}

// X = L^-T Zw ; Zw (npad x w) is consumed
static void solve_lower_t(Ctx& c, Mat L, int npad, Mat Dinv, Mat Zw, Mat X, int w) {
  const int nb = npad / NB, nbig = npad / BIG;
  const Mat Dbig = dbig_of(Dinv, npad);
  for (int k = nb - 1; k >= nbig * OUTER_PANELS && !c.rc; --k) {
    const Mat Dk{Dinv.p + (int64_t)k * NB * NB, NB, Dinv.ps};
    const int64_t r0 = (int64_t)k * NB;
    gemm<true, false>(c, NB, w, NB, 1.0, Dk, Zw.at(r0, 0), 0.0, X.at(r0, 0));
    if (k > 0)  // Zw[0:k] -= L[k, 0:k]' X_k
      gemm<true, false>(c, k * NB, w, NB, -1.0, L.at(r0, 0), X.at(r0, 0), 1.0, Zw);
  }
  for (int ob = nbig - 1; ob >= 0 && !c.rc; --ob) {
    const int64_t r0 = (int64_t)ob * BIG;
    gemm<true, false>(c, BIG, w, BIG, 1.0, Dbig.at(r0, 0), Zw.at(r0, 0), 0.0, X.at(r0, 0));
    if (r0 > 0)  // Zw[0:r0] -= L[r0:r0+256, 0:r0]' X_ob
      gemm<true, false>(c, (int)r0, w, BIG, -1.0, L.at(r0, 0), X.at(r0, 0), 1.0, Zw);
  }
}

// Workspace: every array holds all problems of the batch back to back ([problem][elements]).
struct SolveLayout {
  int npad, mpad, rpad;
  size_t e_Pt, e_Dinv, e_Ap, e_Y, e_Bw, e_S, e_DinvS, e_mr, e_nr;  // elements per problem
  size_t off_Pt, off_Dinv, off_Ap, off_Y, off_Bw, off_S, off_DinvS, off_Bp, off_T1, off_T2, off_Lam,
      off_Z, off_Xt, off_X2, off_scal, total;
};

static SolveLayout solve_layout(int n, int m, int nrhs, int nprob) {
  SolveLayout l;
  l.npad = (int)round_up(n, NB);
  l.mpad = (int)round_up(m, NB);
  l.rpad = (int)round_up(nrhs, NB);
  size_t o = 0;
  auto take = [&](size_t elems) {
    size_t r = o;
    o += (size_t)round_up((int64_t)(elems * nprob * sizeof(double)), 256);
    return r;
  };
  const size_t wmax = (size_t)(l.mpad > l.rpad ? l.mpad : l.rpad);
  l.e_Pt = (size_t)l.npad * l.npad;
  l.e_Dinv = dinv_elems(l.npad);
  l.e_Ap = (size_t)l.mpad * l.npad;
  l.e_Y = (size_t)l.npad * l.mpad;
  l.e_Bw = (size_t)(l.npad > l.mpad ? l.npad : l.mpad) * wmax;
  l.e_S = (size_t)l.mpad * l.mpad;
  l.e_DinvS = dinv_elems(l.mpad);
  l.e_mr = (size_t)l.mpad * l.rpad;
  l.e_nr = (size_t)l.npad * l.rpad;
  l.off_Pt = take(l.e_Pt);
  l.off_Dinv = take(l.e_Dinv);
  l.off_Ap = take(l.e_Ap);
  l.off_Y = take(l.e_Y);
  l.off_Bw = take(l.e_Bw);
  l.off_S = take(l.e_S);
  l.off_DinvS = take(l.e_DinvS);
  l.off_Bp = take(l.e_mr);
  l.off_T1 = take(l.e_mr);
  l.off_T2 = take(l.e_mr);
  l.off_Lam = take(l.e_mr);
  l.off_Z = take(l.e_nr);
  l.off_Xt = take(l.e_nr);
  l.off_X2 = take(l.e_nr);
  l.off_scal = take(4);
  l.total = o;
  return l;
}

static int eq_qp_solve_impl(const double* G, int32_t n, double l2, const double* l2_diag, const double* A,
                            int32_t m, const double* B, int32_t nrhs, double schur_reg, int32_t n_refine,
                            int32_t nprob, double* X, double* stats, void* ws, size_t ws_bytes, void* stream_v,
                            const char* who, const double* AtA = nullptr, const int32_t* perm = nullptr,
                            int32_t a_first_col = 0) {
  if (!G || !A || !X || !stats || !ws) return fail(AGGF_ERR_ARG, "%s: NULL pointer", who);
  if (n <= 0 || m <= 0 || nrhs <= 0 || nprob <= 0) return fail(AGGF_ERR_ARG, "%s: empty problem", who);
  if (!B && nrhs != m) return fail(AGGF_ERR_ARG, "%s: B == NULL needs nrhs == m", who);
  if (!(l2 >= 0.0)) return fail(AGGF_ERR_ARG, "%s: l2 must be >= 0", who);
  if (!(schur_reg >= 0.0) || n_refine < 0 || n_refine > 100) return fail(AGGF_ERR_ARG, "%s: bad schur_reg / n_refine", who);
  if (((uintptr_t)ws & 255) != 0) return fail(AGGF_ERR_ARG, "%s: workspace not 256-byte aligned", who);
  const SolveLayout l = solve_layout(n, m, nrhs, nprob);
  if (ws_bytes < l.total) return fail(AGGF_ERR_WORKSPACE, "%s: workspace too small (%zu < %zu)", who, ws_bytes, l.total);
  if ((int64_t)nprob * 2 * (l.npad / BIG + 1) > 65535) return fail(AGGF_ERR_ARG, "%s: too many problems", who);
  Ctx c;
  c.stream = (hipStream_t)stream_v;
  c.nprob = nprob;
  char* w = (char*)ws;
  const int npad = l.npad, mpad = l.mpad, rpad = l.rpad;
  auto M_ = [&](size_t off, int64_t ld, size_t elems) { return Mat{reinterpret_cast<double*>(w + off), ld, (int64_t)elems}; };
  const Mat Pt = M_(l.off_Pt, npad, l.e_Pt), Dinv = M_(l.off_Dinv, NB, l.e_Dinv), Ap = M_(l.off_Ap, npad, l.e_Ap),
            Y = M_(l.off_Y, mpad, l.e_Y), S = M_(l.off_S, mpad, l.e_S), DinvS = M_(l.off_DinvS, NB, l.e_DinvS),
            Bp = M_(l.off_Bp, rpad, l.e_mr), T1 = M_(l.off_T1, rpad, l.e_mr), T2 = M_(l.off_T2, rpad, l.e_mr),
            Lam = M_(l.off_Lam, rpad, l.e_mr), Z = M_(l.off_Z, rpad, l.e_nr), Xt = M_(l.off_Xt, rpad, l.e_nr),
            X2 = M_(l.off_X2, rpad, l.e_nr);
  const Mat Bw_m = M_(l.off_Bw, mpad, l.e_Bw);  // (npad x mpad) view
  const Mat Bw_r = M_(l.off_Bw, rpad, l.e_Bw);  // (mpad x rpad) view
  double* scal = reinterpret_cast<double*>(w + l.off_scal);  // 4 doubles per problem
  hipStream_t st = c.stream;
  const int np = nprob;
  const int64_t g_ps = (int64_t)n * n, a_ps = (int64_t)m * n, b_ps = (int64_t)m * nrhs;
  // Columns of A (in the factorisation's variable order) before a_first_col are zero in every problem: L^-1 A' is zero
  // above that row, so the forward solve, the Schur complement and every product with A or Y start at the 256-row
  // block holding it.  (Sparse constraint rows -- a slice coordinate map touches 1 + n_basis feature columns per
  // cg site -- with the touched variables permuted to the end: BASELINE config 4 keeps 576 of ~2100-3000 rows.)
  if (a_first_col < 0 || a_first_col > n) return fail(AGGF_ERR_ARG, "%s: a_first_col out of range", who);
  const int ob0 = a_first_col >= n ? (n - 1) / BIG : a_first_col / BIG;  // first 256-row block of Y that is not zero
  const int64_t r0 = (int64_t)(ob0 < npad / BIG ? ob0 : npad / BIG) * BIG;
  const int first_big = (int)(r0 / BIG);

  AGGF_LAUNCH(init_stats_kernel, dim3((unsigned)ceil_div(4 * np, 64)), dim3(64), 0, st, stats, 4 * np);
  AGGF_LAUNCH(max_diag_kernel, dim3(1, np), dim3(256), 0, st, G, n, g_ps, l2, l2_diag, scal, (int64_t)4);
  AGGF_LAUNCH(copy_scalar_kernel, dim3(np), dim3(1), 0, st, scal, (int64_t)4, stats + 3, (int64_t)4);
  AGGF_LAUNCH(build_pt_kernel, flat_grid((int64_t)npad * npad, np), dim3(256), 0, st, G, n, g_ps, npad, l2,
                     l2_diag, scal, (int64_t)4, AtA, perm, Pt.p, Pt.ps);
  AGGF_LAUNCH(pad_copy_kernel, flat_grid((int64_t)mpad * npad, np), dim3(256), 0, st, A, m, n, a_ps, 0, Ap.p,
                     mpad, npad, Ap.ps, perm);
  AGGF_LAUNCH(pad_copy_kernel, flat_grid((int64_t)mpad * rpad, np), dim3(256), 0, st, B,
                     B ? m : (m < nrhs ? m : nrhs), nrhs, b_ps, 0, Bp.p, mpad, rpad, Bp.ps);
  AGGF_LAUNCH_OK();
  // P~ = P/s + A'A  (the factorisation reads the lower triangle only); a caller that knows the structure of its
  // rows has formed A'A itself and build_pt_kernel has added it
  if (!AtA) gemm<true, false>(c, npad, npad, mpad, 1.0, Ap, Ap, 1.0, Pt, 1);
  cholesky(c, Pt, npad, Dinv, stats, 0);
  // Y = L^-1 A'
  AGGF_LAUNCH(pad_copy_kernel, flat_grid((int64_t)npad * mpad, np), dim3(256), 0, st, A, m, n, a_ps, 1, Bw_m.p,
                     npad, mpad, Bw_m.ps, perm);
  solve_lower(c, Pt, npad, Dinv, Bw_m, Y, mpad, first_big);
  // S = Y'Y (identity on the padding), factor it; rows of Y above r0 are zero (and were never written)
  gemm<true, false>(c, mpad, mpad, npad - (int)r0, 1.0, Y.at(r0, 0), Y.at(r0, 0), 0.0, S, 1);
  if (mpad > m) AGGF_LAUNCH(fix_pad_diag_kernel, dim3(1, np), dim3(64), 0, st, S.p, m, mpad, S.ps);
  if (schur_reg > 0.0) AGGF_LAUNCH(schur_reg_kernel, dim3(1, np), dim3(256), 0, st, S.p, m, mpad, S.ps, schur_reg);
  cholesky(c, S, mpad, DinvS, stats, n);
  // Lam = S^-1 Bp ; Xt = L^-T (Y Lam)
  auto schur_solve = [&](Mat rhs, Mat out) {
    if (hipMemcpyAsync(T1.p, rhs.p, (size_t)np * l.e_mr * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess)
      c.rc = fail(AGGF_ERR_HIP, "%s: device copy failed", who);
    solve_lower(c, S, mpad, DinvS, T1, T2, rpad);
    solve_lower_t(c, S, mpad, DinvS, T2, out, rpad);
  };
  schur_solve(Bp, Lam);
  // Z = Y Lam: rows above r0 are zeros
  auto y_times = [&](Mat rhs) {
    if (r0 > 0 && hipMemset2DAsync(Z.p, (size_t)Z.ps * sizeof(double), 0, (size_t)r0 * rpad * sizeof(double), (size_t)np, st) != hipSuccess)
      c.rc = fail(AGGF_ERR_HIP, "%s: device memset failed", who);
    gemm<false, false>(c, npad - (int)r0, rpad, mpad, 1.0, Y.at(r0, 0), rhs, 0.0, Z.at(r0, 0));
  };
  y_times(Lam);
  solve_lower_t(c, Pt, npad, Dinv, Z, Xt, rpad);
  // refinement on the constraint residual R = A Xt - B:  Xt -= P~^-1 A' S^-1 R
  int64_t rb = ceil_div((int64_t)m * nrhs, 256 * 8);
  const unsigned resid_blocks = (unsigned)(rb < 1 ? 1 : rb > 64 ? 64 : rb);
  for (int it = 0; it < n_refine; ++it) {
    gemm<false, false>(c, mpad, rpad, npad - (int)r0, 1.0, Ap.at(0, r0), Xt.at(r0, 0), 0.0, Lam);
    if (it > 0) AGGF_LAUNCH(zero_scalar_kernel, dim3(np), dim3(1), 0, st, scal + 1, (int64_t)4);
    AGGF_LAUNCH(resid_kernel, dim3(resid_blocks, np), dim3(256), 0, st, Lam.p, Bp.p, m, nrhs, rpad, Lam.ps,
                       it == 0 ? stats + 2 : scal + 1, (int64_t)4);
    // padded rows/cols of A Xt - Bp are exact zeros, so the padded residual needs no masking.
    // Bw's storage is reused as the (mpad x rpad) scratch for S^-1 R: per problem it holds at least
    // mpad * rpad elements, but with its own per-problem stride
    const Mat SR{Bw_r.p, rpad, Bw_r.ps};
    schur_solve(Lam, SR);
    y_times(SR);
    solve_lower_t(c, Pt, npad, Dinv, Z, X2, rpad);
    AGGF_LAUNCH(axpy_kernel, flat_grid((int64_t)np * l.e_nr), dim3(256), 0, st, Xt.p, X2.p, -1.0,
                       (int64_t)np * (int64_t)l.e_nr);
  }
  gemm<false, false>(c, mpad, rpad, npad - (int)r0, 1.0, Ap.at(0, r0), Xt.at(r0, 0), 0.0, Lam);
  AGGF_LAUNCH(resid_kernel, dim3(resid_blocks, np), dim3(256), 0, st, Lam.p, Bp.p, m, nrhs, rpad, Lam.ps, stats + 1,
                     (int64_t)4);
  AGGF_LAUNCH(crop_transpose_kernel, flat_grid((int64_t)nrhs * n, np), dim3(256), 0, st, Xt.p, rpad, Xt.ps, n,
                     nrhs, X, perm);
  AGGF_LAUNCH_OK();
  return c.rc;
}

// ---------------------------------------------------------------------------
// One-hot constraint rows (slice coordinate maps: every configuration of BASELINE.json and every test of the
// reference): A x = e_i merely PINS m variables, x[pin[j]] = delta_ij.  With f the free variables,
//     x_f = -P_ff^-1 P[f, pin[i]],
// i.e. one Cholesky factorisation of the (n - m)^2 free block and one pair of triangular solves with m right-hand
// sides -- no A'A product, no Schur complement, no refinement (the constraints hold exactly by construction):
// about 90 of the general path's 330 dependent launches go, 2 ms of 7 at n = 4096, m = 256.

// free[0..n-m) = the indices not in pin, ascending; bad[0] = 1 if a pin is out of range or repeated.  One workgroup.
__global__ __launch_bounds__(256) void pinned_free_list_kernel(const int32_t* __restrict__ pin, int m, int n,
                                                               int32_t* __restrict__ mark, int32_t* __restrict__ free_idx,
                                                               double* __restrict__ stats) {
  __shared__ int part[256];
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  for (int i = threadIdx.x; i < n; i += 256) mark[i] = 0;
  __syncthreads();
  for (int j = threadIdx.x; j < m; j += 256) {
    const int p = pin[j];
    if (p < 0 || p >= n || atomicAdd(&mark[p], 1) != 0) bad = 1;
  }
  __syncthreads();
  const int per = (n + 255) / 256, lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
  int cnt = 0;
  for (int i = lo; i < hi; ++i) cnt += mark[i] == 0;
  part[threadIdx.x] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int t = 0; t < 256; ++t) {
      const int c = part[t];
      part[t] = run;
      run += c;
    }
    if (bad) stats[0] = -1.0;
  }
  __syncthreads();
  int o = part[threadIdx.x];
  for (int i = lo; i < hi; ++i)
    if (mark[i] == 0) free_idx[o++] = i;
}

// Pt ((npad + rpad) x npad): rows < npad = (G[f,f] + l2 diag)/s with the identity on the padding; rows npad + c =
// -G[pin[c], f]/s (zero padded): the right-hand sides B = -P[f, pin], transposed, riding along with the factorisation
__global__ __launch_bounds__(256) void pinned_build_kernel(const double* __restrict__ G, int n, const int32_t* __restrict__ free_idx,
                                                           int nf, const int32_t* __restrict__ pin, int m, int npad, int rpad,
                                                           double l2, const double* __restrict__ l2d,
                                                           const double* __restrict__ scale, double* __restrict__ Pt,
                                                           const double* __restrict__ stats) {
  // a pin out of range or repeated (stats[0] = -1 from pinned_free_list_kernel): the index lists are not a partition
  // of 0..n-1 -- touch nothing (the host reports the status); the factorisation then runs on the workspace as it is
  if (stats[0] < 0.0) {
    const int64_t tot = (int64_t)(npad + rpad) * npad;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
      const int i = (int)(e / npad), j = (int)(e - (int64_t)i * npad);
      Pt[e] = (i == j) ? 1.0 : 0.0;
    }
    return;
  }
  const double inv_s = 1.0 / scale[0];
  const int64_t total = (int64_t)(npad + rpad) * npad;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / npad), j = (int)(e - (int64_t)i * npad);
    double v;
    if (i < npad) {
      if (i < nf && j < nf) {
        const int gi = free_idx[i], gj = free_idx[j];
        v = G[(int64_t)gi * n + gj];
        if (i == j) v += l2 * (l2d ? l2d[gi] : 1.0);
        v *= inv_s;
      } else {
        v = (i == j) ? 1.0 : 0.0;
      }
    } else {
      const int c = i - npad;
      v = (c < m && j < nf) ? -G[(int64_t)pin[c] * n + free_idx[j]] * inv_s : 0.0;
    }
    Pt[e] = v;
  }
}

// X (m x n): X[c, free[i]] = Xt[i, c];  X[c, pin[j]] = (j == c)
__global__ __launch_bounds__(256) void pinned_scatter_kernel(const double* __restrict__ Xt, int rpad,
                                                             const int32_t* __restrict__ free_idx, int nf,
                                                             const int32_t* __restrict__ pin, int m, int n,
                                                             double* __restrict__ X, const double* __restrict__ stats) {
  AGGF_GATED_BODY_BEGIN
  const int64_t total = (int64_t)m * n;
  if (stats[0] < 0.0) {  // bad pins: no scatter through them; X = 0
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) X[e] = 0.0;
    return;
  }
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e / n), k = (int)(e - (int64_t)c * n);  // k-th entry of row c: free ones first, then the pins
    if (k < nf) X[(int64_t)c * n + free_idx[k]] = Xt[(int64_t)k * rpad + c];
    else X[(int64_t)c * n + pin[k - nf]] = (k - nf == c) ? 1.0 : 0.0;
  }
  AGGF_GATED_BODY_END
}

}  // namespace aggf

using namespace aggf;

extern "C" size_t aggf_eq_qp_workspace_bytes(int32_t n, int32_t m, int32_t nrhs) {
  if (n <= 0 || m <= 0 || nrhs <= 0) return 0;
  return solve_layout(n, m, nrhs, 1).total;
}

extern "C" int aggf_eq_qp_solve(const double* G, int32_t n, double l2, const double* l2_diag,
                                const double* A, int32_t m, const double* B, int32_t nrhs,
                                double schur_reg, int32_t n_refine, double* X, double* stats,
                                void* ws, size_t ws_bytes, void* stream_v) {
  return eq_qp_solve_impl(G, n, l2, l2_diag, A, m, B, nrhs, schur_reg, n_refine, 1, X, stats, ws, ws_bytes, stream_v,
                          "aggf_eq_qp_solve");
}

extern "C" size_t aggf_eq_qp_batched_workspace_bytes(int32_t n, int32_t m, int32_t nrhs, int32_t n_problems) {
  if (n <= 0 || m <= 0 || nrhs <= 0 || n_problems <= 0) return 0;
  return solve_layout(n, m, nrhs, n_problems).total;
}

extern "C" int aggf_eq_qp_solve_batched(const double* G, int32_t n, double l2, const double* l2_diag,
                                        const double* A, int32_t m, const double* B, int32_t nrhs,
                                        double schur_reg, int32_t n_refine, int32_t n_problems, double* X,
                                        double* stats, void* ws, size_t ws_bytes, void* stream_v) {
  return eq_qp_solve_impl(G, n, l2, l2_diag, A, m, B, nrhs, schur_reg, n_refine, n_problems, X, stats, ws, ws_bytes,
                          stream_v, "aggf_eq_qp_solve_batched");
}

extern "C" int aggf_eq_qp_solve_batched_shift(const double* G, int32_t n, double l2, const double* l2_diag,
                                              const double* A, const double* AtA, const int32_t* perm,
                                              int32_t a_first_col, int32_t m, const double* B, int32_t nrhs,
                                              double schur_reg, int32_t n_refine, int32_t n_problems, double* X,
                                              double* stats, void* ws, size_t ws_bytes, void* stream_v) {
  if (!AtA) return fail(AGGF_ERR_ARG, "aggf_eq_qp_solve_batched_shift: NULL pointer");
  if (a_first_col != 0 && !perm)
    return fail(AGGF_ERR_ARG, "aggf_eq_qp_solve_batched_shift: a_first_col refers to the order perm gives");
  return eq_qp_solve_impl(G, n, l2, l2_diag, A, m, B, nrhs, schur_reg, n_refine, n_problems, X, stats, ws, ws_bytes,
                          stream_v, "aggf_eq_qp_solve_batched_shift", AtA, perm, a_first_col);
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
  Ctx c;
  c.stream = st;
  AGGF_LAUNCH(pad_copy_kernel, flat_grid((int64_t)npad * npad), dim3(256), 0, st, G, n, n, (int64_t)0, 0, Gp, npad, npad, (int64_t)0);
  AGGF_LAUNCH(pad_copy_kernel, flat_grid((int64_t)mpad * npad), dim3(256), 0, st, X, m, n, (int64_t)0, 0, Xp, mpad, npad, (int64_t)0);
  AGGF_LAUNCH_OK();
  gemm<false, false>(c, mpad, npad, npad, 1.0, Mat{Xp, npad, 0}, Mat{Gp, npad, 0}, 0.0, Mat{Y, npad, 0});  // Y = X G
  if (c.rc) return c.rc;
  AGGF_LAUNCH(rowdot_kernel, dim3(m), dim3(256), 0, st, Xp, (int64_t)npad, Y, (int64_t)npad, n, q);
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_daxpby(int64_t n, double a, const double* x, double b, const double* y, double* out,
                           void* stream_v) {
  hipStream_t st = (hipStream_t)stream_v;
  if (!x || !y || !out) return fail(AGGF_ERR_ARG, "aggf_daxpby: NULL pointer");
  if (n <= 0) return AGGF_OK;
  AGGF_LAUNCH(axpby_kernel, flat_grid(n), dim3(256), 0, st, n, a, x, b, y, out);
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

// ---- one-hot constraint rows: m pinned variables (see pinned_* kernels) -------------------------------
struct PinnedLayout {
  int npad, rpad;
  size_t off_Pt, off_Dinv, off_Z, off_Xt, off_scal, off_idx, total;
};
static PinnedLayout pinned_layout(int n, int m) {
  PinnedLayout l;
  l.npad = (int)round_up(n - m, NB);
  l.rpad = (int)round_up(m, NB);
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t r = o;
    o += (size_t)round_up((int64_t)bytes, 256);
    return r;
  };
  l.off_Pt = take((size_t)(l.npad + l.rpad) * l.npad * 8);
  l.off_Dinv = take(dinv_elems(l.npad) * 8);
  l.off_Z = take((size_t)l.npad * l.rpad * 8);
  l.off_Xt = take((size_t)l.npad * l.rpad * 8);
  l.off_scal = take(32);
  l.off_idx = take((size_t)n * 8);
  l.total = o;
  return l;
}

extern "C" size_t aggf_eq_qp_pinned_workspace_bytes(int32_t n, int32_t m) {
  if (n <= 0 || m <= 0 || m >= n) return 0;
  return pinned_layout(n, m).total;
}

extern "C" int aggf_eq_qp_solve_pinned(const double* G, int32_t n, double l2, const double* l2_diag,
                                       const int32_t* pin_idx, int32_t m, double* X, double* stats, void* ws,
                                       size_t ws_bytes, void* stream_v) {
  const char* who = "aggf_eq_qp_solve_pinned";
  if (!G || !pin_idx || !X || !stats || !ws) return fail(AGGF_ERR_ARG, "%s: NULL pointer", who);
  if (n <= 0 || m <= 0 || m >= n) return fail(AGGF_ERR_ARG, "%s: needs 0 < m < n", who);
  if (n > 256 * 4096) return fail(AGGF_ERR_ARG, "%s: n too large", who);
  if (!(l2 >= 0.0)) return fail(AGGF_ERR_ARG, "%s: l2 must be >= 0", who);
  if (((uintptr_t)ws & 255) != 0) return fail(AGGF_ERR_ARG, "%s: workspace not 256-byte aligned", who);
  const int nf = n - m;
  const PinnedLayout l = pinned_layout(n, m);
  if (ws_bytes < l.total) return fail(AGGF_ERR_WORKSPACE, "%s: workspace too small", who);
  Ctx c;
  c.stream = (hipStream_t)stream_v;
  c.nprob = 1;
  hipStream_t st = c.stream;
  char* w = (char*)ws;
  const int npad = l.npad, rpad = l.rpad;
  const Mat Pt{reinterpret_cast<double*>(w + l.off_Pt), npad, 0}, Dinv{reinterpret_cast<double*>(w + l.off_Dinv), NB, 0},
      Z{reinterpret_cast<double*>(w + l.off_Z), rpad, 0}, Xt{reinterpret_cast<double*>(w + l.off_Xt), rpad, 0};
  double* scal = reinterpret_cast<double*>(w + l.off_scal);
  int32_t* free_idx = reinterpret_cast<int32_t*>(w + l.off_idx);
  int32_t* mark = free_idx + n;
  AGGF_LAUNCH(init_stats_kernel, dim3(1), dim3(64), 0, st, stats, 4);
  AGGF_LAUNCH(pinned_free_list_kernel, dim3(1), dim3(256), 0, st, pin_idx, m, n, mark, free_idx, stats);
  AGGF_LAUNCH(max_diag_kernel, dim3(1, 1), dim3(256), 0, st, G, n, (int64_t)0, l2, l2_diag, scal, (int64_t)4);
  AGGF_LAUNCH(copy_scalar_kernel, dim3(1), dim3(1), 0, st, scal, (int64_t)4, stats + 3, (int64_t)4);
  AGGF_LAUNCH(pinned_build_kernel, flat_grid((int64_t)(npad + rpad) * npad), dim3(256), 0, st, G, n, free_idx, nf,
                     pin_idx, m, npad, rpad, l2, l2_diag, scal, Pt.p, stats);
  AGGF_LAUNCH_OK();
  // the rows below the matrix leave the factorisation as Y' = B' L^-T: no forward solve of its own
  cholesky(c, Pt, npad, Dinv, stats, 0, rpad);
  AGGF_LAUNCH(pad_copy_kernel, flat_grid((int64_t)npad * rpad), dim3(256), 0, st, Pt.p + (int64_t)npad * npad, rpad, npad,
                     (int64_t)0, 1, Z.p, npad, rpad, (int64_t)0);
  AGGF_LAUNCH_OK();
  solve_lower_t(c, Pt, npad, Dinv, Z, Xt, rpad);
  AGGF_LAUNCH_GATED(1024, pinned_scatter_kernel, flat_grid((int64_t)m * n), dim3(256), 0, st, Xt.p, rpad, free_idx, nf, pin_idx, m,
                     n, X, stats);
  AGGF_LAUNCH_OK();
  return c.rc;
}
