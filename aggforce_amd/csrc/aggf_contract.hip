// Per-frame contractions of the featurised path and of per-frame (configuration-dependent) maps.
// All of them stream one large operand from HBM exactly once (the (T, n_cg, N) per-frame map or
// the (T, N, n_feat) feature tensor) and do ~6 flop per element read, so they are HBM-bound:
// coalesced reads along the innermost axis, the small operand of a frame kept in registers / LDS,
// float64 accumulation, wave reductions, no atomics (fixed summation order).
//
//   trjdot_frames_kernel   util.trjdot with a 3-D factor (util.py:119-125), CLAMap.__call__
//                          (map/core.py:428-430):  out[t,c,d] = sum_f factor[t,c,f] points[t,f,d]
//   feat_contract_kernel   regression matrix of a dense featuriser (featlinearmap.py:361-369) and the
//                          force term of CLAMap's application (512-520):
//                          out[t,f,d] = sum_a feat[t,a,f] F[t,a,d] + alpha div[t,f,d]
//   feat_rows_kernel       _constr_arrays (featlinearmap.py:445-459), dense features:
//                          A[(s,c),f] = sum_a M[c,a] feat[idx[s],a,f],  b[(s,c)] = [c == site]
//   gb_rows_kernel         the same rows for the fused [id_feat | gb_feat] features, from the compact
//                          per-channel Gaussians (never forming the one-hot feature tensor)
//   gb_ata_kernel          A'A of those rows from their structure (S multiply-adds per entry instead of S n_cg)
//   feat_weights_kernel    scale_f of _feat_linear_mapping (featlinearmap.py:512-515):
//                          w[t,a] = sum_f feat[t,a,f] coef[f]
#include "aggf_common.h"

namespace aggf {

template <typename T>
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// ---------------------------------------------------------------------------
// One wave = RW consecutive cg rows of one frame; lanes stride over the fine-grained sites, so
// the factor rows are read in 256/512-byte coalesced pieces and each points element loaded
// serves RW rows.
constexpr int TRJ_RW = 4;

template <typename TFa, typename TP, typename TO>
__global__ __launch_bounds__(256) void trjdot_frames_kernel(const TP* __restrict__ P, const TFa* __restrict__ Fa,
                                                            int64_t T, int32_t N, int32_t n_cg,
                                                            const TO* __restrict__ trans, TO* __restrict__ out,
                                                            int vec_ok) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int row_blocks = (n_cg + TRJ_RW - 1) / TRJ_RW;
  const int64_t tasks = T * row_blocks;
  for (int64_t task = wid; task < tasks; task += nw) {
    const int64_t t = task / row_blocks;
    const int c0 = (int)(task - t * row_blocks) * TRJ_RW;
    const TP* p = P + t * (int64_t)N * 3;
    const TFa* fa = Fa + (t * n_cg + c0) * (int64_t)N;
    double acc[TRJ_RW][3];
#pragma unroll
    for (int r = 0; r < TRJ_RW; ++r) acc[r][0] = acc[r][1] = acc[r][2] = 0.0;
    if (vec_ok && (3 * (16 / (int)sizeof(TFa))) % (16 / (int)sizeof(TP)) == 0) {
      // 16-byte loads: lane = VF consecutive sites (VF = 4 floats / 2 doubles of the factor); their 3 VF point
      // values are 3 VF consecutive elements
      constexpr int VF = 16 / (int)sizeof(TFa);
      typedef TFa __attribute__((ext_vector_type(VF))) vfa_t;
      constexpr int VP = 16 / (int)sizeof(TP);
      typedef TP __attribute__((ext_vector_type(VP))) vp_t;
      for (int f0 = lane * VF; f0 < N; f0 += 64 * VF) {
        double xyz[3 * VF];
        const TP* pp = p + (int64_t)f0 * 3;
#pragma unroll
        for (int v = 0; v < 3 * VF / VP; ++v) {
          const vp_t pv = *reinterpret_cast<const vp_t*>(pp + v * VP);
#pragma unroll
          for (int e = 0; e < VP; ++e) xyz[v * VP + e] = (double)pv[e];
        }
#pragma unroll
        for (int r = 0; r < TRJ_RW; ++r) {
          if (c0 + r < n_cg) {
            const vfa_t wv = *reinterpret_cast<const vfa_t*>(fa + (int64_t)r * N + f0);
#pragma unroll
            for (int e = 0; e < VF; ++e) {
              const double w = (double)wv[e];
              acc[r][0] = fma(w, xyz[3 * e + 0], acc[r][0]);
              acc[r][1] = fma(w, xyz[3 * e + 1], acc[r][1]);
              acc[r][2] = fma(w, xyz[3 * e + 2], acc[r][2]);
            }
          }
        }
      }
    } else
    for (int f = lane; f < N; f += 64) {
      const double x = (double)p[(int64_t)f * 3 + 0], y = (double)p[(int64_t)f * 3 + 1],
                   z = (double)p[(int64_t)f * 3 + 2];
#pragma unroll
      for (int r = 0; r < TRJ_RW; ++r) {
        if (c0 + r < n_cg) {
          const double w = (double)fa[(int64_t)r * N + f];
          acc[r][0] = fma(w, x, acc[r][0]);
          acc[r][1] = fma(w, y, acc[r][1]);
          acc[r][2] = fma(w, z, acc[r][2]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < TRJ_RW; ++r)
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const double s = wave_sum<double>(acc[r][d]);
        if (lane == 0 && c0 + r < n_cg) {
          const int64_t o = (t * n_cg + c0 + r) * 3 + d;
          out[o] = trans ? (TO)s + trans[o] : (TO)s;  // rounded product first, then the sum: trjdot(...) + trans
        }
      }
  }
}

// ---------------------------------------------------------------------------
// Workgroup = 256 feature columns of one frame; thread = one column.  The frame's forces go
// through LDS in tiles of FC_TILE atoms (broadcast reads); feat rows are read coalesced along f.
constexpr int FC_TILE = 128;

template <typename TX, typename TF, typename TO>
__global__ __launch_bounds__(256) void feat_contract_kernel(const TF* __restrict__ F, const TX* __restrict__ feat,
                                                            const TX* __restrict__ div, double alpha, int64_t T,
                                                            int32_t N, int32_t n_feat, int32_t ld,
                                                            TO* __restrict__ out) {
  __shared__ double fs[FC_TILE * 3];
  const int col_blocks = (ld + 255) / 256;
  const int64_t tasks = T * col_blocks;
  for (int64_t task = blockIdx.x; task < tasks; task += gridDim.x) {
    const int64_t t = task / col_blocks;
    const int f = (int)(task - t * col_blocks) * 256 + threadIdx.x;
    const bool live = f < n_feat;
    const TX* x = feat + t * (int64_t)N * n_feat + f;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int at0 = 0; at0 < N; at0 += FC_TILE) {
      const int na = N - at0 < FC_TILE ? N - at0 : FC_TILE;
      __syncthreads();
      for (int e = threadIdx.x; e < na * 3; e += 256) fs[e] = (double)F[(t * N + at0) * 3 + e];
      __syncthreads();
      if (live) {
        int a = 0;
        for (; a + 4 <= na; a += 4) {
          const double v0 = (double)x[(int64_t)(at0 + a) * n_feat], v1 = (double)x[(int64_t)(at0 + a + 1) * n_feat],
                       v2 = (double)x[(int64_t)(at0 + a + 2) * n_feat], v3 = (double)x[(int64_t)(at0 + a + 3) * n_feat];
          a0 = fma(v0, fs[a * 3 + 0], a0); a1 = fma(v0, fs[a * 3 + 1], a1); a2 = fma(v0, fs[a * 3 + 2], a2);
          a0 = fma(v1, fs[a * 3 + 3], a0); a1 = fma(v1, fs[a * 3 + 4], a1); a2 = fma(v1, fs[a * 3 + 5], a2);
          a0 = fma(v2, fs[a * 3 + 6], a0); a1 = fma(v2, fs[a * 3 + 7], a1); a2 = fma(v2, fs[a * 3 + 8], a2);
          a0 = fma(v3, fs[a * 3 + 9], a0); a1 = fma(v3, fs[a * 3 + 10], a1); a2 = fma(v3, fs[a * 3 + 11], a2);
        }
        for (; a < na; ++a) {
          const double v = (double)x[(int64_t)(at0 + a) * n_feat];
          a0 = fma(v, fs[a * 3 + 0], a0); a1 = fma(v, fs[a * 3 + 1], a1); a2 = fma(v, fs[a * 3 + 2], a2);
        }
      }
    }
    if (f < ld) {
      TO* o = out + (t * ld + f) * 3;
      if (live) {
        if (div) {
          const TX* dv = div + (t * n_feat + f) * 3;
          a0 = fma(alpha, (double)dv[0], a0);
          a1 = fma(alpha, (double)dv[1], a1);
          a2 = fma(alpha, (double)dv[2], a2);
        }
        o[0] = (TO)a0; o[1] = (TO)a1; o[2] = (TO)a2;
      } else {
        o[0] = o[1] = o[2] = (TO)0;  // padding columns of K1's layout
      }
    }
  }
}

// ---------------------------------------------------------------------------
// grid = (feature blocks of 64, cg blocks of 16, sampled frames); 4 waves, wave w owns cg rows
// c0+4w .. c0+4w+3 of the block, lane = feature column.  M tile (16 x 64 atoms) in LDS.
template <typename TX>
__global__ __launch_bounds__(256) void feat_rows_kernel(const TX* __restrict__ feat, int32_t N, int32_t n_feat,
                                                        const int64_t* __restrict__ idx,
                                                        const double* __restrict__ M, int32_t n_cg, int32_t site,
                                                        double* __restrict__ A, double* __restrict__ b) {
  __shared__ double ms[16][64 + 1];
  const int s = blockIdx.z;
  const int c0 = blockIdx.y * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int f = blockIdx.x * 64 + lane;
  const TX* x = feat + idx[s] * (int64_t)N * n_feat + f;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int a0 = 0; a0 < N; a0 += 64) {
    __syncthreads();
    for (int e = threadIdx.x; e < 16 * 64; e += 256) {
      const int c = e >> 6, a = e & 63;
      ms[c][a] = (c0 + c < n_cg && a0 + a < N) ? M[(int64_t)(c0 + c) * N + a0 + a] : 0.0;
    }
    __syncthreads();
    if (f < n_feat) {
      const int na = N - a0 < 64 ? N - a0 : 64;
      for (int a = 0; a < na; ++a) {
        const double v = (double)x[(int64_t)(a0 + a) * n_feat];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = fma(ms[wave * 4 + q][a], v, acc[q]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = c0 + wave * 4 + q;
    if (c < n_cg && f < n_feat) A[((int64_t)s * n_cg + c) * n_feat + f] = acc[q];
  }
  if (blockIdx.x == 0 && threadIdx.x < 16 && c0 + (int)threadIdx.x < n_cg)
    b[(int64_t)s * n_cg + c0 + threadIdx.x] = (c0 + (int)threadIdx.x == site) ? 1.0 : 0.0;
}

// A[(s,c), g] = Mg[c,g] (g < n_id);  A[(s,c), n_id + j] = Mg[c,ch] gauss[s,ch,k] with (ch,k) the j-th
// Gaussian column (all n_ch*nb of them if cols == NULL, else cols[j] = ch*nb + k);  b one-hot on `site`.
// A has row stride ld >= n_id + n_cols; columns beyond are written as zeros.
template <typename TG>
__global__ __launch_bounds__(256) void gb_rows_kernel(const double* __restrict__ Mg, const TG* __restrict__ gauss,
                                                      int32_t S, int32_t n_cg, int32_t G, int32_t n_id, int32_t n_ch,
                                                      int32_t nb, const int32_t* __restrict__ cols, int32_t n_cols,
                                                      int32_t ld, int32_t site, double* __restrict__ A,
                                                      double* __restrict__ b) {
  const int n_feat = n_id + n_cols;
  const int64_t total = (int64_t)S * n_cg * ld;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(e % ld);
    const int64_t sc = e / ld;
    const int c = (int)(sc % n_cg);
    const int64_t s = sc / n_cg;
    double v = 0.0;
    if (f < n_id) {
      v = Mg[(int64_t)c * G + f];
    } else if (f < n_feat) {
      const int full = cols ? cols[f - n_id] : f - n_id;
      const int ch = full / nb, k = full - ch * nb;
      v = Mg[(int64_t)c * G + ch] * (double)gauss[(s * n_ch + ch) * nb + k];
    }
    A[e] = v;
    if (f == 0) b[sc] = (c == site) ? 1.0 : 0.0;
  }
}

// M2 = Mg' Mg (G x G): the overlap of the constraint-adjusted mapping rows, shared by every site's A'A
__global__ __launch_bounds__(256) void gb_overlap_kernel(const double* __restrict__ Mg, int32_t n_cg, int32_t G,
                                                         double* __restrict__ M2) {
  const int64_t total = (int64_t)G * G;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / G), j = (int)(e - (int64_t)i * G);
    double acc = 0.0;
    for (int c = 0; c < n_cg; ++c) acc = fma(Mg[(int64_t)c * G + i], Mg[(int64_t)c * G + j], acc);
    M2[e] = acc;
  }
}

// A'A of gb_rows_kernel's rows WITHOUT forming the product over the S n_cg rows.  Row (s, c) of A is
// Mg[c, g(f)] w_s(f) with g(f) the constraint group behind column f (f itself for an id column, the channel for a
// Gaussian column) and w_s(f) = 1 or gauss[s, ch, k], so
//     (A'A)[f, f'] = M2[g(f), g(f')] * sum_s w_s(f) w_s(f'):
// S multiply-adds per entry instead of S n_cg (BASELINE config 4: 20 against 1280, and the n x n x 1280 product was a
// quarter of the batched solve's flops).  One workgroup = one 64 x 64 tile of the LOWER triangle (the factorisation
// reads nothing else), each thread a 4 x 4 block of it; the tile's w columns sit in LDS in chunks of GA_SC frames.
// Entries beyond the n_feat columns (up to ld) are written as zeros.
constexpr int GA_SC = 32;

template <typename TG>
__global__ __launch_bounds__(256) void gb_ata_kernel(const double* __restrict__ M2, const TG* __restrict__ gauss,
                                                     int32_t S, int32_t G, int32_t n_id, int32_t n_ch, int32_t nb,
                                                     const int32_t* __restrict__ cols, int32_t n_cols, int32_t ld,
                                                     double* __restrict__ out) {
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj > bi) return;
  __shared__ double wI[GA_SC][64], wJ[GA_SC][64];
  __shared__ int gI[64], gJ[64], oI[64], oJ[64];  // group of a column; offset of its Gaussian in a frame (-1: w = 1, -2: w = 0)
  const int tid = threadIdx.x, n_feat = n_id + n_cols;
  if (tid < 128) {
    const int f = (tid < 64 ? bi : bj) * 64 + (tid & 63);
    int g = 0, o = -2;
    if (f < n_id) {
      g = f;
      o = -1;
    } else if (f < n_feat) {
      const int full = cols ? cols[f - n_id] : f - n_id;
      g = full / nb;
      o = full;  // (ch * nb + k): gauss[(s * n_ch) * nb + full]
    }
    if (tid < 64) {
      gI[tid] = g;
      oI[tid] = o;
    } else {
      gJ[tid - 64] = g;
      oJ[tid - 64] = o;
    }
  }
  const int ty = tid >> 4, tx = tid & 15;
  double acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[r][q] = 0.0;
  for (int s0 = 0; s0 < S; s0 += GA_SC) {
    __syncthreads();
    const int ns = S - s0 < GA_SC ? S - s0 : GA_SC;
    for (int e = tid; e < 2 * GA_SC * 64; e += 256) {
      const int side = e / (GA_SC * 64), r = (e / 64) % GA_SC, col = e & 63;
      const int o = side ? oJ[col] : oI[col];
      double w = 0.0;
      if (r < ns) w = o == -1 ? 1.0 : (o >= 0 ? (double)gauss[((int64_t)(s0 + r) * n_ch) * nb + o] : 0.0);
      (side ? wJ : wI)[r][col] = w;
    }
    __syncthreads();
    for (int r = 0; r < ns; ++r) {
      double a[4], b[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a[q] = wI[r][ty * 4 + q];
        b[q] = wJ[r][tx * 4 + q];
      }
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] = fma(a[p], b[q], acc[p][q]);
    }
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int i = bi * 64 + ty * 4 + p;
    if (i >= ld) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = bj * 64 + tx * 4 + q;
      if (j >= ld) continue;
      double v = 0.0;
      if (i < n_feat && j < n_feat) v = acc[p][q] * M2[(int64_t)gI[ty * 4 + p] * G + gJ[tx * 4 + q]];
      out[(int64_t)i * ld + j] = v;
    }
  }
}

// w[t*ld_t + a] = sum_f feat[t,a,f] coef[f]; one wave per (t,a) row
template <typename TX>
__global__ __launch_bounds__(256) void feat_weights_kernel(const TX* __restrict__ feat, int64_t T, int32_t N,
                                                           int32_t n_feat, const double* __restrict__ coef,
                                                           int64_t ld_t, double* __restrict__ w, int vec_ok) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t row = wid; row < T * N; row += nw) {
    const TX* x = feat + row * n_feat;
    double acc = 0.0;
    if (vec_ok) {
      constexpr int VX = 16 / (int)sizeof(TX);
      typedef TX __attribute__((ext_vector_type(VX))) vx_t;
      for (int f0 = lane * VX; f0 < n_feat; f0 += 64 * VX) {
        const vx_t xv = *reinterpret_cast<const vx_t*>(x + f0);
#pragma unroll
        for (int e = 0; e < VX; ++e) acc = fma((double)xv[e], coef[f0 + e], acc);
      }
    } else {
      for (int f = lane; f < n_feat; f += 64) acc = fma((double)x[f], coef[f], acc);
    }
    acc = wave_sum<double>(acc);
    if (lane == 0) {
      const int64_t t = row / N;
      w[t * ld_t + (row - t * N)] = acc;
    }
  }
}

static inline dim3 stream_grid(int64_t blocks) {
  if (blocks > 65536) blocks = 65536;
  if (blocks < 1) blocks = 1;
  return dim3((unsigned)blocks);
}

}  // namespace aggf

using namespace aggf;

extern "C" int aggf_trjdot_frames(const void* points, int p_dtype, const void* factor, int f_dtype, int64_t T,
                                  int32_t N, int32_t n_cg, const void* trans, void* out, int out_dtype,
                                  void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!points || !factor || !out) return fail(AGGF_ERR_ARG, "aggf_trjdot_frames: NULL pointer");
  if (T <= 0 || N <= 0 || n_cg <= 0) return fail(AGGF_ERR_ARG, "aggf_trjdot_frames: empty problem");
  const bool pf = p_dtype == AGGF_F64, ff = f_dtype == AGGF_F64, of = out_dtype == AGGF_F64;
  if ((p_dtype != AGGF_F32 && !pf) || (f_dtype != AGGF_F32 && !ff) || (out_dtype != AGGF_F32 && !of))
    return fail(AGGF_ERR_ARG, "aggf_trjdot_frames: bad dtype");
  if (of != (pf || ff)) return fail(AGGF_ERR_ARG, "aggf_trjdot_frames: out dtype must be the promoted dtype");
  const dim3 grid = stream_grid(ceil_div(T * ceil_div(n_cg, TRJ_RW), 4)), block(256);
  // 16-byte loads need every factor row and every frame of points to start 16-byte aligned and hold whole vectors
  const int vf = ff ? 2 : 4;
  // (double factor with float points: a lane's 2 sites are 6 floats = 24 bytes of points, not whole vectors)
  const int vec_ok = !(ff && !pf) && (N % vf == 0) && (((uintptr_t)factor & 15) == 0) &&
                     (((uintptr_t)points & 15) == 0) && (((int64_t)N * 3 * (pf ? 8 : 4)) % 16 == 0);
#define AGGF_TRJ(TFa, TP, TO)                                                                                    \
  AGGF_LAUNCH((trjdot_frames_kernel<TFa, TP, TO>), grid, block, 0, stream, (const TP*)points,               \
                     (const TFa*)factor, T, N, n_cg, (const TO*)trans, (TO*)out, vec_ok)
  if (!pf && !ff) AGGF_TRJ(float, float, float);
  else if (pf && !ff) AGGF_TRJ(float, double, double);
  else if (!pf && ff) AGGF_TRJ(double, float, double);
  else AGGF_TRJ(double, double, double);
#undef AGGF_TRJ
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_feat_contract(const void* forces, int f_dtype, const void* feat, const void* div, int x_dtype,
                                  double alpha, int64_t T, int32_t N, int32_t n_feat, int32_t ld, void* out,
                                  int out_dtype, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!forces || !feat || !out) return fail(AGGF_ERR_ARG, "aggf_feat_contract: NULL pointer");
  if (T <= 0 || N <= 0 || n_feat <= 0 || ld < n_feat) return fail(AGGF_ERR_ARG, "aggf_feat_contract: bad shape");
  const bool ff = f_dtype == AGGF_F64, xf = x_dtype == AGGF_F64, of = out_dtype == AGGF_F64;
  if ((f_dtype != AGGF_F32 && !ff) || (x_dtype != AGGF_F32 && !xf) || (out_dtype != AGGF_F32 && !of))
    return fail(AGGF_ERR_ARG, "aggf_feat_contract: bad dtype");
  if (of != (ff || xf)) return fail(AGGF_ERR_ARG, "aggf_feat_contract: out dtype must be the promoted dtype");
  const dim3 grid = stream_grid(T * ceil_div(ld, 256)), block(256);
#define AGGF_FC(TX, TF, TO)                                                                                      \
  AGGF_LAUNCH((feat_contract_kernel<TX, TF, TO>), grid, block, 0, stream, (const TF*)forces,               \
                     (const TX*)feat, (const TX*)div, alpha, T, N, n_feat, ld, (TO*)out)
  if (!xf && !ff) AGGF_FC(float, float, float);
  else if (!xf && ff) AGGF_FC(float, double, double);
  else if (xf && !ff) AGGF_FC(double, float, double);
  else AGGF_FC(double, double, double);
#undef AGGF_FC
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_feat_constraint_rows(const void* feat, int x_dtype, int64_t T, int32_t N, int32_t n_feat,
                                         const int64_t* frame_idx, int32_t S, const double* M, int32_t n_cg,
                                         int32_t site, double* A, double* b, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!feat || !frame_idx || !M || !A || !b) return fail(AGGF_ERR_ARG, "aggf_feat_constraint_rows: NULL pointer");
  if (T <= 0 || N <= 0 || n_feat <= 0 || S <= 0 || S > 65535 || n_cg <= 0 || site < 0 || site >= n_cg)
    return fail(AGGF_ERR_ARG, "aggf_feat_constraint_rows: bad shape");
  const dim3 grid((unsigned)ceil_div(n_feat, 64), (unsigned)ceil_div(n_cg, 16), (unsigned)S), block(256);
  if (grid.y > 65535) return fail(AGGF_ERR_ARG, "aggf_feat_constraint_rows: too many cg sites");
  if (x_dtype == AGGF_F32)
    AGGF_LAUNCH(feat_rows_kernel<float>, grid, block, 0, stream, (const float*)feat, N, n_feat, frame_idx, M, n_cg, site, A, b);
  else if (x_dtype == AGGF_F64)
    AGGF_LAUNCH(feat_rows_kernel<double>, grid, block, 0, stream, (const double*)feat, N, n_feat, frame_idx, M, n_cg, site, A, b);
  else
    return fail(AGGF_ERR_ARG, "aggf_feat_constraint_rows: bad dtype");
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_gb_group_overlap(const double* Mg, int32_t n_cg, int32_t G, double* M2, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!Mg || !M2) return fail(AGGF_ERR_ARG, "aggf_gb_group_overlap: NULL pointer");
  if (n_cg <= 0 || G <= 0) return fail(AGGF_ERR_ARG, "aggf_gb_group_overlap: bad shape");
  AGGF_LAUNCH(gb_overlap_kernel, stream_grid(ceil_div((int64_t)G * G, 256)), dim3(256), 0, stream, Mg, n_cg, G, M2);
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_gb_constraint_gram(const double* M2, const void* gauss, int g_dtype, int32_t S, int32_t G,
                                       int32_t n_id, int32_t n_ch, int32_t n_basis, const int32_t* cols,
                                       int32_t n_cols, int32_t ld, double* AtA, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!cols) n_cols = n_ch * n_basis;
  if (!M2 || !AtA || (n_cols > 0 && !gauss)) return fail(AGGF_ERR_ARG, "aggf_gb_constraint_gram: NULL pointer");
  if (S <= 0 || G <= 0 || n_id < 0 || n_id > G || n_ch < 0 || n_ch > G || n_basis <= 0 || n_cols < 0 ||
      n_cols > n_ch * n_basis || n_id + n_cols <= 0 || ld < n_id + n_cols)
    return fail(AGGF_ERR_ARG, "aggf_gb_constraint_gram: bad shape");
  const unsigned nt = (unsigned)ceil_div(ld, 64);
  if (nt > 65535) return fail(AGGF_ERR_ARG, "aggf_gb_constraint_gram: too many columns");
  const dim3 grid(nt, nt), block(256);
  if (g_dtype == AGGF_F32)
    AGGF_LAUNCH(gb_ata_kernel<float>, grid, block, 0, stream, M2, (const float*)gauss, S, G, n_id, n_ch, n_basis,
                       cols, n_cols, ld, AtA);
  else if (g_dtype == AGGF_F64)
    AGGF_LAUNCH(gb_ata_kernel<double>, grid, block, 0, stream, M2, (const double*)gauss, S, G, n_id, n_ch,
                       n_basis, cols, n_cols, ld, AtA);
  else
    return fail(AGGF_ERR_ARG, "aggf_gb_constraint_gram: bad dtype");
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_gb_constraint_rows(const double* Mg, const void* gauss, int g_dtype, int32_t S, int32_t n_cg, int32_t G,
                                       int32_t n_id, int32_t n_ch, int32_t n_basis, const int32_t* cols,
                                       int32_t n_cols, int32_t ld, int32_t site, double* A, double* b,
                                       void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!cols) n_cols = n_ch * n_basis;
  if (!Mg || !A || !b || (n_cols > 0 && !gauss)) return fail(AGGF_ERR_ARG, "aggf_gb_constraint_rows: NULL pointer");
  if (S <= 0 || n_cg <= 0 || G <= 0 || n_id < 0 || n_id > G || n_ch < 0 || n_ch > G || n_basis <= 0 || n_cols < 0 ||
      n_cols > n_ch * n_basis || n_id + n_cols <= 0 || ld < n_id + n_cols || site < 0 || site >= n_cg)
    return fail(AGGF_ERR_ARG, "aggf_gb_constraint_rows: bad shape");
  const int64_t total = (int64_t)S * n_cg * ld;
  if (g_dtype == AGGF_F32)
    AGGF_LAUNCH(gb_rows_kernel<float>, stream_grid(ceil_div(total, 256)), dim3(256), 0, stream, Mg,
                       (const float*)gauss, S, n_cg, G, n_id, n_ch, n_basis, cols, n_cols, ld, site, A, b);
  else if (g_dtype == AGGF_F64)
    AGGF_LAUNCH(gb_rows_kernel<double>, stream_grid(ceil_div(total, 256)), dim3(256), 0, stream, Mg,
                       (const double*)gauss, S, n_cg, G, n_id, n_ch, n_basis, cols, n_cols, ld, site, A, b);
  else
    return fail(AGGF_ERR_ARG, "aggf_gb_constraint_rows: bad feature dtype");
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_feat_weights(const void* feat, int x_dtype, int64_t T, int32_t N, int32_t n_feat,
                                 const double* coef, int64_t ld_t, double* w, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!feat || !coef || !w) return fail(AGGF_ERR_ARG, "aggf_feat_weights: NULL pointer");
  if (T <= 0 || N <= 0 || n_feat <= 0 || ld_t < N) return fail(AGGF_ERR_ARG, "aggf_feat_weights: bad shape");
  const dim3 grid = stream_grid(ceil_div(T * N, 4)), block(256);
  const int vec_ok = (((uintptr_t)feat & 15) == 0) && (n_feat % (x_dtype == AGGF_F64 ? 2 : 4) == 0);
  if (x_dtype == AGGF_F32)
    AGGF_LAUNCH(feat_weights_kernel<float>, grid, block, 0, stream, (const float*)feat, T, N, n_feat, coef, ld_t, w, vec_ok);
  else if (x_dtype == AGGF_F64)
    AGGF_LAUNCH(feat_weights_kernel<double>, grid, block, 0, stream, (const double*)feat, T, N, n_feat, coef, ld_t, w, vec_ok);
  else
    return fail(AGGF_ERR_ARG, "aggf_feat_weights: bad dtype");
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}
