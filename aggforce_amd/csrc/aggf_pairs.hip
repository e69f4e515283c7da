// K6: pair-distance fluctuation statistics for guess_pairwise_constraints.
//
// Replaces constraints/constfinder.py:46-53 of the reference -- util.distances (util.py:65-72)
// materialises the (T, N, N) distance tensor and takes np.var over frames; here one streaming pass
// accumulates, for every pair i < j, sum_t (d_t - d_0) and sum_t (d_t - d_0)^2 with d_0 the
// distance in the first frame (shifted sums: no cancellation for rigid pairs), from which
//     var[i,j] = E[(d-d0)^2] - E[d-d0]^2.
// One workgroup = one 64x64 tile of pairs x one frame range; thread = 4x4 pairs in registers;
// 8 frames of the two 64-atom position blocks are staged in LDS per barrier.  Partial sums per
// frame range go to slabs and are combined in a fixed order (deterministic).
#include "aggf_common.h"

namespace aggf {

constexpr int PT = 64;   // pair tile edge
constexpr int PFB = 8;   // frames per LDS stage

template <typename TIn>
__global__ __launch_bounds__(256) void pair_stats_kernel(const TIn* __restrict__ X, int64_t T, int32_t N,
                                                         int32_t nt1, int32_t n_tiles,
                                                         int64_t frames_per_split,
                                                         double* __restrict__ slabs) {
  __shared__ double sp[2][PFB][PT][3];  // [i-block | j-block][frame][atom][xyz]
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int ks = b / n_tiles;
  int idx = b - ks * n_tiles, ti = 0;
  {
    int rowlen = nt1;
    while (idx >= rowlen) {
      idx -= rowlen;
      --rowlen;
      ++ti;
    }
  }
  const int tj = ti + idx;
  const int tile_lin = b - ks * n_tiles;
  const int64_t t_begin = (int64_t)ks * frames_per_split;
  int64_t t_end = t_begin + frames_per_split;
  if (t_end > T) t_end = T;
  const int bi = (tid >> 4) * 4, bj = (tid & 15) * 4;  // this thread's 4x4 pairs inside the tile

  auto load_atom = [&](int64_t t, int a, double out[3]) {
    if (a < N) {
      const TIn* p = X + (t * N + a) * 3;
      out[0] = (double)p[0];
      out[1] = (double)p[1];
      out[2] = (double)p[2];
    } else {
      out[0] = out[1] = out[2] = 0.0;
    }
  };
  // reference distances d0 from frame 0
  double d0[4][4];
  {
    double pi[4][3], pj[4][3];
    for (int x = 0; x < 4; ++x) load_atom(0, ti * PT + bi + x, pi[x]);
    for (int y = 0; y < 4; ++y) load_atom(0, tj * PT + bj + y, pj[y]);
    for (int x = 0; x < 4; ++x)
      for (int y = 0; y < 4; ++y) {
        const double dx = pj[y][0] - pi[x][0], dy = pj[y][1] - pi[x][1], dz = pj[y][2] - pi[x][2];
        d0[x][y] = sqrt(dx * dx + dy * dy + dz * dz);
      }
  }
  double s1[4][4], s2[4][4];
  for (int x = 0; x < 4; ++x)
    for (int y = 0; y < 4; ++y) s1[x][y] = s2[x][y] = 0.0;

  for (int64_t t0 = t_begin; t0 < t_end; t0 += PFB) {
    __syncthreads();
    for (int e = tid; e < 2 * PFB * PT * 3; e += 256) {
      const int side = e / (PFB * PT * 3), r = e - side * (PFB * PT * 3);
      const int f = r / (PT * 3), q = r - f * (PT * 3);
      const int a = (side ? tj : ti) * PT + q / 3;
      const int64_t t = t0 + f;
      double v = 0.0;
      if (t < t_end && a < N) v = (double)X[(t * N + a) * 3 + q % 3];
      (&sp[side][f][0][0])[q] = v;
    }
    __syncthreads();
    const int nf = (int)((t_end - t0) < PFB ? (t_end - t0) : PFB);
    for (int f = 0; f < nf; ++f) {
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const double ax = sp[0][f][bi + x][0], ay = sp[0][f][bi + x][1], az = sp[0][f][bi + x][2];
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          const double dx = sp[1][f][bj + y][0] - ax, dy = sp[1][f][bj + y][1] - ay, dz = sp[1][f][bj + y][2] - az;
          const double dd = sqrt(dx * dx + dy * dy + dz * dz) - d0[x][y];
          s1[x][y] += dd;
          s2[x][y] += dd * dd;
        }
      }
    }
  }
  const int ksplit = gridDim.x / n_tiles;
  double* slab = slabs + ((int64_t)tile_lin * ksplit + ks) * (2 * PT * PT);
  for (int x = 0; x < 4; ++x)
    for (int y = 0; y < 4; ++y) {
      slab[(bi + x) * PT + bj + y] = s1[x][y];
      slab[PT * PT + (bi + x) * PT + bj + y] = s2[x][y];
    }
}

// var[i,j] (N x N, symmetric, diagonal 0) from the slabs, fixed summation order
// mean (optional): mean distance over the frames = d0 + shifted mean, d0 recomputed from frame 0 of X
template <typename TIn>
__global__ __launch_bounds__(256) void pair_var_kernel(const double* __restrict__ slabs, int32_t nt1,
                                                       int32_t ksplit, int32_t N, int64_t T,
                                                       double* __restrict__ var, const TIn* __restrict__ X,
                                                       double* __restrict__ mean) {
  int tile = blockIdx.x;
  const int tile_lin = tile;
  int ti = 0;
  {
    int rowlen = nt1;
    while (tile >= rowlen) {
      tile -= rowlen;
      --rowlen;
      ++ti;
    }
  }
  const int tj = ti + tile;
  const double* base = slabs + (int64_t)tile_lin * ksplit * (2 * PT * PT);
  for (int e = threadIdx.x; e < PT * PT; e += 256) {
    const int i = ti * PT + e / PT, j = tj * PT + e % PT;
    if (i >= N || j >= N) continue;
    double a = 0.0, q = 0.0;
    for (int ks = 0; ks < ksplit; ++ks) {
      a += base[(int64_t)ks * (2 * PT * PT) + e];
      q += base[(int64_t)ks * (2 * PT * PT) + PT * PT + e];
    }
    const double m = a / (double)T;
    double v = q / (double)T - m * m;
    if (v < 0.0) v = 0.0;
    if (i == j) v = 0.0;
    var[(int64_t)i * N + j] = v;
    var[(int64_t)j * N + i] = v;
    if (mean) {
      const double dx = (double)X[(int64_t)j * 3 + 0] - (double)X[(int64_t)i * 3 + 0],
                   dy = (double)X[(int64_t)j * 3 + 1] - (double)X[(int64_t)i * 3 + 1],
                   dz = (double)X[(int64_t)j * 3 + 2] - (double)X[(int64_t)i * 3 + 2];
      const double mu = i == j ? 0.0 : sqrt(dx * dx + dy * dy + dz * dz) + m;
      mean[(int64_t)i * N + j] = mu;
      mean[(int64_t)j * N + i] = mu;
    }
  }
}

// out = weight * (var_r + (mean_r - mean)^2): this rank's term of the pooled variance
__global__ __launch_bounds__(256) void pair_pool_kernel(const double* __restrict__ var_r, const double* __restrict__ mean_r,
                                                        const double* __restrict__ mean, double weight, int64_t n,
                                                        double* __restrict__ out) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const double d = mean_r[e] - mean[e];
    out[e] = weight * (var_r[e] + d * d);
  }
}

static void pair_plan(int64_t T, int32_t N, int* nt1, int* n_tiles, int* ksplit, int64_t* fps) {
  *nt1 = (int)ceil_div(N, PT);
  *n_tiles = *nt1 * (*nt1 + 1) / 2;
  int64_t k = ceil_div(2048, *n_tiles);
  const int64_t kmax_frames = ceil_div(T, PFB);
  if (k > kmax_frames) k = kmax_frames;
  const int64_t kmax_mem = (int64_t)(((size_t)1 << 31) / ((size_t)*n_tiles * 2 * PT * PT * 8));
  if (k > kmax_mem) k = kmax_mem;
  if (k < 1) k = 1;
  *ksplit = (int)k;
  *fps = round_up(ceil_div(T, k), PFB);
}

}  // namespace aggf

using namespace aggf;

extern "C" size_t aggf_pair_dist_var_workspace_bytes(int64_t T, int32_t N) {
  if (T <= 0 || N <= 0) return 0;
  int nt1, n_tiles, ksplit;
  int64_t fps;
  pair_plan(T, N, &nt1, &n_tiles, &ksplit, &fps);
  return (size_t)n_tiles * ksplit * 2 * PT * PT * sizeof(double) + 256;
}

static int pair_moments_impl(const void* X, int64_t T, int32_t N, int dtype, double* mean, double* var, void* ws,
                             size_t ws_bytes, void* stream_v, const char* who) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!X || !var || !ws) return fail(AGGF_ERR_ARG, "%s: NULL pointer", who);
  if (T <= 0 || N <= 0) return fail(AGGF_ERR_ARG, "%s: empty problem", who);
  int nt1, n_tiles, ksplit;
  int64_t fps;
  pair_plan(T, N, &nt1, &n_tiles, &ksplit, &fps);
  if (ws_bytes < (size_t)n_tiles * ksplit * 2 * PT * PT * sizeof(double))
    return fail(AGGF_ERR_WORKSPACE, "%s: workspace too small", who);
  double* slabs = reinterpret_cast<double*>(ws);
  const dim3 grid((unsigned)((int64_t)n_tiles * ksplit));
  if (dtype == AGGF_F64) {
    AGGF_LAUNCH(pair_stats_kernel<double>, grid, dim3(256), 0, stream, (const double*)X, T, N, nt1, n_tiles, fps, slabs);
    AGGF_LAUNCH_OK();
    AGGF_LAUNCH(pair_var_kernel<double>, dim3(n_tiles), dim3(256), 0, stream, slabs, nt1, ksplit, N, T, var,
                       (const double*)X, mean);
  } else if (dtype == AGGF_F32) {
    AGGF_LAUNCH(pair_stats_kernel<float>, grid, dim3(256), 0, stream, (const float*)X, T, N, nt1, n_tiles, fps, slabs);
    AGGF_LAUNCH_OK();
    AGGF_LAUNCH(pair_var_kernel<float>, dim3(n_tiles), dim3(256), 0, stream, slabs, nt1, ksplit, N, T, var,
                       (const float*)X, mean);
  } else {
    return fail(AGGF_ERR_ARG, "%s: bad dtype", who);
  }
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_pair_dist_var(const void* X, int64_t T, int32_t N, int dtype, double* var, void* ws,
                                  size_t ws_bytes, void* stream_v) {
  return pair_moments_impl(X, T, N, dtype, nullptr, var, ws, ws_bytes, stream_v, "aggf_pair_dist_var");
}

extern "C" int aggf_pair_dist_moments(const void* X, int64_t T, int32_t N, int dtype, double* mean, double* var,
                                      void* ws, size_t ws_bytes, void* stream_v) {
  if (!mean) return fail(AGGF_ERR_ARG, "aggf_pair_dist_moments: NULL pointer");
  return pair_moments_impl(X, T, N, dtype, mean, var, ws, ws_bytes, stream_v, "aggf_pair_dist_moments");
}

extern "C" int aggf_pair_pool_term(const double* var_r, const double* mean_r, const double* mean, double weight,
                                   int64_t n, double* out, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!var_r || !mean_r || !mean || !out) return fail(AGGF_ERR_ARG, "aggf_pair_pool_term: NULL pointer");
  if (n <= 0) return AGGF_OK;
  int64_t g = ceil_div(n, 256);
  if (g > 4096) g = 4096;
  AGGF_LAUNCH(pair_pool_kernel, dim3((unsigned)g), dim3(256), 0, stream, var_r, mean_r, mean, weight, n, out);
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}
