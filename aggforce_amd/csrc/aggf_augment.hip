// K5: Gaussian conditional-normal augmentation of a trajectory (noised maps).
//
// Replaces JCondNormal.sample / .log_gradient (trajectory/jaxgausstraj.py:213-284; the
// JAX autodiff of the Gaussian log-density is replaced by its closed form) and
// AugmentedTrajectory._augment (trajectory/core.py:382-390):
//     y      = M x + sqrt(var) * eps                      (n_cg x 3 per frame)
//     r      = (y - M x) / var
//     F_aug  = -kbt * r                                   forces on the noise sites
//     F_real = F + kbt * M' r                             corrected forces on real sites
// and writes the concatenated (T, N + n_cg, 3) coordinate and force arrays in one pass.
// `mean` = M x is produced beforehand by the K3 apply / gather kernels.
#include "aggf_common.h"

namespace aggf {

template <typename TIn, typename TA, typename TOut>
__global__ __launch_bounds__(256) void augment_kernel(
    const TIn* __restrict__ coords, const TIn* __restrict__ forces, int64_t T, int32_t N,
    const int32_t* __restrict__ mt_ptr, const int32_t* __restrict__ mt_idx, const TA* __restrict__ mt_val,
    int32_t n_cg, const TA* __restrict__ mean,
    const TA* __restrict__ noise, uint64_t seed, int64_t frame_offset, TA var, TA kbt, int fb,
    TOut* __restrict__ out_coords, TOut* __restrict__ out_forces) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  TA* sR = reinterpret_cast<TA*>(smem_raw);  // [fb][n_cg][3]
  const int64_t t0 = (int64_t)blockIdx.x * fb;
  const int nf = (int)((T - t0) < fb ? (T - t0) : fb);
  const int row_aug = n_cg * 3;
  const int64_t row_out = (int64_t)(N + n_cg) * 3;
  const TA sd = (TA)sqrt((double)var);
  for (int e = threadIdx.x; e < nf * row_aug; e += blockDim.x) {
    const int f = e / row_aug, cd = e - f * row_aug;
    const int64_t t = t0 + f;
    const TA mu = mean[t * row_aug + cd];
    TA eps;
    if (noise) {
      eps = noise[t * row_aug + cd];
    } else {
      const int64_t g = (frame_offset + t) * row_aug + cd;
      double z[4];
      normal_quad(seed, 1, g >> 2, z);
      eps = (TA)z[g & 3];
    }
    const TA y = mu + sd * eps;
    const TA r = (y - mu) / var;
    sR[e] = r;
    out_coords[t * row_out + (int64_t)N * 3 + cd] = (TOut)y;
    out_forces[t * row_out + (int64_t)N * 3 + cd] = (TOut)(kbt * (-r));
  }
  __syncthreads();
  for (int a = threadIdx.x; a < N; a += blockDim.x) {
    TA acc[4][3];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int d = 0; d < 3; ++d) acc[f][d] = 0;
    // column a of M in compressed form (M' r touches only the sites atom a contributes to)
    for (int j = mt_ptr[a]; j < mt_ptr[a + 1]; ++j) {
      const int c = mt_idx[j];
      const TA m = mt_val[j];
#pragma unroll
      for (int f = 0; f < 4; ++f)
        if (f < nf) {
#pragma unroll
          for (int d = 0; d < 3; ++d) acc[f][d] += m * sR[(f * n_cg + c) * 3 + d];
        }
    }
#pragma unroll
    for (int f = 0; f < 4; ++f)
      if (f < nf) {
        const int64_t t = t0 + f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const int64_t i = (t * N + a) * 3 + d;
          const int64_t o = t * row_out + (int64_t)a * 3 + d;
          out_coords[o] = (TOut)coords[i];
          out_forces[o] = (TOut)forces[i] + (TOut)(kbt * acc[f][d]);
        }
      }
  }
}

// The generated sites alone -- y and their forces -kbt r, (T, n_cg, 3) each -- by the SAME expressions as
// augment_kernel's first loop (same Philox stream): the pieces of the extended trajectory that are not copies.
// With them the noised maps never materialise the (T, N + n_cg, 3) arrays (aggf_gram_pair, aggf_augmented_gram).
template <typename TA, typename TOut>
__global__ __launch_bounds__(256) void noise_sites_kernel(const TA* __restrict__ mean, const TA* __restrict__ noise,
                                                          uint64_t seed, int64_t frame_offset, int64_t T, int32_t n_cg,
                                                          TA var, TA kbt, TOut* __restrict__ out_y,
                                                          TOut* __restrict__ out_f) {
  const int row_aug = n_cg * 3;
  const int64_t total = T * row_aug;
  const TA sd = (TA)sqrt((double)var);
  // One thread = one Philox quad = four consecutive elements of the (global) noise stream: with one element per
  // thread every thread generated a whole quad and used a quarter of it (2.5 ms per call at BASELINE config 4's size).
  const int64_t g0 = frame_offset * row_aug;            // global stream index of element 0 of this shard
  const int64_t q_first = g0 >> 2, q_last = (g0 + total - 1) >> 2;
  for (int64_t q = q_first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q <= q_last; q += (int64_t)gridDim.x * blockDim.x) {
    double z[4] = {0.0, 0.0, 0.0, 0.0};
    if (!noise) normal_quad(seed, 1, q, z);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = (q << 2) + k - g0;              // element of this shard
      if (i < 0 || i >= total) continue;
      const TA mu = mean[i];
      const TA eps = noise ? noise[i] : (TA)z[k];
      const TA y = mu + sd * eps;
      const TA r = (y - mu) / var;
      out_y[i] = (TOut)y;
      out_f[i] = (TOut)(kbt * (-r));
    }
  }
}

// G_aug = Tm' Gx Tm with Tm = [[I, 0], [-C, I]]: the Gram matrix of the extended trajectory
// [F - Fa C | Fa] from the Gram matrix Gx of [F | Fa] (n = N + n2 columns).  C (n2 x N) arrives as compressed columns
// (premap_columns: for atom a the entries (c, C[c,a])).  Two kernels: H = Gyy C (n2 x N), then the upper triangle of
// G_aug, mirrored on store so that the result is exactly symmetric.
__global__ __launch_bounds__(256) void auggram_h_kernel(const double* __restrict__ Gx, int32_t N, int32_t n2,
                                                        const int32_t* __restrict__ cp, const int32_t* __restrict__ ci,
                                                        const double* __restrict__ cv, double* __restrict__ H) {
  const int64_t n = (int64_t)N + n2;
  const int64_t total = (int64_t)n2 * N;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e / N), j = (int)(e - (int64_t)c * N);
    double h = 0.0;
    for (int k = cp[j]; k < cp[j + 1]; ++k) h += Gx[(N + c) * n + N + ci[k]] * cv[k];
    H[e] = h;
  }
}

__global__ __launch_bounds__(256) void auggram_kernel(const double* __restrict__ Gx, int32_t N, int32_t n2,
                                                      const int32_t* __restrict__ cp, const int32_t* __restrict__ ci,
                                                      const double* __restrict__ cv, const double* __restrict__ H,
                                                      double* __restrict__ out) {
  const int64_t n = (int64_t)N + n2;
  const int64_t total = n * n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / n, j = e - i * n;
    if (j < i) continue;  // upper triangle; mirrored below
    double v;
    if (i >= N) {
      v = Gx[e];  // Gyy
    } else if (j >= N) {
      // Gxy[i,c] - sum_{c' in col(i)} C[c',i] Gyy[c',c]
      const int c = (int)(j - N);
      v = Gx[e];
      for (int k = cp[i]; k < cp[i + 1]; ++k) v -= cv[k] * Gx[(N + ci[k]) * n + N + c];
    } else {
      // Gxx[i,j] - sum_{col(i)} C[c,i] (Gyx[c,j] - H[c,j]) - sum_{col(j)} Gxy[i,c] C[c,j]
      v = Gx[e];
      for (int k = cp[i]; k < cp[i + 1]; ++k) v -= cv[k] * (Gx[(N + ci[k]) * n + j] - H[(int64_t)ci[k] * N + j]);
      for (int k = cp[j]; k < cp[j + 1]; ++k) v -= Gx[i * n + N + ci[k]] * cv[k];
    }
    out[e] = v;
    if (j > i) out[j * n + i] = v;
  }
}

// G_red[gi, gj] = sum_{a in gi} sum_{b in gj} G[a, b]  (C' G C for the 0/1 constraint matrix C of qplinear.py:147-164),
// upper triangle computed and mirrored
__global__ __launch_bounds__(256) void sym_group_reduce_kernel(const double* __restrict__ G, int32_t n,
                                                               const int32_t* __restrict__ gp,
                                                               const int32_t* __restrict__ ga, int32_t n_red,
                                                               double* __restrict__ out) {
  const int64_t total = (int64_t)n_red * n_red;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int gi = (int)(e / n_red), gj = (int)(e - (int64_t)gi * n_red);
    if (gj < gi) continue;
    double v = 0.0;
    for (int a = gp[gi]; a < gp[gi + 1]; ++a) {
      const double* row = G + (int64_t)ga[a] * n;
      for (int b = gp[gj]; b < gp[gj + 1]; ++b) v += row[ga[b]];
    }
    out[e] = v;
    if (gj > gi) out[(int64_t)gj * n_red + gi] = v;
  }
}

template <typename TIn, typename TA, typename TOut>
static int augment_typed(const void* coords, const void* forces, int64_t T, int32_t N,
                         const int32_t* mt_ptr, const int32_t* mt_idx, const void* mt_val,
                         int32_t n_cg, const void* mean, const void* noise, uint64_t seed,
                         int64_t frame_offset, double var, double kbt, void* out_coords,
                         void* out_forces, hipStream_t stream) {
  int fb = 4;
  while (fb > 1 && (size_t)fb * n_cg * 3 * sizeof(TA) > 60000) fb >>= 1;
  const size_t lds = (size_t)fb * n_cg * 3 * sizeof(TA);
  if (lds > 64000) return fail(AGGF_ERR_ARG, "aggf_condnormal_augment: n_cg too large");
  const int64_t nblocks = ceil_div(T, fb);
  if (nblocks > 0x7fffffffLL) return fail(AGGF_ERR_ARG, "augment grid too large");
  AGGF_LAUNCH((augment_kernel<TIn, TA, TOut>), dim3((unsigned)nblocks), dim3(256), lds, stream,
                     (const TIn*)coords, (const TIn*)forces, T, N, mt_ptr, mt_idx, (const TA*)mt_val, n_cg,
                     (const TA*)mean, (const TA*)noise, seed, frame_offset, (TA)var, (TA)kbt, fb,
                     (TOut*)out_coords, (TOut*)out_forces);
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}


// ---- pieces of the GENERAL Augmenter protocol (trajectory/core.py:382-390 with any sample / log_gradient) ----------
// r = (gen - mean) / var and -r in one pass: the two log-gradients of a scalar-covariance conditional normal
// (jaxgausstraj.py:251-284 in closed form; simplegausstraj.py:100-113).  Either output may be NULL.
template <typename TG, typename TM, typename TO>
__global__ __launch_bounds__(256) void residual_over_var_kernel(const TG* __restrict__ gen, const TM* __restrict__ mean,
                                                                int64_t count, TO var, TO* __restrict__ out_pos,
                                                                TO* __restrict__ out_neg) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const TO r = ((TO)gen[i] - (TO)mean[i]) / var;
    if (out_pos) out_pos[i] = r;
    if (out_neg) out_neg[i] = -r;
  }
}

// out[t, j] = add[t, j] + alpha * sum_k (X[t, k] - S[t, k]) B[j, k]    (add, S optional)
// The flattened (n_frames, 3 n) products of a FULL covariance matrix: y = mean + eps L' and Sigma^-1 (y - mean)
// (jaxgausstraj.py:77-96, 291-329).  MFMA 16x16x4: workgroup = 64 frames x 64 outputs, 4 waves, wave = 16 frames x 64
// outputs; K in chunks of 32 staged through LDS (coalesced along k, zero filled outside the matrix).
template <typename T>
__global__ __launch_bounds__(256) void frames_matmul_kernel(const T* __restrict__ X, const T* __restrict__ S, int64_t nT,
                                                            int32_t K, const T* __restrict__ B, int32_t J,
                                                            const T* __restrict__ add, T alpha, T* __restrict__ out) {
  constexpr int KC = 32, LD = KC + 1;
  __shared__ T sX[64 * LD];
  __shared__ T sB[64 * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t t0 = (int64_t)blockIdx.x * 64;
  const int j0 = blockIdx.y * 64;
  typename Mfma<T>::acc_t acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = acc_zero<T>();
  for (int k0 = 0; k0 < K; k0 += KC) {
    for (int e = tid; e < 64 * KC; e += 256) {
      const int r = e / KC, k = e - r * KC;
      const int64_t t = t0 + r;
      T x = 0, b = 0;
      if (k0 + k < K) {
        if (t < nT) {
          x = X[t * K + k0 + k];
          if (S) x -= S[t * K + k0 + k];
        }
        if (j0 + r < J) b = B[(int64_t)(j0 + r) * K + k0 + k];
      }
      sX[r * LD + k] = x;
      sB[r * LD + k] = b;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KC; kk += 4) {
      const T a = sX[(wave * 16 + (lane & 15)) * LD + kk + (lane >> 4)];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const T b = sB[(c * 16 + (lane & 15)) * LD + kk + (lane >> 4)];
        acc[c] = Mfma<T>::mma(a, b, acc[c]);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t t = t0 + wave * 16 + Mfma<T>::row(lane, r);
      const int j = j0 + c * 16 + (lane & 15);
      if (t < nT && j < J) {
        const T base = add ? add[t * J + j] : (T)0;
        out[t * J + j] = base + alpha * acc[c][r];
      }
    }
}

// [x ; y] and [F + kbt corr ; kbt lgrad] written side by side (trajectory/core.py:384-390): the concatenation of the
// general Augmenter protocol, with the two scaled sums fused into the copy.  One thread = one output element.
template <typename TX, typename TY, typename TO>
__global__ __launch_bounds__(256) void augment_concat_kernel(const TX* __restrict__ coords, const TX* __restrict__ forces,
                                                             const TY* __restrict__ gen, const TY* __restrict__ corr,
                                                             const TY* __restrict__ lgrad, int64_t nT, int32_t N,
                                                             int32_t n_aug, TO kbt, TO* __restrict__ out_c,
                                                             TO* __restrict__ out_f) {
  const int64_t row_in = (int64_t)N * 3, row_aug = (int64_t)n_aug * 3, row_out = row_in + row_aug;
  const int64_t total = nT * row_out;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = e / row_out, c = e - t * row_out;
    if (c < row_in) {
      const int64_t i = t * row_in + c;
      out_c[e] = (TO)coords[i];
      out_f[e] = (TO)forces[i] + kbt * (TO)corr[i];
    } else {
      const int64_t i = t * row_aug + (c - row_in);
      out_c[e] = (TO)gen[i];
      out_f[e] = kbt * (TO)lgrad[i];
    }
  }
}

}  // namespace aggf

using namespace aggf;

extern "C" int aggf_condnormal_augment(const void* coords, const void* forces, int64_t T, int32_t N,
                                       int traj_dtype, const int32_t* mt_ptr, const int32_t* mt_idx,
                                       const void* mt_val, int32_t n_cg, int aug_dtype,
                                       const void* mean, const void* noise, uint64_t seed,
                                       int64_t frame_offset, double var, double kbt,
                                       void* out_coords, void* out_forces, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!coords || !forces || !mt_ptr || !mt_idx || !mt_val || !mean || !out_coords || !out_forces)
    return fail(AGGF_ERR_ARG, "aggf_condnormal_augment: NULL pointer");
  if (T <= 0 || N <= 0 || n_cg <= 0) return fail(AGGF_ERR_ARG, "aggf_condnormal_augment: empty problem");
  if (!(var > 0.0)) return fail(AGGF_ERR_ARG, "aggf_condnormal_augment: var must be positive");
  if (traj_dtype == AGGF_F32 && aug_dtype == AGGF_F32)
    return augment_typed<float, float, float>(coords, forces, T, N, mt_ptr, mt_idx, mt_val, n_cg, mean, noise, seed, frame_offset, var, kbt, out_coords, out_forces, stream);
  if (traj_dtype == AGGF_F64 && aug_dtype == AGGF_F32)
    return augment_typed<double, float, double>(coords, forces, T, N, mt_ptr, mt_idx, mt_val, n_cg, mean, noise, seed, frame_offset, var, kbt, out_coords, out_forces, stream);
  if (traj_dtype == AGGF_F64 && aug_dtype == AGGF_F64)
    return augment_typed<double, double, double>(coords, forces, T, N, mt_ptr, mt_idx, mt_val, n_cg, mean, noise, seed, frame_offset, var, kbt, out_coords, out_forces, stream);
  if (traj_dtype == AGGF_F32 && aug_dtype == AGGF_F64)
    return augment_typed<float, double, double>(coords, forces, T, N, mt_ptr, mt_idx, mt_val, n_cg, mean, noise, seed, frame_offset, var, kbt, out_coords, out_forces, stream);
  return fail(AGGF_ERR_ARG, "aggf_condnormal_augment: bad dtype");
}

extern "C" int aggf_condnormal_sites(const void* mean, const void* noise, uint64_t seed, int64_t frame_offset,
                                     int64_t T, int32_t n_cg, int aug_dtype, double var, double kbt, void* out_y,
                                     void* out_f, int out_dtype, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!mean || !out_y || !out_f) return fail(AGGF_ERR_ARG, "aggf_condnormal_sites: NULL pointer");
  if (T <= 0 || n_cg <= 0) return fail(AGGF_ERR_ARG, "aggf_condnormal_sites: empty problem");
  if (!(var > 0.0)) return fail(AGGF_ERR_ARG, "aggf_condnormal_sites: var must be positive");
  int64_t g = ceil_div(ceil_div(T * n_cg * 3, 4) + 1, 256);
  if (g > 16384) g = 16384;
  const dim3 grid((unsigned)g), block(256);
  if (aug_dtype == AGGF_F32 && out_dtype == AGGF_F32)
    AGGF_LAUNCH((noise_sites_kernel<float, float>), grid, block, 0, stream, (const float*)mean, (const float*)noise, seed, frame_offset, T, n_cg, (float)var, (float)kbt, (float*)out_y, (float*)out_f);
  else if (aug_dtype == AGGF_F32 && out_dtype == AGGF_F64)
    AGGF_LAUNCH((noise_sites_kernel<float, double>), grid, block, 0, stream, (const float*)mean, (const float*)noise, seed, frame_offset, T, n_cg, (float)var, (float)kbt, (double*)out_y, (double*)out_f);
  else if (aug_dtype == AGGF_F64 && out_dtype == AGGF_F64)
    AGGF_LAUNCH((noise_sites_kernel<double, double>), grid, block, 0, stream, (const double*)mean, (const double*)noise, seed, frame_offset, T, n_cg, var, kbt, (double*)out_y, (double*)out_f);
  else
    return fail(AGGF_ERR_ARG, "aggf_condnormal_sites: bad dtype (out must be the augmenter's dtype or float64)");
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" size_t aggf_augmented_gram_workspace_bytes(int32_t N, int32_t n2) {
  return (N > 0 && n2 > 0) ? (size_t)round_up((int64_t)N * n2 * 8, 256) : 0;
}

extern "C" int aggf_augmented_gram(const double* Gx, int32_t N, int32_t n2, const int32_t* c_ptr, const int32_t* c_idx,
                                   const double* c_val, double* G_aug, void* ws, size_t ws_bytes, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!Gx || !c_ptr || !c_idx || !c_val || !G_aug || !ws) return fail(AGGF_ERR_ARG, "aggf_augmented_gram: NULL pointer");
  if (N <= 0 || n2 <= 0) return fail(AGGF_ERR_ARG, "aggf_augmented_gram: empty problem");
  if (Gx == G_aug) return fail(AGGF_ERR_ARG, "aggf_augmented_gram: in-place transform is not supported");
  if (ws_bytes < aggf_augmented_gram_workspace_bytes(N, n2)) return fail(AGGF_ERR_WORKSPACE, "aggf_augmented_gram: workspace too small");
  double* H = reinterpret_cast<double*>(ws);
  int64_t g1 = ceil_div((int64_t)N * n2, 256), g2 = ceil_div(((int64_t)N + n2) * ((int64_t)N + n2), 256);
  if (g1 > 65535) g1 = 65535;
  if (g2 > 65535) g2 = 65535;
  AGGF_LAUNCH(auggram_h_kernel, dim3((unsigned)g1), dim3(256), 0, stream, Gx, N, n2, c_ptr, c_idx, c_val, H);
  AGGF_LAUNCH_OK();
  AGGF_LAUNCH(auggram_kernel, dim3((unsigned)g2), dim3(256), 0, stream, Gx, N, n2, c_ptr, c_idx, c_val, H, G_aug);
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_sym_group_reduce(const double* G, int32_t n, const int32_t* grp_ptr, const int32_t* grp_atoms,
                                     int32_t n_red, double* G_red, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!G || !grp_ptr || !grp_atoms || !G_red) return fail(AGGF_ERR_ARG, "aggf_sym_group_reduce: NULL pointer");
  if (n <= 0 || n_red <= 0 || n_red > n) return fail(AGGF_ERR_ARG, "aggf_sym_group_reduce: bad shape");
  if (G == G_red) return fail(AGGF_ERR_ARG, "aggf_sym_group_reduce: in-place reduction is not supported");
  int64_t g = ceil_div((int64_t)n_red * n_red, 256);
  if (g > 65535) g = 65535;
  AGGF_LAUNCH(sym_group_reduce_kernel, dim3((unsigned)g), dim3(256), 0, stream, G, n, grp_ptr, grp_atoms, n_red, G_red);
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_residual_over_var(const void* gen, int gen_dtype, const void* mean, int mean_dtype, int64_t count,
                                      double var, void* out_pos, void* out_neg, int out_dtype, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!gen || !mean || (!out_pos && !out_neg)) return fail(AGGF_ERR_ARG, "aggf_residual_over_var: NULL pointer");
  if (count < 0) return fail(AGGF_ERR_ARG, "aggf_residual_over_var: negative count");
  if (!(var > 0.0)) return fail(AGGF_ERR_ARG, "aggf_residual_over_var: var must be positive");
  if (count == 0) return AGGF_OK;
  int64_t g = ceil_div(count, 256);
  if (g > 16384) g = 16384;
  const dim3 grid((unsigned)g), block(256);
#define AGGF_ROV(TG, TM, TO)                                                                                         \
  AGGF_LAUNCH((residual_over_var_kernel<TG, TM, TO>), grid, block, 0, stream, (const TG*)gen, (const TM*)mean, \
                     count, (TO)var, (TO*)out_pos, (TO*)out_neg)
  if (gen_dtype == AGGF_F32 && mean_dtype == AGGF_F32 && out_dtype == AGGF_F32) AGGF_ROV(float, float, float);
  else if (gen_dtype == AGGF_F32 && mean_dtype == AGGF_F32 && out_dtype == AGGF_F64) AGGF_ROV(float, float, double);
  else if (gen_dtype == AGGF_F64 && mean_dtype == AGGF_F64 && out_dtype == AGGF_F64) AGGF_ROV(double, double, double);
  else if (gen_dtype == AGGF_F32 && mean_dtype == AGGF_F64 && out_dtype == AGGF_F64) AGGF_ROV(float, double, double);
  else if (gen_dtype == AGGF_F64 && mean_dtype == AGGF_F32 && out_dtype == AGGF_F64) AGGF_ROV(double, float, double);
  else return fail(AGGF_ERR_ARG, "aggf_residual_over_var: bad dtype (out must hold the promotion of the inputs)");
#undef AGGF_ROV
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_frames_matmul(const void* X, const void* S, int64_t T, int32_t K, const void* B, int32_t J,
                                  const void* add, double alpha, int dtype, void* out, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!X || !B || !out) return fail(AGGF_ERR_ARG, "aggf_frames_matmul: NULL pointer");
  if (T < 0 || K <= 0 || J <= 0) return fail(AGGF_ERR_ARG, "aggf_frames_matmul: bad shape");
  if (out == X || out == S) return fail(AGGF_ERR_ARG, "aggf_frames_matmul: out must not alias X or S");
  if (T == 0) return AGGF_OK;
  const int64_t gx = ceil_div(T, 64), gy = ceil_div(J, 64);
  if (gx > 0x7fffffffLL || gy > 65535) return fail(AGGF_ERR_ARG, "aggf_frames_matmul: grid too large");
  const dim3 grid((unsigned)gx, (unsigned)gy), block(256);
  if (dtype == AGGF_F32)
    AGGF_LAUNCH((frames_matmul_kernel<float>), grid, block, 0, stream, (const float*)X, (const float*)S, T, K,
                       (const float*)B, J, (const float*)add, (float)alpha, (float*)out);
  else if (dtype == AGGF_F64)
    AGGF_LAUNCH((frames_matmul_kernel<double>), grid, block, 0, stream, (const double*)X, (const double*)S, T, K,
                       (const double*)B, J, (const double*)add, alpha, (double*)out);
  else
    return fail(AGGF_ERR_ARG, "aggf_frames_matmul: bad dtype");
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_augment_concat(const void* coords, const void* forces, int traj_dtype, const void* gen,
                                   const void* corr, const void* lgrad, int aug_dtype, int64_t T, int32_t N,
                                   int32_t n_aug, double kbt, void* out_coords, void* out_forces, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!coords || !forces || !gen || !corr || !lgrad || !out_coords || !out_forces)
    return fail(AGGF_ERR_ARG, "aggf_augment_concat: NULL pointer");
  if (T < 0 || N <= 0 || n_aug <= 0) return fail(AGGF_ERR_ARG, "aggf_augment_concat: bad shape");
  if (T == 0) return AGGF_OK;
  int64_t g = ceil_div(T * ((int64_t)N + n_aug) * 3, 256);
  if (g > 32768) g = 32768;
  const dim3 grid((unsigned)g), block(256);
#define AGGF_CAT(TX, TY, TO)                                                                                       \
  AGGF_LAUNCH((augment_concat_kernel<TX, TY, TO>), grid, block, 0, stream, (const TX*)coords,               \
                     (const TX*)forces, (const TY*)gen, (const TY*)corr, (const TY*)lgrad, T, N, n_aug, (TO)kbt,   \
                     (TO*)out_coords, (TO*)out_forces)
  if (traj_dtype == AGGF_F32 && aug_dtype == AGGF_F32) AGGF_CAT(float, float, float);
  else if (traj_dtype == AGGF_F64 && aug_dtype == AGGF_F32) AGGF_CAT(double, float, double);
  else if (traj_dtype == AGGF_F32 && aug_dtype == AGGF_F64) AGGF_CAT(float, double, double);
  else if (traj_dtype == AGGF_F64 && aug_dtype == AGGF_F64) AGGF_CAT(double, double, double);
  else return fail(AGGF_ERR_ARG, "aggf_augment_concat: bad dtype");
#undef AGGF_CAT
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}
