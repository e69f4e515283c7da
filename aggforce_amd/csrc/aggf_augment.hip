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
  hipLaunchKernelGGL((augment_kernel<TIn, TA, TOut>), dim3((unsigned)nblocks), dim3(256), lds, stream,
                     (const TIn*)coords, (const TIn*)forces, T, N, mt_ptr, mt_idx, (const TA*)mt_val, n_cg,
                     (const TA*)mean, (const TA*)noise, seed, frame_offset, (TA)var, (TA)kbt, fb,
                     (TOut*)out_coords, (TOut*)out_forces);
  AGGF_LAUNCH_OK();
  return AGGF_OK;
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
