// K4: Gaussian-basis distance featuriser (gb_feat) and the featurised regression matrix,
// without ever materialising the one-hot (T, N, n_feat) feature tensor.
//
// Reference: qp/jaxfeat.py (gb_feat 20-184, gaussian_dist_basis 187-240, clipped_gauss
// 243-276, channel_allocate 279-368, gb_subfeat 371-464, gb_subfeat_jac 467-567),
// map/tools.py:63-104 (smear_map), qp/featlinearmap.py:361-369 (regression matrix) and
// 512-520 (map application).  Facts used: every atom of a constraint group is
// replaced by the group mean before the distance is taken, so all atoms of a channel share
// one distance r[t,ch] = |p[t,ch] - cg[t,c]| and one Gaussian row, and
//   feat[t,a,(ch,k)]  = [ch == channel(a)] * g_k(r[t,ch])
//   div [t,(ch,k),:]  = |ch| * g_k'(r[t,ch]) * (p[t,ch] - cg[t,c]) / r[t,ch]        (closed form of
//                       the reference's jacrev; cg is a constant of the differentiation)
//   R3[t,(ch,k),d]    = g_k(r) * Fg[t,ch,d] + kbt * div[t,(ch,k),d],  Fg = group force sums
// with g_k(r) = max(exp(-((r-c_k)/w)^2), clip) - clip.  The float32 arithmetic of the
// reference (JAX default) is kept for positions, distances and Gaussians.
#include "aggf_common.h"

namespace aggf {

// out[t,g,d] = sum (or mean: each term multiplied by 1/|g| first, like the smear matrix
// product) of X[t,a,d] over the atoms of group g
template <typename TIn, typename TO>
__global__ __launch_bounds__(256) void group_reduce_kernel(const TIn* __restrict__ X, int64_t T, int32_t N,
                                                           const int32_t* __restrict__ grp_ptr,
                                                           const int32_t* __restrict__ grp_atoms,
                                                           int32_t G, int mean, TO* __restrict__ out) {
  const int64_t row_in = (int64_t)N * 3, row_out = (int64_t)G * 3;
  for (int64_t t = blockIdx.x; t < T; t += gridDim.x) {
    const TIn* src = X + t * row_in;
    TO* dst = out + t * row_out;
    for (int e = threadIdx.x; e < (int)row_out; e += blockDim.x) {
      const int g = e / 3, d = e - 3 * g;
      const int b = grp_ptr[g], en = grp_ptr[g + 1];
      // mean weights are the float32 entries of the reference's smear matrix (map/tools.py:94-101), whatever TO
      const TO w = mean ? (TO)(1.0f / (float)(en - b)) : (TO)1;
      TO acc = 0;
      for (int j = b; j < en; ++j) acc += w * (TO)src[(int64_t)grp_atoms[j] * 3 + d];
      dst[e] = acc;
    }
  }
}

// TG = arithmetic type of positions, distances and Gaussians: float (the reference's JAX default; the default
// here) or double (opt-in `feature_dtype=np.float64` of gb_feat: product and float64 oracle then share arithmetic)
template <typename TG>
struct GbParams {
  const TG* centers;  // n_basis grid centres
  int32_t n_basis;
  TG width, clip;
};

__device__ __forceinline__ float gb_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double gb_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float gb_exp(float x) { return expf(x); }
__device__ __forceinline__ double gb_exp(double x) { return exp(x); }
__device__ __forceinline__ float gb_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double gb_max(double a, double b) { return fmax(a, b); }

// arithmetic type of the products with the forces: NumPy promotion of (force dtype, feature dtype)
template <typename TF, typename TG>
struct GbProd { typedef double type; };
template <>
struct GbProd<float, float> { typedef float type; };

// distance, unit vector and Gaussian row of channel ch at frame t
template <typename TG>
__device__ __forceinline__ void gb_geometry(const TG* __restrict__ Pg, const TG* __restrict__ cg,
                                            int64_t t, int32_t G, int32_t ch, int32_t n_cg, int32_t site,
                                            TG& r, TG u[3]) {
  // no fused multiply-add here: whether dx*dx + dy*dy + dz*dz contracts is otherwise decided per kernel the function is
  // inlined into, and the kernels that evaluate a feature (regression matrix, constraint rows, the two application
  // kernels) must agree on it to the last bit -- as the reference's NumPy/JAX float32 arithmetic does
#pragma clang fp contract(off)
  const TG* p = Pg + (t * G + ch) * 3;
  const TG* c = cg + (t * n_cg + site) * 3;
  const TG dx = p[0] - c[0], dy = p[1] - c[1], dz = p[2] - c[2];
  r = gb_sqrt(dx * dx + dy * dy + dz * dz);
  u[0] = dx / r;  // NaN at r == 0, like the gradient of jnp.linalg.norm
  u[1] = dy / r;
  u[2] = dz / r;
}

template <typename TG>
__device__ __forceinline__ void gb_gauss(const GbParams<TG>& gp, TG r, int k, TG& g, TG& dg) {
#pragma clang fp contract(off)
  const TG arg = (r - gp.centers[k]) / gp.width;
  const TG raw = gb_exp(-(arg * arg));
  g = gb_max(raw, gp.clip) - gp.clip;
  dg = raw > gp.clip ? (TG)-2 * arg / gp.width * raw : (TG)0;
}

// compact per-channel features: gauss[t,ch,k], grad[t,ch,k,:] = |ch| g_k'(r) u
template <typename TG>
__global__ __launch_bounds__(256) void gb_channels_kernel(const TG* __restrict__ Pg,
                                                          const TG* __restrict__ cg, int64_t T, int32_t G,
                                                          int32_t n_cg, int32_t site,
                                                          const float* __restrict__ sizes, int32_t n_ch,
                                                          GbParams<TG> gp, TG* __restrict__ gauss,
                                                          TG* __restrict__ grad) {
  const int64_t total = T * n_ch;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / n_ch;
    const int ch = (int)(i - t * n_ch);
    TG r, u[3];
    gb_geometry(Pg, cg, t, G, ch, n_cg, site, r, u);
    const TG m = (TG)sizes[ch];
    for (int k = 0; k < gp.n_basis; ++k) {
      TG g, dg;
      gb_gauss(gp, r, k, g, dg);
      const int64_t o = i * gp.n_basis + k;
      gauss[o] = g;
      grad[o * 3 + 0] = m * dg * u[0];
      grad[o * 3 + 1] = m * dg * u[1];
      grad[o * 3 + 2] = m * dg * u[2];
    }
  }
}

// R3 (T, ld_feat, 3): columns [0, n_id) = group force sums (id_feat block, optional), then
// n_ch * n_basis Gaussian columns; columns up to ld_feat are left untouched (ignored by K1).
// TO = storage type of R3: the products are formed in TF (the reference's arithmetic) and only then widened,
// so a float64 R3 -- K1's in-place operand for float64 products -- holds exactly the float32 matrix.
template <typename TF, typename TG, typename TO>
__global__ __launch_bounds__(256) void gb_regmat_kernel(const TF* __restrict__ Fg, const TG* __restrict__ Pg,
                                                        const TG* __restrict__ cg, int64_t T, int32_t G,
                                                        int32_t n_cg, int32_t site,
                                                        const float* __restrict__ sizes, int32_t n_id,
                                                        int32_t n_ch, GbParams<TG> gp, double kbt_d, int32_t ld_feat,
                                                        TO* __restrict__ R3) {
  typedef typename GbProd<TF, TG>::type TP;
  const TP kbt = (TP)kbt_d;
  const int per_frame = n_id + n_ch;  // work items per frame: id columns, then channels
  const int64_t total = T * per_frame;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / per_frame;
    const int j = (int)(i - t * per_frame);
    TO* row = R3 + t * (int64_t)ld_feat * 3;
    if (j < n_id) {
      const TF* f = Fg + (t * G + j) * 3;
      row[j * 3 + 0] = (TO)f[0];
      row[j * 3 + 1] = (TO)f[1];
      row[j * 3 + 2] = (TO)f[2];
      continue;
    }
    const int ch = j - n_id;
    TG r, u[3];
    gb_geometry(Pg, cg, t, G, ch, n_cg, site, r, u);
    const TF* f = Fg + (t * G + ch) * 3;
    const TP f0 = (TP)f[0], f1 = (TP)f[1], f2 = (TP)f[2];
    const TG m = (TG)sizes[ch];
    TO* o = row + ((int64_t)n_id + (int64_t)ch * gp.n_basis) * 3;
    for (int k = 0; k < gp.n_basis; ++k) {
      TG g, dg;
      gb_gauss(gp, r, k, g, dg);
      const TG s = m * dg;
      o[k * 3 + 0] = (TO)((TP)g * f0 + kbt * (TP)(s * u[0]));
      o[k * 3 + 1] = (TO)((TP)g * f1 + kbt * (TP)(s * u[1]));
      o[k * 3 + 2] = (TO)((TP)g * f2 + kbt * (TP)(s * u[2]));
    }
  }
}

// ---------------------------------------------------------------------------
// Column compaction of the fused fit.  A clipped Gaussian column (ch, k) is identically zero over the
// whole trajectory when the channel never comes within w sqrt(ln(1/clip)) of centre c_k (the point of a
// cut-off basis); such a column contributes a zero row/column to P and zeros to every constraint row, so
// its coefficient in the minimiser is exactly 0 (l2 > 0) and it can be left out of the Gram matrix and of
// the solve.  gb_range_kernel finds, per (site, channel), the range of distances over the frames.

__device__ __forceinline__ void atomic_min_pos_float(float* addr, float v) {
  atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));  // order-preserving for v >= 0
}
__device__ __forceinline__ void atomic_max_pos_float(float* addr, float v) {
  atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
}

// rmin/rmax (n_cg, G) must be initialised to +inf / 0.  grid = (channel blocks, sites, frame slices).
// A NaN distance (non-finite coordinates) marks the channel as spanning everything.
__global__ __launch_bounds__(256) void gb_range_kernel(const float* __restrict__ Pg, const float* __restrict__ cg,
                                                       int64_t T, int32_t G, int32_t n_cg, int32_t n_ch,
                                                       float* __restrict__ rmin, float* __restrict__ rmax) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  const int site = blockIdx.y;
  if (ch >= n_ch) return;
  float lo = INFINITY, hi = 0.0f;
  for (int64_t t = blockIdx.z; t < T; t += gridDim.z) {
    const float* p = Pg + (t * G + ch) * 3;
    const float* c = cg + (t * n_cg + site) * 3;
    const float dx = p[0] - c[0], dy = p[1] - c[1], dz = p[2] - c[2];
    const float r = sqrtf(dx * dx + dy * dy + dz * dz);
    if (!(r == r)) { lo = 0.0f; hi = INFINITY; }
    lo = fminf(lo, r);
    hi = fmaxf(hi, r);
  }
  atomic_min_pos_float(rmin + (int64_t)site * G + ch, lo);
  atomic_max_pos_float(rmax + (int64_t)site * G + ch, hi);
}

// Compact regression matrix: column j < n_id = group force sums; column n_id + j = the Gaussian column
// cols[j] = ch * n_basis + k (only the listed ones).  One thread per (frame, compact column): consecutive
// threads write consecutive 3-vectors (coalesced), the distance of a channel is recomputed per listed k.
template <typename TF, typename TG, typename TO>
__global__ __launch_bounds__(256) void gb_regmat_cols_kernel(const TF* __restrict__ Fg, const TG* __restrict__ Pg,
                                                             const TG* __restrict__ cg, int64_t T, int32_t G,
                                                             int32_t n_cg, int32_t site,
                                                             const float* __restrict__ sizes, int32_t n_id,
                                                             const int32_t* __restrict__ cols, int32_t n_cols,
                                                             GbParams<TG> gp, double kbt_d, int32_t ld_feat,
                                                             TO* __restrict__ R3) {
  typedef typename GbProd<TF, TG>::type TP;
  const TP kbt = (TP)kbt_d;
  const int per_frame = n_id + n_cols;
  const int64_t total = T * per_frame;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / per_frame;
    const int j = (int)(i - t * per_frame);
    TO* o = R3 + (t * (int64_t)ld_feat + j) * 3;
    if (j < n_id) {
      const TF* f = Fg + (t * G + j) * 3;
      o[0] = (TO)f[0];
      o[1] = (TO)f[1];
      o[2] = (TO)f[2];
      continue;
    }
    const int full = cols[j - n_id];
    const int ch = full / gp.n_basis, k = full - ch * gp.n_basis;
    TG r, u[3], g, dg;
    gb_geometry(Pg, cg, t, G, ch, n_cg, site, r, u);
    gb_gauss(gp, r, k, g, dg);
    const TF* f = Fg + (t * G + ch) * 3;
    const TG s = (TG)sizes[ch] * dg;
    o[0] = (TO)((TP)g * (TP)f[0] + kbt * (TP)(s * u[0]));
    o[1] = (TO)((TP)g * (TP)f[1] + kbt * (TP)(s * u[1]));
    o[2] = (TO)((TP)g * (TP)f[2] + kbt * (TP)(s * u[2]));
  }
}

// CLAMap application of the [id | gb] feature-linear map (featlinearmap.py:512-520,
// map/core.py:428-430): out[t,c,:] = sum_f coef[c,f] * (feat_c[t]' F[t] + div_c[t])[f,:]
// -- the divergence enters WITHOUT kbt, exactly as in the reference's trans_f.
// One wave per (block of GB_FR frames, site); lanes stride over id columns and channels.  A lane's coefficients of a
// channel are read ONCE for the GB_FR frames (per frame they were 43 KB of L1/L2 reads per (frame, site) at BASELINE
// config 4 -- 55 GB per application, the kernel's bound: 16 ms), and a channel none of whose columns the fit kept
// costs neither the distance nor the expf.  Per frame the terms are added in the same order as one frame at a time.
constexpr int GB_FR = 4;
constexpr int GB_CB = 8;  // coefficients of a channel held in registers (basis functions beyond: read in the loop)

template <typename TF, typename TG>
__global__ __launch_bounds__(256) void gb_apply_kernel(const TF* __restrict__ Fg, const TG* __restrict__ Pg,
                                                       const TG* __restrict__ cg, int64_t T, int32_t G,
                                                       int32_t n_cg, const float* __restrict__ sizes,
                                                       int32_t n_id, int32_t n_ch, GbParams<TG> gp,
                                                       const double* __restrict__ coef, int32_t n_feat,
                                                       double* __restrict__ out) {
  typedef typename GbProd<TF, TG>::type TP;
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t n_blk = (T + GB_FR - 1) / GB_FR;
  for (int64_t i = wid; i < n_blk * n_cg; i += nw) {
    const int64_t blk = i / n_cg;  // consecutive waves: the same frames, the sites one after the other
    const int site = (int)(i - blk * n_cg);
    const int64_t t0 = blk * GB_FR;
    const int nf = T - t0 < GB_FR ? (int)(T - t0) : GB_FR;
    const double* cf = coef + (int64_t)site * n_feat;
    double acc[GB_FR][3];
#pragma unroll
    for (int f = 0; f < GB_FR; ++f) acc[f][0] = acc[f][1] = acc[f][2] = 0.0;
    for (int g = lane; g < n_id; g += 64) {
      const double c = cf[g];
#pragma unroll
      for (int f = 0; f < GB_FR; ++f) {
        if (f < nf) {
          const TF* fv = Fg + ((t0 + f) * G + g) * 3;
          acc[f][0] += c * (double)fv[0];
          acc[f][1] += c * (double)fv[1];
          acc[f][2] += c * (double)fv[2];
        }
      }
    }
    for (int ch = lane; ch < n_ch; ch += 64) {
      const double* cc = cf + n_id + (int64_t)ch * gp.n_basis;
      double cb[GB_CB];  // the loads are issued together: one memory latency per channel, not one per basis function
#pragma unroll
      for (int k = 0; k < GB_CB; ++k) cb[k] = k < gp.n_basis ? cc[k] : 0.0;
      bool any = false;
#pragma unroll
      for (int k = 0; k < GB_CB; ++k) any |= cb[k] != 0.0;
      for (int k = GB_CB; k < gp.n_basis; ++k) any |= cc[k] != 0.0;
      if (!any) continue;  // no column of this channel was kept by the fit (identically zero over the training frames)
      TG r[GB_FR], u[GB_FR][3];
      TP fr[GB_FR][3];
#pragma unroll
      for (int f = 0; f < GB_FR; ++f) {
        if (f < nf) {
          gb_geometry(Pg, cg, t0 + f, G, ch, n_cg, site, r[f], u[f]);
          const TF* fv = Fg + ((t0 + f) * G + ch) * 3;
          fr[f][0] = (TP)fv[0];
          fr[f][1] = (TP)fv[1];
          fr[f][2] = (TP)fv[2];
        }
      }
      const TG m = (TG)sizes[ch];
      auto add_basis = [&](int k, double c) {
        if (c == 0.0) return;  // a column the fit left out: no expf
#pragma unroll
        for (int f = 0; f < GB_FR; ++f) {
          if (f < nf) {
            TG g, dg;
            gb_gauss(gp, r[f], k, g, dg);
            const TG s = m * dg;
            acc[f][0] += c * ((double)((TP)g * fr[f][0]) + (double)(s * u[f][0]));
            acc[f][1] += c * ((double)((TP)g * fr[f][1]) + (double)(s * u[f][1]));
            acc[f][2] += c * ((double)((TP)g * fr[f][2]) + (double)(s * u[f][2]));
          }
        }
      };
#pragma unroll
      for (int k = 0; k < GB_CB; ++k)
        if (k < gp.n_basis) add_basis(k, cb[k]);
      for (int k = GB_CB; k < gp.n_basis; ++k) add_basis(k, cc[k]);
    }
#pragma unroll
    for (int f = 0; f < GB_FR; ++f) {
      double a0 = acc[f][0], a1 = acc[f][1], a2 = acc[f][2];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        a0 += __shfl_down(a0, off, 64);
        a1 += __shfl_down(a1, off, 64);
        a2 += __shfl_down(a2, off, 64);
      }
      if (lane == 0 && f < nf) {
        double* o = out + ((t0 + f) * n_cg + site) * 3;
        o[0] = a0;
        o[1] = a1;
        o[2] = a2;
      }
    }
  }
}

// The same map from a COMPACT coefficient list: the fit keeps ~a third of the Gaussian columns (2.1 of 8 basis
// functions per channel at BASELINE config 4) and different ones for every channel, so in gb_apply_kernel -- lane =
// channel -- nearly every basis function is executed by the wave for the few lanes that need it.  Here lane = one kept
// column (ch, k) of the site: no idle lanes; the distance of a channel is recomputed for each of its kept columns
// (cheaper than the expf it replaces).  col_ptr[n_cg + 1] / col_idx (ch * n_basis + k) / col_val: the non-zero
// Gaussian coefficients per site; coef_id (n_cg, n_id): the id block, dense.
template <typename TF, typename TG>
__global__ __launch_bounds__(256) void gb_apply_cols_kernel(const TF* __restrict__ Fg, const TG* __restrict__ Pg,
                                                            const TG* __restrict__ cg, int64_t T, int32_t G,
                                                            int32_t n_cg, const float* __restrict__ sizes,
                                                            int32_t n_id, const double* __restrict__ coef_id,
                                                            const int32_t* __restrict__ col_ptr,
                                                            const int32_t* __restrict__ col_idx,
                                                            const double* __restrict__ col_val, GbParams<TG> gp,
                                                            double* __restrict__ out) {
  typedef typename GbProd<TF, TG>::type TP;
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t n_blk = (T + GB_FR - 1) / GB_FR;
  for (int64_t i = wid; i < n_blk * n_cg; i += nw) {
    const int64_t blk = i / n_cg;
    const int site = (int)(i - blk * n_cg);
    const int64_t t0 = blk * GB_FR;
    const int nf = T - t0 < GB_FR ? (int)(T - t0) : GB_FR;
    double acc[GB_FR][3];
#pragma unroll
    for (int f = 0; f < GB_FR; ++f) acc[f][0] = acc[f][1] = acc[f][2] = 0.0;
    for (int g = lane; g < n_id; g += 64) {
      const double c = coef_id[(int64_t)site * n_id + g];
#pragma unroll
      for (int f = 0; f < GB_FR; ++f) {
        if (f < nf) {
          const TF* fv = Fg + ((t0 + f) * G + g) * 3;
          acc[f][0] += c * (double)fv[0];
          acc[f][1] += c * (double)fv[1];
          acc[f][2] += c * (double)fv[2];
        }
      }
    }
    for (int e = col_ptr[site] + lane; e < col_ptr[site + 1]; e += 64) {
      const int full = col_idx[e];
      const int ch = full / gp.n_basis, k = full - ch * gp.n_basis;
      const double c = col_val[e];
      const TG m = (TG)sizes[ch];
#pragma unroll
      for (int f = 0; f < GB_FR; ++f) {
        if (f < nf) {
          TG r, u[3], g, dg;
          gb_geometry(Pg, cg, t0 + f, G, ch, n_cg, site, r, u);
          gb_gauss(gp, r, k, g, dg);
          const TF* fv = Fg + ((t0 + f) * G + ch) * 3;
          const TG sd = m * dg;
          acc[f][0] += c * ((double)((TP)g * (TP)fv[0]) + (double)(sd * u[0]));
          acc[f][1] += c * ((double)((TP)g * (TP)fv[1]) + (double)(sd * u[1]));
          acc[f][2] += c * ((double)((TP)g * (TP)fv[2]) + (double)(sd * u[2]));
        }
      }
    }
#pragma unroll
    for (int f = 0; f < GB_FR; ++f) {
      double a0 = acc[f][0], a1 = acc[f][1], a2 = acc[f][2];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        a0 += __shfl_down(a0, off, 64);
        a1 += __shfl_down(a1, off, 64);
        a2 += __shfl_down(a2, off, 64);
      }
      if (lane == 0 && f < nf) {
        double* o = out + ((t0 + f) * n_cg + site) * 3;
        o[0] = a0;
        o[1] = a1;
        o[2] = a2;
      }
    }
  }
}

static inline dim3 feat_grid(int64_t n) {
  int64_t g = ceil_div(n, 256);
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return dim3((unsigned)g);
}

static int check_gb(const void* centers, int32_t n_basis, double width) {
  if (!centers || n_basis <= 0 || n_basis > 64) return fail(AGGF_ERR_ARG, "gb_feat: bad n_basis / centres");
  if (!(width > 0.0)) return fail(AGGF_ERR_ARG, "gb_feat: width must be positive");
  return AGGF_OK;
}

}  // namespace aggf

using namespace aggf;

extern "C" int aggf_group_reduce(const void* X, int64_t T, int32_t N, int in_dtype,
                                 const int32_t* grp_ptr, const int32_t* grp_atoms, int32_t n_groups,
                                 int mean, int out_dtype, void* out, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!X || !grp_ptr || !grp_atoms || !out) return fail(AGGF_ERR_ARG, "aggf_group_reduce: NULL pointer");
  if (T <= 0 || N <= 0 || n_groups <= 0) return fail(AGGF_ERR_ARG, "aggf_group_reduce: empty problem");
  const dim3 grid((unsigned)(T < 8192 ? T : 8192)), block(256);
  if (in_dtype == AGGF_F32 && out_dtype == AGGF_F32)
    AGGF_LAUNCH((group_reduce_kernel<float, float>), grid, block, 0, stream, (const float*)X, T, N, grp_ptr, grp_atoms, n_groups, mean, (float*)out);
  else if (in_dtype == AGGF_F64 && out_dtype == AGGF_F64)
    AGGF_LAUNCH((group_reduce_kernel<double, double>), grid, block, 0, stream, (const double*)X, T, N, grp_ptr, grp_atoms, n_groups, mean, (double*)out);
  else if (in_dtype == AGGF_F64 && out_dtype == AGGF_F32)
    AGGF_LAUNCH((group_reduce_kernel<double, float>), grid, block, 0, stream, (const double*)X, T, N, grp_ptr, grp_atoms, n_groups, mean, (float*)out);
  else if (in_dtype == AGGF_F32 && out_dtype == AGGF_F64)
    AGGF_LAUNCH((group_reduce_kernel<float, double>), grid, block, 0, stream, (const float*)X, T, N, grp_ptr, grp_atoms, n_groups, mean, (double*)out);
  else
    return fail(AGGF_ERR_ARG, "aggf_group_reduce: bad dtype");
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_gb_channels(const void* Pg, const void* cg, int g_dtype, int64_t T, int32_t G, int32_t n_cg,
                                int32_t site, const float* sizes, int32_t n_ch, const void* centers,
                                int32_t n_basis, double width, double clip, void* gauss, void* grad,
                                void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!Pg || !cg || !sizes || !gauss || !grad) return fail(AGGF_ERR_ARG, "aggf_gb_channels: NULL pointer");
  if (T <= 0 || G <= 0 || n_ch <= 0 || n_ch > G || site < 0 || site >= n_cg)
    return fail(AGGF_ERR_ARG, "aggf_gb_channels: bad shape");
  int rc = check_gb(centers, n_basis, width);
  if (rc) return rc;
  if (g_dtype == AGGF_F32) {
    GbParams<float> gp{(const float*)centers, n_basis, (float)width, (float)clip};
    AGGF_LAUNCH(gb_channels_kernel<float>, feat_grid(T * n_ch), dim3(256), 0, stream, (const float*)Pg, (const float*)cg, T, G, n_cg, site, sizes, n_ch, gp, (float*)gauss, (float*)grad);
  } else if (g_dtype == AGGF_F64) {
    GbParams<double> gp{(const double*)centers, n_basis, width, clip};
    AGGF_LAUNCH(gb_channels_kernel<double>, feat_grid(T * n_ch), dim3(256), 0, stream, (const double*)Pg, (const double*)cg, T, G, n_cg, site, sizes, n_ch, gp, (double*)gauss, (double*)grad);
  } else {
    return fail(AGGF_ERR_ARG, "aggf_gb_channels: bad feature dtype");
  }
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

// dispatch over (force dtype, feature dtype, output dtype): out must be the promoted product dtype or float64
#define AGGF_GB_DISPATCH(WHO, LAUNCH)                                                                          \
  do {                                                                                                         \
    if (g_dtype == AGGF_F32) {                                                                                 \
      GbParams<float> gp{(const float*)centers, n_basis, (float)width, (float)clip};                           \
      typedef float TG;                                                                                        \
      if (f_dtype == AGGF_F32 && out_dtype == AGGF_F32) { typedef float TF; typedef float TO __attribute__((unused)); LAUNCH; }        \
      else if (f_dtype == AGGF_F32 && out_dtype == AGGF_F64) { typedef float TF; typedef double TO __attribute__((unused)); LAUNCH; }  \
      else if (f_dtype == AGGF_F64 && out_dtype == AGGF_F64) { typedef double TF; typedef double TO __attribute__((unused)); LAUNCH; } \
      else return fail(AGGF_ERR_ARG, WHO ": bad dtype (out must be the product dtype or float64)");            \
    } else if (g_dtype == AGGF_F64) {                                                                          \
      GbParams<double> gp{(const double*)centers, n_basis, width, clip};                                       \
      typedef double TG;                                                                                       \
      if (f_dtype == AGGF_F32 && out_dtype == AGGF_F64) { typedef float TF; typedef double TO __attribute__((unused)); LAUNCH; }       \
      else if (f_dtype == AGGF_F64 && out_dtype == AGGF_F64) { typedef double TF; typedef double TO __attribute__((unused)); LAUNCH; } \
      else return fail(AGGF_ERR_ARG, WHO ": bad dtype (float64 features give float64 products)");              \
    } else {                                                                                                   \
      return fail(AGGF_ERR_ARG, WHO ": bad feature dtype");                                                    \
    }                                                                                                          \
  } while (0)

extern "C" int aggf_gb_regmat(const void* Fg, int f_dtype, const void* Pg, const void* cg, int g_dtype, int64_t T,
                              int32_t G, int32_t n_cg, int32_t site, const float* sizes, int32_t n_id,
                              int32_t n_ch, const void* centers, int32_t n_basis, double width,
                              double clip, double kbt, int32_t ld_feat, void* R3, int out_dtype, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!Fg || !Pg || !cg || !sizes || !R3) return fail(AGGF_ERR_ARG, "aggf_gb_regmat: NULL pointer");
  if (T <= 0 || G <= 0 || n_ch < 0 || n_ch > G || n_id < 0 || n_id > G || site < 0 || site >= n_cg ||
      ld_feat < n_id + n_ch * n_basis)
    return fail(AGGF_ERR_ARG, "aggf_gb_regmat: bad shape");
  int rc = check_gb(centers, n_basis, width);
  if (rc) return rc;
  const dim3 grid = feat_grid(T * (n_id + n_ch));
  AGGF_GB_DISPATCH("aggf_gb_regmat",
                   AGGF_LAUNCH((gb_regmat_kernel<TF, TG, TO>), grid, dim3(256), 0, stream, (const TF*)Fg,
                                      (const TG*)Pg, (const TG*)cg, T, G, n_cg, site, sizes, n_id, n_ch, gp, kbt,
                                      ld_feat, (TO*)R3));
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_gb_apply(const void* Fg, int f_dtype, const void* Pg, const void* cg, int g_dtype, int64_t T,
                             int32_t G, int32_t n_cg, const float* sizes, int32_t n_id, int32_t n_ch,
                             const void* centers, int32_t n_basis, double width, double clip,
                             const double* coef, int32_t n_feat, double* out, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!Fg || !Pg || !cg || !sizes || !coef || !out) return fail(AGGF_ERR_ARG, "aggf_gb_apply: NULL pointer");
  if (T <= 0 || G <= 0 || n_cg <= 0 || n_ch < 0 || n_ch > G || n_id < 0 || n_id > G ||
      n_feat != n_id + n_ch * n_basis)
    return fail(AGGF_ERR_ARG, "aggf_gb_apply: bad shape");
  int rc = check_gb(centers, n_basis, width);
  if (rc) return rc;
  const dim3 grid = feat_grid(((T + GB_FR - 1) / GB_FR) * n_cg * 64);
  const int out_dtype = AGGF_F64;
  AGGF_GB_DISPATCH("aggf_gb_apply",
                   AGGF_LAUNCH((gb_apply_kernel<TF, TG>), grid, dim3(256), 0, stream, (const TF*)Fg,
                                      (const TG*)Pg, (const TG*)cg, T, G, n_cg, sizes, n_id, n_ch, gp, coef, n_feat,
                                      out));
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_gb_apply_cols(const void* Fg, int f_dtype, const void* Pg, const void* cg, int g_dtype, int64_t T,
                                  int32_t G, int32_t n_cg, const float* sizes, int32_t n_id, const double* coef_id,
                                  const int32_t* col_ptr, const int32_t* col_idx, const double* col_val,
                                  const void* centers, int32_t n_basis, double width, double clip, double* out,
                                  void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!Fg || !Pg || !cg || !sizes || !col_ptr || !out || (n_id > 0 && !coef_id))
    return fail(AGGF_ERR_ARG, "aggf_gb_apply_cols: NULL pointer");
  if (T <= 0 || G <= 0 || n_cg <= 0 || n_id < 0 || n_id > G) return fail(AGGF_ERR_ARG, "aggf_gb_apply_cols: bad shape");
  int rc = check_gb(centers, n_basis, width);
  if (rc) return rc;
  const dim3 grid = feat_grid(((T + GB_FR - 1) / GB_FR) * n_cg * 64);
  const int out_dtype = AGGF_F64;
  AGGF_GB_DISPATCH("aggf_gb_apply_cols",
                   AGGF_LAUNCH((gb_apply_cols_kernel<TF, TG>), grid, dim3(256), 0, stream, (const TF*)Fg,
                                      (const TG*)Pg, (const TG*)cg, T, G, n_cg, sizes, n_id, coef_id, col_ptr, col_idx,
                                      col_val, gp, out));
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_gb_distance_range(const float* Pg, const float* cg, int64_t T, int32_t G, int32_t n_cg,
                                      int32_t n_ch, float* rmin, float* rmax, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!Pg || !cg || !rmin || !rmax) return fail(AGGF_ERR_ARG, "aggf_gb_distance_range: NULL pointer");
  if (T <= 0 || G <= 0 || n_cg <= 0 || n_cg > 65535 || n_ch <= 0 || n_ch > G)
    return fail(AGGF_ERR_ARG, "aggf_gb_distance_range: bad shape");
  int64_t slices = ceil_div(T, 256);
  if (slices > 256) slices = 256;
  const dim3 grid((unsigned)ceil_div(n_ch, 256), (unsigned)n_cg, (unsigned)slices);
  AGGF_LAUNCH(gb_range_kernel, grid, dim3(256), 0, stream, Pg, cg, T, G, n_cg, n_ch, rmin, rmax);
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_gb_regmat_cols(const void* Fg, int f_dtype, const void* Pg, const void* cg, int g_dtype, int64_t T,
                                   int32_t G, int32_t n_cg, int32_t site, const float* sizes, int32_t n_id,
                                   const int32_t* cols, int32_t n_cols, const void* centers, int32_t n_basis,
                                   double width, double clip, double kbt, int32_t ld_feat, void* R3,
                                   int out_dtype, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!Fg || !Pg || !cg || !sizes || !R3 || (n_cols > 0 && !cols))
    return fail(AGGF_ERR_ARG, "aggf_gb_regmat_cols: NULL pointer");
  if (T <= 0 || G <= 0 || n_cols < 0 || n_id < 0 || n_id > G || n_id + n_cols <= 0 || site < 0 || site >= n_cg ||
      ld_feat < n_id + n_cols)
    return fail(AGGF_ERR_ARG, "aggf_gb_regmat_cols: bad shape");
  int rc = check_gb(centers, n_basis, width);
  if (rc) return rc;
  const dim3 grid = feat_grid(T * (n_id + n_cols));
  AGGF_GB_DISPATCH("aggf_gb_regmat_cols",
                   AGGF_LAUNCH((gb_regmat_cols_kernel<TF, TG, TO>), grid, dim3(256), 0, stream, (const TF*)Fg,
                                      (const TG*)Pg, (const TG*)cg, T, G, n_cg, site, sizes, n_id, cols, n_cols, gp, kbt,
                                      ld_feat, (TO*)R3));
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}
