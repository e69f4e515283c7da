// K3: LinearMap apply  out[t,c,d] = sum_a M[c,a] * P[t,a,d]  on MFMA, plus the one-hot
// (slice map) gather.  Replaces util.trjdot (util.py:119-125) behind
// LinearMap.__call__ (map/core.py:219-240) of the reference.
//
// One workgroup = 64 frames x 64 coarse-grained sites; the atom axis is the MFMA K
// dimension, walked in chunks of 16 atoms.  Frame rows are staged in LDS as they lie
// in HBM ((atom, xyz) interleaved); the three xyz components are three accumulator
// sets fed from the same staged tile, so P is read exactly once per site block.
#include <stdlib.h>

#include <type_traits>

#include "aggf_common.h"

namespace aggf {

// Two tilings of a workgroup: 4 waves = 64 frames x 64 sites (n_cg <= 64), and 8 waves = 64 frames x
// 128 sites (two wave columns share the staged P tile: P is re-read half as often from L2/HBM and
// each thread stages 5 instead of 8 sixteen-byte chunks per 48 MFMAs).
constexpr int AP_KA = 16;   // atoms per stage
constexpr int AP_XS = AP_KA * 3 + 2;  // LDS row stride of the P tile (elements)
constexpr int AP_MS = AP_KA + 2;      // LDS row stride of the M tile

// a 16-byte vector load from an address that is only element-aligned (frame rows of an odd atom count are 8-byte --
// float64 -- or 4-byte -- float32 -- aligned): still ONE global_load_dwordx4, the memory system takes any byte address
// (tools/dma_align_probe.hip); through an element-wise loop the 1001-atom apply took 1.75x the time of 1000 atoms
template <typename V, typename E>
__device__ __forceinline__ V load_vec_elem_aligned(const E* p) {
  struct __attribute__((packed, aligned(alignof(E)))) U {
    V v;
  };
  return reinterpret_cast<const U*>(p)->v;
}

template <typename TC>
__device__ __forceinline__ TC fix_nan(TC v, bool replace, TC fill) {
  return (replace && v != v) ? fill : v;
}

#ifdef AGGF_APPLY_PROF
// tools/apply_probe.hip: shader cycles per wave spent in load issue / MFMA / LDS refill / barrier
__device__ unsigned long long aggf_apply_prof[5];
#define AP_T(x) const uint64_t x = __builtin_readcyclecounter()
#else
#define AP_T(x)
#endif

template <typename TIn, typename TC, bool NANREP, int AP_THREADS, int AP_TC, int AP_WF = 4>
__global__ __launch_bounds__(AP_THREADS, AP_THREADS >= 1024 ? 4 : 2) void apply_kernel(
    const TIn* __restrict__ P, int64_t T, int32_t N, const TC* __restrict__ Mx, int32_t n_cg,
    int32_t ncb, TC nan_fill, int p_vec_ok, int m_vec_ok, TC* __restrict__ out,
    double* __restrict__ sumsq_partials, int32_t* __restrict__ nan_seen) {
  using MF = Mfma<TC>;
  using acc_t = typename MF::acc_t;
  constexpr int NWAVE = AP_THREADS / 64;
  constexpr int WF = AP_WF;              // waves along the frame axis; the others split the site axis
  constexpr int WC = NWAVE / WF;
  constexpr int AP_TF = 16 * WF;         // frames per workgroup (one 16-frame MFMA row block per wave row)
  constexpr int NCT = AP_TC / 16 / WC;   // 16-site column tiles per wave
  constexpr int VI = 16 / sizeof(TIn);   // input elements per 16-byte chunk
  constexpr int VM = 16 / sizeof(TC);
  constexpr int P_CH_ROW = AP_KA * 3 / VI;                 // chunks per P tile row
  constexpr int P_CH = P_CH_ROW * AP_TF;
  constexpr int P_PER_THREAD = (P_CH + AP_THREADS - 1) / AP_THREADS;
  constexpr int M_CH_ROW = AP_KA / VM;
  constexpr int M_CH = M_CH_ROW * AP_TC;
  constexpr int M_PER_THREAD = (M_CH + AP_THREADS - 1) / AP_THREADS;
  constexpr int XBUF = AP_TF * AP_XS;
  constexpr int MBUF = AP_TC * AP_MS;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  TC* sX = reinterpret_cast<TC*>(smem_raw);     // [2][XBUF]
  TC* sM = sX + 2 * XBUF;                       // [2][MBUF]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t bid = blockIdx.x;
  const int cb = (int)(bid % ncb);
  const int64_t fb = bid / ncb;
  const int64_t t0 = fb * AP_TF;
  const int c0 = cb * AP_TC;
  const int64_t rowP = (int64_t)N * 3;
  const int n_stage = (N + AP_KA - 1) / AP_KA;

  // two register sets: the global loads of stage s+2 are issued at the top of stage s and reach LDS
  // at the end of stage s+1.  With one stage of distance the refill waited ~3500 cycles per stage for
  // its loads (a CU keeps only a few dozen cache-line misses in flight; tools/apply_probe.hip).
  TIn rp[2][P_PER_THREAD][VI];
  TC rm[2][M_PER_THREAD][VM];
  bool saw_nan = false;  // any NaN among the P values this thread staged (fused _has_nans scan)

  auto load_stage = [&](int s, auto set_c) {
    constexpr int SET = decltype(set_c)::value;
    const int a0 = s * AP_KA;
    // Whole stage inside the rows and both operands 16-byte aligned (every stage but a ragged last one): straight-line
    // vector loads.  With the element-wise tail handling inside the same loop body the compiler put an
    // s_waitcnt vmcnt(0) after EVERY 16-byte load (the two paths join in a phi), so the loads that were meant to run
    // two stages ahead each waited out a full memory latency right where they were issued.
    if (p_vec_ok && m_vec_ok && a0 + AP_KA <= N) {
#pragma unroll
      for (int q = 0; q < P_PER_THREAD; ++q) {
        const int c = tid + q * AP_THREADS;
        const int r = c / P_CH_ROW, col = (c - r * P_CH_ROW) * VI;  // col in [0,48)
        const int64_t t = t0 + r;
        typedef TIn __attribute__((ext_vector_type(VI))) vin_t;
        vin_t v;
#pragma unroll
        for (int e = 0; e < VI; ++e) v[e] = 0;
        if (c < P_CH && t < T) v = load_vec_elem_aligned<vin_t>(P + t * rowP + (int64_t)a0 * 3 + col);
#pragma unroll
        for (int e = 0; e < VI; ++e) rp[SET][q][e] = v[e];
      }
#pragma unroll
      for (int q = 0; q < M_PER_THREAD; ++q) {
        const int c = tid + q * AP_THREADS;
        const int r = c / M_CH_ROW, col = (c - r * M_CH_ROW) * VM;
        const int cg = c0 + r;
        typedef TC __attribute__((ext_vector_type(VM))) vm_t;
        vm_t v;
#pragma unroll
        for (int e = 0; e < VM; ++e) v[e] = 0;
        if (c < M_CH && cg < n_cg) v = load_vec_elem_aligned<vm_t>(Mx + (int64_t)cg * N + a0 + col);
#pragma unroll
        for (int e = 0; e < VM; ++e) rm[SET][q][e] = v[e];
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < P_PER_THREAD; ++q) {
      const int c = tid + q * AP_THREADS;
      const int r = c / P_CH_ROW, col = (c - r * P_CH_ROW) * VI;  // col in [0,48)
      const int64_t t = t0 + r;
      const int64_t e0 = (int64_t)a0 * 3 + col;                    // element in the frame row
#pragma unroll
      for (int e = 0; e < VI; ++e) rp[SET][q][e] = 0;
      if (c < P_CH && t < T) {
        const TIn* src = P + t * rowP + e0;
        if (p_vec_ok && e0 + VI <= rowP) {
          typedef TIn __attribute__((ext_vector_type(VI))) vin_t;
          vin_t v = load_vec_elem_aligned<vin_t>(src);
#pragma unroll
          for (int e = 0; e < VI; ++e) rp[SET][q][e] = v[e];
        } else {
#pragma unroll
          for (int e = 0; e < VI; ++e)
            if (e0 + e < rowP) rp[SET][q][e] = src[e];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < M_PER_THREAD; ++q) {
      const int c = tid + q * AP_THREADS;
      const int r = c / M_CH_ROW, col = (c - r * M_CH_ROW) * VM;
      const int cg = c0 + r;
      const int a = a0 + col;
#pragma unroll
      for (int e = 0; e < VM; ++e) rm[SET][q][e] = 0;
      if (c < M_CH && cg < n_cg) {
        const TC* src = Mx + (int64_t)cg * N + a;
        if (m_vec_ok && a + VM <= N) {
          typedef TC __attribute__((ext_vector_type(VM))) vm_t;
          vm_t v = load_vec_elem_aligned<vm_t>(src);
#pragma unroll
          for (int e = 0; e < VM; ++e) rm[SET][q][e] = v[e];
        } else {
#pragma unroll
          for (int e = 0; e < VM; ++e)
            if (a + e < N) rm[SET][q][e] = src[e];
        }
      }
    }
  };
  auto store_stage = [&](int buf, auto set_c) {
    constexpr int SET = decltype(set_c)::value;
    TC* x = sX + buf * XBUF;
    TC* m = sM + buf * MBUF;
#pragma unroll
    for (int q = 0; q < P_PER_THREAD; ++q) {
      const int c = tid + q * AP_THREADS;
      const int r = c / P_CH_ROW, col = (c - r * P_CH_ROW) * VI;
      if (c < P_CH) {
#pragma unroll
        for (int e = 0; e < VI; ++e) {
          saw_nan |= (rp[SET][q][e] != rp[SET][q][e]);
          x[r * AP_XS + col + e] = fix_nan<TC>((TC)rp[SET][q][e], NANREP, nan_fill);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < M_PER_THREAD; ++q) {
      const int c = tid + q * AP_THREADS;
      const int r = c / M_CH_ROW, col = (c - r * M_CH_ROW) * VM;
      if (c < M_CH) {
#pragma unroll
        for (int e = 0; e < VM; ++e) m[r * AP_MS + col + e] = rm[SET][q][e];
      }
    }
  };

  acc_t acc[NCT][3];
#pragma unroll
  for (int n = 0; n < NCT; ++n)
#pragma unroll
    for (int d = 0; d < 3; ++d) acc[n][d] = acc_zero<TC>();

  const int wf = wave % WF, wc = wave / WF;
  const int offX = (16 * wf + (lane & 15)) * AP_XS + 3 * (lane >> 4);
  const int offM = (wc * NCT * 16 + (lane & 15)) * AP_MS + (lane >> 4);

#ifdef AGGF_APPLY_PROF
  uint64_t pf[4] = {0, 0, 0, 0};
#endif
  using set0 = std::integral_constant<int, 0>;
  using set1 = std::integral_constant<int, 1>;
  auto stage = [&](int s, auto set_c) {
    // set_c = s & 1: the set that held stage s (already in LDS) is free for stage s+2; the other set
    // holds stage s+1, loaded during stage s-1
    constexpr int SET = decltype(set_c)::value;
    using other = std::integral_constant<int, 1 - SET>;
    const int cur = SET;
    AP_T(q0);
#if !defined(AGGF_APPLY_ABL) || AGGF_APPLY_ABL < 1
    if (s + 2 < n_stage) load_stage(s + 2, set_c);
#endif
    AP_T(q1);
    const TC* x = sX + cur * XBUF;
    const TC* m = sM + cur * MBUF;
#pragma unroll
    for (int kk = 0; kk < AP_KA / 4; ++kk) {
      TC a[3], b[NCT];
#pragma unroll
      for (int d = 0; d < 3; ++d) a[d] = x[offX + 12 * kk + d];
#pragma unroll
      for (int n = 0; n < NCT; ++n) b[n] = m[offM + 16 * n * AP_MS + 4 * kk];
      if (kk == AP_KA / 4 - 1) {
        // LDS refill and barrier IN FRONT of the stage's last MFMAs (their operands are in registers): the six MFMAs
        // run while the 16 waves meet.  Same box, back to back, C3: 106.3 -> 104.4 ms; c5 14.2 -> 13.8 ms.
#if !defined(AGGF_APPLY_ABL) || AGGF_APPLY_ABL < 2
        if (s + 1 < n_stage) store_stage(cur ^ 1, other{});
#endif
        __syncthreads();
      }
#pragma unroll
      for (int n = 0; n < NCT; ++n)
#pragma unroll
        for (int d = 0; d < 3; ++d) acc[n][d] = MF::mma(a[d], b[n], acc[n][d]);
    }
    AP_T(q2);
    AP_T(q3);
#ifdef AGGF_APPLY_PROF
    const uint64_t q4 = __builtin_readcyclecounter();
    pf[0] += q1 - q0;
    pf[1] += q2 - q1;
    pf[2] += q3 - q2;
    pf[3] += q4 - q3;
#endif
  };
  load_stage(0, set0{});
  store_stage(0, set0{});
  if (n_stage > 1) load_stage(1, set1{});
  __syncthreads();
  for (int s = 0; s < n_stage; s += 2) {
    stage(s, set0{});
    if (s + 1 < n_stage) stage(s + 1, set1{});
  }

#ifdef AGGF_APPLY_PROF
  if (lane == 0) {
    for (int i = 0; i < 4; ++i) atomicAdd(&aggf_apply_prof[i], (unsigned long long)pf[i]);
    atomicAdd(&aggf_apply_prof[4], (unsigned long long)n_stage);
  }
#endif
  if (nan_seen && cb == 0 && __any(saw_nan) && lane == 0) atomicOr(nan_seen, 1);

  // epilogue: out[t, c, d]; optional sum of squares (fixed order: lane tree, then waves)
  double ss = 0.0;
#pragma unroll
  for (int n = 0; n < NCT; ++n) {
    const int c = c0 + (wc * NCT + n) * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t t = t0 + 16 * wf + MF::row(lane, r);
      if (t < T && c < n_cg) {
        TC* o = out + (t * n_cg + c) * 3;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const TC v = acc[n][d][r];
          o[d] = v;
          ss += (double)v * (double)v;
        }
      }
    }
  }
  if (sumsq_partials) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
    __shared__ double wsum[NWAVE];
    if (lane == 0) wsum[wave] = ss;
    __syncthreads();
    if (tid == 0) {
      double tot = 0.0;
#pragma unroll
      for (int w = 0; w < NWAVE; ++w) tot += wsum[w];
      sumsq_partials[bid] = tot;
    }
  }
}

// fixed-order final sum of per-workgroup partials: one workgroup, strided then tree
__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ part,
                                                           int64_t n, double* __restrict__ out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += part[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}

// K3b: one-hot (slice) maps -- out[t, c, :] = P[t, idx[c], :] (map/core.py:219-240 on a map whose rows are unit
// vectors).  A thread owns ONE element position e = 3 c + d of the output row for a strided set of frames: its source
// offset 3 idx[c] + d is looked up once, outside the frame loop (the first version divided a 64-bit element index per
// element), and the loop body is UF independent 4/8-byte loads followed by UF stores -- consecutive threads read the
// three components of one atom, then the next selected atom, and write consecutive elements.  Loads and stores are
// non-temporal: every byte is used once, and the kernel runs on a side stream beside K1 / K3, whose panels live in
// the L2.  Fused NaN scan of the gathered values (the slice map's NaN policy: a NaN at a SELECTED site is the case
// in which the reference's NaN -> 0 / NaN -> -1 products differ, map/core.py:226-236).
constexpr int SG_UF = 8;

// `lanes` (a power of two <= 256) threads serve one frame's 3 n_cg output elements and 256 / lanes frames share a
// workgroup pass: with few sites (CLN025: 30 elements) a whole workgroup per frame left 226 of its 256 threads idle
// (c1: 3.7 ms for 5 GB of lines).
template <typename TIn, typename TO>
__global__ __launch_bounds__(256) void slice_gather_kernel(const TIn* __restrict__ P, int64_t T,
                                                           int32_t N, const int32_t* __restrict__ idx,
                                                           int32_t n_cg, TO* __restrict__ out,
                                                           int32_t* __restrict__ nan_seen, int lanes) {
  const int row_out = n_cg * 3;
  const int fpw = 256 / lanes;                       // frames per workgroup pass
  const int f = threadIdx.x / lanes;                 // this thread's frame inside a pass
  const int e = blockIdx.y * lanes + (threadIdx.x & (lanes - 1));
  const bool valid = e < row_out;
  const int c = valid ? e / 3 : 0;
  const int64_t off = (int64_t)idx[c] * 3 + (e - 3 * c);
  const int64_t row_in = (int64_t)N * 3;
  bool saw_nan = false;
  const int64_t step = (int64_t)SG_UF * fpw;
  for (int64_t t0 = (int64_t)blockIdx.x * step + f; t0 < T; t0 += (int64_t)gridDim.x * step) {
    TIn v[SG_UF];
#pragma unroll
    for (int u = 0; u < SG_UF; ++u) {
      v[u] = (TIn)0;
      if (valid && t0 + (int64_t)u * fpw < T) v[u] = __builtin_nontemporal_load(P + (t0 + (int64_t)u * fpw) * row_in + off);
    }
#pragma unroll
    for (int u = 0; u < SG_UF; ++u) {
      saw_nan |= (v[u] != v[u]);
      if (valid && t0 + (int64_t)u * fpw < T) __builtin_nontemporal_store((TO)v[u], out + (t0 + (int64_t)u * fpw) * row_out + e);
    }
  }
  if (nan_seen && __any(saw_nan) && (threadIdx.x & 63) == 0) atomicOr(nan_seen, 1);
}

// ---------------------------------------------------------------------------
// Few coarse-grained sites (n_cg <= 16, e.g. CLN025's 10 beads): the apply is HBM-bound (2 * 3 * n_cg flop per
// 3 s bytes read), and the tile kernel above would multiply 64-site tiles that are 5/6 padding.  Same plan as the
// small-system Gram kernel: a stage = 8 frames = ONE contiguous run of 8 * 3N elements, fetched with 16-byte loads
// into registers during the previous stage's MFMAs and parked in LDS as it lies in HBM; the MFMA operands are read
// straight from there -- rows = sites (the map, zero padded, in LDS), columns = (frame, xyz) pairs, K = atoms:
//   D[c][(t,d)] += M[c][a] * P[t][a][d].
// 8 frames x 3 = 24 columns = two 16-column blocks (waves 0 and 1 multiply, all 8 waves fetch and park); stages are
// dealt round-robin to 2 resident workgroups per CU.  NaN policy, fused NaN scan and sum of squares as in the tile
// kernel.
// Frames per stage KB = 8 / 12 / 20 / 40 (-> 24 / 36 / 60 / 120 columns in NBLK = 2 / 4 / 4 / 8 blocks of 16, the K steps
// of a block split over KQ = 4 / 2 / 2 / 1 waves), chosen so that a stage is ~30 KB of frames whatever the atom count: with 8 frames
// for everything, 32 atoms streamed at 0.36 of 8 TB/s and 64 atoms at 0.50 against 0.68 at 128 atoms (two barriers and
// the partial-sum exchange per 6 KB / 12 KB of frames; profiles/r04_stream_kernels.jsonl).
constexpr int AS_THREADS = 512, AS_NV = 5;   // 5 x 16 B x 512 threads = 40 KB of frames per stage at most

template <typename TIn, typename TC, bool NANREP, int AS_KB, int NBLK>
__global__ __launch_bounds__(AS_THREADS, 4) void apply_small_kernel(
    const TIn* __restrict__ P, int64_t T, int32_t N, const TC* __restrict__ Mx, int32_t n_cg, TC nan_fill,
    int32_t raw_bytes, int32_t n_pad, TC* __restrict__ out, double* __restrict__ sumsq_partials,
    int32_t* __restrict__ nan_seen) {
  using MF = Mfma<TC>;
  using acc_t = typename MF::acc_t;
  typedef float __attribute__((ext_vector_type(4))) v16_t;
  typedef float __attribute__((ext_vector_type(4), aligned(4))) v16g_t;  // a piece in HBM: P is element-aligned only
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  TIn* raw = reinterpret_cast<TIn*>(smem_raw);                       // AS_KB frames as in HBM (+ 64 zero bytes)
  TC* ms = reinterpret_cast<TC*>(smem_raw + raw_bytes + 64);         // [16][n_pad + 2]: the map, zero padded
  const int ms_ld = n_pad + 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t row_in = (int64_t)N * 3;
  const int64_t n_stage_all = (T + AS_KB - 1) / AS_KB;
  const int n_it = blockIdx.x < n_stage_all ? (int)((n_stage_all - 1 - blockIdx.x) / gridDim.x + 1) : 0;
  auto stage_t0 = [&](int k) { return ((int64_t)blockIdx.x + (int64_t)k * gridDim.x) * AS_KB; };

  for (int e = tid; e < 16 * ms_ld; e += AS_THREADS) {
    const int c = e / ms_ld, a = e - c * ms_ld;
    ms[e] = (c < n_cg && a < N) ? Mx[(int64_t)c * N + a] : (TC)0;
  }
  if (tid < 64 / (int)sizeof(TIn)) raw[raw_bytes / (int)sizeof(TIn) + tid] = (TIn)0;  // reads past the last frame: zeros

  const int n_vec = raw_bytes / 16;
  v16_t hold[AS_NV];
  bool saw_nan = false;
  auto fetch = [&](int s) {
    const int64_t t0 = stage_t0(s);
    const int64_t valid = (T - t0 < AS_KB ? T - t0 : AS_KB) * row_in * (int64_t)sizeof(TIn);
    const char* src = reinterpret_cast<const char*>(P + t0 * row_in);
    if (valid >= (int64_t)n_vec * 16) {
      // every frame of the stage exists (all stages but the last of the trajectory): straight-line loads.  With the
      // ragged-end handling in the same loop the compiler put an s_waitcnt vmcnt(0) after EVERY load (the paths
      // join in a phi): the loads of a stage went out one memory latency apart (tools/small_probe.hip: 6500 of
      // 13600 cycles per stage were spent "issuing" five loads)
#pragma unroll
      for (int i = 0; i < AS_NV; ++i) {
        const int v = tid + AS_THREADS * i;
        v16_t x = {0.f, 0.f, 0.f, 0.f};
        // non-temporal: every byte of the trajectory is read once (tools/ldsdma_fill.hip: 6.8 against 6.1 TB/s for
        // this load shape on a stream that is read once)
        if (v < n_vec) x = __builtin_nontemporal_load(reinterpret_cast<const v16g_t*>(src + (int64_t)v * 16));
        hold[i] = x;
      }
    } else {
#pragma unroll
      for (int i = 0; i < AS_NV; ++i) {
        const int v = tid + AS_THREADS * i;
        v16_t x = {0.f, 0.f, 0.f, 0.f};
        if (v < n_vec) {
          const int64_t off = (int64_t)v * 16;
          if (off + 16 <= valid) {
            x = *reinterpret_cast<const v16g_t*>(src + off);
          } else if (off < valid) {  // the ragged end of the trajectory: element by element
            TIn tmp[16 / sizeof(TIn)];
#pragma unroll
            for (int k = 0; k < (int)(16 / sizeof(TIn)); ++k)
              tmp[k] = off + (k + 1) * (int64_t)sizeof(TIn) <= valid ? reinterpret_cast<const TIn*>(src + off)[k] : (TIn)0;
            x = *reinterpret_cast<v16_t*>(tmp);
          }
        }
        hold[i] = x;
      }
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int i = 0; i < AS_NV; ++i) {
      const int v = tid + AS_THREADS * i;
      if (v < n_vec) {
        TIn tmp[16 / sizeof(TIn)];
        *reinterpret_cast<v16_t*>(tmp) = hold[i];
#pragma unroll
        for (int k = 0; k < (int)(16 / sizeof(TIn)); ++k) {
          saw_nan |= (tmp[k] != tmp[k]);
          if (NANREP && tmp[k] != tmp[k]) tmp[k] = (TIn)nan_fill;
        }
        reinterpret_cast<v16_t*>(raw)[v] = *reinterpret_cast<v16_t*>(tmp);
      }
    }
  };

  // MFMA operands.  A[i = site][k = atom]: lane reads ms[lane & 15][a0 + (lane >> 4)].
  // B[k = atom][j = column]: column j = 16 blk + (lane & 15) = 3 t + d (j < 3 KB; others: the zero slot).
  // All 8 waves multiply: wave = (column block blk = wave % NBLK) x (K share kq = wave / NBLK: K steps kq, kq + KQ,
  // ...); the KQ partial accumulators of a block are summed through LDS in a fixed order.  (With only the two
  // block waves multiplying, the 44 dependent LDS-read + MFMA steps of a stage were the critical path: 7.1 ms at
  // CLN025 x 4e6 frames.)
  constexpr int KQ = 8 / NBLK;
  static_assert(NBLK * KQ == 8 && 16 * NBLK >= 3 * AS_KB, "wave split");
  const int blk = wave % NBLK, kq = wave / NBLK;
  const int offA = (lane & 15) * ms_ld + (lane >> 4);
  const int j = 16 * blk + (lane & 15);
  const bool col_ok = j < AS_KB * 3;
  const int jt = j / 3, jd = j - 3 * jt;
  const int offB = col_ok ? jt * (int)row_in + jd + 3 * (lane >> 4) : -1;
  const int zero_off = raw_bytes / (int)sizeof(TIn);
  TC* part = ms + 16 * ms_ld;  // [8 waves][4][64]: partial accumulators
  double ss = 0.0;

  if (n_it > 0) fetch(0);
  for (int s = 0; s < n_it; ++s) {
    // (no barrier in front of park(): the MFMAs of stage s-1 read `raw` in front of that stage's last barrier, and
    // `part`, which waves 0 and 1 may still be reading, is not written before the next barrier)
    park();
    __syncthreads();
    if (s + 1 < n_it) fetch(s + 1);
    acc_t acc = acc_zero<TC>();
    // two K steps per trip: both pairs of operand reads are in flight before the first MFMA
    for (int a0 = 4 * kq; a0 < n_pad; a0 += 8 * KQ) {
      const int a1 = a0 + 4 * KQ;
      const bool second = a1 < n_pad;
      const TC xa = ms[offA + a0];
      // atoms past N (the last K step): the element at that address belongs to the NEXT frame -- read the zero
      // slot instead, or a NaN there would leak into this frame (0 x NaN)
      const TC xb = (TC)((offB >= 0 && a0 + (lane >> 4) < N) ? raw[offB + 3 * a0] : raw[zero_off]);
      const TC ya = second ? ms[offA + a1] : (TC)0;
      const TC yb = (TC)((second && offB >= 0 && a1 + (lane >> 4) < N) ? raw[offB + 3 * a1] : raw[zero_off]);
      acc = MF::mma(xa, xb, acc);
      acc = MF::mma(ya, yb, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[(wave * 4 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (kq == 0) {
      const int64_t t = stage_t0(s) + jt;
      if (col_ok && t < T) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = MF::row(lane, r);
          if (c < n_cg) {
            TC v = part[(blk * 4 + r) * 64 + lane];
#pragma unroll
            for (int q = 1; q < KQ; ++q) v += part[((blk + NBLK * q) * 4 + r) * 64 + lane];
            out[(t * n_cg + c) * 3 + jd] = v;
            ss += (double)v * (double)v;
          }
        }
      }
    }
  }
  if (nan_seen && __any(saw_nan) && lane == 0) atomicOr(nan_seen, 1);
  if (sumsq_partials) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
    __shared__ double wsum[NBLK];
    if (lane == 0 && wave < NBLK) wsum[wave] = ss;  // (the kq = 0 waves wrote the outputs)
    __syncthreads();
    if (tid == 0) {
      double tot = wsum[0];
#pragma unroll
      for (int q = 1; q < NBLK; ++q) tot += wsum[q];
      sumsq_partials[blockIdx.x] = tot;
    }
  }
}

// frames per stage of the few-sites kernel for this frame size (0: the frames of even 8 do not fit the registers)
template <typename TIn>
static int apply_small_frames(int32_t N) {
  const int64_t frame = (int64_t)3 * N * (int64_t)sizeof(TIn), cap = (int64_t)AS_NV * AS_THREADS * 16;
  return 40 * frame <= cap ? 40 : (20 * frame <= cap ? 20 : (12 * frame <= cap ? 12 : (8 * frame <= cap ? 8 : 0)));
}

template <typename TIn, typename TC, int KB, int NBLK>
static int apply_small_launch(const void* P, int64_t T, int32_t N, const void* Mx, int32_t n_cg, int nan_mode,
                              double nan_fill, void* out, double* sumsq, int32_t* nan_seen, void* ws, size_t ws_bytes,
                              hipStream_t stream) {
  const int32_t raw_bytes = (int32_t)((int64_t)KB * 3 * N * sizeof(TIn));  // a multiple of 16 for 4- and 8-byte elements
  const int32_t n_pad = (int32_t)round_up(N, 4);
  const size_t lds = (size_t)raw_bytes + 64 + (size_t)16 * (n_pad + 2) * sizeof(TC) + (size_t)8 * 4 * 64 * sizeof(TC);
  int64_t nwg = 2 * (int64_t)device_cu_count();
  const int64_t n_stage_all = ceil_div(T, KB);
  if (nwg > n_stage_all) nwg = n_stage_all;
  double* partials = nullptr;
  if (sumsq) {
    if (!ws || ws_bytes < (size_t)nwg * sizeof(double))
      return fail(AGGF_ERR_WORKSPACE, "aggf_linearmap_apply: workspace too small for sumsq");
    partials = reinterpret_cast<double*>(ws);
  }
  const bool rep = nan_mode == AGGF_NAN_REPLACE;
  if (lds > 65536) {
    static thread_local PerDeviceOnce once_t, once_f;
    bool& done = rep ? *once_t.flag() : *once_f.flag();
    if (!done) {
      if (rep)
        AGGF_HIP_OK(hipFuncSetAttribute((const void*)apply_small_kernel<TIn, TC, true, KB, NBLK>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
      else
        AGGF_HIP_OK(hipFuncSetAttribute((const void*)apply_small_kernel<TIn, TC, false, KB, NBLK>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
      done = true;
    }
  }
  if (rep)
    AGGF_LAUNCH((apply_small_kernel<TIn, TC, true, KB, NBLK>), dim3((unsigned)nwg), dim3(AS_THREADS), lds, stream,
                       (const TIn*)P, T, N, (const TC*)Mx, n_cg, (TC)nan_fill, raw_bytes, n_pad, (TC*)out, partials, nan_seen);
  else
    AGGF_LAUNCH((apply_small_kernel<TIn, TC, false, KB, NBLK>), dim3((unsigned)nwg), dim3(AS_THREADS), lds, stream,
                       (const TIn*)P, T, N, (const TC*)Mx, n_cg, (TC)0, raw_bytes, n_pad, (TC*)out, partials, nan_seen);
  AGGF_LAUNCH_OK();
  if (sumsq) {
    AGGF_LAUNCH(sum_partials_kernel, dim3(1), dim3(256), 0, stream, partials, nwg, sumsq);
    AGGF_LAUNCH_OK();
  }
  return AGGF_OK;
}

template <typename TIn, typename TC, int THREADS, int TCB, int WFR = 4>
static int apply_launch(const void* P, int64_t T, int32_t N, const void* Mx, int32_t n_cg, int nan_mode,
                        double nan_fill, void* out, double* sumsq, int32_t* nan_seen, void* ws,
                        size_t ws_bytes, hipStream_t stream) {
  constexpr int TF = 16 * WFR;
  const int ncb = (int)ceil_div(n_cg, TCB);
  const int64_t nfb = ceil_div(T, TF);
  const int64_t nblocks = nfb * ncb;
  if (nblocks > 0x7fffffffLL) return fail(AGGF_ERR_ARG, "apply grid too large");
  double* partials = nullptr;
  if (sumsq) {
    if (!ws || ws_bytes < (size_t)nblocks * sizeof(double))
      return fail(AGGF_ERR_WORKSPACE, "aggf_linearmap_apply: workspace too small for sumsq");
    partials = reinterpret_cast<double*>(ws);
  }
  // 16-byte loads wherever a chunk lies inside its row: the pointers need element alignment only
  const int p_vec_ok = ((uintptr_t)P % sizeof(TIn)) == 0;
  const int m_vec_ok = ((uintptr_t)Mx % sizeof(TC)) == 0;
  const size_t lds = (size_t)2 * (TF * AP_XS + TCB * AP_MS) * sizeof(TC);
  if (lds > 65536) {
    static thread_local PerDeviceOnce once_t, once_f;
    bool &done_t = *once_t.flag(), &done_f = *once_f.flag();
    if (!done_t) {
      AGGF_HIP_OK(hipFuncSetAttribute((const void*)apply_kernel<TIn, TC, true, THREADS, TCB, WFR>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      done_t = true;
    }
    if (!done_f) {
      AGGF_HIP_OK(hipFuncSetAttribute((const void*)apply_kernel<TIn, TC, false, THREADS, TCB, WFR>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      done_f = true;
    }
  }
  if (nan_mode == AGGF_NAN_REPLACE)
    AGGF_LAUNCH((apply_kernel<TIn, TC, true, THREADS, TCB, WFR>), dim3((unsigned)nblocks), dim3(THREADS), lds,
                       stream, (const TIn*)P, T, N, (const TC*)Mx, n_cg, ncb, (TC)nan_fill, p_vec_ok,
                       m_vec_ok, (TC*)out, partials, nan_seen);
  else
    AGGF_LAUNCH((apply_kernel<TIn, TC, false, THREADS, TCB, WFR>), dim3((unsigned)nblocks), dim3(THREADS), lds,
                       stream, (const TIn*)P, T, N, (const TC*)Mx, n_cg, ncb, (TC)0, p_vec_ok,
                       m_vec_ok, (TC*)out, partials, nan_seen);
  AGGF_LAUNCH_OK();
  if (sumsq) {
    AGGF_LAUNCH(sum_partials_kernel, dim3(1), dim3(256), 0, stream, partials, nblocks, sumsq);
    AGGF_LAUNCH_OK();
  }
  return AGGF_OK;
}

// ---------------------------------------------------------------------------
// K3 with a float64 map and result for more than 16 sites (round 4: float64 frames, n_cg > 64, N % 16 == 0; round 5: any
// atom count from 32, float32 frames, 17-64 sites -- see the template parameters below): the tiles of apply_kernel with
// the operand tiles travelling HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4) through a ring of three stages, as
// in K1 -- no VGPR round trip, no ds_write, no conversion or NaN-scan VALU work beside the MFMAs (the register-staged
// kernel spends 7.6 % of its time refilling LDS and holds the clock at ~2.2 GHz; K1 runs at 2.38).
//
// K3 contracts over ATOMS, the direction in which a frame row is contiguous, so a stage (16 atoms) is a 384-byte piece
// of each of the 64 frame rows and a 128-byte piece of each of the 128 map rows.  A DMA instruction fills 1 KiB of LDS
// contiguously but every lane brings its own global address, so the 16-byte chunks may land in any order:
//   P tile: row i = 64 doubles (32 slots of 16 bytes, 24 used); chunk c of row i sits in slot (c + i) & 31.  An operand
//           read (lane = frame l & 15, k' = l >> 4: element 3 (4 kk + k') + d) then touches 32 different bank pairs.
//           One DMA instruction = 2 rows x 32 slots (the 8 unused slots of a row are masked lanes).
//   M tile: row j = 16 doubles (8 slots); chunk c of row j sits in slot (c + (j >> 1)) & 7; one DMA instruction = 8 rows.
// 32 + 16 instructions per stage, 3 per wave (16 waves); stage it + 2 is issued during stage it, two pieces beside the
// first two MFMA groups and the third behind the stage's barrier, which sits in front of the last group's MFMAs (their
// operands are in registers by then).  Rows past T / n_cg are masked lanes: their LDS rows keep stale bytes, which only
// reach outputs that are never stored.  The fused NaN scan looks at the OUTPUT (0 x NaN = NaN reaches every site of that
// frame and component): nan_seen is conservative -- an infinity meeting a zero coefficient sets it too.
// Workgroups b and b + 8 (same XCD under round-robin dispatch; speed only) take the two 128-site blocks of one frame
// block: the second reader of a P tile finds it in that XCD's L2.
constexpr int AD_KA = 16;

__device__ __forceinline__ void ad_wait_vmcnt(int n) {
  // s_waitcnt takes an immediate: dispatch on the (wave-uniform) count
  switch (n < 0 ? 0 : n) {
#define AGGF_VMCNT(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    AGGF_VMCNT(0) AGGF_VMCNT(1) AGGF_VMCNT(2) AGGF_VMCNT(3) AGGF_VMCNT(4) AGGF_VMCNT(5) AGGF_VMCNT(6) AGGF_VMCNT(7)
    AGGF_VMCNT(8) AGGF_VMCNT(9) AGGF_VMCNT(10) AGGF_VMCNT(11) AGGF_VMCNT(12) AGGF_VMCNT(13) AGGF_VMCNT(14) AGGF_VMCNT(15)
#undef AGGF_VMCNT
    default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  }
}

// TF frames x TCW sites per workgroup, TF / 16 x TCW / 32 waves (wave tile 16 frames x 32 sites), NBUF ring slots:
//   <2, 64, 3, ., 128> (n_cg > 128): 16 waves, one workgroup per CU (144 KB of LDS), the second wave pair of every SIMD
//                         half a stage behind; the site blocks of a frame block stay in lock-step and share P through the L2;
//   <1, 32, 2, ., 128> (n_cg <= 128): 8 waves, 64 KB of LDS -- two INDEPENDENT workgroups per CU as in K1: one's barrier
//                         and operand-read phase is covered by the other's MFMAs;
//   <1, 64, ., ., 64>  (n_cg <= 64, round 5): 8 waves on 64 frames x 64 sites, two workgroups per CU.
// TS (round 5): the type of the frames.  float: a stage's piece of a frame row is 48 floats = 12 chunks of 16 bytes in
// a row of 16 slots, chunk c of row i in slot (c + i) & 15 (one DMA instruction = 4 rows); an operand read (element
// e = 3 (4 kk + k') + d of row i: float ((e >> 2) + i & 15) * 4 + (e & 3)) touches 64 different banks -- the 16 rows of
// a k' group give the 16 slots, and 3 k' mod 4 gives the four k' groups four different floats of a slot; the operand
// is widened in a register (exact), the map and the result are float64 as in the reference (map/core.py:219-240).
// c3 (n_cg 256), same box: <2, 64, 3> 105.2-105.3 ms, FETCH x 2 109.5 GB; <1, 32, 2> 103.2-103.9 ms (MFMA-busy 0.86 against
// 0.83; the dense variant's two applies 194.8 -> 189.6 ms) but FETCH x 2 145.7 GB: its pairs drift apart.
// With two slots a stage has exactly one stage time to land: spreading its four pieces over the NEXT stage's MFMA groups
// instead of issuing them right behind the barrier costs 6 % (109.5 ms) -- the kernel is sensitive to landing latency,
// and a third slot does not fit twice into 160 KB.
template <int MODE, int TF, int NBUF, typename TS = double, int NWS = 4, int NCT = 2, bool RAGGED = false>
__global__ __launch_bounds__(TF * NWS * 4, TF * NWS * 4 >= 1024 ? 4 : 2) void apply_dma_kernel(
    const TS* __restrict__ P, int64_t T, int32_t N, const double* __restrict__ Mx, int32_t n_cg, int32_t ncb, int64_t nfb,
    double* __restrict__ out, double* __restrict__ sumsq_partials, int32_t* __restrict__ nan_seen) {
  using MF = Mfma<double>;
  static_assert(NBUF == 3 || MODE == 1, "two slots: every piece behind the barrier, no stagger");
  constexpr bool F32 = sizeof(TS) == 4;
  constexpr int TCW = NWS * NCT * 16;             // sites per workgroup: NWS waves along the sites, NCT 16-site tiles each
  constexpr int NWF = TF / 16, NW = NWF * NWS;    // waves along the frames x along the sites
  constexpr int RPP = F32 ? 4 : 2;                // frame rows per DMA piece (a row = 64 elements of TS in LDS)
  constexpr int SLOTS = 64 / RPP;                 // 16-byte slots per row: 32 (24 used) / 16 (12 used)
  constexpr int USED = F32 ? 12 : 24, EPC = F32 ? 4 : 2;  // chunks of a stage's row piece, elements per chunk
  constexpr int PP = (TF / RPP) / NW, MP = (TCW / 8 + NW - 1) / NW;  // DMA pieces per wave and stage: P, M (8 rows each)
  static_assert(PP >= 1 && PP * NW * RPP == TF && TCW % 8 == 0, "piece split");
  constexpr int NPIECE = PP + MP;
  static_assert(NPIECE <= 16, "ad_wait_vmcnt");
  constexpr int P_BYTES = TF * 64 * (int)sizeof(TS), BUF_BYTES = P_BYTES + TCW * 16 * 8;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // block -> (frame block, site block): 8 consecutive frame blocks x ncb site blocks per group of 8 ncb workgroups,
  // the site blocks of one frame block 8 apart
  const int64_t b = blockIdx.x;
  const int64_t grp = b / (8 * ncb);
  const int rem = (int)(b - grp * 8 * ncb);
  const int cb = rem >> 3;
  const int64_t fb = grp * 8 + (rem & 7);
  if (fb >= nfb) return;  // (the grid is padded to whole groups; uniform: no barrier has been reached)
  const int64_t t0 = fb * TF;
  const int c0 = cb * TCW;
  const int64_t rowP = (int64_t)N * 3;
  // RAGGED (N % 16 != 0; its own instantiation: the extra stage form costs the whole-stage kernels registers -- 7 spilled
  // in the 16-wave form): the LAST stage is the window of atoms N - 16 .. N - 1 -- it overlaps the stage before it by 16 - N % 16
  // atoms, so every byte it reads lies inside the rows -- and the lanes of the overlapped atoms feed zeros to the MFMAs
  // in place of what they read (a select, not a product: an infinity counted once stays an infinity)
  const int n_stage = (N + AD_KA - 1) / AD_KA;
  const int tail = RAGGED ? N % AD_KA : 0;

  // this wave's DMA pieces: PP pieces of the P tile (RPP rows each), then MP pieces of the M tile (8 rows each)
  const char* gsrc[NPIECE];
  int lbase[NPIECE];
  bool ok[NPIECE];
#pragma unroll
  for (int q = 0; q < PP; ++q) {
    const int pp = wave + NW * q;
    const int row = RPP * pp + lane / SLOTS, slot = lane % SLOTS;
    const int chunk = (slot - (row & 15)) & (SLOTS - 1);
    ok[q] = chunk < USED && t0 + row < T;
    gsrc[q] = reinterpret_cast<const char*>(P + (t0 + row) * rowP + chunk * EPC);
    lbase[q] = pp * 1024;
  }
#pragma unroll
  for (int q = 0; q < MP; ++q) {
    const int mp = wave + NW * q;
    const int row = 8 * mp + (lane >> 3), slot = lane & 7;
    const int chunk = (slot - (row >> 1)) & 7;
    ok[PP + q] = mp < TCW / 8 && c0 + row < n_cg;
    gsrc[PP + q] = reinterpret_cast<const char*>(Mx + (int64_t)(c0 + row) * N + chunk * 2);
    lbase[PP + q] = P_BYTES + mp * 1024;
  }
  // pieces with no active lane are skipped by the hardware and do not count in vmcnt: wave-uniform tallies
  int n_first = 0, n_all = 0;  // (n_first: the two pieces MODE 0 issues in front of the stage barrier)
#pragma unroll
  for (int q = 0; q < NPIECE; ++q) {
    const int act = __any(ok[q]) ? 1 : 0;
    n_all += act;
    if (q < 2) n_first += act;
  }
  n_first = __builtin_amdgcn_readfirstlane(n_first);
  n_all = __builtin_amdgcn_readfirstlane(n_all);
  auto issue_piece = [&](int s, int q) {
    if (ok[q]) {
      const char* src = gsrc[q] + (int64_t)s * (q >= PP ? AD_KA * 8 : AD_KA * 3 * (int)sizeof(TS));
      if constexpr (RAGGED) {  // (wave-uniform)
        if (s == n_stage - 1) src = gsrc[q] + (int64_t)(N - AD_KA) * (q >= PP ? 8 : 3 * (int)sizeof(TS));
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(smem_raw + (s % NBUF) * BUF_BYTES + lbase[q]),
                                       16, 0, 0);
    }
  };

  // MFMA operand offsets inside a stage buffer: elements of TS (P tile), doubles behind P_BYTES (M tile)
  const int wf = wave % NWF, wc = wave / NWF;
  int offA[4][3], offB[4];
  {
    const int row = 16 * wf + (lane & 15), kq = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const int e = 12 * kk + 3 * kq + d;
        offA[kk][d] = F32 ? row * 64 + (((e >> 2) + (row & 15)) & 15) * 4 + (e & 3)
                          : row * 64 + (((e >> 1) + (row & 15)) & 31) * 2 + (e & 1);
      }
    const int j = 16 * NCT * wc + (lane & 15);  // (the n-th tile of the wave: j + 16 n -- the rotation (j >> 1) & 7 is the same)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int e = 4 * kk + kq;
      offB[kk] = j * 16 + (((e >> 1) + (j >> 1)) & 7) * 2 + (e & 1);
    }
  }

  f64x4 acc[NCT][3];
#pragma unroll
  for (int n = 0; n < NCT; ++n)
#pragma unroll
    for (int d = 0; d < 3; ++d) acc[n][d] = acc_zero<double>();

#pragma unroll
  for (int q = 0; q < NPIECE; ++q) issue_piece(0, q);
  if (n_stage > 1) {
#pragma unroll
    for (int q = 0; q < NPIECE; ++q) issue_piece(1, q);
  }
  ad_wait_vmcnt(n_stage > 1 ? n_all : 0);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // MODE 0: pieces 0, 1 of stage it + 2 beside the first two MFMA groups, piece 2 behind the barrier (in front of the
  //         last group's MFMAs).  MODE 1: all pieces behind the barrier.  MODE 2 (shipped for n_cg > 128): as 1, and waves
  //         NW/2.. (the second pair of every SIMD) meet the barrier two groups EARLIER in their own stream -- they run half a
  //         stage behind waves 0..7, so the two pairs of a SIMD are never in their operand-read / barrier phase together
  //         (a slot is refilled only behind the barrier after which nobody reads it: three slots still suffice).
  //         c3, same box, two runs each: MODE 0 107.1 / 106.6 ms, MODE 1 106.0 / 106.5, MODE 2 104.9 / 105.5 (the
  //         register-staged kernel: 105.4-105.8) -- the staging was never the limiter; what the DMA form buys is the
  //         single fetch of P: rocprofv3 FETCH_SIZE x 2 = 109.5 GB per launch against 202.3 GB (98.3 GB algorithmic).
  const int bar_kk = (MODE == 2 && wave >= NW / 2) ? 1 : 3;
  // One stage.  SLOT (the ring slot, a compile-time constant in the three-slot form below) puts the slot's base into the
  // immediate offset of the LDS reads.  The operand offsets are lane-dependent in a way no single base covers (the
  // rotated chunks), i.e. 16 address registers; with the slot base added at run time every read cost a vector add --
  // 20 of the 1.7 non-MFMA vector instructions per MFMA of this kernel (rocprofv3: SQ_INSTS_VALU / MFMA = 2.7 against
  // 1.6 in K1), and float64 MFMAs share their SIMD's vector datapath with them.
  auto stage = [&](int it, const char* buf, auto masked_c) {
    constexpr bool MASKED = decltype(masked_c)::value;  // the ragged last stage: lanes of atoms already counted feed zeros
    const bool issue_now = it + 2 < n_stage;
    const TS* bufP = reinterpret_cast<const TS*>(buf);
    const double* bufM = reinterpret_cast<const double*>(buf + P_BYTES);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double a[3], bq[NCT];
#pragma unroll
      for (int d = 0; d < 3; ++d) a[d] = (double)bufP[offA[kk][d]];
#pragma unroll
      for (int n = 0; n < NCT; ++n) bq[n] = bufM[offB[kk] + 256 * n];
      if constexpr (MASKED) {
        const bool fresh = 4 * kk + (lane >> 4) >= AD_KA - tail;  // window position of this lane's atom
#pragma unroll
        for (int d = 0; d < 3; ++d) a[d] = fresh ? a[d] : 0.0;
#pragma unroll
        for (int n = 0; n < NCT; ++n) bq[n] = fresh ? bq[n] : 0.0;
      }
      if (MODE == 0 && issue_now && kk < 2) issue_piece(it + 2, kk);
      if (kk == bar_kk) {
        // stage it + 1 has landed (this wave's pieces; pieces of stage it + 2 issued above may stay in flight), every
        // operand of stage it that the early group needs is in registers
        ad_wait_vmcnt(MODE == 0 && issue_now ? n_first : 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (issue_now) {
#pragma unroll
          for (int q = (MODE == 0 ? 2 : 0); q < NPIECE; ++q) issue_piece(it + 2, q);
        }
      }
#pragma unroll
      for (int n = 0; n < NCT; ++n)
#pragma unroll
        for (int d = 0; d < 3; ++d) acc[n][d] = MF::mma(a[d], bq[n], acc[n][d]);
    }
  };
  using plain = std::false_type;
  using masked = std::true_type;
  const int n_plain = RAGGED ? n_stage - 1 : n_stage;  // stages of whole 16-atom blocks
  if constexpr (NBUF == 3) {
    // three stages per pass: the slot of each is a constant
    int it = 0;
    for (; it + 2 < n_plain; it += 3) {
      stage(it, smem_raw, plain{});
      stage(it + 1, smem_raw + BUF_BYTES, plain{});
      stage(it + 2, smem_raw + 2 * BUF_BYTES, plain{});
    }
    if constexpr (!RAGGED) {
      if (it < n_stage) stage(it, smem_raw, plain{});
      if (it + 1 < n_stage) stage(it + 1, smem_raw + BUF_BYTES, plain{});
    } else {
      // the rest of the plain stages, then the ragged one in the slot that follows
      const int left = n_plain - it;  // 0, 1 or 2
      if (left >= 1) stage(it, smem_raw, plain{});
      if (left >= 2) stage(it + 1, smem_raw + BUF_BYTES, plain{});
      if (left == 0) stage(n_plain, smem_raw, masked{});
      else if (left == 1) stage(n_plain, smem_raw + BUF_BYTES, masked{});
      else stage(n_plain, smem_raw + 2 * BUF_BYTES, masked{});
    }
  } else {
    static_assert(NBUF == 2, "ring of two or three slots");
    int it = 0;
    for (; it + 1 < n_plain; it += 2) {
      stage(it, smem_raw, plain{});
      stage(it + 1, smem_raw + BUF_BYTES, plain{});
    }
    if constexpr (!RAGGED) {
      if (it < n_stage) stage(it, smem_raw, plain{});
    } else {
      const int left = n_plain - it;  // 0 or 1
      if (left >= 1) stage(it, smem_raw, plain{});
      if (left == 0) stage(n_plain, smem_raw, masked{});
      else stage(n_plain, smem_raw + BUF_BYTES, masked{});
    }
  }

  double ss = 0.0;
  bool saw_nan = false;
#pragma unroll
  for (int n = 0; n < NCT; ++n) {
    const int c = c0 + (wc * NCT + n) * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t t = t0 + 16 * wf + MF::row(lane, r);
      if (t < T && c < n_cg) {
        double* o = out + (t * n_cg + c) * 3;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const double v = acc[n][d][r];
          o[d] = v;
          saw_nan |= (v != v);
          ss += v * v;
        }
      }
    }
  }
  if (nan_seen && __any(saw_nan) && lane == 0) atomicOr(nan_seen, 1);
  if (sumsq_partials) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
    __shared__ double wsum[NW];
    if (lane == 0) wsum[wave] = ss;
    __syncthreads();
    if (tid == 0) {
      double tot = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) tot += wsum[w];
      sumsq_partials[fb * ncb + cb] = tot;
    }
  }
}

template <int MODE, int TF, int NBUF, typename TS = double, int NWS = 4, int NCT = 2, bool RAGGED = false>
static int apply_dma_launch_r(const TS* P, int64_t T, int32_t N, const double* Mx, int32_t n_cg, double* out,
                            double* sumsq, int32_t* nan_seen, void* ws, size_t ws_bytes, hipStream_t stream) {
  constexpr int TCW = NWS * NCT * 16;
  const int ncb = (int)ceil_div(n_cg, TCW);
  const int64_t nfb = ceil_div(T, TF);
  const int64_t nblocks = round_up(nfb, 8) * ncb;
  if (nblocks > 0x7fffffffLL) return fail(AGGF_ERR_ARG, "apply grid too large");
  double* partials = nullptr;
  if (sumsq) {
    if (!ws || ws_bytes < (size_t)(nfb * ncb) * sizeof(double))
      return fail(AGGF_ERR_WORKSPACE, "aggf_linearmap_apply: workspace too small for sumsq");
    partials = reinterpret_cast<double*>(ws);
  }
  // <64, 3, double, 128>: 144 KB; <32, 2, double, 128>: 64 KB; <64, 3, float, 64>: 72 KB
  constexpr size_t lds = (size_t)NBUF * (TF * 64 * sizeof(TS) + TCW * 16 * sizeof(double));
  static thread_local PerDeviceOnce once;
  bool& done = *once.flag();
  if (!done) {
    AGGF_HIP_OK(hipFuncSetAttribute((const void*)apply_dma_kernel<MODE, TF, NBUF, TS, NWS, NCT, RAGGED>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    done = true;
  }
  AGGF_LAUNCH((apply_dma_kernel<MODE, TF, NBUF, TS, NWS, NCT, RAGGED>), dim3((unsigned)nblocks), dim3(TF * NWS * 4), lds, stream, P, T, N,
              Mx, n_cg, ncb, nfb, out, partials, nan_seen);
  AGGF_LAUNCH_OK();
  if (sumsq) {
    AGGF_LAUNCH(sum_partials_kernel, dim3(1), dim3(256), 0, stream, partials, nfb * ncb, sumsq);
    AGGF_LAUNCH_OK();
  }
  return AGGF_OK;
}

template <int MODE, int TF, int NBUF, typename TS = double, int NWS = 4, int NCT = 2>
static int apply_dma_launch(const TS* P, int64_t T, int32_t N, const double* Mx, int32_t n_cg, double* out,
                            double* sumsq, int32_t* nan_seen, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (N % AD_KA != 0)
    return apply_dma_launch_r<MODE, TF, NBUF, TS, NWS, NCT, true>(P, T, N, Mx, n_cg, out, sumsq, nan_seen, ws, ws_bytes, stream);
  return apply_dma_launch_r<MODE, TF, NBUF, TS, NWS, NCT, false>(P, T, N, Mx, n_cg, out, sumsq, nan_seen, ws, ws_bytes, stream);
}

template <typename TIn, typename TC>
static int apply_typed(const void* P, int64_t T, int32_t N, const void* Mx, int32_t n_cg,
                       int nan_mode, double nan_fill, void* out, double* sumsq, int32_t* nan_seen,
                       void* ws, size_t ws_bytes, hipStream_t stream) {
  // 64 frames x 128 sites with 16 waves (4 per SIMD) when n_cg > 64, else 64 x 64 with 4 waves.  Measured and dropped
  // (profiles/r04_pruned_variants.patch): 8 waves on 64 x 128 (104.5 against 102.4 ms at c3), 32 x 128 with two
  // workgroups per CU (103 ms), 64 x 256 at 4 waves per SIMD (spills: 282 ms), 32 x 256 at 2 waves per SIMD (P read
  // once, 232 registers: 109.8 against 108.4 ms).
  // few sites: the streaming kernel (frames of a stage must fit AS_NV 16-byte loads per thread; P at any
  // element-aligned address)
  const int small_kb = apply_small_frames<TIn>(N);
  if (n_cg <= 16 && small_kb > 0 && T >= 64) {
    if (small_kb == 40)
      return apply_small_launch<TIn, TC, 40, 8>(P, T, N, Mx, n_cg, nan_mode, nan_fill, out, sumsq, nan_seen, ws, ws_bytes, stream);
    if (small_kb == 20)
      return apply_small_launch<TIn, TC, 20, 4>(P, T, N, Mx, n_cg, nan_mode, nan_fill, out, sumsq, nan_seen, ws, ws_bytes, stream);
    if (small_kb == 12)
      return apply_small_launch<TIn, TC, 12, 4>(P, T, N, Mx, n_cg, nan_mode, nan_fill, out, sumsq, nan_seen, ws, ws_bytes, stream);
    return apply_small_launch<TIn, TC, 8, 2>(P, T, N, Mx, n_cg, nan_mode, nan_fill, out, sumsq, nan_seen, ws, ws_bytes, stream);
  }
  // LDS-DMA form: float64 map and result, frames float64 or float32 (widened out of LDS), at least two 16-atom stages
  // (an atom count that is not a multiple of 16: the RAGGED instantiations)
  constexpr bool dma_types = std::is_same<TC, double>::value;
  const char* k3_route = getenv("AGGF_APPLY_ROUTE");  // measurement: "reg" = the register-staged kernel everywhere
  // (P and the map at any element-aligned address: the LDS-DMA takes every byte address)
  const bool dma_ok = dma_types && nan_mode != AGGF_NAN_REPLACE && N >= 2 * AD_KA && !(k3_route && k3_route[0] == 'r');
  if (n_cg > 64) {
    if constexpr (dma_types) {
      if (dma_ok) {
        // More than one 128-site block: the 16-wave form, whose site blocks of a frame block run in lock-step on one XCD
        // and share the P tile through its L2 (FETCH x 2 = 109.5 GB per launch at c3; the two-workgroup form drifts:
        // 145.7 GB).  A single site block has no second reader: the faster two-workgroup form.
        if (n_cg > 128)
          return apply_dma_launch<2, 64, 3, TIn, 4, 2>((const TIn*)P, T, N, (const double*)Mx, n_cg, (double*)out, sumsq,
                                                      nan_seen, ws, ws_bytes, stream);
        return apply_dma_launch<1, 32, 2, TIn, 4, 2>((const TIn*)P, T, N, (const double*)Mx, n_cg, (double*)out, sumsq,
                                                    nan_seen, ws, ws_bytes, stream);
      }
    }
    return apply_launch<TIn, TC, 1024, 128>(P, T, N, Mx, n_cg, nan_mode, nan_fill, out, sumsq, nan_seen, ws,
                                            ws_bytes, stream);
  }
  if constexpr (dma_types) {
    if (dma_ok && n_cg > 16) {
      // 17-64 sites (round 5): the same ring with narrower site tiles.  Against the register-staged kernel, same box:
      // 64 sites on 1024 atoms x 1e5 frames (BASELINE config 2's apply) float32 0.84-0.85 -> 0.63-0.70 ms, float64
      // 0.80-0.83 -> 0.68 ms; 35 sites on 576 atoms 2.97 -> 2.67 ms (float32, 1e6 frames) / 1.75 -> 1.54 (float64, 5e5);
      // 20 sites on 320 atoms 2.50 -> 2.09 / 1.54 -> 1.45.  Measured and not kept: one wave column of four site tiles for
      // 64 sites (0.71-0.78 ms), 32-frame workgroups for the 32- and 48-site tiles (2.30 / 2.78 ms float32), the early
      // pieces of MODE 0 for them (2.26 / 2.67).
#define AGGF_DMA(M_, TF_, NB_, NWS_, NCT_)                                                                              \
  return apply_dma_launch<M_, TF_, NB_, TIn, NWS_, NCT_>((const TIn*)P, T, N, (const double*)Mx, n_cg, (double*)out, sumsq, \
                                                         nan_seen, ws, ws_bytes, stream)
      if (n_cg > 48) {
        // 32 frames x 64 sites, 4 waves, three workgroups per CU: float32 frames with three slots and the first two
        // pieces beside the MFMA groups, float64 with two slots
        if constexpr (sizeof(TIn) == 4) AGGF_DMA(0, 32, 3, 2, 2);
        else AGGF_DMA(1, 32, 2, 2, 2);
      }
      if (n_cg > 32) AGGF_DMA(1, 64, 2, 1, 3);  // 64 frames x 48 sites, 4 waves of 16 frames x 48 sites
      AGGF_DMA(1, 64, 2, 1, 2);                 // 64 frames x 32 sites
#undef AGGF_DMA
    }
  }
  // 17-48 sites: a 32- or 48-site tile -- the 64-site tile multiplies its empty 16-site column tiles all the same
  // (20 sites on 320 atoms of float32 frames, float64 map: 6.5 ms for 12 GB, 0.75 of the fp64 MFMA peak in EXECUTED flops)
  if (n_cg <= 32)
    return apply_launch<TIn, TC, 256, 32>(P, T, N, Mx, n_cg, nan_mode, nan_fill, out, sumsq, nan_seen, ws, ws_bytes, stream);
  if (n_cg <= 48)
    return apply_launch<TIn, TC, 256, 48>(P, T, N, Mx, n_cg, nan_mode, nan_fill, out, sumsq, nan_seen, ws, ws_bytes, stream);
  return apply_launch<TIn, TC, 256, 64>(P, T, N, Mx, n_cg, nan_mode, nan_fill, out, sumsq, nan_seen, ws,
                                        ws_bytes, stream);
}

}  // namespace aggf

using namespace aggf;

extern "C" size_t aggf_linearmap_apply_workspace_bytes(int64_t T, int32_t N, int32_t n_cg) {
  (void)N;
  if (T <= 0 || n_cg <= 0) return 256;
  // one partial per workgroup: tile kernel ceil(T/32) x ceil(n_cg/64) at most, streaming kernel 2 per CU
  return (size_t)round_up(ceil_div(T, 32) * ceil_div(n_cg, 64) * 8 + 2 * 8 * (int64_t)device_cu_count(), 256);
}

extern "C" int aggf_linearmap_apply(const void* P, int64_t T, int32_t N, int in_dtype,
                                    const void* Mx, int32_t n_cg, int out_dtype, int nan_mode,
                                    double nan_fill, void* out, double* sumsq, int32_t* nan_seen,
                                    void* ws, size_t ws_bytes, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!P || !Mx || !out) return fail(AGGF_ERR_ARG, "aggf_linearmap_apply: NULL pointer");
  if (T <= 0 || N <= 0 || n_cg <= 0) return fail(AGGF_ERR_ARG, "aggf_linearmap_apply: empty problem");
  if (nan_mode != AGGF_NAN_PROPAGATE && nan_mode != AGGF_NAN_REPLACE)
    return fail(AGGF_ERR_ARG, "aggf_linearmap_apply: bad nan_mode");
  if (in_dtype == AGGF_F64 && out_dtype == AGGF_F64)
    return apply_typed<double, double>(P, T, N, Mx, n_cg, nan_mode, nan_fill, out, sumsq, nan_seen, ws, ws_bytes, stream);
  if (in_dtype == AGGF_F32 && out_dtype == AGGF_F64)
    return apply_typed<float, double>(P, T, N, Mx, n_cg, nan_mode, nan_fill, out, sumsq, nan_seen, ws, ws_bytes, stream);
  if (in_dtype == AGGF_F32 && out_dtype == AGGF_F32)
    return apply_typed<float, float>(P, T, N, Mx, n_cg, nan_mode, nan_fill, out, sumsq, nan_seen, ws, ws_bytes, stream);
  if (in_dtype == AGGF_F64 && out_dtype == AGGF_F32)
    return apply_typed<double, float>(P, T, N, Mx, n_cg, nan_mode, nan_fill, out, sumsq, nan_seen, ws, ws_bytes, stream);
  return fail(AGGF_ERR_ARG, "aggf_linearmap_apply: unsupported dtype combination (in %d, out %d)",
              in_dtype, out_dtype);
}

extern "C" int aggf_slice_gather(const void* P, int64_t T, int32_t N, int in_dtype,
                                 const int32_t* idx, int32_t n_cg, int out_dtype, void* out,
                                 int32_t* nan_seen, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!P || !idx || !out) return fail(AGGF_ERR_ARG, "aggf_slice_gather: NULL pointer");
  if (T <= 0 || N <= 0 || n_cg <= 0) return fail(AGGF_ERR_ARG, "aggf_slice_gather: empty problem");
  int lanes = 256;  // threads per frame: the smallest power of two that covers the frame's 3 n_cg elements
  while (lanes > 1 && lanes / 2 >= n_cg * 3) lanes /= 2;
  const int64_t gy = ceil_div((int64_t)n_cg * 3, lanes);
  if (gy > 65535) return fail(AGGF_ERR_ARG, "aggf_slice_gather: too many sites");
  // enough workgroups to keep the memory system busy on their own (8 frames in flight per thread), few enough that
  // the kernel stays a guest beside a compute kernel on another stream; AGGF_GATHER_WGS overrides (benchmarks)
  static const char* force = getenv("AGGF_GATHER_WGS");
  int64_t gx = force ? atoll(force) : 4 * (int64_t)device_cu_count() / gy;
  const int64_t need = ceil_div(T, (int64_t)SG_UF * (256 / lanes));
  if (gx > need) gx = need;
  if (gx < 1) gx = 1;
  const dim3 grid((unsigned)gx, (unsigned)gy), block(256);
  if (in_dtype == AGGF_F64 && out_dtype == AGGF_F64)
    AGGF_LAUNCH((slice_gather_kernel<double, double>), grid, block, 0, stream, (const double*)P, T, N, idx, n_cg, (double*)out, nan_seen, lanes);
  else if (in_dtype == AGGF_F32 && out_dtype == AGGF_F64)
    AGGF_LAUNCH((slice_gather_kernel<float, double>), grid, block, 0, stream, (const float*)P, T, N, idx, n_cg, (double*)out, nan_seen, lanes);
  else if (in_dtype == AGGF_F32 && out_dtype == AGGF_F32)
    AGGF_LAUNCH((slice_gather_kernel<float, float>), grid, block, 0, stream, (const float*)P, T, N, idx, n_cg, (float*)out, nan_seen, lanes);
  else if (in_dtype == AGGF_F64 && out_dtype == AGGF_F32)
    AGGF_LAUNCH((slice_gather_kernel<double, float>), grid, block, 0, stream, (const double*)P, T, N, idx, n_cg, (float*)out, nan_seen, lanes);
  else
    return fail(AGGF_ERR_ARG, "aggf_slice_gather: unsupported dtype combination");
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}
