// K1: Gram / normal matrix of (constraint-reduced) forces on MFMA.
//
// Replaces qp/qplinear.py:66-71 of the reference (qp_form copy, `@ con_mat`,
// `reg_mat.T @ reg_mat`).  Three kernels:
//   pack_groups_kernel   (T,N,3) -> (T,n_pad,3): constraint-group column sums, dtype
//                        conversion and zero padding to a multiple of 128 columns
//                        (skipped when the input already has that shape and dtype);
//   gram_tile_kernel     split-K SYRK: one 128x128 upper-triangle tile x one frame
//                        range per workgroup, MFMA 16x16x4 (f64 or f32), frame rows
//                        staged through LDS exactly as they lie in HBM -- the
//                        (t,d)-major / atom-minor transpose the reference pays a
//                        full copy for (qp_form) is done by the LDS read pattern;
//   gram_reduce_kernel   fixed-order fp64 sum of the split-K slabs into G (both
//                        triangles) -- no float atomics, bit-reproducible.
#include <stdlib.h>

#include "aggf_common.h"

namespace aggf {

constexpr int TILE = 128;         // output tile edge (reduced atoms)
constexpr int ROW_ELEMS = TILE * 3;
constexpr int ROW_PAD = 16;       // row stride == 128 B (f64) / 64 B (f32) mod bank period:
                                  // the two k-rows a 32-lane half reads hit disjoint banks
constexpr int ROW_STRIDE = ROW_ELEMS + ROW_PAD;
constexpr int GRAM_THREADS = 256;

template <typename T>
struct GramCfg;
template <>
struct GramCfg<double> {
  static constexpr int KB = 4;  // frames per LDS stage
};
template <>
struct GramCfg<float> {
  static constexpr int KB = 8;
};

// ---------------------------------------------------------------------------
// One workgroup = 768 consecutive elements (256 reduced columns x xyz) of the packed row, for a strided set of
// frames: a thread owns three fixed output elements, so the member atoms of their constraint groups -- the chain
// grp_ptr -> grp_atoms -> element offset, three dependent loads per element and frame in the first version of this
// kernel -- are looked up ONCE, outside the frame loop, and held in registers (up to PK_FAST members; larger groups
// finish through the CSR arrays).  Inside the loop every load is independent and neighbouring threads read
// neighbouring addresses (xyz of one atom, then the next group's atoms); workgroups with the same blockIdx.y walk
// the same frames, so a frame row is fetched from HBM once.  Sums run in CSR order like the column sum of `@ con_mat`.
constexpr int PK_FAST = 4;
constexpr int PK_ELEMS = 3;  // output elements per thread

// NT: non-temporal loads and stores -- the overlapped pipeline packs the next chunk while the tile kernel lives on
// the panels it shares through the L2
template <typename TIn, typename TC, bool NT = false>
__global__ __launch_bounds__(256) void pack_groups_kernel(
    const TIn* __restrict__ F, int64_t T, int32_t N, const int32_t* __restrict__ grp_ptr,
    const int32_t* __restrict__ grp_atoms, int32_t n_red, int32_t n_pad, TC* __restrict__ out) {
  const int64_t row_in = (int64_t)N * 3;
  const int64_t row_out = (int64_t)n_pad * 3;
  int off[PK_ELEMS][PK_FAST], cnt[PK_ELEMS], first[PK_ELEMS], eo[PK_ELEMS];
  int max_cnt = 0;
#pragma unroll
  for (int q = 0; q < PK_ELEMS; ++q) {
    const int e = blockIdx.x * (256 * PK_ELEMS) + q * 256 + threadIdx.x;
    eo[q] = e < (int)row_out ? e : -1;
    const int g = e / 3, d = e - 3 * g;
    int b = 0, n = 0;
    if (e < (int)row_out && g < n_red) {
      b = grp_ptr ? grp_ptr[g] : g;
      n = grp_ptr ? grp_ptr[g + 1] - b : 1;
    }
    cnt[q] = n;
    first[q] = b;
    max_cnt = n > max_cnt ? n : max_cnt;
#pragma unroll
    for (int j = 0; j < PK_FAST; ++j) off[q][j] = j < n ? 3 * (grp_atoms ? grp_atoms[b + j] : b + j) + d : -1;
  }
  const bool big = __syncthreads_or(max_cnt > PK_FAST);
  for (int64_t t = blockIdx.y; t < T; t += gridDim.y) {
    const TIn* src = F + t * row_in;
    TC* dst = out + t * row_out;
    TC acc[PK_ELEMS];
#pragma unroll
    for (int q = 0; q < PK_ELEMS; ++q) {
      TC a = 0;
#pragma unroll
      for (int j = 0; j < PK_FAST; ++j)
        if (off[q][j] >= 0) a += (TC)(NT ? __builtin_nontemporal_load(src + off[q][j]) : src[off[q][j]]);
      acc[q] = a;
    }
    if (big) {
#pragma unroll
      for (int q = 0; q < PK_ELEMS; ++q) {
        const int d = off[q][0] % 3;
        for (int j = PK_FAST; j < cnt[q]; ++j) acc[q] += (TC)src[(int64_t)grp_atoms[first[q] + j] * 3 + d];
      }
    }
#pragma unroll
    for (int q = 0; q < PK_ELEMS; ++q)
      if (eo[q] >= 0) {
        if (NT) __builtin_nontemporal_store(acc[q], dst + eo[q]); else dst[eo[q]] = acc[q];
      }
  }
}

// ---------------------------------------------------------------------------
// X: (rows, ld) with ld = 3*n_pad elements, n_pad % 128 == 0, 16-byte aligned rows.
// grid.x = ksplit * n_tiles, tile fastest (co-running workgroups share a frame range).
// ABL (ablation switches, tools/gram_ablate.hip only; the library always uses 0):
//   1 = no global loads after the first stage, 2 = additionally no LDS refill/barrier,
//   3 = additionally operands read from LDS once (MFMA only).
template <typename T, int ABL = 0, bool STAGGER = true>
__global__ __launch_bounds__(GRAM_THREADS, 2) void gram_tile_kernel(
    const T* __restrict__ X, int64_t n_rows, int64_t ld, int32_t nt1, int32_t n_tiles,
    int64_t frames_per_split, T* __restrict__ slabs) {
  using M = Mfma<T>;
  using acc_t = typename M::acc_t;
  using vec_t = typename Vec16<T>::type;
  constexpr int KB = GramCfg<T>::KB;
  constexpr int VN = Vec16<T>::N;
  constexpr int CH_PER_ROW = ROW_ELEMS / VN;          // 16-byte chunks per panel row
  constexpr int CH_PER_PANEL = CH_PER_ROW * KB;       // 768 for both dtypes
  constexpr int CH_PER_THREAD = CH_PER_PANEL / GRAM_THREADS;  // 3
  static_assert(CH_PER_PANEL % GRAM_THREADS == 0, "staging split");
  constexpr int PANEL_ELEMS = KB * ROW_STRIDE;
  constexpr int BUF_ELEMS = 2 * PANEL_ELEMS;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int b = blockIdx.x;
  const int ks = b / n_tiles;
  int tile = b - ks * n_tiles;
  // tile -> (ti, tj), upper triangle, row-major
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
  const bool diag = (ti == tj);
  const int tile_lin = b - ks * n_tiles;

  // De-phase the two workgroups that share a CU.  All workgroups do identical work, so two
  // co-resident ones that start together stay in lock-step and hit their refill/barrier
  // phases at the same time, leaving the MFMA pipe idle (tools/gram_ablate.hip: 12 %).  A
  // pseudo-random start delay of up to one stage (conserved for the life of the pair) makes
  // one of them compute while the other refills.
  if (STAGGER) {
    const unsigned h = ((unsigned)b * 2654435761u) >> 25;  // 0..127
    for (unsigned i = 0; i < h; ++i) __builtin_amdgcn_s_sleep(1);  // 64 clocks each
  }

  const int64_t t_begin = (int64_t)ks * frames_per_split;
  int64_t t_end = t_begin + frames_per_split;
  if (t_end > n_rows) t_end = n_rows;
  const int n_it = t_begin < t_end ? (int)((t_end - t_begin + KB - 1) / KB) : 0;

  // per-thread staging coordinates (same for every stage)
  int st_row[CH_PER_THREAD], st_col[CH_PER_THREAD];
#pragma unroll
  for (int q = 0; q < CH_PER_THREAD; ++q) {
    const int c = tid + q * GRAM_THREADS;
    st_row[q] = c / CH_PER_ROW;
    st_col[q] = (c - st_row[q] * CH_PER_ROW) * VN;
  }
  const T* gA = X + (int64_t)ti * ROW_ELEMS;
  const T* gB = X + (int64_t)tj * ROW_ELEMS;

  vec_t ra[CH_PER_THREAD], rb[CH_PER_THREAD];
  auto load_stage = [&](int it) {
    const int64_t t0 = t_begin + (int64_t)it * KB;
#pragma unroll
    for (int q = 0; q < CH_PER_THREAD; ++q) {
      const int64_t t = t0 + st_row[q];
      vec_t z;
#pragma unroll
      for (int e = 0; e < VN; ++e) z[e] = 0;
      ra[q] = z;
      rb[q] = z;
      if (t < t_end) {
        ra[q] = *reinterpret_cast<const vec_t*>(gA + t * ld + st_col[q]);
        if (!diag) rb[q] = *reinterpret_cast<const vec_t*>(gB + t * ld + st_col[q]);
      }
    }
  };
  auto store_stage = [&](int buf) {
    T* pa = smem + buf * BUF_ELEMS;
    T* pb = pa + PANEL_ELEMS;
#pragma unroll
    for (int q = 0; q < CH_PER_THREAD; ++q) {
      *reinterpret_cast<vec_t*>(pa + st_row[q] * ROW_STRIDE + st_col[q]) = ra[q];
      if (!diag) *reinterpret_cast<vec_t*>(pb + st_row[q] * ROW_STRIDE + st_col[q]) = rb[q];
    }
  };

  acc_t acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = acc_zero<T>();

  // lane-constant LDS read offsets: k-row (lane>>4), column 3*(wave tile + lane&15)
  const int offA = (lane >> 4) * ROW_STRIDE + 3 * (wm * 64 + (lane & 15));
  const int offB = (lane >> 4) * ROW_STRIDE + 3 * (wn * 64 + (lane & 15));

  if (n_it > 0) {
    load_stage(0);
    store_stage(0);
  }
  __syncthreads();

  T a[4], bb[4];
  for (int it = 0; it < n_it; ++it) {
    const int cur = (ABL == 2 || ABL == 3) ? 0 : (it & 1);
    if (ABL == 0 && it + 1 < n_it) load_stage(it + 1);
    const T* pa = smem + cur * BUF_ELEMS;
    const T* pb = diag ? pa : pa + PANEL_ELEMS;
#pragma unroll
    for (int kk = 0; kk < KB / 4; ++kk) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if (ABL != 3 || it == 0) {
#pragma unroll
          for (int m = 0; m < 4; ++m) a[m] = pa[offA + kk * 4 * ROW_STRIDE + 48 * m + d];
#pragma unroll
          for (int n = 0; n < 4; ++n) bb[n] = pb[offB + kk * 4 * ROW_STRIDE + 48 * n + d];
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[m][n] = M::mma(a[m], bb[n], acc[m][n]);
      }
    }
    if (ABL < 2 || ABL == 4 || ABL == 5) {
      if (ABL != 5 && it + 1 < n_it) store_stage(cur ^ 1);   // 5: barrier without LDS refill
      if (ABL != 4) __syncthreads();                         // 4: LDS refill without barrier (racy, timing only)
    }
  }

  // partial tile -> slab [(tile_lin * ksplit + ks)][128][128]
  const int ksplit = gridDim.x / n_tiles;
  T* slab = slabs + ((int64_t)tile_lin * ksplit + ks) * (TILE * TILE);
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wm * 64 + m * 16 + M::row(lane, r);
        const int col = wn * 64 + n * 16 + (lane & 15);
        slab[row * TILE + col] = acc[m][n][r];
      }
}

// ---------------------------------------------------------------------------
// Same tiling as gram_tile_kernel, but the panels travel HBM/L2 -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip, no ds_write -- the register-staged refill
// costs 6-12 % of the MFMA time, tools/gram_ablate.hip) through a 3-stage LDS ring with a
// counted vmcnt, so the DMAs of stage s+2 stay in flight across the barrier of stage s.
// One DMA piece = one wave-instruction = 64 lanes x 16 B = 1 KiB contiguous in LDS (every lane brings
// its own global address).  f64: a panel row (384 elements = 3 KiB) is 3 pieces.  f32: a row is 1.5 KiB,
// so rows are staged in PAIRS (r, r+4) that lie back to back in LDS -- 3 KiB = exactly 3 full pieces, the
// middle one gathering the tail of row r (lanes 0-31) and the head of row r+4 (lanes 32-63).  (The first
// f32 version used one full and one half-used piece per row: 32 instead of 24 DMA instructions per stage,
// and the waves stall in the DMA *issue*.)  An MFMA operand read touches rows kk*4 + 0..3, i.e. the four
// pairs at the same member, so the padded stride that keeps the stride-3 reads conflict-free is the one
// between PAIRS (784 floats = 16 banks mod 64, like the 400-element row stride of the f64 layout).
template <typename T>
struct DmaCfg;
template <>
struct DmaCfg<double> {
  static constexpr int ROW_PIECES = 3;       // (row-wise staging of the opt-in pair-tile kernel: pieces per row)
  static constexpr int UNITS = 4;            // LDS units per panel and stage: the KB = 4 rows
  static constexpr int UNIT_STRIDE = ROW_STRIDE;
  static constexpr int PIECE_ELEMS = 128;    // elements per piece
  static constexpr int KK_STRIDE = 4 * ROW_STRIDE;  // LDS distance between the row groups kk and kk+1
  __host__ __device__ static constexpr int row_off(int r) { return r * ROW_STRIDE; }
};
template <>
struct DmaCfg<float> {
  static constexpr int ROW_PIECES = 2;       // (pair-tile kernel: one full and one half-used piece per row)
  static constexpr int UNITS = 4;            // the 4 row pairs (r, r+4) of the KB = 8 rows
  static constexpr int UNIT_STRIDE = 2 * ROW_ELEMS + ROW_PAD;  // 784
  static constexpr int PIECE_ELEMS = 256;
  static constexpr int KK_STRIDE = ROW_ELEMS;       // member 1 of every pair follows member 0 directly
  __host__ __device__ static constexpr int row_off(int r) { return (r & 3) * UNIT_STRIDE + (r >> 2) * ROW_ELEMS; }
};
template <typename T>
constexpr int dma_panel_elems() { return DmaCfg<T>::UNITS * DmaCfg<T>::UNIT_STRIDE; }  // 1600 (f64) / 3136 (f32)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else static_assert(N == 0, "unsupported vmcnt");
}

template <int MAXN>
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
  // s_waitcnt takes an immediate: dispatch on the (wave-uniform) count
  if (n <= 0) wait_vmcnt<0>();
  else if (n == 1) wait_vmcnt<1>();
  else if (n == 2) wait_vmcnt<2>();
  else if (n == 3) wait_vmcnt<3>();
  else if (n == 4) wait_vmcnt<4>();
  else if (n == 5) wait_vmcnt<5>();
  else if (n == 6) wait_vmcnt<6>();
  else if (n == 7) wait_vmcnt<7>();
  else wait_vmcnt<8>();
  static_assert(MAXN <= 8, "extend the dispatch");
}

#ifdef AGGF_SMALL_PROF
// tools/small_probe.hip: shader cycles per wave of the small-system kernel spent in park (incl. the wait for the
// fetched frames) / group sums / MFMA / the three barriers, and the stage count
__device__ unsigned long long aggf_small_prof[9];
#define AGGF_SP_T(x) const uint64_t x = __builtin_readcyclecounter()
#else
#define AGGF_SP_T(x)
#endif
#ifdef AGGF_GRAM_PROF
// tools/clock_probe.hip: shader cycles per wave spent issuing DMAs / in ds_read+MFMA / in waitcnt+barrier
__device__ unsigned long long aggf_gram_prof[4];
#define AGGF_PROF_T(x) const uint64_t x = __builtin_readcyclecounter()
#else
#define AGGF_PROF_T(x)
#endif

// M32 (float, NW = 8 only): v_mfma_f32_32x32x2_f32 instead of 16x16x4 -- the same operand reads per flop (they
// depend on the 64 x 32 wave tile only) in half as many MFMA instructions of twice the length.
// pieces q of a stage (piece q goes with MFMA group q * groups / ppw) issued up to and including group g
// (float32 has 6 groups for 3 pieces: groups 0, 2, 4; shifted to 1, 3, 5 -- the last one behind the barrier -- measured the same)
constexpr int dma_piece_group(int groups, int ppw, int q) { return q * groups / ppw; }
constexpr int dma_pieces_upto(int groups, int ppw, int g) {
  int n = 0;
  for (int q = 0; q < ppw; ++q) n += (dma_piece_group(groups, ppw, q) <= g) ? 1 : 0;
  return n;
}

// TWO: the operand is the column-concatenation [X | X2] of two arrays that lie apart in HBM (panels 0..np1-1 from X,
// row stride ld; the rest from X2, row stride ld2) -- the noised maps' [forces | generated-site forces], which round 2
// materialised as one (T, N + n_cg, 3) array twice per step (aggf_gram_pair).
template <typename T, int ABL = 0, int NBUF = 3, int WPS = 2, int NW = 4, bool SPREAD_DMA = false, bool M32 = false,
          int ES = 0, bool ES_DMA_AFTER = false, bool TWO = false>
__global__ __launch_bounds__(64 * NW, WPS) void gram_tile_dma_kernel(
    const T* __restrict__ X, int64_t n_rows, int64_t ld, int32_t nt1, int32_t n_tiles, int32_t ksplit,
    const int32_t* __restrict__ tile_table, int64_t frames_per_split, T* __restrict__ slabs,
    const T* __restrict__ X2 = nullptr, int64_t ld2 = 0, int32_t np1 = 0) {
  using M = Mfma<T>;
  using acc_t = typename M::acc_t;
  constexpr int KB = GramCfg<T>::KB;
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int PE = DmaCfg<T>::PIECE_ELEMS;
  constexpr int UNITS = DmaCfg<T>::UNITS;
  constexpr int PANELS = 2;  // diagonal tiles stage their panel twice (one code path, fixed vmcnt)
  constexpr int PIECES = PANELS * UNITS * 3;         // per stage: 24 for both dtypes
  constexpr int PPW = PIECES / NW;                   // per wave: 6 with 4 waves, 3 with 8
  static_assert(PIECES % NW == 0, "piece split");
  // NW waves tile the 128x128 output as 2 x (NW/2): wave tile 64 x 64 (NW = 4) or 64 x 32 (NW = 8,
  // four waves per SIMD with two workgroups per CU: half the accumulators and half the DMA
  // instructions per wave, twice the waves to cover each other's stalls)
  constexpr int WN = NW / 2;
  constexpr int WCOLS = TILE / WN;                   // columns per wave: 64 / 32
  constexpr int NACC = WCOLS / 16;                   // 16-column accumulator tiles per wave: 4 / 2
  constexpr int NTHREADS = 64 * NW;
  constexpr int PANEL_ELEMS = dma_panel_elems<T>();
  constexpr int BUF_ELEMS = PANELS * PANEL_ELEMS;
  constexpr int AHEAD = NBUF - 1;  // stages in flight ahead of the one being computed
  constexpr bool EARLY_SYNC = ES > 0 && SPREAD_DMA && !M32 && ABL == 0;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // workgroup -> (split, tile), XCD-aware.  Workgroups b and b+8 land on the same XCD (observed
  // round-robin dispatch; only speed depends on it), and an XCD runs 64 of them at a time (32 CUs x
  // 2).  Give each XCD 64 CONSECUTIVE entries of the (split, tile) list, whose tile order walks
  // 8x8 super-blocks of the tile grid (tile_table): those 64 workgroups then share 16 panels in
  // their XCD's L2 instead of streaming 128 panels past it (FETCH_SIZE: 2.9 TB per launch before).
  const int b = blockIdx.x;
  const int v = (((b >> 3) >> 6) * 8 + (b & 7)) * 64 + ((b >> 3) & 63);
  if (v >= ksplit * n_tiles) return;  // grid is padded to a multiple of 512
  const int ks = v / n_tiles;
  const int packed = tile_table[v - ks * n_tiles];
  const int ti = packed >> 16, tj = packed & 0xffff;
  const int tile_lin = ti * nt1 - ti * (ti - 1) / 2 + (tj - ti);  // row-major upper-triangle index

  const int64_t t_begin = (int64_t)ks * frames_per_split;
  int64_t t_end = t_begin + frames_per_split;
  if (t_end > n_rows) t_end = n_rows;
  const int n_it = t_begin < t_end ? (int)((t_end - t_begin + KB - 1) / KB) : 0;

  // this wave's DMA pieces: global element offset within a stage (per lane), the frame row the lane
  // reads (for the ragged last stage) and the LDS element offset of the piece
  int64_t g_off[PPW];
  int l_off[PPW], p_row[PPW];
  const T* g_base[TWO ? PPW : 1];
  int64_t g_ld[TWO ? PPW : 1];
  (void)g_base;
  (void)g_ld;
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    const int p = wave + NW * q;
    const int panel = p / (UNITS * 3);
    const int unit = (p - panel * UNITS * 3) / 3;
    const int cp = p % 3;
    int r, elem;
    if (F32) {
      // pair (unit, unit + 4): piece 0 = row unit [0, 256); piece 1 = row unit [256, 384) | row unit+4 [0, 128);
      // piece 2 = row unit+4 [128, 384)
      const bool second = cp == 2 || (cp == 1 && lane >= 32);
      r = unit + (second ? 4 : 0);
      elem = cp == 0 ? lane * 4 : cp == 2 ? 128 + lane * 4 : (lane < 32 ? 256 + lane * 4 : (lane - 32) * 4);
    } else {
      r = unit;
      elem = cp * PE + lane * 2;
    }
    p_row[q] = r;
    if constexpr (TWO) {
      const int pj = panel ? tj : ti;
      const bool second = pj >= np1;
      g_base[q] = second ? X2 : X;
      g_ld[q] = second ? ld2 : ld;
      g_off[q] = (int64_t)r * g_ld[q] + (int64_t)(second ? pj - np1 : pj) * ROW_ELEMS + elem;
    } else {
      const int64_t col = (int64_t)(panel ? tj : ti) * ROW_ELEMS;
      g_off[q] = (int64_t)r * ld + col + elem;
    }
    l_off[q] = panel * PANEL_ELEMS + unit * DmaCfg<T>::UNIT_STRIDE + cp * PE;
  }

  // rows past the end of this split's frame range must read as zeros (last stage only)
  // Stage order is rotated by (ti + tj) mod 8: the 8 workgroups of an XCD group that share a
  // panel then ask for a given (panel, stage) in 8 different iterations, so the first request
  // misses and the other 7 hit the XCD's L2.  In lock-step (all at once) every request misses,
  // because concurrent misses to one line are not merged (TCC_MISS == TCC_EA0_RDREQ before).
  const int skew = n_it > 16 ? ((ti + tj) & 7) : 0;  // (0, &1, &3, &7: within 0.5 % of each other since the DMAs are spread)
  auto stage_of = [&](int seq) { const int v = seq + skew; return v >= n_it ? v - n_it : v; };
  // sequence position of the one stage with fewer than KB valid rows (last split only), or -1
  const bool ragged = n_it > 0 && (t_end - t_begin) % KB != 0;
  const int ragged_seq = ragged ? (n_it - 1 - skew + (n_it - 1 - skew < 0 ? n_it : 0)) : -1;
  auto prep_stage = [&](int seq) {
    const int s = seq;  // ring slot follows the sequence position
    const int64_t t0 = t_begin + (int64_t)stage_of(seq) * KB;
    if (t0 + KB > t_end) {
      T* lbase = smem + (s % NBUF) * BUF_ELEMS;
      const int first = (int)(t_end - t0);
      for (int e = tid; e < PANELS * (KB - first) * ROW_ELEMS; e += NTHREADS) {
        const int panel = e / ((KB - first) * ROW_ELEMS);
        const int rem = e - panel * (KB - first) * ROW_ELEMS;
        const int r = first + rem / ROW_ELEMS, c = rem % ROW_ELEMS;
        lbase[panel * PANEL_ELEMS + DmaCfg<T>::row_off(r) + c] = 0;
      }
    }
  };
  auto issue_piece = [&](int s, int q) {
    // ABL 3 (ablation): every stage re-reads the first rows of the split -> all DMAs hit the L2
    // ABL 4: cycle over 8 stages -> DMAs miss the L1 but hit the L2
    const int64_t t0 = t_begin + (ABL == 3 ? 0 : ABL == 4 ? (int64_t)(stage_of(s) & 7) * KB : (int64_t)stage_of(s) * KB);
    const bool row_ok = t0 + p_row[q] < t_end;
    if (row_ok) {
      const T* src = TWO ? g_base[TWO ? q : 0] + t0 * g_ld[TWO ? q : 0] + g_off[q] : X + t0 * ld + g_off[q];
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)src,
          (__attribute__((address_space(3))) void*)(smem + (s % NBUF) * BUF_ELEMS + l_off[q]), 16, 0, 0);
    }
  };
  auto issue_stage = [&](int s) {
    prep_stage(s);
#pragma unroll
    for (int q = 0; q < PPW; ++q) issue_piece(s, q);
  };

  acc_t acc[4][NACC];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < NACC; ++n) acc[m][n] = acc_zero<T>();
  typedef float __attribute__((ext_vector_type(16))) f32x16;
  f32x16 acc32[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc32[m][r] = 0.f;
  // 32x32x2 operands: A[i = atom (lane & 31)][k = row (lane >> 5)], two K steps of two rows per row group
  const int offA32 = (lane >> 5) * DmaCfg<T>::UNIT_STRIDE + 3 * (wm * 64 + (lane & 31));
  const int offB32 = PANEL_ELEMS + (lane >> 5) * DmaCfg<T>::UNIT_STRIDE + 3 * (wn * WCOLS + (lane & 31));

  // MFMA operand of row group kk: rows kk*4 + (lane >> 4) -- f64: four consecutive rows; f32: member kk of
  // the four pairs
  const int offA = (lane >> 4) * DmaCfg<T>::UNIT_STRIDE + 3 * (wm * 64 + (lane & 15));
  const int offB = PANEL_ELEMS + (lane >> 4) * DmaCfg<T>::UNIT_STRIDE + 3 * (wn * WCOLS + (lane & 15));
  constexpr int KKS = DmaCfg<T>::KK_STRIDE;

  if (n_it > 0) issue_stage(0);
  if (AHEAD > 1 && n_it > 1) issue_stage(1);
  if (AHEAD > 1 && n_it > 1 && ragged_seq != 1) wait_vmcnt<PPW>(); else wait_vmcnt<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

#ifdef AGGF_GRAM_PROF
  uint64_t prof_issue = 0, prof_compute = 0, prof_sync = 0;
#endif
  for (int it = 0; it < n_it; ++it) {
    AGGF_PROF_T(p0);
    // DMAs of stage it+2 first (placing them between the MFMA groups instead, or raising the
    // wave priority around the MFMA groups, measured no better: tools/gram_ablate.hip history)
    // SPREAD_DMA: one share of the stage's DMAs before each MFMA group instead of all up front
    // (the DMA path accepts ~1 KiB per 100+ cycles under this load; a burst of 24 KiB after every
    // barrier queues the waves behind it)
    constexpr int GROUPS = 3 * KB / 4;
    constexpr bool SPREAD = SPREAD_DMA;
    const bool issue_now = (ABL == 0 || ABL >= 3) && it + AHEAD < n_it;
    if (issue_now) {
      if (SPREAD) prep_stage(it + AHEAD); else issue_stage(it + AHEAD);
    }
    const T* pa = smem + (((ABL == 1 || ABL == 2) ? it % 2 : it % NBUF)) * BUF_ELEMS;
    AGGF_PROF_T(p1);
    T aL[ES > 0 ? ES : 1][4], bL[ES > 0 ? ES : 1][NACC];  // operands of the groups behind an early barrier
    (void)aL;
    (void)bL;
#pragma unroll
    for (int kk = 0; kk < KB / 4; ++kk) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if constexpr (M32) {
          float a32[2][2], b32[2];  // [K step of two rows][32-atom tile]
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int m = 0; m < 2; ++m) a32[s2][m] = (float)pa[offA32 + kk * KKS + s2 * 2 * DmaCfg<T>::UNIT_STRIDE + 96 * m + d];
            b32[s2] = (float)pa[offB32 + kk * KKS + s2 * 2 * DmaCfg<T>::UNIT_STRIDE + d];
          }
          if (SPREAD && issue_now) {
#pragma unroll
            for (int q = 0; q < PPW; ++q)
              if (q * GROUPS / PPW == kk * 3 + d) issue_piece(it + AHEAD, q);
          }
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int m = 0; m < 2; ++m)
              acc32[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a32[s2][m], b32[s2], acc32[m], 0, 0, 0);
        } else {
        constexpr int ESG = EARLY_SYNC ? ES : 0;  // MFMA groups that run behind the stage barrier
        const int g = kk * 3 + d;
        T a[4], bb[NACC];
        if (ESG > 0 && g == GROUPS - ESG) {
          // operands of all remaining groups: once these reads have returned, nobody needs the slot any more
          // (reading them one group earlier: 764 against 743 ms at C3 -- slower)
#pragma unroll
          for (int h = 0; h < (ESG > 0 ? ESG : 1); ++h) {
            const int kh = (GROUPS - ESG + h) / 3, dh = (GROUPS - ESG + h) % 3;
#pragma unroll
            for (int m = 0; m < 4; ++m) aL[h][m] = pa[offA + kh * KKS + 48 * m + dh];
#pragma unroll
            for (int n = 0; n < NACC; ++n) bL[h][n] = pa[offB + kh * KKS + 48 * n + dh];
          }
        }
        if (ESG > 0 && g >= GROUPS - ESG) {
#pragma unroll
          for (int m = 0; m < 4; ++m) a[m] = aL[g - (GROUPS - ESG)][m];
#pragma unroll
          for (int n = 0; n < NACC; ++n) bb[n] = bL[g - (GROUPS - ESG)][n];
        } else {
#pragma unroll
          for (int m = 0; m < 4; ++m) a[m] = pa[offA + kk * KKS + 48 * m + d];
#pragma unroll
          for (int n = 0; n < NACC; ++n) bb[n] = pa[offB + kk * KKS + 48 * n + d];
        }
        // between the operand reads and the MFMAs of the group: the reads are in flight while the DMA
        // waits to be accepted (before the reads: +3.5 %, after the MFMAs: +1 %, tools/clock_probe.hip)
        constexpr bool DMA_AFTER = ES_DMA_AFTER && ESG > 0;  // the barrier group issues its DMA piece behind the barrier
        if (SPREAD && issue_now && !(DMA_AFTER && g == GROUPS - ESG)) {
#pragma unroll
          for (int q = 0; q < PPW; ++q)
            if (dma_piece_group(GROUPS, PPW, q) == g) issue_piece(it + AHEAD, q);  // piece q goes with group q*GROUPS/PPW
        }
        if (ESG > 0 && g == GROUPS - ESG) {
          // the stage's barrier BEFORE the MFMAs of its last group(s): their operands are in registers once the LDS
          // reads have returned, so the slot is free for the next DMA, and the MFMAs run while the waves meet.
          // (Pieces of stage it+2 that go with later groups are not issued yet: the counted wait allows the rest.)
          constexpr int ISSUED = dma_pieces_upto(GROUPS, PPW, GROUPS - ESG - (DMA_AFTER ? 1 : 0));
          static_assert(ISSUED >= 0 && ISSUED <= PPW, "pieces issued by the barrier group");
          if (AHEAD > 1 && it + 2 < n_it && it + 2 != ragged_seq) wait_vmcnt<ISSUED>(); else wait_vmcnt<0>();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          // (right behind the barrier; behind the group's MFMAs instead: 744.5 against 737.9 ms at C3, same box)
          if (DMA_AFTER && SPREAD && issue_now) {
#pragma unroll
            for (int q = 0; q < PPW; ++q)
              if (dma_piece_group(GROUPS, PPW, q) == g) issue_piece(it + AHEAD, q);
          }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < NACC; ++n) acc[m][n] = M::mma(a[m], bb[n], acc[m][n]);
        }
      }
    }
    AGGF_PROF_T(p2);
    // stage it+1 must have landed (this wave's pieces), stage it+2 may stay in flight
    if (ABL != 2 && !EARLY_SYNC) {
      // (the ragged stage of the last split issues fewer DMAs: a counted wait would let pieces
      // of stage it+1 slip through)
      if (AHEAD > 1 && it + 2 < n_it && it + 2 != ragged_seq) wait_vmcnt<PPW>(); else wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
#ifdef AGGF_GRAM_PROF
    const uint64_t p3 = __builtin_readcyclecounter();
    prof_issue += p1 - p0;
    prof_compute += p2 - p1;
    prof_sync += p3 - p2;
#endif
  }
#ifdef AGGF_GRAM_PROF
  if (lane == 0) {
    atomicAdd(&aggf_gram_prof[0], (unsigned long long)prof_issue);
    atomicAdd(&aggf_gram_prof[1], (unsigned long long)prof_compute);
    atomicAdd(&aggf_gram_prof[2], (unsigned long long)prof_sync);
    atomicAdd(&aggf_gram_prof[3], (unsigned long long)n_it);
  }
#endif

  T* slab = slabs + ((int64_t)tile_lin * ksplit + ks) * (TILE * TILE);
  if constexpr (M32) {
    // D of v_mfma_f32_32x32x2_f32: lane holds column (lane & 31), rows 8 (r / 4) + 4 (lane >> 5) + r % 4
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + m * 32 + 8 * (r / 4) + 4 * (lane >> 5) + (r % 4);
        const int col = wn * WCOLS + (lane & 31);
        slab[row * TILE + col] = (T)acc32[m][r];
      }
    return;
  }
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < NACC; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wm * 64 + m * 16 + M::row(lane, r);
        const int col = wn * WCOLS + n * 16 + (lane & 15);
        slab[row * TILE + col] = acc[m][n][r];
      }
}

// ---------------------------------------------------------------------------
// The tile kernel for everything that is NOT a ready-made panel in HBM: constraint groups (`@ con_mat`,
// qplinear.py:69-70), float32 trajectories with float64 products (the reference's arithmetic: con_mat is float64),
// site counts that are no multiple of 128.  Round 2 wrote a packed float64 copy of the trajectory for these
// (pack_groups_kernel: 98 GB read + 66 GB written at C3 with bond-pair constraints) and ran the panel kernel on it.
// Here the raw frame rows travel HBM -> LDS by the same LDS-DMA ring, exactly as they lie in HBM, and the group sums,
// the dtype conversion and the zero padding happen in the MFMA OPERAND READ: lane (column c, row r) adds up the
// members of column c from the raw row (up to MAXM of them; a bit mask selects the ones that exist).
//   col_off[c * MAXM + j] = 3 * atom of member j of reduced column c (element offset inside a frame), -1 = none;
//   panel_lo[p]           = first atom of the window of panel p (its 128 columns' members lie in
//                           [panel_lo[p], panel_lo[p] + span_atoms));
//   row_slot              = LDS bytes per staged row: the window (span_atoms * 3 * sizeof(TIn) + 16 for the 16-byte
//                           alignment of the first piece), rounded to 16, + 32 so that the four rows of an operand
//                           read start in different banks.
// A row's window starts at an arbitrary 4/8-byte aligned address: the DMA pieces start at the 16-byte boundary below
// it and the operand offsets carry the misalignment, which depends on the row only through (row mod 4) because every
// stage starts at a multiple of 4 frames.  Requirements (checked by the host side, else the pack path is taken):
// F 16-byte aligned, T * 3N * sizeof(TIn) a multiple of 16, at most MAXM <= 4 members per column, two stages in 80 KB.
// ---------------------------------------------------------------------------
// float32 products, "quad" shape: FOUR 4-wave workgroups per CU, each one 128x128 tile with 64x64 wave tiles, 4 frame
// rows per stage.  Why: the float32 MFMA takes 32 cycles instead of 64, so in the 8-wave shape above (64x32 wave
// tiles, 8 MFMAs between two operand fetches) every fixed per-group cost -- operand-read latency, s_waitcnt, the
// barrier -- weighs twice as much as in float64 (MFMA pipes busy 81-85 % against 89 %).  Here a group is 16 MFMAs
// (512 cycles, as in float64) fed by 8 operand reads instead of 2 x 6, the barrier joins 4 waves instead of 8, and the
// smaller stage (2 panels x 4 rows x 1.6 KB) lets four workgroups share a CU, so each SIMD still has four waves.
// Rows lie one by one in LDS (stride 384 + 16 floats: the four k-rows of an operand read start 16 banks apart), a row
// is one full and one half-used 1-KiB DMA piece.  [X | X2] as in the TWO instantiation of gram_tile_dma_kernel.
constexpr int Q_KB = 4, Q_NW = 4, Q_THREADS = 256, Q_NBUF = 3;
constexpr int Q_PANEL = Q_KB * ROW_STRIDE;          // 1600 floats
constexpr int Q_BUF = 2 * Q_PANEL;                  // 3200 floats = 12.8 KB per stage
constexpr int Q_PPW = 2 * Q_KB * 2 / Q_NW;          // DMA pieces per wave and stage: 4

__global__ __launch_bounds__(Q_THREADS, 4) void gram_tile_f32q_kernel(
    const float* __restrict__ X, int64_t n_rows, int64_t ld, int32_t nt1, int32_t n_tiles, int32_t ksplit,
    const int32_t* __restrict__ tile_table, int64_t frames_per_split, float* __restrict__ slabs,
    const float* __restrict__ X2, int64_t ld2, int32_t np1) {
  using M = Mfma<float>;
  using acc_t = typename M::acc_t;
  constexpr int KB = Q_KB, NBUF = Q_NBUF, AHEAD = NBUF - 1, PPW = Q_PPW, GROUPS = 3;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int b = blockIdx.x;
  const int v = (((b >> 3) >> 6) * 8 + (b & 7)) * 64 + ((b >> 3) & 63);  // XCD-aware order, see gram_tile_dma_kernel
  if (v >= ksplit * n_tiles) return;
  const int ks = v / n_tiles;
  const int packed = tile_table[v - ks * n_tiles];
  const int ti = packed >> 16, tj = packed & 0xffff;
  const int tile_lin = ti * nt1 - ti * (ti - 1) / 2 + (tj - ti);

  const int64_t t_begin = (int64_t)ks * frames_per_split;
  int64_t t_end = t_begin + frames_per_split;
  if (t_end > n_rows) t_end = n_rows;
  const int n_it = t_begin < t_end ? (int)((t_end - t_begin + KB - 1) / KB) : 0;

  // this wave's DMA pieces: piece p = wave + 4 q -> (panel, row, half); a row = [0, 256) by all lanes + [256, 384) by
  // lanes 0..31
  const float* g_base[PPW];
  int64_t g_ld[PPW], g_off[PPW];
  int l_off[PPW], p_row[PPW];
  bool lane_on[PPW];
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    const int p = wave + Q_NW * q;
    const int panel = p >> 3, row = (p & 7) >> 1, cp = p & 1;
    const int pj = panel ? tj : ti;
    const bool second = pj >= np1;
    g_base[q] = second ? X2 : X;
    g_ld[q] = second ? ld2 : ld;
    g_off[q] = (int64_t)row * g_ld[q] + (int64_t)(second ? pj - np1 : pj) * ROW_ELEMS + cp * 256 + lane * 4;
    l_off[q] = panel * Q_PANEL + row * ROW_STRIDE + cp * 256;
    p_row[q] = row;
    lane_on[q] = cp == 0 || lane < 32;
  }
  const int skew = n_it > 16 ? ((ti + tj) & 7) : 0;
  auto stage_of = [&](int seq) { const int u = seq + skew; return u >= n_it ? u - n_it : u; };
  const bool ragged = n_it > 0 && (t_end - t_begin) % KB != 0;
  const int ragged_seq = ragged ? (n_it - 1 - skew + (n_it - 1 - skew < 0 ? n_it : 0)) : -1;
  auto prep_stage = [&](int seq) {
    const int64_t t0 = t_begin + (int64_t)stage_of(seq) * KB;
    if (t0 + KB > t_end) {
      float* lbase = smem + (seq % NBUF) * Q_BUF;
      const int first = (int)(t_end - t0);
      for (int e = tid; e < 2 * (KB - first) * ROW_ELEMS; e += Q_THREADS) {
        const int panel = e / ((KB - first) * ROW_ELEMS);
        const int rem = e - panel * (KB - first) * ROW_ELEMS;
        const int r = first + rem / ROW_ELEMS, c = rem % ROW_ELEMS;
        lbase[panel * Q_PANEL + r * ROW_STRIDE + c] = 0.f;
      }
    }
  };
  auto issue_piece = [&](int seq, int q) {
    const int64_t t0 = t_begin + (int64_t)stage_of(seq) * KB;
    if (lane_on[q] && t0 + p_row[q] < t_end) {
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(g_base[q] + t0 * g_ld[q] + g_off[q]),
          (__attribute__((address_space(3))) void*)(smem + (seq % NBUF) * Q_BUF + l_off[q]), 16, 0, 0);
    }
  };

  acc_t acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = acc_zero<float>();
  const int offA = (lane >> 4) * ROW_STRIDE + 3 * (wm * 64 + (lane & 15));
  const int offB = Q_PANEL + (lane >> 4) * ROW_STRIDE + 3 * (wn * 64 + (lane & 15));

  if (n_it > 0) {
    prep_stage(0);
#pragma unroll
    for (int q = 0; q < PPW; ++q) issue_piece(0, q);
  }
  if (n_it > 1) {
    prep_stage(1);
#pragma unroll
    for (int q = 0; q < PPW; ++q) issue_piece(1, q);
  }
  if (n_it > 1 && ragged_seq != 1) wait_vmcnt<PPW>(); else wait_vmcnt<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  for (int it = 0; it < n_it; ++it) {
    const bool issue_now = it + AHEAD < n_it;
    if (issue_now) prep_stage(it + AHEAD);
    const float* pa = smem + (it % NBUF) * Q_BUF;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float a[4], bb[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) a[m] = pa[offA + 48 * m + d];
#pragma unroll
      for (int n = 0; n < 4; ++n) bb[n] = pa[offB + 48 * n + d];
      // pieces 0, 1 go with group 0, piece 2 with group 1, piece 3 with the last group -- behind its barrier
      if (issue_now) {
        if (d == 0) {
          issue_piece(it + AHEAD, 0);
          issue_piece(it + AHEAD, 1);
        } else if (d == 1) {
          issue_piece(it + AHEAD, 2);
        }
      }
      if (d == 2) {
        // the stage's barrier in front of the MFMAs of its last group (operands in registers: the slot is free)
        if (it + 2 < n_it && it + 2 != ragged_seq) wait_vmcnt<3>(); else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (issue_now) issue_piece(it + AHEAD, 3);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = M::mma(a[m], bb[n], acc[m][n]);
    }
  }

  float* slab = slabs + ((int64_t)tile_lin * ksplit + ks) * (TILE * TILE);
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wm * 64 + m * 16 + M::row(lane, r);
        const int col = wn * 64 + n * 16 + (lane & 15);
        slab[row * TILE + col] = acc[m][n][r];
      }
}

constexpr int GA_HDR = 32;     // zeroed bytes in front of every staged row: what absent members and padding columns read
constexpr int GA_MAXPPW = 6;   // DMA pieces per wave and stage, at most (host-checked)

template <typename TIn, typename TC, int MAXM>
__global__ __launch_bounds__(512, MAXM <= 2 ? 4 : 2) void gram_tile_gather_kernel(
    const TIn* __restrict__ F, int64_t n_rows, int32_t N, const int32_t* __restrict__ col_off,
    const int32_t* __restrict__ panel_lo, int32_t span_bytes, int32_t row_slot, int32_t nbuf, int32_t nt1,
    int32_t n_tiles, int32_t ksplit, const int32_t* __restrict__ tile_table, int64_t frames_per_split,
    TC* __restrict__ slabs) {
  using M = Mfma<TC>;
  using acc_t = typename M::acc_t;
  constexpr int KB = GramCfg<TC>::KB;   // 4 (float64 products) or 8 (float32 products)
  constexpr int NW = 8, WN = 4, WCOLS = 32, NACC = 2;
  constexpr int SI = (int)sizeof(TIn);
  constexpr int GROUPS = 3 * KB / 4;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  char* smem = smem_raw;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int b = blockIdx.x;
  const int v = (((b >> 3) >> 6) * 8 + (b & 7)) * 64 + ((b >> 3) & 63);  // XCD-aware order, see gram_tile_dma_kernel
  if (v >= ksplit * n_tiles) return;
  const int ks = v / n_tiles;
  const int packed = tile_table[v - ks * n_tiles];
  const int ti = packed >> 16, tj = packed & 0xffff;
  const int tile_lin = ti * nt1 - ti * (ti - 1) / 2 + (tj - ti);

  const int64_t t_begin = (int64_t)ks * frames_per_split;  // a multiple of KB (hence of 4)
  int64_t t_end = t_begin + frames_per_split;
  if (t_end > n_rows) t_end = n_rows;
  const int n_it = t_begin < t_end ? (int)((t_end - t_begin + KB - 1) / KB) : 0;
  const int64_t row_bytes = (int64_t)N * 3 * SI;
  const int stage_bytes = 2 * KB * row_slot;

  // windows of the two panels: first atom (clamped: a bad table must not send a DMA outside the array)
  int lo[2];
  lo[0] = panel_lo[ti];
  lo[1] = panel_lo[tj];
#pragma unroll
  for (int p = 0; p < 2; ++p) lo[p] = lo[p] < 0 ? 0 : (lo[p] > N - 1 ? N - 1 : lo[p]);
  const char* Fb = reinterpret_cast<const char*>(F);
  const int64_t f_end = (int64_t)n_rows * row_bytes;  // a multiple of 16 (host-checked)
  // byte misalignment of row r's window (r mod 4 decides: every stage starts at a multiple of 4 frames)
  auto mis_of = [&](int p, int r) { return (int)((((int64_t)r * row_bytes) + (int64_t)lo[p] * 3 * SI) & 15); };

  // zero headers of every row slot of every ring slot (the DMAs never write there)
  for (int e = tid; e < nbuf * 2 * KB * (GA_HDR / 4); e += 512) {
    const int row = e / (GA_HDR / 4), w = e - row * (GA_HDR / 4);
    *reinterpret_cast<int*>(smem + row * row_slot + w * 4) = 0;
  }

  // operand offsets (bytes inside a stage): row slot + header + misalignment + member offset; an absent member (and
  // every member of a padding column) reads the zero header of the lane's row.  f64 products: the lane's row is
  // (lane >> 4); f32 products: rows (lane >> 4) and (lane >> 4) + 4 (same misalignment), 4 * row_slot apart.
  const int r_lane = lane >> 4;
  int offA[4][MAXM], offB[NACC][MAXM];
  {
    const int baseA = r_lane * row_slot, baseB = KB * row_slot + r_lane * row_slot;
    const int dataA = GA_HDR + mis_of(0, r_lane), dataB = GA_HDR + mis_of(1, r_lane);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int c = ti * TILE + wm * 64 + 16 * m + (lane & 15);
#pragma unroll
      for (int j = 0; j < MAXM; ++j) {
        const int e = col_off[(int64_t)c * MAXM + j];
        const int rel = (e - 3 * lo[0]) * SI;
        const bool ok = e >= 0 && rel >= 0 && rel + 3 * SI <= span_bytes;
        offA[m][j] = baseA + (ok ? dataA + rel : 0);
      }
    }
#pragma unroll
    for (int n = 0; n < NACC; ++n) {
      const int c = tj * TILE + wn * WCOLS + 16 * n + (lane & 15);
#pragma unroll
      for (int j = 0; j < MAXM; ++j) {
        const int e = col_off[(int64_t)c * MAXM + j];
        const int rel = (e - 3 * lo[1]) * SI;
        const bool ok = e >= 0 && rel >= 0 && rel + 3 * SI <= span_bytes;
        offB[n][j] = baseB + (ok ? dataB + rel : 0);
      }
    }
  }

  // DMA pieces of a stage, fixed per wave: piece index pc = wave + 8 q -> (panel, row, k-th KiB of the row's aligned
  // window).  Per lane: the byte offset from the stage's first frame row (g_off) and whether the lane takes part.
  const int ppr = (span_bytes + 15 + 1023) / 1024;        // pieces per row (15: largest misalignment)
  const int n_pieces = 2 * KB * ppr;                        // a multiple of 8
  const int ppw = n_pieces / NW;                            // pieces per wave, <= GA_MAXPPW
  int64_t g_off[GA_MAXPPW];
  int l_dst[GA_MAXPPW], p_row[GA_MAXPPW];
  unsigned act = 0;
#pragma unroll
  for (int q = 0; q < GA_MAXPPW; ++q) {
    const int pc = wave + NW * q;
    const int panel = pc / (KB * ppr), rem = pc - panel * KB * ppr, r = rem / ppr, k = rem - r * ppr;
    const int pn = panel > 1 ? 1 : panel;
    const int64_t start = (int64_t)r * row_bytes + (int64_t)lo[pn] * 3 * SI;
    const int64_t al = start & ~(int64_t)15;
    const int need = (int)(start - al) + span_bytes;
    const int off = k * 1024 + lane * 16;
    g_off[q] = al + k * 1024;  // wave-uniform (scalar registers); the lane's 16 bytes are added when the piece is issued
    l_dst[q] = pn * KB * row_slot + r * row_slot + GA_HDR + k * 1024;
    p_row[q] = r;
    // (lane 0 of every piece stays active -- 16 bytes of the same array -- so that a regular stage issues exactly ppw
    // instructions per wave whatever the alignment)
    act |= (q < ppw && (off < need || lane == 0)) ? (1u << q) : 0u;
  }
  const int skew = n_it > 16 ? ((ti + tj) & 7) : 0;
  auto stage_of = [&](int seq) { const int u = seq + skew; return u >= n_it ? u - n_it : u; };
  // The stage that holds the split's last frames is "irregular": rows past t_end issue nothing and a piece that would
  // cross the end of the array is dropped -- the counted waits must not assume a full stage there.
  const int irr_seq = n_it > 0 ? (n_it - 1 - skew + (n_it - 1 - skew < 0 ? n_it : 0)) : -1;
  auto issue_piece = [&](int seq, int q) {
    const int64_t t0 = t_begin + (int64_t)stage_of(seq) * KB;
    const int64_t src = t0 * row_bytes + g_off[q] + lane * 16;
    bool ok = (act >> q) & 1u;
    if (seq == irr_seq) ok = ok && (t0 + p_row[q] < t_end) && (src + 16 <= f_end);
    if (ok) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Fb + src),
                                       (__attribute__((address_space(3))) void*)(smem + (seq % nbuf) * stage_bytes + l_dst[q]),
                                       16, 0, 0);
    }
  };
  auto prep_stage = [&](int seq) {  // rows past the end of the split read as zeros (one ragged stage at most)
    if (seq != irr_seq) return;
    const int64_t t0 = t_begin + (int64_t)stage_of(seq) * KB;
    if (t0 + KB > t_end) {
      const int first = (int)(t_end - t0);
      char* base = smem + (seq % nbuf) * stage_bytes;
      for (int e = tid * 4; e < 2 * (KB - first) * row_slot; e += 512 * 4) {
        const int panel = e / ((KB - first) * row_slot), rem = e - panel * (KB - first) * row_slot;
        *reinterpret_cast<int*>(base + panel * KB * row_slot + first * row_slot + rem) = 0;
      }
    }
  };

  acc_t acc[4][NACC];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < NACC; ++n) acc[m][n] = acc_zero<TC>();

  const int ahead = nbuf - 1;  // 2 (three ring slots) or 1
  // which MFMA group a piece goes with: spread over the groups (three slots) or all in front of the first (two slots:
  // the stage's own duration is all the time its successor's DMAs get)
  auto group_of_piece = [&](int q) { return ahead > 1 ? q * GROUPS / ppw : 0; };
  for (int s0 = 0; s0 < ahead && s0 < n_it; ++s0) {
    prep_stage(s0);
#pragma unroll
    for (int q = 0; q < GA_MAXPPW; ++q)
      if (q < ppw) issue_piece(s0, q);
  }
  if (ahead > 1 && n_it > 1 && irr_seq != 1) wait_vmcnt_dyn<8>(ppw); else wait_vmcnt<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  for (int it = 0; it < n_it; ++it) {
    const bool issue_now = it + ahead < n_it;
    if (issue_now) prep_stage(it + ahead);
    const int so = (it % nbuf) * stage_bytes;
    // pieces of stage it + ahead issued in front of the barrier group (the last group's own pieces follow the barrier)
    int issued_before = 0;
#pragma unroll
    for (int kk = 0; kk < KB / 4; ++kk) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const int g = kk * 3 + d;
        const int sh = so + kk * 4 * row_slot + d * SI;
        TC a[4], bb[NACC];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          TC sum = (TC) * reinterpret_cast<const TIn*>(smem + offA[m][0] + sh);
#pragma unroll
          for (int j = 1; j < MAXM; ++j) sum += (TC) * reinterpret_cast<const TIn*>(smem + offA[m][j] + sh);  // table order, like `@ con_mat`
          a[m] = sum;
        }
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
          TC sum = (TC) * reinterpret_cast<const TIn*>(smem + offB[n][0] + sh);
#pragma unroll
          for (int j = 1; j < MAXM; ++j) sum += (TC) * reinterpret_cast<const TIn*>(smem + offB[n][j] + sh);
          bb[n] = sum;
        }
        const bool last = g == GROUPS - 1;
        if (issue_now && !last) {
#pragma unroll
          for (int q = 0; q < GA_MAXPPW; ++q)
            if (q < ppw && group_of_piece(q) == g) {
              issue_piece(it + ahead, q);
              ++issued_before;
            }
        }
        if (last) {
          // the stage's barrier IN FRONT of the MFMAs of its last group: their operands are in registers once the LDS
          // reads have returned, so the slot is free, and the MFMAs run while the waves meet (as in the panel kernel)
          if (ahead > 1 && it + 2 < n_it && it + 2 != irr_seq) wait_vmcnt_dyn<8>(issued_before); else wait_vmcnt<0>();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          if (issue_now && ahead > 1) {
#pragma unroll
            for (int q = 0; q < GA_MAXPPW; ++q)
              if (q < ppw && group_of_piece(q) == g) issue_piece(it + ahead, q);
          }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < NACC; ++n) acc[m][n] = M::mma(a[m], bb[n], acc[m][n]);
      }
    }
  }

  TC* slab = slabs + ((int64_t)tile_lin * ksplit + ks) * (TILE * TILE);
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < NACC; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wm * 64 + m * 16 + M::row(lane, r);
        const int col = wn * WCOLS + n * 16 + (lane & 15);
        slab[row * TILE + col] = acc[m][n][r];
      }
}

// tile_table[k] = (ti << 16) | tj of the k-th upper-triangle tile in 8x8 super-block order
// (tiles with tj < first_tile -- a leading block the caller already has -- are left out), i.e. the order of
//     for si, for sj >= si, for ti in super-row si, for tj in super-column sj: keep (ti, tj) if tj >= ti, tj >= first_tile
// ONE workgroup of 4 waves: a wave takes one super-block per pass (lane = its 64 candidates, kept ones ranked by a
// ballot), the passes are chained through a running count.  (Until round 3 this was one thread walking the loop
// nest: 42 us per Gram launch at nt1 = 24, 2.7 ms per step of BASELINE config 4's 64 launches.)
__global__ __launch_bounds__(256) void build_tile_table_kernel(int32_t nt1, int32_t* __restrict__ table,
                                                              int32_t first_tile = 0) {
  __shared__ int wave_count[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t nsb = (nt1 + 7) / 8, n_super = nsb * (nsb + 1) / 2;
  int64_t base = 0;
  for (int64_t b0 = 0; b0 < n_super; b0 += 4) {
    const int64_t sb = b0 + wave;
    bool keep = false;
    int ti = 0, tj = 0;
    if (sb < n_super) {
      // super-block sb -> (si, sj): row si starts at si nsb - si (si - 1) / 2
      const double w = 2.0 * (double)nsb + 1.0;
      int64_t si = (int64_t)((w - sqrt(w * w - 8.0 * (double)sb)) * 0.5);
      if (si < 0) si = 0;
      if (si >= nsb) si = nsb - 1;
      while (si > 0 && si * nsb - si * (si - 1) / 2 > sb) --si;
      while (si + 1 < nsb && (si + 1) * nsb - (si + 1) * si / 2 <= sb) ++si;
      const int64_t sj = si + (sb - (si * nsb - si * (si - 1) / 2));
      ti = (int)si * 8 + (lane >> 3);
      tj = (int)sj * 8 + (lane & 7);
      keep = ti < nt1 && tj < nt1 && tj >= ti && tj >= first_tile;
    }
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) wave_count[wave] = __popcll(mask);
    __syncthreads();
    int64_t off = base;
    for (int w2 = 0; w2 < wave; ++w2) off += wave_count[w2];
    if (keep) table[off + __popcll(mask & ((1ull << lane) - 1ull))] = (ti << 16) | tj;
    base += wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// Pair-tile LDS-DMA kernel: one workgroup of 8 waves = TWO 128x128 output units.
//
// Why: with one 128x128 unit per 4-wave workgroup (2 workgroups per CU) the kernel is bound by the
// number of cache-line misses the CU's vector L1 keeps in flight, not by MFMA, LDS or HBM
// (tools/clock_probe.hip: 283 ms without DMAs, 293 ms with DMAs that hit the L1, 304 ms with L2 hits,
// 317 ms shipped; the waves stall in the DMA *issue*, for about one memory latency per stage).  The
// lever is bytes per flop through the L1: the upper wave group computes unit (ua_i, ua_j), the lower
// group unit (ub_i, ub_j), and the panels they have in common are staged once.  Regular entries are
// (2I, j) + (2I+1, j) with j >= 2I+2 -- 3 panels for 2 units, 25 % fewer bytes; the 3 units per
// row pair that touch the diagonal are paired among themselves, so no flop is wasted.
// Same slab format as the unit kernels (gram_reduce_kernel is shared).
constexpr int PAIR_THREADS = 512;
constexpr int PAIR_SLOTS = 4;   // distinct panels staged per LDS stage (at most)
constexpr int PAIR_NBUF = 3;    // ring depth: 3 x 4 x 12.8 KB = 153.6 KB of the 160 KB

struct PairEntry {
  int32_t a, b;  // (i | j << 16) of the upper / lower unit; b = -1: no lower unit
};

// number of entries for nt1 unit rows (host and device agree by construction)
__host__ __device__ inline int pair_entry_count(int nt1) {
  const int nrb = nt1 / 2;
  int64_t regular = 0;
  for (int I = 0; I < nrb; ++I) regular += nt1 - (2 * I + 2) > 0 ? nt1 - (2 * I + 2) : 0;
  const int left = 3 * nrb + (nt1 & 1);
  return (int)(regular + (left + 1) / 2);
}

// entries in XCD-friendly order: 4 row pairs x 8 columns = 32 consecutive regular entries share
// 8 + 8 panels; leftovers (diagonal units) at the end, paired in sequence
__global__ void build_pair_table_kernel(int32_t nt1, PairEntry* __restrict__ table) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int nrb = nt1 / 2;
  int k = 0;
  for (int si = 0; si * 4 < nrb; ++si)
    for (int sj = 0; sj * 8 < nt1; ++sj)
      for (int I = si * 4; I < si * 4 + 4 && I < nrb; ++I)
        for (int j = sj * 8; j < sj * 8 + 8 && j < nt1; ++j)
          if (j >= 2 * I + 2) {
            table[k].a = (2 * I) | (j << 16);
            table[k].b = (2 * I + 1) | (j << 16);
            ++k;
          }
  int pending = -1;
  auto leftover = [&](int i, int j) {
    const int u = i | (j << 16);
    if (pending < 0) {
      pending = u;
    } else {
      table[k].a = pending;
      table[k].b = u;
      ++k;
      pending = -1;
    }
  };
  for (int I = 0; I < nrb; ++I) {
    leftover(2 * I, 2 * I);
    leftover(2 * I, 2 * I + 1);
    leftover(2 * I + 1, 2 * I + 1);
  }
  if (nt1 & 1) leftover(nt1 - 1, nt1 - 1);
  if (pending >= 0) {
    table[k].a = pending;
    table[k].b = -1;
  }
}


template <typename T, int ABL = 0>
__global__ __launch_bounds__(PAIR_THREADS, 1) void gram_pair_dma_kernel(
    const T* __restrict__ X, int64_t n_rows, int64_t ld, int32_t nt1, int32_t n_entries, int32_t ksplit,
    const PairEntry* __restrict__ table, int64_t frames_per_split, T* __restrict__ slabs) {
  using M = Mfma<T>;
  using acc_t = typename M::acc_t;
  constexpr int KB = GramCfg<T>::KB;
  constexpr int RP = DmaCfg<T>::ROW_PIECES;
  constexpr int PE = DmaCfg<T>::PIECE_ELEMS;
  constexpr int SLOT_PIECES = KB * RP;                         // 12 (f64) / 16 (f32)
  constexpr int MAXPPW = (PAIR_SLOTS * SLOT_PIECES + 7) / 8;   // 6 / 8
  constexpr int PANEL_ELEMS = KB * ROW_STRIDE;
  constexpr int BUF_ELEMS = PAIR_SLOTS * PANEL_ELEMS;
  constexpr int NBUF = PAIR_NBUF;
  constexpr int AHEAD = NBUF - 1;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;              // 0: upper unit, 1: lower unit
  const int wm = (wave >> 1) & 1, wn = wave & 1;

  // workgroup -> (split, entry), XCD-aware: workgroups b and b+8 share an XCD, which runs 32 of
  // them at a time (one per CU); give each XCD 32 consecutive entries of the list
  const int b = blockIdx.x;
  const int v = (((b >> 3) >> 5) * 8 + (b & 7)) * 32 + ((b >> 3) & 31);
  if (v >= ksplit * n_entries) return;  // grid is padded to a multiple of 256
  const int ks = v / n_entries;
  const PairEntry ent = table[v - ks * n_entries];
  const int ua_i = ent.a & 0xffff, ua_j = (ent.a >> 16) & 0xffff;
  const bool has_b = ent.b >= 0;
  const int ub_i = has_b ? (ent.b & 0xffff) : ua_i, ub_j = has_b ? ((ent.b >> 16) & 0xffff) : ua_j;

  // distinct panels -> LDS slots (all wave-uniform, straight-line: nothing may end up in scratch,
  // whose loads would count against vmcnt)
  const int s0 = ua_i;
  int s1 = -1, s2 = -1, s3 = -1, ns = 1;
  const int a_up = 0;
  int b_up = 0, a_lo = 0, b_lo = 0;
  if (ua_j != s0) { s1 = ua_j; b_up = 1; ns = 2; }
  if (ub_i == s0) a_lo = 0;
  else if (ns > 1 && ub_i == s1) a_lo = 1;
  else if (ns == 1) { s1 = ub_i; a_lo = 1; ns = 2; }
  else { s2 = ub_i; a_lo = 2; ns = 3; }
  if (ub_j == s0) b_lo = 0;
  else if (ns > 1 && ub_j == s1) b_lo = 1;
  else if (ns > 2 && ub_j == s2) b_lo = 2;
  else if (ns == 1) { s1 = ub_j; b_lo = 1; ns = 2; }
  else if (ns == 2) { s2 = ub_j; b_lo = 2; ns = 3; }
  else { s3 = ub_j; b_lo = 3; ns = 4; }
  const int slotA = grp ? a_lo : a_up, slotB = grp ? b_lo : b_up;
  const bool active = grp == 0 || has_b;
  const int my_i = grp ? ub_i : ua_i, my_j = grp ? ub_j : ua_j;

  const int64_t t_begin = (int64_t)ks * frames_per_split;
  int64_t t_end = t_begin + frames_per_split;
  if (t_end > n_rows) t_end = n_rows;
  const int n_it = t_begin < t_end ? (int)((t_end - t_begin + KB - 1) / KB) : 0;

  // this wave's DMA pieces: piece p = wave + 8 q -> (slot, row, column piece)
  const int n_pieces = ns * SLOT_PIECES;
  const int my_cnt = n_pieces > wave ? (n_pieces - wave + 7) / 8 : 0;
  int64_t g_off[MAXPPW];
  int l_off[MAXPPW];
#pragma unroll
  for (int q = 0; q < MAXPPW; ++q) {
    const int p = wave + 8 * q;
    const int slot = p / SLOT_PIECES;
    const int r = (p - slot * SLOT_PIECES) / RP;
    const int cp = p % RP;
    const int panel = slot == 0 ? s0 : slot == 1 ? s1 : slot == 2 ? s2 : s3;
    g_off[q] = (int64_t)r * ld + (int64_t)panel * ROW_ELEMS + cp * PE + lane * (16 / (int)sizeof(T));
    l_off[q] = slot * PANEL_ELEMS + r * ROW_STRIDE + cp * PE;
  }

  // stage order rotated per entry (see gram_tile_dma_kernel); the one ragged stage of the last
  // split issues fewer DMAs, so the counted wait must not assume a full newest stage there
  const int skew = n_it > 16 ? ((ua_i + ua_j) & 7) : 0;
  auto stage_of = [&](int seq) { const int u = seq + skew; return u >= n_it ? u - n_it : u; };
  const bool ragged = n_it > 0 && (t_end - t_begin) % KB != 0;
  const int ragged_seq = ragged ? (n_it - 1 - skew + (n_it - 1 - skew < 0 ? n_it : 0)) : -1;
  auto prep_stage = [&](int seq) {
    const int64_t t0 = t_begin + (int64_t)stage_of(seq) * KB;
    if (t0 + KB > t_end) {
      T* lbase = smem + (seq % NBUF) * BUF_ELEMS;
      const int first = (int)(t_end - t0);
      const int per_slot = (KB - first) * ROW_ELEMS;
      for (int e = tid; e < ns * per_slot; e += PAIR_THREADS) {
        const int slot = e / per_slot;
        const int rem = e - slot * per_slot;
        const int r = first + rem / ROW_ELEMS, c = rem % ROW_ELEMS;
        lbase[slot * PANEL_ELEMS + r * ROW_STRIDE + c] = 0;
      }
    }
  };
  auto issue_stage = [&](int seq) {
    prep_stage(seq);
    const int64_t t0 = t_begin + (int64_t)stage_of(seq) * KB;
#pragma unroll
    for (int q = 0; q < MAXPPW; ++q) {
      const int p = wave + 8 * q;
      const int r = (p % SLOT_PIECES) / RP;
      const bool row_ok = t0 + r < t_end;
      const bool lane_ok = !(sizeof(T) == 4 && (p % RP) == 1) || lane < 32;
      if (p < n_pieces && row_ok && lane_ok) {
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(X + t0 * ld + g_off[q]),
            (__attribute__((address_space(3))) void*)(smem + (seq % NBUF) * BUF_ELEMS + l_off[q]), 16, 0, 0);
      }
    }
  };
  // wait until everything but the newest stage's pieces has landed (newest = seq_newest)
  auto wait_landed = [&](int seq_newest) {
    if (seq_newest >= n_it || seq_newest == ragged_seq) wait_vmcnt<0>();
    else wait_vmcnt_dyn<MAXPPW>(my_cnt);
  };

  acc_t acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = acc_zero<T>();

  const int offA = slotA * PANEL_ELEMS + (lane >> 4) * ROW_STRIDE + 3 * (wm * 64 + (lane & 15));
  const int offB = slotB * PANEL_ELEMS + (lane >> 4) * ROW_STRIDE + 3 * (wn * 64 + (lane & 15));

  if (n_it > 0) issue_stage(0);
  if (n_it > 1) issue_stage(1);
  wait_landed(1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  for (int it = 0; it < n_it; ++it) {
    if (ABL == 0 && it + AHEAD < n_it) issue_stage(it + AHEAD);
    const T* pa = smem + (ABL ? it % 2 : it % NBUF) * BUF_ELEMS;
    if (active) {
#pragma unroll
      for (int kk = 0; kk < KB / 4; ++kk) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          T a[4], bb[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) a[m] = pa[offA + kk * 4 * ROW_STRIDE + 48 * m + d];
#pragma unroll
          for (int n = 0; n < 4; ++n) bb[n] = pa[offB + kk * 4 * ROW_STRIDE + 48 * n + d];
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = M::mma(a[m], bb[n], acc[m][n]);
        }
      }
    }
    // stage it+1 must have landed before anyone reads it; stage it+2 may stay in flight
    if (ABL < 2) {
      wait_landed(it + 2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }

  if (!active) return;
  const int tile_lin = my_i * nt1 - my_i * (my_i - 1) / 2 + (my_j - my_i);
  T* slab = slabs + ((int64_t)tile_lin * ksplit + ks) * (TILE * TILE);
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wm * 64 + m * 16 + M::row(lane, r);
        const int col = wn * 64 + n * 16 + (lane & 15);
        slab[row * TILE + col] = acc[m][n][r];
      }
}

// ---------------------------------------------------------------------------
// G[(ti,tj) tile] (+)= sum_ks slab, and the mirrored tile.  grid = (n_tiles, 16): each
// workgroup handles 8 rows of a tile... (16 row-groups of 8 rows x 128 cols, 256 threads,
// 4 elements per thread).
template <typename T>
__global__ __launch_bounds__(256) void gram_reduce_kernel(const T* __restrict__ slabs,
                                                          int32_t nt1, int32_t ksplit,
                                                          int32_t n_red, int accumulate,
                                                          double* __restrict__ G, int32_t first_tile = 0) {
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
  if (tj < first_tile) return;  // a tile of the leading block the caller keeps: no slabs were written for it
  const T* base = slabs + (int64_t)tile_lin * ksplit * (TILE * TILE);
  const int r0 = blockIdx.y * 8;
  for (int e = threadIdx.x; e < 8 * TILE; e += blockDim.x) {
    const int row = r0 + e / TILE, col = e % TILE;
    double s = 0.0;
    for (int ks = 0; ks < ksplit; ++ks)
      s += (double)base[(int64_t)ks * (TILE * TILE) + row * TILE + col];
    const int gi = ti * TILE + row, gj = tj * TILE + col;
    if (gi < n_red && gj < n_red) {
      if (ti != tj || gj >= gi) {
        double* p = G + (int64_t)gi * n_red + gj;
        *p = accumulate ? *p + s : s;
      }
      if (ti != tj || gj > gi) {
        // mirror (for diagonal tiles only the strict upper part is mirrored so that
        // G stays exactly symmetric)
        double* p = G + (int64_t)gj * n_red + gi;
        *p = accumulate ? *p + s : s;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Small systems: n_red <= 128 (one output tile), e.g. CLN025 (175 atoms, 97 reduced variables).
// There the Gram build is HBM-bound (n_red (n_red+1) / (N s) = 6.8 flop/B at CLN025 against a machine
// balance of ~10), so the kernel is organised around ONE pass over the forces, read the way they lie
// in HBM: the 8 frames of a stage are one contiguous run of 8 x 3N elements, fetched with 16-byte
// loads into registers while the MFMAs of the previous stage run (2 workgroups per CU x 8 frames:
// ~64 KB per CU in flight), parked in LDS as they are, and only there turned into the panel the
// MFMAs read -- constraint-group column sums (`@ con_mat`), dtype conversion and zero padding, all
// LDS -> LDS.  No packed copy of the trajectory exists (the pack + tile pipeline above reads F,
// writes a padded copy and reads that again: 3.4x the bytes at CLN025), and only the 16x16 blocks
// of the upper triangle are multiplied (8 waves, <= 5 blocks each, dealt round-robin).
// Two shapes: 8 frames per stage x 8 waves (default: 60 KB of LDS at CLN025, two workgroups per CU, 67 KB in
// flight per CU) and 4 frames x 4 waves (AGGF_GRAM_SMALL=4: three smaller workgroups per CU; measured slower,
// 10.2 against 7.5 ms at CLN025 x 4e6 frames before the group sums lost their loops).
constexpr int SM_MAXVEC = 8;        // 16-byte loads per thread and stage (template NV = 3, 5 or 8)
constexpr int SM_FAST_MEMBERS = 4;  // group members summed without a loop (larger groups: generic tail loop)

template <typename TIn>
static size_t small_raw_bytes(int32_t N, int kbs) { return (size_t)round_up((int64_t)kbs * 3 * N * sizeof(TIn), 16) + 16; }

template <typename TIn, typename TC, int NV, int KBS, int NWV>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 3 : 4) void gram_small_kernel(
    const TIn* __restrict__ F, int64_t T, int32_t N, const int32_t* __restrict__ grp_ptr,
    const int32_t* __restrict__ grp_atoms, int32_t n_red, int64_t frames_per_split, int32_t raw_bytes,
    TC* __restrict__ slabs) {
  using M = Mfma<TC>;
  using acc_t = typename M::acc_t;
  constexpr int SM_THREADS = 64 * NWV;
  constexpr int SM_ENT = KBS * ROW_ELEMS / SM_THREADS;      // panel entries per thread and stage: 6
  constexpr int SM_MAXBLK = (36 + NWV - 1) / NWV;           // upper-triangle blocks per wave: 9 / 5
  static_assert(KBS * ROW_ELEMS % SM_THREADS == 0, "entry split");
  typedef float __attribute__((ext_vector_type(4))) v16_t;  // one 16-byte piece, whatever the dtype
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  TC* panel = reinterpret_cast<TC*>(smem_raw);                         // [SM_KB][ROW_STRIDE]
  TIn* raw = reinterpret_cast<TIn*>(smem_raw + KBS * ROW_STRIDE * sizeof(TC));  // SM_KB frames as in HBM
  // (raw_bytes includes one extra zeroed 16-byte piece: the "no member" slot of the table below)
  int32_t* atoms_s = reinterpret_cast<int32_t*>(smem_raw + KBS * ROW_STRIDE * sizeof(TC) + raw_bytes);  // [N]
  int32_t* ptr_s = atoms_s + N;                                                                           // [129]
  // per panel column c = 3 g + d: offsets (3 atom + d) of the first 4 members of group g inside a frame,
  // 0xFFFF = none -- one 8-byte LDS read instead of a chain of dependent ones per member
  unsigned short* memb_s = reinterpret_cast<unsigned short*>(smem_raw + KBS * ROW_STRIDE * sizeof(TC) + raw_bytes +
                                                             (((int64_t)N + TILE + 1) * 4 + 15) / 16 * 16);  // [384][4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Stages are dealt round-robin to the workgroups (stage = blockIdx.x + k * gridDim.x): the workgroups that run
  // at the same time then read one contiguous window of the trajectory.  With a private contiguous range per
  // workgroup, 512 far-apart streams hit the HBM channels at once and the row-buffer locality is gone
  // (measured: 2.2 TB/s at CLN025 however the loads were issued).
  (void)frames_per_split;
  const int64_t n_stage_all = (T + KBS - 1) / KBS;
  const int n_it = blockIdx.x < n_stage_all ? (int)((n_stage_all - 1 - blockIdx.x) / gridDim.x + 1) : 0;
  const int64_t t_end = T;
  auto stage_t0 = [&](int k) { return ((int64_t)blockIdx.x + (int64_t)k * gridDim.x) * KBS; };
  const int64_t row_in = (int64_t)N * 3;
  // column -> member atoms (CSR) in LDS; without constraint groups column g is atom g
  for (int a = tid; a < N; a += SM_THREADS) atoms_s[a] = grp_atoms ? grp_atoms[a] : a;
  for (int g = tid; g <= TILE; g += SM_THREADS) ptr_s[g] = g <= n_red ? (grp_ptr ? grp_ptr[g] : g) : (grp_ptr ? grp_ptr[n_red] : n_red);
  const int zero_idx = (raw_bytes - 16) / (int)sizeof(TIn);
  if (tid < 16 / (int)sizeof(TIn)) raw[zero_idx + tid] = (TIn)0;
  __syncthreads();
  bool big_groups = false;  // uniform: some group has more than SM_FAST_MEMBERS members
  for (int g = 0; g < n_red; ++g) big_groups |= ptr_s[g + 1] - ptr_s[g] > SM_FAST_MEMBERS;
  for (int c = tid; c < ROW_ELEMS; c += SM_THREADS) {
    const int g = c / 3, d = c - 3 * g;
#pragma unroll
    for (int j = 0; j < SM_FAST_MEMBERS; ++j)
      memb_s[c * 4 + j] = (ptr_s[g] + j < ptr_s[g + 1]) ? (unsigned short)(3 * atoms_s[ptr_s[g] + j] + d) : (unsigned short)0xFFFF;
  }

  // stage s+1 travels HBM -> registers (16-byte pieces of the contiguous run of frames) during the
  // MFMAs of stage s.  fps is a multiple of 8 frames, so every stage starts 16-byte aligned.
  const int n_vec = raw_bytes / 16 - 1;  // (the last piece is the zero slot)
  v16_t hold[NV];
  auto fetch = [&](int s) {
    const int64_t t0 = stage_t0(s);
    const int64_t valid = (t_end - t0 < KBS ? t_end - t0 : KBS) * row_in * (int64_t)sizeof(TIn);  // bytes
    const char* src = reinterpret_cast<const char*>(F + t0 * row_in);
    if (valid >= (int64_t)n_vec * 16) {
      // every frame of the stage exists (all stages but the last of the trajectory): straight-line loads.  With the
      // ragged-end handling in the same loop the compiler put an s_waitcnt vmcnt(0) after EVERY load (the paths
      // join in a phi): the loads of a stage went out one memory latency apart (tools/small_probe.hip: 6500 of
      // 13600 cycles per stage were spent "issuing" five loads)
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int v = tid + SM_THREADS * i;
        v16_t x = {0.f, 0.f, 0.f, 0.f};
        if (v < n_vec) x = *reinterpret_cast<const v16_t*>(src + (int64_t)v * 16);
        hold[i] = x;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int v = tid + SM_THREADS * i;
        v16_t x = {0.f, 0.f, 0.f, 0.f};
        if (v < n_vec) {
          const int64_t off = (int64_t)v * 16;
          if (off + 16 <= valid) {
            x = *reinterpret_cast<const v16_t*>(src + off);
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
    for (int i = 0; i < NV; ++i) {
      const int v = tid + SM_THREADS * i;
      if (v < n_vec) reinterpret_cast<v16_t*>(raw)[v] = hold[i];
    }
  };
  // raw frames -> MFMA panel: group sums in the compute dtype (frames past t_end were fetched as zeros)
  auto reduce_groups = [&]() {
    // all entries' table reads first, then all raw reads, then the sums: six independent LDS chains in flight at
    // once (with the rare-large-group loop inside the per-entry body the chains ran one after the other)
    TC sum[SM_ENT];
    uint2 mem[SM_ENT];
#pragma unroll
    for (int i = 0; i < SM_ENT; ++i) {
      const int e = tid + SM_THREADS * i;  // (frame in stage, reduced column, xyz); padding columns sum nothing
      mem[i] = *reinterpret_cast<const uint2*>(memb_s + (e % ROW_ELEMS) * 4);
    }
#pragma unroll
    for (int i = 0; i < SM_ENT; ++i) {
      const int e = tid + SM_THREADS * i;
      const int base = (e / ROW_ELEMS) * (int)row_in;
      const int o0 = mem[i].x & 0xFFFF, o1 = mem[i].x >> 16, o2 = mem[i].y & 0xFFFF, o3 = mem[i].y >> 16;
      const TC v0 = (TC)raw[o0 == 0xFFFF ? zero_idx : base + o0], v1 = (TC)raw[o1 == 0xFFFF ? zero_idx : base + o1],
               v2 = (TC)raw[o2 == 0xFFFF ? zero_idx : base + o2], v3 = (TC)raw[o3 == 0xFFFF ? zero_idx : base + o3];
      sum[i] = ((v0 + v1) + v2) + v3;  // members in CSR order, like the column sum of `@ con_mat`
    }
    if (big_groups) {
#pragma unroll
      for (int i = 0; i < SM_ENT; ++i) {
        const int e = tid + SM_THREADS * i;
        const int r = e / ROW_ELEMS, c = e - r * ROW_ELEMS;
        const int g = c / 3, d = c - 3 * g;
        for (int j = ptr_s[g] + SM_FAST_MEMBERS; j < ptr_s[g + 1]; ++j) sum[i] += (TC)raw[r * (int)row_in + 3 * atoms_s[j] + d];
      }
    }
#pragma unroll
    for (int i = 0; i < SM_ENT; ++i) {
      const int e = tid + SM_THREADS * i;
      const int r = e / ROW_ELEMS, c = e - r * ROW_ELEMS;
      panel[r * ROW_STRIDE + c] = sum[i];
    }
  };

  // this wave's 16x16 blocks of the upper triangle: q = wave, wave + NWV, ... in row-major order
  const int nb = (n_red + 15) / 16;
  int b_i[SM_MAXBLK], b_j[SM_MAXBLK];
#pragma unroll
  for (int k = 0; k < SM_MAXBLK; ++k) {
    int q = wave + NWV * k, bi = 0, rowlen = nb;
    while (bi < nb && q >= rowlen) {
      q -= rowlen;
      --rowlen;
      ++bi;
    }
    b_i[k] = bi < nb ? bi : -1;
    b_j[k] = bi + q;
  }
  acc_t acc[SM_MAXBLK];
#pragma unroll
  for (int k = 0; k < SM_MAXBLK; ++k) acc[k] = acc_zero<TC>();
  const int off = (lane >> 4) * ROW_STRIDE + 3 * (lane & 15);

  // De-phase the workgroups that share a CU: they do identical work, so two that start together stay in lock-step
  // and sit in the same phase (fetch wait / LDS group sums / MFMA) at the same time -- the three phases then add up
  // instead of overlapping (measured: HBM 2.1 + LDS 2.2 + MFMA 2.2 ms against 6.2 ms total).  A pseudo-random start
  // delay of up to ~one stage, as in the tile kernel.
  {
    const unsigned h = ((unsigned)blockIdx.x * 2654435761u) >> 25;  // 0..127
    for (unsigned i = 0; i < h; ++i) __builtin_amdgcn_s_sleep(1);   // 64 clocks each
  }
#ifdef AGGF_SMALL_PROF
  uint64_t pf[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
  if (n_it > 0) fetch(0);
  for (int s = 0; s < n_it; ++s) {
    AGGF_SP_T(q0);
    // (no barrier here: `raw` was last read by the group sums of stage s-1, in front of that stage's third barrier;
    // the panel, which the MFMAs of stage s-1 may still be reading, is not written before the next barrier)
    AGGF_SP_T(q1);
    park();
    AGGF_SP_T(q2);
    __syncthreads();
    AGGF_SP_T(q3);
    if (s + 1 < n_it) fetch(s + 1);
    AGGF_SP_T(q3b);
    reduce_groups();
    AGGF_SP_T(q4);
    __syncthreads();
    AGGF_SP_T(q5);
#pragma unroll
    for (int kk = 0; kk < KBS / 4; ++kk)
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int k = 0; k < SM_MAXBLK; ++k)
          if (b_i[k] >= 0) {
            // (one LDS round trip per MFMA.  Tried: all operands of a group read first -- 20 more VGPRs, spills,
            // 5.8 -> 9.0 ms; one item ahead -- 10 spills, 6.5 ms; a compile-time block count with straight-line code --
            // the scheduler hoists every read: 79-108 spilled VGPRs.  The staged frames take the registers.)
            const TC a = panel[off + kk * 4 * ROW_STRIDE + 48 * b_i[k] + d];
            const TC b = panel[off + kk * 4 * ROW_STRIDE + 48 * b_j[k] + d];
            acc[k] = M::mma(a, b, acc[k]);
          }
#ifdef AGGF_SMALL_PROF
    const uint64_t q6 = __builtin_readcyclecounter();
    pf[0] += q1 - q0;  // barrier 1
    pf[1] += q2 - q1;  // park (waits for the global loads)
    pf[2] += q3 - q2;  // barrier 2
    pf[3] += q4 - q3b;  // group sums
    pf[6] += q3b - q3;  // fetch issue
    pf[4] += q5 - q4;  // barrier 3
    pf[5] += q6 - q5;  // MFMA phase
#endif
  }
#ifdef AGGF_SMALL_PROF
  if (lane == 0) {
    for (int i = 0; i < 7; ++i) atomicAdd(&aggf_small_prof[i], (unsigned long long)pf[i]);
    atomicAdd(&aggf_small_prof[7], (unsigned long long)n_it);
    atomicAdd(&aggf_small_prof[8], 1ull);
  }
#endif
  TC* slab = slabs + (int64_t)blockIdx.x * (TILE * TILE);
#pragma unroll
  for (int k = 0; k < SM_MAXBLK; ++k)
    if (b_i[k] >= 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(b_i[k] * 16 + M::row(lane, r)) * TILE + b_j[k] * 16 + (lane & 15)] = acc[k][r];
    }
}

// ---------------------------------------------------------------------------
// The same small-system kernel with the frames travelling HBM -> LDS by LDS-DMA into a 3-stage ring.
// Why: with register staging the loads of a stage are issued in one burst and waited for before the next
// burst goes out, so the bytes in flight per CU (2 x 33 KB) are only in flight part of the time: 2.2 TB/s
// at CLN025 (7.5 ms for 4e6 frames), bound by latency.  With the ring, two stages per workgroup are in flight
// WHILE a third is being consumed, nothing is held in registers, and the waves wait on a counted vmcnt exactly
// like K1.  4 frames per stage, 8 waves, 2 workgroups per CU (67 KB of LDS at CLN025).
constexpr int SD_KB = 4, SD_NW = 8, SD_NBUF = 3, SD_THREADS = 64 * SD_NW;
constexpr int SD_ENT = SD_KB * ROW_ELEMS / SD_THREADS;  // 3
constexpr int SD_MAXBLK = (36 + SD_NW - 1) / SD_NW;     // 5
constexpr int SD_MAXPPW = 8;                            // DMA pieces per wave and stage (64 KB per stage)

// ABL (ablation, AGGF_SMALL_ABL; measurements only): 1 = no MFMAs, 2 = no group sums, 3 = no DMAs.
// NBW = 16x16 blocks per wave, a compile-time count: the upper-triangle blocks are dealt round-robin and a wave
// that gets one block fewer multiplies a spare copy of block (0,0) that is never stored.  With a run-time count
// every MFMA sat behind its own wave-uniform branch, its two operand reads and an s_waitcnt lgkmcnt(0) -- a chain
// of LDS round trips (PMC: MFMA pipes busy 44 %); with straight-line code the reads of a group are in flight
// together and the next group's are hoisted above the MFMAs.
template <typename TIn, typename TC, int ABL = 0, int NBW = SD_MAXBLK>
__global__ __launch_bounds__(SD_THREADS, 4) void gram_small_dma_kernel(
    const TIn* __restrict__ F, int64_t T, int32_t N, const int32_t* __restrict__ grp_ptr,
    const int32_t* __restrict__ grp_atoms, int32_t n_red, int64_t frames_per_split, int32_t raw_bytes,
    TC* __restrict__ slabs) {
  using M = Mfma<TC>;
  using acc_t = typename M::acc_t;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  TC* panel = reinterpret_cast<TC*>(smem_raw);                              // [SD_KB][ROW_STRIDE]
  char* ring = smem_raw + SD_KB * ROW_STRIDE * sizeof(TC);                  // [SD_NBUF][raw_bytes]: frames as in HBM
  TIn* zero_s = reinterpret_cast<TIn*>(ring + SD_NBUF * raw_bytes);         // 16 bytes of zeros ("no member")
  int32_t* atoms_s = reinterpret_cast<int32_t*>(ring + SD_NBUF * raw_bytes + 16);  // [N]
  int32_t* ptr_s = atoms_s + N;                                                      // [129]
  unsigned short* memb_s = reinterpret_cast<unsigned short*>(ring + SD_NBUF * raw_bytes + 16 +
                                                             (((int64_t)N + TILE + 1) * 4 + 15) / 16 * 16);  // [384][4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // stages dealt round-robin to the workgroups (see gram_small_kernel): co-running workgroups read one window
  (void)frames_per_split;
  const int64_t n_stage_all = (T + SD_KB - 1) / SD_KB;
  const int n_it = blockIdx.x < n_stage_all ? (int)((n_stage_all - 1 - blockIdx.x) / gridDim.x + 1) : 0;
  const int64_t t_end = T;
  auto stage_t0 = [&](int k) { return ((int64_t)blockIdx.x + (int64_t)k * gridDim.x) * SD_KB; };
  const int64_t row_in = (int64_t)N * 3;
  const int64_t row_bytes = row_in * (int64_t)sizeof(TIn);

  for (int a = tid; a < N; a += SD_THREADS) atoms_s[a] = grp_atoms ? grp_atoms[a] : a;
  for (int g = tid; g <= TILE; g += SD_THREADS) ptr_s[g] = g <= n_red ? (grp_ptr ? grp_ptr[g] : g) : (grp_ptr ? grp_ptr[n_red] : n_red);
  if (tid < 16 / (int)sizeof(TIn)) zero_s[tid] = (TIn)0;
  __syncthreads();
  bool big_groups = false;
  for (int g = 0; g < n_red; ++g) big_groups |= ptr_s[g + 1] - ptr_s[g] > SM_FAST_MEMBERS;
  for (int c = tid; c < ROW_ELEMS; c += SD_THREADS) {
    const int g = c / 3, d = c - 3 * g;
#pragma unroll
    for (int j = 0; j < SM_FAST_MEMBERS; ++j)
      memb_s[c * 4 + j] = (ptr_s[g] + j < ptr_s[g + 1]) ? (unsigned short)(3 * atoms_s[ptr_s[g] + j] + d) : (unsigned short)0xFFFF;
  }

  // DMA pieces of a stage: piece p = bytes [1024 p, 1024 (p+1)) of the contiguous run of SD_KB frames;
  // wave w issues pieces w, w + 8, ...; my_pieces = how many of them exist (the same for every full stage)
  const int n_pieces = (raw_bytes + 1023) / 1024;
  const int my_pieces = wave < n_pieces ? (n_pieces - 1 - wave) / SD_NW + 1 : 0;
  // the last stage of the trajectory may have fewer than SD_KB frames; it is the last stage of whoever owns it
  const bool ragged = n_it > 0 && T % SD_KB != 0 && (n_stage_all - 1) % gridDim.x == blockIdx.x;
  auto issue_stage = [&](int s) {
    const int64_t t0 = stage_t0(s);
    const int64_t valid = (t_end - t0 < SD_KB ? t_end - t0 : SD_KB) * row_bytes;  // bytes that exist
    const char* src = reinterpret_cast<const char*>(F + t0 * row_in);
    char* dst = ring + (s % SD_NBUF) * raw_bytes;
    if (valid < raw_bytes) {
      // the ragged last stage: missing frames read as zeros; a 16-byte piece that straddles the end of the
      // data is copied element-wise instead of by DMA (no read past the end of the trajectory)
      const int64_t whole = valid / 16 * 16;
      for (int64_t b = whole + (int64_t)tid * sizeof(TIn); b < raw_bytes; b += (int64_t)SD_THREADS * sizeof(TIn))
        *reinterpret_cast<TIn*>(dst + b) = b < valid ? *reinterpret_cast<const TIn*>(src + b) : (TIn)0;
    }
#pragma unroll
    for (int q = 0; q < SD_MAXPPW; ++q) {
      const int p = wave + SD_NW * q;
      if (p < n_pieces) {
        const int64_t off = (int64_t)p * 1024 + lane * 16;
        if (ABL != 3 && off + 16 <= valid)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                           (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
      }
    }
  };
  // wait until at most `my_pieces` of this wave's DMAs (= the newest stage) are outstanding
  auto wait_older = [&](bool newest_is_full) {
    if (!newest_is_full) { wait_vmcnt<0>(); return; }
    switch (my_pieces) {
      case 0: wait_vmcnt<0>(); break;
      case 1: wait_vmcnt<1>(); break;
      case 2: wait_vmcnt<2>(); break;
      case 3: wait_vmcnt<3>(); break;
      case 4: wait_vmcnt<4>(); break;
      case 5: wait_vmcnt<5>(); break;
      case 6: wait_vmcnt<6>(); break;
      case 7: wait_vmcnt<7>(); break;
      default: wait_vmcnt<8>(); break;
    }
  };
  auto reduce_groups = [&](const TIn* raw) {
    // table reads of all entries, then all raw reads, then the sums: independent LDS chains in flight together
    TC sum[SD_ENT];
    uint2 mem[SD_ENT];
#pragma unroll
    for (int i = 0; i < SD_ENT; ++i) mem[i] = *reinterpret_cast<const uint2*>(memb_s + ((tid + SD_THREADS * i) % ROW_ELEMS) * 4);
#pragma unroll
    for (int i = 0; i < SD_ENT; ++i) {
      const int e = tid + SD_THREADS * i;  // (frame in stage, reduced column, xyz); padding columns sum nothing
      const TIn* fr = raw + (e / ROW_ELEMS) * (int)row_in;
      const int o0 = mem[i].x & 0xFFFF, o1 = mem[i].x >> 16, o2 = mem[i].y & 0xFFFF, o3 = mem[i].y >> 16;
      const TC v0 = (TC) * (o0 == 0xFFFF ? zero_s : fr + o0), v1 = (TC) * (o1 == 0xFFFF ? zero_s : fr + o1),
               v2 = (TC) * (o2 == 0xFFFF ? zero_s : fr + o2), v3 = (TC) * (o3 == 0xFFFF ? zero_s : fr + o3);
      sum[i] = ((v0 + v1) + v2) + v3;  // members in CSR order, like the column sum of `@ con_mat`
    }
    if (big_groups) {
#pragma unroll
      for (int i = 0; i < SD_ENT; ++i) {
        const int e = tid + SD_THREADS * i;
        const int r = e / ROW_ELEMS, c = e - r * ROW_ELEMS;
        const int g = c / 3, d = c - 3 * g;
        for (int j = ptr_s[g] + SM_FAST_MEMBERS; j < ptr_s[g + 1]; ++j) sum[i] += (TC)raw[r * (int)row_in + 3 * atoms_s[j] + d];
      }
    }
#pragma unroll
    for (int i = 0; i < SD_ENT; ++i) {
      const int e = tid + SD_THREADS * i;
      const int r = e / ROW_ELEMS, c = e - r * ROW_ELEMS;
      panel[r * ROW_STRIDE + c] = sum[i];
    }
  };

  const int nb = (n_red + 15) / 16;
  int b_i[NBW], b_j[NBW];
  bool b_live[NBW];
#pragma unroll
  for (int k = 0; k < NBW; ++k) {
    int q = wave + SD_NW * k, bi = 0, rowlen = nb;
    while (bi < nb && q >= rowlen) {
      q -= rowlen;
      --rowlen;
      ++bi;
    }
    b_live[k] = bi < nb;
    b_i[k] = b_live[k] ? bi : 0;
    b_j[k] = b_live[k] ? bi + q : 0;
  }
  acc_t acc[NBW];
#pragma unroll
  for (int k = 0; k < NBW; ++k) acc[k] = acc_zero<TC>();
  const int off = (lane >> 4) * ROW_STRIDE + 3 * (lane & 15);
  int addr_a[NBW], addr_b[NBW];
#pragma unroll
  for (int k = 0; k < NBW; ++k) {
    addr_a[k] = off + 48 * b_i[k];
    addr_b[k] = off + 48 * b_j[k];
  }

  __syncthreads();  // tables complete
  // De-phase the workgroups that share a CU: they do identical work, so two that start together stay in lock-step
  // and sit in the same phase (fetch wait / LDS group sums / MFMA) at the same time -- the three phases then add up
  // instead of overlapping (measured: HBM 2.1 + LDS 2.2 + MFMA 2.2 ms against 6.2 ms total).  A pseudo-random start
  // delay of up to ~one stage, as in the tile kernel.
  {
    const unsigned h = ((unsigned)blockIdx.x * 2654435761u) >> 25;  // 0..127
    for (unsigned i = 0; i < h; ++i) __builtin_amdgcn_s_sleep(1);   // 64 clocks each
  }
  if (n_it > 0) issue_stage(0);
  if (n_it > 1) issue_stage(1);
  wait_older(n_it > 1 && !(ragged && n_it == 2));  // stage 0 has landed (this wave's pieces)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  for (int it = 0; it < n_it; ++it) {
    if (ABL != 2) reduce_groups(reinterpret_cast<const TIn*>(ring + (it % SD_NBUF) * raw_bytes));
    __syncthreads();  // panel complete; slot (it + 2) % 3 was last read in iteration it - 1
    const bool more = it + 2 < n_it;
    if (more) issue_stage(it + 2);
    if (ABL != 1) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        TC av[NBW], bv[NBW];
#pragma unroll
        for (int k = 0; k < NBW; ++k) {
          av[k] = panel[addr_a[k] + d];
          bv[k] = panel[addr_b[k] + d];
        }
#pragma unroll
        for (int k = 0; k < NBW; ++k) acc[k] = M::mma(av[k], bv[k], acc[k]);
      }
    }
    // stage it + 1 must have landed; stage it + 2 may stay in flight (unless it is the ragged one: its DMA
    // count is not the usual one)
    wait_older(more && !(ragged && it + 2 == n_it - 1));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }
  TC* slab = slabs + (int64_t)blockIdx.x * (TILE * TILE);
#pragma unroll
  for (int k = 0; k < NBW; ++k)
    if (b_live[k]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(b_i[k] * 16 + M::row(lane, r)) * TILE + b_j[k] * 16 + (lane & 15)] = acc[k][r];
    }
}

// Slab sum of the single-tile (small-system) path: one workgroup per row of G instead of the generic kernel's 16
// workgroups per tile (those took 0.84 ms for the 512 slabs of CLN025 -- 15 % of the Gram build).  Thread =
// (column, parity of the slab index); fixed summation order; upper triangle written and mirrored.
template <typename T>
__global__ __launch_bounds__(256) void gram_reduce_small_kernel(const T* __restrict__ slabs, int32_t ksplit,
                                                                int32_t n_red, int accumulate,
                                                                double* __restrict__ G) {
  __shared__ double part[2][TILE];
  const int row = blockIdx.x, col = threadIdx.x & (TILE - 1), half = threadIdx.x >> 7;
  double s = 0.0;
  for (int ks = half; ks < ksplit; ks += 2) s += (double)slabs[((int64_t)ks * TILE + row) * TILE + col];
  part[half][col] = s;
  __syncthreads();
  if (half == 0 && row < n_red && col < n_red && col >= row) {
    const double tot = part[0][col] + part[1][col];
    double* p = G + (int64_t)row * n_red + col;
    *p = accumulate ? *p + tot : tot;
    if (col > row) {
      double* q = G + (int64_t)col * n_red + row;
      *q = accumulate ? *q + tot : tot;
    }
  }
}

// ---------------------------------------------------------------------------
enum GramStaging { STAGE_REG = 0, STAGE_DMA = 1, STAGE_PAIR = 2, STAGE_DMA8 = 3, STAGE_SMALL = 4, STAGE_QUAD = 5 };

struct GramPlan {
  int32_t n_pad, nt1, n_tiles;
  int32_t first_tile = 0;  // tiles with tj < first_tile are skipped (aggf_gram_from_column; DMA8 direct path only)
  int staging;         // GramStaging
  int32_t n_entries;   // work items per split: n_tiles (unit kernels) or pair entries
  bool direct;         // gram kernel reads F in place
  bool small_dma = false;  // STAGE_SMALL: LDS-DMA ring variant
  int ksplit;
  int64_t frames_per_split;
  int64_t chunk_frames;  // frames per pack chunk (direct: T)
  size_t slab_bytes, pack_bytes;
};

static int choose_ksplit(int n_tiles, int64_t frames, int kb, int slots, int64_t max_splits,
                         int units_per_wg = 1) {
  // Minimise a simple time model over the split count k: workgroups run in rounds of `slots`
  // (2 per CU); a workgroup costs its frames plus a fixed prologue/epilogue, and every
  // workgroup writes (and the reducer re-reads) one 128x128 slab.
  int64_t hi = ceil_div(frames, (int64_t)kb * 8);  // at least 8 stages per split
  if (hi < 1) hi = 1;
  if (hi > max_splits) hi = max_splits;
  if (hi > 1024) hi = 1024;
  static const char* force_k = getenv("AGGF_GRAM_KSPLIT");  // measurement: a fixed split count (tools/gram_ksplit_bench.py)
  if (force_k && atoi(force_k) > 0) return (int)(atoi(force_k) < hi ? atoi(force_k) : hi);
  const double us_per_frame = 0.64 * 4.0 / kb;     // one LDS stage = 48 MFMAs per wave, 2 waves per SIMD
  const double fixed_frames = 48.0;                // pipeline fill + slab store, in frame units
  const double slab_us = units_per_wg * 2.0 * TILE * TILE * (kb == 4 ? 8 : 4) / 2.0e6;  // write + read at ~2 TB/s
  double best_cost = 1e300;
  int64_t best = 1;
  for (int64_t k = 1; k <= hi; ++k) {
    const int64_t blocks = k * n_tiles;
    const double rounds = (double)ceil_div(blocks, slots);
    const double fpb = (double)round_up(ceil_div(frames, k), kb);
    const double cost = rounds * (fpb + fixed_frames) * us_per_frame + blocks * slab_us;
    if (cost < best_cost * (1.0 - 1e-6)) {
      best_cost = cost;
      best = k;
    }
  }
  return (int)best;
}

static size_t dtype_size(int dt) { return dt == AGGF_F64 ? 8 : 4; }
static size_t table_bytes(const GramPlan& p) { return (size_t)round_up((int64_t)p.n_tiles * 8, 256); }

// Default for both dtypes: LDS-DMA ring, 8 waves per tile, the stage's DMAs spread over the MFMA groups.
// fp64 at C3: 757 ms (4 waves with the DMAs up front 786 ms, register staging 810 ms); fp32 (panel
// rows are 1.5 DMA pieces, uneven shares per group): c2 3.12 ms against 3.26 ms register-staged, c5
// 60.7 against 60.3 ms.  AGGF_GRAM_STAGING = "8waves" | "dma" | "pair" | "reg" overrides (benchmarks, tests).
static int choose_staging(int compute_dtype, int nt1) {
  static const char* force = getenv("AGGF_GRAM_STAGING");
  int st = STAGE_DMA8;
  (void)compute_dtype;
  if (force) st = force[0] == 'p' ? STAGE_PAIR : force[0] == 'd' ? STAGE_DMA : force[0] == '8' ? STAGE_DMA8 : force[0] == 'q' ? STAGE_QUAD : STAGE_REG;
  if (st == STAGE_QUAD && compute_dtype != AGGF_F32) st = STAGE_DMA8;  // the quad shape exists for float32 products only
  if (st == STAGE_PAIR && nt1 < 2) st = STAGE_DMA;
  return st;
}

static int make_plan(int64_t T, int32_t N, int32_t n_red, int in_dtype, int compute_dtype,
                     bool has_groups, bool aligned, size_t ws_bytes, bool query, GramPlan* p, int32_t first_col = 0) {
  p->n_pad = (int32_t)round_up(n_red, TILE);
  p->nt1 = p->n_pad / TILE;
  p->n_tiles = p->nt1 * (p->nt1 + 1) / 2;
  p->direct = !has_groups && (N % TILE == 0) && in_dtype == compute_dtype && aligned;
  p->staging = choose_staging(compute_dtype, p->nt1);
  static const char* no_small = getenv("AGGF_GRAM_NO_SMALL");  // tests: force the tiled pipeline on small systems
  // small-system variants (AGGF_GRAM_SMALL): default "8" = register-staged, 8 frames x 8 waves (6.5 ms at CLN025 x
  // 4e6 frames); "dma" = LDS-DMA ring, 4 frames per stage (6.8-7.0 ms); "4" = register-staged 4 x 4 waves.
  // tools/c1_ablate.sh (DMA variant): 7.0 ms complete, 5.65 ms without any DMA, 4.2 ms without the MFMAs, 5.6 ms
  // without the group sums -- the three phases (HBM 2.1 ms at 8 TB/s, MFMA 2.2 ms for the 28 upper-triangle
  // blocks = 1.5x the algorithmic flops, LDS ~2.2 ms) are of equal size and overlap only partly.
  static const char* small_shape = getenv("AGGF_GRAM_SMALL");
  const bool small_dma_wanted = small_shape && small_shape[0] == 'd';
  const size_t raw4 = (size_t)SD_KB * 3 * N * dtype_size(in_dtype);
  const size_t lds_dma = (size_t)SD_KB * ROW_STRIDE * dtype_size(compute_dtype) + SD_NBUF * raw4 + 16 +
                         (size_t)round_up(((int64_t)N + TILE + 1) * 4, 16) + (size_t)ROW_ELEMS * 8;
  const bool small_dma = small_dma_wanted && raw4 % 16 == 0 && raw4 <= (size_t)SD_MAXPPW * SD_NW * 1024 && lds_dma <= 159 * 1024;
  const int small_kbs = small_dma ? SD_KB : (small_shape && small_shape[0] == '4') ? 4 : 8;
  const size_t raw_small = (size_t)round_up((int64_t)small_kbs * 3 * N * (int64_t)dtype_size(in_dtype), 16);
  // (16-byte loads per thread and stage <= SM_MAXVEC; 3 N + xyz must fit the 16-bit member table)
  if (p->nt1 == 1 && !no_small && (small_dma || raw_small <= (size_t)SM_MAXVEC * 64 * small_kbs * 16) && N < 21000 && aligned) {
    // one output tile: the fused streaming kernel (group sums + conversion on the way into LDS, upper
    // triangle blocks only); one slab per workgroup, ~4 workgroups per CU over the frame axis
    p->staging = STAGE_SMALL;
    p->n_entries = 1;
    p->direct = true;
    p->chunk_frames = T;
    p->pack_bytes = 0;
    p->small_dma = small_dma;
    // one resident generation of workgroups (2 per CU; 3 for the 4 x 4 shape), each looping over strided stages
    int64_t nwg = (int64_t)(small_kbs == 4 && !small_dma ? 3 : 2) * device_cu_count();
    const int64_t n_stage_all = ceil_div(T, small_kbs);
    if (nwg > n_stage_all) nwg = n_stage_all;
    const size_t slab1s = (size_t)TILE * TILE * dtype_size(compute_dtype);
    if (!query) {
      if (ws_bytes < table_bytes(*p) + slab1s + 512) return fail(AGGF_ERR_WORKSPACE, "gram workspace too small");
      const int64_t max_splits = (int64_t)((ws_bytes - table_bytes(*p) - 512) / slab1s);
      if (nwg > max_splits) nwg = max_splits;
    }
    p->frames_per_split = 0;
    p->ksplit = (int)nwg;
    p->slab_bytes = (size_t)p->ksplit * slab1s;
    return AGGF_OK;
  }
  const bool pair = p->staging == STAGE_PAIR;
  p->n_entries = pair ? pair_entry_count(p->nt1) : p->n_tiles;
  p->first_tile = 0;
  if (first_col >= TILE && p->staging == STAGE_DMA8 && p->direct) {
    p->first_tile = first_col / TILE;
    if (p->first_tile >= p->nt1) p->first_tile = p->nt1 - 1;  // always at least the last tile column
    p->n_entries = p->n_tiles - p->first_tile * (p->first_tile + 1) / 2;
  }
  const int upw = pair ? 2 : 1;
  const bool quad = p->staging == STAGE_QUAD;
  const int kb = quad ? Q_KB : compute_dtype == AGGF_F64 ? GramCfg<double>::KB : GramCfg<float>::KB;
  const size_t cs = dtype_size(compute_dtype);
  const int slots = (pair ? 1 : quad ? 4 : 2) * device_cu_count();
  const size_t slab1 = (size_t)p->n_tiles * TILE * TILE * cs;  // one split
  const size_t row_bytes = (size_t)p->n_pad * 3 * cs;
  if (query) {
    // recommended: slabs for the preferred split count + a pack chunk of up to 16 GiB (and at most a quarter of the
    // HBM that is free right now).  Round 2 capped the chunk at 1 GiB: 119 pack + table + tile + reduce launches at C3
    // with the pair constraints, 13 ms of slab sums and 2.5 ms of tile tables per step for nothing.
    p->chunk_frames = T;
    if (!p->direct) {
      size_t free_b = 0, total_b = 0;
      size_t cap_b = (size_t)16 << 30;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b / 4 < cap_b) cap_b = free_b / 4;
      if (cap_b < ((size_t)1 << 28)) cap_b = (size_t)1 << 28;
      int64_t cf = (int64_t)(cap_b / row_bytes);
      if (cf < 256) cf = 256;
      if (cf > T) cf = T;
      p->chunk_frames = cf > 0 ? cf : 1;
    }
    p->ksplit = choose_ksplit(p->n_entries, p->chunk_frames, kb, slots, 1 << 20, upw);
    p->slab_bytes = slab1 * p->ksplit;
    p->pack_bytes = p->direct ? 0 : round_up((int64_t)(p->chunk_frames * row_bytes), 256);
    return AGGF_OK;
  }
  // fit into the given workspace: the slabs of the split count the whole trajectory would like come first (at most
  // half of the space), the pack chunk takes what is left
  p->pack_bytes = 0;
  p->chunk_frames = T;
  if (ws_bytes < table_bytes(*p) + 1024) return fail(AGGF_ERR_WORKSPACE, "gram workspace too small");
  ws_bytes -= table_bytes(*p);
  size_t avail = ws_bytes;
  if (!p->direct) {
    size_t slab_budget = slab1 * (size_t)choose_ksplit(p->n_entries, T, kb, slots, 1 << 20, upw) + 1024;
    if (slab_budget > ws_bytes / 2) slab_budget = ws_bytes / 2;
    int64_t cf = (int64_t)((ws_bytes - slab_budget) / row_bytes);
    if (cf > T) cf = T;
    if (cf < 1) return fail(AGGF_ERR_WORKSPACE, "gram workspace too small for one packed frame");
    p->chunk_frames = cf;
    p->pack_bytes = (size_t)round_up((int64_t)(cf * row_bytes), 256);
    avail = ws_bytes - p->pack_bytes;
  }
  avail = avail > 512 ? avail - 512 : 0;  // room for the 256-byte roundings
  const int64_t max_splits = (int64_t)(avail / slab1);
  if (max_splits < 1) return fail(AGGF_ERR_WORKSPACE, "gram workspace too small for one slab set");
  p->ksplit = choose_ksplit(p->n_entries, p->chunk_frames, kb, slots, max_splits, upw);
  p->slab_bytes = slab1 * p->ksplit;
  return AGGF_OK;
}

template <typename T>
static int launch_gram(const T* X, int64_t rows, int64_t ld, const GramPlan& p, T* slabs,
                       int32_t* tile_table, double* G, int32_t n_red, int accumulate, hipStream_t stream) {
  constexpr int KB = GramCfg<T>::KB;
  const int ksplit = p.ksplit;
  int64_t fps = round_up(ceil_div(rows, ksplit), KB);
  if (fps < KB) fps = KB;
  const int64_t nblocks = (int64_t)ksplit * p.n_tiles;
  if (nblocks > 0x7fffffffLL) return fail(AGGF_ERR_ARG, "gram grid too large");
  if (p.staging == STAGE_PAIR) {
    const size_t lds = (size_t)PAIR_NBUF * PAIR_SLOTS * KB * ROW_STRIDE * sizeof(T);  // 153.6 KB
    static thread_local PerDeviceOnce attr_once;
    bool& attr_done = *attr_once.flag();
    if (!attr_done) {
      AGGF_HIP_OK(hipFuncSetAttribute((const void*)gram_pair_dma_kernel<T>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_done = true;
    }
    const int64_t nwg = (int64_t)ksplit * p.n_entries;
    if (nwg > 0x7fffff00LL) return fail(AGGF_ERR_ARG, "gram grid too large");
    PairEntry* table = reinterpret_cast<PairEntry*>(tile_table);
    hipLaunchKernelGGL(build_pair_table_kernel, dim3(1), dim3(1), 0, stream, p.nt1, table);
    AGGF_LAUNCH_OK();
    hipLaunchKernelGGL((gram_pair_dma_kernel<T>), dim3((unsigned)round_up(nwg, 256)), dim3(PAIR_THREADS), lds,
                       stream, X, rows, ld, p.nt1, p.n_entries, ksplit, table, fps, slabs);
    AGGF_LAUNCH_OK();
    hipLaunchKernelGGL((gram_reduce_kernel<T>), dim3(p.n_tiles, TILE / 8), dim3(256), 0, stream,
                       slabs, p.nt1, ksplit, n_red, accumulate, G);
    AGGF_LAUNCH_OK();
    return AGGF_OK;
  }
  if constexpr (sizeof(T) == 4) {
    if (p.staging == STAGE_QUAD) {
      int64_t fq = round_up(ceil_div(rows, ksplit), Q_KB);
      if (fq < Q_KB) fq = Q_KB;
      hipLaunchKernelGGL(build_tile_table_kernel, dim3(1), dim3(256), 0, stream, p.nt1, tile_table, 0);
      AGGF_LAUNCH_OK();
      hipLaunchKernelGGL(gram_tile_f32q_kernel, dim3((unsigned)round_up(nblocks, 512)), dim3(Q_THREADS),
                         (size_t)Q_NBUF * Q_BUF * sizeof(float), stream, X, rows, ld, p.nt1, p.n_tiles, ksplit, tile_table, fq,
                         slabs, (const float*)nullptr, (int64_t)0, p.nt1);
      AGGF_LAUNCH_OK();
      hipLaunchKernelGGL((gram_reduce_kernel<T>), dim3(p.n_tiles, TILE / 8), dim3(256), 0, stream, slabs, p.nt1, ksplit, n_red,
                         accumulate, G, 0);
      AGGF_LAUNCH_OK();
      return AGGF_OK;
    }
  }
  if (p.staging == STAGE_DMA8) {
    const size_t lds3 = (size_t)3 * 2 * dma_panel_elems<T>() * sizeof(T);
    static thread_local PerDeviceOnce attr_once;
    bool& attr_done = *attr_once.flag();
    if (!attr_done) {
      AGGF_HIP_OK(hipFuncSetAttribute((const void*)gram_tile_dma_kernel<T, 0, 3, 2, 8, true>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
      attr_done = true;
    }
    hipLaunchKernelGGL(build_tile_table_kernel, dim3(1), dim3(256), 0, stream, p.nt1, tile_table, p.first_tile);
    AGGF_LAUNCH_OK();
    const int64_t nblk = (int64_t)ksplit * p.n_entries;  // n_entries = tiles actually computed
    static const char* f32_mfma = getenv("AGGF_GRAM_F32_MFMA");  // "32": v_mfma_f32_32x32x2_f32 (measurement)
    if constexpr (sizeof(T) == 4) {
      if (f32_mfma && f32_mfma[0] == '3') {
        static thread_local PerDeviceOnce once32;
        bool& done32 = *once32.flag();
        if (!done32) {
          AGGF_HIP_OK(hipFuncSetAttribute((const void*)gram_tile_dma_kernel<T, 0, 3, 2, 8, true, true>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
          done32 = true;
        }
        hipLaunchKernelGGL((gram_tile_dma_kernel<T, 0, 3, 2, 8, true, true>), dim3((unsigned)round_up(nblk, 512)),
                           dim3(512), lds3, stream, X, rows, ld, p.nt1, p.n_entries, ksplit, tile_table, fps, slabs);
        AGGF_LAUNCH_OK();
        hipLaunchKernelGGL((gram_reduce_kernel<T>), dim3(p.n_tiles, TILE / 8), dim3(256), 0, stream,
                           slabs, p.nt1, ksplit, n_red, accumulate, G, p.first_tile);
        AGGF_LAUNCH_OK();
        return AGGF_OK;
      }
    }
    // The stage barrier in front of the last MFMA group instead of behind it (its operands are in registers by then,
    // the 8 MFMAs run while the waves meet), and the DMA piece that goes with that group issued BEHIND the barrier.
    // Same box, back to back: float32 (no piece in the barrier group) c5 60.2 -> 58.6 ms, c2 3.25 -> 3.16 ms; float64
    // C3 756.4 -> 747.9 ms, c4 297.0 -> 293.9 ms.  With the piece in front of the barrier float64 is SLOWER than the
    // late barrier (757 -> 768 ms): the waves then meet right after the instruction that stalls longest, and nobody
    // multiplies until the slowest DMA issue is through.  Two groups behind the barrier (template ES = 2): c5 59.2
    // against 57.3 ms, C3 789 against 746 ms: much worse.  AGGF_GRAM_EARLY_SYNC=0: the late barrier (measurement).
    static const char* es_env = getenv("AGGF_GRAM_EARLY_SYNC");
    if (es_env && es_env[0] == '0') {
      hipLaunchKernelGGL((gram_tile_dma_kernel<T, 0, 3, 2, 8, true>), dim3((unsigned)round_up(nblk, 512)), dim3(512),
                         lds3, stream, X, rows, ld, p.nt1, p.n_entries, ksplit, tile_table, fps, slabs);
    } else {
      static thread_local PerDeviceOnce once_es;
      bool& done_es = *once_es.flag();
      if (!done_es) {
        AGGF_HIP_OK(hipFuncSetAttribute((const void*)gram_tile_dma_kernel<T, 0, 3, 2, 8, true, false, 1, true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
        done_es = true;
      }
      hipLaunchKernelGGL((gram_tile_dma_kernel<T, 0, 3, 2, 8, true, false, 1, true>), dim3((unsigned)round_up(nblk, 512)),
                         dim3(512), lds3, stream, X, rows, ld, p.nt1, p.n_entries, ksplit, tile_table, fps, slabs);
    }
    AGGF_LAUNCH_OK();
    hipLaunchKernelGGL((gram_reduce_kernel<T>), dim3(p.n_tiles, TILE / 8), dim3(256), 0, stream,
                       slabs, p.nt1, ksplit, n_red, accumulate, G, p.first_tile);
    AGGF_LAUNCH_OK();
    return AGGF_OK;
  }
  const bool use_dma = p.staging == STAGE_DMA;
  if (use_dma) {
    const size_t lds3 = (size_t)3 * 2 * dma_panel_elems<T>() * sizeof(T);  // 3-stage ring, 76.8 / 75.3 KB
    static thread_local PerDeviceOnce attr_once;
    bool& attr_done = *attr_once.flag();
    if (!attr_done) {
      AGGF_HIP_OK(hipFuncSetAttribute((const void*)gram_tile_dma_kernel<T>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
      attr_done = true;
    }
    hipLaunchKernelGGL(build_tile_table_kernel, dim3(1), dim3(256), 0, stream, p.nt1, tile_table);
    AGGF_LAUNCH_OK();
    hipLaunchKernelGGL((gram_tile_dma_kernel<T>), dim3((unsigned)round_up(nblocks, 512)), dim3(GRAM_THREADS),
                       lds3, stream, X, rows, ld, p.nt1, p.n_tiles, ksplit, tile_table, fps, slabs);
    AGGF_LAUNCH_OK();
  } else {
    const size_t lds = (size_t)2 * 2 * KB * ROW_STRIDE * sizeof(T);
    hipLaunchKernelGGL((gram_tile_kernel<T>), dim3((unsigned)nblocks), dim3(GRAM_THREADS), lds,
                       stream, X, rows, ld, p.nt1, p.n_tiles, fps, slabs);
    AGGF_LAUNCH_OK();
  }
  hipLaunchKernelGGL((gram_reduce_kernel<T>), dim3(p.n_tiles, TILE / 8), dim3(256), 0, stream,
                     slabs, p.nt1, ksplit, n_red, accumulate, G);
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

// side stream + events of the overlapped pack pipeline (gram_typed), created once per host thread and device
struct PackPipe {
  hipStream_t side = nullptr;
  hipEvent_t packed[2] = {nullptr, nullptr}, consumed[2] = {nullptr, nullptr};
  bool ok = false, tried = false;
};
static PackPipe* pack_pipe() {
  static thread_local PackPipe pipes[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  PackPipe& pp = pipes[dev];
  if (!pp.tried) {
    pp.tried = true;
    bool ok = hipStreamCreateWithFlags(&pp.side, hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; k < 2 && ok; ++k)
      ok = hipEventCreateWithFlags(&pp.packed[k], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&pp.consumed[k], hipEventDisableTiming) == hipSuccess;
    pp.ok = ok;
  }
  return pp.ok ? &pp : nullptr;
}

template <typename TIn, typename TC>
static int gram_typed(const void* Fv, int64_t T, int32_t N, const int32_t* grp_ptr,
                      const int32_t* grp_atoms, int32_t n_red, double* G, int accumulate,
                      const GramPlan& p, char* ws, hipStream_t stream) {
  // workspace: [tile table | slabs | pack chunk]
  int32_t* tile_table = reinterpret_cast<int32_t*>(ws);
  ws += table_bytes(p);
  TC* slabs = reinterpret_cast<TC*>(ws);
  if (p.staging == STAGE_SMALL && p.small_dma) {
    const size_t raw_bytes = (size_t)SD_KB * 3 * N * sizeof(TIn);
    const size_t lds = (size_t)SD_KB * ROW_STRIDE * sizeof(TC) + SD_NBUF * raw_bytes + 16 +
                       (size_t)round_up(((int64_t)N + TILE + 1) * 4, 16) + (size_t)ROW_ELEMS * 4 * sizeof(unsigned short);
    static thread_local PerDeviceOnce attr_once;
    bool& attr_done = *attr_once.flag();
    if (!attr_done) {
      AGGF_HIP_OK(hipFuncSetAttribute((const void*)gram_small_dma_kernel<TIn, TC>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
      attr_done = true;
    }
    static const char* abl_env = getenv("AGGF_SMALL_ABL");
    const int abl = abl_env ? atoi(abl_env) : 0;
    const int nbk = (n_red + 15) / 16;
    const int nbw = (nbk * (nbk + 1) / 2 + SD_NW - 1) / SD_NW;  // blocks per wave: 1 .. 5
#define AGGF_SD(A, W)                                                                                               \
  do {                                                                                                               \
    AGGF_HIP_OK(hipFuncSetAttribute((const void*)gram_small_dma_kernel<TIn, TC, A, W>,                               \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));                 \
    hipLaunchKernelGGL((gram_small_dma_kernel<TIn, TC, A, W>), dim3((unsigned)p.ksplit), dim3(SD_THREADS), lds,      \
                       stream, reinterpret_cast<const TIn*>(Fv), T, N, grp_ptr, grp_atoms, n_red,                    \
                       p.frames_per_split, (int32_t)raw_bytes, slabs);                                               \
  } while (0)
    if (abl == 1) AGGF_SD(1, 5);
    else if (abl == 2) AGGF_SD(2, 5);
    else if (abl == 3) AGGF_SD(3, 5);
    else if (nbw <= 1) AGGF_SD(0, 1);
    else if (nbw == 2) AGGF_SD(0, 2);
    else if (nbw == 3) AGGF_SD(0, 3);
    else if (nbw == 4) AGGF_SD(0, 4);
    else AGGF_SD(0, 5);
#undef AGGF_SD
    AGGF_LAUNCH_OK();
    hipLaunchKernelGGL((gram_reduce_small_kernel<TC>), dim3(TILE), dim3(256), 0, stream, slabs, p.ksplit, n_red,
                       accumulate, G);
    AGGF_LAUNCH_OK();
    return AGGF_OK;
  }
  if (p.staging == STAGE_SMALL) {
    static const char* shape = getenv("AGGF_GRAM_SMALL");
    const bool big = !(shape && shape[0] == '4');
    const int kbs = big ? 8 : 4, threads = big ? 512 : 256;
    const size_t raw_bytes = small_raw_bytes<TIn>(N, kbs);
    const size_t lds = (size_t)kbs * ROW_STRIDE * sizeof(TC) + raw_bytes + (size_t)round_up(((int64_t)N + TILE + 1) * 4, 16) +
                       (size_t)ROW_ELEMS * 4 * sizeof(unsigned short);
    const int nv = (int)ceil_div((int64_t)(raw_bytes / 16 - 1), threads);
    if (nv > SM_MAXVEC) return fail(AGGF_ERR_ARG, "aggf_gram: small-system kernel: frame too large");
#define AGGF_SMALL(NVC, KBC, NWC)                                                                                    \
  do {                                                                                                               \
    if (lds > 65536) {                                                                                               \
      static thread_local PerDeviceOnce attr_once;                                                                   \
      bool& attr_done = *attr_once.flag();                                                                           \
      if (!attr_done) {                                                                                              \
        AGGF_HIP_OK(hipFuncSetAttribute((const void*)gram_small_kernel<TIn, TC, NVC, KBC, NWC>,                      \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));             \
        attr_done = true;                                                                                            \
      }                                                                                                              \
    }                                                                                                                \
    hipLaunchKernelGGL((gram_small_kernel<TIn, TC, NVC, KBC, NWC>), dim3((unsigned)p.ksplit), dim3(64 * NWC), lds,   \
                       stream, reinterpret_cast<const TIn*>(Fv), T, N, grp_ptr, grp_atoms, n_red,                    \
                       p.frames_per_split, (int32_t)raw_bytes, slabs);                                               \
  } while (0)
    if (big) {
      if (nv <= 3) AGGF_SMALL(3, 8, 8);
      else if (nv <= 5) AGGF_SMALL(5, 8, 8);
      else AGGF_SMALL(8, 8, 8);
    } else {
      if (nv <= 3) AGGF_SMALL(3, 4, 4);
      else if (nv <= 5) AGGF_SMALL(5, 4, 4);
      else AGGF_SMALL(8, 4, 4);
    }
#undef AGGF_SMALL
    AGGF_LAUNCH_OK();
    hipLaunchKernelGGL((gram_reduce_small_kernel<TC>), dim3(TILE), dim3(256), 0, stream, slabs, p.ksplit, n_red,
                       accumulate, G);
    AGGF_LAUNCH_OK();
    return AGGF_OK;
  }
  if (p.direct) {
    // only reachable with TIn == TC
    return launch_gram<TC>(reinterpret_cast<const TC*>(Fv), T, (int64_t)N * 3, p, slabs, tile_table, G,
                           n_red, accumulate, stream);
  }
  TC* pack = reinterpret_cast<TC*>(ws + round_up((int64_t)p.slab_bytes, 256));
  const TIn* F = reinterpret_cast<const TIn*>(Fv);
  // The pack pass is HBM-bound, the tile kernel MFMA-bound: the chunk is split into two half-size buffers and chunk
  // i + 1 is packed on a side stream while chunk i is multiplied; a trajectory that fits one chunk is still cut into
  // PACK_MIN_CHUNKS pieces for the same reason.  The pack beside the tile kernel is a SMALL grid of long-lived
  // workgroups with non-temporal accesses (one per CU; AGGF_GRAM_PACK_WGS): a full-size grid's workgroups take turns
  // with the tile kernel's for the CUs and break up the cohorts that share panels in the L2.  C3 with bond pairs,
  // Gram stage (tools/pack_ab.sh): serial 404-407 ms; overlapped with the full grid 562-581; with 32 / 64 / 128 / 192 /
  // 256 / 384 / 512 workgroups 1017 / 633 / 430 / 396 / 397 / 399 / 409 -- 35 ms of pack, 8 of them hidden.
  // AGGF_GRAM_PACK=serial: one buffer, one stream; =chunked: the overlapped form's chunks on one stream (measurements).
  static const char* pack_env = getenv("AGGF_GRAM_PACK");
  constexpr int PACK_MIN_CHUNKS = 8;
  // a chunk must be worth its launches and event waits: >= 8192 frames and >= 4e11 flop (~6 ms of tile kernel)
  int64_t PACK_MIN_FRAMES = (int64_t)(4e11 / (3.0 * (double)p.n_pad * (double)p.n_pad)) + 1;
  if (PACK_MIN_FRAMES < 8192) PACK_MIN_FRAMES = 8192;
  const size_t row_elems = (size_t)p.n_pad * 3;
  int64_t cf = p.chunk_frames;
  bool overlap = !(pack_env && pack_env[0] == 's') && T >= 2 * PACK_MIN_FRAMES && p.chunk_frames / 2 >= PACK_MIN_FRAMES;
  const bool same_stream = pack_env && pack_env[0] == 'c';  // (measurement: the overlapped form's chunks, one stream)
  PackPipe* pipe = overlap ? pack_pipe() : nullptr;
  if (overlap && !pipe) overlap = false;  // no side stream: the serial form
  if (overlap) {
    cf = p.chunk_frames / 2;                              // two buffers in the space of one
    const int64_t want = ceil_div(T, (int64_t)PACK_MIN_CHUNKS);
    if (cf > want) cf = want;
    if (cf < PACK_MIN_FRAMES) cf = PACK_MIN_FRAMES;
    if (cf > p.chunk_frames / 2) cf = p.chunk_frames / 2;
    cf = (cf / 8) * 8;                                    // whole stages (both dtypes)
    if (cf < 8) overlap = false;
  }
  if (!overlap) cf = p.chunk_frames;
  TC* bufs[2] = {pack, overlap ? pack + (size_t)cf * row_elems : pack};
  auto launch_pack = [&](int64_t t0, int64_t rows, TC* dst, hipStream_t st) {
    const unsigned gx = (unsigned)ceil_div((int64_t)p.n_pad * 3, 256 * PK_ELEMS);
    int64_t gy = ceil_div((int64_t)16 * device_cu_count(), gx);  // ~16 workgroups per CU in all
    if (st != stream) {
      // beside the tile kernel: a FEW long-lived workgroups that trickle the chunk through (a full-size grid's
      // workgroups take turns with the tile kernel's for the CUs and break up the cohorts that share panels in the L2)
      static const char* wg_env = getenv("AGGF_GRAM_PACK_WGS");
      const int64_t budget = wg_env && atoi(wg_env) > 0 ? atoi(wg_env) : device_cu_count();  // one per CU
      gy = ceil_div(budget, (int64_t)gx);
    }
    if (gy > rows) gy = rows;
    if (st != stream)
      hipLaunchKernelGGL((pack_groups_kernel<TIn, TC, true>), dim3(gx, (unsigned)gy), dim3(256), 0, st,
                         F + t0 * (int64_t)N * 3, rows, N, grp_ptr, grp_atoms, n_red, p.n_pad, dst);
    else
      hipLaunchKernelGGL((pack_groups_kernel<TIn, TC>), dim3(gx, (unsigned)gy), dim3(256), 0, st,
                         F + t0 * (int64_t)N * 3, rows, N, grp_ptr, grp_atoms, n_red, p.n_pad, dst);
  };
  int acc = accumulate;
  if (overlap) {
    // side stream: starts behind everything already queued on `stream` (the caller's data), ends before the last
    // tile kernel starts (that kernel waits for its pack), so the caller's stream order covers the whole call
    AGGF_HIP_OK(hipEventRecord(pipe->consumed[0], stream));
    AGGF_HIP_OK(hipStreamWaitEvent(pipe->side, pipe->consumed[0], 0));
  }
  int64_t i = 0;
  for (int64_t t0 = 0; t0 < T; t0 += cf, ++i) {
    const int64_t rows = (T - t0 < cf) ? T - t0 : cf;
    const int b = (int)(i & 1);
    if (overlap && same_stream) {
      launch_pack(t0, rows, bufs[b], stream);
      AGGF_LAUNCH_OK();
    } else if (overlap) {
      if (i >= 2) AGGF_HIP_OK(hipStreamWaitEvent(pipe->side, pipe->consumed[b], 0));  // the kernel that read this buffer
      launch_pack(t0, rows, bufs[b], pipe->side);
      AGGF_LAUNCH_OK();
      AGGF_HIP_OK(hipEventRecord(pipe->packed[b], pipe->side));
      AGGF_HIP_OK(hipStreamWaitEvent(stream, pipe->packed[b], 0));
    } else {
      launch_pack(t0, rows, bufs[b], stream);
      AGGF_LAUNCH_OK();
    }
    int rc = launch_gram<TC>(bufs[b], rows, (int64_t)p.n_pad * 3, p, slabs, tile_table, G, n_red, acc, stream);
    if (rc) return rc;
    if (overlap && !same_stream) AGGF_HIP_OK(hipEventRecord(pipe->consumed[b], stream));
    acc = 1;
  }
  return AGGF_OK;
}

}  // namespace aggf

using namespace aggf;

extern "C" size_t aggf_gram_workspace_bytes(int64_t T, int32_t N, int32_t n_red, int in_dtype,
                                            int compute_dtype, int has_groups) {
  if (T <= 0 || N <= 0 || n_red <= 0) return 0;
  GramPlan p;
  make_plan(T, N, n_red, in_dtype, compute_dtype, has_groups != 0, true, 0, true, &p);
  return table_bytes(p) + (size_t)round_up((int64_t)p.slab_bytes, 256) + p.pack_bytes + 1024;
}

static int gram_impl(const void* F, int64_t T, int32_t N, int in_dtype, int compute_dtype,
                     const int32_t* grp_ptr, const int32_t* grp_atoms, int32_t n_red, int32_t first_col,
                     double* G, int accumulate, void* ws, size_t ws_bytes, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!F || !G || !ws) return fail(AGGF_ERR_ARG, "aggf_gram: NULL pointer");
  if (T <= 0 || N <= 0 || n_red <= 0) return fail(AGGF_ERR_ARG, "aggf_gram: empty problem");
  if ((in_dtype != AGGF_F32 && in_dtype != AGGF_F64) ||
      (compute_dtype != AGGF_F32 && compute_dtype != AGGF_F64))
    return fail(AGGF_ERR_ARG, "aggf_gram: bad dtype");
  if (in_dtype == AGGF_F64 && compute_dtype == AGGF_F32)
    return fail(AGGF_ERR_ARG, "aggf_gram: float64 input with float32 products is not supported");
  const bool has_groups = grp_ptr != nullptr;
  if (has_groups != (grp_atoms != nullptr))
    return fail(AGGF_ERR_ARG, "aggf_gram: grp_ptr and grp_atoms must be given together");
  if (n_red > N) return fail(AGGF_ERR_ARG, "aggf_gram: n_red > N");
  if (first_col < 0 || first_col % TILE != 0) return fail(AGGF_ERR_ARG, "aggf_gram_from_column: first_col must be a multiple of 128");
  // the leading block is skipped only by the in-place LDS-DMA plan; every other plan computes (and with `accumulate`
  // would ADD) it, so the result of accumulate + first_col would depend on pointer alignment: refused
  if (first_col > 0 && accumulate) return fail(AGGF_ERR_ARG, "aggf_gram_from_column: accumulate != 0 needs first_col == 0");
  if (((uintptr_t)ws & 255) != 0) return fail(AGGF_ERR_ARG, "aggf_gram: workspace not 256-byte aligned");
  const bool aligned = ((uintptr_t)F & 15) == 0;
  GramPlan p;
  int rc = make_plan(T, N, n_red, in_dtype, compute_dtype, has_groups, aligned, ws_bytes, false, &p, first_col);
  if (rc) return rc;
  if (table_bytes(p) + (size_t)round_up((int64_t)p.slab_bytes, 256) + p.pack_bytes > ws_bytes)
    return fail(AGGF_ERR_WORKSPACE, "aggf_gram: workspace too small");
  char* w = reinterpret_cast<char*>(ws);
  if (in_dtype == AGGF_F64)
    return gram_typed<double, double>(F, T, N, grp_ptr, grp_atoms, n_red, G, accumulate, p, w, stream);
  if (compute_dtype == AGGF_F64)
    return gram_typed<float, double>(F, T, N, grp_ptr, grp_atoms, n_red, G, accumulate, p, w, stream);
  return gram_typed<float, float>(F, T, N, grp_ptr, grp_atoms, n_red, G, accumulate, p, w, stream);
}

extern "C" int aggf_gram(const void* F, int64_t T, int32_t N, int in_dtype, int compute_dtype,
                         const int32_t* grp_ptr, const int32_t* grp_atoms, int32_t n_red,
                         double* G, int accumulate, void* ws, size_t ws_bytes, void* stream_v) {
  return gram_impl(F, T, N, in_dtype, compute_dtype, grp_ptr, grp_atoms, n_red, 0, G, accumulate, ws, ws_bytes, stream_v);
}

// Gram matrix of the column-concatenation [F | F2] without materialising it: both (T, ., 3) arrays of the same dtype
// (which is also the arithmetic type of the products), N and N2 multiples of 128, 16-byte aligned.
template <typename T>
static int gram_pair_typed(const T* F, const T* F2, int64_t rows, int32_t N, int32_t N2, const GramPlan& p, char* ws,
                           double* G, int accumulate, hipStream_t stream) {
  constexpr int KB = GramCfg<T>::KB;
  int32_t* tile_table = reinterpret_cast<int32_t*>(ws);
  T* slabs = reinterpret_cast<T*>(ws + table_bytes(p));
  const int ksplit = p.ksplit;
  int64_t fps = round_up(ceil_div(rows, ksplit), KB);
  if (fps < KB) fps = KB;
  const size_t lds3 = (size_t)3 * 2 * dma_panel_elems<T>() * sizeof(T);
  static thread_local PerDeviceOnce attr_once;
  bool& attr_done = *attr_once.flag();
  if (!attr_done) {
    AGGF_HIP_OK(hipFuncSetAttribute((const void*)gram_tile_dma_kernel<T, 0, 3, 2, 8, true, false, 1, true, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
    attr_done = true;
  }
  hipLaunchKernelGGL(build_tile_table_kernel, dim3(1), dim3(256), 0, stream, p.nt1, tile_table, 0);
  AGGF_LAUNCH_OK();
  const int64_t nblk = (int64_t)ksplit * p.n_tiles;
  if (nblk > 0x7fffff00LL) return fail(AGGF_ERR_ARG, "gram grid too large");
  hipLaunchKernelGGL((gram_tile_dma_kernel<T, 0, 3, 2, 8, true, false, 1, true, true>), dim3((unsigned)round_up(nblk, 512)),
                     dim3(512), lds3, stream, F, rows, (int64_t)N * 3, p.nt1, p.n_tiles, ksplit, tile_table, fps, slabs, F2,
                     (int64_t)N2 * 3, N / TILE);
  AGGF_LAUNCH_OK();
  hipLaunchKernelGGL((gram_reduce_kernel<T>), dim3(p.n_tiles, TILE / 8), dim3(256), 0, stream, slabs, p.nt1, ksplit,
                     N + N2, accumulate, G, 0);
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" size_t aggf_gram_pair_workspace_bytes(int64_t T, int32_t N, int32_t N2, int dtype) {
  if (T <= 0 || N <= 0 || N2 <= 0) return 0;
  return aggf_gram_workspace_bytes(T, N + N2, N + N2, dtype, dtype, 0);
}

extern "C" int aggf_gram_pair(const void* F, int32_t N, const void* F2, int32_t N2, int64_t T, int dtype, double* G,
                              int accumulate, void* ws, size_t ws_bytes, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!F || !F2 || !G || !ws) return fail(AGGF_ERR_ARG, "aggf_gram_pair: NULL pointer");
  if (T <= 0 || N <= 0 || N2 <= 0) return fail(AGGF_ERR_ARG, "aggf_gram_pair: empty problem");
  if (dtype != AGGF_F32 && dtype != AGGF_F64) return fail(AGGF_ERR_ARG, "aggf_gram_pair: bad dtype");
  if (N % TILE != 0 || N2 % TILE != 0)
    return fail(AGGF_ERR_ARG, "aggf_gram_pair: both site counts must be multiples of 128 (concatenate and use aggf_gram otherwise)");
  if ((((uintptr_t)F | (uintptr_t)F2) & 15) != 0) return fail(AGGF_ERR_ARG, "aggf_gram_pair: arrays must be 16-byte aligned");
  if (((uintptr_t)ws & 255) != 0) return fail(AGGF_ERR_ARG, "aggf_gram_pair: workspace not 256-byte aligned");
  GramPlan p;
  int rc = make_plan(T, N + N2, N + N2, dtype, dtype, false, true, ws_bytes, false, &p, 0);
  if (rc) return rc;
  if (!p.direct || p.staging == STAGE_SMALL) return fail(AGGF_ERR_ARG, "aggf_gram_pair: unsupported layout");
  p.staging = STAGE_DMA8;
  p.n_entries = p.n_tiles;
  if (table_bytes(p) + (size_t)round_up((int64_t)p.slab_bytes, 256) > ws_bytes)
    return fail(AGGF_ERR_WORKSPACE, "aggf_gram_pair: workspace too small");
  char* w = reinterpret_cast<char*>(ws);
  if (dtype == AGGF_F64)
    return gram_pair_typed<double>((const double*)F, (const double*)F2, T, N, N2, p, w, G, accumulate, stream);
  return gram_pair_typed<float>((const float*)F, (const float*)F2, T, N, N2, p, w, G, accumulate, stream);
}

// ---- fused constraint sums / conversion / padding: the raw trajectory through the gather tile kernel -----------
struct GatherGeom {
  int span_bytes, row_slot, nbuf;
};
// LDS geometry for a window of span_atoms atoms of `in_size`-byte elements; nbuf == 0: does not fit two stages
static GatherGeom gather_geom(int32_t span_atoms, int in_size, int kb) {
  GatherGeom g;
  g.span_bytes = span_atoms * 3 * in_size;
  g.row_slot = GA_HDR + (int)round_up(g.span_bytes + 16, 16);
  if ((g.row_slot / 32) % 2 == 0) g.row_slot += 32;  // odd multiple of 32 bytes: the 4 rows of an operand read start 8 banks apart
  const int stage = 2 * kb * g.row_slot;
  g.nbuf = 3 * stage <= 80 * 1024 ? 3 : 2 * stage <= 80 * 1024 ? 2 : 0;
  const int ppr = (g.span_bytes + 15 + 1023) / 1024;
  if (2 * kb * ppr / 8 > GA_MAXPPW) g.nbuf = 0;  // more DMA pieces per wave and stage than the kernel keeps track of
  return g;
}

template <typename TIn, typename TC, int MAXM>
static int gram_gather_launch(const TIn* F, int64_t T, int32_t N, const int32_t* col_off, const int32_t* panel_lo,
                              const GatherGeom& g, const GramPlan& p, char* ws, double* G, int32_t n_red, int accumulate,
                              hipStream_t stream) {
  constexpr int KB = GramCfg<TC>::KB;
  int32_t* tile_table = reinterpret_cast<int32_t*>(ws);
  TC* slabs = reinterpret_cast<TC*>(ws + table_bytes(p));
  const int ksplit = p.ksplit;
  int64_t fps = round_up(ceil_div(T, ksplit), KB);
  if (fps < KB) fps = KB;
  const size_t lds = (size_t)g.nbuf * 2 * KB * g.row_slot;
  static thread_local PerDeviceOnce attr_once;
  bool& attr_done = *attr_once.flag();
  if (!attr_done) {
    AGGF_HIP_OK(hipFuncSetAttribute((const void*)gram_tile_gather_kernel<TIn, TC, MAXM>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL(build_tile_table_kernel, dim3(1), dim3(256), 0, stream, p.nt1, tile_table, 0);
  AGGF_LAUNCH_OK();
  const int64_t nblk = (int64_t)ksplit * p.n_tiles;
  if (nblk > 0x7fffff00LL) return fail(AGGF_ERR_ARG, "gram grid too large");
  hipLaunchKernelGGL((gram_tile_gather_kernel<TIn, TC, MAXM>), dim3((unsigned)round_up(nblk, 512)), dim3(512), lds, stream,
                     F, T, N, col_off, panel_lo, g.span_bytes, g.row_slot, g.nbuf, p.nt1, p.n_tiles, ksplit, tile_table,
                     fps, slabs);
  AGGF_LAUNCH_OK();
  hipLaunchKernelGGL((gram_reduce_kernel<TC>), dim3(p.n_tiles, TILE / 8), dim3(256), 0, stream, slabs, p.nt1, ksplit,
                     n_red, accumulate, G, 0);
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_gram_gather_supported(int64_t T, int32_t N, int32_t n_red, int in_dtype, int compute_dtype,
                                          int32_t max_members, int32_t span_atoms) {
  if (T <= 0 || N <= 0 || n_red <= TILE || n_red > N || max_members < 1 || max_members > 4 || span_atoms < 1) return 0;
  if ((in_dtype != AGGF_F32 && in_dtype != AGGF_F64) || (compute_dtype != AGGF_F32 && compute_dtype != AGGF_F64)) return 0;
  if (in_dtype == AGGF_F64 && compute_dtype == AGGF_F32) return 0;
  const int in_size = (int)dtype_size(in_dtype);
  if ((T * (int64_t)N * 3 * in_size) % 16 != 0) return 0;
  return gather_geom(span_atoms, in_size, compute_dtype == AGGF_F64 ? 4 : 8).nbuf > 0;
}

extern "C" size_t aggf_gram_gather_workspace_bytes(int64_t T, int32_t n_red, int compute_dtype) {
  if (T <= 0 || n_red <= 0) return 0;
  const int32_t n_pad = (int32_t)round_up(n_red, TILE);
  return aggf_gram_workspace_bytes(T, n_pad, n_pad, compute_dtype, compute_dtype, 0);  // tile table + slabs, no pack chunk
}

extern "C" int aggf_gram_gather(const void* F, int64_t T, int32_t N, int in_dtype, int compute_dtype,
                                const int32_t* col_off, int32_t max_members, const int32_t* panel_lo, int32_t span_atoms,
                                int32_t n_red, double* G, int accumulate, void* ws, size_t ws_bytes, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!F || !col_off || !panel_lo || !G || !ws) return fail(AGGF_ERR_ARG, "aggf_gram_gather: NULL pointer");
  if (!aggf_gram_gather_supported(T, N, n_red, in_dtype, compute_dtype, max_members, span_atoms))
    return fail(AGGF_ERR_ARG, "aggf_gram_gather: layout not supported (see aggf_gram_gather_supported); use aggf_gram");
  if (((uintptr_t)F & 15) != 0) return fail(AGGF_ERR_ARG, "aggf_gram_gather: F must be 16-byte aligned");
  if (((uintptr_t)ws & 255) != 0) return fail(AGGF_ERR_ARG, "aggf_gram_gather: workspace not 256-byte aligned");
  const int32_t n_pad = (int32_t)round_up(n_red, TILE);
  GramPlan p;
  int rc = make_plan(T, n_pad, n_pad, compute_dtype, compute_dtype, false, true, ws_bytes, false, &p, 0);
  if (rc) return rc;
  if (!p.direct || p.staging == STAGE_SMALL) return fail(AGGF_ERR_ARG, "aggf_gram_gather: unsupported layout");
  p.staging = STAGE_DMA8;
  p.n_entries = p.n_tiles;
  if (table_bytes(p) + (size_t)round_up((int64_t)p.slab_bytes, 256) > ws_bytes)
    return fail(AGGF_ERR_WORKSPACE, "aggf_gram_gather: workspace too small");
  const GatherGeom g = gather_geom(span_atoms, (int)dtype_size(in_dtype), compute_dtype == AGGF_F64 ? 4 : 8);
  char* w = reinterpret_cast<char*>(ws);
  const int mm = max_members <= 1 ? 1 : max_members <= 2 ? 2 : 4;
#define AGGF_GG(TIN, TCC)                                                                                             \
  do {                                                                                                                \
    if (mm == 1) return gram_gather_launch<TIN, TCC, 1>((const TIN*)F, T, N, col_off, panel_lo, g, p, w, G, n_red, accumulate, stream); \
    if (mm == 2) return gram_gather_launch<TIN, TCC, 2>((const TIN*)F, T, N, col_off, panel_lo, g, p, w, G, n_red, accumulate, stream); \
    return gram_gather_launch<TIN, TCC, 4>((const TIN*)F, T, N, col_off, panel_lo, g, p, w, G, n_red, accumulate, stream);  \
  } while (0)
  if (in_dtype == AGGF_F64) AGGF_GG(double, double);
  if (compute_dtype == AGGF_F64) AGGF_GG(float, double);
  AGGF_GG(float, float);
#undef AGGF_GG
}

extern "C" int aggf_gram_from_column(const void* F, int64_t T, int32_t N, int in_dtype, int compute_dtype,
                                     int32_t n_red, int32_t first_col, double* G, int accumulate, void* ws,
                                     size_t ws_bytes, void* stream_v) {
  return gram_impl(F, T, N, in_dtype, compute_dtype, nullptr, nullptr, n_red, first_col, G, accumulate, ws, ws_bytes,
                   stream_v);
}
