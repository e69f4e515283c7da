// K1: Gram / normal matrix of (constraint-reduced) forces on MFMA.
//
// Replaces qp/qplinear.py:66-71 of the reference (qp_form copy, `@ con_mat`,
// `reg_mat.T @ reg_mat`).  Three kernels:
//   pack_groups_kernel   (T,N,3) -> (T,n_pad,3): constraint-group column sums, dtype
//                        conversion and zero padding to a multiple of 128 columns
//                        (skipped when the tile kernel can read the input where it lies: no
//                        groups, rows of whole 16-byte pieces, the product dtype -- or float32
//                        frames with float64 products, widened out of LDS);
//   gram_tile_dma_kernel split-K SYRK: one 128x128 upper-triangle tile x one frame
//                        range per workgroup, MFMA 16x16x4 (f64 or f32), frame rows
//                        staged through LDS exactly as they lie in HBM -- the
//                        (t,d)-major / atom-minor transpose the reference pays a
//                        full copy for (qp_form) is done by the LDS read pattern;
//   gram_reduce_kernel   fixed-order fp64 sum of the split-K slabs into G (both
//                        triangles) -- no float atomics, bit-reproducible.
#include <stdlib.h>

#include "aggf_common.h"
#include "aggf_routing.h"

#include <type_traits>

namespace aggf {

constexpr int TILE = 128;         // output tile edge (reduced atoms)
constexpr int ROW_ELEMS = TILE * 3;
constexpr int ROW_PAD = 16;       // row stride == 128 B (f64) / 64 B (f32) mod bank period:
                                  // the two k-rows a 32-lane half reads hit disjoint banks
constexpr int ROW_STRIDE = ROW_ELEMS + ROW_PAD;

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
// The tile kernel.  X: (rows, ld) with ld = 3*n_pad elements, n_pad % 128 == 0, 16-byte aligned rows; one workgroup =
// one 128x128 upper-triangle tile x one frame range (split-K).  The panels travel HBM/L2 -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip, no ds_write -- a register-staged refill cost 6-12 % of the MFMA time,
// profiles/r04_pruned_variants.patch) through a 3-stage LDS ring with a counted vmcnt, so the DMAs of stage s+2 stay
// in flight across the barrier of stage s.
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
  static constexpr int UNITS = 4;            // LDS units per panel and stage: the KB = 4 rows
  static constexpr int UNIT_STRIDE = ROW_STRIDE;
  static constexpr int PIECE_ELEMS = 128;    // elements per piece
  static constexpr int KK_STRIDE = 4 * ROW_STRIDE;  // LDS distance between the row groups kk and kk+1
  __host__ __device__ static constexpr int row_off(int r) { return r * ROW_STRIDE; }
};
template <>
struct DmaCfg<float> {
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

// AGGF_SMALL_ABL (tools/small_ablate.hip only; the library is built without it): the single-tile kernel with one of its
// three co-limiting phases removed -- 1 = no MFMA phase (operand reads and MFMAs), 2 = no global loads after the first
// stage (the fetched registers are re-parked), 3 = no group sums (the panel keeps its first contents).
#ifndef AGGF_SMALL_ABL
#define AGGF_SMALL_ABL 0
#endif
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
// EDGE: X is read where it lies although its rows are NOT padded to whole panels (ld = row_elems = 3 N with
// N % 128 != 0; rows of whole 16-byte pieces, or `straddle` -- the LDS-DMA takes any byte address,
// tools/dma_align_probe.hip): lanes whose chunk lies past the end of a row issue no DMA -- their
// LDS slots are zeroed once and stay zero -- and a piece without an active lane does not count in vmcnt, so the
// counted waits take their numbers from wave-uniform tallies instead of the template's constants.
template <typename T, int ABL = 0, int NBUF = 3, int WPS = 2, int NW = 4, bool SPREAD_DMA = false, int ES = 0,
          bool ES_DMA_AFTER = false, bool TWO = false, bool EDGE = false, typename TS = T>
__global__ __launch_bounds__(64 * NW, WPS) void gram_tile_dma_kernel(
    const TS* __restrict__ X, int64_t n_rows, int64_t ld, int32_t nt1, int32_t n_tiles, int32_t ksplit,
    const int32_t* __restrict__ tile_table, int64_t frames_per_split, T* __restrict__ slabs,
    const TS* __restrict__ X2 = nullptr, int64_t ld2 = 0, int32_t np1 = 0, int32_t row_elems = 0, int32_t straddle = 0) {
  using M = Mfma<T>;
  using acc_t = typename M::acc_t;
  // TS: the type of the frames in HBM and in the LDS ring (its stage layout and row count); T: the type of the products.
  // TS = float with T = double: float32 frames staged as they lie, every MFMA operand widened on its way out of LDS
  // (v_cvt_f64_f32: exact) -- the reference's float64 products of a float32 trajectory without a converted copy.
  constexpr int KB = GramCfg<TS>::KB;
  constexpr bool F32 = sizeof(TS) == 4;
  constexpr int PE = DmaCfg<TS>::PIECE_ELEMS;
  constexpr int UNITS = DmaCfg<TS>::UNITS;
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
  constexpr int PANEL_ELEMS = dma_panel_elems<TS>();
  constexpr int BUF_ELEMS = PANELS * PANEL_ELEMS;
  constexpr int AHEAD = NBUF - 1;  // stages in flight ahead of the one being computed
  constexpr bool EARLY_SYNC = ES > 0 && SPREAD_DMA && ABL == 0;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  TS* smem = reinterpret_cast<TS*>(smem_raw);

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
  bool c_ok[PPW];
  const TS* g_base[TWO ? PPW : 1];
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
    l_off[q] = panel * PANEL_ELEMS + unit * DmaCfg<TS>::UNIT_STRIDE + cp * PE;
    // straddle (rows that are not whole 16-byte pieces): the piece that starts inside the row is read although it ends
    // in the next row -- what it brings beyond the row's end lands in columns past n_red, whose products nobody reads;
    // the caller keeps the array's last row out of this launch (gram_tail_row_kernel adds it)
    c_ok[q] = !EDGE || (panel ? tj : ti) * ROW_ELEMS + elem + (straddle ? 1 : (int)(16 / sizeof(TS))) <= row_elems;
  }
  // EDGE: pieces with an active lane (wave-uniform; the others never count in vmcnt), all and those in front of the
  // early barrier
  int n_act = PPW, n_act_early = 0;
  if constexpr (EDGE) {
    n_act = 0;
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const int a = __any(c_ok[q]) ? 1 : 0;
      n_act += a;
      constexpr int GROUPS_ = 3 * KB / 4;
      if (ES > 0 && SPREAD_DMA && dma_piece_group(GROUPS_, PPW, q) <= GROUPS_ - ES - (ES_DMA_AFTER ? 1 : 0)) n_act_early += a;
    }
    n_act = __builtin_amdgcn_readfirstlane(n_act);
    n_act_early = __builtin_amdgcn_readfirstlane(n_act_early);
    for (int e = tid; e < NBUF * BUF_ELEMS; e += NTHREADS) smem[e] = 0;
    __syncthreads();
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
      TS* lbase = smem + (s % NBUF) * BUF_ELEMS;
      const int first = (int)(t_end - t0);
      for (int e = tid; e < PANELS * (KB - first) * ROW_ELEMS; e += NTHREADS) {
        const int panel = e / ((KB - first) * ROW_ELEMS);
        const int rem = e - panel * (KB - first) * ROW_ELEMS;
        const int r = first + rem / ROW_ELEMS, c = rem % ROW_ELEMS;
        lbase[panel * PANEL_ELEMS + DmaCfg<TS>::row_off(r) + c] = 0;
      }
    }
  };
  auto issue_piece = [&](int s, int q) {
    // ABL 3 (ablation): every stage re-reads the first rows of the split -> all DMAs hit the L2
    // ABL 4: cycle over 8 stages -> DMAs miss the L1 but hit the L2
    const int64_t t0 = t_begin + (ABL == 3 ? 0 : ABL == 4 ? (int64_t)(stage_of(s) & 7) * KB : (int64_t)stage_of(s) * KB);
    const bool row_ok = t0 + p_row[q] < t_end && c_ok[q];
    if (row_ok) {
      const TS* src = TWO ? g_base[TWO ? q : 0] + t0 * g_ld[TWO ? q : 0] + g_off[q] : X + t0 * ld + g_off[q];
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
  // MFMA operand of row group kk: rows kk*4 + (lane >> 4) -- f64: four consecutive rows; f32: member kk of
  // the four pairs
  const int offA = (lane >> 4) * DmaCfg<TS>::UNIT_STRIDE + 3 * (wm * 64 + (lane & 15));
  const int offB = PANEL_ELEMS + (lane >> 4) * DmaCfg<TS>::UNIT_STRIDE + 3 * (wn * WCOLS + (lane & 15));
  constexpr int KKS = DmaCfg<TS>::KK_STRIDE;

  if (n_it > 0) issue_stage(0);
  if (AHEAD > 1 && n_it > 1) issue_stage(1);
  if (AHEAD > 1 && n_it > 1 && ragged_seq != 1) {
    if constexpr (EDGE) wait_vmcnt_dyn<8>(n_act); else wait_vmcnt<PPW>();
  } else {
    wait_vmcnt<0>();
  }
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
    const TS* pa = smem + (((ABL == 1 || ABL == 2) ? it % 2 : it % NBUF)) * BUF_ELEMS;
    AGGF_PROF_T(p1);
    T aL[ES > 0 ? ES : 1][4], bL[ES > 0 ? ES : 1][NACC];  // operands of the groups behind an early barrier
    (void)aL;
    (void)bL;
#pragma unroll
    for (int kk = 0; kk < KB / 4; ++kk) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
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
            for (int m = 0; m < 4; ++m) aL[h][m] = (T)pa[offA + kh * KKS + 48 * m + dh];
#pragma unroll
            for (int n = 0; n < NACC; ++n) bL[h][n] = (T)pa[offB + kh * KKS + 48 * n + dh];
          }
        }
        if (ESG > 0 && g >= GROUPS - ESG) {
#pragma unroll
          for (int m = 0; m < 4; ++m) a[m] = aL[g - (GROUPS - ESG)][m];
#pragma unroll
          for (int n = 0; n < NACC; ++n) bb[n] = bL[g - (GROUPS - ESG)][n];
        } else {
#pragma unroll
          for (int m = 0; m < 4; ++m) a[m] = (T)pa[offA + kk * KKS + 48 * m + d];
#pragma unroll
          for (int n = 0; n < NACC; ++n) bb[n] = (T)pa[offB + kk * KKS + 48 * n + d];
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
          if (AHEAD > 1 && it + 2 < n_it && it + 2 != ragged_seq) {
            if constexpr (EDGE) wait_vmcnt_dyn<8>(n_act_early); else wait_vmcnt<ISSUED>();
          } else {
            wait_vmcnt<0>();
          }
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
    AGGF_PROF_T(p2);
    // stage it+1 must have landed (this wave's pieces), stage it+2 may stay in flight
    if (ABL != 2 && !EARLY_SYNC) {
      // (the ragged stage of the last split issues fewer DMAs: a counted wait would let pieces
      // of stage it+1 slip through)
      if (AHEAD > 1 && it + 2 < n_it && it + 2 != ragged_seq) {
        if constexpr (EDGE) wait_vmcnt_dyn<8>(n_act); else wait_vmcnt<PPW>();
      } else {
        wait_vmcnt<0>();
      }
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
// G[(ti,tj) tile] (+)= sum_ks slab, and the mirrored tile.  grid = (n_tiles, 16): each
// workgroup handles 8 rows of a tile... (16 row-groups of 8 rows x 128 cols, 256 threads,
// 4 elements per thread).
template <typename T>
__global__ __launch_bounds__(256) void gram_reduce_kernel(const T* __restrict__ slabs,
                                                          int32_t nt1, int32_t ksplit,
                                                          int32_t n_red, int accumulate,
                                                          double* __restrict__ G, int32_t first_tile = 0) {
  AGGF_GATED_BODY_BEGIN
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
  AGGF_GATED_BODY_END
}

// ---------------------------------------------------------------------------
// Small and mid-size systems: n_red <= 512, e.g. CLN025 (175 atoms, 97 reduced variables).
// At CLN025 the Gram build sits on the ridge (n_red (n_red+1) / (N s) = 6.8 flop/B against a machine balance of
// ~12), below it the HBM is the bound, above it the MFMA pipe -- in all three the kernel is organised around ONE pass
// over the forces, read the way they lie in HBM: the frames of a stage are one contiguous run of KBS x 3N elements,
// fetched with 16-byte loads into registers while the MFMAs of the previous stage run, parked in LDS as they are, and
// only there turned into the panel the MFMAs read -- constraint-group column sums (`@ con_mat`), dtype conversion and
// zero padding, all LDS -> LDS.  No packed copy of the trajectory exists (the pack + tile pipeline reads F, writes a
// padded copy and reads that again: 3.4x the bytes at CLN025), and only the 16x16 blocks of the upper triangle are
// multiplied, dealt to the waves in equal contiguous shares (template C).  Shapes (template W, KBS, NWV; make_plan
// picks): panel width 32 / 64 / 128 with 32 / 16 / 8 frames per stage, 8 waves, two workgroups per CU; 256 columns
// with 8 or 4 frames, 16 waves, one workgroup per CU; 512 columns with 4 frames, 16 waves and gridDim.y workgroups
// sharing a frame range.
constexpr int SM_MAXVEC = 8;        // 16-byte loads per thread and stage (template NV = 3, 5 or 8)
constexpr int SM_FAST_MEMBERS = 4;  // group members summed without a loop (larger groups: generic tail loop)

template <typename TIn>
static size_t small_raw_bytes(int32_t N, int kbs) { return (size_t)round_up((int64_t)kbs * 3 * N * sizeof(TIn), 16) + 16; }

template <typename TIn, typename TC, int NV, int KBS, int NWV, int W, int C>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 3 : 4) void gram_small_kernel(
    const TIn* __restrict__ F, int64_t T, int32_t N, const int32_t* __restrict__ grp_ptr,
    const int32_t* __restrict__ grp_atoms, int32_t n_red, int64_t frames_per_split, int32_t raw_bytes,
    TC* __restrict__ slabs) {
  using M = Mfma<TC>;
  using acc_t = typename M::acc_t;
  constexpr int SM_THREADS = 64 * NWV;
  // W = panel width in reduced columns (128, 64 or 32, n_red <= W) with KBS = 8 * 128 / W frames per stage: a stage
  // is the same amount of panel whatever the system's size, so the per-stage costs (three barriers, the table-driven
  // group sums over ALL panel columns, the fetch latency) are spread over 2x / 4x the frames of a 64- / 32-column
  // system.  (With the 128-wide panel for everything, 32 atoms streamed at 0.19 of 8 TB/s and 64 atoms at 0.30
  // against CLN025's 0.45: profiles/r04_stream_kernels.jsonl.)
  constexpr int RE = 3 * W, RS = RE + ROW_PAD;              // panel row: elements / stride
  constexpr int WT = W > TILE ? W : TILE;                   // edge of the slab (and of the column tables)
  constexpr int SM_ENT = KBS * RE / SM_THREADS;             // panel entries per thread and stage: 6
  // C = 16x16 blocks of the upper triangle per ACTIVE wave: wave w owns blocks w C .. w C + C - 1 of the row-major list
  // (n_blocks <= NWV C); the waves behind the list skip the MFMA phase, the last active one pads its share with
  // repeats of block 0 whose accumulators are dropped.  Every active wave thus runs the SAME unconditional sequence
  // of MFMAs: with a test per block (blocks dealt round-robin, 3 or 4 per wave at CLN025) the compiler wrapped every
  // conditional MFMA in copies of the accumulator set behind an s_nop for the MFMA's full latency -- ~150 cycles per
  // block step and nothing overlapped (tools/small_probe.hip: MFMA phase 3500 of 10300 cycles per stage at CLN025).
  constexpr int SM_MAXBLK = C;
  static_assert(KBS * RE % SM_THREADS == 0 && KBS >= 4 && (KBS * W == 8 * TILE || (KBS == 4 && W > 2 * TILE) || (KBS == 8 && W == 2 * TILE)), "entry split");
  typedef float __attribute__((ext_vector_type(4))) v16_t;  // one 16-byte piece, whatever the dtype
  // ... as it lies in HBM: F needs element alignment only (a frame block that starts at an odd row of a larger array is
  // 8- or 4-byte aligned) -- still one global_load_dwordx4 per piece, the memory system takes any byte address
  typedef float __attribute__((ext_vector_type(4), aligned(4))) v16g_t;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  TC* panel = reinterpret_cast<TC*>(smem_raw);                         // [KBS][RS]
  TIn* raw = reinterpret_cast<TIn*>(smem_raw + KBS * RS * sizeof(TC));  // SM_KB frames as in HBM
  // (raw_bytes includes one extra zeroed 16-byte piece: the "no member" slot of the table below)
  int32_t* atoms_s = reinterpret_cast<int32_t*>(smem_raw + KBS * RS * sizeof(TC) + raw_bytes);  // [N]
  int32_t* ptr_s = atoms_s + N;                                                                           // [WT + 1]
  // per panel column c = 3 g + d: offsets (3 atom + d) of the first 4 members of group g inside a frame,
  // 0xFFFF = none -- one 8-byte LDS read instead of a chain of dependent ones per member
  unsigned short* memb_s = reinterpret_cast<unsigned short*>(smem_raw + KBS * RS * sizeof(TC) + raw_bytes +
                                                             (((int64_t)N + WT + 1) * 4 + 15) / 16 * 16);  // [RE][4]
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
  for (int g = tid; g <= WT; g += SM_THREADS) ptr_s[g] = g <= n_red ? (grp_ptr ? grp_ptr[g] : g) : (grp_ptr ? grp_ptr[n_red] : n_red);
  const int zero_idx = (raw_bytes - 16) / (int)sizeof(TIn);
  if (tid < 16 / (int)sizeof(TIn)) raw[zero_idx + tid] = (TIn)0;
  __syncthreads();
  bool big_groups = false;  // uniform: some group has more than SM_FAST_MEMBERS members
  bool pair_groups = true;  // uniform: no group has more than two (bond pairs: the common constraint pattern)
  for (int g = 0; g < n_red; ++g) {
    big_groups |= ptr_s[g + 1] - ptr_s[g] > SM_FAST_MEMBERS;
    pair_groups &= ptr_s[g + 1] - ptr_s[g] <= 2;
  }
  for (int c = tid; c < RE; c += SM_THREADS) {
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
      // non-temporal where every byte is read once (one workgroup per frame range; tools/ldsdma_fill.hip: 6.8 against
      // 6.1 TB/s for this load shape); with `parts` workgroups per frame range the siblings meet the frames in their L2
      // Addresses: ONE 32-bit lane offset (tid x 16) beside a wave-uniform base per piece (scalar registers): the loads
      // take the scalar-base form.  With a 64-bit address per piece (5-8 register pairs held across the stage) every
      // instantiation of this kernel spilled at its 128-register cap, and a spill RELOAD is a scratch load: the
      // `s_waitcnt vmcnt(0)` in front of its use also waited for every global load issued before it -- the loads of a
      // stage went out one HBM latency apart (round 5: 2800 of 10300 cycles per stage at CLN025 "issuing" five loads).
      const unsigned voff = (unsigned)tid * 16u;
      if (gridDim.y == 1) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int v = tid + SM_THREADS * i;
          const char* base_i = src + (int64_t)i * (SM_THREADS * 16);
          v16_t x = {0.f, 0.f, 0.f, 0.f};
          if (v < n_vec) x = __builtin_nontemporal_load(reinterpret_cast<const v16g_t*>(base_i + voff));
          hold[i] = x;
        }
      } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int v = tid + SM_THREADS * i;
          const char* base_i = src + (int64_t)i * (SM_THREADS * 16);
          v16_t x = {0.f, 0.f, 0.f, 0.f};
          if (v < n_vec) x = *reinterpret_cast<const v16g_t*>(base_i + voff);
          hold[i] = x;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int v = tid + SM_THREADS * i;
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
    if (!grp_ptr) {
      // no constraint groups: panel column c IS element c of the frame (one LDS read per entry instead of the
      // table's one + four: 32 unconstrained atoms 0.50 -> 0.67 of 8 TB/s)
#pragma unroll
      for (int i = 0; i < SM_ENT; ++i) {
        const int e = tid + SM_THREADS * i;
        const int r = e / RE, c = e - r * RE;
        sum[i] = (TC)raw[c < 3 * n_red ? r * (int)row_in + c : zero_idx];
      }
#pragma unroll
      for (int i = 0; i < SM_ENT; ++i) {
        const int e = tid + SM_THREADS * i;
        const int r = e / RE, c = e - r * RE;
        panel[r * RS + c] = sum[i];
      }
      return;
    }
    if (pair_groups) {
      // at most two members everywhere: the same straight-line sequence with half the table and half the frame reads
      // (one 4-byte table entry and two values per panel entry instead of 8 bytes and four)
      unsigned mem2[SM_ENT];
#pragma unroll
      for (int i = 0; i < SM_ENT; ++i) {
        const int e = tid + SM_THREADS * i;
        mem2[i] = *reinterpret_cast<const unsigned*>(memb_s + (e % RE) * 4);
      }
#pragma unroll
      for (int i = 0; i < SM_ENT; ++i) {
        const int e = tid + SM_THREADS * i;
        const int base = (e / RE) * (int)row_in;
        const int o0 = mem2[i] & 0xFFFF, o1 = mem2[i] >> 16;
        const TC v0 = (TC)raw[o0 == 0xFFFF ? zero_idx : base + o0], v1 = (TC)raw[o1 == 0xFFFF ? zero_idx : base + o1];
        sum[i] = v0 + v1;
      }
#pragma unroll
      for (int i = 0; i < SM_ENT; ++i) {
        const int e = tid + SM_THREADS * i;
        const int r = e / RE, c = e - r * RE;
        panel[r * RS + c] = sum[i];
      }
      return;
    }
    uint2 mem[SM_ENT];
#pragma unroll
    for (int i = 0; i < SM_ENT; ++i) {
      const int e = tid + SM_THREADS * i;  // (frame in stage, reduced column, xyz); padding columns sum nothing
      mem[i] = *reinterpret_cast<const uint2*>(memb_s + (e % RE) * 4);
    }
#pragma unroll
    for (int i = 0; i < SM_ENT; ++i) {
      const int e = tid + SM_THREADS * i;
      const int base = (e / RE) * (int)row_in;
      const int o0 = mem[i].x & 0xFFFF, o1 = mem[i].x >> 16, o2 = mem[i].y & 0xFFFF, o3 = mem[i].y >> 16;
      const TC v0 = (TC)raw[o0 == 0xFFFF ? zero_idx : base + o0], v1 = (TC)raw[o1 == 0xFFFF ? zero_idx : base + o1],
               v2 = (TC)raw[o2 == 0xFFFF ? zero_idx : base + o2], v3 = (TC)raw[o3 == 0xFFFF ? zero_idx : base + o3];
      // (member slots a system never uses are still read: skipping them under a uniform test, or walking only the real
      // columns of the panel, made the compiler wait for every LDS read in turn -- CLN025 4.7 -> 11 / 5.0 ms)
      sum[i] = ((v0 + v1) + v2) + v3;  // members in CSR order, like the column sum of `@ con_mat`
    }
    if (big_groups) {
#pragma unroll
      for (int i = 0; i < SM_ENT; ++i) {
        const int e = tid + SM_THREADS * i;
        const int r = e / RE, c = e - r * RE;
        const int g = c / 3, d = c - 3 * g;
        for (int j = ptr_s[g] + SM_FAST_MEMBERS; j < ptr_s[g + 1]; ++j) sum[i] += (TC)raw[r * (int)row_in + 3 * atoms_s[j] + d];
      }
    }
#pragma unroll
    for (int i = 0; i < SM_ENT; ++i) {
      const int e = tid + SM_THREADS * i;
      const int r = e / RE, c = e - r * RE;
      panel[r * RS + c] = sum[i];
    }
  };

  // this wave's 16x16 blocks of the upper triangle: q = first + wave C + k in row-major order.  More than 256 reduced
  // columns (up to 528 blocks: more accumulators than a CU has registers): gridDim.y workgroups share a frame range,
  // each stages the WHOLE panel and multiplies ITS contiguous share of the block list.
  const int nb = (n_red + 15) / 16;
  const int n_blocks_all = nb * (nb + 1) / 2;
  const int per_part = (n_blocks_all + (int)gridDim.y - 1) / (int)gridDim.y;
  const int first_block = (int)blockIdx.y * per_part;
  const int n_blocks = n_blocks_all - first_block < per_part ? n_blocks_all - first_block : per_part;  // of this workgroup
  const bool mfma_wave = wave * C < n_blocks;
  int b_i[SM_MAXBLK], b_j[SM_MAXBLK];
  bool b_real[SM_MAXBLK];
#pragma unroll
  for (int k = 0; k < SM_MAXBLK; ++k) {
    int q = wave * C + k, bi = 0, rowlen = nb;
    b_real[k] = q < n_blocks;
    q = b_real[k] ? q + first_block : 0;
    while (q >= rowlen) {
      q -= rowlen;
      --rowlen;
      ++bi;
    }
    b_i[k] = bi;
    b_j[k] = bi + q;
  }
  acc_t acc[SM_MAXBLK];
#pragma unroll
  for (int k = 0; k < SM_MAXBLK; ++k) acc[k] = acc_zero<TC>();
  const int off = (lane >> 4) * RS + 3 * (lane & 15);

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
    if (s + 1 < n_it && (AGGF_SMALL_ABL != 2 || s == 0)) fetch(s + 1);
    AGGF_SP_T(q3b);
    if (AGGF_SMALL_ABL != 3 || s == 0) reduce_groups();
    AGGF_SP_T(q4);
    __syncthreads();
    AGGF_SP_T(q5);
    if (AGGF_SMALL_ABL != 1 && mfma_wave) {
      if constexpr (NWV == 16) {
        // 256- and 512-column panels (one workgroup per CU): software pipeline, one MFMA deep -- the operands of product
        // i + 1 are read while product i runs.  The plain loop below reads two operands, waits for them (`s_waitcnt
        // lgkmcnt(0)`: a full LDS latency) and only then issues the MFMA, for every one of the (KBS / 4) x 3 x C products:
        // the compiler reuses ONE operand register pair because everything hoisted further spills at the 128-register
        // cap.  Two pairs (4 more registers) and scheduling barriers around each product keep exactly one read-ahead in
        // flight: the wait in front of an MFMA is for reads issued a whole MFMA earlier.  Same products in the same
        // order: bit-identical sums.  12 GB of frames: 320 atoms 13.8 -> 12.6 ms, 400 atoms 17.5 -> 15.8, 250 atoms
        // 9.1 -> 8.8; with two 8-wave workgroups per CU (the classes below) the other workgroup already covers the
        // latency and the extra registers cost more than they bring (CLN025 4.85 -> 5.06 ms): not used there.
        constexpr int NPROD = (KBS / 4) * 3 * C;
        auto addr_a = [&](int i) { return off + (i / (3 * C)) * 4 * RS + 48 * b_i[i % C] + (i / C) % 3; };
        auto addr_b = [&](int i) { return off + (i / (3 * C)) * 4 * RS + 48 * b_j[i % C] + (i / C) % 3; };
        TC oa[2], ob[2];
        oa[0] = panel[addr_a(0)];
        ob[0] = panel[addr_b(0)];
#pragma unroll
        for (int i = 0; i < NPROD; ++i) {
          if (i + 1 < NPROD) {
            oa[(i + 1) & 1] = panel[addr_a(i + 1)];
            ob[(i + 1) & 1] = panel[addr_b(i + 1)];
          }
          __builtin_amdgcn_sched_barrier(0);  // (the reads stay IN FRONT of the MFMA: otherwise they reuse its operand registers)
          acc[i % C] = M::mma(oa[i & 1], ob[i & 1], acc[i % C]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < KBS / 4; ++kk)
#pragma unroll
          for (int d = 0; d < 3; ++d) {
#pragma unroll
            for (int k = 0; k < C; ++k) {
              const TC a = panel[off + kk * 4 * RS + 48 * b_i[k] + d];
              const TC b = panel[off + kk * 4 * RS + 48 * b_j[k] + d];
              acc[k] = M::mma(a, b, acc[k]);
            }
            // (operand reads of the next group stay behind this point: hoisted over the whole phase they spill 70-100
            // registers at the 128-register cap of two workgroups per CU)
            __builtin_amdgcn_sched_barrier(0);
          }
      }
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
  TC* slab = slabs + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * (WT * WT);
#pragma unroll
  for (int k = 0; k < SM_MAXBLK; ++k)
    if (b_real[k]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(b_i[k] * 16 + M::row(lane, r)) * WT + b_j[k] * 16 + (lane & 15)] = acc[k][r];
    }
}

// Slab sum of the single-tile (small-system) path: one workgroup per row of G instead of the generic kernel's 16
// workgroups per tile (those took 0.84 ms for the 512 slabs of CLN025 -- 15 % of the Gram build).  Thread =
// (column, parity of the slab index); fixed summation order; upper triangle written and mirrored.
template <typename T, int WT>
__global__ __launch_bounds__(2 * WT > 1024 ? 1024 : 2 * WT) void gram_reduce_small_kernel(const T* __restrict__ slabs, int32_t ksplit,
                                                                                      int32_t n_red, int accumulate,
                                                                                      double* __restrict__ G, int32_t parts) {
  constexpr int HALVES = 2 * WT > 1024 ? 1 : 2;
  __shared__ double part_sum[HALVES][WT];
  AGGF_GATED_BODY_BEGIN
  const int row = blockIdx.x, col = threadIdx.x & (WT - 1), half = threadIdx.x / WT;
  // the slab that holds entry (row, col): the workgroup whose share of the block list contains block (row / 16, col / 16)
  int owner = 0;
  if (parts > 1 && col >= row) {
    const int nb = (n_red + 15) / 16, bi = row >> 4, bj = col >> 4;
    const int n_blocks = nb * (nb + 1) / 2, per_part = (n_blocks + parts - 1) / parts;
    owner = (bi * nb - bi * (bi - 1) / 2 + (bj - bi)) / per_part;
  }
  double s = 0.0;
  for (int ks = half; ks < ksplit; ks += HALVES) s += (double)slabs[(((int64_t)ks * parts + owner) * WT + row) * WT + col];
  part_sum[half][col] = s;
  __syncthreads();
  if (half == 0 && row < n_red && col < n_red && col >= row) {
    double tot = part_sum[0][col];
    if (HALVES == 2) tot += part_sum[HALVES - 1][col];
    double* p = G + (int64_t)row * n_red + col;
    *p = accumulate ? *p + tot : tot;
    if (col > row) {
      double* q = G + (int64_t)col * n_red + row;
      *q = accumulate ? *q + tot : tot;
    }
  }
  AGGF_GATED_BODY_END
}

// ---------------------------------------------------------------------------
enum GramStaging { STAGE_DMA8 = 3, STAGE_SMALL = 4 };  // 8-wave LDS-DMA tile kernel / single-tile streaming kernel

struct GramPlan {
  int32_t n_pad, nt1, n_tiles;
  int32_t first_tile = 0;  // tiles with tj < first_tile are skipped (aggf_gram_from_column; DMA8 direct path only)
  int staging;         // GramStaging
  int32_t n_entries;   // tiles computed per split (n_tiles, or fewer with first_tile > 0)
  bool direct;         // gram kernel reads F in place
  int ksplit;
  bool edge = false;     // tile kernel reads rows that are not padded to whole panels (N % 128 != 0, in place)
  bool straddle = false; // ... and the rows are not whole 16-byte pieces: the last frame goes through gram_tail_row_kernel
  bool wide256 = false;  // small-system kernel: 113-128 columns on the 256-column panel (16 waves, 3 blocks per wave)
  int parts = 1;       // small-system kernel above 256 columns: workgroups that share a frame range and split the block list
  int64_t frames_per_split;
  int64_t chunk_frames;  // frames per pack chunk (direct: T)
  size_t slab_bytes, pack_bytes;
};

static int choose_ksplit(int n_tiles, int64_t frames, int kb, int slots, int64_t max_splits, int stage_rows = 0) {
  // kb: 4 = float64 products, 8 = float32 (cost per frame and slab size); stage_rows: frames per LDS stage when that is
  // not kb (float32 frames with float64 products: 8)
  if (stage_rows <= 0) stage_rows = kb;
  // Minimise a simple time model over the split count k: workgroups run in rounds of `slots`
  // (2 per CU); a workgroup costs its frames plus a fixed prologue/epilogue, and every
  // workgroup writes (and the reducer re-reads) one 128x128 slab.
  int64_t hi = ceil_div(frames, (int64_t)stage_rows * 8);  // at least 8 stages per split
  if (hi < 1) hi = 1;
  if (hi > max_splits) hi = max_splits;
  if (hi > 1024) hi = 1024;
  static const char* force_k = getenv("AGGF_GRAM_KSPLIT");  // measurement: a fixed split count (tools/gram_ksplit_bench.py)
  if (force_k && atoi(force_k) > 0) return (int)(atoi(force_k) < hi ? atoi(force_k) : hi);
  const double us_per_frame = 0.64 * 4.0 / kb;     // one LDS stage = 48 MFMAs per wave, 2 waves per SIMD
  const double fixed_frames = 48.0;                // pipeline fill + slab store, in frame units
  const double slab_us = 2.0 * TILE * TILE * (kb == 4 ? 8 : 4) / 2.0e6;  // write + read at ~2 TB/s
  double best_cost = 1e300;
  int64_t best = 1;
  for (int64_t k = 1; k <= hi; ++k) {
    const int64_t blocks = k * n_tiles;
    const double rounds = (double)ceil_div(blocks, slots);
    const double fpb = (double)round_up(ceil_div(frames, k), stage_rows);
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

// One staging for both dtypes: LDS-DMA ring, 8 waves per tile, the stage's DMAs spread over the MFMA groups.
// fp64 at C3: 757 ms (4 waves with the DMAs up front 786 ms, register staging 810 ms, pair tiles and the float32
// "quad" shape no better: profiles/r04_pruned_variants.patch holds those kernels).
static int make_plan(int64_t T, int32_t N, int32_t n_red, int in_dtype, int compute_dtype,
                     bool has_groups, bool aligned, size_t ws_bytes, bool query, GramPlan* p, int32_t first_col = 0,
                     bool tiles_only = false) {  // tiles_only: the caller's kernel is the tile kernel (aggf_gram_pair)
  p->n_pad = (int32_t)round_up(n_red, TILE);
  p->nt1 = p->n_pad / TILE;
  p->n_tiles = p->nt1 * (p->nt1 + 1) / 2;
  // in place: no groups (`aligned`: the two-array form of aggf_gram_pair asks for 16-byte aligned arrays of whole
  // panels; aggf_gram's kernels read F at any element-aligned address); N % 128 != 0 takes the EDGE form of the tile kernel
  // (float32 frames with float64 products too: the tile kernel widens the operands as it reads them from LDS)
  p->edge = N % TILE != 0;
  const bool widen = in_dtype == AGGF_F32 && compute_dtype == AGGF_F64;
  const bool rows16 = ((int64_t)3 * N * (int64_t)dtype_size(in_dtype)) % 16 == 0;
  // rows that are not whole 16-byte pieces (an odd atom count; float32: N % 4 != 0): the piece across a row's end is
  // read into columns nobody uses, and the array's last row -- behind which nothing may be read -- is added by a
  // rank-3 update kernel.  Not for aggf_gram_from_column (the caller's leading block must stay what it is).
  p->straddle = p->edge && !rows16 && !tiles_only && first_col == 0 && T >= 2;
  p->direct = !has_groups && (in_dtype == compute_dtype || (widen && !tiles_only)) && aligned &&
              (!p->edge || (!tiles_only && (rows16 || p->straddle)));
  if (!p->direct) p->straddle = false;
  p->staging = STAGE_DMA8;
  static const char* no_small = getenv("AGGF_GRAM_NO_SMALL");  // tests: force the tiled pipeline on small systems
  // The streaming kernel (gram_small_kernel: the frames pass through LDS once, group sums / conversion / padding on the
  // way, only the 16 x 16 blocks of the upper triangle are multiplied) takes
  //   * n_red <= 128: 8 waves, two workgroups per CU, panel width 32 / 64 / 128 (6.5 ms at CLN025 x 4e6 frames when it
  //     was written; an LDS-DMA ring of 4 frames per stage took 6.8-7.0 ms, 4 x 4 waves more:
  //     profiles/r04_pruned_variants.patch);
  //   * 128 < n_red <= 256 (round 4): a 256-column panel, 16 waves, ONE workgroup per CU -- no packed copy, no padding
  //     to whole 128-tiles inside the products (the pack + tile pipeline spent 22.6 ms on 12 GB of 144-atom frames,
  //     17.7 ms on 192 atoms; now 4.7 / 5.4).  113-128 columns WITHOUT constraint groups too: their 36 blocks are 5 per
  //     wave of the 8-wave form, which spills (128 atoms 5.7 -> 4.4 ms); with groups the table-driven sums over all 768
  //     panel columns cost more than that (117 columns of 175 atoms in pairs: 6.8 against 6.5 ms);
  //   * 256 < n_red <= 480 (512 with float32 products) for layouts the tile kernel cannot read in place: a 512-column
  //     panel and two to four workgroups per frame range, each with a contiguous share of at most 144 of the up to
  //     528 blocks (more accumulators than one CU has registers); every one stages the whole panel -- the frames are
  //     read `parts` times, which 32+ flop/B affords (260 atoms 37.9 -> 9.1 ms, 400 atoms 29.5 -> 17.6).  A layout the
  //     tile kernel reads in place (N a multiple of 128, no groups, no conversion) keeps it from three tiles on (384
  //     atoms 11.4 against 15.2 ms), and from ~480 columns the pack + tile pipeline wins again with float64 products
  //     (500 atoms 19.6 against 27.9 ms).
  const size_t raw_small = (size_t)round_up((int64_t)8 * 3 * N * (int64_t)dtype_size(in_dtype), 16);
  const size_t raw_wide = (size_t)round_up((int64_t)4 * 3 * N * (int64_t)dtype_size(in_dtype), 16);  // 4 frames, 16 waves
  const bool wide_fits = first_col == 0 && raw_wide <= (size_t)5 * 64 * 16 * 16;
  // (three tiles: the streaming kernel also beats the EDGE form of the tile kernel up to ~320 columns -- 260 atoms 9.1
  // against ~17 ms, 360 atoms 18.6 against 12.5; four tiles: the EDGE form wins
  // from ~400 columns with float64 products -- 448 atoms 16.5 against 24.5 ms -- and from ~480 with float32 -- 400 atoms
  // 19.6 against 13.2)
  // The thresholds are the constants of aggf_routing.h, generated from profiles/r05_routing.json (the measured crossovers;
  // tools/routing_sweep.py re-measures them).  AGGF_GRAM_ROUTE (measurement; read per call): "stream" = the streaming
  // kernel wherever it can run, "tile" = never.
  const char* route = getenv("AGGF_GRAM_ROUTE");
  const bool force_stream = route && route[0] == 's', force_tile = route && route[0] == 't';
  const bool f64p = compute_dtype == AGGF_F64;
  const int edge3_max = widen ? routing::stream_edge3_max_cols_widen : routing::stream_edge3_max_cols;
  const int edge4_max = widen ? routing::stream_edge4_max_cols_widen
                              : (f64p ? routing::stream_edge4_max_cols_f64 : routing::stream_edge4_max_cols_f32);
  const bool wide = wide_fits && (p->nt1 == 2 || (force_stream && p->nt1 <= 4) ||
                                  (p->nt1 == 3 && (!p->direct || (p->edge && n_red <= edge3_max))) ||
                                  (p->nt1 == 4 && !p->direct && n_red <= (f64p ? routing::stream_pack4_max_cols_f64 : routing::stream_pack4_max_cols_f32)) ||
                                  (p->nt1 == 4 && p->direct && p->edge && n_red <= edge4_max));
  p->parts = 1;
  p->wide256 = wide_fits && p->nt1 == 1 && n_red > routing::wide256_min_cols && !has_groups;
  // (16-byte loads per thread and stage <= SM_MAXVEC; 3 N + xyz must fit the 16-bit member table)
  if (((p->nt1 == 1 && raw_small <= (size_t)SM_MAXVEC * 64 * 8 * 16) || wide || p->wide256) && !no_small && !force_tile && !tiles_only && N < 21000 && aligned) {
    // one output tile: the fused streaming kernel (group sums + conversion on the way into LDS, upper
    // triangle blocks only); one slab per workgroup, 2 workgroups per CU over the frame axis
    p->staging = STAGE_SMALL;
    p->n_entries = 1;
    p->direct = true;
    p->chunk_frames = T;
    p->pack_bytes = 0;
    const int nb16 = (n_red + 15) / 16, n_blocks = nb16 * (nb16 + 1) / 2;
    // A block's accumulators are 8 registers in float64 and 4 in float32: with float32 products a workgroup takes up
    // to 192 blocks (12 per wave) -- 257-304 columns in ONE workgroup per frame range, up to 512 in three (a workgroup
    // stages the whole panel whatever its share of the blocks).  12 GB of float32 frames, 144 / 192 / 224 / 288 blocks:
    // 267 columns (400 atoms in bond pairs) 10.8 / 7.6 / 7.8 / 7.6 ms, 388 (villin's size) 13.9 / 10.8 / 10.8 / 10.8,
    // 512 columns 15.1 / 12.5 / 12.4 / 47.6 -- beyond 13 blocks per wave the accumulators spill.
    if (p->nt1 > 2) p->parts = (int)ceil_div((int64_t)n_blocks, compute_dtype == AGGF_F32 ? 192 : 144);
    // one resident generation of workgroups (2 per CU; 1 with the 256- and 512-column panels), each looping over
    // strided stages
    int64_t nwg = (int64_t)(p->nt1 >= 2 || p->wide256 ? 1 : 2) * device_cu_count() / p->parts;
    if (nwg < 1) nwg = 1;
    const int64_t n_stage_all = ceil_div(T, 8);
    if (nwg > n_stage_all) nwg = n_stage_all;
    const int64_t edge = p->nt1 > 2 ? 4 * TILE : (p->wide256 ? 2 * TILE : p->n_pad);
    const size_t slab1s = (size_t)p->parts * edge * edge * dtype_size(compute_dtype);
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
  p->n_entries = p->n_tiles;
  p->first_tile = 0;
  if (first_col >= TILE && p->direct) {
    p->first_tile = first_col / TILE;
    if (p->first_tile >= p->nt1) p->first_tile = p->nt1 - 1;  // always at least the last tile column
    p->n_entries = p->n_tiles - p->first_tile * (p->first_tile + 1) / 2;
  }
  const int kb = compute_dtype == AGGF_F64 ? GramCfg<double>::KB : GramCfg<float>::KB;
  const int stage_rows = p->direct && widen ? GramCfg<float>::KB : kb;
  const size_t cs = dtype_size(compute_dtype);
  const int slots = 2 * device_cu_count();
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
    p->ksplit = choose_ksplit(p->n_entries, p->chunk_frames, kb, slots, 1 << 20, stage_rows);
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
    size_t slab_budget = slab1 * (size_t)choose_ksplit(p->n_entries, T, kb, slots, 1 << 20) + 1024;
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
  p->ksplit = choose_ksplit(p->n_entries, p->chunk_frames, kb, slots, max_splits, stage_rows);
  p->slab_bytes = slab1 * p->ksplit;
  return AGGF_OK;
}

// G += x x' over the three xyz components of ONE frame (float64 products whatever the dtypes of the call: exact
// products of the stored values, the same in both triangles): the last frame of an array whose rows are not whole
// 16-byte pieces (GramPlan::straddle).  A thread owns one entry of G; runs behind gram_reduce_kernel on the stream.
template <typename TIn>
__global__ __launch_bounds__(256) void gram_tail_row_kernel(const TIn* __restrict__ row, int32_t n_red, double* __restrict__ G) {
  const int i = blockIdx.y * 16 + threadIdx.y, j = blockIdx.x * 16 + threadIdx.x;
  if (i >= n_red || j >= n_red) return;
  double s = 0.0;
#pragma unroll
  for (int d = 0; d < 3; ++d) s += (double)row[3 * i + d] * (double)row[3 * j + d];
  G[(int64_t)i * n_red + j] += s;
}

template <typename T, bool EDGE = false, typename TS = T>
static int launch_gram(const TS* X, int64_t rows, int64_t ld, const GramPlan& p, T* slabs,
                       int32_t* tile_table, double* G, int32_t n_red, int accumulate, hipStream_t stream,
                       bool straddle = false) {
  constexpr int KB = GramCfg<TS>::KB;
  const int ksplit = p.ksplit;
  int64_t fps = round_up(ceil_div(rows, ksplit), KB);
  if (fps < KB) fps = KB;
  const int64_t nblocks = (int64_t)ksplit * p.n_tiles;
  if (nblocks > 0x7fffffffLL) return fail(AGGF_ERR_ARG, "gram grid too large");
  const size_t lds3 = (size_t)3 * 2 * dma_panel_elems<TS>() * sizeof(TS);
  // The stage barrier sits in front of the last MFMA group instead of behind it (its operands are in registers by then,
  // the 8 MFMAs run while the waves meet), and the DMA piece that goes with that group is issued BEHIND the barrier
  // (template ES = 1, ES_DMA_AFTER).  Same box, back to back: float32 c5 60.2 -> 58.6 ms, c2 3.25 -> 3.16 ms; float64
  // C3 756.4 -> 747.9 ms, c4 297.0 -> 293.9 ms.  With the piece in front of the barrier float64 is SLOWER than the
  // late barrier (757 -> 768 ms): the waves then meet right after the instruction that stalls longest.  Two groups
  // behind the barrier (ES = 2): c5 59.2 against 57.3 ms, C3 789 against 746 ms.
  static thread_local PerDeviceOnce attr_once;
  bool& attr_done = *attr_once.flag();
  if (!attr_done) {
    AGGF_HIP_OK(hipFuncSetAttribute((const void*)gram_tile_dma_kernel<T, 0, 3, 2, 8, true, 1, true, false, EDGE, TS>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
    attr_done = true;
  }
  AGGF_LAUNCH(build_tile_table_kernel, dim3(1), dim3(256), 0, stream, p.nt1, tile_table, p.first_tile);
  AGGF_LAUNCH_OK();
  const int64_t nblk = (int64_t)ksplit * p.n_entries;  // n_entries = tiles actually computed
  AGGF_LAUNCH((gram_tile_dma_kernel<T, 0, 3, 2, 8, true, 1, true, false, EDGE, TS>), dim3((unsigned)round_up(nblk, 512)),
                     dim3(512), lds3, stream, X, rows, ld, p.nt1, p.n_entries, ksplit, tile_table, fps, slabs,
                     (const TS*)nullptr, (int64_t)0, 0, (int32_t)(EDGE ? ld : 0), (int32_t)(straddle ? 1 : 0));
  AGGF_LAUNCH_OK();
  AGGF_LAUNCH_GATED(1024, (gram_reduce_kernel<T>), dim3(p.n_tiles, TILE / 8), dim3(256), 0, stream,
                     slabs, p.nt1, ksplit, n_red, accumulate, G, p.first_tile);
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
  if (p.staging == STAGE_SMALL) {
    constexpr int threads = 512;
    // panel width 32 / 64 / 128 reduced columns with 32 / 16 / 8 frames per stage (see gram_small_kernel); the wider
    // class when the stage's raw frames would not fit the registers that carry them (SM_MAXVEC 16-byte pieces per thread)
    int width = n_red <= 32 ? 32 : (n_red <= 64 ? 64 : TILE);
    while (width < TILE && small_raw_bytes<TIn>(N, 8 * TILE / width) / 16 - 1 > (size_t)SM_MAXVEC * threads) width *= 2;
    if (p.nt1 == 2 || p.wide256) width = 2 * TILE;
    if (p.nt1 > 2) width = 4 * TILE;
    int kbs = width > 2 * TILE ? 4 : 8 * TILE / width;  // (the MFMA's K = 4 frames is the smallest stage)
    // 256 columns: 8 frames per stage where they fit (one workgroup per CU: nothing covers a stage's barriers and group
    // sums but its own MFMAs, and 4 frames are 15 MFMAs per wave)
    // -- up to 5 blocks per wave (~200 columns): beyond, the longer MFMA phase spills (224 atoms 6.3 -> 9.2 ms)
    if (width == 2 * TILE && small_raw_bytes<TIn>(N, 8) / 16 - 1 <= (size_t)5 * 1024 &&
        ((n_red + 15) / 16) * ((n_red + 15) / 16 + 1) / 2 <= 5 * 16)
      kbs = 8;
    const int wt = width > TILE ? width : TILE;
    const int n_thr = width > TILE ? 1024 : threads;
    const size_t raw_bytes = small_raw_bytes<TIn>(N, kbs);
    const size_t lds = (size_t)kbs * (3 * width + ROW_PAD) * sizeof(TC) + raw_bytes +
                       (size_t)round_up(((int64_t)N + wt + 1) * 4, 16) + (size_t)3 * width * 4 * sizeof(unsigned short);
    const int nv = (int)ceil_div((int64_t)(raw_bytes / 16 - 1), n_thr);
    if (nv > SM_MAXVEC || (width > TILE && nv > 5)) return fail(AGGF_ERR_ARG, "aggf_gram: small-system kernel: frame too large");
    const int nb16 = (n_red + 15) / 16, n_blocks = nb16 * (nb16 + 1) / 2;
    const int per_part = (int)ceil_div((int64_t)n_blocks, p.parts);
    const int per_wave = (int)ceil_div((int64_t)per_part, width > TILE ? 16 : 8);  // blocks per active wave (template C)
#define AGGF_SMALL(NVC, KBC, NWC, WC, CC)                                                                            \
  do {                                                                                                               \
    if (lds > 65536) {                                                                                               \
      static thread_local PerDeviceOnce attr_once;                                                                   \
      bool& attr_done = *attr_once.flag();                                                                           \
      if (!attr_done) {                                                                                              \
        AGGF_HIP_OK(hipFuncSetAttribute((const void*)gram_small_kernel<TIn, TC, NVC, KBC, NWC, WC, CC>,              \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));             \
        attr_done = true;                                                                                            \
      }                                                                                                              \
    }                                                                                                                \
    AGGF_LAUNCH((gram_small_kernel<TIn, TC, NVC, KBC, NWC, WC, CC>), dim3((unsigned)p.ksplit, (unsigned)p.parts), \
                       dim3(64 * NWC), lds, stream, reinterpret_cast<const TIn*>(Fv), T, N, grp_ptr, grp_atoms,      \
                       n_red, p.frames_per_split, (int32_t)raw_bytes, slabs);                                        \
  } while (0)
#define AGGF_SMALL_NV(KBC, WC, CC)                                                                                   \
  do {                                                                                                               \
    if (nv <= 3) AGGF_SMALL(3, KBC, 8, WC, CC);                                                                      \
    else if (nv <= 5) AGGF_SMALL(5, KBC, 8, WC, CC);                                                                 \
    else AGGF_SMALL(8, KBC, 8, WC, CC);                                                                              \
  } while (0)
#define AGGF_SMALL_WIDE(WC, CC)                                                                                      \
  do {                                                                                                               \
    if (nv <= 3) AGGF_SMALL(3, 4, 16, WC, CC);                                                                       \
    else AGGF_SMALL(5, 4, 16, WC, CC);                                                                               \
  } while (0)
#define AGGF_SMALL_WIDE8(CC)                                                                                         \
  do {                                                                                                               \
    if (nv <= 3) AGGF_SMALL(3, 8, 16, 2 * TILE, CC);                                                                 \
    else AGGF_SMALL(5, 8, 16, 2 * TILE, CC);                                                                         \
  } while (0)
    if (width == 32) AGGF_SMALL_NV(32, 32, 1);                       // 1 or 3 blocks
    else if (width == 64 && per_wave <= 1) AGGF_SMALL_NV(16, 64, 1);  // 6 blocks
    else if (width == 64) AGGF_SMALL_NV(16, 64, 2);                  // 10
    else if (width == TILE && per_wave <= 2) AGGF_SMALL_NV(8, TILE, 2);   // 15
    else if (width == TILE && per_wave == 3) AGGF_SMALL_NV(8, TILE, 3);   // 21
    else if (width == TILE && per_wave == 4) AGGF_SMALL_NV(8, TILE, 4);   // 28
    else if (width == TILE) AGGF_SMALL_NV(8, TILE, 5);                    // 36
    else if (width == 2 * TILE && kbs == 8 && per_wave <= 3) AGGF_SMALL_WIDE8(3);  // 256-column panel, 16 waves, 8 frames
    else if (width == 2 * TILE && kbs == 8 && per_wave == 4) AGGF_SMALL_WIDE8(4);
    else if (width == 2 * TILE && kbs == 8) AGGF_SMALL_WIDE8(5);
    else if (width == 2 * TILE && per_wave <= 3) AGGF_SMALL_WIDE(2 * TILE, 3);   // 4 frames (the frames of 8 do not fit): 45 blocks
    else if (width == 2 * TILE && per_wave == 4) AGGF_SMALL_WIDE(2 * TILE, 4);   // 55
    else if (width == 2 * TILE && per_wave == 5) AGGF_SMALL_WIDE(2 * TILE, 5);   // 66, 78
    else if (width == 2 * TILE && per_wave == 6) AGGF_SMALL_WIDE(2 * TILE, 6);   // 91
    else if (width == 2 * TILE && per_wave == 7) AGGF_SMALL_WIDE(2 * TILE, 7);   // 105
    else if (width == 2 * TILE && per_wave == 8) AGGF_SMALL_WIDE(2 * TILE, 8);   // 120
    else if (width == 2 * TILE) AGGF_SMALL_WIDE(2 * TILE, 9);                    // 136
    else if (per_wave <= 6) {
      // 512-column panel: 77 - 96 blocks per workgroup (float64 products: with float32 a workgroup has at least 97)
      if constexpr (std::is_same<TC, float>::value) {
        AGGF_SMALL_WIDE(4 * TILE, 7);
      } else {
        if (per_wave <= 5) AGGF_SMALL_WIDE(4 * TILE, 5);
        else AGGF_SMALL_WIDE(4 * TILE, 6);
      }
    }
    else if (per_wave == 7) AGGF_SMALL_WIDE(4 * TILE, 7);
    else if (per_wave == 8) AGGF_SMALL_WIDE(4 * TILE, 8);
    else if (per_wave == 9) AGGF_SMALL_WIDE(4 * TILE, 9);            // <= 144
    else {
      // float32 products only (make_plan): up to 12 blocks per wave, 192 per workgroup
      if constexpr (std::is_same<TC, float>::value) {
        if (per_wave == 10) AGGF_SMALL_WIDE(4 * TILE, 10);
        else if (per_wave == 11) AGGF_SMALL_WIDE(4 * TILE, 11);
        else AGGF_SMALL_WIDE(4 * TILE, 12);
      } else {
        return fail(AGGF_ERR_ARG, "aggf_gram: streaming kernel: more than 144 blocks per workgroup with float64 products");
      }
    }
#undef AGGF_SMALL_WIDE
#undef AGGF_SMALL_WIDE8
#undef AGGF_SMALL_NV
#undef AGGF_SMALL
    AGGF_LAUNCH_OK();
    if (width == 4 * TILE)
      AGGF_LAUNCH_GATED(256, (gram_reduce_small_kernel<TC, 4 * TILE>), dim3(4 * TILE), dim3(1024), 0, stream, slabs, p.ksplit,
                         n_red, accumulate, G, p.parts);
    else if (width > TILE)
      AGGF_LAUNCH_GATED(256, (gram_reduce_small_kernel<TC, 2 * TILE>), dim3(2 * TILE), dim3(4 * TILE), 0, stream, slabs, p.ksplit,
                         n_red, accumulate, G, 1);
    else
      AGGF_LAUNCH_GATED(256, (gram_reduce_small_kernel<TC, TILE>), dim3(TILE), dim3(2 * TILE), 0, stream, slabs, p.ksplit, n_red,
                         accumulate, G, 1);
    AGGF_LAUNCH_OK();
    return AGGF_OK;
  }
  if (p.direct) {
    // TIn == TC, or float32 frames widened inside the tile kernel
    if (p.edge && p.straddle) {
      // all frames but the last through the tile kernel (the piece across the end of row t reads the head of row t + 1),
      // the last one as a rank-3 update of G
      const TIn* F = reinterpret_cast<const TIn*>(Fv);
      int rc = launch_gram<TC, true, TIn>(F, T - 1, (int64_t)N * 3, p, slabs, tile_table, G, n_red, accumulate, stream, true);
      if (rc) return rc;
      AGGF_LAUNCH((gram_tail_row_kernel<TIn>), dim3((unsigned)ceil_div((int64_t)n_red, 16), (unsigned)ceil_div((int64_t)n_red, 16)),
                  dim3(16, 16), 0, stream, F + (T - 1) * (int64_t)N * 3, n_red, G);
      AGGF_LAUNCH_OK();
      return AGGF_OK;
    }
    if (p.edge)
      return launch_gram<TC, true, TIn>(reinterpret_cast<const TIn*>(Fv), T, (int64_t)N * 3, p, slabs, tile_table, G,
                                        n_red, accumulate, stream);
    return launch_gram<TC, false, TIn>(reinterpret_cast<const TIn*>(Fv), T, (int64_t)N * 3, p, slabs, tile_table, G,
                                       n_red, accumulate, stream);
  }
  TC* pack = reinterpret_cast<TC*>(ws + round_up((int64_t)p.slab_bytes, 256));
  const TIn* F = reinterpret_cast<const TIn*>(Fv);
  // The pack pass is HBM-bound, the tile kernel MFMA-bound: the chunk is split into two half-size buffers and chunk
  // i + 1 is packed on a side stream while chunk i is multiplied; a trajectory that fits one chunk is still cut into
  // PACK_MIN_CHUNKS pieces for the same reason.  The pack beside the tile kernel is a SMALL grid of long-lived
  // workgroups with non-temporal accesses (one per CU): a full-size grid's workgroups take turns
  // with the tile kernel's for the CUs and break up the cohorts that share panels in the L2.  C3 with bond pairs,
  // Gram stage (tools/pack_ab.sh): serial 404-407 ms; overlapped with the full grid 562-581; with 32 / 64 / 128 / 192 /
  // 256 / 384 / 512 workgroups 1017 / 633 / 430 / 396 / 397 / 399 / 409 -- 35 ms of pack, 8 of them hidden.
  // AGGF_GRAM_PACK=serial: one buffer, one stream; =chunked: the overlapped form's chunks on one stream (measurements
  // and tests).  AGGF_GRAM_PACK_MIN_FRAMES: test hook -- the smallest chunk, so that small inputs reach the pipeline.
  // Both are read on every call (a test compares the forms inside one process).
  const char* pack_env = getenv("AGGF_GRAM_PACK");
  const char* min_env = getenv("AGGF_GRAM_PACK_MIN_FRAMES");
  constexpr int PACK_MIN_CHUNKS = 8;
  // a chunk must be worth its launches and event waits: >= 8192 frames and >= 4e11 flop (~6 ms of tile kernel)
  int64_t PACK_MIN_FRAMES = (int64_t)(4e11 / (3.0 * (double)p.n_pad * (double)p.n_pad)) + 1;
  if (PACK_MIN_FRAMES < 8192) PACK_MIN_FRAMES = 8192;
  if (min_env && atoll(min_env) >= 8) PACK_MIN_FRAMES = atoll(min_env);
  const size_t row_elems = (size_t)p.n_pad * 3;
  int64_t cf = p.chunk_frames;
  // (up to 1024 columns the tile kernel is short against the pack pass and the throttled pack beside it loses: 700 atoms
  // 31.0 against 27.1 ms, 1000 atoms 32.9 against 31.4 -- serial there; the min-frames hook still reaches the pipeline)
  bool overlap = !(pack_env && pack_env[0] == 's') && T >= 2 * PACK_MIN_FRAMES && p.chunk_frames / 2 >= PACK_MIN_FRAMES &&
                 (p.n_pad > routing::pack_overlap_min_pad || min_env);
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
      // workgroups take turns with the tile kernel's for the CUs and break up the cohorts that share panels in the L2):
      // one per CU (32 / 64 / 128 / 192 / 256 / 384 / 512 workgroups: 1017 / 633 / 430 / 396 / 397 / 399 / 409 ms)
      gy = ceil_div((int64_t)device_cu_count(), (int64_t)gx);
    }
    if (gy > rows) gy = rows;
    if (st != stream)
      AGGF_LAUNCH((pack_groups_kernel<TIn, TC, true>), dim3(gx, (unsigned)gy), dim3(256), 0, st,
                         F + t0 * (int64_t)N * 3, rows, N, grp_ptr, grp_atoms, n_red, p.n_pad, dst);
    else
      AGGF_LAUNCH((pack_groups_kernel<TIn, TC>), dim3(gx, (unsigned)gy), dim3(256), 0, st,
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
    if (rc) {
      // the side stream may still be writing the caller's workspace: join it into `stream` before handing back
      if (overlap && !same_stream && hipEventRecord(pipe->packed[b], pipe->side) == hipSuccess)
        (void)hipStreamWaitEvent(stream, pipe->packed[b], 0);
      return rc;
    }
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
  // (the kernels take F at any element-aligned address: the launch plan does not depend on where a frame block starts,
  // so the workspace query, which sees no pointer, always describes the plan of the call)
  if (((uintptr_t)F & (dtype_size(in_dtype) - 1)) != 0) return fail(AGGF_ERR_ARG, "aggf_gram: F is not aligned to its element size");
  const bool aligned = true;
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
    AGGF_HIP_OK(hipFuncSetAttribute((const void*)gram_tile_dma_kernel<T, 0, 3, 2, 8, true, 1, true, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
    attr_done = true;
  }
  AGGF_LAUNCH(build_tile_table_kernel, dim3(1), dim3(256), 0, stream, p.nt1, tile_table, 0);
  AGGF_LAUNCH_OK();
  const int64_t nblk = (int64_t)ksplit * p.n_tiles;
  if (nblk > 0x7fffff00LL) return fail(AGGF_ERR_ARG, "gram grid too large");
  AGGF_LAUNCH((gram_tile_dma_kernel<T, 0, 3, 2, 8, true, 1, true, true>), dim3((unsigned)round_up(nblk, 512)),
                     dim3(512), lds3, stream, F, rows, (int64_t)N * 3, p.nt1, p.n_tiles, ksplit, tile_table, fps, slabs, F2,
                     (int64_t)N2 * 3, N / TILE);
  AGGF_LAUNCH_OK();
  AGGF_LAUNCH_GATED(1024, (gram_reduce_kernel<T>), dim3(p.n_tiles, TILE / 8), dim3(256), 0, stream, slabs, p.nt1, ksplit,
                     N + N2, accumulate, G, 0);
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" size_t aggf_gram_pair_workspace_bytes(int64_t T, int32_t N, int32_t N2, int dtype) {
  if (T <= 0 || N <= 0 || N2 <= 0) return 0;
  GramPlan p;
  make_plan(T, N + N2, N + N2, dtype, dtype, false, true, 0, true, &p, 0, true);
  return table_bytes(p) + (size_t)round_up((int64_t)p.slab_bytes, 256) + p.pack_bytes + 1024;
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
  int rc = make_plan(T, N + N2, N + N2, dtype, dtype, false, true, ws_bytes, false, &p, 0, true);
  if (rc) return rc;
  if (!p.direct || p.staging == STAGE_SMALL) return fail(AGGF_ERR_ARG, "aggf_gram_pair: unsupported layout");
  if (table_bytes(p) + (size_t)round_up((int64_t)p.slab_bytes, 256) > ws_bytes)
    return fail(AGGF_ERR_WORKSPACE, "aggf_gram_pair: workspace too small");
  char* w = reinterpret_cast<char*>(ws);
  if (dtype == AGGF_F64)
    return gram_pair_typed<double>((const double*)F, (const double*)F2, T, N, N2, p, w, G, accumulate, stream);
  return gram_pair_typed<float>((const float*)F, (const float*)F2, T, N, N2, p, w, G, accumulate, stream);
}

extern "C" int aggf_gram_from_column(const void* F, int64_t T, int32_t N, int in_dtype, int compute_dtype,
                                     int32_t n_red, int32_t first_col, double* G, int accumulate, void* ws,
                                     size_t ws_bytes, void* stream_v) {
  return gram_impl(F, T, N, in_dtype, compute_dtype, nullptr, nullptr, n_red, first_col, G, accumulate, ws, ws_bytes,
                   stream_v);
}
