// K1w: the streaming Gram kernel with PRODUCER and CONSUMER waves (included by aggf_gram.hip).
//
// Same job as gram_small_kernel -- `reg_mat.T @ reg_mat` with the column sums of `@ con_mat`, dtype conversion and
// padding done on the way through LDS, one pass over the forces as they lie in HBM (qp/qplinear.py:66-71) -- for up
// to 512 reduced columns, but the two halves of a stage no longer take turns:
//   * 4 PRODUCER waves fetch the next stage's frames (16-byte global loads into registers, one stage ahead), park them
//     in LDS as they are and turn them into the MFMA panel (group sums LDS -> LDS) in the panel buffer the consumers
//     are NOT reading;
//   * 8 CONSUMER waves (two per SIMD) do nothing but operand reads and MFMAs on the other panel buffer: C blocks of
//     the upper triangle each, one unconditional sequence.
// gram_small_kernel ran fetch wait -> park -> barrier -> sums -> barrier -> MFMA -> barrier with every wave in every
// phase: with one workgroup per CU (the 256- and 512-column classes) nothing covered the non-MFMA phases (400 atoms:
// 0.44 of the fp64 MFMA peak; with constraint groups 0.37-0.45), with two per CU (CLN025) the phases overlapped only
// partly (0.43 of 8 TB/s, profiles/r04_small_ablate.jsonl).  Here the matrix pipe and the LDS / VALU work of the
// group sums run side by side on every SIMD (separate pipes: MI355X_MICROARCH.md, Wave scheduling), and a stage has
// TWO workgroup barriers: park | sums on the producer side, half the MFMAs each on the consumer side.
//
// More than 8 C blocks (C <= 16: 128 accumulator registers of the 168 a wave may hold at three waves per SIMD):
// `parts` workgroups share a frame range and split the block list, as in gram_small_kernel; the workgroups of a frame
// range sit on ONE XCD (consecutive multiples of 8 in the block index) so that the frames come from HBM once.
#pragma once

namespace aggf {

constexpr int WS_PROD = 4, WS_CONS = 8, WS_THREADS = 64 * (WS_PROD + WS_CONS);

// AGGF_WS_ABL (tools/ws_probe.hip only): one of the three activities removed -- 1 = no operand reads / MFMAs,
// 2 = no global loads and no parking after the prologue, 3 = no group sums after the prologue
#ifndef AGGF_WS_ABL
#define AGGF_WS_ABL 0
#endif
#ifdef AGGF_WS_PROF
// tools/ws_probe.hip: shader cycles per wave -- producers: [0] load issue, [1] group sums, [2] load wait + park, [3] barrier;
// consumers: [4] operand reads + MFMAs, [5] barrier; [6] stages x producer waves, [7] stages x consumer waves
__device__ unsigned long long aggf_ws_prof[8];
#define AGGF_WP_T(x) const uint64_t x = __builtin_readcyclecounter()
#else
#define AGGF_WP_T(x)
#endif
// blocks per consumer wave: accumulators of at most 96 (float64: 12 blocks x 8) / 64 (float32: 16 x 4) registers of
// the 168 a wave may hold at three waves per SIMD (14 and 16 float64 blocks spilled: -Rpass-analysis=kernel-resource-usage)
template <typename TC>
constexpr int ws_max_c() { return sizeof(TC) == 8 ? 12 : 16; }

// LDS accesses of the PRODUCER waves beside their own pending LDS-DMAs.  The compiler's wait-count pass puts an
// `s_waitcnt vmcnt(0)` in front of every LDS access it can see while an LDS-DMA of the same wave is in flight (it cannot
// tell the two raw buffers apart), which would serialise "issue the next stage's DMAs, then sum the landed stage" into
// one HBM latency per stage (measured: 7900 of 14000 cycles per stage "issuing" eight DMAs).  Inline assembly is
// invisible to that pass: the group sums read and write LDS through these, and wait for their reads themselves
// (ws_lds_wait).  Correctness is the kernel's own business: a buffer is read only after the vmcnt(0) + barrier that
// followed its DMAs.
__device__ __forceinline__ unsigned ws_lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ float ws_lds_read(const float* p) {
  float v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(ws_lds_addr(p)));
  return v;
}
__device__ __forceinline__ double ws_lds_read(const double* p) {
  double v;
  asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(ws_lds_addr(p)));
  return v;
}
__device__ __forceinline__ uint2 ws_lds_read(const uint2* p) {
  unsigned long long v;
  asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(ws_lds_addr(p)));
  return make_uint2((unsigned)v, (unsigned)(v >> 32));
}
__device__ __forceinline__ void ws_lds_write(float* p, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(ws_lds_addr(p)), "v"(v) : "memory"); }
__device__ __forceinline__ void ws_lds_write(double* p, double v) { asm volatile("ds_write_b64 %0, %1" ::"v"(ws_lds_addr(p)), "v"(v) : "memory"); }
__device__ __forceinline__ void ws_lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// a value read by ws_lds_read may be used only behind the wait: the empty statement (ordered behind the wait like every
// volatile asm) redefines the value as far as the compiler knows, so no use of it can be scheduled in front
__device__ __forceinline__ void ws_lds_ready(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void ws_lds_ready(double& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void ws_lds_ready(uint2& v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); }

// panel row stride (elements): 3 n16 + pad with stride % 32 == 16 -- the two frame rows a 32-lane half of an operand
// read touches then hit disjoint banks (f64: 2 RS dwords = 32 mod 64; f32: RS dwords = 16 mod 32)
__host__ __device__ static inline int ws_row_stride(int n16) { return 3 * n16 + (((3 * n16) & 31) == 0 ? 16 : 32); }

struct WsLds {
  size_t panel_bytes, raw_bytes, table_bytes, total;
  int32_t n_vec;  // 16-byte pieces of a stage's frames
};
template <typename TIn, typename TC>
static WsLds ws_lds(int32_t N, int32_t n_red, int kbs) {
  WsLds l;
  const int n16 = (int)round_up(n_red, 16);
  l.panel_bytes = (size_t)kbs * ws_row_stride(n16) * sizeof(TC);
  const int64_t raw = round_up((int64_t)kbs * 3 * N * (int64_t)sizeof(TIn), 16);
  l.n_vec = (int32_t)(raw / 16);
  l.raw_bytes = (size_t)round_up(raw + 16, 1024);  // + the zeroed "no member" slot; whole 1 KiB DMA pieces
  l.table_bytes = (size_t)round_up(((int64_t)N + n16 + 1) * 4, 16) + (size_t)3 * n16 * 4 * sizeof(unsigned short);
  l.total = 2 * l.panel_bytes + 2 * l.raw_bytes + l.table_bytes;
  return l;
}

template <typename TIn, typename TC, int KBS, int C>
__global__ __launch_bounds__(WS_THREADS, 3) void gram_ws_kernel(
    const TIn* __restrict__ F, int64_t T, int32_t N, const int32_t* __restrict__ grp_ptr,
    const int32_t* __restrict__ grp_atoms, int32_t n_red, int32_t nwg_x, int32_t parts, int32_t slab_edge,
    int32_t raw_bytes, TC* __restrict__ slabs) {
  using M = Mfma<TC>;
  using acc_t = typename M::acc_t;
  static_assert(KBS == 4 || KBS == 8, "frames per stage");
  const int n16 = (n_red + 15) & ~15, RE = 3 * n16, RS = ws_row_stride(n16);
  const int64_t row_in = (int64_t)N * 3;
  const int64_t stage_bytes = (int64_t)KBS * row_in * (int64_t)sizeof(TIn);  // a multiple of 16 (KBS >= 4)
  const int n_vec = (int)(stage_bytes / 16);
  const int zero_idx = (int)(stage_bytes / (int64_t)sizeof(TIn));             // the zeroed 16-byte piece behind the frames

  extern __shared__ __attribute__((aligned(16))) char ws_smem[];
  TC* panel0 = reinterpret_cast<TC*>(ws_smem);
  TC* panel1 = panel0 + (size_t)KBS * RS;
  char* raw0 = ws_smem + 2 * (size_t)KBS * RS * sizeof(TC);                    // two raw buffers: frames as in HBM
  char* tab = raw0 + 2 * (size_t)raw_bytes;
  int32_t* atoms_s = reinterpret_cast<int32_t*>(tab);      // [N]
  int32_t* ptr_s = atoms_s + N;                            // [n16 + 1]
  unsigned short* memb_s = reinterpret_cast<unsigned short*>(tab + (((int64_t)N + n16 + 1) * 4 + 15) / 16 * 16);  // [RE][4]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= WS_CONS;                   // waves 8..11: one per SIMD beside two consumers

  // workgroup -> (frame-range index x, part): the parts of a frame range are consecutive multiples of 8 apart
  const int b = blockIdx.x;
  const int full = (nwg_x / 8) * 8;
  int x, part;
  if (b < full * parts) {
    part = (b >> 3) % parts;
    x = (b & 7) + 8 * (b / (8 * parts));
  } else {
    const int r = b - full * parts;
    x = full + r / parts;
    part = r % parts;
  }
  // stages are dealt round-robin to the frame-range indices: the workgroups that run together read one dense window
  const int64_t n_stage_all = (T + KBS - 1) / KBS;
  const int n_it = x < n_stage_all ? (int)((n_stage_all - 1 - x) / nwg_x + 1) : 0;
  auto stage_t0 = [&](int k) { return ((int64_t)x + (int64_t)k * nwg_x) * KBS; };

  // ---- tables (all waves), as in gram_small_kernel
  for (int a = tid; a < N; a += WS_THREADS) atoms_s[a] = grp_atoms ? grp_atoms[a] : a;
  for (int g = tid; g <= n16; g += WS_THREADS) ptr_s[g] = g <= n_red ? (grp_ptr ? grp_ptr[g] : g) : (grp_ptr ? grp_ptr[n_red] : n_red);
  if (tid < 2 * (16 / (int)sizeof(TIn))) {
    const int which = tid / (16 / (int)sizeof(TIn)), k = tid % (16 / (int)sizeof(TIn));
    reinterpret_cast<TIn*>(raw0 + (size_t)which * raw_bytes)[zero_idx + k] = (TIn)0;
  }
  __syncthreads();
  bool big_groups = false;
  for (int g = 0; g < n_red; ++g) big_groups |= ptr_s[g + 1] - ptr_s[g] > SM_FAST_MEMBERS;
  for (int c = tid; c < RE; c += WS_THREADS) {
    const int g = c / 3, d = c - 3 * g;
#pragma unroll
    for (int j = 0; j < SM_FAST_MEMBERS; ++j)
      memb_s[c * 4 + j] = (ptr_s[g] + j < ptr_s[g + 1]) ? (unsigned short)(3 * atoms_s[ptr_s[g] + j] + d) : (unsigned short)0xFFFF;
  }
  __syncthreads();

  if (producer) {
    // =========================== PRODUCER =====================================================================
    const int pw = wave - WS_CONS;                         // 0..3
    // The producers are the youngest waves of their SIMDs and lose every issue arbitration to the two consumers beside
    // them (same priority: by age): eight global loads took 8000 cycles to issue, the group sums 3x their time alone
    // (profiles/r05_ws_ablation.txt).  They issue a few hundred instructions per stage against the consumers' MFMA
    // stream: at a higher priority they get them in at once and cost the consumers almost nothing.
    __builtin_amdgcn_s_setprio(3);
    // stage s -> registers (16-byte pieces of the contiguous run of KBS frames, all of them in flight at once) -> raw
    // buffer s & 1.  (The first version filled the raw buffers by LDS-DMA: beside the consumers' operand reads an
    // LDS-DMA instruction took ~960 cycles to issue -- 7900 cycles per stage at 175 atoms, 2.6 TB/s for the chip; a
    // global load to registers issues at once, and the ds_write of the parked pieces is ordinary LDS traffic.)
    constexpr int NVMAX = 12;                              // pieces per producer thread and stage (host: <= 48 KB per stage)
    const int pt = pw * 64 + lane;
    typedef float __attribute__((ext_vector_type(4))) v16_t;
    v16_t hold[NVMAX];
    bool hold_full = false;                                // (uniform) the held stage is a full one: park all pieces
    int64_t hold_valid = 0;
    auto fetch = [&](int s) {
      const int64_t t0 = stage_t0(s);
      const char* src = reinterpret_cast<const char*>(F + t0 * row_in);
      hold_full = t0 + KBS <= T;
      hold_valid = (T - t0 < KBS ? T - t0 : KBS) * row_in * (int64_t)sizeof(TIn);
      if (hold_full) {
        // non-temporal where every byte is read once (6.8 against 6.1 TB/s: profiles/r05_ldsdma_fill.jsonl); with
        // `parts` workgroups per frame range the siblings meet the frames in their XCD's L2: default policy
        if (parts == 1) {
#pragma unroll
          for (int i = 0; i < NVMAX; ++i) {
            const int v = pt + 256 * i;
            v16_t xv = {0.f, 0.f, 0.f, 0.f};
            if (v < n_vec) xv = __builtin_nontemporal_load(reinterpret_cast<const v16_t*>(src + (int64_t)v * 16));
            hold[i] = xv;
          }
        } else {
#pragma unroll
          for (int i = 0; i < NVMAX; ++i) {
            const int v = pt + 256 * i;
            v16_t xv = {0.f, 0.f, 0.f, 0.f};
            if (v < n_vec) xv = *reinterpret_cast<const v16_t*>(src + (int64_t)v * 16);
            hold[i] = xv;
          }
        }
      } else {
        // the last stage of the trajectory holds fewer frames: element by element, zeros behind the end
#pragma unroll
        for (int i = 0; i < NVMAX; ++i) {
          const int v = pt + 256 * i;
          v16_t xv = {0.f, 0.f, 0.f, 0.f};
          if (v < n_vec) {
            const int64_t off = (int64_t)v * 16;
            TIn tmp[16 / sizeof(TIn)];
#pragma unroll
            for (int k = 0; k < (int)(16 / sizeof(TIn)); ++k)
              tmp[k] = off + (k + 1) * (int64_t)sizeof(TIn) <= hold_valid ? reinterpret_cast<const TIn*>(src + off)[k] : (TIn)0;
            xv = *reinterpret_cast<v16_t*>(tmp);
          }
          hold[i] = xv;
        }
      }
    };
    auto park = [&](int s) {
      char* dst = raw0 + (size_t)(s & 1) * raw_bytes;
#pragma unroll
      for (int i = 0; i < NVMAX; ++i) {
        const int v = pt + 256 * i;
        if (v < n_vec) reinterpret_cast<v16_t*>(dst)[v] = hold[i];
      }
    };
    // raw frames -> panel: wave pw takes frames pw, pw + 4 of the stage, a lane four columns 64 apart per pass.  Every
    // LDS access goes through the ws_lds_* accessors (see there); the values of a pass are read, waited for, summed.
    auto sums = [&](const char* rawb, TC* panel) {
      const TIn* rw = reinterpret_cast<const TIn*>(rawb);
      for (int r = pw; r < KBS; r += WS_PROD) {
        const int base = r * (int)row_in;
        TC* prow = panel + r * RS;
        if (!grp_ptr) {
          for (int c0 = lane; c0 < RE; c0 += 256) {
            TIn v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int c = c0 + 64 * u;
              v[u] = ws_lds_read(rw + (c < 3 * n_red ? base + c : zero_idx));
            }
            ws_lds_wait();
#pragma unroll
            for (int u = 0; u < 4; ++u) ws_lds_ready(v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (c0 + 64 * u < RE) ws_lds_write(prow + c0 + 64 * u, (TC)v[u]);
          }
          continue;
        }
        for (int c0 = lane; c0 < RE; c0 += 256) {
          uint2 mem[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int c = c0 + 64 * u;
            mem[u] = ws_lds_read(reinterpret_cast<const uint2*>(memb_s + (c < RE ? c : 0) * 4));
          }
          ws_lds_wait();
#pragma unroll
          for (int u = 0; u < 4; ++u) ws_lds_ready(mem[u]);
          TIn v[4][4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int o0 = mem[u].x & 0xFFFF, o1 = mem[u].x >> 16, o2 = mem[u].y & 0xFFFF, o3 = mem[u].y >> 16;
            v[u][0] = ws_lds_read(rw + (o0 == 0xFFFF ? zero_idx : base + o0));
            v[u][1] = ws_lds_read(rw + (o1 == 0xFFFF ? zero_idx : base + o1));
            v[u][2] = ws_lds_read(rw + (o2 == 0xFFFF ? zero_idx : base + o2));
            v[u][3] = ws_lds_read(rw + (o3 == 0xFFFF ? zero_idx : base + o3));
          }
          ws_lds_wait();
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) ws_lds_ready(v[u][q]);
          TC sum[4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            sum[u] = (((TC)v[u][0] + (TC)v[u][1]) + (TC)v[u][2]) + (TC)v[u][3];  // members in CSR order, like the column sum of `@ con_mat`
          if (big_groups) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int c = c0 + 64 * u;
              if (c >= RE) continue;
              const int g = c / 3, d = c - 3 * g;
              for (int j = ptr_s[g] + SM_FAST_MEMBERS; j < ptr_s[g + 1]; ++j) {
                TIn x = ws_lds_read(rw + base + 3 * atoms_s[j] + d);
                ws_lds_wait();
                ws_lds_ready(x);
                sum[u] += (TC)x;
              }
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (c0 + 64 * u < RE) ws_lds_write(prow + c0 + 64 * u, sum[u]);
        }
      }
    };
    // prologue: stage 0 -> panel0, stage 1 parked
    if (n_it > 0) {
      fetch(0);
      park(0);
    }
    __syncthreads();                                       // (P1) raw[0] = stage 0, from every producer wave
    if (n_it > 1) fetch(1);
    if (n_it > 0) sums(raw0, panel0);
    if (n_it > 1) park(1);
    ws_lds_wait();
    __syncthreads();                                       // (P2) panel0 = stage 0, raw[1] = stage 1
#ifdef AGGF_WS_PROF
    uint64_t pf[4] = {0, 0, 0, 0};
#endif
    for (int s = 0; s < n_it; ++s) {
      // the consumers multiply stage s (panel s & 1).  Stage s + 2 leaves HBM for the registers; stage s + 1
      // (raw[(s + 1) & 1], parked before the last barrier) becomes the other panel; then the registers are parked in
      // raw[s & 1], whose frames -- stage s -- were summed one stage ago.
      AGGF_WP_T(q0);
      if (AGGF_WS_ABL != 2 && s + 2 < n_it) fetch(s + 2);
      AGGF_WP_T(q1);
      if (AGGF_WS_ABL != 3 && s + 1 < n_it) sums(raw0 + (size_t)((s + 1) & 1) * raw_bytes, (s & 1) ? panel0 : panel1);
      ws_lds_wait();
      AGGF_WP_T(q2);
      if (AGGF_WS_ABL != 2 && s + 2 < n_it) park(s + 2);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      AGGF_WP_T(q3);
      __syncthreads();
#ifdef AGGF_WS_PROF
      const uint64_t q4 = __builtin_readcyclecounter();
      pf[0] += q1 - q0; pf[1] += q2 - q1; pf[2] += q3 - q2; pf[3] += q4 - q3;
#endif
    }
#ifdef AGGF_WS_PROF
    if (lane == 0) {
      for (int i = 0; i < 4; ++i) atomicAdd(&aggf_ws_prof[i], (unsigned long long)pf[i]);
      atomicAdd(&aggf_ws_prof[6], (unsigned long long)n_it);
    }
#endif
    return;
  }

  // ============================= CONSUMER =======================================================================
  const int nb = n16 >> 4;
  const int n_blocks_all = nb * (nb + 1) / 2;
  const int per_part = (n_blocks_all + parts - 1) / parts;
  const int first_block = part * per_part;
  const int n_blocks = n_blocks_all - first_block < per_part ? n_blocks_all - first_block : per_part;
  const bool mfma_wave = wave * C < n_blocks;
  int b_i[C], b_j[C];
  bool b_real[C];
#pragma unroll
  for (int k = 0; k < C; ++k) {
    int q = wave * C + k, bi = 0, rowlen = nb;
    b_real[k] = q < n_blocks;
    q = b_real[k] ? q + first_block : 0;
    while (q >= rowlen) {
      q -= rowlen;
      --rowlen;
      ++bi;
    }
    b_i[k] = __builtin_amdgcn_readfirstlane(48 * bi);
    b_j[k] = __builtin_amdgcn_readfirstlane(48 * (bi + q));
  }
  acc_t acc[C];
#pragma unroll
  for (int k = 0; k < C; ++k) acc[k] = acc_zero<TC>();
  const int off = (lane >> 4) * RS + 3 * (lane & 15);
  __syncthreads();                                         // (P1)
  __syncthreads();                                         // (P2)
#ifdef AGGF_WS_PROF
  uint64_t cf[2] = {0, 0};
#endif
  for (int s = 0; s < n_it; ++s) {
    const TC* panel = (s & 1) ? panel1 : panel0;
    AGGF_WP_T(c0);
    if (AGGF_WS_ABL != 1 && mfma_wave) {
#pragma unroll
      for (int kk = 0; kk < KBS / 4; ++kk)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
#pragma unroll
          for (int k = 0; k < C; ++k) {
            const TC a = panel[off + kk * 4 * RS + b_i[k] + d];
            const TC bv = panel[off + kk * 4 * RS + b_j[k] + d];
            acc[k] = M::mma(a, bv, acc[k]);
          }
          // (operand reads of the next step stay behind this point: hoisted over the whole stage they spill)
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every operand read of this stage has returned
    AGGF_WP_T(c1);
    __syncthreads();
#ifdef AGGF_WS_PROF
    const uint64_t c2 = __builtin_readcyclecounter();
    cf[0] += c1 - c0; cf[1] += c2 - c1;
#endif
  }
#ifdef AGGF_WS_PROF
  if (lane == 0) {
    atomicAdd(&aggf_ws_prof[4], (unsigned long long)cf[0]);
    atomicAdd(&aggf_ws_prof[5], (unsigned long long)cf[1]);
    atomicAdd(&aggf_ws_prof[7], (unsigned long long)n_it);
  }
#endif
  TC* slab = slabs + ((int64_t)x * parts + part) * ((int64_t)slab_edge * slab_edge);
#pragma unroll
  for (int k = 0; k < C; ++k)
    if (b_real[k]) {
      const int bi = b_i[k] / 48, bj = b_j[k] / 48;
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(bi * 16 + M::row(lane, r)) * slab_edge + bj * 16 + (lane & 15)] = acc[k][r];
    }
}

}  // namespace aggf
