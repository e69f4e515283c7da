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
// blocks per consumer wave: accumulators of at most 96 (float64: 12 blocks x 8) / 64 (float32: 16 x 4) registers of
// the 168 a wave may hold at three waves per SIMD (14 and 16 float64 blocks spilled: -Rpass-analysis=kernel-resource-usage)
template <typename TC>
constexpr int ws_max_c() { return sizeof(TC) == 8 ? 12 : 16; }

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
    // stage s -> raw buffer s & 1 by LDS-DMA: one wave-instruction = 64 lanes x 16 B = 1 KiB, contiguous in LDS and in
    // HBM (a stage is one contiguous run of KBS frames); the last stage of the trajectory may hold fewer frames: it goes
    // through registers, element by element, with zeros behind the end
    auto fetch = [&](int s) {
      const int64_t t0 = stage_t0(s);
      char* dst = raw0 + (size_t)(s & 1) * raw_bytes;
      if (t0 + KBS <= T) {
        const char* src = reinterpret_cast<const char*>(F + t0 * row_in);
        // non-temporal (aux = 2) where every byte is read once: 7.1 against 6.4 TB/s (profiles/r05_ldsdma_fill.jsonl);
        // with `parts` workgroups per frame range the siblings meet the frames in their XCD's L2: default policy
        if (parts == 1) {
          for (int v0 = pw * 64; v0 < n_vec; v0 += 64 * WS_PROD) {
            if (v0 + lane < n_vec)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (int64_t)(v0 + lane) * 16),
                                               (__attribute__((address_space(3))) void*)(dst + (size_t)v0 * 16), 16, 0, 2);
          }
        } else {
          for (int v0 = pw * 64; v0 < n_vec; v0 += 64 * WS_PROD) {
            if (v0 + lane < n_vec)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (int64_t)(v0 + lane) * 16),
                                               (__attribute__((address_space(3))) void*)(dst + (size_t)v0 * 16), 16, 0, 0);
          }
        }
      } else {
        const int64_t valid = (T - t0) * row_in, all = (int64_t)KBS * row_in;
        const TIn* src = F + t0 * row_in;
        for (int64_t e = pw * 64 + lane; e < all; e += 64 * WS_PROD) reinterpret_cast<TIn*>(dst)[e] = e < valid ? src[e] : (TIn)0;
      }
    };
    // raw frames -> panel: wave pw takes frames pw, pw + 4 of the stage, a lane four columns 64 apart per pass
    auto sums = [&](const char* rawb, TC* panel) {
      const TIn* rw = reinterpret_cast<const TIn*>(rawb);
      for (int r = pw; r < KBS; r += WS_PROD) {
        const int base = r * (int)row_in;
        TC* prow = panel + r * RS;
        if (!grp_ptr) {
          for (int c0 = lane; c0 < RE; c0 += 256) {
            TC v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int c = c0 + 64 * u;
              v[u] = (TC)rw[c < 3 * n_red ? base + c : zero_idx];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (c0 + 64 * u < RE) prow[c0 + 64 * u] = v[u];
          }
          continue;
        }
        for (int c0 = lane; c0 < RE; c0 += 256) {
          uint2 mem[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int c = c0 + 64 * u;
            mem[u] = *reinterpret_cast<const uint2*>(memb_s + (c < RE ? c : 0) * 4);
          }
          TC sum[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int o0 = mem[u].x & 0xFFFF, o1 = mem[u].x >> 16, o2 = mem[u].y & 0xFFFF, o3 = mem[u].y >> 16;
            const TC v0 = (TC)rw[o0 == 0xFFFF ? zero_idx : base + o0], v1 = (TC)rw[o1 == 0xFFFF ? zero_idx : base + o1],
                     v2 = (TC)rw[o2 == 0xFFFF ? zero_idx : base + o2], v3 = (TC)rw[o3 == 0xFFFF ? zero_idx : base + o3];
            sum[u] = ((v0 + v1) + v2) + v3;  // members in CSR order, like the column sum of `@ con_mat`
          }
          if (big_groups) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int c = c0 + 64 * u;
              if (c >= RE) continue;
              const int g = c / 3, d = c - 3 * g;
              for (int j = ptr_s[g] + SM_FAST_MEMBERS; j < ptr_s[g + 1]; ++j) sum[u] += (TC)rw[base + 3 * atoms_s[j] + d];
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (c0 + 64 * u < RE) prow[c0 + 64 * u] = sum[u];
        }
      }
    };
    auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); };

    // prologue: stage 0 -> panel0, stage 1 on its way
    if (n_it > 0) fetch(0);
    landed();
    __syncthreads();                                       // (P1) raw[0] = stage 0, from every producer wave
    if (n_it > 1) fetch(1);
    if (n_it > 0) sums(raw0, panel0);
    landed();
    __syncthreads();                                       // (P2) panel0 = stage 0, raw[1] = stage 1
    for (int s = 0; s < n_it; ++s) {
      // the consumers multiply stage s (panel s & 1); stage s + 2 leaves HBM for raw[s & 1] (whose frames -- stage s --
      // were summed one stage ago), stage s + 1 (raw[(s + 1) & 1], landed before the last barrier) becomes the other panel
      if (s + 2 < n_it) fetch(s + 2);
      if (s + 1 < n_it) sums(raw0 + (size_t)((s + 1) & 1) * raw_bytes, (s & 1) ? panel0 : panel1);
      landed();
      __syncthreads();
    }
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
  for (int s = 0; s < n_it; ++s) {
    const TC* panel = (s & 1) ? panel1 : panel0;
    if (mfma_wave) {
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
    __syncthreads();
  }
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
