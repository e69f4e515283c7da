// Shader-clock probe: is the Gram kernel's gap to the MFMA peak a clock (power) effect?
// A one-wave kernel on a second stream samples s_memtime (shader cycles) against s_memrealtime
// (100 MHz reference) while the Gram kernel variants run; the ratio is the shader frequency.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/clock_probe.hip aggforce_amd/csrc/aggf_util.hip -o /tmp/clock_probe
#define AGGF_GRAM_PROF 1
#include "../aggforce_amd/csrc/aggf_gram.hip"

using namespace aggf;

__global__ void probe_kernel(uint64_t duration_ticks, uint64_t* samples, int max_samples, int* n_out) {
  if (threadIdx.x != 0) return;
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  int n = 0;
  uint64_t next = r0;
  while (true) {
    const uint64_t r = __builtin_amdgcn_s_memrealtime();
    if (r - r0 >= duration_ticks || n >= max_samples) break;
    if (r >= next) {
      samples[2 * n] = r;
      samples[2 * n + 1] = __builtin_readcyclecounter();
      ++n;
      next = r + 100000;  // 1 ms at 100 MHz
    }
    __builtin_amdgcn_s_sleep(64);
  }
  *n_out = n;
}

template <typename F>
static void with_probe(const char* name, double ms_budget, F&& launch) {
  static hipStream_t sp = nullptr, sm = nullptr;
  static uint64_t* samples = nullptr;
  static int* n_out = nullptr;
  if (!sp) {
    hipStreamCreateWithFlags(&sp, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sm, hipStreamNonBlocking);
    hipHostMalloc(&samples, 2 * 4096 * sizeof(uint64_t));
    hipHostMalloc(&n_out, sizeof(int));
  }
  *n_out = 0;
  hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, sp, (uint64_t)(ms_budget * 1e5), samples, 4096, n_out);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipEventRecord(a, sm);
  launch(sm);
  hipEventRecord(b, sm);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const int n = *n_out;
  // frequency over consecutive samples; report min / mean / max inside the kernel's run time
  double fmin = 1e9, fmax = 0, fsum = 0;
  int cnt = 0;
  const int last = (int)(ms) < n - 1 ? (int)ms : n - 1;
  for (int i = 2; i < last; ++i) {
    const double f = (double)(samples[2 * i + 1] - samples[2 * i - 1]) / (double)(samples[2 * i] - samples[2 * i - 2]) * 100.0;
    fmin = f < fmin ? f : fmin;
    fmax = f > fmax ? f : fmax;
    fsum += f;
    ++cnt;
  }
  unsigned long long prof[4] = {0, 0, 0, 0}, zero[4] = {0, 0, 0, 0};
  hipMemcpyFromSymbol(prof, HIP_SYMBOL(aggf_gram_prof), sizeof prof);
  hipMemcpyToSymbol(HIP_SYMBOL(aggf_gram_prof), zero, sizeof zero);
  printf("%-38s %.1f ms | clock MHz mean %.0f min %.0f", name, ms, cnt ? fsum / cnt : 0.0, fmin);
  if (prof[3])
    printf(" | cycles per wave-stage: issue %.0f compute %.0f sync %.0f", (double)prof[0] / prof[3], (double)prof[1] / prof[3],
           (double)prof[2] / prof[3]);
  printf("\n");
}

int main() {
  using T = double;
  constexpr int KB = GramCfg<T>::KB;
  const int64_t rows = 200000;
  const int n_pad = 4096, ksplit = 32;
  const int nt1 = n_pad / TILE, n_tiles = nt1 * (nt1 + 1) / 2;
  const int64_t fps = round_up(ceil_div(rows, ksplit), KB);
  T *X, *slabs;
  int32_t* table;
  hipMalloc(&X, (size_t)rows * n_pad * 3 * sizeof(T));
  hipMalloc(&slabs, (size_t)ksplit * n_tiles * TILE * TILE * sizeof(T));
  hipMalloc(&table, 1 << 20);
  aggf_synth_normal(X, rows, n_pad, AGGF_F64, 1, 0, 0.0, 30.0, 0.0, nullptr);
  hipLaunchKernelGGL(build_tile_table_kernel, dim3(1), dim3(1), 0, 0, nt1, table);
  hipDeviceSynchronize();
  const size_t lds_reg = (size_t)2 * 2 * KB * ROW_STRIDE * sizeof(T);
  const size_t lds_dma = (size_t)3 * 2 * KB * ROW_STRIDE * sizeof(T);
  hipFuncSetAttribute((const void*)gram_tile_dma_kernel<T, 0, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma);
  hipFuncSetAttribute((const void*)gram_tile_dma_kernel<T, 1, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma);
  hipFuncSetAttribute((const void*)gram_tile_dma_kernel<T, 3, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma);
  hipFuncSetAttribute((const void*)gram_tile_dma_kernel<T, 4, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma);
  const double flops = (double)n_tiles * TILE * TILE * 2.0 * 3.0 * rows;
  with_probe("idle (no kernel)", 50, [&](hipStream_t) {});
  for (int rep = 0; rep < 1; ++rep) {
    with_probe("mfma only (ABL 3)", 400, [&](hipStream_t s) {
      for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((gram_tile_kernel<T, 3, false>), dim3(ksplit * n_tiles), dim3(GRAM_THREADS), lds_reg, s, X, rows, (int64_t)n_pad * 3, nt1, n_tiles, fps, slabs);
    });
    with_probe("mfma + lds reads + barrier (ABL 2)", 400, [&](hipStream_t s) {
      for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((gram_tile_kernel<T, 2, false>), dim3(ksplit * n_tiles), dim3(GRAM_THREADS), lds_reg, s, X, rows, (int64_t)n_pad * 3, nt1, n_tiles, fps, slabs);
    });
    with_probe("dma ring, no DMA in loop (ABL 1)", 400, [&](hipStream_t s) {
      for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((gram_tile_dma_kernel<T, 1, 3, 2>), dim3((unsigned)round_up((int64_t)ksplit * n_tiles, 512)), dim3(GRAM_THREADS), lds_dma, s, X, rows, (int64_t)n_pad * 3, nt1, n_tiles, ksplit, table, fps, slabs);
    });
    with_probe("dma ring, full (shipped)", 400, [&](hipStream_t s) {
      for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((gram_tile_dma_kernel<T, 0, 3, 2>), dim3((unsigned)round_up((int64_t)ksplit * n_tiles, 512)), dim3(GRAM_THREADS), lds_dma, s, X, rows, (int64_t)n_pad * 3, nt1, n_tiles, ksplit, table, fps, slabs);
    });
    with_probe("dma ring, DMAs all hit L2 (ABL 3)", 400, [&](hipStream_t s) {
      for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((gram_tile_dma_kernel<T, 3, 3, 2>), dim3((unsigned)round_up((int64_t)ksplit * n_tiles, 512)), dim3(GRAM_THREADS), lds_dma, s, X, rows, (int64_t)n_pad * 3, nt1, n_tiles, ksplit, table, fps, slabs);
    });
    with_probe("dma ring, DMAs miss L1, hit L2 (ABL 4)", 400, [&](hipStream_t s) {
      for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((gram_tile_dma_kernel<T, 4, 3, 2>), dim3((unsigned)round_up((int64_t)ksplit * n_tiles, 512)), dim3(GRAM_THREADS), lds_dma, s, X, rows, (int64_t)n_pad * 3, nt1, n_tiles, ksplit, table, fps, slabs);
    });
    {
      const int n_entries = pair_entry_count(nt1);
      const size_t lds_pair = (size_t)PAIR_NBUF * PAIR_SLOTS * KB * ROW_STRIDE * sizeof(T);
      PairEntry* ptable = reinterpret_cast<PairEntry*>(table);
      hipLaunchKernelGGL(build_pair_table_kernel, dim3(1), dim3(1), 0, 0, nt1, ptable);
      hipDeviceSynchronize();
      hipFuncSetAttribute((const void*)gram_pair_dma_kernel<T, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pair);
      hipFuncSetAttribute((const void*)gram_pair_dma_kernel<T, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pair);
      printf("pair entries %d (units %d)\n", n_entries, n_tiles);
      with_probe("PAIR tiles (8 waves), no DMA in loop", 400, [&](hipStream_t s) {
        for (int i = 0; i < 2; ++i)
          hipLaunchKernelGGL((gram_pair_dma_kernel<T, 1>), dim3((unsigned)round_up((int64_t)ksplit * n_entries, 256)), dim3(PAIR_THREADS), lds_pair, s, X, rows, (int64_t)n_pad * 3, nt1, n_entries, ksplit, ptable, fps, slabs);
      });
      with_probe("PAIR tiles (8 waves), full", 400, [&](hipStream_t s) {
        for (int i = 0; i < 2; ++i)
          hipLaunchKernelGGL((gram_pair_dma_kernel<T, 0>), dim3((unsigned)round_up((int64_t)ksplit * n_entries, 256)), dim3(PAIR_THREADS), lds_pair, s, X, rows, (int64_t)n_pad * 3, nt1, n_entries, ksplit, ptable, fps, slabs);
      });
      hipLaunchKernelGGL(build_tile_table_kernel, dim3(1), dim3(1), 0, 0, nt1, table);
      hipDeviceSynchronize();
    }
    hipFuncSetAttribute((const void*)gram_tile_dma_kernel<T, 0, 3, 2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma);
    hipFuncSetAttribute((const void*)gram_tile_dma_kernel<T, 1, 3, 2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma);
    with_probe("8 WAVES per tile (4/SIMD), no DMA in loop", 400, [&](hipStream_t s) {
      for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((gram_tile_dma_kernel<T, 1, 3, 2, 8>), dim3((unsigned)round_up((int64_t)ksplit * n_tiles, 512)), dim3(512), lds_dma, s, X, rows, (int64_t)n_pad * 3, nt1, n_tiles, ksplit, table, fps, slabs);
    });
    with_probe("8 WAVES per tile (4/SIMD), full", 400, [&](hipStream_t s) {
      for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((gram_tile_dma_kernel<T, 0, 3, 2, 8>), dim3((unsigned)round_up((int64_t)ksplit * n_tiles, 512)), dim3(512), lds_dma, s, X, rows, (int64_t)n_pad * 3, nt1, n_tiles, ksplit, table, fps, slabs);
    });
    hipFuncSetAttribute((const void*)gram_tile_dma_kernel<T, 0, 3, 2, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma);
    hipFuncSetAttribute((const void*)gram_tile_dma_kernel<T, 0, 3, 2, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma);
    with_probe("8 WAVES per tile, DMAs spread over MFMA groups", 400, [&](hipStream_t s) {
      for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((gram_tile_dma_kernel<T, 0, 3, 2, 8, true>), dim3((unsigned)round_up((int64_t)ksplit * n_tiles, 512)), dim3(512), lds_dma, s, X, rows, (int64_t)n_pad * 3, nt1, n_tiles, ksplit, table, fps, slabs);
    });
    with_probe("4 waves per tile, DMAs spread over MFMA groups", 400, [&](hipStream_t s) {
      for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((gram_tile_dma_kernel<T, 0, 3, 2, 4, true>), dim3((unsigned)round_up((int64_t)ksplit * n_tiles, 512)), dim3(256), lds_dma, s, X, rows, (int64_t)n_pad * 3, nt1, n_tiles, ksplit, table, fps, slabs);
    });
    hipFuncSetAttribute((const void*)gram_tile_dma_kernel<T, 3, 3, 2, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma);
    hipFuncSetAttribute((const void*)gram_tile_dma_kernel<T, 4, 3, 2, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma);
    with_probe("8 waves spread, DMAs all hit L1 (ABL 3)", 400, [&](hipStream_t s) {
      for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((gram_tile_dma_kernel<T, 3, 3, 2, 8, true>), dim3((unsigned)round_up((int64_t)ksplit * n_tiles, 512)), dim3(512), lds_dma, s, X, rows, (int64_t)n_pad * 3, nt1, n_tiles, ksplit, table, fps, slabs);
    });
    with_probe("8 waves spread, DMAs miss L1, hit L2 (ABL 4)", 400, [&](hipStream_t s) {
      for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((gram_tile_dma_kernel<T, 4, 3, 2, 8, true>), dim3((unsigned)round_up((int64_t)ksplit * n_tiles, 512)), dim3(512), lds_dma, s, X, rows, (int64_t)n_pad * 3, nt1, n_tiles, ksplit, table, fps, slabs);
    });
    with_probe("register-staged, full", 400, [&](hipStream_t s) {
      for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((gram_tile_kernel<T, 0, false>), dim3(ksplit * n_tiles), dim3(GRAM_THREADS), lds_reg, s, X, rows, (int64_t)n_pad * 3, nt1, n_tiles, fps, slabs);
    });
  }
  printf("flops per launch (executed) %.3e; 2 launches per line\n", flops);
  // row stride: 4096 atoms x 24 B = 96 KiB is a power-of-two multiple of the cache line; does the
  // L2 alias?  Same kernel on the first 4096 atoms of rows that are 16 / 5 / 1 atoms longer.
  for (int extra : {16, 80, 1040}) {
    T* Xp;
    const int64_t ldp = (int64_t)(n_pad + extra) * 3;
    if (hipMalloc(&Xp, (size_t)rows * ldp * sizeof(T)) != hipSuccess) break;
    aggf_synth_normal(Xp, rows, n_pad + extra, AGGF_F64, 1, 0, 0.0, 30.0, 0.0, nullptr);
    hipDeviceSynchronize();
    char name[64];
    snprintf(name, sizeof name, "dma full, row stride +%d atoms", extra);
    with_probe(name, 400, [&](hipStream_t s) {
      for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((gram_tile_dma_kernel<T, 0, 3, 2>), dim3((unsigned)round_up((int64_t)ksplit * n_tiles, 512)), dim3(GRAM_THREADS), lds_dma, s, Xp, rows, ldp, nt1, n_tiles, ksplit, table, fps, slabs);
    });
    hipFree(Xp);
  }
  // tile length sweep: shorter frame ranges per workgroup = less time for the workgroups that share
  // panels in an XCD's L2 to drift apart
  hipFree(slabs);
  for (int ks : {64}) {
    const int64_t f2 = round_up(ceil_div(rows, ks), KB);
    hipMalloc(&slabs, (size_t)ks * n_tiles * TILE * TILE * sizeof(T));
    char name[64];
    snprintf(name, sizeof name, "dma full, ksplit %d (%ld stages)", ks, (long)(f2 / KB));
    with_probe(name, 400, [&](hipStream_t s) {
      for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((gram_tile_dma_kernel<T, 0, 3, 2>), dim3((unsigned)round_up((int64_t)ks * n_tiles, 512)), dim3(GRAM_THREADS), lds_dma, s, X, rows, (int64_t)n_pad * 3, nt1, n_tiles, ks, table, f2, slabs);
    });
    hipFree(slabs);
  }
  return 0;
}
