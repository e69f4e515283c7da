// Ablation bench of the Gram tile kernel (GPU box): where do the cycles go?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gram_ablate.hip aggforce_amd/csrc/aggf_util.hip -o /tmp/gram_ablate
#include "../aggforce_amd/csrc/aggf_gram.hip"

using namespace aggf;

template <typename T, int ABL, bool STAG = false>
static double run(const T* X, int64_t rows, int n_pad, int ksplit, T* slabs, size_t extra_lds = 0) {
  constexpr int KB = GramCfg<T>::KB;
  const int nt1 = n_pad / TILE, n_tiles = nt1 * (nt1 + 1) / 2;
  int64_t fps = round_up(ceil_div(rows, ksplit), KB);
  const size_t lds = (size_t)2 * 2 * KB * ROW_STRIDE * sizeof(T) + extra_lds;
  if (extra_lds) hipFuncSetAttribute((const void*)gram_tile_kernel<T, ABL, STAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((gram_tile_kernel<T, ABL, STAG>), dim3(ksplit * n_tiles), dim3(GRAM_THREADS), lds, 0, X, rows,
                       (int64_t)n_pad * 3, nt1, n_tiles, fps, slabs);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best;
}

static int32_t* g_table = nullptr;

template <typename T, int ABL, int NBUF = 3, int WPS = 2>
static double run_dma(const T* X, int64_t rows, int n_pad, int ksplit, T* slabs) {
  constexpr int KB = GramCfg<T>::KB;
  const int nt1 = n_pad / TILE, n_tiles = nt1 * (nt1 + 1) / 2;
  int64_t fps = round_up(ceil_div(rows, ksplit), KB);
  const size_t lds = (size_t)NBUF * 2 * KB * ROW_STRIDE * sizeof(T);
  if (!g_table) hipMalloc(&g_table, 1 << 20);
  hipLaunchKernelGGL(build_tile_table_kernel, dim3(1), dim3(1), 0, 0, nt1, g_table);
  hipFuncSetAttribute((const void*)gram_tile_dma_kernel<T, ABL, NBUF, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((gram_tile_dma_kernel<T, ABL, NBUF, WPS>), dim3((unsigned)round_up((int64_t)ksplit * n_tiles, 512)), dim3(GRAM_THREADS), lds, 0, X, rows,
                       (int64_t)n_pad * 3, nt1, n_tiles, ksplit, g_table, fps, slabs);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best;
}

template <typename T>
static void sweep(int64_t rows, int n_pad, double peak) {
  const int nt1 = n_pad / TILE, n_tiles = nt1 * (nt1 + 1) / 2;
  T* X;
  hipMalloc(&X, (size_t)rows * n_pad * 3 * sizeof(T));
  aggf_synth_normal(X, rows, n_pad, sizeof(T) == 8 ? AGGF_F64 : AGGF_F32, 1, 0, 0.0, 30.0, 0.0, nullptr);
  const double flops_exec = (double)n_tiles * TILE * TILE * 2.0 * 3.0 * rows;
  for (int ksplit : {32}) {
    T* slabs;
    hipMalloc(&slabs, (size_t)ksplit * n_tiles * TILE * TILE * sizeof(T));
    double t0 = run<T, 0>(X, rows, n_pad, ksplit, slabs);
    double t1 = run<T, 1>(X, rows, n_pad, ksplit, slabs);
    double t2 = run<T, 2>(X, rows, n_pad, ksplit, slabs);
    double t3 = run<T, 3>(X, rows, n_pad, ksplit, slabs);
    double t4 = run<T, 4>(X, rows, n_pad, ksplit, slabs);
    double t5 = run<T, 5>(X, rows, n_pad, ksplit, slabs);
    double t1b = run<T, 1>(X, rows, n_pad, ksplit, slabs, 40 * 1024);
    double t0b = run<T, 0>(X, rows, n_pad, ksplit, slabs, 40 * 1024);
    printf("   no-gload: refill w/o barrier %.2f (%.1f TF) | barrier w/o refill %.2f (%.1f TF) || ONE block/CU: no-gload %.2f (%.1f TF) full %.2f (%.1f TF)\n",
           t4, flops_exec / t4 / 1e9, t5, flops_exec / t5 / 1e9, t1b, flops_exec / t1b / 1e9, t0b, flops_exec / t0b / 1e9);
    double d0 = run_dma<T, 0>(X, rows, n_pad, ksplit, slabs);
    double d1 = run_dma<T, 1>(X, rows, n_pad, ksplit, slabs);
    double d2 = run_dma<T, 2>(X, rows, n_pad, ksplit, slabs);
    printf("   LDS-DMA ring: full %.2f ms (%.1f TF exec, %.0f%%) | no DMA in loop %.2f (%.1f TF) | + no barrier %.2f (%.1f TF)\n",
           d0, flops_exec / d0 / 1e9, 100 * flops_exec / d0 / 1e9 / peak, d1, flops_exec / d1 / 1e9, d2, flops_exec / d2 / 1e9);
    double e23 = run_dma<T, 0, 2, 3>(X, rows, n_pad, ksplit, slabs);
    double e22 = run_dma<T, 0, 2, 2>(X, rows, n_pad, ksplit, slabs);
    printf("   LDS-DMA 2-stage ring: 3 waves/SIMD %.2f ms (%.1f TF) | 2 waves/SIMD %.2f ms (%.1f TF)\n", e23,
           flops_exec / e23 / 1e9, e22, flops_exec / e22 / 1e9);
    double ts = run<T, 0, true>(X, rows, n_pad, ksplit, slabs);
    double ts1 = run<T, 1, true>(X, rows, n_pad, ksplit, slabs);
    printf("   staggered: full %.2f ms (%.1f TF exec, %.0f%%) | no-gload %.2f (%.1f TF)\n", ts, flops_exec / ts / 1e9,
           100 * flops_exec / ts / 1e9 / peak, ts1, flops_exec / ts1 / 1e9);
    printf("%s N=%d T=%ld ksplit=%d blocks=%d: full %.2f ms (%.1f TF exec, %.0f%%) | no-gload %.2f (%.1f TF) | no-refill/barrier %.2f (%.1f TF) | mfma-only %.2f (%.1f TF)\n",
           sizeof(T) == 8 ? "f64" : "f32", n_pad, (long)rows, ksplit, ksplit * n_tiles, t0, flops_exec / t0 / 1e9,
           100 * flops_exec / t0 / 1e9 / peak, t1, flops_exec / t1 / 1e9, t2, flops_exec / t2 / 1e9, t3,
           flops_exec / t3 / 1e9);
    hipFree(slabs);
  }
  hipFree(X);
}

int main(int argc, char** argv) {
  sweep<double>(200000, 4096, 78.6);
  if (argc > 1) return 0;  // "quick" mode for PMC passes
  sweep<float>(400000, 4096, 157.3);
  sweep<float>(100000, 1024, 157.3);
  return 0;
}
