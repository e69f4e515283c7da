// Phase timers of the producer-consumer streaming Gram kernel (gram_ws_kernel, aggf_gram_ws.h):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ws_probe.hip aggforce_amd/csrc/aggf_util.hip -o /tmp/ws_probe -ldl
//   AGGF_GRAM_WS=1 /tmp/ws_probe <atoms> [pairs] [f32]
#define AGGF_WS_PROF 1
#include "../aggforce_amd/csrc/aggf_gram.hip"

#include <vector>

int main(int argc, char** argv) {
  const int32_t N = argc > 1 ? atoi(argv[1]) : 175;
  bool pairs = false, f32 = false;
  for (int i = 2; i < argc; ++i) {
    if (!strcmp(argv[i], "pairs")) pairs = true;
    if (!strcmp(argv[i], "f32")) f32 = true;
  }
  const int es = f32 ? 4 : 8;
  const int64_t T = (int64_t)(12e9 / (3.0 * es * N)) / 64 * 64;
  const int32_t n_groups = pairs ? N / 3 : 0, n_red = N - n_groups;
  // make_bond_constraint_matrix's column order: unconstrained atoms (and anchors) first ... here simply: singles, then pairs
  std::vector<int32_t> ptr(n_red + 1), atoms(N);
  int a = 0, g = 0;
  for (int i = 0; i < N; ++i) {
    const bool in_pair = pairs && i / 3 < n_groups && i % 3 < 2;
    if (!in_pair) { ptr[g++] = a; atoms[a++] = i; }
  }
  for (int p = 0; p < n_groups; ++p) { ptr[g++] = a; atoms[a++] = 3 * p; atoms[a++] = 3 * p + 1; }
  ptr[n_red] = N;
  void* F;
  hipMalloc(&F, (size_t)T * N * 3 * es);
  aggf_synth_normal(F, T, N, f32 ? AGGF_F32 : AGGF_F64, 1, 0, 0.0, 30.0, 0.0, nullptr);
  int32_t *dptr, *datoms;
  hipMalloc(&dptr, ptr.size() * 4);
  hipMalloc(&datoms, atoms.size() * 4);
  hipMemcpy(dptr, ptr.data(), ptr.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(datoms, atoms.data(), atoms.size() * 4, hipMemcpyHostToDevice);
  double* G;
  hipMalloc(&G, (size_t)n_red * n_red * 8);
  const int dt = f32 ? AGGF_F32 : AGGF_F64;
  const size_t need = aggf_gram_workspace_bytes(T, N, n_red, dt, AGGF_F64, pairs ? 1 : 0);
  void* ws;
  hipMalloc(&ws, need);
  for (int rep = 0; rep < 2; ++rep) {
    unsigned long long zero[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(aggf::aggf_ws_prof), zero, sizeof(zero));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    int rc = aggf_gram(F, T, N, dt, AGGF_F64, pairs ? dptr : nullptr, pairs ? datoms : nullptr, n_red, G, 0, ws, need, nullptr);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long pf[8];
    hipMemcpyFromSymbol(pf, HIP_SYMBOL(aggf::aggf_ws_prof), sizeof(pf));
    const double flop = 3.0 * T * n_red * (n_red + 1.0);
    printf("{\"atoms\": %d, \"n_red\": %d, \"pairs\": %s, \"in\": \"%s\", \"frames\": %ld, \"rc\": %d, \"ms\": %.3f, \"frac_mfma_fp64\": %.3f, \"TBps\": %.2f",
           N, n_red, pairs ? "true" : "false", f32 ? "f32" : "f64", (long)T, rc, ms, flop / (ms * 1e-3) / 78.6e12,
           (double)T * N * 3 * es / (ms * 1e-3) / 1e12);
    if (pf[6] && pf[7])
      printf(", \"producer_cycles_per_stage\": {\"load_issue\": %.0f, \"sums\": %.0f, \"wait_park\": %.0f, \"barrier\": %.0f}, "
             "\"consumer_cycles_per_stage\": {\"mfma\": %.0f, \"barrier\": %.0f}",
             (double)pf[0] / pf[6], (double)pf[1] / pf[6], (double)pf[2] / pf[6], (double)pf[3] / pf[6], (double)pf[4] / pf[7],
             (double)pf[5] / pf[7]);
    printf("}\n");
  }
  return 0;
}
