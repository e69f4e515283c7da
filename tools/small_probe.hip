// Phase timers of the small-system Gram kernel (gram_small_kernel) at CLN025 size:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/small_probe.hip aggforce_amd/csrc/aggf_util.hip -o /tmp/small_probe -ldl && /tmp/small_probe
#define AGGF_SMALL_PROF 1
#include "../aggforce_amd/csrc/aggf_gram.hip"

#include <vector>

// usage: small_probe            CLN025-like (175 atoms, 97 reduced columns)
//        small_probe <atoms>    unconstrained system of that many atoms, 16.8 GB of frames
int main(int argc, char** argv) {
  //        small_probe <atoms> pairs   bond pairs {3i, 3i+1} on that many atoms (n_red = N - N / 3)
  const bool pairs = argc > 2 && argv[2][0] == 'p';
  const bool plain = argc > 1 && !pairs;
  const int32_t N = argc > 1 ? atoi(argv[1]) : 175, n_red = plain ? N : (pairs ? N - N / 3 : 97);
  const int64_t T = argc > 1 ? (int64_t)(16.8e9 / (24.0 * N)) : 4000000;
  // CLN025-like groups: 59 groups (38 anchors alone ... here: first 38 atoms alone, then groups of 2-3)
  std::vector<int32_t> ptr(n_red + 1), atoms(N);
  int a = 0;
  if (pairs) {
    int g = 0;
    for (int i = 0; i < N; ++g) {
      ptr[g] = a;
      const bool two = i % 3 == 0 && i / 3 < N / 3 && i + 1 < N;
      atoms[a++] = i++;
      if (two) atoms[a++] = i++;
    }
    if (g != n_red) { printf("pair layout: %d groups for n_red %d\n", g, n_red); return 1; }
  } else {
    for (int g = 0; g < n_red; ++g) {
      ptr[g] = a;
      const int size = g < 38 ? 1 : (a + 3 * (n_red - g) <= N ? 3 : 2);
      for (int j = 0; j < size && a < N; ++j) atoms[a++] = a;
    }
    while (a < N) { atoms[a] = a; ++a; }
  }
  ptr[n_red] = N;
  double* F;
  hipMalloc(&F, (size_t)T * N * 3 * 8);
  aggf_synth_normal(F, T, N, AGGF_F64, 1, 0, 0.0, 30.0, 0.0, nullptr);
  int32_t *dptr, *datoms;
  hipMalloc(&dptr, ptr.size() * 4);
  hipMalloc(&datoms, atoms.size() * 4);
  hipMemcpy(dptr, ptr.data(), ptr.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(datoms, atoms.data(), atoms.size() * 4, hipMemcpyHostToDevice);
  double* G;
  hipMalloc(&G, (size_t)n_red * n_red * 8);
  const size_t need = aggf_gram_workspace_bytes(T, N, n_red, AGGF_F64, AGGF_F64, 1);
  void* ws;
  hipMalloc(&ws, need);
  for (int rep = 0; rep < 2; ++rep) {
    unsigned long long zero[9] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(aggf::aggf_small_prof), zero, sizeof(zero));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    int rc = aggf_gram(F, T, N, AGGF_F64, AGGF_F64, plain ? nullptr : dptr, plain ? nullptr : datoms, n_red, G, 0, ws, need, nullptr);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long pf[9];
    hipMemcpyFromSymbol(pf, HIP_SYMBOL(aggf::aggf_small_prof), sizeof(pf));
    const double waves = (double)pf[8], stages = (double)pf[7] / waves;
    printf("%d atoms, %d columns: rc %d  %.3f ms  waves %.0f  stages/wave %.1f\n", N, n_red, rc, ms, waves, stages);
    const char* names[7] = {"barrier1", "park+loadwait", "barrier2", "group sums", "barrier3", "MFMA phase", "fetch issue"};
    double tot = 0;
    for (int i = 0; i < 7; ++i) tot += (double)pf[i];
    for (int i = 0; i < 7; ++i)
      printf("  %-26s %8.0f cycles per stage per wave (%4.1f %%)\n", names[i], (double)pf[i] / pf[7], 100.0 * pf[i] / tot);
    printf("  total %.0f cycles per stage per wave\n", tot / pf[7]);
  }
  return 0;
}
