// The two streaming kernels of the small-system regime (CLN025: 175 atoms, 97 reduced variables, 10 beads, 4e6
// frames, float64) with one co-limiting phase removed at a time -- what bounds them when HBM, MFMA and LDS are of
// equal size.  One binary per variant:
//   for a in 0 1 2 3; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAGGF_SMALL_ABL=$a tools/small_ablate.hip \
//       aggforce_amd/csrc/aggf_util.hip -o /tmp/small_ablate_$a -ldl && /tmp/small_ablate_$a; done
// Gram (gram_small_kernel): 0 complete, 1 no MFMA phase, 2 no global loads after the first stage, 3 no group sums.
// Apply (apply_small_kernel): timed beside it, always complete (its isolated rate; inside project_forces the coordinate
// gather shares the HBM with it).
#include "../aggforce_amd/csrc/aggf_gram.hip"
#include "../aggforce_amd/csrc/aggf_apply.hip"

#include <vector>

int main() {
  const int64_t T = 4000000;
  const int32_t N = 175, n_red = 97, n_cg = 10;
  std::vector<int32_t> ptr(n_red + 1), atoms(N);
  int a = 0;
  for (int g = 0; g < n_red; ++g) {
    ptr[g] = a;
    const int size = g < 38 ? 1 : (a + 3 * (n_red - g) <= N ? 3 : 2);
    for (int j = 0; j < size && a < N; ++j) atoms[a++] = a;
  }
  while (a < N) { atoms[a] = a; ++a; }
  ptr[n_red] = N;
  double *F, *M, *out, *G;
  hipMalloc(&F, (size_t)T * N * 3 * 8);
  hipMalloc(&M, (size_t)n_cg * (N / 3 + 1) * 3 * 8);
  hipMalloc(&out, (size_t)T * n_cg * 3 * 8);
  hipMalloc(&G, (size_t)n_red * n_red * 8);
  aggf_synth_normal(F, T, N, AGGF_F64, 1, 0, 0.0, 30.0, 0.0, nullptr);
  aggf_synth_normal(M, n_cg, N / 3 + 1, AGGF_F64, 2, 0, 0.0, 1.0, 0.0, nullptr);
  int32_t *dptr, *datoms;
  hipMalloc(&dptr, ptr.size() * 4);
  hipMalloc(&datoms, atoms.size() * 4);
  hipMemcpy(dptr, ptr.data(), ptr.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(datoms, atoms.data(), atoms.size() * 4, hipMemcpyHostToDevice);
  const size_t need = aggf_gram_workspace_bytes(T, N, n_red, AGGF_F64, AGGF_F64, 1) +
                      aggf_linearmap_apply_workspace_bytes(T, N, n_cg);
  void* ws;
  hipMalloc(&ws, need);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const double gb = (double)T * N * 3 * 8 / 1e9;
  for (int which = 0; which < 2; ++which) {
    float best = 1e30f;
    int rc = 0;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      if (which == 0)
        rc = aggf_gram(F, T, N, AGGF_F64, AGGF_F64, dptr, datoms, n_red, G, 0, ws, need, nullptr);
      else
        rc = aggf_linearmap_apply(F, T, N, AGGF_F64, M, n_cg, AGGF_F64, AGGF_NAN_PROPAGATE, 0.0, out, nullptr, nullptr, ws,
                                  need, nullptr);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    printf("{\"kernel\": \"%s\", \"ablation\": %d, \"rc\": %d, \"ms\": %.3f, \"GB\": %.2f, \"GB_per_s\": %.0f, \"frac_of_8TBps\": %.3f}\n",
           which == 0 ? "aggf_gram (gram_small_kernel + gram_reduce_small_kernel), CLN025 x 4e6 frames" : "aggf_linearmap_apply (apply_small_kernel, complete), CLN025 x 4e6 frames",
           AGGF_SMALL_ABL, rc, best, gb, gb / best * 1e3, gb / best * 1e3 / 8000.0);
  }
  return 0;
}
