// Phase timers of the K3 apply kernel (per wave and stage): load issue / MFMA / LDS refill / barrier.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAGGF_APPLY_PROF tools/apply_probe.hip aggforce_amd/csrc/aggf_util.hip -o tools/apply_probe
#define AGGF_APPLY_PROF 1
#include "../aggforce_amd/csrc/aggf_apply.hip"
using namespace aggf;
// args: [frames atoms sites [f32]]  (f32: float32 frames, float64 map and result)
int main(int argc, char** argv) {
  const int64_t T = argc > 1 ? atoll(argv[1]) : 200000;
  const int N = argc > 2 ? atoi(argv[2]) : 4096, n_cg = argc > 3 ? atoi(argv[3]) : 256;
  const bool f32 = argc > 4 && argv[4][0] == 'f' && argv[4][1] == '3';
  const int pdt = f32 ? AGGF_F32 : AGGF_F64;
  double *P, *M, *out;
  hipMalloc(&P, (size_t)T * N * 3 * 8);
  hipMalloc(&M, (size_t)n_cg * (N / 3 + 1) * 3 * 8);  // the synthetic fill below writes whole (atom, xyz) triples
  hipMalloc(&out, (size_t)T * n_cg * 3 * 8);
  aggf_synth_normal(P, T, N, pdt, 1, 0, 0.0, 30.0, 0.0, nullptr);
  aggf_synth_normal(M, n_cg, N / 3 + 1, AGGF_F64, 2, 0, 0.0, 1.0, 0.0, nullptr);
  size_t need = aggf_linearmap_apply_workspace_bytes(T, N, n_cg) + 256;
  void* ws;
  hipMalloc(&ws, need);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    unsigned long long zero[5] = {0, 0, 0, 0, 0};
    hipMemcpyToSymbol(HIP_SYMBOL(aggf_apply_prof), zero, sizeof zero);
    hipEventRecord(a);
    int rc = aggf_linearmap_apply(P, T, N, pdt, M, n_cg, AGGF_F64, AGGF_NAN_PROPAGATE, 0.0, out, nullptr, nullptr, ws, need, nullptr);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    unsigned long long pf[5];
    hipMemcpyFromSymbol(pf, HIP_SYMBOL(aggf_apply_prof), sizeof pf);
    printf("rc %d apply %.2f ms (%.1f TF) | cycles per wave-stage: load issue %.0f, mfma %.0f, refill %.0f, barrier %.0f\n", rc, ms,
           2.0 * 3 * T * N * n_cg / ms / 1e9, (double)pf[0] / pf[4], (double)pf[1] / pf[4], (double)pf[2] / pf[4], (double)pf[3] / pf[4]);
  }
  return 0;
}
