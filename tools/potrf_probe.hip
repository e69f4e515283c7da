// In-kernel phase timing of K2's 64 x 64 diagonal-block kernel (potrf_diag_mfma_kernel):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAGGF_POTRF_PROF -I aggforce_amd/csrc tools/potrf_probe.hip \
//         aggforce_amd/csrc/aggf_util.hip -o tools/potrf_probe && tools/potrf_probe
#include "../aggforce_amd/csrc/aggf_solve.hip"

#include <vector>

int main() {
  const int n = 4096;
  std::vector<double> h((size_t)n * n, 0.0);
  for (int i = 0; i < 64; ++i)
    for (int j = 0; j <= i; ++j) h[(size_t)i * n + j] = (i == j ? 80.0 : 0.0) + 1.0 / (1.0 + i + j);
  double *A, *L, *info;
  hipMalloc(&A, h.size() * 8);
  hipMalloc(&L, 64 * 64 * 8);
  hipMalloc(&info, 32);
  hipMemset(info, 0, 32);
  hipFuncSetAttribute((const void*)aggf::potrf_diag_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, aggf::POTRF3_LDS);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipEventRecord(e0);
    hipLaunchKernelGGL(aggf::potrf_diag_mfma_kernel, dim3(1), dim3(256), aggf::POTRF3_LDS, 0, A, (int64_t)n, L, info, 0,
                       (int64_t)0, (int64_t)0, (int64_t)4);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long p[8];
    hipMemcpyFromSymbol(p, HIP_SYMBOL(aggf::aggf_potrf_prof), sizeof(p));
    printf("rep %d: event %.1f us | cycles: load %llu, factor + inverse levels %llu, store L %llu, store X %llu, total %llu\n", rep,
           ms * 1e3, p[1] - p[0], p[2] - p[1], p[3] - p[2], p[4] - p[3], p[4] - p[0]);
    unsigned long long ph[6];
    hipMemcpyFromSymbol(ph, HIP_SYMBOL(aggf::aggf_potrf_phase), sizeof(ph));
    printf("        inside the factor: sub-blocks %llu, panels %llu, trailing %llu, inverse levels %llu\n", ph[0], ph[1], ph[2], ph[3]);
  }
  return 0;
}
