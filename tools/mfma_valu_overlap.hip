// Do fp64 MFMA and fp64 VALU FMAs overlap on gfx950?  (Could K1 give part of its flops to the VALU?)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_valu_overlap.hip -o tools/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NV, bool MF>
__global__ __launch_bounds__(256, 2) void k(double* out, int iters, double seed) {
  f64x4 acc[16];
  double v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    acc[i] = f64x4{seed, seed, seed, seed};
    v[i] = seed + i + threadIdx.x;
  }
  double a = seed + threadIdx.x, b = seed * 0.5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MF) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV / 16; ++j) {
        const int r = (i * (NV / 16) + j) % 16;
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(v[r]) : "v"(a), "v"(b));
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
  if (s == 12345.678) out[threadIdx.x] = s;
}

template <int NV, bool MF>
static void run(const char* name, double* out) {
  const int iters = 20000;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k<NV, MF>), dim3(512), dim3(256), 0, 0, out, iters, 1.0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  const double mf = MF ? 512.0 * 4 * iters * 16 * 2048.0 : 0.0;      // flops in MFMAs
  const double vf = 512.0 * 4 * iters * (double)NV * 64 * 2.0;       // flops in VALU FMAs
  printf("%-40s %8.2f ms | MFMA %6.1f TF + VALU %6.1f TF = %6.1f TF\n", name, best, mf / best / 1e9, vf / best / 1e9,
         (mf + vf) / best / 1e9);
}

int main() {
  double* out;
  hipMalloc(&out, 4096);
  run<0, true>("16 MFMA", out);
  run<16, false>("16 v_fma_f64", out);
  run<64, false>("64 v_fma_f64", out);
  run<16, true>("16 MFMA + 16 v_fma_f64", out);
  run<32, true>("16 MFMA + 32 v_fma_f64", out);
  run<64, true>("16 MFMA + 64 v_fma_f64", out);
  run<128, true>("16 MFMA + 128 v_fma_f64", out);
  return 0;
}
