// Micro-benchmark (GPU box): sustained MFMA rate of the two instructions the Gram/apply kernels
// use, v_mfma_f64_16x16x4_f64 and v_mfma_f32_16x16x4_f32, one or two waves per SIMD, operands in
// registers.  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double __attribute__((ext_vector_type(4))) f64x4;
typedef float __attribute__((ext_vector_type(4))) f32x4;

template <int NACC>
__global__ __launch_bounds__(256) void k_f64(double* out, int iters) {
  f64x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_f32(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
static double time_ms(F launch) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}
int main() {
  void* buf; hipMalloc(&buf, 1 << 26);
  const int iters = 20000;
  for (int bpc = 1; bpc <= 2; ++bpc) {
    const int grid = 256 * bpc;
    double ms = time_ms([&] { hipLaunchKernelGGL(k_f64<16>, dim3(grid), dim3(256), 0, 0, (double*)buf, iters); });
    double fl = (double)grid * 4 * iters * 16 * 2048.0;
    printf("f64 16x16x4, 16 acc, %d wave(s)/SIMD: %.1f TFLOP/s (%.3f ms)\n", bpc, fl / ms / 1e9, ms);
    ms = time_ms([&] { hipLaunchKernelGGL(k_f32<16>, dim3(grid), dim3(256), 0, 0, (float*)buf, iters); });
    printf("f32 16x16x4, 16 acc, %d wave(s)/SIMD: %.1f TFLOP/s (%.3f ms)\n", bpc, fl / ms / 1e9, ms);
  }
  double ms = time_ms([&] { hipLaunchKernelGGL(k_f64<1>, dim3(256), dim3(256), 0, 0, (double*)buf, iters * 16); });
  printf("f64 16x16x4, 1 acc (dependent chain), 1 wave/SIMD: %.1f TFLOP/s\n", 256.0 * 4 * iters * 16 * 2048.0 / ms / 1e9);
  return 0;
}
