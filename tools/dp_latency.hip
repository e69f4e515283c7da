// fp64 VALU timing on one wave per SIMD (the regime of K2's diagonal-block kernel): cycles per DEPENDENT v_fma_f64,
// per independent one (4 chains), per v_rsq_f64, and the relative error of v_rsq_f64 (raw, after one third-order step,
// after the second step).   hipcc --offload-arch=gfx950 -O3 tools/dp_latency.hip -o tools/dp_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

#define FMA64(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c))
#define FMA32(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(bf), "v"(cf))
#define MUL64(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b))
__global__ void lat_kernel(double* out, unsigned long long* cyc, double seed) {
  double a = seed + threadIdx.x * 1e-9, b = 1.0000001 + seed * 1e-12, c = 1e-9 * seed;
  float bf = (float)b, cf = (float)c;
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 256; ++i) FMA64(a);
  unsigned long long t1 = __builtin_readcyclecounter();
  double p = a, q = a + 1, r = a + 2, s = a + 3;
#pragma unroll
  for (int i = 0; i < 64; ++i) { FMA64(p); FMA64(q); FMA64(r); FMA64(s); }
  unsigned long long t2 = __builtin_readcyclecounter();
  double u = p + q + r + s;
#pragma unroll
  for (int i = 0; i < 128; ++i) { asm volatile("v_rsq_f64 %0, %0" : "+v"(u)); asm volatile("s_nop 0"); }
  unsigned long long t3 = __builtin_readcyclecounter();
  float f = (float)u;
#pragma unroll
  for (int i = 0; i < 256; ++i) FMA32(f);
  unsigned long long t4 = __builtin_readcyclecounter();
  double m = u;
#pragma unroll
  for (int i = 0; i < 256; ++i) MUL64(m);
  unsigned long long t5 = __builtin_readcyclecounter();
  out[threadIdx.x] = u + f + m;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; }
}

__global__ void acc_kernel(const double* x, double* e0, double* e1, double* e2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = x[i];
  double rs = __builtin_amdgcn_rsq(d);
  e0[i] = rs;
  double e = fma(-d * rs, rs, 1.0);
  rs = fma(rs * e, fma(e, 0.375, 0.5), rs);
  e1[i] = rs;
  e = fma(-d * rs, rs, 1.0);
  e2[i] = fma(rs * 0.5, e, rs);
}

int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 64);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(lat_kernel, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
    hipDeviceSynchronize();
    unsigned long long h[5];
    hipMemcpy(h, cyc, 40, hipMemcpyDeviceToHost);
    printf("cycles per op: dependent v_fma_f64 %.1f | 4 independent chains %.1f per fma | dependent v_rsq_f64 %.1f | dependent v_fma_f32 %.1f | dependent v_mul_f64 %.1f\n",
           h[0] / 256.0, h[1] / 256.0, h[2] / 128.0, h[3] / 256.0, h[4] / 256.0);
  }
  const int n = 1 << 20;
  std::vector<double> x(n), r0(n), r1(n), r2(n);
  for (int i = 0; i < n; ++i) x[i] = std::ldexp(1.0 + (i + 0.5) / n * 3.0, (i % 41) - 20);
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(acc_kernel, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
  hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost);
  double m0 = 0, m1 = 0, m2 = 0;
  for (int i = 0; i < n; ++i) {
    const long double t = 1.0L / sqrtl((long double)x[i]);
    m0 = fmax(m0, (double)fabsl((r0[i] - t) / t));
    m1 = fmax(m1, (double)fabsl((r1[i] - t) / t));
    m2 = fmax(m2, (double)fabsl((r2[i] - t) / t));
  }
  printf("max relative error of 1/sqrt: raw v_rsq_f64 %.3g | after the third-order step %.3g | after the second step %.3g (2^-53 = %.3g)\n",
         m0, m1, m2, std::ldexp(1.0, -53));
  return 0;
}
