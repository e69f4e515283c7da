// Does the cache policy of a load change how much of a 128-byte line a 24-byte pick costs?  Picks of 3 doubles every
// 384 bytes (the c3 coordinate gather: one site in 16) from a 12.6 GB array, one pick element per thread and frame,
// with the load spelled plain / nt / sc0 / sc1 / sc0 sc1 / sc0 sc1 nt.  HBM-bound: the time IS the traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o tools/gather_probe && tools/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__device__ __forceinline__ double ld(const double* p) {
  double v;
  if (MODE == 0) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  if (MODE == 1) asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  if (MODE == 2) asm volatile("global_load_dwordx2 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
  if (MODE == 3) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  if (MODE == 4) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  if (MODE == 5) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int MODE>
__global__ __launch_bounds__(256) void pick_kernel(const double* __restrict__ P, int64_t T, int row_in, int n_pick, int stride,
                                                   double* __restrict__ out) {
  const int e = blockIdx.y * 256 + threadIdx.x;
  const int row_out = n_pick * 3;
  if (e >= row_out) return;
  const int c = e / 3;
  const int64_t off = (int64_t)c * stride + (e - 3 * c);
  for (int64_t t0 = (int64_t)blockIdx.x * 8; t0 < T; t0 += (int64_t)gridDim.x * 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = t0 + u < T ? ld<MODE>(P + (t0 + u) * row_in + off) : 0.0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (t0 + u < T) __builtin_nontemporal_store(v[u], out + (t0 + u) * row_out + e);
  }
}

template <int MODE>
static void run(const char* name, const double* P, int64_t T, int row_in, int n_pick, int stride, double* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(pick_kernel<MODE>, dim3(1024 / ((n_pick * 3 + 255) / 256), (n_pick * 3 + 255) / 256), dim3(256), 0, 0, P, T, row_in,
                       n_pick, stride, out);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  const double lines = (double)T * n_pick * 128.0;
  printf("%-14s %.3f ms   (%.0f GB/s if every pick costs its 128-byte line, %.0f GB/s useful)\n", name, best, lines / best / 1e6,
         (double)T * n_pick * 48.0 / best / 1e6);
}

int main() {
  const int64_t T = 128000;
  const int N = 4096, n_pick = 256, row_in = N * 3, stride = (N / n_pick) * 3;
  double *P, *out;
  hipMalloc(&P, (size_t)T * row_in * 8);
  hipMalloc(&out, (size_t)T * n_pick * 3 * 8);
  hipMemset(P, 0, (size_t)T * row_in * 8);
  run<0>("plain", P, T, row_in, n_pick, stride, out);
  run<1>("nt", P, T, row_in, n_pick, stride, out);
  run<2>("sc0", P, T, row_in, n_pick, stride, out);
  run<3>("sc1", P, T, row_in, n_pick, stride, out);
  run<4>("sc0 sc1", P, T, row_in, n_pick, stride, out);
  run<5>("sc0 sc1 nt", P, T, row_in, n_pick, stride, out);
  return 0;
}
