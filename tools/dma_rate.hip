// LDS-DMA (global_load_lds_dwordx4) throughput per CU, alone and against ds_read traffic.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dma_rate.hip -o tools/dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// MODE 0: DMA only; 1: DMA + ds_read_b64 stream; 2: global_load_dwordx4 to VGPR (no LDS);
// FOOT: bytes of source per workgroup that the loop cycles through (L1: 16 KB, L2: 512 KB, HBM: 64 MB)
template <int MODE>
__global__ __launch_bounds__(256, 2) void dma_kernel(const double* __restrict__ src, int64_t foot_elems, int iters,
                                                     double* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* smem = reinterpret_cast<double*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const double* base = src + (int64_t)blockIdx.x * foot_elems;
  double acc = 0.0;
  double4 r4 = {0, 0, 0, 0};
  // per iteration: 6 pieces of 1 KiB per wave (= one Gram stage of the unit kernel)
  for (int it = 0; it < iters; ++it) {
    const int64_t off = ((int64_t)it * 3072) % foot_elems;  // 24 KiB per workgroup and iteration
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const double* g = base + off + (wave * 6 + q) * 128 + lane * 2;
      if (MODE == 2) {
        const double4 v = *reinterpret_cast<const double4*>(base + off + ((wave * 6 + q) * 128 + lane * 2) % 3072);
        r4.x += v.x; r4.y += v.y;
      } else {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(smem + ((it % 3) * 24 + wave * 6 + q) * 128),
                                         16, 0, 0);
      }
    }
    if (MODE == 1) {
      // 24 ds_read_b64 per wave and stage, like the MFMA operand reads
#pragma unroll
      for (int k = 0; k < 24; ++k) acc += smem[9216 + ((k * 67 + lane * 3 + wave * 400) % 3072)];
    }
    if (MODE != 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 123.456 || r4.x + r4.y == 123.456) sink[tid] = acc + smem[tid];
}

template <int MODE>
static void run(const char* name, const double* src, int64_t foot_bytes, double* sink) {
  const int iters = 20000, nb = 512;
  const size_t lds = 3 * 24 * 1024 + 3072 * 8;
  hipFuncSetAttribute((const void*)dma_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(dma_kernel<MODE>, dim3(nb), dim3(256), lds, 0, src, foot_bytes / 8, iters, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  const double bytes = (double)nb * iters * 24576.0;
  printf("%-44s foot %8ld KiB/WG: %7.2f ms  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz)\n", name, (long)(foot_bytes >> 10), best,
         bytes / best / 1e9, bytes / (best * 1e-3) / 256 / 2.4e9);
}

int main() {
  double *src, *sink;
  const size_t total = (size_t)512 * (64 << 20);  // 32 GiB
  if (hipMalloc(&src, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&sink, 4096);
  hipMemset(src, 0, total);
  for (int64_t foot : {(int64_t)24 << 10, (int64_t)480 << 10, (int64_t)60 << 20}) {
    run<0>("LDS-DMA dwordx4 only", src, foot, sink);
    run<1>("LDS-DMA dwordx4 + ds_read_b64 stream", src, foot, sink);
    run<2>("global_load_dwordx4 -> VGPR", src, foot, sink);
  }
  return 0;
}
