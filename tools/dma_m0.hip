// LDS-DMA: does rewriting M0 between instructions serialise a wave's DMAs?  Three pieces of a row with ONE M0 and
// instruction offsets 0/1024/2048 (added to both the global and the LDS address) vs one M0 write per piece.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const double* __restrict__ src, int64_t foot_elems, int iters, double* sink,
                                            int check) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* smem = reinterpret_cast<double*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const double* base = src + (int64_t)blockIdx.x * foot_elems;
  for (int it = 0; it < iters; ++it) {
    const int64_t off = ((int64_t)it * 3072) % foot_elems;
#pragma unroll
    for (int row = 0; row < 2; ++row) {
      const double* g = base + off + (wave * 2 + row) * 384 + lane * 2;     // one 3 KiB row = 3 pieces
      double* l = smem + ((it % 3) * 8 + wave * 2 + row) * 384;
      if (MODE == 0) {
#pragma unroll
        for (int cp = 0; cp < 3; ++cp)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + cp * 128),
                                           (__attribute__((address_space(3))) void*)(l + cp * 128), 16, 0, 0);
      } else {
        const uint32_t lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)l;
        asm volatile(
            "s_mov_b32 m0, %0\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, off\n\t"
            "global_load_lds_dwordx4 %1, off offset:1024\n\t"
            "global_load_lds_dwordx4 %1, off offset:2048\n\t"
            :
            : "s"(lds_addr), "v"(g)
            : "memory", "m0");
      }
    }
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (check) {
    // after the last iteration the ring slot holds rows [off .. off+3072) of this block's source
    const int it = iters - 1;
    const int64_t off = ((int64_t)it * 3072) % foot_elems;
    int bad = 0;
    for (int e = tid; e < 3072; e += 256)
      if (smem[(it % 3) * 3072 + e] != base[off + e]) ++bad;
    if (bad) atomicAdd((int*)sink, bad);
  }
}

template <int MODE>
static void run(const char* name, const double* src, int64_t foot_bytes, double* sink) {
  const int iters = 20000, nb = 512;
  const size_t lds = 3 * 24 * 1024;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipMemset(sink, 0, 8);
  hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), lds, 0, src, foot_bytes / 8, 7, sink, 1);
  int bad = -1;
  hipMemcpy(&bad, sink, 4, hipMemcpyDeviceToHost);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), lds, 0, src, foot_bytes / 8, iters, sink, 0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  const double bytes = (double)nb * iters * 24576.0;
  printf("%-36s foot %8ld KiB/WG: %7.2f ms  %6.2f TB/s  %5.1f B/clk/CU  (wrong elements in check: %d)\n", name,
         (long)(foot_bytes >> 10), best, bytes / best / 1e9, bytes / (best * 1e-3) / 256 / 2.4e9, bad);
}

int main() {
  double *src, *sink;
  const size_t total = (size_t)512 * (8 << 20);
  hipMalloc(&src, total);
  hipMalloc(&sink, 4096);
  // distinct values so that misplaced pieces are seen
  double* h = (double*)malloc(total);
  for (size_t i = 0; i < total / 8; ++i) h[i] = (double)i;
  hipMemcpy(src, h, total, hipMemcpyHostToDevice);
  for (int64_t foot : {(int64_t)24 << 10, (int64_t)6 << 20}) {
    run<0>("M0 rewritten per piece (builtin)", src, foot, sink);
    run<1>("one M0 per row + inst offsets", src, foot, sink);
  }
  return 0;
}
