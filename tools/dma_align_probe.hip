// Does global_load_lds_dwordx4 (the LDS-DMA piece of K1's tile kernel) accept global addresses that are only 8- or
// 4-byte aligned?  One wave copies 1 KiB pieces from src + shift bytes into LDS and writes them back out; the host
// compares.  (Rows of an odd atom count are 8-byte (float64) or 4-byte (float32) aligned, not 16.)
//   hipcc --offload-arch=gfx950 -O3 tools/dma_align_probe.hip -o /tmp/dma_align_probe && /tmp/dma_align_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ __launch_bounds__(64) void probe(const char* __restrict__ src, int shift, int pieces, char* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  for (int q = 0; q < pieces; ++q)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + shift + q * 1024 + lane * 16),
                                     (__attribute__((address_space(3))) void*)(smem + q * 1024), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int e = lane; e < pieces * 1024; e += 64) out[e] = smem[e];
}

int main() {
  const int pieces = 8, bytes = pieces * 1024 + 64;
  std::vector<char> h(bytes);
  for (int i = 0; i < bytes; ++i) h[i] = (char)(i * 37 + 11);
  char *src, *out;
  hipMalloc(&src, bytes);
  hipMalloc(&out, pieces * 1024);
  hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice);
  for (int shift : {0, 8, 4, 12, 2, 1}) {
    hipMemset(out, 0, pieces * 1024);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), pieces * 1024, 0, src, shift, pieces, out);
    hipError_t e = hipDeviceSynchronize();
    std::vector<char> r(pieces * 1024);
    hipMemcpy(r.data(), out, pieces * 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < pieces * 1024; ++i) bad += r[i] != h[i + shift];
    printf("{\"probe\": \"global_load_lds_dwordx4 alignment\", \"shift_bytes\": %d, \"status\": \"%s\", \"wrong_bytes\": %d}\n", shift,
           hipGetErrorString(e), bad);
    if (e != hipSuccess) return 1;
  }
  return 0;
}
