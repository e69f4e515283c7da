// LDS-DMA stream from HBM in the shape of MI355X_MICROARCH.md's "ldsdma-fill" row: ONE workgroup per CU, a few LOADER
// waves that fill a ring of 16 KiB LDS slots with 16 x 1 KiB global_load_lds_dwordx4 each and keep F fills in flight
// behind a counted vmcnt; nothing consumes the slots.  Question: does the stream of one workgroup per CU
// reach the guide's 6.4 TB/s (6.5-6.8 with nt), or the 3.5-4.4 TB/s that round 4's wave-specialised Gram kernel saw?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ldsdma_fill.hip -o tools/ldsdma_fill && tools/ldsdma_fill
// Prints one JSON line per configuration.  Data: 16.8 GB (CLN025 x 4e6 frames of 525 float64), every byte read once.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int SLOT = 16384;  // bytes per fill
constexpr int RING = 8;      // slots (128 KiB)

// loaders: waves that issue DMAs (the workgroup has `loaders` waves in all); in_flight: fills kept in flight per loader
// wave (<= RING / loaders... the ring is shared round-robin); interleave: fill f of the trajectory goes to workgroup
// f % gridDim.x (the workgroups together sweep one dense window) or every workgroup owns a contiguous range
template <bool NT, int IN_FLIGHT>
__global__ __launch_bounds__(256) void fill_kernel(const char* __restrict__ src, int64_t n_fills, int interleave,
                                                   unsigned long long* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_waves = blockDim.x >> 6;
  // fills of this workgroup: k-th fill = global fill index g(k)
  const int64_t per_wg = (n_fills + gridDim.x - 1) / gridDim.x;
  int64_t count = interleave ? (blockIdx.x < n_fills ? (n_fills - 1 - blockIdx.x) / gridDim.x + 1 : 0)
                             : (per_wg * blockIdx.x < n_fills ? (n_fills - per_wg * blockIdx.x < per_wg ? n_fills - per_wg * blockIdx.x : per_wg) : 0);
  // this wave takes fills wave, wave + n_waves, ... of the workgroup's list; slot = (its own counter) % (RING / n_waves)
  const int slots_per_wave = RING / n_waves;
  int issued = 0;
  for (int64_t k = wave; k < count; k += n_waves, ++issued) {
    const int64_t g = interleave ? (int64_t)blockIdx.x + k * gridDim.x : per_wg * blockIdx.x + k;
    const char* base = src + g * SLOT;
    char* slot = smem + ((wave * slots_per_wave) + (issued % slots_per_wave)) * SLOT;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + q * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(slot + q * 1024), 16, 0, NT ? 2 : 0);
    }
    // keep IN_FLIGHT fills (16 pieces each) outstanding
    if constexpr (IN_FLIGHT == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (IN_FLIGHT == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (IN_FLIGHT == 3) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (sink && threadIdx.x == 0 && smem[17] == 123) sink[blockIdx.x] = smem[5];
}

// the same bytes through registers: global_load_dwordx4, 8 loads in flight per thread, 512 threads, 2 workgroups per CU
// (gram_small_kernel's fetch shape)
template <bool NT>
__global__ __launch_bounds__(512) void reg_kernel(const char* __restrict__ src, int64_t n_fills, unsigned long long* __restrict__ sink) {
  typedef float __attribute__((ext_vector_type(4))) v4;
  v4 acc = {0, 0, 0, 0};
  // a "stage" = 2 fills = 32 KiB: 4 x 16 B per thread
  const int64_t n_stage = n_fills / 2;
  for (int64_t s = blockIdx.x; s < n_stage; s += gridDim.x) {
    const v4* p = reinterpret_cast<const v4*>(src + s * 2 * SLOT);
    v4 h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = NT ? __builtin_nontemporal_load(p + threadIdx.x + 512 * i) : p[threadIdx.x + 512 * i];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc += h[i];
  }
  if (sink && acc.x == 123.456f) sink[blockIdx.x] = 1;
}

template <typename F>
static float time_ms(F&& launch) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(a);
    launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (rep > 0 && ms < best) best = ms;
  }
  return best;
}

template <bool NT, int IN_FLIGHT>
static void run_fill(const char* src, int64_t n_fills, int loaders, int wgs, int interleave) {
  hipFuncSetAttribute((const void*)fill_kernel<NT, IN_FLIGHT>, hipFuncAttributeMaxDynamicSharedMemorySize, RING * SLOT);
  const float ms = time_ms([&] {
    hipLaunchKernelGGL((fill_kernel<NT, IN_FLIGHT>), dim3(wgs), dim3(64 * loaders), RING * SLOT, 0, src, n_fills, interleave,
                       (unsigned long long*)nullptr);
  });
  const double bytes = (double)n_fills * SLOT;
  printf("{\"probe\": \"ldsdma_fill\", \"loader_waves\": %d, \"fills_in_flight_per_wave\": %d, \"KiB_in_flight_per_CU\": %d, \"nt\": %s, "
         "\"workgroups\": %d, \"interleaved\": %s, \"ms\": %.3f, \"TBps\": %.3f, \"frac_of_8TBps\": %.3f}\n",
         loaders, IN_FLIGHT, loaders * IN_FLIGHT * 16, NT ? "true" : "false", wgs, interleave ? "true" : "false", ms,
         bytes / ms / 1e9, bytes / ms / 1e9 / 8.0);
  fflush(stdout);
}

int main() {
  const int64_t bytes = (int64_t)4000000 * 525 * 8;  // 16.8 GB
  const int64_t n_fills = bytes / SLOT;
  char* src;
  if (hipMalloc(&src, (size_t)n_fills * SLOT) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(src, 1, (size_t)n_fills * SLOT);
  hipDeviceSynchronize();
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  for (int interleave : {1, 0}) {
    for (int loaders : {1, 2, 4}) {
      run_fill<false, 1>(src, n_fills, loaders, cus, interleave);
      run_fill<false, 2>(src, n_fills, loaders, cus, interleave);
      if (loaders <= 2) run_fill<false, 3>(src, n_fills, loaders, cus, interleave);
      if (loaders <= 2) run_fill<false, 4>(src, n_fills, loaders, cus, interleave);
      run_fill<true, 2>(src, n_fills, loaders, cus, interleave);
      if (loaders <= 2) run_fill<true, 4>(src, n_fills, loaders, cus, interleave);
    }
  }
  for (int nt = 0; nt < 2; ++nt) {
    const float ms = nt ? time_ms([&] { hipLaunchKernelGGL((reg_kernel<true>), dim3(2 * cus), dim3(512), 0, 0, src, n_fills, (unsigned long long*)nullptr); })
                        : time_ms([&] { hipLaunchKernelGGL((reg_kernel<false>), dim3(2 * cus), dim3(512), 0, 0, src, n_fills, (unsigned long long*)nullptr); });
    printf("{\"probe\": \"global_load_dwordx4 -> VGPR, 2 x 512 threads per CU, 4 loads in flight per thread\", \"nt\": %s, \"ms\": %.3f, "
           "\"TBps\": %.3f, \"frac_of_8TBps\": %.3f}\n", nt ? "true" : "false", ms, (double)n_fills * SLOT / ms / 1e9,
           (double)n_fills * SLOT / ms / 1e9 / 8.0);
  }
  return 0;
}
