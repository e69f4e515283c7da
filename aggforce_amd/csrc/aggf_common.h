// Shared device/host helpers for libaggf (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>

#include "../../include/aggf.h"

namespace aggf {

// ---- error plumbing (thread-local last error string; see aggf_last_error) ----
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

#define AGGF_HIP_OK(expr)                                                         \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess)                                                         \
      return ::aggf::fail(AGGF_ERR_HIP, "%s failed: %s (%s:%d)", #expr,           \
                          hipGetErrorString(_e), __FILE__, __LINE__);             \
  } while (0)

#define AGGF_LAUNCH_OK()                                                          \
  do {                                                                            \
    hipError_t _e = hipGetLastError();                                            \
    if (_e != hipSuccess)                                                         \
      return ::aggf::fail(AGGF_ERR_HIP, "kernel launch failed: %s (%s:%d)",       \
                          hipGetErrorString(_e), __FILE__, __LINE__);             \
  } while (0)

// ---- launch coverage (aggf_coverage_dump): every kernel launch of the library goes through AGGF_LAUNCH, which counts
// the launch under the kernel's host handle -- the set of template instantiations a process actually executed, resolved
// to symbol names on request.  tests/test_gpu_zz_coverage.py compares it with the kernels the library contains.
void cover_hit(const void* kernel_handle);
#define AGGF_LAUNCH(kernel, ...)                                   \
  do {                                                             \
    ::aggf::cover_hit(reinterpret_cast<const void*>(kernel));      \
    hipLaunchKernelGGL(kernel, __VA_ARGS__);                       \
  } while (0)

// ---- adversarial dispatch order (TEST-ONLY build: `make order` -> build_order/libaggf_order.so, compiled with
// -DAGGF_ORDER_TEST; the shipped library contains none of this).  Launches that read and write one buffer from several
// workgroups (K2's step / panel / trailing / back-substitution products, the slab sums with `accumulate`, the triangle
// unpack, the pinned scatter: DESIGN.md section 5b lists them) are made through AGGF_LAUNCH_GATED; in the test build
// their workgroups then run STRICTLY ONE AFTER THE OTHER, in ascending (AGGF_ORDER=forward) or descending
// (AGGF_ORDER=reverse) block order: a result that depends on which workgroup runs first differs between the two
// orders and from the oracle -- deterministically, in one run, instead of once in a thousand.  A workgroup waits for
// its turn on a device counter (bounded spin: a grid larger than `limit` -- what is surely resident at once -- is
// not gated, a spin that runs out is counted and reported), the launch is followed by a synchronise + read-back of
// the time-outs.  aggf_order_note() (aggf_util.hip) keeps the process totals and prints them when the library unloads.
#ifdef AGGF_ORDER_TEST
struct OrderGate {
  unsigned int done, mode, timeouts, pad;
};
void order_note(int gated, int timeouts, const char* kernel);
int order_mode();  // 0 off, 1 forward, 2 reverse (AGGF_ORDER)
static __device__ OrderGate g_order_gate;  // one per translation unit (no relocatable device code)
static __global__ void order_arm_kernel(unsigned int mode) {
  g_order_gate.done = 0;
  g_order_gate.mode = mode;
  g_order_gate.timeouts = 0;
}
__device__ __forceinline__ void order_enter() {
  const unsigned mode = g_order_gate.mode;
  if (mode == 0) return;
  const unsigned nb = gridDim.x * gridDim.y * gridDim.z;
  const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const unsigned turn = mode == 1 ? lin : nb - 1 - lin;
  if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(&g_order_gate.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != turn) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 20)) {  // ~0.25 s: never hang the GPU
        atomicAdd(&g_order_gate.timeouts, 1u);
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__device__ __forceinline__ void order_exit() {
  if (g_order_gate.mode == 0) return;
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
    __threadfence();
    atomicAdd(&g_order_gate.done, 1u);
  }
}
static inline void order_arm(hipStream_t stream, dim3 grid, unsigned limit) {
  const uint64_t nb = (uint64_t)grid.x * grid.y * grid.z;
  const unsigned mode = nb <= limit ? (unsigned)order_mode() : 0u;
  hipLaunchKernelGGL(order_arm_kernel, dim3(1), dim3(1), 0, stream, mode);
}
static inline void order_collect(hipStream_t stream, dim3 grid, unsigned limit, const char* kernel) {
  const uint64_t nb = (uint64_t)grid.x * grid.y * grid.z;
  OrderGate g = {0, 0, 0, 0};
  if (order_mode() == 0) return;
  if (hipStreamSynchronize(stream) == hipSuccess) (void)hipMemcpyFromSymbol(&g, HIP_SYMBOL(g_order_gate), sizeof(g));
  order_note(nb <= limit ? 1 : 0, (int)g.timeouts, kernel);
}
// the kernel's body between the two macros runs as a lambda so that its early `return`s still reach the exit
#define AGGF_GATED_BODY_BEGIN ::aggf::order_enter(); [&]() {
#define AGGF_GATED_BODY_END }(); ::aggf::order_exit();
#define AGGF_LAUNCH_GATED(limit, kernel, grid, block, lds, stream, ...)      \
  do {                                                                       \
    const dim3 og_ = (grid);                                                 \
    ::aggf::order_arm(stream, og_, (limit));                                 \
    AGGF_LAUNCH(kernel, og_, block, lds, stream, __VA_ARGS__);               \
    ::aggf::order_collect(stream, og_, (limit), #kernel);                    \
  } while (0)
#else
#define AGGF_GATED_BODY_BEGIN
#define AGGF_GATED_BODY_END
#define AGGF_LAUNCH_GATED(limit, kernel, ...) AGGF_LAUNCH(kernel, __VA_ARGS__)
#endif

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int64_t ceil_div(int64_t x, int64_t m) { return (x + m - 1) / m; }
int device_cu_count();

// One flag per (call site, device): hipFuncSetAttribute applies to the CURRENT device only, so a
// process that drives several GPUs must repeat it on each of them.
struct PerDeviceOnce {
  bool done[64] = {};
  bool* flag() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    return &done[dev];
  }
};

typedef double __attribute__((ext_vector_type(4))) f64x4;
typedef float __attribute__((ext_vector_type(4))) f32x4;

// ---- MFMA 16x16x4 wrappers.  A: lane l holds A[i=l&15][k=l>>4]; B: B[k=l>>4][j=l&15].
//      C/D: col = l&15 for both; row = (l>>4)+4*r for f64, (l>>4)*4+r for f32
//      (cdna_hip_programming.md section 3, "Fragment layout").
template <typename T>
struct Mfma;

template <>
struct Mfma<double> {
  using acc_t = f64x4;
  __device__ static __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  __device__ static __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};

template <>
struct Mfma<float> {
  using acc_t = f32x4;
  __device__ static __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  __device__ static __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};

template <typename T>
__device__ __forceinline__ typename Mfma<T>::acc_t acc_zero() {
  typename Mfma<T>::acc_t z = {0, 0, 0, 0};
  return z;
}

// 16-byte vector of T
template <typename T>
struct Vec16;
template <>
struct Vec16<double> {
  typedef double __attribute__((ext_vector_type(2))) type;
  static constexpr int N = 2;
};
template <>
struct Vec16<float> {
  typedef float __attribute__((ext_vector_type(4))) type;
  static constexpr int N = 4;
};

// ---- Philox4x32-10 (Salmon et al., SC'11): counter = 64-bit quad index, key = seed ----
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c[0];
    const uint64_t p1 = (uint64_t)M1 * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += W0; k1 += W1;
  }
}

// four standard normals for quad index q of stream `seed` (Box-Muller on 32-bit uniforms)
__device__ __forceinline__ void normal_quad(uint64_t seed, uint64_t stream, int64_t q, double z[4]) {
  uint32_t c[4] = {(uint32_t)q, (uint32_t)((uint64_t)q >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const double u1 = ((double)c[2 * h] + 0.5) * (1.0 / 4294967296.0);
    const double u2 = ((double)c[2 * h + 1] + 0.5) * (1.0 / 4294967296.0);
    const double rad = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincos(6.283185307179586476925 * u2, &sn, &cs);
    z[2 * h] = rad * cs;
    z[2 * h + 1] = rad * sn;
  }
}

}  // namespace aggf
