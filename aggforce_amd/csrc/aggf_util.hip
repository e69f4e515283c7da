// Error plumbing, small streaming kernels (NaN scan, allclose, sum of squares, map
// expansion) and the counter-based synthetic trajectory generator.
#include <cxxabi.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdlib.h>

#include <atomic>

#include "aggf_common.h"

namespace aggf {

// ---- launch coverage: a fixed open-addressing table kernel handle -> launch count, lock-free (a launch costs two
// atomic operations).  A process that has AGGF_COVERAGE_FILE set appends its table to that file when the library is
// unloaded (one line per kernel: label <tab> mangled name <tab> count; the label is AGGF_COVERAGE_LABEL at that moment):
// the test session collects the launches of its child processes that way.
constexpr int COVER_SLOTS = 4096;  // power of two, ~10x the kernels of the library
static std::atomic<const void*> g_cover_key[COVER_SLOTS];
static std::atomic<uint64_t> g_cover_cnt[COVER_SLOTS];    // since the last aggf_coverage_reset
static std::atomic<uint64_t> g_cover_total[COVER_SLOTS];  // since the library was loaded

void cover_hit(const void* h) {
  size_t i = ((uintptr_t)h >> 3) * 0x9E3779B97F4A7C15ull >> 52;
  for (int probe = 0; probe < COVER_SLOTS; ++probe, i = (i + 1) & (COVER_SLOTS - 1)) {
    const void* k = g_cover_key[i].load(std::memory_order_relaxed);
    if (k == nullptr) {
      const void* expected = nullptr;
      if (g_cover_key[i].compare_exchange_strong(expected, h, std::memory_order_relaxed)) k = h; else k = expected;
    }
    if (k == h) {
      g_cover_cnt[i].fetch_add(1, std::memory_order_relaxed);
      g_cover_total[i].fetch_add(1, std::memory_order_relaxed);
      return;
    }
  }
}

#ifdef AGGF_ORDER_TEST
// process totals of the adversarial-order test build (aggf_common.h), printed when the library unloads
static std::atomic<int> g_order_gated{0}, g_order_ungated{0}, g_order_timeouts{0};
int order_mode() {
  const char* e = getenv("AGGF_ORDER");  // read per launch: a test may switch between the orders
  if (!e) return 0;
  return e[0] == 'f' ? 1 : e[0] == 'r' ? 2 : 0;
}
void order_note(int gated, int timeouts, const char* kernel) {
  (gated ? g_order_gated : g_order_ungated).fetch_add(1);
  if (timeouts > 0) {
    g_order_timeouts.fetch_add(timeouts);
    fprintf(stderr, "AGGF_ORDER: %d workgroup(s) of %s gave up waiting for their turn\n", timeouts, kernel);
  }
}
namespace {
struct OrderAtExit {
  ~OrderAtExit() {
    fprintf(stderr, "AGGF_ORDER_SUMMARY mode=%d gated=%d ungated=%d timeouts=%d\n", order_mode(), g_order_gated.load(),
            g_order_ungated.load(), g_order_timeouts.load());
  }
};
static OrderAtExit g_order_at_exit;
}  // namespace
#endif

// "mangled name<tab>demangled name<tab>launches since the last reset<tab>launches in all<newline>" per executed kernel into buf (always NUL-terminated when
// n > 0), with `label<tab>` in front of every line if given; returns the bytes the full text needs (without the NUL)
static size_t cover_dump(char* buf, size_t n, const char* label) {
  size_t need = 0;
  if (n > 0) buf[0] = 0;
  for (int i = 0; i < COVER_SLOTS; ++i) {
    const void* k = g_cover_key[i].load(std::memory_order_relaxed);
    if (!k) continue;
    Dl_info info;
    char addr[32];
    const char* name = nullptr;
    if (dladdr(k, &info) && info.dli_sname && info.dli_saddr == k) name = info.dli_sname;
    if (!name) {
      snprintf(addr, sizeof(addr), "?%p", k);
      name = addr;
    }
    int status = 1;
    char* pretty = name[0] == '_' ? abi::__cxa_demangle(name, nullptr, nullptr, &status) : nullptr;
    std::string line;
    if (label) line.append(label).append("\t");
    line.append(name).append("\t").append(status == 0 && pretty ? pretty : name).append("\t");
    line.append(std::to_string((unsigned long long)g_cover_cnt[i].load())).append("\t");
    line.append(std::to_string((unsigned long long)g_cover_total[i].load())).append("\n");
    free(pretty);
    if (need + line.size() < n) {
      memcpy(buf + need, line.data(), line.size());
      buf[need + line.size()] = 0;
    }
    need += line.size();
  }
  return need;
}

namespace {
struct CoverAtExit {
  ~CoverAtExit() {
    const char* path = getenv("AGGF_COVERAGE_FILE");
    if (!path || !path[0]) return;
    const char* label = getenv("AGGF_COVERAGE_LABEL");
    const size_t need = cover_dump(nullptr, 0, label ? label : "-");
    if (need == 0) return;
    std::string text(need + 1, '\0');
    cover_dump(&text[0], need + 1, label ? label : "-");
    FILE* f = fopen(path, "a");  // O_APPEND: one write per process, whole lines
    if (!f) return;
    fwrite(text.data(), 1, need, f);
    fclose(f);
  }
};
static CoverAtExit g_cover_at_exit;
}  // namespace

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

int device_cu_count() {
  static thread_local int cached = 0;
  if (cached > 0) return cached;
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
    n = 256;  // MI355X
  cached = n;
  return n;
}

// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void has_nan_kernel(const T* __restrict__ x, int64_t n,
                                                      int32_t* __restrict__ flag) {
  bool found = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const T v = x[i];
    found |= (v != v);
  }
  if (__any(found) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

template <typename T>
__global__ __launch_bounds__(256) void not_close_kernel(const T* __restrict__ a,
                                                        const T* __restrict__ b, int64_t n,
                                                        double rtol, double atol,
                                                        int32_t* __restrict__ flag) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const double x = (double)a[i], y = (double)b[i];
    // np.isclose: finite -> |x-y| <= atol + rtol|y|; equal infinities are close; NaN never
    const bool ok = (x == y) || (fabs(x - y) <= atol + rtol * fabs(y));
    bad |= !ok;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

constexpr int SUMSQ_BLOCKS = 1024;

template <typename T>
__global__ __launch_bounds__(256) void sumsq_kernel(const T* __restrict__ x, int64_t n,
                                                    double* __restrict__ partials) {
  // Fixed order, whatever the hardware does: workgroup w takes the 4-KiB chunks w, w + G, w + 2G, ... (G = the fixed
  // grid: together the workgroups sweep the array as one dense front -- with a private contiguous slice per workgroup
  // 1024 far-apart streams hit the HBM channels at once: 4.2 TB/s), a thread one 16-byte piece of each chunk, four
  // chunks in flight, four accumulators combined in a fixed order; then lane tree, wave sum.
  constexpr int V = 16 / sizeof(T);
  typedef T __attribute__((ext_vector_type(V))) vec_t;
  const int64_t n_vec = (((uintptr_t)x & 15) == 0) ? n / V : 0;  // whole 16-byte pieces (an unaligned array: scalar tail only)
  const int64_t n_chunk = n_vec / 256;                              // whole chunks of 256 pieces
  const vec_t* xv = reinterpret_cast<const vec_t*>(x);
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  int64_t c = blockIdx.x;
  for (; c + 3 * (int64_t)gridDim.x < n_chunk; c += 4 * (int64_t)gridDim.x) {
    vec_t v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(xv + (c + u * (int64_t)gridDim.x) * 256 + threadIdx.x);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < V; ++e) acc[u] += (double)v[u][e] * (double)v[u][e];
  }
  for (; c < n_chunk; c += gridDim.x) {
    const vec_t v = __builtin_nontemporal_load(xv + c * 256 + threadIdx.x);
#pragma unroll
    for (int e = 0; e < V; ++e) acc[0] += (double)v[e] * (double)v[e];
  }
  // what is left behind the last whole chunk (fewer than 256 pieces + a ragged end): the last workgroup, in order
  if (blockIdx.x == gridDim.x - 1)
    for (int64_t i = n_chunk * 256 * V + threadIdx.x; i < n; i += 256) acc[1] += (double)x[i] * (double)x[i];
  double s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  __shared__ double w[4];
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (w[0] + w[1]) + (w[2] + w[3]);
}

__global__ __launch_bounds__(256) void sum_fixed_kernel(const double* __restrict__ part, int n,
                                                        double* __restrict__ out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}

__global__ __launch_bounds__(256) void expand_map_kernel(const double* __restrict__ X,
                                                         int32_t n_rows, int32_t n_red,
                                                         const int32_t* __restrict__ goa,
                                                         int32_t N, double* __restrict__ W) {
  const int64_t total = (int64_t)n_rows * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / N;
    const int a = (int)(i - r * N);
    W[i] = X[r * n_red + goa[a]];
  }
}

// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void synth_normal_kernel(T* __restrict__ out, int64_t T_frames,
                                                           int32_t N, uint64_t seed,
                                                           int64_t frame_offset, double mean,
                                                           double sigma, double lattice) {
  const int64_t row = (int64_t)N * 3;
  const int64_t g_begin = frame_offset * row;
  const int64_t g_end = g_begin + T_frames * row;
  const int64_t q_begin = g_begin / 4, q_end = (g_end + 3) / 4;
  int side = 1;
  if (lattice != 0.0) {
    side = (int)ceil(cbrt((double)N));
    while ((int64_t)side * side * side < N) ++side;
  }
  for (int64_t q = q_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < q_end;
       q += (int64_t)gridDim.x * blockDim.x) {
    double z[4];
    normal_quad(seed, 0, q, z);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t g = 4 * q + e;
      if (g < g_begin || g >= g_end) continue;
      double mu = mean;
      if (lattice != 0.0) {
        const int64_t in_row = g % row;
        const int a = (int)(in_row / 3), d = (int)(in_row - 3 * (int64_t)a);
        const int coord = d == 0 ? a % side : (d == 1 ? (a / side) % side : a / (side * side));
        mu += lattice * coord;
      }
      out[g - g_begin] = (T)(mu + sigma * z[e]);
    }
  }
}

static dim3 stream_grid(int64_t n) {
  int64_t g = ceil_div(n, 256);
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return dim3((unsigned)g);
}

// ---- frame-sized housekeeping of the Python layer as kernels of the library (the reference does these with NumPy
// fancy indexing, np.concatenate and scalar products on (n_frames, n_sites, 3) arrays): ------------------------------
// out[i, :] = src[idx[i], :] for whole frames (rows of row_elems elements): fold / sample selection
// (agg.py:208-231 `coords[train_inds]`, featlinearmap.py:447-452).  One workgroup walks rows, 16 bytes per lane when
// the row allows it.
template <typename T>
__global__ __launch_bounds__(256) void take_frames_kernel(const T* __restrict__ src, int64_t n_src, int64_t row_elems,
                                                          const int64_t* __restrict__ idx, int64_t n, int vec_ok,
                                                          T* __restrict__ out) {
  constexpr int V = 16 / sizeof(T);
  typedef T __attribute__((ext_vector_type(V))) vec_t;
  for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
    const int64_t r = idx[i];
    if (r < 0 || r >= n_src) {  // (the host checks the range; a bypassed check shows as NaN rows, never as a read outside the array)
      for (int64_t e = threadIdx.x; e < row_elems; e += blockDim.x) out[i * row_elems + e] = (T)NAN;
      continue;
    }
    const T* s = src + r * row_elems;
    T* o = out + i * row_elems;
    if (vec_ok) {
      for (int64_t e = threadIdx.x; e < row_elems / V; e += blockDim.x)
        reinterpret_cast<vec_t*>(o)[e] = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(s) + e);
    } else {
      for (int64_t e = threadIdx.x; e < row_elems; e += blockDim.x) o[e] = s[e];
    }
  }
}

// out[t] = [a[t] ; b[t]] along the site axis, converted to the output type (np.concatenate(..., axis=1) of
// trajectory/core.py:388-390 and map/tmap.py:430-436)
template <typename TA, typename TB, typename TO>
__global__ __launch_bounds__(256) void concat_sites_kernel(const TA* __restrict__ a, int64_t row_a, const TB* __restrict__ b,
                                                           int64_t row_b, int64_t nT, TO* __restrict__ out) {
  const int64_t row_o = row_a + row_b, total = nT * row_o;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = e / row_o, c = e - t * row_o;
    out[e] = c < row_a ? (TO)a[t * row_a + c] : (TO)b[t * row_b + (c - row_a)];
  }
}

// out = alpha * x (NullForcesTMap's `fill_value * coords`, map/tmap.py:399-401; jgauss.py:573 `0 * traj.forces`):
// a product, not a fill -- NaN x anything and 0 x inf stay what NumPy makes of them
template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(const T* __restrict__ x, int64_t n, T alpha, T* __restrict__ out) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    out[e] = alpha * x[e];
}

}  // namespace aggf

using namespace aggf;

extern "C" int aggf_version(void) { return AGGF_VERSION; }

extern "C" const char* aggf_last_error(void) { return aggf::g_last_error.c_str(); }

extern "C" size_t aggf_coverage_dump(char* buf, size_t buf_bytes) { return aggf::cover_dump(buf, buf_bytes, nullptr); }

extern "C" int aggf_coverage_reset(void) {
  for (int i = 0; i < aggf::COVER_SLOTS; ++i) aggf::g_cover_cnt[i].store(0);
  return AGGF_OK;
}

extern "C" int aggf_device_info(int32_t* cu_count, size_t* free_bytes, size_t* total_bytes) {
  int dev = 0, n = 0;
  AGGF_HIP_OK(hipGetDevice(&dev));
  AGGF_HIP_OK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
  size_t f = 0, t = 0;
  AGGF_HIP_OK(hipMemGetInfo(&f, &t));
  if (cu_count) *cu_count = n;
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return AGGF_OK;
}

extern "C" int aggf_has_nan(const void* x, int64_t count, int dtype, int32_t* flag, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!x || !flag) return fail(AGGF_ERR_ARG, "aggf_has_nan: NULL pointer");
  if (count <= 0) return AGGF_OK;
  if (dtype == AGGF_F64)
    AGGF_LAUNCH(has_nan_kernel<double>, stream_grid(count), dim3(256), 0, stream, (const double*)x, count, flag);
  else if (dtype == AGGF_F32)
    AGGF_LAUNCH(has_nan_kernel<float>, stream_grid(count), dim3(256), 0, stream, (const float*)x, count, flag);
  else
    return fail(AGGF_ERR_ARG, "aggf_has_nan: bad dtype");
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_not_close(const void* a, const void* b, int64_t count, int dtype, double rtol,
                              double atol, int32_t* flag, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!a || !b || !flag) return fail(AGGF_ERR_ARG, "aggf_not_close: NULL pointer");
  if (count <= 0) return AGGF_OK;
  if (dtype == AGGF_F64)
    AGGF_LAUNCH(not_close_kernel<double>, stream_grid(count), dim3(256), 0, stream, (const double*)a, (const double*)b, count, rtol, atol, flag);
  else if (dtype == AGGF_F32)
    AGGF_LAUNCH(not_close_kernel<float>, stream_grid(count), dim3(256), 0, stream, (const float*)a, (const float*)b, count, rtol, atol, flag);
  else
    return fail(AGGF_ERR_ARG, "aggf_not_close: bad dtype");
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" size_t aggf_sumsq_workspace_bytes(void) { return SUMSQ_BLOCKS * sizeof(double); }

extern "C" int aggf_sumsq(const void* x, int64_t count, int dtype, double* out, void* ws,
                          size_t ws_bytes, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!x || !out || !ws) return fail(AGGF_ERR_ARG, "aggf_sumsq: NULL pointer");
  if (ws_bytes < SUMSQ_BLOCKS * sizeof(double)) return fail(AGGF_ERR_WORKSPACE, "aggf_sumsq: workspace too small");
  if (count < 0) return fail(AGGF_ERR_ARG, "aggf_sumsq: negative count");
  double* part = (double*)ws;
  if (dtype == AGGF_F64)
    AGGF_LAUNCH(sumsq_kernel<double>, dim3(SUMSQ_BLOCKS), dim3(256), 0, stream, (const double*)x, count, part);
  else if (dtype == AGGF_F32)
    AGGF_LAUNCH(sumsq_kernel<float>, dim3(SUMSQ_BLOCKS), dim3(256), 0, stream, (const float*)x, count, part);
  else
    return fail(AGGF_ERR_ARG, "aggf_sumsq: bad dtype");
  AGGF_LAUNCH_OK();
  AGGF_LAUNCH(sum_fixed_kernel, dim3(1), dim3(256), 0, stream, part, SUMSQ_BLOCKS, out);
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

// ---- packed upper triangle of symmetric matrices (the payload of the Gram all-reduce) ----
// packed[b][i * n - i (i - 1) / 2 + (j - i)] = G[b][i][j] for j >= i
__device__ __forceinline__ int64_t sym_row_start(int64_t i, int64_t n) { return i * n - i * (i - 1) / 2; }

__global__ __launch_bounds__(256) void sym_pack_kernel(const double* __restrict__ G, int32_t n, int64_t g_ps,
                                                       double* __restrict__ packed, int64_t p_ps) {
  const int64_t i = blockIdx.y, j = i + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < n) packed[blockIdx.z * p_ps + sym_row_start(i, n) + (j - i)] = G[blockIdx.z * g_ps + i * n + j];
}

// both triangles of G from the packed upper one; 32 x 32 tiles, the mirrored tile goes through LDS so that
// both writes are row-contiguous
__global__ __launch_bounds__(256) void sym_unpack_kernel(const double* __restrict__ packed, int32_t n, int64_t p_ps,
                                                         double* __restrict__ G, int64_t g_ps) {
  __shared__ double tile[32][33];
  AGGF_GATED_BODY_BEGIN
  const int ti = blockIdx.y, tj = blockIdx.x;
  if (tj < ti) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const double* P = packed + blockIdx.z * p_ps;
  double* Gb = G + blockIdx.z * g_ps;
  for (int r = ty; r < 32; r += 8) {
    const int64_t i = (int64_t)ti * 32 + r, j = (int64_t)tj * 32 + tx;
    double v = 0.0;
    if (i < n && j < n) {
      const int64_t a = i < j ? i : j, b = i < j ? j : i;  // (only the diagonal tile has j < i)
      v = P[sym_row_start(a, n) + (b - a)];
      Gb[i * n + j] = v;
    }
    tile[r][tx] = v;
  }
  if (tj == ti) return;
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int64_t i = (int64_t)tj * 32 + r, j = (int64_t)ti * 32 + tx;  // element (i, j) of the mirrored tile
    if (i < n && j < n) Gb[i * n + j] = tile[tx][r];
  }
  AGGF_GATED_BODY_END
}

extern "C" int aggf_sym_pack_upper(const double* G, int32_t n, int32_t batch, double* packed, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!G || !packed) return fail(AGGF_ERR_ARG, "aggf_sym_pack_upper: NULL pointer");
  if (n <= 0 || batch <= 0 || batch > 65535 || n > 65535) return fail(AGGF_ERR_ARG, "aggf_sym_pack_upper: bad size");
  const int64_t per = (int64_t)n * (n + 1) / 2;
  AGGF_LAUNCH(sym_pack_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)n, (unsigned)batch), dim3(256), 0, stream,
                     G, n, (int64_t)n * n, packed, per);
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_sym_unpack_upper(const double* packed, int32_t n, int32_t batch, double* G, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!G || !packed) return fail(AGGF_ERR_ARG, "aggf_sym_unpack_upper: NULL pointer");
  if (n <= 0 || batch <= 0 || batch > 65535 || n > 65535 * 32) return fail(AGGF_ERR_ARG, "aggf_sym_unpack_upper: bad size");
  const int64_t per = (int64_t)n * (n + 1) / 2;
  const unsigned nt = (unsigned)((n + 31) / 32);
  {
    // every workgroup reads its part of `packed` and writes two tiles of G: the buffers must not overlap (unpacking a
    // triangle into its own storage would let one workgroup overwrite what another still reads)
    const uintptr_t p0 = (uintptr_t)packed, p1 = p0 + (uintptr_t)batch * per * 8, g0 = (uintptr_t)G,
                    g1 = g0 + (uintptr_t)batch * n * n * 8;
    if (p0 < g1 && g0 < p1) return fail(AGGF_ERR_ARG, "aggf_sym_unpack_upper: packed and G overlap");
  }
  AGGF_LAUNCH_GATED(1024, sym_unpack_kernel, dim3(nt, nt, (unsigned)batch), dim3(256), 0, stream, packed, n, per, G,
                     (int64_t)n * n);
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_expand_map(const double* X, int32_t n_rows, int32_t n_red,
                               const int32_t* group_of_atom, int32_t N, double* W, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!X || !group_of_atom || !W) return fail(AGGF_ERR_ARG, "aggf_expand_map: NULL pointer");
  if (n_rows <= 0 || n_red <= 0 || N <= 0) return fail(AGGF_ERR_ARG, "aggf_expand_map: empty problem");
  AGGF_LAUNCH(expand_map_kernel, stream_grid((int64_t)n_rows * N), dim3(256), 0, stream, X, n_rows, n_red, group_of_atom, N, W);
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_synth_normal(void* out, int64_t T, int32_t N, int dtype, uint64_t seed,
                                 int64_t frame_offset, double mean, double sigma, double lattice,
                                 void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!out) return fail(AGGF_ERR_ARG, "aggf_synth_normal: NULL pointer");
  if (T <= 0 || N <= 0 || frame_offset < 0) return fail(AGGF_ERR_ARG, "aggf_synth_normal: bad shape");
  const int64_t quads = T * (int64_t)N * 3 / 4 + 2;
  dim3 grid = stream_grid(quads);
  if (dtype == AGGF_F64)
    AGGF_LAUNCH(synth_normal_kernel<double>, grid, dim3(256), 0, stream, (double*)out, T, N, seed, frame_offset, mean, sigma, lattice);
  else if (dtype == AGGF_F32)
    AGGF_LAUNCH(synth_normal_kernel<float>, grid, dim3(256), 0, stream, (float*)out, T, N, seed, frame_offset, mean, sigma, lattice);
  else
    return fail(AGGF_ERR_ARG, "aggf_synth_normal: bad dtype");
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_take_frames(const void* src, int64_t n_src, int64_t row_elems, int dtype, const int64_t* idx, int64_t n,
                                void* out, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!src || !idx || !out) return fail(AGGF_ERR_ARG, "aggf_take_frames: NULL pointer");
  if (n_src <= 0 || row_elems <= 0 || n < 0) return fail(AGGF_ERR_ARG, "aggf_take_frames: bad shape");
  if (dtype != AGGF_F32 && dtype != AGGF_F64) return fail(AGGF_ERR_ARG, "aggf_take_frames: bad dtype");
  if (n == 0) return AGGF_OK;
  const size_t es = dtype == AGGF_F64 ? 8 : 4;
  const int vec_ok = (((uintptr_t)src | (uintptr_t)out) & 15) == 0 && (row_elems * es) % 16 == 0;
  int64_t g = n < 8192 ? n : 8192;
  if (dtype == AGGF_F64)
    AGGF_LAUNCH(take_frames_kernel<double>, dim3((unsigned)g), dim3(256), 0, stream, (const double*)src, n_src, row_elems, idx, n, vec_ok, (double*)out);
  else
    AGGF_LAUNCH(take_frames_kernel<float>, dim3((unsigned)g), dim3(256), 0, stream, (const float*)src, n_src, row_elems, idx, n, vec_ok, (float*)out);
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_concat_sites(const void* a, int32_t Na, int a_dtype, const void* b, int32_t Nb, int b_dtype, int64_t T,
                                 void* out, int out_dtype, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!a || !b || !out) return fail(AGGF_ERR_ARG, "aggf_concat_sites: NULL pointer");
  if (Na <= 0 || Nb <= 0 || T < 0) return fail(AGGF_ERR_ARG, "aggf_concat_sites: bad shape");
  if (T == 0) return AGGF_OK;
  const int64_t ra = (int64_t)Na * 3, rb = (int64_t)Nb * 3;
  const dim3 grid = stream_grid(T * (ra + rb)), block(256);
#define AGGF_CC(TA_, TB_, TO_) \
  AGGF_LAUNCH((concat_sites_kernel<TA_, TB_, TO_>), grid, block, 0, stream, (const TA_*)a, ra, (const TB_*)b, rb, T, (TO_*)out)
  if (a_dtype == AGGF_F32 && b_dtype == AGGF_F32 && out_dtype == AGGF_F32) AGGF_CC(float, float, float);
  else if (a_dtype == AGGF_F64 && b_dtype == AGGF_F64 && out_dtype == AGGF_F64) AGGF_CC(double, double, double);
  else if (a_dtype == AGGF_F32 && b_dtype == AGGF_F64 && out_dtype == AGGF_F64) AGGF_CC(float, double, double);
  else if (a_dtype == AGGF_F64 && b_dtype == AGGF_F32 && out_dtype == AGGF_F64) AGGF_CC(double, float, double);
  else return fail(AGGF_ERR_ARG, "aggf_concat_sites: out_dtype must hold the promotion of the inputs");
#undef AGGF_CC
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}

extern "C" int aggf_scale(const void* x, int64_t count, int dtype, double alpha, void* out, void* stream_v) {
  hipStream_t stream = (hipStream_t)stream_v;
  if (!x || !out) return fail(AGGF_ERR_ARG, "aggf_scale: NULL pointer");
  if (count < 0) return fail(AGGF_ERR_ARG, "aggf_scale: negative count");
  if (count == 0) return AGGF_OK;
  if (dtype == AGGF_F64)
    AGGF_LAUNCH(scale_kernel<double>, stream_grid(count), dim3(256), 0, stream, (const double*)x, count, alpha, (double*)out);
  else if (dtype == AGGF_F32)
    AGGF_LAUNCH(scale_kernel<float>, stream_grid(count), dim3(256), 0, stream, (const float*)x, count, (float)alpha, (float*)out);
  else
    return fail(AGGF_ERR_ARG, "aggf_scale: bad dtype");
  AGGF_LAUNCH_OK();
  return AGGF_OK;
}
