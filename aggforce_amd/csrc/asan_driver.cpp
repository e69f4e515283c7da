// Host-side sanitizer harness of the C ABI (`make asan`, CPU box, no GPU needed): libaggf's HOST code -- argument
// validation, launch planning, the hand-computed workspace layouts of its 62 entry points -- built with
// -fsanitize=address,undefined (host pass only: the kernels are not compiled) and driven through
//   * every *_workspace_bytes query over a grid of shapes (empty, tiny, ragged, BASELINE-sized, absurd),
//   * every compute entry with NULL pointers and with bad shapes / dtypes: must refuse with an error code,
//   * every compute entry with plausible arguments and a workspace of the queried size: the host logic runs up to the
//     first HIP call, which fails cleanly without a device (the pointers are never dereferenced on the host).
// Exit code 0 = every call returned a documented status and the sanitizers stayed silent.
#include <initializer_list>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/aggf.h"

static int n_calls = 0, n_bad = 0;
static void status(const char* what, int rc, bool must_fail) {
  ++n_calls;
  const bool known = rc == AGGF_OK || rc == AGGF_ERR_ARG || rc == AGGF_ERR_HIP || rc == AGGF_ERR_WORKSPACE || rc == AGGF_ERR_COMM;
  if (!known || (must_fail && rc == AGGF_OK)) {
    ++n_bad;
    printf("UNEXPECTED %s: rc %d (%s)\n", what, rc, aggf_last_error());
  }
}
#define REFUSED(call) status(#call, (call), true)
#define RUNS(call) status(#call, (call), false)

int main() {
  const size_t BUF = (size_t)64 << 20;
  char* raw = (char*)malloc(BUF + 512);
  char* buf = (char*)(((uintptr_t)raw + 255) & ~(uintptr_t)255);  // 256-byte aligned stand-in for every device pointer
  memset(buf, 0, 1 << 20);
  void *p = buf, *ws = buf + (1 << 20);
  double* d = (double*)buf;
  int32_t* i32 = (int32_t*)buf;
  const size_t WS = BUF - (1 << 20);
  printf("version %d\n", aggf_version());
  {
    // launch coverage: sizing call, a buffer that is too small (must stay NUL-terminated and in bounds), reset
    char small[8];
    const size_t need = aggf_coverage_dump(nullptr, 0);
    (void)aggf_coverage_dump(small, sizeof(small));
    if (small[sizeof(small) - 1] != 0 && need >= sizeof(small)) { /* (truncated text still ends inside the buffer) */ }
    RUNS(aggf_coverage_reset());
  }
  int32_t cu = 0;
  size_t fr = 0, tot = 0;
  RUNS(aggf_device_info(&cu, &fr, &tot));

  // ---- workspace queries over a grid of shapes
  const int64_t Ts[] = {0, 1, 7, 64, 1000, 100000, 1000000, (int64_t)1 << 40};
  const int32_t Ns[] = {0, 1, 3, 97, 128, 175, 1024, 4096, 20000};
  size_t sink = 0;
  for (int64_t T : Ts)
    for (int32_t N : Ns) {
      for (int in = 0; in < 2; ++in)
        for (int cd = 0; cd < 2; ++cd)
          for (int g = 0; g < 2; ++g) sink += aggf_gram_workspace_bytes(T, N, N > 3 ? N - N / 3 : N, in, cd, g);
      sink += aggf_linearmap_apply_workspace_bytes(T, N, N / 16 + 1);
      sink += aggf_gram_pair_workspace_bytes(T, N, 128, AGGF_F64);
      sink += aggf_pair_dist_var_workspace_bytes(T, N);
    }
  for (int32_t n : Ns)
    for (int32_t m : {0, 1, 10, 64, 256, 1300}) {
      sink += aggf_eq_qp_workspace_bytes(n, m, m);
      sink += aggf_eq_qp_batched_workspace_bytes(n, m, 1, 32);
      sink += aggf_eq_qp_pinned_workspace_bytes(n, m);
      sink += aggf_augmented_gram_workspace_bytes(n, m);
      sink += aggf_gram_quadform_workspace_bytes(n, m);
    }
  sink += aggf_sumsq_workspace_bytes();
  printf("workspace queries done (checksum %zu)\n", sink);

  // ---- NULL pointers / bad shapes must be refused
  REFUSED(aggf_gram(nullptr, 10, 10, 1, 1, nullptr, nullptr, 10, nullptr, 0, nullptr, 0, nullptr));
  REFUSED(aggf_gram(p, 0, 10, 1, 1, nullptr, nullptr, 10, d, 0, ws, WS, nullptr));
  REFUSED(aggf_gram(p, 10, 10, 7, 1, nullptr, nullptr, 10, d, 0, ws, WS, nullptr));
  REFUSED(aggf_gram(p, 10, 10, 1, 0, nullptr, nullptr, 10, d, 0, ws, WS, nullptr));       // f64 in, f32 products
  REFUSED(aggf_gram(p, 10, 10, 1, 1, i32, nullptr, 10, d, 0, ws, WS, nullptr));            // half a CSR
  REFUSED(aggf_gram(p, 10, 10, 1, 1, nullptr, nullptr, 11, d, 0, ws, WS, nullptr));        // n_red > N
  REFUSED(aggf_gram(p, 10, 10, 1, 1, nullptr, nullptr, 10, d, 0, (char*)ws + 8, WS, nullptr));  // misaligned workspace
  REFUSED(aggf_gram(p, 1000, 4096, 1, 1, nullptr, nullptr, 4096, d, 0, ws, 1024, nullptr));  // workspace too small
  REFUSED(aggf_gram_from_column(p, 100, 256, 1, 1, 256, 100, d, 0, ws, WS, nullptr));      // first_col % 128
  REFUSED(aggf_gram_from_column(p, 100, 256, 1, 1, 256, 128, d, 1, ws, WS, nullptr));      // accumulate with first_col
  REFUSED(aggf_eq_qp_solve(nullptr, 10, 0, nullptr, nullptr, 2, nullptr, 2, 0, 1, nullptr, nullptr, nullptr, 0, nullptr));
  REFUSED(aggf_eq_qp_solve(d, 10, -1.0, nullptr, d, 2, nullptr, 2, 0, 1, d, d, ws, WS, nullptr));
  REFUSED(aggf_eq_qp_solve(d, 10, 0.0, nullptr, d, 2, nullptr, 3, 0, 1, d, d, ws, WS, nullptr));   // B == NULL, nrhs != m
  REFUSED(aggf_eq_qp_solve(d, 10, 0.0, nullptr, d, 2, nullptr, 2, 0, 1, d, d, ws, 64, nullptr));
  REFUSED(aggf_eq_qp_solve_batched(d, 10, 0.0, nullptr, d, 2, nullptr, 2, 0, 1, 0, d, d, ws, WS, nullptr));
  REFUSED(aggf_eq_qp_solve_batched_shift(d, 10, 0.0, nullptr, d, d, i32, 11, 2, nullptr, 2, 0, 1, 2, d, d, ws, WS, nullptr));
  REFUSED(aggf_eq_qp_solve_pinned(d, 10, 0.0, nullptr, i32, 10, d, d, ws, WS, nullptr));  // m >= n
  REFUSED(aggf_eq_qp_solve_pinned(d, 10, 0.0, nullptr, nullptr, 2, d, d, ws, WS, nullptr));
  REFUSED(aggf_sym_pack_upper(nullptr, 10, 1, d, nullptr));
  REFUSED(aggf_sym_unpack_upper(d, 0, 1, d, nullptr));
  REFUSED(aggf_expand_map(nullptr, 2, 5, i32, 10, d, nullptr));
  REFUSED(aggf_linearmap_apply(nullptr, 10, 10, 1, p, 2, 1, 0, 0, p, nullptr, nullptr, ws, WS, nullptr));
  REFUSED(aggf_linearmap_apply(p, 10, 10, 1, p, 2, 1, 5, 0, p, nullptr, nullptr, ws, WS, nullptr));   // nan_mode
  REFUSED(aggf_linearmap_apply(p, 10, 10, 3, p, 2, 1, 0, 0, p, nullptr, nullptr, ws, WS, nullptr));   // dtype
  REFUSED(aggf_linearmap_apply(p, 100, 4096, 1, p, 256, 1, 0, 0, p, d, nullptr, ws, 8, nullptr));     // sumsq workspace
  REFUSED(aggf_slice_gather(p, 10, 10, 1, nullptr, 2, 1, p, nullptr, nullptr));
  REFUSED(aggf_slice_gather(p, 10, 10, 1, i32, 1 << 24, 1, p, nullptr, nullptr));
  REFUSED(aggf_has_nan(nullptr, 10, 1, i32, nullptr));
  REFUSED(aggf_not_close(p, nullptr, 10, 1, 1e-5, 1e-8, i32, nullptr));
  REFUSED(aggf_sumsq(nullptr, 10, 1, d, ws, WS, nullptr));
  REFUSED(aggf_condnormal_augment(p, p, 10, 10, 1, i32, i32, p, 2, 0, p, nullptr, 1, 0, -1.0, 1.0, p, p, nullptr));
  REFUSED(aggf_condnormal_augment(nullptr, p, 10, 10, 1, i32, i32, p, 2, 0, p, nullptr, 1, 0, 1.0, 1.0, p, p, nullptr));
  REFUSED(aggf_condnormal_sites(p, nullptr, 1, 0, 10, 2, 1, 1.0, 1.0, p, p, 0, nullptr));   // f64 sites into f32 outputs
  REFUSED(aggf_gram_pair(p, 100, p, 128, 10, 1, d, 0, ws, WS, nullptr));                     // N % 128
  REFUSED(aggf_gram_pair(p, 128, (char*)p + 8, 128, 10, 1, d, 0, ws, WS, nullptr));          // alignment
  REFUSED(aggf_augmented_gram(d, 10, 2, i32, i32, d, d, ws, WS, nullptr));                   // in place
  REFUSED(aggf_sym_group_reduce(d, 10, i32, i32, 11, d + 1000, nullptr));
  REFUSED(aggf_residual_over_var(p, 1, p, 1, 10, 0.0, p, p, 1, nullptr));
  REFUSED(aggf_residual_over_var(p, 1, p, 1, 10, 1.0, nullptr, nullptr, 1, nullptr));
  REFUSED(aggf_residual_over_var(p, 1, p, 1, 10, 1.0, p, p, 0, nullptr));
  REFUSED(aggf_frames_matmul(p, nullptr, 10, 0, p, 3, nullptr, 1.0, 1, (char*)p + 4096, nullptr));
  REFUSED(aggf_frames_matmul(p, nullptr, 10, 3, p, 3, nullptr, 1.0, 1, p, nullptr));          // out aliases X
  REFUSED(aggf_augment_concat(p, p, 1, p, p, nullptr, 1, 10, 5, 2, 1.0, p, p, nullptr));
  REFUSED(aggf_group_reduce(nullptr, 10, 10, 1, i32, i32, 3, 0, 1, p, nullptr));
  REFUSED(aggf_gb_channels(nullptr, p, 0, 10, 5, 2, 0, (float*)p, 4, p, 8, 1.0, 1e-5, p, p, nullptr));
  REFUSED(aggf_gb_regmat(nullptr, 0, p, p, 0, 10, 5, 2, 0, (float*)p, 5, 4, p, 8, 1.0, 1e-5, 0.6, 128, p, 1, nullptr));
  REFUSED(aggf_gb_distance_range(nullptr, (float*)p, 10, 5, 2, 4, (float*)p, (float*)p, nullptr));
  REFUSED(aggf_gb_regmat_cols(nullptr, 0, p, p, 0, 10, 5, 2, 0, (float*)p, 5, i32, 3, p, 8, 1.0, 1e-5, 0.6, 128, p, 1, nullptr));
  REFUSED(aggf_gb_apply(nullptr, 0, p, p, 0, 10, 5, 2, (float*)p, 5, 4, p, 8, 1.0, 1e-5, d, 37, d, nullptr));
  REFUSED(aggf_gb_apply_cols(nullptr, 0, p, p, 0, 10, 5, 2, (float*)p, 5, d, i32, i32, d, p, 8, 1.0, 1e-5, d, nullptr));
  REFUSED(aggf_trjdot_frames(nullptr, 1, p, 1, 10, 5, 2, nullptr, p, 1, nullptr));
  REFUSED(aggf_feat_contract(nullptr, 0, p, p, 0, 1.0, 10, 5, 7, 128, p, 1, nullptr));
  REFUSED(aggf_feat_constraint_rows(nullptr, 0, 10, 5, 7, (int64_t*)p, 3, d, 2, 0, d, d, nullptr));
  REFUSED(aggf_gb_constraint_rows(nullptr, p, 0, 3, 2, 5, 5, 4, 8, i32, 3, 128, 0, d, d, nullptr));
  REFUSED(aggf_gb_group_overlap(nullptr, 2, 5, d, nullptr));
  REFUSED(aggf_gb_constraint_gram(nullptr, p, 0, 3, 5, 5, 4, 8, i32, 3, 128, d, nullptr));
  REFUSED(aggf_feat_weights(nullptr, 0, 10, 5, 7, d, 10, d, nullptr));
  REFUSED(aggf_pair_dist_var(nullptr, 10, 5, 1, d, ws, WS, nullptr));
  REFUSED(aggf_pair_dist_moments(p, 10, 5, 1, nullptr, d, ws, WS, nullptr));
  REFUSED(aggf_pair_pool_term(nullptr, d, d, 1.0, 10, d, nullptr));
  REFUSED(aggf_gram_quadform(nullptr, 10, d, 2, d, ws, WS, nullptr));
  REFUSED(aggf_daxpby(10, 1.0, nullptr, 1.0, d, d, nullptr));
  REFUSED(aggf_comm_unique_id(nullptr, 128));
  REFUSED(aggf_comm_unique_id(p, 8));
  REFUSED(aggf_comm_init(p, 8, 0, 1, (void**)p));
  REFUSED(aggf_comm_init(p, 128, 3, 2, (void**)p));
  RUNS(aggf_comm_destroy(nullptr));  // (like free(NULL): nothing to do)
  REFUSED(aggf_allreduce_sum(nullptr, 4, 1, nullptr, nullptr));
  REFUSED(aggf_synth_normal(nullptr, 10, 5, 1, 1, 0, 0, 1, 0, nullptr));
  REFUSED(aggf_synth_normal(p, 10, 5, 9, 1, 0, 0, 1, 0, nullptr));

  // ---- plausible calls: planning and workspace arithmetic run; without a device the first HIP call fails cleanly
  struct Shape { int64_t T; int32_t N, n_red, n_cg; };
  const Shape shapes[] = {{500, 6, 6, 2}, {4000, 175, 97, 10}, {5000, 1024, 683, 64}, {3000, 4096, 4096, 256}, {777, 333, 200, 7}};
  for (const Shape& s : shapes) {
    for (int in = 0; in < 2; ++in)
      for (int cd = in; cd < 2; ++cd) {
        const bool groups = s.n_red != s.N;
        const size_t need = aggf_gram_workspace_bytes(s.T, s.N, s.n_red, in, cd, groups);
        if (need <= WS) RUNS(aggf_gram(p, s.T, s.N, in, cd, groups ? i32 : nullptr, groups ? i32 : nullptr, s.n_red, d, 0, ws, need, nullptr));
        if (need <= WS) RUNS(aggf_gram(p, s.T, s.N, in, cd, groups ? i32 : nullptr, groups ? i32 : nullptr, s.n_red, d, 1, ws, need / 2 + 4096 & ~(size_t)255, nullptr));
      }
    const size_t wa = aggf_linearmap_apply_workspace_bytes(s.T, s.N, s.n_cg);
    for (int in = 0; in < 2; ++in)
      for (int od = 0; od < 2; ++od)
        for (int nm = 0; nm < 2; ++nm)
          RUNS(aggf_linearmap_apply(p, s.T, s.N, in, p, s.n_cg, od, nm, -1.0, (char*)p + 4096, d, i32, ws, wa, nullptr));
    RUNS(aggf_slice_gather(p, s.T, s.N, 1, i32, s.n_cg, 1, (char*)p + 4096, i32, nullptr));
    if (s.n_red <= 1100) {
      const size_t w1 = aggf_eq_qp_workspace_bytes(s.n_red, s.n_cg, s.n_cg);
      if (w1 <= WS) RUNS(aggf_eq_qp_solve(d, s.n_red, 0.5, nullptr, d, s.n_cg, nullptr, s.n_cg, 1e-12, 2, d, d, ws, w1, nullptr));
      const size_t w2 = aggf_eq_qp_pinned_workspace_bytes(s.n_red, s.n_cg);
      if (w2 <= WS) RUNS(aggf_eq_qp_solve_pinned(d, s.n_red, 0.0, d, i32, s.n_cg, d, d, ws, w2, nullptr));
      const size_t w3 = aggf_eq_qp_batched_workspace_bytes(s.n_red, s.n_cg, 1, 3);
      if (w3 <= WS) RUNS(aggf_eq_qp_solve_batched(d, s.n_red, 10.0, nullptr, d, s.n_cg, d, 1, 1e-12, 3, 3, d, d, ws, w3, nullptr));
      if (w3 <= WS) RUNS(aggf_eq_qp_solve_batched_shift(d, s.n_red, 10.0, nullptr, d, d, i32, s.n_red / 2, s.n_cg, d, 1, 1e-12, 3, 3, d, d, ws, w3, nullptr));
    }
    RUNS(aggf_sumsq(p, s.T * s.N * 3, 1, d, ws, aggf_sumsq_workspace_bytes(), nullptr));
    RUNS(aggf_condnormal_sites(p, nullptr, 42100, 17, s.T, s.n_cg, 0, 0.01, 0.6955215, (char*)p + 4096, (char*)p + 8192, 1, nullptr));
    RUNS(aggf_condnormal_augment(p, p, s.T, s.N, 1, i32, i32, p, s.n_cg, 0, p, nullptr, 42100, 0, 0.01, 0.6955215, (char*)p + 4096, (char*)p + 8192, nullptr));
    RUNS(aggf_residual_over_var(p, 0, p, 0, s.T * s.n_cg * 3, 0.01, (char*)p + 4096, nullptr, 0, nullptr));
    RUNS(aggf_frames_matmul(p, p, s.T, 3 * s.n_cg, p, 3 * s.n_cg, nullptr, -1.0, 1, (char*)p + 4096, nullptr));
    RUNS(aggf_augment_concat(p, p, 1, p, p, p, 0, s.T, s.N, s.n_cg, 0.6955215, (char*)p + 4096, (char*)p + 8192, nullptr));
    const size_t wp = aggf_pair_dist_var_workspace_bytes(s.T, s.N);
    if (wp <= WS) RUNS(aggf_pair_dist_var(p, s.T, s.N, 1, d, ws, wp, nullptr));
    RUNS(aggf_synth_normal(p, s.T, s.N, 1, 42100, 5, 0.0, 30.0, 1.5, nullptr));
  }
  {
    const size_t wpair = aggf_gram_pair_workspace_bytes(2000, 2048, 128, AGGF_F32);
    if (wpair <= WS) RUNS(aggf_gram_pair(p, 2048, (char*)p + 4096, 128, 2000, AGGF_F32, d, 0, ws, wpair, nullptr));
    const size_t wag = aggf_augmented_gram_workspace_bytes(2048, 128);
    if (wag <= WS) RUNS(aggf_augmented_gram(d, 2048, 128, i32, i32, d, d + 4096, ws, wag, nullptr));
    const size_t wq = aggf_gram_quadform_workspace_bytes(4096, 256);
    if (wq <= WS) RUNS(aggf_gram_quadform(d, 4096, d, 256, d + 4096, ws, wq, nullptr));
    RUNS(aggf_sym_pack_upper(d, 4096, 1, d + 4096, nullptr));
    REFUSED(aggf_sym_unpack_upper(d, 4096, 1, d + 4096, nullptr));  // packed and G overlap
    RUNS(aggf_sym_unpack_upper(d, 64, 1, d + 4096, nullptr));
    RUNS(aggf_expand_map(d, 256, 2731, i32, 4096, d + 4096, nullptr));
    RUNS(aggf_has_nan(p, 1000, 0, i32, nullptr));
    RUNS(aggf_not_close(p, (char*)p + 4096, 1000, 1, 1e-5, 1e-6, i32, nullptr));
    RUNS(aggf_daxpby(1000, 1.0, d, -1.0, d + 1000, d + 2000, nullptr));
  }
  free(raw);
  printf("%d calls, %d unexpected statuses\n", n_calls, n_bad);
  return n_bad ? 1 : 0;
}
