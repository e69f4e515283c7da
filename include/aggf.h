/*
 * libaggf -- C ABI of the MI355X (gfx950) implementation of the aggforce
 * force-map optimisation hot path.
 *
 * The reference (noegroup/aggforce) is pure Python and has no FFI of its own: its
 * boundary for this path is a set of NumPy call sites.  Each entry point below
 * names the reference call site(s) it replaces (paths relative to
 * /root/reference/src/aggforce).  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add at each of those sites.
 *
 * Conventions
 *  - every function returns 0 (AGGF_OK) or a negative AGGF_ERR_* code;
 *    aggf_last_error() returns a thread-local message for the last failure;
 *  - all array pointers are CALLER-OWNED DEVICE pointers (HBM), never retained
 *    or freed by the library; arrays are dense, C-contiguous ("row-major");
 *  - scratch memory is caller-provided: ask aggf_*_workspace_bytes() first;
 *  - `stream` is a hipStream_t passed as void*; calls are asynchronous on it and
 *    never synchronise the device; no hidden global state;
 *  - dtype codes: AGGF_F32 / AGGF_F64.
 */
#ifndef AGGF_H
#define AGGF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGGF_VERSION 100 /* 0.1.0 */

#define AGGF_OK 0
#define AGGF_ERR_ARG (-1)       /* bad argument (shape, dtype, alignment, NULL) */
#define AGGF_ERR_HIP (-2)       /* HIP runtime error */
#define AGGF_ERR_WORKSPACE (-3) /* workspace too small */
#define AGGF_ERR_COMM (-4)      /* RCCL unavailable or a collective failed */

#define AGGF_F32 0
#define AGGF_F64 1

/* nan_mode of aggf_linearmap_apply */
#define AGGF_NAN_PROPAGATE 0 /* plain product (LinearMap(handle_nans=False)) */
#define AGGF_NAN_REPLACE 1   /* NaN inputs are read as `nan_fill` */

int aggf_version(void);
const char* aggf_last_error(void);
/* Launch coverage (no reference counterpart: test infrastructure of the dispatch tables).  Every kernel launch of the
 * library is counted under its template instantiation.  aggf_coverage_dump writes "mangled kernel name <tab> demangled name
 * <tab> launches since the last aggf_coverage_reset <tab> launches since the library was loaded <newline>" for every kernel
 * this process has launched into buf, NUL-terminated, and returns the bytes the full text needs (call with buf_bytes = 0
 * to size the buffer).  A process with AGGF_COVERAGE_FILE set appends the same lines with "label <tab>" in front to that
 * file when the library is unloaded (label = AGGF_COVERAGE_LABEL): tests/test_gpu_zz_coverage.py. */
size_t aggf_coverage_dump(char* buf, size_t buf_bytes);
int aggf_coverage_reset(void);
/* number of compute units of the current device; free/total HBM bytes */
int aggf_device_info(int32_t* cu_count, size_t* free_bytes, size_t* total_bytes);

/* ---------------------------------------------------------------------------
 * K1  Gram / normal matrix of the constraint-reduced forces.
 *
 * Replaces qp/qplinear.py:66-71:  qp_form(forces) (91-103), reshaped_fs @ con_mat
 * (70) and reg_mat.T @ reg_mat (71):
 *     G[j,k] = sum_{t,d} (sum_{a in g_j} F[t,a,d]) (sum_{b in g_k} F[t,b,d])
 * F: (T, N, 3) in `in_dtype`.  Constraint groups are given in CSR form
 * (grp_ptr[n_red+1], grp_atoms[N]) -- column j of the reference's con_mat
 * (make_bond_constraint_matrix, qplinear.py:147-164) has ones at
 * grp_atoms[grp_ptr[j] .. grp_ptr[j+1]); pass NULL,NULL for no constraints
 * (then column j is atom j and n_red <= N: only the first n_red atoms are used, the
 * rest of each frame row is ignored -- row padding).  Products are formed in `compute_dtype`
 * (AGGF_F64 reproduces the reference, whose con_mat is float64 even for float32
 * forces; AGGF_F32 uses fp32 MFMA with partial sums combined in fp64).  float32 F with
 * AGGF_F64 products and no constraint groups is read in place: the operands are widened
 * inside the kernel, no converted copy is made and the workspace holds partial tiles only.
 * F needs the alignment of its element type only (a frame block may start at any row of
 * a larger array) and rows need not be multiples of 16 bytes (odd N); nothing is read
 * beyond F + T*3*N elements.  The launch plan -- and with it the workspace size -- depends
 * on (T, N, n_red, dtypes, has_groups) alone.
 * G: (n_red, n_red) float64, full symmetric matrix; accumulate != 0 adds to it
 * (frame chunks, cross-validation folds).  Partial sums are combined in a fixed
 * order: two runs are bit-identical.
 * ------------------------------------------------------------------------- */
size_t aggf_gram_workspace_bytes(int64_t T, int32_t N, int32_t n_red, int in_dtype,
                                 int compute_dtype, int has_groups);
int aggf_gram(const void* F, int64_t T, int32_t N, int in_dtype, int compute_dtype,
              const int32_t* grp_ptr, const int32_t* grp_atoms, int32_t n_red, double* G,
              int accumulate, void* ws, size_t ws_bytes, void* stream);

/* The same Gram matrix when the caller already holds its leading first_col x first_col block
 * (first_col a multiple of 128): only the 128 x 128 tiles that reach column first_col or beyond
 * are computed and written (both triangles); G[0:first_col, 0:first_col] is left untouched.
 * Used by the featurised fit (featlinearmap.py:361-372 per cg site): the id_feat block of the
 * regression matrix -- the group force sums -- is the same for every site, so its Gram block is
 * formed once and pasted.  No constraint groups here (F is a regression matrix).  Guarantee:
 * with accumulate == 0 every entry outside the leading block is written; the leading block is
 * either left untouched (in-place tile kernel: N % 128 == 0, in_dtype == compute_dtype or
 * float32 frames with float64 products) or overwritten with its own correct values (any other layout computes the
 * whole matrix), so the caller may paste its copy afterwards either way.  accumulate != 0 with
 * first_col > 0 is refused (AGGF_ERR_ARG): the two cases would differ there. */
int aggf_gram_from_column(const void* F, int64_t T, int32_t N, int in_dtype, int compute_dtype,
                          int32_t n_red, int32_t first_col, double* G, int accumulate, void* ws,
                          size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * K2  Equality-constrained least squares shared by all coarse-grained sites.
 *
 * Replaces the solver loop qp/qplinear.py:76-86 (and featlinearmap.py:370-381):
 * for every right-hand side i
 *     x_i = argmin 1/2 x'(G + l2*diag(l2_diag)) x   s.t.  A x = B[:, i]
 * (what qpsolvers.solve_qp(P, q=0, A, b) is asked for).  One scaled, shifted
 * Cholesky factorisation P~ = P/s + A'A serves all right-hand sides; the Schur
 * complement A P~^-1 A' is factorised the same way; one refinement step on the
 * constraint residual follows.
 * G: (n, n) float64 (not modified).  l2_diag: n float64 or NULL (= ones).
 * A: (m, n) float64.  B: (m, nrhs) float64 or NULL (= identity, nrhs must equal m).
 * X: (nrhs, n) float64, row i = x_i.
 * schur_reg: 0 for full-row-rank A; > 0 adds schur_reg * trace (>= the largest eigenvalue) to the Schur
 * complement so that redundant (consistent) constraint rows -- the sampled rows of
 * featlinearmap._constr_arrays -- can be factorised; the n_refine refinement steps
 * x -= P~^-1 A' S^-1 (A x - b) then remove the bias of that shift.
 * stats (device, 4 doubles): [0] 0 if ok, else 1-based index of the first
 * non-positive pivot (k for P~, n+k for the Schur complement); [1] max |A x - b|
 * after refinement; [2] max |A x - b| before refinement; [3] scale s.
 * ------------------------------------------------------------------------- */
size_t aggf_eq_qp_workspace_bytes(int32_t n, int32_t m, int32_t nrhs);
int aggf_eq_qp_solve(const double* G, int32_t n, double l2, const double* l2_diag,
                     const double* A, int32_t m, const double* B, int32_t nrhs,
                     double schur_reg, int32_t n_refine, double* X, double* stats, void* ws,
                     size_t ws_bytes, void* stream);

/* The same solve for n_problems INDEPENDENT problems of identical shape at once -- the
 * per-site loop of the featurised fit, featlinearmap.py:349-384, where every cg site has
 * its own P and A: problem p reads G + p*n*n, A + p*m*n, B + p*m*nrhs (B == NULL:
 * identity for all), writes X + p*nrhs*n and stats + 4*p; l2 and l2_diag are shared.
 * Every step of the factorisation is ONE launch over all problems (the problem index is
 * a grid dimension), so the chain of small dependent kernels that bounds a single solve
 * is paid once per batch instead of once per site.  Results are bit-identical to
 * n_problems separate aggf_eq_qp_solve calls. */
size_t aggf_eq_qp_batched_workspace_bytes(int32_t n, int32_t m, int32_t nrhs, int32_t n_problems);
int aggf_eq_qp_solve_batched(const double* G, int32_t n, double l2, const double* l2_diag,
                             const double* A, int32_t m, const double* B, int32_t nrhs,
                             double schur_reg, int32_t n_refine, int32_t n_problems, double* X,
                             double* stats, void* ws, size_t ws_bytes, void* stream);

/* aggf_eq_qp_solve_batched with the positive shift A'A of every problem formed by the CALLER (AtA: n x n per
 * problem, lower triangle read) -- for constraint rows whose structure makes it cheap (aggf_gb_constraint_gram:
 * S multiply-adds per entry instead of m = S*n_cg).  Same workspace, same results up to the rounding of A'A.
 * perm (n int32 per problem, or NULL): the factorisation takes the variables in the order perm[0], perm[1], ...
 * (G, l2_diag, A, AtA and X stay in the caller's order).  a_first_col (needs perm; 0 = unknown): in that order the
 * columns of A before a_first_col are zero in EVERY problem -- sparse constraint rows with the variables they touch
 * moved to the end; the forward solve L^-1 A', the Schur complement and the products with A then start at the 256-row
 * block that holds a_first_col (the fused featurised fit with a slice coordinate map: 576 of ~2100-3000 rows). */
int aggf_eq_qp_solve_batched_shift(const double* G, int32_t n, double l2, const double* l2_diag,
                                   const double* A, const double* AtA, const int32_t* perm, int32_t a_first_col,
                                   int32_t m, const double* B, int32_t nrhs, double schur_reg, int32_t n_refine,
                                   int32_t n_problems, double* X, double* stats, void* ws, size_t ws_bytes,
                                   void* stream);

/* The same problem when every row of A is a unit vector with the 1 at pin_idx[i] (all distinct) and B is the
 * identity -- the constraint rows of a slice coordinate map, `coord_map.standard_matrix @ con_mat` of
 * qplinear.py:82 for every configuration of the reference's tests: the constraints pin m variables,
 * x_i[pin_idx[j]] = delta_ij, and the rest follows from ONE factorisation of the free block,
 * x_f = -P_ff^-1 P[f, pin_idx[i]] -- no A'A product, no Schur complement, no refinement.
 * pin_idx: m int32 (device).  X: (m, n).  stats as above ([1] = [2] = 0: the constraints hold exactly;
 * [0] = -1 if pin_idx holds an index twice or outside 0..n-1: then no element of G or X is addressed through a pin
 * and X comes back as zeros). */
size_t aggf_eq_qp_pinned_workspace_bytes(int32_t n, int32_t m);
int aggf_eq_qp_solve_pinned(const double* G, int32_t n, double l2, const double* l2_diag,
                            const int32_t* pin_idx, int32_t m, double* X, double* stats, void* ws,
                            size_t ws_bytes, void* stream);

/* Packed upper triangle of `batch` symmetric n x n float64 matrices, row-major:
 *   packed[b][i n - i (i - 1) / 2 + (j - i)] = G[b][i][j], j >= i   (n (n + 1) / 2 elements per matrix).
 * The payload of the Gram all-reduce: the reference has no distributed code; the build sums the per-rank Gram
 * matrices of reg_mat.T @ reg_mat (qplinear.py:71, featlinearmap.py:359) -- symmetric, so half the bytes travel.
 * aggf_sym_unpack_upper writes both triangles (exactly symmetric). */
int aggf_sym_pack_upper(const double* G, int32_t n, int32_t batch, double* packed, void* stream);
int aggf_sym_unpack_upper(const double* packed, int32_t n, int32_t batch, double* G, void* stream);

/* W[i, a] = X[i, group_of_atom[a]]  -- `con_mat @ gen_map`, qp/qplinear.py:86.
 * X: (n_rows, n_red) float64; group_of_atom: N int32; W: (n_rows, N) float64. */
int aggf_expand_map(const double* X, int32_t n_rows, int32_t n_red,
                    const int32_t* group_of_atom, int32_t N, double* W, void* stream);

/* ---------------------------------------------------------------------------
 * K3  LinearMap apply:  out[t,c,d] = sum_a M[c,a] * P[t,a,d]
 *
 * Replaces util.trjdot (util.py:119-125) as called by LinearMap.__call__
 * (map/core.py:219-240).  P: (T, N, 3) in `in_dtype`; M: (n_cg, N) and out:
 * (T, n_cg, 3) in `out_dtype` (NumPy's promotion of points and matrix).
 * nan_mode AGGF_NAN_REPLACE reads NaN inputs as nan_fill (the reference's two
 * passes with NaN->0 and NaN->-1, map/core.py:226-229).  If sumsq != NULL the
 * sum of squares of `out` is written there (float64; agg.force_smoothness,
 * agg.py:297, is sumsq / (3 T n_cg)), combined in a fixed order.  If nan_seen != NULL,
 * nan_seen[0] is set to 1 when P contains a NaN (the _has_nans scan of map/core.py:13-16
 * fused into the pass that reads P anyway; the caller zeroes it).  The flag is conservative: it
 * is never left at 0 when P holds a NaN, but the float64 LDS-DMA tile kernel scans its OUTPUT
 * (0 x NaN = NaN reaches every site of that frame and component), so an infinity of P that meets a
 * zero coefficient sets it too -- a caller that needs the exact input property follows a set flag
 * with aggf_has_nan.
 * ------------------------------------------------------------------------- */
size_t aggf_linearmap_apply_workspace_bytes(int64_t T, int32_t N, int32_t n_cg);
int aggf_linearmap_apply(const void* P, int64_t T, int32_t N, int in_dtype, const void* M,
                         int32_t n_cg, int out_dtype, int nan_mode, double nan_fill, void* out,
                         double* sumsq, int32_t* nan_seen, void* ws, size_t ws_bytes, void* stream);

/* K3b  one-hot rows (slice maps): out[t,c,:] = P[t, idx[c], :], converted to
 * out_dtype.  Same call site as K3 when every row of M is a unit vector.  0 <= idx[c] < N (not
 * checked on the device).  nan_seen (may be NULL; zeroed by the caller): set to 1 when a GATHERED
 * value is NaN -- for a slice map exactly the case in which the reference's NaN -> 0 and NaN -> -1
 * products differ (map/core.py:226-236). */
int aggf_slice_gather(const void* P, int64_t T, int32_t N, int in_dtype, const int32_t* idx,
                      int32_t n_cg, int out_dtype, void* out, int32_t* nan_seen, void* stream);

/* flag[0] = 1 if any element of x is NaN (map/core.py:13-16 _has_nans); flag must be
 * zeroed by the caller. */
int aggf_has_nan(const void* x, int64_t count, int dtype, int32_t* flag, void* stream);
/* flag[0] = 1 unless |a-b| <= atol + rtol*|b| everywhere (np.allclose, map/core.py:230-232);
 * flag must be zeroed by the caller. */
int aggf_not_close(const void* a, const void* b, int64_t count, int dtype, double rtol,
                   double atol, int32_t* flag, void* stream);
/* out[0] = sum of squares of x in float64, fixed summation order (agg.py:297).
 * ws: at least aggf_sumsq_workspace_bytes() bytes. */
size_t aggf_sumsq_workspace_bytes(void);
int aggf_sumsq(const void* x, int64_t count, int dtype, double* out, void* ws, size_t ws_bytes,
               void* stream);

/* ---------------------------------------------------------------------------
 * K5  Gaussian augmentation of a trajectory (noised maps).
 *
 * Replaces JCondNormal.sample / .log_gradient with cov = var*I and a LinearMap
 * premap (trajectory/jaxgausstraj.py:213-284, closed form of the autodiff) and
 * AugmentedTrajectory._augment (trajectory/core.py:382-390):
 *     y = M x + sqrt(var) eps,  r = (y - M x)/var,
 *     out_coords = [x ; y],  out_forces = [F + kbt M' r ; -kbt r]   (T, N+n_cg, 3)
 * coords/forces: (T, N, 3) in traj_dtype; the premap M (n_cg, N) is given by its columns in
 * compressed form -- atom a contributes M[mt_idx[j], a] = mt_val[j] for j in
 * [mt_ptr[a], mt_ptr[a+1]) -- (mt_val in aug_dtype); mean = M x: (T, n_cg, 3) and
 * noise: (T, n_cg, 3) in aug_dtype (the augmenter's dtype, float32 by default in
 * the reference); noise == NULL draws eps from Philox4x32-10 keyed by
 * (seed, frame_offset + t, site, dim), independent of sharding.  Outputs are in
 * NumPy's promotion of the two dtypes.  `mean` comes from aggf_linearmap_apply /
 * aggf_slice_gather with out_dtype = aug_dtype.
 * ------------------------------------------------------------------------- */
int aggf_condnormal_augment(const void* coords, const void* forces, int64_t T, int32_t N,
                            int traj_dtype, const int32_t* mt_ptr, const int32_t* mt_idx,
                            const void* mt_val, int32_t n_cg, int aug_dtype,
                            const void* mean, const void* noise, uint64_t seed,
                            int64_t frame_offset, double var, double kbt, void* out_coords,
                            void* out_forces, void* stream);

/* The noised maps without the extended trajectory (qp/jgauss.py:114-131 runs qp_linear_map on the
 * (T, N + n_cg, 3) arrays that AugmentedTrajectory._augment concatenates, trajectory/core.py:382-390).
 * The extended forces are [F - Fa C | Fa] with Fa = -kbt r the generated sites' forces (T, n_cg, 3) and
 * C (n_cg x N) the premap (or premap x source_postmap'), so their Gram matrix is Tm' Gx Tm with Gx the
 * Gram matrix of the plain concatenation [F | Fa] and Tm = [[I, 0], [-C, I]]:
 *   aggf_condnormal_sites  y (generated coordinates) and Fa, (T, n_cg, 3) each, by the same expressions and
 *                          the same Philox stream as aggf_condnormal_augment (out_dtype: aug_dtype or AGGF_F64);
 *   aggf_gram_pair         Gx ((N + N2)^2 float64) of [F | F2] read where they lie: same dtype for both arrays
 *                          and the products, N % 128 == 0, N2 % 128 == 0, 16-byte aligned (otherwise
 *                          AGGF_ERR_ARG: concatenate and call aggf_gram); workspace: aggf_gram_pair_workspace_bytes;
 *   aggf_augmented_gram    G_aug = Tm' Gx Tm (exactly symmetric); C as compressed columns: for atom a the entries
 *                          c_idx[k], c_val[k], k in [c_ptr[a], c_ptr[a+1]);
 *   aggf_sym_group_reduce  C' G C for the 0/1 constraint matrix of qplinear.py:147-164 given as CSR groups
 *                          (what aggf_gram's fused group sums give when it reads the trajectory itself).
 * The map is applied the same way: W [F - Fa C | Fa] = W_N F + (W_a - W_N C') Fa -- two aggf_linearmap_apply
 * calls and an aggf_daxpby, mapped coordinates = y. */
int aggf_condnormal_sites(const void* mean, const void* noise, uint64_t seed, int64_t frame_offset, int64_t T,
                          int32_t n_cg, int aug_dtype, double var, double kbt, void* out_y, void* out_f,
                          int out_dtype, void* stream);
size_t aggf_gram_pair_workspace_bytes(int64_t T, int32_t N, int32_t N2, int dtype);
int aggf_gram_pair(const void* F, int32_t N, const void* F2, int32_t N2, int64_t T, int dtype, double* G,
                   int accumulate, void* ws, size_t ws_bytes, void* stream);
size_t aggf_augmented_gram_workspace_bytes(int32_t N, int32_t n2);
int aggf_augmented_gram(const double* Gx, int32_t N, int32_t n2, const int32_t* c_ptr, const int32_t* c_idx,
                        const double* c_val, double* G_aug, void* ws, size_t ws_bytes, void* stream);
int aggf_sym_group_reduce(const double* G, int32_t n, const int32_t* grp_ptr, const int32_t* grp_atoms,
                          int32_t n_red, double* G_red, void* stream);

/* The GENERAL Augmenter protocol (any sample / log_gradient pair; trajectory/core.py:382-390, trajectory/augment.py)
 * and the parts of JCondNormal's interface that the fused calls above do not cover (trajectory/jaxgausstraj.py):
 *   aggf_residual_over_var  r = (gen - mean) / var into out_pos and -r into out_neg (either may be NULL): the two
 *                           log-gradients of a scalar-covariance conditional normal (jaxgausstraj.py:251-284 in closed
 *                           form, simplegausstraj.py:100-113); out_dtype = NumPy promotion of the two input dtypes
 *                           (or float64);
 *   aggf_frames_matmul      out[t, j] = add[t, j] + alpha sum_k (X[t, k] - S[t, k]) B[j, k] on flattened frames
 *                           (T, K) -> (T, J), B (J, K) row-major, S and add optional (NULL), one dtype throughout:
 *                           the products of a FULL (3 n x 3 n) covariance -- y = mean + eps L' with L its Cholesky
 *                           factor (jaxgausstraj.py:291-329) and Sigma^-1 (y - mean) (jaxgausstraj.py:77-96);
 *   aggf_augment_concat     out_coords = [coords ; gen], out_forces = [forces + kbt corr ; kbt lgrad], (T, N + n_aug,
 *                           3) each in the NumPy promotion of traj_dtype and aug_dtype (trajectory/core.py:384-390);
 *                           gen, lgrad: (T, n_aug, 3), corr: (T, N, 3), all three in aug_dtype. */
int aggf_residual_over_var(const void* gen, int gen_dtype, const void* mean, int mean_dtype, int64_t count, double var,
                           void* out_pos, void* out_neg, int out_dtype, void* stream);
int aggf_frames_matmul(const void* X, const void* S, int64_t T, int32_t K, const void* B, int32_t J, const void* add,
                       double alpha, int dtype, void* out, void* stream);
int aggf_augment_concat(const void* coords, const void* forces, int traj_dtype, const void* gen, const void* corr,
                        const void* lgrad, int aug_dtype, int64_t T, int32_t N, int32_t n_aug, double kbt,
                        void* out_coords, void* out_forces, void* stream);

/* ---------------------------------------------------------------------------
 * K4  Gaussian-basis distance featuriser (gb_feat) and the featurised regression
 *     matrix, without the one-hot (T, N, n_feat) feature tensor.
 *
 * Replaces qp/jaxfeat.py:20-567 (gb_feat, gaussian_dist_basis, clipped_gauss,
 * channel_allocate, gb_subfeat, gb_subfeat_jac; the JAX jacrev is replaced by its
 * closed form), map/tools.py:63-104 (smear_map product) and the dense einsum of
 * qp/featlinearmap.py:361-369 / 512-520.  "Groups" are the constraint groups in
 * id_feat label order (featlinearmap.py:598-609): group g = feature channel g.
 * All atoms of a group share the group-mean position, hence one distance
 * r[t,ch] = |Pg[t,ch] - cg[t,site]| and one Gaussian row
 * g_k(r) = max(exp(-((r - centers[k])/width)^2), clip) - clip.
 * g_dtype = arithmetic type of positions, distances and Gaussians (Pg, cg, centers, gauss,
 * grad are arrays of that type): AGGF_F32 is the reference's (JAX default float32) and the
 * default of the Python layer; AGGF_F64 (gb_feat(feature_dtype=np.float64)) evaluates the
 * same expressions in float64.  Products with the forces are formed in the NumPy-promoted
 * type of (f_dtype, g_dtype).  `sizes` (group sizes) is float32 either way.
 * ------------------------------------------------------------------------- */
/* out[t,g,:] = sum (mean != 0: mean) over the atoms of group g of X[t,a,:];
 * X: (T, N, 3) in in_dtype; out: (T, n_groups, 3) in out_dtype; CSR groups. */
int aggf_group_reduce(const void* X, int64_t T, int32_t N, int in_dtype, const int32_t* grp_ptr,
                      const int32_t* grp_atoms, int32_t n_groups, int mean, int out_dtype,
                      void* out, void* stream);
/* compact features of cg site `site`: gauss (T, n_ch, n_basis) and
 * grad (T, n_ch, n_basis, 3) = |ch| g_k'(r) (Pg - cg)/r  (the per-channel divergence,
 * jaxfeat.py:544-565).  Pg: (T, G, 3) group means, cg: (T, n_cg, 3), sizes: G. */
int aggf_gb_channels(const void* Pg, const void* cg, int g_dtype, int64_t T, int32_t G, int32_t n_cg,
                     int32_t site, const float* sizes, int32_t n_ch, const void* centers,
                     int32_t n_basis, double width, double clip, void* gauss, void* grad,
                     void* stream);
/* R3 (T, ld_feat, 3) in out_dtype (the product type, or AGGF_F64: float32 products widened on
 * store, which with ld_feat % 128 == 0 is aggf_gram's in-place operand for float64 products;
 * float64 features always give AGGF_F64), the regression matrix of featlinearmap.py:361-369 in the
 * layout aggf_gram consumes: columns [0, n_id) = Fg (id_feat block, n_id = 0 or G), then
 * R3[t, n_id + ch*n_basis + k, d] = g_k(r) Fg[t,ch,d] + kbt |ch| g_k'(r) u_d for ch < n_ch.
 * Fg: (T, G, 3) group force sums in f_dtype.  Columns beyond n_id + n_ch*n_basis are not
 * written (pass n_red = that count to aggf_gram, which ignores the rest). */
int aggf_gb_regmat(const void* Fg, int f_dtype, const void* Pg, const void* cg, int g_dtype, int64_t T,
                   int32_t G, int32_t n_cg, int32_t site, const float* sizes, int32_t n_id,
                   int32_t n_ch, const void* centers, int32_t n_basis, double width, double clip,
                   double kbt, int32_t ld_feat, void* R3, int out_dtype, void* stream);
/* Column compaction of the fused fit (no reference counterpart: the reference multiplies the
 * zeros).  A clipped Gaussian column (ch, k) of gb_feat (jaxfeat.py:272-276) is identically
 * zero over the trajectory if channel ch never comes within width*sqrt(ln(1/clip)) of centre
 * c_k; it then contributes nothing to P = R'R nor to the constraint rows, its coefficient in
 * the minimiser of featlinearmap.py:370-381 is exactly 0 (l2 > 0), and it can be dropped.
 * aggf_gb_distance_range (always float32 positions: a superset test with a safety margin):
 *   rmin/rmax (n_cg, G) float32, caller-initialised to +inf / 0, receive
 *   the min / max over frames of |Pg[t,ch] - cg[t,site]| for ch < n_ch (atomic min/max, so
 *   several calls -- frame chunks, ranks -- accumulate).
 * aggf_gb_regmat_cols: aggf_gb_regmat restricted to the Gaussian columns cols[j] = ch*n_basis+k
 *   (j < n_cols), stored compactly: R3[t, n_id + j, :]. */
int aggf_gb_distance_range(const float* Pg, const float* cg, int64_t T, int32_t G, int32_t n_cg,
                           int32_t n_ch, float* rmin, float* rmax, void* stream);
int aggf_gb_regmat_cols(const void* Fg, int f_dtype, const void* Pg, const void* cg, int g_dtype, int64_t T,
                        int32_t G, int32_t n_cg, int32_t site, const float* sizes, int32_t n_id,
                        const int32_t* cols, int32_t n_cols, const void* centers, int32_t n_basis,
                        double width, double clip, double kbt, int32_t ld_feat, void* R3,
                        int out_dtype, void* stream);
/* out (T, n_cg, 3) float64: application of the feature-linear force map
 * (featlinearmap.py:512-520 + map/core.py:428-430) for all sites;
 * coef: (n_cg, n_feat) float64, n_feat = n_id + n_ch*n_basis. */
int aggf_gb_apply(const void* Fg, int f_dtype, const void* Pg, const void* cg, int g_dtype, int64_t T,
                  int32_t G, int32_t n_cg, const float* sizes, int32_t n_id, int32_t n_ch,
                  const void* centers, int32_t n_basis, double width, double clip,
                  const double* coef, int32_t n_feat, double* out, void* stream);
/* The same application from the NON-ZERO Gaussian coefficients only (a fit with a cut-off basis leaves most of them
 * exactly zero, and different ones for every channel): site c owns entries col_ptr[c] .. col_ptr[c+1]-1 of
 * col_idx (= ch*n_basis + k) / col_val; coef_id (n_cg, n_id) is the id block, dense (NULL if n_id == 0).
 * One lane per kept column instead of one per channel: no idle lanes. */
int aggf_gb_apply_cols(const void* Fg, int f_dtype, const void* Pg, const void* cg, int g_dtype, int64_t T,
                       int32_t G, int32_t n_cg, const float* sizes, int32_t n_id, const double* coef_id,
                       const int32_t* col_ptr, const int32_t* col_idx, const double* col_val,
                       const void* centers, int32_t n_basis, double width, double clip, double* out,
                       void* stream);

/* ---------------------------------------------------------------------------
 * K3c  trjdot with a per-frame (3-D) factor.
 *
 * Replaces util.trjdot's second branch, np.einsum("...fd,...cf->...cd") (util.py:119-125),
 * which CLAMap.__call__ (map/core.py:428-430) uses to apply a configuration-dependent map:
 *     out[t,c,d] = sum_f factor[t,c,f] * points[t,f,d]  (+ trans[t,c,d] if trans != NULL)
 * points (T, N, 3) in p_dtype, factor (T, n_cg, N) in f_dtype, out and trans (T, n_cg, 3)
 * in out_dtype = the promoted dtype (float64 if either input is).  HBM-bound: factor is
 * read exactly once.
 * ------------------------------------------------------------------------- */
int aggf_trjdot_frames(const void* points, int p_dtype, const void* factor, int f_dtype, int64_t T,
                       int32_t N, int32_t n_cg, const void* trans, void* out, int out_dtype, void* stream);

/* ---------------------------------------------------------------------------
 * K4b/K4c  Dense-featuriser contractions of qp_feat_linear_map (any featuriser that
 * follows the reference's protocol: feats (T, N, n_feat), divs (T, n_feat, 3) per site).
 *
 * aggf_feat_contract: featlinearmap.py:361-369 (alpha = kbt) and the force + divergence
 *   term of the map's application, featlinearmap.py:512-520 (alpha = 1):
 *     out[t,f,d] = sum_a feat[t,a,f] * F[t,a,d] + alpha * div[t,f,d]      (div may be NULL)
 *   out is (T, ld, 3), ld >= n_feat; columns n_feat..ld-1 are written as zeros, so a
 *   multiple of 128 gives aggf_gram its in-place layout.  feat and div share x_dtype;
 *   out_dtype = promoted dtype of forces and features.
 * aggf_feat_constraint_rows: _constr_arrays (featlinearmap.py:445-459) for one cg site:
 *     A[(s,c),f] = sum_a M[c,a] * feat[frame_idx[s],a,f],   b[(s,c)] = (c == site)
 *   feat (T, N, n_feat); frame_idx[S] int64 device array; M (n_cg, N) float64;
 *   A (S*n_cg, n_feat), b (S*n_cg) float64.
 * aggf_gb_constraint_rows: the same rows for the fused [id_feat | gb_feat] features:
 *     A[(s,c), g]        = Mg[c,g]                        g < n_id
 *     A[(s,c), n_id+j]   = Mg[c,ch] * gauss[s,ch,k]       j-th Gaussian column = (ch, k)
 *   Mg (n_cg, G) float64 = coordinate map summed over each constraint group, gauss
 *   (S, n_ch, nb) in g_dtype from aggf_gb_channels on the sampled frames.  cols == NULL: all
 *   n_ch*nb Gaussian columns in (ch, k) order; else cols[j] = ch*nb + k for the n_cols
 *   columns kept by the compacted fit (see aggf_gb_distance_range).  A has row stride
 *   ld >= n_id + n_cols; columns beyond are zero.
 * aggf_gb_group_overlap / aggf_gb_constraint_gram: A'A of those rows (the positive shift of the solve,
 *   aggf_eq_qp_solve_batched_shift) from their structure instead of a product over the S*n_cg rows:
 *     (A'A)[f,f'] = M2[g(f), g(f')] * sum_s w_s(f) w_s(f'),   M2 = Mg' Mg  (G x G, aggf_gb_group_overlap),
 *   g(f) = f for an id column, ch for a Gaussian column; w_s(f) = 1 resp. gauss[s,ch,k].  AtA is (ld, ld),
 *   ld >= n_id + n_cols; the LOWER triangle is written (whole 64 x 64 tiles of it), zeros beyond the columns.
 * aggf_feat_weights: scale_f of _feat_linear_mapping (featlinearmap.py:512-515):
 *     w[t*ld_t + a] = sum_f feat[t,a,f] * coef[f]     (ld_t >= N lets the caller stack sites)
 * ------------------------------------------------------------------------- */
int aggf_feat_contract(const void* forces, int f_dtype, const void* feat, const void* div, int x_dtype,
                       double alpha, int64_t T, int32_t N, int32_t n_feat, int32_t ld, void* out,
                       int out_dtype, void* stream);
int aggf_feat_constraint_rows(const void* feat, int x_dtype, int64_t T, int32_t N, int32_t n_feat,
                              const int64_t* frame_idx, int32_t S, const double* M, int32_t n_cg,
                              int32_t site, double* A, double* b, void* stream);
int aggf_gb_constraint_rows(const double* Mg, const void* gauss, int g_dtype, int32_t S, int32_t n_cg, int32_t G,
                            int32_t n_id, int32_t n_ch, int32_t n_basis, const int32_t* cols,
                            int32_t n_cols, int32_t ld, int32_t site, double* A, double* b, void* stream);
int aggf_gb_group_overlap(const double* Mg, int32_t n_cg, int32_t G, double* M2, void* stream);
int aggf_gb_constraint_gram(const double* M2, const void* gauss, int g_dtype, int32_t S, int32_t G, int32_t n_id,
                            int32_t n_ch, int32_t n_basis, const int32_t* cols, int32_t n_cols, int32_t ld,
                            double* AtA, void* stream);
int aggf_feat_weights(const void* feat, int x_dtype, int64_t T, int32_t N, int32_t n_feat,
                      const double* coef, int64_t ld_t, double* w, void* stream);

/* ---------------------------------------------------------------------------
 * K6  Pair-distance fluctuations for guess_pairwise_constraints.
 *
 * Replaces constraints/constfinder.py:46-53 (util.distances, util.py:65-72, then
 * np.var over frames): var[i,j] = Var_t |x_j(t) - x_i(t)|  (population variance) for all
 * pairs, in one streaming pass with shifted sums (no (T, N, N) tensor).
 * X: (T, N, 3) in dtype; var: (N, N) float64, symmetric, zero diagonal.
 * ------------------------------------------------------------------------- */
size_t aggf_pair_dist_var_workspace_bytes(int64_t T, int32_t N);
int aggf_pair_dist_var(const void* X, int64_t T, int32_t N, int dtype, double* var, void* ws,
                       size_t ws_bytes, void* stream);
/* The same pass with the mean distance as a second output (mean, var: (N, N) float64): what a
 * frame-sharded caller needs to combine the per-rank statistics exactly (Chan et al.):
 *     n = sum n_r,  mean = sum n_r mean_r / n,  var = sum n_r (var_r + (mean_r - mean)^2) / n
 * (aggforce_amd.guess_pairwise_constraints(..., comm=): two all-reduces of (N, N)).  Same workspace
 * as aggf_pair_dist_var. */
int aggf_pair_dist_moments(const void* X, int64_t T, int32_t N, int dtype, double* mean, double* var,
                           void* ws, size_t ws_bytes, void* stream);
/* out[e] = weight * (var_r[e] + (mean_r[e] - mean[e])^2) on n float64 elements: one rank's term of the
 * pooled variance above (weight = n_r / n; out may alias var_r). */
int aggf_pair_pool_term(const double* var_r, const double* mean_r, const double* mean, double weight,
                        int64_t n, double* out, void* stream);

/* ---------------------------------------------------------------------------
 * Gram algebra for cross-validation.  Replaces, inside project_forces_grid_cv
 * (agg.py:142-235), the n_folds x |grid| repeated passes over the training and
 * validation frames: G is additive over frames, so per-fold Grams are formed once
 * (aggf_gram), the training Gram is total - fold (aggf_daxpby), and the hold-out
 * score of a map with reduced coefficients X is
 *     mean((W F_val)^2) = sum_i x_i' G_fold x_i / (3 T_val n_cg)   (agg.py:224-227,291-297).
 * aggf_gram_quadform: q[i] = x_i' G x_i for the m rows of X (m, n); G (n, n) float64.
 * aggf_daxpby: out = a*x + b*y on n float64 elements (out may alias x or y).
 * ------------------------------------------------------------------------- */
size_t aggf_gram_quadform_workspace_bytes(int32_t n, int32_t m);
int aggf_gram_quadform(const double* G, int32_t n, const double* X, int32_t m, double* q, void* ws,
                       size_t ws_bytes, void* stream);
int aggf_daxpby(int64_t n, double a, const double* x, double b, const double* y, double* out,
                void* stream);

/* ---------------------------------------------------------------------------
 * C1  The path's one collective: sum over the GPUs of a node of the per-shard Gram
 * matrices (and of the residual's two scalars).  No reference counterpart (the reference is
 * single-process); the objective of qplinear.py:66-77 is a sum over frames, so frames shard
 * over ranks, every rank calls aggf_gram on its shard, ONE all-reduce combines G, and every
 * rank runs the identical aggf_eq_qp_solve (replicated, no broadcast).  RCCL over xGMI,
 * loaded on first use (librccl.so.1).  One process per GPU: rank 0 calls
 * aggf_comm_unique_id and passes the 128 bytes to the other ranks by any host channel;
 * every rank calls aggf_comm_init with the same id (collective), then
 *     aggf_gram(...);  aggf_allreduce_sum(G, n_red*n_red, AGGF_F64, comm, stream);  aggf_eq_qp_solve(...)
 * in place, asynchronous on `stream`.  (The Python host does the same through
 * torch.distributed, backend "nccl" = RCCL: aggforce_amd/distributed.py.)
 * ------------------------------------------------------------------------- */
#define AGGF_COMM_ID_BYTES 128
int aggf_comm_unique_id(void* id_out, size_t id_bytes);
int aggf_comm_init(const void* id, size_t id_bytes, int32_t rank, int32_t world, void** comm_out);
int aggf_comm_destroy(void* comm);
int aggf_allreduce_sum(void* buf, int64_t count, int dtype, void* comm, void* stream);

/* Frame-sized housekeeping of the host layer (the reference does it in NumPy on (n_frames, n_sites, 3) arrays):
 *   aggf_take_frames   out[i, :] = src[idx[i], :], whole frames of row_elems elements each (fold / sample selection:
 *                      agg.py:208-231 `coords[train_inds]`, featlinearmap.py:447-452); idx: int64 on the device, every
 *                      entry in [0, n_src) (the host validates; an entry outside gives a NaN row, never a read outside src);
 *   aggf_concat_sites  out[t] = [a[t] ; b[t]] along the site axis (np.concatenate(axis=1) of trajectory/core.py:388-390,
 *                      map/tmap.py:430-436), a: (T, Na, 3), b: (T, Nb, 3), out_dtype = the NumPy promotion of the two (anything else is refused);
 *   aggf_scale         out = alpha x, elementwise in `dtype` (map/tmap.py:399-401 `fill_value * coords`: a product,
 *                      so NaN and 0 x inf behave as in NumPy). */
int aggf_take_frames(const void* src, int64_t n_src, int64_t row_elems, int dtype, const int64_t* idx, int64_t n,
                     void* out, void* stream);
int aggf_concat_sites(const void* a, int32_t Na, int a_dtype, const void* b, int32_t Nb, int b_dtype, int64_t T,
                      void* out, int out_dtype, void* stream);
int aggf_scale(const void* x, int64_t count, int dtype, double alpha, void* out, void* stream);

/* ---------------------------------------------------------------------------
 * Synthetic trajectories for benchmarks and full-size property tests (no
 * reference counterpart).  out[t,a,d] = mean + sigma * z(seed, frame_offset+t, a, d)
 * with z a counter-based standard normal (Philox4x32-10 + Box-Muller), so any
 * sharding of the frame axis reproduces the same data.  If lattice != 0 the mean
 * of atom a is its position on a cubic lattice with that spacing (coordinates).
 * ------------------------------------------------------------------------- */
int aggf_synth_normal(void* out, int64_t T, int32_t N, int dtype, uint64_t seed,
                      int64_t frame_offset, double mean, double sigma, double lattice,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AGGF_H */
