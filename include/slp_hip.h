/*
 * slp_hip.h -- C ABI of libslp_hip.so: the MI355X (gfx950) device side of the
 * first-order sparse-LP hot path of PySparseLP (Chambolle-Pock and ADMM behind
 * SparseLP.solve(method=...)).
 *
 * Plain pointers and sizes only; the library copies what it is given to the
 * GPU and never keeps a host pointer after a call returns.  One host thread
 * per handle, one HIP stream per process (slp_init).  Every function that
 * returns int returns 0 on success; a function that returns a handle returns
 * NULL on failure; slp_last_error() then describes the failure (including
 * "no HIP device": there is no CPU fallback anywhere in the library).
 *
 * Reference citations are file:line under the reference tree's pysparselp/.
 * All values are IEEE double, column indices int32, row pointers int64.
 * Matrices are CSR; entries inside a row are used in the order given (the
 * reference never sorts them on this path).
 */
#ifndef SLP_HIP_H
#define SLP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Summation order of the per-row / per-column dot products.
 * SLP_ORDER_SEQUENTIAL: one thread walks a row in storage order, one rounding
 *   per multiply and per add -- bit-identical to scipy's csr_matvec /
 *   csc_matvec as called by the reference.
 * SLP_ORDER_TREE: a row is spread over 2..64 lanes of a wavefront and reduced
 *   with cross-lane shuffles -- same terms, different association (fp64
 *   rounding differences only).
 * SLP_ORDER_AUTO: SEQUENTIAL while the mean row length is <= 16, else TREE. */
enum { SLP_ORDER_AUTO = 0, SLP_ORDER_SEQUENTIAL = 1, SLP_ORDER_TREE = 2 };

/* ---- library / device -------------------------------------------------- */
int slp_version(void);
/* 0 for the shipped library.  Bit 0: a -DSLP_ABLATION build (timing experiments whose kernels give WRONG results by
 * design), bit 1: a kernel-lab variant (`make variant`).  The Python loader refuses a non-zero library unless
 * SLP_LIB_VARIANT names it. */
int slp_build_flags(void);
/* Number of HIP devices visible, or -1 (see slp_last_error). */
int slp_device_count(void);
/* Bind this process to `device` and create the library's stream. */
int slp_init(int device);
int slp_synchronize(void);
/* Device memory the library has freed is kept in a cache (keyed by size) for its next allocation instead of being
 * returned to the driver; slp_trim() returns it all, slp_cached_bytes() tells how much is parked. */
int slp_trim(void);
int64_t slp_cached_bytes(void);
/* What the library's device allocations cost and weigh: out[0] seconds the calling thread spent inside hipMalloc / hipFree
 * (or waiting for a reservation), out[1] the most bytes the library ever held from the driver at once (live + cached),
 * out[2] bytes held now, out[3] driver calls, out[4] seconds of hipMalloc the reservation helper thread absorbed beside
 * other work (slp_matrix_chunked_expect) -- all since the last call with reset != 0 (the peak restarts from what is
 * held).  A set-up's allocation time and peak footprint as the bench line reports them. */
int slp_alloc_stats(double out[5], int reset);
const char *slp_last_error(void);
/* Milliseconds the GPU spent between the two most recent slp_timer_start /
 * slp_timer_stop marks on the library's stream (HIP events). */
int slp_timer_start(void);
int slp_timer_stop(double *ms);

/* ---- sparse matrix: replaces the scipy.sparse.csr_matrix operands ------- *
 * `a * x`  -> scipy csr_matvec (ChambollePockPPD.py:235,240 ; ADMM.py:220,262)
 * `y * a`  -> scipy csc_matvec over a.T (ChambollePockPPD.py:206,216 ;
 *             ADMM.py:95,148).  The transposed copy is built on the device. */
typedef struct slp_matrix slp_matrix;
slp_matrix *slp_matrix_create(int64_t nrow, int64_t ncol, const int64_t *indptr,
                              const int32_t *indices, const double *data);
void slp_matrix_destroy(slp_matrix *a);
int64_t slp_matrix_nnz(const slp_matrix *a);
/* y[nrow] = A x[ncol]   (host vectors in, host vector out) */
int slp_matrix_spmv(slp_matrix *a, const double *x, double *y, int order);
/* out[ncol] = A^T y[nrow] */
int slp_matrix_spmv_t(slp_matrix *a, const double *y, double *out, int order);
/* y = |A|^p x (transposed = 0) or |A|^p^T x (transposed = 1), entry-wise power: the products whose results with a vector
 * of ones are the sums behind Chambolle-Pock's diagonal preconditioners (ChambollePockPPD.py:122-179: np.abs(A) ** p,
 * column sums in row order, row sums in storage order).  Runs on the strip / tall-cell copy of that orientation with its
 * value table raised to the power (what slp_cp_create_on does at set-up); an error when the copy cannot (no copy, or fp64
 * wide strips).  A checker's entry point: lets the set-up sums of a matrix without CSR be verified slice by slice. */
int slp_matrix_spmv_abs_pow(slp_matrix *a, int transposed, double p, const double *x, double *y);
/* Copies of the device arrays: CSR (transposed = 0) or the device-built
 * transposed CSR, i.e. CSC of A (transposed = 1).  Any pointer may be NULL. */
int slp_matrix_download(slp_matrix *a, int transposed, int64_t *indptr,
                        int32_t *indices, double *data);
/* The same for rows row0 .. row0 + count of that orientation (a slice of a matrix too large to copy whole): indptr[count + 1]
 * keeps the ABSOLUTE offsets of the device array (entries of the slice: indptr[count] - indptr[0], starting at indptr[0]);
 * call once for indptr, then with buffers of that size for indices / data. */
int slp_matrix_download_rows(slp_matrix *a, int transposed, int64_t row0, int64_t count, int64_t *indptr,
                             int32_t *indices, double *data);
/* Device-resident benchmark kernels: `reps` back-to-back launches on resident
 * vectors, no host traffic; *ms = average GPU time per launch (HIP events). */
int slp_matrix_bench_spmv(slp_matrix *a, int transposed, int order, int reps, double *ms);
/* Measurement only (no reference counterpart; bench.py's roofline.timed_region): while on, a pair of HIP events is recorded
 * around every single-vector product the solvers run through the strip / tall-cell copies (the products of ADMM.py:148,262 and
 * ChambollePockPPD.py:206,216,235,240), on the stream it runs on.  slp_product_timing(1) resets the record; _read (after
 * switching off) gives out[0] = products, out[1] = the sum of their durations in ms, out[2] = the longest one. */
int slp_product_timing(int on);
int slp_product_timing_read(double out[3]);
/* New device-resident matrix whose row r is scale[r] * (row rows[r] of a): the one-sided stacking [A[up]; -A[lo]] of
 * ChambollePockPPD.py:74-88 (and any row selection) without a round trip of the CSR through the host.
 * scale 1 copies, -1 negates exactly. */
slp_matrix *slp_matrix_gather_rows(slp_matrix *a, int64_t count, const int64_t *rows, const double *scale);
/* Which kernel serves that orientation: 1 = LDS-tiled strip kernel (k_strip_spmv; long sorted rows,
 * >= 3e7 stored entries), 3 / 2 = the same over a value-dictionary copy (at most 2048 distinct stored values):
 * 3 = k_qstrip_spmv (4096-row blocks, 3-byte entries), 2 = k_dstrip_spmv (2048-row blocks, 4-byte entries;
 * SLP_DICT_VARIANT=1), 4 / 5 = wide strips (k_wstrip_spmv: strips of 131072 columns, x gathered from L2; rows too sparse
 * for the LDS tile over a width far beyond an L2) with value-dictionary / fp64 entries -- since round 3 only the fallback of
 * 6 / 7 = tall cells (k_tall_spmv<true / false>, csrc/slp_tall_spmv.hip; format: csrc/slp_tall.hip: row blocks of ~1e4 rows x strips of 4096 columns, running
 * sums and x-tile in LDS, 5-byte value-dictionary items / 4-byte items + fp64 values; 0.05 - 2.5 entries per (row, 4096
 * columns): the per-rank slice of a 1e7-variable LP and its transpose),
 * 0 = row-per-lane-group CSR kernel (k_spmv), -1 = error. */
int slp_matrix_spmv_kernel(slp_matrix *a, int transposed);
/* M = gamma_eq A^T A + gamma_ineq I (N x N CSR, sorted rows, exact zeros dropped) formed on the device: replaces
 * `(gamma_eq * a.T * a + gamma_ineq * eye).tocsr()` of ADMM.py:93-101, i.e. scipy's SMMP csr_matmat, with the same
 * accumulation order per entry (shared rows increasingly, from 0.0) -- bit-identical values. */
slp_matrix *slp_matrix_normal(slp_matrix *a, double gamma_eq, double gamma_ineq);
/* a[:, keep] (keep[ncol] = 0 / 1; entries stay in storage order, columns renumbered) as a new device matrix: the
 * column compaction of SparseLP.remove_fixed_variables (SparseLP.py:632-674).  With shift[ncol] != NULL also
 * a_shift[nrow] = A * shift (csr_matvec order, unreduced matrix) for `b - A * shift` (:646-650); both host vectors. */
slp_matrix *slp_matrix_remove_columns(slp_matrix *a, const unsigned char *keep, const double *shift, double *a_shift);
/* Which derived copies the products may use: 0 (default) = the best the matrix qualifies for; 1 = no value dictionary
 * (fp64 entries in the strips: the general, any-values path); 2 = CSR kernels only.  Frees the strip copies built so
 * far; fails while a solver created on the matrix is alive. */
int slp_matrix_set_format(slp_matrix *a, int policy);
/* Drop the CSR entries (column indices and values, both orientations) of a matrix whose products run on strip copies in
 * both orientations (slp_matrix_spmv_kernel >= 1): afterwards the products, the solvers' iterations AND their periodic
 * reports (slp_cp_report, slp_admm_cg_report: formed through the strip copies) work; downloads, row gathers, format
 * changes and new solver set-ups fail with a clear error.  Refused while a solver created on the matrix iterates on
 * the CSR arrays (Chambolle-Pock in SLP_ORDER_SEQUENTIAL with equality and inequality rows).  At BASELINE config 3
 * this takes the resident matrix data from 66 GB (two CSR orientations + value-dictionary strips) to 17 GB. */
int slp_matrix_release_csr(slp_matrix *a);
/* hipMemGetInfo of the bound device. */
int slp_device_memory(int64_t *free_bytes, int64_t *total_bytes);
/* Bytes of the matrix copy that kernel reads per product (entries + per-strip metadata; CSR: 12 nnz + 8 (rows + 1)),
 * i.e. the matrix part of the HBM traffic one launch must generate; -1 = error. */
int64_t slp_matrix_format_bytes(slp_matrix *a, int transposed);

/* ---- chunked matrix: an LP larger than one CSR copy of itself ------------ *
 * No counterpart in the reference (its operands are whole scipy matrices: ChambollePockPPD.py:206-240, ADMM.py:148,262).
 * The constraint matrix is handed over (slp_matrix_create) or generated (slp_matrix_random) in ROW CHUNKS, in row order;
 * slp_matrix_chunked_append converts a chunk into its product copies for both orientations -- the copy of the chunk's
 * transpose comes straight from the chunk's CSR, no transposed CSR is formed -- keeps the per-row sums the ADMM row
 * scaling needs, releases the chunk's CSR and TAKES OWNERSHIP of the chunk (do not destroy or use it afterwards).  The
 * chunked matrix then serves slp_matrix_spmv / _spmv_t / _bench_spmv / _format_bytes / _spmv_kernel, slp_cp_create_on and
 * slp_admm_cg_create_on* (value-dictionary copies) like any other slp_matrix; whatever needs CSR entries fails with an
 * error.  A x: every chunk writes its rows.  A^T y: chunk k continues the column sums chunk k - 1 left, so every column
 * is one chain of additions in row order -- the unchunked product and scipy's csc_matvec bit for bit, for any chunking.
 * Every chunk but the last needs an even number of rows; a chunk must qualify for strip copies (>= 3e7 entries, sorted
 * rows).  BASELINE config 4 at density 1e-4 (1e7 x 2e7, 2e10 entries: 240 GB of CSR per orientation) is resident on one
 * 288 GB GPU this way (208 GB of tall cells for both orientations). */
slp_matrix *slp_matrix_chunked_create(int64_t ncol);
int slp_matrix_chunked_append(slp_matrix *chunked, slp_matrix *chunk);
/* Optional: how many chunks will be appended in all.  With SLP_RESERVE=1 in the environment every append that is not the last
 * then asks a helper thread to hipMalloc the next chunk's two packet-stream buffers (sizes of the chunk just appended) beside
 * the generation and conversion of that chunk: on boxes whose hipMalloc runs at ~27 ms per GB this hid 3.4 of config 4's
 * 10.4 s of set-up; on quick-malloc boxes it LOST time and it raises the peak footprint by the blocks taken ahead (measured:
 * 279.8 -> 302.9 GB), hence off unless asked for.  Never changes results. */
int slp_matrix_chunked_expect(slp_matrix *chunked, int64_t chunks);
/* The same with the row count of the whole matrix: every chunk then brings its exact SHARE of a multiple of the CU
 * count of tall row blocks (cumulative shares rounded), whatever the chunks' sizes -- e.g. an LP whose equality rows
 * are cut off into chunks of their own (slp_cp_create_on with 0 < m_eq < m).  rows = 0: as slp_matrix_chunked_expect. */
int slp_matrix_chunked_expect_rows(slp_matrix *chunked, int64_t chunks, int64_t rows);
/* Chunks appended so far; 0 for an ordinary matrix, -1 for NULL. */
int64_t slp_matrix_chunks(const slp_matrix *a);
/* Product-kernel launches one y = A x (transposed: A^T y) takes: 1 for an ordinary matrix, up to one per row chunk for a
 * chunked one (bench.py: "per product" vs "per launch" figures). */
int64_t slp_matrix_product_launches(const slp_matrix *a, int transposed);
/* Columns per strip of the orientation's product copy (builds it if need be): 4096 / 2048 / 1024 for tall cells -- the width the
 * build's cost model chose (slp_tall.hip, tall_build) --, the strip formats' widths otherwise; 0 without a copy (CSR kernels), -1 for
 * NULL.  A chunked matrix reports its first chunk's.  Diagnostics: what a density sweep or a test asserts. */
int64_t slp_matrix_strip_width(slp_matrix *a, int transposed);

/* ---- Chambolle-Pock: replaces chambolle_pock_ppd's loop ----------------- *
 * ChambollePockPPD.py:122-179 (preconditioners T, Sigma) and :195-343 (loop).
 * K = [A_eq; A_ineq] stacked by rows (m_eq rows first), b = [b_eq; b_ineq]
 * with the inequalities already one-sided (K_ineq x <= b_ineq, :74-88).
 * State: x = x0 (zeros if NULL), y = 0.  One iteration:
 *   d = (c + A_eq^T y_eq) + A_ineq^T y_ineq ; x+ = clip(x - T d, lb, ub)
 *   z = (1+theta) x+ - theta x ; x = x+
 *   y_eq += S_eq (A_eq z - b_eq) ; y_ineq = max(y_ineq + S_ineq (A_ineq z - b_ineq), 0) */
typedef struct slp_cp slp_cp;
slp_cp *slp_cp_create(int64_t n, int64_t m_eq, int64_t m_ineq, const int64_t *indptr,
                      const int32_t *indices, const double *data, const double *b,
                      const double *c, const double *lb, const double *ub,
                      const double *x0, double alpha, double theta, int order);
void slp_cp_destroy(slp_cp *s);
/* k whole iterations, enqueued without host synchronisation. */
int slp_cp_iterate(slp_cp *s, int64_t k);
/* The two halves of one iteration, for the iterations on which the reference
 * reports (:242-329 sits between the residual :231-240 and the dual update
 * :333-342): primal_step, then slp_cp_report, then dual_step. */
int slp_cp_primal_step(slp_cp *s);
int slp_cp_dual_step(slp_cp *s);
/* How d = (c + y_eq * a_eq) + y_ineq * a_ineq (:206,216) is formed when the LP has
 * both kinds of rows and runs on strip copies: 1 -- two products over copies of
 * A_e^T and A_i^T (the chunks of a chunked matrix cut at m_eq, or copies of the
 * two row ranges built for this solver); 2 -- two products over the copy of the
 * whole K^T with the other kind of rows masked out of y; 0 -- one kind of rows,
 * or the CSR / ELL walk that forms both sums in one pass.  Bit for bit the
 * reference's order in every form. */
int slp_cp_split_form(const slp_cp *s);
/* out[0] energy1 (:248,267,271)  out[1] energy2 (:260-272)
 * out[2] max |A_eq z - b_eq| (:269; 0 without equalities)
 * out[3] max (A_ineq x - b_ineq) (:283; -inf without inequalities)
 * out[4] max |A_eq x - b_eq| (:280; 0 without equalities) */
int slp_cp_report(slp_cp *s, double out[5]);
int slp_cp_get_x(slp_cp *s, double *x);                 /* n  */
int slp_cp_get_y(slp_cp *s, double *y);                 /* m_eq + m_ineq */
int slp_cp_get_preconditioners(slp_cp *s, double *t, double *sigma); /* n, m */
/* Average GPU milliseconds per iteration over `k` iterations (HIP events on
 * the library stream) and per kernel: ms[0] iteration, ms[1] primal kernel
 * (SpMV^T + update), ms[2] dual kernel (SpMV + update). */
int slp_cp_bench(slp_cp *s, int64_t k, double ms[3]);

/* ---- projected Gauss-Seidel: replaces gaussSiedel.pyx ------------------- *
 * boundedGaussSeidelClass.__init__ (gaussSiedel.pyx:87-92) and .solve
 * (:95-153).  The sweep keeps the reference's lexicographic data dependence
 * (row i sees rows < i updated, rows > i not yet) through a level schedule, so
 * x is bit-identical to the sequential sweep. */
typedef struct slp_gs slp_gs;
slp_gs *slp_gs_create(int64_t n, const int64_t *indptr, const int32_t *indices,
                      const double *data);
void slp_gs_destroy(slp_gs *g);
int64_t slp_gs_num_levels(const slp_gs *g);
/* Which sweep the plan chose: 0 one launch per dependency level, 1 one workgroup for the whole (small) system, 2 runs of
 * narrow levels in one workgroup with a register ring (x gathered from memory), 3 the same with the entries classed at plan
 * time (products with values that are final formed chip-wide beforehand, recent results read from an LDS ring).  All four give
 * the sequential sweep's x bit for bit.  SLP_GS_PIPELINED=0/1 and SLP_GS_WINDOW=0 override the choice (tests, timing). */
int slp_gs_sweep_kind(const slp_gs *g);
/* band records of the plan: runs of narrow levels cut into row ranges, one workgroup (compute unit) each; 0 = none
 * (SLP_GS_BANDS=P forces P per run where possible, 0 forbids; default: by the plan's timing model). */
int slp_gs_num_bands(const slp_gs *g);
/* x[n] is updated in place (host buffer), maxiter sweeps, relaxation w;
 * lower/upper may hold -inf/+inf. */
int slp_gs_solve(slp_gs *g, const double *b, const double *lower, const double *upper,
                 double *x, int maxiter, double w);

/* ---- ADMM: replaces lp_admm's loop (ADMM.py:143-268) -------------------- *
 * Inputs are the standard-form, row-normalised problem of ADMM.py:76-101:
 * A (m x N), b, c, lb, ub, x0 (all length N resp. m), M = gamma_eq A^T A +
 * gamma_ineq I (N x N, CSR; m_indptr == NULL: formed on the device by slp_matrix_normal).  One iteration:
 *   y = -c + gamma_eq A^T b + gamma_ineq xp - A^T lambda      (:148)
 *   one projected Gauss-Seidel sweep of M x = y, in place on x (:162)
 *   xp = x (alias, :259) ; lambda += gamma_eq (A x - b)        (:261-263) */
typedef struct slp_admm slp_admm;
slp_admm *slp_admm_create(int64_t N, int64_t m, const int64_t *a_indptr,
                          const int32_t *a_indices, const double *a_data,
                          const double *b, const double *c, const double *lb,
                          const double *ub, const double *x0, const int64_t *m_indptr,
                          const int32_t *m_indices, const double *m_data,
                          double gamma_eq, double gamma_ineq, int order);
/* The same solver from the LP as lp_admm receives it -- the whole setup chain of ADMM.py:73-101 runs on the device:
 * both constraint blocks are uploaded once, row normalisation of each (tools.py:272-290), slack standard form
 * (tools.py:88-127), row normalisation of the stacked system (use_preconditioning), M and A^T b are computed in HBM with
 * the reference's entry and accumulation orders (bit-identical state).  eq_indptr == NULL: no equality block;
 * b_lower / b_upper == NULL: -inf / +inf; x0 == NULL: zeros.  The inequality block is required (tools.py:92). */
slp_admm *slp_admm_create_lp(int64_t n, int64_t m_eq, const int64_t *eq_indptr, const int32_t *eq_indices,
                             const double *eq_data, const double *b_eq, int64_t m_ineq, const int64_t *in_indptr,
                             const int32_t *in_indices, const double *in_data, const double *b_lower,
                             const double *b_upper, const double *c, const double *lb, const double *ub,
                             const double *x0, double gamma_eq, double gamma_ineq, int use_preconditioning, int order);
/* The setup transforms on their own (tests, other callers): rows scaled to unit 2-norm -- new matrix with scipy's entry
 * order for `diags(1/s) * A` (each row reversed, exact zeros dropped), b / b2 (host, may be NULL) scaled in place -- and
 * the slack standard form [[A_eq, 0], [A_ineq, -I]] with rows sorted by column (a_eq may be NULL). */
slp_matrix *slp_matrix_precondition_rows(slp_matrix *a, double *b, double *b2);
slp_matrix *slp_matrix_standard_form(slp_matrix *a_eq, slp_matrix *a_ineq);
void slp_admm_destroy(slp_admm *s);
/* x-step of the iteration; call before the first iteration.
 * 0 (default): one projected Gauss-Seidel sweep, the flags the reference ships (ADMM.py:66-71,:162).
 * 1: the reference's use_unbounded_gauss_siedel branch (ADMM.py:164-181, gaussSiedel.pyx:21-79): one plain
 *    Gauss-Seidel sweep, x = 1.4 x - 0.4 xp, then xp = clip(x + lambda_ineq/g_ineq), lambda_ineq += g_ineq (x - xp)
 *    (:253-256) before the lambda_eq update. */
int slp_admm_set_xstep(slp_admm *s, int mode);
int slp_admm_iterate(slp_admm *s, int64_t k);
/* Halves of one iteration around the reference's report (:213-248). */
int slp_admm_sweep_step(slp_admm *s);
int slp_admm_multiplier_step(slp_admm *s);
/* out[0] augmented-Lagrangian energy (:124-132)  out[1] max |A x - b| (:221)
 * out[2] max(0, -min x) (:222) */
int slp_admm_report(slp_admm *s, double out[3]);
int slp_admm_get_x(slp_admm *s, double *x, int64_t count); /* first `count` entries */
int slp_admm_get_lambda(slp_admm *s, double *lam);          /* m */
int64_t slp_admm_num_levels(const slp_admm *s);
/* band records of M's Gauss-Seidel plan (slp_gs_num_bands) */
int slp_admm_num_bands(const slp_admm *s);
int slp_admm_bench(slp_admm *s, int64_t k, double *ms);

/* ---- ADMM, matrix-free conjugate-gradient x-step ------------------------- *
 * The reference's own alternative x-step (ADMM.py:182-201 with
 * conjugateGradientLinearSolver.py:30-52, selected by its hard-coded flags
 * ADMM.py:66-71: use_cg), with M v evaluated as gamma_eq A^T (A v) +
 * gamma_ineq v so that M is never formed: the only ADMM form that exists at
 * 1e6 x 2e6 (M would be dense).  Per iteration (ten passes over A):
 *   y = -c + g_eq A^T b + g_ineq xp - A^T lambda_eq - lambda_ineq      (:148)
 *   exact line search along the previous displacement                  (:190-194)
 *   one conjugate-gradient step on M x = y                             (:199)
 *   speed = x - xprev ; x = 1.4 x + (1 - 1.4) xp                       (:200-201)
 *   xp = clip(x + lambda_ineq / g_ineq, lb, ub) ; lambda_ineq += g_ineq (x - xp)   (:253-256)
 *   lambda_eq += g_eq (A x - b)                                        (:261-263)
 * slp_admm_cg_create: A, b, c, lb, ub, x0 = the standard-form, row-normalised
 * problem of ADMM.py:76-91 (any mix of equalities and inequalities).
 * slp_admm_cg_create_on: all-inequality LP  A_ineq x <= b_upper  whose matrix is
 * already resident; row normalisation and the slack standard form
 * [A' -I] (tools.py:272-290, :88-127) are applied ON THE DEVICE.  When the
 * matrix runs on value-dictionary strips the two row scalings are kept as a
 * vector and the matrix is left untouched (any number of solvers may share it);
 * otherwise its values are scaled IN PLACE, once: a second ADMM set-up on the
 * same matrix, or one while another solver created on it is alive, fails with
 * an error instead of silently solving another problem.  a_ineq must outlive
 * the solver.
 * With slp_comm_init active the rows (and their slack variables) are this
 * rank's block; the n original variables are replicated. */
typedef struct slp_admm_cg slp_admm_cg;
slp_admm_cg *slp_admm_cg_create(int64_t N, int64_t m, const int64_t *indptr, const int32_t *indices,
                                const double *data, const double *b, const double *c,
                                const double *lb, const double *ub, const double *x0,
                                double gamma_eq, double gamma_ineq, int order);
slp_admm_cg *slp_admm_cg_create_on(slp_matrix *a_ineq, const double *b_upper, const double *c,
                                   const double *lb, const double *ub, double gamma_eq,
                                   double gamma_ineq, int order);
/* Same with the first m_eq rows of the resident matrix being equalities a_i x = b_i (b = [b_eq; b_upper]):
 * they get no slack entry (standard form [A_eq 0; A_ineq -I], tools.py:96-107). */
slp_admm_cg *slp_admm_cg_create_on_mixed(slp_matrix *a, int64_t m_eq, const double *b, const double *c,
                                         const double *lb, const double *ub, double gamma_eq,
                                         double gamma_ineq, int order);
/* The same with two-sided inequality rows  b_lower_i <= a_i x <= b_upper_i  (b_lower may be NULL = all -inf; entries of
 * the m_eq equality rows are ignored): b_lower is scaled with its row like b_upper (tools.py:286-288) and becomes the
 * lower bound of the row's slack variable (tools.py:117-121). */
slp_admm_cg *slp_admm_cg_create_on_two_sided(slp_matrix *a, int64_t m_eq, const double *b_lower, const double *b_upper,
                                             const double *c, const double *lb, const double *ub, double gamma_eq,
                                             double gamma_ineq, int order);
void slp_admm_cg_destroy(slp_admm_cg *s);
/* Products of A per iteration (same mathematics, fp64 rounding differences only); default 0:
 * 0  ten, as the reference writes the iteration;
 * 1  eight: the CG residual y - M(x + step dir) is formed from M x and M dir of the line search (M is linear);
 * 2  six: additionally A^T (g_eq A x + lambda_eq) is one product (y = ... - A^T lambda_eq is never formed);
 * 3  five: additionally A dir of the new direction dir = step dir_old + a_cg r is step (A dir_old) + a_cg (A r)
 *    (both at hand); taken as a product again every 64 iterations so that the recurrence cannot drift.
 * 4  four: additionally M dir = step (M dir_old) + a_cg (M r) (M r is computed for the CG step anyway), so the A^T pass
 *    of the line search carries one vector; refreshed together with A dir.
 * With the strip kernels, A [x, dir] and the two A^T products of the line search each share ONE sweep over the
 * matrix (two-vector pass): 5 sweeps at level 1, 4 at level 2. */
int slp_admm_cg_set_reuse(slp_admm_cg *s, int reuse);
int slp_admm_cg_iterate(slp_admm_cg *s, int64_t k);
/* Halves of one iteration around the reference's report (:213-248). */
int slp_admm_cg_xstep(slp_admm_cg *s);
int slp_admm_cg_multiplier_step(slp_admm_cg *s);
/* out[0] augmented-Lagrangian energy (:124-132)  out[1] max |A x - b|  out[2] max(0, -min x) */
int slp_admm_cg_report(slp_admm_cg *s, double out[3]);
int slp_admm_cg_get_x(slp_admm_cg *s, double *x, int64_t count);

/* ---- ADMM with one copy of the variables per constraint block ------------ *
 * Replaces the loop of lp_admm_block_decomposition (ADMMBlocks.py:264-307) and its per-block sparse LU
 * factorisations (:178-243): the per-block projections onto {A_g x = b_g} are computed matrix-free, all blocks
 * together, by conjugate gradients on A^ A^T (A^ = the constraint matrix with one column per (block, variable)
 * copy, m x P).  The host passes A^ in CSR, `owner[p]` = the variable a copy belongs to, and the copies of every
 * variable in block order as a CSR list (copy_ptr[N+1], copy_idx[P]).  xp0 = the initial consensus variable
 * (clamped to the bounds, :84-86).  Parity with the LU form is a tolerance (slp_blocks_set_cg: relative residual
 * of the projection systems, default 1e-13, at most max_steps CG steps per iteration). */
typedef struct slp_blocks slp_blocks;
slp_blocks *slp_blocks_create(int64_t P, int64_t m, int64_t N, const int64_t *indptr, const int32_t *indices,
                              const double *data, const double *b, const double *c, const double *lb, const double *ub,
                              const double *xp0, const int32_t *owner, const int64_t *copy_ptr, const int32_t *copy_idx,
                              double gamma);
/* One block per rank over a device-resident row block (at scale / multi-GPU): the rows of `a` -- the first m_eq
 * equalities a_i x = b_upper_i, the others b_lower_i <= a_i x <= b_upper_i (b_lower may be NULL = -inf) -- form ONE block
 * of the standard form [A_eq 0; A_ineq -I] (slack column kept implicit).  With slp_comm_init active every rank holds
 * its own block and the consensus sum is one all-reduce of n doubles per iteration; the per-block conjugate-gradient
 * runs need no exchange.  x0 = 0.  `a` stays owned by the caller. */
slp_blocks *slp_blocks_create_on(slp_matrix *a, int64_t m_eq, const double *b_lower, const double *b_upper, const double *c,
                                 const double *lb, const double *ub, double gamma);
/* Several row blocks on one rank (the `blocks` metadata of ADMMBlocks.py at scale): each block its own slp_blocks from
 * slp_blocks_create_on over its own row-block matrix (same n, same gamma; under slp_comm_init every rank creates the
 * same number of blocks).  slp_blocks_group_link once after creating them (the per-variable copy counts become the
 * group's), then slp_blocks_group_iterate: every block's projection -- under slp_comm_init each block's consensus summand
 * is all-reduced on a second stream while the next block's projection computes (count all-reduces of n doubles per
 * iteration, all but the last overlapped) -- and the consensus update in every block.  slp_blocks_get_xp of any member
 * returns the consensus variable. */
int slp_blocks_group_link(slp_blocks **blocks, int count);
int slp_blocks_group_iterate(slp_blocks **blocks, int count, int64_t k);
void slp_blocks_destroy(slp_blocks *s);
int slp_blocks_set_cg(slp_blocks *s, double tol, int max_steps);
/* How well the last block update of a solver from slp_blocks_create_on projected: out[0] = || rhs - S sol ||_2 of its
 * projection system with the operator applied afresh (not the conjugate-gradient recurrence), out[1] = || rhs ||_2; the
 * bar of slp_blocks_set_cg is out[0] <= tol * out[1].  (In the dual form rhs - S nu IS the constraint residual A z - z_s - b
 * of the projected point.)  Costs two products; for tests and diagnostics. */
int slp_blocks_projection_residual(slp_blocks *s, double out[2]);
/* Jacobi (diagonal) preconditioner for the per-block conjugate gradients of a solver from slp_blocks_create_on
 * (1 = on).  Same projection up to the CG tolerance; measured: no fewer steps on the benchmark LPs, whose systems have an
 * essentially constant diagonal -- an option for badly scaled constraint matrices. */
int slp_blocks_set_precond(slp_blocks *s, int jacobi);
int slp_blocks_iterate(slp_blocks *s, int64_t k);
/* out[0] = the augmented-Lagrangian energy of ADMMBlocks.py:246-253, out[1] = CG steps taken so far. */
int slp_blocks_report(slp_blocks *s, double out[2]);
/* Conjugate-gradient steps this rank has taken so far (local: no exchange, unlike slp_blocks_report under slp_comm_init). */
int64_t slp_blocks_cg_steps(const slp_blocks *s);
int slp_blocks_get_xp(slp_blocks *s, double *xp, int64_t count);

/* ---- synthetic random LP on the device (randomLP.py:14-75) -------------- *
 * Row r of A_ineq (global row index row_offset + r): every entry is non-zero
 * with probability `density`, value round(N(0,1)*100)/100, exact zeros
 * dropped (:14-26).  Counter-based generator keyed by (seed, global row), so
 * any row block can be regenerated on any GPU.  Rows come out sorted by
 * column. */
slp_matrix *slp_matrix_random(int64_t nrow, int64_t ncol, double density,
                              uint64_t seed, int64_t row_offset);
/* The vectors of the same LP, keyed by (seed, global index):
 * feasible_x, c, lb, ub of length n (:33,51-55) and, for the local rows of
 * `a`, b_upper = ceil((A feasible_x + |rand_sparse|) * 1000) / 1000 (:43-46).
 * Host output buffers; any may be NULL.  `a` may be a chunked matrix (or one
 * whose CSR was released): A feasible_x is then formed by its product copies,
 * for all rows at once after the last append; row_offset = the global index
 * of its first row. */
int slp_random_lp_vectors(slp_matrix *a, double density, uint64_t seed, int64_t row_offset,
                          double *feasible_x, double *c, double *lb, double *ub,
                          double *b_upper);
/* The 10 %-equality variant of the same LP (randomLP.py:62-68): the first m_eq
 * local rows of `a` are equalities a_i x = b_i with b_eq = A_e feasible_x
 * (:63, no ceiling); the others as above.  The right-hand sides land in
 * b_upper[0 .. m_eq).  m_eq = 0: slp_random_lp_vectors. */
int slp_random_lp_vectors_eq(slp_matrix *a, double density, uint64_t seed, int64_t row_offset,
                             int64_t m_eq, double *feasible_x, double *c, double *lb,
                             double *ub, double *b_upper);
/* Chambolle-Pock state over a device-resident matrix (no host copy of A):
 * all rows are inequalities; takes ownership of nothing (a must outlive it). */
slp_cp *slp_cp_create_on(slp_matrix *a, int64_t m_eq, const double *b, const double *c,
                         const double *lb, const double *ub, const double *x0,
                         double alpha, double theta, int order);

/* ---- multi-GPU: constraint rows partitioned over the GPUs of one node --- *
 * No counterpart in the reference (single process).  Each rank owns a row
 * block of K and the matching slices of b, y, Sigma; x, z, c, T, lb, ub are
 * replicated.  Per iteration one RCCL sum-all-reduce of the n partial column
 * sums K_g^T y_g; the preconditioner T needs one all-reduce at setup; report
 * scalars are reduced with sum/max all-reduces.  RCCL is loaded with dlopen on
 * first use, so single-GPU use does not depend on it. */
int slp_comm_unique_id(char id[128]);
int slp_comm_init(int nranks, int rank, const char id[128]);
/* Test transport: instead of RCCL, every all-reduce copies its buffer to the host, calls
 * fn(buf, count, op, user) -- which must reduce buf[0..count) in place over the ranks (op 0 sum, 1 max) and return 0 -- and
 * copies the result back.  Lets several ranks of the real partitioned device code share one GPU (RCCL refuses two ranks on
 * one device) and lets tests record the sequence of collectives. */
typedef int (*slp_host_allreduce_fn)(double *buf, int64_t count, int op, void *user);
int slp_comm_init_host(int nranks, int rank, slp_host_allreduce_fn fn, void *user);
int slp_comm_finalize(void);
/* In-place all-reduce of a small host vector (op: 0 sum, 1 max) through the
 * device, for bench timing and report scalars. */
int slp_comm_allreduce_host(double *v, int64_t count, int op);
int slp_comm_barrier(void);
/* Preflight of the exchange (tools/rccl_preflight.py; no reference counterpart): `reps` sum-all-reduces of `count`
 * doubles on a device buffer through the path the solvers' exchange takes -- the vector all-reduced once per
 * Chambolle-Pock iteration is A^T y, n doubles (ChambollePockPPD.py:206,216).  The first one is checked (every rank
 * contributes rank + 1).  out[0] = ms per all-reduce (HIP events), out[1] = max |error|, out[2] = ranks. */
int slp_comm_bench_allreduce(int64_t count, int reps, double out[3]);
/* All-reduces this process has issued since slp_comm_init (0 without a communicator): the data-path exchange steps
 * can be counted per iteration (Chambolle-Pock 1, matrix-free ADMM at reuse level 4: 2, block-splitting ADMM 1). */
long long slp_comm_collectives(void);
/* Exchange timing for the bench line (no reference counterpart: the reference is single-process).  slp_comm_timing(1) resets and
 * starts recording: every collective issued from then on is bracketed by a pair of HIP events on the stream it is issued on
 * (events are only recorded -- nothing synchronises inside the timed region); slp_comm_timing(0) stops.  slp_comm_timing_read
 * (after the stop; synchronises) fills out[6] = { collectives recorded, ms inside them, payload bytes (this rank), the slowest
 * single collective in ms, the part of the ms spent on the second stream (block groups: overlapped with the next block's
 * projection), collectives that did not fit the ring of 16384 event pairs }.  Under RCCL a pair brackets the collective's
 * kernel, whose duration includes the wait for the slowest peer.  (The host transport's asynchronous worker -- tests only -- is
 * timed with the host clock around each job.) */
int slp_comm_timing(int on);
int slp_comm_timing_read(double out[6]);
/* The communicator of this process: *nranks / *rank (1 / 0 without one); returns 1 when slp_comm_init* is active, else 0.
 * The host-API solvers (SparseLP.solve -> lp_admm(xstep="cg") / chambolle_pock_ppd) consult it: under a communicator every
 * rank holds the whole LP, hands over only its row block (equal stored entries) and returns the same x. */
int slp_comm_info(int *nranks, int *rank);

#ifdef __cplusplus
}
#endif
#endif /* SLP_HIP_H */
