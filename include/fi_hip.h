/* fi_hip.h -- C ABI of the MI355X-native field_interpolation solver core (libfi_hip.so).
 *
 * This is the drop-in boundary: plain C, opaque context, POD structs, caller-owned buffers, integer
 * status codes, no exceptions, no aborts.  Everything above it (the C++ headers in
 * include/field_interpolation/, the ctypes mirror in field_interpolation_amd/) is host plumbing;
 * everything below it is hand-written HIP for gfx950.
 *
 * Each entry point names the reference interface it replaces (file:line under the reference tree,
 * emilk/field_interpolation).
 *
 * Conventions
 *   - all calls are synchronous: they return after the GPU work they started has completed;
 *   - `memory` says where caller buffers live: FI_HOST (malloc'ed) or FI_DEVICE (HBM pointers of the
 *     current device, e.g. torch tensor data_ptr(); the caller must have finished writing them);
 *   - positions / normals are interleaved xyzxyz... fp32 in LATTICE coordinates, exactly like
 *     field_interpolation.hpp:153-173;
 *   - a context is not thread safe; different contexts may be used from different threads.
 *   - return value 0 = FI_OK; otherwise fi_last_error() holds a message for the calling thread.
 */
#ifndef FI_HIP_H
#define FI_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define FI_OK 0
#define FI_ERR_INVALID 1      /* bad argument (the reference CHECK_F / ABORT_F cases) */
#define FI_ERR_HIP 2          /* HIP runtime failure */
#define FI_ERR_STATE 3        /* call order violated (e.g. solve before assemble) */
#define FI_ERR_COMM 4         /* RCCL failure */
#define FI_ERR_UNSUPPORTED 5  /* valid in the reference, not available in this mode */
#define FI_ERR_BREAKDOWN 6    /* solver breakdown (non-finite or non-positive curvature): reference returns {} */
#define FI_ERR_TIMEOUT 7      /* the wall-clock guard of a solve (FI_SOLVE_TIMEOUT_S, default 600 s) stopped it; over slabs
                                 all ranks stop in the same round.  The iterate is kept in the context. */

#define FI_HOST 0
#define FI_DEVICE 1

#define FI_F32 0 /* all vectors and operator coefficients fp32, reductions fp64 */
#define FI_F64 1 /* everything fp64 */

#define FI_MAX_DIM 3 /* field_interpolation.hpp:44 */

/* field_interpolation.hpp:47-59 */
#define FI_VALUE_NEAREST_NEIGHBOR 0
#define FI_VALUE_LINEAR_INTERPOLATION 1
#define FI_GRADIENT_NEAREST_NEIGHBOR 0
#define FI_GRADIENT_CELL_EDGES 1
#define FI_GRADIENT_LINEAR_INTERPOLATION 2

typedef struct fi_ctx fi_ctx;

/* field_interpolation.hpp:75-95 `Weights`, field for field. */
typedef struct fi_weights {
	float data_pos;
	float data_gradient;
	float model_0, model_1, model_2, model_3, model_4;
	float gradient_smoothness;
	int   value_kernel;
	int   gradient_kernel;
} fi_weights;

/* sparse_linear.hpp:66-73 `SolveOptions`. */
typedef struct fi_solve_options {
	int   tile;
	int   tile_size;
	int   cg;
	int   max_iterations;
	float error_tolerance;
} fi_solve_options;

/* sparse_linear.hpp:8-15 `Triplet` (12 bytes). */
typedef struct fi_triplet {
	int   row, col;
	float value;
} fi_triplet;

/* Counters and timings of the last assemble / solve (the reference logs these through loguru:
 * sparse_linear.cpp:122-125,208-209,438-439). */
typedef struct fi_stats {
	long   num_unknowns;       /* owned lattice points of this rank */
	long   num_data_rows;      /* data rows accepted (value + gradient rows inside the lattice) */
	long   num_cells;          /* distinct lattice cells holding data */
	long   num_generic_rows;   /* rows held in generic (COO) form */
	int    iterations;         /* CG iterations of the last solve */
	int    converged;          /* 1: ||r|| <= tol*||Atb|| reached; 0: stopped by max_iterations.  Under FI_OPT_FIELD_TOLERANCE: 1 when the
	                              field estimate is within the tolerance (or an fp64 recurrence has reached its floor: as converged as
	                              the arithmetic allows); 0 also for an fp32 solve that ends at ITS floor with the estimate above it */
	double rel_residual;       /* recurrence ||r||/||Atb|| at exit */
	double assemble_ms;        /* GPU time of the last fi_assemble */
	double solve_ms;           /* GPU time of the last solve */
	double spmv_ms_avg;        /* mean duration of the AtA-apply launches sampled with HIP events */
	int    spmv_samples;
	double spmv_bytes;         /* algorithmic bytes of one AtA apply */
	int    restarts;           /* residual replacements: b - A x was evaluated this many times at convergence */
	double verified_residual;  /* ||b - A x||/||b|| at the last such evaluation (-1: none) */
	int    num_levels;         /* 1 + coarser levels built by the last fi_assemble */
	int    coarse_iterations;  /* CG iterations spent on coarser levels by the last solve (cascade start); a level solved without
	                              a look at its stop flag (same problem shape and tolerance as the solve before) counts what its
	                              flag said afterwards */
	double prec_ms_avg;        /* mean duration of the sampled Chebyshev-step launches of the polynomial preconditioner */
	int    prec_samples;
	double prec_bytes;         /* mean algorithmic bytes of the sampled launches (every step of the sampled polynomials: 2.5 / 3.5 / 4.5 lattice passes) */
	int    operator_applies;   /* full operator applications + preconditioner steps of the last solve (finest level) */
	int    halo_exchanges;     /* slabs: halo exchanges of the finest level during the last solve (polynomial PCG) */
	int    reductions;         /* slabs: dot-product reductions across the ranks during the last solve (polynomial PCG) */
	int    coarse_unconverged; /* levels of the coarse-to-fine start that had NOT met their tolerance after the iterations their
	                              previous solve had needed (no look at the flag in between): the finest level then started from a
	                              poorer guess and still converged to ITS tolerance; those levels watch their flag again next time */
	double field_estimate;     /* FI_OPT_FIELD_TOLERANCE: bound on ||x - x*||_inf / ||x||_inf of the returned field (-1: the
	                              residual rule ran) */
	double field_per_residual; /* ... change of the field (relative, maximum norm) per unit of relative residual dropped over the
	                              last iteration */
	double stop_residual;      /* the relative residual the last solve ended at */
	int    field_rounds;       /* 1: the last solve stopped by the field; 0: by the residual */
} fi_stats;

const char* fi_last_error(void);
/* Number of visible HIP devices (does not initialise a context). */
int fi_device_count(int* count);

/* ---- context --------------------------------------------------------------------------------
 * Replaces `LatticeField{sizes}` (field_interpolation.hpp:97-111): x (sizes[0]) is the fastest axis.
 * The context lives on the current HIP device. */
int fi_ctx_create(fi_ctx** out, int ndim, const int* sizes, int dtype);

/* Slab-decomposed context for one rank of `nranks` (one process per GPU): the slowest axis
 * (sizes[ndim-1]) is split into contiguous slabs; `sizes` are the GLOBAL lattice sizes.  Every rank
 * passes every data point (or at least those within one cell of its slab); each rank keeps the cells
 * that touch its slab.  Follow with fi_comm_init before fi_assemble. */
int fi_ctx_create_slab(fi_ctx** out, int ndim, const int* sizes, int dtype, int rank, int nranks);
int fi_ctx_destroy(fi_ctx* ctx);

/* Owned range [lo, hi) of the slowest axis for this rank. */
int fi_slab_range(const fi_ctx* ctx, int* lo, int* hi);
/* Data points this rank has to be given: those whose coordinate z along the slowest axis lies in [*lo, *hi).  The
 * range covers every cell that touches the slab on the finest level AND on each of the FI_OPT_LEVELS coarser levels
 * set so far (a coarse cell of level l spans 2^l fine planes, and the coarse replicas are assembled from the rank's
 * own points): call it after fi_set_option(FI_OPT_LEVELS).  Points outside the range are never needed; points inside
 * it that touch no owned plane are dropped by the library. */
int fi_slab_point_range(const fi_ctx* ctx, float* lo, float* hi);
/* The partition rule itself (pure host arithmetic, no GPU): planes [lo, hi) of `planes` for `rank`. */
int fi_slab_partition(int planes, int rank, int nranks, int* lo, int* hi);
/* Ghost planes a slab keeps on each side: the reach of the widest enabled model stencil (model_k -> k),
 * at least 1 (cell blocks and gradient_smoothness reach one plane). */
int fi_halo_width(const fi_weights* w, int* width);

/* Diagnostic: runs the RCCL call pattern of the slab exchange (grouped ncclSend/ncclRecv on a stream, in-place
 * fp64 all-reduce) on a ONE-rank communicator of `device` and checks the data -- the part of the multi-GPU path
 * a single-GPU machine can execute against the real library. */
int fi_comm_self_test(int device, long count);

/* RCCL bootstrap (no reference counterpart: the reference is single-process).  Rank 0 calls
 * fi_comm_unique_id, the 128 bytes are broadcast by the launcher (torch.distributed), every rank calls
 * fi_comm_init.  Halo planes of the CG search direction and the dot products then travel over xGMI. */
int fi_comm_unique_id(void* out128);
int fi_comm_init(fi_ctx* ctx, const void* unique_id128);
/* What the transport of a slab context looks like from the inside (a bench line can then show that RCCL really carried N
 * ranks): out[0] = ranks the communicator counts (ncclCommCount; the test transport: its segment's; 0: no transport),
 * out[1] = this rank's index in it (ncclCommUserRank), out[2] = the HIP device the communicator is bound to
 * (ncclCommCuDevice), out[3] = 1 RCCL / 2 host-staged test transport / 0 none, out[4] = ghost planes one exchange moves
 * to each neighbour (the stencil reach), out[5] = ghost planes stored (the polynomial's deep exchange moves that many
 * once per polynomial), out[6] = bytes of one lattice plane in the context's precision. */
int fi_comm_info(const fi_ctx* ctx, long out[7]);
/* TEST transport for ranks that share one GPU (RCCL refuses two ranks on one device): the halo planes and the dot
 * products travel through the POSIX shared-memory segment `name` ("/...") by host copies, behind the same two internal
 * operations (exchange_halo, allreduce_sum) the RCCL path implements.  Rank 0 passes create = 1 BEFORE the other ranks
 * attach (the launcher orders the calls).  Everything above the wire -- slab geometry, per-rank assembly, the rank-set
 * solvers, bench.py --gpus N -- then runs for real on a single-GPU machine.  Slow by construction; never the product path. */
int fi_comm_init_host(fi_ctx* ctx, const char* name, int create);

/* ---- assembly -------------------------------------------------------------------------------
 * fi_set_model replaces add_field_constraints(field, weights) (field_interpolation.cpp:326-341):
 * the model_0..model_4 and gradient_smoothness rows of add_model_constraint (:243-316) are never
 * materialised; the operator applies them matrix-free.  Only the model_* and gradient_smoothness
 * fields of `w` are read here. */
int fi_set_model(fi_ctx* ctx, const fi_weights* w);

/* Replaces add_points (field_interpolation.cpp:343-371) and, with n == 1, add_value_constraint (:57-80),
 * add_value_constraint_nearest_neighbor (:82-107) and add_gradient_constraint (:123-240).
 *   per point i:  w_i = point_weights ? point_weights[i] : 1
 *     value row    with weight w_i*value_weight  and target values ? values[i] : 0
 *     gradient rows with weight w_i*gradient_weight  when normals != NULL
 * A zero weight skips the rows, points outside the lattice are ignored, exactly as the reference.
 * FI_VALUE_NEAREST_NEIGHBOR requires normals (reference CHECK_NOTNULL_F, :361) -> FI_ERR_INVALID.
 * May be called several times; rows accumulate until fi_assemble. */
int fi_add_points(fi_ctx* ctx, long n, const float* positions, const float* normals, const float* point_weights,
                  const float* values, float value_weight, int value_kernel, float gradient_weight,
                  int gradient_kernel, int memory);

/* The border prior of the reference's SDF application (src/sdf_field.cpp:218-246, generate_sdf_field with
 * boundary_weight > 0): every lattice point on the border of the lattice gets the row [1] * weight = d * weight, d = its
 * distance to the nearest data point given to fi_add_points so far (fp32: sum of squared coordinate differences, min,
 * sqrt -- the reference's arithmetic).  Brute force O(border x points) like the reference, on the device; the rows join the
 * data rows (and the coarser levels).  Call it after fi_add_points, before fi_assemble.  weight == 0 adds nothing. */
int fi_add_border_prior(fi_ctx* ctx, float weight);

/* Generic rows: replaces handing an arbitrary `LinearEquation` (sparse_linear.hpp:18-22) to the solvers,
 * as src/bipolar_2d.cpp:177-302 and src/line_2d.cpp:49-104 do.  Duplicate (row, col) entries are summed
 * (sparse_linear.hpp:43).  Row indices are local to this call (0..nrows-1). */
int fi_add_rows_coo(fi_ctx* ctx, long nrows, long ntriplets, const fi_triplet* triplets, const float* rhs,
                    int memory);

/* Replaces as_sparse_matrix_float + make_square + A^T*b (sparse_linear.cpp:59-70,105-113,120): bins the
 * data rows by lattice cell, accumulates the per-cell A^T A blocks, A^T b and diag(A^T A) on the GPU.
 * Returns when everything is ENQUEUED on the context's stream (sizes the host needs have been read back on the way): the
 * next call on the context -- normally fi_solve_cg -- queues behind it.  fi_stats.assemble_ms is the time between two events
 * around the call's device work; fi_get_stats waits for the second one. */
int fi_assemble(fi_ctx* ctx);

/* Drops all data rows (model weights are kept). */
int fi_clear_points(fi_ctx* ctx);

/* ---- solve ----------------------------------------------------------------------------------
 * Replaces solve_sparse_linear_with_guess (sparse_linear.cpp:186-212) and the CG phase of
 * solve_tiled_with_guess (:427-440).  Jacobi-preconditioned conjugate gradients on A^T A x = A^T b
 * (the reference runs Eigen::BiCGSTAB with the same diagonal preconditioner and the same stop rule
 * ||r||_2 <= tol * ||A^T b||_2).  max_iterations <= 0: 2*N (Eigen default); tol <= 0: fp32 epsilon.
 * guess == NULL means zeros.  `guess`/`out` hold the owned unknowns of this rank (fp32). */
int fi_solve_cg(fi_ctx* ctx, const float* guess, int max_iterations, float tol, float* out, int* iterations,
                float* rel_residual, int memory);

/* Solver options.  FI_OPT_VERIFY_RESIDUAL (default 1): when the recurrence residual meets the tolerance,
 * evaluate b - A x; if it misses the tolerance (fp32 drift) restart CG from it, at most 3 times.  0 gives
 * the reference's stop rule on the recurrence residual alone.
 * FI_OPT_LEVELS (default 0): coarser replicas of the problem (lattice halved per level, weights rescaled,
 * the same data points) built by fi_assemble; with guess == NULL, fi_solve_cg then starts from a coarse-to-fine
 * cascade (each level solved to FI_OPT_COARSE_TOLERANCE, default 1e-3, and interpolated to the next) -- the
 * reference's recipe for large lattices (src/sdf_field.cpp:272-288, README.md "My resolution is huge"),
 * generalised to several levels and kept on the device. */
#define FI_OPT_VERIFY_RESIDUAL 1
#define FI_OPT_LEVELS 2
#define FI_OPT_COARSE_TOLERANCE 3
/* FI_OPT_MULTIGRID (default 0): with levels, precondition CG with one V-cycle over them (Chebyshev-Jacobi
 * smoothing, R = P^T, coarse operators re-assembled from the same points) instead of the Jacobi diagonal. */
#define FI_OPT_MULTIGRID 4
/* FI_OPT_MIXED_PRECISION (default 0; FI_F64 contexts, with FI_OPT_LEVELS and FI_OPT_MULTIGRID): fi_assemble also
 * builds an fp32 replica of the problem and gives IT the levels; CG (x, r, p, the operator apply, every dot
 * product and the stop test) stays in fp64, the V-cycle preconditioner -- most of the HBM traffic of an iteration
 * -- runs on the replica in fp32 (z = s V32(r / s), s = ||r||/||b||).  Same iteration counts and answers as the
 * pure fp64 solve, about 0.7x the time.  The SDF configurations need fp64 residuals: in fp32 alone b - A x stalls
 * near 1e-4 (kappa ~ side^4).  With guess == NULL the coarse-to-fine start also runs on the replica. */
#define FI_OPT_MIXED_PRECISION 5
/* FI_OPT_POLY_TERMS (default 0 = the Jacobi diagonal, Eigen's DiagonalPreconditioner as in sparse_linear.cpp:199): with
 * d >= 2, CG is preconditioned by a Chebyshev polynomial of d terms in Dinv (A_model + diag(A_data)) over the interval
 * [hi / FI_OPT_POLY_RATIO, hi] (default ratio 10; hi = 1.1 x the largest eigenvalue of the Jacobi-scaled model operator,
 * found once per model by the power method).  Same stop rule, same answers; about the same number of operator
 * applications as Jacobi-PCG but d - 1 of every d run as ONE 5-pass launch without cell records, and dot products
 * are reduced twice per d applications.  3-D lattices with model_0 / model_1 / model_2; other contexts ignore it. */
#define FI_OPT_POLY_TERMS 6
#define FI_OPT_POLY_RATIO 7
/* FI_OPT_MG_SMOOTHER (default 1): the V-cycle's smoother on fp32 3-D levels with model_0 / model_1 / model_2.
 * 1: the polynomial of FI_OPT_POLY_TERMS' kind in A_model + f diag(A_data) -- plain-stencil launches without cell
 * records, two full applies per level and cycle; f = FI_OPT_MG_SAFE_FACTOR (default 4; 2^D makes that operator a bound of
 * the full one, so the smoother converges for any data).  0: the Chebyshev polynomial in the full operator on every
 * level (the only smoother of 2-D / fp64 levels).  Both give symmetric positive definite preconditioners. */
#define FI_OPT_MG_SMOOTHER 8
#define FI_OPT_MG_SAFE_FACTOR 9
/* FI_OPT_MG_TERMS (default 5, 2..16) and FI_OPT_MG_RATIO (default 30, (1, 1000]): the polynomial smoother's number of
 * terms (terms - 1 launches of the plain marching kernel per smoothing pass) and the ratio of its interval [hi / ratio, hi].
 * Measured on config 4 (fp64 CG + fp32 V-cycle to a field within 1e-5; profiles/r4_ablation.md): 4 terms / ratio 10 (the
 * setting of round 3) 7 iterations at 256^3 and 8 at 512^3, 5 / 30 five and six -- each cycle 8 % dearer, the solve 20 %
 * cheaper. */
#define FI_OPT_MG_TERMS 10
#define FI_OPT_MG_RATIO 11
/* FI_OPT_FIELD_TOLERANCE (default 0 = off; V-cycle PCG; over slabs up to 16 of them -- their maxima travel with the r . r
 * sum of the iteration's all-reduce, every rank decides on the same numbers): stop by the FIELD, not by the residual --
 * the north-star's accuracy "values within 1e-5 of the CPU reference's double solve" (sparse_linear.cpp:154-184) is a statement
 * about x, and what a residual buys in x varies with the lattice, the data and the weights by three orders of magnitude
 * (kappa ~ side^4).  x* - x_k is the sum of the steps still to come; every iteration the solver holds the last step's size,
 * ||x_k - x_(k-1)||_inf = |alpha| ||p||_inf (a by-product of the pass that updates x), and the history of the residual norms.
 * If the steps shrink by sigma per iteration, ||e_k||_inf <= ||x_k - x_(k-1)||_inf sigma / (1 - sigma).  sigma is the SLOWEST
 * mean decay of the residual norm over the last 1, 2, 4, 8, 16 iterations and over the whole solve, the step the largest of
 * the last nine carried forward at that rate (a slowly converging solve has lucky single drops; its late, faster phases
 * are not the tail's rate; and CG on an ill-conditioned system converges in stairs -- a lull of several iterations with tiny
 * steps and a falling residual while the error stands still, then the next stair); where every one of those windows gains
 * more than a factor 3 per iteration -- a healthy V-cycle -- the last step and its own ratio ||r_k|| / ||r_(k-1)|| are used.
 * The solve ends when twice that (a margin for the smooth modes, which converge last) is within the tolerance times
 * ||x_k||_inf; no estimate is formed while the residual falls by less than 5 % per iteration (the residual floor then ends
 * the solve) and no stop before the third iteration (CG's first steps remove the rough part of the error: small steps, a
 * falling residual, the smooth part not yet moved) -- the eighth behind a caller's guess, whose error may be smooth from the start.  The `tol` of fi_solve_cg is ignored (the precision's floor stands in:
 * 1e-13 in fp64, 2e-7 in fp32 -- an fp32 solve that ends there with the estimate above the tolerance reports converged = 0);
 * no constant depends on the workload.  An estimate, not a bound: over 200 random 3-D, 150 random 2-D and 100 random fp32
 * problems (tests/stress_field_rule.py: value data and oriented points, 1 to 5 levels, 6 to 650 iterations) the true error
 * exceeded the tolerance in 3, 7 and 0 cases (warm starts included; 3 and 9 of 150 + 150 with FI_OPT_MG_KCYCLE), by at most 1.5 x and 2.4 x --
 * but for warm starts on hierarchies whose cold solves take a thousand iterations (2 cases, 15 x); the goldens of configs 2 to 5 end 8 to 100 x
 * below it.  fi_stats: field_estimate, field_per_residual. */
#define FI_OPT_FIELD_TOLERANCE 12
/* FI_OPT_MG_KCYCLE (default 0 = off; V-cycle PCG on an undivided lattice, fp32 levels): a K-cycle -- the correction of the
 * first `value` coarse levels is not one application of the coarser level's cycle but TWO steps of flexible CG on that
 * level's system, each preconditioned by the level's cycle (Notay & Vassilevski: the lengths of the two corrections come from
 * a line search in the energy norm, so a level whose own cycle overcorrects -- the re-discretised coarse levels of
 * oriented-point data do, profiles/r6_ablation.md section 11 -- cannot make the preconditioner indefinite the way a W-cycle
 * does).  The preconditioner then depends on its argument: the outer CG takes the flexible beta, -alpha z_(k+1) . A p_k /
 * (z_k . r_k), one more dot product per iteration.  Set before fi_assemble. */
#define FI_OPT_MG_KCYCLE 13
/* FI_OPT_MG_CHEB_DEGREE (0 or 2..16) and FI_OPT_MG_CHEB_RATIO (0 or (1, 1000]): degree and interval [lambda / ratio, 1.1 lambda] of
 * the Chebyshev smoother in the full operator (2-D lattices, oriented points, fp64 levels).  0 (default): by the lattice's
 * dimension -- 4 over [lambda / 10] in 2-D, 5 over [lambda / 40] in 3-D (swept under the field rule with the V-cycle).  With
 * the K-cycle on a deep hierarchy the coarse correction is strong and the smoother need not reach down: config 5 runs (4, 10)
 * (profiles/r6_ablation.md section 12).  Set before fi_assemble. */
#define FI_OPT_MG_CHEB_DEGREE 14
#define FI_OPT_MG_CHEB_RATIO 15
int fi_set_option(fi_ctx* ctx, int option, double value);

/* Replaces jacobi_iterations (sparse_linear.cpp:214-241): x <- x + w*(Atb - AtA x)/diag, true Jacobi. */
int fi_jacobi(fi_ctx* ctx, const float* guess, int num_iterations, float weight, float* out, int memory);

/* Replaces tile_solver_square (sparse_linear.cpp:246-390), the pre-solver behind SolveOptions.tile
 * (solve_tiled_with_guess :415-425): non-overlapping tile_size^D tiles, every tile solved for its own unknowns
 * with the couplings to other tiles moved to the right-hand side using `guess` (the reference moves every
 * off-tile coupling twice, :327-334 -- reproduced), 1e-6 added to the tile diagonals (:296-300).  The tiles are
 * independent SPD systems; they are solved together by one CG run on the block-diagonal tile operator
 * (fp32 contexts to a relative residual of 1e-6, fp64 to 1e-12) instead of one sparse Cholesky per tile.
 * Lattice rows (fi_set_model / fi_add_points) and fi_add_rows_coo rows alike -- the rows of a materialised
 * LinearEquation are split into their per-tile pieces on the fly; in a context of such rows only, a tile without
 * any entry keeps the guess, as the reference skips it.  tile_size >= 2 (:254). */
int fi_tile_pass(fi_ctx* ctx, const float* guess, int tile_size, float* out, int memory);

/* Replaces generate_error_map (field_interpolation.cpp:402-429): the blame heat-map of a solution.  Every
 * equation (model row, data row, fi_add_rows_coo row) distributes its squared residual (rhs - a.x)^2 over its
 * unknowns in proportion to a_j^2.  `solution` and `out`: owned unknowns, x fastest, fp32.  Rows added by
 * fi_add_points are read from the row tables of the last batches (kept until fi_clear_points). */
int fi_error_map(fi_ctx* ctx, const float* solution, float* out, int memory);

/* fp64 copy of the last solution (FI_F64 contexts keep full precision; FI_F32 widens). Host buffer. */
int fi_get_solution_f64(fi_ctx* ctx, double* out);

/* True residual ||Atb - AtA x||_2 / ||Atb||_2 of the last solution, evaluated on the GPU. */
int fi_true_residual(fi_ctx* ctx, double* rel_residual);

/* ---- test / measurement hooks (host buffers, owned unknowns) ------------------------------------ */
int fi_apply_AtA_f64(fi_ctx* ctx, const double* x, double* y); /* y = (A^T A) x */
int fi_get_Atb_f64(fi_ctx* ctx, double* out);
int fi_get_diag_f64(fi_ctx* ctx, double* out);
int fi_get_stats(const fi_ctx* ctx, fi_stats* out);
/* Device memory of destroyed contexts is kept (per device, at most 8 GiB) and handed to the next context: hipMalloc /
 * hipFree synchronise the device and dominate the cost of a context that lives for one solve -- what a caller of the
 * reference's stateless solve_sparse_linear* (sparse_linear.hpp:75-96) creates per call.  Frees pooled blocks of the
 * CURRENT device down to keep_bytes (0: everything; negative: nothing) and reports what stays cached. */
int fi_memory_pool(long long keep_bytes, long long* cached_bytes);
/* Launches the AtA apply `reps` times on the context's stream between two HIP events. */
int fi_time_apply(fi_ctx* ctx, int reps, double* ms_per_launch);

/* ---- loop-back group (test facility) ---------------------------------------------------------------
 * All `nranks` slabs of a decomposition in ONE process on the current device: the same kernels, slab
 * geometry, halo widths and ownership rules as the RCCL path, with halo planes moved by device-to-device
 * copies and dot products summed by a kernel.  Lets a single GPU check the decomposed solve against the
 * undivided one.  Per-rank assembly goes through fi_group_rank(g, r) and the fi_set_model / fi_add_points
 * calls above; vectors passed to the group calls are host buffers holding the WHOLE lattice. */
typedef struct fi_group fi_group;
int     fi_group_create(fi_group** out, int ndim, const int* sizes, int dtype, int nranks);
int     fi_group_destroy(fi_group* g);
int     fi_group_size(const fi_group* g);
fi_ctx* fi_group_rank(fi_group* g, int rank);
int     fi_group_assemble(fi_group* g);
int     fi_group_solve_cg(fi_group* g, const float* guess, int max_iterations, float tol, float* out, int* iterations,
                          float* rel_residual);
int     fi_group_apply_AtA_f64(fi_group* g, const double* x, double* y);
int     fi_group_true_residual(fi_group* g, double* rel_residual);
int     fi_group_get_solution_f64(fi_group* g, double* out);
int     fi_group_tile_pass(fi_group* g, const float* guess, int tile_size, float* out);
int     fi_group_error_map(fi_group* g, const float* solution, float* out);

/* ---- helpers either side of the path ---------------------------------------------------------
 * Replaces upscale_field (field_interpolation.cpp:431-485): multilinear resampling small -> large. */
int fi_upscale_field(const float* small_field, int ndim, const int* small_sizes, const int* large_sizes,
                     float* out, int memory);

#ifdef __cplusplus
}
#endif
#endif /* FI_HIP_H */
