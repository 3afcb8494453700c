// fi_internal.h -- shared declarations of the HIP solver core (not installed).
#pragma once

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "../../include/fi_hip.h"

namespace fi {

// ---- environment switches ------------------------------------------------------------------------------
// test_switch: alternative code paths the test-suite compares with each other (FI_NO_FUSE, FI_NO_MARCH, FI_NO_TILE2D,
// FI_NO_GATHER, FI_NO_PACK, FI_NO_OVERLAP, FI_ZC) -- every one of them computes the same answers; present in the shipped
// library.  tuning_switch: experiment knobs of the ablations in profiles/ (chunk shapes, smoother degrees, launch
// structure) and the FI_DBG timing modes whose results are wrong by construction: compiled in only with
// -DFI_TIMING_BUILD (tools/build_variant.sh); the shipped library never reads them.
inline const char* test_switch(const char* name) { return getenv(name); }
#ifdef FI_TIMING_BUILD
inline const char* tuning_switch(const char* name) { return getenv(name); }
#else
inline const char* tuning_switch(const char*) { return nullptr; }
#endif

// FI_ASM_CHAIN_TIMES (diagnostic): host time of the assembly's stages, per thread, since fi_assemble began -- the
// assembly of a config-4 step is bound by its threads' launch and round-trip times, not by the GPU (profiles/r6_ablation.md)
inline std::chrono::steady_clock::time_point& chain_origin()
{
	static std::chrono::steady_clock::time_point t0;
	return t0;
}
inline void chain_mark(const char* what, int level = -1)
{
	if (!tuning_switch("FI_ASM_CHAIN_TIMES")) { return; }
	if (!what) {
		chain_origin() = std::chrono::steady_clock::now();
		return;
	}
	const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - chain_origin()).count();
	std::fprintf(stderr, "fi_assemble host %8.1f us  %s%s%d\n", us, what, level >= 0 ? " level " : " ", level);
}

// ---- error plumbing: nothing throws or aborts across the C ABI ------------------------------------
void set_error(const char* fmt, ...);

struct Fail {
	int code;
};

#define FI_HIP_TRY(expr)                                                                              \
	do {                                                                                              \
		hipError_t e_ = (expr);                                                                       \
		if (e_ != hipSuccess) {                                                                       \
			::fi::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_));      \
			throw ::fi::Fail{FI_ERR_HIP};                                                             \
		}                                                                                             \
	} while (0)

#define FI_REQUIRE(cond, code, ...)                                                                   \
	do {                                                                                              \
		if (!(cond)) {                                                                                \
			::fi::set_error(__VA_ARGS__);                                                             \
			throw ::fi::Fail{code};                                                                   \
		}                                                                                             \
	} while (0)

// ---- device buffer --------------------------------------------------------------------------------
// Device memory of destroyed contexts is kept for the next one (fi_pool.hip): hipMalloc / hipFree synchronise the device
// and cost 50-500 us per block -- a 256^3 context holds ~60 blocks (first assemble + solve 14 ms against 8 warm, destroy
// 16-24 ms).  A block enters the pool only while `pool_quiescent` is set: fi_ctx_destroy sets it after synchronising
// every stream of the context (all work on a context's blocks is enqueued on its own streams; other streams and threads
// of the device are not stalled), so nothing in flight can still touch a pooled block; every other release (a buffer that
// grows, a temporary) is a plain hipFree, as before.  An allocation that runs out of memory empties the pool and tries
// once more.  FI_NO_POOL: no pooling (tests).
// (`used`: the bytes its last owner asked for -- the 64 bytes behind them are still zero, no kernel ever writes there)
void*  pool_take(size_t capacity_wanted, size_t* capacity, size_t* used);  // a pooled block of at least / at most twice that capacity, or nullptr
bool   pool_give(void* p, size_t capacity, size_t used);                   // false: the pool is full or off, the caller frees the block
size_t pool_trim(size_t keep_bytes);                          // frees pooled blocks down to keep_bytes; returns what stays
extern thread_local bool pool_quiescent;
// the stream a buffer allocated by this thread will FIRST be used on (set for the length of fi_assemble / a solve / a level's
// assembly on a helper thread): the 64 zero bytes behind a pooled block of another size are then written by an asynchronous
// fill on that stream instead of a synchronous hipMemset (24 us each, 86 per context of the headline solver); nullptr: synchronous
extern thread_local hipStream_t alloc_stream;
struct AllocStream {
	hipStream_t was;
	explicit AllocStream(hipStream_t st) : was(alloc_stream) { alloc_stream = st; }
	~AllocStream() { alloc_stream = was; }
	AllocStream(const AllocStream&) = delete;
	AllocStream& operator=(const AllocStream&) = delete;
};
hipStream_t stream_take();                           // a non-blocking stream: a pooled one of a destroyed context, or a new one
void        stream_give(hipStream_t st, bool drained);   // back to the pool (drained: nothing in flight on it), else destroyed
void*       pinned_take(size_t bytes, size_t* capacity);  // pinned host memory of at least `bytes`
void        pinned_give(void* p, size_t capacity);

struct DevBuf {
	void*  p     = nullptr;
	size_t bytes = 0;  // usable bytes (the request); 64 zeroed bytes follow
	size_t cap   = 0;  // bytes of the block
	DevBuf() = default;
	DevBuf(const DevBuf&) = delete;
	DevBuf& operator=(const DevBuf&) = delete;
	~DevBuf() { release(); }
	void release()
	{
		if (p && !(pool_quiescent && pool_give(p, cap, bytes))) { (void)hipFree(p); }
		p     = nullptr;
		bytes = 0;
		cap   = 0;
	}
	// 64 zeroed bytes of slack behind every buffer: the tiled kernels read whole 16-byte groups, and the group at
	// the end of a row whose length is not a multiple of the group reads on into the next row -- behind the very
	// last row that is this slack (finite values, only ever multiplied by zero masks / zero coefficients)
	void alloc(size_t nbytes)
	{
		if (nbytes <= bytes && p) { return; }
		release();
		if (nbytes == 0) { nbytes = 16; }
		size_t used = 0;
		p = pool_take(nbytes + 64, &cap, &used);
		if (!p) {
			cap = nbytes + 64;
			hipError_t e = hipMalloc(&p, cap);
			if (e == hipErrorOutOfMemory) {
				// the pool may be what fills the device (blocks no request of this process fits any more, or memory another
				// allocator of the process -- torch -- now wants): give everything back and ask once more
				(void)hipGetLastError();
				p = nullptr;
				(void)pool_trim(0);
				e = hipMalloc(&p, cap);
			}
			if (e != hipSuccess) {
				p   = nullptr;
				cap = 0;
				FI_HIP_TRY(e);
			}
		}
		if (used != nbytes) {
			// (asynchronously on the stream the caller named -- fi_assemble and fi_solve_cg name the context's stream, on which the
			// buffer's first reader runs; FI_SYNC_ALLOC_FILL=1 forces the synchronous fill, so that two runs can be compared bit for
			// bit when a buffer's first use on ANOTHER stream is suspected -- ADVICE r5)
			if (alloc_stream && !test_switch("FI_SYNC_ALLOC_FILL")) {
				FI_HIP_TRY(hipMemsetAsync(static_cast<char*>(p) + nbytes, 0, 64, alloc_stream));
			} else {
				FI_HIP_TRY(hipMemset(static_cast<char*>(p) + nbytes, 0, 64));
			}
		}
		bytes = nbytes;
	}
	void swap(DevBuf& o)
	{
		std::swap(p, o.p);
		std::swap(bytes, o.bytes);
		std::swap(cap, o.cap);
	}
	template <typename T>
	T* as() const { return static_cast<T*>(p); }
};

// ---- geometry: the (slab of the) lattice one context owns ------------------------------------------
// The slowest axis L = ndim-1 is the decomposed one.  Local storage holds `halo` ghost planes on both
// sides of the owned planes along L; every kernel takes global coordinates from `off`.
struct Geom {
	int     ndim;
	int     n[3];       // local extent incl. ghost planes
	int     gn[3];      // global extent
	int     off[3];     // global coordinate of local index 0
	int     own_lo[3];  // owned local range [lo, hi)
	int     own_hi[3];
	int64_t stride[3];  // local strides (x fastest)
	int64_t nloc;       // local elements incl. ghosts
	int64_t nown;       // owned elements
	int64_t own_first;  // local linear index of the first owned element (ghosts come first along L)
	// extended cell grid (cell origins -1 .. size-1): local extents and the global origin of cell 0
	int     cn[3];
	int     coff[3];
	// coarser levels: a data point at p on the caller's lattice sits at p / 2^level + pshift[d] on this one (non-zero along
	// the axes that were halved cell-centred: fi_ctx::cc)
	float   pshift[3];
};

// Coefficients of the model rows exactly as the reference stores them in A: fp32(stencil * weight)
// (sparse_linear.cpp:43), widened to T.  field_interpolation.cpp:257-315.
template <typename T>
struct ModelCoef {
	T   c[5][5];  // c[k][m], k = difference order 1..4, m = 0..k
	T   w0sq;     // model_0^2 (added once per axis)
	T   gs;       // gradient_smoothness (row coefficient +-gs)
	int on[6];    // on[k] for k = 0..4, on[5] = gradient smoothness
	int maxk;     // widest enabled difference order (0 if none)
};

// ---- per-cell data operator -----------------------------------------------------------------------
struct CellData {
	int64_t ncell = 0;
	DevBuf  cell_id;  // uint32[ncell], extended local cell id, ascending
	DevBuf  blk;      // T[ncell][nb]  upper triangle of the symmetric 2^D x 2^D block, one cell after another
	bool    blk_valid = false;  // 3-D contexts of the fused kernel: formed from the factor rows on first use (ensure_cell_blocks)
	DevBuf  nrow;     // uint32[ncell] number of data rows accumulated into the cell
	DevBuf  row1;     // T[ncell][2^D] the row itself for cells holding exactly one row (block = row row^T)
	DevBuf  mrow;     // T[ncell][2^D][2^D] (3-D only) up to 2^D factor rows a_k with block = sum a_k a_k^T
	DevBuf  nfac;     // uint32[ncell] number of factor rows
	int     nb = 0;   // entries per block: 2^D(2^D+1)/2
	bool    pack = false;  // cells of >= 3 rows are kept as packed blocks in mrow (contexts of mostly multi-row cells)
};

// Tiling of the z-marching stencil kernel (fi_stencil.hip) and, per workgroup and z-layer, the list of
// occupied cells whose corners touch the workgroup's tile (cells on a tile border are listed by every
// workgroup they touch).
struct MarchParams {
	int     nx, ny;          // lattice extent in x and y
	int     nzl;             // local planes (incl. ghost planes)
	int     gz;              // global extent of z
	int     zoff;            // global z of local plane 0
	int     own_z0, own_z1;  // owned local planes [z0, z1)
	int     tiles_x, tiles_y, chunks, zc;
	int     nwg;
	int     tx, ty;          // tile extent in x and y (lattice points)
	int     txt;             // threads along x (32: wide tiles, 16: 64 x 16)
	int     dbg;             // timing experiments only (FI_DBG): 1 = no halo loads, 2 = no stores
	int     dense_min;       // a layer with more records than this is scattered by all four waves (fi_stencil.hip)
	int64_t plane;           // nx * ny
};

struct MarchState {
	bool        valid = false;  // the marching kernel applies to this context
	bool        fused = false;  // cell blocks are applied inside the marching kernel
	bool        strip_lists = false;  // ... by the strip kernel (fi_strip.hip): the lists live in fi_ctx::strip, this state keeps none
	bool        no_lists = false;  // ... but never by IT: a small level of a hierarchy whose full operator runs as k_full_direct3
	                               // (diagonals) keeps no per-workgroup cell lists (fi_stencil.hip, stencil_prepare)
	bool        wide = false;   // the model has rows the marching kernel does not carry (model_3, model_4, gradient_smoothness,
	                            // field_interpolation.cpp:282-315): it applies model_0/1/2 and the cells, k_add_wide3 (fi_operator.hip)
	                            // adds the rest onto its result.  Whoever takes the marching kernel for the WHOLE operator -- the
	                            // epilogue recurrences, the polynomial, the split launches over slabs -- must look at this flag
	MarchParams P{};
	MarchParams Pplain{};  // chunking for launches of the plain (model-only) variant over the whole lattice: the
	                       // polynomial preconditioner's steps (4 workgroups per CU, longer chunks)
	// self-contained records per (workgroup, layer); two kinds: a cell holding a single data row (8
	// coefficients) or a cell holding several: up to 8 factor rows in a 64-coefficient slot
	DevBuf      lay_row, lay_blk;  // uint32[nwg*(zc+1)+1] record ranges
	DevBuf      pos_row, pos_blk;  // uint32[n] (tcx+1) | k << 8 | (tcy+1) << 16: tile-relative origin, k rows
	DevBuf      coef_row;          // T[n_row][8]
	DevBuf      coef_blk;          // T[n_blk][8][8]
	int64_t     n_row = 0, n_blk = 0;          // records (cells on tile borders are listed more than once)
	int64_t     cells_row = 0, cells_blk = 0;  // distinct cells of each kind
	DevBuf      wg_cells, wg_plain;            // workgroup ids with cells (ascending) / first workgroups of the plain runs
	DevBuf      wg_runs;                       // chunks per plain run (consecutive empty chunks of one tile)
	int         n_wg_cells = 0, n_wg_plain = 0;
	// slabs: workgroups of the first and last z-chunk (they read ghost planes) and all the others, for P and for
	// Pplain -- the interior launch runs while the ghost planes are still on the wire (apply_overlapped)
	DevBuf      wg_edge, wg_inner, wgp_edge, wgp_inner;
	int         n_edge = 0, n_inner = 0, np_edge = 0, np_inner = 0;
};

// Tiling of the 2-D tile kernel (fi_stencil2d.hip): one workgroup per TX x 16 tile of the owned rows.
struct Tile2Params {
	int nx;              // lattice extent in x
	int nyl;             // local rows (incl. ghost rows)
	int gy;              // global extent of y
	int yoff;            // global y of local row 0
	int own_y0, own_y1;  // owned local rows [y0, y1)
	int tx;              // tile extent in x
	int tiles_x, tiles_y, ntiles;
};

struct Tile2State {
	bool        valid = false, fused = false;
	Tile2Params P{};
	DevBuf      off;   // uint32[ntiles + 1] record range of every tile
	DevBuf      pos;   // uint32[nrec] (tcx+1) | (tcy+1) << 16
	DevBuf      blk;   // T[nrec][16] full symmetric 4x4 block
	int64_t     nrec = 0;
};

// Arbitrary sparse rows (fi_add_rows_coo and GradientKernel::kLinearInterpolation): fi_generic.hip
struct GenericRows {
	int64_t ntrip = 0, nrows = 0;  // accumulated input
	DevBuf  key;                   // uint64[ntrip]  (row << 32) | col
	DevBuf  val;                   // float[ntrip]
	DevBuf  rhs;                   // float[nrows]
	int64_t nnz = 0, ncols = 0;    // assembled: distinct (row, col) entries, columns that hold entries
	DevBuf  csr_ptr, csr_col, csr_val;            // A by rows
	DevBuf  csc_cols, csc_ptr, csc_row, csc_val;  // A by columns (transposed product without atomics)
	DevBuf  t;                                    // T[nrows]  t = A x
};

struct Pending {  // one fi_add_points batch, already turned into cell rows on the device
	int64_t nrows = 0;  // slots (valid or not)
	DevBuf  key;        // uint32[nrows]   extended local cell id or 0xFFFFFFFF
	DevBuf  coef;       // float[nrows * 2^D]
	DevBuf  rhs;        // float[nrows]
};

// The caller's points of one fi_add_points call, kept on the device so that coarser levels of the same
// problem (multilevel solve) can be assembled from them without another upload.
struct PointBatch {
	long   n = 0;
	DevBuf pos, nrm, pw, val;  // float; nrm/pw/val may be empty
	bool   has_nrm = false, has_pw = false, has_val = false;
	float  vw = 0, gw = 0;
	int    vk = 0, gk = 0;
	bool   prior = false;  // the rows of fi_add_border_prior: lattice points, not data -- no distance source of a later call
};

struct Comm;  // RCCL state (fi_comm.cpp)

struct CgScalars {  // lives in device memory; kernels read/write it, the host polls it
	double rz, rz_new, pq, rr, bb, tol2, alpha, beta, true_rr;
	double sums[4];
	// FI_OPT_FIELD_TOLERANCE over slabs: every slab's two maxima travel with the r.r sum -- slab s writes entries 2 s and
	// 2 s + 1, zeros elsewhere, and the SUM over the slabs (the all-reduce the iteration makes anyway, over sums[0 .. 3] and
	// these) hands every rank all of them; the stop test takes their maximum.  Directly behind sums[]: one contiguous run.
	double rank_max[2 * 16];
	double tscale;  // mixed precision: the scale the fp32 copy of the current residual was divided by
	int    iter, done, max_iter, restarts;
	int    field_min_iter, pad2_;  // FI_OPT_FIELD_TOLERANCE: no stop before this iteration (kFieldMinIter; more behind a caller's guess)
	int    tag, field_ranks;  // tag: second slot only (single-rank fused CG): the iteration whose first half filled it;
	                          // field_ranks: slabs whose maxima rank_max[] holds (0: an undivided lattice, dmax_bits / xmax_bits)
	// FI_OPT_FIELD_TOLERANCE (V-cycle PCG, fi_multigrid.hip): the stop test on the field.  The step kernel leaves
	// max |x_k - x_(k-1)| = |alpha| max |p| and max |x_k| here (bit patterns of non-negative doubles: atomicMax)
	double field_tol, field_est, field_kappa;
	unsigned long long dmax_bits, xmax_bits;
	// ... and the last kFieldHist iterations' relative residuals and relative steps (slot: iteration % kFieldHist)
	double hist_r[32], hist_s[32], hist_r0;  // (hist_r0: the start residual)
	double hist_t[32];  // ... and the steps' squared A-norms, alpha_k r_k . z_k (||x* - x_k||_A^2 is the sum of those still to come)
};
constexpr int kFieldHist = 32;
constexpr int kFieldRanks = 16;  // slabs the field rule runs over (CgScalars::rank_max)

}  // namespace fi

struct fi_ctx {
	int        dtype = FI_F32;
	int        device = 0;
	int        rank = 0, nranks = 1;
	int        n_halo_exchanges = 0;  // halo exchanges of this level since the solve reset it (cg_run_mg: fi_stats.halo_exchanges)
	int        halo = 0;    // ghost planes STORED on each side of the slab along the slowest axis (>= reach)
	int        min_slab = 0;  // the thinnest slab of this level over all ranks (a deep exchange needs that many planes to send)
	int        reach = 1;   // planes a stencil / transfer reads beyond the slab = the default width of an exchange.  halo > reach:
	                        // the polynomial preconditioner exchanges 2 (d - 1) planes of r ONCE and runs its steps redundantly on
	                        // the shrinking ghost zone (cg_run_poly, "deep halo")
	int        slab_lo = 0, slab_hi = 0;
	bool       slab_fixed = false;   // coarser levels: the slab range follows the finest level's, not the equal split
	fi::Geom   g{};
	fi_weights w{};
	bool       model_set = false;
	bool       assembled = false;
	bool       vectors_ready = false;
	bool       vectors_stale = true;   // the solver vectors must be zeroed before their next use (new, or the local geometry changed)
	hipStream_t stream = nullptr;

	std::vector<fi::Pending*> pending;
	std::vector<fi::Pending*> pending_pool;  // buffers of cleared batches, reused by the next fi_add_points
	std::vector<fi::PointBatch*> batches, batches_pool;  // the points themselves (for coarser levels)

	// multilevel: coarser replicas of this problem (lattice halved per level), owned by the finest context
	int        levels_wanted = 0;
	double     coarse_tol = 1e-3;
	fi_ctx*    coarse = nullptr;   // next coarser level
	fi_ctx*    finer = nullptr;
	int        level = 0;
	// A level of the REPLICATED TAIL of a slab hierarchy: once a level's slabs would be thinner than 4 planes, the levels
	// below are undivided lattices that every rank holds and solves in full (a 16^3 lattice over 8 GPUs is not worth one
	// message); the right-hand side reaches them through one all-reduce per V-cycle (build_levels, vcycle)
	bool       replicated = false;
	// how this level was derived from the finer one, per axis.  1 (even fine extent n): cell-centred -- n / 2 coarse points,
	// coarse point j halfway between fine 2j and 2j+1; every fine point interpolates (3/4, 1/4), the first and last
	// extrapolate (5/4, -1/4).  0 (odd n): vertex-centred -- (n + 1) / 2 points, coarse j ON fine 2j.  An even extent halved
	// vertex-centred leaves the last fine point beyond the last coarse one: its constant extrapolation costs the V-cycle
	// a factor 3 in its condition number (tools/proto_multilevel.py: lattice-edge modes at 0.31 and 1.98 against 0.62 .. 1.18)
	int        cc[3] = {0, 0, 0};
	float      pos_shift[3] = {0, 0, 0};
	fi::CellData              cells;
	fi::MarchState            march;
	fi::MarchState            strip;   // fi_strip.hip: the apply of undivided 3-D fp64 lattices as wave-private strips (P: tx x ty = one
	                                   // wave's strip, nwg = waves of the launch; the lists as the marching kernel's, per strip)
	fi::Tile2State            tile2;
	fi::GenericRows           generic;

	// operator pieces (T arrays over local storage)
	fi::DevBuf atb, diag, dinv;
	fi::DevBuf dinv16;  // dinv truncated to bfloat16 (k_invert_diag): the scaling of the epilogue recurrences
	// the V-cycle's polynomial smoother (fi_solver.hip, poly_smooth) scales by 1 / (m + f d), m = the model diagonal, d = the
	// data diagonal: a cell block sum a a^T is bounded by 2^D diag(a_i^2), so A <= A_model + 2^D diag(A_data) and the
	// polynomial in that operator is a convergent smoother of A whatever the data (f = mg_safe)
	fi::DevBuf dinv16s;
	bool       dinv16s_valid = false;
	// The fp32 replica of a mixed-precision context whose data are value rows (fi_solver.hip, twin_assemble_lumped): its
	// finest level runs on the LUMPED operator A~ = A_model + diag(dlump), dlump = the row sums of the data term (a value row
	// a >= 0 has a a^T <= (sum a) diag(a): A <= A~, equal to second order on smooth fields).  No rows, no cells, no sort for
	// the replica's finest level; CG on the exact fp64 operator takes the same number of iterations (tools/proto_lumped.py)
	bool       lumped = false;
	fi::DevBuf dlump;   // float[nloc]
	// on the fp64 context of such a pair: the row sums of ITS data term, formed by the assembly beside A^T b and the
	// diagonal (fi_assembly.hip: a third entry of every cell's record) when fi_assemble sets want_lump
	hipEvent_t ev_asm0 = nullptr, ev_asm1 = nullptr;  // around the last fi_assemble; read by finish_assemble_timing
	bool       asm_time_pending = false;
	bool       want_lump = false;
	fi::DevBuf lump;    // float[nloc]
	bool       data_pinned = false;  // (levels of <= 2^16 points, set with dinv16s) every point's data diagonal reaches its model
	                                 // diagonal: value rows that dense hold every smooth mode, and two sweeps of the polynomial
	                                 // smoother solve such a coarsest level as well as an exact solve (tools/proto_cc.py)
	double     mg_safe = 4.0;
	int        mg_terms = 5;      // the polynomial smoother: terms and interval ratio (FI_OPT_MG_TERMS / FI_OPT_MG_RATIO)
	double     mg_pratio = 30.0;
	int        mg_cheb_degree = 0;  // FI_OPT_MG_CHEB_DEGREE / _RATIO: the full-operator Chebyshev smoother (0: by the lattice's dimension)
	double     mg_cheb_ratio = 0.0;
	int        mg_kcycle = 0;      // FI_OPT_MG_KCYCLE: coarse levels (from the first one down) whose correction is two flexible-CG steps
	fi::DevBuf kc;                 // ... and their coefficients (KcScalars, fi_multigrid.hip)
	int        mg_smoother = 1;   // 1: the polynomial in A_model + f diag(A_data) where the marching kernel runs it; 0: Chebyshev in A
	// Data facts that decide which kernels AND WHICH COLLECTIVES a solve runs -- triplet rows anywhere (any_trip), gradient
	// rows anywhere (value_rows_only is its negation) -- are agreed over the ranks at the start of fi_assemble (one
	// all-reduce; a loop-back group decides over its members: facts_forced): a rank whose slab happens to hold no points
	// would otherwise take another branch than its neighbours (poly_ok, poly_smoother_ok, the deep halo, `beside`)
	bool       any_trip = false;
	bool       facts_forced = false, forced_trip = false, forced_grad = false;
	bool       value_rows_only = false;  // set by fi_assemble (levels and replicas: from the context that holds the points): no
	                                     // gradient rows.  The polynomial smoother is used for such data only: a gradient row
	                                     // a (+-w/4 on the cell's corners) makes a a^T large exactly where diag(a_i^2) is not --
	                                     // config 5 takes 34 iterations with it against 25 with the Chebyshev smoother in A
	bool       scaling_ghosts = false;  // slabs: the ghost planes of diag / dinv / dinv16 hold the neighbours' values
	                                    // (exchanged by operator_prepare when a transport exists)
	bool       defer_scaling_exchange = false;  // a level built by fi_assemble's helper thread: the exchange is the
	                                            // calling thread's to do (operator_finish_ghosts), after the join
	// solver vectors
	fi::DevBuf x, r, p, q;
	// multigrid work vectors of this level: V-cycle rhs / result, smoother residual and direction
	fi::DevBuf mg_b, mg_x, mg_r, mg_d;
	double     lambda_max = 0;    // estimate of the largest eigenvalue of Dinv * AtA on this level
	int        poly_terms = 0;    // > 1: CG preconditioned by a Chebyshev polynomial of that many terms (cg_run_poly)
	double     poly_ratio = 10.0; // the polynomial's interval is [hi / ratio, hi], hi = 1.1 * poly_lambda
	double     poly_lambda = 0;   // largest eigenvalue of diag(A_model)^-1 A_model (power method, once per model)
	int        last_mg_iterations = 0;     // of the previous V-cycle PCG solve of this context, its tolerance, and whether the
	double     last_mg_tol = 0;            // solve about to run starts the way that one did (no caller's guess): cg_run_mg
	bool       predictable_start = false;
	double     last_cg_tol = 0;            // ... and its tolerance
	bool       unwatched_pending = false;  // a coarse-to-fine level solved without a look at its flag: the flag's copy is in pin[2]
	int        unwatched_expected = 0;
	hipEvent_t ev_unwatched = nullptr;
	int        last_cg_iterations = 0;     // of the previous Jacobi-PCG solve of this context (coarser levels: first look at the stop flag)
	double     field_tol = 0;          // FI_OPT_FIELD_TOLERANCE (0: the residual rule)
	bool       pred_recalled[2] = {false, false};  // a fresh context has asked the process-wide record of iteration counts once
	bool       cg_count_recalled = false;  // last_cg_iterations came from that record, not from a solve of this context: the next
	                                       // solve is scheduled by it but WATCHED (cg_run)
	                                       // (fi_cg.hip, recall_iterations: [0] Jacobi-PCG, [1] V-cycle PCG)
	int        last_outer_iterations = 0;  // of the previous polynomial-PCG solve of this context (first look at the stop flag)
	int        mg_mode = 0;       // 0: Jacobi-PCG (+ cascade start when levels exist); 1: V-cycle preconditioned CG
	fi::DevBuf partial;       // double[4 * max_blocks]
	fi::DevBuf scal;          // CgScalars
	fi::CgScalars* scal_host = nullptr;  // pinned
	// pinned staging of the assembly's small host copies (slot 0: read-backs of sizes, slot 1: uploads of workgroup lists).
	// A copy from / to pageable memory is staged by the runtime inside the call, which waits for the stream -- with several
	// threads assembling levels side by side the others' launches queued up behind it (gaps of 100-150 us in their chains).
	void*      pin[3] = {nullptr, nullptr, nullptr};  // (slot 2: the stop flag of an unwatched coarse-level solve, cg_run)
	size_t     scal_host_cap = 0;
	size_t     pin_bytes[3] = {0, 0, 0};
	int        max_blocks = 0;

	// assembly temporaries, kept between fi_assemble calls (hipMalloc/hipFree are slow and synchronising)
	fi::DevBuf scratch[36];

	// the small-level engine (fi_tail.h): this level and every level below it (all of <= 4 096 unknowns) run their share of a
	// V-cycle in one launch of one workgroup.  tail_dia: the data rows as 3^D diagonals; tail_prog: the levels' table and the
	// stages of the cycle (built at the first cycle after an assemble / a change of the smoothers' bounds)
	// fp32 replica of a mixed-precision solve: the fp64 CG's r . z is b . x of the V-cycle, and the cycle's last launch on the
	// finest level (the post-smoothing polynomial's last step) can sum it on the way (ChebEpi::dotv): bx_dot_wanted is set by
	// cg_run_mg, bx_dot_done by the cycle that delivered the partials (fi_multigrid.hip)
	bool       bx_dot_wanted = false, bx_dot_done = false;
	bool       tail_ok = false;
	fi::DevBuf tail_dia, tail_map, tail_prog;
	bool       tail_prog_valid = false;
	bool       dia_valid = false;  // tail_dia holds THIS assembly's data rows (fi_levels.hip: levels of up to 2^19 points keep them for
	                               // the full operator's direct launches, fi_stencil.hip k_full_direct3, not for the engine alone)
	int        tail_nlev = 0, tail_nops = 0;
	int        tail_lds_floats = 0;  // LDS the program's levels take (fi_tail.h: tail_level_floats)

	fi::Comm*  comm = nullptr;
	hipStream_t comm_stream = nullptr;   // slabs over RCCL: the halo exchange runs here beside the interior launch
	hipStream_t level_stream = nullptr;  // fi_assemble: the coarser levels are assembled here, by a helper thread, beside
	                                     // the finest level on `stream`
	hipEvent_t  ev_level = nullptr;
	hipStream_t level_stream2 = nullptr;  // mixed precision: the replica's coarser levels (a second helper thread)
	hipStream_t build_stream = nullptr;   // a coarser level beyond the first: the stream (and thread) its assembly runs on
	hipEvent_t  ev_build = nullptr;
	hipEvent_t  ev_level2 = nullptr;
	hipEvent_t  ev_ready = nullptr, ev_halo = nullptr;
	fi::DevBuf group_scal;    // loop-back group: CgScalars* of every member (held by member 0)
	bool       owns_stream = true;
	bool       owns_comm = true;     // coarser levels share the RCCL communicator of the finest level
	// mixed precision (FI_OPT_MIXED_PRECISION on an FI_F64 context): an fp32 replica of the same problem carries
	// the levels and runs the V-cycle preconditioner; CG itself stays on this context in fp64
	int        mixed = 0;
	fi_ctx*    twin = nullptr;
	int        tile_ts = 0;          // > 0 while fi_tile_pass runs: apply_AtA applies the tile operator of that tile size
	int        verify_residual = 1;  // check b - A x when the recurrence converges, restart CG if it misses
	fi_stats   stats{};
	std::vector<hipEvent_t> ev;       // sampled events around AtA applies
	std::vector<hipEvent_t> ev_prec;  // ... around Chebyshev steps of the polynomial preconditioner
	// V-cycle PCG: the finest level's pre-smoothing chains (the polynomial's launches from zero) are timed for the first
	// prec_budget cycles of a solve; prec_taken pairs of ev_prec are valid, prec_chain_bytes = algorithmic bytes of one chain
	int        prec_budget = 0, prec_taken = 0, prec_chain_launches = 0;
	double     prec_chain_bytes = 0;
};

namespace fi {

size_t elem_size(const fi_ctx* c);

// fi_operator.hip
// model diag etc. after assemble; with_scaling: the polynomial smoother's scaling too (prepare_safe_scaling, fused into the
// same pass on undivided 3-D levels); lump_in: a lumped replica takes its data diagonal from these row sums
void operator_prepare(fi_ctx* c, bool with_scaling = false, const float* lump_in = nullptr);
void apply_AtA(fi_ctx* c, const void* x, void* y, double* pq_partial);  // y = AtA x (+ fused x.y partials)
int  apply_num_partials(const fi_ctx* c);
double apply_algorithmic_bytes(const fi_ctx* c);
void error_map(fi_ctx* c, const void* x, void* out);                 // generate_error_map; x with valid ghost planes
void exchange_halo(fi_ctx* c, void* v, int width = 0);               // fi_comm.hip; width planes next to the slab (0: reach)
void exchange_halo_on(fi_ctx* c, void* v, hipStream_t stream, int width = 0);  // the same on another stream
bool comm_ready(const fi_ctx* c);                                   // a transport exists (fi_comm_init / fi_comm_init_host)
void operator_finish_ghosts(fi_ctx* c);  // the deferred exchange of the diagonal's ghost planes + the scaling over them
void operator_rescale_with_ghosts(fi_ctx* c);  // dinv / dinv16 over all local planes from diag as it stands (loop-back group)
void prepare_safe_scaling(fi_ctx* c);    // dinv16s (see fi_ctx) from diag and the model diagonal, ghost planes included

// fi_stencil.hip: LDS-tiled z-marching kernel for 3-D lattices (model_0/1/2); false => use the generic kernel
void stencil_prepare(fi_ctx* c);   // after assemble(): tiling + per-workgroup cell lists
int  stencil_partials(const fi_ctx* c);
bool stencil_apply(fi_ctx* c, const void* x, void* y, double* partial);
// part 1: the workgroups that read no ghost plane, part 2: the others (first and last z-chunk); false when the context's
// apply is not one launch over all workgroups (then the caller exchanges first and applies in one go)
bool stencil_apply_part(fi_ctx* c, const void* x, void* y, double* partial, int part);

// fi_strip.hip
bool strip_wanted(const fi_ctx* c);
void strip_setup(fi_ctx* c);       // fi_ctx::strip.P (valid only where the kernel applies); called by stencil_prepare
void strip_apply(fi_ctx* c, const void* x, void* y, double* partial);

bool cells_fused(const fi_ctx* c);  // the stencil kernel of this context also applies the cell blocks
// one step of the Chebyshev polynomial preconditioner / of the power method through the plain marching kernel
// (fi_stencil.hip, ChebEpi); z / v with valid ghost planes; partials: one per workgroup of the launch
bool stencil_cheb_available(const fi_ctx* c);
int  stencil_cheb_partials(const fi_ctx* c);
void stencil_cheb_step(fi_ctx* c, const void* z, const void* zprev, const void* r, void* znew, double c1, double c2,
                       double* partial, int part = 0, double zprev_scale = 0.0, double pro_scale = 0.0,
                       const unsigned short* scaling = nullptr,   // scaling: bfloat16 array (default: the context's dinv16)
                       int extend = 0,    // slabs, deep exchange: the step also covers `extend` ghost planes on either side
                       int fmt = 0,       // bfloat16 storage of z (bit 0), z_prev (bit 1), z_new (bit 2): fp32 3-D levels, ChebEpi::fmt
                       const void* acc = nullptr,   // z_new = acc + the step's result (ChebEpi::acc; may be znew itself)
                       const void* dotv = nullptr); // the partials are those of dotv . z_new instead of r . z_new (ChebEpi::dotv)
int  stencil_cheb_partials_max(const fi_ctx* c);  // room for the partials of a step extended over the whole ghost zone
// small undivided fp32 levels: a step whose caller wants no partials (partial == nullptr) runs as one thread per point with
// direct neighbour loads instead of the z-marching kernel (fi_stencil.hip, k_cheb_direct3)
bool stencil_cheb_direct(const fi_ctx* c);
bool stencil_full_direct_wanted(const fi_ctx* c);  // the level's data rows are worth keeping as diagonals (asked while the level is assembled)
void stencil_power_step(fi_ctx* c, const void* v, void* vnew, double* partial);
// z_new = a z - c1 z_prev + c2 Dinv (r - A z) on the FULL operator (residual: z_new = r - A z) in one pass of the
// marching kernel(s) over the lattice; z with valid ghost planes.  Dinv is the context's bfloat16 copy (dinv16).
bool stencil_full_epi_available(const fi_ctx* c);
void stencil_full_step(fi_ctx* c, const void* z, const void* zprev, const void* r, bool residual, void* znew, double a,
                       double c1, double c2);
void stencil_lumped_residual(fi_ctx* c, const void* x, const void* b, void* out);  // out = b - (A_model + diag(dlump)) x

// fi_stencil2d.hip: LDS-tiled kernel for 2-D lattices (model_0/1/2), called through the stencil_* entry points
void tile2d_prepare(fi_ctx* c);
int  tile2d_partials(const fi_ctx* c);
bool tile2d_apply(fi_ctx* c, const void* x, void* y, double* partial);
void tile2d_cheb_step(fi_ctx* c, const void* z, const void* zprev, const void* r, void* znew, double c1, double c2, double* partial,
                      double zprev_scale, const unsigned short* scaling);   // stencil_cheb_step for 2-D lattices
void tile2d_power_step(fi_ctx* c, const void* v, void* vnew, double* partial);
bool tile2d_full_epi_available(const fi_ctx* c);  // the smoother's recurrence step / residual in the tile kernel's epilogue
void tile2d_full_step(fi_ctx* c, const void* z, const void* zprev, const void* r, bool residual, void* znew, double a, double c1,
                      double c2);

// fi_generic.hip
void generic_add_coo(fi_ctx* c, int64_t nrows, int64_t ntrip, const fi_triplet* trip, const float* rhs, int memory);
void generic_add_gradient_linear(fi_ctx* c, long n, const float* pos, const float* nrm, const float* pw, float gw,
                                 float pos_scale = 1.0f, float nrm_scale = 1.0f);
void generic_clear(fi_ctx* c);
void generic_assemble(fi_ctx* c);                      // after assemble(): adds A^T b and diag, builds CSR/CSC
int  generic_num_partials(const fi_ctx* c);
void generic_apply(fi_ctx* c, const void* x, void* y, double* partial);  // y += A^T (A x)
void generic_apply_tile(fi_ctx* c, const void* x, void* y, double* partial, int ts);  // the tile operator (fi_tile_pass)
void generic_keep_guess_in_empty_tiles(fi_ctx* c, int ts, const void* guess, void* x);  // rows-only contexts
void generic_error_map(fi_ctx* c, const void* x, void* out);             // out += blame of the generic rows

// fi_assembly.hip
// positions are multiplied by pos_scale and normals by nrm_scale before use (coarser levels: 1/2^l and 2^l)
void emit_point_rows(fi_ctx* c, long n, const float* pos, const float* nrm, const float* pw, const float* val,
                     float vw, int vk, float gw, int gk, float pos_scale = 1.0f, float nrm_scale = 1.0f);
void assemble(fi_ctx* c);
void ensure_cell_blocks(fi_ctx* c);     // CellData::blk of every cell, on the context's stream (no-op when they are there)
bool stencil_will_fuse(const fi_ctx* c);  // will stencil_prepare() put this context's cells into the marching kernel?
void* pinned(fi_ctx* c, int slot, size_t bytes);  // the context's pinned staging buffer `slot`, at least `bytes` long (contents not kept when it grows)
// border prior (src/sdf_field.cpp:218-246): coordinates of the lattice's border points and their distance to the nearest
// data point added so far, as device buffers; returns their number
int64_t border_prior_points(fi_ctx* c, DevBuf& pos, DevBuf& val);
// fi_solver.hip: a batch of points whose arrays already sit in HBM (kept for the coarser levels, rows emitted)
void add_points_device(fi_ctx* c, long n, const float* p, const float* g, const float* w, const float* v, float value_weight,
                       int value_kernel, float gradient_weight, int gradient_kernel);

// fi_comm.cpp
void comm_destroy(Comm* cm);
void allreduce_sum(fi_ctx* c, double* dev, int count);
long comm_allreduces(const fi_ctx* c);  // all-reduces this context's communicator has issued so far (0 without one)
void allreduce_sum_vec(fi_ctx* c, void* dev, int64_t count, bool f64);  // a whole vector, in place

}  // namespace fi
