// Host side of the drop-in library: sparse_linear.hpp on top of libfi_hip (include/fi_hip.h).
//
// The reference builds Eigen matrices here (sparse_linear.cpp:59-113) and calls SimplicialLLT / BiCGSTAB.
// This file uploads the triplets as generic sparse rows (fi_add_rows_coo) and iterates A^T A x = A^T b on
// the GPU: Jacobi-preconditioned CG with the reference's stop rule; "exact"/"fast" iterate to a tight
// tolerance in fp64.  Failure conventions follow the reference: log + empty vector.  Device contexts are cached per
// lattice shape between the stateless calls.
#include "field_interpolation/sparse_linear.hpp"

#include <climits>
#include <cstdio>
#include <cstdlib>
#include <ostream>

#include <fi_hip.h>

#include "recipe.hpp"

namespace field_interpolation {

std::ostream& operator<<(std::ostream& os, const LinearEquation& eq)
{
	std::vector<std::vector<const Triplet*>> by_row(eq.rhs.size());
	for (const Triplet& t : eq.triplets) { by_row[t.row].push_back(&t); }
	for (size_t r = 0; r < by_row.size(); ++r) {
		os << eq.rhs[r] << " = ";
		for (size_t k = 0; k < by_row[r].size(); ++k) {
			os << by_row[r][k]->value << " * x" << by_row[r][k]->col;
			if (k + 1 < by_row[r].size()) { os << "  +  "; }
		}
		os << "\n";
	}
	return os;
}

void add_equation(LinearEquation* eq, Weight weight, Rhs rhs, std::initializer_list<LinearEquationPair> pairs)
{
	if (weight.value == 0) { return; }
	const int row = static_cast<int>(eq->rhs.size());
	bool any = false;
	for (const LinearEquationPair& p : pairs) {
		if (p.value != 0) {
			eq->triplets.emplace_back(row, p.column, p.value * weight.value);
			any = true;
		}
	}
	if (any) { eq->rhs.emplace_back(rhs.value * weight.value); }
}

namespace {

static_assert(sizeof(Triplet) == sizeof(fi_triplet), "Triplet must stay 12 bytes: it is handed to the C ABI as is");

void warn(const char* what) { std::fprintf(stderr, "field_interpolation: %s: %s\n", what, fi_last_error()); }

// Device contexts are kept between calls, keyed by (lattice shape, precision): the reference API is stateless and
// its per-frame caller (src/bipolar_2d.cpp:730: a 128^2 warm-started solve every frame) would otherwise pay context
// creation -- stream, hipMalloc of every vector, the sort buffers -- on each call (SURVEY.md 8(b) allows exactly this
// cache).  A handful of shapes, least recently used first out; one cache per thread, like the contexts themselves.
struct CachedCtx {
	std::vector<int> shape;
	int              dtype = 0;
	fi_ctx*          ctx = nullptr;
	unsigned long    used = 0;
};
struct CtxCache {
	std::vector<CachedCtx> slots;
	unsigned long          clock = 0;
	~CtxCache()
	{
		for (CachedCtx& s : slots) { fi_ctx_destroy(s.ctx); }
	}
	// Only small systems are kept: the cache exists for the per-frame caller's latency (a 128^2 system), and a context
	// holds every vector and sort buffer of its system -- a few cached 256^3 contexts would pin gigabytes of HBM until the
	// thread ends.  Larger systems get a context of their own, destroyed with the call (RowsOnGpu::owned).
	static bool cacheable(const std::vector<int>& shape)
	{
		long long n = 1;
		for (int s : shape) { n *= s; }
		return n <= (1LL << 20);
	}
	void clear()
	{
		for (CachedCtx& s : slots) { fi_ctx_destroy(s.ctx); }
		slots.clear();
	}
	fi_ctx* get(const std::vector<int>& shape, int dtype)
	{
		++clock;
		for (CachedCtx& s : slots) {
			if (s.dtype == dtype && s.shape == shape) {
				s.used = clock;
				return s.ctx;
			}
		}
		fi_ctx* ctx = nullptr;
		if (fi_ctx_create(&ctx, static_cast<int>(shape.size()), shape.data(), dtype) != FI_OK) {
			warn("fi_ctx_create");
			return nullptr;
		}
		if (slots.size() >= 6) {
			size_t old = 0;
			for (size_t i = 1; i < slots.size(); ++i) {
				if (slots[i].used < slots[old].used) { old = i; }
			}
			fi_ctx_destroy(slots[old].ctx);
			slots.erase(slots.begin() + static_cast<long>(old));
		}
		slots.push_back(CachedCtx{shape, dtype, ctx, clock});
		return ctx;
	}
	void drop(fi_ctx* ctx)  // a context that failed is not reused
	{
		for (size_t i = 0; i < slots.size(); ++i) {
			if (slots[i].ctx == ctx) {
				fi_ctx_destroy(ctx);
				slots.erase(slots.begin() + static_cast<long>(i));
				return;
			}
		}
	}
};
CtxCache& cache()
{
	static thread_local CtxCache c;
	return c;
}

// The caller's rows on the device: a context that holds nothing else -- a 1-D "lattice" of num_columns unknowns (or the
// caller's lattice, which only the tile pre-solver looks at), no model rows.
struct RowsOnGpu {
	fi_ctx* ctx = nullptr;
	bool    owned = false;  // not from the cache: destroyed with this object
	~RowsOnGpu()
	{
		if (owned && ctx) { fi_ctx_destroy(ctx); }
	}
	// The rows' note (recipe.hpp) still holds: the noted rows matrix-free on their lattice -- fi_set_model for the model rows
	// (never uploaded: config 3's 100 M triplets stay on the host), fi_add_points for the data rows from the noted copies of
	// the point arrays -- and only the rows nobody vouches for as triplets.  false: nothing was set up (the caller's edits
	// broke a checksum, two sets of model weights, a triplet appended to a noted row, a lattice of another shape, more than
	// three dimensions): the generic path below takes all rows, as before round 6.
	bool from_recipe(const LinearEquation& eq, int num_columns, int dtype, const std::vector<int>* lattice)
	{
		const detail::Recipe* r = eq.recipe.get();
		if (!r || r->segments.empty() || std::getenv("FI_DROPIN_NO_RECIPE")) { return false; }
		if (lattice && !lattice->empty() && *lattice != r->sizes) { return false; }
		if (r->sizes.empty() || r->sizes.size() > 3) { return false; }
		long long n = 1;
		for (int s : r->sizes) { n *= s; }
		if (n != num_columns) { return false; }
		const detail::Segment* model = nullptr;
		size_t row_end = 0, trip_end = 0;
		for (const detail::Segment& s : r->segments) {
			if (s.row0 < row_end || s.trip0 < trip_end || s.row1 < s.row0 || s.trip1 < s.trip0 || s.row1 > eq.rhs.size() ||
			    s.trip1 > eq.triplets.size()) {
				return false;
			}
			row_end  = s.row1;
			trip_end = s.trip1;
			if (detail::sample_checksum(eq.triplets, s.trip0, s.trip1, eq.rhs, s.row0, s.row1) != s.checksum) { return false; }
			if (s.kind == detail::Segment::kModel) {
				if (model) { return false; }
				model = &s;
			}
		}
		// the rows between the noted ranges, renumbered from 0 in their order
		std::vector<Triplet> extra;
		std::vector<float>   extra_rhs;
		{
			size_t t = 0, row = 0, shift = 0;  // shift: noted rows in front of the current gap
			auto gap = [&](size_t t_end, size_t row_endx) -> bool {
				for (; t < t_end; ++t) {
					const Triplet& q = eq.triplets[t];
					if (q.row < 0 || static_cast<size_t>(q.row) < row || static_cast<size_t>(q.row) >= row_endx) { return false; }
					extra.emplace_back(static_cast<int>(static_cast<size_t>(q.row) - shift), q.col, q.value);
				}
				for (; row < row_endx; ++row) { extra_rhs.push_back(eq.rhs[row]); }
				return true;
			};
			for (const detail::Segment& s : r->segments) {
				if (!gap(s.trip0, s.row0)) { return false; }
				t = s.trip1;
				row = s.row1;
				shift += s.row1 - s.row0;
			}
			if (!gap(eq.triplets.size(), eq.rhs.size())) { return false; }
		}
		if (CtxCache::cacheable(r->sizes)) {
			ctx = cache().get(r->sizes, dtype);
		} else if (fi_ctx_create(&ctx, static_cast<int>(r->sizes.size()), r->sizes.data(), dtype) == FI_OK) {
			owned = true;
		} else {
			warn("fi_ctx_create");
			ctx = nullptr;
		}
		if (!ctx) { return false; }
		fi_weights w = {1, 1, 0, 0, 0, 0, 0, 0, FI_VALUE_LINEAR_INTERPOLATION, FI_GRADIENT_CELL_EDGES};
		if (model) {
			const Weights& m = model->weights;
			w = fi_weights{m.data_pos, m.data_gradient, m.model_0, m.model_1, m.model_2, m.model_3, m.model_4, m.gradient_smoothness,
			               static_cast<int>(m.value_kernel), static_cast<int>(m.gradient_kernel)};
		}
		bool ok = fi_clear_points(ctx) == FI_OK && fi_set_model(ctx, &w) == FI_OK;
		for (const detail::Segment& s : r->segments) {
			if (!ok || s.kind != detail::Segment::kPoints || s.num_points <= 0) { continue; }
			ok = fi_add_points(ctx, s.num_points, s.positions.data(), s.normals.empty() ? nullptr : s.normals.data(),
			                   s.point_weights.empty() ? nullptr : s.point_weights.data(), nullptr, s.value_weight,
			                   static_cast<int>(s.value_kernel), s.gradient_weight, static_cast<int>(s.gradient_kernel), FI_HOST) == FI_OK;
		}
		if (ok && !extra_rhs.empty()) {
			ok = fi_add_rows_coo(ctx, static_cast<long>(extra_rhs.size()), static_cast<long>(extra.size()),
			                     reinterpret_cast<const fi_triplet*>(extra.data()), extra_rhs.data(), FI_HOST) == FI_OK;
		}
		ok = ok && fi_assemble(ctx) == FI_OK;
		if (!ok) {  // (a state the lattice path does not take: the generic path answers)
			if (owned) { fi_ctx_destroy(ctx); } else { cache().drop(ctx); }
			owned = false;
			ctx   = nullptr;
			return false;
		}
		matrix_free = true;
		return true;
	}
	bool matrix_free = false;  // the noted rows were applied on their lattice (from_recipe)

	RowsOnGpu(const LinearEquation& eq, int num_columns, int dtype, const std::vector<int>* lattice = nullptr)
	{
		if (num_columns >= 1 && from_recipe(eq, num_columns, dtype, lattice)) {
			last_matrix_free() = true;
			return;
		}
		last_matrix_free() = false;
		upload(eq.triplets, eq.rhs, num_columns, dtype, lattice);
	}
	RowsOnGpu(const std::vector<Triplet>& triplets, const std::vector<float>& rhs, int num_columns, int dtype,
	          const std::vector<int>* lattice = nullptr)
	{
		last_matrix_free() = false;
		upload(triplets, rhs, num_columns, dtype, lattice);
	}
	static bool& last_matrix_free()
	{
		static thread_local bool flag = false;
		return flag;
	}
	void upload(const std::vector<Triplet>& triplets, const std::vector<float>& rhs, int num_columns, int dtype,
	            const std::vector<int>* lattice)
	{
		struct { const std::vector<Triplet>& triplets; const std::vector<float>& rhs; } eq{triplets, rhs};
		if (num_columns < 1) { return; }
		const bool nd = lattice && !lattice->empty() && lattice->size() <= 3;
		const std::vector<int> shape = nd ? *lattice : std::vector<int>{num_columns};
		if (CtxCache::cacheable(shape)) {
			ctx = cache().get(shape, dtype);
		} else if (fi_ctx_create(&ctx, static_cast<int>(shape.size()), shape.data(), dtype) == FI_OK) {
			owned = true;
		} else {
			warn("fi_ctx_create");
			ctx = nullptr;
		}
		if (!ctx) { return; }
		const fi_weights none = {1, 1, 0, 0, 0, 0, 0, 0, FI_VALUE_LINEAR_INTERPOLATION, FI_GRADIENT_CELL_EDGES};
		const bool ok = fi_clear_points(ctx) == FI_OK && fi_set_model(ctx, &none) == FI_OK &&
		                fi_add_rows_coo(ctx, static_cast<long>(eq.rhs.size()), static_cast<long>(eq.triplets.size()),
		                                reinterpret_cast<const fi_triplet*>(eq.triplets.data()), eq.rhs.data(), FI_HOST) == FI_OK &&
		                fi_assemble(ctx) == FI_OK;
		if (!ok) {
			warn("assembling the linear equation");
			if (owned) { fi_ctx_destroy(ctx); } else { cache().drop(ctx); }
			owned = false;
			ctx = nullptr;
		}
	}
	RowsOnGpu(const RowsOnGpu&) = delete;
};

// Eigen::BiCGSTAB applies the operator twice per iteration, the CG used here once: a caller's iteration budget
// (src/bipolar_2d.cpp:328: solve_sparse_linear_with_guess(eq, last, 100, 0)) buys the same number of operator
// applications -- two CG steps per requested BiCGSTAB step.  0 keeps meaning "the solver's default".
int cg_budget(int bicgstab_iterations)
{
	if (bicgstab_iterations <= 0) { return 0; }
	return bicgstab_iterations > INT_MAX / 2 ? INT_MAX : 2 * bicgstab_iterations;
}

std::vector<float> iterate(const LinearEquation& eq, const std::vector<float>* guess, int num_columns, int dtype,
                           int max_iterations, float tolerance, const std::vector<int>* lattice = nullptr)
{
	RowsOnGpu gpu(eq, num_columns, dtype, lattice);
	if (!gpu.ctx) { return {}; }
	std::vector<float> out(static_cast<size_t>(num_columns));
	int   iterations = 0;
	float error      = 0;
	if (fi_solve_cg(gpu.ctx, guess ? guess->data() : nullptr, max_iterations, tolerance, out.data(), &iterations, &error,
	                FI_HOST) != FI_OK) {
		warn("solver failed");  // the reference: LOG_F(WARNING, "solver.solve failed"); return {};
		return {};
	}
	return out;
}

// solve_sparse_linear_exact / _fast: the reference factorises A^T A (SimplicialLLT) and returns {} when the
// factorisation fails (sparse_linear.cpp:137-149, 169-172).  Here the normal equations are iterated in fp64:
//   * an unknown without any equation (a zero on the diagonal of A^T A -- the zero pivot that stops LLT) -> {} like the
//     reference; so does a solver breakdown (non-finite values, non-positive curvature);
//   * otherwise the best iterate is returned.  A well-posed but ill-conditioned system (kappa of A^T A grows like
//     side^4..8 for model_2..4 lattices with few data) may not reach the requested relative residual in floating
//     point -- the attainable one is about eps * kappa -- and the factorisation the reference runs would still answer:
//     the iterate is returned with a warning instead of {}.
std::vector<float> solve_direct_like(const LinearEquation& eq, int num_columns, float tolerance)
{
	RowsOnGpu gpu(eq, num_columns, FI_F64);
	if (!gpu.ctx) { return {}; }
	{
		std::vector<double> diag(static_cast<size_t>(num_columns));
		if (fi_get_diag_f64(gpu.ctx, diag.data()) != FI_OK) {
			warn("fi_get_diag_f64");
			return {};
		}
		for (double d : diag) {
			if (!(d > 0.0)) {
				std::fprintf(stderr, "field_interpolation: singular system (an unknown without equations): LLT would fail\n");
				return {};
			}
		}
	}
	std::vector<float> out(static_cast<size_t>(num_columns));
	int   iterations = 0;
	float error      = 0;
	const int budget = num_columns > (1 << 24) ? (1 << 30) : 50 * num_columns + 1000;
	if (fi_solve_cg(gpu.ctx, nullptr, budget, tolerance, out.data(), &iterations, &error, FI_HOST) != FI_OK) {
		warn("solver failed");
		return {};
	}
	if (!(error <= tolerance * 1.0001f)) {
		std::fprintf(stderr, "field_interpolation: iterative stand-in for the factorisation stopped at relative residual %g "
		                     "after %d iterations (asked for %g); returning that iterate\n", error, iterations, tolerance);
	}
	return out;
}

}  // namespace

std::vector<float> solve_sparse_linear_exact(const LinearEquation& eq, int num_columns)
{
	return solve_direct_like(eq, num_columns, 1e-12f);
}

// SimplicialLLT in float in the reference: same system, fp32 accuracy.
std::vector<float> solve_sparse_linear_fast(const LinearEquation& eq, int num_columns)
{
	return solve_direct_like(eq, num_columns, 1e-7f);
}

std::vector<float> solve_sparse_linear_with_guess(const LinearEquation& eq, const std::vector<float>& guess,
                                                  int max_iterations, float error_tolerance)
{
	return iterate(eq, &guess, static_cast<int>(guess.size()), FI_F32, cg_budget(max_iterations), error_tolerance);
}

std::vector<float> jacobi_iterations(const LinearEquation& eq, const std::vector<float>& guess, const int num_iterations,
                                     const float weight)
{
	if (num_iterations <= 0) { return guess; }
	RowsOnGpu gpu(eq, static_cast<int>(guess.size()), FI_F32);
	if (!gpu.ctx) { return {}; }
	std::vector<float> out(guess.size());
	if (fi_jacobi(gpu.ctx, guess.data(), num_iterations, weight, out.data(), FI_HOST) != FI_OK) {
		warn("jacobi_iterations");
		return {};
	}
	return out;
}

std::vector<float> solve_tiled_with_guess(const LinearEquation& eq, const std::vector<float>& guess,
                                          const std::vector<int>& sizes, const SolveOptions& options)
{
	size_t n = 1;
	for (int s : sizes) { n *= static_cast<size_t>(s); }
	if (guess.size() != n) {
		std::fprintf(stderr, "field_interpolation: Incomplete guess.\n");
		return {};
	}
	if (!options.tile) {
		if (!options.cg) { return guess; }
		// (the caller's lattice shapes the context of generic rows too: a context's extent per axis is limited to 2^20, and a
		// 1024^2 system as ONE axis of 2^20 unknowns would be refused)
		return iterate(eq, &guess, static_cast<int>(n), FI_F32, cg_budget(options.max_iterations), options.error_tolerance,
		               sizes.size() <= 3 ? &sizes : nullptr);
	}
	// tile_solver_square (sparse_linear.cpp:246-390, 415-425) on the device: fi_tile_pass solves every tile_size^D
	// tile of the lattice with its couplings to the other tiles taken from the guess, then the iteration starts
	// from the tile solutions (:427-440).  More than 3 lattice dimensions: no tiles to speak of here, plain iteration.
	if (sizes.size() > 3) {
		std::fprintf(stderr, "field_interpolation: SolveOptions.tile needs a lattice of at most 3 dimensions; ignored\n");
		if (!options.cg) { return guess; }
		return iterate(eq, &guess, static_cast<int>(n), FI_F32, cg_budget(options.max_iterations), options.error_tolerance);
	}
	RowsOnGpu gpu(eq, static_cast<int>(n), FI_F32, &sizes);
	if (!gpu.ctx) { return {}; }
	std::vector<float> tiled(n);
	if (fi_tile_pass(gpu.ctx, guess.data(), options.tile_size, tiled.data(), FI_HOST) != FI_OK) {
		warn("tile pre-solver");
		return {};
	}
	if (!options.cg) { return tiled; }
	std::vector<float> out(n);
	int   iterations = 0;
	float error      = 0;
	if (fi_solve_cg(gpu.ctx, tiled.data(), cg_budget(options.max_iterations), options.error_tolerance, out.data(), &iterations,
	                &error, FI_HOST) != FI_OK) {
		warn("solver failed");
		return {};
	}
	return out;
}

// (gpu_field.hpp) whether this thread's last solver call applied the rows' note matrix-free on its lattice
bool last_solve_was_matrix_free() { return RowsOnGpu::last_matrix_free(); }

// Frees the device contexts this thread's stateless calls have cached (gpu_field.hpp).
void clear_context_cache()
{
	cache().clear();
	(void)fi_memory_pool(0, nullptr);  // and the device blocks the library keeps of destroyed contexts
}

namespace detail {

// generate_error_map (field_interpolation.cpp:402-429) through the device: the caller's rows go up like for a solve
// (fi_add_rows_coo), fi_error_map distributes the squared row residuals.  {} when the GPU path fails.
std::vector<float> error_map_on_gpu(const std::vector<Triplet>& triplets, const std::vector<float>& solution,
                                    const std::vector<float>& rhs)
{
	if (solution.empty()) { return {}; }
	RowsOnGpu rows(triplets, rhs, static_cast<int>(solution.size()), FI_F32);
	if (!rows.ctx) { return {}; }
	std::vector<float> blame(solution.size(), 0.0f);
	if (fi_error_map(rows.ctx, solution.data(), blame.data(), FI_HOST) != FI_OK) {
		warn("fi_error_map");
		return {};
	}
	return blame;
}

}  // namespace detail

}  // namespace field_interpolation
