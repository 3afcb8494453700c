// Host side of the drop-in library: sparse_linear.hpp on top of libfi_hip (include/fi_hip.h).
//
// The reference builds Eigen matrices here (sparse_linear.cpp:59-113) and calls SimplicialLLT / BiCGSTAB.
// This file uploads the triplets as generic sparse rows (fi_add_rows_coo) and iterates A^T A x = A^T b on
// the GPU: Jacobi-preconditioned CG with the reference's stop rule; "exact"/"fast" iterate to a tight
// tolerance in fp64 / fp32.  Failure conventions follow the reference: log + empty vector.
#include "field_interpolation/sparse_linear.hpp"

#include <cstdio>
#include <ostream>

#include <fi_hip.h>

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

// A context that holds nothing but the caller's rows: 1-D "lattice" of num_columns unknowns (or the caller's lattice,
// which only the tile pre-solver looks at), no model rows.
struct RowsOnGpu {
	fi_ctx* ctx = nullptr;
	RowsOnGpu(const LinearEquation& eq, int num_columns, int dtype, const std::vector<int>* lattice = nullptr)
	{
		const int flat[1] = {num_columns};
		const bool nd = lattice && !lattice->empty() && lattice->size() <= 3;
		if (num_columns < 1 || fi_ctx_create(&ctx, nd ? static_cast<int>(lattice->size()) : 1, nd ? lattice->data() : flat, dtype) != FI_OK) {
			warn("fi_ctx_create");
			ctx = nullptr;
			return;
		}
		const fi_weights none = {1, 1, 0, 0, 0, 0, 0, 0, FI_VALUE_LINEAR_INTERPOLATION, FI_GRADIENT_CELL_EDGES};
		const bool ok = fi_set_model(ctx, &none) == FI_OK &&
		                fi_add_rows_coo(ctx, static_cast<long>(eq.rhs.size()), static_cast<long>(eq.triplets.size()),
		                                reinterpret_cast<const fi_triplet*>(eq.triplets.data()), eq.rhs.data(), FI_HOST) == FI_OK &&
		                fi_assemble(ctx) == FI_OK;
		if (!ok) {
			warn("assembling the linear equation");
			fi_ctx_destroy(ctx);
			ctx = nullptr;
		}
	}
	~RowsOnGpu() { fi_ctx_destroy(ctx); }
	RowsOnGpu(const RowsOnGpu&) = delete;
};

std::vector<float> iterate(const LinearEquation& eq, const std::vector<float>* guess, int num_columns, int dtype,
                           int max_iterations, float tolerance, bool must_converge)
{
	RowsOnGpu gpu(eq, num_columns, dtype);
	if (!gpu.ctx) { return {}; }
	std::vector<float> out(static_cast<size_t>(num_columns));
	int   iterations = 0;
	float error      = 0;
	if (fi_solve_cg(gpu.ctx, guess ? guess->data() : nullptr, max_iterations, tolerance, out.data(), &iterations, &error,
	                FI_HOST) != FI_OK) {
		warn("solver failed");  // the reference: LOG_F(WARNING, "solver.solve failed"); return {};
		return {};
	}
	if (must_converge && !(error <= tolerance * 1.0001f)) {
		std::fprintf(stderr, "field_interpolation: solver did not converge (residual %g after %d iterations)\n", error,
		             iterations);
		return {};
	}
	return out;
}

}  // namespace

// The reference factorises A^T A (SimplicialLLT, double) and returns {} when that fails, e.g. for a singular
// system.  Here: fp64 CG to 1e-12; a system CG cannot drive there (singular / inconsistent) returns {}.
std::vector<float> solve_sparse_linear_exact(const LinearEquation& eq, int num_columns)
{
	return iterate(eq, nullptr, num_columns, FI_F64, 50 * num_columns + 1000, 1e-12f, true);
}

// SimplicialLLT in float in the reference: same system, fp32 accuracy.
std::vector<float> solve_sparse_linear_fast(const LinearEquation& eq, int num_columns)
{
	return iterate(eq, nullptr, num_columns, FI_F64, 50 * num_columns + 1000, 1e-7f, true);
}

std::vector<float> solve_sparse_linear_with_guess(const LinearEquation& eq, const std::vector<float>& guess,
                                                  int max_iterations, float error_tolerance)
{
	return iterate(eq, &guess, static_cast<int>(guess.size()), FI_F32, max_iterations, error_tolerance, false);
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
		return iterate(eq, &guess, static_cast<int>(n), FI_F32, options.max_iterations, options.error_tolerance, false);
	}
	// tile_solver_square (sparse_linear.cpp:246-390, 415-425) on the device: fi_tile_pass solves every tile_size^D
	// tile of the lattice with its couplings to the other tiles taken from the guess, then the iteration starts
	// from the tile solutions (:427-440).  More than 3 lattice dimensions: no tiles to speak of here, plain iteration.
	if (sizes.size() > 3) {
		std::fprintf(stderr, "field_interpolation: SolveOptions.tile needs a lattice of at most 3 dimensions; ignored\n");
		if (!options.cg) { return guess; }
		return iterate(eq, &guess, static_cast<int>(n), FI_F32, options.max_iterations, options.error_tolerance, false);
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
	if (fi_solve_cg(gpu.ctx, tiled.data(), options.max_iterations, options.error_tolerance, out.data(), &iterations, &error,
	                FI_HOST) != FI_OK) {
		warn("solver failed");
		return {};
	}
	return out;
}

}  // namespace field_interpolation
