// GpuLatticeField: the matrix-free fast path (include/field_interpolation/gpu_field.hpp) over fi_hip.h.
#include "field_interpolation/gpu_field.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>

#include <fi_hip.h>

namespace field_interpolation {

namespace {
void warn(const char* what) { std::fprintf(stderr, "field_interpolation: %s: %s\n", what, fi_last_error()); }
fi_weights to_c(const Weights& w)
{
	return fi_weights{w.data_pos, w.data_gradient, w.model_0, w.model_1, w.model_2, w.model_3, w.model_4,
	                  w.gradient_smoothness, static_cast<int>(w.value_kernel), static_cast<int>(w.gradient_kernel)};
}
}  // namespace

GpuLatticeField::GpuLatticeField(const std::vector<int>& sizes, bool double_precision) : sizes_(sizes)
{
	if (fi_ctx_create(&ctx_, static_cast<int>(sizes.size()), sizes.data(), double_precision ? FI_F64 : FI_F32) != FI_OK) {
		warn("fi_ctx_create");
		std::abort();  // the reference CHECK_Fs the dimensionality (field_interpolation.cpp:24)
	}
}

GpuLatticeField::~GpuLatticeField() { fi_ctx_destroy(ctx_); }

void GpuLatticeField::set_levels(int levels, bool multigrid, bool mixed_precision)
{
	if (fi_set_option(ctx_, FI_OPT_LEVELS, levels) != FI_OK || fi_set_option(ctx_, FI_OPT_MULTIGRID, multigrid ? 1 : 0) != FI_OK ||
	    fi_set_option(ctx_, FI_OPT_MIXED_PRECISION, mixed_precision ? 1 : 0) != FI_OK) {
		warn("set_levels");
	}
	dirty_ = true;
}

bool GpuLatticeField::set_option(int option, double value)
{
	if (fi_set_option(ctx_, option, value) != FI_OK) {
		warn("set_option");
		return false;
	}
	dirty_ = true;
	return true;
}

size_t GpuLatticeField::num_unknowns() const
{
	size_t n = 1;
	for (int s : sizes_) { n *= static_cast<size_t>(s); }
	return n;
}

void GpuLatticeField::add_field_constraints(const Weights& weights)
{
	const fi_weights w = to_c(weights);
	if (fi_set_model(ctx_, &w) != FI_OK) { warn("fi_set_model"); }
	dirty_ = true;
}

void GpuLatticeField::add_points(float value_weight, ValueKernel value_kernel, float gradient_weight,
                                 GradientKernel gradient_kernel, int num_points, const float positions[],
                                 const float* normals, const float* point_weights)
{
	if (fi_add_points(ctx_, num_points, positions, normals, point_weights, nullptr, value_weight,
	                  static_cast<int>(value_kernel), gradient_weight, static_cast<int>(gradient_kernel), FI_HOST) != FI_OK) {
		warn("add_points");
		std::abort();  // CHECK_NOTNULL_F / ABORT_F in the reference (field_interpolation.cpp:238,361)
	}
	dirty_ = true;
}

bool GpuLatticeField::add_border_prior(float weight)
{
	if (weight == 0) { return false; }
	if (fi_add_border_prior(ctx_, weight) != FI_OK) {
		warn("add_border_prior");
		return false;
	}
	dirty_ = true;
	return true;
}

bool GpuLatticeField::add_value_constraint(const float pos[], float value, float weight)
{
	if (weight == 0) { return false; }
	for (size_t d = 0; d < sizes_.size(); ++d) {  // some corner of the cell floor(pos) lies inside the lattice
		const float fl = std::floor(pos[d]);
		if (!(fl >= -1.0f && fl <= static_cast<float>(sizes_[d] - 1))) { return false; }
	}
	if (fi_add_points(ctx_, 1, pos, nullptr, nullptr, &value, weight, FI_VALUE_LINEAR_INTERPOLATION, 0.0f,
	                  FI_GRADIENT_CELL_EDGES, FI_HOST) != FI_OK) {
		warn("add_value_constraint");
		return false;
	}
	dirty_ = true;
	return true;
}

bool GpuLatticeField::add_value_constraint_nearest_neighbor(const float pos[], const float gradient[], float value,
                                                            float weight)
{
	for (size_t d = 0; d < sizes_.size(); ++d) {
		const float q = std::round(pos[d]);
		if (!(q >= 0.0f && q <= static_cast<float>(sizes_[d] - 1))) { return false; }
	}
	if (fi_add_points(ctx_, 1, pos, gradient, nullptr, &value, weight, FI_VALUE_NEAREST_NEIGHBOR, 0.0f,
	                  FI_GRADIENT_CELL_EDGES, FI_HOST) != FI_OK) {
		warn("add_value_constraint_nearest_neighbor");
		return false;
	}
	dirty_ = true;
	return true;
}

bool GpuLatticeField::add_gradient_constraint(const float pos[], const float gradient[], float weight,
                                              GradientKernel kernel)
{
	if (weight == 0) { return false; }
	const bool lin = kernel == GradientKernel::kLinearInterpolation;
	bool any_sample = !lin;
	for (size_t d = 0; d < sizes_.size(); ++d) {
		const float fl = std::floor(lin ? pos[d] - 0.5f : pos[d]);
		if (lin) {
			// kept samples need 0 <= q and q + 1 < size for q in {fl, fl + 1}
			if (!(fl >= -1.0f && fl + 1.0f < static_cast<float>(sizes_[d]))) { return false; }
		} else if (!(fl >= 0.0f && fl + 1.0f < static_cast<float>(sizes_[d]))) {
			return false;  // cell_index, field_interpolation.cpp:116
		}
	}
	(void)any_sample;
	if (fi_add_points(ctx_, 1, pos, gradient, nullptr, nullptr, 0.0f, FI_VALUE_LINEAR_INTERPOLATION, weight,
	                  static_cast<int>(kernel), FI_HOST) != FI_OK) {
		warn("add_gradient_constraint");
		std::abort();  // unknown kernel: ABORT_F in the reference
	}
	dirty_ = true;
	return true;
}

bool GpuLatticeField::assemble()
{
	if (!dirty_) { return true; }
	if (fi_assemble(ctx_) != FI_OK) {
		warn("fi_assemble");
		return false;
	}
	dirty_ = false;
	return true;
}

size_t GpuLatticeField::num_data_rows() const
{
	fi_stats st{};
	fi_get_stats(ctx_, &st);
	return static_cast<size_t>(st.num_data_rows + st.num_generic_rows);
}

std::vector<float> GpuLatticeField::solve_with_guess(const std::vector<float>& guess, int max_iterations,
                                                     float error_tolerance)
{
	if (guess.size() != num_unknowns() || !assemble()) { return {}; }
	std::vector<float> out(guess.size());
	if (fi_solve_cg(ctx_, guess.data(), max_iterations, error_tolerance, out.data(), &iterations_, &error_, FI_HOST) !=
	    FI_OK) {
		warn("solver failed");
		return {};
	}
	return out;
}

std::vector<float> GpuLatticeField::solve(int max_iterations, float error_tolerance)
{
	if (!assemble()) { return {}; }
	std::vector<float> out(num_unknowns());
	if (fi_solve_cg(ctx_, nullptr, max_iterations, error_tolerance, out.data(), &iterations_, &error_, FI_HOST) != FI_OK) {
		warn("solver failed");
		return {};
	}
	return out;
}

std::vector<float> GpuLatticeField::solve_tiled_with_guess(const std::vector<float>& guess, const SolveOptions& options)
{
	if (guess.size() != num_unknowns()) {
		std::fprintf(stderr, "field_interpolation: Incomplete guess.\n");  // sparse_linear.cpp:402-405
		return {};
	}
	std::vector<float> start = guess;
	if (options.tile) {  // tile_solver_square pre-pass (sparse_linear.cpp:415-425)
		if (!assemble()) { return {}; }
		if (fi_tile_pass(ctx_, guess.data(), options.tile_size, start.data(), FI_HOST) != FI_OK) {
			warn("tile pre-solver");
			return {};
		}
	}
	if (!options.cg) { return start; }
	return solve_with_guess(start, options.max_iterations, options.error_tolerance);
}

std::vector<float> GpuLatticeField::jacobi_iterations(const std::vector<float>& guess, int num_iterations, float weight)
{
	if (num_iterations <= 0) { return guess; }
	if (guess.size() != num_unknowns() || !assemble()) { return {}; }
	std::vector<float> out(guess.size());
	if (fi_jacobi(ctx_, guess.data(), num_iterations, weight, out.data(), FI_HOST) != FI_OK) {
		warn("jacobi_iterations");
		return {};
	}
	return out;
}

std::vector<float> GpuLatticeField::generate_error_map(const std::vector<float>& solution)
{
	if (solution.size() != num_unknowns() || !assemble()) { return {}; }
	std::vector<float> out(solution.size());
	if (fi_error_map(ctx_, solution.data(), out.data(), FI_HOST) != FI_OK) {
		warn("generate_error_map");
		return {};
	}
	return out;
}

std::unique_ptr<GpuLatticeField> gpu_sdf_from_points(const std::vector<int>& sizes, const Weights& weights,
                                                     int num_points, const float positions[], const float* normals,
                                                     const float* point_weights)
{
	if (!positions) {
		std::fprintf(stderr, "field_interpolation: sdf_from_points: positions is null\n");
		std::abort();
	}
	std::unique_ptr<GpuLatticeField> field(new GpuLatticeField(sizes));
	field->add_field_constraints(weights);
	field->add_points(weights.data_pos, weights.value_kernel, weights.data_gradient, weights.gradient_kernel, num_points,
	                  positions, normals, point_weights);
	return field;
}

}  // namespace field_interpolation
