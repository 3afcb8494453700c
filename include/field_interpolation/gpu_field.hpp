// gpu_field.hpp -- the matrix-free fast path for lattice problems (not in the reference).
//
// GpuLatticeField is the GPU twin of LatticeField: same call sequence (add_field_constraints, add_points /
// add_value_constraint / add_gradient_constraint, then a solver), but no row is ever written to host memory:
// everything goes straight to libfi_hip (include/fi_hip.h).
#pragma once

#include <memory>
#include <vector>

#include "field_interpolation.hpp"

struct fi_ctx;

namespace field_interpolation {

class GpuLatticeField
{
public:
	// double_precision: keep all lattice vectors in fp64 (for tolerances below ~1e-5).
	explicit GpuLatticeField(const std::vector<int>& sizes, bool double_precision = false);
	~GpuLatticeField();
	GpuLatticeField(const GpuLatticeField&) = delete;
	GpuLatticeField& operator=(const GpuLatticeField&) = delete;

	const std::vector<int>& sizes() const { return sizes_; }
	size_t num_unknowns() const;

	// Large lattices (the reference's recipe is app code: src/sdf_field.cpp:272-288).  `levels` coarser replicas of
	// the problem are built on the GPU; a zero-length guess then starts from the coarse-to-fine cascade.
	// multigrid: V-cycle preconditioned CG over the levels (needed by SDF problems from oriented points).
	// mixed_precision (double_precision fields): CG in fp64, the V-cycle on an fp32 replica.
	void set_levels(int levels, bool multigrid = false, bool mixed_precision = false);
	// Any solver option of the C ABI by number (include/fi_hip.h FI_OPT_*: e.g. FI_OPT_FIELD_TOLERANCE = 12 stops by the field,
	// FI_OPT_MG_KCYCLE = 13 corrects that many coarse levels by two flexible-CG steps each -- a third of the iterations on SDF
	// problems).  false: the library refused the value.
	bool set_option(int option, double value);

	void add_field_constraints(const Weights& weights);
	bool add_value_constraint(const float pos[], float value, float weight);
	bool add_value_constraint_nearest_neighbor(const float pos[], const float gradient[], float value, float weight);
	bool add_gradient_constraint(const float pos[], const float gradient[], float weight, GradientKernel kernel);
	void add_points(float value_weight, ValueKernel value_kernel, float gradient_weight, GradientKernel gradient_kernel,
	                int num_points, const float positions[], const float* normals, const float* point_weights);
	// The border prior of the reference's SDF application (src/sdf_field.cpp:218-246: options.boundary_weight): every
	// border lattice point is pulled towards its distance to the nearest point added so far.  After add_points.
	bool add_border_prior(float weight);

	// solve_sparse_linear_with_guess / solve_tiled_with_guess / jacobi_iterations of the reference.
	// An empty result means failure (wrong guess length, solver breakdown), as in the reference.
	std::vector<float> solve_with_guess(const std::vector<float>& guess, int max_iterations, float error_tolerance);
	// No guess: starts from the coarse-to-fine cascade when set_levels() built levels, from zero otherwise.
	std::vector<float> solve(int max_iterations, float error_tolerance);
	std::vector<float> solve_tiled_with_guess(const std::vector<float>& guess, const SolveOptions& options);
	std::vector<float> jacobi_iterations(const std::vector<float>& guess, int num_iterations, float weight);
	// generate_error_map(field.eq.triplets, solution, field.eq.rhs) of the reference, from the rows on the device.
	std::vector<float> generate_error_map(const std::vector<float>& solution);

	int    last_iterations() const { return iterations_; }
	float  last_error() const { return error_; }
	size_t num_data_rows() const;   // rows accepted from points (what eq.rhs.size() would have grown by)

private:
	bool assemble();
	fi_ctx*          ctx_ = nullptr;
	std::vector<int> sizes_;
	bool             dirty_ = true;
	int              iterations_ = 0;
	float            error_ = 0;
};

// sdf_from_points without the triplet list.
std::unique_ptr<GpuLatticeField> gpu_sdf_from_points(const std::vector<int>& sizes, const Weights& weights,
                                                     int num_points, const float positions[], const float* normals,
                                                     const float* point_weights);

// The stateless solver calls of sparse_linear.hpp keep device contexts between calls, keyed by shape and precision (at
// most six, systems of up to 2^20 unknowns only; larger systems get a context per call): the per-frame caller's latency
// (src/bipolar_2d.cpp:730).  This frees the calling thread's cached contexts (they are freed at thread exit otherwise)
// and the device memory the library keeps of destroyed contexts (fi_memory_pool).
void clear_context_cache();

// Whether this thread's last stateless solver call (solve_sparse_linear_with_guess, solve_tiled_with_guess, ...) found the
// note add_field_constraints / add_points leave beside the rows (LinearEquation::recipe) still valid and applied those rows
// matrix-free on their lattice -- the path of the reference's own call sequence, sdf_from_points followed by
// solve_tiled_with_guess(field.eq, ...) (src/sdf_field.cpp:251-304) -- instead of uploading every triplet.
bool last_solve_was_matrix_free();

} // namespace field_interpolation
