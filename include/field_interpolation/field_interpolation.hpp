// field_interpolation.hpp -- drop-in for field_interpolation/field_interpolation.hpp of emilk/field_interpolation.
//
// Same namespace, names, member order, defaults and signatures as the reference header
// (field_interpolation.hpp:44-183).  A field on a regular lattice (1-3 D, coordinates 0 .. size-1 per axis) is
// estimated from weighted linear constraints: model rows (how smooth the field is) and data rows (values /
// gradients known at points).  The functions below append those rows to `LatticeField::eq`; the solvers of
// sparse_linear.hpp minimise the weighted squared error.  For large lattices use GpuLatticeField
// (gpu_field.hpp): it never materialises the rows and runs assembly and solve on the GPU.
#pragma once

#include <iosfwd>
#include <vector>

#include "sparse_linear.hpp"

namespace field_interpolation {

const int MAX_DIM = 3;

enum class ValueKernel  // how a value known at a point becomes a row
{
	kNearestNeighbor,      // closest lattice point, corrected along the gradient
	kLinearInterpolation,  // the 2^D corners of the enclosing cell
};

enum class GradientKernel  // how a gradient known at a point becomes one row per axis
{
	kNearestNeighbor,      // the cell edge at the cell origin
	kCellEdges,            // mean of all cell edges along the axis
	kLinearInterpolation,  // the nearest edges along the axis, multilinearly weighted
};

// Row weights.  model_k asks for a vanishing k-th difference (0: Tikhonov, 1: flat, 2: smooth, ...); with the
// resolution r, model_0 grows like r, model_1 is resolution independent, model_2 ~ 1/r, model_3 ~ 1/r^2.
struct Weights
{
	float data_pos = 1.00f;
	float data_gradient = 1.00f;
	float model_0 = 0.00f;
	float model_1 = 0.00f;
	float model_2 = 0.50f;
	float model_3 = 0.00f;
	float model_4 = 0.00f;
	float gradient_smoothness = 0.0f;  // neighbouring parallel edges carry equal gradients
	ValueKernel value_kernel = ValueKernel::kLinearInterpolation;
	GradientKernel gradient_kernel = GradientKernel::kCellEdges;
};

struct LatticeField
{
	LinearEquation eq;         // rows accumulated so far
	std::vector<int> sizes;    // extent per axis
	std::vector<int> strides;  // index distance between neighbours per axis (x is the fastest)

	LatticeField() = default;

	explicit LatticeField(const std::vector<int>& extents) : sizes(extents)
	{
		int step = 1;
		for (int extent : sizes) {
			strides.push_back(step);
			step *= extent;
		}
	}

	int num_dim() const { return static_cast<int>(sizes.size()); }
};

// ---- model rows -------------------------------------------------------------------------------------------

// Every enabled smoothness order, for every lattice point and axis.
void add_field_constraints(LatticeField* field, const Weights& weights);

// ---- data rows: all return false when the position is ignored (outside the lattice, zero weight) ----------

// f(pos) = value
bool add_value_constraint(LatticeField* field, const float pos[], float value, float weight);

// f(nearest lattice point) = value - (pos - nearest) . gradient
bool add_value_constraint_nearest_neighbor(LatticeField* field, const float pos[], const float gradient[], float value,
                                           float weight);

// grad f(pos) = gradient
bool add_gradient_constraint(LatticeField* field, const float pos[], const float gradient[], float weight,
                             GradientKernel kernel);

// A whole point cloud: value rows with target 0 and, when normals are given, gradient rows.  positions are
// interleaved (xyxy.. / xyzxyz..); normals and point_weights may be null.
void add_points(LatticeField* field, float value_weight, ValueKernel value_kernel, float gradient_weight,
                GradientKernel gradient_kernel, const int num_points, const float positions[], const float* normals,
                const float* point_weights);

// Signed-distance style field from oriented points: the model rows, then add_points with the weights' kernels.
LatticeField sdf_from_points(const std::vector<int>& sizes, const Weights& weights, const int num_points,
                             const float positions[], const float* normals, const float* point_weights);

// ---- helpers ------------------------------------------------------------------------------------------------

// Per-unknown share of the squared row residuals (rhs - A x)^2, split by squared coefficient.
std::vector<float> generate_error_map(const std::vector<Triplet>& triplets, const std::vector<float>& solution,
                                      const std::vector<float>& rhs);

// Multilinear resampling of a lattice field onto a lattice of other sizes.
std::vector<float> upscale_field(const float* field, const std::vector<int>& small_sizes,
                                 const std::vector<int>& large_sizes);

}  // namespace field_interpolation
