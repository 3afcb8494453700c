// field_interpolation.hpp -- drop-in for field_interpolation/field_interpolation.hpp of emilk/field_interpolation.
//
// Same namespace, names, defaults and signatures as the reference header (field_interpolation.hpp:44-183).
// A field on a regular lattice (1-3 D, coordinates 0 .. size-1 per axis) is estimated from weighted linear
// constraints: model rows (how smooth the field is) and data rows (values / gradients known at points).
// The functions below append those rows to `LatticeField::eq`; the solvers of sparse_linear.hpp minimise the
// weighted squared error.  For large lattices use GpuLatticeField (gpu_field.hpp): it never materialises the
// rows and runs assembly and solve on the GPU.
#pragma once

#include <iosfwd>
#include <vector>

#include "sparse_linear.hpp"

namespace field_interpolation {

const int MAX_DIM = 3;

// How a value known at a point enters the system.
enum class ValueKernel
{
	kNearestNeighbor,      // one row on the closest lattice point, corrected along the gradient
	kLinearInterpolation,  // one row on the 2^D corners of the enclosing cell
};

// How a gradient known at a point enters the system (one row per axis).
enum class GradientKernel
{
	kNearestNeighbor,      // the cell edge starting at the cell origin
	kCellEdges,            // the mean of all cell edges along the axis
	kLinearInterpolation,  // the two nearest edges along the axis, multilinearly weighted
};

// Row weights.  Rule of thumb from the reference: model_0 scales with resolution, model_1 is resolution
// independent, model_2 ~ 1/resolution, model_3 ~ 1/resolution^2.
struct Weights
{
	float data_pos      = 1.00f; // trust in point values
	float data_gradient = 1.00f; // trust in point gradients
	float model_0       = 0.00f; // f = 0            (Tikhonov)
	float model_1       = 0.00f; // f' = 0           (flat)
	float model_2       = 0.50f; // f'' = 0          (smooth)
	float model_3       = 0.00f; // f''' = 0
	float model_4       = 0.00f; // f'''' = 0

	float gradient_smoothness = 0.0f; // neighbouring parallel edges carry equal gradients

	ValueKernel    value_kernel    = ValueKernel::kLinearInterpolation;
	GradientKernel gradient_kernel = GradientKernel::kCellEdges;
};

struct LatticeField
{
	LinearEquation   eq;      // rows accumulated so far
	std::vector<int> sizes;   // extent per axis
	std::vector<int> strides; // index distance between neighbours per axis (x is fastest)

	LatticeField() = default;
	explicit LatticeField(const std::vector<int>& sizes_arg) : sizes(sizes_arg)
	{
		int stride = 1;
		for (int size : sizes) {
			strides.push_back(stride);
			stride *= size;
		}
	}

	int num_dim() const { return static_cast<int>(sizes.size()); }
};

// Model rows for every lattice point and axis.
void add_field_constraints(
	LatticeField*  field,
	const Weights& weights);

// f(pos) = value.  Returns false when the position is ignored (outside the lattice or zero weight).
bool add_value_constraint(
	LatticeField* field,
	const float   pos[],
	float         value,
	float         weight);

// f(nearest lattice point) = value - (pos - nearest) . gradient.  False when outside the lattice.
bool add_value_constraint_nearest_neighbor(
	LatticeField* field,
	const float   pos[],
	const float   gradient[],
	float         value,
	float         weight);

// grad f(pos) = gradient.  False when the position is ignored.
bool add_gradient_constraint(
	LatticeField*  field,
	const float    pos[],
	const float    gradient[],
	float          weight,
	GradientKernel kernel);

// add_value_constraint* (target 0) and add_gradient_constraint for a whole point cloud.
void add_points(
	LatticeField*  field,
	float          value_weight,
	ValueKernel    value_kernel,
	float          gradient_weight,
	GradientKernel gradient_kernel,
	const int      num_points,
	const float    positions[],    // interleaved xyxy.. / xyzxyz..
	const float*   normals,        // may be null
	const float*   point_weights); // may be null

// Signed-distance style field from oriented points: model rows, then the point rows.
LatticeField sdf_from_points(
	const std::vector<int>& sizes,
	const Weights&          weights,
	const int               num_points,
	const float             positions[],
	const float*            normals,
	const float*            point_weights);

// Per-unknown share of the squared row residuals (A x - b)^2.
std::vector<float> generate_error_map(const std::vector<Triplet>& triplets,
    const std::vector<float>& solution, const std::vector<float>& rhs);

// Multilinear resampling of a lattice field onto a lattice of other sizes.
std::vector<float> upscale_field(
    const float* field, const std::vector<int>& small_sizes, const std::vector<int>& large_sizes);

} // namespace field_interpolation
