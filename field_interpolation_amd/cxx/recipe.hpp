// recipe.hpp (internal to libfield_interpolation) -- how a LinearEquation's rows were made, recorded beside the rows.
//
// The reference's application calls sdf_from_points(...) and then solve_tiled_with_guess(field.eq, ...) (src/sdf_field.cpp:251-304):
// only `eq` reaches the solver, a COO list of every model row of the lattice (config 3: 100 M triplets, 1.2 GB).  The row
// builders of this library (add_field_constraints, add_points) note WHAT they appended -- model weights, or copies of the
// point arrays -- and where (row / triplet ranges, a checksum of sampled triplets); the solvers use the matrix-free lattice
// path (fi_set_model + fi_add_points, the stencil kernels) for the ranges that are still as recorded and upload only the rows
// nobody vouches for as triplets (fi_add_rows_coo).  A caller that edits the recorded rows in place fails the checksum and
// gets the generic path, as before (the checksum SAMPLES the noted rows -- 4096 triplets and 4096 right-hand sides per range, the counts:
// a caller that rewrites noted rows in place should reset eq.recipe; the reference's callers only append).  The header stays
// source-compatible: LinearEquation gains a trailing shared_ptr.
#pragma once

#include <cstdint>
#include <memory>
#include <vector>

#include "field_interpolation/field_interpolation.hpp"

namespace field_interpolation {
namespace detail {

struct Segment {
	enum Kind { kModel, kPoints } kind = kModel;
	size_t row0 = 0, row1 = 0, trip0 = 0, trip1 = 0;  // the rows / triplets this call appended
	uint64_t checksum = 0;                            // of up to 4096 triplets sampled from [trip0, trip1) and as many right-hand sides
	Weights weights;                                  // kModel
	float value_weight = 0, gradient_weight = 0;      // kPoints
	ValueKernel value_kernel = ValueKernel::kLinearInterpolation;
	GradientKernel gradient_kernel = GradientKernel::kCellEdges;
	int num_points = 0;
	std::vector<float> positions, normals, point_weights;  // copies (normals / point_weights empty when null)
};

struct Recipe {
	std::vector<int>     sizes;
	std::vector<Segment> segments;
};

// of the triplets [a, b) and the right-hand sides of the rows [r0, r1): up to 4096 samples of each, the counts
inline uint64_t sample_checksum(const std::vector<Triplet>& t, size_t a, size_t b, const std::vector<float>& rhs, size_t r0, size_t r1)
{
	uint64_t h = 1469598103934665603ull;
	static_assert(sizeof(float) == 4, "float");
	auto triplet = [&](size_t i) {
		uint32_t v;
		__builtin_memcpy(&v, &t[i].value, 4);
		const uint64_t w[3] = {static_cast<uint64_t>(static_cast<uint32_t>(t[i].row)), static_cast<uint64_t>(static_cast<uint32_t>(t[i].col)), v};
		for (uint64_t x : w) { h = (h ^ x) * 1099511628211ull; }
	};
	auto value = [&](size_t i) {
		uint32_t v;
		__builtin_memcpy(&v, &rhs[i], 4);
		h = (h ^ static_cast<uint64_t>(v)) * 1099511628211ull;
	};
	const size_t n = b - a, step = n > 4096 ? n / 4096 : 1;
	for (size_t i = a; i < b; i += step) { triplet(i); }
	if (n > 0) { triplet(b - 1); }  // (both ends of a range are always among the samples)
	const size_t m = r1 - r0, rstep = m > 4096 ? m / 4096 : 1;
	for (size_t i = r0; i < r1; i += rstep) { value(i); }
	if (m > 0) { value(r1 - 1); }
	return h ^ static_cast<uint64_t>(n) ^ (static_cast<uint64_t>(m) << 32);
}

}  // namespace detail
}  // namespace field_interpolation
