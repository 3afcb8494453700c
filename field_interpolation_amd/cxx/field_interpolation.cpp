// Host side of the drop-in library: the row builders of field_interpolation.hpp.
//
// These functions exist because the reference's API hands the rows to the caller (`LatticeField::eq` is a
// public member that applications read: src/field_1d.cpp:86, src/sdf_field.cpp:673-674), so they have to be
// produced in host memory; the arithmetic follows field_interpolation.cpp:15-400 line by line in meaning
// (fp32, same operation order) and is checked against the oracle.  Solving happens on the GPU
// (sparse_linear.cpp of this directory); large lattices should use GpuLatticeField, which skips this file.
#include "field_interpolation/field_interpolation.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>

#include <fi_hip.h>

#include "recipe.hpp"

namespace field_interpolation {

namespace {

// the note beside the rows (recipe.hpp): a copy of the old note plus one segment (the note is shared by copies of eq)
void note_segment(LatticeField* field, detail::Segment&& seg)
{
	auto next = std::make_shared<detail::Recipe>();
	if (field->eq.recipe && field->eq.recipe->sizes == field->sizes) { *next = *field->eq.recipe; }
	next->sizes = field->sizes;
	seg.row1  = field->eq.rhs.size();
	seg.trip1 = field->eq.triplets.size();
	if (seg.trip1 == seg.trip0) { return; }  // nothing appended
	seg.checksum = detail::sample_checksum(field->eq.triplets, seg.trip0, seg.trip1, field->eq.rhs, seg.row0, seg.row1);
	next->segments.push_back(std::move(seg));
	field->eq.recipe = std::move(next);
}

[[noreturn]] void fatal(const char* what)
{
	std::fprintf(stderr, "field_interpolation: %s\n", what);  // the reference aborts through loguru here
	std::abort();
}

struct Sample {
	int   index;
	float weight;
};

// Multilinear weights of the 2^D lattice points around pos (field_interpolation.cpp:15-55): corner c takes
// the upper neighbour on axis d when bit d of c is set; corners outside [0, size - margin) are dropped.
int corner_samples(Sample* out, const LatticeField& f, const float* pos, int margin)
{
	const int D = f.num_dim();
	if (D < 1 || D > MAX_DIM) { fatal("lattice dimension must be 1..3"); }
	int   lo[MAX_DIM];
	float frac[MAX_DIM];
	for (int d = 0; d < D; ++d) {
		lo[d]   = static_cast<int>(std::floor(pos[d]));
		frac[d] = pos[d] - static_cast<float>(lo[d]);
	}
	int n = 0;
	for (int c = 0; c < (1 << D); ++c) {
		Sample s{0, 1.0f};
		bool   inside = true;
		for (int d = 0; d < D; ++d) {
			const int up = (c >> d) & 1;
			const int q  = lo[d] + up;
			s.index += f.strides[d] * q;
			s.weight *= up ? frac[d] : 1.0f - frac[d];
			inside = inside && 0 <= q && q + margin < f.sizes[d];
		}
		if (inside) { out[n++] = s; }
	}
	return n;
}

int cell_origin(const LatticeField& f, const float* pos)  // field_interpolation.cpp:110-121
{
	int index = 0;
	for (int d = 0; d < f.num_dim(); ++d) {
		const int q = static_cast<int>(std::floor(pos[d]));
		if (q < 0 || q + 1 >= f.sizes[d]) { return -1; }
		index += q * f.strides[d];
	}
	return index;
}

}  // namespace

bool add_value_constraint(LatticeField* field, const float pos[], float value, float weight)
{
	if (weight == 0) { return false; }
	Sample s[8];
	const int n = corner_samples(s, *field, pos, 0);
	if (n == 0) { return false; }
	const int row = static_cast<int>(field->eq.rhs.size());
	float total = 0;
	for (int k = 0; k < n; ++k) {
		const float c = s[k].weight * weight;
		field->eq.triplets.emplace_back(row, s[k].index, c);
		total += c;
	}
	field->eq.rhs.emplace_back(total * value);
	return true;
}

bool add_value_constraint_nearest_neighbor(LatticeField* field, const float pos[], const float gradient[], float value,
                                           float weight)
{
	int   index = 0;
	float along = 0;
	for (int d = 0; d < field->num_dim(); ++d) {
		const int q = static_cast<int>(std::round(pos[d]));
		if (q < 0 || q >= field->sizes[d]) { return false; }
		along += (pos[d] - static_cast<float>(q)) * gradient[d];
		index += q * field->strides[d];
	}
	add_equation(&field->eq, Weight{weight}, Rhs{value - along}, {{index, 1.0f}});
	return true;
}

bool add_gradient_constraint(LatticeField* field, const float pos[], const float gradient[], float weight,
                             GradientKernel kernel)
{
	if (weight == 0) { return false; }
	const int D = field->num_dim();
	switch (kernel) {
	case GradientKernel::kNearestNeighbor: {
		const int o = cell_origin(*field, pos);
		if (o < 0) { return false; }
		for (int d = 0; d < D; ++d) {
			add_equation(&field->eq, Weight{weight}, Rhs{gradient[d]}, {{o, -1.0f}, {o + field->strides[d], +1.0f}});
		}
		return true;
	}
	case GradientKernel::kCellEdges: {
		const int o = cell_origin(*field, pos);
		if (o < 0) { return false; }
		const int corners = 1 << D;
		for (int d = 0; d < D; ++d) {
			const int   row  = static_cast<int>(field->eq.rhs.size());
			const float term = weight * 2.0f / static_cast<float>(corners);
			for (int c = 0; c < corners; ++c) {
				int col = o;
				for (int a = 0; a < D; ++a) { col += field->strides[a] * ((c >> a) & 1); }
				field->eq.triplets.emplace_back(row, col, (((c >> d) & 1) ? +1.0f : -1.0f) * term);
			}
			field->eq.rhs.emplace_back(weight * gradient[d]);
		}
		return true;
	}
	case GradientKernel::kLinearInterpolation: {
		float shifted[MAX_DIM] = {0, 0, 0};
		for (int d = 0; d < D; ++d) { shifted[d] = pos[d] - 0.5f; }
		Sample s[8];
		const int n = corner_samples(s, *field, shifted, 1);
		if (n == 0) { return false; }
		for (int d = 0; d < D; ++d) {
			const int row = static_cast<int>(field->eq.rhs.size());
			float total = 0;
			for (int k = 0; k < n; ++k) {
				const float c = s[k].weight * weight;
				field->eq.triplets.emplace_back(row, s[k].index, -c);
				field->eq.triplets.emplace_back(row, s[k].index + field->strides[d], +c);
				total += c;
			}
			field->eq.rhs.emplace_back(total * gradient[d]);
		}
		return true;
	}
	}
	fatal("Unknown gradient kernel");
}

void add_field_constraints(LatticeField* field, const Weights& w)
{
	const int D = field->num_dim();
	long n = 1;
	for (int s : field->sizes) { n *= s; }
	LinearEquation* eq = &field->eq;
	detail::Segment seg;
	seg.kind    = detail::Segment::kModel;
	seg.row0    = eq->rhs.size();
	seg.trip0   = eq->triplets.size();
	seg.weights = w;
	struct Note {
		LatticeField* f; detail::Segment* s;
		~Note() { note_segment(f, std::move(*s)); }
	} note{field, &seg};
	for (int index = 0; index < n; ++index) {
		int coord[MAX_DIM] = {0, 0, 0};
		for (int d = 0, rest = index; d < D; ++d) {
			coord[d] = rest % field->sizes[d];
			rest /= field->sizes[d];
		}
		for (int d = 0; d < D; ++d) {
			const int size = field->sizes[d], s = field->strides[d], c = coord[d];
			if (w.model_0 > 0) { add_equation(eq, Weight{w.model_0}, Rhs{0}, {{index, 1.0f}}); }
			if (w.model_1 > 0 && c + 1 < size) {
				add_equation(eq, Weight{w.model_1}, Rhs{0}, {{index, -1.0f}, {index + s, +1.0f}});
			}
			if (w.model_2 > 0 && c + 2 < size) {
				add_equation(eq, Weight{w.model_2}, Rhs{0}, {{index, +1.0f}, {index + s, -2.0f}, {index + 2 * s, +1.0f}});
			}
			if (w.model_3 > 0 && c + 3 < size) {
				add_equation(eq, Weight{w.model_3}, Rhs{0},
				             {{index, +1.0f}, {index + s, -3.0f}, {index + 2 * s, +3.0f}, {index + 3 * s, -1.0f}});
			}
			if (w.model_4 > 0 && c + 4 < size) {
				add_equation(eq, Weight{w.model_4}, Rhs{0},
				             {{index, +1.0f}, {index + s, -4.0f}, {index + 2 * s, +6.0f}, {index + 3 * s, -4.0f},
				              {index + 4 * s, +1.0f}});
			}
			if (w.gradient_smoothness > 0 && c + 1 < size) {
				for (int o = 0; o < D; ++o) {
					if (o == d || coord[o] + 1 >= field->sizes[o]) { continue; }
					const int so = field->strides[o];
					add_equation(eq, Weight{w.gradient_smoothness}, Rhs{0},
					             {{index, -1.0f}, {index + s, +1.0f}, {index + so, +1.0f}, {index + so + s, -1.0f}});
				}
			}
		}
	}
}

void add_points(LatticeField* field, float value_weight, ValueKernel value_kernel, float gradient_weight,
                GradientKernel gradient_kernel, const int num_points, const float positions[], const float* normals,
                const float* point_weights)
{
	const int D = field->num_dim();
	detail::Segment seg;
	seg.kind  = detail::Segment::kPoints;
	seg.row0  = field->eq.rhs.size();
	seg.trip0 = field->eq.triplets.size();
	seg.value_weight    = value_weight;
	seg.value_kernel    = value_kernel;
	seg.gradient_weight = gradient_weight;
	seg.gradient_kernel = gradient_kernel;
	seg.num_points      = num_points;
	if (num_points > 0 && D >= 1 && D <= MAX_DIM) {
		seg.positions.assign(positions, positions + static_cast<size_t>(num_points) * D);
		if (normals) { seg.normals.assign(normals, normals + static_cast<size_t>(num_points) * D); }
		if (point_weights) { seg.point_weights.assign(point_weights, point_weights + num_points); }
	}
	struct Note {
		LatticeField* f; detail::Segment* s;
		~Note() { note_segment(f, std::move(*s)); }
	} note{field, &seg};
	for (int i = 0; i < num_points; ++i) {
		const float  w   = point_weights ? point_weights[i] : 1.0f;
		const float* pos = positions + static_cast<size_t>(i) * D;
		const float* g   = normals ? normals + static_cast<size_t>(i) * D : nullptr;
		if (value_kernel == ValueKernel::kNearestNeighbor) {
			if (!normals) { fatal("add_points: the nearest-neighbour value kernel needs normals"); }
			add_value_constraint_nearest_neighbor(field, pos, g, 0.0f, w * value_weight);
		} else {
			add_value_constraint(field, pos, 0.0f, w * value_weight);
		}
		if (normals) { add_gradient_constraint(field, pos, g, w * gradient_weight, gradient_kernel); }
	}
}

LatticeField sdf_from_points(const std::vector<int>& sizes, const Weights& weights, const int num_points,
                             const float positions[], const float* normals, const float* point_weights)
{
	if (!positions) { fatal("sdf_from_points: positions is null"); }
	LatticeField field{sizes};
	add_field_constraints(&field, weights);
	add_points(&field, weights.data_pos, weights.value_kernel, weights.data_gradient, weights.gradient_kernel, num_points,
	           positions, normals, point_weights);
	return field;
}

// generate_error_map (field_interpolation.cpp:402-429): the caller's rows are uploaded (fi_add_rows_coo) and the blame of
// the squared row residuals is computed on the device (fi_error_map), like every other consumer of the rows.
namespace detail {
std::vector<float> error_map_on_gpu(const std::vector<Triplet>& triplets, const std::vector<float>& solution,
                                    const std::vector<float>& rhs);  // sparse_linear.cpp
}
std::vector<float> generate_error_map(const std::vector<Triplet>& triplets, const std::vector<float>& solution,
                                      const std::vector<float>& rhs)
{
	for (const Triplet& t : triplets) {  // the reference indexes without checks; a bad index must not reach the device
		if (t.row < 0 || static_cast<size_t>(t.row) >= rhs.size() || t.col < 0 || static_cast<size_t>(t.col) >= solution.size()) {
			fatal("generate_error_map: triplet index out of range");
		}
	}
	std::vector<float> blame = detail::error_map_on_gpu(triplets, solution, rhs);
	// The reference's loop cannot fail and its callers index the result without a check: a device failure here ends the
	// program with the library's message (like the index check above) instead of handing back an empty vector.  There is no
	// host fallback in this library by design.
	if (blame.size() != solution.size()) { fatal("generate_error_map: the device path failed (see the message above)"); }
	return blame;
}

std::vector<float> upscale_field(const float* field, const std::vector<int>& small_sizes,
                                 const std::vector<int>& large_sizes)
{
	if (small_sizes.size() != large_sizes.size()) { fatal("upscale_field: dimension mismatch"); }
	size_t n = 1;
	for (int s : large_sizes) { n *= static_cast<size_t>(s); }
	std::vector<float> out(n);
	if (fi_upscale_field(field, static_cast<int>(small_sizes.size()), small_sizes.data(), large_sizes.data(), out.data(),
	                     FI_HOST) != FI_OK) {
		std::fprintf(stderr, "field_interpolation: upscale_field: %s\n", fi_last_error());
		return {};
	}
	return out;
}

}  // namespace field_interpolation
