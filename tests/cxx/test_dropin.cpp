// Exercises the C++ drop-in (include/field_interpolation/*.hpp + libfield_interpolation.so) the way the
// reference's demo app uses its library.  Prints "ok ..." lines; exits non-zero on the first failure.
//   1. src/field_1d.cpp:98-110 verbatim call sequence at the default input (:20-29), checked against the
//      float64 solution recorded in SURVEY.md 8(c);
//   2. src/sdf_field.cpp:212-304 style: sdf_from_points + solve_sparse_linear_exact (rows through the generic
//      GPU path) against GpuLatticeField (matrix-free GPU path) on the same input;
//   3. failure conventions: wrong guess length -> {}, singular system -> {}, num_iterations <= 0 -> guess;
//   4. jacobi_iterations: legacy rows vs fast path; generate_error_map, upscale_field, operator<<;
//   5. the per-frame call pattern of src/bipolar_2d.cpp:323-332 on a 128^2 system: the iteration budget rule (two CG
//      steps per requested BiCGSTAB step), the singular-system {} convention, and the latency of warm calls (the
//      device context is cached per lattice shape).
#include <chrono>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <vector>

#include <fi_hip.h>

#include <field_interpolation/field_interpolation.hpp>
#include <field_interpolation/gpu_field.hpp>
#include <field_interpolation/sparse_linear.hpp>

namespace fi = field_interpolation;

static void require(bool ok, const char* what)
{
	if (!ok) {
		std::printf("FAILED: %s\n", what);
		std::exit(1);
	}
	std::printf("ok   %s\n", what);
}

static float max_rel(const std::vector<float>& a, const std::vector<float>& b)
{
	float num = 0, den = 0;
	for (size_t i = 0; i < a.size(); ++i) {
		num = std::fmax(num, std::fabs(a[i] - b[i]));
		den = std::fmax(den, std::fabs(b[i]));
	}
	return num / den;
}

int main()
{
	// ---- 1. field_1d.cpp -------------------------------------------------------------------------
	{
		struct Point1D { float pos, value, gradient; };
		const std::vector<Point1D> points{{0.2f, 0, +1}, {0.8f, 0, -1}};
		const int   resolution = 12;
		fi::Weights weights;
		fi::LatticeField field{{resolution}};
		for (const auto& point : points) {
			float pos_lattice      = point.pos * (resolution - 1);
			float gradient_lattice = point.gradient / (resolution - 1);
			add_value_constraint(&field, &pos_lattice, point.value, weights.data_pos);
			add_gradient_constraint(&field, &pos_lattice, &gradient_lattice, weights.data_gradient, weights.gradient_kernel);
		}
		add_field_constraints(&field, weights);
		require(field.eq.rhs.size() == 14 && field.eq.triplets.size() == 38, "field_1d: 14 rows / 38 triplets (SURVEY 8c)");
		const size_t num_unknowns = resolution;
		auto interpolated = solve_sparse_linear_exact(field.eq, num_unknowns);
		require(interpolated.size() == num_unknowns, "field_1d: solve_sparse_linear_exact returns a solution");
		const float expected[12] = {-0.1846154f, -0.1006993f, -0.0167832f, 0.0671329f, 0.1230769f, 0.1510490f,
		                            0.1510490f, 0.1230769f, 0.0671329f, -0.0167832f, -0.1006993f, -0.1846154f};
		float worst = 0;
		for (int i = 0; i < 12; ++i) { worst = std::fmax(worst, std::fabs(interpolated[i] - expected[i])); }
		require(worst <= 2e-6f, "field_1d: solution equals the SURVEY 8(c) known answer to 2e-6");
		std::ostringstream os;
		os << field.eq;
		require(os.str().find(" * x") != std::string::npos, "operator<<(LinearEquation) prints equations");
	}

	// ---- 2. sdf_field.cpp style -------------------------------------------------------------------
	const std::vector<int> sizes{24, 20};
	std::vector<float> positions, normals;
	for (int i = 0; i < 160; ++i) {
		const float a = 6.2831853f * i / 160.0f;
		positions.push_back(11.5f + 6.0f * std::cos(a) + 0.1f * std::sin(17.0f * i));
		positions.push_back(9.5f + 6.0f * std::sin(a) + 0.1f * std::cos(13.0f * i));
		normals.push_back(std::cos(a));
		normals.push_back(std::sin(a));
	}
	fi::Weights weights;
	const fi::LatticeField field = fi::sdf_from_points(sizes, weights, 160, positions.data(), normals.data(), nullptr);
	const size_t n = 24 * 20;
	std::vector<float> exact = fi::solve_sparse_linear_exact(field.eq, n);
	require(exact.size() == n, "sdf: solve_sparse_linear_exact (generic rows on the GPU)");
	{
		fi::GpuLatticeField gpu(sizes, /*double_precision=*/true);
		gpu.add_field_constraints(weights);
		gpu.add_points(weights.data_pos, weights.value_kernel, weights.data_gradient, weights.gradient_kernel, 160,
		               positions.data(), normals.data(), nullptr);
		std::vector<float> fast = gpu.solve_with_guess(std::vector<float>(n, 0.0f), 100000, 1e-10f);
		require(fast.size() == n && gpu.last_error() <= 1e-10f, "sdf: GpuLatticeField converges to 1e-10");
		require(max_rel(fast, exact) <= 1e-5f, "sdf: matrix-free path == generic-row path to 1e-5");
		require(gpu.num_data_rows() + (24 - 2) * 20 + 24 * (20 - 2) == field.eq.rhs.size(),
		        "sdf: data rows + closed-form model rows == eq.rhs.size()");
	}
	{
		fi::SolveOptions options;  // defaults: cg, tolerance 1e-3
		auto approx = fi::solve_tiled_with_guess(field.eq, std::vector<float>(n, 0.0f), field.sizes, options);
		require(approx.size() == n, "sdf: solve_tiled_with_guess with default SolveOptions");
		// SolveOptions.tile on materialised rows (sparse_linear.cpp:415-425): the tile pre-solver alone moves the zero
		// guess towards the solution, and tile + CG converges like CG alone
		fi::SolveOptions tiles_only;
		tiles_only.tile = true;
		tiles_only.tile_size = 8;
		tiles_only.cg = false;
		auto pre = fi::solve_tiled_with_guess(field.eq, std::vector<float>(n, 0.0f), field.sizes, tiles_only);
		require(pre.size() == n, "sdf: tile pre-solver on materialised rows returns a field");
		{
			double d0 = 0, d1 = 0;
			for (size_t i = 0; i < n; ++i) {
				d0 += static_cast<double>(exact[i]) * exact[i];
				d1 += (static_cast<double>(pre[i]) - exact[i]) * (static_cast<double>(pre[i]) - exact[i]);
			}
			require(d1 < d0, "sdf: the tile solutions are closer to the solution than the zero guess");
		}
		fi::SolveOptions tiles_cg;
		tiles_cg.tile = true;
		tiles_cg.tile_size = 8;
		tiles_cg.error_tolerance = 1e-7f;
		tiles_cg.max_iterations = 100000;
		auto both = fi::solve_tiled_with_guess(field.eq, std::vector<float>(n, 0.0f), field.sizes, tiles_cg);
		require(both.size() == n && max_rel(both, exact) <= 1e-3f, "sdf: tile pre-solver + CG reaches the solution");
		auto from_exact = fi::solve_sparse_linear_with_guess(field.eq, exact, 50, 1e-6f);
		require(from_exact.size() == n && max_rel(from_exact, exact) <= 1e-3f, "sdf: warm start stays at the solution");
	}

	// ---- 3. failure conventions -------------------------------------------------------------------
	{
		fi::SolveOptions options;
		require(fi::solve_tiled_with_guess(field.eq, std::vector<float>(n - 1, 0.0f), field.sizes, options).empty(),
		        "solve_tiled_with_guess: incomplete guess -> {} (sparse_linear.cpp:402-405)");
		// an inconsistent-direction test of the {} convention: the solver must not be able to reach 1e-12 in 1 step
		require(fi::solve_sparse_linear_with_guess(field.eq, std::vector<float>(n, 0.0f), 1, 1e-12f).size() == n,
		        "solve_sparse_linear_with_guess stops at max_iterations and still returns the iterate");
		const std::vector<float> guess(n, 1.5f);
		require(fi::jacobi_iterations(field.eq, guess, 0, 0.5f) == guess, "jacobi_iterations(0) returns the guess (:220)");
		const float outside[2] = {-5.0f, 3.0f};
		fi::LatticeField f2{{8, 8}};
		require(!add_value_constraint(&f2, outside, 1.0f, 1.0f), "add_value_constraint outside the lattice -> false");
		const float inside[2] = {2.5f, 3.5f};
		const float g[2] = {1, 0};
		require(!add_gradient_constraint(&f2, inside, g, 0.0f, fi::GradientKernel::kCellEdges), "zero weight -> false");
	}

	// ---- 4. Jacobi, error map, upscale ---------------------------------------------------------------
	{
		std::vector<float> guess(n);
		for (size_t i = 0; i < n; ++i) { guess[i] = std::sin(0.37f * i); }
		auto legacy = fi::jacobi_iterations(field.eq, guess, 7, 2.0f / 3.0f);
		fi::GpuLatticeField gpu(sizes);
		gpu.add_field_constraints(weights);
		gpu.add_points(weights.data_pos, weights.value_kernel, weights.data_gradient, weights.gradient_kernel, 160,
		               positions.data(), normals.data(), nullptr);
		auto fast = gpu.jacobi_iterations(guess, 7, 2.0f / 3.0f);
		require(legacy.size() == n && fast.size() == n && max_rel(fast, legacy) <= 2e-5f, "jacobi_iterations: rows == matrix-free");
		auto heat = fi::generate_error_map(field.eq.triplets, exact, field.eq.rhs);
		float total = 0;
		for (float h : heat) { total += h; }
		require(heat.size() == n && total > 0, "generate_error_map");
		{   // the definition itself (blame_j = sum over rows of (a_ij^2 / |a_i|^2) (b_i - a_i x)^2), in double on the host
			std::vector<double> res(field.eq.rhs.begin(), field.eq.rhs.end()), n2(field.eq.rhs.size(), 0.0), want(n, 0.0);
			for (const auto& t : field.eq.triplets) {
				res[t.row] -= static_cast<double>(exact[t.col]) * t.value;
				n2[t.row] += static_cast<double>(t.value) * t.value;
			}
			for (const auto& t : field.eq.triplets) {
				if (n2[t.row] != 0) { want[t.col] += static_cast<double>(t.value) * t.value / n2[t.row] * res[t.row] * res[t.row]; }
			}
			double worst = 0, scale = 0;
			for (size_t i = 0; i < n; ++i) {
				worst = std::max(worst, std::fabs(heat[i] - want[i]));
				scale = std::max(scale, std::fabs(want[i]));
			}
			require(worst <= 1e-4 * scale, "generate_error_map: the device's blame == the definition");
		}
		auto heat_gpu = gpu.generate_error_map(exact);
		require(heat_gpu.size() == n && max_rel(heat_gpu, heat) <= 1e-4f, "generate_error_map: device rows == host triplets");
		fi::SolveOptions tiled;
		tiled.tile = true;
		tiled.tile_size = 8;
		tiled.error_tolerance = 1e-5f;
		auto via_tiles = gpu.solve_tiled_with_guess(std::vector<float>(n, 0.0f), tiled);
		require(via_tiles.size() == n && max_rel(via_tiles, exact) <= 5e-3f, "SolveOptions.tile: tile pre-solver + CG");
		// levels + V-cycle + mixed precision through the C++ fast path
		fi::GpuLatticeField deep(sizes, true);
		deep.set_levels(2, true, true);
		deep.add_field_constraints(weights);
		deep.add_points(weights.data_pos, weights.value_kernel, weights.data_gradient, weights.gradient_kernel, 160,
		                positions.data(), normals.data(), nullptr);
		auto leveled = deep.solve(0, 1e-7f);
		require(leveled.size() == n && deep.last_error() <= 1e-7f && max_rel(leveled, exact) <= 1e-4f,
		        "set_levels(2, multigrid, mixed): V-cycle CG with an fp32 replica");
		// ... and the K-cycle through set_option (FI_OPT_MG_KCYCLE): the same field in no more iterations; a bad value is refused
		{
			const int it_v = deep.last_iterations();
			fi::GpuLatticeField kdeep(sizes, true);
			kdeep.set_levels(2, true, true);
			require(kdeep.set_option(FI_OPT_MG_KCYCLE, 1) && !kdeep.set_option(FI_OPT_MG_KCYCLE, -3), "set_option: accepted and refused values");
			kdeep.add_field_constraints(weights);
			kdeep.add_points(weights.data_pos, weights.value_kernel, weights.data_gradient, weights.gradient_kernel, 160,
			                 positions.data(), normals.data(), nullptr);
			auto kfield = kdeep.solve(0, 1e-7f);
			require(kfield.size() == n && kdeep.last_error() <= 1e-7f && max_rel(kfield, exact) <= 1e-4f && kdeep.last_iterations() <= it_v,
			        "set_option(FI_OPT_MG_KCYCLE): K-cycle CG reaches the same field");
		}
		auto big = fi::upscale_field(exact.data(), sizes, {47, 39});
		require(big.size() == 47u * 39u && std::fabs(big[0] - exact[0]) < 1e-6f && std::fabs(big.back() - exact.back()) < 1e-6f,
		        "upscale_field keeps the corners");
	}
	// ---- 5. per-frame solves: budget, singular systems, latency --------------------------------------------
	{
		const std::vector<int> s128{128, 128};
		std::vector<float> pos, nrm;
		for (int i = 0; i < 600; ++i) {
			const float a = 6.2831853f * i / 600.0f;
			pos.push_back(63.5f + 40.0f * std::cos(a) + 0.2f * std::sin(23.0f * i));
			pos.push_back(63.5f + 40.0f * std::sin(a) + 0.2f * std::cos(19.0f * i));
			nrm.push_back(std::cos(a));
			nrm.push_back(std::sin(a));
		}
		const fi::LatticeField f128 = fi::sdf_from_points(s128, fi::Weights(), 600, pos.data(), nrm.data(), nullptr);
		const size_t n128 = 128 * 128;
		const std::vector<float> zero(n128, 0.0f);
		// budget: solve_sparse_linear_with_guess(eq, guess, k, 0) runs 2k CG steps (one operator application each, like
		// the 2 per BiCGSTAB step of the reference): the same iterate as 2k steps through the C ABI
		// (on the GENERIC rows, like the C ABI context below: an unconverged iterate after exactly 100 steps is compared bit-closely,
		// and the rows' note -- section 6 -- would apply them matrix-free, in another order of fp32 sums)
		setenv("FI_DROPIN_NO_RECIPE", "1", 1);
		auto x50 = fi::solve_sparse_linear_with_guess(f128.eq, zero, 50, 0.0f);
		unsetenv("FI_DROPIN_NO_RECIPE");
		require(x50.size() == n128, "bipolar pattern: solve_sparse_linear_with_guess(eq, guess, 50, 0) returns an iterate");
		{
			auto x50mf = fi::solve_sparse_linear_with_guess(f128.eq, zero, 50, 0.0f);   // the same budget, matrix-free
			require(x50mf.size() == n128 && fi::last_solve_was_matrix_free() && max_rel(x50mf, x50) <= 2e-2f,
			        "budget rule on the matrix-free path: the same 100 CG steps up to fp32 rounding");
		}
		{
			fi_ctx* ctx = nullptr;
			const int shape[1] = {static_cast<int>(n128)};
			const fi_weights none = {1, 1, 0, 0, 0, 0, 0, 0, FI_VALUE_LINEAR_INTERPOLATION, FI_GRADIENT_CELL_EDGES};
			std::vector<float> x100(n128), x50cg(n128);
			int it = 0;
			float err = 0;
			const bool ok = fi_ctx_create(&ctx, 1, shape, FI_F32) == FI_OK && fi_set_model(ctx, &none) == FI_OK &&
			                fi_add_rows_coo(ctx, static_cast<long>(f128.eq.rhs.size()), static_cast<long>(f128.eq.triplets.size()),
			                                reinterpret_cast<const fi_triplet*>(f128.eq.triplets.data()), f128.eq.rhs.data(), FI_HOST) == FI_OK &&
			                fi_assemble(ctx) == FI_OK &&
			                fi_solve_cg(ctx, zero.data(), 100, 0.0f, x100.data(), &it, &err, FI_HOST) == FI_OK;
			require(ok && it == 100, "C ABI: 100 CG steps on the same rows");
			const bool ok2 = fi_solve_cg(ctx, zero.data(), 50, 0.0f, x50cg.data(), &it, &err, FI_HOST) == FI_OK;
			fi_ctx_destroy(ctx);
			require(ok2 && max_rel(x50, x100) <= 1e-6f, "budget rule: 50 requested BiCGSTAB steps == 100 CG steps");
			require(max_rel(x50cg, x100) > 1e-4f, "... and not 50 CG steps");
		}
		// singular system: an unknown no equation touches is the zero pivot that stops SimplicialLLT -> {} (sparse_linear.cpp:169-172)
		{
			fi::LinearEquation eq;
			fi::add_equation(&eq, fi::Weight{1.0f}, fi::Rhs{1.0f}, {{0, 1.0f}, {1, 1.0f}});
			fi::add_equation(&eq, fi::Weight{1.0f}, fi::Rhs{0.0f}, {{0, 1.0f}, {1, -1.0f}});
			require(fi::solve_sparse_linear_exact(eq, 2).size() == 2, "exact: a regular 2 x 2 system is solved");
			require(fi::solve_sparse_linear_exact(eq, 3).empty(), "exact: an unknown without equations -> {} like the failed factorisation");
			require(fi::solve_sparse_linear_fast(eq, 3).empty(), "fast: the same");
			// a consistent system of rank 1 with positive diagonals: the factorisation's outcome is rounding luck in the
			// reference; here the least-squares iterate comes back
			fi::LinearEquation r1;
			fi::add_equation(&r1, fi::Weight{1.0f}, fi::Rhs{2.0f}, {{0, 1.0f}, {1, 1.0f}});
			auto ls = fi::solve_sparse_linear_exact(r1, 2);
			require(ls.size() == 2 && std::fabs(ls[0] + ls[1] - 2.0f) <= 1e-5f, "exact: a consistent rank-deficient system returns a least-squares solution");
		}
		// latency of warm calls (contexts cached per shape): the per-frame pattern of bipolar_2d.cpp:323-332
		{
			using clock = std::chrono::steady_clock;
			std::vector<float> last = fi::solve_sparse_linear_with_guess(f128.eq, zero, 100, 0.0f);  // first call: creates the context
			const int reps = 20;
			auto t0 = clock::now();
			for (int r = 0; r < reps; ++r) { last = fi::solve_sparse_linear_with_guess(f128.eq, last, 100, 0.0f); }
			const double ms_cg = std::chrono::duration<double, std::milli>(clock::now() - t0).count() / reps;
			t0 = clock::now();
			for (int r = 0; r < reps; ++r) { last = fi::jacobi_iterations(f128.eq, last, 100, 0.5f); }
			const double ms_jac = std::chrono::duration<double, std::milli>(clock::now() - t0).count() / reps;
			std::printf("latency 128^2 (%zu rows, %zu triplets), warm-started, context cached: solve_sparse_linear_with_guess(eq, last, 100, 0) "
			            "%.2f ms/call; jacobi_iterations(eq, last, 100, 0.5) %.2f ms/call\n", f128.eq.rhs.size(), f128.eq.triplets.size(), ms_cg, ms_jac);
			require(last.size() == n128 && ms_cg < 200.0 && ms_jac < 200.0, "per-frame calls stay interactive");
		}
	}
	// ---- 6. the reference application's own call sequence, unchanged (src/sdf_field.cpp:251-304): sdf_from_points, then
	//         solve_tiled_with_guess(field.eq, guess, field.sizes, options).  The rows carry the note of how they were made
	//         (LinearEquation::recipe): the solver applies them matrix-free on the lattice and uploads only what the note does
	//         not cover (here: border rows appended with add_equation, sdf_field.cpp:218-246) -- and falls back to the generic
	//         rows when the caller has edited noted rows.
	{
		auto circle = [](int side, int count, std::vector<float>* pos, std::vector<float>* nrm) {
			for (int i = 0; i < count; ++i) {
				const float a = 6.2831853f * i / count, c = 0.5f * (side - 1), r = 0.3f * (side - 1);
				pos->push_back(c + r * std::cos(a) + 0.3f * std::sin(17.0f * i));
				pos->push_back(c + r * std::sin(a) + 0.3f * std::cos(13.0f * i));
				nrm->push_back(std::cos(a));
				nrm->push_back(std::sin(a));
			}
		};
		auto with_border = [](fi::LatticeField* f) {  // a few rows the note does not cover
			const int sx = f->sizes[0], sy = f->sizes[1];
			for (int x = 0; x < sx; x += 7) {
				fi::add_equation(&f->eq, fi::Weight{0.05f}, fi::Rhs{0.3f * sx}, {{x, 1.0f}});
				fi::add_equation(&f->eq, fi::Weight{0.05f}, fi::Rhs{0.3f * sx}, {{(sy - 1) * sx + x, 1.0f}});
			}
		};
		// (a) a small lattice: both paths to the fp32 floor, the same solution
		{
			std::vector<float> pos, nrm;
			circle(96, 600, &pos, &nrm);
			fi::LatticeField f = fi::sdf_from_points({96, 80}, fi::Weights{}, 600, pos.data(), nrm.data(), nullptr);
			with_border(&f);
			fi::SolveOptions o;
			o.error_tolerance = 1e-7f;
			o.max_iterations = 20000;
			const std::vector<float> zero(96 * 80, 0.0f);
			auto a = fi::solve_tiled_with_guess(f.eq, zero, f.sizes, o);
			require(a.size() == zero.size() && fi::last_solve_was_matrix_free(), "recipe: the unchanged call sequence runs matrix-free");
			setenv("FI_DROPIN_NO_RECIPE", "1", 1);
			auto b = fi::solve_tiled_with_guess(f.eq, zero, f.sizes, o);
			unsetenv("FI_DROPIN_NO_RECIPE");
			require(b.size() == zero.size() && !fi::last_solve_was_matrix_free(), "recipe: FI_DROPIN_NO_RECIPE takes the generic rows");
			std::printf("   matrix-free vs generic rows at 96 x 80: max relative difference %.2e\n", max_rel(a, b));
			require(max_rel(a, b) <= 5e-4f, "recipe: matrix-free == generic rows (fp32 CG to 1e-7)");
			auto c = fi::solve_sparse_linear_with_guess(f.eq, zero, 20000, 1e-7f);
			require(c.size() == zero.size() && fi::last_solve_was_matrix_free() && max_rel(a, c) <= 5e-4f,
			        "recipe: solve_sparse_linear_with_guess takes the lattice from the note");
			// a caller that edits a noted row in place: the checksum fails, the generic path answers
			fi::LatticeField g = f;
			g.eq.triplets.front().value *= 1.5f;
			auto d = fi::solve_tiled_with_guess(g.eq, zero, g.sizes, o);
			require(d.size() == zero.size() && !fi::last_solve_was_matrix_free(), "recipe: edited rows fall back to the generic path");
			// ... or a noted row's right-hand side (the data rows' targets: the note would re-make them from the points)
			fi::LatticeField g2 = f;
			const size_t border_rows = 2 * ((96 + 6) / 7);  // (with_border's rows come last and are nobody's: the row in front of them)
			g2.eq.rhs[g2.eq.rhs.size() - border_rows - 1] += 0.5f;
			auto d2 = fi::solve_tiled_with_guess(g2.eq, zero, g2.sizes, o);
			require(d2.size() == zero.size() && !fi::last_solve_was_matrix_free(), "recipe: an edited right-hand side falls back to the generic path");
			// a copy of the field keeps the note; rows appended to an EXISTING noted row cannot be expressed: generic
			fi::LatticeField h = f;
			h.eq.triplets.emplace_back(0, 5, 0.25f);
			auto e = fi::solve_tiled_with_guess(h.eq, zero, h.sizes, o);
			require(e.size() == zero.size() && !fi::last_solve_was_matrix_free(), "recipe: a triplet added to a noted row falls back");
		}
		// (b) BASELINE config 3's shape at 1024^2, 20 000 oriented points: what the note saves
		{
			std::vector<float> pos, nrm;
			circle(1024, 20000, &pos, &nrm);
			const auto t0 = std::chrono::steady_clock::now();
			fi::LatticeField f = fi::sdf_from_points({1024, 1024}, fi::Weights{}, 20000, pos.data(), nrm.data(), nullptr);
			with_border(&f);
			const auto t1 = std::chrono::steady_clock::now();
			fi::SolveOptions o;   // the reference's defaults: cg, tolerance 1e-3
			const std::vector<float> zero(1024 * 1024, 0.0f);
			auto warm = fi::solve_tiled_with_guess(f.eq, zero, f.sizes, o);
			const auto t2 = std::chrono::steady_clock::now();
			auto a = fi::solve_tiled_with_guess(f.eq, zero, f.sizes, o);
			const auto t3 = std::chrono::steady_clock::now();
			require(a.size() == zero.size() && fi::last_solve_was_matrix_free(), "recipe: 1024^2 runs matrix-free");
			setenv("FI_DROPIN_NO_RECIPE", "1", 1);
			auto b = fi::solve_tiled_with_guess(f.eq, zero, f.sizes, o);
			const auto t4 = std::chrono::steady_clock::now();
			unsetenv("FI_DROPIN_NO_RECIPE");
			require(b.size() == zero.size() && !fi::last_solve_was_matrix_free(), "recipe: 1024^2 generic rows");
			auto ms = [](std::chrono::steady_clock::time_point x, std::chrono::steady_clock::time_point y) {
				return std::chrono::duration<double, std::milli>(y - x).count();
			};
			std::printf("   1024^2, %zu rows / %zu triplets: host row building %.0f ms; solve_tiled_with_guess matrix-free %.1f ms "
			            "(first call %.1f), generic rows %.1f ms; max relative difference %.2e\n", f.eq.rhs.size(), f.eq.triplets.size(),
			            ms(t0, t1), ms(t2, t3), ms(t1, t2), ms(t3, t4), max_rel(a, b));
			require(max_rel(a, b) <= 5e-2f, "recipe: the two paths agree at the reference's default tolerance (1e-3 residual)");
		}
	}
	// (c) the note in other orders and dimensions: a 3-D lattice whose rows were made by hand in the order points, rows of the
	//     caller's own, model rows, more points (value rows with another kernel) -- and a 1-D lattice; both paths, the same field
	{
		auto both_paths = [&](const fi::LatticeField& f, const char* what, float tol_between) {
			fi::SolveOptions o;
			o.error_tolerance = 1e-7f;
			o.max_iterations = 20000;
			size_t n = 1;
			for (int s : f.sizes) { n *= static_cast<size_t>(s); }
			const std::vector<float> zero(n, 0.0f);
			auto a = fi::solve_tiled_with_guess(f.eq, zero, f.sizes, o);
			const bool mf = fi::last_solve_was_matrix_free();
			setenv("FI_DROPIN_NO_RECIPE", "1", 1);
			auto b = fi::solve_tiled_with_guess(f.eq, zero, f.sizes, o);
			unsetenv("FI_DROPIN_NO_RECIPE");
			char msg[160];
			std::snprintf(msg, sizeof msg, "recipe: %s runs matrix-free and equals the generic rows", what);
			require(a.size() == n && b.size() == n && mf && !fi::last_solve_was_matrix_free() && max_rel(a, b) <= tol_between, msg);
		};
		{
			const int S[3] = {24, 20, 28};
			std::vector<float> pos, nrm, pos2;
			for (int i = 0; i < 300; ++i) {
				const float a = 6.2831853f * i / 300, b = 3.1415927f * (0.15f + 0.7f * ((i * 37) % 300) / 300.0f);
				const float d[3] = {std::sin(b) * std::cos(a), std::sin(b) * std::sin(a), std::cos(b)};
				for (int k = 0; k < 3; ++k) {
					pos.push_back(0.5f * (S[k] - 1) + 0.3f * (S[k] - 1) * d[k]);
					nrm.push_back(d[k]);
				}
			}
			for (int i = 0; i < 40; ++i) {
				for (int k = 0; k < 3; ++k) { pos2.push_back(1.5f + std::fmod(7.31f * i + 2.17f * k, static_cast<float>(S[k]) - 3.0f)); }
			}
			fi::LatticeField f{{S[0], S[1], S[2]}};
			fi::Weights w;
			fi::add_points(&f, w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, 300, pos.data(), nrm.data(), nullptr);
			fi::add_equation(&f.eq, fi::Weight{0.2f}, fi::Rhs{3.0f}, {{0, 1.0f}});                       // nobody's rows, in between
			fi::add_equation(&f.eq, fi::Weight{0.2f}, fi::Rhs{-2.0f}, {{S[0] * S[1] * S[2] - 1, 1.0f}, {5, 0.5f}});
			fi::add_field_constraints(&f, w);
			fi::add_points(&f, 0.7f, fi::ValueKernel::kLinearInterpolation, 0.0f, w.gradient_kernel, 40, pos2.data(), nullptr, nullptr);
			both_paths(f, "a 3-D lattice (points, own rows, model, more points)", 2e-3f);
		}
		{
			fi::LatticeField f{{300}};
			fi::Weights w;
			w.model_1 = 0.1f;
			fi::add_field_constraints(&f, w);
			std::vector<float> pos, nrm;
			for (int i = 0; i < 25; ++i) {
				pos.push_back(3.3f + 11.7f * i);
				nrm.push_back(i % 2 ? 1.0f : -1.0f);
			}
			fi::add_points(&f, w.data_pos, w.value_kernel, w.data_gradient, w.gradient_kernel, 25, pos.data(), nrm.data(), nullptr);
			fi::add_equation(&f.eq, fi::Weight{1.0f}, fi::Rhs{4.0f}, {{150, 1.0f}});
			both_paths(f, "a 1-D lattice", 2e-3f);
		}
	}
	fi::clear_context_cache();   // (and a call after it still works: the cache refills)
	{
		fi::LinearEquation eq;
		fi::add_equation(&eq, fi::Weight{1.0f}, fi::Rhs{2.0f}, {{0, 1.0f}});
		fi::add_equation(&eq, fi::Weight{1.0f}, fi::Rhs{4.0f}, {{1, 2.0f}});
		auto x = fi::solve_sparse_linear_exact(eq, 2);
		require(x.size() == 2 && std::fabs(x[0] - 2.0f) < 1e-5f && std::fabs(x[1] - 2.0f) < 1e-5f, "solve after clear_context_cache");
	}
	std::printf("all drop-in checks passed\n");
	return 0;
}
