// sparse_linear.hpp -- drop-in for field_interpolation/sparse_linear.hpp of emilk/field_interpolation.
//
// Same namespace, type names, member names, defaults and function signatures as the reference header
// (sparse_linear.hpp:8-80), so existing callers compile unchanged.  The solver bodies live in
// libfield_interpolation (field_interpolation_amd/cxx/) and run on an MI355X through the C ABI of
// fi_hip.h: the triplets are uploaded as generic sparse rows and A^T A x = A^T b is iterated on the GPU
// (no Eigen, no A^T A matrix).
#pragma once

#include <initializer_list>
#include <iosfwd>
#include <vector>

namespace field_interpolation {

// One coefficient of the sparse system: A[row][col] = value.  12 bytes, like the reference's Triplet.
struct Triplet
{
	int   row, col;
	float value;

	Triplet() {}
	Triplet(int r, int c, float v) : row(r), col(c), value(v) {}
};

// A x = rhs in coordinate form; entries sharing (row, col) add up.
struct LinearEquation
{
	std::vector<Triplet> triplets;
	std::vector<float>   rhs;
};

// Text dump, one equation per line:  rhs = v * xCOL  +  v * xCOL ...
std::ostream& operator<<(std::ostream& os, const LinearEquation& eq);

struct LinearEquationPair
{
	int   column;
	float value;
};

struct Weight { float value; };
struct Rhs    { float value; };

// Appends the row  sum(pair.value * x[pair.column]) = rhs, scaled by weight.  A zero weight adds nothing;
// zero coefficients are skipped; a row without coefficients adds no rhs either.
void add_equation(
	LinearEquation* eq, Weight weight, Rhs rhs, std::initializer_list<LinearEquationPair> pairs);

// Least-squares solution of the triplets (rows = eq.rhs.size(), num_columns unknowns).
// An empty vector means the solver failed (e.g. singular normal equations), as in the reference.
std::vector<float> solve_sparse_linear_fast(const LinearEquation& eq, int num_columns);

std::vector<float> solve_sparse_linear_exact(const LinearEquation& eq, int num_columns);

// Iterative least squares from a starting guess (guess.size() unknowns).
std::vector<float> solve_sparse_linear_with_guess(
	const LinearEquation&     eq,
	const std::vector<float>& guess,
	int                       max_iterations,   // 0: default (twice the problem size)
	float                     error_tolerance); // 0: default (float epsilon)

// Weighted Jacobi sweeps on the normal equations; num_iterations <= 0 returns the guess.
std::vector<float> jacobi_iterations(
	const LinearEquation&     eq,
	const std::vector<float>& guess,
	const int                 num_iterations,
	const float               weight);

struct SolveOptions
{
	bool  tile             = false;   // tile pre-pass before the iterative phase
	int   tile_size        = 16;      // tile side (tile_size^D unknowns)
	bool  cg               = true;    // iterative phase
	int   max_iterations   = 0;       // 0: default
	float error_tolerance  = 1e-3f;   // 0: default (float epsilon)
};

// Guess (+ optional tile pre-pass) + iterative phase on a lattice of the given sizes.
// A guess of the wrong length returns an empty vector.
std::vector<float> solve_tiled_with_guess(
	const LinearEquation&     eq,
	const std::vector<float>& guess,
	const std::vector<int>&   sizes,
	const SolveOptions&       options);

} // namespace field_interpolation
