// fi_tail.h -- the small-level engine: the coarse TAIL of a level hierarchy -- every level of at most kTailMaxPoints unknowns
// below the first such level -- runs its share of a V-cycle in ONE launch of ONE workgroup (fi_tail.hip): all the tail's
// vectors live in LDS, a stage of the cycle ends in __syncthreads, and nothing crosses a kernel boundary between the
// restricted residual coming in and the correction going out.  (Round 4 ran the stages in a cooperative kernel over many
// workgroups: a grid barrier across the 8 XCDs cost what the launch it replaced cost.  tools/micro/graph_chain.hip, round 5:
// a dependent chain of tiny kernels costs 2.7 us per kernel as stream launches and 2.6 us replayed as a hipGraph -- the cost is
// the kernel boundary itself -- against 0.74 us per stage inside one workgroup.)
// Reference role: the exact small solves of tile_solver_square (sparse_linear.cpp:246-390) and the coarse solve of
// src/sdf_field.cpp:272-288.
#pragma once

#include "fi_internal.h"
#include "fi_transfer.h"

namespace fi {

constexpr int64_t kTailMaxPoints = 4096;   // 64^2 / 16^3: five vectors of every tail level fit the 160 KB of LDS of one CU
constexpr int     kTailMaxLevels = 6;
constexpr int     kTailThreads   = 1024;
constexpr int     kTailVectors   = 5;      // per level: right-hand side, result, residual, two work vectors

// One level as the engine sees it.  Operator = model rows matrix-free (model_0 / model_1 / model_2 from the global
// coordinates, like the tiled kernels) + the data rows as 3^D diagonals (`dia`, built from the cell blocks by
// tail_build_operator; global memory: read-only, cache-resident).
struct TailLevel {
	int      ndim;
	int      n[3];
	int      nn;           // unknowns
	// LDS layout of the level (offsets in floats): kTailVectors vectors of `vstride` = nn + 2 guard floats each -- vector v's
	// point 0 at base + v * vstride, `guard` = two rows (2-D) / two planes (3-D) of zeros on either side, never written: a
	// neighbour index needs no clamp, what lies beyond a vector only ever meets a zero coefficient --, then the points' packed
	// coordinates (ctab, nn words) and the model rows' coefficients per coordinate (ktab: 5 floats per coordinate, axis after axis)
	int      base, vstride, guard, ctab, ktab;
	float    w0sq, w1sq, w2sq;
	const float* dia;      // [3^D][nn] or null (no data)
	LevelPair to_coarse;   // transfers to the next level of the tail (unused on the last one)
};
// floats of LDS a level of extents n takes
inline int tail_level_floats(int ndim, const int* n)
{
	const int nn = n[0] * n[1] * (ndim > 2 ? n[2] : 1);
	const int guard = 2 * (ndim > 2 ? n[0] * n[1] : n[0]);
	return kTailVectors * (nn + 2 * guard) + nn + 5 * (n[0] + n[1] + (ndim > 2 ? n[2] : 0));
}

// The V-cycle as a straight-line program of stages, one workgroup barrier behind each.  a, b, c: input vectors, out / acc:
// outputs -- LDS offsets in floats, -1: none; `scale`: the bfloat16 Jacobi-type scaling the stage uses (the polynomial
// smoother's or the operator's; global memory).
enum TailOpKind {
	kTailScale = 0,    // out = s0 * scale * a                                       (first term of either polynomial)
	kTailPolyStep,     // s = scale (A_model z - m z) + z;  zn = s0 z - s1 z_prev + s2 (scale r - s);  out = zn, acc += zn
	                   //   (a = z, b = z_prev or none, c = r)                       (polynomial smoother, ChebEpi mode 0)
	kTailChebStep,     // out = s0 x - s1 x_prev + s2 scale (rhs - A x)              (a = x, b = x_prev or none, c = rhs; mode 2)
	kTailResidual,     // out = rhs - A x                                            (a = x, c = rhs; mode 3)
	kTailRestrict,     // out (level + 1) = R a (level)
	kTailProlongAdd,   // out (level) += P a (level + 1)
};
struct TailOp {
	int          kind, level;
	int          a, b, c, out, acc;
	int          pad_;
	const unsigned short* scale;
	float        s0, s1, s2;
	int          pad2_;
};

bool tail_level_supported(const fi_ctx* c);   // a level the engine can run (size, geometry, model rows, no triplet rows)
void tail_build_operator(fi_ctx* c);          // `dia` of an assembled level, on the level's stream
TailLevel tail_level_of(const fi_ctx* c);
// runs the program (device array of `nops` stages over `nlev` levels, both in `prog`: TailLevel[kTailMaxLevels] then the ops):
// b (global, top level) -> LDS vector 0 of level 0, the stages, LDS vector 1 of level 0 -> x (global); lds_floats: the sum of
// tail_level_floats over the levels
void tail_run(fi_ctx* top, const void* prog, int nlev, int nops, int lds_floats, const float* b, float* x);

}  // namespace fi
