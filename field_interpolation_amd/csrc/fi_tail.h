// fi_tail.h -- the small-level engine: the coarse TAIL of a level hierarchy (every level of at most kTailMaxPoints
// unknowns) runs its share of a V-cycle in ONE cooperative launch (fi_tail.hip) instead of 15-25 launches of 4-7 us per
// level -- launches the GPU finishes faster than the host can issue them.  Reference role: the exact small solves of
// tile_solver_square (sparse_linear.cpp:246-390) and the coarse solve of src/sdf_field.cpp:272-288.
#pragma once

#include "fi_internal.h"
#include "fi_transfer.h"

namespace fi {

constexpr int64_t kTailMaxPoints = 1 << 18;  // 64^3 / 512^2: the data part of the operator (3^D coefficients per point) stays
                                             // within reach of the caches; larger levels keep the tiled kernels
constexpr int     kTailMaxLevels = 8;

// One level as the engine sees it.  Operator = model rows matrix-free (model_0 / model_1 / model_2 from the global
// coordinates, like the tiled kernels) + the data rows as 3^D diagonals (`dia`, built from the cell blocks by
// tail_build_operator).
struct TailLevel {
	int      ndim;
	int      n[3];
	int      nn;           // unknowns
	float    w0sq, w1sq, w2sq;
	const float* dia;      // [3^D][nn] or null (no data)
	LevelPair to_coarse;   // transfers to the next level of the tail (unused on the last one)
};

// The V-cycle as a straight-line program of stages, one grid barrier behind each.  a, b, c: input vectors, out / acc:
// outputs; `scale`: the bfloat16 Jacobi-type scaling the stage uses (the polynomial smoother's or the operator's).
enum TailOpKind {
	kTailScale = 0,    // out = s0 * scale * a                                       (first term of either polynomial)
	kTailPolyStep,     // s = scale (A_model z - m z) + z;  zn = s0 z - s1 z_prev + s2 (scale r - s);  out = zn, acc += zn
	                   //   (a = z, b = z_prev or null, c = r)                       (polynomial smoother, ChebEpi mode 0)
	kTailChebStep,     // out = s0 x - s1 x_prev + s2 scale (rhs - A x)              (a = x, b = x_prev or null, c = rhs; mode 2)
	kTailResidual,     // out = rhs - A x                                            (a = x, c = rhs; mode 3)
	kTailRestrict,     // out (level + 1) = R a (level)
	kTailProlongAdd,   // out (level) += P a (level + 1)
};
struct TailOp {
	int          kind, level;
	const float* a;
	const float* b;
	const float* c;
	float*       out;
	float*       acc;
	const unsigned short* scale;
	float        s0, s1, s2;
	int          pad_;
};

bool tail_level_supported(const fi_ctx* c);   // a level the engine can run (geometry, model rows, no triplet rows)
void tail_build_operator(fi_ctx* c);          // `dia` of an assembled level, on the level's stream
TailLevel tail_level_of(const fi_ctx* c);
// runs the program (device array of `nops` stages over `nlev` levels, both in `prog`: TailLevel[kTailMaxLevels] then the ops)
void tail_run(fi_ctx* top, const void* prog, int nlev, int nops, int64_t widest);

}  // namespace fi
