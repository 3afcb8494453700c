// fi_transfer.h -- the geometry of a level pair and the per-axis taps of interpolation / restriction, shared by the
// transfer kernels of fi_solver.hip and the small-level engine of fi_tail.hip.
#pragma once

#include "fi_internal.h"

namespace fi {

// Coarse -> fine interpolation between two levels of the multilevel hierarchy, per axis (fi_ctx::cc):
//   vertex-centred (odd fine extent): fine point 2i coincides with coarse point i, odd fine points take the mean of their
//     two coarse neighbours (an even extent halved this way: the last fine point copies its only neighbour);
//   cell-centred (even fine extent): coarse point j sits between fine 2j and 2j+1; fine 2j takes 3/4 of coarse j and
//     1/4 of j-1, fine 2j+1 takes 3/4 of j and 1/4 of j+1; the first and the last fine point extrapolate (5/4, -1/4).
struct LevelPair {
	int ndim;
	int nf[3], nc[3];  // GLOBAL extents of the fine and the coarse lattice
	int cc[3];         // the axis was halved cell-centred
	// slabs (slowest axis L = ndim-1): the kernels walk `f_planes` owned fine planes starting at global plane
	// f_z0 / `c_planes` owned coarse planes from c_z0; local storage of either level starts at global plane *_base
	int f_z0, f_planes, f_base;
	int c_z0, c_planes, c_base;
};


// the two coarse points fine index f interpolates from along one axis, and their weights
template <typename T>
__device__ inline void prolong_taps(int f, int nc, int cc, int* i0, int* i1, T* w0, T* w1)
{
	const int j = f >> 1;
	if (cc) {
		const int nb = (f & 1) ? j + 1 : j - 1;
		const bool in = nb >= 0 && nb < nc;
		*i0 = j;
		*i1 = in ? nb : ((f & 1) ? j - 1 : j + 1);
		*w0 = in ? T(0.75) : T(1.25);
		*w1 = in ? T(0.25) : T(-0.25);
	} else {
		const int c0 = j > nc - 1 ? nc - 1 : j;
		*i0 = c0;
		*i1 = c0 + 1 < nc ? c0 + 1 : c0;
		*w1 = (f & 1) ? T(0.5) : T(0);
		*w0 = T(1) - *w1;
	}
}


// restriction = transpose of the interpolation.  Vertex-centred axis: coarse point c gathers fine 2c (weight 1) and 2c-1,
// 2c+1 (weight 1/2; the last coarse point also takes the full weight of a fine point beyond it).  Cell-centred axis: fine
// 2c-1 .. 2c+2 with (1/4, 3/4, 3/4, 1/4); the end points' extrapolation puts 5/4 of fine 0 on coarse 0 and -1/4 of it on
// coarse 1 (mirrored at the other end): five taps.  Indices are relative to `base` and clamped where the weight is 0.
constexpr int kRTaps = 5;
template <typename T>
__device__ inline void restrict_taps(int c, int nf, int nc, int cc, int base, int* f, T* w)
{
	if (cc) {
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int ff = 2 * c - 1 + k;
			const bool in = ff >= 0 && ff < nf;
			T ww = (k == 1 || k == 2) ? T(0.75) : T(0.25);
			if (in && (ff == 0 || ff == nf - 1)) { ww = T(1.25); }  // c == 0 / c == nc - 1: the extrapolated end point
			f[k] = (in ? ff : 2 * c) - base;
			w[k] = in ? ww : T(0);
		}
		const bool lo = c == 1, hi = c == nc - 2;  // (extents >= 8: never both)
		f[4] = (lo ? 0 : (hi ? nf - 1 : 2 * c)) - base;
		w[4] = (lo || hi) ? T(-0.25) : T(0);
	} else {
#pragma unroll
		for (int k = 0; k < 3; ++k) {
			const int ff = 2 * c + k - 1;
			const bool in = ff >= 0 && ff < nf;
			T ww = (k == 1) ? T(1) : T(0.5);
			if (k == 2 && c + 1 >= nc && in) { ww = T(1); }  // fine point 2c+1 when coarse c+1 does not exist
			f[k] = (in ? ff : 2 * c) - base;
			w[k] = in ? ww : T(0);
		}
		f[3] = f[4] = 2 * c - base;
		w[3] = w[4] = T(0);
	}
}


}  // namespace fi
