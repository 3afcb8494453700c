// fi_stencil_common.h -- what the marching kernel (fi_stencil.hip) and the builder of its per-workgroup cell lists
// (fi_stencil_lists.hip) share.
#pragma once

#include "fi_internal.h"

#ifndef FI_BASE_WAVES
#define FI_BASE_WAVES 4  // waves per SIMD the model-only variant is register-allocated for
#endif

namespace fi {

template <typename T>
struct VecOf;
template <>
struct VecOf<float> {
	using V = float4;
	static constexpr int VX = 4;
};
template <>
struct VecOf<double> {
	using V = double2;
	static constexpr int VX = 2;
};


// per-workgroup record lists of the fused kernel from the context's cells (fi_stencil_lists.hip); T = the context's precision
// (m: the context's marching state, or its strip state -- fi_strip.hip -- whose "workgroup" is one wave's strip)
template <typename T>
void build_cell_lists(fi_ctx* c, MarchState& m);

}  // namespace fi
