// fi_sort.h -- radix sort of (uint32 key, uint32 value) pairs for the assembly's lists.  rocPRIM's default picks a merge
// sort below 2^20 items: log2(n / block) passes of two small launches each -- 25 launches for the 637 k slots of config 4's
// 64^3 level, where Onesweep needs 8 (a histogram, a scan and a pass per 8 key bits).  An assemble is bound by the number of
// launches it issues (~5 us each, ~560 per step in round 3), so the lists are sorted by Onesweep from 16 k items on.
#pragma once

#include <rocprim/rocprim.hpp>

#include "fi_internal.h"

namespace fi {

inline hipError_t sort_pairs_u32(void* tmp, size_t& bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* values_in,
                                 uint32_t* values_out, unsigned int n, int begin_bit, int end_bit, hipStream_t stream)
{
	using config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 16384>;
	return rocprim::radix_sort_pairs<config>(tmp, bytes, keys_in, keys_out, values_in, values_out, n, static_cast<unsigned int>(begin_bit),
	                                         static_cast<unsigned int>(end_bit), stream);
}

}  // namespace fi
