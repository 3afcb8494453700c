// fi_sort.h -- radix sort of (uint32 key, uint32 value) pairs for the assembly's lists.  rocPRIM's default picks a merge
// sort below 2^20 items: log2(n / block) passes of two small launches each -- 25 launches for the 637 k slots of config 4's
// 64^3 level, where Onesweep needs 8 (a histogram, a scan and a pass per 8 key bits).  An assemble is bound by the number of
// launches it issues (~5 us each, ~560 per step in round 3), so the lists are sorted by Onesweep from 16 k items on.
#pragma once

#include <cstring>
#include <iterator>

#include <rocprim/rocprim.hpp>

#include "fi_internal.h"

namespace fi {

// Onesweep tuned on MI355X for the sizes the assembly sorts (tools/scratch/bench_sort.hip, profiles/r4_ablation.md section
// 8): workgroups of 1024 threads x 8 items and digits of 9 bits -- 1 M pairs of 25-bit keys (the rows of config 4's 256^3
// level) in 3 passes and 76 us against the library default's 4 passes (1024 x 16 items, 8 bits: 62 workgroups for 1 M
// items) and 138 us; 20 M pairs 437 against 663 us.  Digits of 10 bits where that saves a pass (19-20, 28-30 key bits).
namespace sort_detail {
template <unsigned int RadixBits>
using onesweep = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                            rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<1024, 8>,
                                                                                RadixBits, rocprim::block_radix_rank_algorithm::match>,
                                            16384>;
inline bool ten_bit_digits(int bits) { return (bits + 9) / 10 < (bits + 8) / 9; }
}  // namespace sort_detail

inline hipError_t sort_pairs_u32(void* tmp, size_t& bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* values_in,
                                 uint32_t* values_out, unsigned int n, int begin_bit, int end_bit, hipStream_t stream)
{
	const unsigned int b0 = static_cast<unsigned int>(begin_bit), b1 = static_cast<unsigned int>(end_bit);
	if (sort_detail::ten_bit_digits(end_bit - begin_bit)) {
		return rocprim::radix_sort_pairs<sort_detail::onesweep<10>>(tmp, bytes, keys_in, keys_out, values_in, values_out, n, b0, b1, stream);
	}
	return rocprim::radix_sort_pairs<sort_detail::onesweep<9>>(tmp, bytes, keys_in, keys_out, values_in, values_out, n, b0, b1, stream);
}

}  // namespace fi
