// fi_prim.h -- the device-wide primitives of the assembly (scans, run-length encoding, selection, reductions by key, 64-bit
// pair sorts) straight on rocPRIM, ROCm's own primitive library (rounds 1-4 went through hipCUB, the CUB-shaped layer over
// it).  Same call shape as the library's: a first call with a null workspace returns the bytes it wants.
#pragma once

#include <cstring>
#include <iterator>

#include <rocprim/rocprim.hpp>

#include "fi_internal.h"

namespace fi {
namespace prim {

template <typename In, typename Out>
inline hipError_t exclusive_sum(void* tmp, size_t& bytes, In in, Out out, size_t n, hipStream_t st)
{
	using T = typename std::iterator_traits<Out>::value_type;
	return rocprim::exclusive_scan(tmp, bytes, in, out, T(0), n, rocprim::plus<T>(), st);
}

// runs of equal keys: the distinct keys, the run lengths, the number of runs
template <typename In, typename Unique, typename Counts, typename Runs>
inline hipError_t run_length_encode(void* tmp, size_t& bytes, In in, Unique unique_out, Counts counts_out, Runs runs_out, size_t n,
                                    hipStream_t st)
{
	return rocprim::run_length_encode(tmp, bytes, in, static_cast<unsigned int>(n), unique_out, counts_out, runs_out, st);
}

// out = the indices 0 .. n-1 the predicate accepts, in order; *count_out = how many
template <typename Out, typename Count, typename Pred>
inline hipError_t select_indices(void* tmp, size_t& bytes, Out out, Count count_out, size_t n, Pred pred, hipStream_t st)
{
	return rocprim::select(tmp, bytes, rocprim::counting_iterator<uint32_t>(0u), out, count_out, n, pred, st);
}

// sums of the values of every run of equal keys (keys sorted)
template <typename K, typename V, typename Runs>
inline hipError_t sum_by_key(void* tmp, size_t& bytes, const K* keys, K* unique_out, const V* values, V* sums_out, Runs runs_out, size_t n,
                             hipStream_t st)
{
	return rocprim::reduce_by_key(tmp, bytes, keys, values, static_cast<unsigned int>(n), unique_out, sums_out, runs_out, rocprim::plus<V>(),
	                              rocprim::equal_to<K>(), st);
}

// stable radix sort of (key, value) pairs over key bits [begin_bit, end_bit); 64-bit keys: Onesweep with the workgroup shape
// tuned for the 32-bit sorts of fi_sort.h (1024 threads x 8 items; the library's default merge-sorts below 2^20 items: a dozen
// pairs of small launches where Onesweep takes one histogram, one scan and a pass per digit)
namespace detail {
using onesweep64 = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                              rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<1024, 8>, 8,
                                                                                  rocprim::block_radix_rank_algorithm::match>,
                                              16384>;
}
template <typename V>
inline hipError_t sort_pairs_u64(void* tmp, size_t& bytes, const uint64_t* keys_in, uint64_t* keys_out, const V* values_in, V* values_out,
                                 size_t n, int begin_bit, int end_bit, hipStream_t st)
{
	return rocprim::radix_sort_pairs<detail::onesweep64>(tmp, bytes, keys_in, keys_out, values_in, values_out, static_cast<unsigned int>(n),
	                                                     static_cast<unsigned int>(begin_bit), static_cast<unsigned int>(end_bit), st);
}

}  // namespace prim
}  // namespace fi
