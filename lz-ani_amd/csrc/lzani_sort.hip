// lzani_sort.hip -- the one library call of the engine: hipCUB's device radix sort, for the per-genome k-mer lists of
// the join form of candidate detection (long genomes; lzani_kernels_pairs.h: DevWave::join).  A translation unit of
// its own so that the hipCUB templates are instantiated once and stay out of the kernels' compile.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

// Sorts n 64-bit keys by their bits [begin_bit, end_bit).  tmp == nullptr: only reports the temporary bytes needed.
int lzani_sort_keys(const unsigned long long* in, unsigned long long* out, size_t n, int begin_bit, int end_bit,
                    void* tmp, size_t* tmp_bytes, hipStream_t stream)
{
    if (n > 0x7FFFFFF0ull) return (int)hipErrorInvalidValue;      // the caller sorts groups of genomes below 2^31 keys
    return (int)hipcub::DeviceRadixSort::SortKeys(tmp, *tmp_bytes, in, out, (int)n, begin_bit, end_bit, stream);
}
