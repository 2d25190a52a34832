// sort_util.hip -- key/value radix sort of the occupancy-grid update's density samples by cell index (library sort: rocPRIM).
// The density network is evaluated on ~10^6 samples per update whose cells come from a multiplicative hash of the sample number
// (testbed_nerf.cu:2476-2592, generate_grid_samples_nerf_nonuniform): in generation order consecutive samples are far apart and every
// hash-grid level misses the caches.  Sorted by (cascade, Morton cell index) consecutive samples are neighbours in space, like the samples
// of a ray.  The update's result does not depend on the order (each sample is splatted into its own cell with atomicMax).
// rocPRIM's default configuration sorts up to 2^20 items -- exactly one cascade's sample count -- with its merge sort (a block sort + ten merge
// passes: ~12 launches, 170 us whatever the key width, profiles/r04_exp_grid_sort_bits.log); MergeSortLimit = 0 selects Onesweep: one histogram
// launch + one pass per 8 key bits (two for the 15 bits of a cascade's 4x4x4-cell blocks).
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "ngp_kernels.hpp"

namespace ngp {

struct GridSamplePos { float x, y, z; };
using GridSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;

size_t grid_sample_sort_temp_bytes(uint32_t n) {
	size_t bytes = 0;
	(void)rocprim::radix_sort_pairs<GridSortConfig>(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const GridSamplePos*)nullptr, (GridSamplePos*)nullptr, (size_t)n, 0u, 32u, (hipStream_t)nullptr);
	return bytes;
}
int grid_sample_sort(hipStream_t s, void* temp, size_t temp_bytes, const uint32_t* idx_in, uint32_t* idx_out, const float* pos_in, float* pos_out, uint32_t n, uint32_t begin_bit, uint32_t end_bit) {
	if (n == 0) return 0;
	return rocprim::radix_sort_pairs<GridSortConfig>(temp, temp_bytes, idx_in, idx_out, (const GridSamplePos*)pos_in, (GridSamplePos*)pos_out, (size_t)n, begin_bit, end_bit, s) == hipSuccess ? 0 : 1;
}

// SDF ground truth: the query points of a batch ordered by (cost class, Morton curve) (csrc/sdf_kernels.hip) -- (31-bit key, point index) pairs
size_t sdf_point_sort_temp_bytes(uint32_t n) {
	size_t bytes = 0;
	(void)rocprim::radix_sort_pairs<GridSortConfig>(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, 0u, 31u, (hipStream_t)nullptr);
	return bytes;
}
int sdf_point_sort(hipStream_t s, void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* idx_in, uint32_t* idx_out, uint32_t n) {
	if (n == 0) return 0;
	return rocprim::radix_sort_pairs<GridSortConfig>(temp, temp_bytes, keys_in, keys_out, idx_in, idx_out, (size_t)n, 0u, 31u, s) == hipSuccess ? 0 : 1;
}

} // namespace ngp
