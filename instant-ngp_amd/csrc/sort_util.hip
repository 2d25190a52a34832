// sort_util.hip -- key/value radix sort of the occupancy-grid update's density samples by cell index (library sort: hipCUB / rocPRIM).
// The density network is evaluated on ~10^6 samples per update whose cells come from a multiplicative hash of the sample number
// (testbed_nerf.cu:2476-2592, generate_grid_samples_nerf_nonuniform): in generation order consecutive samples are far apart and every
// hash-grid level misses the caches.  Sorted by (cascade, Morton cell index) consecutive samples are neighbours in space, like the samples
// of a ray.  The update's result does not depend on the order (each sample is splatted into its own cell with atomicMax).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include "ngp_kernels.hpp"

namespace ngp {

struct GridSamplePos { float x, y, z; };

size_t grid_sample_sort_temp_bytes(uint32_t n) {
	size_t bytes = 0;
	(void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const GridSamplePos*)nullptr, (GridSamplePos*)nullptr, (int)n, 0, 32, (hipStream_t)nullptr);
	return bytes;
}
int grid_sample_sort(hipStream_t s, void* temp, size_t temp_bytes, const uint32_t* idx_in, uint32_t* idx_out, const float* pos_in, float* pos_out, uint32_t n, uint32_t begin_bit, uint32_t end_bit) {
	if (n == 0) return 0;
	return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, idx_in, idx_out, (const GridSamplePos*)pos_in, (GridSamplePos*)pos_out, (int)n, (int)begin_bit, (int)end_bit, s) == hipSuccess ? 0 : 1;
}

} // namespace ngp
