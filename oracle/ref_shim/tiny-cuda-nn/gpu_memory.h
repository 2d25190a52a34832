// oracle/ref_shim (TEST INFRASTRUCTURE ONLY; see tiny-cuda-nn/common.h here): what src/triangle_bvh.cu needs of tcnn's gpu_memory.h in order to compile and RUN on the CPU.
// GPUMemory is a host vector; linear_kernel(kernel, shmem, stream, n, args...) runs the "__global__" function once per element with blockIdx.x = element, blockDim.x = 1,
// threadIdx.x = 0, which is what `blockIdx.x * blockDim.x + threadIdx.x` needs.  Nothing here stands in for arithmetic.
#pragma once
#include <tiny-cuda-nn/common.h>
#include <stdexcept>
#include <utility>
#include <vector>
typedef void* cudaStream_t;
namespace tcnn {
template <typename T> class GPUMemory {
public:
	GPUMemory() {}
	explicit GPUMemory(size_t n) : m(n) {}
	T* data() const { return const_cast<T*>(m.data()); }
	size_t size() const { return m.size(); }
	size_t get_num_elements() const { return m.size(); }
	size_t bytes() const { return m.size() * sizeof(T); }
	void resize(size_t n) { m.resize(n); }
	void enlarge(size_t n) { if (n > m.size()) m.resize(n); }
	void memset(int v) { std::memset((void*)m.data(), v, bytes()); }
	void copy_from_host(const T* p, size_t n) { std::copy(p, p + n, m.begin()); }
	void copy_from_host(const std::vector<T>& v) { copy_from_host(v.data(), v.size()); }
	void copy_to_host(T* p, size_t n) const { std::copy(m.begin(), m.begin() + n, p); }
	void copy_to_host(std::vector<T>& v) const { copy_to_host(v.data(), v.size()); }
	void resize_and_copy_from_host(const T* p, size_t n) { m.assign(p, p + n); }
	void resize_and_copy_from_host(const std::vector<T>& v) { m = v; }
	void free_memory() { m.clear(); m.shrink_to_fit(); }
private:
	std::vector<T> m;
};
template <typename K, typename... Types> inline void linear_kernel(K kernel, uint32_t, cudaStream_t, uint32_t n_elements, Types... args) {
	for (uint32_t i = 0; i < n_elements; ++i) { blockIdx.x = i; kernel(n_elements, args...); }
	blockIdx.x = 0;
}
} // namespace tcnn
