// image_kernels.hip -- gfx950 kernels of the image primitive's trainer: batch generation (random / stratified uv positions +
// targets read from the image) and the full-image MSE.  Reference: src/testbed_image.cu (stratify2_kernel :66-82,
// eval_image_kernel_and_snap :175-229, train_image :231-302, image_coords_from_idx / image_mse_kernel / compute_image_mse :470-547).
// Compiled with -ffp-contract=off like the ray-marching code: positions and (linear_colors) targets are bit-exact against the oracle.
#include "ngp_device.hpp"
#include "ngp_kernels.hpp"
#include <hip/hip_fp16.h>

namespace ngp {

static __device__ __forceinline__ void read_texel(const void* __restrict__ tex, int type, int w, int x, int y, bool linear_colors, float out[4]) {
	const size_t i = ((size_t)y * w + x) * 4;
	if (type == NGP_IMAGE_FLOAT) { const float* p = (const float*)tex + i; out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; out[3] = p[3]; }
	else { const __half* p = (const __half*)tex + i; out[0] = __half2float(p[0]); out[1] = __half2float(p[1]); out[2] = __half2float(p[2]); out[3] = __half2float(p[3]); }
	if (!linear_colors) { out[0] = linear_to_srgb(out[0]); out[1] = linear_to_srgb(out[1]); out[2] = linear_to_srgb(out[2]); }
}
// eval_image_kernel_and_snap<T, 3>: positions are snapped in place, result = rgb
static __device__ __forceinline__ void eval_image_and_snap(const ImageBatchArgs& a, float& px, float& py, bool snap, float rgb[3]) {
	const float rx = (float)a.width, ry = (float)a.height;
	float v[4];
	if (snap) {
		const int ix = (int)floorf(px * rx), iy = (int)floorf(py * ry);
		px = ((float)ix + 0.5f) / rx; py = ((float)iy + 0.5f) / ry;
		read_texel(a.pixels, a.image_data_type, a.width, min(max(ix, 0), a.width - 1), min(max(iy, 0), a.height - 1), a.linear_colors != 0, v);
	} else {
		const float fx = fminf(fmaxf(px * rx - 0.5f, 0.0f), rx - (1.0f + 1e-4f)), fy = fminf(fmaxf(py * ry - 0.5f, 0.0f), ry - (1.0f + 1e-4f));
		const int ix = (int)fx, iy = (int)fy;
		const float wx = fx - (float)ix, wy = fy - (float)iy;
		const int x0 = min(max(ix, 0), a.width - 2), y0 = min(max(iy, 0), a.height - 2);
		float v00[4], v10[4], v01[4], v11[4];
		const bool lin = a.linear_colors != 0;
		read_texel(a.pixels, a.image_data_type, a.width, x0, y0, lin, v00); read_texel(a.pixels, a.image_data_type, a.width, x0 + 1, y0, lin, v10);
		read_texel(a.pixels, a.image_data_type, a.width, x0, y0 + 1, lin, v01); read_texel(a.pixels, a.image_data_type, a.width, x0 + 1, y0 + 1, lin, v11);
		for (int k = 0; k < 4; ++k) v[k] = (1 - wx) * (1 - wy) * v00[k] + (wx) * (1 - wy) * v10[k] + (1 - wx) * (wy) * v01[k] + (wx) * (wy) * v11[k];
	}
	rgb[0] = v[0]; rgb[1] = v[1]; rgb[2] = v[2];
}

// generate_random_uniform [tcnn: element e <- draw e of the pcg32 stream] + stratify2_kernel + eval_image_kernel_and_snap
__global__ void __launch_bounds__(256) k_image_generate_batch(ImageBatchArgs a) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i == 0 && a.zero_word) *a.zero_word = 0.f;
	if (i >= a.n) return;
	Rng rng(a.rng);
	rng.advance((uint64_t)i * 2ull);
	float px = rng.next_float(), py = rng.next_float();
	if (a.stratify_log2) { // batch is a power of four: one sample per cell of a 2^(log2/2) x 2^(log2/2) grid
		const uint32_t log2_size = a.stratify_log2 / 2, size = 1u << log2_size;
		const uint32_t in_batch = i & ((1u << a.stratify_log2) - 1u);
		const uint32_t x = in_batch & (size - 1u), y = in_batch >> log2_size;
		px = px / (float)size + ((float)x / (float)size); py = py / (float)size + ((float)y / (float)size);
	}
	float rgb[3];
	eval_image_and_snap(a, px, py, a.snap_to_pixel_centers != 0, rgb);
	a.positions[(size_t)i * 2 + 0] = px; a.positions[(size_t)i * 2 + 1] = py;
	a.targets[(size_t)i * 3 + 0] = rgb[0]; a.targets[(size_t)i * 3 + 1] = rgb[1]; a.targets[(size_t)i * 3 + 2] = rgb[2];
}

// image_coords_from_idx + eval_image_kernel_and_snap(snap = true): positions / targets of pixels [offset, offset + n)
__global__ void __launch_bounds__(256) k_image_pixel_batch(ImageBatchArgs a, uint32_t offset) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= a.n) return;
	const uint32_t idx = i + offset;
	const int x = min(max((int)(idx % (uint32_t)a.width), 0), a.width - 1), y = min(max((int)(idx / (uint32_t)a.width), 0), a.height - 1);
	float px = ((float)x + 0.5f) / (float)a.width, py = ((float)y + 0.5f) / (float)a.height;
	float rgb[3];
	eval_image_and_snap(a, px, py, true, rgb);
	a.positions[(size_t)i * 2 + 0] = px; a.positions[(size_t)i * 2 + 1] = py;
	a.targets[(size_t)i * 3 + 0] = rgb[0]; a.targets[(size_t)i * 3 + 1] = rgb[1]; a.targets[(size_t)i * 3 + 2] = rgb[2];
}
// image_mse_kernel + reduce_sum: sum over the batch of dot(diff, diff) / 3 into one double accumulator
__global__ void __launch_bounds__(256) k_image_mse(uint32_t n, const float* __restrict__ targets, const __half* __restrict__ pred, uint32_t pred_stride, int quantize_to_byte,
		double* __restrict__ sum) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	float se = 0.f;
	if (i < n) {
		float d2 = 0.f;
		for (int k = 0; k < 3; ++k) {
			float p = __half2float(pred[(size_t)i * pred_stride + k]);
			if (quantize_to_byte) p = (float)min(max((int)(p * 255.0f + 0.5f), 0), 255) / 255.0f;
			const float d = targets[(size_t)i * 3 + k] - p;
			d2 += d * d;
		}
		se = d2 / 3.0f;
	}
	__shared__ float sm[4];
#pragma unroll
	for (int dd = 32; dd >= 1; dd >>= 1) se += __shfl_xor(se, dd, 64);
	if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = se;
	__syncthreads();
	if (threadIdx.x == 0) atomicAdd(sum, (double)((sm[0] + sm[1]) + (sm[2] + sm[3])));
}

void launch_image_generate_batch(hipStream_t s, const ImageBatchArgs& a) {
	if (a.n) hipLaunchKernelGGL(k_image_generate_batch, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
}
void launch_image_pixel_batch(hipStream_t s, const ImageBatchArgs& a, uint32_t offset) {
	if (a.n) hipLaunchKernelGGL(k_image_pixel_batch, dim3((a.n + 255) / 256), dim3(256), 0, s, a, offset);
}
void launch_image_mse(hipStream_t s, uint32_t n, const float* targets, const ngp_half* pred, uint32_t pred_stride, int quantize, double* sum) {
	if (n) hipLaunchKernelGGL(k_image_mse, dim3((n + 255) / 256), dim3(256), 0, s, n, targets, (const __half*)pred, pred_stride, quantize, sum);
}

} // namespace ngp
