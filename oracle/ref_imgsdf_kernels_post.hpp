// oracle/_ref/libngpimgsdf_ref.so, part 3 (see ref_imgsdf_kernels_pre.hpp) -- TEST INFRASTRUCTURE ONLY
} // namespace ngp
using namespace ngp;
#define REF extern "C" __attribute__((visibility("default")))
#define FOR_EACH_THREAD(n) for (uint32_t tid_ = 0; tid_ < (n) && ((blockIdx.x = tid_), true); ++tid_)
static vec3 V3(const float* p) { return {p[0], p[1], p[2]}; }
// train_image's generate_training_data for a float image (testbed_image.cu:243-287) from the uniform randoms on: stratify2_kernel when the batch is a square power of two,
// then eval_image_kernel_and_snap<float, 3>.  positions_inout: n x 2 uniforms in, training positions out; targets: n x 3
REF void ref_image_generate_batch_from_uniforms(const float* rgba, int w, int h, uint32_t n, int stratified, int snap, int linear_colors, float* positions_inout, float* targets) {
	if (stratified) {
		uint32_t log2_batch_size = 0; while ((1u << log2_batch_size) < n) ++log2_batch_size;
		if ((1u << log2_batch_size) == n && log2_batch_size % 2 == 0) FOR_EACH_THREAD(n) stratify2_kernel(n, log2_batch_size, (vec2*)positions_inout);
	}
	FOR_EACH_THREAD(n) eval_image_kernel_and_snap<float, 3>(n, rgba, (vec2*)positions_inout, ivec2{w, h}, targets, snap != 0, linear_colors != 0);
	blockIdx.x = 0;
}
// compute_image_mse's two kernels (:459-488): pixel-centre coordinates of a run of pixels, per-pixel squared error (optionally of the byte-quantised prediction)
REF void ref_image_coords_from_idx(uint32_t n, uint32_t offset, int w, int h, float* pos_out) { FOR_EACH_THREAD(n) image_coords_from_idx(n, offset, (vec2*)pos_out, ivec2{w, h}); blockIdx.x = 0; }
REF void ref_image_mse(uint32_t n, const float* target3, const float* prediction3, float* result, int quantize_to_byte) {
	FOR_EACH_THREAD(n) image_mse_kernel(n, (const vec3*)target3, (const vec3*)prediction3, result, quantize_to_byte != 0);
	blockIdx.x = 0;
}
// generate_training_samples_sdf (testbed_sdf.cu:1449-1544) from the random numbers on, without the octree: positions_inout holds n x 3 uniforms; the first n_surface become
// points on the mesh (sample_uniform_on_triangle_kernel), the rest are scaled into the box; distances: 0 on the surface, |perturbation| * 1.001 for the offset points
// (perturb_sdf_samples; perturbations: (n_surface - n_exact) x 3 logistic variates), |diag| * 1.001 for the uniform ones -- the upper bounds the BVH query then tightens
REF void ref_sdf_generate_positions_from_randoms(const float* tris9, uint32_t n_tris, const float* cdf, uint32_t n, uint32_t n_exact, uint32_t n_surface, const float* perturbations,
		ngp_aabb box, float* positions_inout, float* distances) {
	static_assert(sizeof(Triangle) == 9 * sizeof(float), "Triangle layout");
	FOR_EACH_THREAD(n_surface) sample_uniform_on_triangle_kernel(n_surface, cdf, n_tris, (const Triangle*)tris9, (vec3*)positions_inout);
	for (uint32_t i = 0; i < n_exact; ++i) distances[i] = 0.0f;
	const BoundingBox aabb{V3(box.min), V3(box.max)};
	FOR_EACH_THREAD(n - n_surface) scale_to_aabb_kernel(n - n_surface, aabb, (vec3*)positions_inout + n_surface);
	FOR_EACH_THREAD(n - n_surface) assign_float(n - n_surface, length(aabb.diag()) * 1.001f, distances + n_surface);
	FOR_EACH_THREAD(n_surface - n_exact) perturb_sdf_samples(n_surface - n_exact, (const vec3*)perturbations, (vec3*)positions_inout + n_exact, distances + n_exact);
	blockIdx.x = 0;
}
// compare_signs_kernel without an octree (:540-569): counters[0..7]
REF void ref_sdf_compare_signs(uint32_t n, const float* positions, const float* distances_ref, const float* distances_model, uint32_t* counters8) {
	FOR_EACH_THREAD(n) compare_signs_kernel(n, (const vec3*)positions, distances_ref, distances_model, counters8, nullptr, 0);
	blockIdx.x = 0;
}
// NerfDataset::set_training_image's image conversions (nerf_loader.cu:41-105, 778-827): convert_rgba32 for byte images; from_rgba32<__half> + sharpen<__half> when a sharpening
// amount is set; copy_depth<uint16_t> for integer depth images
REF void ref_convert_rgba32(uint64_t n, const uint8_t* pixels, uint8_t* out, int white_2_transparent, int black_2_transparent, uint32_t mask_color) {
	FOR_EACH_THREAD((uint32_t)n) convert_rgba32(n, pixels, out, white_2_transparent != 0, black_2_transparent != 0, mask_color);
	blockIdx.x = 0;
}
REF void ref_sharpen_rgba8(const uint8_t* pixels, uint32_t w, uint32_t h, float sharpen_amount, int white_2_transparent, int black_2_transparent, uint32_t mask_color, uint16_t* out_half) {
	const uint64_t n = (uint64_t)w * h;
	std::vector<__half> lin(n * 4);
	FOR_EACH_THREAD((uint32_t)n) from_rgba32<__half>(n, pixels, lin.data(), white_2_transparent != 0, black_2_transparent != 0, mask_color);
	const float center_w = 4.f + 1.f / sharpen_amount;
	FOR_EACH_THREAD((uint32_t)n) sharpen<__half>(n, w, lin.data(), (__half*)out_half, center_w, 1.f / (center_w - 4.f));
	blockIdx.x = 0;
}
REF void ref_copy_depth_u16(uint64_t n, float* dst, const uint16_t* src, float depth_scale) {
	FOR_EACH_THREAD((uint32_t)n) copy_depth<uint16_t>(n, dst, src, depth_scale);
	blockIdx.x = 0;
}
