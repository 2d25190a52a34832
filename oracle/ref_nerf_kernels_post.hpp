// oracle/_ref/libngpkern_ref.so, part 3 of 3 (see ref_nerf_kernels_pre.hpp) -- TEST INFRASTRUCTURE ONLY.  Still inside `namespace ngp` of the reference's file.
} // namespace ngp
using namespace ngp;
#define REF extern "C" __attribute__((visibility("default")))
static vec3 V3(const float* p) { return {p[0], p[1], p[2]}; }
static BoundingBox BB(const ngp_aabb& a) { return BoundingBox{V3(a.min), V3(a.max)}; }
static pcg32 RNG(ngp_pcg32 r) { pcg32 g; g.state = r.state; g.inc = r.inc; return g; }
static std::vector<TrainingImageMetadata> META(uint32_t n, const ngp_image_meta* m) {
	std::vector<TrainingImageMetadata> out(n);
	for (uint32_t i = 0; i < n; ++i) {
		out[i].pixels = m[i].pixels; out[i].image_data_type = (EImageDataType)m[i].image_data_type; out[i].depth = m[i].depth;
		out[i].lens.mode = (ELensMode)m[i].lens_mode; for (int k = 0; k < 7; ++k) out[i].lens.params[k] = m[i].lens_params[k];
		out[i].resolution = ivec2{m[i].resolution[0], m[i].resolution[1]}; out[i].principal_point = vec2{m[i].principal_point[0], m[i].principal_point[1]};
		out[i].focal_length = vec2{m[i].focal_length[0], m[i].focal_length[1]};
		out[i].rolling_shutter = vec4{m[i].rolling_shutter[0], m[i].rolling_shutter[1], m[i].rolling_shutter[2], m[i].rolling_shutter[3]};
	}
	return out;
}
static std::vector<TrainingXForm> XFORMS(uint32_t n, const ngp_xform* x) {
	std::vector<TrainingXForm> out(n);
	for (uint32_t i = 0; i < n; ++i) { out[i].start = mat4x3{V3(x[i].start), V3(x[i].start + 3), V3(x[i].start + 6), V3(x[i].start + 9)}; out[i].end = mat4x3{V3(x[i].end), V3(x[i].end + 3), V3(x[i].end + 6), V3(x[i].end + 9)}; }
	return out;
}
#define FOR_EACH_THREAD(n) for (uint32_t tid_ = 0; tid_ < (n) && ((blockIdx.x = tid_), true); ++tid_)

// error-proportional sampling state, like the oracle's hook (ora_set_error_sampling)
static const float* g_cdf_x_cond_y = nullptr; static const float* g_cdf_y = nullptr; static const float* g_cdf_img = nullptr; static ivec2 g_cdf_res{0, 0};
static float* g_error_map = nullptr; static ivec2 g_error_map_res{0, 0};
REF void ref_set_error_sampling(const float* cdf_x_cond_y, const float* cdf_y, const float* cdf_img, const int32_t* cdf_res, float* error_map, const int32_t* error_map_res) {
	g_cdf_x_cond_y = cdf_x_cond_y; g_cdf_y = cdf_y; g_cdf_img = cdf_img; g_cdf_res = cdf_res ? ivec2{cdf_res[0], cdf_res[1]} : ivec2{0, 0};
	g_error_map = error_map; g_error_map_res = error_map_res ? ivec2{error_map_res[0], error_map_res[1]} : ivec2{0, 0};
}

// generate_training_samples_nerf (testbed_nerf.cu:691-850) launched as train_nerf_step launches it (:3064-3096): no envmap, no distortion map, no extra dims,
// max_level_rand_training off.  Rays [ray_begin, ray_end) of the n_rays of the batch (the data-parallel shard), in order.
REF void ref_k_generate_training_samples(uint32_t n_rays, uint32_t ray_begin, uint32_t ray_end, ngp_aabb aabb, uint32_t max_samples, ngp_pcg32 rng,
		uint32_t* ray_counter, uint32_t* numsteps_counter, uint32_t* ray_indices_out, ngp_ray* rays_out, uint32_t* numsteps_out, float* coords_out,
		uint32_t n_images, const ngp_image_meta* meta, const ngp_xform* xforms, const uint8_t* bitfield, uint32_t max_mip, int snap, float cone_angle_constant) {
	static_assert(sizeof(Ray) == sizeof(ngp_ray) && sizeof(NerfCoordinate) == 7 * sizeof(float), "layouts");
	const auto m = META(n_images, meta); const auto x = XFORMS(n_images, xforms);
	*ray_counter = 0; *numsteps_counter = 0;
	for (uint32_t i = ray_begin; i < ray_end; ++i) {
		blockIdx.x = i;
		generate_training_samples_nerf(n_rays, BB(aabb), max_samples, 0u, RNG(rng), ray_counter, numsteps_counter, ray_indices_out, (Ray*)rays_out, numsteps_out,
			PitchedPtr<NerfCoordinate>((NerfCoordinate*)coords_out, 1, 0, 0), n_images, m.data(), x.data(), bitfield, max_mip, false, nullptr, snap != 0, false, cone_angle_constant,
			Buffer2DView<const vec2>{}, g_cdf_x_cond_y, g_cdf_y, g_cdf_img, g_cdf_res, nullptr, 0);
	}
	blockIdx.x = 0;
}
// the same launch with extra dims (:3089-3096: extra_dims_gpu.data(), n_extra_dims; coords pitched by (7 + n_extra_dims) floats, :3044-3046)
REF void ref_k_generate_training_samples_extra(uint32_t n_rays, uint32_t ray_begin, uint32_t ray_end, ngp_aabb aabb, uint32_t max_samples, ngp_pcg32 rng,
		uint32_t* ray_counter, uint32_t* numsteps_counter, uint32_t* ray_indices_out, ngp_ray* rays_out, uint32_t* numsteps_out, float* coords_out /* (7 + n_extra) floats per sample */,
		uint32_t n_images, const ngp_image_meta* meta, const ngp_xform* xforms, const uint8_t* bitfield, uint32_t max_mip, int snap, float cone_angle_constant,
		const float* extra_dims, uint32_t n_extra) {
	const auto m = META(n_images, meta); const auto x = XFORMS(n_images, xforms);
	*ray_counter = 0; *numsteps_counter = 0;
	for (uint32_t i = ray_begin; i < ray_end; ++i) {
		blockIdx.x = i;
		generate_training_samples_nerf(n_rays, BB(aabb), max_samples, 0u, RNG(rng), ray_counter, numsteps_counter, ray_indices_out, (Ray*)rays_out, numsteps_out,
			PitchedPtr<NerfCoordinate>((NerfCoordinate*)coords_out, 1, 0, n_extra * sizeof(float)), n_images, m.data(), x.data(), bitfield, max_mip, false, nullptr, snap != 0, false, cone_angle_constant,
			Buffer2DView<const vec2>{}, g_cdf_x_cond_y, g_cdf_y, g_cdf_img, g_cdf_res, extra_dims, n_extra);
	}
	blockIdx.x = 0;
}
// compute_extra_dims_gradient_train_nerf (:1293-1330) as train_nerf_step launches it (:3325-3340): coords_gradient = the network's input gradient, pitched like the coordinates
REF void ref_k_extra_dims_gradient(uint32_t n_rays, uint32_t n_rays_total, uint32_t rays_counter, float* extra_dims_gradient, uint32_t n_extra, uint32_t n_images,
		const uint32_t* ray_indices_in, uint32_t* numsteps_in, float* coords_gradient /* (7 + n_extra) floats per row */) {
	FOR_EACH_THREAD(rays_counter) compute_extra_dims_gradient_train_nerf(n_rays, n_rays_total, &rays_counter, extra_dims_gradient, n_extra, n_images, ray_indices_in, numsteps_in,
		PitchedPtr<NerfCoordinate>((NerfCoordinate*)coords_gradient, 1, 0, n_extra * sizeof(float)), g_cdf_img);
	blockIdx.x = 0;
}
// compute_loss_kernel_train_nerf (:852-1181) as train_nerf_step launches it (:3171-3228): no envmap, no sharpness, no exposure training, padded_output_width = out_stride
static float g_depth_lambda = 0.f; static int g_depth_loss_type = NGP_LOSS_L1;
REF void ref_set_depth_supervision(float lambda, int loss_type) { g_depth_lambda = lambda; g_depth_loss_type = loss_type; }
REF void ref_k_compute_loss(uint32_t n_rays, uint32_t rays_counter, ngp_aabb aabb, ngp_pcg32 rng, uint32_t max_samples_compacted, float loss_scale,
		const float* background_color, int color_space_srgb, int random_bg, int linear_colors, uint32_t n_images, const ngp_image_meta* meta,
		const uint16_t* network_output, uint32_t out_stride, uint32_t* numsteps_counter_compacted, const uint32_t* ray_indices_in, const ngp_ray* rays_in,
		uint32_t* numsteps_inout, const float* coords_in, float* coords_out, uint16_t* dloss, uint32_t dl_stride, int loss_type, float* loss_output,
		int rgb_act, int density_act, int snap, float mean_density, float near_distance) {
	const auto m = META(n_images, meta);
	if (dl_stride != out_stride) { std::fprintf(stderr, "ref_k_compute_loss: the reference has one padded_output_width for the network output and its gradient\n"); std::abort(); }
	*numsteps_counter_compacted = 0;
	std::vector<vec3> exposure(n_images, vec3(0.0f)), exposure_gradient(n_images, vec3(0.0f));
	FOR_EACH_THREAD(rays_counter) {
		compute_loss_kernel_train_nerf(n_rays, BB(aabb), 0u, RNG(rng), max_samples_compacted, &rays_counter, loss_scale, (int)out_stride,
			Buffer2DView<const vec4>{}, (float*)nullptr, ivec2{0, 0}, ELossType::L2, V3(background_color), color_space_srgb ? EColorSpace::SRGB : EColorSpace::Linear, random_bg != 0, linear_colors != 0,
			n_images, m.data(), (const network_precision_t*)network_output, numsteps_counter_compacted, ray_indices_in, (const Ray*)rays_in, numsteps_inout,
			PitchedPtr<const NerfCoordinate>((const NerfCoordinate*)coords_in, 1, 0, 0), PitchedPtr<NerfCoordinate>((NerfCoordinate*)coords_out, 1, 0, 0), (network_precision_t*)dloss,
			(ELossType)loss_type, (ELossType)g_depth_loss_type, loss_output, false, nullptr, (ENerfActivation)rgb_act, (ENerfActivation)density_act, snap != 0,
			g_error_map, g_cdf_x_cond_y, g_cdf_y, g_cdf_img, g_error_map_res, g_cdf_res, nullptr, ivec2{0, 0}, nullptr, nullptr, &mean_density, 0u,
			exposure.data(), exposure_gradient.data(), g_depth_lambda, near_distance);
	}
	blockIdx.x = 0;
}
// occupancy grid (:87-162, 216-257, 259-284, 316-338, 348-429)
REF void ref_k_mark_untrained_density_grid(uint32_t n_elements, float* grid, uint32_t n_images, const ngp_image_meta* meta, const ngp_xform* xforms, int clear) {
	const auto m = META(n_images, meta); const auto x = XFORMS(n_images, xforms);
	FOR_EACH_THREAD(n_elements) mark_untrained_density_grid(n_elements, grid, n_images, m.data(), x.data(), clear != 0);
	blockIdx.x = 0;
}
REF void ref_k_generate_grid_samples(uint32_t n, ngp_pcg32 rng, uint32_t step, ngp_aabb aabb, const float* grid_in, float* pos_out, uint32_t* indices, uint32_t n_cascades, float thresh) {
	static_assert(sizeof(NerfPosition) == 3 * sizeof(float), "NerfPosition layout");
	FOR_EACH_THREAD(n) generate_grid_samples_nerf_nonuniform(n, RNG(rng), step, BB(aabb), grid_in, (NerfPosition*)pos_out, indices, n_cascades, thresh);
	blockIdx.x = 0;
}
REF void ref_k_splat_grid_samples(uint32_t n, const uint32_t* indices, const uint16_t* net_out, uint32_t stride, float* grid_out, int density_act) {
	if (stride != 1) { std::fprintf(stderr, "ref_k_splat_grid_samples: the reference reads a packed density column (stride 1)\n"); std::abort(); }
	FOR_EACH_THREAD(n) splat_grid_samples_nerf_max_nearest_neighbor(n, indices, (const network_precision_t*)net_out, grid_out, ENerfActivation::None, (ENerfActivation)density_act);
	blockIdx.x = 0;
}
REF void ref_k_ema_grid_samples(uint32_t n, float decay, float* grid_out, const float* grid_in) {
	FOR_EACH_THREAD(n) ema_grid_samples_nerf(n, decay, 0u, grid_out, grid_in);
	blockIdx.x = 0;
}
// update_density_grid_mean_and_bitfield's two kernels with its launch parameters (:2594-2633): grid_to_bitfield over all NERF_CASCADES() levels (zero beyond max_cascade),
// then bitfield_max_pool level by level.  bitfield: grid_mip_offset(NERF_CASCADES()) / 8 bytes.
REF void ref_k_grid_to_bitfield(const float* grid, uint32_t max_cascade, uint8_t* bitfield, float mean) {
	const uint32_t n_elements = NERF_GRID_N_CELLS();
	FOR_EACH_THREAD(n_elements / 8 * NERF_CASCADES()) grid_to_bitfield(n_elements / 8 * NERF_CASCADES(), n_elements / 8 * (max_cascade + 1), grid, bitfield, &mean);
	for (uint32_t level = 1; level < NERF_CASCADES(); ++level) {
		const uint8_t* prev = bitfield + grid_mip_offset(level - 1) / 8; uint8_t* next = bitfield + grid_mip_offset(level) / 8;
		FOR_EACH_THREAD(n_elements / 64) bitfield_max_pool(n_elements / 64, prev, next);
	}
	blockIdx.x = 0;
}
// error-map CDFs (:1530-1580), launched like :2807-2829: construct_cdf_2d on a 2-D grid (x = row, y = image), construct_cdf_1d per image.  cdf_img_raw is what the kernels
// leave in cdf_img: the per-image sums, which the reference then turns into the image CDF on the host (:2831-2846, a Testbed member: not compiled here)
REF void ref_k_construct_error_cdfs(uint32_t n_images, uint32_t width, uint32_t height, const float* data, float* cdf_x_cond_y, float* cdf_y, float* cdf_img_raw) {
	for (uint32_t img = 0; img < n_images; ++img) for (uint32_t y = 0; y < height; ++y) { blockIdx.x = y; blockIdx.y = img; construct_cdf_2d(n_images, height, width, data, cdf_x_cond_y, cdf_y); }
	blockIdx.y = 0;
	FOR_EACH_THREAD(n_images) construct_cdf_1d(n_images, height, cdf_y, cdf_img_raw);
	blockIdx.x = 0;
}
