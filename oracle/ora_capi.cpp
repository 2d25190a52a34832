// ORACLE -- TEST INFRASTRUCTURE ONLY (see ora_math.hpp header).  PARITY UNPINNED.
// C entry points of liboracle.so for ctypes (tests/, smoke(), bench.py cpu_baseline). All pointers are HOST pointers.
#include "ora_nerf.hpp"
#include "ora_encmlp.hpp"
#include "ora_sdf.hpp"
#include <omp.h>

using namespace ora;
#define API extern "C" __attribute__((visibility("default")))

static thread_local char g_err[512] = "";
API const char* ora_last_error() { return g_err; }
#define TRY(...) try { __VA_ARGS__; return 0; } catch (const std::exception& e) { snprintf(g_err, sizeof(g_err), "%s", e.what()); return 1; }

API int ora_num_threads() { return omp_get_max_threads(); }
API void ora_set_num_threads(int n) { omp_set_num_threads(n); }

// ---- scalar helpers --------------------------------------------------------------------------
API void ora_f2h(const float* in, uint16_t* out, uint64_t n) { for (uint64_t i = 0; i < n; ++i) out[i] = f2h(in[i]); }
API void ora_h2f(const uint16_t* in, float* out, uint64_t n) { for (uint64_t i = 0; i < n; ++i) out[i] = h2f(in[i]); }
API uint16_t ora_hfma(uint16_t a, uint16_t b, uint16_t c) { return hfma(a, b, c); }
API void ora_pcg32_seed(ngp_pcg32* s, uint64_t initstate, uint64_t initseq) { Pcg32 r(initstate, initseq); *s = r.pod(); }
API uint32_t ora_pcg32_next_uint(ngp_pcg32* s) { Pcg32 r(*s); uint32_t v = r.next_uint(); *s = r.pod(); return v; }
API float ora_pcg32_next_float(ngp_pcg32* s) { Pcg32 r(*s); float v = r.next_float(); *s = r.pod(); return v; }
API void ora_pcg32_advance(ngp_pcg32* s, int64_t delta) { Pcg32 r(*s); r.advance(delta); *s = r.pod(); }
API uint32_t ora_morton3D(uint32_t x, uint32_t y, uint32_t z) { return morton3D(x, y, z); }
API uint32_t ora_morton3D_invert(uint32_t x) { return morton3D_invert(x); }
API void ora_cascaded_grid_idx_at(const float* pos, uint32_t n, uint32_t mip, uint32_t* out) { for (uint32_t i = 0; i < n; ++i) out[i] = cascaded_grid_idx_at(V3(pos + 3 * i), mip); }
API void ora_mip_from_pos(const float* pos, uint32_t n, uint32_t max_cascade, uint32_t* out) { for (uint32_t i = 0; i < n; ++i) out[i] = mip_from_pos(V3(pos + 3 * i), max_cascade); }
API void ora_mip_from_dt(const float* dt, const float* pos, uint32_t n, uint32_t max_cascade, uint32_t* out) { for (uint32_t i = 0; i < n; ++i) out[i] = mip_from_dt(dt[i], V3(pos + 3 * i), max_cascade); }
API float ora_calc_dt(float t, float cone_angle) { return calc_dt(t, cone_angle); }
API float ora_advance_n_steps(float t, float cone_angle, float n) { return advance_n_steps(t, cone_angle, n); }
API float ora_to_stepping_space(float t, float c) { return to_stepping_space(t, c); }
API float ora_from_stepping_space(float n, float c) { return from_stepping_space(n, c); }
API float ora_advance_to_next_voxel(float t, float cone, const float* pos, const float* dir, uint32_t mip) {
	vec3 d = V3(dir); return advance_to_next_voxel(t, cone, V3(pos), d, V3(1.0f) / d, mip);
}
API float ora_warp_dt(float dt) { return warp_dt(dt); }
API float ora_unwarp_dt(float dt) { return unwarp_dt(dt); }
API float ora_ld_random_val(uint32_t index, uint32_t seed, uint32_t dim) { return ld_random_val(index, seed, dim); }
API uint32_t ora_sobol(uint32_t index, uint32_t dim) { return sobol(index, dim); }
API void ora_sh4(const float* dirs01, uint32_t n, uint16_t* out) { for (uint32_t i = 0; i < n; ++i) sh4(dirs01 + 3 * i, out + 16 * i); }
API float ora_srgb_to_linear(float x) { return srgb_to_linear(x); }
API float ora_linear_to_srgb(float x) { return linear_to_srgb(x); }
API void ora_aabb_ray_intersect(const ngp_aabb* a, const float* o, const float* d, float* out2) { vec2 r = Aabb(*a).ray_intersect(V3(o), V3(d)); out2[0] = r.x; out2[1] = r.y; }
API void ora_loss_and_gradient(const float* target, const float* pred, int type, float* loss3, float* grad3) {
	LossAndGradient lg = loss_and_gradient(V3(target), V3(pred), type);
	for (int k = 0; k < 3; ++k) { loss3[k] = lg.loss[k]; grad3[k] = lg.gradient[k]; }
}
API int ora_uv_to_ray(const float* uv, const ngp_image_meta* m, const float* xform12, float* o3, float* d3) {
	vec3 o, d; const bool ok = uv_to_ray({uv[0], uv[1]}, m->resolution, m->focal_length, M43(xform12), m->principal_point, m->lens_mode, m->lens_params, 0.f, o, d);
	for (int k = 0; k < 3; ++k) { o3[k] = o[k]; d3[k] = d[k]; }
	return ok ? 1 : 0;
}
API void ora_pos_to_uv(const float* pos3, const ngp_image_meta* m, const float* xform12, float* uv2) {
	const vec2 uv = pos_to_uv(V3(pos3), m->resolution, m->focal_length, M43(xform12), m->principal_point, m->lens_mode, m->lens_params);
	uv2[0] = uv.x; uv2[1] = uv.y;
}

// twins of the exports of oracle/_ref/libngpdev_ref.so (the reference's own device headers compiled for the CPU): tests/test_ref_device.py compares them bit for bit
API float ora_distance_to_next_voxel(const float* pos, const float* dir, float res) { const vec3 d = V3(dir); return distance_to_next_voxel(V3(pos), d, V3(1.0f) / d, res); }
API int ora_density_grid_occupied_at(const float* pos, const uint8_t* bitfield, uint32_t mip) { return density_grid_occupied_at(V3(pos), bitfield, mip) ? 1 : 0; }
API float ora_if_unoccupied_advance_to_next_occupied_voxel(float t, float cone, const float* o, const float* d, const uint8_t* bitfield, uint32_t min_mip, uint32_t max_mip, const ngp_aabb* box) {
	const vec3 dir = V3(d); return if_unoccupied_advance_to_next_occupied_voxel(t, cone, V3(o), dir, V3(1.0f) / dir, bitfield, min_mip, max_mip, Aabb(*box));
}
API void ora_warp_position(const float* pos, const ngp_aabb* box, float* out) { const vec3 r = warp_position(V3(pos), Aabb(*box)); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
API void ora_unwarp_position(const float* pos, const ngp_aabb* box, float* out) { const vec3 r = unwarp_position(V3(pos), Aabb(*box)); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
API void ora_warp_direction(const float* dir, float* out) { const vec3 r = warp_direction(V3(dir)); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
API float ora_network_to_rgb(float v, int act) { return network_to_rgb(v, act); }
API float ora_network_to_rgb_derivative(float v, int act) { return network_to_rgb_derivative(v, act); }
API float ora_network_to_density(float v, int act) { return network_to_density(v, act); }
API float ora_network_to_density_derivative(float v, int act) { return network_to_density_derivative(v, act); }
API void ora_ld_random_pixel_offset(uint32_t spp, float* out2) { const vec2 r = ld_random_pixel_offset(spp); out2[0] = r.x; out2[1] = r.y; }
API int ora_aabb_contains(const ngp_aabb* a, const float* p) { return Aabb(*a).contains(V3(p)) ? 1 : 0; }
API void ora_read_rgba_byte(const float* uv, const int32_t* res, const void* pixels, float* out4) { const vec4 r = read_rgba({uv[0], uv[1]}, res, pixels, NGP_IMAGE_BYTE); out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w; }
API float ora_read_depth(const float* uv, const int32_t* res, const float* depth) { return read_depth({uv[0], uv[1]}, res, depth); }

// ---- model -----------------------------------------------------------------------------------
API void ora_set_mlp_half_accumulate(int on) { mlp_half_accumulate() = on; }
API int ora_get_mlp_half_accumulate() { return mlp_half_accumulate(); }
API float ora_mlp_dot(uint32_t n, const uint16_t* a, const uint16_t* b) { return mlp_dot(n, [&](uint32_t k) { return h2f(a[k]); }, [&](uint32_t k) { return h2f(b[k]); }); } // test hook: the accumulator switch's arithmetic
API int ora_model_create(const ngp_model_config* cfg, uint64_t seed, void** out) { TRY(*out = new Model(*cfg, seed)) }
API void ora_model_destroy(void* m) { delete (Model*)m; }
API uint64_t ora_model_n_params(void* m) { return ((Model*)m)->n_params; }
API uint64_t ora_model_n_mlp_params(void* m) { return ((Model*)m)->n_mlp; }
API float* ora_model_params_fp(void* m) { return ((Model*)m)->params_fp.data(); }
API uint16_t* ora_model_params(void* m) { return ((Model*)m)->params.data(); }
API uint16_t* ora_model_params_inference(void* m) { return ((Model*)m)->params_inf.data(); }
API uint16_t* ora_model_gradients(void* m) { return ((Model*)m)->grads.data(); }
API float* ora_model_adam_m(void* m) { return ((Model*)m)->adam_m.data(); }
API float* ora_model_adam_v(void* m) { return ((Model*)m)->adam_v.data(); }
API uint32_t* ora_model_adam_steps(void* m) { return ((Model*)m)->adam_steps.data(); }
API float* ora_model_ema(void* m) { return ((Model*)m)->ema_tmp.data(); }
API void ora_model_set_step(void* m, uint32_t step, float lr) { ((Model*)m)->step = step; ((Model*)m)->lr = lr; }
API void ora_model_sync_half(void* m) { ((Model*)m)->sync_half(); }
API uint32_t ora_model_step(void* m) { return ((Model*)m)->step; }
API float ora_model_learning_rate(void* m) { return ((Model*)m)->lr; }
API void ora_model_grid_layout(void* mm, uint32_t* offsets, uint32_t* resolutions, float* scales) {
	Model* m = (Model*)mm;
	for (uint32_t i = 0; i <= m->grid.n_levels; ++i) offsets[i] = m->grid.offsets[i];
	for (uint32_t i = 0; i < m->grid.n_levels; ++i) { resolutions[i] = m->grid.resolutions[i]; scales[i] = m->grid.scales[i]; }
}
API void ora_model_encode(void* m, const float* pos, uint32_t stride, uint32_t n, uint16_t* out) { ((Model*)m)->encode(pos, stride, n, out); }
API void ora_model_inference(void* m, const float* in, uint32_t in_stride, uint32_t n, uint16_t* out, uint32_t out_stride, int use_inf) {
	((Model*)m)->inference(in, in_stride, n, out, out_stride, use_inf != 0);
}
API void ora_model_density(void* m, const float* pos, uint32_t stride, uint32_t n, uint16_t* out, uint32_t out_stride, int use_inf) {
	((Model*)m)->density(pos, stride, n, out, out_stride, use_inf != 0);
}
API void ora_model_training_step_exact_sums(void* m, const float* in, uint32_t in_stride, uint32_t n, const uint16_t* dL_dy, uint32_t dy_stride, int mode) {
	((Model*)m)->training_step(in, in_stride, n, dL_dy, dy_stride, mode);
}
API void ora_model_training_step_extra(void* m, const float* in, uint32_t in_stride, uint32_t n, const uint16_t* dL_dy, uint32_t dy_stride, float* dL_dextra) {
	((Model*)m)->training_step(in, in_stride, n, dL_dy, dy_stride, 0, dL_dextra);
}
API void ora_model_training_step(void* m, const float* in, uint32_t in_stride, uint32_t n, const uint16_t* dL_dy, uint32_t dy_stride) {
	((Model*)m)->training_step(in, in_stride, n, dL_dy, dy_stride);
}
API void ora_model_optimizer_step(void* m, float loss_scale) { ((Model*)m)->optimizer_step(loss_scale); }
API void ora_model_set_trainable(void* m, int net, int enc) { ((Model*)m)->train_network = net; ((Model*)m)->train_encoding = enc; }
// full intermediate state of one sample (for unit tests of the fused kernel)
API void ora_model_eval_debug(void* mm, const float* coord, int use_inf, uint16_t* out4, uint16_t* enc, uint16_t* dact, uint16_t* rgb_in, uint16_t* ract, uint16_t* rgb_out16) {
	((Model*)mm)->eval(coord, use_inf != 0, out4, enc, dact, rgb_in, ract, rgb_out16);
}

// ---- stand-alone kernels (same argument meaning as ngp_k_* in include/ngp_hip.h) ----------------
static ErrorCdf g_hook_cdf; static float* g_hook_error_map = nullptr; static int32_t g_hook_error_map_res[2] = {0, 0}; // error-proportional sampling of the stand-alone K1 / K3
API void ora_set_error_sampling(const float* cdf_x_cond_y, const float* cdf_y, const float* cdf_img, const int32_t* cdf_res, float* error_map, const int32_t* error_map_res) {
	g_hook_cdf = ErrorCdf(); g_hook_cdf.x_cond_y = cdf_x_cond_y; g_hook_cdf.y = cdf_y; g_hook_cdf.img = cdf_img;
	if (cdf_res) { g_hook_cdf.res[0] = cdf_res[0]; g_hook_cdf.res[1] = cdf_res[1]; }
	g_hook_error_map = error_map;
	if (error_map_res) { g_hook_error_map_res[0] = error_map_res[0]; g_hook_error_map_res[1] = error_map_res[1]; }
}
API void ora_k_generate_training_samples(uint32_t n_rays, uint32_t ray_begin, uint32_t ray_end, ngp_aabb aabb, uint32_t max_samples, ngp_pcg32 rng,
		uint32_t* ray_counter, uint32_t* numsteps_counter, uint32_t* ray_indices_out, ngp_ray* rays_out, uint32_t* numsteps_out, float* coords_out,
		uint32_t n_images, const ngp_image_meta* meta, const ngp_xform* xforms, const uint8_t* bitfield, uint32_t max_mip, int snap, float cone_angle_constant) {
	K1Out k = generate_training_samples(n_rays, ray_begin, ray_end, Aabb(aabb), max_samples, Pcg32(rng), ray_indices_out, rays_out, numsteps_out, coords_out,
		n_images, meta, xforms, bitfield, max_mip, snap != 0, cone_angle_constant, &g_hook_cdf);
	*ray_counter = k.ray_counter; *numsteps_counter = k.numsteps_counter;
}
// CPU model of the production (sample-parallel) marcher, see ora_nerf.hpp lattice_march_counts; mode 0 = independent test, 1 = exact-skip walk
API void ora_k1_lattice_counts(int mode, uint32_t n_rays, uint32_t ray_begin, uint32_t ray_end, ngp_aabb aabb, ngp_pcg32 rng, uint32_t n_images, const ngp_image_meta* meta,
		const ngp_xform* xforms, const uint8_t* bitfield, uint32_t max_mip, int snap, float cone_angle_constant, uint32_t* out_counts, uint32_t max_lattice_points) {
	lattice_march_counts(mode, n_rays, ray_begin, ray_end, Aabb(aabb), Pcg32(rng), n_images, meta, xforms, bitfield, max_mip, snap != 0, cone_angle_constant, out_counts, max_lattice_points);
}
// per ray {cause, count_ref, count_lattice, t_ulps, face_distance_cells} (5 x 4 bytes), see ora_nerf.hpp lattice_vs_reference_divergence
API void ora_k1_lattice_divergence(uint32_t n_rays, ngp_aabb aabb, ngp_pcg32 rng, uint32_t n_images, const ngp_image_meta* meta, const ngp_xform* xforms, const uint8_t* bitfield,
		uint32_t max_mip, int snap, float cone_angle_constant, void* out, uint32_t max_lattice_points) {
	static_assert(sizeof(K1Divergence) == 20, "K1Divergence layout");
	lattice_vs_reference_divergence(n_rays, Aabb(aabb), Pcg32(rng), n_images, meta, xforms, bitfield, max_mip, snap != 0, cone_angle_constant, (K1Divergence*)out, max_lattice_points);
}
API void ora_xform_given_rolling_shutter(const ngp_xform* X, const float* rs, const float* uv, float motionblur_time, float* out12) {
	const mat4x3 m = get_xform_given_rolling_shutter(*X, rs, vec2{uv[0], uv[1]}, motionblur_time);
	for (int c = 0; c < 4; ++c) { out12[c * 3 + 0] = m.c[c].x; out12[c * 3 + 1] = m.c[c].y; out12[c * 3 + 2] = m.c[c].z; }
}
static int g_k3_train_mode = 0; // ETrainMode of the stand-alone ora_k_compute_loss
API void ora_set_train_mode(int mode) { g_k3_train_mode = mode; }
static float g_k3_depth_lambda = 0.f; static int g_k3_depth_loss_type = NGP_LOSS_L1; // depth supervision of the stand-alone ora_k_compute_loss
API void ora_set_depth_supervision(float lambda, int loss_type) { g_k3_depth_lambda = lambda; g_k3_depth_loss_type = loss_type; }
API void ora_k_compute_loss(uint32_t n_rays, uint32_t rays_counter, ngp_aabb aabb, ngp_pcg32 rng, uint32_t max_samples_compacted, float loss_scale,
		const float* background_color, int color_space_srgb, int random_bg, int linear_colors, uint32_t n_images, const ngp_image_meta* meta,
		const uint16_t* network_output, uint32_t out_stride, uint32_t* numsteps_counter_compacted, const uint32_t* ray_indices_in, const ngp_ray* rays_in,
		uint32_t* numsteps_inout, const float* coords_in, float* coords_out, uint16_t* dloss, uint32_t dl_stride, int loss_type, float* loss_output,
		int rgb_act, int density_act, int snap, float mean_density, float near_distance) {
	K3Opts o; o.loss_scale = loss_scale; o.background_color = V3(background_color); o.color_space_srgb = color_space_srgb; o.random_bg = random_bg;
	o.linear_colors = linear_colors; o.snap = snap; o.loss_type = loss_type; o.rgb_act = rgb_act; o.density_act = density_act; o.near_distance = near_distance;
	o.train_mode = g_k3_train_mode; o.depth_lambda = g_k3_depth_lambda; o.depth_loss_type = g_k3_depth_loss_type;
	*numsteps_counter_compacted = compute_loss(n_rays, rays_counter, Aabb(aabb), Pcg32(rng), max_samples_compacted, o, n_images, meta, network_output, out_stride,
		ray_indices_in, rays_in, numsteps_inout, coords_in, coords_out, dloss, dl_stride, loss_output, mean_density, &g_hook_cdf, g_hook_error_map, g_hook_error_map_res);
}
API void ora_extra_dims_gradient(uint32_t n_rays_total, uint32_t rays_counter, float* grad_out, uint32_t n_extra, uint32_t n_images, const uint32_t* ray_indices, const uint32_t* numsteps, const float* dextra) {
	extra_dims_gradient(n_rays_total, n_rays_total, rays_counter, grad_out, n_extra, n_images, ray_indices, numsteps, dextra);
}
API void ora_var_adam_step(uint32_t n, float* variable, const float* gradient_scaled, float* m, float* v, uint32_t iter, float lr, float loss_scale) { var_adam_step(n, variable, gradient_scaled, m, v, iter, lr, loss_scale); }
API void ora_k_fill_rollover(uint32_t n_elements, uint32_t n_input, float* coords, uint32_t coord_stride, uint16_t* dloss, uint32_t dl_stride) {
	fill_rollover_and_rescale_h(n_elements, dl_stride, n_input, dloss);
	fill_rollover_f(n_elements, coord_stride, n_input, coords);
}
API void ora_k_mark_untrained_density_grid(uint32_t n_elements, float* grid, uint32_t n_images, const ngp_image_meta* meta, const ngp_xform* xforms, int clear) {
	mark_untrained_density_grid(n_elements, grid, n_images, meta, xforms, clear != 0);
}
API void ora_k_generate_grid_samples(uint32_t n, ngp_pcg32 rng, uint32_t step, ngp_aabb aabb, const float* grid_in, float* pos_out, uint32_t* indices, uint32_t n_cascades, float thresh) {
	generate_grid_samples_nonuniform(n, Pcg32(rng), step, Aabb(aabb), grid_in, pos_out, indices, n_cascades, thresh);
}
API void ora_k_splat_grid_samples(uint32_t n, const uint32_t* indices, const uint16_t* net_out, uint32_t stride, float* grid_out, int density_act) {
	splat_grid_samples(n, indices, net_out, stride, grid_out, density_act);
}
API void ora_k_ema_grid_samples(uint32_t n, float decay, float* grid_out, const float* grid_in) { ema_grid_samples(n, decay, grid_out, grid_in); }
API float ora_k_density_grid_mean(const float* grid) { return density_grid_mean(grid); }
API void ora_k_grid_to_bitfield(const float* grid, uint32_t max_cascade, uint8_t* bitfield, float mean) { grid_to_bitfield_and_pool(grid, max_cascade, bitfield, mean); }

// ---- trainer ---------------------------------------------------------------------------------
API int ora_nerf_create(void* model, const ngp_nerf_options* o, ngp_aabb aabb, void** out) { TRY(*out = new NerfTrainer((Model*)model, *o, aabb)) }
API void ora_nerf_destroy(void* t) { delete (NerfTrainer*)t; }
API void ora_nerf_set_dataset(void* t, uint32_t n, const ngp_image_meta* meta, const ngp_xform* xforms) { ((NerfTrainer*)t)->set_dataset(n, meta, xforms); }
API int ora_nerf_train(void* t, uint32_t n_steps) { TRY(for (uint32_t i = 0; i < n_steps; ++i) ((NerfTrainer*)t)->train_step()) }
API int ora_nerf_train_prep(void* tt) { NerfTrainer* t = (NerfTrainer*)tt;
	TRY(uint32_t skip = (uint32_t)clampi((int)t->training_step / 16, 1, 16); if (t->training_prep_skip_counter % skip == 0) t->training_prep(); ++t->training_prep_skip_counter) }
API int ora_nerf_train_forward_backward(void* t) { TRY(((NerfTrainer*)t)->forward_backward()) }
API int ora_nerf_train_finish(void* t) { TRY(((NerfTrainer*)t)->finish()) }
API int ora_nerf_update_density_grid(void* t, float decay, uint32_t n_uniform, uint32_t n_nonuniform) { TRY(((NerfTrainer*)t)->update_density_grid(decay, n_uniform, n_nonuniform)) }
API void ora_nerf_update_mean_and_bitfield(void* t) { ((NerfTrainer*)t)->update_mean_and_bitfield(); }
API void ora_nerf_get_stats(void* tt, ngp_nerf_stats* s) {
	NerfTrainer* t = (NerfTrainer*)tt;
	s->training_step = t->training_step; s->rays_per_batch = t->rays_per_batch; s->n_rays_last = t->n_rays_last;
	s->measured_batch_size = t->measured_batch_size; s->measured_batch_size_before_compaction = t->measured_batch_size_before_compaction;
	s->loss = t->loss_scalar; s->total_rays = t->total_rays; s->total_samples = t->total_samples;
}
API float* ora_nerf_density_grid(void* t) { return ((NerfTrainer*)t)->density_grid.data(); }
API uint8_t* ora_nerf_bitfield(void* t) { return ((NerfTrainer*)t)->bitfield.data(); }
API float ora_nerf_mean_density(void* t) { return ((NerfTrainer*)t)->mean_density; }
API void ora_nerf_set_mean_density(void* t, float m) { ((NerfTrainer*)t)->mean_density = m; }
// error map / CDFs of the trainer (pointers into its vectors; resolutions {x, y})
API void ora_nerf_error_map(void* tt, float** error_map, int32_t* error_map_res, float** cdf_x_cond_y, float** cdf_y, float** cdf_img, int32_t* cdf_res, int* valid, uint32_t* n_between, uint32_t* n_since) {
	NerfTrainer* t = (NerfTrainer*)tt;
	if (error_map) *error_map = t->error_map.data();
	if (error_map_res) { error_map_res[0] = t->error_map_res[0]; error_map_res[1] = t->error_map_res[1]; }
	if (cdf_x_cond_y) *cdf_x_cond_y = t->cdf_x_cond_y.data();
	if (cdf_y) *cdf_y = t->cdf_y.data();
	if (cdf_img) *cdf_img = t->cdf_img.data();
	if (cdf_res) { cdf_res[0] = t->cdf_res[0]; cdf_res[1] = t->cdf_res[1]; }
	if (valid) *valid = t->is_cdf_valid;
	if (n_between) *n_between = t->n_steps_between_error_map_updates;
	if (n_since) *n_since = t->n_steps_since_error_map_update;
}
API void ora_nerf_set_error_cdfs(void* tt, const float* cdf_x_cond_y, const float* cdf_y, const float* cdf_img, const int32_t* cdf_res) {
	NerfTrainer* t = (NerfTrainer*)tt;
	const size_t n_img = t->meta.size();
	t->cdf_x_cond_y.assign(cdf_x_cond_y, cdf_x_cond_y + (size_t)cdf_res[0] * cdf_res[1] * n_img);
	t->cdf_y.assign(cdf_y, cdf_y + (size_t)cdf_res[1] * n_img);
	t->cdf_img.assign(cdf_img, cdf_img + n_img);
	t->cdf_res[0] = cdf_res[0]; t->cdf_res[1] = cdf_res[1]; t->is_cdf_valid = true;
}
API void ora_nerf_set_error_map_interval(void* tt, uint32_t n) { ((NerfTrainer*)tt)->n_steps_between_error_map_updates = n; }
API void ora_nerf_set_options(void* tt, const ngp_nerf_options* o) { ((NerfTrainer*)tt)->opt = *o; }
API void ora_construct_error_cdfs(uint32_t n_images, uint32_t width, uint32_t height, const float* data, float* cdf_x_cond_y, float* cdf_y, float* cdf_img) {
	construct_error_cdfs(n_images, width, height, data, cdf_x_cond_y, cdf_y, cdf_img);
}
API void ora_sample_cdf_2d(const float* sample, uint32_t img, const int32_t* res, const float* cdf_x_cond_y, const float* cdf_y, float* uv_out, float* pdf_inout) {
	vec2 r = sample_cdf_2d({sample[0], sample[1]}, img, res, cdf_x_cond_y, cdf_y, pdf_inout); uv_out[0] = r.x; uv_out[1] = r.y;
}
API uint32_t ora_image_idx_cdf(uint32_t base_idx, uint32_t n_images, const float* cdf, float* pdf) { return image_idx_cdf(base_idx, n_images, cdf, pdf); }
API void ora_nerf_set_rays_per_batch(void* t, uint32_t r) { ((NerfTrainer*)t)->rays_per_batch = r; }
API void ora_nerf_set_measured(void* t, uint32_t before_compaction, uint32_t compacted) { ((NerfTrainer*)t)->measured_batch_size_before_compaction = before_compaction; ((NerfTrainer*)t)->measured_batch_size = compacted; }
API void ora_nerf_set_rng(void* t, const ngp_pcg32* rng) { ((NerfTrainer*)t)->rng = Pcg32(*rng); }
API void ora_nerf_set_training_step(void* t, uint32_t step) { ((NerfTrainer*)t)->training_step = step; }
API void ora_nerf_get_rng(void* t, ngp_pcg32* rng, ngp_pcg32* grid_rng) { *rng = ((NerfTrainer*)t)->rng.pod(); *grid_rng = ((NerfTrainer*)t)->density_grid_rng.pod(); }
API int ora_nerf_render(void* t, const ngp_render_params* rp, float* frame, float* depth) { TRY(((NerfTrainer*)t)->render(*rp, frame, depth)) }
// scratch access after forward_backward (tests)
API uint32_t ora_nerf_scratch(void* tt, uint32_t** ray_indices, ngp_ray** rays, uint32_t** numsteps, float** coords, uint16_t** mlp_out,
		float** coords_compacted, uint16_t** dloss, uint32_t* counter_before, uint32_t* counter_compacted) {
	NerfTrainer* t = (NerfTrainer*)tt;
	*ray_indices = t->ray_indices.data(); *rays = t->rays.data(); *numsteps = t->numsteps.data(); *coords = t->coords.data(); *mlp_out = t->mlp_out.data();
	*coords_compacted = t->coords_compacted.data(); *dloss = t->dloss.data(); *counter_before = t->counter_before; *counter_compacted = t->counter_compacted;
	return t->n_rays_last;
}

// ---- encoding + MLP (image / SDF primitives' model), forward only --------------------------------
API int ora_encmlp_create(const ngp_encmlp_config* cfg, uint64_t seed, void** out) { TRY(*out = new EncMlp(*cfg, seed)) }
API void ora_encmlp_destroy(void* h) { delete (EncMlp*)h; }
API uint64_t ora_encmlp_n_params(void* h) { return ((EncMlp*)h)->n_params; }
API uint64_t ora_encmlp_n_mlp(void* h) { return ((EncMlp*)h)->n_mlp; }
API float* ora_encmlp_params_fp(void* h) { return ((EncMlp*)h)->params_fp.data(); }
API void ora_encmlp_sync_half(void* h) { ((EncMlp*)h)->sync_half(); }
API void ora_encmlp_grid_layout(void* h, uint32_t* offsets, uint32_t* resolutions, float* scales) {
	const GridLayoutND& g = ((EncMlp*)h)->grid;
	for (uint32_t l = 0; l <= g.n_levels; ++l) offsets[l] = g.offsets[l];
	for (uint32_t l = 0; l < g.n_levels; ++l) { resolutions[l] = g.resolutions[l]; scales[l] = g.scales[l]; }
}
API int ora_encmlp_encode(void* h, const float* in, uint32_t stride, uint32_t n, uint16_t* out) {
	const EncMlp& m = *(EncMlp*)h;
	TRY(for (uint32_t i = 0; i < n; ++i) grid_encode_nd(m.grid, m.params.data() + m.n_mlp, in + (size_t)i * stride, out + (size_t)i * m.net.in))
}
API int ora_encmlp_inference(void* h, const float* in, uint32_t stride, uint32_t n, uint16_t* out, uint32_t out_stride) {
	TRY(((EncMlp*)h)->inference(in, stride, n, out, out_stride))
}
API int ora_encmlp_training_step(void* h, const float* in, uint32_t stride, uint32_t n, const uint16_t* dL_dy, uint32_t dy_stride) {
	TRY(((EncMlp*)h)->training_step(in, stride, n, dL_dy, dy_stride))
}
API int ora_encmlp_optimizer_step(void* h, float loss_scale) { TRY(((EncMlp*)h)->optimizer_step(loss_scale)) }
API void ora_encmlp_set_optimizer(void* h, const ngp_optimizer_config* o) { ((EncMlp*)h)->opt = *o; ((EncMlp*)h)->lr = o->learning_rate; }
API uint16_t* ora_encmlp_params(void* h) { return ((EncMlp*)h)->params.data(); }
API void ora_image_generate_batch(const float* rgba, int w, int h, uint32_t n, ngp_pcg32 rng, int stratified, int snap, int linear_colors, float* positions, float* targets) {
	image_generate_batch(rgba, w, h, n, Pcg32(rng), stratified != 0, snap != 0, linear_colors != 0, positions, targets);
}
API uint16_t* ora_encmlp_gradients(void* h) { return ((EncMlp*)h)->grads.data(); }
API float ora_encmlp_loss_and_gradient(void* h, int mape, const uint16_t* pred, uint32_t pred_stride, const float* target, uint32_t target_stride, uint32_t n, float loss_scale,
		uint16_t* dL_dy) {
	return ((EncMlp*)h)->loss_and_gradient(mape != 0, pred, pred_stride, target, target_stride, n, loss_scale, dL_dy);
}

// ---- SDF data path (brute force, no BVH) ------------------------------------------------------
API void ora_sdf_signed_distance(const float* tris9, uint32_t n_tris, const float* positions, uint32_t n, const float* max_dist, float* out) {
	sdf_signed_distance_brute((const Tri*)tris9, n_tris, positions, n, max_dist, out);
}
// twins of the reference's triangle / sampling primitives for tests/test_ref_sdf.py (the reference side: oracle/ref_bvh_wrapper.cpp)
API float ora_tri_distance_sq(const float* t9, const float* p) { return tri_distance_sq(*(const Tri*)t9, V3(p)); }
API float ora_tri_ray_intersect(const float* t9, const float* ro, const float* rd) { return tri_ray_intersect(*(const Tri*)t9, V3(ro), V3(rd)); }
API void ora_fibonacci_dir32(uint32_t i, const float* offset2, float* out) { const vec3 r = fibonacci_dir32(i, offset2[0], offset2[1]); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
API float ora_logistic_from_uniform(float x, float stddev) { return logistic_from_uniform(x, stddev); }
API uint32_t ora_cdf_search(float val, const float* cdf, uint32_t n) { return cdf_search(val, cdf, n); }
API void ora_sdf_generate_positions(const float* tris9, uint32_t n_tris, const float* cdf, uint32_t n, uint32_t n_exact, uint32_t n_surface, ngp_pcg32 rng, float stddev,
		ngp_aabb box, float* positions, float* distances) {
	sdf_generate_positions((const Tri*)tris9, n_tris, cdf, n, n_exact, n_surface, Pcg32(rng), stddev, box, positions, distances);
}
