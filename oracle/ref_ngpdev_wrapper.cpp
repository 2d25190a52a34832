/* oracle/_ref/libngpdev_ref.so -- TEST INFRASTRUCTURE ONLY: the reference's OWN device headers (include/neural-graphics-primitives/{nerf_device, common_device, random_val,
 * bounding_box}.cuh), compiled for the CPU from the reference's tree where they lie, against oracle/ref_shim (a stand-in for tiny-cuda-nn's vector types and pcg32, which
 * are absent from the mount).  Every export calls ONE reference function and has the signature of the oracle's ora_* twin, so that tests/test_ref_device.py can hold the
 * oracle's restatement against the reference's formulas bit for bit.  Never shipped, never linked by the product; built by oracle/Makefile when /root/reference is present. */
#include <neural-graphics-primitives/nerf_device.cuh>
#include "../include/ngp_hip.h"

using namespace ngp;
#define REF extern "C" __attribute__((visibility("default")))
static vec3 V3(const float* p) { return {p[0], p[1], p[2]}; }
static mat4x3 M43(const float* p) { return mat4x3{V3(p), V3(p + 3), V3(p + 6), V3(p + 9)}; }
static BoundingBox BB(const ngp_aabb* a) { return BoundingBox{V3(a->min), V3(a->max)}; }
static Lens LENS(const ngp_image_meta* m) { Lens l; l.mode = (ELensMode)m->lens_mode; for (int k = 0; k < 7; ++k) l.params[k] = m->lens_params[k]; return l; }

REF void ref_cascaded_grid_idx_at(const float* pos, uint32_t n, uint32_t mip, uint32_t* out) { for (uint32_t i = 0; i < n; ++i) out[i] = cascaded_grid_idx_at(V3(pos + 3 * i), mip); }
REF void ref_mip_from_pos(const float* pos, uint32_t n, uint32_t max_cascade, uint32_t* out) { for (uint32_t i = 0; i < n; ++i) out[i] = mip_from_pos(V3(pos + 3 * i), max_cascade); }
REF void ref_mip_from_dt(const float* dt, const float* pos, uint32_t n, uint32_t max_cascade, uint32_t* out) { for (uint32_t i = 0; i < n; ++i) out[i] = mip_from_dt(dt[i], V3(pos + 3 * i), max_cascade); }
REF float ref_calc_dt(float t, float cone_angle) { return calc_dt(t, cone_angle); }
REF float ref_advance_n_steps(float t, float cone_angle, float n) { return advance_n_steps(t, cone_angle, n); }
REF float ref_to_stepping_space(float t, float c) { return to_stepping_space(t, c); }
REF float ref_from_stepping_space(float n, float c) { return from_stepping_space(n, c); }
REF float ref_advance_to_next_voxel(float t, float cone, const float* pos, const float* dir, uint32_t mip) { const vec3 d = V3(dir); return advance_to_next_voxel(t, cone, V3(pos), d, vec3(1.0f) / d, mip); }
REF float ref_distance_to_next_voxel(const float* pos, const float* dir, float res) { const vec3 d = V3(dir); return distance_to_next_voxel(V3(pos), d, vec3(1.0f) / d, res); }
REF float ref_calc_cone_angle(float cosine, const float* focal_length, float cone_angle_constant) { return calc_cone_angle(cosine, vec2{focal_length[0], focal_length[1]}, cone_angle_constant); }
REF int ref_density_grid_occupied_at(const float* pos, const uint8_t* bitfield, uint32_t mip) { return density_grid_occupied_at(V3(pos), bitfield, mip) ? 1 : 0; }
REF float ref_if_unoccupied_advance_to_next_occupied_voxel(float t, float cone, const float* o, const float* d, const uint8_t* bitfield, uint32_t min_mip, uint32_t max_mip, const ngp_aabb* box) {
	const vec3 dir = V3(d);
	return if_unoccupied_advance_to_next_occupied_voxel(t, cone, Ray{V3(o), dir}, vec3(1.0f) / dir, bitfield, min_mip, max_mip, BB(box), mat3::identity());
}
REF float ref_warp_dt(float dt) { return warp_dt(dt); }
REF float ref_unwarp_dt(float dt) { return unwarp_dt(dt); }
REF void ref_warp_position(const float* pos, const ngp_aabb* box, float* out) { const vec3 r = warp_position(V3(pos), BB(box)); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
REF void ref_unwarp_position(const float* pos, const ngp_aabb* box, float* out) { const vec3 r = unwarp_position(V3(pos), BB(box)); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
REF void ref_warp_direction(const float* dir, float* out) { const vec3 r = warp_direction(V3(dir)); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
REF float ref_network_to_rgb(float v, int act) { return network_to_rgb(v, (ENerfActivation)act); }
REF float ref_network_to_rgb_derivative(float v, int act) { return network_to_rgb_derivative(v, (ENerfActivation)act); }
REF float ref_network_to_density(float v, int act) { return network_to_density(v, (ENerfActivation)act); }
REF float ref_network_to_density_derivative(float v, int act) { return network_to_density_derivative(v, (ENerfActivation)act); }
REF float ref_ld_random_val(uint32_t index, uint32_t seed, uint32_t dim) { return ld_random_val(index, seed, dim); }
REF uint32_t ref_sobol(uint32_t index, uint32_t dim) { return sobol(index, dim); }
REF void ref_ld_random_pixel_offset(uint32_t spp, float* out2) { const vec2 r = ld_random_pixel_offset(spp); out2[0] = r.x; out2[1] = r.y; }
REF float ref_srgb_to_linear(float x) { return srgb_to_linear(x); }
REF float ref_linear_to_srgb(float x) { return linear_to_srgb(x); }
REF void ref_aabb_ray_intersect(const ngp_aabb* a, const float* o, const float* d, float* out2) { const vec2 r = BB(a).ray_intersect(V3(o), V3(d)); out2[0] = r.x; out2[1] = r.y; }
REF int ref_aabb_contains(const ngp_aabb* a, const float* p) { return BB(a).contains(V3(p)) ? 1 : 0; }
REF void ref_loss_and_gradient(const float* target, const float* pred, int type, float* loss3, float* grad3) {
	const LossAndGradient lg = loss_and_gradient(V3(target), V3(pred), (ELossType)type);
	for (int k = 0; k < 3; ++k) { loss3[k] = lg.loss[k]; grad3[k] = lg.gradient[k]; }
}
/* training-ray set-up as generate_training_samples_nerf calls it (testbed_nerf.cu:759-778): spp 0, no parallax / DoF / foveation / distortion map; 1 = valid ray */
REF int ref_uv_to_ray(const float* uv, const ngp_image_meta* m, const float* xform12, float* o3, float* d3) {
	const Ray ray = uv_to_ray(0, vec2{uv[0], uv[1]}, ivec2{m->resolution[0], m->resolution[1]}, vec2{m->focal_length[0], m->focal_length[1]}, M43(xform12), vec2{m->principal_point[0], m->principal_point[1]},
		vec3(0.0f), 0.0f, 1.0f, 0.0f, {}, {}, LENS(m));
	for (int k = 0; k < 3; ++k) { o3[k] = ray.o[k]; d3[k] = ray.d[k]; }
	return ray.is_valid() ? 1 : 0;
}
REF void ref_pos_to_uv(const float* pos3, const ngp_image_meta* m, const float* xform12, float* uv2) {
	const vec2 uv = pos_to_uv(V3(pos3), ivec2{m->resolution[0], m->resolution[1]}, vec2{m->focal_length[0], m->focal_length[1]}, M43(xform12), vec2{m->principal_point[0], m->principal_point[1]}, vec3(0.0f), {}, LENS(m));
	uv2[0] = uv.x; uv2[1] = uv.y;
}
REF uint32_t ref_image_idx(uint32_t base_idx, uint32_t n_rays, uint32_t n_rays_total, uint32_t n_images, const float* cdf, float* pdf) { return image_idx(base_idx, n_rays, n_rays_total, n_images, cdf, pdf); }
REF void ref_sample_cdf_2d(const float* sample, uint32_t img, const int32_t* res, const float* cdf_x_cond_y, const float* cdf_y, float* uv_out, float* pdf_inout) {
	const vec2 r = sample_cdf_2d(vec2{sample[0], sample[1]}, img, ivec2{res[0], res[1]}, cdf_x_cond_y, cdf_y, pdf_inout); uv_out[0] = r.x; uv_out[1] = r.y;
}
/* read_rgba of a byte image (sRGB -> linear, premultiplied; common_device.cuh:846-868) and read_depth (:874-878) */
REF void ref_read_rgba_byte(const float* uv, const int32_t* res, const void* pixels, float* out4) {
	const vec4 r = read_rgba(vec2{uv[0], uv[1]}, ivec2{res[0], res[1]}, pixels, EImageDataType::Byte); for (int k = 0; k < 4; ++k) out4[k] = r[k];
}
REF float ref_read_depth(const float* uv, const int32_t* res, const float* depth) { return read_depth(vec2{uv[0], uv[1]}, ivec2{res[0], res[1]}, depth); }
