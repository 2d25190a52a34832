// host_hooks.hip -- TEST HOOKS (include/ngp_hip_host_hooks.h): the leaf functions of csrc/ngp_device.hpp evaluated on the host from the same source the kernels compile,
// so that tests/test_ref_device.py can compare them bit for bit with the reference's own headers compiled for the CPU.  Compiled like the marching kernels
// (-ffp-contract=off).  The product never calls anything in this file.
#include "ngp_device.hpp"
#include "../../include/ngp_hip_host_hooks.h"

using namespace ngp;
static f3 P3(const float* p) { return mk3(p[0], p[1], p[2]); }
static void S3(f3 v, float* o) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

extern "C" {
void ngp_host_cascaded_grid_idx_at(const float* pos, uint32_t n, uint32_t mip, uint32_t* out) { for (uint32_t i = 0; i < n; ++i) out[i] = cascaded_grid_idx_at(P3(pos + 3 * i), mip); }
void ngp_host_mip_from_pos(const float* pos, uint32_t n, uint32_t max_cascade, uint32_t* out) { for (uint32_t i = 0; i < n; ++i) out[i] = mip_from_pos(P3(pos + 3 * i), max_cascade); }
void ngp_host_mip_from_dt(const float* dt, const float* pos, uint32_t n, uint32_t max_cascade, uint32_t* out) { for (uint32_t i = 0; i < n; ++i) out[i] = mip_from_dt(dt[i], P3(pos + 3 * i), max_cascade); }
int ngp_host_density_grid_occupied_at(const float* pos, const uint8_t* bitfield, uint32_t mip) { return occupied_at(P3(pos), bitfield, mip) ? 1 : 0; }
float ngp_host_distance_to_next_voxel(const float* pos, const float* dir, float res) { const f3 d = P3(dir); return distance_to_next_voxel(P3(pos), d, mk3(1.0f) / d, res); }
float ngp_host_advance_to_next_voxel(float t, float cone_angle, const float* pos, const float* dir, uint32_t mip) { const f3 d = P3(dir); return advance_to_next_voxel(t, cone_angle, P3(pos), d, mk3(1.0f) / d, mip); }
float ngp_host_if_unoccupied_advance_to_next_occupied_voxel(float t, float cone_angle, const float* o, const float* d, const uint8_t* bitfield, uint32_t min_mip, uint32_t max_mip, const ngp_aabb* box) {
	const f3 dir = P3(d);
	return skip_to_next_occupied(t, cone_angle, P3(o), dir, mk3(1.0f) / dir, bitfield, min_mip, max_mip, Box(*box));
}
float ngp_host_calc_dt(float t, float cone_angle) { return calc_dt(t, cone_angle); }
float ngp_host_advance_n_steps(float t, float cone_angle, float n) { return advance_n_steps(t, cone_angle, n); }
float ngp_host_to_stepping_space(float t, float cone_angle) { return to_stepping_space(t, cone_angle); }
float ngp_host_from_stepping_space(float n, float cone_angle) { return from_stepping_space(n, cone_angle); }
float ngp_host_warp_dt(float dt) { return warp_dt(dt); }
float ngp_host_unwarp_dt(float dt) { return unwarp_dt(dt); }
void ngp_host_warp_position(const float* pos, const ngp_aabb* box, float* out) { S3(warp_position(P3(pos), Box(*box)), out); }
void ngp_host_unwarp_position(const float* pos, const ngp_aabb* box, float* out) { S3(unwarp_position(P3(pos), Box(*box)), out); }
void ngp_host_warp_direction(const float* dir, float* out) { S3(warp_direction(P3(dir)), out); }
float ngp_host_network_to_rgb(float v, int activation) { return act_rgb(v, activation); }
float ngp_host_network_to_rgb_derivative(float v, int activation) { return act_rgb_d(v, activation); }
float ngp_host_network_to_density(float v, int activation) { return act_density(v, activation); }
float ngp_host_network_to_density_derivative(float v, int activation) { return act_density_d(v, activation); }
void ngp_host_loss_and_gradient(const float* target, const float* prediction, int loss_type, float* loss3, float* gradient3) {
	f3 l, g; loss_and_gradient(P3(target), P3(prediction), loss_type, l, g); S3(l, loss3); S3(g, gradient3);
}
void ngp_host_aabb_ray_intersect(const ngp_aabb* box, const float* origin, const float* dir, float* tminmax2) { const f2 r = Box(*box).ray_intersect(P3(origin), P3(dir)); tminmax2[0] = r.x; tminmax2[1] = r.y; }
int ngp_host_aabb_contains(const ngp_aabb* box, const float* pos) { return Box(*box).contains(P3(pos)) ? 1 : 0; }
float ngp_host_ld_random_val(uint32_t index, uint32_t seed, uint32_t dim) { return ld_random_val(index, seed, dim); }
uint32_t ngp_host_sobol(uint32_t index, uint32_t dim) { return sobol01(index, dim); }
void ngp_host_ld_random_pixel_offset(uint32_t spp, float* out2) { const f2 r = ld_random_pixel_offset(spp); out2[0] = r.x; out2[1] = r.y; }
float ngp_host_srgb_to_linear(float x) { return srgb_to_linear(x); }
float ngp_host_linear_to_srgb(float x) { return linear_to_srgb(x); }
void ngp_host_read_rgba_byte(const float* uv, const int32_t* resolution, const void* pixels, float* out4) {
	const f4 r = read_rgba({uv[0], uv[1]}, resolution, pixels, NGP_IMAGE_BYTE); out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
}
float ngp_host_read_depth(const float* uv, const int32_t* resolution, const float* depth) { return read_depth({uv[0], uv[1]}, resolution, depth); }
uint32_t ngp_host_image_idx_cdf(uint32_t base_idx, uint32_t n_images, const float* cdf, float* pdf) { return image_idx_cdf(base_idx, n_images, cdf, pdf); }
void ngp_host_sample_cdf_2d(const float* sample, uint32_t img, const int32_t* resolution, const float* cdf_x_cond_y, const float* cdf_y, float* uv_out, float* pdf_inout) {
	const f2 r = sample_cdf_2d({sample[0], sample[1]}, img, resolution, cdf_x_cond_y, cdf_y, pdf_inout); uv_out[0] = r.x; uv_out[1] = r.y;
}
}
