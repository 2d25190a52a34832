/* ngp_hip_host_hooks.h -- TEST HOOKS, no GPU needed.  The leaf functions of the device code (instant-ngp_amd/csrc/ngp_device.hpp: stepping space, occupancy indexing and
 * skipping, warps, activations, losses, sampling sequences, colour transfer, texel reads, CDF sampling) are __host__ __device__; these entry points evaluate that very source
 * ON THE HOST so that tests/test_ref_device.py can hold it, bit for bit, against the reference's own headers compiled for the CPU (oracle/_ref/libngpdev_ref.so) -- the
 * product's source against the reference's, without the oracle in between.  Each has the signature of the reference function it mirrors (cited) flattened to plain
 * pointers.  The product never calls them. */
#ifndef NGP_HIP_HOST_HOOKS_H
#define NGP_HIP_HOST_HOOKS_H
#include "ngp_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* nerf_device.cuh:317-368, 431-495 */
void ngp_host_cascaded_grid_idx_at(const float* pos, uint32_t n, uint32_t mip, uint32_t* out);
void ngp_host_mip_from_pos(const float* pos, uint32_t n, uint32_t max_cascade, uint32_t* out);
void ngp_host_mip_from_dt(const float* dt, const float* pos, uint32_t n, uint32_t max_cascade, uint32_t* out);
int ngp_host_density_grid_occupied_at(const float* pos, const uint8_t* bitfield, uint32_t mip);
float ngp_host_distance_to_next_voxel(const float* pos, const float* dir, float res);
float ngp_host_advance_to_next_voxel(float t, float cone_angle, const float* pos, const float* dir, uint32_t mip);
float ngp_host_if_unoccupied_advance_to_next_occupied_voxel(float t, float cone_angle, const float* o, const float* d, const uint8_t* bitfield, uint32_t min_mip, uint32_t max_mip, const ngp_aabb* box);
/* nerf_device.cuh:379-429 */
float ngp_host_calc_dt(float t, float cone_angle);
float ngp_host_advance_n_steps(float t, float cone_angle, float n);
float ngp_host_to_stepping_space(float t, float cone_angle);
float ngp_host_from_stepping_space(float n, float cone_angle);
/* nerf_device.cuh:75-143, 204-310, 601-616 */
float ngp_host_warp_dt(float dt);
float ngp_host_unwarp_dt(float dt);
void ngp_host_warp_position(const float* pos, const ngp_aabb* box, float* out);
void ngp_host_unwarp_position(const float* pos, const ngp_aabb* box, float* out);
void ngp_host_warp_direction(const float* dir, float* out);
float ngp_host_network_to_rgb(float v, int activation);
float ngp_host_network_to_rgb_derivative(float v, int activation);
float ngp_host_network_to_density(float v, int activation);
float ngp_host_network_to_density_derivative(float v, int activation);
void ngp_host_loss_and_gradient(const float* target, const float* prediction, int loss_type, float* loss3, float* gradient3);
/* bounding_box.cuh:163-222 */
void ngp_host_aabb_ray_intersect(const ngp_aabb* box, const float* origin, const float* dir, float* tminmax2);
int ngp_host_aabb_contains(const ngp_aabb* box, const float* pos);
/* random_val.cuh:162-325, common_device.cuh:61-103, 846-878, nerf_device.cuh:497-599 */
float ngp_host_ld_random_val(uint32_t index, uint32_t seed, uint32_t dim);
uint32_t ngp_host_sobol(uint32_t index, uint32_t dim);
void ngp_host_ld_random_pixel_offset(uint32_t spp, float* out2);
float ngp_host_srgb_to_linear(float x);
float ngp_host_linear_to_srgb(float x);
void ngp_host_read_rgba_byte(const float* uv, const int32_t* resolution, const void* pixels, float* out4);
float ngp_host_read_depth(const float* uv, const int32_t* resolution, const float* depth);
uint32_t ngp_host_image_idx_cdf(uint32_t base_idx, uint32_t n_images, const float* cdf, float* pdf);
void ngp_host_sample_cdf_2d(const float* sample, uint32_t img, const int32_t* resolution, const float* cdf_x_cond_y, const float* cdf_y, float* uv_out, float* pdf_inout);
#ifdef __cplusplus
}
#endif
#endif
