// render_kernels.hip -- NeRF renderer for gfx950.
//
// Semantics: the fused per-pixel renderer fused_kernels/render_nerf.cuh:22-184 (== the wavefront NerfTracer,
// testbed_nerf.cu:1591-1860: init_rays :1414, advance_pos :431, generate_next_nerf_network_inputs :523,
// composite_kernel_nerf :579, shade_kernel_nerf :1333) in Shade mode: premultiplied linear RGBA + depth.
//
// MI355X structure: like the training marcher, a pixel's march visits the lattice t_j = from_stepping_space(n' + j)
// and emits exactly the lattice points that lie in an occupied voxel (if_unoccupied_advance_to_next_occupied_voxel,
// nerf_device.cuh:462-495, only skips voxels that are empty at some cascade, and coarser cascades are max-pooled).
// One wavefront tests 64 lattice points of one ray per iteration (ballot -> occupancy mask); the masks are then
// consumed in rounds of RENDER_STEPS samples per live ray: emit -> batched MFMA network inference -> composite ->
// compaction, so early-terminated rays stop costing network evaluations (the reference's 1..8 steps between
// compactions, testbed_nerf.cu:52-53, 1744-1746).
#include "ngp_device.hpp"
#include "ngp_kernels.hpp"

namespace ngp {

static __device__ __forceinline__ float rlattice_t(const RenderRay& r, uint32_t j, float cone_angle) {
	return j == 0 ? r.startt : from_stepping_space(r.nprime + (float)j, cone_angle);
}

// init_rays_with_payload_kernel_nerf + the jittered start of render_nerf.cuh:59-102
__global__ void __launch_bounds__(128) k_render_setup(RenderArgs a, uint32_t pixel_begin, uint32_t n_pixels) {
	const uint32_t li = threadIdx.x + blockIdx.x * blockDim.x;
	if (li >= n_pixels) return;
	const uint32_t idx = pixel_begin + li;
	const uint32_t x = idx % (uint32_t)a.p.resolution[0], y = idx / (uint32_t)a.p.resolution[0];
	const f2 off = ld_random_pixel_offset(a.p.snap_to_pixel_centers ? 0 : a.p.spp_index);
	const f2 uv = {((float)x + off.x) / (float)a.p.resolution[0], ((float)y + off.y) / (float)a.p.resolution[1]};
	const M43 cam = ldm43(a.p.camera);
	f3 ro, rd;
	const bool has_ray = uv_to_ray(uv, a.p.resolution, a.p.focal_length, cam, a.p.screen_center, a.p.lens_mode, a.p.lens_params, a.p.near_distance, ro, rd);
	if (!has_ray) rd = cam.c[2]; // outside the lens' field of view (f-theta): the pixel stays empty (init_rays_with_payload: payload.alive = false)
	rd = normalize3(rd);
	const Box box(a.p.render_aabb);
	float t = fmaxf(box.ray_intersect(ro, rd).x, 0.0f) + 1e-6f;
	const bool alive = has_ray && box.contains(ro + rd * t);
	t = advance_n_steps(t, a.cone_angle, ld_random_val(a.p.spp_index, idx * 786433u));
	RenderRay r;
	r.o[0] = ro.x; r.o[1] = ro.y; r.o[2] = ro.z; r.d[0] = rd.x; r.d[1] = rd.y; r.d[2] = rd.z;
	r.startt = t; r.nprime = to_stepping_space(t, a.cone_angle);
	r.rgba[0] = r.rgba[1] = r.rgba[2] = r.rgba[3] = 0.f;
	r.depth = K_MAX_DEPTH; r.max_weight = 0.f;
	r.cursor = 0; r.n_chunks = 0; r.alive = alive ? 1u : 0u; r.n_emitted = 0;
	a.rays[li] = r;
}

// one wavefront per ray: occupancy masks over the lattice
__global__ void __launch_bounds__(256) k_render_masks(RenderArgs a, uint32_t n_pixels) {
	const uint32_t li = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
	if (li >= n_pixels) return;
	RenderRay r = a.rays[li];
	uint32_t n_chunks = 0, any = 0;
	if (r.alive) {
		const Box box(a.p.render_aabb);
		const f3 ro = ld3(r.o), rd = ld3(r.d);
		for (uint32_t ch = 0; ch < RENDER_MAX_CHUNKS; ++ch) {
			const float t = rlattice_t(r, ch * 64 + lane, a.cone_angle);
			const f3 pos = ro + rd * t;
			const bool inside = t < K_MAX_DEPTH && box.contains(pos);
			bool occ = false;
			if (inside) occ = occupied_at(pos, a.bitfield, min(mip_from_pos(pos), a.max_mip));
			const uint64_t m = __ballot(occ);
			if (lane == 0) a.masks[(size_t)li * RENDER_MAX_CHUNKS + ch] = m;
			any |= m ? 1u : 0u;
			n_chunks = ch + 1;
			if (__ballot(inside) == 0ull) break;
		}
	}
	if (lane == 0) {
		a.rays[li].n_chunks = n_chunks;
		if (!any) a.rays[li].alive = 0;
	}
}

// compaction of live rays into an index list (order is irrelevant for the result)
__global__ void __launch_bounds__(256) k_render_compact(RenderArgs a, uint32_t n_pixels, uint32_t* __restrict__ alive_list, uint32_t* __restrict__ n_alive) {
	const uint32_t li = threadIdx.x + blockIdx.x * blockDim.x;
	const bool alive = li < n_pixels && a.rays[li].alive;
	const uint64_t m = __ballot(alive);
	uint32_t base = 0;
	if ((threadIdx.x & 63u) == 0 && m) { base = atomicAdd(n_alive, (uint32_t)__popcll(m)); atomicAdd(n_alive + 1, (uint32_t)__popcll(m) * RENDER_STEPS); } // [1] = network queries of the next round
	base = __shfl(base, 0, 64);
	if (alive) alive_list[base + (uint32_t)__popcll(m & ((1ull << (threadIdx.x & 63u)) - 1ull))] = li;
}

// generate_next_nerf_network_inputs: the next RENDER_STEPS lattice samples of every live ray
__global__ void __launch_bounds__(128) k_render_emit(RenderArgs a, const uint32_t* __restrict__ alive_list, const uint32_t* __restrict__ n_alive_ptr, float* __restrict__ coords) {
	const uint32_t s = threadIdx.x + blockIdx.x * blockDim.x;
	if (s >= *n_alive_ptr) return;
	const uint32_t li = alive_list[s];
	RenderRay r = a.rays[li];
	const Box train_box(a.train_aabb);
	const f3 ro = ld3(r.o), rd = ld3(r.d);
	const f3 wd = warp_direction(rd);
	uint32_t cursor = r.cursor, emitted = 0;
	const uint32_t end = r.n_chunks * 64;
	const uint32_t cs = 7u + a.n_extra;
	float* c = coords + (size_t)s * RENDER_STEPS * cs;
	while (emitted < RENDER_STEPS && cursor < end) {
		const uint64_t m = a.masks[(size_t)li * RENDER_MAX_CHUNKS + (cursor >> 6)] >> (cursor & 63u);
		if (!m) { cursor = (cursor | 63u) + 1; continue; }
		cursor += (uint32_t)__ffsll((long long)m) - 1;
		const float t = rlattice_t(r, cursor, a.cone_angle);
		const float dt = calc_dt(t, a.cone_angle);
		const f3 wp = warp_position(ro + rd * t, train_box);
		float* cc = c + (size_t)emitted * cs;
		cc[0] = wp.x; cc[1] = wp.y; cc[2] = wp.z; cc[3] = warp_dt(dt); cc[4] = wd.x; cc[5] = wd.y; cc[6] = wd.z;
		for (uint32_t x = 7; x < cs; ++x) cc[x] = a.extra_dims[x - 7];
		++emitted; ++cursor;
	}
	for (uint32_t k = emitted; k < RENDER_STEPS; ++k) { // unused slots: a harmless, in-range query
		float* cc = c + (size_t)k * cs;
		cc[0] = cc[1] = cc[2] = 0.5f; cc[3] = 0.f; cc[4] = cc[5] = cc[6] = 0.5f;
		for (uint32_t x = 7; x < cs; ++x) cc[x] = a.extra_dims[x - 7];
	}
	a.rays[li].cursor = cursor;
	a.rays[li].n_emitted = emitted;
}

// composite_kernel_nerf, testbed_nerf.cu:579-689 / render_nerf.cuh:146-166
__global__ void __launch_bounds__(128) k_render_composite(RenderArgs a, const uint32_t* __restrict__ alive_list, const uint32_t* __restrict__ n_alive_ptr,
		const float* __restrict__ coords, const __half* __restrict__ net_out) {
	const uint32_t s = threadIdx.x + blockIdx.x * blockDim.x;
	if (s >= *n_alive_ptr) return;
	const uint32_t li = alive_list[s];
	RenderRay r = a.rays[li];
	const Box train_box(a.train_aabb);
	const M43 cam = ldm43(a.p.camera);
	const f3 cam_fwd = cam.c[2], cam_pos = cam.c[3];
	bool alive = true;
	for (uint32_t k = 0; k < r.n_emitted; ++k) {
		const float* cc = coords + ((size_t)s * RENDER_STEPS + k) * (7u + a.n_extra);
		const __half* o = net_out + ((size_t)s * RENDER_STEPS + k) * 4;
		const float dt = unwarp_dt(cc[3]);
		const f3 pos = unwarp_position(mk3(cc[0], cc[1], cc[2]), train_box);
		const float alpha = 1.f - __expf(-act_density(__half2float(o[3]), a.density_activation) * dt);
		const float weight = alpha * (1.0f - r.rgba[3]);
		r.rgba[0] += act_rgb(__half2float(o[0]), a.rgb_activation) * weight;
		r.rgba[1] += act_rgb(__half2float(o[1]), a.rgb_activation) * weight;
		r.rgba[2] += act_rgb(__half2float(o[2]), a.rgb_activation) * weight;
		r.rgba[3] += weight;
		if (weight > r.max_weight) { r.max_weight = weight; r.depth = dot3(cam_fwd, pos - cam_pos); }
		if (r.rgba[3] > (1.0f - a.p.min_transmittance)) {
			const float inv = r.rgba[3];
			r.rgba[0] /= inv; r.rgba[1] /= inv; r.rgba[2] /= inv; r.rgba[3] /= inv;
			alive = false;
			break;
		}
	}
	if (r.cursor >= r.n_chunks * 64 || r.n_emitted < RENDER_STEPS) alive = false; // march left the box
	r.alive = alive ? 1u : 0u;
	a.rays[li] = r;
}

// shade_kernel_nerf :1333-1378 / render_nerf.cuh:169-183
__global__ void __launch_bounds__(256) k_render_finish(RenderArgs a, uint32_t pixel_begin, uint32_t n_pixels, float* __restrict__ frame, float* __restrict__ depth) {
	const uint32_t li = threadIdx.x + blockIdx.x * blockDim.x;
	if (li >= n_pixels) return;
	const RenderRay r = a.rays[li];
	float c0 = r.rgba[0], c1 = r.rgba[1], c2 = r.rgba[2];
	if (!a.linear_colors) { c0 = srgb_to_linear(c0); c1 = srgb_to_linear(c1); c2 = srgb_to_linear(c2); }
	const size_t idx = (size_t)pixel_begin + li;
	frame[idx * 4 + 0] = c0; frame[idx * 4 + 1] = c1; frame[idx * 4 + 2] = c2; frame[idx * 4 + 3] = r.rgba[3];
	if (depth) depth[idx] = r.rgba[3] > 0.2f ? r.depth : K_MAX_DEPTH;
}

// accumulate_kernel (render_buffer.cu:228-260): running mean of the per-spp frames; tonemap_kernel (:511-560, Identity curve): exposure,
// background behind the premultiplied colour, optional linear -> sRGB
__global__ void __launch_bounds__(256) k_render_accumulate(uint32_t n_floats, const float* __restrict__ frame, float* __restrict__ accum, float weight) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i < n_floats) { const float a = accum[i]; accum[i] = a + (frame[i] - a) * weight; }
}
__global__ void __launch_bounds__(256) k_render_tonemap(uint32_t n_pixels, float* __restrict__ rgba, float exposure_scale, float bg0, float bg1, float bg2, float bg3, int to_srgb, int curve) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_pixels) return;
	const float4 c = ((float4*)rgba)[i];
	const f4 r = tonemap_pixel({c.x, c.y, c.z, c.w}, exposure_scale, bg0, bg1, bg2, bg3, to_srgb, curve); // ngp_device.hpp (host-testable)
	((float4*)rgba)[i] = make_float4(r.x, r.y, r.z, r.w);
}
void launch_render_accumulate(hipStream_t s, uint32_t n_floats, const float* frame, float* accum, float weight) {
	if (n_floats) hipLaunchKernelGGL(k_render_accumulate, dim3((n_floats + 255) / 256), dim3(256), 0, s, n_floats, frame, accum, weight);
}
void launch_render_tonemap(hipStream_t s, uint32_t n_pixels, float* rgba, float exposure_scale, const float bg[4], int to_srgb, int curve) {
	if (n_pixels) hipLaunchKernelGGL(k_render_tonemap, dim3((n_pixels + 255) / 256), dim3(256), 0, s, n_pixels, rgba, exposure_scale, bg[0], bg[1], bg[2], bg[3], to_srgb, curve);
}

static inline uint32_t nblk(uint32_t n, uint32_t t) { return (n + t - 1) / t; }
void launch_render_setup(hipStream_t s, const RenderArgs& a, uint32_t pixel_begin, uint32_t n) {
	hipLaunchKernelGGL(k_render_setup, dim3(nblk(n, 128)), dim3(128), 0, s, a, pixel_begin, n);
	hipLaunchKernelGGL(k_render_masks, dim3(nblk(n, 4)), dim3(256), 0, s, a, n);
}
void launch_render_compact(hipStream_t s, const RenderArgs& a, uint32_t n, uint32_t* alive_list, uint32_t* n_alive) {
	(void)hipMemsetAsync(n_alive, 0, 8, s);
	hipLaunchKernelGGL(k_render_compact, dim3(nblk(n, 256)), dim3(256), 0, s, a, n, alive_list, n_alive);
}
void launch_render_emit(hipStream_t s, const RenderArgs& a, uint32_t n_alive_host, const uint32_t* alive_list, const uint32_t* n_alive, float* coords) {
	hipLaunchKernelGGL(k_render_emit, dim3(nblk(n_alive_host, 128)), dim3(128), 0, s, a, alive_list, n_alive, coords);
}
void launch_render_composite(hipStream_t s, const RenderArgs& a, uint32_t n_alive_host, const uint32_t* alive_list, const uint32_t* n_alive, const float* coords, const ngp_half* net_out) {
	hipLaunchKernelGGL(k_render_composite, dim3(nblk(n_alive_host, 128)), dim3(128), 0, s, a, alive_list, n_alive, coords, (const __half*)net_out);
}
void launch_render_finish(hipStream_t s, const RenderArgs& a, uint32_t pixel_begin, uint32_t n, float* frame, float* depth) {
	hipLaunchKernelGGL(k_render_finish, dim3(nblk(n, 256)), dim3(256), 0, s, a, pixel_begin, n, frame, depth);
}

} // namespace ngp
