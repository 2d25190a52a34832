// nerf_kernels.hip -- gfx950 kernels for the ngp-side of the NeRF training step:
//   K1 ray generation + occupancy-skipping march (reference testbed_nerf.cu:691-849)
//   K3 compositing, loss, adjoint and sample compaction (testbed_nerf.cu:852-1180)
//   K4 roll-over padding (launches testbed_nerf.cu:3298-3306)
//   occupancy-grid maintenance (testbed_nerf.cu:87-396, 2476-2633)
//   the on-device batch-size controller replacing NerfCounters::update_after_training's host
//   round-trip (testbed_nerf.cu:2678-2702).
//
// MI355X notes: one thread per ray like the reference, but span reservation is done with ONE global
// atomic per 64-lane wavefront (wave-level exclusive scan of the per-ray sample counts), which also
// makes a wavefront's samples contiguous in HBM -> the fused encoding kernel that consumes them reads
// ray-coherent (cache-line sharing) positions.  Compiled with -ffp-contract=off: sample positions and
// occupancy indices must equal the un-contracted IEEE arithmetic of the oracle bit for bit.
#include "ngp_device.hpp"
#include "ngp_kernels.hpp"

namespace ngp {
// mip of a marched point (nerf_device.cuh:450-460).  Default: the reference's arithmetic with tcnn's lower-bound-first clamp (a pooled level above max_mip for long steps);
// ablation DBG_K1_MIP_CLAMP_MIN_MAX: min(max()) = never above max_mip.
static __device__ __forceinline__ uint32_t k1_mip(const K1Args& a, float dt, f3 pos) { const uint32_t m = mip_from_dt(dt, pos, a.max_mip); return a.clamp_min_max ? min(m, a.max_mip) : m; }


uint32_t g_debug_flags = 0;
uint32_t g_debug_flags2 = 0;

static __device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t& total) {
	const uint32_t lane = threadIdx.x & 63u;
	uint32_t x = v;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		uint32_t y = __shfl_up(x, d, 64);
		if (lane >= (uint32_t)d) x += y;
	}
	total = __shfl(x, 63, 64);
	return x - v;
}
static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
	return v;
}

// ------------------------------------------------------------------------------------------------
// K1
// ------------------------------------------------------------------------------------------------
// The training pixel of global ray i (testbed_nerf.cu:726-730 in K1 = :955-961 in K3): image, uv and the density the pair was drawn with
// (1 without CDFs); leaves rng behind the two uv draws.
static __device__ __forceinline__ f2 training_pixel(const ErrorCdf& cdf, Rng& rng, uint32_t i, uint32_t n_rays, uint32_t n_images, const ngp_image_meta* __restrict__ metadata,
		int snap, uint32_t& img, float& pdf) {
	float img_pdf = 1.0f, uv_pdf = 1.0f;
	img = cdf.img ? image_idx_cdf(i, n_images, cdf.img, &img_pdf) : image_idx(i, n_rays, n_images);
	const f2 uv = cdf.x_cond_y ? random_image_pos_training_cdf(rng, metadata[img].resolution, snap != 0, cdf.x_cond_y, cdf.y, cdf.res, img, &uv_pdf)
	                           : random_image_pos_training(rng, metadata[img].resolution, snap != 0);
	pdf = img_pdf * uv_pdf;
	return uv;
}
// testbed_nerf.cu:1042-1071 (no sharpness data): the ray's mean loss goes into the four cells around its pixel
static __device__ __forceinline__ void deposit_error(float* __restrict__ error_map, const int32_t emres[2], const int32_t resolution[2], uint32_t img, f2 uv, float mean_loss) {
	const float px = fminf(fmaxf(uv.x * (float)emres[0] - 0.5f, 0.0f), (float)emres[0] - (1.0f + 1e-4f));
	const float py = fminf(fmaxf(uv.y * (float)emres[1] - 0.5f, 0.0f), (float)emres[1] - (1.0f + 1e-4f));
	const int ix = (int)px, iy = (int)py;
	const float wx = px - (float)ix, wy = py - (float)iy;
	const int x = clampi(ix, 0, resolution[0] - 2), y = clampi(iy, 0, resolution[1] - 2); // (the reference clamps against the IMAGE resolution, :1047)
	float* e = error_map + (size_t)img * emres[0] * emres[1];
	atomicAdd(&e[y * emres[0] + x], (1 - wx) * (1 - wy) * mean_loss);
	atomicAdd(&e[y * emres[0] + x + 1], wx * (1 - wy) * mean_loss);
	atomicAdd(&e[(y + 1) * emres[0] + x], (1 - wx) * wy * mean_loss);
	atomicAdd(&e[(y + 1) * emres[0] + x + 1], wx * wy * mean_loss);
}
__global__ void __launch_bounds__(128) k_generate_training_samples(K1Args a) {
	const uint32_t n_rays = a.n_rays_ptr ? *a.n_rays_ptr : a.n_rays;
	const uint32_t max_samples = a.max_samples_ptr ? min(*a.max_samples_ptr, a.max_samples) : a.max_samples;
	// this rank's slice of the global ray range; world_size == 1 -> [0, n_rays)
	const uint32_t ray_begin = (uint32_t)(((uint64_t)n_rays * a.rank) / a.world_size);
	const uint32_t ray_end = (uint32_t)(((uint64_t)n_rays * (a.rank + 1)) / a.world_size);
	const uint32_t i = ray_begin + threadIdx.x + blockIdx.x * blockDim.x;
	if (blockIdx.x * blockDim.x >= ray_end - ray_begin) return; // whole block out of range (uniform)
	const Box aabb(a.aabb);
	const uint32_t cs = 7u + a.n_extra; // floats per NerfCoordinate (PitchedPtr stride, testbed_nerf.cu:3010-3011)
	const float* extra = nullptr;          // the ray's extra dims = its image's (testbed_nerf.cu:744)

	bool valid = i < ray_end;
	uint32_t numsteps = 0;
	f3 ro = mk3(0.f), rd = mk3(0.f), rdn = mk3(0.f, 0.f, 1.f), idir = mk3(1.f);
	float startt = 0.f, cone_angle = a.cone_angle_constant;
	if (valid) {
		Rng rng(a.rng);
		rng.advance((uint64_t)(i * N_RANDOM_PER_RAY));
		uint32_t img; float pix_pdf;
		f2 uv = training_pixel(a.cdf, rng, i, n_rays, a.n_images, a.metadata, a.snap_to_pixel_centers, img, pix_pdf);
		const ngp_image_meta& m = a.metadata[img];
		if (a.n_extra) extra = a.extra_dims + (size_t)img * a.n_extra;
		if (read_rgba(uv, m.resolution, m.pixels, m.image_data_type).x < 0.0f) valid = false;
		if (valid) {
			const float motionblur_time = rng.next_float();
			const M43 xform = xform_given_rolling_shutter(a.xforms[img], m.rolling_shutter, uv, motionblur_time); // common_device.cuh:670-674
			if (!uv_to_ray(uv, m.resolution, m.focal_length, xform, m.principal_point, m.lens_mode, m.lens_params, 0.0f, ro, rd)) { ro = xform.c[3]; rd = xform.c[2]; } // testbed_nerf.cu:776-778
			rdn = normalize3(rd);
			f2 tminmax = aabb.ray_intersect(ro, rdn);
			tminmax.x = fmaxf(tminmax.x, 0.0f);
			startt = advance_n_steps(tminmax.x, cone_angle, rng.next_float());
			idir = mk3(1.0f) / rdn;
			uint32_t j = 0;
			float t = startt;
			f3 pos;
			while (aabb.contains(pos = ro + t * rdn) && j < N_STEPS) {
				float dt = calc_dt(t, cone_angle);
				uint32_t mip = k1_mip(a, dt, pos);
				if (occupied_at(pos, a.bitfield, mip)) { ++j; t += dt; }
				else t = advance_to_next_voxel(t, cone_angle, pos, rdn, idir, mip);
			}
			numsteps = j;
			if (j == 0) valid = false;
		}
	}
	// one atomic per wavefront reserves the contiguous spans of its 64 rays
	uint32_t wave_total;
	uint32_t offset = wave_excl_scan(valid ? numsteps : 0u, wave_total);
	uint32_t wave_base = 0;
	if ((threadIdx.x & 63u) == 0 && wave_total) wave_base = atomicAdd(a.numsteps_counter, wave_total);
	wave_base = __shfl(wave_base, 0, 64);
	const uint32_t base = wave_base + offset;
	if (valid && base + numsteps > max_samples) valid = false;
	uint32_t n_valid_total;
	uint32_t ray_off = wave_excl_scan(valid ? 1u : 0u, n_valid_total);
	uint32_t ray_base = 0;
	if ((threadIdx.x & 63u) == 0 && n_valid_total) ray_base = atomicAdd(a.ray_counter, n_valid_total);
	ray_base = __shfl(ray_base, 0, 64);
	if (!valid) return;

	const uint32_t ray_idx = ray_base + ray_off;
	a.ray_indices_out[ray_idx] = i;
	ngp_ray r; r.o[0] = ro.x; r.o[1] = ro.y; r.o[2] = ro.z; r.d[0] = rd.x; r.d[1] = rd.y; r.d[2] = rd.z;
	a.rays_out[ray_idx] = r;
	a.numsteps_out[ray_idx * 2 + 0] = numsteps;
	a.numsteps_out[ray_idx * 2 + 1] = base;

	float* co = a.coords_out + (size_t)base * cs;
	const f3 wd = warp_direction(rdn);
	float t = startt;
	uint32_t j = 0;
	f3 pos;
	while (aabb.contains(pos = ro + t * rdn) && j < numsteps) {
		float dt = calc_dt(t, cone_angle);
		uint32_t mip = k1_mip(a, dt, pos);
		if (occupied_at(pos, a.bitfield, mip)) {
			f3 wp = warp_position(pos, aabb);
			float* c = co + (size_t)j * cs;
			c[0] = wp.x; c[1] = wp.y; c[2] = wp.z; c[3] = warp_dt(dt); c[4] = wd.x; c[5] = wd.y; c[6] = wd.z;
			for (uint32_t k = 7; k < cs; ++k) c[k] = extra[k - 7]; // set_with_optional_extra_dims, testbed_nerf.cu:833
			++j; t += dt;
		} else t = advance_to_next_voxel(t, cone_angle, pos, rdn, idir, mip);
	}
}


// ------------------------------------------------------------------------------------------------
// K1, sample-parallel ("lattice") formulation -- the production ray marcher on MI355X.
//
// The reference's per-ray loop (testbed_nerf.cu:798-807) is a sequential float recurrence, but its
// RESULT has a closed form: every visited t is  t_j = from_stepping_space(n' + j),  n' =
// to_stepping_space(startt), j = 0,1,2,...  (each update is t <- from(to(t) + k), k integer: k = 1
// for a step through an occupied voxel, k = ceil(...) for a skip to the first lattice point past an
// empty voxel), and a lattice point is emitted as a sample iff it lies in an occupied voxel: points
// inside an empty voxel are the ones the skip jumps over, and the landing point of a skip is the first
// lattice point of the next voxel.  So the samples of a ray are { t_j : occupied(pos(t_j)) } in order
// of j, cut at NERF_STEPS samples -- all j can be tested independently.
// One wavefront marches one ray, 64 lattice points per iteration (ballot -> 64-bit occupancy mask);
// sample spans come from a prefix sum over the per-ray counts, so the output order is deterministic
// (ray-index order) and needs no atomics.  58k rays x ~16 chunks keep all 256 CUs busy, where the
// thread-per-ray loop ran < 1 wavefront per SIMD with ~10^3-instruction dependent chains.
// Numerically the closed form differs from the sequential recurrence only through fp32 double
// rounding of (n + k1) + k2 vs n + (k1 + k2) and of fl(fl(x*M)/M): the great majority of rays are
// bit-identical, the rest carry <= 2-ulp offsets in t (tests/test_gpu_nerf.py quantifies both).
// ------------------------------------------------------------------------------------------------
// the seven floats of a NerfCoordinate (a 28-byte record: 4-byte aligned only) as one 16-byte and one 12-byte access instead of seven 4-byte ones at a 28-byte stride
typedef float f4u_t __attribute__((ext_vector_type(4), aligned(4)));
typedef float f3u_t __attribute__((ext_vector_type(3), aligned(4)));
constexpr uint32_t LAT_MAX_CHUNKS = 32;   // 2048 lattice points per ray
constexpr uint32_t LAT_MAX_POINTS = LAT_MAX_CHUNKS * 64;
constexpr uint32_t SCAN_BLOCK = 1024;     // elements per scan block (256 threads x 4)
constexpr uint32_t K1_TICKET_CLASSES = 32; // sub-counters of k1_count's workgroup ticket
constexpr uint32_t K1_MAX_RANGE = 256;    // most slots a k1_count / k1_write workgroup owns (grid >= ceil(max rays / 256))

static __device__ __forceinline__ uint32_t k1_slot_to_ray(uint32_t li, uint32_t n_local) { return (uint32_t)(((uint64_t)li * K1_SCRAMBLE_PRIME) % n_local); }

// Thread per (ray, role).  The per-ray work is a long dependent chain of scalar arithmetic (twelve powf of the sRGB conversions of the
// target colour, normalisation, box intersection, stepping-space conversion) on fewer threads than the chip has lanes, so it is split
// into four independent roles on separate wavefronts (blockIdx.y): role 0 = ray geometry, roles 1..3 = one colour channel of the
// target / background each (the conversions are component-wise: same arithmetic, same results).  Measured: 48 -> see DESIGN 8.
// PLAIN (round 5): the instance for the datasets of BASELINE.json -- every image 8-bit, Perspective or OpenCV lens, still camera, uniform pixel sampling, no depth
// supervision (checked on the host: K1Args::plain_dataset and the launch).  The general instance carries seven lens models, the rolling-shutter slerp, three pixel
// formats and the CDF samplers: 37 KiB of instructions that these short-lived threads mostly fetch and skip (the slow class of boxes charges ~12 of the kernel's ~30 us
// for it, docs/ROUND_NOTES.md round 3).  Same functions with constant arguments, so the same arithmetic on the paths that remain.
template <bool PLAIN>
__global__ void __launch_bounds__(128) k1_setup(K1Args a, RaySetup* __restrict__ rs) {
	const uint32_t n_rays = a.n_rays_ptr ? *a.n_rays_ptr : a.n_rays;
	const uint32_t ray_begin = (uint32_t)(((uint64_t)n_rays * a.rank) / a.world_size);
	const uint32_t ray_end = (uint32_t)(((uint64_t)n_rays * (a.rank + 1)) / a.world_size);
	const uint32_t role = blockIdx.y;
	// grid-stride over the slots (round 6): the launch used to be sized for the ray cap (2^18 slots -> 8192 workgroups of which a step's ~6 10^4 rays use a quarter; the rest cost
	// dispatch time only); now a fixed grid covers 2^16 slots per pass
	for (uint32_t li = threadIdx.x + blockIdx.x * blockDim.x; li < ray_end - ray_begin; li += gridDim.x * blockDim.x) {
	// Slot li marches global ray ray_begin + pi(li), pi = a fixed bijection of [0, n_local) (multiplication by a prime > 2^18 x world):
	// slots are filled in slot order, so the rays that K1's sample cap (testbed_nerf.cu:813-815) and K3's batch clamp drop -- the LAST
	// slots -- are spread evenly over the ray range.  In plain index order they were always the rays of the last image(s)
	// (img = i * n_img / R): the reference drops whichever rays reserve their spans last (atomic order), i.e. a random subset, and the
	// index-ordered variant cost 0.2 - 0.35 dB of held-out PSNR at 5k - 20k steps (profiles/r02_bench_ab_psnr_before_scramble.json).
	const uint32_t i = ray_begin + k1_slot_to_ray(li, ray_end - ray_begin);
	Rng rng(a.rng);
	rng.advance((uint64_t)(i * N_RANDOM_PER_RAY));
	uint32_t img; float pix_pdf;
	f2 uv = training_pixel(PLAIN ? ErrorCdf{} : a.cdf, rng, i, n_rays, a.n_images, a.metadata, a.snap_to_pixel_centers, img, pix_pdf);
	const ngp_image_meta& m = a.metadata[img];
	const int pixel_type = PLAIN ? (int)NGP_IMAGE_BYTE : m.image_data_type;
	RaySetup& out = rs[li];
	if (role != 0) {
		// K3's target colour (testbed_nerf.cu:930-960): same rng stream position, same arithmetic -- channel c of every 3-vector
		const uint32_t c = role - 1;
		float tex_w; const float tc = read_rgba_channel(uv, m.resolution, m.pixels, pixel_type, c, tex_w);
		const bool masked = pixel_type == NGP_IMAGE_BYTE ? tc < 0.0f : read_rgba_masked(uv, m.resolution, m.pixels, pixel_type); // masked-away pixel (red < 0): the ray is not marched
		float tgt = 0.f, bg = 0.f;
		if (!masked) {
			(void)rng.next_float(); // motionblur_time
			bg = a.background_color[c];
			if (a.random_bg_color) { for (uint32_t k = 0; k <= c; ++k) bg = rng.next_float(); }
			bg = srgb_to_linear(bg);
			if (a.linear_colors || !a.color_space_srgb) {
				tgt = tc + (1.0f - tex_w) * bg;
				if (!a.linear_colors) { tgt = linear_to_srgb(tgt); bg = linear_to_srgb(bg); }
			} else {
				bg = linear_to_srgb(bg);
				if (tex_w > 0) tgt = linear_to_srgb(tc / tex_w) * tex_w + (1.0f - tex_w) * bg;
				else tgt = bg;
			}
		}
		out.tgt[c] = tgt; out.tgt[3 + c] = bg;
		continue;
	}
	const bool masked = read_rgba_masked(uv, m.resolution, m.pixels, pixel_type);
	const Box aabb(a.aabb);
	float o[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 1.f}, dn[3] = {0.f, 0.f, 1.f}, startt = 0.f, nprime = 0.f; uint32_t n_in = 0;
	if (!masked) {
		const float motionblur_time = rng.next_float();
		const M43 xform = PLAIN ? ldm43(a.xforms[img].start) : xform_given_rolling_shutter(a.xforms[img], m.rolling_shutter, uv, motionblur_time); // common_device.cuh:670-674 (a still camera's matrix is returned untouched)
		const int lens_mode = PLAIN ? (m.lens_mode == NGP_LENS_OPENCV ? (int)NGP_LENS_OPENCV : (int)NGP_LENS_PERSPECTIVE) : m.lens_mode;
		f3 ro, rd;
		if (!uv_to_ray(uv, m.resolution, m.focal_length, xform, m.principal_point, lens_mode, m.lens_params, 0.0f, ro, rd)) { ro = xform.c[3]; rd = xform.c[2]; } // testbed_nerf.cu:776-778
		const f3 rdn = normalize3(rd);
		f2 tminmax = aabb.ray_intersect(ro, rdn);
		tminmax.x = fmaxf(tminmax.x, 0.0f);
		startt = advance_n_steps(tminmax.x, a.cone_angle_constant, rng.next_float());
		o[0] = ro.x; o[1] = ro.y; o[2] = ro.z; d[0] = rd.x; d[1] = rd.y; d[2] = rd.z; dn[0] = rdn.x; dn[1] = rdn.y; dn[2] = rdn.z;
		nprime = to_stepping_space(startt, a.cone_angle_constant);
		// The lattice points inside the box are a PREFIX of the lattice (every coordinate of ro + t_j * rdn is monotonic in j, in fp32 as well): their number by bisection
		// with the marchers' own arithmetic.  0 = the march starts outside the box (no samples).
		if (aabb.contains(ro + startt * rdn)) {
			uint32_t lo = 0u, hi = LAT_MAX_POINTS; // invariant: point lo is inside, point hi is outside (or past the lattice)
			while (hi - lo > 1u) {
				const uint32_t mid = (lo + hi) >> 1;
				if (aabb.contains(ro + from_stepping_space(nprime + (float)mid, a.cone_angle_constant) * rdn)) lo = mid; else hi = mid;
			}
			n_in = hi;
		}
		// K3's target depth (testbed_nerf.cu:1027): distance along the unnormalised direction; <= 0 = not supervised
		out.tgt[6] = sqrtf(dot3(rd, rd)) * ((!PLAIN && a.depth_lambda > 0.0f && m.depth) ? read_depth(uv, m.resolution, m.depth) : -1.0f);
	} else out.tgt[6] = -1.0f;
	out.o[0] = o[0]; out.o[1] = o[1]; out.o[2] = o[2]; out.d[0] = d[0]; out.d[1] = d[1]; out.d[2] = d[2]; out.rdn[0] = dn[0]; out.rdn[1] = dn[1]; out.rdn[2] = dn[2];
	out.startt = startt; out.nprime = nprime; out.count = 0; out.flags = n_in; out.ray_index = i; out.img = img;
	if (!a.ray_targets_out) for (int k = 0; k < 6; ++k) out.tgt[k] = 0.f;
	}
}

static __device__ __forceinline__ float lattice_t(const RaySetup& r, uint32_t j, float cone_angle) {
	return j == 0 ? r.startt : from_stepping_space(r.nprime + (float)j, cone_angle);
}

// ---- exclusive prefix sum over packed {samples (low 32), rays (high 32)} ------------------------
template <uint32_t NW = 4 /* wavefronts per workgroup */>
static __device__ __forceinline__ uint64_t block_excl_scan_1024(uint64_t v[4], uint64_t* sm /* NW */, uint64_t& block_total) {
	// each thread owns 4 consecutive elements; returns the exclusive prefix of the thread's first element
	const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
	uint64_t tsum = v[0] + v[1] + v[2] + v[3];
	uint64_t x = tsum;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		uint64_t y = __shfl_up((unsigned long long)x, d, 64);
		if (lane >= (uint32_t)d) x += y;
	}
	if (lane == 63) sm[wid] = x;
	__syncthreads();
	uint64_t woff = 0;
	for (uint32_t w = 0; w < wid; ++w) woff += sm[w];
	block_total = sm[0] + sm[1] + sm[2] + sm[3];
	if constexpr (NW == 8) block_total += (sm[4] + sm[5]) + (sm[6] + sm[7]);
	return woff + x - tsum;
}
// Tail of both counting kernels: publish the workgroup's packed {samples, rays} total; the last workgroup to arrive turns the totals into exclusive offsets + the two global counters.
template <uint32_t NW = 4 /* wavefronts per workgroup */>
static __device__ __forceinline__ void k1_publish_and_scan(const K1Args& a, uint64_t wave_total, uint64_t* __restrict__ partial, uint32_t* __restrict__ done, uint64_t* s_tot /* NW */, uint64_t* s_scan /* NW */, uint32_t& s_ticket) {
	const uint32_t lane = threadIdx.x & 63u;
	if (lane == 0) s_tot[threadIdx.x >> 6] = wave_total;
	__syncthreads();
	if (threadIdx.x == 0) {
		// The total is published with a RETURNING device-scope atomic (performed at the coherence point; its return value is awaited before the
		// ticket is drawn) instead of store + __threadfence(): an agent-scope release fence writes back the XCD's whole L2 on this multi-XCD
		// part, and 2048 of them made this kernel 47 us slower (profiles/r02_k1_experiments.txt).
		uint64_t block_sum = (s_tot[0] + s_tot[1]) + (s_tot[2] + s_tot[3]);
		if constexpr (NW == 8) block_sum += (s_tot[4] + s_tot[5]) + (s_tot[6] + s_tot[7]);
		const uint64_t prev = atomicExch((unsigned long long*)(partial + blockIdx.x), (unsigned long long)block_sum);
		// two-level ticket: one counter word retires only ~90 returning atomics per microsecond, so workgroup b draws from sub-counter
		// b % 32 and only the last of each residue class draws from the top counter
		const uint32_t cls = blockIdx.x % K1_TICKET_CLASSES, n_cls = (gridDim.x - cls + K1_TICKET_CLASSES - 1) / K1_TICKET_CLASSES;
		uint32_t last = 0u;
		if (atomicAdd(done + 1 + cls, (uint32_t)(prev >> 63) + 1u) == n_cls - 1) { // (prev >> 63 == 0: the data dependence orders the ticket behind the exchange)
			atomicExch(done + 1 + cls, 0u);
			last = atomicAdd(done, 1u) == min(gridDim.x, K1_TICKET_CLASSES) - 1 ? 1u : 0u;
		}
		s_ticket = last;
	}
	__syncthreads();
	if (!s_ticket) return;
	// last workgroup: every total has been exchanged in; read them at the coherence point as well (atomic RMW with 0)
	uint64_t run = 0ull;
	for (uint32_t b0 = 0; b0 < gridDim.x; b0 += NW * 256u) {
		uint64_t v[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) { const uint32_t e = b0 + threadIdx.x * 4 + k; v[k] = e < gridDim.x ? (uint64_t)atomicAdd((unsigned long long*)(partial + e), 0ull) : 0ull; }
		uint64_t tot;
		uint64_t pre = run + block_excl_scan_1024<NW>(v, s_scan, tot);
#pragma unroll
		for (int k = 0; k < 4; ++k) { const uint32_t e = b0 + threadIdx.x * 4 + k; if (e < gridDim.x) partial[e] = pre; pre += v[k]; }
		run += tot;
		__syncthreads(); // s_scan is reused by the next round
	}
	if (threadIdx.x == 0) { *a.numsteps_counter = (uint32_t)run; *a.ray_counter = (uint32_t)(run >> 32); atomicExch(done, 0u); }
}

// Prefix sum over the per-ray counts without scan launches: workgroup b owns the CONTIGUOUS slot range [b n / G, (b + 1) n / G) (slots are
// scrambled rays, so the ranges are statistically equal), publishes the packed {samples, rays} total of its range, and the last workgroup to
// finish (ticket counter) turns the G totals into exclusive offsets + the two global counters.  k1_write, launched with the same G, re-scans
// the <= 128 counts of its own range in LDS.  Same slot-ordered spans as a global scan: deterministic, no atomics on the sample buffer.
// K1_GROUP = lattice chunks tested (= occupancy loads in flight) per iteration.  SINGLE_CASCADE (aabb_scale 1: max_mip == 0, AND a constant step: cone_angle == 0,
// i.e. dt * 256 < 1): every point's mip is 0 (mip_from_dt returns mip_from_pos), so dt, both frexp chains and the mip scaling drop out -- the kernel is VALU bound (~110
// instructions per lattice point, 4 cycles each per wavefront), this removes a third of them.
template <uint32_t K1_GROUP, bool SINGLE_CASCADE>
__global__ void __launch_bounds__(256) k1_count(K1Args a, RaySetup* __restrict__ rs, uint64_t* __restrict__ masks, uint64_t* __restrict__ partial, uint32_t* __restrict__ done) {
	__shared__ uint64_t s_tot[4];
	__shared__ uint64_t s_scan[4];
	__shared__ uint32_t s_ticket;
	const uint32_t n_rays = a.n_rays_ptr ? *a.n_rays_ptr : a.n_rays;
	const uint32_t ray_begin = (uint32_t)(((uint64_t)n_rays * a.rank) / a.world_size);
	const uint32_t ray_end = (uint32_t)(((uint64_t)n_rays * (a.rank + 1)) / a.world_size);
	const uint32_t n_local = ray_end - ray_begin;
	// Persistent grid: the launch is sized for the ray CAP (2^18) while a step marches ~5*10^4 rays, and a kernel of 262k
	// one-ray wavefronts is bound by the wavefront launch rate (~1-2.5 waves/clk chip-wide, measured 85 us) -- so a fixed
	// number of wavefronts loops over the rays instead.
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t li_begin = (uint32_t)(((uint64_t)n_local * blockIdx.x) / gridDim.x), li_end = (uint32_t)(((uint64_t)n_local * (blockIdx.x + 1)) / gridDim.x);
	// coarse occupancy (one bit per 4x4x4 cells) of every cascade in LDS: most lattice points lie in empty space and are rejected without a memory
	// access, and a chunk without any coarse hit costs no memory latency at all
	extern __shared__ uint32_t s_coarse[];
	const bool prefilter = a.bitfield_coarse != nullptr && a.bitfield_linear != nullptr;
	if (prefilter && li_begin < li_end) {
		// every level the march can ask for: mip_from_dt returns max(mip_from_pos, exponent of the step), which exceeds max_mip for long steps (pooled levels)
		// (in LDS: the first n_mips_lds levels -- the dataset's cascades and the next pooled ones, where nearly every point falls; a point of a higher level reads the coarse bit from
		// memory: 4 KiB of LDS per level decide how many workgroups a CU holds)
		const uint32_t n_words = (SINGLE_CASCADE ? 1u : a.n_mips_lds) * COARSE_WORDS;
		for (uint32_t w = threadIdx.x; w < n_words; w += blockDim.x) s_coarse[w] = a.bitfield_coarse[w];
	}
	__syncthreads();
	uint64_t wave_total = 0ull; // packed {samples (low 32), rays with samples (high 32)} of this wavefront's rays
	// Round 6: the workgroup marches FOUR rays at a time in two phases.  A ray's chunks used to be one wavefront's serial work -- and a launch lasts as long as its longest ray: on
	// fox (aabb_scale 4: ~4.3 k rays per step, ~7 chunks on average, 32 at most) 8192 wavefronts held one ray or none and the kernel took 125 us for 12 us of average work.
	//   phase A: the 4 x 4 (ray, group of eight chunks) evaluations of the batch, spread over the four wavefronts so that ONE ray's groups run side by side (task (k, g) -> wavefront
	//            (k + g) mod 4); the chunk records {occupied mask, inside mask, mip, uniform?, per-point skip length} go to LDS;
	//   phase B: wavefront k replays ray k's exit tests / exact-skip orbit over the records in chunk order -- the loop that used to follow each group's evaluation, word for word,
	//            so masks, counts and n_chunks are those of the one-wavefront-per-ray kernel.
	const uint32_t wid = threadIdx.x >> 6;
	const uint32_t coarse_words = prefilter ? (SINGLE_CASCADE ? 1u : a.n_mips_lds) * COARSE_WORDS : 0u;
	uint64_t* s_m = (uint64_t*)(s_coarse + coarse_words);               // [4][LAT_MAX_CHUNKS]
	uint64_t* s_in = s_m + 4 * LAT_MAX_CHUNKS;                          // [4][LAT_MAX_CHUNKS]
	uint16_t* s_skip = (uint16_t*)(s_in + 4 * LAT_MAX_CHUNKS);          // [4][LAT_MAX_CHUNKS][64]
	uint8_t* s_mip0 = (uint8_t*)(s_skip + 4 * LAT_MAX_CHUNKS * 64);     // [4][LAT_MAX_CHUNKS]: the chunk's mip (of its first point) | 0x80 if every inside point shares it
	uint8_t* s_mark = s_mip0 + 4 * LAT_MAX_CHUNKS + wid * 64u;          // [4][64]: this wavefront's visited marks of the chunk it walks (phase B)
	// The reference's skip rule differs from "every lattice point on its own" only where the mip changes along a skipped voxel, and the
	// mip of a lattice point depends on dt only when cone_angle > 0 (mip_from_dt): with cone_angle == 0 both are the same algorithm.
	const bool exact_skip = a.exact_skip != 0 && a.cone_angle_constant > 1e-5f;
	const Box aabb(a.aabb);
	static_assert(LAT_MAX_CHUNKS == 4 * K1_GROUP, "k1_count: four groups of K1_GROUP chunks per ray, one per wavefront");
	for (uint32_t lb = li_begin; lb < li_end; lb += 4) {
	// ---- phase A ----
#pragma unroll 1
	for (uint32_t k = 0; k < 4; ++k) {
		const uint32_t li = lb + k, g = (wid + 4u - k) & 3u, ch0 = g * K1_GROUP;
		if (li >= li_end) break;
		const RaySetup& r = rs[li];
		if (!r.flags) continue; // the march starts outside the box (k1_setup): no record is read
		const f3 ro = ld3(r.o), rdn = normalize3(ld3(r.d));
		const f3 idir = mk3(1.0f) / rdn;
		const float startt = r.startt, nprime = r.nprime;
		auto lat_t = [&](uint32_t j) { return j == 0 ? startt : from_stepping_space(nprime + (float)j, a.cone_angle_constant); }; // lattice_t
		// one lattice point per lane: inside the box? occupied at its own mip? (64 consecutive lattice points span ~14 voxels: the byte
		// loads of a wavefront coalesce into a few L1/L2 lines, and thousands of resident wavefronts hide their latency)
		auto eval_point = [&](uint32_t j, bool want_skip, bool& inside, bool& occ, uint32_t& mip, uint32_t& skip) {
			const float t = lat_t(j);
			const f3 pos = ro + t * rdn;
			inside = aabb.contains(pos);
			occ = false; mip = 0u; skip = 1u;
			if (inside) {
				if (!SINGLE_CASCADE) {
					const float dt = calc_dt(t, a.cone_angle_constant);
					mip = k1_mip(a, dt, pos);
				}
				occ = prefilter ? ((SINGLE_CASCADE || mip < a.n_mips_lds) ? occupied_at_linear_prefiltered(pos, a.bitfield_linear, s_coarse, mip) : occupied_at_linear_prefiltered(pos, a.bitfield_linear, a.bitfield_coarse, mip))
					: a.bitfield_linear ? occupied_at_linear(pos, a.bitfield_linear, mip) : occupied_at(pos, a.bitfield, mip);
				if (want_skip && !occ) {
					// advance_to_next_voxel (nerf_device.cuh:431-441) lands on lattice point j + ceil(max(to(t_target) - to(t), 0.5))
					const float res = scalbnf((float)GRIDSIZE, -(int)mip);
					const float t_target = t + distance_to_next_voxel(pos, rdn, idir, res);
					skip = (uint32_t)ceilf(fmaxf(to_stepping_space(t_target, a.cone_angle_constant) - to_stepping_space(t, a.cone_angle_constant), 0.5f));
				}
			}
		};
		// The in-box lattice points of a ray are a contiguous range (every coordinate of ro + t * rdn is monotonic in t, in fp32 as well, and so is t in
		// j), so a chunk whose FIRST point is outside the box lies wholly outside: one test per chunk (lane u = chunk u of the group) instead of 64 point
		// evaluations (same masks: a chunk evaluated outside the box yields exactly these values).
		const uint32_t first_in = (uint32_t)(__ballot(lane < K1_GROUP && aabb.contains(ro + lat_t((ch0 + (lane < K1_GROUP ? lane : 0u)) * 64u) * rdn)) & ((1ull << K1_GROUP) - 1ull));
		// Eight chunks per task so that eight independent occupancy loads are in flight (a chain of dependent ~1 us loads otherwise)
#pragma unroll
		for (uint32_t u = 0; u < K1_GROUP; ++u) {
			uint64_t m = 0ull, in = 0ull; uint32_t mip0 = 0u; bool uni = true; uint32_t skip = 1u;
			if (((first_in >> u) & 1u) || a.no_first_point_skip) {
				bool inside, occ; uint32_t mip;
				eval_point((ch0 + u) * 64 + lane, exact_skip, inside, occ, mip, skip);
				m = __ballot(occ);
				in = __ballot(inside);
				mip0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)mip);
				uni = __ballot(inside && mip != mip0) == 0ull;
			}
			const uint32_t rec = k * LAT_MAX_CHUNKS + ch0 + u;
			if (lane == 0) { s_m[rec] = m; s_in[rec] = in; s_mip0[rec] = (uint8_t)(mip0 | (uni ? 0x80u : 0u)); }
			if (exact_skip) s_skip[rec * 64u + lane] = (uint16_t)min(skip, 0xffffu); // (a jump past the lattice's 2048 points ends the march whatever its length)
		}
	}
	__syncthreads();
	// ---- phase B ----
	if (lb + wid < li_end) {
	const uint32_t li = lb + wid;
	const uint32_t ray_flags = rs[li].flags;
	uint32_t cnt = 0, n_chunks = 0;
	if (ray_flags) {
		bool done = false, prev_walked = false;
		uint32_t jnext = 0;        // exact skip: next lattice point the reference's loop visits (valid while prev_walked)
		uint32_t prev_mip = 0xffu; // mip of the previous chunk if it was uniform
		// (Round 3 also tried a per-ray bracket: <= 31 points along the ray tested against the dilated coarse grid, only the chunks between the first and the
		// last hit evaluated, rays without a hit rejected outright.  Exact -- all K1 tests passed with it -- but worth nothing on the bench scene: 0.188 vs 0.190 ms
		// for K1, profiles/r03_microbench_k1_coarse_bracket_no_gain.log: a ray that misses every occupied cell still passes within one coarse cell of the
		// dilated occupancy almost always.  Removed again.)
		for (uint32_t ch0 = 0; ch0 < LAT_MAX_CHUNKS && !done; ch0 += K1_GROUP) {
			uint64_t m[K1_GROUP], in[K1_GROUP];
			uint32_t mip0[K1_GROUP]; bool uni[K1_GROUP]; // exact skip: the chunk's mip (of its first point) / is it the same for all points inside the box?
#pragma unroll
			for (uint32_t u = 0; u < K1_GROUP; ++u) {
				const uint32_t rec = wid * LAT_MAX_CHUNKS + ch0 + u;
				m[u] = s_m[rec]; in[u] = s_in[rec];
				const uint32_t q = s_mip0[rec]; mip0[u] = q & 0x7fu; uni[u] = (q & 0x80u) != 0u;
			}
			// the exit tests are replayed in chunk order, so masks, counts and n_chunks are exactly those of the one-chunk-at-a-time loop
#pragma unroll
			for (uint32_t u = 0; u < K1_GROUP; ++u) {
				if (done || cnt >= N_STEPS) { done = true; break; }
				uint64_t sm = m[u];
				if (exact_skip) {
					// A chunk whose points all share one mip, between neighbours of that same mip, is "plain": whatever the reference's loop
					// skips there is empty at that mip anyway, so the independent test is exact and no skip has to be followed.  Around a mip
					// change (and at the end of a group, where the next chunk is not known yet) the orbit of the reference's update rule is
					// walked: runs of occupied points are taken whole (j -> j + 1), an empty point jumps by its own skip length.
					const bool next_same = u + 1 < K1_GROUP && (in[u + 1 < K1_GROUP ? u + 1 : u] == 0ull || (uni[u + 1 < K1_GROUP ? u + 1 : u] && mip0[u + 1 < K1_GROUP ? u + 1 : u] == mip0[u]));
					const bool plain = uni[u] && prev_mip == mip0[u] && next_same;
					if (!plain) {
						// The orbit of the reference's update rule through the chunk: an occupied point goes to its successor, an empty one jumps by its skip length, a point
						// outside the box ends the march.  One scalar step per visited point (~100 cycles each) made a far chunk -- step ~ voxel, every jump one or two points
						// long -- cost ~5 us, and a fox ray has thirty of them: the kernel's critical path.  Here the orbit is marked by pointer doubling over the 64 lanes
						// (round k: the visited set, a prefix of the orbit of length 2^k, is joined by its image under next^(2^k)): <= 6 rounds of three LDS operations,
						// ended as soon as a round adds nothing.
						const uint32_t skip = s_skip[(size_t)(wid * LAT_MAX_CHUNKS + ch0 + u) * 64u + lane]; // the jump lengths of the chunk's points (phase A), one per lane
						const uint32_t base = (ch0 + u) * 64;
						const uint32_t j0 = (prev_walked && jnext > base) ? jnext - base : 0u; // behind a plain chunk of the same mip the entry point is immaterial
						const uint32_t nxt = ((m[u] >> lane) & 1ull) ? lane + 1u : ((in[u] >> lane) & 1ull) ? lane + skip : 0x10000u;
						uint64_t V = j0 < 64u ? 1ull << j0 : 0ull;
						if (V) {
							uint32_t J = nxt;
							s_mark[lane] = 0;
#pragma unroll 1
							for (int k = 0; k < 6; ++k) {
								if (((V >> lane) & 1ull) && J < 64u) s_mark[J] = 1;
								__builtin_amdgcn_wave_barrier(); // (a wavefront's LDS operations complete in order)
								const uint64_t Vn = V | __ballot(s_mark[lane] != 0);
								if (Vn == V) break;
								V = Vn;
								const uint32_t Jn = (uint32_t)__shfl((int)J, (int)(J & 63u), 64);
								J = J < 64u ? Jn : J;
							}
							__builtin_amdgcn_wave_barrier();
							sm = V & m[u];
							if (V & ~in[u]) done = true; // the march left the box
							const uint32_t jlast = 63u - (uint32_t)__builtin_clzll(V);
							jnext = base + (uint32_t)__builtin_amdgcn_readlane((int)nxt, (int)jlast);
						} else { sm = 0ull; jnext = base + j0; }
					}
					prev_walked = !plain;
					prev_mip = uni[u] ? mip0[u] : 0xffu;
				}
				if (lane == 0) masks[(size_t)li * LAT_MAX_CHUNKS + ch0 + u] = sm;
				cnt += (uint32_t)__popcll(sm);
				n_chunks = ch0 + u + 1;
				if (in[u] == 0ull) done = true; // the whole chunk is past the box: so is everything after it
			}
		}
		cnt = min(cnt, N_STEPS);
	}
	if (lane == 0) {
		rs[li].count = cnt;
		rs[li].flags = n_chunks;
	}
	wave_total += (uint64_t)cnt | ((uint64_t)(cnt > 0 ? 1u : 0u) << 32);
	}
	__syncthreads(); // the records are rewritten by the next batch (and rs[].flags, which phase A reads, by this phase)
	}
	k1_publish_and_scan(a, wave_total, partial, done, s_tot, s_scan, s_ticket);
}

__global__ void __launch_bounds__(256) k1_write(K1Args a, const RaySetup* __restrict__ rs, const uint64_t* __restrict__ masks, const uint64_t* __restrict__ partial) {
	__shared__ uint64_t s_scan[4];
	__shared__ uint64_t s_off[K1_MAX_RANGE];
	const uint32_t n_rays = a.n_rays_ptr ? *a.n_rays_ptr : a.n_rays;
	const uint32_t max_samples = a.max_samples_ptr ? min(*a.max_samples_ptr, a.max_samples) : a.max_samples;
	const uint32_t ray_begin = (uint32_t)(((uint64_t)n_rays * a.rank) / a.world_size);
	const uint32_t ray_end = (uint32_t)(((uint64_t)n_rays * (a.rank + 1)) / a.world_size);
	const uint32_t lane = threadIdx.x & 63u;
	const Box aabb(a.aabb);
	const uint32_t cs = 7u + a.n_extra;
	// this workgroup's slot range (the same split as k1_count's) and the spans of its rays: offset of the range + exclusive scan inside it
	const uint32_t n_local = ray_end - ray_begin;
	const uint32_t li_begin = (uint32_t)(((uint64_t)n_local * blockIdx.x) / gridDim.x), li_end = (uint32_t)(((uint64_t)n_local * (blockIdx.x + 1)) / gridDim.x);
	{
		uint64_t v[4] = {0ull, 0ull, 0ull, 0ull};
		if (li_begin + threadIdx.x < li_end) { const uint32_t c = rs[li_begin + threadIdx.x].count; v[0] = (uint64_t)c | ((uint64_t)(c > 0 ? 1u : 0u) << 32); }
		uint64_t tot;
		const uint64_t pre = block_excl_scan_1024(v, s_scan, tot);
		if (threadIdx.x < K1_MAX_RANGE) s_off[threadIdx.x] = partial[blockIdx.x] + pre;
		__syncthreads();
	}
	for (uint32_t li = li_begin + (threadIdx.x >> 6); li < li_end; li += 4) { // persistent grid, see k1_count
	const RaySetup r = rs[li];
	const uint32_t count = r.count;
	if (count == 0) continue;
	const uint64_t so = s_off[li - li_begin];
	const uint32_t base = (uint32_t)so, slot = (uint32_t)(so >> 32);
	const bool fits = base + count <= max_samples; // testbed_nerf.cu:813-815: rays past the cap are dropped
	if (lane == 0) {
		a.ray_indices_out[slot] = r.ray_index;
		ngp_ray rr; rr.o[0] = r.o[0]; rr.o[1] = r.o[1]; rr.o[2] = r.o[2]; rr.d[0] = r.d[0]; rr.d[1] = r.d[1]; rr.d[2] = r.d[2];
		a.rays_out[slot] = rr;
		a.numsteps_out[slot * 2 + 0] = fits ? count : 0u;
		a.numsteps_out[slot * 2 + 1] = base;
		if (a.k2_tiles0_out) { // lazy K2, round 0: the first 32 samples of this ray (dropped rays: an empty tile)
			const uint32_t c0 = fits ? count : 0u;
			a.k2_tiles0_out[slot] = make_uint4(base, min(c0, a.k2_tile_w), slot, c0 - min(c0, a.k2_tile_w));
		}
		if (a.ray_targets_out) {
			float4* tg = (float4*)(a.ray_targets_out + (size_t)slot * 8);
			tg[0] = make_float4(r.tgt[0], r.tgt[1], r.tgt[2], r.tgt[3]); tg[1] = make_float4(r.tgt[4], r.tgt[5], r.tgt[6], 0.f);
		}
	}
	if (!fits) continue;
	const f3 ro = ld3(r.o), rdn = normalize3(ld3(r.d));
	const f3 wd = warp_direction(rdn);
	float* co = a.coords_out + (size_t)base * cs;
	uint32_t written = 0;
	const uint32_t n_chunks = r.flags;
	// all chunk masks of the ray with ONE load (lane = chunk), then broadcast from registers: the loop has no memory latency in it
	const uint64_t my_mask = lane < n_chunks ? masks[(size_t)li * LAT_MAX_CHUNKS + lane] : 0ull;
	for (uint32_t ch = 0; ch < n_chunks && written < count; ++ch) {
		const uint64_t m = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_mask >> 32), (int)ch) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_mask, (int)ch);
		const uint32_t k = written + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
		if (((m >> lane) & 1ull) && k < count) {
			const float t = lattice_t(r, ch * 64 + lane, a.cone_angle_constant);
			const f3 pos = ro + t * rdn;
			const float dt = calc_dt(t, a.cone_angle_constant);
			const f3 wp = warp_position(pos, aabb);
			float* c = co + (size_t)k * cs;
			c[0] = wp.x; c[1] = wp.y; c[2] = wp.z; c[3] = warp_dt(dt); c[4] = wd.x; c[5] = wd.y; c[6] = wd.z;
			for (uint32_t x = 7; x < cs; ++x) c[x] = a.extra_dims[(size_t)r.img * a.n_extra + (x - 7)];
		}
		written += (uint32_t)__popcll(m);
	}
	}
}

// ---- K1 for one cascade and a constant step (aabb_scale 1, cone_angle 0: the headline configuration), round 4 ------------------------------------------------------------
// k1_count evaluates every lattice point up to the ray's exit -- 3.8 10^7 points per step for the 1.7 10^6 that become samples -- at ~70 VALU instructions per point: the
// kernel is bound by instruction issue, not by memory.  Here the lattice is cut into SEGMENTS of 8 points.  k1_setup (thread per ray) has counted the points inside the box
// (a prefix of the lattice, by bisection); one prepass lane per segment tests (conservatively) whether anything is occupied near the segment: its midpoint against the
// dilated mid grid in LDS (k_build_mid_dilated_bitfield).  The segments that pass are compacted into a list (prefix popcount -> LDS), and only their points are evaluated,
// sixteen segments (two independent occupancy loads per lane) per iteration, with exactly the arithmetic of k1_count's eval_point.  Candidates are visited in lattice
// order, so the ballot of the occupied ones IS the ray's sample list in output order: the kernel stores each sample's lattice index j (16 bits, N_STEPS entries per slot)
// and k1_write_list reads it back -- one lane per SAMPLE, where k1_write spends a 64-lane iteration per chunk on ~3 samples.  Same samples bit for bit
// (tests/test_gpu_nerf.py::test_k1_ablation_variants_are_the_same_marcher runs the chunk kernels as the ablation).
constexpr uint32_t K1_SEG = 8;                                   // lattice points per segment
constexpr uint32_t K1_MAX_SEGS = LAT_MAX_POINTS / K1_SEG;         // 256 per ray
// this workgroup's contiguous slot range (32-bit arithmetic when it cannot overflow: the 64-bit division is ~110 scalar instructions)
static __device__ __forceinline__ void k1_slot_range(uint32_t n_local, uint32_t b, uint32_t g, uint32_t& li_begin, uint32_t& li_end) {
	if (n_local <= (1u << 19) && g <= (1u << 12)) { li_begin = n_local * b / g; li_end = n_local * (b + 1u) / g; }
	else { li_begin = (uint32_t)(((uint64_t)n_local * b) / g); li_end = (uint32_t)(((uint64_t)n_local * (b + 1u)) / g); }
}
// NW = wavefronts per workgroup (round 6): the 36 KiB LDS image allows four workgroups per CU whatever their size, the kernel holds < 64 registers and waits for its loads
// (RaySetup -> segment list -> occupancy bytes, a dependent chain per ray) 62 % of its wave cycles: eight wavefronts per workgroup = 8 per SIMD instead of 4.
template <uint32_t NW>
__global__ void __launch_bounds__(64 * NW) k1_count_segments(K1Args a, RaySetup* __restrict__ rs, uint16_t* __restrict__ jlist, uint64_t* __restrict__ partial, uint32_t* __restrict__ done) {
	__shared__ uint64_t s_tot[NW];
	__shared__ uint64_t s_scan[NW];
	__shared__ uint32_t s_ticket;
	__shared__ uint32_t s_next; // next slot of the workgroup's range to hand out: a ray costs between nothing (misses the box) and hundreds of lattice points, so the wavefronts take rays as they finish (round 6) instead of every NW-th one
	__shared__ uint8_t s_seg[NW][K1_MAX_SEGS]; // per wavefront: the segments of its ray that passed the prepass, in lattice order
	extern __shared__ uint32_t s_pre[];       // [COARSE_WORDS coarse of cascade 0][MID_WORDS dilated mid grid]
	const uint32_t n_rays = a.n_rays_ptr ? *a.n_rays_ptr : a.n_rays;
	const uint32_t ray_begin = (uint32_t)(((uint64_t)n_rays * a.rank) / a.world_size);
	const uint32_t ray_end = (uint32_t)(((uint64_t)n_rays * (a.rank + 1)) / a.world_size);
	const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
	uint32_t li_begin, li_end; k1_slot_range(ray_end - ray_begin, blockIdx.x, gridDim.x, li_begin, li_end);
	if (threadIdx.x == 0) s_next = li_begin + NW; // (the first NW rays are taken by wavefront index)
	if (li_begin < li_end) { // 36 KiB per workgroup: all 16-byte loads of a thread in flight before the first LDS store
		static_assert(COARSE_WORDS == 4 * 256 && MID_WORDS == 8 * 4 * 256, "one + eight uint4 per thread of a 256-thread workgroup");
		const uint4* src_c = (const uint4*)a.bitfield_coarse; const uint4* src_m = (const uint4*)(a.bitfield_coarse + (size_t)a.n_mips * COARSE_WORDS);
		constexpr uint32_t T = 64u * NW, MK = 8u * 256u / T; // mid-grid uint4 per thread
		uint4 v[1 + MK];
		if (threadIdx.x < 256u) v[0] = src_c[threadIdx.x];
#pragma unroll
		for (uint32_t k = 0; k < MK; ++k) v[1 + k] = src_m[threadIdx.x + T * k];
		uint4* dst = (uint4*)s_pre;
		if (threadIdx.x < 256u) dst[threadIdx.x] = v[0];
#pragma unroll
		for (uint32_t k = 0; k < MK; ++k) dst[256u + threadIdx.x + T * k] = v[1 + k];
	}
	__syncthreads();
	const uint32_t* s_mid = s_pre + COARSE_WORDS;
	uint64_t wave_total = 0ull;
	for (uint32_t li = li_begin + wid; li < li_end; ) {
		const RaySetup& r = rs[li];
		const uint32_t n_in = r.flags; // lattice points inside the box: [0, n_in)
		uint32_t cnt = 0;
		if (n_in) {
			const f3 ro = ld3(r.o), rdn = ld3(r.rdn);
			const float startt = r.startt, nprime = r.nprime;
			const uint32_t n_segs = (n_in + K1_SEG - 1u) / K1_SEG;
			// prepass: one lane per segment
			uint32_t n_hit = 0;
			for (uint32_t s0 = 0; s0 < n_segs; s0 += 64u) {
				const uint32_t sg = s0 + lane;
				const f3 pm = ro + ((nprime + ((float)(sg * K1_SEG) + 0.5f * (float)(K1_SEG - 1))) * MIN_CONE_STEP) * rdn;
				const uint32_t mx = (uint32_t)clampi((int)(pm.x * (float)MID_SIZE), 0, (int)MID_SIZE - 1), my = (uint32_t)clampi((int)(pm.y * (float)MID_SIZE), 0, (int)MID_SIZE - 1),
					mz = (uint32_t)clampi((int)(pm.z * (float)MID_SIZE), 0, (int)MID_SIZE - 1);
				const uint32_t midx = mx + MID_SIZE * (my + MID_SIZE * mz);
				const bool hit = sg < n_segs && ((s_mid[midx >> 5] >> (midx & 31u)) & 1u) != 0u;
				const uint64_t hm = __ballot(hit);
				if (hit) s_seg[wid][n_hit + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))] = (uint8_t)sg;
				n_hit += (uint32_t)__popcll(hm);
			}
			__builtin_amdgcn_wave_barrier(); // the list is read by other lanes of the same wavefront (LDS operations of a wavefront complete in order)
			uint16_t* jl = jlist + (size_t)li * N_STEPS;
			for (uint32_t h0 = 0; h0 < n_hit && cnt < N_STEPS; h0 += 16u) {
				bool occ[2]; uint32_t jj[2];
#pragma unroll
				for (uint32_t u = 0; u < 2; ++u) {
					const uint32_t h = min(h0 + 8u * u + (lane >> 3), n_hit - 1u);
					const uint32_t j = (uint32_t)s_seg[wid][h] * K1_SEG + (lane & 7u);
					jj[u] = j;
					const f3 pos = ro + (j == 0u ? startt : (nprime + (float)j) * MIN_CONE_STEP) * rdn; // lattice_t at cone_angle 0
					occ[u] = h0 + 8u * u + (lane >> 3) < n_hit && j < n_in && occupied_at_linear_prefiltered(pos, a.bitfield_linear, s_pre, 0u);
				}
#pragma unroll
				for (uint32_t u = 0; u < 2; ++u) {
					const uint64_t m = __ballot(occ[u]);
					const uint32_t k = cnt + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
					if (occ[u] && k < N_STEPS) jl[k] = (uint16_t)jj[u];
					cnt += (uint32_t)__popcll(m);
				}
			}
			cnt = min(cnt, N_STEPS);
			__builtin_amdgcn_wave_barrier(); // s_seg is rewritten for the next ray
		}
		if (lane == 0) rs[li].count = cnt;
		wave_total += (uint64_t)cnt | ((uint64_t)(cnt > 0 ? 1u : 0u) << 32);
		uint32_t nxt = 0u;
		if (lane == 0) nxt = atomicAdd(&s_next, 1u);
		li = (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt);
	}
	k1_publish_and_scan<NW>(a, wave_total, partial, done, s_tot, s_scan, s_ticket);
}

// One lane per SAMPLE of the workgroup's slot range (a contiguous span of the sample buffer): the sample's ray by bisection over the range's span offsets in LDS, its lattice
// index from the list k1_count_segments left, the position with k1_write's arithmetic; the 7-float records of 64 samples are transposed through LDS so that the wavefront
// stores 7 x 256 contiguous bytes instead of 7 x 64 dwords at a 28-byte stride.  K1_WRITE_SPLIT workgroups share one slot range (its 64-sample batches interleaved), so
// that the grid has twice the wavefronts of the counting kernel's (whose LDS image limits it to four workgroups per CU).
constexpr uint32_t K1_WRITE_SPLIT = 2;
__global__ void __launch_bounds__(256) k1_write_list(K1Args a, const RaySetup* __restrict__ rs, const uint16_t* __restrict__ jlist, const uint64_t* __restrict__ partial) {
	__shared__ uint64_t s_scan[4];
	__shared__ uint32_t s_base[K1_MAX_RANGE + 1]; // exclusive prefix of the range's sample counts (relative to the range's first sample)
	__shared__ uint32_t s_cut;                    // first sample (relative) of the first ray that does not fit under the sample cap
	const uint32_t n_rays = a.n_rays_ptr ? *a.n_rays_ptr : a.n_rays;
	const uint32_t max_samples = a.max_samples_ptr ? min(*a.max_samples_ptr, a.max_samples) : a.max_samples;
	const uint32_t ray_begin = (uint32_t)(((uint64_t)n_rays * a.rank) / a.world_size);
	const uint32_t ray_end = (uint32_t)(((uint64_t)n_rays * (a.rank + 1)) / a.world_size);
	const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
	const Box aabb(a.aabb);
	const uint32_t range = blockIdx.x / K1_WRITE_SPLIT, part = blockIdx.x % K1_WRITE_SPLIT;
	uint32_t li_begin, li_end; k1_slot_range(ray_end - ray_begin, range, gridDim.x / K1_WRITE_SPLIT, li_begin, li_end);
	const uint32_t n_slots = li_end - li_begin; // <= K1_MAX_RANGE
	if (threadIdx.x == 0) s_cut = 0xffffffffu;
	const uint64_t range_off = partial[range]; // {first sample, first ray slot} of the range
	uint32_t my_count = 0;
	{
		uint64_t v[4] = {0ull, 0ull, 0ull, 0ull};
		if (threadIdx.x < n_slots) { my_count = rs[li_begin + threadIdx.x].count; v[0] = (uint64_t)my_count | ((uint64_t)(my_count > 0 ? 1u : 0u) << 32); }
		uint64_t tot;
		const uint64_t pre = block_excl_scan_1024(v, s_scan, tot); // (contains a barrier: s_cut is initialised behind it)
		if (threadIdx.x < n_slots) s_base[threadIdx.x] = (uint32_t)pre;
		if (threadIdx.x == 0) s_base[n_slots] = (uint32_t)tot;
		// the ray records: one thread per ray of the range (shared between the range's workgroups)
		if (my_count > 0) {
			const uint32_t base = (uint32_t)range_off + (uint32_t)pre, slot = (uint32_t)(range_off >> 32) + (uint32_t)(pre >> 32);
			const bool fits = base + my_count <= max_samples; // testbed_nerf.cu:813-815: rays past the cap are dropped
			if (!fits) atomicMin(&s_cut, (uint32_t)pre);
			if (threadIdx.x % K1_WRITE_SPLIT == part) {
				const RaySetup& r = rs[li_begin + threadIdx.x];
				a.ray_indices_out[slot] = r.ray_index;
				ngp_ray rr; rr.o[0] = r.o[0]; rr.o[1] = r.o[1]; rr.o[2] = r.o[2]; rr.d[0] = r.d[0]; rr.d[1] = r.d[1]; rr.d[2] = r.d[2];
				a.rays_out[slot] = rr;
				a.numsteps_out[slot * 2 + 0] = fits ? my_count : 0u;
				a.numsteps_out[slot * 2 + 1] = base;
				if (a.k2_tiles0_out) { // lazy K2, round 0: the first tile of this ray (dropped rays: an empty tile)
					const uint32_t c0 = fits ? my_count : 0u;
					a.k2_tiles0_out[slot] = make_uint4(base, min(c0, a.k2_tile_w), slot, c0 - min(c0, a.k2_tile_w));
				}
				if (a.ray_targets_out) {
					float4* tg = (float4*)(a.ray_targets_out + (size_t)slot * 8);
					tg[0] = make_float4(r.tgt[0], r.tgt[1], r.tgt[2], r.tgt[3]); tg[1] = make_float4(r.tgt[4], r.tgt[5], r.tgt[6], 0.f);
				}
			}
		}
		__syncthreads();
	}
	// samples past the cap's cut are not written (dropped rays form a suffix of the slot order: the span base is monotone)
	const uint32_t n_total = min(s_base[n_slots], s_cut);
	for (uint32_t e0 = (part * 4u + wid) * 64u; e0 < n_total; e0 += K1_WRITE_SPLIT * 4u * 64u) {
		const uint32_t e = min(e0 + lane, n_total - 1u), n = min(n_total - e0, 64u);
		// the sample's slot: the last one whose span starts at or before e (empty slots share their successor's offset and lose against it)
		uint32_t lo = 0u, hi = n_slots;
		while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (s_base[mid] <= e) lo = mid; else hi = mid; }
		const RaySetup& r = rs[li_begin + lo];
		const uint32_t j = (uint32_t)jlist[(size_t)(li_begin + lo) * N_STEPS + (e - s_base[lo])];
		const f3 ro = ld3(r.o), rdn = ld3(r.rdn);
		const float t = j == 0u ? r.startt : from_stepping_space(r.nprime + (float)j, a.cone_angle_constant); // lattice_t
		const f3 pos = ro + t * rdn;
		const float dt = calc_dt(t, a.cone_angle_constant);
		const f3 wp = warp_position(pos, aabb), wd = warp_direction(rdn);
		if (a.n_extra) { // records of 7 + n_extra floats (the ray's image's extra dims behind the NerfCoordinate): stored by their own lanes
			if (lane < n) {
				const uint32_t cs = 7u + a.n_extra;
				float* c = a.coords_out + ((size_t)(uint32_t)range_off + e0 + lane) * cs;
				c[0] = wp.x; c[1] = wp.y; c[2] = wp.z; c[3] = warp_dt(dt); c[4] = wd.x; c[5] = wd.y; c[6] = wd.z;
				for (uint32_t x = 7; x < cs; ++x) c[x] = a.extra_dims[(size_t)r.img * a.n_extra + (x - 7)];
			}
			continue;
		}
		// the lane's 28-byte record as one 16-byte and one 12-byte store (the wavefront's 1792 bytes are contiguous: the L2 merges the pieces).  Rounds 4-5 transposed the
		// records through LDS for seven coalesced dword stores: measured equal (tools/batches/r06_p.sh), removed.
		if (lane < n) {
			float* co = a.coords_out + ((size_t)(uint32_t)range_off + e0 + lane) * 7;
			const f4u_t va = {wp.x, wp.y, wp.z, warp_dt(dt)}; const f3u_t vb = {wd.x, wd.y, wd.z};
			*(f4u_t*)co = va; *(f3u_t*)(co + 4) = vb;
		}
	}
}

// x-major copy of the Morton-ordered bitfield (all cascades): one thread per output byte = 8 x-consecutive cells (x0 a multiple of 8).  Their Morton indices are
// m0 + {0, 1, 8, 9, 64, 65, 72, 73} with m0 = morton(x0, y, z) (x's bits land on Morton bits 0, 3, 6; m0's own bits 0, 3, 6 are clear): two bit pairs in the 16-bit
// word at byte m0 / 8 (even: Morton bit 3 is clear) and two in the word eight bytes further -- two loads and four shifts instead of eight Morton codes and eight byte
// loads (round 5: 29.3 -> 9.4 us per update, profiles/r04_final_kernel_trace_summary_nooverlap.txt vs r05_final_kernel_trace_summary_nooverlap.txt).
__global__ void k_build_linear_bitfield(const uint8_t* __restrict__ bitfield, uint8_t* __restrict__ linear, uint32_t n_bytes) {
	const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n_bytes) return;
	const uint32_t casc = b / (GRID_N_CELLS / 8), lb = b % (GRID_N_CELLS / 8);
	const uint32_t cell0 = lb * 8, x0 = cell0 % GRIDSIZE, y = (cell0 / GRIDSIZE) % GRIDSIZE, z = cell0 / (GRIDSIZE * GRIDSIZE);
	const uint8_t* src = bitfield + grid_mip_offset(casc) / 8;
	const uint32_t m0 = morton3D(x0, y, z), sh = m0 & 7u; // sh in {0, 2, 4, 6}
	const uint32_t w0 = *(const uint16_t*)(src + (m0 >> 3)), w1 = *(const uint16_t*)(src + (m0 >> 3) + 8);
	linear[b] = (uint8_t)(((w0 >> sh) & 3u) | (((w0 >> (sh + 8u)) & 3u) << 2) | (((w1 >> sh) & 3u) << 4) | (((w1 >> (sh + 8u)) & 3u) << 6));
}

// one bit per 4x4x4 block of cells of the x-major copy: one thread per coarse cell, 32 coarse cells (one word) per 32 threads
__global__ void __launch_bounds__(256) k_build_coarse_bitfield(const uint8_t* __restrict__ linear, uint32_t* __restrict__ coarse, uint32_t n_cascades) {
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t per = COARSE_SIZE * COARSE_SIZE * COARSE_SIZE;
	const uint32_t casc = t / per, c = t % per;
	bool any = false;
	if (casc < n_cascades) {
		const uint32_t cx = c % COARSE_SIZE, cy = (c / COARSE_SIZE) % COARSE_SIZE, cz = c / (COARSE_SIZE * COARSE_SIZE);
		const uint8_t* src = linear + (size_t)casc * (GRID_N_CELLS / 8);
		for (uint32_t dz = 0; dz < 4; ++dz)
			for (uint32_t dy = 0; dy < 4; ++dy) {
				const uint32_t cell0 = 4 * cx + GRIDSIZE * ((4 * cy + dy) + GRIDSIZE * (4 * cz + dz)); // 4 x-consecutive cells: one nibble
				any |= ((src[cell0 >> 3] >> (cell0 & 7u)) & 0xfu) != 0u;
			}
	}
	const uint64_t m = __ballot(any);
	if (casc < n_cascades && (threadIdx.x & 31u) == 0) coarse[t >> 5] = (uint32_t)(m >> (threadIdx.x & 32u));
}
// "Mid" grid of cascade 0: one bit per 2x2x2 cells, DILATED by one mid cell in every direction (OR over the 6x6x6 fine cells around the mid cell).  A clear bit proves that
// every fine cell within one mid-cell edge (2 fine cells = 1/64) of ANY position inside that mid cell is empty -- k1_count_segments rejects whole 8-point lattice segments
// (7 steps = 0.012 long, i.e. every point within 0.006 of the midpoint in every coordinate) by one test of their midpoint.  One thread per mid cell.
__global__ void __launch_bounds__(256) k_build_mid_dilated_bitfield(const uint8_t* __restrict__ linear, uint32_t* __restrict__ mid) {
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; // grid = MID_SIZE^3 exactly
	const int cx = (int)(t % MID_SIZE), cy = (int)((t / MID_SIZE) % MID_SIZE), cz = (int)(t / (MID_SIZE * MID_SIZE));
	const int x0 = max(2 * cx - 2, 0), x1 = min(2 * cx + 3, (int)GRIDSIZE - 1); // inclusive range of fine x: at most 6 bits inside one 16-bit window
	bool any = false;
	for (int z = max(2 * cz - 2, 0); z <= min(2 * cz + 3, (int)GRIDSIZE - 1); ++z)
		for (int y = max(2 * cy - 2, 0); y <= min(2 * cy + 3, (int)GRIDSIZE - 1); ++y) {
			const uint32_t row = GRIDSIZE * ((uint32_t)y + GRIDSIZE * (uint32_t)z); // multiple of 8: the row's bits start at a byte
			const uint32_t b0 = (row + (uint32_t)x0) >> 3, b1 = (row + (uint32_t)x1) >> 3;
			const uint32_t w = (uint32_t)linear[b0] | ((uint32_t)linear[b1] << (8u * (b1 - b0)));
			any |= ((w >> ((uint32_t)x0 & 7u)) & ((1u << (uint32_t)(x1 - x0 + 1)) - 1u)) != 0u;
		}
	const uint64_t m = __ballot(any);
	if ((threadIdx.x & 31u) == 0) mid[t >> 5] = (uint32_t)(m >> (threadIdx.x & 32u));
}

// ------------------------------------------------------------------------------------------------
// K3
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(128) k_compute_loss(K3Args a) {
	const uint32_t cs = a.cstride;
	const uint32_t n_rays = a.n_rays_ptr ? *a.n_rays_ptr : a.n_rays;
	const uint32_t n_active = *a.rays_counter;
	if (blockIdx.x * blockDim.x >= n_active) return;
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	const bool active = i < n_active;
	const Box aabb(a.aabb);

	uint32_t numsteps = 0, base = 0, compacted_numsteps = 0;
	const float* cin = nullptr;
	const __half* no = nullptr;
	float T = 1.f;
	f3 rgb_ray = mk3(0.f), ray_o = mk3(0.f), loss_bg = mk3(0.f);
	f3 rgbtarget = mk3(0.f), background_color = ld3(a.background_color);
	float depth_ray = 0.f, target_depth = -1.f;
	uint32_t pix_img = 0; float pix_pdf = 1.0f; f2 pix_uv = {0.f, 0.f};
	if (active) {
		numsteps = a.numsteps_inout[i * 2 + 0];
		base = a.numsteps_inout[i * 2 + 1];
		cin = a.coords_in + (size_t)base * cs;
		no = (const __half*)a.network_output + (size_t)base * a.output_stride;
		ray_o = ld3(a.rays_in[i].o);
		const uint32_t ray_idx = a.ray_indices_in[i];
		Rng rng(a.rng);
		rng.advance((uint64_t)(ray_idx * N_RANDOM_PER_RAY));
		uint32_t img;
		const f2 uv = training_pixel(a.cdf, rng, ray_idx, n_rays, a.n_images, a.metadata, a.snap_to_pixel_centers, img, pix_pdf);
		pix_img = img; pix_uv = uv;
		const ngp_image_meta& m = a.metadata[img];
		rng.advance(1); // motionblur_time
		if (a.random_bg_color) { background_color.x = rng.next_float(); background_color.y = rng.next_float(); background_color.z = rng.next_float(); }
		background_color = srgb_to_linear3(background_color);
		const f4 tex = read_rgba(uv, m.resolution, m.pixels, m.image_data_type);
		const f3 trgb = mk3(tex.x, tex.y, tex.z);
		if (a.linear_colors || !a.color_space_srgb) {
			rgbtarget = trgb + (1.0f - tex.w) * background_color;
			if (!a.linear_colors) { rgbtarget = linear_to_srgb3(rgbtarget); background_color = linear_to_srgb3(background_color); }
		} else {
			background_color = linear_to_srgb3(background_color);
			if (tex.w > 0) rgbtarget = linear_to_srgb3(trgb / tex.w) * tex.w + (1.0f - tex.w) * background_color;
			else rgbtarget = background_color;
		}
		target_depth = len3(ld3(a.rays_in[i].d)) * ((a.depth_lambda > 0.0f && m.depth) ? read_depth(uv, m.resolution, m.depth) : -1.0f); // testbed_nerf.cu:1027
		const float EPSILON = 1e-4f;
		for (; compacted_numsteps < numsteps; ++compacted_numsteps) {
			if (T < EPSILON) break;
			const __half* lo = no + (size_t)compacted_numsteps * a.output_stride;
			const f3 rgb = mk3(act_rgb(__half2float(lo[0]), a.rgb_activation), act_rgb(__half2float(lo[1]), a.rgb_activation), act_rgb(__half2float(lo[2]), a.rgb_activation));
			const float dt = unwarp_dt(cin[(size_t)compacted_numsteps * cs + 3]);
			const float density = act_density(__half2float(lo[3]), a.density_activation);
			const float alpha = 1.f - __expf(-density * dt);
			const float weight = alpha * T;
			rgb_ray = rgb_ray + weight * rgb;
			if (a.depth_lambda > 0.0f) { const float* ck = cin + (size_t)compacted_numsteps * cs; depth_ray += weight * dist3(unwarp_position(mk3(ck[0], ck[1], ck[2]), aabb), ray_o); }
			if (a.train_mode == 1) { f3 ll, lg2; loss_and_gradient(rgbtarget, rgb, a.loss_type, ll, lg2); loss_bg = loss_bg + weight * ll; }
			T *= (1.f - alpha);
		}
		if (compacted_numsteps == numsteps) {
			rgb_ray = rgb_ray + T * background_color;
			if (a.train_mode == 1) { f3 ll, lg2; loss_and_gradient(rgbtarget, background_color, a.loss_type, ll, lg2); loss_bg = loss_bg + T * ll; }
		}
	}

	// span reservation in the compacted batch: one atomic per wavefront
	uint32_t wave_total;
	uint32_t offset = wave_excl_scan(active ? compacted_numsteps : 0u, wave_total);
	uint32_t wave_base = 0;
	if ((threadIdx.x & 63u) == 0 && wave_total) wave_base = atomicAdd(a.numsteps_counter_compacted, wave_total);
	wave_base = __shfl(wave_base, 0, 64);
	const uint32_t compacted_base = wave_base + offset;
	float my_loss = 0.f;
	if (active) {
		compacted_numsteps = min(a.max_samples_compacted - min(a.max_samples_compacted, compacted_base), compacted_numsteps);
		a.numsteps_inout[i * 2 + 0] = compacted_numsteps;
		a.numsteps_inout[i * 2 + 1] = compacted_base;
	}
	if (active && compacted_numsteps > 0) {
		float* cout = a.coords_out + (size_t)compacted_base * cs;
		__half* dl = (__half*)a.dloss_doutput + (size_t)compacted_base * a.dloss_stride;
		f3 lloss, lgrad;
		loss_and_gradient(rgbtarget, rgb_ray, a.loss_type, lloss, lgrad);
		if (a.cdf.x_cond_y || a.cdf.img) lloss = lloss / pix_pdf; // testbed_nerf.cu:1024 (the gradient is deliberately NOT divided, :1031-1035)
		const float mean_loss = (lloss.x + lloss.y + lloss.z) / 3.0f;
		my_loss = mean_loss / (float)n_rays;
		if (a.error_map) deposit_error(a.error_map, a.error_map_res, a.metadata[pix_img].resolution, pix_img, pix_uv, mean_loss);
		const float loss_scale = a.loss_scale / n_rays;
		const float output_l2_reg = a.rgb_activation == NGP_ACT_EXPONENTIAL ? 1e-4f : 0.0f;
		const float output_l1_reg_density = (a.train_mode == 0 && *a.mean_density_ptr < MIN_OPTICAL_THICKNESS) ? 1e-4f : 0.0f;
		float depth_loss_gradient = 0.f, depth_ray2 = 0.f;
		if (target_depth > 0.0f) { f3 dl_, dg_; loss_and_gradient(mk3(target_depth), mk3(depth_ray), a.depth_loss_type, dl_, dg_); depth_loss_gradient = a.depth_lambda * dg_.x; } // testbed_nerf.cu:1028-1029
		f3 rgb_ray2 = mk3(0.f), loss_bg2 = mk3(0.f);
		T = 1.f;
		for (uint32_t j = 0; j < compacted_numsteps; ++j) {
			const float* ci = cin + (size_t)j * cs;
			float* cj = cout + (size_t)j * cs;
			const float c0 = ci[0], c1 = ci[1], c2 = ci[2], c3 = ci[3];
			cj[0] = c0; cj[1] = c1; cj[2] = c2; cj[3] = c3; cj[4] = ci[4]; cj[5] = ci[5]; cj[6] = ci[6];
			for (uint32_t k = 7; k < cs; ++k) cj[k] = ci[k];
			if (a.src_index_out) a.src_index_out[compacted_base + j] = base + j;
			const f3 pos = unwarp_position(mk3(c0, c1, c2), aabb);
			const float depth = dist3(pos, ray_o);
			const float dt = unwarp_dt(c3);
			const __half* lo = no + (size_t)j * a.output_stride;
			const float l0 = __half2float(lo[0]), l1 = __half2float(lo[1]), l2 = __half2float(lo[2]), l3 = __half2float(lo[3]);
			const f3 rgb = mk3(act_rgb(l0, a.rgb_activation), act_rgb(l1, a.rgb_activation), act_rgb(l2, a.rgb_activation));
			const float density = act_density(l3, a.density_activation);
			const float alpha = 1.f - __expf(-density * dt);
			const float weight = alpha * T;
			rgb_ray2 = rgb_ray2 + weight * rgb;
			depth_ray2 += weight * depth;
			T *= (1.f - alpha);
			const f3 suffix = rgb_ray - rgb_ray2;
			f3 dloss_by_drgb = weight * lgrad;
			const float density_derivative = act_density_d(l3, a.density_activation);
			const float depth_supervision = depth_loss_gradient * (T * depth - (depth_ray - depth_ray2)); // testbed_nerf.cu:1126-1127
			float dloss_by_dmlp = density_derivative * (dt * (dot3(lgrad, T * rgb - suffix) + depth_supervision));
			if (a.train_mode == 1) { // Rfl, fused_kernels/train_nerf.cuh:391-396
				f3 ll, lgl; loss_and_gradient(rgbtarget, rgb, a.loss_type, ll, lgl);
				loss_bg2 = loss_bg2 + weight * ll;
				dloss_by_drgb = weight * lgl;
				const f3 v = T * ll - (loss_bg - loss_bg2);
				dloss_by_dmlp = density_derivative * (dt * (v.x + v.y + v.z));
			} else if (a.train_mode == 2) { // RflRelax, train_nerf.cuh:397-405
				const f3 rgb_bg = suffix / fmaxf(1e-6f, T);
				const f3 rgb_lerp = (1 - alpha) * rgb_bg + alpha * rgb;
				f3 ll, lgl; loss_and_gradient(rgbtarget, rgb_lerp, a.loss_type, ll, lgl);
				dloss_by_drgb = weight * lgl;
				dloss_by_dmlp = density_derivative * (dt * (dot3(lgl, T * rgb - suffix) + 0.0f));
			}
			const float d0 = loss_scale * (dloss_by_drgb.x * act_rgb_d(l0, a.rgb_activation) + fmaxf(0.0f, output_l2_reg * l0));
			const float d1 = loss_scale * (dloss_by_drgb.y * act_rgb_d(l1, a.rgb_activation) + fmaxf(0.0f, output_l2_reg * l1));
			const float d2 = loss_scale * (dloss_by_drgb.z * act_rgb_d(l2, a.rgb_activation) + fmaxf(0.0f, output_l2_reg * l2));
			const float d3 = loss_scale * dloss_by_dmlp + (l3 < 0.0f ? -output_l1_reg_density : 0.0f) + (l3 > -10.0f && depth < a.near_distance ? 1e-4f : 0.0f);
			__half* d = dl + (size_t)j * a.dloss_stride;
			d[0] = __float2half(d0); d[1] = __float2half(d1); d[2] = __float2half(d2); d[3] = __float2half(d3);
		}
	}
	if (a.loss_output) {
		float s = wave_sum(my_loss);
		if ((threadIdx.x & 63u) == 0 && s != 0.f) atomicAdd(a.loss_output, s);
	}
}


// ------------------------------------------------------------------------------------------------
// K3, MI355X version: one wavefront per ray, lanes = samples.
// Transmittance is an exclusive prefix product over the lanes (wave scan), the composited colour a
// wave reduction, the adjoint's "suffix" an inclusive prefix sum -- so the three sequential per-ray
// loops of the reference become O(log 64) scans per 64 samples and every ray gets 64 lanes instead of
// one.  16 rays (waves) share a workgroup and reserve their compacted spans with ONE global atomic.
// Same arithmetic per sample as k_compute_loss; only the association order of the sums/products
// differs (prefix scans), i.e. results agree to fp32 round-off.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t K3_RAYS_PER_BLOCK = 16;
typedef _Float16 h4v __attribute__((ext_vector_type(4)));

// Wave-wide inclusive scans with DPP (row_shr within the 16-lane rows, then row_bcast:15 / row_bcast:31 across rows): 7 data-parallel
// moves + 7 operations per scan instead of 6 ds_bpermute round trips.  `ident` fills the lanes that have no source.
#define NGP_DPP(old_f, src_f, ctrl, row_mask) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old_f), __builtin_bit_cast(int, src_f), ctrl, row_mask, 0xf, false))
template <bool PROD>
static __device__ __forceinline__ float wave_incl_scan(float x) {
	const float ident = PROD ? 1.f : 0.f;
	auto op = [](float a, float b) { return PROD ? a * b : a + b; };
	float a = op(x, NGP_DPP(ident, x, 0x111, 0xf)); // row_shr:1
	a = op(a, NGP_DPP(ident, x, 0x112, 0xf));       // row_shr:2 (of x)
	a = op(a, NGP_DPP(ident, x, 0x113, 0xf));       // row_shr:3 (of x)
	a = op(a, NGP_DPP(ident, a, 0x114, 0xf));       // row_shr:4
	a = op(a, NGP_DPP(ident, a, 0x118, 0xf));       // row_shr:8  -> inclusive scan of every row
	a = op(a, NGP_DPP(ident, a, 0x142, 0xa));       // row_bcast:15 into rows 1 and 3
	a = op(a, NGP_DPP(ident, a, 0x143, 0xc));       // row_bcast:31 into rows 2 and 3
	return a;
}
static __device__ __forceinline__ float wave_incl_prod(float x, uint32_t) { return wave_incl_scan<true>(x); }
static __device__ __forceinline__ float wave_incl_sum(float x, uint32_t) { return wave_incl_scan<false>(x); }
static __device__ __forceinline__ float wave_total(float x) { return __shfl(wave_incl_scan<false>(x), 63, 64); } // sum of all 64 lanes, in every lane

// Segmented variants of the wave scans for the two-rays-per-wavefront kernel: inclusive scan inside each 32-lane half (the DPP row operations stay inside
// 16-lane rows, row_bcast:15 carries row 0 -> 1 and row 2 -> 3; without row_bcast:31 nothing crosses the halves).
template <bool PROD>
static __device__ __forceinline__ float half_incl_scan(float x) {
	const float ident = PROD ? 1.f : 0.f;
	auto op = [](float a, float b) { return PROD ? a * b : a + b; };
	float a = op(x, NGP_DPP(ident, x, 0x111, 0xf));
	a = op(a, NGP_DPP(ident, x, 0x112, 0xf));
	a = op(a, NGP_DPP(ident, x, 0x113, 0xf));
	a = op(a, NGP_DPP(ident, a, 0x114, 0xf));
	a = op(a, NGP_DPP(ident, a, 0x118, 0xf));
	a = op(a, NGP_DPP(ident, a, 0x142, 0xa)); // row_bcast:15 into rows 1 and 3
	return a;
}

// RPW = rays per wavefront.  1: one wavefront per ray (64 samples per pass iteration).  2 (production): the two 32-lane halves of a wavefront work on two rays --
// a trained scene keeps ~10 samples per ray, so a 64-lane wavefront per ray wastes 5/6 of every instruction (the kernel is VALU-issue bound);
// the per-ray values move from scalar to per-lane registers, the scans become segmented, and both halves iterate until the longer ray is done.
// (Four rays per wavefront -- 16-lane DPP rows -- were measured in round 4: 0.0632 / 0.0646 / 0.0651 ms against 0.0615 / 0.0631 / 0.0629 ms for two, interleaved on one box,
// profiles/r04_microbench_k3_four_rays_per_wave_no_gain.log: below 32 lanes per ray the kernel is bound by its per-ray dependent loads and the workgroup's span atomic, not
// by instruction issue.  Not kept.)
// ERR: error-proportional pixel sampling / error-map accumulation compiled in (off in the production instance: the extra live values cost it 16 bytes of scratch).
// PLAIN: train_mode Nerf and no depth supervision, as compile-time facts (the production instance): the Rfl / depth accumulators and their scans disappear.
// TGT (round 5): the rays' targets come from k1_setup (K3Args::ray_targets; the trainer's lattice path) -- the instance does not carry the target-pixel chain at all
// (pixel sampler, three pixel formats, six sRGB conversions: a fifth of the PLAIN instance's code).
// WPB = wavefronts per workgroup, MINW = wavefronts per SIMD the register allocation must allow (round 6): a workgroup's wavefronts all wait at the same two barriers and on
// the same returning span atomic, so ONE 16-wavefront workgroup per CU (what 96 registers allow) leaves the CU idle during every such wait; smaller workgroups take turns.
template <int RPW, bool ERR, bool PLAIN, bool TGT = false, int WPB = 16, int MINW = 4>
__global__ void __launch_bounds__(64 * WPB, MINW) k_compute_loss_v2(K3Args a) {
	const uint32_t cs = PLAIN ? 7u : a.cstride; // (the production instance keeps the NerfCoordinate's 7 floats as a compile-time stride: models with extra dims run the generic one)
	const int train_mode = PLAIN ? 0 : a.train_mode;
	const float depth_lambda = PLAIN ? 0.f : a.depth_lambda;
	constexpr uint32_t LPR = 64u / RPW, RPB = (uint32_t)WPB * RPW; // lanes per ray, rays per workgroup
	__shared__ uint32_t s_cnt[RPB];
	__shared__ float s_loss[RPB];
	__shared__ uint32_t s_base;
	const uint32_t n_rays = a.n_rays_ptr ? *a.n_rays_ptr : a.n_rays;
	const uint32_t n_active = *a.rays_counter;
	const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
	const uint32_t sub = lane / LPR, sl = lane % LPR, seg0 = sub * LPR; // which ray of the wavefront, lane inside the ray's segment, the segment's first lane
	auto uni = [&](uint32_t v) { return RPW == 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)v) : v; }; // per-ray values: scalar when the wavefront has one ray
	static_assert(RPW == 1 || RPW == 2, "rays per wavefront");
	auto seg_prod = [&](float x) { return RPW == 1 ? wave_incl_scan<true>(x) : half_incl_scan<true>(x); };
	auto seg_sum = [&](float x) { return RPW == 1 ? wave_incl_scan<false>(x) : half_incl_scan<false>(x); };
	auto seg_total = [&](float x) { return __shfl(seg_sum(x), (int)(seg0 + LPR - 1u), 64); };       // sum over the ray's segment, in every lane of it
	constexpr uint64_t SEG_BITS = LPR >= 64u ? ~0ull : ((1ull << (LPR & 63u)) - 1ull);
	auto seg_mask = [&](bool pred) { const uint64_t m = __ballot(pred); return RPW == 1 ? m : (m >> seg0) & SEG_BITS; }; // ballot restricted to the segment
	const Box aabb(a.aabb);
	const float EPSILON = 1e-4f;
	float block_loss = 0.f; // thread 0 only
	// persistent grid (see k1_count): each workgroup loops over groups of RPB rays
	for (uint32_t grp = blockIdx.x; grp * RPB < n_active; grp += gridDim.x) {
	const uint32_t i = grp * RPB + wid * RPW + sub; // the lane's ray (wave-uniform for RPW == 1)
	const bool active = i < n_active;

	uint32_t numsteps = 0, base = 0, compacted = 0;
	const float* cin = nullptr;
	const __half* no = nullptr;
	f3 rgb_ray = mk3(0.f), ray_o = mk3(0.f), rgbtarget = mk3(0.f), background_color = ld3(a.background_color), loss_bg = mk3(0.f);
	float T_final = 1.f;
	float depth_ray = 0.f, target_depth = -1.f; // depth supervision (depth_lambda > 0, wave-uniform)
	// the first LPR samples of the ray stay in registers for the adjoint pass (most rays keep fewer)
	// (the NerfCoordinate as two VECTOR values, never an array: the 7-float array of rounds 3-5 lived in scratch -- a store in pass 1, a load in pass 2, 32 bytes per lane)
	float k_l0 = 0.f, k_l1 = 0.f, k_l2 = 0.f, k_l3 = 0.f; f4u_t k_ca = {0.f, 0.f, 0.f, 0.f}; f3u_t k_cb = {0.f, 0.f, 0.f};
	const bool vec_out = a.output_stride == 4, vec_dl = a.dloss_stride == 4;
	auto load_out = [&](const __half* lo, float& l0, float& l1, float& l2, float& l3) {
		if (vec_out) {
			const uint2 raw = *(const uint2*)lo;
			const h4v v = __builtin_bit_cast(h4v, raw);
			l0 = (float)v[0]; l1 = (float)v[1]; l2 = (float)v[2]; l3 = (float)v[3];
		} else { l0 = __half2float(lo[0]); l1 = __half2float(lo[1]); l2 = __half2float(lo[2]); l3 = __half2float(lo[3]); }
	};
	if (active) {
		// per-ray values: uniform across the ray's lanes (scalar registers when the wavefront has one ray)
		numsteps = uni(a.numsteps_inout[i * 2 + 0]);
		base = uni(a.numsteps_inout[i * 2 + 1]);
		ray_o = ld3(a.rays_in[i].o);
		f4 tex = {0.f, 0.f, 0.f, 0.f};
		if (TGT || a.ray_targets) { // computed once per ray by k1_setup
			const float4 t0 = ((const float4*)(a.ray_targets + (size_t)i * 8))[0], t1 = ((const float4*)(a.ray_targets + (size_t)i * 8))[1];
			rgbtarget = mk3(t0.x, t0.y, t0.z); background_color = mk3(t0.w, t1.x, t1.y); target_depth = PLAIN ? -1.f : t1.z;
		} else {
			const uint32_t ray_idx = uni(a.ray_indices_in[i]);
			// The target-pixel chain (ray index -> image metadata -> texel) is issued BEFORE the sample pass so that its three
			// dependent memory latencies overlap the sample loads instead of following them (uniform across the ray's lanes).
			Rng rng(a.rng);
			rng.advance((uint64_t)(ray_idx * N_RANDOM_PER_RAY));
			uint32_t img; float pdf_unused;
			const f2 uv = training_pixel(ERR ? a.cdf : ErrorCdf{}, rng, ray_idx, n_rays, a.n_images, a.metadata, a.snap_to_pixel_centers, img, pdf_unused);
			const ngp_image_meta& m = a.metadata[img];
			rng.advance(1); // motionblur_time
			if (a.random_bg_color) { background_color.x = rng.next_float(); background_color.y = rng.next_float(); background_color.z = rng.next_float(); }
			tex = read_rgba(uv, m.resolution, m.pixels, m.image_data_type);
			target_depth = len3(ld3(a.rays_in[i].d)) * ((depth_lambda > 0.0f && m.depth) ? read_depth(uv, m.resolution, m.depth) : -1.0f);
			// target colour and background: identical to the sequential kernel; needed BEFORE the sample pass by the Rfl mode
			background_color = srgb_to_linear3(background_color);
			const f3 trgb = mk3(tex.x, tex.y, tex.z);
			if (a.linear_colors || !a.color_space_srgb) {
				rgbtarget = trgb + (1.0f - tex.w) * background_color;
				if (!a.linear_colors) { rgbtarget = linear_to_srgb3(rgbtarget); background_color = linear_to_srgb3(background_color); }
			} else {
				background_color = linear_to_srgb3(background_color);
				if (tex.w > 0) rgbtarget = linear_to_srgb3(trgb / tex.w) * tex.w + (1.0f - tex.w) * background_color;
				else rgbtarget = background_color;
			}
		}
		cin = a.coords_in + (size_t)base * cs;
		no = (const __half*)a.network_output + (size_t)base * a.output_stride;
	}
	{ // ---- pass 1: composite front to back until the transmittance cut; every lane of the wavefront takes part in the scans ----
		float T_run = 1.f;
		bool fin = !active; // this ray's pass is over (cut found / all samples seen / no ray)
		for (uint32_t c0 = 0; ; c0 += LPR) {
			const bool on = !fin && c0 < numsteps;
			if (__ballot(on) == 0ull) break;
			const uint32_t s = c0 + sl;
			const bool valid = on && s < numsteps;
			float alpha = 0.f, sdepth = 0.f; f3 rgb = mk3(0.f);
			if (valid) {
				float l0, l1, l2, l3, dtw;
				load_out(no + (size_t)s * a.output_stride, l0, l1, l2, l3);
				if (c0 == 0) {
					const float* ci = cin + (size_t)s * cs;
					k_ca = *(const f4u_t*)ci; k_cb = *(const f3u_t*)(ci + 4);
					k_l0 = l0; k_l1 = l1; k_l2 = l2; k_l3 = l3;
					dtw = k_ca[3];
				} else dtw = cin[(size_t)s * cs + 3];
				rgb = mk3(act_rgb(l0, a.rgb_activation), act_rgb(l1, a.rgb_activation), act_rgb(l2, a.rgb_activation));
				const float dt = unwarp_dt(dtw);
				alpha = 1.f - __expf(-act_density(l3, a.density_activation) * dt);
				if (depth_lambda > 0.0f) { const float* ci = cin + (size_t)s * cs; sdepth = dist3(unwarp_position(mk3(ci[0], ci[1], ci[2]), aabb), ray_o); }
			}
			const float incl = seg_prod(1.f - alpha);
			float excl = __shfl_up(incl, 1, 64);
			if (sl == 0) excl = 1.f;
			const float T_k = T_run * excl;
			const uint64_t vm = seg_mask(valid), fail = seg_mask(valid && !(T_k >= EPSILON)); // `if (T < EPSILON) break;`
			const uint32_t n_proc = fail ? (uint32_t)(__ffsll((long long)fail) - 1) : (uint32_t)__popcll(vm);
			const bool proc = sl < n_proc;
			const float w = proc ? alpha * T_k : 0.f;
			// lanes behind the cut may hold unevaluated network outputs (lazy K2): select, never multiply
			rgb_ray = rgb_ray + mk3(seg_total(proc ? w * rgb.x : 0.f), seg_total(proc ? w * rgb.y : 0.f), seg_total(proc ? w * rgb.z : 0.f));
			if (depth_lambda > 0.0f) depth_ray += seg_total(proc ? w * sdepth : 0.f);
			if (train_mode == 1) { // Rfl: sum of weight * per-sample loss (train_nerf.cuh:219)
				f3 ll, lgl; loss_and_gradient(rgbtarget, rgb, a.loss_type, ll, lgl);
				loss_bg = loss_bg + mk3(seg_total(proc ? w * ll.x : 0.f), seg_total(proc ? w * ll.y : 0.f), seg_total(proc ? w * ll.z : 0.f));
			}
			const float T_last = __shfl(incl, (int)(seg0 + (n_proc ? n_proc - 1u : 0u)), 64);
			if (on) {
				if (n_proc) T_run = T_run * T_last;
				compacted += n_proc;
				if (fail) fin = true;
			}
		}
		T_final = T_run;
		if (active && compacted == numsteps) {
			rgb_ray = rgb_ray + T_final * background_color;
			if (train_mode == 1) { f3 ll, lgl; loss_and_gradient(rgbtarget, background_color, a.loss_type, ll, lgl); loss_bg = loss_bg + T_final * ll; }
		}
	}
	// one global atomic per workgroup reserves the spans of its rays
	if (sl == 0) s_cnt[wid * RPW + sub] = active ? compacted : 0u;
	__syncthreads();
	if (threadIdx.x == 0) { // exclusive prefix of the workgroup's counts (left in s_cnt) + its total
		uint32_t tot = 0;
		for (uint32_t w = 0; w < RPB; ++w) { const uint32_t cw = s_cnt[w]; s_cnt[w] = tot; tot += cw; }
		s_base = tot ? atomicAdd(a.numsteps_counter_compacted, tot) : 0u;
	}
	__syncthreads();
	const uint32_t compacted_base = s_base + s_cnt[wid * RPW + sub];
	float my_loss = 0.f;
	if (active) {
		compacted = min(a.max_samples_compacted - min(a.max_samples_compacted, compacted_base), compacted);
		if (sl == 0) { a.numsteps_inout[i * 2 + 0] = compacted; a.numsteps_inout[i * 2 + 1] = compacted_base; }
	} else compacted = 0;
	{ // ---- pass 2: adjoint + compaction ----
		float* cout = a.coords_out + (size_t)compacted_base * cs;
		__half* dl = (__half*)a.dloss_doutput + (size_t)compacted_base * a.dloss_stride;
		f3 lloss = mk3(0.f), lgrad = mk3(0.f);
		loss_and_gradient(rgbtarget, rgb_ray, a.loss_type, lloss, lgrad);
		if (ERR && (a.error_map || a.cdf.x_cond_y || a.cdf.img) && compacted > 0) {
			// error-proportional sampling (off by default): the ray's pixel is re-derived from its index, as the reference's K3 does (testbed_nerf.cu:955-961),
			// the loss is divided by the density it was drawn with (:1024) and splatted into the error map (:1042-1071)
			Rng rng(a.rng);
			const uint32_t ray_idx = a.ray_indices_in[i];
			rng.advance((uint64_t)(ray_idx * N_RANDOM_PER_RAY));
			uint32_t img; float pdf;
			const f2 uv = training_pixel(a.cdf, rng, ray_idx, n_rays, a.n_images, a.metadata, a.snap_to_pixel_centers, img, pdf);
			if (a.cdf.x_cond_y || a.cdf.img) lloss = lloss / pdf;
			if (a.error_map && sl == 0) deposit_error(a.error_map, a.error_map_res, a.metadata[img].resolution, img, uv, (lloss.x + lloss.y + lloss.z) / 3.0f);
		}
		if (compacted > 0) my_loss = ((lloss.x + lloss.y + lloss.z) / 3.0f) / (float)n_rays;
		const float loss_scale = a.loss_scale / n_rays;
		const float output_l2_reg = a.rgb_activation == NGP_ACT_EXPONENTIAL ? 1e-4f : 0.0f;
		const float output_l1_reg_density = (train_mode == 0 && *a.mean_density_ptr < MIN_OPTICAL_THICKNESS) ? 1e-4f : 0.0f;
		float T_run = 1.f;
		f3 ray2_run = mk3(0.f), lb2_run = mk3(0.f);
		float depth_loss_gradient = 0.f, depth2_run = 0.f;
		if (target_depth > 0.0f) { f3 dl_, dg_; loss_and_gradient(mk3(target_depth), mk3(depth_ray), a.depth_loss_type, dl_, dg_); depth_loss_gradient = depth_lambda * dg_.x; } // testbed_nerf.cu:1028-1029
		for (uint32_t c0 = 0; ; c0 += LPR) {
			if (__ballot(c0 < compacted) == 0ull) break;
			const uint32_t s = c0 + sl;
			const bool valid = s < compacted;
			float alpha = 0.f, l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f, dt = 0.f, depth = 0.f;
			f3 rgb = mk3(0.f);
			f4u_t ca = {0.f, 0.f, 0.f, 0.f}; f3u_t cb = {0.f, 0.f, 0.f};
			if (valid) {
				if (c0 == 0) {
					ca = k_ca; cb = k_cb;
					l0 = k_l0; l1 = k_l1; l2 = k_l2; l3 = k_l3;
				} else {
					const float* ci = cin + (size_t)s * cs;
					ca = *(const f4u_t*)ci; cb = *(const f3u_t*)(ci + 4);
					load_out(no + (size_t)s * a.output_stride, l0, l1, l2, l3);
				}
				rgb = mk3(act_rgb(l0, a.rgb_activation), act_rgb(l1, a.rgb_activation), act_rgb(l2, a.rgb_activation));
				dt = unwarp_dt(ca[3]);
				alpha = 1.f - __expf(-act_density(l3, a.density_activation) * dt);
				depth = dist3(unwarp_position(mk3(ca[0], ca[1], ca[2]), aabb), ray_o);
			}
			const float incl = seg_prod(1.f - alpha);
			float excl = __shfl_up(incl, 1, 64);
			if (sl == 0) excl = 1.f;
			const float T_k = T_run * excl, T_after = T_run * incl;
			const float weight = alpha * T_k;
			const f3 ray2 = ray2_run + mk3(seg_sum(weight * rgb.x), seg_sum(weight * rgb.y), seg_sum(weight * rgb.z));
			float depth2 = depth2_run;
			if (depth_lambda > 0.0f) depth2 = depth2_run + seg_sum(weight * depth);
			f3 lloc = mk3(0.f), gloc = mk3(0.f), lb2 = lb2_run;
			if (train_mode == 1) { // Rfl: per-sample loss against the target and its running (inclusive) weighted sum
				loss_and_gradient(rgbtarget, rgb, a.loss_type, lloc, gloc);
				lb2 = lb2_run + mk3(seg_sum(weight * lloc.x), seg_sum(weight * lloc.y), seg_sum(weight * lloc.z));
			}
			if (valid) {
				float* cj = cout + (size_t)s * cs;
				*(f4u_t*)cj = ca; *(f3u_t*)(cj + 4) = cb;
				for (uint32_t k = 7; k < cs; ++k) cj[k] = cin[(size_t)s * cs + k];
				if (a.src_index_out) a.src_index_out[compacted_base + s] = base + s; // which K2 sample this batch row is (EncStashIn)
				const f3 suffix = rgb_ray - ray2;
				f3 dloss_by_drgb = weight * lgrad;
				float dmlp_inner = dot3(lgrad, T_after * rgb - suffix) + depth_loss_gradient * (T_after * depth - (depth_ray - depth2)); // depth supervision, testbed_nerf.cu:1126-1129
				if (train_mode == 1) { // fused_kernels/train_nerf.cuh:391-396
					dloss_by_drgb = weight * gloc;
					const f3 v = T_after * lloc - (loss_bg - lb2);
					dmlp_inner = v.x + v.y + v.z;
				} else if (train_mode == 2) { // train_nerf.cuh:397-405
					const f3 rgb_bg = suffix / fmaxf(1e-6f, T_after);
					const f3 rgb_lerp = (1 - alpha) * rgb_bg + alpha * rgb;
					f3 ll, lgl; loss_and_gradient(rgbtarget, rgb_lerp, a.loss_type, ll, lgl);
					dloss_by_drgb = weight * lgl;
					dmlp_inner = dot3(lgl, T_after * rgb - suffix) + 0.0f;
				}
				const float d0 = loss_scale * (dloss_by_drgb.x * act_rgb_d(l0, a.rgb_activation) + fmaxf(0.0f, output_l2_reg * l0));
				const float d1 = loss_scale * (dloss_by_drgb.y * act_rgb_d(l1, a.rgb_activation) + fmaxf(0.0f, output_l2_reg * l1));
				const float d2 = loss_scale * (dloss_by_drgb.z * act_rgb_d(l2, a.rgb_activation) + fmaxf(0.0f, output_l2_reg * l2));
				const float dloss_by_dmlp = act_density_d(l3, a.density_activation) * (dt * dmlp_inner);
				const float d3 = loss_scale * dloss_by_dmlp + (l3 < 0.0f ? -output_l1_reg_density : 0.0f) + (l3 > -10.0f && depth < a.near_distance ? 1e-4f : 0.0f);
				__half* d = dl + (size_t)s * a.dloss_stride;
				if (vec_dl) {
					const h4v v = {(_Float16)d0, (_Float16)d1, (_Float16)d2, (_Float16)d3};
					*(uint2*)d = __builtin_bit_cast(uint2, v);
				} else { d[0] = __float2half(d0); d[1] = __float2half(d1); d[2] = __float2half(d2); d[3] = __float2half(d3); }
			}
			const int last = (int)(seg0 + LPR - 1u);
			T_run = T_run * __shfl(incl, last, 64);
			ray2_run = mk3(__shfl(ray2.x, last, 64), __shfl(ray2.y, last, 64), __shfl(ray2.z, last, 64));
			if (depth_lambda > 0.0f) depth2_run = __shfl(depth2, last, 64);
			if (train_mode == 1) lb2_run = mk3(__shfl(lb2.x, last, 64), __shfl(lb2.y, last, 64), __shfl(lb2.z, last, 64));
		}
	}
	if (sl == 0) s_loss[wid * RPW + sub] = my_loss;
	__syncthreads(); // also protects s_cnt / s_base against the next group
	if (threadIdx.x == 0) for (uint32_t w = 0; w < RPB; ++w) block_loss += s_loss[w];
	}
	if (a.loss_output && threadIdx.x == 0 && block_loss != 0.f) atomicAdd(a.loss_output, block_loss);
}

// ------------------------------------------------------------------------------------------------
// K3 in two passes (ablation DBG_K3_TWO_PASS; the one-pass kernel above is faster: 72 us against 34 + 57 us): PASS 0 composites every active ray front to back (wave per ray) and leaves a 16-float record per
// ray {compacted count, rgb_ray, loss_bg (Rfl), target, background}; the workgroups' totals are scanned by the last workgroup (same
// scheme as k1_count).  PASS 1 places every ray's compacted samples at its slot-ordered offset and writes the adjoint.
// Against the one-pass kernel above: no __syncthreads / span atomic per 16 rays, 256-thread workgroups at <= 64 registers (twice the
// rays in flight per CU; the kernel is a chain of dependent loads per ray), and the compacted order is DETERMINISTIC (ray-slot order
// instead of the order in which workgroups win the span atomic) -- two runs from the same state produce the same batch bit for bit.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t K3_REC = 16; // floats per ray record
template <int PASS>
__global__ void __launch_bounds__(256) k_compute_loss_v3(K3Args a, float* __restrict__ rec, uint64_t* __restrict__ partial, uint32_t* __restrict__ done) {
	const uint32_t cs = a.cstride;
	__shared__ uint64_t s_tot[4];
	__shared__ uint64_t s_scan[4];
	__shared__ uint32_t s_ticket;
	__shared__ uint32_t s_off[K1_MAX_RANGE];
	__shared__ float s_loss[4];
	const uint32_t n_rays = a.n_rays_ptr ? *a.n_rays_ptr : a.n_rays;
	const uint32_t n_active = *a.rays_counter;
	const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
	const Box aabb(a.aabb);
	const float EPSILON = 1e-4f;
	const bool vec_out = a.output_stride == 4, vec_dl = a.dloss_stride == 4;
	auto load_out = [&](const __half* lo, float& l0, float& l1, float& l2, float& l3) {
		if (vec_out) {
			const uint2 raw = *(const uint2*)lo;
			const h4v v = __builtin_bit_cast(h4v, raw);
			l0 = (float)v[0]; l1 = (float)v[1]; l2 = (float)v[2]; l3 = (float)v[3];
		} else { l0 = __half2float(lo[0]); l1 = __half2float(lo[1]); l2 = __half2float(lo[2]); l3 = __half2float(lo[3]); }
	};
	const uint32_t r_begin = (uint32_t)(((uint64_t)n_active * blockIdx.x) / gridDim.x), r_end = (uint32_t)(((uint64_t)n_active * (blockIdx.x + 1)) / gridDim.x);
	if (PASS == 0) {
		uint64_t wtot = 0ull;
		for (uint32_t i = r_begin + wid; i < r_end; i += 4) {
			const uint32_t numsteps = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.numsteps_inout[i * 2 + 0]);
			const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.numsteps_inout[i * 2 + 1]);
			f3 rgb_ray = mk3(0.f), rgbtarget = mk3(0.f), background_color = ld3(a.background_color), loss_bg = mk3(0.f);
			if (a.ray_targets) { // computed once per ray by k1_setup
				const float4 t0 = ((const float4*)(a.ray_targets + (size_t)i * 8))[0], t1 = ((const float4*)(a.ray_targets + (size_t)i * 8))[1];
				rgbtarget = mk3(t0.x, t0.y, t0.z); background_color = mk3(t0.w, t1.x, t1.y);
			} else {
				const uint32_t ray_idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.ray_indices_in[i]);
				Rng rng(a.rng);
				rng.advance((uint64_t)(ray_idx * N_RANDOM_PER_RAY));
				uint32_t img; float pdf_unused;
				const f2 uv = training_pixel(a.cdf, rng, ray_idx, n_rays, a.n_images, a.metadata, a.snap_to_pixel_centers, img, pdf_unused);
				const ngp_image_meta& m = a.metadata[img];
				rng.advance(1); // motionblur_time
				if (a.random_bg_color) { background_color.x = rng.next_float(); background_color.y = rng.next_float(); background_color.z = rng.next_float(); }
				const f4 tex = read_rgba(uv, m.resolution, m.pixels, m.image_data_type);
				background_color = srgb_to_linear3(background_color);
				const f3 trgb = mk3(tex.x, tex.y, tex.z);
				if (a.linear_colors || !a.color_space_srgb) {
					rgbtarget = trgb + (1.0f - tex.w) * background_color;
					if (!a.linear_colors) { rgbtarget = linear_to_srgb3(rgbtarget); background_color = linear_to_srgb3(background_color); }
				} else {
					background_color = linear_to_srgb3(background_color);
					if (tex.w > 0) rgbtarget = linear_to_srgb3(trgb / tex.w) * tex.w + (1.0f - tex.w) * background_color;
					else rgbtarget = background_color;
				}
			}
			const float* cin = a.coords_in + (size_t)base * cs;
			const __half* no = (const __half*)a.network_output + (size_t)base * a.output_stride;
			float T_run = 1.f;
			uint32_t compacted = 0;
			for (uint32_t c0 = 0; c0 < numsteps; c0 += 64) {
				const uint32_t s = c0 + lane;
				const bool valid = s < numsteps;
				float alpha = 0.f; f3 rgb = mk3(0.f);
				if (valid) {
					float l0, l1, l2, l3;
					load_out(no + (size_t)s * a.output_stride, l0, l1, l2, l3);
					const float dtw = cin[(size_t)s * cs + 3];
					rgb = mk3(act_rgb(l0, a.rgb_activation), act_rgb(l1, a.rgb_activation), act_rgb(l2, a.rgb_activation));
					const float dt = unwarp_dt(dtw);
					alpha = 1.f - __expf(-act_density(l3, a.density_activation) * dt);
				}
				const float incl = wave_incl_prod(1.f - alpha, lane);
				float excl = __shfl_up(incl, 1, 64);
				if (lane == 0) excl = 1.f;
				const float T_k = T_run * excl;
				const uint64_t vm = __ballot(valid), fail = __ballot(valid && !(T_k >= EPSILON)); // `if (T < EPSILON) break;`
				const uint32_t n_proc = fail ? (uint32_t)(__ffsll((long long)fail) - 1) : (uint32_t)__popcll(vm);
				const bool proc = lane < n_proc;
				const float w = proc ? alpha * T_k : 0.f;
				// lanes behind the cut may hold unevaluated network outputs (lazy K2): select, never multiply
				rgb_ray = rgb_ray + mk3(wave_total(proc ? w * rgb.x : 0.f), wave_total(proc ? w * rgb.y : 0.f), wave_total(proc ? w * rgb.z : 0.f));
				if (a.train_mode == 1) { // Rfl: sum of weight * per-sample loss (train_nerf.cuh:219)
					f3 ll, lgl; loss_and_gradient(rgbtarget, rgb, a.loss_type, ll, lgl);
					loss_bg = loss_bg + mk3(wave_total(proc ? w * ll.x : 0.f), wave_total(proc ? w * ll.y : 0.f), wave_total(proc ? w * ll.z : 0.f));
				}
				if (n_proc) T_run = T_run * __shfl(incl, (int)n_proc - 1, 64);
				compacted += n_proc;
				if (fail) break;
			}
			if (compacted == numsteps) {
				rgb_ray = rgb_ray + T_run * background_color;
				if (a.train_mode == 1) { f3 ll, lgl; loss_and_gradient(rgbtarget, background_color, a.loss_type, ll, lgl); loss_bg = loss_bg + T_run * ll; }
			}
			if (lane == 0) {
				float4* r4 = (float4*)(rec + (size_t)i * K3_REC);
				r4[0] = make_float4(__uint_as_float(compacted), rgb_ray.x, rgb_ray.y, rgb_ray.z);
				r4[1] = make_float4(loss_bg.x, loss_bg.y, loss_bg.z, 0.f);
				r4[2] = make_float4(rgbtarget.x, rgbtarget.y, rgbtarget.z, 0.f);
				r4[3] = make_float4(background_color.x, background_color.y, background_color.z, 0.f);
			}
			wtot += compacted;
		}
		if (lane == 0) s_tot[wid] = wtot;
		__syncthreads();
		if (threadIdx.x == 0) { // see k1_count: returning exchange + two-level ticket, no fence
			const uint64_t prev = atomicExch((unsigned long long*)(partial + blockIdx.x), (unsigned long long)((s_tot[0] + s_tot[1]) + (s_tot[2] + s_tot[3])));
			const uint32_t cls = blockIdx.x % K1_TICKET_CLASSES, n_cls = (gridDim.x - cls + K1_TICKET_CLASSES - 1) / K1_TICKET_CLASSES;
			uint32_t last = 0u;
			if (atomicAdd(done + 1 + cls, (uint32_t)(prev >> 63) + 1u) == n_cls - 1) {
				atomicExch(done + 1 + cls, 0u);
				last = atomicAdd(done, 1u) == min(gridDim.x, K1_TICKET_CLASSES) - 1 ? 1u : 0u;
			}
			s_ticket = last;
		}
		__syncthreads();
		if (!s_ticket) return;
		uint64_t run = *a.numsteps_counter_compacted; // the spans start behind whatever the counter holds (0 in the training loop)
		for (uint32_t b0 = 0; b0 < gridDim.x; b0 += SCAN_BLOCK) {
			uint64_t v[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) { const uint32_t e = b0 + threadIdx.x * 4 + k; v[k] = e < gridDim.x ? (uint64_t)atomicAdd((unsigned long long*)(partial + e), 0ull) : 0ull; }
			uint64_t tot;
			uint64_t pre = run + block_excl_scan_1024(v, s_scan, tot);
#pragma unroll
			for (int k = 0; k < 4; ++k) { const uint32_t e = b0 + threadIdx.x * 4 + k; if (e < gridDim.x) partial[e] = pre; pre += v[k]; }
			run += tot;
			__syncthreads();
		}
		if (threadIdx.x == 0) { *a.numsteps_counter_compacted = (uint32_t)min(run, (uint64_t)0xffffffffu); atomicExch(done, 0u); }
		return;
	}
	// ---- PASS 1 ----
	{
		uint64_t v[4] = {0ull, 0ull, 0ull, 0ull};
		if (r_begin + threadIdx.x < r_end) v[0] = __float_as_uint(rec[(size_t)(r_begin + threadIdx.x) * K3_REC]);
		uint64_t tot;
		const uint64_t pre = block_excl_scan_1024(v, s_scan, tot) + partial[blockIdx.x];
		if (threadIdx.x < K1_MAX_RANGE) s_off[threadIdx.x] = (uint32_t)min(pre, (uint64_t)0xffffffffu);
		__syncthreads();
	}
	float wave_loss = 0.f;
	for (uint32_t i = r_begin + wid; i < r_end; i += 4) {
		const float4* r4 = (const float4*)(rec + (size_t)i * K3_REC);
		const float4 q0 = r4[0], q1 = r4[1], q2 = r4[2], q3 = r4[3];
		const uint32_t compacted_base = s_off[i - r_begin];
		uint32_t compacted = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(q0.x));
		const f3 rgb_ray = mk3(q0.y, q0.z, q0.w), loss_bg = mk3(q1.x, q1.y, q1.z), rgbtarget = mk3(q2.x, q2.y, q2.z);
		const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.numsteps_inout[i * 2 + 1]);
		const f3 ray_o = ld3(a.rays_in[i].o);
		compacted = min(a.max_samples_compacted - min(a.max_samples_compacted, compacted_base), compacted);
		if (lane == 0) { a.numsteps_inout[i * 2 + 0] = compacted; a.numsteps_inout[i * 2 + 1] = compacted_base; }
		if (compacted == 0) continue;
		const float* cin = a.coords_in + (size_t)base * cs;
		const __half* no = (const __half*)a.network_output + (size_t)base * a.output_stride;
		float* cout = a.coords_out + (size_t)compacted_base * cs;
		__half* dl = (__half*)a.dloss_doutput + (size_t)compacted_base * a.dloss_stride;
		f3 lloss, lgrad;
		loss_and_gradient(rgbtarget, rgb_ray, a.loss_type, lloss, lgrad);
		wave_loss += ((lloss.x + lloss.y + lloss.z) / 3.0f) / (float)n_rays;
		const float loss_scale = a.loss_scale / n_rays;
		const float output_l2_reg = a.rgb_activation == NGP_ACT_EXPONENTIAL ? 1e-4f : 0.0f;
		const float output_l1_reg_density = (a.train_mode == 0 && *a.mean_density_ptr < MIN_OPTICAL_THICKNESS) ? 1e-4f : 0.0f;
		float T_run = 1.f;
		f3 ray2_run = mk3(0.f), lb2_run = mk3(0.f);
		for (uint32_t c0 = 0; c0 < compacted; c0 += 64) {
			const uint32_t s = c0 + lane;
			const bool valid = s < compacted;
			float alpha = 0.f, l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f, dt = 0.f, depth = 0.f;
			f3 rgb = mk3(0.f);
			float cc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
			if (valid) {
				const float* ci = cin + (size_t)s * cs;
#pragma unroll
				for (int k = 0; k < 7; ++k) cc[k] = ci[k];
				load_out(no + (size_t)s * a.output_stride, l0, l1, l2, l3);
				rgb = mk3(act_rgb(l0, a.rgb_activation), act_rgb(l1, a.rgb_activation), act_rgb(l2, a.rgb_activation));
				dt = unwarp_dt(cc[3]);
				alpha = 1.f - __expf(-act_density(l3, a.density_activation) * dt);
				depth = dist3(unwarp_position(mk3(cc[0], cc[1], cc[2]), aabb), ray_o);
			}
			const float incl = wave_incl_prod(1.f - alpha, lane);
			float excl = __shfl_up(incl, 1, 64);
			if (lane == 0) excl = 1.f;
			const float T_k = T_run * excl, T_after = T_run * incl;
			const float weight = alpha * T_k;
			const f3 ray2 = ray2_run + mk3(wave_incl_sum(weight * rgb.x, lane), wave_incl_sum(weight * rgb.y, lane), wave_incl_sum(weight * rgb.z, lane));
			f3 lloc = mk3(0.f), gloc = mk3(0.f), lb2 = lb2_run;
			if (a.train_mode == 1) { // Rfl: per-sample loss against the target and its running (inclusive) weighted sum
				loss_and_gradient(rgbtarget, rgb, a.loss_type, lloc, gloc);
				lb2 = lb2_run + mk3(wave_incl_sum(weight * lloc.x, lane), wave_incl_sum(weight * lloc.y, lane), wave_incl_sum(weight * lloc.z, lane));
			}
			if (valid) {
				float* cj = cout + (size_t)s * cs;
#pragma unroll
				for (int k = 0; k < 7; ++k) cj[k] = cc[k];
				for (uint32_t k = 7; k < cs; ++k) cj[k] = cin[(size_t)s * cs + k];
				const f3 suffix = rgb_ray - ray2;
				f3 dloss_by_drgb = weight * lgrad;
				float dmlp_inner = dot3(lgrad, T_after * rgb - suffix) + 0.0f;
				if (a.train_mode == 1) { // fused_kernels/train_nerf.cuh:391-396
					dloss_by_drgb = weight * gloc;
					const f3 v = T_after * lloc - (loss_bg - lb2);
					dmlp_inner = v.x + v.y + v.z;
				} else if (a.train_mode == 2) { // train_nerf.cuh:397-405
					const f3 rgb_bg = suffix / fmaxf(1e-6f, T_after);
					const f3 rgb_lerp = (1 - alpha) * rgb_bg + alpha * rgb;
					f3 ll, lgl; loss_and_gradient(rgbtarget, rgb_lerp, a.loss_type, ll, lgl);
					dloss_by_drgb = weight * lgl;
					dmlp_inner = dot3(lgl, T_after * rgb - suffix) + 0.0f;
				}
				const float d0 = loss_scale * (dloss_by_drgb.x * act_rgb_d(l0, a.rgb_activation) + fmaxf(0.0f, output_l2_reg * l0));
				const float d1 = loss_scale * (dloss_by_drgb.y * act_rgb_d(l1, a.rgb_activation) + fmaxf(0.0f, output_l2_reg * l1));
				const float d2 = loss_scale * (dloss_by_drgb.z * act_rgb_d(l2, a.rgb_activation) + fmaxf(0.0f, output_l2_reg * l2));
				const float dloss_by_dmlp = act_density_d(l3, a.density_activation) * (dt * dmlp_inner);
				const float d3 = loss_scale * dloss_by_dmlp + (l3 < 0.0f ? -output_l1_reg_density : 0.0f) + (l3 > -10.0f && depth < a.near_distance ? 1e-4f : 0.0f);
				__half* d = dl + (size_t)s * a.dloss_stride;
				if (vec_dl) {
					const h4v v = {(_Float16)d0, (_Float16)d1, (_Float16)d2, (_Float16)d3};
					*(uint2*)d = __builtin_bit_cast(uint2, v);
				} else { d[0] = __float2half(d0); d[1] = __float2half(d1); d[2] = __float2half(d2); d[3] = __float2half(d3); }
			}
			T_run = T_run * __shfl(incl, 63, 64);
			ray2_run = mk3(__shfl(ray2.x, 63, 64), __shfl(ray2.y, 63, 64), __shfl(ray2.z, 63, 64));
			if (a.train_mode == 1) lb2_run = mk3(__shfl(lb2.x, 63, 64), __shfl(lb2.y, 63, 64), __shfl(lb2.z, 63, 64));
		}
		(void)q3;
	}
	if (lane == 0) s_loss[wid] = wave_loss;
	__syncthreads();
	if (a.loss_output && threadIdx.x == 0) { const float bl = (s_loss[0] + s_loss[1]) + (s_loss[2] + s_loss[3]); if (bl != 0.f) atomicAdd(a.loss_output, bl); }
}

// NerfCounters::update_after_training (testbed_nerf.cu:2669-2702) on the device-resident counters: one thread
static __device__ __forceinline__ void update_counters_body(TrainCounters* c, uint32_t target_batch_size, uint32_t world_size) {
	const uint32_t before = c->numsteps_counter, compacted = c->numsteps_counter_compacted;
	c->n_rays_last = c->ray_counter; c->rays_per_batch_last = c->rays_per_batch;
	c->total_rays += c->rays_per_batch;
	c->training_step += 1;
	if (before == 0 || compacted == 0) {
		c->measured_batch_size = 0; c->measured_batch_size_before_compaction = 0; c->loss_scalar = 0.f;
	} else {
		c->measured_batch_size_before_compaction = before;
		c->measured_batch_size = compacted;
		c->total_samples += compacted;
		c->loss_scalar = c->loss_sum * (float)compacted / (float)target_batch_size;
		uint32_t r = (uint32_t)((float)c->rays_per_batch * (float)target_batch_size / (float)compacted);
		r = ((r + 255u) / 256u) * 256u;
		// cap: 2^18 rays per rank (the per-rank ray buffers), i.e. the reference's 1<<18 (testbed_nerf.cu:2699) at world_size 1
		c->rays_per_batch = min(r, (1u << 18) * world_size);
	}
	// max_inference for the next step (testbed_nerf.cu:3055-3060)
	const uint32_t max_samples = target_batch_size * 16u;
	const uint32_t mb = c->measured_batch_size_before_compaction;
	c->max_inference = mb == 0 ? max_samples : ((min(mb, max_samples) + 255u) / 256u) * 256u;
	if (mb == 0) c->measured_batch_size_before_compaction = max_samples;
	c->numsteps_counter = 0; c->numsteps_counter_compacted = 0; c->ray_counter = 0; c->loss_sum = 0.f;
	c->n_valid_compacted = 0; for (int r = 0; r < 8; ++r) c->k2_tiles[r] = 0; c->k2_samples_last = c->k2_samples; c->k2_samples = 0;
}
// ------------------------------------------------------------------------------------------------
// K4: fill_rollover<float> on coords + fill_rollover_and_rescale<half> on dL/doutput, one launch.
// ------------------------------------------------------------------------------------------------
// With `ctl` the batch-size controller (k_update_counters) rides along: the last workgroup to finish -- a ticket counter; every read of
// K3's counter precedes the workgroup's ticket -- runs it, so the step has one launch (and one inter-kernel gap) less on its critical path.
__global__ void __launch_bounds__(256) k_fill_rollover(uint32_t n_elements, const uint32_t* __restrict__ n_input_ptr, float* __restrict__ coords, uint32_t cstride,
		__half* __restrict__ dloss, uint32_t dstride, const uint32_t* __restrict__ publish_src2, uint32_t* __restrict__ publish_dst2, const float* __restrict__ publish_loss,
		TrainCounters* ctl, uint32_t ctl_world_size, int zero_padding) {
	// the words every rank must agree on (8e) are published here instead of by a separate copy: {marched, compacted} samples and this rank's
	// share of the loss (K3 normalises by the GLOBAL ray count, so the sum over ranks is the loss of the union batch) as unsigned fixed point
	// in units of 2^-24 -- one integer all-reduce(sum) covers all three
	if (publish_dst2 && blockIdx.x == 0 && threadIdx.x == 0) {
		publish_dst2[0] = publish_src2[0]; publish_dst2[1] = publish_src2[1];
		if (publish_loss) { const float l = *publish_loss; publish_dst2[2] = l > 0.f ? (uint32_t)(fminf(l, 16.f) * 16777216.f) : 0u; } // NaN -> 0
		publish_dst2[3] = min(*n_input_ptr, n_elements); // THIS rank's valid batch rows (not all-reduced; T1 reads it, EncStashIn::n_valid_ptr -- the controller resets K3's counter)
	}
	const uint32_t n_in = min(*n_input_ptr, n_elements); // K3's counter may overshoot the batch (its spans are clamped)
	if (n_in != 0 && n_in < n_elements) {
		const float n_input = (float)(n_in * dstride), n_total = (float)(n_elements * dstride);
		for (uint32_t e = n_in + blockIdx.x * blockDim.x + threadIdx.x; e < n_elements; e += gridDim.x * blockDim.x) { // destination element
			const uint32_t src = e % n_in;
			// coords: element-wise wrap is identical to the reference's flat index wrap because
			// (e*stride + k) % (n_in*stride) == (e % n_in)*stride + k
			for (uint32_t k = 0; k < cstride; ++k) coords[(size_t)e * cstride + k] = coords[(size_t)src * cstride + k];
			for (uint32_t k = 0; k < dstride; ++k) {
				float v = __half2float(dloss[(size_t)src * dstride + k]);
				dloss[(size_t)e * dstride + k] = __float2half(zero_padding ? 0.f : v * n_input / n_total);
			}
		}
	}
	if (!ctl) return;
	__syncthreads(); // every thread of the workgroup has read the counter
	if (threadIdx.x != 0) return;
	// no fence: the counters the controller reads were written by earlier kernels, and this workgroup's own read of K3's counter has
	// completed (its value was used above) before the ticket is drawn
	if (atomicAdd(&ctl->k4_ticket, 1u) != gridDim.x - 1) return;
	atomicExch(&ctl->k4_ticket, 0u);
	update_counters_body(ctl, n_elements, ctl_world_size);
}

// ------------------------------------------------------------------------------------------------
// occupancy grid
// ------------------------------------------------------------------------------------------------
__global__ void k_mark_untrained(uint32_t n_elements, float* __restrict__ grid, uint32_t n_images, const ngp_image_meta* __restrict__ metadata,
		const ngp_xform* __restrict__ xforms, int clear_visible) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	const uint32_t level = i / GRID_N_CELLS, pos_idx = i % GRID_N_CELLS;
	const uint32_t x = morton3D_invert(pos_idx >> 0), y = morton3D_invert(pos_idx >> 1), z = morton3D_invert(pos_idx >> 2);
	const float voxel_size = scalbnf(1.0f / GRIDSIZE, (int)level);
	const f3 pos = (mk3((float)x, (float)y, (float)z) / (float)GRIDSIZE - 0.5f) * scalbnf(1.0f, (int)level) + 0.5f;
	uint32_t count = 0;
	for (uint32_t j = 0; j < n_images && count < 1; ++j) {
		const M43 xf = ldm43(xforms[j].start);
		const ngp_image_meta& m = metadata[j];
		// f-theta lenses have no forward mapping and are assumed to see everything; lat-long / equirectangular ones do (testbed_nerf.cu:131-136)
		if (m.lens_mode == NGP_LENS_FTHETA || lens_is_360(m.lens_mode)) { ++count; continue; }
		for (uint32_t k = 0; k < 8; ++k) {
			const f3 corner = pos + mk3((k & 1) ? voxel_size : 0.f, (k & 2) ? voxel_size : 0.f, (k & 4) ? voxel_size : 0.f);
			const f3 dir = normalize3(corner - xf.c[3]);
			if (dot3(dir, xf.c[2]) < 1e-4f) continue;
			const f2 uv = pos_to_uv(corner, m.resolution, m.focal_length, xf, m.principal_point, m.lens_mode, m.lens_params);
			f3 ro, rd;
			uv_to_ray(uv, m.resolution, m.focal_length, xf, m.principal_point, m.lens_mode, m.lens_params, 0.0f, ro, rd);
			if (dist3(normalize3(rd), dir) < 1e-3f && uv.x > 0.0f && uv.y > 0.0f && uv.x < 1.0f && uv.y < 1.0f) { ++count; break; }
		}
	}
	if (clear_visible || (grid[i] < 0) != (count < 1)) grid[i] = (count >= 1) ? 0.f : -1.f;
}

__global__ void k_generate_grid_samples(uint32_t n_elements, ngp_pcg32 rng_in, const uint32_t* __restrict__ step_ptr, uint32_t step_imm, ngp_aabb box,
		const float* __restrict__ grid_in, float* __restrict__ out_pos, uint32_t* __restrict__ indices, uint32_t n_cascades, float thresh) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	const uint32_t step = step_ptr ? *step_ptr : step_imm;
	Rng rng(rng_in);
	rng.advance((uint64_t)(i * 4u));
	const uint32_t level = (uint32_t)(rng.next_float() * n_cascades) % n_cascades;
	// up to ten candidate cells, the first one above the threshold (the last one if none is): the first candidate alone (the uniform samples nearly always keep it), then the
	// other nine with their loads in flight together instead of one dependent round trip per rejected candidate (the non-uniform samples reject most)
	const uint32_t h0 = (i + step * n_elements) * 56924617u + 96925573u;
	uint32_t idx = h0 % GRID_N_CELLS + level * GRID_N_CELLS;
	if (!(grid_in[idx] > thresh)) {
		uint32_t cand[9]; float val[9];
#pragma unroll
		for (uint32_t j = 1; j < 10; ++j) { cand[j - 1] = (h0 + j * 19349663u) % GRID_N_CELLS + level * GRID_N_CELLS; val[j - 1] = grid_in[cand[j - 1]]; }
		idx = cand[8];
#pragma unroll
		for (int j = 7; j >= 0; --j) idx = val[j] > thresh ? cand[j] : idx;
	}
	const uint32_t pos_idx = idx % GRID_N_CELLS;
	const uint32_t x = morton3D_invert(pos_idx >> 0), y = morton3D_invert(pos_idx >> 1), z = morton3D_invert(pos_idx >> 2);
	f3 r; r.x = rng.next_float(); r.y = rng.next_float(); r.z = rng.next_float();
	const f3 pos = ((mk3((float)x, (float)y, (float)z) + r) / (float)GRIDSIZE - 0.5f) * scalbnf(1.0f, (int)level) + 0.5f;
	const f3 wp = warp_position(pos, Box(box));
	out_pos[(size_t)i * 3 + 0] = wp.x; out_pos[(size_t)i * 3 + 1] = wp.y; out_pos[(size_t)i * 3 + 2] = wp.z;
	indices[i] = idx;
}

__global__ void k_splat_grid_samples(uint32_t n, const uint32_t* __restrict__ indices, const __half* __restrict__ net_out, uint32_t stride,
		float* __restrict__ grid_out, int density_activation) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n) return;
	const float mlp = act_density(__half2float(net_out[(size_t)i * stride]), density_activation);
	const float optical_thickness = mlp * scalbnf(MIN_CONE_STEP, 0);
	// positive floats order like their uint bit patterns
	atomicMax((uint32_t*)&grid_out[indices[i]], __float_as_uint(optical_thickness));
}

__global__ void k_ema_grid_samples(uint32_t n, float decay, float* __restrict__ grid_out, const float* __restrict__ grid_in) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n) return;
	const float prev = grid_out[i];
	grid_out[i] = (prev < 0.f) ? prev : fmaxf(prev * decay, grid_in[i]);
}

// deterministic two-stage mean of max(v,0)/N over cascade 0 (reduce_sum, testbed_nerf.cu:2602-2608)
constexpr uint32_t MEAN_BLOCKS = 256;
__global__ void __launch_bounds__(256) k_grid_mean_partial(const float* __restrict__ grid, float* __restrict__ partial) {
	__shared__ float sm[4];
	float s = 0.f;
	const uint32_t per_block = GRID_N_CELLS / MEAN_BLOCKS;
	const float* g = grid + (size_t)blockIdx.x * per_block;
	for (uint32_t k = threadIdx.x; k < per_block; k += 256) s += fmaxf(g[k], 0.f) / (float)GRID_N_CELLS;
	s = wave_sum(s);
	if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) partial[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
__global__ void __launch_bounds__(64) k_grid_mean_final(const float* __restrict__ partial, float* __restrict__ mean_out) {
	float s = 0.f;
	for (uint32_t k = threadIdx.x; k < MEAN_BLOCKS; k += 64) s += partial[k];
	s = wave_sum(s);
	if (threadIdx.x == 0) *mean_out = s;
}

__global__ void k_grid_to_bitfield(uint32_t n_elements, uint32_t n_nonzero, const float* __restrict__ grid, uint8_t* __restrict__ bitfield,
		const float* __restrict__ mean_ptr) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	if (i >= n_nonzero) { bitfield[i] = 0; return; }
	const float thresh = fminf(MIN_OPTICAL_THICKNESS, *mean_ptr);
	const float4 lo = ((const float4*)grid)[(size_t)i * 2], hi = ((const float4*)grid)[(size_t)i * 2 + 1];
	uint8_t bits = 0;
	bits |= lo.x > thresh ? 1 : 0; bits |= lo.y > thresh ? 2 : 0; bits |= lo.z > thresh ? 4 : 0; bits |= lo.w > thresh ? 8 : 0;
	bits |= hi.x > thresh ? 16 : 0; bits |= hi.y > thresh ? 32 : 0; bits |= hi.z > thresh ? 64 : 0; bits |= hi.w > thresh ? 128 : 0;
	bitfield[i] = bits;
}
__global__ void k_bitfield_max_pool(uint32_t n_elements, const uint8_t* __restrict__ prev_level, uint8_t* __restrict__ next_level) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	const uint64_t p = ((const uint64_t*)prev_level)[i];
	uint8_t bits = 0;
#pragma unroll
	for (uint32_t j = 0; j < 8; ++j) bits |= ((p >> (8 * j)) & 0xffull) ? (uint8_t)(1u << j) : 0;
	const uint32_t x = morton3D_invert(i >> 0) + GRIDSIZE / 8, y = morton3D_invert(i >> 1) + GRIDSIZE / 8, z = morton3D_invert(i >> 2) + GRIDSIZE / 8;
	next_level[morton3D(x, y, z)] |= bits;
}

// ------------------------------------------------------------------------------------------------
// on-device NerfCounters::update_after_training (testbed_nerf.cu:2678-2702) + counter reset
// (prepare_for_training_steps :2669-2676) for the NEXT step.  c = TrainCounters in device memory.
// ------------------------------------------------------------------------------------------------
__global__ void k_update_counters(TrainCounters* c, uint32_t target_batch_size, uint32_t world_size) {
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	update_counters_body(c, target_batch_size, world_size);
}
// clamp the compacted counter to B for K4 / statistics (the reference relies on fill_rollover's guard)
__global__ void k_clamp_compacted(TrainCounters* c, uint32_t target_batch_size) {
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	c->n_valid_compacted = min(c->numsteps_counter_compacted, target_batch_size);
	c->n_inference = min(c->numsteps_counter, c->max_inference);
}

// compute_extra_dims_gradient_train_nerf (testbed_nerf.cu:1293-1330): one thread per compacted ray sums its samples' dL/d(extra dims) (the network's input gradient, left
// per batch row by T1) into its image's gradient with float atomics.  numsteps = {compacted count, compacted base} as K3 left them; rays whose span was cut by the batch
// size carry count 0.
__global__ void __launch_bounds__(128) k_extra_dims_gradient(const uint32_t* __restrict__ n_rays_total_ptr, const uint32_t* __restrict__ rays_counter, float* __restrict__ extra_grad,
		uint32_t n_extra, uint32_t n_images, const uint32_t* __restrict__ ray_indices, const uint32_t* __restrict__ numsteps, const float* __restrict__ dextra, uint32_t max_rows,
		const float* __restrict__ cdf_img) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= *rays_counter) return;
	const uint32_t count = numsteps[i * 2 + 0], base = numsteps[i * 2 + 1];
	if (count == 0 || base + count > max_rows) return;
	const uint32_t ray_idx = ray_indices[i];
	const uint32_t img = cdf_img ? image_idx_cdf(ray_idx, n_images, cdf_img, nullptr) : image_idx(ray_idx, *n_rays_total_ptr, n_images);
	float* g = extra_grad + (size_t)img * n_extra;
	for (uint32_t k = 0; k < n_extra; ++k) {
		float sum = 0.f; // (the reference issues one atomic per sample and dim; the ray's samples are summed here first -- same set of fp32 additions up to their order)
		for (uint32_t j = 0; j < count; ++j) sum += dextra[(size_t)(base + j) * n_extra + k];
		atomicAdd(g + k, sum);
	}
}
// VarAdamOptimizer::step (adam_optimizer.h:37-47) for every image's variable at once: one thread per (image, dim); gradient / LOSS_SCALE as in testbed_nerf.cu:2868.
// iters != nullptr: every image has its own optimizer with its own iteration count (std::vector<VarAdamOptimizer> extra_dims_opt: an image that joins the training set
// later starts at 0); iters[image] holds the count BEFORE this step, k_extra_dims_iter_inc advances the counts behind this kernel.  iters == nullptr: `iter` for all.
__global__ void __launch_bounds__(128) k_extra_dims_adam(uint32_t n, float* __restrict__ variable, const float* __restrict__ gradient, float* __restrict__ m, float* __restrict__ v,
		uint32_t iter, float lr, float loss_scale, const uint32_t* __restrict__ iters, uint32_t n_extra) {
#pragma clang fp contract(off)
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n) return;
	if (iters) iter = iters[i / n_extra] + 1u;
	const float beta1 = 0.9f, beta2 = 0.99f, epsilon = 1e-8f; // VarAdamOptimizer(n_extra_dims, 1e-4f): the defaults of adam_optimizer.h:29
	const float actual_learning_rate = lr * sqrtf(1.0f - powf(beta2, (float)iter)) / (1.0f - powf(beta1, (float)iter));
	const float g = gradient[i] / loss_scale;
	const float fm = m[i] = beta1 * m[i] + (1.0f - beta1) * g;
	const float sm = v[i] = beta2 * v[i] + (1.0f - beta2) * g * g;
	variable[i] -= actual_learning_rate * fm / (sqrtf(sm) + epsilon);
}
__global__ void __launch_bounds__(128) k_extra_dims_iter_inc(uint32_t n_images, uint32_t* __restrict__ iters) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i < n_images) ++iters[i];
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline uint32_t blocks(uint32_t n, uint32_t t) { return (n + t - 1) / t; }
void launch_extra_dims_gradient(hipStream_t s, uint32_t max_rays, const uint32_t* n_rays_total_ptr, const uint32_t* rays_counter, float* extra_grad, uint32_t n_extra, uint32_t n_images,
		const uint32_t* ray_indices, const uint32_t* numsteps, const float* dextra, uint32_t max_rows, const float* cdf_img) {
	hipLaunchKernelGGL(k_extra_dims_gradient, dim3(blocks(max_rays, 128)), dim3(128), 0, s, n_rays_total_ptr, rays_counter, extra_grad, n_extra, n_images, ray_indices, numsteps, dextra, max_rows, cdf_img);
}
void launch_extra_dims_adam(hipStream_t s, uint32_t n, float* variable, const float* gradient, float* m, float* v, uint32_t iter, float lr, float loss_scale, uint32_t* iters, uint32_t n_extra) {
	if (!n) return;
	hipLaunchKernelGGL(k_extra_dims_adam, dim3(blocks(n, 128)), dim3(128), 0, s, n, variable, gradient, m, v, iter, lr, loss_scale, iters, n_extra ? n_extra : 1u);
	if (iters) hipLaunchKernelGGL(k_extra_dims_iter_inc, dim3(blocks(n / n_extra, 128)), dim3(128), 0, s, n / n_extra, iters);
}

void launch_generate_training_samples(hipStream_t s, const K1Args& a, uint32_t max_rays_this_rank) {
	if (max_rays_this_rank == 0) return;
	hipLaunchKernelGGL(k_generate_training_samples, dim3(blocks(max_rays_this_rank, 128)), dim3(128), 0, s, a);
}
// persistent grid: up to 8 workgroups of 4 wavefronts per CU; every workgroup owns at most K1_MAX_RANGE consecutive slots
// The marcher's grid must equal the number of RESIDENT workgroups: every workgroup owns a contiguous slot range, so workgroups that start only when others retire
// are a second round at partial occupancy.  k1_count<8, true> (one cascade) holds 77 registers = 6 workgroups of 4 wavefronts per CU -- the grid of 8 per CU used up
// to round 3a ran 1.33 rounds: K1 0.181 -> 0.171 ms, step 0.600 -> 0.587 ms with 6 (profiles/r03_microbench_k1_grid.log).  k1_count<8, false>: one round as well since
// round 6 (its critical path is a workgroup's batches of four rays).  Scratch is sized for the largest grid.
constexpr uint32_t K1_MAX_BLOCKS_PER_CU = 16;
static uint32_t k1_blocks_per_cu(bool single_cascade) { return single_cascade ? 6u : 4u; } // (multi-cascade, two-phase kernel with ~50 KiB of LDS: fox step 556 / 525 / 525 / 528 us at 8 / 4 / 3 / 2, profiles/r06_ab_k1_count_two_phase.txt)
static uint32_t k1_blocks_per_cu_segments() { return 4u; } // (36 KiB of LDS per workgroup)
static uint32_t k1_grid(uint32_t max_local_rays, uint32_t blocks_per_cu = K1_MAX_BLOCKS_PER_CU) { return std::max(std::min<uint32_t>(blocks(max_local_rays, 4), 256u * blocks_per_cu), blocks(max_local_rays, K1_MAX_RANGE)); }
// byte offset of the workgroup totals behind the RaySetup and mask arrays (64-bit atomics: naturally aligned)
static size_t k1_partial_offset(uint32_t max_local_rays) { return ((size_t)max_local_rays * (sizeof(RaySetup) + LAT_MAX_CHUNKS * 8) + 15) / 16 * 16; }
// ... and, behind the ticket counters, the per-slot sample lists of k1_count_segments (N_STEPS 16-bit lattice indices per slot: 512 MiB for 2^18 slots, of which a step touches
// the first ~64 bytes of ~5 10^4)
static size_t k1_jlist_offset(uint32_t max_local_rays) { return k1_partial_offset(max_local_rays) + ((size_t)k1_grid(max_local_rays) * 8 + 256 + 15) / 16 * 16; }
size_t k1_lattice_scratch_bytes(uint32_t max_local_rays) {
	return k1_jlist_offset(max_local_rays) + (size_t)max_local_rays * N_STEPS * 2;
}
// the ticket counter behind the workgroup totals must start at zero (k1_count's last workgroup leaves it at zero again)
int k1_lattice_scratch_init(hipStream_t s, void* scratch, uint32_t max_local_rays) {
	char* p = (char*)scratch + k1_partial_offset(max_local_rays);
	return hipMemsetAsync(p, 0, (size_t)k1_grid(max_local_rays) * 8 + 256, s) == hipSuccess ? 0 : 1;
}
// two-pass K3: per-ray records + workgroup totals + ticket counters (zeroed once, like K1's)
size_t k3_scratch_bytes(uint32_t max_rays) { return (size_t)max_rays * K3_REC * 4 + (size_t)k1_grid(max_rays) * 8 + 256; }
int k3_scratch_init(hipStream_t s, void* scratch, uint32_t max_rays) {
	return hipMemsetAsync((char*)scratch + (size_t)max_rays * K3_REC * 4, 0, (size_t)k1_grid(max_rays) * 8 + 256, s) == hipSuccess ? 0 : 1;
}
// K1 as three launches: setup (thread per ray), count (wave per ray + the prefix sum over workgroup totals), write (wave per ray)
void launch_generate_training_samples_lattice(hipStream_t s, const K1Args& a, uint32_t max_local_rays, void* scratch, bool count_only) {
	if (max_local_rays == 0) return;
	char* p = (char*)scratch;
	RaySetup* rs = (RaySetup*)p; p += (size_t)max_local_rays * sizeof(RaySetup);
	uint64_t* masks = (uint64_t*)p;
	// single-cascade instance: max_mip == 0 and a constant step below one cell (dt * 2 * GRIDSIZE < 1: mip_from_dt never looks at the step)
	const bool single = a.max_mip == 0 && a.cone_angle_constant <= 1e-5f;
	const uint32_t ray_grid = k1_grid(max_local_rays, k1_blocks_per_cu(single));
	p = (char*)scratch + k1_partial_offset(max_local_rays);
	uint64_t* partial = (uint64_t*)p; p += (size_t)ray_grid * 8;
	uint32_t* done = (uint32_t*)p;
	const bool plain = a.plain_dataset && !(g_debug_flags2 & DBG2_K1_SETUP_GENERAL) && !a.cdf.img && !a.cdf.x_cond_y && !(a.depth_lambda > 0.0f);
	const uint32_t setup_grid = std::min<uint32_t>(blocks(max_local_rays, 128), 512u); // 2^16 slots per pass of the kernel's grid-stride loop
	if (plain) hipLaunchKernelGGL(k1_setup<true>, dim3(setup_grid, a.ray_targets_out ? 4 : 1), dim3(128), 0, s, a, rs);
	else hipLaunchKernelGGL(k1_setup<false>, dim3(setup_grid, a.ray_targets_out ? 4 : 1), dim3(128), 0, s, a, rs);
	if (single && a.bitfield_coarse && a.bitfield_linear && !a.chunk_march && !a.no_first_point_skip) {
		// one cascade, constant step: segment prepass + sample lists.  36 KiB of LDS per workgroup: 4 resident per CU = the persistent grid
		uint16_t* jlist = (uint16_t*)((char*)scratch + k1_jlist_offset(max_local_rays));
		const uint32_t seg_grid = k1_grid(max_local_rays, k1_blocks_per_cu_segments());
		static const uint32_t seg_waves = getenv("NGP_K1_SEG_WAVES") && atoi(getenv("NGP_K1_SEG_WAVES")) == 4 ? 4u : 8u; // 4: the round-4/5 shape (ablation)
		if (seg_waves == 8) hipLaunchKernelGGL(k1_count_segments<8>, dim3(seg_grid), dim3(512), (COARSE_WORDS + MID_WORDS) * 4, s, a, rs, jlist, partial, done);
		else hipLaunchKernelGGL(k1_count_segments<4>, dim3(seg_grid), dim3(256), (COARSE_WORDS + MID_WORDS) * 4, s, a, rs, jlist, partial, done);
		if (!count_only) hipLaunchKernelGGL(k1_write_list, dim3(seg_grid * K1_WRITE_SPLIT), dim3(256), 0, s, a, rs, jlist, partial);
		return;
	}
	// 8 chunks (512 lattice points) in flight per iteration; 16 measured slower (143 -> 159 us: SGPR pressure, profiles/r02_k1_experiments.txt)
	constexpr uint32_t rec_bytes = 4u * LAT_MAX_CHUNKS * (8u + 8u + 64u * 2u + 1u) + 4u * 64u; // k1_count's chunk records of four rays: masks, skip lengths, mip; the walk's marks
	const bool prefilter = a.bitfield_coarse != nullptr && a.bitfield_linear != nullptr;
	if (single) hipLaunchKernelGGL((k1_count<8, true>), dim3(ray_grid), dim3(256), (prefilter ? COARSE_WORDS * 4 : 0) + rec_bytes, s, a, rs, masks, partial, done);
	else {
		K1Args a2 = a;
		a2.n_mips_lds = std::min<uint32_t>(a.n_mips, std::max<uint32_t>(a.max_mip + 2u, 4u)); // fox (max_mip 2): 4 levels = 16 KiB + 18.6 KiB of records: four workgroups per CU (all eight levels: three)
		hipLaunchKernelGGL((k1_count<8, false>), dim3(ray_grid), dim3(256), (prefilter ? a2.n_mips_lds * COARSE_WORDS * 4 : 0) + rec_bytes, s, a2, rs, masks, partial, done);
	}
	if (!count_only) hipLaunchKernelGGL(k1_write, dim3(ray_grid), dim3(256), 0, s, a, rs, masks, partial);
}
void launch_build_linear_bitfield(hipStream_t s, const uint8_t* bitfield, uint8_t* linear, uint32_t n_cascades, uint32_t* coarse) {
	const uint32_t n_bytes = GRID_N_CELLS / 8 * n_cascades;
	hipLaunchKernelGGL(k_build_linear_bitfield, dim3(blocks(n_bytes, 256)), dim3(256), 0, s, bitfield, linear, n_bytes);
	if (coarse) { // k1_prefilter_words(n_cascades): [n_cascades x COARSE_WORDS coarse][MID_WORDS dilated mid grid of cascade 0]
		hipLaunchKernelGGL(k_build_coarse_bitfield, dim3(blocks(COARSE_WORDS * 32 * n_cascades, 256)), dim3(256), 0, s, linear, coarse, n_cascades);
		hipLaunchKernelGGL(k_build_mid_dilated_bitfield, dim3(MID_WORDS * 32 / 256), dim3(256), 0, s, linear, coarse + (size_t)COARSE_WORDS * n_cascades);
	}
}
// ---- CDFs of the accumulated error (testbed_nerf.cu:1530-1580, 2795-2847) ----
// construct_cdf_2d: one thread per (row, image): running sum along x (+1e-10 per cell), the row total goes to cdf_y, the row is normalised and mixed with 1 % uniform
__global__ void __launch_bounds__(128) k_construct_cdf_2d(uint32_t n_images, uint32_t height, uint32_t width, const float* __restrict__ data, float* __restrict__ cdf_x_cond_y, float* __restrict__ cdf_y) {
	const uint32_t y = threadIdx.x + blockIdx.x * blockDim.x, img = blockIdx.y;
	if (y >= height || img >= n_images) return;
	const size_t offset_xy = ((size_t)img * height + y) * width;
	data += offset_xy; cdf_x_cond_y += offset_xy;
	const float MIN_PDF = 0.01f;
	float cum = 0;
	for (uint32_t x = 0; x < width; ++x) { cum += data[x] + 1e-10f; cdf_x_cond_y[x] = cum; }
	cdf_y[img * height + y] = cum;
	const float norm = 1.0f / cum; // __frcp_rn: correctly rounded, like this division
	for (uint32_t x = 0; x < width; ++x) cdf_x_cond_y[x] = (1.0f - MIN_PDF) * cdf_x_cond_y[x] * norm + MIN_PDF * (float)(x + 1) / (float)width;
}
// construct_cdf_1d: one thread per image over its rows; the image total goes to cdf_img
__global__ void __launch_bounds__(64) k_construct_cdf_1d(uint32_t n_images, uint32_t height, float* __restrict__ cdf_y, float* __restrict__ cdf_img) {
	const uint32_t img = threadIdx.x + blockIdx.x * blockDim.x;
	if (img >= n_images) return;
	cdf_y += (size_t)img * height;
	const float MIN_PDF = 0.01f;
	float cum = 0;
	for (uint32_t y = 0; y < height; ++y) { cum += cdf_y[y]; cdf_y[y] = cum; }
	cdf_img[img] = cum;
	const float norm = 1.0f / cum;
	for (uint32_t y = 0; y < height; ++y) cdf_y[y] = (1.0f - MIN_PDF) * cdf_y[y] * norm + MIN_PDF * (float)(y + 1) / (float)height;
}
// testbed_nerf.cu:2832-2847 (a host loop in the reference: "single-threaded anyway"): same order of additions, one lane, no round trip to the host
__global__ void k_construct_cdf_img(uint32_t n_images, float* __restrict__ cdf_img) {
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	const float MIN_PMF = 0.1f;
	float cum = 0;
	for (uint32_t i = 0; i < n_images; ++i) { cum += cdf_img[i]; cdf_img[i] = cum; }
	const float norm = 1.0f / cum;
	for (uint32_t i = 0; i < n_images; ++i) cdf_img[i] = (1.0f - MIN_PMF) * cdf_img[i] * norm + MIN_PMF * (float)(i + 1) / (float)n_images;
}
void launch_construct_error_cdfs(hipStream_t s, uint32_t n_images, uint32_t width, uint32_t height, const float* error_map, float* cdf_x_cond_y, float* cdf_y, float* cdf_img) {
	hipLaunchKernelGGL(k_construct_cdf_2d, dim3(blocks(height, 128), n_images), dim3(128), 0, s, n_images, height, width, error_map, cdf_x_cond_y, cdf_y);
	hipLaunchKernelGGL(k_construct_cdf_1d, dim3(blocks(n_images, 64)), dim3(64), 0, s, n_images, height, cdf_y, cdf_img);
	hipLaunchKernelGGL(k_construct_cdf_img, dim3(1), dim3(64), 0, s, n_images, cdf_img);
}
void launch_compute_loss(hipStream_t s, const K3Args& a, uint32_t max_rays) {
	if (max_rays == 0) return;
	if (g_debug_flags & DBG_K3_THREAD_PER_RAY) hipLaunchKernelGGL(k_compute_loss, dim3(blocks(max_rays, 128)), dim3(128), 0, s, a);
	else if (!a.k3_scratch || !(g_debug_flags & DBG_K3_TWO_PASS)) {
		const bool err = a.error_map || a.cdf.x_cond_y || a.cdf.img;
		const dim3 g1(std::min<uint32_t>(blocks(max_rays, K3_RAYS_PER_BLOCK), 256u * 2u)), g2(std::min<uint32_t>(blocks(max_rays, K3_RAYS_PER_BLOCK * 2), 256u * 2u));
		const bool plain = a.train_mode == 0 && !(a.depth_lambda > 0.f) && !(g_debug_flags & DBG_K3_GENERIC) && a.cstride == 7;
		if (g_debug_flags & DBG_K3_ONE_RAY_PER_WAVE) { if (err) hipLaunchKernelGGL((k_compute_loss_v2<1, true, false>), g1, dim3(1024), 0, s, a); else hipLaunchKernelGGL((k_compute_loss_v2<1, false, false>), g1, dim3(1024), 0, s, a); }
		else if (err) hipLaunchKernelGGL((k_compute_loss_v2<2, true, false>), g2, dim3(1024), 0, s, a);
		else if (plain && a.ray_targets) {
			// production instance: 8 wavefronts (16 rays) per workgroup at 5 wavefronts per SIMD = two independent workgroups per CU (round 6, profiles/r06_ab_k3_workgroup_shape.txt;
			// 4-wavefront workgroups double the span atomics once more and are 20 - 35 us SLOWER: one counter word retires ~100 returning atomics per microsecond)
			static const bool wg16 = getenv("NGP_K3_WG16") && atoi(getenv("NGP_K3_WG16")) != 0; // the round-3..5 shape (ablation)
			if (wg16) hipLaunchKernelGGL((k_compute_loss_v2<2, false, true, true>), g2, dim3(1024), 0, s, a);
			// (79 registers since the coordinate lives in vector registers: three 8-wavefront workgroups per CU; two -- NGP_K3_OCC=5 in tools/batches/r06_p.sh -- measured the same)
			else hipLaunchKernelGGL((k_compute_loss_v2<2, false, true, true, 8, 6>), dim3(std::min<uint32_t>(blocks(max_rays, 16), 256u * 3u)), dim3(512), 0, s, a);
		}
		else if (plain) hipLaunchKernelGGL((k_compute_loss_v2<2, false, true>), g2, dim3(1024), 0, s, a);
		else hipLaunchKernelGGL((k_compute_loss_v2<2, false, false>), g2, dim3(1024), 0, s, a);
	}
	else {
		const uint32_t grid = k1_grid(max_rays, 8);
		float* rec = (float*)a.k3_scratch; uint64_t* partial = (uint64_t*)(rec + (size_t)max_rays * K3_REC); uint32_t* done = (uint32_t*)(partial + grid);
		hipLaunchKernelGGL((k_compute_loss_v3<0>), dim3(grid), dim3(256), 0, s, a, rec, partial, done);
		hipLaunchKernelGGL((k_compute_loss_v3<1>), dim3(grid), dim3(256), 0, s, a, rec, partial, done);
	}
}
void launch_fill_rollover(hipStream_t s, uint32_t n_elements, const uint32_t* n_input_ptr, float* coords, uint32_t cstride, ngp_half* dloss, uint32_t dstride,
		const uint32_t* publish_src2, uint32_t* publish_dst2, TrainCounters* ctl, uint32_t ctl_world_size, const float* publish_loss) {
	// grid-stride over the padding (typically 5 - 15 % of the batch); few workgroups keep the controller's ticket cheap
	hipLaunchKernelGGL(k_fill_rollover, dim3(std::min<uint32_t>(blocks(n_elements, 256), 128u)), dim3(256), 0, s, n_elements, n_input_ptr, coords, cstride, (__half*)dloss, dstride,
		publish_src2, publish_dst2, publish_loss, ctl, ctl_world_size, (g_debug_flags & DBG_K4_ZERO_PADDING) ? 1 : 0);
}
void launch_mark_untrained(hipStream_t s, uint32_t n, float* grid, uint32_t n_images, const ngp_image_meta* m, const ngp_xform* x, int clear) {
	hipLaunchKernelGGL(k_mark_untrained, dim3(blocks(n, 128)), dim3(128), 0, s, n, grid, n_images, m, x, clear);
}
void launch_generate_grid_samples(hipStream_t s, uint32_t n, ngp_pcg32 rng, const uint32_t* step_ptr, uint32_t step, ngp_aabb box, const float* grid_in,
		float* pos, uint32_t* idx, uint32_t n_cascades, float thresh) {
	if (n == 0) return;
	hipLaunchKernelGGL(k_generate_grid_samples, dim3(blocks(n, 256)), dim3(256), 0, s, n, rng, step_ptr, step, box, grid_in, pos, idx, n_cascades, thresh);
}
void launch_splat_grid_samples(hipStream_t s, uint32_t n, const uint32_t* idx, const ngp_half* out, uint32_t stride, float* grid, int act) {
	if (n == 0) return;
	hipLaunchKernelGGL(k_splat_grid_samples, dim3(blocks(n, 256)), dim3(256), 0, s, n, idx, (const __half*)out, stride, grid, act);
}
void launch_ema_grid_samples(hipStream_t s, uint32_t n, float decay, float* grid_out, const float* grid_in) {
	hipLaunchKernelGGL(k_ema_grid_samples, dim3(blocks(n, 256)), dim3(256), 0, s, n, decay, grid_out, grid_in);
}
void launch_grid_mean(hipStream_t s, const float* grid, float* partial256, float* mean_out) {
	hipLaunchKernelGGL(k_grid_mean_partial, dim3(MEAN_BLOCKS), dim3(256), 0, s, grid, partial256);
	hipLaunchKernelGGL(k_grid_mean_final, dim3(1), dim3(64), 0, s, partial256, mean_out);
}
void launch_grid_to_bitfield(hipStream_t s, const float* grid, uint32_t max_cascade, uint8_t* bitfield, const float* mean_ptr) {
	const uint32_t n = GRID_N_CELLS / 8 * N_CASCADES, nz = GRID_N_CELLS / 8 * (max_cascade + 1);
	hipLaunchKernelGGL(k_grid_to_bitfield, dim3(blocks(n, 256)), dim3(256), 0, s, n, nz, grid, bitfield, mean_ptr);
	for (uint32_t level = 1; level < N_CASCADES; ++level) {
		hipLaunchKernelGGL(k_bitfield_max_pool, dim3(blocks(GRID_N_CELLS / 64, 256)), dim3(256), 0, s, GRID_N_CELLS / 64,
			bitfield + grid_mip_offset(level - 1) / 8, bitfield + grid_mip_offset(level) / 8);
	}
}
void launch_update_counters(hipStream_t s, TrainCounters* c, uint32_t B, uint32_t world_size) { hipLaunchKernelGGL(k_update_counters, dim3(1), dim3(64), 0, s, c, B, world_size); }
void launch_clamp_compacted(hipStream_t s, TrainCounters* c, uint32_t B) { hipLaunchKernelGGL(k_clamp_compacted, dim3(1), dim3(64), 0, s, c, B); }

} // namespace ngp
