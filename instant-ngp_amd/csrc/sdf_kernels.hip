// sdf_kernels.hip -- gfx950 kernels of the SDF primitive's data path: training-sample generation on / around / off the mesh surface and
// the ground-truth signed distance through a triangle BVH (unsigned closest-triangle distance + 32 Fibonacci stab rays for the sign).
// Reference: src/testbed_sdf.cu (generate_training_samples_sdf :1449-1544, sample_uniform_on_triangle_kernel :702-716,
// perturb_sdf_samples :232-245, scale_to_aabb_kernel :531-538, compare_signs_kernel :540-567), include/.../triangle.cuh (Triangle::
// sample_uniform_position, distance_sq, ray_intersect), src/triangle_bvh.cu (closest_triangle :520-568, ray_intersect :475-518,
// signed_distance_raystab :631-650, kernels :893-909), random_val.cuh:45-101 (fibonacci_dir).
// MI355X notes: the BVH here is binary (built on the host, sdf part of ngp_api.hip): a wavefront's 64 query points are neighbours only
// by accident (i.i.d. samples), so traversal is divergent either way; the near child is visited first and a 64-entry stack lives in
// registers / scratch.  The stab rays only need to know WHETHER anything is hit (`.first < 0` in the reference), so their traversal
// stops at the first hit.  Compiled with -ffp-contract=off: the distances are compared against a brute-force oracle.
#include "ngp_device.hpp"
#include "ngp_kernels.hpp"
#include <hip/hip_fp16.h>

namespace ngp {

constexpr float SDF_MAX_DIST = 10.0f; // triangle_bvh.cu:42

static __host__ __device__ __forceinline__ f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static __host__ __device__ __forceinline__ float len2(f3 a) { return dot3(a, a); }
static __host__ __device__ __forceinline__ float sgnf(float x) { return copysignf(1.0f, x); } // tcnn's sign(): copysign(1, x), never 0
static __host__ __device__ __forceinline__ float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

// Triangle::distance_sq, triangle.cuh:108-129 (iq's triangle distance)
static __host__ __device__ __forceinline__ float tri_distance_sq(const SdfTriangle& t, f3 pos) {
	const f3 a = ld3(t.a), b = ld3(t.b), c = ld3(t.c);
	const f3 v21 = b - a, p1 = pos - a, v32 = c - b, p2 = pos - b, v13 = a - c, p3 = pos - c;
	const f3 nor = cross3(v21, v13);
	if (sgnf(dot3(cross3(v21, nor), p1)) + sgnf(dot3(cross3(v32, nor), p2)) + sgnf(dot3(cross3(v13, nor), p3)) < 2.0f) {
		const float d1 = len2(v21 * clamp01(dot3(v21, p1) / len2(v21)) - p1);
		const float d2 = len2(v32 * clamp01(dot3(v32, p2) / len2(v32)) - p2);
		const float d3 = len2(v13 * clamp01(dot3(v13, p3) / len2(v13)) - p3);
		return fminf(fminf(d1, d2), d3);
	}
	return dot3(nor, p1) * dot3(nor, p1) / len2(nor);
}
// Triangle::ray_intersect, triangle.cuh:87-101
static __host__ __device__ __forceinline__ float tri_ray_intersect(const SdfTriangle& tr, f3 ro, f3 rd) {
	const f3 a = ld3(tr.a), v1v0 = ld3(tr.b) - a, v2v0 = ld3(tr.c) - a, rov0 = ro - a;
	const f3 n = cross3(v1v0, v2v0), q = cross3(rov0, rd);
	const float d = 1.0f / dot3(rd, n);
	const float u = d * -dot3(q, v2v0), v = d * dot3(q, v1v0);
	float t = d * -dot3(n, rov0);
	if (u < 0.0f || u > 1.0f || v < 0.0f || (u + v) > 1.0f || t < 0.0f) t = 3.402823466e+38f;
	return t;
}
// ---- traversal (round 4) ----------------------------------------------------------------------------------------------------
// Both walks are chains of dependent memory round trips (the query points are i.i.d., so a wavefront's lanes diverge from the root on), and a batch has fewer points
// than the chip has lanes: what counts is the number of round trips in the slowest lane.  Hence
//   * 4-wide nodes: a node carries the boxes of its (up to) four children -- the binary median-split tree with every second level folded away -- in exactly one 128-byte
//     line (SdfBvhNode4, eight 16-byte loads issued together): half the depth of the binary tree, one round trip per inner step (rounds 2-3: node, then its two
//     children, then the next node);
//   * a child reference is either an inner node's index (>= 0) or a leaf ~((first triangle << 3) | count) with count <= SDF_LEAF_TRIS = 4: the leaf's triangles are
//     loaded in one batch (clamped indices, surplus results ignored) -- one round trip per leaf instead of one per triangle behind an early-out branch;
//   * "while-while": every lane first descends through inner nodes to its next leaf, then the wavefront's leaves are evaluated together -- the long leaf code runs once
//     per round instead of in every step in which some lane happens to be at a leaf;
//   * the reference stack (a depth-first walk keeps at most 3 * depth + 1 references of a 4-wide tree) lives in LDS on the device, one column per thread (entry i of
//     thread t at [i * 256 + t]: conflict-free, sized by the tree's depth at launch), instead of 512 bytes of scratch per thread; on the host (test hook) a local array.
constexpr int SDF_STACK_MAX = 48; // 3 * 15 + 1: 4-wide depth <= 15 (binary depth <= 29, ngp_sdf_create checks)
constexpr int SDF_LEAF_TRIS = 4;
constexpr int SDF_DONE = 0x7fffffff;
struct SdfLocalStack { int v[SDF_STACK_MAX]; __host__ __device__ __forceinline__ int get(int i) const { return v[i]; } __host__ __device__ __forceinline__ void set(int i, int x) { v[i] = x; } };
struct SdfLdsStack { int* col; __device__ __forceinline__ int get(int i) const { return col[i * 256]; } __device__ __forceinline__ void set(int i, int x) { col[i * 256] = x; } };

// the whole node (one 128-byte line) in eight 16-byte loads issued back to back
static __host__ __device__ __forceinline__ SdfBvhNode4 load_node(const SdfBvhNode4* __restrict__ nodes, int ref) {
	struct alignas(16) Q { uint32_t w[4]; };
	const Q* q = (const Q*)(nodes + ref);
	Q v[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) v[k] = q[k];
	SdfBvhNode4 n;
	__builtin_memcpy(&n, v, sizeof(n));
	return n;
}
// ascending sort of four (key, reference) pairs, branch-free (5 compare-exchanges)
static __host__ __device__ __forceinline__ void sort4(float (&k)[4], int (&r)[4]) {
#define SDF_CX(i, j) { const bool sw = k[j] < k[i]; const float ka = sw ? k[j] : k[i], kb = sw ? k[i] : k[j]; const int ra = sw ? r[j] : r[i], rb = sw ? r[i] : r[j]; k[i] = ka; k[j] = kb; r[i] = ra; r[j] = rb; }
	SDF_CX(0, 1) SDF_CX(2, 3) SDF_CX(0, 2) SDF_CX(1, 3) SDF_CX(1, 2)
#undef SDF_CX
}
// BoundingBox::distance_sq of child c
static __host__ __device__ __forceinline__ float child_distance_sq(const SdfBvhNode4& n, int c, f3 p) {
	const float dx = fmaxf(fmaxf(n.lo[0][c] - p.x, p.x - n.hi[0][c]), 0.f), dy = fmaxf(fmaxf(n.lo[1][c] - p.y, p.y - n.hi[1][c]), 0.f), dz = fmaxf(fmaxf(n.lo[2][c] - p.z, p.z - n.hi[2][c]), 0.f);
	return dx * dx + dy * dy + dz * dz; // (an empty slot's box is [+inf, -inf]: +inf)
}
// closest_triangle(...).second: distance to the nearest triangle, bounded above by sqrt(max_distance_sq).  Nearest child first; the result is the minimum of
// tri_distance_sq over every triangle whose boxes are not farther than the running minimum, i.e. independent of the visiting order and of the tree's shape.
template <class Stack>
static __host__ __device__ __forceinline__ float bvh_unsigned_distance(f3 p, const SdfBvhNode4* __restrict__ nodes, int root, const SdfTriangle* __restrict__ tris, float max_distance_sq, Stack& st) {
	int sp = 0, ref = root;
	float best = max_distance_sq; bool found = false;
	for (;;) {
		while (ref >= 0 && ref != SDF_DONE) {
			const SdfBvhNode4 n = load_node(nodes, ref);
			float d[4]; int r[4];
#pragma unroll
			for (int c = 0; c < 4; ++c) { d[c] = child_distance_sq(n, c, p); r[c] = n.ref[c]; }
			sort4(d, r);
			// farther children onto the stack, farthest first, so that the nearer ones are popped first (a reference that became farther than the best hit while it waited
			// costs one wasted step: its children are inside its box and fail the test)
			if (d[3] <= best) st.set(sp++, r[3]);
			if (d[2] <= best) st.set(sp++, r[2]);
			if (d[1] <= best) st.set(sp++, r[1]);
			ref = d[0] <= best ? r[0] : (sp > 0 ? st.get(--sp) : SDF_DONE);
		}
		if (ref == SDF_DONE) break;
		{
			const int first = (~ref) >> 3, cnt = (~ref) & 7;
			SdfTriangle T[SDF_LEAF_TRIS]; // all loads first: one round trip
#pragma unroll
			for (int k = 0; k < SDF_LEAF_TRIS; ++k) T[k] = tris[first + (k < cnt ? k : cnt - 1)];
#pragma unroll
			for (int k = 0; k < SDF_LEAF_TRIS; ++k) { const float dd = tri_distance_sq(T[k], p); const bool better = (k < cnt) & (dd <= best); best = better ? dd : best; found = found | better; }
		}
		if (sp == 0) break;
		ref = st.get(--sp);
	}
	return found ? sqrtf(best) : 0.0f; // "No closest triangle found": the reference returns 0 as well (triangle_bvh.cu:562-566)
}
// Slab test of a stab ray against child c with the reciprocal direction (six multiplications instead of the six IEEE divisions of BoundingBox::ray_intersect).  Only a
// conservative filter in front of the exact triangle tests: a box is accepted when [t_enter, t_exit] is non-empty after t_exit has been widened by 4 ulp (covers the
// rounding of both bounds), t_exit >= 0 (the triangle test rejects t < 0) and t_enter < SDF_MAX_DIST.  fminf / fmaxf drop the NaN of 0 * inf (origin on a face,
// axis-parallel ray); an empty slot ([+inf, -inf]) has t_exit < t_enter or NaNs only and is rejected.
static __host__ __device__ __forceinline__ float child_ray_entry(const SdfBvhNode4& n, int c, f3 o, f3 inv) {
	const float x1 = (n.lo[0][c] - o.x) * inv.x, x2 = (n.hi[0][c] - o.x) * inv.x;
	const float y1 = (n.lo[1][c] - o.y) * inv.y, y2 = (n.hi[1][c] - o.y) * inv.y;
	const float z1 = (n.lo[2][c] - o.z) * inv.z, z2 = (n.hi[2][c] - o.z) * inv.z;
	const float t_enter = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fminf(z1, z2));
	const float t_exit = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fmaxf(z1, z2));
	return (n.ref[c] != SDF_DONE && t_exit >= 0.0f && t_enter <= t_exit * 1.0000005f) ? t_enter : 3.402823466e+38f;
}
// ray_intersect(...).first >= 0: is any triangle hit within SDF_MAX_DIST?  Any-hit, so the order is free: the child the ray enters first is walked first (a hit
// ends the walk).  `stop` (device: an LDS flag shared by the 32 rays of one point, nullptr elsewhere) ends it from outside; the return value is then meaningless.
template <class Stack>
static __host__ __device__ __forceinline__ bool bvh_ray_hits_anything(f3 o, f3 d, const SdfBvhNode4* __restrict__ nodes, int root, const SdfTriangle* __restrict__ tris, Stack& st, const volatile int* stop) {
	const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
	int sp = 0, ref = root;
	for (;;) {
		if (stop && *stop) return true;
		for (uint32_t it = 1; ref >= 0 && ref != SDF_DONE; ++it) { // inner nodes, down to this lane's next leaf
			if (stop && (it & 7u) == 0u && *stop) return true; // (a long descent of a grazing ray: look at the flag every eighth step)
			const SdfBvhNode4 n = load_node(nodes, ref);
			float t[4]; int r[4];
#pragma unroll
			for (int c = 0; c < 4; ++c) { t[c] = child_ray_entry(n, c, o, inv); r[c] = n.ref[c]; }
			sort4(t, r);
			if (t[3] < SDF_MAX_DIST) st.set(sp++, r[3]);
			if (t[2] < SDF_MAX_DIST) st.set(sp++, r[2]);
			if (t[1] < SDF_MAX_DIST) st.set(sp++, r[1]);
			ref = t[0] < SDF_MAX_DIST ? r[0] : (sp > 0 ? st.get(--sp) : SDF_DONE);
		}
		if (ref == SDF_DONE) return false;
		{
			const int first = (~ref) >> 3, cnt = (~ref) & 7;
			SdfTriangle T[SDF_LEAF_TRIS]; // all loads first: one round trip
#pragma unroll
			for (int k = 0; k < SDF_LEAF_TRIS; ++k) T[k] = tris[first + (k < cnt ? k : cnt - 1)];
			bool hit = false;
#pragma unroll
			for (int k = 0; k < SDF_LEAF_TRIS; ++k) hit = hit | ((k < cnt) & (tri_ray_intersect(T[k], o, d) < SDF_MAX_DIST));
			if (hit) return true;
		}
		if (sp == 0) return false;
		ref = st.get(--sp);
	}
}
// cylindrical_to_dir / fibonacci_dir<32>, random_val.cuh:45-101
static __host__ __device__ __forceinline__ f3 fibonacci_dir32(uint32_t i, float ox, float oy) {
	const float epsilon = 1.33f; // N_DIRS = 32 >= 24
	const float GOLDEN_RATIO = 1.6180339887498948482045868343656f;
	float px = (i + epsilon) / (32 - 1 + 2 * epsilon) + ox; px = px - floorf(px);
	float py = i / GOLDEN_RATIO + oy; py = py - floorf(py);
	const float cos_theta = -2.0f * px + 1.0f, phi = 2.0f * 3.14159265358979323846f * (py - 0.5f);
	const float sin_theta = sqrtf(fmaxf(1.0f - cos_theta * cos_theta, 0.0f));
	float sp, cp; sincosf(phi, &sp, &cp);
	return mk3(sin_theta * cp, sin_theta * sp, cos_theta);
}
// the stab rays' random lattice offset: per-element rng of signed_distance_raystab_kernel (triangle_bvh.cu:893-909), default-constructed pcg32 advanced by 2 i
static __host__ __device__ __forceinline__ void stab_offset(uint32_t i, float& ox, float& oy) {
	ngp_pcg32 dflt; dflt.state = 0x853c49e6748fea9bULL; dflt.inc = 0xda3e39cb94b95bdbULL;
	Rng rng(dflt);
	rng.advance((uint64_t)(i * 2u));
	ox = rng.next_float(); oy = rng.next_float();
}
// signed_distance_raystab, triangle_bvh.cu:631-650, one point after the other (the host's test hook; the device splits the same functions over two kernels, below)
static float bvh_signed_distance_raystab_serial(uint32_t i, f3 p, const SdfBvhNode4* __restrict__ nodes, int root, const SdfTriangle* __restrict__ tris, float max_distance) {
	SdfLocalStack st;
	const float distance = bvh_unsigned_distance(p, nodes, root, tris, max_distance * max_distance, st);
	float ox, oy; stab_offset(i, ox, oy);
	for (uint32_t k = 0; k < 32; ++k)
		if (!bvh_ray_hits_anything(p, fibonacci_dir32(k, ox, oy), nodes, root, tris, st, nullptr)) return distance; // a stab ray escapes: outside
	return -distance;
}

// binary_search, common.h:207-230
static __device__ __forceinline__ uint32_t cdf_search(float val, const float* __restrict__ data, uint32_t length) {
	uint32_t first = 0, count = length;
	while (count > 0) {
		const uint32_t step = count / 2, it = first + step;
		if (data[it] < val) { first = it + 1; count -= step + 1; } else count = step;
	}
	return min(first, length - 1);
}
// [tcnn random.h generate_random_logistic, from memory]: logit of a uniform draw, scaled to the requested standard deviation
static __device__ __forceinline__ float logistic_from_uniform(float x, float stddev) {
	x = fminf(fmaxf(x, 1e-9f), 1.0f - 1e-9f);
	return -logf(1.0f / x - 1.0f) * stddev * 0.551328895421792049f;
}

// generate_training_samples_sdf, testbed_sdf.cu:1449-1544, positions and distance upper bounds.  Sample i: [0, n_exact) on the surface,
// [n_exact, n_surface) surface + logistic offset, [n_surface, n) uniform in the (inflated) box.  Random numbers: element e of the
// uniform block <- draw e of the pcg32 stream, element e of the perturbation block <- draw 3 n + e [tcnn, element<->draw mapping from memory].
__global__ void __launch_bounds__(256) k_sdf_generate_positions(SdfSampleArgs a) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= a.n) return;
	Rng rng(a.rng);
	rng.advance((uint64_t)i * 3ull);
	f3 s; s.x = rng.next_float(); s.y = rng.next_float(); s.z = rng.next_float();
	float dist = 0.f;
	if (i < a.n_surface) {
		const SdfTriangle& t = a.triangles[cdf_search(s.x, a.cdf, a.n_triangles)];
		const float sqrt_x = sqrtf(s.y), f0 = 1.0f - sqrt_x, f1 = sqrt_x * (1.0f - s.z), f2 = sqrt_x * s.z; // Triangle::sample_uniform_position(sample.yz())
		s = f0 * ld3(t.a) + f1 * ld3(t.b) + f2 * ld3(t.c);
		if (i >= a.n_exact) {
			Rng r2(a.rng);
			r2.advance((uint64_t)a.n * 3ull + (uint64_t)(i - a.n_exact) * 3ull);
			f3 pert;
			pert.x = logistic_from_uniform(r2.next_float(), a.stddev); pert.y = logistic_from_uniform(r2.next_float(), a.stddev); pert.z = logistic_from_uniform(r2.next_float(), a.stddev);
			s = s + pert;
			dist = sqrtf(len2(pert)) * 1.001f; // "Small epsilon above 1 to ensure a triangle is always found."
		}
	} else {
		const f3 mn = ld3(a.aabb.min), mx = ld3(a.aabb.max);
		s = mn + s * (mx - mn);                    // scale_to_aabb_kernel
		dist = sqrtf(len2(mx - mn)) * 1.001f;      // assign_float(length(aabb.diag()) * 1.001f)
	}
	a.positions[(size_t)i * 3 + 0] = s.x; a.positions[(size_t)i * 3 + 1] = s.y; a.positions[(size_t)i * 3 + 2] = s.z;
	a.distances[i] = dist;
}
// signed_distance_gpu(n, Raystab, positions + n_exact, distances + n_exact, ..., use_existing_distances_as_upper_bounds) in three launches:
//   k_sdf_distance_first_rays  one lane per (point, role), the role by blockIdx.y: 0 = unsigned distance, 1 .. first_rays = stab ray role - 1 of the Fibonacci lattice.  The
//                              walks are chains of dependent loads and a batch has fewer points than the chip has lanes, so the independent walks of a point run side by
//                              side instead of one after the other.  A ray that escapes marks its point: outside, settled (for an outside point a ray towards the open
//                              side usually does).  Measured (profiles/r04_f4_sdf_ground_truth.txt): the launch lasts as long as its slowest distance walk -- a uniform
//                              point's walk is ~125 inner steps + ~40 leaves (max 360 + 165) at ~2 us per dependent step.
//   k_sdf_compact_survivors    points none of whose first rays escaped -> survivor list, -distance provisionally
//   k_sdf_stab_rays            32 lanes per survivor, lane k = stab ray k (the first ones are known hits and idle).  Inside points -- every ray hits -- cost the longest
//                              of the remaining short any-hit walks instead of their sum; the first ray that finishes without a hit raises the point's LDS flag, which
//                              stops the others and flips the sign.
// The answer is "does ANY of the 32 rays escape", whatever the order, so the split returns what the serial loop of the reference returns.
// (Ablation only, NGP_SDF_POINT_ORDER: the production order is the batch's.)  Sort key of a query point: bit 30 = cost class (a point whose upper bound is large -- the batch's uniform points, bound = the box diagonal -- walks ~5 x more nodes than a
// near-surface point whose bound is its offset: the classes must not share wavefronts, a wavefront lasts as long as its slowest lane), bits 0..29 = Morton code of the position
// (10 bits per axis over the unit cube): neighbours along the curve walk nearly the same nodes, so a wavefront's lanes agree on most branches.
static __device__ __forceinline__ uint32_t spread10(uint32_t v) { v &= 1023u; v = (v | (v << 16)) & 0x030000FFu; v = (v | (v << 8)) & 0x0300F00Fu; v = (v | (v << 4)) & 0x030C30C3u; v = (v | (v << 2)) & 0x09249249u; return v; }
__global__ void __launch_bounds__(256) k_sdf_point_keys(uint32_t n, const float* __restrict__ positions, const float* __restrict__ upper_bounds /* nullptr: one class */, uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t q[3];
#pragma unroll
	for (int a = 0; a < 3; ++a) { const float x = positions[(size_t)i * 3 + a]; q[a] = (uint32_t)fminf(fmaxf(x * 1024.0f, 0.0f), 1023.0f); } // (NaN -> 0)
	const uint32_t cls = upper_bounds && upper_bounds[i] > 0.0625f ? 1u : 0u;
	keys[i] = (cls << 30) | spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
	idx[i] = i;
}
__global__ void __launch_bounds__(256) k_sdf_distance_first_rays(uint32_t n, const uint32_t* __restrict__ order, const float* __restrict__ positions, float* __restrict__ distances,
		const SdfBvhNode4* __restrict__ nodes, int root, const SdfTriangle* __restrict__ tris, int use_upper_bounds, uint32_t* __restrict__ escaped) {
	extern __shared__ int s_stack[]; // stack entries x 256 (launch_sdf_signed_distance)
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, role = blockIdx.y;
	if (t >= n) return;
	const uint32_t i = order[t];
	SdfLdsStack st; st.col = s_stack + threadIdx.x;
	const f3 p = mk3(positions[(size_t)i * 3], positions[(size_t)i * 3 + 1], positions[(size_t)i * 3 + 2]);
	if (role == 0) {
		const float max_distance = use_upper_bounds ? distances[i] : SDF_MAX_DIST;
		distances[i] = bvh_unsigned_distance(p, nodes, root, tris, max_distance * max_distance, st);
	} else {
		float ox, oy; stab_offset(i, ox, oy);
		// The point's mark doubles as the stop flag of its other first rays: an escaping ray that grazes the mesh walks hundreds of nodes (up to ~1200 on armadillo) --
		// once one ray of the point has escaped, the others' answers are not needed any more.
		if (!bvh_ray_hits_anything(p, fibonacci_dir32(role - 1u, ox, oy), nodes, root, tris, st, (const volatile int*)(escaped + i))) escaped[i] = 1u;
	}
}
__global__ void __launch_bounds__(256) k_sdf_compact_survivors(uint32_t n, const uint32_t* __restrict__ order, float* __restrict__ distances, uint32_t* __restrict__ escaped, uint32_t* __restrict__ survivors, uint32_t* __restrict__ n_survivors) {
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	bool survivor = false;
	uint32_t i = 0;
	if (t < n) {
		i = order[t];
		survivor = escaped[i] == 0u;
		escaped[i] = 0u; // clean for the next call
		if (survivor) distances[i] = -distances[i];
	}
	// wave-aggregated append (a wavefront's share of the list keeps the sorted order)
	const uint64_t m = __ballot(survivor);
	if (m) {
		const uint32_t lane = threadIdx.x & 63u;
		uint32_t base = 0;
		if (lane == (uint32_t)__ffsll((unsigned long long)m) - 1u) base = atomicAdd(n_survivors, (uint32_t)__popcll(m));
		base = __shfl(base, __ffsll((unsigned long long)m) - 1);
		if (survivor) survivors[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
	}
}
__global__ void __launch_bounds__(256, 5) k_sdf_stab_rays(const float* __restrict__ positions, float* __restrict__ distances, const SdfBvhNode4* __restrict__ nodes, int root,
		const SdfTriangle* __restrict__ tris, const uint32_t* __restrict__ survivors, const uint32_t* __restrict__ n_survivors, uint32_t first_rays) {
	extern __shared__ int s_stack[]; // stack entries x 256 (launch_sdf_signed_distance)
	__shared__ int s_escaped[8];
	const uint32_t n = *n_survivors;
	const uint32_t grp = threadIdx.x >> 5, k = threadIdx.x & 31u;
	if (k == 0) s_escaped[grp] = 0;
	__syncthreads();
	// grid-stride over the survivors, eight per workgroup and round; a group's 32 lanes are half a wavefront, so its flag is written and read by one wavefront only
	for (uint32_t g = blockIdx.x * 8u + grp; g < n; g += gridDim.x * 8u) {
		const uint32_t i = survivors[g];
		const f3 p = mk3(positions[(size_t)i * 3], positions[(size_t)i * 3 + 1], positions[(size_t)i * 3 + 2]);
		float ox, oy; stab_offset(i, ox, oy);
		SdfLdsStack st; st.col = s_stack + threadIdx.x;
		volatile int* flag = &s_escaped[grp];
		if (k >= first_rays && !bvh_ray_hits_anything(p, fibonacci_dir32(k, ox, oy), nodes, root, tris, st, flag)) {
			*flag = 1;
			distances[i] = fabsf(distances[i]); // every escaping lane writes the same value
		}
		__builtin_amdgcn_wave_barrier();
		if (k == 0) *flag = 0; // (wavefront-private flag: program order is enough)
		__builtin_amdgcn_wave_barrier();
	}
}
// ---- round 5: ONE persistent launch, lanes refilled from work lists -----------------------------------------------------------------------------------------------
// What the three launches above leave on the table: a wavefront lasts as long as its slowest lane, and the walk lengths have heavy tails (a near-surface point's distance walk
// 39 steps on average, a uniform point's 163 with a maximum of ~530; an escaping first ray: median 6 steps, but up to ~1200 when it grazes the mesh; an inside point's 32 rays
// ~35 each) -- most lanes of most wavefront-steps are idle, and a wavefront-step costs its ~250 VALU instructions whatever the number of live lanes.  Here the walks are ITEMS
// of two lists -- distance item t = point t; ray item r * n_pad + t = stab ray r of point t, ray-major, so that by the time ray r of a point comes up its earlier rays have
// usually decided it (an escaped point's remaining rays are dropped at the fetch: one coalesced load of 64 marks) -- and a lane whose walk has ended takes the next live item
// of its wavefront's reservation (SDF_FETCH_CHUNK items per returning atomic; candidates -> lanes by ballot ranks through 64 words of LDS).  Inner steps come in rounds of at
// most SDF_INNER_STEPS, then the wavefront's leaves are evaluated together (while-while with a bound: a grazing ray's long chain of inner nodes does not hold the others'
// leaves back).  A wavefront works on one list at a time (the two walks are different code) and moves to the other list when its own is drained.  k_sdf_finalize applies the
// sign once every walk has ended: negative iff none of the point's rays escaped -- what the reference's serial loop returns, whatever the order.
// Measured after it (profiles/r05_f4_sdf_half_nodes_ab.jsonl): the same walker on 64-byte nodes (half-precision boxes rounded outwards: boxes only prune, so the answers stay
// bit-identical; half the lines and half the load instructions per step) is NOT faster (batch 3.08 - 3.15 vs 3.05 - 3.10 ms, uniform points slower); neither are 2, 3 or 6
// workgroups per CU instead of 4, nor polling the marks less often (profiles/r05_f4_sdf_walk_occupancy.txt): the launch is as long as its longest dependent chain (one lane's
// ~500 - 1200 rounds), which is why the distance items are handed out long walks first.
constexpr uint32_t SDF_FETCH_CHUNK = 256; // items per reservation (n_pad is a multiple of it: a reservation holds one ray index)
constexpr uint32_t SDF_REFILL_MIN = 16;   // idle lanes that make a wavefront look for work before its next round
constexpr uint32_t SDF_INNER_STEPS = 8;   // inner-node steps per round
struct SdfWalkArgs {
	uint32_t n, n_pad, stack_entries, poll_mask, dist_reversed; int root, use_upper_bounds; // poll_mask: a walking ray looks at its point's mark in the rounds with (round & poll_mask) == poll_mask
	uint32_t m, groups, group_stride; // n = groups x m points: point j of group b lies at element b * group_stride + j of positions / distances, its lattice offset is stab_offsets[j] (round 6: the ground truth of several training batches in one launch)
	const float* positions; float* distances; const SdfBvhNode4* nodes; const SdfTriangle* tris;
	uint32_t* escaped; const float* stab_offsets; uint32_t* ctr; // ctr[0]: distance items reserved, ctr[1]: ray items reserved (zero on entry; k_sdf_finalize clears them)
};
__global__ void __launch_bounds__(256) k_sdf_stab_offsets(uint32_t n, float* __restrict__ offsets) { // stab_offset(i) depends on i alone: computed once per trainer
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { float ox, oy; stab_offset(i, ox, oy); offsets[2 * i] = ox; offsets[2 * i + 1] = oy; }
}
// the marks cross XCDs while the kernel runs: device-scope (write-through / L1-bypassing) accesses; a stale read only costs a walk whose answer is not needed
static __device__ __forceinline__ uint32_t sdf_mark_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ void sdf_mark_set(uint32_t* p) { __hip_atomic_store(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// One wavefront's reservation of a list and the hand-out of its live items to idle lanes.
struct SdfFetch {
	uint32_t lo = 0, hi = 0; bool exhausted = false; // wavefront-uniform
	// idle lanes receive the next live items (`item`, return value true).  Afterwards every idle lane has one, or the list is exhausted.
	template <bool RAYS>
	__device__ __forceinline__ bool fill(const SdfWalkArgs& a, bool idle, uint32_t* s_pick /* 64 words of this wavefront */, uint32_t& item) {
		const uint32_t lane = threadIdx.x & 63u; const uint64_t lt = (1ull << lane) - 1ull;
		const uint32_t total = RAYS ? 32u * a.n_pad : a.n;
		bool got = false;
		for (;;) {
			const uint64_t im = __ballot(idle && !got);
			if (!im) break;
			if (lo >= hi) {
				if (exhausted) break;
				uint32_t base = 0;
				if (lane == 0) base = atomicAdd(a.ctr + (RAYS ? 1 : 0), SDF_FETCH_CHUNK);
				base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
				if (base >= total) { exhausted = true; break; }
				lo = base; hi = min(base + SDF_FETCH_CHUNK, total);
			}
			const uint32_t cand = lo + lane;
			bool live = cand < hi;
			if (RAYS) { const uint32_t t = cand - (lo / a.n_pad) * a.n_pad; live = live && t < a.n; if (live) live = sdf_mark_load(a.escaped + t) == 0u; }
			const uint64_t lm = __ballot(live);
			const uint32_t n_live = (uint32_t)__popcll(lm), n_idle = (uint32_t)__popcll(im);
			if (live) s_pick[__popcll(lm & lt)] = cand;
			__builtin_amdgcn_wave_barrier();
			const uint32_t rank = (uint32_t)__popcll(im & lt);
			if (idle && !got && rank < n_live) { item = s_pick[rank]; got = true; }
			// live candidates nobody took stay in the reservation
			lo = n_live <= n_idle ? min(lo + 64u, hi) : (uint32_t)__builtin_amdgcn_readfirstlane((int)s_pick[n_idle - 1u]) + 1u;
			__builtin_amdgcn_wave_barrier();
		}
		return got;
	}
};
template <bool RAYS>
static __device__ __forceinline__ void sdf_walk_list(const SdfWalkArgs& a, SdfLdsStack& st, uint32_t* s_pick) {
	SdfFetch f;
	bool busy = false, found = false;
	uint32_t i = 0, round = 0; // i: the point's number in the launch (its mark), mi: its element in positions / distances
	size_t mi = 0;
	int ref = SDF_DONE, sp = 0;
	f3 p = mk3(0.f, 0.f, 0.f), d = mk3(0.f, 0.f, 1.f), inv = mk3(0.f, 0.f, 0.f);
	float best = 0.f;
	for (;;) {
		// 1. work for the idle lanes
		if (!f.exhausted && (uint32_t)__popcll(__ballot(!busy)) >= SDF_REFILL_MIN) {
			uint32_t item = 0;
			if (f.template fill<RAYS>(a, !busy, s_pick, item)) {
				busy = true; sp = 0; ref = a.root; round = 0;
				const uint32_t r = RAYS ? item / a.n_pad : 0u;
				// distance items from the END of the batch first: the launch is as long as its longest dependent chain, and the long distance walks (the uniform points, whose upper
				// bound is the box diagonal: 163 rounds on average, up to ~530) sit behind the near-surface points in a training batch -- they must not start last.  With several
				// batches in the launch the items interleave them (item -> batch item % groups, point m - 1 - item / groups), so that every batch's tail comes first.
				uint32_t b, j;
				if (RAYS) { i = item - r * a.n_pad; b = i / a.m; j = i - b * a.m; }
				else if (a.dist_reversed) { const uint32_t q = item / a.groups; b = item - q * a.groups; j = a.m - 1u - q; i = b * a.m + j; }
				else { i = item; b = i / a.m; j = i - b * a.m; }
				mi = (size_t)b * a.group_stride + j;
				p = mk3(a.positions[mi * 3], a.positions[mi * 3 + 1], a.positions[mi * 3 + 2]);
				if (RAYS) { d = fibonacci_dir32(r, a.stab_offsets[2 * j], a.stab_offsets[2 * j + 1]); inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); }
				else { const float md = a.use_upper_bounds ? a.distances[mi] : SDF_MAX_DIST; best = md * md; found = false; }
			}
		}
		if (!__ballot(busy)) break; // (the hand-out gives every idle lane an item unless the list is exhausted)
		// 2. another ray of the point has escaped meanwhile: this one's answer is not needed
		if (RAYS && busy && (round & a.poll_mask) == a.poll_mask && sdf_mark_load(a.escaped + i) != 0u) { busy = false; ref = SDF_DONE; }
		++round;
		// 3. inner nodes, towards each lane's next leaf
		for (uint32_t it = 0; it < SDF_INNER_STEPS && ref >= 0 && ref != SDF_DONE; ++it) {
			const SdfBvhNode4 n = load_node(a.nodes, ref);
			float t[4]; int r[4];
#pragma unroll
			for (int c = 0; c < 4; ++c) { t[c] = RAYS ? child_ray_entry(n, c, p, inv) : child_distance_sq(n, c, p); r[c] = n.ref[c]; }
			sort4(t, r);
			const float lim = RAYS ? SDF_MAX_DIST : best; // rays: entered before the far end; distance: not farther than the running minimum
			const bool k3 = RAYS ? t[3] < lim : t[3] <= lim, k2 = RAYS ? t[2] < lim : t[2] <= lim, k1 = RAYS ? t[1] < lim : t[1] <= lim, k0 = RAYS ? t[0] < lim : t[0] <= lim;
			if (k3) st.set(sp++, r[3]);
			if (k2) st.set(sp++, r[2]);
			if (k1) st.set(sp++, r[1]);
			ref = k0 ? r[0] : (sp > 0 ? st.get(--sp) : SDF_DONE);
		}
		// 4. the wavefront's leaves together; walks that have run out of nodes end
		if (busy && ref < 0) {
			const int first = (~ref) >> 3, cnt = (~ref) & 7;
			SdfTriangle T[SDF_LEAF_TRIS]; // all loads first: one round trip
#pragma unroll
			for (int k = 0; k < SDF_LEAF_TRIS; ++k) T[k] = a.tris[first + (k < cnt ? k : cnt - 1)];
			bool hit = false;
			if (RAYS) {
#pragma unroll
				for (int k = 0; k < SDF_LEAF_TRIS; ++k) hit = hit | ((k < cnt) & (tri_ray_intersect(T[k], p, d) < SDF_MAX_DIST));
			} else {
#pragma unroll
				for (int k = 0; k < SDF_LEAF_TRIS; ++k) { const float dd = tri_distance_sq(T[k], p); const bool better = (k < cnt) & (dd <= best); best = better ? dd : best; found = found | better; }
			}
			if (hit) { busy = false; ref = SDF_DONE; }        // any hit settles a stab ray
			else ref = sp > 0 ? st.get(--sp) : SDF_DONE;
		}
		if (busy && ref == SDF_DONE) {                         // nothing left to visit
			if (RAYS) sdf_mark_set(a.escaped + i);             // the ray escaped: the point is outside
			else a.distances[mi] = found ? sqrtf(best) : 0.0f; // "No closest triangle found": 0, as the reference (triangle_bvh.cu:562-566)
			busy = false;
		}
	}
}
// OCC = workgroups per CU the instance is compiled for: 4 (98 registers, no scratch) or 6 (80 registers, 32 bytes of scratch; six 26.6 KiB stack images are what a CU's LDS
// holds for armadillo's depth) -- the walks wait 75 % of their cycles (profiles/r05_pmc_sdf_walks.txt), so more of them in flight per SIMD is the lever that is left.
template <int OCC>
__global__ void __launch_bounds__(256, OCC) k_sdf_walks(SdfWalkArgs a) {
	extern __shared__ int s_stack[]; // [stack_entries][256] reference stacks | [4][64] hand-out words
	SdfLdsStack st; st.col = s_stack + threadIdx.x;
	uint32_t* s_pick = (uint32_t*)(s_stack + (size_t)a.stack_entries * 256u) + (threadIdx.x >> 6) * 64u;
	// the distance walks are ~1/8 of the steps: every eighth workgroup starts on them, everybody helps with the other list when its own is drained
	if ((blockIdx.x & 7u) == 0u) { sdf_walk_list<false>(a, st, s_pick); sdf_walk_list<true>(a, st, s_pick); }
	else { sdf_walk_list<true>(a, st, s_pick); sdf_walk_list<false>(a, st, s_pick); }
}
__global__ void __launch_bounds__(256) k_sdf_finalize(uint32_t n, uint32_t m, uint32_t group_stride, float* __restrict__ distances, uint32_t* __restrict__ escaped, uint32_t* __restrict__ ctr) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { const uint32_t b = i / m; const size_t mi = (size_t)b * group_stride + (i - b * m); if (escaped[i] == 0u) distances[mi] = -distances[mi]; escaped[i] = 0u; } // clean for the next call
	if (i == 0) { ctr[0] = 0u; ctr[1] = 0u; }
}
// the same functions on the host (test hook; the product never calls it)
void host_sdf_signed_distance(uint32_t n, const float* positions, float* distances, const SdfBvhNode4* nodes, int root, const SdfTriangle* tris, int use_upper_bounds) {
	for (uint32_t i = 0; i < n; ++i) {
		const f3 p = mk3(positions[(size_t)i * 3], positions[(size_t)i * 3 + 1], positions[(size_t)i * 3 + 2]);
		distances[i] = bvh_signed_distance_raystab_serial(i, p, nodes, root, tris, use_upper_bounds ? distances[i] : SDF_MAX_DIST);
	}
}
// compare_signs_kernel (no octree): counters[0..5] = ref inside / outside, model inside / outside, intersection, union
__global__ void __launch_bounds__(256) k_sdf_compare_signs(uint32_t n, const float* __restrict__ ref, const __half* __restrict__ model, uint32_t model_stride, uint32_t* __restrict__ counters) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	const bool in = i < n;
	const bool inside1 = in && ref[i] <= 0.f, inside2 = in && __half2float(model[(size_t)i * model_stride]) <= 0.f;
	const uint64_t m1 = __ballot(inside1), m2 = __ballot(inside2), mv = __ballot(in);
	if ((threadIdx.x & 63) == 0 && mv) {
		atomicAdd(&counters[0], (uint32_t)__popcll(m1)); atomicAdd(&counters[1], (uint32_t)__popcll(mv & ~m1));
		atomicAdd(&counters[2], (uint32_t)__popcll(m2)); atomicAdd(&counters[3], (uint32_t)__popcll(mv & ~m2));
		atomicAdd(&counters[4], (uint32_t)__popcll(m1 & m2)); atomicAdd(&counters[5], (uint32_t)__popcll(m1 | m2));
	}
}

void launch_sdf_stab_offsets(hipStream_t s, uint32_t n, float* offsets) { if (n) hipLaunchKernelGGL(k_sdf_stab_offsets, dim3((n + 255) / 256), dim3(256), 0, s, n, offsets); }
void launch_sdf_generate_positions(hipStream_t s, const SdfSampleArgs& a) { if (a.n) hipLaunchKernelGGL(k_sdf_generate_positions, dim3((a.n + 255) / 256), dim3(256), 0, s, a); }
int launch_sdf_signed_distance(hipStream_t s, uint32_t n, const float* positions, float* distances, const SdfBvhNode4* nodes, int root, uint32_t stack_entries, const SdfTriangle* tris,
		int use_upper_bounds, const SdfQueryScratch& q, uint32_t groups, uint32_t group_stride) {
	if (!n || !groups) return 0;
	static const bool persistent = !(getenv("NGP_SDF_PERSISTENT") && atoi(getenv("NGP_SDF_PERSISTENT")) == 0); // 0: the round-4 three-launch path (ablation)
	if (persistent || groups > 1) {
		SdfWalkArgs a;
		a.m = n; a.groups = groups; a.group_stride = groups > 1 ? group_stride : n; n *= groups; // n: points of the launch from here on
		a.n = n; a.n_pad = (n + SDF_FETCH_CHUNK - 1u) / SDF_FETCH_CHUNK * SDF_FETCH_CHUNK; a.stack_entries = std::min<uint32_t>(std::max<uint32_t>(stack_entries, 4u), (uint32_t)SDF_STACK_MAX);
		a.root = root; a.use_upper_bounds = use_upper_bounds; a.positions = positions; a.distances = distances; a.nodes = nodes; a.tris = tris;
		a.escaped = q.escaped; a.stab_offsets = q.stab_offsets; a.ctr = q.work_ctr;
		// how often a walking ray polls its point's mark (a device-scope load per lane: 64 uncoalesced memory-side requests per wavefront): every 2nd / 8th / ... round, or never
		// (0xffffffff: the marks are then only read at the hand-out).  Ablation knob NGP_SDF_POLL_MASK; measured: profiles/r05_f4_sdf_walk_occupancy.txt
		constexpr uint32_t poll_mask = 7u;
		a.poll_mask = poll_mask;
		constexpr bool dist_batch_order = false; // (distance items in batch order: +15 %, profiles/r05_pytest_sdf_dist_order.log)
		a.dist_reversed = dist_batch_order ? 0u : 1u;
		static const uint32_t occ = getenv("NGP_SDF_WALK_OCC") ? (uint32_t)std::min(std::max(atoi(getenv("NGP_SDF_WALK_OCC")), 1), 6) : 2u; // workgroups per CU (ablation knob; round 5 measured 2 / 3 / 4 as equal, profiles/r05_f4_sdf_walk_occupancy.txt; round 6: 2 is 4 - 6 % faster alone and leaves the CUs room for the training step that now runs beside the walks, profiles/r06_ab_sdf_prefetch.txt)
		const uint32_t lds = a.stack_entries * 256u * 4u + 4u * 64u * 4u, per_cu = std::max(1u, std::min(occ, (160u * 1024u) / (lds + 64u)));
		// a grid of resident workgroups (no more than there are reservations to make)
		const uint32_t grid = std::min<uint32_t>(256u * per_cu, (uint32_t)(((uint64_t)33u * a.n_pad / SDF_FETCH_CHUNK + 3u) / 4u) + 8u);
		if (per_cu > 4u) hipLaunchKernelGGL(k_sdf_walks<6>, dim3(grid), dim3(256), lds, s, a); else hipLaunchKernelGGL(k_sdf_walks<4>, dim3(grid), dim3(256), lds, s, a);
		hipLaunchKernelGGL(k_sdf_finalize, dim3((n + 255) / 256), dim3(256), 0, s, n, a.m, a.group_stride, distances, q.escaped, q.work_ctr);
		return 0;
	}
	constexpr uint32_t first_rays = 4u; // stab rays walked next to the distance query before a point goes to the 32-lane kernel
	// 0: batch order (production); 1: (cost class, Morton) order; 2: Morton order, one class.  Ablation knob: coherent wavefronts measured SLOWER (batch 3.5 -> 4.3 ms, 2^18 uniform points
	// 3.2 -> 5.6 / 6.2 ms, profiles/r04_f4_sdf_ground_truth.txt v7) -- sorted points make concurrently running wavefronts hammer the same lines.
	constexpr int point_order = 0; // the batch's order (Morton / cost-class orders measured slower in round 4)
	const uint32_t entries = std::min<uint32_t>(std::max<uint32_t>(stack_entries, 4u), (uint32_t)SDF_STACK_MAX), lds = entries * 256u * 4u; // (<= 48 KiB)
	(void)hipMemsetAsync(q.n_survivors, 0, 4, s); // (`escaped` is zero: ngp_sdf_create clears it, k_sdf_compact_survivors leaves it clean)
	hipLaunchKernelGGL(k_sdf_point_keys, dim3((n + 255) / 256), dim3(256), 0, s, n, positions, (use_upper_bounds && point_order == 1) ? distances : nullptr, q.keys, point_order ? q.idx : q.order);
	if (point_order && sdf_point_sort(s, q.sort_temp, q.sort_temp_bytes, q.keys, q.keys_sorted, q.idx, q.order, n)) return 1;
	hipLaunchKernelGGL(k_sdf_distance_first_rays, dim3((n + 255) / 256, 1 + first_rays), dim3(256), lds, s, n, q.order, positions, distances, nodes, root, tris, use_upper_bounds, q.escaped);
	hipLaunchKernelGGL(k_sdf_compact_survivors, dim3((n + 255) / 256), dim3(256), 0, s, n, q.order, distances, q.escaped, q.survivors, q.n_survivors);
	// a grid of resident workgroups walks the survivor list (its length is only known on the device): as many as the stacks let a CU hold
	const uint32_t per_cu = std::max(1u, std::min(5u, (160u * 1024u) / (lds + 64u)));
	hipLaunchKernelGGL(k_sdf_stab_rays, dim3(std::min<uint32_t>((n + 7) / 8, 256u * per_cu)), dim3(256), lds, s, positions, distances, nodes, root, tris, q.survivors, q.n_survivors, first_rays);
	return 0;
}
void launch_sdf_compare_signs(hipStream_t s, uint32_t n, const float* ref, const ngp_half* model, uint32_t model_stride, uint32_t* counters) {
	if (n) hipLaunchKernelGGL(k_sdf_compare_signs, dim3((n + 255) / 256), dim3(256), 0, s, n, ref, (const __half*)model, model_stride, counters);
}

} // namespace ngp
