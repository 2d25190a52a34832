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
// BoundingBox::distance_sq, bounding_box.cuh:228-230
static __host__ __device__ __forceinline__ float bb_distance_sq(const SdfBvhNode& n, f3 p) {
	const float dx = fmaxf(fmaxf(n.bmin[0] - p.x, p.x - n.bmax[0]), 0.f), dy = fmaxf(fmaxf(n.bmin[1] - p.y, p.y - n.bmax[1]), 0.f), dz = fmaxf(fmaxf(n.bmin[2] - p.z, p.z - n.bmax[2]), 0.f);
	return dx * dx + dy * dy + dz * dz;
}
// BoundingBox::ray_intersect(...).x, bounding_box.cuh:163-210: entry parameter of the slab test (FLT_MAX on a miss; may be negative inside the box)
static __host__ __device__ __forceinline__ float bb_ray_entry(const SdfBvhNode& n, f3 o, f3 d) {
	float tmin = (n.bmin[0] - o.x) / d.x, tmax = (n.bmax[0] - o.x) / d.x;
	if (tmin > tmax) { const float t = tmin; tmin = tmax; tmax = t; }
	float tymin = (n.bmin[1] - o.y) / d.y, tymax = (n.bmax[1] - o.y) / d.y;
	if (tymin > tymax) { const float t = tymin; tymin = tymax; tymax = t; }
	if (tmin > tymax || tymin > tmax) return 3.402823466e+38f;
	if (tymin > tmin) tmin = tymin;
	if (tymax < tmax) tmax = tymax;
	float tzmin = (n.bmin[2] - o.z) / d.z, tzmax = (n.bmax[2] - o.z) / d.z;
	if (tzmin > tzmax) { const float t = tzmin; tzmin = tzmax; tzmax = t; }
	if (tmin > tzmax || tzmin > tmax) return 3.402823466e+38f;
	if (tzmin > tmin) tmin = tzmin;
	return tmin;
}

// closest_triangle(...).second: distance to the nearest triangle, bounded above by sqrt(max_distance_sq)
static __host__ __device__ float bvh_unsigned_distance(f3 p, const SdfBvhNode* __restrict__ nodes, const SdfTriangle* __restrict__ tris, float max_distance_sq) {
	// stack depth: ngp_sdf_create checks the tree's depth against the 64 entries.  An entry carries its box distance, so a node that became
	// farther than the best hit while it waited on the stack is dropped when it is popped.
	int stack[64]; float sdist[64]; int sp = 0;
	stack[sp] = 0; sdist[sp++] = 0.f;
	float best = max_distance_sq; bool found = false;
	while (sp > 0) {
		--sp;
		if (sdist[sp] > best) continue;
		const SdfBvhNode& n = nodes[stack[sp]];
		if (n.left < 0) {
			for (int i = -n.left - 1; i < -n.right - 1; ++i) { const float d = tri_distance_sq(tris[i], p); if (d <= best) { best = d; found = true; } }
		} else {
			const float dl = bb_distance_sq(nodes[n.left], p), dr = bb_distance_sq(nodes[n.right], p);
			// far child first onto the stack, so the near one is popped next
			if (dl <= dr) { if (dr <= best) { stack[sp] = n.right; sdist[sp++] = dr; } if (dl <= best) { stack[sp] = n.left; sdist[sp++] = dl; } }
			else { if (dl <= best) { stack[sp] = n.left; sdist[sp++] = dl; } if (dr <= best) { stack[sp] = n.right; sdist[sp++] = dr; } }
		}
	}
	return found ? sqrtf(best) : 0.0f; // "No closest triangle found": the reference returns 0 as well (triangle_bvh.cu:562-566)
}
// ray_intersect(...).first >= 0: is any triangle hit within SDF_MAX_DIST?
static __host__ __device__ bool bvh_ray_hits_anything(f3 o, f3 d, const SdfBvhNode* __restrict__ nodes, const SdfTriangle* __restrict__ tris) {
	int stack[64]; int sp = 0;
	stack[sp++] = 0;
	while (sp > 0) {
		const SdfBvhNode& n = nodes[stack[--sp]];
		if (n.left < 0) {
			for (int i = -n.left - 1; i < -n.right - 1; ++i) if (tri_ray_intersect(tris[i], o, d) < SDF_MAX_DIST) return true;
		} else {
			if (bb_ray_entry(nodes[n.right], o, d) < SDF_MAX_DIST) stack[sp++] = n.right; // depth checked at creation (ngp_sdf_create)
			if (bb_ray_entry(nodes[n.left], o, d) < SDF_MAX_DIST) stack[sp++] = n.left;
		}
	}
	return false;
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
// signed_distance_raystab, triangle_bvh.cu:631-650 with the per-element rng of signed_distance_raystab_kernel (:893-909)
static __host__ __device__ float bvh_signed_distance_raystab(uint32_t i, f3 p, const SdfBvhNode* __restrict__ nodes, const SdfTriangle* __restrict__ tris, float max_distance) {
	const float distance = bvh_unsigned_distance(p, nodes, tris, max_distance * max_distance);
	ngp_pcg32 dflt; dflt.state = 0x853c49e6748fea9bULL; dflt.inc = 0xda3e39cb94b95bdbULL; // default-constructed pcg32
	Rng rng(dflt);
	rng.advance((uint64_t)(i * 2u));
	const float ox = rng.next_float(), oy = rng.next_float();
	for (uint32_t k = 0; k < 32; ++k)
		if (!bvh_ray_hits_anything(p, fibonacci_dir32(k, ox, oy), nodes, tris)) return distance; // a stab ray escapes: outside
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
// signed_distance_gpu(n, Raystab, positions + n_exact, distances + n_exact, ..., use_existing_distances_as_upper_bounds)
__global__ void __launch_bounds__(256) k_sdf_signed_distance(uint32_t n, const float* __restrict__ positions, float* __restrict__ distances, const SdfBvhNode* __restrict__ nodes,
		const SdfTriangle* __restrict__ tris, int use_upper_bounds) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const f3 p = mk3(positions[(size_t)i * 3], positions[(size_t)i * 3 + 1], positions[(size_t)i * 3 + 2]);
	distances[i] = bvh_signed_distance_raystab(i, p, nodes, tris, use_upper_bounds ? distances[i] : SDF_MAX_DIST);
}
// the same function on the host (test hook; the product never calls it)
void host_sdf_signed_distance(uint32_t n, const float* positions, float* distances, const SdfBvhNode* nodes, const SdfTriangle* tris, int use_upper_bounds) {
	for (uint32_t i = 0; i < n; ++i) {
		const f3 p = mk3(positions[(size_t)i * 3], positions[(size_t)i * 3 + 1], positions[(size_t)i * 3 + 2]);
		distances[i] = bvh_signed_distance_raystab(i, p, nodes, tris, use_upper_bounds ? distances[i] : SDF_MAX_DIST);
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

void launch_sdf_generate_positions(hipStream_t s, const SdfSampleArgs& a) { if (a.n) hipLaunchKernelGGL(k_sdf_generate_positions, dim3((a.n + 255) / 256), dim3(256), 0, s, a); }
void launch_sdf_signed_distance(hipStream_t s, uint32_t n, const float* positions, float* distances, const SdfBvhNode* nodes, const SdfTriangle* tris, int use_upper_bounds) {
	if (n) hipLaunchKernelGGL(k_sdf_signed_distance, dim3((n + 255) / 256), dim3(256), 0, s, n, positions, distances, nodes, tris, use_upper_bounds);
}
void launch_sdf_compare_signs(hipStream_t s, uint32_t n, const float* ref, const ngp_half* model, uint32_t model_stride, uint32_t* counters) {
	if (n) hipLaunchKernelGGL(k_sdf_compare_signs, dim3((n + 255) / 256), dim3(256), 0, s, n, ref, (const __half*)model, model_stride, counters);
}

} // namespace ngp
