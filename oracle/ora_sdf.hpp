// ORACLE -- TEST INFRASTRUCTURE ONLY (see ora_math.hpp header).  PARITY PINNED for the ground-truth signed distance against the reference's triangle BVH compiled for
// the CPU (oracle/_ref/libngpbvh_ref.so, tests/test_ref_sdf.py: |distance| bit for bit); PINNED for the sample generator from the random numbers on against the reference's kernels of
// testbed_sdf.cu (oracle/_ref/libngpimgsdf_ref.so); tcnn's random-number generators themselves are [tcnn].
// ora_sdf.hpp: the SDF primitive's data path restated on the CPU WITHOUT an acceleration structure: ground-truth signed distances by
// brute force over all triangles (Triangle::distance_sq / ray_intersect, triangle.cuh:87-129; signed_distance_raystab with 32 Fibonacci stab
// rays, triangle_bvh.cu:631-650, per-element rng :893-909; fibonacci_dir random_val.cuh:45-101) and generate_training_samples_sdf's
// positions / distance upper bounds (testbed_sdf.cu:1449-1544).  The BVH of the product only prunes: every value here must be reproduced
// exactly by any correct traversal.
#pragma once
#include "ora_math.hpp"

namespace ora {

struct Tri { float a[3], b[3], c[3]; };
inline vec3 cross3(vec3 a, vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float len2(vec3 a) { return dot(a, a); }
inline float sgnf(float x) { return std::copysign(1.0f, x); } // tcnn's sign(): copysign(1, x), never 0 [vec.h, from memory]
inline float clamp01(float x) { return std::fmin(std::fmax(x, 0.0f), 1.0f); }

inline float tri_distance_sq(const Tri& t, vec3 pos) {
	const vec3 a = V3(t.a), b = V3(t.b), c = V3(t.c);
	const vec3 v21 = b - a, p1 = pos - a, v32 = c - b, p2 = pos - b, v13 = a - c, p3 = pos - c;
	const vec3 nor = cross3(v21, v13);
	if (sgnf(dot(cross3(v21, nor), p1)) + sgnf(dot(cross3(v32, nor), p2)) + sgnf(dot(cross3(v13, nor), p3)) < 2.0f) {
		const float d1 = len2(v21 * clamp01(dot(v21, p1) / len2(v21)) - p1);
		const float d2 = len2(v32 * clamp01(dot(v32, p2) / len2(v32)) - p2);
		const float d3 = len2(v13 * clamp01(dot(v13, p3) / len2(v13)) - p3);
		return std::fmin(std::fmin(d1, d2), d3);
	}
	return dot(nor, p1) * dot(nor, p1) / len2(nor);
}
inline float tri_ray_intersect(const Tri& tr, vec3 ro, vec3 rd) {
	const vec3 a = V3(tr.a), v1v0 = V3(tr.b) - a, v2v0 = V3(tr.c) - a, rov0 = ro - a;
	const vec3 n = cross3(v1v0, v2v0), q = cross3(rov0, rd);
	const float d = 1.0f / dot(rd, n);
	const float u = d * -dot(q, v2v0), v = d * dot(q, v1v0);
	float t = d * -dot(n, rov0);
	if (u < 0.0f || u > 1.0f || v < 0.0f || (u + v) > 1.0f || t < 0.0f) t = std::numeric_limits<float>::max();
	return t;
}
inline vec3 fibonacci_dir32(uint32_t i, float ox, float oy) {
	const float epsilon = 1.33f, GOLDEN_RATIO = 1.6180339887498948482045868343656f;
	float px = (i + epsilon) / (32 - 1 + 2 * epsilon) + ox; px = px - std::floor(px);
	float py = i / GOLDEN_RATIO + oy; py = py - std::floor(py);
	const float cos_theta = -2.0f * px + 1.0f, phi = 2.0f * 3.14159265358979323846f * (py - 0.5f);
	const float sin_theta = std::sqrt(std::fmax(1.0f - cos_theta * cos_theta, 0.0f));
	return {sin_theta * std::cos(phi), sin_theta * std::sin(phi), cos_theta};
}
// distances_out[i] = signed distance of positions[i] (Raystab); max_dist[i] (optional) = upper bounds like the device path
inline void sdf_signed_distance_brute(const Tri* tris, uint32_t n_tris, const float* positions, uint32_t n, const float* max_dist, float* out) {
	#pragma omp parallel for schedule(dynamic, 16)
	for (int64_t ii = 0; ii < (int64_t)n; ++ii) {
		const uint32_t i = (uint32_t)ii;
		const vec3 p = V3(positions + (size_t)i * 3);
		const float md = max_dist ? max_dist[i] : 10.0f;
		float best = md * md; bool found = false;
		for (uint32_t k = 0; k < n_tris; ++k) { const float d = tri_distance_sq(tris[k], p); if (d <= best) { best = d; found = true; } }
		const float distance = found ? std::sqrt(best) : 0.0f;
		Pcg32 rng; // default-constructed pcg32 (state 0x853c49e6748fea9b, inc 0xda3e39cb94b95bdb)
		rng.state = 0x853c49e6748fea9bULL; rng.inc = 0xda3e39cb94b95bdbULL;
		rng.advance((int64_t)(uint32_t)(i * 2u));
		const float ox = rng.next_float(), oy = rng.next_float();
		bool inside = true;
		for (uint32_t r = 0; r < 32 && inside; ++r) {
			const vec3 d = fibonacci_dir32(r, ox, oy);
			bool hit = false;
			for (uint32_t k = 0; k < n_tris && !hit; ++k) hit = tri_ray_intersect(tris[k], p, d) < 10.0f;
			if (!hit) inside = false;
		}
		out[i] = inside ? -distance : distance;
	}
}
inline uint32_t cdf_search(float val, const float* data, uint32_t length) {
	uint32_t first = 0, count = length;
	while (count > 0) {
		const uint32_t step = count / 2, it = first + step;
		if (data[it] < val) { first = it + 1; count -= step + 1; } else count = step;
	}
	return std::min(first, length - 1);
}
inline float logistic_from_uniform(float x, float stddev) { // [tcnn generate_random_logistic, from memory]
	x = std::fmin(std::fmax(x, 1e-9f), 1.0f - 1e-9f);
	return -std::log(1.0f / x - 1.0f) * stddev * 0.551328895421792049f;
}
inline void sdf_generate_positions(const Tri* tris, uint32_t n_tris, const float* cdf, uint32_t n, uint32_t n_exact, uint32_t n_surface, const Pcg32& rng_in, float stddev,
		const ngp_aabb& box, float* positions, float* distances) {
	#pragma omp parallel for schedule(static)
	for (int64_t ii = 0; ii < (int64_t)n; ++ii) {
		const uint32_t i = (uint32_t)ii;
		Pcg32 rng = rng_in;
		rng.advance((int64_t)i * 3);
		vec3 s; s.x = rng.next_float(); s.y = rng.next_float(); s.z = rng.next_float();
		float dist = 0.f;
		if (i < n_surface) {
			const Tri& t = tris[cdf_search(s.x, cdf, n_tris)];
			const float sqrt_x = std::sqrt(s.y), f0 = 1.0f - sqrt_x, f1 = sqrt_x * (1.0f - s.z), f2 = sqrt_x * s.z;
			s = f0 * V3(t.a) + f1 * V3(t.b) + f2 * V3(t.c);
			if (i >= n_exact) {
				Pcg32 r2 = rng_in;
				r2.advance((int64_t)n * 3 + (int64_t)(i - n_exact) * 3);
				vec3 pert;
				pert.x = logistic_from_uniform(r2.next_float(), stddev); pert.y = logistic_from_uniform(r2.next_float(), stddev); pert.z = logistic_from_uniform(r2.next_float(), stddev);
				s = s + pert;
				dist = std::sqrt(len2(pert)) * 1.001f;
			}
		} else {
			const vec3 mn = V3(box.min), mx = V3(box.max);
			s = mn + s * (mx - mn);
			dist = std::sqrt(len2(mx - mn)) * 1.001f;
		}
		positions[(size_t)i * 3 + 0] = s.x; positions[(size_t)i * 3 + 1] = s.y; positions[(size_t)i * 3 + 2] = s.z;
		distances[i] = dist;
	}
}

} // namespace ora
