// oracle/_ref/libngpbvh_ref.so -- TEST INFRASTRUCTURE ONLY.  The reference's triangle BVH (src/triangle_bvh.cu: build with std::nth_element splits, the 4-wide
// traversal with its sorting network, closest_triangle, the 32-ray Fibonacci stab test, the watertight variant, ray tracing) and its triangle / box primitives
// (triangle.cuh, bounding_box.cuh) compiled for the CPU from where they lie, against oracle/ref_shim (tcnn's vector types, GPUMemory as a host vector, linear_kernel as a
// loop).  Nothing of the reference is copied: the translation unit below is the reference's file.  tests/test_ref_sdf.py compares the oracle's brute-force restatement
// (oracle/ora_sdf.hpp, what the HIP path is checked against on the GPU) with it.
#include <array>
#include <cassert>
#include <memory>
#include <../src/triangle_bvh.cu>
#include <neural-graphics-primitives/discrete_distribution.h>

using namespace ngp;
#define REF extern "C" __attribute__((visibility("default")))
static vec3 V3(const float* p) { return vec3{p[0], p[1], p[2]}; }
static Triangle TRI(const float* p) { Triangle t; t.a = V3(p); t.b = V3(p + 3); t.c = V3(p + 6); return t; }

struct RefBvh { std::vector<Triangle> tris; std::unique_ptr<TriangleBvh> bvh; };
// TriangleBvh::make() + build(triangles, 8) as Testbed::load_mesh does (testbed_sdf.cu:1419-1420); build() reorders the triangles in place
REF void* ref_bvh_create(const float* tris9, uint32_t n_tris, uint32_t n_primitives_per_leaf) {
	auto* h = new RefBvh; h->tris.resize(n_tris);
	for (uint32_t i = 0; i < n_tris; ++i) h->tris[i] = TRI(tris9 + (size_t)i * 9);
	h->bvh = TriangleBvh::make(); h->bvh->build(h->tris, n_primitives_per_leaf);
	return h;
}
REF void ref_bvh_destroy(void* h) { delete (RefBvh*)h; }
REF void ref_bvh_triangles(void* h, float* tris9_out) {
	auto& t = ((RefBvh*)h)->tris;
	for (size_t i = 0; i < t.size(); ++i) for (int k = 0; k < 3; ++k) { tris9_out[i * 9 + k] = t[i].a[k]; tris9_out[i * 9 + 3 + k] = t[i].b[k]; tris9_out[i * 9 + 6 + k] = t[i].c[k]; }
}
// signed_distance_gpu (triangle_bvh.cu:664-708) through the kernels at :879-923; mode 0 Watertight, 1 Raystab; distances in/out (upper bounds when the flag is set)
REF void ref_bvh_signed_distance(void* hh, int mode, const float* positions, uint32_t n, float* distances, int use_existing_distances_as_upper_bounds) {
	auto* h = (RefBvh*)hh;
	std::vector<vec3> p(n); for (uint32_t i = 0; i < n; ++i) p[i] = V3(positions + (size_t)i * 3);
	h->bvh->signed_distance_gpu(n, (EMeshSdfMode)mode, p.data(), distances, h->tris.data(), use_existing_distances_as_upper_bounds != 0, nullptr);
}
REF void ref_bvh_unsigned_distance(void* hh, const float* positions, uint32_t n, float* distances, int use_existing_distances_as_upper_bounds) {
	auto* h = (RefBvh*)hh;
	std::vector<vec3> p(n); for (uint32_t i = 0; i < n; ++i) p[i] = V3(positions + (size_t)i * 3);
	linear_kernel(unsigned_distance_kernel, 0, nullptr, n, (const vec3*)p.data(), (const TriangleBvhNode*)h->bvh->nodes_gpu(), (const Triangle*)h->tris.data(), distances, use_existing_distances_as_upper_bounds != 0);
}
// ray_trace_gpu (:710-726): positions advance to the hit (or by MAX_DIST), directions become the hit triangle's normal
REF void ref_bvh_ray_trace(void* hh, float* positions, float* directions, uint32_t n) {
	auto* h = (RefBvh*)hh;
	std::vector<vec3> p(n), d(n); for (uint32_t i = 0; i < n; ++i) { p[i] = V3(positions + (size_t)i * 3); d[i] = V3(directions + (size_t)i * 3); }
	h->bvh->ray_trace_gpu(n, p.data(), d.data(), h->tris.data(), nullptr);
	for (uint32_t i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) { positions[i * 3 + k] = p[i][k]; directions[i * 3 + k] = d[i][k]; }
}
REF int ref_bvh_touches_triangle(void* hh, const float* bmin, const float* bmax) { auto* h = (RefBvh*)hh; return h->bvh->touches_triangle(BoundingBox{V3(bmin), V3(bmax)}, h->tris.data()) ? 1 : 0; }

// triangle.cuh / bounding_box.cuh members, one call each
REF float ref_tri_distance_sq(const float* t9, const float* p) { return TRI(t9).distance_sq(V3(p)); }
REF float ref_tri_ray_intersect(const float* t9, const float* ro, const float* rd) { return TRI(t9).ray_intersect(V3(ro), V3(rd)); }
REF void ref_tri_closest_point(const float* t9, const float* p, float* out) { const vec3 r = TRI(t9).closest_point(V3(p)); for (int k = 0; k < 3; ++k) out[k] = r[k]; }
REF void ref_tri_sample_uniform_position(const float* t9, const float* sample2, float* out) { const vec3 r = TRI(t9).sample_uniform_position(vec2{sample2[0], sample2[1]}); for (int k = 0; k < 3; ++k) out[k] = r[k]; }
REF float ref_tri_surface_area(const float* t9) { return TRI(t9).surface_area(); }
REF void ref_tri_normal(const float* t9, float* out) { const vec3 r = TRI(t9).normal(); for (int k = 0; k < 3; ++k) out[k] = r[k]; }
REF int ref_tri_point_in_triangle(const float* t9, const float* p) { return TRI(t9).point_in_triangle(V3(p)) ? 1 : 0; }
REF float ref_box_distance_sq(const float* bmin, const float* bmax, const float* p) { return BoundingBox{V3(bmin), V3(bmax)}.distance_sq(V3(p)); }
REF float ref_box_signed_distance(const float* bmin, const float* bmax, const float* p) { return BoundingBox{V3(bmin), V3(bmax)}.signed_distance(V3(p)); }
REF int ref_box_intersects_triangle(const float* bmin, const float* bmax, const float* t9) { return BoundingBox{V3(bmin), V3(bmax)}.intersects(TRI(t9)) ? 1 : 0; }
REF void ref_fibonacci_dir32(uint32_t i, const float* offset2, float* out) { const vec3 r = fibonacci_dir<32>(i, vec2{offset2[0], offset2[1]}); for (int k = 0; k < 3; ++k) out[k] = r[k]; }
// DiscreteDistribution over the triangles' surface areas (discrete_distribution.h:21-42; testbed_sdf.cu:1428-1434) and its sampling
REF void ref_discrete_distribution_build(const float* weights, uint32_t n, float* cdf_out) {
	DiscreteDistribution d; d.build(std::vector<float>(weights, weights + n)); std::copy(d.cdf.begin(), d.cdf.end(), cdf_out);
}
REF uint32_t ref_discrete_distribution_sample(const float* cdf, uint32_t n, float val) { DiscreteDistribution d; d.cdf.assign(cdf, cdf + n); return d.sample(val); }
