// ngp_device.hpp -- device-side math of the NeRF hot path for gfx950 (wave64).
//
// Mirrors the *semantics* of the reference's device headers so that occupancy-grid indices and ray
// samples are bit-identical to the reference arithmetic (IEEE fp32, no FMA contraction in the
// translation units that include this for ray marching -- they are compiled with -ffp-contract=off):
//   nerf_device.cuh   (constants :25-43, activations :204-264, warps :266-315, grid index :317-341,
//                      stepping :360-460, occupancy skipping :462-495, sampling :553-599, losses :75-143)
//   common_device.cuh (sRGB :61-103, lens :268-345, uv_to_ray :413-490, pos_to_uv :527-577, read_rgba :846-872)
//   bounding_box.cuh  (:82-84, :173-227)   random_val.cuh (:60-291)   [tcnn] pcg32, morton3D
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "../../include/ngp_hip.h"

namespace ngp {

#define NGP_D __device__ __forceinline__
#define NGP_HD __host__ __device__ __forceinline__

struct f3 { float x, y, z; };
struct f2 { float x, y; };
struct f4 { float x, y, z, w; };

NGP_HD f3 mk3(float a, float b, float c) { f3 r; r.x = a; r.y = b; r.z = c; return r; }
NGP_HD f3 mk3(float a) { return mk3(a, a, a); }
NGP_HD f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
NGP_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
NGP_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
NGP_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
NGP_HD f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
NGP_HD f3 operator+(f3 a, float b) { return mk3(a.x + b, a.y + b, a.z + b); }
NGP_HD f3 operator-(f3 a, float b) { return mk3(a.x - b, a.y - b, a.z - b); }
NGP_HD f3 operator*(f3 a, float b) { return mk3(a.x * b, a.y * b, a.z * b); }
NGP_HD f3 operator*(float b, f3 a) { return mk3(b * a.x, b * a.y, b * a.z); }
NGP_HD f3 operator/(f3 a, float b) { return mk3(a.x / b, a.y / b, a.z / b); }
NGP_HD float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
NGP_HD float len3(f3 a) { return sqrtf(dot3(a, a)); }
NGP_HD f3 normalize3(f3 a) { return a / len3(a); }
NGP_HD float dist3(f3 a, f3 b) { return len3(a - b); }
NGP_HD float sgn(float x) { return copysignf(1.0f, x); }
NGP_HD float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
NGP_HD int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }
// tcnn's scalar clamp (vec.h: `a < b ? b : (c < a ? c : a)`): LOWER BOUND FIRST.  Same as clampi unless the bounds cross (lo > hi), where it returns lo.  The hot path has one
// call with crossed bounds: mip_from_dt (nerf_device.cuh:459).  Decision of round 4, DESIGN.md section 5 (vii).
NGP_HD int clampi_lower_first(int v, int lo, int hi) { return v < lo ? lo : (hi < v ? hi : v); }
NGP_HD float logisticf(float x) { return 1.0f / (1.0f + expf(-x)); }

struct M43 { f3 c[4]; };
NGP_HD M43 ldm43(const float* p) { M43 m; m.c[0] = ld3(p); m.c[1] = ld3(p + 3); m.c[2] = ld3(p + 6); m.c[3] = ld3(p + 9); return m; }
NGP_HD f3 mul3(const M43& m, f3 v) { return m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z; }

// ---- pcg32 -----------------------------------------------------------------------------------
struct Rng {
	uint64_t state, inc;
	NGP_HD Rng() {}
	NGP_HD explicit Rng(ngp_pcg32 p) : state(p.state), inc(p.inc) {}
	NGP_HD uint32_t next_uint() {
		uint64_t old = state;
		state = old * 0x5851f42d4c957f2dULL + inc;
		uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
		uint32_t rot = (uint32_t)(old >> 59u);
		return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
	}
	NGP_HD float next_float() {
		union { uint32_t u; float f; } x;
		x.u = (next_uint() >> 9) | 0x3f800000u;
		return x.f - 1.0f;
	}
	// pcg32::advance, the reference's loop.  (Round 5 tried compile-time tables of M^(2^b) and (1 + M + ... + M^(2^b - 1)) so that only the set bits of delta cost
	// multiplications -- exact, but the unrolled form is ~1 KiB of code per call site and K1's set-up / K3 got 5 - 7 us SLOWER on a slow-class box: these kernels pay for
	// code size, profiles/r05_ab_rng_tables_adam_cache.txt.)
	NGP_HD void advance(uint64_t delta) {
		uint64_t cur_mult = 0x5851f42d4c957f2dULL, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
		while (delta > 0) {
			if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
			cur_plus = (cur_mult + 1) * cur_plus;
			cur_mult *= cur_mult;
			delta >>= 1;
		}
		state = acc_mult * state + acc_plus;
	}
};

// ---- Morton ----------------------------------------------------------------------------------
NGP_HD uint32_t expand_bits(uint32_t v) {
	v = (v * 0x00010001u) & 0xFF0000FFu;
	v = (v * 0x00000101u) & 0x0F00F00Fu;
	v = (v * 0x00000011u) & 0xC30C30C3u;
	v = (v * 0x00000005u) & 0x49249249u;
	return v;
}
NGP_HD uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) { return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2); }
NGP_HD uint32_t morton3D_invert(uint32_t x) {
	x = x & 0x49249249;
	x = (x | (x >> 2)) & 0xc30c30c3;
	x = (x | (x >> 4)) & 0x0f00f00f;
	x = (x | (x >> 8)) & 0xff0000ff;
	x = (x | (x >> 16)) & 0x0000ffff;
	return x;
}

// ---- colour ----------------------------------------------------------------------------------
NGP_HD float srgb_to_linear(float s) { return s <= 0.04045f ? s / 12.92f : powf((s + 0.055f) / 1.055f, 2.4f); }
NGP_HD float linear_to_srgb(float l) { return l < 0.0031308f ? 12.92f * l : 1.055f * powf(l, 0.41666f) - 0.055f; }
NGP_HD f3 srgb_to_linear3(f3 v) { return mk3(srgb_to_linear(v.x), srgb_to_linear(v.y), srgb_to_linear(v.z)); }
NGP_HD f3 linear_to_srgb3(f3 v) { return mk3(linear_to_srgb(v.x), linear_to_srgb(v.y), linear_to_srgb(v.z)); }

// ---- AABB ------------------------------------------------------------------------------------
struct Box {
	f3 mn, mx;
	NGP_HD Box() {}
	NGP_HD explicit Box(const ngp_aabb& a) : mn(ld3(a.min)), mx(ld3(a.max)) {}
	NGP_HD f3 diag() const { return mx - mn; }
	NGP_HD f3 relative_pos(f3 p) const { return (p - mn) / diag(); }
	NGP_HD bool contains(f3 p) const { return p.x >= mn.x && p.x <= mx.x && p.y >= mn.y && p.y <= mx.y && p.z >= mn.z && p.z <= mx.z; }
	NGP_HD f2 ray_intersect(f3 pos, f3 dir) const {
		const float FMAX = 3.402823466e+38f;
		float tmin = (mn.x - pos.x) / dir.x, tmax = (mx.x - pos.x) / dir.x;
		if (tmin > tmax) { float t = tmin; tmin = tmax; tmax = t; }
		float tymin = (mn.y - pos.y) / dir.y, tymax = (mx.y - pos.y) / dir.y;
		if (tymin > tymax) { float t = tymin; tymin = tymax; tymax = t; }
		if (tmin > tymax || tymin > tmax) return {FMAX, FMAX};
		if (tymin > tmin) tmin = tymin;
		if (tymax < tmax) tmax = tymax;
		float tzmin = (mn.z - pos.z) / dir.z, tzmax = (mx.z - pos.z) / dir.z;
		if (tzmin > tzmax) { float t = tzmin; tzmin = tzmax; tzmax = t; }
		if (tmin > tzmax || tzmin > tmax) return {FMAX, FMAX};
		if (tzmin > tmin) tmin = tzmin;
		if (tzmax < tmax) tmax = tzmax;
		return {tmin, tmax};
	}
};

// ---- constants -------------------------------------------------------------------------------
constexpr uint32_t GRIDSIZE = 128;
constexpr uint32_t GRID_N_CELLS = GRIDSIZE * GRIDSIZE * GRIDSIZE;
constexpr uint32_t N_STEPS = 1024;
constexpr uint32_t N_CASCADES = 8;
constexpr float K_SQRT3 = 1.73205080757f;
constexpr float K_STEPSIZE = K_SQRT3 / N_STEPS;
constexpr float MIN_CONE_STEP = K_STEPSIZE;
constexpr float MAX_CONE_STEP = K_STEPSIZE * (1 << (N_CASCADES - 1)) * N_STEPS / GRIDSIZE;
constexpr uint32_t N_RANDOM_PER_RAY = 16;
constexpr float MIN_OPTICAL_THICKNESS = 0.01f;
constexpr float K_MAX_DEPTH = 16384.0f;

NGP_HD float act_rgb(float v, int a) {
	switch (a) {
		case NGP_ACT_NONE: return v;
		case NGP_ACT_RELU: return v > 0.0f ? v : 0.0f;
		case NGP_ACT_LOGISTIC: return logisticf(v);
		default: return expf(clampf(v, -10.0f, 10.0f));
	}
}
NGP_HD float act_rgb_d(float v, int a) {
	switch (a) {
		case NGP_ACT_NONE: return 1.0f;
		case NGP_ACT_RELU: return v > 0.0f ? 1.0f : 0.0f;
		case NGP_ACT_LOGISTIC: { float d = logisticf(v); return d * (1 - d); }
		default: return expf(clampf(v, -10.0f, 10.0f));
	}
}
NGP_HD float act_density(float v, int a) {
	switch (a) {
		case NGP_ACT_NONE: return v;
		case NGP_ACT_RELU: return v > 0.0f ? v : 0.0f;
		case NGP_ACT_LOGISTIC: return logisticf(v);
		default: return expf(v);
	}
}
NGP_HD float act_density_d(float v, int a) {
	switch (a) {
		case NGP_ACT_NONE: return 1.0f;
		case NGP_ACT_RELU: return v > 0.0f ? 1.0f : 0.0f;
		case NGP_ACT_LOGISTIC: { float d = logisticf(v); return d * (1 - d); }
		default: return expf(clampf(v, -15.0f, 15.0f));
	}
}

NGP_HD f3 warp_position(f3 p, const Box& b) { return b.relative_pos(p); }
NGP_HD f3 unwarp_position(f3 p, const Box& b) { return b.mn + p * b.diag(); }
NGP_HD f3 warp_direction(f3 d) { return (d + 1.0f) * 0.5f; }
NGP_HD float warp_dt(float dt) {
	float max_stepsize = MIN_CONE_STEP * (1 << (N_CASCADES - 1));
	return (dt - MIN_CONE_STEP) / (max_stepsize - MIN_CONE_STEP);
}
NGP_HD float unwarp_dt(float dt) {
	float max_stepsize = MIN_CONE_STEP * (1 << (N_CASCADES - 1));
	return dt * (max_stepsize - MIN_CONE_STEP) + MIN_CONE_STEP;
}

// ---- occupancy grid index math (bit-exact integer results) ---------------------------------------
NGP_HD uint32_t cascaded_grid_idx_at(f3 pos, uint32_t mip) {
	float mip_scale = scalbnf(1.0f, -(int)mip);
	pos = pos - mk3(0.5f);
	pos = pos * mip_scale;
	pos = pos + mk3(0.5f);
	int ix = (int)(pos.x * (float)GRIDSIZE), iy = (int)(pos.y * (float)GRIDSIZE), iz = (int)(pos.z * (float)GRIDSIZE);
	if (ix < 0 || ix >= (int)GRIDSIZE || iy < 0 || iy >= (int)GRIDSIZE || iz < 0 || iz >= (int)GRIDSIZE) return 0xFFFFFFFFu;
	return morton3D((uint32_t)ix, (uint32_t)iy, (uint32_t)iz);
}
NGP_HD uint32_t grid_mip_offset(uint32_t mip) { return GRID_N_CELLS * mip; }
// x-major copy of the bitfield (a derived acceleration structure of the ray marchers: the canonical grid stays in Morton order).
// Same cell arithmetic as cascaded_grid_idx_at, but the index costs 3 instead of ~40 integer operations and consecutive
// lattice points of a ray fall into the same bytes.
NGP_HD bool occupied_at_linear(f3 pos, const uint8_t* __restrict__ bitfield_linear, uint32_t mip) {
	float mip_scale = scalbnf(1.0f, -(int)mip);
	pos = pos - mk3(0.5f);
	pos = pos * mip_scale;
	pos = pos + mk3(0.5f);
	int ix = (int)(pos.x * (float)GRIDSIZE), iy = (int)(pos.y * (float)GRIDSIZE), iz = (int)(pos.z * (float)GRIDSIZE);
	if (ix < 0 || ix >= (int)GRIDSIZE || iy < 0 || iy >= (int)GRIDSIZE || iz < 0 || iz >= (int)GRIDSIZE) return false;
	const uint32_t idx = (uint32_t)ix + GRIDSIZE * ((uint32_t)iy + GRIDSIZE * (uint32_t)iz);
	return bitfield_linear[idx / 8 + grid_mip_offset(mip) / 8] & (1 << (idx % 8));
}
// The same test behind a conservative prefilter: `coarse` (LDS) holds one bit per 4x4x4 block of cells (OR of its 64 cells; 32^3 bits =
// 1024 words per cascade, built by k_build_coarse_bitfield from the linear copy), so a clear coarse bit proves the cell empty without a
// memory access.  Identical result to occupied_at_linear by construction.
constexpr uint32_t COARSE_SIZE = GRIDSIZE / 4, COARSE_WORDS = COARSE_SIZE * COARSE_SIZE * COARSE_SIZE / 32;
constexpr uint32_t MID_SIZE = GRIDSIZE / 2, MID_WORDS = MID_SIZE * MID_SIZE * MID_SIZE / 32; // one bit per 2x2x2 cells (k_build_mid_dilated_bitfield)
constexpr uint32_t k1_prefilter_words(uint32_t n_cascades) { return COARSE_WORDS * n_cascades + MID_WORDS; } // launch_build_linear_bitfield's `coarse` output
NGP_D bool occupied_at_linear_prefiltered(f3 pos, const uint8_t* __restrict__ bitfield_linear, const uint32_t* coarse, uint32_t mip) {
	float mip_scale = scalbnf(1.0f, -(int)mip);
	pos = pos - mk3(0.5f);
	pos = pos * mip_scale;
	pos = pos + mk3(0.5f);
	int ix = (int)(pos.x * (float)GRIDSIZE), iy = (int)(pos.y * (float)GRIDSIZE), iz = (int)(pos.z * (float)GRIDSIZE);
	if (ix < 0 || ix >= (int)GRIDSIZE || iy < 0 || iy >= (int)GRIDSIZE || iz < 0 || iz >= (int)GRIDSIZE) return false;
	const uint32_t cidx = ((uint32_t)ix >> 2) + COARSE_SIZE * (((uint32_t)iy >> 2) + COARSE_SIZE * ((uint32_t)iz >> 2));
	if (!((coarse[mip * COARSE_WORDS + (cidx >> 5)] >> (cidx & 31u)) & 1u)) return false;
	const uint32_t idx = (uint32_t)ix + GRIDSIZE * ((uint32_t)iy + GRIDSIZE * (uint32_t)iz);
	return bitfield_linear[idx / 8 + grid_mip_offset(mip) / 8] & (1 << (idx % 8));
}
NGP_HD bool occupied_at(f3 pos, const uint8_t* __restrict__ bitfield, uint32_t mip) {
	uint32_t idx = cascaded_grid_idx_at(pos, mip);
	if (idx == 0xFFFFFFFFu) return false;
	return bitfield[idx / 8 + grid_mip_offset(mip) / 8] & (1 << (idx % 8));
}

NGP_HD float distance_to_next_voxel(f3 pos, f3 dir, f3 idir, float res) {
	f3 p = res * (pos - 0.5f);
	float tx = (floorf(p.x + 0.5f + 0.5f * sgn(dir.x)) - p.x) * idir.x;
	float ty = (floorf(p.y + 0.5f + 0.5f * sgn(dir.y)) - p.y) * idir.y;
	float tz = (floorf(p.z + 0.5f + 0.5f * sgn(dir.z)) - p.z) * idir.z;
	float t = fminf(fminf(tx, ty), tz);
	return fmaxf(t / res, 0.0f);
}
NGP_HD float to_stepping_space(float t, float cone_angle) {
	if (cone_angle <= 1e-5f) return t / MIN_CONE_STEP;
	float log1p_c = logf(1.0f + cone_angle);
	float a = (logf(MIN_CONE_STEP) - logf(log1p_c)) / log1p_c;
	float b = (logf(MAX_CONE_STEP) - logf(log1p_c)) / log1p_c;
	float at = expf(a * log1p_c), bt = expf(b * log1p_c);
	if (t <= at) return (t - at) / MIN_CONE_STEP + a;
	else if (t <= bt) return logf(t) / log1p_c;
	else return (t - bt) / MAX_CONE_STEP + b;
}
NGP_HD float from_stepping_space(float n, float cone_angle) {
	if (cone_angle <= 1e-5f) return n * MIN_CONE_STEP;
	float log1p_c = logf(1.0f + cone_angle);
	float a = (logf(MIN_CONE_STEP) - logf(log1p_c)) / log1p_c;
	float b = (logf(MAX_CONE_STEP) - logf(log1p_c)) / log1p_c;
	float at = expf(a * log1p_c), bt = expf(b * log1p_c);
	if (n <= a) return (n - a) * MIN_CONE_STEP + at;
	else if (n <= b) return expf(n * log1p_c);
	else return (n - b) * MAX_CONE_STEP + bt;
}
NGP_HD float advance_n_steps(float t, float cone_angle, float n) { return from_stepping_space(to_stepping_space(t, cone_angle) + n, cone_angle); }
NGP_HD float calc_dt(float t, float cone_angle) { return advance_n_steps(t, cone_angle, 1.0f) - t; }
NGP_HD float advance_to_next_voxel(float t, float cone_angle, f3 pos, f3 dir, f3 idir, uint32_t mip) {
	float res = scalbnf((float)GRIDSIZE, -(int)mip);
	float t_target = t + distance_to_next_voxel(pos, dir, idir, res);
	t = to_stepping_space(t, cone_angle);
	t_target = to_stepping_space(t_target, cone_angle);
	return from_stepping_space(t + ceilf(fmaxf(t_target - t, 0.5f)), cone_angle);
}
NGP_HD uint32_t mip_from_pos(f3 pos, uint32_t max_cascade = N_CASCADES - 1) {
	int exponent;
	float maxval = fmaxf(fmaxf(fabsf(pos.x - 0.5f), fabsf(pos.y - 0.5f)), fabsf(pos.z - 0.5f));
	frexpf(maxval, &exponent);
	return (uint32_t)clampi(exponent + 1, 0, (int)max_cascade);
}
NGP_HD uint32_t mip_from_dt(float dt, f3 pos, uint32_t max_cascade = N_CASCADES - 1) {
	uint32_t mip = mip_from_pos(pos, max_cascade);
	dt *= 2 * GRIDSIZE;
	if (dt < 1.0f) return mip;
	int exponent;
	frexpf(dt, &exponent);
	// clamp(mip, exponent, max_cascade) with exponent > max_cascade for long steps (far samples of multi-cascade scenes): tcnn's clamp tests the lower bound first and
	// returns `exponent`, i.e. the march goes through the POOLED bitfield levels above max_cascade (update_density_grid_mean_and_bitfield pools all NERF_CASCADES levels) --
	// the behaviour of the pre-tcnn code as well (min(NERF_CASCADES() - 1, max(exponent, mip)): capped by the number of levels, not by the scene's max_cascade).
	// exponent <= 6 (calc_dt caps dt at MIN_CONE_STEP * 128), so the result is always a valid bitfield level.
	return (uint32_t)clampi_lower_first((int)mip, exponent, (int)max_cascade);
}
NGP_HD float skip_to_next_occupied(float t, float cone_angle, f3 o, f3 d, f3 idir, const uint8_t* __restrict__ grid,
		uint32_t min_mip, uint32_t max_mip, const Box& aabb) {
	while (true) {
		f3 pos = o + d * t;
		if (t >= K_MAX_DEPTH || !aabb.contains(pos)) return K_MAX_DEPTH;
		uint32_t mip = (uint32_t)clampi((int)mip_from_pos(pos), (int)min_mip, (int)max_mip);
		if (!grid || occupied_at(pos, grid, mip)) return t;
		while (mip < max_mip && !occupied_at(pos, grid, mip + 1)) ++mip;
		t = advance_to_next_voxel(t, cone_angle, pos, d, idir, mip);
	}
}

// ---- low-discrepancy sampler (Sobol dims 0/1 generated, Owen scrambling) ------------------------
NGP_HD uint32_t sobol01(uint32_t index, uint32_t dim) {
	uint32_t X = 0, v = 0x80000000u;
	for (uint32_t bit = 0; bit < 32; ++bit) {
		uint32_t dirn = (dim == 0) ? (0x80000000u >> bit) : v;
		if ((index >> bit) & 1u) X ^= dirn;
		v = v ^ (v >> 1);
	}
	return X;
}
NGP_HD uint32_t hash_combine(uint32_t seed, uint32_t v) { return seed ^ (v + (seed << 6) + (seed >> 2)); }
NGP_HD uint32_t reverse_bits32(uint32_t x) {
	x = (((x & 0xaaaaaaaa) >> 1) | ((x & 0x55555555) << 1));
	x = (((x & 0xcccccccc) >> 2) | ((x & 0x33333333) << 2));
	x = (((x & 0xf0f0f0f0) >> 4) | ((x & 0x0f0f0f0f) << 4));
	x = (((x & 0xff00ff00) >> 8) | ((x & 0x00ff00ff) << 8));
	return ((x >> 16) | (x << 16));
}
NGP_HD uint32_t lk_perm(uint32_t x, uint32_t seed) {
	x += seed; x ^= x * 0x6c50b47cu; x ^= x * 0xb82f1e52u; x ^= x * 0xc7afe638u; x ^= x * 0x8d22f6e6u; return x;
}
NGP_HD uint32_t owen_scramble(uint32_t x, uint32_t seed) { return reverse_bits32(lk_perm(reverse_bits32(x), seed)); }
NGP_HD float ld_random_val(uint32_t index, uint32_t seed, uint32_t dim = 0) {
	const float S = (float)(1.0 / 4294967296.0);
	index = owen_scramble(index, seed);
	return (float)owen_scramble(sobol01(index, dim), hash_combine(seed, dim)) * S;
}
NGP_HD f2 ld_random_pixel_offset(uint32_t spp) {
	float ax = ld_random_val(0, 0xdeadbeef, 0), ay = ld_random_val(0, 0xdeadbeef, 1);
	float bx = ld_random_val(spp, 0xdeadbeef, 0), by = ld_random_val(spp, 0xdeadbeef, 1);
	float ox = 0.5f - ax + bx, oy = 0.5f - ay + by;
	return {ox - floorf(ox), oy - floorf(oy)};
}

// ---- rolling shutter / motion blur: the training camera at a pixel's exposure time --------------------------------------------------------
// get_xform_given_rolling_shutter + camera_slerp, common_device.cuh:665-674: rotation = slerp(mat3(start), mat3(end), t), position = mix(start[3], end[3], t),
// t = rs.x + rs.y u + rs.z v + rs.w motionblur_time.  slerp(mat3, mat3, t) is tiny-cuda-nn's (vec.h, GLM-derived; the submodule is absent from the mount,
// restated from the published algorithm): matrix -> quaternion by the largest-diagonal rule, quaternion slerp along the short arc (linear when the
// quaternions almost coincide), quaternion -> matrix.  A frame WITHOUT motion data (start == end, rolling_shutter == 0) keeps its matrix untouched: the
// reference sends it through the same round trip at t = 0, which returns the matrix up to quaternion rounding that cannot be pinned without tcnn's sources.
struct Quat { float x, y, z, w; };
NGP_HD Quat quat_normalize(Quat q) { const float l = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w); return {q.x / l, q.y / l, q.z / l, q.w / l}; }
NGP_HD Quat quat_from_mat3(f3 c0, f3 c1, f3 c2) { // columns c0, c1, c2; m[col][row]
	const float fx = c0.x - c1.y - c2.z, fy = c1.y - c0.x - c2.z, fz = c2.z - c0.x - c1.y, fw = c0.x + c1.y + c2.z;
	int big = 0; float fb = fw;
	if (fx > fb) { fb = fx; big = 1; }
	if (fy > fb) { fb = fy; big = 2; }
	if (fz > fb) { fb = fz; big = 3; }
	const float bv = sqrtf(fb + 1.0f) * 0.5f, mult = 0.25f / bv;
	switch (big) {
		case 0: return {(c1.z - c2.y) * mult, (c2.x - c0.z) * mult, (c0.y - c1.x) * mult, bv};
		case 1: return {bv, (c0.y + c1.x) * mult, (c2.x + c0.z) * mult, (c1.z - c2.y) * mult};
		case 2: return {(c0.y + c1.x) * mult, bv, (c1.z + c2.y) * mult, (c2.x - c0.z) * mult};
		default: return {(c2.x + c0.z) * mult, (c1.z + c2.y) * mult, bv, (c0.y - c1.x) * mult};
	}
}
NGP_HD void mat3_from_quat(Quat q, f3& c0, f3& c1, f3& c2) {
	const float qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z, qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z, qwx = q.w * q.x, qwy = q.w * q.y, qwz = q.w * q.z;
	c0 = mk3(1.0f - 2.0f * (qyy + qzz), 2.0f * (qxy + qwz), 2.0f * (qxz - qwy));
	c1 = mk3(2.0f * (qxy - qwz), 1.0f - 2.0f * (qxx + qzz), 2.0f * (qyz + qwx));
	c2 = mk3(2.0f * (qxz + qwy), 2.0f * (qyz - qwx), 1.0f - 2.0f * (qxx + qyy));
}
NGP_HD Quat quat_slerp(Quat x, Quat y, float a) {
	Quat z = y;
	float cos_theta = x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
	if (cos_theta < 0.0f) { z = {-y.x, -y.y, -y.z, -y.w}; cos_theta = -cos_theta; } // the short way round
	if (cos_theta > 1.0f - 1.1920929e-07f) return {x.x + a * (z.x - x.x), x.y + a * (z.y - x.y), x.z + a * (z.z - x.z), x.w + a * (z.w - x.w)}; // mix
	const float angle = acosf(cos_theta), s0 = sinf((1.0f - a) * angle), s1 = sinf(a * angle), sd = sinf(angle);
	return {(s0 * x.x + s1 * z.x) / sd, (s0 * x.y + s1 * z.y) / sd, (s0 * x.z + s1 * z.z) / sd, (s0 * x.w + s1 * z.w) / sd};
}
NGP_HD M43 camera_slerp(const M43& a, const M43& b, float t) {
	const Quat q = quat_normalize(quat_slerp(quat_normalize(quat_from_mat3(a.c[0], a.c[1], a.c[2])), quat_normalize(quat_from_mat3(b.c[0], b.c[1], b.c[2])), t));
	M43 r;
	mat3_from_quat(q, r.c[0], r.c[1], r.c[2]);
	r.c[3] = a.c[3] * (1.0f - t) + b.c[3] * t; // mix
	return r;
}
NGP_HD M43 xform_given_rolling_shutter(const ngp_xform& X, const float rs[4], f2 uv, float motionblur_time) {
	bool moving = rs[0] != 0.f || rs[1] != 0.f || rs[2] != 0.f || rs[3] != 0.f;
	for (int k = 0; k < 12; ++k) moving |= X.start[k] != X.end[k];
	if (!moving) return ldm43(X.start);
	const float pixel_t = rs[0] + rs[1] * uv.x + rs[2] * uv.y + rs[3] * motionblur_time;
	return camera_slerp(ldm43(X.start), ldm43(X.end), pixel_t);
}

// ---- camera ----------------------------------------------------------------------------------
NGP_HD void opencv_distortion_delta(const float* p, float u, float v, float* du, float* dv) {
	const float k1 = p[0], k2 = p[1], p1 = p[2], p2 = p[3];
	const float u2 = u * u, uv = u * v, v2 = v * v, r2 = u2 + v2;
	const float radial = k1 * r2 + k2 * r2 * r2;
	*du = u * radial + 2.f * p1 * uv + p2 * (r2 + 2.f * u2);
	*dv = v * radial + 2.f * p2 * uv + p1 * (r2 + 2.f * v2);
}
// Inverse of x -> x + delta(x) by Newton's method with a central-difference Jacobian (the reference's iterative_lens_undistortion,
// common_device.cuh:307-345), one loop for every distortion function
template <typename Delta>
NGP_HD void newton_undistort(const Delta& delta, const float* params, float* u, float* v) {
	const float eps = 1.1920929e-07f;
	const float x00 = *u, x01 = *v;
	float x0 = *u, x1 = *v;
	for (uint32_t i = 0; i < 100; ++i) {
		const float step0 = fmaxf(eps, fabsf(1e-6f * x0)), step1 = fmaxf(eps, fabsf(1e-6f * x1));
		float dx0, dx1, b0x, b0y, f0x, f0y, b1x, b1y, f1x, f1y;
		delta(params, x0, x1, &dx0, &dx1);
		delta(params, x0 - step0, x1, &b0x, &b0y);
		delta(params, x0 + step0, x1, &f0x, &f0y);
		delta(params, x0, x1 - step1, &b1x, &b1y);
		delta(params, x0, x1 + step1, &f1x, &f1y);
		float J00 = 1 + (f0x - b0x) / (2 * step0), J10 = (f1x - b1x) / (2 * step1);
		float J01 = (f0y - b0y) / (2 * step0), J11 = 1 + (f1y - b1y) / (2 * step1);
		float r0 = x0 + dx0 - x00, r1 = x1 + dx1 - x01;
		float det = J00 * J11 - J10 * J01;
		float s0 = (J11 * r0 - J10 * r1) / det, s1 = (-J01 * r0 + J00 * r1) / det;
		x0 -= s0; x1 -= s1;
		if (s0 * s0 + s1 * s1 < 1e-10f) break;
	}
	*u = x0; *v = x1;
}
struct OpencvDelta { NGP_HD void operator()(const float* p, float u, float v, float* du, float* dv) const { opencv_distortion_delta(p, u, v, du, dv); } };
NGP_HD void opencv_undistort(const float* params, float* u, float* v) { newton_undistort(OpencvDelta{}, params, u, v); }
// ---- the other lens models of common_device.cuh:283-411 (own restatement; the Perspective / OpenCV arithmetic above is untouched) ----
NGP_HD void opencv_fisheye_distortion_delta(const float* p, float u, float v, float* du, float* dv) {
	const float r = sqrtf(u * u + v * v);
	*du = 0.f; *dv = 0.f;
	if (r > 2.220446049250313e-16f) { // (T)std::numeric_limits<double>::epsilon() in the reference
		const float theta = atanf(r), t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
		const float thetad = theta * (1.f + p[0] * t2 + p[1] * t4 + p[2] * t6 + p[3] * t8);
		*du = u * thetad / r - u;
		*dv = v * thetad / r - v;
	}
}
struct OpencvFisheyeDelta { NGP_HD void operator()(const float* p, float u, float v, float* du, float* dv) const { opencv_fisheye_distortion_delta(p, u, v, du, dv); } };
NGP_HD void opencv_fisheye_undistort(const float* params, float* u, float* v) { newton_undistort(OpencvFisheyeDelta{}, params, u, v); }
// f-theta: params = polynomial r0..r4 in the pixel radius, then the resolution the intrinsics refer to; (0,0,0) = no ray
NGP_HD f3 f_theta_direction(float u, float v, const float* params) {
	const float xpix = u * params[5], ypix = v * params[6];
	const float norm = sqrtf(xpix * xpix + ypix * ypix);
	const float alpha = params[0] + norm * (params[1] + norm * (params[2] + norm * (params[3] + norm * params[4])));
	float sa = sinf(alpha), ca = cosf(alpha);
	if (ca <= 1.17549435e-38f || norm == 0.f) return mk3(0.f);
	sa *= 1.f / norm;
	return mk3(sa * xpix, sa * ypix, ca);
}
constexpr float NGP_PI = 3.14159265358979323846f;
NGP_HD f3 latlong_to_dir(f2 uv) {
	const float theta = (uv.y - 0.5f) * NGP_PI, phi = (uv.x - 0.5f) * NGP_PI * 2.0f;
	const float st = sinf(theta), ct = cosf(theta), sp = sinf(phi), cp = cosf(phi);
	return mk3(sp * ct, st, cp * ct);
}
NGP_HD f3 equirectangular_to_dir(f2 uv) {
	const float ct = (uv.y - 0.5f) * 2.0f, st = sqrtf(fmaxf(1.0f - ct * ct, 0.0f)), phi = (uv.x - 0.5f) * NGP_PI * 2.0f;
	return mk3(sinf(phi) * st, ct, cosf(phi) * st);
}
NGP_HD f2 dir_to_latlong(f3 dir) { return {atan2f(dir.x, dir.z) / (NGP_PI * 2.0f) + 0.5f, asinf(dir.y) / NGP_PI + 0.5f}; }
NGP_HD f2 dir_to_equirectangular(f3 dir) { return {atan2f(dir.x, dir.z) / (NGP_PI * 2.0f) + 0.5f, dir.y / 2.0f + 0.5f}; }
NGP_HD bool lens_is_360(int lens_mode) { return lens_mode == NGP_LENS_LATLONG || lens_mode == NGP_LENS_EQUIRECTANGULAR; }

// uv_to_ray, common_device.cuh:413-490 with the default foveation (a clamp of uv to the unit square), without hidden-area mask / distortion map / aperture, parallax_shift = 0.
// Returns false where the lens has no ray for this uv (f-theta outside its field of view): the caller drops the ray / pixel.
NGP_HD bool uv_to_ray(f2 uv, const int32_t res[2], const float focal[2], const M43& cam, const float center[2], int lens_mode,
		const float* lens_params, float near_distance, f3& o, f3& d) {
	// foveation.warp(uv) (:429) with the default Foveation: clamp(x, 0, 1) * 1 + 0 per axis (common_device.cuh:215-224); the identity for every uv the samplers produce
	uv = {uv.x < 0.0f ? 0.0f : (uv.x > 1.0f ? 1.0f : uv.x), uv.y < 0.0f ? 0.0f : (uv.y > 1.0f ? 1.0f : uv.y)};
	f3 dir, head = mk3(0.f);
	if (lens_mode == NGP_LENS_PERSPECTIVE || lens_mode == NGP_LENS_OPENCV) { // the lenses of the BASELINE datasets: arithmetic unchanged since round 1
		dir = mk3((uv.x - center[0]) * (float)res[0] / focal[0], (uv.y - center[1]) * (float)res[1] / focal[1], 1.0f);
		if (lens_mode == NGP_LENS_OPENCV) opencv_undistort(lens_params, &dir.x, &dir.y);
		dir = mul3(cam, dir);
		f3 origin = cam.c[3];
		origin = origin + dir * near_distance;
		o = origin; d = dir;
		return true;
	}
	if (lens_mode == NGP_LENS_FTHETA) {
		dir = f_theta_direction(uv.x - center[0], uv.y - center[1], lens_params);
		if (dir.x == 0.f && dir.y == 0.f && dir.z == 0.f) { o = cam.c[3]; d = dir; return false; }
	} else if (lens_mode == NGP_LENS_LATLONG) {
		dir = latlong_to_dir(uv);
	} else if (lens_mode == NGP_LENS_EQUIRECTANGULAR) {
		dir = equirectangular_to_dir(uv);
	} else if (lens_mode == NGP_LENS_ORTHOGRAPHIC) {
		dir = mk3(0.f, 0.f, 1.f);
		head = mk3((uv.x - center[0]) * (float)res[0] / focal[0], (uv.y - center[1]) * (float)res[1] / focal[1], 0.0f);
	} else { // NGP_LENS_OPENCV_FISHEYE
		dir = mk3((uv.x - center[0]) * (float)res[0] / focal[0], (uv.y - center[1]) * (float)res[1] / focal[1], 1.0f);
		opencv_fisheye_undistort(lens_params, &dir.x, &dir.y);
	}
	dir = mul3(cam, dir);
	f3 origin = mul3(cam, head) + cam.c[3];
	origin = origin + dir * near_distance;
	o = origin; d = dir;
	return true;
}
// pos_to_uv, common_device.cuh:527-577 (f-theta has no forward mapping: treated like Perspective, as the reference's release build does).  The reference
// ends in foveation.unwarp(uv) (:576), and with the default Foveation of this path (testbed_nerf.cu:147) that clamps each axis to [0, 1] (common_device.cuh:226-235).
NGP_HD f2 pos_to_uv_before_foveation(f3 pos, const int32_t res[2], const float focal[2], const M43& cam, const float center[2], int lens_mode, const float* lens_params) {
	f3 dir = pos - cam.c[3];
	const f3 a = cam.c[0], b = cam.c[1], c = cam.c[2];
	float det = a.x * (b.y * c.z - c.y * b.z) - b.x * (a.y * c.z - c.y * a.z) + c.x * (a.y * b.z - b.y * a.z);
	float id = 1.0f / det;
	f3 r0 = mk3((b.y * c.z - c.y * b.z) * id, -(b.x * c.z - c.x * b.z) * id, (b.x * c.y - c.x * b.y) * id);
	f3 r1 = mk3(-(a.y * c.z - c.y * a.z) * id, (a.x * c.z - c.x * a.z) * id, -(a.x * c.y - c.x * a.y) * id);
	f3 r2 = mk3((a.y * b.z - b.y * a.z) * id, -(a.x * b.z - b.x * a.z) * id, (a.x * b.y - b.x * a.y) * id);
	dir = mk3(dot3(r0, dir), dot3(r1, dir), dot3(r2, dir));
	if (lens_mode == NGP_LENS_ORTHOGRAPHIC) return {dir.x * focal[0] / (float)res[0] + center[0], dir.y * focal[1] / (float)res[1] + center[1]};
	if (lens_is_360(lens_mode)) {
		dir = dir / sqrtf(dot3(dir, dir));
		return lens_mode == NGP_LENS_EQUIRECTANGULAR ? dir_to_equirectangular(dir) : dir_to_latlong(dir);
	}
	dir = dir / dir.z;
	float du = 0.f, dv = 0.f;
	if (lens_mode == NGP_LENS_OPENCV) opencv_distortion_delta(lens_params, dir.x, dir.y, &du, &dv);
	else if (lens_mode == NGP_LENS_OPENCV_FISHEYE) opencv_fisheye_distortion_delta(lens_params, dir.x, dir.y, &du, &dv);
	dir.x += du; dir.y += dv;
	return {dir.x * focal[0] / (float)res[0] + center[0], dir.y * focal[1] / (float)res[1] + center[1]};
}
NGP_HD f2 pos_to_uv(f3 pos, const int32_t res[2], const float focal[2], const M43& cam, const float center[2], int lens_mode, const float* lens_params) {
	const f2 uv = pos_to_uv_before_foveation(pos, res, focal, cam, center, lens_mode, lens_params);
	return {uv.x < 0.0f ? 0.0f : (uv.x > 1.0f ? 1.0f : uv.x), uv.y < 0.0f ? 0.0f : (uv.y > 1.0f ? 1.0f : uv.y)};
}

// tonemap(vec3, ETonemapCurve), render_buffer.cu:264-321: Identity 0, ACES 1 (Narkowicz fit incl. the 0.6 pre-exposure), Hable 2 (Uncharted-2
// operator with exposure bias 2 and white point 11.2 folded into the coefficients), Reinhard 3 (luminance based)
NGP_HD f3 tonemap_curve(f3 x, int curve) {
	if (curve == 0) return x;
	x = mk3(fmaxf(x.x, 0.f), fmaxf(x.y, 0.f), fmaxf(x.z, 0.f));
	float k0, k1, k2, k3, k4, k5;
	if (curve == 1) {
		k0 = 0.6f * 0.6f * 2.51f; k1 = 0.6f * 0.03f; k2 = 0.0f; k3 = 0.6f * 0.6f * 2.43f; k4 = 0.6f * 0.59f; k5 = 0.14f;
	} else if (curve == 2) {
		const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
		k0 = A * F - A * E; k1 = C * B * F - B * E; k2 = 0.0f; k3 = A * F; k4 = B * F; k5 = D * F * F;
		const float W = 11.2f;
		const float nom = k0 * (W * W) + k1 * W + k2, denom = k3 * (W * W) + k4 * W + k5;
		const float white_scale = denom / nom;
		k0 = 4.0f * k0 * white_scale; k1 = 2.0f * k1 * white_scale; k2 = k2 * white_scale; k3 = 4.0f * k3; k4 = 2.0f * k4;
	} else {
		const float Y = 0.2126f * x.x + 0.7152f * x.y + 0.0722f * x.z;
		return x * (1.f / (Y + 1.0f));
	}
	const f3 sq = x * x;
	const f3 nom = sq * k0 + x * k1 + mk3(k2), denom = sq * k3 + x * k4 + mk3(k5);
	return nom / denom;
}
// tonemap_kernel (render_buffer.cu:511-548) for one pixel of the accumulated linear frame: background behind the premultiplied colour
// (weight (1 - a) * bg.a), exposure, curve, optional linear -> sRGB.  `bg` is linear here (the host converts the sRGB background colour).
NGP_HD f4 tonemap_pixel(f4 c, float exposure_scale, float bg0, float bg1, float bg2, float bg3, int to_srgb, int curve) {
	const float weight = (1.f - c.w) * bg3;
	f3 rgb = mk3(c.x + bg0 * weight, c.y + bg1 * weight, c.z + bg2 * weight);
	rgb = tonemap_curve(rgb * exposure_scale, curve);
	if (to_srgb) rgb = mk3(linear_to_srgb(rgb.x), linear_to_srgb(rgb.y), linear_to_srgb(rgb.z));
	return {rgb.x, rgb.y, rgb.z, c.w + weight};
}

// read_depth, common_device.cuh:874-878
NGP_HD float read_depth(f2 uv, const int32_t res[2], const float* __restrict__ depth) {
	const int px = clampi((int)(uv.x * (float)res[0]), 0, res[0] - 1), py = clampi((int)(uv.y * (float)res[1]), 0, res[1] - 1);
	return depth[(size_t)px + (size_t)py * res[0]];
}
NGP_HD f4 read_rgba(f2 uv, const int32_t res[2], const void* __restrict__ pixels, int type) {
	int px = clampi((int)(uv.x * (float)res[0]), 0, res[0] - 1);
	int py = clampi((int)(uv.y * (float)res[1]), 0, res[1] - 1);
	size_t idx = (size_t)px + (size_t)py * res[0];
	if (type == NGP_IMAGE_BYTE) {
		uint32_t val = ((const uint32_t*)pixels)[idx];
		if (val == 0x00FF00FFu) return {-1.f, -1.f, -1.f, -1.f};
		f4 r = {((val & 0x000000FFu) >> 0) * (1.0f / 255.0f), ((val & 0x0000FF00u) >> 8) * (1.0f / 255.0f),
		        ((val & 0x00FF0000u) >> 16) * (1.0f / 255.0f), ((val & 0xFF000000u) >> 24) * (1.0f / 255.0f)};
		r.x = srgb_to_linear(r.x) * r.w; r.y = srgb_to_linear(r.y) * r.w; r.z = srgb_to_linear(r.z) * r.w;
		return r;
	} else if (type == NGP_IMAGE_HALF) {
		const __half* p = (const __half*)pixels + idx * 4;
		return {__half2float(p[0]), __half2float(p[1]), __half2float(p[2]), __half2float(p[3])};
	} else if (type == NGP_IMAGE_FLOAT) {
		const float* p = (const float*)pixels + idx * 4;
		return {p[0], p[1], p[2], p[3]};
	}
	return {5.0f, 0.0f, 0.0f, 1.0f};
}

// One component of read_rgba (c = 0..2 premultiplied colour, same arithmetic) + alpha, and the masked-pixel test alone: k1_setup's threads need one channel or only
// the mask, and the byte path costs one powf per channel.
NGP_HD bool read_rgba_masked(f2 uv, const int32_t res[2], const void* __restrict__ pixels, int type) {
	int px = clampi((int)(uv.x * (float)res[0]), 0, res[0] - 1);
	int py = clampi((int)(uv.y * (float)res[1]), 0, res[1] - 1);
	size_t idx = (size_t)px + (size_t)py * res[0];
	if (type == NGP_IMAGE_BYTE) return ((const uint32_t*)pixels)[idx] == 0x00FF00FFu;
	if (type == NGP_IMAGE_HALF) return __half2float(((const __half*)pixels)[idx * 4]) < 0.0f;
	if (type == NGP_IMAGE_FLOAT) return ((const float*)pixels)[idx * 4] < 0.0f;
	return false;
}
NGP_HD float read_rgba_channel(f2 uv, const int32_t res[2], const void* __restrict__ pixels, int type, uint32_t c, float& alpha) {
	int px = clampi((int)(uv.x * (float)res[0]), 0, res[0] - 1);
	int py = clampi((int)(uv.y * (float)res[1]), 0, res[1] - 1);
	size_t idx = (size_t)px + (size_t)py * res[0];
	if (type == NGP_IMAGE_BYTE) {
		uint32_t val = ((const uint32_t*)pixels)[idx];
		if (val == 0x00FF00FFu) { alpha = -1.f; return -1.f; }
		alpha = ((val & 0xFF000000u) >> 24) * (1.0f / 255.0f);
		return srgb_to_linear(((val >> (8u * c)) & 0xFFu) * (1.0f / 255.0f)) * alpha;
	} else if (type == NGP_IMAGE_HALF) {
		const __half* p = (const __half*)pixels + idx * 4;
		alpha = __half2float(p[3]); return __half2float(p[c]);
	} else if (type == NGP_IMAGE_FLOAT) {
		const float* p = (const float*)pixels + idx * 4;
		alpha = p[3]; return p[c];
	}
	alpha = 1.0f; return c == 0 ? 5.0f : 0.0f;
}

NGP_HD uint32_t image_idx(uint32_t base_idx, uint32_t n_rays, uint32_t n_images) { return ((base_idx * n_images) / n_rays) % n_images; }
NGP_HD f2 random_image_pos_training(Rng& rng, const int32_t res[2], bool snap) {
	f2 uv; uv.x = rng.next_float(); uv.y = rng.next_float();
	if (snap) {
		uv.x = ((float)clampi((int)(uv.x * (float)res[0]), 0, res[0] - 1) + 0.5f) / (float)res[0];
		uv.y = ((float)clampi((int)(uv.y * (float)res[1]), 0, res[1] - 1) + 0.5f) / (float)res[1];
	}
	return uv;
}

// ---- training pixels drawn in proportion to the accumulated error (nerf_device.cuh:497-599; the CDFs: testbed_nerf.cu:1530-1580, 2795-2855) ----
// binary_search, common.h:207-230: first element >= val, clamped to the last one
NGP_HD uint32_t cdf_lower_bound(float val, const float* data, uint32_t length) {
	if (length == 0) return 0;
	uint32_t first = 0, count = length;
	while (count > 0) {
		const uint32_t step = count / 2, it = first + step;
		if (data[it] < val) { first = it + 1; count -= step + 1; } else count = step;
	}
	return first < length - 1 ? first : length - 1;
}
// sample_cdf_2d, nerf_device.cuh:499-528: half of the draws stay uniform (UNIFORM_SAMPLING_FRACTION; *pdf is left untouched for them, as in the reference)
NGP_HD f2 sample_cdf_2d(f2 sample, uint32_t img, const int32_t res[2], const float* cdf_x_cond_y, const float* cdf_y, float* pdf) {
	const float UNIFORM_SAMPLING_FRACTION = 0.5f;
	if (sample.x < UNIFORM_SAMPLING_FRACTION) { sample.x /= UNIFORM_SAMPLING_FRACTION; return sample; }
	sample.x = (sample.x - UNIFORM_SAMPLING_FRACTION) / (1.0f - UNIFORM_SAMPLING_FRACTION);
	cdf_y += (size_t)img * res[1];
	const uint32_t y = cdf_lower_bound(sample.y, cdf_y, (uint32_t)res[1]);
	float prev = y > 0 ? cdf_y[y - 1] : 0.0f;
	const float pmf_y = cdf_y[y] - prev;
	sample.y = (sample.y - prev) / pmf_y;
	cdf_x_cond_y += (size_t)img * res[1] * res[0] + (size_t)y * res[0];
	const uint32_t x = cdf_lower_bound(sample.x, cdf_x_cond_y, (uint32_t)res[0]);
	prev = x > 0 ? cdf_x_cond_y[x - 1] : 0.0f;
	const float pmf_x = cdf_x_cond_y[x] - prev;
	sample.x = (sample.x - prev) / pmf_x;
	if (pdf) *pdf = pmf_x * pmf_y * (float)(res[0] * res[1]);
	f2 r; r.x = ((float)x + sample.x) / (float)res[0]; r.y = ((float)y + sample.y) / (float)res[1];
	return r;
}
// image_idx with an image CDF, nerf_device.cuh:578-591: one Owen-scrambled Sobol draw per ray index
NGP_HD uint32_t image_idx_cdf(uint32_t base_idx, uint32_t n_images, const float* cdf, float* pdf) {
	const float sample = ld_random_val(base_idx, 0xdeadbeefu);
	const uint32_t img = cdf_lower_bound(sample, cdf, n_images);
	if (pdf) { const float prev = img > 0 ? cdf[img - 1] : 0.0f; *pdf = (cdf[img] - prev) * (float)n_images; }
	return img;
}
// nerf_random_image_pos_training with the pixel CDFs, nerf_device.cuh:553-576
NGP_HD f2 random_image_pos_training_cdf(Rng& rng, const int32_t res[2], bool snap, const float* cdf_x_cond_y, const float* cdf_y, const int32_t cdf_res[2], uint32_t img, float* pdf) {
	f2 uv; uv.x = rng.next_float(); uv.y = rng.next_float();
	if (cdf_x_cond_y) uv = sample_cdf_2d(uv, img, cdf_res, cdf_x_cond_y, cdf_y, pdf);
	else if (pdf) *pdf = 1.0f;
	if (snap) {
		uv.x = ((float)clampi((int)(uv.x * (float)res[0]), 0, res[0] - 1) + 0.5f) / (float)res[0];
		uv.y = ((float)clampi((int)(uv.y * (float)res[1]), 0, res[1] - 1) + 0.5f) / (float)res[1];
	}
	return uv;
}

// losses: gradient/loss per channel
NGP_HD void loss_and_gradient(f3 target, f3 pred, int type, f3& loss, f3& grad) {
	const float t[3] = {target.x, target.y, target.z}, p[3] = {pred.x, pred.y, pred.z};
	float lo[3], gr[3];
	for (int k = 0; k < 3; ++k) {
		float df = p[k] - t[k];
		switch (type) {
			case NGP_LOSS_RELATIVE_L2: { float den = p[k] * p[k] + 1e-2f; lo[k] = df * df / den; gr[k] = 2.0f * df / den; break; }
			case NGP_LOSS_L1: { lo[k] = fabsf(df); gr[k] = copysignf(1.0f, df); break; }
			case NGP_LOSS_MAPE: { float den = fabsf(p[k]) + 1e-2f; lo[k] = fabsf(df) / den; gr[k] = copysignf(1.0f / den, df); break; }
			case NGP_LOSS_SMAPE: { float den = 0.5f * (fabsf(p[k]) + fabsf(t[k])) + 1e-2f; lo[k] = fabsf(df) / den; gr[k] = copysignf(1.0f / den, df); break; }
			case NGP_LOSS_HUBER: {
				const float alpha = 0.1f;
				float ad = fabsf(df), sq = 0.5f / alpha * df * df;
				lo[k] = (ad > alpha ? (ad - 0.5f * alpha) : sq) / 5.0f;
				gr[k] = (ad > alpha ? (df > 0 ? 1.0f : -1.0f) : (df / alpha)) / 5.0f;
				break; }
			case NGP_LOSS_LOGL1: { float dv = fabsf(df) + 1.0f; lo[k] = logf(dv); gr[k] = copysignf(1.0f / dv, df); break; }
			default: { lo[k] = df * df; gr[k] = 2.0f * df; break; }
		}
	}
	loss = mk3(lo[0], lo[1], lo[2]); grad = mk3(gr[0], gr[1], gr[2]);
}

} // namespace ngp
