// ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into, imported by or executed from the product
// path (libngp_hip.so / pyngp).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// may use it, and only as the checker.
//
// CPU restatement (scalar C++17) of the arithmetic of the instant-ngp NeRF hot path.
// PARITY UNPINNED: the reference ships no tests / golden vectors for this path and the
// tiny-cuda-nn submodule (all network / encoding / optimizer arithmetic) is absent from
// /root/reference, so every "[tcnn]" function below restates the PUBLISHED algorithm of
// github.com/NVlabs/tiny-cuda-nn (API generation >= 2.0, pinned SHA unknown) and is anchored on
// the reference's call sites.  Independent pins used by tests/: the pcg32 reference stream
// (pcg-c-basic demo values), closed-form spherical harmonics (scipy), finite-difference gradients,
// and brute-force definitions of Morton / occupancy indices.  Two results the reference itself logged (notebooks/instant_ngp.ipynb: the hash
// grid's parameter count for a 16 x 2 configuration, the fox cameras' bounding box after loading) are pinned in tests/test_oracle_pins.py and
// tests/test_host_logic.py; nothing else is.
// Round 3: the reference's own code, compiled for the CPU from where it lies against a stand-in for tcnn's vector types (oracle/ref_shim, oracle/Makefile `ref`), now
// pins what is IN the reference's repository bit for bit: the device headers (tests/test_ref_device.py), the kernels of testbed_nerf.cu -- K1, K3, occupancy grid, error-map
// CDFs (tests/test_ref_kernels.py) -- and the triangle BVH (tests/test_ref_sdf.py).  Still unpinned: everything marked [tcnn] (hash grid, MLPs, optimizer, pcg32, slerp).
//
// ora_math.hpp: half, pcg32, vec3, Morton codes, colour transfer, AABB, ray stepping and
// occupancy-grid index math.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <limits>

#include "../include/ngp_hip.h"  // POD interface structs only (no code)

namespace ora {

// ---------------------------------------------------------------------------------------------
// IEEE binary16 <-> binary32, round-to-nearest-even (what `(__half)x` does on the device).
// ---------------------------------------------------------------------------------------------
inline uint16_t f2h(float f) {
	uint32_t x; std::memcpy(&x, &f, 4);
	uint32_t sign = (x >> 16) & 0x8000u;
	uint32_t abs = x & 0x7fffffffu;
	if (abs >= 0x7f800000u) { // inf / nan
		return (uint16_t)(sign | 0x7c00u | ((abs > 0x7f800000u) ? 0x200u : 0u));
	}
	if (abs >= 0x477ff000u) { // >= 65520 -> inf
		return (uint16_t)(sign | 0x7c00u);
	}
	if (abs < 0x33000001u) { // < 2^-25 (or exactly 2^-25 -> ties to even = 0)
		return (uint16_t)sign;
	}
	int e = (int)(abs >> 23) - 127;
	uint32_t m = (abs & 0x7fffffu) | 0x800000u; // 24-bit mantissa with hidden one
	int shift;
	uint32_t base;
	if (e < -14) { // subnormal half: value = m * 2^(e-23); unit = 2^-24
		shift = (-14 - e) + 13;
		base = 0;
	} else {
		shift = 13;
		base = (uint32_t)(e + 15) << 10;
		m &= 0x7fffffu;
	}
	uint32_t q = m >> shift;
	uint32_t rem = m & ((1u << shift) - 1u);
	uint32_t half = 1u << (shift - 1);
	if (rem > half || (rem == half && (q & 1u))) ++q;
	return (uint16_t)(sign | (base + q)); // carry into the exponent is the correct rounding
}

inline float h2f(uint16_t h) {
	uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
	uint32_t e = (h >> 10) & 0x1fu;
	uint32_t m = h & 0x3ffu;
	uint32_t x;
	if (e == 0) {
		if (m == 0) { x = sign; }
		else {
			int s = 0;
			while (!(m & 0x400u)) { m <<= 1; ++s; }
			m &= 0x3ffu;
			x = sign | ((uint32_t)(127 - 15 - s + 1) << 23) | (m << 13);
		}
	} else if (e == 31) {
		x = sign | 0x7f800000u | (m << 13);
	} else {
		x = sign | ((e + 112u) << 23) | (m << 13);
	}
	float f; std::memcpy(&f, &x, 4);
	return f;
}

// round a float through half precision
inline float rh(float f) { return h2f(f2h(f)); }

// double -> half with ONE rounding (round-to-nearest-even on the exact double value).
inline uint16_t d2h(double r) {
	double a = std::fabs(r);
	if (!(a == a)) return 0x7e00u;
	if (a == 0.0) return std::signbit(r) ? 0x8000u : 0u;
	if (a >= 65520.0) return (uint16_t)((std::signbit(r) ? 0x8000u : 0u) | 0x7c00u);
	int e;
	std::frexp(a, &e);                        // a = m * 2^e, m in [0.5, 1)
	int ulp_exp = std::max(e - 1, -14) - 10;  // exponent of one half-precision ulp at this magnitude
	double q = std::nearbyint(std::ldexp(a, -ulp_exp)); // default rounding mode = ties-to-even
	float v = (float)std::ldexp(q, ulp_exp);  // exactly representable in binary16 (or 65536 -> inf)
	return f2h(std::signbit(r) ? -v : v);
}

// Fused half multiply-add `__hfma(a, b, c)` (one rounding): a*b is exact in double and the double
// sum carries > 40 guard bits, so rounding the double once to half is the fused result.
inline uint16_t hfma(uint16_t a, uint16_t b, uint16_t c) {
	return d2h((double)h2f(a) * (double)h2f(b) + (double)h2f(c));
}

// ---------------------------------------------------------------------------------------------
// pcg32 [tcnn: dependencies/pcg32/pcg32.h, Wenzel Jakob's port of M. O'Neill's pcg32].
// `default_rng_t`, random_val.cuh:26; tcnn's copy gives advance() a default delta of 1<<32
// (used at testbed_nerf.cu:2541, 3377).
// ---------------------------------------------------------------------------------------------
struct Pcg32 {
	uint64_t state = 0x853c49e6748fea9bULL;
	uint64_t inc = 0xda3e39cb94b95bdbULL;
	static constexpr uint64_t MULT = 0x5851f42d4c957f2dULL;

	Pcg32() {}
	explicit Pcg32(uint64_t initstate, uint64_t initseq = 1u) { seed(initstate, initseq); }
	explicit Pcg32(const ngp_pcg32& p) : state(p.state), inc(p.inc) {}
	ngp_pcg32 pod() const { return {state, inc}; }

	void seed(uint64_t initstate, uint64_t initseq = 1) {
		state = 0u;
		inc = (initseq << 1u) | 1u;
		next_uint();
		state += initstate;
		next_uint();
	}
	uint32_t next_uint() {
		uint64_t oldstate = state;
		state = oldstate * MULT + inc;
		uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
		uint32_t rot = (uint32_t)(oldstate >> 59u);
		return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
	}
	float next_float() {
		uint32_t u = (next_uint() >> 9) | 0x3f800000u;
		float f; std::memcpy(&f, &u, 4);
		return f - 1.0f;
	}
	void advance(int64_t delta_ = (1ll << 32)) {
		uint64_t cur_mult = MULT, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
		uint64_t delta = (uint64_t)delta_;
		while (delta > 0) {
			if (delta & 1) {
				acc_mult *= cur_mult;
				acc_plus = acc_plus * cur_mult + cur_plus;
			}
			cur_plus = (cur_mult + 1) * cur_plus;
			cur_mult *= cur_mult;
			delta /= 2;
		}
		state = acc_mult * state + acc_plus;
	}
};

// ---------------------------------------------------------------------------------------------
// vec3 [tcnn vec.h]: plain float struct, component-wise operators.
// ---------------------------------------------------------------------------------------------
struct vec2 { float x, y; };
struct vec3 {
	float x, y, z;
	float& operator[](int i) { return (&x)[i]; }
	float operator[](int i) const { return (&x)[i]; }
};
struct vec4 { float x, y, z, w; };
inline vec3 V3(float a) { return {a, a, a}; }
inline vec3 V3(const float* p) { return {p[0], p[1], p[2]}; }
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator/(vec3 a, vec3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline vec3 operator+(vec3 a, float b) { return {a.x + b, a.y + b, a.z + b}; }
inline vec3 operator-(vec3 a, float b) { return {a.x - b, a.y - b, a.z - b}; }
inline vec3 operator*(vec3 a, float b) { return {a.x * b, a.y * b, a.z * b}; }
inline vec3 operator*(float b, vec3 a) { return {b * a.x, b * a.y, b * a.z}; }
inline vec3 operator/(vec3 a, float b) { return {a.x / b, a.y / b, a.z / b}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
inline vec3& operator-=(vec3& a, vec3 b) { a = a - b; return a; }
inline vec3& operator*=(vec3& a, float b) { a = a * b; return a; }
inline vec3& operator/=(vec3& a, float b) { a = a / b; return a; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
inline float distance(vec3 a, vec3 b) { return length(a - b); }
inline vec3 normalize(vec3 a) { return a / length(a); }
inline float mean(vec3 a) { return (a.x + a.y + a.z) / 3.0f; }
inline vec3 vabs(vec3 a) { return {std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)}; }
inline float vmax(vec3 a) { return std::max(std::max(a.x, a.y), a.z); }
inline float sign(float x) { return std::copysign(1.0f, x); } // [tcnn vec.h]
inline float clampf(float v, float lo, float hi) { return std::min(std::max(v, lo), hi); }
inline int clampi(int v, int lo, int hi) { return std::min(std::max(v, lo), hi); }
inline float logistic(float x) { return 1.0f / (1.0f + std::exp(-x)); } // [tcnn common_device.h]

// mat4x3 = 4 columns of vec3, column-major (camera: [0..2] rotation columns, [3] position).
struct mat4x3 {
	vec3 c[4];
	const vec3& operator[](int i) const { return c[i]; }
};
inline mat4x3 M43(const float* p) { return {{V3(p), V3(p + 3), V3(p + 6), V3(p + 9)}}; }
inline vec3 mul3(const mat4x3& m, vec3 v) { return m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z; } // mat3(m) * v

// ---------------------------------------------------------------------------------------------
// get_xform_given_rolling_shutter / camera_slerp, common_device.cuh:665-674.  slerp(mat3, mat3, t) is tiny-cuda-nn's [tcnn vec.h, GLM-derived;
// from the published algorithm]: quat_cast by the largest diagonal term, short-arc quaternion slerp (linear mix when cos(theta) > 1 - epsilon),
// mat3_cast.  Frames without motion data keep their matrix (see csrc/ngp_device.hpp for the reason).
// ---------------------------------------------------------------------------------------------
struct quat { float x, y, z, w; };
inline quat normalize(quat q) { float l = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w); return {q.x / l, q.y / l, q.z / l, q.w / l}; }
inline quat quat_cast(const mat4x3& m) {
	const vec3 &a = m.c[0], &b = m.c[1], &c = m.c[2];
	float four_x = a.x - b.y - c.z, four_y = b.y - a.x - c.z, four_z = c.z - a.x - b.y, four_w = a.x + b.y + c.z;
	int biggest = 0; float four_biggest = four_w;
	if (four_x > four_biggest) { four_biggest = four_x; biggest = 1; }
	if (four_y > four_biggest) { four_biggest = four_y; biggest = 2; }
	if (four_z > four_biggest) { four_biggest = four_z; biggest = 3; }
	float biggest_val = std::sqrt(four_biggest + 1.0f) * 0.5f, mult = 0.25f / biggest_val;
	if (biggest == 0) return {(b.z - c.y) * mult, (c.x - a.z) * mult, (a.y - b.x) * mult, biggest_val};
	if (biggest == 1) return {biggest_val, (a.y + b.x) * mult, (c.x + a.z) * mult, (b.z - c.y) * mult};
	if (biggest == 2) return {(a.y + b.x) * mult, biggest_val, (b.z + c.y) * mult, (c.x - a.z) * mult};
	return {(c.x + a.z) * mult, (b.z + c.y) * mult, biggest_val, (a.y - b.x) * mult};
}
inline quat slerp(quat x, quat y, float a) {
	quat z = y;
	float cos_theta = x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
	if (cos_theta < 0.0f) { z = {-y.x, -y.y, -y.z, -y.w}; cos_theta = -cos_theta; }
	if (cos_theta > 1.0f - std::numeric_limits<float>::epsilon()) return {x.x + a * (z.x - x.x), x.y + a * (z.y - x.y), x.z + a * (z.z - x.z), x.w + a * (z.w - x.w)};
	float angle = std::acos(cos_theta), s0 = std::sin((1.0f - a) * angle), s1 = std::sin(a * angle), sd = std::sin(angle);
	return {(s0 * x.x + s1 * z.x) / sd, (s0 * x.y + s1 * z.y) / sd, (s0 * x.z + s1 * z.z) / sd, (s0 * x.w + s1 * z.w) / sd};
}
inline mat4x3 camera_slerp(const mat4x3& a, const mat4x3& b, float t) {
	quat q = normalize(slerp(normalize(quat_cast(a)), normalize(quat_cast(b)), t));
	float qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z, qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z, qwx = q.w * q.x, qwy = q.w * q.y, qwz = q.w * q.z;
	mat4x3 r;
	r.c[0] = {1.0f - 2.0f * (qyy + qzz), 2.0f * (qxy + qwz), 2.0f * (qxz - qwy)};
	r.c[1] = {2.0f * (qxy - qwz), 1.0f - 2.0f * (qxx + qzz), 2.0f * (qyz + qwx)};
	r.c[2] = {2.0f * (qxz + qwy), 2.0f * (qyz - qwx), 1.0f - 2.0f * (qxx + qyy)};
	r.c[3] = a.c[3] * (1.0f - t) + b.c[3] * t;
	return r;
}
inline mat4x3 get_xform_given_rolling_shutter(const ngp_xform& X, const float rs[4], vec2 uv, float motionblur_time) {
	bool moving = rs[0] != 0.f || rs[1] != 0.f || rs[2] != 0.f || rs[3] != 0.f;
	for (int k = 0; k < 12; ++k) moving = moving || X.start[k] != X.end[k];
	if (!moving) return M43(X.start);
	float pixel_t = rs[0] + rs[1] * uv.x + rs[2] * uv.y + rs[3] * motionblur_time;
	return camera_slerp(M43(X.start), M43(X.end), pixel_t);
}

// ---------------------------------------------------------------------------------------------
// Morton codes [tcnn common_device.h: expand_bits / morton3D / morton3D_invert]
// ---------------------------------------------------------------------------------------------
inline uint32_t expand_bits(uint32_t v) {
	v = (v * 0x00010001u) & 0xFF0000FFu;
	v = (v * 0x00000101u) & 0x0F00F00Fu;
	v = (v * 0x00000011u) & 0xC30C30C3u;
	v = (v * 0x00000005u) & 0x49249249u;
	return v;
}
inline uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) {
	return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
inline uint32_t morton3D_invert(uint32_t x) {
	x = x & 0x49249249;
	x = (x | (x >> 2)) & 0xc30c30c3;
	x = (x | (x >> 4)) & 0x0f00f00f;
	x = (x | (x >> 8)) & 0xff0000ff;
	x = (x | (x >> 16)) & 0x0000ffff;
	return x;
}

// ---------------------------------------------------------------------------------------------
// colour transfer, common_device.cuh:61-103
// ---------------------------------------------------------------------------------------------
inline float srgb_to_linear(float srgb) {
	if (srgb <= 0.04045f) return srgb / 12.92f;
	return std::pow((srgb + 0.055f) / 1.055f, 2.4f);
}
inline vec3 srgb_to_linear(vec3 x) { return {srgb_to_linear(x.x), srgb_to_linear(x.y), srgb_to_linear(x.z)}; }
inline float linear_to_srgb(float linear) {
	if (linear < 0.0031308f) return 12.92f * linear;
	return 1.055f * std::pow(linear, 0.41666f) - 0.055f;
}
inline vec3 linear_to_srgb(vec3 x) { return {linear_to_srgb(x.x), linear_to_srgb(x.y), linear_to_srgb(x.z)}; }

// ---------------------------------------------------------------------------------------------
// BoundingBox, bounding_box.cuh:82-84 (relative_pos), :173-216 (ray_intersect), :222-227 (contains)
// ---------------------------------------------------------------------------------------------
struct Aabb {
	vec3 min, max;
	Aabb() : min(V3(0.f)), max(V3(1.f)) {}
	explicit Aabb(const ngp_aabb& a) : min(V3(a.min)), max(V3(a.max)) {}
	vec3 diag() const { return max - min; }
	vec3 relative_pos(vec3 p) const { return (p - min) / diag(); }
	bool contains(vec3 p) const {
		return p.x >= min.x && p.x <= max.x && p.y >= min.y && p.y <= max.y && p.z >= min.z && p.z <= max.z;
	}
	vec2 ray_intersect(vec3 pos, vec3 dir) const {
		const float FMAX = std::numeric_limits<float>::max();
		float tmin = (min.x - pos.x) / dir.x;
		float tmax = (max.x - pos.x) / dir.x;
		if (tmin > tmax) std::swap(tmin, tmax);
		float tymin = (min.y - pos.y) / dir.y;
		float tymax = (max.y - pos.y) / dir.y;
		if (tymin > tymax) std::swap(tymin, tymax);
		if (tmin > tymax || tymin > tmax) return {FMAX, FMAX};
		if (tymin > tmin) tmin = tymin;
		if (tymax < tmax) tmax = tymax;
		float tzmin = (min.z - pos.z) / dir.z;
		float tzmax = (max.z - pos.z) / dir.z;
		if (tzmin > tzmax) std::swap(tzmin, tzmax);
		if (tmin > tzmax || tzmin > tmax) return {FMAX, FMAX};
		if (tzmin > tmin) tmin = tzmin;
		if (tzmax < tmax) tmax = tzmax;
		return {tmin, tmax};
	}
};

// ---------------------------------------------------------------------------------------------
// nerf_device.cuh constants :25-43, common_device.cuh:33
// ---------------------------------------------------------------------------------------------
constexpr uint32_t NERF_GRIDSIZE = 128;
constexpr uint32_t NERF_GRID_N_CELLS = NERF_GRIDSIZE * NERF_GRIDSIZE * NERF_GRIDSIZE;
constexpr uint32_t NERF_STEPS = 1024;
constexpr uint32_t NERF_CASCADES = 8;
constexpr float SQRT3 = 1.73205080757f;
constexpr float STEPSIZE = SQRT3 / NERF_STEPS;
constexpr float MIN_CONE_STEPSIZE = STEPSIZE;
constexpr float MAX_CONE_STEPSIZE = STEPSIZE * (1 << (NERF_CASCADES - 1)) * NERF_STEPS / NERF_GRIDSIZE;
constexpr uint32_t N_MAX_RANDOM_SAMPLES_PER_RAY = 16;
constexpr float NERF_MIN_OPTICAL_THICKNESS = 0.01f;
constexpr float MAX_DEPTH = 16384.0f;

// activations, nerf_device.cuh:204-264
inline float network_to_rgb(float val, int act) {
	switch (act) {
		case NGP_ACT_NONE: return val;
		case NGP_ACT_RELU: return val > 0.0f ? val : 0.0f;
		case NGP_ACT_LOGISTIC: return logistic(val);
		case NGP_ACT_EXPONENTIAL: return std::exp(clampf(val, -10.0f, 10.0f));
	}
	return 0.0f;
}
inline float network_to_rgb_derivative(float val, int act) {
	switch (act) {
		case NGP_ACT_NONE: return 1.0f;
		case NGP_ACT_RELU: return val > 0.0f ? 1.0f : 0.0f;
		case NGP_ACT_LOGISTIC: { float d = logistic(val); return d * (1 - d); }
		case NGP_ACT_EXPONENTIAL: return std::exp(clampf(val, -10.0f, 10.0f));
	}
	return 0.0f;
}
inline float network_to_density(float val, int act) {
	switch (act) {
		case NGP_ACT_NONE: return val;
		case NGP_ACT_RELU: return val > 0.0f ? val : 0.0f;
		case NGP_ACT_LOGISTIC: return logistic(val);
		case NGP_ACT_EXPONENTIAL: return std::exp(val);
	}
	return 0.0f;
}
inline float network_to_density_derivative(float val, int act) {
	switch (act) {
		case NGP_ACT_NONE: return 1.0f;
		case NGP_ACT_RELU: return val > 0.0f ? 1.0f : 0.0f;
		case NGP_ACT_LOGISTIC: { float d = logistic(val); return d * (1 - d); }
		case NGP_ACT_EXPONENTIAL: return std::exp(clampf(val, -15.0f, 15.0f));
	}
	return 0.0f;
}

// warps, nerf_device.cuh:266-315
inline vec3 warp_position(vec3 pos, const Aabb& aabb) { return aabb.relative_pos(pos); }
inline vec3 unwarp_position(vec3 pos, const Aabb& aabb) { return aabb.min + pos * aabb.diag(); }
inline vec3 warp_direction(vec3 dir) { return (dir + 1.0f) * 0.5f; }
inline float warp_dt(float dt) {
	float max_stepsize = MIN_CONE_STEPSIZE * (1 << (NERF_CASCADES - 1));
	return (dt - MIN_CONE_STEPSIZE) / (max_stepsize - MIN_CONE_STEPSIZE);
}
inline float unwarp_dt(float dt) {
	float max_stepsize = MIN_CONE_STEPSIZE * (1 << (NERF_CASCADES - 1));
	return dt * (max_stepsize - MIN_CONE_STEPSIZE) + MIN_CONE_STEPSIZE;
}

// occupancy-grid index math, nerf_device.cuh:317-341
inline uint32_t cascaded_grid_idx_at(vec3 pos, uint32_t mip) {
	float mip_scale = std::scalbn(1.0f, -(int)mip);
	pos -= V3(0.5f);
	pos *= mip_scale;
	pos += V3(0.5f);
	int ix = (int)(pos.x * (float)NERF_GRIDSIZE);
	int iy = (int)(pos.y * (float)NERF_GRIDSIZE);
	int iz = (int)(pos.z * (float)NERF_GRIDSIZE);
	if (ix < 0 || ix >= (int)NERF_GRIDSIZE || iy < 0 || iy >= (int)NERF_GRIDSIZE || iz < 0 || iz >= (int)NERF_GRIDSIZE) {
		return 0xFFFFFFFFu;
	}
	return morton3D((uint32_t)ix, (uint32_t)iy, (uint32_t)iz);
}
inline uint32_t grid_mip_offset(uint32_t mip) { return NERF_GRID_N_CELLS * mip; }
inline bool density_grid_occupied_at(vec3 pos, const uint8_t* bitfield, uint32_t mip) {
	uint32_t idx = cascaded_grid_idx_at(pos, mip);
	if (idx == 0xFFFFFFFFu) return false;
	return bitfield[idx / 8 + grid_mip_offset(mip) / 8] & (1 << (idx % 8));
}
inline float cascaded_grid_at(vec3 pos, const float* grid, uint32_t mip) {
	uint32_t idx = cascaded_grid_idx_at(pos, mip);
	if (idx == 0xFFFFFFFFu) return 0.0f;
	return grid[idx + grid_mip_offset(mip)];
}

// stepping, nerf_device.cuh:360-460
inline float distance_to_next_voxel(vec3 pos, vec3 dir, vec3 idir, float res) {
	vec3 p = res * (pos - 0.5f);
	float tx = (std::floor(p.x + 0.5f + 0.5f * sign(dir.x)) - p.x) * idir.x;
	float ty = (std::floor(p.y + 0.5f + 0.5f * sign(dir.y)) - p.y) * idir.y;
	float tz = (std::floor(p.z + 0.5f + 0.5f * sign(dir.z)) - p.z) * idir.z;
	float t = std::min(std::min(tx, ty), tz);
	return std::fmax(t / res, 0.0f);
}
inline float to_stepping_space(float t, float cone_angle) {
	if (cone_angle <= 1e-5f) return t / MIN_CONE_STEPSIZE;
	float log1p_c = std::log(1.0f + cone_angle);
	float a = (std::log(MIN_CONE_STEPSIZE) - std::log(log1p_c)) / log1p_c;
	float b = (std::log(MAX_CONE_STEPSIZE) - std::log(log1p_c)) / log1p_c;
	float at = std::exp(a * log1p_c);
	float bt = std::exp(b * log1p_c);
	if (t <= at) return (t - at) / MIN_CONE_STEPSIZE + a;
	else if (t <= bt) return std::log(t) / log1p_c;
	else return (t - bt) / MAX_CONE_STEPSIZE + b;
}
inline float from_stepping_space(float n, float cone_angle) {
	if (cone_angle <= 1e-5f) return n * MIN_CONE_STEPSIZE;
	float log1p_c = std::log(1.0f + cone_angle);
	float a = (std::log(MIN_CONE_STEPSIZE) - std::log(log1p_c)) / log1p_c;
	float b = (std::log(MAX_CONE_STEPSIZE) - std::log(log1p_c)) / log1p_c;
	float at = std::exp(a * log1p_c);
	float bt = std::exp(b * log1p_c);
	if (n <= a) return (n - a) * MIN_CONE_STEPSIZE + at;
	else if (n <= b) return std::exp(n * log1p_c);
	else return (n - b) * MAX_CONE_STEPSIZE + bt;
}
inline float advance_n_steps(float t, float cone_angle, float n) {
	return from_stepping_space(to_stepping_space(t, cone_angle) + n, cone_angle);
}
inline float calc_dt(float t, float cone_angle) { return advance_n_steps(t, cone_angle, 1.0f) - t; }
inline float advance_to_next_voxel(float t, float cone_angle, vec3 pos, vec3 dir, vec3 idir, uint32_t mip) {
	float res = std::scalbn((float)NERF_GRIDSIZE, -(int)mip);
	float t_target = t + distance_to_next_voxel(pos, dir, idir, res);
	t = to_stepping_space(t, cone_angle);
	t_target = to_stepping_space(t_target, cone_angle);
	return from_stepping_space(t + std::ceil(std::fmax(t_target - t, 0.5f)), cone_angle);
}
inline uint32_t mip_from_pos(vec3 pos, uint32_t max_cascade = NERF_CASCADES - 1) {
	int exponent;
	float maxval = vmax(vabs(pos - 0.5f));
	std::frexp(maxval, &exponent);
	return (uint32_t)clampi(exponent + 1, 0, (int)max_cascade);
}
inline uint32_t mip_from_dt(float dt, vec3 pos, uint32_t max_cascade = NERF_CASCADES - 1) {
	uint32_t mip = mip_from_pos(pos, max_cascade);
	dt *= 2 * NERF_GRIDSIZE;
	if (dt < 1.0f) return mip;
	int exponent;
	std::frexp(dt, &exponent);
	// tcnn's scalar clamp tests the lower bound first (vec.h: a < b ? b : (c < a ? c : a)): with exponent > max_cascade (long steps) the result is `exponent`, a pooled
	// level above max_cascade -- also what the pre-tcnn code did (min(NERF_CASCADES() - 1, max(exponent, mip))).  Pinned against the reference compiled with that clamp
	// (oracle/_ref/libngpkern_ref.so, tests/test_ref_kernels.py); the min(max()) variant is kept as libngpkern_ref_clamp_min_max.so to state the difference.
	const int m = (int)mip, hi = (int)max_cascade;
	return (uint32_t)(m < exponent ? exponent : (hi < m ? hi : m));
}
// nerf_device.cuh:462-495 (MIP_FROM_DT = false instantiation; aabb_to_local = identity)
inline float if_unoccupied_advance_to_next_occupied_voxel(float t, float cone_angle, vec3 o, vec3 d, vec3 idir,
		const uint8_t* density_grid, uint32_t min_mip, uint32_t max_mip, const Aabb& aabb) {
	while (true) {
		vec3 pos = o + d * t;
		if (t >= MAX_DEPTH || !aabb.contains(pos)) return MAX_DEPTH;
		uint32_t mip = (uint32_t)clampi((int)mip_from_pos(pos), (int)min_mip, (int)max_mip);
		if (!density_grid || density_grid_occupied_at(pos, density_grid, mip)) return t;
		while (mip < max_mip && !density_grid_occupied_at(pos, density_grid, mip + 1)) ++mip;
		t = advance_to_next_voxel(t, cone_angle, pos, d, idir, mip);
	}
}

// ---------------------------------------------------------------------------------------------
// low-discrepancy sampler (renderer jitter), random_val.cuh:60-291.  Only Sobol dimensions 0 and 1
// are consumed on this path; their direction numbers are generated instead of tabulated:
// dim 0 = van der Corput (bit reversal), dim 1: v[k] = v[k-1] ^ (v[k-1] >> 1), v[0] = 1<<31.
// ---------------------------------------------------------------------------------------------
inline uint32_t sobol(uint32_t index, uint32_t dim) {
	uint32_t X = 0, v = 0x80000000u;
	for (uint32_t bit = 0; bit < 32; ++bit) {
		uint32_t dirn = (dim == 0) ? (0x80000000u >> bit) : v;
		if ((index >> bit) & 1u) X ^= dirn;
		v = v ^ (v >> 1);
	}
	return X;
}
inline uint32_t hash_combine(uint32_t seed, uint32_t v) { return seed ^ (v + (seed << 6) + (seed >> 2)); }
inline uint32_t reverse_bits(uint32_t x) {
	x = (((x & 0xaaaaaaaa) >> 1) | ((x & 0x55555555) << 1));
	x = (((x & 0xcccccccc) >> 2) | ((x & 0x33333333) << 2));
	x = (((x & 0xf0f0f0f0) >> 4) | ((x & 0x0f0f0f0f) << 4));
	x = (((x & 0xff00ff00) >> 8) | ((x & 0x00ff00ff) << 8));
	return ((x >> 16) | (x << 16));
}
inline uint32_t laine_karras_permutation(uint32_t x, uint32_t seed) {
	x += seed;
	x ^= x * 0x6c50b47cu;
	x ^= x * 0xb82f1e52u;
	x ^= x * 0xc7afe638u;
	x ^= x * 0x8d22f6e6u;
	return x;
}
inline uint32_t nested_uniform_scramble_base2(uint32_t x, uint32_t seed) {
	x = reverse_bits(x);
	x = laine_karras_permutation(x, seed);
	x = reverse_bits(x);
	return x;
}
inline float ld_random_val(uint32_t index, uint32_t seed, uint32_t dim = 0) {
	constexpr float S = float(1.0 / (1ull << 32));
	index = nested_uniform_scramble_base2(index, seed);
	return (float)nested_uniform_scramble_base2(sobol(index, dim), hash_combine(seed, dim)) * S;
}
inline vec2 ld_random_val_2d(uint32_t index, uint32_t seed) {
	constexpr float S = float(1.0 / (1ull << 32));
	index = nested_uniform_scramble_base2(index, seed);
	uint32_t x0 = nested_uniform_scramble_base2(sobol(index, 0), hash_combine(seed, 0));
	uint32_t x1 = nested_uniform_scramble_base2(sobol(index, 1), hash_combine(seed, 1));
	return {(float)x0 * S, (float)x1 * S};
}
inline float fractf(float x) { return x - std::floor(x); }
inline vec2 ld_random_pixel_offset(uint32_t spp) {
	vec2 a = ld_random_val_2d(0, 0xdeadbeef), b = ld_random_val_2d(spp, 0xdeadbeef);
	return {fractf(0.5f - a.x + b.x), fractf(0.5f - a.y + b.y)};
}

} // namespace ora
