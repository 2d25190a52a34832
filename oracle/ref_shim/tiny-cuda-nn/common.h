// oracle/ref_shim -- TEST INFRASTRUCTURE ONLY.  A stand-in for the parts of tiny-cuda-nn's vector-math headers (absent from the reference mount: empty submodule) that the
// reference's own device headers (include/neural-graphics-primitives/{common.h, common_device.cuh, nerf_device.cuh, random_val.cuh, bounding_box.cuh, triangle.cuh}) need in order
// to compile on the CPU with g++.  With it, oracle/Makefile builds oracle/_ref/libngpdev_ref.so FROM THE REFERENCE'S HEADERS WHERE THEY LIE, and tests/test_ref_device.py checks
// the oracle's restatement of those formulas (stepping space, occupancy indexing, losses, activations, warps, lens models, Sobol / Halton, box intersection ...) against the
// reference's own code, bit for bit.  What this pins: every constant, branch and operation order the reference wrote.  What it cannot pin: tcnn's vector semantics themselves --
// written here the GLSL way (componentwise operators, column-major matrices, m[c] = column c), as tcnn documents its types.  Never shipped, never linked by the product.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __restrict__
#include <type_traits>
#define TCNN_HOST_DEVICE
#define TCNN_DEVICE
#define TCNN_HOST

namespace tcnn {
using std::min; using std::max; using std::abs; using std::sqrt; using std::pow; using std::exp; using std::log; using std::floor; using std::ceil; using std::copysign;
using std::isfinite; using std::isnan; using std::sin; using std::cos; using std::tan; using std::atan; using std::atan2; using std::asin; using std::acos; using std::fmod;

// vectors of other lengths (the fused kernels' network input, vec<7>): plain arrays
template <typename T, uint32_t N> struct tvec {
	T data[N];
	tvec() : data{} {}
	tvec(T s) { for (uint32_t i = 0; i < N; ++i) data[i] = s; }
	T& operator[](uint32_t i) { return data[i]; }
	const T& operator[](uint32_t i) const { return data[i]; }
	static constexpr uint32_t size() { return N; }
};
template <typename T> struct tvec<T, 2> {
	union { T x, r; }; union { T y, g; };
	tvec() : x{}, y{} {}
	tvec(T s) : x{s}, y{s} {}
	tvec(T a, T b) : x{a}, y{b} {}
	template <typename U> tvec(const tvec<U, 2>& o) : x{(T)o.x}, y{(T)o.y} {}
	template <typename U> tvec(const tvec<U, 3>& o) : x{(T)o.x}, y{(T)o.y} {}
	T& operator[](uint32_t i) { return i == 0 ? x : y; }
	const T& operator[](uint32_t i) const { return i == 0 ? x : y; }
	static constexpr uint32_t size() { return 2; }
};
template <typename T> struct tvec<T, 3> {
	union { T x, r; }; union { T y, g; }; union { T z, b; };
	tvec() : x{}, y{}, z{} {}
	tvec(T s) : x{s}, y{s}, z{s} {}
	tvec(T a, T b_, T c) : x{a}, y{b_}, z{c} {}
	tvec(const tvec<T, 2>& v, T c) : x{v.x}, y{v.y}, z{c} {}
	template <typename U> tvec(const tvec<U, 3>& o) : x{(T)o.x}, y{(T)o.y}, z{(T)o.z} {}
	template <typename U> tvec(const tvec<U, 4>& o) : x{(T)o.x}, y{(T)o.y}, z{(T)o.z} {}
	T& operator[](uint32_t i) { return i == 0 ? x : i == 1 ? y : z; }
	const T& operator[](uint32_t i) const { return i == 0 ? x : i == 1 ? y : z; }
	tvec<T, 2>& xy() { return *reinterpret_cast<tvec<T, 2>*>(this); } // (x, y are the first two members: swizzles that can be assigned to)
	const tvec<T, 2>& xy() const { return *reinterpret_cast<const tvec<T, 2>*>(this); }
	tvec<T, 2>& yz() { return *reinterpret_cast<tvec<T, 2>*>(&y); }
	const tvec<T, 2>& yz() const { return *reinterpret_cast<const tvec<T, 2>*>(&y); }
	tvec<T, 3>& rgb() { return *this; }
	const tvec<T, 3>& rgb() const { return *this; }
	static constexpr uint32_t size() { return 3; }
};
template <typename T> struct tvec<T, 4> {
	union { T x, r; }; union { T y, g; }; union { T z, b; }; union { T w, a; };
	tvec() : x{}, y{}, z{}, w{} {}
	tvec(T s) : x{s}, y{s}, z{s}, w{s} {}
	tvec(T a_, T b_, T c, T d) : x{a_}, y{b_}, z{c}, w{d} {}
	tvec(const tvec<T, 3>& v, T d) : x{v.x}, y{v.y}, z{v.z}, w{d} {}
	template <typename U> tvec(const tvec<U, 4>& o) : x{(T)o.x}, y{(T)o.y}, z{(T)o.z}, w{(T)o.w} {}
	T& operator[](uint32_t i) { return i == 0 ? x : i == 1 ? y : i == 2 ? z : w; }
	const T& operator[](uint32_t i) const { return i == 0 ? x : i == 1 ? y : i == 2 ? z : w; }
	tvec<T, 3>& xyz() { return *reinterpret_cast<tvec<T, 3>*>(this); }
	const tvec<T, 3>& xyz() const { return *reinterpret_cast<const tvec<T, 3>*>(this); }
	tvec<T, 3>& rgb() { return *reinterpret_cast<tvec<T, 3>*>(this); }
	const tvec<T, 3>& rgb() const { return *reinterpret_cast<const tvec<T, 3>*>(this); }
	tvec<T, 2>& xy() { return *reinterpret_cast<tvec<T, 2>*>(this); }
	const tvec<T, 2>& xy() const { return *reinterpret_cast<const tvec<T, 2>*>(this); }
	static constexpr uint32_t size() { return 4; }
};
#define NGP_SHIM_BINOP(op) \
	template <typename T, uint32_t N> tvec<T, N> operator op(const tvec<T, N>& a, const tvec<T, N>& b) { tvec<T, N> r; for (uint32_t i = 0; i < N; ++i) r[i] = a[i] op b[i]; return r; } \
	template <typename T, uint32_t N> tvec<T, N> operator op(const tvec<T, N>& a, T b) { tvec<T, N> r; for (uint32_t i = 0; i < N; ++i) r[i] = a[i] op b; return r; } \
	template <typename T, uint32_t N> tvec<T, N> operator op(T a, const tvec<T, N>& b) { tvec<T, N> r; for (uint32_t i = 0; i < N; ++i) r[i] = a op b[i]; return r; } \
	template <typename T, uint32_t N> tvec<T, N>& operator op##=(tvec<T, N>& a, const tvec<T, N>& b) { for (uint32_t i = 0; i < N; ++i) a[i] = a[i] op b[i]; return a; } \
	template <typename T, uint32_t N> tvec<T, N>& operator op##=(tvec<T, N>& a, T b) { for (uint32_t i = 0; i < N; ++i) a[i] = a[i] op b; return a; }
NGP_SHIM_BINOP(+) NGP_SHIM_BINOP(-) NGP_SHIM_BINOP(*) NGP_SHIM_BINOP(/)
#undef NGP_SHIM_BINOP
template <typename T, uint32_t N> tvec<T, N> operator-(const tvec<T, N>& a) { tvec<T, N> r; for (uint32_t i = 0; i < N; ++i) r[i] = -a[i]; return r; }
template <typename T, uint32_t N> bool operator==(const tvec<T, N>& a, const tvec<T, N>& b) { for (uint32_t i = 0; i < N; ++i) if (!(a[i] == b[i])) return false; return true; }
template <typename T, uint32_t N> bool operator!=(const tvec<T, N>& a, const tvec<T, N>& b) { return !(a == b); }
// integer vectors: shifts / modulo / bit operations
template <uint32_t N> tvec<int, N> operator%(const tvec<int, N>& a, const tvec<int, N>& b) { tvec<int, N> r; for (uint32_t i = 0; i < N; ++i) r[i] = a[i] % b[i]; return r; }
template <uint32_t N> tvec<int, N> operator%(const tvec<int, N>& a, int b) { tvec<int, N> r; for (uint32_t i = 0; i < N; ++i) r[i] = a[i] % b; return r; }
template <uint32_t N> tvec<uint32_t, N> operator>>(const tvec<uint32_t, N>& a, uint32_t b) { tvec<uint32_t, N> r; for (uint32_t i = 0; i < N; ++i) r[i] = a[i] >> b; return r; }

#define NGP_SHIM_MAP1(name, expr) template <typename T, uint32_t N> tvec<T, N> name(const tvec<T, N>& a) { tvec<T, N> r; for (uint32_t i = 0; i < N; ++i) { const T v = a[i]; r[i] = (T)(expr); } return r; }
NGP_SHIM_MAP1(abs, std::abs(v)) NGP_SHIM_MAP1(floor, std::floor(v)) NGP_SHIM_MAP1(ceil, std::ceil(v)) NGP_SHIM_MAP1(sqrt, std::sqrt(v)) NGP_SHIM_MAP1(exp, std::exp(v)) NGP_SHIM_MAP1(log, std::log(v))
NGP_SHIM_MAP1(sin, std::sin(v)) NGP_SHIM_MAP1(cos, std::cos(v)) NGP_SHIM_MAP1(sign, std::copysign(T(1), v)) NGP_SHIM_MAP1(isfinite, std::isfinite(v))
#undef NGP_SHIM_MAP1
template <typename T, uint32_t N> tvec<T, N> min(const tvec<T, N>& a, const tvec<T, N>& b) { tvec<T, N> r; for (uint32_t i = 0; i < N; ++i) r[i] = std::min(a[i], b[i]); return r; }
template <typename T, uint32_t N> tvec<T, N> max(const tvec<T, N>& a, const tvec<T, N>& b) { tvec<T, N> r; for (uint32_t i = 0; i < N; ++i) r[i] = std::max(a[i], b[i]); return r; }
template <typename T, uint32_t N> tvec<T, N> min(const tvec<T, N>& a, T b) { return min(a, tvec<T, N>(b)); }
template <typename T, uint32_t N> tvec<T, N> max(const tvec<T, N>& a, T b) { return max(a, tvec<T, N>(b)); }
template <typename T, uint32_t N> T min(const tvec<T, N>& a) { T r = a[0]; for (uint32_t i = 1; i < N; ++i) r = std::min(r, a[i]); return r; }
template <typename T, uint32_t N> T max(const tvec<T, N>& a) { T r = a[0]; for (uint32_t i = 1; i < N; ++i) r = std::max(r, a[i]); return r; }
template <typename T, uint32_t N> tvec<T, N> pow(const tvec<T, N>& a, T b) { tvec<T, N> r; for (uint32_t i = 0; i < N; ++i) r[i] = std::pow(a[i], b); return r; }
template <typename T, uint32_t N> tvec<T, N> pow(const tvec<T, N>& a, const tvec<T, N>& b) { tvec<T, N> r; for (uint32_t i = 0; i < N; ++i) r[i] = std::pow(a[i], b[i]); return r; }
template <typename T, uint32_t N> tvec<T, N> copysign(const tvec<T, N>& a, const tvec<T, N>& b) { tvec<T, N> r; for (uint32_t i = 0; i < N; ++i) r[i] = std::copysign(a[i], b[i]); return r; }
// Scalar clamp.  tcnn's vec.h defines clamp(a, b, c) = a < b ? b : (c < a ? c : a): the LOWER bound is tested first.  It only matters where the bounds cross, and the reference
// has one such call on the hot path: mip_from_dt ends in clamp((int)mip, exponent, (int)max_cascade) (nerf_device.cuh:459) with exponent > max_cascade for long steps; the
// lower-bound-first form yields `exponent` (a pooled bitfield level above max_cascade), which is also what the pre-tcnn-vector code base computed
// (min(NERF_CASCADES() - 1, max(exponent, mip)): capped by the number of levels, not by max_cascade).  The oracle and the HIP kernels follow it since round 4.
// GLSL's min(max(x, lo), hi) would stay at max_cascade: -DNGP_SHIM_CLAMP_MIN_MAX builds that variant (oracle/_ref/libngpkern_ref_clamp_min_max.so) so that
// tests/test_ref_kernels.py can state what the choice changes.
#ifdef NGP_SHIM_CLAMP_MIN_MAX
template <typename T> T clamp(T v, T lo, T hi) { return std::min(std::max(v, lo), hi); }
#else
template <typename T> T clamp(T v, T lo, T hi) { return v < lo ? lo : (hi < v ? hi : v); }
#endif
template <typename T, uint32_t N> tvec<T, N> clamp(const tvec<T, N>& v, const tvec<T, N>& lo, const tvec<T, N>& hi) { return min(max(v, lo), hi); }
template <typename T, uint32_t N> tvec<T, N> clamp(const tvec<T, N>& v, T lo, T hi) { return min(max(v, tvec<T, N>(lo)), tvec<T, N>(hi)); }
template <typename T, uint32_t N> tvec<T, N> clamp(const tvec<T, N>& v, T lo, const tvec<T, N>& hi) { return min(max(v, tvec<T, N>(lo)), hi); }
template <typename T, uint32_t N> tvec<T, N> clamp(const tvec<T, N>& v, const tvec<T, N>& lo, T hi) { return min(max(v, lo), tvec<T, N>(hi)); }
template <typename T> T mix(T a, T b, T t) { return a * (T(1) - t) + b * t; }
template <typename T, uint32_t N> tvec<T, N> mix(const tvec<T, N>& a, const tvec<T, N>& b, T t) { return a * (T(1) - t) + b * t; }
template <typename T, uint32_t N> tvec<T, N> mix(const tvec<T, N>& a, const tvec<T, N>& b, const tvec<T, N>& t) { return a * (tvec<T, N>(T(1)) - t) + b * t; }
template <typename T, uint32_t N> T dot(const tvec<T, N>& a, const tvec<T, N>& b) { T r = a[0] * b[0]; for (uint32_t i = 1; i < N; ++i) r += a[i] * b[i]; return r; }
template <typename T, uint32_t N> T length2(const tvec<T, N>& a) { return dot(a, a); }
template <typename T, uint32_t N> T length(const tvec<T, N>& a) { return std::sqrt(length2(a)); }
template <typename T, uint32_t N> T distance(const tvec<T, N>& a, const tvec<T, N>& b) { return length(a - b); }
template <typename T, uint32_t N> tvec<T, N> normalize(const tvec<T, N>& a) { const T len = length(a); if (len == T(0)) { tvec<T, N> r{T(0)}; r[0] = T(1); return r; } return a / len; }
template <typename T> tvec<T, 3> cross(const tvec<T, 3>& a, const tvec<T, 3>& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
template <typename T, uint32_t N> T mean(const tvec<T, N>& a) { T r = a[0]; for (uint32_t i = 1; i < N; ++i) r += a[i]; return r / (T)N; }
template <typename T, uint32_t N> T sum(const tvec<T, N>& a) { T r = a[0]; for (uint32_t i = 1; i < N; ++i) r += a[i]; return r; }
template <typename T, uint32_t N> T product(const tvec<T, N>& a) { T r = a[0]; for (uint32_t i = 1; i < N; ++i) r *= a[i]; return r; }
// sign: tcnn's is copysign(1, x) [vec.h, from memory] -- never 0, unlike GLSL's; it matters for direction components that are exactly 0 (axis-parallel rays in
// distance_to_next_voxel: with a three-valued sign the step to the next voxel collapses to 0 for half of them).  The oracle and the HIP code use the same form.
inline float sign(float v) { return std::copysign(1.0f, v); }
inline float logistic(float x) { return 1.0f / (1.0f + std::exp(-x)); }
inline float logit(float x) { return -std::log(1.0f / (std::fmin(std::fmax(x, 1e-9f), 1.0f - 1e-9f)) - 1.0f); }
inline float fract(float x) { return x - std::floor(x); }

using vec2 = tvec<float, 2>; using vec3 = tvec<float, 3>; using vec4 = tvec<float, 4>;
using ivec2 = tvec<int, 2>; using ivec3 = tvec<int, 3>; using ivec4 = tvec<int, 4>;
using uvec2 = tvec<uint32_t, 2>; using uvec3 = tvec<uint32_t, 3>; using uvec4 = tvec<uint32_t, 4>;
using u16vec3 = tvec<uint16_t, 3>;
template <uint32_t N> using vec = tvec<float, N>;
using bvec3 = tvec<bool, 3>;
using u16vec2 = tvec<uint16_t, 2>;

// column-major matrices, C columns of R rows: m[c] = column c
template <typename T, uint32_t C, uint32_t R> struct tmat {
	tvec<T, R> m[C];
	tmat() {}
	explicit tmat(T diag) { for (uint32_t c = 0; c < C; ++c) { m[c] = tvec<T, R>(T(0)); if (c < R) m[c][c] = diag; } }
	tmat(const tvec<T, R>& c0, const tvec<T, R>& c1, const tvec<T, R>& c2) { static_assert(C == 3, ""); m[0] = c0; m[1] = c1; m[2] = c2; }
	tmat(const tvec<T, R>& c0, const tvec<T, R>& c1, const tvec<T, R>& c2, const tvec<T, R>& c3) { static_assert(C == 4, ""); m[0] = c0; m[1] = c1; m[2] = c2; m[3] = c3; }
	template <uint32_t C2, uint32_t R2> tmat(const tmat<T, C2, R2>& o) { for (uint32_t c = 0; c < C; ++c) for (uint32_t r = 0; r < R; ++r) m[c][r] = (c < C2 && r < R2) ? o.m[c][r] : (c == r ? T(1) : T(0)); }
	tvec<T, R>& operator[](uint32_t c) { return m[c]; }
	const tvec<T, R>& operator[](uint32_t c) const { return m[c]; }
	static tmat identity() { return tmat(T(1)); }
};
template <typename T, uint32_t C, uint32_t R> tvec<T, R> operator*(const tmat<T, C, R>& a, const tvec<T, C>& v) { tvec<T, R> r = a.m[0] * v[0]; for (uint32_t c = 1; c < C; ++c) r += a.m[c] * v[c]; return r; }
template <typename T, uint32_t C, uint32_t R, uint32_t C2> tmat<T, C2, R> operator*(const tmat<T, C, R>& a, const tmat<T, C2, C>& b) { tmat<T, C2, R> r; for (uint32_t c = 0; c < C2; ++c) r.m[c] = a * b.m[c]; return r; }
template <typename T, uint32_t C, uint32_t R> tmat<T, C, R> operator*(const tmat<T, C, R>& a, T s) { tmat<T, C, R> r; for (uint32_t c = 0; c < C; ++c) r.m[c] = a.m[c] * s; return r; }
template <typename T, uint32_t C, uint32_t R> tmat<T, R, C> transpose(const tmat<T, C, R>& a) { tmat<T, R, C> r; for (uint32_t c = 0; c < C; ++c) for (uint32_t q = 0; q < R; ++q) r.m[q][c] = a.m[c][q]; return r; }
template <typename T, uint32_t C, uint32_t R> bool operator==(const tmat<T, C, R>& a, const tmat<T, C, R>& b) { for (uint32_t c = 0; c < C; ++c) if (!(a.m[c] == b.m[c])) return false; return true; }
template <typename T, uint32_t C, uint32_t R> bool operator!=(const tmat<T, C, R>& a, const tmat<T, C, R>& b) { return !(a == b); }
using mat2x3 = tmat<float, 2, 3>; using mat2 = tmat<float, 2, 2>; using mat3 = tmat<float, 3, 3>; using mat4 = tmat<float, 4, 4>; using mat4x3 = tmat<float, 4, 3>; using mat3x4 = tmat<float, 3, 4>;
template <typename T> tvec<T, 3> row(const tmat<T, 3, 3>& a, uint32_t r) { return {a.m[0][r], a.m[1][r], a.m[2][r]}; }
// [tcnn vec.h, from memory] row r of any matrix, and a copy of the matrix with row r replaced (nerf_loader.h:113-116 cycles the axes of a mat4x3 with them)
template <typename T, uint32_t C, uint32_t R, typename = typename std::enable_if<!(C == 3 && R == 3)>::type> tvec<T, C> row(const tmat<T, C, R>& a, uint32_t r) { tvec<T, C> v; for (uint32_t c = 0; c < C; ++c) v[c] = a.m[c][r]; return v; }
template <typename T, uint32_t C, uint32_t R> tmat<T, C, R> row(const tmat<T, C, R>& a, uint32_t r, const tvec<T, C>& v) { tmat<T, C, R> o = a; for (uint32_t c = 0; c < C; ++c) o.m[c][r] = v[c]; return o; }

// quaternion {x, y, z, w} with the handful of operations camera_path.h / .cu use (GLM-style: to_mat3 = mat3_cast, quat(mat3) = quat_cast by the largest diagonal term)
struct quat {
	float x = 0, y = 0, z = 0, w = 1;
	quat() {}
	quat(float x_, float y_, float z_, float w_) : x{x_}, y{y_}, z{z_}, w{w_} {}
	explicit quat(const tmat<float, 3, 3>& m) {
		const float fx = m[0][0] - m[1][1] - m[2][2], fy = m[1][1] - m[0][0] - m[2][2], fz = m[2][2] - m[0][0] - m[1][1], fw = m[0][0] + m[1][1] + m[2][2];
		int big = 0; float fb = fw;
		if (fx > fb) { fb = fx; big = 1; }
		if (fy > fb) { fb = fy; big = 2; }
		if (fz > fb) { fb = fz; big = 3; }
		const float bv = std::sqrt(fb + 1.0f) * 0.5f, mult = 0.25f / bv;
		switch (big) {
			case 0: w = bv; x = (m[1][2] - m[2][1]) * mult; y = (m[2][0] - m[0][2]) * mult; z = (m[0][1] - m[1][0]) * mult; break;
			case 1: w = (m[1][2] - m[2][1]) * mult; x = bv; y = (m[0][1] + m[1][0]) * mult; z = (m[2][0] + m[0][2]) * mult; break;
			case 2: w = (m[2][0] - m[0][2]) * mult; x = (m[0][1] + m[1][0]) * mult; y = bv; z = (m[1][2] + m[2][1]) * mult; break;
			default: w = (m[0][1] - m[1][0]) * mult; x = (m[2][0] + m[0][2]) * mult; y = (m[1][2] + m[2][1]) * mult; z = bv; break;
		}
	}
};
inline quat operator*(const quat& q, float f) { return {q.x * f, q.y * f, q.z * f, q.w * f}; }
inline quat operator+(const quat& a, const quat& b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline quat operator-(const quat& q) { return {-q.x, -q.y, -q.z, -q.w}; }
inline float dot(const quat& a, const quat& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
inline quat normalize(const quat& q) { const float l = std::sqrt(dot(q, q)); return {q.x / l, q.y / l, q.z / l, q.w / l}; }
inline tmat<float, 3, 3> to_mat3(const quat& q) {
	const float qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z, qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z, qwx = q.w * q.x, qwy = q.w * q.y, qwz = q.w * q.z;
	tmat<float, 3, 3> r;
	r[0] = tvec<float, 3>(1.0f - 2.0f * (qyy + qzz), 2.0f * (qxy + qwz), 2.0f * (qxz - qwy));
	r[1] = tvec<float, 3>(2.0f * (qxy - qwz), 1.0f - 2.0f * (qxx + qzz), 2.0f * (qyz + qwx));
	r[2] = tvec<float, 3>(2.0f * (qxz + qwy), 2.0f * (qyz - qwx), 1.0f - 2.0f * (qxx + qyy));
	return r;
}
inline quat slerp(const quat&, const quat&, float) { std::fprintf(stderr, "oracle/ref_shim: slerp(quat) is tcnn code that is absent from the mount\n"); std::abort(); }
// 3 x 3 inverse (pos_to_uv's camera-space transform): reciprocal of the determinant times the cofactors, in the glm formulation tcnn's types follow.  tcnn itself is
// absent from the mount, so the rounding of this function is an ASSUMPTION the oracle shares, not something this shim can pin.
inline mat3 inverse(const mat3& m) {
	const float one_over_det = 1.0f / (m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2]) - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2]) + m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]));
	mat3 r;
	r[0][0] = +(m[1][1] * m[2][2] - m[2][1] * m[1][2]) * one_over_det; r[1][0] = -(m[1][0] * m[2][2] - m[2][0] * m[1][2]) * one_over_det; r[2][0] = +(m[1][0] * m[2][1] - m[2][0] * m[1][1]) * one_over_det;
	r[0][1] = -(m[0][1] * m[2][2] - m[2][1] * m[0][2]) * one_over_det; r[1][1] = +(m[0][0] * m[2][2] - m[2][0] * m[0][2]) * one_over_det; r[2][1] = -(m[0][0] * m[2][1] - m[2][0] * m[0][1]) * one_over_det;
	r[0][2] = +(m[0][1] * m[1][2] - m[1][1] * m[0][2]) * one_over_det; r[1][2] = -(m[0][0] * m[1][2] - m[1][0] * m[0][2]) * one_over_det; r[2][2] = +(m[0][0] * m[1][1] - m[1][0] * m[0][1]) * one_over_det;
	return r;
}
// 2 x 2 inverse (the Newton step of the lens undistortion): reciprocal of the determinant times the adjugate, the glm formulation tcnn's types follow
inline mat2 inverse(const mat2& m) {
	const float one_over_det = 1.0f / (m[0][0] * m[1][1] - m[1][0] * m[0][1]);
	mat2 r; r[0] = vec2(m[1][1] * one_over_det, -m[0][1] * one_over_det); r[1] = vec2(-m[1][0] * one_over_det, m[0][0] * one_over_det);
	return r;
}
// Morton code of a 10-bit cell coordinate triple (the bit-interleaving every GPU text book has; tcnn's common_device.h)
inline uint32_t expand_bits(uint32_t v) { v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu; v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u; return v; }
inline uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) { return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2); }
inline uint32_t morton3D_invert(uint32_t x) { x = x & 0x49249249u; x = (x | (x >> 2)) & 0xc30c30c3u; x = (x | (x >> 4)) & 0x0f00f00fu; x = (x | (x >> 8)) & 0xff0000ffu; x = (x | (x >> 16)) & 0x0000ffffu; return x; }
// camera interpolation (rolling shutter / motion blur): tcnn's own mat_log / mat_exp / slerp are not reproducible from the mount -- NOT pinned, calling them aborts
[[noreturn]] inline void ngp_shim_unpinned(const char* what) { std::fprintf(stderr, "oracle/ref_shim: %s is tcnn code that is absent from the mount\n", what); std::abort(); }
inline tmat<float, 4, 4> inverse(const tmat<float, 4, 4>&) { ngp_shim_unpinned("inverse(mat4)"); }
inline tmat<float, 4, 4> mat_log(const tmat<float, 4, 4>&) { ngp_shim_unpinned("mat_log"); }
inline tmat<float, 4, 4> mat_exp(const tmat<float, 4, 4>&) { ngp_shim_unpinned("mat_exp"); }
inline mat3 mat_log(const mat3&) { ngp_shim_unpinned("mat_log"); }
inline mat3 mat_exp(const mat3&) { ngp_shim_unpinned("mat_exp"); }
// camera_slerp of a camera WITHOUT motion (start == end, every shipped dataset): the matrix itself -- the assumption the oracle and the HIP path make as well (tcnn routes
// through quaternions; whether that round trip returns the input bits cannot be known from the mount).  Cameras with motion: not pinned here.
inline mat3 slerp(const mat3& a, const mat3& b, float) { if (a == b) return a; ngp_shim_unpinned("slerp of two different rotations"); }
template <uint32_t N> tvec<float, N> tan(const tvec<float, N>& a) { tvec<float, N> r; for (uint32_t i = 0; i < N; ++i) r[i] = std::tan(a[i]); return r; }
template <uint32_t N> tvec<float, N> atan(const tvec<float, N>& a) { tvec<float, N> r; for (uint32_t i = 0; i < N; ++i) r[i] = std::atan(a[i]); return r; }
// a ray as the reference's headers use it (origin, direction, point at t, validity = non-zero direction)
struct Ray {
	vec3 o; vec3 d;
	vec3 operator()(float t) const { return o + t * d; }
	void advance(float t) { o += d * t; }
	float distance_to(const vec3& p) const { vec3 nearest = p - o; nearest -= d * dot(nearest, d) / length2(d); return length(nearest); }
	bool is_valid() const { return d != vec3(0.0f); }
	static Ray invalid() { return {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}}; }
};

template <typename T> T div_round_up(T a, T b) { return (a + b - 1) / b; }
template <typename T> T next_multiple(T a, T b) { return div_round_up(a, b) * b; }
template <typename T> void host_device_swap(T& a, T& b) { T c = a; a = b; b = c; }
inline uint32_t lane_id() { return 0; }

} // namespace tcnn
// IEEE binary16 storage type with the two conversions the headers use
struct __half {
	uint16_t bits = 0;
	__half() {}
	__half(float f) { uint32_t x; std::memcpy(&x, &f, 4); const uint32_t sign = (x >> 16) & 0x8000u; int32_t e = (int32_t)((x >> 23) & 0xff) - 127 + 15; uint32_t m = x & 0x7fffffu;
		if (((x >> 23) & 0xff) == 0xff) { bits = (uint16_t)(sign | 0x7c00u | (m ? 0x200u : 0)); return; }
		if (e >= 31) { bits = (uint16_t)(sign | 0x7c00u); return; }
		if (e <= 0) { if (e < -10) { bits = (uint16_t)sign; return; } m |= 0x800000u; const int sh = 14 - e; uint32_t h = m >> sh; const uint32_t rem = m & ((1u << sh) - 1u), half = 1u << (sh - 1); if (rem > half || (rem == half && (h & 1))) ++h; bits = (uint16_t)(sign | h); return; }
		uint32_t h = ((uint32_t)e << 10) | (m >> 13); const uint32_t rem = m & 0x1fffu; if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h; bits = (uint16_t)(sign | h); }
	operator float() const { const uint32_t sign = (uint32_t)(bits & 0x8000u) << 16; uint32_t e = (bits >> 10) & 31u, m = bits & 1023u, x;
		if (e == 0) { if (m == 0) x = sign; else { int sh = 0; while (!(m & 0x400u)) { m <<= 1; ++sh; } m &= 1023u; x = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13); } }
		else if (e == 31) x = sign | 0x7f800000u | (m << 13); else x = sign | ((e - 15 + 127) << 23) | (m << 13);
		float f; std::memcpy(&f, &x, 4); return f; }
};
// "atomics" of a one-thread-at-a-time CPU run of the reference's kernels (ref_nerf_kernels_wrapper): plain read-modify-write, returning the old value like CUDA's
template <typename T, typename U> inline T atomicAdd(T* p, U v) { const T old = *p; *p = (T)(old + (T)v); return old; }
template <typename T, typename U> inline T atomicMax(T* p, U v) { const T old = *p; if ((T)v > old) *p = (T)v; return old; }
template <typename T, typename U> inline T atomicMin(T* p, U v) { const T old = *p; if ((T)v < old) *p = (T)v; return old; }
inline __half operator*(__half a, __half b) { return __half((float)a * (float)b); }
inline __half operator+(__half a, __half b) { return __half((float)a + (float)b); }
namespace tcnn {
using network_precision_t = __half; // TCNN_HALF_PRECISION builds (every GPU the reference's README lists as supported at full speed)
// tcnn's pitched pointer (common.h): rows of `stride_in_bytes` bytes; operator()(row) addresses a row, += / -= move by rows
template <typename T> struct PitchedPtr {
	PitchedPtr() : ptr{nullptr}, stride_in_bytes{sizeof(T)} {}
	PitchedPtr(T* ptr, size_t stride_in_elements, size_t offset = 0, size_t extra_stride_bytes = 0) : ptr{ptr + offset}, stride_in_bytes{stride_in_elements * sizeof(T) + extra_stride_bytes} {}
	template <typename U> explicit PitchedPtr(PitchedPtr<U> other) : ptr{(T*)other.ptr}, stride_in_bytes{other.stride_in_bytes} {}
	T* operator()(uint32_t y) const { return (T*)((const char*)ptr + y * stride_in_bytes); }
	void operator+=(uint32_t y) { ptr = (T*)((const char*)ptr + y * stride_in_bytes); }
	void operator-=(uint32_t y) { ptr = (T*)((const char*)ptr - y * stride_in_bytes); }
	explicit operator bool() const { return ptr; }
	T* ptr; size_t stride_in_bytes;
};
} // namespace tcnn
struct ngp_shim_dim3 { uint32_t x = 0, y = 0, z = 0; };
static const ngp_shim_dim3 threadIdx, blockDim{1, 1, 1}, gridDim{1, 1, 1};
static thread_local ngp_shim_dim3 blockIdx; // written by the CPU stand-in for linear_kernel (gpu_memory.h here)
inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }
// CUDA builtins the host-compilable inline functions mention
using tcnn::min; using tcnn::max;
inline float __expf(float x) { return std::exp(x); }
inline float __logf(float x) { return std::log(x); }
inline float __powf(float x, float y) { return std::pow(x, y); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline float __fdividef(float a, float b) { return a / b; }
inline float __saturatef(float x) { return std::fmin(std::fmax(x, 0.0f), 1.0f); }
inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
