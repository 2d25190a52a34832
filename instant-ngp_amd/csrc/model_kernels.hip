// model_kernels.hip -- the fused multiresolution-hash-grid + tiny-MLP kernels for gfx950 (CDNA4).
//
// Replaces tiny-cuda-nn's GridEncoding + FullyFusedMLP as consumed by NerfNetwork
// (reference nerf_network.h:105-139 inference, :145-268 forward/backward, :270-280 density;
// call sites testbed_nerf.cu:3235, 3313-3323, 2570, 1772) and Trainer::optimizer_step (:2770).
//
// Design (MI355X-first, not a translation of tcnn's 32-lane WMMA tiling):
//  * One 64-lane wavefront owns 64 samples = two 32-column MFMA tiles.  All layers are evaluated in
//    TRANSPOSED form  H^T = W * X^T  with v_mfma_f32_32x32x16_f16: weights are the A operand (rows =
//    neurons), samples are the B operand / the lane dimension.  The C/D layout of a 32x32 tile
//    (lane -> column = sample, register r -> row (r&3)+8*(r>>2)+4*(lane>>5)) is *already* a valid B
//    operand for the next layer if the contraction index is permuted by
//        k(s, hi, j) = 16*s + 8*(j>>2) + 4*hi + (j&3)            (s = k-step, hi = lane>>5, j = 0..7)
//    so activations never leave registers: no LDS round trip, no barrier between layers.  The weights
//    are kept in HBM/LDS pre-permuted in that "fragment order" (one 16-byte chunk per lane per
//    fragment, refreshed by the optimizer kernel), so an A operand is a conflict-free ds_read_b128.
//  * The hash-grid lookup is fused in front: lane (n, hi) gathers the levels whose features land in
//    its own B-operand slots (levels 4s+2g+hi for F=4), i.e. the encoding is produced directly in MFMA
//    operand registers.  Features accumulate with packed half FMAs (v_pk_fma_f16), matching the
//    reference's `fma((T)weight, grid_val, result)`.
//  * Swapping the MFMA operands yields the transposed product (lane = neuron, registers = samples),
//    which is exactly the operand layout of the weight-gradient GEMM dW = dY * X^T (contraction over
//    samples).  The backward therefore needs no LDS transposes either.  Production (base.json's shape, round 5): ONE kernel,
//    k_train_fused -- two roles per SIMD, each recomputing the part of the chain it needs: role A the density network's
//    weight gradients + dL/d(enc), role B the colour network's; every other shape: kernel T1 (k_train_fwd_bwd: forward + dgrad,
//    low VGPR) and kernel W (k_wgrad2 / k_wgrad_nr: recompute + wgrad) as separate launches.
//  * ReLU state is kept as 1 bit per neuron (one VGPR per layer per tile) instead of saved activations.
//  * Hash-table gradients: EVERY level goes through record lists (k_grad_bin: counting sort by table chunk in LDS; k_grad_accumulate:
//    exact 64-bit fixed-point sums in LDS, each gradient written once, the optimizer applied in the epilogue on one GPU) -- no global
//    atomics on the production path.  Packed-half atomics (global_atomic_pk_add_f16, the reference's atomicAdd(__half2)) remain as the
//    fallback for table sizes the lists do not cover, for overflowing lists, and as the ablation DBG_T1_NO_BINNING.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "ngp_kernels.hpp"
#include <algorithm>

namespace ngp {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define DEV static __device__ __forceinline__

// fragment indices (see header of this file / DESIGN.md)
// density net: D1 (32 -> 64), D2 (64 -> 16); colour net: R1 (32 -> 64), NR - 1 layers R2 (64 -> 64, 8 fragments each), R3 (64 -> 16).
// FW_R3 / BW_R3 are the positions for NR = 2 hidden colour layers (configs/nerf/base.json); fw_r3(NR) / bw_r3(NR) in general.
constexpr int FW_D1 = 0, FW_D2 = 4, FW_R1 = 8, FW_R2 = 12, FW_R3 = 20;
constexpr int BW_D1 = 0, BW_D2 = 4, BW_R1 = 6, BW_R2 = 10, BW_R3 = 18;
constexpr int fw_r3(int nr) { return FW_R2 + 8 * (nr - 1); }
constexpr int bw_r3(int nr) { return BW_R2 + 8 * (nr - 1); }
constexpr int n_fw(int nr) { return fw_r3(nr) + 4; }
constexpr int n_bw(int nr) { return bw_r3(nr) + 2; }

DEV f16v zero16() { f16v z; for (int i = 0; i < 16; ++i) z[i] = 0.f; return z; }
DEV h8 zero8() { h8 z; for (int i = 0; i < 8; ++i) z[i] = (_Float16)0.f; return z; }
DEV f16v mfma(h8 a, h8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
DEV h8 lds_frag(const h8* frags, int idx, int lane) { return frags[idx * 64 + lane]; }

// D tile (fp32, regs 8q..8q+7) -> half fragment, optional ReLU; returns the ReLU bit mask (bit j)
template <bool RELU>
DEV h8 to_frag(const f16v& d, int q, uint32_t& mask_bits) {
	h8 r;
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		float v = d[8 * q + j];
		if (RELU) v = v > 0.f ? v : 0.f;
		r[j] = (_Float16)v;
		// the ReLU state is that of the STORED (half) activation, as in tcnn's backward and the oracle (oracle/ora_model.hpp: `!(h2f(in[k]) > 0.f)`): a pre-activation in
		// (0, 2^-25) rounds to a zero activation and passes no gradient (rounds 1-4 tested the fp32 value; k_train_fused's packed conversions test the half as well)
		if (RELU) mask_bits |= ((float)r[j] > 0.f ? 1u : 0u) << (8 * q + j);
	}
	return r;
}
DEV h8 to_frag_masked(const f16v& d, int q, uint32_t mask16) {
	h8 r;
#pragma unroll
	for (int j = 0; j < 8; ++j) r[j] = ((mask16 >> (8 * q + j)) & 1u) ? (_Float16)d[8 * q + j] : (_Float16)0.f;
	return r;
}

// ---------------------------------------------------------------------------------------------
// hash grid: one level for one sample.  [tcnn grid.h kernel_grid / grid_index / pos_fract]
// ---------------------------------------------------------------------------------------------
struct LevelConst { float scale; uint32_t res, hs, offset; bool hashed; };
// pieces of a NerfCoordinate (28-byte records: 4-byte aligned only) as ONE 16- / 12-byte access instead of dwords at a 28-byte stride
typedef float f4u_t __attribute__((ext_vector_type(4), aligned(4)));
typedef float f3u_t __attribute__((ext_vector_type(3), aligned(4)));
// the lane's level is one of two compile-time levels, selected by hi (= lane >> 5): a select between two uniform values
DEV LevelConst level_const2(const GridMeta* __restrict__ gmp, int lvl_lo, int lvl_hi, int hi) {
	const GridMeta& gm = *gmp;
	LevelConst c;
	c.scale = hi ? gm.scale[lvl_hi] : gm.scale[lvl_lo];
	c.res = hi ? gm.resolution[lvl_hi] : gm.resolution[lvl_lo];
	c.hs = hi ? gm.hashmap_size[lvl_hi] : gm.hashmap_size[lvl_lo];
	c.offset = hi ? gm.offset[lvl_hi] : gm.offset[lvl_lo];
	c.hashed = (uint64_t)c.res * c.res * c.res > (uint64_t)c.hs;
	return c;
}
DEV LevelConst level_const(const GridMeta* __restrict__ gmp, int lvl_even, int hi) { return level_const2(gmp, lvl_even, lvl_even + 1, hi); }
DEV LevelConst level_const_uniform(const GridMeta* __restrict__ gm, uint32_t level) {
	LevelConst lc;
	lc.scale = gm->scale[level]; lc.res = gm->resolution[level]; lc.hs = gm->hashmap_size[level]; lc.offset = gm->offset[level];
	lc.hashed = (uint64_t)lc.res * lc.res * lc.res > (uint64_t)lc.hs;
	return lc;
}
struct Corners { uint32_t idx[8]; float w[8]; uint32_t cell_xy, cell_z; };
DEV void level_corners(const LevelConst& lc, float x, float y, float z, Corners& out) {
	// No contraction here: under -ffp-contract=fast LLVM rewrites (1 - p) * w into fma(-p, w, w), which changes the
	// interpolation weights by an ulp relative to the reference's two-step rounding (only the explicit fmaf is fused).
#pragma clang fp contract(off)
	float p0 = fmaf(lc.scale, x, 0.5f), p1 = fmaf(lc.scale, y, 0.5f), p2 = fmaf(lc.scale, z, 0.5f);
	float f0 = floorf(p0), f1 = floorf(p1), f2 = floorf(p2);
	uint32_t g0 = (uint32_t)(int)f0, g1 = (uint32_t)(int)f1, g2 = (uint32_t)(int)f2;
	p0 -= f0; p1 -= f1; p2 -= f2;
	out.cell_xy = g0 | (g1 << 16); out.cell_z = g2;
#pragma unroll
	for (int c = 0; c < 8; ++c) {
		float w = 1.f;
		uint32_t a0, a1, a2;
		if ((c & 1) == 0) { w *= 1 - p0; a0 = g0; } else { w *= p0; a0 = g0 + 1; }
		if ((c & 2) == 0) { w *= 1 - p1; a1 = g1; } else { w *= p1; a1 = g1 + 1; }
		if ((c & 4) == 0) { w *= 1 - p2; a2 = g2; } else { w *= p2; a2 = g2 + 1; }
		// materialise the fp32 weight: otherwise the last multiply is folded into the half conversion
		// (v_fma_mixlo_f16, one rounding) while the reference rounds to fp32 first and then to half
		asm volatile("" : "+v"(w));
		uint32_t idx;
		if (lc.hashed) {
			idx = (a0 * 1u) ^ (a1 * 2654435761u) ^ (a2 * 805459861u);
			idx = idx & (lc.hs - 1u); // a hashed level always has hs = 2^log2_hashmap_size, so & == %
		} else {
			idx = a0 + a1 * lc.res + a2 * lc.res * lc.res; // positions in [0, 1]: < 2*hs, one conditional subtract == % hs
			if (idx >= lc.hs) { idx -= lc.hs; if (idx >= lc.hs) idx %= lc.hs; } // anything else (out-of-range / NaN input) wraps like tcnn's `% hashmap_size`: never out of bounds
		}
		out.idx[c] = idx; out.w[c] = w;
	}
}
// F = 4: returns the 4 interpolated features of one level (half fma accumulation, corner order 0..7).
// PAIR (ablation, off): the x-adjacent corners (c, c^1) of a sample are neighbouring 8-byte entries on dense levels (and
// for even x on hashed ones), so a lane pair (L, L^1) can load {L's c, L's c^1} in ONE instruction and hand the values
// back with shuffles. Unlike for the atomics of the backward pass this does NOT pay for loads (0.262 vs 0.233 ms for K2):
// the gathers are L2/MALL hits that coalesce well enough, and the shuffles + 40 extra VGPRs cost more than they save.
template <bool PAIR>
DEV h4 level_features4(const __half* __restrict__ table, const LevelConst& lc, float x, float y, float z, int lane) {
	Corners cr;
	level_corners(lc, x, y, z, cr);
	const uint2* t = (const uint2*)table + lc.offset;
	uint2 v[8];
	if (PAIR) {
		const bool odd = (lane & 1) != 0;
#pragma unroll
		for (int c = 0; c < 8; c += 2) {
			// what the partner lane needs from me: my idx[c] if I am odd (it loads my corner c), my idx[c+1] if I am even
			const uint32_t idx_nb = (uint32_t)__shfl_xor((int)(odd ? cr.idx[c] : cr.idx[c + 1]), 1, 64);
			const uint2 A = t[odd ? idx_nb : cr.idx[c]];       // step A: the even lane's corner pair {c, c+1}
			const uint2 B = t[odd ? cr.idx[c + 1] : idx_nb];   // step B: the odd lane's corner pair {c, c+1}
			// even keeps A (own c) and needs own c+1 = partner's A; odd keeps B (own c+1) and needs own c = partner's B
			const uint32_t sx = odd ? A.x : B.x, sy = odd ? A.y : B.y;
			const uint32_t rx = (uint32_t)__shfl_xor((int)sx, 1, 64), ry = (uint32_t)__shfl_xor((int)sy, 1, 64);
			if (odd) { v[c].x = rx; v[c].y = ry; v[c + 1] = B; }
			else { v[c] = A; v[c + 1].x = rx; v[c + 1].y = ry; }
		}
	} else {
#pragma unroll
		// (round 4: the same gathers with the nt policy -- global_load ... nt, L1 bypassed -- make K2 0.133 -> 0.21 ms: the x-adjacent corner pairs and the coarse
		// levels live on L1 hits; profiles/r04_microbench_k2_nt_gathers_slower.log)
		for (int c = 0; c < 8; ++c) v[c] = t[cr.idx[c]];  // 8 independent 8-byte gathers in flight
	}
	h2 r0 = {(_Float16)0.f, (_Float16)0.f}, r1 = r0;
#pragma unroll
	for (int c = 0; c < 8; ++c) {
		_Float16 wh = (_Float16)cr.w[c];
		h2 w2 = {wh, wh};
		h2 a = __builtin_bit_cast(h2, v[c].x), b = __builtin_bit_cast(h2, v[c].y);
		r0 = __builtin_elementwise_fma(w2, a, r0);
		r1 = __builtin_elementwise_fma(w2, b, r1);
	}
	h4 r = {r0[0], r0[1], r1[0], r1[1]};
	// keep the scheduler from hoisting the next level's 8 gathers above this point: 8 loads per wave in
	// flight x 16 waves/CU already exceeds what the memory pipeline tracks, and hoisting all 32-64 gathers
	// costs >64 VGPRs (= occupancy, which is what actually hides the gather latency).
	__builtin_amdgcn_sched_barrier(0);
	return r;
}

// F = 2 (L = 16: the reference's 2022 configs/nerf/base.json, notebooks/instant_ngp.ipynb): 4-byte entries, same corner order and half fma chain
DEV h2 level_features2_3d(const __half* __restrict__ table, const LevelConst& lc, float x, float y, float z) {
	Corners cr;
	level_corners(lc, x, y, z, cr);
	const uint32_t* t = (const uint32_t*)table + lc.offset;
	uint32_t v[8];
#pragma unroll
	for (int c = 0; c < 8; ++c) v[c] = t[cr.idx[c]];
	h2 r = {(_Float16)0.f, (_Float16)0.f};
#pragma unroll
	for (int c = 0; c < 8; ++c) {
		const _Float16 wh = (_Float16)cr.w[c];
		const h2 w2 = {wh, wh};
		r = __builtin_elementwise_fma(w2, __builtin_bit_cast(h2, v[c]), r);
	}
	__builtin_amdgcn_sched_barrier(0); // see level_features4
	return r;
}

// F = 4, two levels' gathers in flight (16 independent loads per lane between waits instead of 8): the per-tile chain of the lazy K2 is position load -> four dependent gather
// rounds -> MFMA chain, and the kernel keeps only 3 wavefronts per SIMD busy; this halves the dependent rounds.  Same loads, same half fma chain per level (corner order 0..7).
DEV void level_features4_x2(const __half* __restrict__ table, const LevelConst& la, const LevelConst& lb, float x, float y, float z, h4& ra, h4& rb) {
	Corners ca, cb;
	level_corners(la, x, y, z, ca);
	level_corners(lb, x, y, z, cb);
	const uint2* ta = (const uint2*)table + la.offset; const uint2* tb = (const uint2*)table + lb.offset;
	uint2 va[8], vb[8];
#pragma unroll
	for (int c = 0; c < 8; ++c) va[c] = ta[ca.idx[c]];
#pragma unroll
	for (int c = 0; c < 8; ++c) vb[c] = tb[cb.idx[c]];
	h2 a0 = {(_Float16)0.f, (_Float16)0.f}, a1 = a0, b0 = a0, b1 = a0;
#pragma unroll
	for (int c = 0; c < 8; ++c) {
		const _Float16 wh = (_Float16)ca.w[c]; const h2 w2 = {wh, wh};
		a0 = __builtin_elementwise_fma(w2, __builtin_bit_cast(h2, va[c].x), a0);
		a1 = __builtin_elementwise_fma(w2, __builtin_bit_cast(h2, va[c].y), a1);
	}
#pragma unroll
	for (int c = 0; c < 8; ++c) {
		const _Float16 wh = (_Float16)cb.w[c]; const h2 w2 = {wh, wh};
		b0 = __builtin_elementwise_fma(w2, __builtin_bit_cast(h2, vb[c].x), b0);
		b1 = __builtin_elementwise_fma(w2, __builtin_bit_cast(h2, vb[c].y), b1);
	}
	ra = h4{a0[0], a0[1], a1[0], a1[1]}; rb = h4{b0[0], b0[1], b1[0], b1[1]};
	__builtin_amdgcn_sched_barrier(0);
}
// (Round 6 also tried x-adjacent corner pairs as ONE 16-byte gather per lane where the table makes them neighbours -- always on dense levels, for even x on hashed ones --
// with an exec-masked 8-byte load for the rest: 168 registers + 228 B of scratch and K2 106 -> 193 us, profiles/r06_ab_k2_pair_gathers_slower.txt.  Not kept; rounds 1 and 5
// had measured the lane-pair + shuffle form and the L1-resident case with the same verdict.)
// The level constants from a 16-byte-per-level LDS table {scale, resolution | hashed << 31, hashmap_size, offset} (fill_level_table) instead of from GridMeta in global memory:
// level_const's `hi ? gm.x[l + 1] : gm.x[l]` compiles to four VECTOR global loads per level (the address depends on the lane), i.e. a second memory round trip in front
// of every level's gathers -- the lazy K2 paid eight dependent round trips per tile where four are needed (round 6).
DEV void fill_level_table(uint4* lct, const GridMeta* __restrict__ gm) {
	const uint32_t l = threadIdx.x;
	if (l < 16u) {
		uint4 v = make_uint4(0u, 0u, 0u, 0u);
		if (l < gm->n_levels) {
			const uint32_t res = gm->resolution[l], hs = gm->hashmap_size[l];
			v = make_uint4(__float_as_uint(gm->scale[l]), res | (((uint64_t)res * res * res > (uint64_t)hs) ? 0x80000000u : 0u), hs, gm->offset[l]);
		}
		lct[l] = v;
	}
}
DEV LevelConst level_const_lds(const uint4* lct, int level) {
	const uint4 v = lct[level];
	LevelConst c; c.scale = __uint_as_float(v.x); c.res = v.y & 0x7fffffffu; c.hashed = (v.y >> 31) != 0u; c.hs = v.z; c.offset = v.w;
	return c;
}
template <int DEPTH>
DEV void encode_sample_lds(const uint4* lct, const __half* __restrict__ table, float x, float y, float z, int hi, h8 out[2]) {
	const int lane = threadIdx.x & 63;
#pragma unroll
	for (int s = 0; s < 2; ++s) {
		h4 a, b;
		if constexpr (DEPTH == 2) level_features4_x2(table, level_const_lds(lct, 4 * s + 0 + hi), level_const_lds(lct, 4 * s + 2 + hi), x, y, z, a, b);
		else { a = level_features4<false>(table, level_const_lds(lct, 4 * s + 0 + hi), x, y, z, lane); b = level_features4<false>(table, level_const_lds(lct, 4 * s + 2 + hi), x, y, z, lane); }
		out[s] = h8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
	}
}

// Encoding of one sample column into the lane's two B-operand fragments (k-steps 0,1).  Fragment element (s, hi, j) is input feature
// k = 16 s + 8 (j >> 2) + 4 hi + (j & 3):  F = 4: feature j & 3 of level 4 s + 2 (j >> 2) + hi;  F = 2: feature j & 1 of level 8 s + 4 (j >> 2) + 2 hi + ((j & 3) >> 1).
template <int F = 4, bool PAIR = false>
DEV void encode_sample(const GridMeta* __restrict__ gm, const __half* __restrict__ table, float x, float y, float z, int hi, h8 out[2]) {
	const int lane = threadIdx.x & 63;
#pragma unroll
	for (int s = 0; s < 2; ++s) {
		if (F == 4) {
			h4 a = level_features4<PAIR>(table, level_const(gm, 4 * s + 0, hi), x, y, z, lane); // level 4s+hi   -> j = 0..3
			h4 b = level_features4<PAIR>(table, level_const(gm, 4 * s + 2, hi), x, y, z, lane); // level 4s+2+hi -> j = 4..7
			h8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
			out[s] = r;
		} else {
			h8 r;
#pragma unroll
			for (int q = 0; q < 4; ++q) { // q = j >> 1
				const int l0 = 8 * s + 4 * (q >> 1) + (q & 1);
				// 16 levels x 4 per-lane constants would stay live across the whole sample loop (they are loop invariant: 32 VGPRs, spills);
				// an opaque copy of `hi` makes the four selects of a level part of the level's own code instead
				int hi_l = hi; asm volatile("" : "+v"(hi_l));
				const h2 f = level_features2_3d(table, level_const2(gm, l0, l0 + 2, hi_l), x, y, z);
				r[2 * q] = f[0]; r[2 * q + 1] = f[1];
			}
			out[s] = r;
		}
	}
}

// [tcnn spherical_harmonics.h] degree 4, d in [0,1]^3; the lane keeps SH indices 8*(j>>2)+4*hi+(j&3)
DEV h8 sh4_frag(float dx, float dy, float dz, int hi) {
	const float x = dx * 2.f - 1.f, y = dy * 2.f - 1.f, z = dz * 2.f - 1.f;
	const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
	float o[16];
	o[0] = 0.28209479177387814f;
	o[1] = -0.48860251190291987f * y;
	o[2] = 0.48860251190291987f * z;
	o[3] = -0.48860251190291987f * x;
	o[4] = 1.0925484305920792f * xy;
	o[5] = -1.0925484305920792f * yz;
	o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
	o[7] = -1.0925484305920792f * xz;
	o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
	o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
	o[10] = 2.8906114426405538f * xy * z;
	o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
	o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
	o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
	o[14] = 1.4453057213202769f * z * (x2 - y2);
	o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
	h8 r;
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		const int lo = 8 * (j >> 2) + (j & 3); // index for hi = 0; hi = 1 adds 4
		r[j] = (_Float16)(hi ? o[lo + 4] : o[lo]);
	}
	return r;
}

// Extra (latent / light-direction) dims of the dir encoding (nerf_network.h:84, Composite: Identity over the dims behind the direction): third k-step of the colour
// network's first layer.  xp = the sample's n_extra floats behind its NerfCoordinate (in + 7); the lane keeps k-slots 8*(j>>2)+4*hi+(j&3); [tcnn] padding columns = 1.
DEV h8 extra_frag(const float* __restrict__ xp, uint32_t n_extra, int hi) {
	h8 r;
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		const uint32_t k = 8u * (j >> 2) + 4u * hi + (j & 3);
		r[j] = k < n_extra ? (_Float16)xp[k] : (_Float16)1.f;
	}
	return r;
}
// fragment indices of a model with extra dims: behind the regular ones, so that every other index is the same in both kinds of model
// fw: n_fw(NR) + mt (k-step 2 of R1, mt = 0, 1);  bw: n_bw(NR) + s (dgrad rows 32..63 of R1^T -- 32..47 are the extra dims --, k-steps s = 0..3)
constexpr int N_FW_EXTRA = 2, N_BW_EXTRA = 4;

// copy the fragment-ordered weights into LDS (all threads of the block)
// (loads are issued in batches of 8 / 4 before their LDS stores: one load per loop trip made every block start with 5 - 10 dependent round trips to L2)
DEV void load_frags_to_lds(h8* dst, const ngp_half* __restrict__ src, int n_frags) {
	const uint4* s = (const uint4*)src;
	uint4* d = (uint4*)dst;
	const int n = n_frags * 64, step = (int)blockDim.x;
	int i = (int)threadIdx.x;
	for (; i + 7 * step < n; i += 8 * step) {
		uint4 r[8];
#pragma unroll
		for (int k = 0; k < 8; ++k) r[k] = s[i + k * step];
#pragma unroll
		for (int k = 0; k < 8; ++k) d[i + k * step] = r[k];
	}
	for (; i + 3 * step < n; i += 4 * step) {
		uint4 r[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) r[k] = s[i + k * step];
#pragma unroll
		for (int k = 0; k < 4; ++k) d[i + k * step] = r[k];
	}
	for (; i < n; i += step) d[i] = s[i];
}

// ---------------------------------------------------------------------------------------------
// forward chain shared by inference / T1 / W.  CT = number of 32-sample column tiles per wave.
// ---------------------------------------------------------------------------------------------
template <int CT>
struct FwdState {
	h8 enc[CT][2];      // encoding fragments (k-steps 0,1)
	h8 rin[CT][3];      // rgb-net input: [0] = density-net output (16), [1] = SH (16), [2] (models with extra dims only) = the extra dims + padding ones (16)
	h8 hb[CT][4];       // current 64-wide hidden activation fragments
	uint32_t m1d[CT], m1r[CT], m2r[2][CT]; // ReLU bit masks: bit (16*mt + r); m2r[k] = the k-th 64 x 64 colour layer
	float sigma[CT];    // density logit (valid on hi == 0 lanes)
};

template <int CT>
DEV void fwd_density_l1(const h8* fw, int lane, FwdState<CT>& st) {
	f16v acc[2][CT];
#pragma unroll
	for (int mt = 0; mt < 2; ++mt)
#pragma unroll
		for (int c = 0; c < CT; ++c) acc[mt][c] = zero16();
#pragma unroll
	for (int mt = 0; mt < 2; ++mt)
#pragma unroll
		for (int s = 0; s < 2; ++s) {
			h8 a = lds_frag(fw, FW_D1 + mt * 2 + s, lane);
#pragma unroll
			for (int c = 0; c < CT; ++c) acc[mt][c] = mfma(a, st.enc[c][s], acc[mt][c]);
		}
#pragma unroll
	for (int c = 0; c < CT; ++c) {
		st.m1d[c] = 0;
#pragma unroll
		for (int mt = 0; mt < 2; ++mt) {
			uint32_t mb = 0;
			st.hb[c][2 * mt + 0] = to_frag<true>(acc[mt][c], 0, mb);
			st.hb[c][2 * mt + 1] = to_frag<true>(acc[mt][c], 1, mb);
			st.m1d[c] |= mb << (16 * mt);
		}
	}
}
template <int CT>
DEV void fwd_density_l2(const h8* fw, int lane, FwdState<CT>& st) {
	f16v acc[CT];
#pragma unroll
	for (int c = 0; c < CT; ++c) acc[c] = zero16();
#pragma unroll
	for (int s = 0; s < 4; ++s) {
		h8 a = lds_frag(fw, FW_D2 + s, lane);
#pragma unroll
		for (int c = 0; c < CT; ++c) acc[c] = mfma(a, st.hb[c][s], acc[c]);
	}
#pragma unroll
	for (int c = 0; c < CT; ++c) {
		uint32_t dummy = 0;
		st.rin[c][0] = to_frag<false>(acc[c], 0, dummy);
		st.sigma[c] = (float)st.rin[c][0][0]; // neuron 0 lives in reg 0 of the hi == 0 lanes (half-rounded)
	}
}
template <int CT, int EX = 0>
DEV void fwd_rgb_l1(const h8* fw, int lane, FwdState<CT>& st, const int xfrag = 0 /* EX: index of the first extra fragment */) {
	f16v acc[2][CT];
#pragma unroll
	for (int mt = 0; mt < 2; ++mt)
#pragma unroll
		for (int c = 0; c < CT; ++c) acc[mt][c] = zero16();
#pragma unroll
	for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
		for (int s = 0; s < 2; ++s) {
			h8 a = lds_frag(fw, FW_R1 + mt * 2 + s, lane);
#pragma unroll
			for (int c = 0; c < CT; ++c) acc[mt][c] = mfma(a, st.rin[c][s], acc[mt][c]);
		}
		if constexpr (EX != 0) {
			h8 a = lds_frag(fw, xfrag + mt, lane);
#pragma unroll
			for (int c = 0; c < CT; ++c) acc[mt][c] = mfma(a, st.rin[c][2], acc[mt][c]);
		}
	}
#pragma unroll
	for (int c = 0; c < CT; ++c) {
		st.m1r[c] = 0;
#pragma unroll
		for (int mt = 0; mt < 2; ++mt) {
			uint32_t mb = 0;
			st.hb[c][2 * mt + 0] = to_frag<true>(acc[mt][c], 0, mb);
			st.hb[c][2 * mt + 1] = to_frag<true>(acc[mt][c], 1, mb);
			st.m1r[c] |= mb << (16 * mt);
		}
	}
}
template <int CT>
DEV void fwd_rgb_l2(const h8* fw, int lane, FwdState<CT>& st, const int k = 0 /* which of the 64 x 64 layers */) {
	f16v acc[2][CT];
#pragma unroll
	for (int mt = 0; mt < 2; ++mt)
#pragma unroll
		for (int c = 0; c < CT; ++c) acc[mt][c] = zero16();
#pragma unroll
	for (int mt = 0; mt < 2; ++mt)
#pragma unroll
		for (int s = 0; s < 4; ++s) {
			h8 a = lds_frag(fw, FW_R2 + 8 * k + mt * 4 + s, lane);
#pragma unroll
			for (int c = 0; c < CT; ++c) acc[mt][c] = mfma(a, st.hb[c][s], acc[mt][c]);
		}
#pragma unroll
	for (int c = 0; c < CT; ++c) {
		st.m2r[k][c] = 0;
#pragma unroll
		for (int mt = 0; mt < 2; ++mt) {
			uint32_t mb = 0;
			st.hb[c][2 * mt + 0] = to_frag<true>(acc[mt][c], 0, mb);
			st.hb[c][2 * mt + 1] = to_frag<true>(acc[mt][c], 1, mb);
			st.m2r[k][c] |= mb << (16 * mt);
		}
	}
}
// rgb output layer: returns the D tile (rows 0..2 = rgb logits on hi == 0 lanes, regs 0..2)
template <int CT>
DEV void fwd_rgb_l3(const h8* fw, int lane, const FwdState<CT>& st, f16v out[CT], const int base = FW_R3) {
#pragma unroll
	for (int c = 0; c < CT; ++c) out[c] = zero16();
#pragma unroll
	for (int s = 0; s < 4; ++s) {
		h8 a = lds_frag(fw, base + s, lane);
#pragma unroll
		for (int c = 0; c < CT; ++c) out[c] = mfma(a, st.hb[c][s], out[c]);
	}
}

// the colour network with NR hidden layers (configs/nerf/base_1layer / base(_2layer) / base_3layer.json): hidden activations only
template <int CT, int NR, int EX = 0>
DEV void fwd_rgb_hidden(const h8* fw, int lane, FwdState<CT>& st) {
	fwd_rgb_l1<CT, EX>(fw, lane, st, n_fw(NR));
#pragma unroll
	for (int k = 0; k < NR - 1; ++k) fwd_rgb_l2<CT>(fw, lane, st, k);
}

// ---------------------------------------------------------------------------------------------
// inference kernel (K2, density-grid queries, renderer): persistent waves, 64 samples per iteration
// ---------------------------------------------------------------------------------------------
template <bool DENSITY_ONLY, int CT, bool PAIR, int MINW, int F = 4, int NR = 2, int EX = 0>
__global__ void __launch_bounds__(256, MINW) k_inference(const GridMeta* __restrict__ gm, ModelPtrs mp, const float* __restrict__ in, uint32_t in_stride, uint32_t n_max,
		const uint32_t* __restrict__ n_ptr, __half* __restrict__ out, uint32_t out_stride, uint32_t dir_offset) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	h8* fw = (h8*)smem;
	__shared__ uint4 s_lct[16];
	constexpr bool LCT = F == 4 && !PAIR; // level constants from an LDS table (round 6, see fill_level_table)
	if constexpr (LCT) fill_level_table(s_lct, gm);
	load_frags_to_lds(fw, mp.fw_frags, DENSITY_ONLY ? 8 : n_fw(NR) + (EX ? N_FW_EXTRA : 0));
	__syncthreads();
	const uint32_t n = n_ptr ? min(*n_ptr, n_max) : n_max;
	const int lane = threadIdx.x & 63, col = lane & 31, hi = lane >> 5;
	const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
	const __half* table = (const __half*)mp.grid;
	constexpr uint32_t TS = 32 * CT; // samples per wave iteration
	for (uint32_t tile = wave; (uint64_t)tile * TS < n; tile += n_waves) {
		FwdState<CT> st;
		uint32_t sidx[CT];
#pragma unroll
		for (int c = 0; c < CT; ++c) {
			const uint32_t s_raw = tile * TS + c * 32 + col;
			sidx[c] = s_raw;
			const float* p = in + (size_t)min(s_raw, n - 1) * in_stride;
			if constexpr (LCT) encode_sample_lds<1>(s_lct, table, p[0], p[1], p[2], hi, st.enc[c]);
			else encode_sample<F, PAIR>(gm, table, p[0], p[1], p[2], hi, st.enc[c]);
			if (!DENSITY_ONLY) st.rin[c][1] = sh4_frag(p[dir_offset], p[dir_offset + 1], p[dir_offset + 2], hi);
			if constexpr (!DENSITY_ONLY && EX != 0) st.rin[c][2] = extra_frag(p + dir_offset + 3, mp.n_extra, hi);
		}
		fwd_density_l1<CT>(fw, lane, st);
		fwd_density_l2<CT>(fw, lane, st);
		if (DENSITY_ONLY) {
#pragma unroll
			for (int c = 0; c < CT; ++c)
				if (hi == 0 && sidx[c] < n) out[(size_t)sidx[c] * out_stride] = __float2half(st.sigma[c]);
			continue;
		}
		fwd_rgb_hidden<CT, NR, EX>(fw, lane, st);
		f16v o[CT];
		fwd_rgb_l3<CT>(fw, lane, st, o, fw_r3(NR));
#pragma unroll
		for (int c = 0; c < CT; ++c) {
			if (hi == 0 && sidx[c] < n) {
				h4 r = {(_Float16)o[c][0], (_Float16)o[c][1], (_Float16)o[c][2], (_Float16)st.sigma[c]};
				*(uint2*)(out + (size_t)sidx[c] * out_stride) = __builtin_bit_cast(uint2, r);
			}
		}
	}
}

// (Round 4's level-per-XCD encoding stage in front of the lazy K2 -- k_encode_tiles_xcd: one hash-grid level per XCD, so that its 4 MB table stays in that XCD's L2 -- was
// measured at 0.105 ms against 0.120 ms for the whole fused K2, profiles/r04_microbench_k2_xcd_encode.log, and removed in round 6; git history has the kernel.)
// ---------------------------------------------------------------------------------------------
// Lazy (front-to-back) K2.  The reference evaluates the network on EVERY marched sample and then discards all samples behind
// the point where a ray's transmittance drops below 1e-4 (compute_loss_kernel_train_nerf: `if (T < EPSILON) break`).  Samples
// behind the cut influence nothing, so the network is evaluated in rounds of 32-sample tiles (one tile = 32 consecutive samples of
// ONE ray).  K1 writes the round-0 list (the first tile of every active ray); a tile that leaves its ray still transparent appends
// the ray's next tile to the next round's list (the last round takes everything that is left), with a 1 % safety margin on the
// threshold so that K3's own test can never walk into an unevaluated sample.  Same results as the eager order (DBG_K2_EAGER),
// tests/test_gpu_train.py::test_lazy_k2_matches_eager.
// ---------------------------------------------------------------------------------------------
// TW = tile width in samples: 32 (one tile per wavefront) or 16 (two tiles of two different rays share the wavefront's 32 MFMA
// columns -- rays end after ~12 compacted samples, so 16-wide tiles evaluate fewer samples behind the cut and fill the columns).
template <uint32_t TW, int F = 4, int NR = 2, int DEPTH = 0 /* F = 4: 0 = level constants from GridMeta (rounds 1-5), 1 | 2 = from the LDS table, with one | two levels' gathers in flight */,
	bool DYN = false /* a workgroup owns a contiguous range of tile pairs and its wavefronts take them as they finish (LDS counter) instead of every n_waves-th pair: rays need 1 .. 60+ tiles */>
__global__ void __launch_bounds__(256, 3) k_inference_tiles(const GridMeta* __restrict__ gm, ModelPtrs mp, const float* __restrict__ in, uint32_t in_stride,
		K2LazyArgs la, __half* __restrict__ out, uint32_t out_stride, uint32_t dir_offset) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	constexpr uint32_t TPW = 32u / TW; // tiles per wavefront
	const uint32_t r = la.round;
	const uint32_t n_tiles = min(r == 0 ? *la.n_rays_ptr : la.n_tiles_ptr[r], la.tile_cap);
	h8* fw = (h8*)smem;
	__shared__ uint4 s_lct[16];
	__shared__ uint32_t s_next_wt;
	const uint32_t n_wt = (n_tiles + TPW - 1u) / TPW; // tile pairs (TW = 16) / tiles
	const uint32_t wt_begin = DYN ? (uint32_t)(((uint64_t)n_wt * blockIdx.x) / gridDim.x) : 0u, wt_end = DYN ? (uint32_t)(((uint64_t)n_wt * (blockIdx.x + 1u)) / gridDim.x) : n_wt;
	if (DYN ? wt_begin >= wt_end : blockIdx.x * 4 * TPW >= n_tiles) return; // uniform: nothing for this workgroup (late rounds / the first training steps are small)
	if (DYN && threadIdx.x == 0) s_next_wt = wt_begin + (blockDim.x >> 6);
	if constexpr (DEPTH != 0) fill_level_table(s_lct, gm);
	load_frags_to_lds(fw, mp.fw_frags, n_fw(NR));
	__syncthreads();
	const int lane = threadIdx.x & 63, col = lane & 31, hi = lane >> 5;
	const uint32_t slot = (uint32_t)col / TW, tcol = (uint32_t)col % TW; // which of the wavefront's tiles, column inside it
	const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
	const __half* table = (const __half*)mp.grid;
	const uint4* __restrict__ tiles = la.tiles[r & 1u];
	uint4* __restrict__ next = la.tiles[(r + 1u) & 1u];
	const bool next_is_last = r + 2 == la.n_rounds;
	uint32_t n_eval = 0; // tile leaders: samples evaluated (statistics)
	// Continuations are collected per wavefront and appended to the next round's list in bulk: one returning atomic per flush instead of one
	// per surviving tile (a single counter word retires only ~90 returning atomics per microsecond chip-wide -- MI355X_MICROARCH.md
	// "dequeue" -- which bounded round 0 at ~24k rays: 114 us for 16-wide and 32-wide tiles alike, profiles/r02_kernel_trace_summary_*.txt).
	constexpr uint32_t PEND_CAP = 32;
	__shared__ volatile uint32_t s_pend[4][PEND_CAP][3]; // {first sample of the continuation, samples left, ray}
	const uint32_t wid = threadIdx.x >> 6;
	uint32_t n_pend = 0; // wave-uniform
	auto flush = [&]() {
		const uint32_t e = (uint32_t)lane;
		uint32_t first = 0, rest = 0, ray = 0, nt = 0;
		if (e < n_pend) { first = s_pend[wid][e][0]; rest = s_pend[wid][e][1]; ray = s_pend[wid][e][2]; nt = next_is_last ? (rest + TW - 1u) / TW : 1u; }
		uint32_t incl = nt;
#pragma unroll
		for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, dd, 64); if (lane >= dd) incl += y; }
		const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
		uint32_t off = 0;
		if (lane == 0) off = atomicAdd(la.n_tiles_ptr + r + 1, total);
		off = (uint32_t)__shfl((int)off, 0, 64) + incl - nt;
		const uint32_t n = next_is_last ? rest : min(rest, TW);
		for (uint32_t j = 0; j < nt; ++j)
			if (off + j < la.tile_cap) next[off + j] = make_uint4(first + TW * j, min(TW, n - TW * j), ray, rest - min(rest, TW * (j + 1u)));
		n_pend = 0;
	};
	for (uint32_t wt = DYN ? wt_begin + wid : wave; wt < wt_end; ) {
		const uint32_t wt_cur = wt;
		if constexpr (DYN) { uint32_t nx = 0u; if (lane == 0) nx = atomicAdd(&s_next_wt, 1u); wt = (uint32_t)__builtin_amdgcn_readfirstlane((int)nx); } else wt += n_waves;
		{ const uint32_t wt = wt_cur; // (the body below names the current pair `wt`)
		const uint32_t tile = wt * TPW + slot;
		uint4 d = make_uint4(0u, 0u, 0u, 0u);
		if (tile < n_tiles) d = tiles[tile];
		float T_wave = 1.f; // n_rounds == 1: transmittance behind the tiles this wavefront has evaluated of its ray
		bool first_tile = r == 0; // (wave-uniform) the tile comes from K1's round-0 list
		for (;;) {
		// d.y == 0: a ray K1 dropped at its sample cap (its base may lie outside the buffers) or no tile in this slot: nothing to evaluate
		const bool valid = tcol < d.y;
		const uint32_t sample = valid ? d.x + tcol : 0u;
		if (__ballot(valid) == 0ull) break;
		FwdState<1> st;
		const float* p = in + (size_t)sample * in_stride;
		const f4u_t pc = *(const f4u_t*)p;                 // position + warped dt
		const f3u_t pd = *(const f3u_t*)(p + dir_offset);  // direction
		if constexpr (DEPTH != 0 && F == 4) encode_sample_lds<DEPTH>(s_lct, table, pc[0], pc[1], pc[2], hi, st.enc[0]);
		else encode_sample<F, false>(gm, table, pc[0], pc[1], pc[2], hi, st.enc[0]);
		if (la.enc_out && valid) { // for T1 (EncStashIn): this lane's half of the sample's encoding, 32 contiguous bytes
			uint4* e = la.enc_out + (size_t)sample * 4 + (uint32_t)hi * 2;
			e[0] = __builtin_bit_cast(uint4, st.enc[0][0]); e[1] = __builtin_bit_cast(uint4, st.enc[0][1]);
		}
		first_tile = false;
		st.rin[0][1] = sh4_frag(pd[0], pd[1], pd[2], hi);
		fwd_density_l1<1>(fw, lane, st);
		fwd_density_l2<1>(fw, lane, st);
		fwd_rgb_hidden<1, NR>(fw, lane, st);
		f16v o[1];
		fwd_rgb_l3<1>(fw, lane, st, o, fw_r3(NR));
		if (hi == 0 && valid) {
			h4 rr = {(_Float16)o[0][0], (_Float16)o[0][1], (_Float16)o[0][2], (_Float16)st.sigma[0]};
			*(uint2*)(out + (size_t)sample * out_stride) = __builtin_bit_cast(uint2, rr);
		}
		const bool leader = hi == 0 && tcol == 0;
		if (leader) n_eval += d.y;
		if (la.n_rounds == 1) {
			// One launch, no lists: the wavefront that evaluated a tile of a still transparent ray goes on with the ray's next tile itself.
			// Strictly lazier than the round scheme, whose last round takes everything that is left, and two launches + their ramps and gaps
			// shorter.  TW == 16: the two tiles of the wavefront belong to different rays and continue (or end) independently -- the slot of a
			// finished ray idles (its lanes issue no gathers) until the other one is done.
			float od = 0.f;
			if (hi == 0 && valid) {
				const float x = st.sigma[0];
				const float sg = la.density_activation == NGP_ACT_NONE ? x : la.density_activation == NGP_ACT_RELU ? fmaxf(x, 0.f)
					: la.density_activation == NGP_ACT_LOGISTIC ? 1.f / (1.f + __expf(-x)) : __expf(x);
				od = sg * (pc[3] * la.dt_unwarp_scale + la.dt_unwarp_offset);
			}
#pragma unroll
			for (int dd = (int)TW / 2; dd >= 1; dd >>= 1) od += __shfl_xor(od, dd, 64); // sum over the tile's TW lanes (hi == 0 half)
			T_wave *= __expf(-od);
			const bool cont_l = d.w != 0u && !(T_wave < 0.99e-4f); // meaningful in the tile's leader lane; NaN stays alive, like in K3
			const bool cont = __shfl((int)cont_l, (int)(slot * TW), 64) != 0;
			if (__ballot(cont) == 0ull) break;
			const uint32_t take = cont ? min(d.w, TW) : 0u;
			d = make_uint4(d.x + TW, take, d.z, d.w - take);
			continue;
		}
		// Transmittance behind this tile (an estimate with a safety margin: K3 recomputes the exact compositing).  A ray that is
		// still transparent gets its next tile -- or, if the next round is the last one, all its remaining tiles -- appended to the
		// next round's list: 1 % below K3's threshold, so K3's own test can never walk into an unevaluated sample; NaN stays alive.
		if (r + 1 < la.n_rounds && __ballot(d.w != 0u) != 0ull) {
			float od = 0.f; // optical depth of the lane's sample
			if (hi == 0 && valid) {
				const float x = st.sigma[0];
				const float sg = la.density_activation == NGP_ACT_NONE ? x : la.density_activation == NGP_ACT_RELU ? fmaxf(x, 0.f)
					: la.density_activation == NGP_ACT_LOGISTIC ? 1.f / (1.f + __expf(-x)) : __expf(x);
				od = sg * (pc[3] * la.dt_unwarp_scale + la.dt_unwarp_offset);
			}
#pragma unroll
			for (int dd = (int)TW / 2; dd >= 1; dd >>= 1) od += __shfl_xor(od, dd, 64); // sum over the tile's TW lanes (hi == 0 half)
			bool cont = false;
			if (leader && d.w != 0u) {
				const float T = (r == 0 ? 1.f : la.T_run[d.z]) * __expf(-od);
				if (!(T < 0.99e-4f)) { la.T_run[d.z] = T; cont = true; }
			}
			const uint64_t cm = __ballot(cont);
			if (cm) {
				if (cont) {
					const uint32_t e = n_pend + (uint32_t)__popcll(cm & ((1ull << lane) - 1ull));
					s_pend[wid][e][0] = d.x + TW; s_pend[wid][e][1] = d.w; s_pend[wid][e][2] = d.z;
				}
				n_pend += (uint32_t)__popcll(cm);
				if (n_pend + TPW > PEND_CAP) flush();
			}
		}
		break;
		}
		}
	}
	if (n_pend) flush();
	if (hi == 0 && tcol == 0 && n_eval) atomicAdd(la.n_eval_ptr, n_eval);
}

// ---------------------------------------------------------------------------------------------
// Image / SDF primitives' model (tcnn::NetworkWithInputEncoding: HashGrid L = 16, F = 2 over a 2-D or 3-D position, then a
// FullyFusedMLP 32 -> 64 -> 64 -> 16): forward only.  The MLP has exactly the shape of the NeRF colour network, so the
// fragment layout (FW_R1 / FW_R2 / FW_R3) and the register-resident chain are shared; the encoding again lands directly in the
// lane's B-operand registers: with F = 2, fragment element (s, hi, j) is feature j & 1 of level 8 s + 4 (j >> 2) + 2 hi + ((j & 3) >> 1).
// ---------------------------------------------------------------------------------------------
template <int D> struct CornersND { uint32_t idx[1 << D]; float w[1 << D]; };
// [tcnn grid.h] grid_index + interpolation weights of one level for a D-dimensional position (F = 2 tables)
template <int D>
DEV void level_corners_nd(const LevelConst& lc, const float* __restrict__ x, CornersND<D>& out) {
#pragma clang fp contract(off)
	const float scale = lc.scale;
	const uint32_t res = lc.res, hs = lc.hs;
	uint32_t g[D]; float p[D];
#pragma unroll
	for (int d = 0; d < D; ++d) {
		const float q = fmaf(scale, x[d], 0.5f), f = floorf(q);
		g[d] = (uint32_t)(int)f; p[d] = q - f;
	}
	constexpr int NC = 1 << D;
#pragma unroll
	for (int c = 0; c < NC; ++c) {
		float wc = 1.f;
		uint32_t a[D];
#pragma unroll
		for (int d = 0; d < D; ++d) { if ((c & (1 << d)) == 0) { wc *= 1 - p[d]; a[d] = g[d]; } else { wc *= p[d]; a[d] = g[d] + 1; } }
		asm volatile("" : "+v"(wc)); // keep the fp32 rounding of the weight before the half conversion (see level_corners)
		uint32_t idx;
		if (!lc.hashed) {
			uint32_t stride = 1; idx = 0;
#pragma unroll
			for (int d = 0; d < D; ++d) { idx += a[d] * stride; stride *= res; }
			// [tcnn] `% hashmap_size`: the dense index exceeds the (8-aligned) level size only at the last cell row (positions on the upper faces): one compare instead of a
			// 32-bit division per corner (~30 instructions x 128 corners per 3-D sample; round 6)
			if (idx >= hs) idx %= hs;
		} else {
			const uint32_t primes[3] = {1u, 2654435761u, 805459861u};
			idx = 0;
#pragma unroll
			for (int d = 0; d < D; ++d) idx ^= a[d] * primes[d];
			idx = (hs & (hs - 1u)) == 0u ? idx & (hs - 1u) : idx % hs; // a hashed level has hs = 2^log2_hashmap_size
		}
		out.idx[c] = idx; out.w[c] = wc;
	}
}
template <int D>
DEV LevelConst level_const_nd(const GridMeta* __restrict__ gm, uint32_t level) {
	LevelConst lc; lc.scale = gm->scale[level]; lc.res = gm->resolution[level]; lc.hs = gm->hashmap_size[level]; lc.offset = gm->offset[level];
	uint64_t cells = 1; bool dense = true; // dense iff res^D <= hs
#pragma unroll
	for (int d = 0; d < D; ++d) { cells *= lc.res; if (cells > lc.hs) dense = false; }
	lc.hashed = !dense;
	return lc;
}
template <int D>
DEV void level_corners_nd(const GridMeta* __restrict__ gm, uint32_t level, const float* __restrict__ x, CornersND<D>& out) { level_corners_nd<D>(level_const_nd<D>(gm, level), x, out); }
// (level, hi)-dependent constants of the F = 2 kernels from an LDS table (see fill_level_table; D decides which levels are dense)
template <int D>
DEV void fill_level_table_nd(uint4* lct, const GridMeta* __restrict__ gm) {
	const uint32_t l = threadIdx.x;
	if (l < 16u) {
		uint4 v = make_uint4(0u, 0u, 0u, 0u);
		if (l < gm->n_levels) { const LevelConst lc = level_const_nd<D>(gm, l); v = make_uint4(__float_as_uint(lc.scale), lc.res | (lc.hashed ? 0x80000000u : 0u), lc.hs, lc.offset); }
		lct[l] = v;
	}
}
template <int D>
DEV h2 level_features2_lds(const __half* __restrict__ table, const uint4* lct, uint32_t level, const float* __restrict__ x) {
	const uint4 v = lct[level];
	LevelConst lc; lc.scale = __uint_as_float(v.x); lc.res = v.y & 0x7fffffffu; lc.hashed = (v.y >> 31) != 0u; lc.hs = v.z; lc.offset = v.w;
	CornersND<D> cr;
	level_corners_nd<D>(lc, x, cr);
	const uint32_t* t = (const uint32_t*)table + lc.offset;
	constexpr int NC = 1 << D;
	uint32_t vv[NC];
#pragma unroll
	for (int c = 0; c < NC; ++c) vv[c] = t[cr.idx[c]];
	h2 r = {(_Float16)0.f, (_Float16)0.f};
#pragma unroll
	for (int c = 0; c < NC; ++c) {
		const _Float16 wh = (_Float16)cr.w[c];
		const h2 w2 = {wh, wh};
		r = __builtin_elementwise_fma(w2, __builtin_bit_cast(h2, vv[c]), r);
	}
	__builtin_amdgcn_sched_barrier(0);
	return r;
}

template <int D>
__global__ void __launch_bounds__(256, 3) k_encmlp_inference(const GridMeta* __restrict__ gm, const __half* __restrict__ table, const ngp_half* __restrict__ frags,
		const float* __restrict__ in, uint32_t in_stride, uint32_t n, __half* __restrict__ out, uint32_t out_stride, uint32_t n_out) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	h8* fw = (h8*)smem;
	__shared__ uint4 s_lct[16];
	fill_level_table_nd<D>(s_lct, gm);
	load_frags_to_lds(fw, frags, (int)N_FW_FRAGS);
	__syncthreads();
	const int lane = threadIdx.x & 63, col = lane & 31, hi = lane >> 5;
	const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
	for (uint32_t tile = wave; (uint64_t)tile * 32 < n; tile += n_waves) {
		const uint32_t s_raw = tile * 32 + col;
		const float* p = in + (size_t)min(s_raw, n - 1) * in_stride;
		float x[D];
#pragma unroll
		for (int d = 0; d < D; ++d) x[d] = p[d];
		FwdState<1> st;
#pragma unroll
		for (int s = 0; s < 2; ++s) {
			h8 e;
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const h2 f = level_features2_lds<D>(table, s_lct, (uint32_t)(8 * s + 4 * (q >> 1) + 2 * hi + (q & 1)), x);
				e[2 * q] = f[0]; e[2 * q + 1] = f[1];
			}
			st.rin[0][s] = e;
		}
		fwd_rgb_l1<1>(fw, lane, st);
		fwd_rgb_l2<1>(fw, lane, st);
		f16v o[1];
		fwd_rgb_l3<1>(fw, lane, st, o);
		if (s_raw < n) {
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				const uint32_t row = (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * hi);
				if (row < n_out) out[(size_t)s_raw * out_stride + row] = __float2half(o[0][r]);
			}
		}
	}
}

// encoding only (unit-test hook): out[i][32] halfs in NATURAL feature order (level-major)
template <int F>
__global__ void __launch_bounds__(256) k_encode_only(const GridMeta* __restrict__ gm, const __half* __restrict__ table, const float* __restrict__ pos, uint32_t stride,
		uint32_t n, __half* __restrict__ out) {
	const int lane = threadIdx.x & 63, col = lane & 31, hi = lane >> 5;
	const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	const uint32_t s_raw = wave * 32 + col;
	if (wave * 32 >= n) return;
	const float* p = pos + (size_t)min(s_raw, n - 1) * stride;
	h8 e[2];
	encode_sample<F>(gm, table, p[0], p[1], p[2], hi, e);
	if (s_raw >= n) return;
#pragma unroll
	for (int s = 0; s < 2; ++s)
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const int f = 16 * s + 8 * (j >> 2) + 4 * hi + (j & 3);
			((_Float16*)out)[(size_t)s_raw * 32 + f] = e[s][j];
		}
}

// ---------------------------------------------------------------------------------------------
// T1: forward + dgrad + hash-grid scatter; stashes the encoding fragments for kernel W.
// ---------------------------------------------------------------------------------------------
DEV void atomic_add_h2(__half* addr, h2 v) {
	typedef __attribute__((address_space(1))) h2 gh2;
	__builtin_amdgcn_global_atomic_fadd_v2f16((gh2*)addr, v);
}

// SCATTER = false: every level's dL/d(enc) goes to denc_lv and the kernel issues no atomics (production: all levels through the bin lists);
// compiled separately so that the scatter code's registers do not limit the occupancy of the gather-latency-bound forward / dgrad part.
// STASH: the encodings come from the lazy K2's per-sample records (EncStashIn) instead of the hash tables.
template <int CT, int MINW, bool SCATTER, int F = 4, int NR = 2, bool STASH = false, int EX = 0>
__global__ void __launch_bounds__(256, MINW) k_train_fwd_bwd(const GridMeta* __restrict__ gm, ModelPtrs mp, const float* __restrict__ in, uint32_t in_stride, uint32_t n,
		const __half* __restrict__ dL_dy, uint32_t dy_stride, __half* __restrict__ grid_grad, uint4* __restrict__ enc_stash, uint32_t flags,
		uint2* __restrict__ denc_lv, uint32_t denc_cap, EncStashIn stash_in = EncStashIn(), float* __restrict__ dextra_out = nullptr /* EX: optional dL/d(extra dims), n x n_extra floats */) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	h8* fw = (h8*)smem;
	h8* bw = fw + (n_fw(NR) + (EX ? N_FW_EXTRA : 0)) * 64;
	load_frags_to_lds(fw, mp.fw_frags, n_fw(NR) + (EX ? N_FW_EXTRA : 0));
	load_frags_to_lds(bw, mp.bw_frags, n_bw(NR) + (EX ? N_BW_EXTRA : 0));
	__syncthreads();
	const int lane = threadIdx.x & 63, col = lane & 31, hi = lane >> 5;
	const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
	const __half* table = (const __half*)mp.grid;
	constexpr uint32_t TS = 32 * CT;
	uint32_t n_valid_rows = 0;
	if constexpr (STASH) n_valid_rows = min(*stash_in.n_valid_ptr, n);
#pragma unroll 1
	for (uint32_t tile = wave; (uint64_t)tile * TS < n; tile += n_waves) {
		FwdState<CT> st;
		uint32_t sidx[CT];
		float px[CT], py[CT], pz[CT];
#pragma unroll
		for (int c = 0; c < CT; ++c) {
			const uint32_t s_raw = tile * TS + c * 32 + col;
			sidx[c] = s_raw;
			const float* p = in + (size_t)min(s_raw, n - 1) * in_stride;
			px[c] = p[0]; py[c] = p[1]; pz[c] = p[2];
			if constexpr (STASH) {
				uint32_t row = min(s_raw, n - 1);
				if (row >= n_valid_rows) row = n_valid_rows ? row % n_valid_rows : 0u; // K4's wrap-around padding: row e is a copy of row e % n_valid
				const uint32_t src = n_valid_rows ? stash_in.src_index[row] : 0u;
				const uint4* e = stash_in.enc + (size_t)src * 4 + (uint32_t)hi * 2;
				st.enc[c][0] = __builtin_bit_cast(h8, e[0]); st.enc[c][1] = __builtin_bit_cast(h8, e[1]);
				__builtin_amdgcn_sched_barrier(0); // without it the scheduler hoists the layers' LDS fragment loads above these two loads: 168 registers + 328 B of scratch instead of 130 + 0
			} else
			encode_sample<F>(gm, table, px[c], py[c], pz[c], hi, st.enc[c]);
			st.rin[c][1] = sh4_frag(p[4], p[5], p[6], hi);
			if constexpr (EX != 0) st.rin[c][2] = extra_frag(p + 7, mp.n_extra, hi);
			// stash for kernel W: [32-sample tile][s][lane] 16-byte chunks (lane-linear, coalesced)
			enc_stash[(((size_t)tile * CT + c) * 2 + 0) * 64 + lane] = __builtin_bit_cast(uint4, st.enc[c][0]);
			enc_stash[(((size_t)tile * CT + c) * 2 + 1) * 64 + lane] = __builtin_bit_cast(uint4, st.enc[c][1]);
		}
		fwd_density_l1<CT>(fw, lane, st);
		fwd_density_l2<CT>(fw, lane, st);
		fwd_rgb_hidden<CT, NR, EX>(fw, lane, st);

		// ---- backward (dgrad chain) ----
		h8 dy0[CT];          // dL/d(rgb output) fragment, k-step 0
		_Float16 dsig[CT];   // dL/d(sigma logit)
#pragma unroll
		for (int c = 0; c < CT; ++c) {
			dy0[c] = zero8(); dsig[c] = (_Float16)0.f;
			if (hi == 0 && sidx[c] < n) {
				const uint2 raw = *(const uint2*)(dL_dy + (size_t)sidx[c] * dy_stride);
				const h4 g = __builtin_bit_cast(h4, raw);
				dy0[c][0] = g[0]; dy0[c][1] = g[1]; dy0[c][2] = g[2];
				dsig[c] = g[3];
			}
		}
		h8 dh[CT][4];
		// rgb L3^T : d_h = W3^T * d_out   (kin tiles mt = 0,1; one k-step), masked by the last hidden layer's ReLU state
#pragma unroll
		for (int mt = 0; mt < 2; ++mt) {
			h8 a = lds_frag(bw, bw_r3(NR) + mt, lane);
#pragma unroll
			for (int c = 0; c < CT; ++c) {
				f16v d = mfma(a, dy0[c], zero16());
				const uint32_t mask = NR == 1 ? st.m1r[c] : st.m2r[NR >= 2 ? NR - 2 : 0][c];
				dh[c][2 * mt + 0] = to_frag_masked(d, 0, mask >> (16 * mt));
				dh[c][2 * mt + 1] = to_frag_masked(d, 1, mask >> (16 * mt));
			}
		}
		// the 64 x 64 layers, last to first: d_h(k) = W2(k)^T * d_h(k+1), masked by the ReLU state of the layer below
#pragma unroll
		for (int k = NR - 2; k >= 0; --k) {
			f16v acc[2][CT];
#pragma unroll
			for (int mt = 0; mt < 2; ++mt)
#pragma unroll
				for (int c = 0; c < CT; ++c) acc[mt][c] = zero16();
#pragma unroll
			for (int mt = 0; mt < 2; ++mt)
#pragma unroll
				for (int s = 0; s < 4; ++s) {
					h8 a = lds_frag(bw, BW_R2 + 8 * k + mt * 4 + s, lane);
#pragma unroll
					for (int c = 0; c < CT; ++c) acc[mt][c] = mfma(a, dh[c][s], acc[mt][c]);
				}
#pragma unroll
			for (int mt = 0; mt < 2; ++mt)
#pragma unroll
				for (int c = 0; c < CT; ++c) {
					const uint32_t mask = k == 0 ? st.m1r[c] : st.m2r[k >= 1 ? k - 1 : 0][c];
					dh[c][2 * mt + 0] = to_frag_masked(acc[mt][c], 0, mask >> (16 * mt));
					dh[c][2 * mt + 1] = to_frag_masked(acc[mt][c], 1, mask >> (16 * mt));
				}
		}
		// rgb L1^T : d_rin = W1^T * d_h1 ; only rows 0..15 (density-net output) are consumed
		if constexpr (EX != 0) if (dextra_out) {
			// ... and rows 32..47 = dL/d(extra dims): the colour network's input gradient is a half matrix in the reference, the Identity encoding's backward hands it on
			// as float (nerf_network.h:238-252 -> coords_gradient, testbed_nerf.cu:1326)
			f16v acc[CT];
#pragma unroll
			for (int c = 0; c < CT; ++c) acc[c] = zero16();
#pragma unroll
			for (int s = 0; s < 4; ++s) {
				h8 a = lds_frag(bw, n_bw(NR) + s, lane);
#pragma unroll
				for (int c = 0; c < CT; ++c) acc[c] = mfma(a, dh[c][s], acc[c]);
			}
#pragma unroll
			for (int c = 0; c < CT; ++c) {
				if (sidx[c] >= n) continue;
#pragma unroll
				for (int r = 0; r < 8; ++r) { // tile rows (r & 3) + 8 (r >> 2) + 4 hi < 16
					const uint32_t k = (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * hi);
					if (k < mp.n_extra) dextra_out[(size_t)sidx[c] * mp.n_extra + k] = (float)(_Float16)acc[c][r];
				}
			}
		}
		h8 ddens[CT];
		{
			f16v acc[CT];
#pragma unroll
			for (int c = 0; c < CT; ++c) acc[c] = zero16();
#pragma unroll
			for (int s = 0; s < 4; ++s) {
				h8 a = lds_frag(bw, BW_R1 + s, lane);
#pragma unroll
				for (int c = 0; c < CT; ++c) acc[c] = mfma(a, dh[c][s], acc[c]);
			}
#pragma unroll
			for (int c = 0; c < CT; ++c) {
				uint32_t dummy = 0;
				ddens[c] = to_frag<false>(acc[c], 0, dummy);
				// add_density_gradient (nerf_network.h:235): half add into density-net output 0
				if (hi == 0) ddens[c][0] = (_Float16)((float)ddens[c][0] + (float)dsig[c]);
			}
		}
		// density L2^T : d_h1d = W2d^T * d_densout
#pragma unroll
		for (int mt = 0; mt < 2; ++mt) {
			h8 a = lds_frag(bw, BW_D2 + mt, lane);
#pragma unroll
			for (int c = 0; c < CT; ++c) {
				f16v d = mfma(a, ddens[c], zero16());
				dh[c][2 * mt + 0] = to_frag_masked(d, 0, st.m1d[c] >> (16 * mt));
				dh[c][2 * mt + 1] = to_frag_masked(d, 1, st.m1d[c] >> (16 * mt));
			}
		}
		// density L1^T : d_enc = W1d^T * d_h1d  (32 features = one row tile)
		f16v denc[CT];
#pragma unroll
		for (int c = 0; c < CT; ++c) denc[c] = zero16();
#pragma unroll
		for (int s = 0; s < 4; ++s) {
			h8 a = lds_frag(bw, BW_D1 + s, lane);
#pragma unroll
			for (int c = 0; c < CT; ++c) denc[c] = mfma(a, dh[c][s], denc[c]);
		}
		// ---- hash-grid scatter: lane (n, hi) owns levels 2*rr + hi (rr = r>>2), features r&3 ----
		// Consecutive lanes of a 32-lane half are consecutive samples of (mostly) one ray; on the coarse
		// levels whole runs of them fall into the same grid cell and would hit the same 8 table entries.
		// Runs are found with one ballot per level and summed with a segmented shuffle reduction in
		// packed half (the same precision class as the table's half atomics), so only the run head
		// issues global_atomic_pk_add_f16.
		if (flags & DBG_T1_NO_SCATTER) continue;
		if (F == 2) {
			// F = 2, L = 16: register pair (2m, 2m+1) of the dL/d(enc) tile = features 0,1 of level (m & 1) + 4 (m >> 1) + 2 hi.  Levels that go through the
			// record lists (all of them in production) leave their 4 bytes per sample level-major in denc_lv; the others (no lists: table sizes the
			// binning does not cover, DBG_T1_NO_BINNING) are scattered here with one packed-half atomic per corner like the reference's atomicAdd(__half2).
			uint32_t* __restrict__ denc32 = (uint32_t*)denc_lv;
#pragma unroll
			for (int c = 0; c < CT; ++c) {
				const bool sv = sidx[c] < n;
#pragma unroll
				for (int m = 0; m < 8; ++m) {
					const int l0 = (m & 1) + 4 * (m >> 1);
					const h2 g = {(_Float16)denc[c][2 * m], (_Float16)denc[c][2 * m + 1]}; // dL/d(enc) is a half matrix in the reference
					if (!SCATTER) { if (sv) denc32[(size_t)(l0 + 2 * hi) * denc_cap + sidx[c]] = __builtin_bit_cast(uint32_t, g); continue; }
					const LevelConst lc = level_const2(gm, l0, l0 + 2, hi);
					const bool binned = denc_lv != nullptr && (lc.hashed || (flags & T1_DENSE_EXTERNAL));
					if (binned) { if (sv) denc32[(size_t)(l0 + 2 * hi) * denc_cap + sidx[c]] = __builtin_bit_cast(uint32_t, g); }
					if (__ballot(!binned) == 0ull) continue;
					if (sv && !binned) {
						Corners cr;
						level_corners(lc, px[c], py[c], pz[c], cr);
						const float g0 = (float)g[0], g1 = (float)g[1];
						__half* gt = grid_grad + (size_t)lc.offset * 2;
#pragma unroll
						for (int k = 0; k < 8; ++k) {
							const h2 v = {(_Float16)(g0 * cr.w[k]), (_Float16)(g1 * cr.w[k])};
							atomic_add_h2(gt + (size_t)cr.idx[k] * 2, v);
						}
					}
					__builtin_amdgcn_sched_barrier(0);
				}
			}
			continue;
		}
		if (!SCATTER) {
#pragma unroll
			for (int c = 0; c < CT; ++c) {
				if (sidx[c] >= n) continue;
#pragma unroll
				for (int rr = 0; rr < 4; ++rr) {
					const h4 g = {(_Float16)denc[c][4 * rr + 0], (_Float16)denc[c][4 * rr + 1], (_Float16)denc[c][4 * rr + 2], (_Float16)denc[c][4 * rr + 3]};
					denc_lv[(size_t)(2 * rr + hi) * denc_cap + sidx[c]] = __builtin_bit_cast(uint2, g);
				}
			}
			continue;
		}
#pragma unroll
		for (int c = 0; c < CT; ++c) {
			const bool sv = sidx[c] < n;
#pragma unroll
			for (int rr = 0; rr < 4; ++rr) {
				if ((flags & DBG_T1_NO_COARSE_LEVELS) && rr < 2) continue;
				if ((flags & DBG_T1_NO_FINE_LEVELS) && rr >= 2) continue;
				const LevelConst lc = level_const(gm, 2 * rr, hi);
				// Hashed levels (binned mode): the memory side retires only ~14 G atomic requests/s, and hashed corners neither
				// merge nor coalesce -- they were 85 % of T1's atomic requests.  Their dL/d(enc) goes to memory level-major
				// (8 bytes per sample and level, coalesced) and k_grad_bin / k_grad_accumulate turn it into table gradients
				// through LDS accumulators, without global atomics.  Dense levels: merged + coalesced atomics, either below (production) or (T1_DENSE_EXTERNAL,
				// ablation) in k_grad_dense, which runs beside k_grad_bin / k_grad_accumulate / W on its own stream.
				const bool binned = denc_lv != nullptr && (lc.hashed || (flags & T1_DENSE_EXTERNAL));
				if (binned && sv) {
					const h4 g = {(_Float16)denc[c][4 * rr + 0], (_Float16)denc[c][4 * rr + 1], (_Float16)denc[c][4 * rr + 2], (_Float16)denc[c][4 * rr + 3]};
					denc_lv[(size_t)(2 * rr + hi) * denc_cap + sidx[c]] = __builtin_bit_cast(uint2, g);
				}
				if (__ballot(!binned) == 0ull) continue; // both levels of this register group are binned
				const bool svl = sv && !binned;
				Corners cr;
				level_corners(lc, px[c], py[c], pz[c], cr);
				// dL/d(enc) is rounded to half first (it is a half matrix in the reference)
				const float g0 = svl ? (float)(_Float16)denc[c][4 * rr + 0] : 0.f, g1 = svl ? (float)(_Float16)denc[c][4 * rr + 1] : 0.f;
				const float g2 = svl ? (float)(_Float16)denc[c][4 * rr + 2] : 0.f, g3 = svl ? (float)(_Float16)denc[c][4 * rr + 3] : 0.f;
				h2 v0[8], v1[8];
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					const float w = cr.w[k];
					h2 a0 = {(_Float16)(g0 * w), (_Float16)(g1 * w)}, a1 = {(_Float16)(g2 * w), (_Float16)(g3 * w)};
					v0[k] = a0; v1[k] = a1;
				}
				// run detection: same grid cell as the previous lane of this half => all 8 corner indices equal
				const uint32_t key = cr.cell_xy, key7 = cr.cell_z;
				const uint32_t pkey = (uint32_t)__shfl_up((int)key, 1, 32), pkey7 = (uint32_t)__shfl_up((int)key7, 1, 32);
				const bool head = (col == 0) || key != pkey || key7 != pkey7;
				const uint64_t hm = __ballot(head);
				bool issue = true;
				if (!(flags & DBG_T1_NO_MERGE) && __popcll(hm) <= 48) { // wave-uniform: enough followers to pay for the reduction
					const uint32_t hmask = hi ? (uint32_t)(hm >> 32) : (uint32_t)hm;
					const uint32_t rest = col == 31 ? 0u : (hmask >> (col + 1));
					const uint32_t run_right = rest ? (uint32_t)(__ffs((int)rest) - 1) : (uint32_t)(31 - col);
#pragma unroll
					for (int d = 1; d < 32; d <<= 1) {
						const bool take = run_right >= (uint32_t)d;
#pragma unroll
						for (int k = 0; k < 8; ++k) {
							const h2 t0 = __builtin_bit_cast(h2, __shfl_down(__builtin_bit_cast(int, v0[k]), d, 32));
							const h2 t1 = __builtin_bit_cast(h2, __shfl_down(__builtin_bit_cast(int, v1[k]), d, 32));
							if (take) { v0[k] += t0; v1[k] += t1; }
						}
					}
					issue = head;
				}
				__half* gt = grid_grad + (size_t)lc.offset * 4;
				if (!(flags & (DBG_T1_NO_PAIR_HALVES | DBG_T1_NO_QUADS))) {
					// A lane QUAD issues, per instruction, the 16 bytes of one sample's x-adjacent corner pair: {corner 2p: half 0,
					// half 1; corner 2p+1: half 0, half 1}.  On dense levels (and for even x on hashed ones, prime_x = 1) the two
					// entries are adjacent, so the four lane-atomics fall into one cache line and leave the CU as ONE request.
					const bool own = issue && svl;
					const int r4 = lane & 3;
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						const bool go = __shfl((int)own, q, 4) != 0;
#pragma unroll
						for (int p = 0; p < 4; ++p) {
							const uint32_t i0 = (uint32_t)__shfl((int)cr.idx[2 * p], q, 4), i1 = (uint32_t)__shfl((int)cr.idx[2 * p + 1], q, 4);
							const int a0 = __shfl(__builtin_bit_cast(int, v0[2 * p]), q, 4), a1 = __shfl(__builtin_bit_cast(int, v1[2 * p]), q, 4);
							const int b0 = __shfl(__builtin_bit_cast(int, v0[2 * p + 1]), q, 4), b1 = __shfl(__builtin_bit_cast(int, v1[2 * p + 1]), q, 4);
							const uint32_t ix = (r4 & 2) ? i1 : i0;
							const int val = r4 == 0 ? a0 : r4 == 1 ? a1 : r4 == 2 ? b0 : b1;
							if (go) atomic_add_h2(gt + (size_t)ix * 4 + (r4 & 1) * 2, __builtin_bit_cast(h2, val));
						}
					}
				} else if (!(flags & DBG_T1_NO_PAIR_HALVES)) {
					// both 4-byte halves of an 8-byte entry go out in the SAME instruction from a lane pair (L, L^1): the vector
					// memory pipeline coalesces same-line atomic lanes of one instruction into one memory-side request, like it
					// does for stores (measured: 1.06 -> 0.59 ms per step, profiles/r01_microbench_ablation2.log)
					const bool own = issue && svl;
					const bool odd = (lane & 1) != 0;
					const bool nb_own = __shfl_xor((int)own, 1, 64) != 0;
#pragma unroll
					for (int k = 0; k < 8; ++k) {
						const uint32_t idx_nb = (uint32_t)__shfl_xor((int)cr.idx[k], 1, 64);
						const h2 v0_nb = __builtin_bit_cast(h2, __shfl_xor(__builtin_bit_cast(int, v0[k]), 1, 64));
						const h2 v1_nb = __builtin_bit_cast(h2, __shfl_xor(__builtin_bit_cast(int, v1[k]), 1, 64));
						// step A: the even lane's entry; step B: the odd lane's entry
						{
							const bool go = odd ? nb_own : own;
							if (go) atomic_add_h2(gt + (size_t)(odd ? idx_nb : cr.idx[k]) * 4 + (odd ? 2 : 0), odd ? v1_nb : v0[k]);
						}
						{
							const bool go = odd ? own : nb_own;
							if (go) atomic_add_h2(gt + (size_t)(odd ? cr.idx[k] : idx_nb) * 4 + (odd ? 2 : 0), odd ? v1[k] : v0_nb);
						}
					}
				} else if (issue && svl) {
#pragma unroll
					for (int k = 0; k < 8; ++k) {
						__half* dst = gt + (size_t)cr.idx[k] * 4;
						atomic_add_h2(dst, v0[k]);
						atomic_add_h2(dst + 2, v1[k]);
					}
				}
				__builtin_amdgcn_sched_barrier(0);
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------
// Binned hash-grid scatter for the HASHED levels (no global atomics).
//   k_grad_bin:        one block = 512 samples of one level.  The 4096 corner records {value: 4 halfs, index} are counting-
//                      sorted in LDS by table chunk (4096 entries = 32 KiB of table per chunk), one returning atomic per
//                      (block, chunk) reserves the slots in the chunk's list, and the sorted records leave the CU as
//                      contiguous runs (coalesced 8-byte value / 2-byte index streams).
//   k_grad_accumulate: one block = one chunk x one feature pair.  Records are summed EXACTLY into 64-bit fixed-point LDS
//                      accumulators (integer ds_add_u64) and the chunk's gradients are written once (the reference sums in
//                      half with atomicAdd(__half2): one rounding per contribution, order dependent).
// The hash spreads every sample's corners uniformly over the chunks, so the lists are balanced for any scene; a list that
// still overflows its capacity (2x the mean) falls back to global atomics on the spot.
// ---------------------------------------------------------------------------------------------

// record value of one corner: the F halfs of the entry's gradient contribution (F = 4: 8 bytes, F = 2: 4 bytes)
// workgroup barrier that orders LDS traffic only: outstanding global loads / atomics stay in flight (the caller must not rely on them)
#define NGP_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
template <int F> struct BinVal;
template <> struct BinVal<4> { typedef uint2 type; };
template <> struct BinVal<2> { typedef uint32_t type; };
template <int F> DEV typename BinVal<F>::type pack_halfs(const float* v);
template <> __device__ __forceinline__ uint2 pack_halfs<4>(const float* v) { const h4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]}; return __builtin_bit_cast(uint2, hv); }
template <> __device__ __forceinline__ uint32_t pack_halfs<2>(const float* v) { const h2 hv = {(_Float16)v[0], (_Float16)v[1]}; return __builtin_bit_cast(uint32_t, hv); }

// THREADS: 256 (two samples per thread at 512 samples per block) or 512 (one): the block's LDS image is the same, so 512 threads double the wavefronts per CU
// (3 blocks of 50 KiB: 12 -> 24) -- the kernel is latency bound (position / gradient loads, the cursor atomics, four barriers), not LDS or issue bound.
// D = 2: the image primitive's 2-D grid (4 corners per sample; its samples are i.i.d. pixels, so runs are not merged)
template <uint32_t CL2, uint32_t GRAD_BIN_SAMPLES /* samples of one level per block: 256 | 512 */, int F = 4, uint32_t THREADS = 256, int D = 3>
__global__ void __launch_bounds__(THREADS) k_grad_bin(GradBinArgs a) {
	typedef typename BinVal<F>::type val_t;
	static_assert(D == 3 || (D == 2 && F == 2), "k_grad_bin: 3-D grids, or the 2-D F = 2 grid of the image primitive");
	constexpr int NC = 1 << D; // corners per sample
	constexpr uint32_t NCH = (1u << GRAD_BIN_MAX_TABLE_LOG2) >> CL2; // most chunks a level can have (128 / 256)
	__shared__ uint32_t s_cnt[NCH], s_start[NCH], s_gbase[NCH];
	__shared__ uint32_t s_wsum[THREADS / 64];
	__shared__ val_t s_val[GRAD_BIN_SAMPLES * NC];
	__shared__ uint32_t s_key[GRAD_BIN_SAMPLES * NC];
	const uint32_t tid = threadIdx.x, ly = blockIdx.y, level = a.levels[ly];
	LevelConst lc = level_const_uniform(a.gm, level);
	if (D == 2) lc.hashed = (uint64_t)lc.res * lc.res > (uint64_t)lc.hs;
	// Chunk of a table entry.  Hashed level: 2^CL2 consecutive entries (the hash spreads every sample's corners over all chunks).  Dense level
	// (a.dense_too): entries are INTERLEAVED over all NCH chunks (chunk = index mod NCH, local = index / NCH), so that the spatially clustered
	// samples of a scene still fill the lists evenly; a dense level has at most 2^19 entries, i.e. local < 2^CL2.
	constexpr uint32_t NCH_LOG2 = GRAD_BIN_MAX_TABLE_LOG2 - CL2;
	const bool dense = !lc.hashed;
	auto chunk_of = [&](uint32_t idx) { return dense ? idx & (NCH - 1u) : idx >> CL2; };
	auto local_of = [&](uint32_t idx) { return dense ? idx >> NCH_LOG2 : idx & ((1u << CL2) - 1u); };
	const uint32_t n_chunks = dense ? NCH : lc.hs >> CL2;
	for (uint32_t c = tid; c < NCH; c += THREADS) s_cnt[c] = 0;
	__syncthreads();
	constexpr int SPT = GRAD_BIN_SAMPLES / THREADS; // samples per thread
	static_assert(GRAD_BIN_SAMPLES % THREADS == 0 && NCH <= THREADS, "k_grad_bin: block shape");
	uint32_t idx[SPT][NC], rank[SPT][NC];
	val_t val[SPT][NC];
	bool valid[SPT];
	const uint32_t lane = tid & 63u;
#pragma unroll
	for (int u = 0; u < SPT; ++u) {
		const uint32_t s = blockIdx.x * GRAD_BIN_SAMPLES + u * THREADS + tid;
		valid[u] = s < a.n;
		float g[F];
#pragma unroll
		for (int f = 0; f < F; ++f) g[f] = 0.f;
		struct { uint32_t idx[NC]; float w[NC]; uint32_t cell_xy, cell_z; } cr;
		{
			const uint32_t sc = valid[u] ? s : a.n - 1; // every lane takes part in the shuffles below
			const float* p = a.in + (size_t)sc * a.in_stride;
			if constexpr (D == 3) {
				Corners c3;
				const f3u_t p3 = *(const f3u_t*)p;
				level_corners(lc, p3[0], p3[1], p3[2], c3);
#pragma unroll
				for (int k = 0; k < NC; ++k) { cr.idx[k] = c3.idx[k]; cr.w[k] = c3.w[k]; }
				cr.cell_xy = c3.cell_xy; cr.cell_z = c3.cell_z;
			} else {
				CornersND<D> c2;
				const float x2[2] = {p[0], p[1]};
				level_corners_nd<D>(a.gm, level, x2, c2);
#pragma unroll
				for (int k = 0; k < NC; ++k) { cr.idx[k] = c2.idx[k]; cr.w[k] = c2.w[k]; }
				cr.cell_xy = s; cr.cell_z = 0u; // (never merged)
			}
			if (valid[u]) {
				const val_t raw = ((const val_t*)a.denc_lv)[(size_t)level * a.denc_cap + sc];
				const _Float16* gh = (const _Float16*)&raw;
#pragma unroll
				for (int f = 0; f < F; ++f) g[f] = (float)gh[f];
			}
		}
		// Consecutive lanes are consecutive samples of (mostly) one ray; on the coarser hashed levels runs of them share a grid cell, i.e. all
		// eight table entries.  Such runs are summed here (fp32, segmented shuffle reduction like T1's, stopped at the longest run of the
		// wavefront) and only the run head emits records: 20-25 % fewer records of the hashed levels and ~60 % fewer of the dense ones to sort,
		// write, read and accumulate (without it the dense levels' lists overflow).  Wave-uniform decision: worth it when >= 1/4 of the lanes are followers.
		const uint32_t key_xy = cr.cell_xy, key_z = cr.cell_z;
		const uint32_t pxy = (uint32_t)__shfl_up((int)key_xy, 1, 64), pz = (uint32_t)__shfl_up((int)key_z, 1, 64);
		const bool pvalid = __shfl_up((int)valid[u], 1, 64) != 0;
		// (runs are cut at the 16-lane DPP rows -- lanes 0, 16, 32, 48 are heads -- so that the reduction below stays inside a row: round 6)
		const bool head = (lane & 15u) == 0u || key_xy != pxy || key_z != pz || !valid[u] || !pvalid;
		const uint64_t hm = __ballot(head);
		const bool merge = D == 3 && (a.merge_runs || (dense && !a.no_dense_merge)) && __popcll(hm) <= 48; // dense (coarse) levels: most lanes are followers
		bool emit = valid[u];
		if (merge) {
			const uint64_t rest = lane == 63 ? 0ull : (hm >> (lane + 1));
			const uint32_t run_right = rest ? (uint32_t)(__ffsll((long long)rest) - 1) : (63u - lane); // followers to my right that belong to my run
			float v[NC][F];
#pragma unroll
			for (int k = 0; k < NC; ++k) { const float w = cr.w[k];
#pragma unroll
				for (int f = 0; f < F; ++f) v[k][f] = g[f] * w; }
			// segmented suffix sums by doubling, through DPP row shifts (v_mov_b32 row_shl:d: lane i reads lane i + d of its 16-lane row, 0 past the row's end) instead of
			// __shfl_down = ds_bpermute_b32: 32 values x up to 4 steps were 128 LDS-pipe instructions per thread of a kernel that is VALU / LDS-issue bound (round 6).  A run never
			// crosses a row (forced heads above), so a lane that takes never reads past its row.
			auto step = [&](auto dconst) {
				constexpr int d = decltype(dconst)::value;
				const bool take = run_right >= (uint32_t)d;
				if (__ballot(take) == 0ull) return false; // no run reaches this far (wave-uniform): runs are a ray's samples in one cell
#pragma unroll
				for (int k = 0; k < NC; ++k)
#pragma unroll
					for (int f = 0; f < F; ++f) {
						const float t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[k][f]), 0x100 + d, 0xf, 0xf, true));
						v[k][f] += take ? t : 0.f;
					}
				return true;
			};
			if (step(std::integral_constant<int, 1>{}) && step(std::integral_constant<int, 2>{}) && step(std::integral_constant<int, 4>{})) (void)step(std::integral_constant<int, 8>{});
			emit = valid[u] && head;
#pragma unroll
			for (int k = 0; k < NC; ++k) val[u][k] = pack_halfs<F>(v[k]);
		} else {
#pragma unroll
			for (int k = 0; k < NC; ++k) {
				const float w = cr.w[k];
				float v[F];
#pragma unroll
				for (int f = 0; f < F; ++f) v[f] = g[f] * w;
				val[u][k] = pack_halfs<F>(v);
			}
		}
		valid[u] = emit;
		if (!emit) continue;
#pragma unroll
		for (int k = 0; k < NC; ++k) {
			idx[u][k] = cr.idx[k];
			rank[u][k] = atomicAdd(&s_cnt[chunk_of(cr.idx[k])], 1u);
		}
	}
	__syncthreads();
	// exclusive prefix of the chunk counts (<= 256 chunks: one per thread) + slot reservation in the global lists
	uint32_t gbase_reg = 0u;
	{
		const uint32_t cnt = tid < n_chunks ? s_cnt[tid < NCH ? tid : 0u] : 0u;
		// The reservation only needs the count: the returning atomic is issued here and its result is first used behind the LDS scatter below, so its
		// round trip to the memory side overlaps the prefix sum and the scatter.  The two barriers in between wait for LDS only (s_waitcnt lgkmcnt(0)):
		// __syncthreads() would wait for the atomic as well.
		if (tid < NCH && cnt) gbase_reg = atomicAdd(&a.cursors[ly * a.max_chunks + tid], cnt);
		uint32_t x = cnt;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)x, d, 64); if ((tid & 63u) >= (uint32_t)d) x += y; }
		if ((tid & 63u) == 63u) s_wsum[tid >> 6] = x;
		NGP_LDS_BARRIER();
		uint32_t woff = 0;
		for (uint32_t w = 0; w < (tid >> 6); ++w) woff += s_wsum[w];
		if (tid < NCH) s_start[tid] = woff + x - cnt;
	}
	NGP_LDS_BARRIER();
#pragma unroll
	for (int u = 0; u < SPT; ++u) {
		if (!valid[u]) continue;
#pragma unroll
		for (int k = 0; k < NC; ++k) {
			const uint32_t c = chunk_of(idx[u][k]);
			const uint32_t pos = s_start[c] + rank[u][k];
			s_val[pos] = val[u][k];
			s_key[pos] = (c << 16) | local_of(idx[u][k]);
		}
	}
	if (tid < NCH) s_gbase[tid] = gbase_reg;
	__syncthreads();
	uint32_t total = 0;
#pragma unroll
	for (uint32_t w = 0; w < THREADS / 64; ++w) total += s_wsum[w];
	for (uint32_t i = tid; i < total; i += THREADS) {
		const uint32_t key = s_key[i], c = key >> 16, local = key & 0xffffu;
		const uint32_t d = s_gbase[c] + (i - s_start[c]);
		const val_t v = s_val[i];
		if (d < a.cap) {
			const size_t o = ((size_t)ly * a.max_chunks + c) * a.cap + d;
			((val_t*)a.vals)[o] = v;
			a.idxs[o] = (uint16_t)local;
		} else { // list full: straight to the table (k_grad_accumulate adds its sums on top)
			__half* dst = (__half*)a.grid_grad_ + ((size_t)lc.offset + (dense ? ((size_t)local << NCH_LOG2) + c : ((size_t)c << CL2) + local)) * F;
			if constexpr (F == 4) { atomic_add_h2(dst, __builtin_bit_cast(h2, v.x)); atomic_add_h2(dst + 2, __builtin_bit_cast(h2, v.y)); }
			else atomic_add_h2(dst, __builtin_bit_cast(h2, v));
		}
	}
}

// [tcnn optimizers/adam.h adam_step + ema.h ema_step_half_precision], one sweep over all parameters, 4 parameters
// (= one F=4 hash-table entry) per thread: 8-byte gradient / half-parameter accesses, 16-byte fp32 state accesses; the
// Adam state of an entry is only touched when one of its gradients is non-zero (sparse update of the reference).
// (no fp contraction in the two update rules: they are instantiated in k_optimizer AND in k_grad_accumulate's fused epilogue, and both must round identically --
// tests/test_gpu_train.py::test_fused_optimizer_epilogue_is_the_separate_sweep; it is also the arithmetic of the un-contracted oracle)
DEV float adam_update(const AdamArgs& a, bool matrix, float gradient, float weight_fp, float& m, float& v, uint16_t& step) {
#pragma clang fp contract(off)
	if (matrix) gradient += a.l2_reg * weight_fp;
	const float gradient_sq = gradient * gradient;
	const float first = m = a.beta1 * m + (1 - a.beta1) * gradient;
	const float second = v = a.beta2 * v + (1 - a.beta2) * gradient_sq;
	// per-parameter step counter, 16 bits SATURATING: it only feeds the two debias factors, and 1 - beta^t is exactly 1.0f in fp32 long before
	// t = 65,535 (beta2 = 0.99: t > 1,700; beta2 = 0.999: t > 17,000) -- the result is the one of a 32-bit counter, at half the bytes per updated entry
	const uint32_t current_step = step == 0xFFFFu ? 0xFFFFu : (uint32_t)(++step);
	float lr = a.lr;
	// beta^t as exp(t ln beta): the per-parameter step counters make powf the dominant cost of the sweep otherwise.  (Round 5: evaluating this factor once per distinct
	// step count of an entry's four parameters -- they nearly always share it -- removes 15 % of k_grad_accumulate's VALU instructions and not a microsecond of its time:
	// the kernel waits for its LDS atomics and its records, profiles/r05_ab_rng_tables_adam_cache.txt.)
	lr *= sqrtf(1 - __expf((float)current_step * a.log_beta2)) / (1 - __expf((float)current_step * a.log_beta1));
	const float effective_lr = fminf(fmaxf(lr / (sqrtf(second) + a.eps), 0.0f), 3.402823466e+38f);
	float nw = weight_fp - effective_lr * first;
	asm volatile("" : "+v"(nw)); // the fp32 value exists before the caller rounds it to half (no v_fma_mixlo_f16 from the unrounded expression: the master weight and its half copy must agree)
	return nw;
}
// [tcnn optimizers/ema.h] debiased exponential moving average, the two kernels of EmaOptimizer::step:
//   ema_step_half_precision (the default, "full_precision": false): the state IS the network-precision EMA buffer (= the inference parameters): ema_old = (float)weights_ema[i],
//                            w = (float)weights[i] (the half copy), weights_ema[i] = (T)filtered;
//   ema_step_full_precision ("full_precision": true): ema_old = tmp[i] (an fp32 buffer), w = weights_full_precision[i] (the fp32 master), tmp[i] = filtered, weights_ema[i] = (T)filtered.
// Rounds 1-5 ran a hybrid (fp32 state, half weights) that is neither; AdamArgs::ema_full_precision selects between the two since round 6.
DEV float ema_update(const AdamArgs& a, float ema_old, float w) {
#pragma clang fp contract(off)
	float r = (ema_old * a.ema_decay * a.ema_debias_old + w * (1 - a.ema_decay)) * a.ema_debias_new;
	asm volatile("" : "+v"(r)); // as in adam_update: the inference half weight is the rounding of THIS fp32 value in every instantiation
	return r;
}
// half -> integer multiple of 2^-24 (every finite half is one: subnormal step 2^-24, largest 65504 = 2047 << 29 units)
DEV long long half_bits_to_fixed(uint32_t hbits) {
	const uint32_t e = (hbits >> 10) & 31u, m = hbits & 1023u;
	const long long mag = e ? (long long)(1024u + m) << (e - 1u) : (long long)m; // e == 31 (inf/nan) -> > 65504: converts back to inf
	return (hbits & 0x8000u) ? -mag : mag;
}
// 64-bit FIXED-POINT accumulators (units of 2^-24): sums of halfs are exact, so the result does not depend on the order in
// which the records arrive -- the hashed levels' gradients are bitwise reproducible -- and integer LDS atomics run at full
// rate where ds_add_f32 / ds_pk_add_f16 do not (measured per step: 0.28 / 0.15 ms vs 0.04 ms for this kernel,
// profiles/r01_microbench_ablation7_binning.log).
//   SPLIT = false: one block = one chunk, all four features of an entry (2^CL2 x 4 x 8 B of LDS: 128 KiB at CL2 = 12, 64 KiB at 11):
//                  every record byte is fetched once;
//   SPLIT = true:  one block = one chunk x one feature pair (round-1 layout: half the LDS, but both blocks fetch every record).
// The block also empties its list for the next step (no separate reset launch).
// THREADS: 1024 (NeRF: lists of ~13 k records, 4096-entry epilogues) or 256 (the image primitive: 2048 blocks whose lists hold ~2 k records and whose chunks own a few hundred entries --
// sixteen wavefronts per block were mostly launch and barrier cost there; round 6)
template <uint32_t CL2, bool SPLIT, int F = 4, bool ADAM = false, uint32_t THREADS = 1024>
__global__ void __launch_bounds__(THREADS) k_grad_accumulate(GradBinArgs a) {
	static_assert(F == 4 || !SPLIT, "the split layout exists for F = 4 only");
	static_assert(!ADAM || (F == 4 && !SPLIT), "the fused optimizer epilogue exists for the production layout (F = 4, one block per chunk)");
	constexpr uint32_t E = 1u << CL2, NF = SPLIT ? 2u : (uint32_t)F;
	__shared__ unsigned long long acc[E * NF];
	const uint32_t tid = threadIdx.x, c = blockIdx.x, ly = blockIdx.y + a.acc_ly_begin, fp = SPLIT ? blockIdx.z : 0u, level = a.levels[ly];
	const uint32_t hs = a.gm->hashmap_size[level], offset = a.gm->offset[level];
	constexpr uint32_t NCH_LOG2 = GRAD_BIN_MAX_TABLE_LOG2 - CL2;
	const uint64_t res = a.gm->resolution[level];
	const bool dense = (a.n_pos_dims == 2 ? res * res : res * res * res) <= (uint64_t)hs; // interleaved chunks: entry = local * NCH + chunk (see k_grad_bin)
	if (c >= (dense ? (1u << NCH_LOG2) : (hs >> CL2))) return;
	uint32_t* cursor = a.cursors + ly * a.max_chunks + c;
	const uint32_t n_raw = *cursor;
	const uint32_t n = min(n_raw, a.cap);
	// entries this chunk owns: a dense level's chunk holds entries c, c + NCH, c + 2 NCH, ... < hs -- a few hundred for the coarse levels (the image model's sixteen levels
	// launch 2048 blocks for 7 10^5 entries), not 2^CL2
	const uint32_t n_entries = dense ? (hs > c ? (hs - c + (1u << NCH_LOG2) - 1u) >> NCH_LOG2 : 0u) : E;
	if (n_raw == 0u && !(ADAM && !dense)) return; // nothing listed: the gradients stay what they are (zero: cleared by the optimizer sweep / the step's memset); the cursor is zero already
	for (uint32_t f = 0; f < NF; ++f) for (uint32_t i = tid; i < n_entries; i += THREADS) acc[f * E + i] = 0ull;
	__syncthreads(); // every thread has read the cursor
	if (tid == 0) { // the list is empty again for the next step; SPLIT: the second of the two blocks that share it does that
		uint32_t* done = a.cursor_done + ly * a.max_chunks + c;
		if (!SPLIT) *cursor = 0u;
		else if (atomicAdd(done, 1u) == 1u) { *cursor = 0u; *done = 0u; }
	}
	const size_t base = ((size_t)ly * a.max_chunks + c) * a.cap;
	const uint16_t* idxs = a.idxs + base;
	constexpr int U = 8; // records per thread in flight
	if constexpr (SPLIT) {
		const uint32_t* vals32 = (const uint32_t*)((const uint2*)a.vals + base) + fp; // this block's half2 of every 8-byte record
		for (uint32_t i0 = tid; i0 < n; i0 += U * THREADS) {
			uint32_t v[U], id[U];
#pragma unroll
			for (int u = 0; u < U; ++u) { const uint32_t i = i0 + u * THREADS; if (i < n) { v[u] = vals32[(size_t)i * 2]; id[u] = idxs[i]; } }
#pragma unroll
			for (int u = 0; u < U; ++u) {
				if (i0 + u * THREADS >= n) break;
				atomicAdd(&acc[id[u]], (unsigned long long)half_bits_to_fixed(v[u] & 0xffffu));
				atomicAdd(&acc[E + id[u]], (unsigned long long)half_bits_to_fixed(v[u] >> 16));
			}
		}
	} else if constexpr (F == 4) {
		const uint2* vals = (const uint2*)a.vals + base;
		for (uint32_t i0 = tid; i0 < n; i0 += U * THREADS) {
			uint2 v[U]; uint32_t id[U];
#pragma unroll
			for (int u = 0; u < U; ++u) { const uint32_t i = i0 + u * THREADS; if (i < n) { v[u] = vals[i]; id[u] = idxs[i]; } }
#pragma unroll
			for (int u = 0; u < U; ++u) {
				if (i0 + u * THREADS >= n) break;
				atomicAdd(&acc[id[u]], (unsigned long long)half_bits_to_fixed(v[u].x & 0xffffu));
				atomicAdd(&acc[E + id[u]], (unsigned long long)half_bits_to_fixed(v[u].x >> 16));
				atomicAdd(&acc[2 * E + id[u]], (unsigned long long)half_bits_to_fixed(v[u].y & 0xffffu));
				atomicAdd(&acc[3 * E + id[u]], (unsigned long long)half_bits_to_fixed(v[u].y >> 16));
			}
		}
	} else {
		const uint32_t* vals = (const uint32_t*)a.vals + base;
		for (uint32_t i0 = tid; i0 < n; i0 += U * THREADS) {
			uint32_t v[U], id[U];
#pragma unroll
			for (int u = 0; u < U; ++u) { const uint32_t i = i0 + u * THREADS; if (i < n) { v[u] = vals[i]; id[u] = idxs[i]; } }
#pragma unroll
			for (int u = 0; u < U; ++u) {
				if (i0 + u * THREADS >= n) break;
				atomicAdd(&acc[id[u]], (unsigned long long)half_bits_to_fixed(v[u] & 0xffffu));
				atomicAdd(&acc[E + id[u]], (unsigned long long)half_bits_to_fixed(v[u] >> 16));
			}
		}
	}
	__syncthreads();
	if constexpr (SPLIT) {
		h2* gt = (h2*)((__half*)a.grid_grad_ + ((size_t)offset + ((size_t)c << CL2)) * 4) + fp;
		for (uint32_t e = tid; e < E; e += THREADS) {
			const h2 old = gt[(size_t)e * 2]; // zero unless a list overflowed
			const float s0 = (float)(long long)acc[e] * 0x1p-24f, s1 = (float)(long long)acc[E + e] * 0x1p-24f;
			const h2 r = {(_Float16)((float)old[0] + s0), (_Float16)((float)old[1] + s1)};
			gt[(size_t)e * 2] = r;
		}
	} else {
		typedef typename BinVal<F>::type val_t;
		val_t* gt = (val_t*)((__half*)a.grid_grad_ + ((size_t)offset + (dense ? (size_t)c : ((size_t)c << CL2))) * F);
		const uint32_t n_local = n_entries;
		if constexpr (ADAM) if (!dense) {
			// Fused optimizer step for this chunk (k_optimizer's arithmetic, entry by entry): the gradient the sweep would read is the half-rounded sum -- formed here the
			// same way and used from registers.  Only a list that overflowed has gradient mass in the table (k_grad_bin's fallback atomics): read and cleared then.
			const AdamArgs& o = a.adam;
			const bool overflow = n_raw > a.cap;
			const uint64_t i4_0 = o.n_mlp / 4 + (uint64_t)offset + ((uint64_t)c << CL2); // index in 4-parameter units (= table entries behind the MLP block)
			for (uint32_t e = tid; e < E; e += THREADS) {
				const uint64_t i4 = i4_0 + e;
				float r[4];
#pragma unroll
				for (int f = 0; f < 4; ++f) r[f] = (float)(long long)acc[f * E + e] * 0x1p-24f;
				if (overflow) {
					const uint2 oldv = gt[e];
					const _Float16* old = (const _Float16*)&oldv;
#pragma unroll
					for (int f = 0; f < 4; ++f) r[f] += (float)old[f];
					if (oldv.x | oldv.y) gt[e] = make_uint2(0u, 0u);
				}
				const h4 g4 = __builtin_bit_cast(h4, pack_halfs<4>(r));
				float g[4]; bool upd[4]; bool any = false;
#pragma unroll
				for (int k = 0; k < 4; ++k) { g[k] = (float)g4[k] / o.loss_scale; upd[k] = o.optimize_non_matrix != 0 && g[k] != 0.f; any |= upd[k]; }
				// the half parameters are ALWAYS the rounding of the fp32 masters (adam below, model_refresh_half): an entry whose masters are read anyway does not read them
				h4 w4; float4 mw = make_float4(0.f, 0.f, 0.f, 0.f);
				float* mwp = (float*)&mw;
				if (any || o.ema_full_precision) mw = ((const float4*)o.master)[i4];
				if (any) {
#pragma unroll
					for (int k = 0; k < 4; ++k) w4[k] = (_Float16)mwp[k];
					float4 m4 = ((const float4*)o.m)[i4], v4 = ((const float4*)o.v)[i4];
					uint2 st = ((const uint2*)o.steps)[i4];
					float* mp = (float*)&m4; float* vp = (float*)&v4; uint16_t* sp = (uint16_t*)&st;
#pragma unroll
					for (int k = 0; k < 4; ++k) {
						if (!upd[k]) continue;
						const float nw = adam_update(o, false, g[k], mwp[k], mp[k], vp[k], sp[k]);
						mwp[k] = nw;
						w4[k] = (_Float16)nw;
					}
					((float4*)o.master)[i4] = mw; ((float4*)o.m)[i4] = m4; ((float4*)o.v)[i4] = v4; ((uint2*)o.steps)[i4] = st;
					((uint2*)o.params)[i4] = __builtin_bit_cast(uint2, w4);
				} else w4 = __builtin_bit_cast(h4, ((const uint2*)o.params)[i4]);
				h4 inf4;
				if (o.ema_decay == 0.f) inf4 = w4; // no Ema wrapper: the inference parameters are the parameters
				else if (o.ema_full_precision) {
					float4 e4 = ((const float4*)o.ema)[i4];
					float* ep = (float*)&e4;
#pragma unroll
					for (int k = 0; k < 4; ++k) { const float filtered = ema_update(o, ep[k], mwp[k]); ep[k] = filtered; inf4[k] = (_Float16)filtered; }
					((float4*)o.ema)[i4] = e4;
				} else {
					const h4 old4 = __builtin_bit_cast(h4, ((const uint2*)o.params_inf)[i4]);
#pragma unroll
					for (int k = 0; k < 4; ++k) inf4[k] = (_Float16)ema_update(o, (float)old4[k], (float)w4[k]);
				}
				((uint2*)o.params_inf)[i4] = __builtin_bit_cast(uint2, inf4);
			}
			return;
		}
		for (uint32_t e = tid; e < n_local; e += THREADS) {
			const size_t o = dense ? ((size_t)e << NCH_LOG2) : (size_t)e;
			const val_t oldv = gt[o]; // zero unless a list overflowed
			const _Float16* old = (const _Float16*)&oldv;
			float r[F];
#pragma unroll
			for (int f = 0; f < F; ++f) r[f] = (float)old[f] + (float)(long long)acc[f * E + e] * 0x1p-24f;
			gt[o] = pack_halfs<F>(r);
		}
	}
}

// ---------------------------------------------------------------------------------------------
// Dense levels' scatter (the levels whose grid fits the table: 16^3 .. 64^3 in base.json), from the level-major dL/d(enc) that T1
// leaves in memory.  One lane = one sample, 64 consecutive samples (mostly of one ray) per wavefront: runs of samples in the same
// grid cell are summed in packed half by a segmented shuffle reduction and only the run head issues atomics; a lane QUAD issues the
// 16 bytes of an x-adjacent corner pair in one instruction (one memory-side request).  The memory side retires ~14 G atomic requests
// per second whatever the kernel around them does, so these requests are issued from a small kernel that shares the chip with the
// LDS-bound k_grad_bin / k_grad_accumulate and the MFMA-bound W instead of from T1's critical path.  Measured: T1 189 -> 90 us, this
// kernel 117 us, step time unchanged (the chip is throughput bound, not critical-path bound) -- kept as ablation DBG_T1_DENSE_EXTERNAL.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_grad_dense(GradDenseArgs a) {
	const uint32_t level = a.levels[blockIdx.y];
	const LevelConst lc = level_const_uniform(a.gm, level);
	const int lane = threadIdx.x & 63;
	const uint32_t s = blockIdx.x * 256 + threadIdx.x;
	if ((s & ~63u) >= a.n) return; // whole wavefront past the end
	const bool sv = s < a.n;
	const uint32_t sc = sv ? s : a.n - 1;
	const float* p = a.in + (size_t)sc * a.in_stride;
	Corners cr;
	level_corners(lc, p[0], p[1], p[2], cr);
	float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
	if (sv) { const h4 g = __builtin_bit_cast(h4, a.denc_lv[(size_t)level * a.denc_cap + sc]); g0 = (float)g[0]; g1 = (float)g[1]; g2 = (float)g[2]; g3 = (float)g[3]; }
	h2 v0[8], v1[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const float w = cr.w[k];
		h2 a0 = {(_Float16)(g0 * w), (_Float16)(g1 * w)}, a1 = {(_Float16)(g2 * w), (_Float16)(g3 * w)};
		v0[k] = a0; v1[k] = a1;
	}
	const uint32_t key = cr.cell_xy, key7 = cr.cell_z;
	const uint32_t pkey = (uint32_t)__shfl_up((int)key, 1, 64), pkey7 = (uint32_t)__shfl_up((int)key7, 1, 64);
	const bool head = lane == 0 || key != pkey || key7 != pkey7;
	const uint64_t hm = __ballot(head);
	bool issue = true;
	if (a.merge_runs && __popcll(hm) <= 48) {
		const uint64_t rest = lane == 63 ? 0ull : (hm >> (lane + 1));
		const uint32_t run_right = rest ? (uint32_t)(__ffsll((long long)rest) - 1) : (uint32_t)(63 - lane);
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const bool take = run_right >= (uint32_t)d;
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				const h2 t0 = __builtin_bit_cast(h2, __shfl_down(__builtin_bit_cast(int, v0[k]), d, 64));
				const h2 t1 = __builtin_bit_cast(h2, __shfl_down(__builtin_bit_cast(int, v1[k]), d, 64));
				if (take) { v0[k] += t0; v1[k] += t1; }
			}
		}
		issue = head;
	}
	__half* gt = (__half*)a.grid_grad_ + (size_t)lc.offset * 4;
	const bool own = issue && sv;
	const int r4 = lane & 3;
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		const bool go = __shfl((int)own, q, 4) != 0;
#pragma unroll
		for (int pp = 0; pp < 4; ++pp) {
			const uint32_t i0 = (uint32_t)__shfl((int)cr.idx[2 * pp], q, 4), i1 = (uint32_t)__shfl((int)cr.idx[2 * pp + 1], q, 4);
			const int a0 = __shfl(__builtin_bit_cast(int, v0[2 * pp]), q, 4), a1 = __shfl(__builtin_bit_cast(int, v1[2 * pp]), q, 4);
			const int b0 = __shfl(__builtin_bit_cast(int, v0[2 * pp + 1]), q, 4), b1 = __shfl(__builtin_bit_cast(int, v1[2 * pp + 1]), q, 4);
			const uint32_t ix = (r4 & 2) ? i1 : i0;
			const int val = r4 == 0 ? a0 : r4 == 1 ? a1 : r4 == 2 ? b0 : b1;
			if (go) atomic_add_h2(gt + (size_t)ix * 4 + (r4 & 1) * 2, __builtin_bit_cast(h2, val));
		}
	}
}

// ---------------------------------------------------------------------------------------------
// W: recompute forward + dgrad in chain AND swapped form, accumulate all weight gradients in registers.
// One 32-sample column tile per wave iteration; 12 accumulator tiles (192 registers) per wave.
// ---------------------------------------------------------------------------------------------
DEV h8 ident_frag(int s, int lane) { // fw_frag(I_32, mt = 0, s): transposes a fragment through the MFMA
	const int col = lane & 31, hi = lane >> 5;
	h8 r;
#pragma unroll
	for (int j = 0; j < 8; ++j) r[j] = (col == 16 * s + 8 * (j >> 2) + 4 * hi + (j & 3)) ? (_Float16)1.f : (_Float16)0.f;
	return r;
}
// swapped-layout tile (lane = neuron, regs = samples) -> two half operand fragments (q = 0,1)
DEV void sw_to_frags(const f16v& d, bool relu, h8 out[2]) {
#pragma unroll
	for (int q = 0; q < 2; ++q)
#pragma unroll
		for (int j = 0; j < 8; ++j) { float v = d[8 * q + j]; if (relu) v = v > 0.f ? v : 0.f; out[q][j] = (_Float16)v; }
}
// gradient tile in swapped layout, masked by the sign of the (post-ReLU) forward activation
DEV void sw_grad_to_frags(const f16v& d, const h8 act[2], h8 out[2]) {
#pragma unroll
	for (int q = 0; q < 2; ++q)
#pragma unroll
		for (int j = 0; j < 8; ++j) out[q][j] = ((float)act[q][j] > 0.f) ? (_Float16)d[8 * q + j] : (_Float16)0.f;
}

constexpr int N_DW_TILES = 12; // NR = 2: d1:(0,1) d2:(2,3) r1:(4,5) r2:(6..9) r3:(10,11)
constexpr int n_dw_tiles(int nr) { return 8 + 4 * (nr - 1); } // d1:(0,1) d2:(2,3) r1:(4,5) r2_k:(6+4k .. 9+4k) r3: the last two

// One wavefront keeps all dW tiles (128 / 192 / 256 accumulator registers for NR = 1 / 2 / 3 hidden colour layers) and owns its SIMD's register file.
// NR = 2 is the round-1 kernel (ablation DBG_W_SINGLE_ROLE; production runs the two-role k_wgrad2 below); NR = 1 and 3 (configs/nerf/base_1layer.json,
// base_3layer.json) run this one.  Per 32-sample tile: forward chain with the SWAPPED activations (lane = neuron, registers = samples) of every hidden
// colour layer kept for the weight-gradient products, then the dgrad chain in both layouts, layer by layer from the output.
template <int NR, int EX = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_wgrad_nr(ModelPtrs mp, const float* __restrict__ in, uint32_t in_stride, uint32_t n, const __half* __restrict__ dL_dy, uint32_t dy_stride,
		const uint4* __restrict__ enc_stash, float* __restrict__ partials) {
	constexpr int NT = n_dw_tiles(NR); // EX: + 2 tiles (columns 32..63 of the first colour layer's 64 x 48 matrix, 32..47 real) behind them
	extern __shared__ __attribute__((aligned(16))) char smem[];
	h8* fw = (h8*)smem;
	h8* bw = fw + (n_fw(NR) + (EX ? N_FW_EXTRA : 0)) * 64;
	load_frags_to_lds(fw, mp.fw_frags, n_fw(NR) + (EX ? N_FW_EXTRA : 0));
	load_frags_to_lds(bw, mp.bw_frags, n_bw(NR));
	__syncthreads();
	const int lane = threadIdx.x & 63, col = lane & 31, hi = lane >> 5, wid = threadIdx.x >> 6;
	const uint32_t wave = blockIdx.x * 4 + wid, n_waves = gridDim.x * 4;
	f16v dW[NT + (EX ? 2 : 0)];
#pragma unroll
	for (int t = 0; t < NT + (EX ? 2 : 0); ++t) dW[t] = zero16();
	const h8 I0 = ident_frag(0, lane), I1 = ident_frag(1, lane);

	for (uint32_t ct = wave; (uint64_t)ct * 32 < n; ct += n_waves) { // ct = 32-sample column tile
		asm volatile("" ::: "memory"); // the weight fragments in LDS are loop invariant: without this the compiler hoists their loads into (spilled) registers
		const uint32_t s_raw = ct * 32 + col;
		const bool valid = s_raw < n;
		const float* p = in + (size_t)min(s_raw, n - 1) * in_stride;
		FwdState<1> st;
		st.enc[0][0] = __builtin_bit_cast(h8, enc_stash[((size_t)ct * 2 + 0) * 64 + lane]);
		st.enc[0][1] = __builtin_bit_cast(h8, enc_stash[((size_t)ct * 2 + 1) * 64 + lane]);
		st.rin[0][1] = sh4_frag(p[4], p[5], p[6], hi);
		if constexpr (EX != 0) st.rin[0][2] = extra_frag(p + 7, mp.n_extra, hi);
		h8 dy0 = zero8(); _Float16 dsig = (_Float16)0.f;
		if (hi == 0 && valid) { // out-of-range columns contribute nothing: their output gradient is zero
			const h4 g = __builtin_bit_cast(h4, *(const uint2*)(dL_dy + (size_t)s_raw * dy_stride));
			dy0[0] = g[0]; dy0[1] = g[1]; dy0[2] = g[2]; dsig = g[3];
		}
		// ---- forward chain (ReLU masks) + swapped activations of the hidden colour layers ----
		fwd_density_l1<1>(fw, lane, st);
		fwd_density_l2<1>(fw, lane, st);
		h8 H_sw[NR][2][2]; // [layer][32-neuron tile][16-sample half]
#pragma unroll
		for (int kt = 0; kt < 2; ++kt) {
			f16v t = zero16();
#pragma unroll
			for (int s = 0; s < 2; ++s) t = mfma(st.rin[0][s], lds_frag(fw, FW_R1 + kt * 2 + s, lane), t);
			if constexpr (EX != 0) t = mfma(st.rin[0][2], lds_frag(fw, n_fw(NR) + kt, lane), t);
			sw_to_frags(t, true, H_sw[0][kt]);
		}
		fwd_rgb_l1<1, EX>(fw, lane, st, n_fw(NR));
#pragma unroll
		for (int k = 0; k < NR - 1; ++k) {
#pragma unroll
			for (int kt = 0; kt < 2; ++kt) {
				f16v t = zero16();
#pragma unroll
				for (int s = 0; s < 4; ++s) t = mfma(st.hb[0][s], lds_frag(fw, FW_R2 + 8 * k + kt * 4 + s, lane), t);
				sw_to_frags(t, true, H_sw[k + 1][kt]);
			}
			fwd_rgb_l2<1>(fw, lane, st, k);
		}
		// ---- backward ----
		h8 g_sw[2]; // an output gradient in swapped layout
		// output layer: dW = d_out * H_last^T
		{ f16v t = mfma(dy0, I0, zero16()); sw_to_frags(t, false, g_sw); }
#pragma unroll
		for (int kt = 0; kt < 2; ++kt)
#pragma unroll
			for (int q = 0; q < 2; ++q) dW[NT - 2 + kt] = mfma(g_sw[q], H_sw[NR - 1][kt][q], dW[NT - 2 + kt]);
		// gradient at the last hidden layer, chain form (masked by the ReLU bits) and swapped (masked by the activation's sign)
		h8 dh[4];
		h8 d_sw[2][2];
#pragma unroll
		for (int mt = 0; mt < 2; ++mt) {
			const h8 a = lds_frag(bw, bw_r3(NR) + mt, lane);
			const uint32_t mask = NR == 1 ? st.m1r[0] : st.m2r[NR >= 2 ? NR - 2 : 0][0];
			f16v d = mfma(a, dy0, zero16());
			dh[2 * mt + 0] = to_frag_masked(d, 0, mask >> (16 * mt));
			dh[2 * mt + 1] = to_frag_masked(d, 1, mask >> (16 * mt));
			f16v dsw = mfma(dy0, a, zero16());
			sw_grad_to_frags(dsw, H_sw[NR - 1][mt], d_sw[mt]);
		}
		// the 64 x 64 layers, last to first: dW(k)[it][kt] = d_H(k+2)[it] * H(k+1)[kt]^T, then the gradient at H(k+1)
#pragma unroll
		for (int k = NR - 2; k >= 0; --k) {
#pragma unroll
			for (int it = 0; it < 2; ++it)
#pragma unroll
				for (int kt = 0; kt < 2; ++kt)
#pragma unroll
					for (int q = 0; q < 2; ++q) dW[6 + 4 * k + it * 2 + kt] = mfma(d_sw[it][q], H_sw[k][kt][q], dW[6 + 4 * k + it * 2 + kt]);
			h8 dhn[4];
			h8 dn_sw[2][2];
#pragma unroll
			for (int mt = 0; mt < 2; ++mt) {
				f16v d = zero16(), dsw = zero16();
#pragma unroll
				for (int s = 0; s < 4; ++s) {
					const h8 a = lds_frag(bw, BW_R2 + 8 * k + mt * 4 + s, lane);
					d = mfma(a, dh[s], d);
					dsw = mfma(dh[s], a, dsw);
				}
				const uint32_t mask = k == 0 ? st.m1r[0] : st.m2r[k >= 1 ? k - 1 : 0][0];
				dhn[2 * mt + 0] = to_frag_masked(d, 0, mask >> (16 * mt));
				dhn[2 * mt + 1] = to_frag_masked(d, 1, mask >> (16 * mt));
				sw_grad_to_frags(dsw, H_sw[k][mt], dn_sw[mt]);
			}
#pragma unroll
			for (int x = 0; x < 4; ++x) dh[x] = dhn[x];
#pragma unroll
			for (int x = 0; x < 2; ++x) { d_sw[x][0] = dn_sw[x][0]; d_sw[x][1] = dn_sw[x][1]; }
		}
		// first colour layer: dW[it][0] = d_H1[it] * rin^T
		{
			h8 rin_sw[2];
			f16v t = mfma(st.rin[0][0], I0, zero16()); t = mfma(st.rin[0][1], I1, t); sw_to_frags(t, false, rin_sw);
#pragma unroll
			for (int it = 0; it < 2; ++it)
#pragma unroll
				for (int q = 0; q < 2; ++q) dW[4 + it] = mfma(d_sw[it][q], rin_sw[q], dW[4 + it]);
			if constexpr (EX != 0) { // columns 32..47: the extra dims and the padding ones (lanes 16..31 of the swapped tile stay zero)
				h8 x_sw[2];
				f16v tx = mfma(st.rin[0][2], I0, zero16()); sw_to_frags(tx, false, x_sw);
#pragma unroll
				for (int it = 0; it < 2; ++it)
#pragma unroll
					for (int q = 0; q < 2; ++q) dW[NT + it] = mfma(d_sw[it][q], x_sw[q], dW[NT + it]);
			}
		}
		// d_rin -> d_densout (+ dsigma on neuron 0)
		h8 ddens;
		{
			f16v d = zero16();
#pragma unroll
			for (int s = 0; s < 4; ++s) d = mfma(lds_frag(bw, BW_R1 + s, lane), dh[s], d);
			uint32_t dummy = 0;
			ddens = to_frag<false>(d, 0, dummy);
			if (hi == 0) ddens[0] = (_Float16)((float)ddens[0] + (float)dsig);
		}
		{ f16v t = mfma(ddens, I0, zero16()); sw_to_frags(t, false, g_sw); }
		// h1d swapped (from the encoding); density L2: dW[0][kt] = d_densout * h1d[kt]^T
		h8 h1d_sw[2][2];
#pragma unroll
		for (int kt = 0; kt < 2; ++kt) {
			f16v t = zero16();
#pragma unroll
			for (int s = 0; s < 2; ++s) t = mfma(st.enc[0][s], lds_frag(fw, FW_D1 + kt * 2 + s, lane), t);
			sw_to_frags(t, true, h1d_sw[kt]);
		}
#pragma unroll
		for (int kt = 0; kt < 2; ++kt)
#pragma unroll
			for (int q = 0; q < 2; ++q) dW[2 + kt] = mfma(g_sw[q], h1d_sw[kt][q], dW[2 + kt]);
		// d_h1d swapped; density L1: dW[it][0] = d_h1d[it] * enc^T
		h8 enc_sw[2];
		{ f16v t = mfma(st.enc[0][0], I0, zero16()); t = mfma(st.enc[0][1], I1, t); sw_to_frags(t, false, enc_sw); }
#pragma unroll
		for (int it = 0; it < 2; ++it) {
			const h8 a = lds_frag(bw, BW_D2 + it, lane);
			f16v dsw = mfma(ddens, a, zero16());
			h8 dd_sw[2];
			sw_grad_to_frags(dsw, h1d_sw[it], dd_sw);
#pragma unroll
			for (int q = 0; q < 2; ++q) dW[0 + it] = mfma(dd_sw[q], enc_sw[q], dW[0 + it]);
		}
	}

	// ---- reduce the 4 waves of the block through LDS (re-using the fragment region), 4 tiles at a time: 16 KiB <= the 28 KiB of fragments at NR = 1 ----
	float* red = (float*)smem;
	float* dstp = partials + (size_t)blockIdx.x * ((NT + (EX ? 2 : 0)) * 16 * 64);
#pragma unroll
	for (int part = 0; part < NT / 4 + (EX ? 1 : 0); ++part) {
		const int nt = part < NT / 4 ? 4 : 2; // the two extra tiles are a last, half-sized part
		__syncthreads();
		for (int w = 0; w < 4; ++w) {
			if (wid == w) {
#pragma unroll
				for (int t = 0; t < 4; ++t)
#pragma unroll
					for (int r = 0; r < 16; ++r) {
						if (t >= nt) continue;
						float* dst = red + ((size_t)t * 16 + r) * 64 + lane;
						*dst = (w == 0 ? 0.f : *dst) + dW[part * 4 + t < NT + (EX ? 2 : 0) ? part * 4 + t : 0][r];
					}
			}
			__syncthreads();
		}
		for (int i = threadIdx.x; i < nt * 16 * 64; i += blockDim.x) dstp[part * 4 * 16 * 64 + i] = red[i];
	}
}

// ---------------------------------------------------------------------------------------------
// W, two roles per workgroup (round 2).  k_wgrad keeps all 12 dW tiles in one wavefront: 192 accumulators + the working set spill 104
// VGPRs to scratch, the wave owns its SIMD's whole register file (nothing else can run there) and has nobody to hide its LDS / scratch /
// global latencies behind (matrix pipe busy 7.7 % of the kernel, profiles/r02_pmc_mfma_tcc.txt).  Here a workgroup has 8 waves:
// waves 0-3 (role A) accumulate the density-net tiles and the colour net's first layer  (d1, d2, r1: 6 tiles),
// waves 4-7 (role B) the colour net's second and third layer (r2, r3: 6 tiles);
// wave w and wave w+4 land on the same SIMD and work on the SAME 32-sample tiles, each recomputing the part of the forward / dgrad chain
// it needs (~16 % more MFMA work in total) with 96 accumulators and no spills, two waves per SIMD.
// ---------------------------------------------------------------------------------------------
template <int ROLE>
DEV void wgrad_tile(const h8* fw, const h8* bw, int lane, int hi, bool valid, const float* __restrict__ p, const uint4* __restrict__ enc_stash, uint32_t ct,
		const __half* __restrict__ dL_dy, uint32_t dy_stride, uint32_t s_raw, const h8& I0, const h8& I1, f16v dW[6]) {
	FwdState<1> st;
	st.enc[0][0] = __builtin_bit_cast(h8, enc_stash[((size_t)ct * 2 + 0) * 64 + lane]);
	st.enc[0][1] = __builtin_bit_cast(h8, enc_stash[((size_t)ct * 2 + 1) * 64 + lane]);
	st.rin[0][1] = sh4_frag(p[4], p[5], p[6], hi);
	h8 dy0 = zero8(); _Float16 dsig = (_Float16)0.f;
	if (hi == 0 && valid) {
		const h4 g = __builtin_bit_cast(h4, *(const uint2*)(dL_dy + (size_t)s_raw * dy_stride));
		dy0[0] = g[0]; dy0[1] = g[1]; dy0[2] = g[2]; dsig = g[3];
	}
	fwd_density_l1<1>(fw, lane, st);
	fwd_density_l2<1>(fw, lane, st);
	fwd_rgb_l1<1>(fw, lane, st);
	// swapped activations of the colour net's first hidden layer: both roles mask gradients with them
	h8 h1r_sw[2][2];
#pragma unroll
	for (int kt = 0; kt < 2; ++kt) {
		f16v t = zero16();
#pragma unroll
		for (int s = 0; s < 2; ++s) t = mfma(st.rin[0][s], lds_frag(fw, FW_R1 + kt * 2 + s, lane), t);
		sw_to_frags(t, true, h1r_sw[kt]);
	}
	if (ROLE == 1) {
		// ---- role B: r3 = d_out x h2r^T, r2 = d_h2 x h1r^T ----
		h8 h2r_sw[2][2];
#pragma unroll
		for (int kt = 0; kt < 2; ++kt) {
			f16v t = zero16();
#pragma unroll
			for (int s = 0; s < 4; ++s) t = mfma(st.hb[0][s], lds_frag(fw, FW_R2 + kt * 4 + s, lane), t);
			sw_to_frags(t, true, h2r_sw[kt]);
		}
		h8 g_sw[2];
		{ f16v t = mfma(dy0, I0, zero16()); sw_to_frags(t, false, g_sw); }
#pragma unroll
		for (int kt = 0; kt < 2; ++kt)
#pragma unroll
			for (int q = 0; q < 2; ++q) dW[4 + kt] = mfma(g_sw[q], h2r_sw[kt][q], dW[4 + kt]);
#pragma unroll
		for (int it = 0; it < 2; ++it) {
			const h8 a = lds_frag(bw, BW_R3 + it, lane);
			f16v dsw = mfma(dy0, a, zero16());
			h8 d2_sw[2];
			sw_grad_to_frags(dsw, h2r_sw[it], d2_sw);
#pragma unroll
			for (int kt = 0; kt < 2; ++kt)
#pragma unroll
				for (int q = 0; q < 2; ++q) dW[it * 2 + kt] = mfma(d2_sw[q], h1r_sw[kt][q], dW[it * 2 + kt]);
		}
		return;
	}
	// ---- role A: r1 = d_h1r x rin^T, d2 = d_densout x h1d^T, d1 = d_h1d x enc^T ----
	fwd_rgb_l2<1>(fw, lane, st); // only the ReLU mask m2r is consumed below
	h8 dh[4];
#pragma unroll
	for (int mt = 0; mt < 2; ++mt) {
		f16v d = mfma(lds_frag(bw, BW_R3 + mt, lane), dy0, zero16());
		dh[2 * mt + 0] = to_frag_masked(d, 0, st.m2r[0][0] >> (16 * mt));
		dh[2 * mt + 1] = to_frag_masked(d, 1, st.m2r[0][0] >> (16 * mt));
	}
	h8 dh1[4];
	{
		h8 rin_sw[2];
		{ f16v t = mfma(st.rin[0][0], I0, zero16()); t = mfma(st.rin[0][1], I1, t); sw_to_frags(t, false, rin_sw); }
#pragma unroll
		for (int mt = 0; mt < 2; ++mt) {
			f16v d = zero16(), dsw = zero16();
#pragma unroll
			for (int s = 0; s < 4; ++s) {
				const h8 a = lds_frag(bw, BW_R2 + mt * 4 + s, lane);
				d = mfma(a, dh[s], d);
				dsw = mfma(dh[s], a, dsw);
			}
			dh1[2 * mt + 0] = to_frag_masked(d, 0, st.m1r[0] >> (16 * mt));
			dh1[2 * mt + 1] = to_frag_masked(d, 1, st.m1r[0] >> (16 * mt));
			h8 d1_sw[2];
			sw_grad_to_frags(dsw, h1r_sw[mt], d1_sw);
#pragma unroll
			for (int q = 0; q < 2; ++q) dW[4 + mt] = mfma(d1_sw[q], rin_sw[q], dW[4 + mt]);
		}
	}
	h8 ddens;
	{
		f16v d = zero16();
#pragma unroll
		for (int s = 0; s < 4; ++s) d = mfma(lds_frag(bw, BW_R1 + s, lane), dh1[s], d);
		uint32_t dummy = 0;
		ddens = to_frag<false>(d, 0, dummy);
		if (hi == 0) ddens[0] = (_Float16)((float)ddens[0] + (float)dsig);
	}
	h8 g_sw[2];
	{ f16v t = mfma(ddens, I0, zero16()); sw_to_frags(t, false, g_sw); }
	h8 h1d_sw[2][2];
#pragma unroll
	for (int kt = 0; kt < 2; ++kt) {
		f16v t = zero16();
#pragma unroll
		for (int s = 0; s < 2; ++s) t = mfma(st.enc[0][s], lds_frag(fw, FW_D1 + kt * 2 + s, lane), t);
		sw_to_frags(t, true, h1d_sw[kt]);
	}
#pragma unroll
	for (int kt = 0; kt < 2; ++kt)
#pragma unroll
		for (int q = 0; q < 2; ++q) dW[2 + kt] = mfma(g_sw[q], h1d_sw[kt][q], dW[2 + kt]);
	h8 enc_sw[2];
	{ f16v t = mfma(st.enc[0][0], I0, zero16()); t = mfma(st.enc[0][1], I1, t); sw_to_frags(t, false, enc_sw); }
#pragma unroll
	for (int it = 0; it < 2; ++it) {
		f16v dsw = mfma(ddens, lds_frag(bw, BW_D2 + it, lane), zero16());
		h8 dd_sw[2];
		sw_grad_to_frags(dsw, h1d_sw[it], dd_sw);
#pragma unroll
		for (int q = 0; q < 2; ++q) dW[0 + it] = mfma(dd_sw[q], enc_sw[q], dW[0 + it]);
	}
}

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_wgrad2(ModelPtrs mp, const float* __restrict__ in, uint32_t in_stride, uint32_t n, const __half* __restrict__ dL_dy, uint32_t dy_stride,
		const uint4* __restrict__ enc_stash, float* __restrict__ partials) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	h8* fw = (h8*)smem;
	h8* bw = fw + N_FW_FRAGS * 64;
	load_frags_to_lds(fw, mp.fw_frags, N_FW_FRAGS);
	load_frags_to_lds(bw, mp.bw_frags, N_BW_FRAGS);
	__syncthreads();
	const int lane = threadIdx.x & 63, col = lane & 31, hi = lane >> 5, wid = threadIdx.x >> 6, role = wid >> 2, w4 = wid & 3;
	const uint32_t wave = blockIdx.x * 4 + w4, n_waves = gridDim.x * 4; // the two roles walk the same tiles
	f16v dW[6]; // role A: d1 (0,1), d2 (2,3), r1 (4,5) = tiles 0..5; role B: r2 (0..3), r3 (4,5) = tiles 6..11
#pragma unroll
	for (int t = 0; t < 6; ++t) dW[t] = zero16();
	const h8 I0 = ident_frag(0, lane), I1 = ident_frag(1, lane);
	for (uint32_t ct = wave; (uint64_t)ct * 32 < n; ct += n_waves) {
		const uint32_t s_raw = ct * 32 + col;
		const bool valid = s_raw < n;
		const float* p = in + (size_t)min(s_raw, n - 1) * in_stride;
		if (role == 0) wgrad_tile<0>(fw, bw, lane, hi, valid, p, enc_stash, ct, dL_dy, dy_stride, s_raw, I0, I1, dW);
		else wgrad_tile<1>(fw, bw, lane, hi, valid, p, enc_stash, ct, dL_dy, dy_stride, s_raw, I0, I1, dW);
	}
	// reduce the 4 waves of each role through LDS (re-using the fragment region: 6 tiles * 16 regs * 64 lanes * 4 B = 24 KiB), one role at a time
	float* red = (float*)smem;
	float* dstp = partials + (size_t)blockIdx.x * (N_DW_TILES * 16 * 64);
	for (int r = 0; r < 2; ++r) {
		__syncthreads();
		for (int w = 0; w < 4; ++w) {
			if (role == r && w4 == w) {
#pragma unroll
				for (int t = 0; t < 6; ++t)
#pragma unroll
					for (int q = 0; q < 16; ++q) {
						float* dst = red + ((size_t)t * 16 + q) * 64 + lane;
						*dst = (w == 0 ? 0.f : *dst) + dW[t][q];
					}
			}
			__syncthreads();
		}
		for (int i = threadIdx.x; i < 6 * 16 * 64; i += blockDim.x) dstp[r * 6 * 16 * 64 + i] = red[i];
	}
}

// ---------------------------------------------------------------------------------------------
// T1 + W in ONE kernel (round 5; base.json's shape with the lazy K2's encoding stash): the two roles of k_wgrad2, role A also emitting what T1 emits.
// k_wgrad2's role A already evaluates T1's whole chain (forward, dgrad down to d(density-net hidden layer)) to get its operands, so T1 was a second, separate
// evaluation of the same chain (25 us) whose only product of its own is dL/d(enc) = W1d^T * d_h1d: here role A computes it from the d_h1d it holds (2 + 4 more
// MFMAs per 32 samples) and stores it level-major for k_grad_bin.  The encodings come straight from K2's per-sample records (EncStashIn), no lane-linear copy
// in between.
// Fragment conversions (the VALU between two MFMAs was the bound of both kernels: 2,420 VALU instructions for 108 MFMAs per tile pair in k_wgrad2, i.e. 9,700 issue
// cycles against 3,500): fp32 -> half as v_cvt_pk_f16_f32 (two elements per instruction, RNE like the scalar conversion), ReLU as v_pk_max_i16 on the half bit
// patterns (exact: negative halfs and -0 are negative int16), the per-lane ReLU bit masks of the chain layout accumulated with v_dot2_u32_u16 over the 0/1 flags of a
// packed pair (v_pk_min_u16(a, 1)), and gradients masked with the half-domain test `activation != 0` -- the stored forward activation, as tcnn's backward and the
// oracle do (the scalar helpers above test the fp32 pre-activation: the two differ only for 0 < x < 2^-25, which rounds to a zero activation).
// ---------------------------------------------------------------------------------------------
typedef float f2v __attribute__((ext_vector_type(2)));
typedef short s2v __attribute__((ext_vector_type(2)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
DEV uint32_t cvt_pk(float a, float b) { const f2v v = {a, b}; return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h2)); }
DEV uint32_t relu_pk(uint32_t h) { const s2v z = {0, 0}; return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s2v, h), z)); } // v_pk_max_i16
// 1 per non-zero half (a >= +0 as a half: after relu_pk).  (inline asm: LLVM canonicalises umin(x, 1) to a compare + select per half, four instructions for this one)
DEV uint32_t flags_pk(uint32_t a) { uint32_t r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(0x00010001u)); return r; }
DEV uint32_t nz_mask_pk(uint32_t a) { return __umul24(flags_pk(a), 0xffffu); } // 0xffff per non-zero half
// D tile (fp32, regs 8Q .. 8Q+7) -> half fragment
template <int Q> DEV h8 frag_plain(const f16v& d) {
	u4v r;
#pragma unroll
	for (int p = 0; p < 4; ++p) r[p] = cvt_pk(d[8 * Q + 2 * p], d[8 * Q + 2 * p + 1]);
	return __builtin_bit_cast(h8, r);
}
// ... with ReLU.  `bits` records which elements are non-zero, one 16-bit pattern per D tile spread over both halves of the word so that a packed pair's two flags
// (bits 0 and 16 of flags_pk) go in with one shift-or: element 2p + h of registers 8Q .. 8Q+7 <-> bit 16 h + 4 Q + p (a second D tile of the layer sits 8 bits higher)
template <int Q> DEV h8 frag_relu_bits(const f16v& d, uint32_t& bits) {
	u4v r;
#pragma unroll
	for (int p = 0; p < 4; ++p) {
		r[p] = relu_pk(cvt_pk(d[8 * Q + 2 * p], d[8 * Q + 2 * p + 1]));
		bits |= flags_pk(r[p]) << (4 * Q + p);
	}
	return __builtin_bit_cast(h8, r);
}
// ... the ReLU bits alone (the activation itself is not needed)
template <int Q> DEV void relu_bits_only(const f16v& d, uint32_t& bits) {
#pragma unroll
	for (int p = 0; p < 4; ++p) bits |= flags_pk(relu_pk(cvt_pk(d[8 * Q + 2 * p], d[8 * Q + 2 * p + 1]))) << (4 * Q + p);
}
// ... masked by the ReLU bits of its D tile (layout above)
template <int Q> DEV h8 frag_masked_bits(const f16v& d, uint32_t bits) {
	u4v r;
#pragma unroll
	for (int p = 0; p < 4; ++p) r[p] = cvt_pk(d[8 * Q + 2 * p], d[8 * Q + 2 * p + 1]) & __umul24((bits >> (4 * Q + p)) & 0x00010001u, 0xffffu);
	return __builtin_bit_cast(h8, r);
}
// swapped-layout tile (lane = neuron, regs = samples) -> two operand fragments: plain, with ReLU, or masked by the (post-ReLU) forward activation
DEV void swf_plain(const f16v& d, h8 out[2]) { out[0] = frag_plain<0>(d); out[1] = frag_plain<1>(d); }
DEV void swf_relu(const f16v& d, h8 out[2]) {
#pragma unroll
	for (int q = 0; q < 2; ++q) {
		u4v r;
#pragma unroll
		for (int p = 0; p < 4; ++p) r[p] = relu_pk(cvt_pk(d[8 * q + 2 * p], d[8 * q + 2 * p + 1]));
		out[q] = __builtin_bit_cast(h8, r);
	}
}
DEV void swf_masked(const f16v& d, const h8 act[2], h8 out[2]) {
#pragma unroll
	for (int q = 0; q < 2; ++q) {
		u4v r; const u4v a = __builtin_bit_cast(u4v, act[q]);
#pragma unroll
		for (int p = 0; p < 4; ++p) r[p] = cvt_pk(d[8 * q + 2 * p], d[8 * q + 2 * p + 1]) & nz_mask_pk(a[p]);
		out[q] = __builtin_bit_cast(h8, r);
	}
}
// the density network + the colour network's first layer in chain layout (what both roles need): hb = the colour net's first hidden activation, rin[0] = the
// density net's output, ReLU bits of the two 64-wide layers (frag_relu_bits' layout, row tile mt shifted by 8 mt)
struct FusedFwd { h8 enc[2]; h8 rin[2]; h8 hb[4]; uint32_t m1d, m1r; };
template <bool NEED_M1D>
DEV void fused_fwd_to_h1r(const h8* fw, int lane, FusedFwd& st) {
	h8 h1d[4];
	st.m1d = 0;
#pragma unroll
	for (int mt = 0; mt < 2; ++mt) {
		f16v acc = zero16();
#pragma unroll
		for (int s = 0; s < 2; ++s) acc = mfma(lds_frag(fw, FW_D1 + mt * 2 + s, lane), st.enc[s], acc);
		uint32_t mb = 0;
		if (NEED_M1D) { h1d[2 * mt] = frag_relu_bits<0>(acc, mb); h1d[2 * mt + 1] = frag_relu_bits<1>(acc, mb); st.m1d |= mb << (8 * mt); }
		else { h8 t[2]; swf_relu(acc, t); h1d[2 * mt] = t[0]; h1d[2 * mt + 1] = t[1]; }
	}
	{
		f16v acc = zero16();
#pragma unroll
		for (int s = 0; s < 4; ++s) acc = mfma(lds_frag(fw, FW_D2 + s, lane), h1d[s], acc);
		st.rin[0] = frag_plain<0>(acc);
	}
	st.m1r = 0;
#pragma unroll
	for (int mt = 0; mt < 2; ++mt) {
		f16v acc = zero16();
#pragma unroll
		for (int s = 0; s < 2; ++s) acc = mfma(lds_frag(fw, FW_R1 + mt * 2 + s, lane), st.rin[s], acc);
		uint32_t mb = 0;
		st.hb[2 * mt] = frag_relu_bits<0>(acc, mb); st.hb[2 * mt + 1] = frag_relu_bits<1>(acc, mb);
		st.m1r |= mb << (8 * mt);
	}
}
// Roles (two wavefronts per SIMD work on the SAME 32-sample tile): role A = the density network's weight gradients (d1, d2: 4 tiles) + dL/d(enc) -- it needs the whole
// chain, forward and backward --, role B = the colour network's (r1, r2, r3: 8 tiles), which needs the forward chain and the colour network's backward products only.
// (k_wgrad2 gave r1 to role A: 75 MFMAs + 720 VALU instructions per tile against 39 + 250 in role B, and role A is the pole; this split is 63 / 61 MFMAs.)
constexpr int FUSED_NT_A = 4, FUSED_NT_B = 8;
template <int ROLE>
DEV void fused_tile(const h8* fw, const h8* bw, const h8* idf, int lane, int hi, bool valid, const float* __restrict__ p, const uint4* __restrict__ e,
		const __half* __restrict__ dL_dy, uint32_t dy_stride, uint32_t s_raw, f16v dW[FUSED_NT_B], uint2* __restrict__ denc_lv, uint32_t denc_cap) {
	FusedFwd st;
	st.enc[0] = __builtin_bit_cast(h8, e[0]); st.enc[1] = __builtin_bit_cast(h8, e[1]);
	__builtin_amdgcn_sched_barrier(0); // (as in T1: keeps the layers' LDS fragment loads behind the two global loads)
	st.rin[1] = sh4_frag(p[4], p[5], p[6], hi);
	h8 dy0 = zero8(); _Float16 dsig = (_Float16)0.f;
	if (hi == 0 && valid) {
		const h4 g = __builtin_bit_cast(h4, *(const uint2*)(dL_dy + (size_t)s_raw * dy_stride));
		dy0[0] = g[0]; dy0[1] = g[1]; dy0[2] = g[2]; dsig = g[3];
	}
	fused_fwd_to_h1r<ROLE == 0>(fw, lane, st);
	// ReLU state of the colour net's second hidden layer in chain layout: only the bits are needed (both roles mask R3^T * d_out with them)
	uint32_t m2r = 0;
#pragma unroll
	for (int mt = 0; mt < 2; ++mt) {
		f16v acc = zero16();
#pragma unroll
		for (int s = 0; s < 4; ++s) acc = mfma(lds_frag(fw, FW_R2 + mt * 4 + s, lane), st.hb[s], acc);
		uint32_t mb = 0;
		relu_bits_only<0>(acc, mb); relu_bits_only<1>(acc, mb);
		m2r |= mb << (8 * mt);
	}
	auto colour_dh = [&](h8 dh[4]) { // d(second hidden colour layer) in chain layout = ReLU-masked R3^T * d_out
#pragma unroll
		for (int mt = 0; mt < 2; ++mt) {
			f16v d = mfma(lds_frag(bw, BW_R3 + mt, lane), dy0, zero16());
			dh[2 * mt + 0] = frag_masked_bits<0>(d, m2r >> (8 * mt));
			dh[2 * mt + 1] = frag_masked_bits<1>(d, m2r >> (8 * mt));
		}
	};
	if (ROLE == 1) {
		// ---- role B: r3 = d_out x h2r^T, r2 = d_h2 x h1r^T, r1 = d_h1r x rin^T  (dW[0..3] = r2, dW[4..5] = r3, dW[6..7] = r1) ----
		h8 h1r_sw[2][2]; // swapped activations of the colour net's first hidden layer
#pragma unroll
		for (int kt = 0; kt < 2; ++kt) {
			f16v t = zero16();
#pragma unroll
			for (int s = 0; s < 2; ++s) t = mfma(st.rin[s], lds_frag(fw, FW_R1 + kt * 2 + s, lane), t);
			swf_relu(t, h1r_sw[kt]);
		}
		{
			h8 h2r_sw[2][2];
#pragma unroll
			for (int kt = 0; kt < 2; ++kt) {
				f16v t = zero16();
#pragma unroll
				for (int s = 0; s < 4; ++s) t = mfma(st.hb[s], lds_frag(fw, FW_R2 + kt * 4 + s, lane), t);
				swf_relu(t, h2r_sw[kt]);
			}
			h8 g_sw[2];
			{ f16v t = mfma(dy0, lds_frag(idf, 0, lane), zero16()); swf_plain(t, g_sw); }
#pragma unroll
			for (int kt = 0; kt < 2; ++kt)
#pragma unroll
				for (int q = 0; q < 2; ++q) dW[4 + kt] = mfma(g_sw[q], h2r_sw[kt][q], dW[4 + kt]);
#pragma unroll
			for (int it = 0; it < 2; ++it) {
				f16v dsw = mfma(dy0, lds_frag(bw, BW_R3 + it, lane), zero16());
				h8 d2_sw[2];
				swf_masked(dsw, h2r_sw[it], d2_sw);
#pragma unroll
				for (int kt = 0; kt < 2; ++kt)
#pragma unroll
					for (int q = 0; q < 2; ++q) dW[it * 2 + kt] = mfma(d2_sw[q], h1r_sw[kt][q], dW[it * 2 + kt]);
			}
		}
		h8 rin_sw[2];
		{ f16v t = mfma(st.rin[0], lds_frag(idf, 0, lane), zero16()); t = mfma(st.rin[1], lds_frag(idf, 1, lane), t); swf_plain(t, rin_sw); }
		__builtin_amdgcn_sched_barrier(0); // (r1 behind r2 / r3: h2r_sw is dead before dh comes alive, 128 accumulator registers leave no room for both)
		h8 dh[4];
		colour_dh(dh);
#pragma unroll
		for (int mt = 0; mt < 2; ++mt) {
			f16v dsw = zero16();
#pragma unroll
			for (int s = 0; s < 4; ++s) dsw = mfma(dh[s], lds_frag(bw, BW_R2 + mt * 4 + s, lane), dsw);
			h8 d1_sw[2];
			swf_masked(dsw, h1r_sw[mt], d1_sw);
#pragma unroll
			for (int q = 0; q < 2; ++q) dW[6 + mt] = mfma(d1_sw[q], rin_sw[q], dW[6 + mt]);
		}
		return;
	}
	// ---- role A: d2 = d_densout x h1d^T, d1 = d_h1d x enc^T, and T1's dL/d(enc)  (dW[0..1] = d1, dW[2..3] = d2) ----
	h8 dh[4];
	colour_dh(dh);
	h8 dh1[4];
#pragma unroll
	for (int mt = 0; mt < 2; ++mt) {
		f16v d = zero16();
#pragma unroll
		for (int s = 0; s < 4; ++s) d = mfma(lds_frag(bw, BW_R2 + mt * 4 + s, lane), dh[s], d);
		dh1[2 * mt + 0] = frag_masked_bits<0>(d, st.m1r >> (8 * mt));
		dh1[2 * mt + 1] = frag_masked_bits<1>(d, st.m1r >> (8 * mt));
	}
	h8 ddens;
	{
		f16v d = zero16();
#pragma unroll
		for (int s = 0; s < 4; ++s) d = mfma(lds_frag(bw, BW_R1 + s, lane), dh1[s], d);
		ddens = frag_plain<0>(d);
		if (hi == 0) ddens[0] = (_Float16)((float)ddens[0] + (float)dsig); // add_density_gradient (nerf_network.h:235): half add into density-net output 0
	}
	h8 g_sw[2];
	{ f16v t = mfma(ddens, lds_frag(idf, 0, lane), zero16()); swf_plain(t, g_sw); }
	h8 h1d_sw[2][2];
#pragma unroll
	for (int kt = 0; kt < 2; ++kt) {
		f16v t = zero16();
#pragma unroll
		for (int s = 0; s < 2; ++s) t = mfma(st.enc[s], lds_frag(fw, FW_D1 + kt * 2 + s, lane), t);
		swf_relu(t, h1d_sw[kt]);
	}
#pragma unroll
	for (int kt = 0; kt < 2; ++kt)
#pragma unroll
		for (int q = 0; q < 2; ++q) dW[2 + kt] = mfma(g_sw[q], h1d_sw[kt][q], dW[2 + kt]);
	h8 enc_sw[2];
	{ f16v t = mfma(st.enc[0], lds_frag(idf, 0, lane), zero16()); t = mfma(st.enc[1], lds_frag(idf, 1, lane), t); swf_plain(t, enc_sw); }
	h8 dh1d[4]; // chain layout: the operand of T1's last product
#pragma unroll
	for (int it = 0; it < 2; ++it) {
		const h8 a = lds_frag(bw, BW_D2 + it, lane);
		f16v dsw = mfma(ddens, a, zero16());
		h8 dd_sw[2];
		swf_masked(dsw, h1d_sw[it], dd_sw);
#pragma unroll
		for (int q = 0; q < 2; ++q) dW[0 + it] = mfma(dd_sw[q], enc_sw[q], dW[0 + it]);
		f16v d = mfma(a, ddens, zero16());
		dh1d[2 * it + 0] = frag_masked_bits<0>(d, st.m1d >> (8 * it));
		dh1d[2 * it + 1] = frag_masked_bits<1>(d, st.m1d >> (8 * it));
	}
	// density L1^T : dL/d(enc) = W1d^T * d_h1d  (32 features = one row tile), stored level-major as T1 stores it: lane (sample, hi) owns levels 2 rr + hi
	f16v denc = zero16();
#pragma unroll
	for (int s = 0; s < 4; ++s) denc = mfma(lds_frag(bw, BW_D1 + s, lane), dh1d[s], denc);
	if (valid) {
#pragma unroll
		for (int rr = 0; rr < 4; ++rr) {
			const uint2 g = {cvt_pk(denc[4 * rr + 0], denc[4 * rr + 1]), cvt_pk(denc[4 * rr + 2], denc[4 * rr + 3])};
			denc_lv[(size_t)(2 * rr + hi) * denc_cap + s_raw] = g;
		}
	}
}

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_train_fused(ModelPtrs mp, const float* __restrict__ in, uint32_t in_stride, uint32_t n, const __half* __restrict__ dL_dy, uint32_t dy_stride,
		EncStashIn stash_in, uint2* __restrict__ denc_lv, uint32_t denc_cap, float* __restrict__ partials) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	h8* fw = (h8*)smem;
	h8* bw = fw + N_FW_FRAGS * 64;
	load_frags_to_lds(fw, mp.fw_frags, N_FW_FRAGS);
	load_frags_to_lds(bw, mp.bw_frags, N_BW_FRAGS);
	const int lane = threadIdx.x & 63, col = lane & 31, hi = lane >> 5, wid = threadIdx.x >> 6, role = wid >> 2, w4 = wid & 3;
	h8* idf = bw + N_BW_FRAGS * 64; // the two identity fragments (transposes through the MFMA) live in LDS like the weights: 8 registers less per lane
	if (wid < 2) idf[wid * 64 + lane] = ident_frag(wid, lane);
	__syncthreads();
	const uint32_t wave = blockIdx.x * 4 + w4, n_waves = gridDim.x * 4; // the two roles walk the same tiles
	f16v dW[FUSED_NT_B]; // role A: d1 (0,1), d2 (2,3) = tiles 0..3 of the partials; role B: r2 (0..3), r3 (4,5), r1 (6,7) = tiles 6..9, 10..11, 4..5
#pragma unroll
	for (int t = 0; t < FUSED_NT_B; ++t) dW[t] = zero16();
	const uint32_t n_valid_rows = min(*stash_in.n_valid_ptr, n);
	for (uint32_t ct = wave; (uint64_t)ct * 32 < n; ct += n_waves) {
		const uint32_t s_raw = ct * 32 + col;
		const bool valid = s_raw < n;
		uint32_t row = min(s_raw, n - 1);
		const float* p = in + (size_t)row * in_stride;
		if (row >= n_valid_rows) row = n_valid_rows ? row % n_valid_rows : 0u; // K4's wrap-around padding: row e is a copy of row e % n_valid
		const uint32_t src = n_valid_rows ? stash_in.src_index[row] : 0u;
		const uint4* e = stash_in.enc + (size_t)src * 4 + (uint32_t)hi * 2;
		if (role == 0) fused_tile<0>(fw, bw, idf, lane, hi, valid, p, e, dL_dy, dy_stride, s_raw, dW, denc_lv, denc_cap);
		else fused_tile<1>(fw, bw, idf, lane, hi, valid, p, e, dL_dy, dy_stride, s_raw, dW, denc_lv, denc_cap);
	}
	// reduce the 4 waves of each role through LDS (re-using the fragment region: <= 8 tiles * 16 regs * 64 lanes * 4 B = 32 KiB), one role at a time, in a fixed order
	float* red = (float*)smem;
	float* dstp = partials + (size_t)blockIdx.x * (N_DW_TILES * 16 * 64);
	for (int r = 0; r < 2; ++r) {
		const int nt = r == 0 ? FUSED_NT_A : FUSED_NT_B;
		__syncthreads();
		for (int w = 0; w < 4; ++w) {
			if (role == r && w4 == w) {
#pragma unroll
				for (int t = 0; t < FUSED_NT_B; ++t) {
					if (t >= nt) break;
#pragma unroll
					for (int q = 0; q < 16; ++q) {
						float* dst = red + ((size_t)t * 16 + q) * 64 + lane;
						*dst = (w == 0 ? 0.f : *dst) + dW[t][q];
					}
				}
			}
			__syncthreads();
		}
		// partial tile order (k_wgrad_reduce): d1 (0,1), d2 (2,3), r1 (4,5), r2 (6..9), r3 (10,11)
		for (int i = threadIdx.x; i < nt * 16 * 64; i += blockDim.x) {
			const int t = i / (16 * 64), dst_t = r == 0 ? t : (t < 6 ? 6 + t : t - 2);
			dstp[dst_t * 16 * 64 + (i % (16 * 64))] = red[i];
		}
	}
}

// sum the per-block partials, un-permute the D tiles into row-major [out][in] and round to half.
// One block per 64 consecutive elements (= one register row of a tile): 4 waves each sum a quarter of the partials with
// coalesced 256-byte reads, then combine through LDS in a fixed order (deterministic).
// 16 wavefronts per 64 elements (each sums every 16th partial, 4 independent loads in flight), then a fixed-order tree over the 16 sums:
// the 12.6 MB of partials stream at memory speed instead of as 64 dependent loads per wavefront (17.7 -> see DESIGN 8).
__global__ void __launch_bounds__(1024) k_wgrad_reduce(const float* __restrict__ partials, uint32_t n_partials, __half* __restrict__ mlp_grad, uint32_t nr /* hidden colour layers */, uint32_t ex /* 1: model with extra dims (first colour layer 64 x 48, two more tiles behind the regular ones) */) {
	__shared__ float sm[16][64];
	const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
	const uint32_t e = blockIdx.x * 64 + lane; // element of [tile][r][lane]
	float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
	const uint32_t n_tiles = 8 + 4 * (nr - 1);
	const size_t PS = (size_t)(n_tiles + 2 * ex) * 16 * 64;
	uint32_t g = wid;
	for (; g + 48 < n_partials; g += 64) {
		s0 += partials[(size_t)g * PS + e]; s1 += partials[(size_t)(g + 16) * PS + e];
		s2 += partials[(size_t)(g + 32) * PS + e]; s3 += partials[(size_t)(g + 48) * PS + e];
	}
	for (; g < n_partials; g += 16) s0 += partials[(size_t)g * PS + e];
	sm[wid][lane] = (s0 + s1) + (s2 + s3);
	__syncthreads();
	if (wid != 0) return;
	float s;
	{
		float t[16];
#pragma unroll
		for (int k = 0; k < 16; ++k) t[k] = sm[k][lane];
#pragma unroll
		for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
			for (int k = 0; k < w; ++k) t[k] = t[k] + t[k + w];
		s = t[0];
	}
	const int t = e / (16 * 64), r = (e / 64) % 16;
	int layer_off, R, C, it, kt;
	const int c1 = 32 + 16 * (int)ex, xo = 64 * 16 * (int)ex; // width of the first colour layer; what it adds to the offsets of the layers behind it
	if (t < 2) { layer_off = 0; R = 64; C = 32; it = t; kt = 0; }
	else if (t < 4) { layer_off = 2048; R = 16; C = 64; it = 0; kt = t - 2; }
	else if (t < 6) { layer_off = 3072; R = 64; C = c1; it = t - 4; kt = 0; }
	else if (t < (int)n_tiles - 2) { const int k = (t - 6) >> 2, u = (t - 6) & 3; layer_off = 5120 + xo + 4096 * k; R = 64; C = 64; it = u >> 1; kt = u & 1; }
	else if (t < (int)n_tiles) { layer_off = 5120 + xo + 4096 * ((int)nr - 1); R = 16; C = 64; it = 0; kt = t - ((int)n_tiles - 2); }
	else { layer_off = 3072; R = 64; C = c1; it = t - (int)n_tiles; kt = 1; }
	const int i = it * 32 + (r & 3) + 8 * (r >> 2) + 4 * ((int)lane >> 5);
	const int k = kt * 32 + ((int)lane & 31);
	if (i < R && k < C) mlp_grad[layer_off + i * C + k] = __float2half(s);
}

// ---------------------------------------------------------------------------------------------
// Training step of the image / SDF primitives' model (Trainer::training_step, reference call sites testbed_image.cu:289,
// testbed_sdf.cu:1557): forward with ReLU masks, loss + loss gradient [tcnn losses/l2.h, mape.h, relative_l2.h] (or an external
// dL/dy), dgrad chain, GridEncoding backward with atomicAdd(__half2) into the gradient table -- one launch; the encoding fragments
// and dL/dy are stashed for the weight-gradient kernel.  Lane (n, hi) owns levels 4q + 2hi + b (q = 0..3, b = 0,1), both features.
// ---------------------------------------------------------------------------------------------
template <int D, bool EXTERNAL_DY>
__global__ void __launch_bounds__(256, 3) k_encmlp_train_fwd_bwd(EncTrainArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	h8* fw = (h8*)smem;
	h8* bw = fw + N_FW_FRAGS * 64;
	__shared__ uint4 s_lct[16];
	fill_level_table_nd<D>(s_lct, a.gm);
	load_frags_to_lds(fw, a.fw_frags, N_FW_FRAGS);
	load_frags_to_lds(bw, a.bw_frags, N_BW_FRAGS);
	__syncthreads();
	const int lane = threadIdx.x & 63, col = lane & 31, hi = lane >> 5;
	const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
	const uint32_t n = a.n;
	const float n_total = (float)n * (float)a.n_out;
	float loss_acc = 0.f;
	for (uint32_t tile = wave; (uint64_t)tile * 32 < n; tile += n_waves) {
		const uint32_t s_raw = tile * 32 + col;
		const bool sv = s_raw < n;
		const float* p = a.in + (size_t)min(s_raw, n - 1) * a.in_stride;
		float x[D];
#pragma unroll
		for (int d = 0; d < D; ++d) x[d] = p[d];
		FwdState<1> st;
#pragma unroll
		for (int s = 0; s < 2; ++s) {
			h8 e;
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const h2 f = level_features2_lds<D>(a.table, s_lct, (uint32_t)(8 * s + 4 * (q >> 1) + 2 * hi + (q & 1)), x);
				e[2 * q] = f[0]; e[2 * q + 1] = f[1];
			}
			st.rin[0][s] = e;
			a.enc_stash[((size_t)tile * 2 + s) * 64 + lane] = __builtin_bit_cast(uint4, e);
		}
		fwd_rgb_l1<1>(fw, lane, st);
		fwd_rgb_l2<1>(fw, lane, st);
		f16v o[1];
		fwd_rgb_l3<1>(fw, lane, st, o);
		// ---- loss and dL/dy: outputs 0..3 live in registers 0..3 of the hi == 0 lanes ----
		h8 dy0 = zero8();
		if (hi == 0 && sv) {
			h4 dy4 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
			if (EXTERNAL_DY) {
				const _Float16* g = (const _Float16*)a.dy_in + (size_t)s_raw * a.dy_stride;
				for (uint32_t k = 0; k < a.n_out; ++k) dy4[k] = g[k];
			} else {
				const float* t = a.target + (size_t)s_raw * a.target_stride;
				for (uint32_t k = 0; k < a.n_out; ++k) {
					const float pr = (float)(_Float16)o[0][k], tg = t[k], diff = pr - tg; // the prediction is the network's half output
					float value, grad;
					if (a.loss_type == NGP_LOSS_MAPE) { const float sc = 1.0f / (fabsf(tg) + 0.01f); value = fabsf(diff) * sc / n_total; grad = (diff > 0.f ? 1.f : diff < 0.f ? -1.f : 0.f) * sc / n_total; }
					else if (a.loss_type == NGP_LOSS_RELATIVE_L2) { const float sc = 1.0f / (pr * pr + 0.01f); value = diff * diff * sc / n_total; grad = 2 * diff * sc / n_total; }
					else if (a.loss_type == NGP_LOSS_L1) { value = fabsf(diff) / n_total; grad = (diff > 0.f ? 1.f : diff < 0.f ? -1.f : 0.f) / n_total; }
					else { value = diff * diff / n_total; grad = 2 * diff / n_total; }
					loss_acc += value;
					dy4[k] = (_Float16)(a.loss_scale * grad);
				}
				if (a.pred_out) for (uint32_t k = 0; k < a.n_out; ++k) ((_Float16*)a.pred_out)[(size_t)s_raw * a.pred_stride + k] = (_Float16)o[0][k];
			}
			dy0[0] = dy4[0]; dy0[1] = dy4[1]; dy0[2] = dy4[2]; dy0[3] = dy4[3];
			a.dy_stash[s_raw] = __builtin_bit_cast(uint2, dy4);
		}
		// ---- dgrad chain ----
		h8 dh[4];
#pragma unroll
		for (int mt = 0; mt < 2; ++mt) {
			const h8 aa = lds_frag(bw, BW_R3 + mt, lane);
			const f16v d = mfma(aa, dy0, zero16());
			dh[2 * mt + 0] = to_frag_masked(d, 0, st.m2r[0][0] >> (16 * mt));
			dh[2 * mt + 1] = to_frag_masked(d, 1, st.m2r[0][0] >> (16 * mt));
		}
		h8 dh1[4];
#pragma unroll
		for (int mt = 0; mt < 2; ++mt) {
			f16v acc = zero16();
#pragma unroll
			for (int s = 0; s < 4; ++s) acc = mfma(lds_frag(bw, BW_R2 + mt * 4 + s, lane), dh[s], acc);
			dh1[2 * mt + 0] = to_frag_masked(acc, 0, st.m1r[0] >> (16 * mt));
			dh1[2 * mt + 1] = to_frag_masked(acc, 1, st.m1r[0] >> (16 * mt));
		}
		f16v denc = zero16();
#pragma unroll
		for (int s = 0; s < 4; ++s) denc = mfma(lds_frag(bw, BW_R1 + s, lane), dh1[s], denc);
		// ---- GridEncoding backward: register pair (2m, 2m+1) = features 0,1 of level (m & 1) + 4 (m >> 1) + 2 hi ----
		if (a.denc_lv && sv) {
			// the scatter runs through the record lists (k_grad_bin / k_grad_accumulate behind this kernel): dL/d(enc) leaves level-major, one half2 per (level, sample)
#pragma unroll
			for (int m = 0; m < 8; ++m) {
				const uint32_t level = (uint32_t)((m & 1) + 4 * (m >> 1) + 2 * hi);
				const h2 g = {(_Float16)denc[2 * m], (_Float16)denc[2 * m + 1]};
				a.denc_lv[(size_t)level * a.denc_cap + s_raw] = __builtin_bit_cast(uint32_t, g);
			}
		} else if (a.grid_grad && sv) {
#pragma unroll
			for (int m = 0; m < 8; ++m) {
				const uint32_t level = (uint32_t)((m & 1) + 4 * (m >> 1) + 2 * hi);
				const float g0 = (float)(_Float16)denc[2 * m], g1 = (float)(_Float16)denc[2 * m + 1]; // dL/d(enc) is a half matrix in the reference
				CornersND<D> cr;
				level_corners_nd<D>(a.gm, level, x, cr);
				__half* gt = a.grid_grad + (size_t)a.gm->offset[level] * 2;
#pragma unroll
				for (int c = 0; c < (1 << D); ++c) {
					const h2 v = {(_Float16)(g0 * cr.w[c]), (_Float16)(g1 * cr.w[c])};
					atomic_add_h2(gt + (size_t)cr.idx[c] * 2, v);
				}
				__builtin_amdgcn_sched_barrier(0);
			}
		}
	}
	if (!EXTERNAL_DY && a.loss_sum) {
#pragma unroll
		for (int dd = 32; dd >= 1; dd >>= 1) loss_acc += __shfl_xor(loss_acc, dd, 64);
		if (lane == 0 && loss_acc != 0.f) atomicAdd(a.loss_sum, loss_acc);
	}
}

// Weight gradients of the 32 -> 64 -> 64 -> 16 MLP: the colour-network part of k_wgrad (same swapped-operand scheme), 8 dW tiles
// r1:(0,1) r2:(2..5) r3:(6,7) = 128 accumulator registers per wave.
constexpr int N_EDW_TILES = 8;
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
k_encmlp_wgrad(const ngp_half* __restrict__ fw_frags, const ngp_half* __restrict__ bw_frags, uint32_t n, const uint2* __restrict__ dy_stash,
		const uint4* __restrict__ enc_stash, float* __restrict__ partials) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	h8* fw = (h8*)smem;
	h8* bw = fw + N_FW_FRAGS * 64;
	load_frags_to_lds(fw, fw_frags, N_FW_FRAGS);
	load_frags_to_lds(bw, bw_frags, N_BW_FRAGS);
	__syncthreads();
	const int lane = threadIdx.x & 63, col = lane & 31, hi = lane >> 5, wid = threadIdx.x >> 6;
	const uint32_t wave = blockIdx.x * 4 + wid, n_waves = gridDim.x * 4;
	f16v dW[N_EDW_TILES];
#pragma unroll
	for (int t = 0; t < N_EDW_TILES; ++t) dW[t] = zero16();
	const h8 I0 = ident_frag(0, lane), I1 = ident_frag(1, lane);
	for (uint32_t ct = wave; (uint64_t)ct * 32 < n; ct += n_waves) {
		const uint32_t s_raw = ct * 32 + col;
		FwdState<1> st;
		st.rin[0][0] = __builtin_bit_cast(h8, enc_stash[((size_t)ct * 2 + 0) * 64 + lane]);
		st.rin[0][1] = __builtin_bit_cast(h8, enc_stash[((size_t)ct * 2 + 1) * 64 + lane]);
		h8 dy0 = zero8();
		if (hi == 0 && s_raw < n) { // out-of-range columns contribute nothing: their output gradient is zero
			const h4 g = __builtin_bit_cast(h4, dy_stash[s_raw]);
			dy0[0] = g[0]; dy0[1] = g[1]; dy0[2] = g[2]; dy0[3] = g[3];
		}
		fwd_rgb_l1<1>(fw, lane, st);
		h8 h2_sw[2][2];
#pragma unroll
		for (int kt = 0; kt < 2; ++kt) {
			f16v t = zero16();
#pragma unroll
			for (int s = 0; s < 4; ++s) t = mfma(st.hb[0][s], lds_frag(fw, FW_R2 + kt * 4 + s, lane), t);
			sw_to_frags(t, true, h2_sw[kt]);
		}
		fwd_rgb_l2<1>(fw, lane, st); // only the ReLU mask m2r is consumed below
		h8 g_sw[2];
		{ f16v t = mfma(dy0, I0, zero16()); sw_to_frags(t, false, g_sw); }
#pragma unroll
		for (int kt = 0; kt < 2; ++kt)
#pragma unroll
			for (int q = 0; q < 2; ++q) dW[6 + kt] = mfma(g_sw[q], h2_sw[kt][q], dW[6 + kt]);
		h8 dh[4];
		h8 d2_sw[2][2];
#pragma unroll
		for (int mt = 0; mt < 2; ++mt) {
			const h8 aa = lds_frag(bw, BW_R3 + mt, lane);
			f16v d = mfma(aa, dy0, zero16());
			dh[2 * mt + 0] = to_frag_masked(d, 0, st.m2r[0][0] >> (16 * mt));
			dh[2 * mt + 1] = to_frag_masked(d, 1, st.m2r[0][0] >> (16 * mt));
			f16v dsw = mfma(dy0, aa, zero16());
			sw_grad_to_frags(dsw, h2_sw[mt], d2_sw[mt]);
		}
		h8 h1_sw[2][2];
#pragma unroll
		for (int kt = 0; kt < 2; ++kt) {
			f16v t = zero16();
#pragma unroll
			for (int s = 0; s < 2; ++s) t = mfma(st.rin[0][s], lds_frag(fw, FW_R1 + kt * 2 + s, lane), t);
			sw_to_frags(t, true, h1_sw[kt]);
		}
#pragma unroll
		for (int it = 0; it < 2; ++it)
#pragma unroll
			for (int kt = 0; kt < 2; ++kt)
#pragma unroll
				for (int q = 0; q < 2; ++q) dW[2 + it * 2 + kt] = mfma(d2_sw[it][q], h1_sw[kt][q], dW[2 + it * 2 + kt]);
		h8 d1_sw[2][2];
#pragma unroll
		for (int mt = 0; mt < 2; ++mt) {
			f16v dsw = zero16();
#pragma unroll
			for (int s = 0; s < 4; ++s) dsw = mfma(dh[s], lds_frag(bw, BW_R2 + mt * 4 + s, lane), dsw);
			sw_grad_to_frags(dsw, h1_sw[mt], d1_sw[mt]);
		}
		h8 rin_sw[2];
		{ f16v t = mfma(st.rin[0][0], I0, zero16()); t = mfma(st.rin[0][1], I1, t); sw_to_frags(t, false, rin_sw); }
#pragma unroll
		for (int it = 0; it < 2; ++it)
#pragma unroll
			for (int q = 0; q < 2; ++q) dW[0 + it] = mfma(d1_sw[it][q], rin_sw[q], dW[0 + it]);
	}
	// reduce the 4 waves of the block through LDS (re-using the fragment region), 4 tiles at a time
	float* red = (float*)smem; // 4 tiles * 16 regs * 64 lanes * 4 B = 16 KiB
	float* dstp = partials + (size_t)blockIdx.x * (N_EDW_TILES * 16 * 64);
#pragma unroll
	for (int half = 0; half < 2; ++half) {
		__syncthreads();
		for (int w = 0; w < 4; ++w) {
			if (wid == w) {
#pragma unroll
				for (int t = 0; t < 4; ++t)
#pragma unroll
					for (int r = 0; r < 16; ++r) {
						float* dst = red + ((size_t)t * 16 + r) * 64 + lane;
						*dst = (w == 0 ? 0.f : *dst) + dW[half * 4 + t][r];
					}
			}
			__syncthreads();
		}
		for (int i = threadIdx.x; i < 4 * 16 * 64; i += blockDim.x) dstp[half * 4 * 16 * 64 + i] = red[i];
	}
}
// (16 wavefronts per 64 elements, each summing every 16th partial, then a fixed-order tree -- the shape of k_wgrad_reduce; round 6: four wavefronts walked 64 dependent loads each, 17 us)
__global__ void __launch_bounds__(1024) k_encmlp_wgrad_reduce(const float* __restrict__ partials, uint32_t n_partials, __half* __restrict__ mlp_grad) {
	__shared__ float sm[16][64];
	const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
	const uint32_t e = blockIdx.x * 64 + lane; // element of [tile][r][lane]
	float s = 0.f;
	for (uint32_t g = wid; g < n_partials; g += 16) s += partials[(size_t)g * (N_EDW_TILES * 16 * 64) + e];
	sm[wid][lane] = s;
	__syncthreads();
	if (wid != 0) return;
	s = (((sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane])) + ((sm[4][lane] + sm[5][lane]) + (sm[6][lane] + sm[7][lane])))
	  + (((sm[8][lane] + sm[9][lane]) + (sm[10][lane] + sm[11][lane])) + ((sm[12][lane] + sm[13][lane]) + (sm[14][lane] + sm[15][lane])));
	const int t = e / (16 * 64), r = (e / 64) % 16;
	int layer_off, R, C, it, kt;
	if (t < 2) { layer_off = 0; R = 64; C = 32; it = t; kt = 0; }
	else if (t < 6) { layer_off = 2048; R = 64; C = 64; it = (t - 2) >> 1; kt = (t - 2) & 1; }
	else { layer_off = 6144; R = 16; C = 64; it = 0; kt = t - 6; }
	const int i = it * 32 + (r & 3) + 8 * (r >> 2) + 4 * ((int)lane >> 5);
	const int k = kt * 32 + ((int)lane & 31);
	if (i < R && k < C) mlp_grad[layer_off + i * C + k] = __float2half(s);
}

// ---------------------------------------------------------------------------------------------
// fragment build (after set_params) and the fused Adam + ExponentialDecay + EMA sweep
// ---------------------------------------------------------------------------------------------
__global__ void k_build_frags(const __half* __restrict__ mlp_params, uint32_t n_mlp, const uint32_t* __restrict__ fw_perm, const uint32_t* __restrict__ bw_perm,
		__half* __restrict__ fw, __half* __restrict__ bw) {
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n_mlp) return;
	const __half v = mlp_params[p];
	if (fw) fw[fw_perm[p]] = v;
	if (bw) bw[bw_perm[p]] = v;
}

// Sweeps parameters [a.range_begin, a.n_params) (both multiples of 4).  a.ema_only: the Adam step of this range ran elsewhere and the new half parameters are in place
// (sharded data-parallel step: another rank owns the range and its parameters arrived by all-gather) -- only the EMA / inference copy is advanced and the consumed
// gradients are cleared.
__global__ void __launch_bounds__(256) k_optimizer(AdamArgs a) {
	const uint64_t i4 = a.range_begin / 4 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t i = i4 * 4;
	if (i >= a.n_params) return;
	const bool matrix = i < a.n_mlp; // n_mlp is a multiple of 4
	const h4 g4 = __builtin_bit_cast(h4, ((const uint2*)a.grads)[i4]);
	if (a.zero_grid_grads && !matrix) ((uint2*)a.grads)[i4] = make_uint2(0u, 0u); // consumed: the next step's scatter starts from zero
	h4 w4 = __builtin_bit_cast(h4, ((const uint2*)a.params)[i4]);
	float g[4]; bool upd[4]; bool any = false;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		g[k] = (float)g4[k] / a.loss_scale;
		upd[k] = matrix ? (a.optimize_matrix != 0) : (a.optimize_non_matrix != 0 && g[k] != 0.f);
		any |= upd[k];
	}
	if (a.ema_only) any = false;
	float4 mw = make_float4(0.f, 0.f, 0.f, 0.f);
	float* mwp = (float*)&mw;
	if (any || (a.ema_full_precision && a.ema_decay != 0.f)) mw = ((const float4*)a.master)[i4];
	if (any) {
		float4 m4 = ((const float4*)a.m)[i4], v4 = ((const float4*)a.v)[i4];
		uint2 st = ((const uint2*)a.steps)[i4];
		float* mp = (float*)&m4; float* vp = (float*)&v4; uint16_t* sp = (uint16_t*)&st;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			if (!upd[k]) continue;
			const float nw = adam_update(a, matrix, g[k], mwp[k], mp[k], vp[k], sp[k]);
			mwp[k] = nw;
			w4[k] = (_Float16)nw;
			if (matrix) {
				((_Float16*)a.fw_frags)[a.fw_perm[i + k]] = w4[k];
				((_Float16*)a.bw_frags)[a.bw_perm[i + k]] = w4[k];
			}
		}
		((float4*)a.master)[i4] = mw; ((float4*)a.m)[i4] = m4; ((float4*)a.v)[i4] = v4; ((uint2*)a.steps)[i4] = st;
		((uint2*)a.params)[i4] = __builtin_bit_cast(uint2, w4);
	}
	h4 inf4;
	if (a.ema_decay == 0.f) inf4 = w4; // no Ema wrapper (image / SDF configs): the inference parameters are the parameters
	else if (a.ema_full_precision) { // ema_step_full_precision: fp32 state, fp32 master weights (sharded data parallelism keeps the all-reduce step in this mode: foreign masters are stale)
		float4 e4 = ((const float4*)a.ema)[i4];
		float* ep = (float*)&e4;
#pragma unroll
		for (int k = 0; k < 4; ++k) { const float filtered = ema_update(a, ep[k], mwp[k]); ep[k] = filtered; inf4[k] = (_Float16)filtered; }
		((float4*)a.ema)[i4] = e4;
	} else { // ema_step_half_precision: the state is the inference buffer itself
		const h4 old4 = __builtin_bit_cast(h4, ((const uint2*)a.params_inf)[i4]);
#pragma unroll
		for (int k = 0; k < 4; ++k) inf4[k] = (_Float16)ema_update(a, (float)old4[k], (float)w4[k]);
	}
	if (matrix) {
#pragma unroll
		for (int k = 0; k < 4; ++k) ((_Float16*)a.fw_frags_inf)[a.fw_perm[i + k]] = inf4[k];
	}
	((uint2*)a.params_inf)[i4] = __builtin_bit_cast(uint2, inf4);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static int g_num_cus = 0;
static int num_cus() {
	if (!g_num_cus) {
		hipDeviceProp_t prop;
		int dev = 0;
		(void)hipGetDevice(&dev);
		if (hipGetDeviceProperties(&prop, dev) == hipSuccess) g_num_cus = prop.multiProcessorCount;
		if (g_num_cus <= 0) g_num_cus = 256;
	}
	return g_num_cus;
}

void launch_inference(hipStream_t s, const GridMeta* gm, const ModelPtrs& mp, const float* in, uint32_t in_stride, uint32_t n_max, const uint32_t* n_ptr,
		ngp_half* out, uint32_t out_stride, bool density_only, uint32_t dir_offset, uint32_t F) {
	if (n_max == 0) return;
	const uint32_t tiles = (n_max + 31) / 32;
	const uint32_t grid = (uint32_t)std::min<uint64_t>((tiles + 3) / 4, (uint64_t)num_cus() * 4);
	const bool pair = (g_debug_flags & DBG_FWD_PAIR_LOADS) != 0; // measured slower than plain per-lane gathers (profiles/r01_microbench_ablation4_gather.log)
	const bool occ4 = (g_debug_flags & DBG_FWD_OCC4) != 0; // 4 waves/SIMD (128 VGPRs, spills) instead of 3 (168 VGPRs)
	const uint32_t nr = mp.n_rgb_hidden;
#define NGP_LAUNCH_INF(D, P, W, LDS, FF, NRR) hipLaunchKernelGGL((k_inference<D, 1, P, W, FF, NRR>), dim3(grid), dim3(256), LDS, s, gm, mp, in, in_stride, n_max, n_ptr, (__half*)out, out_stride, dir_offset)
	if (density_only) { // the density network is the same for every colour-network depth
		if (F == 2) NGP_LAUNCH_INF(true, false, 3, 8 * 1024, 2, 2);
		else if (pair) { if (occ4) NGP_LAUNCH_INF(true, true, 4, 8 * 1024, 4, 2); else NGP_LAUNCH_INF(true, true, 3, 8 * 1024, 4, 2); }
		else { if (occ4) NGP_LAUNCH_INF(true, false, 4, 8 * 1024, 4, 2); else NGP_LAUNCH_INF(true, false, 3, 8 * 1024, 4, 2); }
	} else if (mp.n_extra) { // extra dims: L = 8, F = 4, two hidden colour layers (ngp_model_create)
		hipLaunchKernelGGL((k_inference<false, 1, false, 3, 4, 2, 1>), dim3(grid), dim3(256), (n_fw(2) + N_FW_EXTRA) * 1024, s, gm, mp, in, in_stride, n_max, n_ptr, (__half*)out, out_stride, dir_offset);
	} else if (F == 2 || nr != 2) { // L = 16, F = 2 and / or 1 or 3 hidden colour layers (no ablation variants)
		if (F == 2) { if (nr == 1) NGP_LAUNCH_INF(false, false, 3, n_fw(1) * 1024, 2, 1); else if (nr == 3) NGP_LAUNCH_INF(false, false, 3, n_fw(3) * 1024, 2, 3); else NGP_LAUNCH_INF(false, false, 3, n_fw(2) * 1024, 2, 2); }
		else { if (nr == 1) NGP_LAUNCH_INF(false, false, 3, n_fw(1) * 1024, 4, 1); else NGP_LAUNCH_INF(false, false, 3, n_fw(3) * 1024, 4, 3); }
	} else {
		if (pair) { if (occ4) NGP_LAUNCH_INF(false, true, 4, N_FW_FRAGS * 1024, 4, 2); else NGP_LAUNCH_INF(false, true, 3, N_FW_FRAGS * 1024, 4, 2); }
		else { if (occ4) NGP_LAUNCH_INF(false, false, 4, N_FW_FRAGS * 1024, 4, 2); else NGP_LAUNCH_INF(false, false, 3, N_FW_FRAGS * 1024, 4, 2); }
	}
#undef NGP_LAUNCH_INF
}
void launch_inference_lazy(hipStream_t s, const GridMeta* gm, const ModelPtrs& mp, const float* in, uint32_t in_stride, uint32_t max_rays, uint32_t max_samples,
		ngp_half* out, uint32_t out_stride, uint32_t dir_offset, const K2LazyArgs& la_in, uint32_t F) {
	if (max_rays == 0) return;
	K2LazyArgs la = la_in;
	const uint32_t tpw = 32u / la.tile_w;
	// Resident blocks only (3 per CU, 137 registers).  Measured and rejected in round 3 (profiles/r03_microbench_bin_threads_k2_grid.log): 2 - 16 x more blocks than
	// resident slots (dispatcher hands tiles out as blocks retire) 0.13 -> 0.16 - 0.21 ms; 4 blocks per CU at 128 registers (28 B of scratch) 0.130 -> 0.134 ms; 8-wide
	// tiles 0.131 -> 0.142 ms.  The kernel moves 400 MB of 128-byte lines for 8-byte table entries (profiles/r03_pmc_summary.txt): it runs at the memory side's
	// random-line rate (3.3 TB/s), not at a latency or occupancy limit.
	constexpr uint32_t k2_bpc = 3u; // resident workgroups per CU (fewer, to leave wave slots to a concurrent kernel, bought nothing: profiles/r04_exp_dummy_k1_under_k2.log)
	const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)la.tile_cap / tpw + 3) / 4, (uint64_t)num_cus() * k2_bpc);
	const uint32_t nr = mp.n_rgb_hidden;
#define NGP_LAUNCH_TILES(TW, FF, NRR) hipLaunchKernelGGL((k_inference_tiles<TW, FF, NRR>), dim3(grid), dim3(256), n_fw(NRR) * 1024, s, gm, mp, in, in_stride, la, (__half*)out, out_stride, dir_offset)
#define NGP_LAUNCH_TILES_W(FF, NRR) do { if (la.tile_w == 8) NGP_LAUNCH_TILES(8, FF, NRR); else if (la.tile_w == 16) NGP_LAUNCH_TILES(16, FF, NRR); else NGP_LAUNCH_TILES(32, FF, NRR); } while (0)
	static const int k2_depth = getenv("NGP_K2_DEPTH") ? atoi(getenv("NGP_K2_DEPTH")) : 2; // 0: level constants from GridMeta (rounds 1-5, ablation); 1 / 2: from the LDS table (profiles/r06_ab_k2_level_table.txt)
	for (uint32_t r = 0; r < la.n_rounds; ++r) {
		la.round = r;
		static const bool k2_dyn = !(getenv("NGP_K2_STATIC") && atoi(getenv("NGP_K2_STATIC")) != 0) && la.n_rounds == 1;
		if (k2_depth == 2 && F == 4 && nr == 2 && la.tile_w == 16 && k2_dyn) { hipLaunchKernelGGL((k_inference_tiles<16, 4, 2, 2, true>), dim3(grid), dim3(256), n_fw(2) * 1024, s, gm, mp, in, in_stride, la, (__half*)out, out_stride, dir_offset); continue; }
		if (k2_depth == 2 && F == 4 && nr == 2 && la.tile_w == 16) { hipLaunchKernelGGL((k_inference_tiles<16, 4, 2, 2>), dim3(grid), dim3(256), n_fw(2) * 1024, s, gm, mp, in, in_stride, la, (__half*)out, out_stride, dir_offset); continue; }
		if (k2_depth == 1 && F == 4 && nr == 2 && la.tile_w == 16) { hipLaunchKernelGGL((k_inference_tiles<16, 4, 2, 1>), dim3(grid), dim3(256), n_fw(2) * 1024, s, gm, mp, in, in_stride, la, (__half*)out, out_stride, dir_offset); continue; }
		if (F == 2) { if (nr == 1) NGP_LAUNCH_TILES_W(2, 1); else if (nr == 3) NGP_LAUNCH_TILES_W(2, 3); else NGP_LAUNCH_TILES_W(2, 2); }
		else { if (nr == 1) NGP_LAUNCH_TILES_W(4, 1); else if (nr == 3) NGP_LAUNCH_TILES_W(4, 3); else NGP_LAUNCH_TILES_W(4, 2); }
	}
#undef NGP_LAUNCH_TILES_W
#undef NGP_LAUNCH_TILES
	(void)max_samples;
}
void launch_encmlp_inference(hipStream_t s, const GridMeta* gm, uint32_t n_pos_dims, const ngp_half* grid, const ngp_half* fw_frags, const float* in, uint32_t in_stride,
		uint32_t n, ngp_half* out, uint32_t out_stride, uint32_t n_out) {
	if (n == 0) return;
	const uint32_t tiles = (n + 31) / 32;
	const uint32_t grid_dim = (uint32_t)std::min<uint64_t>((tiles + 3) / 4, (uint64_t)num_cus() * 4);
	if (n_pos_dims == 2)
		hipLaunchKernelGGL((k_encmlp_inference<2>), dim3(grid_dim), dim3(256), N_FW_FRAGS * 1024, s, gm, (const __half*)grid, fw_frags, in, in_stride, n, (__half*)out, out_stride, n_out);
	else
		hipLaunchKernelGGL((k_encmlp_inference<3>), dim3(grid_dim), dim3(256), N_FW_FRAGS * 1024, s, gm, (const __half*)grid, fw_frags, in, in_stride, n, (__half*)out, out_stride, n_out);
}
void launch_encmlp_train(hipStream_t s, const EncTrainArgs& a, uint32_t n_pos_dims, bool external_dy, float* wgrad_partials, uint32_t n_partials, ngp_half* mlp_grad) {
	if (a.n == 0) return;
	const uint32_t tiles = (a.n + 31) / 32;
	const uint32_t grid_dim = (uint32_t)std::min<uint64_t>((tiles + 3) / 4, (uint64_t)num_cus() * 3);
	const uint32_t lds = (N_FW_FRAGS + N_BW_FRAGS) * 1024;
	if (n_pos_dims == 2) {
		if (external_dy) hipLaunchKernelGGL((k_encmlp_train_fwd_bwd<2, true>), dim3(grid_dim), dim3(256), lds, s, a);
		else hipLaunchKernelGGL((k_encmlp_train_fwd_bwd<2, false>), dim3(grid_dim), dim3(256), lds, s, a);
	} else {
		if (external_dy) hipLaunchKernelGGL((k_encmlp_train_fwd_bwd<3, true>), dim3(grid_dim), dim3(256), lds, s, a);
		else hipLaunchKernelGGL((k_encmlp_train_fwd_bwd<3, false>), dim3(grid_dim), dim3(256), lds, s, a);
	}
	hipLaunchKernelGGL(k_encmlp_wgrad, dim3(n_partials), dim3(256), lds, s, a.fw_frags, a.bw_frags, a.n, (const uint2*)a.dy_stash, (const uint4*)a.enc_stash, wgrad_partials);
	hipLaunchKernelGGL(k_encmlp_wgrad_reduce, dim3(N_EDW_TILES * 16), dim3(1024), 0, s, wgrad_partials, n_partials, (__half*)mlp_grad);
}
void launch_encode_only(hipStream_t s, const GridMeta* gm, const ngp_half* grid, const float* pos, uint32_t stride, uint32_t n, ngp_half* out, uint32_t F) {
	if (n == 0) return;
	const uint32_t waves = (n + 31) / 32;
	if (F == 2) hipLaunchKernelGGL(k_encode_only<2>, dim3((waves + 3) / 4), dim3(256), 0, s, gm, (const __half*)grid, pos, stride, n, (__half*)out);
	else hipLaunchKernelGGL(k_encode_only<4>, dim3((waves + 3) / 4), dim3(256), 0, s, gm, (const __half*)grid, pos, stride, n, (__half*)out);
}
void launch_build_frags(hipStream_t s, const ngp_half* mlp_params, uint32_t n_mlp, const uint32_t* fw_perm, const uint32_t* bw_perm, ngp_half* fw, ngp_half* bw) {
	hipLaunchKernelGGL(k_build_frags, dim3((n_mlp + 255) / 256), dim3(256), 0, s, (const __half*)mlp_params, n_mlp, fw_perm, bw_perm, (__half*)fw, (__half*)bw);
}
uint32_t wgrad_n_partials() { return (uint32_t)num_cus(); }
#define REQUIRE_VOID(c) do { if (!(c)) { fprintf(stderr, "launch_grad_bin: unsupported layout (%s)\n", #c); abort(); } } while (0)
// what: bit 0 = k_grad_bin (every listed level), bit 1 = k_grad_accumulate over the listed levels [a.acc_ly_begin, a.acc_ly_begin + a.acc_ly_count) (count 0 = all of them).
// The sharded data-parallel step accumulates in two launches so that the first bucket's reduce-scatter runs beside the second launch.
void launch_grad_bin(hipStream_t s, const GradBinArgs& a_in, uint32_t what) {
	if (a_in.n == 0 || a_in.n_hashed == 0) return;
	GradBinArgs a = a_in;
	if (a.acc_ly_count == 0) { a.acc_ly_begin = 0; a.acc_ly_count = a.n_hashed; }
	const bool do_bin = (what & 1u) != 0, do_acc = (what & 2u) != 0;
	const uint32_t ny = a.acc_ly_count;
	constexpr uint32_t ns = 512u; // samples per k_grad_bin workgroup (256 / 1024 measured in rounds 2-3: profiles/r03_microbench_bin_threads_k2_grid.log)
	const dim3 gb((a.n + ns - 1) / ns, a.n_hashed);
	if (a.n_features == 2 && a.n_pos_dims == 2) { // the image primitive's grid (encmlp trainer): 4 corners per sample
		REQUIRE_VOID(a.chunk_log2 == 12);
		if (do_bin) hipLaunchKernelGGL((k_grad_bin<12, 512, 2, 256, 2>), dim3((a.n + 511) / 512, a.n_hashed), dim3(256), 0, s, a);
		if (do_acc) hipLaunchKernelGGL((k_grad_accumulate<12, false, 2, false, 256>), dim3(a.max_chunks, ny, 1), dim3(256), 0, s, a);
		return;
	}
	if (a.n_features == 2) { // L = 16, F = 2: one block per chunk, both features (4-byte record values)
		if (a.chunk_log2 == 11) {
			if (do_bin) hipLaunchKernelGGL((k_grad_bin<11, 512, 2>), dim3((a.n + 511) / 512, a.n_hashed), dim3(256), 0, s, a);
			if (do_acc) hipLaunchKernelGGL((k_grad_accumulate<11, false, 2>), dim3(a.max_chunks, ny, 1), dim3(1024), 0, s, a);
		} else {
			if (do_bin) hipLaunchKernelGGL((k_grad_bin<12, 512, 2>), dim3((a.n + 511) / 512, a.n_hashed), dim3(256), 0, s, a);
			if (do_acc) hipLaunchKernelGGL((k_grad_accumulate<12, false, 2>), dim3(a.max_chunks, ny, 1), dim3(1024), 0, s, a);
		}
		return;
	}
	if (a.chunk_log2 == 11) {
		if (do_bin) {
			static const bool bin256 = getenv("NGP_BIN_THREADS_256") && atoi(getenv("NGP_BIN_THREADS_256")) != 0; // ablation: two samples per thread (what 2048-entry chunks ran until this commit)
			if (bin256) hipLaunchKernelGGL((k_grad_bin<11, 512>), gb, dim3(256), 0, s, a);
			else hipLaunchKernelGGL((k_grad_bin<11, 512, 4, 512>), gb, dim3(512), 0, s, a); // one sample per thread, as with 4096-entry chunks (round 3: 80 -> 66 us)
		}
		if (do_acc) {
			if (a.fuse_adam && !a.split) hipLaunchKernelGGL((k_grad_accumulate<11, false, 4, true>), dim3(a.max_chunks, ny, 1), dim3(1024), 0, s, a);
			else if (a.split) hipLaunchKernelGGL((k_grad_accumulate<11, true>), dim3(a.max_chunks, ny, 2), dim3(1024), 0, s, a);
			else hipLaunchKernelGGL((k_grad_accumulate<11, false>), dim3(a.max_chunks, ny, 1), dim3(1024), 0, s, a);
		}
	} else {
		constexpr uint32_t bin_threads = 512u; // one sample per thread (256 threads x 2 samples, the round-2 shape: 80 -> 66 us, profiles/r03_microbench_bin_threads_k2_grid.log)
		if (do_bin) {
			if (ns == 256) hipLaunchKernelGGL((k_grad_bin<12, 256>), gb, dim3(256), 0, s, a);
			// (1024 samples / threads per block -- twice the run length, half the cursor atomics, but one 100 KiB block per CU: unit 0.150 -> 0.164 ms, rejected)
			else if (bin_threads == 512) hipLaunchKernelGGL((k_grad_bin<12, 512, 4, 512>), gb, dim3(512), 0, s, a);
			else hipLaunchKernelGGL((k_grad_bin<12, 512>), gb, dim3(256), 0, s, a);
		}
		if (do_acc) {
			if (a.fuse_adam && !a.split) hipLaunchKernelGGL((k_grad_accumulate<12, false, 4, true>), dim3(a.max_chunks, ny, 1), dim3(1024), 0, s, a);
			else if (a.split) hipLaunchKernelGGL((k_grad_accumulate<12, true>), dim3(a.max_chunks, ny, 2), dim3(1024), 0, s, a);
			else hipLaunchKernelGGL((k_grad_accumulate<12, false>), dim3(a.max_chunks, ny, 1), dim3(1024), 0, s, a);
		}
	}
}
void launch_train_fwd_bwd(hipStream_t s, const GridMeta* gm, const ModelPtrs& mp, const float* in, uint32_t in_stride, uint32_t n,
		const ngp_half* dL_dy, uint32_t dy_stride, ngp_half* grid_grad, ngp_half* enc_stash, uint32_t flags, void* denc_lv, uint32_t denc_cap, uint32_t F, const EncStashIn* stash_in, float* dextra_out) {
	if (n == 0) return;
	const uint32_t tiles = (n + 31) / 32;
	const uint32_t grid = (uint32_t)std::min<uint64_t>((tiles + 3) / 4, (uint64_t)num_cus() * 3);
	const uint32_t nr = mp.n_rgb_hidden;
	if (mp.n_extra) { // extra dims: L = 8, F = 4, two hidden colour layers; every level's dL/d(enc) through the lists (the caller provides them) or, without lists, atomics
		const uint32_t lds = (uint32_t)(n_fw(2) + N_FW_EXTRA + n_bw(2) + N_BW_EXTRA) * 1024;
		if ((flags & T1_DENSE_EXTERNAL) && denc_lv) hipLaunchKernelGGL((k_train_fwd_bwd<1, 3, false, 4, 2, false, 1>), dim3(grid), dim3(256), lds, s, gm, mp, in, in_stride, n,
			(const __half*)dL_dy, dy_stride, (__half*)grid_grad, (uint4*)enc_stash, flags, (uint2*)denc_lv, denc_cap, EncStashIn(), dextra_out);
		else hipLaunchKernelGGL((k_train_fwd_bwd<1, 3, true, 4, 2, false, 1>), dim3(grid), dim3(256), lds, s, gm, mp, in, in_stride, n,
			(const __half*)dL_dy, dy_stride, (__half*)grid_grad, (uint4*)enc_stash, flags, (uint2*)denc_lv, denc_cap, EncStashIn(), dextra_out);
		return;
	}
	if (F == 2 || nr != 2) { // L = 16, F = 2 and / or 1 or 3 hidden colour layers: every level through the lists (no atomics in T1) or, without lists, atomics
		const bool no_scatter = (flags & T1_DENSE_EXTERNAL) && denc_lv;
#define NGP_LAUNCH_T1(SC, FF, NRR) hipLaunchKernelGGL((k_train_fwd_bwd<1, 3, SC, FF, NRR>), dim3(grid), dim3(256), (n_fw(NRR) + n_bw(NRR)) * 1024, s, gm, mp, in, in_stride, n, \
			(const __half*)dL_dy, dy_stride, (__half*)grid_grad, (uint4*)enc_stash, flags, (uint2*)denc_lv, denc_cap)
#define NGP_LAUNCH_T1_SC(FF, NRR) do { if (no_scatter) NGP_LAUNCH_T1(false, FF, NRR); else NGP_LAUNCH_T1(true, FF, NRR); } while (0)
		if (F == 2) { if (nr == 1) NGP_LAUNCH_T1_SC(2, 1); else if (nr == 3) NGP_LAUNCH_T1_SC(2, 3); else NGP_LAUNCH_T1_SC(2, 2); }
		else { if (nr == 1) NGP_LAUNCH_T1_SC(4, 1); else NGP_LAUNCH_T1_SC(4, 3); }
#undef NGP_LAUNCH_T1_SC
#undef NGP_LAUNCH_T1
		return;
	}
	// 3 wavefronts per SIMD without spills (164 registers) beat 4 with 36 spilled registers: T1 + bin + accumulate 0.219 vs 0.244 ms (profiles/r02_t1_occupancy.txt)
	constexpr int t1_occ = 3; // wavefronts per SIMD of T1 (4 spills: profiles/r02_t1_occupancy.txt)
	// (4 blocks per CU -- the stash variant compiles to 91 registers when asked -- measured no better: unit 0.189 vs 0.180 ms, profiles/r03_microbench_t1_stash.log)
	if ((flags & T1_DENSE_EXTERNAL) && denc_lv && !(flags & DBG_T1_OCC2) && t1_occ == 3 && stash_in && stash_in->enc && !(flags & DBG_T1_NO_K2_STASH))
		hipLaunchKernelGGL((k_train_fwd_bwd<1, 3, false, 4, 2, true>), dim3(grid), dim3(256), (N_FW_FRAGS + N_BW_FRAGS) * 1024, s, gm, mp, in, in_stride, n,
			(const __half*)dL_dy, dy_stride, (__half*)grid_grad, (uint4*)enc_stash, flags, (uint2*)denc_lv, denc_cap, *stash_in);
	else if ((flags & T1_DENSE_EXTERNAL) && denc_lv && !(flags & DBG_T1_OCC2) && t1_occ == 3)
		hipLaunchKernelGGL((k_train_fwd_bwd<1, 3, false>), dim3(grid), dim3(256), (N_FW_FRAGS + N_BW_FRAGS) * 1024, s, gm, mp, in, in_stride, n,
			(const __half*)dL_dy, dy_stride, (__half*)grid_grad, (uint4*)enc_stash, flags, (uint2*)denc_lv, denc_cap);
	else if ((flags & T1_DENSE_EXTERNAL) && denc_lv && !(flags & DBG_T1_OCC2)) // no atomics in T1: every level's dL/d(enc) goes to denc_lv
		hipLaunchKernelGGL((k_train_fwd_bwd<1, 4, false>), dim3(std::min<uint32_t>((tiles + 3) / 4, (uint32_t)num_cus() * 4)), dim3(256), (N_FW_FRAGS + N_BW_FRAGS) * 1024, s, gm, mp, in, in_stride, n,
			(const __half*)dL_dy, dy_stride, (__half*)grid_grad, (uint4*)enc_stash, flags, (uint2*)denc_lv, denc_cap);
	else if (flags & DBG_T1_OCC2)
		hipLaunchKernelGGL((k_train_fwd_bwd<1, 2, true>), dim3(grid), dim3(256), (N_FW_FRAGS + N_BW_FRAGS) * 1024, s, gm, mp, in, in_stride, n, (const __half*)dL_dy, dy_stride,
			(__half*)grid_grad, (uint4*)enc_stash, flags, (uint2*)denc_lv, denc_cap);
	else
		hipLaunchKernelGGL((k_train_fwd_bwd<1, 3, true>), dim3(grid), dim3(256), (N_FW_FRAGS + N_BW_FRAGS) * 1024, s, gm, mp, in, in_stride, n, (const __half*)dL_dy, dy_stride,
			(__half*)grid_grad, (uint4*)enc_stash, flags, (uint2*)denc_lv, denc_cap);
}
void launch_grad_dense(hipStream_t s, const GradDenseArgs& a) {
	if (a.n == 0 || a.n_levels == 0) return;
	hipLaunchKernelGGL(k_grad_dense, dim3((a.n + 255) / 256, a.n_levels), dim3(256), 0, s, a);
}
void launch_wgrad(hipStream_t s, const ModelPtrs& mp, const float* in, uint32_t in_stride, uint32_t n, const ngp_half* dL_dy, uint32_t dy_stride,
		const ngp_half* enc_stash, float* wgrad_partials, uint32_t n_partials) {
	if (n == 0) return;
	const uint32_t nr = mp.n_rgb_hidden, lds = (uint32_t)(n_fw((int)nr) + n_bw((int)nr)) * 1024;
#define NGP_LAUNCH_W(NRR) hipLaunchKernelGGL(k_wgrad_nr<NRR>, dim3(n_partials), dim3(256), lds, s, mp, in, in_stride, n, (const __half*)dL_dy, dy_stride, (const uint4*)enc_stash, wgrad_partials)
	if (mp.n_extra) hipLaunchKernelGGL((k_wgrad_nr<2, 1>), dim3(n_partials), dim3(256), lds + N_FW_EXTRA * 1024, s, mp, in, in_stride, n, (const __half*)dL_dy, dy_stride, (const uint4*)enc_stash, wgrad_partials);
	else if (nr == 1) NGP_LAUNCH_W(1);
	else if (nr == 3) NGP_LAUNCH_W(3);
	else if (g_debug_flags & DBG_W_SINGLE_ROLE) NGP_LAUNCH_W(2);
	else hipLaunchKernelGGL(k_wgrad2, dim3(n_partials), dim3(512), lds, s, mp, in, in_stride, n, (const __half*)dL_dy, dy_stride, (const uint4*)enc_stash, wgrad_partials);
#undef NGP_LAUNCH_W
}
// T1 + W fused (base.json's shape, encodings from the lazy K2's stash): dL/d(enc) of every level -> denc_lv (level-major), weight-gradient partials -> wgrad_partials
void launch_train_fused(hipStream_t s, const ModelPtrs& mp, const float* in, uint32_t in_stride, uint32_t n, const ngp_half* dL_dy, uint32_t dy_stride,
		const EncStashIn& stash_in, void* denc_lv, uint32_t denc_cap, float* wgrad_partials, uint32_t n_partials) {
	if (n == 0) return;
	hipLaunchKernelGGL(k_train_fused, dim3(n_partials), dim3(512), (N_FW_FRAGS + N_BW_FRAGS + 2) * 1024, s, mp, in, in_stride, n, (const __half*)dL_dy, dy_stride, stash_in,
		(uint2*)denc_lv, denc_cap, wgrad_partials);
}
void launch_wgrad_reduce(hipStream_t s, const float* partials, uint32_t n_partials, ngp_half* mlp_grad, uint32_t n_rgb_hidden, uint32_t n_extra) {
	const uint32_t ex = n_extra ? 1u : 0u;
	hipLaunchKernelGGL(k_wgrad_reduce, dim3((n_dw_tiles((int)n_rgb_hidden) + 2 * ex) * 16), dim3(1024), 0, s, partials, n_partials, (__half*)mlp_grad, n_rgb_hidden, ex);
}
void launch_optimizer_step(hipStream_t s, const AdamArgs& a) {
	if (a.n_params <= a.range_begin) return;
	hipLaunchKernelGGL(k_optimizer, dim3((uint32_t)(((a.n_params - a.range_begin) / 4 + 255) / 256)), dim3(256), 0, s, a);
}

} // namespace ngp
