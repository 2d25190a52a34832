// ORACLE -- TEST INFRASTRUCTURE ONLY (see ora_math.hpp header).  PARITY UNPINNED: everything in this file restates tiny-cuda-nn, whose source is absent from the mount.
//
// ora_model.hpp: CPU restatement of the tiny-cuda-nn objects the reference's NeRF path consumes:
// GridEncoding (hash grid) fwd/bwd, FullyFusedMLP fwd/bwd, SphericalHarmonics, the NerfNetwork
// composition (nerf_network.h:81-280, 357-390), and Trainer + Adam / ExponentialDecay / Ema.
// Everything marked [tcnn] restates github.com/NVlabs/tiny-cuda-nn (absent from /root/reference);
// rounding points follow the reference's mixed precision: fp16 tables / weights / activations,
// fp32 accumulation inside a matrix product, fp16 gradient table.
#pragma once
#include "ora_math.hpp"
#include <vector>
#include <stdexcept>
#include <cstdio>

namespace ora {

// [tcnn encodings/grid.h] grid_scale / grid_resolution
inline float grid_scale(uint32_t level, float log2_per_level_scale, uint32_t base_resolution) {
	return std::exp2(level * log2_per_level_scale) * base_resolution - 1.0f;
}
inline uint32_t grid_resolution(float scale) { return (uint32_t)std::ceil(scale) + 1; }
inline uint32_t next_multiple(uint32_t v, uint32_t d) { return ((v + d - 1) / d) * d; }

struct GridLayout {
	uint32_t n_levels = 0, F = 0;
	std::vector<uint32_t> offsets;     // n_levels+1, in entries
	std::vector<float> scales;         // per level
	std::vector<uint32_t> resolutions; // per level
	uint32_t n_entries() const { return offsets.back(); }

	// [tcnn GridEncodingTemplated ctor] offset table for GridType::Hash
	void build(const ngp_model_config& c) {
		n_levels = c.n_levels; F = c.n_features_per_level;
		offsets.assign(n_levels + 1, 0); scales.resize(n_levels); resolutions.resize(n_levels);
		float l2 = std::log2(c.per_level_scale);
		uint32_t offset = 0;
		for (uint32_t i = 0; i < n_levels; ++i) {
			scales[i] = grid_scale(i, l2, c.base_resolution);
			uint32_t res = resolutions[i] = grid_resolution(scales[i]);
			uint32_t max_params = std::numeric_limits<uint32_t>::max() / 2;
			uint32_t params_in_level = std::pow((float)res, 3.0f) > (float)max_params ? max_params : res * res * res;
			params_in_level = next_multiple(params_in_level, 8u);
			params_in_level = std::min(params_in_level, 1u << c.log2_hashmap_size);
			offsets[i] = offset;
			offset += params_in_level;
		}
		offsets[n_levels] = offset;
	}
};

// [tcnn grid.h] grid_index<3, CoherentPrime>
inline uint32_t grid_index(uint32_t hashmap_size, uint32_t res, const uint32_t pg[3]) {
	uint32_t stride = 1, index = 0;
	for (uint32_t dim = 0; dim < 3 && stride <= hashmap_size; ++dim) {
		index += pg[dim] * stride;
		stride *= res;
	}
	if (hashmap_size < stride) {
		index = (pg[0] * 1u) ^ (pg[1] * 2654435761u) ^ (pg[2] * 805459861u);
	}
	return index % hashmap_size;
}

// [tcnn grid.h kernel_grid] one level of one sample; linear interpolation; half fma accumulation.
// pos in [0,1]^3. out: F halfs.  Also returns the 8 corner indices / weights for the backward pass.
inline void grid_level_lookup(const GridLayout& g, uint32_t level, const float pos_in[3], uint32_t idx_out[8], float w_out[8]) {
	const float scale = g.scales[level];
	const uint32_t res = g.resolutions[level];
	const uint32_t hashmap_size = g.offsets[level + 1] - g.offsets[level];
	float pos[3]; uint32_t pg[3];
	for (int d = 0; d < 3; ++d) {
		float p = std::fma(scale, pos_in[d], 0.5f);
		float tmp = std::floor(p);
		pg[d] = (uint32_t)(int)tmp;
		pos[d] = p - tmp;
	}
	for (uint32_t c = 0; c < 8; ++c) {
		float weight = 1;
		uint32_t pl[3];
		for (uint32_t d = 0; d < 3; ++d) {
			if ((c & (1u << d)) == 0) { weight *= 1 - pos[d]; pl[d] = pg[d]; }
			else { weight *= pos[d]; pl[d] = pg[d] + 1; }
		}
		idx_out[c] = grid_index(hashmap_size, res, pl);
		w_out[c] = weight;
	}
}

inline void grid_encode(const GridLayout& g, const uint16_t* table, const float pos[3], uint16_t* out /* L*F */) {
	for (uint32_t l = 0; l < g.n_levels; ++l) {
		uint32_t idx[8]; float w[8];
		grid_level_lookup(g, l, pos, idx, w);
		const uint16_t* lvl = table + (size_t)g.offsets[l] * g.F;
		for (uint32_t f = 0; f < g.F; ++f) {
			uint16_t r = 0;
			for (uint32_t c = 0; c < 8; ++c) r = hfma(f2h(w[c]), lvl[(size_t)idx[c] * g.F + f], r);
			out[l * g.F + f] = r;
		}
	}
}

// [tcnn encodings/spherical_harmonics.h] degree-4 real SH; input d in [0,1]^3 -> x = 2d-1.
// Constants written as their closed forms (see test_oracle_pins.py for the scipy cross-check).
inline void sh4(const float d[3], uint16_t* out16) {
	const float x = d[0] * 2.f - 1.f, y = d[1] * 2.f - 1.f, z = d[2] * 2.f - 1.f;
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
	for (int i = 0; i < 16; ++i) out16[i] = f2h(o[i]);
}

// [tcnn networks/fully_fused_mlp] ReLU hidden layers of `width`, no output activation, weights
// row-major [out x in] fp16, outputs padded to 16.
struct MlpShape {
	uint32_t in = 32, width = 64, n_hidden = 1, out = 16;
	uint32_t n_layers() const { return n_hidden + 1; }
	uint32_t rows(uint32_t l) const { return l == n_hidden ? out : width; }
	uint32_t cols(uint32_t l) const { return l == 0 ? in : width; }
	uint32_t offset(uint32_t l) const { uint32_t o = 0; for (uint32_t i = 0; i < l; ++i) o += rows(i) * cols(i); return o; }
	uint32_t n_params() const { return offset(n_layers()); }
};

// Appendix-A switch: the precision of the matrix-multiply ACCUMULATOR.
//   0 (default): fp32 accumulation over the whole contraction, one rounding to half per output -- what this repository's MFMA kernels do (v_mfma_f32_*_f16);
//   1: tcnn's FullyFusedMLP keeps its WMMA accumulator fragments in __half (wmma::fragment<accumulator, 16, 16, 16, __half>): every 16-wide k-step is one
//      mma_sync whose result is rounded to half -- modelled as acc = half(float(acc) + sum of the step's 16 products in fp32).
// tcnn's source is absent from this mount, so which one the CUDA build runs cannot be settled here; the switch puts a number on the difference
// (tools/ab_half_accumulate.py): it is a process-wide test switch of the ORACLE only (ora_set_mlp_half_accumulate / ORA_MLP_HALF_ACCUMULATE).
inline int& mlp_half_accumulate() { static int v = [] { const char* e = getenv("ORA_MLP_HALF_ACCUMULATE"); return e ? atoi(e) : 0; }(); return v; }
// dot product of `n` half pairs under the selected accumulator precision (n a multiple of 16 in every layer of the models here; a tail is one more step)
template <typename FA, typename FB>
inline float mlp_dot(uint32_t n, FA a, FB b) {
	if (!mlp_half_accumulate()) { float acc = 0.f; for (uint32_t k = 0; k < n; ++k) acc += a(k) * b(k); return acc; }
	float acc = 0.f;
	for (uint32_t k0 = 0; k0 < n; k0 += 16) {
		float part = 0.f;
		for (uint32_t k = k0; k < n && k < k0 + 16; ++k) part += a(k) * b(k);
		acc = h2f(f2h(acc + part));
	}
	return acc;
}

// forward: acts[l] (l < n_hidden) = half(relu(W_l x)), out = half(W_last h)
inline void mlp_forward(const MlpShape& s, const uint16_t* w, const uint16_t* x, uint16_t* acts /* n_hidden*width */, uint16_t* out) {
	const uint16_t* in = x;
	for (uint32_t l = 0; l < s.n_layers(); ++l) {
		const uint16_t* W = w + s.offset(l);
		uint32_t R = s.rows(l), C = s.cols(l);
		uint16_t* dst = (l == s.n_hidden) ? out : acts + l * s.width;
		for (uint32_t i = 0; i < R; ++i) {
			float acc = mlp_dot(C, [&](uint32_t k) { return h2f(W[i * C + k]); }, [&](uint32_t k) { return h2f(in[k]); });
			if (l != s.n_hidden) acc = acc > 0.f ? acc : 0.f;
			dst[i] = f2h(acc);
		}
		in = dst;
	}
}

// backward: dW (fp32 accumulators, += ), dL_dx (half, may be null)
inline void mlp_backward(const MlpShape& s, const uint16_t* w, const uint16_t* x, const uint16_t* acts, const uint16_t* dL_dout,
		float* dW, uint16_t* dL_dx) {
	std::vector<uint16_t> dcur(dL_dout, dL_dout + s.out), dnext;
	for (int l = (int)s.n_hidden; l >= 0; --l) {
		const uint16_t* W = w + s.offset(l);
		uint32_t R = s.rows(l), C = s.cols(l);
		const uint16_t* in = (l == 0) ? x : acts + (l - 1) * s.width;
		float* dWl = dW + s.offset(l);
		for (uint32_t i = 0; i < R; ++i) {
			float d = h2f(dcur[i]);
			if (d == 0.f) continue;
			for (uint32_t k = 0; k < C; ++k) dWl[i * C + k] += d * h2f(in[k]);
		}
		if (l == 0 && !dL_dx) break;
		dnext.assign(C, 0);
		for (uint32_t k = 0; k < C; ++k) {
			float acc = mlp_dot(R, [&](uint32_t i) { return h2f(W[i * C + k]); }, [&](uint32_t i) { return h2f(dcur[i]); }); // (the dgrad chain is the same fused kernel in tcnn: same accumulator type)
			if (l > 0) { // ReLU backward through the stored forward activation
				if (!(h2f(in[k]) > 0.f)) acc = 0.f;
			}
			dnext[k] = f2h(acc);
		}
		if (l == 0) { for (uint32_t k = 0; k < C; ++k) dL_dx[k] = dnext[k]; }
		dcur.swap(dnext);
	}
}

struct Model {
	ngp_model_config cfg;
	GridLayout grid;
	MlpShape density_net, rgb_net;
	uint32_t n_enc = 32;       // L*F
	uint32_t n_extra = 0, rgb_in = 32; // extra (latent / light-direction) dims of the dir encoding; width of the colour network's input
	size_t n_mlp = 0, n_params = 0, off_density = 0, off_rgb = 0, off_grid = 0;

	std::vector<float> params_fp;             // Trainer::params_full_precision
	std::vector<uint16_t> params, params_inf, grads; // params / params_inference (EMA) / gradients
	std::vector<float> adam_m, adam_v, ema_tmp;
	std::vector<uint32_t> adam_steps;
	uint32_t step = 0;                        // Adam::m_current_step
	float lr = 1e-2f;
	bool train_network = true, train_encoding = true;

	Model(const ngp_model_config& c, uint64_t seed) : cfg(c) {
		if (c.sh_degree != 4) throw std::runtime_error("oracle: only SH degree 4");
		grid.build(c);
		n_enc = c.n_levels * c.n_features_per_level;
		if (n_enc % 16) throw std::runtime_error("oracle: L*F must be a multiple of 16");
		density_net = {n_enc, c.n_neurons, c.n_hidden_layers, 16};
		// nerf_network.h:81-95: the dir encoding (Composite: SphericalHarmonics over 3 dims, Identity over the n_extra_dims others) is padded to the colour network's
		// alignment (16): 16 + n_extra -> 32 columns when n_extra > 0; the colour network's input = density output (16) + that.
		// [tcnn, restated: encodings fill their padding columns with ONES (encodings/identity.h, composite.h), so the padded columns act as a learned bias]
		n_extra = c.n_extra_dims;
		if (n_extra > 16) throw std::runtime_error("oracle: at most 16 extra dims");
		rgb_in = 16 + 16 + (n_extra ? 16u : 0u);
		rgb_net = {rgb_in, c.n_neurons, c.n_hidden_layers_rgb, 16};
		off_density = 0; off_rgb = density_net.n_params();
		n_mlp = off_rgb + rgb_net.n_params();
		off_grid = n_mlp;
		n_params = n_mlp + (size_t)grid.n_entries() * grid.F;
		params_fp.resize(n_params); params.resize(n_params); params_inf.resize(n_params); grads.assign(n_params, 0);
		adam_m.assign(n_params, 0.f); adam_v.assign(n_params, 0.f); ema_tmp.assign(n_params, 0.f); adam_steps.assign(n_params, 0);
		lr = c.learning_rate;
		initialize(seed);
	}

	// Trainer::initialize_params -> nerf_network.h:374-386 order; [tcnn] Xavier-uniform for matrices
	// (rnd.next_float()*2*s - s, s = sqrt(6/(fan_in+fan_out))), U(-1e-4,1e-4) for the grid.
	// NOTE [tcnn, unverifiable]: element j takes draw j of the pcg32{seed} stream.
	void initialize(uint64_t seed) {
		Pcg32 rnd(seed);
		size_t p = 0;
		for (const MlpShape* s : {&density_net, &rgb_net}) {
			for (uint32_t l = 0; l < s->n_layers(); ++l) {
				uint32_t R = s->rows(l), C = s->cols(l);
				float scale = std::sqrt(6.0f / (float)(R + C));
				for (uint32_t i = 0; i < R * C; ++i) params_fp[p++] = rnd.next_float() * 2.0f * scale - scale;
			}
		}
		for (; p < n_params; ++p) params_fp[p] = rnd.next_float() * (1e-4f - (-1e-4f)) + (-1e-4f);
		sync_half();
	}
	void sync_half() { for (size_t i = 0; i < n_params; ++i) params_inf[i] = params[i] = f2h(params_fp[i]); }

	const uint16_t* P(bool inference) const { return inference ? params_inf.data() : params.data(); }

	// NerfNetwork::inference_mixed_precision_impl, nerf_network.h:105-139.  coord: NerfCoordinate
	// {pos[3], dt, dir[3]}; out4: rgb logits + sigma logit (extract_density :132-138).
	void eval(const float* coord, bool inference, uint16_t out4[4], uint16_t* enc_out = nullptr, uint16_t* dact = nullptr,
			uint16_t* rgb_in_out = nullptr, uint16_t* ract = nullptr, uint16_t* rgb_out16 = nullptr) const {
		const uint16_t* p = P(inference);
		std::vector<uint16_t> enc(n_enc), da(density_net.n_hidden * density_net.width), rin(rgb_in), ra(rgb_net.n_hidden * rgb_net.width);
		uint16_t ro[16];
		grid_encode(grid, p + off_grid, coord, enc.data());
		mlp_forward(density_net, p + off_density, enc.data(), da.data(), rin.data()); // density out -> rows 0..15 of rgb input
		sh4(coord + 4, rin.data() + 16);                                             // dir enc -> rows 16..31 (nerf_network.h:122)
		for (uint32_t k = 0; k + 32 < rgb_in; ++k) rin[32 + k] = f2h(k < n_extra ? coord[7 + k] : 1.0f); // Identity over the extra dims (scale 1, offset 0), padding = 1
		mlp_forward(rgb_net, p + off_rgb, rin.data(), ra.data(), ro);
		out4[0] = ro[0]; out4[1] = ro[1]; out4[2] = ro[2]; out4[3] = rin[0];
		if (enc_out) std::copy(enc.begin(), enc.end(), enc_out);
		if (dact) std::copy(da.begin(), da.end(), dact);
		if (rgb_in_out) std::copy(rin.begin(), rin.end(), rgb_in_out);
		if (ract) std::copy(ra.begin(), ra.end(), ract);
		if (rgb_out16) std::copy(ro, ro + 16, rgb_out16);
	}

	void inference(const float* coords, uint32_t stride, uint32_t n, uint16_t* out, uint32_t out_stride, bool use_inference) const {
		#pragma omp parallel for schedule(static)
		for (int64_t i = 0; i < (int64_t)n; ++i) {
			uint16_t o[4];
			eval(coords + (size_t)i * stride, use_inference, o);
			for (int k = 0; k < 4; ++k) out[(size_t)i * out_stride + k] = o[k];
		}
	}

	// NerfNetwork::density, nerf_network.h:270-280: pos encoding + density network; out = sigma logit.
	void density(const float* pos, uint32_t stride, uint32_t n, uint16_t* out, uint32_t out_stride, bool use_inference) const {
		const uint16_t* p = P(use_inference);
		#pragma omp parallel for schedule(static)
		for (int64_t i = 0; i < (int64_t)n; ++i) {
			std::vector<uint16_t> enc(n_enc), da(density_net.n_hidden * density_net.width);
			uint16_t o[16];
			grid_encode(grid, p + off_grid, pos + (size_t)i * stride, enc.data());
			mlp_forward(density_net, p + off_density, enc.data(), da.data(), o);
			out[(size_t)i * out_stride] = o[0];
		}
	}

	// encoding only (test hook for the grid kernel)
	void encode(const float* pos, uint32_t stride, uint32_t n, uint16_t* out) const {
		#pragma omp parallel for schedule(static)
		for (int64_t i = 0; i < (int64_t)n; ++i) grid_encode(grid, params.data() + off_grid, pos + (size_t)i * stride, out + (size_t)i * n_enc);
	}

	// Trainer::training_step with external dL_dy (testbed_nerf.cu:3313-3323) =
	// NerfNetwork::forward_impl (nerf_network.h:145-187) + backward_impl (:189-268), GradientMode::Overwrite.
	// dL_dy: [0..2] -> rgb net output gradient (extract_rgb :206), [3] added to density-net output 0
	// (add_density_gradient :235).  Grid gradient: [tcnn kernel_grid_backward] half atomicAdd of
	// (half)(dL_dy_f * weight) -- accumulated here in sample order.
	// exact_grid_sums (test aid, NOT the reference's arithmetic): the SAME half contributions (half)(dL/d(enc) * weight) are summed per table entry in double and rounded
	// to half once -- the order-independent sum that the reference's chain of half atomicAdds approximates with one rounding per contribution.  The HIP path sums exactly
	// (64-bit fixed point): tests use this variant to show that its distance from the reference-order result is the reference's own accumulation noise.
	// grid_sum_mode (test aids, NOT the reference's arithmetic): 0 = the reference (chain of half adds); 1 = exact_grid_sums as described above; 2 = the UNROUNDED products
	// dL/d(enc) * weight summed in double and rounded to half once -- the gradient the half dL/d(enc) implies, with no per-contribution rounding at all (the device merges runs of
	// samples in one cell in fp32 before it rounds, DESIGN 3.1, so on the coarse levels it is closer to this than to mode 1).
	// dL_dextra (optional, n x n_extra floats): dL/d(input) of the extra dims = the Identity encoding's backward of the colour network's (half) input gradient
	// (nerf_network.h:238-252 with dL_dinput; consumed by compute_extra_dims_gradient_train_nerf, testbed_nerf.cu:1293-1330)
	void training_step(const float* coords, uint32_t stride, uint32_t n, const uint16_t* dL_dy, uint32_t dy_stride, int grid_sum_mode = 0, float* dL_dextra = nullptr) {
		const bool exact_grid_sums = grid_sum_mode != 0;
		std::vector<float> dW(n_mlp, 0.f);
		std::vector<uint16_t> dL_denc((size_t)n * n_enc);
		#pragma omp parallel
		{
			std::vector<float> dWt(n_mlp, 0.f);
			std::vector<uint16_t> enc(n_enc), da(density_net.n_hidden * density_net.width), rin(rgb_in), ra(rgb_net.n_hidden * rgb_net.width);
			#pragma omp for schedule(static)
			for (int64_t i = 0; i < (int64_t)n; ++i) {
				uint16_t o4[4];
				const float* c = coords + (size_t)i * stride;
				eval(c, false, o4, enc.data(), da.data(), rin.data(), ra.data());
				uint16_t drgb[16] = {0};
				for (int k = 0; k < 3; ++k) drgb[k] = dL_dy[(size_t)i * dy_stride + k];
				uint16_t drin[48];
				mlp_backward(rgb_net, params.data() + off_rgb, rin.data(), ra.data(), drgb, dWt.data() + off_rgb, drin);
				if (dL_dextra) for (uint32_t k = 0; k < n_extra; ++k) dL_dextra[(size_t)i * n_extra + k] = h2f(drin[32 + k]);
				// add_density_gradient: half add
				drin[0] = f2h(h2f(drin[0]) + h2f(dL_dy[(size_t)i * dy_stride + 3]));
				mlp_backward(density_net, params.data() + off_density, enc.data(), da.data(), drin, dWt.data() + off_density,
					dL_denc.data() + (size_t)i * n_enc);
			}
			#pragma omp critical
			for (size_t k = 0; k < n_mlp; ++k) dW[k] += dWt[k];
		}
		std::fill(grads.begin(), grads.end(), 0);
		for (size_t k = 0; k < n_mlp; ++k) grads[k] = f2h(dW[k]);
		uint16_t* gg = grads.data() + off_grid;
		#pragma omp parallel for schedule(dynamic, 1)
		for (int64_t l = 0; l < (int64_t)grid.n_levels; ++l) {
			uint16_t* lvl = gg + (size_t)grid.offsets[l] * grid.F;
			std::vector<double> acc;
			if (exact_grid_sums) acc.assign((size_t)(grid.offsets[l + 1] - grid.offsets[l]) * grid.F, 0.0);
			for (uint32_t i = 0; i < n; ++i) {
				uint32_t idx[8]; float w[8];
				grid_level_lookup(grid, (uint32_t)l, coords + (size_t)i * stride, idx, w);
				for (uint32_t f = 0; f < grid.F; ++f) {
					float g = h2f(dL_denc[(size_t)i * n_enc + l * grid.F + f]);
					for (uint32_t c = 0; c < 8; ++c) {
						uint16_t v = f2h(g * w[c]);
						if (exact_grid_sums) { acc[(size_t)idx[c] * grid.F + f] += grid_sum_mode == 2 ? (double)g * (double)w[c] : (double)h2f(v); continue; }
						uint16_t& dst = lvl[(size_t)idx[c] * grid.F + f];
						dst = f2h(h2f(dst) + h2f(v));
					}
				}
			}
			if (exact_grid_sums) for (size_t k = 0; k < acc.size(); ++k) lvl[k] = f2h((float)acc[k]);
		}
	}

	// Trainer::optimizer_step(loss_scale) (testbed_nerf.cu:2770): Ema( ExponentialDecay( Adam ) ).
	// [tcnn optimizers/adam.h adam_step, exponential_decay.h, ema.h]
	void optimizer_step(float loss_scale) {
		++step; // Adam::step(): ++m_current_step
		const float beta1 = cfg.beta1, beta2 = cfg.beta2, eps = cfg.epsilon, l2 = cfg.l2_reg;
		#pragma omp parallel for schedule(static)
		for (int64_t i = 0; i < (int64_t)n_params; ++i) {
			float gradient = h2f(grads[i]) / loss_scale;
			bool matrix = (size_t)i < n_mlp;
			if (!matrix) { if (!train_encoding || gradient == 0) continue; }
			else { if (!train_network) continue; }
			const float weight_fp = params_fp[i];
			if (matrix) gradient += l2 * weight_fp;
			const float gradient_sq = gradient * gradient;
			float first = adam_m[i] = beta1 * adam_m[i] + (1 - beta1) * gradient;
			const float second = adam_v[i] = beta2 * adam_v[i] + (1 - beta2) * gradient_sq;
			float learning_rate = lr;
			const uint32_t current_step = ++adam_steps[i];
			learning_rate *= std::sqrt(1 - std::pow(beta2, (float)current_step)) / (1 - std::pow(beta1, (float)current_step));
			const float effective_lr = std::fmin(std::fmax(learning_rate / (std::sqrt(second) + eps), 0.0f), std::numeric_limits<float>::max());
			const float new_weight = weight_fp - effective_lr * first;
			params_fp[i] = new_weight;
			params[i] = f2h(new_weight);
		}
		// ExponentialDecay::step
		if (cfg.decay_interval > 0 && step >= cfg.decay_start && step % cfg.decay_interval == 0) lr *= cfg.decay_base;
		// EmaOptimizer::step [tcnn optimizers/ema.h, restated from memory: Appendix-A switch `ema_full_precision` = the optimizer's "full_precision" hyperparameter, default false]:
		//   ema_step_half_precision: filtered = ((float)weights_ema[i] * decay * debias_old + (float)weights[i] * (1 - decay)) * debias_new;  weights_ema[i] = (T)filtered
		//   ema_step_full_precision: filtered = (tmp[i] * decay * debias_old + weights_full_precision[i] * (1 - decay)) * debias_new;  tmp[i] = filtered;  weights_ema[i] = (T)filtered
		// (rounds 1-5 restated a hybrid -- fp32 state, half weights -- which is neither kernel)
		const float d = cfg.ema_decay;
		const float debias_old = 1 - std::pow(d, (float)(step - 1));
		const float debias_new = 1 / (1 - std::pow(d, (float)step));
		const bool full = cfg.ema_full_precision != 0;
		#pragma omp parallel for schedule(static)
		for (int64_t i = 0; i < (int64_t)n_params; ++i) {
			if (d == 0.f) { params_inf[i] = params[i]; ema_tmp[i] = h2f(params[i]); continue; } // no Ema wrapper
			const float filtered = full ? (ema_tmp[i] * d * debias_old + params_fp[i] * (1 - d)) * debias_new
			                            : (h2f(params_inf[i]) * d * debias_old + h2f(params[i]) * (1 - d)) * debias_new;
			params_inf[i] = f2h(filtered);
			ema_tmp[i] = full ? filtered : h2f(params_inf[i]); // half mode: the buffer mirrors the state (the tests move trainer states through it)
		}
	}
};

// fill_rollover / fill_rollover_and_rescale [tcnn common_device.h], launches testbed_nerf.cu:3298-3306
inline void fill_rollover_f(uint32_t n_elements, uint32_t stride, uint32_t n_input, float* inout) {
	if (n_input == 0) return;
	for (size_t i = (size_t)n_input * stride; i < (size_t)n_elements * stride; ++i) inout[i] = inout[i % ((size_t)n_input * stride)];
}
inline void fill_rollover_and_rescale_h(uint32_t n_elements, uint32_t stride, uint32_t n_input, uint16_t* inout) {
	if (n_input == 0) return;
	// [tcnn, from memory] same guard as fill_rollover (i < n_input returns early), so only the
	// rolled-over copies are rescaled: result = (T)((float)inout[i % n_input] * n_input / n_total).
	size_t tot = (size_t)n_elements * stride, per = (size_t)n_input * stride;
	for (size_t i = per; i < tot; ++i) {
		inout[i] = f2h(h2f(inout[i % per]) * (float)per / (float)tot);
	}
}

} // namespace ora
