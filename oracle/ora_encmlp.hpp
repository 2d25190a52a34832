// CPU oracle (TEST INFRASTRUCTURE ONLY, parity unpinned -- see ora_math.hpp): the image / SDF primitives' model, a HashGrid
// encoding of a 2-D or 3-D position followed by one FullyFusedMLP (tcnn::NetworkWithInputEncoding; reference call sites
// testbed.cu:4354-4363, testbed_image.cu:383,541, testbed_sdf.cu:1658).  Forward path only.
// [tcnn] GridEncodingTemplated<T, N_POS_DIMS, N_FEATURES_PER_LEVEL> restated for general D (grid.h kernel_grid, grid_index,
// hash primes), NetworkWithInputEncoding parameter order = network, then encoding.
#pragma once
#include "ora_model.hpp"

namespace ora {

struct GridLayoutND {
	uint32_t D = 3, n_levels = 0, F = 0;
	std::vector<uint32_t> offsets; std::vector<float> scales; std::vector<uint32_t> resolutions;
	uint32_t n_entries() const { return offsets.back(); }
	void build(const ngp_encmlp_config& c) {
		D = c.n_pos_dims; n_levels = c.n_levels; F = c.n_features_per_level;
		offsets.assign(n_levels + 1, 0); scales.resize(n_levels); resolutions.resize(n_levels);
		const float l2 = std::log2(c.per_level_scale);
		uint32_t offset = 0;
		for (uint32_t i = 0; i < n_levels; ++i) {
			scales[i] = grid_scale(i, l2, c.base_resolution);
			const uint32_t res = resolutions[i] = grid_resolution(scales[i]);
			const uint32_t max_params = std::numeric_limits<uint32_t>::max() / 2;
			uint32_t params_in_level = std::pow((float)res, (float)D) > (float)max_params ? max_params : (uint32_t)std::pow((float)res, (float)D);
			params_in_level = next_multiple(params_in_level, 8u);
			params_in_level = std::min(params_in_level, 1u << c.log2_hashmap_size);
			offsets[i] = offset; offset += params_in_level;
		}
		offsets[n_levels] = offset;
	}
};

// [tcnn grid.h] grid_index<N_POS_DIMS, Hash> with the coherent prime hash
inline uint32_t grid_index_nd(uint32_t D, uint32_t hashmap_size, uint32_t res, const uint32_t* pg) {
	static const uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
	uint32_t stride = 1, index = 0;
	for (uint32_t dim = 0; dim < D && stride <= hashmap_size; ++dim) { index += pg[dim] * stride; stride *= res; }
	if (hashmap_size < stride) { index = 0; for (uint32_t dim = 0; dim < D; ++dim) index ^= pg[dim] * primes[dim]; }
	return index % hashmap_size;
}

inline void grid_encode_nd(const GridLayoutND& g, const uint16_t* table, const float* pos_in, uint16_t* out /* L*F */) {
	const uint32_t D = g.D, NC = 1u << D;
	for (uint32_t l = 0; l < g.n_levels; ++l) {
		const float scale = g.scales[l];
		const uint32_t res = g.resolutions[l], hashmap_size = g.offsets[l + 1] - g.offsets[l];
		float pos[4]; uint32_t pg[4];
		for (uint32_t d = 0; d < D; ++d) { const float p = std::fma(scale, pos_in[d], 0.5f), tmp = std::floor(p); pg[d] = (uint32_t)(int)tmp; pos[d] = p - tmp; }
		uint32_t idx[16]; float w[16];
		for (uint32_t c = 0; c < NC; ++c) {
			float weight = 1; uint32_t pl[4];
			for (uint32_t d = 0; d < D; ++d) {
				if ((c & (1u << d)) == 0) { weight *= 1 - pos[d]; pl[d] = pg[d]; } else { weight *= pos[d]; pl[d] = pg[d] + 1; }
			}
			idx[c] = grid_index_nd(D, hashmap_size, res, pl); w[c] = weight;
		}
		const uint16_t* lvl = table + (size_t)g.offsets[l] * g.F;
		for (uint32_t f = 0; f < g.F; ++f) {
			uint16_t r = 0;
			for (uint32_t c = 0; c < NC; ++c) r = hfma(f2h(w[c]), lvl[(size_t)idx[c] * g.F + f], r);
			out[l * g.F + f] = r;
		}
	}
}

struct EncMlp {
	ngp_encmlp_config cfg; GridLayoutND grid; MlpShape net;
	size_t n_mlp = 0, n_params = 0;
	std::vector<float> params_fp; std::vector<uint16_t> params;
	EncMlp(const ngp_encmlp_config& c, uint64_t seed) : cfg(c) {
		if (c.n_pos_dims < 1 || c.n_pos_dims > 4) throw std::runtime_error("oracle: n_pos_dims must be 1..4");
		if ((c.n_levels * c.n_features_per_level) % 16) throw std::runtime_error("oracle: L*F must be a multiple of 16");
		grid.build(c);
		net = {c.n_levels * c.n_features_per_level, c.n_neurons, c.n_hidden_layers, 16};
		n_mlp = net.n_params(); n_params = n_mlp + (size_t)grid.n_entries() * grid.F;
		params_fp.resize(n_params); params.resize(n_params);
		// same conventions as Model::initialize: Xavier-uniform matrices, then U(-1e-4, 1e-4) for the grid, one pcg32 stream [tcnn]
		Pcg32 rnd(seed);
		size_t p = 0;
		for (uint32_t l = 0; l < net.n_layers(); ++l) {
			const uint32_t R = net.rows(l), C = net.cols(l);
			const float scale = std::sqrt(6.0f / (float)(R + C));
			for (uint32_t i = 0; i < R * C; ++i) params_fp[p++] = rnd.next_float() * 2.0f * scale - scale;
		}
		for (; p < n_params; ++p) params_fp[p] = rnd.next_float() * (1e-4f - (-1e-4f)) + (-1e-4f);
		sync_half();
	}
	void sync_half() { for (size_t i = 0; i < n_params; ++i) params[i] = f2h(params_fp[i]); }
	std::vector<uint16_t> grads; // Trainer::gradients, written by training_step (GradientMode::Overwrite)

	// Trainer::training_step with an external dL/dy (n x 16 halfs, only the first n_output_dims used): forward with saved activations,
	// FullyFusedMLP backward (fp32 weight-gradient accumulation, half dL/dx), GridEncoding backward (half atomicAdd per contribution,
	// here in sample order) -- the same scheme as Model::training_step, for general D / F.  [tcnn]
	void training_step(const float* in, uint32_t stride, uint32_t n, const uint16_t* dL_dy, uint32_t dy_stride) {
		const uint32_t n_enc = net.in, D = grid.D, NC = 1u << D;
		std::vector<float> dW(n_mlp, 0.f);
		std::vector<uint16_t> dL_denc((size_t)n * n_enc);
		#pragma omp parallel
		{
			std::vector<float> dWt(n_mlp, 0.f);
			std::vector<uint16_t> enc(n_enc), acts(net.n_hidden * net.width);
			#pragma omp for schedule(static)
			for (int64_t i = 0; i < (int64_t)n; ++i) {
				uint16_t o[16], dout[16] = {0};
				grid_encode_nd(grid, params.data() + n_mlp, in + (size_t)i * stride, enc.data());
				mlp_forward(net, params.data(), enc.data(), acts.data(), o);
				for (uint32_t k = 0; k < cfg.n_output_dims; ++k) dout[k] = dL_dy[(size_t)i * dy_stride + k];
				mlp_backward(net, params.data(), enc.data(), acts.data(), dout, dWt.data(), dL_denc.data() + (size_t)i * n_enc);
			}
			#pragma omp critical
			for (size_t k = 0; k < n_mlp; ++k) dW[k] += dWt[k];
		}
		grads.assign(n_params, 0);
		for (size_t k = 0; k < n_mlp; ++k) grads[k] = f2h(dW[k]);
		uint16_t* gg = grads.data() + n_mlp;
		#pragma omp parallel for schedule(dynamic, 1)
		for (int64_t l = 0; l < (int64_t)grid.n_levels; ++l) {
			const float scale = grid.scales[l];
			const uint32_t res = grid.resolutions[l], hashmap_size = grid.offsets[l + 1] - grid.offsets[l];
			uint16_t* lvl = gg + (size_t)grid.offsets[l] * grid.F;
			for (uint32_t i = 0; i < n; ++i) {
				const float* x = in + (size_t)i * stride;
				float pos[4]; uint32_t pg[4];
				for (uint32_t d = 0; d < D; ++d) { const float p = std::fma(scale, x[d], 0.5f), tmp = std::floor(p); pg[d] = (uint32_t)(int)tmp; pos[d] = p - tmp; }
				for (uint32_t c = 0; c < NC; ++c) {
					float w = 1; uint32_t pl[4];
					for (uint32_t d = 0; d < D; ++d) { if ((c & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; } else { w *= pos[d]; pl[d] = pg[d] + 1; } }
					const uint32_t idx = grid_index_nd(D, hashmap_size, res, pl);
					for (uint32_t f = 0; f < grid.F; ++f) {
						const uint16_t v = f2h(h2f(dL_denc[(size_t)i * n_enc + l * grid.F + f]) * w);
						uint16_t& dst = lvl[(size_t)idx * grid.F + f];
						dst = f2h(h2f(dst) + h2f(v));
					}
				}
			}
		}
	}

	// [tcnn losses/l2.h, mape.h] per-element loss value and loss_scale-d gradient in half; n_total = n * n_output_dims.
	// L2 (configs/image/base.json): (p - t)^2 / n_total, gradient 2 (p - t) / n_total.  MAPE (configs/sdf/base.json):
	// |p - t| / (|t| + 0.01) / n_total, gradient sign(p - t) / (|t| + 0.01) / n_total.
	float loss_and_gradient(bool mape, const uint16_t* pred, uint32_t pred_stride, const float* target, uint32_t target_stride, uint32_t n, float loss_scale,
			uint16_t* dL_dy /* n x 16 halfs */) const {
		const uint32_t no = cfg.n_output_dims;
		const float n_total = (float)n * (float)no;
		double total = 0;
		for (uint32_t i = 0; i < n; ++i) {
			for (uint32_t k = 0; k < 16; ++k) dL_dy[(size_t)i * 16 + k] = 0;
			for (uint32_t k = 0; k < no; ++k) {
				const float p = h2f(pred[(size_t)i * pred_stride + k]), t = target[(size_t)i * target_stride + k], diff = p - t;
				float value, grad;
				if (mape) { const float scale = 1.0f / (std::fabs(t) + 0.01f); value = std::fabs(diff) * scale / n_total; grad = (diff > 0 ? 1.f : diff < 0 ? -1.f : 0.f) * scale / n_total; }
				else { value = diff * diff / n_total; grad = 2 * diff / n_total; }
				total += value;
				dL_dy[(size_t)i * 16 + k] = f2h(loss_scale * grad);
			}
		}
		return (float)total;
	}

	void inference(const float* in, uint32_t stride, uint32_t n, uint16_t* out, uint32_t out_stride) const {
		#pragma omp parallel for schedule(static)
		for (int64_t i = 0; i < (int64_t)n; ++i) {
			std::vector<uint16_t> enc(net.in), acts(net.n_hidden * net.width);
			uint16_t o[16];
			grid_encode_nd(grid, params.data() + n_mlp, in + (size_t)i * stride, enc.data());
			mlp_forward(net, params.data(), enc.data(), acts.data(), o);
			for (uint32_t k = 0; k < cfg.n_output_dims; ++k) out[(size_t)i * out_stride + k] = o[k];
		}
	}
};

} // namespace ora
