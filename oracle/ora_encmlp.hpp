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
	// Trainer::optimizer_step: ExponentialDecay( Adam ) of configs/image|sdf/base.json (no Ema wrapper) -- the same per-parameter
	// arithmetic as Model::optimizer_step [tcnn optimizers/adam.h, exponential_decay.h]
	ngp_optimizer_config opt{1e-2f, 0.9f, 0.99f, 1e-15f, 1e-6f, 0.f, 20000, 10000, 0.33f};
	std::vector<float> adam_m, adam_v; std::vector<uint32_t> adam_steps; uint32_t step = 0; float lr = 1e-2f;
	void optimizer_step(float loss_scale) {
		if (adam_m.empty()) { adam_m.assign(n_params, 0.f); adam_v.assign(n_params, 0.f); adam_steps.assign(n_params, 0); lr = opt.learning_rate; }
		++step;
		const float beta1 = opt.beta1, beta2 = opt.beta2, eps = opt.epsilon, l2 = opt.l2_reg;
		#pragma omp parallel for schedule(static)
		for (int64_t i = 0; i < (int64_t)n_params; ++i) {
			float gradient = h2f(grads[i]) / loss_scale;
			const bool matrix = (size_t)i < n_mlp;
			if (!matrix && gradient == 0) continue;
			const float weight_fp = params_fp[i];
			if (matrix) gradient += l2 * weight_fp;
			const float gradient_sq = gradient * gradient;
			const float first = adam_m[i] = beta1 * adam_m[i] + (1 - beta1) * gradient;
			const float second = adam_v[i] = beta2 * adam_v[i] + (1 - beta2) * gradient_sq;
			float learning_rate = lr;
			const uint32_t current_step = ++adam_steps[i];
			learning_rate *= std::sqrt(1 - std::pow(beta2, (float)current_step)) / (1 - std::pow(beta1, (float)current_step));
			const float effective_lr = std::fmin(std::fmax(learning_rate / (std::sqrt(second) + eps), 0.0f), std::numeric_limits<float>::max());
			params_fp[i] = weight_fp - effective_lr * first;
			params[i] = f2h(params_fp[i]);
		}
		if (opt.decay_interval > 0 && step >= opt.decay_start && step % opt.decay_interval == 0) lr *= opt.decay_base;
	}
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

// -------------------------------------------------------------------------------------------------
// image primitive: training batch of train_image (testbed_image.cu:231-302): uniform positions from the pcg32 stream (element e <-
// draw e [tcnn generate_random_uniform, mapping from memory]), stratify2_kernel (:66-82), eval_image_kernel_and_snap<float, 3> (:175-229)
// -------------------------------------------------------------------------------------------------
inline void image_eval_and_snap(const float* rgba, int w, int h, bool snap, bool linear_colors, float& px, float& py, float* rgb) {
	auto read_val = [&](int x, int y, float* o) {
		const float* p = rgba + ((size_t)y * w + x) * 4;
		for (int k = 0; k < 4; ++k) o[k] = p[k];
		if (!linear_colors) for (int k = 0; k < 3; ++k) o[k] = linear_to_srgb(o[k]);
	};
	const float rx = (float)w, ry = (float)h;
	float v[4];
	if (snap) {
		const int ix = (int)std::floor(px * rx), iy = (int)std::floor(py * ry);
		px = ((float)ix + 0.5f) / rx; py = ((float)iy + 0.5f) / ry;
		read_val(clampi(ix, 0, w - 1), clampi(iy, 0, h - 1), v);
	} else {
		const float fx = std::fmin(std::fmax(px * rx - 0.5f, 0.0f), rx - (1.0f + 1e-4f)), fy = std::fmin(std::fmax(py * ry - 0.5f, 0.0f), ry - (1.0f + 1e-4f));
		const int ix = (int)fx, iy = (int)fy;
		const float wx = fx - (float)ix, wy = fy - (float)iy;
		const int x0 = clampi(ix, 0, w - 2), y0 = clampi(iy, 0, h - 2);
		float v00[4], v10[4], v01[4], v11[4];
		read_val(x0, y0, v00); read_val(x0 + 1, y0, v10); read_val(x0, y0 + 1, v01); read_val(x0 + 1, y0 + 1, v11);
		for (int k = 0; k < 4; ++k) v[k] = (1 - wx) * (1 - wy) * v00[k] + (wx) * (1 - wy) * v10[k] + (1 - wx) * (wy) * v01[k] + (wx) * (wy) * v11[k];
	}
	rgb[0] = v[0]; rgb[1] = v[1]; rgb[2] = v[2];
}
inline void image_generate_batch(const float* rgba, int w, int h, uint32_t n, const Pcg32& rng_in, bool stratified, bool snap, bool linear_colors, float* positions, float* targets) {
	uint32_t l2 = 0; while ((1u << l2) < n) ++l2;
	const bool strat = stratified && (1u << l2) == n && l2 % 2 == 0;
	#pragma omp parallel for schedule(static)
	for (int64_t ii = 0; ii < (int64_t)n; ++ii) {
		const uint32_t i = (uint32_t)ii;
		Pcg32 rng = rng_in;
		rng.advance((int64_t)i * 2);
		float px = rng.next_float(), py = rng.next_float();
		if (strat) {
			const uint32_t log2_size = l2 / 2, size = 1u << log2_size, in_batch = i & ((1u << l2) - 1u);
			const uint32_t x = in_batch & (size - 1u), y = in_batch >> log2_size;
			px = px / (float)size + ((float)x / (float)size); py = py / (float)size + ((float)y / (float)size);
		}
		image_eval_and_snap(rgba, w, h, snap, linear_colors, px, py, targets + (size_t)i * 3);
		positions[(size_t)i * 2] = px; positions[(size_t)i * 2 + 1] = py;
	}
}

} // namespace ora
