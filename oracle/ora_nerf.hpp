// ORACLE -- TEST INFRASTRUCTURE ONLY (see ora_math.hpp header).  PARITY PINNED for K1, K3, the occupancy-grid kernels and the error-map CDF kernels against the
// reference's own kernels compiled for the CPU (oracle/_ref/libngpkern_ref.so, tests/test_ref_kernels.py: bit for bit), for render() against the reference's fused
// renderer (libngprender_ref.so: bit for bit) and for the Rfl / RflRelax train modes against its fused training kernel (libngptrain_ref.so: 2 fp16 ulp); UNPINNED for the
// trainer's host logic (Testbed members).
//
// ora_nerf.hpp: CPU restatement of the ngp-side NeRF kernels: camera model, K1
// generate_training_samples_nerf, K3 compute_loss_kernel_train_nerf, the occupancy-grid update chain,
// the fused per-pixel renderer, and the Testbed::train driver (prep cadence, counters, rng).
#pragma once
#include "ora_model.hpp"

namespace ora {

// common_device.cuh:778-793
inline void image_pos(vec2 pos, const int32_t res[2], int& px, int& py) {
	px = clampi((int)(pos.x * (float)res[0]), 0, res[0] - 1);
	py = clampi((int)(pos.y * (float)res[1]), 0, res[1] - 1);
}

// read_rgba, common_device.cuh:846-872 (host pointer version of `pixels`)
inline vec4 read_rgba(vec2 uv, const int32_t res[2], const void* pixels, int type) {
	int px, py; image_pos(uv, res, px, py);
	size_t idx = (size_t)px + (size_t)py * res[0];
	switch (type) {
		case NGP_IMAGE_BYTE: {
			uint32_t val = ((const uint32_t*)pixels)[idx];
			if (val == 0x00FF00FFu) return {-1.f, -1.f, -1.f, -1.f};
			vec4 r = {((val & 0x000000FFu) >> 0) * (1.0f / 255.0f), ((val & 0x0000FF00u) >> 8) * (1.0f / 255.0f),
			          ((val & 0x00FF0000u) >> 16) * (1.0f / 255.0f), ((val & 0xFF000000u) >> 24) * (1.0f / 255.0f)};
			r.x = srgb_to_linear(r.x) * r.w; r.y = srgb_to_linear(r.y) * r.w; r.z = srgb_to_linear(r.z) * r.w;
			return r;
		}
		case NGP_IMAGE_HALF: {
			const uint16_t* p = (const uint16_t*)pixels + idx * 4;
			return {h2f(p[0]), h2f(p[1]), h2f(p[2]), h2f(p[3])};
		}
		case NGP_IMAGE_FLOAT: {
			const float* p = (const float*)pixels + idx * 4;
			return {p[0], p[1], p[2], p[3]};
		}
		default: return {5.0f, 0.0f, 0.0f, 1.0f};
	}
}

// opencv_lens_distortion_delta + iterative_lens_undistortion, common_device.cuh:268-345
inline void opencv_lens_distortion_delta(const float* p, float u, float v, float* du, float* dv) {
	const float k1 = p[0], k2 = p[1], p1 = p[2], p2 = p[3];
	const float u2 = u * u, uv = u * v, v2 = v * v, r2 = u2 + v2;
	const float radial = k1 * r2 + k2 * r2 * r2;
	*du = u * radial + 2.f * p1 * uv + p2 * (r2 + 2.f * u2);
	*dv = v * radial + 2.f * p2 * uv + p1 * (r2 + 2.f * v2);
}
inline void iterative_opencv_lens_undistortion(const float* params, float* u, float* v) {
	const uint32_t kNumIterations = 100;
	const float kMaxStepNorm = 1e-10f, kRelStepSize = 1e-6f;
	const float eps = std::numeric_limits<float>::epsilon();
	const float x0[2] = {*u, *v};
	float x[2] = {*u, *v};
	for (uint32_t i = 0; i < kNumIterations; ++i) {
		const float step0 = std::max(eps, std::fabs(kRelStepSize * x[0]));
		const float step1 = std::max(eps, std::fabs(kRelStepSize * x[1]));
		float dx[2], d0b[2], d0f[2], d1b[2], d1f[2];
		opencv_lens_distortion_delta(params, x[0], x[1], &dx[0], &dx[1]);
		opencv_lens_distortion_delta(params, x[0] - step0, x[1], &d0b[0], &d0b[1]);
		opencv_lens_distortion_delta(params, x[0] + step0, x[1], &d0f[0], &d0f[1]);
		opencv_lens_distortion_delta(params, x[0], x[1] - step1, &d1b[0], &d1b[1]);
		opencv_lens_distortion_delta(params, x[0], x[1] + step1, &d1f[0], &d1f[1]);
		// J is column-major: J[c][r]
		float J00 = 1 + (d0f[0] - d0b[0]) / (2 * step0);
		float J10 = (d1f[0] - d1b[0]) / (2 * step1);
		float J01 = (d0f[1] - d0b[1]) / (2 * step0);
		float J11 = 1 + (d1f[1] - d1b[1]) / (2 * step1);
		// step_x = inverse(J) * (x + dx - x0); matrix M = [[J00, J10],[J01, J11]] (row r, col c)
		float r0 = x[0] + dx[0] - x0[0], r1 = x[1] + dx[1] - x0[1];
		float det = J00 * J11 - J10 * J01;
		float s0 = (J11 * r0 - J10 * r1) / det;
		float s1 = (-J01 * r0 + J00 * r1) / det;
		x[0] -= s0; x[1] -= s1;
		if (s0 * s0 + s1 * s1 < kMaxStepNorm) break;
	}
	*u = x[0]; *v = x[1];
}

// opencv_fisheye_lens_distortion_delta, common_device.cuh:283-305
inline void opencv_fisheye_lens_distortion_delta(const float* p, float u, float v, float* du, float* dv) {
	const float r = std::sqrt(u * u + v * v);
	if (r > (float)std::numeric_limits<double>::epsilon()) {
		const float theta = std::atan(r);
		const float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
		const float thetad = theta * (1.f + p[0] * theta2 + p[1] * theta4 + p[2] * theta6 + p[3] * theta8);
		*du = u * thetad / r - u;
		*dv = v * thetad / r - v;
	} else { *du = 0.f; *dv = 0.f; }
}
// iterative_lens_undistortion (common_device.cuh:307-345) for an arbitrary distortion function
template <typename F> inline void iterative_lens_undistortion(const float* params, float* u, float* v, F distortion_fun) {
	const float eps = std::numeric_limits<float>::epsilon();
	const float x0[2] = {*u, *v};
	float x[2] = {*u, *v};
	for (uint32_t i = 0; i < 100; ++i) {
		const float step0 = std::max(eps, std::fabs(1e-6f * x[0])), step1 = std::max(eps, std::fabs(1e-6f * x[1]));
		float dx[2], d0b[2], d0f[2], d1b[2], d1f[2];
		distortion_fun(params, x[0], x[1], &dx[0], &dx[1]);
		distortion_fun(params, x[0] - step0, x[1], &d0b[0], &d0b[1]);
		distortion_fun(params, x[0] + step0, x[1], &d0f[0], &d0f[1]);
		distortion_fun(params, x[0], x[1] - step1, &d1b[0], &d1b[1]);
		distortion_fun(params, x[0], x[1] + step1, &d1f[0], &d1f[1]);
		const float J00 = 1 + (d0f[0] - d0b[0]) / (2 * step0), J10 = (d1f[0] - d1b[0]) / (2 * step1);
		const float J01 = (d0f[1] - d0b[1]) / (2 * step0), J11 = 1 + (d1f[1] - d1b[1]) / (2 * step1);
		const float r0 = x[0] + dx[0] - x0[0], r1 = x[1] + dx[1] - x0[1];
		const float det = J00 * J11 - J10 * J01;
		const float s0 = (J11 * r0 - J10 * r1) / det, s1 = (-J01 * r0 + J00 * r1) / det;
		x[0] -= s0; x[1] -= s1;
		if (s0 * s0 + s1 * s1 < 1e-10f) break;
	}
	*u = x[0]; *v = x[1];
}
// f_theta_undistortion, latlong / equirectangular mappings, common_device.cuh:368-411
inline vec3 f_theta_undistortion(vec2 uv, const float* params, vec3 error_direction) {
	const float xpix = uv.x * params[5], ypix = uv.y * params[6];
	const float norm = std::sqrt(xpix * xpix + ypix * ypix);
	const float alpha = params[0] + norm * (params[1] + norm * (params[2] + norm * (params[3] + norm * params[4])));
	float sin_alpha = std::sin(alpha), cos_alpha = std::cos(alpha);
	if (cos_alpha <= std::numeric_limits<float>::min() || norm == 0.f) return error_direction;
	sin_alpha *= 1.f / norm;
	return {sin_alpha * xpix, sin_alpha * ypix, cos_alpha};
}
constexpr float ORA_PI = 3.14159265358979323846f;
inline vec3 latlong_to_dir(vec2 uv) {
	const float theta = (uv.y - 0.5f) * ORA_PI, phi = (uv.x - 0.5f) * ORA_PI * 2.0f;
	return {std::sin(phi) * std::cos(theta), std::sin(theta), std::cos(phi) * std::cos(theta)};
}
inline vec3 equirectangular_to_dir(vec2 uv) {
	const float ct = (uv.y - 0.5f) * 2.0f, st = std::sqrt(std::max(1.0f - ct * ct, 0.0f)), phi = (uv.x - 0.5f) * ORA_PI * 2.0f;
	return {std::sin(phi) * st, ct, std::cos(phi) * st};
}
inline vec2 dir_to_latlong(vec3 dir) { return {std::atan2(dir.x, dir.z) / (ORA_PI * 2.0f) + 0.5f, std::asin(dir.y) / ORA_PI + 0.5f}; }
inline vec2 dir_to_equirectangular(vec3 dir) { return {std::atan2(dir.x, dir.z) / (ORA_PI * 2.0f) + 0.5f, dir.y / 2.0f + 0.5f}; }

// uv_to_ray, common_device.cuh:413-490: all seven lens modes; default foveation (a clamp of uv to the unit square); no hidden-area mask / distortion map / aperture;
// parallax_shift = 0.  Returns false for Ray::invalid() (f-theta outside its field of view).
inline bool uv_to_ray(vec2 uv, const int32_t res[2], const float focal[2], const mat4x3& cam, const float screen_center[2],
		int lens_mode, const float* lens_params, float near_distance, vec3& o, vec3& d) {
	// warped_uv = foveation.warp(uv) (:429): with the default Foveation every caller of this path passes, clamp(x, 0, 1) * 1 + 0 per axis (common_device.cuh:215-224)
	uv = {uv.x < 0.0f ? 0.0f : (uv.x > 1.0f ? 1.0f : uv.x), uv.y < 0.0f ? 0.0f : (uv.y > 1.0f ? 1.0f : uv.y)};
	vec3 head_pos = {0.f, 0.f, 0.f};
	vec3 dir;
	if (lens_mode == NGP_LENS_FTHETA) {
		dir = f_theta_undistortion({uv.x - screen_center[0], uv.y - screen_center[1]}, lens_params, {0.f, 0.f, 0.f});
		if (dir.x == 0.f && dir.y == 0.f && dir.z == 0.f) { o = cam[3]; d = dir; return false; }
	} else if (lens_mode == NGP_LENS_LATLONG) {
		dir = latlong_to_dir(uv);
	} else if (lens_mode == NGP_LENS_EQUIRECTANGULAR) {
		dir = equirectangular_to_dir(uv);
	} else if (lens_mode == NGP_LENS_ORTHOGRAPHIC) {
		dir = {0.0f, 0.0f, 1.0f};
		head_pos += vec3{(uv.x - screen_center[0]) * (float)res[0] / focal[0], (uv.y - screen_center[1]) * (float)res[1] / focal[1], 0.0f};
	} else {
		dir = {(uv.x - screen_center[0]) * (float)res[0] / focal[0], (uv.y - screen_center[1]) * (float)res[1] / focal[1], 1.0f};
		if (lens_mode == NGP_LENS_OPENCV) iterative_opencv_lens_undistortion(lens_params, &dir.x, &dir.y);
		else if (lens_mode == NGP_LENS_OPENCV_FISHEYE) iterative_lens_undistortion(lens_params, &dir.x, &dir.y, opencv_fisheye_lens_distortion_delta);
		else if (lens_mode != NGP_LENS_PERSPECTIVE) throw std::runtime_error("oracle: unknown lens mode");
	}
	dir = mul3(cam, dir);
	vec3 origin = mul3(cam, head_pos) + cam[3];
	origin += dir * near_distance;
	o = origin; d = dir;
	return true;
}

// image_idx (no cdf), nerf_device.cuh:578-599
inline uint32_t image_idx(uint32_t base_idx, uint32_t n_rays, uint32_t n_training_images) {
	return ((base_idx * n_training_images) / n_rays) % n_training_images; // uint32 arithmetic, as on the device
}
// nerf_random_image_pos_training (no cdf), nerf_device.cuh:553-576
inline vec2 random_image_pos_training(Pcg32& rng, const int32_t res[2], bool snap) {
	vec2 uv; uv.x = rng.next_float(); uv.y = rng.next_float();
	if (snap) {
		uv.x = ((float)clampi((int)(uv.x * (float)res[0]), 0, res[0] - 1) + 0.5f) / (float)res[0];
		uv.y = ((float)clampi((int)(uv.y * (float)res[1]), 0, res[1] - 1) + 0.5f) / (float)res[1];
	}
	return uv;
}

// ---- training pixels drawn in proportion to the accumulated error ----
struct ErrorCdf { const float* x_cond_y = nullptr; const float* y = nullptr; const float* img = nullptr; int32_t res[2] = {0, 0}; };
// binary_search, common.h:207-230
inline uint32_t binary_search(float val, const float* data, uint32_t length) {
	if (length == 0) return 0;
	uint32_t it, count = length, step, first = 0;
	while (count > 0) {
		it = first; step = count / 2; it += step;
		if (data[it] < val) { first = ++it; count -= step + 1; } else count = step;
	}
	return std::min(first, length - 1);
}
// sample_cdf_2d, nerf_device.cuh:499-528 (UNIFORM_SAMPLING_FRACTION = 0.5: *pdf stays untouched on the uniform branch)
inline vec2 sample_cdf_2d(vec2 sample, uint32_t img, const int32_t res[2], const float* cdf_x_cond_y, const float* cdf_y, float* pdf) {
	const float UNIFORM_SAMPLING_FRACTION = 0.5f;
	if (sample.x < UNIFORM_SAMPLING_FRACTION) { sample.x /= UNIFORM_SAMPLING_FRACTION; return sample; }
	sample.x = (sample.x - UNIFORM_SAMPLING_FRACTION) / (1.0f - UNIFORM_SAMPLING_FRACTION);
	cdf_y += (size_t)img * res[1];
	uint32_t y = binary_search(sample.y, cdf_y, (uint32_t)res[1]);
	float prev = y > 0 ? cdf_y[y - 1] : 0.0f;
	float pmf_y = cdf_y[y] - prev;
	sample.y = (sample.y - prev) / pmf_y;
	cdf_x_cond_y += (size_t)img * res[1] * res[0] + (size_t)y * res[0];
	uint32_t x = binary_search(sample.x, cdf_x_cond_y, (uint32_t)res[0]);
	prev = x > 0 ? cdf_x_cond_y[x - 1] : 0.0f;
	float pmf_x = cdf_x_cond_y[x] - prev;
	sample.x = (sample.x - prev) / pmf_x;
	if (pdf) *pdf = pmf_x * pmf_y * (float)(res[0] * res[1]);
	return {((float)x + sample.x) / (float)res[0], ((float)y + sample.y) / (float)res[1]};
}
// compute_extra_dims_gradient_train_nerf (testbed_nerf.cu:1293-1330): every compacted ray adds its samples' dL/d(extra dims) -- the extra-dims columns of the network's
// input gradient, coords_gradient(j)->get_extra_dims() -- to its image's gradient, in ray and sample order (the reference: float atomicAdd per sample and dim).
inline void extra_dims_gradient(uint32_t n_rays, uint32_t n_rays_total, uint32_t rays_counter, float* extra_dims_gradient_out, uint32_t n_extra_dims, uint32_t n_training_images,
		const uint32_t* ray_indices_in, const uint32_t* numsteps_in, const float* coords_gradient_extra /* row-major [row][n_extra_dims] */) {
	(void)n_rays;
	for (uint32_t i = 0; i < rays_counter; ++i) {
		const uint32_t numsteps = numsteps_in[i * 2 + 0];
		if (numsteps == 0) continue;
		const uint32_t base = numsteps_in[i * 2 + 1];
		const uint32_t img = image_idx(ray_indices_in[i], n_rays_total, n_training_images);
		float* g = extra_dims_gradient_out + (size_t)n_extra_dims * img;
		for (uint32_t j = 0; j < numsteps; ++j)
			for (uint32_t k = 0; k < n_extra_dims; ++k) g[k] += coords_gradient_extra[(size_t)(base + j) * n_extra_dims + k];
	}
}
// VarAdamOptimizer::step (adam_optimizer.h:37-47) as driven by testbed_nerf.cu:2860-2878: gradient / LOSS_SCALE, the network optimizer's learning rate, hyperparameters of
// VarAdamOptimizer(n_extra_dims, 1e-4f) (epsilon 1e-8, beta1 0.9, beta2 0.99); iter = the optimizer's iteration count AFTER this step (>= 1)
inline void var_adam_step(uint32_t n, float* variable, const float* gradient_scaled, float* first_moment, float* second_moment, uint32_t iter, float learning_rate, float loss_scale) {
	const float epsilon = 1e-8f, beta1 = 0.9f, beta2 = 0.99f;
	const float actual_learning_rate = learning_rate * std::sqrt(1.0f - std::pow(beta2, (float)iter)) / (1.0f - std::pow(beta1, (float)iter));
	for (uint32_t i = 0; i < n; ++i) {
		const float g = gradient_scaled[i] / loss_scale;
		first_moment[i] = beta1 * first_moment[i] + (1.0f - beta1) * g;
		second_moment[i] = beta2 * second_moment[i] + (1.0f - beta2) * g * g;
		variable[i] -= actual_learning_rate * first_moment[i] / (std::sqrt(second_moment[i]) + epsilon);
	}
}

// image_idx with a CDF over the images, nerf_device.cuh:578-591
inline uint32_t image_idx_cdf(uint32_t base_idx, uint32_t n_training_images, const float* cdf, float* pdf) {
	float sample = ld_random_val(base_idx, 0xdeadbeef);
	uint32_t img = binary_search(sample, cdf, n_training_images);
	if (pdf) { float prev = img > 0 ? cdf[img - 1] : 0.0f; *pdf = (cdf[img] - prev) * n_training_images; }
	return img;
}
// the pixel of global ray i as K1 (testbed_nerf.cu:726-730) and K3 (:955-961) derive it; rng is left behind the two uv draws
inline vec2 training_pixel(const ErrorCdf* cdf, Pcg32& rng, uint32_t i, uint32_t n_rays, uint32_t n_images, const ngp_image_meta* meta, bool snap, uint32_t& img, float& img_pdf, float& uv_pdf) {
	img_pdf = 1.0f; uv_pdf = 1.0f;
	img = (cdf && cdf->img) ? image_idx_cdf(i, n_images, cdf->img, &img_pdf) : image_idx(i, n_rays, n_images);
	const int32_t* res = meta[img].resolution;
	vec2 uv; uv.x = rng.next_float(); uv.y = rng.next_float();
	if (cdf && cdf->x_cond_y) uv = sample_cdf_2d(uv, img, cdf->res, cdf->x_cond_y, cdf->y, &uv_pdf);
	if (snap) {
		uv.x = ((float)clampi((int)(uv.x * (float)res[0]), 0, res[0] - 1) + 0.5f) / (float)res[0];
		uv.y = ((float)clampi((int)(uv.y * (float)res[1]), 0, res[1] - 1) + 0.5f) / (float)res[1];
	}
	return uv;
}
// construct_cdf_2d / construct_cdf_1d / the host loop over the images, testbed_nerf.cu:1530-1580, 2832-2847
inline void construct_error_cdfs(uint32_t n_images, uint32_t width, uint32_t height, const float* data, float* cdf_x_cond_y, float* cdf_y, float* cdf_img) {
	const float MIN_PDF = 0.01f;
	for (uint32_t img = 0; img < n_images; ++img) for (uint32_t y = 0; y < height; ++y) {
		const size_t off = ((size_t)img * height + y) * width;
		float cum = 0;
		for (uint32_t x = 0; x < width; ++x) { cum += data[off + x] + 1e-10f; cdf_x_cond_y[off + x] = cum; }
		cdf_y[img * height + y] = cum;
		float norm = 1.0f / cum; // __frcp_rn
		for (uint32_t x = 0; x < width; ++x) cdf_x_cond_y[off + x] = (1.0f - MIN_PDF) * cdf_x_cond_y[off + x] * norm + MIN_PDF * (float)(x + 1) / (float)width;
	}
	for (uint32_t img = 0; img < n_images; ++img) {
		float* cy = cdf_y + (size_t)img * height;
		float cum = 0;
		for (uint32_t y = 0; y < height; ++y) { cum += cy[y]; cy[y] = cum; }
		cdf_img[img] = cum;
		float norm = 1.0f / cum;
		for (uint32_t y = 0; y < height; ++y) cy[y] = (1.0f - MIN_PDF) * cy[y] * norm + MIN_PDF * (float)(y + 1) / (float)height;
	}
	float cum = 0;
	for (uint32_t i = 0; i < n_images; ++i) { cum += cdf_img[i]; cdf_img[i] = cum; }
	float norm = 1.0f / cum;
	const float MIN_PMF = 0.1f;
	for (uint32_t i = 0; i < n_images; ++i) cdf_img[i] = (1.0f - MIN_PMF) * cdf_img[i] * norm + MIN_PMF * (float)(i + 1) / (float)n_images;
}

// losses, nerf_device.cuh:75-143, 601-616
struct LossAndGradient { vec3 loss, gradient; };
inline LossAndGradient loss_and_gradient(vec3 target, vec3 pred, int type) {
	vec3 diff = pred - target;
	LossAndGradient r;
	auto cps = [](float mag, float s) { return std::copysign(mag, s); };
	switch (type) {
		case NGP_LOSS_RELATIVE_L2: { vec3 den = pred * pred + 1e-2f; r.loss = diff * diff / den; r.gradient = 2.0f * diff / den; break; }
		case NGP_LOSS_L1: { r.loss = vabs(diff); r.gradient = {cps(1.f, diff.x), cps(1.f, diff.y), cps(1.f, diff.z)}; break; }
		case NGP_LOSS_MAPE: { vec3 den = vabs(pred) + 1e-2f; r.loss = vabs(diff) / den;
			r.gradient = {cps(1.f / den.x, diff.x), cps(1.f / den.y, diff.y), cps(1.f / den.z, diff.z)}; break; }
		case NGP_LOSS_SMAPE: { vec3 den = 0.5f * (vabs(pred) + vabs(target)) + 1e-2f; r.loss = vabs(diff) / den;
			r.gradient = {cps(1.f / den.x, diff.x), cps(1.f / den.y, diff.y), cps(1.f / den.z, diff.z)}; break; }
		case NGP_LOSS_HUBER: {
			const float alpha = 0.1f;
			for (int k = 0; k < 3; ++k) {
				float df = diff[k], ad = std::fabs(df), sq = 0.5f / alpha * df * df;
				r.loss[k] = (ad > alpha ? (ad - 0.5f * alpha) : sq) / 5.0f;
				r.gradient[k] = (ad > alpha ? (df > 0 ? 1.0f : -1.0f) : (df / alpha)) / 5.0f;
			}
			break; }
		case NGP_LOSS_LOGL1: { vec3 dv = vabs(diff) + 1.0f; r.loss = {std::log(dv.x), std::log(dv.y), std::log(dv.z)};
			r.gradient = {cps(1.f / dv.x, diff.x), cps(1.f / dv.y, diff.y), cps(1.f / dv.z, diff.z)}; break; }
		default: { r.loss = diff * diff; r.gradient = 2.0f * diff; break; }
	}
	return r;
}

// -------------------------------------------------------------------------------------------------
// K1: generate_training_samples_nerf, testbed_nerf.cu:691-849.  Threads (= rays) run in index order,
// so `base` offsets are deterministic here (the device's atomics order is not).
// ray range [ray_begin, ray_end) of the global range [0, n_rays) = this rank's shard.
// -------------------------------------------------------------------------------------------------
struct K1Out {
	uint32_t ray_counter = 0, numsteps_counter = 0;
};
inline K1Out generate_training_samples(uint32_t n_rays, uint32_t ray_begin, uint32_t ray_end, const Aabb& aabb, uint32_t max_samples,
		const Pcg32& rng_in, uint32_t* ray_indices_out, ngp_ray* rays_out, uint32_t* numsteps_out, float* coords_out /* 7 floats each */,
		uint32_t n_images, const ngp_image_meta* meta, const ngp_xform* xforms, const uint8_t* bitfield, uint32_t max_mip,
		bool snap_to_pixel_centers, float cone_angle_constant, const ErrorCdf* cdf = nullptr) {
	K1Out k;
	for (uint32_t i = ray_begin; i < ray_end && i < n_rays; ++i) {
		Pcg32 rng = rng_in;
		rng.advance((int64_t)(i * N_MAX_RANDOM_SAMPLES_PER_RAY));
		uint32_t img; float img_pdf, uv_pdf;
		vec2 uv = training_pixel(cdf, rng, i, n_rays, n_images, meta, snap_to_pixel_centers, img, img_pdf, uv_pdf);
		const ngp_image_meta& m = meta[img];
		if (read_rgba(uv, m.resolution, m.pixels, m.image_data_type).x < 0.0f) continue; // masked away
		/* max_level_rand_training = false: max_level = 1, no draw */
		float motionblur_time = rng.next_float();
		const mat4x3 xform = get_xform_given_rolling_shutter(xforms[img], m.rolling_shutter, uv, motionblur_time); // common_device.cuh:670-674
		vec3 ro, rd;
		if (!uv_to_ray(uv, m.resolution, m.focal_length, xform, m.principal_point, m.lens_mode, m.lens_params, 0.0f, ro, rd)) { ro = xform[3]; rd = xform[2]; } // testbed_nerf.cu:776-778
		vec3 rdn = normalize(rd);
		vec2 tminmax = aabb.ray_intersect(ro, rdn);
		float cone_angle = cone_angle_constant; // calc_cone_angle, nerf_device.cuh:370-377
		tminmax.x = std::fmax(tminmax.x, 0.0f);
		float startt = advance_n_steps(tminmax.x, cone_angle, rng.next_float());
		vec3 idir = V3(1.0f) / rdn;

		uint32_t j = 0;
		float t = startt;
		vec3 pos;
		while (aabb.contains(pos = ro + t * rdn) && j < NERF_STEPS) {
			float dt = calc_dt(t, cone_angle);
			uint32_t mip = mip_from_dt(dt, pos, max_mip);
			if (density_grid_occupied_at(pos, bitfield, mip)) { ++j; t += dt; }
			else t = advance_to_next_voxel(t, cone_angle, pos, rdn, idir, mip);
		}
		if (j == 0) continue;
		uint32_t numsteps = j;
		uint32_t base = k.numsteps_counter; k.numsteps_counter += numsteps;
		if (base + numsteps > max_samples) continue;
		float* co = coords_out + (size_t)base * 7;
		uint32_t ray_idx = k.ray_counter++;
		ray_indices_out[ray_idx] = i;
		rays_out[ray_idx] = {{ro.x, ro.y, ro.z}, {rd.x, rd.y, rd.z}};
		numsteps_out[ray_idx * 2 + 0] = numsteps;
		numsteps_out[ray_idx * 2 + 1] = base;
		vec3 wd = warp_direction(rdn);
		t = startt; j = 0;
		while (aabb.contains(pos = ro + t * rdn) && j < numsteps) {
			float dt = calc_dt(t, cone_angle);
			uint32_t mip = mip_from_dt(dt, pos, max_mip);
			if (density_grid_occupied_at(pos, bitfield, mip)) {
				vec3 wp = warp_position(pos, aabb);
				float* c = co + (size_t)j * 7;
				c[0] = wp.x; c[1] = wp.y; c[2] = wp.z; c[3] = warp_dt(dt); c[4] = wd.x; c[5] = wd.y; c[6] = wd.z;
				++j; t += dt;
			} else t = advance_to_next_voxel(t, cone_angle, pos, rdn, idir, mip);
		}
	}
	return k;
}

// -------------------------------------------------------------------------------------------------
// CPU model of the PRODUCTION ray marcher of libngp_hip (csrc/nerf_kernels.hip k1_setup / k1_count): NOT a restatement of a
// reference function but of this repo's sample-parallel reformulation of testbed_nerf.cu:798-807, kept here so that the tests can
// (i) quantify on the CPU how the reformulation differs from the sequential loop above and (ii) compare the device kernel with the
// same algorithm on the host.  Every visited march parameter of the reference loop is a lattice point t_j = from_stepping_space(n'+j),
// n' = to_stepping_space(startt): a step through an occupied voxel is j -> j+1 (`t += dt`, dt = from(to(t)+1) - t), and a skip
// is j -> j + ceil(max(to(t_target) - to(t), 0.5)) (advance_to_next_voxel, nerf_device.cuh:431-441).
//   mode 0 ("independent"): lattice point j is a sample iff it lies in an occupied voxel at ITS OWN mip (exact while the mip is
//                           constant along a skipped voxel, i.e. cone_angle == 0);
//   mode 1 ("walk"):        the orbit of j = 0 under the reference's own update rule, evaluated on the lattice.
// out_counts[i - ray_begin] = number of samples of global ray i (0 for masked / empty rays).
// -------------------------------------------------------------------------------------------------
inline void lattice_march_counts(int mode, uint32_t n_rays, uint32_t ray_begin, uint32_t ray_end, const Aabb& aabb, const Pcg32& rng_in, uint32_t n_images,
		const ngp_image_meta* meta, const ngp_xform* xforms, const uint8_t* bitfield, uint32_t max_mip, bool snap_to_pixel_centers, float cone_angle,
		uint32_t* out_counts, uint32_t max_lattice_points) {
#pragma omp parallel for schedule(dynamic, 64)
	for (int64_t ii = ray_begin; ii < (int64_t)std::min(ray_end, n_rays); ++ii) {
		const uint32_t i = (uint32_t)ii;
		out_counts[i - ray_begin] = 0;
		uint32_t img = image_idx(i, n_rays, n_images);
		const ngp_image_meta& m = meta[img];
		Pcg32 rng = rng_in;
		rng.advance((int64_t)(i * N_MAX_RANDOM_SAMPLES_PER_RAY));
		vec2 uv = random_image_pos_training(rng, m.resolution, snap_to_pixel_centers);
		if (read_rgba(uv, m.resolution, m.pixels, m.image_data_type).x < 0.0f) continue;
		const float motionblur_time = rng.next_float();
		const mat4x3 xform = get_xform_given_rolling_shutter(xforms[img], m.rolling_shutter, uv, motionblur_time);
		vec3 ro, rd;
		if (!uv_to_ray(uv, m.resolution, m.focal_length, xform, m.principal_point, m.lens_mode, m.lens_params, 0.0f, ro, rd)) { ro = xform[3]; rd = xform[2]; }
		vec3 rdn = normalize(rd);
		vec2 tminmax = aabb.ray_intersect(ro, rdn);
		tminmax.x = std::fmax(tminmax.x, 0.0f);
		const float startt = advance_n_steps(tminmax.x, cone_angle, rng.next_float());
		const float nprime = to_stepping_space(startt, cone_angle);
		const vec3 idir = V3(1.0f) / rdn;
		uint32_t cnt = 0, j = 0;
		while (j < max_lattice_points && cnt < NERF_STEPS) {
			const float t = j == 0 ? startt : from_stepping_space(nprime + (float)j, cone_angle);
			const vec3 pos = ro + t * rdn;
			if (!aabb.contains(pos)) break;
			const float dt = calc_dt(t, cone_angle);
			const uint32_t mip = mip_from_dt(dt, pos, max_mip);
			if (density_grid_occupied_at(pos, bitfield, mip)) { ++cnt; ++j; continue; }
			if (mode == 0) { ++j; continue; }
			// the reference's skip, in lattice units
			const float res = scalbnf((float)NERF_GRIDSIZE, -(int)mip);
			const float t_target = t + distance_to_next_voxel(pos, rdn, idir, res);
			const float k = ceilf(std::fmax(to_stepping_space(t_target, cone_angle) - to_stepping_space(t, cone_angle), 0.5f));
			j += (uint32_t)k;
		}
		out_counts[i - ray_begin] = cnt;
	}
}

// WHERE does the lattice walk (mode 1 above, the production marcher's algorithm) leave the reference's loop (testbed_nerf.cu:798-807)?  Both marches of one ray
// are run in lockstep over their visited points: the reference's t (accumulated: t += dt, t = advance_to_next_voxel(...)) against the closed-form lattice
// t_j.  While all decisions agree the two visit "the same" points up to rounding of t; the first disagreement is classified:
//   1 BOX_FACE    one position is inside the box and the other is not
//   2 MIP         the two positions / step sizes select different cascades (cone_angle > 0 only)
//   3 VOXEL_FACE  same cascade, different occupancy: the two positions must lie in DIFFERENT cells (else the test is the same bit)
//   4 SKIP        same empty cell, but the skip lands on different lattice points (t_target sits on a lattice point up to rounding)
//   5 OTHER       anything else (must not happen: same cell, same bit, same skip)
// out per ray: cause (0 = the marches agree to the end), sample counts of both, |t_ref - t_lattice| in ulps of t at the divergence (max over the walk if none),
// and for 3 the distance (in cells of that cascade) of the reference's position to the nearest cell face, for 4 the distance (in lattice steps) of the skip length to
// the nearest integer (its ceil() picks the landing point).
struct K1Divergence { uint32_t cause, count_ref, count_lattice; float t_ulps, face_distance_cells; };
inline void lattice_vs_reference_divergence(uint32_t n_rays, const Aabb& aabb, const Pcg32& rng_in, uint32_t n_images, const ngp_image_meta* meta, const ngp_xform* xforms,
		const uint8_t* bitfield, uint32_t max_mip, bool snap_to_pixel_centers, float cone_angle, K1Divergence* out, uint32_t max_lattice_points) {
#pragma omp parallel for schedule(dynamic, 64)
	for (int64_t ii = 0; ii < (int64_t)n_rays; ++ii) {
		const uint32_t i = (uint32_t)ii;
		K1Divergence r = {0u, 0u, 0u, 0.f, 0.f};
		out[i] = r;
		uint32_t img = image_idx(i, n_rays, n_images);
		const ngp_image_meta& m = meta[img];
		Pcg32 rng = rng_in;
		rng.advance((int64_t)(i * N_MAX_RANDOM_SAMPLES_PER_RAY));
		vec2 uv = random_image_pos_training(rng, m.resolution, snap_to_pixel_centers);
		if (read_rgba(uv, m.resolution, m.pixels, m.image_data_type).x < 0.0f) continue;
		const float motionblur_time = rng.next_float();
		const mat4x3 xform = get_xform_given_rolling_shutter(xforms[img], m.rolling_shutter, uv, motionblur_time);
		vec3 ro, rd;
		if (!uv_to_ray(uv, m.resolution, m.focal_length, xform, m.principal_point, m.lens_mode, m.lens_params, 0.0f, ro, rd)) { ro = xform[3]; rd = xform[2]; }
		const vec3 rdn = normalize(rd);
		vec2 tminmax = aabb.ray_intersect(ro, rdn);
		tminmax.x = std::fmax(tminmax.x, 0.0f);
		const float startt = advance_n_steps(tminmax.x, cone_angle, rng.next_float());
		const float nprime = to_stepping_space(startt, cone_angle);
		const vec3 idir = V3(1.0f) / rdn;
		auto ulps = [](float a, float b) { int e; std::frexp(std::fmax(std::fabs(a), std::fabs(b)), &e); return std::fabs(a - b) / std::scalbn(1.0f, e - 24); };
		auto cell = [&](vec3 pos, uint32_t mip, int c[3]) { const float sc = std::scalbn(1.0f, -(int)mip); vec3 q = (pos - V3(0.5f)) * sc + V3(0.5f); c[0] = (int)(q.x * (float)NERF_GRIDSIZE); c[1] = (int)(q.y * (float)NERF_GRIDSIZE); c[2] = (int)(q.z * (float)NERF_GRIDSIZE); };
		float t_ref = startt; uint32_t j = 0; bool diverged = false;
		while (j < max_lattice_points && r.count_ref < NERF_STEPS && r.count_lattice < NERF_STEPS) {
			const float t_lat = j == 0 ? startt : from_stepping_space(nprime + (float)j, cone_angle);
			const vec3 p_ref = ro + t_ref * rdn, p_lat = ro + t_lat * rdn;
			const float u = ulps(t_ref, t_lat);
			r.t_ulps = std::fmax(r.t_ulps, u);
			const bool in_ref = aabb.contains(p_ref), in_lat = aabb.contains(p_lat);
			if (in_ref != in_lat) { r.cause = 1; r.t_ulps = u; diverged = true; break; }
			if (!in_ref) break;
			const float dt_ref = calc_dt(t_ref, cone_angle), dt_lat = calc_dt(t_lat, cone_angle);
			const uint32_t mip_ref = mip_from_dt(dt_ref, p_ref, max_mip), mip_lat = mip_from_dt(dt_lat, p_lat, max_mip);
			if (mip_ref != mip_lat) { r.cause = 2; r.t_ulps = u; diverged = true; break; }
			const bool occ_ref = density_grid_occupied_at(p_ref, bitfield, mip_ref), occ_lat = density_grid_occupied_at(p_lat, bitfield, mip_lat);
			int c_ref[3], c_lat[3]; cell(p_ref, mip_ref, c_ref); cell(p_lat, mip_lat, c_lat);
			const bool same_cell = c_ref[0] == c_lat[0] && c_ref[1] == c_lat[1] && c_ref[2] == c_lat[2];
			if (occ_ref != occ_lat) {
				r.cause = same_cell ? 5 : 3; r.t_ulps = u; diverged = true;
				const float sc = std::scalbn(1.0f, -(int)mip_ref); const vec3 q = ((p_ref - V3(0.5f)) * sc + V3(0.5f)) * (float)NERF_GRIDSIZE;
				auto fd = [](float x) { return std::fabs(x - std::nearbyint(x)); };
				r.face_distance_cells = std::fmin(fd(q.x), std::fmin(fd(q.y), fd(q.z)));
				break;
			}
			if (occ_ref) { ++r.count_ref; ++r.count_lattice; t_ref += dt_ref; ++j; continue; }
			const float t_next = advance_to_next_voxel(t_ref, cone_angle, p_ref, rdn, idir, mip_ref);
			const float res = scalbnf((float)NERF_GRIDSIZE, -(int)mip_lat);
			const float t_target = t_lat + distance_to_next_voxel(p_lat, rdn, idir, res);
			const uint32_t k = (uint32_t)ceilf(std::fmax(to_stepping_space(t_target, cone_angle) - to_stepping_space(t_lat, cone_angle), 0.5f));
			const float t_lat_next = from_stepping_space(nprime + (float)(j + k), cone_angle);
			// the same landing point?  compare in stepping space: a different lattice point is >= 1 apart there
			if (std::fabs(to_stepping_space(t_next, cone_angle) - to_stepping_space(t_lat_next, cone_angle)) > 0.5f) { r.cause = same_cell ? 4 : 3; r.t_ulps = u; diverged = true;
				if (same_cell) { // how close was the skip length (in lattice steps) to an integer on either side?  (ceil() of it decides the landing point)
					const float res_r = scalbnf((float)NERF_GRIDSIZE, -(int)mip_ref);
					const float x_ref = to_stepping_space(t_ref + distance_to_next_voxel(p_ref, rdn, idir, res_r), cone_angle) - to_stepping_space(t_ref, cone_angle);
					const float x_lat = to_stepping_space(t_target, cone_angle) - to_stepping_space(t_lat, cone_angle);
					auto fi = [](float x) { return std::fabs(x - std::nearbyint(x)); };
					r.face_distance_cells = std::fmin(fi(x_ref), fi(x_lat));
				}
				if (!same_cell) { const float sc = std::scalbn(1.0f, -(int)mip_ref); const vec3 q = ((p_ref - V3(0.5f)) * sc + V3(0.5f)) * (float)NERF_GRIDSIZE; auto fd = [](float x) { return std::fabs(x - std::nearbyint(x)); }; r.face_distance_cells = std::fmin(fd(q.x), std::fmin(fd(q.y), fd(q.z))); }
				break; }
			t_ref = t_next; j += k;
		}
		if (diverged) { // the counts of the two complete marches, for the statistics
			uint32_t jj = 0, cnt = 0;
			while (jj < max_lattice_points && cnt < NERF_STEPS) {
				const float t = jj == 0 ? startt : from_stepping_space(nprime + (float)jj, cone_angle);
				const vec3 pos = ro + t * rdn;
				if (!aabb.contains(pos)) break;
				const uint32_t mip = mip_from_dt(calc_dt(t, cone_angle), pos, max_mip);
				if (density_grid_occupied_at(pos, bitfield, mip)) { ++cnt; ++jj; continue; }
				const float res = scalbnf((float)NERF_GRIDSIZE, -(int)mip);
				jj += (uint32_t)ceilf(std::fmax(to_stepping_space(t + distance_to_next_voxel(pos, rdn, idir, res), cone_angle) - to_stepping_space(t, cone_angle), 0.5f));
			}
			r.count_lattice = cnt;
			float t = startt; vec3 pos; cnt = 0;
			while (aabb.contains(pos = ro + t * rdn) && cnt < NERF_STEPS) {
				const float dt = calc_dt(t, cone_angle);
				const uint32_t mip = mip_from_dt(dt, pos, max_mip);
				if (density_grid_occupied_at(pos, bitfield, mip)) { ++cnt; t += dt; } else t = advance_to_next_voxel(t, cone_angle, pos, rdn, idir, mip);
			}
			r.count_ref = cnt;
		}
		out[i] = r;
	}
}

// -------------------------------------------------------------------------------------------------
// K3: compute_loss_kernel_train_nerf, testbed_nerf.cu:852-1180 (no envmap / error-map / exposure: off by default; depth supervision
// :1027-1029, :1126-1129 behind depth_lambda > 0).  __expf is restated as expf.
// -------------------------------------------------------------------------------------------------
struct K3Opts {
	float loss_scale = 128.f;
	vec3 background_color = {0, 0, 0};
	bool color_space_srgb = false, random_bg = true, linear_colors = false, snap = true;
	int loss_type = NGP_LOSS_HUBER, rgb_act = NGP_ACT_LOGISTIC, density_act = NGP_ACT_EXPONENTIAL;
	float near_distance = 0.1f;
	int train_mode = 0; // ETrainMode: 0 Nerf, 1 Rfl, 2 RflRelax (fused_kernels/train_nerf.cuh:391-410)
	float depth_lambda = 0.f; int depth_loss_type = NGP_LOSS_L1; // depth_supervision_lambda, depth_loss_type (testbed.h:796, 824)
};
// read_depth, common_device.cuh:874-878 (image_pos: pixel of a uv, clamped)
inline float read_depth(vec2 uv, const int32_t res[2], const float* depth) {
	int px = std::min(std::max((int)(uv.x * (float)res[0]), 0), res[0] - 1), py = std::min(std::max((int)(uv.y * (float)res[1]), 0), res[1] - 1);
	return depth[(size_t)px + (size_t)py * res[0]];
}
inline uint32_t compute_loss(uint32_t n_rays, uint32_t rays_counter, const Aabb& aabb, const Pcg32& rng_in, uint32_t max_samples_compacted,
		const K3Opts& o, uint32_t n_images, const ngp_image_meta* meta, const uint16_t* network_output, uint32_t out_stride,
		const uint32_t* ray_indices_in, const ngp_ray* rays_in, uint32_t* numsteps_inout, const float* coords_in, float* coords_out,
		uint16_t* dloss_doutput, uint32_t dl_stride, float* loss_output, float mean_density,
		const ErrorCdf* cdf = nullptr, float* error_map = nullptr, const int32_t* error_map_res = nullptr) {
	uint32_t counter = 0;
	for (uint32_t i = 0; i < rays_counter; ++i) {
		uint32_t numsteps = numsteps_inout[i * 2 + 0];
		uint32_t base = numsteps_inout[i * 2 + 1];
		const float* cin = coords_in + (size_t)base * 7;
		const uint16_t* no = network_output + (size_t)base * out_stride;

		uint32_t ray_idx = ray_indices_in[i];
		Pcg32 rng = rng_in;
		rng.advance((int64_t)(ray_idx * N_MAX_RANDOM_SAMPLES_PER_RAY));
		uint32_t img; float img_pdf, uv_pdf;
		vec2 uv = training_pixel(cdf, rng, ray_idx, n_rays, n_images, meta, o.snap, img, img_pdf, uv_pdf);
		const ngp_image_meta& m = meta[img];
		rng.advance(1); // motionblur_time
		vec3 background_color = o.background_color;
		if (o.random_bg) { background_color.x = rng.next_float(); background_color.y = rng.next_float(); background_color.z = rng.next_float(); }
		background_color = srgb_to_linear(background_color);

		vec4 texsamp = read_rgba(uv, m.resolution, m.pixels, m.image_data_type);
		vec3 trgb = {texsamp.x, texsamp.y, texsamp.z};
		vec3 rgbtarget;
		if (o.linear_colors || !o.color_space_srgb) {
			rgbtarget = trgb + (1.0f - texsamp.w) * background_color; // exposure_scale = exp(0) = 1
			if (!o.linear_colors) {
				rgbtarget = linear_to_srgb(rgbtarget);
				background_color = linear_to_srgb(background_color);
			}
		} else {
			background_color = linear_to_srgb(background_color);
			if (texsamp.w > 0) rgbtarget = linear_to_srgb(trgb / texsamp.w) * texsamp.w + (1.0f - texsamp.w) * background_color;
			else rgbtarget = background_color;
		}
		vec3 loss_bg = V3(0.f); // Rfl: sum of weight * per-sample loss (train_nerf.cuh:219)
		float T = 1.f;
		const float EPSILON = 1e-4f;
		vec3 rgb_ray = V3(0.f);
		float depth_ray = 0.f;
		uint32_t compacted_numsteps = 0;
		vec3 ray_o = V3(rays_in[i].o);
		for (; compacted_numsteps < numsteps; ++compacted_numsteps) {
			if (T < EPSILON) break;
			const uint16_t* lo = no + (size_t)compacted_numsteps * out_stride;
			vec3 rgb = {network_to_rgb(h2f(lo[0]), o.rgb_act), network_to_rgb(h2f(lo[1]), o.rgb_act), network_to_rgb(h2f(lo[2]), o.rgb_act)};
			const float dt = unwarp_dt(cin[compacted_numsteps * 7 + 3]);
			float density = network_to_density(h2f(lo[3]), o.density_act);
			const float alpha = 1.f - std::exp(-density * dt);
			const float weight = alpha * T;
			rgb_ray += weight * rgb;
			depth_ray += weight * distance(unwarp_position(V3(cin + (size_t)compacted_numsteps * 7), aabb), ray_o);
			if (o.train_mode == 1) loss_bg += weight * loss_and_gradient(rgbtarget, rgb, o.loss_type).loss;
			T *= (1.f - alpha);
		}

		if (compacted_numsteps == numsteps) { rgb_ray += T * background_color; if (o.train_mode == 1) loss_bg += T * loss_and_gradient(rgbtarget, background_color, o.loss_type).loss; }

		uint32_t compacted_base = counter; counter += compacted_numsteps;
		compacted_numsteps = std::min(max_samples_compacted - std::min(max_samples_compacted, compacted_base), compacted_numsteps);
		numsteps_inout[i * 2 + 0] = compacted_numsteps;
		numsteps_inout[i * 2 + 1] = compacted_base;
		if (compacted_numsteps == 0) continue;

		float* cout = coords_out + (size_t)compacted_base * 7;
		uint16_t* dl = dloss_doutput + (size_t)compacted_base * dl_stride;

		LossAndGradient lg = loss_and_gradient(rgbtarget, rgb_ray, o.loss_type);
		if (cdf && (cdf->x_cond_y || cdf->img)) lg.loss = lg.loss / (img_pdf * uv_pdf); // :1024; the gradient is deliberately not divided (:1031-1035)
		float mean_loss = mean(lg.loss);
		if (loss_output) loss_output[i] = mean_loss / (float)n_rays;
		if (error_map) { // testbed_nerf.cu:1042-1071 (no sharpness data)
			const float px = std::fmin(std::fmax(uv.x * (float)error_map_res[0] - 0.5f, 0.0f), (float)error_map_res[0] - (1.0f + 1e-4f));
			const float py = std::fmin(std::fmax(uv.y * (float)error_map_res[1] - 0.5f, 0.0f), (float)error_map_res[1] - (1.0f + 1e-4f));
			const int ix = (int)px, iy = (int)py;
			const float wx = px - (float)ix, wy = py - (float)iy;
			const int x = clampi(ix, 0, m.resolution[0] - 2), y = clampi(iy, 0, m.resolution[1] - 2);
			float* e = error_map + (size_t)img * error_map_res[0] * error_map_res[1];
			e[y * error_map_res[0] + x] += (1 - wx) * (1 - wy) * mean_loss;
			e[y * error_map_res[0] + x + 1] += wx * (1 - wy) * mean_loss;
			e[(y + 1) * error_map_res[0] + x] += (1 - wx) * wy * mean_loss;
			e[(y + 1) * error_map_res[0] + x + 1] += wx * wy * mean_loss;
		}
		// testbed_nerf.cu:1027-1029: the depth image holds distances along the UNNORMALISED ray direction
		const float target_depth = length(V3(rays_in[i].d)) * ((o.depth_lambda > 0.0f && m.depth) ? read_depth(uv, m.resolution, m.depth) : -1.0f);
		const LossAndGradient lg_depth = loss_and_gradient(V3(target_depth), V3(depth_ray), o.depth_loss_type);
		const float depth_loss_gradient = target_depth > 0.0f ? o.depth_lambda * lg_depth.gradient.x : 0.f;

		float loss_scale = o.loss_scale / n_rays;
		const float output_l2_reg = o.rgb_act == NGP_ACT_EXPONENTIAL ? 1e-4f : 0.0f;
		// the fused kernel (the only one with Rfl modes in the reference) has this regulariser switched off (train_nerf.cuh:307)
		const float output_l1_reg_density = (o.train_mode == 0 && mean_density < NERF_MIN_OPTICAL_THICKNESS) ? 1e-4f : 0.0f;

		vec3 rgb_ray2 = V3(0.f), loss_bg2 = V3(0.f);
		float depth_ray2 = 0.f;
		T = 1.f;
		for (uint32_t j = 0; j < compacted_numsteps; ++j) {
			for (int k = 0; k < 7; ++k) cout[(size_t)j * 7 + k] = cin[(size_t)j * 7 + k];
			const vec3 pos = unwarp_position(V3(cin + (size_t)j * 7), aabb);
			float depth = distance(pos, ray_o);
			float dt = unwarp_dt(cin[(size_t)j * 7 + 3]);
			const uint16_t* lo = no + (size_t)j * out_stride;
			float l0 = h2f(lo[0]), l1 = h2f(lo[1]), l2 = h2f(lo[2]), l3 = h2f(lo[3]);
			const vec3 rgb = {network_to_rgb(l0, o.rgb_act), network_to_rgb(l1, o.rgb_act), network_to_rgb(l2, o.rgb_act)};
			const float density = network_to_density(l3, o.density_act);
			const float alpha = 1.f - std::exp(-density * dt);
			const float weight = alpha * T;
			rgb_ray2 += weight * rgb;
			depth_ray2 += weight * depth;
			T *= (1.f - alpha);
			const vec3 suffix = rgb_ray - rgb_ray2;
			vec3 dloss_by_drgb = weight * lg.gradient;
			float density_derivative = network_to_density_derivative(l3, o.density_act);
			const float depth_suffix = depth_ray - depth_ray2;
			const float depth_supervision = depth_loss_gradient * (T * depth - depth_suffix); // :1126-1127
			float dloss_by_dmlp = density_derivative * (dt * (dot(lg.gradient, T * rgb - suffix) + depth_supervision));
			if (o.train_mode == 1) { // radiance field loss, train_nerf.cuh:391-396
				LossAndGradient local_lg = loss_and_gradient(rgbtarget, rgb, o.loss_type);
				loss_bg2 += weight * local_lg.loss;
				dloss_by_drgb = weight * local_lg.gradient;
				const vec3 v = T * local_lg.loss - (loss_bg - loss_bg2);
				dloss_by_dmlp = density_derivative * (dt * (v.x + v.y + v.z));
			} else if (o.train_mode == 2) { // relaxation between volume and surface reconstruction, train_nerf.cuh:397-405
				const vec3 rgb_bg = suffix / std::fmax(1e-6f, T);
				const vec3 rgb_lerp = (1 - alpha) * rgb_bg + alpha * rgb;
				LossAndGradient local_lg = loss_and_gradient(rgbtarget, rgb_lerp, o.loss_type);
				dloss_by_drgb = weight * local_lg.gradient;
				dloss_by_dmlp = density_derivative * (dt * (dot(local_lg.gradient, T * rgb - suffix) + 0.0f));
			}
			float d0 = loss_scale * (dloss_by_drgb.x * network_to_rgb_derivative(l0, o.rgb_act) + std::fmax(0.0f, output_l2_reg * l0));
			float d1 = loss_scale * (dloss_by_drgb.y * network_to_rgb_derivative(l1, o.rgb_act) + std::fmax(0.0f, output_l2_reg * l1));
			float d2 = loss_scale * (dloss_by_drgb.z * network_to_rgb_derivative(l2, o.rgb_act) + std::fmax(0.0f, output_l2_reg * l2));
			float d3 = loss_scale * dloss_by_dmlp + (l3 < 0.0f ? -output_l1_reg_density : 0.0f) + (l3 > -10.0f && depth < o.near_distance ? 1e-4f : 0.0f);
			uint16_t* d = dl + (size_t)j * dl_stride;
			d[0] = f2h(d0); d[1] = f2h(d1); d[2] = f2h(d2); d[3] = f2h(d3);
		}
	}
	return counter;
}

// -------------------------------------------------------------------------------------------------
// occupancy grid, testbed_nerf.cu:87-396, 2476-2633
// -------------------------------------------------------------------------------------------------
// pos_to_uv, common_device.cuh:527-577 (f-theta has no forward mapping: the reference asserts in debug builds and falls through to the
// undistorted perspective mapping otherwise).  The reference returns foveation.unwarp(uv) (:576); with the default Foveation this path uses ({} at testbed_nerf.cu:147)
// that is (clamp(y, 0, 1) - 0) / 1 per axis (common_device.cuh:226-235): the result is CLAMPED to the unit square.
inline vec2 pos_to_uv_before_foveation(vec3 pos, const int32_t res[2], const float focal[2], const mat4x3& cam, const float screen_center[2],
		int lens_mode, const float* lens_params) {
	vec3 dir = pos - cam[3];
	// inverse(mat3(cam)) * dir via cofactors
	const vec3 &a = cam.c[0], &b = cam.c[1], &c = cam.c[2];
	float det = a.x * (b.y * c.z - c.y * b.z) - b.x * (a.y * c.z - c.y * a.z) + c.x * (a.y * b.z - b.y * a.z);
	float id = 1.0f / det;
	vec3 r0 = {(b.y * c.z - c.y * b.z) * id, -(b.x * c.z - c.x * b.z) * id, (b.x * c.y - c.x * b.y) * id};
	vec3 r1 = {-(a.y * c.z - c.y * a.z) * id, (a.x * c.z - c.x * a.z) * id, -(a.x * c.y - c.x * a.y) * id};
	vec3 r2 = {(a.y * b.z - b.y * a.z) * id, -(a.x * b.z - b.x * a.z) * id, (a.x * b.y - b.x * a.y) * id};
	dir = {dot(r0, dir), dot(r1, dir), dot(r2, dir)};
	if (lens_mode == NGP_LENS_ORTHOGRAPHIC) return {dir.x * focal[0] / (float)res[0] + screen_center[0], dir.y * focal[1] / (float)res[1] + screen_center[1]};
	if (lens_mode == NGP_LENS_LATLONG || lens_mode == NGP_LENS_EQUIRECTANGULAR) { // lens.is_360(): normalise by the length
		dir /= std::sqrt(dot(dir, dir));
		return lens_mode == NGP_LENS_EQUIRECTANGULAR ? dir_to_equirectangular(dir) : dir_to_latlong(dir);
	}
	dir /= dir.z;
	float du = 0.f, dv = 0.f;
	if (lens_mode == NGP_LENS_OPENCV) opencv_lens_distortion_delta(lens_params, dir.x, dir.y, &du, &dv);
	else if (lens_mode == NGP_LENS_OPENCV_FISHEYE) opencv_fisheye_lens_distortion_delta(lens_params, dir.x, dir.y, &du, &dv);
	dir.x += du; dir.y += dv;
	return {dir.x * focal[0] / (float)res[0] + screen_center[0], dir.y * focal[1] / (float)res[1] + screen_center[1]};
}

inline float default_foveation_unwarp(float y) { y = y < 0.0f ? 0.0f : (y > 1.0f ? 1.0f : y); return (y - 0.0f) / 1.0f; }
inline vec2 pos_to_uv(vec3 pos, const int32_t res[2], const float focal[2], const mat4x3& cam, const float screen_center[2], int lens_mode, const float* lens_params) {
	const vec2 uv = pos_to_uv_before_foveation(pos, res, focal, cam, screen_center, lens_mode, lens_params);
	return {default_foveation_unwarp(uv.x), default_foveation_unwarp(uv.y)};
}

// mark_untrained_density_grid, testbed_nerf.cu:87-162
inline void mark_untrained_density_grid(uint32_t n_elements, float* grid, uint32_t n_images, const ngp_image_meta* meta,
		const ngp_xform* xforms, bool clear_visible_voxels) {
	#pragma omp parallel for schedule(static)
	for (int64_t ii = 0; ii < (int64_t)n_elements; ++ii) {
		uint32_t i = (uint32_t)ii;
		uint32_t level = i / NERF_GRID_N_CELLS, pos_idx = i % NERF_GRID_N_CELLS;
		uint32_t x = morton3D_invert(pos_idx >> 0), y = morton3D_invert(pos_idx >> 1), z = morton3D_invert(pos_idx >> 2);
		float voxel_size = std::scalbn(1.0f / NERF_GRIDSIZE, (int)level);
		vec3 pos = (vec3{(float)x, (float)y, (float)z} / (float)NERF_GRIDSIZE - 0.5f) * std::scalbn(1.0f, (int)level) + 0.5f;
		vec3 corners[8];
		for (int k = 0; k < 8; ++k) corners[k] = pos + vec3{(k & 1) ? voxel_size : 0.f, (k & 2) ? voxel_size : 0.f, (k & 4) ? voxel_size : 0.f};
		const uint32_t min_count = 1;
		uint32_t count = 0;
		for (uint32_t j = 0; j < n_images && count < min_count; ++j) {
			const mat4x3 xf = M43(xforms[j].start);
			const ngp_image_meta& m = meta[j];
			// testbed_nerf.cu:131-136: f-theta lenses have no forward mapping and are assumed to see everything; 360 lenses do
			if (m.lens_mode == NGP_LENS_FTHETA || m.lens_mode == NGP_LENS_LATLONG || m.lens_mode == NGP_LENS_EQUIRECTANGULAR) { ++count; continue; }
			for (uint32_t k = 0; k < 8; ++k) {
				vec3 dir = normalize(corners[k] - xf[3]);
				if (dot(dir, xf[2]) < 1e-4f) continue;
				vec2 uv = pos_to_uv(corners[k], m.resolution, m.focal_length, xf, m.principal_point, m.lens_mode, m.lens_params);
				vec3 ro, rd;
				uv_to_ray(uv, m.resolution, m.focal_length, xf, m.principal_point, m.lens_mode, m.lens_params, 0.0f, ro, rd);
				if (distance(normalize(rd), dir) < 1e-3f && uv.x > 0.0f && uv.y > 0.0f && uv.x < 1.0f && uv.y < 1.0f) { ++count; break; }
			}
		}
		if (clear_visible_voxels || (grid[i] < 0) != (count < min_count)) grid[i] = (count >= min_count) ? 0.f : -1.f;
	}
}

// generate_grid_samples_nerf_nonuniform, testbed_nerf.cu:216-257
inline void generate_grid_samples_nonuniform(uint32_t n_elements, const Pcg32& rng_in, uint32_t step, const Aabb& aabb, const float* grid_in,
		float* out_pos /* 3 floats */, uint32_t* indices, uint32_t n_cascades, float thresh) {
	#pragma omp parallel for schedule(static)
	for (int64_t ii = 0; ii < (int64_t)n_elements; ++ii) {
		uint32_t i = (uint32_t)ii;
		Pcg32 rng = rng_in;
		rng.advance((int64_t)(i * 4u));
		uint32_t level = (uint32_t)(rng.next_float() * n_cascades) % n_cascades;
		uint32_t idx = 0;
		for (uint32_t j = 0; j < 10; ++j) {
			idx = ((i + step * n_elements) * 56924617u + j * 19349663u + 96925573u) % NERF_GRID_N_CELLS;
			idx += level * NERF_GRID_N_CELLS;
			if (grid_in[idx] > thresh) break;
		}
		uint32_t pos_idx = idx % NERF_GRID_N_CELLS;
		uint32_t x = morton3D_invert(pos_idx >> 0), y = morton3D_invert(pos_idx >> 1), z = morton3D_invert(pos_idx >> 2);
		vec3 r; r.x = rng.next_float(); r.y = rng.next_float(); r.z = rng.next_float();
		vec3 pos = ((vec3{(float)x, (float)y, (float)z} + r) / (float)NERF_GRIDSIZE - 0.5f) * std::scalbn(1.0f, (int)level) + 0.5f;
		vec3 wp = warp_position(pos, aabb);
		out_pos[(size_t)i * 3 + 0] = wp.x; out_pos[(size_t)i * 3 + 1] = wp.y; out_pos[(size_t)i * 3 + 2] = wp.z;
		indices[i] = idx;
	}
}

// splat_grid_samples_nerf_max_nearest_neighbor, testbed_nerf.cu:259-284 (atomicMax on uint bits of a
// non-negative float == float max)
inline void splat_grid_samples(uint32_t n, const uint32_t* indices, const uint16_t* net_out, uint32_t stride, float* grid_out, int density_act) {
	for (uint32_t i = 0; i < n; ++i) {
		float mlp = network_to_density(h2f(net_out[(size_t)i * stride]), density_act);
		float thick = mlp * std::scalbn(MIN_CONE_STEPSIZE, 0);
		uint32_t a, b; std::memcpy(&a, &grid_out[indices[i]], 4); std::memcpy(&b, &thick, 4);
		if (b > a) std::memcpy(&grid_out[indices[i]], &b, 4);
	}
}
// ema_grid_samples_nerf, testbed_nerf.cu:316-338
inline void ema_grid_samples(uint32_t n, float decay, float* grid_out, const float* grid_in) {
	for (uint32_t i = 0; i < n; ++i) {
		float prev = grid_out[i];
		grid_out[i] = (prev < 0.f) ? prev : std::fmax(prev * decay, grid_in[i]);
	}
}
// reduce_sum lambda, testbed_nerf.cu:2602-2608 (summation order differs from the device; compared with tolerance)
inline float density_grid_mean(const float* grid) {
	double s = 0;
	for (uint32_t i = 0; i < NERF_GRID_N_CELLS; ++i) s += (double)(std::fmax(grid[i], 0.f) / (float)NERF_GRID_N_CELLS);
	return (float)s;
}
// grid_to_bitfield :348-374 + bitfield_max_pool :376-396 (launch loop :2621-2630)
inline void grid_to_bitfield_and_pool(const float* grid, uint32_t max_cascade, uint8_t* bitfield, float mean_density) {
	const uint32_t n_elements = NERF_GRID_N_CELLS / 8 * NERF_CASCADES;
	const uint32_t n_nonzero = NERF_GRID_N_CELLS / 8 * (max_cascade + 1);
	float thresh = std::min(NERF_MIN_OPTICAL_THICKNESS, mean_density);
	for (uint32_t i = 0; i < n_elements; ++i) {
		if (i >= n_nonzero) { bitfield[i] = 0; continue; }
		uint8_t bits = 0;
		for (uint8_t j = 0; j < 8; ++j) bits |= grid[(size_t)i * 8 + j] > thresh ? ((uint8_t)1 << j) : 0;
		bitfield[i] = bits;
	}
	for (uint32_t level = 1; level < NERF_CASCADES; ++level) {
		const uint8_t* prev = bitfield + grid_mip_offset(level - 1) / 8;
		uint8_t* next = bitfield + grid_mip_offset(level) / 8;
		for (uint32_t i = 0; i < NERF_GRID_N_CELLS / 64; ++i) {
			uint8_t bits = 0;
			for (uint8_t j = 0; j < 8; ++j) bits |= prev[(size_t)i * 8 + j] > 0 ? ((uint8_t)1 << j) : 0;
			uint32_t x = morton3D_invert(i >> 0) + NERF_GRIDSIZE / 8;
			uint32_t y = morton3D_invert(i >> 1) + NERF_GRIDSIZE / 8;
			uint32_t z = morton3D_invert(i >> 2) + NERF_GRIDSIZE / 8;
			next[morton3D(x, y, z)] |= bits;
		}
	}
}

// -------------------------------------------------------------------------------------------------
// Trainer = the NeRF half of ngp::Testbed (train loop: testbed.cu:4561-4647; train_nerf
// testbed_nerf.cu:2704-2790; train_nerf_step :3007-3382; training_prep_nerf :3385-3398;
// NerfCounters :2669-2702; reset_network rng plumbing testbed.cu:4163-4178).
// -------------------------------------------------------------------------------------------------
struct NerfTrainer {
	Model* model;
	ngp_nerf_options opt;
	Aabb aabb;
	std::vector<ngp_image_meta> meta;
	std::vector<ngp_xform> xforms;
	std::vector<float> density_grid;      // 128^3 * (max_cascade+1)
	std::vector<uint8_t> bitfield;        // 128^3/8 * 8
	float mean_density = 0.f;
	uint32_t ema_step = 0;
	Pcg32 rng, density_grid_rng;
	uint32_t training_step = 0;
	uint32_t rays_per_batch = 1u << 12, n_rays_total = 0, n_images_marked = 0;
	uint32_t measured_batch_size = 0, measured_batch_size_before_compaction = 0;
	uint32_t n_rays_last = 0;
	float loss_scalar = 0.f;
	uint64_t total_rays = 0, total_samples = 0;
	// scratch kept for inspection by tests
	std::vector<uint32_t> ray_indices, numsteps;
	std::vector<ngp_ray> rays;
	std::vector<float> coords, coords_compacted, loss;
	std::vector<uint16_t> mlp_out, dloss;

	// error map (testbed.h:745-756, 810-815): accumulated on every step like the reference (testbed_nerf.cu:2793)
	std::vector<float> error_map, cdf_x_cond_y, cdf_y, cdf_img;
	int32_t error_map_res[2] = {0, 0}, cdf_res[2] = {0, 0};
	bool is_cdf_valid = false;
	uint32_t n_steps_between_error_map_updates = 128, n_steps_since_error_map_update = 0;
	ErrorCdf cdf_args() const {
		ErrorCdf c;
		if (!is_cdf_valid) return c;
		if (opt.sample_focal_plane_proportional_to_error) { c.x_cond_y = cdf_x_cond_y.data(); c.y = cdf_y.data(); }
		if (opt.sample_image_proportional_to_error) c.img = cdf_img.data();
		c.res[0] = cdf_res[0]; c.res[1] = cdf_res[1];
		return c;
	}

	NerfTrainer(Model* m, const ngp_nerf_options& o, const ngp_aabb& box) : model(m), opt(o), aabb(box) {
		rng = Pcg32(o.seed);                         // testbed.cu:4163
		density_grid_rng = Pcg32(rng.next_uint());   // testbed.cu:4178
		density_grid.assign((size_t)NERF_GRID_N_CELLS * (o.max_cascade + 1), 0.f);
		bitfield.assign((size_t)NERF_GRID_N_CELLS / 8 * NERF_CASCADES, 0);
	}
	void set_dataset(uint32_t n, const ngp_image_meta* m, const ngp_xform* x) { meta.assign(m, m + n); xforms.assign(x, x + n); }

	void update_mean_and_bitfield() {
		mean_density = density_grid_mean(density_grid.data());
		grid_to_bitfield_and_pool(density_grid.data(), opt.max_cascade, bitfield.data(), mean_density);
	}

	// update_density_grid_nerf, testbed_nerf.cu:2476-2592
	void update_density_grid(float decay, uint32_t n_uniform, uint32_t n_nonuniform) {
		const uint32_t n_elements = NERF_GRID_N_CELLS * (opt.max_cascade + 1);
		const uint32_t n_samples = n_uniform + n_nonuniform;
		if (training_step == 0 || (uint32_t)meta.size() != n_images_marked) { // testbed_nerf.cu:2500-2517: again whenever the number of training images changed
			n_images_marked = (uint32_t)meta.size();
			if (training_step == 0) ema_step = 0;
			mark_untrained_density_grid(n_elements, density_grid.data(), (uint32_t)meta.size(), meta.data(), xforms.data(), training_step == 0);
		}
		std::vector<float> positions((size_t)n_samples * 3), tmp(n_elements, 0.f);
		std::vector<uint32_t> indices(n_samples);
		std::vector<uint16_t> out(n_samples);
		generate_grid_samples_nonuniform(n_uniform, density_grid_rng, ema_step, aabb, density_grid.data(), positions.data(), indices.data(), opt.max_cascade + 1, -0.01f);
		density_grid_rng.advance();
		generate_grid_samples_nonuniform(n_nonuniform, density_grid_rng, ema_step, aabb, density_grid.data(), positions.data() + (size_t)n_uniform * 3,
			indices.data() + n_uniform, opt.max_cascade + 1, NERF_MIN_OPTICAL_THICKNESS);
		density_grid_rng.advance();
		model->density(positions.data(), 3, n_samples, out.data(), 1, false);
		splat_grid_samples(n_samples, indices.data(), out.data(), 1, tmp.data(), opt.density_activation);
		ema_grid_samples(n_elements, decay, density_grid.data(), tmp.data());
		++ema_step;
		update_mean_and_bitfield();
	}

	// training_prep_nerf, testbed_nerf.cu:3385-3398
	void training_prep() {
		uint32_t n_cascades = opt.max_cascade + 1;
		if (training_step < 256) update_density_grid(opt.density_grid_decay, NERF_GRID_N_CELLS * n_cascades, 0);
		else update_density_grid(opt.density_grid_decay, NERF_GRID_N_CELLS / 4 * n_cascades, NERF_GRID_N_CELLS / 4 * n_cascades);
	}

	// train_nerf_step (forward + backward), testbed_nerf.cu:3007-3382
	void forward_backward() {
		const uint32_t B = opt.target_batch_size;
		const uint32_t max_samples = B * 16;
		uint32_t max_inference;
		if (measured_batch_size_before_compaction == 0) { measured_batch_size_before_compaction = max_inference = max_samples; }
		else max_inference = next_multiple(std::min(measured_batch_size_before_compaction, max_samples), 256u);
		const uint32_t R = rays_per_batch;
		if (n_steps_since_error_map_update == 0 && !meta.empty()) { // testbed_nerf.cu:2753-2759
			uint32_t n_samples_per_image = (n_steps_between_error_map_updates * rays_per_batch) / (uint32_t)meta.size();
			int r = (int)(std::sqrt(std::sqrt((float)n_samples_per_image)) * 3.5f);
			error_map_res[0] = std::min(r, meta[0].resolution[0]); error_map_res[1] = std::min(r, meta[0].resolution[1]);
			error_map.assign((size_t)error_map_res[0] * error_map_res[1] * meta.size(), 0.f);
		}
		const ErrorCdf cdf = cdf_args();
		if (training_step == 0) n_rays_total = 0;
		n_rays_total += R;
		ray_indices.assign(R, 0); rays.assign(R, ngp_ray{}); numsteps.assign((size_t)R * 2, 0);
		coords.assign((size_t)max_inference * 7, 0.f); mlp_out.assign((size_t)max_inference * 4, 0);
		coords_compacted.assign((size_t)B * 7, 0.f); dloss.assign((size_t)B * 4, 0); loss.assign(R, 0.f);
		uint32_t rb = (uint32_t)((uint64_t)R * opt.rank / std::max(1u, opt.world_size));
		uint32_t re = (uint32_t)((uint64_t)R * (opt.rank + 1) / std::max(1u, opt.world_size));
		K1Out k1 = generate_training_samples(R, rb, re, aabb, max_inference, rng, ray_indices.data(), rays.data(), numsteps.data(), coords.data(),
			(uint32_t)meta.size(), meta.data(), xforms.data(), bitfield.data(), opt.max_cascade, opt.snap_to_pixel_centers, opt.cone_angle_constant, &cdf);
		uint32_t n_inf = std::min(k1.numsteps_counter, max_inference);
		model->inference(coords.data(), 7, n_inf, mlp_out.data(), 4, false);
		K3Opts ko;
		ko.loss_scale = opt.loss_scale; ko.background_color = V3(opt.background_color); ko.color_space_srgb = opt.color_space_srgb;
		ko.random_bg = opt.random_bg_color; ko.linear_colors = opt.linear_colors; ko.snap = opt.snap_to_pixel_centers;
		ko.loss_type = opt.loss_type; ko.rgb_act = opt.rgb_activation; ko.density_act = opt.density_activation; ko.near_distance = opt.near_distance;
		ko.train_mode = opt.train_mode; ko.depth_lambda = opt.depth_supervision_lambda; ko.depth_loss_type = opt.depth_loss_type;
		uint32_t compacted = compute_loss(R, k1.ray_counter, aabb, rng, B, ko, (uint32_t)meta.size(), meta.data(), mlp_out.data(), 4,
			ray_indices.data(), rays.data(), numsteps.data(), coords.data(), coords_compacted.data(), dloss.data(), 4, loss.data(), mean_density, &cdf, error_map.data(), error_map_res);
		n_rays_last = k1.ray_counter;
		counter_before = k1.numsteps_counter; counter_compacted = compacted;
		uint32_t n_valid = std::min(compacted, B);
		fill_rollover_and_rescale_h(B, 4, n_valid, dloss.data());
		fill_rollover_f(B, 7, n_valid, coords_compacted.data());
		model->training_step(coords_compacted.data(), 7, B, dloss.data(), 4);
		rng.advance();
	}
	uint32_t counter_before = 0, counter_compacted = 0;

	// optimizer_step + NerfCounters::update_after_training, testbed_nerf.cu:2770-2778, 2678-2702
	void finish() {
		model->optimizer_step(opt.loss_scale);
		++training_step;
		total_rays += rays_per_batch;
		struct AtExit { NerfTrainer* t; ~AtExit() { t->update_error_cdfs(); } } at_exit{this}; // testbed_nerf.cu:2791-2855 follows the counter update
		measured_batch_size = 0; measured_batch_size_before_compaction = 0;
		if (counter_before == 0 || counter_compacted == 0) { loss_scalar = 0.f; return; }
		measured_batch_size_before_compaction = counter_before;
		measured_batch_size = counter_compacted;
		total_samples += measured_batch_size;
		double s = 0; for (float l : loss) s += l;
		loss_scalar = (float)s * (float)measured_batch_size / (float)opt.target_batch_size;
		rays_per_batch = (uint32_t)((float)rays_per_batch * (float)opt.target_batch_size / (float)measured_batch_size);
		rays_per_batch = std::min(next_multiple(rays_per_batch, 256u), 1u << 18);
	}

	void update_error_cdfs() {
		n_steps_since_error_map_update += 1;
		if (n_steps_since_error_map_update < n_steps_between_error_map_updates) return;
		cdf_res[0] = error_map_res[0]; cdf_res[1] = error_map_res[1];
		const size_t n_img = meta.size();
		cdf_x_cond_y.assign((size_t)cdf_res[0] * cdf_res[1] * n_img, 0.f); cdf_y.assign((size_t)cdf_res[1] * n_img, 0.f); cdf_img.assign(n_img, 0.f);
		construct_error_cdfs((uint32_t)n_img, (uint32_t)cdf_res[0], (uint32_t)cdf_res[1], error_map.data(), cdf_x_cond_y.data(), cdf_y.data(), cdf_img.data());
		n_steps_since_error_map_update = 0;
		is_cdf_valid = true;
		n_steps_between_error_map_updates = (uint32_t)(n_steps_between_error_map_updates * 1.5f);
	}

	// Testbed::train, testbed.cu:4561-4647 (one call = one optimizer step)
	uint32_t training_prep_skip_counter = 0;
	void train_step() {
		uint32_t n_prep_to_skip = (uint32_t)clampi((int)training_step / 16, 1, 16);
		if (training_prep_skip_counter % n_prep_to_skip == 0) training_prep();
		++training_prep_skip_counter;
		forward_backward();
		finish();
	}

	// fused per-pixel renderer, fused_kernels/render_nerf.cuh:22-184 (Shade mode, no envmap / DoF /
	// foveation; render_aabb_to_local = identity).
	void render(const ngp_render_params& rp, float* frame, float* depth) const {
		const int W = rp.resolution[0], H = rp.resolution[1];
		Aabb render_aabb(rp.render_aabb);
		const mat4x3 cam = M43(rp.camera);
		#pragma omp parallel for schedule(dynamic, 64)
		for (int64_t idx64 = 0; idx64 < (int64_t)W * H; ++idx64) {
			uint32_t idx = (uint32_t)idx64, x = idx % W, y = idx / W;
			vec2 off = ld_random_pixel_offset(rp.snap_to_pixel_centers ? 0 : rp.spp_index);
			vec2 uv = {((float)x + off.x) / (float)W, ((float)y + off.y) / (float)H};
			vec3 ro, rd;
			const bool has_ray = uv_to_ray(uv, rp.resolution, rp.focal_length, cam, rp.screen_center, rp.lens_mode, rp.lens_params, rp.near_distance, ro, rd);
			if (!has_ray) rd = cam[2];
			rd = normalize(rd);
			float t = std::fmax(render_aabb.ray_intersect(ro, rd).x, 0.0f) + 1e-6f;
			bool alive = has_ray && render_aabb.contains(ro + rd * t);
			vec3 idir = V3(1.0f) / rd;
			float color[4] = {0, 0, 0, 0};
			vec3 cam_fwd = cam[2], cam_pos = cam[3];
			float best_depth = MAX_DEPTH, max_weight = 0.f;
			float cone_angle = opt.cone_angle_constant;
			t = advance_n_steps(t, cone_angle, ld_random_val(rp.spp_index, idx * 786433u));
			while (alive) {
				t = if_unoccupied_advance_to_next_occupied_voxel(t, cone_angle, ro, rd, idir, bitfield.data(), 0, opt.max_cascade, render_aabb);
				if (t >= MAX_DEPTH) break;
				vec3 pos = ro + rd * t;
				float dt = calc_dt(t, cone_angle);
				vec3 wp = warp_position(pos, aabb), wd = warp_direction(rd);
				float c[7] = {wp.x, wp.y, wp.z, warp_dt(dt), wd.x, wd.y, wd.z};
				uint16_t o4[4];
				model->eval(c, rp.use_inference_params != 0, o4);
				t += dt;
				float alpha = 1.f - std::exp(-network_to_density(h2f(o4[3]), opt.density_activation) * dt);
				float weight = alpha * (1.0f - color[3]);
				vec3 rgb = {network_to_rgb(h2f(o4[0]), opt.rgb_activation), network_to_rgb(h2f(o4[1]), opt.rgb_activation), network_to_rgb(h2f(o4[2]), opt.rgb_activation)};
				color[0] += rgb.x * weight; color[1] += rgb.y * weight; color[2] += rgb.z * weight; color[3] += weight;
				if (weight > max_weight) { max_weight = weight; best_depth = dot(cam_fwd, pos - cam_pos); }
				if (color[3] > (1.0f - rp.min_transmittance)) {
					float a = color[3];
					for (int k = 0; k < 4; ++k) color[k] /= a;
					break;
				}
			}
			if (!opt.linear_colors) { color[0] = srgb_to_linear(color[0]); color[1] = srgb_to_linear(color[1]); color[2] = srgb_to_linear(color[2]); }
			depth[idx] = color[3] > 0.2f ? best_depth : MAX_DEPTH;
			for (int k = 0; k < 4; ++k) frame[(size_t)idx * 4 + k] = color[k];
		}
	}
};

} // namespace ora
