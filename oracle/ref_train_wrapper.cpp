// oracle/_ref/libngptrain_ref.so -- TEST INFRASTRUCTURE ONLY.  The reference's fused training kernel (include/neural-graphics-primitives/fused_kernels/train_nerf.cuh: ray
// generation, march, network, compositing, loss and dL/doutput in one kernel; the only place where the Rfl / RflRelax training modes are written down) compiled for the CPU
// from where it lies, against oracle/ref_shim, with the caller's network as its eval_nerf (see ref_render_wrapper.cpp).  Rays run one at a time in order.
// tests/test_ref_kernels.py feeds the samples this kernel chose to the oracle's K3 and compares the losses and dL/doutput.
#include <tiny-cuda-nn/common.h>
#include <neural-graphics-primitives/common.h>
#include "../include/ngp_hip.h"
#include <vector>
static constexpr uint32_t N_EXTRA_DIMS = 0;
#define __all_sync(mask, predicate) (predicate)
typedef void (*ref_infer_fn)(void* model, const float* in, uint32_t in_stride, uint32_t n, uint16_t* out, uint32_t out_stride, int use_inference_params);
static ref_infer_fn g_infer = nullptr; static void* g_model = nullptr;
static inline tcnn::vec4 eval_nerf(const tcnn::vec<7>& in, const tcnn::network_precision_t*) {
	uint16_t o[4]; g_infer(g_model, &in[0], 7, 1, o, 4, 0);
	__half h[4]; for (int k = 0; k < 4; ++k) h[k].bits = o[k];
	return {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
#include <neural-graphics-primitives/nerf_device.cuh>
#include <neural-graphics-primitives/envmap.cuh>
using namespace ngp;
#include <neural-graphics-primitives/fused_kernels/train_nerf.cuh>

#define REF extern "C" __attribute__((visibility("default")))
static vec3 V3(const float* p) { return {p[0], p[1], p[2]}; }
REF void ref_fused_train_nerf(uint32_t n_rays, ngp_aabb aabb, uint32_t max_samples, ngp_pcg32 rng, uint32_t n_images, const ngp_image_meta* meta, const ngp_xform* xforms,
		const uint8_t* bitfield, uint32_t max_mip, int snap, float cone_angle_constant, float loss_scale, const float* background_color, int color_space_srgb, int random_bg,
		int linear_colors, int loss_type, int rgb_act, int density_act, float mean_density, float near_distance, int train_mode, ref_infer_fn infer, void* model,
		uint32_t* ray_counter, uint32_t* numsteps_counter, uint32_t* ray_indices_out, uint32_t* numsteps_out, float* coords_out, uint16_t* dloss, float* loss_output) {
	g_infer = infer; g_model = model;
	std::vector<TrainingImageMetadata> m(n_images); std::vector<TrainingXForm> x(n_images);
	for (uint32_t i = 0; i < n_images; ++i) {
		m[i].pixels = meta[i].pixels; m[i].image_data_type = (EImageDataType)meta[i].image_data_type; m[i].depth = meta[i].depth;
		m[i].lens.mode = (ELensMode)meta[i].lens_mode; for (int k = 0; k < 7; ++k) m[i].lens.params[k] = meta[i].lens_params[k];
		m[i].resolution = ivec2{meta[i].resolution[0], meta[i].resolution[1]}; m[i].principal_point = vec2{meta[i].principal_point[0], meta[i].principal_point[1]};
		m[i].focal_length = vec2{meta[i].focal_length[0], meta[i].focal_length[1]};
		x[i].start = mat4x3{V3(xforms[i].start), V3(xforms[i].start + 3), V3(xforms[i].start + 6), V3(xforms[i].start + 9)}; x[i].end = x[i].start;
	}
	pcg32 g; g.state = rng.state; g.inc = rng.inc;
	std::vector<vec3> exposure(n_images, vec3(0.0f));
	std::vector<Ray> rays(n_rays);
	*ray_counter = 0; *numsteps_counter = 0;
	for (uint32_t i = 0; i < n_rays; ++i) {
		blockIdx.x = i;
		train_nerf(n_rays, BoundingBox{V3(aabb.min), V3(aabb.max)}, max_samples, 0u, g, ray_counter, numsteps_counter, ray_indices_out, rays.data(), numsteps_out,
			PitchedPtr<NerfCoordinate>((NerfCoordinate*)coords_out, 1, 0, 0), n_images, m.data(), x.data(), bitfield, loss_output, false, nullptr, max_mip, snap != 0, false,
			cone_angle_constant, Buffer2DView<const vec2>{}, nullptr, nullptr, nullptr, ivec2{0, 0}, nullptr, nullptr, (ENerfActivation)density_act, (ENerfActivation)rgb_act,
			loss_scale, 4, Buffer2DView<const vec4>{}, nullptr, ivec2{0, 0}, ELossType::L2, V3(background_color), color_space_srgb ? EColorSpace::SRGB : EColorSpace::Linear,
			random_bg != 0, linear_colors != 0, (network_precision_t*)dloss, (ELossType)loss_type, ELossType::L1, nullptr, ivec2{0, 0}, nullptr, ivec2{0, 0}, nullptr,
			&mean_density, exposure.data(), nullptr, 0.0f, near_distance, (ETrainMode)train_mode);
	}
	blockIdx.x = 0;
}
