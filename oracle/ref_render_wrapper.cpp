// oracle/_ref/libngprender_ref.so -- TEST INFRASTRUCTURE ONLY.  The reference's fused per-pixel NeRF renderer (include/neural-graphics-primitives/fused_kernels/render_nerf.cuh,
// the kernel its JIT path compiles around the network) compiled for the CPU from where it lies, against oracle/ref_shim.  The header expects three things from the code that
// embeds it: N_EXTRA_DIMS, the network as `eval_nerf(input, params)`, and warp voting.  Here N_EXTRA_DIMS = 0, eval_nerf calls back into the caller (the test passes the
// oracle's model, whose four fp16 outputs are widened to float as the fused network's are), and __all_sync of a one-thread "warp" is its own predicate.  Pixels run one at
// a time.  tests/test_ref_kernels.py compares the oracle's render() -- what the HIP renderer is compared with on the GPU -- with it.
#include <tiny-cuda-nn/common.h>
#include <neural-graphics-primitives/common.h>
#include "../include/ngp_hip.h"
static constexpr uint32_t N_EXTRA_DIMS = 0;
#define __launch_bounds__(...)
#define __all_sync(mask, predicate) (predicate)
typedef void (*ref_infer_fn)(void* model, const float* in, uint32_t in_stride, uint32_t n, uint16_t* out, uint32_t out_stride, int use_inference_params);
static ref_infer_fn g_infer = nullptr; static void* g_model = nullptr; static int g_use_inf = 0;
static inline tcnn::vec4 eval_nerf(const tcnn::vec<7>& in, const tcnn::network_precision_t*) {
	uint16_t o[4]; g_infer(g_model, &in[0], 7, 1, o, 4, g_use_inf);
	__half h[4]; for (int k = 0; k < 4; ++k) h[k].bits = o[k];
	return {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
#include <neural-graphics-primitives/fused_kernels/render_nerf.cuh>

#define REF extern "C" __attribute__((visibility("default")))
static vec3 V3(const float* p) { return {p[0], p[1], p[2]}; }
// render_nerf as Testbed::render_nerf launches the fused kernel for a static camera (testbed_nerf.cu:1955-2010): Shade mode, no envmap / DoF / foveation / distortion map /
// hidden-area mask, render_aabb_to_local = identity, min_mip 0, no surface rendering
REF void ref_render_nerf(const ngp_render_params* rp, ngp_aabb train_aabb, const uint8_t* bitfield, uint32_t max_mip, float cone_angle_constant, int rgb_act, int density_act,
		int train_in_linear_colors, ref_infer_fn infer, void* model, float* frame, float* depth) {
	g_infer = infer; g_model = model; g_use_inf = rp->use_inference_params;
	const mat4x3 cam{V3(rp->camera), V3(rp->camera + 3), V3(rp->camera + 6), V3(rp->camera + 9)};
	Lens lens; lens.mode = (ELensMode)rp->lens_mode; for (int k = 0; k < 7; ++k) lens.params[k] = rp->lens_params[k];
	for (uint32_t y = 0; y < (uint32_t)rp->resolution[1]; ++y) for (uint32_t x = 0; x < (uint32_t)rp->resolution[0]; ++x) {
		blockIdx.x = x; blockIdx.y = y;
		render_nerf((uint32_t)rp->spp_index, ivec2{rp->resolution[0], rp->resolution[1]}, vec2{rp->focal_length[0], rp->focal_length[1]}, cam, cam, vec4(0.0f),
			vec2{rp->screen_center[0], rp->screen_center[1]}, vec3(0.0f), rp->snap_to_pixel_centers != 0, BoundingBox{V3(rp->render_aabb.min), V3(rp->render_aabb.max)}, mat3::identity(),
			rp->near_distance, 1.0f, 0.0f, Foveation{}, lens, BoundingBox{V3(train_aabb.min), V3(train_aabb.max)}, bitfield, 0u, max_mip, cone_angle_constant, Buffer2DView<const vec4>{},
			(vec4*)frame, depth, Buffer2DView<const uint8_t>{}, Buffer2DView<const vec2>{}, ERenderMode::Shade, nullptr, nullptr, (ENerfActivation)density_act, (ENerfActivation)rgb_act,
			rp->min_transmittance, train_in_linear_colors != 0, false, 1.0f);
	}
	blockIdx.x = 0; blockIdx.y = 0;
}
