// oracle/_ref/libngprb_ref.so, part 3 (see ref_renderbuffer_pre.hpp) -- TEST INFRASTRUCTURE ONLY
} // namespace ngp
using namespace ngp;
#define REF extern "C" __attribute__((visibility("default")))
// CudaRenderBuffer::accumulate / tonemap for a Linear render colour space (render_buffer.cu:613-676): n pixels in a row
REF void ref_accumulate(uint32_t n, float* frame_rgba, float* accumulate_rgba, float sample_count, int color_space_srgb) {
	for (uint32_t i = 0; i < n; ++i) { blockIdx.x = i; blockIdx.y = 0; accumulate_kernel(ivec2{(int)n, 1}, (vec4*)frame_rgba, (vec4*)accumulate_rgba, sample_count, color_space_srgb ? EColorSpace::SRGB : EColorSpace::Linear); }
	blockIdx.x = 0;
}
REF void ref_tonemap(uint32_t n, float exposure, const float* background_srgb4, float* accumulate_rgba, int output_srgb, int curve, int clamp_output_color, int unmultiply_alpha, float* out_rgba) {
	ngp_shim_surface s{(float4*)out_rgba, n};
	for (uint32_t i = 0; i < n; ++i) {
		blockIdx.x = i; blockIdx.y = 0;
		tonemap_kernel(ivec2{(int)n, 1}, exposure, vec4{background_srgb4[0], background_srgb4[1], background_srgb4[2], background_srgb4[3]}, (vec4*)accumulate_rgba, EColorSpace::Linear,
			output_srgb ? EColorSpace::SRGB : EColorSpace::Linear, (ETonemapCurve)curve, clamp_output_color != 0, unmultiply_alpha != 0, &s);
	}
	blockIdx.x = 0;
}
