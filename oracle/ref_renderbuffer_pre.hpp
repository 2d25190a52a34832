// oracle/_ref/libngprb_ref.so, part 1 -- TEST INFRASTRUCTURE ONLY.  accumulate_kernel, the two tonemap() functions and tonemap_kernel of /root/reference/src/render_buffer.cu,
// picked out by ref_extract_functions.awk and compiled for the CPU on a pipe between this file and ref_renderbuffer_post.hpp (see ref_nerf_kernels_pre.hpp).  The CUDA surface
// the kernel writes to is a host array here.
#include <neural-graphics-primitives/common.h>
#include <neural-graphics-primitives/common_device.cuh>
struct float4 { float x, y, z, w; };
inline float4 to_float4(const tcnn::vec4& v) { return {v.x, v.y, v.z, v.w}; }
struct ngp_shim_surface { float4* data; uint32_t width; };
typedef ngp_shim_surface* cudaSurfaceObject_t;
inline void surf2Dwrite(float4 v, cudaSurfaceObject_t s, size_t x_bytes, uint32_t y) { s->data[(size_t)y * s->width + x_bytes / sizeof(float4)] = v; }
namespace ngp {
