// render_kernels.hip -- NeRF renderer kernels (reference testbed_nerf.cu:431-689, 1333-1528, fused_kernels/render_nerf.cuh).
// (filled in below; see ngp_nerf_render in ngp_api.hip)
#include "ngp_device.hpp"
#include "ngp_kernels.hpp"
namespace ngp {}
