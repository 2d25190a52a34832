// oracle/_ref/libngpkern_ref.so, part 1 of 3 -- TEST INFRASTRUCTURE ONLY.  The translation unit is assembled by oracle/Makefile on a pipe (nothing is written to disk):
//   this file  +  the kernel section of /root/reference/src/testbed_nerf.cu read where it lies (`namespace ngp {` up to, not including, the first Testbed member
//   function after the kernels; Testbed::network_dims_nerf() skipped)  +  ref_nerf_kernels_post.hpp (C-ABI exports with the signatures of the oracle's ora_k_* twins).
// The kernels are the reference's own text, compiled for the CPU against oracle/ref_shim and run one "thread" at a time in element order (blockIdx.x = element,
// blockDim.x = 1), so every atomicAdd hands out the slot the sequential oracle hands out.  What this pins: generate_training_samples_nerf (K1), compute_loss_kernel_train_nerf
// (K3: compositing, losses, adjoint, compaction, error map, depth supervision), the occupancy-grid kernels and the error-map CDF kernels -- every line the reference wrote,
// down to tcnn's vector arithmetic (shim, GLSL semantics) and the camera of a frame without motion (see ref_shim's slerp).
#include <neural-graphics-primitives/common.h>
#include <neural-graphics-primitives/common_device.cuh>
#include <neural-graphics-primitives/nerf_device.cuh>
#include <neural-graphics-primitives/random_val.cuh>
#include <neural-graphics-primitives/envmap.cuh>
#include "../include/ngp_hip.h"
