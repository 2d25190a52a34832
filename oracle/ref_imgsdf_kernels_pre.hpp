// oracle/_ref/libngpimgsdf_ref.so, part 1 -- TEST INFRASTRUCTURE ONLY.  Like ref_nerf_kernels_pre.hpp: oracle/Makefile pipes this file, selected __global__ functions of
// /root/reference/src/testbed_image.cu (stratify2_kernel, eval_image_kernel_and_snap, image_coords_from_idx, image_mse_kernel) and src/testbed_sdf.cu (perturb_sdf_samples,
// scale_to_aabb_kernel, compare_signs_kernel, assign_float, sample_discrete, sample_uniform_on_triangle_kernel) and src/nerf_loader.cu (convert_rgba32, copy_depth, sharpen) -- each read where it lies, from its first line to its closing
// brace -- and ref_imgsdf_kernels_post.hpp (C-ABI exports) into g++.  Nothing is written to disk.
#include <neural-graphics-primitives/common.h>
#include <neural-graphics-primitives/common_device.cuh>
#include <neural-graphics-primitives/random_val.cuh>
#include <neural-graphics-primitives/bounding_box.cuh>
#include <neural-graphics-primitives/triangle.cuh>
#include <neural-graphics-primitives/triangle_octree_device.cuh>
#include "../include/ngp_hip.h"
#include <vector>
namespace ngp {
