// oracle/ref_shim (TEST INFRASTRUCTURE ONLY): shadows the reference's common_host.h for src/triangle_bvh.cu.  The real header declares host utilities (camera predictors on
// tcnn's se3 functions, GPU images on GPUMatrix, path helpers) that the BVH does not use and that would pull in the rest of tiny-cuda-nn; the BVH needs only what follows.
#pragma once
#include <neural-graphics-primitives/common.h>
#include <tiny-cuda-nn/gpu_memory.h>
#include <tinylogger/tinylogger.h>
#include <filesystem/path.h>
#include <string>
namespace fmt { template <typename... A> inline std::string format(const char* f, const A&...) { return f; } } // status / error strings only
namespace ngp { namespace fs = filesystem; }
