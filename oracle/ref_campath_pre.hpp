// oracle/_ref/libngpcampath_ref.so, part 1 -- TEST INFRASTRUCTURE ONLY.  CameraKeyframe / CameraPath of include/neural-graphics-primitives/camera_path.h with the blends and
// the time -> segment search of src/camera_path.cu (normalize, spline_cubic, spline_quadratic, spline_linear, the timestamp helpers, CameraPath::get_pos), picked out by
// ref_extract_functions.awk and compiled for the CPU on a pipe (see ref_nerf_kernels_pre.hpp).  tests/test_pyngp.py holds host/camera_path_lite.hpp against it.
#include <algorithm>
#include <neural-graphics-primitives/camera_path.h>
namespace ngp {
