// oracle/_ref/libngpcampath_ref.so, part 3 (see ref_campath_pre.hpp) -- TEST INFRASTRUCTURE ONLY
CameraKeyframe lerp(const CameraKeyframe&, const CameraKeyframe&, float, float, float) { std::abort(); } // (declared by the header; needs tcnn's quaternion slerp; not on the spline path)
} // namespace ngp
using namespace ngp;
#define REF extern "C" __attribute__((visibility("default")))
// keyframes: n x 12 floats {R[4] (x, y, z, w), T[3], slice, scale, fov, aperture_size, timestamp}; out: camera matrix 3 x 4 row-major, then fov, scale
REF void ref_eval_camera_path(const float* keyframes, uint32_t n, int spline_order, int loop, int sanitize, float playtime, float* out14) {
	CameraPath p; p.spline_order = spline_order; p.loop = loop != 0;
	for (uint32_t i = 0; i < n; ++i) {
		const float* k = keyframes + (size_t)i * 12;
		p.keyframes.emplace_back(quat{k[0], k[1], k[2], k[3]}, vec3{k[4], k[5], k[6]}, k[7], k[8], k[9], k[10], k[11]);
	}
	if (sanitize) p.sanitize_keyframes();
	const CameraKeyframe r = p.eval_camera_path(playtime);
	const mat4x3 m = r.m();
	for (int row = 0; row < 3; ++row) for (int c = 0; c < 4; ++c) out14[row * 4 + c] = m[c][row];
	out14[12] = r.fov; out14[13] = r.scale;
}
