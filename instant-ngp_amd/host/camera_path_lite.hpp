// camera_path_lite.hpp -- camera paths for video rendering (reference include/neural-graphics-primitives/camera_path.h:32-193, src/camera_path.cu:31-250): the keyframe
// record, its JSON form ("path": [{R, T, slice, scale, fov, aperture_size | dof, timestamp}], "loop", "spline_order", "duration_seconds"), time -> (segment, fraction) by
// binary search over the timestamps, and the order-0..3 uniform B-spline blends with sign-aligned quaternion sums.  Headless subset: no editing kernels, no GUI.
// Quaternions are stored {x, y, z, w}, the order tcnn's JSON binding writes them in [tcnn vec_json.h, from memory: the mount holds no camera path file].
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>
#include "../csrc/mini_json.hpp"

namespace ngp_host {

struct CameraKeyframe {
	std::array<float, 4> R{0.f, 0.f, 0.f, 1.f}; // x, y, z, w
	std::array<float, 3> T{0.f, 0.f, 0.f};
	float slice = 0.f, scale = 1.f, fov = 50.625f, aperture_size = 0.f, timestamp = 0.f;

	// m(): rotation of the normalised quaternion in the first three columns, T in the fourth (column-major mat4x3)
	std::array<float, 12> m() const {
		float x = R[0], y = R[1], z = R[2], w = R[3];
		const float l = std::sqrt(x * x + y * y + z * z + w * w); x /= l; y /= l; z /= l; w /= l;
		const float xx = x * x, yy = y * y, zz = z * z, xz = x * z, xy = x * y, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
		return {1.f - 2.f * (yy + zz), 2.f * (xy + wz), 2.f * (xz - wy),
		        2.f * (xy - wz), 1.f - 2.f * (xx + zz), 2.f * (yz + wx),
		        2.f * (xz + wy), 2.f * (yz - wx), 1.f - 2.f * (xx + yy),
		        T[0], T[1], T[2]};
	}
	CameraKeyframe scaled(float f) const { CameraKeyframe k = *this; for (auto& v : k.R) v *= f; for (auto& v : k.T) v *= f; k.slice *= f; k.scale *= f; k.fov *= f; k.aperture_size *= f; return k; }
	// operator+ of the reference: the right-hand quaternion is flipped onto the left one's hemisphere; the sum takes the right-hand timestamp
	CameraKeyframe plus(const CameraKeyframe& rhs) const {
		CameraKeyframe k = *this;
		const float d = R[0] * rhs.R[0] + R[1] * rhs.R[1] + R[2] * rhs.R[2] + R[3] * rhs.R[3], s = d < 0.f ? -1.f : 1.f;
		for (int i = 0; i < 4; ++i) k.R[i] = R[i] + s * rhs.R[i];
		for (int i = 0; i < 3; ++i) k.T[i] = T[i] + rhs.T[i];
		k.slice += rhs.slice; k.scale += rhs.scale; k.fov += rhs.fov; k.aperture_size += rhs.aperture_size; k.timestamp = rhs.timestamp;
		return k;
	}
	CameraKeyframe normalized() const { CameraKeyframe k = *this; const float l = std::sqrt(R[0] * R[0] + R[1] * R[1] + R[2] * R[2] + R[3] * R[3]); for (auto& v : k.R) v /= l; return k; }
};

struct CameraPath {
	std::vector<CameraKeyframe> keyframes;
	bool loop = false;
	int spline_order = 3;
	float duration_seconds = 0.f;

	bool empty() const { return keyframes.empty(); }
	bool has_valid_timestamps() const { float prev = 0.f; for (const auto& k : keyframes) { if (!(k.timestamp > prev)) return false; prev = k.timestamp; } return true; }
	void sanitize_keyframes() { // invalid timestamps: spread the keyframes evenly over one second (camera_path.cu:217-231)
		if (has_valid_timestamps()) return;
		for (size_t i = 0; i < keyframes.size(); ++i) keyframes[i].timestamp = (float)(i + 1) / (float)keyframes.size();
		duration_seconds = 1.f;
	}
	const CameraKeyframe& get_keyframe(int i) const {
		const int n = (int)keyframes.size();
		return loop ? keyframes[(size_t)(((i % n) + n) % n)] : keyframes[(size_t)std::min(std::max(i, 0), n - 1)];
	}
	// get_pos, camera_path.cu:233-257: playtime in [0, 1] -> keyframe index and the fraction of the segment that ENDS at that keyframe
	void get_pos(float playtime, int& kfidx, float& t) const {
		if (keyframes.empty()) { kfidx = -1; t = 0.f; return; }
		if (keyframes.size() == 1) { kfidx = 0; t = playtime; return; }
		const float duration = loop ? keyframes.back().timestamp : keyframes[keyframes.size() - 2].timestamp;
		playtime *= duration;
		const auto it = std::upper_bound(keyframes.begin(), keyframes.end(), playtime, [](float v, const CameraKeyframe& k) { return v < k.timestamp; });
		const int i = std::min(std::max((int)(it - keyframes.begin()), 0), (int)keyframes.size() - (loop ? 1 : 2));
		const float prev = i == 0 ? 0.f : keyframes[(size_t)i - 1].timestamp;
		kfidx = i; t = (playtime - prev) / (keyframes[(size_t)i].timestamp - prev);
	}
	CameraKeyframe eval_camera_path(float playtime) const { // camera_path.h:172-190 with the blends of camera_path.cu:66-86
		if (keyframes.empty()) return {};
		int i; float t; get_pos(playtime, i, t);
		switch (spline_order) {
			case 0: return get_keyframe(i + (int)std::round(t));
			case 1: return get_keyframe(i).scaled(1.f - t).plus(get_keyframe(i + 1).scaled(t)).normalized();
			case 2: { const float tt = t * t, a = (1 - t) * (1 - t) * 0.5f, b = (-2.f * tt + 2.f * t + 1.f) * 0.5f, c = tt * 0.5f;
				return get_keyframe(i - 1).scaled(a).plus(get_keyframe(i).scaled(b)).plus(get_keyframe(i + 1).scaled(c)).normalized(); }
			case 3: { const float tt = t * t, ttt = t * t * t, a = (1 - t) * (1 - t) * (1 - t) * (1.f / 6.f), b = (3.f * ttt - 6.f * tt + 4.f) * (1.f / 6.f),
					c = (-3.f * ttt + 3.f * tt + 3.f * t + 1.f) * (1.f / 6.f), d = ttt * (1.f / 6.f);
				return get_keyframe(i - 1).scaled(a).plus(get_keyframe(i).scaled(b)).plus(get_keyframe(i + 1).scaled(c)).plus(get_keyframe(i + 2).scaled(d)).normalized(); }
			default: throw std::runtime_error{"Spline of order " + std::to_string(spline_order) + " is not supported."};
		}
	}
	void load_json_text(const std::string& text, const std::string& name) { // CameraPath::load, camera_path.cu:135-168 (load_relative_to_first is off in the reference)
		mini_json::Value j; std::string err;
		if (!mini_json::parse(text.c_str(), j, err)) throw std::runtime_error{"Camera path " + name + ": " + err};
		keyframes.clear();
		if (j.has("loop")) loop = j.boolean("loop", false);
		if (j.has("path")) for (size_t k = 0; k < j["path"].size(); ++k) {
			const mini_json::Value& el = j["path"].at(k);
			if (el["R"].size() != 4 || el["T"].size() != 3) throw std::runtime_error{"Camera path " + name + ": keyframe without R[4] / T[3]"};
			CameraKeyframe p;
			for (int i = 0; i < 4; ++i) p.R[i] = (float)el["R"].at(i).n;
			for (int i = 0; i < 3; ++i) p.T[i] = (float)el["T"].at(i).n;
			p.slice = (float)el.num("slice", 0.0); p.scale = (float)el.num("scale", 1.0); p.fov = (float)el.num("fov", 50.625);
			p.aperture_size = (float)(el.has("dof") ? el.num("dof", 0.0) : el.num("aperture_size", 0.0));
			p.timestamp = (float)el.num("timestamp", 0.0);
			keyframes.push_back(p);
		}
		duration_seconds = (float)j.num("duration_seconds", 0.0);
		spline_order = (int)j.num("spline_order", 3);
		sanitize_keyframes();
	}
};

} // namespace ngp_host
