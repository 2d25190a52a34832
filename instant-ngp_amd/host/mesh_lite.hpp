// mesh_lite.hpp -- minimal Wavefront OBJ and binary STL readers for the SDF primitive's data path (reference: load_obj via tinyobjloader,
// src/tinyobj_loader_wrapper.cpp; dependencies/tinyobjloader is not used).  Output: 3 vertices per triangle, faces with more than three
// corners triangulated like tinyobjloader's `triangulate` default: quads along the shorter diagonal, five and more corners by its ear clipping; texture / normal indices,
// groups and materials are ignored.
// load_stl: testbed_sdf.cu:1328-1361 (binary only: 80-byte header, uint32 face count, 50-byte faces = normal, 3 vertices, attribute word).
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace mesh_lite {

// tinyobjloader's rule for quads (tiny_obj_loader.h, `triangulate`): [0, 1, 2] + [0, 2, 3] iff |v2 - v0|^2 < |v3 - v1|^2 in fp32, else [0, 1, 3] + [1, 2, 3].
// Evaluated without FMA contraction, like the reference's host build (pinned against the library itself: tests/test_ref_loaders.py).
#if defined(__GNUC__) && !defined(__clang__)
__attribute__((optimize("fp-contract=off")))
#endif
inline bool quad_splits_along_02(const float* v0, const float* v1, const float* v2, const float* v3) {
	const float ax = v2[0] - v0[0], ay = v2[1] - v0[1], az = v2[2] - v0[2], bx = v3[0] - v1[0], by = v3[1] - v1[1], bz = v3[2] - v1[2];
	volatile float a0 = ax * ax, a1 = ay * ay, a2 = az * az, b0 = bx * bx, b1 = by * by, b2 = bz * bz; // volatile: every product is rounded to fp32 before the sums
	const float sqr02 = a0 + a1 + a2, sqr13 = b0 + b1 + b2;
	return sqr02 < sqr13;
}

// tinyobjloader's triangulation of faces with five and more corners (its built-in ear clipping), restated from the library's behaviour and pinned against the library
// itself (tests/test_ref_loaders.py): the face is projected onto two coordinate axes chosen from its first non-degenerate corner, then corners are clipped in a fixed
// walk -- a candidate (v0, v1, v2) starting at `guess` is an ear unless cross * "area" < 0 (the library's own sign test) or another remaining corner lies inside it
// (crossing-number test); an ear removes v1; a rejected candidate advances `guess`; the walk gives up when no corner could be removed for a full round.  fp32, no
// FMA contraction (see quad_splits_along_02).  Appends vertex indices (three per triangle) to `tri`.
#if defined(__GNUC__) && !defined(__clang__)
__attribute__((optimize("fp-contract=off")))
#endif
inline void ear_clip_like_tinyobj(const std::vector<float>& v, const std::vector<long>& face, std::vector<long>& tri) {
	auto vol = [](float x) { volatile float y = x; return y; }; // every product / sum rounded to fp32 on its own
	size_t axes[2] = {1, 2};
	const size_t n0 = face.size();
	for (size_t k = 0; k < n0; ++k) {
		const float* a = &v[(size_t)face[k % n0] * 3]; const float* b = &v[(size_t)face[(k + 1) % n0] * 3]; const float* c = &v[(size_t)face[(k + 2) % n0] * 3];
		const float e0x = b[0] - a[0], e0y = b[1] - a[1], e0z = b[2] - a[2], e1x = c[0] - b[0], e1y = c[1] - b[1], e1z = c[2] - b[2];
		const float cx = std::fabs(vol(vol(e0y * e1z) - vol(e0z * e1y))), cy = std::fabs(vol(vol(e0z * e1x) - vol(e0x * e1z))), cz = std::fabs(vol(vol(e0x * e1y) - vol(e0y * e1x)));
		const float eps = 1.1920928955078125e-07f; // std::numeric_limits<float>::epsilon()
		if (cx > eps || cy > eps || cz > eps) {
			if (!(cx > cy && cx > cz)) { axes[0] = 0; if (cz > cx && cz > cy) axes[1] = 1; }
			break;
		}
	}
	std::vector<long> rem = face;
	size_t guess = 0, remaining_iterations = rem.size(), previous = rem.size();
	while (rem.size() > 3 && remaining_iterations > 0) {
		const size_t np = rem.size();
		if (guess >= np) guess -= np;
		if (previous != np) { previous = np; remaining_iterations = np; } else --remaining_iterations;
		long ind[3]; float vx[3], vy[3];
		for (size_t k = 0; k < 3; ++k) { ind[k] = rem[(guess + k) % np]; vx[k] = v[(size_t)ind[k] * 3 + axes[0]]; vy[k] = v[(size_t)ind[k] * 3 + axes[1]]; }
		const float e0x = vx[1] - vx[0], e0y = vy[1] - vy[0], e1x = vx[2] - vx[1], e1y = vy[2] - vy[1];
		const float cross = vol(vol(e0x * e1y) - vol(e0y * e1x));
		const float area = vol(vol(vol(vx[0] * vy[1]) - vol(vy[0] * vx[1])) * 0.5f);
		if (vol(cross * area) < 0.0f) { guess += 1; continue; }
		bool overlap = false;
		for (size_t other = 3; other < np && !overlap; ++other) {
			const size_t idx = (guess + other) % np;
			const float tx = v[(size_t)rem[idx] * 3 + axes[0]], ty = v[(size_t)rem[idx] * 3 + axes[1]];
			int c = 0; // crossing number of (tx, ty) against the candidate triangle
			for (int i = 0, j = 2; i < 3; j = i++)
				if (((vy[i] > ty) != (vy[j] > ty)) && (tx < vol(vol(vol(vol(vx[j] - vx[i]) * vol(ty - vy[i])) / vol(vy[j] - vy[i])) + vx[i]))) c = !c;
			overlap = c != 0;
		}
		if (overlap) { guess += 1; continue; }
		tri.push_back(ind[0]); tri.push_back(ind[1]); tri.push_back(ind[2]);
		rem.erase(rem.begin() + (ptrdiff_t)((guess + 1) % np));
	}
	if (rem.size() == 3) { tri.push_back(rem[0]); tri.push_back(rem[1]); tri.push_back(rem[2]); }
}

inline std::vector<float> load_obj(const std::string& path) {
	std::ifstream f{path};
	if (!f) throw std::runtime_error{"obj: could not open '" + path + "'"};
	std::vector<float> v, out;
	std::string line;
	std::vector<long> idx, tri;
	while (std::getline(f, line)) {
		const char* p = line.c_str();
		while (*p == ' ' || *p == '\t') ++p;
		if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
			char* e = nullptr;
			p += 2;
			for (int k = 0; k < 3; ++k) { v.push_back(std::strtof(p, &e)); p = e; }
		} else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
			p += 2; idx.clear();
			while (*p) {
				while (*p == ' ' || *p == '\t' || *p == '\r') ++p;
				if (!*p) break;
				char* e = nullptr;
				long i = std::strtol(p, &e, 10);
				if (e == p) break;
				p = e;
				while (*p && *p != ' ' && *p != '\t' && *p != '\r') ++p; // skip /vt/vn
				const long nv = (long)(v.size() / 3);
				i = i < 0 ? nv + i : i - 1;
				if (i < 0 || i >= nv) throw std::runtime_error{"obj: face index out of range in '" + path + "'"};
				idx.push_back(i);
			}
			auto emit = [&](long q0, long q1, long q2) { for (long q : {q0, q1, q2}) for (int c = 0; c < 3; ++c) out.push_back(v[(size_t)q * 3 + c]); };
			if (idx.size() == 4 && !quad_splits_along_02(&v[(size_t)idx[0] * 3], &v[(size_t)idx[1] * 3], &v[(size_t)idx[2] * 3], &v[(size_t)idx[3] * 3])) {
				emit(idx[0], idx[1], idx[3]); emit(idx[1], idx[2], idx[3]); // tinyobjloader: a quad is split along its SHORTER diagonal (1-3 here, and on a tie)
			} else if (idx.size() <= 4) {
				for (size_t k = 2; k < idx.size(); ++k) emit(idx[0], idx[k - 1], idx[k]);
			} else {
				tri.clear(); ear_clip_like_tinyobj(v, idx, tri);
				for (size_t k = 0; k + 2 < tri.size(); k += 3) emit(tri[k], tri[k + 1], tri[k + 2]);
			}
		}
	}
	if (out.empty()) throw std::runtime_error{"obj: no faces in '" + path + "'"};
	return out;
}

// binary STL: 3 vertices per triangle in file order (the face normals are not used by the SDF path); a truncated file yields the complete
// faces that were read, an ASCII file ("solid ...") or a zero face count is refused -- as the reference does
inline std::vector<float> load_stl(const std::string& path) {
	std::ifstream f{path, std::ios::in | std::ios::binary};
	if (!f) throw std::runtime_error{"Mesh file '" + path + "' not found"};
	unsigned char head[84] = {};
	f.read((char*)head, 84);
	if (f.gcount() < 84) throw std::runtime_error{"Mesh file '" + path + "' too small for STL header"};
	uint32_t n_faces = 0;
	std::memcpy(&n_faces, head + 80, 4);
	if (std::memcmp(head, "solid", 5) == 0 || n_faces == 0) throw std::runtime_error{"ASCII STL file '" + path + "' not supported"};
	std::vector<float> out;
	out.reserve((size_t)n_faces * 9);
	unsigned char face[50];
	for (uint32_t i = 0; i < n_faces; ++i) {
		f.read((char*)face, 50);
		if (f.gcount() < 50) break;
		float v[9];
		std::memcpy(v, face + 12, 36); // behind the 12-byte normal
		out.insert(out.end(), v, v + 9);
	}
	if (out.empty()) throw std::runtime_error{"stl: no faces in '" + path + "'"};
	return out;
}

} // namespace mesh_lite
