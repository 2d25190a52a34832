// mesh_lite.hpp -- minimal Wavefront OBJ and binary STL readers for the SDF primitive's data path (reference: load_obj via tinyobjloader,
// src/tinyobj_loader_wrapper.cpp; dependencies/tinyobjloader is not used).  Output: 3 vertices per triangle, faces with more than three
// corners fan-triangulated like tinyobjloader's `triangulate` default; texture / normal indices, groups and materials are ignored.
// load_stl: testbed_sdf.cu:1328-1361 (binary only: 80-byte header, uint32 face count, 50-byte faces = normal, 3 vertices, attribute word).
#pragma once
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace mesh_lite {

inline std::vector<float> load_obj(const std::string& path) {
	std::ifstream f{path};
	if (!f) throw std::runtime_error{"obj: could not open '" + path + "'"};
	std::vector<float> v, out;
	std::string line;
	std::vector<long> idx;
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
			for (size_t k = 2; k < idx.size(); ++k)
				for (long q : {idx[0], idx[k - 1], idx[k]}) for (int c = 0; c < 3; ++c) out.push_back(v[(size_t)q * 3 + c]);
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
