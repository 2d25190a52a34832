/* oracle/_ref: three more of the reference's own dependencies that compile from its tree without its build system -- TEST INFRASTRUCTURE ONLY, never shipped or linked
 * by the product; built by oracle/Makefile when /root/reference is present, from the sources where they lie (nothing is copied):
 *   tinyexr          (dependencies/tinyexr/tinyexr.h, miniz bundled): LoadEXR, what src/tinyexr_wrapper.cu calls          -> validates host/exr_lite.hpp
 *   NaturalSort      (dependencies/NaturalSort/natural_sort.hpp): SI::natural::compare, nerf_loader.cu:347-349              -> validates the loader's frame order
 *   tinyobjloader    (dependencies/tinyobjloader/tiny_obj_loader.h): LoadObj called as src/tinyobj_loader_wrapper.cu does   -> validates host/mesh_lite.hpp */
#define TINYEXR_IMPLEMENTATION
#include "tinyexr.h"
#include "natural_sort.hpp"
#define TINYOBJLOADER_IMPLEMENTATION
#include "tiny_obj_loader.h"
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

extern "C" {
/* LoadEXR: RGBA float32, row-major; out == NULL -> size only. returns 1 on success */
__attribute__((visibility("default"))) int ref_exr_load_rgba(const char* path, int* w, int* h, float* out) {
	float* rgba = nullptr; const char* err = nullptr;
	if (LoadEXR(&rgba, w, h, path, &err) != TINYEXR_SUCCESS) { if (err) FreeEXRErrorMessage(err); return 0; }
	if (out) std::memcpy(out, rgba, (size_t)(*w) * (size_t)(*h) * 4 * sizeof(float));
	free(rgba);
	return 1;
}
__attribute__((visibility("default"))) int ref_natural_less(const char* a, const char* b) { return SI::natural::compare<std::string>(std::string(a), std::string(b)) ? 1 : 0; }
/* triangles as 9 floats each (the vertex positions of every 3-vertex face of every shape, in file order; other faces skipped): out == NULL -> count only */
__attribute__((visibility("default"))) long long ref_obj_load_triangles(const char* path, float* out, long long cap_floats) {
	tinyobj::attrib_t attrib; std::vector<tinyobj::shape_t> shapes; std::vector<tinyobj::material_t> materials; std::string warn, err;
	std::ifstream f{path, std::ios::in | std::ios::binary};
	if (!f) return -1;
	tinyobj::LoadObj(&attrib, &shapes, &materials, &warn, &err, &f);
	if (!err.empty()) return -2;
	long long n = 0;
	for (const auto& s : shapes) {
		size_t off = 0;
		for (size_t fi = 0; fi < s.mesh.num_face_vertices.size(); ++fi) {
			const size_t fv = s.mesh.num_face_vertices[fi];
			if (fv == 3) for (size_t v = 0; v < 3; ++v) {
				const tinyobj::index_t idx = s.mesh.indices[off + v];
				for (int k = 0; k < 3; ++k) { if (out && n < cap_floats) out[n] = attrib.vertices[3 * idx.vertex_index + k]; ++n; }
			}
			off += fv;
		}
	}
	return n;
}
}
