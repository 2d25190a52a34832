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
/* SaveEXRImageToFile: RGBA float32 in, channels A B G R stored as half (as_half != 0) or float, compression = TINYEXR_COMPRESSIONTYPE_* (0 NONE, 1 RLE, 2 ZIPS, 3 ZIP, 4 PIZ).
 * Test files for the host reader's RLE / PIZ paths are written by the reference's own encoder. returns 1 on success */
__attribute__((visibility("default"))) int ref_exr_save_rgba(const char* path, int w, int h, const float* rgba, int compression, int as_half) {
	EXRHeader header; InitEXRHeader(&header);
	EXRImage image; InitEXRImage(&image);
	std::vector<float> planes[4];
	for (int c = 0; c < 4; ++c) { planes[c].resize((size_t)w * h); for (size_t i = 0; i < (size_t)w * h; ++i) planes[c][i] = rgba[i * 4 + (3 - c)]; } /* A B G R */
	float* ptrs[4] = {planes[0].data(), planes[1].data(), planes[2].data(), planes[3].data()};
	image.images = (unsigned char**)ptrs; image.width = w; image.height = h; image.num_channels = 4;
	header.num_channels = 4;
	EXRChannelInfo ch[4]; std::memset(ch, 0, sizeof(ch));
	const char* names[4] = {"A", "B", "G", "R"};
	int in_types[4], out_types[4];
	for (int c = 0; c < 4; ++c) { std::strncpy(ch[c].name, names[c], 255); in_types[c] = TINYEXR_PIXELTYPE_FLOAT; out_types[c] = as_half ? TINYEXR_PIXELTYPE_HALF : TINYEXR_PIXELTYPE_FLOAT; }
	header.channels = ch; header.pixel_types = in_types; header.requested_pixel_types = out_types; header.compression_type = compression;
	const char* err = nullptr;
	const int rc = SaveEXRImageToFile(&image, &header, path, &err);
	if (err) FreeEXRErrorMessage(err);
	return rc == TINYEXR_SUCCESS ? 1 : 0;
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
