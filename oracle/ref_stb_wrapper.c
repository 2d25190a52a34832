/* oracle/_ref: the reference's own image decoder (stb_image, vendored under /root/reference/dependencies/stb_image and used by nerf_loader.cu:570-603),
 * compiled from the reference's tree where it lies -- TEST INFRASTRUCTURE: validates host/jpeg_lite.hpp and the PNG reader bit for bit; never shipped,
 * never linked by the product.  Built by oracle/Makefile only when /root/reference is present. */
#define STB_IMAGE_IMPLEMENTATION
#define STBI_NO_THREAD_LOCALS
#include "stb_image.h"
#include <string.h>

/* load_stbi(path, &w, &h, &comp, 4): RGBA8; returns 1 on success, `out` must hold w * h * 4 bytes (call with out = NULL to get the size) */
int ref_stbi_load_rgba(const char* path, int* w, int* h, unsigned char* out) {
	int comp = 0;
	unsigned char* p = stbi_load(path, w, h, &comp, 4);
	if (!p) return 0;
	if (out) memcpy(out, p, (size_t)(*w) * (size_t)(*h) * 4);
	stbi_image_free(p);
	return 1;
}
/* load_stbi_16(path, &w, &h, &comp, 1): one 16-bit channel (depth images, nerf_loader.cu:633) */
int ref_stbi_load_gray16(const char* path, int* w, int* h, unsigned short* out) {
	int comp = 0;
	unsigned short* p = stbi_load_16(path, w, h, &comp, 1);
	if (!p) return 0;
	if (out) memcpy(out, p, (size_t)(*w) * (size_t)(*h) * 2);
	stbi_image_free(p);
	return 1;
}
