// exr_lite.hpp -- minimal OpenEXR reader for the image primitive's data path (reference: load_exr_to_gpu / tinyexr, testbed_image.cu
// load_image; dependencies/tinyexr is not used).  Scope: single-part scan-line files, NO / ZIPS / ZIP compression, half or float
// channels named R, G, B (A optional; a single channel Y is replicated) -- what data/image/albert.exr (1024x1024 RGBA float32, ZIP)
// and image exports of the usual tools use.  Output: RGBA float32, row-major, top scan line first.
#pragma once
#include <zlib.h>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace exr_lite {

inline float half_to_float(uint16_t h) {
	const uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31u, m = h & 1023u;
	uint32_t bits;
	if (e == 0) {
		if (m == 0) bits = s;
		else { int sh = 0; uint32_t mm = m; while (!(mm & 1024u)) { mm <<= 1; ++sh; } bits = s | ((uint32_t)(113 - sh) << 23) | ((mm & 1023u) << 13); }
	} else if (e == 31) bits = s | 0x7f800000u | (m << 13);
	else bits = s | ((e + 112u) << 23) | (m << 13);
	float f; std::memcpy(&f, &bits, 4); return f;
}

struct Channel { std::string name; int type; }; // 0 uint, 1 half, 2 float

inline void read_rgba(const std::string& path, int& width, int& height, std::vector<float>& rgba) {
	std::ifstream f{path, std::ios::binary};
	if (!f) throw std::runtime_error{"exr: could not open '" + path + "'"};
	std::vector<uint8_t> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
	// every size and offset below comes from the file: checked without wrap-around before anything is read
	auto need = [&](size_t p, size_t n) { if (n > b.size() || p > b.size() - n) throw std::runtime_error{"exr: truncated or malformed file '" + path + "'"}; };
	auto bad = [&](const char* what) { return std::runtime_error{std::string("exr: ") + what + " in '" + path + "'"}; };
	need(0, 8);
	uint32_t magic, version; std::memcpy(&magic, &b[0], 4); std::memcpy(&version, &b[4], 4);
	if (magic != 20000630u) throw std::runtime_error{"exr: bad magic in '" + path + "'"};
	if ((version & 0xffu) != 2 || (version & 0x1a00u)) throw std::runtime_error{"exr: only single-part scan-line files are supported ('" + path + "')"};
	size_t p = 8;
	std::vector<Channel> channels;
	int compression = -1, xmin = 0, ymin = 0, xmax = -1, ymax = -1, line_order = 0;
	auto cstr = [&](size_t& q) { size_t e = q; while (e < b.size() && b[e]) ++e; need(e, 1); std::string s((const char*)&b[q], e - q); q = e + 1; return s; };
	while (true) {
		const std::string name = cstr(p);
		if (name.empty()) break;
		const std::string type = cstr(p);
		need(p, 4); int32_t size; std::memcpy(&size, &b[p], 4); p += 4;
		if (size < 0) throw bad("negative attribute size");
		need(p, (size_t)size);
		const size_t v = p; p += (size_t)size;
		if (name == "channels") {
			size_t q = v;
			while (q < v + (size_t)size && b[q]) {
				Channel c; c.name = cstr(q); need(q, 16); if (q + 16 > v + (size_t)size) throw bad("channel list overruns its attribute");
				int32_t t; std::memcpy(&t, &b[q], 4); c.type = t; q += 16;
				if (t < 0 || t > 2) throw bad("unknown channel type");
				channels.push_back(c);
				if (channels.size() > 64) throw bad("too many channels");
			}
		} else if (name == "compression") { if (size < 1) throw bad("short compression attribute"); compression = b[v]; }
		else if (name == "dataWindow") { if (size < 16) throw bad("short dataWindow attribute"); int32_t w[4]; std::memcpy(w, &b[v], 16); xmin = w[0]; ymin = w[1]; xmax = w[2]; ymax = w[3]; }
		else if (name == "lineOrder") { if (size < 1) throw bad("short lineOrder attribute"); line_order = b[v]; }
	}
	if (channels.empty() || xmax < xmin || ymax < ymin) throw std::runtime_error{"exr: missing channels / dataWindow in '" + path + "'"};
	if (compression != 0 && compression != 2 && compression != 3) throw std::runtime_error{"exr: only NO / ZIPS / ZIP compression is supported ('" + path + "')"};
	const int64_t w64 = (int64_t)xmax - xmin + 1, h64 = (int64_t)ymax - ymin + 1;
	if (w64 > (1 << 16) || h64 > (1 << 16) || w64 * h64 > (int64_t)1 << 28) throw bad("dataWindow too large (limit 2^28 pixels)"); // 4 GiB of RGBA floats
	width = (int)w64; height = (int)h64;
	const int lines_per_block = compression == 3 ? 16 : 1;
	const int n_blocks = (height + lines_per_block - 1) / lines_per_block;
	size_t line_bytes = 0;
	std::vector<size_t> ch_off(channels.size());
	for (size_t c = 0; c < channels.size(); ++c) { ch_off[c] = line_bytes; line_bytes += (size_t)width * (channels[c].type == 1 ? 2 : 4); }
	std::map<std::string, int> idx;
	for (size_t c = 0; c < channels.size(); ++c) idx[channels[c].name] = (int)c;
	const bool grey = !idx.count("R") && idx.count("Y");
	if (!grey && (!idx.count("R") || !idx.count("G") || !idx.count("B"))) throw std::runtime_error{"exr: channels R, G, B (or Y) expected in '" + path + "'"};
	const int src[4] = {grey ? idx["Y"] : idx["R"], grey ? idx["Y"] : idx["G"], grey ? idx["Y"] : idx["B"], idx.count("A") ? idx["A"] : -1};
	rgba.assign((size_t)width * height * 4, 1.0f);
	need(p, (size_t)n_blocks * 8);
	std::vector<uint8_t> raw, tmp;
	for (int blk = 0; blk < n_blocks; ++blk) {
		uint64_t off; std::memcpy(&off, &b[p + (size_t)blk * 8], 8);
		if (off > b.size()) throw bad("block offset outside the file");
		need((size_t)off, 8);
		int32_t y0, csize; std::memcpy(&y0, &b[off], 4); std::memcpy(&csize, &b[off + 4], 4);
		if (csize < 0) throw bad("negative block size");
		need((size_t)off + 8, (size_t)csize);
		if (y0 < ymin || y0 > ymax) throw bad("block outside the dataWindow");
		const int n_lines = std::min(lines_per_block, ymax - y0 + 1);
		const size_t usize = line_bytes * (size_t)n_lines;
		if (compression == 0 && (size_t)csize != usize) throw bad("uncompressed block of the wrong size");
		if ((size_t)csize > usize) throw bad("block larger than its scan lines");
		raw.resize(usize);
		if ((size_t)csize == usize) std::memcpy(raw.data(), &b[off + 8], usize); // stored raw (NO compression, or a block zlib could not shrink)
		else {
			tmp.resize(usize);
			uLongf dl = (uLongf)usize;
			if (uncompress(tmp.data(), &dl, &b[off + 8], (uLong)csize) != Z_OK || dl != usize) throw std::runtime_error{"exr: zlib error in '" + path + "'"};
			for (size_t i = 1; i < usize; ++i) tmp[i] = (uint8_t)(tmp[i - 1] + tmp[i] - 128); // predictor
			const size_t half = (usize + 1) / 2;                                             // de-interleave
			for (size_t i = 0; i < usize; ++i) raw[i] = (i & 1) ? tmp[half + i / 2] : tmp[i / 2];
		}
		for (int l = 0; l < n_lines; ++l) {
			const int y = y0 - ymin + l;
			const uint8_t* line = raw.data() + line_bytes * (size_t)l;
			for (int k = 0; k < 4; ++k) {
				if (src[k] < 0) continue;
				const Channel& c = channels[(size_t)src[k]];
				const uint8_t* cp = line + ch_off[(size_t)src[k]];
				float* dst = &rgba[((size_t)y * width) * 4 + k];
				for (int x = 0; x < width; ++x) {
					float v;
					if (c.type == 1) { uint16_t h; std::memcpy(&h, cp + (size_t)x * 2, 2); v = half_to_float(h); }
					else if (c.type == 2) std::memcpy(&v, cp + (size_t)x * 4, 4);
					else { uint32_t u; std::memcpy(&u, cp + (size_t)x * 4, 4); v = (float)u; }
					dst[(size_t)x * 4] = v;
				}
			}
		}
	}
	(void)line_order; // blocks carry their own y coordinate
}

} // namespace exr_lite
