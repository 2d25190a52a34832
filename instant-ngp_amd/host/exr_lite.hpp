// exr_lite.hpp -- minimal OpenEXR reader for the image primitive's data path (reference: load_exr_to_gpu / tinyexr, testbed_image.cu
// load_image; dependencies/tinyexr is not used).  Scope: single-part scan-line files, NO / RLE / ZIPS / ZIP / PIZ compression (what tinyexr reads, minus tiles), half or float
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

// ---- RLE (OpenEXR compression 1): signed run bytes -- n < 0: -n literal bytes follow, n >= 0: the next byte repeated n + 1 times.  false on malformed input.
inline bool rle_decode(const uint8_t* in, size_t n_in, uint8_t* out, size_t n_out) {
	size_t i = 0, o = 0;
	while (i < n_in) {
		const int c = (int8_t)in[i++];
		if (c < 0) { const size_t k = (size_t)(-c); if (k > n_in - i || k > n_out - o) return false; std::memcpy(out + o, in + i, k); i += k; o += k; }
		else { const size_t k = (size_t)c + 1; if (i >= n_in || k > n_out - o) return false; std::memset(out + o, in[i++], k); o += k; }
	}
	return o == n_out;
}

// ---- PIZ (OpenEXR compression 4), restated from the published format: [u16 min, u16 max][bitmap bytes min..max][i32 n][Huffman stream of n bytes].  The 16-bit words of a
// block (channel after channel, every channel as ny rows of nx * size words, size = 1 for half and 2 for float / uint) were mapped through a lookup table that squeezes out the
// unused values, Haar-wavelet transformed per 16-bit plane and Huffman coded with a canonical code over 65537 symbols (the last one = "repeat the previous word").
namespace piz {
constexpr int HUF_ENCBITS = 16, HUF_ENCSIZE = (1 << HUF_ENCBITS) + 1, SHORT_ZEROCODE_RUN = 59, LONG_ZEROCODE_RUN = 63, SHORTEST_LONG_RUN = 2 + LONG_ZEROCODE_RUN - SHORT_ZEROCODE_RUN;
struct Bits { // MSB-first bit reader over [p, e)
	const uint8_t* p; const uint8_t* e; uint64_t acc = 0; int n = 0; bool overrun = false;
	uint32_t get(int k) { while (n < k) { if (p < e) acc = (acc << 8) | *p++; else { acc <<= 8; overrun = true; } n += 8; } n -= k; return (uint32_t)((acc >> n) & ((1ull << k) - 1ull)); }
};
// Huffman stream -> n_raw 16-bit words; false on malformed input
inline bool huf_uncompress(const uint8_t* in, size_t n_in, uint16_t* raw, size_t n_raw) {
	if (n_in < 20) return n_raw == 0 && n_in == 0;
	uint32_t hdr[5]; std::memcpy(hdr, in, 20);
	const uint32_t im = hdr[0], iM = hdr[1], n_bits = hdr[3];
	if (im >= (uint32_t)HUF_ENCSIZE || iM >= (uint32_t)HUF_ENCSIZE || im > iM) return false;
	// code lengths of the symbols im .. iM, 6 bits each, runs of zeros packed (59..62: 2..5 zeros, 63 + 8 bits: 6..261 zeros)
	std::vector<uint8_t> len((size_t)HUF_ENCSIZE, 0);
	Bits tb{in + 20, in + n_in};
	for (uint32_t s = im; s <= iM; ++s) {
		const uint32_t l = tb.get(6);
		if (tb.overrun) return false;
		if (l == (uint32_t)LONG_ZEROCODE_RUN) { uint32_t z = tb.get(8) + (uint32_t)SHORTEST_LONG_RUN; if (s + z > iM + 1) return false; s += z - 1; }
		else if (l >= (uint32_t)SHORT_ZEROCODE_RUN) { uint32_t z = l - (uint32_t)SHORT_ZEROCODE_RUN + 2; if (s + z > iM + 1) return false; s += z - 1; }
		else len[s] = (uint8_t)l;
	}
	if (tb.overrun) return false;
	const uint8_t* data = tb.p; // the packed table ends on a byte boundary of what was consumed
	if ((size_t)(in + n_in - data) * 8 < n_bits) return false;
	// canonical code: the longest codes start at 0, every shorter length continues at (first code of the longer length + its count) >> 1; symbols of one length in ascending order
	uint64_t count[59] = {0}, base[59] = {0};
	for (int s = 0; s < HUF_ENCSIZE; ++s) { if (len[(size_t)s] > 58) return false; ++count[len[(size_t)s]]; }
	{ uint64_t c = 0; for (int l = 58; l > 0; --l) { const uint64_t nc = (c + count[l]) >> 1; base[l] = c; c = nc; } }
	std::vector<uint32_t> first(60, 0), syms; // symbols sorted by (length, symbol); first[l] = index of the first symbol of length l
	{ uint32_t k = 0; for (int l = 1; l <= 58; ++l) { first[(size_t)l] = k; k += (uint32_t)count[l]; } first[59] = k; syms.resize(k);
	  std::vector<uint32_t> fill(first.begin(), first.end());
	  for (int s = 0; s < HUF_ENCSIZE; ++s) if (len[(size_t)s]) syms[fill[len[(size_t)s]]++] = (uint32_t)s; }
	Bits db{data, in + n_in};
	uint64_t left = n_bits; size_t o = 0;
	const uint32_t rlc = iM; // run-length symbol
	while (left > 0) {
		uint64_t v = 0; int l = 0; bool found = false;
		while (l < 58 && left > 0) {
			v = (v << 1) | db.get(1); ++l; --left;
			if (count[l] && v >= base[l] && v - base[l] < count[l]) { found = true; break; }
		}
		if (!found) { // trailing padding bits of the last byte are not a code
			return o == n_raw;
		}
		const uint32_t sym = syms[first[(size_t)l] + (uint32_t)(v - base[l])];
		if (sym == rlc) {
			if (left < 8 || o == 0) return false;
			const uint32_t rep = db.get(8); left -= 8;
			if (rep > n_raw - o) return false;
			for (uint32_t k = 0; k < rep; ++k, ++o) raw[o] = raw[o - 1];
		} else { if (o >= n_raw) return false; raw[o++] = (uint16_t)sym; }
	}
	return o == n_raw && !db.overrun;
}
inline void wdec14(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
	const int ls = (int16_t)l, hs = (int16_t)h;
	const int ai = ls + (hs & 1) + (hs >> 1);
	a = (uint16_t)(int16_t)ai; b = (uint16_t)(int16_t)(ai - hs);
}
inline void wdec16(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
	const int m = l, d = h;
	const int bb = (m - (d >> 1)) & 0xffff;
	const int aa = (d + bb - 0x8000) & 0xffff;
	b = (uint16_t)bb; a = (uint16_t)aa;
}
// inverse 2-D Haar-style wavelet over nx x ny words with strides ox / oy (in words); mx = largest value in the data before the transform
inline void wav2_decode(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t mx) {
	const bool w14 = mx < (1 << 14);
	auto dec = [w14](uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) { if (w14) wdec14(l, h, a, b); else wdec16(l, h, a, b); };
	const int n = nx > ny ? ny : nx;
	int p = 1;
	while (p <= n) p <<= 1;
	p >>= 1;
	int p2 = p;
	p >>= 1;
	while (p >= 1) {
		const ptrdiff_t ey = (ptrdiff_t)oy * (ny - p2), oy1 = (ptrdiff_t)oy * p, oy2 = (ptrdiff_t)oy * p2, ox1 = (ptrdiff_t)ox * p, ox2 = (ptrdiff_t)ox * p2, ex_rel = (ptrdiff_t)ox * (nx - p2);
		ptrdiff_t py = 0;
		uint16_t i00, i01, i10, i11;
		for (; py <= ey; py += oy2) {
			ptrdiff_t px = py;
			for (; px <= py + ex_rel; px += ox2) {
				uint16_t &q00 = in[px], &q01 = in[px + ox1], &q10 = in[px + oy1], &q11 = in[px + oy1 + ox1];
				dec(q00, q10, i00, i10); dec(q01, q11, i01, i11); dec(i00, i01, q00, q01); dec(i10, i11, q10, q11);
			}
			if (nx & p) { uint16_t &q00 = in[px], &q10 = in[px + oy1]; dec(q00, q10, i00, q10); q00 = i00; }
		}
		if (ny & p) {
			ptrdiff_t px = py;
			for (; px <= py + ex_rel; px += ox2) { uint16_t &q00 = in[px], &q01 = in[px + ox1]; dec(q00, q01, i00, q01); q00 = i00; }
		}
		p2 = p;
		p >>= 1;
	}
}
// one PIZ block -> the block's scan lines in file layout (line after line, channel after channel); false on malformed input
inline bool decode_block(const uint8_t* in, size_t n_in, const std::vector<Channel>& channels, int nx, int ny, uint8_t* out, size_t n_out) {
	size_t total = 0;
	for (const Channel& c : channels) total += (size_t)nx * (size_t)ny * (c.type == 1 ? 1u : 2u);
	if (total * 2 != n_out || n_in < 4) return false;
	uint16_t mn, mx; std::memcpy(&mn, in, 2); std::memcpy(&mx, in + 2, 2);
	size_t p = 4;
	std::vector<uint8_t> bitmap(8192, 0);
	if (mn <= mx) { if (mx >= 8192 || (size_t)(mx - mn + 1) > n_in - p) return false; std::memcpy(&bitmap[mn], in + p, (size_t)(mx - mn + 1)); p += (size_t)(mx - mn + 1); }
	std::vector<uint16_t> lut(65536, 0);
	uint32_t k = 0;
	for (uint32_t i = 0; i < 65536; ++i) if (i == 0 || (bitmap[i >> 3] & (1u << (i & 7u)))) lut[k++] = (uint16_t)i;
	const uint16_t max_value = (uint16_t)(k - 1);
	if (n_in - p < 4) return false;
	int32_t length; std::memcpy(&length, in + p, 4); p += 4;
	if (length < 0 || (size_t)length > n_in - p) return false;
	std::vector<uint16_t> tmp(total);
	if (!huf_uncompress(in + p, (size_t)length, tmp.data(), total)) return false;
	size_t start = 0;
	std::vector<size_t> ch_start(channels.size());
	for (size_t c = 0; c < channels.size(); ++c) {
		const int size = channels[c].type == 1 ? 1 : 2;
		ch_start[c] = start;
		for (int j = 0; j < size; ++j) wav2_decode(tmp.data() + start + j, nx, size, ny, nx * size, max_value);
		start += (size_t)nx * (size_t)ny * (size_t)size;
	}
	for (uint16_t& v : tmp) v = lut[v];
	uint8_t* o = out;
	for (int y = 0; y < ny; ++y)
		for (size_t c = 0; c < channels.size(); ++c) {
			const size_t row = (size_t)nx * (channels[c].type == 1 ? 1u : 2u);
			std::memcpy(o, tmp.data() + ch_start[c] + (size_t)y * row, row * 2);
			o += row * 2;
		}
	return true;
}
} // namespace piz

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
	if (compression < 0 || compression > 4) throw std::runtime_error{"exr: only NO / RLE / ZIPS / ZIP / PIZ compression is supported ('" + path + "')"};
	const int64_t w64 = (int64_t)xmax - xmin + 1, h64 = (int64_t)ymax - ymin + 1;
	if (w64 > (1 << 16) || h64 > (1 << 16) || w64 * h64 > (int64_t)1 << 28) throw bad("dataWindow too large (limit 2^28 pixels)"); // 4 GiB of RGBA floats
	width = (int)w64; height = (int)h64;
	const int lines_per_block = compression == 3 ? 16 : compression == 4 ? 32 : 1;
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
		else if (compression == 4) {
			if (!piz::decode_block(&b[off + 8], (size_t)csize, channels, width, n_lines, raw.data(), usize)) throw std::runtime_error{"exr: malformed PIZ block in '" + path + "'"};
		} else {
			tmp.resize(usize);
			uLongf dl = (uLongf)usize;
			if (compression == 1) { if (!rle_decode(&b[off + 8], (size_t)csize, tmp.data(), usize)) throw std::runtime_error{"exr: malformed RLE block in '" + path + "'"}; }
			else if (uncompress(tmp.data(), &dl, &b[off + 8], (uLong)csize) != Z_OK || dl != usize) throw std::runtime_error{"exr: zlib error in '" + path + "'"};
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
