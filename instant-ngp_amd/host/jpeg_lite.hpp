// jpeg_lite.hpp -- baseline and progressive JPEG decoder for the NeRF loader's data path (reference: load_stbi -> stbi_load, nerf_loader.cu:570-603; the reference vendors
// stb_image).  Scope: what cameras and colmap pipelines write -- 8-bit baseline sequential DCT (SOF0 / SOF1 Huffman), 1 or 3 components, any sampling
// factors, restart intervals, interleaved and per-component scans -- and progressive DCT (SOF2: spectral selection + successive approximation, round 3); arithmetic /
// lossless / CMYK files return false (the caller's fallback decoder gets them).  The three places where JPEG decoders legitimately differ by a level -- the integer inverse DCT (12-bit constants, rounding in both passes), the
// chroma up-sampling (3:1 tent filter in each direction) and the fixed-point YCbCr -> RGB matrix -- follow stb_image's arithmetic, so that the training
// pixels are the ones the reference trains on (checked bit for bit against stb_image compiled from the reference's tree: oracle/_ref, tests/test_jpeg.py).
#pragma once
#include <cstdint>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

namespace jpeg_lite {

struct Huffman {
	uint8_t fast[512];       // 9-bit prefix -> symbol index (255 = longer code)
	uint16_t code[256];
	uint8_t values[256], size[257];
	uint32_t maxcode[18];
	int delta[17];
	bool defined = false; // a DHT segment has filled this table (a scan that names an undefined table is rejected: read_sos)
	bool build(const uint8_t counts[16], const uint8_t* vals, int n_vals) {
		int k = 0;
		for (int i = 0; i < 16; ++i) for (int j = 0; j < counts[i]; ++j) { if (k >= 256) return false; size[k++] = (uint8_t)(i + 1); }
		if (k != n_vals) return false;
		size[k] = 0;
		std::memcpy(values, vals, (size_t)n_vals);
		uint32_t c = 0; k = 0;
		for (int len = 1; len <= 16; ++len) {
			delta[len] = k - (int)c;
			if (size[k] == len) { while (size[k] == len) code[k++] = (uint16_t)c++; if (c - 1 >= (1u << len)) return false; }
			maxcode[len] = c << (16 - len);
			c <<= 1;
		}
		maxcode[17] = 0xffffffffu;
		std::memset(fast, 255, sizeof(fast));
		for (int i = 0; i < k; ++i) {
			const int s = size[i];
			if (s <= 9) { const int c0 = code[i] << (9 - s), m = 1 << (9 - s); for (int j = 0; j < m; ++j) fast[c0 + j] = (uint8_t)i; }
		}
		defined = true;
		return true;
	}
};

struct Component {
	int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0, dc_pred = 0;
	int x = 0, y = 0, w2 = 0, h2 = 0; // size in samples, padded plane size
	std::vector<uint8_t> plane;
	std::vector<int16_t> coeff; // progressive files: 64 coefficients per block of the padded plane, block row-major
};

class Decoder {
public:
	bool decode(const uint8_t* data, size_t size, int& w, int& h, std::vector<uint8_t>& rgba) {
		buf = data; end = data + size; pos = data;
		if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return false;
		pos += 2;
		bool have_frame = false;
		for (;;) {
			const int m = next_marker();
			if (m < 0) return false;
			if (m == 0xD9) break; // EOI
			if (m == 0xDA) { // SOS
				if (!have_frame || !read_sos() || !(progressive ? decode_scan_progressive() : decode_scan())) return false;
				continue;
			}
			if (m == 0xC0 || m == 0xC1 || m == 0xC2) { if (have_frame) return false; progressive = m == 0xC2; if (!read_sof()) return false; have_frame = true; continue; }
			if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) return false; // lossless / arithmetic / hierarchical
			if (!read_segment(m)) return false;
		}
		if (!have_frame || !scanned) return false;
		if (progressive) finish_progressive();
		return output(w, h, rgba);
	}

private:
	const uint8_t *buf = nullptr, *end = nullptr, *pos = nullptr;
	Huffman hdc[4] = {}, hac[4] = {};
	uint16_t dequant[4][64] = {};
	Component comp[3];
	int n_comp = 0, img_w = 0, img_h = 0, h_max = 1, v_max = 1, mcu_w = 0, mcu_h = 0, mcu_x = 0, mcu_y = 0;
	int restart_interval = 0, todo = 0, scan_n = 0, order[3] = {0, 0, 0};
	bool jfif = false, scanned = false; int adobe_transform = -1;
	// progressive (SOF2): coefficients of every block are kept over the scans (spectral selection ss..se, successive approximation ah / al), the inverse DCT runs at the end
	bool progressive = false; int ss = 0, se = 63, ah = 0, al = 0, eob_run = 0;
	// entropy-coded segment reader
	uint32_t code_buffer = 0; int code_bits = 0; bool nomore = false; int marker = -1;

	static int dezigzag(int i) {
		static const uint8_t z[64 + 15] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
			30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};
		return z[i];
	}
	int get8() { return pos < end ? *pos++ : 0; }
	int get16() { const int a = get8(); return (a << 8) | get8(); }
	int next_marker() {
		if (marker >= 0) { const int m = marker; marker = -1; return m; }
		int x = get8();
		while (x != 0xFF) { if (pos >= end) return -1; x = get8(); } // (garbage between segments is skipped)
		while (x == 0xFF) { if (pos >= end) return -1; x = get8(); }
		return x;
	}
	bool read_segment(int m) {
		if (m == 0xDD) { if (get16() != 4) return false; restart_interval = get16(); return true; } // DRI
		if (m == 0xDB) { // DQT
			int L = get16() - 2;
			while (L > 0) {
				const int q = get8(), p = q >> 4, t = q & 15;
				if ((p != 0 && p != 1) || t > 3) return false;
				for (int i = 0; i < 64; ++i) dequant[t][dezigzag(i)] = (uint16_t)(p ? get16() : get8());
				L -= p ? 129 : 65;
			}
			return L == 0;
		}
		if (m == 0xC4) { // DHT
			int L = get16() - 2;
			while (L > 0) {
				const int q = get8(), tc = q >> 4, th = q & 15;
				if (tc > 1 || th > 3) return false;
				uint8_t counts[16]; int n = 0;
				for (int i = 0; i < 16; ++i) { counts[i] = (uint8_t)get8(); n += counts[i]; }
				if (n > 256 || pos + n > end) return false;
				if (!(tc == 0 ? hdc[th] : hac[th]).build(counts, pos, n)) return false;
				pos += n; L -= 17 + n;
			}
			return L == 0;
		}
		if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) { // APPn / COM
			int L = get16();
			if (L < 2) return false;
			L -= 2;
			if (m == 0xE0 && L >= 5 && pos + 5 <= end && !std::memcmp(pos, "JFIF\0", 5)) jfif = true;
			if (m == 0xEE && L >= 12 && pos + 12 <= end && !std::memcmp(pos, "Adobe\0", 6)) adobe_transform = pos[11];
			if (pos + L > end) return false;
			pos += L;
			return true;
		}
		return false; // unknown marker
	}
	bool read_sof() {
		const int L = get16();
		if (get8() != 8) return false; // 8-bit samples only
		img_h = get16(); img_w = get16(); n_comp = get8();
		if (img_w <= 0 || img_h <= 0 || (uint64_t)img_w * img_h > (1ull << 28) || (n_comp != 1 && n_comp != 3) || L != 8 + 3 * n_comp) return false;
		h_max = v_max = 1;
		for (int i = 0; i < n_comp; ++i) {
			Component& c = comp[i];
			c.id = get8(); const int q = get8(); c.h = q >> 4; c.v = q & 15; c.tq = get8();
			if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return false;
			h_max = std::max(h_max, c.h); v_max = std::max(v_max, c.v);
		}
		for (int i = 0; i < n_comp; ++i) if (h_max % comp[i].h || v_max % comp[i].v) return false;
		mcu_w = h_max * 8; mcu_h = v_max * 8;
		mcu_x = (img_w + mcu_w - 1) / mcu_w; mcu_y = (img_h + mcu_h - 1) / mcu_h;
		for (int i = 0; i < n_comp; ++i) {
			Component& c = comp[i];
			c.x = (img_w * c.h + h_max - 1) / h_max; c.y = (img_h * c.v + v_max - 1) / v_max;
			c.w2 = mcu_x * c.h * 8; c.h2 = mcu_y * c.v * 8;
			c.plane.assign((size_t)c.w2 * c.h2, 0);
			if (progressive) c.coeff.assign((size_t)c.w2 * c.h2, 0);
		}
		return true;
	}
	bool read_sos() {
		const int L = get16();
		scan_n = get8();
		if (scan_n < 1 || scan_n > n_comp || L != 6 + 2 * scan_n) return false;
		for (int i = 0; i < scan_n; ++i) {
			const int id = get8(), q = get8();
			int which = -1;
			for (int k = 0; k < n_comp; ++k) if (comp[k].id == id) which = k;
			if (which < 0) return false;
			comp[which].hd = q >> 4; comp[which].ha = q & 15;
			if (comp[which].hd > 3 || comp[which].ha > 3) return false;
			order[i] = which;
		}
		ss = get8(); se = get8(); const int a = get8(); ah = a >> 4; al = a & 15;
		for (int i = 0; i < scan_n; ++i) { // every table the scan will decode with must have been defined (corrupt files: clean failure instead of garbage tables)
			const Component& c = comp[order[i]];
			const bool needs_dc = !progressive || ss == 0, needs_ac = !progressive || se > 0;
			if ((needs_dc && ah == 0 && !hdc[c.hd].defined) || (needs_ac && !hac[c.ha].defined)) return false;
		}
		if (progressive) { if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13) return false; }
		else if (ss != 0 || ah != 0 || al != 0) return false; // baseline: the whole spectrum, no successive approximation
		return true;
	}
	void reset_entropy() {
		code_bits = 0; code_buffer = 0; nomore = false; marker = -1; eob_run = 0;
		for (int i = 0; i < n_comp; ++i) comp[i].dc_pred = 0;
		todo = restart_interval ? restart_interval : 0x7fffffff;
	}
	void grow() {
		do {
			int b = nomore ? 0 : get8();
			if (b == 0xFF) {
				int c = get8();
				while (c == 0xFF) c = get8();
				if (c != 0) { marker = c; nomore = true; return; }
			}
			code_buffer |= (uint32_t)b << (24 - code_bits);
			code_bits += 8;
		} while (code_bits <= 24);
	}
	int huff_decode(const Huffman& h) {
		if (code_bits < 16) grow();
		const int c = (code_buffer >> 23) & 511, k = h.fast[c];
		if (k < 255) {
			const int s = h.size[k];
			if (s > code_bits) return -1;
			code_buffer <<= s; code_bits -= s;
			return h.values[k];
		}
		const uint32_t temp = code_buffer >> 16;
		int len = 10;
		for (; len < 17; ++len) if (temp < h.maxcode[len]) break;
		if (len == 17) { code_bits -= 16; return -1; }
		if (len > code_bits) return -1;
		const int idx = (int)((code_buffer >> (32 - len)) & ((1u << len) - 1u)) + h.delta[len];
		if (idx < 0 || idx > 255) return -1;
		code_bits -= len; code_buffer <<= len;
		return h.values[idx];
	}
	int extend_receive(int n) { // n bits, sign-extended the JPEG way
		if (code_bits < n) grow();
		if (code_bits < n) return 0;
		const int sgn = (int)(code_buffer >> 31); // 1 = non-negative value class
		const uint32_t k = (code_buffer << n) | (code_buffer >> (32 - n)); // rotate left
		const uint32_t mask = (1u << n) - 1u;
		code_buffer = k & ~mask;
		code_bits -= n;
		const int v = (int)(k & mask);
		return sgn ? v : v - (int)mask;
	}
	int get_bits(int n) { // n raw bits
		if (code_bits < n) grow();
		if (code_bits < n) return 0; // truncated stream (a marker stopped grow())
		const uint32_t k = (code_buffer << n) | (code_buffer >> (32 - n)), mask = (1u << n) - 1u;
		code_buffer = k & ~mask;
		code_bits -= n;
		return (int)(k & mask);
	}
	bool get_bit() {
		if (code_bits < 1) grow();
		if (code_bits < 1) return false; // truncated stream
		const uint32_t k = code_buffer;
		code_buffer <<= 1; --code_bits;
		return (k & 0x80000000u) != 0;
	}
	// progressive DC scan of one block: first pass = the difference-coded DC shifted by al (and the block is cleared), refinement = one more bit
	bool decode_block_prog_dc(int16_t* data, Component& c) {
		if (se != 0) return false; // a scan carries DC or AC coefficients, not both
		if (code_bits < 16) grow();
		if (ah == 0) {
			std::memset(data, 0, 64 * sizeof(int16_t));
			const int t = huff_decode(hdc[c.hd]);
			if (t < 0 || t > 15) return false;
			const int diff = t ? extend_receive(t) : 0;
			const int dc = c.dc_pred + diff;
			c.dc_pred = dc;
			data[0] = (int16_t)(dc * (1 << al));
		} else if (get_bit()) data[0] = (int16_t)(data[0] + (int16_t)(1 << al));
		return true;
	}
	// progressive AC scan of one block (ITU T.81 G.1.2.2 / G.1.2.3): first pass = run / size pairs with end-of-band runs over several blocks, refinement = one correction
	// bit for every coefficient that is already non-zero and +-(1 << al) for the newly non-zero ones
	bool decode_block_prog_ac(int16_t* data, Component& c) {
		if (ss == 0) return false;
		const Huffman& h = hac[c.ha];
		if (ah == 0) {
			if (eob_run) { --eob_run; return true; }
			int k = ss;
			do {
				if (code_bits < 16) grow();
				const int rs = huff_decode(h);
				if (rs < 0) return false;
				const int s = rs & 15, r = rs >> 4;
				if (s == 0) {
					if (r < 15) { eob_run = 1 << r; if (r) eob_run += get_bits(r); --eob_run; break; }
					k += 16;
				} else {
					k += r;
					const int zig = dezigzag(k++);
					data[zig] = (int16_t)(extend_receive(s) * (1 << al));
				}
			} while (k <= se);
			return true;
		}
		const int16_t bit = (int16_t)(1 << al);
		auto refine = [&](int16_t* p) { if (get_bit() && (*p & bit) == 0) *p = (int16_t)(*p > 0 ? *p + bit : *p - bit); };
		if (eob_run) {
			--eob_run;
			for (int k = ss; k <= se; ++k) { int16_t* p = &data[dezigzag(k)]; if (*p != 0) refine(p); }
			return true;
		}
		int k = ss;
		do {
			const int rs = huff_decode(h);
			if (rs < 0) return false;
			int s = rs & 15, r = rs >> 4;
			if (s == 0) {
				if (r < 15) { eob_run = (1 << r) - 1; if (r) eob_run += get_bits(r); r = 64; } // the rest of the band: only refinements
			} else {
				if (s != 1) return false;
				s = get_bit() ? bit : -bit;
			}
			while (k <= se) { // skip r zero-history coefficients, refining the non-zero ones on the way; then place the new coefficient
				int16_t* p = &data[dezigzag(k++)];
				if (*p != 0) refine(p);
				else { if (r == 0) { *p = (int16_t)s; break; } --r; }
			}
		} while (k <= se);
		return true;
	}
	bool decode_scan_progressive() {
		reset_entropy();
		if (scan_n == 1) {
			Component& c = comp[order[0]];
			const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3, cw = c.w2 >> 3;
			for (int j = 0; j < bh; ++j) for (int i = 0; i < bw; ++i) {
				int16_t* data = &c.coeff[64 * ((size_t)i + (size_t)j * cw)];
				if (!(ss == 0 ? decode_block_prog_dc(data, c) : decode_block_prog_ac(data, c))) return false;
				if (!restart_or_end()) { j = bh; break; }
			}
		} else {
			for (int j = 0; j < mcu_y; ++j) for (int i = 0; i < mcu_x; ++i) {
				for (int k = 0; k < scan_n; ++k) {
					Component& c = comp[order[k]];
					const int cw = c.w2 >> 3;
					for (int y = 0; y < c.v; ++y) for (int x = 0; x < c.h; ++x)
						if (!decode_block_prog_dc(&c.coeff[64 * ((size_t)(i * c.h + x) + (size_t)(j * c.v + y) * cw)], c)) return false;
				}
				if (!restart_or_end()) { j = mcu_y; break; }
			}
		}
		scanned = true;
		if (marker < 0 && !nomore) {
			while (pos < end) { if (*pos == 0xFF && pos + 1 < end && pos[1] != 0 && !(pos[1] >= 0xD0 && pos[1] <= 0xD7)) break; ++pos; }
		}
		return true;
	}
	void finish_progressive() { // dequantise with the tables in force at the end of the file, inverse DCT
		for (int n = 0; n < n_comp; ++n) {
			Component& c = comp[n];
			const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3, cw = c.w2 >> 3;
			for (int j = 0; j < bh; ++j) for (int i = 0; i < bw; ++i) {
				int16_t* data = &c.coeff[64 * ((size_t)i + (size_t)j * cw)];
				for (int k = 0; k < 64; ++k) data[k] = (int16_t)(data[k] * dequant[c.tq][k]);
				idct_block(&c.plane[(size_t)c.w2 * j * 8 + (size_t)i * 8], c.w2, data);
			}
		}
	}
	bool decode_block(int16_t data[64], Component& c) {
		if (code_bits < 16) grow();
		const int t = huff_decode(hdc[c.hd]);
		if (t < 0 || t > 15) return false;
		std::memset(data, 0, 64 * sizeof(int16_t));
		const int diff = t ? extend_receive(t) : 0;
		const int dc = c.dc_pred + diff;
		c.dc_pred = dc;
		data[0] = (int16_t)(dc * dequant[c.tq][0]);
		int k = 1;
		do {
			if (code_bits < 16) grow();
			const int rs = huff_decode(hac[c.ha]);
			if (rs < 0) return false;
			const int s = rs & 15, r = rs >> 4;
			if (s == 0) { if (rs != 0xF0) break; k += 16; }
			else {
				k += r;
				const int zig = dezigzag(k++);
				data[zig] = (int16_t)(extend_receive(s) * dequant[c.tq][zig]);
			}
		} while (k < 64);
		return true;
	}
	static uint8_t clamp8(int x) { return (unsigned)x > 255u ? (x < 0 ? 0 : 255) : (uint8_t)x; }
	// integer inverse DCT with 12-bit constants: columns first (kept 2 extra bits), then rows (+128 level shift), the arithmetic of stb_image's stbi__idct_block
	static void idct_block(uint8_t* out, int out_stride, const int16_t data[64]) {
		// (int)(x * 4096 + 0.5) truncates toward zero, for the negative constants as well: -1.847759065 -> -7567, not -7568
		auto f2f = [](float x) { return (int)(x * 4096 + 0.5); };
		static const int c0541 = f2f(0.5411961f), cm1847 = f2f(-1.847759065f), c0765 = f2f(0.765366865f), c1175 = f2f(1.175875602f), c0298 = f2f(0.298631336f), c2053 = f2f(2.053119869f),
			c3072 = f2f(3.072711026f), c1501 = f2f(1.501321110f), cm0899 = f2f(-0.899976223f), cm2562 = f2f(-2.562915447f), cm1961 = f2f(-1.961570560f), cm0390 = f2f(-0.390180644f);
		int val[64];
		auto pass = [&](int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7, int& x0, int& x1, int& x2, int& x3, int& t0, int& t1, int& t2, int& t3) {
			int p2 = s2, p3 = s6;
			int p1 = (p2 + p3) * c0541;
			t2 = p1 + p3 * cm1847; t3 = p1 + p2 * c0765;
			p2 = s0; p3 = s4;
			t0 = (p2 + p3) * 4096; t1 = (p2 - p3) * 4096;
			x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;
			t0 = s7; t1 = s5; t2 = s3; t3 = s1;
			p3 = t0 + t2; int p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;
			const int p5 = (p3 + p4) * c1175;
			t0 = t0 * c0298; t1 = t1 * c2053; t2 = t2 * c3072; t3 = t3 * c1501;
			p1 = p5 + p1 * cm0899; p2 = p5 + p2 * cm2562; p3 = p3 * cm1961; p4 = p4 * cm0390;
			t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;
		};
		for (int i = 0; i < 8; ++i) {
			const int16_t* d = data + i; int* v = val + i;
			if (d[8] == 0 && d[16] == 0 && d[24] == 0 && d[32] == 0 && d[40] == 0 && d[48] == 0 && d[56] == 0) {
				const int dcterm = d[0] * 4;
				v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dcterm;
			} else {
				int x0, x1, x2, x3, t0, t1, t2, t3;
				pass(d[0], d[8], d[16], d[24], d[32], d[40], d[48], d[56], x0, x1, x2, x3, t0, t1, t2, t3);
				x0 += 512; x1 += 512; x2 += 512; x3 += 512;
				v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10; v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
				v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10; v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
			}
		}
		for (int i = 0; i < 8; ++i) {
			const int* v = val + i * 8; uint8_t* o = out + (size_t)i * out_stride;
			int x0, x1, x2, x3, t0, t1, t2, t3;
			pass(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], x0, x1, x2, x3, t0, t1, t2, t3);
			const int bias = 65536 + (128 << 17);
			x0 += bias; x1 += bias; x2 += bias; x3 += bias;
			o[0] = clamp8((x0 + t3) >> 17); o[7] = clamp8((x0 - t3) >> 17); o[1] = clamp8((x1 + t2) >> 17); o[6] = clamp8((x1 - t2) >> 17);
			o[2] = clamp8((x2 + t1) >> 17); o[5] = clamp8((x2 - t1) >> 17); o[3] = clamp8((x3 + t0) >> 17); o[4] = clamp8((x3 - t0) >> 17);
		}
	}
	bool restart_or_end() { // every MCU counts down the restart interval; false = the interval is over and no RSTn marker follows: the scan ends here
		if (--todo > 0) return true;
		if (code_bits < 24) grow();
		if (marker < 0xD0 || marker > 0xD7) return false;
		reset_entropy();
		return true;
	}
	bool decode_scan() {
		reset_entropy();
		int16_t data[64];
		if (scan_n == 1) { // one component: its own block grid, no MCU interleaving
			Component& c = comp[order[0]];
			const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
			for (int j = 0; j < bh; ++j) for (int i = 0; i < bw; ++i) {
				if (!decode_block(data, c)) return false;
				idct_block(&c.plane[(size_t)c.w2 * j * 8 + (size_t)i * 8], c.w2, data);
				if (!restart_or_end()) { j = bh; break; }
			}
		} else {
			for (int j = 0; j < mcu_y; ++j) for (int i = 0; i < mcu_x; ++i) {
				for (int k = 0; k < scan_n; ++k) {
					Component& c = comp[order[k]];
					for (int y = 0; y < c.v; ++y) for (int x = 0; x < c.h; ++x) {
						if (!decode_block(data, c)) return false;
						idct_block(&c.plane[(size_t)c.w2 * ((size_t)(j * c.v + y) * 8) + (size_t)(i * c.h + x) * 8], c.w2, data);
					}
				}
				if (!restart_or_end()) { j = mcu_y; break; }
			}
		}
		scanned = true;
		if (marker < 0 && !nomore) { // find the marker that ends the entropy-coded segment
			while (pos < end) { if (*pos == 0xFF && pos + 1 < end && pos[1] != 0 && !(pos[1] >= 0xD0 && pos[1] <= 0xD7)) break; ++pos; }
		}
		return true;
	}
	// ---- chroma up-sampling: 3:1 tent filters with stb_image's rounding ----
	static const uint8_t* up_v2(uint8_t* out, const uint8_t* nr, const uint8_t* fr, int w, int) { for (int i = 0; i < w; ++i) out[i] = (uint8_t)((3 * nr[i] + fr[i] + 2) >> 2); return out; }
	static const uint8_t* up_h2(uint8_t* out, const uint8_t* in, const uint8_t*, int w, int) {
		if (w == 1) { out[0] = out[1] = in[0]; return out; }
		out[0] = in[0]; out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
		int i;
		for (i = 1; i < w - 1; ++i) { const int n = 3 * in[i] + 2; out[i * 2 + 0] = (uint8_t)((n + in[i - 1]) >> 2); out[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2); }
		out[i * 2 + 0] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2); out[i * 2 + 1] = in[w - 1];
		return out;
	}
	static const uint8_t* up_hv2(uint8_t* out, const uint8_t* nr, const uint8_t* fr, int w, int) {
		if (w == 1) { out[0] = out[1] = (uint8_t)((3 * nr[0] + fr[0] + 2) >> 2); return out; }
		int t1 = 3 * nr[0] + fr[0], t0;
		out[0] = (uint8_t)((t1 + 2) >> 2);
		for (int i = 1; i < w; ++i) { t0 = t1; t1 = 3 * nr[i] + fr[i]; out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4); out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4); }
		out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
		return out;
	}
	static const uint8_t* up_generic(uint8_t* out, const uint8_t* nr, const uint8_t*, int w, int hs) { for (int i = 0; i < w; ++i) for (int j = 0; j < hs; ++j) out[i * hs + j] = nr[i]; return out; }
	bool output(int& w, int& h, std::vector<uint8_t>& rgba) {
		w = img_w; h = img_h;
		rgba.assign((size_t)w * h * 4, 255);
		struct Up { const uint8_t* (*fn)(uint8_t*, const uint8_t*, const uint8_t*, int, int); int hs, vs, ystep, w_lores, ypos; const uint8_t *line0, *line1; std::vector<uint8_t> buf; };
		Up up[3];
		for (int k = 0; k < n_comp; ++k) {
			Up& r = up[k];
			r.hs = h_max / comp[k].h; r.vs = v_max / comp[k].v; r.ystep = r.vs >> 1; r.w_lores = (img_w + r.hs - 1) / r.hs; r.ypos = 0;
			r.line0 = r.line1 = comp[k].plane.data();
			r.buf.resize((size_t)img_w + 8 * r.hs + 8);
			r.fn = (r.hs == 1 && r.vs == 1) ? nullptr : (r.hs == 1 && r.vs == 2) ? up_v2 : (r.hs == 2 && r.vs == 1) ? up_h2 : (r.hs == 2 && r.vs == 2) ? up_hv2 : up_generic;
		}
		// three components are YCbCr unless the file says RGB (component ids 'R','G','B', or an Adobe marker with transform 0 and no JFIF header)
		const bool is_rgb = n_comp == 3 && ((comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') || (adobe_transform == 0 && !jfif));
		auto fix = [](float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; };
		const int k_cr_r = fix(1.40200f), k_cr_g = -fix(0.71414f), k_cb_g = -fix(0.34414f), k_cb_b = fix(1.77200f);
		const uint8_t* row[3] = {nullptr, nullptr, nullptr};
		for (int j = 0; j < img_h; ++j) {
			for (int k = 0; k < n_comp; ++k) {
				Up& r = up[k];
				const bool y_bot = r.ystep >= (r.vs >> 1);
				const uint8_t *nr = y_bot ? r.line1 : r.line0, *fr = y_bot ? r.line0 : r.line1;
				row[k] = r.fn ? r.fn(r.buf.data(), nr, fr, r.w_lores, r.hs) : nr;
				if (++r.ystep >= r.vs) { r.ystep = 0; r.line0 = r.line1; if (++r.ypos < comp[k].y) r.line1 += comp[k].w2; }
			}
			uint8_t* o = &rgba[(size_t)j * w * 4];
			if (n_comp == 1) for (int i = 0; i < w; ++i) { o[4 * i] = o[4 * i + 1] = o[4 * i + 2] = row[0][i]; }
			else if (is_rgb) for (int i = 0; i < w; ++i) { o[4 * i] = row[0][i]; o[4 * i + 1] = row[1][i]; o[4 * i + 2] = row[2][i]; }
			else for (int i = 0; i < w; ++i) {
				const int y_fixed = (row[0][i] << 20) + (1 << 19), cr = row[2][i] - 128, cb = row[1][i] - 128;
				int rr = y_fixed + cr * k_cr_r, g = y_fixed + cr * k_cr_g + (int)(((unsigned)(cb * k_cb_g)) & 0xffff0000u), b = y_fixed + cb * k_cb_b;
				rr >>= 20; g >>= 20; b >>= 20;
				o[4 * i] = clamp8(rr); o[4 * i + 1] = clamp8(g); o[4 * i + 2] = clamp8(b);
			}
		}
		return true;
	}
};

inline bool decode_file(const std::string& path, int& w, int& h, std::vector<uint8_t>& rgba) {
	std::ifstream f{path, std::ios::binary};
	if (!f) return false;
	std::vector<uint8_t> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
	Decoder d;
	return d.decode(b.data(), b.size(), w, h, rgba);
}

} // namespace jpeg_lite
