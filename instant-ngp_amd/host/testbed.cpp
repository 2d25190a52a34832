// testbed.cpp -- see testbed.hpp.  Host logic only; every device operation is a C-ABI call.
#include "testbed.hpp"
#include "exr_lite.hpp"
#include "jpeg_lite.hpp"
#include "mesh_lite.hpp"
#include "msgpack_lite.hpp"

#include <hip/hip_runtime_api.h>
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace fs = std::filesystem;

namespace ngp_host {

ImageDecoder Testbed::s_fallback_decoder;
std::string Testbed::s_default_root_dir;

#define NGP_CHECK(x) do { if ((x) != 0) throw std::runtime_error{std::string{ngp_last_error()}}; } while (0)
#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error{std::string{#x ": "} + hipGetErrorString(e_)}; } while (0)

static std::string read_text(const fs::path& p) {
	std::ifstream f{p, std::ios::binary};
	if (!f) throw std::runtime_error{"File '" + p.string() + "' does not exist."};
	std::stringstream ss; ss << f.rdbuf();
	return ss.str();
}
static std::string lower(std::string s) { for (auto& c : s) c = (char)tolower(c); return s; }

// ---- PNG reader on top of zlib: every colour type (gray, RGB, palette, gray + alpha, RGBA), bit depths 1 / 2 / 4 / 8 / 16, Adam7 interlacing, tRNS transparency -- the
// files stb_image reads (nerf_loader.cu:570-603), with its conventions for the 8-bit interface (pinned bit for bit against stb_image itself, tests/test_jpeg.py) ----
static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
// un-filtered, expanded image: `ch` channels of `depth` (8 or 16; big-endian samples for 16) bits, row-major.  Low bit depths are widened to 8 bits (gray: v * 255 / (2^d - 1),
// the way stb_image scales them), palette indices are replaced by their colours (RGB, or RGBA when the file has a tRNS chunk), a tRNS colour key adds an alpha channel
// (0 where the pixel equals the key, else opaque).
static bool decode_png_raw(const std::string& path, int& w, int& h, int& ch, int& depth, std::vector<uint8_t>& img) {
	std::ifstream f{path, std::ios::binary};
	if (!f) return false;
	std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
	static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
	if (buf.size() < 33 || memcmp(buf.data(), sig, 8) != 0) return false;
	size_t pos = 8;
	int ctype = -1, interlace = 0, bits = 0;
	std::vector<uint8_t> idat, plte, trns;
	while (pos + 12 <= buf.size()) {
		const uint32_t len = be32(&buf[pos]);
		const std::string type((const char*)&buf[pos + 4], 4);
		const uint8_t* data = &buf[pos + 8];
		if (len > buf.size() || pos + 12 + len > buf.size()) return false;
		if (type == "IHDR") { if (len < 13) return false; w = (int)be32(data); h = (int)be32(data + 4); bits = data[8]; ctype = data[9]; interlace = data[12]; }
		else if (type == "PLTE") plte.assign(data, data + len);
		else if (type == "tRNS") trns.assign(data, data + len);
		else if (type == "IDAT") idat.insert(idat.end(), data, data + len);
		else if (type == "IEND") break;
		pos += 12 + len;
	}
	if (w <= 0 || h <= 0 || (uint64_t)w * h > (1ull << 28) || interlace > 1) return false;
	const int file_ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
	const bool depth_ok = ctype == 0 ? (bits == 1 || bits == 2 || bits == 4 || bits == 8 || bits == 16) : ctype == 3 ? (bits == 1 || bits == 2 || bits == 4 || bits == 8) : (bits == 8 || bits == 16);
	if (!file_ch || !depth_ok) return false;
	if (ctype == 3 && (plte.empty() || plte.size() % 3 || plte.size() > 768)) return false;
	const int px_bits = file_ch * bits;                        // bits per pixel in the file
	const int fbpp = std::max(1, px_bits / 8);                  // the filters' "previous pixel" distance in bytes
	auto row_bytes = [&](int n) { return ((size_t)n * px_bits + 7) / 8; };
	// the inflated stream: one image (or seven Adam7 passes), every row = filter byte + packed samples
	static const int xo[7] = {0, 4, 0, 2, 0, 1, 0}, yo[7] = {0, 0, 4, 0, 2, 0, 1}, xs[7] = {8, 8, 4, 4, 2, 2, 1}, ys[7] = {8, 8, 8, 4, 4, 2, 2};
	const int n_pass = interlace ? 7 : 1;
	size_t total = 0;
	for (int p = 0; p < n_pass; ++p) {
		const int pw = interlace ? (w - xo[p] + xs[p] - 1) / xs[p] : w, ph = interlace ? (h - yo[p] + ys[p] - 1) / ys[p] : h;
		if (pw > 0 && ph > 0) total += (row_bytes(pw) + 1) * (size_t)ph;
	}
	std::vector<uint8_t> raw(total);
	uLongf out_len = (uLongf)raw.size();
	if (idat.empty() || uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) return false;
	// samples of the file, one per byte (8-bit and lower depths) or two (16 bits), in image order
	const int sbytes = bits == 16 ? 2 : 1;
	std::vector<uint8_t> smp((size_t)w * h * file_ch * sbytes);
	size_t off = 0;
	std::vector<uint8_t> prev, cur;
	for (int p = 0; p < n_pass; ++p) {
		const int pw = interlace ? (w - xo[p] + xs[p] - 1) / xs[p] : w, ph = interlace ? (h - yo[p] + ys[p] - 1) / ys[p] : h;
		if (pw <= 0 || ph <= 0) continue;
		const size_t rb = row_bytes(pw);
		prev.assign(rb, 0); cur.assign(rb, 0);
		for (int y = 0; y < ph; ++y) {
			const uint8_t ft = raw[off]; const uint8_t* src = &raw[off + 1]; off += rb + 1;
			if (ft > 4) return false;
			for (size_t x = 0; x < rb; ++x) {
				const int a = x >= (size_t)fbpp ? cur[x - fbpp] : 0, b = y ? prev[x] : 0, c = (y && x >= (size_t)fbpp) ? prev[x - fbpp] : 0;
				int v = src[x];
				switch (ft) {
					case 1: v += a; break;
					case 2: v += b; break;
					case 3: v += (a + b) >> 1; break;
					case 4: { const int q = a + b - c, pa = abs(q - a), pb = abs(q - b), pc = abs(q - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
					default: break;
				}
				cur[x] = (uint8_t)v;
			}
			const int oy = interlace ? yo[p] + y * ys[p] : y;
			for (int x = 0; x < pw; ++x) {
				const int ox = interlace ? xo[p] + x * xs[p] : x;
				uint8_t* d = &smp[((size_t)oy * w + ox) * file_ch * sbytes];
				if (bits >= 8) memcpy(d, &cur[(size_t)x * file_ch * sbytes], (size_t)file_ch * sbytes);
				else { const size_t bit = (size_t)x * bits; d[0] = (uint8_t)((cur[bit >> 3] >> (8 - bits - (bit & 7))) & ((1 << bits) - 1)); } // one channel (gray / palette), MSB first
			}
			prev.swap(cur);
		}
	}
	depth = bits == 16 ? 16 : 8;
	const size_t n_px = (size_t)w * h;
	if (ctype == 3) { // palette -> RGB(A)
		const bool has_a = !trns.empty();
		ch = has_a ? 4 : 3;
		const size_t n_pal = plte.size() / 3;
		img.assign(n_px * ch, 0);
		for (size_t i = 0; i < n_px; ++i) {
			const size_t k = smp[i];
			if (k >= n_pal) return false; // (stb_image reads an uninitialised palette entry here: nothing to match)
			img[i * ch + 0] = plte[k * 3]; img[i * ch + 1] = plte[k * 3 + 1]; img[i * ch + 2] = plte[k * 3 + 2];
			if (has_a) img[i * ch + 3] = k < trns.size() ? trns[k] : 255;
		}
		return true;
	}
	if (bits < 8) { const int scale = bits == 1 ? 255 : bits == 2 ? 85 : 17; for (uint8_t& v : smp) v = (uint8_t)(v * scale); }
	const bool key = !trns.empty() && (ctype == 0 || ctype == 2);
	if (!key) { ch = file_ch; img.swap(smp); return true; }
	if (trns.size() < (size_t)file_ch * 2) return false;
	ch = file_ch + 1;
	img.assign(n_px * ch * sbytes, 0);
	uint8_t kb[6]; // the key in the layout of the samples
	for (int c = 0; c < file_ch; ++c) {
		if (bits == 16) { kb[c * 2] = trns[c * 2]; kb[c * 2 + 1] = trns[c * 2 + 1]; }
		else kb[c] = (uint8_t)(trns[c * 2 + 1] * (bits == 1 ? 255 : bits == 2 ? 85 : bits == 4 ? 17 : 1)); // low byte of the 16-bit key, scaled like the samples
	}
	for (size_t i = 0; i < n_px; ++i) {
		const uint8_t* sp = &smp[i * file_ch * sbytes];
		uint8_t* d = &img[i * ch * sbytes];
		memcpy(d, sp, (size_t)file_ch * sbytes);
		const bool transparent = memcmp(sp, kb, (size_t)file_ch * sbytes) == 0;
		d[file_ch * sbytes] = transparent ? 0 : 255;
		if (sbytes == 2) d[file_ch * sbytes + 1] = transparent ? 0 : 255;
	}
	return true;
}
static bool decode_png(const std::string& path, int& w, int& h, std::vector<uint8_t>& rgba) {
	int ch = 0, depth = 0; std::vector<uint8_t> img;
	if (!decode_png_raw(path, w, h, ch, depth, img)) return false;
	if (depth == 16) { // 16 -> 8 bits like stb_image's 8-bit interface: the high byte
		std::vector<uint8_t> hi((size_t)w * h * ch);
		for (size_t i = 0; i < hi.size(); ++i) hi[i] = img[2 * i];
		img.swap(hi);
	}
	rgba.resize((size_t)w * h * 4);
	for (size_t i = 0; i < (size_t)w * h; ++i) {
		const uint8_t* p = &img[i * ch];
		uint8_t* o = &rgba[i * 4];
		if (ch == 1) { o[0] = o[1] = o[2] = p[0]; o[3] = 255; }
		else if (ch == 2) { o[0] = o[1] = o[2] = p[0]; o[3] = p[1]; }
		else if (ch == 3) { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = 255; }
		else { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = p[3]; }
	}
	return true;
}

// load_stbi_16(path, &w, &h, &comp, 1) (nerf_loader.cu:633): one 16-bit channel.  Gray (+ alpha) files give their gray channel, colour files stb_image's
// integer luma (77 r + 150 g + 29 b) >> 8, 8-bit files are widened by v * 257 (tests/test_jpeg.py: bit-exact against stb_image itself).
static bool decode_png_gray16(const std::string& path, int& w, int& h, std::vector<uint16_t>& out) {
	int ch = 0, depth = 0; std::vector<uint8_t> img;
	if (!decode_png_raw(path, w, h, ch, depth, img)) return false;
	out.resize((size_t)w * h);
	// stb_image converts to the requested channel count in the file's own bit depth and widens afterwards: the luma of an 8-bit colour file is taken in 8 bits
	auto sample = [&](size_t i, int c) -> uint32_t { return depth == 16 ? ((uint32_t)img[(i * ch + c) * 2] << 8) | img[(i * ch + c) * 2 + 1] : (uint32_t)img[i * ch + c]; };
	for (size_t i = 0; i < out.size(); ++i) {
		const uint32_t y = ch >= 3 ? (sample(i, 0) * 77u + sample(i, 1) * 150u + sample(i, 2) * 29u) >> 8 : sample(i, 0);
		out[i] = (uint16_t)(depth == 16 ? y : y * 257u);
	}
	return true;
}

// load_stbi (nerf_loader.cu:570-603) with the built-in readers: PNG and JPEG (baseline + progressive) decode natively in C++ (host/jpeg_lite.hpp); anything else
// (arithmetic-coded or CMYK JPEG, BMP, TGA, ...) is left to the decoder hook a Python host may register
static bool decode_builtin(const std::string& path, int& w, int& h, std::vector<uint8_t>& rgba) {
	const std::string ext = lower(fs::path(path).extension().string());
	if (ext == ".png") return decode_png(path, w, h, rgba);
	if (ext == ".jpg" || ext == ".jpeg") return jpeg_lite::decode_file(path, w, h, rgba);
	return false;
}
bool Testbed::read_image_builtin(const std::string& path, int& w, int& h, std::vector<uint8_t>& rgba) { return decode_builtin(path, w, h, rgba); }
bool Testbed::read_depth_png16(const std::string& path, int& w, int& h, std::vector<uint16_t>& gray) { return decode_png_gray16(path, w, h, gray); }

// Natural ordering of frame paths, nerf_loader.cu:347-349: `SI::natural::compare<std::string>` of the NaturalSort library the reference vendors, restated from its
// behaviour and pinned against the library itself (tests/test_ref_loaders.py, oracle/_ref): characters compare case-INSENSITIVELY; runs of digits compare as numbers
// (leading zeros ignored, then length, then digits), except that a run directly behind a '.' is a fractional part (digit by digit, trailing zeros ignored; against a
// non-fractional run it is "greater" on the left and a tie on the right -- the library's `return true / false` from an int function); a run of blanks behind a blank is
// skipped; when one string ends first it is the smaller one.  Equivalent names ("r_01" / "r_1") keep their file order here (stable sort; the reference's std::sort leaves
// their order unspecified).
static bool natural_less(const std::string& a, const std::string& b) {
	auto at = [](const std::string& s, size_t i) -> unsigned char { return i < s.size() ? (unsigned char)s[i] : 0; };
	auto lower_less = [](unsigned char x, unsigned char y) { return tolower(x) < tolower(y); };
	auto digits_end = [](const std::string& s, size_t i) { while (i < s.size() && isdigit((unsigned char)s[i])) ++i; return i; };
	size_t i = 0, j = 0;
	bool blank1 = false, blank2 = false;
	while (i < a.size() && j < b.size()) {
		while (blank1 && i < a.size() && a[i] == ' ') ++i;
		blank1 = at(a, i) == ' ';
		while (blank2 && j < b.size() && b[j] == ' ') ++j;
		blank2 = at(b, j) == ' ';
		const unsigned char ca = at(a, i), cb = at(b, j);
		if (!isdigit(ca) || !isdigit(cb)) {
			if (lower_less(ca, cb)) return true;
			if (lower_less(cb, ca)) return false;
			if (i >= a.size() || j >= b.size()) break; // (both ran out behind trailing blanks: equivalent)
			++i; ++j;
			continue;
		}
		const size_t ie = digits_end(a, i), je = digits_end(b, j);
		const bool frac1 = i > 0 && a[i - 1] == '.', frac2 = j > 0 && b[j - 1] == '.';
		int r = 0;
		if (frac1 && !frac2) r = 1;
		else if (!frac1 && frac2) r = 0;
		else if (frac1) {
			size_t p = i, q = j;
			while (p < ie && q < je && r == 0) { r = a[p] < b[q] ? -1 : a[p] > b[q] ? 1 : 0; if (r == 0) { ++p; ++q; } }
			if (r == 0) {
				while (p < ie && a[p] == '0') ++p;
				while (q < je && b[q] == '0') ++q;
				r = (p == ie && q != je) ? -1 : (p != ie && q == je) ? 1 : 0;
			}
		} else {
			size_t p = i, q = j;
			while (p < ie && a[p] == '0') ++p;
			while (q < je && b[q] == '0') ++q;
			if (ie - p != je - q) r = ie - p < je - q ? -1 : 1;
			else for (; p < ie && r == 0; ++p, ++q) r = a[p] < b[q] ? -1 : a[p] > b[q] ? 1 : 0;
		}
		if (r < 0) return true;
		if (r > 0) return false;
		i = ie; j = je;
	}
	if (i >= a.size() && j >= b.size()) return false;
	return i >= a.size();
}
bool Testbed::natural_path_less(const std::string& a, const std::string& b) { return natural_less(a, b); }

// ------------------------------------------------------------------------------------------------
Testbed::Testbed() {
	root_dir = s_default_root_dir.empty() ? fs::current_path().string() : s_default_root_dir;
	nerf.training.owner = this;
	if (ngp_device_available()) (void)ngp_init(); // helper streams before any communicator the host may create later (include/ngp_hip.h)
}
Testbed::~Testbed() {
	destroy_trainer();
	if (m_frame_dev) (void)hipFree(m_frame_dev);
}
void Testbed::destroy_trainer() {
	// ngp_nerf_destroy tears the RCCL communicator down with the trainer, and a unique id creates ONE communicator: the next trainer needs a fresh
	// comm_unique_id() / comm_init() on every rank (ensure_trainer says so instead of handing RCCL a consumed id, which hangs or fails on all ranks)
	if (m_comm_up) { m_comm_up = false; m_comm_id.clear(); }
	if (m_image) { ngp_image_destroy(m_image); m_image = nullptr; }
	if (m_sdf) { ngp_sdf_destroy(m_sdf); m_sdf = nullptr; }
	if (m_encmlp) { ngp_encmlp_destroy(m_encmlp); m_encmlp = nullptr; }
	if (m_nerf) { ngp_nerf_destroy(m_nerf); m_nerf = nullptr; }
	if (m_model) { ngp_model_destroy(m_model); m_model = nullptr; }
	m_extra_dims_installed = false; // the next trainer starts from reset_extra_dims' values again (reset_network, testbed.cu:4272)
}

// load_network_config with recursive "parent" merge-patch, testbed.cu:86-97, 254-310
static mini_json::Value load_config_recursive(const fs::path& path, int depth = 0) {
	if (depth > 8) throw std::runtime_error{"network config: parent chain too deep"};
	mini_json::Value v; std::string err;
	if (!mini_json::parse(read_text(path).c_str(), v, err)) throw std::runtime_error{"network config '" + path.string() + "': " + err};
	if (v.has("parent")) {
		mini_json::Value parent = load_config_recursive(path.parent_path() / v.str("parent", ""), depth + 1);
		mini_json::merge_patch(parent, v);
		return parent;
	}
	return v;
}

void Testbed::reload_network_from_file(const std::string& path_in) {
	fs::path path = path_in;
	const char* mode_dir = mode == ETestbedMode::Image ? "image" : mode == ETestbedMode::Sdf ? "sdf" : "nerf"; // get_filename_in_data_path_with_suffix / to_string(mode), testbed.cu:254-270
	if (path_in.empty()) { path = fs::path(root_dir) / "configs" / mode_dir / "base.json"; if (!fs::exists(path)) path = fs::path(s_default_root_dir) / "configs" / mode_dir / "base.json"; }
	else if (!fs::exists(path)) {
		// relative names resolve against configs/<mode>/ (testbed.cu:254-270)
		fs::path alt = fs::path(root_dir) / "configs" / mode_dir / path_in;
		if (!fs::exists(alt) && alt.extension().empty()) alt += ".json";
		if (fs::exists(alt)) path = alt; else throw std::runtime_error{"Network config '" + path_in + "' does not exist."};
	}
	m_network_config = load_config_recursive(path);
	m_network_config_path = path.string();
	reset_network();
}

// reload_network_from_json, testbed.cu:346-351: the config as a JSON document; a "parent" entry resolves against config_base_path (a file name, or a directory with a trailing slash)
void Testbed::reload_network_from_json_text(const std::string& json_text, const std::string& config_base_path) {
	mini_json::Value v; std::string err;
	if (!mini_json::parse(json_text.c_str(), v, err)) throw std::runtime_error{"reload_network_from_json: " + err};
	if (v.has("parent")) {
		mini_json::Value parent = load_config_recursive(fs::path(config_base_path).parent_path() / v.str("parent", ""), 1);
		mini_json::merge_patch(parent, v);
		v = parent;
	}
	m_network_config = v;
	reset_network();
}

void Testbed::reset_network() {
	destroy_trainer(); // rebuilt lazily with the current dataset / options
	training_step = 0;
	loss = 0.f;
}

ngp_aabb Testbed::scene_aabb() const { // testbed_nerf.cu:2424-2425
	const float h = 0.5f * (float)std::min(128, nerf.training.dataset.aabb_scale);
	return ngp_aabb{{0.5f - h, 0.5f - h, 0.5f - h}, {0.5f + h, 0.5f + h, 0.5f + h}};
}

ngp_nerf_options Testbed::current_options() const {
	ngp_nerf_options o; memset(&o, 0, sizeof(o));
	o.rgb_activation = nerf.rgb_activation >= 0 ? nerf.rgb_activation : (nerf.training.dataset.is_hdr ? NGP_ACT_EXPONENTIAL : NGP_ACT_LOGISTIC); // testbed_nerf.cu:2354 unless set from Python
	o.density_activation = nerf.density_activation;
	const std::string lt = lower(m_network_config["loss"].str("otype", "L2")); // string_to_loss_type, testbed.cu:4209
	o.loss_type = lt == "huber" ? NGP_LOSS_HUBER : lt == "l1" ? NGP_LOSS_L1 : lt == "mape" ? NGP_LOSS_MAPE : lt == "smape" ? NGP_LOSS_SMAPE :
		lt == "logl1" ? NGP_LOSS_LOGL1 : lt == "relativel2" ? NGP_LOSS_RELATIVE_L2 : NGP_LOSS_L2;
	if (nerf.training.loss_type >= 0) o.loss_type = nerf.training.loss_type; // nerf.training.loss_type written from Python (python_api.cu:785)
	o.random_bg_color = nerf.training.random_bg_color;
	o.snap_to_pixel_centers = nerf.training.snap_to_pixel_centers;
	o.linear_colors = nerf.training.linear_colors;
	o.color_space_srgb = color_space == EColorSpace::SRGB;
	for (int k = 0; k < 3; ++k) o.background_color[k] = background_color[k];
	o.near_distance = nerf.training.near_distance;
	o.density_grid_decay = nerf.training.density_grid_decay;
	o.cone_angle_constant = nerf.cone_angle_constant;
	o.max_cascade = (uint32_t)nerf.max_cascade;
	o.target_batch_size = training_batch_size;
	o.loss_scale = 128.f; // default_loss_scale<__half>, testbed.h:307-311
	o.seed = seed;
	o.rank = m_rank; o.world_size = m_world_size;
	// Rfl / RflRelax: the reference needs its JIT-fused kernel for these (testbed_nerf.cu:3091-3094); here K3 evaluates their gradients
	o.train_mode = nerf.training.train_mode == ETrainMode::Rfl ? 1 : nerf.training.train_mode == ETrainMode::RflRelax ? 2 : 0;
	o.depth_supervision_lambda = nerf.training.depth_supervision_lambda; o.depth_loss_type = nerf.training.depth_loss_type;
	o.sample_focal_plane_proportional_to_error = nerf.training.sample_focal_plane_proportional_to_error; o.sample_image_proportional_to_error = nerf.training.sample_image_proportional_to_error;
	o.accumulate_error_map = nerf.training.accumulate_error_map;
	return o;
}

void Testbed::ensure_trainer() {
	if (nerf.training.dataset.n_images == 0) throw std::runtime_error{"No training data loaded."};
	if (m_network_config.type != mini_json::Value::Object) reload_network_from_file("");
	if (!m_model) {
		ngp_model_config cfg;
		NGP_CHECK(ngp_model_config_from_json(mini_json::dump(m_network_config).c_str(), (uint32_t)nerf.training.dataset.aabb_scale, nerf.training.dataset.n_extra_dims(), &cfg)); // testbed.cu:4281-4294: NerfNetwork(..., n_extra_dims, ...)
		NGP_CHECK(ngp_model_create(&cfg, seed, &m_model));
	}
	if (!m_nerf) {
		ngp_nerf_options o = current_options();
		NGP_CHECK(ngp_nerf_create(m_model, &o, scene_aabb(), &m_nerf));
		m_dataset_dirty = true;
		m_comm_up = false;
	}
	if (m_world_size > 1 && !m_comm_up) {
		if (m_comm_id.size() != 128)
			throw std::runtime_error{"The data-parallel communicator went away with the trainer it belonged to (load_snapshot / reload_network_from_file / new training data): "
				"call comm_unique_id() on rank 0 and comm_init(rank, world_size, id) on every rank again before training resumes."};
		NGP_CHECK(ngp_comm_init(m_nerf, m_rank, m_world_size, (const uint8_t*)m_comm_id.data()));
		m_comm_up = true;
	}
	if (nerf.training.n_images_for_training != m_uploaded_n_images_for_training) m_dataset_dirty = true; // written from Python since the last upload
	if (m_dataset_dirty) {
		const NerfDataset& d = nerf.training.dataset;
		std::vector<ngp_image_meta> meta(d.n_images);
		std::vector<ngp_xform> xf(d.n_images);
		std::vector<const void*> pix(d.n_images);
		for (size_t i = 0; i < d.n_images; ++i) {
			memset(&meta[i], 0, sizeof(ngp_image_meta));
			meta[i].image_data_type = NGP_IMAGE_BYTE; meta[i].lens_mode = d.metadata[i].lens_mode;
			for (int k = 0; k < 2; ++k) { meta[i].resolution[k] = d.metadata[i].resolution[k]; meta[i].principal_point[k] = d.metadata[i].principal_point[k]; meta[i].focal_length[k] = d.metadata[i].focal_length[k]; }
			for (int k = 0; k < 7; ++k) meta[i].lens_params[k] = d.metadata[i].lens_params[k];
			for (int k = 0; k < 12; ++k) { xf[i].start[k] = d.xforms[i][k]; xf[i].end[k] = i < d.xforms_end.size() ? d.xforms_end[i][k] : d.xforms[i][k]; }
			for (int k = 0; k < 4; ++k) meta[i].rolling_shutter[k] = d.metadata[i].rolling_shutter[k];
			pix[i] = d.pixels[i].data();
			if (i < d.depth.size() && !d.depth[i].empty()) meta[i].depth = d.depth[i].data(); // host pointer: ngp_nerf_set_dataset_host uploads it
			if (i < d.pixels_half.size() && !d.pixels_half[i].empty()) { meta[i].image_data_type = NGP_IMAGE_HALF; pix[i] = d.pixels_half[i].data(); } // sharpened at load time
			if (i < d.pixels_float.size() && !d.pixels_float[i].empty()) { meta[i].image_data_type = NGP_IMAGE_FLOAT; pix[i] = d.pixels_float[i].data(); } // set_image
			const bool no_pixels = d.pixels[i].empty() && !(i < d.pixels_float.size() && !d.pixels_float[i].empty()) && !(i < d.pixels_half.size() && !d.pixels_half[i].empty());
			if (no_pixels) { // metadata restored from a snapshot, or a slot of create_empty_nerf_dataset that set_image has not filled yet: a 1x1 transparent stand-in keeps the device arrays well formed (render-only)
				static const uint32_t k_no_pixel = 0u;
				pix[i] = &k_no_pixel; meta[i].resolution[0] = meta[i].resolution[1] = 1;
			}
		}
		// rays are drawn from the first n_images_for_training images (testbed_nerf.cu:3075: n_training_images = nerf.training.n_images_for_training)
		const uint32_t n_train = nerf.training.n_images_for_training > 0 ? (uint32_t)std::min<size_t>((size_t)nerf.training.n_images_for_training, d.n_images) : (uint32_t)d.n_images;
		NGP_CHECK(ngp_nerf_set_dataset_host(m_nerf, n_train, meta.data(), xf.data(), pix.data()));
		m_dataset_dirty = false; m_uploaded_n_images_for_training = nerf.training.n_images_for_training;
		if (d.n_extra_dims() > 0 && !m_extra_dims_installed) {
			// Testbed::Nerf::reset_extra_dims as reset_network calls it (testbed.cu:4163-4272), once per trainer: m_rng = rng{seed}; density_grid_rng = rng{m_rng.next_uint()};
			// reset_extra_dims(m_rng) -- the latents are drawn from the TRAINER's ray-stream rng behind the density-grid draw (where ngp_nerf_create leaves it) and advance it, so
			// the initial latents and every training ray after them are the reference's.  Values for EVERY image of the dataset (extra_dims_gpu holds dataset.n_images + 1 vectors).
			ngp_pcg32 rng, grid_rng;
			NGP_CHECK(ngp_nerf_get_rng(m_nerf, &rng, &grid_rng));
			const std::vector<float> e = initial_extra_dims(rng);
			NGP_CHECK(ngp_nerf_set_extra_dims(m_nerf, e.data(), (uint32_t)d.n_images));
			NGP_CHECK(ngp_nerf_set_rng(m_nerf, &rng));
			nerf.rendering_extra_dims_default.assign(e.begin(), e.begin() + d.n_extra_dims());
			m_extra_dims_installed = true;
		}
	}
}

void Testbed::push_options() {
	ngp_nerf_options o = current_options();
	NGP_CHECK(ngp_nerf_set_options(m_nerf, &o));
	if (nerf.training.dataset.n_extra_dims() > 0) {
		// testbed_nerf.cu:2743: the latents train when the dataset has learnable dims and the switch is on (light directions alone are fixed inputs)
		NGP_CHECK(ngp_nerf_set_optimize_extra_dims(m_nerf, nerf.training.dataset.n_extra_learnable_dims > 0 && nerf.training.optimize_extra_dims ? 1 : 0));
		const bool explicit_vals = nerf.rendering_extra_dims_from_training_view < 0 && nerf.rendering_extra_dims.size() == nerf.training.dataset.n_extra_dims();
		NGP_CHECK(ngp_nerf_set_rendering_extra_dims(m_nerf, nerf.rendering_extra_dims_from_training_view, explicit_vals ? nerf.rendering_extra_dims.data() : nullptr));
		NGP_CHECK(ngp_nerf_set_light_dir(m_nerf, nerf.training.dataset.has_light_dirs ? 1 : 0, nerf.light_dir.data())); // get_rendering_extra_dims, testbed_nerf.cu:3697-3706
	}
}

// Testbed::Nerf::reset_extra_dims (testbed_nerf.cu:3656-3683): per image the warped (normalised) light direction in the first three dims when the dataset has light
// directions, uniform random values in [-1, 1) otherwise (random_val(rng) * 2 - 1 = pcg32::next_float [tcnn pcg32.h]), drawn image by image, dim by dim from `rng`, which is
// left advanced by one draw per learnable dim -- the caller hands in the rng the reference hands in (reset_network's m_rng behind the density-grid draw, testbed.cu:4163-4272)
std::vector<float> Testbed::initial_extra_dims(ngp_pcg32& rng) const {
	const NerfDataset& d = nerf.training.dataset;
	const uint32_t n = d.n_extra_dims();
	std::vector<float> out((size_t)d.n_images * n);
	auto next_uint = [&]() { const uint64_t old = rng.state; rng.state = old * 0x5851f42d4c957f2dULL + rng.inc; const uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u); return (xs >> rot) | (xs << ((~rot + 1u) & 31)); };
	auto next_float = [&]() { union { uint32_t u; float f; } x; x.u = (next_uint() >> 9) | 0x3f800000u; return x.f - 1.0f; };
	for (size_t i = 0; i < d.n_images; ++i) {
		const auto& l = d.metadata[i].light_dir;
		const float len = std::sqrt(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]);
		for (uint32_t j = 0; j < n; ++j) {
			if (d.has_light_dirs && j < 3) out[i * n + j] = (l[j] / len + 1.0f) * 0.5f; // warp_direction(normalize(light_dir))
			else out[i * n + j] = next_float() * 2.0f - 1.0f;
		}
	}
	return out;
}
std::vector<float> Testbed::get_extra_dims(int trainview) { // Training::get_extra_dims_cpu, testbed_nerf.cu:1862-1877
	const uint32_t n = nerf.training.dataset.n_extra_dims();
	if (n == 0) return {};
	if (trainview < 0 || (size_t)trainview >= nerf.training.dataset.n_images) throw std::runtime_error{"Invalid training view."};
	ensure_trainer();
	std::vector<float> all((size_t)(trainview + 1) * n);
	NGP_CHECK(ngp_nerf_get_extra_dims(m_nerf, all.data(), (uint32_t)trainview + 1));
	return std::vector<float>(all.end() - n, all.end());
}

// ------------------------------------------------------------------------------------------------
// dataset: transforms.json (+ images), nerf_loader.cu:273-747
// ------------------------------------------------------------------------------------------------
namespace { uint16_t f32_to_f16(float f); float f16_to_f32(uint16_t h); }
// NerfDataset::set_training_image's sharpening (nerf_loader.cu:85-105, 805-827): the RGBA8 image becomes linear premultiplied halfs
// (from_rgba32, common_device.cuh:699-731) and is filtered with the 5-point stencil {center_w, -1, -1, -1, -1} / (center_w - 4),
// center_w = 4 + 1 / amount, on the FLAT pixel index (left / up neighbours clamp at 0, right / down ones wrap) -- on the host here,
// the reference runs it on the device at load time.  The trainer then samples the half image (EImageDataType::Half).
std::vector<uint16_t> sharpen_half(const std::vector<uint16_t>& src, int w, int h, float amount) {
	const int64_t n = (int64_t)w * h;
	std::vector<uint16_t> dst((size_t)n * 4);
	const float center_w = 4.f + 1.f / amount, inv_totalw = 1.f / (center_w - 4.f);
	for (int64_t i = 0; i < n; ++i) {
		int64_t nb[4] = {i - 1, i - w, i + 1, i + w};
		if (nb[0] < 0) nb[0] = 0;
		if (nb[1] < 0) nb[1] = 0;
		if (nb[2] >= n) nb[2] -= n;
		if (nb[3] >= n) nb[3] -= n;
		for (int c = 0; c < 4; ++c) {
			float v = f16_to_f32(src[i * 4 + c]) * center_w;
			for (int k = 0; k < 4; ++k) v -= f16_to_f32(src[nb[k] * 4 + c]);
			dst[i * 4 + c] = f32_to_f16(std::max(0.f, v * inv_totalw));
		}
	}
	return dst;
}
// has_mask: the image came with a dynamic mask; its hot-pink pixels (mask_color 0x00FF00FF) become -1 before the filter, as from_rgba32 makes them (common_device.cuh:729-731)
std::vector<uint16_t> sharpen_rgba8(const std::vector<uint8_t>& rgba, int w, int h, float amount, bool has_mask) {
	const int64_t n = (int64_t)w * h;
	std::vector<uint16_t> src((size_t)n * 4);
	auto s2l = [](float s) { return s <= 0.04045f ? s / 12.92f : std::pow((s + 0.055f) / 1.055f, 2.4f); };
	for (int64_t i = 0; i < n; ++i) {
		const float alpha = rgba[i * 4 + 3] * (1.0f / 255.0f);
		for (int c = 0; c < 3; ++c) src[i * 4 + c] = f32_to_f16(s2l(rgba[i * 4 + c] * (1.0f / 255.0f)) * alpha);
		src[i * 4 + 3] = f32_to_f16(alpha);
		if (has_mask && rgba[i * 4] == 0xFF && rgba[i * 4 + 1] == 0x00 && rgba[i * 4 + 2] == 0xFF && rgba[i * 4 + 3] == 0x00) for (int c = 0; c < 4; ++c) src[i * 4 + c] = f32_to_f16(-1.0f);
	}
	return sharpen_half(src, w, h, amount);
}
std::vector<uint16_t> Testbed::sharpen_rgba8_for_tests(const std::vector<uint8_t>& rgba, int w, int h, float amount, bool has_mask) { return sharpen_rgba8(rgba, w, h, amount, has_mask); }

void Testbed::load_training_data(const std::string& path_in) {
	fs::path path = path_in;
	if (!fs::exists(path)) throw std::runtime_error{"Data path '" + path_in + "' does not exist."};
	// mode_from_scene, common_host.cu:144-160: directory / json -> NeRF, obj / stl -> SDF, nvdb -> volume, anything else -> image
	const std::string ext = lower(path.extension().string());
	if (!fs::is_directory(path) && ext != ".json") {
		if (ext == ".nvdb") throw std::runtime_error{"Volume (.nvdb) scenes are out of scope of this build."};
		destroy_trainer();
		m_network_config = mini_json::Value{};
		if (ext == ".obj" || ext == ".stl") { // load_mesh, testbed_sdf.cu:1363-1447
			m_mesh = ext == ".stl" ? mesh_lite::load_stl(path.string()) : mesh_lite::load_obj(path.string());
			NGP_CHECK(ngp_sdf_normalize_mesh_host(m_mesh.data(), m_mesh.size() / 3, &m_mesh_aabb, &sdf.mesh_scale)); // m_sdf.mesh_scale, testbed_sdf.cu:1404
			aabb = raw_aabb = render_aabb = BoundingBox{{m_mesh_aabb.min[0], m_mesh_aabb.min[1], m_mesh_aabb.min[2]}, {m_mesh_aabb.max[0], m_mesh_aabb.max[1], m_mesh_aabb.max[2]}}; // :1408-1410
			mode = ETestbedMode::Sdf;
		} else { // load_image, testbed_image.cu:393-458: EXR and the .bin format natively (linear), PNG / JPEG natively and other 8-bit formats through the decoder hook (sRGB -> linear)
			if (ext == ".exr") exr_lite::read_rgba(path.string(), m_image_w, m_image_h, m_image_pixels);
			else if (ext == ".bin") { // load_binary_image, testbed_image.cu:439-458: int32 height, int32 width, then height x width RGBA halfs
				std::ifstream f{path, std::ios::binary};
				int32_t hw[2] = {0, 0};
				f.read((char*)hw, 8);
				if (!f || hw[0] <= 0 || hw[1] <= 0 || (uint64_t)hw[0] * (uint64_t)hw[1] > (1ull << 28)) throw std::runtime_error{"Could not load binary image '" + path.string() + "'"};
				m_image_h = hw[0]; m_image_w = hw[1];
				std::vector<uint16_t> halfs((size_t)m_image_w * m_image_h * 4);
				f.read((char*)halfs.data(), (std::streamsize)(halfs.size() * 2));
				if (f.gcount() != (std::streamsize)(halfs.size() * 2)) throw std::runtime_error{"Binary image '" + path.string() + "' is truncated"};
				m_image_pixels.resize(halfs.size());
				for (size_t i = 0; i < halfs.size(); ++i) m_image_pixels[i] = exr_lite::half_to_float(halfs[i]);
			} else {
				std::vector<uint8_t> rgba;
				bool ok = decode_builtin(path.string(), m_image_w, m_image_h, rgba);
				if (!ok && s_fallback_decoder) ok = s_fallback_decoder(path.string(), m_image_w, m_image_h, rgba);
				if (!ok) throw std::runtime_error{"Could not load image '" + path.string() + "'"};
				m_image_pixels.resize(rgba.size());
				for (size_t i = 0; i < rgba.size(); ++i) { const float v = rgba[i] / 255.f; m_image_pixels[i] = (i & 3) == 3 ? v : (v <= 0.04045f ? v / 12.92f : std::pow((v + 0.055f) / 1.055f, 2.4f)); }
			}
			mode = ETestbedMode::Image;
		}
		training_step = 0; loss = 0.f;
		return;
	}
	std::vector<fs::path> jsons;
	if (fs::is_directory(path)) {
		for (auto& e : fs::directory_iterator(path)) if (e.is_regular_file() && lower(e.path().extension().string()) == ".json") jsons.push_back(e.path());
		std::sort(jsons.begin(), jsons.end());
	} else if (lower(path.extension().string()) == ".json") jsons.push_back(path);
	else throw std::runtime_error{"NeRF data path must either be a json file or a directory containing json files."};
	if (jsons.empty()) throw std::runtime_error{"No json files found in '" + path_in + "'."};

	NerfDataset d;
	d.sharpen_amount = nerf.sharpen;
	bool white_transparent = false, black_transparent = false; // json flags (NSVF-style datasets): pure white / black pixels become transparent (convert_rgba32)
	bool fix_premult = false; // json "fix_premult" (nerf_loader.cu:448-450): EXR colours are multiplied by their alpha at load time
	struct Frame { std::string image_path, depth_path; float depth_scale = -1.f; std::array<float, 12> xform, xform_end; ImageMetadata meta; float angle_x = 0.f, angle_y = 0.f; bool principal_in_pixels = false; };
	bool enable_depth_loading = true; // nerf_loader.cu:419, 435-437
	std::vector<Frame> frames;
	for (const fs::path& jp : jsons) {
		mini_json::Value j; std::string err;
		if (!mini_json::parse(read_text(jp).c_str(), j, err)) throw std::runtime_error{jp.string() + ": " + err};
		if (j.has("aabb_scale")) d.aabb_scale = (int)j.num("aabb_scale", 1);
		if (j.has("sharpen")) d.sharpen_amount = (float)j.num("sharpen", 0);
		if (j.has("n_extra_learnable_dims")) d.n_extra_learnable_dims = (uint32_t)j.num("n_extra_learnable_dims", 0); // nerf_loader.cu:482-483
		if (j.has("render_aabb") && j["render_aabb"].size() == 2) // nerf_loader.cu:457-460: [[min], [max]] in the scene's coordinates, as is
			for (int k = 0; k < 3; ++k) { d.render_aabb.min[k] = (float)j["render_aabb"].at(0).at(k).n; d.render_aabb.max[k] = (float)j["render_aabb"].at(1).at(k).n; }
		if (j.has("up") && j["up"].size() == 3) d.up = {(float)j["up"].at(1).n, (float)j["up"].at(2).n, (float)j["up"].at(0).n}; // nerf_loader.cu:528-533: axes permuted like the transforms
		if (j.has("fix_premult")) fix_premult = j.boolean("fix_premult", false);
		if (j.has("enable_depth_loading")) enable_depth_loading = j.boolean("enable_depth_loading", true);
		const float integer_depth_scale = j.has("integer_depth_scale") ? (float)j.num("integer_depth_scale", -1.0) : -1.f; // nerf_loader.cu:490-492: units of the 16-bit depth images
		// nerf_loader.cu:439-514, in the reference's order: Mitsuba convention (its own default scale / offset), then explicit scale / offset, then an
		// "aabb" [[min],[max]] that is mapped isotropically onto the unit cube
		if (j.has("normal_mts_args")) d.from_mitsuba = true;
		if (j.has("from_mitsuba")) d.from_mitsuba = j.boolean("from_mitsuba", false);
		if (d.from_mitsuba) { d.scale = 0.66f; d.offset = {0.25f * d.scale, 0.25f * d.scale, 0.25f * d.scale}; }
		if (j.has("white_transparent")) white_transparent = j.boolean("white_transparent", false);
		if (j.has("black_transparent")) black_transparent = j.boolean("black_transparent", false);
		if (j.has("scale")) d.scale = (float)j.num("scale", 0.33);
		if (j.has("offset")) {
			if (j["offset"].size() == 3) for (int k = 0; k < 3; ++k) d.offset[k] = (float)j["offset"].at(k).n;
			else if (j["offset"].type == mini_json::Value::Number) d.offset = {(float)j["offset"].n, (float)j["offset"].n, (float)j["offset"].n};
		}
		if (j.has("aabb") && j["aabb"].size() == 2) {
			const auto& ab = j["aabb"];
			float len = 0.000001f, lo[3], hi[3];
			for (int k = 0; k < 3; ++k) { lo[k] = (float)ab.at(0).at(k).n; hi[k] = (float)ab.at(1).at(k).n; len = std::max(len, std::fabs(hi[k] - lo[k])); }
			d.scale = 1.f / len;
			for (int k = 0; k < 3; ++k) d.offset[k] = ((hi[k] + lo[k]) * 0.5f) * -d.scale + 0.5f;
		}
		const auto& fr = j["frames"];
		auto resolve = [&](const mini_json::Value& f) { // resolve_path, nerf_loader.cu:314-330: try the supported extensions when none is given
			fs::path ip = jp.parent_path() / f.str("file_path", "");
			if (!fs::exists(ip) || ip.extension().empty()) {
				for (const char* ext : {".png", ".jpg", ".jpeg", ".JPG", ".PNG", ".exr"}) { fs::path c = ip; c += ext; if (fs::exists(c)) { ip = c; break; } }
			}
			return ip;
		};
		// frame selection, nerf_loader.cu:352-388: natural sort by file_path, optional "n_frames" cull, and -- when the frames carry a
		// "sharpness" value (colmap2nerf.py) -- frames whose image is missing or blurrier than `sharpness_discard_threshold` x the mean
		// of their neighbourhood are dropped (fox: transforms.json lists 67 frames, 50 images are shipped)
		std::vector<const mini_json::Value*> sel;
		for (size_t i = 0; i < fr.size(); ++i) sel.push_back(&fr.at(i));
		std::stable_sort(sel.begin(), sel.end(), [](const mini_json::Value* a, const mini_json::Value* b) { return natural_less(a->str("file_path", ""), b->str("file_path", "")); });
		if (j.has("n_frames")) sel.resize(std::min(sel.size(), (size_t)j.num("n_frames", 0)));
		if (!sel.empty() && sel[0]->has("sharpness")) {
			const float thresh = (float)j.num("sharpness_discard_threshold", 0.0);
			std::vector<const mini_json::Value*> kept;
			const int nb = 3, n = (int)sel.size();
			for (int i = 0; i < n; ++i) {
				const int b = std::max(0, i - nb), e = std::min(i + nb, n - 1);
				float mean = 0.f;
				for (int k = b; k < e; ++k) mean += (float)sel[k]->num("sharpness", 1.0);
				mean /= (float)(e - b);
				if (fs::exists(resolve(*sel[i])) && (float)sel[i]->num("sharpness", 1.0) > thresh * mean) kept.push_back(sel[i]);
			}
			sel.swap(kept);
		}
		for (const mini_json::Value* fp : sel) {
			const auto& f = *fp;
			Frame F;
			F.image_path = resolve(f).string();
			if (enable_depth_loading && integer_depth_scale > 0.f && f.has("depth_path")) { // nerf_loader.cu:629-641
				std::string dp = f.str("depth_path", ""); for (auto& c : dp) if (c == '\\') c = '/';
				F.depth_path = (jp.parent_path() / dp).string(); F.depth_scale = integer_depth_scale;
			}
			// per-frame values override the global ones (nerf_loader.cu:487-535, 697-700)
			auto get = [&](const char* k, double dflt) { return f.has(k) ? f.num(k, dflt) : j.num(k, dflt); };
			auto has = [&](const char* k) { return f.has(k) || j.has(k); };
			F.meta.resolution = {(int)get("w", 0), (int)get("h", 0)};
			double flx = 0, fly = 0;
			if (has("fl_x")) flx = get("fl_x", 0);
			if (has("fl_y")) fly = get("fl_y", 0);
			F.meta.focal_length = {(float)flx, (float)fly}; // resolved after the image size is known
			if (has("cx")) F.meta.principal_point[0] = -1.f; // marker, resolved below
			// nerf_loader.cu:668-669, 689-694: "transform_matrix_start" / "transform_matrix_end" bracket the exposure of a moving camera; both default to "transform_matrix"
			const auto& tm_start = f.has("transform_matrix_start") ? f["transform_matrix_start"] : f["transform_matrix"];
			const auto& tm_end = f.has("transform_matrix_end") ? f["transform_matrix_end"] : tm_start;
			auto to_ngp = [&](const mini_json::Value& tm, std::array<float, 12>& out) {
				float m[3][4];
				for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) m[r][c] = (float)tm.at(r).at(c).n;
				// nerf_matrix_to_ngp, nerf_loader.h:101-120
				for (int r = 0; r < 3; ++r) { m[r][1] *= -1.f; m[r][2] *= -1.f; m[r][3] = m[r][3] * d.scale + d.offset[r]; }
				float c3[3][4];
				if (d.from_mitsuba) { for (int r = 0; r < 3; ++r) { m[r][0] *= -1.f; m[r][2] *= -1.f; } for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) c3[r][c] = m[r][c]; }
				else for (int c = 0; c < 4; ++c) { c3[0][c] = m[1][c]; c3[1][c] = m[2][c]; c3[2][c] = m[0][c]; } // cycle the axes xyz <- yzx
				for (int c = 0; c < 4; ++c) for (int r = 0; r < 3; ++r) out[c * 3 + r] = c3[r][c];
			};
			to_ngp(tm_start, F.xform); to_ngp(tm_end, F.xform_end);
			if (f.has("driver_parameters")) { // nerf_loader.cu:671-680: a light direction per frame replaces learnable dims
				const auto& dp = f["driver_parameters"];
				float l[3] = {(float)dp.num("LightX", 0), (float)dp.num("LightY", 0), (float)dp.num("LightZ", 0)};
				const float len = std::sqrt(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]);
				for (float& v : l) v /= len;
				// nerf_direction_to_ngp (nerf_loader.h:141-152): y and z flipped, then -- unless from_mitsuba -- the axes cycled xyz <- yzx
				l[1] *= -1.f; l[2] *= -1.f;
				if (d.from_mitsuba) { l[0] *= -1.f; l[2] *= -1.f; F.meta.light_dir = {l[0], l[1], l[2]}; } else F.meta.light_dir = {l[1], l[2], l[0]};
				d.has_light_dirs = true; d.n_extra_learnable_dims = 0;
			}
			{ // "rolling_shutter": [a, b, c, d?] (nerf_loader.cu:204-215), global with per-frame override (read_lens :699)
				const mini_json::Value* rs = f.has("rolling_shutter") ? &f["rolling_shutter"] : (j.has("rolling_shutter") ? &j["rolling_shutter"] : nullptr);
				if (rs && rs->size() >= 3) F.meta.rolling_shutter = {(float)rs->at(0).n, (float)rs->at(1).n, (float)rs->at(2).n, rs->size() >= 4 ? (float)rs->at(3).n : 0.f};
			}
			{ // read_lens, nerf_loader.cu:175-241: OpenCV / OpenCV-fisheye coefficients select their mode only when one of them is non-zero
				const int opencv_mode = (has("is_fisheye") && get("is_fisheye", 0) != 0) ? NGP_LENS_OPENCV_FISHEYE : NGP_LENS_OPENCV;
				auto coeff = [&](const char* name, int idx) { if (has(name)) { F.meta.lens_params[idx] = (float)get(name, 0); if (F.meta.lens_params[idx] != 0.f) F.meta.lens_mode = opencv_mode; } };
				coeff("k1", 0); coeff("k2", 1); coeff("k3", 2); coeff("k4", 3); coeff("p1", 2); coeff("p2", 3);
				if (has("ftheta_p0")) { // polynomial in the pixel radius + the resolution the intrinsics refer to (stored behind the coefficients)
					for (int k = 0; k < 5; ++k) F.meta.lens_params[k] = (float)get((std::string("ftheta_p") + char('0' + k)).c_str(), 0);
					F.meta.lens_params[5] = (float)get("w", 0); F.meta.lens_params[6] = (float)get("h", 0);
					F.meta.lens_mode = NGP_LENS_FTHETA;
				}
				if (has("latlong")) F.meta.lens_mode = NGP_LENS_LATLONG;
				else if (has("equirectangular")) F.meta.lens_mode = NGP_LENS_EQUIRECTANGULAR;
				else if (has("orthographic")) F.meta.lens_mode = NGP_LENS_ORTHOGRAPHIC;
			}
			// stash intrinsics needing the resolution
			F.angle_x = (float)(has("camera_angle_x") ? get("camera_angle_x", 0) : 0);
			F.angle_y = (float)(has("camera_angle_y") ? get("camera_angle_y", 0) : 0);
			if (has("cx")) { F.meta.principal_point = {(float)get("cx", 0), (float)get("cy", 0)}; F.principal_in_pixels = true; }
			frames.push_back(F);
		}
	}
	std::sort(frames.begin(), frames.end(), [](const Frame& a, const Frame& b) { return natural_less(a.image_path, b.image_path); });
	for (Frame& F : frames) {
		int w = 0, h = 0; std::vector<uint8_t> rgba; bool has_mask = false;
		std::vector<uint16_t> hdr_half; // EXR frames: linear RGBA halfs (EImageDataType::Half), nerf_loader.cu:569-573 + tinyexr_wrapper.cu:39-53
		const std::string ext_l = lower(fs::path(F.image_path).extension().string());
		bool ok = false;
		if (ext_l == ".exr") {
			std::vector<float> f;
			exr_lite::read_rgba(F.image_path, w, h, f);
			hdr_half.resize((size_t)w * h * 4); rgba.resize((size_t)w * h * 4);
			for (size_t i = 0; i < (size_t)w * h; ++i) {
				const float alpha = f[i * 4 + 3], fix = fix_premult ? alpha : 1.0f;
				for (int c = 0; c < 3; ++c) {
					const float v = f[i * 4 + c] * fix;
					hdr_half[i * 4 + c] = f32_to_f16(v);
					const float s8 = v < 0.0031308f ? 12.92f * v : 1.055f * std::pow(v, 0.41666f) - 0.055f; // 8-bit sRGB preview for render_ground_truth
					rgba[i * 4 + c] = (uint8_t)std::lround(255.f * std::min(std::max(s8, 0.f), 1.f));
				}
				hdr_half[i * 4 + 3] = f32_to_f16(alpha);
				rgba[i * 4 + 3] = (uint8_t)std::lround(255.f * std::min(std::max(alpha, 0.f), 1.f));
			}
			d.is_hdr = true; ok = true;
		}
		if (!ok) ok = decode_builtin(F.image_path, w, h, rgba);
		if (!ok && s_fallback_decoder) ok = s_fallback_decoder(F.image_path, w, h, rgba);
		if (!ok) throw std::runtime_error{"Could not load image '" + F.image_path + "'"};
		if (hdr_half.empty()) { // 8-bit images (nerf_loader.cu:581-620, convert_rgba32 :41-63)
			const fs::path ip = F.image_path;
			auto decode_any = [&](const fs::path& p, int& ww, int& hh, std::vector<uint8_t>& px) {
				bool good = decode_builtin(p.string(), ww, hh, px);
				if (!good && s_fallback_decoder) good = s_fallback_decoder(p.string(), ww, hh, px);
				return good;
			};
			const fs::path alpha_path = ip.parent_path() / (ip.stem().string() + ".alpha" + ip.extension().string());
			if (fs::exists(alpha_path)) { // red channel of <name>.alpha.<ext> (sRGB -> linear) replaces the alpha channel
				int wa = 0, ha = 0; std::vector<uint8_t> a;
				if (!decode_any(alpha_path, wa, ha, a)) throw std::runtime_error{"Could not load alpha image " + alpha_path.string()};
				if (wa != w || ha != h) throw std::runtime_error{"Alpha image " + alpha_path.string() + " has wrong resolution."};
				for (size_t i = 0; i < (size_t)w * h; ++i) { const float s = a[i * 4] * (1.f / 255.f); rgba[i * 4 + 3] = (uint8_t)(255.0f * (s <= 0.04045f ? s / 12.92f : std::pow((s + 0.055f) / 1.055f, 2.4f))); }
			}
			const fs::path mask_path = ip.parent_path() / ("dynamic_mask_" + ip.stem().string() + ".png");
			has_mask = fs::exists(mask_path);
			if (has_mask) { // masked pixels become "hot pink" 0x00FF00FF, which read_rgba reports as "no pixel": such rays are not trained
				int wa = 0, ha = 0; std::vector<uint8_t> mk;
				if (!decode_any(mask_path, wa, ha, mk)) throw std::runtime_error{"Dynamic mask " + mask_path.string() + " could not be loaded."};
				if (wa != w || ha != h) throw std::runtime_error{"Dynamic mask " + mask_path.string() + " has wrong resolution."};
				for (size_t i = 0; i < (size_t)w * h; ++i) if (mk[i * 4] || mk[i * 4 + 1] || mk[i * 4 + 2]) { rgba[i * 4] = 0xFF; rgba[i * 4 + 1] = 0x00; rgba[i * 4 + 2] = 0xFF; rgba[i * 4 + 3] = 0x00; }
			}
			if (white_transparent || black_transparent)
				for (size_t i = 0; i < (size_t)w * h; ++i) {
					const uint8_t* q = &rgba[i * 4];
					if ((white_transparent && q[0] == 255 && q[1] == 255 && q[2] == 255) || (black_transparent && q[0] == 0 && q[1] == 0 && q[2] == 0)) rgba[i * 4 + 3] = 0;
				}
		}
		F.meta.resolution = {w, h};
		const float ax = F.angle_x, ay = F.angle_y;
		float flx = F.meta.focal_length[0], fly = F.meta.focal_length[1];
		if (flx <= 0 && ax > 0) flx = 0.5f * (float)w / std::tan(0.5f * ax); // nerf_loader.cu:256-263
		if (fly <= 0 && ay > 0) fly = 0.5f * (float)h / std::tan(0.5f * ay);
		if (flx <= 0 && fly > 0) flx = fly;
		if (fly <= 0) fly = flx;
		if (flx <= 0) throw std::runtime_error{"Couldn't read fov / focal length for '" + F.image_path + "'"};
		F.meta.focal_length = {flx, fly};
		if (F.principal_in_pixels) F.meta.principal_point = {F.meta.principal_point[0] / (float)w, F.meta.principal_point[1] / (float)h};
		else F.meta.principal_point = {0.5f, 0.5f};
		d.pixels_half.emplace_back();
		if (!hdr_half.empty()) d.pixels_half.back() = d.sharpen_amount > 0.f ? sharpen_half(hdr_half, w, h, d.sharpen_amount) : std::move(hdr_half);
		else if (d.sharpen_amount > 0.f) d.pixels_half.back() = sharpen_rgba8(rgba, w, h, d.sharpen_amount, has_mask);
		d.depth.emplace_back();
		if (!F.depth_path.empty() && fs::exists(F.depth_path)) { // copy_depth, nerf_loader.cu:73-82: float depth = integer depth * (integer_depth_scale * dataset scale); 0 = no measurement
			int wa = 0, ha = 0; std::vector<uint16_t> dp;
			if (!decode_png_gray16(F.depth_path, wa, ha, dp)) throw std::runtime_error{"Could not load depth image '" + F.depth_path + "'."};
			if (wa != w || ha != h) throw std::runtime_error{"Depth image " + F.depth_path + " has wrong resolution."};
			std::vector<float>& dst = d.depth.back();
			dst.resize(dp.size());
			const float sc = F.depth_scale * d.scale; // (the dataset scale is final here: it is read from the json before its frames)
			for (size_t i = 0; i < dp.size(); ++i) dst[i] = (float)dp[i] * sc;
		}
		d.metadata.push_back(F.meta); d.xforms.push_back(F.xform); d.xforms_end.push_back(F.xform_end); d.pixels.push_back(std::move(rgba)); d.paths.push_back(F.image_path);
	}
	d.n_images = d.metadata.size();
	if ((d.aabb_scale & (d.aabb_scale - 1)) != 0) throw std::runtime_error{"NeRF dataset's `aabb_scale` must be a power of two."};
	if (d.aabb_scale > 128) throw std::runtime_error{"NeRF dataset must have `aabb_scale <= 128`."};

	const int prev_scale = nerf.training.dataset.aabb_scale; const uint32_t prev_extra = nerf.training.dataset.n_extra_dims();
	const bool had = nerf.training.dataset.n_images > 0;
	nerf.training.dataset = std::move(d);
	mode = ETestbedMode::Nerf;
	load_nerf_post();
	if (had && (prev_scale != nerf.training.dataset.aabb_scale || prev_extra != nerf.training.dataset.n_extra_dims())) destroy_trainer(); // network size depends on aabb_scale (testbed_nerf.cu:2466-2470) and on the extra dims (testbed.cu:4281-4294)
	m_dataset_dirty = true;
	m_render_lens_mode = nerf.training.dataset.metadata[0].lens_mode;
	m_render_lens_params = nerf.training.dataset.metadata[0].lens_params;
}

// load_nerf_post, testbed_nerf.cu:2353-2443: everything the Testbed derives from a freshly loaded (or snapshot-restored) dataset
void Testbed::load_nerf_post() {
	const NerfDataset& d = nerf.training.dataset;
	nerf.training.n_images_for_training = (int)d.n_images;                  // :2371
	const ngp_aabb box = scene_aabb();                                      // :2424-2425
	aabb = BoundingBox{{box.min[0], box.min[1], box.min[2]}, {box.max[0], box.max[1], box.max[2]}};
	raw_aabb = aabb; render_aabb = aabb; render_aabb_to_local = d.render_aabb_to_local; // :2426-2428
	if (!d.render_aabb.is_empty()) render_aabb = d.render_aabb.intersection(aabb);       // :2429-2431
	nerf.max_cascade = 0;
	while ((1 << nerf.max_cascade) < d.aabb_scale) ++nerf.max_cascade;
	nerf.cone_angle_constant = d.aabb_scale <= 1 ? 0.0f : (1.0f / 256.0f);
	nerf.training.optimize_extra_dims = d.n_extra_learnable_dims > 0;       // :2379
	up_dir = d.up;                                                          // :2443
}

void Testbed::load_file(const std::string& path) {
	const std::string ext = lower(fs::path(path).extension().string());
	if (fs::is_directory(path)) { load_training_data(path); return; }
	if (ext == ".ingp" || ext == ".msgpack" || ext == ".snap") { load_snapshot(path); return; }
	if (ext == ".json") {
		mini_json::Value j; std::string err;
		if (!mini_json::parse(read_text(path).c_str(), j, err)) throw std::runtime_error{path + ": " + err};
		// testbed.cu:371-395 in the reference's order: network config, camera path, otherwise training data
		if (j.has("parent") || j.has("network") || j.has("encoding") || j.has("loss") || j.has("optimizer")) reload_network_from_file(path);
		else if (j.has("path")) load_camera_path(path);
		else if (j.has("frames")) load_training_data(path);   // a NeRF scene (mode_from_scene, common_host.cu:144-160)
		else throw std::runtime_error{"File '" + path + "' is not a recognised scene / config / camera-path json."};
		return;
	}
	if (fs::exists(path)) { load_training_data(path); return; } // a mesh (SDF) or an image: mode_from_scene, common_host.cu:144-160
	throw std::runtime_error{"File '" + path + "' does not exist."};
}

// ------------------------------------------------------------------------------------------------
// training
// ------------------------------------------------------------------------------------------------
// NetworkWithInputEncoding + Trainer of the image / SDF modes from the network config (reset_network, testbed.cu:4160-4412)
void Testbed::ensure_encmlp_trainer() {
	if (m_image || m_sdf) return;
	if (m_encmlp) { ngp_encmlp_destroy(m_encmlp); m_encmlp = nullptr; } // left over from a creation that failed after the model was built (bad batch size, out of memory): never stack a second model on it
	if (m_network_config.type != mini_json::Value::Object) reload_network_from_file("");
	const bool image = mode == ETestbedMode::Image;
	const auto& enc = m_network_config["encoding"]; const auto& net = m_network_config["network"];
	ngp_encmlp_config c; memset(&c, 0, sizeof(c));
	c.n_pos_dims = image ? 2 : 3; c.n_output_dims = image ? 3 : 1;
	c.n_features_per_level = (uint32_t)enc.num("n_features_per_level", 2); c.n_levels = (uint32_t)enc.num("n_levels", 16);
	c.log2_hashmap_size = (uint32_t)enc.num("log2_hashmap_size", 15); c.base_resolution = (uint32_t)enc.num("base_resolution", 16);
	c.per_level_scale = (float)enc.num("per_level_scale", 0.0);
	if (c.per_level_scale <= 0.f && c.n_levels > 1) { // testbed.cu:4241-4255: finest level = half the image's larger side / 2048 for a unit-cube SDF
		const float desired = image ? (float)std::max(m_image_w, m_image_h) / 2.0f : 2048.0f;
		c.per_level_scale = std::exp(std::log(desired / (float)c.base_resolution) / (float)(c.n_levels - 1));
	}
	c.n_neurons = (uint32_t)net.num("n_neurons", 64); c.n_hidden_layers = (uint32_t)net.num("n_hidden_layers", 2);
	NGP_CHECK(ngp_encmlp_create(&c, seed, &m_encmlp));
	ngp_optimizer_config oc{1e-3f, 0.9f, 0.999f, 1e-8f, 1e-8f, 0.f, 0, 0, 1.f};
	const mini_json::Value* o = &m_network_config["optimizer"];
	for (int depth = 0; depth < 4 && o->is_object(); ++depth) {
		const std::string t = lower(o->str("otype", ""));
		if (t == "ema") oc.ema_decay = (float)o->num("decay", 0.99);
		else if (t == "exponentialdecay") { oc.decay_start = (uint32_t)o->num("decay_start", 10000); oc.decay_interval = (uint32_t)o->num("decay_interval", 10000); oc.decay_base = (float)o->num("decay_base", 0.33); }
		else if (t == "adam") { oc.learning_rate = (float)o->num("learning_rate", 1e-3); oc.beta1 = (float)o->num("beta1", 0.9); oc.beta2 = (float)o->num("beta2", 0.999); oc.epsilon = (float)o->num("epsilon", 1e-8); oc.l2_reg = (float)o->num("l2_reg", 1e-8); }
		if (!o->has("nested")) break;
		o = &(*o)["nested"];
	}
	NGP_CHECK(ngp_encmlp_set_optimizer(m_encmlp, &oc));
	const std::string lt = lower(m_network_config["loss"].str("otype", "L2"));
	const int loss_type = lt == "mape" ? NGP_LOSS_MAPE : lt == "l1" ? NGP_LOSS_L1 : lt == "relativel2" ? NGP_LOSS_RELATIVE_L2 : NGP_LOSS_L2;
	if (image) {
		ngp_image_options io; memset(&io, 0, sizeof(io));
		if (this->image.random_mode != ERandomMode::Random && this->image.random_mode != ERandomMode::Stratified) throw std::runtime_error{"image.random_mode: Halton / Sobol sampling of the image trainer is not part of this build (Random, Stratified)"};
		io.snap_to_pixel_centers = this->image.training.snap_to_pixel_centers; io.linear_colors = this->image.training.linear_colors; io.stratified = this->image.random_mode == ERandomMode::Stratified; io.loss_type = loss_type; io.loss_scale = 128.f; io.batch_size = training_batch_size; io.seed = seed;
		NGP_CHECK(ngp_image_create(m_encmlp, m_image_pixels.data(), NGP_IMAGE_FLOAT, m_image_w, m_image_h, &io, &m_image));
	} else {
		ngp_sdf_options so; memset(&so, 0, sizeof(so));
		so.loss_type = loss_type; so.loss_scale = 128.f; so.batch_size = training_batch_size; so.seed = seed; so.surface_offset_scale = sdf.training.surface_offset_scale; so.zero_offset = sdf.zero_offset;
		if (sdf.mesh_sdf_mode != EMeshSdfMode::Raystab) throw std::runtime_error{"sdf.mesh_sdf_mode: only Raystab ground truth is part of this build (Watertight needs the averaged-normal query, PathEscape OptiX)"};
		if (sdf.use_triangle_octree) throw std::runtime_error{"sdf.use_triangle_octree: the octree sampler is not part of this build"};
		if (!sdf.training.generate_sdf_data_online) throw std::runtime_error{"sdf.training.generate_sdf_data_online = False: overriding the training data is not part of this build"};
		NGP_CHECK(ngp_sdf_create(m_encmlp, m_mesh.data(), (uint32_t)(m_mesh.size() / 9), m_mesh_aabb, &so, &m_sdf));
	}
}
float Testbed::compute_image_mse(bool quantize_to_byte) {
	if (mode != ETestbedMode::Image) throw std::runtime_error{"compute_image_mse: no image loaded"};
	ensure_encmlp_trainer();
	float mse = 0.f; NGP_CHECK(ngp_image_mse(m_image, quantize_to_byte ? 1 : 0, &mse));
	return mse;
}
double Testbed::calculate_iou(uint32_t n_samples, float, bool, bool) {
	if (mode != ETestbedMode::Sdf) throw std::runtime_error{"calculate_iou: no mesh loaded"};
	ensure_encmlp_trainer();
	double iou = 0; NGP_CHECK(ngp_sdf_iou(m_sdf, n_samples, &iou));
	return iou;
}

void Testbed::train(uint32_t batch_size) {
	if (mode == ETestbedMode::Image || mode == ETestbedMode::Sdf) { // train_image (testbed_image.cu:231) / training_prep_sdf + train_sdf (testbed_sdf.cu:1580-1635) + optimizer_step
		if (batch_size != training_batch_size && !m_image && !m_sdf) training_batch_size = batch_size;
		ensure_encmlp_trainer();
		if (m_image) NGP_CHECK(ngp_image_train(m_image, nullptr, 1)); else NGP_CHECK(ngp_sdf_train(m_sdf, nullptr, 1));
		++training_step;
		if (training_step % 16 == 0 || training_step == 1) { if (m_image) NGP_CHECK(ngp_image_loss(m_image, nullptr, &loss)); else NGP_CHECK(ngp_sdf_loss(m_sdf, nullptr, &loss)); }
		return;
	}
	{
		const NerfDataset& d = nerf.training.dataset;
		auto has_pixels = [&](size_t i) { return !d.pixels[i].empty() || (i < d.pixels_float.size() && !d.pixels_float[i].empty()) || (i < d.pixels_half.size() && !d.pixels_half[i].empty()); };
		if (d.n_images > 0 && nerf.training.n_images_for_training > 0 && !has_pixels(0))
			throw std::runtime_error{"Cannot train: the first training image holds no pixels (a dataset restored from a snapshot's metadata only, or created empty and not yet filled by training.set_image)."};
		if (nerf.training.n_images_for_training <= 0) return; // train_nerf returns at once (testbed_nerf.cu:2705-2707): a streaming client has not delivered an image yet
	}
	if (batch_size != training_batch_size && !m_nerf) training_batch_size = batch_size;
	ensure_trainer();
	push_options();
	NGP_CHECK(ngp_nerf_train(m_nerf, nullptr, 1));
	++training_step;
	if (training_step % 16 == 0 || training_step == 1) { // get_loss_scalar cadence, testbed.cu:4625
		ngp_nerf_stats s = stats();
		loss = s.loss;
		if (s.measured_batch_size == 0 && training_step > 1) { fprintf(stderr, "Warning: Nerf training generated 0 samples. Aborting training.\n"); shall_train = false; }
	}
}
std::string Testbed::comm_unique_id() {
	uint8_t id[128];
	NGP_CHECK(ngp_comm_unique_id(id));
	return std::string((const char*)id, 128);
}
void Testbed::comm_init(uint32_t rank, uint32_t world_size, const std::string& id) {
	if (world_size < 1 || rank >= world_size) throw std::runtime_error{"comm_init: bad rank / world_size"};
	if (id.size() != 128) throw std::runtime_error{"comm_init: the unique id must be the 128 bytes of Testbed.comm_unique_id()"};
	if (m_rank != rank || m_world_size != world_size) destroy_trainer(); // the sharding is fixed when the trainer is created
	else if (m_nerf && m_comm_up) { NGP_CHECK(ngp_comm_destroy(m_nerf)); m_comm_up = false; } // same sharding, live trainer: the new id replaces the communicator
	m_rank = rank; m_world_size = world_size; m_comm_id = id; m_comm_up = false;
}
ngp_nerf_stats Testbed::stats() {
	ngp_nerf_stats s; memset(&s, 0, sizeof(s));
	if (m_nerf) NGP_CHECK(ngp_nerf_get_stats(m_nerf, nullptr, &s));
	return s;
}
bool Testbed::frame() { // headless: one call = one optimizer step when training is on
	if (shall_train && (mode == ETestbedMode::Nerf || mode == ETestbedMode::Image || mode == ETestbedMode::Sdf)) train(training_batch_size);
	return true;
}

// ------------------------------------------------------------------------------------------------
// rendering
// ------------------------------------------------------------------------------------------------
static float srgb_to_lin(float s) { return s <= 0.04045f ? s / 12.92f : std::pow((s + 0.055f) / 1.055f, 2.4f); }
static float lin_to_srgb(float l) { return l < 0.0031308f ? 12.92f * l : 1.055f * std::pow(l, 0.41666f) - 0.055f; }

static std::array<float, 3> v3sub(const std::array<float, 3>& a, const std::array<float, 3>& b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
static std::array<float, 3> v3cross(const std::array<float, 3>& a, const std::array<float, 3>& b) { return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}; }
static float v3dot(const std::array<float, 3>& a, const std::array<float, 3>& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static std::array<float, 3> v3normalize(const std::array<float, 3>& a) { const float l = std::sqrt(v3dot(a, a)); return {a[0] / l, a[1] / l, a[2] / l}; }
void Testbed::set_camera_to_training_view(int i) {
	const NerfDataset& d = nerf.training.dataset;
	if (i < 0 || (size_t)i >= d.n_images) throw std::runtime_error{"set_camera_to_training_view: index out of range"};
	const auto old_look_at = look_at();
	m_camera = d.xforms[i];
	m_scale = std::max(v3dot(v3sub(old_look_at, view_pos()), view_dir()), 0.1f); // testbed.cu:494
	const auto& m = d.metadata[i];
	for (int k = 0; k < 2; ++k) relative_focal_length[k] = m.focal_length[k] / (float)m.resolution[fov_axis];
	render_with_lens_distortion = true;
	m_render_lens_mode = m.lens_mode; m_render_lens_params = m.lens_params;
	screen_center = {1.0f - m.principal_point[0], 1.0f - m.principal_point[1]};
	m_training_view = i; nerf.training.view = i;
}
void Testbed::set_nerf_camera_matrix(const std::array<float, 12>& rm) { // 3x4 row-major NeRF-convention matrix (python_api.cu / testbed.cu:463-466)
	const NerfDataset& d = nerf.training.dataset;
	float m[3][4];
	for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) m[r][c] = rm[r * 4 + c];
	for (int r = 0; r < 3; ++r) { m[r][1] *= -1.f; m[r][2] *= -1.f; m[r][3] = m[r][3] * d.scale + d.offset[r]; }
	float c3[3][4];
	for (int c = 0; c < 4; ++c) { c3[0][c] = m[1][c]; c3[1][c] = m[2][c]; c3[2][c] = m[0][c]; }
	for (int c = 0; c < 4; ++c) for (int r = 0; r < 3; ++r) m_camera[c * 3 + r] = c3[r][c];
}
float Testbed::fov() const { return 2.0f * std::atan(0.5f / relative_focal_length[fov_axis]) * 180.0f / 3.14159265358979f; } // testbed.cu focal_length_to_fov
void Testbed::set_fov(float deg) {
	const float rf = 0.5f / std::tan(0.5f * deg * 3.14159265358979f / 180.0f);
	relative_focal_length = {rf, rf};
}

// ---- camera and training-view helpers with the reference's semantics (testbed.cu:440-528, 4085-4091; testbed_nerf.cu:2151-2292, 3710-3723) ----
std::array<float, 3> Testbed::look_at() const { const auto p = view_pos(), d = view_dir(); return {p[0] + d[0] * m_scale, p[1] + d[1] * m_scale, p[2] + d[2] * m_scale}; }
void Testbed::set_look_at(const std::array<float, 3>& pos) { const auto l = look_at(); for (int k = 0; k < 3; ++k) m_camera[9 + k] += pos[k] - l[k]; }
void Testbed::set_scale(float scale) {
	const auto prev = look_at(), p = view_pos();
	for (int k = 0; k < 3; ++k) m_camera[9 + k] = (p[k] - prev[k]) * (scale / m_scale) + prev[k];
	m_scale = scale;
}
void Testbed::set_view_dir(const std::array<float, 3>& dir) {
	const auto old = look_at();
	const auto c0 = v3normalize(v3cross(dir, up_dir)), c1 = v3normalize(v3cross(dir, c0)), c2 = v3normalize(dir);
	for (int k = 0; k < 3; ++k) { m_camera[k] = c0[k]; m_camera[3 + k] = c1[k]; m_camera[6 + k] = c2[k]; }
	set_look_at(old);
}
std::array<float, 2> Testbed::fov_xy() const { // focal_length_to_fov(ivec2(1), m_relative_focal_length), common_device.cuh:657-659
	return {2.0f * 180.0f / 3.14159265358979323846f * std::atan(1.0f / (relative_focal_length[0] * 2.0f)), 2.0f * 180.0f / 3.14159265358979323846f * std::atan(1.0f / (relative_focal_length[1] * 2.0f))};
}
void Testbed::set_fov_xy(const std::array<float, 2>& deg) { // fov_to_focal_length(ivec2(1), val), common_device.cuh:649-651
	for (int k = 0; k < 2; ++k) relative_focal_length[k] = 0.5f * 1.0f / std::tan(0.5f * deg[k] * (3.14159265358979323846f / 180.0f));
}
std::array<float, 12> Testbed::camera_matrix_row_major() const { std::array<float, 12> r; for (int row = 0; row < 3; ++row) for (int c = 0; c < 4; ++c) r[row * 4 + c] = m_camera[c * 3 + row]; return r; }
void Testbed::set_camera_matrix_row_major(const std::array<float, 12>& m) { for (int row = 0; row < 3; ++row) for (int c = 0; c < 4; ++c) m_camera[c * 3 + row] = m[row * 4 + c]; }
void Testbed::first_training_view() { nerf.training.view = 0; set_camera_to_training_view(nerf.training.view); }
void Testbed::last_training_view() { nerf.training.view = (int)nerf.training.dataset.n_images - 1; set_camera_to_training_view(nerf.training.view); }
void Testbed::previous_training_view() { if (nerf.training.view != 0) nerf.training.view -= 1; set_camera_to_training_view(nerf.training.view); }
void Testbed::next_training_view() { if (nerf.training.view != (int)nerf.training.dataset.n_images - 1) nerf.training.view += 1; set_camera_to_training_view(nerf.training.view); }
void Testbed::reset_camera() {
	fov_axis = 1; zoom = 1.0f; screen_center = {0.5f, 0.5f};
	if (mode == ETestbedMode::Image) { relative_focal_length = {1.0f, 1.0f}; m_scale = 1.0f; }
	else { set_fov(50.625f); m_scale = 1.5f; }
	m_camera = {1, 0, 0, 0, -1, 0, 0, 0, -1, 0.5f, 0.5f, 0.5f}; // m_default_camera, testbed.h:654
	const auto d = view_dir();
	for (int k = 0; k < 3; ++k) m_camera[9 + k] -= m_scale * d[k];
}
void Testbed::create_empty_nerf_dataset(size_t n_images, int aabb_scale, bool is_hdr) {
	NerfDataset d;
	d.n_images = n_images; d.aabb_scale = aabb_scale; d.is_hdr = is_hdr;
	d.metadata.assign(n_images, ImageMetadata{});
	d.xforms.assign(n_images, std::array<float, 12>{1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}); d.xforms_end = d.xforms; // mat4x3::identity()
	d.pixels.assign(n_images, {}); d.pixels_half.assign(n_images, {}); d.pixels_float.assign(n_images, {}); d.depth.assign(n_images, {});
	d.paths.assign(n_images, std::string());
	if ((aabb_scale & (aabb_scale - 1)) != 0 || aabb_scale < 1 || aabb_scale > 128) throw std::runtime_error{"create_empty_nerf_dataset: aabb_scale must be a power of two <= 128"};
	const int prev_scale = nerf.training.dataset.aabb_scale; const bool had = nerf.training.dataset.n_images > 0;
	nerf.training.dataset = std::move(d);
	mode = ETestbedMode::Nerf;
	load_nerf_post();
	nerf.training.n_images_for_training = 0; // testbed_nerf.cu:2349: the caller raises it as images arrive
	if (had && prev_scale != aabb_scale) destroy_trainer();
	m_dataset_dirty = true;
}
void Testbed::set_training_image(int frame_idx, int w, int h, const float* rgba, const float* depth, float depth_scale) {
	NerfDataset& d = nerf.training.dataset;
	if (frame_idx < 0 || (size_t)frame_idx >= d.n_images) throw std::runtime_error{"Invalid frame index"};
	if (w < 1 || h < 1) throw std::runtime_error{"image should be (H,W,C) where C=4"};
	if (d.pixels_float.size() < d.n_images) d.pixels_float.resize(d.n_images);
	if (d.depth.size() < d.n_images) d.depth.resize(d.n_images);
	d.pixels_float[frame_idx].assign(rgba, rgba + (size_t)w * h * 4);
	d.pixels[frame_idx].clear(); if ((size_t)frame_idx < d.pixels_half.size()) d.pixels_half[frame_idx].clear();
	d.metadata[frame_idx].resolution = {w, h};
	// copy_depth<float> (nerf_loader.cu:73-82): no depth data, or a scale <= 0, means "no measurement anywhere"
	d.depth[frame_idx].clear();
	if (depth && depth_scale > 0.f) { d.depth[frame_idx].resize((size_t)w * h); for (size_t i = 0; i < (size_t)w * h; ++i) d.depth[frame_idx][i] = depth[i] * depth_scale; }
	m_dataset_dirty = true;
}
void Testbed::load_camera_path(const std::string& path) {
	if (!fs::exists(path)) throw std::runtime_error{"Camera path " + path + " does not exist."};
	camera_path.load_json_text(read_text(path), path);
}
void Testbed::set_camera_from_time(float t) {
	if (camera_path.keyframes.empty()) return;
	const CameraKeyframe k = camera_path.eval_camera_path(t);
	m_camera = k.m(); m_scale = k.scale; set_fov(k.fov); // set_camera_from_keyframe, testbed.cu:4061-4067 (slice plane and aperture belong to the GUI / depth of field)
}
std::vector<float> Testbed::render_path_frame(int width, int height, int spp, bool linear, float start_t, float end_t, float shutter_fraction, std::vector<float>* depth_out) {
	// The reference moves the camera inside every sample as well (per-pixel time between the sample's start and end matrices); here a sample is rendered from the
	// camera at the MIDDLE of its sub-interval -- the same time samples (python_api.cu:183-196), motion blur resolved per sample instead of per pixel.
	if (end_t < 0.f) end_t = start_t;
	const int n = std::max(spp, 1);
	std::vector<float> acc((size_t)width * height * 4, 0.f), depth;
	for (int i = 0; i < n; ++i) {
		const float start_alpha = (float)i / (float)n * shutter_fraction, end_alpha = ((float)i + 1.0f) / (float)n * shutter_fraction;
		set_camera_from_time(start_t + (end_t - start_t) * (start_alpha + end_alpha) / 2.0f);
		const std::vector<float> one = render(width, height, 1, true, depth_out ? &depth : nullptr);
		for (size_t k = 0; k < acc.size(); ++k) acc[k] += (one[k] - acc[k]) / (float)(i + 1);
	}
	if (depth_out) *depth_out = depth;
	if (!linear) for (size_t i = 0; i < (size_t)width * height; ++i) for (int k = 0; k < 3; ++k) acc[i * 4 + k] = lin_to_srgb(acc[i * 4 + k]);
	return acc;
}
void Testbed::clear_training_data() { // testbed.cu:190-193: the metadata goes, training stops being possible until new data is loaded
	nerf.training.dataset = NerfDataset{};
	nerf.training.n_images_for_training = 0;
	destroy_trainer();
}
size_t Testbed::n_params() {
	if (!m_model) return 0;
	uint64_t n = 0, n_mlp = 0; NGP_CHECK(ngp_model_n_params(m_model, &n, &n_mlp)); return (size_t)n;
}
size_t Testbed::n_encoding_params() { // n_params() - first_encoder_param(): the hash-grid entries
	if (!m_model) return 0;
	uint64_t n = 0, n_mlp = 0; NGP_CHECK(ngp_model_n_params(m_model, &n, &n_mlp)); return (size_t)(n - n_mlp);
}
int Testbed::find_closest_training_view(const std::array<float, 12>& pose) const { // pose: 3 x 4 row-major, ngp convention like camera_matrix
	const NerfDataset& d = nerf.training.dataset;
	int best = nerf.training.view; float best_score = std::numeric_limits<float>::infinity();
	const int n = std::min<int>(nerf.training.n_images_for_training, (int)d.n_images);
	for (int i = 0; i < n; ++i) {
		float dp = 0, dd = 0;
		for (int k = 0; k < 3; ++k) { const float a = d.xforms[i][9 + k] - pose[k * 4 + 3], b = d.xforms[i][6 + k] - pose[k * 4 + 2]; dp += a * a; dd += b * b; }
		const float score = std::sqrt(dp) + 0.25f * std::sqrt(dd);
		if (score < best_score) { best_score = score; best = i; }
	}
	return best;
}
std::array<float, 12> NerfDataset::nerf_matrix_to_ngp(const std::array<float, 12>& rm) const {
	float m[3][4];
	for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) m[r][c] = rm[r * 4 + c];
	for (int r = 0; r < 3; ++r) { m[r][1] *= -1.f; m[r][2] *= -1.f; m[r][3] = m[r][3] * scale + offset[r]; }
	float c3[3][4];
	if (from_mitsuba) { for (int r = 0; r < 3; ++r) { m[r][0] *= -1.f; m[r][2] *= -1.f; } for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) c3[r][c] = m[r][c]; }
	else for (int c = 0; c < 4; ++c) { c3[0][c] = m[1][c]; c3[1][c] = m[2][c]; c3[2][c] = m[0][c]; }
	std::array<float, 12> out;
	for (int c = 0; c < 4; ++c) for (int r = 0; r < 3; ++r) out[c * 3 + r] = c3[r][c];
	return out;
}
std::array<float, 12> NerfDataset::ngp_matrix_to_nerf(const std::array<float, 12>& cm) const {
	float m[3][4];
	for (int c = 0; c < 4; ++c) for (int r = 0; r < 3; ++r) m[r][c] = cm[c * 3 + r];
	if (from_mitsuba) { for (int r = 0; r < 3; ++r) { m[r][0] *= -1.f; m[r][2] *= -1.f; } }
	else { float t[3][4]; for (int c = 0; c < 4; ++c) { t[0][c] = m[2][c]; t[2][c] = m[1][c]; t[1][c] = m[0][c]; } std::memcpy(m, t, sizeof(m)); } // cycle the axes xyz -> yzx
	for (int r = 0; r < 3; ++r) { m[r][1] *= -1.f; m[r][2] *= -1.f; m[r][3] = (m[r][3] - offset[r]) / scale; }
	std::array<float, 12> out;
	for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) out[r * 4 + c] = m[r][c];
	return out;
}
void Testbed::set_camera_intrinsics(int frame_idx, float fx, float fy, float cx, float cy, float k1, float k2, float p1, float p2, float k3, float k4, bool is_fisheye) {
	NerfDataset& d = nerf.training.dataset;
	if (frame_idx < 0 || (size_t)frame_idx >= d.n_images) return;
	if (fx <= 0.f) fx = fy;
	if (fy <= 0.f) fy = fx;
	ImageMetadata& m = d.metadata[frame_idx];
	cx = cx < 0.f ? -cx : cx / (float)m.resolution[0];
	cy = cy < 0.f ? -cy : cy / (float)m.resolution[1];
	m.lens_mode = NGP_LENS_PERSPECTIVE; m.lens_params = {};
	if (k1 || k2 || k3 || k4 || p1 || p2) {
		if (is_fisheye) { m.lens_mode = NGP_LENS_OPENCV_FISHEYE; m.lens_params = {k1, k2, k3, k4, 0, 0, 0}; }
		else { m.lens_mode = NGP_LENS_OPENCV; m.lens_params = {k1, k2, p1, p2, 0, 0, 0}; }
	}
	m.principal_point = {cx, cy}; m.focal_length = {fx, fy};
	m_dataset_dirty = true;
}
void Testbed::set_camera_extrinsics(int frame_idx, const std::array<float, 12>& c2w, bool convert_to_ngp) {
	NerfDataset& d = nerf.training.dataset;
	if (frame_idx < 0 || (size_t)frame_idx >= d.n_images) return;
	std::array<float, 12> m;
	if (convert_to_ngp) m = d.nerf_matrix_to_ngp(c2w);
	else for (int c = 0; c < 4; ++c) for (int r = 0; r < 3; ++r) m[c * 3 + r] = c2w[r * 4 + c];
	d.xforms[frame_idx] = m;
	if ((size_t)frame_idx < d.xforms_end.size()) d.xforms_end[frame_idx] = m;
	d.metadata[frame_idx].rolling_shutter = {0.f, 0.f, 0.f, 0.f};
	m_dataset_dirty = true;
}
std::array<float, 12> Testbed::get_camera_extrinsics(int frame_idx) const {
	const NerfDataset& d = nerf.training.dataset;
	if (frame_idx < 0 || (size_t)frame_idx >= d.n_images) return {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}; // mat4x3::identity()
	return d.ngp_matrix_to_nerf(d.xforms[frame_idx]);
}
void Testbed::training_options_changed() { m_dataset_dirty = true; }
std::array<float, 2> BoundingBox::ray_intersect(const std::array<float, 3>& pos, const std::array<float, 3>& dir) const {
	const float FMAX = std::numeric_limits<float>::max();
	float tmin = (min[0] - pos[0]) / dir[0], tmax = (max[0] - pos[0]) / dir[0];
	if (tmin > tmax) std::swap(tmin, tmax);
	float tymin = (min[1] - pos[1]) / dir[1], tymax = (max[1] - pos[1]) / dir[1];
	if (tymin > tymax) std::swap(tymin, tymax);
	if (tmin > tymax || tymin > tmax) return {FMAX, FMAX};
	if (tymin > tmin) tmin = tymin;
	if (tymax < tmax) tmax = tymax;
	float tzmin = (min[2] - pos[2]) / dir[2], tzmax = (max[2] - pos[2]) / dir[2];
	if (tzmin > tzmax) std::swap(tzmin, tzmax);
	if (tmin > tzmax || tzmin > tmax) return {FMAX, FMAX};
	if (tzmin > tmin) tmin = tzmin;
	if (tzmax < tmax) tmax = tzmax;
	return {tmin, tmax};
}

std::vector<float> Testbed::render(int width, int height, int spp, bool linear, std::vector<float>* depth_out) {
	std::vector<float> out((size_t)width * height * 4, 0.f);
	const float bg[4] = {srgb_to_lin(background_color[0]), srgb_to_lin(background_color[1]), srgb_to_lin(background_color[2]), background_color[3]};
	if (render_ground_truth) {
		// overlay_image_kernel (render_buffer.cu:344-412): point-sample the training image at pixel centres
		const NerfDataset& d = nerf.training.dataset;
		const auto& m = d.metadata[m_training_view];
		if (d.pixels[m_training_view].empty()) throw std::runtime_error{"render_ground_truth: this training image has no 8-bit pixels on the host (set through training.set_image, or restored from a snapshot)"};
		const uint8_t* px = d.pixels[m_training_view].data();
		for (int y = 0; y < height; ++y) for (int x = 0; x < width; ++x) {
			const float u = ((float)x + 0.5f) / (float)width, v = ((float)y + 0.5f) / (float)height;
			const int ix = std::min(std::max((int)(u * m.resolution[0]), 0), m.resolution[0] - 1), iy = std::min(std::max((int)(v * m.resolution[1]), 0), m.resolution[1] - 1);
			const uint8_t* p = px + ((size_t)iy * m.resolution[0] + ix) * 4;
			const float a = p[3] / 255.f;
			// the reference draws the overlay into the frame buffer BEFORE tonemap_kernel (testbed.cu:5076-5091, render_buffer.cu:511-560): the texel, as
			// premultiplied linear RGBA, goes through the same background / exposure / curve arithmetic as a rendered pixel
			const float texel[4] = {srgb_to_lin(p[0] / 255.f) * a, srgb_to_lin(p[1] / 255.f) * a, srgb_to_lin(p[2] / 255.f) * a, a};
			NGP_CHECK(ngp_host_tonemap_pixel(texel, exposure, bg, 0, (int)tonemap_curve, &out[((size_t)y * width + x) * 4]));
		}
	} else {
		ensure_trainer();
		push_options();
		const size_t n = (size_t)width * height * 4;
		if (2 * n + n / 4 > m_frame_dev_floats) { // [frame of one spp | accumulated frame | depth of the last spp]
			if (m_frame_dev) HIP_CHECK(hipFree(m_frame_dev));
			HIP_CHECK(hipMalloc((void**)&m_frame_dev, (2 * n + n / 4) * sizeof(float)));
			m_frame_dev_floats = 2 * n + n / 4;
		}
		float* accum = m_frame_dev + n;
		float* depth_dev = depth_out ? m_frame_dev + 2 * n : nullptr;
		HIP_CHECK(hipMemsetAsync(accum, 0, n * sizeof(float), nullptr));
		ngp_render_params rp; memset(&rp, 0, sizeof(rp));
		rp.resolution[0] = width; rp.resolution[1] = height;
		const int res_axis = fov_axis == 0 ? width : height;
		rp.focal_length[0] = relative_focal_length[0] * (float)res_axis; rp.focal_length[1] = relative_focal_length[1] * (float)res_axis; // calc_focal_length, testbed.cu:4649
		rp.screen_center[0] = (0.5f - screen_center[0]) + 0.5f; rp.screen_center[1] = (0.5f - screen_center[1]) + 0.5f;                      // render_screen_center, testbed.cu:4653
		for (int k = 0; k < 12; ++k) rp.camera[k] = m_camera[k];
		rp.lens_mode = render_with_lens_distortion ? m_render_lens_mode : NGP_LENS_PERSPECTIVE;
		for (int k = 0; k < 7; ++k) rp.lens_params[k] = m_render_lens_params[k];
		rp.snap_to_pixel_centers = snap_to_pixel_centers;
		rp.min_transmittance = nerf.render_min_transmittance;
		rp.near_distance = render_near_distance;
		rp.use_inference_params = 1; // the renderer uses the optimizer's EMA weights (testbed_nerf.cu:1772)
		{ // m_render_aabb (testbed_nerf.cu:2427-2431; writable from Python); its orientation must be the identity in this build
			for (int k = 0; k < 9; ++k) if (render_aabb_to_local[k] != (k % 4 == 0 ? 1.f : 0.f)) throw std::runtime_error{"render: a rotated crop box (render_aabb_to_local != identity) is not supported by this build"};
			const BoundingBox box = render_aabb.is_empty() ? aabb : render_aabb;
			for (int k = 0; k < 3; ++k) { rp.render_aabb.min[k] = box.min[k]; rp.render_aabb.max[k] = box.max[k]; }
		}
		for (int s = 0; s < std::max(spp, 1); ++s) { // accumulate + tonemap stay on the device (render_buffer.cu:228-260, 511-560): one read-back per frame
			rp.spp_index = (uint32_t)s;
			NGP_CHECK(ngp_nerf_render(m_nerf, nullptr, &rp, m_frame_dev, depth_dev));
			NGP_CHECK(ngp_render_accumulate(nullptr, m_frame_dev, accum, n, (uint32_t)s));
		}
		NGP_CHECK(ngp_render_tonemap_curve(nullptr, accum, (uint64_t)width * height, exposure, bg, 0, (int)tonemap_curve));
		HIP_CHECK(hipMemcpy(out.data(), accum, n * sizeof(float), hipMemcpyDeviceToHost));
		if (depth_out) { depth_out->resize(n / 4); HIP_CHECK(hipMemcpy(depth_out->data(), depth_dev, (n / 4) * sizeof(float), hipMemcpyDeviceToHost)); } // render_to_cpu's second array (python_api.cu:231-236): the depth buffer as the last sample left it
	}
	if (!linear) for (size_t i = 0; i < (size_t)width * height; ++i) for (int k = 0; k < 3; ++k) out[i * 4 + k] = lin_to_srgb(out[i * 4 + k]);
	return out;
}

// ------------------------------------------------------------------------------------------------
// snapshots in the reference's wire format (testbed.cu:5288-5514): nlohmann::json::to_msgpack of the network config with a
// "snapshot" object, `.ingp` = the same bytes through zlib (zstr).  ngp-side fields and the vector / bounding-box / dataset
// encodings follow testbed.cu:5288-5343 and json_binding.h; the tcnn-side fields (Trainer::serialize: "n_params", "params_type",
// "params_binary" = inference parameters in half, optional "optimizer") are restated from memory of the public tiny-cuda-nn
// [tcnn, unverifiable here: the submodule is absent and the mount holds no snapshot file].  The optimizer state is written under
// "optimizer" in tcnn's nesting and key names as far as they can be restated here (round 5; rounds 1-4 wrote one private blob,
// which load_snapshot still reads), the fp32 master parameters under the private key "ngp_hip_master_binary".
// ------------------------------------------------------------------------------------------------
namespace {
using mini_json::Value;
Value jnum(double d) { Value v; v.type = Value::Number; v.n = d; return v; }
Value jbool(bool b) { Value v; v.type = Value::Bool; v.b = b; return v; }
Value jstr(const std::string& s) { Value v; v.type = Value::String; v.s = s; return v; }
Value jobj() { Value v; v.type = Value::Object; return v; }
template <typename T> Value jvec(const T* p, size_t n) { Value v; v.type = Value::Array; for (size_t i = 0; i < n; ++i) v.arr.push_back(jnum((double)p[i])); return v; }
Value jmat_cols(const float* p, size_t n_cols, size_t n_rows) { Value v; v.type = Value::Array; for (size_t c = 0; c < n_cols; ++c) v.arr.push_back(jvec(p + c * n_rows, n_rows)); return v; } // [tcnn vec_json.h] array of columns
Value jbox(const ngp_aabb& b) { Value v = jobj(); v.set("min", jvec(b.min, 3)); v.set("max", jvec(b.max, 3)); return v; }
uint16_t f32_to_f16(float f) { // round to nearest even
	uint32_t x; memcpy(&x, &f, 4);
	const uint32_t sign = (x >> 16) & 0x8000u; x &= 0x7fffffffu;
	if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0u));
	if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                        // overflow -> inf
	if (x < 0x38800000u) {                                                            // subnormal half (or zero)
		if (x < 0x33000000u) return (uint16_t)sign;
		const uint32_t e = x >> 23, m = (x & 0x7fffffu) | 0x800000u, shift = 126u - e; // value = m * 2^(e-150); half subnormal unit 2^-24
		uint32_t h = m >> shift; const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1u);
		if (rem > half || (rem == half && (h & 1u))) ++h;
		return (uint16_t)(sign | h);
	}
	uint32_t h = ((x - 0x38000000u) >> 13); const uint32_t rem = x & 0x1fffu;
	if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
	return (uint16_t)(sign | h);
}
float f16_to_f32(uint16_t h) {
	const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 1023u;
	uint32_t x;
	if (e == 0) { if (m == 0) x = sign; else { int sh = 0; uint32_t mm = m; while (!(mm & 0x400u)) { mm <<= 1; ++sh; } x = sign | ((uint32_t)(113 - sh) << 23) | ((mm & 0x3ffu) << 13); } }
	else if (e == 31) x = sign | 0x7f800000u | (m << 13);
	else x = sign | ((e + 112u) << 23) | (m << 13);
	float f; memcpy(&f, &x, 4); return f;
}
bool ends_with_ci(const std::string& s, const char* ext) { const size_t n = strlen(ext); if (s.size() < n) return false; for (size_t i = 0; i < n; ++i) if (tolower((unsigned char)s[s.size() - n + i]) != ext[i]) return false; return true; }
} // namespace

// from_json(NerfDataset) / from_json(Lens) / from_json(TrainingXForm) / from_json(BoundingBox), json_binding.h:30-34, 67-105, 141-190: the dataset METADATA a snapshot
// embeds (no pixels), with the reference reader's dataset-wide defaults and legacy keys ("lens" / "camera_distortion", "principal_point", "rolling_shutter", "focal_length",
// "image_resolution" at the top level; "focal_lengths"; a per-image "metadata" entry overrides them) so that documents of older instant-ngp versions read the same way.
static void lens_from_json(const mini_json::Value& lens, ImageMetadata& m) { // from_json(Lens), json_binding.h:67-100 (anything unrecognised = Perspective)
	m.lens_mode = NGP_LENS_PERSPECTIVE;
	if (!lens.is_object()) return;
	if (lens.has("k1")) {
		if (lens.boolean("is_fisheye", false)) { m.lens_mode = NGP_LENS_OPENCV_FISHEYE; m.lens_params[0] = (float)lens.num("k1", 0); m.lens_params[1] = (float)lens.num("k2", 0); m.lens_params[2] = (float)lens.num("k3", 0); m.lens_params[3] = (float)lens.num("k4", 0); }
		else { m.lens_mode = NGP_LENS_OPENCV; m.lens_params[0] = (float)lens.num("k1", 0); m.lens_params[1] = (float)lens.num("k2", 0); m.lens_params[2] = (float)lens.num("p1", 0); m.lens_params[3] = (float)lens.num("p2", 0); }
	} else if (lens.has("ftheta_p0")) { m.lens_mode = NGP_LENS_FTHETA; for (int k = 0; k < 5; ++k) m.lens_params[k] = (float)lens.num(std::string("ftheta_p") + char('0' + k), 0); m.lens_params[5] = (float)lens.num("w", 0); m.lens_params[6] = (float)lens.num("h", 0); }
	else if (lens.has("latlong")) m.lens_mode = NGP_LENS_LATLONG;
	else if (lens.has("equirectangular")) m.lens_mode = NGP_LENS_EQUIRECTANGULAR;
	else if (lens.has("orthographic")) m.lens_mode = NGP_LENS_ORTHOGRAPHIC;
}
NerfDataset Testbed::dataset_from_json(const mini_json::Value& jd, int default_aabb_scale) {
	if (!jd.is_object() || !jd["xforms"].is_array()) throw std::runtime_error{"Snapshot holds no dataset metadata and no training data is loaded."};
	NerfDataset d;
	const size_t n = jd.has("n_images") ? (size_t)jd.num("n_images", 0) : jd["xforms"].size();
	if (jd["xforms"].size() < n) throw std::runtime_error{"Snapshot dataset: fewer xforms than images."};
	d.aabb_scale = (int)jd.num("aabb_scale", default_aabb_scale); d.scale = (float)jd.num("scale", 0.33); d.is_hdr = jd["is_hdr"].type == Value::Bool && jd["is_hdr"].b; d.from_mitsuba = jd.boolean("from_mitsuba", false);
	if (jd["offset"].size() == 3) for (int k = 0; k < 3; ++k) d.offset[k] = (float)jd["offset"].at(k).n;
	if (jd["up"].size() == 3) for (int k = 0; k < 3; ++k) d.up[k] = (float)jd["up"].at(k).n;
	if (jd["render_aabb"].is_object() && jd["render_aabb"]["min"].size() == 3 && jd["render_aabb"]["max"].size() == 3)
		for (int k = 0; k < 3; ++k) { d.render_aabb.min[k] = (float)jd["render_aabb"]["min"].at(k).n; d.render_aabb.max[k] = (float)jd["render_aabb"]["max"].at(k).n; }
	if (jd["render_aabb_to_local"].size() == 3) for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) d.render_aabb_to_local[c * 3 + r] = (float)jd["render_aabb_to_local"].at(c).at(r).n;
	auto vec_into = [](const Value& v, auto& dst, size_t cnt) { if (v.is_array() && v.size() == cnt) for (size_t k = 0; k < cnt; ++k) dst[k] = (typename std::remove_reference<decltype(dst[0])>::type)v.at(k).n; };
	for (size_t i = 0; i < n; ++i) {
		ImageMetadata m;
		// dataset-wide defaults first, legacy names included (json_binding.h:149-156)
		if (jd.has("lens")) lens_from_json(jd["lens"], m);
		if (jd.has("camera_distortion")) lens_from_json(jd["camera_distortion"], m);
		vec_into(jd["principal_point"], m.principal_point, 2); vec_into(jd["rolling_shutter"], m.rolling_shutter, 4); vec_into(jd["focal_length"], m.focal_length, 2); vec_into(jd["image_resolution"], m.resolution, 2);
		if (jd["focal_lengths"].is_array() && i < jd["focal_lengths"].size()) vec_into(jd["focal_lengths"].at(i), m.focal_length, 2);
		if (jd["metadata"].is_array() && i < jd["metadata"].size()) {
			const Value& jm = jd["metadata"].at(i);
			vec_into(jm["resolution"], m.resolution, 2); vec_into(jm["focal_length"], m.focal_length, 2); vec_into(jm["principal_point"], m.principal_point, 2);
			if (jm.has("lens")) lens_from_json(jm["lens"], m);
			if (jm.has("camera_distortion")) lens_from_json(jm["camera_distortion"], m);
			vec_into(jm["rolling_shutter"], m.rolling_shutter, 4); // (written per image by to_json; the reference's reader takes the dataset-wide value only)
		}
		std::array<float, 12> x{}, xe{};
		const Value& xs = jd["xforms"].at(i)["start"];
		const Value& xend = jd["xforms"].at(i).has("end") ? jd["xforms"].at(i)["end"] : xs;
		for (int c = 0; c < 4; ++c) for (int r = 0; r < 3; ++r) { x[c * 3 + r] = (float)xs.at(c).at(r).n; xe[c * 3 + r] = (float)xend.at(c).at(r).n; }
		d.metadata.push_back(m); d.xforms.push_back(x); d.xforms_end.push_back(xe); d.pixels.emplace_back(); d.pixels_half.emplace_back(); // no pixels
		d.paths.push_back(jd["paths"].is_array() && i < jd["paths"].size() ? jd["paths"].at(i).s : std::string());
	}
	d.n_images = n;
	d.n_extra_learnable_dims = (uint32_t)jd.num("n_extra_learnable_dims", 0); // from_json(NerfDataset), json_binding.h:190
	return d;
}
// test hooks (pyngp: Testbed._nerf_dataset_to_json / _nerf_dataset_from_json): the two functions above through JSON text, no device needed
std::string Testbed::nerf_dataset_to_json_text() const { return mini_json::dump(dataset_to_json()); }
void Testbed::nerf_dataset_from_json_text(const std::string& text) {
	mini_json::Value v; std::string err;
	if (!mini_json::parse(text.c_str(), v, err)) throw std::runtime_error{"nerf_dataset_from_json: " + err};
	destroy_trainer();
	nerf.training.dataset = dataset_from_json(v, nerf.training.dataset.aabb_scale);
	mode = ETestbedMode::Nerf;
	load_nerf_post();
	m_dataset_dirty = true; shall_train = false;
}

// to_json(NerfDataset) / to_json(Lens) / to_json(TrainingXForm) / to_json(BoundingBox), json_binding.h:24-139 (vectors as arrays, matrices as arrays of columns: [tcnn vec_json.h]).
// Pinned against the reference's own functions compiled from where they lie: tests/test_ref_snapshot.py (oracle/ref_json_wrapper.cpp).
mini_json::Value Testbed::dataset_to_json() const {
	const NerfDataset& d = nerf.training.dataset;
	Value jd = jobj(); jd.set("n_images", jnum((double)d.n_images));
	Value paths; paths.type = Value::Array; for (const auto& p : d.paths) paths.arr.push_back(jstr(p)); jd.set("paths", paths);
	Value metas; metas.type = Value::Array; Value xfs; xfs.type = Value::Array;
	for (size_t i = 0; i < d.n_images; ++i) {
		const ImageMetadata& m = d.metadata[i];
		Value jm = jobj(); jm.set("focal_length", jvec(m.focal_length.data(), 2));
		Value lens; // to_json(Lens), json_binding.h:37-65: a Perspective lens sets no key, so nlohmann leaves the value null (NOT an empty object); every other model creates the object
		if (m.lens_mode != NGP_LENS_PERSPECTIVE) lens = jobj();
		if (m.lens_mode == NGP_LENS_OPENCV) { lens.set("is_fisheye", jbool(false)); lens.set("k1", jnum(m.lens_params[0])); lens.set("k2", jnum(m.lens_params[1])); lens.set("p1", jnum(m.lens_params[2])); lens.set("p2", jnum(m.lens_params[3])); }
		else if (m.lens_mode == NGP_LENS_OPENCV_FISHEYE) { lens.set("is_fisheye", jbool(true)); lens.set("k1", jnum(m.lens_params[0])); lens.set("k2", jnum(m.lens_params[1])); lens.set("k3", jnum(m.lens_params[2])); lens.set("k4", jnum(m.lens_params[3])); }
		else if (m.lens_mode == NGP_LENS_FTHETA) { for (int k = 0; k < 5; ++k) lens.set(std::string("ftheta_p") + char('0' + k), jnum(m.lens_params[k])); lens.set("w", jnum(m.lens_params[5])); lens.set("h", jnum(m.lens_params[6])); }
		else if (m.lens_mode == NGP_LENS_LATLONG) lens.set("latlong", jbool(true));
		else if (m.lens_mode == NGP_LENS_EQUIRECTANGULAR) lens.set("equirectangular", jbool(true));
		else if (m.lens_mode == NGP_LENS_ORTHOGRAPHIC) lens.set("orthographic", jbool(true));
		jm.set("lens", lens); jm.set("principal_point", jvec(m.principal_point.data(), 2));
		jm.set("rolling_shutter", jvec(m.rolling_shutter.data(), 4)); jm.set("resolution", jvec(m.resolution.data(), 2));
		metas.arr.push_back(jm);
		Value x = jobj(); x.set("start", jmat_cols(d.xforms[i].data(), 4, 3)); x.set("end", jmat_cols((i < d.xforms_end.size() ? d.xforms_end[i] : d.xforms[i]).data(), 4, 3)); xfs.arr.push_back(x);
	}
	jd.set("metadata", metas); jd.set("xforms", xfs);
	{ // to_json(NerfDataset), json_binding.h: the dataset's own crop box (empty = none: the scene box is written, as before), its orientation and up vector
		ngp_aabb box = scene_aabb();
		if (!d.render_aabb.is_empty()) for (int k = 0; k < 3; ++k) { box.min[k] = d.render_aabb.min[k]; box.max[k] = d.render_aabb.max[k]; }
		jd.set("render_aabb", jbox(box));
	}
	jd.set("render_aabb_to_local", jmat_cols(d.render_aabb_to_local.data(), 3, 3));
	jd.set("up", jvec(d.up.data(), 3)); jd.set("offset", jvec(d.offset.data(), 3));
	const int env[2] = {0, 0}; jd.set("envmap_resolution", jvec(env, 2)); jd.set("scale", jnum(d.scale)); jd.set("aabb_scale", jnum(d.aabb_scale));
	jd.set("from_mitsuba", jbool(d.from_mitsuba)); jd.set("is_hdr", jbool(d.is_hdr)); jd.set("wants_importance_sampling", jbool(true)); jd.set("n_extra_learnable_dims", jnum(d.n_extra_learnable_dims));
	return jd;
}

void Testbed::dp_gather_state() { // COLLECTIVE (every rank): the optimizer state of every rank's own pieces to all ranks -- before save_snapshot(path, True) / parameter read-backs of a sharded run
	ensure_trainer();
	NGP_CHECK(ngp_nerf_dp_gather_state(m_nerf, nullptr));
}

void Testbed::save_snapshot(const std::string& path, bool include_optimizer_state) {
	ensure_trainer();
	// data parallel (sharded step): the fp32 master parameters / Adam state of the other ranks' pieces are gathered first -- a COLLECTIVE: with an optimizer state every rank
	// calls save_snapshot (each may write its own file); the half / inference parameters (all a snapshot without optimizer state holds) are current on every rank anyway
	// (round 6: no longer gathered implicitly -- rank 0 saving alone, the usual pattern, would block forever in the all-gather; the caller makes the collective explicit)
	if (include_optimizer_state && ngp_nerf_dp_state_stale(m_nerf))
		throw std::runtime_error{"save_snapshot with optimizer state under the sharded data-parallel step: this rank holds stale fp32 masters / Adam moments for the other ranks' pieces. "
			"Call testbed.dp_gather_state() on EVERY rank first (a collective), then save on the ranks that should write a file."};
	Value root = m_network_config.type == Value::Object ? m_network_config : jobj();
	Value snap = jobj();
	// ---- tcnn Trainer::serialize ----
	uint64_t n_params = 0, n_mlp = 0;
	NGP_CHECK(ngp_model_n_params(m_model, &n_params, &n_mlp));
	float* master = nullptr; ngp_half* params = nullptr; ngp_half* inference = nullptr; ngp_half* grads = nullptr;
	NGP_CHECK(ngp_model_param_ptrs(m_model, &master, &params, &inference, &grads));
	HIP_CHECK(hipDeviceSynchronize());
	Value pb; pb.type = Value::Binary; pb.bin.resize(n_params * 2);
	HIP_CHECK(hipMemcpy(pb.bin.data(), inference, n_params * 2, hipMemcpyDeviceToHost));
	snap.set("n_params", jnum((double)n_params)); snap.set("params_type", jstr("__half")); snap.set("params_binary", pb);
	if (include_optimizer_state) {
		// snapshot["optimizer"] = m_optimizer->serialize() [tcnn, from memory -- trainer.h Trainer::serialize; optimizers/ema.h, exponential_decay.h, adam.h]: the optimizers nest as
		// the config nests them (configs/nerf/base.json: Ema { ExponentialDecay { Adam } }), each level {"nested": <inner>, own fields}; Adam writes "current_step",
		// "base_learning_rate", "first_moments_binary", "second_moments_binary" (GPUMemory<float> as msgpack bin) and "param_steps_binary" (GPUMemory<uint32_t>); Ema writes
		// its step count and "weights_ema_binary".  Unverifiable in this mount (tcnn's sources are absent): the key names are the best restatement available, and a reader
		// that knows more keys ignores none of these.  The fp32 master parameters -- which tcnn does NOT serialise (its deserialize casts "params_binary" up) -- travel
		// under a private key so that a snapshot of this library resumes bit for bit.
		std::vector<uint8_t> blob(ngp_model_serialized_size(m_model, 1));
		NGP_CHECK(ngp_model_serialize_host(m_model, blob.data(), blob.size(), 1));
		struct { uint64_t n_params; uint32_t step, with_optimizer; float lr; } h;
		NGP_CHECK(ngp_model_state_header(blob.data(), blob.size(), 0, &h.n_params, &h.step, &h.lr, &h.with_optimizer));
		const size_t nb = (size_t)n_params * 4;
		auto bin = [&](int section) { Value v; v.type = Value::Binary; const uint8_t* b = blob.data() + ngp_model_state_offset(n_params, section); v.bin.assign(b, b + nb); return v; };
		ngp_model_config mc; NGP_CHECK(ngp_model_get_config(m_model, &mc));
		Value adam = jobj();
		adam.set("current_step", jnum(h.step)); adam.set("base_learning_rate", jnum(h.lr)); // (the rate Adam steps with now: ExponentialDecay has already applied its factors)
		adam.set("first_moments_binary", bin(NGP_STATE_ADAM_M)); adam.set("second_moments_binary", bin(NGP_STATE_ADAM_V)); adam.set("param_steps_binary", bin(NGP_STATE_ADAM_STEPS));
		Value decay = jobj(); decay.set("nested", adam); decay.set("base_learning_rate", jnum(m_network_config["optimizer"]["nested"]["nested"].num("learning_rate", m_network_config["optimizer"].num("learning_rate", 1e-2))));
		// EmaOptimizer::serialize [tcnn, from memory]: "weights_ema_binary" = m_weights_ema, a GPUMemory<T> -- the EMA weights in NETWORK precision ("params_type"), which is what
		// the inference parameters are (`pb` above).  A "full_precision" EMA's fp32 state travels under a private key next to the master parameters (round-5 files carried an
		// fp32 blob under the reference's key: a tcnn reader sizes its buffer from the byte count and would have read garbage).
		Value ema = jobj(); ema.set("nested", decay); ema.set("ema_step", jnum(h.step)); ema.set("full_precision", jbool(mc.ema_full_precision != 0)); ema.set("weights_ema_binary", pb);
		snap.set("optimizer", ema);
		snap.set("ngp_hip_master_binary", bin(NGP_STATE_MASTER));
		if (mc.ema_full_precision) snap.set("ngp_hip_ema_binary", bin(NGP_STATE_EMA));
	}
	// ---- Testbed::save_snapshot, testbed.cu:5291-5343 ----
	snap.set("version", jnum(1)); snap.set("mode", jstr("nerf"));
	snap.set("density_grid_size", jnum(128));
	float* grid_dev = nullptr;
	NGP_CHECK(ngp_nerf_density_grid_ptrs(m_nerf, &grid_dev, nullptr, nullptr));
	const uint64_t gf = (uint64_t)128 * 128 * 128 * (nerf.max_cascade + 1);
	std::vector<float> grid(gf);
	HIP_CHECK(hipMemcpy(grid.data(), grid_dev, gf * 4, hipMemcpyDeviceToHost));
	Value gb; gb.type = Value::Binary; gb.bin.resize(gf * 2);
	for (uint64_t i = 0; i < gf; ++i) { const uint16_t h = f32_to_f16(grid[i]); memcpy(&gb.bin[i * 2], &h, 2); }
	snap.set("density_grid_binary", gb);
	const ngp_nerf_stats st = stats();
	Value jn = jobj();
	jn.set("aabb_scale", jnum(nerf.training.dataset.aabb_scale));
	{ Value e; e.type = Value::Array; jn.set("cam_pos_offset", e); jn.set("cam_rot_offset", e); } // camera optimisation is not part of this build
	{ // snapshot["nerf"]["extra_dims_opt"] = m_nerf.training.extra_dims_opt (testbed.cu:5311): one VarAdamOptimizer per image (adam_optimizer.h:72-81)
		Value arr; arr.type = Value::Array;
		const uint32_t ne = nerf.training.dataset.n_extra_dims();
		const uint32_t n_img = ne ? (uint32_t)nerf.training.dataset.n_images : 0u; // one optimizer per image of the DATASET (reset_extra_dims, testbed_nerf.cu:3661), trained or not
		if (ne && n_img) {
			std::vector<float> var((size_t)n_img * ne), m1(var.size()), m2(var.size()); std::vector<uint32_t> iter(n_img, 0u);
			NGP_CHECK(ngp_nerf_get_extra_dims(m_nerf, var.data(), n_img));
			NGP_CHECK(ngp_nerf_get_extra_dims_optimizer(m_nerf, m1.data(), m2.data(), iter.data(), n_img));
			const float lr = ngp_nerf_extra_dims_learning_rate(m_nerf); // the rate the last step was taken with (set_learning_rate(m_optimizer->learning_rate()) BEFORE the step, testbed_nerf.cu:2874), not the decayed one behind it
			for (uint32_t i = 0; i < n_img; ++i) {
				Value o = jobj();
				o.set("iter", jnum(iter[i])); o.set("first_moment", jvec(m1.data() + (size_t)i * ne, ne)); o.set("second_moment", jvec(m2.data() + (size_t)i * ne, ne));
				o.set("variable", jvec(var.data() + (size_t)i * ne, ne));
				o.set("learning_rate", jnum(iter[i] ? lr : 1e-4f)); o.set("epsilon", jnum(1e-8f)); o.set("beta1", jnum(0.9f)); o.set("beta2", jnum(0.99f));
				arr.arr.push_back(o);
			}
		}
		jn.set("extra_dims_opt", arr);
	}
	Value rgb = jobj(); rgb.set("rays_per_batch", jnum(st.rays_per_batch)); rgb.set("measured_batch_size", jnum(st.measured_batch_size));
	rgb.set("measured_batch_size_before_compaction", jnum(st.measured_batch_size_before_compaction));
	jn.set("rgb", rgb);
	jn.set("dataset", dataset_to_json()); // to_json(NerfDataset), json_binding.h:114-139
	snap.set("nerf", jn);
	snap.set("training_step", jnum(training_step)); snap.set("loss", jnum(loss));
	const ngp_aabb box = scene_aabb();
	snap.set("aabb", jbox(box)); snap.set("bounding_radius", jnum(1.0));
	snap.set("render_aabb_to_local", jmat_cols(render_aabb_to_local.data(), 3, 3));
	{ // m_render_aabb / m_up_dir, testbed.cu:5319-5320
		ngp_aabb rb = box;
		if (!render_aabb.is_empty()) for (int k = 0; k < 3; ++k) { rb.min[k] = render_aabb.min[k]; rb.max[k] = render_aabb.max[k]; }
		snap.set("render_aabb", jbox(rb));
	}
	{ const float sun[3] = {0.577f, 0.577f, 0.577f}; snap.set("up_dir", jvec(up_dir.data(), 3)); snap.set("sun_dir", jvec(sun, 3)); }
	snap.set("exposure", jnum(exposure)); snap.set("background_color", jvec(background_color.data(), 4));
	Value cam = jobj();
	cam.set("matrix", jmat_cols(m_camera.data(), 4, 3)); cam.set("fov_axis", jnum(fov_axis));
	cam.set("relative_focal_length", jvec(relative_focal_length.data(), 2)); cam.set("screen_center", jvec(screen_center.data(), 2));
	cam.set("zoom", jnum(zoom)); cam.set("scale", jnum(m_scale)); cam.set("aperture_size", jnum(0.0)); cam.set("autofocus", jbool(false));
	{ const float t[3] = {0.5f, 0.5f, 0.5f}; cam.set("autofocus_target", jvec(t, 3)); } cam.set("autofocus_depth", jnum(0.5));
	snap.set("camera", cam);
	root.set("snapshot", snap);

	std::string bytes;
	msgpack_lite::pack(root, bytes);
	if (ends_with_ci(path, ".ingp")) bytes = msgpack_lite::zlib_compress(bytes);
	std::ofstream f{path, std::ios::binary};
	if (!f) throw std::runtime_error{"Could not open '" + path + "' for writing."};
	f.write(bytes.data(), (std::streamsize)bytes.size());
	m_network_config_path = path;
}

void Testbed::load_snapshot(const std::string& path) {
	std::ifstream f{path, std::ios::binary};
	if (!f) throw std::runtime_error{"Snapshot '" + path + "' does not exist."};
	std::string bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
	if (ends_with_ci(path, ".ingp")) bytes = msgpack_lite::zlib_decompress(bytes.data(), bytes.size());
	Value root = msgpack_lite::unpack(bytes.data(), bytes.size());
	if (!root.is_object() || !root.has("snapshot")) throw std::runtime_error{"File '" + path + "' does not contain a snapshot."};
	const Value& snap = root["snapshot"];
	if (snap.num("version", 0) < 1) throw std::runtime_error{"Snapshot uses an old format and can not be loaded."};
	if (snap.has("mode") && snap.str("mode", "") != "nerf") throw std::runtime_error{"Only NeRF snapshots are supported by this build."};
	if ((int)snap.num("density_grid_size", 0) != 128) throw std::runtime_error{"Incompatible grid size."};
	const Value& jn = snap["nerf"];
	const int aabb_scale = (int)jn.num("aabb_scale", nerf.training.dataset.aabb_scale);
	if (nerf.training.dataset.n_images == 0) {
		// No training data loaded: restore the dataset METADATA embedded in the snapshot (from_json(NerfDataset), json_binding.h:141-190;
		// testbed.cu:5386-5400), enough to render / evaluate (`run.py --load_snapshot x.ingp` without --scene).  There are no pixels:
		// training stays off until load_training_data is called.
		nerf.training.dataset = dataset_from_json(jn["dataset"], aabb_scale);
		mode = ETestbedMode::Nerf;
		load_nerf_post();
		m_render_lens_mode = nerf.training.dataset.metadata[0].lens_mode; m_render_lens_params = nerf.training.dataset.metadata[0].lens_params;
		m_dataset_dirty = true; shall_train = false;
	}
	if (aabb_scale != nerf.training.dataset.aabb_scale) throw std::runtime_error{"Snapshot aabb_scale differs from the loaded dataset."};
	m_network_config = root;
	destroy_trainer();
	ensure_trainer();
	uint64_t n_params = 0, n_mlp = 0;
	NGP_CHECK(ngp_model_n_params(m_model, &n_params, &n_mlp));
	const Value& opt = snap.has("ngp_hip_optimizer") ? snap["ngp_hip_optimizer"] : snap["optimizer"]; // ("ngp_hip_optimizer" / "otype": "ngp_hip": files written by rounds 1-4 of this library)
	// tcnn's shape (see save_snapshot): the Adam level is the innermost "nested" that carries the moments
	const Value* adam = nullptr; const Value* ema = nullptr;
	for (const Value* lv = &snap["optimizer"]; lv && lv->is_object(); lv = lv->has("nested") ? &(*lv)["nested"] : nullptr) {
		if (lv->has("weights_ema_binary")) ema = lv;
		if (lv->has("first_moments_binary") && lv->has("second_moments_binary")) { adam = lv; break; }
	}
	if (opt.is_object() && opt.str("otype", "") == "ngp_hip" && opt["state_binary"].type == Value::Binary) {
		NGP_CHECK(ngp_model_deserialize_host(m_model, opt["state_binary"].bin.data(), opt["state_binary"].bin.size()));
	} else if (adam) { // Trainer::deserialize with optimizer state [tcnn, from memory]
		const Value& pb = snap["params_binary"];
		if ((uint64_t)snap.num("n_params", 0) != n_params || pb.type != Value::Binary) throw std::runtime_error{"Snapshot parameters do not match the network config."};
		const size_t nb = (size_t)n_params * 4;
		struct { uint64_t n_params; uint32_t step, with_optimizer; float lr; } h = {n_params, (uint32_t)adam->num("current_step", 0), 1u, (float)adam->num("base_learning_rate", 1e-2)};
		std::vector<uint8_t> blob(ngp_model_state_offset(n_params, NGP_STATE_N_SECTIONS));
		NGP_CHECK(ngp_model_state_header(blob.data(), blob.size(), 1, &h.n_params, &h.step, &h.lr, &h.with_optimizer));
		uint8_t* pbase = blob.data() + ngp_model_state_offset(n_params, 0);
		auto as_f32 = [&](const Value& v, int k, const char* what, bool allow_half) {
			if (v.type == Value::Binary && v.bin.size() == nb) { memcpy(pbase + (size_t)k * nb, v.bin.data(), nb); return; }
			if (allow_half && v.type == Value::Binary && v.bin.size() == nb / 2) { float* d = (float*)(pbase + (size_t)k * nb); for (uint64_t i = 0; i < n_params; ++i) { uint16_t hh; memcpy(&hh, &v.bin[i * 2], 2); d[i] = f16_to_f32(hh); } return; }
			throw std::runtime_error{std::string("Snapshot optimizer state: '") + what + "' has the wrong size."};
		};
		// master parameters: this library's private fp32 copy when present, else "params_binary" cast up (what tcnn's deserialize does)
		if (snap["ngp_hip_master_binary"].type == Value::Binary) as_f32(snap["ngp_hip_master_binary"], 0, "ngp_hip_master_binary", false);
		else if (snap.str("params_type", "__half") == "float") as_f32(pb, 0, "params_binary", false);
		else as_f32(pb, 0, "params_binary", true);
		as_f32((*adam)["first_moments_binary"], 1, "first_moments_binary", false);
		as_f32((*adam)["second_moments_binary"], 2, "second_moments_binary", false);
		if (adam->has("param_steps_binary")) as_f32((*adam)["param_steps_binary"], 3, "param_steps_binary", false); // (uint32 counters: same width)
		else { uint32_t* d = (uint32_t*)(pbase + 3 * nb); for (uint64_t i = 0; i < n_params; ++i) d[i] = h.step; } // [tcnn: older files have no per-parameter counters]
		if (snap["ngp_hip_ema_binary"].type == Value::Binary) as_f32(snap["ngp_hip_ema_binary"], 4, "ngp_hip_ema_binary", false); // this library's fp32 state of a "full_precision" EMA
		else if (ema) as_f32((*ema)["weights_ema_binary"], 4, "weights_ema_binary", true);
		else memcpy(pbase + 4 * nb, pbase, nb); // no EMA level: the inference parameters are the parameters
		NGP_CHECK(ngp_model_deserialize_host(m_model, blob.data(), blob.size()));
	} else { // Trainer::deserialize without optimizer state: parameters only, in the precision named by "params_type"
		const Value& pb = snap["params_binary"];
		if (pb.type != Value::Binary || (uint64_t)snap.num("n_params", 0) != n_params) throw std::runtime_error{"Snapshot parameters do not match the network config."};
		std::vector<float> p(n_params);
		if (snap.str("params_type", "__half") == "float") { if (pb.bin.size() != n_params * 4) throw std::runtime_error{"Snapshot params_binary has the wrong size."}; memcpy(p.data(), pb.bin.data(), n_params * 4); }
		else { if (pb.bin.size() != n_params * 2) throw std::runtime_error{"Snapshot params_binary has the wrong size."}; for (uint64_t i = 0; i < n_params; ++i) { uint16_t h; memcpy(&h, &pb.bin[i * 2], 2); p[i] = f16_to_f32(h); } }
		NGP_CHECK(ngp_model_set_params_host(m_model, p.data(), n_params));
	}
	const Value& gb = snap["density_grid_binary"];
	const uint64_t gf = (uint64_t)128 * 128 * 128 * (nerf.max_cascade + 1);
	if (gb.type == Value::Binary && gb.bin.size() == gf * 2) {
		std::vector<float> grid(gf);
		for (uint64_t i = 0; i < gf; ++i) { uint16_t h; memcpy(&h, &gb.bin[i * 2], 2); grid[i] = f16_to_f32(h); }
		NGP_CHECK(ngp_nerf_set_density_grid_host(m_nerf, nullptr, grid.data(), grid.size()));
	} else if (gb.type == Value::Binary && !gb.bin.empty()) throw std::runtime_error{"Incompatible number of grid cascades."};
	{ // m_nerf.training.extra_dims_opt = snapshot["nerf"]["extra_dims_opt"]; update_extra_dims() (testbed.cu:5482-5486)
		const Value& eo = jn["extra_dims_opt"];
		const uint32_t ne = nerf.training.dataset.n_extra_dims();
		if (ne && eo.is_array() && eo.size() > 0) {
			const uint32_t n_img = (uint32_t)std::min<size_t>(eo.size(), nerf.training.dataset.n_images);
			std::vector<float> var((size_t)n_img * ne), m1(var.size()), m2(var.size()); std::vector<uint32_t> iter(n_img, 0u);
			for (uint32_t i = 0; i < n_img; ++i) {
				const Value& o = eo.at(i);
				if (o["variable"].size() != ne || o["first_moment"].size() != ne || o["second_moment"].size() != ne) throw std::runtime_error{"Snapshot extra_dims_opt does not match the dataset's extra dims."};
				for (uint32_t k = 0; k < ne; ++k) { var[(size_t)i * ne + k] = (float)o["variable"].at(k).n; m1[(size_t)i * ne + k] = (float)o["first_moment"].at(k).n; m2[(size_t)i * ne + k] = (float)o["second_moment"].at(k).n; }
				iter[i] = (uint32_t)o.num("iter", 0);
			}
			NGP_CHECK(ngp_nerf_set_extra_dims_optimizer(m_nerf, var.data(), m1.data(), m2.data(), iter.data(), n_img));
		}
	}
	training_step = (uint32_t)snap.num("training_step", 0);
	loss = (float)snap.num("loss", 0.0);
	NGP_CHECK(ngp_nerf_set_training_step(m_nerf, training_step));
	if (jn["rgb"].is_object() && jn["rgb"].num("rays_per_batch", 0) > 0) NGP_CHECK(ngp_nerf_set_rays_per_batch(m_nerf, (uint32_t)jn["rgb"].num("rays_per_batch", 4096)));
	exposure = (float)snap.num("exposure", exposure);
	if (snap["background_color"].is_array() && snap["background_color"].size() == 4) for (int k = 0; k < 4; ++k) background_color[k] = (float)snap["background_color"].at(k).n;
	const Value& cam = snap["camera"];
	if (cam.is_object()) {
		if (cam["matrix"].is_array() && cam["matrix"].size() == 4) for (int c = 0; c < 4; ++c) for (int r = 0; r < 3; ++r) m_camera[c * 3 + r] = (float)cam["matrix"].at(c).at(r).n;
		fov_axis = (int)cam.num("fov_axis", fov_axis);
		if (cam["relative_focal_length"].size() == 2) for (int k = 0; k < 2; ++k) relative_focal_length[k] = (float)cam["relative_focal_length"].at(k).n;
		if (cam["screen_center"].size() == 2) for (int k = 0; k < 2; ++k) screen_center[k] = (float)cam["screen_center"].at(k).n;
		zoom = (float)cam.num("zoom", zoom); m_scale = (float)cam.num("scale", m_scale);
	}
	{ // m_render_aabb, m_up_dir (testbed.cu:5456-5459); the crop box orientation travels with them
		const Value& rb = snap["render_aabb"];
		if (rb.is_object() && rb["min"].size() == 3 && rb["max"].size() == 3) for (int k = 0; k < 3; ++k) { render_aabb.min[k] = (float)rb["min"].at(k).n; render_aabb.max[k] = (float)rb["max"].at(k).n; }
		if (snap["up_dir"].size() == 3) for (int k = 0; k < 3; ++k) up_dir[k] = (float)snap["up_dir"].at(k).n;
		const Value& rl = snap["render_aabb_to_local"];
		if (rl.is_array() && rl.size() == 3) for (int c = 0; c < 3; ++c) if (rl.at(c).size() == 3) for (int r = 0; r < 3; ++r) render_aabb_to_local[c * 3 + r] = (float)rl.at(c).at(r).n;
	}
	m_network_config_path = path;
}

// test / tooling hook: msgpack (optionally zlib-framed) -> Value -> msgpack, exercising reader and writer on arbitrary documents
std::string msgpack_repack(const std::string& data, bool input_compressed, bool output_compressed) {
	const std::string raw = input_compressed ? msgpack_lite::zlib_decompress(data.data(), data.size()) : data;
	const mini_json::Value v = msgpack_lite::unpack(raw.data(), raw.size());
	std::string out; msgpack_lite::pack(v, out);
	return output_compressed ? msgpack_lite::zlib_compress(out) : out;
}

} // namespace ngp_host
