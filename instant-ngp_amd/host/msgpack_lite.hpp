// msgpack_lite.hpp -- MessagePack (de)serialisation of mini_json::Value plus zlib framing, for the reference's snapshot wire format:
// `.msgpack` = json::to_msgpack(config incl. "snapshot"), `.ingp` = the same bytes through zlib (testbed.cu:5345-5352, zstr::ostream).
// Number policy follows nlohmann::json's to_msgpack: integral values as the smallest (u)int type, other numbers as float32 when that
// is exact and float64 otherwise; binary blobs as bin 8/16/32.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include <zlib.h>
#include "../csrc/mini_json.hpp"

namespace msgpack_lite {
using mini_json::Value;

inline void put_be(std::string& o, uint64_t v, int bytes) { for (int i = bytes - 1; i >= 0; --i) o.push_back((char)((v >> (8 * i)) & 0xff)); }

inline void pack(const Value& v, std::string& o) {
	switch (v.type) {
		case Value::Null: o.push_back((char)0xc0); break;
		case Value::Bool: o.push_back((char)(v.b ? 0xc3 : 0xc2)); break;
		case Value::Number: {
			const double d = v.n;
			if (std::isfinite(d) && d == std::floor(d) && std::fabs(d) < 9.2e18) {
				if (d >= 0) {
					const uint64_t u = (uint64_t)d;
					if (u < 128) o.push_back((char)u);
					else if (u <= 0xff) { o.push_back((char)0xcc); put_be(o, u, 1); }
					else if (u <= 0xffff) { o.push_back((char)0xcd); put_be(o, u, 2); }
					else if (u <= 0xffffffffull) { o.push_back((char)0xce); put_be(o, u, 4); }
					else { o.push_back((char)0xcf); put_be(o, u, 8); }
				} else {
					const int64_t i = (int64_t)d;
					if (i >= -32) o.push_back((char)(int8_t)i);
					else if (i >= -128) { o.push_back((char)0xd0); put_be(o, (uint64_t)i, 1); }
					else if (i >= -32768) { o.push_back((char)0xd1); put_be(o, (uint64_t)i, 2); }
					else if (i >= -2147483648ll) { o.push_back((char)0xd2); put_be(o, (uint64_t)i, 4); }
					else { o.push_back((char)0xd3); put_be(o, (uint64_t)i, 8); }
				}
			} else if ((double)(float)d == d || std::isnan(d)) {
				const float f = (float)d; uint32_t u; memcpy(&u, &f, 4); o.push_back((char)0xca); put_be(o, u, 4);
			} else { uint64_t u; memcpy(&u, &d, 8); o.push_back((char)0xcb); put_be(o, u, 8); }
			break; }
		case Value::String: {
			const size_t n = v.s.size();
			if (n < 32) o.push_back((char)(0xa0 | n));
			else if (n <= 0xff) { o.push_back((char)0xd9); put_be(o, n, 1); }
			else if (n <= 0xffff) { o.push_back((char)0xda); put_be(o, n, 2); }
			else { o.push_back((char)0xdb); put_be(o, n, 4); }
			o.append(v.s); break; }
		case Value::Binary: {
			const size_t n = v.bin.size();
			if (n <= 0xff) { o.push_back((char)0xc4); put_be(o, n, 1); }
			else if (n <= 0xffff) { o.push_back((char)0xc5); put_be(o, n, 2); }
			else { if (n > 0xffffffffull) throw std::runtime_error{"msgpack: binary blob larger than 4 GiB"}; o.push_back((char)0xc6); put_be(o, n, 4); }
			o.append((const char*)v.bin.data(), n); break; }
		case Value::Array: {
			const size_t n = v.arr.size();
			if (n < 16) o.push_back((char)(0x90 | n));
			else if (n <= 0xffff) { o.push_back((char)0xdc); put_be(o, n, 2); }
			else { o.push_back((char)0xdd); put_be(o, n, 4); }
			for (const Value& e : v.arr) pack(e, o);
			break; }
		case Value::Object: {
			const size_t n = v.obj.size();
			if (n < 16) o.push_back((char)(0x80 | n));
			else if (n <= 0xffff) { o.push_back((char)0xde); put_be(o, n, 2); }
			else { o.push_back((char)0xdf); put_be(o, n, 4); }
			for (const auto& kv : v.obj) { Value k; k.type = Value::String; k.s = kv.first; pack(k, o); pack(kv.second, o); }
			break; }
	}
}

struct Reader {
	const uint8_t* p; const uint8_t* end;
	void need(size_t n) const { if ((size_t)(end - p) < n) throw std::runtime_error{"msgpack: truncated input"}; }
	uint64_t be(int bytes) { need(bytes); uint64_t v = 0; for (int i = 0; i < bytes; ++i) v = (v << 8) | *p++; return v; }
	void str(Value& v, size_t n) { need(n); v.type = Value::String; v.s.assign((const char*)p, n); p += n; }
	void bin(Value& v, size_t n) { need(n); v.type = Value::Binary; v.bin.assign(p, p + n); p += n; }
	void arr(Value& v, size_t n) { v.type = Value::Array; v.arr.resize(n); for (size_t i = 0; i < n; ++i) read(v.arr[i]); }
	void map(Value& v, size_t n) {
		v.type = Value::Object; v.obj.reserve(n);
		for (size_t i = 0; i < n; ++i) { Value k; read(k); if (k.type != Value::String) throw std::runtime_error{"msgpack: non-string map key"}; Value e; read(e); v.obj.emplace_back(std::move(k.s), std::move(e)); }
	}
	void num(Value& v, double d) { v.type = Value::Number; v.n = d; }
	void read(Value& v) {
		need(1);
		const uint8_t t = *p++;
		if (t < 0x80) return num(v, t);
		if (t >= 0xe0) return num(v, (int8_t)t);
		if ((t & 0xf0) == 0x80) return map(v, t & 0x0f);
		if ((t & 0xf0) == 0x90) return arr(v, t & 0x0f);
		if ((t & 0xe0) == 0xa0) return str(v, t & 0x1f);
		switch (t) {
			case 0xc0: v.type = Value::Null; return;
			case 0xc2: v.type = Value::Bool; v.b = false; return;
			case 0xc3: v.type = Value::Bool; v.b = true; return;
			case 0xc4: return bin(v, be(1)); case 0xc5: return bin(v, be(2)); case 0xc6: return bin(v, be(4));
			case 0xca: { const uint32_t u = (uint32_t)be(4); float f; memcpy(&f, &u, 4); return num(v, f); }
			case 0xcb: { const uint64_t u = be(8); double d; memcpy(&d, &u, 8); return num(v, d); }
			case 0xcc: return num(v, (double)be(1)); case 0xcd: return num(v, (double)be(2)); case 0xce: return num(v, (double)be(4)); case 0xcf: return num(v, (double)be(8));
			case 0xd0: return num(v, (int8_t)be(1)); case 0xd1: return num(v, (int16_t)be(2)); case 0xd2: return num(v, (int32_t)be(4)); case 0xd3: return num(v, (double)(int64_t)be(8));
			case 0xd9: return str(v, be(1)); case 0xda: return str(v, be(2)); case 0xdb: return str(v, be(4));
			case 0xdc: return arr(v, be(2)); case 0xdd: return arr(v, be(4));
			case 0xde: return map(v, be(2)); case 0xdf: return map(v, be(4));
			default: throw std::runtime_error{"msgpack: unsupported type byte (ext / reserved)"};
		}
	}
};
inline Value unpack(const void* data, size_t n) { Reader r{(const uint8_t*)data, (const uint8_t*)data + n}; Value v; r.read(v); return v; }

// zlib stream (what zstr::ostream writes); reading auto-detects zlib / gzip headers like zstr::istream
inline std::string zlib_compress(const std::string& in, int level = Z_DEFAULT_COMPRESSION) {
	uLongf bound = compressBound((uLong)in.size());
	std::string out(bound, '\0');
	if (compress2((Bytef*)out.data(), &bound, (const Bytef*)in.data(), (uLong)in.size(), level) != Z_OK) throw std::runtime_error{"zlib: compress failed"};
	out.resize(bound);
	return out;
}
inline std::string zlib_decompress(const void* data, size_t n) {
	z_stream zs; memset(&zs, 0, sizeof(zs));
	if (inflateInit2(&zs, 15 + 32) != Z_OK) throw std::runtime_error{"zlib: inflateInit failed"};
	zs.next_in = (Bytef*)data; zs.avail_in = (uInt)n;
	std::string out; std::vector<char> buf(1 << 20);
	int rc;
	do {
		zs.next_out = (Bytef*)buf.data(); zs.avail_out = (uInt)buf.size();
		rc = inflate(&zs, Z_NO_FLUSH);
		if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&zs); throw std::runtime_error{"zlib: corrupt stream"}; }
		out.append(buf.data(), buf.size() - zs.avail_out);
	} while (rc != Z_STREAM_END);
	inflateEnd(&zs);
	return out;
}

} // namespace msgpack_lite
