// mini_json.hpp -- a small self-contained JSON reader (objects, arrays, numbers, strings, bools, null,
// // and /* */ comments) for the network-config files (configs/nerf/*.json) and transforms.json.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace mini_json {

struct Value {
	enum Type { Null, Bool, Number, String, Array, Object, Binary } type = Null; // Binary: msgpack bin (snapshots), dumped as null in JSON text
	bool b = false;
	double n = 0;
	std::string s;
	std::vector<unsigned char> bin;
	std::vector<Value> arr;
	std::vector<std::pair<std::string, Value>> obj; // insertion-ordered

	bool is_object() const { return type == Object; }
	bool is_array() const { return type == Array; }
	bool is_number() const { return type == Number; }
	bool is_string() const { return type == String; }
	size_t size() const { return type == Array ? arr.size() : type == Object ? obj.size() : 0; }
	bool has(const std::string& k) const { for (auto& kv : obj) if (kv.first == k) return true; return false; }
	const Value& operator[](const std::string& k) const {
		static const Value null_value;
		for (auto& kv : obj) if (kv.first == k) return kv.second;
		return null_value;
	}
	const Value& at(size_t i) const { static const Value null_value; return i < arr.size() ? arr[i] : null_value; }
	double num(const std::string& k, double dflt) const { const Value& v = (*this)[k]; return v.type == Number ? v.n : (v.type == Bool ? (v.b ? 1.0 : 0.0) : dflt); }
	std::string str(const std::string& k, const std::string& dflt) const { const Value& v = (*this)[k]; return v.type == String ? v.s : dflt; }
	bool boolean(const std::string& k, bool dflt) const { const Value& v = (*this)[k]; return v.type == Bool ? v.b : (v.type == Number ? v.n != 0 : dflt); }
	Value* find(const std::string& k) { for (auto& kv : obj) if (kv.first == k) return &kv.second; return nullptr; }
	void set(const std::string& k, const Value& v) { if (Value* p = find(k)) *p = v; else obj.push_back({k, v}); }
};

struct Parser {
	const char* p; const char* end; std::string err;
	void ws() {
		for (;;) {
			while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
			if (p + 1 < end && p[0] == '/' && p[1] == '/') { while (p < end && *p != '\n') ++p; continue; }
			if (p + 1 < end && p[0] == '/' && p[1] == '*') { p += 2; while (p + 1 < end && !(p[0] == '*' && p[1] == '/')) ++p; p += 2; continue; }
			break;
		}
	}
	bool fail(const char* m) { if (err.empty()) err = m; return false; }
	bool parse_string(std::string& out) {
		if (p >= end || *p != '"') return fail("expected string");
		++p;
		while (p < end && *p != '"') {
			if (*p == '\\' && p + 1 < end) {
				++p;
				switch (*p) {
					case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
					case 'b': out += '\b'; break; case 'f': out += '\f'; break;
					case 'u': { if (p + 4 < end) { unsigned c = (unsigned)strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16); out += (char)(c < 128 ? c : '?'); p += 4; } break; }
					default: out += *p;
				}
				++p;
			} else out += *p++;
		}
		if (p >= end) return fail("unterminated string");
		++p;
		return true;
	}
	bool parse_value(Value& v) {
		ws();
		if (p >= end) return fail("unexpected end");
		if (*p == '{') {
			v.type = Value::Object; ++p; ws();
			if (p < end && *p == '}') { ++p; return true; }
			for (;;) {
				ws(); std::string k;
				if (!parse_string(k)) return false;
				ws(); if (p >= end || *p != ':') return fail("expected ':'");
				++p; Value c;
				if (!parse_value(c)) return false;
				v.obj.push_back({k, c}); ws();
				if (p < end && *p == ',') { ++p; ws(); if (p < end && *p == '}') { ++p; return true; } continue; }
				if (p < end && *p == '}') { ++p; return true; }
				return fail("expected ',' or '}'");
			}
		}
		if (*p == '[') {
			v.type = Value::Array; ++p; ws();
			if (p < end && *p == ']') { ++p; return true; }
			for (;;) {
				Value c;
				if (!parse_value(c)) return false;
				v.arr.push_back(c); ws();
				if (p < end && *p == ',') { ++p; ws(); if (p < end && *p == ']') { ++p; return true; } continue; }
				if (p < end && *p == ']') { ++p; return true; }
				return fail("expected ',' or ']'");
			}
		}
		if (*p == '"') { v.type = Value::String; return parse_string(v.s); }
		if (end - p >= 4 && std::string(p, p + 4) == "true") { v.type = Value::Bool; v.b = true; p += 4; return true; }
		if (end - p >= 5 && std::string(p, p + 5) == "false") { v.type = Value::Bool; v.b = false; p += 5; return true; }
		if (end - p >= 4 && std::string(p, p + 4) == "null") { v.type = Value::Null; p += 4; return true; }
		char* e = nullptr;
		double d = strtod(p, &e);
		if (e == p) return fail("unexpected token");
		v.type = Value::Number; v.n = d; p = e;
		return true;
	}
};

inline bool parse(const char* text, Value& out, std::string& err) {
	if (!text) { err = "null text"; return false; }
	Parser ps; ps.p = text; ps.end = text + std::char_traits<char>::length(text);
	if (!ps.parse_value(out)) { err = ps.err; return false; }
	ps.ws();
	if (ps.p != ps.end) { err = "trailing characters"; return false; }
	return true;
}

inline void dump_to(const Value& v, std::string& out) {
	switch (v.type) {
		case Value::Null: case Value::Binary: out += "null"; break;
		case Value::Bool: out += v.b ? "true" : "false"; break;
		case Value::Number: { char buf[40]; snprintf(buf, sizeof(buf), "%.17g", v.n); out += buf; break; }
		case Value::String: {
			out += '"';
			for (char c : v.s) { if (c == '"' || c == '\\') { out += '\\'; out += c; } else if (c == '\n') out += "\\n"; else out += c; }
			out += '"'; break; }
		case Value::Array: { out += '['; for (size_t i = 0; i < v.arr.size(); ++i) { if (i) out += ','; dump_to(v.arr[i], out); } out += ']'; break; }
		case Value::Object: { out += '{'; for (size_t i = 0; i < v.obj.size(); ++i) { if (i) out += ','; out += '"' + v.obj[i].first + "\":"; dump_to(v.obj[i].second, out); } out += '}'; break; }
	}
}
inline std::string dump(const Value& v) { std::string s; dump_to(v, s); return s; }

// RFC 7386 merge-patch, as used for the "parent" inheritance of network configs (testbed.cu:86-97)
inline void merge_patch(Value& target, const Value& patch) {
	if (!patch.is_object()) { target = patch; return; }
	if (!target.is_object()) { target = Value(); target.type = Value::Object; }
	for (auto& kv : patch.obj) {
		if (kv.second.type == Value::Null) {
			for (size_t i = 0; i < target.obj.size(); ++i) if (target.obj[i].first == kv.first) { target.obj.erase(target.obj.begin() + i); break; }
		} else {
			Value* t = target.find(kv.first);
			if (!t) { target.obj.push_back({kv.first, Value()}); t = &target.obj.back().second; }
			merge_patch(*t, kv.second);
		}
	}
}

} // namespace mini_json
