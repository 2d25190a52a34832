// oracle/ref_shim: stand-in for nlohmann/json (a tiny-cuda-nn dependency, absent from the reference mount) -- TEST INFRASTRUCTURE ONLY.
// A small value type with the part of nlohmann::json's interface that the reference's OWN (de)serialisation code uses -- include/neural-graphics-primitives/
// json_binding.h (BoundingBox, Lens, TrainingXForm, NerfDataset) and adam_optimizer.h (VarAdamOptimizer::to_json / from_json) -- so that those functions compile from
// where they lie and run: operator[] / at / contains / value / emplace_back / get<T>() / implicit conversion, user types through ADL to_json / from_json like the real
// library, plus a JSON text reader / writer for the test wrappers (oracle/ref_json_wrapper.cpp).  Object keys keep insertion order; numbers are doubles with an
// "integer" flag.  Nothing of nlohmann's implementation is reproduced: only the calling conventions the reference's code relies on.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>
namespace nlohmann {
class json;
namespace shim_detail {
// does `to_json(json&, const T&)` / `from_json(const json&, T&)` exist through ADL?
template <typename T, typename = void> struct has_adl_to : std::false_type {};
template <typename T> struct has_adl_to<T, decltype(to_json(std::declval<json&>(), std::declval<const T&>()), void())> : std::true_type {};
template <typename T, typename = void> struct has_adl_from : std::false_type {};
template <typename T> struct has_adl_from<T, decltype(from_json(std::declval<const json&>(), std::declval<T&>()), void())> : std::true_type {};
template <typename T, typename = void> struct has_member_to : std::false_type {};
template <typename T> struct has_member_to<T, decltype(std::declval<const T&>().to_json(std::declval<json&>()), void())> : std::true_type {};
template <typename T, typename = void> struct has_member_from : std::false_type {};
template <typename T> struct has_member_from<T, decltype(std::declval<T&>().from_json(std::declval<const json&>()), void())> : std::true_type {};
template <typename T> struct is_vector : std::false_type {};
template <typename T, typename A> struct is_vector<std::vector<T, A>> : std::true_type {};
} // namespace shim_detail

class json {
public:
	enum class kind { null, boolean, number, string, array, object };
	kind k = kind::null;
	bool b = false; double num = 0; bool is_int = false; std::string str;
	std::vector<json> arr; std::vector<std::pair<std::string, json>> obj;

	json() = default;
	json(const json&) = default; json(json&&) = default;
	json& operator=(const json&) = default; json& operator=(json&&) = default;
	template <typename T, typename = typename std::enable_if<!std::is_same<typename std::decay<T>::type, json>::value>::type> json(const T& v) { put(v); }
	template <typename T, typename = typename std::enable_if<!std::is_same<typename std::decay<T>::type, json>::value>::type> json& operator=(const T& v) { json t; t.put(v); *this = std::move(t); return *this; }
	static json array() { json j; j.k = kind::array; return j; }
	static json object() { json j; j.k = kind::object; return j; }

	// ---- writing ----
	void put(bool v) { k = kind::boolean; b = v; }
	void put(const char* v) { k = kind::string; str = v; }
	void put(const std::string& v) { k = kind::string; str = v; }
	template <typename T> typename std::enable_if<std::is_arithmetic<T>::value && !std::is_same<T, bool>::value>::type put(const T& v) { k = kind::number; num = (double)v; is_int = std::is_integral<T>::value; }
	template <typename T> typename std::enable_if<std::is_enum<T>::value>::type put(const T& v) { k = kind::number; num = (double)(long long)v; is_int = true; }
	template <typename T> typename std::enable_if<shim_detail::is_vector<T>::value>::type put(const T& v) { k = kind::array; arr.clear(); for (const auto& e : v) { json t; t.put(e); arr.push_back(std::move(t)); } }
	template <typename T> typename std::enable_if<!std::is_arithmetic<T>::value && !std::is_enum<T>::value && !shim_detail::is_vector<T>::value && shim_detail::has_adl_to<T>::value>::type put(const T& v) { json t; to_json(t, v); *this = std::move(t); }
	template <typename T> typename std::enable_if<!std::is_arithmetic<T>::value && !std::is_enum<T>::value && !shim_detail::is_vector<T>::value && !shim_detail::has_adl_to<T>::value && shim_detail::has_member_to<T>::value>::type put(const T& v) { json t; v.to_json(t); *this = std::move(t); }

	json& operator[](const std::string& key) {
		if (k == kind::null) k = kind::object;
		if (k != kind::object) throw std::runtime_error("json: operator[](key) on a non-object");
		for (auto& kv : obj) if (kv.first == key) return kv.second;
		obj.emplace_back(key, json());
		return obj.back().second;
	}
	json& operator[](const char* key) { return (*this)[std::string(key)]; }
	const json& operator[](const std::string& key) const { return at(key); }
	const json& operator[](const char* key) const { return at(std::string(key)); }
	json& operator[](size_t i) { if (k == kind::null) k = kind::array; if (k != kind::array) throw std::runtime_error("json: operator[](index) on a non-array"); if (i >= arr.size()) arr.resize(i + 1); return arr[i]; }
	json& operator[](int i) { return (*this)[(size_t)i]; }
	const json& operator[](size_t i) const { return at(i); }
	const json& operator[](int i) const { return at((size_t)i); }
	json& emplace_back() { if (k == kind::null) k = kind::array; if (k != kind::array) throw std::runtime_error("json: emplace_back on a non-array"); arr.emplace_back(); return arr.back(); }
	template <typename T> void push_back(const T& v) { json t; t.put(v); emplace_back() = std::move(t); }
	void push_back(const json& v) { emplace_back() = v; }

	// ---- reading ----
	bool contains(const std::string& key) const { if (k != kind::object) return false; for (const auto& kv : obj) if (kv.first == key) return true; return false; }
	const json& at(const std::string& key) const { if (k == kind::object) for (const auto& kv : obj) if (kv.first == key) return kv.second; throw std::out_of_range("json: key '" + key + "' not found"); }
	json& at(const std::string& key) { if (k == kind::object) for (auto& kv : obj) if (kv.first == key) return kv.second; throw std::out_of_range("json: key '" + key + "' not found"); }
	const json& at(const char* key) const { return at(std::string(key)); }
	json& at(const char* key) { return at(std::string(key)); }
	const json& at(size_t i) const { if (k != kind::array || i >= arr.size()) throw std::out_of_range("json: array index out of range"); return arr[i]; }
	json& at(size_t i) { if (k != kind::array || i >= arr.size()) throw std::out_of_range("json: array index out of range"); return arr[i]; }
	const json& at(int i) const { return at((size_t)i); }
	json& at(int i) { return at((size_t)i); }
	size_t size() const { return k == kind::array ? arr.size() : k == kind::object ? obj.size() : k == kind::null ? 0 : 1; }
	bool is_array() const { return k == kind::array; }
	bool is_object() const { return k == kind::object; }

	void take(bool& v) const { if (k != kind::boolean) throw std::runtime_error("json: not a boolean"); v = b; }
	void take(std::string& v) const { if (k != kind::string) throw std::runtime_error("json: not a string"); v = str; }
	template <typename T> typename std::enable_if<std::is_arithmetic<T>::value && !std::is_same<T, bool>::value>::type take(T& v) const { if (k == kind::boolean) { v = (T)b; return; } if (k != kind::number) throw std::runtime_error("json: not a number"); v = (T)num; }
	template <typename T> typename std::enable_if<std::is_enum<T>::value>::type take(T& v) const { if (k != kind::number) throw std::runtime_error("json: not a number"); v = (T)(long long)num; }
	template <typename T> typename std::enable_if<shim_detail::is_vector<T>::value>::type take(T& v) const { if (k != kind::array) throw std::runtime_error("json: not an array"); v.clear(); for (const auto& e : arr) { typename T::value_type x{}; e.take(x); v.push_back(std::move(x)); } }
	template <typename T> typename std::enable_if<!std::is_arithmetic<T>::value && !std::is_enum<T>::value && !shim_detail::is_vector<T>::value && shim_detail::has_adl_from<T>::value>::type take(T& v) const { from_json(*this, v); }
	template <typename T> typename std::enable_if<!std::is_arithmetic<T>::value && !std::is_enum<T>::value && !shim_detail::is_vector<T>::value && !shim_detail::has_adl_from<T>::value && shim_detail::has_member_from<T>::value>::type take(T& v) const { v.from_json(*this); }
	template <typename T> T get() const { T v{}; take(v); return v; }
	template <typename T, typename = typename std::enable_if<!std::is_same<typename std::decay<T>::type, json>::value && !std::is_pointer<T>::value && !std::is_same<T, std::nullptr_t>::value>::type> operator T() const { return get<T>(); }
	template <typename T> T value(const std::string& key, const T& def) const { return contains(key) ? at(key).get<T>() : def; }
	std::string value(const std::string& key, const char* def) const { return contains(key) ? at(key).get<std::string>() : std::string(def); }

	// ---- JSON text (test wrappers) ----
	std::string dump() const { std::string s; dump_to(s); return s; }
	static json parse(const std::string& text) { size_t p = 0; json j = parse_value(text, p); skip_ws(text, p); if (p != text.size()) throw std::runtime_error("json: trailing characters"); return j; }

private:
	void dump_to(std::string& s) const {
		char buf[64];
		switch (k) {
			case kind::null: s += "null"; break;
			case kind::boolean: s += b ? "true" : "false"; break;
			case kind::number: if (is_int) snprintf(buf, sizeof buf, "%lld", (long long)num); else snprintf(buf, sizeof buf, "%.17g", num); s += buf; break;
			case kind::string: s += '"'; for (char c : str) { if (c == '"' || c == '\\') { s += '\\'; s += c; } else if (c == '\n') s += "\\n"; else s += c; } s += '"'; break;
			case kind::array: s += '['; for (size_t i = 0; i < arr.size(); ++i) { if (i) s += ','; arr[i].dump_to(s); } s += ']'; break;
			case kind::object: s += '{'; for (size_t i = 0; i < obj.size(); ++i) { if (i) s += ','; json key; key.put(obj[i].first); key.dump_to(s); s += ':'; obj[i].second.dump_to(s); } s += '}'; break;
		}
	}
	static void skip_ws(const std::string& t, size_t& p) { while (p < t.size() && (t[p] == ' ' || t[p] == '\n' || t[p] == '\t' || t[p] == '\r')) ++p; }
	static json parse_value(const std::string& t, size_t& p) {
		skip_ws(t, p);
		if (p >= t.size()) throw std::runtime_error("json: unexpected end");
		json j;
		const char c = t[p];
		if (c == '{') { j.k = kind::object; ++p; skip_ws(t, p); if (t[p] == '}') { ++p; return j; }
			for (;;) { json key = parse_value(t, p); if (key.k != kind::string) throw std::runtime_error("json: key"); skip_ws(t, p); if (t[p] != ':') throw std::runtime_error("json: ':'"); ++p; j.obj.emplace_back(key.str, parse_value(t, p)); skip_ws(t, p); if (t[p] == ',') { ++p; continue; } if (t[p] == '}') { ++p; return j; } throw std::runtime_error("json: object"); } }
		if (c == '[') { j.k = kind::array; ++p; skip_ws(t, p); if (t[p] == ']') { ++p; return j; }
			for (;;) { j.arr.push_back(parse_value(t, p)); skip_ws(t, p); if (t[p] == ',') { ++p; continue; } if (t[p] == ']') { ++p; return j; } throw std::runtime_error("json: array"); } }
		if (c == '"') { j.k = kind::string; ++p; while (p < t.size() && t[p] != '"') { if (t[p] == '\\' && p + 1 < t.size()) { ++p; j.str += t[p] == 'n' ? '\n' : t[p]; } else j.str += t[p]; ++p; } ++p; return j; }
		if (!t.compare(p, 4, "true")) { j.k = kind::boolean; j.b = true; p += 4; return j; }
		if (!t.compare(p, 5, "false")) { j.k = kind::boolean; j.b = false; p += 5; return j; }
		if (!t.compare(p, 4, "null")) { p += 4; return j; }
		char* end = nullptr; j.num = strtod(t.c_str() + p, &end); if (end == t.c_str() + p) throw std::runtime_error("json: value");
		j.k = kind::number; j.is_int = true; for (const char* q = t.c_str() + p; q < end; ++q) if (*q == '.' || *q == 'e' || *q == 'E' || *q == 'n' || *q == 'i') j.is_int = false;
		p = (size_t)(end - t.c_str());
		return j;
	}
};
} // namespace nlohmann
