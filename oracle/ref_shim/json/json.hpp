// oracle/ref_shim: stand-in for nlohmann/json (a tiny-cuda-nn dependency, absent from the reference mount) -- TEST INFRASTRUCTURE ONLY.  Just enough for headers of the
// reference that mention the type in (de)serialisation members the pins never call (adam_optimizer.h: to_json / from_json) to compile.
#pragma once
#include <string>
#include <vector>
namespace nlohmann {
class json {
public:
	json() = default;
	template <typename T> json(const T&) {}
	template <typename T> json& operator=(const T&) { return *this; }
	json& operator[](const std::string&) { return *this; }
	json& operator[](const char*) { return *this; }
	const json& at(const std::string&) const { return *this; }
	template <typename T> T get() const { return T{}; }
	template <typename T> operator T() const { return T{}; }
};
} // namespace nlohmann
