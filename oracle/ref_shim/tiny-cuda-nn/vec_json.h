// oracle/ref_shim: stand-in for tiny-cuda-nn/vec_json.h (absent from the reference mount) -- TEST INFRASTRUCTURE ONLY.
// [tcnn, from memory of the public repository, unverifiable here]: a vector is a JSON array of its N components; a matrix is a JSON array of its COLUMNS, each an array
// of the column's components (tcnn's matrices are column-major: mat4x3 = 4 columns of vec3).  The product (host/testbed.cpp jvec / jmat_cols) writes the same shapes.
#pragma once
#include <json/json.hpp>
#include <tiny-cuda-nn/common.h>
namespace tcnn {
template <typename T, uint32_t N> void to_json(nlohmann::json& j, const tvec<T, N>& v) { j = nlohmann::json::array(); for (uint32_t i = 0; i < N; ++i) j.push_back(v[i]); }
template <typename T, uint32_t N> void from_json(const nlohmann::json& j, tvec<T, N>& v) { for (uint32_t i = 0; i < N; ++i) v[i] = j.at((size_t)i).template get<T>(); }
template <typename T, uint32_t C, uint32_t R> void to_json(nlohmann::json& j, const tmat<T, C, R>& m) { j = nlohmann::json::array(); for (uint32_t c = 0; c < C; ++c) j.push_back(m[c]); }
template <typename T, uint32_t C, uint32_t R> void from_json(const nlohmann::json& j, tmat<T, C, R>& m) { for (uint32_t c = 0; c < C; ++c) from_json(j.at((size_t)c), m[c]); }
} // namespace tcnn
