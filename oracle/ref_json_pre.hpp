// oracle/_ref/libngpjson_ref.so, part 1 of 2 (see ref_json_wrapper.cpp) -- TEST INFRASTRUCTURE ONLY
#include <cstdint>
#include <cstddef>
#include <cmath>
#include <vector>
#include <neural-graphics-primitives/json_binding.h>
namespace ngp {
