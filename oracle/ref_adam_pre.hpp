// oracle/_ref/libngpadam_ref.so, part 1 of 2 (see ref_adam_wrapper.cpp) -- TEST INFRASTRUCTURE ONLY
#include <cstdint>
#include <cstddef>
#include <cmath>
#include <vector>
#include <json/json.hpp>
namespace ngp {
