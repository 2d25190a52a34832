/* oracle/_ref/libngpjson_ref.so -- TEST INFRASTRUCTURE ONLY: the reference's OWN snapshot (de)serialisers, include/neural-graphics-primitives/json_binding.h (BoundingBox,
 * Lens, TrainingXForm, NerfDataset: to_json / from_json) and adam_optimizer.h (VarAdamOptimizer::to_json / from_json), compiled from the reference's tree where they lie
 * against oracle/ref_shim (a value type with nlohmann::json's calling conventions + tcnn's vector encoding restated from memory -- both absent from the mount).
 * Every export parses a JSON text, runs the reference's from_json, then its to_json, and returns the text: tests/test_ref_snapshot.py feeds the subtrees the PRODUCT wrote
 * (snapshot.nerf.dataset, .aabb / .render_aabb, .nerf.extra_dims_opt[i], a dataset's lenses) and demands (a) that the reference's reader accepts them and (b) that what the
 * reference's writer makes of the result is the same document.  Never shipped, never linked by the product; built by oracle/Makefile when /root/reference is present. */
} // namespace ngp   (part 2 of 2: oracle/Makefile pipes ref_json_pre.hpp + the text of `class VarAdamOptimizer` of adam_optimizer.h, read where it lies, + this file)
#include <cstring>
using namespace ngp;
#define REF extern "C" __attribute__((visibility("default")))
static thread_local std::string g_out;
template <typename F> static const char* guarded(F f) {
	try { g_out = f(); } catch (const std::exception& e) { g_out = std::string("!error: ") + e.what(); }
	return g_out.c_str();
}
template <typename T> static std::string roundtrip(const char* text) { T v{}; nlohmann::json::parse(text).take(v); nlohmann::json out; out = v; return out.dump(); }
REF const char* ref_json_roundtrip_bounding_box(const char* text) { return guarded([&] { return roundtrip<BoundingBox>(text); }); }
REF const char* ref_json_roundtrip_lens(const char* text) { return guarded([&] { return roundtrip<Lens>(text); }); }
REF const char* ref_json_roundtrip_xform(const char* text) { return guarded([&] { return roundtrip<TrainingXForm>(text); }); }
REF const char* ref_json_roundtrip_dataset(const char* text) { return guarded([&] { return roundtrip<NerfDataset>(text); }); }
REF const char* ref_json_roundtrip_var_adam(const char* text) { return guarded([&] { VarAdamOptimizer o; o.from_json(nlohmann::json::parse(text)); nlohmann::json out; o.to_json(out); return out.dump(); }); }
/* the lens a from_json'ed Lens holds (mode, params[7]): what the reader UNDERSTOOD, for the modes whose JSON form does not carry every parameter */
REF int ref_json_lens_fields(const char* text, int* mode, float* params7) {
	try { Lens l{}; nlohmann::json::parse(text).take(l); *mode = (int)l.mode; for (int k = 0; k < 7; ++k) params7[k] = l.params[k]; return 0; } catch (...) { return 1; }
}
