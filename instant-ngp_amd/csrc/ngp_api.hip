// ngp_api.hip -- implementation of the C-ABI declared in include/ngp_hip.h (host side of libngp_hip.so).
// Owns device memory, derives layouts (hash-grid offset table, MFMA fragment permutations), and
// sequences the kernels of one training step on one HIP stream without host synchronisation.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include "ngp_device.hpp"
#include "ngp_kernels.hpp"
#include "mini_json.hpp"

using namespace ngp;

static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return 1; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
#define REQUIRE(c, msg) do { if (!(c)) return fail(msg); } while (0)

extern "C" const char* ngp_last_error(void) { return g_err.c_str(); }
extern "C" int ngp_device_available(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n > 0;
}

// ------------------------------------------------------------------------------------------------
// optional per-kernel timing with HIP events recorded on the launch stream (off by default)
// ------------------------------------------------------------------------------------------------
struct ProfEntry { int id; hipEvent_t a, b; };
static bool g_prof_on = false;
static std::vector<ProfEntry> g_prof;
static std::vector<hipEvent_t> g_event_pool;
static double g_prof_ms[P_COUNT]; static uint64_t g_prof_n[P_COUNT];
static const char* kProfNames[P_COUNT] = {"k_generate_training_samples", "k_inference", "k_compute_loss", "k_fill_rollover", "k_train_fwd_bwd", "k_wgrad",
	"k_wgrad_reduce", "k_optimizer", "k_inference<density_only>", "occupancy_grid_misc", "grad_memset", "counters", "k_grad_bin+accumulate", "k_encode_tiles_xcd", "k_train_fused"};
static hipEvent_t prof_event() {
	if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
	hipEvent_t e; (void)hipEventCreate(&e); return e;
}
struct ProfScope {
	int idx = -1; hipStream_t s;
	ProfScope(int id, hipStream_t st) : s(st) {
		if (!g_prof_on) return;
		ProfEntry e; e.id = id; e.a = prof_event(); e.b = prof_event();
		(void)hipEventRecord(e.a, s);
		g_prof.push_back(e); idx = (int)g_prof.size() - 1;
	}
	~ProfScope() { if (idx >= 0) (void)hipEventRecord(g_prof[idx].b, s); }
};
static void prof_collect() {
	for (auto& e : g_prof) {
		(void)hipEventSynchronize(e.b);
		float ms = 0.f;
		if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { g_prof_ms[e.id] += ms; g_prof_n[e.id] += 1; }
		g_event_pool.push_back(e.a); g_event_pool.push_back(e.b);
	}
	g_prof.clear();
}
// Box calibration (bench.py config.calibration): shader-clock probe.  One wavefront per SIMD runs a chain of DEPENDENT fp32 FMAs -- nothing but the VALU issue rate of
// a single wavefront, i.e. the shader clock the box grants a light kernel (the pool's boxes come in two classes that a GEMM or a copy does not tell apart, because those
// run power- or memory-bound: the compute- and latency-bound kernels of this library run 1.1 - 1.5 x slower on one class).  Returns nanoseconds per dependent FMA.
__global__ void __launch_bounds__(64) k_clock_probe(uint32_t iters, float* __restrict__ out) {
	float x = (float)threadIdx.x * 1e-3f, a = 1.0000001f, b = 1e-7f;
	for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
		for (int k = 0; k < 16; ++k) x = __builtin_fmaf(x, a, b);
	}
	if (x == 123.456f) out[0] = x; // never true: keeps the chain alive
}
extern "C" int ngp_debug_clock_probe(float* ns_per_dependent_fma_host) {
	REQUIRE(ns_per_dependent_fma_host, "ngp_debug_clock_probe: null argument");
	float* d = nullptr; HIPCHK(hipMalloc((void**)&d, 4));
	hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
	const uint32_t iters = 1u << 16; // x 16 FMAs = 1.05 M dependent operations per wavefront
	hipLaunchKernelGGL(k_clock_probe, dim3(1024), dim3(64), 0, nullptr, 1024u, d); // warm up (clock ramp)
	HIPCHK(hipEventRecord(e0, nullptr));
	hipLaunchKernelGGL(k_clock_probe, dim3(1024), dim3(64), 0, nullptr, iters, d);
	HIPCHK(hipEventRecord(e1, nullptr));
	HIPCHK(hipEventSynchronize(e1));
	float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
	*ns_per_dependent_fma_host = ms * 1e6f / ((float)iters * 16.f);
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(d);
	return 0;
}
extern "C" int ngp_profile_enable(int on) {
	prof_collect();
	g_prof_on = on != 0;
	if (on) { for (int i = 0; i < P_COUNT; ++i) { g_prof_ms[i] = 0; g_prof_n[i] = 0; } }
	return 0;
}
// host-side evaluation of the camera model of csrc/ngp_device.hpp (test hooks; the functions are __host__ __device__)
extern "C" int ngp_host_uv_to_ray(const ngp_image_meta* m, const float xform12[12], const float uv[2], float o_out[3], float d_out[3]) {
	f3 o, d; const f2 u = {uv[0], uv[1]};
	const bool ok = uv_to_ray(u, m->resolution, m->focal_length, ldm43(xform12), m->principal_point, m->lens_mode, m->lens_params, 0.0f, o, d);
	o_out[0] = o.x; o_out[1] = o.y; o_out[2] = o.z; d_out[0] = d.x; d_out[1] = d.y; d_out[2] = d.z;
	return ok ? 1 : 0;
}
// get_xform_given_rolling_shutter of csrc/ngp_device.hpp evaluated on the host (test hook)
extern "C" int ngp_host_xform_given_rolling_shutter(const ngp_xform* xform, const float rolling_shutter[4], const float uv[2], float motionblur_time, float xform12_out[12]) {
	const f2 u = {uv[0], uv[1]};
	const M43 m = xform_given_rolling_shutter(*xform, rolling_shutter, u, motionblur_time);
	for (int c = 0; c < 4; ++c) { xform12_out[c * 3 + 0] = m.c[c].x; xform12_out[c * 3 + 1] = m.c[c].y; xform12_out[c * 3 + 2] = m.c[c].z; }
	return 0;
}
extern "C" int ngp_host_pos_to_uv(const ngp_image_meta* m, const float xform12[12], const float pos[3], float uv_out[2]) {
	const f2 uv = pos_to_uv(mk3(pos[0], pos[1], pos[2]), m->resolution, m->focal_length, ldm43(xform12), m->principal_point, m->lens_mode, m->lens_params);
	uv_out[0] = uv.x; uv_out[1] = uv.y;
	return 0;
}
// NGP_DEBUG_FLAGS_OR (environment, ablation runs of unmodified callers): bits that stay set whatever ngp_debug_set_flags is given
static uint32_t env_debug_or() { static const uint32_t v = getenv("NGP_DEBUG_FLAGS_OR") ? (uint32_t)strtoul(getenv("NGP_DEBUG_FLAGS_OR"), nullptr, 0) : 0u; return v; }
static const bool g_debug_env_applied = [] { g_debug_flags |= env_debug_or(); return true; }();
extern "C" int ngp_debug_set_flags(uint32_t flags) { g_debug_flags = flags | env_debug_or(); return 0; }
extern "C" uint32_t ngp_debug_get_flags(void) { return g_debug_flags; } // what is in effect, NGP_DEBUG_FLAGS_OR included (bench.py records it: 0 = the production path)
static uint32_t env_debug2_or() { static const uint32_t v = getenv("NGP_DEBUG_FLAGS2_OR") ? (uint32_t)strtoul(getenv("NGP_DEBUG_FLAGS2_OR"), nullptr, 0) : 0u; return v; }
static const bool g_debug2_env_applied = [] { g_debug_flags2 |= env_debug2_or(); return true; }();
extern "C" int ngp_debug_set_flags2(uint32_t flags) { g_debug_flags2 = flags | env_debug2_or(); return 0; }
extern "C" uint32_t ngp_debug_get_flags2(void) { return g_debug_flags2; }
// Per-handle ablation switches (ngp_model_set_debug_flags / ngp_nerf_set_debug_flags): a handle that carries an override runs ITS calls under those switches whatever the
// process-wide ones say -- two trainers of one process can then differ (an A / B of two kernel variants side by side, a caller pinning the production path for its handle
// while a test harness toggles the global bits).  Applied for the duration of a call on the handle (training step, inference, rendering, grid update) and restored behind
// it; like every call on a handle it is thread-compatible, not thread-safe: calls on handles with DIFFERENT overrides must not overlap in time on different threads.
struct DebugOverride { bool on = false; uint32_t flags = 0, flags2 = 0; };
struct FlagScope {
	uint32_t saved = 0, saved2 = 0; bool active = false;
	explicit FlagScope(const DebugOverride& o) { if (o.on) { active = true; saved = g_debug_flags; saved2 = g_debug_flags2; g_debug_flags = o.flags | env_debug_or(); g_debug_flags2 = o.flags2 | env_debug2_or(); } }
	~FlagScope() { if (active) { g_debug_flags = saved; g_debug_flags2 = saved2; } }
};
// layout of the hashed levels' binned scatter (tuning / test hook): table entries per chunk (2^11 or 2^12), one block per chunk
// (split = 0) or per (chunk, feature pair) (split = 1), and a list-capacity override (0 = twice the mean; small values force the
// overflow path of k_grad_bin).  NGP_BIN_CHUNK_LOG2 / NGP_BIN_SPLIT / NGP_BIN_CAP in the environment set the defaults.
static uint32_t env_u32(const char* name, uint32_t dflt) { const char* e = getenv(name); return e ? (uint32_t)strtoul(e, nullptr, 0) : dflt; }
// chunk = 2^11 table entries since round 6 (64 KiB of accumulators per k_grad_accumulate block, two per CU): the kernels' own times equal those of 2^12 (one 128 KiB block per CU),
// but the step is 8-10 us shorter -- the next step's K1 (36 KiB of LDS per workgroup) then fits beside the accumulate blocks (profiles/r06_ab_bin_chunk_2048.txt; NGP_BIN_CHUNK_LOG2=12: rounds 2-5)
static uint32_t g_bin_chunk_log2 = env_u32("NGP_BIN_CHUNK_LOG2", 11) == 12 ? 12 : 11, g_bin_split = env_u32("NGP_BIN_SPLIT", 0) ? 1 : 0, g_bin_cap_override = env_u32("NGP_BIN_CAP", 0);
extern "C" int ngp_debug_set_bin_params(uint32_t chunk_log2, uint32_t split, uint32_t cap_override) {
	if (chunk_log2 != 11 && chunk_log2 != 12) { g_err = "ngp_debug_set_bin_params: chunk_log2 must be 11 or 12"; return 1; }
	g_bin_chunk_log2 = chunk_log2; g_bin_split = split ? 1 : 0; g_bin_cap_override = cap_override;
	return 0;
}
extern "C" int ngp_profile_count(void) { return P_COUNT; }
extern "C" const char* ngp_profile_name(int i) { return (i >= 0 && i < P_COUNT) ? kProfNames[i] : ""; }
extern "C" int ngp_profile_read(double* ms_sum, uint64_t* launches) {
	prof_collect();
	for (int i = 0; i < P_COUNT; ++i) { ms_sum[i] = g_prof_ms[i]; launches[i] = g_prof_n[i]; }
	return 0;
}

// Helper streams are created with the object that uses them, and one tiny command makes HIP create the stream's hardware queue right away.  Created lazily
// (inside the first training step) they used to come into existence AFTER a data-parallel caller's RCCL communicator, and in that order every kernel of the
// step ran 1.3 - 2x longer (+20 us even for one-wavefront kernels): 1.12 ms per step against 0.66 ms with the communicator created after the first steps
// (profiles/r03_dp_overhead.txt).  NGP_LAZY_STREAMS=1 restores the old order (diagnostic).
static bool lazy_streams() { return false; } // (round 3's diagnostic NGP_LAZY_STREAMS=1 -- helper streams created on first use, i.e. possibly behind a communicator -- is gone: profiles/r03_dp_overhead.txt)
static int g_helper_stream_device = -1; // the device the process-wide helper streams belong to (one process per GPU: SURVEY 8e)
static int create_helper_stream(hipStream_t* st, bool high_priority) {
	int dev = 0; HIPCHK(hipGetDevice(&dev));
	if (g_helper_stream_device < 0) g_helper_stream_device = dev;
	// one process drives one GPU: a model / trainer created with another device current would launch its backward pass on the first device's streams
	REQUIRE(dev == g_helper_stream_device, "libngp_hip's helper streams belong to the device that was current at ngp_init() / the first model; select the device (hipSetDevice / torch.cuda.set_device) before that and use one process per GPU");
	if (*st) return 0;
	if (high_priority) { int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi); HIPCHK(hipStreamCreateWithPriority(st, hipStreamNonBlocking, hi)); }
	else HIPCHK(hipStreamCreateWithFlags(st, hipStreamNonBlocking));
	static uint32_t* scratch = nullptr;
	if (!scratch) HIPCHK(hipMalloc((void**)&scratch, 256));
	HIPCHK(hipMemsetAsync(scratch, 0, 4, *st));
	HIPCHK(hipStreamSynchronize(*st));
	return 0;
}
// The helper streams are PROCESS-WIDE (one W stream, one for the dense-level ablation, one high-priority stream for the pre-launched K1) and live until the
// process ends: every model / trainer uses the same three, so a second trainer (bench.py's fox leg, a Testbed rebuilding its trainer after load_snapshot)
// neither adds hardware queues -- two trainers with their own streams ran the second one at 1.17 instead of 0.78 ms per step,
// profiles/r03_bench_n1_slow_box.json -- nor creates streams behind a communicator that exists by then.  ngp_init() creates them explicitly for hosts
// that set up communication before their first model.
// (Three helper streams + the caller's = the four hardware queues HIP maps streams onto by default.  A FIFTH stream shares a queue with another one: measured with one more helper stream
// merely created and used once per 16 steps -- every step 445 -> 660 us, all of it gaps between kernels whose own times did not change, profiles/r06_ab_grid_samples_ahead.txt.)
static hipStream_t g_side_stream = nullptr, g_side2_stream = nullptr, g_k1_stream = nullptr, g_comm_stream = nullptr;
static int ensure_helper_streams() { return create_helper_stream(&g_side_stream, false) || create_helper_stream(&g_k1_stream, true) || create_helper_stream(&g_comm_stream, true); }
extern "C" int ngp_init(void) {
	REQUIRE(ngp_device_available(), "no HIP device visible: libngp_hip has no CPU fallback");
	return ensure_helper_streams();
}
template <typename T> static int dev_alloc(T** p, size_t n) { HIPCHK(hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T))); return 0; }

// ------------------------------------------------------------------------------------------------
// config
// ------------------------------------------------------------------------------------------------
extern "C" int ngp_model_config_from_json(const char* json_host, uint32_t aabb_scale, uint32_t n_extra_dims, ngp_model_config* c) {
	mini_json::Value root;
	std::string err;
	if (!mini_json::parse(json_host, root, err)) return fail("config json: " + err);
	const auto& enc = root["encoding"];
	const auto& net = root["network"];
	const auto& rgb = root["rgb_network"];
	const auto& dir = root["dir_encoding"];
	std::string otype = enc.str("otype", "HashGrid");
	for (auto& ch : otype) ch = (char)tolower(ch);
	REQUIRE(otype == "hashgrid", "only the HashGrid encoding is implemented on this path (got '" + otype + "')");
	// reset_network, testbed.cu:4218-4255
	c->n_features_per_level = (uint32_t)enc.num("n_features_per_level", 2);
	if (enc.has("n_features") && enc.num("n_features", 0) > 0) c->n_levels = (uint32_t)enc.num("n_features", 0) / c->n_features_per_level;
	else c->n_levels = (uint32_t)enc.num("n_levels", 16);
	c->log2_hashmap_size = (uint32_t)enc.num("log2_hashmap_size", 15);
	c->base_resolution = (uint32_t)enc.num("base_resolution", 0);
	if (!c->base_resolution) c->base_resolution = 1u << (c->log2_hashmap_size / 3);
	c->per_level_scale = (float)enc.num("per_level_scale", 0.0);
	if (c->per_level_scale <= 0.0f && c->n_levels > 1) {
		const float desired_resolution = 2048.0f;
		c->per_level_scale = std::exp(std::log(desired_resolution * (float)aabb_scale / (float)c->base_resolution) / (c->n_levels - 1));
	}
	c->n_neurons = (uint32_t)net.num("n_neurons", 64);
	c->n_hidden_layers = (uint32_t)net.num("n_hidden_layers", 1);
	c->n_hidden_layers_rgb = (uint32_t)rgb.num("n_hidden_layers", 2);
	c->sh_degree = 4;
	if (dir.has("nested") && dir["nested"].size() > 0) c->sh_degree = (uint32_t)dir["nested"].at(0).num("degree", 4);
	else if (dir.has("degree")) c->sh_degree = (uint32_t)dir.num("degree", 4);
	c->n_extra_dims = n_extra_dims;
	// optimizer: Ema( ExponentialDecay( Adam ) ), configs/nerf/base.json:5-22; walk the nesting
	c->learning_rate = 1e-2f; c->beta1 = 0.9f; c->beta2 = 0.99f; c->epsilon = 1e-15f; c->l2_reg = 1e-6f;
	c->ema_decay = 0.0f; c->ema_full_precision = 0; c->decay_start = 0; c->decay_interval = 0; c->decay_base = 1.0f;
	const mini_json::Value* o = &root["optimizer"];
	for (int depth = 0; depth < 4 && o->is_object(); ++depth) {
		std::string t = o->str("otype", "");
		for (auto& ch : t) ch = (char)tolower(ch);
		if (t == "ema") { c->ema_decay = (float)o->num("decay", 0.99); c->ema_full_precision = o->boolean("full_precision", false) ? 1u : 0u; } // [tcnn EmaOptimizer::update_hyperparams]
		else if (t == "exponentialdecay") {
			c->decay_start = (uint32_t)o->num("decay_start", 10000); c->decay_interval = (uint32_t)o->num("decay_interval", 10000);
			c->decay_base = (float)o->num("decay_base", 0.33);
		} else if (t == "adam") {
			c->learning_rate = (float)o->num("learning_rate", 1e-3); c->beta1 = (float)o->num("beta1", 0.9); c->beta2 = (float)o->num("beta2", 0.999);
			c->epsilon = (float)o->num("epsilon", 1e-8); c->l2_reg = (float)o->num("l2_reg", 1e-8);
		}
		if (!o->has("nested")) break;
		o = &(*o)["nested"];
	}
	return 0;
}

// ------------------------------------------------------------------------------------------------
// model
// ------------------------------------------------------------------------------------------------
struct ngp_model {
	ngp_model_config cfg;
	GridMeta gm;
	GridMeta* gm_dev = nullptr;
	uint64_t n_params = 0, n_mlp = 0;
	float* master = nullptr; ngp_half* params = nullptr; ngp_half* params_inf = nullptr; ngp_half* grads = nullptr;
	float* adam_m = nullptr; float* adam_v = nullptr; float* ema = nullptr; uint16_t* adam_steps = nullptr;
	uint32_t* fw_perm = nullptr; uint32_t* bw_perm = nullptr;
	ngp_half* fw_frags = nullptr; ngp_half* bw_frags = nullptr; ngp_half* fw_frags_inf = nullptr;
	ngp_half* enc_stash = nullptr; size_t stash_halfs = 0;
	float* wgrad_partials = nullptr; uint32_t n_partials = 0;
	// binned scatter of the hashed levels: dL/d(enc) level-major, per-chunk record lists, list cursors
	void* denc_lv = nullptr; void* bin_vals = nullptr; void* bin_idxs = nullptr; uint32_t* bin_cursors = nullptr; uint32_t bin_n = 0, bin_cap = 0, bin_lists = 0;
	GradBinArgs bin_args{};
	// W (weight gradients, compute bound, 1 wave/SIMD) runs on a side stream next to the hashed levels' bin/accumulate kernels (memory/LDS bound)
	bool bin_dense = false; // the dense levels are scattered through the bin lists as well (production; see ngp_model_training_step)
	hipStream_t side = nullptr, side2 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr;
	// data-parallel step: events that mark the two gradient buckets final (recorded when record_bucket_events is set, see ngp_comm_*)
	bool record_bucket_events = false; hipEvent_t ev_hashed_ready = nullptr, ev_mlp_ready = nullptr;
	// sharded data-parallel step (round 5): k_grad_accumulate runs in two launches -- the first dp_split_ly listed levels (bucket A), then the rest (bucket B) -- and
	// ev_bucket_a marks bucket A's gradients final on the caller's stream, so that its reduce-scatter runs beside bucket B's accumulation (0 = one launch, no event)
	DebugOverride dbg; // per-handle ablation switches (ngp_model_set_debug_flags)
	uint32_t dp_split_level = 0; hipEvent_t ev_bucket_a = nullptr; bool dp_split_done = false /* the last training step accumulated in two launches and recorded ev_bucket_a */;
	uint32_t step = 0; float lr = 1e-2f;
	bool train_network = true, train_encoding = true;
	bool adam_fused_pending = false; uint64_t adam_sweep_end = 0, last_sweep_params = 0; // this step's k_grad_accumulate has already applied the optimizer to the hashed levels: the sweep covers [0, adam_sweep_end) only
	float* dextra_out = nullptr; // this training step also leaves dL/d(extra dims) per sample here (n x n_extra_dims floats; set by the callers that want it)
	bool grads_clean = true; // the hash-grid part of `grads` is all zero (after creation / after an optimizer sweep that zeroed it)
	// sharded data-parallel step: this rank's fp32 masters / Adam moments / step counters of the pieces OTHER ranks own are out of date (only the half parameters and the
	// inference copy travel each step).  Set by the sharded optimizer step, cleared by ngp_nerf_dp_gather_state (a collective); while set, the calls that would hand out or
	// continue from those arrays fail instead of returning stale numbers
	bool dp_state_stale = false;
};

// pcg32(initstate, initseq = 1) [tcnn pcg32.h]
static Rng make_rng(uint64_t seed) { ngp_pcg32 z; z.state = 0; z.inc = 3; Rng r(z); r.next_uint(); r.state += seed; r.next_uint(); return r; }
static ngp_pcg32 pod(const Rng& r) { ngp_pcg32 p; p.state = r.state; p.inc = r.inc; return p; }

static uint32_t next_multiple_u(uint32_t v, uint32_t d) { return ((v + d - 1) / d) * d; }

// [tcnn GridEncodingTemplated ctor] offset table for a Hash grid
static void build_grid_meta(const ngp_model_config& c, GridMeta& g) {
	memset(&g, 0, sizeof(g));
	g.n_levels = c.n_levels; g.F = c.n_features_per_level;
	const float l2 = std::log2(c.per_level_scale);
	uint32_t offset = 0;
	for (uint32_t i = 0; i < c.n_levels; ++i) {
		const float scale = std::exp2(i * l2) * c.base_resolution - 1.0f;
		const uint32_t res = (uint32_t)std::ceil(scale) + 1;
		const uint32_t max_params = 0xFFFFFFFFu / 2;
		uint32_t params_in_level = std::pow((float)res, 3.0f) > (float)max_params ? max_params : res * res * res;
		params_in_level = next_multiple_u(params_in_level, 8u);
		params_in_level = std::min(params_in_level, 1u << c.log2_hashmap_size);
		g.scale[i] = scale; g.resolution[i] = res; g.hashmap_size[i] = params_in_level; g.offset[i] = offset;
		offset += params_in_level;
	}
	g.offset[c.n_levels] = offset;
}

// MLP layers in parameter order (nerf_network.h:357-372): density L1, L2, then the colour network's L1, its NR - 1 layers of 64 x 64, and its output
// layer; row-major [out][in].  Fragment bases as in model_kernels.hip (FW_* / BW_*, fw_r3 / bw_r3).
// Extra dims (n_extra_dims > 0): the colour network's first layer is 64 x 48 (nerf_network.h:84-95: dir encoding = 16 SH + the extra dims, padded to 32 columns); its
// columns 32..47 live in fragments BEHIND the regular ones (xfw / xbw = their first index, 0 = none), so every other fragment index is the same in both kinds of model.
struct LayerDesc { uint32_t R, C, off, fw_base, bw_base, xfw = 0, xbw = 0; };
static std::vector<LayerDesc> nerf_layers(uint32_t n_rgb_hidden, bool extra = false) {
	const uint32_t xo = extra ? 64u * 16u : 0u;
	std::vector<LayerDesc> L = {{64, 32, 0, 0, 0}, {16, 64, 2048, 4, 4}, {64, extra ? 48u : 32u, 3072, 8, 6, extra ? n_fw_frags(n_rgb_hidden) : 0u, extra ? n_bw_frags(n_rgb_hidden) : 0u}};
	for (uint32_t k = 0; k + 1 < n_rgb_hidden; ++k) L.push_back({64, 64, 5120 + xo + 4096 * k, 12 + 8 * k, 10 + 8 * k});
	L.push_back({16, 64, 5120 + xo + 4096 * (n_rgb_hidden - 1), 12 + 8 * (n_rgb_hidden - 1), 10 + 8 * (n_rgb_hidden - 1)});
	return L;
}

// position of W[i][k] inside the forward / dgrad fragment buffers (see model_kernels.hip header):
//   k-index map of a fragment element: k(s, hi, j) = 16 s + 8 (j>>2) + 4 hi + (j&3)
static void build_perms(const std::vector<LayerDesc>& layers, uint32_t n_mlp, std::vector<uint32_t>& fw, std::vector<uint32_t>& bw) {
	fw.assign(n_mlp, 0xFFFFFFFFu); bw.assign(n_mlp, 0xFFFFFFFFu);
	for (const LayerDesc& L : layers) {
		for (uint32_t i = 0; i < L.R; ++i) for (uint32_t k = 0; k < L.C; ++k) {
			const uint32_t p = L.off + i * L.C + k;
			{ // forward: A[row = i][k]; fragment index = base + (i/32) * (C/16) + s
				const uint32_t mt = i / 32, s = k / 16, kk = k % 16;
				const uint32_t j = (kk / 8) * 4 + (kk % 4), hi = (kk % 8) / 4;
				const uint32_t frag = s >= 2 && L.xfw ? L.xfw + mt : L.fw_base + mt * ((L.xfw ? 32u : L.C) / 16) + s, lane = hi * 32 + (i % 32);
				fw[p] = (frag * 64 + lane) * 8 + j;
			}
			{ // dgrad: A[row = k][i]; fragment index = base + (k/32) * ceil(R/16) + s
				const uint32_t mt = k / 32, s = i / 16, ii = i % 16;
				const uint32_t j = (ii / 8) * 4 + (ii % 4), hi = (ii % 8) / 4;
				const uint32_t frag = mt >= 1 && L.xbw ? L.xbw + s : L.bw_base + mt * ((L.R + 15) / 16) + s, lane = hi * 32 + (k % 32);
				bw[p] = (frag * 64 + lane) * 8 + j;
			}
		}
	}
}

static int model_refresh_half(ngp_model* m, hipStream_t s); // master -> params/params_inf + fragments

extern "C" int ngp_model_create(const ngp_model_config* cfg, uint64_t seed, ngp_model** out) {
	REQUIRE(cfg && out, "ngp_model_create: null argument");
	REQUIRE(cfg->n_neurons == 64 && cfg->n_hidden_layers == 1 && cfg->n_hidden_layers_rgb >= 1 && cfg->n_hidden_layers_rgb <= 3,
		"the fused kernels cover the FullyFusedMLP topologies of configs/nerf/base*.json: 64 neurons, density network with 1 hidden layer, colour network with 1, 2 or 3");
	// the encoding feeds the 32-wide first layer directly in MFMA operand registers: L x F = 8 x 4 (configs/nerf/base.json) or 16 x 2 (the reference's 2022
	// base.json, notebooks/instant_ngp.ipynb:5838); any log2_hashmap_size (base_14, small, big: table sizes the record lists do not cover fall back to half atomics)
	REQUIRE((cfg->n_features_per_level == 4 && cfg->n_levels == 8) || (cfg->n_features_per_level == 2 && cfg->n_levels == 16), "the fused kernels take the hash grid as L = 8, F = 4 or L = 16, F = 2");
	REQUIRE(cfg->log2_hashmap_size >= 12 && cfg->log2_hashmap_size <= 24 && cfg->base_resolution >= 2 && cfg->per_level_scale >= 1.0f, "hash grid: log2_hashmap_size in [12, 24], base_resolution >= 2, per_level_scale >= 1");
	REQUIRE(cfg->sh_degree == 4, "only SphericalHarmonics degree 4 is implemented");
	REQUIRE(cfg->n_extra_dims <= 16, "at most 16 extra dims (the dir encoding's Identity part fills one 16-wide k-step of the colour network's input)");
	REQUIRE(cfg->n_extra_dims == 0 || (cfg->n_features_per_level == 4 && cfg->n_hidden_layers_rgb == 2), "extra dims are implemented for configs/nerf/base.json's shape: L = 8, F = 4, two hidden colour layers");
	REQUIRE(ngp_device_available(), "no HIP device visible: libngp_hip has no CPU fallback");
	ngp_model* m = new ngp_model();
	m->cfg = *cfg;
	build_grid_meta(*cfg, m->gm);
	const bool extra = cfg->n_extra_dims > 0;
	const uint32_t nr = cfg->n_hidden_layers_rgb, n_fw_halfs = (n_fw_frags(nr) + (extra ? 2u : 0u)) * FRAG_HALFS, n_bw_halfs = (n_bw_frags(nr) + (extra ? 4u : 0u)) * FRAG_HALFS;
	const std::vector<LayerDesc> layers = nerf_layers(nr, extra);
	m->n_mlp = n_mlp_params(nr) + (extra ? 64u * 16u : 0u);
	m->n_params = m->n_mlp + (uint64_t)m->gm.offset[cfg->n_levels] * cfg->n_features_per_level;
	m->lr = cfg->learning_rate;
	const uint64_t P = m->n_params;
	if (dev_alloc(&m->gm_dev, 1) || dev_alloc(&m->master, P) || dev_alloc(&m->params, P) || dev_alloc(&m->params_inf, P) || dev_alloc(&m->grads, P) ||
		dev_alloc(&m->adam_m, P) || dev_alloc(&m->adam_v, P) || dev_alloc(&m->ema, P) || dev_alloc(&m->adam_steps, P) ||
		dev_alloc(&m->fw_perm, m->n_mlp) || dev_alloc(&m->bw_perm, m->n_mlp) || dev_alloc(&m->fw_frags, n_fw_halfs) ||
		dev_alloc(&m->bw_frags, n_bw_halfs) || dev_alloc(&m->fw_frags_inf, n_fw_halfs)) { delete m; return 1; }
	HIPCHK(hipMemcpy(m->gm_dev, &m->gm, sizeof(GridMeta), hipMemcpyHostToDevice));
	HIPCHK(hipMemset(m->grads, 0, P * 2)); HIPCHK(hipMemset(m->adam_m, 0, P * 4)); HIPCHK(hipMemset(m->adam_v, 0, P * 4));
	HIPCHK(hipMemset(m->ema, 0, P * 4)); HIPCHK(hipMemset(m->adam_steps, 0, P * 2));
	HIPCHK(hipMemset(m->fw_frags, 0, n_fw_halfs * 2)); HIPCHK(hipMemset(m->bw_frags, 0, n_bw_halfs * 2));
	HIPCHK(hipMemset(m->fw_frags_inf, 0, n_fw_halfs * 2));
	std::vector<uint32_t> fwp, bwp;
	build_perms(layers, (uint32_t)m->n_mlp, fwp, bwp);
	HIPCHK(hipMemcpy(m->fw_perm, fwp.data(), fwp.size() * 4, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(m->bw_perm, bwp.data(), bwp.size() * 4, hipMemcpyHostToDevice));
	m->n_partials = wgrad_n_partials();
	if (dev_alloc(&m->wgrad_partials, (size_t)m->n_partials * (8 + 4 * (nr - 1) + (extra ? 2 : 0)) * 16 * 64)) { delete m; return 1; }
	// Trainer::initialize_params: pcg32{seed}; Xavier-uniform matrices, U(-1e-4, 1e-4) grid (element j <- draw j)
	std::vector<float> init(P);
	Rng rnd = make_rng(seed);
	size_t p = 0;
	for (const LayerDesc& L : layers) {
		const float scale = std::sqrt(6.0f / (float)(L.R + L.C));
		for (uint32_t i = 0; i < L.R * L.C; ++i) init[p++] = rnd.next_float() * 2.0f * scale - scale;
	}
	for (; p < P; ++p) init[p] = rnd.next_float() * (1e-4f - (-1e-4f)) + (-1e-4f);
	HIPCHK(hipMemcpy(m->master, init.data(), P * 4, hipMemcpyHostToDevice));
	if (model_refresh_half(m, nullptr)) { delete m; return 1; }
	HIPCHK(hipDeviceSynchronize());
	if (!lazy_streams()) { if (ensure_helper_streams()) { delete m; return 1; } m->side = g_side_stream; } // W's stream (see create_helper_stream)
	*out = m;
	return 0;
}

extern "C" void ngp_model_destroy(ngp_model* m) {
	if (!m) return;
	void* ptrs[] = {m->gm_dev, m->master, m->params, m->params_inf, m->grads, m->adam_m, m->adam_v, m->ema, m->adam_steps, m->fw_perm, m->bw_perm,
		m->fw_frags, m->bw_frags, m->fw_frags_inf, m->enc_stash, m->wgrad_partials, m->denc_lv, m->bin_vals, m->bin_idxs, m->bin_cursors};
	for (void* p : ptrs) if (p) (void)hipFree(p);
	// (the helper streams are process-wide: not destroyed with the model)
	if (m->ev_join2) (void)hipEventDestroy(m->ev_join2);
	if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
	if (m->ev_join) (void)hipEventDestroy(m->ev_join);
	if (m->ev_bucket_a) (void)hipEventDestroy(m->ev_bucket_a);
	if (m->ev_hashed_ready) (void)hipEventDestroy(m->ev_hashed_ready);
	if (m->ev_mlp_ready) (void)hipEventDestroy(m->ev_mlp_ready);
	delete m;
}

__global__ void k_master_to_half(const float* __restrict__ master, __half* __restrict__ params, __half* __restrict__ params_inf, uint64_t n) {
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const __half h = __float2half(master[i]);
	params[i] = h; params_inf[i] = h;
}
static int model_refresh_half(ngp_model* m, hipStream_t s) {
	hipLaunchKernelGGL(k_master_to_half, dim3((uint32_t)((m->n_params + 255) / 256)), dim3(256), 0, s, m->master, (__half*)m->params, (__half*)m->params_inf, m->n_params);
	launch_build_frags(s, m->params, (uint32_t)m->n_mlp, m->fw_perm, m->bw_perm, m->fw_frags, m->bw_frags);
	launch_build_frags(s, m->params_inf, (uint32_t)m->n_mlp, m->fw_perm, m->bw_perm, m->fw_frags_inf, nullptr);
	HIPCHK(hipGetLastError());
	return 0;
}

extern "C" int ngp_model_n_params(const ngp_model* m, uint64_t* n, uint64_t* n_mlp) { if (n) *n = m->n_params; if (n_mlp) *n_mlp = m->n_mlp; return 0; }
extern "C" int ngp_model_param_ptrs(ngp_model* m, float** master, ngp_half** params, ngp_half** inf, ngp_half** grads) {
	if (master) *master = m->master; if (params) *params = m->params; if (inf) *inf = m->params_inf; if (grads) *grads = m->grads; return 0;
}
extern "C" int ngp_model_grid_layout(const ngp_model* m, uint32_t* offsets, uint32_t* resolutions, float* scales) {
	for (uint32_t i = 0; i <= m->gm.n_levels; ++i) offsets[i] = m->gm.offset[i];
	for (uint32_t i = 0; i < m->gm.n_levels; ++i) { resolutions[i] = m->gm.resolution[i]; scales[i] = m->gm.scale[i]; }
	return 0;
}
extern "C" int ngp_model_set_params_host(ngp_model* m, const float* p, uint64_t n) {
	REQUIRE(n == m->n_params, "set_params: size mismatch");
	HIPCHK(hipMemcpy(m->master, p, n * 4, hipMemcpyHostToDevice));
	if (model_refresh_half(m, nullptr)) return 1;
	HIPCHK(hipDeviceSynchronize());
	return 0;
}
static const char* kStaleMsg = "the sharded data-parallel step leaves this rank's fp32 masters / Adam state of the other ranks' pieces out of date: call ngp_nerf_dp_gather_state (Testbed.dp_gather_state) on EVERY rank first";
extern "C" int ngp_model_get_params_host(ngp_model* m, float* p, uint64_t n) {
	REQUIRE(n == m->n_params, "get_params: size mismatch");
	REQUIRE(!m->dp_state_stale, kStaleMsg);
	HIPCHK(hipMemcpy(p, m->master, n * 4, hipMemcpyDeviceToHost));
	return 0;
}

static ModelPtrs model_ptrs(const ngp_model* m, bool inference) {
	ModelPtrs mp;
	mp.grid = (inference ? m->params_inf : m->params) + m->n_mlp;
	mp.fw_frags = inference ? m->fw_frags_inf : m->fw_frags;
	mp.bw_frags = m->bw_frags;
	mp.n_rgb_hidden = m->cfg.n_hidden_layers_rgb;
	mp.n_extra = m->cfg.n_extra_dims;
	return mp;
}

extern "C" int ngp_model_inference(ngp_model* m, void* stream, const float* in, uint32_t in_stride, uint32_t n_max, const uint32_t* n_ptr,
		ngp_half* out, uint32_t out_stride, int use_inference_params) {
	FlagScope flag_scope_(m->dbg);
	REQUIRE(in_stride >= 7 && out_stride >= 4 && out_stride % 4 == 0, "inference: in_stride >= 7, out_stride a multiple of 4 halfs");
	{ ProfScope ps(P_K2_INFERENCE, (hipStream_t)stream);
	  launch_inference((hipStream_t)stream, m->gm_dev, model_ptrs(m, use_inference_params != 0), in, in_stride, n_max, n_ptr, out, out_stride, false, 4, m->gm.F); }
	HIPCHK(hipGetLastError());
	return 0;
}
extern "C" int ngp_model_density(ngp_model* m, void* stream, const float* pos, uint32_t pos_stride, uint32_t n, ngp_half* out, uint32_t out_stride, int use_inference_params) {
	REQUIRE(pos_stride >= 3 && out_stride >= 1, "density: bad strides");
	{ ProfScope ps(P_GRID_DENSITY, (hipStream_t)stream);
	  launch_inference((hipStream_t)stream, m->gm_dev, model_ptrs(m, use_inference_params != 0), pos, pos_stride, n, nullptr, out, out_stride, true, 0, m->gm.F); }
	HIPCHK(hipGetLastError());
	return 0;
}
// test hook: grid encoding only (natural feature order), not part of the reference API surface
extern "C" int ngp_model_encode(ngp_model* m, void* stream, const float* pos, uint32_t pos_stride, uint32_t n, ngp_half* out32) {
	launch_encode_only((hipStream_t)stream, m->gm_dev, m->params + m->n_mlp, pos, pos_stride, n, out32, m->gm.F);
	HIPCHK(hipGetLastError());
	return 0;
}

static AdamArgs make_adam_args(const ngp_model* m, float loss_scale, uint32_t step);
static int model_training_step_impl(ngp_model* m, void* stream, const float* in, uint32_t in_stride, uint32_t n, const ngp_half* dL_dy, uint32_t dy_stride, const EncStashIn* stash_in, float fuse_optimizer_loss_scale = 0.f);
extern "C" int ngp_model_training_step(ngp_model* m, void* stream, const float* in, uint32_t in_stride, uint32_t n, const ngp_half* dL_dy, uint32_t dy_stride) {
	m->dextra_out = nullptr;
	return model_training_step_impl(m, stream, in, in_stride, n, dL_dy, dy_stride, nullptr);
}
extern "C" int ngp_model_training_step_extra(ngp_model* m, void* stream, const float* in, uint32_t in_stride, uint32_t n, const ngp_half* dL_dy, uint32_t dy_stride, float* dL_dextra) {
	REQUIRE(m->cfg.n_extra_dims > 0 || !dL_dextra, "training_step_extra: the model has no extra dims");
	m->dextra_out = dL_dextra;
	const int r = model_training_step_impl(m, stream, in, in_stride, n, dL_dy, dy_stride, nullptr);
	m->dextra_out = nullptr;
	return r;
}
// fuse_optimizer_loss_scale > 0: the caller runs ngp_model_optimizer_step(m, stream, that loss scale) next, with nothing in between that looks at the hash-grid gradients
// (ngp_nerf_train on one GPU): k_grad_accumulate then applies the optimizer to the hashed levels itself (GradBinArgs::fuse_adam).
static int model_training_step_impl(ngp_model* m, void* stream, const float* in, uint32_t in_stride, uint32_t n, const ngp_half* dL_dy, uint32_t dy_stride, const EncStashIn* stash_in, float fuse_optimizer_loss_scale) {
	FlagScope flag_scope_(m->dbg);
	REQUIRE(in_stride >= 7 + m->cfg.n_extra_dims && dy_stride >= 4 && dy_stride % 4 == 0, "training_step: in_stride >= 7 + n_extra_dims, dy_stride a multiple of 4 halfs");
	if (m->cfg.n_extra_dims) stash_in = nullptr; // (the lazy K2 that leaves the encodings behind has no extra-dims instance)
	hipStream_t s = (hipStream_t)stream;
	const size_t need = (size_t)((n + 31) / 32) * 2 * 64 * 8; // halfs
	if (need > m->stash_halfs) {
		HIPCHK(hipStreamSynchronize(s));
		if (m->enc_stash) HIPCHK(hipFree(m->enc_stash));
		m->enc_stash = nullptr; m->stash_halfs = 0;
		if (dev_alloc(&m->enc_stash, need)) return 1;
		m->stash_halfs = need;
	}
	// binned scatter: every hashed level must have a power-of-two table of 2^chunk_log2 .. 2^19 entries (base.json: 2^19)
	GradBinArgs& ba = m->bin_args;
	ba.n_hashed = 0; ba.max_chunks = 0; ba.chunk_log2 = g_bin_chunk_log2; ba.split = g_bin_split && m->gm.F == 4; ba.merge_runs = !(g_debug_flags & DBG_BIN_NO_HASHED_MERGE); ba.no_dense_merge = (g_debug_flags & DBG_BIN_NO_DENSE_MERGE) != 0;
	if (!(g_debug_flags & DBG_T1_NO_BINNING)) {
		bool ok = true;
		// The dense levels go through the lists as well (entries interleaved over all 2^(19 - chunk_log2) chunks, see k_grad_bin): T1 issues no
		// atomics at all -- the memory side retires only ~14 G four-byte atomic operations per second, and the dense levels needed 1.9 M of them
		// per step (~100 us of T1).  Needs the one-block-per-chunk layout; DBG_T1_DENSE_ATOMICS restores the atomics.
		// (and every hashed level must have the full 2^19-entry table, base.json's size: the dense levels always spread over 2^(19 - chunk_log2)
		// lists and all lists share one capacity, so smaller tables would overflow theirs)
		bool dense_too = !(g_debug_flags & DBG_T1_DENSE_ATOMICS) && !ba.split;
		for (uint32_t l = 0; l < m->gm.n_levels; ++l) {
			const uint64_t res = m->gm.resolution[l], hs = m->gm.hashmap_size[l];
			if (res * res * res > hs && hs != (1ull << GRAD_BIN_MAX_TABLE_LOG2)) dense_too = false;
		}
		uint32_t n_dense = 0;
		for (uint32_t l = 0; l < m->gm.n_levels; ++l) {
			const uint64_t res = m->gm.resolution[l], hs = m->gm.hashmap_size[l];
			if (res * res * res <= hs) {
				if (!dense_too) continue;
				if (hs > (1u << GRAD_BIN_MAX_TABLE_LOG2)) { ok = false; break; }
				ba.levels[ba.n_hashed++] = l; ++n_dense;
				ba.max_chunks = std::max<uint32_t>(ba.max_chunks, (1u << GRAD_BIN_MAX_TABLE_LOG2) >> ba.chunk_log2);
				continue;
			}
			if ((hs & (hs - 1)) || hs < (1u << ba.chunk_log2) || hs > (1u << GRAD_BIN_MAX_TABLE_LOG2)) { ok = false; break; }
			ba.levels[ba.n_hashed++] = l;
			ba.max_chunks = std::max<uint32_t>(ba.max_chunks, (uint32_t)(hs >> ba.chunk_log2));
		}
		if (!ok) ba.n_hashed = 0;
		m->bin_dense = ok && n_dense > 0;
	}
	// lists: n_hashed x max_chunks of `cap` records (8-byte values + 2-byte local indices); capacity = twice the mean number of
	// records per chunk.  Re-allocated when the batch grows or the layout (chunk size, capacity override) changes.
	const uint32_t cap_want = !ba.n_hashed ? 0u : g_bin_cap_override ? g_bin_cap_override
		: std::max<uint32_t>(2048u, (uint32_t)((((uint64_t)n * 8 * 2) / ba.max_chunks + 1023) / 1024 * 1024));
	if (ba.n_hashed && (n > m->bin_n || cap_want != m->bin_cap || ba.n_hashed * ba.max_chunks != m->bin_lists)) {
		HIPCHK(hipStreamSynchronize(s));
		for (void* p : {m->denc_lv, m->bin_vals, m->bin_idxs, (void*)m->bin_cursors}) if (p) HIPCHK(hipFree(p));
		const uint32_t n_alloc = std::max(n, m->bin_n);
		m->denc_lv = m->bin_vals = m->bin_idxs = nullptr; m->bin_cursors = nullptr; m->bin_n = 0;
		const size_t n_lists = (size_t)ba.n_hashed * ba.max_chunks;
		HIPCHK(hipMalloc(&m->denc_lv, (size_t)m->gm.n_levels * n_alloc * m->gm.F * 2));
		HIPCHK(hipMalloc(&m->bin_vals, n_lists * cap_want * m->gm.F * 2));
		HIPCHK(hipMalloc(&m->bin_idxs, n_lists * cap_want * 2));
		HIPCHK(hipMalloc((void**)&m->bin_cursors, n_lists * 2 * 4)); // cursors + the split variant's arrival counters
		HIPCHK(hipMemsetAsync(m->bin_cursors, 0, n_lists * 2 * 4, s));
		m->bin_n = n_alloc; m->bin_cap = cap_want; m->bin_lists = (uint32_t)n_lists;
	}
	// GradientMode::Overwrite: clear the hash-grid gradient table (the MLP part is fully rewritten)
	if (!m->grads_clean) { ProfScope ps(P_GRAD_MEMSET, s); HIPCHK(hipMemsetAsync(m->grads + m->n_mlp, 0, (m->n_params - m->n_mlp) * 2, s)); }
	m->grads_clean = false;
	// dense levels (ablation DBG_T1_DENSE_EXTERNAL): T1 leaves their dL/d(enc) in denc_lv as well and k_grad_dense issues the atomics beside the kernels below
	GradDenseArgs da;
	da.n_levels = 0;
	if (ba.n_hashed && m->gm.F == 4 && (g_debug_flags & DBG_T1_DENSE_EXTERNAL) && !m->bin_dense && !(g_debug_flags & DBG_T1_NO_SCATTER))
		for (uint32_t l = 0; l < m->gm.n_levels; ++l) { const uint64_t res = m->gm.resolution[l]; if (res * res * res <= m->gm.hashmap_size[l]) da.levels[da.n_levels++] = l; }
	// T1 + W in one kernel (round 5): base.json's shape with K2's encoding stash and every level's dL/d(enc) going through the lists -- the configuration whose T1 is the
	// STASH instance of k_train_fwd_bwd and whose W is k_wgrad2.  Everything else (and the ablation DBG2_NO_FUSED_T1W) runs the two kernels.
	const bool fused_t1w = stash_in && stash_in->enc && m->gm.F == 4 && m->cfg.n_hidden_layers_rgb == 2 && !m->cfg.n_extra_dims && ba.n_hashed && (da.n_levels || m->bin_dense) && !m->dextra_out
		&& !(g_debug_flags & (DBG_T1_NO_K2_STASH | DBG_T1_OCC2 | DBG_T1_NO_SCATTER | DBG_W_SINGLE_ROLE)) && !(g_debug_flags2 & DBG2_NO_FUSED_T1W);
	if (fused_t1w) { ProfScope ps(P_TRAIN_FUSED, s); launch_train_fused(s, model_ptrs(m, false), in, in_stride, n, dL_dy, dy_stride, *stash_in, m->denc_lv, m->bin_n, m->wgrad_partials, m->n_partials); }
	else { ProfScope ps(P_T1_FWD_BWD_SCATTER, s); launch_train_fwd_bwd(s, m->gm_dev, model_ptrs(m, false), in, in_stride, n, dL_dy, dy_stride, m->grads + m->n_mlp, m->enc_stash,
		g_debug_flags | ((da.n_levels || m->bin_dense) ? T1_DENSE_EXTERNAL : 0u), ba.n_hashed ? m->denc_lv : nullptr, m->bin_n, m->gm.F, stash_in, m->dextra_out); }
	// fork: per-kernel profiling keeps everything on one stream so that the HIP-event times are those of isolated kernels
	const bool overlap = ba.n_hashed && !g_prof_on && !(g_debug_flags & DBG_NO_STREAM_OVERLAP);
	hipStream_t sw = s;
	if (overlap) {
		if (!m->side) { if (create_helper_stream(&g_side_stream, false)) return 1; m->side = g_side_stream; }
		if (!m->ev_fork) { HIPCHK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming)); }
		HIPCHK(hipEventRecord(m->ev_fork, s)); HIPCHK(hipStreamWaitEvent(m->side, m->ev_fork, 0));
		sw = m->side;
	}
	if (da.n_levels) {
		da.gm = m->gm_dev; da.in = in; da.in_stride = in_stride; da.n = n; da.denc_lv = (const uint2*)m->denc_lv; da.denc_cap = m->bin_n;
		da.merge_runs = !(g_debug_flags & DBG_T1_NO_MERGE); da.grid_grad_ = m->grads + m->n_mlp;
		if (overlap) {
			if (!m->side2) { if (create_helper_stream(&g_side2_stream, false)) return 1; m->side2 = g_side2_stream; }
			if (!m->ev_join2) HIPCHK(hipEventCreateWithFlags(&m->ev_join2, hipEventDisableTiming));
			HIPCHK(hipStreamWaitEvent(m->side2, m->ev_fork, 0));
			launch_grad_dense(m->side2, da);
			HIPCHK(hipEventRecord(m->ev_join2, m->side2));
		}
	}
	if (!fused_t1w) { ProfScope ps(P_W_WGRAD, sw); launch_wgrad(sw, model_ptrs(m, false), in, in_stride, n, dL_dy, dy_stride, m->enc_stash, m->wgrad_partials, m->n_partials); }
	{ ProfScope ps(P_WGRAD_REDUCE, sw); launch_wgrad_reduce(sw, m->wgrad_partials, m->n_partials, m->grads, m->cfg.n_hidden_layers_rgb, m->cfg.n_extra_dims); }
	if (ba.n_hashed) {
		ProfScope ps(P_GRAD_BIN, s);
		ba.gm = m->gm_dev; ba.in = in; ba.in_stride = in_stride; ba.n = n; ba.denc_lv = m->denc_lv; ba.denc_cap = m->bin_n; ba.cap = m->bin_cap; ba.n_features = m->gm.F;
		ba.vals = m->bin_vals; ba.idxs = (uint16_t*)m->bin_idxs; ba.cursors = m->bin_cursors; ba.cursor_done = m->bin_cursors + m->bin_lists; ba.grid_grad_ = m->grads + m->n_mlp;
		ba.fuse_adam = 0;
		static const bool no_fuse = getenv("NGP_NO_FUSED_ADAM") && atoi(getenv("NGP_NO_FUSED_ADAM")) != 0; // ablation: separate sweep over all parameters (rounds 1-3)
		if (fuse_optimizer_loss_scale > 0.f && !no_fuse && m->gm.F == 4 && !ba.split && m->bin_dense) {
			// parameter order = MLP, then the levels coarse -> fine: the dense levels (and the MLP) stay with the sweep, everything from the first hashed level on is done here
			uint64_t first_hashed = m->n_params; bool ordered = true;
			for (uint32_t l = 0; l < m->gm.n_levels; ++l) {
				const uint64_t res = m->gm.resolution[l], begin = m->n_mlp + (uint64_t)m->gm.offset[l] * m->gm.F;
				if (res * res * res > m->gm.hashmap_size[l]) first_hashed = std::min(first_hashed, begin); else if (begin >= first_hashed) ordered = false;
			}
			// the sweep's own precondition, checked BEFORE the epilogue may touch a parameter (a refused optimizer step must not leave the hashed levels updated)
			const bool betas_ok = 65535.0f * std::log(m->cfg.beta1) < -18.f && 65535.0f * std::log(m->cfg.beta2) < -18.f;
			if (ordered && betas_ok && first_hashed < m->n_params && first_hashed % 4 == 0) {
				ba.fuse_adam = 1; ba.adam = make_adam_args(m, fuse_optimizer_loss_scale, m->step + 1);
				m->adam_fused_pending = true; m->adam_sweep_end = first_hashed;
			}
		}
		uint32_t split_ly = 0; // listed levels below the bucket boundary
		if (m->dp_split_level) for (uint32_t k = 0; k < ba.n_hashed; ++k) if (ba.levels[k] < m->dp_split_level) ++split_ly;
		m->dp_split_done = false;
		if (split_ly > 0 && split_ly < ba.n_hashed && !ba.fuse_adam && !g_prof_on) {
			if (!m->ev_bucket_a) HIPCHK(hipEventCreateWithFlags(&m->ev_bucket_a, hipEventDisableTiming));
			launch_grad_bin(s, ba, 1u);
			ba.acc_ly_begin = 0; ba.acc_ly_count = split_ly; launch_grad_bin(s, ba, 2u);
			HIPCHK(hipEventRecord(m->ev_bucket_a, s));
			ba.acc_ly_begin = split_ly; ba.acc_ly_count = ba.n_hashed - split_ly; launch_grad_bin(s, ba, 2u);
			ba.acc_ly_begin = ba.acc_ly_count = 0;
			m->dp_split_done = true;
		} else launch_grad_bin(s, ba);
		if (da.n_levels && !overlap) launch_grad_dense(s, da); // profiling / single-stream mode: part of the same scope (one unit of algorithmic work)
	}
	if (da.n_levels && overlap) HIPCHK(hipStreamWaitEvent(sw, m->ev_join2, 0)); // bucket A (MLP + dense levels) is final behind W's stream from here on
	if (m->record_bucket_events) { // bucket B (hashed levels) is final behind k_grad_accumulate, bucket A (MLP + dense levels) behind T1 and k_wgrad_reduce
		if (!m->ev_hashed_ready) { HIPCHK(hipEventCreateWithFlags(&m->ev_hashed_ready, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&m->ev_mlp_ready, hipEventDisableTiming)); }
		HIPCHK(hipEventRecord(m->ev_hashed_ready, s)); HIPCHK(hipEventRecord(m->ev_mlp_ready, sw));
	}
	if (overlap) { HIPCHK(hipEventRecord(m->ev_join, sw)); HIPCHK(hipStreamWaitEvent(s, m->ev_join, 0)); }
	HIPCHK(hipGetLastError());
	return 0;
}

// ------------------------------------------------------------------------------------------------
// encoding + MLP model of the image / SDF primitives (forward only): tcnn::NetworkWithInputEncoding, testbed.cu:4354-4363
// ------------------------------------------------------------------------------------------------
struct ngp_encmlp {
	ngp_encmlp_config cfg{};
	ngp_optimizer_config opt{};
	GridMeta gm{}; GridMeta* gm_dev = nullptr;
	uint64_t n_params = 0, n_mlp = 0;
	// Trainer state on the device, same layout conventions as ngp_model: MLP (row-major [out][in] per layer) then the grid [tcnn]
	float* master = nullptr; ngp_half* params = nullptr; ngp_half* params_inf = nullptr; ngp_half* grads = nullptr;
	float* adam_m = nullptr; float* adam_v = nullptr; float* ema = nullptr; uint16_t* adam_steps = nullptr;
	uint32_t* fw_perm = nullptr; uint32_t* bw_perm = nullptr;
	ngp_half* fw_frags = nullptr; ngp_half* bw_frags = nullptr; ngp_half* fw_frags_inf = nullptr; // only the FW_R* / BW_R* slots are used
	ngp_half* enc_stash = nullptr; void* dy_stash = nullptr; uint32_t stash_n = 0;
	float* wgrad_partials = nullptr; uint32_t n_partials = 0;
	uint32_t step = 0; float lr = 1e-2f;
	bool train_network = true, train_encoding = true;
	bool grads_clean = false; // the grid part of `grads` is all zero (the optimizer sweep zeroed what it consumed): the next training step needs no memset launch
	// scatter through the record lists (k_grad_bin / k_grad_accumulate, as in ngp_model): level-major dL/d(enc) + one list per (level, 4096-entry chunk)
	void* denc_lv = nullptr; void* bin_vals = nullptr; void* bin_idxs = nullptr; uint32_t* bin_cursors = nullptr; uint32_t bin_n = 0, bin_cap = 0, bin_lists = 0;
};
static void build_grid_meta_nd(const ngp_encmlp_config& c, GridMeta& g) {
	memset(&g, 0, sizeof(g));
	g.n_levels = c.n_levels; g.F = c.n_features_per_level;
	const float l2 = std::log2(c.per_level_scale);
	uint32_t offset = 0;
	for (uint32_t i = 0; i < c.n_levels; ++i) {
		const float scale = std::exp2(i * l2) * c.base_resolution - 1.0f;
		const uint32_t res = (uint32_t)std::ceil(scale) + 1;
		const uint32_t max_params = 0xFFFFFFFFu / 2;
		uint32_t params_in_level = std::pow((float)res, (float)c.n_pos_dims) > (float)max_params ? max_params : (uint32_t)std::pow((float)res, (float)c.n_pos_dims);
		params_in_level = next_multiple_u(params_in_level, 8u);
		params_in_level = std::min(params_in_level, 1u << c.log2_hashmap_size);
		g.scale[i] = scale; g.resolution[i] = res; g.hashmap_size[i] = params_in_level; g.offset[i] = offset;
		offset += params_in_level;
	}
	g.offset[c.n_levels] = offset;
}
// the three layers of the 32 -> 64 -> 64 -> 16 network sit in the fragment slots of the NeRF colour network (FW_R* / BW_R*)
static const LayerDesc kEncLayers[3] = {{64, 32, 0, 8, 6}, {64, 64, 2048, 12, 10}, {16, 64, 6144, 20, 18}};
static void build_enc_perms(std::vector<uint32_t>& fw, std::vector<uint32_t>& bw) {
	fw.assign(7168, 0xFFFFFFFFu); bw.assign(7168, 0xFFFFFFFFu);
	for (const LayerDesc& L : kEncLayers)
		for (uint32_t i = 0; i < L.R; ++i) for (uint32_t k = 0; k < L.C; ++k) {
			const uint32_t p = L.off + i * L.C + k;
			{ const uint32_t mt = i / 32, s = k / 16, kk = k % 16, j = (kk / 8) * 4 + (kk % 4), hi = (kk % 8) / 4;
			  fw[p] = ((L.fw_base + mt * (L.C / 16) + s) * 64 + hi * 32 + (i % 32)) * 8 + j; }
			{ const uint32_t mt = k / 32, s = i / 16, ii = i % 16, j = (ii / 8) * 4 + (ii % 4), hi = (ii % 8) / 4;
			  bw[p] = ((L.bw_base + mt * ((L.R + 15) / 16) + s) * 64 + hi * 32 + (k % 32)) * 8 + j; }
		}
}
static int encmlp_refresh_half(ngp_encmlp* m, hipStream_t s) { // master -> params / params_inf + fragments
	hipLaunchKernelGGL(k_master_to_half, dim3((uint32_t)((m->n_params + 255) / 256)), dim3(256), 0, s, m->master, (__half*)m->params, (__half*)m->params_inf, m->n_params);
	launch_build_frags(s, m->params, (uint32_t)m->n_mlp, m->fw_perm, m->bw_perm, m->fw_frags, m->bw_frags);
	launch_build_frags(s, m->params_inf, (uint32_t)m->n_mlp, m->fw_perm, m->bw_perm, m->fw_frags_inf, nullptr);
	HIPCHK(hipGetLastError());
	return 0;
}
extern "C" int ngp_encmlp_create(const ngp_encmlp_config* cfg, uint64_t seed, ngp_encmlp** out) {
	REQUIRE(cfg && out, "ngp_encmlp_create: null argument");
	REQUIRE(cfg->n_pos_dims == 2 || cfg->n_pos_dims == 3, "encmlp: n_pos_dims must be 2 (image) or 3 (SDF)");
	REQUIRE(cfg->n_levels == 16 && cfg->n_features_per_level == 2, "encmlp: the fused kernel is specialised for L = 16, F = 2 (configs/image|sdf/base.json)");
	REQUIRE(cfg->n_neurons == 64 && cfg->n_hidden_layers == 2 && cfg->n_output_dims >= 1 && cfg->n_output_dims <= 16, "encmlp: 64 neurons, 2 hidden layers, 1..16 outputs");
	REQUIRE(ngp_device_available(), "no HIP device visible: libngp_hip has no CPU fallback");
	ngp_encmlp* m = new ngp_encmlp();
	m->cfg = *cfg;
	// configs/image/base.json, configs/sdf/base.json: ExponentialDecay(Adam), no EMA
	m->opt.learning_rate = 1e-2f; m->opt.beta1 = 0.9f; m->opt.beta2 = 0.99f; m->opt.epsilon = 1e-15f; m->opt.l2_reg = 1e-6f;
	m->opt.ema_decay = 0.f; m->opt.decay_start = 20000; m->opt.decay_interval = 10000; m->opt.decay_base = 0.33f;
	m->lr = m->opt.learning_rate;
	build_grid_meta_nd(*cfg, m->gm);
	m->n_mlp = 64 * 32 + 64 * 64 + 16 * 64;
	m->n_params = m->n_mlp + (uint64_t)m->gm.offset[cfg->n_levels] * cfg->n_features_per_level;
	const uint64_t P = m->n_params;
	if (dev_alloc(&m->gm_dev, 1) || dev_alloc(&m->master, P) || dev_alloc(&m->params, P) || dev_alloc(&m->params_inf, P) || dev_alloc(&m->grads, P) ||
		dev_alloc(&m->adam_m, P) || dev_alloc(&m->adam_v, P) || dev_alloc(&m->ema, P) || dev_alloc(&m->adam_steps, P) ||
		dev_alloc(&m->fw_perm, 7168) || dev_alloc(&m->bw_perm, 7168) || dev_alloc(&m->fw_frags, N_FW_FRAGS * FRAG_HALFS) ||
		dev_alloc(&m->bw_frags, N_BW_FRAGS * FRAG_HALFS) || dev_alloc(&m->fw_frags_inf, N_FW_FRAGS * FRAG_HALFS)) { delete m; return 1; }
	HIPCHK(hipMemcpy(m->gm_dev, &m->gm, sizeof(GridMeta), hipMemcpyHostToDevice));
	HIPCHK(hipMemset(m->grads, 0, P * 2)); HIPCHK(hipMemset(m->adam_m, 0, P * 4)); HIPCHK(hipMemset(m->adam_v, 0, P * 4));
	HIPCHK(hipMemset(m->ema, 0, P * 4)); HIPCHK(hipMemset(m->adam_steps, 0, P * 2));
	HIPCHK(hipMemset(m->fw_frags, 0, N_FW_FRAGS * FRAG_HALFS * 2)); HIPCHK(hipMemset(m->bw_frags, 0, N_BW_FRAGS * FRAG_HALFS * 2));
	HIPCHK(hipMemset(m->fw_frags_inf, 0, N_FW_FRAGS * FRAG_HALFS * 2));
	std::vector<uint32_t> fwp, bwp;
	build_enc_perms(fwp, bwp);
	HIPCHK(hipMemcpy(m->fw_perm, fwp.data(), fwp.size() * 4, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(m->bw_perm, bwp.data(), bwp.size() * 4, hipMemcpyHostToDevice));
	m->n_partials = wgrad_n_partials();
	if (dev_alloc(&m->wgrad_partials, (size_t)m->n_partials * 8 * 16 * 64)) { delete m; return 1; }
	// Trainer::initialize_params [tcnn]: Xavier-uniform matrices, then U(-1e-4, 1e-4) for the grid, one pcg32{seed} stream
	std::vector<float> init(P);
	Rng rnd = make_rng(seed);
	uint64_t p = 0;
	for (const LayerDesc& L : kEncLayers) {
		const float scale = std::sqrt(6.0f / (float)(L.R + L.C));
		for (uint32_t i = 0; i < L.R * L.C; ++i) init[p++] = rnd.next_float() * 2.0f * scale - scale;
	}
	for (; p < P; ++p) init[p] = rnd.next_float() * (1e-4f - (-1e-4f)) + (-1e-4f);
	HIPCHK(hipMemcpy(m->master, init.data(), P * 4, hipMemcpyHostToDevice));
	if (encmlp_refresh_half(m, nullptr)) { delete m; return 1; }
	HIPCHK(hipDeviceSynchronize());
	*out = m;
	return 0;
}
extern "C" void ngp_encmlp_destroy(ngp_encmlp* m) {
	if (!m) return;
	void* ptrs[] = {m->gm_dev, m->master, m->params, m->params_inf, m->grads, m->adam_m, m->adam_v, m->ema, m->adam_steps, m->fw_perm, m->bw_perm, m->fw_frags, m->bw_frags,
		m->fw_frags_inf, m->enc_stash, m->dy_stash, m->wgrad_partials, m->denc_lv, m->bin_vals, m->bin_idxs, m->bin_cursors};
	for (void* p : ptrs) if (p) (void)hipFree(p);
	delete m;
}
extern "C" int ngp_encmlp_n_params(const ngp_encmlp* m, uint64_t* n_params, uint64_t* n_mlp) { if (n_params) *n_params = m->n_params; if (n_mlp) *n_mlp = m->n_mlp; return 0; }
extern "C" int ngp_encmlp_param_ptrs(ngp_encmlp* m, float** master, ngp_half** params, ngp_half** inf, ngp_half** grads) {
	if (master) *master = m->master; if (params) *params = m->params; if (inf) *inf = m->params_inf;
	if (grads) { *grads = m->grads; m->grads_clean = false; } // (a caller that holds the pointer may write: the next training step clears the table itself again)
	return 0;
}
extern "C" int ngp_encmlp_set_params_host(ngp_encmlp* m, const float* p, uint64_t n) {
	REQUIRE(n == m->n_params, "encmlp set_params: size mismatch");
	HIPCHK(hipMemcpy(m->master, p, n * 4, hipMemcpyHostToDevice));
	if (encmlp_refresh_half(m, nullptr)) return 1;
	HIPCHK(hipDeviceSynchronize());
	return 0;
}
extern "C" int ngp_encmlp_get_params_host(ngp_encmlp* m, float* p, uint64_t n) {
	REQUIRE(n == m->n_params, "encmlp get_params: size mismatch");
	HIPCHK(hipMemcpy(p, m->master, n * 4, hipMemcpyDeviceToHost));
	return 0;
}
extern "C" int ngp_encmlp_set_optimizer(ngp_encmlp* m, const ngp_optimizer_config* o) {
	REQUIRE(m && o, "encmlp set_optimizer: null argument");
	m->opt = *o; m->lr = o->learning_rate;
	return 0;
}
extern "C" int ngp_encmlp_inference(ngp_encmlp* m, void* stream, const float* in, uint32_t in_stride, uint32_t n, ngp_half* out, uint32_t out_stride) {
	REQUIRE(in_stride >= m->cfg.n_pos_dims && out_stride >= m->cfg.n_output_dims, "encmlp inference: strides too small");
	launch_encmlp_inference((hipStream_t)stream, m->gm_dev, m->cfg.n_pos_dims, m->params_inf + m->n_mlp, m->fw_frags_inf, in, in_stride, n, out, out_stride, m->cfg.n_output_dims);
	HIPCHK(hipGetLastError());
	return 0;
}
// Trainer::training_step(stream, input, target) (testbed_image.cu:289, testbed_sdf.cu:1557): forward, loss [tcnn l2 / mape / relative_l2 / l1],
// backward, gradients overwritten.  `external`: dL/dy is given instead of targets (parity tests / callers with their own loss).
static int encmlp_training_step(ngp_encmlp* m, hipStream_t s, const float* in, uint32_t in_stride, uint32_t n, const float* target, uint32_t target_stride,
		int loss_type, float loss_scale, const ngp_half* dy, uint32_t dy_stride, float* loss_sum_dev, ngp_half* pred_out, uint32_t pred_stride, bool loss_sum_is_zero = false) {
	REQUIRE(m->cfg.n_output_dims <= 4, "encmlp training: at most 4 output dims (image: 3, SDF: 1)");
	REQUIRE(in_stride >= m->cfg.n_pos_dims, "encmlp training: in_stride too small");
	if (n > m->stash_n) {
		HIPCHK(hipStreamSynchronize(s));
		if (m->enc_stash) HIPCHK(hipFree(m->enc_stash)); if (m->dy_stash) HIPCHK(hipFree(m->dy_stash));
		m->enc_stash = nullptr; m->dy_stash = nullptr; m->stash_n = 0;
		if (dev_alloc(&m->enc_stash, (size_t)((n + 31) / 32) * 2 * 64 * 8)) return 1;
		HIPCHK(hipMalloc(&m->dy_stash, (size_t)((n + 31) / 32) * 32 * 8));
		m->stash_n = n;
	}
	if (!m->grads_clean) HIPCHK(hipMemsetAsync(m->grads + m->n_mlp, 0, (m->n_params - m->n_mlp) * 2, s)); // GradientMode::Overwrite (the optimizer sweep leaves the table zeroed: no launch then)
	m->grads_clean = false;
	if (loss_sum_dev && !loss_sum_is_zero) HIPCHK(hipMemsetAsync(loss_sum_dev, 0, 4, s));
	// The encoding's backward pass through the record lists (round 4; rounds 2-3: atomicAdd(__half2) from the fused kernel, 2.2 ms per 2^18 SDF samples -- the memory side
	// retires ~15 G atomic operations per second): every level -- hashed ones by 4096-entry chunk, dense ones interleaved over the 128 chunks -- is counting-sorted by
	// k_grad_bin and summed exactly in LDS by k_grad_accumulate.  Needs base.json's table size (T = 2^19: all lists share one capacity); other sizes keep the atomics.
	GradBinArgs ba;
	ba.n_hashed = 0; ba.max_chunks = (1u << GRAD_BIN_MAX_TABLE_LOG2) >> 12; ba.chunk_log2 = 12; ba.split = 0; ba.merge_runs = 0; ba.no_dense_merge = 0;
	ba.n_features = 2; ba.n_pos_dims = m->cfg.n_pos_dims;
	if (m->train_encoding && !(g_debug_flags & DBG_T1_NO_BINNING) && m->gm.n_levels <= MAX_LEVELS) {
		bool ok = true;
		for (uint32_t l = 0; l < m->gm.n_levels; ++l) {
			const uint64_t res = m->gm.resolution[l], hs = m->gm.hashmap_size[l];
			const bool dense = (m->cfg.n_pos_dims == 2 ? res * res : res * res * res) <= hs;
			if (dense ? hs > (1ull << GRAD_BIN_MAX_TABLE_LOG2) : hs != (1ull << GRAD_BIN_MAX_TABLE_LOG2)) { ok = false; break; }
			ba.levels[ba.n_hashed++] = l;
		}
		if (!ok) ba.n_hashed = 0;
	}
	if (ba.n_hashed) {
		const uint32_t corners = 1u << m->cfg.n_pos_dims;
		const uint32_t cap_want = std::max<uint32_t>(2048u, (uint32_t)((((uint64_t)n * corners * 2) / ba.max_chunks + 1023) / 1024 * 1024)); // twice the mean number of records per list
		const uint32_t n_lists = ba.n_hashed * ba.max_chunks;
		if (n > m->bin_n || cap_want > m->bin_cap || n_lists != m->bin_lists) {
			HIPCHK(hipStreamSynchronize(s));
			for (void* p : {m->denc_lv, m->bin_vals, m->bin_idxs, (void*)m->bin_cursors}) if (p) HIPCHK(hipFree(p));
			m->denc_lv = m->bin_vals = m->bin_idxs = nullptr; m->bin_cursors = nullptr;
			const uint32_t n_alloc = std::max(n, m->bin_n), cap_alloc = std::max(cap_want, m->bin_cap);
			m->bin_n = m->bin_cap = m->bin_lists = 0;
			HIPCHK(hipMalloc(&m->denc_lv, (size_t)m->gm.n_levels * n_alloc * 4));
			HIPCHK(hipMalloc(&m->bin_vals, (size_t)n_lists * cap_alloc * 4));
			HIPCHK(hipMalloc(&m->bin_idxs, (size_t)n_lists * cap_alloc * 2));
			HIPCHK(hipMalloc((void**)&m->bin_cursors, (size_t)n_lists * 2 * 4));
			HIPCHK(hipMemsetAsync(m->bin_cursors, 0, (size_t)n_lists * 2 * 4, s));
			m->bin_n = n_alloc; m->bin_cap = cap_alloc; m->bin_lists = n_lists;
		}
	}
	EncTrainArgs a;
	a.gm = m->gm_dev; a.table = (const __half*)(m->params + m->n_mlp); a.fw_frags = m->fw_frags; a.bw_frags = m->bw_frags;
	a.in = in; a.in_stride = in_stride; a.n = n; a.target = target; a.target_stride = target_stride; a.n_out = m->cfg.n_output_dims;
	a.loss_type = loss_type; a.loss_scale = loss_scale; a.dy_in = dy; a.dy_stride = dy_stride;
	a.grid_grad = m->train_encoding ? (__half*)(m->grads + m->n_mlp) : nullptr;
	a.enc_stash = (uint4*)m->enc_stash; a.dy_stash = (uint2*)m->dy_stash; a.loss_sum = loss_sum_dev; a.pred_out = pred_out; a.pred_stride = pred_stride;
	if (ba.n_hashed) { a.denc_lv = (uint32_t*)m->denc_lv; a.denc_cap = m->bin_n; }
	launch_encmlp_train(s, a, m->cfg.n_pos_dims, dy != nullptr, m->wgrad_partials, m->n_partials, m->grads);
	if (ba.n_hashed) {
		ba.gm = m->gm_dev; ba.in = in; ba.in_stride = in_stride; ba.n = n; ba.denc_lv = m->denc_lv; ba.denc_cap = m->bin_n; ba.cap = m->bin_cap;
		ba.vals = m->bin_vals; ba.idxs = (uint16_t*)m->bin_idxs; ba.cursors = m->bin_cursors; ba.cursor_done = m->bin_cursors + m->bin_lists; ba.grid_grad_ = m->grads + m->n_mlp;
		ba.fuse_adam = 0;
		launch_grad_bin(s, ba);
	}
	HIPCHK(hipGetLastError());
	return 0;
}
extern "C" int ngp_encmlp_training_step(ngp_encmlp* m, void* stream, const float* in, uint32_t in_stride, uint32_t n, const float* target, uint32_t target_stride,
		int loss_type, float loss_scale, float* loss_sum, ngp_half* pred_out, uint32_t pred_stride) {
	REQUIRE(target && target_stride >= m->cfg.n_output_dims, "encmlp training_step: targets missing / stride too small");
	REQUIRE(loss_type == NGP_LOSS_L2 || loss_type == NGP_LOSS_L1 || loss_type == NGP_LOSS_MAPE || loss_type == NGP_LOSS_RELATIVE_L2, "encmlp training_step: loss must be L2, L1, MAPE or RelativeL2");
	return encmlp_training_step(m, (hipStream_t)stream, in, in_stride, n, target, target_stride, loss_type, loss_scale, nullptr, 0, loss_sum, pred_out, pred_stride);
}
extern "C" int ngp_encmlp_training_step_external(ngp_encmlp* m, void* stream, const float* in, uint32_t in_stride, uint32_t n, const ngp_half* dL_dy, uint32_t dy_stride) {
	REQUIRE(dL_dy && dy_stride >= m->cfg.n_output_dims, "encmlp training_step_external: dL/dy missing / stride too small");
	return encmlp_training_step(m, (hipStream_t)stream, in, in_stride, n, nullptr, 0, NGP_LOSS_L2, 1.f, dL_dy, dy_stride, nullptr, nullptr, 0);
}
extern "C" int ngp_encmlp_optimizer_step(ngp_encmlp* m, void* stream, float loss_scale) {
	++m->step;
	AdamArgs a;
	a.n_params = m->n_params; a.n_mlp = m->n_mlp; a.loss_scale = loss_scale; a.lr = m->lr;
	a.beta1 = m->opt.beta1; a.beta2 = m->opt.beta2; a.eps = m->opt.epsilon; a.l2_reg = m->opt.l2_reg;
	a.log_beta1 = std::log(m->opt.beta1); a.log_beta2 = std::log(m->opt.beta2);
	REQUIRE(65535.0f * a.log_beta1 < -18.f && 65535.0f * a.log_beta2 < -18.f, "Adam: beta too close to 1 for the 16-bit saturating per-parameter step counters (1 - beta^65535 must round to 1)");
	a.optimize_matrix = m->train_network; a.optimize_non_matrix = m->train_encoding; a.zero_grid_grads = 1; // (the sweep clears the grid gradients it consumed, as in the NeRF model)
	const float d = m->opt.ema_decay; // 0 without an Ema wrapper: the inference parameters then equal the parameters
	a.ema_decay = d;
	a.ema_debias_old = 1 - std::pow(d, (float)(m->step - 1));
	a.ema_debias_new = 1 / (1 - std::pow(d, (float)m->step));
	a.master = m->master; a.params = m->params; a.params_inf = m->params_inf; a.grads = m->grads;
	a.m = m->adam_m; a.v = m->adam_v; a.steps = m->adam_steps; a.ema = m->ema;
	a.fw_perm = m->fw_perm; a.bw_perm = m->bw_perm; a.fw_frags = m->fw_frags; a.bw_frags = m->bw_frags; a.fw_frags_inf = m->fw_frags_inf;
	launch_optimizer_step((hipStream_t)stream, a);
	HIPCHK(hipGetLastError());
	m->grads_clean = true;
	if (m->opt.decay_interval > 0 && m->step >= m->opt.decay_start && m->step % m->opt.decay_interval == 0) m->lr *= m->opt.decay_base;
	return 0;
}
extern "C" float ngp_encmlp_learning_rate(const ngp_encmlp* m) { return m->lr; }
extern "C" uint32_t ngp_encmlp_step(const ngp_encmlp* m) { return m->step; }

// ------------------------------------------------------------------------------------------------
// image trainer: Testbed::m_image, train_image (testbed_image.cu:231-302), compute_image_mse (:490-547)
// ------------------------------------------------------------------------------------------------
struct ngp_image {
	ngp_encmlp* model = nullptr;
	ngp_image_options opt{};
	void* pixels = nullptr; int type = NGP_IMAGE_FLOAT; int width = 0, height = 0;
	float* positions = nullptr; float* targets = nullptr; uint32_t batch_cap = 0;
	float* loss_sum = nullptr; double* mse_sum = nullptr; ngp_half* pred = nullptr;
	Rng rng; uint32_t training_step = 0;
};
extern "C" int ngp_image_create(ngp_encmlp* model, const void* pixels_host, int32_t type, int32_t w, int32_t h, const ngp_image_options* o, ngp_image** out) {
	REQUIRE(model && pixels_host && o && out, "ngp_image_create: null argument");
	REQUIRE(model->cfg.n_pos_dims == 2 && model->cfg.n_output_dims == 3, "ngp_image_create: the model must map 2-D positions to 3 outputs (network_dims_image, testbed_image.cu:30-36)");
	REQUIRE((type == NGP_IMAGE_FLOAT || type == NGP_IMAGE_HALF) && w > 1 && h > 1, "ngp_image_create: RGBA float32 / half images of at least 2x2 pixels");
	REQUIRE(o->batch_size > 0 && o->batch_size % 32 == 0, "ngp_image_create: batch size must be a positive multiple of 32");
	ngp_image* t = new ngp_image();
	t->model = model; t->opt = *o; t->type = type; t->width = w; t->height = h;
	t->rng = make_rng(o->seed);
	const size_t bytes = (size_t)w * h * 4 * (type == NGP_IMAGE_FLOAT ? 4 : 2);
	t->batch_cap = std::max<uint32_t>(o->batch_size, 1u << 20); // compute_image_mse evaluates batches of 2^20 pixels
	if (dev_alloc((char**)&t->pixels, bytes) || dev_alloc(&t->positions, (size_t)t->batch_cap * 2) || dev_alloc(&t->targets, (size_t)t->batch_cap * 3) ||
		dev_alloc(&t->loss_sum, 1) || dev_alloc(&t->mse_sum, 1) || dev_alloc(&t->pred, (size_t)t->batch_cap * 4)) { delete t; return 1; }
	HIPCHK(hipMemcpy(t->pixels, pixels_host, bytes, hipMemcpyHostToDevice));
	HIPCHK(hipMemset(t->loss_sum, 0, 4));
	*out = t;
	return 0;
}
extern "C" void ngp_image_destroy(ngp_image* t) {
	if (!t) return;
	(void)hipDeviceSynchronize();
	for (void* p : {t->pixels, (void*)t->positions, (void*)t->targets, (void*)t->loss_sum, (void*)t->mse_sum, (void*)t->pred}) if (p) (void)hipFree(p);
	delete t;
}
static ImageBatchArgs image_args(ngp_image* t, uint32_t n) {
	ImageBatchArgs a;
	a.pixels = t->pixels; a.image_data_type = t->type; a.width = t->width; a.height = t->height; a.n = n; a.rng = pod(t->rng); a.stratify_log2 = 0;
	a.snap_to_pixel_centers = t->opt.snap_to_pixel_centers; a.linear_colors = t->opt.linear_colors; a.positions = t->positions; a.targets = t->targets;
	return a;
}
extern "C" int ngp_image_train(ngp_image* t, void* stream, uint32_t n_steps) {
	hipStream_t s = (hipStream_t)stream;
	const uint32_t n = t->opt.batch_size;
	for (uint32_t i = 0; i < n_steps; ++i) {
		ImageBatchArgs a = image_args(t, n);
		if (t->opt.stratified) { // "Can't stratify a non-pot / non-square batch size" (testbed_image.cu:252-259): plain uniform positions then
			uint32_t l2 = 0; while ((1u << l2) < n) ++l2;
			if ((1u << l2) == n && l2 % 2 == 0) a.stratify_log2 = l2;
		}
		a.zero_word = t->loss_sum; // (the step's loss sum starts from zero without a memset launch)
		launch_image_generate_batch(s, a);
		t->rng.advance((uint64_t)n * 2ull); // generate_random_uniform advances the generator by the number of elements [tcnn]
		if (encmlp_training_step(t->model, s, t->positions, 2, n, t->targets, 3, t->opt.loss_type, t->opt.loss_scale, nullptr, 0, t->loss_sum, nullptr, 0, true)) return 1;
		if (ngp_encmlp_optimizer_step(t->model, stream, t->opt.loss_scale)) return 1;
		++t->training_step;
	}
	HIPCHK(hipGetLastError());
	return 0;
}
extern "C" int ngp_image_loss(ngp_image* t, void* stream, float* loss_host) {
	HIPCHK(hipMemcpyAsync(loss_host, t->loss_sum, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
	HIPCHK(hipStreamSynchronize((hipStream_t)stream));
	return 0;
}
extern "C" int ngp_image_mse(ngp_image* t, int quantize_to_byte, float* mse_host) {
	const uint32_t n_elements = (uint32_t)t->width * (uint32_t)t->height, max_batch = 1u << 20;
	HIPCHK(hipMemsetAsync(t->mse_sum, 0, 8, nullptr));
	for (uint32_t offset = 0; offset < n_elements; offset += max_batch) {
		const uint32_t n = std::min(max_batch, n_elements - offset);
		ImageBatchArgs a = image_args(t, n);
		launch_image_pixel_batch(nullptr, a, offset);
		if (ngp_encmlp_inference(t->model, nullptr, t->positions, 2, n, t->pred, 4)) return 1;
		launch_image_mse(nullptr, n, t->targets, t->pred, 4, quantize_to_byte, t->mse_sum);
	}
	double sum = 0;
	HIPCHK(hipMemcpy(&sum, t->mse_sum, 8, hipMemcpyDeviceToHost));
	*mse_host = (float)(sum / (double)n_elements);
	return 0;
}
extern "C" int ngp_image_batch_ptrs(ngp_image* t, float** positions, float** targets) { if (positions) *positions = t->positions; if (targets) *targets = t->targets; return 0; }

// ------------------------------------------------------------------------------------------------
// SDF trainer: Testbed::m_sdf, load_mesh (testbed_sdf.cu:1363-1447), generate_training_samples_sdf (:1449-1544), train_sdf (:1580-1622),
// calculate_iou (:1636-1680); TriangleBvh::build (triangle_bvh.cu:757-840) as a binary tree
// ------------------------------------------------------------------------------------------------
static uint32_t sdf_default_batches_ahead();
struct ngp_sdf {
	ngp_encmlp* model = nullptr;
	ngp_sdf_options opt{};
	ngp_aabb aabb{};
	uint32_t n_triangles = 0;
	SdfTriangle* tris = nullptr; SdfBvhNode4* nodes = nullptr; int root = 0; uint32_t stack_entries = 4; float* cdf = nullptr;
	float* positions = nullptr; float* distances = nullptr; ngp_half* pred = nullptr; uint32_t cap = 0;
	float* loss_sum = nullptr; uint32_t* iou_counters = nullptr;
	uint32_t* stab_list = nullptr; uint32_t* stab_count = nullptr; float* stab_offsets = nullptr; // scratch of the ground-truth launches (sdf_kernels.hip, SdfQueryScratch): 6 x cap words, 4 counter words, cap x 2 lattice offsets
	void* sort_temp = nullptr; size_t sort_temp_bytes = 0;
	SdfQueryScratch query() const { return {stab_list, stab_list + cap, stab_count, stab_list + 2 * (size_t)cap, stab_list + 3 * (size_t)cap, stab_list + 4 * (size_t)cap, stab_list + 5 * (size_t)cap, sort_temp, sort_temp_bytes, stab_offsets, stab_count + 1}; }
	Rng rng; uint32_t training_step = 0;
	// The batches ahead (positions + ground truth: they depend on the rng stream and the mesh, on no parameter) are generated on a side stream while the current ones train, a
	// GROUP of batches per ground-truth launch (round 6): two group-sized buffer pairs and two events.  `start` = the rng state a group's first batch was drawn at: a group serves
	// the trainer only while the trainer's own rng stands where the group's next batch begins (anything else that draws in between -- calculate_iou -- invalidates what was
	// generated ahead; it is then generated again).
	struct Group { float* positions = nullptr; float* distances = nullptr; uint32_t count = 0, next = 0; Rng start; bool in_flight = false; } group[2];
	uint32_t cur_group = 0, group_cap = 0, batches_ahead = 0 /* ngp_sdf_set_batches_ahead; 0 = serial loop */; hipEvent_t ev_free = nullptr, ev_batch = nullptr;
	float* batch_positions = nullptr; float* batch_distances = nullptr; // where the last trained batch lies (null: positions / distances)
};
// load_mesh's normalisation (testbed_sdf.cu:1380-1410): raw box inflated by 0.5 % of its diagonal, scaled by its largest extent and centred in the unit cube
extern "C" int ngp_sdf_normalize_mesh_host(float* v, uint64_t n_vertices, ngp_aabb* aabb_out, float* mesh_scale_out) {
	REQUIRE(v && n_vertices >= 3 && aabb_out, "ngp_sdf_normalize_mesh_host: null / empty");
	float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
	for (uint64_t i = 0; i < n_vertices; ++i) for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], v[i * 3 + k]); mx[k] = std::max(mx[k], v[i * 3 + k]); }
	const float inflation = 0.005f;
	auto diag_len = [&]() { const float d[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]}; return std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]); };
	float amt = diag_len() * inflation;
	for (int k = 0; k < 3; ++k) { mn[k] -= amt; mx[k] += amt; }
	const float scale = std::max(std::max(mx[0] - mn[0], mx[1] - mn[1]), mx[2] - mn[2]);
	for (uint64_t i = 0; i < n_vertices; ++i) for (int k = 0; k < 3; ++k) v[i * 3 + k] = (v[i * 3 + k] - mn[k] - 0.5f * (mx[k] - mn[k])) / scale + 0.5f;
	for (int k = 0; k < 3; ++k) { mn[k] = INFINITY; mx[k] = -INFINITY; }
	for (uint64_t i = 0; i < n_vertices; ++i) for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], v[i * 3 + k]); mx[k] = std::max(mx[k], v[i * 3 + k]); }
	amt = diag_len() * inflation;
	for (int k = 0; k < 3; ++k) { aabb_out->min[k] = std::max(mn[k] - amt, 0.0f); aabb_out->max[k] = std::min(mx[k] + amt, 1.0f); } // intersection with the unit cube
	if (mesh_scale_out) *mesh_scale_out = scale;
	return 0;
}
// returns the depth of the tree (root = 0): the binary tree the 4-wide device tree is folded from (sdf_flatten_bvh)
static uint32_t sdf_build_bvh(std::vector<SdfTriangle>& tris, std::vector<SdfBvhNode>& nodes, uint32_t leaf_size) {
	struct Job { int node; size_t begin, end; uint32_t depth; };
	uint32_t max_depth = 0;
	auto bounds = [&](size_t b, size_t e, SdfBvhNode& n) {
		for (int k = 0; k < 3; ++k) { n.bmin[k] = INFINITY; n.bmax[k] = -INFINITY; }
		for (size_t i = b; i < e; ++i) for (const float* p : {tris[i].a, tris[i].b, tris[i].c}) for (int k = 0; k < 3; ++k) { n.bmin[k] = std::min(n.bmin[k], p[k]); n.bmax[k] = std::max(n.bmax[k], p[k]); }
	};
	auto centroid = [](const SdfTriangle& t, int k) { return (t.a[k] + t.b[k] + t.c[k]) / 3; };
	nodes.clear(); nodes.emplace_back();
	bounds(0, tris.size(), nodes[0]);
	std::vector<Job> stack{{0, 0, tris.size(), 0u}};
	while (!stack.empty()) {
		const Job j = stack.back(); stack.pop_back();
		max_depth = std::max(max_depth, j.depth);
		if (j.end - j.begin <= leaf_size) { nodes[j.node].left = -(int)j.begin - 1; nodes[j.node].right = -(int)j.end - 1; continue; }
		// axis of maximum centroid variance, median split (triangle_bvh.cu:788-809)
		double mean[3] = {0, 0, 0}, var[3] = {0, 0, 0};
		for (size_t i = j.begin; i < j.end; ++i) for (int k = 0; k < 3; ++k) mean[k] += centroid(tris[i], k);
		for (int k = 0; k < 3; ++k) mean[k] /= (double)(j.end - j.begin);
		for (size_t i = j.begin; i < j.end; ++i) for (int k = 0; k < 3; ++k) { const double d = centroid(tris[i], k) - mean[k]; var[k] += d * d; }
		const int axis = var[0] >= var[1] && var[0] >= var[2] ? 0 : (var[1] >= var[2] ? 1 : 2);
		const size_t mid = j.begin + (j.end - j.begin) / 2;
		std::nth_element(tris.begin() + j.begin, tris.begin() + mid, tris.begin() + j.end, [&](const SdfTriangle& x, const SdfTriangle& y) { return centroid(x, axis) < centroid(y, axis); });
		const int l = (int)nodes.size(); nodes.emplace_back(); nodes.emplace_back();
		nodes[j.node].left = l; nodes[j.node].right = l + 1;
		bounds(j.begin, mid, nodes[l]); bounds(mid, j.end, nodes[l + 1]);
		stack.push_back({l, j.begin, mid, j.depth + 1}); stack.push_back({l + 1, mid, j.end, j.depth + 1});
	}
	return max_depth;
}
// Device form of the tree (sdf_kernels.hip): 4-wide nodes -- every second level of the binary tree folded away: a node's children are its binary grandchildren (a binary
// child that is a leaf stays one child).  A child reference is the child's index in `out` (inner node) or ~((first triangle << 3) | count) (leaf); empty slots carry
// 0x7fffffff and an empty box.  Returns the root's reference; *depth4 = depth of the 4-wide tree.
static int sdf_flatten_bvh(const std::vector<SdfBvhNode>& nodes, std::vector<SdfBvhNode4>& out, uint32_t* depth4) {
	auto is_leaf = [&](int i) { return nodes[i].left < 0; };
	auto leaf_ref = [&](int i) { const SdfBvhNode& n = nodes[i]; return ~(int)((((uint32_t)(-n.left - 1)) << 3) | (uint32_t)((-n.right - 1) - (-n.left - 1))); };
	out.clear();
	*depth4 = 0;
	if (is_leaf(0)) return leaf_ref(0);
	struct Job { int bin; int slot; uint32_t depth; };
	out.emplace_back();
	std::vector<Job> stack{{0, 0, 0u}};
	while (!stack.empty()) {
		const Job j = stack.back(); stack.pop_back();
		*depth4 = std::max(*depth4, j.depth + 1); // (its children -- leaves at least -- sit one level below)
		int kids[4]; int nk = 0;
		for (int c : {nodes[j.bin].left, nodes[j.bin].right}) {
			if (is_leaf(c)) kids[nk++] = c; else { kids[nk++] = nodes[c].left; kids[nk++] = nodes[c].right; }
		}
		SdfBvhNode4 o;
		for (int k = 0; k < 4; ++k) {
			for (int a = 0; a < 3; ++a) { o.lo[a][k] = INFINITY; o.hi[a][k] = -INFINITY; }
			o.ref[k] = 0x7fffffff; o.pad[k] = 0;
		}
		for (int k = 0; k < nk; ++k) {
			const SdfBvhNode& c = nodes[kids[k]];
			for (int a = 0; a < 3; ++a) { o.lo[a][k] = c.bmin[a]; o.hi[a][k] = c.bmax[a]; }
			if (is_leaf(kids[k])) o.ref[k] = leaf_ref(kids[k]);
			else { o.ref[k] = (int)out.size(); out.emplace_back(); stack.push_back({kids[k], o.ref[k], j.depth + 1}); }
		}
		out[j.slot] = o;
	}
	return 0;
}
constexpr uint32_t SDF_LEAF_SIZE = 4, SDF_MAX_DEPTH4 = 15; // (sdf_kernels.hip: leaves of <= 4 triangles are fetched in one batch; 3 * 15 + 1 = 46 <= 48 stack entries)
// DiscreteDistribution::build over the surface areas (discrete_distribution.h:21-38) -- of the REORDERED triangles, like the reference
static void sdf_surface_cdf(const std::vector<SdfTriangle>& tris, std::vector<float>& cdf) {
	const uint32_t n_triangles = (uint32_t)tris.size();
	cdf.resize(n_triangles);
	std::vector<float> w(n_triangles);
	float total = 0;
	for (uint32_t i = 0; i < n_triangles; ++i) {
		const SdfTriangle& q = tris[i];
		const float e1[3] = {q.b[0] - q.a[0], q.b[1] - q.a[1], q.b[2] - q.a[2]}, e2[3] = {q.c[0] - q.a[0], q.c[1] - q.a[1], q.c[2] - q.a[2]};
		const float cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
		w[i] = 0.5f * std::sqrt(cx * cx + cy * cy + cz * cz);
		total += w[i];
	}
	const float inv = 1 / total;
	float acc = 0;
	for (uint32_t i = 0; i < n_triangles; ++i) { acc += w[i] * inv; cdf[i] = acc; }
	cdf.back() = 1.0f;
}
// Test hook, no GPU: the mesh setup of ngp_sdf_create (BVH build with its triangle reordering, surface CDF) and the ground-truth signed distance of
// csrc/sdf_kernels.hip evaluated ON THE HOST from the same source.  distances_inout holds upper bounds when use_upper_bounds != 0.
extern "C" int ngp_host_sdf_signed_distance(const float* triangles_host, uint32_t n_triangles, const float* positions_host, uint32_t n, float* distances_inout, int use_upper_bounds,
		float* triangles_ordered_out, float* cdf_out) {
	REQUIRE(triangles_host && n_triangles > 0 && (n == 0 || (positions_host && distances_inout)), "ngp_host_sdf_signed_distance: null / empty argument");
	std::vector<SdfTriangle> tris(n_triangles);
	memcpy(tris.data(), triangles_host, (size_t)n_triangles * sizeof(SdfTriangle));
	std::vector<SdfBvhNode> nodes;
	const uint32_t depth = sdf_build_bvh(tris, nodes, SDF_LEAF_SIZE);
	(void)depth;
	std::vector<SdfBvhNode4> nodes2; uint32_t depth4 = 0;
	const int root = sdf_flatten_bvh(nodes, nodes2, &depth4);
	REQUIRE(depth4 <= SDF_MAX_DEPTH4 && n_triangles < (1u << 28), "ngp_host_sdf_signed_distance: BVH deeper than the traversal stack");
	if (triangles_ordered_out) memcpy(triangles_ordered_out, tris.data(), (size_t)n_triangles * sizeof(SdfTriangle));
	if (cdf_out) { std::vector<float> cdf; sdf_surface_cdf(tris, cdf); memcpy(cdf_out, cdf.data(), cdf.size() * 4); }
	host_sdf_signed_distance(n, positions_host, distances_inout, nodes2.data(), root, tris.data(), use_upper_bounds);
	return 0;
}
extern "C" int ngp_sdf_create(ngp_encmlp* model, const float* triangles_host, uint32_t n_triangles, ngp_aabb aabb, const ngp_sdf_options* o, ngp_sdf** out) {
	REQUIRE(model && triangles_host && o && out && n_triangles > 0, "ngp_sdf_create: null / empty argument");
	REQUIRE(model->cfg.n_pos_dims == 3 && model->cfg.n_output_dims == 1, "ngp_sdf_create: the model must map 3-D positions to 1 output (network_dims_sdf)");
	REQUIRE(o->batch_size >= 256 && o->batch_size % 32 == 0, "ngp_sdf_create: batch size must be a multiple of 32, at least 256");
	ngp_sdf* t = new ngp_sdf();
	t->model = model; t->opt = *o; t->aabb = aabb; t->n_triangles = n_triangles;
	t->rng = make_rng(o->seed);
	t->batches_ahead = sdf_default_batches_ahead();
	std::vector<SdfTriangle> tris(n_triangles);
	memcpy(tris.data(), triangles_host, (size_t)n_triangles * sizeof(SdfTriangle));
	std::vector<SdfBvhNode> nodes;
	// m_sdf.triangle_bvh->build(triangles_cpu, 8) reorders the triangles; so does this tree (a binary median split down to 4 triangles: the ground truth does not depend on the
	// tree, the surface samples' triangle order follows it like the reference's follows its own)
	const uint32_t bvh_depth = sdf_build_bvh(tris, nodes, SDF_LEAF_SIZE);
	(void)bvh_depth;
	std::vector<SdfBvhNode4> nodes2; uint32_t depth4 = 0;
	t->root = sdf_flatten_bvh(nodes, nodes2, &depth4);
	if (depth4 > SDF_MAX_DEPTH4 || n_triangles >= (1u << 28)) { delete t; return fail("ngp_sdf_create: BVH deeper than the device traversal stack"); } // median split: binary depth = ceil(log2(n / 4)) <= 26, 4-wide depth = ceil of half of it
	t->stack_entries = 3 * depth4 + 1;
	if (nodes2.empty()) nodes2.emplace_back(); // (a mesh of <= 4 triangles: the root is a leaf reference, the node array is never read)
	std::vector<float> cdf;
	sdf_surface_cdf(tris, cdf);
	t->cap = std::max<uint32_t>(o->batch_size, 1u << 21); // calculate_iou works in batches of 128^3 = 2^21
	if (dev_alloc(&t->tris, n_triangles) || dev_alloc(&t->nodes, nodes2.size()) || dev_alloc(&t->cdf, n_triangles) || dev_alloc(&t->positions, (size_t)t->cap * 3) ||
		dev_alloc(&t->distances, t->cap) || dev_alloc(&t->pred, t->cap) || dev_alloc(&t->loss_sum, 1) || dev_alloc(&t->iou_counters, 8) || dev_alloc(&t->stab_list, (size_t)t->cap * 6) || dev_alloc((char**)&t->sort_temp, t->sort_temp_bytes = sdf_point_sort_temp_bytes(t->cap)) || dev_alloc(&t->stab_count, 4) || dev_alloc(&t->stab_offsets, (size_t)t->cap * 2)) { delete t; return 1; }
	HIPCHK(hipMemcpy(t->tris, tris.data(), tris.size() * sizeof(SdfTriangle), hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(t->nodes, nodes2.data(), nodes2.size() * sizeof(SdfBvhNode4), hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(t->cdf, cdf.data(), cdf.size() * 4, hipMemcpyHostToDevice));
	HIPCHK(hipMemset(t->loss_sum, 0, 4));
	HIPCHK(hipMemset(t->stab_list + t->cap, 0, (size_t)t->cap * 4)); // the first stab rays' "escaped" marks
	HIPCHK(hipMemset(t->stab_count, 0, 16));
	launch_sdf_stab_offsets(nullptr, t->cap, t->stab_offsets);
	HIPCHK(hipDeviceSynchronize());
	*out = t;
	return 0;
}
extern "C" void ngp_sdf_destroy(ngp_sdf* t) {
	if (!t) return;
	(void)hipDeviceSynchronize();
	for (void* p : {(void*)t->tris, (void*)t->nodes, (void*)t->cdf, (void*)t->positions, (void*)t->distances, (void*)t->pred, (void*)t->loss_sum, (void*)t->iou_counters, (void*)t->stab_list, (void*)t->stab_count, (void*)t->stab_offsets, t->sort_temp, (void*)t->group[0].positions, (void*)t->group[0].distances, (void*)t->group[1].positions, (void*)t->group[1].distances}) if (p) (void)hipFree(p);
	if (t->ev_free) (void)hipEventDestroy(t->ev_free);
	if (t->ev_batch) (void)hipEventDestroy(t->ev_batch);
	delete t;
}
// generate_training_samples_sdf for `count` consecutive batches of `n` samples drawn from `rng` on (batch k at positions + k * n * 3 / distances + k * n); the ground truth of all
// of them in one launch.  Returns the rng draws one batch consumes through *draws_per_batch; the caller's rng is not touched.
static int sdf_generate_batches(ngp_sdf* t, hipStream_t s, uint32_t n, bool uniform_only, Rng rng, uint32_t count, float* positions, float* distances, uint64_t* draws_per_batch = nullptr) {
	const uint32_t base = n / 8;
	SdfSampleArgs a;
	a.n = n; a.n_exact = uniform_only ? 0 : base * 4; a.n_surface = uniform_only ? 0 : base * 7;
	a.stddev = std::sqrt(0.75f) / 1024.0f * t->opt.surface_offset_scale; // m_bounding_radius = length(vec3(0.5)) (testbed_sdf.cu:1424)
	a.aabb = t->aabb;
	for (int k = 0; k < 3; ++k) { a.aabb.min[k] -= t->opt.zero_offset; a.aabb.max[k] += t->opt.zero_offset; } // sdf_aabb.inflate(zero_offset)
	a.cdf = t->cdf; a.n_triangles = t->n_triangles; a.triangles = t->tris;
	const uint64_t draws = (uint64_t)n * 3ull + (uint64_t)(a.n_surface - a.n_exact) * 3ull; // generate_random_uniform(n * 3) + generate_random_logistic(n_offset * 3)
	if (draws_per_batch) *draws_per_batch = draws;
	REQUIRE((uint64_t)count * (n - a.n_exact) <= t->cap, "sdf ground truth: more points in one launch than the scratch holds");
	for (uint32_t k = 0; k < count; ++k) {
		a.rng = pod(rng); a.positions = positions + (size_t)k * n * 3; a.distances = distances + (size_t)k * n;
		launch_sdf_generate_positions(s, a);
		rng.advance(draws);
	}
	if (launch_sdf_signed_distance(s, n - a.n_exact, positions + (size_t)a.n_exact * 3, distances + a.n_exact, t->nodes, t->root, t->stack_entries, t->tris, 1, t->query(), count, n)) return fail("sdf ground truth: point sort failed");
	HIPCHK(hipGetLastError());
	return 0;
}
// one batch into the trainer's own buffers, advancing m_rng like the reference (calculate_iou's uniform points)
static int sdf_generate(ngp_sdf* t, hipStream_t s, uint32_t n, bool uniform_only) {
	// the ground-truth scratch is shared with whatever was generated ahead on the side stream
	for (auto& g : t->group) if (g.in_flight) { HIPCHK(hipStreamSynchronize(g_side_stream)); g.in_flight = false; }
	uint64_t draws = 0;
	if (sdf_generate_batches(t, s, n, uniform_only, t->rng, 1, t->positions, t->distances, &draws)) return 1;
	t->rng.advance(draws);
	return 0;
}
static uint32_t sdf_default_batches_ahead() { // batches per ground-truth launch (NGP_SDF_GROUP; 1 = one batch ahead, round 6's first form; NGP_SDF_NO_PREFETCH=1: the serial loop)
	static const bool no_prefetch = getenv("NGP_SDF_NO_PREFETCH") && atoi(getenv("NGP_SDF_NO_PREFETCH")) != 0;
	static const uint32_t g = getenv("NGP_SDF_GROUP") ? (uint32_t)std::min(std::max(atoi(getenv("NGP_SDF_GROUP")), 1), 16) : 16u;
	return no_prefetch ? 0u : g;
}
extern "C" int ngp_sdf_set_batches_ahead(ngp_sdf* t, uint32_t batches) {
	REQUIRE(t && batches <= 16, "ngp_sdf_set_batches_ahead: 0 (serial loop) .. 16 batches per ground-truth launch");
	t->batches_ahead = batches;
	return 0;
}
extern "C" int ngp_sdf_train(ngp_sdf* t, void* stream, uint32_t n_steps) {
	hipStream_t s = (hipStream_t)stream;
	const uint32_t n = t->opt.batch_size;
	// The ground truth of a batch (BVH distance + stab rays) is a launch as long as its longest walk with most lanes idle: 2.2 ms for one batch, 2.6 for two, 4.0 for four
	// (profiles/r06_exp_sdf_multibatch.jsonl), and it depends on the rng stream and the mesh only; the training part (0.5 - 0.7 ms) depends on the batch.  So the batches are
	// generated a GROUP at a time, group k + 1 on a side stream while group k trains -- also across calls (the Testbed trains one step per call).  Same rng positions, same
	// batches, same order of the training steps as the serial loop (ngp_sdf_set_batches_ahead(t, 0) / NGP_SDF_NO_PREFETCH=1).  The trainer's rng advances as batches are
	// CONSUMED: a draw in between (calculate_iou) sees the reference's order, and what was generated ahead from a state that is not the trainer's any more is dropped.
	// The step in a continuous 320-step window: serial loop 2.97 ms, 4 / 8 / 12 / 16 batches ahead 1.50 / 1.36 / 1.30 / 1.29 ms (profiles/r06_ab_sdf_batches_ahead.txt).
	const bool prefetch = t->batches_ahead > 0 && !g_prof_on;
	const uint32_t n_bvh = n - n / 8 * 4;
	const uint32_t G = prefetch ? std::max(1u, std::min(t->batches_ahead, t->cap / std::max(n_bvh, 1u))) : 1u;
	if (t->group_cap < G) {
		for (auto& g : t->group) {
			if (g.in_flight) { HIPCHK(hipStreamSynchronize(g_side_stream)); g.in_flight = false; }
			if (g.positions) (void)hipFree(g.positions);
			if (g.distances) (void)hipFree(g.distances);
			g.positions = nullptr; g.distances = nullptr; g.count = g.next = 0;
			if (dev_alloc(&g.positions, (size_t)G * n * 3) || dev_alloc(&g.distances, (size_t)G * n)) return 1;
		}
		t->group_cap = G;
	}
	if (prefetch) {
		if (create_helper_stream(&g_side_stream, false)) return 1;
		if (!t->ev_free) { HIPCHK(hipEventCreateWithFlags(&t->ev_free, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&t->ev_batch, hipEventDisableTiming)); }
	}
	uint64_t draws = 0;
	{ const uint32_t base = n / 8; draws = (uint64_t)n * 3ull + (uint64_t)(base * 3) * 3ull; }
	auto begins_at = [&](const ngp_sdf::Group& g, uint32_t k) { Rng r = g.start; r.advance(draws * k); return r.state == t->rng.state && r.inc == t->rng.inc; };
	for (uint32_t i = 0; i < n_steps; ++i) {
		ngp_sdf::Group* g = &t->group[t->cur_group];
		if (!(g->next < g->count && begins_at(*g, g->next))) {
			ngp_sdf::Group* h = &t->group[t->cur_group ^ 1u];
			if (h->count > 0 && h->next == 0 && begins_at(*h, 0)) { // the group generated ahead is the one that comes now
				if (h->in_flight) { HIPCHK(hipStreamWaitEvent(s, t->ev_batch, 0)); h->in_flight = false; }
			} else { // nothing usable ahead (first call, or somebody else drew from the rng): this step waits for its ground truth
				if (h->in_flight) { HIPCHK(hipStreamWaitEvent(s, t->ev_batch, 0)); h->in_flight = false; } // (one scratch for every ground-truth launch)
				h->count = std::min(G, n_steps - i); h->next = 0; h->start = t->rng;
				if (sdf_generate_batches(t, s, n, false, h->start, h->count, h->positions, h->distances)) return 1; // training_prep_sdf with generate_sdf_data_online (the shuffle of train_sdf permutes a full batch: no effect on its sum)
			}
			t->cur_group ^= 1u; g = h;
			if (prefetch) { // the next group into the pair that has just been used up: it was last read by the steps before this point of the caller's stream, and so was the scratch
				ngp_sdf::Group* f = &t->group[t->cur_group ^ 1u];
				f->count = G; f->next = 0; f->start = g->start; f->start.advance(draws * g->count);
				HIPCHK(hipEventRecord(t->ev_free, s)); HIPCHK(hipStreamWaitEvent(g_side_stream, t->ev_free, 0));
				if (sdf_generate_batches(t, g_side_stream, n, false, f->start, f->count, f->positions, f->distances)) return 1;
				HIPCHK(hipEventRecord(t->ev_batch, g_side_stream));
				f->in_flight = true;
			}
		}
		float* pos = g->positions + (size_t)g->next * n * 3; float* dst = g->distances + (size_t)g->next * n;
		++g->next;
		t->rng.advance(draws);
		if (encmlp_training_step(t->model, s, pos, 3, n, dst, 1, t->opt.loss_type, t->opt.loss_scale, nullptr, 0, t->loss_sum, nullptr, 0)) return 1;
		if (ngp_encmlp_optimizer_step(t->model, stream, t->opt.loss_scale)) return 1;
		++t->training_step;
		t->batch_positions = pos; t->batch_distances = dst; // ngp_sdf_batch_ptrs: the last trained batch
	}
	return 0;
}
extern "C" int ngp_sdf_loss(ngp_sdf* t, void* stream, float* loss_host) {
	HIPCHK(hipMemcpyAsync(loss_host, t->loss_sum, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
	HIPCHK(hipStreamSynchronize((hipStream_t)stream));
	return 0;
}
extern "C" int ngp_sdf_iou(ngp_sdf* t, uint32_t n_samples, double* iou_host) {
	HIPCHK(hipMemsetAsync(t->iou_counters, 0, 32, nullptr));
	while (n_samples > 0) {
		const uint32_t n = std::min<uint32_t>(128u * 128u * 128u, n_samples);
		n_samples -= n;
		if (sdf_generate(t, nullptr, n, true)) return 1;
		if (ngp_encmlp_inference(t->model, nullptr, t->positions, 3, n, t->pred, 1)) return 1;
		launch_sdf_compare_signs(nullptr, n, t->distances, t->pred, 1, t->iou_counters);
	}
	uint32_t c[8];
	HIPCHK(hipMemcpy(c, t->iou_counters, 32, hipMemcpyDeviceToHost));
	*iou_host = c[5] ? (double)c[4] / (double)c[5] : 0.0;
	return 0;
}
extern "C" int ngp_sdf_batch_ptrs(ngp_sdf* t, float** positions, float** distances) {
	if (positions) *positions = t->batch_positions ? t->batch_positions : t->positions;
	if (distances) *distances = t->batch_distances ? t->batch_distances : t->distances;
	return 0;
}
extern "C" int ngp_sdf_signed_distance(ngp_sdf* t, void* stream, const float* positions, uint32_t n, float* out) {
	REQUIRE(t && (n == 0 || (positions && out)), "ngp_sdf_signed_distance: null argument");
	for (auto& g : t->group) if (g.in_flight) { HIPCHK(hipStreamSynchronize(g_side_stream)); g.in_flight = false; } // one scratch for every ground-truth launch
	for (uint32_t done = 0; done < n; done += t->cap) // the survivor list of the stab rays holds t->cap points
		if (launch_sdf_signed_distance((hipStream_t)stream, std::min(n - done, t->cap), positions + (size_t)done * 3, out + done, t->nodes, t->root, t->stack_entries, t->tris, 0, t->query())) return fail("sdf ground truth: point sort failed");
	HIPCHK(hipGetLastError());
	return 0;
}

// arguments of optimizer step number `step` (Adam::step has already counted it)
static AdamArgs make_adam_args(const ngp_model* m, float loss_scale, uint32_t step) {
	AdamArgs a;
	a.n_params = m->n_params; a.n_mlp = m->n_mlp; a.loss_scale = loss_scale; a.lr = m->lr;
	a.beta1 = m->cfg.beta1; a.beta2 = m->cfg.beta2; a.eps = m->cfg.epsilon; a.l2_reg = m->cfg.l2_reg;
	a.log_beta1 = std::log(m->cfg.beta1); a.log_beta2 = std::log(m->cfg.beta2);
	a.optimize_matrix = m->train_network; a.optimize_non_matrix = m->train_encoding;
	a.zero_grid_grads = !(g_debug_flags & DBG_NO_GRAD_ZERO_IN_OPTIMIZER);
	const float d = m->cfg.ema_decay;
	a.ema_decay = d;
	a.ema_debias_old = 1 - std::pow(d, (float)(step - 1));
	a.ema_debias_new = 1 / (1 - std::pow(d, (float)step));
	a.master = m->master; a.params = m->params; a.params_inf = m->params_inf; a.grads = m->grads;
	a.m = m->adam_m; a.v = m->adam_v; a.steps = m->adam_steps; a.ema = m->ema; a.ema_full_precision = m->cfg.ema_full_precision ? 1 : 0;
	a.fw_perm = m->fw_perm; a.bw_perm = m->bw_perm; a.fw_frags = m->fw_frags; a.bw_frags = m->bw_frags; a.fw_frags_inf = m->fw_frags_inf;
	return a;
}
extern "C" int ngp_model_optimizer_step(ngp_model* m, void* stream, float loss_scale) {
	FlagScope flag_scope_(m->dbg);
	++m->step; // Adam::step: ++m_current_step
	AdamArgs a = make_adam_args(m, loss_scale, m->step);
	if (!(65535.0f * a.log_beta1 < -18.f && 65535.0f * a.log_beta2 < -18.f)) {
		--m->step; m->adam_fused_pending = false; // nothing was applied (the fused epilogue checks the same condition before it is armed)
		REQUIRE(false, "Adam: beta too close to 1 for the 16-bit saturating per-parameter step counters (1 - beta^65535 must round to 1)");
	}
	if (m->adam_fused_pending) { a.n_params = m->adam_sweep_end; m->adam_fused_pending = false; }
	m->last_sweep_params = a.n_params; // the hashed levels were updated by this step's k_grad_accumulate (same arguments)
	{ ProfScope ps(P_OPTIMIZER, (hipStream_t)stream); launch_optimizer_step((hipStream_t)stream, a); }
	m->grads_clean = a.zero_grid_grads != 0;
	HIPCHK(hipGetLastError());
	// ExponentialDecay::step [tcnn]
	if (m->cfg.decay_interval > 0 && m->step >= m->cfg.decay_start && m->step % m->cfg.decay_interval == 0) m->lr *= m->cfg.decay_base;
	return 0;
}
extern "C" int ngp_model_set_trainable(ngp_model* m, int net, int enc) { m->train_network = net != 0; m->train_encoding = enc != 0; return 0; }
extern "C" float ngp_model_learning_rate(const ngp_model* m) { return m->lr; }
extern "C" uint32_t ngp_model_step(const ngp_model* m) { return m->step; }

struct SerHeader { uint32_t magic, version; uint64_t n_params; uint32_t step, with_optimizer; float lr; uint32_t pad; };
__global__ void k_inference_to_ema(const __half* __restrict__ inf, float* __restrict__ ema, uint64_t n) {
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) ema[i] = __half2float(inf[i]);
}
extern "C" uint64_t ngp_model_state_offset(uint64_t n_params, int section) { return sizeof(SerHeader) + (uint64_t)section * n_params * 4; }
extern "C" int ngp_model_state_header(void* buf, uint64_t size, int write, uint64_t* n_params, uint32_t* step, float* lr, uint32_t* with_optimizer) {
	REQUIRE(buf && size >= sizeof(SerHeader) && n_params && step && lr && with_optimizer, "ngp_model_state_header: null argument / truncated buffer");
	if (write) { const SerHeader h = {0x4E475031u, 1, *n_params, *step, (uint32_t)(*with_optimizer != 0), *lr, 0}; memcpy(buf, &h, sizeof(h)); return 0; }
	SerHeader h; memcpy(&h, buf, sizeof(h));
	REQUIRE(h.magic == 0x4E475031u && h.version == 1, "ngp_model_state_header: bad magic/version");
	*n_params = h.n_params; *step = h.step; *lr = h.lr; *with_optimizer = h.with_optimizer;
	return 0;
}
extern "C" int ngp_model_get_config(const ngp_model* m, ngp_model_config* out) { REQUIRE(m && out, "ngp_model_get_config: null argument"); *out = m->cfg; return 0; }
extern "C" uint64_t ngp_model_serialized_size(const ngp_model* m, int with_optimizer) {
	return sizeof(SerHeader) + m->n_params * 4 * (with_optimizer ? 5 : 1);
}
extern "C" int ngp_model_serialize_host(ngp_model* m, void* buf, uint64_t size, int with_optimizer) {
	REQUIRE(size >= ngp_model_serialized_size(m, with_optimizer), "serialize: buffer too small");
	REQUIRE(!m->dp_state_stale, kStaleMsg);
	SerHeader h = {0x4E475031u, 1, m->n_params, m->step, (uint32_t)(with_optimizer != 0), m->lr, 0};
	char* p = (char*)buf;
	memcpy(p, &h, sizeof(h)); p += sizeof(h);
	const size_t nb = m->n_params * 4;
	HIPCHK(hipMemcpy(p, m->master, nb, hipMemcpyDeviceToHost)); p += nb;
	if (with_optimizer) {
		HIPCHK(hipMemcpy(p, m->adam_m, nb, hipMemcpyDeviceToHost)); p += nb;
		HIPCHK(hipMemcpy(p, m->adam_v, nb, hipMemcpyDeviceToHost)); p += nb;
		{ // the blob keeps 32-bit counters (format version 1); the device holds 16-bit saturating ones
			std::vector<uint16_t> st16(m->n_params);
			HIPCHK(hipMemcpy(st16.data(), m->adam_steps, m->n_params * 2, hipMemcpyDeviceToHost));
			uint32_t* dst = (uint32_t*)p;
			for (uint64_t i = 0; i < m->n_params; ++i) dst[i] = st16[i];
			p += nb;
		}
		// EMA section (fp32 in this private blob): the fp32 state in "full_precision" mode, else the half inference parameters widened (exactly) -- they ARE the state
		if (!m->cfg.ema_full_precision) hipLaunchKernelGGL(k_inference_to_ema, dim3((uint32_t)((m->n_params + 255) / 256)), dim3(256), 0, 0, (const __half*)m->params_inf, m->ema, m->n_params);
		HIPCHK(hipMemcpy(p, m->ema, nb, hipMemcpyDeviceToHost)); p += nb;
	}
	return 0;
}
__global__ void k_ema_to_inference(const float* __restrict__ ema, __half* __restrict__ inf, uint64_t n) {
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) inf[i] = __float2half(ema[i]);
}
extern "C" int ngp_model_deserialize_host(ngp_model* m, const void* buf, uint64_t size) {
	REQUIRE(size >= sizeof(SerHeader), "deserialize: truncated");
	SerHeader h; memcpy(&h, buf, sizeof(h));
	REQUIRE(h.magic == 0x4E475031u && h.version == 1, "deserialize: bad magic/version");
	REQUIRE(h.n_params == m->n_params, "deserialize: parameter count mismatch");
	REQUIRE(size >= sizeof(SerHeader) + m->n_params * 4 * (h.with_optimizer ? 5 : 1), "deserialize: truncated");
	const char* p = (const char*)buf + sizeof(h);
	const size_t nb = m->n_params * 4;
	HIPCHK(hipMemcpy(m->master, p, nb, hipMemcpyHostToDevice)); p += nb;
	if (model_refresh_half(m, nullptr)) return 1;
	if (h.with_optimizer) {
		HIPCHK(hipMemcpy(m->adam_m, p, nb, hipMemcpyHostToDevice)); p += nb;
		HIPCHK(hipMemcpy(m->adam_v, p, nb, hipMemcpyHostToDevice)); p += nb;
		{
			std::vector<uint16_t> st16(m->n_params);
			const uint32_t* src = (const uint32_t*)p;
			for (uint64_t i = 0; i < m->n_params; ++i) st16[i] = (uint16_t)std::min<uint32_t>(src[i], 0xFFFFu);
			HIPCHK(hipMemcpy(m->adam_steps, st16.data(), m->n_params * 2, hipMemcpyHostToDevice));
			p += nb;
		}
		HIPCHK(hipMemcpy(m->ema, p, nb, hipMemcpyHostToDevice)); p += nb;
		m->step = h.step; m->lr = h.lr;
		if (m->step > 0) {
			hipLaunchKernelGGL(k_ema_to_inference, dim3((uint32_t)((m->n_params + 255) / 256)), dim3(256), 0, 0, m->ema, (__half*)m->params_inf, m->n_params);
			launch_build_frags(nullptr, m->params_inf, (uint32_t)m->n_mlp, m->fw_perm, m->bw_perm, m->fw_frags_inf, nullptr);
		}
	}
	HIPCHK(hipDeviceSynchronize());
	return 0;
}

// ------------------------------------------------------------------------------------------------
// stand-alone NeRF kernels (test hooks).  The three hooks that need scratch memory keep it in function-static buffers on the current
// device: they are serialised by one mutex and must be used from one device / one stream at a time (the trainer handles own theirs).
// ------------------------------------------------------------------------------------------------
static std::mutex g_hook_mutex;
// stand-alone ngp_k_generate_training_samples / ngp_k_compute_loss only (test hook): CDFs for error-proportional pixel sampling and the error map K3 splats into (device pointers; null = off)
static ErrorCdf g_hook_cdf; static float* g_hook_error_map = nullptr; static int32_t g_hook_error_map_res[2] = {0, 0};
extern "C" int ngp_debug_set_error_sampling(const float* cdf_x_cond_y, const float* cdf_y, const float* cdf_img, const int32_t cdf_res[2], float* error_map, const int32_t error_map_res[2]) {
	g_hook_cdf = ErrorCdf();
	g_hook_cdf.x_cond_y = cdf_x_cond_y; g_hook_cdf.y = cdf_y; g_hook_cdf.img = cdf_img;
	if (cdf_res) { g_hook_cdf.res[0] = cdf_res[0]; g_hook_cdf.res[1] = cdf_res[1]; }
	g_hook_error_map = error_map;
	if (error_map_res) { g_hook_error_map_res[0] = error_map_res[0]; g_hook_error_map_res[1] = error_map_res[1]; }
	return 0;
}
// stand-alone ngp_k_generate_training_samples / ngp_k_compute_loss only (test hook): the per-image extra dims K1 copies behind every NerfCoordinate of the image's rays
// (extra_dims_gpu, testbed_nerf.cu:718-719, 744, 833) and, with them, the row stride 7 + n_extra of the coords buffers of both kernels (device pointer; n_extra = 0 = off)
static const float* g_hook_extra_dims = nullptr; static uint32_t g_hook_n_extra = 0;
extern "C" int ngp_debug_set_extra_dims(const float* extra_dims_device, uint32_t n_extra) {
	REQUIRE(n_extra <= 16 && (n_extra == 0 || extra_dims_device), "ngp_debug_set_extra_dims: 0..16 extra dims, device pointer");
	g_hook_extra_dims = n_extra ? extra_dims_device : nullptr; g_hook_n_extra = n_extra;
	return 0;
}
// Stand-alone ngp_k_generate_training_samples / ngp_k_compute_loss only (test hook): the per-ray target records {rgbtarget[3], background[3], depth, 0} that the trainer's
// k1_setup computes for K3 (K1Args::ray_targets_out / K3Args::ray_targets) -- K1 writes them into `buf` (device, 8 floats per ray slot) with these colour options, K3 reads
// them instead of walking the target-pixel chain itself (its TGT instance when the mode is plain).  plain_dataset: the caller vouches for 8-bit images, Perspective /
// OpenCV lenses and still cameras (k1_setup<PLAIN>).  buf = null: off.
static float* g_hook_ray_targets = nullptr; static int g_hook_plain_dataset = 0; static float g_hook_bg[3] = {0.f, 0.f, 0.f}; static int g_hook_srgb = 0, g_hook_random_bg = 0, g_hook_linear = 0;
extern "C" int ngp_debug_set_ray_targets(float* buf, int plain_dataset, const float* background_color, int color_space_srgb, int random_bg_color, int linear_colors) {
	g_hook_ray_targets = buf; g_hook_plain_dataset = plain_dataset;
	for (int k = 0; k < 3; ++k) g_hook_bg[k] = background_color ? background_color[k] : 0.f;
	g_hook_srgb = color_space_srgb; g_hook_random_bg = random_bg_color; g_hook_linear = linear_colors;
	return 0;
}
extern "C" int ngp_k_generate_training_samples(void* stream, uint32_t n_rays, uint32_t rank, uint32_t world_size, const uint32_t* n_rays_ptr, ngp_aabb aabb,
		uint32_t max_samples, const uint32_t* max_samples_ptr, ngp_pcg32 rng, uint32_t* ray_counter, uint32_t* numsteps_counter, uint32_t* ray_indices_out,
		ngp_ray* rays_out, uint32_t* numsteps_out, float* coords_out, uint32_t n_training_images, const ngp_image_meta* metadata, const ngp_xform* xforms,
		const uint8_t* bitfield, uint32_t max_mip, int snap_to_pixel_centers, float cone_angle_constant) {
	std::lock_guard<std::mutex> hook_lock(g_hook_mutex);
	REQUIRE(world_size >= 1 && rank < world_size, "generate_training_samples: bad rank/world_size");
	K1Args a;
	a.k2_tiles0_out = nullptr; a.k2_tile_w = 32; a.ray_targets_out = nullptr; a.background_color[0] = a.background_color[1] = a.background_color[2] = 0.f; a.color_space_srgb = a.random_bg_color = a.linear_colors = 0;
	a.n_rays = n_rays; a.n_rays_ptr = n_rays_ptr; a.rank = rank; a.world_size = world_size; a.aabb = aabb; a.max_samples = max_samples;
	a.max_samples_ptr = max_samples_ptr; a.rng = rng; a.ray_counter = ray_counter; a.numsteps_counter = numsteps_counter; a.ray_indices_out = ray_indices_out;
	a.rays_out = rays_out; a.numsteps_out = numsteps_out; a.coords_out = coords_out; a.n_images = n_training_images; a.metadata = metadata; a.xforms = xforms;
	a.bitfield = bitfield; a.max_mip = max_mip; a.snap_to_pixel_centers = snap_to_pixel_centers; a.cone_angle_constant = cone_angle_constant;
	a.exact_skip = !(g_debug_flags & DBG_K1_INDEPENDENT_LATTICE); a.clamp_min_max = (g_debug_flags & DBG_K1_MIP_CLAMP_MIN_MAX) ? 1u : 0u;
	a.cdf = g_hook_cdf;
	a.extra_dims = g_hook_extra_dims; a.n_extra = g_hook_n_extra;
	a.plain_dataset = g_hook_plain_dataset;
	if (g_hook_ray_targets) { a.ray_targets_out = g_hook_ray_targets; for (int k = 0; k < 3; ++k) a.background_color[k] = g_hook_bg[k]; a.color_space_srgb = g_hook_srgb; a.random_bg_color = g_hook_random_bg; a.linear_colors = g_hook_linear; }
	static char* s_scratch = nullptr; static size_t s_scratch_bytes = 0;
	static uint8_t* s_linear = nullptr;
	if (!s_linear && dev_alloc(&s_linear, (size_t)GRID_N_CELLS / 8 * N_CASCADES)) return 1;
	static uint32_t* s_coarse = nullptr;
	if (!s_coarse && dev_alloc(&s_coarse, (size_t)k1_prefilter_words(N_CASCADES))) return 1;
	launch_build_linear_bitfield((hipStream_t)stream, bitfield, s_linear, N_CASCADES, s_coarse);
	a.n_mips = N_CASCADES; a.bitfield_linear = s_linear; a.bitfield_coarse = (g_debug_flags & DBG_K1_NO_PREFILTER) ? nullptr : s_coarse; a.chunk_march = (g_debug_flags & DBG_K1_CHUNK_MARCH) != 0; a.no_first_point_skip = (g_debug_flags & DBG_K1_NO_FIRST_POINT_SKIP) != 0;
	const uint32_t max_local = n_rays / world_size + 1;
	if (g_debug_flags & DBG_K1_REFERENCE_LAYOUT) {
		launch_generate_training_samples((hipStream_t)stream, a, max_local);
	} else {
		REQUIRE(n_rays_ptr == nullptr, "stand-alone lattice K1: pass n_rays as an immediate");
		const size_t need = k1_lattice_scratch_bytes(max_local);
		if (need > s_scratch_bytes) {
			HIPCHK(hipDeviceSynchronize());
			if (s_scratch) HIPCHK(hipFree(s_scratch));
			s_scratch = nullptr; s_scratch_bytes = 0;
			if (dev_alloc(&s_scratch, need)) return 1;
			s_scratch_bytes = need;
		}
		if (k1_lattice_scratch_init((hipStream_t)stream, s_scratch, max_local)) return fail("k1 scratch init");
		// the per-wave atomics of the sequential kernel accumulate into the counters; the scan overwrites them
		launch_generate_training_samples_lattice((hipStream_t)stream, a, max_local, s_scratch);
	}
	HIPCHK(hipGetLastError());
	return 0;
}
static int g_k3_train_mode = 0; // stand-alone ngp_k_compute_loss only (test hook): ETrainMode
extern "C" int ngp_debug_set_train_mode(int mode) { g_k3_train_mode = mode; return 0; }
static float g_k3_depth_lambda = 0.f; static int g_k3_depth_loss_type = NGP_LOSS_L1; // stand-alone ngp_k_compute_loss only (test hook): depth supervision
extern "C" int ngp_debug_set_depth_supervision(float lambda, int loss_type) { g_k3_depth_lambda = lambda; g_k3_depth_loss_type = loss_type; return 0; }
extern "C" int ngp_k_compute_loss(void* stream, uint32_t n_rays, const uint32_t* n_rays_ptr, ngp_aabb aabb, ngp_pcg32 rng, uint32_t max_samples_compacted,
		const uint32_t* rays_counter, float loss_scale, const float background_color[3], int color_space_srgb, int random_bg_color, int linear_colors,
		uint32_t n_training_images, const ngp_image_meta* metadata, const ngp_half* network_output, uint32_t output_stride, uint32_t* numsteps_counter_compacted,
		const uint32_t* ray_indices_in, const ngp_ray* rays_in, uint32_t* numsteps_inout, const float* coords_in, float* coords_out, ngp_half* dloss_doutput,
		uint32_t dloss_stride, int loss_type, float* loss_output, int rgb_activation, int density_activation, int snap_to_pixel_centers,
		const float* mean_density_ptr, float near_distance) {
	K3Args a;
	a.ray_targets = g_hook_ray_targets; a.train_mode = g_k3_train_mode; a.depth_lambda = g_k3_depth_lambda; a.depth_loss_type = g_k3_depth_loss_type;
	a.n_rays = n_rays; a.n_rays_ptr = n_rays_ptr; a.aabb = aabb; a.rng = rng; a.max_samples_compacted = max_samples_compacted; a.rays_counter = rays_counter;
	a.loss_scale = loss_scale; for (int k = 0; k < 3; ++k) a.background_color[k] = background_color[k];
	a.color_space_srgb = color_space_srgb; a.random_bg_color = random_bg_color; a.linear_colors = linear_colors; a.n_images = n_training_images; a.metadata = metadata;
	a.network_output = network_output; a.output_stride = output_stride; a.numsteps_counter_compacted = numsteps_counter_compacted; a.ray_indices_in = ray_indices_in;
	a.rays_in = rays_in; a.numsteps_inout = numsteps_inout; a.coords_in = coords_in; a.coords_out = coords_out; a.dloss_doutput = dloss_doutput; a.dloss_stride = dloss_stride;
	a.loss_type = loss_type; a.loss_output = loss_output; a.rgb_activation = rgb_activation; a.density_activation = density_activation;
	a.snap_to_pixel_centers = snap_to_pixel_centers; a.mean_density_ptr = mean_density_ptr; a.near_distance = near_distance;
	a.cdf = g_hook_cdf; a.error_map = g_hook_error_map; a.error_map_res[0] = g_hook_error_map_res[0]; a.error_map_res[1] = g_hook_error_map_res[1];
	a.cstride = 7u + g_hook_n_extra;
	std::lock_guard<std::mutex> hook_lock(g_hook_mutex);
	{ // scratch of the two-pass kernel (ablation DBG_K3_TWO_PASS); static like the other stand-alone hooks' scratch
		static char* s_k3 = nullptr; static size_t s_k3_bytes = 0;
		const size_t need = k3_scratch_bytes(n_rays);
		if (need > s_k3_bytes) {
			HIPCHK(hipDeviceSynchronize());
			if (s_k3) HIPCHK(hipFree(s_k3));
			s_k3 = nullptr; s_k3_bytes = 0;
			if (dev_alloc(&s_k3, need)) return 1;
			s_k3_bytes = need;
		}
		if (k3_scratch_init((hipStream_t)stream, s_k3, n_rays)) return fail("k3 scratch init");
		a.k3_scratch = s_k3;
		REQUIRE(!((g_debug_flags & DBG_K3_TWO_PASS) && (a.error_map || a.cdf.x_cond_y || a.cdf.img)), "the two-pass K3 (ablation) has no error map");
	}
	launch_compute_loss((hipStream_t)stream, a, n_rays);
	HIPCHK(hipGetLastError());
	return 0;
}
extern "C" int ngp_k_construct_error_cdfs(void* stream, uint32_t n_images, uint32_t width, uint32_t height, const float* error_map, float* cdf_x_cond_y, float* cdf_y, float* cdf_img) {
	REQUIRE(n_images >= 1 && width >= 1 && height >= 1 && error_map && cdf_x_cond_y && cdf_y && cdf_img, "ngp_k_construct_error_cdfs: bad argument");
	launch_construct_error_cdfs((hipStream_t)stream, n_images, width, height, error_map, cdf_x_cond_y, cdf_y, cdf_img);
	HIPCHK(hipGetLastError());
	return 0;
}
extern "C" int ngp_k_fill_rollover(void* stream, uint32_t n_elements, const uint32_t* n_input_ptr, float* coords, uint32_t cstride, ngp_half* dloss, uint32_t dstride) {
	launch_fill_rollover((hipStream_t)stream, n_elements, n_input_ptr, coords, cstride, dloss, dstride);
	HIPCHK(hipGetLastError());
	return 0;
}
extern "C" int ngp_k_mark_untrained_density_grid(void* stream, uint32_t n, float* grid, uint32_t n_images, const ngp_image_meta* meta, const ngp_xform* xf, int clear) {
	launch_mark_untrained((hipStream_t)stream, n, grid, n_images, meta, xf, clear);
	HIPCHK(hipGetLastError());
	return 0;
}
extern "C" int ngp_k_generate_grid_samples(void* stream, uint32_t n, ngp_pcg32 rng, uint32_t step, ngp_aabb aabb, const float* grid_in, float* pos, uint32_t* idx,
		uint32_t n_cascades, float thresh) {
	launch_generate_grid_samples((hipStream_t)stream, n, rng, nullptr, step, aabb, grid_in, pos, idx, n_cascades, thresh);
	HIPCHK(hipGetLastError());
	return 0;
}
extern "C" int ngp_k_splat_grid_samples(void* stream, uint32_t n, const uint32_t* idx, const ngp_half* out, uint32_t stride, float* grid, int act) {
	launch_splat_grid_samples((hipStream_t)stream, n, idx, out, stride, grid, act);
	HIPCHK(hipGetLastError());
	return 0;
}
extern "C" int ngp_k_ema_grid_samples(void* stream, uint32_t n, float decay, float* grid_out, const float* grid_in) {
	launch_ema_grid_samples((hipStream_t)stream, n, decay, grid_out, grid_in);
	HIPCHK(hipGetLastError());
	return 0;
}
static float* g_mean_partial = nullptr;
extern "C" int ngp_k_update_mean_and_bitfield(void* stream, const float* grid, uint32_t max_cascade, uint8_t* bitfield, float* mean_out) {
	std::lock_guard<std::mutex> hook_lock(g_hook_mutex);
	if (!g_mean_partial && dev_alloc(&g_mean_partial, 256)) return 1;
	launch_grid_mean((hipStream_t)stream, grid, g_mean_partial, mean_out);
	launch_grid_to_bitfield((hipStream_t)stream, grid, max_cascade, bitfield, mean_out);
	HIPCHK(hipGetLastError());
	return 0;
}
extern "C" int ngp_k_grid_to_bitfield(void* stream, const float* grid, uint32_t max_cascade, uint8_t* bitfield, const float* mean_ptr) {
	launch_grid_to_bitfield((hipStream_t)stream, grid, max_cascade, bitfield, mean_ptr);
	HIPCHK(hipGetLastError());
	return 0;
}

// ------------------------------------------------------------------------------------------------
// NeRF trainer
// ------------------------------------------------------------------------------------------------
struct ngp_nerf {
	ngp_model* model;
	ngp_nerf_options opt;
	ngp_aabb aabb;
	uint32_t n_images = 0, n_images_marked = 0; // n_images_marked: Nerf::Training::n_images_for_training_prev (testbed.h)
	ngp_image_meta* meta_dev = nullptr; ngp_xform* xforms_dev = nullptr; bool plain_dataset = false; // (set_dataset_common)
	std::vector<void*> owned_pixels;
	float* density_grid = nullptr; float* density_grid_tmp = nullptr; uint8_t* bitfield = nullptr; float* mean = nullptr; float* mean_partial = nullptr;
	float* grid_positions = nullptr; uint32_t* grid_indices = nullptr; ngp_half* grid_mlp_out = nullptr; uint32_t grid_sample_cap = 0;
	float* grid_positions_sorted = nullptr; uint32_t* grid_indices_sorted = nullptr; char* grid_sort_temp = nullptr; size_t grid_sort_temp_bytes = 0;
	TrainCounters* counters = nullptr;
	uint32_t* ray_indices = nullptr; ngp_ray* rays = nullptr; uint32_t* numsteps = nullptr;
	// K1 of step n+1 does not depend on the parameters: it is launched on its own stream as soon as step n's controller has run and
	// overlaps step n's backward pass / optimizer (single-rank training, no grid update pending, no per-kernel profiling)
	bool ctl_done = false; // the batch-size controller of the current step has run
	hipStream_t k1_stream = nullptr; hipEvent_t ev_ctl = nullptr, ev_k1 = nullptr; bool k1_prelaunched = false; uint64_t state_version = 0, k1_version = 0; hipStream_t k1_for_stream = nullptr;
	uint32_t k2_rounds = 1, k2_tile_w = 16; // rounds 1 = one launch, every wavefront follows its rays front to back (default); 2..8 = list-driven rounds (round r < last evaluates samples [r w, r w + w) of the rays that are still transparent, the last round the rest).  Tile width 16: two rays per wavefront, 383k instead of 556k evaluations per step.  Measured per step (profiles/r02_microbench_k2.log, r02_microbench_final.log): 3 rounds x 32 = 0.171 ms, 1 x 32 = 0.131 ms, 1 x 16 = 0.112 ms.  NGP_K2_ROUNDS / NGP_K2_TILE override.
	float* k2_T = nullptr; uint4* k2_tiles = nullptr; uint32_t k2_tile_cap = 0; // lazy K2 round B tile descriptors
	float* ray_targets = nullptr; // per active ray: {rgbtarget, background} from k1_setup for K3
	uint4* k2_enc = nullptr; uint32_t* src_index = nullptr; bool k2_enc_valid = false; // K2's per-sample encodings and K3's row -> sample map for T1 (EncStashIn); valid: written by this step's K2 / K3
	float* coords = nullptr; ngp_half* mlp_out = nullptr; float* coords_compacted = nullptr; ngp_half* dloss = nullptr;
	RenderRay* r_rays = nullptr; uint64_t* r_masks = nullptr; uint32_t* r_alive = nullptr; uint32_t* r_n_alive = nullptr; uint32_t* r_n_inf = nullptr; float* r_coords = nullptr; ngp_half* r_out = nullptr;
	char* k1_scratch = nullptr; // RaySetup / occupancy masks / prefix sums of the sample-parallel K1
	char* k3_scratch = nullptr; // per-ray records / workgroup totals of the two-pass K3
	uint8_t* bitfield_linear = nullptr; // x-major copy of the bitfield for the lattice marchers
	uint32_t* bitfield_coarse = nullptr; // one bit per 4x4x4 cells of it (k1_count's prefilter)
	uint32_t* sync2 = nullptr; // {measured_before, measured, loss sum in units of 2^-24} for the cross-rank all-reduce (4 words allocated)
	// in-library data-parallel step over RCCL (ngp_comm_init): communicator, its stream, bucket-reduced events
	void* comm = nullptr; hipStream_t comm_stream = nullptr; hipEvent_t ev_red_a = nullptr, ev_red_b = nullptr; bool grads_pending = false;
	// Sharded data-parallel step (round 5): the hash table's parameters [dp_begin[b], dp_end[b]) of bucket b = 0, 1 (level groups) are cut into world_size equal
	// pieces; rank r owns piece r of both buckets: it receives their summed gradients (reduce-scatter), runs Adam on them and hands the new half parameters to everybody
	// (all-gather).  The MLP (10,240 parameters) stays replicated (its gradients are all-reduced).  dp_sharded = the layout exists and the mode is on.
	DebugOverride dbg; // per-handle ablation switches (ngp_nerf_set_debug_flags; the trainer's model runs under them as well)
	bool dp_sharded = false; uint64_t dp_begin[2] = {0, 0}, dp_end[2] = {0, 0}; hipEvent_t ev_rs_a = nullptr; AdamArgs dp_adam; /* this step's optimizer arguments (local part -> tail) */
	// error-proportional pixel sampling (testbed.h:745-756, 810-815; off unless one of the option switches is set)
	float* error_map = nullptr; size_t error_map_cap = 0; int32_t error_map_res[2] = {0, 0};
	float* cdf_x_cond_y = nullptr; float* cdf_y = nullptr; float* cdf_img = nullptr; size_t cdf_xy_cap = 0, cdf_y_cap = 0, cdf_img_cap = 0; int32_t cdf_res[2] = {0, 0}; bool cdf_valid = false;
	uint32_t n_steps_between_error_map_updates = 128, n_steps_since_error_map_update = 0; bool error_cycle_open = false;
	// host-side deterministic state (no device read-back needed)
	Rng rng, density_grid_rng;
	uint32_t training_step = 0, prep_skip_counter = 0, ema_step = 0;
	// the NEXT occupancy-grid update's samples (generation + sort: they depend on the grid rng, the EMA step and the grid as the last update left it -- on no parameter), drawn on a
	// side stream while the steps in between train; used if nothing they were derived from has changed by then (grid_ahead_matches), dropped otherwise
	struct GridAhead { bool valid = false, in_flight = false, sorted = false; uint32_t n_uniform = 0, n_nonuniform = 0, ema_step = 0; Rng rng; uint64_t state_version = 0; } grid_ahead;
	hipEvent_t ev_grid_free = nullptr, ev_grid_ahead = nullptr; uint32_t grid_ahead_hits = 0;
	TrainCounters* stats_host = nullptr; // pinned: ngp_nerf_get_stats
	uint32_t max_rays = 1u << 18;
	// extra (latent / light-direction) dims, testbed.h Nerf::Training::extra_dims_gpu / extra_dims_opt / rendering_extra_dims: n_extra floats per image (+ one slot behind them
	// for the dims a rendering uses), their gradient, the per-image VarAdamOptimizer state (adam_optimizer.h:27-47; one iteration count: every image steps every time)
	uint32_t n_extra = 0, extra_cap = 0 /* images the buffers hold: every image of the DATASET (ngp_nerf_set_extra_dims), >= n_images = the ones rays are drawn from */; bool optimize_extra_dims = false;
	float* extra_dims = nullptr; float* extra_grad = nullptr; float* extra_m = nullptr; float* extra_v = nullptr; float* dextra = nullptr /* dL/d(extra dims) per batch row */;
	uint32_t* extra_iter = nullptr;   // VarAdamOptimizer::m_iter per image (device): images that join the training set later step from 0 (std::vector<VarAdamOptimizer>, testbed_nerf.cu:2865-2876)
	float extra_lr_last = 1e-4f;      // the learning rate the optimizers' last step used (set_learning_rate before every step, testbed_nerf.cu:2874; VarAdamOptimizer's own default before the first)
	int rendering_extra_view = -1; std::vector<float> rendering_extra; // testbed_nerf.cu:3685-3707: the dims a rendering uses = those of a training view, or explicit values
	std::vector<float> rendering_extra_default; // reset_extra_dims' copy of image 0's INITIAL dims (testbed_nerf.cu:3679-3682): what a rendering uses until told otherwise
	float light_dir_warped[3] = {0, 0, 0}; // (host staging of the copy below: must outlive the asynchronous copy)
	bool has_light_dirs = false; float light_dir[3] = {0.5f, 0.5f, 0.5f}; // Nerf::light_dir (testbed.h; GUI / python settable): replaces the first three rendering dims of a dataset with light directions (:3697-3706)
};


extern "C" int ngp_nerf_create(ngp_model* model, const ngp_nerf_options* o, ngp_aabb aabb, ngp_nerf** out) {
	REQUIRE(model && o && out, "ngp_nerf_create: null argument");
	REQUIRE(o->world_size >= 1 && o->rank < o->world_size, "ngp_nerf_create: bad rank/world_size");
	REQUIRE(o->max_cascade < N_CASCADES, "ngp_nerf_create: max_cascade must be < 8 (NERF_CASCADES)");
	ngp_nerf* t = new ngp_nerf();
	t->model = model; t->opt = *o; t->aabb = aabb; t->n_extra = model->cfg.n_extra_dims;
	t->rng = make_rng(o->seed);                         // testbed.cu:4163
	t->density_grid_rng = make_rng(t->rng.next_uint()); // testbed.cu:4178
	const uint32_t n_cells = GRID_N_CELLS * (o->max_cascade + 1);
	const uint32_t B = o->target_batch_size, max_samples = B * 16;
	if (const char* e = getenv("NGP_K2_ROUNDS")) t->k2_rounds = std::min<uint32_t>(std::max<int>(atoi(e), 1), K2_ROUNDS);
	if (const char* e = getenv("NGP_K2_TILE")) t->k2_tile_w = atoi(e) == 32 ? 32u : atoi(e) == 8 ? 8u : 16u;
	if (t->k2_tile_w == 8) t->k2_rounds = 1;
	t->grid_sample_cap = n_cells;
	if (dev_alloc(&t->density_grid, n_cells) || dev_alloc(&t->density_grid_tmp, n_cells) || dev_alloc(&t->bitfield, GRID_N_CELLS / 8 * N_CASCADES) ||
		dev_alloc(&t->mean, 1) || dev_alloc(&t->mean_partial, 256) || dev_alloc(&t->grid_positions, (size_t)n_cells * 3) || dev_alloc(&t->grid_indices, n_cells) ||
		dev_alloc(&t->grid_positions_sorted, (size_t)n_cells * 3) || dev_alloc(&t->grid_indices_sorted, n_cells) || dev_alloc(&t->grid_sort_temp, t->grid_sort_temp_bytes = grid_sample_sort_temp_bytes(n_cells)) ||
		dev_alloc(&t->grid_mlp_out, n_cells) || dev_alloc(&t->counters, 1) || dev_alloc(&t->ray_indices, t->max_rays) || dev_alloc(&t->rays, t->max_rays) ||
		dev_alloc(&t->numsteps, (size_t)t->max_rays * 2) || dev_alloc(&t->ray_targets, (size_t)t->max_rays * 8) || dev_alloc(&t->k2_enc, (size_t)max_samples * 4) || dev_alloc(&t->src_index, B) || dev_alloc(&t->k2_T, t->max_rays) || dev_alloc(&t->k2_tiles, (size_t)2 * (t->k2_tile_cap = max_samples / 16 + t->max_rays)) || dev_alloc(&t->coords, (size_t)max_samples * (7 + model->cfg.n_extra_dims)) || dev_alloc(&t->mlp_out, (size_t)max_samples * 4) ||
		dev_alloc(&t->coords_compacted, (size_t)B * (7 + model->cfg.n_extra_dims)) || dev_alloc(&t->dloss, (size_t)B * 4) || dev_alloc(&t->sync2, 4) || dev_alloc(&t->bitfield_linear, (size_t)GRID_N_CELLS / 8 * N_CASCADES) || dev_alloc(&t->bitfield_coarse, (size_t)k1_prefilter_words(N_CASCADES)) ||
		dev_alloc(&t->k1_scratch, k1_lattice_scratch_bytes(t->max_rays)) || dev_alloc(&t->k3_scratch, k3_scratch_bytes(t->max_rays))) { delete t; return 1; }
	if (k1_lattice_scratch_init(nullptr, t->k1_scratch, t->max_rays) || k3_scratch_init(nullptr, t->k3_scratch, t->max_rays) || hipDeviceSynchronize() != hipSuccess) { delete t; return fail("k1 scratch init"); }
	HIPCHK(hipMemset(t->density_grid, 0, (size_t)n_cells * 4));
	HIPCHK(hipMemset(t->bitfield, 0, GRID_N_CELLS / 8 * N_CASCADES));
	HIPCHK(hipMemset(t->mean, 0, 4));
	HIPCHK(hipMemset(t->bitfield_linear, 0, (size_t)GRID_N_CELLS / 8 * N_CASCADES));
	HIPCHK(hipMemset(t->bitfield_coarse, 0, (size_t)k1_prefilter_words(N_CASCADES) * 4));
	TrainCounters c; memset(&c, 0, sizeof(c));
	c.rays_per_batch = 1u << 12;   // reset_network, testbed.cu:4171
	c.max_inference = max_samples; // first step: measured_batch_size_before_compaction == 0 (testbed_nerf.cu:3056-3057)
	c.measured_batch_size_before_compaction = max_samples;
	HIPCHK(hipMemcpy(t->counters, &c, sizeof(c), hipMemcpyHostToDevice));
	HIPCHK(hipMemset(t->sync2, 0, 16));
	if (!lazy_streams()) { if (ensure_helper_streams()) { delete t; return 1; } t->k1_stream = g_k1_stream; } // the pre-launched K1's stream (see create_helper_stream)
	*out = t;
	return 0;
}
// runtime-tunable members of Testbed::m_nerf (python_api.cu:714-853): everything except the sharding and batch size
// Anything K1 depends on is about to change through the API: wait for a pre-launched K1 and make sure it is not consumed.
static void invalidate_k1(ngp_nerf* t) {
	if (t->k1_prelaunched && t->k1_stream) (void)hipStreamSynchronize(t->k1_stream);
	++t->state_version;
}
extern "C" int ngp_nerf_set_options(ngp_nerf* t, const ngp_nerf_options* o) {
	REQUIRE(t && o, "set_options: null argument");
	if (memcmp(&t->opt, o, sizeof(*o)) == 0) return 0; // nothing changes: a pre-launched K1 of the next step stays valid (Testbed pushes its options every frame)
	invalidate_k1(t);
	REQUIRE(o->target_batch_size == t->opt.target_batch_size && o->rank == t->opt.rank && o->world_size == t->opt.world_size && o->max_cascade == t->opt.max_cascade,
		"set_options: batch size, max_cascade and sharding are fixed at creation");
	t->opt = *o;
	return 0;
}
extern "C" void ngp_nerf_destroy(ngp_nerf* t) {
	if (!t) return;
	(void)hipDeviceSynchronize();
	if (t->comm) (void)ngp_comm_destroy(t);
	if (t->k1_prelaunched && t->k1_stream) (void)hipStreamSynchronize(t->k1_stream); // (process-wide stream: not destroyed with the trainer)
	if (t->ev_ctl) (void)hipEventDestroy(t->ev_ctl);
	if (t->ev_k1) (void)hipEventDestroy(t->ev_k1);
	if (t->stats_host) (void)hipHostFree(t->stats_host);
	if (t->ev_grid_free) (void)hipEventDestroy(t->ev_grid_free);
	if (t->ev_grid_ahead) (void)hipEventDestroy(t->ev_grid_ahead);
	void* ptrs[] = {t->meta_dev, t->xforms_dev, t->density_grid, t->density_grid_tmp, t->bitfield, t->mean, t->mean_partial, t->grid_positions, t->grid_indices, t->grid_positions_sorted, t->grid_indices_sorted, t->grid_sort_temp,
		t->grid_mlp_out, t->counters, t->ray_indices, t->rays, t->numsteps, t->ray_targets, t->k2_tiles, t->k2_T, t->coords, t->mlp_out, t->coords_compacted, t->dloss, t->sync2, t->bitfield_linear, t->bitfield_coarse, t->k1_scratch, t->k3_scratch, t->r_rays, t->r_masks, t->r_alive, t->r_n_alive, t->r_coords, t->r_out};
	for (void* p : ptrs) if (p) (void)hipFree(p);
	for (void* p : {(void*)t->error_map, (void*)t->cdf_x_cond_y, (void*)t->cdf_y, (void*)t->cdf_img, (void*)t->k2_enc, (void*)t->src_index, (void*)t->extra_dims, (void*)t->extra_grad, (void*)t->extra_m, (void*)t->extra_v, (void*)t->extra_iter, (void*)t->dextra}) if (p) (void)hipFree(p);
	for (void* p : t->owned_pixels) (void)hipFree(p);
	delete t;
}

static size_t pixel_bytes(int type) { return type == NGP_IMAGE_BYTE ? 4 : type == NGP_IMAGE_HALF ? 8 : 16; }

// extra-dims buffers for n images (+ the slot behind them a rendering composes its dims in, testbed_nerf.cu:3658-3660): values, moments and iteration counts of the images already held are kept
static int extra_dims_reserve(ngp_nerf* t, uint32_t n) {
	if (n <= t->extra_cap) return 0;
	float* nd = nullptr; float* ng = nullptr; float* nm = nullptr; float* nv = nullptr; uint32_t* ni = nullptr;
	const size_t cnt = (size_t)(n + 1) * t->n_extra;
	if (dev_alloc(&nd, cnt) || dev_alloc(&ng, cnt) || dev_alloc(&nm, cnt) || dev_alloc(&nv, cnt) || dev_alloc(&ni, (size_t)n + 1)) return 1;
	HIPCHK(hipMemset(nd, 0, cnt * 4)); HIPCHK(hipMemset(ng, 0, cnt * 4)); HIPCHK(hipMemset(nm, 0, cnt * 4)); HIPCHK(hipMemset(nv, 0, cnt * 4)); HIPCHK(hipMemset(ni, 0, ((size_t)n + 1) * 4));
	if (t->extra_dims) {
		HIPCHK(hipDeviceSynchronize());
		const size_t old = (size_t)t->extra_cap * t->n_extra * 4;
		HIPCHK(hipMemcpy(nd, t->extra_dims, old, hipMemcpyDeviceToDevice)); HIPCHK(hipMemcpy(nm, t->extra_m, old, hipMemcpyDeviceToDevice)); HIPCHK(hipMemcpy(nv, t->extra_v, old, hipMemcpyDeviceToDevice));
		HIPCHK(hipMemcpy(ni, t->extra_iter, (size_t)t->extra_cap * 4, hipMemcpyDeviceToDevice));
		for (void* q : {(void*)t->extra_dims, (void*)t->extra_grad, (void*)t->extra_m, (void*)t->extra_v, (void*)t->extra_iter}) (void)hipFree(q);
	}
	t->extra_dims = nd; t->extra_grad = ng; t->extra_m = nm; t->extra_v = nv; t->extra_iter = ni; t->extra_cap = n;
	return 0;
}
static int set_dataset_common(ngp_nerf* t, uint32_t n, const std::vector<ngp_image_meta>& meta, const ngp_xform* xforms) {
	if (t->meta_dev) { (void)hipFree(t->meta_dev); t->meta_dev = nullptr; }
	if (t->xforms_dev) { (void)hipFree(t->xforms_dev); t->xforms_dev = nullptr; }
	if (dev_alloc(&t->meta_dev, n) || dev_alloc(&t->xforms_dev, n)) return 1;
	HIPCHK(hipMemcpy(t->meta_dev, meta.data(), n * sizeof(ngp_image_meta), hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(t->xforms_dev, xforms, n * sizeof(ngp_xform), hipMemcpyHostToDevice));
	// k1_setup's small instance serves datasets like BASELINE.json's: 8-bit pixels, Perspective / OpenCV lenses, still cameras (the general one every other dataset)
	t->plain_dataset = true;
	for (uint32_t i = 0; i < n && t->plain_dataset; ++i) {
		const ngp_image_meta& mi = meta[i];
		bool plain = mi.image_data_type == NGP_IMAGE_BYTE && (mi.lens_mode == NGP_LENS_PERSPECTIVE || mi.lens_mode == NGP_LENS_OPENCV);
		for (int k = 0; k < 4; ++k) plain = plain && mi.rolling_shutter[k] == 0.f;
		for (int k = 0; k < 12; ++k) plain = plain && xforms[i].start[k] == xforms[i].end[k];
		t->plain_dataset = plain;
	}
	if (n != t->n_images) { // the error map and the three CDFs are laid out per image: an open accumulation cycle and installed CDFs end with the old image count
		t->cdf_valid = false; t->error_cycle_open = false; t->n_steps_since_error_map_update = 0; // (the next step opens a cycle sized for n; dev_grow regrows the buffers)
	}
	t->n_images = n;
	if (t->n_extra && extra_dims_reserve(t, n)) return 1; // the extra dims follow the image count: existing values are kept, new images start at zero (ngp_nerf_set_extra_dims installs the reference's initial values for the whole dataset)
	return 0;
}
extern "C" int ngp_nerf_set_dataset_host(ngp_nerf* t, uint32_t n, const ngp_image_meta* meta, const ngp_xform* xforms, const void* const* pixels_host) {
	invalidate_k1(t);
	REQUIRE(n > 0 && meta && xforms && pixels_host, "set_dataset: null/empty");
	for (void* p : t->owned_pixels) (void)hipFree(p);
	t->owned_pixels.clear();
	std::vector<ngp_image_meta> m(meta, meta + n);
	for (uint32_t i = 0; i < n; ++i) {
		REQUIRE(m[i].lens_mode >= NGP_LENS_PERSPECTIVE && m[i].lens_mode <= NGP_LENS_ORTHOGRAPHIC, "set_dataset: unknown lens mode");
		const size_t bytes = (size_t)m[i].resolution[0] * m[i].resolution[1] * pixel_bytes(m[i].image_data_type);
		void* d = nullptr;
		HIPCHK(hipMalloc(&d, bytes));
		t->owned_pixels.push_back(d);
		HIPCHK(hipMemcpy(d, pixels_host[i], bytes, hipMemcpyHostToDevice));
		m[i].pixels = d;
		if (m[i].depth) { // host pointer to one float per pixel (nerf_loader.cu copy_depth: integer depth * depth_scale): uploaded like the pixels
			const size_t db = (size_t)m[i].resolution[0] * m[i].resolution[1] * sizeof(float);
			void* dd = nullptr;
			HIPCHK(hipMalloc(&dd, db));
			t->owned_pixels.push_back(dd);
			HIPCHK(hipMemcpy(dd, m[i].depth, db, hipMemcpyHostToDevice));
			m[i].depth = (const float*)dd;
		}
	}
	return set_dataset_common(t, n, m, xforms);
}
extern "C" int ngp_nerf_set_dataset_device(ngp_nerf* t, uint32_t n, const ngp_image_meta* meta, const ngp_xform* xforms) {
	invalidate_k1(t);
	REQUIRE(n > 0 && meta && xforms, "set_dataset: null/empty");
	std::vector<ngp_image_meta> m(meta, meta + n);
	for (uint32_t i = 0; i < n; ++i)
		REQUIRE(m[i].lens_mode >= NGP_LENS_PERSPECTIVE && m[i].lens_mode <= NGP_LENS_ORTHOGRAPHIC, "set_dataset: unknown lens mode");
	return set_dataset_common(t, n, m, xforms);
}

// the samples of one occupancy-grid update (testbed_nerf.cu:2525-2557): uniform + non-uniform generation from `rng` (not advanced here) and their sort into (cascade, Morton block)
// order; generate = false: they lie in the buffers already (drawn ahead), only the pointers are returned
static int grid_update_samples(ngp_nerf* t, hipStream_t s, const Rng* rng, uint32_t n_uniform, uint32_t n_nonuniform, bool sorted, bool generate, const float** eval_pos, const uint32_t** eval_idx) {
	const uint32_t n_samples = n_uniform + n_nonuniform;
	if (generate) {
		ProfScope ps(P_GRID_MISC, s);
		Rng r = *rng;
		launch_generate_grid_samples(s, n_uniform, pod(r), nullptr, t->ema_step, t->aabb, t->density_grid, t->grid_positions, t->grid_indices,
			t->opt.max_cascade + 1, -0.01f);
		r.advance(1ull << 32);
		launch_generate_grid_samples(s, n_nonuniform, pod(r), nullptr, t->ema_step, t->aabb, t->density_grid, t->grid_positions + (size_t)n_uniform * 3,
			t->grid_indices + n_uniform, t->opt.max_cascade + 1, MIN_OPTICAL_THICKNESS);
	}
	// evaluation order = (cascade, Morton cell) order: spatially coherent like the samples of a ray (sort_util.hip); the result is order independent
	*eval_pos = t->grid_positions; *eval_idx = t->grid_indices;
	if (sorted) {
		if (generate) {
			ProfScope ps(P_GRID_MISC, s);
			uint32_t key_bits = 21; for (uint32_t c = t->opt.max_cascade; c; c >>= 1) ++key_bits; // 3 x 7 Morton bits + the cascade
			// the low 6 Morton bits (the cell inside its 4x4x4 block) stay unsorted: one radix pass less, the same coherence for the hash-grid levels
			constexpr uint32_t begin_bit = 6u; // Morton blocks of 4 x 4 x 4 cells (coarser blocks = fewer radix passes, less coherence: measured in round 4)
			if (grid_sample_sort(s, t->grid_sort_temp, t->grid_sort_temp_bytes, t->grid_indices, t->grid_indices_sorted, t->grid_positions, t->grid_positions_sorted, n_samples, std::min(begin_bit, key_bits - 1u), key_bits))
				return fail("update_density_grid: sort failed");
		}
		*eval_pos = t->grid_positions_sorted; *eval_idx = t->grid_indices_sorted;
	}
	return 0;
}
// update_density_grid_nerf, testbed_nerf.cu:2476-2592
extern "C" int ngp_nerf_update_density_grid(ngp_nerf* t, void* stream, float decay, uint32_t n_uniform, uint32_t n_nonuniform) {
	FlagScope flag_scope_(t->dbg);
	const uint64_t version_on_entry = t->state_version;
	invalidate_k1(t);
	REQUIRE(t->n_images > 0, "update_density_grid: no dataset");
	hipStream_t s = (hipStream_t)stream;
	bool marked = false;
	if (t->grid_ahead.in_flight) { HIPCHK(hipStreamWaitEvent(s, t->ev_grid_ahead, 0)); t->grid_ahead.in_flight = false; } // used or not: the sample buffers (and the grid it reads) are the side stream's until then
	const uint32_t n_elements = GRID_N_CELLS * (t->opt.max_cascade + 1);
	const uint32_t n_samples = n_uniform + n_nonuniform;
	REQUIRE(n_samples <= t->grid_sample_cap, "update_density_grid: too many samples");
	if (t->training_step == 0 || t->n_images != t->n_images_marked) { // testbed_nerf.cu:2500-2517: at the first step, and again whenever the number of training images changed
		t->n_images_marked = t->n_images;                             // (a streaming client raising n_images_for_training): cells only the new cameras see become trainable
		if (t->training_step == 0) t->ema_step = 0;
		launch_mark_untrained(s, n_elements, t->density_grid, t->n_images, t->meta_dev, t->xforms_dev, t->training_step == 0 ? 1 : 0);
		marked = true; // the grid the samples ahead were thresholded against has changed
	}
	const bool want_sorted = !(g_debug_flags & DBG_GRID_NO_SORT);
	ngp_nerf::GridAhead& ga = t->grid_ahead;
	const bool ahead_hit = ga.valid && !marked && ga.state_version == version_on_entry && ga.n_uniform == n_uniform && ga.n_nonuniform == n_nonuniform && ga.ema_step == t->ema_step &&
		ga.rng.state == t->density_grid_rng.state && ga.rng.inc == t->density_grid_rng.inc && ga.sorted == want_sorted;
	ga.valid = false;
	if (ahead_hit) ++t->grid_ahead_hits;
	{ ProfScope ps(P_GRID_MISC, s);
	HIPCHK(hipMemsetAsync(t->density_grid_tmp, 0, (size_t)n_elements * 4, s)); }
	const float* eval_pos = nullptr; const uint32_t* eval_idx = nullptr;
	if (grid_update_samples(t, s, ahead_hit ? nullptr : &t->density_grid_rng, n_uniform, n_nonuniform, want_sorted, !ahead_hit, &eval_pos, &eval_idx)) return 1;
	t->density_grid_rng.advance(1ull << 32); t->density_grid_rng.advance(1ull << 32); // (one advance per generation launch, testbed_nerf.cu:2541, 2557)
	// NerfNetwork::density with the TRAINING params (use_inference_params = false, testbed_nerf.cu:2570)
	{ ProfScope ps(P_GRID_DENSITY, s);
	  launch_inference(s, t->model->gm_dev, model_ptrs(t->model, false), eval_pos, 3, n_samples, nullptr, t->grid_mlp_out, 1, true, 0, t->model->gm.F); }
	ProfScope ps2(P_GRID_MISC, s);
	launch_splat_grid_samples(s, n_samples, eval_idx, t->grid_mlp_out, 1, t->density_grid_tmp, t->opt.density_activation);
	launch_ema_grid_samples(s, n_elements, decay, t->density_grid, t->density_grid_tmp);
	++t->ema_step;
	launch_grid_mean(s, t->density_grid, t->mean_partial, t->mean);
	launch_grid_to_bitfield(s, t->density_grid, t->opt.max_cascade, t->bitfield, t->mean);
	launch_build_linear_bitfield(s, t->bitfield, t->bitfield_linear, N_CASCADES, t->bitfield_coarse); // all pooled levels: the march may ask for one above max_cascade (mip_from_dt)
	HIPCHK(hipGetLastError());
	// The next update's samples, now: everything they depend on is final (the grid after this EMA, the grid rng, the EMA step).  Steady state only (the sample counts change at
	// step 256), not under the per-kernel profile (its events time the caller's stream).
	// They run on the communication stream, which a trainer without communicator leaves idle (with one, its collectives must not queue behind them: drawn inside the update then).
	if (!(g_debug_flags2 & DBG2_GRID_NO_AHEAD) && !g_prof_on && t->training_step >= 256 + 16 && n_nonuniform > 0 && !t->comm && t->opt.world_size <= 1) {
		hipStream_t g_grid_stream = nullptr;
		if (create_helper_stream(&g_comm_stream, true)) return 1;
		g_grid_stream = g_comm_stream;
		if (!t->ev_grid_free) { HIPCHK(hipEventCreateWithFlags(&t->ev_grid_free, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&t->ev_grid_ahead, hipEventDisableTiming)); }
		HIPCHK(hipEventRecord(t->ev_grid_free, s)); HIPCHK(hipStreamWaitEvent(g_grid_stream, t->ev_grid_free, 0));
		Rng r = t->density_grid_rng;
		const float* p_; const uint32_t* i_;
		if (grid_update_samples(t, g_grid_stream, &r, n_uniform, n_nonuniform, want_sorted, true, &p_, &i_)) return 1;
		HIPCHK(hipEventRecord(t->ev_grid_ahead, g_grid_stream));
		ga.valid = true; ga.in_flight = true; ga.sorted = want_sorted; ga.n_uniform = n_uniform; ga.n_nonuniform = n_nonuniform; ga.ema_step = t->ema_step; ga.rng = t->density_grid_rng;
		ga.state_version = t->state_version;
	}
	return 0;
}

// training_prep_nerf (testbed_nerf.cu:3385-3398) at the cadence of Testbed::train (testbed.cu:4596-4613)
extern "C" int ngp_nerf_train_prep(ngp_nerf* t, void* stream) {
	const uint32_t n_prep_to_skip = (uint32_t)std::min(std::max((int)t->training_step / 16, 1), 16);
	int rc = 0;
	if (t->prep_skip_counter % n_prep_to_skip == 0) {
		const uint32_t n_cascades = t->opt.max_cascade + 1;
		if (t->training_step < 256) rc = ngp_nerf_update_density_grid(t, stream, t->opt.density_grid_decay, GRID_N_CELLS * n_cascades, 0);
		else rc = ngp_nerf_update_density_grid(t, stream, t->opt.density_grid_decay, GRID_N_CELLS / 4 * n_cascades, GRID_N_CELLS / 4 * n_cascades);
	}
	++t->prep_skip_counter;
	return rc;
}

// train_nerf_step, testbed_nerf.cu:3007-3382 (Nerf train mode: K1, K2, K3, K4, K5)
// will the NEXT ngp_nerf_train_prep update the occupancy grid? (same arithmetic, one step ahead: training_step is incremented by
// ngp_nerf_train_finish, prep_skip_counter was already incremented by this step's prep)
static bool next_prep_updates_grid(const ngp_nerf* t) {
	const uint32_t n_prep_to_skip = (uint32_t)std::min(std::max((int)(t->training_step + 1) / 16, 1), 16);
	return t->prep_skip_counter % n_prep_to_skip == 0;
}
__global__ void k_import_sync(TrainCounters* c, const uint32_t* sync2, uint32_t world_size);
// phase bit 1: K1..K4 (forward, loss, compaction); bit 2: controller, T1 / W / scatter (backward).  `global_counters`: the caller has
// all-reduced the published counters (multi-rank), so the controller may run before the backward pass.
// ---- error map and its CDFs (testbed_nerf.cu:2753-2759, 2791-2855) ----
static bool error_map_wanted(const ngp_nerf* t) { return t->opt.accumulate_error_map || t->opt.sample_focal_plane_proportional_to_error || t->opt.sample_image_proportional_to_error; }
static ErrorCdf error_cdf_args(const ngp_nerf* t) {
	ErrorCdf c;
	if (!t->cdf_valid) return c;
	if (t->opt.sample_focal_plane_proportional_to_error) { c.x_cond_y = t->cdf_x_cond_y; c.y = t->cdf_y; }
	if (t->opt.sample_image_proportional_to_error) c.img = t->cdf_img;
	c.res[0] = t->cdf_res[0]; c.res[1] = t->cdf_res[1];
	return c;
}
template <typename T> static int dev_grow(T** p, size_t* cap, size_t n) {
	if (n <= *cap) return 0;
	if (*p) HIPCHK(hipFree(*p));
	*p = nullptr; *cap = 0;
	if (dev_alloc(p, n)) return 1;
	*cap = n;
	return 0;
}
// Start of an accumulation cycle (testbed_nerf.cu:2753-2759): the map's resolution follows the number of rays one image receives until the next CDF
// update; the reference reads rays_per_batch from its host-side counters, here it lives on the device (one 4-byte read-back per cycle, >= 128 steps).
static int error_map_begin_cycle(ngp_nerf* t, hipStream_t s, const ngp_image_meta& meta0) {
	uint32_t rays_per_batch = 0;
	HIPCHK(hipMemcpyAsync(&rays_per_batch, &t->counters->rays_per_batch, 4, hipMemcpyDeviceToHost, s));
	HIPCHK(hipStreamSynchronize(s));
	const uint32_t n_samples_per_image = (t->n_steps_between_error_map_updates * rays_per_batch) / t->n_images; // uint32 arithmetic, as in the reference
	const int r = (int)(std::sqrt(std::sqrt((float)n_samples_per_image)) * 3.5f);
	t->error_map_res[0] = std::min(r, meta0.resolution[0]); t->error_map_res[1] = std::min(r, meta0.resolution[1]);
	REQUIRE(t->error_map_res[0] >= 2 && t->error_map_res[1] >= 2, "error map: fewer than 2 x 2 cells (too few rays per image)");
	const size_t n = (size_t)t->error_map_res[0] * t->error_map_res[1] * t->n_images;
	if (dev_grow(&t->error_map, &t->error_map_cap, n)) return 1;
	HIPCHK(hipMemsetAsync(t->error_map, 0, n * 4, s));
	t->error_cycle_open = true;
	return 0;
}
static int error_map_allreduce(ngp_nerf* t, hipStream_t s, size_t n); // data-parallel: every rank has splatted its own rays only
// End of a cycle (testbed_nerf.cu:2795-2855): CDFs from the map, update interval x 1.5
static int error_map_build_cdfs(ngp_nerf* t, hipStream_t s) {
	if (t->k1_prelaunched && t->k1_stream) HIPCHK(hipStreamSynchronize(t->k1_stream)); // a pre-launched K1 may still be reading the old CDFs; it is discarded below
	++t->state_version;
	const int32_t w = t->error_map_res[0], h = t->error_map_res[1];
	const size_t n = (size_t)w * h * t->n_images;
	if ((t->opt.world_size > 1 || t->comm) && error_map_allreduce(t, s, n)) return 1; // (a one-rank communicator: the identity, exercised by tests/test_gpu_dist.py)
	if (dev_grow(&t->cdf_x_cond_y, &t->cdf_xy_cap, n) || dev_grow(&t->cdf_y, &t->cdf_y_cap, (size_t)h * t->n_images)) return 1;
	if (dev_grow(&t->cdf_img, &t->cdf_img_cap, (size_t)t->n_images)) return 1;
	t->cdf_res[0] = w; t->cdf_res[1] = h;
	launch_construct_error_cdfs(s, t->n_images, (uint32_t)w, (uint32_t)h, t->error_map, t->cdf_x_cond_y, t->cdf_y, t->cdf_img);
	t->n_steps_since_error_map_update = 0;
	t->cdf_valid = true; t->error_cycle_open = false;
	t->n_steps_between_error_map_updates = (uint32_t)(t->n_steps_between_error_map_updates * 1.5f);
	return 0;
}

static int nerf_step_impl(ngp_nerf* t, void* stream, int phase, bool global_counters, bool optimizer_follows = false) {
	FlagScope flag_scope_(t->dbg);
	REQUIRE(t->n_images > 0, "train: no dataset");
	hipStream_t s = (hipStream_t)stream;
	const ngp_nerf_options& o = t->opt;
	const uint32_t B = o.target_batch_size, max_samples = B * 16;
	TrainCounters* c = t->counters;
	const bool lattice = !(g_debug_flags & DBG_K1_REFERENCE_LAYOUT);
	auto make_k1 = [&]() {
		K1Args k1;
		k1.n_rays = 0; k1.n_rays_ptr = &c->rays_per_batch; k1.rank = o.rank; k1.world_size = o.world_size; k1.aabb = t->aabb;
		k1.max_samples = max_samples; k1.max_samples_ptr = &c->max_inference; k1.rng = pod(t->rng);
		k1.n_mips = N_CASCADES; k1.bitfield_linear = t->bitfield_linear; k1.bitfield_coarse = (g_debug_flags & DBG_K1_NO_PREFILTER) ? nullptr : t->bitfield_coarse; k1.chunk_march = (g_debug_flags & DBG_K1_CHUNK_MARCH) != 0; k1.no_first_point_skip = (g_debug_flags & DBG_K1_NO_FIRST_POINT_SKIP) != 0; k1.k2_tiles0_out = lattice ? t->k2_tiles : nullptr; k1.k2_tile_w = t->k2_tile_w;
		k1.ray_targets_out = lattice ? t->ray_targets : nullptr; for (int k = 0; k < 3; ++k) k1.background_color[k] = o.background_color[k];
		k1.color_space_srgb = o.color_space_srgb; k1.random_bg_color = o.random_bg_color; k1.linear_colors = o.linear_colors;
		k1.ray_counter = &c->ray_counter; k1.numsteps_counter = &c->numsteps_counter; k1.ray_indices_out = t->ray_indices; k1.rays_out = t->rays;
		k1.numsteps_out = t->numsteps; k1.coords_out = t->coords; k1.n_images = t->n_images; k1.metadata = t->meta_dev; k1.xforms = t->xforms_dev;
		k1.bitfield = t->bitfield; k1.max_mip = o.max_cascade; k1.snap_to_pixel_centers = o.snap_to_pixel_centers; k1.cone_angle_constant = o.cone_angle_constant;
		k1.exact_skip = !(g_debug_flags & DBG_K1_INDEPENDENT_LATTICE); k1.clamp_min_max = (g_debug_flags & DBG_K1_MIP_CLAMP_MIN_MAX) ? 1u : 0u;
		k1.depth_lambda = o.depth_supervision_lambda;
		k1.cdf = error_cdf_args(t);
		k1.extra_dims = t->extra_dims; k1.n_extra = t->n_extra;
		k1.plain_dataset = t->plain_dataset ? 1 : 0;
		return k1;
	};
	if (phase & 1) {
	if (error_map_wanted(t) && !t->error_cycle_open) {
		REQUIRE(t->n_steps_since_error_map_update == 0, "error map: cycle state");
		ngp_image_meta meta0; HIPCHK(hipMemcpy(&meta0, t->meta_dev, sizeof(meta0), hipMemcpyDeviceToHost));
		if (error_map_begin_cycle(t, s, meta0)) return 1;
	}
	bool have_k1 = false;
	if (t->k1_prelaunched) { // launched by the previous step: valid if nothing it depends on was changed through the API since
		HIPCHK(hipStreamWaitEvent(s, t->ev_k1, 0));
		have_k1 = t->k1_version == t->state_version && t->k1_for_stream == s && lattice;
		t->k1_prelaunched = false; // a stale one is simply overwritten: the lattice K1 writes (never accumulates) its counters
	}
	if (!have_k1) {
		ProfScope ps(P_K1, s);
		const K1Args k1 = make_k1();
		if (g_debug_flags & DBG_K1_REFERENCE_LAYOUT) launch_generate_training_samples(s, k1, t->max_rays);
		else launch_generate_training_samples_lattice(s, k1, t->max_rays, t->k1_scratch);
	}
	const uint32_t cs = 7 + t->n_extra; // floats per NerfCoordinate
	const bool lazy_k2 = lattice && !(g_debug_flags & DBG_K2_EAGER) && !t->n_extra; // the round-0 tile list comes from the lattice K1 (models with extra dims: every sample, like the reference)
	if (!lazy_k2) { ProfScope ps(P_COUNTERS, s); launch_clamp_compacted(s, c, B); } // n_inference for the eager K2
	{ ProfScope ps(P_K2_INFERENCE, s);
	  if (lazy_k2) {
		K2LazyArgs la;
		la.n_rays_ptr = &c->ray_counter; la.tiles[0] = t->k2_tiles; la.tiles[1] = t->k2_tiles + t->k2_tile_cap; la.tile_cap = t->k2_tile_cap;
		la.n_tiles_ptr = c->k2_tiles; la.T_run = t->k2_T; la.round = 0; la.n_rounds = t->k2_rounds; la.tile_w = t->k2_tile_w; la.n_eval_ptr = &c->k2_samples; la.density_activation = o.density_activation;
		{ const float max_stepsize = MIN_CONE_STEP * (float)(1 << (N_CASCADES - 1)); la.dt_unwarp_scale = max_stepsize - MIN_CONE_STEP; la.dt_unwarp_offset = MIN_CONE_STEP; } // unwarp_dt
		// T1 re-uses the encodings of the samples it differentiates (base.json's shape, production K3 kernels): 64 coalesced bytes per sample instead of 64 table gathers
		const bool two_pass_k3 = (g_debug_flags & DBG_K3_TWO_PASS) && !(o.depth_supervision_lambda > 0.f) && !error_map_wanted(t) && !t->error_cycle_open;
		t->k2_enc_valid = t->model->gm.F == 4 && t->model->cfg.n_hidden_layers_rgb == 2 && !(g_debug_flags & DBG_T1_NO_K2_STASH) && !two_pass_k3;
		la.enc_out = t->k2_enc_valid ? t->k2_enc : nullptr;
		launch_inference_lazy(s, t->model->gm_dev, model_ptrs(t->model, false), t->coords, 7, t->max_rays, max_samples, t->mlp_out, 4, 4, la, t->model->gm.F);
	  } else {
	  t->k2_enc_valid = false; // eager K2 (ablation): T1 gathers
	  launch_inference(s, t->model->gm_dev, model_ptrs(t->model, false), t->coords, cs, max_samples, &c->n_inference, t->mlp_out, 4, false, 4, t->model->gm.F); } }
	K3Args k3;
	k3.n_rays = 0; k3.n_rays_ptr = &c->rays_per_batch; k3.aabb = t->aabb; k3.rng = pod(t->rng); k3.max_samples_compacted = B; k3.rays_counter = &c->ray_counter;
	k3.loss_scale = o.loss_scale; for (int k = 0; k < 3; ++k) k3.background_color[k] = o.background_color[k];
	k3.color_space_srgb = o.color_space_srgb; k3.random_bg_color = o.random_bg_color; k3.linear_colors = o.linear_colors; k3.n_images = t->n_images; k3.metadata = t->meta_dev;
	k3.network_output = t->mlp_out; k3.output_stride = 4; k3.numsteps_counter_compacted = &c->numsteps_counter_compacted; k3.ray_indices_in = t->ray_indices;
	k3.rays_in = t->rays; k3.numsteps_inout = t->numsteps; k3.coords_in = t->coords; k3.coords_out = t->coords_compacted; k3.dloss_doutput = t->dloss; k3.dloss_stride = 4;
	k3.loss_type = o.loss_type; k3.loss_output = &c->loss_sum; k3.rgb_activation = o.rgb_activation; k3.density_activation = o.density_activation;
	k3.snap_to_pixel_centers = o.snap_to_pixel_centers; k3.mean_density_ptr = t->mean; k3.near_distance = o.near_distance;
	k3.ray_targets = lattice ? t->ray_targets : nullptr; k3.train_mode = o.train_mode; k3.k3_scratch = t->k3_scratch;
	k3.depth_lambda = o.depth_supervision_lambda; k3.depth_loss_type = o.depth_loss_type;
	if (k3.depth_lambda > 0.f) k3.k3_scratch = nullptr; // the two-pass ablation kernel has no depth term: the one-pass kernel runs
	k3.src_index_out = t->k2_enc_valid ? t->src_index : nullptr;
	k3.cstride = cs;
	k3.cdf = error_cdf_args(t);
	if (t->error_cycle_open) { k3.error_map = t->error_map; k3.error_map_res[0] = t->error_map_res[0]; k3.error_map_res[1] = t->error_map_res[1]; }
	if (k3.error_map || k3.cdf.x_cond_y || k3.cdf.img) k3.k3_scratch = nullptr; // (nor the error map)
	{ ProfScope ps(P_K3, s); launch_compute_loss(s, k3, t->max_rays); }
	// K4 clamps K3's counter itself and publishes {marched, compacted} for the cross-rank all-reduce (8e)
	// single rank, forward and backward in one call: the controller rides on K4's last workgroup (no launch of its own)
	const bool fuse_ctl = (phase & 2) && o.world_size == 1 && !(g_debug_flags & DBG_SEPARATE_CONTROLLER);
	{ ProfScope ps(P_K4, s); launch_fill_rollover(s, B, &c->numsteps_counter_compacted, t->coords_compacted, cs, t->dloss, 4, &c->numsteps_counter, t->sync2, fuse_ctl ? c : nullptr, o.world_size, &c->loss_sum); }
	if (fuse_ctl) t->ctl_done = true;
	}
	if (phase & 2) {
	// The batch-size controller only needs K1's / K3's counters (multi-rank: their all-reduced values), so it runs here instead of after
	// the optimizer and the next step's K1 can start behind it.
	const bool early_ctl = o.world_size == 1 || global_counters;
	if (early_ctl && !t->ctl_done) {
		if (o.world_size > 1) hipLaunchKernelGGL(k_import_sync, dim3(1), dim3(64), 0, s, t->counters, t->sync2, o.world_size);
		ProfScope ps(P_COUNTERS, s); launch_update_counters(s, t->counters, B, o.world_size);
		t->ctl_done = true;
	}
	const bool train_extra = t->n_extra && t->optimize_extra_dims; // (testbed_nerf.cu:2743)
	const bool prelaunch = early_ctl && lattice && !g_prof_on && !(g_debug_flags & DBG_NO_STREAM_OVERLAP) && !next_prep_updates_grid(t) && !train_extra; // (the next K1 copies the extra dims this step's optimizer is about to change)
	if (prelaunch) {
		if (!t->k1_stream) { if (create_helper_stream(&g_k1_stream, true)) return 1; t->k1_stream = g_k1_stream; } // highest priority: K1's small latency-bound kernels slip in between the backward pass's workgroups
		if (!t->ev_ctl) { HIPCHK(hipEventCreateWithFlags(&t->ev_ctl, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&t->ev_k1, hipEventDisableTiming)); }
		HIPCHK(hipEventRecord(t->ev_ctl, s));
	}
	EncStashIn stash_in;
	if (t->k2_enc_valid) { stash_in.enc = t->k2_enc; stash_in.src_index = t->src_index; stash_in.n_valid_ptr = t->sync2 + 3; }
	if (train_extra) {
		REQUIRE(o.world_size == 1, "optimize_extra_dims: single rank only (the per-image gradients are not all-reduced)");
		if (!t->dextra && dev_alloc(&t->dextra, (size_t)B * t->n_extra)) return 1;
		t->model->dextra_out = t->dextra;
	}
	const int tr = model_training_step_impl(t->model, stream, t->coords_compacted, 7 + t->n_extra, B, t->dloss, 4, t->k2_enc_valid ? &stash_in : nullptr, optimizer_follows && o.world_size == 1 && !t->comm ? o.loss_scale : 0.f);
	t->model->dextra_out = nullptr;
	if (tr) return 1;
	if (train_extra) { // testbed_nerf.cu:2745-2750 (clear) + :3325-3340 (compute_extra_dims_gradient_train_nerf over the compacted rays)
		HIPCHK(hipMemsetAsync(t->extra_grad, 0, (size_t)t->n_images * t->n_extra * 4, s));
		// (the controller has closed the step by now -- single rank: it rides on K4 or runs right behind it -- and keeps the step's ray counts in *_last)
		launch_extra_dims_gradient(s, t->max_rays, &c->rays_per_batch_last, &c->n_rays_last, t->extra_grad, t->n_extra, t->n_images, t->ray_indices, t->numsteps, t->dextra, B, error_cdf_args(t).img);
	}
	t->rng.advance(1ull << 32); // m_rng.advance(), testbed_nerf.cu:3377
	if (prelaunch) { // K1 of the NEXT step (its rng position), concurrent with this step's backward pass and optimizer
		HIPCHK(hipStreamWaitEvent(t->k1_stream, t->ev_ctl, 0));
		launch_generate_training_samples_lattice(t->k1_stream, make_k1(), t->max_rays, t->k1_scratch);
		HIPCHK(hipEventRecord(t->ev_k1, t->k1_stream));
		t->k1_prelaunched = true; t->k1_version = t->state_version; t->k1_for_stream = s;
	}
	}
	HIPCHK(hipGetLastError());
	return 0;
}

extern "C" int ngp_nerf_train_forward_backward(ngp_nerf* t, void* stream) { return nerf_step_impl(t, stream, 3, false); }
// Multi-rank order: ngp_nerf_train_forward -> all-reduce(sum) of the two ngp_nerf_counter_ptrs words -> ngp_nerf_train_backward ->
// all-reduce(sum) of the gradients -> ngp_nerf_train_finish.  The controller then runs before the backward pass and the next
// step's K1 overlaps the backward pass AND the gradient all-reduce.
extern "C" int ngp_nerf_train_forward(ngp_nerf* t, void* stream) { return nerf_step_impl(t, stream, 1, false); }
extern "C" int ngp_nerf_train_backward(ngp_nerf* t, void* stream) { return nerf_step_impl(t, stream, 2, true); }

__global__ void k_import_sync(TrainCounters* c, const uint32_t* sync2, uint32_t world_size) {
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	// after an all-reduce(sum) sync2 holds the GLOBAL counts; the controller works on per-rank averages
	const uint32_t avg_before = (sync2[0] + world_size - 1) / world_size;
	c->numsteps_counter = avg_before + avg_before / 8; // slack: K1's cap must cover ranks above the average
	c->numsteps_counter_compacted = (sync2[1] + world_size - 1) / world_size;
	// K3 normalises every ray's loss by the GLOBAL ray count, so the sum over ranks is the loss of the union batch (Testbed.loss under data parallelism)
	c->loss_sum = (float)sync2[2] * (1.0f / 16777216.f);
}

// optimizer_step + NerfCounters::update_after_training, testbed_nerf.cu:2770-2778
static int nerf_finish_impl(ngp_nerf* t, void* stream, bool run_optimizer);
extern "C" int ngp_nerf_train_finish(ngp_nerf* t, void* stream) { return nerf_finish_impl(t, stream, true); }
static int nerf_finish_impl(ngp_nerf* t, void* stream, bool run_optimizer) {
	hipStream_t s = (hipStream_t)stream;
	if (t->grads_pending) { HIPCHK(hipStreamWaitEvent(s, t->ev_red_a, 0)); HIPCHK(hipStreamWaitEvent(s, t->ev_red_b, 0)); t->grads_pending = false; }
	if (run_optimizer && ngp_model_optimizer_step(t->model, stream, t->opt.loss_scale)) return 1;
	if (t->n_extra && t->optimize_extra_dims) // testbed_nerf.cu:2860-2878: one VarAdamOptimizer step per image at the network optimizer's current learning rate
		{ launch_extra_dims_adam(s, t->n_images * t->n_extra, t->extra_dims, t->extra_grad, t->extra_m, t->extra_v, 0, t->model->lr, t->opt.loss_scale, t->extra_iter, t->n_extra); t->extra_lr_last = t->model->lr; }
	if (!t->ctl_done) { // the controller has not run behind K3 (multi-rank caller using ngp_nerf_train_forward_backward)
		if (t->opt.world_size > 1) hipLaunchKernelGGL(k_import_sync, dim3(1), dim3(64), 0, s, t->counters, t->sync2, t->opt.world_size);
		ProfScope ps(P_COUNTERS, s); launch_update_counters(s, t->counters, t->opt.target_batch_size, t->opt.world_size);
	}
	t->ctl_done = false;
	++t->training_step;
	if (t->error_cycle_open) { // testbed_nerf.cu:2791-2855
		if (++t->n_steps_since_error_map_update >= t->n_steps_between_error_map_updates && error_map_build_cdfs(t, s)) return 1;
	}
	HIPCHK(hipGetLastError());
	return 0;
}

// ------------------------------------------------------------------------------------------------
// data-parallel step inside the library: RCCL over xGMI, one process per GPU.  librccl is resolved at run time
// (dlopen; the copy a host process has already loaded -- e.g. PyTorch's -- is reused), so libngp_hip.so has no link-time dependency.
// ------------------------------------------------------------------------------------------------
struct NcclId { char internal[128]; };
struct RcclApi {
	void* lib = nullptr;
	int (*GetUniqueId)(NcclId*) = nullptr;
	int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
	int (*CommDestroy)(void*) = nullptr;
	int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
	int (*ReduceScatter)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr; // (sendbuff, recvbuff, recvcount, datatype, op, comm, stream)
	int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;           // (sendbuff, recvbuff, sendcount, datatype, comm, stream)
	int (*GroupStart)() = nullptr; int (*GroupEnd)() = nullptr;
	const char* (*GetErrorString)(int) = nullptr;
};
static RcclApi g_rccl;
static int rccl_load() {
	if (g_rccl.lib) return 0;
	void* h = nullptr;
	for (const char* name : {"librccl.so.1", "librccl.so"}) if ((h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break; // already in the process
	if (!h) for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
	if (!h) return fail(std::string("librccl not found: ") + dlerror());
	g_rccl.GetUniqueId = (int (*)(NcclId*))dlsym(h, "ncclGetUniqueId");
	g_rccl.CommInitRank = (int (*)(void**, int, NcclId, int))dlsym(h, "ncclCommInitRank");
	g_rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
	g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclAllReduce");
	g_rccl.ReduceScatter = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclReduceScatter");
	g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
	g_rccl.GroupStart = (int (*)())dlsym(h, "ncclGroupStart"); g_rccl.GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
	g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
	if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce) return fail("librccl: missing symbols");
	g_rccl.lib = h;
	return 0;
}
#define RCCLCHK(x) do { int r_ = (x); if (r_ != 0) return fail(std::string(#x) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error")); } while (0)
constexpr int kNcclUint32 = 3, kNcclHalf = 6, kNcclSum = 0; // ncclDataType_t / ncclRedOp_t values of rccl.h

// ---- sharded data-parallel step (round 5) ------------------------------------------------------------------------------------------------------------------
// Rounds 2-4: all-reduce(fp16 gradients) -> the full optimizer sweep replicated on every rank (38 B x P of memory traffic per rank and step, 63 us, and one 23.4 MB ring
// that starts behind the last kernel of the backward pass).  Now: the table is cut into two buckets of whole levels (A = the coarse half of the entries, B = the rest);
// bucket A's reduce-scatter starts when ITS levels are accumulated and runs beside bucket B's accumulation; every rank runs Adam on its 1/G piece of both buckets
// only (+ the replicated 10,240-parameter MLP, whose gradients are all-reduced: 20 KB) and all-gathers the new half parameters, which the next step's K1 (it needs no
// parameters) overlaps; the EMA / inference copy of the foreign pieces is advanced locally from the gathered parameters (12 B x P instead of 38 B x P).
// Same bytes on the wire as the all-reduce (a ring all-reduce IS reduce-scatter + all-gather), less of them exposed, 1/G of the Adam state touched per rank.
// Per-rank optimizer state (fp32 master, Adam moments, step counters) of the FOREIGN pieces goes stale: ngp_nerf_dp_gather_state (collective) refreshes it before a
// snapshot / parameter read-back.
extern "C" int ngp_nerf_dp_gather_state(ngp_nerf* t, void* stream);
static int dp_setup_sharded(ngp_nerf* t, bool on) {
	ngp_model* m = t->model;
	if (m->dp_state_stale && t->dp_sharded && t->comm) { if (ngp_nerf_dp_gather_state(t, nullptr)) return 1; } // leaving (or re-planning) the sharded step continues from every rank's masters: a collective, like the calls that get here (ngp_comm_init / _destroy, ngp_nerf_dp_set_sharded on every rank)
	t->dp_sharded = false; m->dp_split_level = 0;
	if (!on) return 0;
	const uint32_t W = t->opt.world_size, L = m->gm.n_levels, F = m->gm.F;
	REQUIRE(L >= 2, "sharded data-parallel step: the table has one level");
	REQUIRE(!m->cfg.ema_full_precision, "sharded data-parallel step: a full-precision EMA needs every parameter's fp32 master on every rank (falls back to the all-reduce step)");
	const uint64_t total = m->gm.offset[L]; // entries
	// Bucket boundary = half of the LEVELS: k_grad_accumulate runs one block per (chunk, level) with the same number of chunks on every level and one 128 KiB block per CU,
	// so two launches of L / 2 levels each keep whole rounds of blocks (base.json: 2 x 512 blocks on 256 CUs), where a split by bytes (5 + 3 levels: 2.5 + 1.5 rounds)
	// cost 25 us of tails (profiles/r05_dp_sharded_world1.json).  Bucket A is then the smaller one (28 % of the entries): its exchange fits beside bucket B's accumulation.
	const uint32_t ka = L / 2;
	const uint64_t b0 = m->n_mlp, b1 = m->n_mlp + (uint64_t)m->gm.offset[ka] * F, b2 = m->n_mlp + total * F;
	REQUIRE(b2 == m->n_params, "sharded data-parallel step: parameter layout");
	REQUIRE((b1 - b0) % (4ull * W) == 0 && (b2 - b1) % (4ull * W) == 0, "sharded data-parallel step: a bucket does not divide into world_size pieces of whole entries (falls back to the all-reduce step)");
	t->dp_begin[0] = b0; t->dp_end[0] = b1; t->dp_begin[1] = b1; t->dp_end[1] = b2;
	m->dp_split_level = (getenv("NGP_DP_NO_SPLIT") && atoi(getenv("NGP_DP_NO_SPLIT")) != 0) ? 0u : ka; // (ablation: one accumulate launch, both buckets' exchange on the caller's stream)
	if (!t->ev_rs_a) HIPCHK(hipEventCreateWithFlags(&t->ev_rs_a, hipEventDisableTiming));
	t->dp_sharded = true;
	return 0;
}
extern "C" int ngp_nerf_dp_set_sharded(ngp_nerf* t, int on) { REQUIRE(t, "null trainer"); return dp_setup_sharded(t, on != 0); }
extern "C" int ngp_nerf_dp_layout(ngp_nerf* t, uint64_t begin[2], uint64_t end[2]) {
	REQUIRE(t && begin && end, "ngp_nerf_dp_layout: null argument");
	for (int b = 0; b < 2; ++b) { begin[b] = t->dp_sharded ? t->dp_begin[b] : 0; end[b] = t->dp_sharded ? t->dp_end[b] : 0; }
	return 0;
}
// this step's Adam on the replicated MLP and on this rank's pieces (their summed gradients are in place)
static int dp_optimizer_local(ngp_nerf* t, hipStream_t s) {
	ngp_model* m = t->model;
	// (checked BEFORE the step counter moves: a refused step must not skew the EMA debiasing / the decay schedule of the steps behind it)
	REQUIRE(65535.0f * std::log(m->cfg.beta1) < -18.f && 65535.0f * std::log(m->cfg.beta2) < -18.f, "Adam: beta too close to 1 for the 16-bit saturating per-parameter step counters (1 - beta^65535 must round to 1)");
	++m->step;
	AdamArgs a = make_adam_args(m, t->opt.loss_scale, m->step);
	t->dp_adam = a;
	if (t->opt.world_size > 1) m->dp_state_stale = true;
	ProfScope ps(P_OPTIMIZER, s);
	a.range_begin = 0; a.n_params = m->n_mlp; launch_optimizer_step(s, a);
	for (int b = 0; b < 2; ++b) {
		const uint64_t piece = (t->dp_end[b] - t->dp_begin[b]) / t->opt.world_size;
		a.range_begin = t->dp_begin[b] + piece * t->opt.rank; a.n_params = a.range_begin + piece; launch_optimizer_step(s, a);
	}
	HIPCHK(hipGetLastError());
	return 0;
}
// ... and, once every rank's new half parameters have arrived, the EMA / inference copy of the foreign pieces (and their consumed gradients cleared)
static int dp_optimizer_tail(ngp_nerf* t, hipStream_t s) {
	ngp_model* m = t->model;
	AdamArgs a = t->dp_adam;
	a.ema_only = 1;
	{
		ProfScope ps(P_OPTIMIZER, s);
		for (int b = 0; b < 2; ++b) {
			const uint64_t piece = (t->dp_end[b] - t->dp_begin[b]) / t->opt.world_size, own = t->dp_begin[b] + piece * t->opt.rank;
			a.range_begin = t->dp_begin[b]; a.n_params = own; launch_optimizer_step(s, a);
			a.range_begin = own + piece; a.n_params = t->dp_end[b]; launch_optimizer_step(s, a);
		}
	}
	m->last_sweep_params = m->n_params;
	m->grads_clean = a.zero_grid_grads != 0;
	HIPCHK(hipGetLastError());
	if (m->cfg.decay_interval > 0 && m->step >= m->cfg.decay_start && m->step % m->cfg.decay_interval == 0) m->lr *= m->cfg.decay_base; // ExponentialDecay::step [tcnn]
	return 0;
}
static int nerf_finish_impl(ngp_nerf* t, void* stream, bool run_optimizer);
// For callers that run the collectives themselves (tests/test_gpu_dist.py: gloo through host memory):  phase 0 = Adam on the MLP + this rank's pieces (after the
// reduce-scatter / all-reduce put the summed gradients in place), phase 1 = the foreign pieces' EMA + the rest of ngp_nerf_train_finish (after the all-gather of the parameters).
extern "C" int ngp_nerf_train_finish_sharded(ngp_nerf* t, void* stream, int phase) {
	REQUIRE(t && t->dp_sharded, "ngp_nerf_train_finish_sharded: the sharded step is not set up (ngp_nerf_dp_set_sharded / ngp_comm_init)");
	if (phase == 0) return dp_optimizer_local(t, (hipStream_t)stream);
	if (dp_optimizer_tail(t, (hipStream_t)stream)) return 1;
	return nerf_finish_impl(t, stream, false);
}
extern "C" int ngp_comm_unique_id(uint8_t id_out_host[128]) {
	REQUIRE(id_out_host, "ngp_comm_unique_id: null argument");
	if (rccl_load()) return 1;
	NcclId id; RCCLCHK(g_rccl.GetUniqueId(&id));
	memcpy(id_out_host, id.internal, 128);
	return 0;
}
extern "C" int ngp_comm_init(ngp_nerf* t, uint32_t rank, uint32_t world_size, const uint8_t id_host[128]) {
	REQUIRE(t && id_host, "ngp_comm_init: null argument");
	REQUIRE(rank == t->opt.rank && world_size == t->opt.world_size, "ngp_comm_init: rank / world_size differ from the trainer's sharding (ngp_nerf_options)");
	REQUIRE(!t->comm, "ngp_comm_init: communicator already initialised");
	if (rccl_load()) return 1;
	NcclId id; memcpy(id.internal, id_host, 128);
	RCCLCHK(g_rccl.CommInitRank(&t->comm, (int)world_size, id, (int)rank));
	t->model->record_bucket_events = false; // the all-reduce runs on the caller's stream behind the backward pass: no communication stream, no bucket events
	// the sharded step (reduce-scatter -> Adam on this rank's pieces -> all-gather) is the default from two ranks on; NGP_DP_ALLREDUCE=1 keeps the round-2..4 step
	// (all-reduce -> replicated sweep), NGP_DP_SHARDED=1 runs the sharded structure with a communicator of one rank as well (diagnostic / test)
	const bool force_allreduce = getenv("NGP_DP_ALLREDUCE") && atoi(getenv("NGP_DP_ALLREDUCE")) != 0; // (read per communicator: a test sets them between two trainers)
	const bool force_sharded = getenv("NGP_DP_SHARDED") && atoi(getenv("NGP_DP_SHARDED")) != 0;
	const bool want = !force_allreduce && (world_size > 1 || force_sharded) && g_rccl.ReduceScatter && g_rccl.AllGather && g_rccl.GroupStart && g_rccl.GroupEnd;
	if (want) (void)dp_setup_sharded(t, true); // (a layout that does not divide by the world size keeps the all-reduce step)
	return 0;
}
extern "C" int ngp_comm_destroy(ngp_nerf* t) {
	if (!t || !t->comm) return 0;
	(void)hipDeviceSynchronize();
	RCCLCHK(g_rccl.CommDestroy(t->comm));
	t->comm = nullptr; t->model->record_bucket_events = false;
	(void)dp_setup_sharded(t, false);
	if (t->comm_stream) { (void)hipStreamDestroy(t->comm_stream); t->comm_stream = nullptr; }
	if (t->ev_red_a) { (void)hipEventDestroy(t->ev_red_a); (void)hipEventDestroy(t->ev_red_b); t->ev_red_a = t->ev_red_b = nullptr; }
	return 0;
}
// all-reduce(sum) of the whole fp16 gradient buffer on `stream` (the un-bucketed form; tests / callers that sequence the step themselves)
extern "C" int ngp_allreduce_gradients(ngp_nerf* t, void* stream) {
	REQUIRE(t && t->comm, "ngp_allreduce_gradients: ngp_comm_init has not been called");
	RCCLCHK(g_rccl.AllReduce(t->model->grads, t->model->grads, t->model->n_params, kNcclHalf, kNcclSum, t->comm, (hipStream_t)stream));
	return 0;
}
static bool dp_skip_allreduce() { static const bool v = getenv("NGP_DP_SKIP_ALLREDUCE") && atoi(getenv("NGP_DP_SKIP_ALLREDUCE")) != 0; return v; } // diagnostic: the step's structure without the RCCL calls
extern "C" int ngp_allreduce_counters(ngp_nerf* t, void* stream) {
	REQUIRE(t && t->comm, "ngp_allreduce_counters: ngp_comm_init has not been called");
	if (!dp_skip_allreduce()) RCCLCHK(g_rccl.AllReduce(t->sync2, t->sync2, 3, kNcclUint32, kNcclSum, t->comm, (hipStream_t)stream));
	return 0;
}
// Gradient all-reduce behind ngp_nerf_train_backward: ONE ring over the whole fp16 gradient vector on the CALLER's stream.  Every level goes
// through the record lists now, so the big bucket is complete only when the backward pass is (k_grad_accumulate is its last kernel) and a
// second bucket has nothing to overlap with; the next step's K1 runs on its own stream next to the ring.  The round-2a layout -- two
// buckets on a communication stream of their own -- paid 0.6 ms per step for its event chain (two waits and two records on an otherwise
// idle high-priority stream, two waits on the caller's stream): 1.41 vs 0.79 ms at world size 1 (profiles/r02_dp_overhead.txt).
static int dp_reduce_gradients(ngp_nerf* t, hipStream_t s) {
	ngp_model* m = t->model;
	if (!dp_skip_allreduce()) RCCLCHK(g_rccl.AllReduce(m->grads, m->grads, m->n_params, kNcclHalf, kNcclSum, t->comm, s));
	t->grads_pending = false;
	return 0;
}

// reduce-scatter (bucket A beside bucket B's accumulation, on the communication stream) -> local Adam -> all-gather -> foreign EMA, all behind ngp_nerf_train_backward
static int dp_sharded_exchange_and_step(ngp_nerf* t, hipStream_t s) {
	ngp_model* m = t->model;
	const uint32_t W = t->opt.world_size, r = t->opt.rank;
	const bool skip = dp_skip_allreduce();
	ngp_half* g = m->grads; ngp_half* p = m->params;
	const uint64_t pieceA = (t->dp_end[0] - t->dp_begin[0]) / W, pieceB = (t->dp_end[1] - t->dp_begin[1]) / W;
	const bool split = m->dp_split_done && g_comm_stream != nullptr;
	if (split) { // bucket A's gradients are final behind ev_bucket_a: its exchange runs on the communication stream while the caller's stream accumulates bucket B
		HIPCHK(hipStreamWaitEvent(g_comm_stream, m->ev_bucket_a, 0));
		if (!skip) RCCLCHK(g_rccl.ReduceScatter(g + t->dp_begin[0], g + t->dp_begin[0] + pieceA * r, pieceA, kNcclHalf, kNcclSum, t->comm, g_comm_stream));
		HIPCHK(hipEventRecord(t->ev_rs_a, g_comm_stream));
	}
	if (!skip) {
		RCCLCHK(g_rccl.AllReduce(g, g, m->n_mlp, kNcclHalf, kNcclSum, t->comm, s)); // the replicated MLP (20 KB)
		if (!split) RCCLCHK(g_rccl.ReduceScatter(g + t->dp_begin[0], g + t->dp_begin[0] + pieceA * r, pieceA, kNcclHalf, kNcclSum, t->comm, s));
		RCCLCHK(g_rccl.ReduceScatter(g + t->dp_begin[1], g + t->dp_begin[1] + pieceB * r, pieceB, kNcclHalf, kNcclSum, t->comm, s));
	}
	if (split) HIPCHK(hipStreamWaitEvent(s, t->ev_rs_a, 0));
	t->grads_pending = false;
	if (dp_optimizer_local(t, s)) return 1;
	if (!skip) { // the new half parameters of every rank's pieces (the next step's K1 runs on its own stream beside this)
		RCCLCHK(g_rccl.GroupStart());
		RCCLCHK(g_rccl.AllGather(p + t->dp_begin[0] + pieceA * r, p + t->dp_begin[0], pieceA, kNcclHalf, t->comm, s));
		RCCLCHK(g_rccl.AllGather(p + t->dp_begin[1] + pieceB * r, p + t->dp_begin[1], pieceB, kNcclHalf, t->comm, s));
		RCCLCHK(g_rccl.GroupEnd());
	}
	return dp_optimizer_tail(t, s);
}
// Collective: every rank's fp32 master parameters, Adam moments, per-parameter step counters and EMA state of its OWN pieces -> all ranks (the sharded step keeps only
// the half parameters and the inference copy current everywhere).  Call on every rank before ngp_model_get_params_host / ngp_model_serialize_host of a sharded run.
extern "C" int ngp_nerf_dp_gather_state(ngp_nerf* t, void* stream) {
	REQUIRE(t, "null trainer");
	if (!t->dp_sharded || !t->comm || t->opt.world_size < 2) return 0;
	ngp_model* m = t->model; hipStream_t s = (hipStream_t)stream;
	constexpr int kNcclFloat32 = 7, kNcclUint16AsHalf = kNcclHalf; // (16-bit payloads travel as halfs: all-gather does not interpret them)
	RCCLCHK(g_rccl.GroupStart());
	for (int b = 0; b < 2; ++b) {
		const uint64_t piece = (t->dp_end[b] - t->dp_begin[b]) / t->opt.world_size, lo = t->dp_begin[b], own = lo + piece * t->opt.rank;
		RCCLCHK(g_rccl.AllGather(m->master + own, m->master + lo, piece, kNcclFloat32, t->comm, s));
		RCCLCHK(g_rccl.AllGather(m->adam_m + own, m->adam_m + lo, piece, kNcclFloat32, t->comm, s));
		RCCLCHK(g_rccl.AllGather(m->adam_v + own, m->adam_v + lo, piece, kNcclFloat32, t->comm, s));
		RCCLCHK(g_rccl.AllGather(m->adam_steps + own, m->adam_steps + lo, piece, kNcclUint16AsHalf, t->comm, s));
	}
	RCCLCHK(g_rccl.GroupEnd());
	HIPCHK(hipStreamSynchronize(s));
	m->dp_state_stale = false;
	return 0;
}
extern "C" int ngp_nerf_dp_state_stale(const ngp_nerf* t) { return t && t->model && t->model->dp_state_stale ? 1 : 0; }
// for callers that run the collectives themselves (ngp_nerf_train_finish_sharded): they have all-gathered master / Adam m / v / step counters of every piece (ngp_model_param_ptrs + the layout)
extern "C" int ngp_nerf_dp_state_gathered(ngp_nerf* t) { REQUIRE(t, "null trainer"); t->model->dp_state_stale = false; return 0; }

static int error_map_allreduce(ngp_nerf* t, hipStream_t s, size_t n) {
	REQUIRE(t->comm, "error-proportional sampling under data parallelism needs the in-library communicator (ngp_comm_init): the ranks' error maps are summed before the CDFs are built");
	constexpr int kNcclFloat32 = 7;
	RCCLCHK(g_rccl.AllReduce(t->error_map, t->error_map, n, kNcclFloat32, kNcclSum, t->comm, s));
	return 0;
}
extern "C" int ngp_nerf_error_map_ptrs(ngp_nerf* t, float** error_map, int32_t error_map_res[2], float** cdf_x_cond_y, float** cdf_y, float** cdf_img, int32_t cdf_res[2],
		int* cdf_valid, uint32_t* n_steps_between_updates, uint32_t* n_steps_since_update) {
	REQUIRE(t, "ngp_nerf_error_map_ptrs: null argument");
	if (error_map) *error_map = t->error_map;
	if (error_map_res) { error_map_res[0] = t->error_map_res[0]; error_map_res[1] = t->error_map_res[1]; }
	if (cdf_x_cond_y) *cdf_x_cond_y = t->cdf_x_cond_y;
	if (cdf_y) *cdf_y = t->cdf_y;
	if (cdf_img) *cdf_img = t->cdf_img;
	if (cdf_res) { cdf_res[0] = t->cdf_res[0]; cdf_res[1] = t->cdf_res[1]; }
	if (cdf_valid) *cdf_valid = t->cdf_valid ? 1 : 0;
	if (n_steps_between_updates) *n_steps_between_updates = t->n_steps_between_error_map_updates;
	if (n_steps_since_update) *n_steps_since_update = t->n_steps_since_error_map_update;
	return 0;
}
// Testbed::Nerf::Training::n_steps_between_error_map_updates (testbed.h:813; 128 after a reset, x 1.5 per cycle): applies from the next cycle on
extern "C" int ngp_nerf_set_error_map_interval(ngp_nerf* t, uint32_t n_steps) {
	REQUIRE(t && n_steps >= 1, "ngp_nerf_set_error_map_interval: bad argument");
	REQUIRE(!t->error_cycle_open, "ngp_nerf_set_error_map_interval: an accumulation cycle is open (the map's resolution was derived from the old interval)");
	t->n_steps_between_error_map_updates = n_steps;
	return 0;
}
// test / tooling hook: install CDFs computed elsewhere (all three, cdf_res[0] x cdf_res[1] cells per image); the option switches decide which of them the step uses
extern "C" int ngp_nerf_set_error_cdfs_host(ngp_nerf* t, const float* cdf_x_cond_y, const float* cdf_y, const float* cdf_img, const int32_t cdf_res[2]) {
	REQUIRE(t && cdf_x_cond_y && cdf_y && cdf_img && cdf_res && cdf_res[0] >= 1 && cdf_res[1] >= 1 && t->n_images > 0, "ngp_nerf_set_error_cdfs_host: bad argument");
	invalidate_k1(t);
	HIPCHK(hipDeviceSynchronize());
	const size_t n = (size_t)cdf_res[0] * cdf_res[1] * t->n_images;
	if (dev_grow(&t->cdf_x_cond_y, &t->cdf_xy_cap, n) || dev_grow(&t->cdf_y, &t->cdf_y_cap, (size_t)cdf_res[1] * t->n_images)) return 1;
	if (dev_grow(&t->cdf_img, &t->cdf_img_cap, (size_t)t->n_images)) return 1;
	HIPCHK(hipMemcpy(t->cdf_x_cond_y, cdf_x_cond_y, n * 4, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(t->cdf_y, cdf_y, (size_t)cdf_res[1] * t->n_images * 4, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(t->cdf_img, cdf_img, (size_t)t->n_images * 4, hipMemcpyHostToDevice));
	t->cdf_res[0] = cdf_res[0]; t->cdf_res[1] = cdf_res[1]; t->cdf_valid = true;
	return 0;
}
extern "C" int ngp_nerf_train(ngp_nerf* t, void* stream, uint32_t n_steps) {
	for (uint32_t i = 0; i < n_steps; ++i) {
		if (ngp_nerf_train_prep(t, stream)) return 1;
		// diagnostics (world size 1 only): NGP_TRAIN_SPLIT_PHASES=1 runs the split-phase step without a communicator, NGP_DP_FUSED_STEP=1 the
		// single-call step although a communicator exists
		static const bool split_phases = getenv("NGP_TRAIN_SPLIT_PHASES") && atoi(getenv("NGP_TRAIN_SPLIT_PHASES")) != 0;
		static const bool fused_step = getenv("NGP_DP_FUSED_STEP") && atoi(getenv("NGP_DP_FUSED_STEP")) != 0;
		if (!t->comm && split_phases && t->opt.world_size == 1) {
			if (ngp_nerf_train_forward(t, stream)) return 1;
			if (ngp_nerf_train_backward(t, stream)) return 1;
		} else if (t->comm && !(fused_step && t->opt.world_size == 1)) { // data-parallel: forward -> all-reduce(counters) -> backward + bucketed all-reduce(gradients) -> optimizer
			if (ngp_nerf_train_forward(t, stream)) return 1;
			if (ngp_allreduce_counters(t, stream)) return 1;
			if (ngp_nerf_train_backward(t, stream)) return 1;
			if (t->dp_sharded) { // reduce-scatter -> Adam on this rank's pieces -> all-gather -> the rest of the step
				if (dp_sharded_exchange_and_step(t, (hipStream_t)stream)) return 1;
				if (nerf_finish_impl(t, stream, false)) return 1;
				continue;
			}
			if (dp_reduce_gradients(t, (hipStream_t)stream)) return 1;
		} else if (nerf_step_impl(t, stream, 3, false, true)) return 1; // (= ngp_nerf_train_forward_backward, and the optimizer step follows at once: its hashed-level part is fused into the scatter)
		if (ngp_nerf_train_finish(t, stream)) return 1;
	}
	return 0;
}
extern "C" int ngp_nerf_counter_ptrs(ngp_nerf* t, uint32_t** counters2) { *counters2 = t->sync2; return 0; }
extern "C" int ngp_nerf_uses_k2_stash(ngp_nerf* t) { return t && t->k2_enc_valid ? 1 : 0; }
extern "C" uint64_t ngp_model_last_sweep_params(const ngp_model* m) { return m ? m->last_sweep_params : 0; }
extern "C" int ngp_nerf_get_stats(ngp_nerf* t, void* stream, ngp_nerf_stats* out) {
	// (pinned staging buffer: a device-to-host copy into pageable memory goes through the runtime's own staging and costs a few hundred microseconds of an idle GPU per call --
	// Testbed::train reads the loss every 16 steps like the reference, testbed.cu:4625: 0.471 -> see profiles/r06_pyngp_frame_rate.txt)
	if (!t->stats_host) HIPCHK(hipHostMalloc((void**)&t->stats_host, sizeof(TrainCounters), hipHostMallocDefault));
	HIPCHK(hipMemcpyAsync(t->stats_host, t->counters, sizeof(TrainCounters), hipMemcpyDeviceToHost, (hipStream_t)stream));
	HIPCHK(hipStreamSynchronize((hipStream_t)stream));
	const TrainCounters c = *t->stats_host;
	out->training_step = c.training_step; out->rays_per_batch = c.rays_per_batch; out->n_rays_last = c.n_rays_last;
	out->measured_batch_size = c.measured_batch_size; out->measured_batch_size_before_compaction = c.measured_batch_size_before_compaction;
	out->loss = c.loss_scalar; out->total_rays = c.total_rays; out->total_samples = c.total_samples;
	out->network_evaluations = c.k2_samples_last ? c.k2_samples_last : c.measured_batch_size_before_compaction; out->reserved = 0;
	return 0;
}
extern "C" uint32_t ngp_nerf_grid_ahead_hits(ngp_nerf* t) { return t ? t->grid_ahead_hits : 0u; }
extern "C" int ngp_nerf_density_grid_ptrs(ngp_nerf* t, float** grid, uint8_t** bitfield, float** mean) {
	if (grid) *grid = t->density_grid; if (bitfield) *bitfield = t->bitfield; if (mean) *mean = t->mean; return 0;
}
extern "C" int ngp_nerf_set_density_grid_host(ngp_nerf* t, void* stream, const float* grid_host, uint64_t n) {
	invalidate_k1(t);
	REQUIRE(n == (uint64_t)GRID_N_CELLS * (t->opt.max_cascade + 1), "set_density_grid: size mismatch");
	HIPCHK(hipMemcpy(t->density_grid, grid_host, n * 4, hipMemcpyHostToDevice));
	launch_grid_mean((hipStream_t)stream, t->density_grid, t->mean_partial, t->mean);
	launch_grid_to_bitfield((hipStream_t)stream, t->density_grid, t->opt.max_cascade, t->bitfield, t->mean);
	launch_build_linear_bitfield((hipStream_t)stream, t->bitfield, t->bitfield_linear, N_CASCADES, t->bitfield_coarse);
	HIPCHK(hipGetLastError());
	return 0;
}
// test hooks: scratch buffers of the last train_forward_backward, and state overrides
extern "C" int ngp_nerf_scratch_ptrs(ngp_nerf* t, uint32_t** ray_indices, ngp_ray** rays, uint32_t** numsteps, float** coords, ngp_half** mlp_out,
		float** coords_compacted, ngp_half** dloss, void** counters) {
	*ray_indices = t->ray_indices; *rays = t->rays; *numsteps = t->numsteps; *coords = t->coords; *mlp_out = t->mlp_out; *coords_compacted = t->coords_compacted;
	*dloss = t->dloss; *counters = t->counters;
	return 0;
}
extern "C" int ngp_nerf_set_rays_per_batch(ngp_nerf* t, uint32_t r) {
	invalidate_k1(t);
	HIPCHK(hipMemcpy(&t->counters->rays_per_batch, &r, 4, hipMemcpyHostToDevice));
	return 0;
}
// load_snapshot restores m_training_step (testbed.cu:5400-5403): keeps the prep cadence / step-0 grid marking consistent
extern "C" int ngp_nerf_set_training_step(ngp_nerf* t, uint32_t step) {
	invalidate_k1(t);
	t->training_step = step; t->prep_skip_counter = step; t->ema_step = step;
	HIPCHK(hipMemcpy(&t->counters->training_step, &step, 4, hipMemcpyHostToDevice));
	return 0;
}
// lazy K2 tuning knobs (test / ablation hook): 1 = single launch with in-wave continuation, 2..8 = list-driven front-to-back rounds; samples per tile (16 | 32)
extern "C" int ngp_nerf_set_k2_params(ngp_nerf* t, uint32_t rounds, uint32_t tile_w) {
	REQUIRE(rounds >= 1 && rounds <= K2_ROUNDS && (tile_w == 32 || tile_w == 16 || (tile_w == 8 && rounds == 1)), "set_k2_params: rounds in 1..8 (1 = one launch, wavefronts follow their rays), tile width 16 or 32 (8 with rounds = 1)");
	invalidate_k1(t); // a pre-launched K1 wrote its round-0 tiles with the old width
	t->k2_rounds = rounds; t->k2_tile_w = tile_w;
	return 0;
}
extern "C" int ngp_nerf_set_rng(ngp_nerf* t, const ngp_pcg32* rng) { invalidate_k1(t); t->rng.state = rng->state; t->rng.inc = rng->inc; return 0; }
extern "C" int ngp_nerf_get_rng(ngp_nerf* t, ngp_pcg32* rng, ngp_pcg32* grid_rng) { *rng = pod(t->rng); *grid_rng = pod(t->density_grid_rng); return 0; }

// stand-alone launches of the two extra-dims kernels (test hooks like ngp_k_generate_training_samples): device pointers
extern "C" int ngp_k_extra_dims_gradient(void* stream, uint32_t n_rays_total, uint32_t rays_counter, float* grad_out, uint32_t n_extra, uint32_t n_images, const uint32_t* ray_indices,
		const uint32_t* numsteps, const float* dextra, uint32_t max_rows) {
	static uint32_t* s_cnt = nullptr;
	if (!s_cnt && dev_alloc(&s_cnt, 2)) return 1;
	const uint32_t h[2] = {n_rays_total, rays_counter};
	HIPCHK(hipMemcpyAsync(s_cnt, h, 8, hipMemcpyHostToDevice, (hipStream_t)stream));
	launch_extra_dims_gradient((hipStream_t)stream, rays_counter, s_cnt, s_cnt + 1, grad_out, n_extra, n_images, ray_indices, numsteps, dextra, max_rows, nullptr);
	HIPCHK(hipGetLastError());
	return 0;
}
extern "C" int ngp_k_extra_dims_adam(void* stream, uint32_t n, float* variable, const float* gradient_scaled, float* m, float* v, uint32_t iter, float lr, float loss_scale) {
	launch_extra_dims_adam((hipStream_t)stream, n, variable, gradient_scaled, m, v, iter, lr, loss_scale);
	HIPCHK(hipGetLastError());
	return 0;
}
// ---- extra dims (testbed.h Nerf::Training::extra_dims_gpu, optimize_extra_dims; python_api.cu set_rendering_extra_dims*) ----
// values: n_images x n_extra_dims floats on the host (reset_extra_dims computes them: warped light directions / uniform random latents, testbed_nerf.cu:3656-3683) for EVERY
// image of the dataset -- n_images >= the number of images rays are drawn from (ngp_nerf_set_dataset_*'s n = n_images_for_training): images that join the training set later
// find their initial values here, and the getters / snapshots cover the whole dataset like the reference's.  Installs them, resets the per-image optimizers (reset_state)
// and takes the copy of image 0's dims a rendering uses by default (rendering_extra_dims, :3679-3682).
extern "C" int ngp_nerf_set_extra_dims(ngp_nerf* t, const float* values, uint32_t n_images) {
	REQUIRE(t && values && t->n_extra > 0, "set_extra_dims: the model has no extra dims");
	REQUIRE(n_images >= t->n_images && n_images > 0 && t->extra_dims, "set_extra_dims: one vector per image of the dataset, at least the images in training (set the dataset first)");
	invalidate_k1(t);
	HIPCHK(hipDeviceSynchronize());
	if (extra_dims_reserve(t, n_images)) return 1;
	const size_t cnt = (size_t)n_images * t->n_extra;
	HIPCHK(hipMemcpy(t->extra_dims, values, cnt * 4, hipMemcpyHostToDevice));
	HIPCHK(hipMemset(t->extra_m, 0, (size_t)t->extra_cap * t->n_extra * 4)); HIPCHK(hipMemset(t->extra_v, 0, (size_t)t->extra_cap * t->n_extra * 4));
	HIPCHK(hipMemset(t->extra_iter, 0, (size_t)t->extra_cap * 4));
	t->extra_lr_last = 1e-4f;
	t->rendering_extra_default.assign(values, values + t->n_extra);
	return 0;
}
extern "C" int ngp_nerf_get_extra_dims(ngp_nerf* t, float* values, uint32_t n_images) {
	REQUIRE(t && values && t->n_extra > 0 && n_images <= t->extra_cap && t->extra_dims, "get_extra_dims: no extra dims / too many images");
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpy(values, t->extra_dims, (size_t)n_images * t->n_extra * 4, hipMemcpyDeviceToHost));
	return 0;
}
// The per-image optimizers' state (Training::extra_dims_opt, a std::vector<VarAdamOptimizer>: snapshots carry it, testbed.cu:5311 / 5482-5486): first / second moments
// (n_images x n_extra_dims each), the iteration count of every image's optimizer (n_images values) and the learning rate their last step used (learning_rate_out, may be
// null).  set: variables, moments and counts as they are (no reset), for the first n_images images of the dataset.
extern "C" int ngp_nerf_get_extra_dims_optimizer(ngp_nerf* t, float* first_moment, float* second_moment, uint32_t* iter, uint32_t n_images) {
	REQUIRE(t && first_moment && second_moment && iter && t->n_extra > 0 && n_images <= t->extra_cap && t->extra_dims, "get_extra_dims_optimizer: no extra dims / too many images");
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpy(first_moment, t->extra_m, (size_t)n_images * t->n_extra * 4, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(second_moment, t->extra_v, (size_t)n_images * t->n_extra * 4, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(iter, t->extra_iter, (size_t)n_images * 4, hipMemcpyDeviceToHost));
	return 0;
}
extern "C" float ngp_nerf_extra_dims_learning_rate(const ngp_nerf* t) { return t ? t->extra_lr_last : 0.f; }
extern "C" int ngp_nerf_set_extra_dims_optimizer(ngp_nerf* t, const float* variable, const float* first_moment, const float* second_moment, const uint32_t* iter, uint32_t n_images) {
	REQUIRE(t && variable && first_moment && second_moment && iter && t->n_extra > 0 && n_images <= t->extra_cap && t->extra_dims, "set_extra_dims_optimizer: no extra dims / too many images");
	invalidate_k1(t);
	HIPCHK(hipDeviceSynchronize());
	const size_t bytes = (size_t)n_images * t->n_extra * 4;
	HIPCHK(hipMemcpy(t->extra_dims, variable, bytes, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(t->extra_m, first_moment, bytes, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(t->extra_v, second_moment, bytes, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(t->extra_iter, iter, (size_t)n_images * 4, hipMemcpyHostToDevice));
	return 0;
}
extern "C" int ngp_nerf_get_extra_dims_gradient(ngp_nerf* t, float* values, uint32_t n_images) { // the last step's (loss-scaled) per-image gradient: test hook
	REQUIRE(t && values && t->n_extra > 0 && n_images <= t->extra_cap && t->extra_grad, "get_extra_dims_gradient: no extra dims / too many images");
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpy(values, t->extra_grad, (size_t)n_images * t->n_extra * 4, hipMemcpyDeviceToHost));
	return 0;
}
extern "C" int ngp_nerf_set_optimize_extra_dims(ngp_nerf* t, int on) { REQUIRE(t, "null trainer"); invalidate_k1(t); t->optimize_extra_dims = on != 0; return 0; }
// view >= 0: render with that training view's CURRENT dims; view < 0: with `values` (n_extra_dims floats; null = the copy of image 0's initial dims reset_extra_dims took)
extern "C" int ngp_nerf_set_rendering_extra_dims(ngp_nerf* t, int view, const float* values) {
	REQUIRE(t && t->n_extra > 0, "set_rendering_extra_dims: the model has no extra dims");
	t->rendering_extra_view = view;
	t->rendering_extra.clear();
	if (view < 0 && values) t->rendering_extra.assign(values, values + t->n_extra);
	return 0;
}
// Nerf::light_dir + NerfDataset::has_light_dirs (testbed.h, nerf_loader.h): with light directions in the dataset a rendering's first three extra dims are
// warp_direction(normalize(light_dir)) whatever else was selected (get_rendering_extra_dims, testbed_nerf.cu:3697-3706); light_dir defaults to (0.5, 0.5, 0.5).
extern "C" int ngp_nerf_set_light_dir(ngp_nerf* t, int has_light_dirs, const float light_dir[3]) {
	REQUIRE(t && t->n_extra > 0, "set_light_dir: the model has no extra dims");
	t->has_light_dirs = has_light_dirs != 0;
	if (light_dir) for (int k = 0; k < 3; ++k) t->light_dir[k] = light_dir[k];
	return 0;
}

// Testbed::render_nerf (testbed_nerf.cu:1894-2149): one spp of a frame into premultiplied linear RGBA + depth
extern "C" int ngp_nerf_render(ngp_nerf* t, void* stream, const ngp_render_params* rp, float* frame, float* depth) {
	FlagScope flag_scope_(t->dbg);
	REQUIRE(t && rp && frame, "render: null argument");
	REQUIRE(rp->lens_mode >= NGP_LENS_PERSPECTIVE && rp->lens_mode <= NGP_LENS_ORTHOGRAPHIC, "render: unknown lens mode");
	hipStream_t s = (hipStream_t)stream;
	constexpr uint32_t TILE = 1u << 18;
	if (!t->r_rays) {
		if (dev_alloc(&t->r_rays, TILE) || dev_alloc(&t->r_masks, (size_t)TILE * RENDER_MAX_CHUNKS) || dev_alloc(&t->r_alive, TILE) || dev_alloc(&t->r_n_alive, 2) ||
			dev_alloc(&t->r_coords, (size_t)TILE * RENDER_STEPS * (7 + t->n_extra)) || dev_alloc(&t->r_out, (size_t)TILE * RENDER_STEPS * 4)) return 1;
	}
	t->r_n_inf = t->r_n_alive + 1; // live rays x RENDER_STEPS: the inference kernel's device-side element count
	RenderArgs a;
	a.p = *rp; a.train_aabb = t->aabb; a.bitfield = t->bitfield; a.max_mip = t->opt.max_cascade; a.cone_angle = t->opt.cone_angle_constant;
	a.rgb_activation = t->opt.rgb_activation; a.density_activation = t->opt.density_activation; a.linear_colors = t->opt.linear_colors;
	a.rays = t->r_rays; a.masks = t->r_masks;
	if (t->n_extra) { // Testbed::Nerf::get_rendering_extra_dims (testbed_nerf.cu:3685-3707): a training view's dims, or the explicit values, through the slot behind the images'
		REQUIRE(t->extra_dims, "render: a model with extra dims needs a dataset (the extra dims are per image)");
		REQUIRE(t->rendering_extra_view < (int)t->extra_cap, "render: rendering_extra_dims_from_training_view out of range");
		float* slot = t->extra_dims + (size_t)t->extra_cap * t->n_extra;
		if (t->rendering_extra_view >= 0) HIPCHK(hipMemcpyAsync(slot, t->extra_dims + (size_t)t->rendering_extra_view * t->n_extra, t->n_extra * 4, hipMemcpyDeviceToDevice, s));
		else if (t->rendering_extra.size() == t->n_extra) HIPCHK(hipMemcpyAsync(slot, t->rendering_extra.data(), t->n_extra * 4, hipMemcpyHostToDevice, s));
		else if (t->rendering_extra_default.size() == t->n_extra) HIPCHK(hipMemcpyAsync(slot, t->rendering_extra_default.data(), t->n_extra * 4, hipMemcpyHostToDevice, s)); // reset_extra_dims: rendering_extra_dims = a COPY of image 0's initial dims (testbed_nerf.cu:3679-3682), not its trained ones
		else HIPCHK(hipMemcpyAsync(slot, t->extra_dims, t->n_extra * 4, hipMemcpyDeviceToDevice, s)); // (no ngp_nerf_set_extra_dims yet: image 0's)
		if (t->has_light_dirs) { // testbed_nerf.cu:3697-3706: the light direction of the rendering replaces the first three dims
			const float l = sqrtf(t->light_dir[0] * t->light_dir[0] + t->light_dir[1] * t->light_dir[1] + t->light_dir[2] * t->light_dir[2]);
			t->light_dir_warped[0] = (t->light_dir[0] / l + 1.0f) * 0.5f; t->light_dir_warped[1] = (t->light_dir[1] / l + 1.0f) * 0.5f; t->light_dir_warped[2] = (t->light_dir[2] / l + 1.0f) * 0.5f; // warp_direction(normalize(light_dir)), nerf_device.cuh:291
			HIPCHK(hipMemcpyAsync(slot, t->light_dir_warped, std::min<size_t>((size_t)t->n_extra * 4, 12), hipMemcpyHostToDevice, s));
		}
		a.extra_dims = slot; a.n_extra = t->n_extra;
	}
	const uint64_t n_pix = (uint64_t)rp->resolution[0] * rp->resolution[1];
	for (uint64_t begin = 0; begin < n_pix; begin += TILE) {
		const uint32_t n = (uint32_t)std::min<uint64_t>(TILE, n_pix - begin);
		launch_render_setup(s, a, (uint32_t)begin, n);
		launch_render_compact(s, a, n, t->r_alive, t->r_n_alive);
		uint32_t n_alive = 0;
		HIPCHK(hipMemcpyAsync(&n_alive, t->r_n_alive, 4, hipMemcpyDeviceToHost, s));
		HIPCHK(hipStreamSynchronize(s));
		// The reference synchronises after every compaction (testbed_nerf.cu:1735-1736).  Here the live-ray count stays on the device:
		// emit / inference / composite take it from r_n_alive (grids sized for the last count the host has seen, an upper bound since rays
		// only die), and the host looks at it after groups of 2, 4, 8, ... rounds -- a few read-backs per tile instead of one per round.
		const uint32_t max_rounds = (RENDER_MAX_CHUNKS * 64) / RENDER_STEPS + 1;
		for (uint32_t round = 0, group = 2; n_alive > 0 && round < max_rounds; group = std::min(group * 2, 16u)) {
			for (uint32_t g = 0; g < group && round < max_rounds; ++g, ++round) {
				launch_render_emit(s, a, n_alive, t->r_alive, t->r_n_alive, t->r_coords);
				launch_inference(s, t->model->gm_dev, model_ptrs(t->model, rp->use_inference_params != 0), t->r_coords, 7 + t->n_extra, n_alive * RENDER_STEPS, t->r_n_inf, t->r_out, 4, false, 4, t->model->gm.F);
				launch_render_composite(s, a, n_alive, t->r_alive, t->r_n_alive, t->r_coords, t->r_out);
				launch_render_compact(s, a, n, t->r_alive, t->r_n_alive);
			}
			HIPCHK(hipMemcpyAsync(&n_alive, t->r_n_alive, 4, hipMemcpyDeviceToHost, s));
			HIPCHK(hipStreamSynchronize(s));
		}
		launch_render_finish(s, a, (uint32_t)begin, n, frame, depth);
	}
	HIPCHK(hipGetLastError());
	return 0;
}

// CudaRenderBuffer::accumulate / tonemap (render_buffer.cu:228-260, 511-560; Identity tonemap curve): device-side so that a multi-spp
// render leaves the GPU once, as the finished frame
extern "C" int ngp_render_accumulate(void* stream, const float* frame, float* accum, uint64_t n_floats, uint32_t sample_index) {
	REQUIRE(frame && accum, "ngp_render_accumulate: null argument");
	launch_render_accumulate((hipStream_t)stream, (uint32_t)n_floats, frame, accum, 1.0f / (float)(sample_index + 1));
	HIPCHK(hipGetLastError());
	return 0;
}
extern "C" int ngp_render_tonemap_curve(void* stream, float* rgba, uint64_t n_pixels, float exposure, const float background_linear[4], int to_srgb, int curve) {
	REQUIRE(rgba && background_linear, "ngp_render_tonemap: null argument");
	REQUIRE(curve >= 0 && curve <= 3, "ngp_render_tonemap: curve = 0 Identity, 1 ACES, 2 Hable, 3 Reinhard");
	launch_render_tonemap((hipStream_t)stream, (uint32_t)n_pixels, rgba, std::pow(2.0f, exposure), background_linear, to_srgb, curve);
	HIPCHK(hipGetLastError());
	return 0;
}
extern "C" int ngp_render_tonemap(void* stream, float* rgba, uint64_t n_pixels, float exposure, const float background_linear[4], int to_srgb) {
	return ngp_render_tonemap_curve(stream, rgba, n_pixels, exposure, background_linear, to_srgb, 0);
}
// host-side evaluation of the device's tonemapping (csrc/ngp_device.hpp tonemap_pixel, compiled for the host): test hook, no GPU needed
extern "C" int ngp_host_tonemap_pixel(const float rgba[4], float exposure, const float background_linear[4], int to_srgb, int curve, float out[4]) {
	const f4 r = tonemap_pixel({rgba[0], rgba[1], rgba[2], rgba[3]}, std::pow(2.0f, exposure), background_linear[0], background_linear[1], background_linear[2], background_linear[3], to_srgb, curve);
	out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
	return 0;
}

// per-handle ablation switches (see FlagScope): on = 0 removes the override (the handle follows the process-wide switches again)
extern "C" int ngp_model_set_debug_flags(ngp_model* m, int on, uint32_t flags, uint32_t flags2) { REQUIRE(m, "null model"); m->dbg.on = on != 0; m->dbg.flags = flags; m->dbg.flags2 = flags2; return 0; }
extern "C" int ngp_nerf_set_debug_flags(ngp_nerf* t, int on, uint32_t flags, uint32_t flags2) {
	REQUIRE(t, "null trainer");
	invalidate_k1(t);
	t->dbg.on = on != 0; t->dbg.flags = flags; t->dbg.flags2 = flags2;
	t->model->dbg = t->dbg;
	return 0;
}
