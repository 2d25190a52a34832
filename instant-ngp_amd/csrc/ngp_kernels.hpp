// ngp_kernels.hpp -- internal (non-ABI) declarations shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ngp_hip.h"

namespace ngp {

// debug / ablation switches (ngp_debug_set_flags); 0 in production
extern uint32_t g_debug_flags;
extern uint32_t g_debug_flags2; // second word of ablation switches (ngp_debug_set_flags2; the 32 bits of the first are taken)
enum : uint32_t {
	DBG2_NO_FUSED_T1W = 1,      // round-4 backward pass: T1 (k_train_fwd_bwd) and W (k_wgrad2) as two kernels instead of k_train_fused
	DBG2_K1_SETUP_GENERAL = 2,  // k1_setup's general instance (seven lens models, rolling shutter, three pixel formats, CDF samplers) also for plain datasets
	DBG2_GRID_NO_AHEAD = 4,     // occupancy-grid update: samples drawn and sorted inside the update (rounds 1-6a) instead of ahead of it on a side stream
};
enum : uint32_t { DBG_K1_REFERENCE_LAYOUT = 1 /* thread-per-ray sequential march, exact reference recurrence */, DBG_T1_NO_SCATTER = 2, DBG_T1_NO_COARSE_LEVELS = 4, DBG_T1_NO_FINE_LEVELS = 8, DBG_T1_NO_MERGE = 16, DBG_T1_NO_PAIR_HALVES = 64, DBG_T1_NO_QUADS = 128, DBG_FWD_PAIR_LOADS = 256, DBG_FWD_OCC4 = 512, DBG_T1_OCC2 = 1024, DBG_T1_NO_BINNING = 2048 /* hashed levels through global atomics as well */, DBG_NO_STREAM_OVERLAP = 4096, DBG_K2_EAGER = 8192 /* evaluate every marched sample like the reference */, DBG_K3_THREAD_PER_RAY = 32 /* the reference's sequential per-ray loops */,
	DBG_BIN_NO_HASHED_MERGE = 65536 /* k_grad_bin sums same-cell runs before the sort on the dense levels only (hashed levels: one record per sample and corner): bin + accumulate 150 -> 161 us (profiles/r02_microbench_bin_merge.log) */, DBG_NO_GRAD_ZERO_IN_OPTIMIZER = 131072 /* separate gradient memset per step */,
	DBG_K1_CHUNK_MARCH = 33554432 /* single cascade + constant step: the chunk kernels k1_count<8, true> / k1_write (production up to round 4a: every lattice point up to the ray's exit is evaluated, 64 per iteration) instead of k1_count_segments / k1_write_list */,
	DBG_BIN_NO_DENSE_MERGE = 16777216 /* no run merging on the dense levels either: their lists overflow into the atomics fallback (0.16 -> 0.81 ms) */,
	DBG_T1_DENSE_ATOMICS = 8388608 /* dense levels' gradients as merged half atomics issued by T1 (rounds 1-2a) instead of through k_grad_bin / k_grad_accumulate: T1 183 -> 89 us, bin + accumulate 123 -> 173 us (profiles/r02_microbench_dense_bins.log) */,
	DBG_GRID_NO_SORT = 4194304 /* occupancy-grid update evaluates its samples in generation order (round-1 behaviour) */,
	DBG_K1_NO_PREFILTER = 2097152 /* k1_count without the coarse-occupancy prefilter in LDS */,
	DBG_K3_TWO_PASS = 1048576 /* K3 as composite pass + prefix sum + adjoint pass: deterministic (slot-ordered) compaction without span atomics, but 34 + 57 us against 72 us for the one-pass kernel (profiles/r02_k3_two_pass.txt) */,
	DBG_SEPARATE_CONTROLLER = 524288 /* batch-size controller as its own launch behind K4 (round-1 behaviour) */,
	DBG_T1_DENSE_EXTERNAL = 262144 /* dense levels' atomics issued by k_grad_dense on its own stream instead of by T1: T1 189 -> 90 us, k_grad_dense 117 us; same wall time (profiles/r02_microbench_final.log) */,
	DBG_W_SINGLE_ROLE = 32768 /* round-1 weight-gradient kernel: one wave per SIMD holds all 12 dW tiles */,
	DBG_K3_GENERIC = 1073741824 /* K3 without the instance specialised for train_mode Nerf + no depth supervision (rounds 1-3a: one kernel for all modes) */,
	DBG_T1_NO_K2_STASH = 536870912 /* T1 gathers its encodings from the hash tables again (rounds 1-3a) instead of reading the ones K2 left behind for the same samples */,
	DBG_K1_NO_FIRST_POINT_SKIP = 268435456 /* k1_count without the one-test-per-chunk rejection of the chunks behind the ray's exit */,
	DBG_K3_ONE_RAY_PER_WAVE = 134217728 /* K3 with one wavefront per ray (rounds 1-2) instead of two rays per wavefront */,
	DBG_K4_ZERO_PADDING = 67108864 /* test hook: the rows K4 pads the compacted batch with carry a zero loss gradient instead of the rescaled copy (the padding is the only part of a step that is not linear in the set of rays: tests/test_gpu_dist.py compares the 2-rank sum with the 1-rank gradient without it) */,
	DBG_K1_MIP_CLAMP_MIN_MAX = 2147483648u /* ablation of the round-4 decision on mip_from_dt's crossed-bounds clamp (nerf_device.cuh:459): GLSL's min(max()) -- the march never leaves max_cascade -- instead of tcnn's lower-bound-first conditional; only differs with cone_angle > 0 (profiles/r04_ab_psnr_fox_clamp.json) */,
	DBG_K1_INDEPENDENT_LATTICE = 16384 /* lattice K1 without the exact skip rule: every lattice point tested on its own (round-1 behaviour; exact only for cone_angle == 0) */ };

// Device-resident NerfCounters (testbed.h / testbed_nerf.cu:2669-2702) + per-step scratch counters.
struct TrainCounters {
	uint32_t rays_per_batch;                        // R of the current/next step (global, all ranks)
	uint32_t max_inference;                         // K1 sample cap (testbed_nerf.cu:3055-3060)
	uint32_t numsteps_counter;                      // K1: marched samples (may overshoot max_inference)
	uint32_t numsteps_counter_compacted;            // K3: compacted samples (may overshoot B)
	uint32_t ray_counter;                           // K1: rays that produced samples
	uint32_t n_inference;                           // min(numsteps_counter, max_inference)
	uint32_t n_valid_compacted;                     // min(numsteps_counter_compacted, B)
	uint32_t training_step;
	uint32_t measured_batch_size;
	uint32_t measured_batch_size_before_compaction;
	uint32_t n_rays_last;
	float loss_sum;                                 // sum over rays of mean_loss / n_rays
	float loss_scalar;
	uint32_t ema_step;                              // density_grid_ema_step
	uint64_t total_rays;
	uint64_t total_samples;
	uint32_t k2_tiles[8];                           // lazy K2: number of tiles of rounds 1..7 ([0] unused: round 0 = one tile per active ray)
	uint32_t k2_samples, k2_samples_last;           // network evaluations performed by K2 in this / the previous step (statistics)
	uint32_t k4_ticket;                             // K4's workgroup ticket (the last one runs the controller); zero between launches
	uint32_t rays_per_batch_last;                   // R of the step the controller last closed (n_rays_last's companion: the extra-dims gradient maps that step's rays to images)
};

// error-proportional sampling of the training pixels (testbed.h:756 error_map; nerf_device.cuh:497-599): all null = the uniform stream
struct ErrorCdf { const float* x_cond_y = nullptr; const float* y = nullptr; const float* img = nullptr; int32_t res[2] = {0, 0}; };
struct K1Args {
	uint32_t n_rays; const uint32_t* n_rays_ptr;
	uint32_t rank, world_size;
	ngp_aabb aabb;
	uint32_t max_samples; const uint32_t* max_samples_ptr;
	ngp_pcg32 rng;
	uint32_t* ray_counter; uint32_t* numsteps_counter;
	uint32_t* ray_indices_out; ngp_ray* rays_out; uint32_t* numsteps_out; float* coords_out;
	uint32_t n_images; const ngp_image_meta* metadata; const ngp_xform* xforms;
	const uint8_t* bitfield; uint32_t max_mip;
	const uint8_t* bitfield_linear; // optional x-major copy (launch_build_linear_bitfield) for the lattice marcher
	uint32_t clamp_min_max = 0;                // ablation DBG_K1_MIP_CLAMP_MIN_MAX
	uint32_t n_mips_lds = 1; // k1_count: levels of bitfield_coarse it keeps in LDS (set by the launcher)
	uint32_t n_mips = 1; // bitfield levels present in bitfield_linear / bitfield_coarse (N_CASCADES: mip_from_dt may ask for a pooled level above max_mip, nerf_device.cuh:459)
	const uint32_t* bitfield_coarse = nullptr; // optional: k1_prefilter_words(n_mips) words from launch_build_linear_bitfield: one bit per 4x4x4 cells of every level of the x-major copy, then the dilated mid grid of level 0 (2x2x2 cells per bit): the marchers' LDS prefilters
	uint32_t no_first_point_skip = 0;          // ablation DBG_K1_NO_FIRST_POINT_SKIP: k1_count evaluates every chunk of a group, also those behind the ray's exit from the box (rounds 1-2)
	uint32_t chunk_march = 0;                  // ablation DBG_K1_CHUNK_MARCH: the chunk kernels also for one cascade + constant step
	uint4* k2_tiles0_out; uint32_t k2_tile_w; // optional: round-0 tile list of the lazy K2 (one descriptor per active ray: its first k2_tile_w samples)
	int snap_to_pixel_centers; float cone_angle_constant;
	int exact_skip; // lattice K1: follow the reference's skip rule (advance_to_next_voxel) over the lattice instead of testing every point on its own
	// optional (lattice K1 only): per-ray training target {rgbtarget[3], background[3], 0, 0} for K3, computed by the thread-per-ray
	// setup kernel so that K3's wavefronts do not all repeat it (same arithmetic as compute_loss_kernel_train_nerf)
	float* ray_targets_out; float background_color[3]; int color_space_srgb, random_bg_color, linear_colors;
	float depth_lambda = 0.f; // > 0: the ray's target depth (testbed_nerf.cu:1027) goes into slot 6 of its target record
	ErrorCdf cdf;             // testbed_nerf.cu:3152-3155: x_cond_y / y set = sample_focal_plane_proportional_to_error, img set = sample_image_proportional_to_error
	int plain_dataset = 0;    // host-checked: every image has 8-bit pixels, a Perspective / OpenCV lens and a still camera (start == end, no rolling shutter): k1_setup's small instance
	const float* extra_dims = nullptr; uint32_t n_extra = 0; // testbed_nerf.cu:718-719, 744: n_extra floats per image, copied behind every NerfCoordinate of the image's rays (coords_out stride = 7 + n_extra)
};

struct K3Args {
	uint32_t n_rays; const uint32_t* n_rays_ptr;
	ngp_aabb aabb; ngp_pcg32 rng;
	uint32_t max_samples_compacted; const uint32_t* rays_counter;
	float loss_scale; float background_color[3];
	int color_space_srgb, random_bg_color, linear_colors;
	uint32_t n_images; const ngp_image_meta* metadata;
	const ngp_half* network_output; uint32_t output_stride;
	uint32_t* numsteps_counter_compacted;
	const uint32_t* ray_indices_in; const ngp_ray* rays_in; uint32_t* numsteps_inout;
	const float* coords_in; float* coords_out;
	ngp_half* dloss_doutput; uint32_t dloss_stride;
	int loss_type; float* loss_output; // ONE float accumulator (sum over rays)
	int rgb_activation, density_activation, snap_to_pixel_centers;
	const float* mean_density_ptr; float near_distance;
	const float* ray_targets; // optional: K1Args::ray_targets_out (8 floats per active ray)
	int train_mode;           // ETrainMode: 0 Nerf, 1 Rfl, 2 RflRelax (fused_kernels/train_nerf.cuh:391-410)
	float depth_lambda = 0.f; int depth_loss_type = NGP_LOSS_L1; // depth supervision (testbed_nerf.cu:1027-1029, 1126-1129); ray_targets slot 6 = target depth (<= 0: none)
	ErrorCdf cdf;             // must equal K1's: the ray's pixel is re-derived from its index (testbed_nerf.cu:955-961); the loss is divided by the pixel's density (:1024)
	float* error_map = nullptr; int32_t error_map_res[2] = {0, 0}; // testbed_nerf.cu:1042-1071: every ray's mean loss, splatted bilinearly into its image's error map (float atomics)
	uint32_t* src_index_out = nullptr; // optional: src_index_out[compacted row] = index of the row's sample in coords_in / network_output (for T1, see EncStashIn)
	uint32_t cstride = 7;       // floats per NerfCoordinate in coords_in / coords_out (7 + n_extra_dims: the extra dims travel with the compacted samples)
	void* k3_scratch = nullptr; // k3_scratch_bytes(max_rays), initialised by k3_scratch_init: needed by the two-pass kernel (DBG_K3_TWO_PASS)
};
// compute_extra_dims_gradient_train_nerf (testbed_nerf.cu:1293-1330) and the per-image VarAdamOptimizer::step (adam_optimizer.h:37-47, testbed_nerf.cu:2860-2878)
void launch_extra_dims_gradient(hipStream_t s, uint32_t max_rays, const uint32_t* n_rays_total_ptr, const uint32_t* rays_counter, float* extra_grad, uint32_t n_extra, uint32_t n_images,
	const uint32_t* ray_indices, const uint32_t* numsteps, const float* dextra, uint32_t max_rows, const float* cdf_img);
void launch_extra_dims_adam(hipStream_t s, uint32_t n, float* variable, const float* gradient, float* m, float* v, uint32_t iter, float lr, float loss_scale,
	uint32_t* iters = nullptr /* per-image iteration counts before this step (n / n_extra of them; advanced behind the step); null: `iter` for every image */, uint32_t n_extra = 0);
size_t k3_scratch_bytes(uint32_t max_rays);
int k3_scratch_init(hipStream_t s, void* scratch, uint32_t max_rays);

void launch_generate_training_samples(hipStream_t s, const K1Args& a, uint32_t max_rays_this_rank);
// per-ray state of the sample-parallel K1 (k1_setup -> k1_count -> scan -> k1_write)
struct RaySetup { float o[3]; float d[3]; float startt; float nprime; uint32_t count; uint32_t flags; float tgt[7]; uint32_t ray_index; float rdn[3]; uint32_t img; }; // flags: from k1_setup = number of lattice points inside the box (a prefix of the lattice; 0 = the ray is not marched); tgt = {rgb target, background, target depth}; rdn = normalize(d)
constexpr uint64_t K1_SCRAMBLE_PRIME = 2654435761ull; // prime (Knuth's multiplicative-hash constant), larger than every ray count => coprime to it, and well mixed modulo powers of two; see k1_setup
size_t k1_lattice_scratch_bytes(uint32_t max_local_rays);
int k1_lattice_scratch_init(hipStream_t s, void* scratch, uint32_t max_local_rays); // once per allocation (and whenever max_local_rays changes)
void launch_generate_training_samples_lattice(hipStream_t s, const K1Args& a, uint32_t max_local_rays, void* scratch, bool count_only = false /* experiment: setup + count without the write pass */);
void launch_build_linear_bitfield(hipStream_t s, const uint8_t* bitfield, uint8_t* linear, uint32_t n_cascades, uint32_t* coarse = nullptr);
void launch_compute_loss(hipStream_t s, const K3Args& a, uint32_t max_rays);
// testbed_nerf.cu:2795-2847: error map (n_images x height x width) -> cdf_x_cond_y (same shape), cdf_y (n_images x height), cdf_img (n_images)
void launch_construct_error_cdfs(hipStream_t s, uint32_t n_images, uint32_t width, uint32_t height, const float* error_map, float* cdf_x_cond_y, float* cdf_y, float* cdf_img);
void launch_fill_rollover(hipStream_t s, uint32_t n_elements, const uint32_t* n_input_ptr, float* coords, uint32_t cstride, ngp_half* dloss, uint32_t dstride,
	const uint32_t* publish_src2 = nullptr, uint32_t* publish_dst2 = nullptr, TrainCounters* ctl = nullptr /* run the batch-size controller behind the fill */, uint32_t ctl_world_size = 1,
	const float* publish_loss = nullptr /* this rank's loss sum -> publish_dst2[2], unsigned fixed point in units of 2^-24 */);
void launch_mark_untrained(hipStream_t s, uint32_t n, float* grid, uint32_t n_images, const ngp_image_meta* m, const ngp_xform* x, int clear);
void launch_generate_grid_samples(hipStream_t s, uint32_t n, ngp_pcg32 rng, const uint32_t* step_ptr, uint32_t step, ngp_aabb box, const float* grid_in,
	float* pos, uint32_t* idx, uint32_t n_cascades, float thresh);
// sort_util.hip: the update's samples sorted by (cascade, Morton cell) so that the density network sees spatially coherent positions
size_t grid_sample_sort_temp_bytes(uint32_t n);
int grid_sample_sort(hipStream_t s, void* temp, size_t temp_bytes, const uint32_t* idx_in, uint32_t* idx_out, const float* pos_in, float* pos_out, uint32_t n, uint32_t begin_bit, uint32_t end_bit);
void launch_splat_grid_samples(hipStream_t s, uint32_t n, const uint32_t* idx, const ngp_half* out, uint32_t stride, float* grid, int act);
void launch_ema_grid_samples(hipStream_t s, uint32_t n, float decay, float* grid_out, const float* grid_in);
void launch_grid_mean(hipStream_t s, const float* grid, float* partial256, float* mean_out);
void launch_grid_to_bitfield(hipStream_t s, const float* grid, uint32_t max_cascade, uint8_t* bitfield, const float* mean_ptr);
void launch_update_counters(hipStream_t s, TrainCounters* c, uint32_t B, uint32_t world_size);
void launch_clamp_compacted(hipStream_t s, TrainCounters* c, uint32_t B);

// ---- model (model_kernels.hip) ----------------------------------------------------------------
constexpr uint32_t MAX_LEVELS = 16;
struct GridMeta {           // per-level constants, passed by value to kernels
	uint32_t n_levels, F;
	uint32_t offset[MAX_LEVELS + 1]; // entries
	uint32_t hashmap_size[MAX_LEVELS];
	uint32_t resolution[MAX_LEVELS];
	float scale[MAX_LEVELS];
};

// MFMA-fragment-ordered copies of the MLP weights (see model_kernels.hip header).
constexpr uint32_t N_FW_FRAGS = 24;  // forward A-fragments: 4 + 4 + 4 + 8 + 4   (two hidden colour layers: configs/nerf/base.json, and the image / SDF model)
constexpr uint32_t N_BW_FRAGS = 20;  // dgrad A-fragments:   4 + 2 + 4 + 8 + 2
constexpr uint32_t n_fw_frags(uint32_t n_rgb_hidden) { return 16 + 8 * (n_rgb_hidden - 1); } // NR hidden colour layers: NR - 1 layers of 64 x 64 (8 fragments each)
constexpr uint32_t n_bw_frags(uint32_t n_rgb_hidden) { return 12 + 8 * (n_rgb_hidden - 1); }
constexpr uint32_t n_mlp_params(uint32_t n_rgb_hidden) { return 64 * 32 + 16 * 64 + 64 * 32 + 64 * 64 * (n_rgb_hidden - 1) + 16 * 64; }
constexpr uint32_t FRAG_HALFS = 64 * 8;

struct ModelPtrs {
	const ngp_half* grid;       // hash table (params or inference params)
	const ngp_half* fw_frags;   // n_fw_frags(n_rgb_hidden) * 512 halfs
	const ngp_half* bw_frags;   // n_bw_frags(n_rgb_hidden) * 512 halfs
	uint32_t n_rgb_hidden = 2;  // hidden layers of the colour network: 1, 2 (base.json) or 3
	uint32_t n_extra = 0;       // extra dims behind every NerfCoordinate (nerf_network.h:84); > 0: two more forward / four more dgrad fragments behind the regular ones
};

// lazy (front-to-back) K2 in rounds of 32-sample tiles (one tile = 32 consecutive samples of ONE ray): round r evaluates samples
// [32r, 32r+32) of the rays that are still transparent after round r-1; the last round takes everything that is left.
// Tile descriptor = {first sample, valid lanes, ray, samples of the ray behind this tile}.  Round 0's list is written by K1
// (one tile per active ray, tile index = ray slot); a tile of round r appends its ray's next tile(s) to round r+1's list.
constexpr uint32_t K2_ROUNDS = 8; // maximum number of rounds
struct K2LazyArgs {
	const uint32_t* n_rays_ptr;     // active rays (K1's ray counter) = number of round-0 tiles
	uint4* tiles[2]; uint32_t tile_cap; // ping-pong lists: round r reads tiles[r & 1] and appends to tiles[(r + 1) & 1]
	uint32_t* n_tiles_ptr /* [K2_ROUNDS], [0] unused */; uint32_t* n_eval_ptr;
	float* T_run;                   // per active ray: transmittance behind the evaluated samples
	int density_activation; float dt_unwarp_scale, dt_unwarp_offset; // dt = warped * scale + offset (unwarp_dt)
	uint32_t round, n_rounds;
	uint32_t tile_w;                // samples per tile: 16 (two rays' tiles per wavefront) or 32
	uint4* enc_out = nullptr;       // optional: the encoding of every evaluated sample, 4 x 16 bytes per sample at [sample][hi][k-step] (the B-operand registers of its two lanes), for T1
};
void launch_inference_lazy(hipStream_t s, const GridMeta* gm_dev, const ModelPtrs& mp, const float* in, uint32_t in_stride, uint32_t max_rays, uint32_t max_samples,
	ngp_half* out, uint32_t out_stride, uint32_t dir_offset, const K2LazyArgs& la, uint32_t n_features = 4);
void launch_inference(hipStream_t s, const GridMeta* gm_dev, const ModelPtrs& mp, const float* in, uint32_t in_stride, uint32_t n_max, const uint32_t* n_ptr,
	ngp_half* out, uint32_t out_stride, bool density_only, uint32_t dir_offset, uint32_t n_features = 4 /* F: 4 (L = 8) or 2 (L = 16) */);
// encoding (D = 2 / 3, L = 16, F = 2) + MLP 32 -> 64 -> 64 -> 16, forward only (image / SDF primitives' model)
void launch_encmlp_inference(hipStream_t s, const GridMeta* gm_dev, uint32_t n_pos_dims, const ngp_half* grid, const ngp_half* fw_frags, const float* in, uint32_t in_stride,
	uint32_t n, ngp_half* out, uint32_t out_stride, uint32_t n_out);
// training step of the image / SDF model (k_encmlp_train_fwd_bwd -> k_encmlp_wgrad -> k_encmlp_wgrad_reduce)
struct EncTrainArgs {
	const GridMeta* gm; const __half* table; const ngp_half* fw_frags; const ngp_half* bw_frags;
	const float* in; uint32_t in_stride, n;
	const float* target; uint32_t target_stride, n_out;     // internal loss: targets (n_out floats per sample) ...
	int loss_type; float loss_scale;
	const ngp_half* dy_in; uint32_t dy_stride;              // ... or an external dL/dy (n_out halfs per sample used)
	__half* grid_grad;                                      // encoding gradient table (nullptr: the encoding is not trained)
	uint4* enc_stash; uint2* dy_stash;                      // for the weight-gradient kernel
	float* loss_sum; ngp_half* pred_out; uint32_t pred_stride; // optional: sum of the per-element loss values, network outputs
	uint32_t* denc_lv = nullptr; uint32_t denc_cap = 0;     // non-null: dL/d(enc) is left level-major (one half2 per (level, sample)) for the record lists instead of being scattered with atomics
};
void launch_encmlp_train(hipStream_t s, const EncTrainArgs& a, uint32_t n_pos_dims, bool external_dy, float* wgrad_partials, uint32_t n_partials, ngp_half* mlp_grad);
void launch_encode_only(hipStream_t s, const GridMeta* gm_dev, const ngp_half* grid, const float* pos, uint32_t stride, uint32_t n, ngp_half* out, uint32_t n_features = 4);
void launch_build_frags(hipStream_t s, const ngp_half* mlp_params, uint32_t n_mlp, const uint32_t* fw_perm, const uint32_t* bw_perm, ngp_half* fw, ngp_half* bw);
uint32_t wgrad_n_partials();

// binned scatter of the hashed levels (model_kernels.hip)
constexpr uint32_t GRAD_BIN_MAX_TABLE_LOG2 = 19; // hashmap sizes up to 2^19 (2^12 .. 2^19)
// chunk_log2: table entries per chunk (2^12: 128 KiB of 64-bit accumulators for the four features of an entry, one block per CU;
// 2^11: 64 KiB, two blocks per CU).  split: round-1 layout, one block per (chunk, feature pair) -- both blocks fetch every record.
struct AdamArgs {
	uint64_t n_params, n_mlp; // the sweep covers [range_begin, n_params)
	uint64_t range_begin = 0; int ema_only = 0; // sharded data-parallel step: a sub-range of the parameters; ema_only = EMA / inference copy + gradient clearing only (the range's Adam step ran on its owner rank)
	float loss_scale, lr, beta1, beta2, eps, l2_reg, log_beta1, log_beta2;
	int optimize_matrix, optimize_non_matrix;
	int zero_grid_grads; // leave the hash-grid gradients zeroed for the next step's scatter (GradientMode::Overwrite without a memset launch)
	float ema_decay, ema_debias_old, ema_debias_new;
	int ema_full_precision = 0; // [tcnn EmaOptimizer "full_precision"] 0 (default): the EMA state is the half inference buffer; 1: fp32 state `ema` fed with the fp32 masters
	float* master; ngp_half* params; ngp_half* params_inf; ngp_half* grads;
	float* m; float* v; uint16_t* steps /* per-parameter Adam step counters, saturating */; float* ema;
	const uint32_t* fw_perm; const uint32_t* bw_perm; // n_mlp-entry scatter tables into the fragment buffers (0xFFFFFFFF = none)
	ngp_half* fw_frags; ngp_half* bw_frags; ngp_half* fw_frags_inf; 
};
struct GradBinArgs {
	const GridMeta* gm; const float* in; uint32_t in_stride, n;
	const void* denc_lv; uint32_t denc_cap; // level-major dL/d(enc): F halfs per (level, sample)
	uint32_t levels[MAX_LEVELS]; uint32_t n_hashed, max_chunks, cap;
	uint32_t chunk_log2, split, merge_runs, no_dense_merge;
	uint32_t n_features; // F: 4 (8-byte record values) or 2 (4-byte)
	uint32_t n_pos_dims = 3; // 3, or 2 for the image primitive's grid (F = 2 only)
	void* vals; uint16_t* idxs; uint32_t* cursors; uint32_t* cursor_done; ngp_half* grid_grad_;
	uint32_t acc_ly_begin = 0, acc_ly_count = 0; // k_grad_accumulate: the listed levels it covers (count 0 = all); the sharded data-parallel step accumulates bucket by bucket
	// Single-GPU steps: k_grad_accumulate applies the optimizer to the HASHED levels in its epilogue -- the chunk's gradient sums are in LDS, so the 23 MB gradient write,
	// its re-read by the sweep and the sweep's pass over those levels disappear (same arithmetic: the sums are rounded to half exactly as the stored gradient would be).
	// The sweep (k_optimizer) then covers parameters [0, adam.n_params) = MLP + dense levels only.  Off (0): gradients are written for a separate optimizer step
	// (C-ABI callers that look at gradients, the data-parallel all-reduce).
	uint32_t fuse_adam = 0; AdamArgs adam{};
};
void launch_grad_bin(hipStream_t s, const GradBinArgs& a, uint32_t what = 3u /* bit 0: k_grad_bin, bit 1: k_grad_accumulate */);
struct GradDenseArgs { // k_grad_dense: the dense levels' scatter from T1's level-major dL/d(enc)
	const GridMeta* gm; const float* in; uint32_t in_stride, n;
	const uint2* denc_lv; uint32_t denc_cap;
	uint32_t levels[MAX_LEVELS]; uint32_t n_levels, merge_runs;
	ngp_half* grid_grad_;
};
void launch_grad_dense(hipStream_t s, const GradDenseArgs& a);
constexpr uint32_t T1_DENSE_EXTERNAL = 1u << 31; // launch_train_fwd_bwd flag (not a debug flag): dense levels' dL/d(enc) goes to denc_lv too, no scatter in T1
// Encodings left behind by the lazy K2 (K2LazyArgs::enc_out) for the samples T1 is about to differentiate: src_index[j] = K2's sample of batch row j (written by
// K3 next to the row), n_valid_ptr = rows K3 produced (rows behind it are K4's wrap-around copies of row j % n_valid).  T1 then reads 64 contiguous bytes per sample
// instead of gathering 64 table entries through 128-byte lines (306 MB of line traffic per step, profiles/r03_pmc_summary.txt).
struct EncStashIn { const uint4* enc = nullptr; const uint32_t* src_index = nullptr; const uint32_t* n_valid_ptr = nullptr; };
void launch_train_fwd_bwd(hipStream_t s, const GridMeta* gm_dev, const ModelPtrs& mp, const float* in, uint32_t in_stride, uint32_t n,
	const ngp_half* dL_dy, uint32_t dy_stride, ngp_half* grid_grad, ngp_half* enc_stash, uint32_t flags, void* denc_lv, uint32_t denc_cap, uint32_t n_features = 4,
	const EncStashIn* stash_in = nullptr, float* dextra_out = nullptr /* models with extra dims: optional dL/d(extra dims), n x n_extra floats */);
void launch_wgrad(hipStream_t s, const ModelPtrs& mp, const float* in, uint32_t in_stride, uint32_t n, const ngp_half* dL_dy, uint32_t dy_stride,
	const ngp_half* enc_stash, float* wgrad_partials, uint32_t n_partials);
void launch_wgrad_reduce(hipStream_t s, const float* partials, uint32_t n_partials, ngp_half* mlp_grad, uint32_t n_rgb_hidden = 2, uint32_t n_extra = 0);
void launch_train_fused(hipStream_t s, const ModelPtrs& mp, const float* in, uint32_t in_stride, uint32_t n, const ngp_half* dL_dy, uint32_t dy_stride,
	const EncStashIn& stash_in, void* denc_lv, uint32_t denc_cap, float* wgrad_partials, uint32_t n_partials);

void launch_optimizer_step(hipStream_t s, const AdamArgs& a);

// ---- image primitive (image_kernels.hip) ------------------------------------------------------
struct ImageBatchArgs {
	const void* pixels; int image_data_type; int width, height;
	uint32_t n; ngp_pcg32 rng; uint32_t stratify_log2; // 0 = plain uniform positions
	int snap_to_pixel_centers, linear_colors;
	float* positions; float* targets; // vec2 / vec3 per sample
	float* zero_word = nullptr;       // optional: one word this launch clears (the trainer's loss sum of the step that follows: no memset launch)
};
void launch_image_generate_batch(hipStream_t s, const ImageBatchArgs& a);
void launch_image_pixel_batch(hipStream_t s, const ImageBatchArgs& a, uint32_t offset);
void launch_image_mse(hipStream_t s, uint32_t n, const float* targets, const ngp_half* pred, uint32_t pred_stride, int quantize, double* sum);

// ---- SDF primitive (sdf_kernels.hip) -----------------------------------------------------------
struct SdfTriangle { float a[3], b[3], c[3]; };                       // Triangle, triangle.cuh:37
struct SdfBvhNode { float bmin[3], bmax[3]; int left, right; };        // inner node: child indices; leaf: left = -first - 1, right = -end - 1 (TriangleBvhNode convention)
struct SdfSampleArgs {
	uint32_t n, n_exact, n_surface;   // samples [0, n_exact) on the surface, [n_exact, n_surface) surface + offset, [n_surface, n) uniform
	ngp_pcg32 rng; float stddev; ngp_aabb aabb;
	const float* cdf; uint32_t n_triangles; const SdfTriangle* triangles;
	float* positions; float* distances;
};
void launch_sdf_generate_positions(hipStream_t s, const SdfSampleArgs& a);
// device form of the tree: 4-wide nodes of one 128-byte line -- the boxes of up to four children (component-major: lo[axis][child]) and their references: an inner node's
// index (>= 0), a leaf ~((first triangle << 3) | count) with count <= 4, or 0x7fffffff for an empty slot (box [+inf, -inf])
struct SdfBvhNode4 { float lo[3][4], hi[3][4]; int ref[4]; int pad[4]; };
static_assert(sizeof(SdfBvhNode4) == 128, "SdfBvhNode4 is one cache line");
// scratch of one ground-truth call over <= cap points (allocated by ngp_sdf_create)
struct SdfQueryScratch {
	uint32_t* survivors;   // cap: points none of whose first stab rays escaped
	uint32_t* escaped;     // cap: per-point mark of the first stab rays, zero on entry and on exit
	uint32_t* n_survivors; // one word
	uint32_t* keys; uint32_t* keys_sorted; uint32_t* idx; uint32_t* order; // cap each: (cost class, Morton) keys of the points, identity, and the sorted order
	void* sort_temp; size_t sort_temp_bytes;
	float* stab_offsets;   // cap x 2: the stab rays' lattice offsets of point i (a function of i alone)
	uint32_t* work_ctr;    // 2 words: items reserved from the persistent kernel's two lists, zero between calls
};
size_t sdf_point_sort_temp_bytes(uint32_t n);
void launch_sdf_stab_offsets(hipStream_t s, uint32_t n, float* offsets);
int sdf_point_sort(hipStream_t s, void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* idx_in, uint32_t* idx_out, uint32_t n);
int launch_sdf_signed_distance(hipStream_t s, uint32_t n, const float* positions, float* distances, const SdfBvhNode4* nodes, int root, uint32_t stack_entries /* 3 * depth + 1 */,
		const SdfTriangle* tris, int use_upper_bounds, const SdfQueryScratch& q, uint32_t groups = 1 /* the launch covers `groups` x n points: point j of group b at element b * group_stride + j */, uint32_t group_stride = 0);
void host_sdf_signed_distance(uint32_t n, const float* positions, float* distances, const SdfBvhNode4* nodes, int root, const SdfTriangle* tris, int use_upper_bounds); // test hook, host
void launch_sdf_compare_signs(hipStream_t s, uint32_t n, const float* ref, const ngp_half* model, uint32_t model_stride, uint32_t* counters);

// ---- renderer (render_kernels.hip) ------------------------------------------------------------
constexpr uint32_t RENDER_MAX_CHUNKS = 32; // 2048 lattice points per ray
constexpr uint32_t RENDER_STEPS = 8;       // samples per live ray between compactions (MAX_STEPS_INBETWEEN_COMPACTION, testbed_nerf.cu:53)
struct RenderRay {  // NerfPayload (nerf_device.cuh:145-153) + accumulators
	float o[3], d[3];
	float startt, nprime;
	float rgba[4];
	float depth, max_weight;
	uint32_t cursor, n_chunks, alive, n_emitted;
};
struct RenderArgs {
	ngp_render_params p;
	ngp_aabb train_aabb;
	const uint8_t* bitfield;
	uint32_t max_mip;
	float cone_angle;
	int rgb_activation, density_activation, linear_colors;
	RenderRay* rays;
	uint64_t* masks;
	const float* extra_dims = nullptr; uint32_t n_extra = 0; // the rendering's extra dims, written behind every NerfCoordinate (testbed_nerf.cu:211-213, 463-465)
};
void launch_render_setup(hipStream_t s, const RenderArgs& a, uint32_t pixel_begin, uint32_t n);
void launch_render_compact(hipStream_t s, const RenderArgs& a, uint32_t n, uint32_t* alive_list, uint32_t* n_alive);
void launch_render_emit(hipStream_t s, const RenderArgs& a, uint32_t n_alive_host, const uint32_t* alive_list, const uint32_t* n_alive, float* coords);
void launch_render_composite(hipStream_t s, const RenderArgs& a, uint32_t n_alive_host, const uint32_t* alive_list, const uint32_t* n_alive, const float* coords, const ngp_half* net_out);
void launch_render_finish(hipStream_t s, const RenderArgs& a, uint32_t pixel_begin, uint32_t n, float* frame, float* depth);
void launch_render_accumulate(hipStream_t s, uint32_t n_floats, const float* frame, float* accum, float weight);
void launch_render_tonemap(hipStream_t s, uint32_t n_pixels, float* rgba, float exposure_scale, const float bg[4], int to_srgb, int curve);

// ---- optional per-kernel HIP-event timing (bench.py roofline leg) -----------------------------
enum ProfId { P_K1 = 0, P_K2_INFERENCE, P_K3, P_K4, P_T1_FWD_BWD_SCATTER, P_W_WGRAD, P_WGRAD_REDUCE, P_OPTIMIZER, P_GRID_DENSITY, P_GRID_MISC, P_GRAD_MEMSET, P_COUNTERS, P_GRAD_BIN, P_K2_ENCODE_XCD, P_TRAIN_FUSED, P_COUNT };

} // namespace ngp
