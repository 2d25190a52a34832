/*
 * ngp_hip.h -- C-ABI of libngp_hip.so, the MI355X (gfx950) replacement for the NeRF
 * training/rendering hot path of NVlabs/instant-ngp.
 *
 * The reference has no C boundary on this path: `ngp::Testbed` calls C++ virtual classes of
 * tiny-cuda-nn (`Encoding<T>`, `Network<float,T>`, `Trainer`, `Optimizer`) plus its own CUDA
 * kernels, all on one `cudaStream_t`.  Every entry point below names the reference interface
 * (file:line under /root/reference) it replaces.  Conventions:
 *   - plain C: opaque handles, POD structs, raw pointers + sizes; no torch / STL types;
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`;
 *   - `stream` is a `hipStream_t` passed as `void*` (NULL = the null stream); calls enqueue
 *     work and return, exactly like the reference's `(cudaStream_t stream, ...)` methods;
 *   - return value: 0 on success, non-zero on failure; `ngp_last_error()` holds the message
 *     (the C++ host turns it back into the `std::runtime_error` the reference would throw);
 *   - a handle is thread-compatible (one thread at a time), like `ngp::Testbed`.
 *   - `ngp_half` is IEEE binary16 stored as uint16_t (== `network_precision_t` = `__half`,
 *     reference CMakeLists.txt:302).
 */
#ifndef NGP_HIP_H
#define NGP_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t ngp_half;

/* ------------------------------------------------------------------ POD mirrors ---------- */

/* BoundingBox, include/neural-graphics-primitives/bounding_box.cuh:247-248 */
typedef struct ngp_aabb { float min[3]; float max[3]; } ngp_aabb;

/* Ray {vec3 o, d}; tiny-cuda-nn vec.h (usage common_device.cuh:413-490) */
typedef struct ngp_ray { float o[3]; float d[3]; } ngp_ray;

/* pcg32 state (tiny-cuda-nn/dependencies/pcg32/pcg32.h; `default_rng_t`, random_val.cuh:26) */
typedef struct ngp_pcg32 { uint64_t state; uint64_t inc; } ngp_pcg32;

/* ELensMode, common.h:184-192 */
enum { NGP_LENS_PERSPECTIVE = 0, NGP_LENS_OPENCV = 1, NGP_LENS_FTHETA = 2, NGP_LENS_LATLONG = 3,
       NGP_LENS_OPENCV_FISHEYE = 4, NGP_LENS_EQUIRECTANGULAR = 5, NGP_LENS_ORTHOGRAPHIC = 6 };

/* EImageDataType, common_device.cuh:764-769 */
enum { NGP_IMAGE_NONE = 0, NGP_IMAGE_BYTE = 1, NGP_IMAGE_HALF = 2, NGP_IMAGE_FLOAT = 3 };

/* ENerfActivation, common.h (None/ReLU/Logistic/Exponential) */
enum { NGP_ACT_NONE = 0, NGP_ACT_RELU = 1, NGP_ACT_LOGISTIC = 2, NGP_ACT_EXPONENTIAL = 3 };

/* ELossType, common.h */
enum { NGP_LOSS_L2 = 0, NGP_LOSS_L1 = 1, NGP_LOSS_MAPE = 2, NGP_LOSS_SMAPE = 3, NGP_LOSS_HUBER = 4,
       NGP_LOSS_LOGL1 = 5, NGP_LOSS_RELATIVE_L2 = 6 };

/* TrainingImageMetadata, nerf_device.cuh:45-60 (explicit rays / light_dir: out of scope) */
typedef struct ngp_image_meta {
	const void* pixels;        /* device pointer to the image (RGBA8 sRGB, RGBA16F or RGBA32F) */
	int32_t image_data_type;   /* NGP_IMAGE_* */
	int32_t lens_mode;         /* NGP_LENS_* */
	int32_t resolution[2];
	float principal_point[2];
	float focal_length[2];
	float rolling_shutter[4];
	float lens_params[7];
	float _pad;
	const float* depth;        /* optional: one float per pixel in scene units (0 = no measurement), nerf_loader.cu:73-82; NULL = no depth for this image */
} ngp_image_meta;

/* TrainingXForm {mat4x3 start, end}, common.h:177-182; mat4x3 = 4 columns of vec3 (col-major) */
typedef struct ngp_xform { float start[12]; float end[12]; } ngp_xform;

/* Hyper-parameters of NerfNetwork + Trainer (nerf_network.h:81-101, testbed.cu:4160-4412,
 * configs/nerf/base.json). `ngp_model_config_from_json` fills this from the reference's JSON. */
typedef struct ngp_model_config {
	/* "encoding": HashGrid (tiny-cuda-nn encodings/grid.h) */
	uint32_t n_levels;              /* 8  */
	uint32_t n_features_per_level;  /* 4  */
	uint32_t log2_hashmap_size;     /* 19 */
	uint32_t base_resolution;       /* 16 */
	float per_level_scale;          /* testbed.cu:4241-4255 */
	/* "network" / "rgb_network": FullyFusedMLP */
	uint32_t n_neurons;             /* 64 */
	uint32_t n_hidden_layers;       /* density net: 1 */
	uint32_t n_hidden_layers_rgb;   /* rgb net: 2 */
	uint32_t sh_degree;             /* "dir_encoding": SphericalHarmonics degree 4 */
	uint32_t n_extra_dims;          /* NerfDataset::n_extra_dims() (nerf_loader.h:85-87): 0 for lego / fox; 3 = light directions; <= 16 */
	/* "optimizer": Ema( ExponentialDecay( Adam ) ) */
	float learning_rate, beta1, beta2, epsilon, l2_reg;
	float ema_decay;
	uint32_t decay_start, decay_interval;
	float decay_base;
	uint32_t ema_full_precision;    /* [tcnn EmaOptimizer "full_precision", default false] 0: the EMA state is the network-precision inference buffer itself
	                                   (ema_step_half_precision, fed with the half weights); 1: an fp32 state fed with the fp32 master weights (ema_step_full_precision) */
} ngp_model_config;

/* Run-time options of the NeRF trainer; defaults = the reference's member defaults. */
typedef struct ngp_nerf_options {
	int32_t rgb_activation;          /* testbed_nerf.cu:2354  Logistic for LDR data */
	int32_t density_activation;      /* testbed.h:869         Exponential */
	int32_t loss_type;               /* configs/nerf/base.json:2-4  Huber */
	int32_t random_bg_color;         /* testbed.h:793  true  */
	int32_t snap_to_pixel_centers;   /* testbed.h:797  true  */
	int32_t linear_colors;           /* testbed.h:794  false */
	int32_t color_space_srgb;        /* testbed.h:1002 0 = Linear, 1 = SRGB (--nerf_compatibility) */
	float background_color[3];       /* used when !random_bg_color */
	float near_distance;             /* testbed.h:817  0.1  */
	float density_grid_decay;        /* testbed.h:818  0.95 */
	float cone_angle_constant;       /* testbed_nerf.cu:2440 */
	uint32_t max_cascade;            /* testbed_nerf.cu:2433-2436 */
	uint32_t target_batch_size;      /* testbed.h:1089  1<<18 samples */
	float loss_scale;                /* testbed.h:311   128 */
	uint64_t seed;                   /* testbed.h:680   1337 */
	/* data-parallel sharding (new): this rank marches global rays
	 * [rank*R/world, (rank+1)*R/world) of the same global stream */
	uint32_t rank, world_size;
	/* ETrainMode (testbed.h:822; python_api.cu TrainMode): 0 Nerf, 1 Rfl, 2 RflRelax -- the radiance-field-loss gradients of
	 * fused_kernels/train_nerf.cuh:391-410, here evaluated by the unfused K1/K2/K3 pipeline (no JIT needed) */
	int32_t train_mode;
	/* depth supervision (testbed.h:796, 824; testbed_nerf.cu:1027-1029, 1126-1129): weight of the depth term (0 = off) and its loss (default L1) */
	float depth_supervision_lambda;
	int32_t depth_loss_type;
	/* training pixels drawn in proportion to the accumulated error (testbed.h:810-811; python_api.cu:795-796; nerf_device.cuh:497-599). The reference
	 * ALWAYS accumulates the error map (testbed_nerf.cu:2793 `accumulate_error = true`, for its GUI); here the map is accumulated while one of the two
	 * switches is on, or always with accumulate_error_map = 1, so that the default step does not pay for atomics nothing reads. */
	int32_t sample_focal_plane_proportional_to_error;
	int32_t sample_image_proportional_to_error;
	int32_t accumulate_error_map;
} ngp_nerf_options;

/* Counters read back by the host (NerfCounters, testbed.h / testbed_nerf.cu:2669-2702). */
typedef struct ngp_nerf_stats {
	uint32_t training_step;
	uint32_t rays_per_batch;                        /* R used by the NEXT step */
	uint32_t n_rays_last;                           /* rays that produced samples in the last step */
	uint32_t measured_batch_size;                   /* compacted samples in the last step */
	uint32_t measured_batch_size_before_compaction; /* marched samples in the last step */
	float loss;                                     /* last loss scalar (every 16 steps in the reference) */
	uint64_t total_rays;                            /* sum of rays_per_batch over all steps */
	uint64_t total_samples;                         /* sum of measured_batch_size */
	uint32_t network_evaluations;                   /* K2 samples actually evaluated in the last step (== marched samples when eager) */
	uint32_t reserved;
} ngp_nerf_stats;

typedef struct ngp_model ngp_model;      /* NerfNetwork + Trainer + optimizer state */
typedef struct ngp_nerf ngp_nerf;        /* Testbed::m_nerf state for training/rendering */

const char* ngp_last_error(void);
/* 1 when a HIP device is visible to this process. */
int ngp_device_available(void);
/* Optional: create the library's process-wide helper streams now (they are created with the first model / trainer otherwise).  A host that sets up
 * communication (RCCL / torch.distributed "nccl") BEFORE its first model calls this first: on ROCm 7.0 streams that come into existence after an RCCL
 * communicator run their kernels 1.3 - 2x slower (profiles/r03_dp_overhead.txt).  New; no counterpart in the reference. */
int ngp_init(void);
/* Box calibration: nanoseconds per dependent fp32 FMA of a single wavefront (= the shader clock a light kernel gets; bench.py config.calibration). Blocks. */
int ngp_debug_clock_probe(float* ns_per_dependent_fma_host);

/* ------------------------------------------------------------------ model ---------------- */

/* load_network_config + reset_network hyper-parameter derivation: testbed.cu:86-97, 4160-4412.
 * `aabb_scale` feeds per_level_scale (testbed.cu:4241-4255). Host-only (no device needed). */
int ngp_model_config_from_json(const char* json_host, uint32_t aabb_scale, uint32_t n_extra_dims,
                               ngp_model_config* out_host);

/* NerfNetwork ctor (nerf_network.h:81-101) + Trainer ctor (testbed.cu:4383): allocates
 * {master f32 | params f16 | inference(EMA) params f16 | gradients f16 | Adam m, v, steps | EMA f32}
 * and initialises from pcg32{seed} in the order density MLP, rgb MLP, pos enc (nerf_network.h:374-386). */
int ngp_model_create(const ngp_model_config* cfg_host, uint64_t seed, ngp_model** out);
void ngp_model_destroy(ngp_model*);

/* n_params (nerf_network.h:388-390); *n_mlp_params = "matrix" params (first in the layout). */
int ngp_model_n_params(const ngp_model*, uint64_t* n_params, uint64_t* n_mlp_params);
/* Trainer::params / params_inference / gradients device pointers (for snapshots / all-reduce).
 * On one GPU ngp_nerf_train applies the optimizer to the hashed levels inside the scatter kernel's epilogue: after such a step
 * `gradients` holds the MLP and dense-level gradients only, the hashed levels' slots are NOT written (they stay zero).  A caller
 * that reads gradients back uses ngp_nerf_train_forward_backward / ngp_model_backward (never fused), or NGP_NO_FUSED_ADAM=1. */
int ngp_model_param_ptrs(ngp_model*, float** master, ngp_half** params, ngp_half** inference_params,
                         ngp_half** gradients);
/* Per-level offsets (in entries) and hashmap sizes; MultiLevelEncoding::level_params_offset
 * (testbed.cu:4115-4117). Arrays of n_levels+1 / n_levels uint32 on the host. */
int ngp_model_grid_layout(const ngp_model*, uint32_t* offsets_host, uint32_t* resolutions_host, float* scales_host);
/* Trainer::set_params_full_precision (testbed.cu:4407): upload f32 params, refresh f16 copies. */
int ngp_model_set_params_host(ngp_model*, const float* params_host, uint64_t n);
int ngp_model_get_params_host(ngp_model*, float* params_host, uint64_t n);

/* Network::inference_mixed_precision (nerf_network.h:105-139; call sites testbed_nerf.cu:3235, 1772).
 * in: NerfCoordinate AoS, `in_stride` floats per element (7 + n_extra_dims); n_ptr (device uint32,
 * may be NULL) bounds the element count on-device so no host read-back is needed.
 * out: rgb logits in [0..2], sigma logit in [3]; `out_stride` halfs per element (>= 4). */
int ngp_model_inference(ngp_model*, void* stream, const float* in, uint32_t in_stride, uint32_t n_max,
                        const uint32_t* n_ptr, ngp_half* out, uint32_t out_stride, int use_inference_params);

/* NerfNetwork::density (nerf_network.h:270-280; call site testbed_nerf.cu:2570): grid encoding +
 * density MLP only; out[i] = raw sigma logit. `pos_stride` floats per position (NerfPosition = 3). */
int ngp_model_density(ngp_model*, void* stream, const float* pos, uint32_t pos_stride, uint32_t n,
                      ngp_half* out, uint32_t out_stride, int use_inference_params);

/* Trainer::training_step(stream, input, {}, nullptr, false, dL_dinput, false, Overwrite, &dL_dy)
 * (testbed_nerf.cu:3313-3323 -> nerf_network.h:145-268): forward + backward with an EXTERNAL output
 * gradient; parameter gradients are overwritten. dL_dy: `dy_stride` halfs per element, [0..3] used. */
int ngp_model_training_step(ngp_model*, void* stream, const float* in, uint32_t in_stride, uint32_t n,
                            const ngp_half* dL_dy, uint32_t dy_stride);
/* The same with dL_dinput requested for the extra dims (Trainer::training_step(..., dL_dinput, ...) with prepare_input_gradients,
 * testbed_nerf.cu:3309-3323; nerf_network.h:238-252: the dir encoding's backward): dL_dextra (device, n x n_extra_dims floats, may be
 * NULL) receives the extra-dims columns of dL/d(input) -- what compute_extra_dims_gradient_train_nerf (testbed_nerf.cu:1293-1330) reads
 * through coords_gradient(j)->get_extra_dims(). Models created with n_extra_dims > 0 only. */
int ngp_model_training_step_extra(ngp_model*, void* stream, const float* in, uint32_t in_stride, uint32_t n,
                                  const ngp_half* dL_dy, uint32_t dy_stride, float* dL_dextra);

/* Trainer::optimizer_step(stream, loss_scale) (testbed_nerf.cu:2770): Adam -> ExponentialDecay -> EMA. */
int ngp_model_optimizer_step(ngp_model*, void* stream, float loss_scale);
/* Optimizer::update_hyperparams (testbed.cu:4617-4623): optimize_matrix_params / non_matrix. */
int ngp_model_set_trainable(ngp_model*, int train_network, int train_encoding);
float ngp_model_learning_rate(const ngp_model*);
uint32_t ngp_model_step(const ngp_model*);

/* Trainer::serialize / deserialize payload (testbed.cu:5289, 5468): params (f32 master), and when
 * `with_optimizer` Adam m/v/steps + EMA. Layout documented in DESIGN.md. Host buffers. */
uint64_t ngp_model_serialized_size(const ngp_model*, int with_optimizer);
int ngp_model_serialize_host(ngp_model*, void* buffer_host, uint64_t size, int with_optimizer);
int ngp_model_deserialize_host(ngp_model*, const void* buffer_host, uint64_t size);
/* The payload's layout for callers that translate it into another container (Testbed::save_snapshot / load_snapshot write tcnn's per-optimizer keys, testbed.cu:5289, 5468):
 * a header, then sections of n_params x 4 bytes each in the order below; `ngp_model_state_header` reads (buffer -> out arguments, write = 0) or writes (write = 1) the
 * header fields so that no caller carries a copy of the header struct. */
enum { NGP_STATE_MASTER = 0 /* f32 */, NGP_STATE_ADAM_M = 1 /* f32 */, NGP_STATE_ADAM_V = 2 /* f32 */, NGP_STATE_ADAM_STEPS = 3 /* u32 per parameter */,
       NGP_STATE_EMA = 4 /* f32: the EMA state ("full_precision") or the half inference parameters widened */, NGP_STATE_N_SECTIONS = 5 };
uint64_t ngp_model_state_offset(uint64_t n_params, int section);
int ngp_model_state_header(void* buffer_host, uint64_t size, int write, uint64_t* n_params, uint32_t* step, float* learning_rate, uint32_t* with_optimizer);
/* the configuration the model was created with (e.g. for the snapshot writer: ema_full_precision) */
int ngp_model_get_config(const ngp_model*, ngp_model_config* out);

/* ------------------------------------------------------------------ encoding + MLP ------ */
/* The image and SDF primitives' model (configs/image/base.json, configs/sdf/base.json): a HashGrid encoding of a 2-D / 3-D
 * position feeding one FullyFusedMLP -- tcnn::NetworkWithInputEncoding + Trainer, created at testbed.cu:4354-4383, queried through
 * Network::inference at testbed_image.cu:383,541 / testbed_sdf.cu:1658 and trained through Trainer::training_step at
 * testbed_image.cu:289 / testbed_sdf.cu:1557.  The fused kernels are specialised for L*F = 32 encoded features (L = 16, F = 2),
 * 64 neurons, 2 hidden layers, <= 16 outputs (training: <= 4).  Parameter order: MLP weights (row-major [out][in] per layer,
 * output layer padded to 16 rows), then the grid [tcnn]. */
typedef struct ngp_encmlp_config {
	uint32_t n_pos_dims;            /* 2 (image uv) or 3 (SDF position), inputs in [0,1]^D */
	uint32_t n_levels;              /* 16 */
	uint32_t n_features_per_level;  /* 2  */
	uint32_t log2_hashmap_size;     /* 19 in BASELINE.json's configs (24 in configs/image/base.json) */
	uint32_t base_resolution;       /* 16 */
	float per_level_scale;          /* testbed.cu:4241-4255: image 1.25992 (max res 512 from 16), SDF 1.3819 */
	uint32_t n_neurons;             /* 64 */
	uint32_t n_hidden_layers;       /* 2  */
	uint32_t n_output_dims;         /* 3 (rgb) / 1 (distance) */
} ngp_encmlp_config;
/* "optimizer" block of a network config: [Ema(] ExponentialDecay( Adam ) [)]; ema_decay = 0 <=> no Ema wrapper
 * (configs/image/base.json:5-22, configs/sdf/base.json) */
typedef struct ngp_optimizer_config {
	float learning_rate, beta1, beta2, epsilon, l2_reg;
	float ema_decay;
	uint32_t decay_start, decay_interval;
	float decay_base;
} ngp_optimizer_config;
typedef struct ngp_encmlp ngp_encmlp;
int ngp_encmlp_create(const ngp_encmlp_config*, uint64_t seed, ngp_encmlp** out);
void ngp_encmlp_destroy(ngp_encmlp*);
int ngp_encmlp_n_params(const ngp_encmlp*, uint64_t* n_params, uint64_t* n_mlp_params);
int ngp_encmlp_param_ptrs(ngp_encmlp*, float** master, ngp_half** params, ngp_half** inference_params, ngp_half** gradients);
int ngp_encmlp_set_params_host(ngp_encmlp*, const float* params_host, uint64_t n);
int ngp_encmlp_get_params_host(ngp_encmlp*, float* params_host, uint64_t n);
int ngp_encmlp_set_optimizer(ngp_encmlp*, const ngp_optimizer_config* host);   /* defaults: configs/image/base.json */
/* Network::inference (inference parameters): in = n positions (D floats each, `in_stride` floats apart), out = n x n_output_dims halfs */
int ngp_encmlp_inference(ngp_encmlp*, void* stream, const float* in, uint32_t in_stride, uint32_t n, ngp_half* out, uint32_t out_stride);
/* Trainer::training_step(stream, input, target): forward, loss (NGP_LOSS_L2 | L1 | MAPE | RELATIVE_L2, tcnn losses/...: value and gradient
 * normalised by n * n_output_dims, gradient times loss_scale), backward; parameter gradients overwritten.  loss_sum (device float,
 * may be NULL) receives the sum of the per-element loss values; pred_out (may be NULL) the network outputs (n_output_dims halfs). */
int ngp_encmlp_training_step(ngp_encmlp*, void* stream, const float* in, uint32_t in_stride, uint32_t n, const float* target, uint32_t target_stride,
                             int loss_type, float loss_scale, float* loss_sum, ngp_half* pred_out, uint32_t pred_stride);
/* the same with an external output gradient dL/dy (n_output_dims halfs per element used), like ngp_model_training_step */
int ngp_encmlp_training_step_external(ngp_encmlp*, void* stream, const float* in, uint32_t in_stride, uint32_t n, const ngp_half* dL_dy, uint32_t dy_stride);
int ngp_encmlp_optimizer_step(ngp_encmlp*, void* stream, float loss_scale);    /* Trainer::optimizer_step */
float ngp_encmlp_learning_rate(const ngp_encmlp*);
uint32_t ngp_encmlp_step(const ngp_encmlp*);

/* ------------------------------------------------------------------ image trainer -------- */
/* Testbed::m_image + train_image (testbed_image.cu:231-302) and compute_image_mse (:490-547): the image lives on the device as RGBA
 * float32 (EDataType::Float) or half; a training batch is `batch_size` uv positions from pcg32{seed} (stratified over a
 * sqrt(batch) x sqrt(batch) grid, stratify2_kernel :66-82) and their targets (eval_image_kernel_and_snap :175-229: nearest pixel
 * when snap_to_pixel_centers, else bilinear; sRGB-encoded unless linear_colors). */
typedef struct ngp_image_options {
	int32_t snap_to_pixel_centers;   /* testbed.h:966 true  */
	int32_t linear_colors;           /* testbed.h:967 false */
	int32_t stratified;              /* testbed.h:970 ERandomMode::Stratified */
	int32_t loss_type;               /* configs/image/base.json: L2 */
	float loss_scale;                /* 128 */
	uint32_t batch_size;             /* Testbed::m_training_batch_size; BASELINE config 0: 65536 */
	uint64_t seed;                   /* 1337 */
} ngp_image_options;
typedef struct ngp_image ngp_image;
int ngp_image_create(ngp_encmlp* model, const void* pixels_rgba_host, int32_t image_data_type /* NGP_IMAGE_HALF | NGP_IMAGE_FLOAT */, int32_t width, int32_t height,
                     const ngp_image_options* opts_host, ngp_image** out);
void ngp_image_destroy(ngp_image*);
int ngp_image_train(ngp_image*, void* stream, uint32_t n_steps);               /* train_image + optimizer_step, n times, no host sync */
int ngp_image_loss(ngp_image*, void* stream, float* loss_host);               /* loss of the last step (blocking read-back) */
int ngp_image_mse(ngp_image*, int quantize_to_byte, float* mse_host);         /* compute_image_mse (blocking) */
/* device pointers of the last training batch: positions (vec2) and targets (vec3); test hook */
int ngp_image_batch_ptrs(ngp_image*, float** positions, float** targets);

/* ------------------------------------------------------------------ SDF trainer ---------- */
/* Testbed::m_sdf: load_mesh normalisation (testbed_sdf.cu:1380-1410), triangle BVH (triangle_bvh.cu) with EMeshSdfMode::Raystab
 * signed distances (32 Fibonacci stab rays, :631-650), generate_training_samples_sdf (:1449-1544: 4/8 of a batch on the surface, 3/8
 * surface + logistic offset of sigma = |(0.5,0.5,0.5)| / 1024, 1/8 uniform in the box), train_sdf (:1580-1622, MAPE loss),
 * calculate_iou (:1636-1680).  Triangles: 9 floats each (a, b, c), already normalised into the unit cube. */
typedef struct ngp_sdf_options {
	int32_t loss_type;               /* configs/sdf/base.json: MAPE */
	float loss_scale;                /* 128 */
	uint32_t batch_size;             /* 1 << 18 */
	uint64_t seed;                   /* 1337 */
	float surface_offset_scale;      /* testbed.h:938  1.0 */
	float zero_offset;               /* testbed.h:925  0   */
} ngp_sdf_options;
typedef struct ngp_sdf ngp_sdf;
int ngp_sdf_normalize_mesh_host(float* vertices_inout_host, uint64_t n_vertices, ngp_aabb* aabb_out_host, float* mesh_scale_out_host);
int ngp_sdf_create(ngp_encmlp* model, const float* triangles_host, uint32_t n_triangles, ngp_aabb aabb, const ngp_sdf_options* opts_host, ngp_sdf** out);
void ngp_sdf_destroy(ngp_sdf*);
int ngp_sdf_train(ngp_sdf*, void* stream, uint32_t n_steps);                 /* training_prep_sdf + train_sdf + optimizer_step, n times */
/* Batches whose samples + ground truth are generated ahead of the training steps, per ground-truth launch, on a side stream (default 16; 0 = the reference's serial loop
 * generate -> train, testbed_sdf.cu:1580-1635 training_prep_sdf / train_sdf).  The batches, their order and the rng positions are the serial loop's either way. */
int ngp_sdf_set_batches_ahead(ngp_sdf*, uint32_t batches);
int ngp_sdf_loss(ngp_sdf*, void* stream, float* loss_host);
int ngp_sdf_iou(ngp_sdf*, uint32_t n_samples, double* iou_host);             /* calculate_iou(n_samples, 0, blocking) */
int ngp_sdf_batch_ptrs(ngp_sdf*, float** positions, float** distances);      /* last generated batch (device); test hook */
int ngp_sdf_signed_distance(ngp_sdf*, void* stream, const float* positions, uint32_t n, float* distances_out); /* TriangleBvh::signed_distance_gpu (Raystab) */

/* ------------------------------------------------------------------ NeRF kernels --------- */
/* Stand-alone kernels (each mirrors one reference kernel; used by the parity tests and by
 * ngp_nerf_* below). All buffers are caller-owned device memory. */

/* generate_training_samples_nerf, testbed_nerf.cu:691-849 (launch :3195).
 * rank/world_size: this rank marches the slice [n_rays*rank/world, n_rays*(rank+1)/world) of the
 * global ray range (8e); n_rays_ptr / max_samples_ptr (device, may be NULL) override the immediates. */
int ngp_k_generate_training_samples(
	void* stream, uint32_t n_rays, uint32_t rank, uint32_t world_size, const uint32_t* n_rays_ptr,
	ngp_aabb aabb, uint32_t max_samples, const uint32_t* max_samples_ptr, ngp_pcg32 rng,
	uint32_t* ray_counter, uint32_t* numsteps_counter, uint32_t* ray_indices_out, ngp_ray* rays_out,
	uint32_t* numsteps_out, float* coords_out, uint32_t n_training_images, const ngp_image_meta* metadata,
	const ngp_xform* xforms, const uint8_t* density_grid_bitfield, uint32_t max_mip,
	int snap_to_pixel_centers, float cone_angle_constant);

/* compute_loss_kernel_train_nerf, testbed_nerf.cu:852-1180 (launch :3241). */
int ngp_k_compute_loss(
	void* stream, uint32_t n_rays, const uint32_t* n_rays_ptr, ngp_aabb aabb, ngp_pcg32 rng,
	uint32_t max_samples_compacted, const uint32_t* rays_counter, float loss_scale,
	const float background_color[3], int color_space_srgb, int random_bg_color, int linear_colors,
	uint32_t n_training_images, const ngp_image_meta* metadata, const ngp_half* network_output,
	uint32_t output_stride, uint32_t* numsteps_counter_compacted, const uint32_t* ray_indices_in,
	const ngp_ray* rays_in, uint32_t* numsteps_inout, const float* coords_in, float* coords_out,
	ngp_half* dloss_doutput, uint32_t dloss_stride, int loss_type, float* loss_output,
	int rgb_activation, int density_activation, int snap_to_pixel_centers,
	const float* mean_density_ptr, float near_distance);
/* construct_cdf_2d + construct_cdf_1d + the image CDF (testbed_nerf.cu:1530-1580, 2795-2847): error map (n_images x height x width, device) ->
 * cdf_x_cond_y (same shape), cdf_y (n_images x height), cdf_img (n_images). */
int ngp_k_construct_error_cdfs(void* stream, uint32_t n_images, uint32_t width, uint32_t height, const float* error_map, float* cdf_x_cond_y, float* cdf_y, float* cdf_img);
/* fill_rollover_and_rescale<T> / fill_rollover<float>, launches testbed_nerf.cu:3298-3306. */
int ngp_k_fill_rollover(void* stream, uint32_t n_elements, const uint32_t* n_input_ptr,
                        float* coords_inout, uint32_t coord_stride, ngp_half* dloss_inout, uint32_t dloss_stride);

/* mark_untrained_density_grid, testbed_nerf.cu:87-162 */
int ngp_k_mark_untrained_density_grid(void* stream, uint32_t n_elements, float* grid,
	uint32_t n_training_images, const ngp_image_meta* metadata, const ngp_xform* xforms, int clear_visible_voxels);
/* generate_grid_samples_nerf_nonuniform, testbed_nerf.cu:216-257 */
int ngp_k_generate_grid_samples(void* stream, uint32_t n_elements, ngp_pcg32 rng, uint32_t step, ngp_aabb aabb,
	const float* grid_in, float* positions_out, uint32_t* indices_out, uint32_t n_cascades, float thresh);
/* splat_grid_samples_nerf_max_nearest_neighbor, testbed_nerf.cu:259-284 */
int ngp_k_splat_grid_samples(void* stream, uint32_t n_elements, const uint32_t* indices,
	const ngp_half* network_output, uint32_t out_stride, float* grid_out, int density_activation);
/* ema_grid_samples_nerf, testbed_nerf.cu:316-338 */
int ngp_k_ema_grid_samples(void* stream, uint32_t n_elements, float decay, float* grid_out, const float* grid_in);
/* update_density_grid_mean_and_bitfield, testbed_nerf.cu:2594-2633 (reduce_sum + grid_to_bitfield :348
 * + bitfield_max_pool :376). bitfield: 128^3/8 * 8 bytes; mean: 1 float. */
int ngp_k_update_mean_and_bitfield(void* stream, const float* grid, uint32_t max_cascade,
	uint8_t* bitfield, float* mean_out);
/* grid_to_bitfield + max-pool with a caller-supplied mean (bit-exact test hook). */
int ngp_k_grid_to_bitfield(void* stream, const float* grid, uint32_t max_cascade, uint8_t* bitfield,
	const float* mean_ptr);

/* ------------------------------------------------------------------ NeRF trainer --------- */

/* Owns Testbed::m_nerf training state: dataset replicas on device, density grid + bitfield,
 * counters, rng, scratch arena (train_nerf_step's 11 buffers, testbed_nerf.cu:3014-3040). */
int ngp_nerf_create(ngp_model* model, const ngp_nerf_options* opts_host, ngp_aabb aabb, ngp_nerf** out);
void ngp_nerf_destroy(ngp_nerf*);
/* Update the run-time members (python_api.cu:714-853: near_distance, random_bg_color, cone_angle_constant, colour space,
 * activations, loss ...). Batch size, max_cascade and rank/world_size are fixed at creation. */
int ngp_nerf_set_options(ngp_nerf*, const ngp_nerf_options* opts_host);
/* NerfDataset upload (nerf_loader.cu:749-850 set_training_image; metadata/xforms host arrays;
 * pixels_host[i] points at resolution.x*resolution.y pixels of the given type). */
int ngp_nerf_set_dataset_host(ngp_nerf*, uint32_t n_images, const ngp_image_meta* metadata_host,
                              const ngp_xform* xforms_host, const void* const* pixels_host);
/* Same, but pixels already live on the device (metadata[i].pixels are device pointers). */
int ngp_nerf_set_dataset_device(ngp_nerf*, uint32_t n_images, const ngp_image_meta* metadata_host,
                                const ngp_xform* xforms_host);
/* Testbed::train(batch_size) (testbed.cu:4561-4647): [training_prep_nerf every clamp(step/16,1,16)
 * steps] + train_nerf (testbed_nerf.cu:2704) = K1..K6, `n_steps` times, WITHOUT host synchronisation
 * (rays_per_batch adaptation, testbed_nerf.cu:2698-2699, runs on the device). */
int ngp_nerf_train(ngp_nerf*, void* stream, uint32_t n_steps);
/* Pieces of the above, exposed for tests / the multi-GPU driver:
 *   prep      = training_prep_nerf (testbed_nerf.cu:3385) if due
 *   forward_backward = train_nerf_step (testbed_nerf.cu:3007) -> gradients ready for all-reduce
 *   finish    = optimizer_step + counters.update_after_training (testbed_nerf.cu:2770-2778) */
int ngp_nerf_train_prep(ngp_nerf*, void* stream);
int ngp_nerf_train_forward_backward(ngp_nerf*, void* stream);
int ngp_nerf_train_finish(ngp_nerf*, void* stream);
/* Multi-rank order that lets the controller and the next step's ray marching start before the backward pass:
 * train_forward -> all-reduce(sum) of ngp_nerf_counter_ptrs' three words -> train_backward -> all-reduce(sum) of the gradients ->
 * train_finish.  (train_forward_backward + both all-reduces + train_finish stays valid.) */
int ngp_nerf_train_forward(ngp_nerf*, void* stream);
int ngp_nerf_train_backward(ngp_nerf*, void* stream);
/* Data-parallel training inside the library (new; SURVEY 8b "ngp_comm_init / ngp_allreduce_gradients", 8e): one process per GPU,
 * RCCL over xGMI (librccl is resolved at run time).  Rank 0 calls ngp_comm_unique_id (ncclGetUniqueId) and hands the 128 bytes to
 * every rank by any side channel; every rank calls ngp_comm_init with the rank / world_size its trainer was created with.  From
 * then on ngp_nerf_train runs  forward -> all-reduce(sum) of the three counter words -> backward -> all-reduce(sum) of the fp16
 * gradient vector on the caller's stream (the next step's ray marching runs beside it on its own stream) -> optimizer.
 * ngp_allreduce_gradients / ngp_allreduce_counters are the same collectives for callers that sequence the step themselves. */
int ngp_comm_unique_id(uint8_t id_out_host[128]);
int ngp_comm_init(ngp_nerf*, uint32_t rank, uint32_t world_size, const uint8_t id_host[128]);
int ngp_comm_destroy(ngp_nerf*);
int ngp_allreduce_gradients(ngp_nerf*, void* stream);
int ngp_allreduce_counters(ngp_nerf*, void* stream);
/* Sharded data-parallel step (round 5): from two ranks on, ngp_nerf_train under an ngp_comm_init communicator runs reduce-scatter(fp16 gradients, two
 * buckets of whole levels: the first one beside the second one's accumulation) -> Adam on THIS rank's 1 / world_size piece of each bucket (+ the replicated MLP, whose
 * 20 KB of gradients are all-reduced) -> all-gather of the new half parameters -> EMA / inference copy of the foreign pieces from the gathered parameters.
 * NGP_DP_ALLREDUCE=1 keeps the all-reduce + replicated-sweep step of rounds 2-4.  There is no counterpart in the reference (it has no multi-GPU training).
 *   ngp_nerf_dp_set_sharded / ngp_nerf_dp_layout / ngp_nerf_train_finish_sharded: for callers that run the collectives themselves (as with ngp_nerf_train_forward /
 *     _backward / ngp_nerf_counter_ptrs): set_sharded computes the layout (fails if a bucket does not divide into world_size pieces of whole entries); layout returns the
 *     two buckets' parameter ranges [begin, end) -- rank r owns [begin + r * (end - begin) / world, ... + (end - begin) / world) of each; after the backward pass the caller
 *     sums the MLP gradients [0, n_mlp) over all ranks and each rank's pieces over all ranks (reduce-scatter), calls finish_sharded(phase 0) = the local Adam step,
 *     all-gathers the half parameters of the pieces, and calls finish_sharded(phase 1) = foreign EMA + the rest of ngp_nerf_train_finish.
 *   ngp_nerf_dp_gather_state: COLLECTIVE -- the fp32 master parameters, Adam moments and step counters of every rank's own pieces to all ranks (only the half parameters
 *     and the inference copy are kept current everywhere); call on every rank before reading parameters / serialising a sharded run.  Until then
 *     ngp_nerf_dp_state_stale() is 1 and ngp_model_get_params_host / ngp_model_serialize_host FAIL (they would hand out stale numbers for the foreign pieces); leaving the
 *     sharded step (ngp_nerf_dp_set_sharded(t, 0), ngp_comm_destroy -- called on every rank) gathers first. */
int ngp_nerf_dp_set_sharded(ngp_nerf*, int on);
int ngp_nerf_dp_layout(ngp_nerf*, uint64_t begin[2], uint64_t end[2]);
int ngp_nerf_train_finish_sharded(ngp_nerf*, void* stream, int phase);
int ngp_nerf_dp_gather_state(ngp_nerf*, void* stream);
int ngp_nerf_dp_state_stale(const ngp_nerf*);
int ngp_nerf_dp_state_gathered(ngp_nerf*); /* callers that run their own collectives (ngp_nerf_train_finish_sharded): the optimizer state of every piece has been exchanged */
/* Three uint32 {measured_before_compaction, measured, this rank's loss sum in units of 2^-24} to all-reduce(sum) across ranks (8e):
 * every rank then derives the same next rays_per_batch and reports the loss of the union batch. */
int ngp_nerf_counter_ptrs(ngp_nerf*, uint32_t** counters3);
/* Error map and its CDFs (Testbed::Nerf::Training::error_map, testbed.h:745-756; built every n_steps_between_error_map_updates steps, x 1.5 per cycle,
 * testbed_nerf.cu:2753-2759, 2791-2855). Device pointers (null before the first cycle), resolutions {x, y}; any out pointer may be null. */
int ngp_nerf_error_map_ptrs(ngp_nerf*, float** error_map, int32_t error_map_res[2], float** cdf_x_cond_y, float** cdf_y, float** cdf_img, int32_t cdf_res[2],
	int* cdf_valid, uint32_t* n_steps_between_updates, uint32_t* n_steps_since_update);
/* n_steps_between_error_map_updates (testbed.h:813): 128 after a reset, multiplied by 1.5 after every CDF update; only between cycles. */
int ngp_nerf_set_error_map_interval(ngp_nerf*, uint32_t n_steps);
/* Test / tooling hook: install CDFs computed elsewhere (n_images x res[1] x res[0], n_images x res[1], n_images floats on the host). */
int ngp_nerf_set_error_cdfs_host(ngp_nerf*, const float* cdf_x_cond_y, const float* cdf_y, const float* cdf_img, const int32_t cdf_res[2]);
/* Blocking read-back (the reference's copy_to_host, testbed_nerf.cu:2681-2682). */
int ngp_nerf_get_stats(ngp_nerf*, void* stream, ngp_nerf_stats* out_host);
/* update_density_grid_nerf (testbed_nerf.cu:2476-2592) with explicit sample counts. */
int ngp_nerf_update_density_grid(ngp_nerf*, void* stream, float decay, uint32_t n_uniform, uint32_t n_nonuniform);
/* device pointers: density grid (float, 128^3*(max_cascade+1)), bitfield (128^3/8*8), mean (1 float) */
int ngp_nerf_density_grid_ptrs(ngp_nerf*, float** grid, uint8_t** bitfield, float** mean);
/* Diagnostic / test hook.  update_density_grid_nerf's samples (generate_grid_samples_nerf_nonuniform x 2, testbed_nerf.cu:2525-2557) depend on the grid rng, the EMA step and the grid
 * as the previous update left it -- on no parameter: the trainer draws and sorts the NEXT update's samples on a side stream while the steps in between train, and uses them if
 * nothing they were derived from has changed by then.  Returns how many updates found their samples ready.  (ngp_debug_set_flags2 bit 4 / NGP_DEBUG_FLAGS2_OR=4: inside the update, as before.) */
uint32_t ngp_nerf_grid_ahead_hits(ngp_nerf*);
int ngp_nerf_set_density_grid_host(ngp_nerf*, void* stream, const float* grid_host, uint64_t n);
/* m_training_step restored by load_snapshot (testbed.cu:5400-5403). */
int ngp_nerf_set_training_step(ngp_nerf*, uint32_t step);

/* Testbed::render_nerf (testbed_nerf.cu:1894-2149) semantics of the fused per-pixel kernel
 * fused_kernels/render_nerf.cuh:22-184: frame_buffer = premultiplied linear RGBA float4 per pixel,
 * depth_buffer float per pixel. camera = mat4x3 col-major (12 floats, host). */
typedef struct ngp_render_params {
	int32_t resolution[2];
	float focal_length[2];
	float screen_center[2];
	float camera[12];
	int32_t lens_mode; float lens_params[7];
	uint32_t spp_index;              /* sample_index */
	int32_t snap_to_pixel_centers;
	float min_transmittance;         /* testbed.h:890 0.01, eval 1e-4 */
	float near_distance;             /* m_render_near_distance (0 for eval) */
	int32_t use_inference_params;    /* 1: EMA weights (testbed_nerf.cu:1772) */
	ngp_aabb render_aabb;
} ngp_render_params;
int ngp_nerf_render(ngp_nerf*, void* stream, const ngp_render_params* params_host,
                    float* frame_buffer, float* depth_buffer);

/* Extra (latent / light-direction) dims of a model created with n_extra_dims > 0: one vector per training image, copied behind every
 * NerfCoordinate of the image's rays (testbed_nerf.cu:718-744, 833) and, when optimize_extra_dims is on, trained by one
 * VarAdamOptimizer per image (adam_optimizer.h:27-47; testbed_nerf.cu:2743-2750, 2860-2878, 3325-3340).
 * set: Testbed::Nerf::reset_extra_dims (testbed_nerf.cu:3656-3683) -- the host computes the initial values (warped light directions
 *      or uniform random latents) for EVERY image of the dataset (n_images >= the count handed to ngp_nerf_set_dataset_*, which is
 *      n_images_for_training: images that join the training set later find their initial values in place, getters and snapshots cover
 *      the whole dataset); this installs them, resets the optimizers and keeps a copy of image 0's vector as the default rendering dims.
 * get: Training::get_extra_dims_cpu (testbed_nerf.cu:1862-1877) for the first n_images images of the dataset.
 * rendering: set_rendering_extra_dims_from_training_view / set_rendering_extra_dims (testbed_nerf.cu:3685-3735): view >= 0 = that
 *      training view's current dims; view < 0 = `values` (n_extra_dims floats; NULL = the copy of image 0's INITIAL dims taken by
 *      ngp_nerf_set_extra_dims, the state after reset_extra_dims, :3679-3682).
 * light dir: Nerf::light_dir + NerfDataset::has_light_dirs -- with light directions in the dataset get_rendering_extra_dims overwrites the
 *      first three rendering dims with warp_direction(normalize(light_dir)) (:3697-3706; default light_dir (0.5, 0.5, 0.5)). */
int ngp_nerf_set_extra_dims(ngp_nerf*, const float* values_host, uint32_t n_images);
int ngp_nerf_set_light_dir(ngp_nerf*, int has_light_dirs, const float light_dir[3]);
/* Training::extra_dims_opt as snapshots carry it (testbed.cu:5311, 5482-5486; adam_optimizer.h to_json / from_json): the per-image VarAdamOptimizers' moments (n_images x
 * n_extra_dims floats each), every optimizer's iteration count (n_images values: an image that joined the training set later has stepped less often) and the learning rate
 * the last step used (set_learning_rate(m_optimizer->learning_rate()) before every step, testbed_nerf.cu:2874; 1e-4, the class's default, before the first).
 * set installs variables, moments and counts without resetting anything. */
int ngp_nerf_get_extra_dims_optimizer(ngp_nerf*, float* first_moment_host, float* second_moment_host, uint32_t* iter_host /* n_images */, uint32_t n_images);
int ngp_nerf_set_extra_dims_optimizer(ngp_nerf*, const float* variable_host, const float* first_moment_host, const float* second_moment_host, const uint32_t* iter_host /* n_images */, uint32_t n_images);
float ngp_nerf_extra_dims_learning_rate(const ngp_nerf*);
/* stand-alone launches of compute_extra_dims_gradient_train_nerf and of the VarAdamOptimizer step over device buffers (test hooks) */
int ngp_k_extra_dims_gradient(void* stream, uint32_t n_rays_total, uint32_t rays_counter, float* grad_out, uint32_t n_extra, uint32_t n_images,
                              const uint32_t* ray_indices, const uint32_t* numsteps, const float* dextra, uint32_t max_rows);
int ngp_k_extra_dims_adam(void* stream, uint32_t n, float* variable, const float* gradient_scaled, float* m, float* v, uint32_t iter, float lr, float loss_scale);
int ngp_nerf_get_extra_dims(ngp_nerf*, float* values_host, uint32_t n_images);
int ngp_nerf_get_extra_dims_gradient(ngp_nerf*, float* values_host, uint32_t n_images); /* test hook: extra_dims_gradient_gpu of the last step (loss-scaled) */
int ngp_nerf_set_optimize_extra_dims(ngp_nerf*, int on);                                 /* m_nerf.training.optimize_extra_dims (testbed.h) */
int ngp_nerf_set_rendering_extra_dims(ngp_nerf*, int training_view, const float* values_host);

/* CudaRenderBuffer::accumulate (running mean over spp, render_buffer.cu:228-260) and tonemap (Identity curve: 2^exposure, background behind
 * the premultiplied colour, optional linear -> sRGB; :511-560) on device buffers of RGBA float32 */
int ngp_render_accumulate(void* stream, const float* frame, float* accum, uint64_t n_floats, uint32_t sample_index);
int ngp_render_tonemap(void* stream, float* rgba, uint64_t n_pixels, float exposure, const float background_linear[4], int to_srgb);
/* the same with a tonemapping curve (ETonemapCurve, render_buffer.cu:264-321): 0 Identity, 1 ACES, 2 Hable, 3 Reinhard; and its per-pixel
 * arithmetic evaluated on the host (test hook, no GPU needed) */
int ngp_render_tonemap_curve(void* stream, float* rgba, uint64_t n_pixels, float exposure, const float background_linear[4], int to_srgb, int curve);
int ngp_host_tonemap_pixel(const float rgba[4], float exposure, const float background_linear[4], int to_srgb, int curve, float out[4]);

/* ------------------------------------------------------------------ profiling ------------ */
/* Optional per-kernel timing with HIP events recorded on the launch stream (bench.py roofline leg).
 * enable(1) clears the accumulators; read() synchronises the pending events and returns, per kernel
 * class i < ngp_profile_count(), the summed milliseconds and the number of timed launches. */
int ngp_profile_enable(int on);
int ngp_profile_count(void);
const char* ngp_profile_name(int i);
int ngp_profile_read(double* ms_sum_host, uint64_t* launches_host);
/* 1 if the last training step's backward pass read its encodings from the forward pass's stash (base.json's shape, production kernels) instead of gathering them again:
 * decides which byte model bench.py charges the scatter unit (no reference counterpart: tcnn always re-gathers). */
int ngp_nerf_uses_k2_stash(ngp_nerf*);
/* Number of parameters the last ngp_model_optimizer_step swept in k_optimizer itself: all of them, or -- when the step's k_grad_accumulate applied the optimizer to the hashed
 * levels in its epilogue (single-GPU ngp_nerf_train) -- the MLP and the dense levels only.  For bench.py's byte model of the two kernels (no reference counterpart). */
uint64_t ngp_model_last_sweep_params(const ngp_model*);

/* ------------------------------------------------------------------ test hooks ----------- */
/* Not part of the reference's API surface: expose intermediate state to the parity tests. */
int ngp_model_encode(ngp_model*, void* stream, const float* pos, uint32_t pos_stride, uint32_t n, ngp_half* out32);
int ngp_nerf_scratch_ptrs(ngp_nerf*, uint32_t** ray_indices, ngp_ray** rays, uint32_t** numsteps, float** coords,
                          ngp_half** mlp_out, float** coords_compacted, ngp_half** dloss, void** counters);
int ngp_nerf_set_rays_per_batch(ngp_nerf*, uint32_t rays_per_batch);
int ngp_nerf_get_rng(ngp_nerf*, ngp_pcg32* rng, ngp_pcg32* density_grid_rng);
int ngp_nerf_set_rng(ngp_nerf*, const ngp_pcg32* rng);
/* lazy (front-to-back) K2: rounds = 1 (default): one launch, every wavefront follows its ray tile by tile; rounds = 2..8: list-driven rounds;
   samples per tile 16 (default: two rays per wavefront) or 32 (csrc/model_kernels.hip k_inference_tiles) */
int ngp_nerf_set_k2_params(ngp_nerf*, uint32_t rounds, uint32_t tile_w);
/* host-side evaluation of the device's camera model (the same source, csrc/ngp_device.hpp uv_to_ray / pos_to_uv, compiled for the host):
   all seven lens modes of common_device.cuh:413-577.  Test hooks: no GPU needed.  uv_to_ray returns 0 where the lens has no ray. */
int ngp_host_uv_to_ray(const ngp_image_meta* meta, const float xform12[12], const float uv[2], float origin_out[3], float dir_out[3]);
int ngp_host_pos_to_uv(const ngp_image_meta* meta, const float xform12[12], const float pos[3], float uv_out[2]);
/* get_xform_given_rolling_shutter (common_device.cuh:670-674) of csrc/ngp_device.hpp evaluated on the host (test hook): the training camera of a pixel
 * (uv) of a frame with rolling shutter {a, b, c, d} -> t = a + b u + c v + d motionblur_time, interpolated between xform.start and xform.end */
int ngp_host_xform_given_rolling_shutter(const ngp_xform* xform, const float rolling_shutter[4], const float uv[2], float motionblur_time, float xform12_out[12]);
/* Test hook, no GPU: ngp_sdf_create's mesh setup (TriangleBvh::build-style median splits that reorder the triangles, triangle_bvh.cu:757-840; DiscreteDistribution over the
 * surface areas, discrete_distribution.h:21-38) and the Raystab signed distance of csrc/sdf_kernels.hip (triangle_bvh.cu:631-650, 893-909) evaluated on the host from the same
 * source.  distances_inout: upper bounds in (when use_upper_bounds), signed distances out.  triangles_ordered_out (9 floats each) / cdf_out: optional. */
int ngp_host_sdf_signed_distance(const float* triangles_host, uint32_t n_triangles, const float* positions_host, uint32_t n, float* distances_inout, int use_upper_bounds,
                                 float* triangles_ordered_out, float* cdf_out);
/* ablation switches of csrc/ngp_kernels.hpp (0 = production path); process-wide */
int ngp_debug_set_flags(uint32_t flags);
/* The switches in effect (NGP_DEBUG_FLAGS_OR from the environment included); 0 = the production path. */
uint32_t ngp_debug_get_flags(void);
/* second word of ablation switches (DBG2_* of csrc/ngp_kernels.hpp; NGP_DEBUG_FLAGS2_OR): 1 = the backward pass as T1 + W, two kernels (round 4), instead of k_train_fused; 4 = the occupancy-grid update draws its samples itself */
int ngp_debug_set_flags2(uint32_t flags);
uint32_t ngp_debug_get_flags2(void);
/* The same switches PER HANDLE: a handle with an override (on != 0) runs its calls (training step, inference, rendering, grid update, optimizer) under `flags` / `flags2`
 * whatever the process-wide words say; on = 0 removes it.  ngp_nerf_set_debug_flags covers the trainer and its model.  Thread-compatible like every call on a handle:
 * calls on handles with different overrides must not overlap in time on different threads. */
int ngp_model_set_debug_flags(ngp_model*, int on, uint32_t flags, uint32_t flags2);
int ngp_nerf_set_debug_flags(ngp_nerf*, int on, uint32_t flags, uint32_t flags2);
/* train mode of the STAND-ALONE ngp_k_compute_loss (the trainer takes it from ngp_nerf_options) */
int ngp_debug_set_train_mode(int mode);
/* depth supervision of the STAND-ALONE ngp_k_compute_loss (the trainer takes it from ngp_nerf_options; testbed_nerf.cu:1027-1029, 1126-1129) */
int ngp_debug_set_depth_supervision(float depth_supervision_lambda, int depth_loss_type);
/* Stand-alone ngp_k_generate_training_samples / ngp_k_compute_loss only: the per-ray target records the trainer's ray set-up kernel computes for K3 (what
 * compute_loss_kernel_train_nerf derives per ray at testbed_nerf.cu:930-1027: target colour, background, target depth).  K1 writes them to `buf` (device, 8 floats per
 * ray slot) with the given colour options, K3 reads them instead of deriving them.  plain_dataset != 0: the caller vouches for 8-bit images, Perspective / OpenCV lenses
 * and still cameras (the ray set-up kernel's small instance).  buf = null: off. */
int ngp_debug_set_ray_targets(float* buf, int plain_dataset, const float* background_color3, int color_space_srgb, int random_bg_color, int linear_colors);
/* Stand-alone ngp_k_generate_training_samples / ngp_k_compute_loss only: CDFs (device; null = uniform) and the error map K3 splats into (device; null = none). */
int ngp_debug_set_error_sampling(const float* cdf_x_cond_y, const float* cdf_y, const float* cdf_img, const int32_t cdf_res[2], float* error_map, const int32_t error_map_res[2]);
/* Stand-alone ngp_k_generate_training_samples / ngp_k_compute_loss only: the per-image extra dims (extra_dims_gpu, testbed_nerf.cu:718-719, 744, 833; device, n_images x n_extra
 * floats) that K1 copies behind every NerfCoordinate; both kernels' coords rows then are 7 + n_extra floats (PitchedPtr stride, testbed_nerf.cu:3010-3011).  n_extra = 0: off. */
int ngp_debug_set_extra_dims(const float* extra_dims_device, uint32_t n_extra);
/* layout of the hashed levels' binned gradient scatter (csrc/model_kernels.hip k_grad_bin / k_grad_accumulate): table entries per
 * chunk = 2^chunk_log2 (11 or 12), one block per chunk (split = 0) or per (chunk, feature pair) (split = 1), list capacity override
 * in records (0 = twice the mean; a small value forces the list-overflow path for the tests); process-wide */
int ngp_debug_set_bin_params(uint32_t chunk_log2, uint32_t split, uint32_t cap_override);

#ifdef __cplusplus
}
#endif
#endif /* NGP_HIP_H */
