// testbed.hpp -- C++ host of the MI355X NeRF path with the member / method names of ngp::Testbed
// (reference include/neural-graphics-primitives/testbed.h:71-1292) that scripts/run.py and pyngp users touch
// (python_api.cu:439-853).  All device work goes through the C-ABI of libngp_hip.so (include/ngp_hip.h).
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <limits>
#include <functional>
#include <string>
#include <vector>

extern "C" {
#include "../../include/ngp_hip.h"
}
#include "../csrc/mini_json.hpp"
#include "camera_path_lite.hpp"

namespace ngp_host {

enum class ETestbedMode : int { Nerf = 0, Sdf = 1, Image = 2, Volume = 3, None = 4 };   // common.h
enum class ETrainMode : int { Nerf = 0, Rfl = 1, RflRelax = 2 };                          // common.h:47-51
enum class EColorSpace : int { Linear = 0, SRGB = 1, VisPosNeg = 2 };
enum class ETonemapCurve : int { Identity = 0, ACES = 1, Hable = 2, Reinhard = 3 };
enum class ELossType : int { L2 = 0, L1 = 1, Mape = 2, Smape = 3, Huber = 4, LogL1 = 5, RelativeL2 = 6 }; // common.h:99-107 = NGP_LOSS_*

enum class ENerfActivation : int { None = 0, ReLU = 1, Logistic = 2, Exponential = 3 };                  // common.h = NGP_ACT_*

struct BoundingBox {                        // bounding_box.cuh:43-252 as seen from Python (python_api.cu:409-427)
	std::array<float, 3> min{std::numeric_limits<float>::infinity(), std::numeric_limits<float>::infinity(), std::numeric_limits<float>::infinity()};
	std::array<float, 3> max{-std::numeric_limits<float>::infinity(), -std::numeric_limits<float>::infinity(), -std::numeric_limits<float>::infinity()};
	BoundingBox() {}
	BoundingBox(const std::array<float, 3>& a, const std::array<float, 3>& b) : min(a), max(b) {}
	bool is_empty() const { return max[0] < min[0] || max[1] < min[1] || max[2] < min[2]; }
	std::array<float, 3> diag() const { return {max[0] - min[0], max[1] - min[1], max[2] - min[2]}; }
	std::array<float, 3> center() const { return {0.5f * (max[0] + min[0]), 0.5f * (max[1] + min[1]), 0.5f * (max[2] + min[2])}; }
	std::array<float, 3> relative_pos(const std::array<float, 3>& p) const { const auto d = diag(); return {(p[0] - min[0]) / d[0], (p[1] - min[1]) / d[1], (p[2] - min[2]) / d[2]}; }
	bool contains(const std::array<float, 3>& p) const { return p[0] >= min[0] && p[0] <= max[0] && p[1] >= min[1] && p[1] <= max[1] && p[2] >= min[2] && p[2] <= max[2]; }
	void enlarge(const std::array<float, 3>& p) { for (int k = 0; k < 3; ++k) { min[k] = std::min(min[k], p[k]); max[k] = std::max(max[k], p[k]); } }
	void enlarge(const BoundingBox& o) { for (int k = 0; k < 3; ++k) { min[k] = std::min(min[k], o.min[k]); max[k] = std::max(max[k], o.max[k]); } }
	void inflate(float amount) { for (int k = 0; k < 3; ++k) { min[k] -= amount; max[k] += amount; } }
	BoundingBox intersection(const BoundingBox& o) const { BoundingBox r = *this; for (int k = 0; k < 3; ++k) { r.min[k] = std::max(r.min[k], o.min[k]); r.max[k] = std::min(r.max[k], o.max[k]); } return r; }
	bool intersects(const BoundingBox& o) const { return !intersection(o).is_empty(); }
	float distance_sq(const std::array<float, 3>& p) const { float s = 0; for (int k = 0; k < 3; ++k) { const float d = std::max(std::max(min[k] - p[k], p[k] - max[k]), 0.0f); s += d * d; } return s; }
	float distance(const std::array<float, 3>& p) const { return std::sqrt(distance_sq(p)); }
	float signed_distance(const std::array<float, 3>& p) const { // bounding_box.cuh:232-235
		float q[3], outside = 0.f, inside = -std::numeric_limits<float>::infinity();
		for (int k = 0; k < 3; ++k) { q[k] = std::fabs(p[k] - 0.5f * (min[k] + max[k])) - 0.5f * (max[k] - min[k]); const float o = std::max(q[k], 0.0f); outside += o * o; inside = std::max(inside, q[k]); }
		return std::sqrt(outside) + std::min(inside, 0.0f);
	}
	std::array<float, 2> ray_intersect(const std::array<float, 3>& pos, const std::array<float, 3>& dir) const; // bounding_box.cuh:163-210 (testbed.cpp)
	std::vector<std::array<float, 3>> get_vertices() const { std::vector<std::array<float, 3>> v(8); for (int i = 0; i < 8; ++i) v[i] = {(i & 1) ? max[0] : min[0], (i & 2) ? max[1] : min[1], (i & 4) ? max[2] : min[2]}; return v; } // :237-246
};

struct Lens { int mode = NGP_LENS_PERSPECTIVE; std::array<float, 7> params{}; };   // common.h Lens {ELensMode mode; float params[7]} as seen from Python (python_api.cu:429-433)
enum class ELensMode : int { Perspective = 0, OpenCV = 1, FTheta = 2, LatLong = 3, OpenCVFisheye = 4, Equirectangular = 5, Orthographic = 6 }; // = NGP_LENS_*

struct ImageMetadata {                      // TrainingImageMetadata as seen from Python (python_api.cu:766-779)
	std::array<int, 2> resolution{0, 0};
	std::array<float, 2> focal_length{1000.f, 1000.f};
	std::array<float, 2> principal_point{0.5f, 0.5f};
	int lens_mode = NGP_LENS_PERSPECTIVE;
	std::array<float, 7> lens_params{};
	std::array<float, 4> rolling_shutter{};          // {a, b, c, d}: pixel time t = a + b u + c v + d motionblur_time (nerf_loader.cu:204-215)
	std::array<float, 3> light_dir{0.f, 0.f, 0.f};    // json "driver_parameters" LightX / LightY / LightZ, normalised, in ngp's axes (nerf_loader.cu:671-680)
};

struct NerfDataset {                        // nerf_loader.h NerfDataset (subset)
	size_t n_images = 0;
	std::vector<ImageMetadata> metadata;
	std::vector<std::array<float, 12>> xforms;      // ngp convention, column-major mat4x3 (TrainingXForm::start)
	std::vector<std::array<float, 12>> xforms_end;  // TrainingXForm::end: json "transform_matrix_end" (== start without motion data)
	std::vector<std::vector<uint8_t>> pixels;        // RGBA8 per image (host copy; also feeds render_ground_truth)
	std::vector<std::vector<float>> depth;            // per image: empty, or one float per pixel in scene units (json "depth_path" + "integer_depth_scale", nerf_loader.cu:629-641, 73-82)
	std::vector<std::vector<float>> pixels_float;     // images handed over by Training.set_image (python_api.cu:45-72): linear premultiplied RGBA float32, sampled as EImageDataType::Float
	std::vector<std::vector<uint16_t>> pixels_half;  // sharpened images (nerf.sharpen > 0): linear premultiplied RGBA halfs, what the trainer then samples
	float sharpen_amount = 0.f;                      // what the images were loaded with (json "sharpen" overrides nerf.sharpen, nerf_loader.cu:462)
	std::vector<std::string> paths;
	int aabb_scale = 1;
	float scale = 0.33f;                             // NERF_SCALE, nerf_loader.h:29
	std::array<float, 3> offset{0.5f, 0.5f, 0.5f};   // nerf_loader.cu:403-404
	bool is_hdr = false;
	bool from_mitsuba = false;                       // json "from_mitsuba" / "normal_mts_args": Mitsuba axis convention (nerf_loader.h:101-120)
	BoundingBox render_aabb;                         // json "render_aabb" (nerf_loader.cu:457-460); empty = the whole scene box
	std::array<float, 9> render_aabb_to_local{1, 0, 0, 0, 1, 0, 0, 0, 1}; // nerf_loader.h: identity unless a snapshot carries a crop box orientation (column-major mat3)
	std::array<float, 3> up{0.f, 1.f, 0.f};          // json "up", axes permuted like the transforms (nerf_loader.cu:528-533)
	std::array<int, 2> envmap_resolution{0, 0};      // no environment maps in this build
	uint32_t n_extra_learnable_dims = 0;             // json "n_extra_learnable_dims" (nerf_loader.cu:482-483): per-image latent codes
	bool has_light_dirs = false;                     // a frame carried "driver_parameters" (nerf_loader.cu:671-680): three fixed extra dims = the warped light direction
	uint32_t n_extra_dims() const { return (has_light_dirs ? 3u : 0u) + n_extra_learnable_dims; } // nerf_loader.h:85-87
	std::array<float, 12> nerf_matrix_to_ngp(const std::array<float, 12>& row_major_3x4) const; // nerf_loader.h:101-120
	std::array<float, 12> ngp_matrix_to_nerf(const std::array<float, 12>& col_major_4x3) const; // nerf_loader.h:122-139 (row-major 3x4 out)
};

class Testbed;
struct NerfTraining {
	float near_distance = 0.1f;                      // testbed.h:817
	ETrainMode train_mode = ETrainMode::RflRelax;    // testbed.h:822 (forced to Nerf without JIT, testbed_nerf.cu:3091-3094)
	bool random_bg_color = true;                     // testbed.h:793
	bool linear_colors = false;                      // testbed.h:794
	bool snap_to_pixel_centers = true;               // testbed.h:797
	float density_grid_decay = 0.95f;                // testbed.h:818
	float depth_supervision_lambda = 0.f;            // testbed.h:824
	int depth_loss_type = NGP_LOSS_L1;               // testbed.h:796 (ELossType)
	bool sample_focal_plane_proportional_to_error = false; // testbed.h:810
	bool sample_image_proportional_to_error = false;       // testbed.h:811
	bool accumulate_error_map = false;                     // (the reference always accumulates, testbed_nerf.cu:2793; here: while one of the two switches is on, or when this is set)
	int n_images_for_training = 0;                         // testbed.h:771: rays are drawn from the first n images (set to n_images by load_nerf_post, testbed_nerf.cu:2371)
	int loss_type = -1;                                    // ELossType; -1 = as the network config's "loss.otype" says (testbed.cu:4209 writes it into this member)
	int view = 0;                                          // testbed.h:770: the training view the camera was last set to
	bool optimize_extra_dims = false;                      // testbed.h: set when the dataset has learnable dims (testbed_nerf.cu:2379); python_api.cu:789-790
	Testbed* owner = nullptr;                        // the reference's Training methods (set_camera_extrinsics ...) live on this object: route them to the Testbed
	NerfDataset dataset;
};

struct Nerf {
	float sharpen = 0.f;
	float cone_angle_constant = 1.f / 256.f;
	float render_min_transmittance = 0.01f;          // testbed.h:890
	int rgb_activation = -1, density_activation = NGP_ACT_EXPONENTIAL; // ENerfActivation; rgb -1 = Exponential for HDR data, Logistic otherwise (testbed_nerf.cu:2354)
	bool visualize_cameras = false;                  // GUI overlay switch: kept so that scripts that clear it run
	int max_cascade = 0;
	int rendering_extra_dims_from_training_view = -1;   // testbed.h; python_api.cu:725-727
	std::vector<float> rendering_extra_dims;         // set_rendering_extra_dims (python_api.cu:739); empty = the copy of image 0's initial dims reset_extra_dims took
	std::vector<float> rendering_extra_dims_default; // reset_extra_dims' copy of image 0's initial dims (testbed_nerf.cu:3679-3682), filled when the trainer is created
	std::array<float, 3> light_dir{0.5f, 0.5f, 0.5f};   // testbed.h:871: with light directions in the dataset the rendering's first three extra dims = warp_direction(normalize(light_dir)) (testbed_nerf.cu:3697-3706; GUI-set in the reference)
	NerfTraining training;
};

enum class ERandomMode : int { Random = 0, Halton = 1, Sobol = 2, Stratified = 3 };   // common.h
enum class EMeshSdfMode : int { Watertight = 0, Raystab = 1, PathEscape = 2 };         // common.h:118-122
struct ImageTraining { bool snap_to_pixel_centers = true; bool linear_colors = false; };  // testbed.h:966-967
struct ImagePrimitive { ImageTraining training; ERandomMode random_mode = ERandomMode::Stratified; }; // testbed.h:970
struct SdfTraining { bool generate_sdf_data_online = true; float surface_offset_scale = 1.0f; };       // testbed.h:937-938
struct SdfPrimitive {                                                                    // testbed.h:905-950 (what the trainer and the ground truth use)
	SdfTraining training;
	EMeshSdfMode mesh_sdf_mode = EMeshSdfMode::Raystab;
	float mesh_scale = 1.f;          // set by load_mesh (testbed_sdf.cu:1404)
	float zero_offset = 0.f;
	bool use_triangle_octree = false;
	bool calculate_iou_online = false;
};

using ImageDecoder = std::function<bool(const std::string& path, int& w, int& h, std::vector<uint8_t>& rgba)>;

class Testbed {
public:
	Testbed();
	~Testbed();
	Testbed(const Testbed&) = delete;

	// ---- python_api.cu:439-711 ----
	void load_file(const std::string& path);                        // testbed.cu:183-252
	void load_training_data(const std::string& path);               // testbed.cu:156
	void reload_network_from_file(const std::string& path = "");    // testbed.cu:311
	void reload_network_from_json_text(const std::string& json_text, const std::string& config_base_path = ""); // testbed.cu:346-351
	const mini_json::Value& network_config() const { return m_network_config; }
	void reset_network();                                           // testbed.cu:4160
	void load_snapshot(const std::string& path);                    // testbed.cu:5357
	void dp_gather_state();
	void save_snapshot(const std::string& path, bool include_optimizer_state = false); // testbed.cu:5288
	bool frame();                                                   // testbed.cu:3908 (headless: train one step)
	void train(uint32_t batch_size);                                // testbed.cu:4561
	bool want_repl() const { return false; }
	void set_camera_to_training_view(int i);                        // testbed.cu:486
	void first_training_view(); void last_training_view(); void previous_training_view(); void next_training_view(); // testbed.cu:460-484
	void reset_camera();                                            // testbed.cu:507-528
	void load_camera_path(const std::string& path);                 // testbed.cu / camera_path.cu:135-168
	void set_camera_from_time(float t);                             // testbed.cu:4061-4075: camera, scale and fov of the path at playtime t in [0, 1]
	// render_to_cpu with a path animation (python_api.cu:145-236): every sample of the frame is taken at its own time inside the shutter interval
	std::vector<float> render_path_frame(int width, int height, int spp, bool linear, float start_t, float end_t, float shutter_fraction, std::vector<float>* depth_out = nullptr);
	CameraPath camera_path;
	void clear_training_data();                                     // testbed.cu:190-193
	void create_empty_nerf_dataset(size_t n_images, int aabb_scale = 1, bool is_hdr = false); // testbed_nerf.cu:2344-2351, nerf_loader.cu:148-172
	void set_training_image(int frame_idx, int w, int h, const float* rgba, const float* depth_or_null, float depth_scale); // NerfDataset::set_training_image for float data (nerf_loader.cu:749-850)
	size_t n_params(); size_t n_encoding_params();                  // testbed.cu:4089-4091
	// camera (testbed.cu:440-458): column-major mat4x3, columns = side, up, direction, position
	std::array<float, 3> view_pos() const { return {m_camera[9], m_camera[10], m_camera[11]}; }
	std::array<float, 3> view_dir() const { return {m_camera[6], m_camera[7], m_camera[8]}; }
	std::array<float, 3> look_at() const;
	void set_look_at(const std::array<float, 3>& pos);
	void set_view_dir(const std::array<float, 3>& dir);
	float scale() const { return m_scale; }
	void set_scale(float scale);
	std::array<float, 2> fov_xy() const; void set_fov_xy(const std::array<float, 2>& degrees); // testbed.cu:4085-4087
	std::array<float, 12> camera_matrix_row_major() const;          // m_camera as the 3 x 4 numpy array pyngp exposes
	void set_camera_matrix_row_major(const std::array<float, 12>& m);
	int find_closest_training_view(const std::array<float, 12>& pose_row_major_3x4) const; // testbed_nerf.cu:3710-3723
	std::vector<float> get_extra_dims(int trainview); // Training::get_extra_dims_cpu, testbed_nerf.cu:1862-1877
	// training cameras (testbed_nerf.cu:2151-2292)
	void set_camera_intrinsics(int frame_idx, float fx, float fy, float cx, float cy, float k1, float k2, float p1, float p2, float k3, float k4, bool is_fisheye);
	void set_camera_extrinsics(int frame_idx, const std::array<float, 12>& camera_to_world_row_major, bool convert_to_ngp);
	std::array<float, 12> get_camera_extrinsics(int frame_idx) const;
	void training_options_changed();                                // a member of nerf / nerf.training was written from Python: push it to the trainer before the next step
	void set_nerf_camera_matrix(const std::array<float, 12>& m_row_major_3x4);
	// render_to_cpu (python_api.cu:145-236): premultiplied RGBA float [h][w][4]
	std::vector<float> render(int width, int height, int spp, bool linear, std::vector<float>* depth_out = nullptr); // depth_out: render_with_depth (python_api.cu:520-532)
	ngp_nerf_stats stats();
	// image / SDF primitives (testbed_image.cu, testbed_sdf.cu): same entry points (load_training_data, train / frame, loss), plus
	float compute_image_mse(bool quantize_to_byte = false);                    // testbed_image.cu:490
	double calculate_iou(uint32_t n_samples = 128u * 128u * 128u * 4u, float scale_existing_results_factor = 0.f, bool blocking = true, bool force_use_octree = false); // testbed_sdf.cu:1636
	// data-parallel training (new): one process per GPU, ngp_comm_* of libngp_hip (RCCL); call before the first train()
	static std::string comm_unique_id();
	void comm_init(uint32_t rank, uint32_t world_size, const std::string& unique_id_128_bytes);

	// public members, same names as the reference
	std::string root_dir;
	ETestbedMode mode = ETestbedMode::None;
	bool shall_train = true;
	uint32_t training_step = 0;
	float loss = 0.f;
	float exposure = 0.f;
	std::array<float, 4> background_color{0.f, 0.f, 0.f, 1.f};          // testbed.h:1031
	bool snap_to_pixel_centers = false;
	bool render_with_lens_distortion = false;
	bool render_ground_truth = false;
	EColorSpace color_space = EColorSpace::Linear;                   // testbed.h:1002
	ETonemapCurve tonemap_curve = ETonemapCurve::Identity;
	int fov_axis = 1;
	uint32_t training_batch_size = 1u << 18;                         // testbed.h:1089
	uint64_t seed = 1337;                                            // testbed.h:680
	Nerf nerf;
	ImagePrimitive image;                                            // python_api.cu:673, 874-879: settings read when the image trainer is created
	SdfPrimitive sdf;                                                // python_api.cu:672, 855-872
	BoundingBox aabb, raw_aabb, render_aabb;                         // testbed.h:1025-1027; set by load_nerf_post (testbed_nerf.cu:2424-2431)
	std::array<float, 9> render_aabb_to_local{1, 0, 0, 0, 1, 0, 0, 0, 1};
	std::array<float, 3> up_dir{0.f, 1.f, 0.f};                      // testbed.h:672
	bool visualize_unit_cube = false;                                // GUI overlay switch: kept so that scripts that set it run
	float zoom = 1.f;                                                // testbed.h:647 (2-D zoom of the GUI view; kept for scripts)
	Lens render_lens() const { Lens l; l.mode = m_render_lens_mode; l.params = m_render_lens_params; return l; }              // testbed.h m_render_lens
	void set_render_lens(const Lens& l) { m_render_lens_mode = l.mode; m_render_lens_params = l.params; }
	float render_near_distance = 0.f;                                // testbed.h (m_render_near_distance)
	std::array<float, 2> relative_focal_length{1.f, 1.f};            // testbed.h: focal length / resolution[fov_axis]
	std::array<float, 2> screen_center{0.5f, 0.5f};
	float fov() const;
	void set_fov(float degrees);

	static ImageDecoder s_fallback_decoder;
	// the loader's built-in readers (load_stbi, nerf_loader.cu:570-603, 633): PNG + JPEG (baseline, progressive) -> RGBA8, 16-bit PNG -> one channel; test hooks for pyngp
	static bool read_image_builtin(const std::string& path, int& w, int& h, std::vector<uint8_t>& rgba);
	static bool read_depth_png16(const std::string& path, int& w, int& h, std::vector<uint16_t>& gray);
	static std::vector<uint16_t> sharpen_rgba8_for_tests(const std::vector<uint8_t>& rgba, int w, int h, float amount, bool has_mask); // the loader's sharpening (nerf_loader.cu:805-827), for tests
	static bool natural_path_less(const std::string& a, const std::string& b); // the loader's frame order (nerf_loader.cu:347-349: SI::natural::compare)
	static std::string s_default_root_dir;                           // directory that holds configs/ (set by the binding layer)                          // non-PNG images (jpg/exr): provided by the binding layer

	// snapshot["nerf"]["dataset"] (json_binding.h to_json / from_json of NerfDataset) as JSON text: test hooks, host only
	std::string nerf_dataset_to_json_text() const;
	void nerf_dataset_from_json_text(const std::string& text);

private:
	mini_json::Value dataset_to_json() const;
	static NerfDataset dataset_from_json(const mini_json::Value& jd, int default_aabb_scale);
	void ensure_trainer();
	void load_nerf_post();
	void destroy_trainer();
	void push_options();
	std::vector<float> initial_extra_dims(ngp_pcg32& rng) const; // reset_extra_dims' values for every image of the dataset, drawn from (and advancing) `rng` (testbed_nerf.cu:3656-3683)
	bool m_extra_dims_installed = false;            // the CURRENT trainer holds reset_extra_dims' values (cleared whenever the trainer is destroyed: a pointer comparison would be an ABA test, the allocator reuses addresses)
	ngp_nerf_options current_options() const;
	ngp_aabb scene_aabb() const;

	mini_json::Value m_network_config;
	std::string m_network_config_path;
	ngp_model* m_model = nullptr;
	ngp_nerf* m_nerf = nullptr;
	// image / SDF modes: the NetworkWithInputEncoding model + its trainer state
	ngp_encmlp* m_encmlp = nullptr; ngp_image* m_image = nullptr; ngp_sdf* m_sdf = nullptr;
	std::vector<float> m_image_pixels; int m_image_w = 0, m_image_h = 0; // RGBA float32, linear
public:
	const std::vector<float>& image_pixels() const { return m_image_pixels; } int image_width() const { return m_image_w; } int image_height() const { return m_image_h; } // image mode: what load_image read (tests)
private:
	std::vector<float> m_mesh; ngp_aabb m_mesh_aabb{};                   // 9 floats per triangle, normalised into the unit cube (load_mesh)
	void ensure_encmlp_trainer();
	bool m_dataset_dirty = true;
	int m_uploaded_n_images_for_training = -1;
	// camera (testbed.h:453-456): column-major mat4x3
	std::array<float, 12> m_camera{1, 0, 0, 0, 1, 0, 0, 0, 1, 0.5f, 0.5f, 0.5f};
	float m_scale = 1.f;                                                 // testbed.h:643
	int m_render_lens_mode = NGP_LENS_PERSPECTIVE;
	std::array<float, 7> m_render_lens_params{};
	int m_training_view = 0;
	float* m_frame_dev = nullptr; size_t m_frame_dev_floats = 0;
	bool m_warned_train_mode = false;
	uint32_t m_rank = 0, m_world_size = 1; std::string m_comm_id; bool m_comm_up = false;
};

std::string msgpack_repack(const std::string& data, bool input_compressed, bool output_compressed); // msgpack_lite round trip (tests)

} // namespace ngp_host
