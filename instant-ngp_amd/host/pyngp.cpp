// pyngp.cpp -- pybind11 module `pyngp`: the drop-in boundary of scripts/run.py (reference src/python_api.cu:306-968).
// Property / method / enum names follow the reference binding so that `import pyngp as ngp` scripts keep working for the
// NeRF path; members that belong to out-of-scope subsystems (GUI, VR, DLSS, SDF, image, volume, mesh export) raise.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>
#include <pybind11/eval.h>

#include "testbed.hpp"
#include "exr_lite.hpp"
#include "mesh_lite.hpp"

namespace py = pybind11;
using namespace ngp_host;

static py::list dataset_transforms(const NerfDataset& d) {
	py::list out;
	for (size_t i = 0; i < d.n_images; ++i) {
		py::array_t<float> a({3, 4}), b({3, 4});
		const auto& e = i < d.xforms_end.size() ? d.xforms_end[i] : d.xforms[i];
		for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) { a.mutable_at(r, c) = d.xforms[i][c * 3 + r]; b.mutable_at(r, c) = e[c * 3 + r]; }
		out.append(py::make_tuple(a, b));
	}
	return out;
}

static py::array_t<float> render_to_numpy(Testbed& t, int w, int h, int spp, bool linear) {
	std::vector<float> px;
	{
		py::gil_scoped_release release;
		px = t.render(w, h, spp, linear);
	}
	py::array_t<float> out({h, w, 4});
	std::memcpy(out.mutable_data(), px.data(), px.size() * sizeof(float));
	return out;
}

PYBIND11_MODULE(pyngp, m) {
	m.doc() = "MI355X-native instant-ngp NeRF training/rendering (pyngp-compatible API surface)";
	// not part of the reference API: msgpack (optionally zlib-framed) -> document -> msgpack, used by the snapshot wire-format tests
	m.def("_msgpack_repack", [](py::bytes data, bool input_compressed, bool output_compressed) { return py::bytes(msgpack_repack(std::string(data), input_compressed, output_compressed)); },
		py::arg("data"), py::arg("input_compressed") = false, py::arg("output_compressed") = false);

	// not part of the reference API: the host's EXR reader (image primitive data path), RGBA float32 [h][w][4]
	m.def("read_exr", [](const std::string& path) {
		int w = 0, h = 0; std::vector<float> px;
		exr_lite::read_rgba(path, w, h, px);
		py::array_t<float> out({h, w, 4});
		std::memcpy(out.mutable_data(), px.data(), px.size() * sizeof(float));
		return out;
	});
	// not part of the reference API: the host's OBJ reader (SDF primitive data path), float32 [n_triangles][3][3]
	m.def("read_obj", [](const std::string& path) {
		const std::vector<float> v = mesh_lite::load_obj(path);
		py::array_t<float> out({(py::ssize_t)(v.size() / 9), (py::ssize_t)3, (py::ssize_t)3});
		std::memcpy(out.mutable_data(), v.data(), v.size() * sizeof(float));
		return out;
	});
	m.def("_sharpen_rgba8", [](py::array_t<uint8_t, py::array::c_style | py::array::forcecast> img, float amount, bool has_mask) { // not part of the reference API: the loader's sharpening, for tests
		if (img.ndim() != 3 || img.shape(2) != 4) throw std::runtime_error{"_sharpen_rgba8 expects [h, w, 4] uint8"};
		const int h = (int)img.shape(0), w = (int)img.shape(1);
		const auto r = Testbed::sharpen_rgba8_for_tests(std::vector<uint8_t>(img.data(), img.data() + (size_t)w * h * 4), w, h, amount, has_mask);
		py::array_t<uint16_t> out({h, w, 4}); std::memcpy(out.mutable_data(), r.data(), r.size() * 2); return out;
	}, py::arg("rgba8"), py::arg("amount"), py::arg("has_mask") = false);
	m.def("_natural_less", [](const std::string& a, const std::string& b) { return Testbed::natural_path_less(a, b); }); // not part of the reference API: the loader's frame order, for tests
	m.def("read_stl", [](const std::string& path) { // binary STL (testbed_sdf.cu:1328-1361), float32 [n_triangles][3][3]
		const std::vector<float> v = mesh_lite::load_stl(path);
		py::array_t<float> out({(py::ssize_t)(v.size() / 9), (py::ssize_t)3, (py::ssize_t)3});
		std::memcpy(out.mutable_data(), v.data(), v.size() * sizeof(float));
		return out;
	});
	py::enum_<ETestbedMode>(m, "TestbedMode").value("Nerf", ETestbedMode::Nerf).value("Sdf", ETestbedMode::Sdf).value("Image", ETestbedMode::Image)
		.value("Volume", ETestbedMode::Volume).value("None", ETestbedMode::None).export_values();
	py::enum_<ETrainMode>(m, "TrainMode").value("Nerf", ETrainMode::Nerf).value("Rfl", ETrainMode::Rfl).value("RflRelax", ETrainMode::RflRelax).export_values();
	py::enum_<ELossType>(m, "LossType").value("L2", ELossType::L2).value("L1", ELossType::L1).value("Mape", ELossType::Mape).value("Smape", ELossType::Smape)
		.value("Huber", ELossType::Huber).value("SmoothL1", ELossType::Huber).value("LogL1", ELossType::LogL1).value("RelativeL2", ELossType::RelativeL2).export_values(); // python_api.cu:351-362
	py::enum_<EColorSpace>(m, "ColorSpace").value("Linear", EColorSpace::Linear).value("SRGB", EColorSpace::SRGB).value("VisPosNeg", EColorSpace::VisPosNeg).export_values();
	py::enum_<ETonemapCurve>(m, "TonemapCurve").value("Identity", ETonemapCurve::Identity).value("ACES", ETonemapCurve::ACES).value("Hable", ETonemapCurve::Hable)
		.value("Reinhard", ETonemapCurve::Reinhard).export_values();

	py::enum_<ENerfActivation>(m, "NerfActivation").value("None", ENerfActivation::None).value("ReLU", ENerfActivation::ReLU).value("Logistic", ENerfActivation::Logistic)
		.value("Exponential", ENerfActivation::Exponential).export_values(); // python_api.cu:364-369
	// mode_from_scene / mode_from_string, common_host.cu:144-174
	m.def("mode_from_scene", [](const std::string& scene) {
		py::object os = py::module_::import("os");
		if (!os.attr("path").attr("exists")(scene).cast<bool>()) return ETestbedMode::None;
		std::string ext = scene.substr(scene.find_last_of('.') == std::string::npos ? scene.size() : scene.find_last_of('.') + 1);
		for (auto& c : ext) c = (char)std::tolower((unsigned char)c);
		if (os.attr("path").attr("isdir")(scene).cast<bool>() || ext == "json") return ETestbedMode::Nerf;
		if (ext == "obj" || ext == "stl") return ETestbedMode::Sdf;
		if (ext == "nvdb") return ETestbedMode::Volume;
		return ETestbedMode::Image;
	});
	m.def("mode_from_string", [](std::string str) {
		for (auto& c : str) c = (char)std::tolower((unsigned char)c);
		return str == "nerf" ? ETestbedMode::Nerf : str == "sdf" ? ETestbedMode::Sdf : str == "image" ? ETestbedMode::Image : str == "volume" ? ETestbedMode::Volume : ETestbedMode::None;
	});
	m.def("free_temporary_memory", []() {}); // python_api.cu:309: tcnn's memory arenas; this build holds no arena (fixed per-trainer scratch)
	using V3 = std::array<float, 3>;
	py::class_<BoundingBox>(m, "BoundingBox") // python_api.cu:409-427
		.def(py::init<>()).def(py::init<const V3&, const V3&>())
		.def("center", &BoundingBox::center).def("contains", &BoundingBox::contains).def("diag", &BoundingBox::diag).def("distance", &BoundingBox::distance)
		.def("distance_sq", &BoundingBox::distance_sq).def("enlarge", py::overload_cast<const V3&>(&BoundingBox::enlarge)).def("enlarge", py::overload_cast<const BoundingBox&>(&BoundingBox::enlarge))
		.def("get_vertices", &BoundingBox::get_vertices).def("inflate", &BoundingBox::inflate).def("intersection", &BoundingBox::intersection).def("intersects", &BoundingBox::intersects)
		.def("ray_intersect", &BoundingBox::ray_intersect).def("relative_pos", &BoundingBox::relative_pos).def("signed_distance", &BoundingBox::signed_distance)
		.def_readwrite("min", &BoundingBox::min).def_readwrite("max", &BoundingBox::max);

	py::enum_<ELensMode>(m, "LensMode").value("Perspective", ELensMode::Perspective).value("OpenCV", ELensMode::OpenCV).value("FTheta", ELensMode::FTheta).value("LatLong", ELensMode::LatLong)
		.value("OpenCVFisheye", ELensMode::OpenCVFisheye).value("Equirectangular", ELensMode::Equirectangular).value("Orthographic", ELensMode::Orthographic).export_values(); // python_api.cu:391-399
	py::class_<Lens>(m, "Lens").def(py::init<>()) // python_api.cu:429-433; a VALUE here: assign the whole object back (md.lens = l) to change a lens
		.def_property("mode", [](const Lens& l) { return (ELensMode)l.mode; }, [](Lens& l, ELensMode v) { l.mode = (int)v; })
		.def_readwrite("params", &Lens::params);
	py::class_<Testbed> testbed(m, "Testbed");
	py::class_<ImageMetadata>(testbed, "TrainingImageMetadata")
		.def_readonly("resolution", &ImageMetadata::resolution).def_readonly("focal_length", &ImageMetadata::focal_length)
		.def_readonly("principal_point", &ImageMetadata::principal_point).def_readonly("lens_mode", &ImageMetadata::lens_mode)
		.def_readonly("lens_params", &ImageMetadata::lens_params).def_readonly("rolling_shutter", &ImageMetadata::rolling_shutter)
		.def_property_readonly("lens", [](const ImageMetadata& m) { Lens l; l.mode = m.lens_mode; l.params = m.lens_params; return l; })               // python_api.cu:759 (write through training.set_camera_intrinsics)
		.def_property_readonly("camera_distortion", [](const ImageMetadata& m) { Lens l; l.mode = m.lens_mode; l.params = m.lens_params; return l; }) // :758 legacy name
		.def_readonly("light_dir", &ImageMetadata::light_dir);                                                                                            // :764 (nerf_loader.cu:671-680)
	py::class_<NerfDataset>(testbed, "NerfDataset")
		.def_property_readonly("n_images", [](const NerfDataset& d) { return d.n_images; })
		.def_readonly("metadata", &NerfDataset::metadata).def_readonly("aabb_scale", &NerfDataset::aabb_scale)
		.def_readonly("scale", &NerfDataset::scale).def_readonly("offset", &NerfDataset::offset).def_readonly("paths", &NerfDataset::paths)
		.def_readonly("render_aabb", &NerfDataset::render_aabb).def_readonly("render_aabb_to_local", &NerfDataset::render_aabb_to_local).def_readonly("up", &NerfDataset::up)
		.def_readonly("envmap_resolution", &NerfDataset::envmap_resolution)
		.def_property_readonly("transforms", [](const NerfDataset& d) { return dataset_transforms(d); }) // python_api.cu:768: the (start, end) pair per image, here as two 3 x 4 arrays each (ngp convention)
		.def_readonly("is_hdr", &NerfDataset::is_hdr).def_readonly("xforms", &NerfDataset::xforms).def_readonly("xforms_end", &NerfDataset::xforms_end).def_readonly("from_mitsuba", &NerfDataset::from_mitsuba)
		.def("image", [](const NerfDataset& d, size_t i) {
			if (i >= d.n_images) throw std::runtime_error{"image index out of range"};
			py::array_t<uint8_t> out({d.metadata[i].resolution[1], d.metadata[i].resolution[0], 4});
			std::memcpy(out.mutable_data(), d.pixels[i].data(), d.pixels[i].size());
			return out;
		}, "RGBA8 pixels of training image i (host copy)")
		.def_readonly("sharpen_amount", &NerfDataset::sharpen_amount)
		.def_readonly("n_extra_learnable_dims", &NerfDataset::n_extra_learnable_dims).def_readonly("has_light_dirs", &NerfDataset::has_light_dirs) // (read-only views of nerf_loader.h:84-87; the reference binds none of them)
		.def_property_readonly("n_extra_dims", [](const NerfDataset& d) { return d.n_extra_dims(); })
		.def("image_half", [](const NerfDataset& d, size_t i) {
			if (i >= d.n_images || i >= d.pixels_half.size() || d.pixels_half[i].empty()) throw std::runtime_error{"image_half: no sharpened image (nerf.sharpen == 0?)"};
			py::array_t<uint16_t> out({d.metadata[i].resolution[1], d.metadata[i].resolution[0], 4});
			std::memcpy(out.mutable_data(), d.pixels_half[i].data(), d.pixels_half[i].size() * 2);
			return out;
		}, "sharpened training image i as IEEE binary16 bit patterns [h, w, 4] (view as float16): linear premultiplied RGBA, what the trainer samples when nerf.sharpen > 0")
		.def("image_float", [](const NerfDataset& d, size_t i) {
			if (i >= d.n_images || i >= d.pixels_float.size() || d.pixels_float[i].empty()) throw std::runtime_error{"image_float: image was not set through training.set_image"};
			py::array_t<float> out({d.metadata[i].resolution[1], d.metadata[i].resolution[0], 4});
			std::memcpy(out.mutable_data(), d.pixels_float[i].data(), d.pixels_float[i].size() * 4);
			return out;
		}, "training image i as handed to training.set_image: linear premultiplied RGBA float32 [h, w, 4]")
		.def("depth", [](const NerfDataset& d, size_t i) {
			if (i >= d.n_images || i >= d.depth.size() || d.depth[i].empty()) throw std::runtime_error{"depth: image has no depth (json depth_path + integer_depth_scale)"};
			py::array_t<float> out({d.metadata[i].resolution[1], d.metadata[i].resolution[0]});
			std::memcpy(out.mutable_data(), d.depth[i].data(), d.depth[i].size() * 4);
			return out;
		}, "depth image i in scene units [h, w] (0 = no measurement): integer depth x integer_depth_scale x dataset scale");
	py::class_<NerfTraining>(testbed, "NerfTraining")
		.def_readwrite("near_distance", &NerfTraining::near_distance).def_readwrite("train_mode", &NerfTraining::train_mode)
		.def_readwrite("random_bg_color", &NerfTraining::random_bg_color).def_readwrite("linear_colors", &NerfTraining::linear_colors)
		.def_readwrite("snap_to_pixel_centers", &NerfTraining::snap_to_pixel_centers).def_readwrite("density_grid_decay", &NerfTraining::density_grid_decay)
		.def_readwrite("depth_supervision_lambda", &NerfTraining::depth_supervision_lambda)
		.def_property("depth_loss_type", [](const NerfTraining& t) { return (ELossType)t.depth_loss_type; }, [](NerfTraining& t, ELossType v) { t.depth_loss_type = (int)v; })
		.def_readwrite("sample_focal_plane_proportional_to_error", &NerfTraining::sample_focal_plane_proportional_to_error) // python_api.cu:795
		.def_readwrite("sample_image_proportional_to_error", &NerfTraining::sample_image_proportional_to_error)             // python_api.cu:796
		.def_readwrite("accumulate_error_map", &NerfTraining::accumulate_error_map)
		.def_readwrite("n_images_for_training", &NerfTraining::n_images_for_training) // python_api.cu:783
		.def_property("loss_type", [](const NerfTraining& t) { return (ELossType)(t.loss_type < 0 ? 0 : t.loss_type); }, [](NerfTraining& t, ELossType v) { t.loss_type = (int)v; }) // :785 (unset: the config's)
		.def_property_readonly("transforms", [](const NerfTraining& t) { return dataset_transforms(t.dataset); }) // :798 (no extrinsics optimiser here: the dataset's)
		.def("set_camera_intrinsics", [](NerfTraining& t, int i, float fx, float fy, float cx, float cy, float k1, float k2, float p1, float p2, float k3, float k4, bool fisheye) {
				t.owner->set_camera_intrinsics(i, fx, fy, cx, cy, k1, k2, p1, p2, k3, k4, fisheye); },
			py::arg("frame_idx"), py::arg("fx") = 0.f, py::arg("fy") = 0.f, py::arg("cx") = -0.5f, py::arg("cy") = -0.5f, py::arg("k1") = 0.f, py::arg("k2") = 0.f, py::arg("p1") = 0.f,
			py::arg("p2") = 0.f, py::arg("k3") = 0.f, py::arg("k4") = 0.f, py::arg("is_fisheye") = false) // :814-830
		.def("set_camera_extrinsics", [](NerfTraining& t, int i, py::array_t<float, py::array::c_style | py::array::forcecast> a, bool convert) {
				if (a.size() < 12) throw std::runtime_error{"set_camera_extrinsics expects a 3x4 matrix"};
				std::array<float, 12> mm; for (int k = 0; k < 12; ++k) mm[k] = a.data()[k];
				t.owner->set_camera_extrinsics(i, mm, convert); }, py::arg("frame_idx"), py::arg("camera_to_world"), py::arg("convert_to_ngp") = true) // :831-838
		.def("set_image", [](NerfTraining& t, int i, py::array_t<float, py::array::c_style | py::array::forcecast> img, py::array_t<float, py::array::c_style | py::array::forcecast> depth, float depth_scale) {
				if (img.ndim() != 3 || img.shape(2) != 4) throw std::runtime_error{"image should be (H,W,C) where C=4"};
				const bool has_depth = depth.size() == img.shape(0) * img.shape(1);
				t.owner->set_training_image(i, (int)img.shape(1), (int)img.shape(0), img.data(), has_depth ? depth.data() : nullptr, depth_scale); },
			py::arg("frame_idx"), py::arg("img"), py::arg("depth_img"), py::arg("depth_scale") = 1.0f) // python_api.cu:45-72, 845-852
		.def("get_camera_extrinsics", [](NerfTraining& t, int i) {
				const auto mm = t.owner->get_camera_extrinsics(i); py::array_t<float> out({3, 4}); std::memcpy(out.mutable_data(), mm.data(), sizeof(float) * 12); return out; }, py::arg("frame_idx")) // :839-844
		// camera / exposure optimisation and the sharpness-weighted error map are not part of this build: the switches exist, turning one on says so
		.def_property("optimize_extrinsics", [](const NerfTraining&) { return false; }, [](NerfTraining&, bool v) { if (v) throw std::runtime_error{"optimize_extrinsics: camera optimisation is not part of this build"}; })
		.def_property("optimize_exposure", [](const NerfTraining&) { return false; }, [](NerfTraining&, bool v) { if (v) throw std::runtime_error{"optimize_exposure: exposure optimisation is not part of this build"}; })
		.def_property("optimize_distortion", [](const NerfTraining&) { return false; }, [](NerfTraining&, bool v) { if (v) throw std::runtime_error{"optimize_distortion: distortion-map optimisation is not part of this build"}; })
		.def_property("optimize_focal_length", [](const NerfTraining&) { return false; }, [](NerfTraining&, bool v) { if (v) throw std::runtime_error{"optimize_focal_length: intrinsics optimisation is not part of this build"}; })
		.def_readwrite("optimize_extra_dims", &NerfTraining::optimize_extra_dims).def_readwrite("optimize_per_image_latents", &NerfTraining::optimize_extra_dims) // python_api.cu:789-790
		.def("get_extra_dims", [](NerfTraining& t, int i) { return t.owner->get_extra_dims(i); }, py::arg("frame_idx"))                                        // :810-813
		.def_property("include_sharpness_in_error", [](const NerfTraining&) { return false; }, [](NerfTraining&, bool v) { if (v) throw std::runtime_error{"include_sharpness_in_error: the sharpness-weighted error map is not part of this build"}; })
		.def_readonly("dataset", &NerfTraining::dataset);
	py::class_<Nerf>(testbed, "Nerf")
		.def_readwrite("sharpen", &Nerf::sharpen).def_readwrite("cone_angle_constant", &Nerf::cone_angle_constant)
		.def_readwrite("render_min_transmittance", &Nerf::render_min_transmittance).def_readwrite("rendering_min_transmittance", &Nerf::render_min_transmittance) // python_api.cu:719-720
		.def_property("rgb_activation", [](const Nerf& n) { return (ENerfActivation)(n.rgb_activation >= 0 ? n.rgb_activation : (n.training.dataset.is_hdr ? NGP_ACT_EXPONENTIAL : NGP_ACT_LOGISTIC)); },
			[](Nerf& n, ENerfActivation v) { n.rgb_activation = (int)v; })                                                                                     // :716
		.def_property("density_activation", [](const Nerf& n) { return (ENerfActivation)n.density_activation; }, [](Nerf& n, ENerfActivation v) { n.density_activation = (int)v; }) // :717
		.def_readwrite("visualize_cameras", &Nerf::visualize_cameras)
		.def_readwrite("rendering_extra_dims_from_training_view", &Nerf::rendering_extra_dims_from_training_view)                                          // python_api.cu:725-727
		.def_readwrite("light_dir", &Nerf::light_dir)                                                                                                      // testbed.h:871 (the reference sets it from its GUI only; bound here so that scripts can relight)
		.def("set_rendering_extra_dims_from_training_view", [](Nerf& n, int v) { n.rendering_extra_dims_from_training_view = v; })                           // :735-737
		.def("set_rendering_extra_dims", [](Nerf& n, const std::vector<float>& v) { n.rendering_extra_dims = v; n.rendering_extra_dims_from_training_view = -1; }) // :739
		.def("get_rendering_extra_dims", [](Nerf& n) {                                                                                                     // :741 -> get_rendering_extra_dims_cpu, testbed_nerf.cu:3750
				if (n.training.dataset.n_extra_dims() == 0) return std::vector<float>{};
				std::vector<float> v;
				if (n.rendering_extra_dims_from_training_view >= 0) v = n.training.owner->get_extra_dims(n.rendering_extra_dims_from_training_view);
				else if (!n.rendering_extra_dims.empty()) v = n.rendering_extra_dims;
				else { (void)n.training.owner->get_extra_dims(0); v = n.rendering_extra_dims_default; } // (creates the trainer, which runs reset_extra_dims); the copy of image 0's INITIAL dims (testbed_nerf.cu:3679-3682), not its trained ones
				if (n.training.dataset.has_light_dirs) { // testbed_nerf.cu:3697-3706: with light directions the first three dims are warp_direction(normalize(light_dir)) -- what the renderer uses
					const float len = std::sqrt(n.light_dir[0] * n.light_dir[0] + n.light_dir[1] * n.light_dir[1] + n.light_dir[2] * n.light_dir[2]);
					for (size_t k = 0; k < 3 && k < v.size(); ++k) v[k] = (n.light_dir[k] / len + 1.0f) * 0.5f;
				}
				return v; })
		.def("find_closest_training_view", [](Nerf& n, py::array_t<float, py::array::c_style | py::array::forcecast> a) {
				if (a.size() < 12) throw std::runtime_error{"find_closest_training_view expects a 3x4 matrix"};
				std::array<float, 12> mm; for (int k = 0; k < 12; ++k) mm[k] = a.data()[k];
				return n.training.owner->find_closest_training_view(mm); })                                                                                      // :729-733
		.def_readonly("training", &Nerf::training);

	py::enum_<ERandomMode>(m, "RandomMode").value("Random", ERandomMode::Random).value("Halton", ERandomMode::Halton).value("Sobol", ERandomMode::Sobol)
		.value("Stratified", ERandomMode::Stratified).export_values();                                           // python_api.cu:344-349
	py::enum_<EMeshSdfMode>(m, "MeshSdfMode").value("Watertight", EMeshSdfMode::Watertight).value("Raystab", EMeshSdfMode::Raystab).value("PathEscape", EMeshSdfMode::PathEscape).export_values(); // :376-380
	py::class_<ImageTraining>(testbed, "ImageTraining").def_readwrite("snap_to_pixel_centers", &ImageTraining::snap_to_pixel_centers).def_readwrite("linear_colors", &ImageTraining::linear_colors); // :877-879
	py::class_<ImagePrimitive>(testbed, "Image").def_readonly("training", &ImagePrimitive::training).def_readwrite("random_mode", &ImagePrimitive::random_mode);                                   // :874-875
	py::class_<SdfTraining>(testbed, "SdfTraining").def_readwrite("generate_sdf_data_online", &SdfTraining::generate_sdf_data_online).def_readwrite("surface_offset_scale", &SdfTraining::surface_offset_scale); // :870-872
	py::class_<SdfPrimitive>(testbed, "Sdf").def_readonly("training", &SdfPrimitive::training).def_readwrite("mesh_sdf_mode", &SdfPrimitive::mesh_sdf_mode).def_readwrite("mesh_scale", &SdfPrimitive::mesh_scale)
		.def_readwrite("zero_offset", &SdfPrimitive::zero_offset).def_readwrite("use_triangle_octree", &SdfPrimitive::use_triangle_octree).def_readwrite("calculate_iou_online", &SdfPrimitive::calculate_iou_online); // :855-868 (the members the trainer and the ground truth use; the shading ones belong to SDF rendering)

	testbed
		.def(py::init<>())
		.def(py::init([](ETestbedMode) { return std::make_unique<Testbed>(); }))
		.def("load_file", &Testbed::load_file, py::call_guard<py::gil_scoped_release>())
		.def("load_training_data", &Testbed::load_training_data, "Load training data from a given path.")
		.def("reload_network_from_file", &Testbed::reload_network_from_file, py::arg("path") = "")
		.def("reload_network_from_json", [](Testbed& t, py::object json, const std::string& base) { // python_api.cu:544-550: a dict (or a JSON string)
				const std::string text = py::isinstance<py::str>(json) ? json.cast<std::string>() : py::module_::import("json").attr("dumps")(json).cast<std::string>();
				t.reload_network_from_json_text(text, base); }, py::arg("json"), py::arg("config_base_path") = "")
		.def("_network_config_json", [](const Testbed& t) { return mini_json::dump(t.network_config()); }) // not part of the reference API: the merged config, for tests
		.def_property_readonly("bounding_radius", [](const Testbed&) { return 0.8660254037844386f; }) // testbed.h m_bounding_radius = length(vec3(0.5))
		.def_property("jit_fusion", [](Testbed&) { return true; }, [](Testbed&, bool) {}) // python_api.cu: the reference's JIT-fused kernels on / off; the kernels here are fused at build time either way
		.def_property("max_level_rand_training", [](Testbed&) { return false; }, [](Testbed&, bool v) { if (v) throw std::runtime_error{"max_level_rand_training: random level cut-off is not part of this build"}; })
		.def("reset", &Testbed::reset_network).def("reset_network", &Testbed::reset_network)
		.def("load_snapshot", &Testbed::load_snapshot).def("_nerf_dataset_to_json", &Testbed::nerf_dataset_to_json_text)       // test hooks: snapshot["nerf"]["dataset"] as JSON text (json_binding.h to_json / from_json), host only
		.def("_nerf_dataset_from_json", &Testbed::nerf_dataset_from_json_text)
		.def("dp_gather_state", &Testbed::dp_gather_state, "data parallel (sharded step): gather every rank's optimizer state on all ranks -- a collective, call on every rank before save_snapshot(path, True)")
		.def("save_snapshot", &Testbed::save_snapshot, py::arg("path"), py::arg("include_optimizer_state") = false)
		.def("frame", &Testbed::frame, py::call_guard<py::gil_scoped_release>())
		.def("train", &Testbed::train, py::call_guard<py::gil_scoped_release>())
		.def("want_repl", &Testbed::want_repl)
		.def("compute_image_mse", &Testbed::compute_image_mse, py::arg("quantize") = false)
		.def("_image_pixels", [](const Testbed& t) { // not part of the reference API: the image primitive's pixels as load_image left them (linear RGBA float32 [h, w, 4])
			py::array_t<float> out({(py::ssize_t)t.image_height(), (py::ssize_t)t.image_width(), (py::ssize_t)4});
			std::memcpy(out.mutable_data(), t.image_pixels().data(), t.image_pixels().size() * sizeof(float));
			return out;
		})
		.def("calculate_iou", &Testbed::calculate_iou, py::arg("n_samples") = 128u * 128u * 128u * 4u, py::arg("scale_existing_results_factor") = 0.0f, py::arg("blocking") = true, py::arg("force_use_octree") = false)
		.def("set_camera_to_training_view", &Testbed::set_camera_to_training_view)
		.def("first_training_view", &Testbed::first_training_view).def("last_training_view", &Testbed::last_training_view)
		.def("previous_training_view", &Testbed::previous_training_view).def("next_training_view", &Testbed::next_training_view) // python_api.cu:655-658
		.def("reset_camera", &Testbed::reset_camera).def("reset_accumulation", [](Testbed&, bool, bool) {}, py::arg("due_to_camera_movement") = false, py::arg("immediate_redraw") = true) // :535-541 (every render() starts a fresh accumulation here)
		.def("create_empty_nerf_dataset", &Testbed::create_empty_nerf_dataset, py::arg("n_images"), py::arg("aabb_scale") = 1, py::arg("is_hdr") = false) // python_api.cu:444-451
		.def("clear_training_data", &Testbed::clear_training_data)                                                            // :453
		.def("n_params", &Testbed::n_params).def("n_encoding_params", &Testbed::n_encoding_params)                            // :561-562
		.def_readwrite("aabb", &Testbed::aabb).def_readwrite("raw_aabb", &Testbed::raw_aabb).def_readwrite("render_aabb", &Testbed::render_aabb) // :641-645
		.def_readwrite("render_aabb_to_local", &Testbed::render_aabb_to_local)
		.def_property("render_lens", &Testbed::render_lens, &Testbed::set_render_lens) // python_api.cu: m_render_lens
		.def_readwrite("visualize_unit_cube", &Testbed::visualize_unit_cube)
		.def_readwrite("up_dir", &Testbed::up_dir).def_readwrite("zoom", &Testbed::zoom).def_readwrite("render_near_distance", &Testbed::render_near_distance)
		.def_readwrite("relative_focal_length", &Testbed::relative_focal_length).def_readwrite("screen_center", &Testbed::screen_center) // :649-651
		.def_property("fov_xy", &Testbed::fov_xy, &Testbed::set_fov_xy).def_property("scale", &Testbed::scale, &Testbed::set_scale)   // :639, 647
		.def_property("look_at", &Testbed::look_at, &Testbed::set_look_at).def_property("view_dir", &Testbed::view_dir, &Testbed::set_view_dir) // :664-665
		.def_property("camera_matrix", [](Testbed& t) { const auto mm = t.camera_matrix_row_major(); py::array_t<float> out({3, 4}); std::memcpy(out.mutable_data(), mm.data(), sizeof(float) * 12); return out; },
			[](Testbed& t, py::array_t<float, py::array::c_style | py::array::forcecast> a) {
				if (a.size() < 12) throw std::runtime_error{"camera_matrix expects a 3x4 matrix"};
				std::array<float, 12> mm; for (int k = 0; k < 12; ++k) mm[k] = a.data()[k];
				t.set_camera_matrix_row_major(mm); })                                                                             // :660
		.def_property("shall_train_encoding", [](Testbed&) { return true; }, [](Testbed&, bool v) { if (!v) throw std::runtime_error{"shall_train_encoding = False: freezing the encoding is not part of this build"}; }) // :623
		.def_property("shall_train_network", [](Testbed&) { return true; }, [](Testbed&, bool v) { if (!v) throw std::runtime_error{"shall_train_network = False: freezing the MLPs is not part of this build"}; })  // :624
		.def("set_nerf_camera_matrix", [](Testbed& t, py::array_t<float, py::array::c_style | py::array::forcecast> a) {
			if (a.size() < 12) throw std::runtime_error{"set_nerf_camera_matrix expects a 3x4 matrix"};
			std::array<float, 12> mm; for (int i = 0; i < 12; ++i) mm[i] = a.data()[i];
			t.set_nerf_camera_matrix(mm);
		})
		.def("render", [](Testbed& t, int w, int h, int spp, bool linear, float start_t, float end_t, float, float shutter_fraction) {
				if (start_t < 0.f) return render_to_numpy(t, w, h, spp, linear); // no path animation (python_api.cu:155)
				std::vector<float> px;
				{ py::gil_scoped_release release; px = t.render_path_frame(w, h, spp, linear, start_t, end_t, shutter_fraction); }
				py::array_t<float> out({h, w, 4}); std::memcpy(out.mutable_data(), px.data(), px.size() * sizeof(float)); return out; },
			py::arg("width") = 1920, py::arg("height") = 1080, py::arg("spp") = 1, py::arg("linear") = true, py::arg("start_t") = -1.f, py::arg("end_t") = -1.f,
			py::arg("fps") = 30.f, py::arg("shutter_fraction") = 1.0f)
		.def("render_with_depth", [](Testbed& t, int w, int h, int spp, bool linear, float, float, float, float) { // python_api.cu:520-532: (rgba [h, w, 4], depth [h, w])
				std::vector<float> px, depth;
				{ py::gil_scoped_release release; px = t.render(w, h, spp, linear, &depth); }
				py::array_t<float> a({h, w, 4}), d({h, w});
				std::memcpy(a.mutable_data(), px.data(), px.size() * sizeof(float));
				if (depth.size() == (size_t)w * h) std::memcpy(d.mutable_data(), depth.data(), depth.size() * sizeof(float)); else std::fill(d.mutable_data(), d.mutable_data() + (size_t)w * h, 0.f);
				return py::make_tuple(a, d); },
			py::arg("width") = 1920, py::arg("height") = 1080, py::arg("spp") = 1, py::arg("linear") = true, py::arg("start_t") = -1.f, py::arg("end_t") = -1.f,
			py::arg("fps") = 30.f, py::arg("shutter_fraction") = 1.0f)
		.def("init_window", [](Testbed&, int, int, bool, bool) { throw std::runtime_error{"init_window: GUI is out of scope of this build (headless MI355X path)"}; },
			py::arg("width"), py::arg("height"), py::arg("hidden") = false, py::arg("second_window") = false)
		.def("init_vr", [](Testbed&) { throw std::runtime_error{"init_vr: VR is out of scope of this build"}; })
		.def("load_camera_path", &Testbed::load_camera_path) // python_api.cu:563
		.def("set_camera_from_time", &Testbed::set_camera_from_time)
		.def_property("camera_smoothing", [](Testbed&) { return false; }, [](Testbed&, bool v) { if (v) throw std::runtime_error{"camera_smoothing: the exponential camera smoothing (tcnn matrix logarithm) is not part of this build; render the path unsmoothed"}; })
		// data-parallel training (new): rank 0 creates the id, every rank passes the same 128 bytes
		.def_static("comm_unique_id", []() { return py::bytes(Testbed::comm_unique_id()); })
		.def("comm_init", [](Testbed& t, uint32_t rank, uint32_t world, py::bytes id) { t.comm_init(rank, world, std::string(id)); }, py::arg("rank"), py::arg("world_size"), py::arg("unique_id"))
		.def("compute_and_save_marching_cubes_mesh", [](Testbed&, py::args, py::kwargs) { throw std::runtime_error{"marching cubes export is out of scope of this build"}; })
		.def_readwrite("root_dir", &Testbed::root_dir)
		.def_readwrite("mode", &Testbed::mode)
		.def_readwrite("shall_train", &Testbed::shall_train)
		.def_readonly("training_step", &Testbed::training_step)
		.def_readonly("loss", &Testbed::loss)
		.def_readwrite("exposure", &Testbed::exposure)
		.def_readwrite("background_color", &Testbed::background_color)
		.def_readwrite("snap_to_pixel_centers", &Testbed::snap_to_pixel_centers)
		.def_readwrite("render_with_lens_distortion", &Testbed::render_with_lens_distortion)
		.def_readwrite("render_ground_truth", &Testbed::render_ground_truth)
		.def_readwrite("render_groundtruth", &Testbed::render_ground_truth) // the reference's spelling (python_api.cu:625)
		.def_readwrite("color_space", &Testbed::color_space)
		.def_readwrite("tonemap_curve", &Testbed::tonemap_curve)
		.def_readwrite("fov_axis", &Testbed::fov_axis)
		.def_property("fov", &Testbed::fov, &Testbed::set_fov)
		.def_readwrite("training_batch_size", &Testbed::training_batch_size)
		.def_readwrite("seed", &Testbed::seed)
		.def_property_readonly("nerf", [](Testbed& t) -> Nerf& { return t.nerf; }, py::return_value_policy::reference_internal)
		.def_property_readonly("image", [](Testbed& t) -> ImagePrimitive& { return t.image; }, py::return_value_policy::reference_internal)
		.def_property_readonly("sdf", [](Testbed& t) -> SdfPrimitive& { return t.sdf; }, py::return_value_policy::reference_internal)
		.def_property_readonly("rays_per_batch", [](Testbed& t) { return t.stats().rays_per_batch; })
		.def_property_readonly("measured_batch_size", [](Testbed& t) { return t.stats().measured_batch_size; });

	m.def("read_image", [](const std::string& path) {
		int w = 0, h = 0; std::vector<uint8_t> px;
		if (!Testbed::read_image_builtin(path, w, h, px)) throw std::runtime_error{"read_image: '" + path + "' is not a PNG / JPEG the built-in readers decode"};
		py::array_t<uint8_t> out({h, w, 4});
		std::memcpy(out.mutable_data(), px.data(), px.size());
		return out;
	}, "RGBA8 [h, w, 4] by the loader's built-in PNG / JPEG readers (no Pillow)");
	m.def("read_depth_png", [](const std::string& path) {
		int w = 0, h = 0; std::vector<uint16_t> px;
		if (!Testbed::read_depth_png16(path, w, h, px)) throw std::runtime_error{"read_depth_png: could not decode '" + path + "'"};
		py::array_t<uint16_t> out({h, w});
		std::memcpy(out.mutable_data(), px.data(), px.size() * 2);
		return out;
	}, "one 16-bit channel [h, w] of a PNG, as the loader reads depth images");
	// decoder hook for images the built-in readers do not decode (BMP, TGA, arithmetic-coded JPEG, ...): Pillow, if importable
	m.def("_set_image_decoder", [](py::function fn) {
		Testbed::s_fallback_decoder = [fn](const std::string& path, int& w, int& h, std::vector<uint8_t>& rgba) {
			py::gil_scoped_acquire gil;
			py::object r = fn(path);
			if (r.is_none()) return false;
			auto arr = r.cast<py::array_t<uint8_t, py::array::c_style | py::array::forcecast>>();
			if (arr.ndim() != 3 || arr.shape(2) != 4) return false;
			h = (int)arr.shape(0); w = (int)arr.shape(1);
			rgba.assign(arr.data(), arr.data() + (size_t)w * h * 4);
			return true;
		};
	});
	py::exec(R"(
def _pil_decoder(path):
    try:
        from PIL import Image
        import numpy as np
        return np.ascontiguousarray(np.asarray(Image.open(path).convert('RGBA'), dtype=np.uint8))
    except Exception:
        return None
)", m.attr("__dict__"));
	m.attr("_set_image_decoder")(m.attr("_pil_decoder"));
	// drop the Python callable before the interpreter is finalised (static storage outlives the GIL)
	py::module_::import("atexit").attr("register")(py::cpp_function([]() { Testbed::s_fallback_decoder = nullptr; }));
	// root_dir default: the package directory that holds configs/
	m.attr("__package_dir__") = py::module_::import("os").attr("path").attr("dirname")(m.attr("__file__"));
	Testbed::s_default_root_dir = m.attr("__package_dir__").cast<std::string>();
}
