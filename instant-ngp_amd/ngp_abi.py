"""ctypes mirror of include/ngp_hip.h (POD structs + loader for libngp_hip.so).

This is plumbing for tests/ and bench.py: it binds the C-ABI exactly as a reference maintainer's
FFI stub would. No compute happens here; if the HIP library is missing the
loader raises -- there is no CPU fallback on the product path.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NGP_HIP_LIB") or os.path.join(HERE, "libngp_hip.so")  # NGP_HIP_LIB: another build of the same library (A / B measurements of two builds on one box)

u32, i32, f32, u64, u8, u16 = C.c_uint32, C.c_int32, C.c_float, C.c_uint64, C.c_uint8, C.c_uint16
vp = C.c_void_p

LENS_PERSPECTIVE, LENS_OPENCV, LENS_FTHETA, LENS_LATLONG, LENS_OPENCV_FISHEYE, LENS_EQUIRECTANGULAR, LENS_ORTHOGRAPHIC = 0, 1, 2, 3, 4, 5, 6
IMAGE_BYTE, IMAGE_HALF, IMAGE_FLOAT = 1, 2, 3
ACT_NONE, ACT_RELU, ACT_LOGISTIC, ACT_EXPONENTIAL = 0, 1, 2, 3
DBG_K4_ZERO_PADDING = 1 << 26  # ngp_debug_set_flags test hook (csrc/ngp_kernels.hpp)
LOSS_L2, LOSS_L1, LOSS_MAPE, LOSS_SMAPE, LOSS_HUBER, LOSS_LOGL1, LOSS_RELATIVE_L2 = range(7)


class Aabb(C.Structure):
    _fields_ = [("min", f32 * 3), ("max", f32 * 3)]


class Ray(C.Structure):
    _fields_ = [("o", f32 * 3), ("d", f32 * 3)]


class Pcg32(C.Structure):
    _fields_ = [("state", u64), ("inc", u64)]


class ImageMeta(C.Structure):
    _fields_ = [("pixels", vp), ("image_data_type", i32), ("lens_mode", i32), ("resolution", i32 * 2),
                ("principal_point", f32 * 2), ("focal_length", f32 * 2), ("rolling_shutter", f32 * 4),
                ("lens_params", f32 * 7), ("_pad", f32), ("depth", vp)]


class Xform(C.Structure):
    _fields_ = [("start", f32 * 12), ("end", f32 * 12)]


class ModelConfig(C.Structure):
    _fields_ = [("n_levels", u32), ("n_features_per_level", u32), ("log2_hashmap_size", u32), ("base_resolution", u32),
                ("per_level_scale", f32), ("n_neurons", u32), ("n_hidden_layers", u32), ("n_hidden_layers_rgb", u32),
                ("sh_degree", u32), ("n_extra_dims", u32),
                ("learning_rate", f32), ("beta1", f32), ("beta2", f32), ("epsilon", f32), ("l2_reg", f32),
                ("ema_decay", f32), ("decay_start", u32), ("decay_interval", u32), ("decay_base", f32), ("ema_full_precision", u32)]


class EncMlpConfig(C.Structure):
    _fields_ = [("n_pos_dims", u32), ("n_levels", u32), ("n_features_per_level", u32), ("log2_hashmap_size", u32),
                ("base_resolution", u32), ("per_level_scale", f32), ("n_neurons", u32), ("n_hidden_layers", u32), ("n_output_dims", u32)]


class OptimizerConfig(C.Structure):
    _fields_ = [("learning_rate", f32), ("beta1", f32), ("beta2", f32), ("epsilon", f32), ("l2_reg", f32), ("ema_decay", f32),
                ("decay_start", u32), ("decay_interval", u32), ("decay_base", f32)]


class ImageOptions(C.Structure):
    _fields_ = [("snap_to_pixel_centers", i32), ("linear_colors", i32), ("stratified", i32), ("loss_type", i32), ("loss_scale", f32),
                ("batch_size", u32), ("seed", u64)]


def default_image_options(**kw):
    """Testbed::m_image.training defaults (testbed.h:966-970), configs/image/base.json loss, BASELINE config 0 batch"""
    o = ImageOptions(snap_to_pixel_centers=1, linear_colors=0, stratified=1, loss_type=LOSS_L2, loss_scale=128.0, batch_size=1 << 16, seed=1337)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class SdfOptions(C.Structure):
    _fields_ = [("loss_type", i32), ("loss_scale", f32), ("batch_size", u32), ("seed", u64), ("surface_offset_scale", f32), ("zero_offset", f32)]


def default_sdf_options(**kw):
    """Testbed::m_sdf defaults (testbed.h:909-938), configs/sdf/base.json loss (MAPE), m_training_batch_size"""
    o = SdfOptions(loss_type=LOSS_MAPE, loss_scale=128.0, batch_size=1 << 18, seed=1337, surface_offset_scale=1.0, zero_offset=0.0)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _auto_per_level_scale(desired_resolution, base_resolution, n_levels):
    """testbed.cu:4249-4253 in the reference's float arithmetic: std::exp(std::log(desired / base) / (n_levels - 1))"""
    import numpy as np
    return float(np.exp(np.log(np.float32(desired_resolution) / np.float32(base_resolution)) / np.float32(n_levels - 1)))


def image_encmlp_config(log2_hashmap_size=19, image_resolution=1024):
    """BASELINE.json config 0 (configs/image/base.json with T = 2^19): 2-D grid, desired finest resolution = max(image res) / 2."""
    return EncMlpConfig(2, 16, 2, log2_hashmap_size, 16, _auto_per_level_scale(image_resolution / 2.0, 16, 16), 64, 2, 3)


def sdf_encmlp_config(log2_hashmap_size=19):
    """BASELINE.json config 4 (configs/sdf/base.json): 3-D grid, desired finest resolution 2048 => per_level_scale 1.3819."""
    return EncMlpConfig(3, 16, 2, log2_hashmap_size, 16, _auto_per_level_scale(2048.0, 16, 16), 64, 2, 1)


class NerfOptions(C.Structure):
    _fields_ = [("rgb_activation", i32), ("density_activation", i32), ("loss_type", i32), ("random_bg_color", i32),
                ("snap_to_pixel_centers", i32), ("linear_colors", i32), ("color_space_srgb", i32),
                ("background_color", f32 * 3), ("near_distance", f32), ("density_grid_decay", f32),
                ("cone_angle_constant", f32), ("max_cascade", u32), ("target_batch_size", u32), ("loss_scale", f32),
                ("seed", u64), ("rank", u32), ("world_size", u32), ("train_mode", i32),
                ("depth_supervision_lambda", f32), ("depth_loss_type", i32),
                ("sample_focal_plane_proportional_to_error", i32), ("sample_image_proportional_to_error", i32), ("accumulate_error_map", i32)]


class NerfStats(C.Structure):
    _fields_ = [("training_step", u32), ("rays_per_batch", u32), ("n_rays_last", u32), ("measured_batch_size", u32),
                ("measured_batch_size_before_compaction", u32), ("loss", f32), ("total_rays", u64), ("total_samples", u64),
                ("network_evaluations", u32), ("reserved", u32)]


class RenderParams(C.Structure):
    _fields_ = [("resolution", i32 * 2), ("focal_length", f32 * 2), ("screen_center", f32 * 2), ("camera", f32 * 12),
                ("lens_mode", i32), ("lens_params", f32 * 7), ("spp_index", u32), ("snap_to_pixel_centers", i32),
                ("min_transmittance", f32), ("near_distance", f32), ("use_inference_params", i32), ("render_aabb", Aabb)]


def base_model_config(aabb_scale=1, **kw):
    """configs/nerf/base.json + the per_level_scale derivation of testbed.cu:4241-4255."""
    import math
    c = ModelConfig(n_levels=8, n_features_per_level=4, log2_hashmap_size=19, base_resolution=16, per_level_scale=0.0,
                    n_neurons=64, n_hidden_layers=1, n_hidden_layers_rgb=2, sh_degree=4, n_extra_dims=0,
                    learning_rate=1e-2, beta1=0.9, beta2=0.99, epsilon=1e-15, l2_reg=1e-6, ema_decay=0.95,
                    decay_start=20000, decay_interval=10000, decay_base=0.33)
    for k, v in kw.items():
        setattr(c, k, v)
    if c.per_level_scale <= 0:
        import numpy as np
        c.per_level_scale = float(np.exp(np.log(np.float32(2048.0) * np.float32(aabb_scale) / np.float32(c.base_resolution)) / (c.n_levels - 1)))
    return c


def default_nerf_options(aabb_scale=1, **kw):
    """Member defaults of ngp::Testbed for NeRF training (testbed.h:793-822, 869, 1089; testbed_nerf.cu:2433-2440)."""
    max_cascade = 0
    while (1 << max_cascade) < aabb_scale:
        max_cascade += 1
    o = NerfOptions(rgb_activation=ACT_LOGISTIC, density_activation=ACT_EXPONENTIAL, loss_type=LOSS_HUBER, random_bg_color=1,
                    snap_to_pixel_centers=1, linear_colors=0, color_space_srgb=0, near_distance=0.1, density_grid_decay=0.95,
                    cone_angle_constant=0.0 if aabb_scale <= 1 else 1.0 / 256.0, max_cascade=max_cascade,
                    target_batch_size=1 << 18, loss_scale=128.0, seed=1337, rank=0, world_size=1, train_mode=0,
                    depth_supervision_lambda=0.0, depth_loss_type=LOSS_L1)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def scene_aabb(aabb_scale=1):
    """m_aabb = BoundingBox{0.5,0.5}.inflate(0.5*aabb_scale), testbed_nerf.cu:2424-2425."""
    h = 0.5 * min(128, aabb_scale)
    return Aabb((f32 * 3)(0.5 - h, 0.5 - h, 0.5 - h), (f32 * 3)(0.5 + h, 0.5 + h, 0.5 + h))


_lib = None


_hooks = None


def load_testhooks():
    """dlopen libngp_hip_testhooks.so: the device code's leaf functions compiled for the host (include/ngp_hip_host_hooks.h) -- test infrastructure, not part of the product library"""
    global _hooks
    if _hooks is None:
        path = os.path.join(HERE, "libngp_hip_testhooks.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _hooks = C.CDLL(path)
    return _hooks


def load_hip():
    """dlopen libngp_hip.so (built in-tree by __graft_entry__.build()). Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        _lib.ngp_last_error.restype = C.c_char_p
        _lib.ngp_model_learning_rate.restype = f32
        _lib.ngp_model_step.restype = u32
        _lib.ngp_model_serialized_size.restype = u64
        if hasattr(_lib, "ngp_model_state_offset"):  # (absent from older builds loaded through NGP_HIP_LIB)
            _lib.ngp_model_state_offset.restype = u64
            _lib.ngp_model_state_offset.argtypes = [u64, C.c_int]
        _lib.ngp_encmlp_learning_rate.restype = f32
        _lib.ngp_encmlp_step.restype = u32
    return _lib


def check(lib, rc):
    if rc != 0:
        raise RuntimeError(lib.ngp_last_error().decode())
