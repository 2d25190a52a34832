"""ORACLE -- TEST INFRASTRUCTURE ONLY. ctypes loader for oracle/liboracle.so (CPU restatement).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def load():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        f32, u32, u64, u16, vp = C.c_float, C.c_uint32, C.c_uint64, C.c_uint16, C.c_void_p
        L.ora_last_error.restype = C.c_char_p
        for name in ("ora_pcg32_next_float", "ora_calc_dt", "ora_advance_n_steps", "ora_to_stepping_space", "ora_from_stepping_space",
                     "ora_advance_to_next_voxel", "ora_warp_dt", "ora_unwarp_dt", "ora_ld_random_val", "ora_srgb_to_linear",
                     "ora_linear_to_srgb", "ora_k_density_grid_mean", "ora_model_learning_rate", "ora_nerf_mean_density"):
            getattr(L, name).restype = f32
        for name in ("ora_calc_dt", "ora_to_stepping_space", "ora_from_stepping_space"):
            getattr(L, name).argtypes = [f32, f32]
        L.ora_advance_n_steps.argtypes = [f32, f32, f32]
        L.ora_advance_to_next_voxel.argtypes = [f32, f32, vp, vp, u32]
        L.ora_warp_dt.argtypes = [f32]; L.ora_unwarp_dt.argtypes = [f32]
        L.ora_srgb_to_linear.argtypes = [f32]; L.ora_linear_to_srgb.argtypes = [f32]
        L.ora_ld_random_val.argtypes = [u32, u32, u32]
        L.ora_nerf_set_mean_density.argtypes = [vp, f32]
        L.ora_model_optimizer_step.argtypes = [vp, f32]
        L.ora_k_ema_grid_samples.argtypes = [u32, f32, vp, vp]
        L.ora_nerf_update_density_grid.argtypes = [vp, f32, u32, u32]
        L.ora_sobol.restype = u32; L.ora_morton3D.restype = u32; L.ora_morton3D_invert.restype = u32; L.ora_pcg32_next_uint.restype = u32
        L.ora_hfma.restype = u16; L.ora_hfma.argtypes = [u16, u16, u16]
        L.ora_model_set_step.argtypes = [vp, u32, f32]
        for name in ("ora_model_n_params", "ora_model_n_mlp_params"):
            getattr(L, name).restype = u64; getattr(L, name).argtypes = [vp]
        for name in ("ora_model_params_fp", "ora_model_params", "ora_model_params_inference", "ora_model_gradients", "ora_model_adam_m",
                     "ora_model_adam_v", "ora_model_adam_steps", "ora_model_ema", "ora_nerf_density_grid", "ora_nerf_bitfield"):
            getattr(L, name).restype = vp; getattr(L, name).argtypes = [vp]
        for name in ("ora_encmlp_n_params", "ora_encmlp_n_mlp"):
            getattr(L, name).restype = u64; getattr(L, name).argtypes = [vp]
        L.ora_encmlp_params_fp.restype = vp; L.ora_encmlp_params_fp.argtypes = [vp]
        L.ora_encmlp_gradients.restype = vp; L.ora_encmlp_gradients.argtypes = [vp]
        L.ora_encmlp_loss_and_gradient.restype = f32; L.ora_encmlp_loss_and_gradient.argtypes = [vp, C.c_int, vp, u32, vp, u32, u32, f32, vp]
        L.ora_encmlp_destroy.argtypes = [vp]; L.ora_encmlp_sync_half.argtypes = [vp]
        L.ora_encmlp_optimizer_step.argtypes = [vp, f32]; L.ora_encmlp_params.restype = vp; L.ora_encmlp_params.argtypes = [vp]
        for name in ("ora_model_destroy", "ora_nerf_destroy", "ora_model_sync_half", "ora_nerf_update_mean_and_bitfield"):
            getattr(L, name).argtypes = [vp]
        _lib = L
    return _lib
