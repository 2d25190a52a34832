"""Shared helpers for the parity tests: small synthetic dataset, oracle/HIP handles, device buffers."""
import ctypes as C
import math

import numpy as np

import ngp_abi as A

f32p = C.POINTER(C.c_float)


def ptr(a):
    """host pointer of a numpy array"""
    return a.ctypes.data_as(C.c_void_p)


def dptr(t):
    """device pointer of a torch cuda tensor"""
    return C.c_void_p(t.data_ptr())


def half_to_f32(a_u16):
    return np.asarray(a_u16, dtype=np.uint16).view(np.float16).astype(np.float32)


def make_small_dataset(n_images=6, res=48):
    """Tiny synthetic 'lego-like' dataset (host numpy): images RGBA8, metadata + xforms ctypes arrays."""
    import synth_scene
    images, xforms, meta, _ = synth_scene.make_dataset(n_images, res, "cpu")
    imgs = [np.ascontiguousarray(im.numpy()) for im in images]
    return imgs, xforms, meta


def build_meta_arrays(imgs, xforms, meta, pixel_ptrs):
    n = len(imgs)
    M = (A.ImageMeta * n)()
    X = (A.Xform * n)()
    for i in range(n):
        M[i].pixels = pixel_ptrs[i]
        M[i].image_data_type = A.IMAGE_BYTE
        M[i].lens_mode = A.LENS_PERSPECTIVE
        M[i].resolution[0], M[i].resolution[1] = meta["resolution"]
        M[i].principal_point[0], M[i].principal_point[1] = meta["principal_point"]
        M[i].focal_length[0], M[i].focal_length[1] = meta["focal_length"]
        for k in range(12):
            X[i].start[k] = float(xforms[i][k]); X[i].end[k] = float(xforms[i][k])
    return M, X


def host_meta(imgs, xforms, meta):
    return build_meta_arrays(imgs, xforms, meta, [im.ctypes.data for im in imgs])


def device_meta(imgs, xforms, meta, torch):
    dev_imgs = [torch.from_numpy(im).cuda() for im in imgs]
    M, X = build_meta_arrays(imgs, xforms, meta, [t.data_ptr() for t in dev_imgs])
    # the metadata / xform arrays themselves must live on the device for the stand-alone kernels
    Md = torch.from_numpy(np.frombuffer(bytes(M), dtype=np.uint8).copy()).cuda()
    Xd = torch.from_numpy(np.frombuffer(bytes(X), dtype=np.uint8).copy()).cuda()
    return dev_imgs, M, X, Md, Xd


def random_coords(n, seed=0, ray_coherent=False):
    """NerfCoordinate AoS [n,7]: pos in [0,1]^3, dt (warped), dir in [0,1]^3"""
    rng = np.random.default_rng(seed)
    c = np.zeros((n, 7), dtype=np.float32)
    if ray_coherent:
        n_rays = max(1, n // 32)
        o = rng.uniform(0.2, 0.8, (n_rays, 3)); d = rng.normal(size=(n_rays, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        t = (np.arange(n) % 32) * (math.sqrt(3) / 1024)
        r = np.arange(n) // 32 % n_rays
        c[:, 0:3] = np.clip(o[r] + d[r] * t[:, None], 0, 1)
        c[:, 4:7] = (d[r] + 1) * 0.5
    else:
        c[:, 0:3] = rng.uniform(0, 1, (n, 3))
        d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        c[:, 4:7] = (d + 1) * 0.5
    c[:, 3] = 0.0
    return c


class OraModel:
    def __init__(self, L, cfg, seed=1337):
        self.L = L
        self.h = C.c_void_p()
        assert L.ora_model_create(C.byref(cfg), C.c_uint64(seed), C.byref(self.h)) == 0, L.ora_last_error()
        self.n = L.ora_model_n_params(self.h)
        self.n_mlp = L.ora_model_n_mlp_params(self.h)

    def arr(self, fn, dtype):
        p = getattr(self.L, fn)(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(self.n,))

    @property
    def params_fp(self):
        return self.arr("ora_model_params_fp", np.float32)

    @property
    def params(self):
        return self.arr("ora_model_params", np.uint16)

    @property
    def params_inf(self):
        return self.arr("ora_model_params_inference", np.uint16)

    @property
    def grads(self):
        return self.arr("ora_model_gradients", np.uint16)

    def load_serialized(self, blob):
        """the device model's state (ngp_model_serialize_host with optimizer state: header, master, Adam m / v / steps, EMA) into the oracle's trainer"""
        hdr = np.frombuffer(blob[:32].tobytes(), np.uint32)
        step = int(hdr[4]); lr = float(np.frombuffer(blob[24:28].tobytes(), np.float32)[0])
        n = self.n
        body = blob[32:].view(np.uint32).reshape(5, n)
        self.params_fp[:] = body[0].view(np.float32)
        self.arr("ora_model_adam_m", np.float32)[:] = body[1].view(np.float32)
        self.arr("ora_model_adam_v", np.float32)[:] = body[2].view(np.float32)
        self.arr("ora_model_adam_steps", np.uint32)[:] = body[3]
        self.L.ora_model_sync_half(self.h)                       # params = params_inference = half(master) ...
        self.arr("ora_model_ema", np.float32)[:] = body[4].view(np.float32)
        self.params_inf[:] = body[4].view(np.float32).astype(np.float16).view(np.uint16)   # ... then the inference weights = half(EMA)
        self.L.ora_model_set_step(self.h, step, lr)

    def inference(self, coords, use_inf=False):
        n = coords.shape[0]
        out = np.zeros((n, 4), dtype=np.uint16)
        self.L.ora_model_inference(self.h, ptr(coords), 7, n, ptr(out), 4, int(use_inf))
        return out

    def density(self, pos, use_inf=False):
        n = pos.shape[0]
        out = np.zeros((n,), dtype=np.uint16)
        self.L.ora_model_density(self.h, ptr(pos), pos.shape[1], n, ptr(out), 1, int(use_inf))
        return out

    def encode(self, pos):
        n = pos.shape[0]
        out = np.zeros((n, 32), dtype=np.uint16)
        self.L.ora_model_encode(self.h, ptr(pos), pos.shape[1], n, ptr(out))
        return out

    def training_step(self, coords, dl, grid_sum_mode=0):
        """grid_sum_mode (test aids): 0 = the reference (half contributions added one by one in half); 1 = the same half contributions summed per table entry in double, one
        rounding; 2 = the unrounded products dL/d(enc) * weight summed in double, one rounding"""
        if grid_sum_mode:
            self.L.ora_model_training_step_exact_sums(self.h, ptr(coords), 7, coords.shape[0], ptr(dl), dl.shape[1], int(grid_sum_mode))
        else:
            self.L.ora_model_training_step(self.h, ptr(coords), 7, coords.shape[0], ptr(dl), dl.shape[1])

    def __del__(self):
        try:
            self.L.ora_model_destroy(self.h)
        except Exception:
            pass


class HipModel:
    def __init__(self, lib, cfg, seed=1337):
        self.lib = lib
        self.h = C.c_void_p()
        A.check(lib, lib.ngp_model_create(C.byref(cfg), C.c_uint64(seed), C.byref(self.h)))
        n, nm = C.c_uint64(), C.c_uint64()
        lib.ngp_model_n_params(self.h, C.byref(n), C.byref(nm))
        self.n, self.n_mlp = n.value, nm.value

    def ptrs(self):
        m, p, i, g = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        self.lib.ngp_model_param_ptrs(self.h, C.byref(m), C.byref(p), C.byref(i), C.byref(g))
        return m.value, p.value, i.value, g.value

    def read(self, which, torch):
        """copy a parameter-sized device array to numpy: 'master' (f32) | 'params' | 'inference' | 'grads' (u16)"""
        m, p, i, g = self.ptrs()
        src = {"master": m, "params": p, "inference": i, "grads": g}[which]
        dt, nb = (np.float32, 4) if which == "master" else (np.uint16, 2)
        out = np.empty(self.n, dtype=dt)
        hipr = torch.cuda.current_stream()  # noqa: F841 (forces context)
        torch.cuda.synchronize()
        rt = C.CDLL("libamdhip64.so")
        rc = rt.hipMemcpy(ptr(out), C.c_void_p(src), C.c_size_t(self.n * nb), 2)
        assert rc == 0
        return out

    def set_params(self, params_f32):
        p = np.ascontiguousarray(params_f32, dtype=np.float32)
        A.check(self.lib, self.lib.ngp_model_set_params_host(self.h, ptr(p), C.c_uint64(p.size)))

    def __del__(self):
        try:
            self.lib.ngp_model_destroy(self.h)
        except Exception:
            pass
