"""CPU: training pixels drawn in proportion to the accumulated error (Testbed::Nerf::Training::error_map; nerf_device.cuh:497-599,
testbed_nerf.cu:1042-1071, 1530-1580, 2753-2759, 2791-2855) -- the oracle's restatement against independent numpy models and against
the statistics the construction promises (the CDFs are what the reference samples from: frequencies must follow them)."""
import ctypes as C

import numpy as np
import pytest

import ngp_abi as A
from common import OraModel, host_meta, make_small_dataset, ptr


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _cdfs(ora, err):
    n_img, h, w = err.shape
    cxy = np.zeros_like(err); cy = np.zeros((n_img, h), np.float32); ci = np.zeros(n_img, np.float32)
    ora.ora_construct_error_cdfs(n_img, w, h, _f32p(err), _f32p(cxy), _f32p(cy), _f32p(ci))
    return cxy, cy, ci


def _error_field(n_img=5, h=20, w=28, seed=0):
    rs = np.random.default_rng(seed)
    err = rs.uniform(0.0, 1e-3, (n_img, h, w)).astype(np.float32)
    err[1, 3:6, 10:14] += 0.05   # hot spots
    err[3, 15:, :4] += 0.02
    err[2] = 0.0                 # an image that never received a ray: the 1e-10 floor keeps its CDFs finite
    return err


def test_construct_cdfs_vs_numpy(ora):
    err = _error_field()
    n_img, h, w = err.shape
    cxy, cy, ci = _cdfs(ora, err)
    f = np.float32
    # independent model: sequential float32 running sums (np.cumsum accumulates in the array's dtype, in order)
    row = np.cumsum(err + f(1e-10), axis=2, dtype=np.float32)
    tot = row[:, :, -1]
    norm = (f(1.0) / tot)[:, :, None]
    xs = (np.arange(1, w + 1, dtype=np.float32))[None, None, :]
    ref_xy = (f(1.0) - f(0.01)) * row * norm + f(0.01) * xs / f(w)
    assert np.array_equal(cxy, ref_xy.astype(np.float32))
    col = np.cumsum(tot, axis=1, dtype=np.float32)
    itot = col[:, -1]
    ys = np.arange(1, h + 1, dtype=np.float32)[None, :]
    ref_y = (f(1.0) - f(0.01)) * col * (f(1.0) / itot)[:, None] + f(0.01) * ys / f(h)
    assert np.array_equal(cy, ref_y.astype(np.float32))
    icum = np.cumsum(itot, dtype=np.float32)
    ref_i = (f(1.0) - f(0.1)) * icum * (f(1.0) / icum[-1]) + f(0.1) * np.arange(1, n_img + 1, dtype=np.float32) / f(n_img)
    assert np.array_equal(ci, ref_i.astype(np.float32))
    # what the sampler relies on: non-decreasing, ending at 1
    assert np.all(np.diff(cxy, axis=2) >= 0) and np.all(np.diff(cy, axis=1) >= 0) and np.all(np.diff(ci) >= 0)
    assert np.allclose(cxy[:, :, -1], 1.0, atol=2e-6) and np.allclose(cy[:, -1], 1.0, atol=2e-6) and abs(ci[-1] - 1.0) < 2e-6


def test_sample_cdf_2d_follows_the_error(ora):
    """Half of the draws stay uniform (UNIFORM_SAMPLING_FRACTION), the other half lands in cell (x, y) with probability pmf_x|y * pmf_y; the density
    reported for them is pmf * n_cells (a mean of 1 under the CDF's own measure)."""
    err = _error_field()
    n_img, h, w = err.shape
    cxy, cy, ci = _cdfs(ora, err)
    res = (C.c_int32 * 2)(w, h)
    rs = np.random.default_rng(5)
    n = 200_000
    img = 1
    smp = rs.uniform(size=(n, 2)).astype(np.float32)
    uv = np.zeros((n, 2), np.float32); pdf = np.full(n, -7.0, np.float32)
    for k in range(n):
        ora.ora_sample_cdf_2d(_f32p(smp[k]), img, res, _f32p(cxy), _f32p(cy), _f32p(uv[k]), _f32p(pdf[k:k + 1]))
    uni = smp[:, 0] < 0.5
    assert np.all(pdf[uni] == -7.0), "the uniform branch must leave *pdf untouched (the caller initialised it to 1)"
    assert np.allclose(uv[uni, 0], smp[uni, 0] / 0.5) and np.array_equal(uv[uni, 1], smp[uni, 1])
    assert np.all((uv >= 0) & (uv <= 1.0 + 1e-6))
    pm_y = np.diff(np.concatenate([[0], cy[img]])); pm_x = np.diff(np.concatenate([np.zeros((h, 1), np.float32), cxy[img]], axis=1), axis=1)
    pmf = pm_x * pm_y[:, None]
    assert abs(pmf.sum() - 1.0) < 1e-4
    cx = np.minimum((uv[~uni, 0] * w).astype(int), w - 1); cyi = np.minimum((uv[~uni, 1] * h).astype(int), h - 1)
    hist = np.zeros((h, w)); np.add.at(hist, (cyi, cx), 1.0)
    m = (~uni).sum()
    # cell frequencies against the multinomial's standard deviation (5 sigma + 2 counts)
    assert np.all(np.abs(hist - m * pmf) <= 5.0 * np.sqrt(m * pmf * (1 - pmf)) + 2.0)
    hot = pmf[3:6, 10:14].sum()
    assert hot > 0.5 and abs(hist[3:6, 10:14].sum() / m - hot) < 0.01
    assert np.allclose(pdf[~uni], pmf[cyi, cx] * (w * h), rtol=2e-4)


def test_image_idx_cdf_follows_the_error(ora):
    err = _error_field()
    n_img = err.shape[0]
    _, _, ci = _cdfs(ora, err)
    pmf = np.diff(np.concatenate([[0], ci]))
    n = 50_000
    cnt = np.zeros(n_img); pdfs = np.zeros(n_img)
    p = C.c_float()
    for i in range(n):
        k = ora.ora_image_idx_cdf(i, n_img, _f32p(ci), C.byref(p))
        cnt[k] += 1; pdfs[k] = p.value
    # an Owen-scrambled Sobol sequence over consecutive indices: far better than random, 1 % is generous
    assert np.all(np.abs(cnt / n - pmf) < 0.01), (cnt / n, pmf)
    assert np.allclose(pdfs, pmf * n_img, rtol=1e-5)
    assert pmf.min() >= 0.1 / n_img - 1e-6  # MIN_PMF: every image keeps at least 10 % of its uniform share


def _k1(ora, M, X, bf, n_img, n_rays, max_samples, rng):
    rc, nc = C.c_uint32(), C.c_uint32()
    ri = np.zeros(n_rays, np.uint32); rays = np.zeros((n_rays, 6), np.float32); ns = np.zeros((n_rays, 2), np.uint32); co = np.zeros((max_samples, 7), np.float32)
    ora.ora_k_generate_training_samples(n_rays, 0, n_rays, A.scene_aabb(1), max_samples, rng, C.byref(rc), C.byref(nc), ptr(ri), ptr(rays), ptr(ns), ptr(co),
                                        n_img, M, X, ptr(bf), 0, 1, C.c_float(0.0))
    return rc.value, nc.value, ri, rays, ns, co


def test_k1_and_k3_with_cdfs(ora):
    """K1 picks images / pixels through the CDFs, K3 re-derives the same pixel, divides the LOSS (not the gradient) by its density and splats the mean
    loss bilinearly: the map's total equals the sum of the rays' mean losses."""
    imgs, xforms, meta = make_small_dataset(5, 48)
    M, X = host_meta(imgs, xforms, meta)
    n_img = len(imgs)
    grid = np.zeros(128 ** 3, np.float32)
    ora.ora_k_mark_untrained_density_grid(128 ** 3, ptr(grid), n_img, M, X, 1)
    grid = np.where(grid >= 0, 0.05, grid).astype(np.float32)
    bf = np.zeros(128 ** 3, np.uint8)
    ora.ora_k_grid_to_bitfield(ptr(grid), 0, ptr(bf), C.c_float(0.01))
    rng = A.Pcg32(); ora.ora_pcg32_seed(C.byref(rng), C.c_uint64(1337), C.c_uint64(1))
    n_rays, max_samples, B = 1024, 1 << 20, 1 << 20
    err = _error_field(n_img, 20, 28)
    cxy, cy, ci = _cdfs(ora, err)
    cres = (C.c_int32 * 2)(28, 20)
    base = _k1(ora, M, X, bf, n_img, n_rays, max_samples, rng)
    emap = np.zeros((n_img, 16, 16), np.float32); eres = (C.c_int32 * 2)(16, 16)
    try:
        ora.ora_set_error_sampling(_f32p(cxy), _f32p(cy), _f32p(ci), cres, _f32p(emap), eres)
        rc, nc, ri, rays, ns, co = _k1(ora, M, X, bf, n_img, n_rays, max_samples, rng)
        assert rc > 500 and nc <= max_samples and not np.array_equal(rays[:rc], base[3][:rc]), "the CDFs must change the rays"
        # every ray's image follows the image CDF of its index
        p = C.c_float()
        for k in range(0, rc, 37):
            img = ora.ora_image_idx_cdf(int(ri[k]), n_img, _f32p(ci), C.byref(p))
            assert np.allclose(rays[k, :3], np.asarray(X[img].start, np.float32).reshape(4, 3)[3], atol=1e-6), "ray origin = camera position of the CDF's image"
        rs = np.random.default_rng(1)
        net = np.zeros((max_samples, 4), np.float16)
        net[:nc, :3] = rs.normal(0, 1.5, (nc, 3)); net[:nc, 3] = rs.normal(-1.0, 2.5, nc)
        bg = (C.c_float * 3)(0, 0, 0)

        def k3():
            ns2 = ns.copy(); cc = np.zeros((B, 7), np.float32); dl = np.zeros((B, 4), np.uint16); loss = np.zeros(n_rays, np.float32); cnt = C.c_uint32()
            ora.ora_k_compute_loss(n_rays, rc, A.scene_aabb(1), rng, B, C.c_float(128.0), bg, 0, 1, 0, n_img, M, ptr(net.view(np.uint16)), 4, C.byref(cnt), ptr(ri), ptr(rays),
                                   ptr(ns2), ptr(co), ptr(cc), ptr(dl), 4, A.LOSS_HUBER, ptr(loss), A.ACT_LOGISTIC, A.ACT_EXPONENTIAL, 1, C.c_float(0.01), C.c_float(0.1))
            return ns2, dl, loss, cnt.value
        ns_a, dl_a, loss_a, cnt_a = k3()
        assert emap.min() >= 0 and emap.sum() > 0
        assert abs(float(emap.sum(dtype=np.float64)) - float(loss_a.sum(dtype=np.float64)) * n_rays) <= 1e-4 * emap.sum()
        # image CDF only: every ray's pixel density is its image's pmf * n_images, the reported loss is the Huber loss divided by it -> multiplying back must give
        # a value the Huber(0.1)/5 loss can take (bounded by the loss at |difference| = 1 per channel), and rays of an over-sampled image report SMALLER losses
        ora.ora_set_error_sampling(None, None, _f32p(ci), cres, None, None)
        rc, nc, ri, rays, ns, co = _k1(ora, M, X, bf, n_img, n_rays, max_samples, rng)
        net[:nc, :3] = rs.normal(0, 1.5, (nc, 3)); net[:nc, 3] = rs.normal(-1.0, 2.5, nc)
        ns_c, dl_c, loss_c, cnt_c = k3()
        pmf = np.diff(np.concatenate([[0], ci]))
        pdf_of = np.array([pmf[ora.ora_image_idx_cdf(int(r), n_img, _f32p(ci), None)] * n_img for r in ri[:rc]], np.float32)
        undivided = loss_c[:rc] * n_rays * pdf_of
        huber_max = (1.0 - 0.05) / 5.0  # huber(alpha 0.1) at |d| = 1: (|d| - alpha / 2) / 5 (nerf_device.cuh loss_and_gradient)
        assert undivided.max() <= huber_max * (1 + 1e-4) and undivided.max() > 0.2 * huber_max
    finally:
        ora.ora_set_error_sampling(None, None, None, None, None, None)


def test_trainer_cycle_schedule(ora):
    """Resolution of the map, the x 1.5 update interval and the validity flag follow testbed_nerf.cu:2753-2759 / 2795-2855; with both switches on the next
    steps sample through the CDFs and still produce a finite loss."""
    imgs, xforms, meta = make_small_dataset(4, 32)
    M, X = host_meta(imgs, xforms, meta)
    cfg = A.base_model_config(1, log2_hashmap_size=14)
    B = 1 << 12
    opts = A.default_nerf_options(1, target_batch_size=B, sample_focal_plane_proportional_to_error=1, sample_image_proportional_to_error=1)
    om = OraModel(ora, cfg)
    t = C.c_void_p()
    assert ora.ora_nerf_create(om.h, C.byref(opts), A.scene_aabb(1), C.byref(t)) == 0
    ora.ora_nerf_set_dataset(t, len(imgs), M, X)
    ora.ora_nerf_density_grid.restype = C.POINTER(C.c_float)
    g = np.ctypeslib.as_array(ora.ora_nerf_density_grid(t), shape=(128 ** 3,))
    ora.ora_k_mark_untrained_density_grid(128 ** 3, ptr(g), len(imgs), M, X, 1)
    g[g >= 0] = 0.05
    ora.ora_nerf_update_mean_and_bitfield(t)
    ora.ora_nerf_set_rays_per_batch(t, 512)
    ora.ora_nerf_set_error_map_interval(t, 3)

    def state():
        em, cxy, cy, ci = (C.POINTER(C.c_float)() for _ in range(4))
        er, cr = (C.c_int32 * 2)(), (C.c_int32 * 2)()
        valid, nb, nsn = C.c_int(), C.c_uint32(), C.c_uint32()
        ora.ora_nerf_error_map(t, C.byref(em), er, C.byref(cxy), C.byref(cy), C.byref(ci), cr, C.byref(valid), C.byref(nb), C.byref(nsn))
        return em, tuple(er), cxy, cy, ci, tuple(cr), valid.value, nb.value, nsn.value
    rays = 512
    for step in range(1, 9):
        st = A.NerfStats(); ora.ora_nerf_get_stats(t, C.byref(st))
        rays_before = st.rays_per_batch if step > 1 else 512
        assert ora.ora_nerf_train_forward_backward(t) == 0, ora.ora_last_error()
        em, er, cxy, cy, ci, cr, valid, nb, nsn = state()
        if step in (1, 4):  # a cycle starts: resolution from interval * rays / images
            interval = 3 if step == 1 else 4
            r = int(np.sqrt(np.sqrt(np.float32(interval * rays_before // len(imgs)))) * np.float32(3.5))
            assert er == (min(r, 32), min(r, 32)), (step, er, r)
        assert ora.ora_nerf_train_finish(t) == 0
        em, er, cxy, cy, ci, cr, valid, nb, nsn = state()
        if step < 3:
            assert not valid and nb == 3 and nsn == step
        elif step == 3:
            assert valid and nb == 4 and nsn == 0 and cr == er
            cyv = np.ctypeslib.as_array(cy, shape=(len(imgs), cr[1]))
            assert np.allclose(cyv[:, -1], 1.0, atol=1e-5)
            assert np.ctypeslib.as_array(em, shape=(len(imgs) * er[0] * er[1],)).sum() > 0
        elif step == 7:
            assert valid and nb == 6 and nsn == 0
        st = A.NerfStats(); ora.ora_nerf_get_stats(t, C.byref(st))
        assert np.isfinite(st.loss) and st.measured_batch_size > 0
    ora.ora_nerf_destroy(t)
