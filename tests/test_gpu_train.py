"""GPU end-to-end: the on-device training loop (no host sync) trains the synthetic scene; loss decreases, counters behave
like NerfCounters, and the first steps track the CPU oracle's trainer (same seed, same dataset)."""
import ctypes as C

import numpy as np
import pytest

import ngp_abi as A
from common import HipModel, OraModel, half_to_f32, host_meta, make_small_dataset, ptr

pytestmark = pytest.mark.gpu


def _make(ora, hip, B, n_images=8, res=64, rays0=0):
    imgs, xforms, meta = make_small_dataset(n_images, res)
    M, X = host_meta(imgs, xforms, meta)
    cfg = A.base_model_config(1)
    opts = A.default_nerf_options(1, target_batch_size=B)
    aabb = A.scene_aabb(1)
    hm = HipModel(hip, cfg)
    t = C.c_void_p()
    A.check(hip, hip.ngp_nerf_create(hm.h, C.byref(opts), aabb, C.byref(t)))
    pix = (C.c_void_p * len(imgs))(*[im.ctypes.data for im in imgs])
    A.check(hip, hip.ngp_nerf_set_dataset_host(t, len(imgs), M, X, pix))
    om = OraModel(ora, cfg)
    ot = C.c_void_p()
    assert ora.ora_nerf_create(om.h, C.byref(opts), aabb, C.byref(ot)) == 0
    ora.ora_nerf_set_dataset(ot, len(imgs), M, X)
    if rays0:
        # start in a regime where neither K1's sample cap nor K3's batch clamp drops rays (both are order dependent)
        A.check(hip, hip.ngp_nerf_set_rays_per_batch(t, rays0)); ora.ora_nerf_set_rays_per_batch(ot, rays0)
    return dict(hm=hm, t=t, om=om, ot=ot, keep=(imgs, M, X, pix), opts=opts)


def _stats(hip, t):
    s = A.NerfStats()
    A.check(hip, hip.ngp_nerf_get_stats(t, None, C.byref(s)))
    return s


def test_training_loop_tracks_oracle(ora, hip):
    """Device trainer vs the oracle's trainer, step by step, in two regimes.
    (a) From initialisation, four optimizer steps: step counters, marched / compacted samples PER RAY, the batch-size controller's decisions and the loss.  At initialisation
        every occupancy cell sits AT the threshold (density = exp(~0) * min step everywhere, threshold = mean), so half-ulp differences of the density MLP flip occupancy
        bits, and the controller rounds its ray count up to a multiple of 256 -- a discrete decision next to a rounding boundary at these tiny ray counts: the two runs track
        each other only statistically (over 20 device runs step 4 marched 2.0e5 .. 2.7e5 samples against the oracle's 2.27e5, profiles/r02_tracking_test_spread.txt), hence
        the wide bounds of this part.
    (b) From a TRAINED state (64 device steps; parameters, Adam moments, step counters, EMA, occupancy grid, rng and ray count copied into the oracle): three more steps without
        an occupancy-grid update in between.  Nothing sits at a threshold any more, so the two trainers must agree per step to a few per cent: rays that hit (K1's lattice vs the
        reference recurrence: 0.3 % of rays), marched and compacted samples per ray, the controller's next ray count (one 256-step) and the loss.  Bounds = 2 x measured
        (profiles/r04_pytest_gpu.log)."""
    import torch
    B = 1 << 17
    s = _make(ora, hip, B, rays0=256)
    rays_h = rays_o = 256
    for step in range(1, 5):
        A.check(hip, hip.ngp_nerf_train(s["t"], None, 1))
        assert ora.ora_nerf_train(s["ot"], 1) == 0, ora.ora_last_error()
        hs = _stats(hip, s["t"]); os_ = A.NerfStats(); ora.ora_nerf_get_stats(s["ot"], C.byref(os_))
        assert hs.training_step == os_.training_step == step
        print(step, hs.measured_batch_size_before_compaction, os_.measured_batch_size_before_compaction, hs.measured_batch_size, os_.measured_batch_size,
              hs.rays_per_batch, os_.rays_per_batch, hs.loss, os_.loss)
        bh, bo = hs.measured_batch_size_before_compaction / rays_h, os_.measured_batch_size_before_compaction / rays_o
        assert abs(bh - bo) <= 0.15 * bo + 0.25, (step, bh, bo)
        if hs.measured_batch_size < 0.95 * B and os_.measured_batch_size < 0.95 * B:  # (K3 clamps the batch at B)
            ch, co = hs.measured_batch_size / rays_h, os_.measured_batch_size / rays_o
            assert abs(ch - co) <= 0.15 * co + 0.25, (step, ch, co)
        # the controller: next ray count = this one * B / compacted samples, rounded up to a multiple of 256; allow one rounding step
        assert abs(int(hs.rays_per_batch) - int(os_.rays_per_batch)) <= 0.10 * os_.rays_per_batch + 256
        # (which rays the sample cap drops differs: the device fills its ray slots in a scrambled order, the oracle in index order)
        assert abs(hs.loss - os_.loss) <= 0.3 * abs(os_.loss) + 1e-5, (step, hs.loss, os_.loss)
        rays_h, rays_o = hs.rays_per_batch, os_.rays_per_batch
    # ---- (b) from a trained state ----
    t, hm, om, ot = s["t"], s["hm"], s["om"], s["ot"]
    A.check(hip, hip.ngp_nerf_train(t, None, 60))           # step 64: the next three steps have no grid update pending between them (prep is skipped below on both sides)
    size = hip.ngp_model_serialized_size; size.restype = C.c_uint64
    nbytes = size(hm.h, 1)
    blob = np.zeros(nbytes, np.uint8)
    A.check(hip, hip.ngp_model_serialize_host(hm.h, ptr(blob), C.c_uint64(nbytes), 1))
    om.load_serialized(blob)
    gp = C.c_void_p(); hip.ngp_nerf_density_grid_ptrs(t, C.byref(gp), None, None)
    grid = np.empty(128 ** 3, np.float32); rt = C.CDLL("libamdhip64.so"); torch.cuda.synchronize()
    assert rt.hipMemcpy(ptr(grid), gp, C.c_size_t(grid.nbytes), 2) == 0
    C.memmove(ora.ora_nerf_density_grid(ot), grid.ctypes.data, grid.nbytes)
    ora.ora_nerf_update_mean_and_bitfield(ot)
    rng, grng = A.Pcg32(), A.Pcg32(); hip.ngp_nerf_get_rng(t, C.byref(rng), C.byref(grng))
    ora.ora_nerf_set_rng(ot, C.byref(rng))
    hs = _stats(hip, t)
    # 10 % fewer rays than the controller asks for, on both sides: neither K1's sample cap (the previous step's marched count) nor K3's batch clamp binds -- both drop
    # rays in an order that differs by design (scrambled slots vs index order)
    rays = int(hs.rays_per_batch * 0.9) // 256 * 256
    A.check(hip, hip.ngp_nerf_set_rays_per_batch(t, rays))
    ora.ora_nerf_set_rays_per_batch(ot, rays); ora.ora_nerf_set_training_step(ot, hs.training_step)
    ora.ora_nerf_set_measured(ot, hs.measured_batch_size_before_compaction, hs.measured_batch_size)
    for step in range(3):
        A.check(hip, hip.ngp_nerf_train_forward_backward(t, None)); A.check(hip, hip.ngp_nerf_train_finish(t, None))
        assert ora.ora_nerf_train_forward_backward(ot) == 0 and ora.ora_nerf_train_finish(ot) == 0, ora.ora_last_error()
        hs = _stats(hip, t); os_ = A.NerfStats(); ora.ora_nerf_get_stats(ot, C.byref(os_))
        d = dict(hit=(hs.n_rays_last, os_.n_rays_last), marched=(hs.measured_batch_size_before_compaction, os_.measured_batch_size_before_compaction),
                 compacted=(hs.measured_batch_size, os_.measured_batch_size), next_rays=(hs.rays_per_batch, os_.rays_per_batch), loss=(hs.loss, os_.loss))
        print("trained state, step", step, "rays", rays, {k: (v, round(abs(v[0] - v[1]) / max(abs(v[1]), 1e-12), 5)) for k, v in d.items()})
        assert hs.training_step == os_.training_step
        tol = TRACK_TOL if step == 0 else {k: 2 * v for k, v in TRACK_TOL.items()}   # later steps inherit the (tiny) parameter differences of the earlier ones
        for k in ("hit", "marched", "compacted", "loss"):
            assert abs(d[k][0] - d[k][1]) <= tol[k] * abs(d[k][1]), (step, k, d[k])
        assert abs(int(hs.rays_per_batch) - int(os_.rays_per_batch)) <= 256
        # the next step starts from the same ray count on both sides, 5 % fewer than this one: K1's sample cap is the PREVIOUS step's marched count (testbed_nerf.cu:3055-3060),
        # so a step that marches more than its predecessor drops rays -- in slot order on the device, in index order in the oracle (measured: 2622 vs 2508 rays kept)
        rays = int(rays * 0.95) // 256 * 256
        A.check(hip, hip.ngp_nerf_set_rays_per_batch(t, rays)); ora.ora_nerf_set_rays_per_batch(ot, rays)
    hip.ngp_nerf_destroy(s["t"]); ora.ora_nerf_destroy(s["ot"])


# relative bounds of part (b), first step; = 2 x measured in round 4 (see the docstring)
TRACK_TOL = dict(hit=0.005, marched=0.005, compacted=0.01, loss=0.02)   # measured at the first step: 0, 0, 0, 3e-5; at the second: marched 2e-5, compacted 4e-3, loss 1.2e-2


def test_training_converges(hip, ora):
    B = 1 << 16
    s = _make(ora, hip, B, n_images=12, res=96)
    A.check(hip, hip.ngp_nerf_train(s["t"], None, 16))
    l0 = _stats(hip, s["t"]).loss
    A.check(hip, hip.ngp_nerf_train(s["t"], None, 300))
    st = _stats(hip, s["t"])
    assert st.training_step == 316
    assert np.isfinite(st.loss) and st.loss < 0.5 * l0, (l0, st.loss)
    assert 0 < st.measured_batch_size <= B * 1.5
    # the controller drops to ~1k rays while the untrained grid is dense (>= 64 samples per ray) and recovers as the grid gets sparser;
    # the equilibrium on this scene is ~15-16 compacted samples per ray, i.e. rays_per_batch hovers around 4096 = B / 16 (3840 .. 4608
    # depending on the order of the fp16 atomics), so the bound must not sit exactly on it
    assert st.rays_per_batch % 256 == 0 and st.rays_per_batch >= 3072
    hip.ngp_nerf_destroy(s["t"]); ora.ora_nerf_destroy(s["ot"])


def test_lazy_k2_matches_eager(ora, hip):
    """Front-to-back (lazy) K2 evaluates only the samples in front of each ray's transmittance cut; the eager order evaluates
    every marched sample like the reference. From the same trained state, one forward/backward pass must compact exactly the
    same samples and produce the same gradients (up to the order of the fp16 atomics on the dense levels)."""
    import torch
    B = 1 << 18
    s = _make(ora, hip, B, n_images=12, res=96)
    A.check(hip, hip.ngp_nerf_train(s["t"], None, 400))
    st = _stats(hip, s["t"])
    print("network evaluations", st.network_evaluations, "of", st.measured_batch_size_before_compaction, "marched samples")
    assert st.network_evaluations < st.measured_batch_size_before_compaction
    hm = s["hm"]
    size = hip.ngp_model_serialized_size
    size.restype = C.c_uint64
    nbytes = size(hm.h, 0)
    buf = (C.c_uint8 * nbytes)()
    A.check(hip, hip.ngp_model_serialize_host(hm.h, buf, C.c_uint64(nbytes), 0))
    g, bf, mean = C.c_void_p(), C.c_void_p(), C.c_void_p()
    hip.ngp_nerf_density_grid_ptrs(s["t"], C.byref(g), C.byref(bf), C.byref(mean))
    n_cells = 128 ** 3
    grid = np.empty(n_cells, dtype=np.float32)
    torch.cuda.synchronize()
    rt = C.CDLL("libamdhip64.so")
    assert rt.hipMemcpy(ptr(grid), g, C.c_size_t(n_cells * 4), 2) == 0
    rng, grng = A.Pcg32(), A.Pcg32()
    hip.ngp_nerf_get_rng(s["t"], C.byref(rng), C.byref(grng))
    rays = min(st.rays_per_batch, 12000)  # far below K1's sample cap and K3's batch clamp (both order dependent)
    K3_TWO_PASS = 1048576
    res = {}
    for name, flags in (("lazy", 0), ("lazy_rounds", 0), ("lazy_tile16", 0), ("lazy_tile32", 0), ("lazy_tile8", 0), ("eager", 8192)):
        c = _make(ora, hip, B, n_images=12, res=96)
        if name == "lazy_rounds":  # list-driven rounds instead of the default single launch whose wavefronts follow their rays
            A.check(hip, hip.ngp_nerf_set_k2_params(c["t"], 3, 32))
        if name == "lazy_tile16":  # single launch, two 16-sample tiles of different rays per wavefront
            A.check(hip, hip.ngp_nerf_set_k2_params(c["t"], 1, 16))
        if name == "lazy_tile32":
            A.check(hip, hip.ngp_nerf_set_k2_params(c["t"], 1, 32))
        if name == "lazy_tile8":
            A.check(hip, hip.ngp_nerf_set_k2_params(c["t"], 1, 8))
        A.check(hip, hip.ngp_model_deserialize_host(c["hm"].h, buf, C.c_uint64(nbytes)))
        A.check(hip, hip.ngp_nerf_set_density_grid_host(c["t"], None, ptr(grid), C.c_uint64(n_cells)))
        hip.ngp_nerf_set_rng(c["t"], C.byref(rng))
        A.check(hip, hip.ngp_nerf_set_rays_per_batch(c["t"], rays))
        # Every variant runs K3 as the deterministic two-pass kernel (DBG_K3_TWO_PASS: slot-ordered compaction): fill_rollover duplicates the FIRST B - n compacted samples (here
        # ~15 % of the batch), and with the production K3 which ones come first depends on the order of its span atomics -- two runs of the SAME path then differ by a few per
        # cent, occasionally by 10 % (the last tier run of round 6 drew 0.1008 against the 0.1 bar below).  With the slot order the variants see the same batch rows.
        hip.ngp_debug_set_flags(flags | K3_TWO_PASS)
        try:
            A.check(hip, hip.ngp_nerf_train_forward_backward(c["t"], None))
        finally:
            hip.ngp_debug_set_flags(0)
        cp = C.POINTER(C.c_uint32)()
        hip.ngp_nerf_counter_ptrs(c["t"], C.byref(cp))
        cnt = np.empty(2, dtype=np.uint32)
        torch.cuda.synchronize()
        assert rt.hipMemcpy(ptr(cnt), cp, C.c_size_t(8), 2) == 0
        grads = half_to_f32(c["hm"].read("grads", torch))
        A.check(hip, hip.ngp_nerf_train_finish(c["t"], None))
        res[name] = (cnt.copy(), grads, _stats(hip, c["t"]).loss)
        hip.ngp_nerf_destroy(c["t"]); ora.ora_nerf_destroy(c["ot"])
    for variant in ("lazy_rounds", "lazy_tile16", "lazy_tile32", "lazy_tile8"):
        (cr, gr, lr), (ce, ge, le) = res[variant], res["eager"]
        assert abs(lr - le) <= 1e-4 * abs(le) and cr[0] == ce[0] and cr[1] == ce[1], variant
        assert np.linalg.norm(gr - ge) / np.linalg.norm(ge) < 0.1, variant
    (cl, gl, ll), (ce, ge, le) = res["lazy"], res["eager"]
    print("loss", ll, le)
    assert abs(ll - le) <= 1e-4 * abs(le)  # the composited colours (forward pass) are the same up to the summation order
    print("marched, compacted:", cl, ce)
    assert cl[0] == ce[0] and cl[1] == ce[1] and 0 < cl[1] < B
    assert cl[1] < 0.7 * cl[0]  # the cut is active in this state
    err = np.linalg.norm(gl - ge) / np.linalg.norm(ge)
    print("relative gradient difference", err)
    assert err < 0.1  # (measured with the slot-ordered compaction: see the printed value)
    hip.ngp_nerf_destroy(s["t"]); ora.ora_nerf_destroy(s["ot"])


def test_prelaunched_k1_is_the_same_k1(ora, hip):
    """K1 of step n+1 is launched on a side stream behind step n's controller (it does not depend on the parameters). Its output
    must be bit-identical to a K1 launched the normal way from the same state."""
    import torch
    B = 1 << 14
    s = _make(ora, hip, B, n_images=8, res=64)
    t = s["t"]
    A.check(hip, hip.ngp_nerf_train(t, None, 70))  # step 71 has no grid update pending (prep every 4th step) -> K1(71) is in flight
    torch.cuda.synchronize()
    ri, rays, ns, co, mo, cc, dl, cn = (C.c_void_p() for _ in range(8))
    A.check(hip, hip.ngp_nerf_scratch_ptrs(t, C.byref(ri), C.byref(rays), C.byref(ns), C.byref(co), C.byref(mo), C.byref(cc), C.byref(dl), C.byref(cn)))
    rt = C.CDLL("libamdhip64.so")
    n_rays_cap, n_samples = 1 << 18, B * 16

    def grab():
        torch.cuda.synchronize()
        a = np.empty(n_rays_cap, np.uint32); b = np.empty(n_rays_cap * 6, np.float32); c = np.empty(n_samples * 7, np.float32)
        assert rt.hipMemcpy(ptr(a), ri, C.c_size_t(a.nbytes), 2) == 0
        assert rt.hipMemcpy(ptr(b), rays, C.c_size_t(b.nbytes), 2) == 0
        assert rt.hipMemcpy(ptr(c), co, C.c_size_t(c.nbytes), 2) == 0
        return a, b, c
    pre = grab()
    rng, grng = A.Pcg32(), A.Pcg32()
    hip.ngp_nerf_get_rng(t, C.byref(rng), C.byref(grng))
    hip.ngp_nerf_set_rng(t, C.byref(rng))  # same state, but the pre-launched K1 is now considered stale -> K1 runs again in-line
    A.check(hip, hip.ngp_nerf_train_prep(t, None))
    hip.ngp_debug_set_flags(4096)  # DBG_NO_STREAM_OVERLAP: this step must not pre-launch K1(72) over the buffers we compare
    try:
        A.check(hip, hip.ngp_nerf_train_forward_backward(t, None))
    finally:
        hip.ngp_debug_set_flags(0)
    post = grab()
    for name, x, y in zip(("ray_indices", "rays", "coords"), pre, post):
        bad = np.flatnonzero(x.view(np.uint32) != y.view(np.uint32))
        assert bad.size == 0, (name, bad.size, bad[:8].tolist(), x[bad[:4]].tolist(), y[bad[:4]].tolist())
    A.check(hip, hip.ngp_nerf_train_finish(t, None))
    st = _stats(hip, t)
    assert st.training_step == 71 and st.measured_batch_size > 0
    hip.ngp_nerf_destroy(t); ora.ora_nerf_destroy(s["ot"])


def _fox_small():
    """tests/golden/fox_small through the host loader: 13 views, OpenCV lens, aabb_scale 4 (three occupancy cascades, exponential stepping)."""
    import os
    import pyngp as ngp
    tb = ngp.Testbed()
    tb.load_training_data(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fox_small", "transforms.json"))
    d = tb.nerf.training.dataset
    imgs = [np.ascontiguousarray(d.image(i)) for i in range(d.n_images)]
    n = d.n_images
    M = (A.ImageMeta * n)(); X = (A.Xform * n)()
    for i in range(n):
        m = d.metadata[i]
        M[i].pixels = imgs[i].ctypes.data; M[i].image_data_type = A.IMAGE_BYTE; M[i].lens_mode = m.lens_mode
        M[i].resolution[0], M[i].resolution[1] = m.resolution
        M[i].principal_point[0], M[i].principal_point[1] = m.principal_point
        M[i].focal_length[0], M[i].focal_length[1] = m.focal_length
        for k in range(7):
            M[i].lens_params[k] = m.lens_params[k]
        for k in range(12):
            X[i].start[k] = X[i].end[k] = d.xforms[i][k]
    return imgs, M, X


@pytest.mark.parametrize("scene", ["synthetic_aabb1", "fox_small_aabb4"])
def test_render_matches_oracle(ora, hip, scene):
    """ngp_nerf_render (lattice march + rounds of batched inference + compositing + shade) vs the oracle's per-pixel renderer, whose colour AND
    depth are pinned bit for bit to the reference's fused renderer (fused_kernels/render_nerf.cuh:146-183 compiled here, tests/test_ref_render.py;
    unfused: testbed_nerf.cu:579-689, 1333-1378).  200 x 200 pixels, three views, on the single-cascade synthetic scene and on the
    multi-cascade real capture (fox_small: aabb_scale 4, exponential stepping, mip_from_dt on every step).  Both outputs are compared:
      rgba  (premultiplied linear): half network outputs, otherwise the same arithmetic -> bounds = 2 x the measured values below;
      depth (camera-space depth of the max-weight sample where alpha > 0.2, MAX_DEPTH elsewhere): an argmax over samples, so half-ulp noise may
             pick the neighbouring sample (one step) on a few pixels, and alpha within 1e-3 of 0.2 may classify differently.
    Measured (round 4, MI355X): the comment at RENDER_TOL below; printed lines in profiles/r04_pytest_gpu.log."""
    import torch
    from common import dptr
    B = 1 << 16
    multi = scene != "synthetic_aabb1"
    if multi:
        imgs, M, X = _fox_small()
        cfg = A.base_model_config(4); opts = A.default_nerf_options(4, target_batch_size=B); aabb = A.scene_aabb(4)
        hm = HipModel(hip, cfg); t = C.c_void_p()
        A.check(hip, hip.ngp_nerf_create(hm.h, C.byref(opts), aabb, C.byref(t)))
        pix = (C.c_void_p * len(imgs))(*[im.ctypes.data for im in imgs])
        A.check(hip, hip.ngp_nerf_set_dataset_host(t, len(imgs), M, X, pix))
        om = OraModel(ora, cfg); ot = C.c_void_p()
        assert ora.ora_nerf_create(om.h, C.byref(opts), aabb, C.byref(ot)) == 0
        ora.ora_nerf_set_dataset(ot, len(imgs), M, X)
        s = dict(hm=hm, t=t, om=om, ot=ot, keep=(imgs, M, X, pix), opts=opts)
        n_steps, n_casc, views = 600, 3, (0, 5, 9)
    else:
        s = _make(ora, hip, B, n_images=8, res=64)
        M, X = s["keep"][1], s["keep"][2]; aabb = A.scene_aabb(1)
        n_steps, n_casc, views = 200, 1, (3, 0, 6)
    A.check(hip, hip.ngp_nerf_train(s["t"], None, n_steps))
    # copy the trained state into the oracle trainer
    p = np.empty(s["om"].n, np.float32)
    A.check(hip, hip.ngp_model_get_params_host(s["hm"].h, ptr(p), C.c_uint64(p.size)))
    s["om"].params_fp[:] = p; ora.ora_model_sync_half(s["om"].h)
    gp, bp = C.c_void_p(), C.c_void_p(); hip.ngp_nerf_density_grid_ptrs(s["t"], C.byref(gp), C.byref(bp), None)
    n_el = 128 ** 3 * n_casc
    grid = np.empty(n_el, np.float32); rt = C.CDLL("libamdhip64.so"); torch.cuda.synchronize()
    assert rt.hipMemcpy(ptr(grid), gp, C.c_size_t(grid.nbytes), 2) == 0
    C.memmove(ora.ora_nerf_density_grid(s["ot"]), grid.ctypes.data, grid.nbytes)
    ora.ora_nerf_update_mean_and_bitfield(s["ot"])
    bf_d = np.empty(128 ** 3, np.uint8); assert rt.hipMemcpy(ptr(bf_d), bp, C.c_size_t(bf_d.nbytes), 2) == 0
    bf_o = np.ctypeslib.as_array(C.cast(ora.ora_nerf_bitfield(s["ot"]), C.POINTER(C.c_uint8)), shape=(128 ** 3,))
    assert (bf_d != bf_o).sum() <= 8  # same grid, the mean (threshold) differs by summation order only
    # the device renders with the oracle's bitfield so that the few threshold cells cannot differ (the grid chain has its own bit-exact test)
    assert rt.hipMemcpy(bp, ptr(np.ascontiguousarray(bf_o)), C.c_size_t(bf_o.nbytes), 1) == 0
    res = 200
    MAX_DEPTH = 16384.0
    for vi, view in enumerate(views):
        rp = A.RenderParams()
        rp.resolution[0] = rp.resolution[1] = res
        # the training view's field of view along x, square pixels, on a square frame
        rp.focal_length[0] = rp.focal_length[1] = M[view].focal_length[0] * res / M[view].resolution[0]
        rp.screen_center[0] = rp.screen_center[1] = 0.5
        for k in range(12):
            rp.camera[k] = X[view].start[k]
        rp.lens_mode = 0; rp.spp_index = vi; rp.snap_to_pixel_centers = 1 if vi != 1 else 0; rp.min_transmittance = 1e-4
        rp.near_distance = 0.0 if not multi else 0.1; rp.use_inference_params = 0
        rp.render_aabb = aabb
        f_o = np.zeros((res * res, 4), np.float32); d_o = np.zeros(res * res, np.float32)
        assert ora.ora_nerf_render(s["ot"], C.byref(rp), ptr(f_o), ptr(d_o)) == 0
        f_d = torch.zeros((res * res, 4), dtype=torch.float32, device="cuda"); d_d = torch.zeros(res * res, dtype=torch.float32, device="cuda")
        A.check(hip, hip.ngp_nerf_render(s["t"], None, C.byref(rp), dptr(f_d), dptr(d_d)))
        torch.cuda.synchronize()
        f = f_d.cpu().numpy(); d = d_d.cpu().numpy()
        err = np.abs(f - f_o)
        hit_o, hit_d = d_o < MAX_DEPTH, d < MAX_DEPTH
        both = hit_o & hit_d
        derr = np.abs(d[both] - d_o[both])
        cover = float((f_o[:, 3] > 0.5).mean())
        print(f"render {scene} view {view}: coverage {cover:.3f} rgba max err {err.max():.2e} 99th pct {np.quantile(err, 0.99):.2e} | depth: hit pixels {int(both.sum())} "
              f"classification mismatches {int((hit_o != hit_d).sum())} max err {derr.max():.2e} 99th pct {np.quantile(derr, 0.99):.2e} exact-ish (<1e-4) {float((derr < 1e-4).mean()):.4f}")
        assert cover > 0.05 and both.mean() > 0.05  # something is visible
        tol = RENDER_TOL[scene]
        assert np.quantile(err, 0.99) <= tol["rgba_q99"] and (err.max(axis=1) > 2e-3).mean() <= tol["rgba_outliers"] and err.max() <= tol["rgba_max"]
        assert (hit_o != hit_d).mean() <= 0.001                      # alpha next to 0.2 (measured: 0 pixels)
        assert np.isfinite(d).all() and (d[~hit_d] == MAX_DEPTH).all()
        assert (derr < 1e-4).mean() >= tol["depth_same_sample"]      # the same max-weight sample
        assert np.quantile(derr, 0.99) <= tol["depth_q99"]
    hip.ngp_nerf_destroy(s["t"]); ora.ora_nerf_destroy(s["ot"])


# Bounds = 2-3 x what round 4 measured on MI355X (profiles/r04_pytest_gpu.log), six frames of 40,000 pixels:
#   rgba  99th percentile 2.6e-5 .. 5.2e-5; a FEW pixels per frame (< 0.05 %) move by up to 2.6e-2: the device marches the lattice in closed form, the oracle accumulates
#         t += dt, so a point that sits on a voxel face can test the neighbouring cell (tests/test_k1_lattice_model.py classifies these) and one sample appears / disappears;
#   depth the same max-weight sample (|delta| < 1e-4) on 99.90 - 99.92 % of the hit pixels, 99th percentile 8e-7 .. 2e-6; the rest picked another sample of nearly equal
#         weight (any distance along the ray); alpha-0.2 classification: 0 mismatches.
RENDER_TOL = {
    # (rgba_max: 2 x the 2.6e-2 above.  It stood at 2e-2 -- BELOW the measured maximum -- until the last GPU tier of round 6 drew a 2.11e-2 pixel: the maximum over 40,000 pixels of a
    # freshly trained model is one voxel-face sample appearing or not; what the test holds tight are the 99th percentile and the fraction of pixels off by more than 2e-3.)
    "synthetic_aabb1": dict(rgba_q99=1e-4, rgba_outliers=1e-3, rgba_max=5.2e-2, depth_same_sample=0.997, depth_q99=1e-4),
    "fox_small_aabb4": dict(rgba_q99=1.5e-4, rgba_outliers=1e-3, rgba_max=8e-2, depth_same_sample=0.997, depth_q99=1e-4),
}


@pytest.mark.parametrize("train_mode", [1, 2])
def test_rfl_train_modes_converge(hip, ora, train_mode):
    """ngp_nerf_options::train_mode = Rfl / RflRelax (fused_kernels/train_nerf.cuh:391-410, evaluated by the unfused K3): the run.py schedule
    (Nerf first, then the Rfl mode, run.py:229-242) trains, and the loss stays in the range of the Nerf-mode run."""
    B = 1 << 16
    res = {}
    for mode in (0, train_mode):
        s = _make(ora, hip, B, n_images=12, res=96)
        A.check(hip, hip.ngp_nerf_train(s["t"], None, 150))  # warm-up in Nerf mode, like run.py's --rfl_warmup_steps
        o = s["opts"]; o.train_mode = mode
        A.check(hip, hip.ngp_nerf_set_options(s["t"], C.byref(o)))
        A.check(hip, hip.ngp_nerf_train(s["t"], None, 250))
        st = _stats(hip, s["t"])
        res[mode] = st.loss
        assert st.training_step == 400 and np.isfinite(st.loss) and st.measured_batch_size > 0
        hip.ngp_nerf_destroy(s["t"]); ora.ora_nerf_destroy(s["ot"])
    print(f"train_mode {train_mode}: loss {res[train_mode]:.5f} vs Nerf mode {res[0]:.5f}")
    assert res[train_mode] < 4 * res[0] + 1e-4


def test_depth_supervision_end_to_end(hip, ora):
    """The production path of the depth term (k1_setup stores the ray's target depth next to its target colour, k_compute_loss_v2 reads it): training with
    depth images that claim a much smaller depth than the scene's pulls the composited depth sum_j w_j depth_j (not normalised by the opacity, like the
    reference's depth_ray) down -- the model gives up opacity to get there; with depth_supervision_lambda = 0 the same images change nothing.  The arithmetic itself is checked per ray in test_gpu_nerf.py::test_k3_depth_supervision."""
    import torch
    from common import dptr
    B, n_img, res = 1 << 16, 12, 96
    imgs, xforms, meta = make_small_dataset(n_img, res)
    means = {}
    for tag, lam, depth_value in (("off", 0.0, 0.35), ("on", 2.0, 0.35)):
        M, X = host_meta(imgs, xforms, meta)
        dep = None
        if depth_value is not None:
            dep = np.full(res * res, depth_value, np.float32)
            for i in range(n_img):
                M[i].depth = dep.ctypes.data
        hm = HipModel(hip, A.base_model_config(1))
        opts = A.default_nerf_options(1, target_batch_size=B, depth_supervision_lambda=lam, depth_loss_type=A.LOSS_L1)
        t = C.c_void_p()
        A.check(hip, hip.ngp_nerf_create(hm.h, C.byref(opts), A.scene_aabb(1), C.byref(t)))
        pix = (C.c_void_p * n_img)(*[im.ctypes.data for im in imgs])
        A.check(hip, hip.ngp_nerf_set_dataset_host(t, n_img, M, X, pix))
        A.check(hip, hip.ngp_nerf_train(t, None, 400))
        rp = A.RenderParams()
        rp.resolution[0] = rp.resolution[1] = 48
        rp.focal_length[0] = rp.focal_length[1] = M[0].focal_length[0] * 48 / M[0].resolution[0]
        rp.screen_center[0] = rp.screen_center[1] = 0.5
        for k in range(12):
            rp.camera[k] = X[2].start[k]
        rp.lens_mode = 0; rp.spp_index = 0; rp.snap_to_pixel_centers = 1; rp.min_transmittance = 1e-4; rp.near_distance = 0.0; rp.use_inference_params = 0
        rp.render_aabb = A.scene_aabb(1)
        f = torch.zeros((48 * 48, 4), dtype=torch.float32, device="cuda"); dd = torch.zeros(48 * 48, dtype=torch.float32, device="cuda")
        A.check(hip, hip.ngp_nerf_render(t, None, C.byref(rp), dptr(f), dptr(dd)))
        torch.cuda.synchronize()
        st = _stats(hip, t)
        alpha = f[:, 3].cpu().numpy(); depth = dd.cpu().numpy()
        means[tag] = (float(alpha.mean()), float((alpha > 0.5).mean()), float(st.loss))
        assert np.isfinite(st.loss) and st.measured_batch_size > 0
        hip.ngp_nerf_destroy(t)
    print("mean rendered opacity / coverage / loss:", means)
    assert means["off"][1] > 0.1 and means["off"][0] > 0.1  # lambda = 0: the scene trains as if the depth images were not there
    assert means["on"][1] < 0.5 * means["off"][1], means      # the (wrong, far too small) depth targets are met by giving up solid surfaces: no pixel stays opaque


def _error_state(hip, t):
    em, cxy, cy, ci = (C.c_void_p() for _ in range(4))
    er, cr = (C.c_int32 * 2)(), (C.c_int32 * 2)()
    valid, nb, nsn = C.c_int(), C.c_uint32(), C.c_uint32()
    A.check(hip, hip.ngp_nerf_error_map_ptrs(t, C.byref(em), er, C.byref(cxy), C.byref(cy), C.byref(ci), cr, C.byref(valid), C.byref(nb), C.byref(nsn)))
    return dict(em=em, er=tuple(er), cxy=cxy, cy=cy, ci=ci, cr=tuple(cr), valid=valid.value, n_between=nb.value, n_since=nsn.value)


def _dl(ptr_, n, dtype=np.float32):
    rt = C.CDLL("libamdhip64.so")
    a = np.empty(n, dtype)
    assert rt.hipMemcpy(ptr(a), ptr_, C.c_size_t(a.nbytes), 2) == 0
    return a


def test_error_map_cycle_and_sampling(ora, hip):
    """The trainer's error-map cycle (testbed_nerf.cu:2753-2759, 2791-2855) with both sampling switches on: map resolution from interval x rays / images,
    CDFs valid after `interval` steps, interval x 1.5; the CDFs the library built are the oracle's construction of the library's own map (bit-exact); the
    step after the update marches the rays the oracle's K1 derives from those CDFs (a pre-launched K1 of the old CDFs must have been discarded); training
    keeps converging.  Without the switches nothing is allocated."""
    import torch
    B = 1 << 15
    s = _make(ora, hip, B, n_images=8, res=64)
    t, opts = s["t"], s["opts"]
    n_img = 8
    A.check(hip, hip.ngp_nerf_train(t, None, 3))
    st = _error_state(hip, t)
    assert not st["em"].value and not st["valid"] and st["n_since"] == 0, "default: no error map"
    opts.sample_focal_plane_proportional_to_error = 1; opts.sample_image_proportional_to_error = 1
    A.check(hip, hip.ngp_nerf_set_options(t, C.byref(opts)))
    A.check(hip, hip.ngp_nerf_set_error_map_interval(t, 6))
    rays0 = _stats(hip, t).rays_per_batch
    A.check(hip, hip.ngp_nerf_train(t, None, 1))
    st = _error_state(hip, t)
    r = int(np.sqrt(np.sqrt(np.float32(6 * rays0 // n_img))) * np.float32(3.5))
    assert st["er"] == (min(r, 64), min(r, 64)) and st["em"].value and not st["valid"] and st["n_since"] == 1, (st, r, rays0)
    A.check(hip, hip.ngp_nerf_train(t, None, 4))
    torch.cuda.synchronize()
    st = _error_state(hip, t)
    assert not st["valid"] and st["n_since"] == 5
    em_before = _dl(st["em"], n_img * st["er"][0] * st["er"][1])
    assert em_before.min() >= 0 and em_before.sum() > 0
    with pytest.raises(RuntimeError):
        A.check(hip, hip.ngp_nerf_set_error_map_interval(t, 9))  # only between cycles
    A.check(hip, hip.ngp_nerf_train(t, None, 1))
    torch.cuda.synchronize()
    st = _error_state(hip, t)
    assert st["valid"] and st["n_between"] == 9 and st["n_since"] == 0 and st["cr"] == st["er"]
    w, h = st["cr"]
    em = _dl(st["em"], n_img * w * h).reshape(n_img, h, w)
    cxy, cy, ci = _dl(st["cxy"], n_img * w * h).reshape(n_img, h, w), _dl(st["cy"], n_img * h).reshape(n_img, h), _dl(st["ci"], n_img)
    oxy = np.zeros_like(em); oy = np.zeros((n_img, h), np.float32); oi = np.zeros(n_img, np.float32)
    f = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    ora.ora_construct_error_cdfs(n_img, w, h, f(em), f(oxy), f(oy), f(oi))
    assert np.array_equal(cxy, oxy) and np.array_equal(cy, oy) and np.array_equal(ci, oi)
    assert em.sum() >= em_before.sum()
    # the next forward pass: its rays are the oracle K1's rays under these CDFs
    rng, grng = A.Pcg32(), A.Pcg32(); hip.ngp_nerf_get_rng(t, C.byref(rng), C.byref(grng))
    R = _stats(hip, t).rays_per_batch
    A.check(hip, hip.ngp_nerf_train_prep(t, None))
    hip.ngp_debug_set_flags(4096)  # DBG_NO_STREAM_OVERLAP: no K1 of the step after next over the buffers read below
    try:
        A.check(hip, hip.ngp_nerf_train_forward_backward(t, None))
    finally:
        hip.ngp_debug_set_flags(0)
    torch.cuda.synchronize()
    ri_p, rays_p, ns_p, co_p, mo_p, cc_p, dl_p, cn_p = (C.c_void_p() for _ in range(8))
    A.check(hip, hip.ngp_nerf_scratch_ptrs(t, C.byref(ri_p), C.byref(rays_p), C.byref(ns_p), C.byref(co_p), C.byref(mo_p), C.byref(cc_p), C.byref(dl_p), C.byref(cn_p)))
    n_act = _stats(hip, t).n_rays_last  # (the controller has run behind K4)
    imgs, M, X, pix = s["keep"]
    gp, bp = C.c_void_p(), C.c_void_p(); A.check(hip, hip.ngp_nerf_density_grid_ptrs(t, C.byref(gp), C.byref(bp), None))
    bf = _dl(bp, 128 ** 3 // 8 * 8, np.uint8)
    ora.ora_set_error_sampling(f(cxy), f(cy), f(ci), (C.c_int32 * 2)(w, h), None, None)
    try:
        rc, nc = C.c_uint32(), C.c_uint32()
        o_ri = np.zeros(R, np.uint32); o_rays = np.zeros((R, 6), np.float32); o_ns = np.zeros((R, 2), np.uint32); o_co = np.zeros((B * 16, 7), np.float32)
        ora.ora_k_generate_training_samples(R, 0, R, A.scene_aabb(1), B * 16, rng, C.byref(rc), C.byref(nc), ptr(o_ri), ptr(o_rays), ptr(o_ns), ptr(o_co),
                                            n_img, M, X, ptr(bf), 0, 1, C.c_float(0.0))
    finally:
        ora.ora_set_error_sampling(None, None, None, None, None, None)
    assert 0 < n_act <= R and abs(n_act - rc.value) <= 0.01 * rc.value + 2, (n_act, rc.value)
    ri = _dl(ri_p, n_act, np.uint32); rays = _dl(rays_p, n_act * 6).reshape(-1, 6)
    ref = {int(r): i for i, r in enumerate(o_ri[:rc.value])}
    hit = [i for i in range(n_act) if int(ri[i]) in ref]
    assert len(hit) >= 0.99 * n_act
    for i in hit:
        assert np.array_equal(rays[i].view(np.uint32), o_rays[ref[int(ri[i])]].view(np.uint32)), "K1 did not use the new CDFs"
    A.check(hip, hip.ngp_nerf_train_finish(t, None))
    # keeps training: loss after 150 more steps well below the loss now
    l0 = _stats(hip, t).loss
    A.check(hip, hip.ngp_nerf_train(t, None, 150))
    s1 = _stats(hip, t)
    assert np.isfinite(s1.loss) and s1.measured_batch_size > 0.5 * B and s1.loss < 0.7 * l0, (l0, s1.loss)
    st = _error_state(hip, t)
    assert st["valid"] and st["n_between"] > 9
    hip.ngp_nerf_destroy(t); ora.ora_nerf_destroy(s["ot"])


def test_t1_reuses_k2_encodings_bit_exact(hip, ora):
    """Production T1 reads the encodings the lazy K2 left behind for the same samples (64 contiguous bytes per sample through K3's row -> sample map) instead of
    gathering them from the hash tables again.  Same batch, both ways (flag 536870912 = DBG_T1_NO_K2_STASH): every level's gradient goes through the exact
    fixed-point lists and the weight gradients are summed in batch order, so the two gradient vectors must be IDENTICAL -- also when K4 has padded the batch
    with wrap-around copies (early steps: few compacted samples)."""
    import torch
    B = 1 << 15
    s = _make(ora, hip, B, n_images=8, res=64)
    t, hm = s["t"], s["hm"]
    n, n_mlp = C.c_uint64(), C.c_uint64()
    hip.ngp_model_n_params(hm.h, C.byref(n), C.byref(n_mlp))
    g = C.c_void_p(); hip.ngp_model_param_ptrs(hm.h, None, None, None, C.byref(g))
    for steps_before in (2, 60):  # step 3: the batch is mostly padding; step ~63: full batches
        A.check(hip, hip.ngp_nerf_train(t, None, steps_before))
        A.check(hip, hip.ngp_nerf_train_prep(t, None))
        try:
            hip.ngp_debug_set_flags(4096)  # DBG_NO_STREAM_OVERLAP: no K1 of the next step in flight while the step is replayed
            A.check(hip, hip.ngp_nerf_train_forward(t, None))
            A.check(hip, hip.ngp_nerf_train_backward(t, None))
            torch.cuda.synchronize()
            g1 = _dl(g, n.value, np.uint16)
            hip.ngp_debug_set_flags(4096 | 536870912)
            A.check(hip, hip.ngp_nerf_train_backward(t, None))
            torch.cuda.synchronize()
            g2 = _dl(g, n.value, np.uint16)
            # round 5: the production backward pass is ONE kernel (k_train_fused = T1 + W: role A of the weight-gradient kernel also emits dL/d(enc)); flags2 = 1
            # (DBG2_NO_FUSED_T1W) replays the same batch through the round-4 pair k_train_fwd_bwd<STASH> + k_wgrad2: same operands, same order of every sum
            hip.ngp_debug_set_flags(4096); hip.ngp_debug_set_flags2(1)
            A.check(hip, hip.ngp_nerf_train_backward(t, None))
            torch.cuda.synchronize()
            g3 = _dl(g, n.value, np.uint16)
        finally:
            hip.ngp_debug_set_flags(0); hip.ngp_debug_set_flags2(0)
        st = _stats(hip, t)
        assert np.count_nonzero(g1) > 1000
        bad = np.flatnonzero(g1 != g2)
        assert bad.size == 0, (steps_before, st.measured_batch_size, bad.size, bad[:8].tolist(), g1[bad[:4]].tolist(), g2[bad[:4]].tolist())
        bad = np.flatnonzero(g1 != g3)
        assert bad.size == 0, ("fused T1 + W vs the two kernels", steps_before, bad.size, bad[:8].tolist(), g1[bad[:4]].tolist(), g3[bad[:4]].tolist())
        A.check(hip, hip.ngp_nerf_train_finish(t, None))
    s1 = _stats(hip, t)
    assert np.isfinite(s1.loss) and s1.measured_batch_size > 0
    hip.ngp_nerf_destroy(t); ora.ora_nerf_destroy(s["ot"])


def test_error_map_survives_a_growing_dataset(ora, hip):
    """The error map and the three CDFs are laid out per image.  When the image count changes under error-proportional sampling (pyngp raises
    n_images_for_training / re-uploads a streamed dataset), an open cycle and installed CDFs are dropped and the next cycle is sized for the new count
    (the reference sizes them from dataset.n_images, testbed_nerf.cu:2753-2759, so it never sees the problem)."""
    import torch
    B = 1 << 15
    s = _make(ora, hip, B, n_images=4, res=64)
    t, opts = s["t"], s["opts"]
    opts.sample_focal_plane_proportional_to_error = 1; opts.sample_image_proportional_to_error = 1
    A.check(hip, hip.ngp_nerf_set_options(t, C.byref(opts)))
    A.check(hip, hip.ngp_nerf_set_error_map_interval(t, 4))
    A.check(hip, hip.ngp_nerf_train(t, None, 6))   # one full cycle (CDFs valid for 4 images) + two steps into the next
    st = _error_state(hip, t)
    assert st["valid"] and st["n_since"] == 2
    imgs, xforms, meta = make_small_dataset(12, 64)
    M, X = host_meta(imgs, xforms, meta)
    pix = (C.c_void_p * len(imgs))(*[im.ctypes.data for im in imgs])
    A.check(hip, hip.ngp_nerf_set_dataset_host(t, len(imgs), M, X, pix))
    st = _error_state(hip, t)
    assert not st["valid"] and st["n_since"] == 0, "stale per-image CDFs must not survive a dataset of another size"
    A.check(hip, hip.ngp_nerf_train(t, None, 1))
    st = _error_state(hip, t)
    assert st["n_since"] == 1 and not st["valid"]
    n_between = st["n_between"]
    A.check(hip, hip.ngp_nerf_train(t, None, n_between - 1))
    torch.cuda.synchronize()
    st = _error_state(hip, t)
    assert st["valid"] and st["n_since"] == 0
    w, h = st["cr"]
    ci = _dl(st["ci"], 12)
    assert np.all(np.diff(ci) >= 0) and abs(ci[-1] - 1.0) < 1e-5 and ci[3] < 0.999, "the image CDF covers all 12 images"
    em = _dl(st["em"], 12 * w * h).reshape(12, -1)
    assert (em.sum(axis=1) > 0).sum() >= 10
    A.check(hip, hip.ngp_nerf_train(t, None, 20))
    assert np.isfinite(_stats(hip, t).loss)
    hip.ngp_nerf_destroy(t); ora.ora_nerf_destroy(s["ot"])


def test_fused_optimizer_epilogue_is_the_separate_sweep(ora, hip):
    """ngp_nerf_train on one GPU applies the optimizer to the hashed levels inside k_grad_accumulate (the gradient sums are in LDS: no gradient write, no second pass
    over those parameters) and sweeps only the MLP + dense levels; the split calls (forward_backward, then finish) write gradients and sweep everything.  Same arithmetic
    (the sums are rounded to half as the stored gradient would be), so from the same seed both trainers must hold IDENTICAL master parameters, Adam moments, per-parameter
    step counters and EMA weights after 40 steps -- with the deterministic (slot-ordered) K3 compaction, because the production K3 orders the batch rows by atomic arrival
    and the weight gradients are summed in row order.  40 steps cover occupancy-grid updates and the batch-size controller's ramp."""
    B = 1 << 16
    flags = 1048576  # DBG_K3_TWO_PASS
    blobs, losses = [], []
    for fused in (True, False):
        hip.ngp_debug_set_flags(flags)
        try:
            s = _make(ora, hip, B, n_images=8, res=64)
            t = s["t"]
            if fused:
                A.check(hip, hip.ngp_nerf_train(t, None, 40))
            else:
                for _ in range(40):
                    A.check(hip, hip.ngp_nerf_train_prep(t, None))
                    A.check(hip, hip.ngp_nerf_train_forward_backward(t, None))
                    A.check(hip, hip.ngp_nerf_train_finish(t, None))
            st = _stats(hip, t)
            size = hip.ngp_model_serialized_size; size.restype = C.c_uint64
            nbytes = size(s["hm"].h, 1)
            buf = np.zeros(nbytes, np.uint8)
            A.check(hip, hip.ngp_model_serialize_host(s["hm"].h, ptr(buf), C.c_uint64(nbytes), 1))
            inf = np.zeros(s["om"].n, np.uint16); par = np.zeros(s["om"].n, np.uint16)
            pi, pp = C.c_void_p(), C.c_void_p(); hip.ngp_model_param_ptrs(s["hm"].h, None, C.byref(pp), C.byref(pi), None)
            import torch
            torch.cuda.synchronize()
            assert C.CDLL("libamdhip64.so").hipMemcpy(ptr(inf), pi, C.c_size_t(inf.nbytes), 2) == 0 and C.CDLL("libamdhip64.so").hipMemcpy(ptr(par), pp, C.c_size_t(par.nbytes), 2) == 0
            blobs.append((buf, inf, par)); losses.append((st.training_step, st.loss, st.rays_per_batch, st.measured_batch_size))
            hip.ngp_nerf_destroy(t); ora.ora_nerf_destroy(s["ot"])
        finally:
            hip.ngp_debug_set_flags(0)
    print("fused vs split:", losses)
    # (the loss scalar is a float atomic sum over rays: equal up to its order)
    assert losses[0][0] == losses[1][0] == 40 and losses[0][2:] == losses[1][2:] and abs(losses[0][1] - losses[1][1]) <= 1e-5 * losses[1][1]
    n = (blobs[0][0].size - 32) // 4 // 5
    hdr = 32
    for k, name in enumerate(("master", "adam_m", "adam_v", "adam_steps", "ema")):
        a = blobs[0][0][hdr + k * n * 4: hdr + (k + 1) * n * 4].view(np.uint32); b = blobs[1][0][hdr + k * n * 4: hdr + (k + 1) * n * 4].view(np.uint32)
        bad = np.flatnonzero(a != b)
        assert bad.size == 0, (name, bad.size, bad[:8].tolist())
    for k, name in ((1, "inference (EMA) half parameters"), (2, "half parameters")):
        bad = np.flatnonzero(blobs[0][k] != blobs[1][k])
        assert bad.size == 0, (name, bad.size, bad[:8].tolist(), blobs[0][k][bad[:8]].tolist(), blobs[1][k][bad[:8]].tolist())
    assert np.array_equal(blobs[0][0][:hdr], blobs[1][0][:hdr])
