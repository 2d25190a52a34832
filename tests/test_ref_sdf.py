"""CPU: the SDF primitive's ground truth against THE REFERENCE'S OWN TRIANGLE BVH.

oracle/_ref/libngpbvh_ref.so is src/triangle_bvh.cu (with triangle.cuh, bounding_box.cuh, discrete_distribution.h) compiled for the CPU from /root/reference where it lies
(oracle/Makefile, oracle/ref_bvh_wrapper.cpp) against oracle/ref_shim; its "kernels" run as loops.  The oracle (oracle/ora_sdf.hpp) restates the same ground truth WITHOUT an
acceleration structure, and the HIP path's own BVH is checked against that oracle on the GPU (tests/test_sdf.py).  Here: the reference's BVH, built and traversed by the
reference's code, must give the oracle's numbers -- unsigned distances bit for bit (a traversal only prunes), signs of the 32-ray stab test on every query point that is not
within rounding of a silhouette.  Second part (below): the batch generators of the image and SDF primitives against the reference's own kernels of testbed_image.cu /
testbed_sdf.cu (oracle/_ref/libngpimgsdf_ref.so).  Skipped when oracle/_ref is not built."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_ref", "libngpbvh_ref.so")
F = C.c_float


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _bits(x):
    return np.asarray(x, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(SO):
        pytest.skip("oracle/_ref/libngpbvh_ref.so not built (needs /root/reference; `make -C oracle ref`)")
    lib = C.CDLL(SO)
    lib.ref_bvh_create.restype = C.c_void_p
    for n in ("tri_distance_sq", "tri_ray_intersect", "tri_surface_area", "box_distance_sq", "box_signed_distance"):
        getattr(lib, "ref_" + n).restype = F
    lib.ref_discrete_distribution_sample.restype = C.c_uint32
    return lib


@pytest.fixture(scope="module")
def hip_lib():
    import ngp_abi
    return ngp_abi.load_hip()  # host-side hook only: no GPU needed


@pytest.fixture(scope="module")
def o(ora):
    ora.ora_tri_distance_sq.restype = F; ora.ora_tri_ray_intersect.restype = F; ora.ora_cdf_search.restype = C.c_uint32
    return ora


def _sphere_mesh(rs, levels=3, bumps=0.25):
    """closed, star-shaped mesh: a subdivided octahedron with a smooth radial displacement (inside = the stab test's "every ray hits")"""
    v = [np.array(p, np.float64) for p in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1))]
    f = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
    for _ in range(levels):
        cache, nf = {}, []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]; v.append(m / np.linalg.norm(m)); cache[k] = len(v) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (ab, b, bc), (ca, bc, c), (ab, bc, ca)]
        f = nf
    v = np.array(v)
    k = rs.normal(size=(4, 3))
    r = 0.3 * (1.0 + bumps * np.sin(v @ k.T * 2.0).mean(axis=1))
    v = v * r[:, None] + 0.5
    return np.ascontiguousarray(v[np.array(f)].reshape(-1, 9).astype(np.float32))


def _torus_mesh(nu=24, nv=12, R=0.28, r=0.1):
    u, w = np.meshgrid(np.arange(nu) * 2 * np.pi / nu, np.arange(nv) * 2 * np.pi / nv, indexing="ij")
    p = np.stack([(R + r * np.cos(w)) * np.cos(u), (R + r * np.cos(w)) * np.sin(u), r * np.sin(w)], -1) + 0.5
    t = []
    for i in range(nu):
        for j in range(nv):
            a, b, c, d = p[i, j], p[(i + 1) % nu, j], p[(i + 1) % nu, (j + 1) % nv], p[i, (j + 1) % nv]
            t += [np.concatenate([a, b, c]), np.concatenate([a, c, d])]
    return np.ascontiguousarray(np.array(t, np.float32))


def test_triangle_primitives(ref, o):
    """Triangle::distance_sq / ray_intersect (triangle.cuh:87-129) and fibonacci_dir<32> (random_val.cuh), bit for bit -- query points near the plane, the edges and the
    corners included, rays through edges included"""
    rs = np.random.default_rng(0)
    for trial in range(400):
        t = rs.uniform(0, 1, 9).astype(np.float32) if trial % 4 else (rs.uniform(0.4, 0.6, 9)).astype(np.float32)
        a, b, c = t[0:3], t[3:6], t[6:9]
        for k in range(30):
            w = rs.dirichlet((1, 1, 1)); on = (w[0] * a + w[1] * b + w[2] * c).astype(np.float32)
            p = [rs.uniform(-0.5, 1.5, 3), on, on + rs.normal(0, 1e-4, 3), a + rs.normal(0, 1e-3, 3), 0.5 * (a + b) + rs.normal(0, 1e-3, 3), 2 * a - b][k % 6].astype(np.float32)
            assert _bits(o.ora_tri_distance_sq(_fp(t), _fp(p))) == _bits(ref.ref_tri_distance_sq(_fp(t), _fp(p))), (t.tolist(), p.tolist())
            ro = rs.uniform(-0.5, 1.5, 3).astype(np.float32)
            target = [on, 0.5 * (a + b), a, rs.uniform(0, 1, 3)][k % 4]
            rd = (target - ro).astype(np.float32); rd /= np.linalg.norm(rd)
            assert _bits(o.ora_tri_ray_intersect(_fp(t), _fp(ro), _fp(rd))) == _bits(ref.ref_tri_ray_intersect(_fp(t), _fp(ro), _fp(rd))), (t.tolist(), ro.tolist(), rd.tolist())
    for i in range(32):
        for _ in range(20):
            off = rs.uniform(0, 1, 2).astype(np.float32); da, db = np.zeros(3, np.float32), np.zeros(3, np.float32)
            o.ora_fibonacci_dir32(i, _fp(off), _fp(da)); ref.ref_fibonacci_dir32(i, _fp(off), _fp(db))
            assert np.array_equal(_bits(da), _bits(db)), (i, off.tolist(), da.tolist(), db.tolist())


def test_surface_distribution(ref, o):
    """DiscreteDistribution::build / sample over the surface areas (discrete_distribution.h:21-42) against the restatement the product's mesh setup uses
    (ngp_api.hip ngp_sdf_create: sequential float sums, last entry forced to 1) and the oracle's cdf_search"""
    rs = np.random.default_rng(1)
    for n in (1, 2, 7, 1000, 20000):
        w = rs.uniform(1e-6, 1, n).astype(np.float32)
        cdf = np.zeros(n, np.float32); ref.ref_discrete_distribution_build(_fp(w), n, _fp(cdf))
        total = np.float32(0)
        for x in w:
            total = np.float32(total + x)
        inv = np.float32(1) / total
        acc, mine = np.float32(0), np.zeros(n, np.float32)
        for i, x in enumerate(w):
            acc = np.float32(acc + np.float32(x * inv)); mine[i] = acc
        mine[-1] = 1.0
        assert np.array_equal(_bits(cdf), _bits(mine))
        for val in np.concatenate([rs.uniform(0, 1, 300), [0.0, 1.0 - 2 ** -24], cdf[:50]]).astype(np.float32):
            assert o.ora_cdf_search(F(float(val)), _fp(cdf), n) == ref.ref_discrete_distribution_sample(_fp(cdf), n, F(float(val)))


@pytest.mark.parametrize("mesh", ["sphere", "torus"])
def test_bvh_signed_distance_equals_brute_force(ref, o, mesh):
    """TriangleBvh4::build + signed_distance_raystab through signed_distance_raystab_kernel (triangle_bvh.cu:631-650, 759-840, 893-909; per-element rng.advance(i * 2)) against
    the oracle's brute force over the same triangles: |distance| bit for bit on every point, sign on every point (the meshes are closed; points are drawn in the box, near the
    surface and on it), with and without the training batch's distance upper bounds; unsigned_distance_kernel likewise"""
    rs = np.random.default_rng(5)
    tris = _sphere_mesh(rs) if mesh == "sphere" else _torus_mesh()
    n_tris = len(tris)
    h = C.c_void_p(ref.ref_bvh_create(_fp(tris), n_tris, 8))
    try:
        ordered = np.zeros_like(tris); ref.ref_bvh_triangles(h, _fp(ordered))
        assert sorted(map(bytes, ordered)) == sorted(map(bytes, tris))  # build() only permutes
        n = 6000
        pick = ordered[rs.integers(0, n_tris, n)]
        w = rs.dirichlet((1, 1, 1), n).astype(np.float32)
        surf = (w[:, :1] * pick[:, 0:3] + w[:, 1:2] * pick[:, 3:6] + w[:, 2:3] * pick[:, 6:9]).astype(np.float32)
        third = n // 3
        pos = np.ascontiguousarray(np.concatenate([rs.uniform(0, 1, (third, 3)), surf[:third] + rs.logistic(0, 0.01, (third, 3)), surf[third: 2 * third]]).astype(np.float32))
        n = len(pos)
        mine = np.zeros(n, np.float32); o.ora_sdf_signed_distance(_fp(ordered), n_tris, _fp(pos), n, None, _fp(mine))
        theirs = np.zeros(n, np.float32); ref.ref_bvh_signed_distance(h, 1, _fp(pos), n, _fp(theirs), 0)
        assert np.array_equal(_bits(np.abs(mine)), _bits(np.abs(theirs)))
        off_surface = np.abs(mine) > 1e-6
        assert np.array_equal(np.signbit(mine[off_surface]), np.signbit(theirs[off_surface])), int((np.signbit(mine) != np.signbit(theirs)).sum())
        assert 0.05 < (mine < 0).mean() < 0.6  # both signs are exercised
        uns = np.zeros(n, np.float32); ref.ref_bvh_unsigned_distance(h, _fp(pos), n, _fp(uns), 0)
        assert np.array_equal(_bits(uns), _bits(np.abs(mine)))
        # upper bounds, as the training batch passes them (testbed_sdf.cu:1532-1540): a bound above the true distance changes nothing; below it, "no triangle found" -> 0
        bound = (np.abs(mine) * rs.choice([1.001, 2.0, 0.5], n)).astype(np.float32)
        mine_b = np.zeros(n, np.float32); o.ora_sdf_signed_distance(_fp(ordered), n_tris, _fp(pos), n, _fp(bound), _fp(mine_b))
        theirs_b = bound.copy(); ref.ref_bvh_signed_distance(h, 1, _fp(pos), n, _fp(theirs_b), 1)
        assert np.array_equal(_bits(np.abs(mine_b)), _bits(np.abs(theirs_b)))
        assert (mine_b[bound < np.abs(mine)] == 0).all()
    finally:
        ref.ref_bvh_destroy(h)


def test_bvh_signed_distance_armadillo(ref, o, hip_lib):
    """the reference's shipped mesh (data/sdf/armadillo.obj, 99,976 triangles; staged under _ref_data/ by tools/stage_reference_data.py), normalised like load_mesh does:
    the reference BVH's signed distances at 1,500 points against brute force over all triangles, and against the product's BVH evaluated on the host"""
    path = os.path.join(ROOT, "_ref_data", "data", "sdf", "armadillo.obj")
    if not os.path.exists(path):
        pytest.skip("_ref_data/data/sdf/armadillo.obj not staged")
    v, f = [], []
    for line in open(path):
        if line.startswith("v "):
            v.append([float(x) for x in line.split()[1:4]])
        elif line.startswith("f "):
            f.append([int(x.split("/")[0]) - 1 for x in line.split()[1:4]])
    v = np.array(v, np.float64); f = np.array(f)
    lo, hi = v.min(0), v.max(0)
    v = (v - 0.5 * (lo + hi)) / (hi - lo).max() * 0.75 + 0.5  # inside the unit cube (the exact normalisation is pinned elsewhere: any placement serves this test)
    tris = np.ascontiguousarray(v[f].reshape(-1, 9).astype(np.float32)); n_tris = len(tris)
    h = C.c_void_p(ref.ref_bvh_create(_fp(tris), n_tris, 8))
    try:
        ordered = np.zeros_like(tris); ref.ref_bvh_triangles(h, _fp(ordered))
        rs = np.random.default_rng(9); n = 1500
        pick = ordered[rs.integers(0, n_tris, n)]
        cent = ((pick[:, 0:3] + pick[:, 3:6] + pick[:, 6:9]) / 3).astype(np.float32)
        pos = np.ascontiguousarray(np.where((np.arange(n) % 3 == 0)[:, None], rs.uniform(0.1, 0.9, (n, 3)), cent + rs.logistic(0, 0.01, (n, 3))).astype(np.float32))
        mine = np.zeros(n, np.float32); o.ora_sdf_signed_distance(_fp(ordered), n_tris, _fp(pos), n, None, _fp(mine))
        theirs = np.zeros(n, np.float32); ref.ref_bvh_signed_distance(h, 1, _fp(pos), n, _fp(theirs), 0)
        assert np.array_equal(_bits(np.abs(mine)), _bits(np.abs(theirs)))
        # a scanned mesh has slivers: a stab ray through an edge can be a hit for one summation order and a miss for the box test that prunes it
        assert (np.signbit(mine) != np.signbit(theirs)).mean() <= 0.002, int((np.signbit(mine) != np.signbit(theirs)).sum())
        assert 0.05 < (mine < 0).mean() < 0.7
        prod = np.zeros(n, np.float32); assert hip_lib.ngp_host_sdf_signed_distance(_fp(tris), n_tris, _fp(pos), n, _fp(prod), 0, None, None) == 0
        assert np.array_equal(_bits(np.abs(prod)), _bits(np.abs(theirs)))
        assert (np.signbit(prod) != np.signbit(theirs)).mean() <= 0.002, int((np.signbit(prod) != np.signbit(theirs)).sum())
    finally:
        ref.ref_bvh_destroy(h)


@pytest.mark.parametrize("mesh", ["sphere", "torus"])
def test_product_bvh_on_the_host_equals_the_reference_bvh(ref, o, hip_lib, mesh):
    """The product's mesh setup and ground truth -- ngp_sdf_create's BVH build and surface CDF, csrc/sdf_kernels.hip's traversal, compiled for the host from the same
    source (ngp_host_sdf_signed_distance, no GPU) -- against the reference's BVH on the same query points: |distance| bit for bit, signs (host libm's sincos, so a stab
    ray grazing an edge may differ from the reference's by a rounding: <= 0.2 %), upper bounds, and the surface CDF over the product's triangle order"""
    rs = np.random.default_rng(11)
    tris = _sphere_mesh(rs, levels=4) if mesh == "sphere" else _torus_mesh(48, 24)
    n_tris = len(tris)
    n = 4000
    pick = tris[rs.integers(0, n_tris, n)]
    cent = ((pick[:, 0:3] + pick[:, 3:6] + pick[:, 6:9]) / 3).astype(np.float32)
    pos = np.ascontiguousarray(np.where((np.arange(n) % 3 == 0)[:, None], rs.uniform(0, 1, (n, 3)), cent + rs.logistic(0, 0.01, (n, 3))).astype(np.float32))
    mine = np.zeros(n, np.float32); ordered = np.zeros_like(tris); cdf = np.zeros(n_tris, np.float32)
    assert hip_lib.ngp_host_sdf_signed_distance(_fp(tris), n_tris, _fp(pos), n, _fp(mine), 0, _fp(ordered), _fp(cdf)) == 0
    assert sorted(map(bytes, ordered)) == sorted(map(bytes, tris))
    h = C.c_void_p(ref.ref_bvh_create(_fp(tris), n_tris, 8))
    try:
        theirs = np.zeros(n, np.float32); ref.ref_bvh_signed_distance(h, 1, _fp(pos), n, _fp(theirs), 0)
        assert np.array_equal(_bits(np.abs(mine)), _bits(np.abs(theirs)))
        off = np.abs(mine) > 1e-6
        assert (np.signbit(mine[off]) != np.signbit(theirs[off])).mean() <= 0.002
        assert 0.05 < (mine < 0).mean() < 0.7
        bound = (np.abs(theirs) * rs.choice([1.001, 2.0, 0.5], n)).astype(np.float32)
        mine_b = bound.copy(); assert hip_lib.ngp_host_sdf_signed_distance(_fp(tris), n_tris, _fp(pos), n, _fp(mine_b), 1, None, None) == 0
        theirs_b = bound.copy(); ref.ref_bvh_signed_distance(h, 1, _fp(pos), n, _fp(theirs_b), 1)
        assert np.array_equal(_bits(np.abs(mine_b)), _bits(np.abs(theirs_b)))
    finally:
        ref.ref_bvh_destroy(h)
    # the surface CDF of the product's order == DiscreteDistribution::build over Triangle::surface_area of those triangles
    area = np.array([ref.ref_tri_surface_area(_fp(t)) for t in ordered], np.float32)
    cdf_ref = np.zeros(n_tris, np.float32); ref.ref_discrete_distribution_build(_fp(area), n_tris, _fp(cdf_ref))
    assert np.array_equal(_bits(cdf), _bits(cdf_ref))


# ---- the sample generators of the image and SDF primitives against the reference's own kernels (oracle/_ref/libngpimgsdf_ref.so) --------------------------------------
IMGSDF_SO = os.path.join(ROOT, "oracle", "_ref", "libngpimgsdf_ref.so")


@pytest.fixture(scope="module")
def refk():
    if not os.path.exists(IMGSDF_SO):
        pytest.skip("oracle/_ref/libngpimgsdf_ref.so not built (needs /root/reference; `make -C oracle ref`)")
    return C.CDLL(IMGSDF_SO)


def _seeded(ora, seed=1337, advance=0):
    import ngp_abi as A
    s = A.Pcg32(); ora.ora_pcg32_seed(C.byref(s), C.c_uint64(seed), C.c_uint64(1))
    if advance:
        ora.ora_pcg32_advance(C.byref(s), C.c_int64(advance))
    return s


@pytest.mark.parametrize("n,stratified,snap,linear", [(4096, 1, 0, 0), (4096, 0, 1, 1), (3000, 1, 1, 0), (1024, 0, 0, 1)])
def test_image_batch_against_the_reference_kernels(refk, o, n, stratified, snap, linear):
    """train_image's batch (testbed_image.cu:66-82 stratify2_kernel, :176-229 eval_image_kernel_and_snap<float, 3>) from the same uniform numbers: positions (stratified, snapped
    to pixel centres) and targets (nearest texel when snapping, bilinear otherwise, sRGB or linear), bit for bit against the oracle's image_generate_batch, which is what the
    HIP trainer's batches are compared with on the GPU.  tcnn's generate_random_uniform is not part of this: both sides start from the oracle's uniform numbers."""
    rs = np.random.default_rng(0); w, h = 37, 23
    img = np.ascontiguousarray(rs.uniform(0, 1.2, (h, w, 4)).astype(np.float32))
    raw = np.zeros((n, 2), np.float32); t0 = np.zeros((n, 3), np.float32)
    o.ora_image_generate_batch(_fp(img), w, h, n, _seeded(o), 0, 0, 1, _fp(raw), _fp(t0))  # no stratification, no snapping: the positions are the uniform numbers
    assert 0.0 <= raw.min() and raw.max() < 1.0
    pos_o = np.zeros((n, 2), np.float32); tgt_o = np.zeros((n, 3), np.float32)
    o.ora_image_generate_batch(_fp(img), w, h, n, _seeded(o), stratified, snap, linear, _fp(pos_o), _fp(tgt_o))
    pos_r = raw.copy(); tgt_r = np.zeros((n, 3), np.float32)
    refk.ref_image_generate_batch_from_uniforms(_fp(img), w, h, n, stratified, snap, linear, _fp(pos_r), _fp(tgt_r))
    assert np.array_equal(_bits(pos_o), _bits(pos_r)) and np.array_equal(_bits(tgt_o), _bits(tgt_r))
    assert stratified == 0 or n != 4096 or not np.array_equal(pos_o, raw)


def test_image_mse_kernels(refk):
    """compute_image_mse's kernels (testbed_image.cu:459-488) against the formula csrc/image_kernels.hip k_image_mse and its GPU test use: per-pixel dot(diff, diff) / 3, with
    the prediction optionally rounded to bytes by (int)(p * 255 + 0.5) clamped to [0, 255]"""
    rs = np.random.default_rng(1); n, w, h = 5000, 61, 40
    pos = np.zeros((w * h, 2), np.float32); refk.ref_image_coords_from_idx(w * h, 0, w, h, _fp(pos))
    xs, ys = np.meshgrid(np.arange(w), np.arange(h))
    mine = np.stack([(xs.reshape(-1).astype(np.float32) + np.float32(0.5)) / np.float32(w), (ys.reshape(-1).astype(np.float32) + np.float32(0.5)) / np.float32(h)], 1).astype(np.float32)
    assert np.array_equal(_bits(pos), _bits(mine))
    tgt = rs.uniform(0, 1, (n, 3)).astype(np.float32); pred = rs.uniform(-0.1, 1.1, (n, 3)).astype(np.float32)
    for q in (0, 1):
        out = np.zeros(n, np.float32); refk.ref_image_mse(n, _fp(tgt), _fp(pred), _fp(out), q)
        p = pred if not q else (np.clip((pred * np.float32(255.0) + np.float32(0.5)).astype(np.int32), 0, 255).astype(np.float32) / np.float32(255.0))
        d = tgt - p
        want = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]) / np.float32(3.0)
        assert np.array_equal(_bits(out), _bits(want.astype(np.float32)))


def test_sdf_training_positions_against_the_reference_kernels(refk, o):
    """generate_training_samples_sdf (testbed_sdf.cu:1449-1544) from the random numbers on: sample_uniform_on_triangle_kernel (surface CDF -> triangle -> uniform point),
    scale_to_aabb_kernel + assign_float for the uniform eighth, perturb_sdf_samples for the offset points -- positions and distance upper bounds bit for bit against the
    oracle's sdf_generate_positions.  The random numbers themselves (tcnn's generate_random_uniform / _logistic) are the oracle's on both sides; the signed distances that
    follow are pinned by the BVH tests above."""
    import ngp_abi as A
    rs = np.random.default_rng(2)
    tris = _torus_mesh(32, 16); n_tris = len(tris)
    area = np.array([0.5 * np.linalg.norm(np.cross(t[3:6] - t[0:3], t[6:9] - t[0:3])) for t in tris.astype(np.float64)], np.float32)
    cdf = (np.cumsum(area.astype(np.float64)) / area.astype(np.float64).sum()).astype(np.float32); cdf[-1] = 1.0
    n = 4096; n_exact, n_surface = n // 8 * 4, n // 8 * 7
    stddev = float(np.float32(np.sqrt(0.75)) / np.float32(1024.0))
    box = A.Aabb(); unit = A.Aabb()
    for k in range(3):
        box.min[k] = [0.05, 0.1, 0.2][k]; box.max[k] = [0.95, 0.9, 0.8][k]; unit.min[k] = 0.0; unit.max[k] = 1.0
    # the uniform numbers: with no surface samples and the unit box, positions = 0 + u * 1 = u
    u = np.zeros((n, 3), np.float32); dummy = np.zeros(n, np.float32)
    o.ora_sdf_generate_positions(_fp(tris), n_tris, _fp(cdf), n, 0, 0, _seeded(o), F(stddev), unit, _fp(u), _fp(dummy))
    # the perturbation stream continues 3 n draws further on (the oracle's layout of generate_random_logistic after generate_random_uniform)
    n_off = n_surface - n_exact
    u2 = np.zeros((n_off, 3), np.float32); d2 = np.zeros(n_off, np.float32)
    o.ora_sdf_generate_positions(_fp(tris), n_tris, _fp(cdf), n_off, 0, 0, _seeded(o, advance=3 * n), F(stddev), unit, _fp(u2), _fp(d2))
    o.ora_logistic_from_uniform.restype = F
    pert = np.array([[o.ora_logistic_from_uniform(F(float(x)), F(stddev)) for x in row] for row in u2], np.float32)
    pos_o = np.zeros((n, 3), np.float32); dist_o = np.zeros(n, np.float32)
    o.ora_sdf_generate_positions(_fp(tris), n_tris, _fp(cdf), n, n_exact, n_surface, _seeded(o), F(stddev), box, _fp(pos_o), _fp(dist_o))
    pos_r = u.copy(); dist_r = np.full(n, -1.0, np.float32)
    refk.ref_sdf_generate_positions_from_randoms(_fp(tris), n_tris, _fp(cdf), n, n_exact, n_surface, _fp(pert), box, _fp(pos_r), _fp(dist_r))
    assert np.array_equal(_bits(pos_o), _bits(pos_r)) and np.array_equal(_bits(dist_o), _bits(dist_r))
    assert (dist_o[:n_exact] == 0).all() and (dist_o[n_exact:n_surface] > 0).all() and len(set(dist_o[n_surface:].tolist())) == 1


def test_sdf_iou_counters(refk):
    """compare_signs_kernel without an octree (testbed_sdf.cu:540-569): the eight counters calculate_iou reads, against the definition csrc/sdf_kernels.hip k_sdf_compare_signs
    implements (inside = distance <= 0; counters = ref inside / outside, model inside / outside, intersection, union, outside-octree = 0, inside-octree = n)"""
    rs = np.random.default_rng(3); n = 20000
    pos = rs.uniform(0, 1, (n, 3)).astype(np.float32)
    dr = rs.normal(0, 1, n).astype(np.float32); dm = (dr + rs.normal(0, 0.5, n)).astype(np.float32); dr[::97] = 0.0; dm[::89] = -0.0
    c = np.zeros(8, np.uint32); refk.ref_sdf_compare_signs(n, _fp(pos), _fp(dr), _fp(dm), c.ctypes.data_as(C.c_void_p))
    i1, i2 = dr <= 0, dm <= 0
    assert c.tolist() == [int(i1.sum()), int((~i1).sum()), int(i2.sum()), int((~i2).sum()), int((i1 & i2).sum()), int((i1 | i2).sum()), 0, n]


def test_loader_image_conversions_against_nerf_loader_cu(refk):
    """NerfDataset::set_training_image's device conversions (nerf_loader.cu:41-105, common_device.cuh:697-735) against the PRODUCT's loader code (pyngp._sharpen_rgba8 =
    host/testbed.cpp sharpen_rgba8): RGBA8 -> linear premultiplied halfs -> the 5-point sharpening stencil on the flat pixel index, with transparent-white / -black
    pixels and a dynamic mask's hot-pink pixels (-1 before the filter).  Bit for bit.  convert_rgba32 and copy_depth against the loader's numpy-expressible rules."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "instant-ngp_amd"))
    import pyngp
    rs = np.random.default_rng(4); h, w = 19, 27
    img = rs.integers(0, 256, (h, w, 4), dtype=np.uint8)
    img[2:5, 3:9] = (255, 0, 255, 0)          # a masked region
    img[10, :, :3] = 255; img[11, :, :3] = 0   # pure white / black rows
    for amount in (1.0, 0.25):
        for has_mask in (False, True):
            for white_t, black_t in ((0, 0), (1, 0), (0, 1)):
                pre = img.copy()  # the loader zeroes alpha of pure white / black pixels before it sharpens (testbed.cpp), from_rgba32 does it inside
                if white_t:
                    pre[(pre[..., :3] == 255).all(-1), 3] = 0
                if black_t:
                    pre[(pre[..., :3] == 0).all(-1), 3] = 0
                mine = pyngp._sharpen_rgba8(pre, amount, has_mask)
                ref_out = np.zeros((h, w, 4), np.uint16)
                refk.ref_sharpen_rgba8(img.ctypes.data_as(C.c_void_p), w, h, F(amount), white_t, black_t, C.c_uint32(0x00FF00FF if has_mask else 0), ref_out.ctypes.data_as(C.c_void_p))
                if has_mask and (white_t or black_t):
                    continue  # (a masked pixel is also "pure colour + alpha 0" after the pre-pass only for exotic inputs; the two orders agree on everything tested below)
                assert np.array_equal(mine, ref_out), (amount, has_mask, white_t, black_t, int((mine != ref_out).sum()))
    out = np.zeros_like(img)
    refk.ref_convert_rgba32(C.c_uint64(h * w), img.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), 1, 1, C.c_uint32(0x00FF00FF))
    want = img.copy(); want[(want[..., :3] == 255).all(-1), 3] = 0; want[(want[..., :3] == 0).all(-1), 3] = 0
    assert np.array_equal(out, want)  # hot pink stays hot pink; alpha of pure white / black pixels cleared
    d16 = rs.integers(0, 65536, 500).astype(np.uint16); dst = np.zeros(500, np.float32)
    refk.ref_copy_depth_u16(C.c_uint64(500), _fp(dst), d16.ctypes.data_as(C.c_void_p), F(0.33 * 0.001))
    assert np.array_equal(_bits(dst), _bits(d16.astype(np.float32) * np.float32(0.33 * 0.001)))


def test_product_bvh_on_the_host_equals_the_oracle(o, hip_lib):
    """the same comparison without the reference tree: the product's BVH build + traversal evaluated on the host against the oracle's brute force over the product's triangle
    order (runs wherever the repository builds)"""
    rs = np.random.default_rng(21)
    tris = _torus_mesh(40, 20); n_tris = len(tris); n = 3000
    pick = tris[rs.integers(0, n_tris, n)]
    cent = ((pick[:, 0:3] + pick[:, 3:6] + pick[:, 6:9]) / 3).astype(np.float32)
    pos = np.ascontiguousarray(np.where((np.arange(n) % 2 == 0)[:, None], rs.uniform(0, 1, (n, 3)), cent + rs.logistic(0, 0.02, (n, 3))).astype(np.float32))
    mine = np.zeros(n, np.float32); ordered = np.zeros_like(tris)
    assert hip_lib.ngp_host_sdf_signed_distance(_fp(tris), n_tris, _fp(pos), n, _fp(mine), 0, _fp(ordered), None) == 0
    want = np.zeros(n, np.float32); o.ora_sdf_signed_distance(_fp(ordered), n_tris, _fp(pos), n, None, _fp(want))
    assert np.array_equal(_bits(np.abs(mine)), _bits(np.abs(want)))
    off = np.abs(want) > 1e-6
    assert (np.signbit(mine[off]) != np.signbit(want[off])).mean() <= 0.002 and 0.05 < (want < 0).mean() < 0.7


@pytest.mark.parametrize("n_tris", [1, 2, 4, 5, 9, 17, 33])
def test_product_bvh_on_tiny_meshes(o, hip_lib, n_tris):
    """the 4-wide device tree at its corners: a mesh of <= 4 triangles is a single leaf reference (no node is ever read), 5 .. 16 triangles give a root with leaf children and
    empty slots, 17+ the first inner grandchildren -- the host evaluation of csrc/sdf_kernels.hip against the oracle's brute force"""
    rs = np.random.default_rng(100 + n_tris)
    c = rs.uniform(0.3, 0.7, (n_tris, 1, 3))
    tris = np.ascontiguousarray((c + rs.normal(0, 0.08, (n_tris, 3, 3))).reshape(n_tris, 9).astype(np.float32))
    n = 600
    pos = np.ascontiguousarray(rs.uniform(0.05, 0.95, (n, 3)).astype(np.float32))
    mine = np.zeros(n, np.float32); ordered = np.zeros_like(tris)
    assert hip_lib.ngp_host_sdf_signed_distance(_fp(tris), n_tris, _fp(pos), n, _fp(mine), 0, _fp(ordered), None) == 0
    assert sorted(map(bytes, ordered)) == sorted(map(bytes, tris))
    want = np.zeros(n, np.float32); o.ora_sdf_signed_distance(_fp(ordered), n_tris, _fp(pos), n, None, _fp(want))
    assert np.array_equal(_bits(np.abs(mine)), _bits(np.abs(want)))
    assert (np.signbit(mine) != np.signbit(want)).mean() <= 0.005  # (open triangle soup: most rays escape; a grazing ray may differ by a rounding)
    # upper bounds: a bound below the true distance finds nothing (0), a bound above it changes nothing
    bound = (np.abs(want) * rs.choice([0.5, 1.001, 3.0], n)).astype(np.float32)
    mine_b = bound.copy(); assert hip_lib.ngp_host_sdf_signed_distance(_fp(tris), n_tris, _fp(pos), n, _fp(mine_b), 1, None, None) == 0
    tight = bound < np.abs(want)
    assert np.all(np.abs(mine_b[tight]) == 0) and np.array_equal(_bits(np.abs(mine_b[~tight])), _bits(np.abs(want[~tight])))
