"""Synthetic stand-in for nerf_synthetic/lego (the real dataset is not shipped with the reference and
there is no network): an analytic scene of boxes and spheres ray-traced into RGBA8 views with the
SAME on-disk/in-memory conventions as the Blender set -- 800x800, camera_angle_x = 0.6911, cameras on
the upper hemisphere at radius 4.03 looking at the origin, transforms.json + PNG with alpha, aabb_scale 1
(SURVEY.md section 8d item 2).  Plumbing only (torch is used as an array library on cpu or cuda).

NeRF -> NGP conventions restated from nerf_loader.h:29 (NERF_SCALE 0.33), nerf_loader.cu:403-404
(offset 0.5) and nerf_loader.h:101-120 (nerf_matrix_to_ngp).
"""
import json
import math
import os

import numpy as np
import torch

CAMERA_ANGLE_X = 0.6911112070083618
RADIUS = 4.031128874
NERF_SCALE = 0.33
NERF_OFFSET = 0.5

# (kind, params, base colour); boxes: (min, max); spheres: (centre, radius). NeRF world units.
_PRIMS = [
    ("box", ((-0.9, -0.9, -0.35), (0.9, 0.9, -0.25)), (0.55, 0.55, 0.52)),
    ("box", ((-0.55, -0.35, -0.25), (0.25, 0.35, 0.10)), (0.85, 0.65, 0.10)),
    ("box", ((-0.35, -0.25, 0.10), (0.15, 0.25, 0.40)), (0.80, 0.15, 0.12)),
    ("box", ((0.25, -0.12, -0.25), (0.80, 0.12, -0.05)), (0.15, 0.35, 0.75)),
    ("box", ((-0.80, -0.70, -0.25), (-0.62, -0.52, 0.45)), (0.20, 0.60, 0.25)),
    ("sphere", ((0.45, 0.50, -0.02), 0.23), (0.90, 0.90, 0.92)),
    ("sphere", ((-0.10, 0.0, 0.55), 0.16), (0.95, 0.75, 0.20)),
    ("sphere", ((0.55, -0.55, -0.10), 0.15), (0.60, 0.20, 0.70)),
]


def _hard_prims():
    """The "hard" variant (round 6): what makes nerf_synthetic/lego slow to fit and the stand-in above easy is missing there -- thin structures (a grille of
    2-3 cm bars, poles, a stud field), many small occluders, and texture at the scale of a pixel.  Deterministic (seeded), same cameras / format / aabb_scale."""
    rng = np.random.default_rng(20240930)
    prims = [("box", ((-0.95, -0.95, -0.38), (0.95, 0.95, -0.30)), (0.50, 0.50, 0.48))]   # base plate
    # stud field on the plate: 9 x 9 short cylinders approximated by small spheres
    for ix in range(9):
        for iy in range(9):
            if (ix + iy) % 2 == 0:
                prims.append(("sphere", ((-0.8 + 0.2 * ix, -0.8 + 0.2 * iy, -0.29), 0.045), (0.75, 0.72, 0.20)))
    # body: three stacked blocks
    prims += [("box", ((-0.50, -0.30, -0.30), (0.30, 0.30, 0.00)), (0.85, 0.62, 0.08)),
              ("box", ((-0.40, -0.22, 0.00), (0.20, 0.22, 0.22)), (0.80, 0.12, 0.10)),
              ("box", ((0.30, -0.10, -0.30), (0.85, 0.10, -0.12)), (0.12, 0.32, 0.78))]
    # grille: thin bars (2.5 cm) in two directions above the body
    for k in range(9):
        x = -0.45 + 0.1 * k
        prims.append(("box", ((x, -0.34, 0.24), (x + 0.025, 0.34, 0.265)), (0.15, 0.15, 0.17)))
    for k in range(7):
        y = -0.30 + 0.1 * k
        prims.append(("box", ((-0.47, y, 0.265), (0.42, y + 0.025, 0.29)), (0.70, 0.70, 0.74)))
    # poles and a boom of thin boxes
    for (x, y, h) in ((-0.85, -0.75, 0.55), (-0.85, 0.75, 0.35), (0.85, 0.75, 0.65), (0.85, -0.75, 0.25), (0.0, 0.85, 0.50), (0.0, -0.85, 0.45)):
        prims.append(("box", ((x - 0.015, y - 0.015, -0.30), (x + 0.015, y + 0.015, h)), (0.18, 0.55, 0.25)))
    prims.append(("box", ((-0.85, -0.76, 0.52), (0.85, -0.74, 0.54)), (0.85, 0.85, 0.30)))
    prims.append(("box", ((-0.86, -0.75, 0.30), (-0.84, 0.75, 0.32)), (0.85, 0.40, 0.30)))
    # scattered small spheres
    for _ in range(24):
        c = (float(rng.uniform(-0.8, 0.8)), float(rng.uniform(-0.8, 0.8)), float(rng.uniform(-0.2, 0.6)))
        prims.append(("sphere", (c, float(rng.uniform(0.03, 0.07))), tuple(float(v) for v in rng.uniform(0.15, 0.95, 3))))
    return prims


_PRIMS_HARD = _hard_prims()


def camera_poses(n, seed_phase=0.0):
    """n camera-to-world matrices (NeRF / OpenGL convention: -Z forward, +Y up), Fibonacci hemisphere."""
    poses = []
    golden = math.pi * (3.0 - math.sqrt(5.0))
    for i in range(n):
        h = 0.12 + 0.80 * ((i + 0.5) / n)  # z / R in (0.12, 0.92)
        phi = golden * i + seed_phase
        r = math.sqrt(max(0.0, 1.0 - h * h))
        pos = np.array([r * math.cos(phi), r * math.sin(phi), h], dtype=np.float64) * RADIUS
        fwd = -pos / np.linalg.norm(pos)
        up0 = np.array([0.0, 0.0, 1.0])
        right = np.cross(fwd, up0); right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        c2w = np.eye(4)
        c2w[:3, 0] = right; c2w[:3, 1] = up; c2w[:3, 2] = -fwd; c2w[:3, 3] = pos
        poses.append(c2w)
    return poses


def nerf_matrix_to_ngp(c2w):
    """nerf_loader.h:101-120 -> 12 floats, column-major mat4x3."""
    m = np.array(c2w[:3, :4], dtype=np.float64)
    m[:, 1] *= -1.0
    m[:, 2] *= -1.0
    m[:, 3] = m[:, 3] * NERF_SCALE + NERF_OFFSET
    m = m[[1, 2, 0], :]  # cycle axes xyz <- yzx
    return m.T.reshape(-1).astype(np.float32)  # columns contiguous


def _intersect(o, d, variant="lego-format"):
    """o, d: [N,3] torch (NeRF world). Returns rgb [N,3] in linear-ish display space and alpha [N].
    variant "hard": _PRIMS_HARD, texture at 3 x the spatial frequency, and a VIEW-DEPENDENT term (Blinn-Phong highlight) so that the colour network has something to fit."""
    hard = variant == "hard"
    N = o.shape[0]
    dev, dt = o.device, o.dtype
    best_t = torch.full((N,), float("inf"), device=dev, dtype=dt)
    rgb = torch.zeros((N, 3), device=dev, dtype=dt)
    nrm = torch.zeros((N, 3), device=dev, dtype=dt)
    light = torch.tensor([0.35, -0.45, 0.82], device=dev, dtype=dt)
    light = light / light.norm()
    inv = 1.0 / torch.where(d.abs() < 1e-9, torch.full_like(d, 1e-9), d)
    for kind, prm, col in (_PRIMS_HARD if hard else _PRIMS):
        colt = torch.tensor(col, device=dev, dtype=dt)
        if kind == "box":
            lo = torch.tensor(prm[0], device=dev, dtype=dt); hi = torch.tensor(prm[1], device=dev, dtype=dt)
            t0 = (lo - o) * inv; t1 = (hi - o) * inv
            tn = torch.minimum(t0, t1); tf = torch.maximum(t0, t1)
            tnear, axis = tn.max(dim=1); tfar = tf.min(dim=1).values
            hit = (tnear < tfar) & (tnear > 0) & (tnear < best_t)
            n = torch.zeros((N, 3), device=dev, dtype=dt)
            n.scatter_(1, axis[:, None], -torch.sign(torch.gather(d, 1, axis[:, None])))
            t = tnear
        else:
            c = torch.tensor(prm[0], device=dev, dtype=dt); r = prm[1]
            oc = o - c
            b = (oc * d).sum(1); cc = (oc * oc).sum(1) - r * r
            disc = b * b - cc
            t = -b - torch.sqrt(disc.clamp_min(0))
            hit = (disc > 0) & (t > 0) & (t < best_t)
            n = (oc + d * t[:, None]) / r
        p = o + d * t[:, None]
        # procedural albedo variation (studs / stripes) so the views carry high-frequency detail
        if hard:
            tex = 0.70 + 0.30 * torch.sin(p[:, 0] * 61.0) * torch.sin(p[:, 1] * 57.0) * torch.sin(p[:, 2] * 53.0 + 0.5) + 0.12 * torch.sign(torch.sin(p[:, 0] * 140.0 + p[:, 1] * 90.0))
            shade = 0.30 + 0.60 * (n * light).sum(1).clamp_min(0)
            hvec = light[None, :] - d; hvec = hvec / hvec.norm(dim=1, keepdim=True)
            spec = 0.45 * (n * hvec).sum(1).clamp_min(0) ** 24                      # view dependent
            c_out = colt[None, :] * (tex * shade)[:, None] + spec[:, None]
        else:
            tex = 0.82 + 0.18 * torch.sign(torch.sin(p[:, 0] * 19.0) * torch.sin(p[:, 1] * 19.0) * torch.sin(p[:, 2] * 19.0 + 0.5))
            shade = 0.35 + 0.65 * (n * light).sum(1).clamp_min(0)
            c_out = colt[None, :] * (tex * shade)[:, None]
        best_t = torch.where(hit, t, best_t)
        rgb = torch.where(hit[:, None], c_out, rgb)
    alpha = torch.isfinite(best_t).to(dt)
    return rgb.clamp(0, 1), alpha


def render_view(c2w, res, device="cpu", variant="lego-format"):
    """RGBA8 image [H, W, 4] uint8 (sRGB-encoded colour, straight alpha), like a nerf_synthetic PNG."""
    W = H = res
    focal = 0.5 * W / math.tan(0.5 * CAMERA_ANGLE_X)
    ys, xs = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float64), torch.arange(W, device=device, dtype=torch.float64), indexing="ij")
    dirs = torch.stack([(xs + 0.5 - 0.5 * W) / focal, -(ys + 0.5 - 0.5 * H) / focal, -torch.ones_like(xs)], dim=-1).reshape(-1, 3)
    c2w_t = torch.tensor(c2w, device=device, dtype=torch.float64)
    d = dirs @ c2w_t[:3, :3].T
    d = d / d.norm(dim=1, keepdim=True)
    o = c2w_t[:3, 3][None, :].expand_as(d)
    rgb, alpha = _intersect(o, d, variant)
    img = torch.cat([rgb * alpha[:, None], alpha[:, None]], dim=1)  # non-hit pixels are (0,0,0,0)
    return (img.reshape(H, W, 4) * 255.0 + 0.5).clamp(0, 255).to(torch.uint8)


def make_dataset(n_images=100, res=800, device="cpu", phase=0.0, variant="lego-format"):
    """In-memory dataset: list of uint8 [H,W,4] tensors (on `device`), per-image metadata dicts."""
    poses = camera_poses(n_images, phase)
    focal = 0.5 * res / math.tan(0.5 * CAMERA_ANGLE_X)  # nerf_loader.cu:256-263 (camera_angle_x -> focal length)
    images, xforms = [], []
    for c2w in poses:
        images.append(render_view(c2w, res, device, variant))
        xforms.append(nerf_matrix_to_ngp(c2w))
    meta = dict(resolution=(res, res), focal_length=(focal, focal), principal_point=(0.5, 0.5), aabb_scale=1)
    return images, xforms, meta, poses


def write_dataset(path, n_train=100, n_test=8, res=800, device="cpu"):
    """Write transforms_train.json / transforms_test.json + PNGs (the nerf_synthetic layout)."""
    from PIL import Image
    os.makedirs(os.path.join(path, "train"), exist_ok=True)
    os.makedirs(os.path.join(path, "test"), exist_ok=True)
    for split, n, phase in (("train", n_train, 0.0), ("test", n_test, 1.234)):
        images, _, _, poses = make_dataset(n, res, device, phase)
        frames = []
        for i, (img, c2w) in enumerate(zip(images, poses)):
            Image.fromarray(img.cpu().numpy(), "RGBA").save(os.path.join(path, split, f"r_{i}.png"))
            frames.append({"file_path": f"./{split}/r_{i}", "rotation": 0.0, "transform_matrix": [[float(v) for v in row] for row in c2w]})
        with open(os.path.join(path, f"transforms_{split}.json"), "w") as f:
            json.dump({"camera_angle_x": CAMERA_ANGLE_X, "aabb_scale": 1, "frames": frames}, f, indent=1)
