"""bench.py's scene plumbing (round 6): the lego-hard stand-in, the search for a real nerf_synthetic/lego (scripts/scenes.py:25-32) and the path-based loader with its
held-out split.  The real set is not in the mount, so the lego path is exercised with a dataset written in its layout."""
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-ngp_amd")]


def test_hard_variant_is_a_different_and_harder_scene():
    import synth_scene
    easy, _, meta_e, _ = synth_scene.make_dataset(2, 96, "cpu")
    hard, xf, meta_h, _ = synth_scene.make_dataset(2, 96, "cpu", variant="hard")
    assert meta_e == meta_h and len(xf) == 2                       # same cameras / format
    e, h = easy[0].numpy().astype(np.float32), hard[0].numpy().astype(np.float32)
    assert e.shape == h.shape == (96, 96, 4) and not np.array_equal(e, h)
    cov_e, cov_h = (e[..., 3] > 0).mean(), (h[..., 3] > 0).mean()
    assert 0.1 < cov_h < 0.6 and 0.1 < cov_e < 0.6

    def edge_energy(im):  # mean absolute horizontal difference inside the covered region: texture / thin structures
        m = (im[:, 1:, 3] > 0) & (im[:, :-1, 3] > 0)
        return float(np.abs(im[:, 1:, :3] - im[:, :-1, :3])[m].mean())
    assert edge_energy(h) > 1.5 * edge_energy(e)
    assert len(synth_scene._PRIMS_HARD) > 5 * len(synth_scene._PRIMS)
    # deterministic: the scene is seeded, two calls give the same pixels
    again, _, _, _ = synth_scene.make_dataset(2, 96, "cpu", variant="hard")
    assert np.array_equal(again[0].numpy(), hard[0].numpy())


def test_find_lego_search_order(tmp_path, monkeypatch):
    import bench
    monkeypatch.delenv("NGP_LEGO_DIR", raising=False)
    if bench.find_lego() is not None:
        pytest.skip("a real lego copy is present on this box")
    d = tmp_path / "somewhere" / "lego"
    d.mkdir(parents=True)
    monkeypatch.setenv("NGP_LEGO_DIR", str(d))
    assert bench.find_lego() is None                                # no transforms_train.json there
    (d / "transforms_train.json").write_text("{}")
    assert bench.find_lego() == str(d / "transforms_train.json")


@pytest.mark.gpu
def test_scene_from_a_nerf_synthetic_layout(tmp_path, monkeypatch):
    """`--scene auto` with NGP_LEGO_DIR pointing at a dataset in the Blender layout: loaded through the C++ loader, evaluated on transforms_test.json (held-out views)."""
    import torch
    import synth_scene
    import bench
    d = tmp_path / "lego"
    synth_scene.write_dataset(str(d), n_train=6, n_test=3, res=64)
    monkeypatch.setenv("NGP_LEGO_DIR", str(d))
    args = types.SimpleNamespace(scene="auto", images=6, res=64, eval_views=2, eval_res=64, eval_spp=1, batch=1 << 14)
    sc = bench.load_scene(args)
    assert sc["which"] == "lego" and sc["n"] == 6 and sc["aabb_scale"] == 1 and len(sc["eval"]) == 2
    assert "held-out" in sc["eval_kind"] and sc["metric_scene"] == "nerf_synthetic/lego" and sc["data"].startswith("nerf_synthetic/lego") and sc["data"] != "synthetic"
    gt, rp = sc["eval"][0]
    assert tuple(gt.shape) == (64, 64, 4) and rp.resolution[0] == 64 and gt.is_cuda
    # the held-out cameras are NOT training cameras
    train = {tuple(round(v, 5) for v in sc["X"][i].start) for i in range(sc["n"])}
    assert tuple(round(v, 5) for v in rp.camera) not in train
    # an explicit path works the same way, and a scene without a test split falls back to (labelled) training views
    sc2 = bench.load_scene(args, str(d / "transforms_train.json"))
    assert sc2["n"] == 6 and len(sc2["eval"]) == 2
    os.remove(d / "transforms_test.json")
    sc3 = bench.load_scene(args, str(d))
    assert "TRAINING views" in sc3["eval_kind"] and len(sc3["eval"]) == 2
    # ... and trains: a few steps through the library on the loaded scene
    import ngp_abi as A
    lib = A.load_hip(); A.check(lib, lib.ngp_init())
    _, _, model, nerf = bench.make_trainer(lib, sc, args.batch)
    A.check(lib, lib.ngp_nerf_train(nerf, None, 20)); torch.cuda.synchronize()
    st = bench.get_stats(lib, nerf)
    assert st.training_step == 20 and np.isfinite(st.loss)
    assert bench.eval_psnr(lib, nerf, sc, 1) > 5.0
    lib.ngp_nerf_destroy(nerf); lib.ngp_model_destroy(model)
