#!/usr/bin/env python
"""Stage the reference's SHIPPED DATASETS (and its driver script, as a test harness) next to the repo so that they travel to the GPU box.

/root/reference exists only in the build container; `gpurun` / the driver snapshot /root/repo.  Like oracle/_ref (a compiled
reference binary), `_ref_data/` is git-ignored -- nothing of the reference enters this repository's history -- but it is NOT
gpurun-ignored, so `bench.py --scene fox`, the full-resolution fox test, the image / SDF trainers' tests and the
`scripts/run.py` drop-in test find their inputs on the GPU box.  Everything that reads `_ref_data/` skips (tests) or fails loudly
(bench options that were asked for explicitly) when it is absent.

  data/nerf/fox          50 JPEGs 1080x1920 + transforms.json   (BASELINE.json config 2)
  data/image/albert.exr  1024^2 RGBA float32                     (config 0)
  data/sdf/armadillo.obj 49,990 vertices                         (config 4)
  configs/{nerf,image,sdf}/*.json             the reference's network configs (run.py resolves them against ROOT_DIR = the parent of scripts/)
  scripts/{run,common,scenes,constants}.py   executed UNMODIFIED by tests/test_run_py_dropin.py against this repo's pyngp
"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("NGP_REFERENCE_DIR", "/root/reference")
DST = os.path.join(ROOT, "_ref_data")

ITEMS = ["configs/nerf", "configs/image", "configs/sdf", "data/nerf/fox", "data/image/albert.exr", "data/sdf/armadillo.obj", "scripts/run.py", "scripts/common.py", "scripts/scenes.py", "scripts/constants.py"]


def stage(verbose=False):
    if not os.path.isdir(REF):
        return False
    for rel in ITEMS:
        src, dst = os.path.join(REF, rel), os.path.join(DST, rel)
        if not os.path.exists(src):
            continue
        if os.path.isdir(src):
            if not os.path.isdir(dst) or len(os.listdir(dst)) != len(os.listdir(src)):
                shutil.copytree(src, dst, dirs_exist_ok=True)
        elif not os.path.exists(dst) or os.path.getsize(dst) != os.path.getsize(src):
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            shutil.copyfile(src, dst)
        if verbose:
            print("staged", rel)
    return True


if __name__ == "__main__":
    ok = stage(verbose=True)
    print("reference present:", ok, "->", DST)
    sys.exit(0)
