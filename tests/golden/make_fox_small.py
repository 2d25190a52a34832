#!/usr/bin/env python
"""Generates tests/golden/fox_small/ from the reference's shipped capture data/nerf/fox (real photos, OpenCV lens,
aabb_scale 4): every 4th frame, down-scaled 8x (135x240) with the intrinsics scaled accordingly, stored as PNG so that the
C++ loader's built-in decoder reads them. Run in the build container (needs /root/reference); the output is committed
because /root/reference does not exist on the GPU box."""
import json
import os

from PIL import Image

SRC = "/root/reference/data/nerf/fox"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fox_small")
F = 8


def main():
    j = json.load(open(os.path.join(SRC, "transforms.json")))
    os.makedirs(os.path.join(DST, "images"), exist_ok=True)
    out = {k: v for k, v in j.items() if k != "frames"}
    for k in ("fl_x", "fl_y", "cx", "cy", "w", "h"):
        out[k] = j[k] / F
    frames = [f for f in j["frames"] if os.path.exists(os.path.join(SRC, f["file_path"]))]
    out["frames"] = []
    for f in frames[::4]:
        im = Image.open(os.path.join(SRC, f["file_path"])).convert("RGB")
        im = im.resize((im.width // F, im.height // F), Image.LANCZOS)
        name = os.path.splitext(os.path.basename(f["file_path"]))[0] + ".png"
        im.save(os.path.join(DST, "images", name), optimize=True)
        out["frames"].append({"file_path": "images/" + name, "transform_matrix": f["transform_matrix"]})
    json.dump(out, open(os.path.join(DST, "transforms.json"), "w"), indent=1)
    print(len(out["frames"]), "frames ->", DST)


if __name__ == "__main__":
    main()
