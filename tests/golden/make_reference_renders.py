"""Decodes the reference's own renders of the scenes tests/reference_scenes.py restates (renders/**.png, 8-bit RGB) -> tests/golden/ref_renders/<dir>_<name>.npz
(the pixels exactly as the reference wrote them).  Run in the build container: python3 tests/golden/make_reference_renders.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_sphere_mask import read_png  # noqa: E402

NAMES = ["shapes/triangles-alpha-mask", "cameras/perspective", "cameras/orthographic", "cameras/environment", "lights/point", "lights/distant", "lights/spot",
         "lights/infinite-no-map", "lights/goniometric", "objects/instances", "materials/bump", "samplers/halton", "samplers/sobol", "cameras/depth-of-field", "textures/fbm", "textures/marble", "textures/wrinkled", "textures/windy", "textures/dots", "textures/bilerp", "textures/uv", "textures/mix", "textures/scale", "textures/2d-checkerboard", "textures/2d-mappings", "textures/constant", "lights/diffuse", "samplers/random"]
TEX_CROP = (60, 195, 5, 145)  # rows, columns
# the 3-D textures do not depend on a shape's parameterisation: the sphere's corner of the image is kept; the scenes with 2-D (u, v) textures are kept whole
TEX_CROPPED = ("textures/fbm", "textures/wrinkled", "textures/windy", "textures/marble")
if __name__ == "__main__":
    os.makedirs(os.path.join(HERE, "ref_renders"), exist_ok=True)
    for n in NAMES:
        img = read_png(os.path.join("/root/reference/renders", n + ".png"))
        if n in TEX_CROPPED:
            img = img[TEX_CROP[0]:TEX_CROP[1], TEX_CROP[2]:TEX_CROP[3]]   # the sphere of the six-shape texture scenes and the wall around it
        np.savez_compressed(os.path.join(HERE, "ref_renders", n.replace("/", "_") + ".npz"), rgb=np.ascontiguousarray(img, np.uint8))
        print(n, img.shape)
    # the one image INPUT among these scenes (scenes/lights/goniometric.pbrt "string mapname")
    img = read_png("/root/reference/scenes/images/goniometric-upward-downward.png")
    np.savez_compressed(os.path.join(HERE, "ref_renders", "image_goniometric-upward-downward.npz"), rgb=np.ascontiguousarray(img, np.uint8))
    print("goniometric map", img.shape)
    img = read_png("/root/reference/scenes/images/checkerboard.png")   # the image map of scenes/textures/2d-mappings.pbrt
    np.savez_compressed(os.path.join(HERE, "ref_renders", "image_checkerboard.npz"), rgb=np.ascontiguousarray(img, np.uint8))
    print("checkerboard map", img.shape)
