"""-m gpu: regression test for the "wrong values in OTHER lanes" bug HISTORY.md §4 records (found with scripts/bump_probe.py).

An earlier version of the one-lobe shade kernel called the bump-map evaluator out of line with about forty scalar arguments, most of them passed on
the stack; on gfx950 (ROCm 7.2) lanes of the wave that did NOT take the call then carried wrong values — visible as whole columns of differing pixels
on a bump-mapped floor at 1 and 4 samples per pixel, in the one-lobe kernel and, with one non-matte triangle in the scene, in the general-BSDF kernel.
The arguments now travel through one private struct (csrc/texture.h hit_bump / BumpIn).  These are the probe's scenes with its print statements turned
into assertions: every film must equal the oracle's bit for bit."""
import numpy as np
import pytest

import pbrt_hip
from oracle_binding import OracleScene, set_libm_mode
from texture_scenes import make_image, textured_quad_scene

pytestmark = pytest.mark.gpu


def _differing_pixels(host, bump, depth, spp=4, general_kernel=False):
    def material(sc, tex):
        m = sc.add_material_matte((0.6, 0.6, 0.6), 0.0)
        sc.set_material_bump(m, bump(sc))
        return m

    def extra(sc):
        if general_kernel:   # one far-away glass triangle switches the whole scene to shade_kernel<GEN = true>
            gl = sc.add_material_glass((1, 1, 1), (1, 1, 1), 0.0, 0.0, 1.5, True)
            sc.add_mesh(np.array([[30, 30, 30], [31, 30, 30], [30, 31, 30]], np.float32), [0, 1, 2], gl)

    prod = pbrt_hip.Scene(); orc = OracleScene()
    for s in (prod, orc):
        textured_quad_scene(s, host, lambda sc: sc.add_texture_constant((0.5, 0.5, 0.5)), res=48, spp=spp, material=material, extra=extra)
    set_libm_mode(1)
    try:
        o = orc.render_path_ex(max_depth=depth)
    finally:
        set_libm_mode(0)
    g = prod.render_path(max_depth=depth)
    assert (g[2].regular_rays, g[2].shadow_rays) == (o[2].regular_rays, o[2].shadow_rays)
    d = (g[0].view(np.uint32) != o[0].view(np.uint32)).any(axis=2)
    prod.close(); orc.close()
    return int(d.sum()), sorted(set(np.where(d)[1].tolist()))[:12]


IMG = lambda **k: (lambda sc: sc.add_texture_imagemap(sc.add_mipmap(make_image(32, 32, seed=21), as_float=True, **k), su=2.0, sv=2.0))
BUMPS = {
    "const": lambda sc: sc.add_texture_constant(0.3),
    "dots": lambda sc: sc.add_texture_dots(sc.add_texture_constant(0.02), sc.add_texture_constant(0.0), su=6.0, sv=6.0),
    "image_ewa": IMG(),
    "image_trilinear": IMG(trilinear=True),
}


@pytest.mark.parametrize("name", sorted(BUMPS))
@pytest.mark.parametrize("depth", [1, 4])
def test_bumped_floor_one_lobe_kernel_every_lane_right(host, name, depth):
    n, cols = _differing_pixels(host, BUMPS[name], depth)
    assert n == 0, f"{n} pixels differ from the oracle (columns {cols}): lanes outside the out-of-line call carry wrong values"


@pytest.mark.parametrize("general_kernel", [False, True])
@pytest.mark.parametrize("spp", [1, 4])
def test_bumped_floor_both_shade_kernels(host, general_kernel, spp):
    n, cols = _differing_pixels(host, BUMPS["const"], 1, spp=spp, general_kernel=general_kernel)
    assert n == 0, f"{n} pixels differ from the oracle (columns {cols}), general kernel = {general_kernel}, spp = {spp}"
