"""-m gpu: the wavefront path tracer (K1 raygen, K4 shade, K6 resolve, K7 film) against the CPU oracle.

Two comparisons per scene:
  * oracle in libm mode 1 (sin/cos/acos/atan2 evaluated in f64 and rounded): the device evaluates them the same way, every
    other operation is IEEE-exact f32 in the reference's order, so the FILM must match bit for bit;
  * oracle in libm mode 0 (glibc's f32 routines = what the Rust reference links): per-pixel L2 within the stated tolerance
    RMSE <= 1e-3 x mean luminance, outliers (|d| > 1e-2 x mean) <= 0.1 % of pixels.
"""
import numpy as np
import pytest

import pbrt_hip
import scenes
from oracle_binding import OracleScene, set_libm_mode

pytestmark = pytest.mark.gpu

TOL_RMSE = 1e-3
TOL_OUTLIER_FRAC = 1e-3


def _render_both(capture, libm_mode, **kw):
    prod = pbrt_hip.Scene(); orc = OracleScene()
    capture(prod); capture(orc)
    set_libm_mode(libm_mode)
    try:
        oxyz, owt, ost, _ = orc.render_path_ex(**kw)
    finally:
        set_libm_mode(0)
    gxyz, gwt, gst = prod.render_path(**kw)
    return (gxyz, gwt, gst), (oxyz, owt, ost), prod, orc


def _bits_equal(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


def _l2_report(prod, g, o):
    grgb = prod.film_to_rgb(g[0], g[1]); orgb = prod.film_to_rgb(o[0], o[1])
    mean_lum = float(orgb.mean())
    d = grgb - orgb
    rmse = float(np.sqrt((d ** 2).mean()))
    outliers = float((np.abs(d).max(axis=2) > 1e-2 * mean_lum).mean())
    return rmse / max(mean_lum, 1e-20), outliers, mean_lum


def test_camera_rays_bit_exact(host):
    spec = pbrt_hip.SceneSpec(n_tris=10, xres=200, yres=150, spp=8)
    prod = pbrt_hip.Scene(); orc = OracleScene()
    pbrt_hip.capture_spec(spec, prod, host); pbrt_hip.capture_spec(spec, orc, host)
    for s in (0, 3, 7):
        gr, gp = prod.generate_camera_rays([0, 0, 200, 150], s)
        orr, op = orc.generate_camera_rays([0, 0, 200, 150], s)
        assert _bits_equal(gp, op)
        for f in ("o", "d", "t_max", "time"):
            assert _bits_equal(np.ascontiguousarray(gr[f]), np.ascontiguousarray(orr[f])), f


def test_camera_rays_thin_lens(host):
    """lens_radius > 0 goes through concentric_sample_disk (cos/sin): bit-exact against libm mode 1."""
    spec = pbrt_hip.SceneSpec(n_tris=10, xres=64, yres=64, spp=4)
    prod = pbrt_hip.Scene(); orc = OracleScene()
    for s in (prod, orc):
        pbrt_hip.capture_spec(spec, s, host)
        w2c, c2w = host.look_at(spec.eye, spec.look, spec.up)
        s.set_camera_perspective(host.perspective_raster_to_camera(spec.fov, 64, 64), c2w, lens_radius=0.05, focal_distance=4.0, shutter_open=0.25, shutter_close=0.75)
    set_libm_mode(1)
    try:
        orr, op = orc.generate_camera_rays([0, 0, 64, 64], 2)
    finally:
        set_libm_mode(0)
    gr, gp = prod.generate_camera_rays([0, 0, 64, 64], 2)
    for f in ("o", "d", "t_max", "time"):
        assert _bits_equal(np.ascontiguousarray(gr[f]), np.ascontiguousarray(orr[f])), f


@pytest.mark.parametrize("n_tris,res,spp,depth", [(2000, 64, 4, 5), (20000, 96, 8, 3), (300, 40, 16, 8)])
def test_infinite_light_scene_film_bit_exact_and_l2(host, n_tris, res, spp, depth):
    spec = pbrt_hip.SceneSpec(n_tris=n_tris, seed=3, xres=res, yres=res, spp=spp, max_depth=depth)
    cap = lambda s: pbrt_hip.capture_spec(spec, s, host)
    g, o, prod, orc = _render_both(cap, 1, max_depth=depth)
    assert g[2].camera_rays == o[2].camera_rays == res * res * spp
    assert (g[2].regular_rays, g[2].shadow_rays) == (o[2].regular_rays, o[2].shadow_rays)
    assert (g[2].paths_total, g[2].paths_zero_radiance) == (o[2].paths_total, o[2].paths_zero_radiance)
    assert _bits_equal(g[1], o[1]), "filter weight sums differ"
    nb = (g[0].view(np.uint32) != o[0].view(np.uint32)).any(axis=2).sum()
    assert nb == 0, f"{nb} of {res * res} pixels differ from the f64-libm oracle"
    # against the glibc-f32 oracle: tolerance
    g2, o2, prod2, _ = _render_both(cap, 0, max_depth=depth)
    rel_rmse, outliers, mean_lum = _l2_report(prod2, g2, o2)
    assert rel_rmse <= TOL_RMSE and outliers <= TOL_OUTLIER_FRAC, (rel_rmse, outliers, mean_lum)
