"""-m gpu: the STATED tolerance against the reference's own libm, at size (round-3 review, item 3).

Bit-exactness is against the oracle's f64-libm mode.  A Rust build of the reference links glibc's f32 sin / cos / acos / atan2 / log2 (core/src/pbrt/common.rs:282-339),
whose last bit differs from the rounded f64 value for 1 - 16 % of the arguments, so against that build — oracle libm mode 0 — a film is close, not equal, and the contract is the
tolerance BASELINE.md §4 / DESIGN §2 state: per-pixel RMSE <= 1e-3 x mean and <= 0.1 % of the pixels off by more than 1e-2 x mean.  Rounds 1 - 3 showed it for configs[1] at full
size and a configs[2] crop (tests/test_big_configs_gpu.py); here: a configs[3] crop, the north-star's 1 M-triangle scene, every general material class at 256 x 256 @ 64 spp
(the toy-size test of tests/test_materials_gpu.py needed 2e-3 / 0.2 %: at 32 spp one flipped lobe choice is a thirty-second of a pixel) and a 96 x 96 window of configs[4] at its 512 spp
through glass, metal and alpha-masked foliage.  Measured values: profiles/r04_glibc_tolerance.json (scripts/glibc_tolerance.py): RMSE / mean 7e-9 ... 2.2e-4, outliers <= 0.002 %."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
import glibc_tolerance as gt  # noqa: E402  (the measuring script's cases: same crop on the device and in the oracle with libm mode 0, report of RMSE / outliers / ray counts)
import pbrt_hip  # noqa: E402
from oracle_binding import OracleScene, set_libm_mode  # noqa: E402

pytestmark = pytest.mark.gpu
RMSE_OVER_MEAN, OUTLIERS = 1e-3, 1e-3     # the stated tolerance


@pytest.fixture(autouse=True)
def _restore_libm_mode():
    yield
    set_libm_mode(0)


def _check(r, what):
    print(f"{what}: rmse/mean {r['rmse_over_mean']:.3e}, outliers {100 * r['outlier_fraction_1e-2_mean']:.4f} %, bit-equal pixels {100 * r['bit_equal_pixel_fraction']:.1f} %, "
          f"ray-count difference {r['ray_count_rel_diff']:.1e}")
    assert r["rmse_over_mean"] <= RMSE_OVER_MEAN, (what, r)
    assert r["outlier_fraction_1e-2_mean"] <= OUTLIERS, (what, r)
    assert r["ray_count_rel_diff"] <= 2e-5, (what, r)   # only a decision that flips (an edge hit, a Russian-roulette draw, a lobe choice) changes a ray count
    assert r["bit_equal_pixel_fraction"] < 1.0 or r["rmse_over_mean"] == 0.0


def test_config3_crop_glibc_mode(host):
    """configs[3]: 10 M triangles, 2048 x 2048 @ 64 spp — a 98 x 98 crop of the full-size scene."""
    _check(gt.spec_case(host, dict(n_tris=10_000_000, seed=1, xres=2048, yres=2048, spp=64, max_depth=5), (0.47, 0.518, 0.40, 0.448)), "configs[3] crop")


def test_north_star_1M_crop_glibc_mode(host):
    """The 1 M-triangle scene of the north-star's speed target, 512 x 512 @ 64 spp — a 128 x 128 crop."""
    _check(gt.spec_case(host, dict(n_tris=1_000_000, seed=1, xres=512, yres=512, spp=64, max_depth=5), (0.375, 0.625, 0.375, 0.625)), "1 M crop")


@pytest.mark.parametrize("material", ["plastic", "glass", "metal", "uber", "mixed", "textured"])
def test_general_materials_256x256_64spp_glibc_mode(host, material):
    """Every general material class (and five of them mixed, and an image-textured matte) on 100 k triangles, 256 x 256 @ 64 spp, whole frame: the stated tolerance, not the
    loosened one of the 64 x 64 @ 32 spp test."""
    _check(gt.spec_case(host, dict(n_tris=100_000, seed=1, xres=256, yres=256, spp=64, max_depth=5), (0.0, 1.0, 0.0, 1.0), material=material), f"material {material}")


def test_config4_window_through_glass_metal_foliage_glibc_mode(host):
    """configs[4] at its stated size (1920 x 1080 @ 512 spp, the San-Miguel-shaped scene): a 96 x 96 window whose first hits are glass (10 %), metal (14 %) and alpha-masked
    foliage (13 %) among floor, plastic and bumped checker (the window scripts/glibc_tolerance.py --find-c4-window picks: the 96 x 96 window with the largest smallest share)."""
    from pbrt_hip.sanmiguel import SanMiguelScene
    sm = SanMiguelScene(host, scale=1.0)
    x0, y0 = 304, 720
    mm = gt.c4_material_map(host, sm)[y0:y0 + 96, x0:x0 + 96]
    s = pbrt_hip.Scene(); M, _ = sm._materials(s); s.close()
    for group in (("glass", "frosted", "water"), ("copper", "steel", "gold", "mirror"), ("leaf_translucent", "leaf_matte", "leaf_uber")):
        share = float(np.isin(mm, [M[k] for k in group]).mean())
        assert share >= 0.05, (group, share)
    crop = (x0 / 1920, (x0 + 96) / 1920, y0 / 1080, (y0 + 96) / 1080)
    prod = pbrt_hip.Scene(); sm.capture(prod, 1920, 1080, 512, crop=crop, device_build=True)
    g = prod.render_path(max_depth=5)
    orc = OracleScene(); sm.capture(orc, 1920, 1080, 512, crop=crop)
    set_libm_mode(0)
    o = orc.render_path_ex(max_depth=5, threads=16)
    assert g[1].shape == (96, 96)
    _check(gt.report(prod, g, o), "configs[4] window")
    prod.close(); orc.close()
