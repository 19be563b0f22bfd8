"""End-to-end drop-in check of the C++ host: `pbrt_hip_render scene.pbrt` (parser -> Api -> C ABI -> GPU -> PFM) must
produce exactly the image the oracle renders for the same scene captured call by call."""
import subprocess

import numpy as np
import pytest

import driver_scene as ds
import pbrt_hip
from oracle_binding import OracleScene, set_libm_mode

pytestmark = pytest.mark.gpu


def oracle_image(host, filt, strategy, crop=None):
    set_libm_mode(1)
    try:
        with OracleScene() as o:
            cb, sb = ds.capture(o, host, filt, crop)
            xyz, wt, st = o.render_path(max_depth=ds.MAXDEPTH, light_strategy=strategy, pixel_bounds=sb)
            rgb = o.film_to_rgb(xyz, wt)
    finally:
        set_libm_mode(0)
    return rgb.reshape(cb[3] - cb[1], cb[2] - cb[0], 3), st


@pytest.mark.parametrize("filt,strategy", [("gaussian", "power"), ("box", "uniform"), ("mitchell", "power"), ("sinc", "uniform"), ("triangle", "power")])
def test_pbrt_file_renders_the_oracle_image(tmp_path, host, filt, strategy):
    path = ds.write_files(str(tmp_path), filter_line=ds.FILTERS[filt][0], strategy=strategy)
    r = subprocess.run([ds.RENDER_BIN, path], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    img = ds.read_pfm(str(tmp_path / "scene.pfm"))
    ref, st = oracle_image(host, filt, {"uniform": 0, "power": 1}[strategy])
    assert img.shape == ref.shape
    assert np.isfinite(img).all() and img.max() > 0
    assert (img.view(np.uint32) == ref.view(np.uint32)).all(), f"{(img != ref).sum()} differing values, max {np.abs(img - ref).max()}"
    # the statistics line carries the reference's ray counters
    assert f"{st.regular_rays} regular + {st.shadow_rays} shadow rays" in r.stdout


def test_crop_window_and_outfile(tmp_path, host):
    crop = (0.25, 0.75, 0.5, 1.0)
    path = ds.write_files(str(tmp_path), crop=crop)
    r = subprocess.run([ds.RENDER_BIN, "--quiet", "--outfile", str(tmp_path / "o.pfm"), path], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    img = ds.read_pfm(str(tmp_path / "o.pfm"))
    ref, _ = oracle_image(host, "gaussian", 1, crop)
    assert img.shape == ref.shape == (20, 28, 3)
    assert (img.view(np.uint32) == ref.view(np.uint32)).all()


def test_python_capture_matches_driver(tmp_path, host):
    """Same scene through the Python wrapper of the same library: equal bits (both sit on one C ABI)."""
    path = ds.write_files(str(tmp_path))
    r = subprocess.run([ds.RENDER_BIN, "--quiet", path], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    img = ds.read_pfm(str(tmp_path / "scene.pfm"))
    with pbrt_hip.Scene() as s:
        cb, sb = ds.capture(s, host)
        xyz, wt, _ = s.render_path(max_depth=ds.MAXDEPTH, light_strategy=1, pixel_bounds=sb)
        rgb = s.film_to_rgb(xyz, wt).reshape(ds.YRES, ds.XRES, 3)
    assert (img.view(np.uint32) == rgb.view(np.uint32)).all()


def test_default_light_strategy_is_spatial(tmp_path, host):
    """No lightsamplestrategy parameter = "spatial" (path.rs:314), 5 lights: the voxel distributions are built on the GPU."""
    path = ds.write_files(str(tmp_path), strategy=None, filter_line=ds.FILTERS["box"][0])
    r = subprocess.run([ds.RENDER_BIN, "--quiet", path], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    img = ds.read_pfm(str(tmp_path / "scene.pfm"))
    ref, st = oracle_image(host, "box", 2)
    assert st.light_distributions_created > 100
    assert (img.view(np.uint32) == ref.view(np.uint32)).all()
