"""-m gpu: the PRODUCT against output of the reference itself — the ten scenes of tests/test_reference_renders.py rendered by libpbrt_hip.so on the MI355X and held
against the renders the reference commits (renders/**.png), with the same thresholds as the oracle; and, for every scene, the device film against the oracle's film bit
for bit (libm mode 1), so that "oracle == reference render" and "device == oracle" are shown on the same inputs."""
import numpy as np
import pytest

import pbrt_hip
import reference_scenes as R
from oracle_binding import OracleScene, set_libm_mode
from test_reference_renders import DETERMINISTIC, NOISY, check_against_reference

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,spp,noisy", [(n, s, False) for n, s in DETERMINISTIC] + [(n, s, True) for n, s in NOISY])
def test_device_render_equals_the_references_render(host, name, spp, noisy):
    with pbrt_hip.Scene() as s:
        info = getattr(R, name)(s, host, spp=spp)
        xyz, wt, st = s.render_path(max_depth=info["max_depth"])
        rgb = s.film_to_rgb(xyz, wt)
    check_against_reference(rgb, info, noisy)
    assert st.camera_rays == rgb.shape[0] * rgb.shape[1] * spp


@pytest.mark.parametrize("name", [n for n, _ in DETERMINISTIC + NOISY])
def test_device_film_equals_the_oracle_film_on_the_reference_scenes(host, name):
    prod = pbrt_hip.Scene(); orc = OracleScene()
    info = getattr(R, name)(prod, host, spp=8, res=96)
    getattr(R, name)(orc, host, spp=8, res=96)
    set_libm_mode(1)
    try:
        oxyz, owt, ost, _ = orc.render_path_ex(max_depth=info["max_depth"])
    finally:
        set_libm_mode(0)
    gxyz, gwt, gst = prod.render_path(max_depth=info["max_depth"])
    assert np.array_equal(gwt.view(np.uint32), owt.view(np.uint32))
    assert np.array_equal(gxyz.view(np.uint32), oxyz.view(np.uint32)), float(np.abs(gxyz - oxyz).max())
    for f in ("camera_rays", "regular_rays", "shadow_rays"):
        assert getattr(gst, f) == getattr(ost, f), f
    prod.close(); orc.close()
