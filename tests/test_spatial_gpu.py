"""-m gpu: SpatialLightDistribution on the device (csrc/spatial.h) against the oracle's restatement of
core/src/light_distrib/spatial.rs — film bit-exact (f64-libm mode), same set of voxel distributions created."""
import numpy as np
import pytest

import pbrt_hip
import scenes
from oracle_binding import OracleScene, set_libm_mode

pytestmark = pytest.mark.gpu


def _both(cap, **kw):
    prod = pbrt_hip.Scene(); orc = OracleScene()
    cap(prod); cap(orc)
    set_libm_mode(1)
    try:
        oxyz, owt, ost, _ = orc.render_path_ex(light_strategy=2, **kw)
    finally:
        set_libm_mode(0)
    gxyz, gwt, gst = prod.render_path(light_strategy=2, **kw)
    return (gxyz, gwt, gst), (oxyz, owt, ost), prod, orc


def _check(g, o):
    assert (g[2].regular_rays, g[2].shadow_rays, g[2].paths_total, g[2].paths_zero_radiance) == \
           (o[2].regular_rays, o[2].shadow_rays, o[2].paths_total, o[2].paths_zero_radiance), (g[2].as_dict(), o[2].as_dict())
    assert g[2].light_distributions_created == o[2].light_distributions_created > 0
    assert np.array_equal(g[1].view(np.uint32), o[1].view(np.uint32))
    nb = int((g[0].view(np.uint32) != o[0].view(np.uint32)).any(axis=2).sum())
    assert nb == 0, f"{nb} pixels differ"
    assert float(o[0].max()) > 0


@pytest.mark.parametrize("opts", [dict(), dict(two_sided=True, with_normals=True)])
def test_box_scene_two_area_lights(host, opts):
    g, o, _, _ = _both(scenes.cornell_like(host, **opts), max_depth=4)
    _check(g, o)


def _emissive_grid_scene(host, n, extra_lights=True, res=40, spp=4):
    """A floor, an occluder and an n x n emissive quad grid (2 n^2 area lights) + delta and infinite lights."""
    def cap(s):
        if extra_lights:
            s.add_light_infinite((0.2, 0.25, 0.3))
            s.add_light_point((5, 4, 3), (-1.5, -1.0, 1.0))
            s.add_light_distant((0.8, 0.8, 0.7), np.float32([0.3, -0.2, 0.93]) / np.float32(np.linalg.norm([0.3, -0.2, 0.93])))
        white = s.add_material_matte((0.7, 0.7, 0.7), 0.0)
        blue = s.add_material_matte((0.2, 0.3, 0.7), 20.0)
        P, idx = scenes.grid_mesh(8, z=-1.0, size=2.0)
        s.add_mesh(P, idx, white)
        P, idx = scenes.grid_mesh(n, z=1.5, size=0.8)
        lid = s.add_light_diffuse_area((6.0, 5.0, 4.0), len(idx) // 3, two_sided=False)
        s.add_mesh(P, idx, white, first_area_light=lid, reverse_orientation=True)  # faces down
        P, idx = scenes.grid_mesh(2, z=0.2, size=0.5)
        s.add_mesh(P + np.float32([0.3, 0.1, 0]), idx, blue)
        w2c, c2w = host.look_at([0.5, -4.5, 1.0], [0, 0, 0], [0, 0, 1])
        s.set_camera_perspective(host.perspective_raster_to_camera(45.0, res, res), c2w)
        cb, table, sb = host.film_box(res, res)
        s.set_film(res, res, cb, (0.5, 0.5), table)
        s.set_sampler(0, spp, sb)
        s.build_accel(0, 4)
    return cap


def test_mixed_lights_emissive_mesh(host):
    g, o, _, _ = _both(_emissive_grid_scene(host, 6), max_depth=5)
    _check(g, o)
    assert g[2].light_distributions_created > 200


def test_many_lights_cross_the_scan_chunk(host):
    """2 * 34^2 = 2312 area lights + 3: the sequential sum / CDF scan runs over more than one 2048-entry LDS chunk."""
    g, o, _, _ = _both(_emissive_grid_scene(host, 34, res=24, spp=2), max_depth=3)
    _check(g, o)


def test_spatial_chunked_and_tile_parts(host, monkeypatch):
    """Voxel tables persist across sample chunks of one render; two half-frame renders add up to the full frame."""
    cap = _emissive_grid_scene(host, 4, res=32, spp=8)
    with pbrt_hip.Scene() as s:
        cap(s)
        x1, w1, st1 = s.render_path(max_depth=4, light_strategy=2)
        monkeypatch.setenv("PBRT_HIP_MAX_PATHS", str(32 * 32 * 2))
        x2, w2, st2 = s.render_path(max_depth=4, light_strategy=2)
        monkeypatch.delenv("PBRT_HIP_MAX_PATHS")
        assert st2.extend_launches > st1.extend_launches
        assert np.array_equal(x1.view(np.uint32), x2.view(np.uint32))
        assert st1.light_distributions_created == st2.light_distributions_created
        xa, wa, _ = s.render_path(max_depth=4, light_strategy=2, tile_part=0, tile_parts=2)
        xb, wb, _ = s.render_path(max_depth=4, light_strategy=2, tile_part=1, tile_parts=2)
        assert np.array_equal((xa + xb).view(np.uint32), x1.view(np.uint32)) and np.array_equal((wa + wb).view(np.uint32), w1.view(np.uint32))


def test_single_light_forces_uniform(host):
    """create_light_sample_distribution: one light -> UniformLightDistribution whatever the strategy (light_distrib/mod.rs:59-64)."""
    spec = pbrt_hip.SceneSpec(n_tris=500, xres=32, yres=32, spp=2)
    with pbrt_hip.Scene() as s:
        pbrt_hip.capture_spec(spec, s, host)
        x0, w0, st0 = s.render_path(light_strategy=0)
        x2, w2, st2 = s.render_path(light_strategy=2)
    assert np.array_equal(x0.view(np.uint32), x2.view(np.uint32)) and st2.light_distributions_created == 0


def test_pool_exhaustion_is_an_error(host, monkeypatch):
    cap = _emissive_grid_scene(host, 4, res=32, spp=2)
    monkeypatch.setenv("PBRT_HIP_SPATIAL_POOL_BYTES", str(64 * (2 * 35 + 2) * 4))  # room for 64 voxels only
    with pbrt_hip.Scene() as s:
        cap(s)
        with pytest.raises(pbrt_hip.PbrtHipError) as e:
            s.render_path(max_depth=3, light_strategy=2)
        assert e.value.code == pbrt_hip.ERR_OOM
        monkeypatch.delenv("PBRT_HIP_SPATIAL_POOL_BYTES")
        x, w, st = s.render_path(max_depth=3, light_strategy=2)   # the handle stays usable
        assert st.light_distributions_created > 64
