"""-m gpu: the multi-device handle (pbrt_hip_scene_create_multi): ONE handle, several device contexts, tiles dealt round-robin, the film tiles gathered on
the first device and merged in tile order.  The test box has one GPU, so the contexts share it (ordinals repeat: the exchange is a device-to-device copy);
what RCCL contributes on a real node — the send / receive group — is exercised on a one-rank communicator by the library's self-test."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import pbrt_hip
from oracle_binding import OracleScene, set_libm_mode

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPEC = dict(n_tris=30_000, seed=3, xres=200, yres=136, spp=8, max_depth=5)


def _render(host, devices=None, **kw):
    s = pbrt_hip.Scene(devices=devices) if devices is not None else pbrt_hip.Scene()
    # the multi-device handles build their tree on the first GPU (it reaches the other contexts through its host copy), the one-device scene on the host: the same tree
    pbrt_hip.capture_spec(pbrt_hip.SceneSpec(**SPEC), s, host, device_build=devices is not None)
    out = s.render_path(**kw)
    return s, out


@pytest.mark.parametrize("n", [2, 3, 8])
def test_multi_device_film_equals_one_device_film(host, n):
    one, (x1, w1, st1) = _render(host)
    multi, (xn, wn, stn) = _render(host, devices=[0] * n)
    assert multi.devices() == [0] * n and one.devices() == [0]
    assert np.array_equal(xn.view(np.uint32), x1.view(np.uint32)) and np.array_equal(wn, w1)
    assert (stn.camera_rays, stn.regular_rays, stn.shadow_rays, stn.paths_total, stn.paths_zero_radiance) == \
           (st1.camera_rays, st1.regular_rays, st1.shadow_rays, st1.paths_total, st1.paths_zero_radiance)
    # a second frame reuses the replicas (no re-upload) and gives the same film; a changed sampler reaches every context
    x2, w2, _ = multi.render_path()
    assert np.array_equal(x2.view(np.uint32), x1.view(np.uint32))
    cb, table, sb = host.film_box(SPEC["xres"], SPEC["yres"])
    for s in (one, multi):
        s.set_sampler(0, 4, sb)
    xa, wa, _ = one.render_path(); xb, wb, _ = multi.render_path()
    assert np.array_equal(xa.view(np.uint32), xb.view(np.uint32)) and np.array_equal(wa, wb) and float(wa.max()) <= 6
    one.close(); multi.close()


def test_multi_device_handle_as_one_rank_of_several(host):
    """tile_part / tile_parts still work on a multi-device handle: two handles of two devices each = a four-way partition."""
    one, (x1, w1, st1) = _render(host)
    acc = np.zeros_like(x1); accw = np.zeros_like(w1); rays = 0
    for part in range(2):
        m, (x, w, st) = _render(host, devices=[0, 0], tile_part=part, tile_parts=2)
        acc += x; accw += w; rays += st.regular_rays + st.shadow_rays
        m.close()
    assert rays == st1.regular_rays + st1.shadow_rays
    assert np.array_equal(acc.view(np.uint32), x1.view(np.uint32)) and np.array_equal(accw, w1)
    one.close()


def test_multi_device_scene_change_reaches_every_context(host):
    """Capture calls act on the handle; the next render must replicate the change (here: a second light and a rebuilt BVH)."""
    def build(s):
        pbrt_hip.capture_spec(pbrt_hip.SceneSpec(**SPEC), s, host)
        s.render_path()
        s.add_light_point((20, 18, 15), (0.2, -1.5, 1.0))
        P = np.array([[-1, -1, -1.2], [1, -1, -1.2], [0, 1, -1.2]], np.float32)
        s.add_mesh(P, [0, 1, 2], s.add_material_matte((0.8, 0.2, 0.2), 0.0))
        s.build_accel_best(0, 4)
        return s.render_path(light_strategy=1)
    a = pbrt_hip.Scene(); b = pbrt_hip.Scene(devices=[0, 0, 0])
    xa, wa, sa = build(a); xb, wb, sb = build(b)
    assert np.array_equal(xa.view(np.uint32), xb.view(np.uint32)) and np.array_equal(wa, wb)
    assert (sa.regular_rays, sa.shadow_rays) == (sb.regular_rays, sb.shadow_rays)
    a.close(); b.close()


def test_rccl_binding_selftest(host):
    """dlopen(librccl), ncclCommInitAll and the ncclSend / ncclRecv group of the film-tile gather, on the communicator this box allows (one rank)."""
    with pbrt_hip.Scene() as s:
        assert s.selftest_rccl_gather(1 << 20) == 0
    with pbrt_hip.Scene(devices=[0, 0]) as s:    # RCCL refuses one device twice: the library says so instead of hanging
        with pytest.raises(pbrt_hip.PbrtHipError) as e:
            s.selftest_rccl_gather(1024)
        assert e.value.code == pbrt_hip.ERR_UNSUPPORTED


def test_front_end_renders_on_several_devices(tmp_path):
    """pbrt_hip_render --devices 0,0 writes the same image as --device 0."""
    exe = os.path.join(ROOT, "pbrt-v3-rs_amd", "pbrt_hip_render")
    scene = tmp_path / "s.pbrt"
    scene.write_text('LookAt 0 -4 1  0 0 0  0 0 1\nCamera "perspective" "float fov" 40\nFilm "image" "integer xresolution" 96 "integer yresolution" 64 "string filename" "o.pfm"\n'
                     'Sampler "halton" "integer pixelsamples" 4\nIntegrator "path" "integer maxdepth" 4\nWorldBegin\nLightSource "infinite" "rgb L" [0.8 0.9 1]\n'
                     'LightSource "point" "rgb I" [10 10 10] "point from" [0 -1 2]\nMaterial "matte" "rgb Kd" [0.6 0.5 0.4]\n'
                     'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-2 -2 0 2 -2 0 2 2 0 -2 2 0]\n'
                     'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [-0.5 0 0.2 0.5 0 0.2 0 0 1.2]\nWorldEnd\n')
    digests = []
    for flags in (["--device", "0"], ["--devices", "0,0"]):
        out = tmp_path / ("a" + "".join(flags).replace(",", "_") + ".pfm")
        r = subprocess.run([exe, "--quiet", "--outfile", str(out)] + flags + [str(scene)], capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
        assert r.returncode == 0, r.stderr
        digests.append(hashlib.sha256(out.read_bytes()).hexdigest())
    assert digests[0] == digests[1]
