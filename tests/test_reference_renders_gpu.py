"""-m gpu: the PRODUCT against output of the reference itself (for the five delta-light scenes PIXEL FOR PIXEL: at the reference's 128 spp the device's path integrator at maxdepth 1
draws the same camera samples and evaluates the same light term as the reference's Whitted render) — the ten scenes of tests/test_reference_renders.py rendered by libpbrt_hip.so on the MI355X and held
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


# ---- the same scenes as scene-description TEXT through the front end (pbrt_hip_render): parser, CTM / attribute stacks, `blackbody` parameters, ObjectBegin / ObjectInstance,
# the orthographic camera's directive order.  The text is written here from the numbers of tests/reference_scenes.py (only the Integrator line differs from the reference's files:
# the hot path is the path integrator, at maxdepth 1 it is Whitted's direct term for these scenes).
def _nums(a):
    return " ".join(repr(float(v)) for v in np.asarray(a, np.float32).ravel())


CUBE_TXT = f'Shape "trianglemesh" "point P" [{_nums(R.CUBE_P)}] "float st" [{_nums(R.CUBE_ST)}] "integer indices" [{" ".join(str(int(i)) for i in R.CUBE_IDX)}]'


def _floor_txt(tex1):
    return (f'AttributeBegin\n Translate 0 0 -1\n Texture "checks" "spectrum" "checkerboard" "float uscale" [24] "float vscale" [24] "rgb tex1" [{tex1} {tex1} {tex1}] "rgb tex2" [.8 .8 .8]\n'
            f' Material "matte" "texture Kd" "checks"\n Shape "trianglemesh" "point P" [{_nums(R.quad(20.0))}] "float st" [{_nums(R.QUAD_ST)}] "integer indices" [0 1 2 0 2 3]\nAttributeEnd\n')


def _head(lookat, camera, xres, yres, spp, extra=""):
    return (f'LookAt {lookat}\n{extra}Camera {camera}\nSampler "halton" "integer pixelsamples" {spp}\nIntegrator "path" "integer maxdepth" 1\n'
            f'Film "image" "string filename" "out.pfm" "integer xresolution" [{xres}] "integer yresolution" [{yres}]\nWorldBegin\n')


SKY_AND_SUN = 'LightSource "infinite" "rgb L" [.4 .45 .5]\nLightSource "distant" "point from" [ -30 40 100 ] "blackbody L" [3000 1.5]\n'
RED_CUBE = f'AttributeBegin\n Rotate 45 0 0 1\n Material "matte" "rgb Kd" [.2 .01 .01]\n {CUBE_TXT}\nAttributeEnd\n'
SCENE_TEXTS = {
    "lights_distant": (lambda spp: _head("0 5 3  0 0 0  0 0 1", '"perspective" "float fov" 90', 400, 400, spp) +
                       'LightSource "distant" "point from" [ -5 0 5 ] "point to" [0 0 0] "blackbody L" [4500 1.5]\n' + RED_CUBE + _floor_txt(.3) + "WorldEnd\n", 128, False),
    "cameras_orthographic": (lambda spp: _head("0 10 10  0 0 0  0 0 1", '"orthographic"', 400, 400, spp) + SKY_AND_SUN + "Scale 0.25 0.25 0.25\n" + RED_CUBE + _floor_txt(.1) + "WorldEnd\n",
                             64, True),
    "objects_instances": (lambda spp: _head("0 7 15  0 0 0  0 0 1", '"perspective" "float fov" 45', 400, 400, spp, extra="Translate 0 -1 0\n") + SKY_AND_SUN +
                          f'Material "matte" "rgb Kd" [.8 .1 .01]\nObjectBegin "cube"\n {CUBE_TXT}\nObjectEnd\n' +
                          "".join(f'AttributeBegin\n{"" if k == 0 else f" Rotate {36 * k} 0 0 1" + chr(10)} Translate 0 5 0\n Rotate 45 0 0 1\n ObjectInstance "cube"\nAttributeEnd\n' for k in range(10)) +
                          _floor_txt(.1) + "WorldEnd\n", 64, True),
}


CHECKS_ALPHA = ('AttributeBegin\n Texture "alpha" "float" "dots" "float inside" 1 "float outside" 0 "float uscale" 10 "float vscale" 10\n Rotate 135 0 0 1\n'
                f' Material "matte" "rgb Kd" [.2 .01 .01]\n {CUBE_TXT} "texture alpha" "alpha"\nAttributeEnd\n')
POINT = 'LightSource "point" "rgb I" [.4 .45 .5] "point from" [-5 0 5] "rgb scale" [200 200 200]\n'
HEAD_LIGHTS = lambda spp: _head("0 5 3  0 0 0  0 0 1", '"perspective" "float fov" 90', 400, 400, spp)
SCENE_TEXTS.update({
    # the dots texture with the file's own "inside" 1 / "outside" 0 (the front end swaps them as the reference does, quirk B13), as alpha mask of the cube
    "triangles_alpha_mask": (lambda spp: HEAD_LIGHTS(spp) + POINT + CHECKS_ALPHA + _floor_txt(.3) + "WorldEnd\n", 128, False),
    "lights_point": (lambda spp: HEAD_LIGHTS(spp) + POINT + RED_CUBE + _floor_txt(.3) + "WorldEnd\n", 128, False),
    # "float conedelta" 20 is what the reference's file says; nothing reads it (spot.rs:156 asks for "conedeltaangle"): the default rim of 5 degrees applies
    "lights_spot": (lambda spp: HEAD_LIGHTS(spp) + 'LightSource "spot" "rgb I" [.4 .45 .5] "point from" [-5 0 5] "point to" [0 0 0] "rgb scale" [200 200 200] "float coneangle" 25 "float conedelta" 20\n' +
                    RED_CUBE + _floor_txt(.3) + "WorldEnd\n", 128, False),
    # the goniometric map is written next to the scene as a PNG (from the fixture's pixels) and read back by the front end's own PNG decoder
    "lights_goniometric": (lambda spp: HEAD_LIGHTS(spp) + 'AttributeBegin\n Translate -5 0 5\n Rotate 135 1 0 0\n Rotate 60 0 1 0\n LightSource "goniometric" "rgb I" [.4 .45 .5] "rgb scale" [200 200 200] '
                           '"float fov" 45 "string mapname" "gonio.png"\nAttributeEnd\n' + RED_CUBE + _floor_txt(.3) + "WorldEnd\n", 128, False),
    "cameras_environment": (lambda spp: _head("0 0 1  0 1 0  0 0 1", '"environment"', 800, 400, spp) + SKY_AND_SUN + 'Material "matte" "rgb Kd" [.8 .1 .01]\n' +
                            "".join(f'AttributeBegin\n{"" if k == 0 else f" Rotate {36 * k} 0 0 1" + chr(10)} Translate 0 5 0\n Rotate 45 0 0 1\n {CUBE_TXT}\nAttributeEnd\n' for k in range(10)) +
                            _floor_txt(.1) + "WorldEnd\n", 64, True),
})


def _write_png(path, rgb_u8):
    import struct
    import zlib
    h, w, _ = rgb_u8.shape
    raw = b"".join(b"\x00" + rgb_u8[y].tobytes() for y in range(h))
    chunk = lambda t, c: struct.pack(">I", len(c)) + t + c + struct.pack(">I", zlib.crc32(t + c) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


@pytest.mark.parametrize("name", sorted(SCENE_TEXTS))
def test_front_end_renders_the_references_scene_text_like_the_reference(tmp_path, name):
    import subprocess
    import driver_scene as ds
    make, spp, noisy = SCENE_TEXTS[name]
    (tmp_path / "scene.pbrt").write_text(make(spp))
    if name == "lights_goniometric":
        import os
        _write_png(str(tmp_path / "gonio.png"), np.load(os.path.join(R.HERE, "golden", "ref_renders", "image_goniometric-upward-downward.npz"))["rgb"])
    r = subprocess.run([ds.RENDER_BIN, "--quiet", str(tmp_path / "scene.pbrt")], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    img = ds.read_pfm(str(tmp_path / "out.pfm"))
    check_against_reference(img, dict(render={"triangles_alpha_mask": "shapes_triangles-alpha-mask"}.get(name, name)), noisy)
