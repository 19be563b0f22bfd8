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


def test_hlbvh_scene_file_is_built_on_the_device_and_renders_the_same_image(tmp_path, host):
    """Accelerator "bvh" "string splitmethod" "hlbvh": the front end has the tree made by the device builder (csrc/bvh_device.hip).  Closest hits do not depend on the
    tree (no exact-t ties in this scene), so the image equals the oracle's, which traces its SAH tree."""
    path = ds.write_files(str(tmp_path), filter_line=ds.FILTERS["box"][0], strategy="uniform")
    text = open(path).read().replace('Accelerator "bvh" "integer maxnodeprims" 2', 'Accelerator "bvh" "string splitmethod" "hlbvh" "integer maxnodeprims" 2')
    assert "hlbvh" in text
    open(path, "w").write(text)
    r = subprocess.run([ds.RENDER_BIN, path], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    img = ds.read_pfm(str(tmp_path / "scene.pfm"))
    ref, st = oracle_image(host, "box", 0)
    assert (img.view(np.uint32) == ref.view(np.uint32)).all(), f"{(img != ref).sum()} differing values"
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


def test_object_instances_in_a_scene_file(tmp_path, host):
    """ObjectBegin / ObjectInstance through the parser and Api: same bits as the oracle fed the same instances call by call."""
    import scenes
    Pq = np.array([[-0.5, -0.5, 0], [0.5, -0.5, 0], [0.5, 0.5, 0.3], [-0.5, 0.5, 0.3]], np.float32)
    text = f"""LookAt 0 -6 2.5  0 0 0.3  0 0 1
Camera "perspective" "float fov" 45
Film "image" "integer xresolution" 40 "integer yresolution" 32 "string filename" "i.pfm"
Sampler "halton" "integer pixelsamples" 4
Integrator "path" "integer maxdepth" 3
WorldBegin
LightSource "infinite" "rgb L" [0.7 0.8 0.9]
LightSource "point" "point from" [0 -1 3] "rgb I" [20 20 20]
Material "matte" "rgb Kd" [0.5 0.5 0.5]
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-3 -3 -0.5 3 -3 -0.5 3 3 -0.5 -3 3 -0.5]
ObjectBegin "tile"
  Material "matte" "rgb Kd" [0.8 0.3 0.2] "float sigma" 20
  Rotate 20 1 0 0
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [{ds.fl(Pq)}]
ObjectEnd
ObjectBegin "single"
  Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 0.6 0 0.2 0 0.6 0.4]
ObjectEnd
AttributeBegin
  Translate -1.5 0 0.2
  ObjectInstance "tile"
  Translate 1.5 0.5 0.4
  Scale 1.5 1 -1
  ObjectInstance "tile"
  ObjectInstance "single"
AttributeEnd
ObjectInstance "single"
WorldEnd
"""
    (tmp_path / "i.pbrt").write_text(text)
    r = subprocess.run([ds.RENDER_BIN, "--quiet", str(tmp_path / "i.pbrt")], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    img = ds.read_pfm(str(tmp_path / "i.pfm"))

    I = (np.eye(4, dtype=np.float32).reshape(16),) * 2
    mul = host.compose
    set_libm_mode(1)
    try:
        with OracleScene() as o:
            w2c = mul(I, host.look_at([0, -6, 2.5], [0, 0, 0.3], [0, 0, 1]))
            o.add_light_infinite((0.7, 0.8, 0.9))
            o.add_light_point(np.float32([20, 20, 20]), host.point_position(I[0], I[1], [0, -1, 3]))
            grey = o.add_material_matte((0.5, 0.5, 0.5), 0.0)
            o.add_mesh(np.float32([[-3, -3, -0.5], [3, -3, -0.5], [3, 3, -0.5], [-3, 3, -0.5]]), [0, 1, 2, 0, 2, 3], grey)
            tile = o.object_begin()
            red = o.add_material_matte((0.8, 0.3, 0.2), 20.0)
            t = mul(I, host.rotate(20, [1, 0, 0]))
            o.add_mesh(host.transform_points(t[0], Pq), [0, 1, 2, 0, 2, 3], red, swaps_handedness=host.swaps_handedness(t[0]))
            o.object_end()
            single = o.object_begin()
            o.add_mesh(np.float32([[0, 0, 0], [0.6, 0, 0.2], [0, 0.6, 0.4]]), [0, 1, 2], grey)   # Material reverts at ObjectEnd (attribute pop)
            o.object_end()
            a = mul(I, host.translate([-1.5, 0, 0.2]))
            o.add_instance(tile, a[0], a[1])
            b = mul(mul(a, host.translate([1.5, 0.5, 0.4])), host.scale([1.5, 1, -1]))
            o.add_instance(tile, b[0], b[1]); o.add_instance(single, b[0], b[1])
            o.add_instance(single, I[0], I[1])
            o.set_camera_perspective(host.perspective_raster_to_camera(45.0, 40, 32), w2c[1])
            cb, table, sb = host.film_box(40, 32)
            o.set_film(40, 32, cb, (0.5, 0.5), table)
            o.set_sampler(0, 4, sb)
            o.build_accel(0, 4)
            xyz, wt, st = o.render_path(max_depth=3, light_strategy=2, pixel_bounds=sb)
            ref = o.film_to_rgb(xyz, wt).reshape(32, 40, 3)
    finally:
        set_libm_mode(0)
    assert (img.view(np.uint32) == ref.view(np.uint32)).all(), f"{(img != ref).sum()} values differ"


def test_materials_in_a_scene_file(tmp_path, host):
    """Material / MakeNamedMaterial directives for mirror, plastic, glass, metal, uber with shape-level overrides: same bits as the
    oracle given the same constructor arguments."""
    quad = "[-1 -1 0 1 -1 0 1 1 0 -1 1 0]"
    text = f"""LookAt 0 -5 2  0 0 0.4  0 0 1
Camera "perspective" "float fov" 50
Film "image" "integer xresolution" 36 "integer yresolution" 28 "string filename" "m.pfm"
Sampler "halton" "integer pixelsamples" 8
Integrator "path" "integer maxdepth" 6
WorldBegin
LightSource "infinite" "rgb L" [0.8 0.9 1.0]
LightSource "distant" "point from" [1 -1 2] "point to" [0 0 0] "rgb L" [2 2 2]
MakeNamedMaterial "gold" "string type" "metal" "rgb eta" [0.14 0.37 1.44] "rgb k" [3.98 2.38 1.6] "float roughness" 0.1
MakeNamedMaterial "wet" "string type" "uber" "rgb Kd" [0.2 0.3 0.2] "rgb Ks" [0.3 0.3 0.3] "rgb Kr" [0.1 0.1 0.1] "float index" 1.33 "rgb opacity" [0.8 0.8 0.8]
AttributeBegin
  Material "plastic" "rgb Kd" [0.4 0.4 0.5] "float roughness" 0.2
  Scale 2.5 2.5 1
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" {quad}
AttributeEnd
AttributeBegin
  NamedMaterial "gold"
  Translate -1.2 0.3 0.6
  Rotate 60 1 0 0
  Scale 0.6 0.6 0.6
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" {quad}
AttributeEnd
AttributeBegin
  Material "glass" "float eta" 1.4
  Translate 0 -0.4 0.7
  Rotate 75 1 0 0
  Scale 0.5 0.5 0.5
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" {quad}
  Translate 0 0 -0.6
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" {quad} "float uroughness" 0.2 "float vroughness" 0.2
AttributeEnd
AttributeBegin
  NamedMaterial "wet"
  Translate 1.3 0.2 0.5
  Rotate 80 1 0.2 0
  Scale 0.6 0.6 0.6
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" {quad}
  Material "mirror"
  Translate 0 0 -0.8
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" {quad} "rgb Kr" [0.5 0.6 0.7]
AttributeEnd
WorldEnd
"""
    (tmp_path / "m.pbrt").write_text(text)
    r = subprocess.run([ds.RENDER_BIN, "--quiet", str(tmp_path / "m.pbrt")], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    img = ds.read_pfm(str(tmp_path / "m.pfm"))
    Q = np.float32([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]]); QI = [0, 1, 2, 0, 2, 3]
    I = (np.eye(4, dtype=np.float32).reshape(16),) * 2
    mul = host.compose

    def put(o, t, mat):
        o.add_mesh(host.transform_points(t[0], Q), QI, mat, swaps_handedness=host.swaps_handedness(t[0]))
    set_libm_mode(1)
    try:
        with OracleScene() as o:
            w2c = mul(I, host.look_at([0, -5, 2], [0, 0, 0.4], [0, 0, 1]))
            o.add_light_infinite((0.8, 0.9, 1.0))
            o.add_light_distant(np.float32([2, 2, 2]), host.distant_direction(I[0], [1, -1, 2], [0, 0, 0]))
            put(o, mul(I, host.scale([2.5, 2.5, 1])), o.add_material_plastic((0.4, 0.4, 0.5), (0.25, 0.25, 0.25), 0.2, True))
            t = mul(mul(mul(I, host.translate([-1.2, 0.3, 0.6])), host.rotate(60, [1, 0, 0])), host.scale([0.6, 0.6, 0.6]))
            put(o, t, o.add_material_metal((0.14, 0.37, 1.44), (3.98, 2.38, 1.6), 0.1, 0.1, True))
            t = mul(mul(mul(I, host.translate([0, -0.4, 0.7])), host.rotate(75, [1, 0, 0])), host.scale([0.5, 0.5, 0.5]))
            # the file says "float eta" 1.4, which the reference never reads (quirk B14: glass.rs:158 looks up a float texture NAMED "eta", then the parameter "index"): 1.5
            put(o, t, o.add_material_glass((1, 1, 1), (1, 1, 1), 0.0, 0.0, 1.5, True))
            t = mul(t, host.translate([0, 0, -0.6]))
            put(o, t, o.add_material_glass((1, 1, 1), (1, 1, 1), 0.2, 0.2, 1.5, True))
            t = mul(mul(mul(I, host.translate([1.3, 0.2, 0.5])), host.rotate(80, [1, 0.2, 0])), host.scale([0.6, 0.6, 0.6]))
            put(o, t, o.add_material_uber((0.2, 0.3, 0.2), (0.3, 0.3, 0.3), (0.1, 0.1, 0.1), (0, 0, 0), (0.8, 0.8, 0.8), 0.1, 0.1, 1.33, True))
            t = mul(t, host.translate([0, 0, -0.8]))
            put(o, t, o.add_material_mirror((0.5, 0.6, 0.7)))
            o.set_camera_perspective(host.perspective_raster_to_camera(50.0, 36, 28), w2c[1])
            cb, table, sb = host.film_box(36, 28)
            o.set_film(36, 28, cb, (0.5, 0.5), table)
            o.set_sampler(0, 8, sb)
            o.build_accel(0, 4)
            xyz, wt, st = o.render_path(max_depth=6, light_strategy=2, pixel_bounds=sb)
            ref = o.film_to_rgb(xyz, wt).reshape(28, 36, 3)
    finally:
        set_libm_mode(0)
    assert (img.view(np.uint32) == ref.view(np.uint32)).all(), f"{(img != ref).sum()} values differ"


def test_mix_substrate_translucent_and_folded_textures_in_a_scene_file(tmp_path, host):
    quad = "[-1 -1 0 1 -1 0 1 1 0 -1 1 0]"
    text = f"""LookAt 0 -5 2  0 0 0.4  0 0 1
Camera "perspective" "float fov" 50
Film "image" "integer xresolution" 32 "integer yresolution" 24 "string filename" "x.pfm"
Sampler "halton" "integer pixelsamples" 8
Integrator "path" "integer maxdepth" 5
WorldBegin
LightSource "infinite" "rgb L" [0.8 0.9 1.0]
LightSource "spot" "point from" [0 -2 3] "point to" [0 0 0] "rgb I" [30 30 30] "float coneangle" 40 "float conedeltaangle" 15
Texture "a" "color" "constant" "rgb value" [0.8 0.6 0.4]
Texture "b" "color" "constant" "rgb value" [0.5 0.5 1.0]
Texture "ab" "color" "scale" "texture tex1" "a" "texture tex2" "b"
Texture "half" "float" "constant" "float value" 0.25
Texture "blend" "color" "mix" "texture tex1" "a" "rgb tex2" [0.1 0.9 0.1] "texture amount" "half"
MakeNamedMaterial "base" "string type" "matte" "texture Kd" "ab"
MakeNamedMaterial "coat" "string type" "mirror" "rgb Kr" [0.8 0.8 0.8]
MakeNamedMaterial "both" "string type" "mix" "string namedmaterial1" "base" "string namedmaterial2" "coat" "rgb amount" [0.6 0.6 0.6]
AttributeBegin
  Material "substrate" "texture Kd" "blend" "rgb Ks" [0.2 0.2 0.2] "float uroughness" 0.05
  Scale 2.5 2.5 1
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" {quad}
AttributeEnd
AttributeBegin
  NamedMaterial "both"
  Translate -1.0 0.3 0.7
  Rotate 70 1 0 0
  Scale 0.7 0.7 0.7
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" {quad}
AttributeEnd
AttributeBegin
  Material "translucent" "rgb Kd" [0.5 0.5 0.3] "rgb transmit" [0.7 0.7 0.7]
  Translate 1.0 0 0.8
  Rotate 85 1 0 0.3
  Scale 0.7 0.7 0.7
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" {quad}
AttributeEnd
WorldEnd
"""
    (tmp_path / "x.pbrt").write_text(text)
    r = subprocess.run([ds.RENDER_BIN, "--quiet", str(tmp_path / "x.pbrt")], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    img = ds.read_pfm(str(tmp_path / "x.pfm"))
    Q = np.float32([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]]); QI = [0, 1, 2, 0, 2, 3]
    I = (np.eye(4, dtype=np.float32).reshape(16),) * 2
    mul = host.compose
    f32 = np.float32

    def put(o, t, mat):
        o.add_mesh(host.transform_points(t[0], Q), QI, mat, swaps_handedness=host.swaps_handedness(t[0]))
    set_libm_mode(1)
    try:
        with OracleScene() as o:
            w2c = mul(I, host.look_at([0, -5, 2], [0, 0, 0.4], [0, 0, 1]))
            o.add_light_infinite((0.8, 0.9, 1.0))
            o.add_light_spot(f32([30, 30, 30]), *host.spot(I, [0, -2, 3], [0, 0, 0], 40.0, 15.0))
            a, b = f32([0.8, 0.6, 0.4]), f32([0.5, 0.5, 1.0])
            ab = a * b
            blend = (f32(1.0) - f32(0.25)) * a + f32(0.25) * f32([0.1, 0.9, 0.1])
            put(o, mul(I, host.scale([2.5, 2.5, 1])), o.add_material_substrate(blend, (0.2, 0.2, 0.2), 0.05, 0.1, True))
            t = mul(mul(mul(I, host.translate([-1.0, 0.3, 0.7])), host.rotate(70, [1, 0, 0])), host.scale([0.7, 0.7, 0.7]))
            put(o, t, o.add_material_mix(o.add_material_matte(ab, 0.0), o.add_material_mirror((0.8, 0.8, 0.8)), (0.6, 0.6, 0.6)))
            t = mul(mul(mul(I, host.translate([1.0, 0, 0.8])), host.rotate(85, [1, 0, 0.3])), host.scale([0.7, 0.7, 0.7]))
            put(o, t, o.add_material_translucent((0.5, 0.5, 0.3), (0.25,) * 3, (0.5,) * 3, (0.7, 0.7, 0.7), 0.1, True))
            o.set_camera_perspective(host.perspective_raster_to_camera(50.0, 32, 24), w2c[1])
            cb, table, sb = host.film_box(32, 24)
            o.set_film(32, 24, cb, (0.5, 0.5), table)
            o.set_sampler(0, 8, sb)
            o.build_accel(0, 4)
            xyz, wt, st = o.render_path(max_depth=5, light_strategy=2, pixel_bounds=sb)
            ref = o.film_to_rgb(xyz, wt).reshape(24, 32, 3)
    finally:
        set_libm_mode(0)
    assert (img.view(np.uint32) == ref.view(np.uint32)).all(), f"{(img != ref).sum()} values differ"


def test_textured_pbrt_file_renders_the_oracle_image(tmp_path, host):
    """Texture "imagemap" (PNG with its default gamma, trilinear TGA float map, scaled PFM with black wrap and uv scale), "scale" and
    "mix" textures over them, MatteMaterial "texture Kd": the C++ front end decodes the files and feeds the library; the oracle gets the
    same texels from this test.  Images must be equal bit for bit."""
    import image_files as imf
    rng = np.random.default_rng(12)
    a8 = rng.integers(0, 256, (24, 40, 3), dtype=np.uint8)                     # NPOT
    m8 = (np.indices((16, 16)).sum(0) * 8 % 256).astype(np.uint8)             # grey ramp
    hdr = rng.uniform(0.0, 2.0, (8, 8, 3)).astype(np.float32)
    imf.write_png(str(tmp_path / "a.png"), a8)
    imf.write_tga(str(tmp_path / "m.tga"), m8, grey=True, rle=True)
    imf.write_pfm(str(tmp_path / "h.pfm"), hdr)
    sky = rng.uniform(0.2, 1.0, (8, 16, 3)).astype(np.float32); sky[1, 3] = (30.0, 25.0, 20.0)
    imf.write_pfm(str(tmp_path / "sky.pfm"), sky)
    Q = np.array([[-4, -4, 0], [4, -4, 0], [4, 4, 0], [-4, 4, 0]], np.float32)
    UVQ = np.array([[0, 0], [2, 0], [2, 2], [0, 2]], np.float32)
    W = np.array([[-2, 3, 0], [2, 3, 0], [2, 3, 2.5], [-2, 3, 2.5]], np.float32)
    UVW = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    res, spp, depth = 48, 4, 3
    text = f"""LookAt 0 -6 1.5  0 0 0.5  0 0 1
Camera "perspective" "float fov" [40]
Film "image" "integer xresolution" [{res}] "integer yresolution" [{res}] "string filename" "tex.pfm"
Sampler "halton" "integer pixelsamples" {spp}
PixelFilter "box"
Integrator "path" "integer maxdepth" {depth} "string lightsamplestrategy" "uniform"
WorldBegin
AttributeBegin
  Rotate 40 0 0 1
  LightSource "infinite" "rgb L" [1 0.9 0.8] "rgb scale" [0.5 0.5 0.5] "string mapname" "sky.pfm"
AttributeEnd
Texture "wood" "color" "imagemap" "string filename" "a.png"
Texture "tint" "color" "scale" "texture tex1" "wood" "rgb tex2" [0.9 0.6 0.4]
Texture "amt" "float" "imagemap" "string filename" "m.tga" "bool trilinear" "true" "string wrap" "clamp" "bool gamma" "false"
Texture "blend" "color" "mix" "texture tex1" "tint" "rgb tex2" [0.1 0.2 0.8] "texture amount" "amt"
Texture "cut" "float" "checkerboard" "float uscale" 4 "float vscale" 4 "string aamode" "none"
Texture "hdr" "color" "imagemap" "string filename" "h.pfm" "float scale" 0.5 "float uscale" 2 "float vscale" 2 "string wrap" "black"
Material "matte" "texture Kd" "blend"
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [{ds.fl(Q)}] "float uv" [{ds.fl(UVQ)}]
Material "matte" "texture Kd" "hdr" "float sigma" 20
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [{ds.fl(W)}] "float st" [{ds.fl(UVW)}] "texture alpha" "amt" "texture shadowalpha" "cut"
Texture "bumps" "float" "scale" "texture tex1" "amt" "float tex2" 0.05
Material "plastic" "texture Kd" "wood" "texture Ks" "tint" "float roughness" 0.05 "texture bumpmap" "bumps"
Shape "trianglemesh" "integer indices" [0 1 2] "point P" [-3 1 0.01  -1 1 0.01  -2 2.5 1.5] "float uv" [0 0 1 0 0.5 1]
MakeNamedMaterial "glossy" "string type" "substrate" "texture Kd" "blend" "rgb Ks" [0.05 0.05 0.05] "float uroughness" 0.1 "float vroughness" 0.2
NamedMaterial "glossy"
Shape "trianglemesh" "integer indices" [0 1 2] "point P" [1 1 0.01  3 1 0.01  2 2.5 1.5] "float uv" [0 0 1 0 0.5 1]
Material "mirror" "texture Kr" "wood"
Shape "trianglemesh" "integer indices" [0 1 2] "point P" [-1 -1 0.01  1 -1 0.01  0 0 1.2]
Texture "checks" "color" "checkerboard" "float uscale" 6 "float vscale" 6 "texture tex1" "wood" "rgb tex2" [0.1 0.1 0.1]
Texture "spots" "color" "dots" "float uscale" 3 "float vscale" 3 "texture inside" "checks" "rgb outside" [0.7 0.2 0.2]
Material "matte" "texture Kd" "spots"
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-4 -4 0.005  -2 -4 0.005  -2 -2 0.005  -4 -2 0.005] "float uv" [0 0 1 0 1 1 0 1]
TransformBegin
  Scale 2 2 2
  Rotate 20 0 0 1
  Texture "stone" "color" "marble" "float scale" 2 "float variation" 0.3 "integer octaves" 6
  Texture "cells" "color" "checkerboard" "integer dimension" 3 "texture tex1" "stone" "rgb tex2" [0.05 0.3 0.05]
TransformEnd
Material "matte" "texture Kd" "cells"
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [2 -4 0.005  4 -4 0.005  4 -2 0.005  2 -2 0.005]
Material "translucent" "texture Kd" "wood" "rgb Ks" [0.1 0.1 0.1] "rgb reflect" [0.4 0.4 0.4] "rgb transmit" [0.6 0.6 0.6] "texture roughness" "amt"
Shape "trianglemesh" "integer indices" [0 1 2] "point P" [-1.8 -2.5 0.3  -0.8 -2.5 0.3  -1.3 -2.5 1.3] "float uv" [0 0 1 0 0.5 1]
MakeNamedMaterial "m1" "string type" "plastic" "texture Kd" "wood" "rgb Ks" [0.2 0.2 0.2]
MakeNamedMaterial "m2" "string type" "matte" "texture Kd" "spots"
Material "mix" "string namedmaterial1" "m1" "string namedmaterial2" "m2" "rgb amount" [0.3 0.3 0.3]
Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0.8 -2.5 0.3  1.8 -2.5 0.3  1.3 -2.5 1.3] "float uv" [0 0 1 0 0.5 1]
WorldEnd
"""
    (tmp_path / "tex.pbrt").write_text(text)
    r = subprocess.run([ds.RENDER_BIN, "--quiet", str(tmp_path / "tex.pbrt")], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    img = ds.read_pfm(str(tmp_path / "tex.pfm"))

    def capture(s):
        wood = s.add_texture_imagemap(s.add_mipmap(a8.astype(np.float32) / np.float32(255.0), gamma=True))
        tint = s.add_texture_scale(wood, s.add_texture_constant((0.9, 0.6, 0.4)))
        grey = np.repeat(m8[..., None], 3, axis=2).astype(np.float32) / np.float32(255.0)
        amt = s.add_texture_imagemap(s.add_mipmap(grey, as_float=True, trilinear=True, wrap="clamp", gamma=False))
        blend = s.add_texture_mix(tint, s.add_texture_constant((0.1, 0.2, 0.8)), amt)
        cut = s.add_texture_checkerboard(s.add_texture_constant(1.0), s.add_texture_constant(0.0), su=4.0, sv=4.0, aa="none")
        hd = s.add_texture_imagemap(s.add_mipmap(hdr, scale=0.5, wrap="black"), su=2.0, sv=2.0)
        ident2 = (np.eye(4, dtype=np.float32).reshape(16),) * 2
        lt = host.compose(ident2, host.rotate(40.0, [0, 0, 1]))
        s.add_light_infinite_map(np.float32([1.0, 0.9, 0.8]) * np.float32([0.5, 0.5, 0.5]), sky, lt[0], lt[1])
        s.add_mesh(Q, [0, 1, 2, 0, 2, 3], s.add_material_matte_tex(blend, 0.0), UV=UVQ)
        s.add_mesh(W, [0, 1, 2, 0, 2, 3], s.add_material_matte_tex(hd, 20.0), UV=UVW)
        s.set_last_mesh_alpha_textures(amt, cut)
        tri_uv = np.array([[0, 0], [1, 0], [0.5, 1]], np.float32)
        pl = s.add_material_plastic((1, 1, 1), (1, 1, 1), 0.05, True); s.set_material_texture(pl, "Kd", wood); s.set_material_texture(pl, "Ks", tint)
        s.set_material_bump(pl, s.add_texture_scale(amt, s.add_texture_constant(0.05)))
        s.add_mesh(np.array([[-3, 1, 0.01], [-1, 1, 0.01], [-2, 2.5, 1.5]], np.float32), [0, 1, 2], pl, UV=tri_uv)
        sb_ = s.add_material_substrate((1, 1, 1), (0.05, 0.05, 0.05), 0.1, 0.2, True); s.set_material_texture(sb_, "Kd", blend)
        s.add_mesh(np.array([[1, 1, 0.01], [3, 1, 0.01], [2, 2.5, 1.5]], np.float32), [0, 1, 2], sb_, UV=tri_uv)
        mi = s.add_material_mirror((1, 1, 1)); s.set_material_texture(mi, "Kr", wood)
        s.add_mesh(np.array([[-1, -1, 0.01], [1, -1, 0.01], [0, 0, 1.2]], np.float32), [0, 1, 2], mi)
        checks = s.add_texture_checkerboard(wood, s.add_texture_constant((0.1, 0.1, 0.1)), su=6.0, sv=6.0)
        spots = s.add_texture_dots(s.add_texture_constant((0.7, 0.2, 0.2)), checks, su=3.0, sv=3.0)  # quirk B13: the file's "inside" lands outside the dots (dots.rs:61-66)
        s.add_mesh(np.array([[-4, -4, 0.005], [-2, -4, 0.005], [-2, -2, 0.005], [-4, -2, 0.005]], np.float32), [0, 1, 2, 0, 2, 3], s.add_material_matte_tex(spots, 0.0),
                   UV=np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32))
        ident = (np.eye(4, dtype=np.float32).reshape(16),) * 2
        ctm = host.compose(host.compose(ident, host.scale([2, 2, 2])), host.rotate(20, [0, 0, 1]))[0]   # the CTM at the Texture directive = "tex2world"
        stone = s.add_texture_marble(ctm, 0.5, 6, 2.0, 0.3)
        cells = s.add_texture_checkerboard3d(stone, s.add_texture_constant((0.05, 0.3, 0.05)), ctm)
        s.add_mesh(np.array([[2, -4, 0.005], [4, -4, 0.005], [4, -2, 0.005], [2, -2, 0.005]], np.float32), [0, 1, 2, 0, 2, 3], s.add_material_matte_tex(cells, 0.0))
        leaf = s.add_material_translucent((1, 1, 1), (0.1, 0.1, 0.1), (0.4, 0.4, 0.4), (0.6, 0.6, 0.6), 0.1, True)
        s.set_material_texture(leaf, "Kd", wood); s.set_material_float_texture(leaf, "roughness", amt)
        s.add_mesh(np.array([[-1.8, -2.5, 0.3], [-0.8, -2.5, 0.3], [-1.3, -2.5, 1.3]], np.float32), [0, 1, 2], leaf, UV=tri_uv)
        m1 = s.add_material_plastic((1, 1, 1), (0.2, 0.2, 0.2), 0.1, True); s.set_material_texture(m1, "Kd", wood)
        m2 = s.add_material_matte_tex(spots, 0.0)
        s.add_mesh(np.array([[0.8, -2.5, 0.3], [1.8, -2.5, 0.3], [1.3, -2.5, 1.3]], np.float32), [0, 1, 2], s.add_material_mix(m1, m2, (0.3, 0.3, 0.3)), UV=tri_uv)
        w2c, c2w = host.look_at((0, -6, 1.5), (0, 0, 0.5), (0, 0, 1))
        s.set_camera_perspective(host.perspective_raster_to_camera(40.0, res, res), c2w)
        cb, table, sb = host.film_box(res, res)
        s.set_film(res, res, cb, (0.5, 0.5), table)
        s.set_sampler(0, spp, sb)
        s.build_accel(0, 4)
        return sb
    set_libm_mode(1)
    try:
        with OracleScene() as o:
            sb = capture(o)
            xyz, wt, _ = o.render_path(max_depth=depth, light_strategy=0, pixel_bounds=sb)
            ref = o.film_to_rgb(xyz, wt).reshape(res, res, 3)
    finally:
        set_libm_mode(0)
    assert img.shape == ref.shape and float(img.mean()) > 0.05
    assert (img.view(np.uint32) == ref.view(np.uint32)).all(), f"{(img != ref).sum()} differing values, max {np.abs(img - ref).max()}"
    with pbrt_hip.Scene() as s:   # and the Python wrapper over the same ABI
        sb = capture(s)
        xyz, wt, _ = s.render_path(max_depth=depth, light_strategy=0, pixel_bounds=sb)
        assert (s.film_to_rgb(xyz, wt).reshape(res, res, 3).view(np.uint32) == ref.view(np.uint32)).all()


@pytest.mark.parametrize("camera", ["orthographic", "environment"])
def test_other_cameras_in_a_scene_file(tmp_path, host, camera):
    """Camera "orthographic" with a screen window and a lens (orthographic_camera.rs:188-244), Camera "environment" (environment_camera.rs:86-104),
    over a textured floor: image equal to the oracle's bit for bit."""
    res, spp, depth = (40, 30), 4, 3
    cam_line = 'Camera "orthographic" "float screenwindow" [-2 2 -1.5 1.5] "float lensradius" 0.03 "float focaldistance" 6' if camera == "orthographic" \
        else 'Camera "environment" "float shutteropen" 0.2 "float shutterclose" 0.1'     # swapped with a warning
    Q = np.array([[-3, -3, 0], [3, -3, 0], [3, 3, 0], [-3, 3, 0]], np.float32)
    UVQ = np.array([[0, 0], [2, 0], [2, 2], [0, 2]], np.float32)
    text = f"""LookAt 2 -5 3  0 0 0.3  0 0 1
{cam_line}
Film "image" "integer xresolution" [{res[0]}] "integer yresolution" [{res[1]}] "string filename" "ortho.pfm"
Sampler "halton" "integer pixelsamples" {spp}
PixelFilter "box"
Integrator "path" "integer maxdepth" {depth} "string lightsamplestrategy" "uniform"
WorldBegin
LightSource "infinite" "rgb L" [0.9 0.9 1.0]
AttributeBegin
  Translate 0.5 -0.5 3
  Rotate 175 1 0 0
  LightSource "projection" "rgb I" [30 30 30] "rgb scale" [0.5 1 1] "float fov" 50 "string mapname" "slide.pfm"
  LightSource "goniometric" "rgb I" [4 3 2]
AttributeEnd
Texture "checks" "color" "checkerboard" "float uscale" 4 "float vscale" 4 "rgb tex1" [0.8 0.7 0.2] "rgb tex2" [0.1 0.1 0.3]
Material "matte" "texture Kd" "checks"
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [{ds.fl(Q)}] "float uv" [{ds.fl(UVQ)}]
Material "plastic" "rgb Kd" [0.6 0.2 0.2]
Shape "trianglemesh" "integer indices" [0 1 2] "point P" [-1 0 0.01  1 0 0.01  0 0.5 1.5]
WorldEnd
"""
    import image_files as imf
    slide = np.random.default_rng(3).uniform(0.0, 1.0, (6, 9, 3)).astype(np.float32)
    imf.write_pfm(str(tmp_path / "slide.pfm"), slide)
    (tmp_path / "ortho.pbrt").write_text(text)
    r = subprocess.run([ds.RENDER_BIN, "--quiet", str(tmp_path / "ortho.pbrt")], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    img = ds.read_pfm(str(tmp_path / "ortho.pfm"))
    set_libm_mode(1)
    try:
        with OracleScene() as s:
            s.add_light_infinite((0.9, 0.9, 1.0))
            ident = (np.eye(4, dtype=np.float32).reshape(16),) * 2
            lt = host.compose(host.compose(ident, host.translate([0.5, -0.5, 3])), host.rotate(175.0, [1, 0, 0]))
            s.add_light_projection(np.float32([30, 30, 30]) * np.float32([0.5, 1, 1]), lt[0], lt[1], 50.0, slide)
            s.add_light_goniometric((4, 3, 2), lt[0], lt[1], None)
            checks = s.add_texture_checkerboard(s.add_texture_constant((0.8, 0.7, 0.2)), s.add_texture_constant((0.1, 0.1, 0.3)), su=4.0, sv=4.0)
            s.add_mesh(Q, [0, 1, 2, 0, 2, 3], s.add_material_matte_tex(checks, 0.0), UV=UVQ)
            s.add_mesh(np.array([[-1, 0, 0.01], [1, 0, 0.01], [0, 0.5, 1.5]], np.float32), [0, 1, 2], s.add_material_plastic((0.6, 0.2, 0.2), (0.25,) * 3, 0.1, True))
            w2c, c2w = host.look_at((2, -5, 3), (0, 0, 0.3), (0, 0, 1))
            if camera == "orthographic":
                s.set_camera_orthographic(host.orthographic_raster_to_camera(res[0], res[1], np.float32([-2, 2, -1.5, 1.5])), c2w, lens_radius=0.03, focal_distance=6.0)
            else:
                s.set_camera_environment(c2w, res[0], res[1], shutter_open=0.1, shutter_close=0.2)
            cb, table, sb = host.film_box(res[0], res[1])
            s.set_film(res[0], res[1], cb, (0.5, 0.5), table)
            s.set_sampler(0, spp, sb)
            s.build_accel(0, 4)
            xyz, wt, _ = s.render_path(max_depth=depth, light_strategy=0, pixel_bounds=sb)
            ref = s.film_to_rgb(xyz, wt).reshape(res[1], res[0], 3)
    finally:
        set_libm_mode(0)
    assert img.shape == ref.shape and float(img.mean()) > 0.05
    assert (img.view(np.uint32) == ref.view(np.uint32)).all(), f"{(img != ref).sum()} differing values, max {np.abs(img - ref).max()}"


def test_output_formats(tmp_path, host):
    """Film "filename" with .exr / .png / .tga (core/src/image_io.rs:225-237): the EXR holds the PFM's floats, the 8-bit files its gamma-encoded bytes."""
    import image_files as imf
    path = ds.write_files(str(tmp_path))
    outs = {}
    for ext in ("pfm", "exr", "png"):
        r = subprocess.run([ds.RENDER_BIN, "--quiet", "--outfile", str(tmp_path / ("o." + ext)), path], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
    ref = ds.read_pfm(str(tmp_path / "o.pfm"))
    r = subprocess.run([ds.RENDER_BIN, "--convert-image", str(tmp_path / "o.exr"), str(tmp_path / "back.pfm")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(ds.read_pfm(str(tmp_path / "back.pfm")).view(np.uint32), ref.view(np.uint32))
    png = imf.read_png_rgb8(str(tmp_path / "o.png"))
    g = np.where(ref <= 0.0031308, 12.92 * ref, 1.055 * np.power(np.maximum(ref, 0), 1 / 2.4) - 0.055)
    want = np.clip(255.0 * g + 0.5, 0, 255).astype(np.uint8)
    assert png.shape == want.shape and np.abs(png.astype(int) - want.astype(int)).max() <= 1
