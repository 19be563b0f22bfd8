"""-m gpu: image textures (MIPMap build on the host side of the library, lookups + ray differentials on the device) against the oracle.
Bit-exact with the oracle in libm mode 1 (log2 / trig evaluated in f64 and rounded, as the device does); the glibc-mode comparison
uses the film tolerance of tests/test_render_gpu.py."""
import numpy as np
import pytest

import pbrt_hip
from oracle_binding import OracleScene, set_libm_mode
from texture_scenes import make_image, probe_points, textured_quad_scene

pytestmark = pytest.mark.gpu


def _bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


@pytest.mark.parametrize("w,h", [(16, 8), (5, 3), (33, 20), (1, 7)])
@pytest.mark.parametrize("wrap", ["repeat", "clamp", "black"])
def test_pyramid_equals_the_oracles(w, h, wrap):
    img = make_image(w, h, seed=w * 100 + h)
    for kw in (dict(), dict(as_float=True, gamma=True, scale=0.7), dict(gamma=True)):
        prod = pbrt_hip.Scene(); orc = OracleScene()
        pp = prod.mipmap_pyramid(prod.add_mipmap(img, wrap=wrap, **kw)); po = orc.mipmap_pyramid(orc.add_mipmap(img, wrap=wrap, **kw))
        assert len(pp) == len(po)
        for a, b in zip(pp, po):
            assert a.shape == b.shape and _bits_equal(a, b)


@pytest.mark.parametrize("trilinear", [True, False])
@pytest.mark.parametrize("wrap", ["repeat", "clamp", "black"])
@pytest.mark.parametrize("as_float", [False, True])
def test_lookups_bit_exact(trilinear, wrap, as_float):
    img = make_image(37, 24, seed=9)
    uv, d = probe_points(4000, 5)
    prod = pbrt_hip.Scene(); orc = OracleScene()
    outs = []
    for sc in (prod, orc):
        mp = sc.add_mipmap(img, as_float=as_float, trilinear=trilinear, wrap=wrap, max_anisotropy=8.0 if not trilinear else 0.0)
        t = sc.add_texture_imagemap(mp, su=1.5, sv=0.75, du=0.1, dv=-0.2)
        k = sc.add_texture_constant((0.9, 0.5, 0.3)); amt = sc.add_texture_imagemap(sc.add_mipmap(make_image(8, 8, seed=2), as_float=True, trilinear=True))
        tex = sc.add_texture_mix(sc.add_texture_scale(t, k), t, amt)
        if sc is orc: set_libm_mode(1)
        try:
            outs.append((sc.texture_eval(t, uv, d), sc.texture_eval(tex, uv, d)))
        finally:
            set_libm_mode(0)
    assert _bits_equal(outs[0][0], outs[1][0])
    assert _bits_equal(outs[0][1], outs[1][1])
    assert np.isfinite(outs[0][0]).all()


def _render_pair(builder, **kw):
    host = pbrt_hip.Host()
    prod = pbrt_hip.Scene(); orc = OracleScene()
    textured_quad_scene(prod, host, builder, **kw); textured_quad_scene(orc, host, builder, **kw)
    return prod, orc


def _tex(kind="noise", w=64, h=64, **mip_kw):
    def build(sc):
        return sc.add_texture_imagemap(sc.add_mipmap(make_image(w, h, seed=4, kind=kind), **mip_kw))
    return build


CASES = {
    "ewa": dict(builder=_tex()),
    "trilinear": dict(builder=_tex(trilinear=True)),
    "npot_clamp_gamma": dict(builder=_tex(w=50, h=37, wrap="clamp", gamma=True)),
    "black_wrap_top_view": dict(builder=_tex(wrap="black"), tilt=False),
    "thin_lens": dict(builder=_tex(), lens_radius=0.05),
    "instanced": dict(builder=_tex(kind="ramp"), instance=True),
    "oren_nayar": dict(builder=_tex(), sigma=20.0),
    "orthographic_camera": dict(builder=_tex(), orthographic=True),                       # differentials = offset origins (orthographic_camera.rs:168-173)
    "orthographic_camera_thin_lens": dict(builder=_tex(), orthographic=True, lens_radius=0.05),
    "environment_camera": dict(builder=_tex(), environment=True),                           # finite-difference differentials (core/src/camera.rs:29-78)
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_textured_matte_film_bit_exact(name):
    kw = dict(CASES[name]); builder = kw.pop("builder")
    prod, orc = _render_pair(builder, **kw)
    set_libm_mode(1)
    try:
        oxyz, owt, ost, _ = orc.render_path_ex(max_depth=3)
    finally:
        set_libm_mode(0)
    gxyz, gwt, gst = prod.render_path(max_depth=3)
    assert _bits_equal(gxyz, oxyz) and _bits_equal(gwt, owt)
    assert gst.regular_rays == ost.regular_rays and gst.shadow_rays == ost.shadow_rays
    assert float(gxyz.mean()) > 0.0


def test_textured_matte_with_general_materials_and_none_veil():
    """The general-BSDF kernel with a textured matte among glass / plastic, seen through a 'none' veil: rays respawned at the veil carry no
    differentials (path.rs:146-150), so the texture behind it is filtered with a zero footprint."""
    def extra(sc):
        gl = sc.add_material_glass((1, 1, 1), (1, 1, 1), 0.0, 0.0, 1.5, True)
        pl = sc.add_material_plastic((0.3, 0.5, 0.2), (0.3, 0.3, 0.3), 0.1, True)
        no = sc.add_material_none()
        S = np.array([[-1.5, -1, 0.1], [-0.8, -1, 0.1], [-1.15, -1, 0.9], [0.8, -1, 0.1], [1.5, -1, 0.1], [1.15, -1, 0.9]], np.float32)
        sc.add_mesh(S[:3], np.array([0, 1, 2], np.uint32), gl)
        sc.add_mesh(S[3:], np.array([0, 1, 2], np.uint32), pl)
        V = np.array([[-0.6, -3, 0.0], [0.6, -3, 0.0], [0.6, -3, 1.0], [-0.6, -3, 1.0]], np.float32)
        sc.add_mesh(V, np.array([0, 1, 2, 0, 2, 3], np.uint32), no)
    prod, orc = _render_pair(_tex(kind="checker", w=128, h=128), extra=extra, res=48)
    set_libm_mode(1)
    try:
        oxyz, owt, ost, _ = orc.render_path_ex(max_depth=4)
    finally:
        set_libm_mode(0)
    gxyz, gwt, gst = prod.render_path(max_depth=4)
    assert _bits_equal(gxyz, oxyz) and _bits_equal(gwt, owt)


def test_filtering_removes_the_moire_and_glibc_mode_is_within_tolerance():
    """A one-texel checkerboard tiled to the horizon: with EWA the far field converges to grey; and against the oracle with glibc's own
    log2f / trig (what the Rust reference links) the film stays within the stated tolerance."""
    prod, orc = _render_pair(_tex(kind="checker", w=256, h=256), res=64, spp=8)
    oxyz, owt, _, _ = orc.render_path_ex(max_depth=1)
    gxyz, gwt, _ = prod.render_path(max_depth=1)
    g = prod.film_to_rgb(gxyz, gwt); o = prod.film_to_rgb(oxyz, owt)
    mean = float(o.mean())
    rmse = float(np.sqrt(((g - o) ** 2).mean()))
    assert rmse <= 1e-3 * mean
    assert float((np.abs(g - o).max(axis=2) > 1e-2 * mean).mean()) <= 1e-3
    far = g[31:34, 8:56, 1]    # just below the horizon of the tilted view: many checker periods per pixel -> filtered to grey
    assert far.std() < 0.25 * far.mean() and 0.35 < float(far.mean()) < 0.65   # 0/1 texels under a white sky: about one half


def _plastic(kd=True, ks=True):
    def make(sc, tex):
        m = sc.add_material_plastic((1, 1, 1) if kd else (0.3, 0.2, 0.1), (1, 1, 1) if ks else (0.25, 0.25, 0.25), 0.1, True)
        if kd: sc.set_material_texture(m, "Kd", tex)
        if ks: sc.set_material_texture(m, "Ks", sc.add_texture_scale(tex, sc.add_texture_constant((0.5, 0.5, 0.5))))
        return m
    return make


def _mirror(sc, tex):
    m = sc.add_material_mirror((1, 1, 1)); sc.set_material_texture(m, "Kr", tex); return m


def _substrate(sc, tex):
    m = sc.add_material_substrate((1, 1, 1), (1, 1, 1), 0.2, 0.1, True)
    sc.set_material_texture(m, "Kd", tex)
    sc.set_material_texture(m, "Ks", sc.add_texture_mix(tex, sc.add_texture_constant((0.04, 0.04, 0.04)), sc.add_texture_constant(0.5)))
    return m


def _uber(sc, tex):
    m = sc.add_material_uber((1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1), (0.8, 0.7, 0.9), 0.1, 0.2, 1.4, True)   # opacity < 1: the pass-through lobe exists too
    half = sc.add_texture_scale(tex, sc.add_texture_constant((0.5, 0.5, 0.5)))
    sc.set_material_texture(m, "Kd", tex); sc.set_material_texture(m, "Ks", half)
    sc.set_material_texture(m, "Kr", sc.add_texture_scale(half, half)); sc.set_material_texture(m, "Kt", half)
    return m


def _glass(rough):
    def make(sc, tex):
        m = sc.add_material_glass((1, 1, 1), (1, 1, 1), rough, rough, 1.5, True)
        sc.set_material_texture(m, "Kr", tex); sc.set_material_texture(m, "Kt", sc.add_texture_mix(tex, sc.add_texture_constant((1, 1, 1)), sc.add_texture_constant(0.5)))
        return m
    return make


def _fimg(sc, lo, hi, zeros=False, **kw):
    img = make_image(24, 24, seed=13)
    if zeros: img[:, :8] = 0.0
    t = sc.add_texture_imagemap(sc.add_mipmap(img, as_float=True, **kw), su=2.0, sv=2.0)
    return sc.add_texture_mix(sc.add_texture_constant(lo), sc.add_texture_constant(hi), t) if not zeros else sc.add_texture_scale(t, sc.add_texture_constant(hi))


def _matte_sigma(sc, tex):
    m = sc.add_material_matte_tex(tex, 10.0); sc.set_material_float_texture(m, "sigma", _fimg(sc, 0.0, 70.0, zeros=True)); return m      # sigma = 0 on a third of the map: Lambert there


def _plastic_rough(remap):
    def make(sc, tex):
        m = sc.add_material_plastic((1, 1, 1), (0.4, 0.4, 0.4), 0.1, remap); sc.set_material_texture(m, "Kd", tex)
        sc.set_material_float_texture(m, "roughness", _fimg(sc, 0.02, 0.5, trilinear=True)); return m
    return make


def _uber_rough(sc, tex):
    m = sc.add_material_uber((0.4, 0.3, 0.2), (0.5, 0.5, 0.5), (0.1, 0.1, 0.1), (0, 0, 0), (1, 1, 1), 0.1, 0.1, 1.5, True)
    sc.set_material_float_texture(m, "uroughness", _fimg(sc, 0.01, 0.3)); sc.set_material_float_texture(m, "vroughness", _fimg(sc, 0.2, 0.6)); return m


def _substrate_metal(sc, tex):
    m = sc.add_material_substrate((0.5, 0.3, 0.2), (0.1, 0.1, 0.1), 0.1, 0.1, False)
    sc.set_material_float_texture(m, "uroughness", _fimg(sc, 0.05, 0.4)); sc.set_material_texture(m, "Kd", tex); return m


def _metal_rough(sc, tex):
    m = sc.add_material_metal((0.2, 0.92, 1.1), (3.9, 2.45, 2.14), 0.05, 0.05, True)
    sc.set_material_float_texture(m, "roughness", _fimg(sc, 0.005, 0.3)); return m


def _translucent(kd=True, ks=True, rough=False, reflect=(0.5, 0.4, 0.6), transmit=(0.4, 0.5, 0.3)):
    def make(sc, tex):
        m = sc.add_material_translucent((1, 1, 1) if kd else (0.3, 0.2, 0.1), (1, 1, 1) if ks else (0.2, 0.2, 0.2), reflect, transmit, 0.1, True)
        if kd: sc.set_material_texture(m, "Kd", tex)
        if ks: sc.set_material_texture(m, "Ks", sc.add_texture_scale(tex, sc.add_texture_constant((0.5, 0.5, 0.5))))
        if rough: sc.set_material_float_texture(m, "roughness", _fimg(sc, 0.02, 0.4))
        return m
    return make


def _translucent_black_product(sc, tex):
    # Kd has no red, reflect is red only: r * kd is black wherever kd is not, and the reference still adds that (black) Lambertian lobe
    m = sc.add_material_translucent((1, 1, 1), (0.2, 0.2, 0.2), (0.8, 0, 0), (0.5, 0.5, 0.5), 0.1, True)
    sc.set_material_texture(m, "Kd", sc.add_texture_scale(tex, sc.add_texture_constant((0.0, 1.0, 1.0))))
    return m


def _mix_textured(second):
    def make(sc, tex):
        a = _plastic()(sc, tex)
        b = second(sc, sc.add_texture_scale(tex, sc.add_texture_constant((0.9, 0.6, 0.3))))
        return sc.add_material_mix(a, b, (0.3, 0.5, 0.7))
    return make


def _matte_tex(sc, tex):
    return sc.add_material_matte_tex(tex, 25.0)


MATERIAL_CASES = {
    "translucent_kd_ks_with_black_texels": (_translucent(), _tex(kind="checker", w=8, h=8)),            # all four lobes vanish on the black squares
    "translucent_kd_only_rough_texture": (_translucent(ks=False, rough=True), _tex()),
    "translucent_ks_only_reflect_only": (_translucent(kd=False, transmit=(0, 0, 0)), _tex(w=20, h=20)),
    "translucent_black_product_keeps_the_lobe": (_translucent_black_product, _tex(kind="checker", w=8, h=8)),
    "mix_of_textured_plastic_and_matte": (_mix_textured(_matte_tex), _tex(kind="checker", w=8, h=8)),
    "mix_of_textured_plastic_and_translucent": (_mix_textured(_translucent(ks=False)), _tex()),
    "matte_sigma_texture": (_matte_sigma, _tex()),
    "plastic_roughness_remapped": (_plastic_rough(True), _tex()),
    "plastic_roughness_raw": (_plastic_rough(False), _tex()),
    "uber_uv_roughness": (_uber_rough, _tex()),
    "substrate_uroughness": (_substrate_metal, _tex()),
    "metal_roughness": (_metal_rough, _tex()),
    "uber_all_four": (_uber, _tex(kind="checker", w=8, h=8)),
    "glass_smooth_kr_kt": (_glass(0.0), _tex(kind="checker", w=8, h=8)),
    "glass_rough_kr_kt": (_glass(0.05), _tex(w=20, h=20)),
    "plastic_kd_ks": (_plastic(), _tex()),
    "plastic_kd_checker_with_black_texels": (_plastic(ks=False), _tex(kind="checker", w=16, h=16)),   # Lambert lobe absent on the black squares
    "plastic_ks_only": (_plastic(kd=False), _tex(kind="checker", w=8, h=8, trilinear=True)),         # microfacet lobe absent there
    "mirror_kr": (_mirror, _tex(kind="checker", w=8, h=8)),                                          # no BSDF lobe at all on black squares
    "substrate": (_substrate, _tex(w=50, h=30)),
}


@pytest.mark.parametrize("name", sorted(MATERIAL_CASES))
def test_textured_general_materials_film_bit_exact(name):
    material, builder = MATERIAL_CASES[name]
    prod, orc = _render_pair(builder, material=material, res=48)
    set_libm_mode(1)
    try:
        oxyz, owt, ost, _ = orc.render_path_ex(max_depth=4)
    finally:
        set_libm_mode(0)
    gxyz, gwt, gst = prod.render_path(max_depth=4)
    assert _bits_equal(gxyz, oxyz) and _bits_equal(gwt, owt)
    assert gst.regular_rays == ost.regular_rays and gst.shadow_rays == ost.shadow_rays and gst.paths_zero_radiance == ost.paths_zero_radiance
    assert float(gxyz.mean()) > 0.0


def test_set_material_texture_errors():
    s = pbrt_hip.Scene()
    t = s.add_texture_constant((0.5, 0.5, 0.5))
    glass = s.add_material_glass((1, 1, 1), (1, 1, 1), 0.0, 0.0, 1.5, True)
    metal = s.add_material_metal((0.2, 0.9, 1.1), (3.9, 2.4, 2.1), 0.05, 0.05, True)
    with pytest.raises(pbrt_hip.PbrtHipError, match="no lobe fed by that parameter"):
        s.set_material_texture(metal, "Kr", t)
    with pytest.raises(pbrt_hip.PbrtHipError, match="no lobe fed by that parameter"):
        s.set_material_texture(glass, "Kd", t)
    black = s.add_material_plastic((0, 0, 0), (0.2, 0.2, 0.2), 0.1, True)     # Kd black: the Lambert lobe was never made
    with pytest.raises(pbrt_hip.PbrtHipError, match="non-black placeholder"):
        s.set_material_texture(black, "Kd", t)
    pl = s.add_material_plastic((1, 1, 1), (0.2, 0.2, 0.2), 0.1, True); s.set_material_texture(pl, "Kd", t)
    s.add_material_mix(pl, glass, (0.5, 0.5, 0.5))                            # textured sub-materials mix
    bumpy = s.add_material_matte((0.5, 0.5, 0.5), 0.0); s.set_material_bump(bumpy, s.add_texture_constant(0.1))
    s.add_material_mix(pl, bumpy, (0.5, 0.5, 0.5))                            # bump-mapped children are taken (films: tests/test_textured_params.py)
    mx = s.add_material_mix(bumpy, pl, (0.5, 0.5, 0.5)); s.set_material_texture(mx, "amount", t)
    with pytest.raises(pbrt_hip.PbrtHipError, match="amount is a texture"):
        s.add_material_mix(mx, pl, (0.5, 0.5, 0.5))
    with pytest.raises(pbrt_hip.PbrtHipError, match="opacity belongs to UberMaterial"):
        s.set_material_texture(pl, "opacity", t)
    r1 = s.add_material_plastic((0.5, 0.5, 0.5), (0.2, 0.2, 0.2), 0.1, True); s.set_material_float_texture(r1, "roughness", s.add_texture_constant(0.2))
    r2 = s.add_material_plastic((0.5, 0.5, 0.5), (0.2, 0.2, 0.2), 0.1, True); s.set_material_float_texture(r2, "roughness", s.add_texture_constant(0.3))
    with pytest.raises(pbrt_hip.PbrtHipError, match="one textured roughness"):
        s.add_material_mix(r1, r2, (0.5, 0.5, 0.5))
    with pytest.raises(pbrt_hip.PbrtHipError):
        s.set_material_texture(pl, "Kd", 999)


def test_procedural_2d_lookups_bit_exact():
    uv, d = probe_points(4000, 8)
    uv = (uv * np.float32(3.0)).astype(np.float32)
    prod = pbrt_hip.Scene(); orc = OracleScene()
    outs = []
    for sc in (prod, orc):
        img = sc.add_texture_imagemap(sc.add_mipmap(make_image(16, 16, seed=6)))
        k = sc.add_texture_constant((0.2, 0.4, 0.6))
        texs = [sc.add_texture_checkerboard(img, k, su=3.0, sv=2.0, du=0.1, dv=0.2),
                sc.add_texture_checkerboard(k, img, su=5.0, sv=5.0, aa="none"),
                sc.add_texture_uv(su=1.5, sv=2.5, du=-0.3, dv=0.7),
                sc.add_texture_bilerp((1, 0, 0), (0.2, 1, 0), (0, 0.3, 1), (1, 1, 0.5), su=0.5, sv=0.5),
                sc.add_texture_dots(img, k, su=7.0, sv=7.0)]
        texs.append(sc.add_texture_mix(texs[0], texs[4], sc.add_texture_bilerp(0.0, 1.0, 0.25, 0.75)))
        if sc is orc: set_libm_mode(1)
        try:
            outs.append([sc.texture_eval(t, uv, d) for t in texs])
        finally:
            set_libm_mode(0)
    for a, b in zip(*outs):
        assert _bits_equal(a, b)


def test_checkerboard_and_dots_scene_film_bit_exact():
    def builder(sc):
        a = sc.add_texture_constant((0.8, 0.7, 0.1)); b = sc.add_texture_imagemap(sc.add_mipmap(make_image(32, 32, seed=2)))
        return sc.add_texture_dots(sc.add_texture_uv(su=4.0, sv=4.0), sc.add_texture_checkerboard(a, b, su=6.0, sv=6.0), su=5.0, sv=5.0)
    prod, orc = _render_pair(builder, res=48)
    set_libm_mode(1)
    try:
        oxyz, owt, _, _ = orc.render_path_ex(max_depth=3)
    finally:
        set_libm_mode(0)
    gxyz, gwt, _ = prod.render_path(max_depth=3)
    assert _bits_equal(gxyz, oxyz) and _bits_equal(gwt, owt)


def test_procedural_3d_lookups_and_scene_bit_exact():
    rng = np.random.default_rng(9)
    n = 3000
    pts = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    dx = (rng.normal(size=(n, 3)) * rng.choice([0.0, 1e-3, 0.05, 2.0], size=(n, 1))).astype(np.float32)
    dy = (rng.normal(size=(n, 3)) * rng.choice([0.0, 1e-3, 0.05, 2.0], size=(n, 1))).astype(np.float32)
    uv = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    host = pbrt_hip.Host()
    m = host.compose(host.rotate(30.0, [1, 2, 3]), host.scale([1.5, 0.7, 2.0]))[0]
    prod = pbrt_hip.Scene(); orc = OracleScene()
    outs = []
    for sc in (prod, orc):
        k1 = sc.add_texture_constant((0.9, 0.1, 0.2)); k2 = sc.add_texture_constant((0.1, 0.3, 0.9))
        texs = [sc.add_texture_fbm(m, 0.55, 6), sc.add_texture_fbm(m, 0.45, 7, wrinkled=True), sc.add_texture_windy(m), sc.add_texture_marble(m, 0.5, 8, 2.0, 0.3),
                sc.add_texture_checkerboard3d(k1, k2, m)]
        texs.append(sc.add_texture_mix(texs[3], texs[4], texs[1]))
        if sc is orc: set_libm_mode(1)
        try:
            outs.append([sc.texture_eval(t, uv, p=pts, dpdx=dx, dpdy=dy) for t in texs])
        finally:
            set_libm_mode(0)
    for a, b in zip(*outs):
        assert _bits_equal(a, b)

    def builder(sc):
        return sc.add_texture_mix(sc.add_texture_marble(m, 0.5, 8, 2.0, 0.3), sc.add_texture_checkerboard3d(sc.add_texture_constant((0.9, 0.1, 0.2)), sc.add_texture_constant((0.1, 0.3, 0.9)), m),
                                  sc.add_texture_fbm(m, 0.5, 4, wrinkled=True))
    for kw in (dict(), dict(instance=True)):
        prod, orc = _render_pair(builder, res=40, **kw)
        set_libm_mode(1)
        try:
            oxyz, owt, _, _ = orc.render_path_ex(max_depth=3)
        finally:
            set_libm_mode(0)
        gxyz, gwt, _ = prod.render_path(max_depth=3)
        assert _bits_equal(gxyz, oxyz) and _bits_equal(gwt, owt)


def test_2d_mappings_lookups_and_scene_bit_exact():
    rng = np.random.default_rng(10)
    n = 3000
    pts = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    dx = (rng.normal(size=(n, 3)) * rng.choice([0.0, 1e-3, 0.05, 1.0], size=(n, 1))).astype(np.float32)
    dy = (rng.normal(size=(n, 3)) * rng.choice([0.0, 1e-3, 0.05, 1.0], size=(n, 1))).astype(np.float32)
    host = pbrt_hip.Host()
    w2t = host.compose(host.rotate(40.0, [1, 1, 0]), host.translate([0.2, -0.1, 0.3]))[1]
    planar = [0.3, 0.1, 0.0, -0.1, 0.4, 0.2, 0.05, 0.15]

    def textures(sc):
        out = []
        for kind, prm in (("spherical", w2t), ("cylindrical", w2t), ("planar", planar)):
            im = sc.add_texture_imagemap(sc.add_mipmap(make_image(32, 16, seed=3))); sc.set_texture_mapping(im, kind, prm)
            ck = sc.add_texture_checkerboard(sc.add_texture_constant((0.9, 0.8, 0.1)), sc.add_texture_constant((0.1, 0.1, 0.5)), su=1.0, sv=1.0); sc.set_texture_mapping(ck, kind, prm)
            dt = sc.add_texture_dots(im, ck); sc.set_texture_mapping(dt, kind, prm)
            out += [im, ck, dt]
        return out
    prod = pbrt_hip.Scene(); orc = OracleScene()
    outs = []
    for sc in (prod, orc):
        texs = textures(sc)
        if sc is orc: set_libm_mode(1)
        try:
            outs.append([sc.texture_eval(t, np.zeros((n, 2), np.float32), p=pts, dpdx=dx, dpdy=dy) for t in texs])
        finally:
            set_libm_mode(0)
    for a, b in zip(*outs):
        assert _bits_equal(a, b)
    for pick in (0, 4, 8):
        prod, orc = _render_pair(lambda sc, pick=pick: textures(sc)[pick], res=40)
        set_libm_mode(1)
        try:
            oxyz, owt, _, _ = orc.render_path_ex(max_depth=3)
        finally:
            set_libm_mode(0)
        gxyz, gwt, _ = prod.render_path(max_depth=3)
        assert _bits_equal(gxyz, oxyz) and _bits_equal(gwt, owt)


def _bumped(base, bump_builder):
    def make(sc, tex):
        m = base(sc, tex)
        sc.set_material_bump(m, bump_builder(sc))
        return m
    return make


def _float_image(sc, **kw):
    return sc.add_texture_imagemap(sc.add_mipmap(make_image(32, 32, seed=21), as_float=True, **kw), su=2.0, sv=2.0)


BUMP_CASES = {
    "matte_image_bump": (_bumped(lambda sc, tex: sc.add_material_matte((0.6, 0.6, 0.6), 0.0), _float_image), {}),
    "textured_matte_trilinear_bump_lens": (_bumped(lambda sc, tex: sc.add_material_matte_tex(tex, 10.0), lambda sc: _float_image(sc, trilinear=True)), dict(lens_radius=0.03)),
    "plastic_wrinkled_bump_instanced": (_bumped(_plastic(), lambda sc: sc.add_texture_scale(sc.add_texture_fbm(omega=0.6, octaves=5, wrinkled=True), sc.add_texture_constant(0.05))), dict(instance=True)),
    "glass_dots_bump": (_bumped(_glass(0.0), lambda sc: sc.add_texture_dots(sc.add_texture_constant(0.02), sc.add_texture_constant(0.0), su=6.0, sv=6.0)), {}),
    "mirror_constant_bump": (_bumped(_mirror, lambda sc: sc.add_texture_constant(0.3)), {}),
}


@pytest.mark.parametrize("name", sorted(BUMP_CASES))
def test_bump_mapping_film_bit_exact(name):
    material, kw = BUMP_CASES[name]
    prod, orc = _render_pair(_tex(), material=material, res=48, **kw)
    set_libm_mode(1)
    try:
        oxyz, owt, ost, _ = orc.render_path_ex(max_depth=4)
    finally:
        set_libm_mode(0)
    gxyz, gwt, gst = prod.render_path(max_depth=4)
    assert _bits_equal(gxyz, oxyz) and _bits_equal(gwt, owt)
    assert gst.regular_rays == ost.regular_rays and gst.shadow_rays == ost.shadow_rays


def test_bump_on_a_mesh_with_vertex_normals_and_tangents():
    """dn/du, dn/dv and the shading dp/dv only matter to bump mapping: a curved patch with N, S and UV per vertex, also through an instance."""
    n = 6
    g = np.linspace(-1, 1, n + 1, dtype=np.float32)
    X, Y = np.meshgrid(g, g, indexing="xy")
    Z = (0.3 * np.cos(1.5 * X) * np.cos(1.2 * Y)).astype(np.float32)
    P = np.stack([2 * X.ravel(), 2 * Y.ravel(), Z.ravel() + 0.3], axis=1).astype(np.float32)
    N = np.stack([0.45 * np.sin(1.5 * X) * np.cos(1.2 * Y), 0.36 * np.cos(1.5 * X) * np.sin(1.2 * Y), np.ones_like(X)], axis=2).reshape(-1, 3).astype(np.float32)
    N /= np.linalg.norm(N, axis=1, keepdims=True)
    S = np.tile(np.array([[1.0, 0.2, 0.0]], np.float32), (len(P), 1))
    UV = np.stack([(X.ravel() + 1) * 1.5, (Y.ravel() + 1) * 1.5], axis=1).astype(np.float32)
    idx = []
    for j in range(n):
        for i in range(n):
            a = j * (n + 1) + i; b = a + 1; c = a + n + 1; d = c + 1
            idx += [a, b, d, a, d, c]
    idx = np.array(idx, np.uint32)
    for instanced in (False, True):
        def extra(sc):
            m = sc.add_material_plastic((0.7, 0.3, 0.2), (0.3, 0.3, 0.3), 0.08, True)
            sc.set_material_bump(m, sc.add_texture_imagemap(sc.add_mipmap(make_image(64, 64, seed=5), as_float=True, scale=0.08), su=1.0, sv=1.0))
            if instanced:
                host = pbrt_hip.Host()
                ob = sc.object_begin(); sc.add_mesh(P, idx, m, N=N, S=S, UV=UV); sc.object_end()
                t = host.compose(host.translate([0.0, 0.5, 0.6]), host.rotate(15.0, [1, 0, 0]))
                sc.add_instance(ob, t[0], t[1])
            else:
                sc.add_mesh(P + np.array([0, 0.5, 0.6], np.float32), idx, m, N=N, S=S, UV=UV)
        prod, orc = _render_pair(_tex(), extra=extra, res=48)
        set_libm_mode(1)
        try:
            oxyz, owt, _, _ = orc.render_path_ex(max_depth=3)
        finally:
            set_libm_mode(0)
        gxyz, gwt, _ = prod.render_path(max_depth=3)
        assert _bits_equal(gxyz, oxyz) and _bits_equal(gwt, owt)


@pytest.mark.parametrize("strategy", [0, 1, 2])
def test_environment_map_light_film_bit_exact(strategy):
    """InfiniteAreaLight with a radiance map (NPOT, strongly peaked) + a second light, all three light-sampling strategies, glossy and textured surfaces."""
    rng = np.random.default_rng(31)
    env = rng.uniform(0.0, 0.3, (12, 20, 3)).astype(np.float32)
    env[2:4, 5:8] = (40.0, 30.0, 20.0)            # a sun
    host = pbrt_hip.Host()
    t = host.compose(host.rotate(35.0, [0, 0, 1]), host.rotate(-60.0, [1, 0, 0]))

    def scene(sc):
        tex = sc.add_texture_imagemap(sc.add_mipmap(make_image(32, 32, seed=4)))
        mat = sc.add_material_plastic((1, 1, 1), (0.3, 0.3, 0.3), 0.08, True); sc.set_material_texture(mat, "Kd", tex)
        grey = sc.add_material_matte((0.6, 0.6, 0.6), 0.0)
        P = np.array([[-4, -4, 0], [4, -4, 0], [4, 4, 0], [-4, 4, 0]], np.float32)
        sc.add_mesh(P, [0, 1, 2, 0, 2, 3], mat, UV=np.array([[0, 0], [3, 0], [3, 3], [0, 3]], np.float32))
        B = np.array([[-0.5, -0.5, 0.2], [0.5, -0.5, 0.2], [0.5, 0.5, 0.2], [-0.5, 0.5, 0.2], [0, 0, 1.2]], np.float32)
        sc.add_mesh(B, np.array([0, 1, 4, 1, 2, 4, 2, 3, 4, 3, 0, 4], np.uint32), grey)
        sc.add_light_infinite_map((1.0, 0.9, 0.8), env, t[0], t[1])
        sc.add_light_point((3.0, 3.0, 3.0), (1.5, -1.0, 2.0))
        w2c, c2w = host.look_at((0.0, -6.0, 1.2), (0, 0, 0.2), (0, 0, 1))
        sc.set_camera_perspective(host.perspective_raster_to_camera(40.0, 48, 48), c2w)
        cb, table, sb = host.film_box(48, 48)
        sc.set_film(48, 48, cb, (0.5, 0.5), table); sc.set_sampler(0, 4, sb); sc.build_accel(0, 4)
    prod = pbrt_hip.Scene(); orc = OracleScene()
    scene(prod); scene(orc)
    set_libm_mode(1)
    try:
        oxyz, owt, ost, _ = orc.render_path_ex(max_depth=4, light_strategy=strategy)
    finally:
        set_libm_mode(0)
    gxyz, gwt, gst = prod.render_path(max_depth=4, light_strategy=strategy)
    assert _bits_equal(gxyz, oxyz) and _bits_equal(gwt, owt)
    assert gst.regular_rays == ost.regular_rays and gst.shadow_rays == ost.shadow_rays
    assert gst.light_distributions_created == ost.light_distributions_created
    assert float(gxyz.mean()) > 0.0


def test_alpha_mask_textures_hits_occlusion_and_film_bit_exact():
    """Alpha-mask textures are part of the traversal: closest hits, any-hit occlusion (alpha and shadowalpha) and a film with cut-out leaves, also
    through an instance, against the oracle."""
    import scenes as SC
    host = pbrt_hip.Host()
    rng = np.random.default_rng(12)
    mask = (rng.uniform(0, 1, (16, 16)) > 0.45).astype(np.float32)            # exactly 0 / 1 texels
    img = np.repeat(mask[..., None], 3, axis=2)

    def scene(sc, instanced):
        grey = sc.add_material_matte((0.6, 0.6, 0.6), 0.0); green = sc.add_material_matte((0.2, 0.7, 0.2), 0.0)
        a = sc.add_texture_imagemap(sc.add_mipmap(img, as_float=True, trilinear=True, wrap="clamp"))
        sa = sc.add_texture_checkerboard(sc.add_texture_constant(1.0), sc.add_texture_constant(0.0), su=5.0, sv=5.0, aa="none")
        P, idx = SC.grid_mesh(6, z=0.0, size=2.0)
        UV = ((P[:, :2] + 2.0) / 4.0).astype(np.float32)
        leaves = [(0.4, (a, None)), (0.8, (None, sa)), (1.2, (a, sa))]
        if instanced:
            ob = sc.object_begin()
            sc.add_mesh(P, idx, green, UV=UV); sc.set_last_mesh_alpha_textures(a, sa)
            sc.object_end()
            for z in (0.4, 0.9):
                t = host.compose(host.translate([0.1, 0.0, z]), host.rotate(20.0 * z, [0, 0, 1]))
                sc.add_instance(ob, t[0], t[1])
        else:
            for z, (al, sh) in leaves:
                sc.add_mesh(P + np.float32([0, 0, z]), idx, green, UV=UV); sc.set_last_mesh_alpha_textures(al, sh)
        sc.add_mesh(P * np.float32(2.0) + np.float32([0, 0, -0.2]), idx, grey)                # an opaque floor below
        sc.add_light_infinite((1.0, 1.0, 1.0))
        sc.add_light_point((8.0, 8.0, 8.0), (0.5, -0.5, 3.0))
        w2c, c2w = host.look_at((0.0, -4.0, 4.0), (0, 0, 0.3), (0, 0, 1))
        sc.set_camera_perspective(host.perspective_raster_to_camera(45.0, 48, 48), c2w)
        cb, table, sb = host.film_box(48, 48)
        sc.set_film(48, 48, cb, (0.5, 0.5), table); sc.set_sampler(0, 4, sb); sc.build_accel(0, 4)
    for instanced in (False, True):
        prod = pbrt_hip.Scene(); orc = OracleScene()
        scene(prod, instanced); scene(orc, instanced)
        rays = SC.random_rays(20000, 5, bound=2.2)
        gh = prod.intersect_batch(rays); oh = orc.intersect_batch(rays)
        for f in ("t", "prim", "b0", "b1", "b2"):
            assert _bits_equal(np.ascontiguousarray(gh[f]), np.ascontiguousarray(oh[f])), f
        assert np.array_equal(prod.occluded_batch(rays), orc.occluded_batch(rays))
        holes = int((gh["prim"] == 0xFFFFFFFF).sum())
        set_libm_mode(1)
        try:
            oxyz, owt, ost, _ = orc.render_path_ex(max_depth=3)
        finally:
            set_libm_mode(0)
        gxyz, gwt, gst = prod.render_path(max_depth=3)
        assert _bits_equal(gxyz, oxyz) and _bits_equal(gwt, owt)
        assert gst.regular_rays == ost.regular_rays and gst.shadow_rays == ost.shadow_rays
    # the mask really cuts holes: with an all-ones mask fewer rays get through the leaves
    prod2 = pbrt_hip.Scene()
    img_full = np.ones_like(img)
    def scene_full(sc):
        green = sc.add_material_matte((0.2, 0.7, 0.2), 0.0)
        a = sc.add_texture_imagemap(sc.add_mipmap(img_full, as_float=True, trilinear=True, wrap="clamp"))
        P, idx = SC.grid_mesh(6, z=0.4, size=2.0)
        sc.add_mesh(P, idx, green, UV=((P[:, :2] + 2.0) / 4.0).astype(np.float32)); sc.set_last_mesh_alpha_textures(a, None)
        sc.build_accel(0, 4)
    scene_full(prod2)
    down = np.zeros(4000, pbrt_hip.RAY_DTYPE); down["o"] = np.c_[rng.uniform(-1.9, 1.9, (4000, 2)), np.full(4000, 3.0)].astype(np.float32); down["d"] = (0, 0, -1); down["t_max"] = np.inf
    full = int((prod2.intersect_batch(down)["prim"] != 0xFFFFFFFF).sum())
    prod3 = pbrt_hip.Scene()
    def scene_mask(sc):
        green = sc.add_material_matte((0.2, 0.7, 0.2), 0.0)
        a = sc.add_texture_imagemap(sc.add_mipmap(img, as_float=True, trilinear=True, wrap="clamp"))
        P, idx = SC.grid_mesh(6, z=0.4, size=2.0)
        sc.add_mesh(P, idx, green, UV=((P[:, :2] + 2.0) / 4.0).astype(np.float32)); sc.set_last_mesh_alpha_textures(a, None)
        sc.build_accel(0, 4)
    scene_mask(prod3)
    cut = int((prod3.intersect_batch(down)["prim"] != 0xFFFFFFFF).sum())
    assert full == 4000 and 0.3 * full < cut < 0.97 * full      # bilinear lookups: a hole needs all four neighbouring texels at exactly 0
