"""The structural material parameters as textures — UberMaterial's opacity (uber.rs:126-160), MixMaterial's amount (mix.rs:59-60), GlassMaterial's
u / v roughness (glass.rs:110-141), MetalMaterial's eta / k (metal.rs:121-125) — and bump-mapped children of a mix (mix.rs:63-76).

CPU part (not gpu): closed-form pins of the oracle's per-hit evaluation.  A texture that is CONSTANT must give the film of the constant it stands for,
bit for bit: the per-hit code path (lobe list rebuilt at every hit, colours from the texture pass) and the creation-time path (lobe list made once) are two
restatements of the same reference lines, so their agreement pins both.  A checkerboard of two constants must give, pixel by pixel, one of the two constant
films wherever a path touches only one kind of square on its first bounce — tested at depth 1 on directly visible hits.
GPU part: the same scenes and textured versions of them, device film against oracle film, bit for bit."""
import numpy as np
import pytest

import pbrt_hip
from oracle_binding import OracleScene, set_libm_mode
from texture_scenes import make_image, textured_quad_scene

ONE = (1.0, 1.0, 1.0)


def _film(scene_cls, host, material, depth=4, res=40, spp=4, **kw):
    s = scene_cls()
    textured_quad_scene(s, host, lambda sc: sc.add_texture_constant((0.5, 0.5, 0.5)), res=res, spp=spp, material=lambda sc, tex: material(sc), **kw)
    if isinstance(s, OracleScene):
        set_libm_mode(1)
        try:
            x, w, st, _ = s.render_path_ex(max_depth=depth)
        finally:
            set_libm_mode(0)
    else:
        x, w, st = s.render_path(max_depth=depth)
    s.close()
    return x, w, (st.regular_rays, st.shadow_rays, st.paths_total, st.paths_zero_radiance)


def _same(a, b):
    return np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1], b[1]) and a[2] == b[2]


# ---- the materials, each as (constant version, the same with the parameter as a constant TEXTURE, a genuinely textured version) ------------------------------
def uber(opacity, textured=None, kd_tex=False):
    def make(sc):
        if textured is None:
            m = sc.add_material_uber((0.5, 0.4, 0.3), (0.3, 0.3, 0.3), (0.2, 0.2, 0.2), (0.15, 0.2, 0.25), opacity, 0.1, 0.2, 1.4, True)
        else:
            m = sc.add_material_uber((0.5, 0.4, 0.3), (0.3, 0.3, 0.3), (0.2, 0.2, 0.2), (0.15, 0.2, 0.25), ONE, 0.1, 0.2, 1.4, True)
            sc.set_material_texture(m, "opacity", textured(sc))
        if kd_tex:
            sc.set_material_texture(m, "Kd", sc.add_texture_imagemap(sc.add_mipmap(make_image(16, 16, seed=3))))
        return m
    return make


def mix(amount, textured=None, bump_first=False, bump_second=False):
    def make(sc):
        a = sc.add_material_plastic((0.6, 0.2, 0.2), (0.3, 0.3, 0.3), 0.15, True)
        b = sc.add_material_matte((0.2, 0.5, 0.7), 20.0)
        bump = lambda: sc.add_texture_scale(sc.add_texture_imagemap(sc.add_mipmap(make_image(32, 32, seed=8), as_float=True), su=3.0, sv=3.0), sc.add_texture_constant(0.08))
        if bump_first: sc.set_material_bump(a, bump())
        if bump_second: sc.set_material_bump(b, bump())
        m = sc.add_material_mix(a, b, amount if textured is None else (0.5, 0.5, 0.5))
        if textured is not None: sc.set_material_texture(m, "amount", textured(sc))
        return m
    return make


def glass(ur, vr, utex=None, vtex=None, remap=True, eta=1.5, index_tex=None):
    def make(sc):
        m = sc.add_material_glass((0.9, 0.9, 0.9), (0.8, 0.85, 0.9), ur, vr, eta, remap)
        if utex is not None: sc.set_material_float_texture(m, "uroughness", utex(sc))
        if vtex is not None: sc.set_material_float_texture(m, "vroughness", vtex(sc))
        if index_tex is not None: sc.set_material_float_texture(m, "index", index_tex(sc))   # glass.rs:102: `let eta = self.index.evaluate(..)`
        return m
    return make


def uber_index(opacity, eta, index_tex=None, opacity_tex=None):
    """UberMaterial whose index of refraction (uber.rs:128) is the constant `eta` or a texture; Kt non-black so that BSDF::eta matters."""
    def make(sc):
        m = sc.add_material_uber((0.4, 0.35, 0.3), (0.3, 0.3, 0.3), (0.25, 0.25, 0.25), (0.5, 0.55, 0.6), ONE if opacity_tex is not None else opacity, 0.1, 0.2, eta, True)
        if opacity_tex is not None: sc.set_material_texture(m, "opacity", opacity_tex(sc))
        if index_tex is not None: sc.set_material_float_texture(m, "index", index_tex(sc))
        return m
    return make


def metal(eta, k, eta_tex=None, k_tex=None):
    def make(sc):
        m = sc.add_material_metal(eta, k, 0.05, 0.1, True)
        if eta_tex is not None: sc.set_material_texture(m, "eta", eta_tex(sc))
        if k_tex is not None: sc.set_material_texture(m, "k", k_tex(sc))
        return m
    return make


def translucent(reflect, transmit, rtex=None, ttex=None, kd_tex=False, rough_tex=None, kd=(0.6, 0.5, 0.4), ks=(0.3, 0.3, 0.3)):
    """TranslucentMaterial whose reflect / transmit (translucent.rs:70-71) are constants or textures."""
    def make(sc):
        m = sc.add_material_translucent(kd, ks, reflect, transmit, 0.2, True)
        if kd_tex: sc.set_material_texture(m, "Kd", sc.add_texture_imagemap(sc.add_mipmap(make_image(16, 16, seed=4))))
        if rough_tex is not None: sc.set_material_float_texture(m, "roughness", rough_tex(sc))
        if rtex is not None: sc.set_material_texture(m, "reflect", rtex(sc))
        if ttex is not None: sc.set_material_texture(m, "transmit", ttex(sc))
        return m
    return make


const = lambda v: (lambda sc: sc.add_texture_constant(v))
checker = lambda a, b, n=6.0: (lambda sc: sc.add_texture_checkerboard(sc.add_texture_constant(a), sc.add_texture_constant(b), su=n, sv=n, aa="none"))
image = lambda seed, scale=1.0: (lambda sc: sc.add_texture_scale(sc.add_texture_imagemap(sc.add_mipmap(make_image(24, 24, seed=seed))), sc.add_texture_constant(scale)))

PINS = [   # (name, constant material, the same through a constant texture)
    ("uber opacity 1", uber(ONE), uber(None, const(ONE))),
    ("uber opacity 0.6", uber((0.6, 0.6, 0.6)), uber(None, const((0.6, 0.6, 0.6)))),
    ("uber opacity 0 (pass-through only)", uber((0.0, 0.0, 0.0)), uber(None, const((0.0, 0.0, 0.0)))),
    ("uber opacity rgb + Kd texture", uber((0.9, 0.5, 0.2), kd_tex=True), uber(None, const((0.9, 0.5, 0.2)), kd_tex=True)),
    ("mix amount 0.3", mix((0.3, 0.3, 0.3)), mix(None, const((0.3, 0.3, 0.3)))),
    ("mix amount rgb beyond 1", mix((1.4, 0.5, 0.0)), mix(None, const((1.4, 0.5, 0.0)))),
    ("glass smooth", glass(0.0, 0.0), glass(0.3, 0.3, const(0.0), const(0.0))),
    ("glass rough", glass(0.2, 0.1), glass(0.0, 0.0, const(0.2), const(0.1))),
    ("glass u texture only", glass(0.25, 0.1), glass(0.0, 0.1, const(0.25), None)),
    ("glass rough, no remap", glass(0.2, 0.1, remap=False), glass(0.0, 0.0, const(0.2), const(0.1), remap=False)),
    ("glass smooth, index 1.7", glass(0.0, 0.0, eta=1.7), glass(0.0, 0.0, index_tex=const(1.7))),
    ("glass rough, index 1.33", glass(0.2, 0.1, eta=1.33), glass(0.2, 0.1, index_tex=const(1.33))),
    ("glass roughness textures + index", glass(0.2, 0.1, eta=1.8), glass(0.0, 0.0, const(0.2), const(0.1), index_tex=const(1.8))),
    ("uber index 1.7, opacity 1", uber_index(ONE, 1.7), uber_index(ONE, 1.5, const(1.7))),
    ("uber index 1.3, opacity 0.6", uber_index((0.6, 0.6, 0.6), 1.3), uber_index((0.6, 0.6, 0.6), 1.5, const(1.3))),
    ("uber index 1.3, opacity texture", uber_index((0.7, 0.5, 0.9), 1.3), uber_index(None, 1.5, const(1.3), const((0.7, 0.5, 0.9)))),
    ("translucent reflect texture", translucent((0.7, 0.6, 0.5), (0.3, 0.4, 0.5)), translucent((0.1, 0.1, 0.1), (0.3, 0.4, 0.5), const((0.7, 0.6, 0.5)))),
    ("translucent both textures", translucent((0.7, 0.6, 0.5), (0.3, 0.4, 0.5)), translucent(ONE, ONE, const((0.7, 0.6, 0.5)), const((0.3, 0.4, 0.5)))),
    ("translucent transmit texture, reflect 0", translucent((0.0, 0.0, 0.0), (0.6, 0.6, 0.7)), translucent((0.0, 0.0, 0.0), (0.1, 0.1, 0.1), None, const((0.6, 0.6, 0.7)))),
    ("translucent reflect texture black, transmit only", translucent((0.0, 0.0, 0.0), (0.5, 0.5, 0.5)), translucent((0.4, 0.4, 0.4), (0.5, 0.5, 0.5), const((0.0, 0.0, 0.0)))),
    ("translucent textures + Kd image + roughness texture", translucent((0.7, 0.6, 0.5), (0.3, 0.4, 0.5), kd_tex=True, rough_tex=const(0.3)), translucent(ONE, ONE, const((0.7, 0.6, 0.5)), const((0.3, 0.4, 0.5)), kd_tex=True, rough_tex=const(0.3))),
    ("translucent no Ks", translucent((0.7, 0.6, 0.5), (0.3, 0.4, 0.5), ks=(0.0, 0.0, 0.0)), translucent(ONE, ONE, const((0.7, 0.6, 0.5)), const((0.3, 0.4, 0.5)), ks=(0.0, 0.0, 0.0))),
    ("metal eta k", metal((0.2, 0.9, 1.1), (3.9, 2.4, 2.1)), metal(ONE, ONE, const((0.2, 0.9, 1.1)), const((3.9, 2.4, 2.1)))),
    ("metal k only", metal((0.2, 0.9, 1.1), (3.9, 2.4, 2.1)), metal((0.2, 0.9, 1.1), ONE, None, const((3.9, 2.4, 2.1)))),
]
TEXTURED = [   # genuinely varying parameters: device against oracle
    ("uber opacity checker", uber(None, checker((1.0, 1.0, 1.0), (0.2, 0.3, 0.4)))),
    ("uber opacity image + Kd image", uber(None, image(5), kd_tex=True)),
    ("uber opacity checker 0 / 1", uber(None, checker((0.0, 0.0, 0.0), ONE))),
    ("mix amount image", mix(None, image(6, 1.3))),
    ("mix amount checker", mix(None, checker((0.0, 0.0, 0.0), (1.0, 1.0, 1.0)))),
    ("mix bumped first child", mix((0.4, 0.4, 0.4), bump_first=True)),
    ("mix bumped both + amount image", mix(None, image(7), bump_first=True, bump_second=True)),
    ("glass roughness checker 0 / rough", glass(0.0, 0.0, checker(0.0, 0.25), checker(0.0, 0.25))),
    ("glass u checker, v constant 0", glass(0.0, 0.0, checker(0.0, 0.3, 4.0), None)),
    ("glass roughness image", glass(0.1, 0.1, image(9, 0.4), image(10, 0.4))),
    ("metal eta k images", metal(ONE, ONE, image(11, 2.0), image(12, 4.0))),
    ("translucent reflect checker, transmit image", translucent(ONE, ONE, checker((0.8, 0.7, 0.6), (0.1, 0.2, 0.3)), image(15))),
    ("translucent reflect / transmit checkers with black squares (null BSDF where both are black)", translucent(ONE, ONE, checker((0.0, 0.0, 0.0), (0.7, 0.7, 0.7), 4.0), checker((0.0, 0.0, 0.0), (0.5, 0.6, 0.7), 4.0))),
    ("translucent transmit checker 0 / 1 + Kd image", translucent((0.5, 0.5, 0.5), ONE, None, checker((0.0, 0.0, 0.0), ONE, 3.0), kd_tex=True)),
    ("glass index checker 1.2 / 1.9", glass(0.0, 0.0, index_tex=checker(1.2, 1.9))),
    ("glass rough, index image", glass(0.15, 0.1, index_tex=lambda sc: sc.add_texture_mix(sc.add_texture_constant(1.1), sc.add_texture_constant(2.2), sc.add_texture_imagemap(sc.add_mipmap(make_image(24, 24, seed=13), as_float=True))))),
    ("uber index checker, opacity checker", uber_index(None, 1.5, checker(1.25, 1.8, 5.0), checker(ONE, (0.3, 0.4, 0.5), 3.0))),
    ("uber index image, constant opacity 0.8", uber_index((0.8, 0.8, 0.8), 1.5, lambda sc: sc.add_texture_mix(sc.add_texture_constant(1.2), sc.add_texture_constant(1.9), sc.add_texture_imagemap(sc.add_mipmap(make_image(24, 24, seed=14), as_float=True))))),
]


@pytest.mark.parametrize("name,const_mat,tex_mat", PINS, ids=[p[0] for p in PINS])
def test_oracle_constant_texture_equals_the_constant(host, name, const_mat, tex_mat):
    a = _film(OracleScene, host, const_mat)
    b = _film(OracleScene, host, tex_mat)
    assert float(a[0].max()) > 0
    assert _same(a, b), f"{name}: the per-hit path disagrees with the creation-time path in {(a[0].view(np.uint32) != b[0].view(np.uint32)).any(axis=2).sum()} pixels"


def test_oracle_second_childs_bump_map_changes_nothing_and_the_firsts_does(host):
    plain = _film(OracleScene, host, mix((0.4, 0.4, 0.4)))
    second = _film(OracleScene, host, mix((0.4, 0.4, 0.4), bump_second=True))
    first = _film(OracleScene, host, mix((0.4, 0.4, 0.4), bump_first=True))
    assert _same(plain, second)                      # mix.rs:63-76: m2 bumps a clone; the BSDF is made on the interaction m1 saw
    assert not np.array_equal(plain[0], first[0])


def test_oracle_checkerboard_opacity_is_one_of_the_two_constants_per_pixel(host):
    """Directly visible hits at depth 1 with pixel-centre sampling off: every pixel's radiance comes from first hits only; where all of a pixel's samples
    land on one kind of square the value equals that constant's film."""
    lo, hi = (0.2, 0.3, 0.4), ONE
    c = _film(OracleScene, host, uber(None, checker(hi, lo, 2.0)), depth=1, res=32, spp=1)
    a = _film(OracleScene, host, uber(hi), depth=1, res=32, spp=1)
    b = _film(OracleScene, host, uber(lo), depth=1, res=32, spp=1)
    eq_a = (c[0].view(np.uint32) == a[0].view(np.uint32)).all(axis=2)
    eq_b = (c[0].view(np.uint32) == b[0].view(np.uint32)).all(axis=2)
    single = c[1] == 1.0     # a Halton sample on a pixel edge also lands in the neighbouring pixel (box filter, closed support): such pixels sum two first hits
    assert single.sum() > 900 and (eq_a | eq_b)[single].all()
    assert eq_a.any() and eq_b.any() and not (eq_a & eq_b).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name,const_mat,tex_mat", PINS, ids=[p[0] for p in PINS])
def test_device_constant_texture_films_bit_exact(host, name, const_mat, tex_mat):
    o = _film(OracleScene, host, tex_mat)
    g = _film(pbrt_hip.Scene, host, tex_mat)
    assert _same(g, o), f"{name}: {(g[0].view(np.uint32) != o[0].view(np.uint32)).any(axis=2).sum()} pixels differ, counters {g[2]} vs {o[2]}"
    assert _same(g, _film(pbrt_hip.Scene, host, const_mat))


@pytest.mark.gpu
@pytest.mark.parametrize("name,mat", TEXTURED, ids=[p[0] for p in TEXTURED])
@pytest.mark.parametrize("instance", [False, True])
def test_device_textured_parameter_films_bit_exact(host, name, mat, instance):
    o = _film(OracleScene, host, mat, depth=5, res=48, instance=instance)
    g = _film(pbrt_hip.Scene, host, mat, depth=5, res=48, instance=instance)
    assert float(o[0].max()) > 0
    assert _same(g, o), f"{name}: {(g[0].view(np.uint32) != o[0].view(np.uint32)).any(axis=2).sum()} pixels differ, counters {g[2]} vs {o[2]}"


def test_oracle_checkerboard_index_is_one_of_the_two_constants_per_pixel(host):
    """A glass floor whose index of refraction is a checkerboard of 1.2 and 1.9, seen at depth 1 with one sample per pixel: a pixel whose sample lands on a square of one
    kind shows exactly the film of the glass made with that constant (the Fresnel term, and with it the lobe choice and the reflected radiance, depend on the index)."""
    c = _film(OracleScene, host, glass(0.0, 0.0, index_tex=checker(1.2, 1.9, 2.0)), depth=1, res=32, spp=1)
    a = _film(OracleScene, host, glass(0.0, 0.0, eta=1.2), depth=1, res=32, spp=1)
    b = _film(OracleScene, host, glass(0.0, 0.0, eta=1.9), depth=1, res=32, spp=1)
    eq_a = (c[0].view(np.uint32) == a[0].view(np.uint32)).all(axis=2)
    eq_b = (c[0].view(np.uint32) == b[0].view(np.uint32)).all(axis=2)
    single = c[1] == 1.0
    assert single.sum() > 900 and (eq_a | eq_b)[single].all()
    assert (eq_a & ~eq_b).any() and (eq_b & ~eq_a).any()


def test_oracle_translucent_null_bsdf_where_reflect_and_transmit_are_black(host):
    """translucent.rs:72-74: where reflect and transmit both evaluate to black the material makes NO BSDF for the hit and the path integrator passes through the surface
    without counting a bounce (path.rs:142-150).  A floor whose reflect and transmit are the same 0 / 1 checkerboard, against (a) the same floor made of Material "none"
    and (b) the translucent of the non-black constants: at depth 1 with one sample per pixel every single-sample pixel equals one of the two films."""
    chk = checker((0.0, 0.0, 0.0), ONE, 2.0)
    c = _film(OracleScene, host, translucent(ONE, ONE, chk, chk), depth=1, res=32, spp=1)
    a = _film(OracleScene, host, translucent(ONE, ONE), depth=1, res=32, spp=1)
    b = _film(OracleScene, host, lambda sc: sc.add_material_none(), depth=1, res=32, spp=1)
    eq_a = (c[0].view(np.uint32) == a[0].view(np.uint32)).all(axis=2)
    eq_b = (c[0].view(np.uint32) == b[0].view(np.uint32)).all(axis=2)
    single = c[1] == 1.0
    assert single.sum() > 900 and (eq_a | eq_b)[single].all()
    assert (eq_a & ~eq_b).any() and (eq_b & ~eq_a).any()
    # reflect = transmit = 0 as constants: the same as Material "none" everywhere
    z = _film(OracleScene, host, translucent((0.0, 0.0, 0.0), (0.0, 0.0, 0.0)), depth=3, res=32, spp=2)
    n = _film(OracleScene, host, lambda sc: sc.add_material_none(), depth=3, res=32, spp=2)
    assert _same(z, n)


@pytest.mark.gpu
def test_refusals_match(host):
    for cls in (pbrt_hip.Scene, OracleScene):
        with cls() as s:
            m = s.add_material_matte((0.5, 0.5, 0.5), 0.0)
            t = s.add_texture_constant(ONE)
            for prm in ("opacity", "amount", "eta", "k"):
                with pytest.raises((pbrt_hip.PbrtHipError, RuntimeError)):
                    s.set_material_texture(m, prm, t)
