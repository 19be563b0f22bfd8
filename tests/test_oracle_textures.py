"""CPU: pins for the oracle's restatement of the image textures (oracle/oracle_texture.hpp) by closed forms and by values
re-derived independently in numpy from the formulas of core/src/mipmap/mod.rs.  The reference holds no golden vectors for
MIPMap (its tests cover only TexInfo hashing), so these closed forms are the pin."""
import numpy as np
import pytest

from oracle_binding import OracleScene
from texture_scenes import make_image, probe_points

f32 = np.float32


def test_pow2_pyramid_is_flipped_image_then_box_averages():
    img = make_image(8, 4, seed=3)
    s = OracleScene(); mp = s.add_mipmap(img)
    pyr = s.mipmap_pyramid(mp)
    assert [p.shape[:2] for p in pyr] == [(4, 8), (2, 4), (1, 2), (1, 1)]   # 1 + log2(max(w, h)) levels, each dimension halved, never below 1
    assert np.array_equal(pyr[0], img[::-1])                                # texture space has (0,0) at the lower left (cache.rs:97-104)
    l1 = ((pyr[0][0::2, 0::2] + pyr[0][0::2, 1::2]) + pyr[0][1::2, 0::2] + pyr[0][1::2, 1::2]) * f32(0.25)
    assert np.array_equal(pyr[1], l1.astype(np.float32))
    # 1 x 2 level from a 2 x 4 one: rows are halved, the single row is fetched twice through the wrap mode (repeat: row 1 -> row... 2t+1 = 1)
    l2 = ((pyr[1][0::2, 0::2] + pyr[1][0::2, 1::2]) + pyr[1][1::2, 0::2] + pyr[1][1::2, 1::2]) * f32(0.25)
    assert np.array_equal(pyr[2], l2.astype(np.float32))


def test_scale_gamma_and_float_conversion():
    img = make_image(4, 4, seed=5)
    s = OracleScene()
    lin = s.mipmap_pyramid(s.add_mipmap(img, scale=0.5))[0]
    assert np.array_equal(lin, (f32(0.5) * img[::-1]).astype(np.float32))
    g = s.mipmap_pyramid(s.add_mipmap(img, gamma=True))[0]
    x = img[::-1].astype(np.float64)
    want = np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4)
    assert np.allclose(g, want, rtol=3e-6, atol=1e-7)                       # inv_gamma_correct (pbrt/common.rs:152-158), f32 powf
    fl = s.mipmap_pyramid(s.add_mipmap(img, as_float=True))[0]
    y = (f32(0.212671) * img[..., 0] + f32(0.715160) * img[..., 1]) + f32(0.072169) * img[..., 2]
    assert np.array_equal(fl[..., 0], y[::-1].astype(np.float32)) and np.array_equal(fl[..., 0], fl[..., 2])


def _lanczos(x, tau=2.0):
    x = abs(f32(x))
    if x < f32(1e-5): return f32(1.0)
    if x > f32(1.0): return f32(0.0)
    x = f32(x * f32(np.pi))
    return f32(f32(np.sin(f32(x * f32(tau)), dtype=f32) / f32(x * f32(tau))) * f32(np.sin(x, dtype=f32) / x))


def _resample_weights(old, new):
    out = []
    for i in range(new):
        center = f32(f32(f32(i) + f32(0.5)) * f32(old)) / f32(new)
        first = max(int(np.floor(f32(f32(center - f32(2.0)) + f32(0.5)))), 0)   # `as usize` saturates: never negative
        w = [_lanczos(f32(f32(f32(f32(first) + f32(j)) + f32(0.5)) - center) / f32(2.0)) for j in range(4)]
        inv = f32(1.0) / f32(f32(f32(w[0] + w[1]) + w[2]) + w[3])
        out.append((first, [f32(x * inv) for x in w]))
    return out


@pytest.mark.parametrize("wrap", ["repeat", "clamp", "black"])
def test_npot_resampling_matches_the_formulas(wrap):
    """3 x 5 -> 4 x 8: s pass then t pass with Lanczos weights, first_texel saturated at 0 (mipmap/mod.rs:383-560)."""
    w, h = 3, 5
    img = make_image(w, h, seed=7)
    s = OracleScene()
    got = s.mipmap_pyramid(s.add_mipmap(img, wrap=wrap))[0]
    assert got.shape[:2] == (8, 4)
    src = img[::-1].astype(np.float32)
    sw = _resample_weights(w, 4)
    mid = np.zeros((h, 4, 3), np.float32)

    def wrapi(i, n):
        return i % n if wrap == "repeat" else (min(i, n - 1) if wrap == "clamp" else i)
    for t in range(h):
        for x in range(4):
            px = np.zeros(3, np.float32)
            for j in range(4):
                o = wrapi(sw[x][0] + j, w)
                if o < w: px = (px + src[t, o] * sw[x][1][j]).astype(np.float32)
            mid[t, x] = px
    tw = _resample_weights(h, 8)
    want = np.zeros((8, 4, 3), np.float32)
    for x in range(4):
        for t in range(8):
            px = np.zeros(3, np.float32)
            for j in range(4):
                o = wrapi(tw[t][0] + j, h)
                if o < h: px = (px + mid[o, x] * tw[t][1][j]).astype(np.float32)
            want[t, x] = np.maximum(px, 0.0)                                  # clamp_default
    assert np.allclose(got, want, rtol=2e-6, atol=1e-7)                        # sinf may differ by an ulp between numpy and libm


def test_bilinear_lookup_closed_forms_and_wrap_modes():
    img = np.zeros((2, 2, 3), np.float32)
    img[0, 0] = (1, 0, 0); img[0, 1] = (0, 1, 0); img[1, 0] = (0, 0, 1); img[1, 1] = (1, 1, 1)   # top row first
    s = OracleScene()
    tex = {w: s.add_texture_imagemap(s.add_mipmap(img, trilinear=True, wrap=w)) for w in ("repeat", "clamp", "black")}
    # texel centres of level 0 (zero footprint -> MIPMap::triangle(0, st)): (0.25, 0.25) is texel (0,0) of the FLIPPED image = img[1,0]
    c = s.texture_eval(tex["clamp"], [[0.25, 0.25], [0.75, 0.25], [0.25, 0.75], [0.75, 0.75], [0.5, 0.5]])
    assert np.array_equal(c[0], img[1, 0]) and np.array_equal(c[1], img[1, 1]) and np.array_equal(c[2], img[0, 0]) and np.array_equal(c[3], img[0, 1])
    assert np.allclose(c[4], img.reshape(4, 3).mean(0))
    # outside [0,1]: clamp repeats the edge texel, black fades to zero, repeat wraps
    out = [[-0.25, 0.25]]
    assert np.array_equal(s.texture_eval(tex["clamp"], out)[0], img[1, 0])
    assert np.array_equal(s.texture_eval(tex["black"], out)[0], np.zeros(3, np.float32))
    assert np.array_equal(s.texture_eval(tex["repeat"], out)[0], s.texture_eval(tex["repeat"], [[0.75, 0.25]])[0])
    assert np.allclose(s.texture_eval(tex["repeat"], [[1.3, 2.6]])[0], s.texture_eval(tex["repeat"], [[0.3, 0.6]])[0], atol=2e-6)


@pytest.mark.parametrize("trilinear", [True, False])
def test_constant_image_is_constant_under_every_filter(trilinear):
    img = np.full((8, 16, 3), (0.2, 0.5, 0.7), np.float32)
    s = OracleScene()
    tex = s.add_texture_imagemap(s.add_mipmap(img, trilinear=trilinear))
    uv, d = probe_points(500, 1)
    out = s.texture_eval(tex, uv, d)
    assert np.allclose(out, (0.2, 0.5, 0.7), rtol=3e-6)


def test_level_selection_of_the_trilinear_filter():
    """A 1-texel checkerboard averages to 0.5 from level 1 on: width 2^-k selects level levels-1-k (mod.rs:222-238)."""
    img = make_image(16, 16, kind="checker")
    s = OracleScene()
    tex = s.add_texture_imagemap(s.add_mipmap(img, trilinear=True))
    uv = np.array([[0.4, 0.6]], np.float32)
    fine = s.texture_eval(tex, uv, [[0, 0, 0, 0]])[0]           # level < 0 -> bilinear on level 0: not grey at a generic point
    assert abs(fine[0] - 0.5) > 0.01
    for width in (2.0 ** -3, 2.0 ** -2, 0.5, 1.0, 4.0):       # levels 1, 2, 3, 4 (coarsest) and beyond
        assert np.allclose(s.texture_eval(tex, uv, [[width, 0, 0, 0]])[0], 0.5, atol=1e-6)
    half = s.texture_eval(tex, uv, [[2.0 ** -3.5, 0, 0, 0]])[0]  # between levels 0 and 1: lerp(0.5) of the two
    l0 = s.texture_eval(tex, uv, [[2.0 ** -4, 0, 0, 0]])[0]      # exactly level 0 with delta 0 -> triangle(0)
    assert np.allclose(l0, fine, atol=1e-6)
    assert np.allclose(half, 0.5 * l0 + 0.25, atol=1e-5)


def test_ewa_degenerate_cases_fall_back_to_bilinear_and_clamp_anisotropy():
    img = make_image(32, 32, seed=11)
    s = OracleScene()
    ewa = s.add_texture_imagemap(s.add_mipmap(img))
    tri = s.add_texture_imagemap(s.add_mipmap(img, trilinear=True))
    uv, _ = probe_points(50, 2)
    d = np.zeros((50, 4), np.float32); d[:, 0] = 0.05                     # one axis of length 0 -> minor_length == 0 -> triangle(0, st)
    assert np.array_equal(s.texture_eval(ewa, uv, d), s.texture_eval(tri, uv, np.zeros((50, 4), np.float32)))
    # a footprint far beyond the image: both EWA levels are past the pyramid -> the single coarsest texel
    coarse = s.mipmap_pyramid(0)[-1][0, 0]
    big = np.tile(np.array([[40.0, 0.0, 0.0, 40.0]], np.float32), (50, 1))
    assert np.allclose(s.texture_eval(ewa, uv, big), coarse, rtol=1e-6)


def test_scale_and_mix_textures():
    s = OracleScene()
    a = s.add_texture_constant((0.2, 0.4, 0.8)); b = s.add_texture_constant((0.5, 0.25, 2.0)); amt = s.add_texture_constant(0.25)
    sc = s.add_texture_scale(a, b); mx = s.add_texture_mix(a, b, amt)
    assert np.array_equal(s.texture_eval(sc, [[0, 0]])[0], (np.array([0.2, 0.4, 0.8], np.float32) * np.array([0.5, 0.25, 2.0], np.float32)))
    want = f32(0.75) * np.array([0.2, 0.4, 0.8], np.float32) + f32(0.25) * np.array([0.5, 0.25, 2.0], np.float32)
    assert np.array_equal(s.texture_eval(mx, [[0, 0]])[0], want.astype(np.float32))
    img = make_image(4, 4, seed=1)
    im = s.add_texture_imagemap(s.add_mipmap(img), su=2.0, sv=0.5, du=0.125, dv=0.25)
    plain = s.add_texture_imagemap(0)
    uv = np.array([[0.3, 0.7]], np.float32)
    st = np.array([[f32(2.0) * f32(0.3) + f32(0.125), f32(0.5) * f32(0.7) + f32(0.25)]], np.float32)   # UVMapping2D (uv_2d.rs:52-60)
    assert np.array_equal(s.texture_eval(im, uv)[0], s.texture_eval(plain, st)[0])
    nested = s.add_texture_scale(s.add_texture_mix(im, a, amt), sc)
    assert np.allclose(s.texture_eval(nested, uv)[0], (0.75 * s.texture_eval(im, uv)[0] + 0.25 * np.array([0.2, 0.4, 0.8])) * s.texture_eval(sc, uv)[0], rtol=1e-6)


def test_a_constant_texture_is_the_constant_parameter():
    """set_material_texture with a ConstantTexture must be indistinguishable from passing the constant: same lobes, same film (the
    per-hit lobe list degenerates to the material's own list)."""
    import pbrt_hip
    from texture_scenes import textured_quad_scene
    host = pbrt_hip.Host()

    def film(material):
        s = OracleScene()
        textured_quad_scene(s, host, lambda sc: sc.add_texture_constant((0.3, 0.6, 0.2)), res=24, spp=2, material=material)
        xyz, wt, st, _ = s.render_path_ex(max_depth=3)
        return xyz

    def plastic_tex(sc, tex):
        m = sc.add_material_plastic((1, 1, 1), (1, 1, 1), 0.1, True)
        sc.set_material_texture(m, "Kd", tex); sc.set_material_texture(m, "Ks", tex); return m
    assert np.array_equal(film(plastic_tex), film(lambda sc, tex: sc.add_material_plastic((0.3, 0.6, 0.2), (0.3, 0.6, 0.2), 0.1, True)))

    def substrate_tex(sc, tex):
        m = sc.add_material_substrate((1, 1, 1), (1, 1, 1), 0.2, 0.1, True)
        sc.set_material_texture(m, "Kd", tex); sc.set_material_texture(m, "Ks", tex); return m
    assert np.array_equal(film(substrate_tex), film(lambda sc, tex: sc.add_material_substrate((0.3, 0.6, 0.2), (0.3, 0.6, 0.2), 0.2, 0.1, True)))

    def mirror_tex(sc, tex):
        m = sc.add_material_mirror((1, 1, 1)); sc.set_material_texture(m, "Kr", tex); return m
    assert np.array_equal(film(mirror_tex), film(lambda sc, tex: sc.add_material_mirror((0.3, 0.6, 0.2))))
    assert np.array_equal(film(lambda sc, tex: sc.add_material_matte_tex(tex, 15.0)), film(lambda sc, tex: sc.add_material_matte((0.3, 0.6, 0.2), 15.0)))
    def uber_tex(sc, tex):
        m = sc.add_material_uber((1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1), (0.8, 0.7, 0.9), 0.1, 0.2, 1.4, True)
        for prm in ("Kd", "Ks", "Kr", "Kt"): sc.set_material_texture(m, prm, tex)
        return m
    c = (0.3, 0.6, 0.2)
    assert np.array_equal(film(uber_tex), film(lambda sc, tex: sc.add_material_uber(c, c, c, c, (0.8, 0.7, 0.9), 0.1, 0.2, 1.4, True)))
    for rough in (0.0, 0.05):
        def glass_tex(sc, tex, rough=rough):
            m = sc.add_material_glass((1, 1, 1), (1, 1, 1), rough, rough, 1.5, True)
            sc.set_material_texture(m, "Kr", tex); sc.set_material_texture(m, "Kt", tex); return m
        assert np.array_equal(film(glass_tex), film(lambda sc, tex, rough=rough: sc.add_material_glass(c, c, rough, rough, 1.5, True)))
    # scalar parameters: a constant float texture is the constant (the per-hit remap uses the same polynomial as the constructor)
    def fconst(v): return lambda sc: sc.add_texture_constant(v)
    def plastic_rough(sc, tex):
        m = sc.add_material_plastic(c, c, 0.5, True); sc.set_material_float_texture(m, "roughness", sc.add_texture_constant(0.15)); return m
    assert np.array_equal(film(plastic_rough), film(lambda sc, tex: sc.add_material_plastic(c, c, 0.15, True)))
    def matte_sigma(sc, tex):
        m = sc.add_material_matte(c, 0.0); sc.set_material_float_texture(m, "sigma", sc.add_texture_constant(35.0)); return m
    assert np.array_equal(film(matte_sigma), film(lambda sc, tex: sc.add_material_matte(c, 35.0)))
    def matte_sigma0(sc, tex):
        m = sc.add_material_matte(c, 20.0); sc.set_material_float_texture(m, "sigma", sc.add_texture_constant(0.0)); return m
    assert np.array_equal(film(matte_sigma0), film(lambda sc, tex: sc.add_material_matte(c, 0.0)))
    def metal_uv(sc, tex):
        m = sc.add_material_metal((0.2, 0.9, 1.1), (3.9, 2.4, 2.1), 0.3, 0.3, False)
        sc.set_material_float_texture(m, "uroughness", sc.add_texture_constant(0.05)); sc.set_material_float_texture(m, "vroughness", sc.add_texture_constant(0.2)); return m
    assert np.array_equal(film(metal_uv), film(lambda sc, tex: sc.add_material_metal((0.2, 0.9, 1.1), (3.9, 2.4, 2.1), 0.05, 0.2, False)))
    # a black constant texture removes the lobe exactly as a black constant does
    def black(material):
        s = OracleScene()
        textured_quad_scene(s, host, lambda sc: sc.add_texture_constant((0.0, 0.0, 0.0)), res=24, spp=2, material=material)
        return s.render_path_ex(max_depth=3)[0]
    def plastic_kd_black(sc, tex):
        m = sc.add_material_plastic((1, 1, 1), (0.2, 0.2, 0.2), 0.1, True); sc.set_material_texture(m, "Kd", tex); return m
    assert np.array_equal(black(plastic_kd_black), black(lambda sc, tex: sc.add_material_plastic((0, 0, 0), (0.2, 0.2, 0.2), 0.1, True)))


def test_procedural_2d_textures_closed_forms():
    s = OracleScene()
    a = s.add_texture_constant((1.0, 0.5, 0.25)); b = s.add_texture_constant((0.0, 0.2, 0.4))
    # checkerboard, point sampled: (floor(s) + floor(t)) % 2 == 0 -> tex1; Rust's % keeps the sign, so (-1 + 0) % 2 = -1 selects tex2
    ck = s.add_texture_checkerboard(a, b, su=4.0, sv=4.0, aa="none")
    uv = np.array([[0.1, 0.1], [0.3, 0.1], [0.3, 0.3], [-0.1, 0.1], [-0.1, -0.1], [-0.3, 0.1]], np.float32)
    got = s.texture_eval(ck, uv)
    want = [(1.0, 0.5, 0.25), (0.0, 0.2, 0.4), (1.0, 0.5, 0.25), (0.0, 0.2, 0.4), (1.0, 0.5, 0.25), (1.0, 0.5, 0.25)]
    assert np.array_equal(got, np.array(want, np.float32))
    # closed form: a filter much wider than a check (ds > 1) averages to one half; one inside a check is the point sample
    cf = s.add_texture_checkerboard(a, b, su=4.0, sv=4.0)
    wide = s.texture_eval(cf, [[0.37, 0.41]], [[0.6, 0, 0, 0.6]])[0]            # ds = dt = 2.4 > 1
    assert np.allclose(wide, 0.5 * (np.array(want[0]) + np.array(want[1])), rtol=1e-6)
    assert np.array_equal(s.texture_eval(cf, [[0.1, 0.1]], [[0.001, 0, 0, 0.001]])[0], np.array(want[0], np.float32))
    # box filter centred on a vertical check edge (s = 1), well inside the row: exactly half of each
    edge = s.texture_eval(cf, [[0.25, 0.125]], [[0.05, 0, 0, 0.01]])[0]         # st = (1.0, 0.5), ds = 0.2, dt = 0.04
    assert np.allclose(edge, 0.5 * (np.array(want[0]) + np.array(want[1])), atol=2e-6)
    # uv texture and bilerp
    uvt = s.add_texture_uv(su=2.0, sv=3.0, du=0.25, dv=0.5)
    g = s.texture_eval(uvt, [[0.7, 0.9]])[0]
    st = (np.float32(2.0) * np.float32(0.7) + np.float32(0.25), np.float32(3.0) * np.float32(0.9) + np.float32(0.5))
    assert np.array_equal(g, np.array([st[0] - np.floor(st[0]), st[1] - np.floor(st[1]), 0.0], np.float32))
    bl = s.add_texture_bilerp((1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1))
    assert np.array_equal(s.texture_eval(bl, [[0, 0], [0, 1], [1, 0], [1, 1]]), np.array([(1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1)], np.float32))
    assert np.allclose(s.texture_eval(bl, [[0.5, 0.5]])[0], (0.5, 0.5, 0.5))


def test_noise_table_and_dots():
    """The permutation behind DotsTexture must be the reference's (compared with its text when the tree is present); Perlin noise vanishes
    on the integer lattice and is bounded; dots: every value is one of the two operands and both occur."""
    import os, re
    here = os.path.dirname(os.path.abspath(__file__))
    mine = open(os.path.join(here, "..", "oracle", "oracle_texture.hpp")).read()
    tab = [int(x) for x in re.findall(r"\d+", mine[mine.index("NOISE_PERM[512] = {"):].split("};")[0].split("{", 1)[1])]
    assert len(tab) == 512 and tab[:256] == tab[256:] and sorted(tab[:256]) == list(range(256))
    dev = open(os.path.join(here, "..", "pbrt-v3-rs_amd", "csrc", "texture.h")).read()
    dtab = [int(x) for x in re.findall(r"\d+", dev[dev.index("kNoisePerm[512] = {"):].split("};")[0].split("{", 1)[1])]
    assert dtab == tab
    ref = "/root/reference/core/src/texture/common.rs"
    if os.path.exists(ref):
        txt = open(ref).read()
        rtab = [int(x) for x in re.findall(r"\d+", txt[txt.index("NOISE_PERM: [usize; 2 * NOISE_PERM_SIZE] = ["):].split("];")[0].split("= [", 1)[1])]
        assert rtab == tab
    s = OracleScene()
    a = s.add_texture_constant((1.0, 1.0, 1.0)); b = s.add_texture_constant((0.0, 0.0, 0.0))
    dots = s.add_texture_dots(a, b, su=10.0, sv=10.0)
    rng = np.random.default_rng(3)
    v = s.texture_eval(dots, rng.uniform(0, 1, (4000, 2)).astype(np.float32))[:, 0]
    assert set(np.unique(v)) == {0.0, 1.0} and 0.05 < v.mean() < 0.5


def _perm():
    import os, re
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "oracle_texture.hpp")).read()
    return [int(x) for x in re.findall(r"\d+", src[src.index("NOISE_PERM[512] = {"):].split("};")[0].split("{", 1)[1])]


def _noise3(x, y, z, P):
    """Perlin noise of core/src/texture/common.rs:37-117 in numpy float32 scalars (an independent second implementation)."""
    x, y, z = f32(x), f32(y), f32(z)
    ix, iy, iz = int(np.floor(x)), int(np.floor(y)), int(np.floor(z))
    dx, dy, dz = f32(x - f32(ix)), f32(y - f32(iy)), f32(z - f32(iz))
    ix &= 255; iy &= 255; iz &= 255

    def grad(a, b, c, u_, v_, w_):
        h = P[P[P[a] + b] + c] & 15
        u = u_ if (h < 8 or h in (12, 13)) else v_
        v = v_ if (h < 4 or h in (12, 13)) else w_
        return f32((-u if h & 1 else u) + (-v if h & 2 else v))

    def wgt(t):
        t3 = f32(f32(t * t) * t); t4 = f32(t3 * t)
        return f32(f32(f32(f32(6.0) * t4) * t - f32(f32(15.0) * t4)) + f32(f32(10.0) * t3))

    def lerp(t, a, b):
        return f32(f32(f32(f32(1.0) - t) * a) + f32(t * b))
    one = f32(1.0)
    w = [[[grad(ix + i, iy + j, iz + k, f32(dx - i * one) if i else dx, f32(dy - j * one) if j else dy, f32(dz - k * one) if k else dz) for k in (0, 1)] for j in (0, 1)] for i in (0, 1)]
    wx, wy, wz = wgt(dx), wgt(dy), wgt(dz)
    x00, x10, x01, x11 = lerp(wx, w[0][0][0], w[1][0][0]), lerp(wx, w[0][1][0], w[1][1][0]), lerp(wx, w[0][0][1], w[1][0][1]), lerp(wx, w[0][1][1], w[1][1][1])
    return lerp(wz, lerp(wy, x00, x10), lerp(wy, x01, x11))


def test_fbm_turbulence_windy_marble_against_a_second_implementation():
    P = _perm()
    s = OracleScene()
    rng = np.random.default_rng(4)
    pts = rng.uniform(-3, 3, (12, 3)).astype(np.float32)
    big = np.tile(np.array([[10.0, 0, 0]], np.float32), (12, 1))      # |dp/dx|^2 = 100 -> n = clamp(-1 - 0.5 log2(100)) = 0: only the partial octave, weight smooth_step(0) = 0
    fbm = s.add_texture_fbm(omega=0.6, octaves=5); wr = s.add_texture_fbm(omega=0.6, octaves=5, wrinkled=True)
    uv0 = np.zeros((12, 2), np.float32)
    assert np.array_equal(s.texture_eval(fbm, uv0, p=pts, dpdx=big, dpdy=big)[:, 0], np.zeros(12, np.float32))
    # zero footprint: log2(0) = -inf -> n = octaves: the full sum, and the partial term has weight smooth_step(0.3, 0.7, 0) = 0
    got_f = s.texture_eval(fbm, uv0, p=pts)[:, 0]; got_w = s.texture_eval(wr, uv0, p=pts)[:, 0]
    for q, gf, gw in zip(pts, got_f, got_w):
        sf = sw = f32(0.0); lam = o = f32(1.0)
        for _ in range(5):
            nz = _noise3(lam * q[0], lam * q[1], lam * q[2], P)
            sf = f32(sf + f32(o * nz)); sw = f32(sw + f32(o * abs(nz)))
            lam = f32(lam * f32(1.99)); o = f32(o * f32(0.6))
        nz = _noise3(lam * q[0], lam * q[1], lam * q[2], P)
        sf = f32(sf + f32(f32(o * f32(0.0)) * nz))
        sw = f32(sw + f32(o * f32(f32(f32(1.0) * f32(0.2)) + f32(f32(0.0) * abs(nz)))))   # lerp(0, 0.2, |noise|)
        assert gf == sf and gw == sw
    # noise vanishes on the integer lattice (one octave, lambda = 1)
    lattice = np.array([[1, 2, 3], [-4, 0, 7], [255, 256, -256]], np.float32)
    one = s.add_texture_fbm(omega=0.5, octaves=1)
    assert np.array_equal(s.texture_eval(one, np.zeros((3, 2), np.float32), p=lattice)[:, 0], np.zeros(3, np.float32))
    # the mapping matrix is applied as given: scaling the point by 2 through the matrix equals evaluating at 2p
    m = np.diag([2.0, 2.0, 2.0, 1.0]).astype(np.float32).reshape(16)
    scaled = s.add_texture_fbm(m=m, omega=0.6, octaves=5)
    assert np.array_equal(s.texture_eval(scaled, uv0, p=pts)[:, 0], s.texture_eval(fbm, uv0, p=(pts * np.float32(2.0)).astype(np.float32))[:, 0])
    # windy = |fbm(0.1 p, 0.5, 3)| * fbm(p, 0.5, 6); marble stays inside its spline's hull (x 1.5)
    windy = s.add_texture_windy()
    f3_ = s.add_texture_fbm(omega=0.5, octaves=3); f6 = s.add_texture_fbm(omega=0.5, octaves=6)
    w_ = s.texture_eval(windy, uv0, p=pts)[:, 0]
    a = s.texture_eval(f3_, uv0, p=(np.float32(0.1) * pts).astype(np.float32))[:, 0]; b = s.texture_eval(f6, uv0, p=pts)[:, 0]
    assert np.array_equal(w_, (np.abs(a) * b).astype(np.float32))
    # marble: t = 0.5 + 0.5 sin(scale p.y + variation fbm(scale p)), then the Bezier spline with `first = min(1, floor(6 t))` — the reference's (and
    # C++ pbrt's) clamp, which lets the local parameter run up to 5 and the spline extrapolate; recomputed here in float64 from the oracle's own fbm
    mar = s.texture_eval(s.add_texture_marble(omega=0.5, octaves=8, scale=3.0, variation=0.4), uv0, p=pts)
    f8 = s.texture_eval(s.add_texture_fbm(m=np.diag([3.0, 3.0, 3.0, 1.0]).astype(np.float32).reshape(16), omega=0.5, octaves=8), uv0, p=pts)[:, 0]
    C = np.array([[0.58, 0.58, 0.6], [0.58, 0.58, 0.6], [0.58, 0.58, 0.6], [0.5, 0.5, 0.5], [0.6, 0.59, 0.58], [0.58, 0.58, 0.6], [0.58, 0.58, 0.6], [0.2, 0.2, 0.33], [0.58, 0.58, 0.6]])
    for q, fb, got in zip(pts.astype(np.float64), f8.astype(np.float64), mar):
        t = 0.5 + 0.5 * np.sin(3.0 * q[1] + 0.4 * fb)
        first = min(1, int(np.floor(t * 6)))
        t = t * 6 - first
        c = [C[first + k] for k in range(4)]
        s0, s1, s2 = (1 - t) * c[0] + t * c[1], (1 - t) * c[1] + t * c[2], (1 - t) * c[2] + t * c[3]
        s0, s1 = (1 - t) * s0 + t * s1, (1 - t) * s1 + t * s2
        assert np.allclose(got, 1.5 * ((1 - t) * s0 + t * s1), rtol=2e-4, atol=1e-4)
    # 3D checkerboard: parity of floor(x) + floor(y) + floor(z), Rust's % keeping the sign
    a_ = s.add_texture_constant(1.0); b_ = s.add_texture_constant(0.0)
    c3 = s.add_texture_checkerboard3d(a_, b_)
    cp = np.array([[0.5, 0.5, 0.5], [1.5, 0.5, 0.5], [1.5, 1.5, 0.5], [-0.5, 0.5, 0.5], [-0.5, -0.5, 0.5]], np.float32)
    assert list(s.texture_eval(c3, np.zeros((5, 2), np.float32), p=cp)[:, 0]) == [1.0, 0.0, 1.0, 0.0, 1.0]


def test_spherical_cylindrical_planar_mappings():
    s = OracleScene()
    ident = np.eye(4, dtype=np.float32).reshape(16)
    uvt = [s.add_texture_uv() for _ in range(3)]
    s.set_texture_mapping(uvt[0], "spherical", ident); s.set_texture_mapping(uvt[1], "cylindrical", ident)
    s.set_texture_mapping(uvt[2], "planar", [0.5, 0, 0, 0, 0.25, 0, 0.1, 0.2])
    pts = np.array([[0, 0, 2.0], [1, 0, 0], [0, 3, 0], [-1, -1, 0.5], [0.3, -0.7, -0.2]], np.float32)
    z = np.zeros((5, 2), np.float32)
    sph = s.texture_eval(uvt[0], z, p=pts); cyl = s.texture_eval(uvt[1], z, p=pts); pla = s.texture_eval(uvt[2], z, p=pts)
    v = pts.astype(np.float64) / np.linalg.norm(pts.astype(np.float64), axis=1, keepdims=True)
    theta = np.arccos(np.clip(v[:, 2], -1, 1)) / np.pi
    phi = np.arctan2(v[:, 1], v[:, 0]); phi = np.where(phi < 0, phi + 2 * np.pi, phi) / (2 * np.pi)
    frac = lambda a: a - np.floor(a)
    assert np.allclose(sph[:, 0], frac(theta), atol=1e-6) and np.allclose(sph[:, 1], frac(phi), atol=1e-6)
    assert np.allclose(cyl[:, 0], frac((np.pi + np.arctan2(v[:, 1], v[:, 0])) / (2 * np.pi)), atol=1e-6) and np.allclose(cyl[:, 1], frac(v[:, 2]), atol=1e-6)
    assert np.allclose(pla[:, 0], frac(0.1 + 0.5 * pts[:, 0]), atol=1e-6) and np.allclose(pla[:, 1], frac(0.2 + 0.25 * pts[:, 1]), atol=1e-6)
    # the matrix is applied to the point before the projection
    m = np.array([[0, 0, 1, 0], [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32).reshape(16)   # (x,y,z) -> (z,x,y)
    t2 = s.add_texture_uv(); s.set_texture_mapping(t2, "spherical", m)
    assert np.array_equal(s.texture_eval(t2, z, p=pts), s.texture_eval(uvt[0], z, p=pts[:, [2, 0, 1]].copy()))
    # only 2D textures take one
    with pytest.raises(Exception):
        s.set_texture_mapping(s.add_texture_fbm(), "planar", [1, 0, 0, 0, 1, 0, 0, 0])


def test_bump_mapping_pins():
    """A constant displacement on a flat untangented mesh changes nothing (the differences vanish and dn/du = dn/dv = 0); a displacement that is
    linear in u tilts the shading normal by exactly atan(slope / |dp/du|): checked through the radiance of a mirror looking at a gradient sky is
    overkill — the shading normal is observable through a Lambertian floor under a distant light: L = Kd/pi * E * cos(theta)."""
    import pbrt_hip
    host = pbrt_hip.Host()

    def floor_radiance(bump):
        s = OracleScene()
        mat = s.add_material_matte((0.5, 0.5, 0.5), 0.0)
        if bump is not None: s.set_material_bump(mat, bump(s))
        P = np.array([[-50, -50, 0], [50, -50, 0], [50, 50, 0], [-50, 50, 0]], np.float32)
        UV = np.array([[0, 0], [100, 0], [100, 100], [0, 100]], np.float32)      # |dp/du| = |dp/dv| = 1
        s.add_mesh(P, [0, 1, 2, 0, 2, 3], mat, UV=UV)
        s.add_light_distant((3.0, 3.0, 3.0), (0.0, 0.0, 1.0))                    # light arriving straight down
        w2c, c2w = host.look_at((0, 0, 5), (0, 0, 0), (0, 1, 0))
        s.set_camera_perspective(host.perspective_raster_to_camera(10.0, 8, 8), c2w)
        cb, table, sb = host.film_box(8, 8)
        s.set_film(8, 8, cb, (0.5, 0.5), table); s.set_sampler(0, 1, sb); s.build_accel(0, 4)
        xyz, wt, _, _ = s.render_path_ex(max_depth=1, light_strategy=0)
        return s.film_to_rgb(xyz, wt).reshape(8, 8, 3)[2:6, 2:6, 1].mean()
    flat = floor_radiance(None)
    assert np.isclose(flat, 0.5 / np.pi * 3.0, rtol=1e-5)
    assert floor_radiance(lambda s: s.add_texture_constant(0.7)) == flat
    # displacement d = k * u: dp/du' = dp/du + k n  ->  the shading normal tilts by atan(k); the shading normal only enters |wi . ns|
    # in estimate_direct, while the Lambertian f and the light are unchanged: L = flat * cos(atan(k))
    for k in (0.5, 2.0):
        ramp = lambda s, k=k: s.add_texture_bilerp(0.0, 0.0, k, k)                # v00, v01, v10, v11: value = k * u on [0,1]^2 ... and beyond, linearly
        got = floor_radiance(ramp)
        assert np.isclose(got, flat * np.cos(np.arctan(k)), rtol=2e-4), (got, flat * np.cos(np.arctan(k)))


def test_environment_map_light_closed_forms():
    """InfiniteAreaLight with a radiance map: (1) a constant map is the constant light (same film); (2) a white diffuse floor under a map that is
    `a` on the upper hemisphere's... simpler: under a map with radiance depending on theta only, E = 2 pi int L(theta) cos sin dtheta; with the
    two-band map below (L = 2 for theta < pi/2 band boundaries on texel rows) the floor's radiance is Kd/pi * E, checked to Monte Carlo accuracy;
    (3) sample_li / pdf_li consistency is implied by (2) under MIS: both estimators meet the same closed form."""
    import pbrt_hip
    host = pbrt_hip.Host()

    def film(add_light, spp=64, depth=1):
        s = OracleScene()
        mat = s.add_material_matte((0.8, 0.8, 0.8), 0.0)
        P = np.array([[-50, -50, 0], [50, -50, 0], [50, 50, 0], [-50, 50, 0]], np.float32)
        s.add_mesh(P, [0, 1, 2, 0, 2, 3], mat)
        add_light(s)
        w2c, c2w = host.look_at((0, 0, 5), (0, 0, 0), (0, 1, 0))
        s.set_camera_perspective(host.perspective_raster_to_camera(10.0, 8, 8), c2w)
        cb, table, sb = host.film_box(8, 8)
        s.set_film(8, 8, cb, (0.5, 0.5), table); s.set_sampler(0, spp, sb); s.build_accel(0, 4)
        xyz, wt, _, _ = s.render_path_ex(max_depth=depth, light_strategy=0)
        return s.film_to_rgb(xyz, wt).reshape(8, 8, 3)
    const = np.full((4, 8, 3), (0.5, 1.0, 2.0), np.float32)
    a = film(lambda s: s.add_light_infinite_map((2.0, 1.0, 0.5), const), spp=256)[2:6, 2:6].mean(axis=(0, 1))
    b = film(lambda s: s.add_light_infinite((1.0, 1.0, 1.0)), spp=256)[2:6, 2:6].mean(axis=(0, 1))
    # texels * L = (1, 1, 1) everywhere: the same light, sampled through a finer table -> equal in expectation (white furnace: Kd * L = 0.8)
    assert np.allclose(a, 0.8, rtol=0.02) and np.allclose(b, 0.8, rtol=0.02)
    # rows of the map are bands of theta = [k pi/H, (k+1) pi/H); light_to_world = identity: theta measured from +z.  Upper hemisphere = rows 0..H/2-1.
    H, W = 16, 32
    img = np.zeros((H, W, 3), np.float32)
    img[:4] = 3.0                                # theta in [0, pi/4): a bright cap around the zenith
    img[4:8] = 0.5                               # theta in [pi/4, pi/2)
    got = film(lambda s: s.add_light_infinite_map((1.0, 1.0, 1.0), img), spp=256)[2:6, 2:6, 1].mean()
    # E = 2 pi [3 int_0^{pi/4} cos sin + 0.5 int_{pi/4}^{pi/2} cos sin] = 2 pi [3 * 0.25 + 0.5 * 0.25]; bilinear lookups blur the band edges by half a
    # texel each way, symmetric to first order
    want = 0.8 / np.pi * 2 * np.pi * (3 * 0.25 + 0.5 * 0.25)
    assert np.isclose(got, want, rtol=0.03), (got, want)
    # rotating the light by 180 degrees about x puts the bright cap below the floor: only nothing from above the horizon -> black
    t = host.rotate(180.0, [1, 0, 0])
    dark = film(lambda s: s.add_light_infinite_map((1.0, 1.0, 1.0), img, t[0], t[1]), spp=16)[2:6, 2:6, 1].mean()
    assert dark < 0.02 * want
