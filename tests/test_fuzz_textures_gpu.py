"""-m gpu: randomized differential test of the texture system (incl. alpha-mask textures inside the traversal) on top of tests/test_fuzz_gpu.py's scene generator: random texture trees
(image maps of random size / filter / wrap / gamma, every procedural class, scale / mix nesting, all mappings) on the colour parameters that
take them, random bump maps, an optional radiance-map sky.  Film and counters must equal the oracle's bit for bit."""
import numpy as np
import pytest

import pbrt_hip
import scenes
from oracle_binding import OracleScene, set_libm_mode
import test_fuzz_gpu as F

pytestmark = pytest.mark.gpu


def random_image(rng, hdr=False):
    h, w = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    img = rng.uniform(0.0, 4.0 if hdr else 1.0, (h, w, 3)).astype(np.float32)
    if rng.integers(0, 3) == 0:
        img[rng.uniform(0, 1, (h, w)) < 0.3] = 0.0      # exactly black texels: lobes vanish there
    return img


def random_mapping(s, host, rng, tex):
    k = int(rng.integers(0, 6))
    if k == 3:
        s.set_texture_mapping(tex, "spherical", F.random_transform(host, rng)[1])
    elif k == 4:
        s.set_texture_mapping(tex, "cylindrical", F.random_transform(host, rng)[1])
    elif k == 5:
        s.set_texture_mapping(tex, "planar", np.concatenate([rng.uniform(-1, 1, 6), rng.uniform(-0.5, 0.5, 2)]).astype(np.float32))
    return tex


def random_texture(s, host, rng, as_float, depth=0):
    """A random texture tree of the requested value type (float textures: three equal channels)."""
    m = F.random_transform(host, rng)[0]
    uvp = dict(su=float(rng.uniform(0.3, 5)), sv=float(rng.uniform(0.3, 5)), du=float(rng.uniform(-1, 1)), dv=float(rng.uniform(-1, 1)))
    k = int(rng.integers(0, 12 if depth < 2 else 9))

    def const():
        return s.add_texture_constant(float(rng.uniform(0, 1)) if as_float else tuple(rng.uniform(0, 1, 3)))
    if k == 0:
        return const()
    if k in (1, 2):
        mip = s.add_mipmap(random_image(rng), as_float=as_float, scale=float(rng.choice([1.0, 0.5])), gamma=bool(rng.integers(0, 2)), trilinear=bool(rng.integers(0, 2)),
                           wrap=str(rng.choice(["repeat", "black", "clamp"])), max_anisotropy=float(rng.choice([8.0, 2.0])))
        return random_mapping(s, host, rng, s.add_texture_imagemap(mip, **uvp))
    if k == 3:
        sub = (lambda: random_texture(s, host, rng, as_float, depth + 1)) if depth < 2 else const
        return random_mapping(s, host, rng, s.add_texture_checkerboard(sub(), sub(), aa=str(rng.choice(["none", "closedform"])), **uvp))
    if k == 4:
        if as_float: return s.add_texture_bilerp(*[float(v) for v in rng.uniform(0, 1, 4)], **uvp)
        return random_mapping(s, host, rng, s.add_texture_uv(**uvp))
    if k == 5:
        return random_mapping(s, host, rng, s.add_texture_dots(const(), const(), **uvp))
    if k == 6:
        return s.add_texture_fbm(m, float(rng.uniform(0.3, 0.7)), int(rng.integers(0, 8)), wrinkled=bool(rng.integers(0, 2)))
    if k == 7:
        return s.add_texture_windy(m) if as_float or rng.integers(0, 2) else s.add_texture_marble(m, 0.5, int(rng.integers(1, 8)), float(rng.uniform(0.5, 3)), float(rng.uniform(0, 0.5)))
    if k == 8:
        return s.add_texture_checkerboard3d(const(), const(), m)
    if k == 9:
        return s.add_texture_scale(random_texture(s, host, rng, as_float, depth + 1), random_texture(s, host, rng, as_float, depth + 1))
    if k == 10:
        return s.add_texture_mix(random_texture(s, host, rng, as_float, depth + 1), const(), random_texture(s, host, rng, True, depth + 2))
    return s.add_texture_bilerp(*([float(v) for v in rng.uniform(0, 1, 4)] if as_float else [tuple(rng.uniform(0, 1, 3)) for _ in range(4)]), **uvp)


def random_mask(s, host, rng):
    """A float texture that is exactly 0 somewhere (alpha masks reject only where the value is exactly 0)."""
    k = int(rng.integers(0, 4))
    uvp = dict(su=float(rng.uniform(1, 6)), sv=float(rng.uniform(1, 6)))
    if k == 0:
        return s.add_texture_checkerboard(s.add_texture_constant(1.0), s.add_texture_constant(0.0), aa="none", **uvp)
    if k == 1:
        return s.add_texture_dots(s.add_texture_constant(0.0), s.add_texture_constant(1.0), **uvp)
    if k == 2:
        m = (rng.uniform(0, 1, (int(rng.integers(2, 12)), int(rng.integers(2, 12)))) > 0.5).astype(np.float32)
        return s.add_texture_imagemap(s.add_mipmap(np.repeat(m[..., None], 3, axis=2), as_float=True, trilinear=True, wrap=str(rng.choice(["repeat", "black", "clamp"]))), **uvp)
    return s.add_texture_checkerboard3d(s.add_texture_constant(0.0), s.add_texture_constant(1.0), F.random_transform(host, rng)[0])


def textured_material(s, host, rng):
    one = (1, 1, 1)
    c = lambda lo=0.0, hi=1.0: tuple(rng.uniform(lo, hi, 3))
    k = int(rng.integers(0, 9))
    tex = lambda: random_texture(s, host, rng, False)
    ftex = lambda scale: s.add_texture_scale(random_texture(s, host, rng, True), s.add_texture_constant(float(scale)))
    if k == 0:
        m = s.add_material_matte_tex(tex(), float(rng.choice([0.0, rng.uniform(1, 60)])))
        if rng.integers(0, 3) == 0: s.set_material_float_texture(m, "sigma", ftex(60.0))
    elif k == 1:
        m = s.add_material_plastic(one, one, float(rng.uniform(0.01, 0.4)), bool(rng.integers(0, 2)))
        s.set_material_texture(m, "Kd", tex())
        if rng.integers(0, 2): s.set_material_texture(m, "Ks", tex())
        if rng.integers(0, 3) == 0: s.set_material_float_texture(m, "roughness", ftex(0.5))
    elif k == 2:
        m = s.add_material_mirror(one); s.set_material_texture(m, "Kr", tex())
    elif k == 3:
        m = s.add_material_substrate(one, one, float(rng.uniform(0.02, 0.4)), float(rng.uniform(0.02, 0.4)), bool(rng.integers(0, 2)))
        s.set_material_texture(m, "Kd", tex()); s.set_material_texture(m, "Ks", tex())
        if rng.integers(0, 3) == 0: s.set_material_float_texture(m, str(rng.choice(["uroughness", "vroughness"])), ftex(0.5))
    elif k == 4:
        rough = float(rng.choice([0.0, rng.uniform(0.02, 0.3)]))
        m = s.add_material_glass(one, one, rough, rough, float(rng.uniform(1.1, 1.8)), bool(rng.integers(0, 2)))
        s.set_material_texture(m, "Kr", tex()); s.set_material_texture(m, "Kt", tex())
        if rng.integers(0, 2):   # roughness textures: a checkerboard of 0 and a rough value makes both lobe structures appear in one image (glass.rs:110-141)
            zero_or = lambda: s.add_texture_checkerboard(s.add_texture_constant(0.0), s.add_texture_constant(float(rng.uniform(0.05, 0.4))), su=float(rng.uniform(2, 9)), sv=float(rng.uniform(2, 9)), aa="none")
            which = int(rng.integers(0, 3))
            if which != 1: s.set_material_float_texture(m, "uroughness", zero_or() if rng.integers(0, 2) else ftex(0.4))
            if which != 0: s.set_material_float_texture(m, "vroughness", zero_or() if rng.integers(0, 2) else ftex(0.4))
        if rng.integers(0, 3) == 0:   # glass.rs:102: the index of refraction of every hit
            s.set_material_float_texture(m, "index", s.add_texture_mix(s.add_texture_constant(float(rng.uniform(1.05, 1.4))), s.add_texture_constant(float(rng.uniform(1.5, 2.4))), ftex(1.0)))
    elif k == 5:
        op = float(rng.choice([1.0, rng.uniform(0.3, 0.9)]))
        m = s.add_material_uber(one, one, one, one, (op, op, op), float(rng.uniform(0.02, 0.3)), float(rng.uniform(0.02, 0.3)), float(rng.uniform(1.1, 1.7)), True)
        for prm in ("Kd", "Ks", "Kr", "Kt"):
            if rng.integers(0, 2): s.set_material_texture(m, prm, tex())
        if rng.integers(0, 3) == 0: s.set_material_float_texture(m, "uroughness", ftex(0.4))
        if rng.integers(0, 2): s.set_material_texture(m, "opacity", tex())   # uber.rs:126-160: pass-through lobe, colours and BSDF::eta per hit
        if rng.integers(0, 3) == 0:   # uber.rs:128: `e` per hit (after the opacity: a constant one becomes a per-hit constant)
            s.set_material_float_texture(m, "index", s.add_texture_checkerboard(s.add_texture_constant(float(rng.uniform(1.1, 1.4))), s.add_texture_constant(float(rng.uniform(1.5, 2.0))), su=float(rng.uniform(2, 9)), sv=float(rng.uniform(2, 9)), aa="none"))
    elif k == 6 or k == 7:
        def leaf(rough_ok):
            refl, trans = c(), c()
            if rng.integers(0, 4) == 0: refl = (0.0, 0.0, 0.0)
            elif rng.integers(0, 4) == 0: trans = (0.0, 0.0, 0.0)
            if rng.integers(0, 4) == 0: refl = (refl[0], 0.0, 0.0)          # products with black channels
            t = s.add_material_translucent(one, one if rng.integers(0, 2) else (0.0, 0.0, 0.0), refl, trans, float(rng.uniform(0.02, 0.4)), bool(rng.integers(0, 2)))
            s.set_material_texture(t, "Kd", tex())
            try:
                if rng.integers(0, 2): s.set_material_texture(t, "Ks", tex())
            except (pbrt_hip.PbrtHipError, RuntimeError) as e:
                if "live values" in str(e): raise                           # else: Ks was black, no lobe to feed -- refused identically by both libraries
            if rough_ok and rng.integers(0, 3) == 0:
                try: s.set_material_float_texture(t, "roughness", ftex(0.5))
                except (pbrt_hip.PbrtHipError, RuntimeError) as e:
                    if "live values" in str(e): raise
            return t
        if k == 6:
            m = leaf(True)
            if rng.integers(0, 2):   # translucent.rs:70-74: reflect / transmit per hit; a checkerboard with black squares on both makes hits without any BSDF
                black_or = lambda: s.add_texture_checkerboard(s.add_texture_constant((0.0, 0.0, 0.0)), s.add_texture_constant(c()), su=float(rng.uniform(2, 7)), sv=float(rng.uniform(2, 7)), aa="none")
                which = int(rng.integers(0, 3))
                if which != 1: s.set_material_texture(m, "reflect", black_or() if rng.integers(0, 2) else tex())
                if which != 0: s.set_material_texture(m, "transmit", black_or() if rng.integers(0, 2) else tex())
        else:                                # MixMaterial over textured sub-materials (never bumped: mix.rs has no bump map)
            a = s.add_material_plastic(one, c(0.05, 0.5), float(rng.uniform(0.01, 0.4)), bool(rng.integers(0, 2))); s.set_material_texture(a, "Kd", tex())
            if rng.integers(0, 3) == 0: s.set_material_float_texture(a, "roughness", ftex(0.5))
            b = leaf(False) if rng.integers(0, 2) else s.add_material_matte_tex(tex(), float(rng.choice([0.0, rng.uniform(1, 60)])))
            for sub in (a, b):   # bump-mapped children: the first one's map shapes the mixture's frame, the second one's changes nothing (mix.rs:63-76)
                if rng.integers(0, 3) == 0:
                    s.set_material_bump(sub, s.add_texture_scale(random_texture(s, host, rng, True), s.add_texture_constant(float(rng.uniform(0.005, 0.2)))))
            mx = s.add_material_mix(a, b, c())
            if rng.integers(0, 2):
                try: s.set_material_texture(mx, "amount", tex())
                except (pbrt_hip.PbrtHipError, RuntimeError) as e:
                    if "per-hit colours" not in str(e): raise            # more than six colour slots: the product refuses; the constant amount stays on both sides
                    return mx
            return mx
    elif k == 8 and rng.integers(0, 2):
        m = s.add_material_metal(c(0.1, 2.0), c(1.0, 4.0), float(rng.uniform(0.01, 0.3)), float(rng.uniform(0.01, 0.3)), bool(rng.integers(0, 2)))
        if rng.integers(0, 2): s.set_material_texture(m, "eta", s.add_texture_scale(tex(), s.add_texture_constant(2.0)))
        s.set_material_texture(m, "k", s.add_texture_scale(tex(), s.add_texture_constant(4.0)))
        if rng.integers(0, 3) == 0: s.set_material_float_texture(m, "roughness", ftex(0.5))
    else:
        m = F.random_material(s, rng)       # a constant material, possibly only bumped
        if rng.integers(0, 2) == 0: return m
    mat_none = False
    try:
        if rng.integers(0, 2): s.set_material_bump(m, s.add_texture_scale(random_texture(s, host, rng, True), s.add_texture_constant(float(rng.uniform(0.005, 0.2)))))
    except pbrt_hip.PbrtHipError as e:
        if "live values" in str(e): raise    # a tree the product's evaluator refuses: the whole case is skipped (the oracle would accept it and the two scenes would diverge)
        mat_none = True                      # "none" has no BSDF to bump: refused identically by both libraries
    return m


def build_case(host, seed):
    rng = np.random.default_rng(seed + 50000)
    res = (int(rng.integers(17, 41)), int(rng.integers(13, 37)))
    spp = int(rng.choice([1, 2, 4, 5]))
    sky = rng.integers(0, 2)
    st = rng.integers(0, 2 ** 31)
    lens = float(rng.choice([0.0, 0.05]))

    def cap(s):
        g = np.random.default_rng(st)
        if sky:
            t = F.random_transform(host, g)
            img = random_image(g, hdr=True)
            img[int(g.integers(0, img.shape[0])), int(g.integers(0, img.shape[1]))] = (50.0, 40.0, 30.0)
            s.add_light_infinite_map(tuple(g.uniform(0.2, 1.0, 3)), img, t[0], t[1])
        else:
            s.add_light_infinite(tuple(g.uniform(0.3, 1.0, 3)))
        if g.integers(0, 2): s.add_light_point(tuple(g.uniform(2, 12, 3)), g.uniform(-1.5, 1.5, 3).astype(np.float32))
        mats = [textured_material(s, host, g) for _ in range(4)]
        for k in range(int(g.integers(2, 5))):
            P, idx = host.gen_random_tris(int(g.integers(5, 120)), int(g.integers(1, 1000)))
            N = g.normal(size=P.shape).astype(np.float32) if g.integers(0, 2) else None
            S = g.normal(size=P.shape).astype(np.float32) if g.integers(0, 3) == 0 else None
            UV = g.uniform(-1, 2, (len(P), 2)).astype(np.float32) if g.integers(0, 3) else None
            s.add_mesh(P, idx, mats[k % 4], N=N, S=S, UV=UV, reverse_orientation=bool(g.integers(0, 2)), swaps_handedness=bool(g.integers(0, 2)))
            if g.integers(0, 3) == 0:
                s.set_last_mesh_alpha_textures(random_mask(s, host, g) if g.integers(0, 2) else None, random_mask(s, host, g) if g.integers(0, 2) else None)
        Pg, ig = scenes.grid_mesh(3, z=-1.3, size=2.5)
        s.add_mesh(Pg, ig, mats[3], UV=(Pg[:, :2] * np.float32(0.7)).astype(np.float32))
        if g.integers(0, 2):
            ob = s.object_begin()
            P, idx = host.gen_random_tris(int(g.integers(2, 60)), int(g.integers(1, 1000)))
            s.add_mesh(P * np.float32(0.5), idx, mats[1], N=(g.normal(size=P.shape).astype(np.float32) if g.integers(0, 2) else None), UV=g.uniform(0, 1, (len(P), 2)).astype(np.float32))
            if g.integers(0, 2): s.set_last_mesh_alpha_textures(random_mask(s, host, g), None)
            s.object_end()
            for _ in range(int(g.integers(1, 4))):
                t = F.random_transform(host, g)
                s.add_instance(ob, t[0], t[1])
        w2c, c2w = host.look_at(g.uniform(-0.5, 0.5, 3) + np.array([0, -4.5, 0.5]), [0, 0, 0], [0, 0, 1])
        s.set_camera_perspective(host.perspective_raster_to_camera(float(g.uniform(30, 60)), res[0], res[1]), c2w, lens_radius=lens, focal_distance=4.5)
        cb, table, sb = host.film_box(res[0], res[1])
        s.set_film(res[0], res[1], cb, (0.5, 0.5), table)
        s.set_sampler(int(g.integers(0, 1)), spp, sb)
        s.build_accel_best(0, int(g.choice([1, 4])))
        return cb
    return cap, dict(max_depth=int(rng.integers(1, 7)), light_strategy=int(rng.integers(0, 3)))


def run_case(host, seed):
    cap, kw = build_case(host, seed)
    prod = pbrt_hip.Scene(); orc = OracleScene()
    try:
        cap(prod)
    except pbrt_hip.PbrtHipError as e:
        if "live values" in str(e): return None, "texture tree deeper than the evaluator's value stack"   # refused, not mis-rendered
        raise
    cap(orc)
    set_libm_mode(1)
    try:
        oxyz, owt, ost, _ = orc.render_path_ex(**kw)
    finally:
        set_libm_mode(0)
    gxyz, gwt, gst = prod.render_path(**kw)
    counters = lambda st: (st.regular_rays, st.shadow_rays, st.paths_total, st.paths_zero_radiance, st.light_distributions_created)
    nb = int((gxyz.view(np.uint32) != oxyz.view(np.uint32)).any(axis=2).sum())
    return counters(gst) == counters(ost) and np.array_equal(gwt.view(np.uint32), owt.view(np.uint32)) and nb == 0, (kw, nb, counters(gst), counters(ost))


@pytest.mark.parametrize("seed", list(range(1, 21)))
def test_random_textured_scene_film_bit_exact(host, seed):
    ok, info = run_case(host, seed)
    if ok is None: pytest.skip(info)
    assert ok, (seed, info)
