"""The oracle against OUTPUT OF THE REFERENCE ITSELF (first with the reference's own integrator for those renders, pixel for pixel; then with the path integrator the product has): the renders hackmad/pbrt-v3-rs commits next to its scenes (renders/**.png), for the ten scenes of its `scenes/`
directory that need nothing outside the product's scope (tests/reference_scenes.py restates them through the C ABI; tests/golden/ref_renders/*.npz hold the reference's
pixels, decoded by tests/golden/make_reference_renders.py).  Cameras (perspective, orthographic, environment), lights (point, spot, distant, infinite, goniometric),
checkerboard and dots textures, alpha masks, object instancing, the triangle intersector, the BVH, the Halton sampler, the box-filtered film and the 8-bit output curve all
have to agree for these to pass.  Two divergences of the first version were found this way: the reference builds DotsTexture with its two operands swapped (quirk B13) and
reads the spot light's rim from "conedeltaangle" (the scene file's "conedelta" is ignored)."""
import numpy as np
import pytest

import pbrt_hip
import reference_scenes as R
from oracle_binding import oracle_binding

# (builder, samples per pixel here; the reference used 128)
DETERMINISTIC = [("triangles_alpha_mask", 128), ("lights_point", 128), ("lights_spot", 128), ("lights_goniometric", 128), ("lights_distant", 128)]
NOISY = [("lights_infinite_no_map", 64), ("cameras_perspective", 64), ("cameras_orthographic", 64), ("cameras_environment", 64), ("objects_instances", 64)]


def render(binding, name, spp, **kw):
    host = pbrt_hip.Host()
    with pbrt_hip.Scene(binding) as s:
        info = getattr(R, name)(s, host, spp=spp, **kw)
        xyz, wt, st = s.render_path(max_depth=info["max_depth"])
        return s.film_to_rgb(xyz, wt), info, st


def check_against_reference(rgb, info, noisy):
    c = R.compare(rgb, R.reference_render(info["render"]))
    if not noisy:
        # one delta light on matte surfaces: Whitted's sum and the path integrator's direct term are the same single product; what is left is 8-bit rounding and the
        # pixel-edge estimate (the reference averaged 128 samples per pixel)
        assert c["mean"] < 0.2 and c["bad"] < 0.003 and c["block_mean"] < 0.06 and c["block_bad"] == 0.0, c
        # ... and at the reference's own 128 spp the PATH integrator at maxdepth 1 draws the same camera samples and evaluates the same single light term as Whitted (only the
        # order of two multiplications differs): the render equals the reference's PNG pixel for pixel (measured: 99.999 - 100 % identical, never more than one level apart)
        d = np.abs(R.to_8bit(rgb).astype(np.int32) - R.reference_render(info["render"]).astype(np.int32)).max(-1)
        assert (d == 0).mean() >= 0.9995 and d.max() <= 2, ((d == 0).mean(), d.max())
    else:
        # an infinite light (one light sample in Whitted, light + BSDF sample with MIS here) and / or two lights (both in Whitted, one of the two picked here): same expectation
        assert c["mean"] < 4.0 and c["bad"] < 0.06 and c["block_mean"] < 0.35 and c["block_bad"] < 0.002, c
    assert max(abs(b) for b in c["bias"]) < 0.2, c  # no colour or exposure drift: a twentieth of one 8-bit level
    return c


@pytest.mark.parametrize("name,spp", DETERMINISTIC)
def test_oracle_equals_the_references_render_delta_light_scenes(name, spp):
    rgb, info, _ = render(oracle_binding(), name, spp)
    check_against_reference(rgb, info, noisy=False)


@pytest.mark.parametrize("name,spp", NOISY)
def test_oracle_equals_the_references_render_sky_and_sun_scenes(name, spp):
    rgb, info, _ = render(oracle_binding(), name, spp)
    check_against_reference(rgb, info, noisy=True)


def test_the_comparison_has_teeth():
    """the two divergences this suite found, put back in: each fails the thresholds by a wide margin"""
    host = pbrt_hip.Host()
    ref = R.reference_render("lights_spot")
    with pbrt_hip.Scene(oracle_binding()) as s:  # the spot light with the rim the scene file seems to ask for ("conedelta" 20)
        l2w, w2l, ct, cs = host.spot(R._ident(), (-5.0, 0.0, 5.0), (0.0, 0.0, 0.0), 25.0, 20.0)
        s.add_light_spot((80.0, 90.0, 100.0), l2w, w2l, ct, cs)
        R._cube_and_checker_floor(s, host)
        R.camera_film(s, host, (0, 5, 3), (0, 0, 0), (0, 0, 1), 90.0, 400, 400, 8)
        s.build_accel(0, 4)
        xyz, wt, _ = s.render_path(max_depth=1)
        c = R.compare(s.film_to_rgb(xyz, wt), ref)
    assert c["mean"] > 5.0 and c["block_bad"] > 0.1
    # a one-level exposure error (x 1.01) on a matching render is caught by the bias bound
    rgb, info, _ = render(oracle_binding(), "lights_point", 8)
    assert max(abs(b) for b in R.compare(rgb * 1.02, R.reference_render(info["render"]))["bias"]) > 0.2


def test_oracle_equals_the_references_render_of_the_bump_mapped_sphere():
    """scenes/materials/bump.pbrt: a matte Sphere (ORACLE ONLY, see test_oracle_sphere.py) displaced by the `windy` float texture, sky + sun.  Pins Material::bump, the windy /
    fBm noise stack and the sphere's dpdu / dpdv / dndu / dndv against the reference's pixels: every crease of the relief has to fall where the reference put it."""
    rgb, info, _ = render(oracle_binding(), "materials_bump", 96)
    c = R.compare(rgb, R.reference_render(info["render"]))
    assert c["mean"] < 5.0 and c["block_mean"] < 0.4 and c["block_bad"] < 0.002 and max(abs(b) for b in c["bias"]) < 0.25, c
    # the relief itself, noise averaged out: gradients of the 4 x 4 block means on the sphere correlate with the reference's
    ref = R.reference_render(info["render"]).astype(np.float64).mean(-1); mine = R.to_8bit(rgb).astype(np.float64).mean(-1)
    bm = lambda a: a.reshape(100, 4, 100, 4).mean((1, 3))[25:75, 30:70]
    gx = lambda a: np.diff(bm(a), axis=1).ravel()
    assert np.corrcoef(gx(ref), gx(mine))[0, 1] > 0.9


def test_sampler_scenes_background_masks_equal_the_references_renders():
    """scenes/samplers/{halton,sobol}.pbrt (64 x 64, 16 spp): a matte Sphere (oracle only) seen through a wide thin lens, so every pixel of the blur ring depends on where
    each of its 16 film + lens samples falls.  Whitted and the path integrator draw different dimensions for the lights, but the camera sample (dimensions 0-4) is the same:
    the set of pixels in which ALL 16 rays miss the sphere (pure sky, 8-bit 231) is a function of the sampler's first five dimensions, the pixel-to-sample mapping, the thin lens and
    the sphere alone.  The oracle's set equals the reference's in 99.6 - 99.8 % of the pixels with its own sampler and in only 97 % with the other one (so the comparison
    tells Halton from Sobol); scaling the lens by 2 % already triples the disagreement.  The residue (7 pixels with Halton, 16 with Sobol) is one-sided — the reference's PNG shows
    level 231 where the oracle counts ONE sphere hit among the 16 rays — and is what the 8-bit criterion predicts: Whitted estimates the sky term with one light sample whose value
    0.8 * (0.2 / pi) * cos / pdf ranges up to 1.0, so a lone hit whose estimate lands within 0.056 of 0.8 leaves the pixel at 231.  That is a few per cent of the one-hit pixels
    (206 with Halton, 251 with Sobol: 3.4 % and 6.4 % observed), and it can only add pixels to the reference's set, never remove one — which is what the test asserts."""
    bg8 = R.to_8bit(np.array([0.8, 0.8, 0.8], np.float32))
    host = pbrt_hip.Host()
    masks = {}
    for smp in ("halton", "sobol"):
        with pbrt_hip.Scene(oracle_binding()) as s:
            info = R.samplers_scene(s, host, smp)
            xyz, wt, _ = s.render_path(max_depth=1)
            masks[smp] = np.all(np.abs(s.film_to_rgb(xyz, wt) - 0.8) < 1e-5, -1)
    ref = {smp: np.all(R.reference_render("samplers_" + smp) == bg8, -1) for smp in ("halton", "sobol")}
    for smp, other in (("halton", "sobol"), ("sobol", "halton")):
        assert 0.15 < ref[smp].mean() < 0.25
        same, cross = int((masks[smp] != ref[smp]).sum()), int((masks[smp] != ref[other]).sum())
        assert same <= 20, (smp, same)                 # <= 0.5 % of 4096 pixels
        assert int((masks[smp] & ~ref[smp]).sum()) == 0  # never a hit in the reference that the oracle lacks
        assert cross >= 80, (smp, cross)               # the other sampler's pattern is a different one


# ---- the reference made every one of these renders with its Whitted integrator; the oracle has it too (ORACLE ONLY, oracle_render.hpp li_whitted), so the comparison
# needs no statistics: same samples, same light draws, same pixels.
WHITTED = [("triangles_alpha_mask", 128), ("lights_point", 128), ("lights_spot", 128), ("lights_goniometric", 128), ("lights_distant", 128), ("lights_infinite_no_map", 128),
           ("cameras_perspective", 128), ("cameras_orthographic", 128), ("cameras_environment", 128), ("objects_instances", 128), ("materials_bump", 128),
           ("samplers_halton", 16), ("samplers_sobol", 16), ("samplers_random", 16), ("lights_diffuse", 128)]   # lights_diffuse: a spherical DiffuseAreaLight (oracle only), soft shadows


@pytest.mark.parametrize("name,spp", WHITTED)
def test_oracle_whitted_reproduces_the_references_render_pixel_for_pixel(name, spp):
    """At the reference's own sample count the oracle's Whitted render equals the reference's PNG in (essentially) every pixel — including the Monte Carlo noise of the sky-lit
    scenes, the 16-sample thin-lens renders of the Halton and the Sobol sampler and the bump-mapped sphere.  Measured: 100 % identical pixels on nine scenes, >= 99.92 % on
    the others, where a handful of values sit on an 8-bit rounding boundary and a last-bit libm difference decides.  What this pins against the reference's real output:
    perspective / orthographic / environment cameras and the thin lens, Halton (incl. its permutation tables, through dimension 8) and Sobol, point / spot / distant /
    goniometric / infinite lights with their sample_li and the 2-D distribution, checkerboard / dots / windy textures, bump mapping, alpha masks, object instancing, triangle and
    sphere intersection, the BVH, spawn-ray offsets, the box-filtered film and the output curve."""
    import ctypes as C
    host = pbrt_hip.Host()
    with pbrt_hip.Scene(oracle_binding()) as s:
        info = R.samplers_scene(s, host, name.split("_")[1], spp=spp) if name.startswith("samplers_") else getattr(R, name)(s, host, spp=spp)
        # samplers_random: the RandomSampler (oracle only) — one PCG32 stream per tile, seeded with the tile's index, drawn from in the order render_tile visits pixels and samples:
        # equal pixels mean the same tile enumeration, the same pixel loop and the same number of draws per camera sample as the reference
        s.b.lib.oracle_set_integrator.argtypes = [C.c_void_p, C.c_int]
        assert s.b.lib.oracle_set_integrator(s.h, 1) == 0
        xyz, wt, _ = s.render_path(max_depth=5)   # Integrator "whitted" default maxdepth
        rgb = s.film_to_rgb(xyz, wt)
    d = np.abs(R.to_8bit(rgb).astype(np.int32) - R.reference_render(info["render"]).astype(np.int32)).max(-1)
    assert (d == 0).mean() >= 0.999, ((d == 0).mean(), d.max())
    assert (d <= 1).mean() >= 0.9999 and d.max() <= 6, ((d <= 1).mean(), d.max())


def test_oracle_whitted_reproduces_the_glass_spheres_of_depth_of_field():
    """scenes/cameras/depth-of-field.pbrt: five glass spheres (ORACLE ONLY) on the checkered floor through a lens of radius 0.25, 128 spp, Whitted depth 5 — the specular
    recursion with its ray differentials (sampler_integrator.rs:79-238), smooth glass as SpecularReflection + SpecularTransmission (allow_multiple_lobes = false), the dielectric
    Fresnel term and Snell refraction.  A 400 x 199 crop around the focused green sphere equals the reference's PNG in 99.98 % of its pixels.  It does so with index of refraction
    1.5: the file's `"float eta" 2` is a parameter the reference never reads (quirk B14, glass.rs:158) — with 2.0 a third of the crop differs by up to 125 levels."""
    import ctypes as C
    host = pbrt_hip.Host()
    with pbrt_hip.Scene(oracle_binding()) as s:
        info = R.cameras_depth_of_field(s, host, spp=128, crop=(0.25, 0.75, 0.3, 0.8))
        s.b.lib.oracle_set_integrator.argtypes = [C.c_void_p, C.c_int]
        assert s.b.lib.oracle_set_integrator(s.h, 1) == 0
        xyz, wt, _ = s.render_path(max_depth=5)
        rgb = s.film_to_rgb(xyz, wt)
    cb = info["crop"]
    ref = R.reference_render(info["render"])[cb[1]:cb[3], cb[0]:cb[2]]
    d = np.abs(R.to_8bit(rgb).astype(np.int32) - ref.astype(np.int32)).max(-1)
    assert (d == 0).mean() >= 0.999 and (d <= 1).mean() >= 0.9999 and d.max() <= 4, ((d == 0).mean(), (d <= 1).mean(), d.max())


@pytest.mark.parametrize("which", ["fbm", "wrinkled", "windy", "marble", "dots", "bilerp", "uv", "mix", "scale", "constant", "2d-checkerboard"])
def test_oracle_whitted_reproduces_the_six_shape_texture_scenes_pixel_for_pixel(which):
    """scenes/textures/<which>.pbrt: a sphere, a hyperboloid, a cone, a paraboloid, a cylinder and a disk (all six ORACLE ONLY: sphere.rs, hyperboloid.rs, cone.rs, paraboloid.rs,
    cylinder.rs, disk.rs with EFloat) wearing the file's texture in front of a wall under a white sky, 128 spp.  The oracle's render equals the reference's PNG in 99.996 - 99.999 %
    of the pixels, the rest one 8-bit level apart: fbm, wrinkled (turbulence), windy, marble inside a mix MATERIAL, dots (operands swapped, quirk B13), bilerp, uv, the mix and scale
    TEXTURES over windy x checkerboard, constant, and the 2-D checkerboard with its closed-form filter and negative scales — each on six different (u, v) parameterisations.
    (The four 3-D textures keep only the sphere's corner of the reference image as fixture; the (u, v)-dependent ones are compared whole.)"""
    import ctypes as C
    host = pbrt_hip.Host()
    with pbrt_hip.Scene(oracle_binding()) as s:
        info = R.textures_six_shapes(s, host, which, spp=128)
        s.b.lib.oracle_set_integrator.argtypes = [C.c_void_p, C.c_int]
        assert s.b.lib.oracle_set_integrator(s.h, 1) == 0
        xyz, wt, _ = s.render_path(max_depth=5)
        rgb = s.film_to_rgb(xyz, wt)
    ref = R.reference_render(info["render"])
    mine = R.to_8bit(rgb)
    if ref.shape[0] != 400:
        r0, r1, c0, c1 = R.TEX_CROP
        mine = mine[r0:r1, c0:c1]
    d = np.abs(mine.astype(np.int32) - ref.astype(np.int32)).max(-1)
    assert (d == 0).mean() >= 0.9995 and d.max() <= 2, ((d == 0).mean(), d.max())


def test_oracle_whitted_reproduces_the_image_maps_under_the_four_2d_mappings():
    """scenes/textures/2d-mappings.pbrt: four Hyperboloids (ORACLE ONLY, hyperboloid.rs) wearing scenes/images/checkerboard.png through `imagemap` textures with the uv, spherical,
    cylindrical and planar mappings.  Everything the image-texture path does is in these pixels: read_8_bit's u8 / 255, the inverse gamma of convert_in, the MIPMap pyramid, EWA filtering
    driven by the camera ray's differentials, `repeat` wrapping, the four TextureMapping2D classes with their derivative estimates.  The oracle's render equals the reference's PNG in
    99.99 % of the 320 000 pixels, the rest one 8-bit level apart."""
    import ctypes as C
    host = pbrt_hip.Host()
    with pbrt_hip.Scene(oracle_binding()) as s:
        info = R.textures_2d_mappings(s, host, spp=128)
        s.b.lib.oracle_set_integrator.argtypes = [C.c_void_p, C.c_int]
        assert s.b.lib.oracle_set_integrator(s.h, 1) == 0
        xyz, wt, _ = s.render_path(max_depth=5)
        rgb = s.film_to_rgb(xyz, wt)
    d = np.abs(R.to_8bit(rgb).astype(np.int32) - R.reference_render(info["render"]).astype(np.int32)).max(-1)
    assert (d == 0).mean() >= 0.9995 and d.max() <= 2, ((d == 0).mean(), d.max())
    for k in range(4):   # each mapping's own quarter of the image
        assert (d[:, 200 * k:200 * (k + 1)] == 0).mean() >= 0.999
