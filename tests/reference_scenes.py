"""Scenes of the reference's own `scenes/` directory that lie inside the product's feature set (triangles only, no external assets), restated through the C ABI
(`pbrt_hip.Scene`, product or oracle), and the reference's own renders of them (`renders/**.png`, decoded into tests/golden/ref_renders/*.npz by
tests/golden/make_reference_renders.py).  The reference rendered them with `Integrator "whitted"`: direct lighting from every light plus specular recursion.  For matte
surfaces under delta lights that is the path integrator at maxdepth 1 (one light: the same single term; k lights: uniform_sample_one_light picks each with 1 / k and
divides by it), so the comparison is quantitative: the render through the C ABI, pushed through the reference's 8-bit output curve (image_io.rs `apply_gamma`), against
the reference's PNG, within what 8-bit rounding and pixel-edge sampling allow."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
IDENT = None


def _ident():
    import pbrt_hip
    return (pbrt_hip.IDENTITY.copy(), pbrt_hip.IDENTITY.copy())


def ctm(host, *steps):
    t = _ident()
    for s in steps:
        t = host.compose(t, s)
    return t


def to_8bit(rgb):
    """image_io.rs write_8_bit / apply_gamma: clamp(255 * gamma_correct(v) + 0.5, 0, 255) as u8"""
    v = np.asarray(rgb, np.float32)
    g = np.where(v <= 0.0031308, 12.92 * v, 1.055 * np.power(np.maximum(v, 0.0), 1.0 / 2.4) - 0.055)
    return np.clip(255.0 * g + 0.5, 0.0, 255.0).astype(np.uint8)


def reference_render(name):
    z = np.load(os.path.join(HERE, "golden", "ref_renders", name + ".npz"))
    return z["rgb"]  # (H, W, 3) uint8


def camera_film(s, host, eye, look, up, fov, xres, yres, spp, screen=None):
    w2c, c2w = host.look_at(eye, look, up)
    s.set_camera_perspective(host.perspective_raster_to_camera(fov, xres, yres, screen), c2w)
    cb, table, sb = host.film_box(xres, yres)
    s.set_film(xres, yres, cb, (0.5, 0.5), table)
    s.set_sampler(0, spp, sb)


CUBE_P = np.array([[-1, -1, -1], [-1, 1, -1], [1, 1, -1], [1, -1, -1], [-1, -1, 1], [-1, 1, 1], [1, 1, 1], [1, -1, 1]], np.float32)
CUBE_ST = np.array([[0, 0], [0, 1], [1, 1], [1, 0], [1, 0], [1, 1], [0, 1], [0, 0]], np.float32)
CUBE_IDX = np.array([0, 1, 2, 3, 0, 2, 1, 5, 6, 2, 1, 6, 4, 5, 1, 0, 4, 1, 3, 2, 6, 7, 3, 6, 6, 5, 4, 6, 4, 7, 4, 0, 3, 7, 4, 3], np.uint32)
QUAD_IDX = np.array([0, 1, 2, 0, 2, 3], np.uint32)
QUAD_ST = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)


def quad(size):
    return np.array([[-size, -size, 0], [size, -size, 0], [size, size, 0], [-size, size, 0]], np.float32)


def triangles_alpha_mask(s, host, spp=32, res=400):
    """scenes/shapes/triangles-alpha-mask.pbrt -> renders/shapes/triangles-alpha-mask.png (400 x 400, the reference used 128 spp)"""
    s.add_light_point((0.4 * 200, 0.45 * 200, 0.5 * 200), (-5.0, 0.0, 5.0))
    # "float inside" 1 "float outside" 0 — which the reference's constructor hands over swapped (dots.rs:61-66, quirk B13): alpha is 0 INSIDE the dots
    alpha = s.add_texture_dots(s.add_texture_constant(0.0), s.add_texture_constant(1.0), 10.0, 10.0)
    t = ctm(host, host.rotate(135.0, (0, 0, 1)))
    m1 = s.add_material_matte((0.2, 0.01, 0.01))
    s.add_mesh(host.transform_points(t[0], CUBE_P), CUBE_IDX, m1, UV=CUBE_ST)
    s.set_last_mesh_alpha_textures(alpha=alpha)
    t = ctm(host, host.translate((0, 0, -1)))
    checks = s.add_texture_checkerboard(s.add_texture_constant((0.3, 0.3, 0.3)), s.add_texture_constant((0.8, 0.8, 0.8)), 24.0, 24.0)
    m2 = s.add_material_matte_tex(checks)
    s.add_mesh(host.transform_points(t[0], quad(20.0)), QUAD_IDX, m2, UV=QUAD_ST)
    camera_film(s, host, (0, 5, 3), (0, 0, 0), (0, 0, 1), 90.0, res, res, spp)
    s.build_accel(0, 4)
    return dict(max_depth=1, render="shapes_triangles-alpha-mask")


def _cube_and_checker_floor(s, host, cube_rotate_deg=45.0, tex1=0.3):
    """the stage most of scenes/lights/*.pbrt share: geometry/cube.pbrt rotated about z, matte (.2 .01 .01), above a 40 x 40 quad at z = -1 with a 24 x 24 checkerboard"""
    m1 = s.add_material_matte((0.2, 0.01, 0.01))
    t = ctm(host, host.rotate(cube_rotate_deg, (0, 0, 1)))
    s.add_mesh(host.transform_points(t[0], CUBE_P), CUBE_IDX, m1, UV=CUBE_ST)
    t = ctm(host, host.translate((0, 0, -1)))
    checks = s.add_texture_checkerboard(s.add_texture_constant((tex1, tex1, tex1)), s.add_texture_constant((0.8, 0.8, 0.8)), 24.0, 24.0)
    s.add_mesh(host.transform_points(t[0], quad(20.0)), QUAD_IDX, s.add_material_matte_tex(checks), UV=QUAD_ST)


def lights_point(s, host, spp=32, res=400):
    """scenes/lights/point.pbrt -> renders/lights/point.png"""
    s.add_light_point((80.0, 90.0, 100.0), (-5.0, 0.0, 5.0))  # "rgb I" [.4 .45 .5] x "rgb scale" [200 200 200]
    _cube_and_checker_floor(s, host)
    camera_film(s, host, (0, 5, 3), (0, 0, 0), (0, 0, 1), 90.0, res, res, spp)
    s.build_accel(0, 4)
    return dict(max_depth=1, render="lights_point")


def lights_spot(s, host, spp=32, res=400):
    """scenes/lights/spot.pbrt -> renders/lights/spot.png: "float coneangle" 25; the file's "float conedelta" 20 is not a parameter the reference reads
    (spot.rs:155-156 looks up "conedeltaangle"), so the default 5 applies — its render shows the 5-degree rim"""
    l2w, w2l, cos_total, cos_start = host.spot(_ident(), (-5.0, 0.0, 5.0), (0.0, 0.0, 0.0), 25.0, 5.0)
    s.add_light_spot((80.0, 90.0, 100.0), l2w, w2l, cos_total, cos_start)
    _cube_and_checker_floor(s, host)
    camera_film(s, host, (0, 5, 3), (0, 0, 0), (0, 0, 1), 90.0, res, res, spp)
    s.build_accel(0, 4)
    return dict(max_depth=1, render="lights_spot")


def lights_infinite_no_map(s, host, spp=64, res=400):
    """scenes/lights/infinite-no-map.pbrt -> renders/lights/infinite-no-map.png.  Whitted takes one light sample per camera sample, the path integrator's direct term adds
    the BSDF-sampled MIS partner: same expectation, both noisy (the reference used 128 spp)."""
    s.add_light_infinite((0.4, 0.45, 0.5))
    _cube_and_checker_floor(s, host)
    camera_film(s, host, (0, 5, 3), (0, 0, 0), (0, 0, 1), 90.0, res, res, spp)
    s.build_accel(0, 4)
    return dict(max_depth=1, render="lights_infinite-no-map", noisy=True)


def lights_goniometric(s, host, spp=32, res=400):
    """scenes/lights/goniometric.pbrt -> renders/lights/goniometric.png; the map is scenes/images/goniometric-upward-downward.png (fixture: its pixels / 255 as read_8_bit returns them)"""
    img = np.load(os.path.join(HERE, "golden", "ref_renders", "image_goniometric-upward-downward.npz"))["rgb"].astype(np.float32) / np.float32(255.0)
    t = ctm(host, host.translate((-5, 0, 5)), host.rotate(135.0, (1, 0, 0)), host.rotate(60.0, (0, 1, 0)))
    s.add_light_goniometric((80.0, 90.0, 100.0), t[0], t[1], img)
    _cube_and_checker_floor(s, host)
    camera_film(s, host, (0, 5, 3), (0, 0, 0), (0, 0, 1), 90.0, res, res, spp)
    s.build_accel(0, 4)
    return dict(max_depth=1, render="lights_goniometric")


def blackbody(key):
    """RGB of a `"blackbody L" [T scale]` parameter (tests/golden/blackbody_rgb.json, made by make_blackbody_fixture.py from the reference's CIE tables)"""
    import json
    return tuple(json.load(open(os.path.join(HERE, "golden", "blackbody_rgb.json")))[key])


def lights_distant(s, host, spp=32, res=400):
    """scenes/lights/distant.pbrt -> renders/lights/distant.png: "point from" [-5 0 5] "point to" [0 0 0] "blackbody L" [4500 1.5]"""
    s.add_light_distant(blackbody("4500x1.5"), host.distant_direction(_ident()[0], (-5.0, 0.0, 5.0), (0.0, 0.0, 0.0)))
    _cube_and_checker_floor(s, host)
    camera_film(s, host, (0, 5, 3), (0, 0, 0), (0, 0, 1), 90.0, res, res, spp)
    s.build_accel(0, 4)
    return dict(max_depth=1, render="lights_distant")


def _sky_and_sun(s, host, t=None):
    """LightSource "infinite" "rgb L" [.4 .45 .5] + LightSource "distant" "point from" [-30 40 100] "blackbody L" [3000 1.5] under the CTM `t`; no "point to" in the
    files: the default is (0, 0, 1) (distant.rs:180-181), so the light points along from - to = (-30, 40, 99)"""
    t = t if t is not None else _ident()
    s.add_light_infinite((0.4, 0.45, 0.5), t[0], t[1])
    s.add_light_distant(blackbody("3000x1.5"), host.distant_direction(t[0], (-30.0, 40.0, 100.0), (0.0, 0.0, 1.0)))


def cameras_perspective(s, host, spp=64, res=400):
    """scenes/cameras/perspective.pbrt -> renders/cameras/perspective.png (two lights: Whitted adds both per camera sample, the path integrator picks one of the two)"""
    _sky_and_sun(s, host)
    _cube_and_checker_floor(s, host, tex1=0.1)
    camera_film(s, host, (0, 2, 2), (0, 0, 0), (0, 0, 1), 90.0, res, res, spp)
    s.build_accel(0, 4)
    return dict(max_depth=1, render="cameras_perspective", noisy=True)


def cameras_orthographic(s, host, spp=64, res=400):
    """scenes/cameras/orthographic.pbrt -> renders/cameras/orthographic.png: `Scale 0.25 0.25 0.25` follows the two LightSource lines, so it sits under the shapes only"""
    sc = host.scale((0.25, 0.25, 0.25))
    _sky_and_sun(s, host)
    m1 = s.add_material_matte((0.2, 0.01, 0.01))
    t = ctm(host, sc, host.rotate(45.0, (0, 0, 1)))
    s.add_mesh(host.transform_points(t[0], CUBE_P), CUBE_IDX, m1, UV=CUBE_ST)
    t = ctm(host, sc, host.translate((0, 0, -1)))
    checks = s.add_texture_checkerboard(s.add_texture_constant((0.1, 0.1, 0.1)), s.add_texture_constant((0.8, 0.8, 0.8)), 24.0, 24.0)
    s.add_mesh(host.transform_points(t[0], quad(20.0)), QUAD_IDX, s.add_material_matte_tex(checks), UV=QUAD_ST)
    w2c, c2w = host.look_at((0, 10, 10), (0, 0, 0), (0, 0, 1))
    s.set_camera_orthographic(host.orthographic_raster_to_camera(res, res), c2w)
    cb, table, sb = host.film_box(res, res)
    s.set_film(res, res, cb, (0.5, 0.5), table)
    s.set_sampler(0, spp, sb)
    s.build_accel(0, 4)
    return dict(max_depth=1, render="cameras_orthographic", noisy=True)


def _ring_transforms(host):
    return [ctm(host, host.rotate(36.0 * k, (0, 0, 1)), host.translate((0, 5, 0)), host.rotate(45.0, (0, 0, 1))) if k else
            ctm(host, host.translate((0, 5, 0)), host.rotate(45.0, (0, 0, 1))) for k in range(10)]


def _checker_floor(s, host, tex1):
    t = ctm(host, host.translate((0, 0, -1)))
    checks = s.add_texture_checkerboard(s.add_texture_constant((tex1, tex1, tex1)), s.add_texture_constant((0.8, 0.8, 0.8)), 24.0, 24.0)
    s.add_mesh(host.transform_points(t[0], quad(20.0)), QUAD_IDX, s.add_material_matte_tex(checks), UV=QUAD_ST)


def cameras_environment(s, host, spp=64, res=400):
    """scenes/cameras/environment.pbrt -> renders/cameras/environment.png (800 x 400): ten cubes on a ring around the camera"""
    _sky_and_sun(s, host)
    m = s.add_material_matte((0.8, 0.1, 0.01))
    for t in _ring_transforms(host):
        s.add_mesh(host.transform_points(t[0], CUBE_P), CUBE_IDX, m, UV=CUBE_ST)
    _checker_floor(s, host, 0.1)
    w2c, c2w = host.look_at((0, 0, 1), (0, 1, 0), (0, 0, 1))
    s.set_camera_environment(c2w, 2 * res, res)
    cb, table, sb = host.film_box(2 * res, res)
    s.set_film(2 * res, res, cb, (0.5, 0.5), table)
    s.set_sampler(0, spp, sb)
    s.build_accel(0, 4)
    return dict(max_depth=1, render="cameras_environment", noisy=True)


def objects_instances(s, host, spp=64, res=400):
    """scenes/objects/instances.pbrt -> renders/objects/instances.png: ONE cube object, ten ObjectInstances; `Translate 0 -1 0` follows LookAt, so it is part of the camera transform"""
    _sky_and_sun(s, host)
    m = s.add_material_matte((0.8, 0.1, 0.01))
    ob = s.object_begin(); s.add_mesh(CUBE_P, CUBE_IDX, m, UV=CUBE_ST); s.object_end()
    for t in _ring_transforms(host):
        s.add_instance(ob, t[0], t[1])
    _checker_floor(s, host, 0.1)
    cam = host.compose(host.look_at((0, 7, 15), (0, 0, 0), (0, 0, 1)), host.translate((0, -1, 0)))  # world -> camera and its inverse
    s.set_camera_perspective(host.perspective_raster_to_camera(45.0, res, res), cam[1])
    cb, table, sb = host.film_box(res, res)
    s.set_film(res, res, cb, (0.5, 0.5), table)
    s.set_sampler(0, spp, sb)
    s.build_accel(0, 4)
    return dict(max_depth=1, render="objects_instances", noisy=True)


def materials_bump(s, host, spp=64, res=400):
    """scenes/materials/bump.pbrt -> renders/materials/bump.png: a matte SPHERE (oracle only) whose bump map is the `windy` texture, above a matte floor"""
    from test_oracle_sphere import add_sphere
    s.add_light_infinite((0.8, 0.9, 1.0))
    s.add_light_distant(blackbody("3000x1.5"), host.distant_direction(_ident()[0], (-30.0, 40.0, 100.0), (0.0, 0.0, 1.0)))
    bump = s.add_texture_windy()
    m = s.add_material_matte((0.5, 0.5, 0.5))
    s.set_material_bump(m, bump)
    add_sphere(s, ctm(host, host.scale((1.5, 1.5, 1.5)), host.rotate(135, (1, 0, 0)), host.rotate(-15, (0, 0, 1)), host.rotate(15, (0, 1, 0))), 1.0, material=m)
    t = ctm(host, host.translate((0, 0, -1.5)))
    s.add_mesh(host.transform_points(t[0], quad(20.0)), QUAD_IDX, s.add_material_matte((0.5, 0.5, 0.5)), UV=QUAD_ST)
    camera_film(s, host, (0, 5, 1.5), (0, 0, 0), (0, 0, 1), 45.0, res, res, spp)
    s.build_accel(0, 4)
    return dict(max_depth=1, render="materials_bump", noisy=True)


def samplers_scene(s, host, sampler, spp=16):
    """scenes/samplers/{halton,sobol}.pbrt -> renders/samplers/{halton,sobol}.png (64 x 64, 16 spp): a matte SPHERE (oracle only) seen through a wide thin lens
    (lensradius 0.1, focaldistance 1, the sphere at z = 4: heavily defocused), sky 0.8 + a distant light.  No LookAt: the camera sits at the origin looking down +z."""
    from test_oracle_sphere import add_sphere
    s.add_light_infinite((0.8, 0.8, 0.8))
    s.add_light_distant((1.0, 1.0, 1.0), host.distant_direction(_ident()[0], (-1.0, 1.0, -1.0), (0.0, 0.0, 1.0)))
    m = s.add_material_matte((0.2, 0.2, 0.2))
    add_sphere(s, ctm(host, host.translate((0, 0, 4))), 1.0, material=m)
    ident = _ident()
    s.set_camera_perspective(host.perspective_raster_to_camera(35.0, 64, 64), ident[1], lens_radius=0.1, focal_distance=1.0)
    cb, table, sb = host.film_box(64, 64)
    s.set_film(64, 64, cb, (0.5, 0.5), table)
    if sampler == "sobol":
        z = np.load(os.path.join(HERE, "golden", "sobol_subset.npz"))
        s.set_sobol_tables(z["m32"], z["vdc"], z["vdc_inv"])
        s.set_sampler(1, spp, sb)
    elif sampler == "random":
        s.set_sampler(2, spp, sb)   # oracle only
    else:
        s.set_sampler(0, spp, sb)
    s.build_accel(0, 4)
    return dict(max_depth=1, render="samplers_" + sampler)


def cameras_depth_of_field(s, host, spp=128, crop=(0.0, 1.0, 0.0, 1.0)):
    """scenes/cameras/depth-of-field.pbrt -> renders/cameras/depth-of-field.png (800 x 400): five GLASS SPHERES (oracle only; Kr .2, coloured Kt, eta 2) in a row on the checkered
    floor, lensradius 0.25, focaldistance 7.12.  `crop`: Film "cropwindow" (the samples of a pixel do not depend on it)."""
    from test_oracle_sphere import add_sphere
    _sky_and_sun(s, host)
    kts = [(0.9, 0.2, 0.2), (0.2, 0.9, 0.2), (0.2, 0.2, 0.9), (0.9, 0.9, 0.2), (0.2, 0.9, 0.9)]
    t = ctm(host, host.translate((3, 3, 0)))
    for k, kt in enumerate(kts):
        if k:
            t = host.compose(t, host.translate((-3, -3, 0)))  # the five Translate directives accumulate inside one attribute block
        # "float eta" 2 in the file — which the reference does not read: glass.rs:158 looks up a float TEXTURE named "eta", finds none, and falls back to "index", 1.5 (quirk B14)
        add_sphere(s, t, 1.0, material=s.add_material_glass((0.2, 0.2, 0.2), kt, 0.0, 0.0, 1.5, True))
    _checker_floor(s, host, 0.1)
    w2c, c2w = host.look_at((1, 8, 1), (0, 0, 0), (0, 0, 1))
    s.set_camera_perspective(host.perspective_raster_to_camera(50.0, 800, 400), c2w, lens_radius=0.25, focal_distance=7.12)
    cb, table, sb = host.film_box(800, 400, crop_window=crop)
    s.set_film(800, 400, cb, (0.5, 0.5), table)
    s.set_sampler(0, spp, sb)
    s.build_accel(0, 4)
    return dict(max_depth=5, render="cameras_depth-of-field", crop=[int(v) for v in cb])


TEX_CROP = (60, 195, 5, 145)  # rows, columns of the 400 x 400 texture renders kept in the fixtures: the sphere and the wall around it


def add_hyperboloid(scene, t, p1, p2, phimax=360.0, material=0):
    import ctypes as C
    L = scene.b.lib
    fp = C.POINTER(C.c_float)
    L.oracle_add_hyperboloid.argtypes = [C.c_void_p, fp, fp, fp, fp, C.c_float, C.c_uint32, C.c_uint32]
    f = lambda a: np.ascontiguousarray(a, np.float32).ctypes.data_as(fp)
    scene._chk(L.oracle_add_hyperboloid(scene.h, f(t[0]), f(t[1]), f(p1), f(p2), phimax, material, 0))


def textures_2d_mappings(s, host, spp=128, crop=(0.0, 1.0, 0.0, 1.0)):
    """scenes/textures/2d-mappings.pbrt -> renders/textures/2d-mappings.png (800 x 400): four HYPERBOLOIDS (oracle only) wearing the image map scenes/images/checkerboard.png under the
    uv, spherical, cylindrical and planar mappings (EWA filtering, gamma on, repeat), in front of a wall, under a white sky"""
    img = np.load(os.path.join(HERE, "golden", "ref_renders", "image_checkerboard.npz"))["rgb"].astype(np.float32) / np.float32(255.0)   # read_8_bit: u8 / 255
    s.add_light_infinite((1.0, 1.0, 1.0))
    mip = s.add_mipmap(img, gamma=True)   # imagemap defaults for a .png: gamma on, EWA, maxanisotropy 8, repeat, scale 1
    rot = host.rotate(90.0, (1, 0, 0))
    for k, (x, kind) in enumerate(((-3.0, "uv"), (-1.0, "spherical"), (1.0, "cylindrical"), (3.0, "planar"))):
        t = ctm(host, host.translate((x, 0, 0)))
        tex = s.add_texture_imagemap(mip)
        if kind in ("spherical", "cylindrical"):
            s.set_texture_mapping(tex, kind, host.compose(t, rot)[1])   # world_to_texture = inverse of the CTM the Texture directive saw
        elif kind == "planar":
            s.set_texture_mapping(tex, kind, [0.0, -0.3, -0.3, 0.3, 0.0, 0.3, 0.0, 0.0])
        add_hyperboloid(s, t, (0.6, 0.6, 1.0), (0.6, -0.6, -1.0), material=s.add_material_matte_tex(tex))
    t = ctm(host, host.translate((0, -0.6, 0)))
    s.add_mesh(host.transform_points(t[0], np.array([[-20, 0, -20], [20, 0, -20], [20, 0, 20], [-20, 0, 20]], np.float32)), QUAD_IDX, s.add_material_matte((0.5, 0.5, 0.5)), UV=QUAD_ST)
    w2c, c2w = host.look_at((0, 15, 0), (0, 0, 0), (0, 0, 1))
    s.set_camera_perspective(host.perspective_raster_to_camera(15.0, 800, 400), c2w)
    cb, table, sb = host.film_box(800, 400, crop_window=crop)
    s.set_film(800, 400, cb, (0.5, 0.5), table)
    s.set_sampler(0, spp, sb)
    s.build_accel(0, 4)
    return dict(max_depth=5, render="textures_2d-mappings", crop=[int(v) for v in cb])


def add_quadric(scene, kind, t, radius, a, b, phimax=360.0, material=0):
    """kind: "cylinder" (a, b = zmin, zmax) | "cone" (a = height) | "paraboloid" (zmin, zmax) | "disk" (a = height, b = innerradius) — oracle only"""
    import ctypes as C
    L = scene.b.lib
    fp = C.POINTER(C.c_float)
    L.oracle_add_quadric.argtypes = [C.c_void_p, C.c_int, fp, fp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32]
    f = lambda m: np.ascontiguousarray(m, np.float32).ctypes.data_as(fp)
    scene._chk(L.oracle_add_quadric(scene.h, {"cylinder": 0, "cone": 1, "paraboloid": 2, "disk": 3}[kind], f(t[0]), f(t[1]), radius, a, b, phimax, material, 0))


def textures_six_shapes(s, host, which, spp=128):
    """scenes/textures/<which>.pbrt -> renders/textures/<which>.png in full: sphere, hyperboloid, cone, paraboloid, cylinder, disk (all ORACLE ONLY) in the files' order, each with the
    texture the file gives it, in front of the wall, under a white sky"""
    from test_oracle_sphere import add_sphere
    c = s.add_texture_constant
    matte = s.add_material_matte_tex
    checks = lambda su, sv: s.add_texture_checkerboard(c(1.0), c(0.0), su, sv)
    if which in ("fbm", "wrinkled", "windy", "marble", "dots", "bilerp", "scale"):
        if which == "fbm": m = matte(s.add_texture_fbm())
        elif which == "wrinkled": m = matte(s.add_texture_fbm(wrinkled=True))
        elif which == "windy": m = matte(s.add_texture_windy())
        elif which == "marble": m = s.add_material_mix(matte(s.add_texture_marble(scale=2.0, variation=10.0)), s.add_material_matte((0.01, 0.04, 0.17)), (0.1, 0.1, 0.1))
        elif which == "dots": m = matte(s.add_texture_dots(c((0.211, 0.213, 0.270)), c((0.8, 0.8, 0.8)), 12.0, 12.0))   # quirk B13: operands handed over swapped
        elif which == "bilerp": m = matte(s.add_texture_bilerp((1, 1, 1), (1, 0, 0), (0, 1, 0), (0, 0, 1)))
        else: m = matte(s.add_texture_scale(s.add_texture_windy(), checks(-16.0, -16.0)))
        mats = [m] * 6
    elif which == "uv":
        mats = [matte(s.add_texture_uv(su, sv)) for su, sv in ((-1, -1), (-1, 1), (-1, 1), (1, 1), (1, 1), (-1, 1))]
    elif which == "2d-checkerboard":
        mats = [matte(checks(su, sv)) for su, sv in ((-16, -16), (-16, 16), (-16, 16), (16, 16), (16, 16), (-16, 16))]
    elif which == "mix":
        t1, t2 = s.add_texture_windy(), checks(-16.0, -16.0)
        mats = [matte(s.add_texture_mix(t1, t2, c(a))) for a in (0.0, 0.51, 0.17, 0.68, 0.34, 1.0)]
    elif which == "constant":
        mats = [matte(c(v)) for v in ((1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (0, 1, 1), (1, 0, 1))]
    else:
        raise ValueError(which)
    s.add_light_infinite((1.0, 1.0, 1.0))
    R_ = host.rotate; T_ = host.translate
    add_sphere(s, ctm(host, T_((-1.8, 0, 1)), R_(15, (0, 1, 0)), R_(200, (1, 0, 0)), R_(-30, (0, 0, 1))), 0.8, material=mats[0])
    add_hyperboloid(s, ctm(host, T_((-1.8, 0, -1)), R_(15, (0, 1, 0)), R_(200, (1, 0, 0)), R_(-15, (0, 0, 1))), (0.6, 0.6, 0.6), (0.6, -0.6, -0.6), material=mats[1])
    add_quadric(s, "cone", ctm(host, T_((0, 0, 0.4)), R_(15, (0, 1, 0)), R_(15, (1, 0, 0)), R_(150, (0, 0, 1))), 0.8, 1.4, 0.0, material=mats[2])
    add_quadric(s, "paraboloid", ctm(host, T_((-0.2, 0, -1.8)), R_(15, (0, 1, 0)), R_(15, (1, 0, 0)), R_(30, (0, 0, 1))), 0.8, 0.0, 1.4, material=mats[3])
    add_quadric(s, "cylinder", ctm(host, T_((1.8, 0, 0.75)), R_(15, (0, 1, 0)), R_(15, (1, 0, 0)), R_(30, (0, 0, 1))), 0.8, -0.6, 0.6, material=mats[4])
    add_quadric(s, "disk", ctm(host, T_((1.8, 0, -1)), R_(200, (0, 1, 0)), R_(-150, (1, 0, 0)), R_(210, (0, 0, 1))), 0.8, 0.0, 0.0, material=mats[5])
    t = ctm(host, T_((0, -1, 0)))
    s.add_mesh(host.transform_points(t[0], np.array([[-20, 0, -20], [20, 0, -20], [20, 0, 20], [-20, 0, 20]], np.float32)), QUAD_IDX, s.add_material_matte((0.5, 0.5, 0.5)), UV=QUAD_ST)
    camera_film(s, host, (0, 22, 0), (0, 0, 0), (0, 0, 1), 15.0, 400, 400, spp)
    s.build_accel(0, 4)
    return dict(max_depth=5, render="textures_" + which)


def lights_diffuse(s, host, spp=128, res=400):
    """scenes/lights/diffuse.pbrt -> renders/lights/diffuse.png: a SPHERE of radius 3 (oracle only) as the shape of a DiffuseAreaLight (L = 20), over the cube and the checkered floor"""
    import ctypes as C
    from test_oracle_sphere import add_sphere
    add_sphere(s, ctm(host, host.translate((-10, 0, 10))), 3.0, material=s.add_material_matte((0.0, 0.0, 0.0)))
    L = s.b.lib
    L.oracle_make_last_sphere_a_light.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
    s._chk(L.oracle_make_last_sphere_a_light(s.h, np.array([20.0, 20.0, 20.0], np.float32).ctypes.data_as(C.POINTER(C.c_float)), 0))
    _cube_and_checker_floor(s, host)
    camera_film(s, host, (0, 5, 3), (0, 0, 0), (0, 0, 1), 90.0, res, res, spp)
    s.build_accel(0, 4)
    return dict(max_depth=5, render="lights_diffuse")


def compare(rgb_linear, ref_u8, block=8):
    """-> dict: mean |delta| in 8-bit levels per pixel, fraction of pixels with a channel off by more than 12 levels, and the same two over block x block means (sampling
    noise averages out of those: the path integrator's estimator differs from Whitted's where a scene has an area-like light or several lights)"""
    mine = to_8bit(rgb_linear).astype(np.float64); ref = ref_u8.astype(np.float64)
    d = np.abs(mine - ref)
    h, w, _ = d.shape
    bm = lambda a: a[: h // block * block, : w // block * block].reshape(h // block, block, w // block, block, 3).mean((1, 3))
    db = np.abs(bm(mine) - bm(ref))
    return dict(mean=float(d.mean()), bad=float((d.max(-1) > 12).mean()), block_mean=float(db.mean()), block_bad=float((db.max(-1) > 6).mean()),
                bias=[float(v) for v in (mine - ref).mean((0, 1))])
