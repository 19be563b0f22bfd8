"""Textured test scenes shared by the CPU (oracle pins) and GPU (parity) tests."""
import numpy as np

import pbrt_hip


def make_image(w, h, seed=0, kind="noise"):
    """(h, w, 3) float32 in [0, 1], top row first."""
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.uniform(0.0, 1.0, (h, w, 3)).astype(np.float32)
    if kind == "checker":
        yy, xx = np.mgrid[0:h, 0:w]
        c = ((xx + yy) & 1).astype(np.float32)
        return np.stack([c, c, c], axis=2)
    if kind == "ramp":
        yy, xx = np.mgrid[0:h, 0:w]
        return np.stack([xx / max(w - 1, 1), yy / max(h - 1, 1), 0.25 + 0 * xx], axis=2).astype(np.float32)
    raise ValueError(kind)


def probe_points(n, seed):
    """uv in [-0.5, 1.5]^2 (outside [0,1] exercises the wrap modes) with a spread of footprints: zero, isotropic, anisotropic, huge."""
    rng = np.random.default_rng(seed)
    uv = rng.uniform(-0.5, 1.5, (n, 2)).astype(np.float32)
    d = np.zeros((n, 4), np.float32)
    k = n // 5
    d[k:2 * k] = (rng.uniform(-1, 1, (k, 4)) * 0.002).astype(np.float32)                 # small isotropic-ish
    d[2 * k:3 * k] = (rng.uniform(-1, 1, (k, 4)) * np.array([0.2, 0.001, 0.001, 0.2])).astype(np.float32)
    d[3 * k:4 * k, 0] = rng.uniform(0.001, 0.3, k).astype(np.float32)                      # one axis only: minor length 0
    d[4 * k:] = (rng.uniform(-1, 1, (n - 4 * k, 4)) * 3.0).astype(np.float32)             # beyond the coarsest level
    return uv, d


def textured_quad_scene(scene, host, tex_builder, res=64, spp=4, sigma=0.0, lens_radius=0.0, tilt=True, instance=False, extra=None, material=None, orthographic=False, environment=False):
    """A ground quad with UVs tiled 3 x 3 under a matte material whose Kd is the texture `tex_builder(scene)` returns, seen at a
    grazing angle (strongly anisotropic footprints near the horizon) and lit by a white environment; a small mirror-free matte
    block above it gives the bounce rays something to shadow.  instance=True places the quad through an ObjectInstance.
    material(scene, tex) -> material id replaces the textured matte."""
    tex = tex_builder(scene)
    mat = material(scene, tex) if material is not None else scene.add_material_matte_tex(tex, sigma)
    grey = scene.add_material_matte((0.6, 0.6, 0.6), 0.0)
    P = np.array([[-4, -4, 0], [4, -4, 0], [4, 4, 0], [-4, 4, 0]], np.float32)
    UV = np.array([[0, 0], [3, 0], [3, 3], [0, 3]], np.float32)
    idx = np.array([0, 1, 2, 0, 2, 3], np.uint32)
    if instance:
        ob = scene.object_begin(); scene.add_mesh(P, idx, mat, UV=UV); scene.object_end()
        t = host.compose(host.translate([0.3, 0.1, 0.0]), host.rotate(25.0, [0, 0, 1]))
        scene.add_instance(ob, t[0], t[1])
    else:
        scene.add_mesh(P, idx, mat, UV=UV)
    B = np.array([[-0.5, -0.5, 0.2], [0.5, -0.5, 0.2], [0.5, 0.5, 0.2], [-0.5, 0.5, 0.2], [0, 0, 1.2]], np.float32)
    scene.add_mesh(B, np.array([0, 1, 4, 1, 2, 4, 2, 3, 4, 3, 0, 4], np.uint32), grey)
    if extra is not None:
        extra(scene)
    scene.add_light_infinite((1.0, 1.0, 1.0))
    eye = (0.0, -6.0, 1.2 if tilt else 6.0)
    w2c, c2w = host.look_at(eye, (0, 0, 0.2), (0, 0, 1))
    if environment:    # EnvironmentCamera a little above the floor: most of the lower hemisphere sees the texture at every footprint size
        scene.set_camera_environment(host.look_at((0.3, -1.0, 0.8), (0, 0, 0.2), (0, 0, 1))[1], res, res)
        cb, table, sb = host.film_box(res, res)
        scene.set_film(res, res, cb, (0.5, 0.5), table)
        scene.set_sampler(0, spp, sb)
        scene.build_accel(0, 4)
        return
    if orthographic:   # OrthographicCamera over a 5 x 5 window
        r2c = host.orthographic_raster_to_camera(res, res, np.float32([-2.5, 2.5, -2.5, 2.5]))
        scene.set_camera_orthographic(r2c, c2w, lens_radius=lens_radius, focal_distance=6.0 if lens_radius > 0 else 1e6)
        cb, table, sb = host.film_box(res, res)
        scene.set_film(res, res, cb, (0.5, 0.5), table)
        scene.set_sampler(0, spp, sb)
        scene.build_accel(0, 4)
        return
    r2c = host.perspective_raster_to_camera(40.0, res, res)
    if lens_radius > 0:
        scene.set_camera_perspective(r2c, c2w, lens_radius=lens_radius, focal_distance=6.0)
    else:
        scene.set_camera_perspective(r2c, c2w)
    cb, table, sb = host.film_box(res, res)
    scene.set_film(res, res, cb, (0.5, 0.5), table)
    scene.set_sampler(0, spp, sb)
    scene.build_accel(0, 4)
