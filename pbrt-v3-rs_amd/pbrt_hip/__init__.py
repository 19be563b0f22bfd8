"""pbrt_hip — thin ctypes host binding over libpbrt_hip.so (include/pbrt_hip.h).

This is plumbing, not the product: every compute call goes straight through the C ABI into hand-written gfx950
kernels.  The class layout mirrors the reference's plugin surface for the path: a `Scene` is captured
with the same calls the reference's factories make (api/src/graphics_state.rs:254-720) and `render_path` stands in
for `Integrator::render` (core/src/integrator/mod.rs:16-39).

`Binding(lib, prefix)` is deliberately generic over the symbol prefix so that the TEST harness can drive its CPU
checker through the identical Python code; nothing in this package imports, loads or links the checker.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PBRT_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "libpbrt_hip.so")   # PBRT_HIP_LIB: another build of the library (same-box A/B, scripts/build_variant.sh)

OK = 0
ERR_INVALID_ARG, ERR_STATE, ERR_DEVICE, ERR_NO_DEVICE, ERR_UNSUPPORTED, ERR_OOM = -1, -2, -3, -4, -5, -6

RAY_DTYPE = np.dtype([("o", "<f4", 3), ("t_max", "<f4"), ("d", "<f4", 3), ("time", "<f4")])
HIT_DTYPE = np.dtype([("t", "<f4"), ("prim", "<u4"), ("b0", "<f4"), ("b1", "<f4"), ("b2", "<f4"), ("pad", "<u4", 3)])
assert RAY_DTYPE.itemsize == 32 and HIT_DTYPE.itemsize == 32


class Stats(C.Structure):
    _fields_ = [
        ("camera_rays", C.c_uint64), ("regular_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
        ("paths_zero_radiance", C.c_uint64), ("paths_total", C.c_uint64),
        ("render_seconds", C.c_double), ("extend_seconds", C.c_double), ("shadow_seconds", C.c_double),
        ("shade_seconds", C.c_double), ("extend_launches", C.c_uint64), ("shadow_launches", C.c_uint64),
        ("light_distributions_created", C.c_uint64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class PbrtHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pbrt_hip error {code}: {msg}")
        self.code = code


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype)) if a is not None else None


class Binding:
    """Loads a shared library exporting the include/pbrt_hip.h entry points under `prefix`."""

    def __init__(self, path: str, prefix: str = "pbrt_hip_"):
        if not os.path.exists(path):
            raise PbrtHipError(ERR_NO_DEVICE, f"{path} not built — run __graft_entry__.build() (no CPU fallback exists)")
        self.lib = C.CDLL(path)
        self.prefix = prefix
        self.path = path
        f = self.fn
        vp, fp, ip, u32p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_uint32)
        sig = {
            "device_count": (C.c_int, []),
            "scene_create": (vp, [C.c_int]),
            "scene_destroy": (None, [vp]),
            "last_error": (C.c_char_p, [vp]),
            "add_material_matte": (C.c_int, [vp, fp, C.c_float, u32p]),
            "add_mesh": (C.c_int, [vp, fp, C.c_uint32, u32p, C.c_uint32, fp, fp, fp, C.c_uint32, C.c_int32, C.c_uint32, C.c_float, C.c_float]),
            "add_light_infinite": (C.c_int, [vp, fp, fp, fp]),
            "add_light_distant": (C.c_int, [vp, fp, fp]),
            "add_light_point": (C.c_int, [vp, fp, fp]),
            "add_light_spot": (C.c_int, [vp, fp, fp, fp, C.c_float, C.c_float]),
            "add_light_projection": (C.c_int, [vp, fp, fp, fp, C.c_float, C.c_int, C.c_int, fp]),
            "add_light_goniometric": (C.c_int, [vp, fp, fp, fp, C.c_int, C.c_int, fp]),
            "add_light_diffuse_area": (C.c_int, [vp, fp, C.c_int, C.c_uint32, u32p]),
            "set_camera_perspective": (C.c_int, [vp, fp, fp, C.c_float, C.c_float, C.c_float, C.c_float]),
            "set_camera_orthographic": (C.c_int, [vp, fp, fp, C.c_float, C.c_float, C.c_float, C.c_float]),
            "set_camera_environment": (C.c_int, [vp, fp, C.c_int, C.c_int, C.c_float, C.c_float]),
            "set_film": (C.c_int, [vp, C.c_int, C.c_int, ip, fp, fp, C.c_float, C.c_float]),
            "set_sampler": (C.c_int, [vp, C.c_int, C.c_uint32, ip, C.c_int]),
            "set_sobol_tables": (C.c_int, [vp, u32p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_size_t]),
            "build_accel": (C.c_int, [vp, C.c_int, C.c_int]),
            "world_bound": (C.c_int, [vp, fp]),
            "intersect_batch": (C.c_int, [vp, vp, vp, C.c_uint64]),
            "occluded_batch": (C.c_int, [vp, vp, vp, C.c_uint64]),
            "render_path": (C.c_int, [vp, C.c_int, C.c_float, C.c_int, ip, C.c_int, C.c_int, C.c_int, fp, fp, C.POINTER(Stats)]),
            "film_to_rgb": (C.c_int, [vp, fp, fp, fp]),
            "generate_camera_rays": (C.c_int, [vp, ip, C.c_uint32, vp, fp]),
        }
        for name, (res, args) in sig.items():
            fn = f(name)
            fn.restype, fn.argtypes = res, args
        self._optional = {
            "intersect_batch_device": (C.c_int, [vp, vp, vp, C.c_uint64, fp]),
            "occluded_batch_device": (C.c_int, [vp, vp, vp, C.c_uint64, fp]),
            "tile_buffer_floats": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64)]),
            "render_path_tiles_device": (C.c_int, [vp, C.c_int, C.c_float, C.c_int, ip, C.c_int, C.c_int, C.c_int, vp, C.POINTER(Stats)]),
            "merge_tiles_device": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(vp), fp, fp]),
            "add_material_none": (C.c_int, [vp, u32p]),
            "add_material_mirror": (C.c_int, [vp, fp, u32p]),
            "add_material_plastic": (C.c_int, [vp, fp, fp, C.c_float, C.c_int, u32p]),
            "add_material_glass": (C.c_int, [vp, fp, fp, C.c_float, C.c_float, C.c_float, C.c_int, u32p]),
            "add_material_metal": (C.c_int, [vp, fp, fp, C.c_float, C.c_float, C.c_int, u32p]),
            "add_material_uber": (C.c_int, [vp, fp, fp, fp, fp, fp, C.c_float, C.c_float, C.c_float, C.c_int, u32p]),
            "add_material_substrate": (C.c_int, [vp, fp, fp, C.c_float, C.c_float, C.c_int, u32p]),
            "add_material_translucent": (C.c_int, [vp, fp, fp, fp, fp, C.c_float, C.c_int, u32p]),
            "add_material_mix": (C.c_int, [vp, C.c_uint32, C.c_uint32, fp, u32p]),
            "object_begin": (C.c_int, [vp, u32p]),
            "object_end": (C.c_int, [vp]),
            "add_instance": (C.c_int, [vp, C.c_uint32, fp, fp]),
            "add_mipmap": (C.c_int, [vp, C.c_int, C.c_int, fp, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_float, u32p]),
            "add_texture_constant": (C.c_int, [vp, fp, u32p]),
            "add_texture_scale": (C.c_int, [vp, C.c_uint32, C.c_uint32, u32p]),
            "add_texture_mix": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint32, u32p]),
            "add_texture_imagemap": (C.c_int, [vp, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_float, u32p]),
            "add_texture_checkerboard": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, u32p]),
            "add_texture_uv": (C.c_int, [vp, C.c_float, C.c_float, C.c_float, C.c_float, u32p]),
            "add_texture_bilerp": (C.c_int, [vp, fp, fp, fp, fp, C.c_float, C.c_float, C.c_float, C.c_float, u32p]),
            "add_texture_dots": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_float, u32p]),
            "add_light_infinite_map": (C.c_int, [vp, fp, C.c_int, C.c_int, fp, fp, fp]),
            "add_texture_fbm": (C.c_int, [vp, fp, C.c_float, C.c_int, u32p]),
            "add_texture_wrinkled": (C.c_int, [vp, fp, C.c_float, C.c_int, u32p]),
            "add_texture_windy": (C.c_int, [vp, fp, u32p]),
            "add_texture_marble": (C.c_int, [vp, fp, C.c_float, C.c_int, C.c_float, C.c_float, u32p]),
            "add_texture_checkerboard3d": (C.c_int, [vp, C.c_uint32, C.c_uint32, fp, u32p]),
            "set_texture_mapping": (C.c_int, [vp, C.c_uint32, C.c_int, fp]),
            "add_material_matte_tex": (C.c_int, [vp, C.c_uint32, C.c_float, u32p]),
            "set_material_texture": (C.c_int, [vp, C.c_uint32, C.c_int, C.c_uint32]),
            "set_material_bump": (C.c_int, [vp, C.c_uint32, C.c_uint32]),
            "set_material_float_texture": (C.c_int, [vp, C.c_uint32, C.c_int, C.c_uint32]),
            "set_last_mesh_alpha_textures": (C.c_int, [vp, C.c_uint32, C.c_uint32]),
            "texture_eval_batch": (C.c_int, [vp, C.c_uint32, C.c_uint64, fp, fp]),
            "texture_eval_batch_nodiff": (C.c_int, [vp, C.c_uint32, C.c_uint64, fp, fp]),
            "mipmap_levels": (C.c_int, [vp, C.c_uint32, ip, ip]),
            "mipmap_level_texels": (C.c_int, [vp, C.c_uint32, C.c_int, fp]),
            "set_traversal_counting": (C.c_int, [vp, C.c_int]),
            "get_traversal_counts": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
            "accel_stats": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
            "accel_copy": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_uint64]),
            "build_accel_device": (C.c_int, [vp, C.c_int, C.c_int]),
            "scene_create_multi": (vp, [ip, C.c_int]),
            "scene_devices": (C.c_int, [vp, ip, C.c_int]),
            "selftest_rccl_gather": (C.c_int, [vp, C.c_uint32, C.POINTER(C.c_uint64)]),
        }
        for name, (res, args) in self._optional.items():
            if hasattr(self.lib, prefix + name):
                fn = f(name)
                fn.restype, fn.argtypes = res, args

    def fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def has(self, name):
        return hasattr(self.lib, self.prefix + name)


_default = None


def default_binding() -> Binding:
    """The product library.  Fails loudly when it has not been built."""
    global _default
    if _default is None:
        _default = Binding(LIB_PATH, "pbrt_hip_")
    return _default


class Host:
    """pbrt_hip_host_* helpers (include/pbrt_hip_host.h): the reference's scene set-up arithmetic, host side."""

    def __init__(self, binding: Binding | None = None):
        b = binding or default_binding()
        self.lib = b.lib
        fp = C.POINTER(C.c_float)
        L = self.lib
        L.pbrt_hip_host_look_at.restype = C.c_int
        L.pbrt_hip_host_gen_random_tris.argtypes = [C.c_uint64, C.c_uint64, fp, C.POINTER(C.c_uint32)]
        L.pbrt_hip_host_rotate.argtypes = [C.c_float, fp, fp, fp]
        L.pbrt_hip_host_perspective_raster_to_camera.argtypes = [C.c_float, C.c_int, C.c_int, fp, fp]
        L.pbrt_hip_host_orthographic_raster_to_camera.argtypes = [C.c_int, C.c_int, fp, fp]
        L.pbrt_hip_host_transform_points.argtypes = [fp, fp, fp, C.c_size_t]
        L.pbrt_hip_host_transform_vectors.argtypes = [fp, fp, fp, C.c_size_t]
        L.pbrt_hip_host_transform_normals.argtypes = [fp, fp, fp, C.c_size_t]
        ip = C.POINTER(C.c_int)
        L.pbrt_hip_host_film_filter.argtypes = [C.c_int, fp, C.c_int, C.c_int, fp, fp, ip, fp, ip]
        L.pbrt_hip_host_film_filter.restype = C.c_int
        L.pbrt_hip_host_spot.argtypes = [fp, fp, fp, fp, C.c_float, C.c_float, fp, fp, fp]
        L.pbrt_hip_host_swaps_handedness.argtypes = [fp]
        L.pbrt_hip_host_swaps_handedness.restype = C.c_int

    @staticmethod
    def _m():
        return np.zeros(16, np.float32)

    def look_at(self, pos, look, up):
        m, mi = self._m(), self._m()
        rc = self.lib.pbrt_hip_host_look_at(_ptr(_f32(pos), C.c_float), _ptr(_f32(look), C.c_float), _ptr(_f32(up), C.c_float), _ptr(m, C.c_float), _ptr(mi, C.c_float))
        if rc != 0:
            raise PbrtHipError(ERR_INVALID_ARG, "look_at: up vector and viewing direction are parallel")
        return m, mi  # world->camera, camera->world

    def translate(self, d):
        m, mi = self._m(), self._m()
        self.lib.pbrt_hip_host_translate(_ptr(_f32(d), C.c_float), _ptr(m, C.c_float), _ptr(mi, C.c_float))
        return m, mi

    def scale(self, s):
        m, mi = self._m(), self._m()
        self.lib.pbrt_hip_host_scale(_ptr(_f32(s), C.c_float), _ptr(m, C.c_float), _ptr(mi, C.c_float))
        return m, mi

    def rotate(self, theta_deg, axis):
        m, mi = self._m(), self._m()
        self.lib.pbrt_hip_host_rotate(C.c_float(theta_deg), _ptr(_f32(axis), C.c_float), _ptr(m, C.c_float), _ptr(mi, C.c_float))
        return m, mi

    def compose(self, a, b):
        m, mi = self._m(), self._m()
        self.lib.pbrt_hip_host_compose(_ptr(a[0], C.c_float), _ptr(a[1], C.c_float), _ptr(b[0], C.c_float), _ptr(b[1], C.c_float), _ptr(m, C.c_float), _ptr(mi, C.c_float))
        return m, mi

    def screen_window(self, xres, yres):
        s = np.zeros(4, np.float32)
        self.lib.pbrt_hip_host_screen_window(xres, yres, _ptr(s, C.c_float))
        return s

    def perspective_raster_to_camera(self, fov, xres, yres, screen=None):
        if screen is None:
            screen = self.screen_window(xres, yres)
        m = self._m()
        self.lib.pbrt_hip_host_perspective_raster_to_camera(C.c_float(fov), xres, yres, _ptr(_f32(screen), C.c_float), _ptr(m, C.c_float))
        return m

    def orthographic_raster_to_camera(self, xres, yres, screen=None):
        if screen is None:
            screen = self.screen_window(xres, yres)
        m = self._m()
        self.lib.pbrt_hip_host_orthographic_raster_to_camera(xres, yres, _ptr(_f32(screen), C.c_float), _ptr(m, C.c_float))
        return m

    def film_box(self, xres, yres, crop_window=(0.0, 1.0, 0.0, 1.0), radius=(0.5, 0.5)):
        cb, sb, table = np.zeros(4, np.int32), np.zeros(4, np.int32), np.zeros(256, np.float32)
        self.lib.pbrt_hip_host_film_box(xres, yres, _ptr(_f32(crop_window), C.c_float), _ptr(_f32(radius), C.c_float), _ptr(cb, C.c_int), _ptr(table, C.c_float), _ptr(sb, C.c_int))
        return cb, table, sb

    FILTER_KINDS = {"box": 0, "gaussian": 1, "mitchell": 2, "sinc": 3, "triangle": 4}

    def film_filter(self, kind, xres, yres, radius, params=(0.0, 0.0), crop_window=(0.0, 1.0, 0.0, 1.0)):
        """Film::new for any filter of filters/src (kind: name or 0..4) -> (cropped bounds, 16x16 table, sample bounds)."""
        k = self.FILTER_KINDS[kind] if isinstance(kind, str) else int(kind)
        cb, sb, table = np.zeros(4, np.int32), np.zeros(4, np.int32), np.zeros(256, np.float32)
        rc = self.lib.pbrt_hip_host_film_filter(k, _ptr(_f32(params), C.c_float), xres, yres, _ptr(_f32(crop_window), C.c_float), _ptr(_f32(radius), C.c_float),
                                                _ptr(cb, C.c_int), _ptr(table, C.c_float), _ptr(sb, C.c_int))
        if rc != 0:
            raise PbrtHipError(ERR_INVALID_ARG, f"film_filter: unknown filter kind {kind}")
        return cb, table, sb

    def invert(self, m):
        out = self._m()
        self.lib.pbrt_hip_host_invert(_ptr(_f32(m), C.c_float), _ptr(out, C.c_float))
        return out

    def transform_vectors(self, m, v):
        v = _f32(v, (-1, 3)); out = np.empty_like(v)
        self.lib.pbrt_hip_host_transform_vectors(_ptr(_f32(m), C.c_float), _ptr(v, C.c_float), _ptr(out, C.c_float), len(v))
        return out

    def transform_normals(self, m_inv, n):
        n = _f32(n, (-1, 3)); out = np.empty_like(n)
        self.lib.pbrt_hip_host_transform_normals(_ptr(_f32(m_inv), C.c_float), _ptr(n, C.c_float), _ptr(out, C.c_float), len(n))
        return out

    def swaps_handedness(self, m):
        return bool(self.lib.pbrt_hip_host_swaps_handedness(_ptr(_f32(m), C.c_float)))

    def distant_direction(self, l2w, frm, to):
        w = np.zeros(3, np.float32)
        self.lib.pbrt_hip_host_distant_direction(_ptr(_f32(l2w), C.c_float), _ptr(_f32(frm), C.c_float), _ptr(_f32(to), C.c_float), _ptr(w, C.c_float))
        return w

    def spot(self, ctm, frm, to, cone_angle=30.0, cone_delta=5.0):
        """SpotLight parameters -> (light_to_world, world_to_light, cos_total_width, cos_falloff_start)."""
        l2w, w2l, c = self._m(), self._m(), np.zeros(2, np.float32)
        self.lib.pbrt_hip_host_spot(_ptr(_f32(ctm[0]), C.c_float), _ptr(_f32(ctm[1]), C.c_float), _ptr(_f32(frm), C.c_float), _ptr(_f32(to), C.c_float),
                                    C.c_float(cone_angle), C.c_float(cone_delta), _ptr(l2w, C.c_float), _ptr(w2l, C.c_float), _ptr(c, C.c_float))
        return l2w, w2l, float(c[0]), float(c[1])

    def point_position(self, l2w, l2w_inv, frm):
        p = np.zeros(3, np.float32)
        self.lib.pbrt_hip_host_point_position(_ptr(_f32(l2w), C.c_float), _ptr(_f32(l2w_inv), C.c_float), _ptr(_f32(frm), C.c_float), _ptr(p, C.c_float))
        return p

    def transform_points(self, m, pts):
        pts = _f32(pts, (-1, 3)); out = np.empty_like(pts)
        self.lib.pbrt_hip_host_transform_points(_ptr(_f32(m), C.c_float), _ptr(pts, C.c_float), _ptr(out, C.c_float), len(pts))
        return out

    def gen_random_tris(self, n_tris, seed):
        P = np.empty((3 * n_tris, 3), np.float32)
        idx = np.empty(3 * n_tris, np.uint32)
        self.lib.pbrt_hip_host_gen_random_tris(n_tris, seed, _ptr(P, C.c_float), _ptr(idx, C.c_uint32))
        return P, idx


IDENTITY = np.eye(4, dtype=np.float32).reshape(16)


class Scene:
    """One captured scene on one device (PbrtHipScene*).  Not re-entrant, like the C handle."""

    def __init__(self, binding: Binding | None = None, device: int = 0, devices=None):
        """device: one GPU.  devices: a list of ordinals -> ONE handle driving all of them (pbrt_hip_scene_create_multi; an ordinal may repeat)."""
        self.b = binding or default_binding()
        if devices is not None:
            arr = np.ascontiguousarray(devices, dtype=np.int32)
            self.h = self.b.fn("scene_create_multi")(_ptr(arr, C.c_int), len(arr))
        else:
            self.h = self.b.fn("scene_create")(device)
        if not self.h:
            msg = self.b.fn("last_error")(None)
            raise PbrtHipError(ERR_NO_DEVICE, (msg or b"scene_create failed").decode())
        self.film_shape = None
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self.b.fn("scene_destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def last_error(self) -> str:
        """pbrt_hip_last_error: the text of the handle's last failed call (or a warning a successful call left), "" when there is none."""
        return (self.b.fn("last_error")(self.h) or b"").decode()

    def _chk(self, rc):
        if rc != OK:
            msg = self.b.fn("last_error")(self.h)
            raise PbrtHipError(rc, (msg or b"").decode())

    # ---- capture -------------------------------------------------------------------------------------------------
    def add_material_matte(self, kd=(0.5, 0.5, 0.5), sigma=0.0) -> int:
        out = C.c_uint32(0)
        self._chk(self.b.fn("add_material_matte")(self.h, _ptr(_f32(kd), C.c_float), C.c_float(sigma), C.byref(out)))
        return out.value

    def _mat(self, name, *args):
        mid = C.c_uint32()
        self._chk(self.b.fn(name)(self.h, *args, C.byref(mid)))
        return mid.value

    @staticmethod
    def _rgb(v):
        return _ptr(_f32(v), C.c_float)

    def add_material_none(self) -> int:
        return self._mat("add_material_none")

    def add_material_mirror(self, kr=(0.9, 0.9, 0.9)) -> int:
        return self._mat("add_material_mirror", self._rgb(kr))

    def add_material_plastic(self, kd=(0.25, 0.25, 0.25), ks=(0.25, 0.25, 0.25), roughness=0.1, remap_roughness=True) -> int:
        return self._mat("add_material_plastic", self._rgb(kd), self._rgb(ks), C.c_float(roughness), int(remap_roughness))

    def add_material_glass(self, kr=(1, 1, 1), kt=(1, 1, 1), uroughness=0.0, vroughness=0.0, eta=1.5, remap_roughness=True) -> int:
        return self._mat("add_material_glass", self._rgb(kr), self._rgb(kt), C.c_float(uroughness), C.c_float(vroughness), C.c_float(eta), int(remap_roughness))

    def add_material_metal(self, eta, k, uroughness=0.01, vroughness=0.01, remap_roughness=True) -> int:
        return self._mat("add_material_metal", self._rgb(eta), self._rgb(k), C.c_float(uroughness), C.c_float(vroughness), int(remap_roughness))

    def add_material_uber(self, kd=(0.25, 0.25, 0.25), ks=(0.25, 0.25, 0.25), kr=(0, 0, 0), kt=(0, 0, 0), opacity=(1, 1, 1), uroughness=0.1, vroughness=0.1,
                          eta=1.5, remap_roughness=True) -> int:
        return self._mat("add_material_uber", self._rgb(kd), self._rgb(ks), self._rgb(kr), self._rgb(kt), self._rgb(opacity), C.c_float(uroughness),
                         C.c_float(vroughness), C.c_float(eta), int(remap_roughness))

    def add_material_substrate(self, kd=(0.5, 0.5, 0.5), ks=(0.5, 0.5, 0.5), uroughness=0.1, vroughness=0.1, remap_roughness=True) -> int:
        return self._mat("add_material_substrate", self._rgb(kd), self._rgb(ks), C.c_float(uroughness), C.c_float(vroughness), int(remap_roughness))

    def add_material_translucent(self, kd=(0.25,) * 3, ks=(0.25,) * 3, reflect=(0.5,) * 3, transmit=(0.5,) * 3, roughness=0.1, remap_roughness=True) -> int:
        return self._mat("add_material_translucent", self._rgb(kd), self._rgb(ks), self._rgb(reflect), self._rgb(transmit), C.c_float(roughness), int(remap_roughness))

    def add_material_mix(self, material1, material2, amount=(0.5, 0.5, 0.5)) -> int:
        return self._mat("add_material_mix", int(material1), int(material2), self._rgb(amount))

    def add_mesh(self, P, indices, material, N=None, S=None, UV=None, first_area_light=-1, reverse_orientation=False,
                 swaps_handedness=False, alpha=1.0, shadow_alpha=1.0):
        P = _f32(P, (-1, 3)); idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        N = _f32(N, (-1, 3)) if N is not None else None
        S = _f32(S, (-1, 3)) if S is not None else None
        UV = _f32(UV, (-1, 2)) if UV is not None else None
        flags = (1 if reverse_orientation else 0) | (2 if swaps_handedness else 0)
        self._chk(self.b.fn("add_mesh")(self.h, _ptr(P, C.c_float), len(P), _ptr(idx, C.c_uint32), len(idx) // 3, _ptr(N, C.c_float),
                                        _ptr(S, C.c_float), _ptr(UV, C.c_float), material, first_area_light, flags, C.c_float(alpha), C.c_float(shadow_alpha)))

    def object_begin(self) -> int:
        """ObjectBegin (api/src/lib.rs:911-925): meshes added until object_end() belong to the returned object."""
        oid = C.c_uint32()
        self._chk(self.b.fn("object_begin")(self.h, C.byref(oid)))
        return oid.value

    def object_end(self):
        self._chk(self.b.fn("object_end")(self.h))

    def add_instance(self, object_id, instance_to_world, world_to_instance):
        """ObjectInstance (lib.rs:942-1000): one TransformedPrimitive in the scene's primitive list."""
        self._chk(self.b.fn("add_instance")(self.h, object_id, _ptr(_f32(instance_to_world), C.c_float), _ptr(_f32(world_to_instance), C.c_float)))

    def add_light_infinite(self, L=(1, 1, 1), light_to_world=None, world_to_light=None):
        l2w = _f32(light_to_world if light_to_world is not None else IDENTITY)
        w2l = _f32(world_to_light if world_to_light is not None else IDENTITY)
        self._chk(self.b.fn("add_light_infinite")(self.h, _ptr(_f32(L), C.c_float), _ptr(l2w, C.c_float), _ptr(w2l, C.c_float)))

    def add_light_infinite_map(self, L, image, light_to_world=None, world_to_light=None):
        """InfiniteAreaLight with a radiance map: image (H, W, 3) float32 as an image reader returns it (top row first)."""
        img = np.ascontiguousarray(image, dtype=np.float32)
        l2w = _f32(light_to_world if light_to_world is not None else IDENTITY)
        w2l = _f32(world_to_light if world_to_light is not None else IDENTITY)
        self._chk(self.b.fn("add_light_infinite_map")(self.h, _ptr(_f32(L), C.c_float), img.shape[1], img.shape[0], _ptr(img, C.c_float), _ptr(l2w, C.c_float), _ptr(w2l, C.c_float)))

    def add_light_distant(self, L, w_light_world):
        self._chk(self.b.fn("add_light_distant")(self.h, _ptr(_f32(L), C.c_float), _ptr(_f32(w_light_world), C.c_float)))

    def add_light_point(self, I, p_world):
        self._chk(self.b.fn("add_light_point")(self.h, _ptr(_f32(I), C.c_float), _ptr(_f32(p_world), C.c_float)))

    def add_light_spot(self, I, light_to_world, world_to_light, cos_total_width, cos_falloff_start):
        self._chk(self.b.fn("add_light_spot")(self.h, _ptr(_f32(I), C.c_float), _ptr(_f32(light_to_world), C.c_float), _ptr(_f32(world_to_light), C.c_float),
                                              C.c_float(cos_total_width), C.c_float(cos_falloff_start)))

    def add_light_projection(self, I, light_to_world, world_to_light, fov, image=None):
        """ProjectionLight: image (H, W, 3) float32, top row first, or None (white inside the frustum)."""
        img = None if image is None else np.ascontiguousarray(image, dtype=np.float32)
        self._chk(self.b.fn("add_light_projection")(self.h, _ptr(_f32(I), C.c_float), _ptr(_f32(light_to_world), C.c_float), _ptr(_f32(world_to_light), C.c_float), C.c_float(fov),
                                                    0 if img is None else img.shape[1], 0 if img is None else img.shape[0], None if img is None else _ptr(img, C.c_float)))

    def add_light_goniometric(self, I, light_to_world, world_to_light, image=None):
        """GonioPhotometricLight: image (H, W, 3) float32 indexed by (phi / 2 pi, theta / pi), or None (a point light)."""
        img = None if image is None else np.ascontiguousarray(image, dtype=np.float32)
        self._chk(self.b.fn("add_light_goniometric")(self.h, _ptr(_f32(I), C.c_float), _ptr(_f32(light_to_world), C.c_float), _ptr(_f32(world_to_light), C.c_float),
                                                     0 if img is None else img.shape[1], 0 if img is None else img.shape[0], None if img is None else _ptr(img, C.c_float)))

    def add_light_diffuse_area(self, L, n_tris, two_sided=False) -> int:
        out = C.c_uint32(0)
        self._chk(self.b.fn("add_light_diffuse_area")(self.h, _ptr(_f32(L), C.c_float), 1 if two_sided else 0, n_tris, C.byref(out)))
        return out.value

    def set_camera_perspective(self, raster_to_camera, camera_to_world, lens_radius=0.0, focal_distance=1e6, shutter_open=0.0, shutter_close=1.0):
        self._chk(self.b.fn("set_camera_perspective")(self.h, _ptr(_f32(raster_to_camera), C.c_float), _ptr(_f32(camera_to_world), C.c_float),
                                                      C.c_float(lens_radius), C.c_float(focal_distance), C.c_float(shutter_open), C.c_float(shutter_close)))

    def set_camera_orthographic(self, raster_to_camera, camera_to_world, lens_radius=0.0, focal_distance=1e6, shutter_open=0.0, shutter_close=1.0):
        self._chk(self.b.fn("set_camera_orthographic")(self.h, _ptr(_f32(raster_to_camera), C.c_float), _ptr(_f32(camera_to_world), C.c_float),
                                                       C.c_float(lens_radius), C.c_float(focal_distance), C.c_float(shutter_open), C.c_float(shutter_close)))

    def set_camera_environment(self, camera_to_world, xres, yres, shutter_open=0.0, shutter_close=1.0):
        self._chk(self.b.fn("set_camera_environment")(self.h, _ptr(_f32(camera_to_world), C.c_float), int(xres), int(yres), C.c_float(shutter_open), C.c_float(shutter_close)))

    def set_film(self, xres, yres, cropped_bounds, radius, table, scale=1.0, max_sample_luminance=float("inf")):
        cb = np.ascontiguousarray(cropped_bounds, dtype=np.int32)
        self._chk(self.b.fn("set_film")(self.h, xres, yres, _ptr(cb, C.c_int), _ptr(_f32(radius), C.c_float), _ptr(_f32(table), C.c_float),
                                        C.c_float(scale), C.c_float(max_sample_luminance)))
        self.film_shape = (int(cb[3] - cb[1]), int(cb[2] - cb[0]))
        self.cropped_bounds = cb

    def set_sampler(self, kind, spp, sample_bounds, sample_at_pixel_center=False):
        sb = np.ascontiguousarray(sample_bounds, dtype=np.int32)
        self._chk(self.b.fn("set_sampler")(self.h, kind, spp, _ptr(sb, C.c_int), 1 if sample_at_pixel_center else 0))
        self.sample_bounds = sb

    def set_sobol_tables(self, m32, vdc, vdc_inv):
        """Generator matrices for the Sobol sampler (data tables the host owns; the library embeds none)."""
        m32 = np.ascontiguousarray(m32, dtype=np.uint32); vdc = np.ascontiguousarray(vdc, dtype=np.uint64); vdci = np.ascontiguousarray(vdc_inv, dtype=np.uint64)
        assert len(vdc) == len(vdci)
        self._chk(self.b.fn("set_sobol_tables")(self.h, _ptr(m32, C.c_uint32), len(m32), _ptr(vdc, C.c_uint64), _ptr(vdci, C.c_uint64), len(vdc)))

    def build_accel_device(self, split_method=1, max_prims_in_node=4):
        """The SAH (0) or HLBVH (1) tree constructed on the GPU: same tree as build_accel(split_method, ..)."""
        self._chk(self.b.fn("build_accel_device")(self.h, split_method, max_prims_in_node))

    def build_accel(self, split_method=0, max_prims_in_node=4):
        self._chk(self.b.fn("build_accel")(self.h, split_method, max_prims_in_node))

    def build_accel_best(self, split_method=0, max_prims_in_node=4):
        """build_accel_device where it applies (SAH / HLBVH trees of scenes without object instances), build_accel elsewhere: the same tree either way."""
        if split_method in (0, 1) and self.b.has("build_accel_device"):
            try:
                return self.build_accel_device(split_method, max_prims_in_node)
            except PbrtHipError as e:
                if e.code != ERR_UNSUPPORTED:
                    raise
        self.build_accel(split_method, max_prims_in_node)

    def world_bound(self):
        out = np.zeros(6, np.float32)
        self._chk(self.b.fn("world_bound")(self.h, _ptr(out, C.c_float)))
        return out

    # ---- hot path ------------------------------------------------------------------------------------------------
    def intersect_batch(self, rays):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(len(rays), HIT_DTYPE)
        self._chk(self.b.fn("intersect_batch")(self.h, rays.ctypes.data, hits.ctypes.data, len(rays)))
        return hits

    def occluded_batch(self, rays):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        out = np.zeros(len(rays), np.uint8)
        self._chk(self.b.fn("occluded_batch")(self.h, rays.ctypes.data, out.ctypes.data, len(rays)))
        return out

    def intersect_batch_device(self, d_rays_ptr, d_hits_ptr, n):
        ms = C.c_float(0)
        self._chk(self.b.fn("intersect_batch_device")(self.h, d_rays_ptr, d_hits_ptr, n, C.byref(ms)))
        return ms.value

    def occluded_batch_device(self, d_rays_ptr, d_out_ptr, n):
        ms = C.c_float(0)
        self._chk(self.b.fn("occluded_batch_device")(self.h, d_rays_ptr, d_out_ptr, n, C.byref(ms)))
        return ms.value

    # ---- textures (include/pbrt_hip.h "textures") ------------------------------------------------------------------------------
    WRAP = {"repeat": 0, "black": 1, "clamp": 2}

    def add_mipmap(self, image, as_float=False, scale=1.0, gamma=False, trilinear=False, wrap="repeat", max_anisotropy=8.0):
        """image: (H, W, 3) float32 as an image reader returns it (top row first).  Returns the mipmap id."""
        img = np.ascontiguousarray(image, dtype=np.float32)
        assert img.ndim == 3 and img.shape[2] == 3
        out = C.c_uint32(0)
        self._chk(self.b.fn("add_mipmap")(self.h, img.shape[1], img.shape[0], _ptr(img, C.c_float), 1 if as_float else 0, C.c_float(scale), 1 if gamma else 0,
                                          0 if trilinear else 1, self.WRAP[wrap], C.c_float(max_anisotropy), C.byref(out)))
        return out.value

    def add_texture_constant(self, value):
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(value, np.float32), (3,)), dtype=np.float32)
        out = C.c_uint32(0); self._chk(self.b.fn("add_texture_constant")(self.h, _ptr(v, C.c_float), C.byref(out))); return out.value

    def add_texture_scale(self, t1, t2):
        out = C.c_uint32(0); self._chk(self.b.fn("add_texture_scale")(self.h, t1, t2, C.byref(out))); return out.value

    def add_texture_mix(self, t1, t2, amount):
        out = C.c_uint32(0); self._chk(self.b.fn("add_texture_mix")(self.h, t1, t2, amount, C.byref(out))); return out.value

    def add_texture_imagemap(self, mipmap, su=1.0, sv=1.0, du=0.0, dv=0.0):
        out = C.c_uint32(0)
        self._chk(self.b.fn("add_texture_imagemap")(self.h, mipmap, C.c_float(su), C.c_float(sv), C.c_float(du), C.c_float(dv), C.byref(out))); return out.value

    def add_texture_checkerboard(self, t1, t2, su=1.0, sv=1.0, du=0.0, dv=0.0, aa="closedform"):
        out = C.c_uint32(0)
        self._chk(self.b.fn("add_texture_checkerboard")(self.h, t1, t2, C.c_float(su), C.c_float(sv), C.c_float(du), C.c_float(dv), 0 if aa == "none" else 1, C.byref(out)))
        return out.value

    def add_texture_uv(self, su=1.0, sv=1.0, du=0.0, dv=0.0):
        out = C.c_uint32(0); self._chk(self.b.fn("add_texture_uv")(self.h, C.c_float(su), C.c_float(sv), C.c_float(du), C.c_float(dv), C.byref(out))); return out.value

    def add_texture_bilerp(self, v00, v01, v10, v11, su=1.0, sv=1.0, du=0.0, dv=0.0):
        vs = [np.ascontiguousarray(np.broadcast_to(np.asarray(v, np.float32), (3,)), dtype=np.float32) for v in (v00, v01, v10, v11)]
        out = C.c_uint32(0)
        self._chk(self.b.fn("add_texture_bilerp")(self.h, *[_ptr(v, C.c_float) for v in vs], C.c_float(su), C.c_float(sv), C.c_float(du), C.c_float(dv), C.byref(out)))
        return out.value

    def add_texture_dots(self, inside, outside, su=1.0, sv=1.0, du=0.0, dv=0.0):
        out = C.c_uint32(0)
        self._chk(self.b.fn("add_texture_dots")(self.h, inside, outside, C.c_float(su), C.c_float(sv), C.c_float(du), C.c_float(dv), C.byref(out))); return out.value

    def add_texture_fbm(self, m=IDENTITY, omega=0.5, octaves=8, wrinkled=False):
        out = C.c_uint32(0)
        self._chk(self.b.fn("add_texture_wrinkled" if wrinkled else "add_texture_fbm")(self.h, _ptr(_f32(m), C.c_float), C.c_float(omega), octaves, C.byref(out))); return out.value

    def add_texture_windy(self, m=IDENTITY):
        out = C.c_uint32(0); self._chk(self.b.fn("add_texture_windy")(self.h, _ptr(_f32(m), C.c_float), C.byref(out))); return out.value

    def add_texture_marble(self, m=IDENTITY, omega=0.5, octaves=8, scale=1.0, variation=0.2):
        out = C.c_uint32(0)
        self._chk(self.b.fn("add_texture_marble")(self.h, _ptr(_f32(m), C.c_float), C.c_float(omega), octaves, C.c_float(scale), C.c_float(variation), C.byref(out))); return out.value

    def add_texture_checkerboard3d(self, t1, t2, m=IDENTITY):
        out = C.c_uint32(0); self._chk(self.b.fn("add_texture_checkerboard3d")(self.h, t1, t2, _ptr(_f32(m), C.c_float), C.byref(out))); return out.value

    def set_texture_mapping(self, texture, kind, params):
        """kind "spherical" | "cylindrical" (params = world_to_texture, 16 floats) | "planar" (params = v1, v2, udelta, vdelta: 8 floats)"""
        k = {"spherical": 1, "cylindrical": 2, "planar": 3}[kind]
        prm = np.ascontiguousarray(np.asarray(params, np.float32).reshape(-1), dtype=np.float32)
        assert len(prm) == (8 if k == 3 else 16)
        self._chk(self.b.fn("set_texture_mapping")(self.h, texture, k, _ptr(prm, C.c_float)))

    def add_material_matte_tex(self, kd_texture, sigma=0.0):
        out = C.c_uint32(0); self._chk(self.b.fn("add_material_matte_tex")(self.h, kd_texture, C.c_float(sigma), C.byref(out))); return out.value

    PARAM = {"Kd": 0, "Ks": 1, "Kr": 2, "Kt": 3, "opacity": 4, "amount": 5, "eta": 6, "k": 7, "reflect": 8, "transmit": 9}

    def set_material_texture(self, material, param, texture):
        """param: "Kd" | "Ks" | "Kr" | "Kt" | "opacity" (uber) | "amount" (mix) | "eta" | "k" (metal) | "reflect" | "transmit" (translucent) — that parameter of `material` becomes `texture`, evaluated per hit."""
        self._chk(self.b.fn("set_material_texture")(self.h, material, self.PARAM[param], texture))

    def set_last_mesh_alpha_textures(self, alpha=None, shadow_alpha=None):
        """Float textures for the `alpha` / `shadowalpha` masks of the mesh added last (None keeps the constant)."""
        none = 0xFFFFFFFF
        self._chk(self.b.fn("set_last_mesh_alpha_textures")(self.h, none if alpha is None else alpha, none if shadow_alpha is None else shadow_alpha))

    FPARAM = {"sigma": 0, "uroughness": 1, "vroughness": 2, "index": 3}

    def set_material_float_texture(self, material, fparam, texture):
        """fparam: "sigma" (matte) | "uroughness" | "vroughness" | "roughness" (= both) | "index" (glass, uber) — that scalar of `material` becomes the float texture, evaluated per hit."""
        for f in (("uroughness", "vroughness") if fparam == "roughness" else (fparam,)):
            self._chk(self.b.fn("set_material_float_texture")(self.h, material, self.FPARAM[f], texture))

    def set_material_bump(self, material, texture):
        """Material::bump with the float texture `texture` as displacement map."""
        self._chk(self.b.fn("set_material_bump")(self.h, material, texture))

    def texture_eval(self, texture, uv, derivs=None, p=None, dpdx=None, dpdy=None):
        """Evaluates `texture` at uv (n,2) with (du/dx, dv/dx, du/dy, dv/dy) (n,4) and, for the 3D textures, the hit point p (n,3) with dp/dx, dp/dy;
        returns (n,3).  A probe for the parity tests."""
        uv = np.asarray(uv, np.float32).reshape(-1, 2)
        z3 = np.zeros((len(uv), 3), np.float32)
        d = np.zeros((len(uv), 4), np.float32) if derivs is None else np.asarray(derivs, np.float32).reshape(-1, 4)
        cols = [uv, d] + [z3 if a is None else np.asarray(a, np.float32).reshape(-1, 3) for a in (p, dpdx, dpdy)]
        inp = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype=np.float32)
        out = np.zeros((len(uv), 3), np.float32)
        self._chk(self.b.fn("texture_eval_batch")(self.h, texture, len(uv), _ptr(inp, C.c_float), _ptr(out, C.c_float)))
        if self.b.has("texture_eval_batch_nodiff") and not inp[:, 2:6].any() and not inp[:, 9:15].any():   # (the product only: the oracle has one evaluator)
            # contexts without differentials: the renderer evaluates them with the NODIFF form of the evaluator (every ray but a camera ray) — the two must agree bit for bit
            out2 = np.zeros_like(out)
            self._chk(self.b.fn("texture_eval_batch_nodiff")(self.h, texture, len(uv), _ptr(inp, C.c_float), _ptr(out2, C.c_float)))
            if not np.array_equal(out.view(np.uint32), out2.view(np.uint32)):
                raise AssertionError("texture_eval: the no-differentials evaluator differs from the general one on zero differentials")
        return out

    def mipmap_pyramid(self, mipmap):
        """The pyramid the host built: list of (h, w, 3) float32 arrays, finest first (row 0 = t 0 = the image's bottom row)."""
        n = C.c_int(0); wh = (C.c_int * 64)()
        self._chk(self.b.fn("mipmap_levels")(self.h, mipmap, C.byref(n), wh))
        levels = []
        for i in range(n.value):
            a = np.zeros((wh[2 * i + 1], wh[2 * i], 3), np.float32)
            self._chk(self.b.fn("mipmap_level_texels")(self.h, mipmap, i, _ptr(a, C.c_float)))
            levels.append(a)
        return levels

    def set_traversal_counting(self, on):
        """Measurement aid (include/pbrt_hip.h): traversal launches also tally nodes / triangle tests / rays.  Never timed."""
        self._chk(self.b.fn("set_traversal_counting")(self.h, 1 if on else 0))

    def traversal_counts(self):
        """{closest: nodes_passed, tri_tests, rays, ref_node_visits; any_hit: the same} since the last call (resets the tallies)."""
        c = (C.c_uint64 * 8)()
        self._chk(self.b.fn("get_traversal_counts")(self.h, c))
        return {"closest": {"nodes_passed": int(c[0]), "tri_tests": int(c[1]), "rays": int(c[2]), "ref_node_visits": int(c[2]) + 2 * int(c[0])},
                "any_hit": {"nodes_passed": int(c[3]), "tri_tests": int(c[4]), "rays": int(c[5]), "ref_node_visits": int(c[6])},
                "culled_pops": int(c[7])}

    def devices(self):
        out = np.zeros(64, np.int32)
        n = self.b.fn("scene_devices")(self.h, _ptr(out, C.c_int), 64)
        return [int(v) for v in out[:n]]

    def selftest_rccl_gather(self, n_floats=1 << 20):
        wrong = C.c_uint64(0)
        self._chk(self.b.fn("selftest_rccl_gather")(self.h, n_floats, C.byref(wrong)))
        return int(wrong.value)

    def accel_stats(self):
        """Sizes of the built acceleration structure in the device layout (measurement aid)."""
        c = (C.c_uint64 * 8)()
        self._chk(self.b.fn("accel_stats")(self.h, c))
        return {"interior_nodes": int(c[0]), "leaf_records": int(c[1]), "node_bytes": int(c[2]), "leaf_record_bytes": int(c[3]), "leaves": int(c[4]),
                "depth": int(c[5]), "max_leaf_prims": int(c[6]), "build_seconds": int(c[7]) * 1e-6}

    def accel_copy(self):
        """(nodes (n, 16) uint32, leaf records (m, 12) uint32): the built structure in the device layout, copied from wherever it lives (test aid)."""
        st = self.accel_stats()
        nodes = np.zeros((max(st["interior_nodes"], 1), 16), np.uint32); recs = np.zeros((max(st["leaf_records"], 1), 12), np.uint32)
        self._chk(self.b.fn("accel_copy")(self.h, nodes.ctypes.data, st["interior_nodes"], recs.ctypes.data, st["leaf_records"]))
        return nodes[:st["interior_nodes"]], recs[:st["leaf_records"]]

    def _film_hw(self):
        if self.film_shape is None:
            raise PbrtHipError(ERR_STATE, "set_film must be called before rendering")
        return self.film_shape

    def render_path(self, max_depth=5, rr_threshold=1.0, light_strategy=2, pixel_bounds=None, tile_size=16, tile_part=0, tile_parts=1, out=None):
        """out = (xyz, weight) arrays to fill instead of fresh ones (see merge_tiles_device)."""
        h, w = self._film_hw()
        if out is not None:
            xyz, wt = out
            assert xyz.shape == (h, w, 3) and wt.shape == (h, w) and xyz.dtype == np.float32 and wt.dtype == np.float32 and xyz.flags.c_contiguous and wt.flags.c_contiguous
        else:
            xyz = np.zeros((h, w, 3), np.float32); wt = np.zeros((h, w), np.float32)
        pb = np.ascontiguousarray(pixel_bounds if pixel_bounds is not None else self.sample_bounds, dtype=np.int32)
        st = Stats()
        self._chk(self.b.fn("render_path")(self.h, max_depth, C.c_float(rr_threshold), light_strategy, _ptr(pb, C.c_int), tile_size, tile_part, tile_parts,
                                           _ptr(xyz, C.c_float), _ptr(wt, C.c_float), C.byref(st)))
        return xyz, wt, st

    def film_to_rgb(self, xyz, weight):
        rgb = np.zeros_like(xyz)
        self._chk(self.b.fn("film_to_rgb")(self.h, _ptr(_f32(xyz), C.c_float), _ptr(_f32(weight), C.c_float), _ptr(rgb, C.c_float)))
        return rgb

    def generate_camera_rays(self, pixel_bounds, sample_index):
        pb = np.ascontiguousarray(pixel_bounds, dtype=np.int32)
        n = int((pb[2] - pb[0]) * (pb[3] - pb[1]))
        rays = np.zeros(n, RAY_DTYPE); pf = np.zeros((n, 2), np.float32)
        self._chk(self.b.fn("generate_camera_rays")(self.h, _ptr(pb, C.c_int), sample_index, rays.ctypes.data, _ptr(pf, C.c_float)))
        return rays, pf

    def tile_buffer_floats(self, tile_size, tile_part, tile_parts):
        out = C.c_uint64(0)
        self._chk(self.b.fn("tile_buffer_floats")(self.h, tile_size, tile_part, tile_parts, C.byref(out)))
        return out.value

    def render_path_tiles_device(self, d_tile_buffer_ptr, max_depth=5, rr_threshold=1.0, light_strategy=2, pixel_bounds=None, tile_size=16, tile_part=0, tile_parts=1):
        pb = np.ascontiguousarray(pixel_bounds if pixel_bounds is not None else self.sample_bounds, dtype=np.int32)
        st = Stats()
        self._chk(self.b.fn("render_path_tiles_device")(self.h, max_depth, C.c_float(rr_threshold), light_strategy, _ptr(pb, C.c_int), tile_size, tile_part, tile_parts,
                                                        d_tile_buffer_ptr, C.byref(st)))
        return st

    def merge_tiles_device(self, d_ptrs, tile_size=16, out=None):
        """out = (xyz (h, w, 3) float32, weight (h, w) float32) to fill instead of fresh arrays — page-locked ones (e.g. torch.empty(.., pin_memory=True).numpy())
        make the read-back of a large film a DMA transfer instead of a staged copy into freshly faulted pages."""
        h, w = self._film_hw()
        if out is not None:
            xyz, wt = out
            assert xyz.shape == (h, w, 3) and wt.shape == (h, w) and xyz.dtype == np.float32 and wt.dtype == np.float32 and xyz.flags.c_contiguous and wt.flags.c_contiguous
        else:
            xyz = np.zeros((h, w, 3), np.float32); wt = np.zeros((h, w), np.float32)
        arr = (C.c_void_p * len(d_ptrs))(*d_ptrs)
        self._chk(self.b.fn("merge_tiles_device")(self.h, tile_size, len(d_ptrs), arr, _ptr(xyz, C.c_float), _ptr(wt, C.c_float)))
        return xyz, wt


def read_ply(path):
    """Vertices and triangles of a PLY file as shapes/src/plymesh.rs:21-249 reads them: `x y z` of element vertex (any other vertex property is skipped here), faces
    `vertex_indices` / `vertex_index` of 3 or 4 (a quad a b c d -> a b c, d a c); ascii and binary, either endianness.  Returns (P float32 (n, 3), idx uint32 (3 n_tris,))."""
    with open(path, "rb") as f:
        data = f.read()
    end = data.index(b"end_header") + len(b"end_header")
    end = data.index(b"\n", end) + 1
    header = data[:end].decode("ascii", "replace").split("\n")
    fmt = None; elements = []
    for line in header:
        t = line.split()
        if not t: continue
        if t[0] == "format": fmt = t[1]
        elif t[0] == "element": elements.append({"name": t[1], "n": int(t[2]), "props": []})
        elif t[0] == "property" and elements:
            if t[1] == "list": elements[-1]["props"].append(("list", t[2], t[3], t[4]))
            else: elements[-1]["props"].append(("scalar", t[1], t[2]))
    types = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
             "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}
    P = None; tris = []
    if fmt == "ascii":
        tok = data[end:].split(); pos = 0
        for e in elements:
            if e["name"] == "vertex":
                names = [p[2] for p in e["props"]]; w = len(names)
                a = np.array(tok[pos:pos + w * e["n"]], dtype=np.float64).reshape(e["n"], w); pos += w * e["n"]
                P = np.stack([a[:, names.index(c)] for c in "xyz"], axis=1).astype(np.float32)
            else:
                for _ in range(e["n"]):
                    for pr in e["props"]:
                        if pr[0] == "list":
                            k = int(tok[pos]); v = [int(x) for x in tok[pos + 1:pos + 1 + k]]; pos += 1 + k
                            if e["name"] == "face" and pr[3] in ("vertex_indices", "vertex_index"): tris.append(v)
                        else: pos += 1
    else:
        bo = "<" if fmt == "binary_little_endian" else ">"
        off = end
        for e in elements:
            if all(p[0] == "scalar" for p in e["props"]):
                dt = np.dtype([(p[2], bo + types[p[1]]) for p in e["props"]])
                a = np.frombuffer(data, dt, e["n"], off); off += dt.itemsize * e["n"]
                if e["name"] == "vertex": P = np.stack([a[c] for c in "xyz"], axis=1).astype(np.float32)
            else:
                if e["name"] == "face" and len(e["props"]) == 1:   # the common case: one list per face; try the fixed-size fast paths first
                    _, ct, it, nm = e["props"][0]; cdt = np.dtype(bo + types[ct]); idt = np.dtype(bo + types[it])
                    done = False
                    for k in (3, 4):
                        rec = np.dtype([("n", cdt), ("v", idt, (k,))])
                        if off + rec.itemsize * e["n"] <= len(data):
                            a = np.frombuffer(data, rec, e["n"], off)
                            if (a["n"] == k).all():
                                if nm in ("vertex_indices", "vertex_index"): tris = a["v"].astype(np.int64)
                                off += rec.itemsize * e["n"]; done = True; break
                    if done: continue
                for _ in range(e["n"]):
                    for pr in e["props"]:
                        if pr[0] == "list":
                            cdt = np.dtype(bo + types[pr[1]]); idt = np.dtype(bo + types[pr[2]])
                            k = int(np.frombuffer(data, cdt, 1, off)[0]); off += cdt.itemsize
                            v = np.frombuffer(data, idt, k, off).astype(np.int64).tolist(); off += idt.itemsize * k
                            if e["name"] == "face" and pr[3] in ("vertex_indices", "vertex_index"): tris.append(v)
                        else: off += np.dtype(types[pr[1]]).itemsize
    if P is None: raise ValueError(f"{path}: no vertex element")
    if isinstance(tris, np.ndarray):
        idx = tris.reshape(-1, 3) if tris.shape[1] == 3 else np.concatenate([tris[:, [0, 1, 2]], tris[:, [3, 0, 2]]], axis=1).reshape(-1, 3)
    else:
        out = []
        for v in tris:
            if len(v) == 3: out.append(v)
            elif len(v) == 4: out.append([v[0], v[1], v[2]]); out.append([v[3], v[0], v[2]])   # plymesh.rs:228-236
        idx = np.array(out, np.int64).reshape(-1, 3)
    return np.ascontiguousarray(P), np.ascontiguousarray(idx.reshape(-1), dtype=np.uint32)


@dataclass
class SceneSpec:
    """The synthetic measurement scene of BASELINE.md §3 / SURVEY §8d, as plain data, so that the same description can
    be captured into any Binding (product or, in tests, the oracle)."""
    n_tris: int = 1000
    seed: int = 1
    xres: int = 64
    yres: int = 64
    spp: int = 4
    max_depth: int = 5
    fov: float = 40.0
    kd: tuple = (0.5, 0.5, 0.5)
    sigma: float = 0.0
    env_L: tuple = (1.0, 1.0, 1.0)
    material: str = "matte"   # matte | plastic | glass | metal | uber: the material of every triangle (headline workload: matte)
    eye: tuple = (0.0, -4.0, 0.0)
    look: tuple = (0.0, 0.0, 0.0)
    up: tuple = (0.0, 0.0, 1.0)
    crop_window: tuple = (0.0, 1.0, 0.0, 1.0)   # Film "cropwindow" x0 x1 y0 y1 (fractions of the full resolution)


def capture_spec(spec: SceneSpec, scene: Scene, host: Host, geometry=None, instances: int = 0, device_build: bool = False):
    """LookAt 0 -4 0  0 0 0  0 0 1 / perspective fov 40 / box filter / halton / matte 0.5 / infinite L=1.
    instances = K > 0: the triangles form ONE object placed K times (ObjectInstance) on a jittered lattice inside the unit
    cube, each copy scaled by K^(-1/3) and rotated — K x n_tris instanced triangles behind a two-level BVH."""
    P, idx = geometry if geometry is not None else host.gen_random_tris(spec.n_tris, spec.seed)
    if spec.material == "matte":
        mat = scene.add_material_matte(spec.kd, spec.sigma)
    elif spec.material == "plastic":
        mat = scene.add_material_plastic(spec.kd, (0.25, 0.25, 0.25), 0.1, True)
    elif spec.material == "glass":
        mat = scene.add_material_glass((1, 1, 1), (1, 1, 1), 0.0, 0.0, 1.5, True)
    elif spec.material == "metal":
        mat = scene.add_material_metal((0.2, 0.92, 1.1), (3.9, 2.45, 2.14), 0.05, 0.05, True)
    elif spec.material == "uber":
        mat = scene.add_material_uber(spec.kd, (0.25, 0.25, 0.25), (0.1, 0.1, 0.1), (0.1, 0.1, 0.1), (0.9, 0.9, 0.9), 0.1, 0.1, 1.5, True)
    elif spec.material == "textured":  # MatteMaterial whose Kd is a 1024 x 1024 image map (EWA, repeat); triangles use the default uv (0,0) (1,0) (1,1)
        img = np.random.default_rng(spec.seed + 7).uniform(0.1, 0.9, (1024, 1024, 3)).astype(np.float32)
        mat = scene.add_material_matte_tex(scene.add_texture_imagemap(scene.add_mipmap(img)), spec.sigma)
    elif spec.material == "mixed":
        mat = None
    else:
        raise ValueError(spec.material)
    if spec.env_L is not None:
        scene.add_light_infinite(spec.env_L)
    if mat is None:  # five materials over five equal slices of the triangle list (spatially interleaved: the triangles are random)
        mats = [scene.add_material_matte(spec.kd, spec.sigma), scene.add_material_plastic(spec.kd, (0.25, 0.25, 0.25), 0.1, True),
                scene.add_material_glass((1, 1, 1), (1, 1, 1), 0.0, 0.0, 1.5, True), scene.add_material_metal((0.2, 0.92, 1.1), (3.9, 2.45, 2.14), 0.05, 0.05, True),
                scene.add_material_uber(spec.kd, (0.25, 0.25, 0.25), (0.1, 0.1, 0.1), (0.1, 0.1, 0.1), (0.9, 0.9, 0.9), 0.1, 0.1, 1.5, True)]
        nt = len(idx) // 3
        for k, m in enumerate(mats):
            t0, t1 = nt * k // 5, nt * (k + 1) // 5
            scene.add_mesh(P[3 * t0:3 * t1], idx[3 * t0:3 * t1] - 3 * t0, m)
    elif instances > 0:
        ob = scene.object_begin(); scene.add_mesh(P, idx, mat); scene.object_end()
        side = int(np.ceil(instances ** (1.0 / 3.0)))
        rng = np.random.default_rng(spec.seed + 1000)
        ident = (IDENTITY, IDENTITY)
        sc = 1.0 / side
        for k in range(instances):
            i, j, l = k % side, (k // side) % side, k // (side * side)
            c = (np.array([i, j, l], np.float64) + 0.5) / side * 2.0 - 1.0 + rng.uniform(-0.2, 0.2, 3) / side
            t = host.compose(host.compose(host.compose(ident, host.translate(c)), host.rotate(float(rng.uniform(0, 360)), rng.normal(size=3) + 1e-3)),
                             host.scale([sc, sc, sc]))
            scene.add_instance(ob, t[0], t[1])
    else:
        scene.add_mesh(P, idx, mat)
    w2c, c2w = host.look_at(spec.eye, spec.look, spec.up)
    r2c = host.perspective_raster_to_camera(spec.fov, spec.xres, spec.yres)
    scene.set_camera_perspective(r2c, c2w)
    cb, table, sb = host.film_box(spec.xres, spec.yres, crop_window=spec.crop_window)
    scene.set_film(spec.xres, spec.yres, cb, (0.5, 0.5), table)
    scene.set_sampler(0, spec.spp, sb)
    if device_build: scene.build_accel_device(0, 4)   # the same tree(s), made on the GPU and left there (csrc/bvh_sah_device.hip)
    else: scene.build_accel(0, 4)
    return P, idx
