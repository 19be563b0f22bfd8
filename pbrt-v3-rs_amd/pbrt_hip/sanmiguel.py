"""Synthetic stand-in for BASELINE.json configs[4] ("San Miguel scene (instanced, ~10M tris, many materials), 1920x1080 @ 512 spp").

The San Miguel asset is not available offline, so this module generates a scene of the same SHAPE, deterministically from a seed, as plain
arrays that can be captured into any binding (the product, or the oracle in tests):

  * a courtyard — tiled floor, four walls, a colonnade — as ordinary top-level triangle meshes (image-textured and bump-mapped);
  * 128 object definitions (ObjectBegin / ObjectEnd, api/src/lib.rs:911-940) of eight kinds — trees with alpha-masked leaf cards, bushes, pots, furniture,
    lamps, statues, cloth, tiles —, each with its own materials, vertex normals and uvs;
  * 1 100 ObjectInstance placements (lib.rs:942-1000 -> TransformedPrimitive, core/src/primitives/transformed_primitive.rs:33-73) with rotations and
    non-uniform scales: ~10 M instanced triangles behind a two-level BVH;
  * 26 materials: every material class of the product, image maps (trilinear and EWA), procedural textures, bump maps, alpha masks, a mix;
  * a radiance-map sky, a distant sun, a point light and eight emissive triangles: eleven lights, so the reference's default SpatialLightDistribution is in play.

`scale` shrinks the tessellation (triangle counts ~ scale) so that CPU-side tests can run the same scene description at a fraction of the size.
"""
import numpy as np

IDENTITY = np.eye(4, dtype=np.float32).reshape(16)


# ---- images ---------------------------------------------------------------------------------------------------------------------------------------------
def _value_noise(w, h, cells, rng):
    """Smooth noise in [0, 1]: bilinear interpolation of a cells x cells lattice (tileable)."""
    g = rng.uniform(0.0, 1.0, (cells, cells)).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    fx, fy = xx * (cells / w), yy * (cells / h)
    x0, y0 = np.floor(fx).astype(int) % cells, np.floor(fy).astype(int) % cells
    x1, y1 = (x0 + 1) % cells, (y0 + 1) % cells
    tx, ty = fx - np.floor(fx), fy - np.floor(fy)
    tx, ty = tx * tx * (3 - 2 * tx), ty * ty * (3 - 2 * ty)
    return ((g[y0, x0] * (1 - tx) + g[y0, x1] * tx) * (1 - ty) + (g[y1, x0] * (1 - tx) + g[y1, x1] * tx) * ty).astype(np.float32)


def make_images(seed):
    """The scene's image files, as (h, w, 3) float32 arrays, top row first."""
    rng = np.random.default_rng(seed)
    im = {}
    n = _value_noise(512, 512, 16, rng) * 0.6 + _value_noise(512, 512, 64, rng) * 0.4
    yy, xx = np.mgrid[0:512, 0:512]
    grout = (((xx % 64) < 3) | ((yy % 64) < 3)).astype(np.float32)
    tile = np.stack([0.55 + 0.3 * n, 0.35 + 0.25 * n, 0.25 + 0.2 * n], axis=2) * (1 - 0.6 * grout[..., None])
    im["tiles"] = tile.astype(np.float32)
    im["tiles_height"] = np.repeat(((1 - grout) * (0.8 + 0.2 * n))[..., None], 3, axis=2).astype(np.float32)
    p = _value_noise(256, 256, 8, rng) * 0.5 + _value_noise(256, 256, 32, rng) * 0.5
    im["plaster"] = np.stack([0.75 + 0.2 * p, 0.68 + 0.2 * p, 0.55 + 0.2 * p], axis=2).astype(np.float32)
    b = _value_noise(256, 256, 4, rng) * 0.3 + _value_noise(256, 256, 64, rng) * 0.7
    im["bark"] = np.stack([0.25 + 0.25 * b, 0.17 + 0.2 * b, 0.1 + 0.12 * b], axis=2).astype(np.float32)
    lf = _value_noise(128, 128, 8, rng)
    im["leaf"] = np.stack([0.1 + 0.2 * lf, 0.35 + 0.4 * lf, 0.05 + 0.15 * lf], axis=2).astype(np.float32)
    yy, xx = np.mgrid[0:64, 0:64].astype(np.float32)
    u, v = (xx + 0.5) / 64 - 0.5, (yy + 0.5) / 64 - 0.5
    shape = ((u / 0.32) ** 2 + (v / 0.47) ** 2 < 1.0) & ~((np.abs(u) < 0.02) & (v > 0.3))
    im["leaf_alpha"] = np.repeat(shape.astype(np.float32)[..., None], 3, axis=2)   # exactly 0 outside the leaf: Triangle::intersect's alpha test is `== 0` (triangle.rs:603)
    f = _value_noise(256, 256, 32, rng)
    im["fabric"] = np.stack([0.6 + 0.3 * f, 0.2 + 0.2 * f, 0.2 + 0.2 * f], axis=2).astype(np.float32)
    m = _value_noise(256, 256, 6, rng)
    im["bronze"] = np.stack([0.45 + 0.3 * m, 0.3 + 0.25 * m, 0.12 + 0.15 * m], axis=2).astype(np.float32)
    # sky: 256 x 128 latitude-longitude radiance map, brighter towards the horizon, a warm patch around the sun's direction
    yy, xx = np.mgrid[0:128, 0:256].astype(np.float32)
    theta = (yy + 0.5) / 128 * np.pi
    up = np.clip(np.cos(theta), 0, 1)
    sky = np.stack([0.35 + 0.5 * (1 - up), 0.5 + 0.4 * (1 - up), 0.9 - 0.1 * (1 - up)], axis=2)
    sky[theta[:, 0] > np.pi / 2 + 0.05] = (0.12, 0.1, 0.08)   # below the horizon: ground bounce
    glow = np.exp(-(((xx - 70) / 18.0) ** 2 + ((yy - 30) / 12.0) ** 2))
    sky = sky + glow[..., None] * np.array([3.0, 2.4, 1.6], np.float32)
    im["sky"] = sky.astype(np.float32)
    return im


# ---- meshes ---------------------------------------------------------------------------------------------------------------------------------------------
def _grid(nu, nv, fn, uv_rep=(1.0, 1.0), flip=False):
    """A parametric surface fn(u, v) -> (x, y, z) on [0, 1]^2 tessellated nu x nv: shared vertices, finite-difference vertex normals, uvs."""
    nu, nv = max(int(nu), 2), max(int(nv), 2)
    u, v = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="xy")
    P = np.stack(fn(u, v), axis=2).astype(np.float64)
    e = 1e-3
    du = (np.stack(fn(np.clip(u + e, 0, 1), v), axis=2) - np.stack(fn(np.clip(u - e, 0, 1), v), axis=2))
    dv = (np.stack(fn(u, np.clip(v + e, 0, 1)), axis=2) - np.stack(fn(u, np.clip(v - e, 0, 1)), axis=2))
    N = np.cross(du, dv)
    ln = np.linalg.norm(N, axis=2, keepdims=True)
    N = np.where(ln > 1e-12, N / np.maximum(ln, 1e-12), np.array([0.0, 0.0, 1.0]))
    if flip:
        N = -N
    UV = np.stack([u * uv_rep[0], v * uv_rep[1]], axis=2)
    i = np.arange(nv)[:, None] * (nu + 1) + np.arange(nu)[None, :]
    a, b, c, d = i, i + 1, i + nu + 2, i + nu + 1
    idx = np.stack([a, b, c, a, c, d] if not flip else [a, c, b, a, d, c], axis=2).reshape(-1)
    return (P.reshape(-1, 3).astype(np.float32), idx.astype(np.uint32), N.reshape(-1, 3).astype(np.float32), UV.reshape(-1, 2).astype(np.float32))


def _blob(nu, nv, rng, radius=(0.5, 0.5, 0.5), lumps=0.15, z0=0.0):
    k = rng.integers(2, 6, 3); ph = rng.uniform(0, 6.28, 3)

    def fn(u, v):
        th, phi = v * np.pi, u * 2 * np.pi
        r = 1.0 + lumps * (np.sin(k[0] * phi + ph[0]) * np.sin(k[1] * th + ph[1]) + 0.5 * np.sin(k[2] * (phi + th) + ph[2]))
        return (radius[0] * r * np.sin(th) * np.cos(phi), radius[1] * r * np.sin(th) * np.sin(phi), z0 + radius[2] * r * np.cos(th))
    return _grid(nu, nv, fn, (4.0, 2.0), flip=True)   # (d/du x d/dv of this parameterisation points inwards)


def _cylinder(nu, nv, r0, r1, z0, z1, uv_rep=(3.0, 3.0), wobble=0.0, rng=None):
    ph = rng.uniform(0, 6.28) if rng is not None else 0.0

    def fn(u, v):
        r = r0 + (r1 - r0) * v + wobble * np.sin(7 * v + ph) * r0
        return (r * np.cos(2 * np.pi * u), r * np.sin(2 * np.pi * u), z0 + (z1 - z0) * v)
    return _grid(nu, nv, fn, uv_rep)


def _cards(n, rng, centre, spread, size):
    """n leaf cards: small quads at random places of an ellipsoid, random orientation, uv = the unit square."""
    n = max(int(n), 1)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    c = np.asarray(centre) + d * rng.uniform(0.3, 1.0, (n, 1)) ** (1 / 3) * np.asarray(spread)
    a = rng.normal(size=(n, 3)); a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = np.cross(a, rng.normal(size=(n, 3))); b /= np.linalg.norm(b, axis=1, keepdims=True)
    s = size * rng.uniform(0.6, 1.4, (n, 1))
    P = np.stack([c - a * s - b * s, c + a * s - b * s, c + a * s + b * s, c - a * s + b * s], axis=1).reshape(-1, 3)
    nrm = np.cross(a, b)
    N = np.repeat(nrm[:, None, :], 4, axis=1).reshape(-1, 3)
    UV = np.tile(np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32), (n, 1))
    q = (np.arange(n) * 4)[:, None]
    idx = (q + np.array([0, 1, 2, 0, 2, 3])[None, :]).reshape(-1)
    return P.astype(np.float32), idx.astype(np.uint32), N.astype(np.float32), UV.astype(np.float32)


def _box(lo, hi, n):
    """A box as six n x n grids (flat normals per face, uvs per face)."""
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    parts = []
    for ax in range(3):
        for side in (0, 1):
            a, b = (ax + 1) % 3, (ax + 2) % 3

            def fn(u, v, ax=ax, a=a, b=b, side=side):
                out = [None, None, None]
                out[ax] = np.full_like(u, hi[ax] if side else lo[ax])
                out[a] = lo[a] + (hi[a] - lo[a]) * (u if side else v)
                out[b] = lo[b] + (hi[b] - lo[b]) * (v if side else u)
                return tuple(out)
            parts.append(_grid(n, n, fn))
    return _merge(parts)


def _merge(parts):
    P, I, N, UV = [], [], [], []
    base = 0
    for p, i, n, uv in parts:
        P.append(p); I.append(i + np.uint32(base)); N.append(n); UV.append(uv); base += len(p)
    return np.concatenate(P), np.concatenate(I).astype(np.uint32), np.concatenate(N), np.concatenate(UV)


def _xf(mesh, scale=(1, 1, 1), shift=(0, 0, 0)):
    P, I, N, UV = mesh
    s = np.asarray(scale, np.float32)
    Nn = N / s
    Nn = Nn / np.maximum(np.linalg.norm(Nn, axis=1, keepdims=True), 1e-20)
    return (P * s + np.asarray(shift, np.float32)).astype(np.float32), I, Nn.astype(np.float32), UV


# ---- the scene description --------------------------------------------------------------------------------------------------------------------------------
class SanMiguelScene:
    """Plain-data description; capture() replays it through the C ABI mirror (product Scene or, in tests, the oracle's)."""

    N_OBJECTS = 128
    N_INSTANCES = 1100

    def __init__(self, host, scale=1.0, seed=5, n_objects=None, n_instances=None):
        self.host = host
        self.scale = float(scale)
        self.seed = seed
        rng = np.random.default_rng(seed)
        s = max(self.scale, 1e-3) ** 0.5     # grid resolutions scale with sqrt(scale): triangle counts with scale
        self.images = make_images(seed + 1)
        n_objects = n_objects or self.N_OBJECTS
        n_instances = n_instances or self.N_INSTANCES

        def g(n):
            return max(2, int(round(n * s)))
        # ---- object definitions: (list of (mesh, material name, alpha texture name or None)) ----
        self.objects = []
        kinds = ["tree", "bush", "pot", "chair", "lamp", "statue", "cloth", "tile"]
        for k in range(n_objects):
            kind = kinds[k % len(kinds)]
            r = np.random.default_rng(seed * 1000 + k)
            variant = (k // len(kinds)) % 3
            if kind == "tree":
                trunk = _cylinder(g(24), g(40), 0.06, 0.03, 0.0, 0.55, wobble=0.15, rng=r)
                leaves = _cards(6000 * self.scale, r, (0, 0, 0.75), (0.42, 0.42, 0.32), 0.035)
                parts = [(trunk, "bark", None), (leaves, ["leaf_translucent", "leaf_matte", "leaf_uber"][variant], "leaf_alpha")]
            elif kind == "bush":
                parts = [(_cards(4000 * self.scale, r, (0, 0, 0.22), (0.35, 0.35, 0.2), 0.03), ["leaf_matte", "leaf_uber", "leaf_translucent"][variant], "leaf_alpha")]
            elif kind == "pot":
                parts = [(_xf(_blob(g(64), g(32), r, (0.3, 0.3, 0.3), 0.05), shift=(0, 0, 0.3)), ["terracotta", "glazed", "checker_bumped"][variant], None)]
            elif kind == "chair":
                legs = [_box((x - 0.03, y - 0.03, 0.0), (x + 0.03, y + 0.03, 0.4), g(6)) for x in (-0.25, 0.25) for y in (-0.25, 0.25)]
                seat = _box((-0.3, -0.3, 0.4), (0.3, 0.3, 0.46), g(10)); back = _box((-0.3, 0.25, 0.46), (0.3, 0.3, 0.95), g(10))
                parts = [(_merge(legs), ["steel", "copper", "gold"][variant], None), (_merge([seat, back]), ["plastic_red", "plastic_blue", "white"][variant], None)]
            elif kind == "lamp":
                parts = [(_cylinder(g(16), g(24), 0.03, 0.02, 0.0, 0.8), "steel", None),
                         (_xf(_blob(g(48), g(24), r, (0.16, 0.16, 0.2), 0.02), shift=(0, 0, 0.95)), ["glass", "frosted", "water"][variant], None)]
            elif kind == "statue":
                parts = [(_xf(_blob(g(160), g(96), r, (0.22, 0.2, 0.5), 0.3), shift=(0, 0, 0.62)), ["bronze_uber", "uber", "mirror"][variant], None),
                         (_box((-0.3, -0.3, 0.0), (0.3, 0.3, 0.12), g(8)), "marble", None)]
            elif kind == "cloth":
                ph = r.uniform(0, 6.28, 2)

                def fn(u, v, ph=ph):
                    return ((u - 0.5) * 0.9, 0.08 * np.sin(9 * u + ph[0]) * (1 - v) + 0.02 * np.sin(23 * u + ph[1]), 0.2 + v * 0.9)
                parts = [(_grid(g(72), g(56), fn, (2.0, 2.0)), ["cloth", "fabric", "cloth"][variant], None)]
            else:
                parts = [(_box((-0.45, -0.45, 0.0), (0.45, 0.45, 0.05), g(14)), ["tile_mix", "checker_bumped", "marble"][variant], None)]
            self.objects.append(parts)
        # ---- instances: jittered lattice over the courtyard, rotation about z with a small tilt, non-uniform scale ----
        side = int(np.ceil(np.sqrt(n_instances)))
        self.instances = []
        for k in range(n_instances):
            i, j = k % side, k // side
            c = np.array([(i + 0.5) / side * 1.8 - 0.9 + rng.uniform(-0.3, 0.3) / side, (j + 0.5) / side * 1.8 - 0.9 + rng.uniform(-0.3, 0.3) / side, 0.0])
            ob = int(rng.integers(0, n_objects))
            sc = float(rng.uniform(0.035, 0.06)) * (1.0 + (ob % 8 == 0) * 1.2)   # trees are taller
            t = host.compose(host.compose(host.compose((IDENTITY, IDENTITY), host.translate(c)), host.rotate(float(rng.uniform(0, 360)), np.array([0, 0, 1.0]) + rng.normal(size=3) * 0.03)),
                             host.scale([sc, sc * float(rng.uniform(0.85, 1.15)), sc * float(rng.uniform(0.9, 1.3))]))
            self.instances.append((ob, t))
        # ---- architecture (top-level meshes) ----
        self.top = []
        self.top.append((_grid(g(400), g(400), lambda u, v: (u * 2.4 - 1.2, v * 2.4 - 1.2, 0 * u), (12.0, 12.0)), "floor", None))
        for w in range(4):
            ang = w * np.pi / 2

            def wall(u, v, ang=ang):
                x, y = (u * 2.4 - 1.2), 1.2 + 0 * u
                return (x * np.cos(ang) - y * np.sin(ang), x * np.sin(ang) + y * np.cos(ang), v * 0.7)
            self.top.append((_grid(g(100), g(30), wall, (6.0, 2.0), flip=True), "wall", None))
        cols = []
        for q in range(24):
            a = q / 24 * 2 * np.pi
            cols.append(_xf(_cylinder(g(48), g(32), 0.035, 0.03, 0.0, 0.55), shift=(1.05 * np.cos(a), 1.05 * np.sin(a), 0.0)))
        self.top.append((_merge(cols), "marble", None))
        # lanterns: emissive quads (two triangles each), facing down
        self.lanterns = []
        for q in range(4):
            x, y = 0.55 * np.cos(q * np.pi / 2 + 0.6), 0.55 * np.sin(q * np.pi / 2 + 0.6)
            P = np.array([[x - 0.05, y - 0.05, 0.6], [x + 0.05, y - 0.05, 0.6], [x + 0.05, y + 0.05, 0.6], [x - 0.05, y + 0.05, 0.6]], np.float32)
            self.lanterns.append((P, np.array([0, 2, 1, 0, 3, 2], np.uint32)))

    # ------------------------------------------------------------------------------------------------------------------------------------------------------
    def counts(self):
        obj_tris = [sum(len(m[1]) // 3 for m, _, _ in parts) for parts in self.objects]
        inst = sum(obj_tris[ob] for ob, _ in self.instances)
        top = sum(len(m[1]) // 3 for m, _, _ in self.top) + 2 * len(self.lanterns)
        return {"objects": len(self.objects), "instances": len(self.instances), "unique_object_triangles": int(sum(obj_tris)), "top_level_triangles": int(top),
                "instanced_triangles": int(inst), "total_triangles_as_instanced": int(inst + top)}

    def _materials(self, s):
        """name -> material id; 26 materials."""
        im = self.images
        T = {}
        T["tiles"] = s.add_texture_imagemap(s.add_mipmap(im["tiles"], wrap="repeat"))
        T["tiles_h"] = s.add_texture_scale(s.add_texture_imagemap(s.add_mipmap(im["tiles_height"], as_float=True, trilinear=True)), s.add_texture_constant(0.004))
        T["plaster"] = s.add_texture_imagemap(s.add_mipmap(im["plaster"], trilinear=True))
        T["bark"] = s.add_texture_imagemap(s.add_mipmap(im["bark"]), su=1.0, sv=2.0)
        T["leaf"] = s.add_texture_imagemap(s.add_mipmap(im["leaf"]))
        T["fabric"] = s.add_texture_imagemap(s.add_mipmap(im["fabric"], gamma=True), su=3.0, sv=3.0)
        T["bronze"] = s.add_texture_imagemap(s.add_mipmap(im["bronze"]))
        T["leaf_alpha"] = s.add_texture_imagemap(s.add_mipmap(im["leaf_alpha"], as_float=True, trilinear=True, wrap="clamp"))
        import os
        if os.environ.get("PBRT_SM_SIMPLE_TEXTURES"):   # measurement aid: the procedural textures replaced by image maps / constants (how much of the texture pass is theirs?)
            T["fbm_h"] = T["wrinkled_h"] = T["tiles_h"]; T["marble"] = T["plaster"]; T["checker"] = T["tiles"]; T["dots"] = T["fabric"]
            return self._finish_materials(s, T)
        sc = np.diag([18.0, 18.0, 18.0, 1.0]).astype(np.float32).reshape(16)
        T["fbm_h"] = s.add_texture_scale(s.add_texture_fbm(sc, 0.5, 4), s.add_texture_constant(0.003))
        T["wrinkled_h"] = s.add_texture_scale(s.add_texture_fbm(sc, 0.6, 5, wrinkled=True), s.add_texture_constant(0.004))
        # (the reference's MarbleTexture extrapolates its spline — `min(1, ..)` where pbrt-v3 has `min(NSEGS - 3, ..)`, textures/src/marble.rs:65 — and returns albedos up to ~40:
        #  not used here; the stone is a 3D checkerboard of two greys modulated by windy waves)
        stone = s.add_texture_checkerboard3d(s.add_texture_constant((0.62, 0.6, 0.58)), s.add_texture_constant((0.5, 0.5, 0.52)), np.diag([9.0, 9.0, 9.0, 1.0]).astype(np.float32).reshape(16))
        T["marble"] = s.add_texture_mix(stone, s.add_texture_constant((0.7, 0.68, 0.62)), s.add_texture_scale(s.add_texture_windy(np.diag([3.0, 3.0, 3.0, 1.0]).astype(np.float32).reshape(16)), s.add_texture_constant(0.5)))
        T["checker"] = s.add_texture_checkerboard(s.add_texture_constant((0.8, 0.8, 0.75)), s.add_texture_constant((0.15, 0.2, 0.3)), su=6.0, sv=6.0)
        T["dots"] = s.add_texture_dots(s.add_texture_constant((0.9, 0.3, 0.2)), s.add_texture_constant((0.9, 0.85, 0.7)), su=5.0, sv=5.0)
        return self._finish_materials(s, T)

    def _finish_materials(self, s, T):
        M = {}
        M["floor"] = s.add_material_matte_tex(T["tiles"], 0.0); s.set_material_bump(M["floor"], T["tiles_h"])
        M["wall"] = s.add_material_matte_tex(T["plaster"], 20.0); s.set_material_bump(M["wall"], T["fbm_h"])
        M["marble"] = s.add_material_plastic((0.5, 0.5, 0.5), (0.2, 0.2, 0.2), 0.05, True); s.set_material_texture(M["marble"], "Kd", T["marble"])
        M["bark"] = s.add_material_matte_tex(T["bark"], 0.0); s.set_material_bump(M["bark"], T["wrinkled_h"])
        M["leaf_translucent"] = s.add_material_translucent((0.3, 0.5, 0.2), (0.1, 0.1, 0.1), (0.6, 0.6, 0.6), (0.4, 0.4, 0.4), 0.3, True)
        s.set_material_texture(M["leaf_translucent"], "Kd", T["leaf"])
        M["leaf_matte"] = s.add_material_matte_tex(T["leaf"], 0.0)
        M["leaf_uber"] = s.add_material_uber((0.3, 0.5, 0.2), (0.1, 0.12, 0.08), (0, 0, 0), (0.15, 0.25, 0.1), (1, 1, 1), 0.2, 0.2, 1.5, True)
        s.set_material_texture(M["leaf_uber"], "Kd", T["leaf"])
        M["terracotta"] = s.add_material_matte((0.6, 0.3, 0.2), 30.0)
        M["glazed"] = s.add_material_substrate((0.2, 0.3, 0.5), (0.3, 0.3, 0.3), 0.05, 0.08, True)
        M["checker_bumped"] = s.add_material_matte_tex(T["checker"], 0.0); s.set_material_bump(M["checker_bumped"], T["fbm_h"])
        M["glass"] = s.add_material_glass((1, 1, 1), (1, 1, 1), 0.0, 0.0, 1.5, True)
        M["frosted"] = s.add_material_glass((0.9, 0.9, 0.9), (0.9, 0.95, 0.9), 0.1, 0.1, 1.5, True)
        M["water"] = s.add_material_glass((1, 1, 1), (0.9, 0.95, 1.0), 0.0, 0.0, 1.33, True)
        M["copper"] = s.add_material_metal((0.2, 0.92, 1.1), (3.9, 2.45, 2.14), 0.05, 0.05, True)
        M["steel"] = s.add_material_metal((2.5, 2.4, 2.3), (3.3, 3.2, 3.0), 0.1, 0.1, True)
        M["gold"] = s.add_material_metal((0.14, 0.37, 1.44), (3.98, 2.38, 1.6), 0.02, 0.06, True)
        M["plastic_red"] = s.add_material_plastic((0.6, 0.08, 0.06), (0.3, 0.3, 0.3), 0.08, True)
        M["plastic_blue"] = s.add_material_plastic((0.1, 0.15, 0.6), (0.3, 0.3, 0.3), 0.15, True); s.set_material_texture(M["plastic_blue"], "Ks", T["dots"])
        M["white"] = s.add_material_matte((0.8, 0.8, 0.8), 0.0)
        M["mirror"] = s.add_material_mirror((0.9, 0.9, 0.9))
        M["cloth"] = s.add_material_translucent((0.5, 0.45, 0.4), (0.05, 0.05, 0.05), (0.7, 0.7, 0.7), (0.3, 0.3, 0.3), 0.4, True)
        M["fabric"] = s.add_material_matte_tex(T["fabric"], 40.0)
        M["uber"] = s.add_material_uber((0.3, 0.3, 0.35), (0.25, 0.25, 0.25), (0.1, 0.1, 0.1), (0, 0, 0), (1, 1, 1), 0.1, 0.15, 1.5, True)
        M["bronze_uber"] = s.add_material_uber((0.4, 0.3, 0.15), (0.4, 0.35, 0.2), (0.05, 0.05, 0.05), (0, 0, 0), (1, 1, 1), 0.08, 0.08, 1.5, True)
        s.set_material_texture(M["bronze_uber"], "Kd", T["bronze"])
        tile_a = s.add_material_matte_tex(T["checker"], 0.0)
        tile_b = s.add_material_plastic((0.3, 0.3, 0.3), (0.4, 0.4, 0.4), 0.03, True)
        M["tile_mix"] = s.add_material_mix(tile_a, tile_b, (0.35, 0.35, 0.35))
        M["lantern"] = s.add_material_matte((0.1, 0.1, 0.1), 0.0)
        self.n_materials = len(M) + 2
        return M, T

    def capture(self, s, xres=1920, yres=1080, spp=512, crop=(0.0, 1.0, 0.0, 1.0), device_build=False, split_method=0):
        """Replays the description into `s` (any pbrt_hip.Scene-shaped binding) and builds the accelerator."""
        host = self.host
        M, T = self._materials(s)
        # lights, in Scene::lights order: sky, sun, lamp, then the lanterns' triangles
        l2w = host.compose((IDENTITY, IDENTITY), host.rotate(-90.0, [1, 0, 0]))   # the map's poles along world z
        s.add_light_infinite_map((1.0, 1.0, 1.0), self.images["sky"], l2w[0], l2w[1])
        s.add_light_distant((2.5, 2.3, 1.9), (0.35, -0.45, 0.82))
        s.add_light_point((0.25, 0.22, 0.18), (0.0, 0.0, 0.5))
        for parts in self.objects:
            s.object_begin()
            for (P, I, N, UV), mat, alpha in parts:
                s.add_mesh(P, I, M[mat], N=N, UV=UV)
                if alpha:
                    s.set_last_mesh_alpha_textures(alpha=T[alpha], shadow_alpha=T[alpha])
            s.object_end()
        for (P, I, N, UV), mat, alpha in self.top:
            s.add_mesh(P, I, M[mat], N=N, UV=UV)
        for P, I in self.lanterns:
            lid = s.add_light_diffuse_area((14.0, 11.0, 7.0), 2)
            s.add_mesh(P, I, M["lantern"], first_area_light=lid)
        for ob, t in self.instances:
            s.add_instance(ob, t[0], t[1])
        w2c, c2w = host.look_at((0.95, -1.1, 0.3), (-0.1, 0.1, 0.12), (0, 0, 1))
        s.set_camera_perspective(host.perspective_raster_to_camera(55.0, xres, yres), c2w)
        cb, table, sb = host.film_box(xres, yres, crop_window=crop)
        s.set_film(xres, yres, cb, (0.5, 0.5), table)
        s.set_sampler(0, spp, sb)
        if device_build:
            s.build_accel_device(split_method, 4)
        else:
            s.build_accel(split_method, 4)
