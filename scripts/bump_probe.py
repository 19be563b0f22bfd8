import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pbrt-v3-rs_amd"))
import numpy as np, pbrt_hip
from oracle_binding import OracleScene, set_libm_mode
from texture_scenes import make_image, textured_quad_scene
host = pbrt_hip.Host()
def run(name, bump, depth=4, **kw):
    def material(sc, tex):
        m = sc.add_material_matte((0.6, 0.6, 0.6), 0.0); sc.set_material_bump(m, bump(sc)); return m
    prod = pbrt_hip.Scene(); orc = OracleScene()
    for s in (prod, orc): textured_quad_scene(s, host, lambda sc: sc.add_texture_constant((0.5, 0.5, 0.5)), res=48, material=material, **kw)
    set_libm_mode(1); o = orc.render_path_ex(max_depth=depth); set_libm_mode(0)
    g = prod.render_path(max_depth=depth)
    d = (g[0].view(np.uint32) != o[0].view(np.uint32)).any(axis=2)
    print(name, "depth", depth, "differing pixels", int(d.sum()), "max abs", float(np.abs(g[0] - o[0]).max()), "rows", np.where(d.any(axis=1))[0][:10])
img = lambda **k: (lambda sc: sc.add_texture_imagemap(sc.add_mipmap(make_image(32, 32, seed=21), as_float=True, **k), su=2.0, sv=2.0))
run("const", lambda sc: sc.add_texture_constant(0.3))
run("dots", lambda sc: sc.add_texture_dots(sc.add_texture_constant(0.02), sc.add_texture_constant(0.0), su=6.0, sv=6.0))
run("img ewa", img())
run("img ewa depth1", img(), depth=1)
run("img tri", img(trilinear=True))
run("img tri depth1", img(trilinear=True), depth=1)
print("---- forced general kernel")
def run2(name, bump, depth=1, gen=False, spp=4):
    def material(sc, tex):
        m = sc.add_material_matte((0.6, 0.6, 0.6), 0.0); sc.set_material_bump(m, bump(sc)); return m
    def extra(sc):
        if gen:
            gl = sc.add_material_glass((1, 1, 1), (1, 1, 1), 0.0, 0.0, 1.5, True)
            sc.add_mesh(np.array([[30, 30, 30], [31, 30, 30], [30, 31, 30]], np.float32), [0, 1, 2], gl)
    prod = pbrt_hip.Scene(); orc = OracleScene()
    for s in (prod, orc): textured_quad_scene(s, host, lambda sc: sc.add_texture_constant((0.5, 0.5, 0.5)), res=48, spp=spp, material=material, extra=extra)
    set_libm_mode(1); o = orc.render_path_ex(max_depth=depth); set_libm_mode(0)
    g = prod.render_path(max_depth=depth)
    d = (g[0].view(np.uint32) != o[0].view(np.uint32)).any(axis=2)
    ys, xs = np.where(d)
    print(name, "gen", gen, "differing", int(d.sum()), "cols", sorted(set(xs.tolist()))[:12], "first", (ys[0], xs[0], g[0][ys[0], xs[0]], o[0][ys[0], xs[0]]) if len(ys) else None)
run2("const", lambda sc: sc.add_texture_constant(0.3))
run2("const", lambda sc: sc.add_texture_constant(0.3), gen=True)
run2("const spp1", lambda sc: sc.add_texture_constant(0.3), spp=1)
