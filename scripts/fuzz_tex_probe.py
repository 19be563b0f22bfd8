"""Bisect a failing seed of tests/test_fuzz_textures_gpu.py by switching features off on both sides."""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "pbrt-v3-rs_amd")); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, pbrt_hip
import test_fuzz_textures_gpu as T
host = pbrt_hip.Host()
seed = int(sys.argv[1])
print("full", T.run_case(host, seed))
orig_alpha = pbrt_hip.Scene.set_last_mesh_alpha_textures
orig_bump = pbrt_hip.Scene.set_material_bump
orig_inst = pbrt_hip.Scene.add_instance
pbrt_hip.Scene.set_last_mesh_alpha_textures = lambda self, a=None, b=None: None
print("no alpha", T.run_case(host, seed))
pbrt_hip.Scene.set_last_mesh_alpha_textures = orig_alpha
pbrt_hip.Scene.set_material_bump = lambda self, m, t: None
print("no bump", T.run_case(host, seed))
pbrt_hip.Scene.set_material_bump = orig_bump
pbrt_hip.Scene.set_last_mesh_alpha_textures = lambda self, a=None, b=None: orig_alpha(self, a, None)
print("alpha only (no shadowalpha)", T.run_case(host, seed))
pbrt_hip.Scene.set_last_mesh_alpha_textures = lambda self, a=None, b=None: orig_alpha(self, None, b)
print("shadowalpha only", T.run_case(host, seed))
pbrt_hip.Scene.set_last_mesh_alpha_textures = orig_alpha
pbrt_hip.Scene.add_instance = lambda self, *a: None
print("no instances", T.run_case(host, seed))
pbrt_hip.Scene.add_instance = orig_inst
orig_sky = pbrt_hip.Scene.add_light_infinite_map
pbrt_hip.Scene.add_light_infinite_map = lambda self, L, img, a=None, b=None: pbrt_hip.Scene.add_light_infinite(self, L, a, b)
print("constant sky", T.run_case(host, seed))
pbrt_hip.Scene.add_light_infinite_map = orig_sky
orig_smt = pbrt_hip.Scene.set_material_texture
for skip in ("Kd", "Ks", "Kr", "Kt"):
    pbrt_hip.Scene.set_material_texture = lambda self, m, p, t, skip=skip: None if p == skip else orig_smt(self, m, p, t)
    print("no", skip, T.run_case(host, seed))
pbrt_hip.Scene.set_material_texture = orig_smt
orig_mesh = pbrt_hip.Scene.add_mesh
def no_s(self, P, idx, mat, **kw):
    kw.pop("S", None); return orig_mesh(self, P, idx, mat, **kw)
pbrt_hip.Scene.add_mesh = no_s
print("no S", T.run_case(host, seed))
pbrt_hip.Scene.add_mesh = orig_mesh
pbrt_hip.Scene.set_material_texture = lambda self, m, p, t: None
pbrt_hip.Scene.set_material_bump = lambda self, m, t: None
pbrt_hip.Scene.set_last_mesh_alpha_textures = lambda self, a=None, b=None: None
pbrt_hip.Scene.add_light_infinite_map = lambda self, L, img, a=None, b=None: pbrt_hip.Scene.add_light_infinite(self, L, a, b)
orig_mtex = pbrt_hip.Scene.add_material_matte_tex
pbrt_hip.Scene.add_material_matte_tex = lambda self, t, sg=0.0: pbrt_hip.Scene.add_material_matte(self, (0.5, 0.5, 0.5), sg)
print("nothing textured (plain kernels)", T.run_case(host, seed))
print("---- plain kernels, substituting materials")
S = pbrt_hip.Scene
o_uber, o_mirror, o_sub = S.add_material_uber, S.add_material_mirror, S.add_material_substrate
matte = lambda self, *a, **k: S.add_material_matte(self, (0.5, 0.5, 0.5), 0.0)
S.add_material_uber = matte; print("uber->matte", T.run_case(host, seed)); S.add_material_uber = o_uber
S.add_material_mirror = matte; print("mirror->matte", T.run_case(host, seed)); S.add_material_mirror = o_mirror
S.add_material_substrate = matte; print("substrate->matte", T.run_case(host, seed)); S.add_material_substrate = o_sub
