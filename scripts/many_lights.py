import sys, time
sys.path.insert(0, 'pbrt-v3-rs_amd'); sys.path.insert(0, 'tests')
import numpy as np, pbrt_hip, scenes
host = pbrt_hip.Host()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
with pbrt_hip.Scene() as s:
    white = s.add_material_matte((0.7, 0.7, 0.7), 0.0)
    P, idx = scenes.grid_mesh(8, z=-1.0, size=2.0); s.add_mesh(P, idx, white)
    P, idx = host.gen_random_tris(20000, 3); s.add_mesh(P * np.float32(0.8), idx, white)
    P, idx = scenes.grid_mesh(n, z=1.5, size=1.5)
    lid = s.add_light_diffuse_area((6.0, 5.0, 4.0), len(idx) // 3)
    s.add_mesh(P, idx, white, first_area_light=lid, reverse_orientation=True)
    s.add_light_infinite((0.1, 0.1, 0.1))
    w2c, c2w = host.look_at([0.5, -4.5, 1.0], [0, 0, 0], [0, 0, 1])
    res = 256
    s.set_camera_perspective(host.perspective_raster_to_camera(45.0, res, res), c2w)
    cb, table, sb = host.film_box(res, res)
    s.set_film(res, res, cb, (0.5, 0.5), table); s.set_sampler(0, 16, sb); s.build_accel(0, 4)
    for strat in (1, 2, 2):
        t = time.time(); xyz, wt, st = s.render_path(max_depth=5, light_strategy=strat); dt = time.time() - t
        d = st.as_dict()
        print('lights', 2 * n * n + 1, 'strategy', strat, 'wall %.3f s' % dt, 'render %.3f' % d['render_seconds'], 'created', d['light_distributions_created'],
              'Mrays/s %.1f' % ((d['regular_rays'] + d['shadow_rays']) / d['render_seconds'] / 1e6), 'mean', float(s.film_to_rgb(xyz, wt).mean()))
