"""An emissive quad INSIDE an object definition (api/src/lib.rs:877-881: "Area lights not supported with object instancing"), shared by the oracle pin and the GPU parity test."""
import numpy as np

I4 = (np.eye(4, dtype=np.float32).reshape(16),) * 2
L_EMIT = (9.0, 7.0, 4.0)


def emissive_object_scene(s, host, emissive=True, res=40, spp=4, sun=(0.3, 0.3, 0.3)):
    """A black emissive quad (two triangles) in an object instanced twice — once rotated —, a matte floor, a mirror wall behind, one weak distant light."""
    black = s.add_material_matte((0.0, 0.0, 0.0), 0.0)
    grey = s.add_material_matte((0.6, 0.6, 0.6), 0.0)
    mirror = s.add_material_mirror((0.9, 0.9, 0.9))
    s.add_light_distant(sun, (0.2, -0.3, 0.93))
    Pq = np.array([[-0.4, 0, -0.4], [0.4, 0, -0.4], [0.4, 0, 0.4], [-0.4, 0, 0.4]], np.float32)   # in the xz plane, facing -y (towards the camera)
    ob = s.object_begin()
    lid = s.add_light_diffuse_area(L_EMIT, 2) if emissive else -1
    s.add_mesh(Pq, np.array([0, 1, 2, 0, 2, 3], np.uint32), black, first_area_light=lid)
    s.object_end()
    Pf = np.array([[-4, -4, -1], [4, -4, -1], [4, 4, -1], [-4, 4, -1]], np.float32)
    s.add_mesh(Pf, np.array([0, 1, 2, 0, 2, 3], np.uint32), grey)
    Pm = np.array([[-4, 3, -1], [4, 3, -1], [4, 3, 3], [-4, 3, 3]], np.float32)                    # mirror wall at y = 3 facing the camera
    s.add_mesh(Pm, np.array([0, 2, 1, 0, 3, 2], np.uint32), mirror)
    mul = host.compose
    s.add_instance(ob, *mul(I4, host.translate([-0.9, 0.0, 0.2])))
    s.add_instance(ob, *mul(mul(I4, host.translate([0.9, 0.5, 0.3])), host.rotate(35.0, [0, 0, 1])))
    w2c, c2w = host.look_at([0.0, -5.0, 0.8], [0, 0, 0.2], [0, 0, 1])
    s.set_camera_perspective(host.perspective_raster_to_camera(45.0, res, res), c2w)
    cb, table, sb = host.film_box(res, res)
    s.set_film(res, res, cb, (0.5, 0.5), table)
    s.set_sampler(0, spp, sb)
    s.build_accel(0, 4)
