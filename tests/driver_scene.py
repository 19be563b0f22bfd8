"""One scene written twice: as .pbrt text for the C++ host (pbrt_hip_render) and as direct C-ABI calls through
pbrt_hip.Scene (product or oracle).  The text exercises the CTM stack, attribute stack, named materials, constant
textures, Include, a binary PLY with a quad face, all four light kinds and a non-box filter."""
import os
import struct

import numpy as np

RENDER_BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pbrt-v3-rs_amd", "pbrt_hip_render")


def f(x):
    """shortest decimal that round-trips the f32"""
    return repr(float(np.float32(x)))


def fl(a):
    return " ".join(f(v) for v in np.asarray(a, np.float32).ravel())


XRES, YRES, SPP, MAXDEPTH = 56, 40, 4, 4
QUAD_IDX = [0, 1, 2, 0, 2, 3]
FLOOR = np.array([[-2, -2, 0], [2, -2, 0], [2, 2, 0], [-2, 2, 0]], np.float32)
LAMP = np.array([[-0.5, -0.5, 0], [0.5, -0.5, 0], [0.5, 0.5, 0], [-0.5, 0.5, 0]], np.float32)
PLY_P = np.array([[-0.5, -0.5, 0.5], [0.5, -0.5, 0.5], [0.5, 0.5, 1.0], [-0.5, 0.5, 1.0], [0.0, 0.0, 1.6]], np.float32)
PLY_N = np.array([[0, 0, 1], [0.1, 0, 1], [0, 0.1, 1], [-0.1, 0, 1], [0, 0, 1]], np.float32)
PLY_FACES = [[0, 1, 2, 3], [2, 3, 4]]
TENT_P = np.array([[-1.5, 0.5, 0], [-0.8, 0.5, 0], [-1.2, 1.3, 0.9]], np.float32)
TENT_N = np.array([[0, -0.7, 0.7], [0, -0.7, 0.7], [0.1, -0.7, 0.6]], np.float32)
TENT_UV = np.array([[0, 0], [1, 0], [0.5, 1]], np.float32)


def write_files(d, filter_line='PixelFilter "gaussian" "float xwidth" 1.5 "float ywidth" 1.25 "float alpha" 1.5',
                strategy="power", crop=None):
    """Writes scene.pbrt (+ inc.pbrt, mesh.ply) into directory d; returns the scene path."""
    with open(os.path.join(d, "mesh.ply"), "wb") as fh:  # binary little-endian, normals, one quad + one triangle
        fh.write(b"ply\nformat binary_little_endian 1.0\ncomment test mesh\nelement vertex 5\n"
                 b"property float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\n"
                 b"element face 2\nproperty list uchar int vertex_indices\nend_header\n")
        for p, n in zip(PLY_P, PLY_N):
            fh.write(struct.pack("<6f", *p, *n))
        for face in PLY_FACES:
            fh.write(struct.pack("<B%di" % len(face), len(face), *face))
    with open(os.path.join(d, "inc.pbrt"), "w") as fh:
        fh.write('AttributeBegin\n  Rotate 30 0 0 1\n  Translate 0.2 0 0\n'
                 '  Shape "plymesh" "string filename" "mesh.ply" "rgb Kd" [0.8 0.1 0.1]\nAttributeEnd\n')
    crop_s = f'"float cropwindow" [{fl(crop)}]' if crop is not None else ""
    strategy_s = f'"string lightsamplestrategy" "{strategy}"' if strategy else ""
    text = f"""# host-driver parity scene
Scale -1 1 1   # handedness flip, as most exported scenes have
LookAt 0.5 -5 2.5  0 0 0.5  0 0 1
Camera "perspective" "float fov" [38] "float lensradius" 0.02 "float focaldistance" 5
Film "image" "integer xresolution" [{XRES}] "integer yresolution" {YRES} "string filename" "scene.pfm" {crop_s}
Sampler "halton" "integer pixelsamples" {SPP}
{filter_line}
Integrator "path" "integer maxdepth" {MAXDEPTH} {strategy_s}
Accelerator "bvh" "integer maxnodeprims" 2
WorldBegin
AttributeBegin
  Rotate 25 1 0 0
  LightSource "infinite" "rgb L" [0.4 0.5 0.6] "rgb scale" [0.5 0.5 0.5]
AttributeEnd
LightSource "distant" "point from" [1 1 3] "point to" [0 0 0] "rgb L" [1.5 1.4 1.3]
TransformBegin
  Translate 0 0 0.25
  LightSource "point" "point from" [-1.5 -1 1] "rgb I" [6 6 8]
TransformEnd
AttributeBegin
  Translate 0 0 3
  Rotate 180 1 0 0
  AreaLightSource "diffuse" "rgb L" [9 8 7] "rgb scale" [2 2 2] "bool twosided" "false"
  Shape "trianglemesh" "integer indices" [{" ".join(map(str, QUAD_IDX))}] "point P" [{fl(LAMP)}]
AttributeEnd
Texture "grey" "color" "constant" "rgb value" [0.3 0.4 0.5]
Texture "rough" "float" "constant" "float value" 35
MakeNamedMaterial "floor" "string type" "matte" "texture Kd" "grey" "texture sigma" "rough"
AttributeBegin
  NamedMaterial "floor"
  Shape "trianglemesh" "integer indices" [{" ".join(map(str, QUAD_IDX))}] "point P" [{fl(FLOOR)}]
AttributeEnd
Include "inc.pbrt"
AttributeBegin
  Material "matte" "rgb Kd" [0.2 0.7 0.3] "float sigma" 10
  ReverseOrientation
  CoordSysTransform "camera"
  Translate 0 0 4.5
  Scale 0.5 0.5 0.5
  Shape "trianglemesh" "integer indices" [0 1 2] "point P" [{fl(TENT_P)}] "normal N" [{fl(TENT_N)}] "float uv" [{fl(TENT_UV)}] "float shadowalpha" 0
AttributeEnd
WorldEnd
"""
    path = os.path.join(d, "scene.pbrt")
    with open(path, "w") as fh:
        fh.write(text)
    return path


FILTERS = {
    "gaussian": ('PixelFilter "gaussian" "float xwidth" 1.5 "float ywidth" 1.25 "float alpha" 1.5', ("gaussian", (1.5, 1.25), (1.5, 0.0))),
    "box": ('PixelFilter "box"', ("box", (0.5, 0.5), (0.0, 0.0))),
    "mitchell": ('PixelFilter "mitchell" "float B" 0.25', ("mitchell", (2.0, 2.0), (0.25, 1.0 / 3.0))),
    "sinc": ('PixelFilter "sinc" "float xwidth" 3', ("sinc", (3.0, 4.0), (3.0, 0.0))),
    "triangle": ('PixelFilter "triangle" "float ywidth" 1', ("triangle", (2.0, 1.0), (0.0, 0.0))),
}


def capture(s, host, filt="gaussian", crop=None):
    """The same scene as direct capture calls, in the order the text issues them (light ids follow call order)."""
    I = (np.eye(4, dtype=np.float32).reshape(16),) * 2
    mul = host.compose
    ctm = mul(mul(I, host.scale([-1, 1, 1])), host.look_at([0.5, -5, 2.5], [0, 0, 0.5], [0, 0, 1]))
    c2w = (ctm[1], ctm[0])  # inverse(ctm)
    # world block
    t = mul(I, host.rotate(25, [1, 0, 0]))
    L = np.float32([0.4, 0.5, 0.6]) * np.float32([0.5, 0.5, 0.5])
    s.add_light_infinite(L, t[0], t[1])
    s.add_light_distant(np.float32([1.5, 1.4, 1.3]), host.distant_direction(I[0], [1, 1, 3], [0, 0, 0]))
    t = mul(I, host.translate([0, 0, 0.25]))
    s.add_light_point(np.float32([6, 6, 8]), host.point_position(t[0], t[1], [-1.5, -1, 1]))
    t = mul(mul(I, host.translate([0, 0, 3])), host.rotate(180, [1, 0, 0]))
    default_mat = s.add_material_matte((0.5, 0.5, 0.5), 0.0)
    lid = s.add_light_diffuse_area(np.float32([9, 8, 7]) * np.float32([2, 2, 2]), 2, two_sided=False)
    s.add_mesh(host.transform_points(t[0], LAMP), QUAD_IDX, default_mat, first_area_light=lid, swaps_handedness=host.swaps_handedness(t[0]))
    floor = s.add_material_matte((0.3, 0.4, 0.5), 35.0)
    s.add_mesh(FLOOR, QUAD_IDX, floor)
    t = mul(mul(I, host.rotate(30, [0, 0, 1])), host.translate([0.2, 0, 0]))
    red = s.add_material_matte((0.8, 0.1, 0.1), 0.0)
    s.add_mesh(host.transform_points(t[0], PLY_P), [0, 1, 2, 3, 0, 2, 2, 3, 4], red, N=host.transform_normals(t[1], PLY_N), swaps_handedness=host.swaps_handedness(t[0]))
    t = mul(mul(c2w, host.translate([0, 0, 4.5])), host.scale([0.5, 0.5, 0.5]))
    green = s.add_material_matte((0.2, 0.7, 0.3), 10.0)
    s.add_mesh(host.transform_points(t[0], TENT_P), [0, 1, 2], green, N=host.transform_normals(t[1], TENT_N), UV=TENT_UV, reverse_orientation=True,
               swaps_handedness=host.swaps_handedness(t[0]), shadow_alpha=0.0)
    # options
    kind, radius, params = FILTERS[filt][1]
    cw = (0.0, 1.0, 0.0, 1.0) if crop is None else tuple(crop)
    cb, table, sb = host.film_filter(kind, XRES, YRES, radius, params, cw)
    s.set_film(XRES, YRES, cb, radius, table)
    s.set_camera_perspective(host.perspective_raster_to_camera(38.0, XRES, YRES), c2w[0], lens_radius=0.02, focal_distance=5.0)
    s.set_sampler(0, SPP, sb)
    s.build_accel(0, 2)
    return cb, sb


def read_pfm(path):
    with open(path, "rb") as fh:
        assert fh.readline() == b"PF\n"
        w, h = map(int, fh.readline().split())
        assert fh.readline() == b"-1\n"
        data = np.frombuffer(fh.read(), "<f4").reshape(h, w, 3)
    return data[::-1].copy()  # rows are stored bottom-to-top
