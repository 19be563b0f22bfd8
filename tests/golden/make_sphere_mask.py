"""Sky mask of the reference's own render of scenes/shapes/sphere.pbrt (renders/shapes/sphere.png, 800 x 400, palette PNG) -> sphere_sky_mask.npz.
A pixel is "sky" where the render shows the pure environment (saturated white).  The mask pins the silhouettes of the two spheres and the ground plane's
horizon, i.e. camera + CTM handling + Sphere::intersect incl. the zmin / zmax / phimax cut, against pixels the reference itself produced.
Run in the build container (the reference does not travel): python3 tests/golden/make_sphere_mask.py"""
import os
import struct
import zlib

import numpy as np


def read_png(path):
    d = open(path, "rb").read()
    assert d[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, plte = 8, b"", None
    while pos < len(d):
        n, typ = struct.unpack(">I4s", d[pos:pos + 8]); body = d[pos + 8:pos + 8 + n]; pos += 12 + n
        if typ == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            assert depth == 8 and interlace == 0
        elif typ == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif typ == b"IDAT":
            idat += body
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * ch)
    out = np.zeros((h, w * ch), np.uint8)
    prev = np.zeros(w * ch, np.int32)
    for y in range(h):
        f, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        cur = np.zeros(w * ch, np.int32)
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        else:
            for x in range(w * ch):
                a = cur[x - ch] if x >= ch else 0
                b = prev[x]
                c = prev[x - ch] if x >= ch else 0
                if f == 1:
                    p = a
                elif f == 3:
                    p = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (line[x] + p) & 255
        out[y] = cur; prev = cur
    img = out.reshape(h, w, ch)
    if ctype == 3:
        img = plte[img[:, :, 0]]
    return img[:, :, :3]


if __name__ == "__main__":
    img = read_png("/root/reference/renders/shapes/sphere.png")
    sky = np.all(img >= 254, axis=-1)
    here = os.path.dirname(os.path.abspath(__file__))
    np.savez_compressed(os.path.join(here, "sphere_sky_mask.npz"), sky=np.packbits(sky), shape=np.array(sky.shape))
    print(img.shape, sky.mean())
