"""Writers for the image formats the host front end decodes (tests only): PFM, TGA, PNG."""
import struct
import zlib

import numpy as np


def write_pfm(path, img, little=True, scale=1.0):
    """img (h, w, 3) or (h, w) float32, top row first.  PFM stores rows bottom to top."""
    img = np.asarray(img, np.float32)
    h, w = img.shape[:2]
    ty = b"PF" if img.ndim == 3 else b"Pf"
    with open(path, "wb") as f:
        f.write(ty + b"\n%d %d\n%s\n" % (w, h, repr(-scale if little else scale).encode()))
        f.write(img[::-1].astype("<f4" if little else ">f4").tobytes())


def write_tga(path, img8, rle=False, top_origin=False, grey=False, alpha=False):
    """img8 (h, w, 3) uint8 (or (h, w) for grey), top row first."""
    img8 = np.asarray(img8, np.uint8)
    h, w = img8.shape[:2]
    rows = img8 if top_origin else img8[::-1]
    if grey:
        px = rows.reshape(h * w, 1)
    else:
        bgr = rows[..., ::-1].reshape(h * w, 3)
        px = np.concatenate([bgr, np.full((h * w, 1), 255, np.uint8)], axis=1) if alpha else bgr
    bpp = px.shape[1] * 8
    ty = (11 if grey else 10) if rle else (3 if grey else 2)
    hdr = struct.pack("<BBBHHBHHHHBB", 0, 0, ty, 0, 0, 0, 0, 0, w, h, bpp, (0x20 if top_origin else 0) | (8 if alpha else 0))
    body = bytearray()
    if not rle:
        body += px.tobytes()
    else:
        i, n = 0, len(px)
        while i < n:  # alternate run packets and raw packets
            run = 1
            while i + run < n and run < 128 and (px[i + run] == px[i]).all():
                run += 1
            if run > 1:
                body.append(0x80 | (run - 1)); body += px[i].tobytes(); i += run
            else:
                raw = 1
                while i + raw < n and raw < 128 and not (px[i + raw] == px[i + raw - 1]).all():
                    raw += 1
                body.append(raw - 1); body += px[i:i + raw].tobytes(); i += raw
    with open(path, "wb") as f:
        f.write(hdr + bytes(body))


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def write_png(path, img8, color_type=2, filters=(0, 1, 2, 3, 4), palette=None):
    """8-bit PNG.  color_type 0 grey (h,w), 2 RGB (h,w,3), 3 palette (h,w) + palette (n,3), 4 grey+alpha (h,w,2), 6 RGBA (h,w,4).
    Row y is encoded with filter type filters[y % len(filters)]; several IDAT chunks."""
    a = np.asarray(img8, np.uint8)
    h, w = a.shape[:2]
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]
    rows = a.reshape(h, w * ch).astype(np.int32)
    raw = bytearray()
    for y in range(h):
        ft = filters[y % len(filters)]
        cur = rows[y]; up = rows[y - 1] if y else np.zeros_like(cur)
        out = np.zeros_like(cur)
        for i in range(len(cur)):
            l = cur[i - ch] if i >= ch else 0
            u = up[i]
            ul = up[i - ch] if i >= ch else 0
            pred = (0, l, u, (l + u) // 2, _paeth(int(l), int(u), int(ul)))[ft]
            out[i] = (cur[i] - pred) & 255
        raw.append(ft); raw += out.astype(np.uint8).tobytes()
    comp = zlib.compress(bytes(raw), 6)

    def chunk(ty, data):
        return struct.pack(">I", len(data)) + ty + data + struct.pack(">I", zlib.crc32(ty + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0)))
        if color_type == 3:
            f.write(chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes()))
        third = max(len(comp) // 3, 1)
        for i in range(0, len(comp), third):
            f.write(chunk(b"IDAT", comp[i:i + third]))
        f.write(chunk(b"IEND", b""))


def write_exr(path, img, compression="none", half=False, with_alpha=False, decreasing_y=False):
    """Single-part scanline OpenEXR (tests only): channels [A] B G R, float or half, compression none | zips | zip."""
    img = np.asarray(img, np.float32)
    h, w = img.shape[:2]
    comp = {"none": 0, "zips": 2, "zip": 3}[compression]
    names = (["A"] if with_alpha else []) + ["B", "G", "R"]
    ptype = 1 if half else 2

    def attr(name, ty, val):
        return name.encode() + b"\0" + ty.encode() + b"\0" + struct.pack("<I", len(val)) + val
    ch = b"".join(n.encode() + b"\0" + struct.pack("<IBBBBII", ptype, 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    hdr = struct.pack("<II", 20000630, 2) + attr("channels", "chlist", ch) + attr("compression", "compression", bytes([comp])) + attr("dataWindow", "box2i", box) + \
        attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", bytes([1 if decreasing_y else 0])) + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + \
        attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    lpb = 16 if comp == 3 else 1
    blocks = []
    for y0 in range(0, h, lpb):
        raw = bytearray()
        for y in range(y0, min(y0 + lpb, h)):
            for n in names:
                row = np.ones(w, np.float32) if n == "A" else img[y, :, "RGB".index(n)]
                raw += (row.astype(np.float16) if half else row.astype("<f4")).tobytes()
        raw = bytes(raw)
        data = raw
        if comp:
            a = np.frombuffer(raw, np.uint8)
            t = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)         # even / odd byte split
            p = t.copy(); p[1:] = (t[1:] - t[:-1] + 128 + 256) % 256        # byte predictor
            z = zlib.compress(p.astype(np.uint8).tobytes(), 6)
            data = z if len(z) < len(raw) else raw                          # blocks that do not shrink are stored raw
        blocks.append((y0, data))
    order = list(reversed(blocks)) if decreasing_y else blocks
    table_pos = len(hdr)
    body = bytearray(); offs = {}
    pos = table_pos + 8 * len(blocks)
    for y0, data in order:
        offs[y0] = pos
        body += struct.pack("<iI", y0, len(data)) + data
        pos += 8 + len(data)
    table = b"".join(struct.pack("<Q", offs[y0]) for y0, _ in blocks)
    with open(path, "wb") as f:
        f.write(hdr + table + bytes(body))


def read_png_rgb8(path):
    """Decoder for the PNGs this repo's host writes (8-bit RGB, filter 0 rows)."""
    d = open(path, "rb").read()
    assert d[:8] == b"\x89PNG\r\n\x1a\n"
    pos = 8; idat = b""; w = h = 0
    while pos < len(d):
        n, ty = struct.unpack(">I4s", d[pos:pos + 8])
        body = d[pos + 8:pos + 8 + n]
        assert zlib.crc32(ty + body) & 0xFFFFFFFF == struct.unpack(">I", d[pos + 8 + n:pos + 12 + n])[0]
        if ty == b"IHDR": w, h, depth, ct = struct.unpack(">IIBB", body[:10]); assert (depth, ct) == (8, 2)
        elif ty == b"IDAT": idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 3)
