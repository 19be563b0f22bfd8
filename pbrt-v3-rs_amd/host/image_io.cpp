// read_image (core/src/image_io.rs:42-50): the texel source of ImageTexture.  Returns width*height RGB floats, top row first, exactly as
// the reference hands them to generate_mipmap: PFM values times |scale| (read_pfm, :127-190), 8-bit formats as u8 / 255.0 (read_8_bit,
// :192-224 — the reference decodes them with the `image` crate, v0.25, and converts to RGB8).  Decoders here: PFM, TGA (types 2, 3, 10, 11)
// PNG (8 bits per channel, non-interlaced; inflate by zlib) and OpenEXR (single-part scanline files, NONE / ZIPS / ZIP compression, half or float R G B [A]).
// write_image (:225-237): PFM, 8-bit PNG / TGA through apply_gamma (:379-390), EXR as uncompressed 32-bit float scanlines.  JPEG, tiled / PIZ / multipart EXR
// are not decoded: convert such files.
#include "pbrt_host.hpp"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <stdexcept>
#include <zlib.h>

namespace pbrt_host {
namespace {

bool slurp(const std::string& path, std::vector<unsigned char>& out, std::string& err) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open '" + path + "'"; return false; }
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const bool ok = out.empty() || std::fread(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    if (!ok) err = "short read on '" + path + "'";
    return ok;
}
std::string lower_ext(const std::string& p) {
    const size_t d = p.rfind('.');
    std::string e = d == std::string::npos ? "" : p.substr(d);
    for (char& c : e) c = (char)std::tolower((unsigned char)c);
    return e;
}
void from_u8(const std::vector<unsigned char>& rgb8, std::vector<float>& out) {
    out.resize(rgb8.size());
    for (size_t i = 0; i < rgb8.size(); i++) out[i] = (float)rgb8[i] / 255.0f;
}

// ---- PFM (image_io.rs:127-190): "PF"/"Pf", width, height, scale (negative = little endian), rows bottom to top
bool read_pfm(const std::vector<unsigned char>& d, std::vector<float>& rgb, int& w, int& h, std::string& err) {
    size_t pos = 0;
    auto word = [&](std::string& s) {
        s.clear();
        while (pos < d.size() && !(d[pos] == ' ' || d[pos] == '\n' || d[pos] == '\t')) s.push_back((char)d[pos++]);
        if (pos >= d.size()) return false;
        pos++;  // the single whitespace that ended the word
        return true;
    };
    std::string ty, sw, sh, ss;
    if (!word(ty) || !word(sw) || !word(sh) || !word(ss)) { err = "PFM: truncated header"; return false; }
    const int nc = ty == "PF" ? 3 : (ty == "Pf" ? 1 : 0);
    if (!nc) { err = "PFM: invalid type '" + ty + "'"; return false; }
    w = std::atoi(sw.c_str()); h = std::atoi(sh.c_str());
    float scale = std::strtof(ss.c_str(), nullptr);
    if (w <= 0 || h <= 0 || (long long)w * h > (1ll << 28)) { err = "PFM: bad resolution (or larger than 2^28 pixels)"; return false; }
    const bool little = scale < 0.0f;
    scale = std::fabs(scale);
    const size_t n = (size_t)nc * (size_t)w * (size_t)h;
    if (d.size() - pos < 4 * n) { err = "PFM: truncated pixel data"; return false; }
    rgb.assign(3 * (size_t)w * (size_t)h, 0.0f);
    for (int y = h - 1; y >= 0; y--)
        for (size_t j = 0; j < (size_t)w * nc; j++) {
            unsigned char b[4] = {d[pos], d[pos + 1], d[pos + 2], d[pos + 3]};
            pos += 4;
            if (!little) { std::swap(b[0], b[3]); std::swap(b[1], b[2]); }
            float f; std::memcpy(&f, b, 4);
            f *= scale;
            if (nc == 3) rgb[(size_t)y * w * 3 + j] = f;
            else { float* o = &rgb[((size_t)y * w + j) * 3]; o[0] = o[1] = o[2] = f; }
        }
    return true;
}

// ---- TGA: uncompressed / run-length true colour (24, 32 bpp) and grey (8 bpp); no colour maps
bool read_tga(const std::vector<unsigned char>& d, std::vector<float>& rgb, int& w, int& h, std::string& err) {
    if (d.size() < 18) { err = "TGA: truncated header"; return false; }
    const int id_len = d[0], cmap = d[1], type = d[2], bpp = d[16], desc = d[17];
    w = d[12] | (d[13] << 8); h = d[14] | (d[15] << 8);
    if (cmap != 0 || !(type == 2 || type == 3 || type == 10 || type == 11)) { err = "TGA: only true-colour and grey images without a colour map are decoded"; return false; }
    const int bytes = bpp / 8;
    if (!((type == 2 || type == 10) ? (bytes == 3 || bytes == 4) : bytes == 1) || w <= 0 || h <= 0) { err = "TGA: unsupported pixel depth"; return false; }
    size_t pos = 18 + (size_t)id_len;
    const size_t npx = (size_t)w * (size_t)h;
    std::vector<unsigned char> px(npx * (size_t)bytes);
    if (type == 2 || type == 3) {
        if (d.size() - pos < px.size()) { err = "TGA: truncated pixel data"; return false; }
        std::memcpy(px.data(), d.data() + pos, px.size());
    } else {
        size_t o = 0;
        while (o < px.size()) {
            if (pos >= d.size()) { err = "TGA: truncated run-length data"; return false; }
            const int hd = d[pos++], cnt = (hd & 127) + 1;
            if (hd & 128) {
                if (pos + bytes > d.size()) { err = "TGA: truncated run-length data"; return false; }
                for (int k = 0; k < cnt && o < px.size(); k++, o += bytes) std::memcpy(&px[o], &d[pos], (size_t)bytes);
                pos += (size_t)bytes;
            } else {
                const size_t nb = (size_t)cnt * bytes;
                if (pos + nb > d.size() || o + nb > px.size()) { err = "TGA: truncated run-length data"; return false; }
                std::memcpy(&px[o], &d[pos], nb); pos += nb; o += nb;
            }
        }
    }
    const bool top_first = (desc & 0x20) != 0, right_first = (desc & 0x10) != 0;
    std::vector<unsigned char> rgb8(npx * 3);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int sy = top_first ? y : h - 1 - y, sx = right_first ? w - 1 - x : x;
            const unsigned char* s = &px[((size_t)sy * w + sx) * bytes];
            unsigned char* o = &rgb8[((size_t)y * w + x) * 3];
            if (bytes == 1) o[0] = o[1] = o[2] = s[0];
            else { o[0] = s[2]; o[1] = s[1]; o[2] = s[0]; }  // stored BGR(A)
        }
    from_u8(rgb8, rgb);
    return true;
}

// ---- PNG: 8 bits per channel, colour types 0 (grey), 2 (RGB), 3 (palette), 4 (grey + alpha), 6 (RGBA); alpha is dropped (into_rgb8)
uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
bool read_png(const std::vector<unsigned char>& d, std::vector<float>& rgb, int& w, int& h, std::string& err) {
    static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (d.size() < 8 || std::memcmp(d.data(), sig, 8) != 0) { err = "PNG: bad signature"; return false; }
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, plte;
    bool have_hdr = false;
    while (pos + 8 <= d.size()) {
        const uint32_t len = be32(&d[pos]);
        const char* ty = (const char*)&d[pos + 4];
        if (pos + 12 + (size_t)len > d.size()) { err = "PNG: truncated chunk"; return false; }
        const unsigned char* body = &d[pos + 8];
        if (!std::memcmp(ty, "IHDR", 4)) {
            if (len < 13) { err = "PNG: bad IHDR"; return false; }
            w = (int)be32(body); h = (int)be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12]; have_hdr = true;
        } else if (!std::memcmp(ty, "PLTE", 4)) plte.assign(body, body + len);
        else if (!std::memcmp(ty, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(ty, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (!have_hdr || w <= 0 || h <= 0) { err = "PNG: missing IHDR"; return false; }
    if ((long long)w * h > (1ll << 28)) { err = "PNG: larger than 2^28 pixels"; return false; }
    if (depth != 8 || interlace != 0) { err = "PNG: only 8 bits per channel, non-interlaced images are decoded"; return false; }
    int ch;
    switch (ctype) { case 0: ch = 1; break; case 2: ch = 3; break; case 3: ch = 1; break; case 4: ch = 2; break; case 6: ch = 4; break; default: err = "PNG: bad colour type"; return false; }
    const size_t stride = (size_t)w * ch;
    std::vector<unsigned char> raw((stride + 1) * (size_t)h);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) { err = "PNG: inflate failed"; return false; }
    std::vector<unsigned char> img(stride * (size_t)h);
    for (int y = 0; y < h; y++) {  // undo the per-row filters (PNG specification, section 9)
        const unsigned char* in = &raw[(stride + 1) * (size_t)y];
        const int ft = in[0];
        unsigned char* cur = &img[stride * (size_t)y];
        const unsigned char* up = y ? cur - stride : nullptr;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= (size_t)ch ? cur[i - ch] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)ch) ? up[i - ch] : 0;
            int pred = 0;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) / 2; break;
                case 4: { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: err = "PNG: bad filter type"; return false;
            }
            cur[i] = (unsigned char)(in[1 + i] + pred);
        }
    }
    std::vector<unsigned char> rgb8((size_t)w * h * 3);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const unsigned char* s = &img[i * ch];
        unsigned char* o = &rgb8[3 * i];
        if (ctype == 3) {
            if ((size_t)s[0] * 3 + 2 >= plte.size()) { err = "PNG: palette index out of range"; return false; }
            o[0] = plte[3 * s[0]]; o[1] = plte[3 * s[0] + 1]; o[2] = plte[3 * s[0] + 2];
        } else if (ch <= 2) o[0] = o[1] = o[2] = s[0];
        else { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; }
    }
    from_u8(rgb8, rgb);
    return true;
}

// ---- OpenEXR (OpenEXR file layout, openexr.com/TechnicalIntroduction): magic, version, attributes, scanline offset table, blocks --------------
float half_to_float(uint16_t h) {
    const uint32_t sgn = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31u, m = h & 1023u;
    uint32_t bits;
    if (e == 0) {
        if (m == 0) bits = sgn;
        else {  // subnormal half: renormalise
            int ex = -1; uint32_t mm = m;
            do { ex++; mm <<= 1; } while (!(mm & 1024u));
            bits = sgn | ((uint32_t)(127 - 15 - ex) << 23) | ((mm & 1023u) << 13);
        }
    } else if (e == 31) bits = sgn | 0x7F800000u | (m << 13);
    else bits = sgn | ((e + 112u) << 23) | (m << 13);
    float f; std::memcpy(&f, &bits, 4); return f;
}
bool read_exr(const std::vector<unsigned char>& d, std::vector<float>& rgb, int& w, int& h, std::string& err) {
    auto u32 = [&](size_t p) { return (uint32_t)d[p] | ((uint32_t)d[p + 1] << 8) | ((uint32_t)d[p + 2] << 16) | ((uint32_t)d[p + 3] << 24); };
    auto u64 = [&](size_t p) { return (uint64_t)u32(p) | ((uint64_t)u32(p + 4) << 32); };
    if (d.size() < 8 || u32(0) != 20000630u) { err = "EXR: bad magic number"; return false; }
    const uint32_t ver = u32(4);
    if ((ver & 0xFFu) != 2u || (ver & 0x1A00u)) { err = "EXR: tiled, deep and multipart files are not decoded by this host (convert the map to .pfm, .png or .tga)"; return false; }
    size_t pos = 8;
    struct Chan { std::string name; int type; };
    std::vector<Chan> chans;
    int compression = -1, dw[4] = {0, 0, -1, -1}, line_order = 0;
    for (;;) {  // attributes: name\0 type\0 size value
        if (pos >= d.size()) { err = "EXR: truncated header"; return false; }
        if (d[pos] == 0) { pos++; break; }
        std::string name, type;
        while (pos < d.size() && d[pos]) name.push_back((char)d[pos++]);
        pos++;
        while (pos < d.size() && d[pos]) type.push_back((char)d[pos++]);
        pos++;
        if (pos + 4 > d.size()) { err = "EXR: truncated header"; return false; }
        const uint32_t size = u32(pos); pos += 4;
        if (pos + size > d.size()) { err = "EXR: truncated header"; return false; }
        if (name == "channels") {
            size_t q = pos;
            const size_t end = pos + size;   // every cursor stays inside the attribute (which the check above keeps inside the file)
            while (q < end && d[q]) {
                Chan c;
                while (q < end && d[q]) c.name.push_back((char)d[q++]);
                q++;
                if (q + 16 > end) { err = "EXR: truncated channel list"; return false; }
                c.type = (int)u32(q);
                const uint32_t xs = u32(q + 8), ys = u32(q + 12);
                if (xs != 1 || ys != 1) { err = "EXR: subsampled channels are not decoded by this host (convert the map to .pfm, .png or .tga)"; return false; }
                q += 16;
                chans.push_back(c);
            }
        } else if (name == "compression") { if (size < 1) { err = "EXR: empty compression attribute"; return false; } compression = d[pos]; }
        else if (name == "dataWindow") { if (size < 16) { err = "EXR: short dataWindow attribute"; return false; } for (int k = 0; k < 4; k++) dw[k] = (int)u32(pos + 4 * (size_t)k); }
        else if (name == "lineOrder") { if (size < 1) { err = "EXR: empty lineOrder attribute"; return false; } line_order = d[pos]; }
        pos += size;
    }
    (void)line_order;
    if (!(compression == 0 || compression == 2 || compression == 3)) {
        err = "EXR: compression method " + std::to_string(compression) + " (RLE / PIZ / PXR24 / B44 / DWA) is not decoded by this host (convert the map to .pfm, .png or .tga)";
        return false;
    }
    const long long w64 = (long long)dw[2] - (long long)dw[0] + 1, h64 = (long long)dw[3] - (long long)dw[1] + 1;
    if (w64 <= 0 || h64 <= 0 || w64 > 65536 || h64 > 65536 || w64 * h64 > (1ll << 28) || chans.empty()) { err = "EXR: bad data window (or larger than 2^28 pixels) or no channels"; return false; }
    w = (int)w64; h = (int)h64;
    int ci[3] = {-1, -1, -1};
    size_t row_bytes = 0;
    std::vector<size_t> chan_off(chans.size());
    for (size_t c = 0; c < chans.size(); c++) {
        chan_off[c] = row_bytes;
        if (chans[c].type < 0 || chans[c].type > 2) { err = "EXR: bad pixel type"; return false; }
        row_bytes += (size_t)w * (chans[c].type == 1 ? 2 : 4);
        if (chans[c].name == "R") ci[0] = (int)c; else if (chans[c].name == "G") ci[1] = (int)c; else if (chans[c].name == "B") ci[2] = (int)c;
    }
    if (ci[0] < 0 || ci[1] < 0 || ci[2] < 0) { err = "EXR: the file has no R, G, B channels"; return false; }
    const int lines_per_block = compression == 3 ? 16 : 1;
    const size_t n_blocks = ((size_t)h + lines_per_block - 1) / lines_per_block;
    if (n_blocks > (d.size() - pos) / 8) { err = "EXR: truncated offset table"; return false; }
    rgb.assign(3 * (size_t)w * h, 0.0f);
    std::vector<unsigned char> raw, tmp;
    for (size_t b = 0; b < n_blocks; b++) {
        const uint64_t off = u64(pos + 8 * b);
        if (d.size() < 8 || off > d.size() - 8) { err = "EXR: bad block offset"; return false; }   // `off` comes from the file: no arithmetic on it before this test
        const long long y0l = (long long)(int)u32((size_t)off) - (long long)dw[1];
        const uint32_t sz = u32((size_t)off + 4);
        if (sz > d.size() - 8 - (size_t)off || y0l < 0 || y0l >= h) { err = "EXR: bad scanline block"; return false; }
        const int y0 = (int)y0l;
        const int nl = std::min(lines_per_block, h - y0);
        const size_t want = row_bytes * (size_t)nl;
        const unsigned char* src = &d[(size_t)off + 8];
        if (compression == 0 || sz == want) raw.assign(src, src + std::min<size_t>(sz, want));   // a block that did not shrink is stored raw
        else {
            tmp.resize(want);
            uLongf out_len = (uLongf)want;
            if (uncompress(tmp.data(), &out_len, src, sz) != Z_OK || out_len != want) { err = "EXR: inflate failed"; return false; }
            for (size_t i = 1; i < want; i++) tmp[i] = (unsigned char)(tmp[i - 1] + tmp[i] - 128);   // undo the byte predictor
            raw.resize(want);
            const size_t half = (want + 1) / 2;                                                      // undo the even / odd byte split
            for (size_t i = 0, a = 0, c2 = half; i < want; ) { raw[i++] = tmp[a++]; if (i < want) raw[i++] = tmp[c2++]; }
        }
        if (raw.size() < want) { err = "EXR: short scanline block"; return false; }
        for (int l = 0; l < nl; l++)
            for (int k = 0; k < 3; k++) {
                const Chan& c = chans[(size_t)ci[k]];
                const unsigned char* row = &raw[row_bytes * (size_t)l + chan_off[(size_t)ci[k]]];
                for (int x = 0; x < w; x++) {
                    float v;
                    if (c.type == 1) v = half_to_float((uint16_t)(row[2 * x] | (row[2 * x + 1] << 8)));
                    else if (c.type == 2) std::memcpy(&v, row + 4 * x, 4);
                    else { uint32_t u; std::memcpy(&u, row + 4 * x, 4); v = (float)u; }
                    rgb[3 * ((size_t)(y0 + l) * w + x) + k] = v;
                }
            }
    }
    return true;
}

}  // namespace

static bool read_image_unchecked(const std::string& path, std::vector<float>& rgb, int& w, int& h, std::string& err) {
    const std::string ext = lower_ext(path);
    if (ext.empty()) { err = "Can't determine file type from suffix of filename " + path + "."; return false; }
    if (ext == ".jpg" || ext == ".jpeg" || ext == ".bmp" || ext == ".gif" || ext == ".tif" || ext == ".tiff" || ext == ".hdr") {
        err = "image format '" + ext + "' is not decoded by this host (convert the map to .pfm, .png or .tga)";
        return false;
    }
    std::vector<unsigned char> d;
    if (!slurp(path, d, err)) return false;
    if (ext == ".pfm") return read_pfm(d, rgb, w, h, err);
    if (ext == ".tga") return read_tga(d, rgb, w, h, err);
    if (ext == ".png") return read_png(d, rgb, w, h, err);
    if (ext == ".exr") return read_exr(d, rgb, w, h, err);
    err = "image format '" + ext + "' is not decoded by this host (convert the map to .pfm, .png or .tga)";
    return false;
}

// A file the decoders cannot hold in memory is a file that "cannot be read": the caller warns and goes on as the reference does (imagemap.rs / infinite.rs),
// it does not die in an allocation
bool read_image(const std::string& path, std::vector<float>& rgb, int& w, int& h, std::string& err) {
    try {
        return read_image_unchecked(path, rgb, w, h, err);
    } catch (const std::bad_alloc&) {
        err = "image " + path + ": not enough memory to decode";
        rgb.clear(); w = h = 0;
        return false;
    } catch (const std::length_error&) {
        err = "image " + path + ": implausible size";
        rgb.clear(); w = h = 0;
        return false;
    }
}

// ---- write_image (core/src/image_io.rs:225-237) ------------------------------------------------------------------------------------------
namespace {
inline float gamma_correct(float v) { return v <= 0.0031308f ? 12.92f * v : 1.055f * std::pow(v, 1.0f / 2.4f) - 0.055f; }  // pbrt/common.rs:140-146
inline unsigned char clamp_byte(float v) {  // image_io.rs:387-390: clamp(255 * gamma_correct(v) + 0.5, 0, 255) as u8
    const float x = 255.0f * gamma_correct(v) + 0.5f;
    return (unsigned char)(x < 0.0f ? 0.0f : (x > 255.0f ? 255.0f : x));   // NaN compares false twice and casts to 0, like `as u8`
}
bool put(const std::string& path, const std::vector<unsigned char>& bytes, std::string& err) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { err = "Error saving output image " + path + ": cannot open"; return false; }
    const bool ok = std::fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    std::fclose(f);
    if (!ok) err = "Error saving output image " + path + ": short write";
    return ok;
}
void be32put(std::vector<unsigned char>& o, uint32_t v) { o.push_back((unsigned char)(v >> 24)); o.push_back((unsigned char)(v >> 16)); o.push_back((unsigned char)(v >> 8)); o.push_back((unsigned char)v); }
void le32put(std::vector<unsigned char>& o, uint32_t v) { o.push_back((unsigned char)v); o.push_back((unsigned char)(v >> 8)); o.push_back((unsigned char)(v >> 16)); o.push_back((unsigned char)(v >> 24)); }
void png_chunk(std::vector<unsigned char>& o, const char* ty, const std::vector<unsigned char>& body) {
    be32put(o, (uint32_t)body.size());
    const size_t start = o.size();
    o.insert(o.end(), ty, ty + 4); o.insert(o.end(), body.begin(), body.end());
    be32put(o, (uint32_t)crc32(0L, &o[start], (uInt)(o.size() - start)));
}
}  // namespace
bool write_image(const std::string& path, const float* rgb, int w, int h, std::string& err) {
    const std::string ext = lower_ext(path);
    if (ext == ".pfm") return write_pfm(path, rgb, w, h, err);
    std::vector<unsigned char> out;
    if (ext == ".png" || ext == ".tga") {
        std::vector<unsigned char> px((size_t)w * h * 3);
        for (size_t i = 0; i < px.size(); i++) px[i] = clamp_byte(rgb[i]);
        if (ext == ".tga") {  // true colour, 24 bits, top-left origin
            const unsigned char hd[18] = {0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, (unsigned char)(w & 255), (unsigned char)(w >> 8), (unsigned char)(h & 255), (unsigned char)(h >> 8), 24, 0x20};
            out.assign(hd, hd + 18);
            for (size_t i = 0; i < (size_t)w * h; i++) { out.push_back(px[3 * i + 2]); out.push_back(px[3 * i + 1]); out.push_back(px[3 * i]); }
            return put(path, out, err);
        }
        static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
        out.assign(sig, sig + 8);
        std::vector<unsigned char> ihdr;
        be32put(ihdr, (uint32_t)w); be32put(ihdr, (uint32_t)h);
        const unsigned char tail[5] = {8, 2, 0, 0, 0};
        ihdr.insert(ihdr.end(), tail, tail + 5);
        png_chunk(out, "IHDR", ihdr);
        std::vector<unsigned char> raw;
        raw.reserve(((size_t)w * 3 + 1) * h);
        for (int y = 0; y < h; y++) { raw.push_back(0); raw.insert(raw.end(), &px[(size_t)y * w * 3], &px[(size_t)y * w * 3] + (size_t)w * 3); }
        uLongf clen = compressBound((uLong)raw.size());
        std::vector<unsigned char> comp(clen);
        if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { err = "Error saving output image " + path + ": deflate failed"; return false; }
        comp.resize(clen);
        png_chunk(out, "IDAT", comp);
        png_chunk(out, "IEND", {});
        return put(path, out, err);
    }
    if (ext == ".exr") {  // single-part scanline file, no compression, 32-bit float B G R (channels are stored in alphabetical order), increasing y
        le32put(out, 20000630u); le32put(out, 2u);
        auto attr = [&](const char* name, const char* type, const std::vector<unsigned char>& v) {
            out.insert(out.end(), name, name + std::strlen(name) + 1); out.insert(out.end(), type, type + std::strlen(type) + 1);
            le32put(out, (uint32_t)v.size()); out.insert(out.end(), v.begin(), v.end());
        };
        std::vector<unsigned char> v;
        for (const char* c : {"B", "G", "R"}) { v.push_back((unsigned char)c[0]); v.push_back(0); le32put(v, 2u); le32put(v, 0u); le32put(v, 1u); le32put(v, 1u); }
        v.push_back(0);
        attr("channels", "chlist", v);
        attr("compression", "compression", {0});
        v.clear(); le32put(v, 0u); le32put(v, 0u); le32put(v, (uint32_t)(w - 1)); le32put(v, (uint32_t)(h - 1));
        attr("dataWindow", "box2i", v); attr("displayWindow", "box2i", v);
        attr("lineOrder", "lineOrder", {0});
        auto f32 = [](float f) { std::vector<unsigned char> b(4); std::memcpy(b.data(), &f, 4); return b; };
        attr("pixelAspectRatio", "float", f32(1.0f));
        v = f32(0.0f); { const std::vector<unsigned char> z = f32(0.0f); v.insert(v.end(), z.begin(), z.end()); }
        attr("screenWindowCenter", "v2f", v);
        attr("screenWindowWidth", "float", f32(1.0f));
        out.push_back(0);
        const size_t row_bytes = (size_t)w * 12, table = out.size();
        out.resize(out.size() + 8 * (size_t)h);
        for (int y = 0; y < h; y++) {
            const uint64_t off = out.size();
            for (int k = 0; k < 8; k++) out[table + 8 * (size_t)y + (size_t)k] = (unsigned char)(off >> (8 * k));
            le32put(out, (uint32_t)y); le32put(out, (uint32_t)row_bytes);
            for (int c = 2; c >= 0; c--)
                for (int x = 0; x < w; x++) { unsigned char b[4]; std::memcpy(b, &rgb[3 * ((size_t)y * w + x) + c], 4); out.insert(out.end(), b, b + 4); }
        }
        return put(path, out, err);
    }
    err = ext.empty() ? "Can't determine file type from suffix of filename " + path : "Extension " + ext + " is not supported";
    return false;
}

}  // namespace pbrt_host
