// read_image (core/src/image_io.rs:42-50): the texel source of ImageTexture.  Returns width*height RGB floats, top row first, exactly as
// the reference hands them to generate_mipmap: PFM values times |scale| (read_pfm, :127-190), 8-bit formats as u8 / 255.0 (read_8_bit,
// :192-224 — the reference decodes them with the `image` crate, v0.25, and converts to RGB8).  Decoders here: PFM, TGA (types 2, 3, 10, 11)
// and PNG (8 bits per channel, non-interlaced; inflate by zlib).  OpenEXR and JPEG are not decoded: convert such maps to PFM / PNG.
#include "pbrt_host.hpp"
#include <cmath>
#include <cstdio>
#include <cstring>
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
    if (w <= 0 || h <= 0) { err = "PFM: bad resolution"; return false; }
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

}  // namespace

bool read_image(const std::string& path, std::vector<float>& rgb, int& w, int& h, std::string& err) {
    const std::string ext = lower_ext(path);
    if (ext.empty()) { err = "Can't determine file type from suffix of filename " + path + "."; return false; }
    if (ext == ".exr" || ext == ".jpg" || ext == ".jpeg" || ext == ".bmp" || ext == ".gif" || ext == ".tif" || ext == ".tiff" || ext == ".hdr") {
        err = "image format '" + ext + "' is not decoded by this host (convert the map to .pfm, .png or .tga)";
        return false;
    }
    std::vector<unsigned char> d;
    if (!slurp(path, d, err)) return false;
    if (ext == ".pfm") return read_pfm(d, rgb, w, h, err);
    if (ext == ".tga") return read_tga(d, rgb, w, h, err);
    if (ext == ".png") return read_png(d, rgb, w, h, err);
    err = "image format '" + ext + "' is not decoded by this host (convert the map to .pfm, .png or .tga)";
    return false;
}

}  // namespace pbrt_host
