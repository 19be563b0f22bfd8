// C ABI of include/pbrt_hip.h, texture side: MIPMap construction (the host half of core/src/mipmap/mod.rs), the texture constructors that flatten
// the scene's texture trees into postfix programs (textures/src/*.rs), the calls that attach textures to material parameters, bump maps, alpha masks and
// the radiance map of an infinite light, plus the probes the tests use.  The device half lives in texture.h; the scene record in scene_host.h.
#include "host_math.h"
#include "scene_host.h"
#include "traverse.h"
#include "texture.h"
#include <algorithm>
#include <cstring>

using namespace phost;

namespace {
// Host-side MIPMap::texel / triangle / lookup_triangle (mipmap/mod.rs:222-238, 293-312, 569-608) over the texel pool, for what InfiniteAreaLight does at
// construction time (compute_scalar_image, power): host libm log2f, as in the reference.
struct Rgb { float c[3]; };
Rgb hmip_texel(const PbrtHipScene* s, const MipRec& m, uint32_t level, long long x, long long y) {
    const long long W = (long long)m.level_w[level], H = (long long)m.level_h[level];
    auto rem = [](long long a, long long b) { long long r = a - (a / b) * b; return r < 0 ? r + b : r; };
    if (m.wrap == 0u) { x = rem(x, W); y = rem(y, H); }
    else if (m.wrap == 2u) { x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x); y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y); }
    else if (x < 0 || x >= W || y < 0 || y >= H) return Rgb{{0.0f, 0.0f, 0.0f}};
    const Texel& t = s->texels[(size_t)m.level_off[level] + (size_t)y * (size_t)W + (size_t)x];
    return Rgb{{t.r, t.g, t.b}};
}
Rgb hmip_triangle(const PbrtHipScene* s, const MipRec& m, uint32_t level, float u, float v) {
    if (level > m.n_levels - 1u) level = m.n_levels - 1u;
    const float x = u * (float)m.level_w[level] - 0.5f, y = v * (float)m.level_h[level] - 0.5f;
    const float fx = std::floor(x), fy = std::floor(y);
    const long long x0 = (long long)fx, y0 = (long long)fy;
    const float ds = x - (float)x0, dt = y - (float)y0;
    const Rgb a = hmip_texel(s, m, level, x0, y0), b = hmip_texel(s, m, level, x0, y0 + 1), c = hmip_texel(s, m, level, x0 + 1, y0), d = hmip_texel(s, m, level, x0 + 1, y0 + 1);
    Rgb r;
    for (int k = 0; k < 3; k++) r.c[k] = ((a.c[k] * (1.0f - ds) * (1.0f - dt) + b.c[k] * (1.0f - ds) * dt) + c.c[k] * ds * (1.0f - dt)) + d.c[k] * ds * dt;
    return r;
}
Rgb hmip_lookup_triangle_rgb(const PbrtHipScene* s, const MipRec& m, float u, float v, float width) {
    const uint32_t levels = m.n_levels;
    const float level = (float)levels - 1.0f + std::log2(width > 1e-8f ? width : 1e-8f);
    if (level < 0.0f) return hmip_triangle(s, m, 0u, u, v);
    if (level >= (float)(levels - 1u)) return hmip_texel(s, m, levels - 1u, 0, 0);
    const uint32_t il = (uint32_t)std::floor(level);
    const float delta = level - (float)il;
    const Rgb a = hmip_triangle(s, m, il, u, v), b = hmip_triangle(s, m, il + 1u, u, v);
    Rgb r;
    for (int k = 0; k < 3; k++) r.c[k] = a.c[k] * (1.0f - delta) + b.c[k] * delta;
    return r;
}
}  // namespace

namespace phost {
void hmip_lookup_triangle(const PbrtHipScene* s, const MipRec& m, float u, float v, float width, float out[3]) {
    const Rgb t = hmip_lookup_triangle_rgb(s, m, u, v, width);
    out[0] = t.c[0]; out[1] = t.c[1]; out[2] = t.c[2];
}
}  // namespace phost

extern "C" {

// ---- textures --------------------------------------------------------------------------------------------------------------------
namespace {
inline float inv_gamma_correct(float v) {  // pbrt/common.rs:152-158 (host powf, as in the reference)
    if (v <= 0.04045f) return v * 1.0f / 12.92f;
    return std::pow((v + 0.055f) * 1.0f / 1.055f, 2.4f);
}
inline float lanczos2(float x) {  // texture/common.rs:216-228 with tau = 2
    x = std::fabs(x);
    if (x < 1e-5f) return 1.0f;
    if (x > 1.0f) return 0.0f;
    x *= hm::kPi;
    const float s = std::sin(x * 2.0f) / (x * 2.0f);
    const float l = std::sin(x) / x;
    return s * l;
}
struct RsWeight { size_t first; float w[4]; };
// resample_weights (mipmap/mod.rs:535-560).  `first` is a usize in the reference: a negative start saturates to 0.
void resample_weights(size_t old_res, size_t new_res, std::vector<RsWeight>& wt) {
    wt.resize(new_res);
    for (size_t i = 0; i < new_res; i++) {
        const float center = ((float)i + 0.5f) * (float)old_res / (float)new_res;
        const float f = std::floor((center - 2.0f) + 0.5f);
        wt[i].first = f > 0.0f ? (size_t)f : 0;
        for (int j = 0; j < 4; j++) wt[i].w[j] = lanczos2((((float)wt[i].first + (float)j + 0.5f) - center) / 2.0f);
        const float inv = 1.0f / (wt[i].w[0] + wt[i].w[1] + wt[i].w[2] + wt[i].w[3]);
        for (int j = 0; j < 4; j++) wt[i].w[j] *= inv;
    }
}
inline size_t wrap_index(size_t i, size_t n, int wrap) { return wrap == 0 ? i % n : (wrap == 2 ? (i > n - 1 ? n - 1 : i) : i); }
int push_texture(PbrtHipScene* s, PbrtHipScene::TextureHost&& t, uint32_t* out_id) {
    if (t.stack_need > PH_TEX_STACK) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "texture: the tree needs more than 6 live values");
    s->textures.push_back(std::move(t));
    if (out_id) *out_id = (uint32_t)s->textures.size() - 1;
    s->uploaded = false;
    return PBRT_HIP_OK;
}
}  // namespace

namespace {
// MIPMap::new (mipmap/mod.rs:115-189) on texels already in texture orientation, three floats per texel
int build_pyramid(PbrtHipScene* s, std::vector<float> img, size_t w, size_t h, int filtering, int wrap, bool as_float, float max_anisotropy, uint32_t* out_id) {
    auto is_pow2 = [](size_t v) { return (v & (v - 1)) == 0; };
    if (!is_pow2(w) || !is_pow2(h)) {  // resample_image (mipmap/mod.rs:383-529)
        size_t rw = 1, rh = 1;
        while (rw < w) rw <<= 1;
        while (rh < h) rh <<= 1;
        std::vector<float> r(3 * rw * rh, 0.0f);
        std::vector<RsWeight> sw, tw;
        resample_weights(w, rw, sw);
        for (size_t t = 0; t < h; t++)
            for (size_t x = 0; x < rw; x++) {
                float px[3] = {0.0f, 0.0f, 0.0f};
                for (int j = 0; j < 4; j++) {
                    const size_t o = wrap_index(sw[x].first + (size_t)j, w, wrap);
                    if (o < w) for (int c = 0; c < 3; c++) px[c] += img[3 * (t * w + o) + c] * sw[x].w[j];
                }
                for (int c = 0; c < 3; c++) r[3 * (t * rw + x) + c] += px[c];
            }
        resample_weights(h, rh, tw);
        std::vector<float> col(3 * rh);
        for (size_t x = 0; x < rw; x++) {
            for (size_t t = 0; t < rh; t++) {
                float px[3] = {0.0f, 0.0f, 0.0f};
                for (int j = 0; j < 4; j++) {
                    const size_t o = wrap_index(tw[t].first + (size_t)j, h, wrap);
                    if (o < h) for (int c = 0; c < 3; c++) px[c] += r[3 * (o * rw + x) + c] * tw[t].w[j];
                }
                for (int c = 0; c < 3; c++) col[3 * t + c] = px[c];
            }
            for (size_t t = 0; t < rh; t++)
                for (int c = 0; c < 3; c++) r[3 * (t * rw + x) + c] = hm::clampf(col[3 * t + c], 0.0f, hm::kInf);  // clamp_default
        }
        img.swap(r); w = rw; h = rh;
    }
    MipRec m{};
    m.filtering = (uint32_t)filtering; m.wrap = (uint32_t)wrap; m.is_float = as_float ? 1u : 0u; m.max_anisotropy = max_anisotropy;
    size_t lw = w, lh = h;
    std::vector<float> cur = std::move(img);
    for (;;) {
        if (m.n_levels >= PH_MIP_MAX_LEVELS) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_mipmap: too many pyramid levels");
        const uint32_t L = m.n_levels++;
        if (s->texels.size() + lw * lh > 0xFFFFFFFFull) return set_err(s, PBRT_HIP_ERR_OOM, "add_mipmap: texel pool exceeds 2^32 texels");
        if (lw > (1u << 30) || lh > (1u << 30)) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_mipmap: a level wider or higher than 2^30 texels");  // mip_texel_i's 32-bit coordinates
        m.level_off[L] = (uint32_t)s->texels.size(); m.level_w[L] = (uint32_t)lw; m.level_h[L] = (uint32_t)lh;
        for (size_t i = 0; i < lw * lh; i++) s->texels.push_back(Texel{cur[3 * i], cur[3 * i + 1], cur[3 * i + 2], 0.0f});
        if (lw == 1 && lh == 1) break;
        // next level: four texels of this one, fetched through the wrap mode like MIPMap::texel (mod.rs:150-164)
        const size_t nw = lw / 2 > 1 ? lw / 2 : 1, nh = lh / 2 > 1 ? lh / 2 : 1;
        std::vector<float> nxt(3 * nw * nh);
        auto tex = [&](long long x, long long y, int c) -> float {
            const long long W = (long long)lw, H = (long long)lh;
            if (wrap == 0) { x %= W; y %= H; }            // indices are non-negative here
            else if (wrap == 2) { x = x > W - 1 ? W - 1 : x; y = y > H - 1 ? H - 1 : y; }
            else if (x >= W || y >= H) return 0.0f;
            return cur[3 * ((size_t)y * lw + (size_t)x) + (size_t)c];
        };
        for (size_t t = 0; t < nh; t++)
            for (size_t x = 0; x < nw; x++)
                for (int c = 0; c < 3; c++) {
                    const long long X = 2 * (long long)x, Y = 2 * (long long)t;
                    nxt[3 * (t * nw + x) + c] = (((tex(X, Y, c) + tex(X + 1, Y, c)) + tex(X, Y + 1, c)) + tex(X + 1, Y + 1, c)) * 0.25f;
                }
        cur.swap(nxt); lw = nw; lh = nh;
    }
    s->mipmaps.push_back(m);
    if (out_id) *out_id = (uint32_t)s->mipmaps.size() - 1;
    s->uploaded = false;
    return PBRT_HIP_OK;
}
}  // namespace
// generate_mipmap + MIPMap::new (mipmap/cache.rs:74-120, mipmap/mod.rs:115-189): flip in y, convert texels, resample to powers of
// two, box-filter the pyramid.  Texels are kept as three floats per texel while building.
int pbrt_hip_add_mipmap(PbrtHipScene* s, int width, int height, const float* rgb, int as_float, float scale, int gamma, int filtering, int wrap,
                        float max_anisotropy, uint32_t* out_id) {
    return ph_guard(s, "pbrt_hip_add_mipmap", [&]() -> int {
    if (!s || !rgb) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_mipmap: null argument");
    if (width <= 0 || height <= 0 || width > 32768 || height > 32768) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_mipmap: resolution must be within 1..32768");
    if (filtering < 0 || filtering > 1 || wrap < 0 || wrap > 2) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_mipmap: unknown filtering / wrap mode");
    size_t w = (size_t)width, h = (size_t)height;
    std::vector<float> img(3 * w * h);
    for (size_t y = 0; y < h; y++)
        for (size_t x = 0; x < w; x++) {
            const float* px = rgb + 3 * ((h - 1 - y) * w + x);  // texture space has (0,0) at the lower left
            float* o = &img[3 * (y * w + x)];
            if (as_float) {  // ConvertIn<Float> for RGBSpectrum (convert_in.rs:37-47)
                const float lum = 0.212671f * px[0] + 0.715160f * px[1] + 0.072169f * px[2];
                o[0] = o[1] = o[2] = scale * (gamma ? inv_gamma_correct(lum) : lum);
            } else for (int c = 0; c < 3; c++) o[c] = scale * (gamma ? inv_gamma_correct(px[c]) : px[c]);
        }
    return build_pyramid(s, std::move(img), w, h, filtering, wrap, as_float != 0, max_anisotropy, out_id);
    });
}
int pbrt_hip_add_texture_constant(PbrtHipScene* s, const float v[3], uint32_t* out_id) {
    return ph_guard(s, "pbrt_hip_add_texture_constant", [&]() -> int {
    if (!s || !v) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_texture_constant: null argument");
    PbrtHipScene::TextureHost t; TexOp op{}; op.op = PH_TOP_CONST; std::memcpy(op.c, v, 12); t.prog.push_back(op);
    return push_texture(s, std::move(t), out_id);
    });
}
int pbrt_hip_add_texture_imagemap(PbrtHipScene* s, uint32_t mipmap, float su, float sv, float du, float dv, uint32_t* out_id) {
    return ph_guard(s, "pbrt_hip_add_texture_imagemap", [&]() -> int {
    if (!s || mipmap >= s->mipmaps.size()) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_texture_imagemap: unknown mipmap");
    PbrtHipScene::TextureHost t; TexOp op{}; op.op = PH_TOP_IMAGE; op.mip = mipmap; op.su = su; op.sv = sv; op.du = du; op.dv = dv; t.prog.push_back(op);
    return push_texture(s, std::move(t), out_id);
    });
}
int pbrt_hip_add_texture_scale(PbrtHipScene* s, uint32_t t1, uint32_t t2, uint32_t* out_id) {
    return ph_guard(s, "pbrt_hip_add_texture_scale", [&]() -> int {
    if (!s || t1 >= s->textures.size() || t2 >= s->textures.size()) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_texture_scale: unknown texture");
    PbrtHipScene::TextureHost t; const PbrtHipScene::TextureHost &a = s->textures[t1], &b = s->textures[t2];
    t.prog = a.prog; t.prog.insert(t.prog.end(), b.prog.begin(), b.prog.end());
    TexOp op{}; op.op = PH_TOP_MUL; t.prog.push_back(op);
    t.stack_need = std::max(a.stack_need, 1 + b.stack_need);
    return push_texture(s, std::move(t), out_id);
    });
}
int pbrt_hip_add_texture_mix(PbrtHipScene* s, uint32_t t1, uint32_t t2, uint32_t amount, uint32_t* out_id) {
    return ph_guard(s, "pbrt_hip_add_texture_mix", [&]() -> int {
    if (!s || t1 >= s->textures.size() || t2 >= s->textures.size() || amount >= s->textures.size())
        return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_texture_mix: unknown texture");
    PbrtHipScene::TextureHost t; const PbrtHipScene::TextureHost &a = s->textures[t1], &b = s->textures[t2], &c = s->textures[amount];
    t.prog = a.prog; t.prog.insert(t.prog.end(), b.prog.begin(), b.prog.end()); t.prog.insert(t.prog.end(), c.prog.begin(), c.prog.end());
    TexOp op{}; op.op = PH_TOP_MIX; t.prog.push_back(op);
    t.stack_need = std::max(a.stack_need, std::max(1 + b.stack_need, 2 + c.stack_need));
    return push_texture(s, std::move(t), out_id);
    });
}
int pbrt_hip_add_texture_checkerboard(PbrtHipScene* s, uint32_t t1, uint32_t t2, float su, float sv, float du, float dv, int aa_mode, uint32_t* out_id) {
    return ph_guard(s, "pbrt_hip_add_texture_checkerboard", [&]() -> int {
    if (!s || t1 >= s->textures.size() || t2 >= s->textures.size()) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_texture_checkerboard: unknown texture");
    PbrtHipScene::TextureHost t; const PbrtHipScene::TextureHost &a = s->textures[t1], &b = s->textures[t2];
    t.prog = a.prog; t.prog.insert(t.prog.end(), b.prog.begin(), b.prog.end());
    TexOp op{}; op.op = PH_TOP_CHECKER; op.mip = aa_mode ? 1u : 0u; op.su = su; op.sv = sv; op.du = du; op.dv = dv; t.prog.push_back(op);
    t.stack_need = std::max(a.stack_need, 1 + b.stack_need);
    return push_texture(s, std::move(t), out_id);
    });
}
int pbrt_hip_add_texture_uv(PbrtHipScene* s, float su, float sv, float du, float dv, uint32_t* out_id) {
    return ph_guard(s, "pbrt_hip_add_texture_uv", [&]() -> int {
    if (!s) return PBRT_HIP_ERR_INVALID_ARG;
    PbrtHipScene::TextureHost t; TexOp op{}; op.op = PH_TOP_UV; op.su = su; op.sv = sv; op.du = du; op.dv = dv; t.prog.push_back(op);
    return push_texture(s, std::move(t), out_id);
    });
}
int pbrt_hip_add_texture_bilerp(PbrtHipScene* s, const float v00[3], const float v01[3], const float v10[3], const float v11[3], float su, float sv, float du, float dv,
                                uint32_t* out_id) {
    return ph_guard(s, "pbrt_hip_add_texture_bilerp", [&]() -> int {
    if (!s || !v00 || !v01 || !v10 || !v11) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_texture_bilerp: null argument");
    PbrtHipScene::TextureHost t;
    for (const float* v : {v00, v01, v10, v11}) { TexOp c{}; c.op = PH_TOP_CONST; std::memcpy(c.c, v, 12); t.prog.push_back(c); }
    TexOp op{}; op.op = PH_TOP_BILERP; op.su = su; op.sv = sv; op.du = du; op.dv = dv; t.prog.push_back(op);
    t.stack_need = 4;
    return push_texture(s, std::move(t), out_id);
    });
}
int pbrt_hip_add_texture_dots(PbrtHipScene* s, uint32_t inside, uint32_t outside, float su, float sv, float du, float dv, uint32_t* out_id) {
    return ph_guard(s, "pbrt_hip_add_texture_dots", [&]() -> int {
    if (!s || inside >= s->textures.size() || outside >= s->textures.size()) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_texture_dots: unknown texture");
    PbrtHipScene::TextureHost t; const PbrtHipScene::TextureHost &a = s->textures[inside], &b = s->textures[outside];
    t.prog = a.prog; t.prog.insert(t.prog.end(), b.prog.begin(), b.prog.end());
    TexOp op{}; op.op = PH_TOP_DOTS; op.su = su; op.sv = sv; op.du = du; op.dv = dv; t.prog.push_back(op);
    t.stack_need = std::max(a.stack_need, 1 + b.stack_need);
    return push_texture(s, std::move(t), out_id);
    });
}
namespace {
int push_texture3d(PbrtHipScene* s, uint32_t opc, const float m[16], float omega, int octaves, float scale, float variation, const char* what, uint32_t* out_id,
                   const PbrtHipScene::TextureHost* a = nullptr, const PbrtHipScene::TextureHost* b = nullptr) {
    if (!s || !m) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, std::string(what) + ": null argument");
    if (octaves < 0 || octaves > 64) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, std::string(what) + ": octaves must be within 0..64");
    PbrtHipScene::TextureHost t;
    if (a) { t.prog = a->prog; t.prog.insert(t.prog.end(), b->prog.begin(), b->prog.end()); t.stack_need = std::max(a->stack_need, 1 + b->stack_need); }
    TexOp op{}; op.op = opc; op.octaves = (uint32_t)octaves; op.omega = omega; op.scale = scale; op.variation = variation; std::memcpy(op.m, m, 64);
    t.prog.push_back(op);
    return push_texture(s, std::move(t), out_id);
}
}  // namespace
int pbrt_hip_add_texture_fbm(PbrtHipScene* s, const float m[16], float omega, int octaves, uint32_t* out_id) { return ph_guard(s, "pbrt_hip_add_texture_fbm", [&]() -> int { return push_texture3d(s, PH_TOP_FBM, m, omega, octaves, 1.0f, 0.0f, "add_texture_fbm", out_id); }); }
int pbrt_hip_add_texture_wrinkled(PbrtHipScene* s, const float m[16], float omega, int octaves, uint32_t* out_id) { return ph_guard(s, "pbrt_hip_add_texture_wrinkled", [&]() -> int { return push_texture3d(s, PH_TOP_WRINKLED, m, omega, octaves, 1.0f, 0.0f, "add_texture_wrinkled", out_id); }); }
int pbrt_hip_add_texture_windy(PbrtHipScene* s, const float m[16], uint32_t* out_id) { return ph_guard(s, "pbrt_hip_add_texture_windy", [&]() -> int { return push_texture3d(s, PH_TOP_WINDY, m, 0.5f, 0, 1.0f, 0.0f, "add_texture_windy", out_id); }); }
int pbrt_hip_add_texture_marble(PbrtHipScene* s, const float m[16], float omega, int octaves, float scale, float variation, uint32_t* out_id) {
    return ph_guard(s, "pbrt_hip_add_texture_marble", [&]() -> int {
    return push_texture3d(s, PH_TOP_MARBLE, m, omega, octaves, scale, variation, "add_texture_marble", out_id);
    });
}
int pbrt_hip_add_texture_checkerboard3d(PbrtHipScene* s, uint32_t t1, uint32_t t2, const float m[16], uint32_t* out_id) {
    return ph_guard(s, "pbrt_hip_add_texture_checkerboard3d", [&]() -> int {
    if (!s || t1 >= s->textures.size() || t2 >= s->textures.size()) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_texture_checkerboard3d: unknown texture");
    const PbrtHipScene::TextureHost a = s->textures[t1], b = s->textures[t2];
    return push_texture3d(s, PH_TOP_CHECKER3D, m, 0.0f, 0, 1.0f, 0.0f, "add_texture_checkerboard3d", out_id, &a, &b);
    });
}
// TextureMapping2D other than uv for a 2D texture (imagemap, checkerboard, uv, bilerp, dots).  Programs are copied into their parents when those
// are created, so the mapping has to be set before the texture is used as an operand.
int pbrt_hip_set_texture_mapping(PbrtHipScene* s, uint32_t texture, int kind, const float* params) {
    return ph_guard(s, "pbrt_hip_set_texture_mapping", [&]() -> int {
    if (!s || texture >= s->textures.size() || !params) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "set_texture_mapping: bad argument");
    if (kind < 1 || kind > 3) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "set_texture_mapping: kind must be 1 spherical, 2 cylindrical or 3 planar");
    TexOp& op = s->textures[texture].prog.back();
    if (!(op.op == PH_TOP_IMAGE || op.op == PH_TOP_CHECKER || op.op == PH_TOP_UV || op.op == PH_TOP_BILERP || op.op == PH_TOP_DOTS))
        return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "set_texture_mapping: only imagemap, checkerboard (2D), uv, bilerp and dots textures take a 2D mapping");
    op.mapping = (uint32_t)kind;
    if (kind == 3) { std::memcpy(op.m, params, 24); op.du = params[6]; op.dv = params[7]; }
    else std::memcpy(op.m, params, 64);
    s->uploaded = false;
    return PBRT_HIP_OK;
    });
}
// Replaces a material's constant colour parameter by a texture evaluated at every hit.  The material must have been created with a non-black
// constant for that parameter (so that its lobe exists); which lobes a hit finally gets follows the reference's `is_black` tests on the
// texture's value at that hit.
int pbrt_hip_set_material_texture(PbrtHipScene* s, uint32_t material, int param, uint32_t texture) {
    return ph_guard(s, "pbrt_hip_set_material_texture", [&]() -> int {
    if (!s || material >= s->materials.size() || texture >= s->textures.size()) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "set_material_texture: unknown material or texture");
    if (param < 0 || param > 9) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "set_material_texture: param must be PBRT_HIP_PARAM_KD / KS / KR / KT / OPACITY / AMOUNT / ETA / K / REFLECT / TRANSMIT");
    if (param >= PBRT_HIP_PARAM_OPACITY) {   // parameters that are not one lobe's colour
        PbrtHipScene::MaterialParams& mq = s->material_params[material];
        if (param == PBRT_HIP_PARAM_REFLECT || param == PBRT_HIP_PARAM_TRANSMIT) {   // translucent.rs:70-71
            if (mq.made_as != 5) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "set_material_texture: reflect / transmit belong to TranslucentMaterial");
            const int rc = translucent_rebuild_rt(s, material);
            if (rc) return rc;
            MaterialRec& mt = s->materials[material];
            (param == PBRT_HIP_PARAM_REFLECT ? mt.refl_tex1 : mt.trans_tex1) = texture + 1u;
            s->uploaded = false;
            return PBRT_HIP_OK;
        }
        MaterialRec& mm = s->materials[material];
        if (param == PBRT_HIP_PARAM_OPACITY) {      // uber.rs:126-160
            if (mq.made_as != 1) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "set_material_texture: opacity belongs to UberMaterial");
            const int rc = uber_rebuild_for_opacity(s, material);
            if (rc) return rc;
            mm.opacity_tex1 = texture + 1u;
        } else if (param == PBRT_HIP_PARAM_AMOUNT) {  // mix.rs:59-60
            if (mq.made_as != 4) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "set_material_texture: amount belongs to MixMaterial");
            uint32_t cols = 1;
            for (uint32_t i = 0; i < mm.n_lobes; i++) {
                LobeRec& l = s->lobes[mm.lobe_base + i];
                l.amt = ((int)i < mq.mix_n1 ? 1u : 2u) | ((l.n_scale - 1u) << 8);
                cols += (l.r_tex1 || (l.has_pre == PH_PRE_OPACITY && l.kind != PH_LK_SPEC_T) ? 1u : 0u) + (l.t_tex1 || (l.has_pre == PH_PRE_OPACITY && l.kind == PH_LK_SPEC_T) || l.has_pre == PH_PRE_PASSTHROUGH ? 1u : 0u) +
                        (l.eta_tex1 ? 1u : 0u) + (l.k_tex1 ? 1u : 0u);
            }
            if (cols > PH_HIT_COLS || mm.n_lobes > PH_HIT_LOBES) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "set_material_texture: a mix with a textured amount may have at most 5 more per-hit colours");
            mm.amount_tex1 = texture + 1u;
        } else {                                     // metal.rs:121-125
            if (mq.made_as != 3) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "set_material_texture: eta / k belong to MetalMaterial");
            LobeRec& l = s->lobes[mm.lobe_base];
            (param == PBRT_HIP_PARAM_ETA ? l.eta_tex1 : l.k_tex1) = texture + 1u;
        }
        mm.textured = 1u;
        s->textured_materials = true;
        s->uploaded = false;
        return PBRT_HIP_OK;
    }
    const PbrtHipScene::MaterialParams& mp = s->material_params[material];
    if (mp.lobe[param] < 0)
        return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "set_material_texture: this material has no lobe fed by that parameter (matte Kd, plastic Kd / Ks, mirror Kr, substrate Kd / Ks, "
                                                   "glass Kr / Kt, uber Kd / Ks / Kr / Kt and translucent Kd / Ks take textures; create the material with a non-black placeholder for the parameter)");
    MaterialRec& m = s->materials[material];
    const int fed[2] = {mp.lobe[param], mp.lobe2[param]}, fields[2] = {mp.field[param], mp.field2[param]};
    for (int k = 0; k < 2; k++) if (fed[k] >= 0) {
        LobeRec& l = s->lobes[m.lobe_base + (uint32_t)fed[k]];
        if (fields[k] == 0) l.r_tex1 = texture + 1u; else l.t_tex1 = texture + 1u;
        if (mp.has_pre && l.has_pre < PH_PRE_OPACITY) { l.has_pre = 1u; std::memcpy(l.pre, mp.pre, 12); }
        if ((l.kind == PH_LK_LAMBERT || l.kind == PH_LK_OREN) && m.n_lobes == 1u && l.has_pre == 0u) m.kd_tex1 = texture + 1u;  // MatteMaterial: the one-lobe kernel reads kd_tex1
    }
    m.textured = 1u;
    s->textured_materials = true;
    s->uploaded = false;
    return PBRT_HIP_OK;
    });
}
// Replaces a scalar parameter by a float texture evaluated at every hit: MatteMaterial's sigma (matte.rs:64-70) or the Trowbridge-Reitz roughness of plastic / uber /
// substrate / metal (remapped per hit when the material was created with remap_roughness).  fparam: 0 sigma, 1 uroughness, 2 vroughness (plastic's single `roughness`: set both).
int pbrt_hip_set_material_float_texture(PbrtHipScene* s, uint32_t material, int fparam, uint32_t texture) {
    return ph_guard(s, "pbrt_hip_set_material_float_texture", [&]() -> int {
    if (!s || material >= s->materials.size() || texture >= s->textures.size()) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "set_material_float_texture: unknown material or texture");
    if (fparam < 0 || fparam > 3) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "set_material_float_texture: fparam must be 0 sigma, 1 uroughness, 2 vroughness or 3 index");
    if (fparam == 3) {   // `let eta = self.index.evaluate(..)` (glass.rs:102) / `let e = self.index.evaluate(..)` (uber.rs:128): every dielectric lobe of the hit takes it
        const int made_as = s->material_params[material].made_as;
        if (made_as != 1 && made_as != 2) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "set_material_float_texture: index belongs to GlassMaterial and UberMaterial");
        if (made_as == 1 && !s->materials[material].opacity_tex1) {
            // UberMaterial decides BSDF::eta and the pass-through lobe per hit once anything about them is per hit: the constant opacity becomes a constant texture (same values at every hit)
            const PbrtHipScene::MaterialParams mq = s->material_params[material];
            const float one[3] = {1.0f, 1.0f, 1.0f};
            uint32_t op_tex = 0;
            int rc = pbrt_hip_add_texture_constant(s, mq.has_pre ? mq.pre : one, &op_tex);
            if (rc == PBRT_HIP_OK) rc = pbrt_hip_set_material_texture(s, material, PBRT_HIP_PARAM_OPACITY, op_tex);
            if (rc) return rc;
        }
        MaterialRec& mi = s->materials[material];
        if (mi.n_lobes == 0) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "set_material_float_texture: this material has no lobe (black Kr and Kt)");
        mi.index_tex1 = texture + 1u; mi.textured = 1u;
        s->textured_materials = true; s->general_materials = true;
        s->uploaded = false;
        return PBRT_HIP_OK;
    }
    MaterialRec& m = s->materials[material];
    const PbrtHipScene::MaterialParams& mp = s->material_params[material];
    if (fparam == 0) {
        if (m.none || m.n_lobes != 1u || !(s->lobes[m.lobe_base].kind == PH_LK_LAMBERT || s->lobes[m.lobe_base].kind == PH_LK_OREN))
            return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "set_material_float_texture: sigma belongs to MatteMaterial (created with a non-black Kd)");
        s->lobes[m.lobe_base].sigma_tex1 = texture + 1u; m.sigma_tex1 = texture + 1u;
    } else if (mp.made_as == 2) {   // glass: both alternatives carry the textures (the smooth one for its `== 0` test), glass.rs:110-141
        const int rc = glass_rebuild_for_roughness(s, material);
        if (rc) return rc;
        MaterialRec& mg = s->materials[material];
        if (mg.n_lobes == 0) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "set_material_float_texture: this glass has neither reflection nor transmission");
        for (uint32_t i = 0; i < mg.n_lobes; i++) { LobeRec& l = s->lobes[mg.lobe_base + i]; (fparam == 1 ? l.ax_tex1 : l.ay_tex1) = texture + 1u; l.remap = s->material_params[material].raw_remap ? 1u : 0u; }
    } else {
        if (mp.rough_lobe < 0) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "set_material_float_texture: this material has no microfacet lobe whose roughness could be textured (plastic, uber, substrate, translucent and metal have; glass switches lobes on roughness == 0 and is not wired)");
        const int rl[2] = {mp.rough_lobe, mp.rough_lobe2};
        for (int k = 0; k < 2; k++) if (rl[k] >= 0) { LobeRec& l = s->lobes[m.lobe_base + (uint32_t)rl[k]]; (fparam == 1 ? l.ax_tex1 : l.ay_tex1) = texture + 1u; l.remap = mp.rough_remap ? 1u : 0u; }
    }
    m.textured = 1u;
    s->textured_materials = true;
    s->uploaded = false;
    return PBRT_HIP_OK;
    });
}
// Material::bump's displacement texture (core/src/material.rs:62-101; the `bumpmap` parameter every material takes)
int pbrt_hip_set_material_bump(PbrtHipScene* s, uint32_t material, uint32_t texture) {
    return ph_guard(s, "pbrt_hip_set_material_bump", [&]() -> int {
    if (!s || material >= s->materials.size() || texture >= s->textures.size()) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "set_material_bump: unknown material or texture");
    if (s->materials[material].none) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "set_material_bump: Material \"none\" has no BSDF to bump");
    s->materials[material].bump_tex1 = texture + 1u;
    s->textured_materials = true; s->bump_materials = true;
    s->uploaded = false;
    return PBRT_HIP_OK;
    });
}
int pbrt_hip_add_material_matte_tex(PbrtHipScene* s, uint32_t kd_tex, float sigma_deg, uint32_t* out_id) {  // matte.rs:47-76, Kd a texture
    return ph_guard(s, "pbrt_hip_add_material_matte_tex", [&]() -> int {
    if (!s || kd_tex >= s->textures.size()) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_material_matte_tex: unknown texture");
    const float one[3] = {1.0f, 1.0f, 1.0f};
    uint32_t id = 0;
    int rc = pbrt_hip_add_material_matte(s, one, sigma_deg, &id);
    if (rc == PBRT_HIP_OK) rc = pbrt_hip_set_material_texture(s, id, 0, kd_tex);
    if (rc == PBRT_HIP_OK && out_id) *out_id = id;
    return rc;
    });
}
// ---- texture probes (test aids: the device's texture evaluation on explicit inputs, and the pyramid the host built) ----------------
namespace ph {
__global__ void texture_eval_kernel(DeviceScene sc, uint32_t tex, uint32_t n, const float* in, float* out, uint32_t nodiff) {
    noise_lds_fill();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* q = in + 15 * (size_t)i;
    TexCtx c; c.uv = mk2(q[0], q[1]); c.dudx = q[2]; c.dvdx = q[3]; c.dudy = q[4]; c.dvdy = q[5];
    c.p = mk3(q[6], q[7], q[8]); c.dpdx = mk3(q[9], q[10], q[11]); c.dpdy = mk3(q[12], q[13], q[14]);
    const spec v = nodiff ? tex_eval<false, true>(sc.self, tex, c) : tex_eval<false, false>(sc.self, tex, c);
    out[3 * i] = v.r; out[3 * i + 1] = v.g; out[3 * i + 2] = v.b;
}
}  // namespace ph
static int texture_eval_batch(PbrtHipScene* s, uint32_t tex, uint64_t n, const float* uv_and_derivatives, float* out_rgb, bool nodiff);
int pbrt_hip_texture_eval_batch(PbrtHipScene* s, uint32_t tex, uint64_t n, const float* uv_and_derivatives, float* out_rgb) {
    return ph_guard(s, "pbrt_hip_texture_eval_batch", [&]() -> int { return texture_eval_batch(s, tex, n, uv_and_derivatives, out_rgb, false); });
}
int pbrt_hip_texture_eval_batch_nodiff(PbrtHipScene* s, uint32_t tex, uint64_t n, const float* uv_and_derivatives, float* out_rgb) {
    return ph_guard(s, "pbrt_hip_texture_eval_batch_nodiff", [&]() -> int { return texture_eval_batch(s, tex, n, uv_and_derivatives, out_rgb, true); });
}
static int texture_eval_batch(PbrtHipScene* s, uint32_t tex, uint64_t n, const float* uv_and_derivatives, float* out_rgb, bool nodiff) {
    {
    if (!s || (n && (!uv_and_derivatives || !out_rgb))) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "texture_eval_batch: null argument");
    if (tex >= s->textures.size()) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "texture_eval_batch: unknown texture");
    if (n == 0) return PBRT_HIP_OK;
    if (n > 0xFFFFFFFFull) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "texture_eval_batch: too many points");
    PH_CHECK(s, hipSetDevice(s->device));
    // textures do not need the accelerator: upload what exists (an empty BVH is fine)
    int rc;
    if ((rc = upload_scene(s))) return rc;
    if ((rc = ensure_buf(s, s->d_rays_tmp, n * 60))) return rc;
    if ((rc = ensure_buf(s, s->d_out_tmp, n * 12))) return rc;
    PH_CHECK(s, hipMemcpyAsync(s->d_rays_tmp.p, uv_and_derivatives, n * 60, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(ph::texture_eval_kernel, dim3((uint32_t)((n + 127) / 128)), dim3(128), 0, s->stream, s->ds, tex, (uint32_t)n, (const float*)s->d_rays_tmp.p, (float*)s->d_out_tmp.p, nodiff ? 1u : 0u);
    PH_CHECK(s, hipGetLastError());
    PH_CHECK(s, hipMemcpyAsync(out_rgb, s->d_out_tmp.p, n * 12, hipMemcpyDeviceToHost, s->stream));
    PH_CHECK(s, hipStreamSynchronize(s->stream));
    return PBRT_HIP_OK;
    }
}
int pbrt_hip_mipmap_levels(PbrtHipScene* s, uint32_t mip, int* out_levels, int* out_wh) {
    return ph_guard(s, "pbrt_hip_mipmap_levels", [&]() -> int {
    if (!s || !out_levels || !out_wh || mip >= s->mipmaps.size()) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "mipmap_levels: bad argument");
    const MipRec& m = s->mipmaps[mip];
    *out_levels = (int)m.n_levels;
    for (uint32_t i = 0; i < m.n_levels; i++) { out_wh[2 * i] = (int)m.level_w[i]; out_wh[2 * i + 1] = (int)m.level_h[i]; }
    return PBRT_HIP_OK;
    });
}
int pbrt_hip_mipmap_level_texels(PbrtHipScene* s, uint32_t mip, int level, float* out_rgb) {
    return ph_guard(s, "pbrt_hip_mipmap_level_texels", [&]() -> int {
    if (!s || !out_rgb || mip >= s->mipmaps.size() || level < 0 || (uint32_t)level >= s->mipmaps[mip].n_levels)
        return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "mipmap_level_texels: bad argument");
    const MipRec& m = s->mipmaps[mip];
    const size_t n = (size_t)m.level_w[level] * m.level_h[level];
    for (size_t i = 0; i < n; i++) { const Texel& t = s->texels[m.level_off[level] + i]; out_rgb[3 * i] = t.r; out_rgb[3 * i + 1] = t.g; out_rgb[3 * i + 2] = t.b; }
    return PBRT_HIP_OK;
    });
}
// `alpha` / `shadowalpha` float textures of the mesh added last (TriangleMesh::alpha_mask / shadow_alpha_mask, shapes/src/triangle.rs:291-312); 0xFFFFFFFF keeps the
// constant given to add_mesh.  Candidate hits are then tested in the traversal kernels' ALPHA variants.
int pbrt_hip_set_last_mesh_alpha_textures(PbrtHipScene* s, uint32_t alpha_tex, uint32_t shadow_alpha_tex) {
    return ph_guard(s, "pbrt_hip_set_last_mesh_alpha_textures", [&]() -> int {
    if (!s || s->meshes.empty()) return set_err(s, PBRT_HIP_ERR_STATE, "set_last_mesh_alpha_textures: no mesh has been added");
    if ((alpha_tex != 0xFFFFFFFFu && alpha_tex >= s->textures.size()) || (shadow_alpha_tex != 0xFFFFFFFFu && shadow_alpha_tex >= s->textures.size()))
        return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "set_last_mesh_alpha_textures: unknown texture");
    MeshRec& m = s->meshes.back();
    if (alpha_tex != 0xFFFFFFFFu) m.alpha_tex1 = alpha_tex + 1u;
    if (shadow_alpha_tex != 0xFFFFFFFFu) m.shadow_alpha_tex1 = shadow_alpha_tex + 1u;
    for (uint32_t t = m.tri_base; t < m.tri_base + m.n_tris; t++) {
        uint32_t& tf = s->tri_flags[t];
        if (m.alpha_tex1) tf &= ~PH_TRI_ALPHA0;              // the texture replaces the constant
        if (m.shadow_alpha_tex1) tf &= ~PH_TRI_SALPHA0;
        if (m.alpha_tex1 || m.shadow_alpha_tex1) tf |= PH_TRI_ALPHATEX;
    }
    if (m.alpha_tex1 || m.shadow_alpha_tex1) s->alpha_textures = true;
    s->built = false; s->uploaded = false;
    return PBRT_HIP_OK;
    });
}

// InfiniteAreaLight::new with a texmap (lights/src/infinite.rs:52-100): texels = read_image(texmap) * L — no y flip on this path —, MIPMap (EWA, repeat, 8),
// compute_scalar_image (:326-369) over 2w x 2h and its Distribution2D.
int pbrt_hip_add_light_infinite_map(PbrtHipScene* s, const float L[3], int width, int height, const float* rgb, const float l2w[16], const float w2l[16]) {
    return ph_guard(s, "pbrt_hip_add_light_infinite_map", [&]() -> int {
    if (!s || !L || !rgb || !l2w || !w2l) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_light_infinite_map: null argument");
    if (width <= 0 || height <= 0 || width > 16384 || height > 16384) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_light_infinite_map: resolution must be within 1..16384");
    const size_t lights_before = s->lights.size();
    int rc = pbrt_hip_add_light_infinite(s, L, l2w, w2l);   // fills the common fields (and the constant-light tables, unused with a map)
    if (rc) return rc;
    std::vector<float> img(3 * (size_t)width * (size_t)height);
    for (size_t i = 0; i < (size_t)width * (size_t)height; i++) for (int c = 0; c < 3; c++) img[3 * i + c] = rgb[3 * i + c] * L[c];
    uint32_t mip = 0;
    if ((rc = build_pyramid(s, std::move(img), (size_t)width, (size_t)height, 1, 0, false, 8.0f, &mip))) { s->lights.pop_back(); s->infinite_lights.pop_back(); return rc; }
    const MipRec& m = s->mipmaps[mip];
    const size_t dw = 2 * (size_t)m.level_w[0], dh = 2 * (size_t)m.level_h[0];
    const float fwidth = 0.5f / (float)(dw < dh ? dw : dh);
    LightRec& l = s->lights[lights_before];
    l.map_mip1 = mip + 1u; l.dw = (uint32_t)dw; l.dh = (uint32_t)dh; l.dist_off = (uint32_t)s->light_dist.size();
    s->textured_materials = true;   // the radiance-map code lives in the TEX instantiations of the shade kernels
    std::vector<float> cond_func(dw * dh), cond_cdf((dw + 1) * dh), cond_int(dh), row(dw), cdf;
    for (size_t v = 0; v < dh; v++) {
        const float vp = ((float)v + 0.5f) / (float)dh;
        const float sin_theta = std::sin(hm::kPi * ((float)v + 0.5f) / (float)dh);
        for (size_t u = 0; u < dw; u++) {
            const float up = ((float)u + 0.5f) / (float)dw;
            const Rgb t = hmip_lookup_triangle_rgb(s, m, up, vp, fwidth);
            row[u] = (0.212671f * t.c[0] + 0.715160f * t.c[1] + 0.072169f * t.c[2]) * sin_theta;
        }
        float fi;
        hm::distribution1d(row, cdf, fi);
        std::copy(row.begin(), row.end(), cond_func.begin() + (long)(v * dw));
        std::copy(cdf.begin(), cdf.end(), cond_cdf.begin() + (long)(v * (dw + 1)));
        cond_int[v] = fi;
    }
    float mi;
    hm::distribution1d(cond_int, cdf, mi);
    std::vector<float>& pool = s->light_dist;
    pool.insert(pool.end(), cond_func.begin(), cond_func.end());
    pool.insert(pool.end(), cond_cdf.begin(), cond_cdf.end());
    pool.insert(pool.end(), cond_int.begin(), cond_int.end());
    pool.insert(pool.end(), cond_int.begin(), cond_int.end());   // the marginal's func = the rows' integrals
    pool.insert(pool.end(), cdf.begin(), cdf.end());
    pool.push_back(mi);
    s->uploaded = false;
    return PBRT_HIP_OK;
    });
}

namespace {
// the part ProjectionLight::new and GonioPhotometricLight::new share: p_light, world_to_light and the image's MIPMap (Ewa, Repeat, 8.0; projection.rs:70-80, goniometric.rs:49-58)
int add_image_point_light(PbrtHipScene* s, int type, const float I[3], const float l2w[16], const float w2l[16], int width, int height, const float* rgb, LightRec* out) {
    if (rgb && (width <= 0 || height <= 0 || width > 16384 || height > 16384)) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "image light: resolution must be within 1..16384");
    LightRec l{}; l.type = type; l.prim = 0xFFFFFFFFu;
    for (int c = 0; c < 3; c++) l.L[c] = I[c];
    std::memcpy(l.l2w, l2w, 48); std::memcpy(l.w2l, w2l, 48);
    // p_light = light_to_world.transform_point(Point3f::ZERO) (transform.rs:288-302)
    const float xp = l2w[0] * 0.0f + l2w[1] * 0.0f + l2w[2] * 0.0f + l2w[3], yp = l2w[4] * 0.0f + l2w[5] * 0.0f + l2w[6] * 0.0f + l2w[7];
    const float zp = l2w[8] * 0.0f + l2w[9] * 0.0f + l2w[10] * 0.0f + l2w[11], wp = l2w[12] * 0.0f + l2w[13] * 0.0f + l2w[14] * 0.0f + l2w[15];
    if (wp == 1.0f) { l.v[0] = xp; l.v[1] = yp; l.v[2] = zp; } else { const float inv = 1.0f / wp; l.v[0] = inv * xp; l.v[1] = inv * yp; l.v[2] = inv * zp; }
    if (rgb) {
        std::vector<float> img(rgb, rgb + 3 * (size_t)width * (size_t)height);
        uint32_t mip = 0;
        const int rc = build_pyramid(s, std::move(img), (size_t)width, (size_t)height, 1, 0, false, 8.0f, &mip);
        if (rc) return rc;
        l.map_mip1 = mip + 1u;
        s->textured_materials = true;   // the image lookup lives in the TEX instantiations of the shade kernels, like the infinite light's radiance map
    }
    *out = l;
    return PBRT_HIP_OK;
}
}  // namespace
int pbrt_hip_add_light_projection(PbrtHipScene* s, const float I[3], const float l2w[16], const float w2l[16], float fov_deg, int width, int height, const float* rgb) {  // projection.rs:55-127
    return ph_guard(s, "pbrt_hip_add_light_projection", [&]() -> int {
    if (!s || !I || !l2w || !w2l) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_light_projection: null argument");
    LightRec l;
    const int rc = add_image_point_light(s, PH_L_PROJECTION, I, l2w, w2l, width, height, rgb, &l);
    if (rc) return rc;
    projection_light_setup(fov_deg, rgb ? (float)width / (float)height : 1.0f, l.proj, l.screen, &l.cos_total_width);
    s->lights.push_back(l); s->uploaded = false;
    return PBRT_HIP_OK;
    });
}
int pbrt_hip_add_light_goniometric(PbrtHipScene* s, const float I[3], const float l2w[16], const float w2l[16], int width, int height, const float* rgb) {  // goniometric.rs:34-80
    return ph_guard(s, "pbrt_hip_add_light_goniometric", [&]() -> int {
    if (!s || !I || !l2w || !w2l) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_light_goniometric: null argument");
    LightRec l;
    const int rc = add_image_point_light(s, PH_L_GONIO, I, l2w, w2l, width, height, rgb, &l);
    if (rc) return rc;
    s->lights.push_back(l); s->uploaded = false;
    return PBRT_HIP_OK;
    });
}

}  // extern "C"
