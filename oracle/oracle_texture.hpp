// TEST INFRASTRUCTURE ONLY (see oracle_math.hpp).  CPU restatement of the reference's image textures:
//   MIPMap<T>::new / lookup / triangle / ewa / resample_image / resample_weights / texel   core/src/mipmap/mod.rs:115-608
//   generate_mipmap (y flip + convert_in)                                                  core/src/mipmap/cache.rs:74-120, convert_in.rs:20-47
//   ImageTexture / ScaleTexture / MixTexture / ConstantTexture evaluate                    textures/src/{imagemap,scale,mix,constant}.rs
//   UVMapping2D::map                                                                       core/src/texture/mapping/uv_2d.rs:52-60
//   lanczos                                                                                core/src/texture/common.rs:216-228
// Float-valued textures (ImageTexture<Float>, ...) are carried as Spec with three equal channels: every operation the reference applies
// to them is applied per channel here, except `sum / sum_wts` in ewa(), a true division for Float and a multiply by the reciprocal for
// RGBSpectrum (rgb_spectrum.rs:255-263) — `is_float` selects it.
#pragma once
#include "oracle_math.hpp"
#include <vector>

namespace orc {

inline Float o_log2(Float x) { return g_libm_mode ? (Float)std::log2((double)x) : std::log2(x); }

enum { TEX_WRAP_REPEAT = 0, TEX_WRAP_BLACK = 1, TEX_WRAP_CLAMP = 2 };
enum { TEX_FILTER_TRILINEAR = 0, TEX_FILTER_EWA = 1 };
static const int WEIGHT_LUT_SIZE = 128;  // mipmap/mod.rs:24

inline int64_t rem_i(int64_t a, int64_t b) { int64_t r = a - (a / b) * b; return r < 0 ? r + b : r; }  // pbrt/common.rs:116-126
inline int64_t f2isize(Float f) {  // `as isize`: saturating, NaN -> 0
    if (f != f) return 0;
    if (f >= 9223372036854775808.0f) return INT64_MAX;
    if (f <= -9223372036854775808.0f) return INT64_MIN;
    return (int64_t)f;
}
inline Float lanczos(Float x, Float tau) {  // texture/common.rs:216-228 (host-side f32 sin: libm's, like the reference's)
    x = std::fabs(x);
    if (x < 1e-5f) return 1.0f;
    if (x > 1.0f) return 0.0f;
    x *= PI;
    Float s = std::sin(x * tau) / (x * tau);
    Float l = std::sin(x) / x;
    return s * l;
}
inline Float inv_gamma_correct(Float v) {  // pbrt/common.rs:152-158
    if (v <= 0.04045f) return v * 1.0f / 12.92f;
    return std::pow((v + 0.055f) * 1.0f / 1.055f, 2.4f);
}

struct ResampleWeight { size_t first_texel; Float weight[4]; };
inline std::vector<ResampleWeight> resample_weights(size_t old_res, size_t new_res) {  // mipmap/mod.rs:535-560
    std::vector<ResampleWeight> wt(new_res);
    const Float filterwidth = 2.0f;
    for (size_t i = 0; i < new_res; i++) {
        Float center = ((Float)i + 0.5f) * (Float)old_res / (Float)new_res;
        wt[i].first_texel = f2usize(std::floor((center - filterwidth) + 0.5f));  // `as usize`: negative saturates to 0 (a quirk: C++ pbrt keeps the sign)
        for (int j = 0; j < 4; j++) {
            Float pos = (Float)wt[i].first_texel + (Float)j + 0.5f;
            wt[i].weight[j] = lanczos((pos - center) / filterwidth, 2.0f);
        }
        Float inv_sum = 1.0f / (wt[i].weight[0] + wt[i].weight[1] + wt[i].weight[2] + wt[i].weight[3]);
        for (int j = 0; j < 4; j++) wt[i].weight[j] *= inv_sum;
    }
    return wt;
}

struct MipMap {
    int filtering = TEX_FILTER_EWA, wrap = TEX_WRAP_REPEAT;
    Float max_anisotropy = 8.0f;
    bool is_float = false;
    struct Level { int64_t w, h; std::vector<Spec> t; };
    std::vector<Level> pyr;
    Float weight_lut[WEIGHT_LUT_SIZE];

    Spec texel(size_t level, int64_t s, int64_t t) const {  // mipmap/mod.rs:569-608
        const Level& l = pyr[level];
        switch (wrap) {
            case TEX_WRAP_REPEAT: s = rem_i(s, l.w); t = rem_i(t, l.h); break;
            case TEX_WRAP_CLAMP: s = pclamp<int64_t>(s, 0, l.w - 1); t = pclamp<int64_t>(t, 0, l.h - 1); break;
            default: if (s < 0 || s >= l.w || t < 0 || t >= l.h) return Spec(0.0f);
        }
        return l.t[(size_t)t * (size_t)l.w + (size_t)s];
    }

    // `rgb`: w*h texels as read_image returns them (top row first).  generate_mipmap: flip in y, convert_in, MIPMap::new.
    void build(const Float* rgb, size_t w, size_t h, bool as_float, Float scale, bool gamma, int filtering_, int wrap_, Float max_aniso) {
        filtering = filtering_; wrap = wrap_; max_anisotropy = max_aniso; is_float = as_float;
        std::vector<Spec> img(w * h);
        for (size_t y = 0; y < h; y++)
            for (size_t x = 0; x < w; x++) {
                const Float* px = rgb + 3 * ((h - 1 - y) * w + x);
                Spec in(px[0], px[1], px[2]);
                if (as_float) { Float yv = in.y(); img[y * w + x] = Spec(scale * (gamma ? inv_gamma_correct(yv) : yv)); }
                else img[y * w + x] = Spec(scale * (gamma ? inv_gamma_correct(in.c[0]) : in.c[0]), scale * (gamma ? inv_gamma_correct(in.c[1]) : in.c[1]),
                                           scale * (gamma ? inv_gamma_correct(in.c[2]) : in.c[2]));
            }
        build_from(img, w, h);
    }
    // MIPMap::new on texels that are already in texture orientation (mipmap/mod.rs:115-189)
    void build_from(std::vector<Spec> img, size_t w, size_t h) {
        size_t rw = w, rh = h;
        auto pow2 = [](size_t v) { return v && !(v & (v - 1)); };
        if (!pow2(w) || !pow2(h)) {  // resample_image (mipmap/mod.rs:383-529)
            auto next_pow2 = [](size_t v) { size_t p = 1; while (p < v) p <<= 1; return p; };
            rw = next_pow2(w); rh = next_pow2(h);
            std::vector<Spec> r(rw * rh, Spec(0.0f));
            std::vector<ResampleWeight> sw = resample_weights(w, rw);
            for (size_t t = 0; t < h; t++)
                for (size_t s = 0; s < rw; s++) {
                    Spec pixel(0.0f);
                    for (int j = 0; j < 4; j++) {
                        size_t os = sw[s].first_texel + j;
                        if (wrap == TEX_WRAP_REPEAT) os = os % w;
                        else if (wrap == TEX_WRAP_CLAMP) os = pclamp<size_t>(os, 0, w - 1);
                        if (os < w) pixel += img[t * w + os] * sw[s].weight[j];
                    }
                    r[t * rw + s] += pixel;
                }
            std::vector<ResampleWeight> tw = resample_weights(h, rh);
            std::vector<Spec> work(rh);
            for (size_t s = 0; s < rw; s++) {
                for (size_t t = 0; t < rh; t++) {
                    work[t] = Spec(0.0f);
                    for (int j = 0; j < 4; j++) {
                        size_t off = tw[t].first_texel + j;
                        if (wrap == TEX_WRAP_REPEAT) off = off % h;
                        else if (wrap == TEX_WRAP_CLAMP) off = pclamp<size_t>(off, 0, h - 1);
                        if (off < h) work[t] += r[off * rw + s] * tw[t].weight[j];
                    }
                }
                for (size_t t = 0; t < rh; t++) r[t * rw + s] = spec_clamp0(work[t]);
            }
            img.swap(r);
        }
        size_t m = rw > rh ? rw : rh, n_levels = 1;
        while ((m >>= 1)) n_levels++;  // 1 + log2int(max(w, h))
        pyr.clear(); pyr.reserve(n_levels);
        pyr.push_back(Level{(int64_t)rw, (int64_t)rh, img});
        for (size_t i = 1; i < n_levels; i++) {
            int64_t sr = pmax<int64_t>(1, pyr[i - 1].w / 2), tr = pmax<int64_t>(1, pyr[i - 1].h / 2);
            Level l{sr, tr, std::vector<Spec>((size_t)(sr * tr))};
            for (int64_t t = 0; t < tr; t++)
                for (int64_t s = 0; s < sr; s++)
                    l.t[(size_t)(t * sr + s)] = (texel(i - 1, 2 * s, 2 * t) + texel(i - 1, 2 * s + 1, 2 * t) + texel(i - 1, 2 * s, 2 * t + 1) + texel(i - 1, 2 * s + 1, 2 * t + 1)) * 0.25f;
            pyr.push_back(std::move(l));
        }
        for (int i = 0; i < WEIGHT_LUT_SIZE; i++) {
            Float r2 = (Float)i / (Float)(WEIGHT_LUT_SIZE - 1);
            weight_lut[i] = std::exp(-2.0f * r2) - std::exp(-2.0f);
        }
    }

    // lookup_triangle at BUILD time (InfiniteAreaLight::new, infinite.rs:326-369 / :176-183): host libm log2f, like the reference
    Spec lookup_triangle_host(V2 st, Float width) const {
        const size_t levels = pyr.size();
        Float level = (Float)levels - 1.0f + std::log2(pmax(width, 1e-8f));
        if (level < 0.0f) return triangle(0, st);
        if (level >= (Float)(levels - 1)) return texel(levels - 1, 0, 0);
        size_t il = f2usize(std::floor(level));
        Float delta = level - (Float)il;
        return triangle(il, st) * (1.0f - delta) + triangle(il + 1, st) * delta;
    }
    Spec triangle(size_t level, V2 st) const {  // mipmap/mod.rs:293-312
        level = pclamp<size_t>(level, 0, pyr.size() - 1);
        Float s = st.x * (Float)pyr[level].w - 0.5f, t = st.y * (Float)pyr[level].h - 0.5f;
        int64_t s0 = f2isize(std::floor(s)), t0 = f2isize(std::floor(t));
        Float ds = s - (Float)s0, dt = t - (Float)t0;
        return texel(level, s0, t0) * (1.0f - ds) * (1.0f - dt) + texel(level, s0, t0 + 1) * (1.0f - ds) * dt + texel(level, s0 + 1, t0) * ds * (1.0f - dt) +
               texel(level, s0 + 1, t0 + 1) * ds * dt;
    }
    Spec ewa(size_t level, V2 st, V2 dst0, V2 dst1) const {  // mipmap/mod.rs:320-371
        if (level >= pyr.size()) return texel(pyr.size() - 1, 0, 0);
        Float us = (Float)pyr[level].w, vs = (Float)pyr[level].h;
        Float s = st.x * us - 0.5f, t = st.y * vs - 0.5f;
        Float d0x = dst0.x * us, d0y = dst0.y * vs, d1x = dst1.x * us, d1y = dst1.y * vs;
        Float a = d0y * d0y + d1y * d1y + 1.0f;
        Float b = -2.0f * (d0x * d0y + d1x * d1y);
        Float c = d0x * d0x + d1x * d1x + 1.0f;
        Float inv_f = 1.0f / (a * c - b * b * 0.25f);
        a *= inv_f; b *= inv_f; c *= inv_f;
        Float det = -b * b + 4.0f * a * c;
        Float inv_det = 1.0f / det;
        Float u_sqrt = std::sqrt(det * c), v_sqrt = std::sqrt(a * det);
        int64_t s0 = f2isize(std::ceil(s - 2.0f * inv_det * u_sqrt)), s1 = f2isize(std::floor(s + 2.0f * inv_det * u_sqrt));
        int64_t t0 = f2isize(std::ceil(t - 2.0f * inv_det * v_sqrt)), t1 = f2isize(std::floor(t + 2.0f * inv_det * v_sqrt));
        Spec sum(0.0f); Float sum_wts = 0.0f;
        for (int64_t it = t0; it <= t1; it++) {
            Float tt = (Float)it - t;
            for (int64_t is = s0; is <= s1; is++) {
                Float ss = (Float)is - s;
                Float r2 = a * ss * ss + b * ss * tt + c * tt * tt;
                if (r2 < 1.0f) {
                    size_t index = pmin<size_t>(f2usize(r2 * (Float)WEIGHT_LUT_SIZE), WEIGHT_LUT_SIZE - 1);
                    Float weight = weight_lut[index];
                    sum += texel(level, is, it) * weight;
                    sum_wts += weight;
                }
            }
        }
        if (is_float) return Spec(sum.c[0] / sum_wts);
        return sum / sum_wts;
    }
    Spec lookup(V2 st, V2 dst0, V2 dst1) const {  // mipmap/mod.rs:205-290
        const size_t levels = pyr.size();
        if (filtering == TEX_FILTER_TRILINEAR) {
            Float width = pmax(pmax(pabs(dst0.x), pabs(dst0.y)), pmax(pabs(dst1.x), pabs(dst1.y)));
            Float level = (Float)levels - 1.0f + o_log2(pmax(width, 1e-8f));
            if (level < 0.0f) return triangle(0, st);
            if (level >= (Float)(levels - 1)) return texel(levels - 1, 0, 0);
            size_t il = f2usize(std::floor(level));
            Float delta = level - (Float)il;
            return triangle(il, st) * (1.0f - delta) + triangle(il + 1, st) * delta;
        }
        if (dst0.x * dst0.x + dst0.y * dst0.y < dst1.x * dst1.x + dst1.y * dst1.y) { V2 tmp = dst0; dst0 = dst1; dst1 = tmp; }
        Float major_length = std::sqrt(dst0.x * dst0.x + dst0.y * dst0.y);
        Float minor_length = std::sqrt(dst1.x * dst1.x + dst1.y * dst1.y);
        Float adjusted = minor_length * max_anisotropy;
        if (adjusted < major_length && minor_length > 0.0f) {
            Float sc = major_length / adjusted;
            dst1.x *= sc; dst1.y *= sc;
            minor_length *= sc;
        }
        if (minor_length == 0.0f) return triangle(0, st);
        Float lod = pmax(0.0f, (Float)levels - 1.0f + o_log2(minor_length));
        size_t il = f2usize(std::floor(lod));
        Float t = lod - (Float)il;
        return ewa(il, st, dst0, dst1) * (1.0f - t) + ewa(il + 1, st, dst0, dst1) * t;
    }
};

// ---- Perlin noise (core/src/texture/common.rs:9-117).  The permutation is Ken Perlin's reference table, duplicated so that indices up to 511 work.
static const uint8_t NOISE_PERM[512] = {
    151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30, 69, 142, 8, 99, 37, 240, 21, 10, 23, 190, 6, 148, 247, 120, 234, 75, 0, 26,
    197, 62, 94, 252, 219, 203, 117, 35, 11, 32, 57, 177, 33, 88, 237, 149, 56, 87, 174, 20, 125, 136, 171, 168, 68, 175, 74, 165, 71, 134, 139, 48, 27, 166, 77, 146,
    158, 231, 83, 111, 229, 122, 60, 211, 133, 230, 220, 105, 92, 41, 55, 46, 245, 40, 244, 102, 143, 54, 65, 25, 63, 161, 1, 216, 80, 73, 209, 76, 132, 187, 208, 89,
    18, 169, 200, 196, 135, 130, 116, 188, 159, 86, 164, 100, 109, 198, 173, 186, 3, 64, 52, 217, 226, 250, 124, 123, 5, 202, 38, 147, 118, 126, 255, 82, 85, 212, 207,
    206, 59, 227, 47, 16, 58, 17, 182, 189, 28, 42, 223, 183, 170, 213, 119, 248, 152, 2, 44, 154, 163, 70, 221, 153, 101, 155, 167, 43, 172, 9, 129, 22, 39, 253, 19, 98,
    108, 110, 79, 113, 224, 232, 178, 185, 112, 104, 218, 246, 97, 228, 251, 34, 242, 193, 238, 210, 144, 12, 191, 179, 162, 241, 81, 51, 145, 235, 249, 14, 239, 107,
    49, 192, 214, 31, 181, 199, 106, 157, 184, 84, 204, 176, 115, 121, 50, 45, 127, 4, 150, 254, 138, 236, 205, 93, 222, 114, 67, 29, 24, 72, 243, 141, 128, 195, 78, 66,
    215, 61, 156, 180,
    151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30, 69, 142, 8, 99, 37, 240, 21, 10, 23, 190, 6, 148, 247, 120, 234, 75, 0, 26,
    197, 62, 94, 252, 219, 203, 117, 35, 11, 32, 57, 177, 33, 88, 237, 149, 56, 87, 174, 20, 125, 136, 171, 168, 68, 175, 74, 165, 71, 134, 139, 48, 27, 166, 77, 146,
    158, 231, 83, 111, 229, 122, 60, 211, 133, 230, 220, 105, 92, 41, 55, 46, 245, 40, 244, 102, 143, 54, 65, 25, 63, 161, 1, 216, 80, 73, 209, 76, 132, 187, 208, 89,
    18, 169, 200, 196, 135, 130, 116, 188, 159, 86, 164, 100, 109, 198, 173, 186, 3, 64, 52, 217, 226, 250, 124, 123, 5, 202, 38, 147, 118, 126, 255, 82, 85, 212, 207,
    206, 59, 227, 47, 16, 58, 17, 182, 189, 28, 42, 223, 183, 170, 213, 119, 248, 152, 2, 44, 154, 163, 70, 221, 153, 101, 155, 167, 43, 172, 9, 129, 22, 39, 253, 19, 98,
    108, 110, 79, 113, 224, 232, 178, 185, 112, 104, 218, 246, 97, 228, 251, 34, 242, 193, 238, 210, 144, 12, 191, 179, 162, 241, 81, 51, 145, 235, 249, 14, 239, 107,
    49, 192, 214, 31, 181, 199, 106, 157, 184, 84, 204, 176, 115, 121, 50, 45, 127, 4, 150, 254, 138, 236, 205, 93, 222, 114, 67, 29, 24, 72, 243, 141, 128, 195, 78, 66,
    215, 61, 156, 180};
inline Float noise_grad(int64_t x, int64_t y, int64_t z, Float dx, Float dy, Float dz) {
    int h = NOISE_PERM[NOISE_PERM[NOISE_PERM[x] + y] + z] & 15;
    Float u = (h < 8 || h == 12 || h == 13) ? dx : dy;
    Float v = (h < 4 || h == 12 || h == 13) ? dy : dz;
    return ((h & 1) ? -u : u) + ((h & 2) ? -v : v);
}
inline Float noise_weight(Float t) { Float t3 = t * t * t, t4 = t3 * t; return 6.0f * t4 * t - 15.0f * t4 + 10.0f * t3; }
inline Float noise_3d(Float x, Float y, Float z) {
    int64_t ix = f2isize(std::floor(x)), iy = f2isize(std::floor(y)), iz = f2isize(std::floor(z));
    Float dx = x - (Float)ix, dy = y - (Float)iy, dz = z - (Float)iz;
    ix &= 255; iy &= 255; iz &= 255;
    Float w000 = noise_grad(ix, iy, iz, dx, dy, dz), w100 = noise_grad(ix + 1, iy, iz, dx - 1.0f, dy, dz);
    Float w010 = noise_grad(ix, iy + 1, iz, dx, dy - 1.0f, dz), w110 = noise_grad(ix + 1, iy + 1, iz, dx - 1.0f, dy - 1.0f, dz);
    Float w001 = noise_grad(ix, iy, iz + 1, dx, dy, dz - 1.0f), w101 = noise_grad(ix + 1, iy, iz + 1, dx - 1.0f, dy, dz - 1.0f);
    Float w011 = noise_grad(ix, iy + 1, iz + 1, dx, dy - 1.0f, dz - 1.0f), w111 = noise_grad(ix + 1, iy + 1, iz + 1, dx - 1.0f, dy - 1.0f, dz - 1.0f);
    Float wx = noise_weight(dx), wy = noise_weight(dy), wz = noise_weight(dz);
    auto lerpf = [](Float t, Float a, Float b) { return (1.0f - t) * a + t * b; };
    Float x00 = lerpf(wx, w000, w100), x10 = lerpf(wx, w010, w110), x01 = lerpf(wx, w001, w101), x11 = lerpf(wx, w011, w111);
    Float y0 = lerpf(wy, x00, x10), y1 = lerpf(wy, x01, x11);
    return lerpf(wz, y0, y1);
}
inline Float noise_2d(Float x, Float y) { return noise_3d(x, y, 0.5f); }
inline Float bump_int(Float x) { return std::floor(x / 2.0f) + 2.0f * pmax((x / 2.0f) - std::floor(x / 2.0f) - 0.5f, 0.0f); }  // checkerboard_2d.rs:108-110

// ---- fbm / turbulence (core/src/texture/common.rs:119-214) over IdentityMapping3D (mapping/identity_3d.rs)
inline Float smooth_step(Float mn, Float mx, Float value) { Float v = pclamp((value - mn) / (mx - mn), 0.0f, 1.0f); return v * v * (-2.0f * v + 3.0f); }
inline Float fbm(V3 p, V3 dpdx, V3 dpdy, Float omega, int max_octaves) {
    Float len2 = pmax(length_squared(dpdx), length_squared(dpdy));
    Float n = pclamp(-1.0f - 0.5f * o_log2(len2), 0.0f, (Float)max_octaves);
    size_t n_int = f2usize(std::floor(n));
    Float sum = 0.0f, lambda = 1.0f, o = 1.0f;
    for (size_t i = 0; i < n_int; i++) { sum += o * noise_3d(lambda * p.x, lambda * p.y, lambda * p.z); lambda *= 1.99f; o *= omega; }
    Float n_partial = n - (Float)n_int;
    sum += o * smooth_step(0.3f, 0.7f, n_partial) * noise_3d(lambda * p.x, lambda * p.y, lambda * p.z);
    return sum;
}
inline Float turbulence(V3 p, V3 dpdx, V3 dpdy, Float omega, int max_octaves) {
    Float len2 = pmax(length_squared(dpdx), length_squared(dpdy));
    Float n = pclamp(-1.0f - 0.5f * o_log2(len2), 0.0f, (Float)max_octaves);
    size_t n_int = f2usize(std::floor(n));
    Float sum = 0.0f, lambda = 1.0f, o = 1.0f;
    for (size_t i = 0; i < n_int; i++) { sum += o * pabs(noise_3d(lambda * p.x, lambda * p.y, lambda * p.z)); lambda *= 1.99f; o *= omega; }
    Float n_partial = n - (Float)n_int;
    Float ss = smooth_step(0.3f, 0.7f, n_partial);
    sum += o * ((1.0f - ss) * 0.2f + ss * pabs(noise_3d(lambda * p.x, lambda * p.y, lambda * p.z)));  // lerp(ss, 0.2, |noise|)
    for (size_t i = n_int; i < (size_t)max_octaves; i++) { sum += o * 0.2f; o *= omega; }
    return sum;
}
inline V3 m4_point(const M4& m, V3 p) {  // Transform::transform_point (transform.rs:288-302)
    Float x = m.m[0][0] * p.x + m.m[0][1] * p.y + m.m[0][2] * p.z + m.m[0][3], y = m.m[1][0] * p.x + m.m[1][1] * p.y + m.m[1][2] * p.z + m.m[1][3];
    Float z = m.m[2][0] * p.x + m.m[2][1] * p.y + m.m[2][2] * p.z + m.m[2][3], w = m.m[3][0] * p.x + m.m[3][1] * p.y + m.m[3][2] * p.z + m.m[3][3];
    return w == 1.0f ? V3(x, y, z) : V3(x, y, z) / w;
}
inline V3 m4_vector(const M4& m, V3 v) {  // transform_vector (:373-380)
    return V3(m.m[0][0] * v.x + m.m[0][1] * v.y + m.m[0][2] * v.z, m.m[1][0] * v.x + m.m[1][1] * v.y + m.m[1][2] * v.z, m.m[2][0] * v.x + m.m[2][1] * v.y + m.m[2][2] * v.z);
}
static const Float MARBLE_C[9][3] = {{0.58f, 0.58f, 0.6f}, {0.58f, 0.58f, 0.6f}, {0.58f, 0.58f, 0.6f}, {0.5f, 0.5f, 0.5f}, {0.6f, 0.59f, 0.58f},
                                     {0.58f, 0.58f, 0.6f}, {0.58f, 0.58f, 0.6f}, {0.2f, 0.2f, 0.33f}, {0.58f, 0.58f, 0.6f}};  // marble.rs:104-114

// TextureMapping2D::map of the four 2D mappings (core/src/texture/mapping/{uv_2d,spherical_2d,cylinderical_2d,planar_2d}.rs)
inline V2 map_sphere(const M4& w2t, V3 p) {
    V3 vec = normalize(m4_point(w2t, p) - V3(0, 0, 0));
    return V2(spherical_theta(vec) * INV_PI, spherical_phi(vec) * INV_TWO_PI);
}
inline V2 map_cylinder(const M4& w2t, V3 p) {
    V3 vec = normalize(m4_point(w2t, p) - V3(0, 0, 0));
    return V2((PI + o_atan2(vec.y, vec.x)) * INV_TWO_PI, vec.z);
}
struct Texture;
struct TexCtx;
inline void fix_wrap(Float& d) { if (d > 0.5f) d = 1.0f - d; else if (d < -0.5f) d = -(d + 1.0f); }

enum { TK_CONST = 0, TK_SCALE = 1, TK_MIX = 2, TK_IMAGE = 3, TK_CHECKER = 4, TK_UV = 5, TK_BILERP = 6, TK_DOTS = 7, TK_FBM = 8, TK_WRINKLED = 9, TK_WINDY = 10,
       TK_MARBLE = 11, TK_CHECKER3D = 12 };
struct Texture {
    int kind = TK_CONST;
    Spec c;                       // TK_CONST (float textures: three equal channels)
    int t1 = -1, t2 = -1, amount = -1;
    int mip = -1; Float su = 1, sv = 1, du = 0, dv = 0;  // TK_IMAGE + UVMapping2D (also the mapping of the 2D procedural textures)
    int aa = 1;                   // TK_CHECKER: 0 none, 1 closedform
    Spec v[4];                    // TK_BILERP: v00 v01 v10 v11
    int mapping = 0;              // 2D textures: 0 uv, 1 spherical, 2 cylindrical (w2t), 3 planar (vs, vt, du = ds, dv = dt)
    V3 vs, vt;
    M4 w2t = M4::identity();      // 3D textures: IdentityMapping3D's matrix (the reference hands it tex2world, fbm.rs:65 — kept)
    Float omega = 0.5f, scale = 1.0f, variation = 0.2f; int octaves = 8;
};
struct TexCtx { V2 uv; Float dudx = 0, dvdx = 0, dudy = 0, dvdy = 0; V3 p, dpdx, dpdy; };

inline void map_2d(const Texture& t, const TexCtx& c, V2& st, V2& dstdx, V2& dstdy) {
    if (t.mapping == 0) {
        dstdx = V2(t.su * c.dudx, t.sv * c.dvdx); dstdy = V2(t.su * c.dudy, t.sv * c.dvdy);
        st = V2(t.su * c.uv.x + t.du, t.sv * c.uv.y + t.dv);
    } else if (t.mapping == 3) {
        dstdx = V2(dot(c.dpdx, t.vs), dot(c.dpdx, t.vt)); dstdy = V2(dot(c.dpdy, t.vs), dot(c.dpdy, t.vt));
        st = V2(t.du + dot(c.p, t.vs), t.dv + dot(c.p, t.vt));
    } else {
        const bool sph = t.mapping == 1;
        const Float delta = sph ? 0.1f : 0.01f;
        st = sph ? map_sphere(t.w2t, c.p) : map_cylinder(t.w2t, c.p);
        V2 sx = sph ? map_sphere(t.w2t, c.p + delta * c.dpdx) : map_cylinder(t.w2t, c.p + delta * c.dpdx);
        V2 sy = sph ? map_sphere(t.w2t, c.p + delta * c.dpdy) : map_cylinder(t.w2t, c.p + delta * c.dpdy);
        Float inv = 1.0f / delta;                                  // Vector2 / f multiplies by the reciprocal (vector2.rs:298-310)
        dstdx = V2(inv * (sx.x - st.x), inv * (sx.y - st.y)); dstdy = V2(inv * (sy.x - st.x), inv * (sy.y - st.y));
        fix_wrap(dstdx.y); fix_wrap(dstdy.y);
    }
}

inline Spec tex_eval(const std::vector<Texture>& tex, const std::vector<MipMap>& mips, int id, const TexCtx& c) {
    const Texture& t = tex[(size_t)id];
    switch (t.kind) {
        case TK_SCALE: return tex_eval(tex, mips, t.t1, c) * tex_eval(tex, mips, t.t2, c);
        case TK_MIX: {
            Spec a = tex_eval(tex, mips, t.t1, c), b = tex_eval(tex, mips, t.t2, c);
            Float amt = tex_eval(tex, mips, t.amount, c).c[0];
            return (1.0f - amt) * a + amt * b;
        }
        case TK_IMAGE: {
            V2 st, dstdx, dstdy; map_2d(t, c, st, dstdx, dstdy);
            return mips[(size_t)t.mip].lookup(st, dstdx, dstdy);
        }
        case TK_CHECKER: {  // checkerboard_2d.rs:60-104
            V2 st, dstdx, dstdy; map_2d(t, c, st, dstdx, dstdy);
            Spec a = tex_eval(tex, mips, t.t1, c), b = tex_eval(tex, mips, t.t2, c);
            auto point = [&]() { return ((int32_t)((uint32_t)f2i32(std::floor(st.x)) + (uint32_t)f2i32(std::floor(st.y))) % 2 == 0) ? a : b; };
            if (t.aa == 0) return point();
            Float ds = pmax(pabs(dstdx.x), pabs(dstdy.x)), dt = pmax(pabs(dstdx.y), pabs(dstdy.y));
            Float s0 = st.x - ds, s1 = st.x + ds, t0 = st.y - dt, t1 = st.y + dt;
            if (std::floor(s0) == std::floor(s1) && std::floor(t0) == std::floor(t1)) return point();
            Float sint = (bump_int(s1) - bump_int(s0)) / (2.0f * ds), tint = (bump_int(t1) - bump_int(t0)) / (2.0f * dt);
            Float area2 = (ds > 1.0f || dt > 1.0f) ? 0.5f : sint + tint - 2.0f * sint * tint;
            return a * (1.0f - area2) + b * area2;
        }
        case TK_UV: {  // uv.rs:33-38
            V2 st, dstdx_, dstdy_; map_2d(t, c, st, dstdx_, dstdy_);
            return Spec(st.x - std::floor(st.x), st.y - std::floor(st.y), 0.0f);
        }
        case TK_BILERP: {  // bilerp.rs:58-71
            V2 st, dstdx_, dstdy_; map_2d(t, c, st, dstdx_, dstdy_);
            Float s00 = (1.0f - st.x) * (1.0f - st.y), s01 = (1.0f - st.x) * st.y, s10 = st.x * (1.0f - st.y), s11 = st.x * st.y;
            return (t.v[0] * s00) + (t.v[1] * s01) + (t.v[2] * s10) + (t.v[3] * s11);
        }
        case TK_DOTS: {  // dots.rs:48-69
            V2 st, dstdx_, dstdy_; map_2d(t, c, st, dstdx_, dstdy_);
            Float s_cell = std::floor(st.x + 0.5f), t_cell = std::floor(st.y + 0.5f);
            if (noise_2d(s_cell + 0.5f, t_cell + 0.5f) > 0.0f) {
                const Float radius = 0.35f, max_shift = 0.5f - radius;
                Float s_center = s_cell + max_shift * noise_2d(s_cell + 1.5f, t_cell + 2.8f);
                Float t_center = t_cell + max_shift * noise_2d(s_cell + 4.5f, t_cell + 9.8f);
                Float ddx = st.x - s_center, ddy = st.y - t_center;
                if (ddx * ddx + ddy * ddy < radius * radius) return tex_eval(tex, mips, t.t1, c);
            }
            return tex_eval(tex, mips, t.t2, c);
        }
        case TK_FBM: case TK_WRINKLED: case TK_WINDY: case TK_MARBLE: case TK_CHECKER3D: {
            V3 dpdx = m4_vector(t.w2t, c.dpdx), dpdy = m4_vector(t.w2t, c.dpdy), p = m4_point(t.w2t, c.p);
            if (t.kind == TK_FBM) return Spec(fbm(p, dpdx, dpdy, t.omega, t.octaves));                    // fbm.rs:45-50
            if (t.kind == TK_WRINKLED) return Spec(turbulence(p, dpdx, dpdy, t.omega, t.octaves));        // wrinkled.rs:45-50
            if (t.kind == TK_WINDY) {                                                                      // windy.rs:36-44
                Float wind = fbm(0.1f * p, 0.1f * dpdx, 0.1f * dpdy, 0.5f, 3), wave = fbm(p, dpdx, dpdy, 0.5f, 6);
                return Spec(pabs(wind) * wave);
            }
            if (t.kind == TK_CHECKER3D) {                                                                  // checkerboard_3d.rs:44-53
                Spec a = tex_eval(tex, mips, t.t1, c), b = tex_eval(tex, mips, t.t2, c);
                uint32_t sum = (uint32_t)f2i32(std::floor(p.x)) + (uint32_t)f2i32(std::floor(p.y)) + (uint32_t)f2i32(std::floor(p.z));
                return ((int32_t)sum % 2 == 0) ? a : b;
            }
            p = p * t.scale;                                                                               // marble.rs:54-85
            Float marble = p.y + t.variation * fbm(p, t.scale * dpdx, t.scale * dpdy, t.omega, t.octaves);
            Float tt = 0.5f + 0.5f * o_sin(marble);
            size_t first = pmin<size_t>(1, f2usize(std::floor(tt * 6.0f)));   // `min(1, ..)` as in the reference (and in C++ pbrt)
            tt = tt * 6.0f - (Float)first;
            Spec c0(MARBLE_C[first][0], MARBLE_C[first][1], MARBLE_C[first][2]), c1(MARBLE_C[first + 1][0], MARBLE_C[first + 1][1], MARBLE_C[first + 1][2]);
            Spec c2(MARBLE_C[first + 2][0], MARBLE_C[first + 2][1], MARBLE_C[first + 2][2]), c3(MARBLE_C[first + 3][0], MARBLE_C[first + 3][1], MARBLE_C[first + 3][2]);
            Spec s0 = (1.0f - tt) * c0 + tt * c1, s1 = (1.0f - tt) * c1 + tt * c2, s2 = (1.0f - tt) * c2 + tt * c3;
            s0 = (1.0f - tt) * s0 + tt * s1; s1 = (1.0f - tt) * s1 + tt * s2;
            return 1.5f * ((1.0f - tt) * s0 + tt * s1);
        }
        default: return t.c;
    }
}

}  // namespace orc
