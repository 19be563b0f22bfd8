// Device side of the image textures: MIPMap<T>::lookup / triangle / ewa / texel (core/src/mipmap/mod.rs:205-371, 569-608), the texture
// programs the host flattens from ScaleTexture / MixTexture / ImageTexture / ConstantTexture trees (textures/src/*.rs), UVMapping2D
// (core/src/texture/mapping/uv_2d.rs:52-60), SurfaceInteraction::compute_differentials (core/src/interaction/surface_interaction.rs:203-278)
// and the camera-ray differentials it consumes (cameras/src/perspective_camera.rs:173-200, ray.rs:90-99, transform.rs:464-472).
// f32 throughout, in the reference's order; log2 is evaluated in f64 and rounded (glibc's f32 log2f is not reproducible on the device,
// same policy as the trigonometric functions: DESIGN.md section 2).
#pragma once
#include "pt_device.h"
#include "bsdf_general.h"

namespace ph {

struct TexCtx { f2 uv; float dudx, dvdx, dudy, dvdy; f3 p, dpdx, dpdy; };

PH_DEV float d_log2(float x) { return (float)log2((double)x); }
PH_DEV long long f2ll_sat(float f) {  // `as isize`
    if (f != f) return 0;
    if (f >= 9223372036854775808.0f) return 0x7FFFFFFFFFFFFFFFll;
    if (f <= -9223372036854775808.0f) return (long long)0x8000000000000000ull;
    return (long long)f;
}
PH_DEV long long rem_ll(long long a, long long b) { long long r = a - (a / b) * b; return r < 0 ? r + b : r; }  // pbrt/common.rs:116-126

PH_DEV spec mip_texel(const DeviceScene& sc, const MipRec& m, uint32_t level, long long s, long long t) {
    const long long w = (long long)m.level_w[level], h = (long long)m.level_h[level];
    // repeat: every pyramid level is a power of two wide and high (MIPMap::new resamples other sizes first, mipmap/mod.rs:383-529), and for a power of two `rem(a, b)` with its
    // "negative remainder + b" (pbrt/common.rs:116-126) is the two's-complement mask — no 64-bit division per texel
    if (m.wrap == 0u) { s = s & (w - 1); t = t & (h - 1); }
    else if (m.wrap == 2u) { s = s < 0 ? 0 : (s > w - 1 ? w - 1 : s); t = t < 0 ? 0 : (t > h - 1 ? h - 1 : t); }
    else if (s < 0 || s >= w || t < 0 || t >= h) return mks1(0.0f);
    const float4 v = *reinterpret_cast<const float4*>(sc.texels + ((size_t)m.level_off[level] + (size_t)t * (size_t)w + (size_t)s));
    return mks(v.x, v.y, v.z);
}
// the same texel with 32-bit coordinates, for callers that know them to be small (|s|, |t| < 2^30; a level is at most 2^30 wide and high and the pool holds at most 2^32 texels: textures_api.hip checks both at upload)
PH_DEV spec mip_texel_i(const DeviceScene& sc, const MipRec& m, uint32_t level, int s, int t) {
    const int w = (int)m.level_w[level], h = (int)m.level_h[level];
    if (m.wrap == 0u) { s = s & (w - 1); t = t & (h - 1); }
    else if (m.wrap == 2u) { s = s < 0 ? 0 : (s > w - 1 ? w - 1 : s); t = t < 0 ? 0 : (t > h - 1 ? h - 1 : t); }
    else if (s < 0 || s >= w || t < 0 || t >= h) return mks1(0.0f);
    const float4 v = *reinterpret_cast<const float4*>(sc.texels + ((size_t)m.level_off[level] + (size_t)((uint32_t)t * (uint32_t)w + (uint32_t)s)));
    return mks(v.x, v.y, v.z);
}
PH_DEV spec mip_triangle(const DeviceScene& sc, const MipRec& m, uint32_t level, f2 st) {
    if (level > m.n_levels - 1u) level = m.n_levels - 1u;
    const float s = st.x * (float)m.level_w[level] - 0.5f, t = st.y * (float)m.level_h[level] - 0.5f;
    const long long s0 = f2ll_sat(floorf(s)), t0 = f2ll_sat(floorf(t));
    const float ds = s - (float)s0, dt = t - (float)t0;
    return mip_texel(sc, m, level, s0, t0) * (1.0f - ds) * (1.0f - dt) + mip_texel(sc, m, level, s0, t0 + 1) * (1.0f - ds) * dt +
           mip_texel(sc, m, level, s0 + 1, t0) * ds * (1.0f - dt) + mip_texel(sc, m, level, s0 + 1, t0 + 1) * ds * dt;
}
PH_DEV spec mip_ewa(const DeviceScene& sc, const MipRec& m, uint32_t level, f2 st, f2 dst0, f2 dst1) {
    if (level >= m.n_levels) return mip_texel(sc, m, m.n_levels - 1u, 0, 0);
    const float us = (float)m.level_w[level], vs = (float)m.level_h[level];
    const float s = st.x * us - 0.5f, t = st.y * vs - 0.5f;
    const float d0x = dst0.x * us, d0y = dst0.y * vs, d1x = dst1.x * us, d1y = dst1.y * vs;
    float a = d0y * d0y + d1y * d1y + 1.0f;
    float b = -2.0f * (d0x * d0y + d1x * d1y);
    float c = d0x * d0x + d1x * d1x + 1.0f;
    const float inv_f = ph_div(1.0f, a * c - b * b * 0.25f);
    a *= inv_f; b *= inv_f; c *= inv_f;
    const float det = -b * b + 4.0f * a * c;
    const float inv_det = ph_div(1.0f, det);
    const float u_sqrt = ph_sqrt(det * c), v_sqrt = ph_sqrt(a * det);
    const long long s0 = f2ll_sat(ceilf(s - 2.0f * inv_det * u_sqrt)), s1 = f2ll_sat(floorf(s + 2.0f * inv_det * u_sqrt));
    const long long t0 = f2ll_sat(ceilf(t - 2.0f * inv_det * v_sqrt)), t1 = f2ll_sat(floorf(t + 2.0f * inv_det * v_sqrt));
    spec sum = mks1(0.0f);
    float sum_wts = 0.0f;
    const long long lim = 1ll << 30;
    if (s0 > -lim && s1 < lim && t0 > -lim && t1 < lim) {   // the ellipse's box in 32-bit counters (always, short of degenerate differentials): (float)(int) == (float)(long long) here
        for (int it = (int)t0; it <= (int)t1; it++) {
            const float tt = (float)it - t;
            for (int is = (int)s0; is <= (int)s1; is++) {
                const float ss = (float)is - s;
                const float r2 = a * ss * ss + b * ss * tt + c * tt * tt;
                if (r2 < 1.0f) {
                    uint32_t index = f2u_sat(r2 * (float)PH_EWA_LUT_SIZE);
                    if (index > PH_EWA_LUT_SIZE - 1u) index = PH_EWA_LUT_SIZE - 1u;
                    const float weight = sc.ewa_lut[index];
                    sum = sum + mip_texel_i(sc, m, level, is, it) * weight;
                    sum_wts += weight;
                }
            }
        }
    } else {
        for (long long it = t0; it <= t1; it++) {
            const float tt = (float)it - t;
            for (long long is = s0; is <= s1; is++) {
                const float ss = (float)is - s;
                const float r2 = a * ss * ss + b * ss * tt + c * tt * tt;
                if (r2 < 1.0f) {
                    uint32_t index = f2u_sat(r2 * (float)PH_EWA_LUT_SIZE);
                    if (index > PH_EWA_LUT_SIZE - 1u) index = PH_EWA_LUT_SIZE - 1u;
                    const float weight = sc.ewa_lut[index];
                    sum = sum + mip_texel(sc, m, level, is, it) * weight;
                    sum_wts += weight;
                }
            }
        }
    }
    if (m.is_float) return mks1(ph_div(sum.r, sum_wts));  // Float: a true division; RGBSpectrum: times the reciprocal (rgb_spectrum.rs:255-263)
    return sum / sum_wts;
}
PH_DEV spec mip_lookup(const DeviceScene& sc, const MipRec& m, f2 st, f2 dst0, f2 dst1) {
    const uint32_t levels = m.n_levels;
    if (m.filtering == 0u) {
        const float width = pmaxf(pmaxf(pabs(dst0.x), pabs(dst0.y)), pmaxf(pabs(dst1.x), pabs(dst1.y)));
        const float level = (float)levels - 1.0f + d_log2(pmaxf(width, 1e-8f));
        if (level < 0.0f) return mip_triangle(sc, m, 0u, st);
        if (level >= (float)(levels - 1u)) return mip_texel(sc, m, levels - 1u, 0, 0);
        const uint32_t il = f2u_sat(floorf(level));
        const float delta = level - (float)il;
        return mip_triangle(sc, m, il, st) * (1.0f - delta) + mip_triangle(sc, m, il + 1u, st) * delta;
    }
    if (dst0.x * dst0.x + dst0.y * dst0.y < dst1.x * dst1.x + dst1.y * dst1.y) { const f2 tmp = dst0; dst0 = dst1; dst1 = tmp; }
    const float major_length = ph_sqrt(dst0.x * dst0.x + dst0.y * dst0.y);
    float minor_length = ph_sqrt(dst1.x * dst1.x + dst1.y * dst1.y);
    const float adjusted = minor_length * m.max_anisotropy;
    if (adjusted < major_length && minor_length > 0.0f) {
        const float scl = ph_div(major_length, adjusted);
        dst1.x *= scl; dst1.y *= scl;
        minor_length *= scl;
    }
    if (minor_length == 0.0f) return mip_triangle(sc, m, 0u, st);
    const float lod = pmaxf(0.0f, (float)levels - 1.0f + d_log2(minor_length));
    const uint32_t il = f2u_sat(floorf(lod));
    const float t = lod - (float)il;
    return mip_ewa(sc, m, il, st, dst0, dst1) * (1.0f - t) + mip_ewa(sc, m, il + 1u, st, dst0, dst1) * t;
}

// ---- Perlin noise (core/src/texture/common.rs:9-117); the permutation is Ken Perlin's reference table, twice
static __device__ const uint8_t kNoisePerm[512] = {
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
// The table is read three levels deep per lattice corner, eight corners per noise value, up to eighteen values per bump-mapped hit (three evaluations of an fbm displacement):
// a chain of ~50 dependent byte loads.  Kernels that evaluate procedural textures copy it to LDS once per block (noise_lds_fill) and the chain runs at LDS latency, off the
// texture-addresser (round 4: the bump phase was 35 % of configs[4]'s texture pass, profiles/r04_phase_clock_shade_texture_config4_*).
#if defined(__HIP_DEVICE_COMPILE__)
static __shared__ uint8_t g_noise_perm_lds[512];
#define PH_NOISE_PERM(i) g_noise_perm_lds[i]
// every kernel from which the general evaluator (tex_eval<false, *>) can be reached calls this first, all threads of the block
PH_DEV void noise_lds_fill() {
    for (uint32_t i = threadIdx.x; i < 512u; i += blockDim.x) g_noise_perm_lds[i] = kNoisePerm[i];
    __syncthreads();
}
#else
#define PH_NOISE_PERM(i) kNoisePerm[i]
PH_DEV void noise_lds_fill() {}   // (the host pass of the compiler only parses the kernels)
#endif
PH_DEV float noise_grad(long long x, long long y, long long z, float dx, float dy, float dz) {
    const int h = PH_NOISE_PERM(PH_NOISE_PERM(PH_NOISE_PERM(x) + y) + z) & 15;
    const float u = (h < 8 || h == 12 || h == 13) ? dx : dy;
    const float v = (h < 4 || h == 12 || h == 13) ? dy : dz;
    return ((h & 1) ? -u : u) + ((h & 2) ? -v : v);
}
PH_DEV float noise_weight(float t) { const float t3 = t * t * t, t4 = t3 * t; return 6.0f * t4 * t - 15.0f * t4 + 10.0f * t3; }
PH_DEV float lerpf(float t, float a, float b) { return (1.0f - t) * a + t * b; }
PH_DEV float noise_3d(float x, float y, float z) {
    long long ix = f2ll_sat(floorf(x)), iy = f2ll_sat(floorf(y)), iz = f2ll_sat(floorf(z));
    const float dx = x - (float)ix, dy = y - (float)iy, dz = z - (float)iz;
    ix &= 255; iy &= 255; iz &= 255;
    const float w000 = noise_grad(ix, iy, iz, dx, dy, dz), w100 = noise_grad(ix + 1, iy, iz, dx - 1.0f, dy, dz);
    const float w010 = noise_grad(ix, iy + 1, iz, dx, dy - 1.0f, dz), w110 = noise_grad(ix + 1, iy + 1, iz, dx - 1.0f, dy - 1.0f, dz);
    const float w001 = noise_grad(ix, iy, iz + 1, dx, dy, dz - 1.0f), w101 = noise_grad(ix + 1, iy, iz + 1, dx - 1.0f, dy, dz - 1.0f);
    const float w011 = noise_grad(ix, iy + 1, iz + 1, dx, dy - 1.0f, dz - 1.0f), w111 = noise_grad(ix + 1, iy + 1, iz + 1, dx - 1.0f, dy - 1.0f, dz - 1.0f);
    const float wx = noise_weight(dx), wy = noise_weight(dy), wz = noise_weight(dz);
    const float x00 = lerpf(wx, w000, w100), x10 = lerpf(wx, w010, w110), x01 = lerpf(wx, w001, w101), x11 = lerpf(wx, w011, w111);
    return lerpf(wz, lerpf(wy, x00, x10), lerpf(wy, x01, x11));
}
PH_DEV float noise_2d(float x, float y) { return noise_3d(x, y, 0.5f); }
PH_DEV float bump_int(float x) { return floorf(ph_div(x, 2.0f)) + 2.0f * pmaxf(ph_div(x, 2.0f) - floorf(ph_div(x, 2.0f)) - 0.5f, 0.0f); }  // checkerboard_2d.rs:108-110

// ---- fbm / turbulence (core/src/texture/common.rs:119-214)
PH_DEV float smooth_step(float mn, float mx, float value) { const float v = pclampf(ph_div(value - mn, mx - mn), 0.0f, 1.0f); return v * v * (-2.0f * v + 3.0f); }
PH_DEV float tex_fbm(f3 p, f3 dpdx, f3 dpdy, float omega, uint32_t max_octaves, bool turbulence) {
    const float len2 = pmaxf(length_squared(dpdx), length_squared(dpdy));
    const float n = pclampf(-1.0f - 0.5f * d_log2(len2), 0.0f, (float)max_octaves);
    const uint32_t n_int = f2u_sat(floorf(n));
    float sum = 0.0f, lambda = 1.0f, o = 1.0f;
    for (uint32_t i = 0; i < n_int; i++) {
        const float nz = noise_3d(lambda * p.x, lambda * p.y, lambda * p.z);
        sum += o * (turbulence ? pabs(nz) : nz);
        lambda *= 1.99f; o *= omega;
    }
    const float n_partial = n - (float)n_int;
    const float ss = smooth_step(0.3f, 0.7f, n_partial), nz = noise_3d(lambda * p.x, lambda * p.y, lambda * p.z);
    if (!turbulence) return sum + o * ss * nz;
    sum += o * ((1.0f - ss) * 0.2f + ss * pabs(nz));
    for (uint32_t i = n_int; i < max_octaves; i++) { sum += o * 0.2f; o *= omega; }
    return sum;
}
static __device__ const float kMarbleC[9][3] = {{0.58f, 0.58f, 0.6f}, {0.58f, 0.58f, 0.6f}, {0.58f, 0.58f, 0.6f}, {0.5f, 0.5f, 0.5f}, {0.6f, 0.59f, 0.58f},
                                                {0.58f, 0.58f, 0.6f}, {0.58f, 0.58f, 0.6f}, {0.2f, 0.2f, 0.33f}, {0.58f, 0.58f, 0.6f}};  // marble.rs:104-114
PH_DEV f3 xf_point16(const float* m, f3 p) {  // Transform::transform_point (transform.rs:288-302)
    const float x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    const float z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11], w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    return (w == 1.0f) ? mk3(x, y, z) : mk3(x, y, z) / w;
}

// TextureMapping2D::map of the four 2D mappings (core/src/texture/mapping/{uv_2d,spherical_2d,cylinderical_2d,planar_2d}.rs)
PH_DEV f2 map_sphere(const float* m, f3 p) {
    const f3 v = normalize(xf_point16(m, p) - mk3(0.0f, 0.0f, 0.0f));
    return mk2(spherical_theta(v) * kInvPi, spherical_phi(v) * kInvTwoPi);
}
PH_DEV f2 map_cylinder(const float* m, f3 p) {
    const f3 v = normalize(xf_point16(m, p) - mk3(0.0f, 0.0f, 0.0f));
    return mk2((kPi + d_atan2(v.y, v.x)) * kInvTwoPi, v.z);
}
PH_DEV float fix_wrap(float d) { return d > 0.5f ? 1.0f - d : (d < -0.5f ? -(d + 1.0f) : d); }
PH_DEV void map_2d(const TexOp& op, const TexCtx& c, f2& st, f2& dstdx, f2& dstdy) {
    if (op.mapping == 0u) {
        dstdx = mk2(op.su * c.dudx, op.sv * c.dvdx); dstdy = mk2(op.su * c.dudy, op.sv * c.dvdy);
        st = mk2(op.su * c.uv.x + op.du, op.sv * c.uv.y + op.dv);
    } else if (op.mapping == 3u) {
        const f3 vs = mk3(op.m[0], op.m[1], op.m[2]), vt = mk3(op.m[3], op.m[4], op.m[5]);
        dstdx = mk2(dot(c.dpdx, vs), dot(c.dpdx, vt)); dstdy = mk2(dot(c.dpdy, vs), dot(c.dpdy, vt));
        st = mk2(op.du + dot(c.p, vs), op.dv + dot(c.p, vt));
    } else {
        const bool sph = op.mapping == 1u;
        const float delta = sph ? 0.1f : 0.01f;
        st = sph ? map_sphere(op.m, c.p) : map_cylinder(op.m, c.p);
        const f2 sx = sph ? map_sphere(op.m, c.p + delta * c.dpdx) : map_cylinder(op.m, c.p + delta * c.dpdx);
        const f2 sy = sph ? map_sphere(op.m, c.p + delta * c.dpdy) : map_cylinder(op.m, c.p + delta * c.dpdy);
        const float inv = ph_div(1.0f, delta);
        dstdx = mk2(inv * (sx.x - st.x), fix_wrap(inv * (sx.y - st.y))); dstdy = mk2(inv * (sy.x - st.x), fix_wrap(inv * (sy.y - st.y)));
    }
}

// the (s, t) of map_2d alone, for look-ups without differentials
PH_DEV f2 map_2d_st(const TexOp& op, const TexCtx& c) {
    if (op.mapping == 0u) return mk2(op.su * c.uv.x + op.du, op.sv * c.uv.y + op.dv);
    if (op.mapping == 3u) return mk2(op.du + dot(c.p, mk3(op.m[0], op.m[1], op.m[2])), op.dv + dot(c.p, mk3(op.m[3], op.m[4], op.m[5])));
    return op.mapping == 1u ? map_sphere(op.m, c.p) : map_cylinder(op.m, c.p);
}

// The evaluator's value stack lives in LDS as [depth][channel][thread of the block] (conflict-free, no scratch): indexed dynamically by the program, a per-thread array
// would go to scratch memory — 72 B per thread that the texture pass of configs[4] wrote and re-read 3.2 TB of per frame (round 3).  Kernels that evaluate textures run
// blocks of at most PH_TEX_LDS_THREADS threads (texture_kernel, the ALPHA = 2 traversal kernel: 256).
#define PH_TEX_LDS_THREADS 256
struct TexStack {
#if defined(__HIP_DEVICE_COMPILE__)
    float* base;   // &lds[0][0][threadIdx.x]
    PH_DEV void put(int k, spec v) { base[(3 * k) * PH_TEX_LDS_THREADS] = v.r; base[(3 * k + 1) * PH_TEX_LDS_THREADS] = v.g; base[(3 * k + 2) * PH_TEX_LDS_THREADS] = v.b; }
    PH_DEV spec get(int k) const { return mks(base[(3 * k) * PH_TEX_LDS_THREADS], base[(3 * k + 1) * PH_TEX_LDS_THREADS], base[(3 * k + 2) * PH_TEX_LDS_THREADS]); }
#else
    spec v_[PH_TEX_STACK];
    void put(int k, spec v) { v_[k] = v; }
    spec get(int k) const { return v_[k]; }
#endif
};

// Runs texture `id`'s postfix program.  Kept out of line: the shade kernels call it only for materials that carry a texture.
// `dsc` = DeviceScene::self (the by-value kernel argument must not have its address taken: it would be copied to scratch).
// SIMPLE = the scene's texture programs consist of constants, image maps, scale and mix only (what most scene files use): the procedural classes and their Perlin noise are
// compiled out, the caller's register budget shrinks with them (an out-of-line callee's registers count towards its kernel's)
// NODIFF = the context carries no differentials (every ray but a camera ray: `c`'s du/dv d x/y and dp d x/y are zero): an image map's filtered look-up then ends in
// MIPMap::triangle(0, st) whatever its filter — trilinear: width 0 -> level = levels - 1 + log2(1e-8) < 0 (a pyramid has at most PH_MIP_MAX_LEVELS = 16 levels),
// mipmap/mod.rs:222-231; EWA: minor_length == 0, :250-262 — and the evaluator is compiled without the EWA / trilinear code (the mapping's differentials, zero for finite
// mapping parameters, are not formed).  The procedural classes run their general code on the zero differentials.
template <bool SIMPLE, bool NODIFF>
static __device__ __forceinline__ spec tex_eval_body(const DeviceScene* dsc, uint32_t id, const TexCtx& c) {
    const DeviceScene& sc = *dsc;
    const TexRec tr = sc.textures[id];
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ float tex_stack_lds[PH_TEX_STACK * 3][PH_TEX_LDS_THREADS];
    TexStack stk; stk.base = &tex_stack_lds[0][threadIdx.x];
#else
    TexStack stk;
#endif
    int sp = 0;
    for (uint32_t k = 0; k < tr.n_ops; k++) {
        const TexOp& op = sc.tex_ops[tr.first_op + k];
        switch (op.op) {
        case PH_TOP_CONST: stk.put(sp++, mks(op.c[0], op.c[1], op.c[2])); break;
        case PH_TOP_IMAGE: {
            if (NODIFF) { stk.put(sp++, mip_triangle(sc, sc.mipmaps[op.mip], 0u, map_2d_st(op, c))); break; }
            f2 p, dstdx, dstdy; map_2d(op, c, p, dstdx, dstdy);
            stk.put(sp++, mip_lookup(sc, sc.mipmaps[op.mip], p, dstdx, dstdy));
            break;
        }
        case PH_TOP_MUL: { sp--; stk.put(sp - 1, stk.get(sp - 1) * stk.get(sp)); break; }                          // scale.rs:33
        case PH_TOP_CHECKER: if (!SIMPLE) {  // checkerboard_2d.rs:60-104; stack = tex1, tex2 (both are evaluated: they have no side effects)
            sp--;
            const spec a = stk.get(sp - 1), b = stk.get(sp);
            f2 p, dstdx, dstdy; map_2d(op, c, p, dstdx, dstdy);
            const bool even = (int)((uint32_t)f2i_sat(floorf(p.x)) + (uint32_t)f2i_sat(floorf(p.y))) % 2 == 0;
            spec r = even ? a : b;
            if (op.mip != 0u) {
                const float ds = pmaxf(pabs(dstdx.x), pabs(dstdy.x)), dt = pmaxf(pabs(dstdx.y), pabs(dstdy.y));
                const float s0 = p.x - ds, s1 = p.x + ds, t0 = p.y - dt, t1 = p.y + dt;
                if (!(floorf(s0) == floorf(s1) && floorf(t0) == floorf(t1))) {
                    const float sint = ph_div(bump_int(s1) - bump_int(s0), 2.0f * ds), tint = ph_div(bump_int(t1) - bump_int(t0), 2.0f * dt);
                    const float area2 = (ds > 1.0f || dt > 1.0f) ? 0.5f : sint + tint - 2.0f * sint * tint;
                    r = a * (1.0f - area2) + b * area2;
                }
            }
            stk.put(sp - 1, r);
            } break;
        case PH_TOP_UV: if (!SIMPLE) {  // uv.rs:33-38
            f2 p, dx_, dy_; map_2d(op, c, p, dx_, dy_);
            stk.put(sp++, mks(p.x - floorf(p.x), p.y - floorf(p.y), 0.0f));
            } break;
        case PH_TOP_BILERP: if (!SIMPLE) {  // bilerp.rs:58-71; stack = v00, v01, v10, v11
            sp -= 3;
            f2 p, dx_, dy_; map_2d(op, c, p, dx_, dy_);
            const float s00 = (1.0f - p.x) * (1.0f - p.y), s01 = (1.0f - p.x) * p.y, s10 = p.x * (1.0f - p.y), s11 = p.x * p.y;
            stk.put(sp - 1, (stk.get(sp - 1) * s00) + (stk.get(sp) * s01) + (stk.get(sp + 1) * s10) + (stk.get(sp + 2) * s11));
            } break;
        case PH_TOP_FBM: case PH_TOP_WRINKLED: case PH_TOP_WINDY: case PH_TOP_MARBLE: case PH_TOP_CHECKER3D: if (!SIMPLE) {  // IdentityMapping3D::map (identity_3d.rs)
            const f3 dpdx = xf_vec(op.m, c.dpdx), dpdy = xf_vec(op.m, c.dpdy);
            f3 p = xf_point16(op.m, c.p);
            if (op.op == PH_TOP_FBM) stk.put(sp++, mks1(tex_fbm(p, dpdx, dpdy, op.omega, op.octaves, false)));            // fbm.rs:45-50
            else if (op.op == PH_TOP_WRINKLED) stk.put(sp++, mks1(tex_fbm(p, dpdx, dpdy, op.omega, op.octaves, true)));   // wrinkled.rs:45-50
            else if (op.op == PH_TOP_WINDY) {                                                                          // windy.rs:36-44
                const float wind = tex_fbm(0.1f * p, 0.1f * dpdx, 0.1f * dpdy, 0.5f, 3u, false), wave = tex_fbm(p, dpdx, dpdy, 0.5f, 6u, false);
                stk.put(sp++, mks1(pabs(wind) * wave));
            } else if (op.op == PH_TOP_CHECKER3D) {                                                                    // checkerboard_3d.rs:44-53
                sp--;
                const uint32_t sum = (uint32_t)f2i_sat(floorf(p.x)) + (uint32_t)f2i_sat(floorf(p.y)) + (uint32_t)f2i_sat(floorf(p.z));
                stk.put(sp - 1, ((int)sum % 2 == 0) ? stk.get(sp - 1) : stk.get(sp));
            } else {                                                                                                   // marble.rs:54-85
                p = p * op.scale;
                const float marble = p.y + op.variation * tex_fbm(p, op.scale * dpdx, op.scale * dpdy, op.omega, op.octaves, false);
                float tt = 0.5f + 0.5f * d_sin(marble);
                uint32_t first = f2u_sat(floorf(tt * 6.0f));
                if (first > 1u) first = 1u;   // `min(1, ..)`, as in the reference
                tt = tt * 6.0f - (float)first;
                const spec c0 = mks(kMarbleC[first][0], kMarbleC[first][1], kMarbleC[first][2]), c1 = mks(kMarbleC[first + 1][0], kMarbleC[first + 1][1], kMarbleC[first + 1][2]);
                const spec c2 = mks(kMarbleC[first + 2][0], kMarbleC[first + 2][1], kMarbleC[first + 2][2]), c3 = mks(kMarbleC[first + 3][0], kMarbleC[first + 3][1], kMarbleC[first + 3][2]);
                spec s0 = (1.0f - tt) * c0 + tt * c1, s1 = (1.0f - tt) * c1 + tt * c2;
                const spec s2 = (1.0f - tt) * c2 + tt * c3;
                s0 = (1.0f - tt) * s0 + tt * s1; s1 = (1.0f - tt) * s1 + tt * s2;
                stk.put(sp++, 1.5f * ((1.0f - tt) * s0 + tt * s1));
            }
            } break;
        case PH_TOP_DOTS: if (!SIMPLE) {  // dots.rs:48-69; stack = inside, outside
            sp--;
            f2 p, dx_, dy_; map_2d(op, c, p, dx_, dy_);
            const float s_cell = floorf(p.x + 0.5f), t_cell = floorf(p.y + 0.5f);
            bool inside = false;
            if (noise_2d(s_cell + 0.5f, t_cell + 0.5f) > 0.0f) {
                const float radius = 0.35f, max_shift = 0.5f - radius;
                const float s_center = s_cell + max_shift * noise_2d(s_cell + 1.5f, t_cell + 2.8f);
                const float t_center = t_cell + max_shift * noise_2d(s_cell + 4.5f, t_cell + 9.8f);
                const float ddx = p.x - s_center, ddy = p.y - t_center;
                inside = ddx * ddx + ddy * ddy < radius * radius;
            }
            stk.put(sp - 1, inside ? stk.get(sp - 1) : stk.get(sp));
            } break;
        default: {  // PH_TOP_MIX: (1 - amt) * t1 + amt * t2 (mix.rs:36-41); stack = t1, t2, amount
            sp -= 2;
            const float amt = stk.get(sp + 1).r;
            stk.put(sp - 1, (1.0f - amt) * stk.get(sp - 1) + amt * stk.get(sp));
            break;
        }
        }
    }
    return stk.get(0);
}
// out of line: the evaluator's registers and code stay out of its callers (several call sites per material class)
template <bool SIMPLE, bool NODIFF> struct TexEval { static __device__ __noinline__ spec run(const DeviceScene* dsc, uint32_t id, TexCtx c) { return tex_eval_body<SIMPLE, NODIFF>(dsc, id, c); } };
template <bool SIMPLE = false, bool NODIFF = false> PH_DEV spec tex_eval(const DeviceScene* dsc, uint32_t id, const TexCtx& c) { return TexEval<SIMPLE, NODIFF>::run(dsc, id, c); }

// ---- ray differentials of a camera ray, in world space, scaled as render_tile does (sampler_integrator.rs:358) --------------
struct RayDiff { f3 rx_o, ry_o, rx_d, ry_d; };
PH_DEV f3 xf_point_plain(const float* m, f3 p) {  // Transform::transform_point (transform.rs:288-302)
    const float x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    const float y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    const float z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    const float w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    return (w == 1.0f) ? mk3(x, y, z) : mk3(x, y, z) / w;
}
// Camera::generate_ray_differential's default (core/src/camera.rs:29-78), used by EnvironmentCamera: finite differences of WORLD-space rays over a
// 0.05-pixel shift (every weight is 1: only the first eps of each loop is used); `Vector3 / eps` multiplies by the reciprocal.  Out of line, arguments by pointer.
__device__ __noinline__ void environment_camera_differentials(const CameraRec* cam, f2 p_film, const f3* o_world_p, const f3* d_world_p, uint32_t spp, RayDiff* out) {
    const f3 o_world = *o_world_p, d_world = *d_world_p;
    const float eps = 0.05f, inv = ph_div(1.0f, eps);
    RayIn rx, ry;
    generate_camera_ray(*cam, mk2(p_film.x + eps, p_film.y), 0.0f, mk2(0.0f, 0.0f), rx);
    generate_camera_ray(*cam, mk2(p_film.x, p_film.y + eps), 0.0f, mk2(0.0f, 0.0f), ry);
    RayDiff r;
    r.rx_o = o_world + (mk3(rx.ox, rx.oy, rx.oz) - o_world) * inv; r.rx_d = d_world + (mk3(rx.dx, rx.dy, rx.dz) - d_world) * inv;
    r.ry_o = o_world + (mk3(ry.ox, ry.oy, ry.oz) - o_world) * inv; r.ry_d = d_world + (mk3(ry.dx, ry.dy, ry.dz) - d_world) * inv;
    const float sc = ph_div(1.0f, ph_sqrt((float)spp));
    r.rx_o = o_world + (r.rx_o - o_world) * sc; r.ry_o = o_world + (r.ry_o - o_world) * sc;
    r.rx_d = d_world + (r.rx_d - d_world) * sc; r.ry_d = d_world + (r.ry_d - d_world) * sc;
    *out = r;
}
PH_DEV RayDiff camera_ray_differentials(const CameraRec& cam, f2 p_film, f2 lens_s, f3 o_world, f3 d_world, uint32_t spp) {
    const f3 p_camera = xf_point_plain(cam.r2c, mk3(p_film.x, p_film.y, 0.0f));
    const f3 dxc = mk3(cam.dx_camera[0], cam.dx_camera[1], cam.dx_camera[2]), dyc = mk3(cam.dy_camera[0], cam.dy_camera[1], cam.dy_camera[2]);
    if (cam.kind == PH_CAM_ENVIRONMENT) { RayDiff r; environment_camera_differentials(&cam, p_film, &o_world, &d_world, spp, &r); return r; }
    f3 rx_o, ry_o, rx_d, ry_d;
    if (cam.kind == PH_CAM_ORTHOGRAPHIC) {  // orthographic_camera.rs:151-174
        if (cam.lens_radius > 0.0f) {
            const f2 cd = concentric_sample_disk(lens_s);
            const f2 p_lens = mk2(cam.lens_radius * cd.x, cam.lens_radius * cd.y);
            // the main ray once more (:137-149): `ft` below divides by ITS direction's z, after the lens moved it (:157)
            const f3 z1 = mk3(0.0f, 0.0f, 1.0f);
            float ft = ph_div(cam.focal_distance, z1.z);
            const f3 o_lens = mk3(p_lens.x, p_lens.y, 0.0f);
            const f3 d_main = normalize((p_camera + z1 * ft) - o_lens);
            ft = ph_div(cam.focal_distance, d_main.z);
            rx_o = o_lens; rx_d = normalize(((p_camera + dxc) + (ft * z1)) - rx_o);
            ry_o = o_lens; ry_d = normalize(((p_camera + dyc) + (ft * z1)) - ry_o);
        } else {
            rx_o = p_camera + dxc; ry_o = p_camera + dyc;
            rx_d = mk3(0.0f, 0.0f, 1.0f); ry_d = rx_d;
        }
    } else if (cam.lens_radius > 0.0f) {
        const f2 cd = concentric_sample_disk(lens_s);
        const f2 p_lens = mk2(cam.lens_radius * cd.x, cam.lens_radius * cd.y);
        const f3 dx = normalize(p_camera + dxc);
        float ft = ph_div(cam.focal_distance, dx.z);
        f3 p_focus = mk3(0.0f, 0.0f, 0.0f) + (ft * dx);
        rx_o = mk3(p_lens.x, p_lens.y, 0.0f); rx_d = normalize(p_focus - rx_o);
        const f3 dy = normalize(p_camera + dyc);
        ft = ph_div(cam.focal_distance, dy.z);
        p_focus = mk3(0.0f, 0.0f, 0.0f) + (ft * dy);
        ry_o = mk3(p_lens.x, p_lens.y, 0.0f); ry_d = normalize(p_focus - ry_o);
    } else {
        rx_o = mk3(0.0f, 0.0f, 0.0f); ry_o = rx_o;
        rx_d = normalize(p_camera + dxc); ry_d = normalize(p_camera + dyc);
    }
    RayDiff r;
    r.rx_o = xf_point_plain(cam.c2w, rx_o); r.ry_o = xf_point_plain(cam.c2w, ry_o);
    r.rx_d = xf_vec(cam.c2w, rx_d); r.ry_d = xf_vec(cam.c2w, ry_d);
    const float sc = ph_div(1.0f, ph_sqrt((float)spp));
    r.rx_o = o_world + (r.rx_o - o_world) * sc; r.ry_o = o_world + (r.ry_o - o_world) * sc;
    r.rx_d = d_world + (r.rx_d - d_world) * sc; r.ry_d = d_world + (r.ry_d - d_world) * sc;
    return r;
}
PH_DEV float comp3(f3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
// `dpdu`, `dpdv`: SurfaceInteraction.der (geometric); returns der.du/dv d x/y (zeros where the reference leaves zeros)
PH_DEV void compute_differentials(f3 p, f3 n, f3 dpdu, f3 dpdv, const RayDiff& rd, TexCtx& c) {
    c.dudx = c.dvdx = c.dudy = c.dvdy = 0.0f;
    c.dpdx = mk3(0.0f, 0.0f, 0.0f); c.dpdy = c.dpdx;
    const float d = dot(n, p);
    const float tx = ph_div(-(dot(n, rd.rx_o) - d), dot(n, rd.rx_d));
    if (__builtin_isinf(tx) || tx != tx) return;
    const f3 px = rd.rx_o + tx * rd.rx_d;
    const float ty = ph_div(-(dot(n, rd.ry_o) - d), dot(n, rd.ry_d));
    if (__builtin_isinf(ty) || ty != ty) return;
    const f3 py = rd.ry_o + ty * rd.ry_d;
    c.dpdx = px - p; c.dpdy = py - p;
    int d0, d1;
    if (pabs(n.x) > pabs(n.y) && pabs(n.x) > pabs(n.z)) { d0 = 1; d1 = 2; }
    else if (pabs(n.y) > pabs(n.z)) { d0 = 0; d1 = 2; }
    else { d0 = 0; d1 = 1; }
    const float a00 = comp3(dpdu, d0), a01 = comp3(dpdv, d0), a10 = comp3(dpdu, d1), a11 = comp3(dpdv, d1);
    const float bx0 = comp3(px, d0) - comp3(p, d0), bx1 = comp3(px, d1) - comp3(p, d1);
    const float by0 = comp3(py, d0) - comp3(p, d0), by1 = comp3(py, d1) - comp3(p, d1);
    const float det = a00 * a11 - a01 * a10;  // solve_linear_system_2x2 (matrix4x4.rs:305-318)
    if (pabs(det) < 1e-10f) return;
    float x0 = ph_div(a11 * bx0 - a01 * bx1, det), x1 = ph_div(a00 * bx1 - a10 * bx0, det);
    if (!(x0 != x0 || x1 != x1)) { c.dudx = x0; c.dvdx = x1; }
    x0 = ph_div(a11 * by0 - a01 * by1, det); x1 = ph_div(a00 * by1 - a10 * by0, det);
    if (!(x0 != x0 || x1 != x1)) { c.dudy = x0; c.dvdy = x1; }
}

// ---- InfiniteAreaLight with a radiance map (lights/src/infinite.rs:127-211), declared in pt_device.h -------------------------------------
// l_map.lookup_triangle(st, 0.0): the level is negative for every pyramid that fits memory, i.e. MIPMap::triangle(0, st)
static __device__ __noinline__ spec envmap_lookup(const DeviceScene* dsc, uint32_t mip, float s, float t) {
    const DeviceScene& sc = *dsc;
    return mip_triangle(sc, sc.mipmaps[mip], 0u, mk2(s, t));
}
// Distribution1D::sample_continuous (core/src/sampling/distribution_1d.rs:55-79) on an n-entry function
PH_DEV float distn_sample_continuous(const float* func, const float* cdf, uint32_t n, float func_int, float u, float& pdf, uint32_t& off) {
    const uint32_t offset = find_interval_cdf(cdf, n + 1u, u);
    float du = u - cdf[offset];
    if (cdf[offset + 1] - cdf[offset] > 0.0f) du = ph_div(du, cdf[offset + 1] - cdf[offset]);
    pdf = func_int > 0.0f ? ph_div(func[offset], func_int) : 0.0f;
    off = offset;
    return ph_div((float)offset + du, (float)n);
}
// Distribution2D::sample_continuous (distribution_2d.rs:31-49); returns the map pdf
static __device__ __noinline__ float envmap_sample(const DeviceScene* dsc, uint32_t dist_off, uint32_t dw, uint32_t dh, float u0, float u1, float* d0, float* d1) {
    const float* base = dsc->light_dist + dist_off;
    const float* cond_func = base; const float* cond_cdf = cond_func + (size_t)dw * dh; const float* cond_int = cond_cdf + (size_t)(dw + 1u) * dh;
    const float* marg_func = cond_int + dh; const float* marg_cdf = marg_func + dh; const float marg_int = marg_cdf[dh + 1u];
    float pdf1, pdf0; uint32_t v, dummy;
    *d1 = distn_sample_continuous(marg_func, marg_cdf, dh, marg_int, u1, pdf1, v);
    *d0 = distn_sample_continuous(cond_func + (size_t)v * dw, cond_cdf + (size_t)v * (dw + 1u), dw, cond_int[v], u0, pdf0, dummy);
    return pdf0 * pdf1;
}
// Distribution2D::pdf (distribution_2d.rs:51-65)
static __device__ __noinline__ float envmap_pdf(const DeviceScene* dsc, uint32_t dist_off, uint32_t dw, uint32_t dh, float px, float py) {
    const float* base = dsc->light_dist + dist_off;
    const float* marg_cdf = base + (size_t)dw * dh + (size_t)(dw + 1u) * dh + dh + dh;
    uint32_t iu = f2u_sat(px * (float)dw), iv = f2u_sat(py * (float)dh);
    if (iu > dw - 1u) iu = dw - 1u;
    if (iv > dh - 1u) iv = dh - 1u;
    return ph_div(base[(size_t)iv * dw + iu], marg_cdf[dh + 1u]);
}

// The texture-evaluation context of a hit, out of line.  Everything texture evaluation needs beyond what the integrator carries is rebuilt
// here from the TriRec the traversal reported: uv (triangle.rs:584), the geometric dp/du, dp/dv (:548-574, carried to world space for an
// instance: transform.rs:566-590) and — for camera rays only — du/dv d x/y.
PH_DEV TexCtx hit_tex_ctx(const DeviceScene* dsc, const CameraRec* cam, uint32_t spp, uint32_t tri_index, uint32_t inst,
                                                  f3 bary, f3 p, f3 n, f3 ro, f3 rd, f2 p_film, f2 lens, uint32_t camera_ray) {
    const DeviceScene& sc = *dsc;
    const float4* tp = reinterpret_cast<const float4*>(sc.tris + tri_index);
    const float4 a = tp[0], b = tp[1], c = tp[2];
    const uint32_t prim = __float_as_uint(a.w);
    const MeshRec m = sc.meshes[__float_as_uint(c.w)];
    TriVerts t;
    t.p0 = mk3(a.x, a.y, a.z); t.p1 = mk3(b.x, b.y, b.z); t.p2 = mk3(c.x, c.y, c.z);
    t.i0 = t.i1 = t.i2 = 0;
    f2 uv0 = mk2(0.0f, 0.0f), uv1 = mk2(1.0f, 0.0f), uv2 = mk2(1.0f, 1.0f);
    if (m.flags & PH_MESH_UV) {
        t.i0 = sc.idx[3 * prim]; t.i1 = sc.idx[3 * prim + 1]; t.i2 = sc.idx[3 * prim + 2];
        uv0 = mk2(sc.UV[2 * (size_t)t.i0], sc.UV[2 * (size_t)t.i0 + 1]);
        uv1 = mk2(sc.UV[2 * (size_t)t.i1], sc.UV[2 * (size_t)t.i1 + 1]);
        uv2 = mk2(sc.UV[2 * (size_t)t.i2], sc.UV[2 * (size_t)t.i2 + 1]);
    }
    f3 dpdu, dpdv;
    tri_dpdu(sc, m, t, dpdu, dpdv);
    if (inst != 0u) {
        const InstRec& I = sc.instances[inst - 1u];
        if (!(I.flags & PH_INST_IDENTITY)) { dpdu = xf_vec(I.i2w, dpdu); dpdv = xf_vec(I.i2w, dpdv); }
    }
    TexCtx ctx;
    ctx.uv = mk2((bary.x * uv0.x + bary.y * uv1.x) + bary.z * uv2.x, (bary.x * uv0.y + bary.y * uv1.y) + bary.z * uv2.y);
    ctx.dudx = ctx.dvdx = ctx.dudy = ctx.dvdy = 0.0f;
    ctx.p = p; ctx.dpdx = mk3(0.0f, 0.0f, 0.0f); ctx.dpdy = ctx.dpdx;
    if (camera_ray) {
        const CameraRec cm = *cam;
        const RayDiff rdf = camera_ray_differentials(cm, p_film, lens, ro, rd, spp);
        compute_differentials(p, n, dpdu, dpdv, rdf, ctx);
    }
    return ctx;
}
// `tex.evaluate(..).clamp_default()` (matte.rs:63, plastic.rs:62-70, mirror.rs:53, substrate.rs:60-61)
template <bool SIMPLE = false, bool NODIFF = false>
PH_DEV spec tex_eval_clamped(const DeviceScene* dsc, uint32_t tex, const TexCtx& ctx) {
    const spec v = tex_eval<SIMPLE, NODIFF>(dsc, tex, ctx);
    return mks(pclampf(v.r, 0.0f, kInf), pclampf(v.g, 0.0f, kInf), pclampf(v.b, 0.0f, kInf));
}
// texture pass: the textured colours of a material's template lobes at this hit, in lobe order (r before t), clamped and pre-multiplied
PH_DEV float d_log(float x) { return (float)log((double)x); }
// the lobe's scalar parameters when they are textures: MatteMaterial's sigma -> Oren-Nayar A, B (matte.rs:64-70, oren_nayar.rs:28-39); roughness -> Trowbridge-Reitz alpha,
// remapped per hit (trowbridge_reitz.rs:21-40).  One parametrised lobe per material: the values travel in out.col[0][3], out.col[1][3]
template <bool SIMPLE = false, bool NODIFF = false>
PH_DEV void eval_lobe_scalars(const DeviceScene* dsc, const LobeRec& l, const TexCtx& ctx, TexOut& out) {
    if (l.sigma_tex1) {
        const float sig = pclampf(tex_eval<SIMPLE, NODIFF>(dsc, l.sigma_tex1 - 1u, ctx).r, 0.0f, 90.0f);
        if (sig == 0.0f) { out.lambert |= 1u; out.col[0][3] = 0.0f; out.col[1][3] = 0.0f; }
        else { const float sg = sig * (kPi / 180.0f), s2 = sg * sg; out.col[0][3] = 1.0f - ph_div(s2, 2.0f * (s2 + 0.33f)); out.col[1][3] = ph_div(0.45f * s2, s2 + 0.09f); }
    }
    if (l.ax_tex1 || l.ay_tex1) {
        float a[2] = {l.ax, l.ay};
        float raw[2] = {l.ur_raw, l.vr_raw};   // GlassMaterial: `is_specular = urough == 0 && vrough == 0` on the values as the textures give them (glass.rs:111)
        const uint32_t tx[2] = {l.ax_tex1, l.ay_tex1};
        for (int k = 0; k < 2; k++) if (tx[k]) {
            float r = tex_eval<SIMPLE, NODIFF>(dsc, tx[k] - 1u, ctx).r;
            raw[k] = r;
            if (l.remap) { r = pmaxf(r, 1e-3f); const float x = d_log(r); r = 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x; }
            a[k] = pmaxf(0.001f, r);
        }
        out.col[0][3] = a[0]; out.col[1][3] = a[1];
        if (l.alt && raw[0] == 0.0f && raw[1] == 0.0f) out.lambert |= 2u;
    }
}
template <bool SIMPLE = false, bool NODIFF = false>
PH_DEV void eval_lobe_colours(const DeviceScene* dsc, const MaterialRec& mr, const LobeRec* tmpl, uint32_t n, const TexCtx& ctx, TexOut& out) {
    uint32_t k = 0;
    auto put = [&](spec c) { if (k < PH_HIT_COLS) { out.col[k][0] = c.r; out.col[k][1] = c.g; out.col[k][2] = c.b; k++; } };
    if (mr.index_tex1) out.col[2][3] = tex_eval<SIMPLE, NODIFF>(dsc, mr.index_tex1 - 1u, ctx).r;                         // glass.rs:102 / uber.rs:128: as the texture gives it
    if (mr.amount_tex1) put(tex_eval_clamped<SIMPLE, NODIFF>(dsc, mr.amount_tex1 - 1u, ctx));                           // mix.rs:59: s1 (s2 is made from it in the shade pass)
    spec op = mks1(1.0f);
    if (mr.opacity_tex1) op = tex_eval_clamped<SIMPLE, NODIFF>(dsc, mr.opacity_tex1 - 1u, ctx);                          // uber.rs:126
    spec rt_r = mks1(0.0f), rt_t = mks1(0.0f);
    if (mr.rt_mode) {   // translucent.rs:70-74: reflect and transmit of this hit; both black -> the hit has no BSDF
        rt_r = mr.refl_tex1 ? tex_eval_clamped<SIMPLE, NODIFF>(dsc, mr.refl_tex1 - 1u, ctx) : mks(mr.refl_c[0], mr.refl_c[1], mr.refl_c[2]);
        rt_t = mr.trans_tex1 ? tex_eval_clamped<SIMPLE, NODIFF>(dsc, mr.trans_tex1 - 1u, ctx) : mks(mr.trans_c[0], mr.trans_c[1], mr.trans_c[2]);
        if (is_black(rt_r) && is_black(rt_t)) { out.bumped |= PH_TEXOUT_NULL_BSDF; return; }
    }
    for (uint32_t i = 0; i < n; i++) {
        const LobeRec& l = tmpl[i];
        if (l.sigma_tex1 || l.ax_tex1 || l.ay_tex1) eval_lobe_scalars<SIMPLE, NODIFF>(dsc, l, ctx, out);
        if (l.has_pre == PH_PRE_RT) {   // `if !kd.is_black() { if !r.is_black() { add(r * kd) } if !t.is_black() { add(t * kd) } }` and the same with Ks (translucent.rs:76-98)
            const bool refl = lobe_is_reflection(l);
            const spec A = refl ? rt_r : rt_t;
            const uint32_t tex = refl ? l.r_tex1 : l.t_tex1;
            const spec B = tex ? tex_eval_clamped<SIMPLE, NODIFF>(dsc, tex - 1u, ctx) : mks(l.pre[0], l.pre[1], l.pre[2]);
            if ((is_black(A) || is_black(B)) && k < PH_HIT_COLS) out.bumped |= 1u << (8u + k);
            put(A * B);
            continue;
        }
        if (l.has_pre == PH_PRE_PASSTHROUGH) { const spec t = op * -1.0f + mks1(1.0f); put(mks(pclampf(t.r, 0.0f, kInf), pclampf(t.g, 0.0f, kInf), pclampf(t.b, 0.0f, kInf))); continue; }   // (-op + ONE).clamp_default() (uber.rs:127)
        if (l.has_pre == PH_PRE_OPACITY) {   // op * k.evaluate(..).clamp_default() (uber.rs:141, :147, :169, :175)
            const uint32_t tex = l.kind == PH_LK_SPEC_T ? l.t_tex1 : l.r_tex1;
            const spec base = tex ? tex_eval_clamped<SIMPLE, NODIFF>(dsc, tex - 1u, ctx) : mks(l.pre[0], l.pre[1], l.pre[2]);
            put(op * base);
        } else {
            const spec pre = l.has_pre ? mks(l.pre[0], l.pre[1], l.pre[2]) : mks1(1.0f);
            const uint32_t texs[2] = {l.r_tex1, l.t_tex1};
            for (int f = 0; f < 2; f++) if (texs[f] && k < PH_HIT_COLS) {
                spec c = tex_eval_clamped<SIMPLE, NODIFF>(dsc, texs[f] - 1u, ctx);
                if (l.has_pre == PH_PRE_RAW_TEST && c.r == 0.0f && c.g == 0.0f && c.b == 0.0f) out.bumped |= 1u << (8u + k);
                if (l.has_pre) c = pre * c;
                put(c);
            }
        }
        if (l.eta_tex1) put(tex_eval<SIMPLE, NODIFF>(dsc, l.eta_tex1 - 1u, ctx));   // metal.rs:121-125: as the textures give them, no clamp
        if (l.k_tex1) put(tex_eval<SIMPLE, NODIFF>(dsc, l.k_tex1 - 1u, ctx));
    }
}
// (shade pass: the hit's lobe list is the template seen through this record — bsdf_general.h: patch_lobe / hit_lobe_mask use the same slot rules)

// Material::bump (core/src/material.rs:62-101): the displacement texture is evaluated at the hit and at two shifted contexts, the shading dp/du,
// dp/dv are tilted accordingly and set_shading_geometry(.., false) (surface_interaction.rs:152-173) makes the new shading normal.  Out of line; the shading frame's other half (dp/dv, dn/du, dn/dv: triangle.rs:631-721), which the integrator does not carry, is rebuilt from the TriRec.
struct BumpOut { f3 ns, dpdu_s; };
struct BumpIn { uint32_t tex, tri_index, inst; f3 bary, p, n, ns, dpdu_s; TexCtx c; };
// arguments travel through one private struct: with ~40 scalar arguments (most of them on the stack) this function, out of line, corrupted values of OTHER
// lanes of the wave in the one-lobe kernel on gfx950
template <bool SIMPLE = false, bool NODIFF = false>
PH_DEV void hit_bump(const DeviceScene* dsc, const BumpIn* in, BumpOut* out) {
    const DeviceScene& sc = *dsc;
    const uint32_t tex = in->tex, tri_index = in->tri_index, inst = in->inst;
    const f3 bary = in->bary, p = in->p, n = in->n, ns = in->ns, dpdu_s = in->dpdu_s;
    const TexCtx c = in->c;
    const float4* tp = reinterpret_cast<const float4*>(sc.tris + tri_index);
    const float4 a = tp[0], b = tp[1], cc = tp[2];
    const uint32_t prim = __float_as_uint(a.w);
    const MeshRec m = sc.meshes[__float_as_uint(cc.w)];
    TriVerts t;
    t.p0 = mk3(a.x, a.y, a.z); t.p1 = mk3(b.x, b.y, b.z); t.p2 = mk3(cc.x, cc.y, cc.z);
    t.i0 = t.i1 = t.i2 = 0;
    if (m.flags & (PH_MESH_N | PH_MESH_S | PH_MESH_UV)) { t.i0 = sc.idx[3 * prim]; t.i1 = sc.idx[3 * prim + 1]; t.i2 = sc.idx[3 * prim + 2]; }
    f3 dpdu, dpdv;
    tri_dpdu(sc, m, t, dpdu, dpdv);
    f3 dpdv_s = dpdv, dndu = mk3(0.0f, 0.0f, 0.0f), dndv = dndu;   // SurfaceInteraction::new: shading = geometric, dn/du = dn/dv = 0 for triangles
    if (m.flags & (PH_MESH_N | PH_MESH_S)) {
        f3 ng = normalize(cross(t.p0 - t.p2, t.p1 - t.p2));
        const bool rev = (m.flags & PH_MESH_REV) != 0, swp = (m.flags & PH_MESH_SWAP) != 0;
        if (rev != swp) ng = -ng;
        f3 nsh = ng, n0 = ng, n1 = ng, n2 = ng;
        if (m.flags & PH_MESH_N) {
            n0 = ld3(sc.N + 3 * (size_t)t.i0); n1 = ld3(sc.N + 3 * (size_t)t.i1); n2 = ld3(sc.N + 3 * (size_t)t.i2);
            const f3 ns2 = bary.x * n0 + bary.y * n1 + bary.z * n2;
            nsh = length_squared(ns2) > 0.0f ? normalize(ns2) : ng;
        }
        f3 ss;
        if (m.flags & PH_MESH_S) {
            const f3 ss2 = bary.x * ld3(sc.S + 3 * (size_t)t.i0) + bary.y * ld3(sc.S + 3 * (size_t)t.i1) + bary.z * ld3(sc.S + 3 * (size_t)t.i2);
            ss = length_squared(ss2) > 0.0f ? normalize(ss2) : normalize(dpdu);
        } else ss = normalize(dpdu);
        f3 ts = cross(ss, nsh);
        if (length_squared(ts) > 0.0f) { ts = normalize(ts); ss = cross(ts, nsh); }
        else coordinate_system(nsh, ss, ts);
        if (m.flags & PH_MESH_N) {  // dn/du, dn/dv (triangle.rs:681-715)
            f2 uv0 = mk2(0.0f, 0.0f), uv1 = mk2(1.0f, 0.0f), uv2 = mk2(1.0f, 1.0f);
            if (m.flags & PH_MESH_UV) {
                uv0 = mk2(sc.UV[2 * (size_t)t.i0], sc.UV[2 * (size_t)t.i0 + 1]); uv1 = mk2(sc.UV[2 * (size_t)t.i1], sc.UV[2 * (size_t)t.i1 + 1]);
                uv2 = mk2(sc.UV[2 * (size_t)t.i2], sc.UV[2 * (size_t)t.i2 + 1]);
            }
            const f2 duv02 = mk2(uv0.x - uv2.x, uv0.y - uv2.y), duv12 = mk2(uv1.x - uv2.x, uv1.y - uv2.y);
            const f3 dn1 = n0 - n2, dn2 = n1 - n2;
            const float determinant = duv02.x * duv12.y - duv02.y * duv12.x;
            if (fabsf(determinant) < 1e-8f) {
                const f3 dn = cross(n2 - n0, n1 - n0);
                if (length_squared(dn) != 0.0f) coordinate_system(dn, dndu, dndv);
            } else {
                const float invdet = ph_div(1.0f, determinant);
                dndu = (duv12.y * dn1 - duv02.y * dn2) * invdet;
                dndv = (-duv12.x * dn1 + duv02.x * dn2) * invdet;
            }
        }
        if (rev) ts = -ts;
        dpdv_s = ts;
    }
    if (inst != 0u) {  // transform_surface_interaction (transform.rs:566-590)
        const InstRec& I = sc.instances[inst - 1u];
        if (!(I.flags & PH_INST_IDENTITY)) { dpdv_s = xf_vec(I.i2w, dpdv_s); dndu = xf_normal(I.w2i, dndu); dndv = xf_normal(I.w2i, dndv); }
    }
    float du = 0.5f * (pabs(c.dudx) + pabs(c.dudy));
    if (du == 0.0f) du = 0.0005f;
    TexCtx cu = c; cu.p = p + du * dpdu_s; cu.uv = mk2(c.uv.x + du, c.uv.y + 0.0f);
    const float u_displace = tex_eval<SIMPLE, NODIFF>(dsc, tex, cu).r;
    float dv = 0.5f * (pabs(c.dvdx) + pabs(c.dvdy));
    if (dv == 0.0f) dv = 0.0005f;
    TexCtx cv = c; cv.p = p + dv * dpdv_s; cv.uv = mk2(c.uv.x + 0.0f, c.uv.y + dv);
    const float v_displace = tex_eval<SIMPLE, NODIFF>(dsc, tex, cv).r;
    const float displace = tex_eval<SIMPLE, NODIFF>(dsc, tex, c).r;
    const f3 ndpdu = dpdu_s + ph_div(u_displace - displace, du) * ns + displace * dndu;
    const f3 ndpdv = dpdv_s + ph_div(v_displace - displace, dv) * ns + displace * dndv;
    out->ns = face_forward(normalize(cross(ndpdu, ndpdv)), n);
    out->dpdu_s = ndpdu;
}

// The lean form of the alpha test (traverse.h, ALPHA = 1): texture programs of constants, uv-mapped image maps, scale and mix, evaluated in the red channel only — every one
// of those operations works channel by channel (mix's amount is its operand's red channel, mix.rs:36-41), and the test reads the red channel (a float texture's three are
// equal).  The value stack lives in six registers that shift, so nothing is indexed dynamically.
PH_DEV float alpha_prog_r(const DeviceScene& sc, uint32_t id, f2 uv) {
    const TexRec tr = sc.textures[id];
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f, s4 = 0.0f, s5 = 0.0f;   // s0 = top of the stack
    for (uint32_t k = 0; k < tr.n_ops; k++) {
        const TexOp& op = sc.tex_ops[tr.first_op + k];
        const uint32_t o = op.op;
        if (o == PH_TOP_CONST || o == PH_TOP_IMAGE) {
            float v = op.c[0];
            if (o == PH_TOP_IMAGE) v = mip_triangle(sc, sc.mipmaps[op.mip], 0u, mk2(op.su * uv.x + op.du, op.sv * uv.y + op.dv)).r;   // UVMapping2D::map (uv_2d.rs:52-60); no differentials: MIPMap::lookup ends in triangle(0, st)
            s5 = s4; s4 = s3; s3 = s2; s2 = s1; s1 = s0; s0 = v;
        } else if (o == PH_TOP_MUL) { s0 = s1 * s0; s1 = s2; s2 = s3; s3 = s4; s4 = s5; }                                  // scale.rs:33: tex1 * tex2
        else { s0 = (1.0f - s0) * s2 + s0 * s1; s1 = s3; s2 = s4; s3 = s5; }                                                 // mix.rs:36-41: stack = t1, t2, amount
    }
    return s0;
}
static __device__ __forceinline__ bool alpha_accept_lean(const DeviceScene& sc, uint32_t prim, uint32_t mesh, float b0, float b1, float b2, bool any_hit) {
    const MeshRec& m = sc.meshes[mesh];
    const uint32_t mflags = m.flags, at = m.alpha_tex1, st = m.shadow_alpha_tex1;
    f2 uv0 = mk2(0.0f, 0.0f), uv1 = mk2(1.0f, 0.0f), uv2 = mk2(1.0f, 1.0f);
    if (mflags & PH_MESH_UV) {
        const uint32_t i0 = sc.idx[3 * prim], i1 = sc.idx[3 * prim + 1], i2 = sc.idx[3 * prim + 2];
        uv0 = mk2(sc.UV[2 * (size_t)i0], sc.UV[2 * (size_t)i0 + 1]); uv1 = mk2(sc.UV[2 * (size_t)i1], sc.UV[2 * (size_t)i1 + 1]); uv2 = mk2(sc.UV[2 * (size_t)i2], sc.UV[2 * (size_t)i2 + 1]);
    }
    const f2 uv = mk2((b0 * uv0.x + b1 * uv1.x) + b2 * uv2.x, (b0 * uv0.y + b1 * uv1.y) + b2 * uv2.y);
    if (at && alpha_prog_r(sc, at - 1u, uv) == 0.0f) return false;
    if (any_hit && st && alpha_prog_r(sc, st - 1u, uv) == 0.0f) return false;
    return true;
}

static __device__ __forceinline__ void alpha_general_prepare() { noise_lds_fill(); }
// The alpha-mask test of Triangle::intersect / intersect_p (triangle.rs:587-607, 868-898), declared in traverse.h: isect_local carries the hit point
// (in the triangle's own space), the interpolated uv and no differentials.
static __device__ __noinline__ bool alpha_accept(const DeviceScene* dsc, uint32_t tri_index, float b0, float b1, float b2, uint32_t any_hit) {
    const DeviceScene& sc = *dsc;
    const float4* tp = reinterpret_cast<const float4*>(sc.tris + tri_index);
    const float4 a = tp[0], b = tp[1], c = tp[2];
    const uint32_t prim = __float_as_uint(a.w);
    const MeshRec m = sc.meshes[__float_as_uint(c.w)];
    f2 uv0 = mk2(0.0f, 0.0f), uv1 = mk2(1.0f, 0.0f), uv2 = mk2(1.0f, 1.0f);
    if (m.flags & PH_MESH_UV) {
        const uint32_t i0 = sc.idx[3 * prim], i1 = sc.idx[3 * prim + 1], i2 = sc.idx[3 * prim + 2];
        uv0 = mk2(sc.UV[2 * (size_t)i0], sc.UV[2 * (size_t)i0 + 1]); uv1 = mk2(sc.UV[2 * (size_t)i1], sc.UV[2 * (size_t)i1 + 1]); uv2 = mk2(sc.UV[2 * (size_t)i2], sc.UV[2 * (size_t)i2 + 1]);
    }
    TexCtx ctx;
    ctx.uv = mk2((b0 * uv0.x + b1 * uv1.x) + b2 * uv2.x, (b0 * uv0.y + b1 * uv1.y) + b2 * uv2.y);
    ctx.dudx = ctx.dvdx = ctx.dudy = ctx.dvdy = 0.0f;
    ctx.p = b0 * mk3(a.x, a.y, a.z) + b1 * mk3(b.x, b.y, b.z) + b2 * mk3(c.x, c.y, c.z);
    ctx.dpdx = mk3(0.0f, 0.0f, 0.0f); ctx.dpdy = ctx.dpdx;
    // (no differentials here: the evaluator's NODIFF form, without the filtered look-ups' code)
    if (m.alpha_tex1 && tex_eval<false, true>(dsc, m.alpha_tex1 - 1u, ctx).r == 0.0f) return false;
    if (any_hit && m.shadow_alpha_tex1 && tex_eval<false, true>(dsc, m.shadow_alpha_tex1 - 1u, ctx).r == 0.0f) return false;
    return true;
}

}  // namespace ph
