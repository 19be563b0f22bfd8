// Host-side scalar helpers in the reference's f32 expression order (compiled with -ffp-contract=off).
// Used for the one-off setup work that stays on the host: light tables, sampler tables, camera/film set-up.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace hm {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kInf = __builtin_huge_valf();

inline float fabs_p(float n) { return n < 0.0f ? -n : n; }           // core/src/pbrt/common.rs:66-76
inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }  // clamp.rs:9-20
inline float to_radians(float d) { return d * (kPi / 180.0f); }

struct V3 { float x, y, z; };
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 mulf(V3 a, float f) { return {a.x * f, a.y * f, a.z * f}; }
inline V3 divf(V3 a, float f) { float inv = 1.0f / f; return {inv * a.x, inv * a.y, inv * a.z}; }  // vector3.rs:406-418
inline V3 cross(V3 a, V3 b) { return {(a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)}; }
inline float len2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
inline float len(V3 a) { return std::sqrt(len2(a)); }
inline V3 normalize(V3 a) { return divf(a, len(a)); }
inline void coordinate_system(V3 v1, V3& v2, V3& v3) {  // core/src/geometry/coordinate_system.rs:12-20
    if (std::fabs(v1.x) > std::fabs(v1.y)) v2 = divf({-v1.z, 0.0f, v1.x}, std::sqrt(v1.x * v1.x + v1.z * v1.z));
    else v2 = divf({0.0f, v1.z, -v1.y}, std::sqrt(v1.y * v1.y + v1.z * v1.z));
    v3 = cross(v1, v2);
}
inline V3 ld(const float* p) { return {p[0], p[1], p[2]}; }

// PCG32 (core/src/rng.rs:20-120)
struct Pcg32 {
    uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;  // RNG::default()
    uint32_t next() {
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27), rot = (uint32_t)(old >> 59);
        return (xs >> rot) | (xs << ((~rot + 1) & 31));
    }
    uint32_t bounded(uint32_t lo, uint32_t hi) {
        uint32_t b = hi - lo, threshold = (~b + 1) % b;
        for (;;) { uint32_t r = next(); if (r >= threshold) return lo + r % b; }
    }
};

// radical_inverse (core/src/low_discrepency.rs:401-421, 454-464): base 2 by bit reversal, other bases digit by digit
inline float radical_inverse(uint32_t base, uint64_t a) {
    if (base == 2) {
        uint64_t n = a;
        n = ((n >> 1) & 0x5555555555555555ULL) | ((n & 0x5555555555555555ULL) << 1);
        n = ((n >> 2) & 0x3333333333333333ULL) | ((n & 0x3333333333333333ULL) << 2);
        n = ((n >> 4) & 0x0f0f0f0f0f0f0f0fULL) | ((n & 0x0f0f0f0f0f0f0f0fULL) << 4);
        n = ((n >> 8) & 0x00ff00ff00ff00ffULL) | ((n & 0x00ff00ff00ff00ffULL) << 8);
        n = ((n >> 16) & 0x0000ffff0000ffffULL) | ((n & 0x0000ffff0000ffffULL) << 16);
        n = (n >> 32) | (n << 32);
        return (float)n * 0x1.0p-64f;
    }
    const float inv_base = 1.0f / (float)base;
    uint64_t reversed = 0; float inv_base_n = 1.0f;
    while (a != 0) {
        const uint64_t next = a / base, digit = a - next * base;
        reversed = reversed * base + digit;
        inv_base_n *= inv_base;
        a = next;
    }
    const float v = (float)reversed * inv_base_n;
    return v < 0x1.fffffep-1f ? v : 0x1.fffffep-1f;
}

// PRIMES / PRIME_SUMS (core/src/low_discrepency.rs:13,102) and the Halton digit permutations
// (compute_radical_inverse_permutations :1512-1528 seeded with RNG::default(), samplers/src/halton.rs:16-19)
struct HaltonTables {
    std::vector<uint32_t> primes, prime_sums;
    std::vector<uint16_t> perms;
    std::vector<uint64_t> magic;  // ceil(2^64 / p): floor(a / p) == umul64hi(a, magic) for every a < 2^32
    void build() {
        primes.clear(); prime_sums.clear();
        for (uint32_t c = 2; primes.size() < 1000; c++) {
            bool is_p = true;
            for (uint32_t q : primes) { if (q * q > c) break; if (c % q == 0) { is_p = false; break; } }
            if (is_p) primes.push_back(c);
        }
        uint32_t total = 0;
        magic.clear();
        for (uint32_t p : primes) { prime_sums.push_back(total); total += p; magic.push_back((uint64_t)(((unsigned __int128)1 << 64) / p) + 1); }
        perms.assign(total, 0);
        Pcg32 rng; size_t at = 0;
        for (uint32_t p : primes) {
            for (uint32_t j = 0; j < p; j++) perms[at + j] = (uint16_t)j;
            for (uint32_t j = 0; j < p; j++) {  // RNG::shuffle (rng.rs:110-119)
                uint32_t other = j + rng.bounded(0, p - j);
                uint16_t t = perms[at + j]; perms[at + j] = perms[at + other]; perms[at + other] = t;
            }
            at += p;
        }
    }
};

// Distribution1D::new (core/src/sampling/distribution_1d.rs:22-45)
inline void distribution1d(const std::vector<float>& f, std::vector<float>& cdf, float& func_int) {
    size_t n = f.size();
    cdf.assign(n + 1, 0.0f);
    for (size_t i = 1; i < n + 1; i++) cdf[i] = cdf[i - 1] + f[i - 1] / (float)n;
    func_int = cdf[n];
    if (func_int == 0.0f) for (size_t i = 1; i < n + 1; i++) cdf[i] = (float)i / (float)n;
    else for (size_t i = 1; i < n + 1; i++) cdf[i] /= func_int;
}

// 4x4 matrices, row-major (core/src/geometry/matrix4x4.rs)
struct M4 { float m[4][4]; };
inline M4 m4_identity() { M4 r; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = i == j ? 1.0f : 0.0f; return r; }
inline M4 m4_mul(const M4& a, const M4& b) {  // :181-201
    M4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}
inline M4 m4_transpose(const M4& a) { M4 r; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = a.m[j][i]; return r; }
M4 m4_inverse(const M4& a);  // Gauss-Jordan with full pivoting (:67-143)

}  // namespace hm
