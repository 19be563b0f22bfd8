// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).
// Samplers, camera, lights, BSDF, PathIntegrator::li, Film, SamplerIntegrator::render.
#pragma once
#include "oracle_scene.hpp"
#include <memory>
#include <mutex>
#include <thread>

namespace orc {

// =============================== RNG (core/src/rng.rs:20-120) =====================================================
struct RNG {
    uint64_t state, inc;
    RNG() : state(0x853c49e6748fea9bULL), inc(0xda3e39cb94b95bdbULL) {}
    explicit RNG(uint64_t seq) { set_sequence(seq); }
    void set_sequence(uint64_t init_seq) {
        state = 0; inc = (init_seq << 1) | 1;
        uniform_u32();
        state += 0x853c49e6748fea9bULL;
        uniform_u32();
    }
    uint32_t uniform_u32() {
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27);
        uint32_t rot = (uint32_t)(old >> 59);
        return (xs >> rot) | (xs << ((~rot + 1) & 31));
    }
    uint32_t bounded_uniform_u32(uint32_t lo, uint32_t hi) {
        uint32_t b = hi - lo, threshold = (~b + 1) % b;
        for (;;) { uint32_t r = uniform_u32(); if (r >= threshold) return lo + r % b; }
    }
    Float uniform_float() { return pmin((Float)uniform_u32() * 0x1.0p-32f, ONE_MINUS_EPSILON); }
};

// =============================== low-discrepancy (core/src/low_discrepency.rs) ====================================
static const int PRIME_TABLE_SIZE = 1000;
struct LowDiscrepancyTables {
    std::vector<uint32_t> primes, prime_sums;
    std::vector<uint16_t> perms;  // compute_radical_inverse_permutations with RNG::default() (samplers/src/halton.rs:16-19)
    LowDiscrepancyTables() {
        for (uint32_t c = 2; primes.size() < (size_t)PRIME_TABLE_SIZE; c++) {
            bool is_p = true;
            for (uint32_t q : primes) { if (q * q > c) break; if (c % q == 0) { is_p = false; break; } }
            if (is_p) primes.push_back(c);
        }
        uint32_t s = 0;
        for (int i = 0; i < PRIME_TABLE_SIZE; i++) { prime_sums.push_back(s); s += primes[i]; }
        perms.resize(s);
        RNG rng; size_t p = 0;
        for (int i = 0; i < PRIME_TABLE_SIZE; i++) {  // low_discrepency.rs:1512-1528 + RNG::shuffle (rng.rs:110-119)
            uint32_t n = primes[i];
            for (uint32_t j = 0; j < n; j++) perms[p + j] = (uint16_t)j;
            for (uint32_t j = 0; j < n; j++) {
                uint32_t other = j + rng.bounded_uniform_u32(0, n - j);
                std::swap(perms[p + j], perms[p + other]);
            }
            p += n;
        }
    }
};
inline const LowDiscrepancyTables& ld_tables() { static LowDiscrepancyTables t; return t; }

inline uint64_t reverse_bits_64(uint64_t n) {
    n = ((n >> 1) & 0x5555555555555555ULL) | ((n & 0x5555555555555555ULL) << 1);
    n = ((n >> 2) & 0x3333333333333333ULL) | ((n & 0x3333333333333333ULL) << 2);
    n = ((n >> 4) & 0x0f0f0f0f0f0f0f0fULL) | ((n & 0x0f0f0f0f0f0f0f0fULL) << 4);
    n = ((n >> 8) & 0x00ff00ff00ff00ffULL) | ((n & 0x00ff00ff00ff00ffULL) << 8);
    n = ((n >> 16) & 0x0000ffff0000ffffULL) | ((n & 0x0000ffff0000ffffULL) << 16);
    return (n >> 32) | (n << 32);
}
inline Float radical_inverse_specialized(uint32_t base_, uint64_t a) {  // :401-421
    Float inv_base = 1.0f / (Float)base_;
    uint64_t base = base_, reversed = 0;
    Float inv_base_n = 1.0f;
    while (a != 0) {
        uint64_t next = a / base, digit = a - next * base;
        reversed = reversed * base + digit;
        inv_base_n *= inv_base;
        a = next;
    }
    return pmin((Float)reversed * inv_base_n, ONE_MINUS_EPSILON);
}
inline Float radical_inverse(int base_index, uint64_t a) {  // :454-464 (base 2: no clamp)
    if (base_index == 0) return (Float)reverse_bits_64(a) * 0x1.0p-64f;
    return radical_inverse_specialized(ld_tables().primes[base_index], a);
}
inline Float scrambled_radical_inverse(int base_index, uint64_t a, const uint16_t* perm) {  // :428-449
    uint32_t base_ = ld_tables().primes[base_index];
    Float inv_base = 1.0f / (Float)base_;
    uint64_t base = base_, reversed = 0;
    Float inv_base_n = 1.0f;
    while (a != 0) {
        uint64_t next = a / base; size_t digit = (size_t)(a - next * base);
        reversed = reversed * base + perm[digit];
        inv_base_n *= inv_base;
        a = next;
    }
    return pmin(inv_base_n * ((Float)reversed + inv_base * (Float)perm[0] / (1.0f - inv_base)), ONE_MINUS_EPSILON);
}
inline uint64_t inverse_radical_inverse(uint64_t base, uint64_t inverse, uint64_t n_digits) {  // :1535-1545
    uint64_t index = 0;
    for (uint64_t i = 0; i < n_digits; i++) { uint64_t digit = inverse % base; inverse /= base; index = index * base + digit; }
    return index;
}

// =============================== Samplers =========================================================================
struct SobolTables { const uint32_t* m32 = nullptr; const uint64_t* vdc = nullptr; const uint64_t* vdc_inv = nullptr; };

struct SamplerConfig {
    int kind = 0;  // 0 halton, 1 sobol, 2 random (CPU only)
    uint32_t spp = 16;
    int bounds[4] = {0, 0, 0, 0};  // sample_bounds x0,y0,x1,y1
    bool at_center = false;
    SobolTables sobol;
};

// HaltonSampler (samplers/src/halton.rs) as a per-tile object exactly like clone_sampler() produces
struct HaltonSampler {
    uint64_t base_scales[2], base_exponents[2], sample_stride;
    int64_t mult_inverse[2];
    bool at_center;
    uint32_t spp;
    int px = 0x7fffffff, py = 0x7fffffff;  // pixel_for_offset
    int cur_x = 0, cur_y = 0;
    uint64_t offset_for_pixel = 0, interval_index = 0;
    uint32_t dimension = 0, sample_num = 0;
    static void extended_gcd(uint64_t a, uint64_t b, int64_t& x, int64_t& y) {  // halton.rs:294-302
        if (b == 0) { x = 1; y = 0; return; }
        int64_t d = (int64_t)(a / b), xp, yp;
        extended_gcd(b, a % b, xp, yp);
        x = yp; y = xp - (d * yp);
    }
    static uint64_t multiplicative_inverse(int64_t a, int64_t n) { int64_t x, y; extended_gcd((uint64_t)a, (uint64_t)n, x, y); return (uint64_t)prem<int64_t>(x, n); }
    HaltonSampler(const SamplerConfig& c) : at_center(c.at_center), spp(c.spp) {  // halton.rs:61-100
        int res[2] = {c.bounds[2] - c.bounds[0], c.bounds[3] - c.bounds[1]};
        for (int i = 0; i < 2; i++) {
            uint64_t base = i == 0 ? 2 : 3, scale = 1, e = 0;
            while ((int)scale < pmin(res[i], 128)) { scale *= base; e++; }
            base_scales[i] = scale; base_exponents[i] = e;
        }
        sample_stride = base_scales[0] * base_scales[1];
        mult_inverse[0] = (int64_t)multiplicative_inverse((int64_t)base_scales[1], (int64_t)base_scales[0]);
        mult_inverse[1] = (int64_t)multiplicative_inverse((int64_t)base_scales[0], (int64_t)base_scales[1]);
    }
    uint64_t index_for_sample(uint64_t s) {  // halton.rs:118-144
        if (cur_x != px || cur_y != py) {
            offset_for_pixel = 0;
            if (sample_stride > 1) {
                int pm[2] = {prem(cur_x, 128), prem(cur_y, 128)};
                for (int i = 0; i < 2; i++) {
                    uint64_t dim_offset = inverse_radical_inverse(i == 0 ? 2 : 3, (uint64_t)pm[i], base_exponents[i]);
                    offset_for_pixel += dim_offset * (sample_stride / base_scales[i]) * (uint64_t)mult_inverse[i];
                }
                offset_for_pixel %= sample_stride;
            }
            px = cur_x; py = cur_y;
        }
        return offset_for_pixel + s * sample_stride;
    }
    Float sample_dimension(uint64_t index, uint32_t dim) const {  // halton.rs:146-160
        if (at_center && (dim == 0 || dim == 1)) return 0.5f;
        if (dim == 0) return radical_inverse(0, index >> base_exponents[0]);
        if (dim == 1) return radical_inverse(1, index / base_scales[1]);
        // permutation_for_dimension asserts dim <= PRIME_TABLE_SIZE (halton.rs:106-110)
        return scrambled_radical_inverse((int)dim, index, &ld_tables().perms[ld_tables().prime_sums[dim]]);
    }
    void start_pixel(int x, int y) { cur_x = x; cur_y = y; sample_num = 0; dimension = 0; interval_index = index_for_sample(0); }
    bool start_next_sample() { dimension = 0; interval_index = index_for_sample(sample_num + 1); sample_num++; return sample_num < spp; }
    void set_sample_number(uint32_t s) { dimension = 0; interval_index = index_for_sample(s); sample_num = s; }
    Float get_1d() { Float p = sample_dimension(interval_index, dimension); dimension += 1; return p; }  // array dims: start=end=5, never fire
    V2 get_2d() {
        if (dimension + 1 >= 5 && dimension < 5) dimension = 5;  // halton.rs:238-240 with no arrays
        V2 p(sample_dimension(interval_index, dimension), sample_dimension(interval_index, dimension + 1));
        dimension += 2;
        return p;
    }
};

// RandomSampler (samplers/src/random.rs) — ORACLE ONLY (its draws depend on the order in which a tile's pixels and samples are visited): `clone_sampler(seed)` gives every
// tile an RNG on sequence `seed` = the tile's index (sampler_integrator.rs:322); every value is the next uniform_float of that stream, camera samples and light samples alike
struct RandomSampler {
    RNG rng; uint32_t spp, sample_num = 0;
    RandomSampler(const SamplerConfig& c, uint64_t seed) : rng(seed), spp(c.spp) {}
    void start_pixel(int, int) { sample_num = 0; }   // no sample arrays are requested on this path: start_pixel draws nothing
    bool start_next_sample() { sample_num++; return sample_num < spp; }
    void set_sample_number(uint32_t s) { sample_num = s; }
    Float get_1d() { return rng.uniform_float(); }
    V2 get_2d() { const Float x = rng.uniform_float(); const Float y = rng.uniform_float(); return V2(x, y); }
};

// SobolSampler (samplers/src/sobol.rs:35-93, core/src/low_discrepency.rs:1770-1848)
struct SobolSampler {
    SobolTables tb; uint32_t spp; int bounds[4]; int resolution, log2_res;
    int cur_x = 0, cur_y = 0; uint64_t interval_index = 0; uint32_t dimension = 0, sample_num = 0;
    static bool is_pow2(uint32_t v) { return v && !(v & (v - 1)); }
    static uint32_t round_up_pow2(uint32_t v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; }
    static int log2_int(uint32_t v) { int r = 0; while (v >>= 1) r++; return r; }
    SobolSampler(const SamplerConfig& c) : tb(c.sobol) {
        spp = is_pow2(c.spp) ? c.spp : round_up_pow2(c.spp);  // sobol.rs:38-47 (round_up_pow2)
        for (int i = 0; i < 4; i++) bounds[i] = c.bounds[i];
        int dx = bounds[2] - bounds[0], dy = bounds[3] - bounds[1];
        resolution = (int)round_up_pow2((uint32_t)pmax(dx, dy));
        log2_res = log2_int((uint32_t)resolution);
    }
    uint64_t interval_to_index(uint32_t m, uint64_t frame, int px, int py) const {  // low_discrepency.rs:1770-1824
        if (m == 0) return 0;
        uint32_t m2 = m << 1;
        uint64_t index = frame << m2;
        uint64_t delta = 0;
        for (int c = 0; frame != 0; frame >>= 1, c++)
            if (frame & 1) delta ^= tb.vdc[(size_t)(m - 1) * 52 + c];
        uint64_t b = (((uint64_t)(uint32_t)px << m) | (uint64_t)(uint32_t)py) ^ delta;
        for (int c = 0; b != 0; b >>= 1, c++)
            if (b & 1) index ^= tb.vdc_inv[(size_t)(m - 1) * 52 + c];
        return index;
    }
    Float sobol_sample(uint64_t a, uint32_t dim) const {  // sobol_sample_f32, scramble = 0 (:1826-1848)
        uint32_t v = 0;
        for (size_t i = (size_t)dim * 52; a != 0; a >>= 1, i++) if (a & 1) v ^= tb.m32[i];
        return pmin((Float)v * 0x1.0p-32f, ONE_MINUS_EPSILON);
    }
    uint64_t index_for_sample(uint64_t s) const { return interval_to_index((uint32_t)log2_res, s, cur_x - bounds[0], cur_y - bounds[1]); }
    Float sample_dimension(uint64_t index, uint32_t dim) const {  // sobol.rs:77-93
        Float s = sobol_sample(index, dim);
        if (dim == 0 || dim == 1) {
            s = s * (Float)resolution + (Float)bounds[dim];
            s = pclamp(s - (Float)(dim == 0 ? cur_x : cur_y), 0.0f, ONE_MINUS_EPSILON);
        }
        return s;
    }
    void start_pixel(int x, int y) { cur_x = x; cur_y = y; sample_num = 0; dimension = 0; interval_index = index_for_sample(0); }
    bool start_next_sample() { dimension = 0; interval_index = index_for_sample(sample_num + 1); sample_num++; return sample_num < spp; }
    void set_sample_number(uint32_t s) { dimension = 0; interval_index = index_for_sample(s); sample_num = s; }
    Float get_1d() { Float p = sample_dimension(interval_index, dimension); dimension += 1; return p; }
    V2 get_2d() {
        if (dimension + 1 >= 5 && dimension < 5) dimension = 5;
        V2 p(sample_dimension(interval_index, dimension), sample_dimension(interval_index, dimension + 1));
        dimension += 2;
        return p;
    }
};

// =============================== sampling routines (core/src/sampling/common.rs) ==================================
inline V2 concentric_sample_disk(V2 u) {  // :138-155
    V2 uo(2.0f * u.x - 1.0f, 2.0f * u.y - 1.0f);
    if (uo.x == 0.0f && uo.y == 0.0f) return V2(0, 0);
    Float r, theta;
    if (pabs(uo.x) > pabs(uo.y)) { r = uo.x; theta = PI_OVER_FOUR * (uo.y / uo.x); }
    else { r = uo.y; theta = PI_OVER_TWO - PI_OVER_FOUR * (uo.x / uo.y); }
    return V2(r * o_cos(theta), r * o_sin(theta));
}
inline V3 cosine_sample_hemisphere(V2 u) {  // :207-211
    V2 d = concentric_sample_disk(u);
    Float z = std::sqrt(pmax(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return V3(d.x, d.y, z);
}
inline V2 uniform_sample_triangle(V2 u) { Float su0 = std::sqrt(u.x); return V2(1.0f - su0, u.y * su0); }  // :198-201
inline Float power_heuristic(int nf, Float fp, int ng, Float gp) {  // :239-243
    Float f = (Float)nf * fp, g = (Float)ng * gp;
    return (f * f) / (f * f + g * g);
}

// Distribution1D for small fixed sizes (core/src/sampling/distribution_1d.rs:22-95)
struct Dist1D {
    std::vector<Float> func, cdf; Float func_int = 0;
    void init(const std::vector<Float>& f) {
        func = f; size_t n = f.size(); cdf.assign(n + 1, 0.0f);
        for (size_t i = 1; i < n + 1; i++) cdf[i] = cdf[i - 1] + f[i - 1] / (Float)n;
        func_int = cdf[n];
        if (func_int == 0.0f) for (size_t i = 1; i < n + 1; i++) cdf[i] = (Float)i / (Float)n;
        else for (size_t i = 1; i < n + 1; i++) cdf[i] /= func_int;
    }
    size_t count() const { return func.size(); }
    Float sample_continuous(Float u, Float& pdf, size_t& off) const {
        size_t offset = find_interval(cdf.size(), [&](size_t i) { return cdf[i] <= u; });
        Float du = u - cdf[offset];
        if (cdf[offset + 1] - cdf[offset] > 0.0f) du /= cdf[offset + 1] - cdf[offset];
        pdf = func_int > 0.0f ? func[offset] / func_int : 0.0f;
        off = offset;
        return ((Float)offset + du) / (Float)count();
    }
    size_t sample_discrete(Float u, Float& pdf) const {
        size_t offset = find_interval(cdf.size(), [&](size_t i) { return cdf[i] <= u; });
        pdf = func_int > 0.0f ? func[offset] / (func_int * (Float)count()) : 0.0f;
        return offset;
    }
};

// =============================== lights ===========================================================================
// constant-L InfiniteAreaLight: 1x1 MIPMap::triangle (core/src/mipmap/mod.rs:293-311, texel :580-608)
inline Spec infinite_lookup(const Spec& tx, V2 st) {
    Float s = st.x * 1.0f - 0.5f, t = st.y * 1.0f - 0.5f;
    Float s0 = std::floor(s), t0 = std::floor(t);  // `as isize` then back `as Float`: exact for these magnitudes
    Float ds = s - s0, dt = t - t0;
    return tx * (1.0f - ds) * (1.0f - dt) + tx * (1.0f - ds) * dt + tx * ds * (1.0f - dt) + tx * ds * dt;
}
inline void infinite_light_setup(Light& l) {  // lights/src/infinite.rs:63-107, 326-369
    Float img[2][2];
    for (int v = 0; v < 2; v++) {
        Float vp = ((Float)v + 0.5f) / 2.0f;
        Float sin_theta = std::sin(PI * ((Float)v + 0.5f) / 2.0f);
        for (int u = 0; u < 2; u++) {
            Float up = ((Float)u + 0.5f) / 2.0f;
            img[v][u] = infinite_lookup(l.L, V2(up, vp)).y() * sin_theta;
        }
    }
    Dist1D d; std::vector<Float> marg;
    for (int v = 0; v < 2; v++) {
        d.init({img[v][0], img[v][1]});
        for (int i = 0; i < 2; i++) l.cond_func[v][i] = d.func[i];
        for (int i = 0; i < 3; i++) l.cond_cdf[v][i] = d.cdf[i];
        l.cond_int[v] = d.func_int; marg.push_back(d.func_int);
    }
    d.init(marg);
    for (int i = 0; i < 2; i++) l.marg_func[i] = d.func[i];
    for (int i = 0; i < 3; i++) l.marg_cdf[i] = d.cdf[i];
    l.marg_int = d.func_int;
}
// Distribution1D::sample_continuous over arrays of any length (distribution_1d.rs:55-79)
inline Float distn_sample_continuous(const Float* func, const Float* cdf, size_t n, Float func_int, Float u, Float& pdf, size_t& off) {
    size_t offset = find_interval(n + 1, [&](size_t i) { return cdf[i] <= u; });
    Float du = u - cdf[offset];
    if (cdf[offset + 1] - cdf[offset] > 0.0f) du /= cdf[offset + 1] - cdf[offset];
    pdf = func_int > 0.0f ? func[offset] / func_int : 0.0f;
    off = offset;
    return ((Float)offset + du) / (Float)n;
}
inline Float dist2_sample_continuous(const Float func[2], const Float cdf[3], Float func_int, Float u, Float& pdf, size_t& off) {
    size_t offset = find_interval(3, [&](size_t i) { return cdf[i] <= u; });
    Float du = u - cdf[offset];
    if (cdf[offset + 1] - cdf[offset] > 0.0f) du /= cdf[offset + 1] - cdf[offset];
    pdf = func_int > 0.0f ? func[offset] / func_int : 0.0f;
    off = offset;
    return ((Float)offset + du) / 2.0f;
}

struct SurfaceHit {  // the fields of Hit/SurfaceInteraction/Shading the path integrator consumes
    V3 p, p_error, wo, n;   // Hit (core/src/interaction/mod.rs:107-125)
    V3 ns, dpdu_s;          // shading.n, shading.dpdu
    Float time;
    uint32_t prim;
    V2 uv; V3 dpdu, dpdv;   // SurfaceInteraction.uv, der.dpdu / der.dpdv (geometric)
    Float dudx = 0, dvdx = 0, dudy = 0, dvdy = 0;  // der, filled by compute_differentials
    V3 dpdx, dpdy;
    V3 dpdv_s, dndu_s, dndv_s;  // shading.dpdv, shading.dndu, shading.dndv (consumed by Material::bump only)
};

// Ray::offset_origin (core/src/geometry/ray.rs:107-127)
inline V3 offset_origin(V3 p, V3 p_error, V3 n, V3 w) {
    Float d = dot(vabs(n), p_error);
    V3 offset = d * n;
    if (dot(w, n) < 0.0f) offset = -offset;
    V3 po = p + offset;
    for (int a = 0; a < 3; a++) {
        if (offset[a] > 0.0f) po[a] = next_float_up(po[a]);
        else if (offset[a] < 0.0f) po[a] = next_float_down(po[a]);
    }
    return po;
}
inline Ray spawn_ray(const V3& p, const V3& p_error, const V3& n, Float time, V3 d) {  // interaction/mod.rs:189-192
    return Ray(offset_origin(p, p_error, n, d), d, INF, time);
}
// Hit::spawn_ray_to_hit (interaction/mod.rs:212-223)
inline Ray spawn_ray_to_hit(const V3& p, const V3& p_error, const V3& n, Float time, const V3& hp, const V3& hperr, const V3& hn) {
    V3 origin = offset_origin(p, p_error, n, hp - p);
    V3 target = offset_origin(hp, hperr, hn, origin - hp);
    return Ray(origin, target - origin, 1.0f - SHADOW_EPSILON, time);
}

struct LiSample { V3 wi; Float pdf; Spec value; V3 vp, vperr, vn; bool valid; };  // Li + VisibilityTester.p1

struct RenderStats {  // the reference keeps these per thread and merges at exit (core/src/stats/macros.rs:176-222)
    uint64_t camera_rays = 0, regular_rays = 0, shadow_rays = 0, zero_paths = 0, total_paths = 0;
    uint64_t nv_regular = 0, nt_regular = 0, nv_shadow = 0, nt_shadow = 0;
    void add(const RenderStats& o) {
        camera_rays += o.camera_rays; regular_rays += o.regular_rays; shadow_rays += o.shadow_rays; zero_paths += o.zero_paths; total_paths += o.total_paths;
        nv_regular += o.nv_regular; nt_regular += o.nt_regular; nv_shadow += o.nv_shadow; nt_shadow += o.nt_shadow;
    }
};
inline RenderStats& tls_stats() { static thread_local RenderStats s; return s; }

struct RayRecorder {  // optional capture of every ray handed to Scene::intersect / intersect_p
    std::mutex mu; std::vector<Ray> regular, shadow; size_t cap = 0;
    void add(std::vector<Ray>& v, const Ray& r) { std::lock_guard<std::mutex> g(mu); if (v.size() < cap) v.push_back(r); }
};

struct Camera {  // cameras/src/perspective_camera.rs, orthographic_camera.rs
    int kind = 0;  // 0 perspective, 1 orthographic, 2 environment
    Float full_res[2] = {1, 1};  // EnvironmentCamera reads film.full_resolution (environment_camera.rs:63-64)
    Transform raster_to_camera, camera_to_world;
    Float lens_radius = 0, focal_distance = 1e6f, shutter_open = 0, shutter_close = 1;
    V3 dx_camera, dy_camera;  // perspective_camera.rs:70-74, set with the camera
};
struct FilmCfg {
    int xres = 0, yres = 0; int crop[4] = {0, 0, 0, 0};
    Float radius[2] = {0.5f, 0.5f}; Float table[256]; Float scale = 1.0f, max_lum = INF;
};

struct Renderer {
    Scene* sc = nullptr;
    Camera cam; FilmCfg film; SamplerConfig scfg;
    int max_depth = 5; Float rr_threshold = 1.0f; int light_strategy = 0;
    int pixel_bounds[4] = {0, 0, 0, 0};
    Dist1D light_distrib;
    // SpatialLightDistribution (core/src/light_distrib/spatial.rs): the reference keeps voxel distributions in a lock-free
    // hash table; a voxel's distribution is a pure function of the voxel, so the table is only a cache and is restated
    // here as a dense array filled on first use.  (With several threads the reference's lookup can return None while
    // another thread is still computing the entry, :213-216, and the caller then samples uniformly — a data race; this
    // restatement is the race-free, --nthreads 1 behaviour.)
    bool spatial = false; int n_voxels[3] = {1, 1, 1}; Float wb_lo[3], wb_hi[3];
    std::vector<std::unique_ptr<Dist1D>> voxel_dist; std::vector<std::once_flag> voxel_once;
    uint64_t spatial_created = 0; std::mutex spatial_mu;
    RenderStats total_stats; std::mutex stats_mu; RayRecorder* rec = nullptr; bool count_traversal = false;

    // ---- Scene::intersect / intersect_p with the reference's counters (core/src/scene.rs:88-99)
    bool scene_intersect(Ray& r, uint32_t& prim, TriHit& h) {
        RenderStats& stats = tls_stats();
        stats.regular_rays++;
        if (rec) rec->add(rec->regular, r);
        if (count_traversal) { TraversalStats ts; bool b = sc->intersect(r, prim, h, &ts); stats.nv_regular += ts.nodes_visited; stats.nt_regular += ts.tri_tests; return b; }
        return sc->intersect(r, prim, h);
    }
    bool scene_intersect_p(const Ray& r) {
        RenderStats& stats = tls_stats();
        stats.shadow_rays++;
        if (rec) rec->add(rec->shadow, r);
        if (count_traversal) { TraversalStats ts; bool b = sc->intersect_p(r, &ts); stats.nv_shadow += ts.nodes_visited; stats.nt_shadow += ts.tri_tests; return b; }
        return sc->intersect_p(r);
    }

    // ---- tail of Triangle::intersect (triangle.rs:547-724) + Hit::new (interaction/mod.rs:137-156)
    SurfaceHit make_surface_hit(const Ray& r_world, uint32_t prim, const TriHit& h) const {
        if (h.inst == 0) return make_surface_hit_local(r_world, prim, h);
        // TransformedPrimitive::intersect (transformed_primitive.rs:51-73): the triangle was met by the instance-space ray, its
        // interaction is then carried to world space by transform_surface_interaction (transform.rs:566-590)
        const Instance& in = sc->instances[h.inst - 1];
        Ray ray = transform_ray(in.i2w.inv(), r_world);
        SurfaceHit si = make_surface_hit_local(ray, prim, h);
        if (in.i2w.is_identity()) return si;
        const Transform& t = in.i2w;
        V3 pe; si.p = t.point_with_abs_error(si.p, si.p_error, pe); si.p_error = pe;
        si.wo = normalize(t.vector(si.wo));
        si.n = normalize(t.normal(si.n));
        si.ns = normalize(t.normal(si.ns));
        si.ns = face_forward(si.ns, si.n);
        si.dpdu_s = t.vector(si.dpdu_s);
        si.dpdu = t.vector(si.dpdu); si.dpdv = t.vector(si.dpdv);  // transform.rs:566-590
        si.dpdv_s = t.vector(si.dpdv_s); si.dndu_s = t.normal(si.dndu_s); si.dndv_s = t.normal(si.dndv_s);
        return si;
    }
    // ---- tail of Sphere::intersect (sphere.rs:173-241): parametric form, normal derivatives from the fundamental forms, SurfaceInteraction::new
    // (surface_interaction.rs:69-98), then object_to_world.transform_surface_interaction (transform.rs:566-590)
    SurfaceHit make_sphere_hit(const Ray& r_world, uint32_t prim, const TriHit& h, const Sphere& sp) const {
        SurfaceHit si; si.prim = prim; si.time = r_world.time;
        const V3 p = h.sp; const Float phi = h.sphi;
        const Float u = phi / sp.phi_max;
        const Float theta = o_acos(pclamp(p.z / sp.radius, -1.0f, 1.0f));
        const Float v = (theta - sp.theta_min) / (sp.theta_max - sp.theta_min);
        const Float z_radius = std::sqrt(p.x * p.x + p.y * p.y);
        const Float inv_z_radius = 1.0f / z_radius;
        const Float cos_phi = p.x * inv_z_radius, sin_phi = p.y * inv_z_radius;
        const V3 dpdu(-sp.phi_max * p.y, sp.phi_max * p.x, 0.0f);
        const V3 dpdv = (sp.theta_max - sp.theta_min) * V3(p.z * cos_phi, p.z * sin_phi, -sp.radius * o_sin(theta));
        const V3 d2p_duu = (-sp.phi_max * sp.phi_max) * V3(p.x, p.y, 0.0f);
        const V3 d2p_duv = ((sp.theta_max - sp.theta_min) * p.z * sp.phi_max) * V3(-sin_phi, cos_phi, 0.0f);
        const V3 d2p_dvv = (-(sp.theta_max - sp.theta_min) * (sp.theta_max - sp.theta_min)) * V3(p.x, p.y, p.z);
        const V3 nn = normalize(cross(dpdu, dpdv));
        const Float e1 = dot(dpdu, dpdu), f1 = dot(dpdu, dpdv), g1 = dot(dpdv, dpdv);
        const Float e2 = dot(nn, d2p_duu), f2 = dot(nn, d2p_duv), g2 = dot(nn, d2p_dvv);
        const Float inv_egf_1 = 1.0f / (e1 * g1 - f1 * f1);
        const V3 dndu = ((f2 * f1 - e2 * g1) * inv_egf_1) * dpdu + ((e2 * f1 - f2 * e1) * inv_egf_1) * dpdv;
        const V3 dndv = ((g2 * f1 - f2 * g1) * inv_egf_1) * dpdu + ((f2 * f1 - g2 * e1) * inv_egf_1) * dpdv;
        const V3 p_error = gamma_n(5) * vabs(p);
        quadric_to_world(si, sp.o2w, sp.w2o, sp.reverse_orientation ^ sp.swaps_handedness, r_world, p, p_error, V2(u, v), dpdu, dpdv, dndu, dndv);
        return si;
    }
    // SurfaceInteraction::new (surface_interaction.rs:69-98) with wo = -ray.d of the OBJECT-space ray, then object_to_world.transform_surface_interaction (transform.rs:566-590)
    static void quadric_to_world(SurfaceHit& si, const Transform& o2w, const Transform& w2o, bool flip_n, const Ray& r_world, V3 p, V3 p_error, V2 uv, V3 dpdu, V3 dpdv, V3 dndu, V3 dndv) {
        V3 n = normalize(cross(dpdu, dpdv));
        if (flip_n) n = n * -1.0f;
        V3 wo = -w2o.vector(r_world.d); const Float l2 = length_squared(wo);
        wo = (l2 == 0.0f) ? wo : wo / std::sqrt(l2);  // Hit::new (interaction/mod.rs:137-156)
        V3 pe; si.p = o2w.point_with_abs_error(p, p_error, pe); si.p_error = pe;
        si.wo = normalize(o2w.vector(wo));
        si.n = normalize(o2w.normal(n));
        si.dpdu = o2w.vector(dpdu); si.dpdv = o2w.vector(dpdv);
        si.ns = face_forward(normalize(o2w.normal(n)), si.n);
        si.dpdu_s = o2w.vector(dpdu); si.dpdv_s = o2w.vector(dpdv);
        si.dndu_s = o2w.normal(dndu); si.dndv_s = o2w.normal(dndv);
        si.uv = uv;
    }
    // ---- tails of Cylinder / Cone / Paraboloid / Disk ::intersect (cylinder.rs:146-196, cone.rs:132-190, paraboloid.rs:132-200, disk.rs:106-140)
    SurfaceHit make_quadric_hit(const Ray& r_world, uint32_t prim, const TriHit& h, const Quadric& q) const {
        SurfaceHit si; si.prim = prim; si.time = r_world.time;
        V3 p = h.sp; const Float phi = h.sphi;
        const Float u = phi / q.phi_max;
        Float v = 0.0f;
        V3 dpdu(-q.phi_max * p.y, q.phi_max * p.x, 0.0f), dpdv, dndu(0, 0, 0), dndv(0, 0, 0);
        if (q.kind == Q_DISK) {
            const Float r_hit = std::sqrt(h.sv);  // sv = dist2
            v = (q.radius - r_hit) / (q.radius - q.inner_radius);
            dpdv = V3(p.x, p.y, 0.0f) * (q.inner_radius - q.radius) / r_hit;
            p.z = q.height;  // refine (disk.rs:118)
        } else {
            V3 d2p_duu = (-q.phi_max * q.phi_max) * V3(p.x, p.y, 0.0f), d2p_duv(0, 0, 0), d2p_dvv(0, 0, 0);
            if (q.kind == Q_CYLINDER) {
                v = (p.z - q.z_min) / (q.z_max - q.z_min);
                dpdv = V3(0.0f, 0.0f, q.z_max - q.z_min);
            } else if (q.kind == Q_CONE) {
                v = p.z / q.height;
                dpdv = V3(-p.x / (1.0f - v), -p.y / (1.0f - v), q.height);
                d2p_duv = (q.phi_max / (1.0f - v)) * V3(p.y, -p.x, 0.0f);
            } else {
                v = (p.z - q.z_min) / (q.z_max - q.z_min);
                dpdv = (q.z_max - q.z_min) * V3(p.x / (2.0f * p.z), p.y / (2.0f * p.z), 1.0f);
                d2p_duv = ((q.z_max - q.z_min) * q.phi_max) * V3(-p.y / (2.0f * p.z), p.x / (2.0f * p.z), 0.0f);
                d2p_dvv = (-(q.z_max - q.z_min) * (q.z_max - q.z_min)) * V3(p.x / (4.0f * p.z * p.z), p.y / (4.0f * p.z * p.z), 0.0f);
            }
            const V3 nn = normalize(cross(dpdu, dpdv));
            const Float e1 = dot(dpdu, dpdu), f1 = dot(dpdu, dpdv), g1 = dot(dpdv, dpdv);
            const Float e2 = dot(nn, d2p_duu), f2 = dot(nn, d2p_duv), g2 = dot(nn, d2p_dvv);
            const Float inv_egf_1 = 1.0f / (e1 * g1 - f1 * f1);
            dndu = ((f2 * f1 - e2 * g1) * inv_egf_1) * dpdu + ((e2 * f1 - f2 * e1) * inv_egf_1) * dpdv;
            dndv = ((g2 * f1 - f2 * g1) * inv_egf_1) * dpdu + ((f2 * f1 - g2 * e1) * inv_egf_1) * dpdv;
        }
        quadric_to_world(si, q.o2w, q.w2o, q.reverse_orientation ^ q.swaps_handedness, r_world, p, h.sperr, V2(u, v), dpdu, dpdv, dndu, dndv);
        return si;
    }
    // ---- tail of Hyperboloid::intersect (hyperboloid.rs:192-262)
    SurfaceHit make_hyperboloid_hit(const Ray& r_world, uint32_t prim, const TriHit& h, const Hyperboloid& hy) const {
        SurfaceHit si; si.prim = prim; si.time = r_world.time;
        const V3 p = h.sp; const Float phi = h.sphi;
        const Float u = phi / hy.phi_max;
        const Float cos_phi = o_cos(phi), sin_phi = o_sin(phi);
        const V3 dpdu(-hy.phi_max * p.y, hy.phi_max * p.x, 0.0f);
        const V3 dpdv((hy.p2.x - hy.p1.x) * cos_phi - (hy.p2.y - hy.p1.y) * sin_phi, (hy.p2.x - hy.p1.x) * sin_phi + (hy.p2.y - hy.p1.y) * cos_phi, hy.p2.z - hy.p1.z);
        const V3 d2p_duu = (-hy.phi_max * hy.phi_max) * V3(p.x, p.y, 0.0f);
        const V3 d2p_duv = hy.phi_max * V3(-dpdv.y, dpdv.x, 0.0f);
        const V3 d2p_dvv(0.0f, 0.0f, 0.0f);
        const V3 nn = normalize(cross(dpdu, dpdv));
        const Float e1 = dot(dpdu, dpdu), f1 = dot(dpdu, dpdv), g1 = dot(dpdv, dpdv);
        const Float e2 = dot(nn, d2p_duu), f2 = dot(nn, d2p_duv), g2 = dot(nn, d2p_dvv);
        const Float inv_egf_1 = 1.0f / (e1 * g1 - f1 * f1);
        const V3 dndu = ((f2 * f1 - e2 * g1) * inv_egf_1) * dpdu + ((e2 * f1 - f2 * e1) * inv_egf_1) * dpdv;
        const V3 dndv = ((g2 * f1 - f2 * g1) * inv_egf_1) * dpdu + ((f2 * f1 - g2 * e1) * inv_egf_1) * dpdv;
        quadric_to_world(si, hy.o2w, hy.w2o, hy.reverse_orientation ^ hy.swaps_handedness, r_world, p, h.sperr, V2(u, h.sv), dpdu, dpdv, dndu, dndv);
        return si;
    }
    SurfaceHit make_surface_hit_local(const Ray& r, uint32_t prim, const TriHit& h) const {
        const Scene& s = *sc; const Mesh& m = s.mesh_of(prim);
        if (m.sphere >= 0) return make_sphere_hit(r, prim, h, s.spheres[(size_t)m.sphere]);
        if (m.hyper >= 0) return make_hyperboloid_hit(r, prim, h, s.hyperboloids[(size_t)m.hyper]);
        if (m.quadric >= 0) return make_quadric_hit(r, prim, h, s.quadrics[(size_t)m.quadric]);
        uint32_t i0 = s.idx[3 * prim], i1 = s.idx[3 * prim + 1], i2 = s.idx[3 * prim + 2];
        V3 p0 = s.P[i0], p1 = s.P[i1], p2 = s.P[i2];
        Float b0 = h.b0, b1 = h.b1, b2 = h.b2;
        SurfaceHit si; si.prim = prim; si.time = r.time;
        V3 dpdu, dpdv; s.tri_dpdu_dpdv(prim, dpdu, dpdv);
        V3 dp02 = p0 - p2, dp12 = p1 - p2;
        Float xs = std::fabs(b0 * p0.x) + std::fabs(b1 * p1.x) + std::fabs(b2 * p2.x);
        Float ys = std::fabs(b0 * p0.y) + std::fabs(b1 * p1.y) + std::fabs(b2 * p2.y);
        Float zs = std::fabs(b0 * p0.z) + std::fabs(b1 * p1.z) + std::fabs(b2 * p2.z);
        si.p_error = gamma_n(7) * V3(xs, ys, zs);
        si.p = b0 * p0 + b1 * p1 + b2 * p2;
        V3 wo = -r.d; Float l2 = length_squared(wo);
        si.wo = (l2 == 0.0f) ? wo : wo / std::sqrt(l2);
        si.n = normalize(cross(dp02, dp12));
        if (m.reverse_orientation ^ m.swaps_handedness) si.n = -si.n;
        si.ns = si.n; si.dpdu_s = dpdu;
        si.dpdu = dpdu; si.dpdv = dpdv;
        si.dpdv_s = dpdv; si.dndu_s = V3(0, 0, 0); si.dndv_s = V3(0, 0, 0);  // SurfaceInteraction::new: shading = geometric, triangles pass dndu = dndv = 0
        { V2 uv[3]; s.tri_uvs(prim, uv); si.uv = V2((b0 * uv[0].x + b1 * uv[1].x) + b2 * uv[2].x, (b0 * uv[0].y + b1 * uv[1].y) + b2 * uv[2].y); }  // triangle.rs:584
        if (m.has_n || m.has_s) {  // :631-721
            V3 ns;
            if (m.has_n) {
                V3 ns2 = b0 * s.N[i0] + b1 * s.N[i1] + b2 * s.N[i2];
                ns = length_squared(ns2) > 0.0f ? normalize(ns2) : si.n;
            } else ns = si.n;
            V3 ss;
            if (m.has_s) {
                V3 ss2 = b0 * s.S[i0] + b1 * s.S[i1] + b2 * s.S[i2];
                ss = length_squared(ss2) > 0.0f ? normalize(ss2) : normalize(dpdu);
            } else ss = normalize(dpdu);
            V3 ts = cross(ss, ns);
            if (length_squared(ts) > 0.0f) { ts = normalize(ts); ss = cross(ts, ns); }
            else coordinate_system(ns, ss, ts);
            V3 dndu(0, 0, 0), dndv(0, 0, 0);  // triangle.rs:681-715
            if (m.has_n) {
                V2 uv[3]; s.tri_uvs(prim, uv);
                V2 duv02(uv[0].x - uv[2].x, uv[0].y - uv[2].y), duv12(uv[1].x - uv[2].x, uv[1].y - uv[2].y);
                V3 n0 = s.N[i0], n1 = s.N[i1], n2 = s.N[i2];
                V3 dn1 = n0 - n2, dn2 = n1 - n2;
                Float determinant = duv02.x * duv12.y - duv02.y * duv12.x;
                if (std::fabs(determinant) < 1e-8f) {
                    V3 dn = cross(n2 - n0, n1 - n0);
                    if (length_squared(dn) != 0.0f) coordinate_system(dn, dndu, dndv);
                } else {
                    Float invdet = 1.0f / determinant;
                    dndu = (duv12.y * dn1 - duv02.y * dn2) * invdet;
                    dndv = (-duv12.x * dn1 + duv02.x * dn2) * invdet;
                }
            }
            if (m.reverse_orientation) ts = -ts;
            si.dpdv_s = ts; si.dndu_s = dndu; si.dndv_s = dndv;
            // set_shading_geometry(ss, ts, .., true) (surface_interaction.rs:152-173)
            si.ns = normalize(cross(ss, ts));
            si.n = face_forward(si.n, si.ns);
            si.dpdu_s = ss;
        }
        return si;
    }

    // ---- lights ------------------------------------------------------------------------------------------------
    Spec area_L(const Light& l, V3 n, V3 w) const { return (l.two_sided || dot(n, w) > 0.0f) ? l.L : Spec(0.0f); }  // diffuse.rs:220-226
    // SurfaceInteraction::le (surface_interaction.rs:283-289): the emission of the primitive's area light, whether or not that light is one of the scene's (Scene::emission_only)
    Spec prim_le(const Mesh& m, uint32_t prim, V3 n, V3 w) const {
        if (m.first_light >= 0) return area_L(sc->lights[m.first_light + (prim - m.tri_base)], n, w);
        if (m.first_light <= -2) return area_L(sc->emission_only[(size_t)(-2 - m.first_light) + (prim - m.tri_base)], n, w);
        return Spec(0.0f);
    }
    Spec light_le(const Light& l, const Ray& ray) const {  // Light::le: infinite.rs:188-195; others Spectrum::ZERO
        if (l.type != L_INFINITE) return Spec(0.0f);
        V3 w = normalize(l.l2w.inv().vector(ray.d));
        V2 st(spherical_phi(w) * INV_TWO_PI, spherical_theta(w) * INV_PI);
        if (l.map_mip >= 0) return sc->mipmaps[(size_t)l.map_mip].triangle(0, st);   // lookup_triangle(st, 0.0): level < 0 -> triangle(0, st)
        return infinite_lookup(l.L, st);
    }
    LiSample light_sample_li(const Light& l, const SurfaceHit& hit, V2 u) const {
        LiSample r; r.valid = false; r.pdf = 0; r.vn = V3(); r.vperr = V3();
        const Scene& s = *sc;
        switch (l.type) {
        case L_INFINITE: {  // infinite.rs:133-173
            Float pdf1, pdf0, d1, d0; size_t v, dummy;
            if (l.map_mip >= 0) {  // Distribution2D::sample_continuous (distribution_2d.rs:31-49)
                d1 = distn_sample_continuous(l.d_marg_func.data(), l.d_marg_cdf.data(), (size_t)l.dh, l.d_marg_int, u.y, pdf1, v);
                d0 = distn_sample_continuous(l.d_cond_func.data() + v * (size_t)l.dw, l.d_cond_cdf.data() + v * (size_t)(l.dw + 1), (size_t)l.dw, l.d_cond_int[v], u.x, pdf0, dummy);
            } else {
                d1 = dist2_sample_continuous(l.marg_func, l.marg_cdf, l.marg_int, u.y, pdf1, v);
                d0 = dist2_sample_continuous(l.cond_func[v], l.cond_cdf[v], l.cond_int[v], u.x, pdf0, dummy);
            }
            Float map_pdf = pdf0 * pdf1;
            if (map_pdf == 0.0f) return r;
            Float theta = d1 * PI, phi = d0 * TWO_PI;
            Float cos_theta = o_cos(theta), sin_theta = o_sin(theta), sin_phi = o_sin(phi), cos_phi = o_cos(phi);
            r.wi = l.l2w.vector(V3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta));
            r.pdf = map_pdf / (TWO_PI * PI * sin_theta);
            if (sin_theta == 0.0f) r.pdf = 0.0f;
            r.vp = hit.p + r.wi * (2.0f * s.world_radius);
            r.value = l.map_mip >= 0 ? sc->mipmaps[(size_t)l.map_mip].triangle(0, V2(d0, d1)) : infinite_lookup(l.L, V2(d0, d1));
            r.valid = true; return r;
        }
        case L_DISTANT:  // distant.rs:87-96
            r.wi = l.w_light; r.pdf = 1.0f; r.vp = hit.p + l.w_light * (2.0f * s.world_radius); r.value = l.L; r.valid = true; return r;
        case L_SPOT: {   // spot.rs:75-84 with falloff (:52-66)
            r.wi = normalize(l.p_light - hit.p); r.pdf = 1.0f; r.vp = l.p_light;
            V3 wl = normalize(l.l2w.inv().vector(-r.wi));
            Float cos_theta = wl.z, fall;
            if (cos_theta < l.cos_total_width) fall = 0.0f;
            else if (cos_theta >= l.cos_falloff_start) fall = 1.0f;
            else { Float delta = (cos_theta - l.cos_total_width) / (l.cos_falloff_start - l.cos_total_width); fall = (delta * delta) * (delta * delta); }
            r.value = l.L * fall / distance_squared(l.p_light, hit.p); r.valid = true; return r;
        }
        case L_PROJECTION: {   // projection.rs:180-191 + projection() :145-168
            r.wi = normalize(l.p_light - hit.p); r.pdf = 1.0f; r.vp = l.p_light;
            const Float near_z = 1e-3f;
            Spec pr(0.0f);
            V3 wl = l.l2w.inv().vector(-r.wi);
            if (!(wl.z < near_z)) {
                V3 pp = l.light_projection.point(wl);
                if (pp.x >= l.screen[0] && pp.x <= l.screen[1] && pp.y >= l.screen[2] && pp.y <= l.screen[3]) {
                    if (l.map_mip < 0) pr = Spec(1.0f);
                    else {
                        Float ox = pp.x - l.screen[0], oy = pp.y - l.screen[2];      // Bounds2::offset (bounds2.rs:161-173)
                        if (l.screen[1] > l.screen[0]) ox /= l.screen[1] - l.screen[0];
                        if (l.screen[3] > l.screen[2]) oy /= l.screen[3] - l.screen[2];
                        pr = sc->mipmaps[(size_t)l.map_mip].triangle(0, V2(ox, oy));
                    }
                }
            }
            r.value = l.L * pr / distance_squared(l.p_light, hit.p); r.valid = true; return r;
        }
        case L_GONIO: {        // goniometric.rs:102-113 + scale() :82-96
            r.wi = normalize(l.p_light - hit.p); r.pdf = 1.0f; r.vp = l.p_light;
            V3 wp = normalize(l.l2w.inv().vector(-r.wi));
            { Float t = wp.y; wp.y = wp.z; wp.z = t; }
            Float theta = spherical_theta(wp), phi = spherical_phi(wp);
            Spec scl = l.map_mip < 0 ? Spec(1.0f) : sc->mipmaps[(size_t)l.map_mip].triangle(0, V2(phi * INV_TWO_PI, theta * INV_PI));
            r.value = l.L * scl / distance_squared(l.p_light, hit.p); r.valid = true; return r;
        }
        case L_POINT:    // point.rs:83-93
            r.wi = normalize(l.p_light - hit.p); r.pdf = 1.0f; r.vp = l.p_light;
            r.value = l.L / distance_squared(l.p_light, hit.p); r.valid = true; return r;
        case L_AREA: {   // diffuse.rs:114-129 -> Shape::sample_solid_angle (shape.rs:64-84) -> Triangle::sample (triangle.rs:918-949)
            if (l.sphere >= 0) {   // Sphere::sample_solid_angle (sphere.rs:344-410): uniform over the cone the sphere subtends, or over its area from inside
                const Sphere& sp = s.spheres[(size_t)l.sphere];
                const V3 p_center = sp.o2w.point(V3(0, 0, 0));
                V3 p, n, p_error; Float pdf;
                const V3 p_origin = offset_origin(hit.p, hit.p_error, hit.n, p_center - hit.p);
                if (distance_squared(p_origin, p_center) <= sp.radius * sp.radius) {
                    const Float z = 1.0f - 2.0f * u.x, rr = std::sqrt(pmax(0.0f, 1.0f - z * z)), phi = TWO_PI * u.y;   // uniform_sample_sphere
                    V3 p_obj = sp.radius * V3(rr * o_cos(phi), rr * o_sin(phi), z);
                    n = normalize(sp.o2w.normal(p_obj));
                    if (sp.reverse_orientation) n = n * -1.0f;
                    p_obj = p_obj * (sp.radius / distance(p_obj, V3(0, 0, 0)));
                    V3 pe; p = sp.o2w.point_with_abs_error(p_obj, gamma_n(5) * vabs(p_obj), pe); p_error = pe;
                    pdf = 1.0f / (sp.phi_max * sp.radius * (sp.z_max - sp.z_min));   // 1 / area()
                    V3 wi = p - hit.p;
                    if (length_squared(wi) == 0.0f) pdf = 0.0f;
                    else { wi = normalize(wi); pdf *= distance_squared(hit.p, p) / abs_dot(n, -wi); }
                    if (std::isinf(pdf)) pdf = 0.0f;
                } else {
                    const Float dc = distance(hit.p, p_center), inv_dc = 1.0f / dc;
                    const V3 wc = (p_center - hit.p) * inv_dc;
                    V3 wc_x, wc_y; coordinate_system(wc, wc_x, wc_y);
                    const Float sin_theta_max = sp.radius * inv_dc, sin_theta_max2 = sin_theta_max * sin_theta_max, inv_sin_theta_max = 1.0f / sin_theta_max;
                    const Float cos_theta_max = std::sqrt(pmax(0.0f, 1.0f - sin_theta_max2));
                    Float cos_theta = (cos_theta_max - 1.0f) * u.x + 1.0f, sin_theta2 = 1.0f - cos_theta * cos_theta;
                    if (sin_theta_max2 < 0.00068523f) { sin_theta2 = sin_theta_max2 * u.x; cos_theta = std::sqrt(1.0f - sin_theta2); }
                    const Float cos_alpha = sin_theta2 * inv_sin_theta_max + cos_theta * std::sqrt(pmax(0.0f, 1.0f - sin_theta2 * inv_sin_theta_max * inv_sin_theta_max));
                    const Float sin_alpha = std::sqrt(pmax(0.0f, 1.0f - cos_alpha * cos_alpha));
                    const Float phi = u.y * TWO_PI;
                    const V3 n_world = (sin_alpha * o_cos(phi)) * (-wc_x) + (sin_alpha * o_sin(phi)) * (-wc_y) + cos_alpha * (-wc);   // spherical_direction_in_coord_frame
                    p = p_center + sp.radius * n_world;
                    p_error = gamma_n(5) * vabs(p);
                    n = n_world;
                    if (sp.reverse_orientation) n = n * -1.0f;
                    pdf = 1.0f / (TWO_PI * (1.0f - cos_theta_max));   // uniform_cone_pdf
                }
                V3 wi2 = p - hit.p; const Float l2 = length_squared(wi2);   // DiffuseAreaLight::sample_li (diffuse.rs:114-129)
                if (pdf == 0.0f || l2 == 0.0f) return r;
                wi2 = wi2 / std::sqrt(l2);
                r.wi = wi2; r.pdf = pdf; r.value = area_L(l, n, -wi2); r.vp = p; r.vperr = p_error; r.vn = n; r.valid = true; return r;
            }
            uint32_t prim = l.prim; const Mesh& m = s.mesh_of(prim);
            uint32_t i0 = s.idx[3 * prim], i1 = s.idx[3 * prim + 1], i2 = s.idx[3 * prim + 2];
            V3 p0 = s.P[i0], p1 = s.P[i1], p2 = s.P[i2];
            V2 b = uniform_sample_triangle(u);
            V3 p = b.x * p0 + b.y * p1 + (1.0f - b.x - b.y) * p2;
            V3 n = normalize(cross(p1 - p0, p2 - p0));
            if (m.has_n) {
                V3 ns = b.x * s.N[i0] + b.y * s.N[i1] + (1.0f - b.x - b.y) * s.N[i2];
                n = face_forward(n, ns);
            } else if (m.reverse_orientation ^ m.swaps_handedness) n = n * -1.0f;
            V3 a0 = vabs(b.x * p0), a1 = vabs(b.y * p1), a2 = vabs((1.0f - b.x - b.y) * p2);
            V3 p_abs_sum = a0 + a1 + a2;
            V3 p_error = gamma_n(6) * V3(p_abs_sum.x, p_abs_sum.y, p_abs_sum.z);
            Float pdf = 1.0f / l.area;
            V3 wi = p - hit.p;
            if (length_squared(wi) == 0.0f) pdf = 0.0f;
            else {
                wi = normalize(wi);
                pdf *= distance_squared(hit.p, p) / abs_dot(n, -wi);
                if (std::isinf(pdf)) pdf = 0.0f;
            }
            V3 wi2 = p - hit.p; Float l2 = length_squared(wi2);
            if (pdf == 0.0f || l2 == 0.0f) return r;
            wi2 = wi2 / std::sqrt(l2);
            r.wi = wi2; r.pdf = pdf; r.value = area_L(l, n, -wi2); r.vp = p; r.vperr = p_error; r.vn = n; r.valid = true; return r;
        }
        }
        return r;
    }
    Float light_pdf_li(const Light& l, const SurfaceHit& hit, V3 wi) const {
        const Scene& s = *sc;
        if (l.type == L_INFINITE) {  // infinite.rs:201-211
            V3 w = l.l2w.inv().vector(wi);
            Float theta = spherical_theta(w), phi = spherical_phi(w), sin_theta = o_sin(theta);
            if (sin_theta == 0.0f) return 0.0f;
            V2 p(phi * INV_TWO_PI, theta * INV_PI);  // Distribution2D::pdf (distribution_2d.rs:51-65)
            if (l.map_mip >= 0) {
                size_t iu = pclamp<size_t>(f2usize(p.x * (Float)l.dw), 0, (size_t)l.dw - 1), iv = pclamp<size_t>(f2usize(p.y * (Float)l.dh), 0, (size_t)l.dh - 1);
                return (l.d_cond_func[iv * (size_t)l.dw + iu] / l.d_marg_int) / (TWO_PI * PI * sin_theta);
            }
            size_t iu = pclamp<size_t>(f2usize(p.x * 2.0f), 0, 1), iv = pclamp<size_t>(f2usize(p.y * 2.0f), 0, 1);
            return (l.cond_func[iv][iu] / l.marg_int) / (TWO_PI * PI * sin_theta);
        }
        if (l.type == L_AREA) {  // Shape::pdf_solid_angle (shape.rs:86-107)
            Ray ray = spawn_ray(hit.p, hit.p_error, hit.n, hit.time, wi);
            TriHit h;
            if (!s.tri_intersect(ray, l.prim, false, false, h)) return 0.0f;
            SurfaceHit li = make_surface_hit(ray, l.prim, h);
            Float pdf = distance_squared(hit.p, li.p) / (abs_dot(li.n, -wi) * l.area);
            return std::isinf(pdf) ? 0.0f : pdf;
        }
        return 0.0f;
    }
    Spec light_power(const Light& l) const {
        const Scene& s = *sc;
        switch (l.type) {
        case L_INFINITE:  // infinite.rs:176-183
            if (l.map_mip >= 0) return PI * s.world_radius * s.world_radius * s.mipmaps[(size_t)l.map_mip].lookup_triangle_host(V2(0.5f, 0.5f), 0.5f);
            return PI * s.world_radius * s.world_radius * infinite_lookup(l.L, V2(0.5f, 0.5f));
        case L_DISTANT: return l.L * PI * s.world_radius * s.world_radius;                                      // distant.rs:98-101
        case L_SPOT: return l.L * TWO_PI * (1.0f - 0.5f * (l.cos_falloff_start + l.cos_total_width));                    // spot.rs:86-88
        case L_POINT: return (4.0f * PI) * l.L;                                                                 // point.rs:95-97
        case L_PROJECTION: {   // projection.rs:193-203
            Spec sp = l.map_mip < 0 ? Spec(1.0f) : s.mipmaps[(size_t)l.map_mip].lookup_triangle_host(V2(0.5f, 0.5f), 0.5f);
            return sp * l.L * TWO_PI * (1.0f - l.cos_total_width);
        }
        case L_GONIO: {        // goniometric.rs:115-126
            Spec sp = l.map_mip < 0 ? Spec(1.0f) : s.mipmaps[(size_t)l.map_mip].lookup_triangle_host(V2(0.5f, 0.5f), 0.5f);
            return (4.0f * PI) * l.L * sp;
        }
        default: return (l.two_sided ? 2.0f : 1.0f) * l.L * l.area * PI;                                        // diffuse.rs:131-134
        }
    }

    // ---- BSDF (core/src/reflection/bsdf.rs) over the lobes of a material ------------------------------------------------
    struct BSDF {
        V3 ns, ng, ss, ts; const Lobe* lobes = nullptr; int n = 0; Float eta = 1.0f;
        bool null = false;   // compute_scattering_functions left `bsdf` at None for this hit (TranslucentMaterial where reflect and transmit are both black, translucent.rs:72-74)
        V3 w2l(V3 v) const { return V3(dot(v, ss), dot(v, ts), dot(v, ns)); }
        V3 l2w(V3 v) const {
            return V3(ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z, ss.z * v.x + ts.z * v.y + ns.z * v.z);
        }
        // reflection/common.rs
        static Float cos2(V3 w) { return w.z * w.z; }
        static Float sin2(V3 w) { return pmax(0.0f, 1.0f - cos2(w)); }
        static Float sinth(V3 w) { return std::sqrt(sin2(w)); }
        static Float tanth(V3 w) { return sinth(w) / w.z; }
        static Float tan2(V3 w) { return sin2(w) / cos2(w); }
        static Float cosphi(V3 w) { Float s = sinth(w); return s == 0.0f ? 1.0f : pclamp(w.x / s, -1.0f, 1.0f); }
        static Float sinphi(V3 w) { Float s = sinth(w); return s == 0.0f ? 0.0f : pclamp(w.y / s, -1.0f, 1.0f); }
        static Float cos2phi(V3 w) { Float c = cosphi(w); return c * c; }
        static Float sin2phi(V3 w) { Float c = sinphi(w); return c * c; }
        static bool same_hemisphere(V3 a, V3 b) { return a.z * b.z > 0.0f; }
        static V3 reflect(V3 wo, V3 n) { return -wo + 2.0f * dot(wo, n) * n; }
        static bool refract(V3 wi, V3 n, Float eta, V3& wt) {  // common.rs:103-118
            Float cos_i = dot(n, wi);
            Float sin2_i = pmax(0.0f, 1.0f - cos_i * cos_i);
            Float sin2_t = eta * eta * sin2_i;
            if (sin2_t >= 1.0f) return false;
            Float cos_t = std::sqrt(1.0f - sin2_t);
            wt = eta * -wi + (eta * cos_i - cos_t) * n;
            return true;
        }
        // fresnel.rs:135-170 (f32::max keeps the non-NaN operand; the operands here are never NaN)
        static Float fr_dielectric(Float cos_i, Float eta_i, Float eta_t) {
            cos_i = pclamp(cos_i, -1.0f, 1.0f);
            if (!(cos_i > 0.0f)) { std::swap(eta_i, eta_t); cos_i = pabs(cos_i); }
            Float sin_i = std::sqrt(std::fmax(0.0f, 1.0f - cos_i * cos_i));
            Float sin_t = eta_i / eta_t * sin_i;
            if (sin_t >= 1.0f) return 1.0f;
            Float cos_t = std::sqrt(std::fmax(0.0f, 1.0f - sin_t * sin_t));
            Float r_parl = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_i * cos_t));
            Float r_perp = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
            return (r_parl * r_parl + r_perp * r_perp) / 2.0f;
        }
        // fresnel.rs:172-196 — as written there: `sin_theta_i_2 = 1.0 - cos_theta_i` (not 1 - cos^2), quirk B11
        static Spec fr_conductor(Float cos_i, Spec eta_i, Spec eta_t, Spec k) {
            cos_i = pclamp(cos_i, -1.0f, 1.0f);
            Spec eta = eta_t / eta_i, eta_k = k / eta_i;
            Float cos2_i = cos_i * cos_i, sin2_i = 1.0f - cos_i;
            Spec eta_2 = eta * eta, eta_k_2 = eta_k * eta_k;
            Spec t0 = eta_2 - eta_k_2 - Spec(sin2_i);
            Spec a2_plus_b2 = spec_sqrt(t0 * t0 + 4.0f * eta_2 * eta_k_2);
            Spec t1 = a2_plus_b2 + Spec(cos2_i);
            Spec a = spec_sqrt(0.5f * (a2_plus_b2 + t0));
            Spec t2 = 2.0f * cos_i * a;
            Spec rs = (t1 - t2) / (t1 + t2);
            Spec t3 = cos2_i * a2_plus_b2 + Spec(sin2_i * sin2_i);
            Spec t4 = t2 * sin2_i;
            Spec rp = rs * (t3 - t4) / (t3 + t4);
            return 0.5f * (rp + rs);
        }
        static Spec fresnel_eval(const Lobe& l, Float cos_i) {
            if (l.fresnel == FR_DIEL) return Spec(fr_dielectric(cos_i, l.eta_a, l.eta_b));
            if (l.fresnel == FR_COND) return fr_conductor(pabs(cos_i), l.c_eta_i, l.c_eta_t, l.c_k);
            return Spec(1.0f);
        }
        // microfacet/trowbridge_reitz.rs
        static Float tr_d(const Lobe& l, V3 wh) {
            Float t2 = tan2(wh);
            if (std::isinf(t2)) return 0.0f;
            Float cos4 = cos2(wh) * cos2(wh);
            Float e = (cos2phi(wh) / (l.ax * l.ax) + sin2phi(wh) / (l.ay * l.ay)) * t2;
            return 1.0f / (PI * l.ax * l.ay * cos4 * (1.0f + e) * (1.0f + e));
        }
        static Float tr_lambda(const Lobe& l, V3 w) {
            Float att = pabs(tanth(w));
            if (std::isinf(att)) return 0.0f;
            Float alpha = std::sqrt(cos2phi(w) * l.ax * l.ax + sin2phi(w) * l.ay * l.ay);
            Float a2t2 = (alpha * att) * (alpha * att);
            return (-1.0f + std::sqrt(1.0f + a2t2)) / 2.0f;
        }
        static Float tr_g1(const Lobe& l, V3 w) { return 1.0f / (1.0f + tr_lambda(l, w)); }
        static Float tr_g(const Lobe& l, V3 wo, V3 wi) { return 1.0f / (1.0f + tr_lambda(l, wo) + tr_lambda(l, wi)); }
        static Float tr_pdf(const Lobe& l, V3 wo, V3 wh) { return tr_d(l, wh) * tr_g1(l, wo) * abs_dot(wo, wh) / pabs(wo.z); }  // sample_visible_area = true
        static void tr_sample_11(Float cos_theta, Float u1, Float u2, Float& slope_x, Float& slope_y) {
            if (cos_theta > 0.9999f) {
                Float r = std::sqrt(u1 / (1.0f - u1));
                Float phi = TWO_PI * u2;
                slope_x = r * o_cos(phi); slope_y = r * o_sin(phi);
                return;
            }
            Float sin_theta = std::sqrt(pmax(0.0f, 1.0f - cos_theta * cos_theta));
            Float tan_theta = sin_theta / cos_theta;
            Float a = 1.0f / tan_theta;
            Float g1 = 2.0f / (1.0f + std::sqrt(1.0f + 1.0f / (a * a)));
            a = 2.0f * u1 / g1 - 1.0f;
            Float tmp = 1.0f / (a * a - 1.0f);
            if (tmp > 1e10f) tmp = 1e10f;
            Float b = tan_theta;
            Float d = std::sqrt(pmax(b * b * tmp * tmp - (a * a - b * b) * tmp, 0.0f));
            Float sx1 = b * tmp - d, sx2 = b * tmp + d;
            slope_x = (a < 0.0f || sx2 > 1.0f / tan_theta) ? sx1 : sx2;
            Float sgn;
            if (u2 > 0.5f) { sgn = 1.0f; u2 = 2.0f * (u2 - 0.5f); } else { sgn = -1.0f; u2 = 2.0f * (0.5f - u2); }
            Float z = (u2 * (u2 * (u2 * 0.27385f - 0.73369f) + 0.46341f)) / (u2 * (u2 * (u2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
            slope_y = sgn * z * std::sqrt(1.0f + slope_x * slope_x);
        }
        static V3 tr_sample_wh(const Lobe& l, V3 wo, V2 u) {
            bool flip = wo.z < 0.0f;
            V3 wi = flip ? -wo : wo;
            V3 ws = normalize(V3(l.ax * wi.x, l.ay * wi.y, wi.z));
            Float sx, sy; tr_sample_11(ws.z, u.x, u.y, sx, sy);
            Float tmp = cosphi(ws) * sx - sinphi(ws) * sy;
            sy = sinphi(ws) * sx + cosphi(ws) * sy;
            sx = tmp;
            sx *= l.ax; sy *= l.ay;
            V3 wh = normalize(V3(-sx, -sy, 1.0f));
            return flip ? -wh : wh;
        }
        // ---- per-lobe f / pdf / sample_f ----------------------------------------------------------------------------------
        static Float pow5(Float v) { return (v * v) * (v * v) * v; }  // pbrt/common.rs:345-347
        static Spec lobe_f(const Lobe& l, V3 wo, V3 wi) {  // ScaledBxDF::f (scaled_bxdf.rs:27-29), innermost wrapper first
            Spec f = lobe_f_raw(l, wo, wi);
            for (int k = 0; k < l.n_scale; k++) f = l.scale[k] * f;
            return f;
        }
        static Spec lobe_f_raw(const Lobe& l, V3 wo, V3 wi) {
            switch (l.kind) {
            case LK_LAMBERT: return l.r * INV_PI;  // lambertian_reflection.rs:38-40
            case LK_LAMBERT_T: return l.t * INV_PI;  // lambertian_transmission.rs:24-26
            case LK_FRESNEL_BLEND: {  // fresnel_blend.rs:32-50 (rd = l.r, rs = l.t)
                Spec diffuse = (28.0f / (23.0f * PI)) * l.r * (Spec(1.0f) - l.t) * (1.0f - pow5(1.0f - 0.5f * pabs(wi.z))) * (1.0f - pow5(1.0f - 0.5f * pabs(wo.z)));
                V3 wh = wi + wo;
                if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return Spec(0.0f);
                wh = normalize(wh);
                Spec schlick = l.t + (Spec(1.0f) - l.t) * pow5(1.0f - dot(wi, wh));
                Spec specular = tr_d(l, wh) / (4.0f * abs_dot(wi, wh) * pmax(pabs(wi.z), pabs(wo.z))) * schlick;
                return diffuse + specular;
            }
            case LK_OREN: {  // oren_nayar.rs:46-72
                Float sin_i = sinth(wi), sin_o = sinth(wo), max_cos = 0.0f;
                if (sin_i > 1e-4f && sin_o > 1e-4f) {
                    Float d_cos = cosphi(wi) * cosphi(wo) + sinphi(wi) * sinphi(wo);
                    max_cos = pmax(0.0f, d_cos);
                }
                Float aco = pabs(wo.z), aci = pabs(wi.z), sin_alpha, tan_beta;
                if (aci > aco) { sin_alpha = sin_o; tan_beta = sin_i / aci; }
                else { sin_alpha = sin_i; tan_beta = sin_o / aco; }
                return l.r * INV_PI * (l.a + l.b * max_cos * sin_alpha * tan_beta);
            }
            case LK_MICRO_R: {  // microfacet_reflection.rs:34-52
                Float cos_o = pabs(wo.z), cos_i = pabs(wi.z);
                V3 wh = wi + wo;
                if ((cos_i == 0.0f || cos_o == 0.0f) || (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f)) return Spec(0.0f);
                wh = normalize(wh);
                Spec f = fresnel_eval(l, dot(wi, face_forward(wh, V3(0.0f, 0.0f, 1.0f))));
                return l.r * tr_d(l, wh) * tr_g(l, wo, wi) * f / (4.0f * cos_i * cos_o);
            }
            case LK_MICRO_T: {  // microfacet_transmission.rs:46-95
                if (same_hemisphere(wo, wi)) return Spec(0.0f);
                Float cos_o = wo.z, cos_i = wi.z;
                if (cos_i == 0.0f || cos_o == 0.0f) return Spec(0.0f);
                Float eta = wo.z > 0.0f ? l.eta_b / l.eta_a : l.eta_a / l.eta_b;
                V3 wh = normalize(wo + wi * eta);
                if (wh.z < 0.0f) wh = -wh;
                if (dot(wo, wh) * dot(wi, wh) > 0.0f) return Spec(0.0f);
                Spec f(fr_dielectric(dot(wo, wh), l.eta_a, l.eta_b));
                Float sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
                Float factor = 1.0f / eta;  // TransportMode::Radiance
                return (Spec(1.0f) - f) * l.t *
                       pabs(tr_d(l, wh) * tr_g(l, wo, wi) * eta * eta * abs_dot(wi, wh) * abs_dot(wo, wh) * factor * factor / (cos_i * cos_o * sqrt_denom * sqrt_denom));
            }
            default: return Spec(0.0f);  // specular lobes scatter nothing outside their delta direction
            }
        }
        static Float lobe_pdf(const Lobe& l, V3 wo, V3 wi) {
            switch (l.kind) {
            case LK_LAMBERT: case LK_OREN: return same_hemisphere(wo, wi) ? pabs(wi.z) * INV_PI : 0.0f;  // reflection/mod.rs:160-166
            case LK_LAMBERT_T: return !same_hemisphere(wo, wi) ? pabs(wi.z) * INV_PI : 0.0f;           // lambertian_transmission.rs:36-42
            case LK_FRESNEL_BLEND: {  // fresnel_blend.rs:77-85
                if (!same_hemisphere(wo, wi)) return 0.0f;
                V3 wh = normalize(wo + wi);
                Float pdf_wh = tr_pdf(l, wo, wh);
                return 0.5f * (pabs(wi.z) * INV_PI + pdf_wh / (4.0f * dot(wo, wh)));
            }
            case LK_MICRO_R: {
                if (!same_hemisphere(wo, wi)) return 0.0f;
                V3 wh = normalize(wo + wi);
                return tr_pdf(l, wo, wh) / (4.0f * dot(wo, wh));
            }
            case LK_MICRO_T: {
                if (same_hemisphere(wo, wi)) return 0.0f;
                Float eta = wo.z > 0.0f ? l.eta_b / l.eta_a : l.eta_a / l.eta_b;
                V3 wh = normalize(wo + wi * eta);
                if (dot(wo, wh) * dot(wi, wh) > 0.0f) return 0.0f;
                Float sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
                Float dwh_dwi = pabs((eta * eta * dot(wi, wh)) / (sqrt_denom * sqrt_denom));
                return tr_pdf(l, wo, wh) * dwh_dwi;
            }
            default: return 0.0f;
            }
        }
        // returns the sampled BxDFType; f/pdf/wi zero where the reference returns BxDFSample::from(type)
        static int lobe_sample_f(const Lobe& l, V3 wo, V2 u, Spec& f, Float& pdf, V3& wi) {  // ScaledBxDF::sample_f (scaled_bxdf.rs:31-35)
            int st = lobe_sample_f_raw(l, wo, u, f, pdf, wi);
            for (int k = 0; k < l.n_scale; k++) f = l.scale[k] * f;
            return st;
        }
        static int lobe_sample_f_raw(const Lobe& l, V3 wo, V2 u, Spec& f, Float& pdf, V3& wi) {
            f = Spec(0.0f); pdf = 0.0f; wi = V3();
            switch (l.kind) {
            case LK_LAMBERT_T: {  // lambertian_transmission.rs:28-35
                wi = cosine_sample_hemisphere(u);
                if (wo.z > 0.0f) wi.z *= -1.0f;
                pdf = lobe_pdf(l, wo, wi); f = lobe_f_raw(l, wo, wi);
                return l.type;
            }
            case LK_FRESNEL_BLEND: {  // fresnel_blend.rs:52-75
                if (u.x < 0.5f) {
                    u.x = pmin(2.0f * u.x, ONE_MINUS_EPSILON);
                    wi = cosine_sample_hemisphere(u);
                    if (wo.z < 0.0f) wi.z *= -1.0f;
                } else {
                    u.x = pmin(2.0f * (u.x - 0.5f), ONE_MINUS_EPSILON);
                    V3 wh = tr_sample_wh(l, wo, u);
                    wi = reflect(wo, wh);
                    if (!same_hemisphere(wo, wi)) return l.type;  // f = 0, pdf = 0
                }
                pdf = lobe_pdf(l, wo, wi); f = lobe_f_raw(l, wo, wi);
                return l.type;
            }
            case LK_LAMBERT: case LK_OREN: {  // reflection/mod.rs:132-141
                wi = cosine_sample_hemisphere(u);
                if (wo.z < 0.0f) wi.z *= -1.0f;
                pdf = lobe_pdf(l, wo, wi); f = lobe_f_raw(l, wo, wi);
                return l.type;
            }
            case LK_SPEC_R: {  // specular_reflection.rs:38-44
                wi = V3(-wo.x, -wo.y, wo.z); pdf = 1.0f;
                f = fresnel_eval(l, wi.z) * l.r / pabs(wi.z);
                return l.type;
            }
            case LK_SPEC_T: {  // specular_transmission.rs:45-64
                bool entering = wo.z > 0.0f;
                Float eta_i = entering ? l.eta_a : l.eta_b, eta_t = entering ? l.eta_b : l.eta_a;
                V3 wt;
                if (!refract(wo, face_forward(V3(0.0f, 0.0f, 1.0f), wo), eta_i / eta_t, wt)) return l.type;
                wi = wt; pdf = 1.0f;
                Spec ft = l.t * (Spec(1.0f) - Spec(fr_dielectric(wi.z, l.eta_a, l.eta_b)));
                ft = ft * ((eta_i * eta_i) / (eta_t * eta_t));  // TransportMode::Radiance
                f = ft / pabs(wi.z);
                return l.type;
            }
            case LK_FRESNEL_SPEC: {  // fresnel_specular.rs:42-79
                Float fr = fr_dielectric(wo.z, l.eta_a, l.eta_b);
                if (u.x < fr) {
                    wi = V3(-wo.x, -wo.y, wo.z); pdf = fr;
                    f = fr * l.r / pabs(wi.z);
                    return BX_SPEC | BX_REFL;
                }
                bool entering = wo.z > 0.0f;
                Float eta_i = entering ? l.eta_a : l.eta_b, eta_t = entering ? l.eta_b : l.eta_a;
                V3 wt;
                if (!refract(wo, face_forward(V3(0.0f, 0.0f, 1.0f), wo), eta_i / eta_t, wt)) return BX_SPEC | BX_TRANS;
                wi = wt;
                Spec ft = l.t * (1.0f - fr);
                ft = ft * ((eta_i * eta_i) / (eta_t * eta_t));
                pdf = 1.0f - fr;
                f = ft / pabs(wi.z);
                return BX_SPEC | BX_TRANS;
            }
            case LK_MICRO_R: {  // microfacet_reflection.rs:54-76
                if (wo.z == 0.0f) return l.type;
                V3 wh = tr_sample_wh(l, wo, u);
                if (dot(wo, wh) < 0.0f) return l.type;
                wi = reflect(wo, wh);
                if (!same_hemisphere(wo, wi)) return l.type;  // f = 0, pdf = 0, wi kept
                pdf = tr_pdf(l, wo, wh) / (4.0f * dot(wo, wh));
                f = lobe_f_raw(l, wo, wi);
                return l.type;
            }
            case LK_MICRO_T: {  // microfacet_transmission.rs:97-120
                if (wo.z == 0.0f) return l.type;
                V3 wh = tr_sample_wh(l, wo, u);
                if (dot(wo, wh) < 0.0f) return l.type;
                Float eta = wo.z > 0.0f ? l.eta_a / l.eta_b : l.eta_b / l.eta_a;
                V3 wt;
                if (!refract(wo, wh, eta, wt)) return l.type;
                wi = wt;
                pdf = lobe_pdf(l, wo, wi); f = lobe_f_raw(l, wo, wi);
                return l.type;
            }
            }
            return l.type;
        }
        static bool matches(const Lobe& l, int flags) { return (l.type & flags) == l.type; }
        int num_components(int flags) const { int c = 0; for (int i = 0; i < n; i++) if (matches(lobes[i], flags)) c++; return c; }
        // BSDF::f (bsdf.rs:133-158)
        Spec f(V3 wo_w, V3 wi_w, int flags = BX_ALL) const {
            V3 wi = w2l(wi_w), wo = w2l(wo_w);
            if (wo.z == 0.0f) return Spec(0.0f);
            bool reflect = dot(wi_w, ng) * dot(wo_w, ng) > 0.0f;
            Spec out(0.0f);
            for (int i = 0; i < n; i++) {
                const Lobe& l = lobes[i];
                if (matches(l, flags) && ((reflect && (l.type & BX_REFL)) || (!reflect && (l.type & BX_TRANS)))) out += lobe_f(l, wo, wi);
            }
            return out;
        }
        Float pdf(V3 wo_w, V3 wi_w, int flags = BX_ALL) const {  // bsdf.rs:331-356
            if (n == 0) return 0.0f;
            V3 wo = w2l(wo_w), wi = w2l(wi_w);
            if (wo.z == 0.0f) return 0.0f;
            int matching = 0; Float p = 0.0f;
            for (int i = 0; i < n; i++) if (matches(lobes[i], flags)) { matching++; p += lobe_pdf(lobes[i], wo, wi); }
            return matching > 0 ? p / (Float)matching : 0.0f;
        }
        // BSDF::sample_f (bsdf.rs:160-292); returns false where the reference returns BxDFSample::default()
        bool sample_f(V3 wo_w, V2 u, Spec& f_out, Float& pdf_out, V3& wi_out, int flags = BX_ALL, int* sampled_type = nullptr) const {
            f_out = Spec(0.0f); pdf_out = 0.0f; wi_out = V3();
            if (sampled_type) *sampled_type = 0;
            int matching = num_components(flags);
            if (matching == 0) return false;
            size_t comp = pmin<size_t>(f2usize(std::floor(u.x * (Float)matching)), (size_t)matching - 1);
            int idx = -1; size_t count = comp;
            for (int i = 0; i < n; i++) if (matches(lobes[i], flags)) { if (count == 0) { idx = i; break; } count--; }
            V2 ur(pmin(u.x * (Float)matching - (Float)comp, ONE_MINUS_EPSILON), u.y);
            V3 wo = w2l(wo_w);
            if (wo.z == 0.0f) return false;
            Spec fv; Float pdf; V3 wi;
            int st = lobe_sample_f(lobes[idx], wo, ur, fv, pdf, wi);
            if (pdf == 0.0f) return false;
            V3 wi_w = l2w(wi);
            if (!(st & BX_SPEC) && matching > 1)
                for (int i = 0; i < n; i++) if (i != idx && matches(lobes[i], flags)) pdf += lobe_pdf(lobes[i], wo, wi);
            if (matching > 1) pdf /= (Float)matching;
            if (!(st & BX_SPEC)) {
                bool reflect = dot(wi_w, ng) * dot(wo_w, ng) > 0.0f;
                fv = Spec(0.0f);
                for (int i = 0; i < n; i++) {
                    const Lobe& l = lobes[i];
                    if (matches(l, flags) && ((reflect && (l.type & BX_REFL)) || (!reflect && (l.type & BX_TRANS)))) fv += lobe_f(l, wo, wi);
                }
            }
            f_out = fv; pdf_out = pdf; wi_out = wi_w;
            if (sampled_type) *sampled_type = st;
            return true;
        }
    };
    // Material::bump (core/src/material.rs:62-101) + set_shading_geometry(.., false) (surface_interaction.rs:152-173)
    void bump(SurfaceHit& si) const {
        const Material& m = sc->materials[sc->mesh_of(si.prim).material];
        if (m.bump_tex < 0) return;
        TexCtx c; c.uv = si.uv; c.dudx = si.dudx; c.dvdx = si.dvdx; c.dudy = si.dudy; c.dvdy = si.dvdy; c.p = si.p; c.dpdx = si.dpdx; c.dpdy = si.dpdy;
        Float du = 0.5f * (std::fabs(si.dudx) + std::fabs(si.dudy));
        if (du == 0.0f) du = 0.0005f;
        TexCtx cu = c; cu.p = si.p + du * si.dpdu_s; cu.uv = V2(si.uv.x + du, si.uv.y + 0.0f);
        Float u_displace = tex_eval(sc->textures, sc->mipmaps, m.bump_tex, cu).c[0];
        Float dv = 0.5f * (std::fabs(si.dvdx) + std::fabs(si.dvdy));
        if (dv == 0.0f) dv = 0.0005f;
        TexCtx cv = c; cv.p = si.p + dv * si.dpdv_s; cv.uv = V2(si.uv.x + 0.0f, si.uv.y + dv);
        Float v_displace = tex_eval(sc->textures, sc->mipmaps, m.bump_tex, cv).c[0];
        Float displace = tex_eval(sc->textures, sc->mipmaps, m.bump_tex, c).c[0];
        V3 dpdu = si.dpdu_s + (u_displace - displace) / du * si.ns + displace * si.dndu_s;
        V3 dpdv = si.dpdv_s + (v_displace - displace) / dv * si.ns + displace * si.dndv_s;
        si.ns = face_forward(normalize(cross(dpdu, dpdv)), si.n);
        si.dpdu_s = dpdu; si.dpdv_s = dpdv;
    }
    // `local` receives the per-hit lobe list of a textured material; the returned BSDF points at it, so it must outlive the BSDF
    BSDF make_bsdf(const SurfaceHit& si, Lobe* local) const {
        const Material& m = sc->materials[sc->mesh_of(si.prim).material];
        BSDF b; b.ns = si.ns; b.ng = si.n; b.ss = normalize(si.dpdu_s); b.ts = cross(b.ns, b.ss);  // bsdf.rs:100-116
        b.lobes = m.lobes.data(); b.n = (int)m.lobes.size(); b.eta = m.bsdf_eta;
        if (m.textured) {  // compute_scattering_functions evaluates the textures at this hit (matte.rs:63-71, plastic.rs:62-81, mirror.rs:53-57, substrate.rs:60-80)
            TexCtx c; c.uv = si.uv; c.dudx = si.dudx; c.dvdx = si.dvdx; c.dudy = si.dudy; c.dvdy = si.dvdy; c.p = si.p; c.dpdx = si.dpdx; c.dpdy = si.dpdy;
            int k = 0;
            Spec s1(1.0f), s2(0.0f), op(1.0f);
            if (m.amount_tex >= 0) { s1 = spec_clamp0(tex_eval(sc->textures, sc->mipmaps, m.amount_tex, c)); s2 = spec_clamp0(Spec(1.0f) - s1); }   // mix.rs:59-60
            if (m.opacity_tex >= 0) op = spec_clamp0(tex_eval(sc->textures, sc->mipmaps, m.opacity_tex, c));                                        // uber.rs:126
            Spec rt_r(0.0f), rt_t(0.0f);
            if (m.rt_mode) {   // translucent.rs:70-74
                rt_r = m.refl_tex >= 0 ? spec_clamp0(tex_eval(sc->textures, sc->mipmaps, m.refl_tex, c)) : m.raw_k[2];
                rt_t = m.trans_tex >= 0 ? spec_clamp0(tex_eval(sc->textures, sc->mipmaps, m.trans_tex, c)) : m.raw_k[3];
                if (rt_r.is_black() && rt_t.is_black()) { b.null = true; b.lobes = local; b.n = 0; return b; }
            }
            bool passthrough = false;
            const Float e_hit = m.index_tex >= 0 ? tex_eval(sc->textures, sc->mipmaps, m.index_tex, c).c[0] : 0.0f;   // glass.rs:102 / uber.rs:128
            for (const Lobe& tl : m.lobes) {
                Lobe l = tl;
                if (m.index_tex >= 0 && l.pre_mode != 4 && (l.kind == LK_FRESNEL_SPEC || l.fresnel == FR_DIEL)) l.eta_b = e_hit;   // every dielectric lobe but the opacity pass-through (built with 1, 1: uber.rs:134)
                if (l.sigma_tex >= 0) {  // matte.rs:64-70 + OrenNayar::new (oren_nayar.rs:28-39)
                    Float sig = pclamp(tex_eval(sc->textures, sc->mipmaps, l.sigma_tex, c).c[0], 0.0f, 90.0f);
                    if (sig == 0.0f) { l.kind = LK_LAMBERT; l.a = 0.0f; l.b = 0.0f; }
                    else { l.kind = LK_OREN; Float sg = to_radians(sig), s2o = sg * sg; l.a = 1.0f - (s2o / (2.0f * (s2o + 0.33f))); l.b = 0.45f * s2o / (s2o + 0.09f); }
                }
                bool is_specular = false;
                if (l.ax_tex >= 0 || l.ay_tex >= 0) {  // roughness textures, remapped per hit (trowbridge_reitz.rs:21-40)
                    const Float ur = l.ax_tex >= 0 ? tex_eval(sc->textures, sc->mipmaps, l.ax_tex, c).c[0] : l.ur_raw;
                    const Float vr = l.ay_tex >= 0 ? tex_eval(sc->textures, sc->mipmaps, l.ay_tex, c).c[0] : l.vr_raw;
                    is_specular = ur == 0.0f && vr == 0.0f;   // glass.rs:111, on the values as the textures give them
                    auto alpha_of = [&](int tex, Float raw, Float cur) {
                        if (tex < 0) return cur;
                        Float r = raw;
                        if (l.remap) { r = pmax(r, 1e-3f); Float x = o_log(r); r = 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x; }
                        return pmax(0.001f, r);
                    };
                    l.ax = alpha_of(l.ax_tex, ur, l.ax); l.ay = alpha_of(l.ay_tex, vr, l.ay);
                }
                if ((l.alt == 1 && !is_specular) || (l.alt == 2 && is_specular)) continue;   // glass.rs:112-141: FresnelSpecular, or the microfacet pair
                bool raw_black = false;   // translucent.rs:77-84, :87: the texel itself is tested, then multiplied by reflect / transmit
                if (l.pre_mode == 5) {   // translucent.rs:76-98 with reflect / transmit evaluated at the hit: `if !kd.is_black() { if !r.is_black() { add(r * kd) } .. }`
                    const bool refl = l.kind == LK_LAMBERT || l.kind == LK_MICRO_R;
                    const Spec A = refl ? rt_r : rt_t;
                    const int tex = refl ? l.r_tex : l.t_tex;
                    const Spec B = tex >= 0 ? spec_clamp0(tex_eval(sc->textures, sc->mipmaps, tex, c)) : l.pre;
                    if (A.is_black() || B.is_black()) continue;
                    (refl ? l.r : l.t) = A * B;
                    local[k++] = l;
                    continue;
                }
                if (l.pre_mode == 4) l.t = spec_clamp0(op * -1.0f + Spec(1.0f));    // (-op + ONE).clamp_default() (uber.rs:127)
                else if (l.pre_mode == 3) {   // op * k.evaluate(..).clamp_default() (uber.rs:141, :147, :169, :175)
                    const bool trans = l.kind == LK_SPEC_T;
                    const int tex = trans ? l.t_tex : l.r_tex;
                    const Spec base = tex >= 0 ? spec_clamp0(tex_eval(sc->textures, sc->mipmaps, tex, c)) : l.pre;
                    (trans ? l.t : l.r) = op * base;
                } else {
                    if (l.r_tex >= 0) { l.r = spec_clamp0(tex_eval(sc->textures, sc->mipmaps, l.r_tex, c)); if (l.pre_raw_test && l.r.is_black()) raw_black = true; if (l.has_pre) l.r = l.pre * l.r; }
                    if (l.t_tex >= 0) { l.t = spec_clamp0(tex_eval(sc->textures, sc->mipmaps, l.t_tex, c)); if (l.pre_raw_test && l.t.is_black()) raw_black = true; if (l.has_pre) l.t = l.pre * l.t; }
                }
                if (l.eta_tex >= 0) l.c_eta_t = tex_eval(sc->textures, sc->mipmaps, l.eta_tex, c);   // metal.rs:121-125: no clamp
                if (l.k_tex >= 0) l.c_k = tex_eval(sc->textures, sc->mipmaps, l.k_tex, c);
                if (l.amt_side) l.scale[l.amt_level] = l.amt_side == 1 ? s1 : s2;
                if (l.pre_raw_test) { if (!raw_black) local[k++] = l; continue; }   // an untextured lobe of this material exists because its constant passed the test at creation
                const bool keep = (l.kind == LK_FRESNEL_BLEND || l.kind == LK_FRESNEL_SPEC) ? !(l.r.is_black() && l.t.is_black())
                                : ((l.kind == LK_SPEC_T || l.kind == LK_MICRO_T || l.kind == LK_LAMBERT_T) ? !l.t.is_black() : !l.r.is_black());
                if (keep) { if (l.pre_mode == 4) passthrough = true; local[k++] = l; }
            }
            if (m.opacity_tex >= 0 && m.made_as == 1) b.eta = passthrough ? 1.0f : (m.index_tex >= 0 ? e_hit : m.bsdf_eta_alt);   // uber.rs:128-137
            b.lobes = local; b.n = k;
        }
        return b;
    }

    // ---- SpatialLightDistribution::new / lookup / compute_distribution (light_distrib/spatial.rs:57-245) ------------------
    void spatial_init(int max_voxels) {
        const Scene& s = *sc;
        const V3 blo = s.world_bound.pmin, bhi = s.world_bound.pmax;
        wb_lo[0] = blo.x; wb_lo[1] = blo.y; wb_lo[2] = blo.z; wb_hi[0] = bhi.x; wb_hi[1] = bhi.y; wb_hi[2] = bhi.z;
        Float diag[3] = {wb_hi[0] - wb_lo[0], wb_hi[1] - wb_lo[1], wb_hi[2] - wb_lo[2]};
        int ext = (diag[0] > diag[1] && diag[0] > diag[2]) ? 0 : (diag[1] > diag[2] ? 1 : 2);  // bounds3.rs:122-134
        Float bmax = diag[ext];
        for (int i = 0; i < 3; i++) {
            Float r = std::round(diag[i] / bmax * (Float)max_voxels);
            long long v = (r != r) ? 0 : (r <= 0.0f ? 0 : (r >= 9.2e18f ? (long long)9.2e18 : (long long)r));  // `as usize` saturates, NaN -> 0
            n_voxels[i] = (int)(v < 1 ? 1 : (v > (1 << 20) - 1 ? (1 << 20) - 1 : v));
        }
        size_t nv = (size_t)n_voxels[0] * n_voxels[1] * n_voxels[2];
        voxel_dist.clear(); voxel_dist.resize(nv);
        voxel_once = std::vector<std::once_flag>(nv);
        spatial_created = 0;
    }
    static Float lerp1(Float t, Float a, Float b) { return (1.0f - t) * a + t * b; }  // pbrt/common.rs:167-173
    void spatial_compute(const int pi[3], Dist1D& out) const {
        const Scene& s = *sc;
        Float lo[3], hi[3];
        for (int i = 0; i < 3; i++) {
            Float p0 = (Float)pi[i] / (Float)n_voxels[i], p1 = (Float)(pi[i] + 1) / (Float)n_voxels[i];
            Float a = lerp1(p0, wb_lo[i], wb_hi[i]), b = lerp1(p1, wb_lo[i], wb_hi[i]);
            lo[i] = pmin(a, b); hi[i] = pmax(a, b);  // Bounds3f::new
        }
        const size_t n_samples = 128, n_lights = s.lights.size();
        std::vector<Float> contrib(n_lights, 0.0f);
        for (size_t i = 0; i < n_samples; i++) {
            SurfaceHit intr{};
            intr.p = V3(lerp1(radical_inverse(0, i), lo[0], hi[0]), lerp1(radical_inverse(1, i), lo[1], hi[1]), lerp1(radical_inverse(2, i), lo[2], hi[2]));
            intr.time = 0.0f;
            V2 u(radical_inverse(3, i), radical_inverse(4, i));
            for (size_t j = 0; j < n_lights; j++) {
                LiSample li = light_sample_li(s.lights[j], intr, u);
                if (li.valid && li.pdf > 0.0f) contrib[j] += li.value.y() / li.pdf;
            }
        }
        Float sum = 0.0f;
        for (Float c : contrib) sum += c;
        Float avg = sum / (Float)(n_samples * n_lights);
        Float min_contrib = avg > 0.0f ? 0.001f * avg : 1.0f;
        for (Float& c : contrib) c = pmax(c, min_contrib);
        out.init(contrib);
    }
    void spatial_voxel_of(V3 p, int pi[3]) const {  // lookup's first half (:170-181); Bounds3::offset (bounds3.rs:153-168)
        Float o[3] = {p.x - wb_lo[0], p.y - wb_lo[1], p.z - wb_lo[2]};
        for (int i = 0; i < 3; i++) {
            if (wb_hi[i] > wb_lo[i]) o[i] /= wb_hi[i] - wb_lo[i];
            int v = f2i32(o[i] * (Float)n_voxels[i]);
            pi[i] = pclamp(v, 0, n_voxels[i] - 1);
        }
    }
    const Dist1D& spatial_lookup(V3 p) {
        int pi[3]; spatial_voxel_of(p, pi);
        size_t idx = ((size_t)pi[0] * n_voxels[1] + pi[1]) * n_voxels[2] + pi[2];
        std::call_once(voxel_once[idx], [&] {
            std::unique_ptr<Dist1D> d(new Dist1D());
            spatial_compute(pi, *d);
            voxel_dist[idx] = std::move(d);
            std::lock_guard<std::mutex> g(spatial_mu); spatial_created++;
        });
        return *voxel_dist[idx];
    }

    // ---- estimate_direct + uniform_sample_one_light (core/src/integrator/common.rs:89-299) -----------------------
    template <class S> Spec uniform_sample_one_light(const SurfaceHit& hit, const BSDF& bsdf, S& sampler) {
        const Scene& s = *sc;
        size_t n_lights = s.lights.size();
        if (n_lights == 0) return Spec(0.0f);
        Float sample = sampler.get_1d(), light_pdf;
        const Dist1D& distrib = spatial ? spatial_lookup(hit.p) : light_distrib;  // LightDistribution::lookup(&isect.hit.p) (path.rs:156-157)
        size_t light_num = distrib.sample_discrete(sample, light_pdf);
        if (light_pdf == 0.0f) return Spec(0.0f);
        const Light& light = s.lights[light_num];
        V2 u_light = sampler.get_2d(), u_scattering = sampler.get_2d();
        Spec est = estimate_direct(hit, bsdf, u_scattering, light, (int)light_num, u_light);
        return est / light_pdf;
    }
    Spec estimate_direct(const SurfaceHit& hit, const BSDF& bsdf, V2 u_scattering, const Light& light, int light_num, V2 u_light) {
        const Scene& s = *sc;
        Spec ld(0.0f);
        Float scattering_pdf = 0.0f;
        LiSample ls = light_sample_li(light, hit, u_light);
        V3 wi = ls.valid ? ls.wi : V3(); Float light_pdf = ls.valid ? ls.pdf : 0.0f; Spec li = ls.valid ? ls.value : Spec(0.0f);
        bool is_delta = light.type == L_DISTANT || light.type == L_POINT || light.type == L_SPOT || light.type == L_PROJECTION || light.type == L_GONIO;
        if (light_pdf > 0.0f && !li.is_black()) {
            Spec f = bsdf.f(hit.wo, wi, BX_ALL & ~BX_SPEC) * abs_dot(wi, hit.ns);  // bsdf_flags: specular = false (common.rs:157-161)
            scattering_pdf = bsdf.pdf(hit.wo, wi, BX_ALL & ~BX_SPEC);
            if (!f.is_black()) {
                Ray sr = spawn_ray_to_hit(hit.p, hit.p_error, hit.n, hit.time, ls.vp, ls.vperr, ls.vn);
                if (scene_intersect_p(sr)) li = Spec(0.0f);
                if (!li.is_black()) {
                    if (is_delta) ld += f * li / light_pdf;
                    else { Float weight = power_heuristic(1, light_pdf, 1, scattering_pdf); ld += f * li * weight / light_pdf; }
                }
            }
        }
        if (!is_delta) {
            Spec f1; Float scatter_pdf; V3 wi2;
            int sampled_type = 0;
            bsdf.sample_f(hit.wo, u_scattering, f1, scatter_pdf, wi2, BX_ALL & ~BX_SPEC, &sampled_type);  // on failure: BxDFSample::default() = zeros
            scattering_pdf = scatter_pdf; wi = wi2;
            Spec f = f1 * abs_dot(wi, hit.ns);
            bool sampled_specular = (sampled_type & BX_SPEC) != 0;
            if (!f.is_black() && scattering_pdf > 0.0f) {
                Float weight = 1.0f;
                if (!sampled_specular) {
                    Float lp = light_pdf_li(light, hit, wi);
                    if (lp == 0.0f) return ld;
                    weight = power_heuristic(1, scattering_pdf, 1, lp);
                }
                Ray ray = spawn_ray(hit.p, hit.p_error, hit.n, hit.time, wi);
                uint32_t prim; TriHit h;
                Spec li2(0.0f);
                if (scene_intersect(ray, prim, h)) {
                    const Mesh& m = s.mesh_of(prim);
                    if (m.first_light >= 0 && (int)(m.first_light + (prim - m.tri_base)) == light_num) {
                        SurfaceHit lh = make_surface_hit(ray, prim, h);
                        li2 = area_L(light, lh.n, -wi);  // SurfaceInteraction::le (surface_interaction.rs:283-289)
                    }
                } else li2 = light_le(light, ray);
                if (!li2.is_black()) ld += f * li2 * Spec(1.0f) * weight / scattering_pdf;
            }
        }
        return ld;
    }

    // ---- WhittedIntegrator::li (integrators/src/whitted.rs:51-118) — ORACLE ONLY: every render the reference commits next to its example scenes was made with it, so
    // the oracle can be held against those pixels sample for sample.  One light sample per light and camera sample (`sampler.get_2d()` in the lights' order), no MIS,
    // then the specular recursion (whitted_specular below).
    int integrator = 0;  // 0 path, 1 whitted
    template <class S> Spec li_whitted(Ray ray, S& sampler, int depth) {
        const Scene& s = *sc;
        Spec L(0.0f);
        uint32_t prim; TriHit h;
        if (!scene_intersect(ray, prim, h)) {
            for (const Light& l : s.lights) L += light_le(l, ray);  // every light's le; zero for all but the infinite ones
            return L;
        }
        SurfaceHit isect = make_surface_hit(ray, prim, h);
        if (s.materials[s.mesh_of(isect.prim).material].none) {  // bsdf.is_none(): the ray goes on (whitted.rs:63-66)
            return li_whitted(spawn_ray(isect.p, isect.p_error, isect.n, isect.time, ray.d), sampler, depth);
        }
        compute_differentials(isect, ray);
        bump(isect);
        Lobe hit_lobes[8];
        BSDF bsdf = make_bsdf(isect, hit_lobes);
        // Whitted asks the materials for allow_multiple_lobes = false: smooth glass is then SpecularReflection(Kr, FresnelDielectric(1, eta)) + SpecularTransmission(Kt, 1, eta),
        // not the one FresnelSpecular lobe the path integrator gets (glass.rs:112-129)
        for (int i = 0; i < bsdf.n; i++) {
            if (bsdf.lobes[i].kind != LK_FRESNEL_SPEC) continue;
            Lobe split[8]; int k = 0;
            for (int j = 0; j < bsdf.n; j++) {
                if (j != i) { split[k++] = bsdf.lobes[j]; continue; }
                const Lobe& g = bsdf.lobes[j];
                if (!g.r.is_black()) { Lobe l; l.kind = LK_SPEC_R; l.type = BX_REFL | BX_SPEC; l.fresnel = FR_DIEL; l.eta_a = g.eta_a; l.eta_b = g.eta_b; l.r = g.r; split[k++] = l; }
                if (!g.t.is_black()) { Lobe l; l.kind = LK_SPEC_T; l.type = BX_TRANS | BX_SPEC; l.fresnel = FR_DIEL; l.eta_a = g.eta_a; l.eta_b = g.eta_b; l.t = g.t; split[k++] = l; }
            }
            for (int j = 0; j < k; j++) hit_lobes[j] = split[j];
            bsdf.lobes = hit_lobes; bsdf.n = k;
            break;
        }
        const V3 n = isect.ns, wo = isect.wo;
        const Mesh& m = s.mesh_of(prim);
        L += prim_le(m, prim, isect.n, wo);  // isect.le(&wo)
        for (const Light& light : s.lights) {
            const V2 u = sampler.get_2d();
            const LiSample ls = light_sample_li(light, isect, u);
            if (!ls.valid || ls.value.is_black() || ls.pdf == 0.0f) continue;
            const Spec f = bsdf.f(wo, ls.wi, BX_ALL);
            if (f.is_black()) continue;
            if (scene_intersect_p(spawn_ray_to_hit(isect.p, isect.p_error, isect.n, isect.time, ls.vp, ls.vperr, ls.vn))) continue;
            L += f * ls.value * abs_dot(ls.wi, n) / ls.pdf;
        }
        if (depth + 1 < max_depth) {  // whitted.rs:108-113
            const Spec refl = whitted_specular(ray, isect, bsdf, sampler, depth, false);
            const Spec trans = whitted_specular(ray, isect, bsdf, sampler, depth, true);
            L += refl + trans;
        }
        return L;
    }
    // SamplerIntegrator::specular_reflect / specular_transmit (core/src/integrator/sampler_integrator.rs:79-127, 137-238): the mirror / refracted ray carries
    // differentials derived from the surface's dndu / dndv, so that textures seen in a reflection are filtered as the reference filters them
    template <class S> Spec whitted_specular(const Ray& ray, const SurfaceHit& isect, const BSDF& bsdf, S& sampler, int depth, bool transmit) {
        const V3 wo = isect.wo;
        const V2 u = sampler.get_2d();
        Spec f; Float pdf = 0.0f; V3 wi; int st = 0;
        bsdf.sample_f(wo, u, f, pdf, wi, (transmit ? BX_TRANS : BX_REFL) | BX_SPEC, &st);
        V3 ns = isect.ns;
        if (!(pdf > 0.0f && !f.is_black() && abs_dot(wi, ns) != 0.0f)) return Spec(0.0f);
        Ray rd = spawn_ray(isect.p, isect.p_error, isect.n, isect.time, wi);
        if (ray.has_diff) {
            rd.has_diff = true;
            rd.rx_o = isect.p + isect.dpdx; rd.ry_o = isect.p + isect.dpdy;
            V3 dndx = isect.dndu_s * isect.dudx + isect.dndv_s * isect.dvdx;
            V3 dndy = isect.dndu_s * isect.dudy + isect.dndv_s * isect.dvdy;
            if (!transmit) {
                const V3 dwodx = -ray.rx_d - wo, dwody = -ray.ry_d - wo;
                const Float ddndx = dot(dwodx, ns) + dot(wo, dndx), ddndy = dot(dwody, ns) + dot(wo, dndy);
                rd.rx_d = wi - dwodx + 2.0f * (dot(wo, ns) * dndx + ddndx * ns);
                rd.ry_d = wi - dwody + 2.0f * (dot(wo, ns) * dndy + ddndy * ns);
            } else {
                Float eta = 1.0f / bsdf.eta;  // the BSDF's eta: 1 for glass in this reference (BSDF::new(.., None), glass.rs:109)
                if (dot(wo, ns) < 0.0f) { eta = 1.0f / eta; ns = -ns; dndx = -dndx; dndy = -dndy; }
                const V3 dwodx = -ray.rx_d - wo, dwody = -ray.ry_d - wo;
                const Float ddndx = dot(dwodx, ns) + dot(wo, dndx), ddndy = dot(dwody, ns) + dot(wo, dndy);
                const Float mu = eta * dot(wo, ns) - abs_dot(wi, ns);
                const Float dmudx = (eta - (eta * eta * dot(wo, ns)) / abs_dot(wi, ns)) * ddndx;
                const Float dmudy = (eta - (eta * eta * dot(wo, ns)) / abs_dot(wi, ns)) * ddndy;
                rd.rx_d = wi - eta * dwodx + (mu * dndx + dmudx * ns);
                rd.ry_d = wi - eta * dwody + (mu * dndy + dmudy * ns);
            }
        }
        return f * li_whitted(rd, sampler, depth + 1) * abs_dot(wi, ns) / pdf;
    }

    // ---- PathIntegrator::li (integrators/src/path.rs:103-284) ------------------------------------------------------
    template <class S> Spec li(Ray ray, S& sampler) {
        const Scene& s = *sc;
        Spec L(0.0f), beta(1.0f);
        bool specular_bounce = false;
        Float eta_scale = 1.0f;
        int bounces = 0;
        for (;;) {
            uint32_t prim; TriHit h;
            bool found = scene_intersect(ray, prim, h);
            SurfaceHit isect;
            if (found) isect = make_surface_hit(ray, prim, h);
            if (bounces == 0 || specular_bounce) {
                if (found) {
                    const Mesh& m = s.mesh_of(prim);
                    L += beta * prim_le(m, prim, isect.n, -ray.d);
                } else {
                    for (int li_ : s.infinite_lights) L += beta * light_le(s.lights[li_], ray);
                }
            }
            if (!found || bounces >= max_depth) break;
            if (s.materials[s.mesh_of(isect.prim).material].none) {  // null BSDF: the surface is skipped, bounces does not advance (path.rs:142-150)
                ray = spawn_ray(isect.p, isect.p_error, isect.n, isect.time, ray.d);
                continue;
            }
            compute_differentials(isect, ray);  // SurfaceInteraction::compute_scattering_functions (surface_interaction.rs:176-195)
            bump(isect);
            Lobe hit_lobes[8];
            BSDF bsdf = make_bsdf(isect, hit_lobes);
            if (bsdf.null) {   // `if isect.bsdf.is_none() { ray = spawn_ray; bounces -= 1; continue }` (path.rs:142-150), decided by this hit's textures
                ray = spawn_ray(isect.p, isect.p_error, isect.n, isect.time, ray.d);
                continue;
            }
            V3 shading_n = isect.ns;
            if (spatial) (void)spatial_lookup(isect.p);  // light_distribution.lookup(&isect.hit.p) happens for every vertex (path.rs:156-157)
            if (bsdf.num_components(BX_ALL & ~BX_SPEC) > 0) {
                tls_stats().total_paths++;
                Spec ld = beta * uniform_sample_one_light(isect, bsdf, sampler);
                if (ld.is_black()) tls_stats().zero_paths++;
                L += ld;
            }
            V2 u = sampler.get_2d();
            V3 wo = -ray.d, wi; Spec f; Float pdf; int flags = 0;
            bsdf.sample_f(wo, u, f, pdf, wi, BX_ALL, &flags);
            if (f.is_black() || pdf == 0.0f) break;
            beta *= f * abs_dot(wi, shading_n) / pdf;
            specular_bounce = (flags & BX_SPEC) != 0;
            if ((flags & BX_SPEC) && (flags & BX_TRANS)) {  // path.rs:192-203
                Float eta = bsdf.eta;
                eta_scale *= dot(wo, isect.n) > 0.0f ? eta * eta : 1.0f / (eta * eta);
            }
            ray = spawn_ray(isect.p, isect.p_error, isect.n, isect.time, wi);
            Spec rr_beta = beta * eta_scale;
            if (rr_beta.max_component_value() < rr_threshold && bounces > 3) {
                Float q = pmax(0.05f, 1.0f - rr_beta.max_component_value());
                if (sampler.get_1d() < q) break;
                beta = beta / (1.0f - q);
            }
            bounces++;
        }
        return L;
    }

    // ---- camera (perspective_camera.rs:144-204 + Transform::transform_ray transform.rs:451-476) ---------------------
    // OrthographicCamera::generate_ray_differential (orthographic_camera.rs:121-178): parallel rays from the film point; with a lens both offset
    // rays aim at their own focus points, whose distance `ft` uses the main ray's direction AFTER the lens moved it (:157)
    Ray generate_ray_orthographic(V2 p_film, Float time_s, V2 p_lens_s) const {
        V3 p_camera = cam.raster_to_camera.point(V3(p_film.x, p_film.y, 0.0f));
        Ray ray(p_camera, V3(0, 0, 1), INF, lerp(time_s, cam.shutter_open, cam.shutter_close));
        V3 rx_o, ry_o, rx_d, ry_d;
        if (cam.lens_radius > 0.0f) {
            V2 cd = concentric_sample_disk(p_lens_s);
            V2 p_lens(cam.lens_radius * cd.x, cam.lens_radius * cd.y);
            Float ft = cam.focal_distance / ray.d.z;
            V3 p_focus = ray.o + ray.d * ft;
            ray.o = V3(p_lens.x, p_lens.y, 0.0f);
            ray.d = normalize(p_focus - ray.o);
            ft = cam.focal_distance / ray.d.z;
            p_focus = (p_camera + cam.dx_camera) + (ft * V3(0, 0, 1));
            rx_o = V3(p_lens.x, p_lens.y, 0.0f); rx_d = normalize(p_focus - rx_o);
            p_focus = (p_camera + cam.dy_camera) + (ft * V3(0, 0, 1));
            ry_o = V3(p_lens.x, p_lens.y, 0.0f); ry_d = normalize(p_focus - ry_o);
        } else {
            rx_o = ray.o + cam.dx_camera; ry_o = ray.o + cam.dy_camera;
            rx_d = ray.d; ry_d = ray.d;
        }
        V3 o_err; V3 o = cam.camera_to_world.point_with_error(ray.o, o_err);
        V3 d = cam.camera_to_world.vector(ray.d);
        Float l2 = length_squared(d), t_max = ray.t_max;
        if (l2 > 0.0f) { Float dt = dot(vabs(d), o_err) / l2; o = o + d * dt; t_max -= dt; }  // quirk B2
        Ray out(o, d, t_max, ray.time);
        out.has_diff = true;
        out.rx_o = cam.camera_to_world.point(rx_o); out.ry_o = cam.camera_to_world.point(ry_o);
        out.rx_d = cam.camera_to_world.vector(rx_d); out.ry_d = cam.camera_to_world.vector(ry_d);
        return out;
    }
    // EnvironmentCamera::generate_ray (environment_camera.rs:61-78): the whole sphere of directions, y up in camera space
    Ray generate_ray_environment_main(V2 p_film, Float time_s) const {
        Float theta = PI * p_film.y / cam.full_res[1];
        Float phi = TWO_PI * p_film.x / cam.full_res[0];
        V3 dir(o_sin(theta) * o_cos(phi), o_cos(theta), o_sin(theta) * o_sin(phi));
        Ray ray(V3(0, 0, 0), dir, INF, lerp(time_s, cam.shutter_open, cam.shutter_close));
        V3 o_err; V3 o = cam.camera_to_world.point_with_error(ray.o, o_err);
        V3 d = cam.camera_to_world.vector(ray.d);
        Float l2 = length_squared(d), t_max = ray.t_max;
        if (l2 > 0.0f) { Float dt = dot(vabs(d), o_err) / l2; o = o + d * dt; t_max -= dt; }  // quirk B2
        return Ray(o, d, t_max, ray.time);
    }
    // Camera::generate_ray_differential's default (core/src/camera.rs:29-78): finite differences over a 0.05-pixel shift; every weight is 1, so
    // only the first eps of each loop is ever used
    Ray generate_ray_environment(V2 p_film, Float time_s) const {
        Ray ray = generate_ray_environment_main(p_film, time_s);
        const Float eps = 0.05f;
        Ray rx = generate_ray_environment_main(V2(p_film.x + eps, p_film.y), time_s);
        Ray ry = generate_ray_environment_main(V2(p_film.x, p_film.y + eps), time_s);
        ray.has_diff = true;
        ray.rx_o = ray.o + (rx.o - ray.o) / eps; ray.rx_d = ray.d + (rx.d - ray.d) / eps;
        ray.ry_o = ray.o + (ry.o - ray.o) / eps; ray.ry_d = ray.d + (ry.d - ray.d) / eps;
        return ray;
    }
    Ray generate_ray(V2 p_film, Float time_s, V2 p_lens_s) const {
        if (cam.kind == 1) return generate_ray_orthographic(p_film, time_s, p_lens_s);
        if (cam.kind == 2) return generate_ray_environment(p_film, time_s);
        V3 p_camera = cam.raster_to_camera.point(V3(p_film.x, p_film.y, 0.0f));
        Ray ray(V3(0, 0, 0), normalize(p_camera), INF, lerp(time_s, cam.shutter_open, cam.shutter_close));
        if (cam.lens_radius > 0.0f) {
            V2 cd = concentric_sample_disk(p_lens_s);
            V2 p_lens(cam.lens_radius * cd.x, cam.lens_radius * cd.y);
            Float ft = cam.focal_distance / ray.d.z;
            V3 p_focus = ray.o + ray.d * ft;
            ray.o = V3(p_lens.x, p_lens.y, 0.0f);
            ray.d = normalize(p_focus - ray.o);
        }
        // ray differentials (perspective_camera.rs:173-200), carried to world space by transform_ray (transform.rs:464-472)
        V3 rx_o, ry_o, rx_d, ry_d;
        if (cam.lens_radius > 0.0f) {
            V2 cd = concentric_sample_disk(p_lens_s);
            V2 p_lens(cam.lens_radius * cd.x, cam.lens_radius * cd.y);
            V3 dx = normalize(p_camera + cam.dx_camera);
            Float ft = cam.focal_distance / dx.z;
            V3 p_focus = V3(0, 0, 0) + (ft * dx);
            rx_o = V3(p_lens.x, p_lens.y, 0.0f); rx_d = normalize(p_focus - rx_o);
            V3 dy = normalize(p_camera + cam.dy_camera);
            ft = cam.focal_distance / dy.z;
            p_focus = V3(0, 0, 0) + (ft * dy);
            ry_o = V3(p_lens.x, p_lens.y, 0.0f); ry_d = normalize(p_focus - ry_o);
        } else {
            rx_o = ray.o; ry_o = ray.o;
            rx_d = normalize(p_camera + cam.dx_camera); ry_d = normalize(p_camera + cam.dy_camera);
        }
        V3 o_err; V3 o = cam.camera_to_world.point_with_error(ray.o, o_err);
        V3 d = cam.camera_to_world.vector(ray.d);
        Float l2 = length_squared(d), t_max = ray.t_max;
        if (l2 > 0.0f) { Float dt = dot(vabs(d), o_err) / l2; o = o + d * dt; t_max -= dt; }  // quirk B2
        Ray out(o, d, t_max, ray.time);
        out.has_diff = true;
        out.rx_o = cam.camera_to_world.point(rx_o); out.ry_o = cam.camera_to_world.point(ry_o);
        out.rx_d = cam.camera_to_world.vector(rx_d); out.ry_d = cam.camera_to_world.vector(ry_d);
        return out;
    }
    static void scale_differentials(Ray& r, Float sc) {  // ray.rs:90-99
        if (!r.has_diff) return;
        r.rx_o = r.o + (r.rx_o - r.o) * sc; r.ry_o = r.o + (r.ry_o - r.o) * sc;
        r.rx_d = r.d + (r.rx_d - r.d) * sc; r.ry_d = r.d + (r.ry_d - r.d) * sc;
    }
    // SurfaceInteraction::compute_differentials (surface_interaction.rs:203-278)
    static void compute_differentials(SurfaceHit& si, const Ray& ray) {
        si.dudx = si.dvdx = si.dudy = si.dvdy = 0.0f; si.dpdx = V3(0, 0, 0); si.dpdy = V3(0, 0, 0);
        if (!ray.has_diff) return;
        V3 n = si.n, p = si.p;
        Float d = dot(n, p);
        Float tx = -(dot(n, ray.rx_o) - d) / dot(n, ray.rx_d);
        if (std::isinf(tx) || tx != tx) return;
        V3 px = ray.rx_o + tx * ray.rx_d;
        Float ty = -(dot(n, ray.ry_o) - d) / dot(n, ray.ry_d);
        if (std::isinf(ty) || ty != ty) return;
        V3 py = ray.ry_o + ty * ray.ry_d;
        si.dpdx = px - p; si.dpdy = py - p;
        int dim[2];
        if (std::fabs(n.x) > std::fabs(n.y) && std::fabs(n.x) > std::fabs(n.z)) { dim[0] = 1; dim[1] = 2; }
        else if (std::fabs(n.y) > std::fabs(n.z)) { dim[0] = 0; dim[1] = 2; }
        else { dim[0] = 0; dim[1] = 1; }
        auto comp = [](V3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); };
        Float a[2][2] = {{comp(si.dpdu, dim[0]), comp(si.dpdv, dim[0])}, {comp(si.dpdu, dim[1]), comp(si.dpdv, dim[1])}};
        Float bx[2] = {comp(px, dim[0]) - comp(p, dim[0]), comp(px, dim[1]) - comp(p, dim[1])};
        Float by[2] = {comp(py, dim[0]) - comp(p, dim[0]), comp(py, dim[1]) - comp(p, dim[1])};
        auto solve = [&](const Float b[2], Float& x0, Float& x1) {  // matrix4x4.rs:305-318
            Float det = a[0][0] * a[1][1] - a[0][1] * a[1][0];
            if (std::fabs(det) < 1e-10f) return false;
            x0 = (a[1][1] * b[0] - a[0][1] * b[1]) / det;
            x1 = (a[0][0] * b[1] - a[1][0] * b[0]) / det;
            return !(x0 != x0 || x1 != x1);
        };
        Float u0, u1;
        if (solve(bx, u0, u1)) { si.dudx = u0; si.dvdx = u1; }
        if (solve(by, u0, u1)) { si.dudy = u0; si.dvdy = u1; }
    }

    // ---- Film ---------------------------------------------------------------------------------------------------
    struct FilmTile { int b[4]; std::vector<Float> contrib, wsum; };  // contrib rgb
    void sample_bounds(int out[4]) const {  // film/mod.rs:150-159
        out[0] = f2i32(std::floor((Float)film.crop[0] + 0.5f - film.radius[0]));
        out[1] = f2i32(std::floor((Float)film.crop[1] + 0.5f - film.radius[1]));
        out[2] = f2i32(std::ceil((Float)film.crop[2] - 0.5f + film.radius[0]));
        out[3] = f2i32(std::ceil((Float)film.crop[3] - 0.5f + film.radius[1]));
    }
    FilmTile get_film_tile(const int sb[4]) const {  // film/mod.rs:182-198
        FilmTile t;
        int p0x = f2i32(std::ceil((Float)sb[0] - 0.5f - film.radius[0])), p0y = f2i32(std::ceil((Float)sb[1] - 0.5f - film.radius[1]));
        int p1x = f2i32(std::floor((Float)sb[2] - 0.5f + film.radius[0])) + 1, p1y = f2i32(std::floor((Float)sb[3] - 0.5f + film.radius[1])) + 1;
        const Bounds2i tp = Bounds2i::raw(p0x, p0y, p1x, p1y).intersect(Bounds2i::raw(film.crop[0], film.crop[1], film.crop[2], film.crop[3]));  // Bounds2i { p0, p1 }.intersect(&cropped_pixel_bounds)
        t.b[0] = tp.x0; t.b[1] = tp.y0; t.b[2] = tp.x1; t.b[3] = tp.y1;
        int w = t.b[2] - t.b[0], hgt = t.b[3] - t.b[1];
        size_t n = (w > 0 && hgt > 0) ? (size_t)w * hgt : 0;
        t.contrib.assign(3 * n, 0.0f); t.wsum.assign(n, 0.0f);
        return t;
    }
    void add_sample(FilmTile& t, V2 p_film, Spec l, Float sample_weight) const {  // film_tile.rs:62-108
        Float ly = l.y();
        if (ly > film.max_lum) l = l * film.max_lum / ly;
        V2 pd(p_film.x - 0.5f, p_film.y - 0.5f);
        int p0x = f2i32(std::ceil(pd.x - film.radius[0])), p0y = f2i32(std::ceil(pd.y - film.radius[1]));
        int p1x = f2i32(std::floor(pd.x + film.radius[0])) + 1, p1y = f2i32(std::floor(pd.y + film.radius[1])) + 1;
        p0x = pmax(p0x, t.b[0]); p0y = pmax(p0y, t.b[1]); p1x = pmin(p1x, t.b[2]); p1y = pmin(p1y, t.b[3]);
        Float inv_rx = 1.0f / film.radius[0], inv_ry = 1.0f / film.radius[1];
        int w = t.b[2] - t.b[0];
        for (int y = p0y; y < p1y; y++) {
            Float fy = pabs(((Float)y - pd.y) * inv_ry * 16.0f);
            size_t iy = f2usize(pmin(std::floor(fy), 15.0f));
            for (int x = p0x; x < p1x; x++) {
                Float fx = pabs(((Float)x - pd.x) * inv_rx * 16.0f);
                size_t ix = f2usize(pmin(std::floor(fx), 15.0f));
                Float fw = film.table[iy * 16 + ix];
                size_t off = (size_t)(x - t.b[0]) + (size_t)(y - t.b[1]) * w;
                Spec c = l * sample_weight * fw;
                t.contrib[3 * off] += c.c[0]; t.contrib[3 * off + 1] += c.c[1]; t.contrib[3 * off + 2] += c.c[2];
                t.wsum[off] += fw;
            }
        }
    }

    // ---- render_tile (sampler_integrator.rs:312-415) -----------------------------------------------------------------
    template <class S> void render_tile_with(S& sampler, const int tb[4], FilmTile& ft) {
        const Bounds2i pxb = Bounds2i::raw(pixel_bounds[0], pixel_bounds[1], pixel_bounds[2], pixel_bounds[3]);
        Bounds2i::raw(tb[0], tb[1], tb[2], tb[3]).for_each([&](int x, int y) {  // `for pixel in tile_bounds`: row-major (bounds2.rs:312-359)
                sampler.start_pixel(x, y);
                if (!pxb.contains_exclusive(x, y)) return;  // sampler_integrator.rs:348-350
                for (;;) {
                    V2 fs = sampler.get_2d();
                    V2 p_film((Float)x + fs.x, (Float)y + fs.y);
                    Float time = sampler.get_1d();
                    V2 p_lens = sampler.get_2d();
                    Ray ray = generate_ray(p_film, time, p_lens);
                    scale_differentials(ray, 1.0f / std::sqrt((Float)sampler.spp));  // sampler_integrator.rs:358
                    tls_stats().camera_rays++;
                    Spec L = integrator == 1 ? li_whitted(ray, sampler, 0) : li(ray, sampler);  // ray_weight is always 1.0 for the perspective camera
                    if (L.has_nans()) L = Spec(0.0f);
                    else if (L.y() < -1e-5f) L = Spec(0.0f);
                    else if (std::isinf(L.y())) L = Spec(0.0f);
                    add_sample(ft, p_film, L, 1.0f);
                    if (!sampler.start_next_sample()) break;
                }
            });
    }

    void tile_bounds(int tile_idx, int ntx, const int sb[4], int tile_size, int tb[4]) const {
        int tx = tile_idx % ntx, ty = tile_idx / ntx;
        const int x0 = sb[0] + tx * tile_size, x1 = pmin(x0 + tile_size, sb[2]);
        const int y0 = sb[1] + ty * tile_size, y1 = pmin(y0 + tile_size, sb[3]);
        const Bounds2i b = Bounds2i::make(x0, y0, x1, y1);  // Bounds2i::new(Point2i::new(x0, y0), Point2i::new(x1, y1)) (sampler_integrator.rs:331-336)
        tb[0] = b.x0; tb[1] = b.y0; tb[2] = b.x1; tb[3] = b.y1;
    }

    // SamplerIntegrator::render (sampler_integrator.rs:243-304).  Tiles are merged in increasing tile index
    // (= the reference with --nthreads 1; with more threads the reference's merge order is completion order).
    void render(int tile_size, int tile_part, int tile_parts, int n_threads, Float* out_xyz, Float* out_weight, std::vector<FilmTile>* keep_tiles = nullptr) {
        // Integrator::preprocess: light distribution (path.rs:81-83, light_distrib/mod.rs:49-60)
        const Scene& s = *sc;
        std::vector<Float> lf;
        int strat = s.lights.size() == 1 ? 0 : light_strategy;
        for (const Light& l : s.lights) lf.push_back(strat == 1 ? light_power(l).y() : 1.0f);
        if (!lf.empty()) light_distrib.init(lf);
        spatial = strat == 2;
        if (spatial) spatial_init(64);  // create_light_sample_distribution (light_distrib/mod.rs:59-69)
        int sb[4]; sample_bounds(sb);
        int ntx = (sb[2] - sb[0] + tile_size - 1) / tile_size, nty = (sb[3] - sb[1] + tile_size - 1) / tile_size;
        int tile_count = ntx * nty;
        std::vector<FilmTile> tiles(tile_count);
        std::atomic<int> next{0};
        total_stats = RenderStats();
        auto worker = [&]() {
            tls_stats() = RenderStats();
            for (;;) {
                int t = next++;
                if (t >= tile_count) { std::lock_guard<std::mutex> g(stats_mu); total_stats.add(tls_stats()); break; }
                if (t % tile_parts != tile_part) continue;
                int tb[4]; tile_bounds(t, ntx, sb, tile_size, tb);
                tiles[t] = get_film_tile(tb);
                if (scfg.kind == 0) { HaltonSampler sp(scfg); render_tile_with(sp, tb, tiles[t]); }
                else if (scfg.kind == 2) { RandomSampler sp(scfg, (uint64_t)t); render_tile_with(sp, tb, tiles[t]); }
                else { SobolSampler sp(scfg); render_tile_with(sp, tb, tiles[t]); }
            }
        };
        std::vector<std::thread> th;
        for (int i = 1; i < n_threads; i++) th.emplace_back(worker);
        worker();
        for (auto& t : th) t.join();
        // Film::merge_film_tile (film/mod.rs:220-279)
        int cw = film.crop[2] - film.crop[0], ch = film.crop[3] - film.crop[1];
        size_t npx = (size_t)pmax(cw, 0) * pmax(ch, 0);
        for (size_t i = 0; i < 3 * npx; i++) out_xyz[i] = 0.0f;
        for (size_t i = 0; i < npx; i++) out_weight[i] = 0.0f;
        for (int t = 0; t < tile_count; t++) {
            if (t % tile_parts != tile_part) continue;
            const FilmTile& ft = tiles[t];
            int w = ft.b[2] - ft.b[0];
            for (int y = ft.b[1]; y < ft.b[3]; y++)
                for (int x = ft.b[0]; x < ft.b[2]; x++) {
                    size_t to = (size_t)(x - ft.b[0]) + (size_t)(y - ft.b[1]) * w;
                    size_t fo = (size_t)(x - film.crop[0]) + (size_t)(y - film.crop[1]) * cw;
                    Float xyz[3]; rgb_to_xyz(&ft.contrib[3 * to], xyz);
                    out_xyz[3 * fo] += xyz[0]; out_xyz[3 * fo + 1] += xyz[1]; out_xyz[3 * fo + 2] += xyz[2];
                    out_weight[fo] += ft.wsum[to];
                }
        }
        if (keep_tiles) *keep_tiles = std::move(tiles);
    }
    // Film::get_pixel_rgb (film/mod.rs:392-417), splat = 0
    void film_to_rgb(const Float* xyz, const Float* weight, Float* rgb) const {
        int cw = film.crop[2] - film.crop[0], ch = film.crop[3] - film.crop[1];
        size_t npx = (size_t)pmax(cw, 0) * pmax(ch, 0);
        for (size_t i = 0; i < npx; i++) {
            Float c[3]; xyz_to_rgb(&xyz[3 * i], c);
            Float zero[3] = {0, 0, 0}, srgb[3]; xyz_to_rgb(zero, srgb);
            for (int k = 0; k < 3; k++) {
                Float v = c[k];
                if (weight[i] != 0.0f) { Float inv = 1.0f / weight[i]; v = pmax(0.0f, v * inv); }
                v += 1.0f * srgb[0];  // quirk B3
                v *= film.scale;
                rgb[3 * i + k] = v;
            }
        }
    }
};

}  // namespace orc
