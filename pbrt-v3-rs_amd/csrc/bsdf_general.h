// General BSDF (core/src/reflection/bsdf.rs) over a material's lobe list, for scenes that use more than MatteMaterial.
//
// With constant textures compute_scattering_functions (materials/src/{matte,mirror,plastic,glass,metal,uber}.rs) produces the
// same BxDF list at every hit, so the host builds it once per material (api.hip) and the device only walks it:
//   LambertianReflection, OrenNayar, SpecularReflection, SpecularTransmission, FresnelSpecular (core/src/reflection/*.rs),
//   MicrofacetReflection / MicrofacetTransmission over TrowbridgeReitzDistribution with visible-area sampling
//   (core/src/microfacet/trowbridge_reitz.rs), Fresnel NoOp / Dielectric / Conductor (core/src/reflection/fresnel.rs).
// Expression order is the reference's; TransportMode is Radiance on this path.
#pragma once
#include "pt_device.h"

namespace ph {

enum : uint32_t { BX_REFL = 1u, BX_TRANS = 2u, BX_DIFF = 4u, BX_GLOSSY = 8u, BX_SPEC = 16u, BX_ALL = 31u };  // BxDFType (bsdf.rs:10-20)

// The BSDF of a hit = the material's TEMPLATE lobe list (built by the host, api.hip) seen through the hit: `keep` says which template lobes the reference would have added at this
// hit, and where the material takes colours / roughnesses / an index of refraction from textures, lobe_at() patches the template lobe on the fly from the texture pass's record of
// the hit (TexOut, 128 B per thread, read through L1).  Rounds 1 - 3 materialised a hit's list as up to eight 176-byte LobeRec copies in global memory per thread (1.4 KB slots,
// ~0.7 TB written per configs[4] frame and read back through L2); round 4 reads the template — the same for every lane of a material-sorted wave — and the record.
struct GBsdf {
    f3 ns, ng, ss, ts;
    const LobeRec* lobes;      // the material's template list
    uint32_t n;
    float eta;
    const TexOut* hit;         // null: constant material, the template IS the list
    const MaterialRec* mr;
    uint32_t keep;             // bit i: template lobe i is part of this hit's list
};
PH_DEV GBsdf make_gbsdf(const DeviceScene& sc, const SurfHit& si, uint32_t material) {
    const MaterialRec& m = sc.materials[material];
    GBsdf b;
    b.ns = si.ns; b.ng = si.n; b.ss = normalize(si.dpdu_s); b.ts = cross(b.ns, b.ss);  // bsdf.rs:100-116
    b.lobes = sc.lobes + m.lobe_base; b.n = m.n_lobes; b.eta = m.bsdf_eta;
    b.hit = nullptr; b.mr = &m; b.keep = 0xFFFFFFFFu;
    return b;
}
// ---- a textured material's lobe at a hit -------------------------------------------------------------------------------------------------------------------------
// The hit's own list = template lobes with the texture pass's colours filled in, a lobe dropped where the reference would not add it (`if !kd.is_black()`, plastic.rs:63 / :70,
// mirror.rs:55, matte.rs:66, uber.rs:134-160; FresnelBlend / FresnelSpecular unless both colours are black, substrate.rs:62, glass.rs:76-78).
PH_DEV bool lobe_keep(const LobeRec& l) {
    const bool r_black = l.r[0] == 0.0f && l.r[1] == 0.0f && l.r[2] == 0.0f, t_black = l.t[0] == 0.0f && l.t[1] == 0.0f && l.t[2] == 0.0f;
    // which colour decides whether the reference adds the lobe: both for the two-colour lobes, t for the transmission lobes, r otherwise
    return (l.kind == PH_LK_FRESNEL_BLEND || l.kind == PH_LK_FRESNEL_SPEC) ? !(r_black && t_black)
         : ((l.kind == PH_LK_SPEC_T || l.kind == PH_LK_MICRO_T || l.kind == PH_LK_LAMBERT_T) ? !t_black : !r_black);
}
// does this colour of the lobe come from the texture pass?  (r before t; PH_PRE_OPACITY / PH_PRE_PASSTHROUGH lobes always: their colour depends on the hit's opacity)
PH_DEV bool lobe_is_reflection(const LobeRec& l) { return l.kind == PH_LK_LAMBERT || l.kind == PH_LK_MICRO_R; }   // (of a PH_PRE_RT lobe: LambertianReflection / MicrofacetReflection against their transmission twins)
PH_DEV bool lobe_slot_r(const LobeRec& l) { return l.has_pre == PH_PRE_RT ? lobe_is_reflection(l) : (l.r_tex1 != 0u || (l.has_pre == PH_PRE_OPACITY && l.kind != PH_LK_SPEC_T)); }
PH_DEV bool lobe_slot_t(const LobeRec& l) { return l.has_pre == PH_PRE_RT ? !lobe_is_reflection(l) : (l.t_tex1 != 0u || (l.has_pre == PH_PRE_OPACITY && l.kind == PH_LK_SPEC_T) || l.has_pre == PH_PRE_PASSTHROUGH); }
// template lobe `l` of material `mr` as the hit `in` has it; returns whether a texel that decides the lobe's presence was black (PH_PRE_RAW_TEST / PH_PRE_RT lobes)
PH_DEV bool patch_lobe(const MaterialRec& mr, LobeRec& l, const TexOut* in) {
    // materials whose lobes need neither per-hit scalars nor the raw-black bits leave the record's header unwritten (texture_kernel): do not read it
    const uint32_t hdr_lambert = mr.tex_hdr ? in->lambert : 0u, hdr_bumped = mr.tex_hdr ? in->bumped : 0u;
    if (l.sigma_tex1) { l.kind = (hdr_lambert & 1u) ? PH_LK_LAMBERT : PH_LK_OREN; l.a = in->col[0][3]; l.b = in->col[1][3]; }
    if (l.ax_tex1 || l.ay_tex1) { l.ax = in->col[0][3]; l.ay = in->col[1][3]; }
    // the hit's index of refraction: FresnelSpecular / FresnelDielectric(1, eta) / the transmission lobes (glass.rs:113-139, uber.rs:145-176) — not the opacity pass-through, which is built with (1, 1)
    if (mr.index_tex1 && l.has_pre != PH_PRE_PASSTHROUGH && (l.kind == PH_LK_FRESNEL_SPEC || l.fresnel == PH_FR_DIEL)) l.eta_b = in->col[2][3];
    uint32_t ci = l.slot0;
    bool raw_black = false;
    if (lobe_slot_r(l) && ci < PH_HIT_COLS) { l.r[0] = in->col[ci][0]; l.r[1] = in->col[ci][1]; l.r[2] = in->col[ci][2]; raw_black = raw_black || ((hdr_bumped >> (8u + ci)) & 1u); ci++; }
    if (lobe_slot_t(l) && ci < PH_HIT_COLS) { l.t[0] = in->col[ci][0]; l.t[1] = in->col[ci][1]; l.t[2] = in->col[ci][2]; raw_black = raw_black || ((hdr_bumped >> (8u + ci)) & 1u); ci++; }
    if (l.eta_tex1 && ci < PH_HIT_COLS) { l.c_eta_t[0] = in->col[ci][0]; l.c_eta_t[1] = in->col[ci][1]; l.c_eta_t[2] = in->col[ci][2]; ci++; }
    if (l.k_tex1 && ci < PH_HIT_COLS) { l.c_k[0] = in->col[ci][0]; l.c_k[1] = in->col[ci][1]; l.c_k[2] = in->col[ci][2]; ci++; }
    if (l.amt & 3u) {   // MixMaterial with an `amount` texture: s1 = the hit's amount (the first colour slot), s2 = clamp(1 - s1) (mix.rs:59-60)
        float* sc = ((l.amt >> 8) & 3u) == 0u ? l.scale0 : l.scale1;
        const bool first = (l.amt & 3u) == 1u;
        for (int c = 0; c < 3; c++) { const float s1 = in->col[0][c]; sc[c] = first ? s1 : pclampf(1.0f - s1, 0.0f, kInf); }
    }
    return raw_black;
}
// which template lobes make up the hit's list (bit mask), and BSDF::eta of the hit when the material is an uber with an opacity texture (uber.rs:128-137; untouched otherwise)
PH_DEV uint32_t hit_lobe_mask(const MaterialRec& mr, const LobeRec* tmpl, uint32_t n, const TexOut* in, float& eta_out) {
    const bool is_specular = mr.tex_hdr && (in->lambert & 2u) != 0u;
    bool passthrough = false;
    uint32_t keep = 0u;
    for (uint32_t i = 0; i < n && i < PH_HIT_LOBES; i++) {
        LobeRec l = tmpl[i];
        const bool raw_black = patch_lobe(mr, l, in);
        if ((l.alt == 1u && !is_specular) || (l.alt == 2u && is_specular)) continue;   // glass.rs:112-141: FresnelSpecular, or the microfacet pair
        // TranslucentMaterial's lobes: the texel decided (an untextured one exists because its constant passed the test when the material was made)
        if ((l.has_pre == PH_PRE_RAW_TEST || l.has_pre == PH_PRE_RT) ? !raw_black : lobe_keep(l)) { if (l.has_pre == PH_PRE_PASSTHROUGH) passthrough = true; keep |= 1u << i; }
    }
    if (mr.uber_eta) eta_out = passthrough ? 1.0f : (mr.index_tex1 ? in->col[2][3] : mr.bsdf_eta_alt);
    return keep;
}
PH_DEV bool lobe_in(const GBsdf& b, uint32_t i) { return ((b.keep >> i) & 1u) != 0u; }
PH_DEV LobeRec lobe_at(const GBsdf& b, uint32_t i) {
    LobeRec l = b.lobes[i];
    if (b.hit) (void)patch_lobe(*b.mr, l, b.hit);
    return l;
}
PH_DEV f3 w2l(const GBsdf& b, f3 v) { return mk3(dot(v, b.ss), dot(v, b.ts), dot(v, b.ns)); }
PH_DEV f3 l2w(const GBsdf& b, f3 v) {
    return mk3(b.ss.x * v.x + b.ts.x * v.y + b.ns.x * v.z, b.ss.y * v.x + b.ts.y * v.y + b.ns.y * v.z, b.ss.z * v.x + b.ts.z * v.y + b.ns.z * v.z);
}

// reflection/common.rs
PH_DEV float g_cos2(f3 w) { return w.z * w.z; }
PH_DEV float g_tan(f3 w) { return ph_div(sin_theta(w), w.z); }
PH_DEV float g_tan2(f3 w) { return ph_div(sin2_theta(w), g_cos2(w)); }
PH_DEV float g_cos2phi(f3 w) { const float c = cos_phi(w); return c * c; }
PH_DEV float g_sin2phi(f3 w) { const float c = sin_phi(w); return c * c; }
PH_DEV bool g_same_hemi(f3 a, f3 b) { return a.z * b.z > 0.0f; }
PH_DEV f3 g_reflect(f3 wo, f3 n) { return -wo + 2.0f * dot(wo, n) * n; }
PH_DEV bool g_refract(f3 wi, f3 n, float eta, f3& wt) {  // common.rs:103-118
    const float cos_i = dot(n, wi);
    const float sin2_i = pmaxf(0.0f, 1.0f - cos_i * cos_i);
    const float sin2_t = eta * eta * sin2_i;
    if (sin2_t >= 1.0f) return false;
    const float cos_t = ph_sqrt(1.0f - sin2_t);
    wt = eta * -wi + (eta * cos_i - cos_t) * n;
    return true;
}
PH_DEV spec sp3(const float* v) { return mks(v[0], v[1], v[2]); }
PH_DEV spec spec_div(spec a, spec b) { return mks(ph_div(a.r, b.r), ph_div(a.g, b.g), ph_div(a.b, b.b)); }  // rgb_spectrum.rs:316-327
PH_DEV spec spec_sub(spec a, spec b) { return mks(a.r - b.r, a.g - b.g, a.b - b.b); }
PH_DEV spec spec_sqrt(spec a) { return mks(ph_sqrt(a.r), ph_sqrt(a.g), ph_sqrt(a.b)); }

// fresnel.rs:135-170
PH_DEV float fr_dielectric(float cos_i, float eta_i, float eta_t) {
    cos_i = pclampf(cos_i, -1.0f, 1.0f);
    if (!(cos_i > 0.0f)) { const float t = eta_i; eta_i = eta_t; eta_t = t; cos_i = pabs(cos_i); }
    const float sin_i = ph_sqrt(__builtin_fmaxf(0.0f, 1.0f - cos_i * cos_i));
    const float sin_t = ph_div(eta_i, eta_t) * sin_i;
    if (sin_t >= 1.0f) return 1.0f;
    const float cos_t = ph_sqrt(__builtin_fmaxf(0.0f, 1.0f - sin_t * sin_t));
    const float r_parl = ph_div((eta_t * cos_i) - (eta_i * cos_t), (eta_t * cos_i) + (eta_i * cos_t));
    const float r_perp = ph_div((eta_i * cos_i) - (eta_t * cos_t), (eta_i * cos_i) + (eta_t * cos_t));
    return ph_div(r_parl * r_parl + r_perp * r_perp, 2.0f);
}
// fresnel.rs:172-196 as written: `sin_theta_i_2 = 1.0 - cos_theta_i` (quirk B11); eta_i is Spectrum::ONE for every conductor made on
// this path (metal.rs:84-88), and x / 1.0 == x
PH_DEV spec fr_conductor(float cos_i, spec eta_t, spec k) {
    cos_i = pclampf(cos_i, -1.0f, 1.0f);
    const spec eta = eta_t, eta_k = k;
    const float cos2_i = cos_i * cos_i, sin2_i = 1.0f - cos_i;
    const spec eta_2 = eta * eta, eta_k_2 = eta_k * eta_k;
    const spec t0 = spec_sub(spec_sub(eta_2, eta_k_2), mks1(sin2_i));
    const spec a2_plus_b2 = spec_sqrt(t0 * t0 + 4.0f * eta_2 * eta_k_2);
    const spec t1 = a2_plus_b2 + mks1(cos2_i);
    const spec a = spec_sqrt(0.5f * (a2_plus_b2 + t0));
    const spec t2 = (2.0f * cos_i) * a;
    const spec rs = spec_div(spec_sub(t1, t2), t1 + t2);
    const spec t3 = cos2_i * a2_plus_b2 + mks1(sin2_i * sin2_i);
    const spec t4 = t2 * sin2_i;
    const spec rp = spec_div(rs * spec_sub(t3, t4), t3 + t4);
    return 0.5f * (rp + rs);
}
PH_DEV spec fresnel_eval(const LobeRec& l, float cos_i) {
    if (l.fresnel == PH_FR_DIEL) return mks1(fr_dielectric(cos_i, l.eta_a, l.eta_b));
    if (l.fresnel == PH_FR_COND) return fr_conductor(pabs(cos_i), sp3(l.c_eta_t), sp3(l.c_k));
    return mks1(1.0f);
}

// microfacet/trowbridge_reitz.rs + microfacet/mod.rs (sample_visible_area = true)
PH_DEV float tr_d(const LobeRec& l, f3 wh) {
    const float t2 = g_tan2(wh);
    if (__builtin_isinf(t2)) return 0.0f;
    const float cos4 = g_cos2(wh) * g_cos2(wh);
    const float e = (ph_div(g_cos2phi(wh), l.ax * l.ax) + ph_div(g_sin2phi(wh), l.ay * l.ay)) * t2;
    return ph_div(1.0f, kPi * l.ax * l.ay * cos4 * (1.0f + e) * (1.0f + e));
}
PH_DEV float tr_lambda(const LobeRec& l, f3 w) {
    const float att = pabs(g_tan(w));
    if (__builtin_isinf(att)) return 0.0f;
    const float alpha = ph_sqrt(g_cos2phi(w) * l.ax * l.ax + g_sin2phi(w) * l.ay * l.ay);
    const float a2t2 = (alpha * att) * (alpha * att);
    return ph_div(-1.0f + ph_sqrt(1.0f + a2t2), 2.0f);
}
PH_DEV float tr_g1(const LobeRec& l, f3 w) { return ph_div(1.0f, 1.0f + tr_lambda(l, w)); }
PH_DEV float tr_g(const LobeRec& l, f3 wo, f3 wi) { return ph_div(1.0f, 1.0f + tr_lambda(l, wo) + tr_lambda(l, wi)); }
PH_DEV float tr_pdf(const LobeRec& l, f3 wo, f3 wh) { return ph_div(tr_d(l, wh) * tr_g1(l, wo) * abs_dot(wo, wh), pabs(wo.z)); }
PH_DEV void tr_sample_11(float cos_theta, float u1, float u2, float& slope_x, float& slope_y) {
    if (cos_theta > 0.9999f) {
        const float r = ph_sqrt(ph_div(u1, 1.0f - u1));
        const float phi = kTwoPi * u2;
        float sn, cs; d_sincos(phi, sn, cs);
        slope_x = r * cs; slope_y = r * sn;
        return;
    }
    const float sin_theta_ = ph_sqrt(pmaxf(0.0f, 1.0f - cos_theta * cos_theta));
    const float tan_theta_ = ph_div(sin_theta_, cos_theta);
    float a = ph_div(1.0f, tan_theta_);
    const float g1 = ph_div(2.0f, 1.0f + ph_sqrt(1.0f + ph_div(1.0f, a * a)));
    a = ph_div(2.0f * u1, g1) - 1.0f;
    float tmp = ph_div(1.0f, a * a - 1.0f);
    if (tmp > 1e10f) tmp = 1e10f;
    const float b = tan_theta_;
    const float d = ph_sqrt(pmaxf(b * b * tmp * tmp - (a * a - b * b) * tmp, 0.0f));
    const float sx1 = b * tmp - d, sx2 = b * tmp + d;
    slope_x = (a < 0.0f || sx2 > ph_div(1.0f, tan_theta_)) ? sx1 : sx2;
    float sgn;
    if (u2 > 0.5f) { sgn = 1.0f; u2 = 2.0f * (u2 - 0.5f); } else { sgn = -1.0f; u2 = 2.0f * (0.5f - u2); }
    const float z = ph_div(u2 * (u2 * (u2 * 0.27385f - 0.73369f) + 0.46341f), u2 * (u2 * (u2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
    slope_y = sgn * z * ph_sqrt(1.0f + slope_x * slope_x);
}
PH_DEV f3 tr_sample_wh(const LobeRec& l, f3 wo, f2 u) {
    const bool flip = wo.z < 0.0f;
    const f3 wi = flip ? -wo : wo;
    const f3 ws = normalize(mk3(l.ax * wi.x, l.ay * wi.y, wi.z));
    float sx, sy;
    tr_sample_11(ws.z, u.x, u.y, sx, sy);
    const float tmp = cos_phi(ws) * sx - sin_phi(ws) * sy;
    sy = sin_phi(ws) * sx + cos_phi(ws) * sy;
    sx = tmp;
    sx *= l.ax; sy *= l.ay;
    const f3 wh = normalize(mk3(-sx, -sy, 1.0f));
    return flip ? -wh : wh;
}

// ---- per-lobe f / pdf / sample_f ------------------------------------------------------------------------------------------------
PH_DEV float pow5(float v) { return (v * v) * (v * v) * v; }  // pbrt/common.rs:345-347
PH_DEV spec lobe_f_raw(const LobeRec& l, f3 wo, f3 wi);
PH_DEV spec apply_scales(const LobeRec& l, spec f) {  // ScaledBxDF (scaled_bxdf.rs:27-35), innermost wrapper first
    if (l.n_scale > 0u) f = sp3(l.scale0) * f;
    if (l.n_scale > 1u) f = sp3(l.scale1) * f;
    return f;
}
PH_DEV spec lobe_f(const LobeRec& l, f3 wo, f3 wi) { return apply_scales(l, lobe_f_raw(l, wo, wi)); }
PH_DEV spec lobe_f_raw(const LobeRec& l, f3 wo, f3 wi) {
    switch (l.kind) {
    case PH_LK_LAMBERT_T: return sp3(l.t) * kInvPi;  // lambertian_transmission.rs:24-26
    case PH_LK_FRESNEL_BLEND: {  // fresnel_blend.rs:32-50 (Rd = l.r, Rs = l.t)
        const spec rd = sp3(l.r), rs = sp3(l.t);
        const spec diffuse = ph_div(28.0f, 23.0f * kPi) * rd * spec_sub(mks1(1.0f), rs) * (1.0f - pow5(1.0f - 0.5f * pabs(wi.z))) * (1.0f - pow5(1.0f - 0.5f * pabs(wo.z)));
        f3 wh = wi + wo;
        if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return mks1(0.0f);
        wh = normalize(wh);
        const spec schlick = rs + spec_sub(mks1(1.0f), rs) * pow5(1.0f - dot(wi, wh));
        const spec specular = ph_div(tr_d(l, wh), 4.0f * abs_dot(wi, wh) * pmaxf(pabs(wi.z), pabs(wo.z))) * schlick;
        return diffuse + specular;
    }
    case PH_LK_LAMBERT: return sp3(l.r) * kInvPi;  // lambertian_reflection.rs:38-40
    case PH_LK_OREN: {  // oren_nayar.rs:46-72
        const float sin_i = sin_theta(wi), sin_o = sin_theta(wo);
        float max_cos = 0.0f;
        if (sin_i > 1e-4f && sin_o > 1e-4f) {
            const float d_cos = cos_phi(wi) * cos_phi(wo) + sin_phi(wi) * sin_phi(wo);
            max_cos = pmaxf(0.0f, d_cos);
        }
        const float aco = pabs(wo.z), aci = pabs(wi.z);
        float sin_alpha, tan_beta;
        if (aci > aco) { sin_alpha = sin_o; tan_beta = ph_div(sin_i, aci); }
        else { sin_alpha = sin_i; tan_beta = ph_div(sin_o, aco); }
        return sp3(l.r) * kInvPi * (l.a + l.b * max_cos * sin_alpha * tan_beta);
    }
    case PH_LK_MICRO_R: {  // microfacet_reflection.rs:34-52
        const float cos_o = pabs(wo.z), cos_i = pabs(wi.z);
        f3 wh = wi + wo;
        if ((cos_i == 0.0f || cos_o == 0.0f) || (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f)) return mks1(0.0f);
        wh = normalize(wh);
        const spec f = fresnel_eval(l, dot(wi, face_forward(wh, mk3(0.0f, 0.0f, 1.0f))));
        return sp3(l.r) * tr_d(l, wh) * tr_g(l, wo, wi) * f / (4.0f * cos_i * cos_o);
    }
    case PH_LK_MICRO_T: {  // microfacet_transmission.rs:46-95
        if (g_same_hemi(wo, wi)) return mks1(0.0f);
        const float cos_o = wo.z, cos_i = wi.z;
        if (cos_i == 0.0f || cos_o == 0.0f) return mks1(0.0f);
        const float eta = wo.z > 0.0f ? ph_div(l.eta_b, l.eta_a) : ph_div(l.eta_a, l.eta_b);
        f3 wh = normalize(wo + wi * eta);
        if (wh.z < 0.0f) wh = -wh;
        if (dot(wo, wh) * dot(wi, wh) > 0.0f) return mks1(0.0f);
        const spec f = mks1(fr_dielectric(dot(wo, wh), l.eta_a, l.eta_b));
        const float sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
        const float factor = ph_div(1.0f, eta);
        return spec_sub(mks1(1.0f), f) * sp3(l.t) *
               pabs(ph_div(tr_d(l, wh) * tr_g(l, wo, wi) * eta * eta * abs_dot(wi, wh) * abs_dot(wo, wh) * factor * factor, cos_i * cos_o * sqrt_denom * sqrt_denom));
    }
    default: return mks1(0.0f);
    }
}
PH_DEV float lobe_pdf(const LobeRec& l, f3 wo, f3 wi) {
    switch (l.kind) {
    case PH_LK_LAMBERT: case PH_LK_OREN: return g_same_hemi(wo, wi) ? pabs(wi.z) * kInvPi : 0.0f;  // reflection/mod.rs:160-166
    case PH_LK_LAMBERT_T: return !g_same_hemi(wo, wi) ? pabs(wi.z) * kInvPi : 0.0f;            // lambertian_transmission.rs:36-42
    case PH_LK_FRESNEL_BLEND: {  // fresnel_blend.rs:77-85
        if (!g_same_hemi(wo, wi)) return 0.0f;
        const f3 wh = normalize(wo + wi);
        const float pdf_wh = tr_pdf(l, wo, wh);
        return 0.5f * (pabs(wi.z) * kInvPi + ph_div(pdf_wh, 4.0f * dot(wo, wh)));
    }
    case PH_LK_MICRO_R: {
        if (!g_same_hemi(wo, wi)) return 0.0f;
        const f3 wh = normalize(wo + wi);
        return ph_div(tr_pdf(l, wo, wh), 4.0f * dot(wo, wh));
    }
    case PH_LK_MICRO_T: {
        if (g_same_hemi(wo, wi)) return 0.0f;
        const float eta = wo.z > 0.0f ? ph_div(l.eta_b, l.eta_a) : ph_div(l.eta_a, l.eta_b);
        const f3 wh = normalize(wo + wi * eta);
        if (dot(wo, wh) * dot(wi, wh) > 0.0f) return 0.0f;
        const float sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
        const float dwh_dwi = pabs(ph_div(eta * eta * dot(wi, wh), sqrt_denom * sqrt_denom));
        return tr_pdf(l, wo, wh) * dwh_dwi;
    }
    default: return 0.0f;
    }
}
// returns the sampled BxDFType; f / pdf / wi are zero where the reference returns BxDFSample::from(type)
PH_DEV uint32_t lobe_sample_f_raw(const LobeRec& l, f3 wo, f2 u, spec& f, float& pdf, f3& wi);
PH_DEV uint32_t lobe_sample_f(const LobeRec& l, f3 wo, f2 u, spec& f, float& pdf, f3& wi) {
    const uint32_t st = lobe_sample_f_raw(l, wo, u, f, pdf, wi);
    f = apply_scales(l, f);
    return st;
}
PH_DEV uint32_t lobe_sample_f_raw(const LobeRec& l, f3 wo, f2 u, spec& f, float& pdf, f3& wi) {
    f = mks1(0.0f); pdf = 0.0f; wi = mk3(0.0f, 0.0f, 0.0f);
    switch (l.kind) {
    case PH_LK_LAMBERT_T: {  // lambertian_transmission.rs:28-35
        wi = cosine_sample_hemisphere(u);
        if (wo.z > 0.0f) wi.z *= -1.0f;
        pdf = lobe_pdf(l, wo, wi); f = lobe_f_raw(l, wo, wi);
        return l.type;
    }
    case PH_LK_FRESNEL_BLEND: {  // fresnel_blend.rs:52-75
        if (u.x < 0.5f) {
            u.x = pminf(2.0f * u.x, kOneMinusEps);
            wi = cosine_sample_hemisphere(u);
            if (wo.z < 0.0f) wi.z *= -1.0f;
        } else {
            u.x = pminf(2.0f * (u.x - 0.5f), kOneMinusEps);
            const f3 wh = tr_sample_wh(l, wo, u);
            wi = g_reflect(wo, wh);
            if (!g_same_hemi(wo, wi)) return l.type;
        }
        pdf = lobe_pdf(l, wo, wi); f = lobe_f_raw(l, wo, wi);
        return l.type;
    }
    case PH_LK_LAMBERT: case PH_LK_OREN: {  // reflection/mod.rs:132-141
        wi = cosine_sample_hemisphere(u);
        if (wo.z < 0.0f) wi.z *= -1.0f;
        pdf = lobe_pdf(l, wo, wi); f = lobe_f_raw(l, wo, wi);
        return l.type;
    }
    case PH_LK_SPEC_R: {  // specular_reflection.rs:38-44
        wi = mk3(-wo.x, -wo.y, wo.z); pdf = 1.0f;
        f = fresnel_eval(l, wi.z) * sp3(l.r) / pabs(wi.z);
        return l.type;
    }
    case PH_LK_SPEC_T: {  // specular_transmission.rs:45-64
        const bool entering = wo.z > 0.0f;
        const float eta_i = entering ? l.eta_a : l.eta_b, eta_t = entering ? l.eta_b : l.eta_a;
        f3 wt;
        if (!g_refract(wo, face_forward(mk3(0.0f, 0.0f, 1.0f), wo), ph_div(eta_i, eta_t), wt)) return l.type;
        wi = wt; pdf = 1.0f;
        spec ft = sp3(l.t) * spec_sub(mks1(1.0f), mks1(fr_dielectric(wi.z, l.eta_a, l.eta_b)));
        ft = ft * ph_div(eta_i * eta_i, eta_t * eta_t);
        f = ft / pabs(wi.z);
        return l.type;
    }
    case PH_LK_FRESNEL_SPEC: {  // fresnel_specular.rs:42-79
        const float fr = fr_dielectric(wo.z, l.eta_a, l.eta_b);
        if (u.x < fr) {
            wi = mk3(-wo.x, -wo.y, wo.z); pdf = fr;
            f = fr * sp3(l.r) / pabs(wi.z);
            return BX_SPEC | BX_REFL;
        }
        const bool entering = wo.z > 0.0f;
        const float eta_i = entering ? l.eta_a : l.eta_b, eta_t = entering ? l.eta_b : l.eta_a;
        f3 wt;
        if (!g_refract(wo, face_forward(mk3(0.0f, 0.0f, 1.0f), wo), ph_div(eta_i, eta_t), wt)) return BX_SPEC | BX_TRANS;
        wi = wt;
        spec ft = sp3(l.t) * (1.0f - fr);
        ft = ft * ph_div(eta_i * eta_i, eta_t * eta_t);
        pdf = 1.0f - fr;
        f = ft / pabs(wi.z);
        return BX_SPEC | BX_TRANS;
    }
    case PH_LK_MICRO_R: {  // microfacet_reflection.rs:54-76
        if (wo.z == 0.0f) return l.type;
        const f3 wh = tr_sample_wh(l, wo, u);
        if (dot(wo, wh) < 0.0f) return l.type;
        wi = g_reflect(wo, wh);
        if (!g_same_hemi(wo, wi)) return l.type;
        pdf = ph_div(tr_pdf(l, wo, wh), 4.0f * dot(wo, wh));
        f = lobe_f_raw(l, wo, wi);
        return l.type;
    }
    case PH_LK_MICRO_T: {  // microfacet_transmission.rs:97-120
        if (wo.z == 0.0f) return l.type;
        const f3 wh = tr_sample_wh(l, wo, u);
        if (dot(wo, wh) < 0.0f) return l.type;
        const float eta = wo.z > 0.0f ? ph_div(l.eta_a, l.eta_b) : ph_div(l.eta_b, l.eta_a);
        f3 wt;
        if (!g_refract(wo, wh, eta, wt)) return l.type;
        wi = wt;
        pdf = lobe_pdf(l, wo, wi); f = lobe_f_raw(l, wo, wi);
        return l.type;
    }
    default: return l.type;
    }
}

PH_DEV bool lobe_matches(const LobeRec& l, uint32_t flags) { return (l.type & flags) == l.type; }
// (a lobe's BxDFType is never patched: membership and type tests read the template)
PH_DEV bool lobe_sel(const GBsdf& b, uint32_t i, uint32_t flags) { return lobe_in(b, i) && lobe_matches(b.lobes[i], flags); }
PH_DEV uint32_t bsdf_num_components(const GBsdf& b, uint32_t flags) {
    uint32_t c = 0;
    for (uint32_t i = 0; i < b.n; i++) if (lobe_sel(b, i, flags)) c++;
    return c;
}
PH_DEV spec bsdf_f(const GBsdf& b, f3 wo_w, f3 wi_w, uint32_t flags) {  // bsdf.rs:133-158
    const f3 wi = w2l(b, wi_w), wo = w2l(b, wo_w);
    if (wo.z == 0.0f) return mks1(0.0f);
    const bool reflect = dot(wi_w, b.ng) * dot(wo_w, b.ng) > 0.0f;
    spec f = mks1(0.0f);
    for (uint32_t i = 0; i < b.n; i++) {
        const uint32_t ty = b.lobes[i].type;
        if (lobe_sel(b, i, flags) && ((reflect && (ty & BX_REFL)) || (!reflect && (ty & BX_TRANS)))) f = f + lobe_f(lobe_at(b, i), wo, wi);
    }
    return f;
}
PH_DEV float bsdf_pdf(const GBsdf& b, f3 wo_w, f3 wi_w, uint32_t flags) {  // bsdf.rs:331-356
    if ((b.keep & ((b.n >= 32u) ? 0xFFFFFFFFu : ((1u << b.n) - 1u))) == 0u) return 0.0f;   // `self.bxdfs.len() == 0`
    const f3 wo = w2l(b, wo_w), wi = w2l(b, wi_w);
    if (wo.z == 0.0f) return 0.0f;
    uint32_t matching = 0; float pdf = 0.0f;
    for (uint32_t i = 0; i < b.n; i++) if (lobe_sel(b, i, flags)) { matching++; pdf += lobe_pdf(lobe_at(b, i), wo, wi); }
    return matching > 0 ? ph_div(pdf, (float)matching) : 0.0f;
}
// BSDF::sample_f (bsdf.rs:160-292); a failed sample is BxDFSample::default(): zeros and type NONE
PH_DEV void bsdf_sample_f(const GBsdf& b, f3 wo_w, f2 u, uint32_t flags, spec& f_out, float& pdf_out, f3& wi_out, uint32_t& type_out) {
    f_out = mks1(0.0f); pdf_out = 0.0f; wi_out = mk3(0.0f, 0.0f, 0.0f); type_out = 0u;
    const uint32_t matching = bsdf_num_components(b, flags);
    if (matching == 0u) return;
    uint32_t comp = f2u_sat(floorf(u.x * (float)matching));
    if (comp > matching - 1u) comp = matching - 1u;
    uint32_t idx = 0, count = comp;
    for (uint32_t i = 0; i < b.n; i++) if (lobe_sel(b, i, flags)) { if (count == 0u) { idx = i; break; } count--; }
    const f2 ur = mk2(pminf(u.x * (float)matching - (float)comp, kOneMinusEps), u.y);
    const f3 wo = w2l(b, wo_w);
    if (wo.z == 0.0f) return;
    spec f; float pdf; f3 wi;
    const uint32_t st = lobe_sample_f(lobe_at(b, idx), wo, ur, f, pdf, wi);
    if (pdf == 0.0f) return;
    const f3 wi_w = l2w(b, wi);
    if (!(st & BX_SPEC) && matching > 1u)
        for (uint32_t i = 0; i < b.n; i++) if (i != idx && lobe_sel(b, i, flags)) pdf += lobe_pdf(lobe_at(b, i), wo, wi);
    if (matching > 1u) pdf = ph_div(pdf, (float)matching);
    if (!(st & BX_SPEC)) {
        const bool reflect = dot(wi_w, b.ng) * dot(wo_w, b.ng) > 0.0f;
        f = mks1(0.0f);
        for (uint32_t i = 0; i < b.n; i++) {
            const uint32_t ty = b.lobes[i].type;
            if (lobe_sel(b, i, flags) && ((reflect && (ty & BX_REFL)) || (!reflect && (ty & BX_TRANS)))) f = f + lobe_f(lobe_at(b, i), wo, wi);
        }
    }
    f_out = f; pdf_out = pdf; wi_out = wi_w; type_out = st;
}

}  // namespace ph
