// C ABI of include/pbrt_hip.h: scene capture, BVH build, device upload, batch traversal entry points.
// (The render entry points live in wavefront.hip.)  No CPU fallback exists anywhere in this library: without a
// gfx950 device pbrt_hip_scene_create returns NULL and every compute entry point fails with PBRT_HIP_ERR_NO_DEVICE.
#include "host_math.h"
#include "scene_host.h"
#include "traverse.h"
#include "texture.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>

static std::string g_last_error;
static std::mutex g_err_mu;

namespace phost {

int set_err(PbrtHipScene* s, int code, const std::string& msg) {
    if (s) s->err = msg;
    else { std::lock_guard<std::mutex> g(g_err_mu); g_last_error = msg; }
    return code;
}
int hip_fail(PbrtHipScene* s, hipError_t e, const char* what) {
    std::string m = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();
    return set_err(s, e == hipErrorOutOfMemory ? PBRT_HIP_ERR_OOM : PBRT_HIP_ERR_DEVICE, m);
}
int ensure_buf(PbrtHipScene* s, DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return PBRT_HIP_OK;
    if (b.p) { PH_CHECK(s, hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
    if (bytes == 0) bytes = 16;
    PH_CHECK(s, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    if (bytes <= 4096) PH_CHECK(s, hipMemset(b.p, 0, bytes));  // flags / counters start cleared
    return PBRT_HIP_OK;
}

template <class T> static int upload_vec(PbrtHipScene* s, const std::vector<T>& v, const T** out) {
    *out = nullptr;
    if (v.empty()) return PBRT_HIP_OK;
    void* d = nullptr;
    PH_CHECK(s, hipMalloc(&d, v.size() * sizeof(T)));
    s->owned.push_back(d);
    PH_CHECK(s, hipMemcpyAsync(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s->stream));
    *out = reinterpret_cast<const T*>(d);
    return PBRT_HIP_OK;
}

void free_tree_dev(PbrtHipScene* s) {
    if (s->tree_dev_nodes) (void)hipFree(s->tree_dev_nodes);
    if (s->tree_dev_tris) (void)hipFree(s->tree_dev_tris);
    s->tree_dev_nodes = s->tree_dev_tris = nullptr; s->tree_dev_n_tris = 0;
}

static void free_owned(PbrtHipScene* s) {
    for (void* p : s->owned) (void)hipFree(p);
    s->owned.clear();
    s->uploaded = false;
    s->light_strategy_uploaded = -1;
}

static hm::HaltonTables& halton_tables() {
    static hm::HaltonTables t;
    static std::once_flag once;
    std::call_once(once, []() { t.build(); });
    return t;
}

// Scenes with object instances, once per upload: every instance's leaf record receives what a ray entering it reads (scene_types.h, InstRec) and the hints that let the traversal kernel
// fetch the transform together with the record.  In place on the device arrays (device-built trees never visit the host); idempotent.
__global__ void patch_inst_records_kernel(TriRec* tris, uint32_t n_top, const InstRec* inst, float* extra) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_top) return;
    uint32_t f = tris[i].flags & ~PH_TRI_NEXT_INST;
    if (!(f & PH_TRI_LAST) && i + 1u < n_top && (tris[i + 1u].flags & PH_TRI_INSTANCE)) f |= PH_TRI_NEXT_INST;   // (this kernel never changes a PH_TRI_INSTANCE bit: no race with the neighbour's thread)
    if (f & PH_TRI_INSTANCE) {
        const InstRec& I = inst[tris[i].prim];
        const bool general = !(I.w2i[12] == 0.0f && I.w2i[13] == 0.0f && I.w2i[14] == 0.0f && I.w2i[15] == 1.0f);
        for (int k = 0; k < 3; k++) { tris[i].p0[k] = I.lo[k]; tris[i].p1[k] = I.hi[k]; }
        tris[i].p2[0] = __uint_as_float(I.root_ref); tris[i].p2[1] = __uint_as_float(I.flags | (general ? PH_INST_GENERAL : 0u)); tris[i].p2[2] = 0.0f;
        for (int k = 0; k < 12; k++) extra[12 * (size_t)i + k] = I.w2i[k];
    }
    tris[i].flags = f;
}
__global__ void patch_leaf_refs_kernel(Node64* nodes, uint32_t n_nodes, const TriRec* tris, uint32_t n_top) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    uint32_t c[2] = {nodes[i].c0, nodes[i].c1};
    for (int k = 0; k < 2; k++)
        if ((c[k] & PH_LEAF_BIT) && c[k] < PH_NEED_POP) {
            const uint32_t t = c[k] & ~(PH_LEAF_BIT | PH_LEAF_INST_HINT);
            c[k] = PH_LEAF_BIT | t | ((t < n_top && (tris[t].flags & PH_TRI_INSTANCE)) ? PH_LEAF_INST_HINT : 0u);
        }
    nodes[i].c0 = c[0]; nodes[i].c1 = c[1];
}

int upload_scene(PbrtHipScene* s) {
    if (s->uploaded) return PBRT_HIP_OK;
    free_owned(s);
    DeviceScene& d = s->ds;
    std::memset(&d, 0, sizeof(d));
    int rc;
    if (s->tree_dev_tris) { d.nodes = reinterpret_cast<const Node64*>(s->tree_dev_nodes); d.tris = reinterpret_cast<const TriRec*>(s->tree_dev_tris); }   // built here, never left
    else {
        if ((rc = upload_vec(s, s->bvh.nodes, &d.nodes))) return rc;
        if ((rc = upload_vec(s, s->bvh.tris, &d.tris))) return rc;
    }
    d.root_ref = s->bvh.root_ref;
    d.n_tris = (uint32_t)(s->idx.size() / 3);
    for (int k = 0; k < 3; k++) { d.root_lo[k] = s->bvh.root_lo[k]; d.root_hi[k] = s->bvh.root_hi[k]; d.world_center[k] = s->world_center[k]; }
    d.world_radius = s->world_radius;
    if ((rc = upload_vec(s, s->P, &d.P))) return rc;
    if (s->any_n && (rc = upload_vec(s, s->N, &d.N))) return rc;
    if (s->any_s && (rc = upload_vec(s, s->S, &d.S))) return rc;
    if (s->any_uv && (rc = upload_vec(s, s->UV, &d.UV))) return rc;
    if ((rc = upload_vec(s, s->idx, &d.idx))) return rc;
    if ((rc = upload_vec(s, s->tri_mesh, &d.tri_mesh))) return rc;
    if ((rc = upload_vec(s, s->tri_flags, &d.tri_flags))) return rc;
    if (s->emission_only.empty()) { if ((rc = upload_vec(s, s->meshes, &d.meshes))) return rc; }
    else {   // emission-only area lights sit behind the scene's lights on the device: their meshes point there
        std::vector<MeshRec> meshes(s->meshes);
        for (MeshRec& m : meshes) if (m.first_light <= -2) m.first_light = (int32_t)(s->lights.size() + (size_t)(-2 - m.first_light));
        if ((rc = upload_vec(s, meshes, &d.meshes))) return rc;
    }
    {   // per material: what the texture pass has to hand to the shade pass (texture.h: eval_lobe_colours / build_hit_lobes use the same slot rules)
        for (MaterialRec& m : s->materials) {
            uint32_t cols = m.amount_tex1 ? 1u : 0u; bool hdr = m.bump_tex1 != 0u || m.sigma_tex1 != 0u;
            for (uint32_t k = 0; k < m.n_lobes; k++) {
                s->lobes[m.lobe_base + k].slot0 = cols;   // where this lobe's colours start in TexOut (bsdf_general.h: patch_lobe reads them from there)
                const LobeRec& l = s->lobes[m.lobe_base + k];
                if (l.has_pre == PH_PRE_RT) { cols += 1u; hdr = true; }   // one product colour per lobe, its black bit and the material's null-BSDF bit in the header
                else cols += (l.r_tex1 || (l.has_pre == PH_PRE_OPACITY && l.kind != PH_LK_SPEC_T) ? 1u : 0u) + (l.t_tex1 || (l.has_pre == PH_PRE_OPACITY && l.kind == PH_LK_SPEC_T) || l.has_pre == PH_PRE_PASSTHROUGH ? 1u : 0u) +
                        (l.eta_tex1 ? 1u : 0u) + (l.k_tex1 ? 1u : 0u);
                if (l.sigma_tex1 || l.ax_tex1 || l.ay_tex1 || l.alt || l.has_pre == PH_PRE_RAW_TEST) hdr = true;
            }
            if (m.sigma_tex1 || hdr) cols = std::max(cols, 2u);   // the per-hit scalars travel in the fourth component of the first two colour slots
            if (m.index_tex1) cols = std::max(cols, 3u);          // ... the per-hit index of refraction in the third one's
            if (m.rt_mode) hdr = true;
            m.tex_cols = std::min<uint32_t>(cols, PH_HIT_COLS); m.tex_hdr = hdr ? 1u : 0u;
        }
        s->alpha_lean = s->alpha_textures;
        for (const MeshRec& m : s->meshes)
            for (uint32_t t1 : {m.alpha_tex1, m.shadow_alpha_tex1})
                if (t1)
                    for (const TexOp& op : s->textures[t1 - 1u].prog)
                        if (!(op.op == PH_TOP_CONST || op.op == PH_TOP_MUL || op.op == PH_TOP_MIX || (op.op == PH_TOP_IMAGE && op.mapping == 0u))) s->alpha_lean = false;
        s->simple_textures = true;
        for (const PbrtHipScene::TextureHost& t : s->textures)
            for (const TexOp& op : t.prog)
                if (!(op.op == PH_TOP_CONST || op.op == PH_TOP_MUL || op.op == PH_TOP_MIX || (op.op == PH_TOP_IMAGE && op.mapping == 0u))) s->simple_textures = false;
    }
    if ((rc = upload_vec(s, s->materials, &d.materials))) return rc;
    if ((rc = upload_vec(s, s->lobes, &d.lobes))) return rc;
    {   // keys of the shade-side work queues (matsort.h): materials the texture pass has work for first, then the others, then "emission only" and "no new vertex"
        const uint32_t kMaxTex = 500u, kMaxPlain = 500u;
        std::vector<uint16_t> key(std::max<size_t>(s->materials.size(), 1), 0);
        uint32_t n_tex = 0, n_plain = 0;
        for (const MaterialRec& m : s->materials) if (!m.none && (m.textured || m.bump_tex1)) n_tex++;
        const uint32_t T = std::min(n_tex, kMaxTex);
        uint32_t jt = 0, jp = 0;
        for (size_t i = 0; i < s->materials.size(); i++) {
            const MaterialRec& m = s->materials[i];
            if (!m.none && (m.textured || m.bump_tex1)) key[i] = (uint16_t)std::min(jt++, kMaxTex - 1u);
            else { key[i] = (uint16_t)(T + std::min(jp++, kMaxPlain - 1u)); n_plain++; }
        }
        const uint32_t U = std::min(n_plain, kMaxPlain);
        if ((rc = upload_vec(s, key, &d.mat_key))) return rc;
        d.ms_tex_keys = T; d.ms_key_emit = T + U; d.ms_key_idle = T + U + 1u;
    }
    if (!s->textures.empty() || !s->mipmaps.empty()) {   // MIPMaps also belong to lights (radiance map, projection image, goniometric diagram)
        std::vector<TexRec> recs; std::vector<TexOp> ops;
        for (const PbrtHipScene::TextureHost& t : s->textures) {
            recs.push_back(TexRec{(uint32_t)ops.size(), (uint32_t)t.prog.size()});
            ops.insert(ops.end(), t.prog.begin(), t.prog.end());
        }
        std::vector<float> lut(PH_EWA_LUT_SIZE);
        for (int i = 0; i < PH_EWA_LUT_SIZE; i++) {  // mipmap/mod.rs:168-174 (host expf, as in the reference)
            const float r2 = (float)i / (float)(PH_EWA_LUT_SIZE - 1);
            lut[(size_t)i] = std::exp(-2.0f * r2) - std::exp(-2.0f);
        }
        if ((rc = upload_vec(s, recs, &d.textures))) return rc;
        if ((rc = upload_vec(s, ops, &d.tex_ops))) return rc;
        if ((rc = upload_vec(s, s->mipmaps, &d.mipmaps))) return rc;
        if ((rc = upload_vec(s, s->texels, &d.texels))) return rc;
        if ((rc = upload_vec(s, lut, &d.ewa_lut))) return rc;
    }
    if (s->emission_only.empty()) { if ((rc = upload_vec(s, s->lights, &d.lights))) return rc; }
    else {
        std::vector<LightRec> all(s->lights);
        all.insert(all.end(), s->emission_only.begin(), s->emission_only.end());
        if ((rc = upload_vec(s, all, &d.lights))) return rc;
    }
    if ((rc = upload_vec(s, s->light_dist, &d.light_dist))) return rc;
    d.n_lights = (uint32_t)s->lights.size();
    if ((rc = upload_vec(s, s->infinite_lights, &d.infinite_lights))) return rc;
    d.n_infinite = (uint32_t)s->infinite_lights.size();
    if ((rc = upload_vec(s, s->inst_recs, &d.instances))) return rc;
    d.n_instances = (uint32_t)s->inst_recs.size();
    // a forest whose trees are all single leaves has no interior node at all (d.nodes == nullptr): the records still need their bounds / root / transform
    if (!s->inst_recs.empty() && !s->top_items.empty() && d.tris) {   // instance leaf records + hints (patch_inst_records_kernel)
        const uint32_t n_top = (uint32_t)s->top_items.size();
        const uint32_t n_nodes = (uint32_t)(s->tree_dev_tris ? s->bvh.interior_nodes : s->bvh.nodes.size());
        void* extra = nullptr;
        PH_CHECK(s, hipMalloc(&extra, (size_t)n_top * 48)); s->owned.push_back(extra);
        hipLaunchKernelGGL(patch_inst_records_kernel, dim3((n_top + 255u) / 256u), dim3(256), 0, s->stream, const_cast<TriRec*>(d.tris), n_top, d.instances, static_cast<float*>(extra));
        if (n_nodes) hipLaunchKernelGGL(patch_leaf_refs_kernel, dim3((n_nodes + 255u) / 256u), dim3(256), 0, s->stream, const_cast<Node64*>(d.nodes), n_nodes, d.tris, n_top);
        PH_CHECK(s, hipGetLastError());
        d.inst_extra = static_cast<const float*>(extra);
    }
    hm::HaltonTables& ht = halton_tables();
    if ((rc = upload_vec(s, ht.perms, &d.halton_perms))) return rc;
    if ((rc = upload_vec(s, ht.primes, &d.primes))) return rc;
    if ((rc = upload_vec(s, ht.prime_sums, &d.prime_sums))) return rc;
    if ((rc = upload_vec(s, ht.magic, &d.prime_magic))) return rc;
    if ((rc = upload_vec(s, s->sobol32, &d.sobol32))) return rc;
    if ((rc = upload_vec(s, s->vdc, &d.vdc))) return rc;
    if ((rc = upload_vec(s, s->vdc_inv, &d.vdc_inv))) return rc;
    {   // device-resident copy of the struct itself (DeviceScene::self), for out-of-line device functions
        void* dself = nullptr;
        PH_CHECK(s, hipMalloc(&dself, sizeof(DeviceScene)));
        s->owned.push_back(dself);
        d.self = reinterpret_cast<const DeviceScene*>(dself);
        PH_CHECK(s, hipMemcpyAsync(dself, &d, sizeof(DeviceScene), hipMemcpyHostToDevice, s->stream));
    }
    PH_CHECK(s, hipStreamSynchronize(s->stream));
    s->uploaded = true;
    return PBRT_HIP_OK;
}

// Light::power().y() per light -> Distribution1D (core/src/integrator/common.rs:304-311); uniform = all ones
// (core/src/light_distrib/uniform.rs:24-31).  create_light_sample_distribution forces Uniform for a single light.
int upload_light_distribution(PbrtHipScene* s, int light_strategy) {
    const size_t n = s->lights.size();
    int strat = (n == 1) ? 0 : light_strategy;
    if (s->light_strategy_uploaded == strat) return PBRT_HIP_OK;
    std::vector<float> f(n, 1.0f), cdf;
    float func_int = 0.0f;
    if (strat == 1) {
        const float wr = s->world_radius;
        for (size_t i = 0; i < n; i++) {
            const LightRec& l = s->lights[i];
            float p[3];
            for (int c = 0; c < 3; c++) {
                switch (l.type) {
                case PH_L_INFINITE: {  // infinite.rs:176-183: PI * r * r * lookup_triangle((.5,.5), .5)
                    if (l.map_mip1) { float t[3]; hmip_lookup_triangle(s, s->mipmaps[l.map_mip1 - 1u], 0.5f, 0.5f, 0.5f, t); p[c] = hm::kPi * wr * wr * t[c]; break; }
                    // 1x1 MIPMap::triangle at st=(0.5,0.5): s = 0, ds = 0 -> tx*1*1 + tx*1*0 + tx*0*1 + tx*0*0
                    float tx = l.L[c];
                    float v = tx * (1.0f - 0.0f) * (1.0f - 0.0f) + tx * (1.0f - 0.0f) * 0.0f + tx * 0.0f * (1.0f - 0.0f) + tx * 0.0f * 0.0f;
                    p[c] = hm::kPi * wr * wr * v;
                    break;
                }
                case PH_L_DISTANT: p[c] = l.L[c] * hm::kPi * wr * wr; break;                       // distant.rs:98-101
                case PH_L_POINT: p[c] = (hm::kPi * 4.0f) * l.L[c]; break;                          // point.rs:95-97
                case PH_L_PROJECTION: case PH_L_GONIO: {   // projection.rs:193-203: spectrum * I * TWO_PI * (1 - cos_total_width); goniometric.rs:115-126: FOUR_PI * I * spectrum
                    float t[3] = {1.0f, 1.0f, 1.0f};
                    if (l.map_mip1) hmip_lookup_triangle(s, s->mipmaps[l.map_mip1 - 1u], 0.5f, 0.5f, 0.5f, t);
                    p[c] = l.type == PH_L_PROJECTION ? t[c] * l.L[c] * (2.0f * hm::kPi) * (1.0f - l.cos_total_width) : (hm::kPi * 4.0f) * l.L[c] * t[c];
                    break;
                }
                case PH_L_SPOT: p[c] = l.L[c] * (2.0f * hm::kPi) * (1.0f - 0.5f * (l.cos_falloff_start + l.cos_total_width)); break;  // spot.rs:86-88
                default: p[c] = (l.two_sided ? 2.0f : 1.0f) * l.L[c] * l.area * hm::kPi; break;    // diffuse.rs:131-134
                }
            }
            f[i] = 0.212671f * p[0] + 0.715160f * p[1] + 0.072169f * p[2];
        }
    }
    if (n) hm::distribution1d(f, cdf, func_int);
    int rc;
    if ((rc = ensure_buf(s, s->d_ld_func, std::max<size_t>(n, 1) * 4))) return rc;
    if ((rc = ensure_buf(s, s->d_ld_cdf, (n + 1) * 4))) return rc;
    if (n) {
        PH_CHECK(s, hipMemcpy(s->d_ld_func.p, f.data(), n * 4, hipMemcpyHostToDevice));
        PH_CHECK(s, hipMemcpy(s->d_ld_cdf.p, cdf.data(), (n + 1) * 4, hipMemcpyHostToDevice));
    }
    s->ds.ld_func = (const float*)s->d_ld_func.p; s->ds.ld_cdf = (const float*)s->d_ld_cdf.p; s->ds.ld_func_int = func_int;
    s->light_strategy_uploaded = strat;
    return PBRT_HIP_OK;
}

// The traversal kernel's shape: {LEAF_MIN, REFILL_MIN, LDS_DEPTH, NODE_STEPS, waves per SIMD the kernel is compiled for (0 = the compiler's choice)}.  Slot 0 is what ships: 6 waves per
// SIMD with 12 stack entries in LDS (the kernel needs 59 VGPRs since the round-3 register diet and would fit 8 — but a seventh wave buys nothing and costs the stack an entry: same-box,
// configs[2] / configs[3] 689.5 / 717.0 ms of traversal per frame at 6 waves x 12 entries against 705.5 / 735.5 at 7 x 11; at 6 waves the depth is worth 12 -> 10 -> 8 -> 6 entries:
// 687 -> 692 -> 708 -> 759 ms, a 13th nothing), lanes wait for 16 companions at leaves, 6 node steps per pass (27 shapes swept at 7 waves, 6 more at 6; gpurun r03s - r03x, r03aw, r03ax).
// Round 4: since finished rays are written out at the wave's refill (traverse.h), the refill threshold is worth more — HALF the wave idle before a refill: configs[2] 711.9 / 689.8 /
// 671.2 / 657.6 / 652.0 / 670.9 / 703.1 / 844.8 ms at 12 / 16 / 20 / 28 / 32 / 36 / 40 / 48 idle lanes (leaf threshold 12 / 16 / 20 and 5 / 6 / 8 node steps per pass within 3 ms of
// each other at 32; configs[3] 704.2 -> 679.2, configs[1] 29.8 -> 29.4; gpurun r04ac - r04ae).  Slot 1 is the round-3 shape (refill at 20), kept for A/B runs (PBRT_HIP_TRAV_VARIANT=1).
#define PH_VARIANTS(X) X(0, 16, 32, 12, 6, 6, false) X(1, 16, 20, 12, 6, 6, false)
#define PH_N_VARIANTS 2
#define PH_DEFAULT_INST_VARIANT 2   // the 5-wave instancing kernel (launch_traverse_kernel); 1 = the 4-wave form, kept as the A/B slot
#define PH_DEFAULT_VARIANT 0
static int trav_variant() {
    static int v = -1;
    if (v < 0) { const char* e = std::getenv("PBRT_HIP_TRAV_VARIANT"); v = e ? std::atoi(e) : PH_DEFAULT_VARIANT; if (v < 0 || v >= PH_N_VARIANTS) v = PH_DEFAULT_VARIANT; }
    return v;
}
static int alpha_min() {   // PBRT_HIP_ALPHA_MIN: lanes that wait for an alpha-mask verdict before the wave evaluates the masks together (traverse.h, ALPHA_MIN); A/B aid for the instancing kernel: 0 (inside the leaf step, rounds 2 - 3), 4, 20; default 12
    static const int v = []() { const char* e = std::getenv("PBRT_HIP_ALPHA_MIN"); const int x = e ? std::atoi(e) : 12; return (x == 0 || x == 4 || x == 20) ? x : 12; }();
    return v;
}
static int variant_lds_depth(int v) {
    switch (v) {
#define X(id, lm, rm, ld, ns, wpe, pk) case id: return ld;
        PH_VARIANTS(X)
#undef X
    }
    return PH_LDS_DEPTH;
}

int ensure_traversal_workspace(PbrtHipScene* s) {
    if (!s->trav_blocks) {
        hipDeviceProp_t prop;
        PH_CHECK(s, hipGetDeviceProperties(&prop, s->device));
        int per_cu = 0;
        switch (trav_variant()) {
#define X(id, lm, rm, ld, ns, wpe, pk) case id: PH_CHECK(s, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ph::traverse_kernel<false, false, lm, rm, ld, ns, false, true, 0, wpe>, PH_TRAV_BLOCK, 0)); if (wpe) per_cu = std::min(per_cu, wpe); break;
            PH_VARIANTS(X)
#undef X
        }
        per_cu = std::min(std::max(per_cu, 1), 8);
        if (const char* e = std::getenv("PBRT_HIP_TRAV_BLOCKS_PER_CU")) per_cu = std::min(std::max(std::atoi(e), 1), per_cu);  // measurement aid
        s->trav_blocks = (uint32_t)(prop.multiProcessorCount * per_cu);
    }
    const uint32_t total_threads = s->trav_blocks * PH_TRAV_BLOCK;
    int rc;
    if ((rc = ensure_buf(s, s->d_counter, 64))) return rc;
    if ((rc = ensure_buf(s, s->d_error, 64))) return rc;
    if ((rc = ensure_buf(s, s->d_counts, 64 + 36 * 8))) return rc;
    const int stack_cap = s->inst_recs.empty() ? PH_MAX_STACK : 2 * PH_MAX_STACK;  // with instances the scene-level and object-level entries share one stack
    const int lds_depth = s->inst_recs.empty() ? variant_lds_depth(trav_variant()) : 11;   // (the shallowest LDS stack any instancing kernel is compiled with)
    if ((rc = ensure_buf(s, s->d_spill, (size_t)(stack_cap - lds_depth) * total_threads * sizeof(uint2)))) return rc;
    return PBRT_HIP_OK;
}

// p.spill / total_threads / error_flag / counts are filled here.  mode: 0 closest hit, 1 any hit, 2 both queues in one launch (MIXED)
void launch_traverse_kernel(PbrtHipScene* s, int mode, uint32_t blocks, const ph::TravParams& p_in) {
    ph::TravParams p = p_in;
    p.spill = (uint2*)s->d_spill.p; p.total_threads = s->trav_blocks * PH_TRAV_BLOCK; p.error_flag = (uint32_t*)s->d_error.p;
    p.counts = (unsigned long long*)s->d_counts.p;
#if PH_PHASE_CLOCK
    p.phase = (unsigned long long*)s->d_counts.p + 8;   // (the measurement build's phase tallies sit behind the eight counters)
#endif
    { static int bt = -1; if (bt < 0) { const char* e = std::getenv("PBRT_HIP_TRAV_BATCH"); bt = e ? std::atoi(e) : PH_BATCH; if (bt != 64 && bt != 128 && bt != 256 && bt != 512 && bt != 1024) bt = 64; } p.batch = (uint32_t)bt; }
    const dim3 g(blocks), b(PH_TRAV_BLOCK);
#define PH_LAUNCH3(cnt, lm, rm, ld, ns, inst, wpe, pk)                                                                                                  \
    do {                                                                                                                                           \
        if (mode == 2) hipLaunchKernelGGL((ph::traverse_kernel<false, cnt, lm, rm, ld, ns, inst, true, 0, wpe>), g, b, 0, s->stream, s->ds, p);   \
        else if (mode == 1) hipLaunchKernelGGL((ph::traverse_kernel<true, cnt, lm, rm, ld, ns, inst, false, 0, wpe>), g, b, 0, s->stream, s->ds, p); \
        else hipLaunchKernelGGL((ph::traverse_kernel<false, cnt, lm, rm, ld, ns, inst, false, 0, wpe>), g, b, 0, s->stream, s->ds, p);             \
    } while (0)
#define PH_LAUNCH3AM(cnt, lm, rm, ld, ns, inst, alpha, wpe, amin)                                                                      \
    do {                                                                                                                              \
        if (mode == 2) hipLaunchKernelGGL((ph::traverse_kernel<false, cnt, lm, rm, ld, ns, inst, true, alpha, wpe, amin>), g, b, 0, s->stream, s->ds, p);       \
        else if (mode == 1) hipLaunchKernelGGL((ph::traverse_kernel<true, cnt, lm, rm, ld, ns, inst, false, alpha, wpe, amin>), g, b, 0, s->stream, s->ds, p);  \
        else hipLaunchKernelGGL((ph::traverse_kernel<false, cnt, lm, rm, ld, ns, inst, false, alpha, wpe, amin>), g, b, 0, s->stream, s->ds, p);                \
    } while (0)
#define PH_LAUNCH3A(cnt, lm, rm, ld, ns, inst, alpha, wpe) PH_LAUNCH3AM(cnt, lm, rm, ld, ns, inst, alpha, wpe, 0)
    if (s->alpha_textures) {  // meshes with alpha-mask textures: the ALPHA variants — 1 = the inlined test for image-map masks, 2 = the general evaluator out of line (traverse.h)
        const bool inst = !s->inst_recs.empty();
        if (s->alpha_lean) {
            if (s->count_traversal) { if (inst) PH_LAUNCH3A(true, PH_LEAF_MIN, PH_REFILL_MIN, PH_LDS_DEPTH, 1, true, 1, 0); else PH_LAUNCH3A(true, PH_LEAF_MIN, PH_REFILL_MIN, PH_LDS_DEPTH, 1, false, 1, 0); }
            // Round 4: lanes whose candidate hit needs its alpha mask's verdict wait at their record until 12 of the wave's lanes do (traverse.h, ALPHA_MIN): configs[4]'s traversal
            // 5 301 -> 4 520 ms per frame, same film (thresholds 4 / 8 / 12 / 20: 4 936 / 4 594 / 4 520 / 4 683 ms; same box, gpurun r04n).  PBRT_HIP_ALPHA_MIN=0 is the round-3 form.
            else if (inst && alpha_min() == 0) PH_LAUNCH3A(false, 24, 12, 11, 5, true, 1, 5);   // (the instancing form, five waves per SIMD: 107 registers where the compiler is free, 20 spilled at 96 — and still faster: configs[4]'s traversal 6.04 -> 5.66 s per frame, gpurun r03aq; before the deferred pops it needed 115 and lost)
            else if (inst && alpha_min() == 4) PH_LAUNCH3AM(false, 16, 12, 11, 5, true, 1, 5, 4);
            else if (inst && alpha_min() == 20) PH_LAUNCH3AM(false, 16, 12, 11, 5, true, 1, 5, 20);
            else if (inst) PH_LAUNCH3AM(false, 16, 12, 11, 5, true, 1, 5, 12);   // (leaf threshold 16: with the mask lanes waiting apart, fewer lanes need to gather at leaves — 4 513 -> 4 466 ms; 32: 4 717; refill at 20: 4 627; 3 / 8 node steps per pass: 4 510 / 4 795, gpurun r04q)
            else PH_LAUNCH3AM(false, 24, 12, PH_LDS_DEPTH, 5, false, 1, 0, 12);
        } else {
            if (s->count_traversal) { if (inst) PH_LAUNCH3A(true, PH_LEAF_MIN, PH_REFILL_MIN, PH_LDS_DEPTH, 1, true, 2, 0); else PH_LAUNCH3A(true, PH_LEAF_MIN, PH_REFILL_MIN, PH_LDS_DEPTH, 1, false, 2, 0); }
            else if (inst) PH_LAUNCH3AM(false, PH_LEAF_MIN, PH_REFILL_MIN, PH_LDS_DEPTH, 3, true, 2, 0, 12); else PH_LAUNCH3AM(false, PH_LEAF_MIN, PH_REFILL_MIN, PH_LDS_DEPTH, 3, false, 2, 0, 12);
        }
        return;
    }
    if (!s->inst_recs.empty()) {  // scenes with object instances: the TransformedPrimitive-aware kernels
        if (s->count_traversal) PH_LAUNCH3(true, PH_LEAF_MIN, PH_REFILL_MIN, PH_LDS_DEPTH, 1, true, 0, false);
        else {
            static const int iv = []() { const char* e = std::getenv("PBRT_HIP_INST_VARIANT"); const int v = e ? std::atoi(e) : PH_DEFAULT_INST_VARIANT; return (v < 1 || v > 2) ? PH_DEFAULT_INST_VARIANT : v; }();
            switch (iv) {   // round 3: 96 VGPRs without spills (102 where the compiler is free) and 31 KB of LDS (11 stack entries + 9 parked words per lane): FIVE blocks per CU.  1 000 x 10 k
                            // instances: 1 181 ms of traversal per frame against 1 293 for the 4-wave form of round 2 (slot 1; 108 VGPRs, 12 + 13 words of LDS), gpurun r03ad; 1 115 with the deferred pops (traverse.h).  Seven more
                            // loop shapes around 24 / 12 / 5 (16-24 / 12-20 / 4-8) measured 1 308 - 1 387 ms against 1 295 at 4 waves (gpurun r03z), four at 5 waves 1 114 - 1 135 against 1 117 (r03aq).
                case 1: PH_LAUNCH3(false, 24, 12, PH_LDS_DEPTH, 5, true, 0, false); break;
                default: PH_LAUNCH3(false, 24, 12, 11, 5, true, 5, false); break;   // (round 4, with finished rays written out at the refill: refill at 20 / 28: 1 044 / 1 101 ms against 1 030; leaf 16: 1 041; gpurun r04af)
            }
        }
        return;
    }
    if (s->count_traversal) { PH_LAUNCH3(true, PH_LEAF_MIN, PH_REFILL_MIN, PH_LDS_DEPTH, 1, false, 0, false); return; }
    switch (trav_variant()) {
#define X(id, lm, rm, ld, ns, wpe, pk) case id: PH_LAUNCH3(false, lm, rm, ld, ns, false, wpe, pk); break;
        PH_VARIANTS(X)
#undef X
    }
#undef PH_LAUNCH3
#undef PH_LAUNCH3A
#undef PH_LAUNCH3AM
}

int launch_traverse(PbrtHipScene* s, bool anyhit, const void* d_rays, void* d_out, uint32_t n, float* kernel_ms) {
    if (kernel_ms) *kernel_ms = 0.0f;
    if (n == 0) return PBRT_HIP_OK;
    int rc;
    if ((rc = ensure_traversal_workspace(s))) return rc;
    // fewer rays than resident lanes: shrink the grid so idle waves exit immediately
    const uint32_t blocks = std::min<uint32_t>(s->trav_blocks, (n + PH_TRAV_BLOCK - 1) / PH_TRAV_BLOCK);
    PH_CHECK(s, hipMemsetAsync(s->d_counter.p, 0, 4, s->stream));
    ph::TravParams p{};
    p.rays = (const ph::RayIn*)d_rays; p.out = d_out; p.n = n; p.n_ptr = nullptr; p.counter = (uint32_t*)s->d_counter.p;
    if (kernel_ms) PH_CHECK(s, hipEventRecord(s->ev0, s->stream));
    launch_traverse_kernel(s, anyhit ? 1 : 0, blocks, p);
    PH_CHECK(s, hipGetLastError());
    if (kernel_ms) {
        PH_CHECK(s, hipEventRecord(s->ev1, s->stream));
        PH_CHECK(s, hipEventSynchronize(s->ev1));
        PH_CHECK(s, hipEventElapsedTime(kernel_ms, s->ev0, s->ev1));
    }
    return PBRT_HIP_OK;
}

}  // namespace phost

using namespace phost;

namespace hm {
M4 m4_inverse(const M4& a) {
    int indxc[4] = {0, 0, 0, 0}, indxr[4] = {0, 0, 0, 0}, ipiv[4] = {0, 0, 0, 0};
    float minv[4][4];
    std::memcpy(minv, a.m, sizeof(minv));
    for (int i = 0; i < 4; i++) {
        int irow = 0, icol = 0;
        float big = 0.0f;
        for (int j = 0; j < 4; j++)
            if (ipiv[j] != 1)
                for (int k = 0; k < 4; k++)
                    if (ipiv[k] == 0) { float av = fabs_p(minv[j][k]); if (av >= big) { big = av; irow = j; icol = k; } }
        ipiv[icol] += 1;
        if (irow != icol) for (int k = 0; k < 4; k++) std::swap(minv[irow][k], minv[icol][k]);
        indxr[i] = irow; indxc[i] = icol;
        float pivinv = 1.0f / minv[icol][icol];
        minv[icol][icol] = 1.0f;
        for (int j = 0; j < 4; j++) minv[icol][j] *= pivinv;
        for (int j = 0; j < 4; j++)
            if (j != icol) {
                float save = minv[j][icol];
                minv[j][icol] = 0.0f;
                for (int k = 0; k < 4; k++) minv[j][k] -= minv[icol][k] * save;
            }
    }
    for (int j = 3; j >= 0; j--)
        if (indxr[j] != indxc[j]) for (int k = 0; k < 4; k++) std::swap(minv[k][indxr[j]], minv[k][indxc[j]]);
    M4 r; std::memcpy(r.m, minv, sizeof(minv)); return r;
}
}  // namespace hm

extern "C" {

int pbrt_hip_device_count(void) {
    return ph_guard(nullptr, "pbrt_hip_device_count", [&]() -> int {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); return e == hipErrorNoDevice ? 0 : PBRT_HIP_ERR_DEVICE; }
    return n;
    });
}

PbrtHipScene* pbrt_hip_scene_create(int device_ordinal) {
    return ph_guard_ptr<PbrtHipScene>("pbrt_hip_scene_create", [&]() -> PbrtHipScene* {
    int n = pbrt_hip_device_count();
    if (n <= 0 || device_ordinal < 0 || device_ordinal >= n) {
        set_err(nullptr, PBRT_HIP_ERR_NO_DEVICE, "pbrt_hip_scene_create: no usable HIP device (this library has no CPU path)");
        return nullptr;
    }
    if (hipSetDevice(device_ordinal) != hipSuccess) { set_err(nullptr, PBRT_HIP_ERR_DEVICE, "hipSetDevice failed"); return nullptr; }
    PbrtHipScene* s = new PbrtHipScene();
    s->device = device_ordinal;
    if (hipStreamCreate(&s->stream) != hipSuccess || hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess) {
        set_err(nullptr, PBRT_HIP_ERR_DEVICE, "stream/event creation failed");
        delete s;
        return nullptr;
    }
    return s;
    });
}

void pbrt_hip_scene_destroy(PbrtHipScene* s) {
    ph_guard_void([&]() {
    if (!s) return;
    free_multi(s);  // the other devices' contexts of a multi-device handle
    (void)hipSetDevice(s->device);
    (void)hipStreamSynchronize(s->stream);
    free_wavefront(s);
    free_owned(s);
    free_tree_dev(s);
    for (DevBuf* b : {&s->d_ld_func, &s->d_ld_cdf, &s->d_counter, &s->d_spill, &s->d_error, &s->d_counts, &s->d_rays_tmp, &s->d_out_tmp})
        if (b->p) (void)hipFree(b->p);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
    });
}

const char* pbrt_hip_last_error(const PbrtHipScene* s) {
    if (s) return s->err.c_str();
    std::lock_guard<std::mutex> g(g_err_mu);
    return g_last_error.c_str();
}

// ---- materials: the BxDF list compute_scattering_functions builds for constant textures (materials/src/*.rs) ------------------
namespace {
float roughness_to_alpha(float roughness) {  // TrowbridgeReitzDistribution::roughness_to_alpha (microfacet/trowbridge_reitz.rs:31-40); host libm logf as for the reference
    roughness = roughness > 1e-3f ? roughness : 1e-3f;
    const float x = std::log(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
bool clamp3(const float in[3], float out[3]) {  // clamp_default(); returns !is_black()
    for (int c = 0; c < 3; c++) out[c] = hm::clampf(in[c], 0.0f, hm::kInf);
    return !(out[0] == 0.0f && out[1] == 0.0f && out[2] == 0.0f);
}
LobeRec lobe(uint32_t kind, uint32_t type) { LobeRec l{}; l.kind = kind; l.type = type; l.eta_a = l.eta_b = 1.0f; return l; }
void set_tr(LobeRec& l, float ax, float ay) { l.ax = 0.001f > ax ? 0.001f : ax; l.ay = 0.001f > ay ? 0.001f : ay; }  // TrowbridgeReitzDistribution::new: max(0.001, alpha)
enum : uint32_t { T_REFL = 1, T_TRANS = 2, T_DIFF = 4, T_GLOSSY = 8, T_SPEC = 16 };
int push_material(PbrtHipScene* s, MaterialRec& m, const std::vector<LobeRec>& lobes, bool general, uint32_t* out_id) {
    m.lobe_base = (uint32_t)s->lobes.size(); m.n_lobes = (uint32_t)lobes.size();
    s->lobes.insert(s->lobes.end(), lobes.begin(), lobes.end());
    s->materials.push_back(m);
    s->material_params.resize(s->materials.size());
    if (general) s->general_materials = true;
    if (out_id) *out_id = (uint32_t)s->materials.size() - 1;
    s->uploaded = false;
    return PBRT_HIP_OK;
}
}  // namespace

int pbrt_hip_add_material_matte(PbrtHipScene* s, const float kd[3], float sigma_deg, uint32_t* out_id) {
    return ph_guard(s, "pbrt_hip_add_material_matte", [&]() -> int {
    if (!s || !kd) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_material_matte: null argument");
    MaterialRec m{};
    for (int c = 0; c < 3; c++) m.kd[c] = hm::clampf(kd[c], 0.0f, hm::kInf);  // clamp_default (matte.rs:63)
    m.sigma = hm::clampf(sigma_deg, 0.0f, 90.0f);                              // matte.rs:64
    m.has_bxdf = !(m.kd[0] == 0.0f && m.kd[1] == 0.0f && m.kd[2] == 0.0f);
    m.bsdf_eta = 1.0f;
    if (m.sigma != 0.0f) {  // OrenNayar::new (oren_nayar.rs:28-39)
        float sg = hm::to_radians(m.sigma), s2 = sg * sg;
        m.a = 1.0f - (s2 / (2.0f * (s2 + 0.33f)));
        m.b = 0.45f * s2 / (s2 + 0.09f);
    }
    std::vector<LobeRec> lobes;
    if (m.has_bxdf) {
        LobeRec l = lobe(m.sigma != 0.0f ? PH_LK_OREN : PH_LK_LAMBERT, T_REFL | T_DIFF);
        std::memcpy(l.r, m.kd, 12); l.a = m.a; l.b = m.b;
        lobes.push_back(l);
    }
    const int rc = push_material(s, m, lobes, false, out_id);
    if (rc == PBRT_HIP_OK && m.has_bxdf) s->material_params.back().lobe[0] = 0;
    return rc;
    });
}
int pbrt_hip_add_material_none(PbrtHipScene* s, uint32_t* out_id) {  // Material "none" / "" (graphics_state.rs make_material -> None)
    return ph_guard(s, "pbrt_hip_add_material_none", [&]() -> int {
    if (!s) return PBRT_HIP_ERR_INVALID_ARG;
    MaterialRec m{}; m.bsdf_eta = 1.0f; m.none = 1u;
    s->has_none_material = true;
    return push_material(s, m, {}, false, out_id);
    });
}
int pbrt_hip_add_material_mirror(PbrtHipScene* s, const float kr[3], uint32_t* out_id) {  // mirror.rs:40-62
    return ph_guard(s, "pbrt_hip_add_material_mirror", [&]() -> int {
    if (!s || !kr) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_material_mirror: null argument");
    MaterialRec m{}; m.bsdf_eta = 1.0f;
    std::vector<LobeRec> lobes;
    float r[3];
    if (clamp3(kr, r)) { LobeRec l = lobe(PH_LK_SPEC_R, T_REFL | T_SPEC); l.fresnel = PH_FR_NOOP; std::memcpy(l.r, r, 12); lobes.push_back(l); }
    const int rc = push_material(s, m, lobes, true, out_id);
    if (rc == PBRT_HIP_OK && !lobes.empty()) s->material_params.back().lobe[2] = 0;
    return rc;
    });
}
int pbrt_hip_add_material_plastic(PbrtHipScene* s, const float kd[3], const float ks[3], float roughness, int remap_roughness, uint32_t* out_id) {  // plastic.rs:50-82
    return ph_guard(s, "pbrt_hip_add_material_plastic", [&]() -> int {
    if (!s || !kd || !ks) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_material_plastic: null argument");
    MaterialRec m{}; m.bsdf_eta = 1.0f;
    std::vector<LobeRec> lobes;
    float d[3], sp[3];
    if (clamp3(kd, d)) { LobeRec l = lobe(PH_LK_LAMBERT, T_REFL | T_DIFF); std::memcpy(l.r, d, 12); lobes.push_back(l); }
    if (clamp3(ks, sp)) {
        LobeRec l = lobe(PH_LK_MICRO_R, T_REFL | T_GLOSSY); l.fresnel = PH_FR_DIEL; l.eta_a = 1.5f; l.eta_b = 1.0f; std::memcpy(l.r, sp, 12);
        const float rough = remap_roughness ? roughness_to_alpha(roughness) : roughness;
        set_tr(l, rough, rough);
        lobes.push_back(l);
    }
    const int kd_lobe = (!lobes.empty() && lobes[0].kind == PH_LK_LAMBERT) ? 0 : -1, ks_lobe = (!lobes.empty() && lobes.back().kind == PH_LK_MICRO_R) ? (int)lobes.size() - 1 : -1;
    const int rc = push_material(s, m, lobes, true, out_id);
    if (rc == PBRT_HIP_OK) { PbrtHipScene::MaterialParams& mp = s->material_params.back(); mp.lobe[0] = kd_lobe; mp.lobe[1] = ks_lobe; mp.rough_lobe = ks_lobe; mp.rough_remap = remap_roughness != 0; }
    return rc;
    });
}
int pbrt_hip_add_material_glass(PbrtHipScene* s, const float kr[3], const float kt[3], float urough, float vrough, float eta, int remap_roughness,
                                uint32_t* out_id) {  // glass.rs:62-118 with allow_multiple_lobes = true (path.rs:143)
    return ph_guard(s, "pbrt_hip_add_material_glass", [&]() -> int {
    if (!s || !kr || !kt) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_material_glass: null argument");
    MaterialRec m{}; m.bsdf_eta = 1.0f;  // BSDF::new(.., None) (glass.rs:82)
    const float glass_raw_ur = urough, glass_raw_vr = vrough;   // before remapping: what a later roughness texture's `== 0` test is combined with
    std::vector<LobeRec> lobes;
    float r[3], t[3];
    const bool rb = clamp3(kr, r), tb = clamp3(kt, t);
    if (rb || tb) {
        if (urough == 0.0f && vrough == 0.0f) {
            LobeRec l = lobe(PH_LK_FRESNEL_SPEC, T_REFL | T_TRANS | T_SPEC); std::memcpy(l.r, r, 12); std::memcpy(l.t, t, 12); l.eta_a = 1.0f; l.eta_b = eta;
            lobes.push_back(l);
        } else {
            if (remap_roughness) { urough = roughness_to_alpha(urough); vrough = roughness_to_alpha(vrough); }
            if (rb) { LobeRec l = lobe(PH_LK_MICRO_R, T_REFL | T_GLOSSY); l.fresnel = PH_FR_DIEL; l.eta_a = 1.0f; l.eta_b = eta; std::memcpy(l.r, r, 12); set_tr(l, urough, vrough); lobes.push_back(l); }
            if (tb) { LobeRec l = lobe(PH_LK_MICRO_T, T_TRANS | T_GLOSSY); l.fresnel = PH_FR_DIEL; l.eta_a = 1.0f; l.eta_b = eta; std::memcpy(l.t, t, 12); set_tr(l, urough, vrough); lobes.push_back(l); }
        }
    }
    const float raw_ur = urough, raw_vr = vrough;
    const int rc = push_material(s, m, lobes, true, out_id);
    if (rc == PBRT_HIP_OK) {  // Kr -> the reflection colour, Kt -> the transmission colour (one FresnelSpecular lobe, or the two microfacet lobes)
        PbrtHipScene::MaterialParams& mp = s->material_params.back();
        mp.made_as = 2; std::memcpy(mp.raw_k[2], kr, 12); std::memcpy(mp.raw_k[3], kt, 12); mp.raw_eta = eta; mp.raw_ur = glass_raw_ur; mp.raw_vr = glass_raw_vr; mp.raw_remap = remap_roughness != 0;
        (void)raw_ur; (void)raw_vr;
        for (size_t i = 0; i < lobes.size(); i++) {
            if (lobes[i].kind == PH_LK_FRESNEL_SPEC) { mp.lobe[2] = (int)i; mp.field[2] = 0; mp.lobe[3] = (int)i; mp.field[3] = 1; }
            else if (lobes[i].kind == PH_LK_MICRO_R) { mp.lobe[2] = (int)i; mp.field[2] = 0; }
            else if (lobes[i].kind == PH_LK_MICRO_T) { mp.lobe[3] = (int)i; mp.field[3] = 1; }
        }
    }
    return rc;
    });
}
int pbrt_hip_add_material_metal(PbrtHipScene* s, const float eta[3], const float k[3], float urough, float vrough, int remap_roughness, uint32_t* out_id) {  // metal.rs:62-98
    return ph_guard(s, "pbrt_hip_add_material_metal", [&]() -> int {
    if (!s || !eta || !k) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_material_metal: null argument");
    MaterialRec m{}; m.bsdf_eta = 1.0f;
    if (remap_roughness) { urough = roughness_to_alpha(urough); vrough = roughness_to_alpha(vrough); }
    LobeRec l = lobe(PH_LK_MICRO_R, T_REFL | T_GLOSSY); l.fresnel = PH_FR_COND;
    l.r[0] = l.r[1] = l.r[2] = 1.0f; std::memcpy(l.c_eta_t, eta, 12); std::memcpy(l.c_k, k, 12);
    set_tr(l, urough, vrough);
    const int rc = push_material(s, m, {l}, true, out_id);
    if (rc == PBRT_HIP_OK) { s->material_params.back().rough_lobe = 0; s->material_params.back().rough_remap = remap_roughness != 0; s->material_params.back().made_as = 3; }
    return rc;
    });
}
int pbrt_hip_add_material_uber(PbrtHipScene* s, const float kd[3], const float ks[3], const float kr[3], const float kt[3], const float opacity[3], float urough,
                               float vrough, float eta, int remap_roughness, uint32_t* out_id) {  // uber.rs:116-186
    return ph_guard(s, "pbrt_hip_add_material_uber", [&]() -> int {
    if (!s || !kd || !ks || !kr || !kt || !opacity) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_material_uber: null argument");
    MaterialRec m{};
    const float uber_raw_ur = urough, uber_raw_vr = vrough;
    std::vector<LobeRec> lobes;
    float op[3], t[3], tmp[3];
    clamp3(opacity, op);
    for (int c = 0; c < 3; c++) tmp[c] = op[c] * -1.0f + 1.0f;  // (-op + Spectrum::ONE)
    if (clamp3(tmp, t)) {
        m.bsdf_eta = 1.0f;
        LobeRec l = lobe(PH_LK_SPEC_T, T_TRANS | T_SPEC); l.fresnel = PH_FR_DIEL; std::memcpy(l.t, t, 12); l.eta_a = 1.0f; l.eta_b = 1.0f; lobes.push_back(l);
    } else m.bsdf_eta = eta;
    auto scaled = [&](const float in[3], float out[3]) {  // op * k.clamp_default()
        float c[3]; clamp3(in, c);
        for (int i = 0; i < 3; i++) out[i] = op[i] * c[i];
        return !(out[0] == 0.0f && out[1] == 0.0f && out[2] == 0.0f);
    };
    float v[3];
    if (scaled(kd, v)) { LobeRec l = lobe(PH_LK_LAMBERT, T_REFL | T_DIFF); std::memcpy(l.r, v, 12); lobes.push_back(l); }
    if (scaled(ks, v)) {
        LobeRec l = lobe(PH_LK_MICRO_R, T_REFL | T_GLOSSY); l.fresnel = PH_FR_DIEL; l.eta_a = 1.0f; l.eta_b = eta; std::memcpy(l.r, v, 12);
        if (remap_roughness) { urough = roughness_to_alpha(urough); vrough = roughness_to_alpha(vrough); }
        set_tr(l, urough, vrough);
        lobes.push_back(l);
    }
    if (scaled(kr, v)) { LobeRec l = lobe(PH_LK_SPEC_R, T_REFL | T_SPEC); l.fresnel = PH_FR_DIEL; l.eta_a = 1.0f; l.eta_b = eta; std::memcpy(l.r, v, 12); lobes.push_back(l); }
    int kt_lobe = -1;
    if (scaled(kt, v)) { LobeRec l = lobe(PH_LK_SPEC_T, T_TRANS | T_SPEC); l.fresnel = PH_FR_DIEL; std::memcpy(l.t, v, 12); l.eta_a = 1.0f; l.eta_b = eta; kt_lobe = (int)lobes.size(); lobes.push_back(l); }
    const int rc = push_material(s, m, lobes, true, out_id);
    if (rc == PBRT_HIP_OK) {  // texturable: Kd, Ks, Kr, Kt, each multiplied by the (constant) opacity as in `op * k.evaluate().clamp_default()`
        PbrtHipScene::MaterialParams& mp = s->material_params.back();
        mp.has_pre = true; std::memcpy(mp.pre, op, 12);
        mp.made_as = 1; std::memcpy(mp.raw_k[0], kd, 12); std::memcpy(mp.raw_k[1], ks, 12); std::memcpy(mp.raw_k[2], kr, 12); std::memcpy(mp.raw_k[3], kt, 12);
        mp.raw_eta = eta; mp.raw_ur = uber_raw_ur; mp.raw_vr = uber_raw_vr; mp.raw_remap = remap_roughness != 0;
        for (size_t i = 0; i < lobes.size(); i++) {
            if (lobes[i].kind == PH_LK_LAMBERT) mp.lobe[0] = (int)i;
            else if (lobes[i].kind == PH_LK_MICRO_R) { mp.lobe[1] = (int)i; mp.rough_lobe = (int)i; mp.rough_remap = remap_roughness != 0; }
            else if (lobes[i].kind == PH_LK_SPEC_R) mp.lobe[2] = (int)i;
        }
        if (kt_lobe >= 0) { mp.lobe[3] = kt_lobe; mp.field[3] = 1; }
    }
    return rc;
    });
}

int pbrt_hip_add_material_substrate(PbrtHipScene* s, const float kd[3], const float ks[3], float urough, float vrough, int remap_roughness, uint32_t* out_id) {  // substrate.rs:55-84
    return ph_guard(s, "pbrt_hip_add_material_substrate", [&]() -> int {
    if (!s || !kd || !ks) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_material_substrate: null argument");
    MaterialRec m{}; m.bsdf_eta = 1.0f;
    std::vector<LobeRec> lobes;
    float d[3], sp[3];
    const bool db = clamp3(kd, d), sb = clamp3(ks, sp);
    if (db || sb) {
        if (remap_roughness) { urough = roughness_to_alpha(urough); vrough = roughness_to_alpha(vrough); }
        LobeRec l = lobe(PH_LK_FRESNEL_BLEND, T_REFL | T_GLOSSY); std::memcpy(l.r, d, 12); std::memcpy(l.t, sp, 12); set_tr(l, urough, vrough);
        lobes.push_back(l);
    }
    const int rc = push_material(s, m, lobes, true, out_id);
    if (rc == PBRT_HIP_OK && !lobes.empty()) { PbrtHipScene::MaterialParams& mp = s->material_params.back(); mp.lobe[0] = 0; mp.field[0] = 0; mp.lobe[1] = 0; mp.field[1] = 1; mp.rough_lobe = 0; mp.rough_remap = remap_roughness != 0; }
    return rc;
    });
}
int pbrt_hip_add_material_translucent(PbrtHipScene* s, const float kd[3], const float ks[3], const float reflect[3], const float transmit[3], float roughness,
                                      int remap_roughness, uint32_t* out_id) {  // translucent.rs:57-112
    return ph_guard(s, "pbrt_hip_add_material_translucent", [&]() -> int {
    if (!s || !kd || !ks || !reflect || !transmit) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_material_translucent: null argument");
    MaterialRec m{}; m.bsdf_eta = 1.5f;
    std::vector<LobeRec> lobes;
    float r[3], t[3], d[3], sp[3], v[3];
    const bool rb = clamp3(reflect, r), tb = clamp3(transmit, t);
    if (!rb && !tb) { m.none = 1u; s->has_none_material = true; }   // `return` before a BSDF is made (translucent.rs:72-74): every hit is passed through, as with Material "none" — until a texture takes reflect's / transmit's place
    auto prod = [&](const float a[3], const float b[3]) { for (int c = 0; c < 3; c++) v[c] = a[c] * b[c]; };
    // each lobe keeps the reflect / transmit factor of its product (pre, PH_PRE_RAW_TEST) so that a Kd / Ks texture can replace the other factor per hit
    PbrtHipScene::MaterialParams mp;
    auto feed = [&](int param, int field) { if (mp.lobe[param] < 0) { mp.lobe[param] = (int)lobes.size(); mp.field[param] = field; } else { mp.lobe2[param] = (int)lobes.size(); mp.field2[param] = field; } };
    auto raw = [&](LobeRec& l, const float f[3]) { std::memcpy(l.pre, f, 12); l.has_pre = PH_PRE_RAW_TEST; };
    if (clamp3(kd, d)) {
        if (rb) { LobeRec l = lobe(PH_LK_LAMBERT, T_REFL | T_DIFF); prod(r, d); std::memcpy(l.r, v, 12); raw(l, r); feed(0, 0); lobes.push_back(l); }
        if (tb) { LobeRec l = lobe(PH_LK_LAMBERT_T, T_TRANS | T_DIFF); prod(t, d); std::memcpy(l.t, v, 12); raw(l, t); feed(0, 1); lobes.push_back(l); }
    }
    if (clamp3(ks, sp)) {
        const float rough = remap_roughness ? roughness_to_alpha(roughness) : roughness;
        mp.rough_remap = remap_roughness != 0;
        if (rb) { LobeRec l = lobe(PH_LK_MICRO_R, T_REFL | T_GLOSSY); l.fresnel = PH_FR_DIEL; l.eta_a = 1.0f; l.eta_b = 1.5f; prod(r, sp); std::memcpy(l.r, v, 12); set_tr(l, rough, rough); raw(l, r);
                  feed(1, 0); mp.rough_lobe = (int)lobes.size(); lobes.push_back(l); }
        if (tb) { LobeRec l = lobe(PH_LK_MICRO_T, T_TRANS | T_GLOSSY); l.fresnel = PH_FR_DIEL; l.eta_a = 1.0f; l.eta_b = 1.5f; prod(t, sp); std::memcpy(l.t, v, 12); set_tr(l, rough, rough); raw(l, t);
                  feed(1, 1); (mp.rough_lobe < 0 ? mp.rough_lobe : mp.rough_lobe2) = (int)lobes.size(); lobes.push_back(l); }
    }
    mp.made_as = 5; clamp3(kd, mp.raw_k[0]); clamp3(ks, mp.raw_k[1]); std::memcpy(mp.raw_k[2], r, 12); std::memcpy(mp.raw_k[3], t, 12); mp.raw_ur = roughness; mp.raw_remap = remap_roughness != 0;
    const int rc = push_material(s, m, lobes, true, out_id);
    if (rc == PBRT_HIP_OK) s->material_params.back() = mp;
    return rc;
    });
}
int pbrt_hip_add_material_mix(PbrtHipScene* s, uint32_t material1, uint32_t material2, const float amount[3], uint32_t* out_id) {  // mix.rs:51-88
    return ph_guard(s, "pbrt_hip_add_material_mix", [&]() -> int {
    if (!s || !amount) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_material_mix: null argument");
    if (material1 >= s->materials.size() || material2 >= s->materials.size()) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_material_mix: unknown material id");
    MaterialRec m{}; m.bsdf_eta = 1.0f;
    float s1[3], s2[3], tmp[3];
    clamp3(amount, s1);
    for (int c = 0; c < 3; c++) tmp[c] = 1.0f - s1[c];
    clamp3(tmp, s2);
    const MaterialRec a = s->materials[material1], b = s->materials[material2];
    // mix.rs:63-76: m1 bumps `si` itself, m2 a clone of it, and the mixture's BSDF is made on `si`: the first material's bump map shapes the frame of every
    // lobe, the second one's changes nothing that is used
    m.bump_tex1 = a.bump_tex1;
    if (a.amount_tex1 || b.amount_tex1) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "add_material_mix: a sub-material is a mix whose amount is a texture; not supported");
    if (a.rt_mode || b.rt_mode) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "add_material_mix: a translucent sub-material with a reflect / transmit texture (a per-hit null BSDF) is not supported");
    if (a.index_tex1 || b.index_tex1) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "add_material_mix: a sub-material with a textured index of refraction is not supported");
    if (a.opacity_tex1 && b.opacity_tex1) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "add_material_mix: both sub-materials are uber materials with opacity textures; not supported");
    m.opacity_tex1 = a.opacity_tex1 ? a.opacity_tex1 : b.opacity_tex1;
    if (a.n_lobes + b.n_lobes > 8) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_material_mix: more than MAX_BXDFS = 8 lobes (BSDF::add asserts, bsdf.rs:119-125)");
    std::vector<LobeRec> lobes;
    auto take = [&](const MaterialRec& src, const float sc[3]) {
        for (uint32_t k = 0; k < src.n_lobes; k++) {
            LobeRec l = s->lobes[src.lobe_base + k];
            if (l.n_scale >= 2) return false;
            std::memcpy(l.n_scale == 0 ? l.scale0 : l.scale1, sc, 12); l.n_scale++;
            lobes.push_back(l);
        }
        return true;
    };
    if (!take(a, s1) || !take(b, s2)) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "add_material_mix: mix nested deeper than two levels");
    if (a.textured || b.textured) {
        // the sub-materials' textures are evaluated per hit and each lobe is kept or dropped as its own material would (mix.rs:63-87); the per-hit list has
        // PH_HIT_LOBES lobe slots, PH_HIT_COLS colour slots and ONE pair of per-hit scalars (roughness or sigma)
        uint32_t cols = 0, scal = 0; const LobeRec* first = nullptr;
        for (const LobeRec& l : lobes) {
            cols += (l.r_tex1 || (l.has_pre == PH_PRE_OPACITY && l.kind != PH_LK_SPEC_T) ? 1u : 0u) + (l.t_tex1 || (l.has_pre == PH_PRE_OPACITY && l.kind == PH_LK_SPEC_T) || l.has_pre == PH_PRE_PASSTHROUGH ? 1u : 0u) +
                    (l.eta_tex1 ? 1u : 0u) + (l.k_tex1 ? 1u : 0u);
            if (l.ax_tex1 || l.ay_tex1 || l.sigma_tex1) {
                if (!first) { first = &l; scal = 1; }
                else if (l.ax_tex1 != first->ax_tex1 || l.ay_tex1 != first->ay_tex1 || l.remap != first->remap || l.sigma_tex1 != first->sigma_tex1 || l.ax != first->ax || l.ay != first->ay) scal++;
            }
        }
        if (lobes.size() > PH_HIT_LOBES || cols > PH_HIT_COLS || scal > 1u)
            return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "add_material_mix: a textured mix may have at most 8 lobes, 6 textured colours and one textured roughness / sigma");
        m.textured = 1u;
    }
    const int rc = push_material(s, m, lobes, true, out_id);
    if (rc == PBRT_HIP_OK) { s->material_params.back().made_as = 4; s->material_params.back().mix_n1 = (int)a.n_lobes; if (m.bump_tex1) { s->textured_materials = true; s->bump_materials = true; } }
    return rc;
    });
}

}  // extern "C"

namespace phost {
// TranslucentMaterial with a reflect / transmit texture: the lobe list becomes every lobe the material CAN have (translucent.rs:76-98); a lobe's colour is the product of the hit's
// reflect or transmit with the hit's Kd or Ks, kept where neither factor is black (PH_PRE_RT); where reflect and transmit are both black the hit has no BSDF at all.
int translucent_rebuild_rt(PbrtHipScene* s, uint32_t material) {
    PbrtHipScene::MaterialParams& mp = s->material_params[material];
    MaterialRec& m = s->materials[material];
    if (m.rt_mode) return PBRT_HIP_OK;
    std::vector<LobeRec> old(s->lobes.begin() + m.lobe_base, s->lobes.begin() + m.lobe_base + m.n_lobes), lobes;
    uint32_t btex1[2] = {0u, 0u}, axt = 0u, ayt = 0u;   // Kd / Ks and roughness textures set so far
    for (int k = 0; k < 2; k++) if (mp.lobe[k] >= 0) { const LobeRec& l = old[(size_t)mp.lobe[k]]; btex1[k] = mp.field[k] == 0 ? l.r_tex1 : l.t_tex1; }
    if (mp.rough_lobe >= 0) { axt = old[(size_t)mp.rough_lobe].ax_tex1; ayt = old[(size_t)mp.rough_lobe].ay_tex1; }
    for (int k = 0; k < 4; k++) { mp.lobe[k] = mp.lobe2[k] = -1; mp.field[k] = mp.field2[k] = 0; }
    mp.rough_lobe = mp.rough_lobe2 = -1;
    for (int k = 0; k < 2; k++) {
        const bool black = mp.raw_k[k][0] == 0.0f && mp.raw_k[k][1] == 0.0f && mp.raw_k[k][2] == 0.0f;
        if (black && !btex1[k]) continue;   // Kd / Ks black at every hit
        for (int side = 0; side < 2; side++) {
            LobeRec l;
            if (k == 0) l = side ? lobe(PH_LK_LAMBERT_T, T_TRANS | T_DIFF) : lobe(PH_LK_LAMBERT, T_REFL | T_DIFF);
            else {
                l = side ? lobe(PH_LK_MICRO_T, T_TRANS | T_GLOSSY) : lobe(PH_LK_MICRO_R, T_REFL | T_GLOSSY);
                l.fresnel = PH_FR_DIEL; l.eta_a = 1.0f; l.eta_b = 1.5f;
                const float rough = mp.raw_remap ? roughness_to_alpha(mp.raw_ur) : mp.raw_ur;
                set_tr(l, rough, rough);
                l.ax_tex1 = axt; l.ay_tex1 = ayt; l.remap = mp.raw_remap ? 1u : 0u;
                (side ? mp.rough_lobe2 : mp.rough_lobe) = (int)lobes.size();
            }
            l.has_pre = PH_PRE_RT; std::memcpy(l.pre, mp.raw_k[k], 12);
            (side ? l.t_tex1 : l.r_tex1) = btex1[k];
            (side ? mp.lobe2[k] : mp.lobe[k]) = (int)lobes.size(); (side ? mp.field2[k] : mp.field[k]) = side;
            lobes.push_back(l);
        }
    }
    mp.rough_remap = mp.raw_remap; mp.has_pre = false;
    m.lobe_base = (uint32_t)s->lobes.size(); m.n_lobes = (uint32_t)lobes.size();
    s->lobes.insert(s->lobes.end(), lobes.begin(), lobes.end());
    m.rt_mode = 1u; m.none = 0u; m.textured = 1u;
    std::memcpy(m.refl_c, mp.raw_k[2], 12); std::memcpy(m.trans_c, mp.raw_k[3], 12);
    s->has_none_material = true;   // some hits of this material may have no BSDF: paths can need more rounds than max_depth + 1
    s->textured_materials = true; s->general_materials = true;
    s->uploaded = false;
    return PBRT_HIP_OK;
}
// UberMaterial with an opacity texture: the lobe list becomes every lobe the material CAN have (uber.rs:126-160); colours, presence and BSDF::eta are decided per hit.
// The new list is appended to the lobe pool (the old one stays behind, unused).
int uber_rebuild_for_opacity(PbrtHipScene* s, uint32_t material) {
    PbrtHipScene::MaterialParams& mp = s->material_params[material];
    MaterialRec& m = s->materials[material];
    if (mp.rebuilt) return PBRT_HIP_OK;
    std::vector<LobeRec> old(s->lobes.begin() + m.lobe_base, s->lobes.begin() + m.lobe_base + m.n_lobes), lobes;
    int oldp[4]; for (int k = 0; k < 4; k++) oldp[k] = mp.lobe[k];
    const int old_rough = mp.rough_lobe;
    for (int k = 0; k < 4; k++) { mp.lobe[k] = -1; mp.field[k] = 0; }
    mp.rough_lobe = -1;
    { LobeRec l = lobe(PH_LK_SPEC_T, T_TRANS | T_SPEC); l.fresnel = PH_FR_DIEL; l.eta_a = 1.0f; l.eta_b = 1.0f; l.has_pre = PH_PRE_PASSTHROUGH; lobes.push_back(l); }
    const float e = mp.raw_eta;
    for (int k = 0; k < 4; k++) {
        float base[3];
        const bool nonblack = clamp3(mp.raw_k[k], base);
        const LobeRec* ol = oldp[k] >= 0 ? &old[(size_t)oldp[k]] : nullptr;
        const uint32_t tex1 = ol ? (k == 3 ? ol->t_tex1 : ol->r_tex1) : 0u;
        if (!nonblack && !tex1) continue;   // op * 0 is black at every hit: the lobe is never added
        LobeRec l;
        if (k == 0) { l = lobe(PH_LK_LAMBERT, T_REFL | T_DIFF); l.r_tex1 = tex1; }
        else if (k == 1) {
            l = lobe(PH_LK_MICRO_R, T_REFL | T_GLOSSY); l.fresnel = PH_FR_DIEL; l.eta_a = 1.0f; l.eta_b = e; l.r_tex1 = tex1;
            float ur = mp.raw_ur, vr = mp.raw_vr;
            if (mp.raw_remap) { ur = roughness_to_alpha(ur); vr = roughness_to_alpha(vr); }
            set_tr(l, ur, vr);
            if (old_rough >= 0) { const LobeRec& r = old[(size_t)old_rough]; l.ax_tex1 = r.ax_tex1; l.ay_tex1 = r.ay_tex1; l.remap = r.remap; }
            mp.rough_lobe = (int)lobes.size(); mp.rough_remap = mp.raw_remap;
        }
        else if (k == 2) { l = lobe(PH_LK_SPEC_R, T_REFL | T_SPEC); l.fresnel = PH_FR_DIEL; l.eta_a = 1.0f; l.eta_b = e; l.r_tex1 = tex1; }
        else { l = lobe(PH_LK_SPEC_T, T_TRANS | T_SPEC); l.fresnel = PH_FR_DIEL; l.eta_a = 1.0f; l.eta_b = e; l.t_tex1 = tex1; mp.field[3] = 1; }
        l.has_pre = PH_PRE_OPACITY; std::memcpy(l.pre, base, 12);
        mp.lobe[k] = (int)lobes.size();
        lobes.push_back(l);
    }
    mp.has_pre = false;   // the per-hit opacity takes the constant's place (set_material_texture must not overwrite has_pre = PH_PRE_OPACITY)
    m.lobe_base = (uint32_t)s->lobes.size(); m.n_lobes = (uint32_t)lobes.size();
    s->lobes.insert(s->lobes.end(), lobes.begin(), lobes.end());
    m.bsdf_eta = 1.0f; m.bsdf_eta_alt = e; m.uber_eta = 1u;
    mp.rebuilt = true;
    return PBRT_HIP_OK;
}
// GlassMaterial with a roughness texture: which lobes a hit gets depends on `urough == 0 && vrough == 0` there (glass.rs:110-141): the list holds the smooth
// alternative (FresnelSpecular; allow_multiple_lobes is true on this path) and the rough one (the microfacet pair)
int glass_rebuild_for_roughness(PbrtHipScene* s, uint32_t material) {
    PbrtHipScene::MaterialParams& mp = s->material_params[material];
    MaterialRec& m = s->materials[material];
    if (mp.rebuilt) return PBRT_HIP_OK;
    uint32_t rtex1 = 0, ttex1 = 0;
    for (uint32_t k = 0; k < m.n_lobes; k++) { const LobeRec& l = s->lobes[m.lobe_base + k]; if (l.r_tex1) rtex1 = l.r_tex1; if (l.t_tex1) ttex1 = l.t_tex1; }
    float r[3], t[3];
    const bool rb = clamp3(mp.raw_k[2], r), tb = clamp3(mp.raw_k[3], t);
    std::vector<LobeRec> lobes;
    for (int k = 0; k < 4; k++) { mp.lobe[k] = mp.lobe2[k] = -1; mp.field[k] = mp.field2[k] = 0; }
    mp.rough_lobe = mp.rough_lobe2 = -1;
    const bool has_r = rb || rtex1, has_t = tb || ttex1;
    if (has_r || has_t) {
        float ur = mp.raw_ur, vr = mp.raw_vr;
        if (mp.raw_remap) { ur = roughness_to_alpha(ur); vr = roughness_to_alpha(vr); }
        auto tag = [&](LobeRec& l, uint32_t alt) { l.alt = alt; l.ur_raw = mp.raw_ur; l.vr_raw = mp.raw_vr; set_tr(l, ur, vr); };
        LobeRec f = lobe(PH_LK_FRESNEL_SPEC, T_REFL | T_TRANS | T_SPEC); std::memcpy(f.r, r, 12); std::memcpy(f.t, t, 12); f.r_tex1 = rtex1; f.t_tex1 = ttex1; f.eta_a = 1.0f; f.eta_b = mp.raw_eta;
        tag(f, 1u);
        mp.lobe[2] = 0; mp.field[2] = 0; mp.lobe[3] = 0; mp.field[3] = 1;
        lobes.push_back(f);
        if (has_r) { LobeRec l = lobe(PH_LK_MICRO_R, T_REFL | T_GLOSSY); l.fresnel = PH_FR_DIEL; l.eta_a = 1.0f; l.eta_b = mp.raw_eta; std::memcpy(l.r, r, 12); l.r_tex1 = rtex1; tag(l, 2u);
                     mp.lobe2[2] = (int)lobes.size(); mp.field2[2] = 0; lobes.push_back(l); }
        if (has_t) { LobeRec l = lobe(PH_LK_MICRO_T, T_TRANS | T_GLOSSY); l.fresnel = PH_FR_DIEL; l.eta_a = 1.0f; l.eta_b = mp.raw_eta; std::memcpy(l.t, t, 12); l.t_tex1 = ttex1; tag(l, 2u);
                     mp.lobe2[3] = (int)lobes.size(); mp.field2[3] = 1; lobes.push_back(l); }
    }
    m.lobe_base = (uint32_t)s->lobes.size(); m.n_lobes = (uint32_t)lobes.size();
    s->lobes.insert(s->lobes.end(), lobes.begin(), lobes.end());
    mp.rebuilt = true;
    return PBRT_HIP_OK;
}
}  // namespace phost

extern "C" {

// "the intersection is bogus" (triangle.rs:548-574): depends only on the triangle, so it is decided once here.
static bool triangle_is_bogus(hm::V3 p0, hm::V3 p1, hm::V3 p2, const float* uv0, const float* uv1, const float* uv2) {
    float u0[2] = {0, 0}, u1[2] = {1, 0}, u2[2] = {1, 1};  // default uvs (triangle.rs:384-394)
    if (uv0) { u0[0] = uv0[0]; u0[1] = uv0[1]; u1[0] = uv1[0]; u1[1] = uv1[1]; u2[0] = uv2[0]; u2[1] = uv2[1]; }
    float duv02[2] = {u0[0] - u2[0], u0[1] - u2[1]}, duv12[2] = {u1[0] - u2[0], u1[1] - u2[1]};
    hm::V3 dp02 = hm::sub(p0, p2), dp12 = hm::sub(p1, p2);
    float determinant = duv02[0] * duv12[1] - duv02[1] * duv12[0];
    bool degenerate_uv = std::fabs(determinant) < 1e-8f;
    hm::V3 dpdu{0, 0, 0}, dpdv{0, 0, 0};
    if (!degenerate_uv) {
        float invdet = 1.0f / determinant;
        dpdu = hm::mulf(hm::sub(hm::mulf(dp02, duv12[1]), hm::mulf(dp12, duv02[1])), invdet);
        dpdv = hm::mulf(hm::add(hm::mulf(dp02, -duv12[0]), hm::mulf(dp12, duv02[0])), invdet);
    }
    if (degenerate_uv || hm::len2(hm::cross(dpdu, dpdv)) == 0.0f) {
        hm::V3 ng = hm::cross(hm::sub(p2, p0), hm::sub(p1, p0));
        if (hm::len2(ng) == 0.0f) return true;
    }
    return false;
}

int pbrt_hip_add_mesh(PbrtHipScene* s, const float* P, uint32_t n_verts, const uint32_t* indices, uint32_t n_tris, const float* N,
                      const float* S, const float* UV, uint32_t material_id, int32_t first_area_light_id, uint32_t flags, float alpha,
                      float shadow_alpha) {
    return ph_guard(s, "pbrt_hip_add_mesh", [&]() -> int {
    if (!s || !P || !indices) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_mesh: null argument");
    if (material_id >= s->materials.size()) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_mesh: unknown material id");
    for (uint32_t i = 0; i < 3 * n_tris; i++)
        if (indices[i] >= n_verts) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_mesh: vertex index out of bounds (triangle.rs:252-261)");
    if (first_area_light_id >= 0) {
        if ((size_t)first_area_light_id + n_tris > s->lights.size()) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_mesh: area light ids out of range");
        for (uint32_t k = 0; k < n_tris; k++)
            if (s->lights[first_area_light_id + k].type != PH_L_AREA || s->lights[first_area_light_id + k].prim != 0xFFFFFFFFu)
                return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_mesh: light is not an unbound diffuse area light");
    }
    bool dropped_lights = false;
    if (s->open_object >= 0 && first_area_light_id >= 0) {
        // "Area lights not supported with object instancing" (api/src/lib.rs:877-881): a warning, the primitives keep their area light — a path that looks at such a surface
        // sees its emission (SurfaceInteraction::le) — and the light is not added to the scene's lights: nothing samples it.
        if ((size_t)first_area_light_id + n_tris != s->lights.size())
            return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_mesh: the area lights of a shape inside an object definition must be the lights created last (they leave the scene's light list)");
        const int32_t first = -2 - (int32_t)s->emission_only.size();
        s->emission_only.insert(s->emission_only.end(), s->lights.begin() + first_area_light_id, s->lights.end());
        s->lights.resize((size_t)first_area_light_id);
        first_area_light_id = first;
        dropped_lights = true;
    }
    MeshRec m{};
    m.vert_base = (uint32_t)(s->P.size() / 3); m.tri_base = (uint32_t)(s->idx.size() / 3); m.n_tris = n_tris;
    m.flags = (N ? PH_MESH_N : 0) | (S ? PH_MESH_S : 0) | (UV ? PH_MESH_UV : 0) | ((flags & 1) ? PH_MESH_REV : 0) | ((flags & 2) ? PH_MESH_SWAP : 0);
    m.material = material_id; m.first_light = first_area_light_id;
    const size_t v0 = s->P.size() / 3;
    s->P.insert(s->P.end(), P, P + 3 * (size_t)n_verts);
    // optional attributes stay vertex-aligned with P; arrays are materialised only once some mesh supplies them
    auto grow_attr = [&](std::vector<float>& dst, bool& any, const float* src, int w) {
        if (src && !any) { dst.assign(v0 * w, 0.0f); any = true; }
        if (any) { if (src) dst.insert(dst.end(), src, src + (size_t)w * n_verts); else dst.insert(dst.end(), (size_t)w * n_verts, 0.0f); }
    };
    grow_attr(s->N, s->any_n, N, 3); grow_attr(s->S, s->any_s, S, 3); grow_attr(s->UV, s->any_uv, UV, 2);
    const uint32_t mesh_id = (uint32_t)s->meshes.size();
    for (uint32_t t = 0; t < n_tris; t++) {
        uint32_t i0 = indices[3 * t], i1 = indices[3 * t + 1], i2 = indices[3 * t + 2];
        s->idx.push_back(i0 + m.vert_base); s->idx.push_back(i1 + m.vert_base); s->idx.push_back(i2 + m.vert_base);
        s->tri_mesh.push_back(mesh_id);
        hm::V3 p0 = hm::ld(P + 3 * i0), p1 = hm::ld(P + 3 * i1), p2 = hm::ld(P + 3 * i2);
        uint32_t tf = 0;
        if (triangle_is_bogus(p0, p1, p2, UV ? UV + 2 * i0 : nullptr, UV ? UV + 2 * i1 : nullptr, UV ? UV + 2 * i2 : nullptr)) tf |= PH_TRI_BOGUS;
        if (alpha == 0.0f) tf |= PH_TRI_ALPHA0;
        if (shadow_alpha == 0.0f) tf |= PH_TRI_SALPHA0;
        s->tri_flags.push_back(tf);
        if (first_area_light_id >= 0 || dropped_lights) {
            LightRec& l = dropped_lights ? s->emission_only[(size_t)(-2 - first_area_light_id) + t] : s->lights[first_area_light_id + t];
            l.prim = m.tri_base + t;
            l.area = 0.5f * hm::len(hm::cross(hm::sub(p1, p0), hm::sub(p2, p0)));  // Triangle::area (triangle.rs:906-911)
        }
    }
    s->meshes.push_back(m);
    if (s->open_object >= 0) s->objects[s->open_object].tri1 = m.tri_base + n_tris;
    else for (uint32_t t = 0; t < n_tris; t++) s->top_items.push_back(m.tri_base + t);
    s->built = false; s->uploaded = false;
    if (dropped_lights) s->err = "warning: Area lights not supported with object instancing.";   // the reference's warn! (lib.rs:878); the call succeeds
    return PBRT_HIP_OK;
    });
}

// ObjectBegin / ObjectEnd / ObjectInstance (api/src/lib.rs:911-1000)
int pbrt_hip_object_begin(PbrtHipScene* s, uint32_t* out_object_id) {
    return ph_guard(s, "pbrt_hip_object_begin", [&]() -> int {
    if (!s || !out_object_id) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "object_begin: null argument");
    if (s->open_object >= 0) return set_err(s, PBRT_HIP_ERR_STATE, "object_begin: ObjectBegin called inside of an instance definition");
    PbrtHipScene::ObjectHost ob; ob.tri0 = ob.tri1 = (uint32_t)(s->idx.size() / 3);
    s->objects.push_back(ob);
    s->open_object = (int)s->objects.size() - 1;
    *out_object_id = (uint32_t)s->open_object;
    return PBRT_HIP_OK;
    });
}
int pbrt_hip_object_end(PbrtHipScene* s) {
    return ph_guard(s, "pbrt_hip_object_end", [&]() -> int {
    if (!s) return PBRT_HIP_ERR_INVALID_ARG;
    if (s->open_object < 0) return set_err(s, PBRT_HIP_ERR_STATE, "object_end: ObjectEnd called outside of instance definition");
    s->open_object = -1;
    return PBRT_HIP_OK;
    });
}
int pbrt_hip_add_instance(PbrtHipScene* s, uint32_t object_id, const float i2w[16], const float w2i[16]) {
    return ph_guard(s, "pbrt_hip_add_instance", [&]() -> int {
    if (!s || !i2w || !w2i) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_instance: null argument");
    if (s->open_object >= 0) return set_err(s, PBRT_HIP_ERR_STATE, "add_instance: ObjectInstance can't be called inside of instance definition");
    if (object_id >= s->objects.size()) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_instance: unknown object");
    if (s->objects[object_id].tri1 == s->objects[object_id].tri0) return PBRT_HIP_OK;  // empty object: nothing is added (lib.rs:949-951)
    PbrtHipScene::InstanceHost in; in.object = object_id;
    std::memcpy(in.i2w, i2w, 64); std::memcpy(in.w2i, w2i, 64);
    s->instances.push_back(in);
    s->top_items.push_back(PH_ITEM_INST | (uint32_t)(s->instances.size() - 1));
    s->built = false; s->uploaded = false;
    return PBRT_HIP_OK;
    });
}

// 1x1 MIPMap::triangle (core/src/mipmap/mod.rs:293-311) for a constant environment
static float env_lookup(float tx, float s_, float t_) {
    float s = s_ * 1.0f - 0.5f, t = t_ * 1.0f - 0.5f;
    float s0 = std::floor(s), t0 = std::floor(t);
    float ds = s - s0, dt = t - t0;
    return tx * (1.0f - ds) * (1.0f - dt) + tx * (1.0f - ds) * dt + tx * ds * (1.0f - dt) + tx * ds * dt;
}

int pbrt_hip_add_light_infinite(PbrtHipScene* s, const float L[3], const float l2w[16], const float w2l[16]) {
    return ph_guard(s, "pbrt_hip_add_light_infinite", [&]() -> int {
    if (!s || !L || !l2w || !w2l) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_light_infinite: null argument");
    LightRec l{};
    l.type = PH_L_INFINITE; l.prim = 0xFFFFFFFFu;
    for (int c = 0; c < 3; c++) l.L[c] = L[c];
    std::memcpy(l.l2w, l2w, 48); std::memcpy(l.w2l, w2l, 48);
    // compute_scalar_image (infinite.rs:326-369): 2x2 image, fwidth 0.25 -> level < 0 -> triangle(0, st)
    float img[2][2];
    for (int v = 0; v < 2; v++) {
        float vp = ((float)v + 0.5f) / 2.0f;
        float sin_theta = std::sin(hm::kPi * ((float)v + 0.5f) / 2.0f);
        for (int u = 0; u < 2; u++) {
            float up = ((float)u + 0.5f) / 2.0f;
            float y = 0.212671f * env_lookup(L[0], up, vp) + 0.715160f * env_lookup(L[1], up, vp) + 0.072169f * env_lookup(L[2], up, vp);
            img[v][u] = y * sin_theta;
        }
    }
    std::vector<float> marg, cdf;
    for (int v = 0; v < 2; v++) {  // Distribution2D::new (distribution_2d.rs:21-29)
        float fi;
        hm::distribution1d({img[v][0], img[v][1]}, cdf, fi);
        l.cond_func[2 * v] = img[v][0]; l.cond_func[2 * v + 1] = img[v][1];
        for (int i = 0; i < 3; i++) l.cond_cdf[3 * v + i] = cdf[i];
        l.cond_int[v] = fi; marg.push_back(fi);
    }
    float mi;
    hm::distribution1d(marg, cdf, mi);
    l.marg_func[0] = marg[0]; l.marg_func[1] = marg[1];
    for (int i = 0; i < 3; i++) l.marg_cdf[i] = cdf[i];
    l.marg_int = mi;
    s->infinite_lights.push_back((uint32_t)s->lights.size());
    s->lights.push_back(l);
    s->uploaded = false;
    return PBRT_HIP_OK;
    });
}
int pbrt_hip_add_light_distant(PbrtHipScene* s, const float L[3], const float w[3]) {
    return ph_guard(s, "pbrt_hip_add_light_distant", [&]() -> int {
    if (!s || !L || !w) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_light_distant: null argument");
    LightRec l{}; l.type = PH_L_DISTANT; l.prim = 0xFFFFFFFFu;
    for (int c = 0; c < 3; c++) { l.L[c] = L[c]; l.v[c] = w[c]; }
    s->lights.push_back(l); s->uploaded = false;
    return PBRT_HIP_OK;
    });
}
int pbrt_hip_add_light_point(PbrtHipScene* s, const float I[3], const float p[3]) {
    return ph_guard(s, "pbrt_hip_add_light_point", [&]() -> int {
    if (!s || !I || !p) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_light_point: null argument");
    LightRec l{}; l.type = PH_L_POINT; l.prim = 0xFFFFFFFFu;
    for (int c = 0; c < 3; c++) { l.L[c] = I[c]; l.v[c] = p[c]; }
    s->lights.push_back(l); s->uploaded = false;
    return PBRT_HIP_OK;
    });
}
int pbrt_hip_add_light_spot(PbrtHipScene* s, const float I[3], const float l2w[16], const float w2l[16], float cos_total_width, float cos_falloff_start) {  // spot.rs:27-50
    return ph_guard(s, "pbrt_hip_add_light_spot", [&]() -> int {
    if (!s || !I || !l2w || !w2l) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_light_spot: null argument");
    LightRec l{}; l.type = PH_L_SPOT; l.prim = 0xFFFFFFFFu;
    for (int c = 0; c < 3; c++) l.L[c] = I[c];
    std::memcpy(l.l2w, l2w, 48); std::memcpy(l.w2l, w2l, 48);
    // p_light = light_to_world.transform_point(Point3f::ZERO) (transform.rs:288-302)
    const float xp = l2w[0] * 0.0f + l2w[1] * 0.0f + l2w[2] * 0.0f + l2w[3], yp = l2w[4] * 0.0f + l2w[5] * 0.0f + l2w[6] * 0.0f + l2w[7];
    const float zp = l2w[8] * 0.0f + l2w[9] * 0.0f + l2w[10] * 0.0f + l2w[11], wp = l2w[12] * 0.0f + l2w[13] * 0.0f + l2w[14] * 0.0f + l2w[15];
    if (wp == 1.0f) { l.v[0] = xp; l.v[1] = yp; l.v[2] = zp; } else { const float inv = 1.0f / wp; l.v[0] = inv * xp; l.v[1] = inv * yp; l.v[2] = inv * zp; }
    l.cos_total_width = cos_total_width; l.cos_falloff_start = cos_falloff_start;
    s->lights.push_back(l); s->uploaded = false;
    return PBRT_HIP_OK;
    });
}
int pbrt_hip_add_light_diffuse_area(PbrtHipScene* s, const float L[3], int two_sided, uint32_t n_tris, uint32_t* out_first_id) {
    return ph_guard(s, "pbrt_hip_add_light_diffuse_area", [&]() -> int {
    if (!s || !L) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "add_light_diffuse_area: null argument");
    if (out_first_id) *out_first_id = (uint32_t)s->lights.size();
    for (uint32_t i = 0; i < n_tris; i++) {
        LightRec l{}; l.type = PH_L_AREA; l.two_sided = two_sided ? 1 : 0; l.prim = 0xFFFFFFFFu;
        for (int c = 0; c < 3; c++) l.L[c] = L[c];
        s->lights.push_back(l);
    }
    s->uploaded = false;
    return PBRT_HIP_OK;
    });
}

int pbrt_hip_set_camera_perspective(PbrtHipScene* s, const float r2c[16], const float c2w[16], float lens_radius, float focal_distance,
                                    float shutter_open, float shutter_close) {
    return ph_guard(s, "pbrt_hip_set_camera_perspective", [&]() -> int {
    if (!s || !r2c || !c2w) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "set_camera_perspective: null argument");
    std::memcpy(s->cam.r2c, r2c, 64); std::memcpy(s->cam.c2w, c2w, 64);
    s->cam.lens_radius = lens_radius; s->cam.focal_distance = focal_distance;
    s->cam.shutter_open = shutter_open; s->cam.shutter_close = shutter_close;
    {   // dx_camera / dy_camera (perspective_camera.rs:70-74): raster_to_camera(1,0,0) - raster_to_camera(0,0,0), same for y
        auto pt = [&](float x, float y, float out[3]) {  // Transform::transform_point (transform.rs:288-302)
            const float* m = r2c;
            const float xp = m[0] * x + m[1] * y + m[2] * 0.0f + m[3], yp = m[4] * x + m[5] * y + m[6] * 0.0f + m[7];
            const float zp = m[8] * x + m[9] * y + m[10] * 0.0f + m[11], wp = m[12] * x + m[13] * y + m[14] * 0.0f + m[15];
            if (wp == 1.0f) { out[0] = xp; out[1] = yp; out[2] = zp; }
            else { const float inv = 1.0f / wp; out[0] = inv * xp; out[1] = inv * yp; out[2] = inv * zp; }  // Point3 / f (point3.rs: times the reciprocal)
        };
        float p0[3], px[3], py[3];
        pt(0.0f, 0.0f, p0); pt(1.0f, 0.0f, px); pt(0.0f, 1.0f, py);
        for (int k = 0; k < 3; k++) { s->cam.dx_camera[k] = px[k] - p0[k]; s->cam.dy_camera[k] = py[k] - p0[k]; }
    }
    s->cam.kind = PH_CAM_PERSPECTIVE;
    s->have_camera = true;
    return PBRT_HIP_OK;
    });
}
int pbrt_hip_set_camera_orthographic(PbrtHipScene* s, const float r2c[16], const float c2w[16], float lens_radius, float focal_distance,
                                     float shutter_open, float shutter_close) {  // OrthographicCamera::new (orthographic_camera.rs:39-72)
    return ph_guard(s, "pbrt_hip_set_camera_orthographic", [&]() -> int {
    const int rc = pbrt_hip_set_camera_perspective(s, r2c, c2w, lens_radius, focal_distance, shutter_open, shutter_close);
    if (rc != PBRT_HIP_OK) return rc;
    // dx_camera / dy_camera (:58-64): raster_to_camera.transform_vector((1,0,0)) / ((0,1,0)) (transform.rs:373-380)
    const float* m = r2c;
    const float vx[3] = {1.0f, 0.0f, 0.0f}, vy[3] = {0.0f, 1.0f, 0.0f};
    for (int k = 0; k < 3; k++) {
        s->cam.dx_camera[k] = m[4 * k] * vx[0] + m[4 * k + 1] * vx[1] + m[4 * k + 2] * vx[2];
        s->cam.dy_camera[k] = m[4 * k] * vy[0] + m[4 * k + 1] * vy[1] + m[4 * k + 2] * vy[2];
    }
    s->cam.kind = PH_CAM_ORTHOGRAPHIC;
    return PBRT_HIP_OK;
    });
}

int pbrt_hip_set_camera_environment(PbrtHipScene* s, const float c2w[16], int xres, int yres, float shutter_open, float shutter_close) {  // environment_camera.rs:27-41
    return ph_guard(s, "pbrt_hip_set_camera_environment", [&]() -> int {
    if (!s || !c2w) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "set_camera_environment: null argument");
    if (xres <= 0 || yres <= 0) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "set_camera_environment: the film's full resolution must be positive");
    s->cam = CameraRec{};
    for (int k = 0; k < 4; k++) s->cam.r2c[5 * k] = 1.0f;   // unused by this camera; identity keeps the shared prologue of the ray generator finite
    std::memcpy(s->cam.c2w, c2w, 64);
    s->cam.focal_distance = 1e6f; s->cam.shutter_open = shutter_open; s->cam.shutter_close = shutter_close;
    s->cam.full_res[0] = (float)xres; s->cam.full_res[1] = (float)yres;
    s->cam.kind = PH_CAM_ENVIRONMENT;
    s->have_camera = true;
    return PBRT_HIP_OK;
    });
}

int pbrt_hip_set_film(PbrtHipScene* s, int xres, int yres, const int crop[4], const float radius[2], const float table[256], float scale,
                      float max_lum) {
    return ph_guard(s, "pbrt_hip_set_film", [&]() -> int {
    if (!s || !crop || !radius || !table) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "set_film: null argument");
    if (!(radius[0] > 0.0f) || !(radius[1] > 0.0f)) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "set_film: filter radius must be positive");
    FilmRec& f = s->film;
    f.xres = xres; f.yres = yres;
    for (int i = 0; i < 4; i++) f.crop[i] = crop[i];
    f.radius[0] = radius[0]; f.radius[1] = radius[1];
    f.inv_radius[0] = 1.0f / radius[0]; f.inv_radius[1] = 1.0f / radius[1];  // film_tile.rs:50
    f.scale = scale; f.max_lum = max_lum;
    std::memcpy(f.table, table, sizeof(f.table));
    s->have_film = true;
    return PBRT_HIP_OK;
    });
}

static void ext_gcd(uint64_t a, uint64_t b, int64_t& x, int64_t& y) {  // halton.rs:294-302
    if (b == 0) { x = 1; y = 0; return; }
    int64_t d = (int64_t)(a / b), xp, yp;
    ext_gcd(b, a % b, xp, yp);
    x = yp; y = xp - (d * yp);
}
static uint64_t mult_inverse(int64_t a, int64_t n) {  // halton.rs:304-311
    int64_t x, y;
    ext_gcd((uint64_t)a, (uint64_t)n, x, y);
    int64_t r = x - (x / n) * n;
    return (uint64_t)(r < 0 ? r + n : r);
}

int pbrt_hip_set_sampler(PbrtHipScene* s, int kind, uint32_t spp, const int sb[4], int at_center) {
    return ph_guard(s, "pbrt_hip_set_sampler", [&]() -> int {
    if (!s || !sb) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "set_sampler: null argument");
    if (kind != 0 && kind != 1) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "set_sampler: only halton (0) and sobol (1) are GPU-friendly (SURVEY §8a-S)");
    if (spp == 0) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "set_sampler: spp must be > 0");
    SamplerRec& r = s->sampler;
    std::memset(&r, 0, sizeof(r));
    r.kind = kind; r.spp = spp; r.at_center = at_center ? 1 : 0;
    for (int i = 0; i < 4; i++) r.bounds[i] = sb[i];
    if (kind == 0) {  // HaltonSampler::new (halton.rs:61-100)
        int res[2] = {sb[2] - sb[0], sb[3] - sb[1]};
        for (int i = 0; i < 2; i++) {
            uint64_t base = i == 0 ? 2 : 3, scale = 1, e = 0;
            while ((int)scale < std::min(res[i], 128)) { scale *= base; e++; }
            r.base_scales[i] = (uint32_t)scale; r.base_exponents[i] = (uint32_t)e;
        }
        r.sample_stride = r.base_scales[0] * r.base_scales[1];
        r.mult_inverse[0] = (uint32_t)mult_inverse(r.base_scales[1], r.base_scales[0]);
        r.mult_inverse[1] = (uint32_t)mult_inverse(r.base_scales[0], r.base_scales[1]);
        if (((uint64_t)spp + 1) * r.sample_stride >= (1ull << 32))
            return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "set_sampler: halton index exceeds 32 bits at this spp/resolution");
    } else {  // SobolSampler::new (sobol.rs:35-63)
        auto pow2_up = [](uint32_t v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; };
        if (spp & (spp - 1)) r.spp = pow2_up(spp);
        int ext = std::max(sb[2] - sb[0], sb[3] - sb[1]);
        r.resolution = (int)pow2_up((uint32_t)std::max(ext, 1));
        int lg = 0; for (uint32_t v = (uint32_t)r.resolution; v >>= 1;) lg++;
        r.log2_resolution = lg;
    }
    s->have_sampler = true;
    return PBRT_HIP_OK;
    });
}

int pbrt_hip_set_sobol_tables(PbrtHipScene* s, const uint32_t* m32, size_t n32, const uint64_t* vdc, const uint64_t* vdc_inv, size_t n_each) {
    return ph_guard(s, "pbrt_hip_set_sobol_tables", [&]() -> int {
    if (!s || !m32 || !vdc || !vdc_inv) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "set_sobol_tables: null argument");
    s->sobol32.assign(m32, m32 + n32); s->vdc.assign(vdc, vdc + n_each); s->vdc_inv.assign(vdc_inv, vdc_inv + n_each);
    s->uploaded = false;
    return PBRT_HIP_OK;
    });
}

int pbrt_hip_build_accel(PbrtHipScene* s, int split_method, int max_prims_in_node) {
    return ph_guard(s, "pbrt_hip_build_accel", [&]() -> int {
    if (!s) return PBRT_HIP_ERR_INVALID_ARG;
    if (split_method == 2) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "build_accel: splitmethod 'middle' panics in the reference (quirk B6, sah.rs:67-76)");
    // every DiffuseAreaLight belongs to a shape (api/src/lib.rs:783-812 creates them per triangle): one that no add_mesh claimed would be sampled
    // through a primitive that does not exist
    for (size_t i = 0; i < s->lights.size(); i++)
        if (s->lights[i].type == PH_L_AREA && s->lights[i].prim == 0xFFFFFFFFu)
            return set_err(s, PBRT_HIP_ERR_STATE, "build_accel: area light " + std::to_string(i) + " was never attached to a mesh (pbrt_hip_add_mesh first_area_light_id)");
    // material classes (shade-side sorting key): materials with the same sequence of lobe kinds share a class; at most 7 classes
    {
        std::vector<uint64_t> sigs;
        for (MaterialRec& m : s->materials) {
            uint64_t sig = 1;
            for (uint32_t k = 0; k < m.n_lobes; k++) sig = sig * 8 + (s->lobes[m.lobe_base + k].kind + 1);
            size_t c = 0;
            while (c < sigs.size() && sigs[c] != sig) c++;
            if (c == sigs.size()) sigs.push_back(sig);
            static const bool no_sort = std::getenv("PBRT_HIP_NO_MATERIAL_SORT") != nullptr;  // measurement aid: every material in one class
            m.sort_class = no_sort ? 0u : (uint32_t)std::min<size_t>(c, 6);
            if (m.none) m.sort_class = 7u;
        }
    }
    if (s->tree_dev_tris) { PH_CHECK(s, hipSetDevice(s->device)); free_tree_dev(s); }   // a rebuild: the tree an earlier device build left there goes first
    std::vector<uint32_t> build_flags(s->tri_flags);
    for (size_t t = 0; t < build_flags.size(); t++) {   // shade-side sorting keys travel in the TriRec: the material's class and its id (scene_types.h)
        const uint32_t mat = s->meshes[s->tri_mesh[t]].material;
        build_flags[t] |= (s->materials[mat].sort_class << PH_TRI_CLASS_SHIFT) | (std::min<uint32_t>(mat, 0xFFFu) << PH_TRI_MAT_SHIFT);
    }
    phost::BuildInput in;
    in.P = s->P.data(); in.idx = s->idx.data(); in.n_tris = s->idx.size() / 3; in.tri_flags = build_flags.data(); in.tri_mesh = s->tri_mesh.data();
    auto fail = [&](int brc) {
        if (brc == -2) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "build_accel: the reference's HLBVH build asserts on this input (hlbvh.rs:338/356/418)");
        return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "build_accel: bad arguments");
    };
    s->inst_recs.clear();
    if (s->objects.empty()) {
        int brc;
        if (s->build_on_device) {   // pbrt_hip_build_accel_device: the same tree, made by kernels (HLBVH: bvh_device.hip, SAH: bvh_sah_device.hip)
            PH_CHECK(s, hipSetDevice(s->device));
            std::string e;
            if (split_method == 1) brc = phost::build_hlbvh_device(in, max_prims_in_node, s->stream, s->bvh, e);
            else {
                brc = phost::build_sah_device(in, max_prims_in_node, s->stream, s->bvh, e, &s->tree_dev_nodes, &s->tree_dev_tris);
                if (brc == 0 && s->tree_dev_tris) s->tree_dev_n_tris = in.n_tris;
            }
            if (brc == -1) return set_err(s, PBRT_HIP_ERR_DEVICE, "build_accel_device: " + e);
        } else brc = phost::build_bvh(in, split_method, max_prims_in_node, 0, s->bvh);
        if (brc != 0) return fail(brc);
    } else {
        if (s->open_object >= 0) return set_err(s, PBRT_HIP_ERR_STATE, "build_accel: an object definition is still open (missing ObjectEnd)");
        if (s->build_on_device && split_method != 0) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "build_accel_device: scenes with object instances are built on the device with the SAH method only");
        // One aggregate per instanced object (make_accelerator at ObjectInstance time, lib.rs:953-971) and the scene's own over its triangles and TransformedPrimitives,
        // in one node array and one TriRec array: [scene | objects] (bvh_build.h)
        std::vector<uint32_t> tri0(s->objects.size()), tri1(s->objects.size()), inst_object(s->instances.size());
        std::vector<float> inst_i2w(16 * s->instances.size());
        for (size_t k = 0; k < s->objects.size(); k++) { tri0[k] = s->objects[k].tri0; tri1[k] = s->objects[k].tri1; }
        for (size_t k = 0; k < s->instances.size(); k++) { inst_object[k] = s->instances[k].object; std::memcpy(&inst_i2w[16 * k], s->instances[k].i2w, 64); }
        const phost::InstancedScene isc{tri0.data(), tri1.data(), tri0.size(), inst_object.data(), inst_i2w.data(), inst_object.size(), s->top_items.data(), s->top_items.size()};
        phost::ForestLayout layout;
        phost::forest_layout(isc, layout);
        // bit 30 of a leaf reference is PH_LEAF_INST_HINT in an instanced scene: the forest's records must stay below it (both builders)
        if (layout.items.size() >= 0x40000000ull) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "build_accel: an instanced scene may hold at most 2^30 leaf records");
        std::vector<phost::ForestTreeOut> trees;
        if (s->build_on_device) {
            PH_CHECK(s, hipSetDevice(s->device));
            std::string e;
            const phost::ForestSpec spec = layout.spec(isc);
            const int brc = phost::build_sah_device(in, max_prims_in_node, s->stream, s->bvh, e, &s->tree_dev_nodes, &s->tree_dev_tris, &spec, &trees);
            if (brc == -1) return set_err(s, PBRT_HIP_ERR_DEVICE, "build_accel_device: " + e);
            if (brc != 0) return fail(brc);
            if (s->tree_dev_tris) s->tree_dev_n_tris = layout.items.size();
        } else {
            const int brc = phost::build_forest_host(in, isc, layout, split_method, max_prims_in_node, s->bvh, trees);
            if (brc != 0) return fail(brc);
        }
        for (size_t k = 0; k < s->instances.size(); k++) {
            const PbrtHipScene::InstanceHost& ih = s->instances[k];
            const phost::ForestTreeOut& b = trees[layout.inst_tree[k]];
            InstRec r{};
            std::memcpy(r.w2i, ih.w2i, 64); std::memcpy(r.i2w, ih.i2w, 64);
            for (int a = 0; a < 3; a++) { r.lo[a] = b.lo[a]; r.hi[a] = b.hi[a]; }
            r.root_ref = b.root_ref; r.flags = b.n_items == 1 ? PH_INST_SINGLE : 0u;   // an object with one primitive is used directly, without an aggregate (lib.rs:953-956)
            bool ident = true;
            for (int a = 0; a < 16; a++) if (ih.i2w[a] != ((a % 5 == 0) ? 1.0f : 0.0f)) ident = false;
            if (ident) r.flags |= PH_INST_IDENTITY;
            s->inst_recs.push_back(r);
        }
    }
    // Light::preprocess: bounding sphere of the world bound (bounds3.rs:196-208; infinite.rs:113-117, distant.rs:54-58)
    s->world_radius = 1.0f; s->world_center[0] = s->world_center[1] = s->world_center[2] = 0.0f;
    if (s->bvh.root_ref != PH_INVALID_REF) {
        const float* lo = s->bvh.root_lo; const float* hi = s->bvh.root_hi;
        float c[3];
        for (int k = 0; k < 3; k++) c[k] = (1.0f - 0.5f) * lo[k] + 0.5f * hi[k];  // lerp(0.5, pmin, pmax) (common.rs:166-175)
        bool inside = (c[0] >= lo[0] && c[0] <= hi[0]) && (c[1] >= lo[1] && c[1] <= hi[1]) && (c[2] >= lo[2] && c[2] <= hi[2]);
        float dx = c[0] - hi[0], dy = c[1] - hi[1], dz = c[2] - hi[2];
        s->world_radius = inside ? std::sqrt(dx * dx + dy * dy + dz * dz) : 0.0f;
        for (int k = 0; k < 3; k++) s->world_center[k] = c[k];
    }
    s->built = true; s->uploaded = false;
    return PBRT_HIP_OK;
    });
}

// BVHAccel::from with splitmethod "sah" (the default) or "hlbvh", constructed on the GPU (bvh_sah_device.hip, bvh_device.hip): same tree, same leaf order as pbrt_hip_build_accel
int pbrt_hip_build_accel_device(PbrtHipScene* s, int split_method, int max_prims_in_node) {
    return ph_guard(s, "pbrt_hip_build_accel_device", [&]() -> int {
    if (!s) return PBRT_HIP_ERR_INVALID_ARG;
    if (split_method != 0 && split_method != 1) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "build_accel_device: the device builders make the SAH (0) and the HLBVH (1) tree; EqualCounts is built on the host");
    s->build_on_device = true;
    const int rc = pbrt_hip_build_accel(s, split_method, max_prims_in_node);
    s->build_on_device = false;
    return rc;
    });
}

int pbrt_hip_world_bound(const PbrtHipScene* s, float out[6]) {
    return ph_guard(const_cast<PbrtHipScene*>(s), "pbrt_hip_world_bound", [&]() -> int {
    if (!s || !out) return PBRT_HIP_ERR_INVALID_ARG;
    if (!s->built) return PBRT_HIP_ERR_STATE;
    for (int k = 0; k < 3; k++) { out[k] = s->bvh.root_lo[k]; out[3 + k] = s->bvh.root_hi[k]; }
    return PBRT_HIP_OK;
    });
}

// measurement aid: sizes of the built acceleration structure in the device layout (bvh/common.rs:8-23 keeps the same tallies as statistics)
int pbrt_hip_accel_stats(const PbrtHipScene* s, uint64_t out[8]) {
    return ph_guard(const_cast<PbrtHipScene*>(s), "pbrt_hip_accel_stats", [&]() -> int {
    if (!s || !out) return PBRT_HIP_ERR_INVALID_ARG;
    if (!s->built) return PBRT_HIP_ERR_STATE;
    out[0] = s->tree_dev_tris ? s->bvh.interior_nodes : s->bvh.nodes.size(); out[1] = s->tree_dev_tris ? s->tree_dev_n_tris : s->bvh.tris.size();
    out[2] = out[0] * sizeof(Node64); out[3] = out[1] * sizeof(TriRec);
    out[4] = s->bvh.leaf_nodes; out[5] = (uint64_t)s->bvh.max_depth; out[6] = s->bvh.max_leaf_prims;
    out[7] = (uint64_t)(s->bvh.build_seconds * 1e6);
    return PBRT_HIP_OK;
    });
}

// test aid: the built structure itself (device layout), wherever it lives
int pbrt_hip_accel_copy(PbrtHipScene* s, void* out_nodes, uint64_t node_capacity, void* out_leaf_records, uint64_t record_capacity) {
    return ph_guard(s, "pbrt_hip_accel_copy", [&]() -> int {
    if (!s) return PBRT_HIP_ERR_INVALID_ARG;
    if (!s->built) return set_err(s, PBRT_HIP_ERR_STATE, "accel_copy: build_accel first");
    const size_t nn = s->tree_dev_tris ? s->bvh.interior_nodes : s->bvh.nodes.size(), nt = s->tree_dev_tris ? s->tree_dev_n_tris : s->bvh.tris.size();
    if (node_capacity < nn || record_capacity < nt) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "accel_copy: buffers too small (pbrt_hip_accel_stats gives the sizes)");
    if (s->tree_dev_tris) {
        PH_CHECK(s, hipSetDevice(s->device));
        if (nn && out_nodes) PH_CHECK(s, hipMemcpy(out_nodes, s->tree_dev_nodes, nn * sizeof(Node64), hipMemcpyDeviceToHost));
        if (nt && out_leaf_records) PH_CHECK(s, hipMemcpy(out_leaf_records, s->tree_dev_tris, nt * sizeof(TriRec), hipMemcpyDeviceToHost));
        if (!s->inst_recs.empty()) {   // an upload may have filled the instance records and set the hints in place (patch_inst_records_kernel): hand out the builder's form
            if (out_nodes) { Node64* nd = static_cast<Node64*>(out_nodes); for (size_t i = 0; i < nn; i++) { if ((nd[i].c0 & PH_LEAF_BIT) && nd[i].c0 < PH_NEED_POP) nd[i].c0 &= ~PH_LEAF_INST_HINT; if ((nd[i].c1 & PH_LEAF_BIT) && nd[i].c1 < PH_NEED_POP) nd[i].c1 &= ~PH_LEAF_INST_HINT; } }
            if (out_leaf_records) { TriRec* tr = static_cast<TriRec*>(out_leaf_records); for (size_t i = 0; i < nt; i++) { tr[i].flags &= ~PH_TRI_NEXT_INST; if (tr[i].flags & PH_TRI_INSTANCE) { const uint32_t prim = tr[i].prim, fl = tr[i].flags; std::memset(&tr[i], 0, sizeof(TriRec)); tr[i].prim = prim; tr[i].flags = fl; } } }
        }
    } else {
        if (nn && out_nodes) std::memcpy(out_nodes, s->bvh.nodes.data(), nn * sizeof(Node64));
        if (nt && out_leaf_records) std::memcpy(out_leaf_records, s->bvh.tris.data(), nt * sizeof(TriRec));
    }
    return PBRT_HIP_OK;
    });
}

static int check_error_flag(PbrtHipScene* s) {
    uint32_t flag = 0;
    PH_CHECK(s, hipMemcpy(&flag, s->d_error.p, 4, hipMemcpyDeviceToHost));
    if (flag) {
        (void)hipMemset(s->d_error.p, 0, 4);
        return set_err(s, PBRT_HIP_ERR_DEVICE, "traversal stack exceeded 64 entries (the reference panics here, bvh/mod.rs:185)");
    }
    return PBRT_HIP_OK;
}

static int batch_common(PbrtHipScene* s, bool anyhit, const void* rays, void* out, uint64_t n, bool device_ptrs, float* kernel_ms) {
    if (!s) return PBRT_HIP_ERR_INVALID_ARG;
    if (n && (!rays || !out)) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "batch: null buffer");
    if (!s->built) return set_err(s, PBRT_HIP_ERR_STATE, "batch: call pbrt_hip_build_accel first");
    if (n > 0xFFFF0000ull) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "batch: at most 2^32-65536 rays per call");
    PH_CHECK(s, hipSetDevice(s->device));
    int rc;
    if ((rc = upload_scene(s))) return rc;
    if (n == 0) return PBRT_HIP_OK;
    const size_t out_bytes = anyhit ? (size_t)n : (size_t)n * sizeof(PbrtHipHit);
    const void* d_rays = rays; void* d_out = out;
    if (!device_ptrs) {
        if ((rc = ensure_buf(s, s->d_rays_tmp, (size_t)n * sizeof(PbrtHipRay)))) return rc;
        if ((rc = ensure_buf(s, s->d_out_tmp, out_bytes))) return rc;
        PH_CHECK(s, hipMemcpyAsync(s->d_rays_tmp.p, rays, (size_t)n * sizeof(PbrtHipRay), hipMemcpyHostToDevice, s->stream));
        d_rays = s->d_rays_tmp.p; d_out = s->d_out_tmp.p;
    }
    if ((rc = launch_traverse(s, anyhit, d_rays, d_out, (uint32_t)n, kernel_ms))) return rc;
    if (!device_ptrs) PH_CHECK(s, hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, s->stream));
    PH_CHECK(s, hipStreamSynchronize(s->stream));
    return check_error_flag(s);
}

int pbrt_hip_intersect_batch(PbrtHipScene* s, const PbrtHipRay* rays, PbrtHipHit* hits, uint64_t n) { return ph_guard(s, "pbrt_hip_intersect_batch", [&]() -> int { return batch_common(s, false, rays, hits, n, false, nullptr); }); }
int pbrt_hip_occluded_batch(PbrtHipScene* s, const PbrtHipRay* rays, uint8_t* out, uint64_t n) { return ph_guard(s, "pbrt_hip_occluded_batch", [&]() -> int { return batch_common(s, true, rays, out, n, false, nullptr); }); }
int pbrt_hip_intersect_batch_device(PbrtHipScene* s, const void* d_rays, void* d_hits, uint64_t n, float* ms) { return ph_guard(s, "pbrt_hip_intersect_batch_device", [&]() -> int { return batch_common(s, false, d_rays, d_hits, n, true, ms); }); }
int pbrt_hip_occluded_batch_device(PbrtHipScene* s, const void* d_rays, void* d_occ, uint64_t n, float* ms) { return ph_guard(s, "pbrt_hip_occluded_batch_device", [&]() -> int { return batch_common(s, true, d_rays, d_occ, n, true, ms); }); }

// Roofline bookkeeping: with counting on, every traversal launch also accumulates the work it did.
// out[0..2] closest-hit {interior nodes passed, triangle tests, rays}, out[3..5] any-hit.  Reference-format node visits of
// the closest-hit rays = rays + 2*out[0] (see traverse.h).  Reading resets the counters.
int pbrt_hip_set_traversal_counting(PbrtHipScene* s, int on) {
    return ph_guard(s, "pbrt_hip_set_traversal_counting", [&]() -> int {
    if (!s) return PBRT_HIP_ERR_INVALID_ARG;
    PH_CHECK(s, hipSetDevice(s->device));
    int rc;
    if ((rc = ensure_traversal_workspace(s))) return rc;
    PH_CHECK(s, hipMemset(s->d_counts.p, 0, 64 + 36 * 8));
    s->count_traversal = on != 0;
    return PBRT_HIP_OK;
    });
}
int pbrt_hip_get_traversal_counts(PbrtHipScene* s, uint64_t out[8]) {
    return ph_guard(s, "pbrt_hip_get_traversal_counts", [&]() -> int {
    if (!s || !out) return PBRT_HIP_ERR_INVALID_ARG;
    if (!s->d_counts.p) { for (int i = 0; i < 8; i++) out[i] = 0; return PBRT_HIP_OK; }
    PH_CHECK(s, hipSetDevice(s->device));
    PH_CHECK(s, hipStreamSynchronize(s->stream));
    PH_CHECK(s, hipMemcpy(out, s->d_counts.p, 64, hipMemcpyDeviceToHost));
#if PH_PHASE_CLOCK
    {   // measurement build: the waves' phase clocks since the last call, to stderr
        unsigned long long ph[36];
        PH_CHECK(s, hipMemcpy(ph, (const char*)s->d_counts.p + 64, sizeof ph, hipMemcpyDeviceToHost));
        static const char* names[10] = {"refill", "node_steps", "instance_entry", "triangle_test", "alpha_test", "instance_exit", "retire", "whole_loop", "stack_pops", "leaf_step_whole"};
        for (int k = 0; k < 10; k++)
            std::fprintf(stderr, "PHASE_CLOCK %-15s cycles %14llu (%5.1f %% of the loop)  executions %12llu  mean active lanes %5.1f\n", names[k], ph[k], ph[7] ? 100.0 * (double)ph[k] / (double)ph[7] : 0.0,
                         ph[12 + k], ph[12 + k] ? (double)ph[24 + k] / (double)ph[12 + k] : 0.0);
    }
    PH_CHECK(s, hipMemset(s->d_counts.p, 0, 64 + 36 * 8));
#else
    PH_CHECK(s, hipMemset(s->d_counts.p, 0, 64 + 36 * 8));
#endif
    return PBRT_HIP_OK;
    });
}

// Film::get_pixel_rgb (core/src/film/mod.rs:392-417); splat is identically zero for the path integrator (quirk B3 kept)
int pbrt_hip_film_to_rgb(const PbrtHipScene* s, const float* xyz, const float* weight, float* out_rgb) {
    return ph_guard(const_cast<PbrtHipScene*>(s), "pbrt_hip_film_to_rgb", [&]() -> int {
    if (!s || !xyz || !weight || !out_rgb) return PBRT_HIP_ERR_INVALID_ARG;
    if (!s->have_film) return PBRT_HIP_ERR_STATE;
    const FilmRec& f = s->film;
    const int w = f.crop[2] - f.crop[0], h = f.crop[3] - f.crop[1];
    const size_t n = (size_t)std::max(w, 0) * (size_t)std::max(h, 0);
    auto to_rgb = [](const float* x, float* r) {  // xyz_to_rgb (spectrum/common.rs:337-343)
        r[0] = 3.240479f * x[0] - 1.537150f * x[1] - 0.498535f * x[2];
        r[1] = -0.969256f * x[0] + 1.875991f * x[1] + 0.041556f * x[2];
        r[2] = 0.055648f * x[0] - 0.204043f * x[1] + 1.057311f * x[2];
    };
    const float zero[3] = {0, 0, 0};
    float splat_rgb[3];
    to_rgb(zero, splat_rgb);
    for (size_t i = 0; i < n; i++) {
        float rgb[3];
        to_rgb(xyz + 3 * i, rgb);
        for (int c = 0; c < 3; c++) {
            float v = rgb[c];
            if (weight[i] != 0.0f) { float inv = 1.0f / weight[i]; float q = v * inv; v = 0.0f > q ? 0.0f : q; }
            v += 1.0f * splat_rgb[0];
            v *= f.scale;
            out_rgb[3 * i + c] = v;
        }
    }
    return PBRT_HIP_OK;
    });
}

}  // extern "C"
