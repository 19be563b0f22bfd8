// Multi-device handle: one PbrtHipScene driving several GPUs of one node from one process (SURVEY §8b sketch
// `pbrt_hip_scene_create(const int* device_ordinals, int n_devices)`, §8e).
//
// Replaces the reference's tile loop over worker threads (core/src/integrator/sampler_integrator.rs:254-296): the frame's 16x16 tiles are dealt
// round-robin to the devices (tile t of the handle's share -> device (t / tile_parts) % n, the reference's tile enumeration :254-259, :314-336), every device renders
// its tiles with the scene replicated in its own HBM, the per-tile film buffers are gathered on the first device — the path's ONE exchange step, RCCL
// send / receive over xGMI — and merged there in increasing tile index (Film::merge_film_tile, core/src/film/mod.rs:220-248), so the film equals the
// one-device film bit for bit.
//
// The capture calls (add_mesh, add_light_*, set_film ...) act on the handle itself; the other devices' contexts receive a copy of the captured state
// when a render finds it changed, upload it and drop the bulky host arrays again.  RCCL is loaded at the first exchange (dlopen), so single-device use
// of the library does not depend on it.  Device ordinals may repeat: the contexts then share a GPU and the exchange is a device-to-device copy (RCCL
// refuses one device twice in a communicator) — that is how the path is rehearsed on a one-GPU box.
#include "scene_host.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <thread>

struct MultiDevice {
    std::vector<PbrtHipScene*> replicas;   // contexts of devices 1 .. n-1 (device 0 is the handle itself)
    std::vector<DevBuf> gather;            // on device 0: the tile buffers received from devices 1 .. n-1 ([0] unused)
    std::vector<DevBuf> zeros;             // on device 0: zero stand-ins for tile parts this handle does not render
    bool distinct = true;                  // no ordinal twice -> RCCL
    std::vector<ncclComm_t> comms;         // one per device once initialised
    bool replicas_current = false;         // replicas hold the handle's captured state
};

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string load() {  // "" on success
        if (lib) return "";
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return std::string("cannot load RCCL: ") + dlerror();
        auto sym = [&](const char* n) { return dlsym(lib, n); };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart"); GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend"); Recv = (decltype(Recv))sym("ncclRecv"); GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString) { lib = nullptr; return "RCCL library lacks a required symbol"; }
        return "";
    }
};
Rccl g_rccl;
std::mutex g_rccl_mu;

}  // namespace

namespace phost {

#define PH_NCCL(s, call)                                                                                                   \
    do {                                                                                                                   \
        ncclResult_t r__ = (call);                                                                                         \
        if (r__ != ncclSuccess) return set_err((s), PBRT_HIP_ERR_DEVICE, std::string(#call) + ": " + g_rccl.GetErrorString(r__)); \
    } while (0)

void free_multi(PbrtHipScene* s) {
    MultiDevice* m = s->multi;
    if (!m) return;
    for (ncclComm_t c : m->comms) if (c && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c);
    (void)hipSetDevice(s->device);
    for (DevBuf& b : m->gather) if (b.p) (void)hipFree(b.p);
    for (DevBuf& b : m->zeros) if (b.p) (void)hipFree(b.p);
    for (PbrtHipScene* r : m->replicas) { r->multi = nullptr; pbrt_hip_scene_destroy(r); }
    delete m;
    s->multi = nullptr;
}

// after a replica has uploaded: the arrays only upload_scene reads go back to the allocator (a 10 M-triangle scene keeps ~1.4 GB of them per context)
static void drop_bulk(PbrtHipScene* r) {
    auto drop = [](auto& v) { std::remove_reference_t<decltype(v)> e; v.swap(e); };
    drop(r->P); drop(r->N); drop(r->S); drop(r->UV); drop(r->idx); drop(r->tri_mesh); drop(r->tri_flags); drop(r->bvh.nodes); drop(r->bvh.tris);
}

// The handle's captured state -> every replica.  A bulk copy only when something changed since the replicas last uploaded it, and then ONE replica at a time (copy, upload,
// drop the bulky host arrays again): at most one extra copy of the scene is ever held on the host.  A tree that was built on the first device and lives only there
// (pbrt_hip_build_accel_device) goes to the other devices by hipMemcpyPeer — no 730 MB round trip through the host for 10 M triangles.
static int sync_replicas(PbrtHipScene* s) {
    MultiDevice& m = *s->multi;
    const bool bulk = !m.replicas_current || !s->uploaded;
    for (PbrtHipScene* r : m.replicas) {
        if (bulk) {
            m.replicas_current = false;   // until every replica has the new state
            PH_CHECK(s, hipSetDevice(r->device));
            free_tree_dev(r);
            static_cast<SceneHostState&>(*r) = static_cast<const SceneHostState&>(*s);   // (a device-built tree leaves bvh.nodes / bvh.tris empty: nothing bulky to copy there)
            r->uploaded = false; r->light_strategy_uploaded = -1; r->count_traversal = false;
            if (s->tree_dev_tris) {
                const size_t nb = s->bvh.interior_nodes * sizeof(Node64), tb = s->tree_dev_n_tris * sizeof(TriRec);
                if (hipMalloc(&r->tree_dev_nodes, nb ? nb : 16) != hipSuccess || hipMalloc(&r->tree_dev_tris, tb ? tb : 16) != hipSuccess) {
                    (void)hipGetLastError(); free_tree_dev(r);
                    return set_err(s, PBRT_HIP_ERR_OOM, "render: no device memory for the tree on device " + std::to_string(r->device));
                }
                r->tree_dev_n_tris = s->tree_dev_n_tris;
                if (nb) PH_CHECK(s, hipMemcpyPeer(r->tree_dev_nodes, r->device, s->tree_dev_nodes, s->device, nb));
                if (tb) PH_CHECK(s, hipMemcpyPeer(r->tree_dev_tris, r->device, s->tree_dev_tris, s->device, tb));
            }
            const int rc = upload_scene(r);
            if (rc != PBRT_HIP_OK) { s->err = "device " + std::to_string(r->device) + ": " + r->err; return rc; }
            drop_bulk(r);
        } else {  // what set_camera_* / set_film / set_sampler change without touching the uploaded scene
            r->cam = s->cam; r->film = s->film; r->sampler = s->sampler;
            r->have_camera = s->have_camera; r->have_film = s->have_film; r->have_sampler = s->have_sampler; r->built = s->built;
            r->count_traversal = false;
        }
    }
    PH_CHECK(s, hipSetDevice(s->device));
    m.replicas_current = true;
    return PBRT_HIP_OK;
}

static int exchange(PbrtHipScene* s, const std::vector<PbrtHipScene*>& devs, const std::vector<size_t>& floats) {
    MultiDevice& m = *s->multi;
    const int n = (int)devs.size();
    if (n == 1) return PBRT_HIP_OK;
    if (!m.distinct) {  // rehearsal on shared devices: plain device-to-device copies, ordered behind each context's render by its own stream
        for (int k = 1; k < n; k++) {
            PH_CHECK(s, hipSetDevice(devs[k]->device));
            PH_CHECK(s, hipMemcpyAsync(m.gather[k].p, tile_buffer_of(devs[k]).p, floats[k] * 4, hipMemcpyDeviceToDevice, devs[k]->stream));
            PH_CHECK(s, hipStreamSynchronize(devs[k]->stream));
        }
        return PBRT_HIP_OK;
    }
    {
        std::lock_guard<std::mutex> g(g_rccl_mu);
        const std::string e = g_rccl.load();
        if (!e.empty()) return set_err(s, PBRT_HIP_ERR_DEVICE, e);
    }
    if (m.comms.empty()) {
        std::vector<int> ord;
        for (PbrtHipScene* d : devs) ord.push_back(d->device);
        m.comms.assign((size_t)n, nullptr);
        PH_NCCL(s, g_rccl.CommInitAll(m.comms.data(), n, ord.data()));
    }
    // rank k sends its tile buffer to rank 0; rank 0 posts the matching receives: one group, all streams
    PH_NCCL(s, g_rccl.GroupStart());
    for (int k = 1; k < n; k++) {
        PH_NCCL(s, g_rccl.Send(tile_buffer_of(devs[k]).p, floats[k], ncclFloat, 0, m.comms[k], devs[k]->stream));
        PH_NCCL(s, g_rccl.Recv(m.gather[k].p, floats[k], ncclFloat, k, m.comms[0], s->stream));
    }
    PH_NCCL(s, g_rccl.GroupEnd());
    for (int k = 0; k < n; k++) {
        PH_CHECK(s, hipSetDevice(devs[k]->device));
        PH_CHECK(s, hipStreamSynchronize(devs[k]->stream));
    }
    return PBRT_HIP_OK;
}

int render_path_multi(PbrtHipScene* s, int max_depth, float rr_threshold, int light_strategy, const int pixel_bounds[4], int tile_size, int tile_part, int tile_parts,
                      float* out_xyz, float* out_weight, PbrtHipStats* out_stats) {
    MultiDevice& m = *s->multi;
    std::vector<PbrtHipScene*> devs{s};
    devs.insert(devs.end(), m.replicas.begin(), m.replicas.end());
    const int n = (int)devs.size();
    const int parts = tile_parts * n;   // device k renders the tiles t with t % parts == tile_part + tile_parts * k: the handle's share, dealt round-robin
    if (parts > PH_MAX_TILE_PARTS) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "render: tile_parts x devices exceeds " + std::to_string(PH_MAX_TILE_PARTS));
    int rc;
    if ((rc = sync_replicas(s))) return rc;
    std::vector<size_t> floats((size_t)n);
    for (int k = 0; k < n; k++) {
        floats[k] = tile_buffer_floats_for(s, tile_size, tile_part + tile_parts * k, parts);
        PH_CHECK(s, hipSetDevice(devs[k]->device));
        if ((rc = ensure_buf(devs[k], tile_buffer_of(devs[k]), floats[k] * 4))) { s->err = devs[k]->err; return rc; }
    }
    PH_CHECK(s, hipSetDevice(s->device));
    m.gather.resize((size_t)n);
    for (int k = 1; k < n; k++)
        if ((rc = ensure_buf(s, m.gather[k], floats[k] * 4))) return rc;

    // ---- every device renders its tiles, one host thread per device ------------------------------------------------------------------------
    std::vector<int> rcs((size_t)n, PBRT_HIP_OK);
    std::vector<PbrtHipStats> stats((size_t)n);
    auto work = [&](int k) {
        PbrtHipScene* d = devs[k];
        if (hipSetDevice(d->device) != hipSuccess) { rcs[k] = set_err(d, PBRT_HIP_ERR_DEVICE, "hipSetDevice failed"); return; }
        rcs[k] = ph_guard(d, "render (device worker)", [&]() -> int {
            return render_tiles(d, max_depth, rr_threshold, light_strategy, pixel_bounds, tile_size, tile_part + tile_parts * k, parts, tile_buffer_of(d).p, &stats[k]);
        });
    };
    {
        ThreadGroup th;   // (a thread the system cannot start runs its device's share on this one)
        for (int k = 1; k < n; k++) th.run([&work, k]() { work(k); });
        work(0);
        th.join();
    }
    for (int k = 0; k < n; k++)
        if (rcs[k] != PBRT_HIP_OK) { if (k) s->err = "device " + std::to_string(devs[k]->device) + ": " + devs[k]->err; m.replicas_current = false; return rcs[k]; }

    // ---- the exchange step, then Film::merge_film_tile on the first device ------------------------------------------------------------------
    if ((rc = exchange(s, devs, floats))) return rc;
    PH_CHECK(s, hipSetDevice(s->device));
    std::vector<const void*> bufs((size_t)parts, nullptr);
    for (int k = 0; k < n; k++) bufs[(size_t)(tile_part + tile_parts * k)] = k == 0 ? tile_buffer_of(s).p : m.gather[k].p;
    if (tile_parts > 1) {  // a partial frame (this handle is one of several ranks): zeroed stand-ins for the tiles other ranks hold
        m.zeros.resize((size_t)parts);
        for (int p = 0; p < parts; p++) {
            if (bufs[(size_t)p]) continue;
            const size_t fl = tile_buffer_floats_for(s, tile_size, p, parts);
            if (m.zeros[p].bytes < fl * 4) { if ((rc = ensure_buf(s, m.zeros[p], fl * 4))) return rc; PH_CHECK(s, hipMemset(m.zeros[p].p, 0, fl * 4)); }
            bufs[(size_t)p] = m.zeros[p].p;
        }
    }
    if ((rc = merge_tiles(s, tile_size, parts, bufs.data(), out_xyz, out_weight))) return rc;
    if (out_stats) {
        *out_stats = stats[0];
        for (int k = 1; k < n; k++) {
            const PbrtHipStats& t = stats[k];
            out_stats->camera_rays += t.camera_rays; out_stats->regular_rays += t.regular_rays; out_stats->shadow_rays += t.shadow_rays;
            out_stats->paths_zero_radiance += t.paths_zero_radiance; out_stats->paths_total += t.paths_total;
            out_stats->render_seconds = std::max(out_stats->render_seconds, t.render_seconds);   // the devices run side by side
            out_stats->extend_seconds = std::max(out_stats->extend_seconds, t.extend_seconds);
            out_stats->shadow_seconds = std::max(out_stats->shadow_seconds, t.shadow_seconds);
            out_stats->shade_seconds = std::max(out_stats->shade_seconds, t.shade_seconds);
            out_stats->extend_launches = std::max(out_stats->extend_launches, t.extend_launches);
            out_stats->shadow_launches = std::max(out_stats->shadow_launches, t.shadow_launches);
        }
        // "Distributions created" (spatial.rs:17-21) of a one-device render = the voxels any path touched: here the UNION of the voxels the devices filled
        if (out_stats->light_distributions_created || n > 1) {
            std::vector<uint8_t> touched;
            uint64_t count = 0; bool any = false;
            for (int k = 0; k < n; k++) {
                if (!stats[k].light_distributions_created) continue;
                any = true;
                if ((rc = spatial_voxels_touched(devs[k], touched, &count))) { if (k) s->err = devs[k]->err; return rc; }
            }
            if (any) out_stats->light_distributions_created = count;
        }
        PH_CHECK(s, hipSetDevice(s->device));
    }
    return PBRT_HIP_OK;
}

}  // namespace phost

using namespace phost;

extern "C" {

PbrtHipScene* pbrt_hip_scene_create_multi(const int* device_ordinals, int n_devices) {
    return ph_guard_ptr<PbrtHipScene>("pbrt_hip_scene_create_multi", [&]() -> PbrtHipScene* {
    const int visible = pbrt_hip_device_count();
    std::vector<int> ord;
    if (!device_ordinals || n_devices <= 0) { for (int i = 0; i < visible; i++) ord.push_back(i); }
    else ord.assign(device_ordinals, device_ordinals + n_devices);
    if (visible <= 0 || ord.empty() || (int)ord.size() > PH_MAX_TILE_PARTS) {
        set_err(nullptr, PBRT_HIP_ERR_NO_DEVICE, "pbrt_hip_scene_create_multi: no usable HIP device (this library has no CPU path)");
        return nullptr;
    }
    PbrtHipScene* s = pbrt_hip_scene_create(ord[0]);
    if (!s) return nullptr;
    s->multi = new MultiDevice();
    for (size_t k = 1; k < ord.size(); k++) {
        PbrtHipScene* r = pbrt_hip_scene_create(ord[k]);
        if (!r) { pbrt_hip_scene_destroy(s); return nullptr; }
        s->multi->replicas.push_back(r);
        for (size_t j = 0; j < k; j++) if (ord[j] == ord[k]) s->multi->distinct = false;
    }
    (void)hipSetDevice(s->device);
    return s;
    });
}

int pbrt_hip_scene_devices(const PbrtHipScene* s, int* out_ordinals, int capacity) {
    return ph_guard(const_cast<PbrtHipScene*>(s), "pbrt_hip_scene_devices", [&]() -> int {
    if (!s) return PBRT_HIP_ERR_INVALID_ARG;
    const int n = 1 + (s->multi ? (int)s->multi->replicas.size() : 0);
    if (out_ordinals)
        for (int k = 0; k < n && k < capacity; k++) out_ordinals[k] = k == 0 ? s->device : s->multi->replicas[(size_t)k - 1]->device;
    return n;
    });
}

// Self-test of the RCCL binding on whatever devices the handle has: loads the library, builds the communicator (a one-rank communicator on a
// single-device handle) and sends `n_floats` floats from every device to the first one, itself included; returns the number of wrong floats received.
int pbrt_hip_selftest_rccl_gather(PbrtHipScene* s, uint32_t n_floats, uint64_t* out_wrong) {
    return ph_guard(s, "pbrt_hip_selftest_rccl_gather", [&]() -> int {
    if (!s || !out_wrong || n_floats == 0) return set_err(s, PBRT_HIP_ERR_INVALID_ARG, "selftest_rccl_gather: bad argument");
    std::vector<PbrtHipScene*> devs{s};
    if (s->multi) devs.insert(devs.end(), s->multi->replicas.begin(), s->multi->replicas.end());
    if (s->multi && !s->multi->distinct) return set_err(s, PBRT_HIP_ERR_UNSUPPORTED, "selftest_rccl_gather: RCCL refuses one device twice in a communicator");
    {
        std::lock_guard<std::mutex> g(g_rccl_mu);
        const std::string e = g_rccl.load();
        if (!e.empty()) return set_err(s, PBRT_HIP_ERR_DEVICE, e);
    }
    const int n = (int)devs.size();
    std::vector<int> ord;
    for (PbrtHipScene* d : devs) ord.push_back(d->device);
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    PH_NCCL(s, g_rccl.CommInitAll(comms.data(), n, ord.data()));
    std::vector<void*> src((size_t)n, nullptr), dst((size_t)n, nullptr);
    std::vector<float> h(n_floats);
    int rc = PBRT_HIP_OK;
    auto cleanup = [&]() {
        for (int k = 0; k < n; k++) { (void)hipSetDevice(devs[k]->device); if (src[k]) (void)hipFree(src[k]); }
        (void)hipSetDevice(s->device);
        for (int k = 0; k < n; k++) if (dst[k]) (void)hipFree(dst[k]);
        for (ncclComm_t c : comms) if (c) (void)g_rccl.CommDestroy(c);
    };
    for (int k = 0; k < n && rc == PBRT_HIP_OK; k++) {
        for (uint32_t i = 0; i < n_floats; i++) h[i] = (float)(k * 1000003u + i % 65521u);
        if (hipSetDevice(devs[k]->device) != hipSuccess || hipMalloc(&src[k], (size_t)n_floats * 4) != hipSuccess ||
            hipMemcpy(src[k], h.data(), (size_t)n_floats * 4, hipMemcpyHostToDevice) != hipSuccess) rc = set_err(s, PBRT_HIP_ERR_DEVICE, "selftest: source buffer");
        if (rc == PBRT_HIP_OK && (hipSetDevice(s->device) != hipSuccess || hipMalloc(&dst[k], (size_t)n_floats * 4) != hipSuccess ||
                                  hipMemset(dst[k], 0, (size_t)n_floats * 4) != hipSuccess)) rc = set_err(s, PBRT_HIP_ERR_DEVICE, "selftest: destination buffer");
    }
    if (rc != PBRT_HIP_OK) { cleanup(); return rc; }
    ncclResult_t r = g_rccl.GroupStart();
    for (int k = 0; k < n && r == ncclSuccess; k++) {
        r = g_rccl.Send(src[k], n_floats, ncclFloat, 0, comms[k], devs[k]->stream);
        if (r == ncclSuccess) r = g_rccl.Recv(dst[k], n_floats, ncclFloat, k, comms[0], s->stream);
    }
    if (r == ncclSuccess) r = g_rccl.GroupEnd();
    if (r != ncclSuccess) { cleanup(); return set_err(s, PBRT_HIP_ERR_DEVICE, std::string("selftest: RCCL: ") + g_rccl.GetErrorString(r)); }
    for (int k = 0; k < n; k++) { (void)hipSetDevice(devs[k]->device); (void)hipStreamSynchronize(devs[k]->stream); }
    (void)hipSetDevice(s->device);
    uint64_t wrong = 0;
    for (int k = 0; k < n; k++) {
        if (hipMemcpy(h.data(), dst[k], (size_t)n_floats * 4, hipMemcpyDeviceToHost) != hipSuccess) { cleanup(); return set_err(s, PBRT_HIP_ERR_DEVICE, "selftest: read back"); }
        for (uint32_t i = 0; i < n_floats; i++) if (h[i] != (float)(k * 1000003u + i % 65521u)) wrong++;
    }
    cleanup();
    *out_wrong = wrong;
    return PBRT_HIP_OK;
    });
}

}  // extern "C"
