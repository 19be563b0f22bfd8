// Round-3 experiment: the traversal kernel with rays DECOUPLED from lanes.
//
// traverse.h binds a ray to a lane for its whole life: a lane whose ray sits at a leaf idles while the wave walks interior nodes (12 of 64 lanes on average),
// a leaf step runs with the ~25 lanes that happen to be at leaves, a refill's ray set-up with a fifth of the wave (PMC: VALU lane utilisation 0.46).
// Here a ray lives in a SLOT of LDS — origin, 1/d, the triangle test's shear, t_max, its position in the tree, its stack — and every wave owns S > 64 slots.
// Each pass of the loop sorts the wave's slots by what they need next (ballots over the slots' `cur` words) and hands each kind of work to lanes 0 .. n-1:
//   * the node-phase slots walk K interior-node steps in registers (state in, `cur` / stack height out);
//   * once enough slots wait at leaves, they test one triangle each;
//   * finished slots are written out (barycentrics recomputed from the winning triangle: the slot keeps only its index) and refilled from the queue.
// Per ray the sequence of box tests, triangle tests, pushes and pops is exactly traverse.h's (same visit order bvh/mod.rs:206-214, same `t_min < t_max` re-check at pop
// time, same last-wins ties), so the results are the same bits; only who executes a step changes.  Flat scenes only (no INST / ALPHA / COUNT forms).
//
// What it costs: ~120 B of LDS per slot (60 B of state + an 8-entry stack), so a CU holds ~1 150 - 1 300 rays instead of 1 536, in 3 - 4 blocks instead of 6.
#pragma once
#include "traverse.h"

namespace ph {

#define PH_SLOT_FREE 0xFFFFFFFEu   // `cur` of a slot without a ray (PH_INVALID_REF = a finished ray that waits to be written out)
#define PH_SLOT_NOHIT 0xFFFFFFFFu

template <bool ANYHIT, bool MIXED, int S, int D, int K, int LEAF_MIN, int REFILL_MIN, int WPE, int NODE_MIN = 48>
__global__ __launch_bounds__(PH_TRAV_BLOCK) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void traverse_slots_kernel(DeviceScene sc, TravParams p) {
    static_assert(S > 64 && S <= 128, "a lane looks after the slots lane and lane + 64");
    constexpr int W = PH_TRAV_BLOCK / 64;
    __shared__ float sf[W][10][S];       // ox oy oz | 1/dx 1/dy 1/dz | sx sy sz | t_max
    __shared__ uint32_t su[W][5][S];     // packed (dir_is_neg, kx ky kz, any-hit) | cur | sp | ray index | winning TriRec
    __shared__ uint2 sstack[W][D][S];
    __shared__ uint8_t slist[W][128];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const uint64_t lane_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const uint32_t n_first = p.n_ptr ? *p.n_ptr : p.n;
    const uint32_t n_rays = MIXED ? n_first + *p.n2_ptr : n_first;
    const uint32_t gslot0 = (blockIdx.x * W + w) * (uint32_t)S;   // this wave's slots in the spill region
    float (*F)[S] = sf[w]; uint32_t (*U)[S] = su[w]; uint2 (*ST)[S] = sstack[w]; uint8_t* LIST = slist[w];
    enum { F_OX, F_OY, F_OZ, F_IX, F_IY, F_IZ, F_SX, F_SY, F_SZ, F_TMAX };
    enum { U_PACK, U_CUR, U_SP, U_RAY, U_HIT };

    uint32_t batch_next = 0, batch_end = 0;
    bool exhausted = false;
    uint32_t head_cur = p.n_heads > 1u ? blockIdx.x % p.n_heads : 0u, heads_dry = 0u;
    const uint32_t h0 = lane, h1 = lane + 64u;
    const bool own1 = h1 < (uint32_t)S;
    U[U_CUR][h0] = PH_SLOT_FREE;
    if (own1) U[U_CUR][h1] = PH_SLOT_FREE;

    auto push = [&](uint32_t slot, int& sp, uint32_t ref, float tmin) {
        const uint2 e = make_uint2(ref, __float_as_uint(tmin));
        if (sp < D) ST[sp][slot] = e;
        else if (sp < PH_MAX_STACK) p.spill[(size_t)(sp - D) * p.total_threads + gslot0 + slot] = e;   // total_threads = slots of the whole grid here
        else { *p.error_flag = 1u; return; }
        sp++;
    };
    auto pop = [&](uint32_t slot, int& sp, float t_max) -> uint32_t {
        while (sp > 0) {
            sp--;
            uint2 e;
            if (sp < D) { e = ST[sp][slot]; asm volatile("; stack entry from LDS" : "+v"(e.x), "+v"(e.y)); }
            else { e = p.spill[(size_t)(sp - D) * p.total_threads + gslot0 + slot]; asm volatile("; stack entry from the spill region" : "+v"(e.x), "+v"(e.y)); }
            if (__uint_as_float(e.y) < t_max) return e.x;
        }
        return PH_INVALID_REF;
    };

    for (;;) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- what does each slot need? ---------------------------------------------------------------------------------------------------
        const uint32_t c0 = U[U_CUR][h0], c1 = own1 ? U[U_CUR][h1] : 0u;
        const bool free0 = c0 == PH_SLOT_FREE, free1 = own1 && c1 == PH_SLOT_FREE;
        const bool done0 = c0 == PH_INVALID_REF, done1 = own1 && c1 == PH_INVALID_REF;
        const bool leaf0 = !free0 && !done0 && (c0 & PH_LEAF_BIT), leaf1 = own1 && !free1 && !done1 && (c1 & PH_LEAF_BIT);
        const bool node0 = !(c0 & PH_LEAF_BIT), node1 = own1 && !(c1 & PH_LEAF_BIT);
        const uint64_t mf0 = __ballot(free0), mf1 = __ballot(free1), md0 = __ballot(done0), md1 = __ballot(done1);
        const uint64_t ml0 = __ballot(leaf0), ml1 = __ballot(leaf1), mn0 = __ballot(node0), mn1 = __ballot(node1);
        const uint32_t n_free = (uint32_t)(__popcll(mf0) + __popcll(mf1)), n_done = (uint32_t)(__popcll(md0) + __popcll(md1));
        const uint32_t n_leaf = (uint32_t)(__popcll(ml0) + __popcll(ml1)), n_node = (uint32_t)(__popcll(mn0) + __popcll(mn1));
        const uint32_t n_busy = n_leaf + n_node + n_done;

        // ---- refill free slots from the wave's batch ----------------------------------------------------------------------------------------
        if (!(exhausted && batch_next == batch_end) && (n_free >= (uint32_t)REFILL_MIN || n_busy == 0u)) {
            if (batch_next == batch_end) {
                if (p.n_heads > 1u) {
                    uint32_t b = 0;
                    if (lane == 0) b = atomicAdd(p.heads + 16u * head_cur, p.batch);
                    b = __shfl(b, 0);
                    const uint32_t chunk = b / p.head_chunk;
                    const uint64_t q = ((uint64_t)chunk * p.n_heads + head_cur) * p.head_chunk + (b - chunk * p.head_chunk);
                    if (q < (uint64_t)n_rays) { batch_next = (uint32_t)q; batch_end = ((uint32_t)q + p.batch < n_rays) ? (uint32_t)q + p.batch : n_rays; }
                    else { heads_dry++; head_cur = head_cur + 1u == p.n_heads ? 0u : head_cur + 1u; if (heads_dry >= p.n_heads) exhausted = true; }
                } else {
                    uint32_t b = 0;
                    if (lane == 0) b = atomicAdd(p.counter, p.batch);
                    b = __shfl(b, 0);
                    if (b >= n_rays) exhausted = true;
                    else { batch_next = b; batch_end = (b + p.batch < n_rays) ? b + p.batch : n_rays; }
                }
            }
            const uint32_t avail = batch_end - batch_next;
            if (avail) {
                const uint32_t take = avail < n_free ? avail : n_free;
                const uint32_t r0 = (uint32_t)__popcll(mf0 & lane_lt), r1 = (uint32_t)__popcll(mf0) + (uint32_t)__popcll(mf1 & lane_lt);
#pragma unroll 1
                for (int half = 0; half < 2; half++) {
                    const bool mine = half ? (free1 && r1 < take) : (free0 && r0 < take);
                    if (!mine) continue;
                    const uint32_t slot = half ? h1 : h0, q = batch_next + (half ? r1 : r0);
                    const uint32_t ray_index = p.order ? p.order[q] : q;
                    const bool ah = MIXED ? ray_index >= n_first : ANYHIT;
                    const float4* rp = reinterpret_cast<const float4*>((MIXED && ah) ? p.rays2 + (ray_index - n_first) : p.rays + ray_index);
                    const float4 a = rp[0], b = rp[1];
                    RayIn in; in.ox = a.x; in.oy = a.y; in.oz = a.z; in.t_max = a.w; in.dx = b.x; in.dy = b.y; in.dz = b.z; in.time = b.w;
                    RayState r;
                    ray_setup(r, in);
                    uint32_t cur = PH_INVALID_REF;
                    if (sc.root_ref != PH_INVALID_REF) {   // the reference tests nodes[0].bounds first (bvh/mod.rs:189-190)
                        float tmin;
                        const bool h = box_test(r, r.nx ? sc.root_hi[0] : sc.root_lo[0], r.nx ? sc.root_lo[0] : sc.root_hi[0], r.ny ? sc.root_hi[1] : sc.root_lo[1],
                                                r.ny ? sc.root_lo[1] : sc.root_hi[1], r.nz ? sc.root_hi[2] : sc.root_lo[2], r.nz ? sc.root_lo[2] : sc.root_hi[2], tmin);
                        if (h && tmin < r.t_max) cur = sc.root_ref;
                    }
                    F[F_OX][slot] = r.ox; F[F_OY][slot] = r.oy; F[F_OZ][slot] = r.oz; F[F_IX][slot] = r.ix; F[F_IY][slot] = r.iy; F[F_IZ][slot] = r.iz;
                    F[F_SX][slot] = r.sx; F[F_SY][slot] = r.sy; F[F_SZ][slot] = r.sz; F[F_TMAX][slot] = r.t_max;
                    U[U_PACK][slot] = (uint32_t)(r.nx | (r.ny << 1) | (r.nz << 2) | (r.kx << 3) | (r.ky << 5) | (r.kz << 7)) | (ah ? 512u : 0u);
                    U[U_CUR][slot] = cur; U[U_SP][slot] = 0u; U[U_RAY][slot] = ray_index; U[U_HIT][slot] = PH_SLOT_NOHIT;
                }
                batch_next += take;
                continue;   // the slots have changed hands: sort again
            }
        }
        if (n_busy == 0u) {
            if (exhausted && batch_next == batch_end) break;
            continue;
        }

        // ---- ONE kind of work per pass: the one that fills the wave best (a pass costs a sort of the slots, ~40 instructions, whatever it then does) ----------
        // node steps when at least NODE_MIN slots want them, else triangle tests when LEAF_MIN slots wait at leaves, else the write-out of 16 or more finished rays;
        // when nothing reaches its threshold, whatever has the most takers
        int phase;
        if (n_node >= (uint32_t)NODE_MIN) phase = 0;
        else if (n_leaf >= (uint32_t)LEAF_MIN) phase = 1;
        else if (n_done >= 16u) phase = 2;
        else phase = (n_node >= n_leaf && n_node >= n_done) ? 0 : (n_leaf >= n_done ? 1 : 2);

        // ---- K interior-node steps for (up to 64 of) the node-phase slots ---------------------------------------------------------------------
        if (phase == 0) {
            if (node0) LIST[__popcll(mn0 & lane_lt)] = (uint8_t)h0;
            if (node1) LIST[__popcll(mn0) + __popcll(mn1 & lane_lt)] = (uint8_t)h1;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < n_node) {
                const uint32_t slot = LIST[lane];
                RayState r;
                r.ox = F[F_OX][slot]; r.oy = F[F_OY][slot]; r.oz = F[F_OZ][slot]; r.ix = F[F_IX][slot]; r.iy = F[F_IY][slot]; r.iz = F[F_IZ][slot]; r.t_max = F[F_TMAX][slot];
                const uint32_t pk = U[U_PACK][slot];
                r.nx = (int)(pk & 1u); r.ny = (int)((pk >> 1) & 1u); r.nz = (int)((pk >> 2) & 1u);
                uint32_t cur = U[U_CUR][slot];
                int sp = (int)U[U_SP][slot];
#pragma unroll
                for (int step = 0; step < K; step++)
                    if (!(cur & PH_LEAF_BIT)) {   // (PH_INVALID_REF carries the leaf bit too)
                        const float4* np = reinterpret_cast<const float4*>(sc.nodes + cur);
                        const float4 q0 = np[0], q1 = np[1], q2 = np[2];
                        const uint4 q3 = reinterpret_cast<const uint4*>(np)[3];
                        float t0, t1;
                        bool hh0 = box_test(r, r.nx ? q0.y : q0.x, r.nx ? q0.x : q0.y, r.ny ? q0.w : q0.z, r.ny ? q0.z : q0.w, r.nz ? q1.y : q1.x, r.nz ? q1.x : q1.y, t0);
                        bool hh1 = box_test(r, r.nx ? q1.w : q1.z, r.nx ? q1.z : q1.w, r.ny ? q2.y : q2.x, r.ny ? q2.x : q2.y, r.nz ? q2.w : q2.z, r.nz ? q2.z : q2.w, t1);
                        hh0 = hh0 & (t0 < r.t_max);
                        hh1 = hh1 & (t1 < r.t_max);
                        const int neg_axis = q3.z == 0 ? r.nx : (q3.z == 1 ? r.ny : r.nz);
                        const uint32_t near_ref = neg_axis ? q3.y : q3.x, far_ref = neg_axis ? q3.x : q3.y;
                        const bool near_hit = neg_axis ? hh1 : hh0, far_hit = neg_axis ? hh0 : hh1;
                        const float far_t = neg_axis ? t0 : t1;
                        if (near_hit) { cur = near_ref; if (far_hit) push(slot, sp, far_ref, far_t); }
                        else if (far_hit) cur = far_ref;
                        else cur = pop(slot, sp, r.t_max);
                    }
                U[U_CUR][slot] = cur; U[U_SP][slot] = (uint32_t)sp;
            }
        }

        // ---- one triangle for (up to 64 of) the slots that were waiting at a leaf when the pass began ------------------------------------------
        if (phase == 1) {
            if (leaf0) LIST[__popcll(ml0 & lane_lt)] = (uint8_t)h0;
            if (leaf1) LIST[__popcll(ml0) + __popcll(ml1 & lane_lt)] = (uint8_t)h1;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < n_leaf) {
                const uint32_t slot = LIST[lane];
                RayState r;
                r.ox = F[F_OX][slot]; r.oy = F[F_OY][slot]; r.oz = F[F_OZ][slot]; r.sx = F[F_SX][slot]; r.sy = F[F_SY][slot]; r.sz = F[F_SZ][slot]; r.t_max = F[F_TMAX][slot];
                const uint32_t pk = U[U_PACK][slot];
                r.kx = (int)((pk >> 3) & 3u); r.ky = (int)((pk >> 5) & 3u); r.kz = (int)((pk >> 7) & 3u);
                const bool ah = MIXED ? (pk & 512u) != 0u : ANYHIT;
                uint32_t cur = U[U_CUR][slot];
                const uint32_t ti = cur & ~PH_LEAF_BIT;
                const float4* tp = reinterpret_cast<const float4*>(sc.tris + ti);
                const float4 a = tp[0], b = tp[1], c = tp[2];
                const uint32_t flags = __float_as_uint(b.w);
                float t, b0, b1, b2;
                bool occluded = false;
                if (tri_test(r, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), t, b0, b1, b2)) {
                    const uint32_t reject = ah ? (PH_TRI_BOGUS | PH_TRI_ALPHA0 | PH_TRI_SALPHA0) : (PH_TRI_BOGUS | PH_TRI_ALPHA0);
                    if (!(flags & reject)) {
                        U[U_HIT][slot] = ti;
                        if (ah) occluded = true;
                        else { r.t_max = t; F[F_TMAX][slot] = t; }
                    }
                }
                if (occluded) cur = PH_INVALID_REF;
                else if (flags & PH_TRI_LAST) { int sp = (int)U[U_SP][slot]; cur = pop(slot, sp, r.t_max); U[U_SP][slot] = (uint32_t)sp; }
                else cur = cur + 1u;
                U[U_CUR][slot] = cur;
            }
        }

        // ---- write out the rays that had finished when the pass began; their slots are free again -----------------------------------------------
        if (phase == 2) {
            if (done0) LIST[__popcll(md0 & lane_lt)] = (uint8_t)h0;
            if (done1) LIST[__popcll(md0) + __popcll(md1 & lane_lt)] = (uint8_t)h1;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < n_done) {
                const uint32_t slot = LIST[lane];
                const uint32_t pk = U[U_PACK][slot], ray_index = U[U_RAY][slot], ti = U[U_HIT][slot];
                const bool ah = MIXED ? (pk & 512u) != 0u : ANYHIT;
                if (MIXED && ah) p.out2[ray_index - n_first] = ti != PH_SLOT_NOHIT ? 1 : 0;
                else if (!MIXED && ANYHIT) reinterpret_cast<uint8_t*>(p.out)[ray_index] = ti != PH_SLOT_NOHIT ? 1 : 0;
                else {
                    float4* hp = reinterpret_cast<float4*>(reinterpret_cast<HitOut*>(p.out) + ray_index);
                    const float t_max = F[F_TMAX][slot];
                    if (ti == PH_SLOT_NOHIT) {
                        hp[0] = make_float4(t_max, __uint_as_float(0xFFFFFFFFu), 0.0f, 0.0f);
                        hp[1] = make_float4(0.0f, __uint_as_float(0u), __uint_as_float(0u), __uint_as_float(0u));
                    } else {
                        // the barycentrics of the winning triangle: Triangle::intersect's arithmetic again (they do not depend on t_max; the acceptance was decided in the leaf step)
                        RayState r;
                        r.ox = F[F_OX][slot]; r.oy = F[F_OY][slot]; r.oz = F[F_OZ][slot]; r.sx = F[F_SX][slot]; r.sy = F[F_SY][slot]; r.sz = F[F_SZ][slot];
                        r.kx = (int)((pk >> 3) & 3u); r.ky = (int)((pk >> 5) & 3u); r.kz = (int)((pk >> 7) & 3u);
                        r.t_max = __builtin_inff();
                        const float4* tp = reinterpret_cast<const float4*>(sc.tris + ti);
                        const float4 a = tp[0], b = tp[1], c = tp[2];
                        float t = 0.0f, b0 = 0.0f, b1 = 0.0f, b2 = 0.0f;
                        (void)tri_test(r, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), t, b0, b1, b2);
                        hp[0] = make_float4(t_max, a.w, b0, b1);
                        hp[1] = make_float4(b2, __uint_as_float(ti), __uint_as_float(0u), __uint_as_float((__float_as_uint(b.w) >> PH_TRI_CLASS_SHIFT) & 7u));
                    }
                }
                U[U_CUR][slot] = PH_SLOT_FREE;
            }
        }
    }
}

}  // namespace ph
