// How fast can a wave fetch one random 64-byte record per lane (a BVH node per ray), in dependent chains as in a tree walk?
//   A  every lane loads its own record with 4 x global_load_dwordx4 (each a separate L1 access: 4 accesses per lane and record)
//   B  quad-cooperative: in round i the four lanes of a quad load the four 16-byte quarters of quad-lane i's record (one 64-byte access per quad); data left where it lands
//   D  as B, then every lane writes the four quarters it received to LDS (4 x ds_write_b128) and reads its own record back (4 x ds_read_b128)
//   W > 0 adds W dependent multiply-adds per record (the box tests of a node step), so that the fetch competes with vector work as it does in the kernel
//   C  as B, but straight into LDS (global_load_lds_dwordx4: wave base + lane x 16) and read back by the owning lane with 4 x ds_read_b128
// hipcc --offload-arch=gfx950 -O2 scripts/calib/node_fetch.hip -o /tmp/node_fetch && /tmp/node_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define BLOCK 256
#define ITER 512
template <int MODE, int W> __global__ __launch_bounds__(BLOCK) void k(const uint4* __restrict__ table, uint32_t mask, uint32_t* out, float active_frac) {
    __shared__ uint4 stage[MODE >= 2 ? BLOCK * 4 : 1];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, q = lane & 3u;
    uint32_t idx = (blockIdx.x * BLOCK + tid) * 2654435761u & mask;
    uint32_t acc = 0;
    float fa = 1.0f, fb = 0.5f;
#define WORK(x) if (W) { float v = __uint_as_float(((x) & 0x007fffffu) | 0x3f800000u); _Pragma("unroll") for (int w = 0; w < W; w += 2) { fa = fa * v + 0.25f; fb = fb * v + fa; } }
    // a share of the lanes sits out (as lanes waiting at a leaf do): their records are not wanted
    const bool active = (float)((lane * 37u + 11u) & 63u) < active_frac * 64.0f;
    for (int it = 0; it < ITER; it++) {
        uint4 r0, r1, r2, r3;
        if (MODE == 0) {
            if (active) {
                const uint4* p = table + (size_t)idx * 4;
                r0 = p[0]; r1 = p[1]; r2 = p[2]; r3 = p[3];
                idx = (r0.x ^ r1.y ^ r2.z ^ r3.w) & mask; acc += r0.y + r1.z + r2.w + r3.x; WORK(r1.x)
            }
        } else {
            // quad-lane i's index to all four lanes of the quad (an inactive lane contributes its stale index: that load is harmless)
            const uint32_t i0 = __builtin_amdgcn_mov_dpp(idx, 0x00, 0xf, 0xf, true), i1 = __builtin_amdgcn_mov_dpp(idx, 0x55, 0xf, 0xf, true),
                           i2 = __builtin_amdgcn_mov_dpp(idx, 0xaa, 0xf, 0xf, true), i3 = __builtin_amdgcn_mov_dpp(idx, 0xff, 0xf, 0xf, true);
            if (MODE == 1) {
                r0 = table[(size_t)i0 * 4 + q]; r1 = table[(size_t)i1 * 4 + q]; r2 = table[(size_t)i2 * 4 + q]; r3 = table[(size_t)i3 * 4 + q];
                // keep the chain dependent on loaded data: this lane's own quarter of its own record
                const uint4 own = q == 0 ? r0 : q == 1 ? r1 : q == 2 ? r2 : r3;
                if (active) { idx = (own.x ^ own.y ^ own.z ^ own.w) & mask; acc += r0.y + r1.z + r2.w + r3.x; WORK(own.x) }
            } else if (MODE == 3) {
                r0 = table[(size_t)i0 * 4 + q]; r1 = table[(size_t)i1 * 4 + q]; r2 = table[(size_t)i2 * 4 + q]; r3 = table[(size_t)i3 * 4 + q];
                uint4* base = stage + wave * 256;   // [round i][quad][quarter]: the lane's quarter of quad-lane i's record
                base[0 * 64 + lane] = r0; base[1 * 64 + lane] = r1; base[2 * 64 + lane] = r2; base[3 * 64 + lane] = r3;
                const uint4* mine = base + q * 64 + (lane & ~3u);
                r0 = mine[0]; r1 = mine[1]; r2 = mine[2]; r3 = mine[3];
                if (active) { idx = (r0.x ^ r1.y ^ r2.z ^ r3.w) & mask; acc += r0.y + r1.z + r2.w + r3.x; WORK(r1.x) }
            } else {
                uint4* base = stage + wave * 256;   // 4 KB per wave: round i's 1 KB = 16 quads x 64 B
                __builtin_amdgcn_global_load_lds(table + (size_t)i0 * 4 + q, (__attribute__((address_space(3))) void*)(base + 0), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(table + (size_t)i1 * 4 + q, (__attribute__((address_space(3))) void*)(base + 64), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(table + (size_t)i2 * 4 + q, (__attribute__((address_space(3))) void*)(base + 128), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(table + (size_t)i3 * 4 + q, (__attribute__((address_space(3))) void*)(base + 192), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // quad-lane q's record was loaded in round q by the quad's four lanes: base + q * 64 + (lane / 4) * 4 .. + 3
                const uint4* mine = base + q * 64 + (lane >> 2) * 4;
                r0 = mine[0]; r1 = mine[1]; r2 = mine[2]; r3 = mine[3];
                if (active) { idx = (r0.x ^ r1.y ^ r2.z ^ r3.w) & mask; acc += r0.y + r1.z + r2.w + r3.x; WORK(r1.x) }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the next round's DMA overwrites the stage
            }
        }
    }
    out[blockIdx.x * BLOCK + tid] = acc + idx + (uint32_t)(fa + fb);
}
template <int MODE, int W> double run(const uint4* d_table, uint32_t mask, uint32_t* d_out, int blocks_per_cu, float frac) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL((k<MODE, W>), dim3(grid), dim3(BLOCK), 0, 0, d_table, mask, d_out, frac);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k<MODE, W>), dim3(grid), dim3(BLOCK), 0, 0, d_table, mask, d_out, frac);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return (double)grid * BLOCK * ITER * 3 * frac / (ms * 1e-3);   // wanted records per second
}
int main() {
    for (uint32_t log2n : {14u, 17u, 21u}) {   // 1 MB (L2-resident), 8 MB, 128 MB (Infinity Cache), 1 GB
        const uint32_t n = 1u << log2n;
        std::vector<uint32_t> h((size_t)n * 16);
        uint64_t s = 88172645463325252ull;
        for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s >> 16); }
        uint4* d_table; uint32_t* d_out;
        if (hipMalloc(&d_table, (size_t)n * 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
        (void)hipMalloc(&d_out, 256 * 8 * BLOCK * 4);
        (void)hipMemcpy(d_table, h.data(), (size_t)n * 64, hipMemcpyHostToDevice);
        for (int bpc : {4, 5, 6}) for (float frac : {1.0f, 0.66f}) {
            const uint32_t m = n - 1;
            printf("table %7u KB  blocks/CU %d  lanes wanting a record %.2f   G records/s:  no work  A %6.1f  B %6.1f  C %6.1f  D %6.1f   | 48 FMAs/record  A %6.1f  B %6.1f  D %6.1f   | 96 FMAs/record  A %6.1f  D %6.1f\n", n / 16, bpc, frac,
                   run<0, 0>(d_table, m, d_out, bpc, frac) * 1e-9, run<1, 0>(d_table, m, d_out, bpc, frac) * 1e-9, run<2, 0>(d_table, m, d_out, bpc, frac) * 1e-9, run<3, 0>(d_table, m, d_out, bpc, frac) * 1e-9,
                   run<0, 48>(d_table, m, d_out, bpc, frac) * 1e-9, run<1, 48>(d_table, m, d_out, bpc, frac) * 1e-9, run<3, 48>(d_table, m, d_out, bpc, frac) * 1e-9,
                   run<0, 96>(d_table, m, d_out, bpc, frac) * 1e-9, run<3, 96>(d_table, m, d_out, bpc, frac) * 1e-9);
        }
        (void)hipFree(d_table); (void)hipFree(d_out);
    }
    return 0;
}
