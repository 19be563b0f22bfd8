// Ceiling for the traversal kernel's access pattern: how fast can MI355X serve INDEPENDENT random 64-B records (4 x global_load_dwordx4 per lane, as
// traverse_kernel reads a Node64) from tables of different sizes — L2-resident, Infinity-Cache-resident, HBM-resident — and how fast a DEPENDENT chain
// of such reads (each address computed from the previous record, as a BVH descent does) at the kernel's occupancy of 6 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 gather_rate.hip -o gather_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
__device__ __forceinline__ uint64_t mix(uint64_t x) { x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; return x; }
// every lane reads `steps` records; DEP: the next index depends on the loaded data (tables hold zeros, so the sequence equals the independent one)
template <bool DEP> __global__ __launch_bounds__(256) void gather64(const float4* __restrict__ table, uint64_t n_rec, float* out, uint32_t seed, int steps) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x = mix(gid + seed);
    float s = 0.0f;
    for (int k = 0; k < steps; k++) {
        const float4* p = table + (x % n_rec) * 4;
        const float4 a = p[0], b = p[1], c = p[2], d = p[3];
        const float v = a.x + b.y + c.z + d.w;
        s += v;
        x = mix(x + (DEP ? (uint64_t)__float_as_uint(v) : 0ull) + 1ull);
    }
    if (s == 12345.678f) out[0] = s;  // keep the loads
}
int main() {
    float* o; hipMalloc(&o, 4);
    const size_t sizes_mb[] = {2, 16, 128, 1024, 8192};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    std::printf("table_MiB mode blocks_per_cu GB/s records/us\n");
    for (size_t mb : sizes_mb) {
        const uint64_t n_rec = (mb << 20) / 64;
        float4* t;
        if (hipMalloc(&t, n_rec * 64) != hipSuccess) { std::printf("alloc of %zu MiB failed\n", mb); continue; }
        hipMemset(t, 0, n_rec * 64);
        for (int dep = 0; dep < 2; dep++)
            for (int per_cu : {6, 8}) {
                const int steps = 64;
                const uint32_t blocks = (uint32_t)(cus * per_cu) * 8;  // 8 generations of resident blocks
                for (int rep = 0; rep < 2; rep++) {
                    hipEventRecord(e0);
                    if (dep) hipLaunchKernelGGL(gather64<true>, dim3(blocks), dim3(256), 0, 0, t, n_rec, o, 7919u * rep, steps);
                    else hipLaunchKernelGGL(gather64<false>, dim3(blocks), dim3(256), 0, 0, t, n_rec, o, 7919u * rep, steps);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                    const double recs = (double)blocks * 256 * steps;
                    if (rep == 1) std::printf("%zu %s %d %.1f %.1f\n", mb, dep ? "dependent" : "independent", per_cu, recs * 64 / ms / 1e6, recs / ms / 1e3);
                }
            }
        hipFree(t);
    }
    return 0;
}
