// Calibration of rocprofv3 FETCH_SIZE for THIS library's node access pattern: every lane reads one random, 64-B aligned record
// with 4 x global_load_dwordx4 (as traverse_kernel reads a Node64).  Known bytes = lanes x 64 (records are distinct w.h.p. and the
// 8 GiB table exceeds the 256 MiB Infinity Cache).  Build: hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void gather64(const float4* __restrict__ table, uint64_t n_rec, float* out, uint32_t seed) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x = (gid + seed) * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    const float4* p = table + (x % n_rec) * 4;
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    const float s = a.x + b.y + c.z + d.w;
    if (s == 12345.678f) out[0] = s;  // keep the loads
}
int main() {
    const uint64_t n_rec = (8ull << 30) / 64;
    float4* t; float* o;
    if (hipMalloc(&t, n_rec * 64) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
    hipMemset(t, 0, n_rec * 64);
    const uint32_t lanes = 64u << 20;
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(gather64, dim3(lanes / 256), dim3(256), 0, 0, t, n_rec, o, 7919u * rep);
    hipDeviceSynchronize();
    std::printf("lanes per launch %u, known bytes per launch %llu\n", lanes, (unsigned long long)lanes * 64ull);
    return 0;
}
