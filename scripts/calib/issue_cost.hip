// Issue cost of the vector instructions the traversal kernel's node step is made of (gfx950): one wave, then four waves per SIMD, REP x 32 independent instructions
// between two s_memtime reads.  hipcc --offload-arch=gfx950 -O2 scripts/calib/issue_cost.hip -o /tmp/issue_cost && /tmp/issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 2000
#define I8(x) x x x x x x x x
#define I32(x) I8(x) I8(x) I8(x) I8(x)
template <int KIND> __global__ void k(unsigned long long* out, float seed) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {seed, seed + 1}, p1 = {seed + 2, seed + 3}, p2 = {seed + 4, seed + 5}, p3 = {seed + 6, seed + 7};
    unsigned long long m0 = __ballot(threadIdx.x & 1), m1 = __ballot(threadIdx.x & 2);
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < REP; i++) {
        if (KIND == 0) { I8(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
        if (KIND == 1) { I8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p3));) }
        if (KIND == 2) { I8(asm volatile("v_pk_add_f32 %0, %0, %4 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %4 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p3));) }
        if (KIND == 3) { I8(asm volatile("v_cndmask_b32 %0, %0, %4, %5\n v_cndmask_b32 %1, %1, %4, %6\n v_cndmask_b32 %2, %2, %4, %5\n v_cndmask_b32 %3, %3, %4, %6" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "s"(m0), "s"(m1));) }
        if (KIND == 4) { I8(asm volatile("v_cmp_lt_f32 %0, %2, %3\n v_cmp_lt_f32 %1, %3, %2\n v_cmp_lt_f32 %0, %4, %3\n v_cmp_lt_f32 %1, %3, %4" : "+s"(m0), "+s"(m1) : "v"(a0), "v"(a1), "v"(a2));) }
        if (KIND == 5) { I8(asm volatile("v_max3_f32 %0, %0, %4, %5\n v_min3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_min3_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));) }
        if (KIND == 6) { I8(asm volatile("v_cmp_lt_f32 %0, %2, %3\n v_cndmask_b32 %4, %4, %2, %0\n v_cmp_lt_f32 %1, %3, %2\n v_cndmask_b32 %5, %5, %3, %1" : "+s"(m0), "+s"(m1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }   // compare -> dependent select through an SGPR pair
        if (KIND == 7) { I8(asm volatile("s_and_b64 %0, %0, %1\n s_or_b64 %1, %1, %0\n s_andn2_b64 %0, %0, %1\n s_xor_b64 %1, %1, %0" : "+s"(m0), "+s"(m1));) }
        if (KIND == 8) { I8(asm volatile("v_mul_f32 %0, %0, %4\n s_and_b64 %5, %5, %6\n v_mul_f32 %1, %1, %4\n s_or_b64 %6, %6, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "s"(m0), "s"(m1));) }   // vector and scalar interleaved
        if (KIND == 9) { I8(asm volatile("v_sub_f32 %0, %0, %4\n v_mul_f32 %0, %0, %5\n v_sub_f32 %1, %1, %4\n v_mul_f32 %1, %1, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));) }   // dependent pairs
        if (KIND == 10) { I8(asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_mul_f32 %0, %0, %3\n v_pk_add_f32 %1, %1, %2\n v_pk_mul_f32 %1, %1, %3" : "+v"(p0), "+v"(p1) : "v"(p2), "v"(p3));) }   // dependent packed pairs
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (a0 + a1 + a2 + a3 + p0.x + p1.x + p2.y + p3.y == 12345.f || m0 == 77 || m1 == 99) out[1000] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int KIND> void run(const char* name, unsigned long long* d) {
    for (int waves : {1, 4, 8, 16, 24}) {   // waves per CU in ONE block on one CU: 1 -> one wave on one SIMD; 4 -> one per SIMD; 16 -> four per SIMD
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(64 * (waves > 16 ? 16 : waves)), 0, 0, d, 1.5f);
        hipDeviceSynchronize();
        if (waves > 16) continue;
        std::vector<unsigned long long> h(16);
        hipMemcpy(h.data(), d, 16 * 8, hipMemcpyDeviceToHost);
        double mx = 0; for (int i = 0; i < waves; i++) mx = h[i] > mx ? h[i] : mx;
        const double per_simd_waves = waves <= 4 ? 1 : waves / 4.0;
        printf("%-44s waves/SIMD %4.1f: %6.2f ticks per instruction per wave, %6.2f per instruction per SIMD\n", name, per_simd_waves, mx / (REP * 32.0), mx / (REP * 32.0) / per_simd_waves);
    }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 8192 * 8);
    // s_memtime ticks at a constant 100 MHz on this family; convert with the measured v_mul_f32 cost (4 shader cycles per wave64 instruction)
    run<0>("v_mul_f32", d); run<1>("v_pk_mul_f32", d); run<2>("v_pk_add_f32 (op_sel / neg)", d); run<3>("v_cndmask_b32 (SGPR-pair mask)", d); run<4>("v_cmp_lt_f32 -> SGPR pair", d);
    run<5>("v_max3_f32 / v_min3_f32", d); run<6>("v_cmp -> dependent v_cndmask", d); run<7>("s_and_b64 / s_or_b64 (dependent)", d); run<8>("v_mul_f32 + s_and_b64 interleaved", d);
    run<9>("v_sub_f32 -> dependent v_mul_f32", d); run<10>("v_pk_add_f32 -> dependent v_pk_mul_f32", d);
    return 0;
}
