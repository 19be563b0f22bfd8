// -DPH_PHASE_CLOCK=1 measurement builds (never shipped or timed): a wave clocks the phases of a kernel with s_memtime and adds cycles, executions and active lanes per phase to
// LDS tallies `phc_lds[3 * PHC_N]` the kernel declares and flushes ([phase] cycles, [PHC_N + phase] executions, [2 * PHC_N + phase] active lanes).  No long-lived registers:
// the kernel's occupancy stays what it is.  In a normal build the macros expand to nothing.
#pragma once
#ifndef PH_PHASE_CLOCK
#define PH_PHASE_CLOCK 0
#endif
#define PHC_N 12
#if PH_PHASE_CLOCK && defined(__HIP_DEVICE_COMPILE__)
#define PHC_BEGIN(k) const unsigned long long phc_t##k = __builtin_amdgcn_s_memtime()
#define PHC_END(k)                                                                                                                  \
    do {                                                                                                                            \
        const unsigned long long phc_dt = __builtin_amdgcn_s_memtime() - phc_t##k;                                                  \
        const unsigned long long phc_m = __ballot(true);                                                                            \
        if ((int)(threadIdx.x & 63u) == __ffsll((long long)phc_m) - 1) {                                                            \
            atomicAdd(&phc_lds[k], phc_dt); atomicAdd(&phc_lds[PHC_N + k], 1ull); atomicAdd(&phc_lds[2 * PHC_N + k], (unsigned long long)__popcll(phc_m)); \
        }                                                                                                                           \
    } while (0)
#else
#define PHC_BEGIN(k) do { } while (0)
#define PHC_END(k) do { } while (0)
#endif
