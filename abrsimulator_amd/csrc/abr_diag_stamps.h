// abr_diag_stamps.h -- in-kernel cycle stamps of the role-split kernels, DIAGNOSTIC BUILDS ONLY.
//
// The product library (libabr_hip.so) is built without ABR_SPLIT_STAMPS: every macro below is then
// empty and nothing of this file reaches its code object.  `make libabr_hip_stamps.so` defines it;
// tools/gpu_stamps.py reads the accumulators through abr_debug_read_stamps
// (profiles/r0*_role_stamps*.txt).
#ifndef ABR_DIAG_STAMPS_H
#define ABR_DIAG_STAMPS_H

#ifdef ABR_SPLIT_STAMPS
// cycle accumulators per code region.  Lane 0 of each wave adds the
// cycles since that wave's previous stamp to region n, in LDS; the totals go to global memory
// once, at the end of the kernel (ABR_STAMP_FLUSH); read with abr_debug_read_stamps.
__device__ unsigned long long g_st_acc[32];
__shared__ unsigned long long g_sh_st[3][33];
#define ABR_STAMP(n)                                                                           \
    do {                                                                                       \
        if ((threadIdx.x & 63) == 0) {                                                         \
            const unsigned long long t_ = (unsigned long long)__builtin_amdgcn_s_memtime();    \
            const unsigned w_ = (threadIdx.x >> 6) % 3;                                        \
            g_sh_st[w_][n] += t_ - g_sh_st[w_][32];                                            \
            g_sh_st[w_][32] = t_;                                                              \
        }                                                                                      \
    } while (0)
#define ABR_STAMP_INIT()                                                                       \
    do {                                                                                       \
        if ((threadIdx.x & 63) == 0) {                                                         \
            for (int q_ = 0; q_ < 32; q_++) g_sh_st[(threadIdx.x >> 6) % 3][q_] = 0;           \
            g_sh_st[(threadIdx.x >> 6) % 3][32] = (unsigned long long)__builtin_amdgcn_s_memtime(); \
        }                                                                                      \
    } while (0)
#define ABR_STAMP_FLUSH()                                                                      \
    do {                                                                                       \
        if ((threadIdx.x & 63) == 0)                                                           \
            for (int q_ = 0; q_ < 32; q_++)                                                    \
                if (g_sh_st[(threadIdx.x >> 6) % 3][q_])                                       \
                    atomicAdd(&g_st_acc[q_], g_sh_st[(threadIdx.x >> 6) % 3][q_]);             \
    } while (0)
#else
#define ABR_STAMP_INIT()
#define ABR_STAMP_FLUSH()
#endif

#endif
