// abr_diag_stamps.h -- in-kernel cycle stamps of the role-split kernels, DIAGNOSTIC BUILDS ONLY.
//
// The product library (libabr_hip.so) is built without ABR_SPLIT_STAMPS: every macro below is then
// empty and nothing of this file reaches its code object.  `make libabr_hip_stamps.so` defines it;
// tools/gpu_stamps.py reads the accumulators through abr_debug_read_stamps / abr_debug_stamp_row
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

// per-role work / barrier-wait split of an iteration (two-wave kernel)
#ifdef ABR_SPLIT_STAMPS
#define SPLIT_STAMP_DECL long long st_work = 0, st_wait = 0, st_iters = 0;
#define SPLIT_STAMP_T0 const long long st0 = __builtin_amdgcn_s_memtime();
#define SPLIT_STAMP_T1 const long long st1 = __builtin_amdgcn_s_memtime();
#define SPLIT_STAMP_T2 st_work += st1 - st0; st_wait += __builtin_amdgcn_s_memtime() - st1; st_iters++;
// the unused 4th row of ep_qoe_terms: [work, wait, iterations] of D (slots 0-2) and P (slots 3-5)
#define SPLIT_STAMP_OUT(base)                                                                       \
    if (in_range && l < 3)                                                                           \
        p.ep_qoe_terms[3 * p.n_lanes + (int64_t)blockIdx.x * 64 + l + (base)] =                      \
            (double)(l == 0 ? st_work : (l == 1 ? st_wait : st_iters));
#else
#define SPLIT_STAMP_DECL
#define SPLIT_STAMP_T0
#define SPLIT_STAMP_T1
#define SPLIT_STAMP_T2
#define SPLIT_STAMP_OUT(base)
#endif

#endif
