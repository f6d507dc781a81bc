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
__device__ unsigned long long g_st_acc_xcd[8][32];     // the same, per XCD (HW_REG_XCC_ID): which region differs where
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
        unsigned xcc_f_;                                                                       \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_f_));                  \
        if ((threadIdx.x & 63) == 0)                                                           \
            for (int q_ = 0; q_ < 32; q_++)                                                    \
                if (g_sh_st[(threadIdx.x >> 6) % 3][q_]) {                                     \
                    atomicAdd(&g_st_acc[q_], g_sh_st[(threadIdx.x >> 6) % 3][q_]);             \
                    atomicAdd(&g_st_acc_xcd[xcc_f_ & 7][q_], g_sh_st[(threadIdx.x >> 6) % 3][q_]); \
                }                                                                              \
    } while (0)
// when each role of each workgroup began and ended (s_memtime): [workgroup][0 = begin, 1 = D end, 2 = P end, 3 = S end]
// [4 + role]: where that role's wave ran: HW_REG_XCC_ID << 16 | HW_REG_HW_ID (simd_id[5:4] cu_id[11:8] sh_id[12] se_id[15:13])
__device__ unsigned long long g_wg_t[4096][10];    // [8], [9]: begin / S end on the constant 100 MHz clock (s_memrealtime)
#define ABR_WG_WHERE(role)                                                                     \
    do {                                                                                       \
        unsigned hw_, xcc_;                                                                    \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                      \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                    \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096)                                      \
            g_wg_t[blockIdx.x][4 + (role)] = ((unsigned long long)(xcc_ & 15) << 16) | (hw_ & 0xffff); \
    } while (0)
#define ABR_WG_TIME(slot)                                                                      \
    do {                                                                                       \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096)                                      \
            g_wg_t[blockIdx.x][slot] = (unsigned long long)__builtin_amdgcn_s_memtime();       \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096 && ((slot) == 0 || (slot) == 3))     \
            g_wg_t[blockIdx.x][(slot) == 0 ? 8 : 9] = (unsigned long long)__builtin_amdgcn_s_memrealtime(); \
    } while (0)
// K3: cycles per phase, lane 0 of every wave; kept in registers, added to the global accumulators (slots 26..30) at the end
#define K3_STAMP_DECL long long k3_t_ = __builtin_amdgcn_s_memtime(); long long k3_d_[5] = {0, 0, 0, 0, 0};
#define K3_STAMP(n)                                                                            \
    do {                                                                                       \
        const long long t_ = __builtin_amdgcn_s_memtime();                                     \
        k3_d_[(n) - 26] = t_ - k3_t_;                                                          \
        k3_t_ = t_;                                                                            \
    } while (0)
#define K3_STAMP_FLUSH()                                                                       \
    do {                                                                                       \
        if ((threadIdx.x & 63) == 0 && (blockIdx.x & 63) == 0)     /* a sample: one workgroup in 64 */ \
            for (int q_ = 0; q_ < 5; q_++) atomicAdd(&g_st_acc[26 + q_], (unsigned long long)k3_d_[q_]); \
    } while (0)
#else
#define ABR_WG_WHERE(role)
#define ABR_WG_TIME(slot)
#define ABR_STAMP_INIT()
#define ABR_STAMP_FLUSH()
#define K3_STAMP_DECL
#define K3_STAMP(n)
#define K3_STAMP_FLUSH()
#endif

#endif
