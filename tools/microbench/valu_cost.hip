// valu_cost.hip -- measured per-instruction cost of the instruction kinds the env kernel
// (K1) is made of, as a function of waves per SIMD.  One number per (kind, dependency
// pattern, waves/SIMD): shader cycles per wave-instruction = d(s_memtime) / count.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off valu_cost.hip -o valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define N_REP 64      // outer loop trips
#define UNROLL 32     // instructions per trip per chain

template <int KIND, int CHAINS>
__global__ void k(double *out, long long *cyc, double c0, int n0) {
    double x[CHAINS];
    int xi[CHAINS];
    for (int j = 0; j < CHAINS; j++) { x[j] = out[threadIdx.x + j]; xi[j] = n0 + j + threadIdx.x; }
    double c = c0;
    __builtin_amdgcn_s_barrier();
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < N_REP; r++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#pragma unroll
            for (int j = 0; j < CHAINS; j++) {
                if (KIND == 0) x[j] = x[j] + c;                                  // v_add_f64
                if (KIND == 1) x[j] = x[j] * c;                                  // v_mul_f64
                if (KIND == 2) x[j] = __builtin_amdgcn_rcp(x[j]);                 // v_rcp_f64
                if (KIND == 3) { xi[j] = (int)x[j]; x[j] = (double)xi[j] * c; }      // cvt, cvt, mul
                if (KIND == 4) xi[j] = xi[j] + (xi[j] >> 3);                      // 2 int ops
                if (KIND == 5) x[j] = (x[j] > c) ? x[j] - c : x[j] + c;           // cmp + 2 cndmask + add/sub
                if (KIND == 6) x[j] = fmax(x[j], c) + c;                          // v_max_f64 + add
                if (KIND == 7) { float f = (float)xi[j]; f = f * 1.0001f + 1.0f; xi[j] = (int)f; } // f32 chain
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; int si = 0;
    for (int j = 0; j < CHAINS; j++) { s += x[j]; si += xi[j]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + si;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND, int CHAINS>
static void run(const char *name, int inst_per_unit, int waves_per_simd) {
    int blocks = 256, threads = 256 * waves_per_simd;   // 256 CUs x 4 SIMDs x waves
    if (threads > 1024) { blocks *= threads / 1024; threads = 1024; }
    double *out; long long *cyc;
    size_t n = (size_t)blocks * threads;
    hipMalloc(&out, (n + 64) * sizeof(double)); hipMalloc(&cyc, (n / 64) * sizeof(long long));
    std::vector<double> h(n + 64, 1.0000001);
    hipMemcpy(out, h.data(), (n + 64) * sizeof(double), hipMemcpyHostToDevice);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((k<KIND, CHAINS>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0000000001, 3);
    hipDeviceSynchronize();
    std::vector<long long> hc(n / 64);
    hipMemcpy(hc.data(), cyc, hc.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : hc) avg += (double)v; avg /= hc.size();
    double per = avg / ((double)N_REP * UNROLL * CHAINS * inst_per_unit);
    printf("%-28s chains=%d waves/SIMD=%d  %.2f cyc per wave-instruction (per wave), %.2f cyc/instr per SIMD\n", name, CHAINS,
           waves_per_simd, per, per / waves_per_simd);
    hipFree(out); hipFree(cyc);
}

#define ALL(KIND, NAME, IPU) \
    for (int w : {1, 2, 4}) { run<KIND, 1>(NAME, IPU, w); run<KIND, 2>(NAME, IPU, w); run<KIND, 4>(NAME, IPU, w); }

int main() {
    ALL(0, "v_add_f64", 1)
    ALL(1, "v_mul_f64", 1)
    ALL(2, "v_rcp_f64", 1)
    ALL(3, "cvt_i32_f64+cvt_f64_i32+mul", 3)
    ALL(4, "int add+shift", 2)
    ALL(5, "cmp_f64+2cndmask+add/sub", 5)
    ALL(6, "v_max_f64+v_add_f64", 2)
    ALL(7, "cvt+f32 mul+add+cvt", 4)
    return 0;
}
