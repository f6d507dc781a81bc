// Does an XCD run PURE ALU work slower than another?  1 024 workgroups x 3 waves (the env kernel's shape: four workgroups per CU,
// three waves per SIMD), every wave a fixed chain of dependent float64 additions (plus an LDS round and a barrier every 256
// additions when BARRIER is set), no global memory inside the timed part.  Reports the mean wall time per XCD (s_memrealtime,
// 100 MHz) and the shader cycles (s_memtime).   hipcc --offload-arch=gfx950 -O2 -o tools/microbench/xcd_alu tools/microbench/xcd_alu.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ __launch_bounds__(192) void k(unsigned long long *out, int n, int barrier, double seed) {
    __shared__ double sh[192];
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    double x = seed + threadIdx.x;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < 256; j++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(seed));
        if (barrier) { sh[threadIdx.x] = x; __syncthreads(); x += sh[(threadIdx.x + 64) % 192]; }
    }
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 3 + (threadIdx.x >> 6);
        out[w * 4 + 0] = xcc & 7; out[w * 4 + 1] = r1 - r0; out[w * 4 + 2] = c1 - c0; out[w * 4 + 3] = (unsigned long long)x;
    }
}
int main() {
    const int W = 1024;
    unsigned long long *out;
    if (hipMalloc(&out, W * 3 * 4 * 8) != hipSuccess) return 1;
    for (int barrier = 0; barrier < 2; barrier++)
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(k, dim3(W), dim3(192), 0, 0, out, 400, barrier, 1.0e-3);
            if (hipDeviceSynchronize() != hipSuccess) return 2;
            std::vector<unsigned long long> h(W * 3 * 4);
            if (hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return 3;
            double t[8] = {0}, c[8] = {0}; int n[8] = {0};
            for (int w = 0; w < W * 3; w++) { int x = (int)h[w * 4]; t[x] += h[w * 4 + 1] * 0.01; c[x] += (double)h[w * 4 + 2]; n[x]++; }
            printf("%s rep %d  wall us per wave by XCD:", barrier ? "adds + LDS + barrier" : "adds only           ", rep);
            for (int x = 0; x < 8; x++) printf(" %7.1f", n[x] ? t[x] / n[x] : 0.0);
            printf("   | GHz:");
            for (int x = 0; x < 8; x++) printf(" %.3f", n[x] ? c[x] / t[x] / 1000.0 : 0.0);
            printf("\n");
        }
    return 0;
}
