// Scalar-load latency per XCD: how long does one s_load + s_waitcnt take (a) from the kernel-argument segment, (b) from a
// global buffer in device memory, read by every wave over and over -- the pattern of the role-split kernels, which re-read
// their parameter block inside the iteration loop.  One wave per workgroup, 1 024 workgroups (4 per CU like the env kernel).
//   hipcc --offload-arch=gfx950 -O2 -o tools/microbench/smem_latency tools/microbench/smem_latency.hip && tools/microbench/smem_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
struct Big { unsigned v[160]; };          // 640 B: about the env kernels' parameter block
typedef const unsigned __attribute__((address_space(4))) *cu32p;

__global__ __launch_bounds__(64) void k(Big p, const unsigned *gbuf, unsigned long long *out, int reps, int spin) {
    const cu32p ka = (cu32p)__builtin_amdgcn_kernarg_segment_ptr();
    const cu32p ga = (cu32p)gbuf;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long tk = 0, tg = 0;
    unsigned acc = 0;
    for (int r = 0; r < reps; r++) {
        // something else in between, as the roles have (keeps the wave from owning the cache line back to back)
        for (int s = 0; s < spin; s++) asm volatile("v_mov_b32 %0, %0" : "+v"(acc));
        const unsigned off = (unsigned)(r * 7 % 9) * 16;                 // nine different 64-B pieces of the block
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        unsigned a;
        asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(a) : "s"(ka), "s"(off * 4) : "memory");
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        unsigned b;
        asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(b) : "s"(ga), "s"(off * 4) : "memory");
        unsigned long long t2 = __builtin_amdgcn_s_memtime();
        tk += t1 - t0; tg += t2 - t1; acc += a + b;
    }
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = xcc & 7; out[blockIdx.x * 4 + 1] = tk; out[blockIdx.x * 4 + 2] = tg; out[blockIdx.x * 4 + 3] = acc;
    }
}

int main() {
    const int W = 1024, reps = 2000;
    unsigned *gbuf; unsigned long long *out;
    hipMalloc(&gbuf, 4096); hipMemset(gbuf, 1, 4096);
    hipMalloc(&out, W * 4 * sizeof(unsigned long long));
    Big p; for (int i = 0; i < 160; i++) p.v[i] = i;
    for (int spin : {0, 50, 400}) {
        for (int pass = 0; pass < 2; pass++) hipLaunchKernelGGL(k, dim3(W), dim3(64), 0, 0, p, gbuf, out, reps, spin);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(W * 4);
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        double sk[8] = {0}, sg[8] = {0}; int n[8] = {0};
        for (int w = 0; w < W; w++) { int x = (int)h[w * 4]; sk[x] += (double)h[w * 4 + 1] / reps; sg[x] += (double)h[w * 4 + 2] / reps; n[x]++; }
        printf("spin %3d  cycles per s_load + wait (s_memtime ticks incl. ~2 x timer read), per XCD:\n  kernarg:", spin);
        for (int x = 0; x < 8; x++) printf(" %7.1f", n[x] ? sk[x] / n[x] : 0.0);
        printf("\n  global: ");
        for (int x = 0; x < 8; x++) printf(" %7.1f", n[x] ? sg[x] / n[x] : 0.0);
        printf("\n");
    }
    const char *e = getenv("HIP_FORCE_DEV_KERNARG");
    printf("HIP_FORCE_DEV_KERNARG=%s\n", e ? e : "(unset)");
    return 0;
}
