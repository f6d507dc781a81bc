// Where do the waves of a workgroup land?  Every wave of a grid of 192-thread (three-wave) workgroups records
// HW_REG_HW_ID (SIMD / CU / SE ...) and HW_REG_XCC_ID; the host prints, per wave index inside the workgroup,
// how the waves spread over the four SIMDs of a CU, and how many waves of each index share one SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o hwid hwid.hip && ./hwid [workgroups] [threads]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void probe(unsigned *out, int spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // stay resident for a while so that the whole grid is on the chip at once (as the env kernel's is)
    long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        out[2 * w] = hw; out[2 * w + 1] = xcc;
    }
}

int main(int argc, char **argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 1024, threads = argc > 2 ? atoi(argv[2]) : 192;
    const int wpw = threads / 64, waves = wgs * wpw;
    unsigned *d;
    hipMalloc(&d, waves * 8);
    hipLaunchKernelGGL(probe, dim3(wgs), dim3(threads), 0, 0, d, 2000000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(waves * 2);
    hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost);
    // gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]; XCC_ID[3:0]
    std::vector<std::vector<int>> simd_of(wpw, std::vector<int>(4, 0));
    std::map<unsigned, std::vector<int>> per_simd;     // key: (xcc, se, sh, cu, simd) -> count per wave index
    for (int w = 0; w < waves; w++) {
        const unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 15;
        const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        simd_of[w % wpw][simd]++;
        const unsigned key = (xcc << 16) | (se << 12) | (sh << 8) | (cu << 4) | simd;
        auto &v = per_simd[key];
        if (v.empty()) v.assign(wpw, 0);
        v[w % wpw]++;
    }
    printf("%d workgroups x %d waves; distinct SIMDs seen: %zu\n", wgs, wpw, per_simd.size());
    for (int i = 0; i < wpw; i++)
        printf("wave %d of a workgroup -> SIMD0..3: %d %d %d %d\n", i, simd_of[i][0], simd_of[i][1], simd_of[i][2], simd_of[i][3]);
    // how many SIMDs hold k waves of index i
    for (int i = 0; i < wpw; i++) {
        int hist[16] = {0};
        for (auto &kv : per_simd) hist[kv.second[i] < 15 ? kv.second[i] : 15]++;
        printf("SIMDs holding k waves of index %d: ", i);
        for (int k = 0; k < 8; k++) printf("k=%d:%d ", k, hist[k]);
        printf("\n");
    }
    int tot[16] = {0};
    for (auto &kv : per_simd) { int s = 0; for (int v : kv.second) s += v; tot[s < 15 ? s : 15]++; }
    printf("SIMDs holding k waves in total: ");
    for (int k = 0; k < 10; k++) printf("k=%d:%d ", k, tot[k]);
    printf("\n");
    return 0;
}
