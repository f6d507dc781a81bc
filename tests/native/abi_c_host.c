/* A plain-C host for the C ABI of include/abr_env.h: no Python, no PyTorch.  Device
 * memory comes from hipMalloc; the program steps 512 lanes through an episode with
 * abr_env_reset / abr_env_step and checks every lane's episode QoE and final clock
 * against the CPU oracle (oracle/libabr_oracle.so, test infrastructure) loaded with
 * dlopen.  Built and run by tests/test_env_gpu.py::test_c_host_program. */
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "abr_env.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_ABR(x) do { int rc_ = (x); if (rc_) { \
    fprintf(stderr, "%s -> %d: %s\n", #x, rc_, abr_last_error()); return 3; } } while (0)

/* mirrors of the oracle's structs (oracle/abr_oracle.c) */
typedef struct { int32_t n_rates, video_length; double chunk_length, max_buffer, start_up_length,
                 interval, wr, wv, ws, wl, speed, ladder[16]; const double *br_table;
                 const double *speed_sched; int32_t speed_rows; int64_t speed_stride; } ocfg;
typedef struct { double qoe, rebuffer_time, start_up_time, average_latency, global_time,
                 buffer_level, play_time; int64_t ticks; int32_t play_id, chunk_id; } ofin;
typedef int64_t (*obatch_fn)(const ocfg *, const double *, const int64_t *, const int32_t *,
                             const int32_t *, const int32_t *, const int32_t *, int32_t, void *,
                             double *, ofin *, int64_t);

static uint32_t lcg(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

int main(int argc, char **argv) {
    const char *oracle_path = argc > 1 ? argv[1] : "oracle/libabr_oracle.so";
    enum { N = 512, V = 10, NT = 5, TL = 700, B = 6 };
    static double traces[NT * TL];
    static int64_t toff[NT];
    static int32_t tlen[NT], tid[N], off[N], acts[N * V];
    uint32_t seed = 12345u;
    for (int t = 0; t < NT; t++) { toff[t] = (int64_t)t * TL; tlen[t] = TL; }
    for (int i = 0; i < NT * TL; i++) traces[i] = (double)(float)(0.2 + (lcg(&seed) % 58000) / 10000.0);
    for (int i = 0; i < N; i++) { tid[i] = i % NT; off[i] = (int32_t)(lcg(&seed) % TL); }
    for (int i = 0; i < N * V; i++) acts[i] = (int32_t)(lcg(&seed) % B);

    abr_env_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    const double ladder[B] = {0.3, 0.75, 1.2, 1.85, 2.85, 4.3};
    cfg.n_rates = B; cfg.video_length = V; cfg.chunk_length = 4.0; cfg.max_buffer = 20.0;
    cfg.start_up_length = 8.0; cfg.interval = 1.0; cfg.rebuffer_weight = 4.3;
    cfg.variance_weight = 1.0; cfg.startup_weight = 1.0; cfg.latency_weight = 0.1; cfg.speed = 1.0;
    for (int r = 0; r < B; r++) cfg.ladder[r] = ladder[r];

    if (abr_abi_version() != ABR_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
    size_t ws_bytes = 0;
    CHECK_ABR(abr_env_workspace_bytes(&cfg, N, &ws_bytes));
    double *d_tr; int64_t *d_off; int32_t *d_len, *d_tid, *d_o, *d_act; void *d_ws; double *d_qoe, *d_f64;
    uint8_t *d_done; float *d_obs, *d_rew;
    CHECK_HIP(hipMalloc((void **)&d_tr, sizeof(traces)));
    CHECK_HIP(hipMalloc((void **)&d_off, sizeof(toff)));
    CHECK_HIP(hipMalloc((void **)&d_len, sizeof(tlen)));
    CHECK_HIP(hipMalloc((void **)&d_tid, sizeof(tid)));
    CHECK_HIP(hipMalloc((void **)&d_o, sizeof(off)));
    CHECK_HIP(hipMalloc((void **)&d_act, sizeof(int32_t) * N));
    CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
    CHECK_HIP(hipMalloc((void **)&d_qoe, sizeof(double) * N));
    CHECK_HIP(hipMalloc((void **)&d_f64, sizeof(double) * ABR_F64_DIM * N));
    CHECK_HIP(hipMalloc((void **)&d_done, N));
    CHECK_HIP(hipMalloc((void **)&d_obs, sizeof(float) * ABR_OBS_DIM * N));
    CHECK_HIP(hipMalloc((void **)&d_rew, sizeof(float) * N));
    CHECK_HIP(hipMemcpy(d_tr, traces, sizeof(traces), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_off, toff, sizeof(toff), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_len, tlen, sizeof(tlen), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_tid, tid, sizeof(tid), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_o, off, sizeof(off), hipMemcpyHostToDevice));

    abr_env *env = NULL;
    CHECK_ABR(abr_env_create(&cfg, d_tr, d_off, d_len, NT, N, d_ws, ws_bytes, NULL, &env));
    CHECK_ABR(abr_env_reset(env, d_tid, d_o, NULL, d_obs, NULL));
    static int32_t col[N];
    for (int s = 0; s < V; s++) {
        for (int i = 0; i < N; i++) col[i] = acts[i * V + s];
        CHECK_HIP(hipMemcpy(d_act, col, sizeof(col), hipMemcpyHostToDevice));
        CHECK_ABR(abr_env_step(env, d_act, d_obs, d_rew, d_done, NULL));
    }
    CHECK_ABR(abr_env_episode_qoe(env, d_qoe, NULL));
    CHECK_ABR(abr_env_observe_f64(env, d_f64, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    static double qoe[N], f64[ABR_F64_DIM * N];
    static uint8_t done[N];
    CHECK_HIP(hipMemcpy(qoe, d_qoe, sizeof(qoe), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(f64, d_f64, sizeof(f64), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(done, d_done, sizeof(done), hipMemcpyDeviceToHost));
    /* error paths of the boundary */
    if (abr_env_step(NULL, d_act, NULL, NULL, NULL, NULL) != ABR_E_INVALID) return 4;
    if (abr_env_create(&cfg, d_tr, d_off, d_len, NT, N, d_ws, 16, NULL, &env) != ABR_E_WORKSPACE) return 4;
    CHECK_ABR(abr_env_destroy(env));

    /* the checker: the CPU oracle */
    void *h = dlopen(oracle_path, RTLD_NOW);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", oracle_path, dlerror()); return 5; }
    obatch_fn obatch = (obatch_fn)dlsym(h, "oracle_env_batch");
    ocfg oc;
    memset(&oc, 0, sizeof(oc));
    oc.n_rates = B; oc.video_length = V; oc.chunk_length = 4.0; oc.max_buffer = 20.0;
    oc.start_up_length = 8.0; oc.interval = 1.0; oc.wr = 4.3; oc.wv = 1.0; oc.ws = 1.0; oc.wl = 0.1;
    oc.speed = 1.0;
    for (int r = 0; r < B; r++) oc.ladder[r] = ladder[r];
    static ofin fin[N];
    if (obatch(&oc, traces, toff, tlen, tid, off, acts, N, NULL, NULL, fin, (int64_t)1 << 40) < 0) return 6;
    int bad = 0;
    for (int i = 0; i < N; i++) {
        if (done[i] != ABR_DONE_EPISODE) bad++;
        if (f64[ABR_F64_GLOBAL_TIME * N + i] != fin[i].global_time) bad++;      /* bit-exact */
        if (f64[ABR_F64_BUFFER_LEVEL * N + i] != fin[i].buffer_level) bad++;
        if (f64[ABR_F64_REBUFFER_TIME * N + i] != fin[i].rebuffer_time) bad++;
        if (fabs(qoe[i] - fin[i].qoe) > 1e-10 * fabs(fin[i].qoe)) bad++;
    }
    printf("abi_c_host: %d lanes x %d steps, mismatches %d, qoe[0]=%.17g\n", N, V, bad, qoe[0]);
    return bad ? 7 : 0;
}
