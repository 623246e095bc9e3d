/*
 * abi_demo.c -- the C-ABI of include/hrl_envs.h driven from plain C: no Python, no torch, no C++ types across the boundary.
 * TEST INFRASTRUCTURE (tests/test_gpu_envs.py::test_c_abi_from_plain_c builds and runs it on the GPU box and checks its output against the CPU
 * oracle bit for bit); it is also the shape of what a non-Python host would write around the library:
 *   device buffers from hipMalloc, hrl_create / hrl_reset / hrl_step on a stream, results copied back when wanted.
 *
 *   build:  gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I../../include abi_demo.c -L../../hrl_pybullet_envs_amd -lhrl_envs_hip -L/opt/rocm/lib -lamdhip64 -o abi_demo   (tests/c_abi/Makefile)
 *   run:    abi_demo <kind> <n_envs> <steps> <seed> <out.bin>            (a sixth argument: only show that an uninitialised hrl_buffers is refused)
 * Actions are U(-1, 1) from a 64-bit LCG (the one oracle/orc_impl.h::orc_bench uses), generated on the host and copied per step.
 * out.bin: state[N][32] | items[N][stride] | aux[N][4] (as int32) | obs[N][D] | reward[N] | done[N] (u8) | final_obs[N][D] | truncated[N] (u8), raw.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hrl_envs.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_HRL(x) do { int r_ = (x); if (r_ != HRL_OK) { fprintf(stderr, "%s: %d %s\n", #x, r_, hrl_last_error()); return 3; } } while (0)

int main(int argc, char **argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s kind n_envs steps seed out.bin\n", argv[0]); return 1; }
    const int kind = atoi(argv[1]), n = atoi(argv[2]), steps = atoi(argv[3]);
    hrl_config cfg;
    CHECK_HRL(hrl_default_config(kind, &cfg));
    cfg.num_envs = n; cfg.seed = (uint64_t)atoll(argv[4]); cfg.auto_reset = 1; cfg.max_episode_steps = 40;
    const int D = hrl_obs_dim(&cfg), A = hrl_act_dim(&cfg), S = hrl_items_stride(&cfg);
    hrl_handle *h = NULL;
    CHECK_HRL(hrl_create(&cfg, &h));
    hrl_buffers b;
    CHECK_HRL(hrl_buffers_init(&b)); /* zeroes the record and sets struct_size: pointers this host does not know about stay NULL, and the library
                                        refuses a record that was left as the stack held it (`abi_demo ... uninit` below shows that) */
    if (argc > 6) { /* what a host rebuilt against a newer header without initialising the record would hand over */
        hrl_buffers junk;
        memset(&junk, 0x5a, sizeof junk);
        const int rc = hrl_reset(h, &junk, NULL, NULL);
        printf("uninitialised record: rc %d (%s)\n", rc, hrl_last_error());
        return rc == HRL_ERR_BAD_ARG ? 0 : 5;
    }
    float *d_act = NULL;
    CHECK_HIP(hipMalloc((void **)&b.state, sizeof(float) * n * HRL_STATE_STRIDE));
    CHECK_HIP(hipMalloc((void **)&b.items, sizeof(float) * n * S));
    CHECK_HIP(hipMalloc((void **)&b.aux, sizeof(int32_t) * n * HRL_AUX_STRIDE));
    CHECK_HIP(hipMalloc((void **)&d_act, sizeof(float) * n * A));
    CHECK_HIP(hipMalloc((void **)&b.obs, sizeof(float) * n * D));
    CHECK_HIP(hipMalloc((void **)&b.reward, sizeof(float) * n));
    CHECK_HIP(hipMalloc((void **)&b.done, n));
    CHECK_HIP(hipMalloc((void **)&b.info, sizeof(float) * n * HRL_INFO_STRIDE));
    CHECK_HIP(hipMalloc((void **)&b.final_obs, sizeof(float) * n * D));
    CHECK_HIP(hipMalloc((void **)&b.truncated, n));
    b.actions = d_act;
    CHECK_HIP(hipMemset(b.state, 0, sizeof(float) * n * HRL_STATE_STRIDE));
    CHECK_HIP(hipMemset(b.items, 0, sizeof(float) * n * S));
    CHECK_HIP(hipMemset(b.aux, 0, sizeof(int32_t) * n * HRL_AUX_STRIDE));
    CHECK_HIP(hipMemset(b.final_obs, 0, sizeof(float) * n * D));
    CHECK_HIP(hipMemset(b.truncated, 0, n));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    CHECK_HRL(hrl_reset(h, &b, NULL, stream));
    float *act = (float *)malloc(sizeof(float) * n * A);
    uint64_t lcg = 0x9E3779B97F4A7C15ull;
    for (int t = 0; t < steps; ++t) {
        for (int i = 0; i < n * A; ++i) { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; act[i] = (float)((double)(lcg >> 11) * (2.0 / 9007199254740992.0) - 1.0); }
        CHECK_HIP(hipMemcpyAsync(d_act, act, sizeof(float) * n * A, hipMemcpyHostToDevice, stream));
        CHECK_HRL(hrl_step(h, &b, stream));
        CHECK_HIP(hipStreamSynchronize(stream)); /* the host buffer is refilled next: wait for the copy */
    }
    FILE *f = fopen(argv[5], "wb");
    if (!f) return 4;
#define DUMP(ptr, bytes) do { void *t_ = malloc(bytes); CHECK_HIP(hipMemcpy(t_, ptr, bytes, hipMemcpyDeviceToHost)); fwrite(t_, 1, bytes, f); free(t_); } while (0)
    DUMP(b.state, sizeof(float) * n * HRL_STATE_STRIDE);
    DUMP(b.items, sizeof(float) * n * S);
    DUMP(b.aux, sizeof(int32_t) * n * HRL_AUX_STRIDE);
    DUMP(b.obs, sizeof(float) * n * D);
    DUMP(b.reward, sizeof(float) * n);
    DUMP(b.done, (size_t)n);
    DUMP(b.final_obs, sizeof(float) * n * D);
    DUMP(b.truncated, (size_t)n);
    fclose(f);
    printf("%s kind %d: %d envs x %d steps, obs %d act %d items stride %d\n", hrl_backend(), kind, n, steps, D, A, S);
    CHECK_HRL(hrl_destroy(h));
    return 0;
}
