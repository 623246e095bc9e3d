// Microbenchmark (diagnostic, not part of the product): how W waves on one SIMD share its issue slots.
// Every wave runs ITERS x 64 repetitions of a pattern in a loop (long enough that all waves of the grid overlap); the grid
// is 256 CUs x W workgroups of 256 threads = W waves per SIMD.  Reported: nanoseconds and s_memtime ticks per repetition per
// wave, and repetitions per microsecond per SIMD -- the aggregate rate, whose ratio between W = 1 and W = 4 says whether four
// dependent chains on a SIMD run side by side (latency bound) or queue for issue slots (issue bound).
//   hipcc --offload-arch=gfx950 -O3 -o simd_share simd_share.hip && ./simd_share
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int P>
__global__ __launch_bounds__(256, 1) void k(unsigned long long *out, float *sink, float a, float b, int iters) {
    float c = a + threadIdx.x, lam = b, lo = -1.f, hi = 1.f, C = 0.001f * a, d0 = a, d1 = b, d2 = a * b, d3 = a - b, t = 0.f, u = 0.f;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {a, b}, p1 = {C, C}, p2 = {lam, lam};
    unsigned long long t0, t1;
    const unsigned long long mask = 1ull << 3;
    int sc = 5;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if (P == 0) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c) : "v"(C), "v"(lam));) }  // dependent fma chain
        if (P == 1) { REP64(asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(d0), "+v"(d1) : "v"(C), "v"(lam));) }  // 2 independent
        if (P == 2) { REP64(asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                                        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(C), "v"(lam));) }  // 4 independent
        if (P == 3) { REP64(asm volatile("v_med3_f32 %2, %0, %4, %5\n\tv_sub_f32 %3, %2, %1\n\ts_cmp_lt_u32 %8, 3\n\tv_readlane_b32 s20, %3, 3\n\t"
                                        "v_cndmask_b32_e64 %1, %1, %2, %7\n\ts_nop 1\n\tv_fmac_f32 %0, s20, %6\n\ts_cbranch_scc1 1f\n1:"
                                        : "+v"(c), "+v"(lam), "+v"(t), "+v"(u) : "v"(lo), "v"(hi), "v"(C), "s"(mask), "s"(sc) : "s20", "scc");) }  // the sweep row as compiled
        if (P == 4) { REP64(asm volatile("v_med3_f32 %2, %0, %4, %5\n\tv_sub_f32 %3, %2, %1\n\tv_readlane_b32 s20, %3, 3\n\t"
                                        "v_cndmask_b32_e64 %1, %1, %2, %7\n\ts_nop 1\n\tv_fmac_f32 %0, s20, %6"
                                        : "+v"(c), "+v"(lam), "+v"(t), "+v"(u) : "v"(lo), "v"(hi), "v"(C), "s"(mask) : "s20");) }  // row without compare / branch
        if (P == 5) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_add_u32 s20, s20, 1" : "+v"(c) : "v"(C), "v"(lam) : "s20", "scc");) }  // dependent fma + a scalar op
        if (P == 6) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(p1), "v"(p2));) }  // dependent packed fma
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = c + lam + d0 + d1 + d2 + d3 + t + u + p0.x + p0.y;
}

template <int P> void run(const char *name, int W, int per_v, int per_s) {
    const int blocks = 256 * W, iters = 400;
    unsigned long long *o; float *s;
    (void)hipMalloc(&o, blocks * 4 * 8); (void)hipMalloc(&s, blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<P>, dim3(blocks), dim3(256), 0, 0, o, s, 1.5f, 0.25f, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<P>, dim3(blocks), dim3(256), 0, 0, o, s, 1.5f, 0.25f, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), o, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += (double)v; m /= h.size();
    const double reps = 64.0 * iters;
    printf("%-34s W=%d: %7.2f ns  %7.1f ticks per repetition per wave (%d VALU + %d SALU); per SIMD: %6.1f repetitions/us = %6.1f VALU/us\n", name, W,
           ms * 1e6 / reps, m / reps, per_v, per_s, W * reps / (ms * 1e3), W * reps * per_v / (ms * 1e3));
    (void)hipFree(o); (void)hipFree(s);
}

int main() {
    for (int W : {1, 2, 4}) {
        run<0>("dependent v_fma chain", W, 1, 0);
        run<1>("2 independent v_fma", W, 2, 0);
        run<2>("4 independent v_fma", W, 4, 0);
        run<6>("dependent v_pk_fma_f32 chain", W, 1, 0);
        run<5>("dependent v_fma + s_add", W, 1, 1);
        run<3>("sweep row as compiled", W, 5, 3);
        run<4>("sweep row without cmp/branch", W, 5, 1);
    }
    return 0;
}
