// Microbenchmark (diagnostic, not part of the product): cycles per instruction of ONE wave on its SIMD for the instruction
// patterns of the solver sweep.  hipcc --offload-arch=gfx950 -O3 -o issue_cost issue_cost.hip && ./issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define REP256(x) REP4(REP64(x))

template <int P>
__global__ __launch_bounds__(64, 1) void k(unsigned long long *out, float *sink, float a, float b) {
    float c = a + threadIdx.x, lam = b, lo = -1.f, hi = 1.f, C = 0.001f * a, d0 = a, d1 = b, d2 = a * b, d3 = a - b, t = 0.f, u = 0.f;
    unsigned long long t0, t1;
    const unsigned long long mask = 1ull << 3;
    int sc = 5;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (P == 0) { REP256(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c) : "v"(C), "v"(lam));) }            // dependent fma chain
    if (P == 1) { REP64(asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(C), "v"(lam));) }            // 4 independent chains
    if (P == 2) { REP256(asm volatile("v_med3_f32 %2, %0, %4, %5\n\tv_sub_f32 %3, %2, %1\n\ts_cmp_lt_u32 %8, 3\n\tv_readlane_b32 s20, %3, 3\n\t"
                                     "v_cndmask_b32_e64 %1, %1, %2, %7\n\ts_nop 1\n\tv_fmac_f32 %0, s20, %6\n\ts_cbranch_scc1 1f\n1:"
                                     : "+v"(c), "+v"(lam), "+v"(t), "+v"(u) : "v"(lo), "v"(hi), "v"(C), "s"(mask), "s"(sc) : "s20", "scc");) }  // the sweep row
    if (P == 3) { REP256(asm volatile("v_med3_f32 %2, %0, %4, %5\n\tv_sub_f32 %3, %2, %1\n\tv_readlane_b32 s20, %3, 3\n\t"
                                     "v_cndmask_b32_e64 %1, %1, %2, %7\n\ts_nop 1\n\tv_fmac_f32 %0, s20, %6"
                                     : "+v"(c), "+v"(lam), "+v"(t), "+v"(u) : "v"(lo), "v"(hi), "v"(C), "s"(mask) : "s20");) }  // row without compare/branch
    if (P == 4) { REP256(asm volatile("v_med3_f32 %2, %0, %4, %5\n\tv_readlane_b32 s20, %2, 3\n\ts_nop 3\n\tv_fmac_f32 %0, s20, %6"
                                     : "+v"(c), "+v"(lam), "+v"(t), "+v"(u) : "v"(lo), "v"(hi), "v"(C), "s"(mask) : "s20");) }  // med3 -> readlane -> fmac only
    if (P == 5) { REP256(asm volatile("v_readlane_b32 s20, %0, 3\n\ts_nop 3\n\tv_fmac_f32 %0, s20, %1" : "+v"(c) : "v"(C) : "s20");) }  // readlane -> fmac chain
    if (P == 6) { REP256(asm volatile("s_nop 0");) }
    if (P == 7) { REP256(asm volatile("s_cmp_lt_u32 %0, 3\n\ts_cbranch_scc1 1f\n1:" ::"s"(sc) : "scc");) }        // not-taken branch pairs
    if (P == 8) { REP256(asm volatile("v_med3_f32 %2, %0, %4, %5\n\tv_sub_f32 %3, %2, %1\n\tv_fmac_f32 %0, %3, %6"
                                     : "+v"(c), "+v"(lam), "+v"(t), "+v"(u) : "v"(lo), "v"(hi), "v"(C));) }      // 3 dependent VALU, no cross-lane
    if (P == 9) { REP256(asm volatile("v_mov_b32_dpp %3, %2 row_bcast:15 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32 %0, %3, %6"
                                     : "+v"(c), "+v"(lam), "+v"(t), "+v"(u) : "v"(lo), "v"(hi), "v"(C));) }      // dpp move + fmac
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = c + lam + d0 + d1 + d2 + d3 + t + u;
}

template <int P> void run(const char *name, int blocks, int per) {
    unsigned long long *o; float *s;
    hipMalloc(&o, blocks * 8); hipMalloc(&s, blocks * 64 * 4);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<P>, dim3(blocks), dim3(64), 0, 0, o, s, 1.5f, 0.25f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), o, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += (double)v; m /= blocks;
    printf("%-44s blocks %5d: %7.1f ticks per repetition (%d instructions each -> %.1f per instruction)\n", name, blocks, m / 256.0, per, m / 256.0 / per);
    hipFree(o); hipFree(s);
}

int main() {
    for (int blocks : {256, 1024, 4096}) { // 1, 4 and 16 single-wave workgroups per CU = 1 wave per SIMD ... 4 per SIMD
        run<0>("dependent v_fma chain", blocks, 1);
        run<1>("4 independent v_fma chains (per 4)", blocks, 4);
        run<8>("med3 -> sub -> fmac (dependent, VALU only)", blocks, 3);
        run<5>("readlane -> s_nop 3 -> fmac", blocks, 3);
        run<4>("med3 -> readlane -> s_nop 3 -> fmac", blocks, 4);
        run<3>("sweep row without compare/branch", blocks, 6);
        run<2>("sweep row as compiled (8 instructions)", blocks, 8);
        run<6>("s_nop 0", blocks, 1);
        run<7>("s_cmp + not-taken s_cbranch", blocks, 2);
        run<9>("v_mov_dpp row_bcast -> fmac", blocks, 2);
    }
    return 0;
}
