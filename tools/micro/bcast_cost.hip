// Microbenchmark (diagnostic, not part of the product): cost and semantics of the lane-broadcast forms a solver sweep can use when
// one wave holds the rows of several envs (16 or 32 lanes per env).  hipcc --offload-arch=gfx950 -O3 -o bcast_cost bcast_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define REP256(x) REP4(REP64(x))

template <int P>
__global__ __launch_bounds__(64, 1) void k(unsigned long long *out, float *sink, float a, float b) {
    float c = a + threadIdx.x, lam = b, lo = -1.f, hi = 1.f, C = 0.001f * a, t = 0.f, u = 0.f, t2 = 0.f;
    unsigned long long t0, t1;
    const unsigned long long mask = (1ull << 3) | (1ull << 35);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    // 16 lanes per env: med3 -> sub -> fmac with a DPP row broadcast of lane 3 of each 16-lane row (3 dependent VALU + lam select)
    if (P == 0) { REP256(asm volatile("v_med3_f32 %2, %0, %4, %5\n\tv_sub_f32 %3, %2, %1\n\tv_cndmask_b32_e64 %1, %1, %2, %7\n\ts_nop 0\n\t"
                                     "v_fmac_f32_dpp %0, %3, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                                     : "+v"(c), "+v"(lam), "+v"(t), "+v"(u) : "v"(lo), "v"(hi), "v"(C), "s"(mask));) }
    // 32 lanes per env: two DPP row broadcasts + v_permlane16_swap spread lane 3 of each 32-lane half over its half
    if (P == 1) { REP256(asm volatile("v_med3_f32 %2, %0, %4, %5\n\tv_sub_f32 %3, %2, %1\n\tv_cndmask_b32_e64 %1, %1, %2, %7\n\ts_nop 0\n\t"
                                     "v_mov_b32_dpp %8, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %9, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                                     "s_nop 1\n\tv_permlane16_swap_b32 %8, %9\n\tv_fmac_f32 %0, %8, %6"
                                     : "+v"(c), "+v"(lam), "+v"(t), "+v"(u) : "v"(lo), "v"(hi), "v"(C), "s"(mask), "v"(t2), "v"(u));) }
    // 32 lanes per env: two v_readlane + the apply under EXEC halves
    if (P == 2) { REP256(asm volatile("v_med3_f32 %2, %0, %4, %5\n\tv_sub_f32 %3, %2, %1\n\tv_readlane_b32 s20, %3, 3\n\tv_readlane_b32 s21, %3, 35\n\t"
                                     "v_cndmask_b32_e64 %1, %1, %2, %7\n\ts_mov_b64 exec, 0xffffffff\n\tv_fmac_f32 %0, s20, %6\n\ts_not_b64 exec, exec\n\t"
                                     "v_fmac_f32 %0, s21, %6\n\ts_mov_b64 exec, -1"
                                     : "+v"(c), "+v"(lam), "+v"(t), "+v"(u) : "v"(lo), "v"(hi), "v"(C), "s"(mask) : "s20", "s21");) }
    // today's row: one env per wave (reference point, tools/micro/issue_cost.hip P == 3)
    if (P == 3) { REP256(asm volatile("v_med3_f32 %2, %0, %4, %5\n\tv_sub_f32 %3, %2, %1\n\tv_readlane_b32 s20, %3, 3\n\t"
                                     "v_cndmask_b32_e64 %1, %1, %2, %7\n\ts_nop 1\n\tv_fmac_f32 %0, s20, %6"
                                     : "+v"(c), "+v"(lam), "+v"(t), "+v"(u) : "v"(lo), "v"(hi), "v"(C), "s"(mask) : "s20");) }
    // 32 lanes per env through the LDS crossbar: ds_bpermute of lane (lane & 32) | 3
    if (P == 4) { int addr = ((threadIdx.x & 32) | 3) << 2;
                  REP256(asm volatile("v_med3_f32 %2, %0, %4, %5\n\tv_sub_f32 %3, %2, %1\n\tv_cndmask_b32_e64 %1, %1, %2, %7\n\t"
                                     "ds_bpermute_b32 %8, %9, %3\n\ts_waitcnt lgkmcnt(0)\n\tv_fmac_f32 %0, %8, %6"
                                     : "+v"(c), "+v"(lam), "+v"(t), "+v"(u) : "v"(lo), "v"(hi), "v"(C), "s"(mask), "v"(t2), "v"(addr));) }
    // the slot of round 3's first attempt: scalar lane counter and shifted mask (two SALU per row), no branch
    if (P == 5) { asm volatile("s_mov_b32 s23, 2\n\ts_mov_b64 s[24:25], 8" ::: "s23", "s24", "s25");
                  REP256(asm volatile("v_med3_f32 %2, %0, %4, %5\n\tv_sub_f32 %3, %2, %1\n\ts_add_u32 s23, s23, 0\n\tv_readlane_b32 s22, %3, s23\n\t"
                                     "v_cndmask_b32_e64 %1, %1, %2, s[24:25]\n\ts_lshl_b64 s[24:25], s[24:25], 0\n\tv_fmac_f32 %0, s22, %6"
                                     : "+v"(c), "+v"(lam), "+v"(t), "+v"(u) : "v"(lo), "v"(hi), "v"(C) : "s22", "s23", "s24", "s25", "scc");) }
    // compile-time lane: owner select by v_cmp on the lane id (VCC), no scalar instruction, no branch
    if (P == 6) { int lid = threadIdx.x;
                  REP256(asm volatile("v_med3_f32 %2, %0, %4, %5\n\tv_sub_f32 %3, %2, %1\n\tv_readlane_b32 s22, %3, 3\n\tv_cmp_eq_u32_e32 vcc, 3, %7\n\t"
                                     "v_cndmask_b32_e32 %1, %1, %2, vcc\n\tv_fmac_f32 %0, s22, %6"
                                     : "+v"(c), "+v"(lam), "+v"(t), "+v"(u) : "v"(lo), "v"(hi), "v"(C), "v"(lid) : "s22", "vcc");) }
    // round 2's row as compiled: count test and not-taken branch, mask in an SGPR pair
    if (P == 7) { int sc = 5;
                  REP256(asm volatile("v_med3_f32 %2, %0, %4, %5\n\tv_sub_f32 %3, %2, %1\n\ts_cmp_lt_u32 %8, 3\n\tv_readlane_b32 s22, %3, 3\n\t"
                                     "v_cndmask_b32_e64 %1, %1, %2, %7\n\ts_nop 1\n\tv_fmac_f32 %0, s22, %6\n\ts_cbranch_scc1 1f\n1:"
                                     : "+v"(c), "+v"(lam), "+v"(t), "+v"(u) : "v"(lo), "v"(hi), "v"(C), "s"(mask), "s"(sc) : "s22", "scc");) }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = c + lam + t + u + t2;
}

// semantics: what each lane ends up with
__global__ void sem(float *o) {
    float v = (float)threadIdx.x, a, b;
    asm volatile("v_mov_b32_dpp %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "=&v"(a), "=&v"(b) : "v"(v));
    o[threadIdx.x] = a; o[64 + threadIdx.x] = b;
    float w = 1.f, acc = 100.f;
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(w));
    o[128 + threadIdx.x] = acc;
}

template <int P> void run(const char *name, int blocks, int per) {
    unsigned long long *o; float *s;
    hipMalloc(&o, blocks * 8); hipMalloc(&s, blocks * 64 * 4);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<P>, dim3(blocks), dim3(64), 0, 0, o, s, 1.5f, 0.25f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), o, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += (double)v; m /= blocks;
    printf("%-66s blocks %5d: %7.1f ticks per row slot (%d instructions)\n", name, blocks, m / 256.0, per);
    hipFree(o); hipFree(s);
}

int main() {
    float *o; hipMalloc(&o, 192 * 4);
    hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, o);
    std::vector<float> h(192);
    hipMemcpy(h.data(), o, 192 * 4, hipMemcpyDeviceToHost);
    for (int r = 0; r < 3; ++r) { printf(r == 0 ? "newbcast:3 x2 + permlane16_swap, vdst : " : r == 1 ? "                                 src  : " : "100 + fmac_dpp row_newbcast:5 of lane id: ");
        for (int i = 0; i < 64; i += 4) printf("%g ", h[64 * r + i]); printf("\n"); }
    for (int blocks : {256, 1024, 4096}) {
        run<7>("1 env/wave, round 2's row: + s_cmp, s_nop 1, not-taken s_cbranch", blocks, 8);
        run<5>("1 env/wave, scalar lane counter + shifted mask (2 SALU), no branch", blocks, 7);
        run<6>("1 env/wave, compile-time lane, v_cmp + v_cndmask vcc, no SALU, no branch", blocks, 6);
        run<3>("1 env/wave: med3 sub readlane cndmask nop fmac", blocks, 6);
        run<0>("4 envs/wave (16 lanes): med3 sub cndmask fmac_dpp newbcast", blocks, 4);
        run<1>("2 envs/wave (32 lanes): med3 sub cndmask 2 x mov_dpp permlane16_swap fmac", blocks, 8);
        run<2>("2 envs/wave (32 lanes): med3 sub 2 x readlane cndmask, fmac under exec halves", blocks, 10);
        run<4>("2 envs/wave (32 lanes): med3 sub cndmask ds_bpermute wait fmac", blocks, 6);
    }
    return 0;
}
