// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the ACCESS PATTERN of the step kernels (diagnostic, not part of the product).
// MI355X_MICROARCH.md (HBM): FETCH_SIZE is calibrated (x 2) only for 16-B-per-lane streams; "other access widths are uncalibrated: calibrate on a
// known byte count in your own access pattern before trusting an absolute".  The step kernels move per-env records with 4-B-per-lane accesses
// (lane i <-> float i of a 128-byte record) and small per-env rows (reward 4 B, done 1 B, info 16 B, observation 184 B); these kernels do exactly
// that and nothing else, with the same launch shape (256 threads = four env-waves, the XCD-aware group index), so the bytes are known:
//   k_read   per env: state 128 + items 128 + aux 16 + actions 32                    = 304 B read,  0 written
//   k_write  per env: state 128 + aux 16 + obs 184 + reward 4 + done 1 + info 16     = 349 B written, 0 read
//   k_both   both, the stores behind a dependent chain of ~40 us like the real kernel (lines shared by neighbouring envs arrive far apart in time)
//   hipcc --offload-arch=gfx950 -O2 -o traffic_calib traffic_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/f -- ./traffic_calib [envs]     (and the same with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ int xcd_group() {
    const int nb = (int)gridDim.x, b = (int)blockIdx.x, x = b & 7, per = nb >> 3, rem = nb & 7;
    return x * per + (x < rem ? x : rem) + (b >> 3);
}

struct Bufs { float *state, *items; int *aux; float *act, *obs, *rew; unsigned char *done; float *info; };

__device__ __forceinline__ float load_records(const Bufs &b, int e, int lane) {
    float v = lane < 32 ? b.state[(size_t)e * 32 + lane] : b.items[(size_t)e * 32 + (lane - 32)];
    if (lane < 4) v += (float)b.aux[(size_t)e * 4 + lane];
    if (lane < 8) v += b.act[(size_t)e * 8 + lane];
    return v;
}
__device__ __forceinline__ void store_records(const Bufs &b, int e, int lane, float v) {
    if (lane < 32) b.state[(size_t)e * 32 + lane] = v;
    if (lane < 4) b.aux[(size_t)e * 4 + lane] = (int)v;
    if (lane < 46) b.obs[(size_t)e * 46 + lane] = v;
    if (lane == 0) b.rew[e] = v;
    if (lane == 1) b.done[e] = (unsigned char)(v > 0.f);
    if (lane >= 4 && lane < 8) b.info[(size_t)e * 4 + (lane - 4)] = v;
}

__global__ __launch_bounds__(256) void k_read(Bufs b, float *sink, int n) {
    const int e = xcd_group() * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= n) return;
    const float v = load_records(b, e, lane);
    if (v == 1.2345e38f) sink[0] = v; /* never true: keeps the loads */
}
__global__ __launch_bounds__(256) void k_write(Bufs b, float seed, int n) {
    const int e = xcd_group() * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= n) return;
    store_records(b, e, lane, seed + (float)lane);
}
__global__ __launch_bounds__(256) void k_both(Bufs b, int chain, int n) {
    const int e = xcd_group() * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= n) return;
    float v = load_records(b, e, lane);
    for (int i = 0; i < chain + (e & 3) * (chain >> 3); ++i) v = __builtin_fmaf(v, 0.999f, 1e-3f); /* a dependent chain, a little different per wave of the group */
    store_records(b, e, lane, v);
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4096, launches = 50;
    Bufs b;
    float *sink;
    CK(hipMalloc(&b.state, (size_t)n * 128)); CK(hipMalloc(&b.items, (size_t)n * 128)); CK(hipMalloc(&b.aux, (size_t)n * 16));
    CK(hipMalloc(&b.act, (size_t)n * 32)); CK(hipMalloc(&b.obs, (size_t)n * 184)); CK(hipMalloc(&b.rew, (size_t)n * 4));
    CK(hipMalloc(&b.done, (size_t)n)); CK(hipMalloc(&b.info, (size_t)n * 16)); CK(hipMalloc(&sink, 256));
    CK(hipMemset(b.state, 0, (size_t)n * 128)); CK(hipMemset(b.items, 0, (size_t)n * 128)); CK(hipMemset(b.aux, 0, (size_t)n * 16)); CK(hipMemset(b.act, 0, (size_t)n * 32));
    const int groups = (n + 3) / 4;
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(k_read, dim3(groups), dim3(256), 0, 0, b, sink, n);
    CK(hipDeviceSynchronize());
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(k_write, dim3(groups), dim3(256), 0, 0, b, (float)i, n);
    CK(hipDeviceSynchronize());
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(k_both, dim3(groups), dim3(256), 0, 0, b, 20000, n);
    CK(hipDeviceSynchronize());
    printf("traffic_calib: %d envs, %d launches of each kernel; known bytes per env: k_read 304 read / 0 written, k_write 0 / 349, k_both 304 / 349\n", n, launches);
    return 0;
}
