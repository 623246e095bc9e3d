/*
 * hrl_hip.hip -- gfx950 (MI355X) implementation of include/hrl_envs.h.
 *
 * One wavefront = one environment; four env-waves form a 256-thread workgroup (one for the PointBot).  A wave runs the phases of
 * step_core.h with
 *   - per-wave state staged in LDS (WaveLds 7.5 KB, <= 128 VGPRs -> 16 waves per CU, 4 per SIMD),
 *   - the lane-sparse articulated-body phases of the four envs of a group executed once, 16 lanes per env, by the group's
 *     leader wave (two s_barrier per substep); everything else by the env's own wave behind wave-level LDS fences,
 *   - coalesced 128-byte loads/stores of the packed state / item records (lane i <-> float i of the record),
 *   - the contact/limit solver entirely in lane registers: row r lives in lane r, its impulse change is broadcast with
 *     one v_readlane_b32 (no LDS, no barrier, no reduction on the dependent chain),
 *   - ballot + popcount for compacting active contacts / limits into solver rows,
 *   - the wave's issue priority rotated with the substep (GpuExec::priority): the four workgroups of a CU take turns.
 * There is no cross-workgroup communication, but neighbouring envs SHARE CACHE LINES of the output arrays (reward 4 B, done 1 B, info 16 B per
 * env, observation rows that are no multiple of a line): the dispatcher deals workgroups to the 8 XCDs round-robin by blockIdx, each XCD has its
 * own L2, so with blockIdx = group index eight L2s each held a piece of every such line and wrote it back separately.  xcd_group() gives
 * every XCD one contiguous eighth of the groups instead.
 */
#include <hip/hip_runtime.h>

#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "host_cfg.h"

#ifndef HRL_PRIO_MODE
#define HRL_PRIO_MODE 1 /* 0 = equal wave priorities (A/B builds of tools/variants.py) */
#endif

using namespace hrl;

namespace {

thread_local std::string g_err;
unsigned long long *g_stamps = nullptr; /* set only by the diagnostic entry point hrl_debug_set_stamps */

/* G = envs (= waves) per workgroup.  G = 1: one 64-thread workgroup per env, every phase on the env's own wave.  G = 4: four
 * env-waves share a 256-thread workgroup and the lane-sparse phases of all four run once on wave 0 (step_core.h,
 * ant_group_block); phases of one wave are ordered by a wave-level LDS fence, the two blocks of a substep by s_barrier. */
template <int G>
struct GpuExec {
    WaveLds *Ls; /* the group's records, [G] */
    LaneRegs r;
    int lane, wave;
    __device__ __forceinline__ WaveLds &lds() { return Ls[G == 1 ? 0 : wave]; }
    __device__ __forceinline__ WaveLds &lds(int k) { return Ls[G == 1 ? 0 : k]; }
    __device__ __forceinline__ LaneRegs &reg(int) { return r; }
    /* orders this wave's LDS writes before its later LDS reads (the lanes of a wave exchange data through LDS) */
    __device__ __forceinline__ void wave_sync() {
        if (G == 1) __syncthreads(); /* a 64-thread workgroup: compiles to the wait, no s_barrier */
        else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
    }
    __device__ __forceinline__ void group_sync() { if (G > 1) __syncthreads(); }
    /* a phase of the group block: executed by the leader wave, lane >> 4 = env of the group */
    template <class F>
    __device__ __forceinline__ void leader(F f) {
        if (G == 1 || wave == 0) { f(lane); wave_sync(); }
    }
    __device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
    __device__ __forceinline__ int wave_index() const { return __builtin_amdgcn_readfirstlane(wave); }
    static constexpr __device__ __forceinline__ int group_size() { return G; }
    /* Makes the lane id (and the friction links) opaque to the optimizer at this point.  Without it every
     * lane-derived value of the unrolled solver (44 `lane == r` masks, LDS addresses, ...) is loop-invariant, gets
     * hoisted to the top of the kernel and is spilled to scratch for the whole substep loop. */
    __device__ __forceinline__ void refresh() { asm volatile("" : "+v"(lane), "+v"(r.fn), "+v"(wave)); }
    /* 1.0 on lane r (wave-uniform), 0.0 elsewhere: one v_cndmask on a scalar lane mask */
    __device__ __forceinline__ float lane_one(int, int r) {
        float d;
        const unsigned long long m = 1ull << r;
        asm("v_cndmask_b32_e64 %0, 0, 1.0, %1" : "=v"(d) : "s"(m));
        return d;
    }
    /* Issue arbitration between the waves of a SIMD goes by priority, then age: with four workgroups per CU at equal priority the
     * first-dispatched one runs at its solo speed and the last one gets the leftover slots, so the launch lasts as long as its
     * slowest workgroup while the mean one is done 13 % earlier (tools/wg_times.py).  slot() = the wave's slot on its SIMD
     * (HW_ID.WAVE_ID; with four workgroups of four waves per CU it equals the workgroup's slot on the CU); priority(p) sets the
     * wave's priority to p mod 4: rotating it with the substep hands every wave of a SIMD every rank once per step. */
    __device__ __forceinline__ int slot() const {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(hw));
        return (int)hw;
    }
    __device__ __forceinline__ void priority(int p) const {
#if HRL_PRIO_MODE
        switch (p & 3) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
#endif
        (void)p;
    }
    /* same for a wave-uniform value: comparisons against it are redone (one s_cmp each) after this point */
    __device__ __forceinline__ void refresh_uniform(int &v) { asm volatile("" : "+s"(v)); }
#ifdef HRL_STAMPS
    /* diagnostic build (never shipped): cycles between phase boundaries, summed per phase id; stamp id = the phase that just
     * ENDED.  s_memtime + lgkmcnt(0) as one statement (cdna_hip_programming.md, In-kernel stamps).  The sums live in LDS
     * (one lane adds), not in registers: sixteen 64-bit accumulators per lane would push the kernel into scratch. */
    unsigned long long t_last = 0;
    unsigned long long *acc = nullptr; /* [24] in LDS, this wave's; null in the kernels that do not collect (k_reset, k_set_goals) */
    __device__ __forceinline__ void stamp(int id) {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        if (acc && t_last && lane == 0) acc[id] += t - t_last;
        t_last = t;
    }
    __device__ __forceinline__ void flush_stamps(const DevBufs &b) {
        if (acc && b.stamps && lane == 0)
            for (int i = 0; i < 24; ++i) atomicAdd(&b.stamps[(G == 1 ? 0 : 24 * wave) + i], acc[i]); /* one table per wave of the group */
    }
#else
    __device__ __forceinline__ void stamp(int) {}
    __device__ __forceinline__ void flush_stamps(const DevBufs &) {}
#endif
    template <class F>
    __device__ __forceinline__ void each(F f) {
        f(lane);
        wave_sync();
    }
    template <class P, class W, class Q>
    __device__ __forceinline__ int each_compact(P pred, W write, Q post) {
        auto h = pred(lane);
        const unsigned long long m = __ballot(h.ok);
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        if (h.ok) write(lane, rank, h);
        post(lane, h);
        wave_sync();
        return __popcll(m);
    }
    /* the same per 16-lane slice (four envs per wave): rank and count within the lane's slice; post also gets the slice's count */
    template <class P, class W, class Q>
    __device__ __forceinline__ void each_compact16(P pred, W write, Q post) {
        auto h = pred(lane);
        const unsigned long long m = __ballot(h.ok);
        const unsigned sub = (unsigned)(m >> (lane & 48)) & 0xffffu;
        const int rank = __popc(sub & ((1u << (lane & 15)) - 1u));
        if (h.ok) write(lane, rank, h);
        post(lane, h, __popc(sub));
        wave_sync();
    }
    /* lane mask of a predicate (v_cmp into an SGPR pair) */
    template <class P>
    __device__ __forceinline__ unsigned long long each_ballot(P pred) { return __ballot(pred(lane)); }
    /* solver row `src` (wave-uniform): every lane produces the candidate {impulse, change} of its own row; lane `src`
     * keeps its candidate impulse (v_cndmask on a scalar lane mask) and its change is broadcast to the wave
     * (v_readlane_b32) for every lane to apply */
    template <class P, class C>
    __device__ __forceinline__ void each_row(int src, P produce, C apply) {
        const F2b v = produce(lane);
        const unsigned long long owner = 1ull << src;
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r.lam) : "v"(r.lam), "v"(v.ln), "s"(owner));
        apply(lane, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.dl), src)));
    }
    /* every lane fetches the value of the lane it names (ds_bpermute_b32: LDS crossbar, no memory) */
    template <class V, class I, class C>
    __device__ __forceinline__ void each_shuffle(V value, I index, C consume) {
        const float v = value(lane);
        consume(lane, __int_as_float(__builtin_amdgcn_ds_bpermute(index(lane) << 2, __float_as_int(v))));
    }
};

/* blockIdx -> index of the env group this workgroup steps: XCD (blockIdx & 7) takes the groups [x * per + min(x, rem), ...) in order, so that
 * the envs whose output rows share cache lines are written through ONE L2 (a bijection for every grid size) */
__device__ __forceinline__ int xcd_group() {
    const int nb = (int)gridDim.x, b = (int)blockIdx.x, x = b & 7, per = nb >> 3, rem = nb & 7;
    return x * per + (x < rem ? x : rem) + (b >> 3);
}

/* The ~0.5 KB of constants are read through a pointer (scalar loads on demand, scalar-cache resident) rather than
 * passed by value: by-value kernel arguments were all preloaded into SGPRs and spilled.
 * One kernel per env kind: each contains only its own env's code, which keeps the instruction footprint small
 * (all waves of a CU share one instruction cache). */
template <int KIND, int G>
__global__ __launch_bounds__(64 * G, 4) void k_step(DevBufs b, const DevCfg *__restrict__ cp) {
    __shared__ WaveLds L[G];
    LaneRegs regs; /* deliberately uninitialised: every field is defined by the phase that produces it */
    GpuExec<G> x{L, regs, (int)threadIdx.x & 63, (int)threadIdx.x >> 6};
#ifdef HRL_STAMPS
    __shared__ unsigned long long stamp_acc[G][24];
    if ((threadIdx.x & 63) < 24) stamp_acc[threadIdx.x >> 6][threadIdx.x & 63] = 0;
    x.acc = stamp_acc[threadIdx.x >> 6];
#endif
#ifdef HRL_WGTIME
    /* diagnostic build (never shipped, tools/wg_times.py): start / end clock and hardware id of every env-wave */
    unsigned long long wg_t0, wg_t1;
    if ((threadIdx.x & 63) == 0) L[threadIdx.x >> 6].dbg_rows = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wg_t0)::"memory");
#endif
    const int env = xcd_group() * G + ((int)threadIdx.x >> 6);
    step_entry<KIND>(x, b, *cp, env);
#ifdef HRL_WGTIME
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wg_t1)::"memory");
    if (b.stamps && (threadIdx.x & 63) == 0 && env < cp->n_envs) { /* a wave of a ragged last group has no row in the buffer */
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long *o = b.stamps + 4 * env;
        o[0] = wg_t0; o[1] = wg_t1; o[2] = hw; o[3] = (xcc & 15) | ((unsigned long long)(unsigned)L[threadIdx.x >> 6].dbg_rows << 8);
    }
#endif
}
template <int KIND>
__global__ __launch_bounds__(64, 4) void k_reset(DevBufs b, const DevCfg *__restrict__ cp) {
    __shared__ WaveLds L[1];
    LaneRegs regs;
    GpuExec<1> x{L, regs, (int)threadIdx.x, 0};
    reset_entry<KIND>(x, b, *cp, xcd_group());
}
template <int KIND>
__global__ __launch_bounds__(64, 4) void k_observe(DevBufs b, const DevCfg *__restrict__ cp) {
    __shared__ WaveLds L[1];
    LaneRegs regs;
    GpuExec<1> x{L, regs, (int)threadIdx.x, 0};
    observe_entry<KIND>(x, b, *cp, xcd_group());
}
__global__ __launch_bounds__(64, 4) void k_set_goals(DevBufs b, const DevCfg *__restrict__ cp, const float *goals_xy, int n_goals, uint8_t *ok) {
    __shared__ WaveLds L[1];
    LaneRegs regs;
    GpuExec<1> x{L, regs, (int)threadIdx.x, 0};
    set_goals_entry(x, b, *cp, xcd_group(), goals_xy, n_goals, ok);
}
using kernel_fn = void (*)(DevBufs, const DevCfg *);
/* group = envs per workgroup of the step kernel: 4 for the ant kinds (1 selectable for A/B measurements), 1 for the point bot */
kernel_fn step_kernel(int kind, int group) {
    if (group == 4) switch (kind) {
        case HRL_ANT_FLAT: return k_step<HRL_ANT_FLAT, 4>;
        case HRL_ANT_GATHER: return k_step<HRL_ANT_GATHER, 4>;
        case HRL_ANT_MAZE: return k_step<HRL_ANT_MAZE, 4>;
        case HRL_ANT_MAZE_MJ: return k_step<HRL_ANT_MAZE_MJ, 4>;
        case HRL_ANT_FLAGRUN: return k_step<HRL_ANT_FLAGRUN, 4>;
    }
    switch (kind) {
        case HRL_ANT_FLAT: return k_step<HRL_ANT_FLAT, 1>;
        case HRL_ANT_GATHER: return k_step<HRL_ANT_GATHER, 1>;
        case HRL_ANT_MAZE: return k_step<HRL_ANT_MAZE, 1>;
        case HRL_ANT_MAZE_MJ: return k_step<HRL_ANT_MAZE_MJ, 1>;
        case HRL_ANT_FLAGRUN: return k_step<HRL_ANT_FLAGRUN, 1>;
        default: return k_step<HRL_POINT_GATHER, 1>;
    }
}
kernel_fn reset_kernel(int kind) {
    switch (kind) {
        case HRL_ANT_FLAT: return k_reset<HRL_ANT_FLAT>;
        case HRL_ANT_GATHER: return k_reset<HRL_ANT_GATHER>;
        case HRL_ANT_MAZE: return k_reset<HRL_ANT_MAZE>;
        case HRL_ANT_MAZE_MJ: return k_reset<HRL_ANT_MAZE_MJ>;
        case HRL_ANT_FLAGRUN: return k_reset<HRL_ANT_FLAGRUN>;
        default: return k_reset<HRL_POINT_GATHER>;
    }
}
kernel_fn observe_kernel(int kind) {
    switch (kind) {
        case HRL_ANT_FLAT: return k_observe<HRL_ANT_FLAT>;
        case HRL_ANT_GATHER: return k_observe<HRL_ANT_GATHER>;
        case HRL_ANT_MAZE: return k_observe<HRL_ANT_MAZE>;
        case HRL_ANT_MAZE_MJ: return k_observe<HRL_ANT_MAZE_MJ>;
        case HRL_ANT_FLAGRUN: return k_observe<HRL_ANT_FLAGRUN>;
        default: return k_observe<HRL_POINT_GATHER>;
    }
}
/* packed record <-> split qpos[N][15], qvel[N][14] */
__global__ void k_get_state(const float *state, float *qpos, float *qvel, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, e = i >> 5, k = i & 31;
    if (e >= n) return;
    const float v = state[i];
    if (k < 15) qpos[e * 15 + k] = v;
    else if (k < 29) qvel[e * 14 + (k - 15)] = v;
}
__global__ void k_set_state(float *state, const float *qpos, const float *qvel, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, e = i >> 5, k = i & 31;
    if (e >= n) return;
    if (k < 15) state[i] = qpos[e * 15 + k];
    else if (k < 29) state[i] = qvel[e * 14 + (k - 15)];
}

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
int hip_fail(hipError_t e, const char *what) { return fail(HRL_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }

/* An optional pointer of the record that lies within the caller's struct_size (hrl_buffers: a caller compiled against an older header of this ABI
 * version handed over a shorter record; what lies beyond it is not the caller's to set). */
#define HRL_OPT(b, field) ((b)->struct_size >= offsetof(hrl_buffers, field) + sizeof((b)->field) ? (b)->field : nullptr)
DevBufs to_dev(const hrl_buffers *b, const uint8_t *mask) {
    DevBufs d;
    d.state = b->state; d.items = b->items; d.aux = b->aux; d.actions = b->actions; d.obs = b->obs;
    d.reward = b->reward; d.done = b->done; d.info = b->info; d.mask = mask;
    d.final_obs = HRL_OPT(b, final_obs); d.truncated = HRL_OPT(b, truncated); d.goal = HRL_OPT(b, goal); d.rows = HRL_OPT(b, solver_rows);
    d.stamps = g_stamps;
    return d;
}

}  // namespace

struct hrl_handle {
    hrl_config cfg;
    DevCfg dc;
    DevCfg *d_dc; /* device copy of the constants (the only device memory the library owns) */
    int device;   /* the device that was current at hrl_create(): d_dc lives there, and so must the caller's buffers */
    int group;    /* envs per workgroup of the step kernel */
};

namespace {
/* What every entry point that launches checks first: a handle, a buffer record that was initialised (hrl_buffers_init) and the device.
 * A launch goes to the CURRENT device with the handle's constants pointer: from another device that is a fault in the kernel, not an error
 * code -- so a host that drives several GPUs from one process gets HRL_ERR_BAD_ARG here and is told to hipSetDevice() first (hipGetDevice
 * is a thread-local read: ~20 ns, tools/host_overhead.py). */
int check_call(const hrl_handle *h, const hrl_buffers *b, const char *who) {
    if (!h || !b) return fail(HRL_ERR_BAD_ARG, std::string(who) + ": null handle or buffer record");
    if (b->struct_size < HRL_BUFFERS_SIZE_V7_BASE || b->struct_size > sizeof(hrl_buffers) || b->struct_size % sizeof(void *) != 0)
        return fail(HRL_ERR_BAD_ARG, std::string(who) + ": hrl_buffers.struct_size = " + std::to_string((unsigned long long)b->struct_size) +
                                         " is not the size of a known layout: initialise the record with hrl_buffers_init() (include/hrl_envs.h)");
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != h->device)
        return fail(HRL_ERR_BAD_ARG, std::string(who) + ": the handle was created on HIP device " + std::to_string(h->device) + ", the current device is " +
                                         std::to_string(cur) + ": hipSetDevice(" + std::to_string(h->device) + ") before calling (one handle per device)");
    return HRL_OK;
}
}  // namespace

extern "C" {

int hrl_default_config(int32_t env_kind, hrl_config *cfg) {
    const int rc = default_config(env_kind, cfg);
    return rc == HRL_OK ? rc : fail(rc, "hrl_default_config: bad env_kind or null cfg");
}
int hrl_obs_dim(const hrl_config *cfg) { return cfg ? obs_dim(cfg) : -1; }
int hrl_act_dim(const hrl_config *cfg) { return cfg ? act_dim(cfg) : -1; }
int hrl_items_stride(const hrl_config *cfg) { return cfg ? items_stride(cfg) : -1; }

int hrl_create(const hrl_config *cfg, hrl_handle **out) {
    if (!out) return fail(HRL_ERR_BAD_ARG, "hrl_create: null out");
    *out = nullptr;
    const std::string why = validate(cfg);
    if (!why.empty()) return fail(HRL_ERR_BAD_ARG, "hrl_create: " + why);
    int ndev = 0;
    const hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(HRL_ERR_NO_DEVICE, "hrl_create: no HIP device (this library has no CPU path)");
    hrl_handle *h = new hrl_handle;
    h->cfg = *cfg;
    build_devcfg(*cfg, h->dc);
    h->d_dc = nullptr;
    h->group = cfg->env_kind == HRL_POINT_GATHER ? 1 : 4;
    if (cfg->model.step_group == 1) h->group = 1; /* measurement reference: the one-wave-per-env launch (hrl_model.step_group) */
    hipError_t e2 = hipGetDevice(&h->device);
    if (e2 == hipSuccess) e2 = hipMalloc((void **)&h->d_dc, sizeof(DevCfg));
    if (e2 == hipSuccess) e2 = hipMemcpy(h->d_dc, &h->dc, sizeof(DevCfg), hipMemcpyHostToDevice);
    if (e2 != hipSuccess) {
        if (h->d_dc) (void)hipFree(h->d_dc);
        delete h;
        return hip_fail(e2, "hrl_create: device constants");
    }
    *out = h;
    return HRL_OK;
}
int hrl_destroy(hrl_handle *h) {
    if (h && h->d_dc) (void)hipFree(h->d_dc);
    delete h;
    return HRL_OK;
}

int hrl_update_config(hrl_handle *h, const hrl_config *cfg, void *stream) {
    if (!h) return fail(HRL_ERR_BAD_ARG, "hrl_update_config: null handle");
    const std::string why = validate(cfg);
    if (!why.empty()) return fail(HRL_ERR_BAD_ARG, "hrl_update_config: " + why);
    if (cfg->env_kind != h->cfg.env_kind || cfg->num_envs != h->cfg.num_envs || obs_dim(cfg) != obs_dim(&h->cfg) || act_dim(cfg) != act_dim(&h->cfg) ||
        items_stride(cfg) != items_stride(&h->cfg))
        return fail(HRL_ERR_BAD_ARG, "hrl_update_config: env_kind, num_envs, the observation / action width and the items stride belong to the buffers and cannot change on a live handle");
    /* what decides the MEANING of the records the caller holds: new item slots would be read uninitialised, another goal mode reads the flagrun record differently */
    if (cfg->n_food != h->cfg.n_food || cfg->n_poison != h->cfg.n_poison)
        return fail(HRL_ERR_BAD_ARG, "hrl_update_config: n_food / n_poison cannot change on a live handle (the items record holds exactly these items: make a new handle and reset)");
    if (cfg->flag_manual_goals != h->cfg.flag_manual_goals || (cfg->flag_max_target_dist > 0) != (h->cfg.flag_max_target_dist > 0) || cfg->flag_goal_capacity != h->cfg.flag_goal_capacity)
        return fail(HRL_ERR_BAD_ARG, "hrl_update_config: the goal mode of a flagrun env (manual goals, goals near the robot, the list capacity) cannot change on a live handle");
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != h->device)
        return fail(HRL_ERR_BAD_ARG, "hrl_update_config: the handle was created on HIP device " + std::to_string(h->device) + ", the current device is " + std::to_string(cur));
    DevCfg dc;
    build_devcfg(*cfg, dc);
    const hipError_t e = hipMemcpyAsync(h->d_dc, &dc, sizeof(DevCfg), hipMemcpyHostToDevice, (hipStream_t)stream); /* pageable source: staged before the call returns */
    if (e != hipSuccess) return hip_fail(e, "hrl_update_config: device constants");
    h->cfg = *cfg; h->dc = dc;
    h->group = cfg->env_kind == HRL_POINT_GATHER || cfg->model.step_group == 1 ? 1 : 4;
    return HRL_OK;
}

static bool needs_items(const DevCfg &dc) {
    return dc.kind == HRL_ANT_GATHER || dc.kind == HRL_POINT_GATHER || (dc.kind == HRL_ANT_FLAGRUN && dc.flag_path_on);
}
static const char *items_why = "this env keeps state in the items buffer (gather kinds: the item positions; flagrun: the goal bookkeeping of set_target())";

int hrl_reset(hrl_handle *h, const hrl_buffers *b, const uint8_t *mask, void *stream) {
    if (const int rc = check_call(h, b, "hrl_reset")) return rc;
    if (!b->state || !b->aux || !b->obs) return fail(HRL_ERR_BAD_ARG, "hrl_reset: null buffer");
    if (needs_items(h->dc) && !b->items) return fail(HRL_ERR_BAD_ARG, std::string("hrl_reset: ") + items_why);
    hipLaunchKernelGGL(reset_kernel(h->dc.kind), dim3(h->dc.n_envs), dim3(64), 0, (hipStream_t)stream, to_dev(b, mask), (const DevCfg *)h->d_dc);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? HRL_OK : hip_fail(e, "hrl_reset launch");
}

int hrl_step(hrl_handle *h, const hrl_buffers *b, void *stream) {
    if (const int rc = check_call(h, b, "hrl_step")) return rc;
    if (!b->state || !b->aux || !b->obs || !b->actions || !b->reward || !b->done || !b->info) return fail(HRL_ERR_BAD_ARG, "hrl_step: null buffer");
    if (needs_items(h->dc) && !b->items) return fail(HRL_ERR_BAD_ARG, std::string("hrl_step: ") + items_why);
    const int G = h->group;
    hipLaunchKernelGGL(step_kernel(h->dc.kind, G), dim3((h->dc.n_envs + G - 1) / G), dim3(64 * G), 0, (hipStream_t)stream, to_dev(b, nullptr), (const DevCfg *)h->d_dc);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? HRL_OK : hip_fail(e, "hrl_step launch");
}

int hrl_buffers_init(hrl_buffers *b) {
    if (!b) return fail(HRL_ERR_BAD_ARG, "hrl_buffers_init: null record");
    memset(b, 0, sizeof(*b));
    b->struct_size = sizeof(*b);
    return HRL_OK;
}

int hrl_observe(hrl_handle *h, const hrl_buffers *b, const uint8_t *mask, void *stream) {
    if (const int rc = check_call(h, b, "hrl_observe")) return rc;
    if (!b->state || !b->aux || !b->obs) return fail(HRL_ERR_BAD_ARG, "hrl_observe: null buffer");
    if (needs_items(h->dc) && !b->items) return fail(HRL_ERR_BAD_ARG, std::string("hrl_observe: ") + items_why);
    hipLaunchKernelGGL(observe_kernel(h->dc.kind), dim3(h->dc.n_envs), dim3(64), 0, (hipStream_t)stream, to_dev(b, mask), (const DevCfg *)h->d_dc);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? HRL_OK : hip_fail(e, "hrl_observe launch");
}

int hrl_get_state(hrl_handle *h, const hrl_buffers *b, float *qpos, float *qvel, void *stream) {
    if (const int rc = check_call(h, b, "hrl_get_state")) return rc;
    if (!b->state || !qpos || !qvel) return fail(HRL_ERR_BAD_ARG, "hrl_get_state: null argument");
    const int n = h->dc.n_envs, total = n * 32;
    hipLaunchKernelGGL(k_get_state, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, b->state, qpos, qvel, n);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? HRL_OK : hip_fail(e, "hrl_get_state launch");
}
int hrl_set_state(hrl_handle *h, const hrl_buffers *b, const float *qpos, const float *qvel, void *stream) {
    if (const int rc = check_call(h, b, "hrl_set_state")) return rc;
    if (!b->state || !qpos || !qvel) return fail(HRL_ERR_BAD_ARG, "hrl_set_state: null argument");
    const int n = h->dc.n_envs, total = n * 32;
    hipLaunchKernelGGL(k_set_state, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, b->state, qpos, qvel, n);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? HRL_OK : hip_fail(e, "hrl_set_state launch");
}

int hrl_set_goals(hrl_handle *h, const hrl_buffers *b, const float *goals_xy, int32_t n_goals, const uint8_t *mask, void *stream) {
    if (const int rc = check_call(h, b, "hrl_set_goals")) return rc;
    if (!b->state || !b->aux || !b->obs || !b->items || !goals_xy) return fail(HRL_ERR_BAD_ARG, "hrl_set_goals: null buffer");
    if (h->dc.kind != HRL_ANT_FLAGRUN || !h->dc.flag_manual) return fail(HRL_ERR_BAD_ARG, "hrl_set_goals: only for AntFlagrun with flag_manual_goals (manual_goal_creation, ant_flagrun_env.py:27)");
    if (h->dc.flag_mtd > 0.f) return fail(HRL_ERR_BAD_ARG, "hrl_set_goals: with flag_max_targets < 1 next_target() draws a goal near the robot and ignores the list (ant_flagrun_env.py:113-114): use hrl_next_target");
    if (n_goals < 1 || n_goals > h->cfg.flag_goal_capacity) return fail(HRL_ERR_BAD_ARG, "hrl_set_goals: n_goals must be within 1..flag_goal_capacity (" + std::to_string(h->cfg.flag_goal_capacity) + ")");
    hipLaunchKernelGGL(k_set_goals, dim3(h->dc.n_envs), dim3(64), 0, (hipStream_t)stream, to_dev(b, mask), (const DevCfg *)h->d_dc, goals_xy, (int)n_goals, (uint8_t *)nullptr);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? HRL_OK : hip_fail(e, "hrl_set_goals launch");
}

int hrl_next_target(hrl_handle *h, const hrl_buffers *b, const uint8_t *mask, uint8_t *ok, void *stream) {
    if (const int rc = check_call(h, b, "hrl_next_target")) return rc;
    if (!b->state || !b->aux || !b->obs) return fail(HRL_ERR_BAD_ARG, "hrl_next_target: null buffer");
    if (h->dc.kind != HRL_ANT_FLAGRUN) return fail(HRL_ERR_BAD_ARG, "hrl_next_target: only for AntFlagrun (ant_flagrun_env.py:112-120)");
    if (needs_items(h->dc) && !b->items) return fail(HRL_ERR_BAD_ARG, std::string("hrl_next_target: ") + items_why);
    hipLaunchKernelGGL(k_set_goals, dim3(h->dc.n_envs), dim3(64), 0, (hipStream_t)stream, to_dev(b, mask), (const DevCfg *)h->d_dc, (const float *)nullptr, 0, ok);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? HRL_OK : hip_fail(e, "hrl_next_target launch");
}

const char *hrl_last_error(void) { return g_err.c_str(); }
const char *hrl_backend(void) { return "hip-gfx950"; }

#if defined(HRL_STAMPS) || defined(HRL_WGTIME)
/* diagnostic builds only: device buffer of 4 x 24 u64 that k_step adds its per-phase cycle sums into */
void hrl_debug_set_stamps(unsigned long long *dev16) { g_stamps = dev16; }
#endif

}  // extern "C"
