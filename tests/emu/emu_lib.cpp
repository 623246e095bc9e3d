/*
 * emu_lib.cpp -- TEST INFRASTRUCTURE: lock-step host executor for the phases of csrc/step_core.h.
 *
 * Runs the product's wave phases lane by lane on the CPU so that the kernel logic (indexing, phase ordering,
 * LDS layout) can be checked against the oracle -- under AddressSanitizer/UBSan -- in a container without a GPU,
 * before the same code is launched on an MI355X.  The product never loads this library.
 * `reverse` runs the lanes of every phase in descending order: results must not depend on lane order, which
 * exposes same-phase cross-lane LDS dependencies that a real wave (and the forward emulation) would hide.
 */
#define HRL_EMU 1
#include <pthread.h>

#include <thread>
#include <vector>

#include "../../hrl_pybullet_envs_amd/csrc/host_cfg.h"

using namespace hrl;

/* A workgroup of G env-waves: their LDS records and the workgroup barrier.  G = 1: the one-wave-per-env form.  G = 4: four
 * host threads run the four waves of a group concurrently, exactly as the kernel's workgroup does (step_core.h,
 * ant_group_block): the leader wave executes the group block on all four records while the others wait at the barrier. */
struct CpuGroup {
    int G;
    WaveLds L[4];
    pthread_barrier_t bar;
    explicit CpuGroup(int g) : G(g) { memset(L, 0x7f, sizeof(L)); if (G > 1) pthread_barrier_init(&bar, nullptr, (unsigned)G); } /* poison LDS with large finite floats */
    ~CpuGroup() { if (G > 1) pthread_barrier_destroy(&bar); }
};

struct CpuExec {
    CpuGroup &grp;
    int wave;
    LaneRegs regs[64];
    bool reverse = false;
    CpuExec(CpuGroup &g, int w) : grp(g), wave(w) { memset(regs, 0, sizeof(regs)); }
    WaveLds &lds() { return grp.L[grp.G == 1 ? 0 : wave]; }
    WaveLds &lds(int k) { return grp.L[grp.G == 1 ? 0 : k]; }
    void group_sync() { if (grp.G > 1) pthread_barrier_wait(&grp.bar); }
    template <class F> void leader(F f) { if (grp.G == 1 || wave == 0) each(f); }
    LaneRegs &reg(int lane) { return regs[lane]; }
    int uniform(int v) { return v; }
    int wave_index() const { return wave; }
    int group_size() const { return grp.G; }
    void refresh() {}
    int slot() const { return 0; } /* scheduling hints of the device executor: no effect on results */
    void priority(int) const {}
    void refresh_uniform(int &) {}
    float lane_one(int lane, int r) { return lane == r ? 1.f : 0.f; }
    void stamp(int) {}
    void flush_stamps(const DevBufs &) {}
    template <class F> void each(F f) {
        if (!reverse) for (int lane = 0; lane < 64; ++lane) f(lane);
        else for (int lane = 63; lane >= 0; --lane) f(lane);
    }
    template <class P, class W, class Q> int each_compact(P pred, W write, Q post) {
        decltype(pred(0)) r[64];
        if (!reverse) for (int lane = 0; lane < 64; ++lane) r[lane] = pred(lane);
        else for (int lane = 63; lane >= 0; --lane) r[lane] = pred(lane);
        int rank[64], n = 0;
        for (int lane = 0; lane < 64; ++lane) { rank[lane] = n; if (r[lane].ok) ++n; }
        if (!reverse) { for (int lane = 0; lane < 64; ++lane) { if (r[lane].ok) write(lane, rank[lane], r[lane]); post(lane, r[lane]); } }
        else { for (int lane = 63; lane >= 0; --lane) { if (r[lane].ok) write(lane, rank[lane], r[lane]); post(lane, r[lane]); } }
        return n;
    }
    template <class P, class W, class Q> void each_compact16(P pred, W write, Q post) {
        decltype(pred(0)) r[64];
        if (!reverse) for (int lane = 0; lane < 64; ++lane) r[lane] = pred(lane);
        else for (int lane = 63; lane >= 0; --lane) r[lane] = pred(lane);
        int rank[64], cnt[4] = {0, 0, 0, 0};
        for (int lane = 0; lane < 64; ++lane) { rank[lane] = cnt[lane >> 4]; if (r[lane].ok) ++cnt[lane >> 4]; }
        if (!reverse) { for (int lane = 0; lane < 64; ++lane) { if (r[lane].ok) write(lane, rank[lane], r[lane]); post(lane, r[lane], cnt[lane >> 4]); } }
        else { for (int lane = 63; lane >= 0; --lane) { if (r[lane].ok) write(lane, rank[lane], r[lane]); post(lane, r[lane], cnt[lane >> 4]); } }
    }
    template <class P> unsigned long long each_ballot(P pred) {
        unsigned long long m = 0;
        if (!reverse) { for (int lane = 0; lane < 64; ++lane) if (pred(lane)) m |= 1ull << lane; }
        else { for (int lane = 63; lane >= 0; --lane) if (pred(lane)) m |= 1ull << lane; }
        return m;
    }
    template <class P, class C> void each_row(int src, P produce, C apply) {
        F2b v[64];
        if (!reverse) for (int lane = 0; lane < 64; ++lane) v[lane] = produce(lane);
        else for (int lane = 63; lane >= 0; --lane) v[lane] = produce(lane);
        regs[src].lam = v[src].ln;
        if (!reverse) for (int lane = 0; lane < 64; ++lane) apply(lane, v[src].dl);
        else for (int lane = 63; lane >= 0; --lane) apply(lane, v[src].dl);
    }
    template <class V, class I, class C> void each_shuffle(V value, I index, C consume) {
        float v[64];
        if (!reverse) for (int lane = 0; lane < 64; ++lane) v[lane] = value(lane);
        else for (int lane = 63; lane >= 0; --lane) v[lane] = value(lane);
        if (!reverse) for (int lane = 0; lane < 64; ++lane) consume(lane, v[index(lane) & 63]);
        else for (int lane = 63; lane >= 0; --lane) consume(lane, v[index(lane) & 63]);
    }
};

static DevBufs to_dev(const hrl_buffers *b, const uint8_t *mask) {
    DevBufs d;
    d.state = b->state; d.items = b->items; d.aux = b->aux; d.actions = b->actions; d.obs = b->obs;
    d.reward = b->reward; d.done = b->done; d.info = b->info; d.mask = mask; d.stamps = nullptr;
    d.final_obs = b->final_obs; d.truncated = b->truncated; d.goal = b->goal; d.rows = b->solver_rows;
    return d;
}

extern "C" {
int emu_default_config(int32_t kind, hrl_config *c) { return default_config(kind, c); }
int emu_obs_dim(const hrl_config *c) { return obs_dim(c); }
int emu_act_dim(const hrl_config *c) { return act_dim(c); }
int emu_items_stride(const hrl_config *c) { return items_stride(c); }
const char *emu_validate(const hrl_config *c) { static std::string s; s = validate(c); return s.c_str(); }
int emu_hot_rows(const hrl_config *cfg) { DevCfg c; build_devcfg(*cfg, c); return c.hot_rows; } /* the straggler rule's threshold the host derives (host_cfg.h::standing_rows) */
int emu_reset(const hrl_config *cfg, const hrl_buffers *b, const uint8_t *mask, int reverse) {
    if (!validate(cfg).empty()) return HRL_ERR_BAD_ARG;
    DevCfg c; build_devcfg(*cfg, c);
    DevBufs d = to_dev(b, mask);
    for (int e = 0; e < cfg->num_envs; ++e) { CpuGroup g(1); CpuExec x(g, 0); x.reverse = reverse != 0; reset_dispatch(x, d, c, e); }
    return HRL_OK;
}
int emu_observe(const hrl_config *cfg, const hrl_buffers *b, const uint8_t *mask, int reverse) {
    if (!validate(cfg).empty()) return HRL_ERR_BAD_ARG;
    DevCfg c; build_devcfg(*cfg, c);
    DevBufs d = to_dev(b, mask);
    for (int e = 0; e < cfg->num_envs; ++e) { CpuGroup g(1); CpuExec x(g, 0); x.reverse = reverse != 0; observe_dispatch(x, d, c, e); }
    return HRL_OK;
}
/* group = envs per workgroup: 4 = the product's launch for the ant kinds (four host threads per group), 1 = one wave per env */
int emu_step_group(const hrl_config *cfg, const hrl_buffers *b, int reverse, int group) {
    if (!validate(cfg).empty() || (group != 1 && group != 4)) return HRL_ERR_BAD_ARG;
    DevCfg c; build_devcfg(*cfg, c);
    DevBufs d = to_dev(b, nullptr);
    if (group == 1 || cfg->env_kind == HRL_POINT_GATHER) {
        for (int e = 0; e < cfg->num_envs; ++e) { CpuGroup g(1); CpuExec x(g, 0); x.reverse = reverse != 0; step_dispatch(x, d, c, e); }
        return HRL_OK;
    }
    const int n_groups = (cfg->num_envs + 3) / 4;
    std::vector<CpuGroup *> groups;
    for (int k = 0; k < n_groups; ++k) groups.push_back(new CpuGroup(4));
    std::vector<std::thread> team;
    for (int w = 0; w < 4; ++w)
        team.emplace_back([&, w]() {
            for (int k = 0; k < n_groups; ++k) { CpuExec x(*groups[k], w); x.reverse = reverse != 0; step_dispatch(x, d, c, 4 * k + w); }
        });
    for (auto &t : team) t.join();
    for (auto *g : groups) delete g;
    return HRL_OK;
}
int emu_step(const hrl_config *cfg, const hrl_buffers *b, int reverse) { return emu_step_group(cfg, b, reverse, 4); }
int emu_set_goals(const hrl_config *cfg, const hrl_buffers *b, const float *goals_xy, int n_goals, const uint8_t *mask, int reverse) {
    if (!validate(cfg).empty() || cfg->env_kind != HRL_ANT_FLAGRUN || !cfg->flag_manual_goals || n_goals < 1 || n_goals > cfg->flag_goal_capacity) return HRL_ERR_BAD_ARG;
    DevCfg c; build_devcfg(*cfg, c);
    DevBufs d = to_dev(b, mask);
    if (cfg->flag_max_target_dist > 0.f) return HRL_ERR_BAD_ARG;
    for (int e = 0; e < cfg->num_envs; ++e) { CpuGroup g(1); CpuExec x(g, 0); x.reverse = reverse != 0; set_goals_entry(x, d, c, e, goals_xy, n_goals, nullptr); }
    return HRL_OK;
}
int emu_next_target(const hrl_config *cfg, const hrl_buffers *b, const uint8_t *mask, uint8_t *ok, int reverse) {
    if (!validate(cfg).empty() || cfg->env_kind != HRL_ANT_FLAGRUN) return HRL_ERR_BAD_ARG;
    DevCfg c; build_devcfg(*cfg, c);
    DevBufs d = to_dev(b, mask);
    for (int e = 0; e < cfg->num_envs; ++e) { CpuGroup g(1); CpuExec x(g, 0); x.reverse = reverse != 0; set_goals_entry(x, d, c, e, nullptr, 0, ok); }
    return HRL_OK;
}
int emu_lds_bytes(void) { return (int)sizeof(WaveLds); }
}

/* debug dump of the articulated-body quantities after phases K and B (same layout as orc_dyn_dump_f32 minus qdd) */
extern "C" void emu_dyn_dump(const hrl_config *cfg, const float *q, const float *u, const float *tau, float *out) {
    DevCfg c; build_devcfg(*cfg, c);
    CpuGroup grp(1);
    CpuExec x(grp, 0);
    WaveLds &L = x.lds();
    for (int i = 0; i < 16; ++i) { L.q[0][i] = i < 15 ? q[i] : 0.f; L.u[i] = i < 14 ? u[i] : 0.f; }
    for (int j = 0; j < 8; ++j) L.tau[j] = tau[j];
    x.each([&](int lane) { phase_kin_ankle(c, L, x.reg(lane), L.q[0], lane); });
    x.each([&](int lane) { phase_hip(c, L, x.reg(lane), lane); });
    x.each([&](int lane) { phase_leg_sum(L, lane); });
    x.each([&](int lane) { phase_base(L, x.reg(lane), lane); });
    int o = 0;
    for (int j = 0; j < 8; ++j) for (int k = 0; k < 6; ++k) out[o++] = L.S[j][k];
    for (int j = 0; j < 8; ++j) for (int k = 0; k < 6; ++k) out[o++] = L.U[j][k];
    for (int j = 0; j < 8; ++j) for (int k = 0; k < 6; ++k) out[o++] = L.cb[j][k];
    for (int j = 0; j < 8; ++j) out[o++] = L.invD[j];
    for (int j = 0; j < 8; ++j) out[o++] = L.uterm[j];
    for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) out[o++] = a > b ? L.Lb[tl(a, b)] : (a == b ? L.idb[a] : 0.f);
    for (int k = 0; k < 6; ++k) out[o++] = L.a0[k];
}

/* the product's specified fp32 transcendental functions on arrays: which = 0 sin, 1 cos, 2 atan2(a, b), 3 asin */
extern "C" void emu_spec_math(int which, const float *a, const float *b, float *out, int n) {
    for (int i = 0; i < n; ++i)
        out[i] = which == 0 ? sin_spec(a[i]) : which == 1 ? cos_spec(a[i]) : which == 2 ? atan2_spec(a[i], b[i]) : asin_spec(a[i]);
}
