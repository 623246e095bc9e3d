/* TEST INFRASTRUCTURE: drives the lock-step host executor of csrc/step_core.h (emu_lib.cpp, four host threads per group, exactly the
 * control flow of the kernel's workgroup) under ThreadSanitizer.  The threads share the group's records the way the waves share
 * LDS: any pair of accesses that the kernel's barriers do not order shows up as a data race here.
 * Build + run: tests/test_emu_asan.py::test_group_threads_under_thread_sanitizer */
#include "emu_lib.cpp"

#include <cstdio>
#include <random>
#include <vector>

int main() {
    const int kinds[] = {1, 0, 2, 4, 5, 1};
    int pass = 0;
    for (int kind : kinds) {
        hrl_config cfg;
        if (emu_default_config(kind, &cfg) != 0) return 2;
        const int n = 10; /* two full groups and a ragged third */
        cfg.num_envs = n; cfg.seed = 4; cfg.auto_reset = 1; cfg.max_episode_steps = 25;
        if (++pass == 6) { /* the second gather pass: 40 items in a crowded arena, pickup by contact, positions in a 74-wide observation */
            cfg.n_food = 24; cfg.n_poison = 16; cfg.n_bins = 12; cfg.use_sensor = 0; cfg.robot_coll_dist = 0.f; cfg.world_size[0] = cfg.world_size[1] = 5.f;
            cfg.robot_object_spacing = 0.3f;
        }
        const int od = emu_obs_dim(&cfg), ad = emu_act_dim(&cfg), is = emu_items_stride(&cfg);
        std::vector<float> state(n * 32), items((size_t)n * is), obs((size_t)n * od), rew(n), info(n * 4), act((size_t)n * ad), fin((size_t)n * od);
        std::vector<int32_t> aux(n * 4);
        std::vector<uint8_t> done(n), trunc(n);
        hrl_buffers b{sizeof(hrl_buffers), state.data(), items.data(), aux.data(), act.data(), obs.data(), rew.data(), done.data(), info.data(), fin.data(), trunc.data(), nullptr, nullptr};
        if (emu_reset(&cfg, &b, nullptr, 0) != 0) return 3;
        std::mt19937 rng(1);
        std::uniform_real_distribution<float> u(-1.f, 1.f);
        for (int t = 0; t < 60; ++t) {
            for (auto &a : act) a = u(rng);
            if (emu_step(&cfg, &b, 0) != 0) return 4;
        }
        double s = 0; for (float v : state) s += v;
        std::printf("kind %d: 60 steps of %d envs, checksum %.6f\n", kind, n, s);
    }
    std::printf("tsan-ok\n");
    return 0;
}
