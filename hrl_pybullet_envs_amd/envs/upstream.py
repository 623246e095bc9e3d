"""What the in-tree reference code reaches for on UPSTREAM classes, kept as plain class attributes so that code written against the reference
finds it here.

`WalkerBaseBulletEnv` (pybullet_envs.gym_locomotion_envs) carries the cost weights of its `step()` as CLASS attributes, read
every step: reward = alive + progress + electricity_cost * mean|a * joint_speed| + stall_torque_cost * mean(a^2) + joints_at_limit_cost * n + 0.
`AntFlagrunBulletEnv.reset()` assigns 0 to three of them ON THE CLASS (ant_flagrun_env.py:133-135) -- for itself, and thereby for every other walker
env of the process (an AntMazeBulletEnv that steps after any flagrun reset runs without those costs).  The env classes of this package read the
attributes below before every step and hand a change to the kernel (hrl_update_config), so both the documented use
(`WalkerBaseBulletEnv.electricity_cost = ...` before training) and the side effect behave as in the reference."""


class WalkerBaseBulletEnv:
    electricity_cost = -2.0        # cost for using motors -- this parameter should be carefully tuned against reward for making progress (upstream's words)
    stall_torque_cost = -0.1       # cost for running electric current through a motor even at zero rotational speed
    foot_collision_cost = -1.0     # upstream: touches another leg (never charged: feet_collision_cost stays 0.0 in step())
    joints_at_limit_cost = -0.1    # discourage stuck joints


def walker_costs():
    c = WalkerBaseBulletEnv
    return float(c.electricity_cost), float(c.stall_torque_cost), float(c.joints_at_limit_cost)
