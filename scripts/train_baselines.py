#!/usr/bin/env python3
"""Single-process off-policy baseline loop (DDPG, SAC, CrossQ: `algo=ddpg_algo` / `sac_algo` / `crossq_algo`) -- same shape as the reference's scripts/train_baselines.py:39-72:
warm-up rollout -> replay, then per iteration: rollout, insert, `agent.update_net(memory)`.
    python scripts/train_baselines.py algo=ddpg_algo task.name=Toy num_envs=64 algo.batch_size=256 algo.memory_size=100000 max_step=20000
"""
import os
import sys
import time
from itertools import count

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pql_amd.algo import alg_name_to_path  # noqa: E402
from pql_amd.envs.synthetic import create_task_env  # noqa: E402
from pql_amd.replay.simple_replay import ReplayBuffer  # noqa: E402
from pql_amd.utils.cfg import load_cfg  # noqa: E402
from pql_amd.utils.common import capture_keyboard_interrupt, load_class_from_path, preprocess_cfg, set_random_seed  # noqa: E402
from pql_amd.utils.logger import MetricLogger  # noqa: E402


def main(cfg):
    set_random_seed(cfg.seed)
    capture_keyboard_interrupt()
    if cfg.device == "cuda":
        cfg.device = cfg.sim_device = cfg.rl_device = "cuda:0"
    cfg.algo.v_learner_gpu = cfg.algo.p_learner_gpu = 0
    preprocess_cfg(cfg)
    env = create_task_env(cfg)
    algo_name = cfg.algo.name if "Agent" in cfg.algo.name else "Agent" + cfg.algo.name
    agent = load_class_from_path(algo_name, alg_name_to_path[algo_name])(env=env, cfg=cfg)
    logger = MetricLogger(cfg.logging.get("jsonl") if cfg.get("logging") else None)
    start, global_steps = time.time(), 0
    agent.reset_agent()
    memory = ReplayBuffer(capacity=int(cfg.algo.memory_size), obs_dim=agent.obs_dim, action_dim=agent.action_dim, device=cfg.device)
    trajectory, steps = agent.explore_env(env, cfg.algo.warm_up, random=True)
    memory.add_to_buffer(trajectory)
    global_steps += steps
    log_info = {}
    for iter_t in count():
        trajectory, steps = agent.explore_env(env, cfg.algo.horizon_len, random=False)
        global_steps += steps
        memory.add_to_buffer(trajectory)
        log_info = agent.update_net(memory)
        if iter_t % cfg.algo.log_freq == 0:
            log_info["global_steps"] = global_steps
            logger.log(log_info, global_steps)
        if (cfg.max_step is not None and global_steps > cfg.max_step) or (cfg.max_step is None and time.time() - start > cfg.max_time):
            break
    return {**log_info, "global_steps": global_steps, "iters": iter_t + 1}


if __name__ == "__main__":
    print(main(load_cfg(sys.argv[1:])))
