#!/usr/bin/env python3
"""Parallel Q-Learning entry point -- same command line as the reference's scripts/train_pql.py
(`python scripts/train_pql.py task=AllegroHand algo.distl=True algo.num_gpus=1 ...`), same cfg keys, same
metric names; the Isaac-Gym rollout is replaced by the synthetic vectorised env (task.name picks the shapes).

Orchestration (SURVEY 3.1 -> MI355X): instead of three Ray processes exchanging pickled nn.Modules through the
object store, ONE process drives three launch queues (HIP streams; with `algo.num_gpus=2` the learners' queues sit
on another GPU and every hand-off crosses xGMI through dedicated copy streams).  Hand-offs are event-fenced and
double-buffered (pql_amd/utils/handoff.py); the host never waits for the GPU.  Two ways to keep the components at
the design ratios 1 : critic_sample_ratio : critic_sample_ratio / critic_actor_ratio:

  algo.async_learners=False (default)  the loop itself issues exactly `critic_sample_ratio` V-steps and the matching
      P-steps per rollout iteration -- the fixed point the reference's controller converges to, without sleeping;
  algo.async_learners=True             the reference's topology: the learners free-run in threads (`asyn_v_learner`
      / `asyn_p_learner`) and the reference's sliding-window controller (train_pql.py:127-158, here
      pql_amd.utils.ratio_control.RatioController) tells whoever is too fast how long to sleep per unit.

With torchrun (WORLD_SIZE > 1) the env axis and the replay shard data-parallel and gradients are all-reduced over
RCCL (fixed-ratio loop only: free-running ranks would issue their collectives out of step); `algo.dp_backend=gloo
algo.dp_share_gpu=True` runs the same branch with host-staged collectives and every rank on cuda:0 (rehearsal on a one-GPU box:
RCCL refuses two ranks on one device).  `algo.dp_global=True`
(default) reads num_envs / algo.memory_size / algo.batch_size as the JOB's sizes and gives every rank 1/G of each, so
BASELINE configs[3] (`num_envs=16384` on 8 GPUs) runs as written; `algo.dp_global=False` reads them per rank (weak scaling).
V-learner, P-learner and the running statistics each get their own RCCL communicator (pql_amd/utils/dp.py).
"""
import os
import sys
import threading
import time
from itertools import count

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import pql_amd  # noqa: E402,F401
from pql_amd.algo.pql_actor import PQLActor  # noqa: E402
from pql_amd.algo.pql_p_learner import PQLPLearner, asyn_p_learner  # noqa: E402
from pql_amd.algo.pql_v_learner import PQLVLearner, asyn_v_learner  # noqa: E402
from pql_amd.envs.synthetic import create_task_env  # noqa: E402
from pql_amd.utils.cfg import load_cfg  # noqa: E402
from pql_amd.utils.common import capture_keyboard_interrupt, preprocess_cfg, set_random_seed  # noqa: E402
from pql_amd.utils.dp import broadcast_from_rank0, component_groups, init_data_parallel, shard  # noqa: E402
from pql_amd.utils.evaluator import Evaluator  # noqa: E402
from pql_amd.utils.logger import MetricLogger  # noqa: E402
from pql_amd.utils import rng as R  # noqa: E402
from pql_amd.utils.ratio_control import RatioController  # noqa: E402


def agree_to_stop(stop, pg, device):
    """Data parallel: every rank must leave the loop in the same iteration (the next one opens with collectives), and a
    wall-clock criterion can differ between ranks -- take rank 0's verdict."""
    if pg is None:
        return stop
    flag = torch.tensor([1.0 if stop else 0.0], device=device if torch.distributed.get_backend(pg) == "nccl" else "cpu")
    torch.distributed.broadcast(flag, src=0, group=pg)
    return bool(flag.item())


def main(cfg):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    pg = None
    if world > 1:
        # algo.dp_backend=nccl (RCCL, one GPU per rank); gloo + algo.dp_share_gpu=True rehearses this branch on ONE card
        pg, local = init_data_parallel(cfg.algo.get("dp_backend", "nccl"), bool(cfg.algo.get("dp_share_gpu", False)), local)
        cfg.device = cfg.sim_device = cfg.rl_device = f"cuda:{local}"
        cfg.algo.v_learner_gpu = cfg.algo.p_learner_gpu = local
        cfg.algo.num_gpus = 1
    if bool(cfg.algo.get("async_learners", False)):
        # free-running learners meet a hand-off after an arbitrary number of steps, and every hand-off discards what was gathered
        # ahead (~100 MB at cfg #2 with 8 steps ahead): two steps ahead keep the launch-count benefit without that waste
        for key in ("prefetch_steps", "prefetch_steps_p"):
            if cfg.algo.get(key) is None:
                cfg.algo[key] = 2
    sh = shard(cfg.num_envs, cfg.algo.memory_size, cfg.algo.batch_size, world, rank,
               "strong" if bool(cfg.algo.get("dp_global", True)) else "weak")
    cfg.num_envs, cfg.algo.memory_size, cfg.algo.batch_size = sh.num_envs, sh.memory_size, sh.batch_size
    groups = component_groups(pg)   # (collective: every rank, same point)
    if cfg.algo.num_gpus == 1 and world == 1:
        cfg.algo.v_learner_gpu = 0
        cfg.algo.p_learner_gpu = 0
    if cfg.sim_device == "cuda":
        cfg.sim_device = cfg.device = cfg.rl_device = "cuda:0"
    preprocess_cfg(cfg)
    capture_keyboard_interrupt()
    set_random_seed(cfg.seed + rank)
    env = create_task_env(cfg, env_offset=sh.env_offset)
    sim_device = torch.device(cfg.sim_device)
    v_dev = torch.device(f"cuda:{cfg.algo.v_learner_gpu}")
    p_dev = torch.device(f"cuda:{cfg.algo.p_learner_gpu}")

    if cfg.algo.num_gpus > 1 and world == 1 and v_dev != sim_device and not torch.cuda.can_device_access_peer(v_dev.index, sim_device.index):
        # torch's cross-device copy_ then stages through host memory and synchronises: the copy-stream hand-offs still work,
        # but the "learner stream never waits for the link" property is gone -- say so instead of degrading silently
        print(f"[train_pql] warning: {sim_device} and {v_dev} have no peer access; hand-offs will be host-staged", file=sys.stderr)
    pql_actor = PQLActor(env, cfg, env_offset=sh.env_offset, total_envs=sh.total_envs)
    v_learner = PQLVLearner(env.observation_space.shape, env.action_space.shape[0], cfg, process_group=groups["v"])
    p_learner = PQLPLearner(env.observation_space.shape, env.action_space.shape[0], cfg, process_group=groups["p"])
    if world > 1:
        for t in (v_learner.critic.arena.data, p_learner.actor.arena.data):
            broadcast_from_rank0(t, pg)
        v_learner.critic_target.arena.data.copy_(v_learner.critic.arena.data)
        if pql_actor.obs_rms is not None:
            pql_actor.obs_rms.pg = groups["rms"]
    if rank == 0:   # which source the learners' draws will come from (algo.rng; decided by the on-device check of pql_amd/utils/rng.py)
        want = "torch" if str(cfg.algo.get("rng", "auto")) == "torch" or bool(cfg.algo.get("graph_rng", False)) else (
            "philox (one launch per run of steps, torch's own numbers)" if R.verified(v_dev) is not None else "torch (the philox check failed on this device)")
        print(f"[train_pql] learner draws: {want}", file=sys.stderr)
    critic, critic_update_times, critic_loss = v_learner.start()
    actor, actor_update_times, actor_loss = p_learner.start()
    pql_actor.set_actor(actor)

    logger = MetricLogger(cfg.logging.get("jsonl") if cfg.get("logging") else None) if rank == 0 else None
    global_steps = 0
    # evaluation beside training (train_pql.py:55,171-185): rank 0 only; every rank keeps the stop criterion
    evaluator = Evaluator(cfg=cfg, wandb_run=None, enabled=rank == 0)
    pql_actor.reset_agent()
    p_data, v_data, steps = pql_actor.explore_env(env, cfg.algo.warm_up, random=True)
    global_steps += steps * world
    rms = (lambda dev: pql_actor.obs_rms.get_states(dev)) if pql_actor.obs_rms is not None else (lambda dev: None)
    critic, critic_loss, critic_update_times = v_learner.update(actor, v_data, rms(v_dev), 0)
    actor, actor_loss, actor_update_times = p_learner.update(critic, p_data, rms(p_dev), 0)

    free_running = bool(cfg.algo.get("async_learners", False))
    if free_running and world > 1:
        raise ValueError("algo.async_learners=True is single-process only: data-parallel ranks must issue their "
                         "gradient all-reduces in step, which the fixed-ratio loop guarantees")
    stop_learners, threads, ctl = threading.Event(), [], None
    if free_running:
        # the reference's learners are separate processes with their own RNG streams
        v_learner.use_private_rng(cfg.seed + 1)
        p_learner.use_private_rng(cfg.seed + 2)
        depth = int(cfg.algo.get("max_in_flight", 2))
        threads = [threading.Thread(target=asyn_v_learner, args=(v_learner, cfg, stop_learners, depth), daemon=True),
                   threading.Thread(target=asyn_p_learner, args=(p_learner, cfg, stop_learners, depth), daemon=True)]
        for t in threads:
            t.start()
        ctl = RatioController(cfg.algo.critic_sample_ratio, cfg.algo.critic_actor_ratio, critic_update_times, actor_update_times)

    v_per_iter = int(cfg.algo.critic_sample_ratio)
    p_every = int(cfg.algo.critic_actor_ratio)
    critic_wait = actor_wait = 0
    for iter_t in count():
        p_data, v_data, steps = pql_actor.explore_env(env, cfg.algo.horizon_len, random=False)
        global_steps += steps * world
        # hand-offs (train_pql.py:111-119): newest policy + transitions -> V-learner, newest critic + obs -> P-learner,
        # newest policy -> rollout replica.  What comes back are snapshots, so nobody reads an arena that is being stepped.
        critic, critic_loss, critic_update_times = v_learner.update(actor, v_data, rms(v_dev), critic_wait)
        actor, actor_loss, actor_update_times = p_learner.update(critic, p_data, rms(p_dev), actor_wait)
        pql_actor.set_actor(actor)
        if free_running:
            sim_wait, critic_wait, actor_wait = ctl.observe(critic_update_times, actor_update_times)
            if sim_wait > 0:
                time.sleep(sim_wait)
        else:
            # (issued per learner: the steps of one learner between two hand-offs are ONE hipGraph on its queue -- learn_many --
            #  and the two queues are independent, so this is the interleaved V, V, P, V, V, P ... order as far as results go.
            #  With algo.streams=False there is one queue: the 4 P steps then start after all 8 V steps.  Under data parallelism
            #  the per-iteration order of collectives is V x 8 then P x 4 on every rank -- each learner on its own communicator)
            v_learner.learn_many(v_per_iter)
            p_learner.learn_many(v_per_iter // p_every)
        if rank == 0 and evaluator.parent.poll():
            logger.log(evaluator.parent.recv(), global_steps)
        if rank == 0 and iter_t % cfg.algo.log_freq == 0:
            log_info = {
                "train/critic_loss": critic_loss, "train/actor_loss": actor_loss,
                "train/return": pql_actor.return_tracker.mean(), "train/episode_length": pql_actor.step_tracker.mean(),
                "train/critic_update_times": critic_update_times, "train/actor_update_times": actor_update_times,
                "train/global_steps": global_steps}
            logger.log(log_info, global_steps)
            if iter_t % cfg.algo.eval_freq == 0:
                logger.table(global_steps, log_info)
        if rank == 0 and iter_t % cfg.algo.eval_freq == 0:
            evaluator.eval_policy(pql_actor.actor, critic, normalizer=pql_actor.obs_rms, step=global_steps)
        stop = evaluator.check_if_should_stop(global_steps)
        if cfg.max_step is None:
            stop = agree_to_stop(stop, pg, sim_device)
        if stop:
            break
    stop_learners.set()
    for t in threads:
        t.join()
    if rank == 0:
        while evaluator.parent.pending():   # evaluations still in flight when training stops
            logger.log(evaluator.parent.recv(), global_steps)
        evaluator.close()
    torch.cuda.synchronize()
    import hashlib
    sha = lambda t: hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]   # noqa: E731  (replicas must stay bit-equal)
    fingerprints = dict(critic_sha=sha(v_learner.critic.arena.data), critic_target_sha=sha(v_learner.critic_target.arena.data),
                        actor_sha=sha(p_learner.actor.arena.data))
    if world > 1:
        torch.distributed.barrier(group=pg)
        torch.distributed.destroy_process_group()
    return dict(rank=rank, world=world, **fingerprints, global_steps=global_steps, critic_updates=v_learner.update_count, actor_updates=p_learner.update_count,
                critic_loss=v_learner.loss_mean(), actor_loss=p_learner.loss_mean(), rollout_iterations=iter_t + 1, waits=(None if ctl is None else (ctl.sim_wait_time, ctl.critic_wait_time,
                                                                                 ctl.actor_wait_time)))


if __name__ == "__main__":
    import json
    result = main(load_cfg(sys.argv[1:]))
    sys.stdout.flush()
    os.write(1, ("TRAIN_PQL_RESULT " + json.dumps(result) + "\n").encode())   # one write: the ranks of a torchrun job share the pipe
