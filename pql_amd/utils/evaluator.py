"""Policy evaluation beside training -- drop-in for `pql/utils/evaluator.py` (`Evaluator(cfg, wandb_run, rollout_callback,
create_task_env_func)`, `.eval_policy(policy, value, step, normalizer)`, `.parent.poll()/.recv()`,
`.check_if_should_stop(step)`, `.start_time`; call sites `scripts/train_pql.py:55,171-187`).

What one evaluation computes is the reference's `default_rollout` (evaluator.py:41-121): fresh trackers of capacity
`eval_num_envs`, `env.reset()`, `max_episode_length` steps of `actor(normalizer.normalize(obs))` (actor side: no clamp),
returns / lengths of the episodes that finish pushed into the trackers in env order, result
`{'eval/return', 'eval/episode_length'}` = tracker means (zero-filled windows, like `common.Tracker`), and the best-so-far
policy saved to `<run dir>/model.pth` in the reference checkpoint format.

How it runs is different by design.  The reference forks a second process that builds its own simulator and receives
cloudpickled nn.Modules through a pipe; on one MI355X that is a second HIP context competing for the same CUs.  Here the
default engine is IN-PROCESS: the evaluation is a launch sequence on its own HIP stream, fed from snapshots of the actor
arena / critic arena / normaliser taken at `eval_policy` time (event-fenced arena copies), and advanced cooperatively --
every `parent.poll()` of the training loop enqueues the next `eval_steps_per_poll` env steps and returns True once a
finished evaluation's results have landed in pinned host memory.  Nothing blocks the host except `recv()` on an
unfinished job.  Episode bookkeeping uses the rollout's `DeviceTracker` (masked scatters, no `torch.where(done)[0]` host
sync per step).  A custom `rollout_callback`, or `cfg.eval_subprocess=True`, selects the reference's process + pipe
topology instead; the message is `[pickle(policy spec), pickle(value spec), step, normalizer states]` with specs from
`module_to_spec` (class name, constructor arguments, CPU state_dict) -- plain data, no code objects.
"""
from __future__ import annotations

import multiprocessing as mp
import os
import pickle
import time
from collections import deque
from copy import deepcopy

import numpy as np
import torch

from pql_amd.utils.model_util import save_model


# ------------------------------------------------------------------------------------------------ specs (subprocess mode)
def module_to_spec(module):
    """(class name, constructor kwargs, CPU state_dict) of a pql_amd model, or None."""
    if module is None:
        return None
    return dict(cls=type(module).__name__, kwargs=dict(getattr(module, "init_kwargs", {})),
                state={k: v.detach().cpu() for k, v in module.state_dict().items()})


def spec_to_module(spec, device):
    if spec is None:
        return None
    import pql_amd.models.mlp as M
    kwargs = dict(spec["kwargs"])
    if "device" in kwargs:
        kwargs["device"] = device
    module = getattr(M, spec["cls"])(**kwargs).to(device)
    module.load_state_dict(spec["state"])
    return module


class _Normalizer:
    """Snapshot of RunningMeanStd: `.normalize(x)` without clamp (torch_util.py:83-85), `.get_states()`."""

    def __init__(self, states, device):
        mean, var, eps = states
        self.mean, self.var, self.epsilon = mean.to(device).clone(), var.to(device).clone(), float(eps)

    def normalize(self, x):
        return (x - self.mean) / torch.sqrt(self.var + self.epsilon)

    def get_states(self, device=None):
        return self.mean, self.var, self.epsilon


def _run_dir(cfg, wandb_run):
    d = getattr(wandb_run, "dir", None)
    if d is None and cfg.get("logging") is not None:
        d = cfg.logging.get("dir")
    d = d or "."
    os.makedirs(d, exist_ok=True)
    return d


# ------------------------------------------------------------------------------------------------ one evaluation
class _Job:
    """One evaluation in flight: device state + how far its episode has been enqueued."""

    def __init__(self, actor, critic, normalizer, step, num_envs, device):
        from pql_amd.algo.pql_actor import DeviceTracker
        self.actor, self.critic, self.normalizer, self.step = actor, critic, normalizer, step
        self.return_tracker = DeviceTracker(num_envs, device)
        self.step_tracker = DeviceTracker(num_envs, device)
        self.returns = torch.zeros(num_envs, dtype=torch.float32, device=device)
        self.lengths = torch.zeros(num_envs, dtype=torch.float32, device=device)
        self.obs = None
        self.i_step = 0
        self.done_event = None
        self.host = None   # pinned copies of the two tracker windows


class RolloutEngine:
    """default_rollout (evaluator.py:41-121) as a resumable launch sequence on `stream` (None = the current stream / CPU)."""

    def __init__(self, cfg, wandb_run=None, create_task_env_func=None):
        self.cfg = cfg
        cfg.headless = cfg.eval_headless
        if create_task_env_func is None:
            from pql_amd.envs.synthetic import create_task_env as create_task_env_func
        self.num_envs = int(cfg.eval_num_envs)
        self.env = create_task_env_func(cfg, num_envs=self.num_envs)
        self.max_step = int(self.env.max_episode_length)
        self.device = torch.device(cfg.device)
        self.stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        self.run_dir = _run_dir(cfg, wandb_run)
        self.wandb_run = wandb_run
        self.ret_max = float("-inf")
        if cfg.info_track_keys is not None:
            raise NotImplementedError("info_track_keys needs a simulator's info dict; out of scope")

    def _ctx(self):
        return torch.cuda.stream(self.stream) if self.stream is not None else _NullCtx()

    def start(self, actor, critic, normalizer_states, step):
        """Snapshot the inputs (on the caller's stream, so they are ordered after the training work that produced them)
        and open a job; the eval stream waits for the snapshot."""
        to_dev = lambda m: m.to(self.device) if hasattr(m, "to") else m   # noqa: E731  (a bare callable is allowed as policy)
        actor = to_dev(deepcopy(actor))
        critic = to_dev(deepcopy(critic)) if critic is not None else None
        normalizer = _Normalizer(normalizer_states, self.device) if normalizer_states is not None else None
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with self._ctx():
            return _Job(actor, critic, normalizer, step, self.num_envs, self.device)

    @torch.no_grad()
    def advance(self, job, n_steps):
        """Enqueue up to n_steps env steps of the episode; on the last one queue the read-back.  Returns True when the
        whole episode has been enqueued."""
        if job.i_step >= self.max_step:
            return True
        with self._ctx():
            if job.i_step == 0:
                job.obs = self.env.reset()
            for _ in range(min(n_steps, self.max_step - job.i_step)):
                x = job.normalizer.normalize(job.obs) if (self.cfg.algo.obs_norm and job.normalizer is not None) else job.obs
                action = job.actor(x)
                next_obs, reward, done, _info = self.env.step(action)
                job.returns += reward
                job.lengths += 1
                finished = done.bool()
                job.return_tracker.update(job.returns, finished)
                job.step_tracker.update(job.lengths, finished)
                job.returns.masked_fill_(finished, 0)
                job.lengths.masked_fill_(finished, 0)
                job.obs = next_obs
                job.i_step += 1
            if job.i_step >= self.max_step:
                windows = torch.stack((job.return_tracker.ring[: self.num_envs], job.step_tracker.ring[: self.num_envs]))
                if self.stream is not None:
                    job.host = torch.empty(windows.shape, dtype=windows.dtype, pin_memory=True)
                    job.host.copy_(windows, non_blocking=True)
                    job.done_event = torch.cuda.Event()
                    job.done_event.record(self.stream)
                else:
                    job.host = windows.clone()
        return job.i_step >= self.max_step

    def ready(self, job):
        return job.host is not None and (job.done_event is None or job.done_event.query())

    def finish(self, job):
        """Blocks until the job's results are on the host; returns the reference's result dict and keeps the best model."""
        self.advance(job, self.max_step)
        if job.done_event is not None:
            job.done_event.synchronize()
        w = job.host.numpy().astype(np.float64)
        ret_mean, step_mean = float(np.mean(w[0])), float(np.mean(w[1]))
        result = {"eval/return": ret_mean, "eval/episode_length": step_mean}
        if ret_mean > self.ret_max:
            self.ret_max = ret_mean
            has_sd = lambda m: m is not None and hasattr(m, "state_dict")   # noqa: E731
            save_model(path=os.path.join(self.run_dir, "model.pth"), actor=job.actor if has_sd(job.actor) else {},
                       critic=job.critic if has_sd(job.critic) else {},
                       rms=job.normalizer.get_states() if (self.cfg.algo.obs_norm and job.normalizer is not None) else None,
                       wandb_run=self.wandb_run)
        return result


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _Mailbox:
    """The training loop's end of the reference's pipe (`evaluator.parent`): poll() / recv(), here backed by the engine."""

    def __init__(self, engine, steps_per_poll):
        self.engine, self.steps_per_poll = engine, int(steps_per_poll)
        self.jobs = deque()

    def poll(self, timeout=None):
        if not self.jobs:
            return False
        for job in self.jobs:   # FIFO like the child's recv loop: only the oldest unfinished job is advanced
            if job.i_step < self.engine.max_step:
                self.engine.advance(job, self.steps_per_poll)
                break
        return self.engine.ready(self.jobs[0])

    def recv(self):
        if not self.jobs:
            raise EOFError("no evaluation was requested")
        return self.engine.finish(self.jobs.popleft())

    def pending(self):
        return len(self.jobs)


class _PipeMailbox:
    """The parent end of the pipe in subprocess mode, plus a count of the evaluations still owed."""

    def __init__(self, conn):
        self.conn, self.owed = conn, 0

    def send(self, msg):
        self.owed += 1
        self.conn.send(msg)

    def poll(self, timeout=0.0):
        return self.conn.poll(timeout)

    def recv(self):
        out = self.conn.recv()
        self.owed -= 1
        return out

    def pending(self):
        return self.owed


# ------------------------------------------------------------------------------------------------ subprocess flavour
def default_rollout(cfg, wandb_run, child, create_task_env_func=None):
    """Child-process body with the reference's protocol (evaluator.py:41-121): recv [policy, value, step, normalizer],
    run one evaluation, send the result dict; a None policy ends the loop."""
    engine = RolloutEngine(cfg, wandb_run, create_task_env_func)
    while True:
        actor, critic, step, normalizer = child.recv()
        actor, critic = pickle.loads(actor), pickle.loads(critic)
        if actor is None:
            break
        job = engine.start(spec_to_module(actor, engine.device), spec_to_module(critic, engine.device), normalizer, step)
        child.send(engine.finish(job))
    child.close()


class Evaluator:
    def __init__(self, cfg, wandb_run=None, rollout_callback=None, create_task_env_func=None, enabled=True):
        """`enabled=False` (ranks other than 0 under data parallelism) keeps only the stop criterion."""
        cfg = deepcopy(cfg)
        self.cfg = cfg
        self.process = None
        self.start_time = time.time()
        if not enabled:
            self.engine, self.parent = None, _Mailbox(None, 0)
            return
        if rollout_callback is not None or bool(cfg.get("eval_subprocess")):
            conn, self.child = mp.Pipe()
            self.parent = _PipeMailbox(conn)
            ctx = mp.get_context("spawn")
            self.process = ctx.Process(target=rollout_callback or default_rollout, args=(cfg, wandb_run, self.child, create_task_env_func),
                                       daemon=True)
            self.process.start()
        else:
            self.engine = RolloutEngine(cfg, wandb_run, create_task_env_func)
            self.parent = _Mailbox(self.engine, cfg.get("eval_steps_per_poll") or 32)

    def eval_policy(self, policy, value, step=0, normalizer=None):
        states = normalizer.get_states() if normalizer is not None else None
        if self.process is not None:
            states = None if states is None else tuple(t.detach().cpu() if torch.is_tensor(t) else t for t in states)
            self.parent.send([pickle.dumps(module_to_spec(policy)), pickle.dumps(module_to_spec(value)), step, states])
        else:
            self.parent.jobs.append(self.engine.start(policy, value, states, step))

    def check_if_should_stop(self, step=None):
        if self.cfg.max_step is not None:
            return step > self.cfg.max_step
        return (time.time() - self.start_time) > self.cfg.max_time

    def close(self):
        """Ends the child process, if there is one (the reference relies on `daemon=True` alone)."""
        if self.process is not None:
            self.parent.conn.send([pickle.dumps(None), pickle.dumps(None), 0, None])
            self.process.join(timeout=30)
