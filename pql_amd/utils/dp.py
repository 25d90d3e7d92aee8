"""Data-parallel layout of the PQL path: which slice of the job a rank owns, and which communicator each
component's collectives ride on.

The reference has no data parallelism (its three Ray actors share GPUs by function, scripts/train_pql.py:41-70); BASELINE
configs[3] asks for it at 4/8 GPUs: envs are independent units, so rank r of G owns envs [r N/G, (r+1) N/G), the n-step
windows of those envs, a replay shard of memory_size / G rows filled only from them, and draws batch_size / G samples per
gradient step from it (equal shards -> the union of the draws is uniform over the global ring in distribution; the mean of
the per-rank batch means is the global batch mean).

    scaling = "strong"  the cfg's num_envs / memory_size / batch_size are the JOB's: each rank takes 1/G of each
                        (BASELINE configs[3] as written: 16384 envs over 8 GPUs = 2048 per rank);
    scaling = "weak"    they are PER RANK: the job grows with G (global batch G x 8192).

Collectives: the V-learner's and the P-learner's gradient all-reduces and the rollout's running-statistics all-gather are
issued from three different HIP streams.  On ONE torch process group they would all be funnelled through that group's one
internal RCCL stream and serialise in issue order (a P all-reduce queued behind the V-learner's backward); each component
therefore gets its own communicator (`component_groups`), so the three queues stay independent on the device while the
issue order inside each communicator is the same on every rank (fixed-ratio loop)."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class Shard:
    world: int
    rank: int
    scaling: str
    num_envs: int        # per rank
    memory_size: int     # per rank
    batch_size: int      # per rank
    total_envs: int      # whole job
    env_offset: int      # first global env id of this rank

    @property
    def global_batch(self):
        return self.batch_size * self.world


def shard(num_envs, memory_size, batch_size, world, rank, scaling="strong") -> Shard:
    """Per-rank sizes.  Strong scaling refuses sizes that do not divide: a ragged split would make shards unequal, and with
    unequal shards neither 'uniform over the union' nor 'mean of rank means == global mean' holds."""
    num_envs, memory_size, batch_size, world, rank = int(num_envs), int(memory_size), int(batch_size), int(world), int(rank)
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} of world {world}")
    if scaling == "weak" or world == 1:
        return Shard(world, rank, "weak" if world > 1 else "single", num_envs, memory_size, batch_size, num_envs * world, rank * num_envs)
    if scaling != "strong":
        raise ValueError(f"scaling must be 'strong' or 'weak', got {scaling!r}")
    for name, v in (("num_envs", num_envs), ("batch_size", batch_size)):
        if v % world:
            raise ValueError(f"{name}={v} does not divide over {world} ranks (strong scaling needs equal shards)")
    per_mem = memory_size // world   # rows; a remainder of < world rows of capacity is dropped, every shard equal
    if per_mem < 1:
        raise ValueError(f"memory_size={memory_size} is smaller than the world size {world}")
    return Shard(world, rank, "strong", num_envs // world, per_mem, batch_size // world, num_envs, rank * (num_envs // world))


def init_data_parallel(backend="nccl", share_gpu=False, local_rank=0):
    """torch.distributed set-up of one data-parallel rank; returns (process group, device index).  backend "nccl" = RCCL over xGMI,
    one GPU per rank (production).  "gloo" = the rehearsal backend: collectives are staged through host memory, so the ranks may
    share ONE card (`share_gpu`, every rank on cuda:0) -- RCCL refuses two ranks on one device -- which is how the entry point's
    data-parallel branch is exercised on a one-GPU box."""
    import torch
    import torch.distributed as dist
    backend = str(backend)
    if backend not in ("nccl", "gloo"):
        raise ValueError(f"algo.dp_backend must be 'nccl' (RCCL) or 'gloo' (rehearsal), got {backend!r}")
    if share_gpu and backend == "nccl":
        raise ValueError("algo.dp_share_gpu=True puts every rank on cuda:0, which RCCL refuses (duplicate device): set algo.dp_backend=gloo")
    index = 0 if share_gpu else int(local_rank)
    if index >= torch.cuda.device_count():
        raise RuntimeError(f"rank-local GPU {index} does not exist ({torch.cuda.device_count()} visible); "
                           "algo.dp_backend=gloo algo.dp_share_gpu=True rehearses on one card")
    torch.cuda.set_device(index)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{index}"))
    else:
        dist.init_process_group("gloo")
    return dist.group.WORLD, index


def broadcast_from_rank0(t, pg):
    """In-place broadcast of rank 0's tensor (replicated parameters start equal); gloo: staged through the host."""
    import torch.distributed as dist
    if dist.get_backend(pg) == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.broadcast(h, src=0, group=pg)
        t.copy_(h)
    else:
        dist.broadcast(t, src=0, group=pg)


def component_groups(pg, names=("v", "p", "rms")):
    """One communicator per collective-issuing component.  Every rank must call this at the same point (new_group is itself
    collective).  pg None -> {name: None}."""
    if pg is None:
        return {n: None for n in names}
    import torch.distributed as dist
    backend = dist.get_backend(pg)
    ranks = dist.get_process_group_ranks(pg)
    return {n: dist.new_group(ranks=ranks, backend=backend) for n in names}


# ------------------------------------------------------------------------------------------------------------------
# Gradient buckets (the all-reduce "overlapped with the last dW GEMMs").  The backward walks the layers from the
# head down; a layer's gradient is final as soon as its dW product's split slabs are summed, long before the first layer's
# launches.  `pqlk_mlp_backward_layers` runs the chain in pieces that end with that sum, and each piece's all-reduce is
# issued asynchronously on the component's communicator right behind it: the collective of the upper layers travels over
# xGMI while the MFMA launches of the layers below run.  Every arena element is the same fixed-order sum and the same
# rank-order reduction as in the one-bucket path, only in another launch: the gradient bits do not change.
def layer_buckets(n_layers):
    """[(layer_hi, layer_lo), ...] from the head down.  The head (a few hundred floats) shares the first bucket with the last
    hidden layer: a collective of its own would be all latency."""
    n_layers = int(n_layers)
    if n_layers < 3:
        return [(n_layers - 1, 0)]
    return [(n_layers - 1, n_layers - 2)] + [(l, l) for l in range(n_layers - 3, -1, -1)]


def bucket_views(grads, layout, layer_hi, layer_lo):
    """The arena ranges of layers layer_lo..layer_hi: one contiguous 1-D view of `grads` per net (layers sit in ascending
    order inside a net's block, nets one after the other)."""
    lo = layout.w_off[layer_lo]
    hi = layout.net_stride if layer_hi == layout.n_layers - 1 else layout.w_off[layer_hi + 1]
    return [grads[n * layout.net_stride + lo: n * layout.net_stride + hi] for n in range(layout.n_nets)]


def drain_pending_collectives(pg):
    """Call before a hipGraph capture that records RCCL collectives of `pg`: returns when the communicator's watchdog thread has
    retired every eager work.  The capture puts the communicator's internal stream into capture mode, and HIP then refuses
    hipEventQuery on every event last recorded on that stream -- including the end event of an EAGER collective issued before
    the capture (the warm-up run), which the watchdog is still polling until its next 100-ms sweep: `operation not permitted on
    an event last recorded in a capturing stream` then aborts the process (seen once in ~15 runs of the bucket test)."""
    if pg is None:
        return
    wait = getattr(pg, "_wait_for_pending_works", None)
    if wait is not None:
        wait()
    else:   # (older torch: two watchdog sweeps)
        import time
        import torch
        torch.cuda.synchronize()
        time.sleep(0.25)


class BucketAllReduce:
    """Issues the buckets' sum-all-reduces without waiting for them; `wait()` orders the current stream behind all of them
    (and is where the optimiser may start).  RCCL: one grouped launch per bucket (`allreduce_coalesced`: the per-net ranges of
    a bucket are not adjacent in the arena) on the communicator's own stream, which waits for the issuing stream's position at
    the call.  gloo (CPU tests, one-card rehearsals): staged through host memory, synchronous -- same sums."""

    def __init__(self, pg):
        import torch.distributed as dist
        self.pg = pg
        self.backend = dist.get_backend(pg)
        self.pending = []

    def issue(self, views):
        import torch
        import torch.distributed as dist
        if self.backend == "gloo" and views[0].is_cuda:
            for v in views:
                h = v.cpu()
                dist.all_reduce(h, group=self.pg)
                v.copy_(h)
            return
        opts = dist.AllreduceCoalescedOptions()
        opts.reduceOp = dist.ReduceOp.SUM
        if hasattr(opts, "asyncOp"):
            opts.asyncOp = True
        self.pending.append(self.pg.allreduce_coalesced(list(views), opts))

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []
