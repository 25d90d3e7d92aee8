"""Event-fenced hand-offs between the three concurrent components of PQL (rollout, V-learner, P-learner).

The reference runs the components as Ray processes and moves data by pickling it through the object store
(scripts/train_pql.py:61-70,100-119): a hand-off is a deep copy, so it can never be torn.  Here the components
are launch queues (HIP streams, possibly on different GPUs) of ONE process that share device memory, so every
hand-off needs (1) a *ready* fence -- the consumer's stream waits until the producer's work is visible -- and
(2) a *release* fence -- the producer does not overwrite a buffer while a consumer is still reading it.  Both
are hipEvents; nothing here blocks the host.

Vocabulary
  Lease      ready event + the release events of everyone who read the data since.
  Block      a tuple of tensors carrying a lease (what `PQLActor.explore_env` returns: still "a 5-tuple").
  publish    ArenaPublisher: double-buffered snapshots of a parameter arena taken on the owner's stream; what
             `start()` / `update()` of a learner return when it runs on its own stream (the reference returns a
             pickled copy of the module at this point, pql_v_learner.py:122 through Ray).
  Shipper    src GPU -> dst GPU pipe: a dedicated copy stream on each end (hipMemcpyPeerAsync-class copies over
             xGMI, never on a learner's compute stream) into double-buffered landing blocks on the destination.

Anything WITHOUT a lease follows the usual stream convention: it is valid on the caller's current stream of its
device, and after the hand-off the caller's stream is made to wait (on the device, not the host) for the consumer.
"""
from __future__ import annotations

import os
import threading

import torch

LOCK = threading.RLock()   # lease bookkeeping + the enqueue of the copy it guards are one atomic step between threads
CAPTURE_LOCK = threading.Lock()   # one hipGraph capture at a time per process (learner threads capture lazily)
# Rehearsal switch (tests / one-GPU boxes): route same-device hand-offs through the copy streams and landing blocks as
# if the learners sat on another GPU, so the two-GPU code path runs on one card.
FORCE_SHIP = os.environ.get("PQL_FORCE_SHIP", "0") == "1"


def crosses(src, dst):
    """True when a hand-off from device `src` to device `dst` goes through a Shipper."""
    return FORCE_SHIP or torch.device(src) != torch.device(dst)


def _event(stream):
    ev = torch.cuda.Event()
    ev.record(stream)
    return ev


class Lease:
    __slots__ = ("ready", "consumed", "home")

    def __init__(self, ready=None, home=None):
        self.ready = ready        # None: valid already
        self.consumed = []        # release events since `ready`
        self.home = home          # plain data only: the caller's stream that must wait for the consumer

    def __deepcopy__(self, memo):   # a copy of leased data is ordinary data on the copier's stream
        return None


class Block(tuple):
    """tuple of tensors + lease."""

    def __new__(cls, tensors, lease=None):
        self = super().__new__(cls, tensors)
        self._pql_lease = lease
        return self


def lease_of(obj):
    return getattr(obj, "_pql_lease", None)


def tag(tensor, lease):
    """Attach a lease to a single tensor (the obs block handed to the P-learner)."""
    tensor._pql_lease = lease
    return tensor


def _device_of(obj):
    if torch.is_tensor(obj):
        return obj.device
    arena = getattr(obj, "arena", None)
    if arena is not None:
        return arena.device
    for t in obj:
        if torch.is_tensor(t):
            return t.device
    raise TypeError(f"cannot tell the device of {type(obj)}")


def acquire(obj, stream, home=None):
    """Make `stream` wait until `obj` is valid.  Returns the lease to pass to release() once the reads are enqueued.
    Call with LOCK held when other threads may publish the same object.  `home`: for data without a lease, the
    caller's stream on the data's device when that is no longer torch's current stream (the learners switch to
    their own stream before they look at their arguments)."""
    lease = lease_of(obj)
    if lease is None:
        dev = _device_of(obj)
        if dev.type != "cuda":
            return None
        if home is None or home.device != dev:
            home = torch.cuda.current_stream(dev)
        if home == stream:
            return None
        lease = Lease(_event(home), home=home)
    if lease.ready is not None:
        stream.wait_event(lease.ready)
    return lease


def release(lease, stream):
    if lease is None:
        return
    ev = _event(stream)
    if lease.home is not None:      # plain data: the caller's stream may reuse the memory right after we return
        lease.home.wait_event(ev)
    else:
        lease.consumed.append(ev)


def reclaim(lease, stream):
    """Producer side: `stream` waits for every reader of the previous contents, then the lease starts over."""
    for ev in lease.consumed:
        stream.wait_event(ev)
    lease.consumed = []
    lease.ready = None


class ArenaPublisher:
    """Double-buffered snapshots of `module`'s flat parameter arena."""

    def __init__(self, module, slots=2):
        from copy import deepcopy
        self.live = module
        self.slots = [deepcopy(module) for _ in range(slots)]
        for s in self.slots:
            s._pql_lease = Lease()
            s.requires_grad_(False)
        self.k = 0

    def publish(self):
        """Snapshot on the CURRENT stream of the module's device (the owner's compute stream: ordered after the
        optimiser steps enqueued so far); returns the snapshot module."""
        with LOCK:
            s = self.slots[self.k]
            self.k = (self.k + 1) % len(self.slots)
            st = torch.cuda.current_stream(self.live.arena.device)
            reclaim(s._pql_lease, st)
            s.arena.data.copy_(self.live.arena.data, non_blocking=True)
            s._pql_lease.ready = _event(st)
            return s


_COPY_STREAMS = {}


def copy_stream(device):
    """The one dedicated copy stream of `device` (created on first use)."""
    device = torch.device(device)
    with LOCK:
        st = _COPY_STREAMS.get(device)
        if st is None:
            st = _COPY_STREAMS[device] = torch.cuda.Stream(device)
        return st


_SHIPPERS = {}


def shipper(src, dst, pipe="data", slots=3):
    """The process-wide Shipper of (src GPU, dst GPU, pipe name): separate pipes keep parameter arenas and transition
    blocks (different sizes, different consumers) out of each other's landing slots."""
    key = (torch.device(src), torch.device(dst), pipe)
    with LOCK:
        sh = _SHIPPERS.get(key)
        if sh is None:
            sh = _SHIPPERS[key] = Shipper(src, dst, slots)
        return sh


class Shipper:
    """src GPU -> dst GPU pipe with `slots` landing blocks on dst."""

    def __init__(self, src, dst, slots=2):
        self.src, self.dst = torch.device(src), torch.device(dst)
        self.cs_src, self.cs_dst = copy_stream(self.src), copy_stream(self.dst)
        self.land = [None] * slots          # flat fp32 landing buffers on dst
        self.lease = [Lease() for _ in range(slots)]
        self.keep = []                      # outgrown landing buffers stay allocated (their readers hold no stream record)
        self.k = 0

    def ship(self, tensors, src_lease=None):
        """Copy fp32 `tensors` (on src) into the next landing block; returns Block(tensors on dst) whose lease is
        ready when the bytes have landed.  The source is released (`src_lease`, or the caller's stream) when the
        copy engine has read it."""
        with LOCK:
            k = self.k
            self.k = (k + 1) % len(self.land)
            total = sum(t.numel() for t in tensors)
            lease = self.lease[k]
            reclaim(lease, self.cs_dst)                       # every reader of this landing block is done
            if src_lease is None:
                src_lease = Lease(_event(torch.cuda.current_stream(self.src)), home=torch.cuda.current_stream(self.src))
            if src_lease.ready is not None:
                self.cs_src.wait_event(src_lease.ready)
            with torch.cuda.stream(self.cs_dst):
                if self.land[k] is None or self.land[k].numel() < total:
                    if self.land[k] is not None:
                        self.keep.append(self.land[k])
                    self.land[k] = torch.empty(total, dtype=torch.float32, device=self.dst)
                outs, off = [], 0
                # torch's cross-device copy_ runs on the SOURCE device's current stream and fences the destination
                # device's current stream around it: with both set to the copy streams nothing touches a compute stream
                with torch.cuda.stream(self.cs_src):
                    for t in tensors:
                        n = t.numel()
                        dst = self.land[k][off: off + n].view(t.shape)
                        dst.copy_(t, non_blocking=True)
                        outs.append(dst)
                        off += n
                    release(src_lease, self.cs_src)
                lease.ready = _event(self.cs_dst)
            return Block(outs, lease)
