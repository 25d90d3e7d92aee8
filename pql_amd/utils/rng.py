"""The learners' random draws without per-step ATen launches -- and still torch's own numbers.

Reference: every `PQLVLearner.learn()` draws `torch.randint(cur_capacity, (B,))` (simple_replay.py:87)
then one N(0,1) tensor of shape (B, A) (noise.py:20-21); every `PQLPLearner.learn()` one `randint` (pql_p_learner.py:49), each
on its process's device generator.  Round 2 issued those two ATen kernels in front of every step's hipGraph (~5 us each).

torch's device generators are counter based (Philox4x32-10): a draw is a pure function of (seed, offset, element index), and a
call advances the offset by a known amount.  `libpqlk`'s `pqlk_philox_draws` evaluates that function itself (csrc/philox.hip),
so ONE launch produces the draws of the next K steps ahead of time -- the same numbers, in the same stream order, that K
alternating torch calls would return -- and the generator object is moved along on the host (`set_offset`) as the steps are
consumed, so switching back to torch calls (injected draws, `algo.rng=torch`) continues the very same stream.

Nothing is taken on trust: `verified(device)` compares the kernel with torch.randint / normal_ on the device (two sizes,
two consecutive calls) the first time it is asked and the fused path is used only if every value is bit-equal; another torch or
rocRAND build that generates differently simply keeps the ATen launches (`algo.rng` reports which).
"""
from __future__ import annotations

import threading

import torch

from pql_amd import _lib as L

_VERIFIED = {}   # device index -> contract flag (0 / 1), or None when torch's kernels could not be reproduced
_VERIFY_LOCK = threading.Lock()   # the learners call verified() from their constructors; free-running ones live in threads


def philox_increment(numel: int) -> int:
    return int(L.lib.pqlk_philox_increment(int(numel)))


def launch_draws(seed, offset, rng_range, idx, normal, chunks, contract, device):
    """idx: (chunks, n_idx) int64 or None; normal: (chunks, n_normal...) float32 or None; on torch's current stream."""
    n_idx = idx[0].numel() if idx is not None else 0
    n_nrm = normal[0].numel() if normal is not None else 0
    with torch.cuda.device(device):
        L.check(L.lib.pqlk_philox_draws(int(seed), int(offset), 0, None, int(rng_range), L.ptr(idx), n_idx, L.ptr(normal), n_nrm,
                                        int(chunks), int(contract), L.stream(device)))


def verified(device):
    """Contract flag under which pqlk_philox_draws reproduces torch's randint / normal_ bit for bit on `device`, else None."""
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    with _VERIFY_LOCK:
        if key not in _VERIFIED:
            _VERIFIED[key] = _verify(device)
        return _VERIFIED[key]


def _verify(device):
    """Only a VALUE mismatch (another torch / rocRAND build that generates differently) or a torch without Generator.set_offset
    means "cannot be reproduced" -> None -> the learners keep the ATen launches.  A failing launch is an error and surfaces."""
    result = None
    try:
        for contract in (1, 0):
            ok = True
            for (n_idx, n_nrm, rng_range, seed, off0) in ((8192, 8192 * 16, 1_000_000, 1234567, 0), (32768, 32768 * 21, 4_999_999, 99, 4096),
                                                          (300, 300 * 2, 977, 7, 8)):
                g = torch.Generator(device=device)
                g.manual_seed(seed)
                g.set_offset(off0)
                want_i, want_n = [], []
                for _ in range(2):   # two consecutive "steps": checks the offset increments as well
                    want_i.append(torch.randint(rng_range, (n_idx,), generator=g, device=device))
                    want_n.append(torch.empty(n_nrm, device=device).normal_(generator=g))
                inc = philox_increment(n_idx) + philox_increment(n_nrm)
                if g.get_offset() != off0 + 2 * inc:
                    ok = False
                    break
                idx = torch.empty((2, n_idx), dtype=torch.int64, device=device)
                nrm = torch.empty((2, n_nrm), dtype=torch.float32, device=device)
                launch_draws(seed, off0, rng_range, idx, nrm, 2, contract, device)
                if not (torch.equal(idx, torch.stack(want_i)) and torch.equal(nrm, torch.stack(want_n))):
                    ok = False
                    break
            if ok:
                result = contract
                break
    except AttributeError:   # torch.Generator without get_offset / set_offset
        result = None
    return result


class DrawAhead:
    """The next `depth` steps' draws of ONE learner, produced by one launch and handed out step by step.

    refill(range[, steps])  on the caller's current stream: draws for steps now .. now + depth - 1 (or the first `steps` of them)
                    at the generator's current offset
    take()          slot of the next step; moves the generator past that step's draws (host only)
    invalidate()    drop what is left (the ring / its bound changed); the next refill starts at the generator's offset,
                    i.e. exactly where a sequence of torch calls would be
    """

    def __init__(self, gen, device, n_idx, normal_shape, depth, contract):
        self.gen, self.device, self.depth, self.contract = gen, torch.device(device), int(depth), int(contract)
        self.idx = torch.zeros((self.depth, int(n_idx)), dtype=torch.int64, device=self.device)
        self.normal = (torch.zeros((self.depth, *normal_shape), dtype=torch.float32, device=self.device)
                       if normal_shape is not None else None)
        n_nrm = self.normal[0].numel() if self.normal is not None else 0
        self.inc = philox_increment(n_idx) + philox_increment(n_nrm)
        self.pos, self.valid, self.base = 0, 0, 0

    def refill(self, rng_range, steps=None):
        """steps: draw only the first `steps` (<= depth) steps -- a caller that knows it will make fewer (the last, partial run of a
        timed block).  Same numbers per step either way: a step's draws depend on the generator offset alone."""
        n = self.depth if steps is None else max(1, min(self.depth, int(steps)))
        self.base = int(self.gen.get_offset())
        launch_draws(self.gen.initial_seed(), self.base, rng_range, self.idx, self.normal, n, self.contract, self.device)
        self.pos, self.valid = 0, n

    def take(self):
        slot = self.pos
        self.pos += 1
        self.valid -= 1
        self.gen.set_offset(self.base + self.pos * self.inc)
        return slot

    def invalidate(self):
        self.valid = 0
