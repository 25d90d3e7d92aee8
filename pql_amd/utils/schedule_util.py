"""Exploration-noise std schedules (`noise.decay: linear | exp`), named like pql/utils/schedule_util.py."""
import math


class _Schedule:
    """step() advances and returns the value; val() reads it.  Stepping stops after `total_iters` (+1) calls."""

    def __init__(self, start_val, total_iters):
        self.start_val = start_val
        self.total_iters = total_iters
        self.count = 0
        self.last_val = start_val

    def _advance(self):
        raise NotImplementedError

    def step(self):
        live = self.total_iters is None or self.count <= self.total_iters
        if live:
            self.last_val = self._advance()
            self.count += 1
        return self.last_val

    def val(self):
        return self.last_val


class LinearSchedule(_Schedule):
    """start_val -> end_val in `total_iters` equal steps."""

    def __init__(self, start_val, end_val, total_iters=5):
        super().__init__(start_val, total_iters)
        self.end_val = end_val

    def _advance(self):
        return self.start_val + (self.count / self.total_iters) * (self.end_val - self.start_val)


class ExponentialSchedule(_Schedule):
    """val *= gamma per step until it would pass `end_val` (no floor when end_val is None)."""

    def __init__(self, start_val, gamma, end_val=None):
        n = None if end_val is None else int((math.log(end_val) - math.log(start_val)) / math.log(gamma))
        super().__init__(start_val, n)
        self.gamma, self.end_val = gamma, end_val

    def _advance(self):
        return self.last_val * self.gamma
