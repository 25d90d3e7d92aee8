"""Exploration-noise std schedules (names of pql/utils/schedule_util.py)."""
import math


class LinearSchedule:
    def __init__(self, start_val, end_val, total_iters=5):
        self.start_val, self.end_val, self.total_iters = start_val, end_val, total_iters
        self.count = 0
        self.last_val = start_val

    def step(self):
        if self.count <= self.total_iters:
            frac = self.count / self.total_iters
            self.last_val = self.start_val + frac * (self.end_val - self.start_val)
            self.count += 1
        return self.last_val

    def val(self):
        return self.last_val


class ExponentialSchedule:
    def __init__(self, start_val, gamma, end_val=None):
        self.start_val, self.gamma, self.end_val = start_val, gamma, end_val
        self.total_iters = None if end_val is None else int((math.log(end_val) - math.log(start_val)) / math.log(gamma))
        self.count = 0
        self.last_val = start_val

    def step(self):
        if self.total_iters is None or self.count <= self.total_iters:
            self.last_val = self.last_val * self.gamma
            self.count += 1
        return self.last_val

    def val(self):
        return self.last_val
