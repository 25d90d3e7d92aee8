"""Speed-ratio controller of the three free-running PQL components.

The reference keeps rollout : V-learner : P-learner at 1 : `critic_sample_ratio` : `critic_sample_ratio /
critic_actor_ratio` by measuring each component's time per unit of work over a sliding window of the last 100
rollout iterations and telling the too-fast side how long to sleep per unit (scripts/train_pql.py:72-88 state,
:127-158 control law, :159-166 bookkeeping; the learners sleep what they are told, pql_v_learner.py:140-141).
This class is that law as an object: `observe()` once per rollout iteration with the three running counters,
then sleep `sim_wait_time` and pass `critic_wait_time` / `actor_wait_time` to the learners' `update()`.

Semantics kept from the reference, because they decide the equilibrium:
  * every correction is applied to the wait time the OLDEST sample of the window had (not the current one);
  * rollout vs V-learner is a two-sided rule with exactly one non-zero wait at a time: slack is first taken out
    of the other side's wait, only then added to one's own;
  * the P-learner rule is one-sided (only the P-learner ever waits for the V-learner);
  * nothing moves until the window holds 10 samples and both learners have stepped at least once.
Difference: a window in which a counter did not advance is skipped (the reference would divide by zero).
"""
from __future__ import annotations

import time
from collections import deque
from dataclasses import dataclass


@dataclass
class _Sample:
    time: float
    sim: int
    critic: int
    actor: int
    sim_wait: float
    critic_wait: float
    actor_wait: float


class RatioController:
    WINDOW = 100      # train_pql.py:78
    MIN_SAMPLES = 10  # train_pql.py:127

    def __init__(self, critic_sample_ratio, critic_actor_ratio, critic_updates=0, actor_updates=0, clock=time.time):
        self.r_cs = float(critic_sample_ratio)
        self.r_ca = float(critic_actor_ratio)
        self.clock = clock
        self.sim_count = 0
        self.sim_wait_time = 0.0
        self.critic_wait_time = 0.0
        self.actor_wait_time = 0.0
        self.window = deque(maxlen=self.WINDOW)
        self._push(critic_updates, actor_updates)

    def _push(self, critic_updates, actor_updates):
        self.window.append(_Sample(self.clock(), self.sim_count, critic_updates, actor_updates, self.sim_wait_time,
                                   self.critic_wait_time, self.actor_wait_time))

    def observe(self, critic_updates, actor_updates):
        """One rollout iteration has finished; counters are the learners' `update_count`s as last reported."""
        self.sim_count += 1
        old = self.window[0]
        d_sim, d_cri, d_act = self.sim_count - old.sim, critic_updates - old.critic, actor_updates - old.actor
        if (len(self.window) >= self.MIN_SAMPLES and critic_updates != 0 and actor_updates != 0
                and d_sim > 0 and d_cri > 0 and d_act > 0):
            span = self.clock() - old.time
            sim_unit, critic_unit, actor_unit = span / d_sim, span / d_cri, span / d_act
            # rollout vs V-learner: `slack` > 0 means a V step is quicker than its share of a rollout iteration
            slack = sim_unit / self.r_cs - critic_unit
            if slack > 0:
                if self.sim_wait_time == 0:
                    self.critic_wait_time = old.critic_wait + slack
                else:
                    self.sim_wait_time = max(0, old.sim_wait - slack)
            elif self.critic_wait_time == 0:
                self.sim_wait_time = old.sim_wait - slack
            else:
                self.critic_wait_time = max(0, old.critic_wait + slack)
            # V-learner vs P-learner: only the P-learner waits
            slack = critic_unit * self.r_ca - actor_unit
            self.actor_wait_time = old.actor_wait + slack if slack > 0 else max(0, old.actor_wait + slack)
        self._push(critic_updates, actor_updates)
        return self.sim_wait_time, self.critic_wait_time, self.actor_wait_time
