"""Stdout table + JSONL metrics; replaces the reference's wandb / loguru logging with the same metric names
(scripts/train_pql.py:160-182)."""
import json
import time


class MetricLogger:
    def __init__(self, jsonl_path=None):
        self.t0 = time.time()
        self.fh = open(jsonl_path, "a") if jsonl_path else None
        self.header_done = False

    def log(self, info, step):
        if self.fh:
            self.fh.write(json.dumps({"step": int(step), "time": time.time() - self.t0, **{k: float(v) for k, v in info.items()}}) + "\n")
            self.fh.flush()

    def table(self, step, info):
        if not self.header_done:
            print(f"{'Steps':>12s}{'Time':>12s}{'critic_loss':>12s}{'actor_loss':>12s}{'v-updates':>12s}{'p-updates':>12s}")
            self.header_done = True
        print(f"{step:12.2e}{time.time() - self.t0:>12.1f}{info['train/critic_loss']:12.4f}{info['train/actor_loss']:12.4f}"
              f"{info['train/critic_update_times']:12.0f}{info['train/actor_update_times']:12.0f}", flush=True)
