"""Robustness sweep of scripts/train_pql.py over cfg corners (one process, a few hundred thousand env steps each):
    python tools/config_sweep.py
Every run must finish, keep its update ratios, and end with finite parameters and losses."""
import math
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))

import torch  # noqa: E402

import train_pql  # noqa: E402
from pql_amd.utils.cfg import load_cfg  # noqa: E402

BASE = ["algo.num_gpus=1", "num_envs=512", "algo.batch_size=2048", "algo.memory_size=200000", "algo.warm_up=8", "max_step=300000",
        "algo.eval_freq=50"]
CASES = {
    "default (graph, streams, fused)": ["task=AllegroHand"],
    "distributional": ["task=AllegroHand", "algo.distl=True"],
    "nstep 1": ["task=AllegroHand", "algo.nstep=1"],
    "nstep 5, Humanoid": ["task=Humanoid", "algo.nstep=5"],
    "no obs norm": ["task=AllegroHand", "algo.obs_norm=False"],
    "eager (no graph)": ["task=AllegroHand", "algo.graph=False"],
    "one stream": ["task=AllegroHand", "algo.streams=False"],
    "per-layer GEMMs (no fused forward)": ["task=AllegroHand", "algo.fused=False"],
    "separate tail launches": ["task=AllegroHand", "algo.fused_tail=False"],
    "hidden [256,256]": ["task=AllegroHand", "algo.hidden_layers=[256,256]"],
    "hidden [512,512,256]": ["task=AllegroHand", "algo.hidden_layers=[512,512,256]"],
    "hidden [96,64] (no 128-multiples: dense DPG chain)": ["task=AllegroHand", "algo.hidden_layers=[96,64]"],
    "hidden [1024,512] (wide fused tiles)": ["task=AllegroHand", "algo.hidden_layers=[1024,512]"],
    "Ant shape": ["task=Ant"],
    "ShadowHand PQL-D": ["task=ShadowHand", "algo.distl=True"],
    "free-running learners": ["task=AllegroHand", "algo.async_learners=True"],
    "free-running, distributional, eager": ["task=AllegroHand", "algo.async_learners=True", "algo.distl=True", "algo.graph=False"],
    "batch not a multiple of 128": ["task=AllegroHand", "algo.batch_size=1000"],
    "tiny (64 envs, batch 256)": ["task=AllegroHand", "num_envs=64", "algo.batch_size=256", "max_step=40000"],
    "per-step ATen draws (rng=torch)": ["task=AllegroHand", "algo.rng=torch"],
    "draws 3 steps ahead (not a divisor of 8)": ["task=AllegroHand", "algo.prefetch_steps=3", "algo.prefetch_steps_p=5"],
    "free-running, draws ahead, TD loss as its own launch": ["task=AllegroHand", "algo.async_learners=True", "algo.td_in_head=False"],
    "SAC-style ratios 1:1:4": ["task=AllegroHand", "algo.critic_sample_ratio=4", "algo.critic_actor_ratio=1"],
    "round-3 launch sequences (no actor-ahead, separate DPG launches)": ["task=AllegroHand", "algo.actor_ahead=False", "algo.dpg_fused=False"],
    "BASELINE hidden, free-running, 2 steps ahead": ["task=AllegroHand", "algo.hidden_layers=[512,512,256]", "algo.async_learners=True"],
}


def main():
    bad = 0
    for name, extra in CASES.items():
        t0 = time.time()
        try:
            cfg = load_cfg(extra + [a for a in BASE if a.split("=")[0] not in {e.split("=")[0] for e in extra}])
            out = train_pql.main(cfg)
            torch.cuda.synchronize()
            ok = (out["critic_updates"] > 0 and out["actor_updates"] > 0 and math.isfinite(out["critic_loss"])
                  and math.isfinite(out["actor_loss"]))
            ratio = out["critic_updates"] / max(out["actor_updates"], 1)
            status = "ok" if ok and math.isfinite(ratio) else "SUSPECT"
            print(f"[{status}] {name}: {out['critic_updates']} critic / {out['actor_updates']} actor updates (ratio {ratio:.2f}), losses {out['critic_loss']:.4g} / {out['actor_loss']:.4g}, "
                  f"{out['rollout_iterations']} rollout iterations, {time.time() - t0:.1f} s", flush=True)
            bad += status != "ok"
        except Exception:   # noqa: BLE001
            bad += 1
            print(f"[FAIL] {name}", flush=True)
            traceback.print_exc()
    print(f"{len(CASES) - bad} / {len(CASES)} configurations ran clean")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
