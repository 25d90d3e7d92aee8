"""Soak run of scripts/train_pql.py on one MI355X: `python tools/soak.py <seconds> [overrides...]`.
Samples device memory (allocated / reserved), host RSS and the update counters every few seconds while the entry point runs,
and reports whether anything grows or goes non-finite.  Example:
    python tools/soak.py 60 task=AllegroHand algo.async_learners=True
"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))

import psutil  # noqa: E402
import torch  # noqa: E402

import train_pql  # noqa: E402
from pql_amd.utils.cfg import load_cfg  # noqa: E402


def main():
    seconds = float(sys.argv[1])
    cfg = load_cfg(sys.argv[2:] + [f"max_time={seconds}"])
    samples, stop = [], threading.Event()
    proc = psutil.Process()

    def monitor():
        t0 = time.time()
        while not stop.wait(5.0):
            samples.append((time.time() - t0, torch.cuda.memory_allocated() / 2 ** 20, torch.cuda.memory_reserved() / 2 ** 20,
                            proc.memory_info().rss / 2 ** 20, threading.active_count()))

    th = threading.Thread(target=monitor, daemon=True)
    th.start()
    out = train_pql.main(cfg)
    stop.set()
    th.join()
    for s in samples:
        print("t=%6.1fs  dev alloc %8.1f MiB  reserved %8.1f MiB  host rss %8.1f MiB  threads %d" % s)
    print(out)
    if len(samples) >= 4:
        a0, a1 = samples[1], samples[-1]
        print("growth after the first sample: dev alloc %+.1f MiB, reserved %+.1f MiB, host rss %+.1f MiB" %
              (a1[1] - a0[1], a1[2] - a0[2], a1[3] - a0[3]))


if __name__ == "__main__":
    main()
