#!/usr/bin/env python3
"""Why `test_gradient_buckets_leave_the_bits_of_the_single_collective` aborted once in ~15 runs, reproduced on purpose.

One RCCL rank.  An EAGER all-reduce is issued, then a hipGraph capture records another all-reduce and stays open for 0.4 s (four
sweeps of ProcessGroupNCCL's watchdog thread).  The capture puts the communicator's internal stream into capture mode; the watchdog
is still polling the eager work's end event, which was recorded on that stream, and HIP answers `operation not permitted on an event
last recorded in a capturing stream` -> the watchdog throws -> SIGABRT.  With pql_amd.utils.dp.drain_pending_collectives() between
the two (what the learners do before every capture that records a collective) the watchdog's list is empty and the capture is safe.
    python tools/probes/rccl_capture_probe.py          # runs both variants as child processes and prints their exit status"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def child(drain):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from pql_amd.utils.dp import drain_pending_collectives
    pg = dist.group.WORLD
    x = torch.ones(1 << 20, device="cuda")
    side = torch.cuda.Stream()
    for attempt in range(3):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            dist.all_reduce(x, group=pg)            # eager: its work sits in the watchdog's list until the next sweep
        torch.cuda.current_stream().wait_stream(side)
        if drain:
            drain_pending_collectives(pg)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            dist.all_reduce(x, group=pg)            # captured: the communicator's stream joins the capture
            time.sleep(0.4)                         # ... and stays in it across several watchdog sweeps
        g.replay()
        torch.cuda.synchronize()
    print(f"drain={drain}: three captures survived", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1] == "drain")
    else:
        for mode in ("nodrain", "drain"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), mode], capture_output=True, text=True, timeout=180)
            tail = [l for l in (r.stdout + r.stderr).splitlines() if "captur" in l.lower() or "survived" in l][:3]
            print(f"[{mode}] exit code {r.returncode}" + "".join("\n    " + l[:200] for l in tail), flush=True)
