import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_gather as bg
import torch
from pql_amd import _lib as L
for cap in (50_000, 250_000, 1_000_000, 5_000_000, 20_000_000):
    bg.CFG["x"] = (108, 21, 32768, cap)
    # monkeypatch sweep to a single config
    orig_run = bg.run
    O, A, B, _ = bg.CFG["x"]
    print("ring rows", cap, "=", cap * 1024 / 1e9, "GB")
    import io, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bg.run("x", iters=20)
    for line in buf.getvalue().splitlines():
        if "R=2 waves/CU=24 nopad=1 nt=0" in line or "auto" in line:
            print("   ", line.strip())
    torch.cuda.empty_cache()
