import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import argparse, importlib.util, torch
spec = importlib.util.spec_from_file_location("bench_mod", "bench.py"); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
args = argparse.Namespace(task="AllegroHand", num_envs=4096, batch=8192, replay=200000, nstep=3, hidden="512,512,256", distl=False,
                          no_graph=True, no_streams=True, no_fused=False)
dev = torch.device("cuda:0")
cfg, env, actor, v, p = bench.build_system(args, 0, 1, dev, None)
critic, policy = bench.prefill(actor, v, p, env, cfg, args, dev)
for _ in range(3):
    p.learn()
torch.cuda.synchronize()
ws = p._ws
from pql_amd import _lib as L
import ctypes as C
cl = p.critic.layout
rows_cap = 2 * ((8192 + 127) // 128 * 128)
mld = 512
perm_off = 2 * rows_cap * mld
mn = ws["bwd_c"][perm_off + rows_cap: perm_off + rows_cap + 4].view(torch.int32)
print("mn =", mn.tolist(), "owner counts", torch.bincount(ws["owner"].long(), minlength=4).tolist())
