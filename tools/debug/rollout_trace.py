"""Kernel list of one steady-state rollout iteration (explore_env + both hand-offs), under rocprofv3 --kernel-trace:
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/rt -o t -- python3 tools/debug/rollout_trace.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
sys.argv = [sys.argv[0], "--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda:0"); torch.cuda.set_device(dev); torch.manual_seed(42)
cfg, env, actor, v, p = bench.build_system(args, 0, 1, dev, None)
critic, pol = bench.prefill(actor, v, p, env, cfg, args, dev)
for i in range(40):
    actor.set_actor(pol)
    p_data, v_data, _ = actor.explore_env(env, cfg.algo.horizon_len, random=False)
    critic, _, _ = v.update(pol, v_data, actor.obs_rms.get_states(v.device), 0)
    pol, _, _ = p.update(critic, p_data, actor.obs_rms.get_states(p.device), 0)
torch.cuda.synchronize()
