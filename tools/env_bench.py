"""Timing of the env launches in isolation (torch events on the launch stream)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from melissa_amd.collect import RoundLoop, DecisionLoop
from melissa_amd.env import HipGraphVectorEnv, synthetic_graph_pool
from melissa_amd.networks import LDGNNetwork
from melissa_amd.policy import DQNPolicy

def timed(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

n, B = 50, 1024
graphs = synthetic_graph_pool(n, 16, 0)
net = LDGNNetwork(5, 128, 2, 4, n, dueling_param=({"hidden_sizes": [128, 128]}, {"hidden_sizes": [128, 128]}), device="cuda", backend="hip")
venv = HipGraphVectorEnv(B, n, graph_pool=graphs, dynamic_graph=True, device="cuda", max_moves=48, construct_like_reference=False)
loop = RoundLoop(venv, DQNPolicy(net), episodes_per_env=12, seed=1, eps=0.0)
loop.run(20)
print("load+store only (first=1):", round(timed(lambda: venv.round_device(loop.pool, None, None, loop.live, None, first=True)), 2), "us")
loop.act.zero_()
print("round, all-zero actions  :", round(timed(lambda: venv.round_device(loop.pool, loop.act, loop.offsets, loop.live, loop.table)), 2), "us  (offsets stale: error flags expected)")
venv2 = HipGraphVectorEnv(B, n, graph_pool=graphs, dynamic_graph=True, device="cuda", max_moves=48, construct_like_reference=False)
aec = DecisionLoop(venv2, DQNPolicy(net), episodes_per_env=12, seed=1, eps=0.0)
aec.run(20)
print("aec step+observe         :", round(timed(lambda: venv2.step_device(aec.pool, aec.act, aec.out, aec.table)), 2), "us")
