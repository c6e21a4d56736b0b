"""Eager against HIP-graph-replayed DQN updates (DQNLearner.capture) at the learner leg's size: python tools/captured_update_check.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from melissa_amd.collect import RoundLoop
from melissa_amd.env import HipGraphVectorEnv, synthetic_graph_pool
from melissa_amd.networks import HLDGNNetwork, LDGNNetwork
from melissa_amd.policy import DQNPolicy
from melissa_amd.replay import DQNLearner, RoundReplay
n, envs = 50, 512
duel = lambda: ({"hidden_sizes": [128, 128]}, {"hidden_sizes": [128, 128]})
def make(model, capture):
    torch.manual_seed(9)
    cls = HLDGNNetwork if model == "hl_dgn" else LDGNNetwork
    kw = dict(aggregator="max") if model == "hl_dgn" else {}
    net = cls(5, 128, 2, 4, n, dueling_param=duel(), device="cuda", **kw)
    policy = DQNPolicy(net, torch.optim.Adam(net.parameters(), lr=1e-3), estimation_step=4, target_update_freq=5)
    venv = HipGraphVectorEnv(envs, n, graph_pool=synthetic_graph_pool(n, 64, 0), dynamic_graph=True, device="cuda", max_moves=48,
                             seed=5000, construct_like_reference=False)
    replay = RoundReplay(envs, n, 32, "cuda")
    loop = RoundLoop(venv, policy, seed=5000, eps=0.1, replay=replay)
    L = DQNLearner(policy, replay, batch_size=32, n_step=4, gamma=0.99, seed=0)
    with torch.no_grad():
        loop.run(40)
    return net, policy, loop, L
for model in ("hl_dgn", "l_dgn"):
    net, policy, loop, L = make(model, False)
    for _ in range(3): L.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): L.step()
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 30 * 1e3
    L.capture()
    for _ in range(3): L.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): out = L.step()
    torch.cuda.synchronize(); cap = (time.perf_counter() - t0) / 30 * 1e3
    print(f"{model}: eager update {eager:.2f} ms, captured {cap:.2f} ms, loss {float(out['loss']):.4f}")
    with torch.no_grad():
        loop.run(8)
    torch.cuda.synchronize()
    print("collect after captured updates ok, errors", loop.counters()["errors"])
