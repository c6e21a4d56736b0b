"""Launches per DQN update (HL-DGN 50-node, batch 32): run under `rocprofv3 --kernel-trace --stats` with MODE=eager|captured and
divide the call counts by the 50 updates.   python tools/update_launch_count.py eager|captured"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from melissa_amd.collect import RoundLoop
from melissa_amd.env import HipGraphVectorEnv, synthetic_graph_pool
from melissa_amd.networks import HLDGNNetwork
from melissa_amd.policy import DQNPolicy
from melissa_amd.replay import DQNLearner, RoundReplay
mode = sys.argv[1] if len(sys.argv) > 1 else "eager"
n, envs = 50, 512
duel = lambda: ({"hidden_sizes": [128, 128]}, {"hidden_sizes": [128, 128]})
torch.manual_seed(9)
net = HLDGNNetwork(5, 128, 2, 4, n, aggregator="max", dueling_param=duel(), device="cuda")
policy = DQNPolicy(net, torch.optim.Adam(net.parameters(), lr=1e-3), estimation_step=4, target_update_freq=500)
venv = HipGraphVectorEnv(envs, n, graph_pool=synthetic_graph_pool(n, 64, 0), dynamic_graph=True, device="cuda", max_moves=48,
                         seed=5000, construct_like_reference=False)
replay = RoundReplay(envs, n, 32, "cuda")
loop = RoundLoop(venv, policy, seed=5000, eps=0.1, replay=replay)
L = DQNLearner(policy, replay, batch_size=32, n_step=4, gamma=0.99, seed=0)
with torch.no_grad():
    loop.run(40)
L.step()
if mode == "captured":
    L.capture()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(50):
    L.step()
torch.cuda.synchronize()
print(f"{mode}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per update (50 updates)")
