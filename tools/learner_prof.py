"""Where one DQN update of the learner leg spends its time (HL-DGN 50-node, 512 envs, batch 32): python tools/learner_prof.py [model]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from melissa_amd.collect import RoundLoop
from melissa_amd.env import HipGraphVectorEnv, synthetic_graph_pool
from melissa_amd.networks import DGNRNetwork, HLDGNNetwork, LDGNNetwork
from melissa_amd.policy import DGNPolicy, DQNPolicy
from melissa_amd.replay import DGNLearner, DQNLearner, RoundReplay

model = sys.argv[1] if len(sys.argv) > 1 else "hl_dgn"
n, envs = 50, 512
duel = lambda: ({"hidden_sizes": [128, 128]}, {"hidden_sizes": [128, 128]})
torch.manual_seed(9)
if model == "hl_dgn":
    net = HLDGNNetwork(5, 128, 2, 4, n, aggregator="max", dueling_param=duel(), device="cuda")
elif model == "l_dgn":
    net = LDGNNetwork(5, 128, 2, 4, n, dueling_param=duel(), device="cuda")
else:
    net = DGNRNetwork(5, 128, 2, 4, n, dueling_param=duel(), device="cuda")
P = DGNPolicy if model == "dgn_r" else DQNPolicy
policy = P(net, torch.optim.Adam(net.parameters(), lr=1e-3), estimation_step=4, target_update_freq=500)
venv = HipGraphVectorEnv(envs, n, graph_pool=synthetic_graph_pool(n, 64, 0), dynamic_graph=True, device="cuda", max_moves=48,
                         seed=5000, construct_like_reference=False)
replay = RoundReplay(envs, n, 32, "cuda")
loop = RoundLoop(venv, policy, seed=5000, eps=0.1, replay=replay)
L = (DGNLearner if model == "dgn_r" else DQNLearner)(policy, replay, batch_size=32, n_step=4, gamma=0.99, seed=0)
with torch.no_grad():
    loop.run(40)
for _ in range(3):
    L.step()
torch.cuda.synchronize()
def timed(f, reps=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
sample = lambda: replay.sample(32, 4, 0.99, L.gen)
print(f"{model}: whole update {timed(L.step):.2f} ms")
print(f"  replay sample      {timed(sample):.2f} ms")
b = sample()
with torch.no_grad():
    print(f"  target forward     {timed(lambda: policy.model_old.hip_forward(b['boot_obs'])):.2f} ms")
if model == "dgn_r":            # the dense form: one graph per sampled experience, the head per (experience, node)
    obs = L.sample_batch()["obs_matrix"]
    body = net.torch_forward_all_agents
    rows = L.row_form(L.sample_batch())["segment"].numel()
    print(f"  (row form of this batch: {rows} sibling rows - what the per-sibling forward evaluated before round 3)")
else:
    obs, body = b["obs"], net.torch_forward
def fwd():
    with torch.enable_grad():
        return body(obs)
print(f"  learn forward      {timed(fwd):.2f} ms  ({obs.shape[0]} graphs)")
def fb():
    policy.optim.zero_grad()
    with torch.enable_grad():
        body(obs).pow(2).mean().backward()
print(f"  forward + backward {timed(fb):.2f} ms")
print(f"  optimizer step     {timed(policy.optim.step):.2f} ms")
with torch.no_grad():
    print(f"  4 collect rounds   {timed(lambda: loop.run(4)):.2f} ms")
L.capture()                                     # the same update replayed from HIP graphs (DQNLearner / DGNLearner.capture)
print(f"  whole update, replayed from HIP graphs {timed(L.step):.2f} ms")
if "--kernels" in sys.argv:                     # per-kernel device time of 10 whole updates (torch profiler, kineto/roctracer)
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(10):
            L.step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=70))
