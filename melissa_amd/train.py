"""Data-parallel DQN training over env shards: the distributed counterpart of ``train_agent``
(l_dgn.py:131-269 / hl_dgn.py / dgn_r.py: collect ``step_per_collect`` transitions, then
``update_per_step`` gradient steps).

One process per GPU (``python -m torch.distributed.run --nproc-per-node N -m melissa_amd.train ...``):
every rank owns ``envs`` independent envs (no data-path collective: rank r steps and evaluates its own
shard with the HIP kernels, device-resident replay), replicas start from rank 0's weights, and the ONE
collective of the path is the flat-gradient all-reduce (RCCL over xGMI) between backward and the Adam step,
so all replicas stay bit-identical.  Target-network sync, eps schedule and counters are replicated locally.
"""
from __future__ import annotations

import argparse
import json
import time

# torch and the package's GPU-facing modules are imported inside the functions that need them: ``python -m melissa_amd.train
# --gpus N`` runs this module top to bottom in the LAUNCHER PARENT, which must stay GPU-free (melissa_amd/launch.py) - it
# parses its arguments, starts the ranks and never gets as far as ``train()``.


def build_network(name: str, n_nodes: int, device, hidden=128, heads=4):
    from .networks import DGNRNetwork, HLDGNNetwork, LDGNNetwork
    duel = ({"hidden_sizes": [128, 128]}, {"hidden_sizes": [128, 128]})          # common.py:41-42
    if name == "l_dgn":
        return LDGNNetwork(5, hidden, 2, heads, n_nodes, dueling_param=duel, device=device)
    if name == "hl_dgn":
        return HLDGNNetwork(5, hidden, 2, heads, n_nodes, aggregator="max", dueling_param=duel, device=device)
    if name == "dgn_r":
        return DGNRNetwork(5, hidden, 2, heads, n_nodes, dueling_param=duel, device=device)
    raise ValueError(name)


def train(model="hl_dgn", n_nodes=20, envs=256, updates=20, rounds_per_update=4, batch_size=32, n_step=4,
          gamma=0.99, lr=1e-3, target_update_freq=500, eps=0.1, replay_rounds=64, seed=9, backend=None, log=print,
          probe=None, graphs=16, ring=16, capture_updates=None):
    """``probe(update_index, net, learner, phase)`` (optional) is called with phase "before" / "after" around every
    update - tests use it to re-derive an update's loss from the sampled batch with the oracle.
    ``graphs``: size of the synthetic training-graph dataset (the reference trains on 50 000 graphs per size, README.md:92-93;
    pools >= 4096 go through the on-disk packed cache, ``melissa_amd.env.cached_graph_pool``).  Episodes come from the
    device episode stream: every reset draws a new (graph, source, interested set, movement seed) like World.reset.
    ``capture_updates``: replay the update from HIP graphs (``DQNLearner.capture``; DGN-R too: its dense sibling form has static
    shapes, replay.DGNLearner).  None = on unless a probe is attached; with several ranks the collective stays
    eager between two graphs.  The capture takes two extra (real, untimed) updates first: ``warmup_updates`` in the result."""
    import torch
    from . import launch, parallel
    from .collect import RoundLoop
    from .env import HipGraphVectorEnv, synthetic_graph_pool
    from .policy import DGNPolicy, DQNPolicy
    from .replay import DGNLearner, DQNLearner, RoundReplay
    if backend != "gloo":
        launch.check_rank_device()                             # exit 2 when LOCAL_RANK names a GPU this rank cannot see
    rank, local_rank, world = parallel.init_distributed(backend)
    device = torch.device("cuda", local_rank if backend != "gloo" else 0)
    torch.cuda.set_device(device)
    torch.manual_seed(seed)                                    # same init everywhere, then broadcast anyway
    net = build_network(model, n_nodes, device)
    parallel.broadcast_parameters(net, src=0)
    # dgn_r.py trains with DGNPolicy (summed sibling Q, policies/dgn.py); l_dgn.py / hl_dgn.py with DQNPolicy
    policy_cls, learner_cls = (DGNPolicy, DGNLearner) if model == "dgn_r" else (DQNPolicy, DQNLearner)
    policy = policy_cls(net, torch.optim.Adam(net.parameters(), lr=lr), discount_factor=gamma,
                        estimation_step=n_step, target_update_freq=target_update_freq)
    from .env import cached_graph_pool
    graph_list = cached_graph_pool(n_nodes, graphs, 0) if graphs >= 4096 else synthetic_graph_pool(n_nodes, graphs, first_seed=0)
    venv = HipGraphVectorEnv(envs, n_nodes, graph_pool=graph_list, dynamic_graph=True, device=device, max_moves=48,
                             seed=1000 + rank * envs, construct_like_reference=False)
    replay = RoundReplay(envs, n_nodes, replay_rounds, device)
    # the rounds between two updates replay from one HIP graph (bit-identical to the eager launches: tests/test_gpu_round.py)
    loop = RoundLoop(venv, policy, seed=1000 + rank * envs, eps=eps, replay=replay, ring=ring,
                     use_graph=device.type == "cuda" and probe is None, graph_rounds=max(1, rounds_per_update))
    learner = learner_cls(policy, replay, batch_size=batch_size, n_step=n_step, gamma=gamma,
                         grad_hook=parallel.FlatGradAllReducer(net), seed=seed + rank)
    with torch.no_grad():
        loop.run(max(n_step + 1, 8))                           # pre-fill (l_dgn.py:201)
    if capture_updates is None:
        # on by default, with any number of ranks: the collective stays EAGER between two graphs (pack | all-reduce | unpack +
        # step), so RCCL never enters a capture; rehearsed with two ranks on one GPU (tests/test_gpu_round.py)
        capture_updates = probe is None
    captured = bool(capture_updates) and device.type == "cuda"
    warmup_updates = 0
    if captured:
        learner.capture()                                      # (two warm-up updates, then the graphs)
        warmup_updates = 2
    t0 = time.perf_counter()
    losses = []
    for _ in range(updates):
        with torch.no_grad():
            loop.run(rounds_per_update)
        if probe is not None:
            probe(len(losses), net, learner, "before")
        losses.append(learner.step()["loss"])
        if probe is not None:
            probe(len(losses) - 1, net, learner, "after")
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    losses = [float(x) for x in losses]                        # (device tensors when the update is replayed from graphs)
    c = loop.counters()
    checksum = float(torch.cat([p.detach().flatten() for p in net.parameters()]).double().sum())
    out = dict(rank=rank, world=world, model=model, updates=updates, seconds=dt, loss_first=losses[0],
               loss_last=losses[-1], decisions=c["decisions"], episodes=c["episodes"], errors=c["errors"],
               param_checksum=checksum, updates_from_hip_graphs=captured,
               # the capture's warm-up updates are REAL optimizer steps taken before the timed loop (they advance the policy's
               # iteration counter and the replay sampler's generator): a run with capture on has taken `updates +
               # warmup_updates` steps, `seconds` covers `updates` of them
               warmup_updates=warmup_updates)
    # replicas must be identical after averaged-gradient steps
    same = parallel.all_reduce_max(checksum, device) == parallel.all_reduce_max(-checksum, device) * -1
    out["replicas_identical"] = bool(same)
    if rank == 0:
        log(json.dumps(out))
    parallel.barrier()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="hl_dgn", choices=["l_dgn", "hl_dgn", "dgn_r"])
    ap.add_argument("--nodes", type=int, default=20)
    ap.add_argument("--envs", type=int, default=256, help="envs per GPU")
    ap.add_argument("--updates", type=int, default=20)
    ap.add_argument("--rounds-per-update", type=int, default=4)
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--backend", default=None)
    ap.add_argument("--graphs", type=int, default=16, help="training-graph dataset size (50000 = the reference's)")
    ap.add_argument("--gpus", type=int, default=1, help="ranks to start (one per GPU) when not under torch.distributed.run")
    ap.add_argument("--capture-updates", choices=["auto", "on", "off"], default="auto",
                    help="replay the DQN update from HIP graphs (auto: on; the capture runs 2 extra untimed warm-up updates first, "
                         "reported as warmup_updates - compare on / off runs at equal total updates)")
    a = ap.parse_args()
    import os
    import sys
    from . import launch
    rc = launch.maybe_spawn("-m", ["melissa_amd.train", *sys.argv[1:]], a.gpus, check_devices=a.backend != "gloo")
    if rc is not None:
        raise SystemExit(rc)
    train(model=a.model, n_nodes=a.nodes, envs=a.envs, updates=a.updates, rounds_per_update=a.rounds_per_update,
          batch_size=a.batch_size, backend=a.backend, graphs=a.graphs,
          capture_updates={"auto": None, "on": True, "off": False}[a.capture_updates])


if __name__ == "__main__":
    main()
