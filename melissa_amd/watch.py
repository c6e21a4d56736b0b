"""Evaluation run, the counterpart of the reference's ``--watch`` mode (l_dgn.py:315-319 -> ``watch``: load a policy,
run test episodes on evaluation envs, report the episodes' statistics).

    python -m melissa_amd.watch --model l_dgn --nodes 20 --envs 1 --episodes 10 [--load policy.pth]

Graphs: connected random geometric graphs (the reference reads graph_topologies/testing_N/*; pass your own pool through
``watch(graph_pool=...)``).  ``--load`` takes a state_dict saved by the reference's trainer (keys ``model.*`` /
``model_old.*``, l_dgn.py:311), loaded with ``weights_only=True``.
"""
from __future__ import annotations

import argparse
import json

import torch

from .collect import Collector
from .env import HipGraphVectorEnv, synthetic_graph_pool
from .policy import DQNPolicy
from .train import build_network


def watch(model="l_dgn", n_nodes=20, envs=1, episodes=10, load=None, graph_pool=None, seed=9, device="cuda:0",
          dynamic_graph=True, feature_dtype="f32"):
    torch.manual_seed(seed)
    net = build_network(model, n_nodes, device)
    policy = DQNPolicy(net, target_update_freq=1)
    if load:
        policy.load_state_dict(torch.load(load, map_location=device, weights_only=True))
    net.eval()
    net.set_feature_dtype(feature_dtype)
    pool = graph_pool if graph_pool is not None else synthetic_graph_pool(n_nodes, 16, first_seed=0)
    venv = HipGraphVectorEnv(envs, n_nodes, graph_pool=pool, dynamic_graph=dynamic_graph, device=device, max_moves=64,
                             seed=seed, construct_like_reference=False, is_testing=True, num_test_episodes=episodes)
    per_env = -(-episodes // envs) + 2
    col = Collector(policy, venv, episodes_per_env=per_env, seed=seed, eps=0.0, chunk=4, use_graph=envs >= 64)
    out = col.collect(n_episode=episodes)
    # the scalars of the collect result (counts, speed, mean return / length, mean of every logger_stats key) as a plain dict
    keys = ["n/ep", "n/st", "collect_time", "collect_speed"] + (["rew", "len"] if out.returns_stat is not None else []) \
        + list(out.info.stats)
    return {k: out[k] for k in keys}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="l_dgn", choices=["l_dgn", "hl_dgn", "dgn_r"])
    ap.add_argument("--nodes", type=int, default=20)
    ap.add_argument("--envs", type=int, default=1)
    ap.add_argument("--episodes", type=int, default=10)
    ap.add_argument("--load", default=None)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f32s", "f32a"])
    a = ap.parse_args()
    print(json.dumps(watch(a.model, a.nodes, a.envs, a.episodes, a.load, feature_dtype=a.dtype)))


if __name__ == "__main__":
    main()
