"""Fidelity factor of the CPU baseline: time the ACTUAL reference env
loop next to the oracle's env restatement on the same graph, seed and action tape, in the build container
(needs /root/reference; never runs on the GPU box).

    python tools/fidelity_factor.py [--nodes 50] [--steps 1500]

Prints one JSON line: decisions/s of both and restatement_speed / reference_speed.  A box-side
cpu_baseline of kind "port" divided by this factor reads as reference-equivalent (env half only: the
network half of the reference is not importable here).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def time_reference(n, dynamic, tape):
    import ref_standins
    from make_env_golden import RefPettingZoo, connected_rggs
    ref_graph, _ = ref_standins.import_reference()
    ref_standins.DEFAULT_SEED = 21
    g = connected_rggs(n, 1, 7)[0][1]
    pz = RefPettingZoo(ref_graph.GraphEnv(graph=g, number_of_agents=n, radius=0.2, dynamic_graph=dynamic))
    obs, *_ = pz.reset()
    live = 0
    t0 = time.perf_counter()
    done = 0
    for a in tape:
        live += int(obs["mask"][0])
        obs, term, trunc, info = pz.step(int(a))
        if term or trunc:
            done += 1
            if done == n or info.get("explicit_reset", False):
                obs, *_ = pz.reset()
                done = 0
    return live / (time.perf_counter() - t0), g


def time_oracle(n, dynamic, tape, g):
    from oracle import env_oracle as eo
    adj = [0] * n
    for u, v in g.edges():
        adj[u] |= 1 << v
        adj[v] |= 1 << u
    pos = np.array([g.nodes[i]["pos"] for i in range(n)], dtype=np.float64)
    env = eo.OraclePettingZooEnv(eo.OracleGraphEnv(
        n, graph=eo.GraphSpec(pos, adj), dynamic_graph=dynamic,
        np_random=np.random.Generator(np.random.PCG64(np.random.SeedSequence(21)))))
    obs, _ = env.reset()
    live = 0
    t0 = time.perf_counter()
    done = 0
    for a in tape:
        live += int(obs["mask"][0])
        obs, _r, term, trunc, info = env.step(int(a))
        if term or trunc:
            done += 1
            if done == n or info.get("explicit_reset", False):
                obs, _ = env.reset()
                done = 0
    return live / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=50)
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--static", action="store_true")
    args = ap.parse_args()
    tape = np.random.RandomState(0).randint(0, 2, size=args.steps)
    ref, g = time_reference(args.nodes, not args.static, tape)
    ora = time_oracle(args.nodes, not args.static, tape, g)
    print(json.dumps({"n_nodes": args.nodes, "dynamic_graph": not args.static, "agent_steps": args.steps,
                      "reference_env_decisions_per_s": round(ref, 1), "oracle_env_decisions_per_s": round(ora, 1),
                      "restatement_over_reference": round(ora / ref, 2), "cpus": os.cpu_count()}))


if __name__ == "__main__":
    main()
