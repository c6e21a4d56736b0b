"""Generate golden network vectors (obs rows + expected logits/intermediates).

Run:  python tests/golden/make_net_golden.py          (needs only this repo: no reference import)

Obs rows are captured from the env oracle (oracle/env_oracle.py, itself pinned to the real reference
by env_trace_*.npz) on synthetic connected RGGs, plus hand-made edge cases.  Expected outputs come from
oracle/net_oracle.py formulation "edges"; weights are regenerated from ``weight_seed`` by
``net_oracle.init_weights`` (legacy numpy RandomState stream), so only obs + outputs are stored.
PARITY UNPINNED w.r.t. the real PyG/tianshou wheels (absent here) - see oracle/net_oracle.py header.
"""
import os
import sys

import networkx as nx
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import env_oracle as eo      # noqa: E402
from oracle import net_oracle as no      # noqa: E402


def rgg_pool(n, count, first_seed):
    out, s = [], first_seed
    while len(out) < count:
        g = nx.random_geometric_graph(n, 0.2, seed=s)
        if nx.is_connected(g):
            pos = np.array([g.nodes[i]["pos"] for i in range(n)], dtype=np.float64)
            out.append(eo.GraphSpec(pos))
        s += 1
    return out


def capture_obs(n, rows, seed, dynamic=True):
    """Play oracle episodes with random actions; keep every k-th observation row."""
    env = eo.OracleGraphEnv(n, graph_pool=rgg_pool(n, 4, 300 + n), dynamic_graph=dynamic,
                            np_random=np.random.Generator(np.random.PCG64(seed)))
    pz = eo.OraclePettingZooEnv(env)
    rng = np.random.RandomState(seed)
    out = []
    obs, info = pz.reset()
    t = 0
    while len(out) < rows:
        if t % 3 == 0:
            out.append(obs["obs"].copy())
        obs, rew, term, trunc, info = pz.step(int(rng.randint(2)))
        t += 1
        if term and info.get("explicit_reset"):
            obs, info = pz.reset()
    return np.stack(out).astype(np.float32)


def edge_case_obs(n, kind, rng):
    """Hand-made rows: 'clique' all nodes at one point (degree n-1 > 32 when n=50 -> neighbour cap),
    'isolated' nodes spread on a coarse lattice (no edges: self-loop only attention),
    'scripted' a captured-like row with dm flag 0 on some nodes."""
    row = np.zeros(8 * n + 1, dtype=np.float32)
    m = row[:-1].reshape(n, 8)
    if kind == "clique":
        m[:, 0:2] = 0.5 + 0.01 * rng.uniform(-1, 1, size=(n, 2))
    elif kind == "isolated":
        side = int(np.ceil(np.sqrt(n)))
        for i in range(n):
            m[i, 0], m[i, 1] = 0.25 * (i % side), 0.25 * (i // side)
    else:
        m[:, 0:2] = rng.uniform(0, 1, size=(n, 2))
    m[:, 2] = rng.randint(0, 8, size=n)
    m[:, 3] = rng.randint(0, 4, size=n)
    m[:, 4] = rng.randint(0, 2, size=n)
    m[:, 5] = rng.randint(0, 2, size=n)
    m[:, 6] = rng.randint(0, 2, size=n)
    m[:, 7] = 1.0
    if kind == "scripted":
        m[rng.choice(n, size=n // 3, replace=False), 7] = 0.0
    row[-1] = rng.randint(0, n)
    return row


def make(name, n, rows, seed, weight_seed):
    rng = np.random.RandomState(seed)
    obs = capture_obs(n, rows, seed)
    extra = np.stack([edge_case_obs(n, k, rng) for k in ("clique", "isolated", "scripted", "scripted")])
    obs = np.concatenate([obs, extra]).astype(np.float32)
    out = dict(obs=obs, n=np.int64(n), weight_seed=np.int64(weight_seed))
    with torch.no_grad():
        sd = no.init_weights("l_dgn", seed=weight_seed, random_conv_bias=True)
        logits, inter = no.ldgn_forward(sd, obs, n, return_intermediates=True)
        out["ldgn_logits"] = logits.numpy()
        for k in ("x_1", "x_2", "x_3"):
            out[f"ldgn_{k}"] = inter[k].numpy()
        out["adj"] = np.packbits(inter["adj"].numpy(), axis=-1, bitorder="little")
        sd = no.init_weights("dgn_r", seed=weight_seed + 2)
        logits, inter = no.dgnr_forward(sd, obs, n, return_intermediates=True)
        out["dgnr_logits"] = logits.numpy()
        for k in ("x_1", "x_2", "x_3"):
            out[f"dgnr_{k}"] = inter[k].numpy()
        sd = no.init_weights("hl_dgn", seed=weight_seed + 1, random_conv_bias=True)
        for agg in ("max", "mean", "add"):
            logits, inter = no.hldgn_forward(sd, obs, n, aggregator=agg, return_intermediates=True)
            out[f"hldgn_{agg}_logits"] = logits.numpy()
            out[f"hldgn_{agg}_pooled"] = inter["pooled"].numpy()
    path = os.path.join(HERE, f"net_golden_{name}.npz")
    np.savez_compressed(path, **out)
    print(name, obs.shape, f"{os.path.getsize(path)/1024:.0f} KiB")


if __name__ == "__main__":
    torch.set_num_threads(4)
    only = sys.argv[1] if len(sys.argv) > 1 else None
    if only in (None, "n20"):
        make("n20", 20, 28, seed=21, weight_seed=9)
    if only in (None, "n50"):
        make("n50", 50, 12, seed=22, weight_seed=9)
    if only in (None, "n12"):
        make("n12", 12, 6, seed=23, weight_seed=5)
    if only in (None, "n100"):      # the reference CLI's third size (--n-agents 100): two-word node sets in the kernels
        make("n100", 100, 8, seed=24, weight_seed=9)
