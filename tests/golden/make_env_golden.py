"""Generate golden env traces from the REAL reference ``GraphEnv`` (build container only).

Run:  python tests/golden/make_env_golden.py
Needs /root/reference (read-only) - imported through tests/golden/ref_standins.py.  The traces are
DATA (inputs + expected outputs); no reference source is stored.  ``tests/test_env_oracle.py`` replays
them through ``oracle/env_oracle.py``; the GPU tests replay them through the HIP env kernels.

Every trace drives one env through the [3P] tianshou ``PettingZooEnv.step`` protocol (restated in
``RefPettingZoo`` below) with the collector's reset rule
(multi_agent_collector.py:261-264: reset on ``explicit_reset`` or when N agents reported done).
"""
import os
import pickle
import sys
import tempfile

import networkx as nx
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_standins  # noqa: E402

LOGGER_KEYS = ("total_messages_transmitted", "coverage", "messages_sent", "messages_received",
               "n_neighbours", "interested_agents", "coverage_interested_fraction",
               "coverage_interested_count", "uninterested_with_message", "episode_rewards_sum")


def connected_rggs(n, count, first_seed=0, radius=0.2):
    """random_geometric_graph(n, 0.2, seed=s), s = 0,1,..., connected only."""
    out, s = [], first_seed
    while len(out) < count:
        g = nx.random_geometric_graph(n, radius, seed=s)
        if nx.is_connected(g):
            out.append((s, g))
        s += 1
    return out


def fixture_graph_12():
    """The reference's own test fixture graph (tests/unit/.../test_core.py:23-31): 12 nodes, 13 edges,
    all positions (0, 0)."""
    g = nx.Graph()
    g.add_edges_from([(0, 1), (0, 2), (0, 3), (0, 4), (3, 4), (2, 5), (2, 6), (3, 7), (7, 8), (7, 9),
                      (8, 9), (4, 11), (3, 10)])
    for node in g.nodes:
        g.nodes[node]["pos"] = (0, 0)
    return g


class RefPettingZoo:
    """[3P] tianshou PettingZooEnv.reset/step around the real GraphEnv."""

    def __init__(self, env):
        self.env = env
        self.n = env.number_of_agents
        self.rewards = [0] * self.n

    def _pack(self):
        observation, _r, term, trunc, info = self.env.last()
        return ({"agent_id": self.env.agent_selection, "obs": observation["observation"],
                 "mask": [bool(m == 1) for m in observation["action_mask"]]}, term, trunc, info)

    def reset(self):
        self.env.reset()
        return self._pack()

    def step(self, action):
        self.env.step(action)
        out = self._pack()
        for agent_id, reward in self.env.rewards.items():
            self.rewards[int(agent_id)] = reward
        return out


def mask_of(names):
    m = 0
    for a in names:
        m |= 1 << int(a)
    return m


def words(m, n):
    """A node set as stored in the traces: np.uint64 up to 64 nodes (the layout the HIP kernels use), an array of
    ceil(n / 64) 64-bit words (low word first) beyond."""
    if n <= 64:
        return np.uint64(m)
    return np.array([(m >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range((n + 63) // 64)], dtype=np.uint64)


def record(rows, pz, packed, n):
    obs, term, trunc, info = packed
    env = pz.env
    w = env.world
    stats = info.get("logger_stats")
    rows["agent_id"].append(int(obs["agent_id"]))
    rows["obs"].append(np.asarray(obs["obs"], dtype=np.float32))
    rows["mask"].append(np.asarray(obs["mask"], dtype=bool))
    rows["rew"].append(np.asarray(pz.rewards, dtype=np.float64))
    rows["term"].append(bool(term))
    rows["trunc"].append(bool(trunc))
    rows["env_step"].append(int(info["env_step"]))
    rows["environment_step"].append(bool(info["environment_step"]))
    rows["explicit_reset"].append(bool(info["explicit_reset"]))
    rows["active_nb"].append(np.asarray(info["active_one_hop_neighbors"], dtype=bool))
    rows["has_stats"].append(stats is not None)
    rows["stats"].append(np.array([float(stats[k]) for k in LOGGER_KEYS] if stats is not None
                                  else [0.0] * len(LOGGER_KEYS), dtype=np.float64))
    rows["agents_mask"].append(words(mask_of(env.agents), n))
    rows["alive_mask"].append(words(mask_of(env.terminations.keys()), n))
    rows["terminated_mask"].append(words(mask_of(k for k, v in env.terminations.items() if v), n))
    rows["has_message_mask"].append(words(mask_of(a.name for a in w.agents if a.state.has_message), n))
    rows["interested_mask"].append(words(mask_of(a.name for a in w.agents if a.is_interested), n))
    rows["scripted_mask"].append(words(mask_of(a.name for a in w.agents if a.is_scripted), n))
    rows["origin"].append(int(w.origin_agent))
    rows["pos"].append(np.array([a.pos for a in w.agents], dtype=np.float64))
    rows["one_hop"].append(np.array([words(mask_of(np.where(a.one_hop_neighbours_ids)[0]), n) for a in w.agents],
                                    dtype=np.uint64))
    rows["two_hop"].append(np.array([words(mask_of(np.where(a.two_hop_neighbours_ids)[0]), n) for a in w.agents],
                                    dtype=np.uint64))


def run_trace(name, n, mode, dynamic, steps, env_seed, tape_seed, local_ratio=None, n_graphs=6, num_test_episodes=None,
              scripted_agents_ratio=0.0, heuristic=None):
    """mode: 'pool' (graph_topologies/training_N/* files written here from synthetic RGGs),
    'testing' (is_testing=True over graph_topologies/testing_N/*, core.py:348-370),
    'fixed' (graph= argument, one RGG), 'fixture12' (the reference's test graph)."""
    ref_graph, _ = ref_standins.import_reference()
    ref_standins.DEFAULT_SEED = env_seed
    cwd = os.getcwd()
    meta = {}
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            if mode == "testing":
                os.makedirs(f"graph_topologies/testing_{n}")
                graphs = connected_rggs(n, n_graphs, first_seed=300 * n)
                for s, g in graphs:
                    with open(f"graph_topologies/testing_{n}/rgg_{s:05d}.gpickle", "wb") as f:
                        pickle.dump(g, f)     # a file this script wrote, not a reference artefact
                env = ref_graph.GraphEnv(graph=None, number_of_agents=n, radius=0.2, dynamic_graph=dynamic,
                                         local_ratio=local_ratio, is_testing=True, num_test_episodes=num_test_episodes,
                                         scripted_agents_ratio=scripted_agents_ratio, heuristic=heuristic)
                order = [os.path.basename(p) for p in env.world.test_graphs]          # sorted glob (core.py:170-172)
                by_name = {f"rgg_{s:05d}.gpickle": g for s, g in graphs}
                pool = [by_name[o] for o in order]
            elif mode == "pool":
                os.makedirs(f"graph_topologies/training_{n}")
                graphs = connected_rggs(n, n_graphs, first_seed=100 * n)
                for s, g in graphs:
                    with open(f"graph_topologies/training_{n}/rgg_{s:05d}.gpickle", "wb") as f:
                        pickle.dump(g, f)     # a file this script wrote, not a reference artefact
                env = ref_graph.GraphEnv(graph=None, number_of_agents=n, radius=0.2,
                                         dynamic_graph=dynamic, local_ratio=local_ratio,
                                         scripted_agents_ratio=scripted_agents_ratio, heuristic=heuristic)
                order = [os.path.basename(p) for p in env.world.train_graphs]
                by_name = {f"rgg_{s:05d}.gpickle": g for s, g in graphs}
                pool = [by_name[o] for o in order]          # glob order = choice index order
            else:
                g = fixture_graph_12() if mode == "fixture12" else connected_rggs(n, 1, 7)[0][1]
                pool = [g.copy()]
                env = ref_graph.GraphEnv(graph=g, number_of_agents=n, radius=0.2,
                                         dynamic_graph=dynamic, local_ratio=local_ratio)
            meta["pool_pos"] = np.array([[gg.nodes[i]["pos"] for i in range(n)] for gg in pool],
                                        dtype=np.float64)
            adj_int = [[0] * n for _ in pool]
            for k, gg in enumerate(pool):
                for u, v in gg.edges():
                    adj_int[k][u] |= 1 << v
                    adj_int[k][v] |= 1 << u
            meta["pool_adj"] = np.array([[words(m, n) for m in row] for row in adj_int], dtype=np.uint64)
            pz = RefPettingZoo(env)
            tape = np.random.RandomState(tape_seed).randint(0, 2, size=steps).astype(np.int8)
            rows = {k: [] for k in ("agent_id obs mask rew term trunc env_step environment_step "
                                    "explicit_reset active_nb has_stats stats agents_mask alive_mask "
                                    "terminated_mask has_message_mask interested_mask scripted_mask origin pos "
                                    "one_hop two_hop was_reset").split()}
            packed = pz.reset()
            record(rows, pz, packed, n)
            rows["was_reset"].append(True)
            done_count = 0
            for t in range(steps):
                packed = pz.step(int(tape[t]))
                record(rows, pz, packed, n)
                rows["was_reset"].append(False)
                _, term, trunc, info = packed
                if term or trunc:
                    done_count += 1
                    if done_count == n or info.get("explicit_reset", False):
                        packed = pz.reset()
                        record(rows, pz, packed, n)
                        rows["was_reset"].append(True)
                        done_count = 0
        finally:
            os.chdir(cwd)
    out = {k: np.array(v) for k, v in rows.items()}
    out.update(meta)
    out.update(n=np.int64(n), dynamic=np.bool_(dynamic), env_seed=np.int64(env_seed), tape=tape,
               fixed_graph=np.bool_(mode not in ("pool", "testing")), is_testing=np.bool_(mode == "testing"),
               num_test_episodes=np.int64(num_test_episodes or 0),
               scripted_agents_ratio=np.float64(scripted_agents_ratio), heuristic=np.str_(heuristic or ""),
               local_ratio=np.float64(-1.0 if local_ratio is None else local_ratio))
    path = os.path.join(HERE, f"env_trace_{name}.npz")
    np.savez_compressed(path, **out)
    n_reset = int(np.sum(out["was_reset"]))
    print(f"{name}: {len(out['agent_id'])} rows, {n_reset} resets -> {os.path.getsize(path)/1024:.0f} KiB")


def main(only=None):
    if only == "n100":          # the reference CLI's third size (--n-agents 100, common.py:49): two-word node sets
        run_trace("n100_pool_dynamic", 100, "pool", True, 700, env_seed=23, tape_seed=13, n_graphs=4)
        run_trace("n100_pool_static_lr", 100, "pool", False, 400, env_seed=24, tape_seed=14, local_ratio=0.5, n_graphs=3)
        run_trace("n100_scripted_broadcast", 100, "pool", True, 400, env_seed=25, tape_seed=15, n_graphs=3,
                  scripted_agents_ratio=0.3, heuristic="simple_broadcast")
        run_trace("n70_testing_dynamic", 70, "testing", True, 500, env_seed=26, tape_seed=16, n_graphs=4, num_test_episodes=6)
        return
    run_trace("n20_pool_static", 20, "pool", False, 700, env_seed=11, tape_seed=1)
    run_trace("n20_pool_dynamic", 20, "pool", True, 700, env_seed=12, tape_seed=2)
    run_trace("n50_pool_dynamic", 50, "pool", True, 900, env_seed=13, tape_seed=3)
    run_trace("n50_pool_static_lr", 50, "pool", False, 500, env_seed=14, tape_seed=4, local_ratio=0.5)
    run_trace("n20_fixed_dynamic", 20, "fixed", True, 400, env_seed=15, tape_seed=5)
    run_trace("n12_fixture_static", 12, "fixture12", False, 300, env_seed=16, tape_seed=6)
    run_trace("n12_fixture_dynamic", 12, "fixture12", True, 300, env_seed=17, tape_seed=7)
    run_trace("n20_testing_dynamic", 20, "testing", True, 900, env_seed=18, tape_seed=8, num_test_episodes=7)
    # scripted agents (heuristics/core.py): training mode keeps them out of the active set, testing mode steps them
    run_trace("n20_scripted_broadcast", 20, "pool", True, 600, env_seed=19, tape_seed=9,
              scripted_agents_ratio=0.3, heuristic="simple_broadcast")
    run_trace("n50_scripted_interested", 50, "pool", False, 500, env_seed=20, tape_seed=10,
              scripted_agents_ratio=0.4, heuristic="broadcast_if_any_interested")
    run_trace("n20_scripted_interested_dynamic", 20, "pool", True, 500, env_seed=22, tape_seed=12,
              scripted_agents_ratio=0.4, heuristic="broadcast_if_any_interested")
    run_trace("n20_scripted_silent_testing", 20, "testing", True, 500, env_seed=21, tape_seed=11, num_test_episodes=5,
              scripted_agents_ratio=0.5, heuristic="silent")


    main("n100")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
