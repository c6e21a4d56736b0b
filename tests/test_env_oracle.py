"""The env oracle (oracle/env_oracle.py) against the reference: golden traces produced by the REAL
reference GraphEnv (tests/golden/make_env_golden.py) and the reference's own known-answer tests."""
import glob
import os

import numpy as np
import pytest

from oracle import env_oracle as eo
from tests.trace_replay import replay

TRACES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "env_trace_*.npz")))


def build_oracle_env(tr):
    n = int(tr["n"])
    pool = [eo.GraphSpec(tr["pool_pos"][k], [int(x) for x in tr["pool_adj"][k]])
            for k in range(tr["pool_pos"].shape[0])]
    lr = float(tr["local_ratio"])
    kw = dict(number_of_agents=n, dynamic_graph=bool(tr["dynamic"]),
              local_ratio=None if lr < 0 else lr,
              np_random=np.random.Generator(np.random.PCG64(np.random.SeedSequence(int(tr["env_seed"])))))
    if bool(tr["fixed_graph"]):
        return eo.OracleGraphEnv(graph=pool[0], **kw)
    return eo.OracleGraphEnv(graph_pool=pool, **kw)


def oracle_state(env):
    return dict(agents_mask=env.agents, alive_mask=env.alive, terminated_mask=env.terminated,
                has_message_mask=env.has_message, interested_mask=env.interested,
                origin=env.origin_agent, pos=env.pos, one_hop=env.adj, two_hop=env.two_hop)


@pytest.mark.parametrize("path", TRACES, ids=[os.path.basename(p)[10:-4] for p in TRACES])
def test_oracle_matches_reference_trace(path):
    tr = np.load(path)
    env = build_oracle_env(tr)
    pz = eo.OraclePettingZooEnv.__new__(eo.OraclePettingZooEnv)
    pz.env, pz.n, pz.rewards = env, env.n, [0] * env.n     # the generator does not reset in __init__
    rows = replay(tr, pz, lambda: oracle_state(env))
    assert rows > 300


def test_traces_present():
    assert len(TRACES) >= 7


# ---- the reference's own known-answer pins (tests/unit/graph_env/env/utils/test_core.py) ----------
EDGES = [(0, 1), (0, 2), (0, 3), (0, 4), (3, 4), (2, 5), (2, 6), (3, 7), (7, 8), (7, 9), (8, 9),
         (4, 11), (3, 10)]                                                    # test_core.py:25-27
MOVED = [(0, 1), (0, 3), (1, 5), (2, 3), (2, 5), (2, 6), (5, 6), (3, 4), (3, 7), (7, 8), (4, 11),
         (3, 10), (10, 11)]                                                   # test_core.py:71-74


def rows_to_masks(rows):
    return [sum(1 << j for j, v in enumerate(r) if v) for r in rows]


def test_one_hop_known_answers():                                            # test_core.py:97-110
    g = eo.GraphSpec.from_edges(12, EDGES)
    expect = rows_to_masks([
        [0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
        [1, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0], [1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0],
        [1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1], [0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0],
        [0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 1, 0, 0, 0, 0, 1, 1, 0, 0],
        [0, 0, 0, 0, 0, 0, 0, 1, 0, 1, 0, 0], [0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0],
        [0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0]])
    env = eo.OracleGraphEnv(12, graph=g, np_random=np.random.default_rng(9))
    assert env.adj == expect


def test_two_hop_known_answers():                                            # test_core.py:133-146
    g = eo.GraphSpec.from_edges(12, EDGES)
    expect = rows_to_masks([
        [0, 1, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1], [1, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0],
        [1, 1, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0], [1, 1, 1, 0, 1, 0, 0, 1, 1, 1, 1, 1],
        [1, 1, 1, 1, 0, 0, 0, 1, 0, 0, 1, 1], [1, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0],
        [1, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0], [1, 0, 0, 1, 1, 0, 0, 0, 1, 1, 1, 0],
        [0, 0, 0, 1, 0, 0, 0, 1, 0, 1, 0, 0], [0, 0, 0, 1, 0, 0, 0, 1, 1, 0, 0, 0],
        [1, 0, 0, 1, 1, 0, 0, 1, 0, 0, 0, 0], [1, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0]])
    env = eo.OracleGraphEnv(12, graph=g, np_random=np.random.default_rng(9))
    assert env.two_hop == expect


def test_two_hop_after_edge_rewrite():                                       # test_core.py:149-169
    g = eo.GraphSpec.from_edges(12, MOVED)
    expect = rows_to_masks([
        [0, 1, 1, 1, 1, 1, 0, 1, 0, 0, 1, 0], [1, 0, 1, 1, 0, 1, 1, 0, 0, 0, 0, 0],
        [1, 1, 0, 1, 1, 1, 1, 1, 0, 0, 1, 0], [1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 1, 1],
        [1, 0, 1, 1, 0, 0, 0, 1, 0, 0, 1, 1], [1, 1, 1, 1, 0, 0, 1, 0, 0, 0, 0, 0],
        [0, 1, 1, 1, 0, 1, 0, 0, 0, 0, 0, 0], [1, 0, 1, 1, 1, 0, 0, 0, 1, 0, 1, 0],
        [0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
        [1, 0, 1, 1, 1, 0, 0, 1, 0, 0, 0, 1], [0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 1, 0]])
    assert eo.two_hop_masks(g.adj) == expect
    one_hop_0 = rows_to_masks([[0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0]])[0]      # test_core.py:118,156
    assert g.adj[0] == one_hop_0


def test_source_first_wave_counts():
    """test_core.py:185-192 (first assertions, which do not depend on scripted heuristics): after
    reset the source has transmitted once to exactly its one-hop set and each neighbour received 1."""
    g = eo.GraphSpec.from_edges(12, EDGES)
    env = eo.OracleGraphEnv(12, graph=g, np_random=np.random.default_rng(42))
    src = env.origin_agent
    assert env.agent_msgs[src] == 1 and env.messages_transmitted == 1
    for j in range(12):
        assert env.received_count[j] == ((g.adj[src] >> j) & 1)
    assert env.has_message == (g.adj[src] | (1 << src))
