"""The env oracle (oracle/env_oracle.py) against the reference: golden traces produced by the REAL
reference GraphEnv (tests/golden/make_env_golden.py) and the reference's own known-answer tests."""
import glob
import os

import numpy as np
import pytest

from oracle import env_oracle as eo
from melissa_amd.env.episodes import set_to_int
from tests.trace_replay import replay, scripted_kwargs, set_ints

TRACES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "env_trace_*.npz")))


def build_oracle_env(tr):
    n = int(tr["n"])
    pool = [eo.GraphSpec(tr["pool_pos"][k], set_ints(tr["pool_adj"][k]))
            for k in range(tr["pool_pos"].shape[0])]
    lr = float(tr["local_ratio"])
    kw = dict(number_of_agents=n, dynamic_graph=bool(tr["dynamic"]),
              local_ratio=None if lr < 0 else lr,
              np_random=np.random.Generator(np.random.PCG64(np.random.SeedSequence(int(tr["env_seed"])))))
    kw.update(scripted_kwargs(tr))
    if "is_testing" in tr.files and bool(tr["is_testing"]):          # core.py:348-370 evaluation schedule
        kw.update(is_testing=True, num_test_episodes=int(tr["num_test_episodes"]))
    if bool(tr["fixed_graph"]):
        return eo.OracleGraphEnv(graph=pool[0], **kw)
    return eo.OracleGraphEnv(graph_pool=pool, **kw)


def oracle_state(env):
    return dict(agents_mask=env.agents, alive_mask=env.alive, terminated_mask=env.terminated,
                has_message_mask=env.has_message, interested_mask=env.interested, scripted_mask=env.scripted,
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


# ---- scripted agents: the reference's own tests restated against the oracle ---------------------------------
def _fixture12_spec():
    edges = [(0, 1), (0, 2), (0, 3), (0, 4), (3, 4), (2, 5), (2, 6), (3, 7), (7, 8), (7, 9), (8, 9), (4, 11), (3, 10)]
    return eo.GraphSpec.from_edges(12, edges, pos=np.zeros((12, 2)))


def test_world_simple_broadcast_waves():                                      # test_core.py:173-215
    """ratio 1.0 + simple_broadcast: the source transmits to exactly its one-hop set during reset, and after one
    more world step every first-wave node has forwarded to all of its neighbours."""
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(42)))
    env = eo.OracleGraphEnv(12, graph=_fixture12_spec(), np_random=rng, scripted_agents_ratio=1.0,
                            heuristic="simple_broadcast")
    src = env.origin_agent
    assert env.scripted == env.full                                           # ratio 1: even the origin is scripted
    first_wave = env.adj[src]
    # a scripted first-wave node with a higher id than the source relays in the SAME step (core.py:249-254 walks
    # ids in order and re-reads has_message), so the message may already be past the first wave
    assert env.has_message & first_wave == first_wave and env.agent_msgs[src] == 1
    for nbr in eo.bits(first_wave):
        assert env.received_count[nbr] >= 1
    env._world_step()
    for relay in eo.bits(first_wave):
        assert env.agent_msgs[relay] == 1                                     # has_taken_action: forwards once
        assert env.has_message & env.adj[relay] == env.adj[relay]
    before = list(env.agent_msgs)
    for _ in range(6):
        env._world_step()
    assert env.has_message == env.full and max(env.agent_msgs) == 1           # everyone forwarded exactly once
    assert sum(env.agent_msgs) >= sum(before)


def test_invalid_scripted_ratio_and_heuristic():                              # test_mixed...py:29-36, core.py:143-152
    for bad in (-0.1, 1.1):
        with pytest.raises(ValueError, match=r"must be in \[0.0, 1.0\]"):
            eo.OracleGraphEnv(12, graph=_fixture12_spec(), scripted_agents_ratio=bad)
    with pytest.raises(ValueError, match="no heuristic can be set"):
        eo.OracleGraphEnv(12, graph=_fixture12_spec(), scripted_agents_ratio=0.0, heuristic="silent")
    with pytest.raises(ValueError, match="Unknown heuristic policy"):
        eo.OracleGraphEnv(12, graph=_fixture12_spec(), scripted_agents_ratio=0.5, heuristic="nope")


def test_scripted_sampling_reproducible_and_partition():                      # test_mixed...py:53-100
    mk = lambda s: np.random.Generator(np.random.PCG64(np.random.SeedSequence(s)))
    pool = [eo.GraphSpec(g.pos.copy(), [set_to_int(m) for m in g.one_hop]) for g in
            __import__("melissa_amd.env", fromlist=["synthetic_graph_pool"]).synthetic_graph_pool(20, 2, 50)]
    env = eo.OracleGraphEnv(20, graph_pool=pool, np_random=mk(1), scripted_agents_ratio=0.3, heuristic=None)
    env.reset(seed=123)
    s1 = env.scripted
    assert popcount_ok(s1, 20, 0.3, env.origin_agent)
    env.reset(seed=123)
    assert env.scripted == s1                                                 # same seed, same scripted set
    env.reset(seed=999)
    assert env.scripted != s1
    # partition: scripted | decision makers = all nodes, origin never scripted, dm flag column = not scripted
    env = eo.OracleGraphEnv(20, graph_pool=pool, np_random=mk(2), scripted_agents_ratio=0.5, heuristic="simple_broadcast")
    for _ in range(5):
        env.reset()
        assert not (env.scripted >> env.origin_agent) & 1
        dm = env.obs_matrix[:, 7]
        assert all(dm[i] == (0.0 if (env.scripted >> i) & 1 else 1.0) for i in range(20))
        assert env.agents & env.scripted == 0                                 # training mode: never selected


def popcount_ok(mask, n, ratio, origin):
    cnt = bin(mask).count("1")
    want = int(round(ratio * n))
    return cnt in (want, want - 1) and not (mask >> origin) & 1               # the origin is discarded if drawn
