"""GPU parity of the HIP env kernels: the golden traces recorded from the REAL reference GraphEnv are
replayed through HipGraphVectorEnv (C ABI: mel_env_reset / mel_env_step) - bit-exact on every output."""
import glob
import os

import numpy as np
import pytest
import torch

from tests.trace_replay import replay, scripted_kwargs

pytestmark = pytest.mark.gpu

TRACES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "env_trace_*.npz")))


class OneEnvAdapter:
    """PettingZooEnv-level view of env `k` of a HipGraphVectorEnv."""

    def __init__(self, venv, k):
        self.venv, self.k = venv, k
        self.rewards = [0] * venv.n

    def reset(self):
        obs, info = self.venv.reset([self.k])
        return obs[0], info[0]

    def step(self, a):
        obs, rew, term, trunc, info = self.venv.step([a], [self.k])
        self.rewards = list(rew[0])
        return obs[0], rew[0], bool(term[0]), bool(trunc[0]), info[0]

    def state(self):
        v, k = self.venv, self.k
        sets = v.node_sets()[k].cpu().numpy().view(np.uint64)
        from melissa_amd import _lib as L
        return dict(agents_mask=sets[L.SET_AGENTS], alive_mask=sets[L.SET_ALIVE],
                    terminated_mask=sets[L.SET_TERMINATED], has_message_mask=sets[L.SET_HAS_MESSAGE],
                    interested_mask=sets[L.SET_INTERESTED], scripted_mask=sets[L.SET_SCRIPTED],
                    origin=int(v.scalars()[k, L.S_ORIGIN]),
                    pos=v.positions()[k].cpu().numpy(), one_hop=v.one_hop()[k].cpu().numpy().view(np.uint64),
                    two_hop=v.two_hop()[k].cpu().numpy().view(np.uint64))


def build_venv(tr, env_num=1, slot=0):
    from melissa_amd.env import Graph, HipGraphVectorEnv
    n = int(tr["n"])
    graphs = [Graph(tr["pool_pos"][k].copy(), tr["pool_adj"][k].copy()) for k in range(tr["pool_pos"].shape[0])]
    lr = float(tr["local_ratio"])
    kw = dict(env_num=env_num, number_of_agents=n, dynamic_graph=bool(tr["dynamic"]),
              local_ratio=None if lr < 0 else lr, seed=int(tr["env_seed"]) - slot, max_moves=64,
              construct_like_reference=2)         # the traces drive a bare GraphEnv (no PettingZooEnv.__init__ reset)
    kw.update(scripted_kwargs(tr))
    if "is_testing" in tr.files and bool(tr["is_testing"]):          # core.py:348-370 evaluation schedule
        kw.update(is_testing=True, num_test_episodes=int(tr["num_test_episodes"]))
    if bool(tr["fixed_graph"]):
        return HipGraphVectorEnv(graph=graphs[0], **kw)
    return HipGraphVectorEnv(graph_pool=graphs, **kw)


@pytest.mark.parametrize("path", TRACES, ids=[os.path.basename(p)[10:-4] for p in TRACES])
def test_hip_env_matches_reference_trace(path):
    tr = np.load(path)
    venv = build_venv(tr)
    pz = OneEnvAdapter(venv, 0)
    rows = replay(tr, pz, pz.state)
    assert rows > 300
    assert int(venv.scalars()[0, 11]) == 0          # MEL_S_ERROR: never ran out of movement offsets


def test_hip_env_trace_in_a_busy_batch():
    """The traced env sits in slot 5 of a 9-env batch whose other envs are stepped with other actions in
    the same launches: envs must not interfere."""
    tr = np.load([p for p in TRACES if "n20_pool_dynamic" in p][0])
    venv = build_venv(tr, env_num=9, slot=5)        # env k is seeded seed+k -> slot 5 gets the trace's seed
    n = int(tr["n"])
    rng = np.random.RandomState(0)
    venv.reset([i for i in range(9) if i != 5])

    class Busy(OneEnvAdapter):
        def step(self, a):
            others = [i for i in range(9) if i != 5]
            obs, rew, term, trunc, info = self.venv.step(rng.randint(0, 2, size=8), others)
            for i, t, inf in zip(others, term, info):
                if t and inf.get("explicit_reset"):
                    self.venv.reset([i])
            return super().step(a)

    busy = Busy(venv, 5)
    rows = replay(tr, busy, busy.state)
    assert rows > 300


def test_scripted_argument_checks_and_partition():
    """core.py:143-152 argument errors; scripted | decision makers = all nodes, origin never scripted, dm column =
    not scripted, scripted nodes never selected in training mode (test_mixed_scripted_learned_agents.py:29-100)."""
    from melissa_amd import _lib as L
    from melissa_amd.env import HipGraphVectorEnv, synthetic_graph_pool
    pool = synthetic_graph_pool(20, 4, 10)
    for bad in (-0.1, 1.1):
        with pytest.raises(ValueError, match=r"must be in \[0.0, 1.0\]"):
            HipGraphVectorEnv(2, 20, graph_pool=pool, scripted_agents_ratio=bad)
    with pytest.raises(ValueError, match="no heuristic can be set"):
        HipGraphVectorEnv(2, 20, graph_pool=pool, heuristic="silent")
    with pytest.raises(ValueError, match="Unknown heuristic policy"):
        HipGraphVectorEnv(2, 20, graph_pool=pool, scripted_agents_ratio=0.5, heuristic="probabilistic_gossip")
    venv = HipGraphVectorEnv(16, 20, graph_pool=pool, scripted_agents_ratio=0.5, heuristic="simple_broadcast", seed=3)
    for _ in range(3):
        venv.reset()
        sets = venv.node_sets().cpu().numpy().view(np.uint64)
        origin = venv.scalars()[:, L.S_ORIGIN].cpu().numpy()
        dm = venv.obs_matrix().cpu().numpy().reshape(16, 20, 8)[:, :, 7]
        for b in range(16):
            scripted = int(sets[b, L.SET_SCRIPTED])
            assert bin(scripted).count("1") in (9, 10) and not (scripted >> int(origin[b])) & 1
            assert int(sets[b, L.SET_AGENTS]) & scripted == 0
            np.testing.assert_array_equal(dm[b], [0.0 if (scripted >> i) & 1 else 1.0 for i in range(20)])


def test_distance_rule_on_threshold_pairs():
    """The float64 edge rule dx*dx + dy*dy <= 0.2**2 (nx.geometric_edges, core.py:311) on positions where it is decided
    in the last bits: a lattice of multiples of 0.2 (0.4 - 0.2 gives an edge, 0.8 - 0.6 does not), movement offsets
    zeroed so that every world step re-evaluates the rule on the same points.  The expectation is numpy's float64 expression."""
    from melissa_amd import _lib
    from melissa_amd.collect import RoundLoop, sample_episode_table
    from melissa_amd.env import Graph, HipGraphVectorEnv
    from melissa_amd.networks import LDGNNetwork
    from melissa_amd.policy import DQNPolicy
    n = 30
    xs = np.array([0.2 * k for k in range(6)], dtype=np.float64)
    pos = np.array([[xs[i % 6], xs[i // 6]] for i in range(n)], dtype=np.float64)
    shifted = pos * 3.0 + 1.0                               # second graph: coordinates up to 4, spacing 0.6 (no edges)
    near = pos.copy()
    near[:, 0] += np.linspace(-3e-9, 3e-9, n)               # third: the same lattice perturbed at the 1e-9 level
    graphs = [Graph.from_positions(p, 0.2) for p in (pos, shifted, near)]
    venv = HipGraphVectorEnv(env_num=6, number_of_agents=n, graph_pool=graphs, dynamic_graph=True, seed=3, max_moves=8,
                             device="cuda", construct_like_reference=False)
    packed, table = sample_episode_table(venv, 6, seed=11)
    packed["moves"][:] = 0.0
    net = LDGNNetwork(5, 128, 2, 4, n, dueling_param=({"hidden_sizes": [128, 128]}, {"hidden_sizes": [128, 128]}),
                      device="cuda", backend="hip")
    loop = RoundLoop(venv, DQNPolicy(net), episodes=(packed, table), seed=11, eps=0.5, use_graph=False)
    seen = set()
    for _ in range(12):
        loop.step()
        torch.cuda.synchronize()
        p = venv.positions().cpu().numpy()
        hop = venv.one_hop().cpu().numpy().view(np.uint64)
        for b in range(6):
            dx = p[b, :, None, 0] - p[b, None, :, 0]
            dy = p[b, :, None, 1] - p[b, None, :, 1]
            adj = (dx * dx + dy * dy) <= 0.2 ** 2
            np.fill_diagonal(adj, False)
            want = (adj.astype(np.uint64) << np.arange(n, dtype=np.uint64)[None, :]).sum(axis=1, dtype=np.uint64)
            np.testing.assert_array_equal(hop[b, :n], want)
            seen.add(int(adj.sum()))
    assert len(seen) >= 2                                    # lattices with and without edges were both exercised
    # (bit 0 = an episode outlived max_moves - the edgeless lattice does; anything else is a fault)
    assert int((venv.scalars()[:, _lib.S_ERROR] & ~1).sum()) == 0
