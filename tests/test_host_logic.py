"""Host-side logic that needs no GPU: episode sampling mirrors World.reset's RNG protocol, pool packing,
graph helpers, network drop-in surface (ctor / state_dict keys / errors / torch path), policy helpers."""
import numpy as np
import pytest
import torch

from melissa_amd.env.episodes import (EpisodeSampler, Graph, movement_offsets, pack_episodes,
                                      synthetic_graph_pool)
from melissa_amd.networks import DGNRNetwork, HLDGNNetwork, LDGNNetwork
from melissa_amd.policy import DQNPolicy, MultiAgentSharedPolicy
from oracle import env_oracle as eo
from oracle import net_oracle as no

DUEL = lambda: ({"hidden_sizes": [128, 128]}, {"hidden_sizes": [128, 128]})


def test_sampler_matches_oracle_world_reset():
    pool = synthetic_graph_pool(20, 3, 400)
    opool = [eo.GraphSpec(p.pos.copy(), [int(m) for m in p.one_hop]) for p in pool]
    mk = lambda: np.random.Generator(np.random.PCG64(np.random.SeedSequence(5)))
    env = eo.OracleGraphEnv(20, graph_pool=opool, dynamic_graph=True, np_random=mk())
    s = EpisodeSampler(20, mk(), 3, False)
    s.sample()
    for _ in range(4):
        ep = s.sample()
        assert (ep.origin, ep.interested, ep.graph_index) == (env.origin_agent, env.interested, int(env.selected_graph))
        # first movement of the episode = the one consumed by the forced source step of reset
        g = pool[ep.graph_index]
        moved = g.pos + movement_offsets(ep.movement_seed, 20, 4)[0].T
        np.testing.assert_array_equal(moved, env.pos)
        env.reset()


def test_geometric_graph_rule_and_connectivity():
    g = Graph.from_positions(np.array([[0, 0], [0.2, 0], [0.5, 0.5]]))
    d2 = 0.2 * 0.2
    assert bool(int(g.one_hop[0]) & 2) == (d2 <= 0.2 ** 2)           # <= in float64, like nx.geometric_edges
    assert int(g.one_hop[2]) == 0 and not g.is_connected()
    assert all(p.is_connected() for p in synthetic_graph_pool(20, 4, 0))
    h = Graph.from_edges(4, [(0, 1), (1, 2), (2, 3)])
    assert [int(m) for m in h.one_hop] == [2, 5, 10, 4] and h.is_connected()


def test_pack_episodes_layout():
    pool = synthetic_graph_pool(12, 2, 7)
    s = EpisodeSampler(12, np.random.default_rng(1), 2, False)
    eps = [s.sample() for _ in range(5)]
    p = pack_episodes(eps, pool, 12, 6, dynamic=True)
    assert p["pos"].shape == (5, 12, 2) and p["moves"].shape == (5, 6, 2, 12) and p["one_hop"].dtype == np.uint64
    np.testing.assert_array_equal(p["moves"][3], movement_offsets(eps[3].movement_seed, 12, 6))
    assert np.abs(p["moves"]).max() <= 0.06
    assert pack_episodes(eps, pool, 12, 6, dynamic=False)["moves"].shape == (5, 1, 2, 12)


def test_network_dropin_surface():
    dp = DUEL()
    net = LDGNNetwork(5, 128, 2, 4, 20, dueling_param=dp, backend="torch")
    assert dp[0]["input_dim"] == 1152 and dp[1]["output_dim"] == 1          # ctor mutates the dicts (l_dgn.py:71-84)
    assert set(net.state_dict()) == set(no.init_weights("l_dgn"))
    assert sum(p.numel() for p in net.parameters()) == 1005315
    hl = HLDGNNetwork(5, 128, 2, 4, 20, aggregator="max", dueling_param=DUEL(), backend="torch")
    assert set(hl.state_dict()) == set(no.init_weights("hl_dgn"))
    assert sum(p.numel() for p in hl.parameters()) == 315139
    with pytest.raises(KeyError):
        HLDGNNetwork(5, 128, 2, 4, 20, aggregator="median", dueling_param=DUEL())
    dr = DGNRNetwork(5, 128, 2, 4, 20, dueling_param=DUEL(), backend="torch")
    assert set(dr.state_dict()) == set(no.init_weights("dgn_r")) and sum(p.numel() for p in dr.parameters()) == 1660675
    g = np.load("tests/golden/net_golden_n20.npz")
    dr.load_state_dict(no.init_weights("dgn_r", seed=int(g["weight_seed"]) + 2))
    with torch.enable_grad():
        out, _ = dr(g["obs"])
    np.testing.assert_allclose(out.detach().numpy(), g["dgnr_logits"], atol=1e-5, rtol=0)
    single = LDGNNetwork(5, 128, 2, 4, 20, dueling_param=None, backend="torch")
    assert "out_linear.weight" in single.state_dict() and single.out_linear.in_features == 1152
    with pytest.raises(ValueError, match="Expected obs to be 2D"):
        net(np.zeros(161, np.float32))
    with pytest.raises(ValueError, match="Expected 160 feature cols for nodes, got 161"):
        net(np.zeros((3, 162), np.float32))
    out, state = hl(np.zeros((3, 161), np.float32), state="s")
    assert out.shape == (3, 2) and state == "s"


def test_torch_path_matches_oracle_and_has_gradients():
    g = np.load("tests/golden/net_golden_n12.npz")
    sd = no.init_weights("l_dgn", seed=int(g["weight_seed"]), random_conv_bias=True)
    net = LDGNNetwork(5, 128, 2, 4, 12, dueling_param=DUEL(), backend="auto")     # grad enabled -> torch path on CPU
    net.load_state_dict(sd)
    out, _ = net(g["obs"])
    np.testing.assert_allclose(out.detach().numpy(), g["ldgn_logits"], atol=1e-5, rtol=0)
    out.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU"):
        net(g["obs"])                                       # inference has no CPU fallback


def test_policy_manager_keeps_batch_order():
    net = HLDGNNetwork(5, 64, 2, 2, 12, aggregator="mean", dueling_param=({"hidden_sizes": [64]}, {"hidden_sizes": [64]}),
                       backend="torch")
    policy = DQNPolicy(net, target_update_freq=5)
    assert any(k.startswith("model_old.") for k in policy.state_dict()) and any(k.startswith("model.") for k in policy.state_dict())
    manager = MultiAgentSharedPolicy(policy, agents=[str(i) for i in range(12)])
    obs = np.random.RandomState(0).rand(6, 97).astype(np.float32)
    mask = np.array([[1, 1], [0, 0], [1, 0], [0, 1], [1, 1], [1, 1]], dtype=bool)
    batch = {"obs": {"agent_id": np.array(list("315903")), "obs": obs, "mask": mask}}
    out = manager(batch)
    logits = out.out.logits.detach()
    want = no.dqn_act(logits, mask)
    assert out.act.tolist() == want.tolist()
    policy.set_eps(1.0)
    np.random.seed(0)
    act = manager.exploration_noise(np.zeros(6, dtype=np.int64), batch)
    np.random.seed(0)
    rand_mask = np.random.rand(6) < 1.0
    q = np.random.rand(6, 2) + mask
    assert act.tolist() == q.argmax(1).tolist() and rand_mask.all()
