"""Host-side logic that needs no GPU: episode sampling mirrors World.reset's RNG protocol, pool packing,
graph helpers, network drop-in surface (ctor / state_dict keys / errors / torch path), policy helpers."""
import numpy as np
import pytest
import torch

from melissa_amd.env.episodes import (EpisodeSampler, Graph, movement_offsets, pack_episodes,
                                      synthetic_graph_pool)
from melissa_amd.env.episodes import int_to_set, set_to_int
from melissa_amd.networks import DGNRNetwork, HLDGNNetwork, LDGNNetwork
from melissa_amd.policy import DQNPolicy, MultiAgentSharedPolicy
from oracle import env_oracle as eo
from oracle import net_oracle as no

DUEL = lambda: ({"hidden_sizes": [128, 128]}, {"hidden_sizes": [128, 128]})


@pytest.mark.parametrize("testing", [False, True])
def test_sampler_matches_oracle_world_reset(testing):
    """training mode (core.py:371-395) and the evaluation schedule (is_testing, core.py:348-370; the oracle is
    pinned to the real reference in both by the golden traces)."""
    pool = synthetic_graph_pool(20, 3, 400)
    opool = [eo.GraphSpec(p.pos.copy(), [int(m) for m in p.one_hop]) for p in pool]
    mk = lambda: np.random.Generator(np.random.PCG64(np.random.SeedSequence(5)))
    kw = dict(is_testing=True, num_test_episodes=3) if testing else {}
    env = eo.OracleGraphEnv(20, graph_pool=opool, dynamic_graph=True, np_random=mk(), **kw)
    s = EpisodeSampler(20, mk(), 3, False, **kw)
    s.sample()
    for _ in range(7 if testing else 4):          # testing: walks the 3-seed list more than twice (wrap-around)
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


# ---------------------------------------------------------------------------------------------------
# DGN-R learn path (policies/dgn.py) and the collective collector's sibling bookkeeping
# ---------------------------------------------------------------------------------------------------
def _filled_round_replay(n_envs=3, n=6, cap=5, rounds=7, seed=0):
    """A RoundReplay filled on the CPU the way mel_env_round fills it (ring per env, cursor counts writes)."""
    from melissa_amd.replay import RoundReplay
    rng = np.random.RandomState(seed)
    rp = RoundReplay(n_envs, n, cap, "cpu")
    for r in range(rounds):
        for e in range(n_envs):
            k = r % cap
            acted = rng.choice(n, size=rng.randint(1, n), replace=False)
            m = 0
            for a in acted:
                m |= 1 << int(a)
            words = torch.from_numpy(np.atleast_1d(int_to_set(m, n)).view(np.int64))       # one word up to 64 nodes, two beyond
            rp.acted[e, k] = words if n > 64 else words[0]
            rp.done[e, k] = (words if n > 64 else words[0]) if r % 4 == 3 else 0
            rp.obs[e, k] = torch.from_numpy(rng.uniform(0, 1, 8 * n).astype(np.float32))
            rp.obs_next[e, k] = torch.from_numpy(rng.uniform(0, 1, 8 * n).astype(np.float32))
            rp.act[e, k] = torch.from_numpy(rng.randint(0, 2, n).astype(np.int8))
            rp.rew[e, k] = torch.from_numpy(rng.uniform(-1, 1, n).astype(np.float32))
            rp.episode[e, k] = r // 4
            rp.cursor[e] = r + 1
    return rp


@pytest.mark.parametrize("n", [6, 100])
def test_export_transitions_sibling_indices(n):
    """collective_experience_collector.py:70-80: ``indices[j]`` of an experience = buffer index of agent j's
    transition of the SAME env round, -1 if agent j did not act; buffer_id = env * N + agent (:251-256).
    (n = 100: the acted / done sets are two 64-bit words.)"""
    rp = _filled_round_replay(n=n)
    ex = rp.export_transitions()
    T = len(ex["act"])
    assert T == len(rp) and ex["indices"].shape == (T, rp.n)
    key = list(zip(ex["env_id"].tolist(), ex["record_slot"].tolist()))
    for t in range(T):
        e, k, i = int(ex["env_id"][t]), int(ex["record_slot"][t]), int(ex["agent_id"][t])
        acted = set_to_int(rp.acted[e, k].numpy())
        assert ex["buffer_id"][t] == e * rp.n + i and (acted >> i) & 1
        for j in range(rp.n):
            s = int(ex["indices"][t, j])
            if (acted >> j) & 1:
                assert key[s] == (e, k) and ex["agent_id"][s] == j
            else:
                assert s == -1
        assert ex["indices"][t, i] == t                          # an experience is its own sibling (dgn.py sums it too)
        np.testing.assert_array_equal(ex["obs"][t, :-1], rp.obs[e, k].numpy())
        assert ex["obs"][t, -1] == i and ex["act"][t] == int(rp.act[e, k, i])
        assert ex["rew_agent"][t] == float(rp.rew[e, k, i]) and bool(ex["done"][t]) == bool((set_to_int(rp.done[e, k].numpy()) >> i) & 1)
    # oldest first inside an env: record order follows write order
    for e in range(rp.B):
        eps = ex["episode"][ex["env_id"] == e]
        assert (np.diff(eps) >= 0).all()


def test_dgn_learn_equals_the_reference_loop():
    """One batched forward + segment sum == the per-experience loop of policies/dgn.py:31-64 (same loss, same
    gradients), driven through the reference's own data layout (info.indices + active_obs.index)."""
    from melissa_amd.policy import DGNPolicy
    n = 6
    rp = _filled_round_replay(n=n)
    ex = rp.export_transitions()
    rng = np.random.RandomState(1)
    pick = rng.choice(len(ex["act"]), size=8, replace=False)
    returns = rng.uniform(-1, 1, size=8).astype(np.float32)
    # reference layout: batch.active_obs = the buffer rows named by the valid sibling indices (with repeats)
    indices = ex["indices"][pick]
    active_index = indices[indices >= 0]
    active_obs, active_act = ex["obs"][active_index], ex["act"][active_index]
    gather, segment = DGNPolicy.segments_from_indices(indices, active_index)

    def make():
        torch.manual_seed(4)
        net = DGNRNetwork(5, 32, 2, 2, n, dueling_param=({"hidden_sizes": [64]}, {"hidden_sizes": [64]}), device="cpu",
                          backend="torch")
        return net

    # (a) the reference's loop
    net_a = make()
    batch_q = []
    for i in range(len(pick)):
        rows = [int(np.where(active_index == idx)[0][0]) for idx in indices[i][indices[i] >= 0]]
        q = net_a(torch.from_numpy(active_obs[rows]))[0]
        batch_q.append(q[torch.arange(len(rows)), torch.from_numpy(active_act[rows])].sum())
    loss_a = (torch.from_numpy(returns) - torch.stack(batch_q)).pow(2).mean()
    loss_a.backward()
    # (b) DGNPolicy.learn (lr = 0: only the gradients matter)
    net_b = make()
    pol = DGNPolicy(net_b, torch.optim.SGD(net_b.parameters(), lr=0.0))
    out = pol.learn(dict(active_obs=torch.from_numpy(active_obs[gather]), active_act=active_act[gather],
                         segment=segment, returns=returns))
    assert abs(out["loss"] - float(loss_a.detach())) < 1e-6
    for (name, pa), pb in zip(net_a.named_parameters(), net_b.parameters()):
        if pa.grad is None:
            assert pb.grad is None or float(pb.grad.abs().max()) == 0.0, name      # lin_skip: unused (dgn_r.py)
            continue
        torch.testing.assert_close(pb.grad, pa.grad, atol=1e-6, rtol=1e-5)
    # huber variant (clip_loss_grad, dgn.py:59-62)
    pol2 = DGNPolicy(make(), None, clip_loss_grad=True)
    pol2.optim = torch.optim.SGD(pol2.model.parameters(), lr=0.0)
    out2 = pol2.learn(dict(active_obs=torch.from_numpy(active_obs[gather]), active_act=active_act[gather],
                           segment=segment, returns=returns))
    want = torch.nn.functional.huber_loss(torch.stack(batch_q).detach().reshape(-1, 1), torch.from_numpy(returns).reshape(-1, 1))
    assert abs(out2["loss"] - float(want)) < 1e-6


@pytest.mark.parametrize("n", [6, 70])
def test_sample_collective_siblings_and_dgn_learner_step(n):
    from melissa_amd.env.episodes import sets_to_bool
    from melissa_amd.policy import DGNPolicy
    from melissa_amd.replay import DGNLearner
    rp = _filled_round_replay(n=n)
    g = torch.Generator().manual_seed(3)
    b = rp.sample_collective(16, n_step=2, gamma=0.9, generator=g)
    bits = torch.from_numpy(sets_to_bool(b["sibling_mask"].numpy(), n))           # [16, n] (one- or two-word sets)
    assert b["segment"].numel() == int(bits.sum()) and (torch.bincount(b["segment"], minlength=16) == bits.sum(1)).all()
    for r in range(b["segment"].numel()):
        i = int(b["segment"][r])
        e, k, j = int(b["env"][i]), int(b["slot"][i]), int(b["active_obs"][r, -1])
        assert bits[i, j] and int(b["active_act"][r]) == int(rp.act[e, k, j])
        assert torch.equal(b["active_obs"][r, :-1], rp.obs[e, k])
    assert all(bool(bits[i, int(b["agent"][i])]) for i in range(16))             # the experience itself is a sibling
    torch.manual_seed(0)
    net = DGNRNetwork(5, 32, 2, 2, n, dueling_param=({"hidden_sizes": [64]}, {"hidden_sizes": [64]}), device="cpu",
                      backend="torch")
    pol = DGNPolicy(net, torch.optim.Adam(net.parameters(), lr=1e-3), target_update_freq=2)
    learner = DGNLearner(pol, rp, batch_size=8, n_step=2, gamma=0.9, seed=1)
    before = [p.detach().clone() for p in net.parameters()]
    losses = [learner.step()["loss"] for _ in range(3)]
    assert all(np.isfinite(losses))
    assert any(not torch.equal(a, p.detach()) for a, p in zip(before, net.parameters()))


@pytest.mark.parametrize("model,n", [("dgn_r", 6), ("l_dgn", 9), ("dgn_r", 70)])
def test_dgn_dense_form_equals_the_row_form(model, n):
    """The learn path's dense form - network body once per sampled experience, head once per (experience, node), siblings as a
    [B, N] mask (GraphQNetwork.torch_forward_all_agents, DGNPolicy.loss_backward) - against the reference's row form (one
    observation row per sibling, policies/dgn.py:31-55): same Q values, same loss, same gradients."""
    from melissa_amd.networks import LDGNNetwork
    from melissa_amd.policy import DGNPolicy
    from melissa_amd.replay import DGNLearner
    rp = _filled_round_replay(n=n)

    def make():
        torch.manual_seed(4)
        cls = DGNRNetwork if model == "dgn_r" else LDGNNetwork
        return cls(5, 32, 2, 2, n, dueling_param=({"hidden_sizes": [64]}, {"hidden_sizes": [64]}), device="cpu", backend="torch")

    net_a, net_b = make(), make()
    pol_a = DGNPolicy(net_a, torch.optim.SGD(net_a.parameters(), lr=0.0))
    pol_b = DGNPolicy(net_b, torch.optim.SGD(net_b.parameters(), lr=0.0))
    learner = DGNLearner(pol_a, rp, batch_size=8, n_step=2, gamma=0.9, seed=1)
    dense = learner.sample_batch()
    rows = learner.row_form(dense)
    assert rows["segment"].numel() == int(dense["sibling"].sum()) and dense["obs_matrix"].shape == (8, 8 * n)
    # Q of every (experience, node) equals the per-row forward on [obs_matrix | node]
    with torch.no_grad():
        q_all = net_a.torch_forward_all_agents(dense["obs_matrix"])
        per_row = net_a.torch_forward(rows["active_obs"])
    seg, agent = rows["segment"], rows["active_obs"][:, -1].long()
    torch.testing.assert_close(q_all[seg, agent], per_row, atol=1e-6, rtol=1e-5)
    la = pol_a.loss_backward({k: dense[k] for k in ("obs_matrix", "act_all", "sibling", "returns")})
    lb = pol_b.loss_backward(dict(rows))
    assert abs(float(la) - float(lb)) <= 1e-6 * max(1.0, abs(float(lb)))
    for (name, pa), pb in zip(net_a.named_parameters(), net_b.parameters()):
        if pb.grad is None:
            assert pa.grad is None or float(pa.grad.abs().max()) == 0.0, name
            continue
        torch.testing.assert_close(pa.grad, pb.grad, atol=2e-6, rtol=1e-4)


def test_construct_time_samplings_line_up_with_the_wrapped_oracle():
    """PettingZooEnv(GraphEnv(...)) samples THREE episodes while being constructed (core.py:190, graph.py:118, [3P]
    tianshou PettingZooEnv.__init__ -> env.reset()); HipGraphVectorEnv(construct_like_reference=True) replays that
    many, so its first reset() yields the sampler's 4th episode - what __graft_entry__.smoke() relies on."""
    n = 20
    pool = synthetic_graph_pool(n, 3, 10)
    opool = [eo.GraphSpec(g.pos.copy(), [set_to_int(m) for m in g.one_hop]) for g in pool]
    mk = lambda: np.random.Generator(np.random.PCG64(np.random.SeedSequence(7)))
    sampler = EpisodeSampler(n, mk(), 3, False)
    episodes = [sampler.sample() for _ in range(5)]
    wrapped = eo.OraclePettingZooEnv(eo.OracleGraphEnv(n, graph_pool=opool, dynamic_graph=True, np_random=mk()))
    assert wrapped.env.origin_agent == episodes[2].origin and wrapped.env.interested == episodes[2].interested
    wrapped.reset()
    assert wrapped.env.origin_agent == episodes[3].origin and wrapped.env.interested == episodes[3].interested


def test_bench_cpu_baseline_variants_run():
    """bench.py's CPU baseline legs (the oracle on the host cores): single process and env-worker processes."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    one = bench.cpu_baseline(12, envs=4, warm=1, reps=3, rep_seconds=0.2)
    assert one["kind"] == "port" and one["value"] > 0 and one["value_bs1"] > 0 and one["cores"] >= 1
    assert one["min"] <= one["median"] <= one["max"] and one["repetitions"] == 3 and one["cpu_model"]
    sub = bench.cpu_baseline_subproc(12, budget_s=1.0, workers=2, envs_per_worker=2)
    assert sub["value"] > 0 and sub["envs"] == 4
    fl = bench.stage_flops((10.0, 20.0, 5.0))
    assert fl["conv2_lin"] == 2.0 * 15 * 512 * 512 and fl["conv1_lin"] == 2.0 * 30 * 512 * 128
    ft = bench.stage_flops((10.0, 20.0, 5.0), table_rows=2000)                  # node-feature table: priced at its rows
    assert ft["conv1_lin"] == 2.0 * 4000 * 512 * 128 and ft["encoder"] == 2000 * 34048 and ft["conv2_lin"] == fl["conv2_lin"]


def test_graph_size_limits_and_two_word_node_sets():
    """The reference CLI offers --n-agents 20 / 50 / 100 (common.py:49): all three are accepted (beyond 64 nodes a node set is
    two 64-bit words); beyond 128 nodes the constructors refuse up front with a clear message."""
    from melissa_amd.env.episodes import Graph, int_to_set, pack_episodes, set_to_int, sets_to_bool, Episode
    with pytest.raises(ValueError, match="outside \\[1, 128\\]"):
        LDGNNetwork(5, 128, 2, 4, 129, dueling_param=DUEL(), backend="torch")
    from melissa_amd.env import HipGraphVectorEnv
    with pytest.raises(ValueError, match="outside \\[1, 128\\]"):
        HipGraphVectorEnv(2, 200, graph_pool=synthetic_graph_pool(12, 1, 0), device="cpu")
    LDGNNetwork(5, 128, 2, 4, 100, dueling_param=DUEL(), backend="torch")
    HLDGNNetwork(5, 128, 2, 4, 65, aggregator="max", dueling_param=DUEL(), backend="torch")
    # packing: node sets of a 100-node graph are [.., 2] words, low word first; up to 64 nodes nothing changes shape
    g = synthetic_graph_pool(100, 1, 0)[0]
    assert g.one_hop.shape == (100, 2) and g.one_hop.dtype == np.uint64 and g.is_connected()
    member = sets_to_bool(g.one_hop, 100)
    assert member.shape == (100, 100) and (member == member.T).all() and not member.diagonal().any()
    dx = g.pos[:, None, :] - g.pos[None, :, :]
    want = ((dx ** 2).sum(-1) <= 0.2 ** 2) & ~np.eye(100, dtype=bool)
    np.testing.assert_array_equal(member, want)
    m = (1 << 99) | (1 << 64) | (1 << 63) | 5
    assert set_to_int(int_to_set(m, 100)) == m and int_to_set(m, 100).tolist() == [(1 << 63) | 5, (1 << 35) | 1]
    assert int_to_set(5, 50) == np.uint64(5) and set_to_int(np.uint64(5)) == 5
    p = pack_episodes([Episode(0, 77, m, 1, 0)], [g], 100, 3, True)
    assert p["one_hop"].shape == (1, 100, 2) and p["interested"].shape == (1, 2) and p["scripted"].shape == (1, 2)
    assert set_to_int(p["interested"][0]) == m and p["moves"].shape == (1, 3, 2, 100)
    assert synthetic_graph_pool(50, 1, 0)[0].one_hop.shape == (50,)


def test_packed_graph_pool_round_trip(tmp_path):
    """The on-disk / in-HBM graph dataset (pos f64 [G, N, 2] + adj u64 [G, N]; replaces graph_topologies/*.pickle,
    core.py:165-175,450-452): same graphs as the list generator, lossless save / load, cached second call."""
    from melissa_amd.env import cached_graph_pool, load_graph_pool, packed_graph_pool, save_graph_pool
    pos, hop, seeds = packed_graph_pool(20, 30, first_seed=0)
    ref = synthetic_graph_pool(20, 30, 0)
    assert pos.shape == (30, 20, 2) and hop.dtype == np.uint64 and len(seeds) == 30 and seeds[0] >= 0
    for g in range(30):
        np.testing.assert_array_equal(pos[g], ref[g].pos)
        np.testing.assert_array_equal(hop[g], ref[g].one_hop)
    path = str(tmp_path / "pool.npz")
    save_graph_pool(path, pos, hop, seeds)
    back = load_graph_pool(path)
    assert len(back) == 30 and all(b.is_connected() for b in back)
    np.testing.assert_array_equal(back[7].pos, ref[7].pos)
    a = cached_graph_pool(20, 12, 0, cache_dir=str(tmp_path))
    b = cached_graph_pool(20, 12, 0, cache_dir=str(tmp_path))          # second call reads the .npz
    assert len(a) == len(b) == 12 and all(np.array_equal(x.one_hop, y.one_hop) for x, y in zip(a, b))


def test_captured_update_refuses_what_it_cannot_capture():
    """DQNLearner.capture() (HIP-graph replay of the update) needs a GPU replay and an optimizer with a capturable mode; the
    flat-gradient reducer exposes the three phases a captured update replays / issues separately."""
    import torch
    from melissa_amd import parallel
    from melissa_amd.networks import LDGNNetwork
    from melissa_amd.policy import DQNPolicy
    from melissa_amd.replay import DQNLearner, RoundReplay
    net = LDGNNetwork(5, 128, 2, 4, 12, dueling_param=({"hidden_sizes": [128, 128]}, {"hidden_sizes": [128, 128]}), device="cpu",
                      backend="torch")
    policy = DQNPolicy(net, torch.optim.Adam(net.parameters(), lr=1e-3))
    learner = DQNLearner(policy, RoundReplay(4, 12, 8, "cpu"))
    with pytest.raises(ValueError, match="ROCm / CUDA device"):
        learner.capture()
    assert learner.captured is None
    red = parallel.FlatGradAllReducer(net)
    assert not red.active()                                        # no process group: __call__ is a no-op
    for p in net.parameters():
        p.grad = torch.full_like(p, 2.0)
    red.pack()
    assert float(red.flat.min()) == 2.0 == float(red.flat.max())
    red.flat.mul_(0.5)
    red.unpack()
    assert all(float(p.grad.min()) == 1.0 == float(p.grad.max()) for p in net.parameters())


def test_collect_result_has_the_reference_field_set():
    """The names the reference's collectors return (collector.py:14-36 ``DictOfSequenceSummaryStats`` / ``CollectStatsWithInfo``
    on top of [3P] tianshou 1.0.0 ``CollectStats``; filled at multi_agent_collector.py:341-353) and the constructor / call
    keywords of ``MultiAgentCollector`` (multi_agent_collector.py:31-42, 89-97; call sites l_dgn.py:119-127, 185-201)."""
    import dataclasses
    import inspect
    from melissa_amd import _lib
    from melissa_amd.collect import (CollectStatsWithInfo, DictOfSequenceSummaryStats, MultiAgentCollector, SequenceSummaryStats,
                                     result_from_episode_log)
    reference_fields = ["n_collected_episodes", "n_collected_steps", "collect_time", "collect_speed", "returns", "returns_stat",
                        "lens", "lens_stat", "info"]                       # multi_agent_collector.py:341-353, in that order
    names = [f.name for f in dataclasses.fields(CollectStatsWithInfo)]
    assert names[:len(reference_fields)] == reference_fields
    assert [f.name for f in dataclasses.fields(SequenceSummaryStats)] == ["mean", "std", "max", "min"]       # [3P] tianshou 1.0.0
    assert [f.name for f in dataclasses.fields(DictOfSequenceSummaryStats)] == ["stats"]                     # collector.py:17
    ctor = inspect.signature(MultiAgentCollector.__init__).parameters
    assert list(ctor)[:6] == ["self", "agents_num", "policy", "env", "buffer", "exploration_noise"]
    assert ctor["buffer"].default is None and ctor["exploration_noise"].default is False
    call = inspect.signature(MultiAgentCollector.collect).parameters
    assert list(call)[:7] == ["self", "n_step", "n_episode", "random", "render", "no_grad", "gym_render_kwargs"]
    assert call["random"].default is False and call["no_grad"].default is False
    # two finished episodes, as the device log hands them over
    stats = np.zeros((2, len(_lib.LOGGER_KEYS)))
    stats[:, _lib.LOGGER_KEYS.index("episode_rewards_sum")] = [1.5, -0.5]
    stats[:, _lib.LOGGER_KEYS.index("coverage")] = [0.5, 1.0]
    meta = np.array([[0, 3, 7], [1, 4, 9]], dtype=np.int32)
    res = result_from_episode_log(stats, meta, total=2, steps=40, dt=2.0)
    assert res.n_collected_episodes == 2 and res.n_collected_steps == 40 and res.collect_speed == 20.0
    assert res.returns.tolist() == [1.5, -0.5] and res.lens.tolist() == [7, 9]
    assert res.returns_stat == SequenceSummaryStats(mean=0.5, std=1.0, max=1.5, min=-0.5) and res.lens_stat.mean == 8.0
    assert res.info.stats["coverage"].mean == 0.75 and set(res.info.stats) == set(_lib.LOGGER_KEYS)
    assert res["n/ep"] == 2 and res["n/st"] == 40 and res["coverage"] == 0.75 and res["len"] == 8.0 and "nope" not in res
    empty = result_from_episode_log(stats[:0], meta[:0], total=0, steps=5, dt=1.0)
    assert empty.returns_stat is None and empty.lens_stat is None and len(empty.returns) == 0 and empty.info.stats == {}


def test_fused_relu_mlp_and_optimizer_wrapper_on_host_tensors():
    """Host-side halves of two round-3 learn-path changes: ``networks.common.mlp`` folds a Linear + ReLU pair (and the callers'
    trailing ``F.relu``) into one call per layer - same values as the module itself -, and ``optim.adam_step`` leaves anything its
    kernel does not cover (here: CPU parameters) to ``torch.optim.Adam.step`` unchanged."""
    import copy
    import torch.nn.functional as F
    from melissa_amd.networks.common import MLP, mlp
    from melissa_amd.optim import adam_step
    torch.manual_seed(1)
    module = MLP(5, 128, [128])
    x = torch.randn(37, 5)
    np.testing.assert_array_equal(mlp(module, x, hip=False).detach().numpy(), module(x).detach().numpy())
    np.testing.assert_array_equal(mlp(module, x, hip=False, final_relu=True).detach().numpy(), F.relu(module(x)).detach().numpy())
    twin = copy.deepcopy(module)
    a, b = torch.optim.Adam(module.parameters(), lr=1e-2), torch.optim.Adam(twin.parameters(), lr=1e-2)
    for _ in range(3):
        for m, o in ((module, a), (twin, b)):
            o.zero_grad()
            m(x).pow(2).mean().backward()
        adam_step(a)
        b.step()
    for p, q in zip(module.parameters(), twin.parameters()):
        assert torch.equal(p, q)
