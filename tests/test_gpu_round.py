"""GPU parity of the round-batched path: mel_ldgn_forward_agents rows equal the per-row forward / the
oracle, and mel_env_round leaves every env in exactly the state the oracle reaches by replaying the same
actions one AEC step at a time."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.trace_replay import set_int, set_ints      # node sets in any form (uint64 scalar / words beyond 64 nodes) -> int


def agent_masks(bs, n):
    """Zeroed agent-set array in the ABI's layout: uint64 [bs] up to 64 nodes, [bs, 2] words beyond."""
    return np.zeros((bs,) if n <= 64 else (bs, (n + 63) // 64), dtype=np.uint64)


def add_agent(masks, b, a):
    if masks.ndim == 1:
        masks[b] |= np.uint64(1) << np.uint64(a)
    else:
        masks[b, a // 64] |= np.uint64(1) << np.uint64(a % 64)
TOL = 1e-4
DUEL = lambda: ({"hidden_sizes": [128, 128]}, {"hidden_sizes": [128, 128]})


def make_ldgn(n, seed=9, model="l_dgn"):
    from melissa_amd.networks import DGNRNetwork, LDGNNetwork
    from oracle import net_oracle as no
    sd = no.init_weights(model, seed=seed, random_conv_bias=True)
    cls = LDGNNetwork if model == "l_dgn" else DGNRNetwork
    net = cls(5, 128, 2, 4, n, dueling_param=DUEL(), device="cuda", backend="hip")
    net.load_state_dict(sd)
    return net, sd


def random_obs_matrix(rng, bs, n):
    m = np.zeros((bs, n, 8), dtype=np.float32)
    m[:, :, 0:2] = rng.uniform(0, 1, size=(bs, n, 2))
    m[:, :, 2] = rng.randint(0, 9, size=(bs, n))
    m[:, :, 3] = rng.randint(0, 4, size=(bs, n))
    m[:, :, 4:7] = rng.randint(0, 2, size=(bs, n, 3))
    m[:, :, 7] = (rng.uniform(size=(bs, n)) > 0.1)
    return m


@pytest.mark.parametrize("model", ["l_dgn", "dgn_r"])
@pytest.mark.parametrize("n,bs", [(20, 64), (50, 40), (64, 9), (5, 7), (100, 12), (65, 5), (128, 3)])
def test_forward_agents_rows_match_oracle_and_per_row_forward(n, bs, model):
    from oracle import net_oracle as no
    rng = np.random.RandomState(n * 100 + bs)
    mat = random_obs_matrix(rng, bs, n)
    masks = agent_masks(bs, n)
    rows = []
    for b in range(bs):
        k = rng.randint(0, min(n, 12) + 1)                     # 0..12 agents, some envs with none
        for a in sorted(rng.choice(n, size=k, replace=False)):
            add_agent(masks, b, int(a))
            rows.append((b, int(a)))
    net, sd = make_ldgn(n, model=model)
    cap = bs * n
    obs_matrix = torch.from_numpy(mat.reshape(bs, n * 8)).cuda()
    with torch.no_grad():
        logits, offsets = net.hip_forward_agents(obs_matrix, torch.from_numpy(masks.view(np.int64)).cuda(), cap)
    offsets = offsets.cpu().numpy()
    assert offsets[-1] == len(rows)
    assert list(offsets[:-1]) == list(np.concatenate([[0], np.cumsum([bin(set_int(m)).count("1") for m in masks])])[:-1])
    # the same (env, agent) rows as explicit observation rows with the index column
    obs_rows = np.concatenate([mat.reshape(bs, -1)[[b for b, _ in rows]],
                               np.array([[a] for _, a in rows], dtype=np.float32)], axis=1)
    with torch.no_grad():
        per_row = net.hip_forward(torch.from_numpy(obs_rows).cuda()).cpu().numpy()
        want = (no.ldgn_forward if model == "l_dgn" else no.dgnr_forward)(sd, obs_rows, n).numpy()
    got = logits[:len(rows)].cpu().numpy()
    np.testing.assert_allclose(got, want, atol=TOL, rtol=0)
    np.testing.assert_allclose(got, per_row, atol=2e-5, rtol=0)
    totals = net.hip_tap(2, bs, cap).cpu().numpy()
    assert totals[2] == len(rows) and totals[0] <= totals[1] <= bs * n


def oracle_round(pz, act_of_agent):
    """Replay one env round on the oracle exactly like mel_env_round: dead steps, then each active agent.
    Returns what the round's world step produced (None if no agent acted): obs_matrix, rewards, terminated."""
    n = pz.env.n
    outcome = None
    for _ in range(3 * n + 4):
        env = pz.env
        sel = env.agent_selection
        dead = (env.terminated >> sel) & 1
        moves = env.num_moves
        obs, rew, term, trunc, info = pz.step(0 if dead else int(act_of_agent[sel]))
        if pz.env.num_moves != moves:
            # the world step ran (environment_step stays False when the observation right after it raises
            # explicit_reset, graph.py:205-211, so the move counter is the reliable signal)
            outcome = dict(obs_next=pz.env.obs_matrix.copy(), rew=list(pz.env.rewards),
                           terminated=pz.env.terminated)
        if term:
            pz.done_count += 1
            if info.get("explicit_reset") or pz.done_count == n:
                if not hasattr(pz, "finished"):
                    pz.finished = []
                pz.finished.append((dict(info["logger_stats"]), pz.env.num_moves))     # what the collectors log per episode
                pz.reset()
                pz.done_count = 0
                return outcome
        if info.get("environment_step"):
            return outcome
    raise AssertionError("round did not terminate")


@pytest.mark.parametrize("n,dynamic,supply", [(20, True, "table"), (12, False, "table"), (50, True, "table"),
                                               (50, False, "table"), (20, True, "stream"), (12, False, "stream"),
                                               (50, True, "stream"), (100, True, "table"), (100, True, "stream"),
                                               (70, False, "stream")])
def test_round_loop_matches_oracle(n, dynamic, supply):
    """``supply`` "stream": the episodes come from the device sampler through a 5-slot ring (3 slots at N = 50; refilled every 2 rounds - every round -
    on a side stream), so every env's k-th reset - far beyond the first ring - must equal the oracle env's k-th reset."""
    round_loop_vs_oracle(n, dynamic, supply)


def test_round_forward_at_the_benchmark_batch_matches_oracle():
    """1024 envs of 50 nodes - the benchmark's own batch (BASELINE.json configs[2], bench.py's default): from a few thousand agent
    rows per round on, the default precision (MEL_PREC_F32_AUTO) sends conv2's projections and the heads' first layer to the
    128 x 128 split-bf16 kernels - the launches the benchmark's step spends most of its time in.  Every logit of those rounds
    against the oracle (<= 1e-4), every chosen action against the oracle's argmax wherever its margin exceeds the tolerance.
    (The env half does not depend on the batch size: the per-env comparisons of test_round_loop_matches_oracle cover it.)"""
    from melissa_amd.collect import RoundLoop
    from melissa_amd.env import HipGraphVectorEnv, synthetic_graph_pool
    from melissa_amd.policy import DQNPolicy
    from oracle import net_oracle as no
    n, B = 50, 1024
    venv = HipGraphVectorEnv(B, n, graph_pool=synthetic_graph_pool(n, 16, first_seed=50), dynamic_graph=True, device="cuda",
                             max_moves=48, seed=123, construct_like_reference=False)
    net, sd = make_ldgn(n)
    assert net.feature_dtype == "f32a"
    loop = RoundLoop(venv, DQNPolicy(net), eps=0.0, seed=5)
    torch.set_num_threads(16)
    big_rounds = 0
    for it in range(12):
        live = loop.live.cpu().numpy().view(np.uint64).reshape(B, -1)[:, 0].copy()
        mat = venv.obs_matrix().cpu().numpy().copy()
        loop.step()
        torch.cuda.synchronize()
        envs, agents = np.nonzero((live[:, None] >> np.arange(n, dtype=np.uint64)[None, :]) & np.uint64(1))
        assert int(loop.offsets[-1]) == len(envs)
        if len(envs) < 3072 or big_rounds >= 3:      # (ceil(U1 / 128) + ceil(R / 128)) * 4 >= 192 big tiles needs U1 >= R >= 3072
            continue
        big_rounds += 1
        obs_rows = np.concatenate([mat[envs], agents[:, None].astype(np.float32)], axis=1)
        want = no.ldgn_forward(sd, obs_rows, n).numpy()
        got = loop.logits[:len(envs)].cpu().numpy()
        np.testing.assert_allclose(got, want, atol=TOL, rtol=0)
        act = loop.act[:len(envs)].cpu().numpy()
        srt = np.sort(want, axis=1)
        clear = srt[:, -1] - srt[:, -2] > 2 * TOL
        np.testing.assert_array_equal(act[clear], want.argmax(axis=1)[clear])
        print(f"round {it}: {len(envs)} agent rows, max |logit error| {np.abs(got - want).max():.2e}")
    assert big_rounds >= 2 and loop.counters()["errors"] == 0


def round_loop_vs_oracle(n, dynamic, supply, B=6, K=40):
    from melissa_amd import _lib as L
    from melissa_amd.collect import RoundLoop, sample_episode_table
    from melissa_amd.env import HipGraphVectorEnv, synthetic_graph_pool
    from melissa_amd.policy import DQNPolicy
    from oracle import env_oracle as eo
    from oracle import net_oracle as no
    seed = 77
    graphs = synthetic_graph_pool(n, 3, first_seed=50)
    venv = HipGraphVectorEnv(B, n, graph_pool=graphs, dynamic_graph=dynamic, device="cuda", max_moves=48,
                             construct_like_reference=False)
    venv.enable_episode_log(256)
    net, sd = make_ldgn(n)
    # episodes: the oracle env consumes two samplings while it is constructed, so the device starts at #1
    packed, table = sample_episode_table(venv, 14, seed)
    from melissa_amd.replay import RoundReplay
    replay = RoundReplay(B, n, 8, "cuda")
    if supply == "stream":
        # (beyond 20 nodes the greedy random-weight policy plays ~12-round episodes, so a 3-slot ring refilled EVERY round)
        ring = 5 if n <= 20 else 3
        loop = RoundLoop(venv, DQNPolicy(net), eps=0.0, seed=seed, replay=replay, ring=ring, discard=1)
        K = 70 if n <= 50 else 110
    else:
        loop = RoundLoop(venv, DQNPolicy(net), eps=0.0, seed=seed, replay=replay,
                         episodes=({k: v for k, v in packed.items()}, np.ascontiguousarray(table[:, 1:])))
    refs = []
    for b in range(B):
        env = eo.OracleGraphEnv(n, graph_pool=[eo.GraphSpec(g.pos.copy(), set_ints(g.one_hop)) for g in graphs],
                                dynamic_graph=dynamic,
                                np_random=np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed + b))))
        pz = eo.OraclePettingZooEnv.__new__(eo.OraclePettingZooEnv)
        pz.env, pz.n, pz.rewards, pz.done_count = env, n, [0] * n, 0
        env.last()                                    # the observe after reset (clears is_new_round like the device)
        refs.append(pz)
    sets = lambda: venv.node_sets().cpu().numpy().view(np.uint64)
    checked_rows = recorded_rounds = 0
    for it in range(K):
        live = loop.live.cpu().numpy().view(np.uint64).copy()
        for b, pz in enumerate(refs):
            assert set_int(live[b]) == pz.env.sel_active, (it, b)
        mat = venv.obs_matrix().cpu().numpy().copy()
        loop.step()
        torch.cuda.synchronize()
        offsets = loop.offsets.cpu().numpy()
        act = loop.act.cpu().numpy()
        logits = loop.logits.cpu().numpy()
        rows = [(b, a) for b in range(B) for a in range(n) if (set_int(live[b]) >> a) & 1]
        assert offsets[-1] == len(rows)
        if rows:
            obs_rows = np.concatenate([mat[[b for b, _ in rows]], np.array([[a] for _, a in rows], np.float32)], axis=1)
            want = no.ldgn_forward(sd, obs_rows, n).numpy()
            np.testing.assert_allclose(logits[:len(rows)], want, atol=TOL, rtol=0)
            checked_rows += len(rows)
        cursor = replay.cursor.cpu().numpy()
        for b, pz in enumerate(refs):
            np.testing.assert_array_equal(mat[b].reshape(n, 8), pz.env.obs_matrix)
            acts = {a: act[offsets[b] + k] for k, a in enumerate(a for a in range(n) if (set_int(live[b]) >> a) & 1)}
            ep_before = pz.env_episode = getattr(pz, "env_episode", 0)
            outcome = oracle_round(pz, acts)
            if set_int(live[b]):                               # the replay record of this round
                slot = (int(cursor[b]) - 1) % replay.K
                assert outcome is not None
                np.testing.assert_array_equal(replay.obs[b, slot].cpu().numpy(), mat[b])
                np.testing.assert_array_equal(replay.obs_next[b, slot].cpu().numpy(), outcome["obs_next"].reshape(-1))
                assert set_int(replay.acted[b, slot].cpu().numpy().view(np.uint64)) == set_int(live[b])
                assert set_int(replay.done[b, slot].cpu().numpy().view(np.uint64)) == outcome["terminated"] & set_int(live[b])
                rec_act, rec_rew = replay.act[b, slot].cpu().numpy(), replay.rew[b, slot].cpu().numpy()
                for a_id, a_val in acts.items():
                    assert rec_act[a_id] == a_val
                    assert rec_rew[a_id] == np.float32(outcome["rew"][a_id])
                recorded_rounds += 1
        s = sets()
        sc = venv.scalars().cpu().numpy()
        pos = venv.positions().cpu().numpy()
        one_hop = venv.one_hop().cpu().numpy().view(np.uint64)
        for b, pz in enumerate(refs):
            e = pz.env
            assert set_int(s[b, L.SET_HAS_MESSAGE]) == e.has_message and set_int(s[b, L.SET_AGENTS]) == e.agents
            assert set_int(s[b, L.SET_TERMINATED]) == e.terminated and set_int(s[b, L.SET_ALIVE]) == e.alive
            assert int(sc[b, L.S_SELECTION]) == e.agent_selection and int(sc[b, L.S_NUM_MOVES]) == e.num_moves
            np.testing.assert_array_equal(pos[b], e.pos)
            assert set_ints(one_hop[b]) == e.adj
    c = loop.counters()
    assert c["errors"] == 0 and c["episodes"] >= 3 and checked_rows > 100 and recorded_rounds > 100
    if supply == "stream":                     # every env went round its ring at least once
        assert int(venv.scalars()[:, L.S_EP_CURSOR].min()) > ring
    # episode log: one row per finished episode = the logger_stats of its final observation (graph.py:166-178), per env
    # in the order the episodes ended
    stats, meta, total = venv.read_episode_log()
    assert total == len(stats) == sum(len(getattr(pz, "finished", [])) for pz in refs) and total >= B
    for b, pz in enumerate(refs):
        mine = [k for k in range(total) if meta[k, 0] == b]
        assert len(mine) == len(getattr(pz, "finished", []))
        for k, (want, moves) in zip(mine, pz.finished):
            np.testing.assert_array_equal(stats[k], [float(want[key]) for key in L.LOGGER_KEYS])
            assert meta[k, 2] == moves
    # the collectors' result built from that log (collect.result_from_episode_log, what MultiAgentCollector.collect returns):
    # `returns` = the oracle env's per-episode reward sums (GraphEnv.episode_rewards_sum, graph.py:237,389: every reward the world
    # steps of the episode handed out - the oracle's bookkeeping is pinned by the real reference's traces, logger_stats included),
    # `lens` = its episode lengths, in the reference's field names
    from melissa_amd.collect import result_from_episode_log
    res = result_from_episode_log(stats, meta, total, steps=c["decisions"], dt=1.0)
    assert res.n_collected_episodes == total and res.n_collected_steps == c["decisions"] and len(res.returns) == total
    for b, pz in enumerate(refs):
        mine = [k for k in range(total) if meta[k, 0] == b]
        np.testing.assert_array_equal(res.returns[mine], [w["episode_rewards_sum"] for w, _ in pz.finished])
        np.testing.assert_array_equal(res.lens[mine], [m for _, m in pz.finished])
    assert res.returns_stat.mean == pytest.approx(float(np.mean(res.returns))) and res.info.stats["coverage"].max <= 1.0



def test_graph_replay_matches_eager_launches():
    """The round step captured once into a HIP graph and replayed must walk exactly the same trajectory as
    the eager launch sequence (incl. the eps-greedy stream, which is keyed on a device-side round counter) - replayed round by
    round, and in groups of four rounds per graph (RoundLoop.run: 14 group replays + 2 single rounds here), with the episode
    stream's refills paced on their side stream in all three."""
    from melissa_amd.collect import RoundLoop
    from melissa_amd.env import HipGraphVectorEnv, synthetic_graph_pool
    from melissa_amd.policy import DQNPolicy
    n, B = 20, 64
    graphs = synthetic_graph_pool(n, 4, first_seed=5)
    net, _ = make_ldgn(n)
    finals = []
    for use_graph, rounds_per_graph in ((False, 1), (True, 1), (True, 4)):
        venv = HipGraphVectorEnv(B, n, graph_pool=graphs, dynamic_graph=True, device="cuda", max_moves=48,
                                 construct_like_reference=False)
        from melissa_amd.replay import RoundReplay
        replay = RoundReplay(B, n, 8, "cuda")                # (the learner's loops record while they replay graphs)
        loop = RoundLoop(venv, DQNPolicy(net), episodes_per_env=10, seed=3, eps=0.05, use_graph=use_graph,
                         graph_rounds=rounds_per_graph, replay=replay)
        loop.run(2)                      # warm-up + capture of the one-round graph
        loop.run(58)
        torch.cuda.synchronize()
        assert (loop.group_graph is not None) == (rounds_per_graph == 4)
        finals.append((venv.scalars().cpu().numpy().copy(), venv.node_sets().cpu().numpy().copy(),
                       venv.positions().cpu().numpy().copy(), loop.counters(),
                       [t.cpu().numpy().copy() for t in (replay.obs, replay.obs_next, replay.acted, replay.done, replay.act,
                                                        replay.rew, replay.episode, replay.cursor)]))
    for other in finals[1:]:
        np.testing.assert_array_equal(finals[0][0], other[0])
        np.testing.assert_array_equal(finals[0][1], other[1])
        np.testing.assert_array_equal(finals[0][2], other[2])
        assert finals[0][3] == other[3]
        for mine, theirs in zip(finals[0][4], other[4]):
            np.testing.assert_array_equal(mine, theirs)
    assert finals[0][3]["errors"] == 0 and finals[0][3]["episodes"] > 20 and finals[0][3]["iterations"] == 60


def test_replay_sampling_and_dqn_learner():
    """Collect with the round loop into the device replay, check the n-step sampler against a slow walk over
    the records, then take a few DQN steps (target net through the HIP forward, autograd learn step)."""
    from melissa_amd.collect import RoundLoop
    from melissa_amd.env import HipGraphVectorEnv, synthetic_graph_pool
    from melissa_amd.networks import LDGNNetwork
    from melissa_amd.policy import DQNPolicy
    from melissa_amd.replay import DQNLearner, RoundReplay
    n, B, K, n_step, gamma = 20, 32, 16, 4, 0.99
    graphs = synthetic_graph_pool(n, 4, first_seed=5)
    torch.manual_seed(0)
    net = LDGNNetwork(5, 128, 2, 4, n, dueling_param=DUEL(), device="cuda", backend="auto")
    policy = DQNPolicy(net, torch.optim.Adam(net.parameters(), lr=1e-3), estimation_step=n_step, target_update_freq=2)
    venv = HipGraphVectorEnv(B, n, graph_pool=graphs, dynamic_graph=True, device="cuda", max_moves=48,
                             construct_like_reference=False)
    replay = RoundReplay(B, n, K, "cuda")
    with pytest.raises(ValueError):
        replay.sample(8, n_step, gamma)                       # nothing recorded yet: an error, not a batch of garbage
    loop = RoundLoop(venv, policy, episodes_per_env=10, seed=3, eps=0.3, replay=replay)
    with torch.no_grad():
        loop.run(40)
    torch.cuda.synchronize()
    assert len(replay) > 500 and int(replay.cursor.min()) >= K
    s = replay.sample(256, n_step, gamma, torch.Generator(device="cuda").manual_seed(1))
    acted = replay.acted.cpu().numpy().view(np.uint64)
    done = replay.done.cpu().numpy().view(np.uint64)
    rew, epi, cur = replay.rew.cpu().numpy(), replay.episode.cpu().numpy(), replay.cursor.cpu().numpy()
    for e, k, a, ret, bw, obs, act in zip(*(s[x].cpu().numpy() for x in ("env", "slot", "agent", "ret", "boot_w", "obs", "act"))):
        assert (int(acted[e, k]) >> a) & 1 and obs[-1] == a and act == replay.act[e, k, a].item()
        want, w, kk, newest = 0.0, 1.0, k, (cur[e] - 1) % K
        for j in range(n_step):
            if epi[e, kk] != epi[e, k] or not (int(acted[e, kk]) >> a) & 1:
                break
            want += gamma ** j * rew[e, kk, a]
            w = gamma ** (j + 1)
            if (int(done[e, kk]) >> a) & 1:
                w = 0.0
                break
            if kk == newest:
                break
            kk = (kk + 1) % K
        assert abs(ret - want) < 1e-5 and abs(bw - w) < 1e-6
    # obs / boot_obs rows: the record's obs_matrix (the bootstrap row: obs_next of the LAST slot the walk accepted) | agent id
    np.testing.assert_array_equal(s["obs"][:, :-1].cpu().numpy(), replay.obs[s["env"], s["slot"]].cpu().numpy())
    assert torch.equal(s["boot_obs"][:, -1], s["agent"].float()) and s["boot_obs"].shape == s["obs"].shape
    # the one-launch sampler (mel_replay_sample) is uniform over (record, acting agent) pairs: 64 batches of 1024 draws against the
    # pair counts per env (the ring is full: every slot valid), and every call advances the device-side draw counter
    gen = torch.Generator(device="cuda").manual_seed(5)
    draws = torch.cat([replay.sample(1024, n_step, gamma, gen)["env"] for _ in range(64)])
    pairs = np.array([[bin(int(m)).count("1") for m in row] for row in acted]).sum(axis=1).astype(np.float64)
    freq = np.bincount(draws.cpu().numpy(), minlength=B).astype(np.float64)
    expect = pairs / pairs.sum() * freq.sum()
    chi2 = float(((freq - expect) ** 2 / expect).sum())
    assert chi2 < 2.0 * B, (chi2, B)                      # B - 1 degrees of freedom: mean B - 1, sd ~ sqrt(2 B)
    a, b = replay.sample(256, n_step, gamma, gen), replay.sample(256, n_step, gamma, gen)
    assert not torch.equal(a["env"] * K + a["slot"], b["env"] * K + b["slot"])
    learner = DQNLearner(policy, replay, batch_size=64, n_step=n_step, gamma=gamma, seed=2)
    before = torch.cat([p.detach().flatten().clone() for p in net.parameters()])
    losses = [learner.step()["loss"] for _ in range(5)]
    after = torch.cat([p.detach().flatten() for p in net.parameters()])
    assert all(np.isfinite(l) for l in losses) and not torch.equal(before, after)
    assert any(k.startswith("model_old.") for k in policy.state_dict())


def test_hldgn_round_forward_at_the_per_gpu_share_matches_oracle():
    """HL-DGN, 50 nodes, 512 envs - one GPU's share of BASELINE.json configs[3] (4096 envs over 8 GPUs): every env's logits of
    several rounds of the device-resident loop against the oracle (<= 1e-4), every live agent's action against the oracle's
    argmax of its env's row wherever the margin exceeds the tolerance."""
    from melissa_amd.collect import RoundLoop
    from melissa_amd.env import HipGraphVectorEnv, synthetic_graph_pool
    from melissa_amd.networks import HLDGNNetwork
    from melissa_amd.policy import DQNPolicy
    from oracle import net_oracle as no
    n, B = 50, 512
    venv = HipGraphVectorEnv(B, n, graph_pool=synthetic_graph_pool(n, 16, first_seed=50), dynamic_graph=True, device="cuda",
                             max_moves=48, seed=321, construct_like_reference=False)
    sd = no.init_weights("hl_dgn", seed=9, random_conv_bias=True)
    net = HLDGNNetwork(5, 128, 2, 4, n, aggregator="max", dueling_param=DUEL(), device="cuda", backend="hip")
    net.load_state_dict(sd)
    loop = RoundLoop(venv, DQNPolicy(net), eps=0.0, seed=5)
    torch.set_num_threads(16)
    checked = 0
    for it in range(9):
        live = loop.live.cpu().numpy().view(np.uint64).reshape(B, -1)[:, 0].copy()
        mat = venv.obs_matrix().cpu().numpy().copy()
        loop.step()
        torch.cuda.synchronize()
        if it % 3 != 2:
            continue
        checked += 1
        want = no.hldgn_forward(sd, np.concatenate([mat, np.zeros((B, 1), np.float32)], axis=1), n, aggregator="max").numpy()
        got = loop.logits.cpu().numpy()
        np.testing.assert_allclose(got, want, atol=TOL, rtol=0)
        act = loop.act.cpu().numpy().reshape(B, n)
        srt = np.sort(want, axis=1)
        clear = srt[:, -1] - srt[:, -2] > 2 * TOL
        member = ((live[:, None] >> np.arange(n, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(bool)
        best = np.broadcast_to(want.argmax(axis=1)[:, None], (B, n))
        sel = member & clear[:, None]
        np.testing.assert_array_equal(act[sel], best[sel])
        print(f"round {it}: {int(member.sum())} live agents, max |logit error| {np.abs(got - want).max():.2e}")
    assert checked == 3 and loop.counters()["errors"] == 0


@pytest.mark.parametrize("n", [20, 50, 100])
@pytest.mark.parametrize("scripted", [None, (0.3, "simple_broadcast"), (0.4, "broadcast_if_any_interested")],
                         ids=["all-policy", "scripted-broadcast", "scripted-interested"])
def test_hldgn_round_loop_matches_oracle(scripted, n):
    """HL-DGN in the round loop: one logits row per env (hl_dgn.py:108 ignores the controlling index), dense
    per-agent actions; env state after every round equals the oracle replaying the same actions.  With scripted
    agents the round's active set excludes them, their rows are zeroed by the dm mask before pooling (hl_dgn.py:105)
    and they relay by their heuristic inside the world step."""
    from melissa_amd import _lib as L
    from melissa_amd.collect import RoundLoop, sample_episode_table
    from melissa_amd.env import HipGraphVectorEnv, synthetic_graph_pool
    from melissa_amd.networks import HLDGNNetwork
    from melissa_amd.policy import DQNPolicy
    from oracle import env_oracle as eo
    from oracle import net_oracle as no
    B, seed, K = 5, 41, 30
    graphs = synthetic_graph_pool(n, 3, first_seed=50)
    skw = dict(scripted_agents_ratio=scripted[0], heuristic=scripted[1]) if scripted else {}
    venv = HipGraphVectorEnv(B, n, graph_pool=graphs, dynamic_graph=True, device="cuda", max_moves=48,
                             construct_like_reference=False, **skw)
    sd = no.init_weights("hl_dgn", seed=9, random_conv_bias=True)
    net = HLDGNNetwork(5, 128, 2, 4, n, aggregator="max", dueling_param=DUEL(), device="cuda", backend="hip")
    net.load_state_dict(sd)
    packed, table = sample_episode_table(venv, 12, seed)
    loop = RoundLoop(venv, DQNPolicy(net), eps=0.0, seed=seed,
                     episodes=(packed, np.ascontiguousarray(table[:, 1:])))
    refs = []
    for b in range(B):
        env = eo.OracleGraphEnv(n, graph_pool=[eo.GraphSpec(g.pos.copy(), set_ints(g.one_hop)) for g in graphs],
                                dynamic_graph=True,
                                np_random=np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed + b))), **skw)
        pz = eo.OraclePettingZooEnv.__new__(eo.OraclePettingZooEnv)
        pz.env, pz.n, pz.rewards, pz.done_count = env, n, [0] * n, 0
        env.last()
        refs.append(pz)
    saw_scripted = False
    for it in range(K):
        live = loop.live.cpu().numpy().view(np.uint64).copy()
        mat = venv.obs_matrix().cpu().numpy().copy()
        if scripted:
            sets = venv.node_sets().cpu().numpy().view(np.uint64)
            for b, pz in enumerate(refs):
                assert set_int(sets[b, L.SET_SCRIPTED]) == pz.env.scripted and set_int(live[b]) & pz.env.scripted == 0
                saw_scripted |= pz.env.scripted != 0
        loop.step()
        torch.cuda.synchronize()
        logits = loop.logits.cpu().numpy()
        act = loop.act.cpu().numpy().reshape(B, n)
        obs_rows = np.concatenate([mat, np.zeros((B, 1), np.float32)], axis=1)
        want = no.hldgn_forward(sd, obs_rows, n, aggregator="max").numpy()
        np.testing.assert_allclose(logits, want, atol=TOL, rtol=0)
        for b, pz in enumerate(refs):
            acts = {a: act[b, a] for a in range(n) if (set_int(live[b]) >> a) & 1}
            assert all(v == int(np.argmax(logits[b])) for v in acts.values())
            oracle_round(pz, acts)
        s = venv.node_sets().cpu().numpy().view(np.uint64)
        for b, pz in enumerate(refs):
            assert set_int(s[b, L.SET_HAS_MESSAGE]) == pz.env.has_message and set_int(s[b, L.SET_AGENTS]) == pz.env.agents
            np.testing.assert_array_equal(venv.positions()[b].cpu().numpy(), pz.env.pos)
    assert loop.counters()["errors"] == 0
    assert saw_scripted == bool(scripted)


@pytest.mark.parametrize("model", ["hl_dgn", "dgn_r"])
def test_training_loop_single_rank(model):
    """collect (HIP) -> device replay -> n-step targets (HIP target net) -> autograd step with the flat-gradient
    hook, end to end on one rank (the N > 1 gradient averaging itself is covered on CPU with gloo)."""
    from melissa_amd.train import train
    out = train(model=model, n_nodes=12, envs=48, updates=4, rounds_per_update=3, batch_size=32, log=lambda *_: None)
    assert out["errors"] == 0 and out["decisions"] > 200 and out["replicas_identical"]
    assert np.isfinite(out["loss_first"]) and np.isfinite(out["loss_last"])


@pytest.mark.parametrize("model", ["l_dgn", "dgn_r", "hl_dgn"])
def test_training_loop_n50_first_update_matches_oracle_autograd(model):
    """BASELINE configs 4 / 5 at their graph size (N = 50): the training loop's FIRST update is re-derived by the oracle -
    the loss of policies/dgn.py:49-64 (DGN-R: sampled returns regress on the summed sibling Q) or [3P] DQNPolicy.learn
    (L-DGN / HL-DGN) from the oracle's forward on the sampled batch with the pre-update weights, and the gradient of
    that loss by torch autograd through the oracle against the gradient the HIP learn path produced (recovered from
    the Adam step: first step from zero moments moves every weight by lr * sign(g) ... so the check is on the loss and
    on the parameter delta's sign pattern where |g| is not negligible)."""
    from melissa_amd.train import train
    from oracle import net_oracle as no
    n, lr = 50, 1e-3
    cap = {}

    def probe(k, net, learner, phase):
        if k != 0:
            return
        if phase == "before":
            cap["sd"] = {key: v.detach().cpu().clone() for key, v in net.state_dict().items()}
        else:
            cap["batch"] = {key: v.detach().cpu() for key, v in learner.last_batch.items()}
            cap["after"] = {key: v.detach().cpu().clone() for key, v in net.state_dict().items()}
            cap["grad"] = {key: p.grad.detach().cpu().clone() for key, p in net.named_parameters() if p.grad is not None}

    out = train(model=model, n_nodes=n, envs=64, updates=2, rounds_per_update=3, batch_size=32, lr=lr,
                log=lambda *_: None, probe=probe)
    assert out["errors"] == 0 and out["replicas_identical"] and out["decisions"] > 500
    sd = {k: v.clone().requires_grad_(True) for k, v in cap["sd"].items()}
    b = cap["batch"]
    torch.set_num_threads(8)
    fwd = {"l_dgn": no.ldgn_forward, "dgn_r": no.dgnr_forward, "hl_dgn": no.hldgn_forward}[model]
    if model == "dgn_r":
        logits = fwd(sd, b["active_obs"].numpy(), n)
        q = logits[torch.arange(len(b["active_act"])), b["active_act"]]
        batch_q = torch.zeros_like(b["returns"]).index_add(0, b["segment"], q)            # dgn.py:43-55
        loss = (b["returns"] - batch_q).pow(2).mean()                                       # dgn.py:57-64
    else:
        logits = fwd(sd, b["obs"].numpy(), n)
        q = logits[torch.arange(len(b["act"])), b["act"]]
        loss = (b["returns"] - q).pow(2).mean()
    loss.backward()
    assert abs(float(loss.detach()) - out["loss_first"]) <= 1e-4 * max(1.0, abs(float(loss.detach()))), (float(loss.detach()), out["loss_first"])
    # Adam's first step from zero moments: delta = -lr * g / (|g| + eps) -> -lr * sign(g) wherever |g| >> 1e-8
    checked = agree = 0
    for k, g in ((k, v.grad) for k, v in sd.items()):
        if g is None:
            continue
        delta = cap["after"][k] - cap["sd"][k]
        big = g.abs() > 1e-5 * float(g.abs().max() + 1e-30)
        checked += int(big.sum())
        agree += int((torch.sign(delta[big]) == -torch.sign(g[big])).sum())
    assert checked > 10000 and agree >= 0.999 * checked, (checked, agree)
    # ... and the gradient itself, number for number: what the HIP learn path (csrc/grad.hip backward kernels + the library's
    # GEMMs) left in .grad against oracle autograd on the same batch and weights - every tensor within 2e-4 of its largest entry
    # (the bar of tests/test_gpu_grad.py, here at the training loop's own size: N = 50, 32 sampled experiences, DGN-R: all their
    # siblings)
    compared = 0
    overall = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    for k, g in ((k, v.grad) for k, v in sd.items()):
        if g is None:
            continue
        assert k in cap["grad"], k
        got = cap["grad"][k]
        # (a tensor whose gradient is zero in exact arithmetic - TransformerConv's key bias: the softmax is invariant to it - holds
        # rounding noise ~1e-10 on both sides: its bar is set by the gradient's overall scale, not by its own)
        scale = max(float(g.abs().max()), 1e-3 * overall)
        assert float((got - g).abs().max()) <= 2e-4 * scale, (k, float((got - g).abs().max()), scale)
        compared += g.numel()
    assert compared > 300000


@pytest.mark.parametrize("n,B", [(20, 32), (100, 8)])
def test_collector_surface_counts_and_episode_stats(n, B):
    """melissa_amd.collect.Collector: collect(n_step) / collect(n_episode) like the reference's collectors, evaluation
    envs built with is_testing=True (l_dgn.py:92-129).  (100 nodes: two-word node sets through the same surface.)"""
    from melissa_amd import _lib as L
    from melissa_amd.collect import Collector
    from melissa_amd.env import HipGraphVectorEnv, synthetic_graph_pool
    from melissa_amd.policy import DQNPolicy
    graphs = synthetic_graph_pool(n, 4, first_seed=50)
    net, _ = make_ldgn(n)
    venv = HipGraphVectorEnv(B, n, graph_pool=graphs, dynamic_graph=True, device="cuda", max_moves=48, seed=5,
                             construct_like_reference=False, is_testing=True, num_test_episodes=10)
    col = Collector(DQNPolicy(net), venv, episodes_per_env=40, seed=5, chunk=4)
    with pytest.raises(ValueError):
        col.collect()
    out = col.collect(n_step=2000)
    assert out["n/st"] >= 2000 and out["collect_speed"] > 0 and out["n/ep"] == len(out["lens"])
    out = col.collect(n_episode=50)
    assert out["n/ep"] >= 50 and set(L.LOGGER_KEYS) <= set(out)
    assert 0.0 < out["coverage"] <= 1.0 and out["total_messages_transmitted"] >= 1.0 and out["len"] >= 1.0
    assert (out["episode_info"]["coverage"] <= 1.0).all() and (out["lens"] >= 1).all()
    assert col.collect_step >= 2000 and col.collect_episode >= 50


def test_multi_agent_collector_is_called_like_the_reference():
    """l_dgn.py:119-127 / 185-201: ``MultiAgentCollector(agents_num=..., policy=masp_policy, env=envs, buffer=...,
    exploration_noise=...)``, ``.reset()``, ``.collect(n_step=...)`` / ``.collect(n_episode=...)`` -> the reference's result fields."""
    from melissa_amd import _lib as L
    from melissa_amd.collect import MultiAgentCollector
    from melissa_amd.env import HipGraphVectorEnv, synthetic_graph_pool
    from melissa_amd.policy import DQNPolicy, MultiAgentSharedPolicy
    from melissa_amd.replay import RoundReplay
    n, B = 20, 32
    net, _ = make_ldgn(n)
    policy = DQNPolicy(net)
    venv = HipGraphVectorEnv(B, n, graph_pool=synthetic_graph_pool(n, 4, first_seed=50), dynamic_graph=True, device="cuda",
                             max_moves=48, seed=5, construct_like_reference=False)
    masp = MultiAgentSharedPolicy(policy, venv)              # shared_policy.py:23-30: (policy, env)
    assert masp.agents == [str(i) for i in range(n)]
    buf = RoundReplay(B, n, 16, "cuda")
    policy.set_eps(0.3)
    train = MultiAgentCollector(agents_num=n, policy=masp, env=venv, buffer=buf, exploration_noise=True)
    train.reset()
    res = train.collect(n_step=32 * 8)
    assert res.n_collected_steps >= 256 and res.collect_speed > 0 and res.collect_time > 0
    assert len(res.returns) == res.n_collected_episodes == len(res.lens)
    assert int(buf.cursor.max()) > 0                          # the rounds were recorded
    policy.set_eps(0.0)
    res = train.collect(n_episode=40)
    assert res.n_collected_episodes >= 40 and res.returns_stat.min <= res.returns_stat.mean <= res.returns_stat.max
    assert set(res.info.stats) == set(L.LOGGER_KEYS) and 0.0 < res.info.stats["coverage"].mean <= 1.0
    rnd = train.collect(n_step=64, random=True)               # uniformly random actions
    assert rnd.n_collected_steps >= 64
    with pytest.raises(AssertionError):
        train.collect(n_step=10, n_episode=2)
    with pytest.raises(TypeError):
        train.collect()
    with pytest.raises(ValueError):
        MultiAgentCollector(agents_num=n + 1, policy=masp, env=venv)


@pytest.mark.parametrize("n", [50, 100])
def test_decision_loop_runs_clean_and_matches_round_loop_counts(n):
    """The AEC-order loop (one agent decision per env per step, the reference collector's granularity) and the round loop are
    two schedules of the same trajectories: with a greedy policy and the same episode stream seeds both run clean and
    finish their envs' episodes at the same pace (100 nodes: two-word node sets)."""
    from melissa_amd.collect import DecisionLoop, RoundLoop
    from melissa_amd.env import HipGraphVectorEnv, synthetic_graph_pool
    from melissa_amd.policy import DQNPolicy
    B = 16
    graphs = synthetic_graph_pool(n, 3, first_seed=50)
    net, _ = make_ldgn(n)
    mk = lambda: HipGraphVectorEnv(B, n, graph_pool=graphs, dynamic_graph=True, device="cuda", max_moves=64,
                                   construct_like_reference=False)
    rl = RoundLoop(mk(), DQNPolicy(net), eps=0.0, seed=11)
    rl.run(40)
    c_round = rl.counters()
    dl = DecisionLoop(mk(), DQNPolicy(net), eps=0.0, seed=11)
    # run the AEC loop until it has made at least as many decisions, then compare at equal episode counts per env
    for _ in range(60):
        dl.run(100)
        if dl.counters()["decisions"] >= c_round["decisions"]:
            break
    c_aec = dl.counters()
    assert c_round["errors"] == 0 and c_aec["errors"] == 0
    assert c_round["episodes"] >= B and c_aec["decisions"] >= c_round["decisions"]
    # (the AEC loop was stopped by its TOTAL decision count, so an env may be an episode short of its round-loop twin)
    sc_r, sc_a = rl.venv.scalars().cpu().numpy(), dl.venv.scalars().cpu().numpy()
    from melissa_amd import _lib as L
    assert (sc_a[:, L.S_EPISODES_DONE] >= sc_r[:, L.S_EPISODES_DONE] - 1).all()


@pytest.mark.parametrize("model", ["l_dgn", "hl_dgn"])
def test_watch_single_env(model):
    """BASELINE config 0: --watch, 20-node graphs, a single env (evaluation schedule, greedy policy)."""
    from melissa_amd.watch import watch
    out = watch(model=model, n_nodes=20, envs=1, episodes=6)
    assert out["n/ep"] >= 6 and 0.0 < out["coverage"] <= 1.0 and out["total_messages_transmitted"] >= 1


@pytest.mark.parametrize("n", [20, 100])
@pytest.mark.parametrize("dueling", [True, False])
def test_hldgn_fused_selection_equals_the_separate_launch(dueling, n):
    """mel_hldgn_forward_envs_select writes, for every agent of live[b], the action mel_select_action_envs draws from the
    same logits (same counter-based stream), for the dueling heads (fused finish kernel) and a plain out_linear head."""
    import ctypes as C
    from melissa_amd import _lib
    from melissa_amd.networks import HLDGNNetwork
    bs = 300
    torch.manual_seed(4)
    net = HLDGNNetwork(5, 128, 2, 4, n, aggregator="max", dueling_param=DUEL() if dueling else None, device="cuda",
                       backend="hip")
    rng = np.random.RandomState(8)
    m = np.zeros((bs, n, 8), dtype=np.float32)
    m[:, :, 0:2] = rng.uniform(0, 1, size=(bs, n, 2))
    m[:, :, 2:7] = rng.randint(0, 3, size=(bs, n, 5))
    m[:, :, 7] = 1.0
    obs = torch.from_numpy(m.reshape(bs, n * 8)).cuda()
    member = rng.randint(0, 2, size=(bs, n)).astype(bool)          # the agents of every env, as node sets (one / two words)
    member[::7] = False
    live_np = agent_masks(bs, n)
    for b, a in zip(*np.nonzero(member)):
        add_agent(live_np, int(b), int(a))
    live = torch.from_numpy(live_np.view(np.int64)).cuda()
    rounds = torch.tensor([5], dtype=torch.int32, device="cuda")
    lib = _lib.load()
    for eps in (0.0, 0.4):
        act = torch.full((bs * n,), -1, dtype=torch.int32, device="cuda")
        sel = _lib.MelSelect()
        sel.act, sel.eps, sel.seed, sel.step_dev = act.data_ptr(), eps, 1234, rounds.data_ptr()
        sel.live, sel.n_nodes = live.data_ptr(), n
        with torch.no_grad():
            logits = net.hip_forward_envs(obs, select=sel)
            plain = net.hip_forward_envs(obs)
        assert torch.equal(logits, plain)
        want = torch.full((bs * n,), -1, dtype=torch.int32, device="cuda")
        _lib.check(lib.mel_select_action_envs(logits.data_ptr(), live.data_ptr(), bs, n, 2, C.c_float(eps), 1234,
                                              rounds.data_ptr(), want.data_ptr(), _lib.current_stream_ptr()))
        assert torch.equal(act, want)
        assert torch.equal(act.view(bs, n).cpu() >= 0, torch.from_numpy(member))      # exactly the member agents got an action


@pytest.mark.parametrize("dtype", ["f32s", "bf16"])
def test_captured_update_targets_follow_the_target_network_sync(dtype):
    """The target network only ever runs INSIDE the captured graph, and on the split / bf16 precisions its launches read CONVERTED
    projection weights from a buffer captured with them.  After every sync_weight() those planes must be reconverted (eagerly,
    into the same buffer - a replay runs no Python): the graph's `returns` must equal ret + boot_w * max_a Q_target(boot_obs)
    recomputed eagerly through model_old after each update, across several target syncs."""
    from melissa_amd.collect import RoundLoop
    from melissa_amd.env import HipGraphVectorEnv, synthetic_graph_pool
    from melissa_amd.networks import HLDGNNetwork
    from melissa_amd.policy import DQNPolicy
    from melissa_amd.replay import DQNLearner, RoundReplay
    n, envs = 20, 64
    torch.manual_seed(3)
    net = HLDGNNetwork(5, 128, 2, 4, n, aggregator="max", dueling_param=DUEL(), device="cuda")
    net.set_feature_dtype(dtype)
    policy = DQNPolicy(net, torch.optim.Adam(net.parameters(), lr=1e-2), estimation_step=4, target_update_freq=2)
    policy.model_old.set_feature_dtype(dtype)
    venv = HipGraphVectorEnv(envs, n, graph_pool=synthetic_graph_pool(n, 8, 0), dynamic_graph=True, device="cuda", max_moves=48,
                             seed=11, construct_like_reference=False)
    replay = RoundReplay(envs, n, 16, "cuda")
    loop = RoundLoop(venv, policy, seed=11, eps=0.1, replay=replay)
    with torch.no_grad():
        loop.run(20)
    learner = DQNLearner(policy, replay, batch_size=32, n_step=4, gamma=0.99, seed=2)
    learner.capture()
    moved = 0.0
    for k in range(7):                                        # target syncs at iterations 2, 4, 6, 8 (two warm-up updates ran)
        before = [p.detach().clone() for p in policy.model_old.parameters()]
        learner.step()
        torch.cuda.synchronize()
        b = {key: v.clone() for key, v in learner.last_batch.items()}
        moved = max(moved, max(float((p.detach() - q).abs().max()) for p, q in zip(policy.model_old.parameters(), before)))
        fresh = HLDGNNetwork(5, 128, 2, 4, n, aggregator="max", dueling_param=DUEL(), device="cuda")
        fresh.set_feature_dtype(dtype)
        fresh.load_state_dict(policy.model_old.state_dict())  # a network that converts these very weights now
        with torch.no_grad():
            want = b["ret"] + b["boot_w"] * fresh.hip_forward(b["boot_obs"]).max(dim=1).values
        assert torch.equal(b["returns"], want), (k, float((b["returns"] - want).abs().max()))
    assert moved > 1e-4                                       # the target weights really changed in between


@pytest.mark.parametrize("collective", [False, True], ids=["one_graph", "pack_reduce_unpack"])
@pytest.mark.parametrize("model", ["hl_dgn", "l_dgn", "dgn_r"])
def test_captured_update_equals_the_eager_update(model, collective):
    """DQNLearner.capture(): the update replayed from HIP graphs (sample -> n-step targets -> forward / backward ->
    [pack | eager reduce | unpack] -> Adam) must move the parameters exactly like an eager update on the same batch from the same
    weights and optimizer state - for several consecutive replays (Adam's step counter lives on the device and advances in the
    graph), across a target-network sync, and the collect loop must see the new weights afterwards (its prepared bf16 planes are
    keyed on torch's version counters, which a replay does not bump)."""
    import copy
    from melissa_amd import parallel
    from melissa_amd.collect import RoundLoop
    from melissa_amd.env import HipGraphVectorEnv, synthetic_graph_pool
    from melissa_amd.networks import DGNRNetwork, HLDGNNetwork, LDGNNetwork
    from melissa_amd.policy import DGNPolicy, DQNPolicy
    from melissa_amd.replay import DGNLearner, DQNLearner, RoundReplay
    n, envs = 20, 64
    # dgn_r: DGNPolicy's summed-sibling loss in its dense form (static shapes: one graph per sampled experience)
    policy_cls, learner_cls = (DGNPolicy, DGNLearner) if model == "dgn_r" else (DQNPolicy, DQNLearner)

    def make_policy():
        torch.manual_seed(3)
        if model == "hl_dgn":
            net = HLDGNNetwork(5, 128, 2, 4, n, aggregator="max", dueling_param=DUEL(), device="cuda")
        elif model == "dgn_r":
            net = DGNRNetwork(5, 128, 2, 4, n, dueling_param=DUEL(), device="cuda")
        else:
            net = LDGNNetwork(5, 128, 2, 4, n, dueling_param=DUEL(), device="cuda")
        return net, policy_cls(net, torch.optim.Adam(net.parameters(), lr=1e-3), estimation_step=4, target_update_freq=3)

    net, policy = make_policy()
    venv = HipGraphVectorEnv(envs, n, graph_pool=synthetic_graph_pool(n, 8, 0), dynamic_graph=True, device="cuda", max_moves=48,
                             seed=11, construct_like_reference=False)
    replay = RoundReplay(envs, n, 16, "cuda")
    loop = RoundLoop(venv, policy, seed=11, eps=0.1, replay=replay)
    with torch.no_grad():
        loop.run(20)

    class LocalReducer(parallel.FlatGradAllReducer):        # the three phases with the collective of a one-rank world
        active = staticmethod(lambda: True)

        def reduce(self):
            self.flat.mul_(1.0)

    hook = LocalReducer(net) if collective else None
    learner = learner_cls(policy, replay, batch_size=32, n_step=4, gamma=0.99, grad_hook=hook, seed=2)
    learner.capture()
    twin_net, twin = make_policy()
    for k in range(5):                                       # covers two target syncs (every 3 updates; 2 warm-up updates ran)
        twin_net.load_state_dict(net.state_dict())
        twin.model_old.load_state_dict(policy.model_old.state_dict())
        twin.optim.load_state_dict(copy.deepcopy(policy.optim.state_dict()))
        twin._iter = policy._iter
        out = learner.step()
        batch = {key: v.clone() for key, v in learner.last_batch.items()}
        want = twin.learn(batch)
        assert abs(float(out["loss"]) - want["loss"]) <= 1e-5 * max(1.0, abs(want["loss"]))
        for (name, p), q in zip(net.named_parameters(), twin_net.parameters()):
            diff = float((p.detach() - q.detach()).abs().max())
            # (every gradient is a deterministic sum - conv*.att / conv*.bias too since round 3, per workgroup and then in
            # workgroup order - so the twin's update lands on the same values)
            assert diff <= 2e-6, (k, name, diff)
        for p, q in zip(policy.model_old.parameters(), twin.model_old.parameters()):
            assert torch.equal(p, q)
    with torch.no_grad():                                    # the collect loop picks the updated weights up
        loop.run(6)
        obs = venv.obs_matrix()[:8]
        idx = torch.zeros(8, 1, device="cuda")
        got = net.hip_forward(torch.cat([obs, idx], 1))
        twin_net.load_state_dict(net.state_dict())
        assert torch.equal(got, twin_net.hip_forward(torch.cat([obs, idx], 1)))
    assert loop.counters()["errors"] == 0
