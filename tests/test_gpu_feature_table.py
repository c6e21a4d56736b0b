"""GPU tests of the node-feature table (MEL_FWD_INTEGER_FEATURES, csrc/plan_masks.hpp): with env-produced observations the
encoder and the conv1 projections are evaluated once per distinct feature tuple instead of once per receptive-field row.
Per row it is the same arithmetic in the same order, so the bar is BIT-IDENTICAL logits with the row-list path (which the
other suites hold to 1e-4 of the oracle), plus the oracle itself on a batch large enough to take the table."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DUEL = lambda: ({"hidden_sizes": [128, 128]}, {"hidden_sizes": [128, 128]})


def env_like_obs(n, bs, seed, index_col=True):
    """Observation rows with GraphEnv's feature ranges (graph.py:261-269)."""
    rng = np.random.RandomState(seed)
    obs = np.zeros((bs, 8 * n + 1), dtype=np.float32)
    m = obs[:, :-1].reshape(bs, n, 8)
    m[:, :, 0:2] = rng.uniform(0, 1, size=(bs, n, 2))
    m[:, :, 2] = rng.randint(0, n, size=(bs, n))           # degree
    m[:, :, 3] = rng.randint(0, 5, size=(bs, n))           # messages transmitted
    m[:, :, 4:7] = rng.randint(0, 2, size=(bs, n, 3))      # last action, interested, has message
    m[:, :, 7] = (rng.uniform(size=(bs, n)) > 0.1)
    obs[:, -1] = rng.randint(0, n, size=bs)
    return obs if index_col else obs[:, :-1].copy()


def make(model, n, seed=9):
    from melissa_amd.networks import DGNRNetwork, HLDGNNetwork, LDGNNetwork
    from oracle import net_oracle as no
    sd = no.init_weights(model, seed=seed, random_conv_bias=True)
    cls = {"l_dgn": LDGNNetwork, "dgn_r": DGNRNetwork, "hl_dgn": HLDGNNetwork}[model]
    kw = dict(aggregator="max") if model == "hl_dgn" else {}
    net = cls(5, 128, 2, 4, n, dueling_param=DUEL(), device="cuda", backend="hip", **kw)
    net.load_state_dict(sd)
    return net, sd


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f32s", "f32a"])
@pytest.mark.parametrize("model", ["l_dgn", "dgn_r", "hl_dgn"])
@pytest.mark.parametrize("n,bs", [(20, 512), (50, 700), (100, 640)])
def test_table_path_is_bit_identical_to_row_lists(model, n, bs, dtype):
    """fp32 modes: bit-identical.  bf16: see below."""
    obs = torch.from_numpy(env_like_obs(n, bs, 5 + n)).cuda()
    net, _ = make(model, n)
    net.set_feature_dtype(dtype)
    with torch.no_grad():
        rows = net.hip_forward(obs).clone()
        assert int(net.hip_tap(3, bs)[0]) == 0                                  # row lists
        table = net.hip_forward(obs, integer_features=True).clone()
        t = net.hip_tap(3, bs).cpu().numpy()
    assert t[0] == n * 40 and t[1:].sum() == 0                                   # the table was used, every feature in range
    if dtype == "bf16":
        # the bf16 feature path's table takes its encoder rows from the exact-fp32 tile that rides with the plan lists (fp32
        # weights, rows rounded to bf16 once) where the row-list path runs a bf16 GEMM on bf16 weights: the same values up to
        # one more bf16 rounding of the encoder's weights, far inside that path's 2e-2 bar
        scale = float(rows.abs().max())
        assert float((rows - table).abs().max()) <= 1e-2 * max(1.0, scale)
        assert float((rows.argmax(dim=1) == table.argmax(dim=1)).float().mean()) >= 0.99
    else:
        assert torch.equal(rows, table)


@pytest.mark.parametrize("model", ["l_dgn", "hl_dgn"])
def test_table_path_matches_oracle(model):
    from oracle import net_oracle as no
    n, bs = 20, 300
    obs = env_like_obs(n, bs, 77)
    net, sd = make(model, n, seed=4)
    with torch.no_grad():
        got = net.hip_forward(torch.from_numpy(obs).cuda(), integer_features=True).cpu().numpy()
        assert int(net.hip_tap(3, bs)[0]) == n * 40
        torch.set_num_threads(8)
        want = (no.ldgn_forward if model == "l_dgn" else no.hldgn_forward)(sd, obs, n).numpy()
    np.testing.assert_allclose(got, want, atol=1e-4, rtol=0)


def test_small_batches_and_foreign_features():
    """Below ~2 x 64 N rows the table would be more work than the row lists: not used.  A feature that is not one of the env's
    integers is clamped into the table and its env flagged (a caller that sets the flag on foreign observations finds out)."""
    n = 20
    net, _ = make("l_dgn", n)
    small = torch.from_numpy(env_like_obs(n, 16, 1)).cuda()
    with torch.no_grad():
        a = net.hip_forward(small, integer_features=True).clone()
        assert int(net.hip_tap(3, 16)[0]) == 0 and torch.equal(a, net.hip_forward(small))
        obs = env_like_obs(n, 512, 2)
        obs[7, 8 * 3 + 2] = 2.5          # env 7, node 3: a fractional degree
        obs[9, 8 * 0 + 3] = 5.0          # env 9, node 0: five messages (a policy agent acts at most four times)
        out = net.hip_forward(torch.from_numpy(obs).cuda(), integer_features=True)
        t = net.hip_tap(3, 512).cpu().numpy()
    assert t[0] == n * 40 and sorted(np.nonzero(t[1:])[0].tolist()) == [7, 9] and torch.isfinite(out).all()


@pytest.mark.parametrize("model", ["l_dgn", "hl_dgn"])
def test_round_loop_with_and_without_table_walks_the_same_trajectory(model):
    """The benchmarked loop: 256 envs for 40 rounds with the table (default) and with row lists - bit-identical logits mean
    identical actions, hence identical env state, episode counts and replayed decisions."""
    from melissa_amd.collect import RoundLoop
    from melissa_amd.env import HipGraphVectorEnv, synthetic_graph_pool
    from melissa_amd.policy import DQNPolicy
    n, B = 20, 256
    graphs = synthetic_graph_pool(n, 8, first_seed=5)
    net, _ = make(model, n)
    finals = []
    for use_table in (True, False):
        venv = HipGraphVectorEnv(B, n, graph_pool=graphs, dynamic_graph=True, device="cuda", max_moves=48,
                                 construct_like_reference=False)
        loop = RoundLoop(venv, DQNPolicy(net), seed=3, eps=0.05, use_graph=True)
        loop.integer_features = use_table
        loop.run(40)
        torch.cuda.synchronize()
        ft = loop.feature_table()
        assert ft["bad_envs"] == 0 and (ft["table_rows"] == n * 40) == use_table
        finals.append((venv.scalars().cpu().numpy().copy(), venv.node_sets().cpu().numpy().copy(),
                       venv.positions().cpu().numpy().copy(), loop.logits.cpu().numpy().copy(), loop.counters()))
    for a, b in zip(finals[0][:4], finals[1][:4]):
        np.testing.assert_array_equal(a, b)
    assert finals[0][4] == finals[1][4] and finals[0][4]["errors"] == 0 and finals[0][4]["episodes"] > 100


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("model", ["l_dgn", "dgn_r", "hl_dgn"])
def test_prepared_tables_equal_per_call_tables(model, dtype):
    """Opt-in: the table prepared once per weight version (mel_prepare_feature_tables) gives the same bits as the table every
    call evaluates, and follows a weight change."""
    n, bs = 20, 512
    obs = torch.from_numpy(env_like_obs(n, bs, 3)).cuda()
    net, _ = make(model, n)
    net.set_feature_dtype(dtype)
    with torch.no_grad():
        per_call = net.hip_forward(obs, integer_features=True).clone()
        net.prepared_tables = True
        prepared = net.hip_forward(obs, integer_features=True).clone()
        assert net._weights().tables and torch.equal(per_call, prepared)
        net.conv1.lin_l.weight.mul_(0.5) if model != "dgn_r" else net.conv1.lin_key.weight.mul_(0.5)
        changed = net.hip_forward(obs, integer_features=True).clone()
        net.prepared_tables = False
        assert not torch.equal(changed, per_call) and torch.equal(changed, net.hip_forward(obs, integer_features=True))
