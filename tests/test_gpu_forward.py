"""GPU parity of the HIP forward (through the C ABI) against the network oracle and the committed
golden vectors.  Tolerance: 1e-4 absolute on fp32 logits (BASELINE.json north_star)."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDENS = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "net_golden_*.npz")))
TOL = 1e-4
DUEL = lambda: ({"hidden_sizes": [128, 128]}, {"hidden_sizes": [128, 128]})


def golden_named(tag):
    """The committed golden ``net_golden_<tag>.npz`` - by NAME (a position in the sorted list changes whenever a golden is added)."""
    hits = [p for p in GOLDENS if os.path.basename(p) == f"net_golden_{tag}.npz"]
    assert len(hits) == 1, f"golden {tag} missing from tests/golden ({[os.path.basename(p) for p in GOLDENS]})"
    return hits[0]


def make_net(model, n, seed, agg="max", dueling=True):
    from melissa_amd.networks import DGNRNetwork, HLDGNNetwork, LDGNNetwork
    from oracle import net_oracle as no
    sd = no.init_weights(model, seed=seed, random_conv_bias=True)
    if model == "dgn_r":
        net = DGNRNetwork(5, 128, 2, 4, n, dueling_param=DUEL(), device="cuda", backend="hip")
    elif model == "l_dgn":
        net = LDGNNetwork(5, 128, 2, 4, n, dueling_param=DUEL(), device="cuda", backend="hip")
    else:
        net = HLDGNNetwork(5, 128, 2, 4, n, aggregator=agg, dueling_param=DUEL(), device="cuda", backend="hip")
    net.load_state_dict(sd)
    return net, sd


@pytest.mark.parametrize("path", GOLDENS, ids=[os.path.basename(p)[11:-4] for p in GOLDENS])
def test_ldgn_matches_golden(path):
    g = np.load(path)
    n, obs = int(g["n"]), g["obs"]
    net, _ = make_net("l_dgn", n, int(g["weight_seed"]))
    with torch.no_grad():
        logits, state = net(obs)                       # numpy in, like the reference collector
    assert state is None and logits.is_cuda and logits.dtype == torch.float32
    np.testing.assert_allclose(logits.cpu().numpy(), g["ldgn_logits"], atol=TOL, rtol=0)
    # intermediates: adjacency bit-exact, head input (x_1 | x_2 | x_3) within tolerance
    from melissa_amd.env.episodes import sets_to_bool
    adj = sets_to_bool(net.hip_tap(0, obs.shape[0]).cpu().numpy().view(np.uint64), n)      # [bs, n, n] membership
    want = np.unpackbits(g["adj"], axis=-1, bitorder="little")[..., :n].astype(bool)
    np.testing.assert_array_equal(adj, want)
    xcat = net.hip_tap(1, obs.shape[0]).cpu().numpy()
    np.testing.assert_allclose(xcat, np.concatenate([g["ldgn_x_1"], g["ldgn_x_2"], g["ldgn_x_3"]], axis=1),
                               atol=TOL, rtol=0)


@pytest.mark.parametrize("path", GOLDENS, ids=[os.path.basename(p)[11:-4] for p in GOLDENS])
def test_dgnr_matches_golden(path):
    """TransformerConv path (dgn_r.py): logits and the x_1|x_2|x_3 head input; the goldens include a clique
    (neighbour cap) and a lattice with isolated nodes (no self-loops -> zero rows)."""
    g = np.load(path)
    n, obs = int(g["n"]), g["obs"]
    net, _ = make_net("dgn_r", n, int(g["weight_seed"]) + 2)
    with torch.no_grad():
        logits, state = net(obs, state=None)
    np.testing.assert_allclose(logits.cpu().numpy(), g["dgnr_logits"], atol=TOL, rtol=0)
    xcat = net.hip_tap(1, obs.shape[0]).cpu().numpy()
    np.testing.assert_allclose(xcat, np.concatenate([g["dgnr_x_1"], g["dgnr_x_2"], g["dgnr_x_3"]], axis=1),
                               atol=TOL, rtol=0)


@pytest.mark.parametrize("agg", ["max", "mean", "add"])
@pytest.mark.parametrize("path", GOLDENS, ids=[os.path.basename(p)[11:-4] for p in GOLDENS])
def test_hldgn_matches_golden(path, agg):
    g = np.load(path)
    n, obs = int(g["n"]), g["obs"]
    net, _ = make_net("hl_dgn", n, int(g["weight_seed"]) + 1, agg=agg)
    with torch.no_grad():
        logits, state = net(torch.from_numpy(obs).cuda(), state="kept")
    assert state == "kept"
    np.testing.assert_allclose(logits.cpu().numpy(), g[f"hldgn_{agg}_logits"], atol=TOL, rtol=0)
    pooled = net.hip_tap(1, obs.shape[0]).cpu().numpy()
    np.testing.assert_allclose(pooled, g[f"hldgn_{agg}_pooled"], atol=TOL, rtol=0)


@pytest.mark.parametrize("model", ["l_dgn", "hl_dgn", "dgn_r"])
# 9000 > 8192: separate scan launch; 2048 / 4800 / 9000 rows: the heads' first layer is cut along K in 2 / 3 / 4
@pytest.mark.parametrize("n,bs", [(20, 256), (50, 300), (64, 37), (1, 5), (7, 1), (5, 9000), (5, 2048), (5, 4800)])
def test_matches_oracle_on_random_batches(model, n, bs):
    """Fresh seeded inputs at sizes the oracle finishes in seconds (incl. ragged / tiny / max-N cases)."""
    from oracle import net_oracle as no
    rng = np.random.RandomState(100 + n + bs)
    obs = np.zeros((bs, 8 * n + 1), dtype=np.float32)
    m = obs[:, :-1].reshape(bs, n, 8)
    m[:, :, 0:2] = rng.uniform(0, 1, size=(bs, n, 2))
    m[:, :, 2] = rng.randint(0, 9, size=(bs, n))
    m[:, :, 3] = rng.randint(0, 4, size=(bs, n))
    m[:, :, 4:7] = rng.randint(0, 2, size=(bs, n, 3))
    m[:, :, 7] = (rng.uniform(size=(bs, n)) > 0.1)
    obs[:, -1] = rng.randint(-1, n + 1, size=bs)        # also exercises the clamp (common.py:63)
    net, sd = make_net(model, n, seed=31)
    with torch.no_grad():
        got = net(obs)[0].cpu().numpy()
        torch.set_num_threads(8)
        fwd = {"l_dgn": no.ldgn_forward, "hl_dgn": no.hldgn_forward, "dgn_r": no.dgnr_forward}[model]
        want = fwd(sd, obs, n).numpy()
    np.testing.assert_allclose(got, want, atol=TOL, rtol=0)


@pytest.mark.parametrize("hidden", [(64,), (128, 64), (128, 128, 64), (128, 128)])
@pytest.mark.parametrize("bs", [40, 700])
def test_ldgn_head_shapes(hidden, bs):
    """Dueling heads other than the CLI default [128, 128] (tianshou MLP, any depth): the first layer still goes through the
    split-K launch + plane sum, the rest through the generic hidden-layer launches instead of the fused finish kernel;
    (128, 128) is the fused case at the same inputs."""
    from melissa_amd.networks import LDGNNetwork
    from oracle import net_oracle as no
    n = 20
    rng = np.random.RandomState(7 + bs)
    obs = np.zeros((bs, 8 * n + 1), dtype=np.float32)
    m = obs[:, :-1].reshape(bs, n, 8)
    m[:, :, 0:2] = rng.uniform(0, 1, size=(bs, n, 2))
    m[:, :, 2:7] = rng.randint(0, 3, size=(bs, n, 5))
    m[:, :, 7] = 1.0
    obs[:, -1] = rng.randint(0, n, size=bs)
    sd = no.init_weights("l_dgn", seed=5, dueling_hidden=hidden, random_conv_bias=True)
    net = LDGNNetwork(5, 128, 2, 4, n, dueling_param=({"hidden_sizes": list(hidden)}, {"hidden_sizes": list(hidden)}),
                      device="cuda", backend="hip")
    net.load_state_dict(sd)
    with torch.no_grad():
        got = net(obs)[0].cpu().numpy()
        want = no.ldgn_forward(sd, obs, n).numpy()
    np.testing.assert_allclose(got, want, atol=TOL, rtol=0)


@pytest.mark.parametrize("n_actions", [3, 5])
def test_ldgn_action_counts(n_actions):
    """More than the reference's two actions: up to four stay on the fused finish kernel, five take the generic tail."""
    from melissa_amd.networks import LDGNNetwork
    from oracle import net_oracle as no
    n, bs = 20, 333
    rng = np.random.RandomState(3 + n_actions)
    obs = np.zeros((bs, 8 * n + 1), dtype=np.float32)
    m = obs[:, :-1].reshape(bs, n, 8)
    m[:, :, 0:2] = rng.uniform(0, 1, size=(bs, n, 2))
    m[:, :, 2:7] = rng.randint(0, 3, size=(bs, n, 5))
    m[:, :, 7] = 1.0
    obs[:, -1] = rng.randint(0, n, size=bs)
    sd = no.init_weights("l_dgn", seed=11, n_actions=n_actions, random_conv_bias=True)
    net = LDGNNetwork(5, 128, n_actions, 4, n, dueling_param=DUEL(), device="cuda", backend="hip")
    net.load_state_dict(sd)
    with torch.no_grad():
        got = net(obs)[0].cpu().numpy()
        want = no.ldgn_forward(sd, obs, n).numpy()
    assert got.shape == (bs, n_actions)
    np.testing.assert_allclose(got, want, atol=TOL, rtol=0)


def test_full_size_linearity_property():
    """BASELINE size (N=50, 1024 rows): property checks that need no oracle run - row independence
    (a row's logits do not depend on its batch mates / position) and determinism."""
    n, bs = 50, 1024
    rng = np.random.RandomState(3)
    obs = np.zeros((bs, 8 * n + 1), dtype=np.float32)
    m = obs[:, :-1].reshape(bs, n, 8)
    m[:, :, 0:2] = rng.uniform(0, 1, size=(bs, n, 2))
    m[:, :, 2:7] = rng.randint(0, 3, size=(bs, n, 5))
    m[:, :, 7] = 1
    obs[:, -1] = rng.randint(0, n, size=bs)
    for model, dtype in (("l_dgn", "f32"), ("hl_dgn", "f32"), ("l_dgn", "f32a")):
        net, _ = make_net(model, n, seed=9)
        net.set_feature_dtype(dtype)
        t = torch.from_numpy(obs).cuda()
        with torch.no_grad():
            a = net(t)[0].clone()
            b = net(t)[0].clone()
            perm = torch.randperm(bs, device="cuda")
            c = net(t[perm])[0]
            d = net(t[:77])[0]
        assert torch.equal(a, b)
        assert torch.equal(a[perm], c)
        if dtype == "f32":
            assert torch.equal(a[:77], d)
        else:       # "f32a" picks each launch's arithmetic by its size: 1024 rows and 77 rows take different (fp32-accurate) kernels
            assert float((a[:77] - d).abs().max()) <= 1e-5
        assert torch.isfinite(a).all()


def test_shape_errors_and_no_fallback():
    from melissa_amd.networks import LDGNNetwork
    net = LDGNNetwork(5, 128, 2, 4, 20, dueling_param=DUEL(), device="cuda", backend="hip")
    with pytest.raises(ValueError, match="Expected obs to be 2D"):
        net(np.zeros(161, np.float32))
    with pytest.raises(ValueError, match="feature cols for nodes"):
        net(np.zeros((2, 160), np.float32))
    with pytest.raises(RuntimeError, match="no CPU"):
        net.hip_forward(torch.zeros(2, 161))
    empty, _ = net(np.zeros((0, 161), np.float32))             # empty batch: empty logits, no launch
    assert empty.shape == (0, 2) and empty.is_cuda


@pytest.mark.parametrize("model", ["l_dgn", "hl_dgn", "dgn_r"])
def test_out_linear_head_matches_oracle(model):
    """dueling_param=None: the single out_linear head (l_dgn.py:90,149; hl_dgn.py:80,117; dgn_r.py:80,127) against the
    oracle's out_linear branch (not the product's own torch formulation)."""
    from melissa_amd.networks import DGNRNetwork, HLDGNNetwork, LDGNNetwork
    from oracle import net_oracle as no
    for path in GOLDENS:
        g = np.load(path)
        n, obs = int(g["n"]), g["obs"]
        sd = no.init_weights(model, seed=23, random_conv_bias=True, dueling=False)
        cls = {"l_dgn": LDGNNetwork, "hl_dgn": HLDGNNetwork, "dgn_r": DGNRNetwork}[model]
        kw = dict(aggregator="max") if model == "hl_dgn" else {}
        net = cls(5, 128, 2, 4, n, dueling_param=None, device="cuda", backend="hip", **kw)
        net.load_state_dict(sd)
        fwd = {"l_dgn": no.ldgn_forward, "hl_dgn": no.hldgn_forward, "dgn_r": no.dgnr_forward}[model]
        with torch.no_grad():
            got = net(obs)[0].cpu().numpy()
            want = fwd(sd, obs, n).numpy()
        np.testing.assert_allclose(got, want, atol=TOL, rtol=0)


def test_non_dueling_head_and_select_action():
    import ctypes as C
    from melissa_amd import _lib
    from melissa_amd.networks import HLDGNNetwork
    from oracle import net_oracle as no
    n = 20
    g = np.load(golden_named("n20"))
    obs = g["obs"]
    assert int(g["n"]) == n
    sd = no.init_weights("hl_dgn", seed=0, random_conv_bias=True, dueling=False)
    net = HLDGNNetwork(5, 128, 2, 4, n, aggregator="max", dueling_param=None, device="cuda", backend="hip")
    net.load_state_dict(sd)
    with torch.no_grad():
        got = net(obs)[0]
        want = no.hldgn_forward(sd, obs, n)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=TOL, rtol=0)
    # DQN masking + argmax + eps-greedy
    lib = _lib.load()
    bs = got.shape[0]
    mask = torch.ones(bs, 2, dtype=torch.uint8, device="cuda")
    mask[::3] = 0
    act = torch.empty(bs, dtype=torch.int32, device="cuda")
    scratch = torch.empty(64, dtype=torch.float32, device="cuda")
    _lib.check(lib.mel_select_action(got.data_ptr(), mask.data_ptr(), bs, 2, C.c_float(0.0), None, None,
                                     act.data_ptr(), scratch.data_ptr(), _lib.current_stream_ptr()))
    want_act = no.dqn_act(got.cpu(), mask.cpu().numpy())
    assert act.cpu().tolist() == want_act.tolist()
    ru = torch.rand(bs, device="cuda")
    rq = torch.rand(bs, 2, device="cuda")
    _lib.check(lib.mel_select_action(got.data_ptr(), None, bs, 2, C.c_float(0.5), ru.data_ptr(), rq.data_ptr(),
                                     act.data_ptr(), scratch.data_ptr(), _lib.current_stream_ptr()))
    base = got.argmax(1)
    expect = torch.where(ru < 0.5, rq.argmax(1), base)
    assert act.cpu().tolist() == expect.cpu().tolist()


def _distinct_uniforms(rng, bs, na):
    """fp32 uniforms on a 2^-16 grid, distinct inside a row, so that ``q + mask`` is exact in fp32 and in the oracle's float64
    alike and no row has a tie (the reference draws float64; a tie there has probability ~0)."""
    base = rng.randint(0, 1 << 16, size=(bs, na)).astype(np.float64) / (1 << 16)
    return (base + np.arange(na)[None, :] / float(1 << 20)).astype(np.float32)


@pytest.mark.parametrize("bs,na", [(1, 2), (257, 2), (1000, 4), (4097, 8)])
def test_select_action_mask_and_eps_match_oracle(bs, na):
    """N10: [3P] tianshou DQNPolicy.forward (mask illegal actions with the batch-wide min - max - 1, argmax) and
    exploration_noise (eps-greedy over ``rand + mask``), reached through shared_policy.py:154 and :81-91 - mel_select_action
    against oracle.dqn_act + oracle.dqn_exploration_noise on the same logits, masks and uniform draws.  Masks: random per
    action, rows with a single legal action, rows with NO legal action (the env's dead-agent mask [0, 0], graph.py:190-192)."""
    import ctypes as C
    from melissa_amd import _lib
    from oracle import net_oracle as no
    lib = _lib.load()
    rng = np.random.RandomState(1000 + bs + na)
    logits_np = rng.standard_normal((bs, na)).astype(np.float32)
    logits_np[rng.uniform(size=bs) < 0.1] = 0.25            # ties: argmax takes the first maximum
    mask_np = (rng.uniform(size=(bs, na)) < 0.6).astype(np.uint8)
    mask_np[::5] = 0                                       # no legal action
    mask_np[1::5] = 0
    mask_np[1::5, (np.arange(len(mask_np[1::5])) % na)] = 0
    one = np.arange(bs)[1::5]
    mask_np[one, one % na] = 1                             # exactly one legal action
    mask_np[2::5] = 1                                      # all legal
    logits = torch.from_numpy(logits_np).cuda()
    mask = torch.from_numpy(mask_np).cuda()
    act = torch.empty(bs, dtype=torch.int32, device="cuda")
    scratch = torch.empty(64, dtype=torch.float32, device="cuda")
    ru_np = rng.uniform(size=bs).astype(np.float32)
    rq_np = _distinct_uniforms(rng, bs, na)
    ru, rq = torch.from_numpy(ru_np).cuda(), torch.from_numpy(rq_np).cuda()
    for use_mask in (True, False):
        m_dev = mask.data_ptr() if use_mask else None
        m_np = mask_np if use_mask else None
        greedy = no.dqn_act(torch.from_numpy(logits_np), m_np).numpy()
        _lib.check(lib.mel_select_action(logits.data_ptr(), m_dev, bs, na, C.c_float(0.0), None, None, act.data_ptr(),
                                         scratch.data_ptr(), _lib.current_stream_ptr()))
        np.testing.assert_array_equal(act.cpu().numpy(), greedy)
        if use_mask:                                       # an illegal action is never taken where a legal one exists
            legal_rows = mask_np.any(axis=1)
            assert (mask_np[np.arange(bs), greedy][legal_rows] == 1).all()
        for eps in (0.3, 1.0):
            want = no.dqn_exploration_noise(greedy, eps, ru_np, rq_np, m_np)
            _lib.check(lib.mel_select_action(logits.data_ptr(), m_dev, bs, na, C.c_float(eps), ru.data_ptr(), rq.data_ptr(),
                                             act.data_ptr(), scratch.data_ptr(), _lib.current_stream_ptr()))
            np.testing.assert_array_equal(act.cpu().numpy(), want)
            if bs > 100:
                assert (want != greedy).any()              # the noise really replaced actions


def _mix32(x):
    """The library's documented exploration stream (include/melissa_hip.h, mel_select_action_rows): lowbias32."""
    x = np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & 0xFFFFFFFF
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & 0xFFFFFFFF
    x ^= x >> np.uint64(16)
    return x


def exploration_stream(seed, step, keys, na):
    """(rand_u [rows], rand_q [rows, na]) of the counter-based stream for the given row keys."""
    keys = np.asarray(keys, dtype=np.uint64)
    base = _mix32(np.uint64(seed) ^ _mix32((np.uint64(step) * np.uint64(0x9E3779B9) + keys) & 0xFFFFFFFF))
    u01 = lambda h: (h >> np.uint64(8)).astype(np.float64) / 16777216.0
    rq = np.stack([u01(_mix32((base + np.uint64(0x85EBCA6B) * np.uint64(a + 1)) & 0xFFFFFFFF)) for a in range(na)], axis=1)
    return u01(base), rq


@pytest.mark.parametrize("model", ["l_dgn", "dgn_r"])
def test_fused_selection_follows_the_oracle_rule_on_the_documented_stream(model):
    """The argmax / eps-greedy fused into the launch that produces the logits (head_finish_kernel, mel_select of
    mel_ldgn_forward_agents) and the separate mel_select_action_rows: both must equal oracle.dqn_act +
    oracle.dqn_exploration_noise fed with the library's documented counter-based stream (restated above)."""
    import ctypes as C
    from melissa_amd import _lib
    from tests.test_gpu_round import add_agent, agent_masks
    from oracle import net_oracle as no
    n, bs = 20, 200
    g = np.load(golden_named("n20"))
    rng = np.random.RandomState(77)
    mat = np.repeat(g["obs"][:, :-1], -(-bs // g["obs"].shape[0]), axis=0)[:bs].copy()
    mat.reshape(bs, n, 8)[:, :, 0:2] += rng.uniform(-0.05, 0.05, size=(bs, n, 2)).astype(np.float32)
    member = rng.uniform(size=(bs, n)) < 0.25
    member[::9] = False
    masks = agent_masks(bs, n)
    for b, a in zip(*np.nonzero(member)):
        add_agent(masks, int(b), int(a))
    rows = int(member.sum())
    net, sd = make_net(model, n, seed=3)
    obs = torch.from_numpy(mat).cuda()
    am = torch.from_numpy(masks.view(np.int64)).cuda()
    rounds = torch.tensor([41], dtype=torch.int32, device="cuda")
    lib = _lib.load()
    envs, agents = np.nonzero(member)
    for eps in (0.0, 0.35, 1.0):
        act = torch.full((bs * n,), -7, dtype=torch.int32, device="cuda")
        sel = _lib.MelSelect()
        sel.act, sel.eps, sel.seed, sel.step_dev = act.data_ptr(), eps, 991, rounds.data_ptr()
        with torch.no_grad():
            logits, offsets = net.hip_forward_agents(obs, am, bs * n, select=sel)
        torch.cuda.synchronize()
        assert int(offsets[-1]) == rows
        got_logits = logits[:rows].cpu().numpy()
        fwd = no.ldgn_forward if model == "l_dgn" else no.dgnr_forward
        want_logits = fwd(sd, np.concatenate([mat[envs], agents[:, None].astype(np.float32)], axis=1), n).numpy()
        np.testing.assert_allclose(got_logits, want_logits, atol=TOL, rtol=0)
        greedy = no.dqn_act(torch.from_numpy(got_logits)).numpy()
        ru, rq = exploration_stream(991, 41, np.arange(rows), 2)
        want = no.dqn_exploration_noise(greedy, eps, ru, rq)
        np.testing.assert_array_equal(act[:rows].cpu().numpy(), want)
        assert (act[rows:] == -7).all()
        sep = torch.full((rows,), -7, dtype=torch.int32, device="cuda")
        _lib.check(lib.mel_select_action_rows(logits.data_ptr(), None, rows, None, 2, C.c_float(eps), 991, 0, rounds.data_ptr(),
                                              sep.data_ptr(), _lib.current_stream_ptr()))
        np.testing.assert_array_equal(sep.cpu().numpy(), want)
        if eps == 0.35:
            assert 0.15 < (ru < eps).mean() < 0.55 and (want != greedy).any()


@pytest.mark.parametrize("model", ["l_dgn", "hl_dgn", "dgn_r"])
@pytest.mark.parametrize("n,bs", [(20, 256), (50, 300), (64, 37), (7, 1)])
def test_split_precision_meets_the_fp32_bar(model, n, bs):
    """MEL_PREC_F32_SPLIT ("f32s"): fp32 features, projections on the bf16 matrix cores by exact operand splitting
    (csrc/gemm_split.hpp).  Same bar as the native fp32 path: logits within 1e-4 of the oracle; measured error is
    reported next to the native path's."""
    from oracle import net_oracle as no
    rng = np.random.RandomState(400 + n + bs)
    obs = np.zeros((bs, 8 * n + 1), dtype=np.float32)
    m = obs[:, :-1].reshape(bs, n, 8)
    m[:, :, 0:2] = rng.uniform(0, 1, size=(bs, n, 2))
    m[:, :, 2] = rng.randint(0, 9, size=(bs, n))
    m[:, :, 3] = rng.randint(0, 4, size=(bs, n))
    m[:, :, 4:7] = rng.randint(0, 2, size=(bs, n, 3))
    m[:, :, 7] = (rng.uniform(size=(bs, n)) > 0.1)
    obs[:, -1] = rng.randint(0, n, size=bs)
    net, sd = make_net(model, n, seed=31)
    with torch.no_grad():
        native = net(obs)[0].cpu().numpy()
        net.set_feature_dtype("f32s")
        got = net(obs)[0].cpu().numpy()
        torch.set_num_threads(8)
        fwd = {"l_dgn": no.ldgn_forward, "hl_dgn": no.hldgn_forward, "dgn_r": no.dgnr_forward}[model]
        want = fwd(sd, obs, n).numpy()
    print(f"{model} N={n}: max |logit error| split {np.abs(got - want).max():.2e}, native fp32 MFMA {np.abs(native - want).max():.2e}")
    np.testing.assert_allclose(got, want, atol=TOL, rtol=0)
    assert not np.array_equal(got, native) or bs == 1          # a different summation: the split path really ran


@pytest.mark.parametrize("path", GOLDENS, ids=[os.path.basename(p)[11:-4] for p in GOLDENS])
def test_split_precision_matches_golden(path):
    g = np.load(path)
    n, obs = int(g["n"]), g["obs"]
    net, _ = make_net("l_dgn", n, int(g["weight_seed"]))
    net.set_feature_dtype("f32s")
    with torch.no_grad():
        logits, _ = net(obs)
    np.testing.assert_allclose(logits.cpu().numpy(), g["ldgn_logits"], atol=TOL, rtol=0)
    xcat = net.hip_tap(1, obs.shape[0]).cpu().numpy()
    np.testing.assert_allclose(xcat, np.concatenate([g["ldgn_x_1"], g["ldgn_x_2"], g["ldgn_x_3"]], axis=1), atol=TOL, rtol=0)
