"""GPU tests of the learn-path kernels (csrc/grad.hip): forward and hand-written backward of
the GATv2 / TransformerConv edge softmax + aggregation and of the graph pool, through torch.autograd.Function,
against the dense torch formulation of the same networks (which the CPU suite pins to the oracle).
Tolerance: forward 1e-4 absolute (the inference bar); gradients 2e-4 relative to the largest gradient entry of the
tensor (fp32 atomics reorder sums; the dense path itself rounds differently)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DUEL = lambda: ({"hidden_sizes": [128, 128]}, {"hidden_sizes": [128, 128]})


def random_obs(n, bs, seed):
    rng = np.random.RandomState(seed)
    obs = np.zeros((bs, 8 * n + 1), dtype=np.float32)
    m = obs[:, :-1].reshape(bs, n, 8)
    m[:, :, 0:2] = rng.uniform(0, 1, size=(bs, n, 2))
    m[:, :, 2] = rng.randint(0, 9, size=(bs, n))
    m[:, :, 3] = rng.randint(0, 4, size=(bs, n))
    m[:, :, 4:7] = rng.randint(0, 2, size=(bs, n, 3))
    m[:, :, 7] = (rng.uniform(size=(bs, n)) > 0.2)
    obs[:, -1] = rng.randint(0, n, size=bs)
    return obs


def make(model, n, agg="max"):
    from melissa_amd.networks import DGNRNetwork, HLDGNNetwork, LDGNNetwork
    from oracle import net_oracle as no
    sd = no.init_weights(model, seed=17, random_conv_bias=True)
    if model == "dgn_r":
        net = DGNRNetwork(5, 128, 2, 4, n, dueling_param=DUEL(), device="cuda", backend="torch")
    elif model == "l_dgn":
        net = LDGNNetwork(5, 128, 2, 4, n, dueling_param=DUEL(), device="cuda", backend="torch")
    else:
        net = HLDGNNetwork(5, 128, 2, 4, n, aggregator=agg, dueling_param=DUEL(), device="cuda", backend="torch")
    net.load_state_dict(sd)
    return net


def test_radius_graph_matches_dense_rule():
    from melissa_amd.networks.autograd_ops import radius_graph
    from melissa_amd.networks.common import radius_adjacency
    for n, bs in [(20, 33), (50, 64), (64, 5), (1, 3)]:
        obs = torch.from_numpy(random_obs(n, bs, n + bs)).cuda()
        if n == 64:
            obs[:, :-1].view(bs, n, 8)[:, :, 0:2] *= 0.2          # dense clique: exercises the 32-neighbour cap
        adj = radius_graph(obs, n, 5).view(bs, n).cpu().numpy().view(np.uint64)
        want = radius_adjacency(obs[:, :-1].view(bs, n, 8)[:, :, :2]).cpu().numpy()          # [bs, i, j]
        got = ((adj[:, :, None] >> np.arange(n, dtype=np.uint64)) & np.uint64(1)).astype(bool)
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("model,agg", [("l_dgn", "max"), ("dgn_r", "max"), ("hl_dgn", "max"), ("hl_dgn", "mean"), ("hl_dgn", "add")])
@pytest.mark.parametrize("n,bs", [(20, 48), (50, 16), (7, 3)])
def test_hip_autograd_matches_dense_formulation(model, agg, n, bs):
    obs = torch.from_numpy(random_obs(n, bs, 300 + n)).cuda()
    act = torch.from_numpy(np.random.RandomState(5).randint(0, 2, bs)).cuda()
    target = torch.from_numpy(np.random.RandomState(6).uniform(-1, 1, bs).astype(np.float32)).cuda()
    grads, outs = {}, {}
    for kernels in ("dense", "hip"):
        net = make(model, n, agg)
        net.learn_kernels = kernels
        logits = net.torch_forward(obs)
        loss = (logits[torch.arange(bs), act] - target).pow(2).mean()
        loss.backward()
        outs[kernels] = logits.detach()
        grads[kernels] = {k: p.grad for k, p in net.named_parameters()}
    torch.testing.assert_close(outs["hip"], outs["dense"], atol=1e-4, rtol=0)
    checked = 0
    for k, gd in grads["dense"].items():
        gh = grads["hip"][k]
        if gd is None:
            assert gh is None or float(gh.abs().max()) == 0.0, k
            continue
        scale = float(gd.abs().max())
        assert gh is not None, k
        assert float((gh - gd).abs().max()) <= 2e-4 * scale + 1e-7, (k, float((gh - gd).abs().max()), scale)
        checked += 1
    assert checked >= 10


def test_learn_step_uses_hip_kernels_by_default():
    """policy.learn on a CUDA batch goes through the HIP attention backward (no opt-in needed) and the
    parameters move."""
    from melissa_amd.policy import DQNPolicy
    n, bs = 20, 32
    net = make("l_dgn", n)
    assert getattr(net, "learn_kernels", "hip") == "hip"
    pol = DQNPolicy(net, torch.optim.Adam(net.parameters(), lr=1e-3))
    before = net.conv1.att.detach().clone()
    out = pol.learn(dict(obs=torch.from_numpy(random_obs(n, bs, 1)).cuda(), act=np.zeros(bs, dtype=np.int64),
                         returns=np.ones(bs, dtype=np.float32)))
    assert np.isfinite(out["loss"]) and not torch.equal(before, net.conv1.att.detach())
