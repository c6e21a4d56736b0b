"""GPU tests of the learn-path kernels (csrc/grad.hip): forward and hand-written backward of
the GATv2 / TransformerConv edge softmax + aggregation and of the graph pool, through torch.autograd.Function,
against torch autograd THROUGH THE ORACLE (oracle/net_oracle.py is plain differentiable torch: the weights of the
restatement are made leaves and the same loss is back-propagated on the CPU).
Tolerance: forward 1e-4 absolute (the inference bar); gradients 2e-4 relative to the largest gradient entry of the
tensor (the sums run in another order than the CPU oracle's, which itself rounds differently)."""
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
    from melissa_amd.env.episodes import sets_to_bool
    for n, bs in [(20, 33), (50, 64), (64, 5), (1, 3), (100, 9), (128, 4), (65, 3)]:
        obs = torch.from_numpy(random_obs(n, bs, n + bs)).cuda()
        if n in (64, 128):
            obs[:, :-1].view(bs, n, 8)[:, :, 0:2] *= 0.2          # dense clique: exercises the 32-neighbour cap
        adj = radius_graph(obs, n, 5).view(bs, n, -1).squeeze(-1).cpu().numpy().view(np.uint64)
        want = radius_adjacency(obs[:, :-1].view(bs, n, 8)[:, :, :2]).cpu().numpy()          # [bs, i, j]
        got = sets_to_bool(adj, n)
        np.testing.assert_array_equal(got, want)


def oracle_loss_and_grads(model, agg, sd, obs_np, n, act_np, target_np, dueling=True):
    """loss = mean((Q_oracle(obs)[act] - target)^2) and d loss / d every weight, by torch autograd on the CPU oracle."""
    from oracle import net_oracle as no
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    torch.set_num_threads(8)
    if model == "l_dgn":
        logits = no.ldgn_forward(leaves, obs_np, n)
    elif model == "dgn_r":
        logits = no.dgnr_forward(leaves, obs_np, n)
    else:
        logits = no.hldgn_forward(leaves, obs_np, n, aggregator=agg)
    bs = obs_np.shape[0]
    loss = (logits[torch.arange(bs), torch.from_numpy(act_np)] - torch.from_numpy(target_np)).pow(2).mean()
    loss.backward()
    return logits.detach(), float(loss.detach()), {k: v.grad for k, v in leaves.items()}


@pytest.mark.parametrize("model,agg", [("l_dgn", "max"), ("dgn_r", "max"), ("hl_dgn", "max"), ("hl_dgn", "mean"), ("hl_dgn", "add")])
@pytest.mark.parametrize("n,bs", [(20, 48), (50, 16), (7, 3), (100, 6)])
def test_hip_autograd_matches_oracle_autograd(model, agg, n, bs):
    from oracle import net_oracle as no
    obs_np = random_obs(n, bs, 300 + n)
    act_np = np.random.RandomState(5).randint(0, 2, bs)
    target_np = np.random.RandomState(6).uniform(-1, 1, bs).astype(np.float32)
    sd = no.init_weights(model, seed=17, random_conv_bias=True)
    want_logits, want_loss, want = oracle_loss_and_grads(model, agg, sd, obs_np, n, act_np, target_np)
    net = make(model, n, agg)
    net.learn_kernels = "hip"
    obs = torch.from_numpy(obs_np).cuda()
    logits = net.torch_forward(obs)
    loss = (logits[torch.arange(bs), torch.from_numpy(act_np).cuda()] - torch.from_numpy(target_np).cuda()).pow(2).mean()
    loss.backward()
    torch.testing.assert_close(logits.detach().cpu(), want_logits, atol=1e-4, rtol=0)
    assert abs(float(loss.detach()) - want_loss) <= 1e-4 * max(1.0, abs(want_loss))
    checked = 0
    for k, p in net.named_parameters():
        gd = want[k]
        gh = p.grad
        if gd is None or float(gd.abs().max()) == 0.0:              # lin_skip (unused by TransformerConv), V bias under mean...
            assert gh is None or float(gh.abs().max()) <= 1e-7, k
            continue
        scale = float(gd.abs().max())
        assert gh is not None, k
        err = float((gh.cpu() - gd).abs().max())
        assert err <= 2e-4 * scale + 1e-7, (k, err, scale)
        checked += 1
    assert checked >= 10


@pytest.mark.parametrize("model", ["l_dgn", "hl_dgn"])
def test_out_linear_head_gradients_match_oracle(model):
    """dueling_param=None (one out_linear head): forward and gradients of the learn path against the oracle's branch."""
    from melissa_amd.networks import HLDGNNetwork, LDGNNetwork
    from oracle import net_oracle as no
    n, bs = 20, 24
    obs_np = random_obs(n, bs, 77)
    act_np = np.random.RandomState(1).randint(0, 2, bs)
    target_np = np.random.RandomState(2).uniform(-1, 1, bs).astype(np.float32)
    sd = no.init_weights(model, seed=3, random_conv_bias=True, dueling=False)
    want_logits, want_loss, want = oracle_loss_and_grads(model, "max", sd, obs_np, n, act_np, target_np)
    kw = dict(aggregator="max") if model == "hl_dgn" else {}
    net = (LDGNNetwork if model == "l_dgn" else HLDGNNetwork)(5, 128, 2, 4, n, dueling_param=None, device="cuda",
                                                              backend="torch", **kw)
    net.load_state_dict(sd)
    logits = net.torch_forward(torch.from_numpy(obs_np).cuda())
    loss = (logits[torch.arange(bs), torch.from_numpy(act_np).cuda()] - torch.from_numpy(target_np).cuda()).pow(2).mean()
    loss.backward()
    torch.testing.assert_close(logits.detach().cpu(), want_logits, atol=1e-4, rtol=0)
    for k, p in net.named_parameters():
        scale = float(want[k].abs().max())
        assert float((p.grad.cpu() - want[k]).abs().max()) <= 2e-4 * scale + 1e-7, k


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


@pytest.mark.parametrize("m,k,n", [(37, 128, 512), (1600, 512, 512), (1, 1152, 128), (250, 128, 128), (64, 640, 256),
                                   (33600, 128, 128), (5003, 256, 128), (2048, 128, 640)])     # long batches: split-K dW
def test_hip_linear_matches_torch_linear(m, k, n):
    """The learn path's dense layers on the library's own GEMM (autograd_ops.hip_linear: y = x W^T + b, dX = dY W,
    dW = dY^T X through mel_gemm_f32 + mel_transpose_f32) against float64 torch on the CPU."""
    from melissa_amd.networks.autograd_ops import hip_linear
    g = torch.Generator().manual_seed(m + k + n)
    x64 = torch.randn(m, k, generator=g, dtype=torch.float64)
    w64 = torch.randn(n, k, generator=g, dtype=torch.float64) / k ** 0.5
    b64 = torch.randn(n, generator=g, dtype=torch.float64)
    dy64 = torch.randn(m, n, generator=g, dtype=torch.float64)
    leaves64 = [t.clone().requires_grad_(True) for t in (x64, w64, b64)]
    torch.nn.functional.linear(*leaves64).backward(dy64)
    leaves = [t.float().cuda().requires_grad_(True) for t in (x64, w64, b64)]
    y = hip_linear(*leaves)
    y.backward(dy64.float().cuda())
    want_y = torch.nn.functional.linear(x64, w64, b64)
    assert float((y.detach().cpu().double() - want_y).abs().max()) <= 2e-5 * max(1.0, float(want_y.abs().max()))
    for got, want in zip(leaves, leaves64):
        scale = float(want.grad.abs().max())
        assert float((got.grad.cpu().double() - want.grad).abs().max()) <= 1e-5 * scale + 1e-7


def test_learn_path_dense_layers_run_on_the_library_gemm(monkeypatch):
    """torch_forward on a ROCm device: every projection the GEMM's tiling takes goes through mel_gemm_f32 (no library
    GEMM); only the 5-wide encoder input and the 2- / 1-wide last head layers are left to F.linear."""
    import torch.nn.functional as F
    seen = []
    real = F.linear
    monkeypatch.setattr(F, "linear", lambda x, w, b=None: (seen.append(tuple(w.shape)), real(x, w, b))[1])
    for model in ("l_dgn", "dgn_r", "hl_dgn"):
        seen.clear()
        net = make(model, 20)
        obs = torch.from_numpy(random_obs(20, 8, 3)).cuda()
        net.torch_forward(obs).sum().backward()
        assert seen and all(s[1] == 5 or s[0] <= 2 for s in seen), (model, seen)
        assert all(p.grad is not None for n_, p in net.named_parameters() if "lin_skip" not in n_)


@pytest.mark.gpu
@pytest.mark.parametrize("weight_decay", [0.0, 1e-2])
def test_fused_adam_step_is_torch_adam(weight_decay):
    """melissa_amd.optim.adam_step: torch.optim.Adam's update on torch's own optimizer state in one launch - against
    torch.optim.Adam itself over eight updates on the same gradients, in the eager form (host step counters) and in the
    capturable form (device step counters, the one a HIP-graph capture uses), and the state a twin optimizer loads from it."""
    import copy
    from melissa_amd.optim import adam_step
    torch.manual_seed(4)
    shapes = [(128, 5), (128,), (512, 128), (4, 128), (128, 1152), (1,), (3, 128)]
    base = [torch.randn(*s, device="cuda") for s in shapes]

    def make(capturable):
        ps = [torch.nn.Parameter(b.clone()) for b in base]
        return ps, torch.optim.Adam(ps, lr=1e-3, weight_decay=weight_decay, capturable=capturable)

    ref_p, ref = make(False)
    ref._mel_disable_fused = True
    runs = [make(False), make(True)]
    for step in range(8):
        grads = [torch.randn(*s, device="cuda") * (0.1 if step % 2 else 10.0) for s in shapes]
        for ps, opt in [(ref_p, ref)] + runs:
            for p, g in zip(ps, grads):
                p.grad = g.clone()
        ref.step()
        for ps, opt in runs:
            adam_step(opt)                       # (the first call is torch's own step: it creates the state)
        for ps, opt in runs:
            for p, q in zip(ps, ref_p):
                assert float((p - q).detach().abs().max()) <= 2e-7 * max(1.0, float(q.detach().abs().max())), step
    for ps, opt in runs:
        for p, q in zip(ps, ref_p):
            np.testing.assert_allclose(opt.state[p]["exp_avg"].cpu().numpy(), ref.state[q]["exp_avg"].cpu().numpy(), rtol=1e-5, atol=2e-6)
            np.testing.assert_allclose(opt.state[p]["exp_avg_sq"].cpu().numpy(), ref.state[q]["exp_avg_sq"].cpu().numpy(), rtol=1e-5, atol=1e-5)
            assert float(opt.state[p]["step"]) == 8.0
    twin_p, twin = make(False)
    twin.load_state_dict(copy.deepcopy(runs[0][1].state_dict()))
    assert all(float(twin.state[p]["step"]) == 8.0 for p in twin_p)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,out_f,in_f", [(1600, 128, 128), (32, 128, 1152), (1600, 512, 128), (96, 64, 64), (33, 128, 64)])
def test_linear_backward_products_without_transposed_copies(rows, out_f, in_f):
    """mel_gemm_f32_t: dX = dY W and dW = dY^T X with the operands read in place.  Bit-identical to the library's own GEMM on
    transposed copies (same tile, K order and MFMA sequence) and equal to torch's products within fp32 rounding; hip_linear's
    backward uses it (rows % 32 != 0 keeps the padded-copy form for dW)."""
    from melissa_amd.networks import autograd_ops as ops
    torch.manual_seed(rows + out_f)
    x = torch.randn(rows, in_f, device="cuda")
    w = torch.randn(out_f, in_f, device="cuda") / in_f ** 0.5
    dy = torch.randn(rows, out_f, device="cuda")
    dx = ops._gemm_t(dy, False, w, torch.empty_like(x), rows, in_f, out_f)
    want_dx = ops._gemm(dy, ops._transpose(w), None, torch.empty_like(x))
    assert torch.equal(dx, want_dx)
    np.testing.assert_allclose(dx.cpu().numpy(), (dy.double() @ w.double()).float().cpu().numpy(), rtol=1e-4, atol=1e-4)
    if rows % 32 == 0:
        dw = ops._gemm_t(dy, True, x, torch.empty_like(w), out_f, in_f, rows)
        want_dw = ops._gemm(ops._transpose(dy, 32), ops._transpose(x, 32), None, torch.empty_like(w))
        assert torch.equal(dw, want_dw)
        np.testing.assert_allclose(dw.cpu().numpy(), (dy.double().t() @ x.double()).float().cpu().numpy(), rtol=1e-4, atol=2e-4)
    xg = x.clone().requires_grad_(True)
    wg = w.clone().requires_grad_(True)
    ops.hip_linear(xg, wg, None, True).backward(dy)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    torch.relu(torch.nn.functional.linear(xr, wr)).backward(dy)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.cpu().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(wg.grad.cpu().numpy(), wr.grad.cpu().numpy(), rtol=1e-4, atol=2e-4)
