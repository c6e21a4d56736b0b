"""GPU tests of the bf16 FEATURE PATH (BASELINE config "L-DGN 50-node, 1024 vectorised envs, bf16 feature path";
SURVEY.md 8(d) parity gates: "bf16 path: report max abs/rel error; 1e-4 is not attainable - <= 2e-2 abs with
identical argmax on >= 99.9 % of rows").

The bf16 path is NOT the reference's arithmetic: feature rows and projection weights are rounded to bf16, the
contraction accumulates in fp32, attention / softmax / biases / logits stay fp32.  Stated bounds:
  * the bf16 GEMM itself: exact products, fp32 accumulation -> equal to torch's fp32 matmul of the same bf16
    inputs up to summation order (1e-3 relative), bf16 outputs within one rounding;
  * logits vs the fp32 oracle: <= 2e-2 absolute; the greedy action agrees wherever the fp32 action gap exceeds
    twice the measured logit error (a tie inside the rounding noise may flip) and on >= 99 % of all rows.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ABS_TOL = 2e-2
DUEL = lambda: ({"hidden_sizes": [128, 128]}, {"hidden_sizes": [128, 128]})


def make_net(model, n, seed, agg="max"):
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


def random_obs(n, bs, seed):
    rng = np.random.RandomState(seed)
    obs = np.zeros((bs, 8 * n + 1), dtype=np.float32)
    m = obs[:, :-1].reshape(bs, n, 8)
    m[:, :, 0:2] = rng.uniform(0, 1, size=(bs, n, 2))
    m[:, :, 2] = rng.randint(0, 9, size=(bs, n))
    m[:, :, 3] = rng.randint(0, 4, size=(bs, n))
    m[:, :, 4:7] = rng.randint(0, 2, size=(bs, n, 3))
    m[:, :, 7] = (rng.uniform(size=(bs, n)) > 0.1)
    obs[:, -1] = rng.randint(0, n, size=bs)
    return obs


@pytest.mark.parametrize("M,N,K,relu,y_f32,tile", [(1000, 512, 512, 0, 0, 0), (77, 64, 64, 1, 1, 0), (4096, 256, 1152, 1, 0, 0),
                                                   (333, 512, 128, 0, 1, 2), (1, 64, 128, 0, 0, 0), (5000, 128, 128, 1, 0, 2),
                                                   # tile 3: the 128 x 256 kernel of the large launches (ragged rows, one row,
                                                   # several work items per workgroup, two-stage and eighteen-stage streams)
                                                   (1000, 512, 512, 1, 0, 3), (1, 256, 128, 0, 1, 3), (40000, 512, 512, 0, 0, 3),
                                                   (129, 256, 1152, 1, 1, 3), (15473, 1024, 512, 1, 0, 3)])
def test_gemm_bf16_matches_torch(M, N, K, relu, y_f32, tile):
    from melissa_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g)
    Y = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float32 if y_f32 else torch.bfloat16)
    _lib.check(lib.mel_gemm_bf16(A.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), N, M, N, K, relu, y_f32, tile,
                                 _lib.current_stream_ptr()))
    want = A.float() @ W.float().T + b
    if relu:
        want = want.relu()
    if y_f32:
        torch.testing.assert_close(Y, want, atol=1e-4, rtol=1e-4)
    else:       # one bf16 rounding (2^-8 relative) on top of the fp32 result
        torch.testing.assert_close(Y.float(), want, atol=1e-3, rtol=2 ** -7)


def test_convert_bf16_is_round_to_nearest_even():
    from melissa_amd import _lib
    lib = _lib.load()
    x = torch.randn(4096, device="cuda") * torch.logspace(-6, 6, 4096, device="cuda")
    out = torch.empty(4096, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.mel_convert_bf16(x.data_ptr(), out.data_ptr(), x.numel(), _lib.current_stream_ptr()))
    assert torch.equal(out, x.to(torch.bfloat16))
    assert lib.mel_convert_bf16(x.data_ptr(), out.data_ptr(), 12, _lib.current_stream_ptr()) != 0


def check_against_fp32(got, want, label):
    err = np.abs(got - want).max()
    gap = np.abs(want[:, 0] - want[:, 1])
    same = got.argmax(1) == want.argmax(1)
    decided = gap > 2 * max(err, 1e-6)
    print(f"{label}: max abs logit error {err:.2e}, argmax agreement {same.mean() * 100:.2f} % "
          f"({decided.mean() * 100:.1f} % of rows decided by more than twice the error)")
    assert err <= ABS_TOL
    assert same[decided].all()
    assert same.mean() >= 0.99


@pytest.mark.parametrize("model", ["l_dgn", "hl_dgn", "dgn_r"])
@pytest.mark.parametrize("n,bs", [(20, 256), (50, 300), (64, 37), (7, 1)])
def test_bf16_forward_close_to_fp32_oracle(model, n, bs):
    from oracle import net_oracle as no
    obs = random_obs(n, bs, 200 + n + bs)
    net, sd = make_net(model, n, seed=31)
    net.set_feature_dtype("bf16")
    with torch.no_grad():
        got = net(obs)[0].cpu().numpy()
        torch.set_num_threads(8)
        fwd = {"l_dgn": no.ldgn_forward, "hl_dgn": no.hldgn_forward, "dgn_r": no.dgnr_forward}[model]
        want = fwd(sd, obs, n).numpy()
    assert got.dtype == np.float32
    err = np.abs(got - want).max()
    assert 0 < err <= ABS_TOL          # > 0: the bf16 path really ran
    if bs >= 100:
        check_against_fp32(got, want, f"{model} N={n}")
    # same weights, fp32 switch back: the reference bar again
    net.set_feature_dtype("f32")
    with torch.no_grad():
        np.testing.assert_allclose(net(obs)[0].cpu().numpy(), want, atol=1e-4, rtol=0)


def test_bf16_sees_weight_updates_without_a_cache():
    """The bf16 weight copies are refreshed by every call (stateless library): an in-place parameter update
    changes the next forward."""
    obs = random_obs(20, 64, 5)
    net, _ = make_net("l_dgn", 20, seed=3)
    net.set_feature_dtype("bf16")
    with torch.no_grad():
        a = net(obs)[0].clone()
        net.conv2.lin_l.weight.mul_(1.5)
        b = net(obs)[0].clone()
        net.conv2.lin_l.weight.div_(1.5)
        c = net(obs)[0].clone()
    assert not torch.allclose(a, b)
    assert torch.equal(a, c)


def test_bf16_round_forward_matches_fp32_round_forward():
    """Agent-set forward (the round-batched loop's entry point) at BASELINE size: bf16 vs the fp32 HIP path on
    the same rows (the fp32 path is the one pinned to the oracle)."""
    n, bs = 50, 1024
    rng = np.random.RandomState(11)
    obs = random_obs(n, bs, 77)[:, :-1].copy()
    mask = np.zeros(bs, dtype=np.uint64)
    for b in range(bs):
        for a in rng.choice(n, size=rng.randint(1, 9), replace=False):
            mask[b] |= np.uint64(1) << np.uint64(a)
    net, _ = make_net("l_dgn", n, seed=9)
    t = torch.from_numpy(obs).cuda()
    m = torch.from_numpy(mask.view(np.int64)).cuda()
    rows_cap = bs * n
    with torch.no_grad():
        want, off = net.hip_forward_agents(t, m, rows_cap)
        rows = int(off[-1])
        want = want[:rows].cpu().numpy()
        net.set_feature_dtype("bf16")
        got, off2 = net.hip_forward_agents(t, m, rows_cap)
        got = got[:rows].cpu().numpy()
    assert torch.equal(off, off2)
    check_against_fp32(got, want, f"round forward, {rows} agent rows")


@pytest.mark.parametrize("dtype", ["bf16", "f32s", "f32a"])
def test_prepared_weights_follow_weight_versions(dtype):
    """The converted projection weights are prepared once per weight VERSION (mel_prepare_weights into a caller-owned
    buffer), not per call: a forward after an in-place parameter change (optimizer step, load_state_dict) must see the new
    weights, and the prepared path must give exactly what the stateless per-call conversion gives."""
    import ctypes as C
    from melissa_amd import _lib
    n, bs = 20, 64
    obs = torch.from_numpy(random_obs(n, bs, 11)).cuda()
    net, _ = make_net("l_dgn", n, seed=5)
    net.set_feature_dtype(dtype)
    with torch.no_grad():
        first = net(obs)[0].clone()
        assert net._weights().prepared and net._prepared[1].numel() >= int(_lib.load().mel_prepared_weights_bytes(C.byref(net._weights())))
        buf_id = net._prepared[1].data_ptr()
        again = net(obs)[0].clone()
        assert torch.equal(first, again) and net._prepared[1].data_ptr() == buf_id          # no re-conversion, same buffer
        # stateless path (prepared = NULL): the library converts into the workspace on every call
        w = net._weights()
        keep, w.prepared = w.prepared, None
        out = torch.empty(bs, 2, device="cuda")
        ws = torch.empty(int(_lib.load().mel_workspace_bytes(C.byref(w), bs, n)), dtype=torch.uint8, device="cuda")
        _lib.check(_lib.load().mel_ldgn_forward(C.byref(w), obs.data_ptr(), bs, n, obs.shape[1], out.data_ptr(), ws.data_ptr(),
                                                ws.numel(), _lib.current_stream_ptr()))
        w.prepared = keep
        assert torch.equal(out, first)
        # a new weight version
        net.conv2.lin_l.weight.mul_(1.5)
        changed = net(obs)[0].clone()
        fresh, _ = make_net("l_dgn", n, seed=5)
        fresh.set_feature_dtype(dtype)
        fresh.conv2.lin_l.weight.mul_(1.5)
        assert not torch.equal(changed, first) and torch.equal(changed, fresh(obs)[0])


@pytest.mark.parametrize("m,n,k,tile,ksplit", [
    (1000, 512, 512, 1, 0), (1000, 512, 512, 2, 0),          # ragged rows (1000 = 7 x 128 + 104)
    (4820, 256, 1152, 2, 3), (4820, 256, 1152, 2, 0),        # the heads' first layer, split-K and not
    (129, 128, 128, 2, 0), (1, 128, 128, 2, 0),              # one row beyond a tile; a single row
    (333, 384, 256, 2, 2), (333, 384, 256, 2, 4), (2048, 640, 160, 1, 0),
    (40000, 512, 512, 2, 0), (70000, 128, 128, 2, 0),        # several work items per workgroup; eight-step work items
    (4000, 1024, 160, 2, 0),                                 # a ten-step stream
    (1000, 512, 512, 3, 0), (129, 256, 128, 3, 0), (1, 256, 128, 3, 0),   # 128 x 256 planes kernel: ragged rows, one row beyond a
    (40000, 512, 512, 3, 0), (70000, 256, 128, 3, 0),                    # tile, a single row; several work items per
    (4000, 1536, 160, 3, 0), (15473, 1024, 512, 3, 0)])                  # workgroup; ten-step stream, widest N; DGN-R's conv2
def test_split_gemm_matches_float64(m, n, k, tile, ksplit):
    """mel_gemm_f32_split (MEL_PREC_F32_SPLIT's projections on their own): fp32 operands split exactly into three bf16
    pieces, six partial products per term on the bf16 matrix cores, fp32 accumulation - as close to the float64 product as
    an fp32 GEMM is.  The tile shapes (1: 64 x 64, 2: 128 x 128, 3: 128 x 256 fed from plane blocks on both sides -
    gemm_planes_kernel, conv2's kernel in large forwards) and the 128 x 128 kernel's split-K.  Tile 3 adds the same six
    products in the same order as tile 2: its result must be bit-identical."""
    from melissa_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g).cuda()
    w = (torch.randn(n, k, generator=g) / k ** 0.5).cuda()
    b = torch.randn(n, generator=g).cuda()
    y = torch.full((m, n), float("nan"), device="cuda")
    scratch = torch.empty(6 * n * k + 256 + 4 * max(ksplit, 1) * m * n + 6 * m * k, dtype=torch.uint8, device="cuda")
    _lib.check(lib.mel_gemm_f32_split(a.data_ptr(), k, w.data_ptr(), b.data_ptr(), y.data_ptr(), n, m, n, k, 1, tile, ksplit,
                                      scratch.data_ptr(), scratch.numel(), _lib.current_stream_ptr()))
    want = torch.relu(torch.addmm(b.double(), a.double(), w.double().t()))
    err = float((y.double() - want).abs().max())
    native = float((torch.relu(torch.addmm(b, a, w.t())).double() - want).abs().max())
    print(f"split gemm {m}x{n}x{k} tile {tile} ksplit {ksplit}: max error {err:.1e} (torch fp32 matmul: {native:.1e})")
    assert err <= 4e-6 * max(1.0, float(want.abs().max()))
    if tile == 3:
        y2 = torch.full((m, n), float("nan"), device="cuda")
        _lib.check(lib.mel_gemm_f32_split(a.data_ptr(), k, w.data_ptr(), b.data_ptr(), y2.data_ptr(), n, m, n, k, 1, 2, 0,
                                          scratch.data_ptr(), scratch.numel(), _lib.current_stream_ptr()))
        assert torch.equal(y, y2)


def test_split_gemm_rejects_shapes_the_big_tile_cannot_take():
    from melissa_amd import _lib
    lib = _lib.load()
    a = torch.zeros(64, 128, device="cuda"); w = torch.zeros(192, 128, device="cuda"); y = torch.zeros(64, 192, device="cuda")
    scratch = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    st = lib.mel_gemm_f32_split(a.data_ptr(), 128, w.data_ptr(), None, y.data_ptr(), 192, 64, 192, 128, 0, 2, 0, scratch.data_ptr(),
                                scratch.numel(), _lib.current_stream_ptr())
    assert st == _lib.ERR_UNSUPPORTED                                  # N = 192 is no multiple of 128
    st = lib.mel_gemm_f32_split(a.data_ptr(), 128, w.data_ptr(), None, y.data_ptr(), 192, 64, 192, 128, 0, 1, 0, scratch.data_ptr(),
                                16, _lib.current_stream_ptr())
    assert st == _lib.ERR_UNSUPPORTED                                  # scratch too small


def test_split_precision_at_the_benchmark_size():
    """L-DGN N = 50, 1024 envs at MEL_PREC_F32_SPLIT: conv2's projections and the heads' first layer then run on the
    128 x 128 split kernel (split-K for the heads).  Here: logits within 1e-5 of the native fp32 path on the same random
    observations; the ORACLE check at this size is tests/test_gpu_round.py::test_round_forward_at_the_benchmark_batch_matches_oracle
    (the default precision, which sends the same two launches to the same kernels)."""
    n, bs = 50, 1024
    obs = torch.from_numpy(random_obs(n, bs, 3)).cuda()
    net, _ = make_net("l_dgn", n, seed=2)
    with torch.no_grad():
        net.set_feature_dtype("f32")
        native = net(obs)[0].clone()
        net.set_feature_dtype("f32s")
        got = net(obs)[0].clone()
    err = float((got - native).abs().max())
    print(f"l_dgn N=50 bs=1024: split vs native fp32 logits {err:.2e}")
    assert err <= 1e-5 and not torch.equal(got, native)


@pytest.mark.parametrize("model", ["l_dgn", "dgn_r", "hl_dgn"])
def test_default_precision_chooses_per_launch(model):
    """The networks' default precision "f32a" (MEL_PREC_F32_AUTO): fp32-accurate results with the arithmetic chosen per launch
    by the expected row counts.  A small batch runs entirely on the exact-fp32 matrix instruction (bit-identical to "f32");
    at the benchmark's size conv2 and the heads' first layer take the split-bf16 kernels (L-DGN / DGN-R; HL-DGN has no conv2 and
    one head row per env) - different summation, same bar: within 1e-5 of the exact-fp32 path, which the full-size tests of
    test_gpu_forward.py hold within 1e-4 of the oracle."""
    n = 50
    net, _ = make_net(model, n, seed=4)
    assert net.feature_dtype == "f32a"
    for bs, switched in ((16, False), (1024, model != "hl_dgn")):
        obs = torch.from_numpy(random_obs(n, bs, 5)).cuda()
        with torch.no_grad():
            net.set_feature_dtype("f32a")
            auto = net(obs)[0].clone()
            net.set_feature_dtype("f32")
            exact = net(obs)[0].clone()
        err = float((auto - exact).abs().max())
        print(f"{model} bs={bs}: auto vs exact fp32 {err:.2e}")
        assert err <= 1e-5
        assert torch.equal(auto, exact) != switched
