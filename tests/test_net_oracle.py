"""Network oracle self-consistency: two independent formulations agree, goldens reproduce, shape
errors match the reference's (common.py:20-29).  Parity with the real PyG wheels is UNPINNED (absent)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import net_oracle as no

GOLDENS = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "net_golden_*.npz")))
TOL = 1e-4          # north_star: action logits within 1e-4 fp32


@pytest.mark.parametrize("path", GOLDENS, ids=[os.path.basename(p)[11:-4] for p in GOLDENS])
def test_formulations_agree_and_match_golden(path):
    torch.set_num_threads(4)
    g = np.load(path)
    n, obs, ws = int(g["n"]), g["obs"], int(g["weight_seed"])
    with torch.no_grad():
        sd = no.init_weights("l_dgn", seed=ws, random_conv_bias=True)
        e, ie = no.ldgn_forward(sd, obs, n, formulation="edges", return_intermediates=True)
        d, idn = no.ldgn_forward(sd, obs, n, formulation="dense", return_intermediates=True)
        assert torch.equal(ie["adj"], idn["adj"])
        np.testing.assert_allclose(e.numpy(), d.numpy(), atol=2e-5, rtol=0)
        np.testing.assert_allclose(e.numpy(), g["ldgn_logits"], atol=1e-6, rtol=0)
        for k in ("x_1", "x_2", "x_3"):
            np.testing.assert_allclose(idn[k].numpy(), g[f"ldgn_{k}"], atol=2e-5, rtol=0)
        adj_bits = np.unpackbits(g["adj"], axis=-1, bitorder="little")[..., :n].astype(bool)
        assert np.array_equal(adj_bits, ie["adj"].numpy())
        sd = no.init_weights("dgn_r", seed=ws + 2)
        e = no.dgnr_forward(sd, obs, n, formulation="edges")
        d = no.dgnr_forward(sd, obs, n, formulation="dense")
        np.testing.assert_allclose(e.numpy(), d.numpy(), atol=2e-5, rtol=0)
        np.testing.assert_allclose(e.numpy(), g["dgnr_logits"], atol=1e-6, rtol=0)
        sd = no.init_weights("hl_dgn", seed=ws + 1, random_conv_bias=True)
        for agg in ("max", "mean", "add"):
            e = no.hldgn_forward(sd, obs, n, aggregator=agg, formulation="edges")
            d = no.hldgn_forward(sd, obs, n, aggregator=agg, formulation="dense")
            np.testing.assert_allclose(e.numpy(), d.numpy(), atol=2e-5, rtol=0)
            np.testing.assert_allclose(e.numpy(), g[f"hldgn_{agg}_logits"], atol=1e-6, rtol=0)


def test_shape_errors_match_reference():
    sd = no.init_weights("hl_dgn")
    with pytest.raises(ValueError, match="Expected obs to be 2D"):
        no.hldgn_forward(sd, np.zeros((161,), np.float32), 20)
    with pytest.raises(ValueError, match="feature cols for nodes"):
        no.hldgn_forward(sd, np.zeros((2, 160), np.float32), 20)


def test_radius_rule_strict_fp32_and_cap():
    # two points exactly r apart in fp32 arithmetic are NOT neighbours (strict <), just inside are
    pos = torch.zeros(1, 3, 2)
    pos[0, 1, 0] = 0.2
    pos[0, 2, 0] = 0.19999
    adj = no.radius_adjacency(pos)
    d2 = (pos[0, 1, 0] - pos[0, 0, 0]) ** 2
    assert bool(adj[0, 0, 1]) == bool(d2 < torch.tensor(0.2 * 0.2, dtype=torch.float64).float())
    assert adj[0, 0, 2] and adj[0, 2, 0] and not adj[0, 0, 0]
    # neighbour cap: 50 coincident points -> every target keeps sources from the first 33 indices
    adj = no.radius_adjacency(torch.full((1, 50, 2), 0.5))
    assert int(adj[0, 0].sum()) == 32 and int(adj[0, 40].sum()) == 33
    assert not adj[0, 40, 33:].any()


def test_state_dict_names_and_counts():
    sd = no.init_weights("l_dgn")
    assert sum(v.numel() for v in sd.values()) == 1005315
    assert sd["conv2.lin_l.weight"].shape == (512, 512) and sd["conv1.att"].shape == (1, 4, 128)
    sd = no.init_weights("dgn_r")
    assert sum(v.numel() for v in sd.values()) == 1660675          # incl. the unused lin_skip
    assert sd["conv2.lin_key.weight"].shape == (512, 512) and "conv1.att" not in sd
    sd = no.init_weights("hl_dgn")
    assert sum(v.numel() for v in sd.values()) == 315139
    assert "conv2.att" not in sd and sd["Q.model.0.weight"].shape == (128, 512)


def test_dqn_act_masking():
    logits = torch.tensor([[0.3, 0.1], [0.0, 0.2], [0.5, 0.4]])
    assert no.dqn_act(logits).tolist() == [0, 1, 0]
    assert no.dqn_act(logits, [[1, 1], [1, 0], [0, 1]]).tolist() == [0, 0, 1]


def test_dqn_act_and_exploration_noise_known_answers():
    """[3P] DQNPolicy.forward masking and exploration_noise on hand-computed cases."""
    logits = torch.tensor([[1.0, 2.0], [3.0, -1.0], [0.5, 0.5], [-4.0, 7.0]])
    mask = np.array([[1, 0], [1, 1], [0, 1], [0, 0]])
    # row 0: the better action is illegal; row 2: tie broken by the mask; row 3: nothing legal -> both shifted alike
    assert no.dqn_act(logits, mask).tolist() == [0, 0, 1, 1]
    assert no.dqn_act(logits).tolist() == [1, 0, 0, 1]
    greedy = np.array([0, 0, 1, 1])
    ru = np.array([0.1, 0.9, 0.29, 0.31])
    rq = np.array([[0.2, 0.9], [0.9, 0.1], [0.8, 0.3], [0.6, 0.7]])
    # eps 0.3: rows 0 and 2 explore; row 0's random favourite (1) is illegal -> 0.2 + 1 beats 0.9 + 0; row 2 -> 0.3 + 1
    assert no.dqn_exploration_noise(greedy, 0.3, ru, rq, mask).tolist() == [0, 0, 1, 1]
    assert no.dqn_exploration_noise(greedy, 0.3, ru, rq).tolist() == [1, 0, 0, 1]
    assert no.dqn_exploration_noise(greedy, 1.0, ru, rq).tolist() == [1, 0, 0, 1]
    assert no.dqn_exploration_noise(greedy, 0.0, ru, rq).tolist() == greedy.tolist()
