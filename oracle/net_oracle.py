"""CPU restatement of the reference network half (TEST INFRASTRUCTURE - the oracle).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.

PARITY UNPINNED: the arithmetic of this half lives in third-party wheels that are absent from the
build container and from /root/reference (torch-geometric ~=2.2.0 ``GATv2Conv`` / ``global_*_pool`` /
``radius_graph`` -> torch_cluster, tianshou ==1.0.0 ``MLP``; requirements.txt:5,10,20-24) and the
reference has no test or golden tensor for it.  This file restates their published
algorithms at the reference's call sites:

* ``build_pyg_batch_time``   graph_env/env/utils/networks/common.py:6-64
* ``LDGNNetwork.forward``    graph_env/env/utils/networks/l_dgn.py:92-151
* ``HLDGNNetwork.forward``   graph_env/env/utils/networks/hl_dgn.py:82-119
* ``DGNRNetwork.forward``    graph_env/env/utils/networks/dgn_r.py:82-129 ([3P] ``TransformerConv``, A.2)
* [3P] ``radius_graph(pos, r=0.2, loop=False)``: fp32, dist^2 < r^2 strict, first 33 hits in index
  order per target then self dropped (torch_cluster CUDA kernel semantics for max_num_neighbors=32)
* [3P] ``GATv2Conv`` (A.1), ``global_{max,mean,add}_pool`` (A.4), tianshou ``MLP`` (A.0),
  ``DQNPolicy.forward`` mask+argmax (A.5)

Two independent formulations are kept and cross-checked by tests/test_net_oracle.py:
``formulation="edges"`` (edge list + scatter, the way PyG computes it) and ``formulation="dense"``
(per-graph N x N masked attention).  Goldens in tests/golden/net_golden_*.npz come from the "edges"
formulation (generator: tests/golden/make_net_golden.py).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

RADIUS_OF_INFLUENCE = 0.20
NODE_COLS = 8                    # x, y, 5 features, dm flag (graph.py:80)
MAX_NUM_NEIGHBORS = 32           # torch_cluster default


# ----------------------------------------------------------------------------------------------
# weights (reference state_dict key names)
# ----------------------------------------------------------------------------------------------
def _uniform(gen, bound, *shape):
    """U(-bound, bound) from a legacy numpy RandomState (its stream is frozen across numpy versions,
    so a seed alone reproduces the golden weights; fixtures then only store obs + logits)."""
    return torch.from_numpy(gen.uniform(-bound, bound, size=shape).astype(np.float32))


def _linear_init(gen, out_f, in_f):
    """torch.nn.Linear default init (kaiming_uniform a=sqrt(5) == U(+-1/sqrt(in)) for W and b)."""
    bound = 1.0 / math.sqrt(in_f)
    return _uniform(gen, bound, out_f, in_f), _uniform(gen, bound, out_f)


def _glorot(gen, *shape):
    return _uniform(gen, math.sqrt(6.0 / (shape[-2] + shape[-1])), *shape)


def init_weights(model: str, input_dim=5, hidden=128, heads=4, n_actions=2, seed=9,
                 dueling_hidden=(128, 128), random_conv_bias=False, dueling=True):
    """Random-init weights with the reference's parameter names and shapes.  ``model`` in
    {"l_dgn", "hl_dgn"}.  Distribution follows the [3P] defaults (nn.Linear default for MLPs, glorot
    for lin_l/lin_r/att, zeros for the conv bias unless ``random_conv_bias`` - tests set it so that a
    missing bias add is caught).  ``dueling=False``: the single ``out_linear`` head the reference builds when
    ``dueling_param`` is None (l_dgn.py:90, hl_dgn.py:80, dgn_r.py:80)."""
    gen = np.random.RandomState(seed)
    sd = {}

    def mlp(prefix, sizes):
        for k, (i, o) in enumerate(zip(sizes[:-1], sizes[1:])):
            w, b = _linear_init(gen, o, i)
            sd[f"{prefix}.model.{2 * k}.weight"] = w
            sd[f"{prefix}.model.{2 * k}.bias"] = b

    def conv(prefix, in_f):
        hc = heads * hidden
        for lin in ("lin_l", "lin_r"):
            sd[f"{prefix}.{lin}.weight"] = _glorot(gen, hc, in_f)
            sd[f"{prefix}.{lin}.bias"] = _uniform(gen, 1.0 / math.sqrt(in_f), hc)
        sd[f"{prefix}.att"] = _glorot(gen, 1, heads, hidden)
        sd[f"{prefix}.bias"] = _uniform(gen, 0.1, hc) if random_conv_bias else torch.zeros(hc)

    def tconv(prefix, in_f):
        # PyG 2.2 TransformerConv(in, C, heads, root_weight=False): lin_key / lin_query / lin_value and a
        # lin_skip that exists as a parameter but is never used (dgn_r.py:47-58)
        hc = heads * hidden
        for lin in ("lin_key", "lin_query", "lin_value", "lin_skip"):
            sd[f"{prefix}.{lin}.weight"] = _glorot(gen, hc, in_f)
            sd[f"{prefix}.{lin}.bias"] = _uniform(gen, 1.0 / math.sqrt(in_f), hc)

    mlp("encoder", [input_dim, hidden, hidden])
    if model == "dgn_r":
        tconv("conv1", hidden)
        tconv("conv2", hidden * heads)
        latent = hidden + 2 * hidden * heads                     # dgn_r.py:63
    elif model == "l_dgn":
        conv("conv1", hidden)
        conv("conv2", hidden * heads)
        latent = hidden + 2 * hidden * heads                     # l_dgn.py:44
    elif model == "hl_dgn":
        conv("conv1", hidden)
        latent = hidden * heads                                  # hl_dgn.py:64
    else:
        raise ValueError(model)
    if dueling:
        mlp("Q", [latent, *dueling_hidden, n_actions])
        mlp("V", [latent, *dueling_hidden, 1])
    else:
        sd["out_linear.weight"], sd["out_linear.bias"] = _linear_init(gen, n_actions, latent)
    return sd


# ----------------------------------------------------------------------------------------------
# N1: unpack (common.py:6-64) and N2: radius adjacency
# ----------------------------------------------------------------------------------------------
def unpack_obs(obs, agents_num, input_dim=5):
    if obs.ndim != 2:
        raise ValueError(f"Expected obs to be 2D, but got shape {obs.shape}")
    bs, dim = obs.shape
    expected = agents_num * (input_dim + 3)
    if dim - 1 != expected:
        raise ValueError(f"Expected {expected} feature cols for nodes, got {dim - 1}")
    node = obs[:, :dim - 1].reshape(bs, agents_num, input_dim + 3).float()
    pos = node[:, :, :2]
    feats = node[:, :, 2:2 + input_dim]
    dm_mask = node[:, :, -1:]
    agent_idx = obs[:, -1].clamp(0, agents_num - 1).long()
    return pos, feats, dm_mask, agent_idx


def radius_adjacency(pos: torch.Tensor, r=RADIUS_OF_INFLUENCE, cap=MAX_NUM_NEIGHBORS):
    """adj[b, i, j] = True iff there is an edge j -> i (source j, target i).  fp32, strict <,
    threshold float(r*r computed in double); per target the first cap+1 hits in index order
    (self included) survive, then self is dropped."""
    r2 = torch.tensor(float(r) * float(r), dtype=torch.float64).to(torch.float32)
    dx = pos[:, :, None, 0] - pos[:, None, :, 0]
    dy = pos[:, :, None, 1] - pos[:, None, :, 1]
    d2 = dx * dx + dy * dy                       # two roundings + one add, no fma
    within = d2 < r2                             # includes the diagonal (0 < r2)
    rank = torch.cumsum(within.to(torch.int32), dim=2)
    within = within & (rank <= cap + 1)
    n = pos.shape[1]
    eye = torch.eye(n, dtype=torch.bool)
    return within & ~eye


# ----------------------------------------------------------------------------------------------
# small pieces
# ----------------------------------------------------------------------------------------------
def _mlp(sd, prefix, x, n_layers):
    for k in range(n_layers):
        x = F.linear(x, sd[f"{prefix}.model.{2 * k}.weight"], sd[f"{prefix}.model.{2 * k}.bias"])
        if k < n_layers - 1:
            x = F.relu(x)
    return x


def _n_layers(sd, prefix):
    k = 0
    while f"{prefix}.model.{2 * k}.weight" in sd:
        k += 1
    return k


def gatv2_edges(sd, prefix, x, adj, heads):
    """GATv2Conv as PyG computes it (A.1): edge list (+ self loops) and scatter ops.
    x [bs*N, in]; adj [bs, N, N] bool (target i, source j)."""
    bs, n, _ = adj.shape
    hc = sd[f"{prefix}.lin_l.weight"].shape[0]
    c = hc // heads
    x_l = F.linear(x, sd[f"{prefix}.lin_l.weight"], sd[f"{prefix}.lin_l.bias"]).view(-1, heads, c)
    x_r = F.linear(x, sd[f"{prefix}.lin_r.weight"], sd[f"{prefix}.lin_r.bias"]).view(-1, heads, c)
    b_idx, i_idx, j_idx = torch.nonzero(adj, as_tuple=True)
    src = b_idx * n + j_idx
    dst = b_idx * n + i_idx
    loops = torch.arange(bs * n)
    src = torch.cat([src, loops])
    dst = torch.cat([dst, loops])
    m = F.leaky_relu(x_r[dst] + x_l[src], 0.2)                          # [E, H, C]
    e = (m * sd[f"{prefix}.att"]).sum(dim=-1)                          # [E, H]
    e_max = torch.full((bs * n, heads), -float("inf")).scatter_reduce(
        0, dst[:, None].expand(-1, heads), e, reduce="amax", include_self=True)
    p = (e - e_max[dst]).exp()
    denom = torch.zeros(bs * n, heads).index_add_(0, dst, p) + 1e-16
    alpha = p / denom[dst]
    out = torch.zeros(bs * n, heads, c).index_add_(0, dst, x_l[src] * alpha[:, :, None])
    return out.reshape(bs * n, hc) + sd[f"{prefix}.bias"]


def gatv2_dense(sd, prefix, x, adj, heads):
    """Same operator as a per-graph dense masked attention (independent formulation)."""
    bs, n, _ = adj.shape
    hc = sd[f"{prefix}.lin_l.weight"].shape[0]
    c = hc // heads
    x_l = (x @ sd[f"{prefix}.lin_l.weight"].t() + sd[f"{prefix}.lin_l.bias"]).view(bs, n, heads, c)
    x_r = (x @ sd[f"{prefix}.lin_r.weight"].t() + sd[f"{prefix}.lin_r.bias"]).view(bs, n, heads, c)
    mask = adj | torch.eye(n, dtype=torch.bool)[None]
    s = x_r[:, :, None] + x_l[:, None, :]                              # [bs, i, j, H, C]
    s = torch.where(s > 0, s, 0.2 * s)
    e = torch.einsum("bijhc,hc->bijh", s, sd[f"{prefix}.att"][0])
    e = e.masked_fill(~mask[..., None], -float("inf"))
    e_max = e.max(dim=2, keepdim=True).values
    p = torch.exp(e - e_max)
    p = torch.where(mask[..., None], p, torch.zeros(()))
    alpha = p / (p.sum(dim=2, keepdim=True) + 1e-16)
    out = torch.einsum("bijh,bjhc->bihc", alpha, x_l)
    return out.reshape(bs * n, hc) + sd[f"{prefix}.bias"]


_GAT = {"edges": gatv2_edges, "dense": gatv2_dense}


def transformer_edges(sd, prefix, x, adj, heads):
    """[3P] PyG TransformerConv(root_weight=False) (A.2): q = lin_query(x_i), k = lin_key(x_j),
    v = lin_value(x_j); e = q.k / sqrt(C); softmax over incoming edges (eps 1e-16); NO self-loops (an
    isolated target gets zeros); heads concatenated; no output bias, lin_skip unused."""
    bs, n, _ = adj.shape
    hc = sd[f"{prefix}.lin_key.weight"].shape[0]
    c = hc // heads
    q = F.linear(x, sd[f"{prefix}.lin_query.weight"], sd[f"{prefix}.lin_query.bias"]).view(-1, heads, c)
    k = F.linear(x, sd[f"{prefix}.lin_key.weight"], sd[f"{prefix}.lin_key.bias"]).view(-1, heads, c)
    v = F.linear(x, sd[f"{prefix}.lin_value.weight"], sd[f"{prefix}.lin_value.bias"]).view(-1, heads, c)
    b_idx, i_idx, j_idx = torch.nonzero(adj, as_tuple=True)
    src = b_idx * n + j_idx
    dst = b_idx * n + i_idx
    e = (q[dst] * k[src]).sum(dim=-1) / math.sqrt(c)
    e_max = torch.full((bs * n, heads), -float("inf")).scatter_reduce(
        0, dst[:, None].expand(-1, heads), e, reduce="amax", include_self=True)
    p = (e - e_max[dst]).exp()
    denom = torch.zeros(bs * n, heads).index_add_(0, dst, p) + 1e-16
    alpha = p / denom[dst]
    out = torch.zeros(bs * n, heads, c).index_add_(0, dst, v[src] * alpha[:, :, None])
    return out.reshape(bs * n, hc)


def transformer_dense(sd, prefix, x, adj, heads):
    bs, n, _ = adj.shape
    hc = sd[f"{prefix}.lin_key.weight"].shape[0]
    c = hc // heads
    q = (x @ sd[f"{prefix}.lin_query.weight"].t() + sd[f"{prefix}.lin_query.bias"]).view(bs, n, heads, c)
    k = (x @ sd[f"{prefix}.lin_key.weight"].t() + sd[f"{prefix}.lin_key.bias"]).view(bs, n, heads, c)
    v = (x @ sd[f"{prefix}.lin_value.weight"].t() + sd[f"{prefix}.lin_value.bias"]).view(bs, n, heads, c)
    e = torch.einsum("bihc,bjhc->bijh", q, k) / math.sqrt(c)
    e = e.masked_fill(~adj[..., None], -float("inf"))
    e_max = e.max(dim=2, keepdim=True).values
    e_max = torch.where(torch.isfinite(e_max), e_max, torch.zeros(()))     # isolated targets
    p = torch.where(adj[..., None], torch.exp(e - e_max), torch.zeros(()))
    alpha = p / (p.sum(dim=2, keepdim=True) + 1e-16)
    return torch.einsum("bijh,bjhc->bihc", alpha, v).reshape(bs * n, hc)


_TCONV = {"edges": transformer_edges, "dense": transformer_dense}


def _dueling(sd, x):
    if "out_linear.weight" in sd:                                         # l_dgn.py:149, hl_dgn.py:117, dgn_r.py:127
        return F.linear(x, sd["out_linear.weight"], sd["out_linear.bias"])
    q = _mlp(sd, "Q", x, _n_layers(sd, "Q"))
    v = _mlp(sd, "V", x, _n_layers(sd, "V"))
    return q - q.mean(dim=1, keepdim=True) + v                            # l_dgn.py:142-147


# ----------------------------------------------------------------------------------------------
# the two forwards
# ----------------------------------------------------------------------------------------------
def ldgn_forward(sd, obs, agents_num, heads=4, formulation="edges", return_intermediates=False):
    """l_dgn.py:92-151.  obs: float tensor / ndarray [bs, 8N+1] -> logits [bs, n_actions] fp32."""
    obs = torch.as_tensor(np.asarray(obs) if not torch.is_tensor(obs) else obs)
    pos, feats, dm, g = unpack_obs(obs, agents_num, sd["encoder.model.0.weight"].shape[1])
    bs, n = pos.shape[:2]
    adj = radius_adjacency(pos)
    gat = _GAT[formulation]
    x = F.relu(_mlp(sd, "encoder", feats.reshape(bs * n, -1), 2))         # :117-118
    gi = torch.arange(bs) * n + g                                        # :121
    x_1 = x[gi]
    x = F.relu(gat(sd, "conv1", x, adj, heads))                          # :125-126
    x_2 = x[gi]                                                          # :127 (before the mask)
    x = x * dm.reshape(bs * n, 1)                                        # :128
    h1 = x
    x = F.relu(gat(sd, "conv2", x, adj, heads))                          # :133-134
    x_3 = x[gi]
    x_cat = torch.cat([x_1, x_2, x_3], dim=1)                            # :139
    out = _dueling(sd, x_cat)
    if return_intermediates:
        return out, dict(adj=adj, x_1=x_1, x_2=x_2, x_3=x_3, h1=h1)
    return out


def dgnr_forward(sd, obs, agents_num, heads=4, formulation="edges", return_intermediates=False):
    """dgn_r.py:82-129: same skeleton as L-DGN with TransformerConv layers."""
    obs = torch.as_tensor(np.asarray(obs) if not torch.is_tensor(obs) else obs)
    pos, feats, dm, g = unpack_obs(obs, agents_num, sd["encoder.model.0.weight"].shape[1])
    bs, n = pos.shape[:2]
    adj = radius_adjacency(pos)
    conv = _TCONV[formulation]
    x = F.relu(_mlp(sd, "encoder", feats.reshape(bs * n, -1), 2))         # :97-98
    gi = torch.arange(bs) * n + g
    x_1 = x[gi]
    x = F.relu(conv(sd, "conv1", x, adj, heads))                         # :104-105
    x_2 = x[gi]
    x = x * dm.reshape(bs * n, 1)                                        # :109
    x = F.relu(conv(sd, "conv2", x, adj, heads))                         # :112-113
    x_3 = x[gi]
    out = _dueling(sd, torch.cat([x_1, x_2, x_3], dim=1))
    if return_intermediates:
        return out, dict(adj=adj, x_1=x_1, x_2=x_2, x_3=x_3)
    return out


def hldgn_forward(sd, obs, agents_num, heads=4, aggregator="max", formulation="edges",
                  return_intermediates=False):
    """hl_dgn.py:82-119."""
    obs = torch.as_tensor(np.asarray(obs) if not torch.is_tensor(obs) else obs)
    pos, feats, dm, _g = unpack_obs(obs, agents_num, sd["encoder.model.0.weight"].shape[1])
    bs, n = pos.shape[:2]
    adj = radius_adjacency(pos)
    x = F.relu(_mlp(sd, "encoder", feats.reshape(bs * n, -1), 2))
    x = F.relu(_GAT[formulation](sd, "conv1", x, adj, heads))
    x = (x * dm.reshape(bs * n, 1)).view(bs, n, -1)                      # :105
    if aggregator == "max":
        pooled = x.max(dim=1).values
    elif aggregator == "mean":
        pooled = x.mean(dim=1)
    elif aggregator == "add":
        pooled = x.sum(dim=1)
    else:
        raise KeyError(aggregator)
    out = _dueling(sd, pooled)
    if return_intermediates:
        return out, dict(adj=adj, pooled=pooled)
    return out


def dqn_act(logits, mask=None):
    """[3P] tianshou DQNPolicy.forward / compute_q_value (A.5): illegal actions pushed below the
    batch minimum, then argmax."""
    q = logits
    if mask is not None:
        m = torch.as_tensor(mask, dtype=q.dtype)
        q = q + (1 - m) * (q.min() - q.max() - 1.0)
    return q.argmax(dim=1)


def dqn_exploration_noise(act, eps, rand_u, rand_q, mask=None):
    """[3P] tianshou 1.0.0 ``DQNPolicy.exploration_noise``, reached through
    ``MultiAgentSharedPolicy.exploration_noise`` (policies/multi_agent_managers/shared_policy.py:81-91):

        rand_mask = np.random.rand(bs) < eps;  q = np.random.rand(bs, n_act) (+ mask)
        act[rand_mask] = q.argmax(1)[rand_mask]

    The two uniform draws are arguments (``rand_u`` [bs], ``rand_q`` [bs, n_act]) so that a caller can feed the
    stream the product used; eps == 0 leaves ``act`` alone, as the reference's ``np.isclose(eps, 0)`` guard does."""
    act = np.array(act, dtype=np.int64, copy=True)
    if np.isclose(eps, 0.0):
        return act
    rand_mask = np.asarray(rand_u) < eps
    q = np.array(rand_q, dtype=np.float64, copy=True)
    if mask is not None:
        q += np.asarray(mask, dtype=np.float64)
    rand_act = q.argmax(axis=1)
    act[rand_mask] = rand_act[rand_mask]
    return act
