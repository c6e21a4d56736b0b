"""``HLDGNNetwork`` drop-in (reference: graph_env/env/utils/networks/hl_dgn.py:14-119): encoder ->
one GATv2 layer -> decision-maker mask -> global max/mean/add pool -> dueling head.  Same constructor
(``aggregator`` before ``dueling_param``), same state_dict keys, ``forward -> (logits, state)``."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .. import _lib
from .common import GATv2Conv, GraphQNetwork, conv_relu, learn_adjacency, mlp, unpack, use_hip_autograd


class HLDGNNetwork(GraphQNetwork):
    _MODEL = _lib.MODEL_HLDGN

    def __init__(self, input_dim, hidden_dim, output_dim, num_heads, agents_num, aggregator="mean", dueling_param=None,
                 device="cpu", edge_attributes=False, backend="auto"):
        super().__init__()
        self._setup(input_dim, hidden_dim, output_dim, num_heads, agents_num, device, edge_attributes, backend)
        if aggregator not in _lib.AGG:                # the reference's pooling table raises KeyError too (hl_dgn.py:56-60)
            raise KeyError(aggregator)
        self.aggregator_name = aggregator
        self.conv1 = GATv2Conv(hidden_dim, hidden_dim, heads=num_heads)
        self._build_heads(hidden_dim * num_heads, dueling_param)
        self.to(device)

    def torch_forward(self, obs: torch.Tensor) -> torch.Tensor:
        """hl_dgn.py:97-115 in differentiable ops: one conv, dm mask, pool over the graph, head."""
        obs = obs.to(self.device)
        pos, feats, dm, _g = unpack(obs, self.input_dim, self.agents_num)
        bs, n = pos.shape[:2]
        hip = use_hip_autograd(self, obs)
        x = mlp(self.encoder, feats.reshape(bs * n, -1), hip, final_relu=True)
        x = conv_relu(self.conv1, x, learn_adjacency(obs, pos, n, self.input_dim, hip), n, hip)
        if hip:
            from .autograd_ops import graph_pool
            return self._head(graph_pool(x, dm, n, self.aggregator_name), hip)
        x = (x * dm.reshape(bs * n, 1)).view(bs, n, -1)
        pooled = {"max": lambda t: t.max(dim=1).values, "mean": lambda t: t.mean(dim=1), "add": lambda t: t.sum(dim=1)}
        return self._head(pooled[self.aggregator_name](x))
