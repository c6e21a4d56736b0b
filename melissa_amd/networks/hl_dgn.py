"""``HLDGNNetwork`` drop-in (reference: graph_env/env/utils/networks/hl_dgn.py:14-119): encoder ->
one GATv2 layer -> decision-maker mask -> global max/mean/add pool -> dueling head.  Same constructor
(``aggregator`` before ``dueling_param``), same state_dict keys, ``forward -> (logits, state)``."""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from .common import (MLP, GATv2Conv, HipForwardMixin, conv_relu, gatv2_dense, learn_adjacency, radius_adjacency, unpack,
                     use_hip_autograd)


class HLDGNNetwork(HipForwardMixin, nn.Module):
    _MODEL = _lib.MODEL_HLDGN

    def __init__(self, input_dim: int, hidden_dim: int, output_dim: int, num_heads: int, agents_num: int,
                 aggregator: str = "mean", dueling_param: Optional[Tuple[Dict[str, Any], Dict[str, Any]]] = None,
                 device: str = "cpu", edge_attributes: bool = False, backend: str = "auto"):
        super().__init__()
        self.device = device
        self.input_dim, self.hidden_dim, self.output_dim = input_dim, hidden_dim, output_dim
        self.num_heads, self.agents_num = num_heads, agents_num
        self.edge_attributes = edge_attributes
        self.backend = backend
        if aggregator not in _lib.AGG:                # hl_dgn.py:56-60 raises KeyError too
            raise KeyError(aggregator)
        self.aggregator_name = aggregator
        self.encoder = MLP(input_dim=input_dim, hidden_sizes=[hidden_dim], output_dim=hidden_dim, device=device)
        self.conv1 = GATv2Conv(hidden_dim, hidden_dim, heads=num_heads)
        self.use_dueling = dueling_param is not None
        in_head_dim = hidden_dim * num_heads
        if self.use_dueling:
            q_kwargs, v_kwargs = dueling_param                                       # hl_dgn.py:66-76
            q_kwargs.update({"input_dim": in_head_dim, "output_dim": output_dim, "device": device})
            v_kwargs.update({"input_dim": in_head_dim, "output_dim": 1, "device": device})
            self.Q = MLP(**q_kwargs)
            self.V = MLP(**v_kwargs)
        else:
            self.out_linear = nn.Linear(in_head_dim, output_dim)
        self.to(device)

    def forward(self, obs, state=None, info={}):
        obs = self._prepare_obs(obs)
        return self._dispatch(obs), state

    def torch_forward(self, obs: torch.Tensor) -> torch.Tensor:
        obs = obs.to(self.device)
        pos, feats, dm, _g = unpack(obs, self.input_dim, self.agents_num)
        bs, n = pos.shape[:2]
        x = F.relu(self.encoder.model(feats.reshape(bs * n, -1)))
        hip = use_hip_autograd(self, obs)
        x = conv_relu(self.conv1, x, learn_adjacency(obs, pos, n, self.input_dim, hip), n, hip)
        if hip:
            from .autograd_ops import graph_pool
            pooled = graph_pool(x, dm, n, self.aggregator_name)
            if self.use_dueling:
                q, v = self.Q.model(pooled), self.V.model(pooled)
                return q - q.mean(dim=1, keepdim=True) + v
            return self.out_linear(pooled)
        x = (x * dm.reshape(bs * n, 1)).view(bs, n, -1)
        if self.aggregator_name == "max":
            pooled = x.max(dim=1).values
        elif self.aggregator_name == "mean":
            pooled = x.mean(dim=1)
        else:
            pooled = x.sum(dim=1)
        if self.use_dueling:
            q, v = self.Q.model(pooled), self.V.model(pooled)
            return q - q.mean(dim=1, keepdim=True) + v
        return self.out_linear(pooled)
