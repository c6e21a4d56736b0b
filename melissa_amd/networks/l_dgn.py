"""``LDGNNetwork`` drop-in (reference: graph_env/env/utils/networks/l_dgn.py:12-151).

Same constructor signature, same parameter tree / state_dict keys
(``encoder.model.{0,2}``, ``conv1.*``, ``conv2.*``, ``Q.model.{0,2,4}``, ``V.model.{0,2,4}`` or
``out_linear``), same ``forward(obs, state=None, info={}) -> (logits, None)`` contract; the inference
arithmetic runs in hand-written HIP through the C ABI (mel_ldgn_forward).
"""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from .common import (MLP, GATv2Conv, HipForwardMixin, conv_relu, gatv2_dense, learn_adjacency, radius_adjacency, unpack,
                     use_hip_autograd)


class LDGNNetwork(HipForwardMixin, nn.Module):
    _MODEL = _lib.MODEL_LDGN

    def __init__(self, input_dim: int, hidden_dim: int, output_dim: int, num_heads: int, agents_num: int,
                 dueling_param: Optional[Tuple[Dict[str, Any], Dict[str, Any]]] = None, device: str = "cpu",
                 edge_attributes=False, backend: str = "auto"):
        super().__init__()
        self.device = device
        self.input_dim, self.hidden_dim, self.output_dim = input_dim, hidden_dim, output_dim
        self.num_heads, self.agents_num = num_heads, agents_num
        self.edge_attributes = edge_attributes      # computed-but-unused in the reference (l_dgn.py:125,133)
        self.backend = backend                      # "auto" | "hip" | "torch"
        self.final_latent_dim = hidden_dim + hidden_dim * num_heads * 2            # l_dgn.py:44
        self.use_dueling = dueling_param is not None
        self.encoder = MLP(input_dim=input_dim, hidden_sizes=[hidden_dim], output_dim=hidden_dim, device=device)
        self.conv1 = GATv2Conv(hidden_dim, hidden_dim, heads=num_heads)
        self.conv2 = GATv2Conv(hidden_dim * num_heads, hidden_dim, heads=num_heads)
        if self.use_dueling:
            q_kwargs, v_kwargs = dueling_param                                       # mutated, as l_dgn.py:71-84
            q_output_dim = q_kwargs.pop("output_dim", output_dim)
            v_output_dim = v_kwargs.pop("output_dim", 1)
            q_kwargs.update({"input_dim": self.final_latent_dim, "output_dim": q_output_dim, "device": device})
            v_kwargs.update({"input_dim": self.final_latent_dim, "output_dim": v_output_dim, "device": device})
            self.Q = MLP(**q_kwargs)
            self.V = MLP(**v_kwargs)
            self.output_dim = q_output_dim
        else:
            self.out_linear = nn.Linear(self.final_latent_dim, output_dim)
        self.to(device)

    def forward(self, obs, state=None, info={}):
        obs = self._prepare_obs(obs)
        return self._dispatch(obs), None

    def torch_forward(self, obs: torch.Tensor) -> torch.Tensor:
        """l_dgn.py:117-149 in differentiable torch ops (learn path)."""
        obs = obs.to(self.device)
        pos, feats, dm, g = unpack(obs, self.input_dim, self.agents_num)
        bs, n = pos.shape[:2]
        hip = use_hip_autograd(self, obs)
        adj = learn_adjacency(obs, pos, n, self.input_dim, hip)
        x = F.relu(self.encoder.model(feats.reshape(bs * n, -1)))
        gi = torch.arange(bs, device=x.device) * n + g
        x_1 = x[gi]
        x = conv_relu(self.conv1, x, adj, n, hip)
        x_2 = x[gi]
        x = x * dm.reshape(bs * n, 1)
        x = conv_relu(self.conv2, x, adj, n, hip)
        x_cat = torch.cat([x_1, x_2, x[gi]], dim=1)
        if self.use_dueling:
            q, v = self.Q.model(x_cat), self.V.model(x_cat)
            return q - q.mean(dim=1, keepdim=True) + v
        return self.out_linear(x_cat)
