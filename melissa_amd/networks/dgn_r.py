"""``DGNRNetwork`` drop-in (reference: graph_env/env/utils/networks/dgn_r.py:13-129): encoder -> two
``TransformerConv(root_weight=False)`` layers with controlling-agent snapshots -> dueling head.  Same
constructor, same state_dict keys (``conv{1,2}.lin_{key,query,value,skip}.*``), ``forward -> (logits, state)``;
inference through mel_dgnr_forward."""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from .common import (MLP, HipForwardMixin, TransformerConv, conv_relu, learn_adjacency, radius_adjacency, transformer_dense,
                     unpack, use_hip_autograd)


class DGNRNetwork(HipForwardMixin, nn.Module):
    _MODEL = _lib.MODEL_DGNR

    def __init__(self, input_dim: int, hidden_dim: int, output_dim: int, num_heads: int, agents_num: int,
                 dueling_param: Optional[Tuple[Dict[str, Any], Dict[str, Any]]] = None, device: str = "cpu",
                 edge_attributes: bool = False, backend: str = "auto"):
        super().__init__()
        self.device = device
        self.input_dim, self.hidden_dim, self.output_dim = input_dim, hidden_dim, output_dim
        self.num_heads, self.agents_num = num_heads, agents_num
        self.edge_attributes = edge_attributes
        self.backend = backend
        self.encoder = MLP(input_dim=input_dim, hidden_sizes=[hidden_dim], output_dim=hidden_dim, device=device)
        self.conv1 = TransformerConv(hidden_dim, hidden_dim, heads=num_heads, root_weight=False)
        self.conv2 = TransformerConv(hidden_dim * num_heads, hidden_dim, heads=num_heads, root_weight=False)
        self.use_dueling = dueling_param is not None
        self.final_latent_dim = hidden_dim + hidden_dim * num_heads * 2               # dgn_r.py:63
        if self.use_dueling:
            q_kwargs, v_kwargs = dueling_param                                          # dgn_r.py:66-78
            q_kwargs.update({"input_dim": self.final_latent_dim, "output_dim": output_dim, "device": device})
            v_kwargs.update({"input_dim": self.final_latent_dim, "output_dim": 1, "device": device})
            self.Q = MLP(**q_kwargs)
            self.V = MLP(**v_kwargs)
        else:
            self.out_linear = nn.Linear(self.final_latent_dim, output_dim)
        self.to(device)

    def forward(self, obs, state=None, info={}):
        obs = self._prepare_obs(obs)
        return self._dispatch(obs), state

    def torch_forward(self, obs: torch.Tensor) -> torch.Tensor:
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
