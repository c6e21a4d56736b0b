"""``LDGNNetwork`` drop-in (reference: graph_env/env/utils/networks/l_dgn.py:12-151).

Same constructor signature, same parameter tree / state_dict keys (``encoder.model.{0,2}``, ``conv1.*``, ``conv2.*``,
``Q.model.{0,2,4}``, ``V.model.{0,2,4}`` or ``out_linear``), same ``forward(obs, state=None, info={}) -> (logits, None)``
contract; the inference arithmetic runs in hand-written HIP through the C ABI (mel_ldgn_forward).
"""
from __future__ import annotations

from .. import _lib
from .common import GATv2Conv, GraphQNetwork


class LDGNNetwork(GraphQNetwork):
    _MODEL = _lib.MODEL_LDGN
    _RETURNS_STATE = False

    def __init__(self, input_dim, hidden_dim, output_dim, num_heads, agents_num, dueling_param=None, device="cpu",
                 edge_attributes=False, backend="auto"):
        super().__init__()
        self._setup(input_dim, hidden_dim, output_dim, num_heads, agents_num, device, edge_attributes, backend)
        self.conv1 = GATv2Conv(hidden_dim, hidden_dim, heads=num_heads)
        self.conv2 = GATv2Conv(hidden_dim * num_heads, hidden_dim, heads=num_heads)
        self.final_latent_dim = hidden_dim + 2 * hidden_dim * num_heads               # x_1 | x_2 | x_3, l_dgn.py:44
        self._build_heads(self.final_latent_dim, dueling_param, honour_output_dim_key=True)
        self.to(device)

    torch_forward = GraphQNetwork._two_conv_torch_forward
