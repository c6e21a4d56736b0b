"""``DGNRNetwork`` drop-in (reference: graph_env/env/utils/networks/dgn_r.py:13-129): encoder -> two
``TransformerConv(root_weight=False)`` layers with controlling-agent snapshots -> dueling head.  Same constructor, same
state_dict keys (``conv{1,2}.lin_{key,query,value,skip}.*``), ``forward -> (logits, state)``; inference through
mel_dgnr_forward."""
from __future__ import annotations

from .. import _lib
from .common import GraphQNetwork, TransformerConv


class DGNRNetwork(GraphQNetwork):
    _MODEL = _lib.MODEL_DGNR

    def __init__(self, input_dim, hidden_dim, output_dim, num_heads, agents_num, dueling_param=None, device="cpu",
                 edge_attributes=False, backend="auto"):
        super().__init__()
        self._setup(input_dim, hidden_dim, output_dim, num_heads, agents_num, device, edge_attributes, backend)
        self.conv1 = TransformerConv(hidden_dim, hidden_dim, heads=num_heads, root_weight=False)
        self.conv2 = TransformerConv(hidden_dim * num_heads, hidden_dim, heads=num_heads, root_weight=False)
        self.final_latent_dim = hidden_dim + 2 * hidden_dim * num_heads               # dgn_r.py:63
        self._build_heads(self.final_latent_dim, dueling_param)
        self.to(device)

    torch_forward = GraphQNetwork._two_conv_torch_forward
