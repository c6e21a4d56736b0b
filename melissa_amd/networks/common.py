"""Host-side pieces shared by the network drop-ins.

Mirrors graph_env/env/utils/networks/common.py (``build_pyg_batch_time``: shape checks, unpack) and
the [3P] modules the reference networks are assembled from (tianshou ``MLP``, PyG ``GATv2Conv``) -
only as *parameter containers with the same state_dict keys*; inference arithmetic is in the HIP
library.  ``torch_forward`` is the autograd formulation used on the learn path (policy.learn needs
gradients; backward kernels are a later round).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib

RADIUS_OF_INFLUENCE = 0.20      # graph_env/env/utils/constants.py:1
MAX_NUM_NEIGHBORS = 32


class MLP(nn.Module):
    """Parameter-compatible with tianshou 1.0.0 ``MLP`` (state_dict keys ``model.{0,2,4}.*``):
    Linear, ReLU, ..., Linear with no activation after the last layer."""

    def __init__(self, input_dim: int, output_dim: int = 0, hidden_sizes: Sequence[int] = (),
                 device="cpu", **_ignored):
        super().__init__()
        self.device = device
        sizes = [input_dim, *hidden_sizes]
        layers = []
        for i, o in zip(sizes[:-1], sizes[1:]):
            layers += [nn.Linear(i, o), nn.ReLU()]
        if output_dim > 0:
            layers += [nn.Linear(sizes[-1], output_dim)]
        self.output_dim = output_dim or sizes[-1]
        self.model = nn.Sequential(*layers)

    def linears(self):
        return [m for m in self.model if isinstance(m, nn.Linear)]

    def forward(self, obs):
        obs = torch.as_tensor(obs, device=self.device, dtype=torch.float32)
        return self.model(obs.flatten(1))


class GATv2Conv(nn.Module):
    """Parameter container with PyG 2.2 ``GATv2Conv`` names/shapes/initialisation:
    ``att [1,H,C]``, ``bias [H*C]``, ``lin_l`` / ``lin_r`` Linear(in, H*C) with bias."""

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1):
        super().__init__()
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.lin_l = nn.Linear(in_channels, heads * out_channels)
        self.lin_r = nn.Linear(in_channels, heads * out_channels)
        self.att = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.empty(heads * out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        for lin in (self.lin_l, self.lin_r):      # glorot weight, PyG Linear default bias
            a = math.sqrt(6.0 / (lin.in_features + lin.out_features))
            nn.init.uniform_(lin.weight, -a, a)
            nn.init.uniform_(lin.bias, -1.0 / math.sqrt(lin.in_features), 1.0 / math.sqrt(lin.in_features))
        a = math.sqrt(6.0 / (self.heads + self.out_channels))
        nn.init.uniform_(self.att, -a, a)
        nn.init.zeros_(self.bias)


class TransformerConv(nn.Module):
    """Parameter container with PyG 2.2 ``TransformerConv(in, C, heads, root_weight=False)`` names and shapes
    (SURVEY.md A.2): ``lin_key`` / ``lin_query`` / ``lin_value`` Linear(in, H*C) with bias, plus ``lin_skip``,
    which PyG creates but never uses when ``root_weight=False`` (kept so state_dicts load)."""

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, root_weight: bool = False):
        super().__init__()
        assert not root_weight, "the reference builds TransformerConv with root_weight=False (dgn_r.py:47-58)"
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.lin_key = nn.Linear(in_channels, heads * out_channels)
        self.lin_query = nn.Linear(in_channels, heads * out_channels)
        self.lin_value = nn.Linear(in_channels, heads * out_channels)
        self.lin_skip = nn.Linear(in_channels, heads * out_channels)
        for lin in (self.lin_key, self.lin_query, self.lin_value, self.lin_skip):
            a = math.sqrt(6.0 / (lin.in_features + lin.out_features))
            nn.init.uniform_(lin.weight, -a, a)


def check_obs(obs: torch.Tensor, input_dim: int, agents_num: int):
    """The two shape errors of build_pyg_batch_time (networks/common.py:20-29)."""
    if obs.ndim != 2:
        raise ValueError(f"Expected obs to be 2D, but got shape {obs.shape}")
    expected = agents_num * (input_dim + 2 + 1)
    if obs.shape[1] - 1 != expected:
        raise ValueError(f"Expected {expected} feature cols for nodes, got {obs.shape[1] - 1}")


# ---------------------------------------------------------------------------------------------------
# ctypes views of the parameters (borrowed pointers: optimizer steps / load_state_dict are seen with
# no copy; the struct is rebuilt only if a storage moved)
# ---------------------------------------------------------------------------------------------------
def _lin(struct: _lib.MelLinear, lin: nn.Linear):
    struct.weight = lin.weight.data_ptr()
    struct.bias = lin.bias.data_ptr()
    struct.in_dim, struct.out_dim = lin.in_features, lin.out_features


def _mlp(struct: _lib.MelMlp, linears):
    if len(linears) > _lib.MAX_HEAD_LAYERS:
        raise RuntimeError(f"MLP depth {len(linears)} > {_lib.MAX_HEAD_LAYERS}")
    struct.n_layers = len(linears)
    for k, lin in enumerate(linears):
        _lin(struct.layer[k], lin)


def _gat(struct: _lib.MelGatv2, conv):
    struct.heads, struct.channels = conv.heads, conv.out_channels
    if isinstance(conv, TransformerConv):
        struct.kind = _lib.CONV_TRANSFORMER
        _lin(struct.lin_l, conv.lin_key)          # sources
        _lin(struct.lin_v, conv.lin_value)        # sources
        _lin(struct.lin_r, conv.lin_query)        # targets
        struct.att, struct.bias = None, None
        return
    struct.kind = _lib.CONV_GATV2
    _lin(struct.lin_l, conv.lin_l)
    _lin(struct.lin_r, conv.lin_r)
    struct.att, struct.bias = conv.att.data_ptr(), conv.bias.data_ptr()


class HipForwardMixin:
    """Shared machinery of the two network drop-ins: parameter views, workspace, dispatch."""

    _MODEL = None          # set by subclasses

    # "f32": the reference's arithmetic (logits within 1e-4).  "bf16": BASELINE's "bf16 feature path" - feature
    # rows and projection weights in bf16, fp32 accumulation / softmax / logits.  "f32s": fp32 features and
    # fp32-accurate projections evaluated on the bf16 matrix cores by operand splitting (MEL_PREC_F32_SPLIT).  "f32a":
    # fp32-accurate like both, the arithmetic chosen per launch by size (MEL_PREC_F32_AUTO): the split kernels for the large
    # projections of a big batch, the exact-fp32 matrix instruction for everything else.
    feature_dtype = "f32a"

    def set_feature_dtype(self, name: str):
        if name not in ("f32", "bf16", "f32s", "f32a"):
            raise ValueError(f"feature_dtype must be 'f32', 'bf16', 'f32s' or 'f32a', got {name!r}")
        self.feature_dtype = name
        self._w_cache = None
        return self

    def _param_key(self):
        return (self.feature_dtype,) + tuple((p.data_ptr(), p.dtype, p.is_contiguous()) for p in self.parameters())

    # Opt-in: the node-feature table (MEL_FWD_INTEGER_FEATURES) prepared once per weight version instead of evaluated by
    # every forward (it depends on the weights only).  Off by default: a forward then does all of its own arithmetic.
    prepared_tables = False

    def _refresh_tables(self, w, device):
        if not self.prepared_tables or self.input_dim != 5:
            w.tables, w.tables_nodes = None, 0
            return
        lib = _lib.load()
        key = (self._param_key(), tuple(p._version for p in self.parameters()), self.agents_num)
        cached = getattr(self, "_tables", None)
        if cached is None or cached[0] != key or cached[1].device != device:
            need = int(lib.mel_feature_tables_bytes(C.byref(w), self.agents_num))
            buf = cached[1] if cached is not None and cached[1].device == device and cached[1].numel() >= max(need, 16) else \
                torch.empty(max(need, 16), dtype=torch.uint8, device=device)       # same buffer: captured launches hold its address
            w.tables, w.tables_nodes = None, 0
            _lib.check(lib.mel_prepare_feature_tables(C.byref(w), self.agents_num, buf.data_ptr(), buf.numel(),
                                                      _lib.current_stream_ptr(device)), "mel_prepare_feature_tables")
            self._tables = cached = (key, buf)
        w.tables, w.tables_nodes = cached[1].data_ptr(), self.agents_num

    def _refresh_prepared(self, w, device):
        """bf16 / split precision: the converted projection weights live in a buffer this module owns and are converted
        once per weight VERSION (every in-place change of a parameter - optimizer step, load_state_dict - bumps its torch
        version counter), not once per forward.  Called outside HIP-graph capture first (the loops warm up eagerly), so a
        captured step carries no conversion launch."""
        if self.feature_dtype == "f32":
            w.prepared = None
            return
        lib = _lib.load()
        version = tuple(p._version for p in self.parameters())
        cached = getattr(self, "_prepared", None)
        if cached is not None and cached[0] == (self._param_key(), version) and cached[1].device == device:
            w.prepared = cached[1].data_ptr()
            return
        need = int(lib.mel_prepared_weights_bytes(C.byref(w)))
        buf = cached[1] if cached is not None and cached[1].numel() >= need and cached[1].device == device else \
            torch.empty(max(need, 16), dtype=torch.uint8, device=device)
        w.prepared = None
        _lib.check(lib.mel_prepare_weights(C.byref(w), buf.data_ptr(), buf.numel(), _lib.current_stream_ptr(device)),
                   "mel_prepare_weights")
        w.prepared = buf.data_ptr()
        self._prepared = ((self._param_key(), version), buf)

    def ensure_prepared(self, device=None) -> None:
        """Bring the per-weight-version caches (converted projection weights, prepared feature tables) up to date NOW, eagerly
        on the current stream and INTO THE SAME BUFFERS - for callers whose launches were captured into a HIP graph: a replay
        runs no Python, so nothing reconverts the planes the captured launches read unless the caller asks before replaying
        (``RoundLoop`` in graph mode, ``replay.CapturedUpdate`` after a target-network sync).  Cheap when nothing changed."""
        if self.feature_dtype == "f32" and not self.prepared_tables:
            return
        # fast path (a couple of microseconds: this runs before every graph replay): nothing bumped a version counter, nobody
        # marked the caches stale, same precision
        plist = getattr(self, "_plist", None)
        if plist is None:
            plist = self._plist = list(self.parameters())
        vsum = 0
        for p in plist:
            vsum += p._version
        stamp = (vsum, self.feature_dtype, self.prepared_tables, id(getattr(self, "_prepared", None)), id(getattr(self, "_tables", None)))
        if getattr(self, "_ensure_stamp", None) == stamp:
            return
        device = device if device is not None else plist[0].device
        w = self._weights()
        self._refresh_prepared(w, device)
        self._refresh_tables(w, device)
        self._ensure_stamp = (vsum, self.feature_dtype, self.prepared_tables, id(getattr(self, "_prepared", None)),
                              id(getattr(self, "_tables", None)))

    def mark_weights_changed(self) -> None:
        """The parameters were changed behind torch's version counters (an optimizer step replayed from a HIP graph): the next
        ``ensure_prepared`` / forward converts them again (same buffers: captured launches hold their addresses)."""
        if getattr(self, "_prepared", None) is not None:
            self._prepared = (("stale",), self._prepared[1])
        if getattr(self, "_tables", None) is not None:
            self._tables = (("stale",), self._tables[1])

    def _weights(self) -> _lib.MelWeights:
        key = self._param_key()
        cached = getattr(self, "_w_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        for name, p in self.named_parameters():
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError(f"parameter {name} must be contiguous float32 for the HIP path")
        w = _lib.MelWeights()
        w.model, w.in_dim, w.n_actions = self._MODEL, self.input_dim, self.output_dim
        _mlp(w.encoder, self.encoder.linears())
        _gat(w.conv1, self.conv1)
        if self._MODEL != _lib.MODEL_HLDGN:
            _gat(w.conv2, self.conv2)
        if self.use_dueling:
            w.dueling = 1
            _mlp(w.q_head, self.Q.linears())
            _mlp(w.v_head, self.V.linears())
        else:
            w.dueling = 0
            _mlp(w.q_head, [self.out_linear])
            _mlp(w.v_head, [self.out_linear])     # ignored by the kernels when dueling == 0
            w.v_head.layer[0].out_dim = 1
        w.precision = {"f32": _lib.PREC_F32, "bf16": _lib.PREC_BF16, "f32s": _lib.PREC_F32_SPLIT,
                       "f32a": _lib.PREC_F32_AUTO}[self.feature_dtype]
        self._w_cache = (key, w)
        return w

    def _workspace(self, w, bs: int, device) -> torch.Tensor:
        lib = _lib.load()
        need = int(lib.mel_workspace_bytes(C.byref(w), bs, self.agents_num))
        if need == 0:
            raise RuntimeError("mel_workspace_bytes rejected the configuration")
        ws = getattr(self, "_ws", None)
        if ws is None or ws.numel() < need or ws.device != device:
            ws = torch.empty(need, dtype=torch.uint8, device=device)
            self._ws = ws
        return ws

    def hip_forward(self, obs: torch.Tensor, out: torch.Tensor | None = None, integer_features: bool = False) -> torch.Tensor:
        """Inference through libmelissa_hip.so on the current HIP stream.  obs: CUDA float32 [bs, 8N+1].
        ``integer_features``: the caller guarantees the observations come from the env (node features are the integers
        GraphEnv writes), which lets large batches go through the node-feature table (MEL_FWD_INTEGER_FEATURES)."""
        lib = _lib.load()
        if not obs.is_cuda:
            raise RuntimeError("the HIP forward needs the observation on a ROCm device; there is no CPU "
                               "fallback (use torch_forward for autograd on any device)")
        obs = obs.contiguous()
        if obs.dtype != torch.float32:
            obs = obs.float()
        bs = obs.shape[0]
        if bs == 0:             # an empty batch has empty logits (nothing to launch), as the torch ops of the reference give
            return out if out is not None else torch.empty(0, self.output_dim, dtype=torch.float32, device=obs.device)
        w = self._weights()
        self._refresh_prepared(w, obs.device)
        self._refresh_tables(w, obs.device)
        ws = self._workspace(w, bs, obs.device)
        if out is None:
            out = torch.empty(bs, self.output_dim, dtype=torch.float32, device=obs.device)
        stream = _lib.current_stream_ptr(obs.device)
        w.flags = _lib.FWD_INTEGER_FEATURES if integer_features else 0
        if self._MODEL in (_lib.MODEL_LDGN, _lib.MODEL_DGNR):
            fn = lib.mel_ldgn_forward if self._MODEL == _lib.MODEL_LDGN else lib.mel_dgnr_forward
            st = fn(C.byref(w), obs.data_ptr(), bs, self.agents_num, obs.shape[1], out.data_ptr(), ws.data_ptr(),
                    ws.numel(), stream)
        else:
            st = lib.mel_hldgn_forward(C.byref(w), _lib.AGG[self.aggregator_name], obs.data_ptr(), bs,
                                       self.agents_num, obs.shape[1], out.data_ptr(), ws.data_ptr(),
                                       ws.numel(), stream)
        w.flags = 0
        _lib.check(st, type(self).__name__ + ".forward")
        return out

    def hip_tap(self, kind: int, bs: int, rows_cap: int = 0, workspace: torch.Tensor | None = None) -> torch.Tensor:
        """Parity tap after a HIP forward: 0 = adjacency masks [bs, N] int64 bit patterns ([bs, N, 2] beyond 64 nodes), 1 = head input
        [rows, latent], 2 = int32 [3] rows processed (sum|U1|, sum|U2|, agent rows).  ``rows_cap`` = 0 after
        ``hip_forward``, else the cap given to ``hip_forward_agents``."""
        lib = _lib.load()
        w = self._weights()
        ws = workspace if workspace is not None else (self._ws_agents if rows_cap else self._ws)
        if kind == 0:
            out = torch.empty(bs, self.agents_num, *(() if self.agents_num <= 64 else (_lib.set_words(self.agents_num),)),
                              dtype=torch.int64, device=ws.device)
        elif kind == 1:
            out = torch.empty(rows_cap or bs, w.q_head.layer[0].in_dim, device=ws.device,
                              dtype=torch.bfloat16 if self.feature_dtype == "bf16" else torch.float32)
        elif kind == 3:           # [0] node-feature table rows the last forward used, [1 + b] out-of-range feature flags
            out = torch.zeros(1 + bs, dtype=torch.int32, device=ws.device)
        else:
            out = torch.zeros(3, dtype=torch.int32, device=ws.device)
        _lib.check(lib.mel_forward_tap(C.byref(w), kind, bs, self.agents_num, rows_cap, ws.data_ptr(), out.data_ptr(),
                                       _lib.current_stream_ptr(ws.device)), "mel_forward_tap")
        return out

    def plan_pointers(self, bs: int, rows_cap: int, workspace: torch.Tensor):
        """Device addresses (adj, live, u1, u2, cnt) of the plan masks inside ``workspace`` for a forward of these dimensions
        (``rows_cap`` = 0: the hip_forward / hip_forward_envs layout) - the values of mel_env_batch.plan_*."""
        out = (C.c_void_p * 5)()
        _lib.check(_lib.load().mel_plan_pointers(C.byref(self._weights()), bs, self.agents_num, rows_cap, workspace.data_ptr(),
                                                 out), "mel_plan_pointers")
        return [out[i] for i in range(5)]

    def hip_forward_envs(self, obs_matrix: torch.Tensor, out: torch.Tensor | None = None,
                         workspace: torch.Tensor | None = None, select: "_lib.MelSelect | None" = None,
                         plan_ready: bool = False, integer_features: bool = False) -> torch.Tensor:
        """HL-DGN on env rows without an index column (round-batched loop): one logits row per env.  ``select`` (with
        ``live`` / ``n_nodes`` set): the per-(env, agent) argmax / eps-greedy fused into the launch that writes the logits."""
        if self._MODEL != _lib.MODEL_HLDGN:
            raise RuntimeError("hip_forward_envs is the HL-DGN entry point")
        lib = _lib.load()
        assert obs_matrix.is_cuda and obs_matrix.dtype == torch.float32 and obs_matrix.stride(-1) == 1
        bs = obs_matrix.shape[0]
        w = self._weights()
        self._refresh_prepared(w, obs_matrix.device)
        self._refresh_tables(w, obs_matrix.device)
        ws = workspace if workspace is not None else self._workspace(w, bs, obs_matrix.device)
        if out is None:
            out = torch.empty(bs, self.output_dim, dtype=torch.float32, device=obs_matrix.device)
        # (the struct is cached: always set, never left behind)
        w.flags = (_lib.FWD_PLAN_READY if plan_ready else 0) | (_lib.FWD_INTEGER_FEATURES if integer_features else 0)
        if select is not None:
            st = lib.mel_hldgn_forward_envs_select(C.byref(w), _lib.AGG[self.aggregator_name], obs_matrix.data_ptr(), bs,
                                                   self.agents_num, obs_matrix.stride(0), out.data_ptr(), C.byref(select),
                                                   ws.data_ptr(), ws.numel(), _lib.current_stream_ptr(obs_matrix.device))
            w.flags = 0
            _lib.check(st, "mel_hldgn_forward_envs_select")
            return out
        st = lib.mel_hldgn_forward_envs(C.byref(w), _lib.AGG[self.aggregator_name], obs_matrix.data_ptr(), bs,
                                        self.agents_num, obs_matrix.stride(0), out.data_ptr(), ws.data_ptr(),
                                        ws.numel(), _lib.current_stream_ptr(obs_matrix.device))
        w.flags = 0
        _lib.check(st, "mel_hldgn_forward_envs")
        return out

    def agents_workspace_bytes(self, bs: int, rows_cap: int) -> int:
        if self._MODEL == _lib.MODEL_HLDGN:
            return int(_lib.load().mel_workspace_bytes(C.byref(self._weights()), bs, self.agents_num))
        return int(_lib.load().mel_workspace_bytes_agents(C.byref(self._weights()), bs, self.agents_num, rows_cap))

    def hip_forward_agents(self, obs_matrix: torch.Tensor, agent_mask: torch.Tensor, rows_cap: int,
                           out: torch.Tensor | None = None, row_offsets: torch.Tensor | None = None,
                           select: "_lib.MelSelect | None" = None, workspace: torch.Tensor | None = None,
                           plan_ready: bool = False, integer_features: bool = False):
        """L-DGN for a set of controlling agents per env (round-batched loop).  ``obs_matrix``: CUDA fp32
        [bs, >= 8N] (row b = env b's obs_matrix, any row stride), ``agent_mask``: CUDA int64 [bs] bit
        patterns ([bs, 2] words beyond 64 nodes).  Returns (logits [rows_cap, A] - rows ordered by env then agent id -, row_offsets [bs+1])."""
        if self._MODEL == _lib.MODEL_HLDGN:
            raise RuntimeError("hip_forward_agents is the L-DGN / DGN-R entry point")
        lib = _lib.load()
        assert obs_matrix.is_cuda and obs_matrix.dtype == torch.float32 and obs_matrix.stride(-1) == 1
        bs = obs_matrix.shape[0]
        w = self._weights()
        self._refresh_prepared(w, obs_matrix.device)
        self._refresh_tables(w, obs_matrix.device)
        need = int(lib.mel_workspace_bytes_agents(C.byref(w), bs, self.agents_num, rows_cap))
        if workspace is not None:                  # caller-owned scratch (one per concurrent stream)
            ws = workspace
            if ws.numel() < need:
                raise RuntimeError(f"workspace of {ws.numel()} bytes < {need} needed")
        else:
            ws = getattr(self, "_ws_agents", None)
            if ws is None or ws.numel() < need or ws.device != obs_matrix.device:
                ws = torch.empty(need, dtype=torch.uint8, device=obs_matrix.device)
                self._ws_agents = ws
        if out is None:
            out = torch.empty(rows_cap, self.output_dim, dtype=torch.float32, device=obs_matrix.device)
        if row_offsets is None:
            row_offsets = torch.empty(bs + 1, dtype=torch.int32, device=obs_matrix.device)
        fn = lib.mel_ldgn_forward_agents if self._MODEL == _lib.MODEL_LDGN else lib.mel_dgnr_forward_agents
        # plan_ready: mel_env_round's plan sink wrote this call's masks; integer_features: obs rows come from the env
        w.flags = (_lib.FWD_PLAN_READY if plan_ready else 0) | (_lib.FWD_INTEGER_FEATURES if integer_features else 0)
        st = fn(C.byref(w), obs_matrix.data_ptr(), bs, self.agents_num, obs_matrix.stride(0), agent_mask.data_ptr(),
                rows_cap, out.data_ptr(), row_offsets.data_ptr(), C.byref(select) if select is not None else None,
                ws.data_ptr(), ws.numel(), _lib.current_stream_ptr(obs_matrix.device))
        w.flags = 0
        _lib.check(st, "mel_ldgn_forward_agents")
        return out, row_offsets

    def _prepare_obs(self, obs):
        if isinstance(obs, np.ndarray):                       # l_dgn.py:103-104
            obs = torch.as_tensor(obs, device=self.device)
        check_obs(obs, self.input_dim, self.agents_num)
        return obs

    def _dispatch(self, obs: torch.Tensor) -> torch.Tensor:
        """HIP kernels for inference; the autograd formulation when a gradient is required."""
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if self.backend == "torch" or (self.backend == "auto" and needs_grad):
            return self.torch_forward(obs)
        return self.hip_forward(obs.to(self.device))


# ---------------------------------------------------------------------------------------------------
# autograd formulation (learn path).  Dense per-graph masked attention in plain torch ops.
# ---------------------------------------------------------------------------------------------------
def radius_adjacency(pos: torch.Tensor) -> torch.Tensor:
    """adj[b, i, j]: edge j -> i.  fp32, strict <, first 33 hits per target."""
    r2 = torch.tensor(RADIUS_OF_INFLUENCE * RADIUS_OF_INFLUENCE, dtype=torch.float64).to(torch.float32).to(pos.device)
    dx = pos[:, :, None, 0] - pos[:, None, :, 0]
    dy = pos[:, :, None, 1] - pos[:, None, :, 1]
    within = (dx * dx + dy * dy) < r2
    within = within & (torch.cumsum(within.to(torch.int32), dim=2) <= MAX_NUM_NEIGHBORS + 1)
    eye = torch.eye(pos.shape[1], dtype=torch.bool, device=pos.device)
    return within & ~eye


def gatv2_dense(conv: GATv2Conv, x: torch.Tensor, adj: torch.Tensor) -> torch.Tensor:
    bs, n, _ = adj.shape
    h, c = conv.heads, conv.out_channels
    x_l = conv.lin_l(x).view(bs, n, h, c)
    x_r = conv.lin_r(x).view(bs, n, h, c)
    mask = adj | torch.eye(n, dtype=torch.bool, device=adj.device)[None]
    e = (F.leaky_relu(x_r[:, :, None] + x_l[:, None, :], 0.2) * conv.att[0]).sum(-1)     # [bs, i, j, H]
    e = e.masked_fill(~mask[..., None], -float("inf"))
    p = torch.exp(e - e.max(dim=2, keepdim=True).values)
    p = torch.where(mask[..., None], p, torch.zeros((), device=p.device))
    alpha = p / (p.sum(dim=2, keepdim=True) + 1e-16)
    out = torch.einsum("bijh,bjhc->bihc", alpha, x_l)
    return out.reshape(bs * n, h * c) + conv.bias


def transformer_dense(conv: TransformerConv, x: torch.Tensor, adj: torch.Tensor) -> torch.Tensor:
    bs, n, _ = adj.shape
    h, c = conv.heads, conv.out_channels
    q = conv.lin_query(x).view(bs, n, h, c)
    k = conv.lin_key(x).view(bs, n, h, c)
    v = conv.lin_value(x).view(bs, n, h, c)
    e = torch.einsum("bihc,bjhc->bijh", q, k) / math.sqrt(c)
    e = e.masked_fill(~adj[..., None], -float("inf"))
    e_max = e.max(dim=2, keepdim=True).values
    e_max = torch.where(torch.isfinite(e_max), e_max, torch.zeros((), device=e.device))
    p = torch.where(adj[..., None], torch.exp(e - e_max), torch.zeros((), device=e.device))
    alpha = p / (p.sum(dim=2, keepdim=True) + 1e-16)
    return torch.einsum("bijh,bjhc->bihc", alpha, v).reshape(bs * n, h * c)


def use_hip_autograd(module, obs: torch.Tensor) -> bool:
    """Learn path on a ROCm device: edge softmax / aggregation / pool run as HIP kernels with hand-written
    backward (autograd_ops.py) unless the module opts out with ``learn_kernels = "dense"``; CPU tensors always take
    the dense torch formulation."""
    return obs.is_cuda and getattr(module, "learn_kernels", "hip") == "hip"


def learn_adjacency(obs: torch.Tensor, pos: torch.Tensor, n: int, input_dim: int, hip: bool):
    if hip:
        from .autograd_ops import radius_graph
        return radius_graph(obs.float().contiguous(), n, input_dim)
    return radius_adjacency(pos)


def linear(lin: nn.Linear, x: torch.Tensor, hip: bool, relu: bool = False) -> torch.Tensor:
    """``lin(x)`` (``relu``: ``F.relu(lin(x))``) - on the learn path of a ROCm device through the library's own fp32-MFMA GEMM,
    forward and backward (autograd_ops.hip_linear; the ReLU in the GEMM's epilogue), for every layer its tiling takes;
    ``F.linear`` otherwise (CPU tensors, ``learn_kernels = "dense"``, the 5-wide encoder input and the 2- / 1-wide last head
    layers)."""
    if hip and x.dim() == 2 and x.dtype == torch.float32:
        from .autograd_ops import hip_linear, hip_linear_supported
        if hip_linear_supported(lin.in_features, lin.out_features):
            return hip_linear(x, lin.weight, lin.bias, relu)
    return F.relu(lin(x)) if relu else lin(x)


def mlp(module: "MLP", x: torch.Tensor, hip: bool, final_relu: bool = False) -> torch.Tensor:
    """``module.model(x)`` with its Linear layers routed through :func:`linear`; a Linear followed by ``nn.ReLU`` is one fused
    call.  ``final_relu``: ``F.relu`` of the result (the callers' ``F.relu(mlp(...))``), fused into the last Linear."""
    layers = list(module.model)
    i = 0
    while i < len(layers):
        layer = layers[i]
        if isinstance(layer, nn.Linear):
            nxt_relu = i + 1 < len(layers) and type(layers[i + 1]) is nn.ReLU
            last = i == len(layers) - 1
            x = linear(layer, x, hip, relu=nxt_relu or (last and final_relu))
            if last and final_relu:
                final_relu = False
            i += 2 if nxt_relu else 1
        else:
            x = layer(x)
            i += 1
    return F.relu(x) if final_relu else x


def conv_relu(conv, x: torch.Tensor, adj, n: int, hip: bool) -> torch.Tensor:
    """relu(conv(x)) for a GATv2Conv / TransformerConv over full graphs (adj: uint64 masks when hip, else dense)."""
    if hip:
        from .autograd_ops import gat_attention, transformer_attention
        if isinstance(conv, TransformerConv):
            return transformer_attention(linear(conv.lin_key, x, hip), linear(conv.lin_value, x, hip),
                                         linear(conv.lin_query, x, hip), adj, n, conv.heads, conv.out_channels)
        return gat_attention(linear(conv.lin_l, x, hip), linear(conv.lin_r, x, hip), conv.att, conv.bias, adj, n, conv.heads,
                             conv.out_channels)
    dense = transformer_dense if isinstance(conv, TransformerConv) else gatv2_dense
    return F.relu(dense(conv, x, adj))


def unpack(obs: torch.Tensor, input_dim: int, n: int):
    bs = obs.shape[0]
    node = obs[:, :-1].reshape(bs, n, input_dim + 3).float()
    g = obs[:, -1].clamp(0, n - 1).long()
    return node[:, :, :2], node[:, :, 2:2 + input_dim], node[:, :, -1:], g


# ---------------------------------------------------------------------------------------------------
# shared body of the three drop-in networks (they differ in the conv type, the pooling and two details of how the
# reference's constructors treat ``dueling_param``)
# ---------------------------------------------------------------------------------------------------
class GraphQNetwork(HipForwardMixin, nn.Module):
    _RETURNS_STATE = True          # forward returns (logits, state); L-DGN returns (logits, None) (l_dgn.py:151)

    def _setup(self, input_dim, hidden_dim, output_dim, num_heads, agents_num, device, edge_attributes, backend):
        _lib.check_n_nodes(agents_num, type(self).__name__)
        self.device, self.backend = device, backend                     # backend: "auto" | "hip" | "torch"
        self.input_dim, self.hidden_dim, self.output_dim = input_dim, hidden_dim, output_dim
        self.num_heads, self.agents_num = num_heads, agents_num
        self.edge_attributes = edge_attributes                          # computed-but-unused in the reference
        self.encoder = MLP(input_dim=input_dim, hidden_sizes=[hidden_dim], output_dim=hidden_dim, device=device)

    def _build_heads(self, latent_dim: int, dueling_param, honour_output_dim_key: bool = False):
        """Dueling Q / V MLPs (the caller's kwargs dicts are MUTATED, as in the reference) or one ``out_linear``."""
        self.use_dueling = dueling_param is not None
        if not self.use_dueling:
            self.out_linear = nn.Linear(latent_dim, self.output_dim)
            return
        q_kwargs, v_kwargs = dueling_param
        q_out, v_out = self.output_dim, 1
        if honour_output_dim_key:                                       # only LDGNNetwork pops it (l_dgn.py:71-84)
            q_out, v_out = q_kwargs.pop("output_dim", q_out), v_kwargs.pop("output_dim", v_out)
        q_kwargs.update({"input_dim": latent_dim, "output_dim": q_out, "device": self.device})
        v_kwargs.update({"input_dim": latent_dim, "output_dim": v_out, "device": self.device})
        self.Q, self.V = MLP(**q_kwargs), MLP(**v_kwargs)
        self.output_dim = q_out

    def _head(self, latent: torch.Tensor, hip: bool = False) -> torch.Tensor:
        if self.use_dueling:
            q, v = mlp(self.Q, latent, hip), mlp(self.V, latent, hip)
            return q - q.mean(dim=1, keepdim=True) + v
        return linear(self.out_linear, latent, hip)

    def forward(self, obs, state=None, info={}):
        logits = self._dispatch(self._prepare_obs(obs))
        return logits, (state if self._RETURNS_STATE else None)

    def torch_forward_all_agents(self, obs_matrix: torch.Tensor) -> torch.Tensor:
        """Q values of EVERY node of every graph as controlling agent: ``obs_matrix`` [G, 8N] (a round's shared observation,
        graph.py:186-188: the rows of a round's agents differ only in the index column) -> logits [G, N, A] with
        ``[g, j] == torch_forward(cat(obs_matrix[g], j))`` - encoder and both convolutions run ONCE per graph (they do not
        depend on the controlling agent, l_dgn.py:117-135 / dgn_r.py:95-125), only the head runs per (graph, agent).  The learn
        path of DGN-R (policies/dgn.py:31-55 sums Q over all agents that acted in the sampled round) evaluates one graph per
        sampled experience instead of one per sibling, with static shapes."""
        if not hasattr(self, "conv2"):
            raise RuntimeError("torch_forward_all_agents is for the two-convolution networks (L-DGN, DGN-R)")
        obs_matrix = obs_matrix.to(self.device)
        G, n = obs_matrix.shape[0], self.agents_num
        node = obs_matrix.reshape(G, n, self.input_dim + 3).float()
        pos, feats, dm = node[:, :, :2], node[:, :, 2:2 + self.input_dim], node[:, :, -1:]
        hip = use_hip_autograd(self, obs_matrix)
        # (the adjacency kernels take observation rows with an index column: any value, it is not read)
        adj = learn_adjacency(torch.cat([obs_matrix.float(), obs_matrix.new_zeros(G, 1, dtype=torch.float32)], dim=1), pos, n,
                              self.input_dim, hip)
        x_1 = mlp(self.encoder, feats.reshape(G * n, -1), hip, final_relu=True)
        x_2 = conv_relu(self.conv1, x_1, adj, n, hip)
        x_3 = conv_relu(self.conv2, x_2 * dm.reshape(G * n, 1), adj, n, hip)
        return self._head(torch.cat([x_1, x_2, x_3], dim=1), hip).view(G, n, -1)

    def _two_conv_torch_forward(self, obs: torch.Tensor) -> torch.Tensor:
        """encoder -> conv1 -> dm mask -> conv2 with the controlling agent's rows gathered after each stage
        (l_dgn.py:117-149, dgn_r.py:95-127) in differentiable ops; on ROCm devices the convolutions' attention runs in
        the HIP kernels of csrc/grad.hip."""
        obs = obs.to(self.device)
        pos, feats, dm, g = unpack(obs, self.input_dim, self.agents_num)
        bs, n = pos.shape[:2]
        hip = use_hip_autograd(self, obs)
        adj = learn_adjacency(obs, pos, n, self.input_dim, hip)
        x = mlp(self.encoder, feats.reshape(bs * n, -1), hip, final_relu=True)
        gi = torch.arange(bs, device=x.device) * n + g
        x_1 = x[gi]
        x = conv_relu(self.conv1, x, adj, n, hip)
        x_2 = x[gi]
        x = conv_relu(self.conv2, x * dm.reshape(bs * n, 1), adj, n, hip)
        return self._head(torch.cat([x_1, x_2, x[gi]], dim=1), hip)
