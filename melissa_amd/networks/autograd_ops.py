"""torch.autograd.Function wrappers of the learn-path HIP kernels (csrc/grad.hip).

``gat_attention``  relu(GATv2Conv / TransformerConv) given the dense projections, forward AND backward in HIP
                   (edge softmax + aggregation; l_dgn.py:125-126,133-134, hl_dgn.py:101-102, dgn_r.py:103-113);
``graph_pool``     (x * dm) -> global max / mean / add pool with its backward (hl_dgn.py:105-108);
``radius_graph``   the fp32 radius adjacency as uint64 source masks (networks/common.py:47-48), no gradient.

``hip_linear``     every dense projection of the learn path (lin_l / lin_r / key / query / value, the encoder's second
                   layer, the heads' hidden layers) on the library's own fp32-MFMA GEMM (``mel_gemm_f32``), forward and
                   backward: y = x W^T + b, dX = dY W, dW = dY^T X as three launches of the same kernel (the operands that
                   are not contiguous along the contraction index go through ``mel_transpose_f32`` first), db = column
                   sums.  Layers the kernel's tiling does not take (K % 32 or N % 64: the 5-wide encoder input, the 2- /
                   1-wide last head layers - 0.1 % of the FLOPs) stay ``F.linear``.

These ops need a ROCm device and fail loudly without the library: no CPU fallback (the CPU autograd formulation is
``gatv2_dense`` / ``transformer_dense`` + ``nn.Linear`` in common.py).
"""
from __future__ import annotations

import os

import torch

from .. import _lib


def _stream(t: torch.Tensor):
    return _lib.current_stream_ptr(t.device)


def radius_graph(obs: torch.Tensor, n_nodes: int, in_dim: int) -> torch.Tensor:
    """obs: CUDA fp32 [bs, >= n*(in_dim+3)] -> int64 [bs*n] bit patterns (bit j = node j is a source); [bs*n, 2] words
    beyond 64 nodes (MEL_SET_WORDS)."""
    assert obs.is_cuda and obs.dtype == torch.float32 and obs.stride(-1) == 1
    bs = obs.shape[0]
    adj = torch.empty(bs * n_nodes, *(() if n_nodes <= 64 else (_lib.set_words(n_nodes),)), dtype=torch.int64, device=obs.device)
    _lib.check(_lib.load().mel_radius_graph(obs.data_ptr(), bs, n_nodes, obs.stride(0), in_dim, adj.data_ptr(), _stream(obs)),
               "mel_radius_graph")
    return adj


class _GatAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xl, xv, xr, att, bias, adj, n_nodes, heads, channels, kind):
        lib = _lib.load()
        xl, xr = xl.contiguous(), xr.contiguous()
        xv = xv.contiguous() if xv is not None else None
        rows = xl.shape[0]
        out = torch.empty(rows, heads * channels, dtype=torch.float32, device=xl.device)
        _lib.check(lib.mel_gat_forward(xl.data_ptr(), xv.data_ptr() if xv is not None else None, xr.data_ptr(),
                                       att.data_ptr() if att is not None else None,
                                       bias.data_ptr() if bias is not None else None, adj.data_ptr(), rows // n_nodes,
                                       n_nodes, heads, channels, kind, out.data_ptr(), _stream(xl)), "mel_gat_forward")
        ctx.save_for_backward(xl, xv, xr, att, adj, out)
        ctx.meta = (n_nodes, heads, channels, kind, bias is not None)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        xl, xv, xr, att, adj, out = ctx.saved_tensors
        n_nodes, heads, channels, kind, has_bias = ctx.meta
        grad_out = grad_out.contiguous()
        dxl, dxr = torch.empty_like(xl), torch.empty_like(xr)            # written, not accumulated (no atomics on rows)
        dxv = torch.empty_like(xv) if xv is not None else None
        # scratch: per (target, head) m, 1/l, S + the workgroups' partial sums of d att | d bias (MEL_GAT_PARTIAL_GROUPS rows)
        stats = torch.empty(xl.shape[0] * heads * 4 + 256 * 2 * heads * channels, dtype=torch.float32, device=xl.device)
        datt = torch.empty(heads * channels, dtype=torch.float32, device=xl.device) if att is not None else None
        dbias = torch.empty(heads * channels, dtype=torch.float32, device=xl.device) if has_bias else None
        p = lambda t: t.data_ptr() if t is not None else None
        _lib.check(lib.mel_gat_backward(xl.data_ptr(), p(xv), xr.data_ptr(), p(att), adj.data_ptr(), out.data_ptr(),
                                        grad_out.data_ptr(), xl.shape[0] // n_nodes, n_nodes, heads, channels, kind,
                                        dxl.data_ptr(), p(dxv), dxr.data_ptr(), p(datt), p(dbias), stats.data_ptr(), _stream(xl)),
                   "mel_gat_backward")
        if datt is not None:
            datt = datt.view_as(att)
        return dxl, dxv, dxr, datt, dbias, None, None, None, None, None


def gat_attention(xl, xr, att, bias, adj, n_nodes: int, heads: int, channels: int) -> torch.Tensor:
    """relu(GATv2 attention + bias): xl = lin_l(x) (sources), xr = lin_r(x) (targets), [bs*n, heads*channels]."""
    return _GatAttention.apply(xl, None, xr, att, bias, adj, n_nodes, heads, channels, _lib.CONV_GATV2)


def transformer_attention(k, v, q, adj, n_nodes: int, heads: int, channels: int) -> torch.Tensor:
    """relu(TransformerConv(root_weight=False)) given key / value / query projections."""
    return _GatAttention.apply(k, v, q, None, None, adj, n_nodes, heads, channels, _lib.CONV_TRANSFORMER)


class _GraphPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dm, n_nodes, aggregator):
        lib = _lib.load()
        x, dm = x.contiguous(), dm.contiguous().float()
        hc = x.shape[1]
        bs = x.shape[0] // n_nodes
        pooled = torch.empty(bs, hc, dtype=torch.float32, device=x.device)
        arg = torch.empty(bs, hc, dtype=torch.int32, device=x.device) if aggregator == 0 else None
        _lib.check(lib.mel_pool_forward(x.data_ptr(), dm.data_ptr(), bs, n_nodes, hc, aggregator, pooled.data_ptr(),
                                        arg.data_ptr() if arg is not None else None, _stream(x)), "mel_pool_forward")
        ctx.save_for_backward(dm, arg)
        ctx.meta = (bs, n_nodes, hc, aggregator)
        return pooled

    @staticmethod
    def backward(ctx, grad_pooled):
        lib = _lib.load()
        dm, arg = ctx.saved_tensors
        bs, n_nodes, hc, aggregator = ctx.meta
        grad_pooled = grad_pooled.contiguous()
        dx = torch.empty(bs * n_nodes, hc, dtype=torch.float32, device=grad_pooled.device)
        _lib.check(lib.mel_pool_backward(grad_pooled.data_ptr(), dm.data_ptr(), arg.data_ptr() if arg is not None else None,
                                         bs, n_nodes, hc, aggregator, dx.data_ptr(), _stream(grad_pooled)), "mel_pool_backward")
        return dx, None, None, None


def graph_pool(x: torch.Tensor, dm: torch.Tensor, n_nodes: int, aggregator: str) -> torch.Tensor:
    """x [bs*n, HC], dm [bs*n] decision-maker flags (no gradient) -> [bs, HC]."""
    return _GraphPool.apply(x, dm.reshape(-1), n_nodes, _lib.AGG[aggregator])


def _gemm(a: torch.Tensor, w: torch.Tensor, bias, out: torch.Tensor, relu: bool = False):
    """out[m, n] = sum_k a[m, k] * w[n, k] (+ bias[n]); a [M, lda >= K], w [N, K] contiguous, out [M, N] contiguous."""
    m, k = a.shape[0], w.shape[1]
    _lib.check(_lib.load().mel_gemm_f32(a.data_ptr(), a.stride(0), w.data_ptr(), bias.data_ptr() if bias is not None else None,
                                        out.data_ptr(), out.stride(0), m, w.shape[0], k, int(relu), 0, _stream(a)),
               "mel_gemm_f32")
    return out


_COPIES = bool(os.environ.get("MEL_BACKWARD_TRANSPOSED_COPIES"))      # A/B switch: round 2's transposed-copy form


def _gemm_t(a: torch.Tensor, a_t: bool, w: torch.Tensor, out: torch.Tensor, m: int, n: int, k: int):
    """out[m, n] = sum_k A'[m, k] W'[n, k] with W' = w^T read in place and, ``a_t``, A' = a^T likewise (mel_gemm_f32_t): the backward
    products of a linear layer without transposed copies."""
    _lib.check(_lib.load().mel_gemm_f32_t(a.data_ptr(), a.stride(0), int(a_t), w.data_ptr(), w.stride(0), 1, out.data_ptr(),
                                          out.stride(0), m, n, k, _stream(a)), "mel_gemm_f32_t")
    return out


def _gemm_splitk(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, ksplit: int):
    """``_gemm`` with the contraction cut into ``ksplit`` chunks (mel_gemm_f32_splitk): few output tiles, long K."""
    m, k, n = a.shape[0], w.shape[1], w.shape[0]
    parts = torch.empty(ksplit * m * n, dtype=torch.float32, device=a.device)
    _lib.check(_lib.load().mel_gemm_f32_splitk(a.data_ptr(), a.stride(0), w.data_ptr(), None, out.data_ptr(), out.stride(0), m, n, k,
                                               0, ksplit, parts.data_ptr(), parts.numel(), _stream(a)), "mel_gemm_f32_splitk")
    return out


def _weight_grad(dy: torch.Tensor, x: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """dW [N, K] = dY^T [N, M] . X [M, K]: the contraction runs over the M rows of the batch.  N x K is a few dozen 64 x 64
    tiles however long the batch is, so from a few thousand rows on the contraction is cut into chunks that run as
    independent work items (M padded with zero rows to a multiple of 32 * ksplit)."""
    m = dy.shape[0]
    tiles = (like.shape[0] // 64) * (like.shape[1] // 64)
    ksplit = 1
    if m >= 2048 and tiles < 512:
        ksplit = min(16, max(2, 1024 // tiles), m // 64)          # ~1 000 work items, at least two 32-steps per chunk
    if ksplit < 2:
        return _gemm(_transpose(dy, 32), _transpose(x, 32), None, torch.empty_like(like))
    return _gemm_splitk(_transpose(dy, 32 * ksplit), _transpose(x, 32 * ksplit), torch.empty_like(like), ksplit)


def _transpose(src: torch.Tensor, pad_to: int = 1) -> torch.Tensor:
    """[R, C] -> [C, Rp] with Rp = R rounded up to ``pad_to`` and zeros in the padding."""
    r, c = src.shape
    rp = (r + pad_to - 1) // pad_to * pad_to
    dst = (torch.zeros if rp != r else torch.empty)(c, rp, dtype=torch.float32, device=src.device)
    _lib.check(_lib.load().mel_transpose_f32(src.data_ptr(), src.stride(0), r, c, dst.data_ptr(), rp, _stream(src)),
               "mel_transpose_f32")
    return dst


def hip_linear_supported(in_features: int, out_features: int) -> bool:
    """mel_gemm_f32 tiles 64 output columns by 32 contraction steps; forward needs K % 32 == 0 and N % 64 == 0, the two
    backward products additionally K % 64 == 0 (dX has K columns, dW too)."""
    return in_features % 64 == 0 and out_features % 64 == 0


class _HipLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        x = x.contiguous()
        w = weight.contiguous()
        y = torch.empty(x.shape[0], w.shape[0], dtype=torch.float32, device=x.device)
        if x.shape[0]:
            _gemm(x, w, bias, y, relu=relu)                 # relu(x W^T + b) in the GEMM's epilogue: no separate launch
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        if y is not None:                                   # relu': the gradient of the rows' positive outputs
            dy = torch.ops.aten.threshold_backward(dy.contiguous(), y, 0.0)
        dy = dy.contiguous()
        m = x.shape[0]
        dx = dw = db = None
        if m == 0:
            return (torch.zeros_like(x), torch.zeros_like(w), torch.zeros(w.shape[0], device=w.device) if ctx.has_bias else None,
                    None)
        in_place = not _COPIES
        if ctx.needs_input_grad[0]:                       # dX [M, K] = dY [M, N] . W [N, K]: contraction over N, W read in place
            dx = (_gemm_t(dy, False, w, torch.empty_like(x), m, w.shape[1], w.shape[0]) if in_place
                  else _gemm(dy, _transpose(w), None, torch.empty_like(x)))
        if ctx.needs_input_grad[1]:                       # dW [N, K] = dY^T [N, M] . X [M, K]: contraction over M
            if in_place and m % 32 == 0 and m < 2048:     # both operands read transposed in place (mel_gemm_f32_t)
                dw = _gemm_t(dy, True, x, torch.empty_like(w), w.shape[0], w.shape[1], m)
            else:                                         # long batches: split-K over the rows (padded transposed copies)
                dw = _weight_grad(dy, x, w)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(dim=0)
        return dx, dw, db, None


def hip_linear(x: torch.Tensor, weight: torch.Tensor, bias, relu: bool = False) -> torch.Tensor:
    """``F.linear(x, weight, bias)`` - ``relu=True``: ``F.relu`` of it, in the same launch - on the library's fp32-MFMA GEMM with
    its own backward (x: CUDA fp32 [M, K])."""
    return _HipLinear.apply(x, weight, bias, relu)
