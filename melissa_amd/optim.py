"""The optimizer step of the learn path on the device in one launch.

The reference trains with ``torch.optim.Adam(net.parameters(), lr=...)`` (l_dgn.py:207, hl_dgn.py, dgn_r.py).  ``adam_step(opt)``
performs exactly that optimizer's update ON ITS OWN STATE (``opt.state[p]["exp_avg" | "exp_avg_sq" | "step"]`` stay torch's
tensors: ``state_dict()`` / ``load_state_dict()`` keep working, a twin ``torch.optim.Adam`` continues from them) through
``mel_adam_step``: every parameter tensor of a group in one launch.  torch's own step is ~25 launches eagerly and ~200 in the
capturable form a HIP-graph capture needs (the bias corrections become a dozen one-element launches per parameter tensor), two
thirds of a captured DQN update.  Anything the kernel does not cover (amsgrad, maximize, sparse or non-fp32 gradients, a
parameter without state yet, CPU tensors) goes through ``opt.step()`` unchanged.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


# The kernel writes the parameters through raw pointers: torch's version counters (what every "did the weights change?" cache
# in this package and in autograd keys on) are advanced by hand afterwards, as an in-place torch op would have done.
_set_versions = getattr(torch._C._autograd, "_unsafe_set_version_counter", None)


def _eligible(opt) -> bool:
    return type(opt) is torch.optim.Adam and _set_versions is not None and not getattr(opt, "_mel_disable_fused", False)


def adam_step(opt) -> None:
    if not _eligible(opt):
        opt.step()
        return
    plan = []
    for group in opt.param_groups:
        if group.get("amsgrad") or group.get("maximize") or group.get("differentiable"):
            opt.step()
            return
        tensors = []
        for p in group["params"]:
            if p.grad is None:
                continue
            st = opt.state.get(p)
            if (not st or "exp_avg" not in st or not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32
                    or p.grad.is_sparse or not p.is_contiguous() or not p.grad.is_contiguous()
                    or not torch.is_tensor(st.get("step"))):
                opt.step()                    # first update (torch creates the state), or a layout the kernel does not take
                return
            tensors.append((p, st))
        plan.append((group, tensors))
    lib = _lib.load()
    for group, tensors in plan:
        if not tensors:
            continue
        on_device = tensors[0][1]["step"].is_cuda
        if any(st["step"].is_cuda != on_device for _, st in tensors):
            opt.step()
            return
        beta1, beta2 = group["betas"]
        lr = group["lr"]
        lr = float(lr) if not torch.is_tensor(lr) else float(lr.item())
        host_step = 0.0
        if not on_device:                     # torch's eager form: the counters are host tensors, advanced here
            for _, st in tensors:
                st["step"] += 1
            host_step = float(tensors[0][1]["step"])
            if any(float(st["step"]) != host_step for _, st in tensors):
                for _, st in tensors:
                    st["step"] -= 1
                opt.step()
                return
        stream = _lib.current_stream_ptr(tensors[0][0].device)
        for at in range(0, len(tensors), _lib.ADAM_MAX_TENSORS):
            chunk = tensors[at: at + _lib.ADAM_MAX_TENSORS]
            t = _lib.MelAdamTensors()
            t.count = len(chunk)
            for i, (p, st) in enumerate(chunk):
                t.param[i], t.grad[i] = p.data_ptr(), p.grad.data_ptr()
                t.exp_avg[i], t.exp_avg_sq[i] = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                t.step[i] = st["step"].data_ptr() if on_device else None
                t.numel[i] = p.numel()
            _lib.check(lib.mel_adam_step(C.byref(t), lr, beta1, beta2, group["eps"], group["weight_decay"], host_step, stream),
                       "mel_adam_step")
        changed = [p for p, _ in tensors]
        _set_versions(changed, [p._version + 1 for p in changed])
