"""Multi-GPU layer: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The hot path shards by env: envs are independent (l_dgn.py:137-146 builds independent envs; the forward
is row-independent given replicated weights), so each rank steps and evaluates its own contiguous block
of envs with NO data-path collective.  The one real exchange is the gradient step of the DGN learner:
a single flat fp32 buffer (L-DGN 1,005,315 / HL-DGN 315,139 parameters) is summed with
ONE all-reduce per update and divided by the world size, then every replica takes the same Adam step.
At 1-4 MB the collective is latency bound, so it is a single call on one flat buffer, never per tensor.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_distributed(backend: str | None = None):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, local_rank, world).  A single process (no WORLD_SIZE) needs no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            # RCCL: bind the communicator to this rank's GPU up front (no lazy "guess the device" at the first collective)
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(total: int, world: int, rank: int):
    """Contiguous block of env ids owned by ``rank`` (env k -> rank k // ceil(total / world))."""
    per = (total + world - 1) // world
    lo = min(rank * per, total)
    return lo, min(lo + per, total)


class FlatGradAllReducer:
    """Sum-then-average all the gradients of ``model`` with one collective on one flat fp32 buffer."""

    def __init__(self, model: torch.nn.Module):
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=ref.device)

    @staticmethod
    def active() -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def __call__(self, model=None):
        if not self.active():
            return
        self.pack()
        self.reduce()
        self.unpack()

    # the three phases on their own: a captured update (melissa_amd.replay.CapturedUpdate) replays pack and unpack from HIP
    # graphs and issues the collective between them eagerly
    def pack(self):
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n

    def reduce(self):
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(dist.get_world_size())

    def unpack(self):
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = self.flat[off:off + n].view_as(p).clone()
            else:
                p.grad.copy_(self.flat[off:off + n].view_as(p))
            off += n


def broadcast_parameters(model: torch.nn.Module, src: int = 0):
    """Make every replica start from rank ``src``'s weights (one flat broadcast)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    params = list(model.parameters())
    flat = torch.cat([p.data.reshape(-1).float() for p in params])
    dist.broadcast(flat, src=src)
    off = 0
    for p in params:
        n = p.numel()
        p.data.copy_(flat[off:off + n].view_as(p))
        off += n


def _scalar_device(device):
    """gloo reduces host tensors, nccl (RCCL) device tensors."""
    return torch.device("cpu") if dist.get_backend() == "gloo" else device


def all_reduce_max(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_scalar_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_reduce_sum(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_scalar_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
