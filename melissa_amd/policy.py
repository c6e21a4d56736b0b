"""Policy side of the drop-in boundary.

* ``DQNPolicy``  - counterpart of [3P] tianshou 1.0.0 ``DQNPolicy`` as the reference uses it
  (l_dgn.py:70-78): ``forward`` = model -> mask illegal actions -> argmax,
  ``exploration_noise`` (eps-greedy), target network ``model_old`` (state_dict holds ``model.*`` and
  ``model_old.*`` like tianshou's), ``learn`` = n-step TD regression with Adam.
* ``MultiAgentSharedPolicy`` - counterpart of policies/multi_agent_managers/shared_policy.py:14-216.
  The reference fans the batch out per agent id (N python iterations, shared_policy.py:125-163) and
  stitches the actions back; all agents share ONE policy, so the result equals one forward over the
  whole batch - which is what this class does (same ``forward(batch) -> act`` contract).

Inference goes through the HIP kernels (network.hip_forward + mel_select_action); ``learn`` uses the
autograd formulation of the networks (backward kernels: SURVEY.md 8(f) #4).
"""
from __future__ import annotations

import copy
import ctypes as C
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .optim import adam_step


class Batch(dict):
    """Tiny attribute-dict standing in for tianshou's ``Batch`` at this boundary."""
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v


class DQNPolicy(nn.Module):
    def __init__(self, model: nn.Module, optim: Optional[torch.optim.Optimizer] = None, discount_factor: float = 0.99,
                 estimation_step: int = 1, target_update_freq: int = 0, is_double: bool = True,
                 clip_loss_grad: bool = False):
        super().__init__()
        self.model = model
        self.optim = optim
        self.eps = 0.0
        self._gamma, self._n_step = discount_factor, estimation_step
        self._target = target_update_freq > 0
        self._freq, self._iter = target_update_freq, 0
        self._is_double, self._clip_loss_grad = is_double, clip_loss_grad
        if self._target:
            self.model_old = copy.deepcopy(model)
            self.model_old.eval()
        self._scratch = None
        self.max_action_num = None

    def set_eps(self, eps: float):
        self.eps = eps

    def sync_weight(self):
        self.model_old.load_state_dict(self.model.state_dict())

    # ------------------------------------------------------------------ inference
    def compute_logits(self, obs) -> torch.Tensor:
        logits, _ = self.model(obs)
        return logits

    def select_action(self, logits: torch.Tensor, mask: torch.Tensor | None = None, eps: float = 0.0,
                      rand_u: torch.Tensor | None = None, rand_q: torch.Tensor | None = None,
                      out: torch.Tensor | None = None) -> torch.Tensor:
        """Device-side mask + argmax (+ eps-greedy when rand_u / rand_q are given)."""
        lib = _lib.load()
        bs, na = logits.shape
        if out is None:
            out = torch.empty(bs, dtype=torch.int32, device=logits.device)
        if self._scratch is None or self._scratch.device != logits.device:
            self._scratch = torch.empty(64, dtype=torch.float32, device=logits.device)
        _lib.check(lib.mel_select_action(
            logits.data_ptr(), mask.data_ptr() if mask is not None else None, bs, na, C.c_float(eps),
            rand_u.data_ptr() if rand_u is not None else None, rand_q.data_ptr() if rand_q is not None else None,
            out.data_ptr(), self._scratch.data_ptr(), _lib.current_stream_ptr(logits.device)), "mel_select_action")
        return out

    def forward(self, batch, state=None, model: str = "model", **kwargs) -> Batch:
        """batch.obs is either a dict/Batch with ``obs`` (+ ``mask``) or the raw observation array."""
        net = getattr(self, model)
        obs = batch["obs"] if isinstance(batch, dict) else batch.obs
        obs_next = obs["obs"] if isinstance(obs, dict) and "obs" in obs else obs
        logits, hidden = net(obs_next, state=state)
        mask = obs.get("mask") if isinstance(obs, dict) else None
        if logits.is_cuda:
            mask_t = None
            if mask is not None:
                mask_t = torch.as_tensor(np.asarray(mask), device=logits.device).to(torch.uint8).contiguous()
            act = self.select_action(logits, mask_t)
        else:                                            # learn-path helper on CPU tensors
            q = logits
            if mask is not None:
                m = torch.as_tensor(np.asarray(mask), dtype=q.dtype)
                q = q + (1 - m) * (q.min() - q.max() - 1.0)
            act = q.argmax(dim=1).to(torch.int32)
        return Batch(logits=logits, act=act, state=hidden)

    def exploration_noise(self, act: np.ndarray, batch) -> np.ndarray:
        """[3P] DQNPolicy.exploration_noise: numpy RNG, same call order as tianshou (A.5)."""
        if isinstance(act, np.ndarray) and not np.isclose(self.eps, 0.0):
            bsz = len(act)
            rand_mask = np.random.rand(bsz) < self.eps
            q = np.random.rand(bsz, self.max_action_num or 2)
            obs = batch["obs"] if isinstance(batch, dict) else batch.obs
            if isinstance(obs, dict) and obs.get("mask") is not None:
                q += np.asarray(obs["mask"])
            act[rand_mask] = q.argmax(axis=1)[rand_mask]
        return act

    # ------------------------------------------------------------------ learning (autograd path)
    def learn(self, batch, grad_hook=None) -> dict:
        """One DQN update on ``batch`` = dict(obs, act, returns[, weight]).  ``grad_hook(model)`` runs
        between backward and the optimizer step (the RCCL gradient all-reduce plugs in here)."""
        if self._target and self._iter % self._freq == 0:
            self.sync_weight()
        loss = self.loss_backward(batch)
        if grad_hook is not None:
            grad_hook(self.model)
        adam_step(self.optim)      # torch.optim.Adam's update on its own state, one launch (melissa_amd.optim)
        self._iter += 1
        return {"loss": float(loss)}

    def loss_backward(self, batch) -> torch.Tensor:
        """zero_grad + forward + loss + backward of one update; returns the detached loss (a device tensor: no host
        synchronisation here, so that a captured update - melissa_amd.replay.CapturedUpdate - can contain it)."""
        self.optim.zero_grad(set_to_none=True)
        with torch.enable_grad():
            logits, _ = self.model(batch["obs"])
            act = torch.as_tensor(batch["act"], device=logits.device, dtype=torch.long)
            q = logits[torch.arange(len(act), device=logits.device), act]
            returns = torch.as_tensor(batch["returns"], device=logits.device, dtype=q.dtype).flatten()
            td = returns - q
            loss = torch.nn.functional.huber_loss(q, returns) if self._clip_loss_grad else td.pow(2).mean()
            loss.backward()
        return loss.detach()


class DGNPolicy(DQNPolicy):
    """Counterpart of policies/dgn.py:21-71 (DGN-R training, BASELINE config 5).  The reference loss regresses
    every sampled experience's n-step return on the SUM of the Q values of all agents that acted in the same env
    round ("siblings": ``exp.info.indices`` holds their buffer indices, -1 for agents that did not act):

        q_j   = Q(active_obs_j)[act_j]                for every sibling j of experience i   (dgn.py:43-52)
        loss  = mean_i (returns_i - sum_j q_j)^2 * w   or   huber(sum_j q_j, returns_i)      (dgn.py:57-64)

    The reference evaluates the siblings with one small forward per experience (a 32-iteration Python loop,
    dgn.py:31-55); here ALL sibling observations of the batch go through ONE forward and a segment sum
    (SURVEY.md 8(f) #3).  Same loss value and gradients (tests/test_host_logic.py compares with the loop)."""

    def learn(self, batch, grad_hook=None) -> dict:
        """batch: ``returns`` [B] and EITHER the reference's row form - ``active_obs`` [M, 8N+1] (all sibling observations,
        index column = the sibling), ``active_act`` [M], ``segment`` [M] = experience each sibling row belongs to - OR the
        dense form ``obs_matrix`` [B, 8N], ``act_all`` [B, N], ``sibling`` [B, N] (who acted in the sampled round): the same
        loss, one graph evaluation per experience instead of one per sibling.  Optional ``weight``."""
        if self._target and self._iter % self._freq == 0:
            self.sync_weight()
        loss = self.loss_backward(batch)
        if grad_hook is not None:
            grad_hook(self.model)
        adam_step(self.optim)      # torch.optim.Adam's update on its own state, one launch (melissa_amd.optim)
        self._iter += 1
        return {"loss": float(loss)}

    def loss_backward(self, batch) -> torch.Tensor:
        """zero_grad + forward + DGN loss + backward; returns the detached loss (a device tensor, no host synchronisation:
        the dense form has static shapes and can be captured by replay.CapturedUpdate)."""
        self.optim.zero_grad(set_to_none=True)
        with torch.enable_grad():
            if "obs_matrix" in batch:
                q_all = self.model.torch_forward_all_agents(batch["obs_matrix"])               # [B, N, A]
                dev = q_all.device
                act = torch.as_tensor(batch["act_all"], device=dev, dtype=torch.long)
                sib = torch.as_tensor(batch["sibling"], device=dev).to(q_all.dtype)
                returns = torch.as_tensor(batch["returns"], device=dev, dtype=q_all.dtype).flatten()
                q = q_all.gather(2, act[..., None]).squeeze(2)                                  # [B, N]
                batch_q = (q * sib).sum(dim=1)                                                  # dgn.py:43-55
            else:
                logits, _ = self.model(batch["active_obs"])
                dev = logits.device
                act = torch.as_tensor(batch["active_act"], device=dev, dtype=torch.long)
                seg = torch.as_tensor(batch["segment"], device=dev, dtype=torch.long)
                returns = torch.as_tensor(batch["returns"], device=dev, dtype=logits.dtype).flatten()
                q = logits[torch.arange(len(act), device=dev), act]
                batch_q = torch.zeros_like(returns).index_add(0, seg, q)
            td = returns - batch_q
            if self._clip_loss_grad:
                loss = torch.nn.functional.huber_loss(batch_q.reshape(-1, 1), returns.reshape(-1, 1), reduction="mean")
            else:
                weight = batch.get("weight") if isinstance(batch, dict) else None
                if weight is None:                   # (no host -> device copy: this runs inside HIP-graph captures)
                    loss = td.pow(2).mean()
                else:
                    loss = (td.pow(2) * torch.as_tensor(weight, device=dev, dtype=td.dtype)).mean()
            loss.backward()
        if isinstance(batch, dict) and "obs_matrix" not in batch:
            batch["weight"] = td.detach()            # prio-buffer hook, dgn.py:66
        return loss.detach()

    @staticmethod
    def segments_from_indices(indices: np.ndarray, active_index: np.ndarray):
        """The reference's sibling lookup (dgn.py:31-47) as index arrays: ``indices`` [B, N] buffer indices of each
        experience's siblings (-1 = none), ``active_index`` [M] buffer index of every row of ``batch.active_obs``.
        Returns (gather [R], segment [R]): sibling row ``active_obs[gather[r]]`` contributes to experience
        ``segment[r]`` - the first matching row, like ``np.where(index == i)[0][0]``."""
        indices = np.asarray(indices)
        active_index = np.asarray(active_index)
        first = {}
        for pos, idx in enumerate(active_index.tolist()):
            first.setdefault(idx, pos)
        gather, segment = [], []
        for i, row in enumerate(indices):
            for idx in row[row >= 0].tolist():
                gather.append(first[idx])
                segment.append(i)
        return np.asarray(gather, dtype=np.int64), np.asarray(segment, dtype=np.int64)


class MultiAgentSharedPolicy(nn.Module):
    """One shared policy for every agent id (shared_policy.py:14-31): ``forward`` returns the actions
    in the batch's original order, exactly what the reference's split / stitch produces."""

    def __init__(self, policy: DQNPolicy, env=None, agents=None, **_):
        """``env``: as in the reference (shared_policy.py:23-30: a PettingZooEnv whose ``agents`` / ``agent_idx`` are taken) - here
        a :class:`melissa_amd.env.HipGraphVectorEnv` (agents are named by their id, core.py:46) or anything with ``agents``;
        a plain list of agent names is accepted too."""
        super().__init__()
        self.policy = policy
        if agents is not None:
            pass
        elif hasattr(env, "agents"):
            agents = env.agents
        elif hasattr(env, "n") and hasattr(env, "env_num"):
            agents = [str(i) for i in range(env.n)]
        else:
            agents = env
        self.agents = list(agents)
        self.agent_idx = getattr(env, "agent_idx", None) or {a: i for i, a in enumerate(self.agents)}

    def forward(self, batch, state=None, **kwargs) -> Batch:
        out = self.policy(batch, state=state, **kwargs)
        return Batch(act=out.act, state=out.state, out=out)

    def exploration_noise(self, act, batch):
        return self.policy.exploration_noise(act, batch)

    def learn(self, batch, **kwargs):
        return self.policy.learn(batch, **kwargs)
