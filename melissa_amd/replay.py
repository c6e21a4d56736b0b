"""Device-resident replay for the round-batched loop + n-step return sampling.

Reference: the collectors route each agent's transition into one of ``num_envs x num_agents`` Tianshou
sub-buffers, waiting for the agent's next observation before the record is complete
(multi_agent_collector.py:229-271); ``DQNPolicy.process_fn`` then builds n-step returns
(``estimation_step`` = 4, common.py:28).  In the round-batched loop a whole env round is ONE record
(``mel_round_replay`` in include/melissa_hip.h) written by ``mel_env_round``: all of the round's agents share
``obs`` / ``obs_next`` (they observe the same obs_matrix, graph.py:186-188), so a transition is
``(record, agent)`` and an agent's trajectory is the run of consecutive records of its env in which it acts.
"""
from __future__ import annotations

import torch

from . import _lib
from .optim import adam_step


class RoundReplay:
    def __init__(self, n_envs: int, n_nodes: int, capacity: int, device):
        self.B, self.n, self.K = n_envs, n_nodes, capacity
        dev = torch.device(device)
        self.obs = torch.zeros(n_envs, capacity, 8 * n_nodes, dtype=torch.float32, device=dev)
        self.obs_next = torch.zeros_like(self.obs)
        # node sets (who acted / who was terminated by the round): one int64 word up to 64 nodes, [W] words beyond
        self.W = _lib.set_words(n_nodes)
        ws = () if self.W == 1 else (self.W,)
        self.acted = torch.zeros(n_envs, capacity, *ws, dtype=torch.int64, device=dev)
        self.done = torch.zeros(n_envs, capacity, *ws, dtype=torch.int64, device=dev)
        self.act = torch.zeros(n_envs, capacity, n_nodes, dtype=torch.int8, device=dev)
        self.rew = torch.zeros(n_envs, capacity, n_nodes, dtype=torch.float32, device=dev)
        self.episode = torch.full((n_envs, capacity), -1, dtype=torch.int32, device=dev)
        self.cursor = torch.zeros(n_envs, dtype=torch.int32, device=dev)
        s = _lib.MelRoundReplay()
        s.capacity = capacity
        s.obs, s.obs_next = self.obs.data_ptr(), self.obs_next.data_ptr()
        s.acted, s.done, s.act = self.acted.data_ptr(), self.done.data_ptr(), self.act.data_ptr()
        s.rew, s.episode, s.cursor = self.rew.data_ptr(), self.episode.data_ptr(), self.cursor.data_ptr()
        self.struct = s
        node = torch.arange(n_nodes, device=dev)
        self._word, self._shift = node // 64, node % 64

    def __len__(self) -> int:
        """Transitions currently held (synchronises)."""
        valid = self._valid_slots()
        return int(self._popcount(self.acted[valid]).sum())

    def _members(self, m: torch.Tensor) -> torch.Tensor:
        """node sets int64 [..., (W)] -> bool [..., N]."""
        if self.W == 1:
            m = m[..., None]
        return ((m[..., self._word] >> self._shift) & 1) != 0

    def _has(self, m: torch.Tensor, agent: torch.Tensor) -> torch.Tensor:
        """node sets int64 [bs, (W)], agent [bs] -> bool [bs]: is agent[i] a member of m[i]?"""
        if self.W == 1:
            m = m[..., None]
        return ((m.gather(-1, (agent // 64)[:, None]).squeeze(-1) >> (agent % 64)) & 1) != 0

    def _popcount(self, m: torch.Tensor) -> torch.Tensor:
        return self._members(m).sum(-1)

    def _valid_slots(self) -> torch.Tensor:
        filled = torch.clamp(self.cursor.long(), max=self.K)                       # [B]
        return torch.arange(self.K, device=self.cursor.device)[None, :] < filled[:, None]

    def sample(self, batch_size: int, n_step: int, gamma: float, generator: torch.Generator | None = None):
        """Uniform over (record, acting agent) pairs.  Returns dict of device tensors:
        obs [bs, 8N+1], act [bs], ret [bs] (discounted n-step reward sum), boot_obs [bs, 8N+1] (observation to
        bootstrap from), boot_w [bs] (gamma^steps, 0 when the agent terminated inside the window), env / slot / agent [bs].

        On the GPU this is ONE launch (``mel_replay_sample``; the index arithmetic below is ~150 small launches, 0.6 ms of a
        1.7 ms update replayed from HIP graphs): draws are a counter-based function of the generator's seed and a device-side
        counter that every call advances, so a captured call keeps drawing new batches when replayed.  Host tensors (CPU tests
        of the host logic) take the torch formulation below - same distribution, torch's own random stream."""
        dev = self.obs.device
        if dev.type == "cuda":
            return self._sample_device(batch_size, n_step, gamma, generator)
        valid = self._valid_slots()
        # never start a chain in the slot about to be overwritten next
        cnt = self._popcount(self.acted) * valid
        flat = cnt.flatten().float()
        idx = torch.multinomial(flat, batch_size, replacement=True, generator=generator)
        e, k = idx // self.K, idx % self.K
        mask = self.acted[e, k]
        # pick the j-th acting agent uniformly
        bits = self._members(mask)                                                 # [bs, N]
        u = torch.rand(batch_size, device=dev, generator=generator)
        target = (u * bits.sum(1)).floor().long().clamp(max=self.n - 1)
        order = torch.cumsum(bits.long(), dim=1) - 1
        agent = ((order == target[:, None]) & bits).float().argmax(dim=1)
        ret = torch.zeros(batch_size, device=dev)
        boot_w = torch.ones(batch_size, device=dev)
        alive = torch.ones(batch_size, dtype=torch.bool, device=dev)
        boot_slot = k.clone()
        ep0 = self.episode[e, k]
        newest = (self.cursor[e].long() - 1) % self.K
        kk = k.clone()
        for j in range(n_step):
            ok = alive & (self.episode[e, kk] == ep0) & self._has(self.acted[e, kk], agent) & valid[e, kk]
            ret = ret + torch.where(ok, (gamma ** j) * self.rew[e, kk, agent], torch.zeros((), device=dev))
            boot_slot = torch.where(ok, kk, boot_slot)
            boot_w = torch.where(ok, torch.full((), gamma ** (j + 1), device=dev), boot_w)
            finished = ok & self._has(self.done[e, kk], agent)
            boot_w = torch.where(finished, torch.zeros((), device=dev), boot_w)
            alive = ok & ~finished & (kk != newest)          # cannot walk past the newest record
            kk = (kk + 1) % self.K
        idx_col = agent.float()[:, None]
        obs = torch.cat([self.obs[e, k], idx_col], dim=1)
        boot_obs = torch.cat([self.obs_next[e, boot_slot], idx_col], dim=1)
        return dict(obs=obs, act=self.act[e, k, agent].long(), ret=ret, boot_obs=boot_obs, boot_w=boot_w,
                    env=e, slot=k, agent=agent)


    def _sample_device(self, batch_size: int, n_step: int, gamma: float, generator):
        import ctypes as C
        dev, lib = self.obs.device, _lib.load()
        if not getattr(self, "_nonempty", False):               # (one host read, until the first record is seen)
            if int(self.cursor.max()) < 1:
                raise ValueError("cannot sample from an empty replay: run the collect loop first")
            self._nonempty = True
        if not hasattr(self, "_draws"):
            self._draws = torch.zeros(1, dtype=torch.int64, device=dev)            # device-side draw counter
            self._prefix = torch.empty(self.B * self.K + 1, dtype=torch.int32, device=dev)
        width = 8 * self.n + 1
        out = dict(obs=torch.empty(batch_size, width, device=dev), boot_obs=torch.empty(batch_size, width, device=dev),
                   act=torch.empty(batch_size, dtype=torch.int64, device=dev), ret=torch.empty(batch_size, device=dev),
                   boot_w=torch.empty(batch_size, device=dev), env=torch.empty(batch_size, dtype=torch.int64, device=dev),
                   slot=torch.empty(batch_size, dtype=torch.int64, device=dev),
                   agent=torch.empty(batch_size, dtype=torch.int64, device=dev))
        b = _lib.MelReplayBatch()
        for name, t in out.items():
            setattr(b, name, t.data_ptr())
        disc = (C.c_float * (n_step + 1))(*[gamma ** j for j in range(n_step + 1)])
        seed = (generator.initial_seed() if generator is not None else torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        _lib.check(lib.mel_replay_sample(C.byref(self.struct), self.B, self.n, batch_size, n_step, disc, seed,
                                         self._draws.data_ptr(), self._prefix.data_ptr(), C.byref(b),
                                         _lib.current_stream_ptr(dev)), "mel_replay_sample")
        return out

    def sample_collective(self, batch_size: int, n_step: int, gamma: float, generator: torch.Generator | None = None):
        """``sample`` plus every sampled transition's SIBLINGS - the agents that acted in the same env round
        (what collective_experience_collector.py:70-80 records as ``info.indices``): here they are simply the
        ``acted`` set of the same record.  Adds ``active_obs`` [M, 8N+1], ``active_act`` [M], ``segment`` [M]
        (rows ordered by experience, then agent id) for :class:`melissa_amd.policy.DGNPolicy`."""
        b = self.sample(batch_size, n_step, gamma, generator)
        e, k = b["env"], b["slot"]
        bits = self._members(self.acted[e, k])                                     # [bs, N]
        seg, agent = torch.nonzero(bits, as_tuple=True)                            # row-major: by experience, then id
        obs = torch.cat([self.obs[e[seg], k[seg]], agent.float()[:, None]], dim=1)
        b.update(active_obs=obs, active_act=self.act[e[seg], k[seg], agent].long(), segment=seg, sibling_mask=self.acted[e, k])
        return b

    def export_transitions(self):
        """The buffer in the reference collectors' layout (multi_agent_collector.py:229-271,
        collective_experience_collector.py:70-80,251-309): one row per (record, acting agent) transition with
        ``buffer_id = env * N + agent`` (the Tianshou VectorReplayBuffer sub-buffer it is routed to) and
        ``indices`` [T, N]: row of each SIBLING transition (agent j acted in the same env round) or -1.
        Rows are ordered by env, record age (oldest first), agent id.  Host NumPy; synchronises."""
        import numpy as np
        valid = self._valid_slots()
        B, K, n = self.B, self.K, self.n
        # oldest-first slot order per env: the ring's next write position is cursor % K
        start = torch.where(self.cursor.long() > K, self.cursor.long() % K, torch.zeros_like(self.cursor.long()))
        order = (start[:, None] + torch.arange(K, device=start.device)[None, :]) % K          # [B, K]
        env = torch.arange(B, device=start.device)[:, None].expand(B, K)
        ok = valid.gather(1, order)
        e, k = env[ok], order[ok]
        bits = self._members(self.acted[e, k])
        rec, agent = torch.nonzero(bits, as_tuple=True)
        T = rec.numel()
        row_of = torch.full((e.numel(), n), -1, dtype=torch.int64, device=start.device)
        row_of[rec, agent] = torch.arange(T, device=start.device)
        ee, kk = e[rec], k[rec]
        idx_col = agent.float()[:, None]
        out = dict(obs=torch.cat([self.obs[ee, kk], idx_col], 1), obs_next_matrix=self.obs_next[ee, kk],
                   act=self.act[ee, kk, agent].long(), rew=self.rew[ee, kk], rew_agent=self.rew[ee, kk, agent],
                   done=self._has(self.done[ee, kk], agent), env_id=ee, agent_id=agent,
                   buffer_id=ee * n + agent, record_slot=kk, episode=self.episode[ee, kk].long(), indices=row_of[rec])
        return {name: t.cpu().numpy() for name, t in out.items()}


class DQNLearner:
    """n-step DQN update over a :class:`RoundReplay` (the learn half of the reference's training loop,
    l_dgn.py:246-261 -> [3P] DQNPolicy.process_fn / learn): target = ret + boot_w * max_a Q_target(boot_obs)
    with the target network evaluated by the HIP forward, then one autograd step (``grad_hook`` = the flat
    RCCL gradient all-reduce of melissa_amd.parallel)."""

    def __init__(self, policy, replay: RoundReplay, batch_size: int = 32, n_step: int = 4, gamma: float = 0.99,
                 grad_hook=None, seed: int = 0):
        self.policy, self.replay = policy, replay
        self.batch_size, self.n_step, self.gamma, self.grad_hook = batch_size, n_step, gamma, grad_hook
        self.gen = torch.Generator(device=replay.obs.device)
        self.gen.manual_seed(seed)
        self.captured = None

    def sample_batch(self) -> dict:
        """Sample + n-step targets: dict(obs, act, returns) of device tensors, no host synchronisation."""
        b = self.replay.sample(self.batch_size, self.n_step, self.gamma, self.gen)
        with torch.no_grad():
            target_net = self.policy.model_old if getattr(self.policy, "_target", False) else self.policy.model
            q_next = target_net.hip_forward(b["boot_obs"]) if b["boot_obs"].is_cuda else target_net.torch_forward(b["boot_obs"])
            if self.policy._is_double:                      # double DQN: argmax from the online net
                online = self.policy.model.hip_forward(b["boot_obs"]) if b["boot_obs"].is_cuda else self.policy.model.torch_forward(b["boot_obs"])
                best = q_next.gather(1, online.argmax(dim=1, keepdim=True)).squeeze(1)
            else:
                best = q_next.max(dim=1).values
            returns = b["ret"] + b["boot_w"] * best
        # (boot_obs / ret / boot_w: what `returns` was built from - tests recompute it through the target network)
        return dict(obs=b["obs"], act=b["act"], returns=returns, boot_obs=b["boot_obs"], ret=b["ret"], boot_w=b["boot_w"])

    def step(self) -> dict:
        if self.captured is not None:
            return self.captured.step()
        self.last_batch = self.sample_batch()                                      # what the update regressed on (tests)
        return self.policy.learn(dict(self.last_batch), grad_hook=self.grad_hook)

    def capture(self) -> "CapturedUpdate":
        """From now on ``step`` replays the update from HIP graphs (see :class:`CapturedUpdate`)."""
        self.captured = CapturedUpdate(self)
        return self.captured


class CapturedUpdate:
    """One DQN update replayed from HIP graphs instead of issued launch by launch from Python.

    An update is a few hundred small launches (replay sampling, the target forward, forward + backward, Adam) behind
    ~3 ms of Python; the device needs a third of that.  All shapes of a DQN update are static (batch x (8 N + 1)
    observations), so after two eager warm-up updates the whole chain is captured once:

        graph A   sample -> n-step targets (target network, HIP forward) -> zero_grad -> forward -> loss -> backward
                  [-> gradients packed into the reducer's flat buffer]
        eager     [one all-reduce of the flat buffer + division: RCCL stays outside the capture]
        graph B   [flat buffer unpacked into the gradients ->] optimizer step

    (one graph when there is no collective).  What stays in Python per update: the target network's periodic weight sync
    and invalidating the networks' per-weight-version caches (prepared bf16 planes / feature tables are keyed on torch's
    parameter version counters, which a replay does not bump).  The optimizer must be a torch optimizer with a
    ``capturable`` flag (Adam / AdamW / ...): it is switched on here and the step counters moved to the device.
    ``step`` returns the loss as a device tensor (``float()`` it to synchronise); ``last_batch`` aliases the graph's static
    tensors (valid until the next replay)."""

    def __init__(self, learner: "DQNLearner", warmup: int = 2):
        self.learner = L = learner
        policy = L.policy
        dev = L.replay.obs.device
        if dev.type != "cuda":
            raise ValueError("CapturedUpdate needs a ROCm / CUDA device")
        opt = policy.optim
        for group in opt.param_groups:
            if "capturable" not in group:
                raise ValueError(f"{type(opt).__name__} has no capturable mode; cannot capture its step")
            group["capturable"] = True
        for st in opt.state.values():
            if "step" in st and torch.is_tensor(st["step"]) and not st["step"].is_cuda:
                st["step"] = st["step"].to(dev)
        hook = L.grad_hook
        self.collective = hook is not None and getattr(hook, "active", lambda: True)()
        if self.collective and not all(hasattr(hook, m) for m in ("pack", "reduce", "unpack")):
            raise ValueError("a captured update needs a grad_hook with pack / reduce / unpack phases (FlatGradAllReducer)")
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                       # warm-up off the capture stream (lazy inits, optimizer state)
            for _ in range(warmup):
                self._sync_target()
                batch = L.sample_batch()
                policy.loss_backward(batch)
                if self.collective:
                    hook.pack(), hook.reduce(), hook.unpack()
                adam_step(opt)
                policy._iter += 1
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph_a = torch.cuda.CUDAGraph()
        self.graph_a.register_generator_state(L.gen)
        with torch.cuda.graph(self.graph_a):
            self.batch = L.sample_batch()
            self.loss = policy.loss_backward(self.batch)
            if self.collective:
                hook.pack()
            else:
                adam_step(opt)
        self.graph_b = None
        if self.collective:
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool()):
                hook.unpack()
                adam_step(opt)
        L.last_batch = self.batch

    def _sync_target(self):
        policy = self.learner.policy
        if policy._target and policy._iter % policy._freq == 0:
            policy.sync_weight()
            # the target network only ever runs INSIDE graph A, whose launches read its converted projection weights from the
            # buffer captured with them: reconvert the new weights into that buffer now (no Python runs during a replay)
            policy.model_old.ensure_prepared(self.learner.replay.obs.device)

    def step(self) -> dict:
        policy = self.learner.policy
        self._sync_target()
        if policy._is_double:                  # graph A also evaluates the ONLINE network through the HIP forward
            policy.model.ensure_prepared(self.learner.replay.obs.device)
        self.graph_a.replay()
        if self.collective:
            self.learner.grad_hook.reduce()
            self.graph_b.replay()
        policy._iter += 1
        # the replay changed the parameters behind torch's version counters: mark the per-weight-version caches stale (the
        # prepared planes keep their buffer - launches captured elsewhere hold its address - and are converted again by the
        # next forward)
        policy.model.mark_weights_changed()
        return {"loss": self.loss.clone()}                     # (self.loss is the graph's static tensor: the next replay overwrites it)


class DGNLearner(DQNLearner):
    """DGN-R update (policies/dgn.py): the sampled experiences' n-step returns regress on the summed Q of their
    siblings (policy = :class:`melissa_amd.policy.DGNPolicy`).  All siblings of an experience acted in the same env round and
    so share its observation matrix: the network body runs once per sampled experience and its head once per (experience,
    node) - ``GraphQNetwork.torch_forward_all_agents`` - instead of one whole forward per sibling row (the reference's loop,
    dgn.py:31-55: ~21 siblings per experience at N = 50).  Every tensor of the update then has a static shape, so it replays
    from HIP graphs like the DQN update (``capture()``)."""

    def sample_batch(self) -> dict:
        """Sample + n-step targets in the DENSE form (static shapes, no host synchronisation): obs_matrix [B, 8N], act_all
        [B, N], sibling [B, N] (the agents that acted in the sampled round), returns [B]."""
        b = self.replay.sample(self.batch_size, self.n_step, self.gamma, self.gen)
        e, k = b["env"], b["slot"]
        with torch.no_grad():
            target_net = self.policy.model_old if getattr(self.policy, "_target", False) else self.policy.model
            fwd = (lambda net, o: net.hip_forward(o)) if b["boot_obs"].is_cuda else (lambda net, o: net.torch_forward(o))
            q_next = fwd(target_net, b["boot_obs"])
            if self.policy._is_double:
                best = q_next.gather(1, fwd(self.policy.model, b["boot_obs"]).argmax(dim=1, keepdim=True)).squeeze(1)
            else:
                best = q_next.max(dim=1).values
            returns = b["ret"] + b["boot_w"] * best
        return dict(obs_matrix=self.replay.obs[e, k], act_all=self.replay.act[e, k].long(),
                    sibling=self.replay._members(self.replay.acted[e, k]), returns=returns, boot_obs=b["boot_obs"], ret=b["ret"],
                    boot_w=b["boot_w"], env=e, slot=k)

    def row_form(self, batch: dict) -> dict:
        """The same batch in the reference's row form (what collective_experience_collector.py:70-80 records as ``info.indices``:
        one row per sibling, ordered by experience then agent id): active_obs [M, 8N+1], active_act [M], segment [M].  M depends
        on the data (host synchronisation): for tests and for callers that keep the reference's loss loop."""
        seg, agent = torch.nonzero(batch["sibling"], as_tuple=True)
        obs = torch.cat([batch["obs_matrix"][seg], agent.float()[:, None]], dim=1)
        return dict(active_obs=obs, active_act=batch["act_all"][seg, agent], segment=seg, returns=batch["returns"])

    def step(self) -> dict:
        if self.captured is not None:
            return self.captured.step()
        batch = self.sample_batch()
        self.last_batch = dict(batch, **self.row_form(batch))
        return self.policy.learn(dict(batch), grad_hook=self.grad_hook)
