"""The collector hot loop, device resident.

Reference: ``MultiAgentCollector.collect``'s ``while True`` loop (multi_agent_collector.py:150-308):
one iteration = policy forward over the ready envs -> eps-greedy -> ``env.step`` -> next obs, with a
host<->device round trip and O(envs) Python bookkeeping per iteration.  ``DecisionLoop`` runs the same
iteration as a fixed sequence of HIP launches on one stream with no host synchronisation:

    obs [B, 8N+1] (HBM) -> mel_{l,hl}dgn_forward -> logits -> mel_select_action -> act
        -> mel_env_step (+ last() + on-device episode reset) -> obs (same buffer)

Finished episodes are re-seeded on the device from a pre-sampled episode pool (the packed replacement
of the reference's per-reset pickle.load + RNG draws).  Replay-buffer routing
(multi_agent_collector.py:229-271) is not part of this loop.
"""
from __future__ import annotations

import dataclasses

import numpy as np
import torch

from . import _lib
from .env.episodes import EpisodeSampler, pack_episodes
from .env.stream import make_supply
from .env.vector_env import HipGraphVectorEnv, ObsBuffers


def sample_episode_table(venv: HipGraphVectorEnv, episodes_per_env: int, seed: int = 0):
    """Pre-draw ``episodes_per_env`` episodes for every env with the reference's RNG protocol (env k's
    generator seeded ``seed + k``) -> (packed pool dict, episode_table int32 [B, K])."""
    episodes, table = [], np.zeros((venv.env_num, episodes_per_env), dtype=np.int32)
    for b in range(venv.env_num):
        sampler = venv.make_sampler(seed + b)          # same evaluation schedule / scripted ratio / density as the env
        for k in range(episodes_per_env):
            table[b, k] = len(episodes)
            episodes.append(sampler.sample())
    packed = pack_episodes(episodes, venv.graphs, venv.n, venv.max_moves, venv.dynamic_graph)
    return packed, table


def _counters_of(scalars: torch.Tensor, iterations: int) -> dict:
    sc = scalars.cpu().numpy()
    return dict(decisions=int(sc[:, _lib.S_DECISIONS].sum()), episodes=int(sc[:, _lib.S_EPISODES_DONE].sum()),
                errors=int(np.bitwise_or.reduce(sc[:, _lib.S_ERROR])), iterations=iterations)


class DecisionLoop:
    """AEC-order loop: one agent decision per env per iteration (the reference collector's granularity)."""

    def __init__(self, venv: HipGraphVectorEnv, policy, episodes_per_env: int = 8, seed: int = 0,
                 eps: float = 0.0, episodes=None, stream: bool | None = None, ring: int = 64, discard: int = 0):
        """Episodes: a device stream (fresh episodes for ever, ``melissa_amd.env.stream``) unless ``episodes`` =
        (packed, table) is given or ``stream`` is False (a host-drawn table of ``episodes_per_env`` per env)."""
        self.venv, self.policy, self.eps = venv, policy, eps
        dev = venv.device
        self.supply = make_supply(venv, seed, episodes, episodes_per_env, stream, ring, discard, reset_snapshots=False)
        self.pool, self.table = self.supply.pool, self.supply.table
        self.n_actions = policy.model.output_dim
        self.obs = torch.empty(venv.env_num, 8 * venv.n + 1, dtype=torch.float32, device=dev)
        self.out = ObsBuffers(venv.env_num, venv.n, dev, obs=self.obs)
        self.logits = torch.empty(venv.env_num, self.n_actions, dtype=torch.float32, device=dev)
        self.act = torch.empty(venv.env_num, dtype=torch.int32, device=dev)
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed)
        self.iterations = 0
        venv.reset_device(self.pool, self.supply.first_episode_ids(), self.out)

    def step(self):
        """One collector iteration for every env (one agent decision per env)."""
        self.supply.before_step(self.iterations)
        net = self.policy.model
        # observations written by the env kernels: integer node features (unless scripted agents relay beyond four times)
        net.hip_forward(self.obs, out=self.logits,
                        integer_features=self.venv._sampler_kw["scripted_agents_ratio"] == 0.0)
        if self.eps > 0.0:
            rand_u = torch.rand(self.venv.env_num, device=self.obs.device, generator=self.gen)
            rand_q = torch.rand(self.venv.env_num, self.n_actions, device=self.obs.device, generator=self.gen)
            self.policy.select_action(self.logits, self.out.action_mask, self.eps, rand_u, rand_q, out=self.act)
        else:
            self.policy.select_action(self.logits, self.out.action_mask, out=self.act)
        self.venv.step_device(self.pool, self.act, self.out, self.table)
        self.iterations += 1

    def run(self, iterations: int):
        for _ in range(iterations):
            self.step()

    def snapshot_counters(self):
        """Device-side copy of the per-env counters as they stand when the launches enqueued so far have run (no host
        synchronisation: the copy is one more launch in the queue); ``counters(snapshot)`` reads it later."""
        return (self.venv.scalars().clone(), self.iterations)

    def counters(self, snapshot=None) -> dict:
        """Host read of the per-env counters (synchronises)."""
        if snapshot is not None:
            return _counters_of(*snapshot)
        return _counters_of(self.venv.scalars(), self.iterations)


class RoundLoop:
    """Round-batched loop: one iteration = one whole env ROUND for every env.

    Within a round every active agent observes the same ``obs_matrix`` (graph.py:186-188: the rows differ
    only in the controlling-agent column) and no agent's action is visible to another before the world
    step (graph.py:324-347), so all of a round's decisions are taken by ONE forward
    (``mel_ldgn_forward_agents``: shared encoder / conv1 over the union of the receptive fields) and applied
    by ONE env launch (``mel_env_round``) that replays the reference's AEC order - pending dead agents first,
    then each active agent with its own action, then the world step.  Per-env trajectories are identical to
    the AEC-order loop under the same actions; an iteration yields ~N/6 decisions per env instead of one.
    """

    def __init__(self, venv: HipGraphVectorEnv, policy, episodes_per_env: int = 8, seed: int = 0,
                 eps: float = 0.0, episodes=None, rows_cap: int | None = None, use_graph: bool = False,
                 stream: "torch.cuda.Stream | None" = None, replay=None, episode_stream: bool | None = None,
                 ring: int = 16, discard: int = 0, graph_rounds: int = 4):
        """Episodes: by default a device STREAM (``melissa_amd.env.stream.EpisodeStream``: every reset draws a new episode
        like World.reset does, core.py:372-394; ``ring`` slots per env, ``discard`` construction-time samplings dropped);
        ``episodes`` = (packed, table) or ``episode_stream=False`` give a fixed table of ``episodes_per_env`` episodes."""
        self.venv, self.policy, self.eps, self.seed = venv, policy, eps, seed
        self.use_graph, self.graph = use_graph, None
        # run(): whole groups of `graph_rounds` rounds are replayed from ONE HIP graph - a graph launch costs the device ~8 us
        # of idle time between the last kernel of one replay and the first of the next (rocprofv3 kernel trace of the
        # one-round graph: tools/gap_trace.py), a quarter of that per round with four rounds per graph
        self.graph_rounds, self.group_graph = max(1, int(graph_rounds)), None
        # the observations are the env kernels' own, so their node features are integers in known ranges and the forward may
        # evaluate encoder / conv1 projections once per feature TUPLE (MEL_FWD_INTEGER_FEATURES); False forces row lists
        # (scripted agents relay without a step budget, so their message counts leave the table's range: row lists then)
        self.integer_features = venv._sampler_kw["scripted_agents_ratio"] == 0.0
        self.stream = stream                       # None: torch's current stream
        self.replay = replay                       # optional melissa_amd.replay.RoundReplay
        dev = venv.device
        # episode ends load their next state from reset snapshots (static tables: precomputed here; streams: by the refill)
        if stream is not None:
            with torch.cuda.stream(stream):
                self.supply = make_supply(venv, seed, episodes, episodes_per_env, episode_stream, ring, discard)
        else:
            self.supply = make_supply(venv, seed, episodes, episodes_per_env, episode_stream, ring, discard)
        self.pool, self.table = self.supply.pool, self.supply.table
        self.n_actions = policy.model.output_dim
        # HL-DGN's logits do not depend on the agent (hl_dgn.py:108): one row per env, dense action layout
        self.per_env_logits = policy.model._MODEL == _lib.MODEL_HLDGN
        self.rows_cap = venv.env_num if self.per_env_logits else int(rows_cap or venv.env_num * venv.n)
        self.live = torch.zeros(venv.env_num, *(() if venv.n <= 64 else (_lib.set_words(venv.n),)), dtype=torch.int64,
                                device=dev)         # the round's active sets (node sets: [B], or [B, W] beyond 64 nodes)
        self.offsets = torch.zeros(venv.env_num + 1, dtype=torch.int32, device=dev)
        self.logits = torch.zeros(self.rows_cap, self.n_actions, dtype=torch.float32, device=dev)
        self.act = torch.zeros(venv.env_num * venv.n if self.per_env_logits else self.rows_cap, dtype=torch.int32,
                               device=dev)
        self.iterations = 0
        self._rounds_base = 0
        self.rounds = torch.zeros(1, dtype=torch.int32, device=dev)     # device-side round counter (RNG step, refill pacing)
        self._select = _lib.MelSelect()
        self._select.act, self._select.eps = self.act.data_ptr(), float(eps)
        self._select.seed, self._select.step_dev = seed & 0xFFFFFFFF, self.rounds.data_ptr()
        if self.per_env_logits:                    # every agent of live[b] draws from env b's logits row
            self._select.live, self._select.n_nodes = self.live.data_ptr(), venv.n
        self._obs_matrix = venv.obs_matrix()
        # own forward scratch: several loops may share one network on different streams
        self.workspace = torch.empty(policy.model.agents_workspace_bytes(venv.env_num, self.rows_cap),
                                     dtype=torch.uint8, device=dev)
        # the env round also writes the plan masks of the next forward into this workspace (mel_env_batch.plan_*): one
        # launch less per round
        self._plan = policy.model.plan_pointers(venv.env_num, 0 if self.per_env_logits else self.rows_cap, self.workspace)
        torch.cuda.synchronize(dev)
        venv.reset_device(self.pool, self.supply.first_episode_ids(), None)
        self._bind_plan()
        venv.round_device(self.pool, None, None, self.live, None, first=True)
        torch.cuda.synchronize(dev)

    def _bind_plan(self):
        """Point the venv's plan sink at THIS loop's workspace (cheap; done before every launch because the struct belongs to
        the venv and another loop may have used it in between)."""
        e = self.venv.env
        e.plan_adj, e.plan_live, e.plan_u1, e.plan_u2, e.plan_cnt = self._plan

    def _launch(self):
        """The fixed launch sequence of one round (no host reads, no allocation: capturable)."""
        lib = _lib.load()
        net = self.policy.model
        dev = self.venv.device
        if self.per_env_logits:
            # forward + per-(env, agent) argmax / eps-greedy in the launch that writes the logits (same stream of draws as
            # mel_select_action_envs)
            net.hip_forward_envs(self._obs_matrix, out=self.logits, workspace=self.workspace, select=self._select,
                                 plan_ready=True, integer_features=self.integer_features)
            self._bind_plan()
            self.venv.round_device(self.pool, self.act, None, self.live, self.table, round_counter=self.rounds,
                                   replay=self.replay)
            return
        # forward + fused argmax / eps-greedy (the dueling tail writes the action next to the logits)
        # (integer_features: the obs rows are the env kernel's own -> encoder / conv1 projections per feature tuple)
        net.hip_forward_agents(self._obs_matrix, self.live, self.rows_cap, out=self.logits, row_offsets=self.offsets,
                               select=self._select, workspace=self.workspace, plan_ready=True,
                               integer_features=self.integer_features)
        self._bind_plan()
        self.venv.round_device(self.pool, self.act, self.offsets, self.live, self.table, round_counter=self.rounds,
                               replay=self.replay)

    def step(self):
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                self._step()
        else:
            self._step()

    def _step(self):
        # episode stream: refill on a side stream every few rounds, paced by the device-side round counter; the pacing gate
        # goes into its queue AFTER the launches it waits for (EpisodeStream.after_step)
        it = self.iterations - self._rounds_base
        self.supply.before_step(it, self.rounds)
        if self.use_graph:
            if self.graph is None:
                self._launch()                        # warm-up outside capture (lazy init)
                self.iterations += 1
                self.supply.after_step(it, self.rounds)
                torch.cuda.synchronize(self.venv.device)
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):    # launches land on the capture stream (torch's current one)
                    self._launch()
                return                                # capture does not execute: the next call replays
            # a replay runs no Python: whatever changed the weights since the last step (an optimizer step, a checkpoint load)
            # is picked up here - the converted projection weights the captured launches read are brought up to date first
            self.policy.model.ensure_prepared(self.venv.device)
            self.graph.replay()
        else:
            self._launch()
        self.iterations += 1
        self.supply.after_step(it, self.rounds)

    def _run_groups(self, groups: int):
        """``groups`` replays of the graph that holds ``graph_rounds`` rounds back to back (same launches, same buffers, same
        device-side round counter as the one-round graph: bit-identical trajectories)."""
        G, dev = self.graph_rounds, self.venv.device
        if self.group_graph is None:
            torch.cuda.synchronize(dev)
            self.group_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.group_graph):
                for _ in range(G):
                    self._launch()
        for _ in range(groups):
            it = self.iterations - self._rounds_base
            for k in range(G):
                self.supply.before_step(it + k, self.rounds)
            self.policy.model.ensure_prepared(dev)
            self.group_graph.replay()
            self.iterations += G
            for k in range(G):                    # the refill gates go into the side stream's queue behind the rounds they wait for
                self.supply.after_step(it + k, self.rounds)

    def run(self, iterations: int):
        paced = getattr(self.supply, "_paced", None)            # (event-ordered refills touch the main stream between rounds)
        while self.use_graph and self.graph is None and iterations > 0:      # eager warm-up round, then the capture
            self.step()
            iterations -= 1
        if (self.use_graph and self.graph is not None and self.graph_rounds > 1 and iterations >= self.graph_rounds
                and (paced is None or paced(self.rounds))):
            groups = iterations // self.graph_rounds
            if self.stream is not None:
                with torch.cuda.stream(self.stream):
                    self._run_groups(groups)
            else:
                self._run_groups(groups)
            iterations -= groups * self.graph_rounds
        for _ in range(iterations):
            self.step()

    def snapshot_counters(self):
        """Device-side copy of the per-env counters as they stand when the launches enqueued so far have run (no host
        synchronisation: the copy is one more launch on the loop's stream); ``counters(snapshot)`` reads it later."""
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                return (self.venv.scalars().clone(), self.iterations)
        return (self.venv.scalars().clone(), self.iterations)

    def counters(self, snapshot=None) -> dict:
        if snapshot is not None:
            return _counters_of(*snapshot)
        return _counters_of(self.venv.scalars(), self.iterations)

    def feature_table(self) -> dict:
        """What the last forward did with the node-feature table: rows used (0 = row-list path) and how many envs had a
        node feature outside the integer ranges the table assumes (must be 0 for env-produced observations)."""
        t = self.policy.model.hip_tap(3, self.venv.env_num, 0 if self.per_env_logits else self.rows_cap,
                                      workspace=self.workspace).cpu().numpy()
        return dict(table_rows=int(t[0]), bad_envs=int(t[1:].sum()))


class MultiStreamRoundLoop:
    """The GPU's envs split into independent sub-batches, each a :class:`RoundLoop` on its own HIP stream.
    One serial launch chain leaves the chip under-filled (latency-bound plan / env / attention launches, GEMM
    tails); sub-batches on separate streams fill those gaps with each other's work.  Envs are independent, so
    this changes nothing about any env's trajectory."""

    def __init__(self, make_venv, policy, n_envs: int, streams: int = 2, episodes_per_env: int = 8, seed: int = 0,
                 eps: float = 0.0, use_graph: bool = True, ring: int = 16):
        per = (n_envs + streams - 1) // streams
        self.loops = []
        for k in range(streams):
            lo, hi = k * per, min((k + 1) * per, n_envs)
            if hi <= lo:
                break
            venv = make_venv(hi - lo, seed + lo)
            self.loops.append(RoundLoop(venv, policy, episodes_per_env=episodes_per_env, seed=seed + lo, eps=eps,
                                        use_graph=use_graph, stream=torch.cuda.Stream(device=venv.device), ring=ring))
        self.iterations = 0

    @property
    def use_graph(self):
        return self.loops[0].use_graph

    @use_graph.setter
    def use_graph(self, v):
        for l in self.loops:
            l.use_graph = v

    def step(self):
        for l in self.loops:
            l.step()
        self.iterations += 1

    def run(self, iterations: int):
        for _ in range(iterations):
            self.step()

    def snapshot_counters(self):
        return [l.snapshot_counters() for l in self.loops]

    def counters(self, snapshot=None) -> dict:
        out = dict(decisions=0, episodes=0, errors=0, iterations=self.iterations)
        for k, l in enumerate(self.loops):
            c = l.counters(snapshot[k] if snapshot is not None else None)
            out["decisions"] += c["decisions"]
            out["episodes"] += c["episodes"]
            out["errors"] |= c["errors"]
        return out


@dataclasses.dataclass
class SequenceSummaryStats:
    """[3P] tianshou 1.0.0 ``SequenceSummaryStats``: mean / std / max / min of a sequence (what ``returns_stat`` / ``lens_stat`` /
    ``info.stats[key]`` of the reference's collect result hold, collector.py:14-36)."""
    mean: float
    std: float
    max: float
    min: float

    @classmethod
    def from_sequence(cls, sequence) -> "SequenceSummaryStats":
        a = np.asarray(sequence, dtype=np.float64)
        return cls(mean=float(a.mean()), std=float(a.std()), max=float(a.max()), min=float(a.min()))


@dataclasses.dataclass
class DictOfSequenceSummaryStats:
    """collector.py:14-30: ``stats[key]`` summarises the values of ``logger_stats[key]`` (graph.py:166-178)."""
    stats: dict

    @classmethod
    def from_dict(cls, stats: dict) -> "DictOfSequenceSummaryStats":
        return cls(stats={k: SequenceSummaryStats.from_sequence(v) for k, v in stats.items() if len(v)})


@dataclasses.dataclass
class CollectStatsWithInfo:
    """The reference collectors' return value, field for field (collector.py:33-36 on top of [3P] tianshou ``CollectStats``;
    built at multi_agent_collector.py:341-353).  ``returns[i]`` is episode i's reward sum as the env reports it at the
    episode's last step (``logger_stats['episode_rewards_sum']``, graph.py:166-178); ``info.stats[key]`` summarises the
    episodes' final ``logger_stats`` (the reference pools the ``logger_stats`` of EVERY collected step: same keys, a mean over
    more samples).  ``episode_info``: the per-episode values behind ``info``.  The mapping protocol (``result["n/ep"]`` ...)
    is the Tianshou-0.x spelling of the same numbers, kept for callers written against it."""
    n_collected_episodes: int = 0
    n_collected_steps: int = 0
    collect_time: float = 0.0
    collect_speed: float = 0.0
    returns: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(0))
    returns_stat: "SequenceSummaryStats | None" = None
    lens: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(0, int))
    lens_stat: "SequenceSummaryStats | None" = None
    info: "DictOfSequenceSummaryStats | None" = None
    episode_info: dict = dataclasses.field(default_factory=dict)

    _ALIASES = {"n/ep": "n_collected_episodes", "n/st": "n_collected_steps"}

    def __getitem__(self, key):
        if key in self._ALIASES:
            return getattr(self, self._ALIASES[key])
        if key in ("rew", "len"):
            stat = self.returns_stat if key == "rew" else self.lens_stat
            if stat is None:
                raise KeyError(key)
            return stat.mean
        if key != "info" and self.info is not None and key in self.info.stats:
            return self.info.stats[key].mean
        if key.startswith("_") or not hasattr(self, key):
            raise KeyError(key)
        return getattr(self, key)

    def __contains__(self, key):
        try:
            self[key]
            return True
        except KeyError:
            return False

    def get(self, key, default=None):
        return self[key] if key in self else default

    def keys(self):
        names = [f.name for f in dataclasses.fields(self)] + ["n/ep", "n/st"]
        if self.returns_stat is not None:
            names += ["rew", "len"]
        return names + (list(self.info.stats) if self.info is not None else [])

    def __iter__(self):
        return iter(self.keys())

    def items(self):
        return [(k, self[k]) for k in self.keys()]


def result_from_episode_log(stats: np.ndarray, meta: np.ndarray, total: int, steps: int, dt: float) -> CollectStatsWithInfo:
    """The collect result from the device's episode log (``HipGraphVectorEnv.read_episode_log``: one row per finished episode
    = the ``logger_stats`` of its last step and (env, pool episode, num_moves)), ``steps`` collected decisions, ``dt`` seconds."""
    episode_info = {k: stats[:, i].copy() for i, k in enumerate(_lib.LOGGER_KEYS)}
    returns = episode_info["episode_rewards_sum"].copy()
    lens = meta[:, 2].astype(int)
    return CollectStatsWithInfo(
        n_collected_episodes=total, n_collected_steps=steps, collect_time=dt, collect_speed=steps / dt,
        returns=returns, returns_stat=SequenceSummaryStats.from_sequence(returns) if len(returns) else None,
        lens=lens, lens_stat=SequenceSummaryStats.from_sequence(lens) if len(lens) else None,
        info=DictOfSequenceSummaryStats.from_dict(episode_info), episode_info=episode_info)


class Collector:
    """The collect() surface of the reference's collectors (multi_agent_collector.py:89-353: ``collect(n_step=...)`` /
    ``collect(n_episode=...)`` returning counts, speed and the episodes' ``logger_stats``) on top of the device-resident
    :class:`RoundLoop`: rounds are issued in chunks without host reads, the device counters and the on-device episode
    log (``mel_env_batch.log_*``) are read between chunks.  ``n_step`` counts live agent decisions (what the reference
    counts as collected transitions), ``n_episode`` finished episodes.  Evaluation as in l_dgn.py:92-129: build the
    vector env with ``is_testing=True`` and call ``collect(n_episode=...)``."""

    def __init__(self, policy, venv: HipGraphVectorEnv, episodes_per_env: int = 16, seed: int = 0, eps: float = 0.0,
                 replay=None, log_capacity: int = 65536, chunk: int = 8, use_graph: bool = True):
        self.venv, self.policy, self.chunk = venv, policy, int(chunk)
        venv.enable_episode_log(log_capacity)
        self.loop = RoundLoop(venv, policy, episodes_per_env=episodes_per_env, seed=seed, eps=eps, replay=replay,
                              use_graph=use_graph)
        self._decisions = self.loop.counters()["decisions"]
        self.collect_step, self.collect_episode, self.collect_time = 0, 0, 0.0

    def collect(self, n_step: int | None = None, n_episode: int | None = None) -> CollectStatsWithInfo:
        import time
        if (n_step is None) == (n_episode is None):
            raise ValueError("give exactly one of n_step / n_episode")        # the reference asserts the same
        self.venv.log_cursor.zero_()
        t0 = time.perf_counter()
        steps = episodes = 0
        with torch.no_grad():
            while (n_step is not None and steps < n_step) or (n_episode is not None and episodes < n_episode):
                self.loop.run(self.chunk)
                c = self.loop.counters()                                      # synchronises
                steps = c["decisions"] - self._decisions
                episodes = int(self.venv.log_cursor.item())
                if c["errors"]:
                    raise RuntimeError(f"env error flags {c['errors']:#x} (episode pool exhausted or desynchronised actions)")
        dt = max(time.perf_counter() - t0, 1e-9)
        stats, meta, total = self.venv.read_episode_log()
        self._decisions += steps
        self.collect_step += steps
        self.collect_episode += total
        self.collect_time += dt
        return result_from_episode_log(stats, meta, total, steps, dt)

    def set_eps(self, eps: float) -> None:
        """Exploration rate of the following collects.  The rate is an argument of the selection fused into the forward's last
        launch: in HIP-graph mode a new value means a new capture (taken by the next round)."""
        eps = float(eps)
        if eps != self.loop.eps:
            self.loop.eps = eps
            self.loop._select.eps = eps
            self.loop.graph = None


class MultiAgentCollector(Collector):
    """``MultiAgentCollector(agents_num, policy=..., env=..., buffer=None, exploration_noise=False)`` with
    ``collect(n_step | n_episode, random, render, no_grad)`` -> :class:`CollectStatsWithInfo` - the reference's constructor and
    call (multi_agent_collector.py:31-42, 89-118; call sites l_dgn.py:119-127, 185-201) in front of the device-resident
    :class:`RoundLoop`.  One line changes in the reference's scripts: the import.

    * ``env``: a :class:`melissa_amd.env.HipGraphVectorEnv` (the vector env of the scripts' ``DummyVectorEnv([...])``);
      ``agents_num`` must be its node count.
    * ``buffer``: a :class:`melissa_amd.replay.RoundReplay` or None.  The reference routes every agent's transition into one of
      ``env_num * agents_num`` sub-buffers and holds it back until the agent's next observation arrives
      (multi_agent_collector.py:229-271); here a whole env round is one record and ``RoundReplay.export_transitions()``
      yields the same transitions with the reference's ``buffer_id = env * agents_num + agent``.
    * ``exploration_noise``: eps-greedy with ``policy.eps`` (set by ``policy.set_eps`` in the scripts' ``train_fn``) when True,
      greedy when False - [3P] ``DQNPolicy.exploration_noise`` through shared_policy.py:81-91, drawn on the device.
    * ``random=True``: uniformly random actions (eps = 1), as ``self._action_space[i].sample()`` does.
    * ``render``: not offered (the reference's matplotlib view, graph.py:466-484, is out of scope): a truthy value raises.
    * ``no_grad``: accepted and ignored - the collect path never builds an autograd graph (the HIP forward has none)."""

    def __init__(self, agents_num, policy=None, env=None, buffer=None, exploration_noise: bool = False, preprocess_fn=None,
                 seed: int = 0, episodes_per_env: int = 16, chunk: int = 8, use_graph: bool = True, log_capacity: int = 65536):
        if policy is None or env is None:
            raise TypeError("MultiAgentCollector needs policy= and env=")
        if preprocess_fn is not None:
            raise NotImplementedError("preprocess_fn: the rounds never leave the device; nothing to hook per step")
        if int(agents_num) != env.n:
            raise ValueError(f"agents_num={agents_num} but the env has {env.n} agents")
        self.agents_num = int(agents_num)
        self.exploration_noise = bool(exploration_noise)
        self.env, self.buffer = env, buffer
        policy = getattr(policy, "policy", policy)          # a MultiAgentSharedPolicy manager wraps the one shared policy
        eps = float(getattr(policy, "eps", 0.0)) if exploration_noise else 0.0
        super().__init__(policy, env, episodes_per_env=episodes_per_env, seed=seed, eps=eps, replay=buffer,
                         log_capacity=log_capacity, chunk=chunk, use_graph=use_graph)

    @property
    def env_num(self) -> int:
        return self.venv.env_num

    def reset(self, reset_buffer: bool = True, gym_reset_kwargs=None) -> None:
        """[3P] ``Collector.reset``: statistics (and the buffer's write cursors) back to zero; the envs keep running - every
        episode end already draws a fresh episode on the device."""
        self.reset_stat()
        if reset_buffer:
            self.reset_buffer()

    def reset_stat(self) -> None:
        self.collect_step, self.collect_episode, self.collect_time = 0, 0, 0.0

    def reset_buffer(self, keep_statistics: bool = False) -> None:
        if self.buffer is not None:
            self.buffer.cursor.zero_()
            self.buffer.episode.fill_(-1)

    def reset_env(self, gym_reset_kwargs=None) -> None:
        """(the reference resets all envs after an n_episode collect; the device envs reset themselves at every episode end)"""

    def collect(self, n_step: int | None = None, n_episode: int | None = None, random: bool = False, render=False,
                no_grad: bool = False, gym_render_kwargs=None) -> CollectStatsWithInfo:
        if render:
            raise NotImplementedError("render: the reference's matplotlib view (graph.py:466-484) is out of scope")
        if n_step is not None and n_episode is not None:
            raise AssertionError(f"Only one of n_step or n_episode is allowed in Collector.collect, got n_step={n_step}, "
                                 f"n_episode={n_episode}.")
        if n_step is None and n_episode is None:
            raise TypeError("Please specify at least one (either n_step or n_episode) in AsyncCollector.collect().")
        if (n_step is not None and n_step <= 0) or (n_episode is not None and n_episode <= 0):
            raise AssertionError("n_step / n_episode must be positive")
        eps = 1.0 if random else (float(getattr(self.policy, "eps", 0.0)) if self.exploration_noise else 0.0)
        self.set_eps(eps)
        return super().collect(n_step=n_step, n_episode=n_episode)
