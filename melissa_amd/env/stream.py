"""Continuous episode supply for the device-resident loops.

The reference draws a fresh episode on EVERY reset (``World.reset``, graph_env/env/utils/core.py:343-437: episode seed
and graph from the env's generator, then source / interested set / movement seed from ``RandomState(episode_seed)``).
A pre-drawn table that wraps would make a training run replay the same handful of episodes per env for ever, so the
loops (``melissa_amd.collect``) take their episodes from an :class:`EpisodeStream` by default:

* per env a RING of ``ring`` pool slots (episode j of env b lives in slot ``b * ring + j % ring``), the reset snapshot of
  every slot next to it;
* ``mel_episode_refill`` (csrc/episode_stream.hpp) draws the next episodes of every env ON THE DEVICE with numpy's exact
  algorithms (PCG64 ``Generator.integers`` / ``choice``; legacy MT19937 ``RandomState`` randint / uniform / choice /
  the ``0.06 * uniform(-1, 1)`` movement offsets), copies the graph out of the packed dataset in HBM (the replacement of
  ``pickle.load`` per episode, core.py:450-452), and runs ``GraphEnv.reset`` + ``World.reset`` into the slot's snapshot;
* the loop calls :meth:`before_step` / :meth:`after_step` once per iteration: every ``period`` iterations the refill is
  issued on a SIDE stream - ordered after the main stream's work so far by an event (it reads exact episode cursors; the
  main stream waits for it one period later), or, in the round loop, paced by a gate on the device-side round counter.  An env starts at most one episode per iteration, so with
  ``2 * period <= ring - 1`` no env can reach a slot that is being written or an episode that is not there yet; the env
  kernels check that anyway (``MEL_ENV_ERR_EPISODE_UNDERRUN``).

:class:`StaticSupply` is the old behaviour for explicitly given tables (tests) and for the modes the device sampler
does not cover (evaluation schedule, scripted agents, a moving fixed graph): a fixed table; wrapping raises the same
error flag unless the table is periodic by construction (the evaluation schedule is, core.py:351-352).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from .. import _lib
from .episodes import pack_episodes


def stream_supported(venv) -> bool:
    kw = venv._sampler_kw
    return not (kw["is_testing"] or kw["scripted_agents_ratio"] > 0.0 or (venv.fixed_graph and venv.dynamic_graph))


class StaticSupply:
    """A pre-drawn episode table (``packed`` pool dict, ``table`` int32 [B, K])."""

    kind = "static table"

    def __init__(self, venv, packed: dict, table: np.ndarray, reset_snapshots: bool, check_wrap: bool = True):
        if venv.fixed_graph and venv.dynamic_graph:
            # GraphEnv(graph=...) with dynamic_graph mutates its ONE graph across episodes (core.py:130,303-314: no reload
            # at reset), so an episode starts from wherever the previous one left the nodes.  The on-device resets of the
            # loops load every episode's positions from the pool (keep_graph = 0) and would restart from the original
            # graph: refuse instead of silently diverging.  The host path (HipGraphVectorEnv.reset / step) handles it.
            raise ValueError("a fixed graph that moves carries its positions over between episodes (core.py:303-314); the "
                             "device-resident loops reset from the episode pool - drive this configuration through "
                             "HipGraphVectorEnv.reset()/step(), or use a graph pool")
        self.venv = venv
        self.pool = venv.load_pool(packed, reset_snapshots=reset_snapshots)
        self.table = torch.from_numpy(np.ascontiguousarray(table, dtype=np.int32)).to(venv.device)
        self.ring = int(table.shape[1])
        kw = venv._sampler_kw
        periodic = kw["is_testing"] and kw["scripted_agents_ratio"] == 0.0 and self.ring % kw["num_test_episodes"] == 0
        if check_wrap and not periodic:          # starting episode number >= K would replay one: flag it
            self.produced = torch.full((venv.env_num,), self.ring, dtype=torch.int32, device=venv.device)
            self.pool.struct.produced = self.produced.data_ptr()

    def first_episode_ids(self) -> torch.Tensor:
        return self.table[:, 0].contiguous()

    def before_step(self, iteration: int, round_counter=None):
        pass

    def after_step(self, iteration: int, round_counter=None):
        pass

    def describe(self) -> dict:
        return {"mode": "static table", "episodes_per_env": self.ring, "graphs": len(self.venv.graphs)}


class EpisodeStream:
    """Device-resident sampler + ring pool (see the module docstring).  ``seed``: env b's generator is seeded
    ``seed + b`` (tianshou ``BaseVectorEnv.seed``), exactly like ``HipGraphVectorEnv.make_sampler(seed + b)``.
    ``discard``: episodes every env draws and drops first (the reference samples 2-3 episodes while an env is
    constructed: World.__init__ core.py:190, GraphEnv.__init__ graph.py:118, [3P] PettingZooEnv.__init__)."""

    kind = "device stream"

    def __init__(self, venv, seed, ring: int = 16, discard: int = 0, period: int | None = None):
        if not stream_supported(venv):
            raise ValueError("the device sampler covers training mode without scripted agents (and no moving fixed graph)")
        if ring < 3:
            raise ValueError("ring must be >= 3")
        self.venv, self.ring = venv, int(ring)
        self.period = int(period) if period is not None else max(1, (self.ring - 1) // 2)
        if 2 * self.period > self.ring - 1:
            raise ValueError(f"period {self.period} needs ring >= {2 * self.period + 1}")
        self.lib = _lib.load()
        dev, B, n, K = venv.device, venv.env_num, venv.n, self.ring
        # ---- the envs' generators: numpy's own seeding (SeedSequence -> PCG64 state), uploaded once
        pcg = np.zeros((B, 4), dtype=np.uint64)
        half = np.zeros((B, 2), dtype=np.uint32)
        mask = (1 << 64) - 1
        for b in range(B):
            st = np.random.PCG64(np.random.SeedSequence(None if seed is None else seed + b)).state
            pcg[b] = [st["state"]["state"] & mask, st["state"]["state"] >> 64, st["state"]["inc"] & mask, st["state"]["inc"] >> 64]
            half[b] = [st["has_uint32"], st["uinteger"]]
        self.pcg = torch.from_numpy(pcg.view(np.int64)).to(dev)
        self.pcg_half = torch.from_numpy(half.view(np.int32)).to(dev)
        self.produced = torch.zeros(B, dtype=torch.int32, device=dev)
        self.draw_seed = torch.zeros(B * K, dtype=torch.int32, device=dev)
        self.draw_graph = torch.zeros(B * K, dtype=torch.int32, device=dev)
        self.work = torch.zeros(1 + 2 * B * K, dtype=torch.int32, device=dev)
        self.new_count = torch.zeros(B, dtype=torch.int32, device=dev)
        # ---- the packed graph dataset in HBM
        self.graph_pos = torch.from_numpy(np.stack([g.pos for g in venv.graphs]).astype(np.float64)).to(dev)
        self.graph_hop = torch.from_numpy(np.stack([g.one_hop for g in venv.graphs]).astype(np.uint64).view(np.int64)).to(dev)
        self.graphs = _lib.MelGraphPool()
        self.graphs.n_graphs, self.graphs.n_nodes = len(venv.graphs), n
        self.graphs.pos, self.graphs.one_hop = self.graph_pos.data_ptr(), self.graph_hop.data_ptr()
        # ---- the ring pool (allocated on the device: B*K slots) + its snapshot batch
        from .vector_env import DevicePool
        self.pool = DevicePool.empty(B * K, n, venv.max_moves, venv.dynamic_graph, dev)
        self.pool.alloc_snapshots(self.lib, venv.env)
        self.pool.struct.produced = self.produced.data_ptr()
        self.table = torch.arange(B * K, dtype=torch.int32, device=dev).view(B, K).contiguous()
        s = _lib.MelEpisodeStream()
        s.n_envs, s.ring, s.fixed_graph = B, K, int(venv.fixed_graph)
        dens = venv._sampler_kw["fixed_interest_density"]
        s.has_density, s.fixed_interest_density = int(dens is not None), float(dens or 0.0)
        s.pcg, s.pcg_half, s.produced = self.pcg.data_ptr(), self.pcg_half.data_ptr(), self.produced.data_ptr()
        s.draw_seed, s.draw_graph = self.draw_seed.data_ptr(), self.draw_graph.data_ptr()
        s.work, s.new_count = self.work.data_ptr(), self.new_count.data_ptr()
        self.struct = s
        self.side = torch.cuda.Stream(device=dev)
        self._ev_main = torch.cuda.Event()
        self._ev_done = None
        self.refills = 0
        self._max_new = int(os.environ["MEL_STREAM_MAX_NEW"]) if "MEL_STREAM_MAX_NEW" in os.environ else None   # tuning only
        if "MEL_STREAM_PERIOD" in os.environ:
            self.period = int(os.environ["MEL_STREAM_PERIOD"])
        # first fill on the current stream: the loop resets its envs from slot 0 right after
        self.refill(discard=discard)

    def refill(self, max_new: int | None = None, discard: int = 0):
        """Issue one refill on the CURRENT stream (the cursors it reads are the ones that stream has produced)."""
        _lib.check(self.lib.mel_episode_refill(C.byref(self.struct), C.byref(self.graphs), C.byref(self.pool.struct),
                                               C.byref(self.venv.env), self.ring if max_new is None else int(max_new),
                                               int(discard), _lib.current_stream_ptr(self.venv.device)), "mel_episode_refill")
        self.refills += 1

    def first_episode_ids(self) -> torch.Tensor:
        return self.table[:, 0].contiguous()

    def _paced(self, round_counter) -> bool:
        # paced mode needs slack: a refill gated behind iteration i reads the cursors that iteration left, finishes within the
        # next iteration or two, and must cover every episode an env can start until the NEXT refill has finished, i.e. up to
        # period + 1 more: period <= ring - 3 (one more of margin kept).  Small test rings fall back to events.
        return (round_counter is not None and self.period <= self.ring - 4
                and os.environ.get("MEL_STREAM_SYNC", "paced") == "paced")

    def before_step(self, iteration: int, round_counter: "torch.Tensor | None" = None):
        """Call before launching iteration number ``iteration`` (0-based) on the current stream.  Every ``period``
        iterations one refill is issued on the side stream, ordered after the main stream's progress in one of two ways:

        * ``round_counter`` given (the device counter mel_env_round advances once per round, RoundLoop): nothing happens
          here - the loop calls :meth:`after_step` once the iteration's launches are enqueued.
        * no counter (DecisionLoop): events - the side stream waits for an event recorded on the main stream, and the main
          stream waits for the refill one period later."""
        if iteration % self.period or self._paced(round_counter):
            return
        main = torch.cuda.current_stream(self.venv.device)
        if self._ev_done is not None:
            main.wait_event(self._ev_done)            # the refill issued one period ago must be complete from here on
        self._ev_main.record(main)
        self.side.wait_event(self._ev_main)           # exact episode cursors: everything issued so far has run
        with torch.cuda.stream(self.side):
            self.refill(max_new=self._max_new)
            if self._ev_done is None:
                self._ev_done = torch.cuda.Event()
            self._ev_done.record(self.side)

    def after_step(self, iteration: int, round_counter: "torch.Tensor | None" = None):
        """Call after iteration number ``iteration``'s launches are ENQUEUED on the current stream (RoundLoop).  Paced mode:
        a gate (mel_wait_counter) on the SIDE stream polls the device-side round counter until that iteration's env launch
        has started, then the refill runs; nothing touches the main stream (an event record / wait between HIP-graph
        replays costs the replayed step ~8 us on this stack, measured).  The gate only ever waits for work that is already
        in a queue: issued BEFORE the launch it waits for (as round 2's first version did) it stalls for its whole 2 s bound
        whenever the runtime maps the side stream and the main stream onto the same in-order hardware queue (seen in a
        process that had created other streams before: 28 refills x 2 s in a 200-step run).  Nothing orders the main
        stream behind the refill either: a refill takes ~0.15 ms and no env can need its episodes before
        ``ring - period - 3`` more iterations; should that ever fail, the env kernels' produced check raises
        MEL_ENV_ERR_EPISODE_UNDERRUN (an error, never a stale slot)."""
        if iteration % self.period or not self._paced(round_counter):
            return
        with torch.cuda.stream(self.side):
            # (the 2 s bound only guards the process exit; the host can run hundreds of replays ahead of the GPU)
            _lib.check(self.lib.mel_wait_counter(round_counter.data_ptr(), (iteration + 1) & 0xFFFFFFFF, 2000000,
                                                 _lib.current_stream_ptr(self.venv.device)), "mel_wait_counter")
            self.refill(max_new=self._max_new)

    def drawn(self) -> int:
        """Episodes drawn so far over all envs (synchronises)."""
        return int(self.produced.sum().item())

    def describe(self) -> dict:
        return {"mode": "device stream", "ring": self.ring, "refill_every": self.period, "graphs": len(self.venv.graphs),
                "refills": self.refills}


def make_supply(venv, seed, episodes=None, episodes_per_env: int = 8, stream: bool | None = None, ring: int = 16,
                discard: int = 0, reset_snapshots: bool = True):
    """``episodes`` = (packed, table): that static table.  Otherwise a device stream where the sampler covers the env's
    mode (``stream`` None / True) or a host-drawn table of ``episodes_per_env`` episodes (``stream`` False, or a mode
    the device sampler does not cover)."""
    from ..collect import sample_episode_table
    if episodes is not None:
        return StaticSupply(venv, episodes[0], episodes[1], reset_snapshots)
    if stream is None:
        stream = stream_supported(venv)
    if stream:
        return EpisodeStream(venv, seed, ring=ring, discard=discard)
    packed, table = sample_episode_table(venv, episodes_per_env, seed)
    return StaticSupply(venv, packed, table, reset_snapshots)
