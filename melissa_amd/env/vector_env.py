"""``HipGraphVectorEnv``: B message-dissemination envs stepped by HIP kernels, behind the surface the
reference's collectors use on a tianshou vector env of ``PettingZooEnv(GraphEnv)``
(multi_agent_collector.py:119,192-195,296): ``__len__``, ``is_async``, ``env_num``,
``action_space``, ``seed``, ``reset(id)``, ``step(action, id)`` returning stacked NumPy
(obs = array of ``{agent_id, obs, mask}`` dicts, ``rew [len(id), N]``, ``terminated``, ``truncated``,
``info`` dicts with ``env_id`` / ``env_step`` / ``environment_step`` / ``explicit_reset`` /
``active_one_hop_neighbors`` / ``logger_stats``).

The same object also exposes the device-resident path (``step_device`` / ``reset_device``) that the
vectorised decision loop uses: no host round trip, observations land directly in the network's input
buffer.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib
from .episodes import EpisodeSampler, Graph, pack_episodes, set_shape, set_words, sets_to_bool

LOGGER_KEYS = ("total_messages_transmitted", "coverage", "messages_sent", "messages_received", "n_neighbours",
               "interested_agents", "coverage_interested_fraction", "coverage_interested_count",
               "uninterested_with_message", "episode_rewards_sum")          # graph.py:167-177


class Discrete:
    """Minimal gym ``Discrete(n)`` stand-in for ``action_space`` (graph.py:103)."""

    def __init__(self, n, rng=None):
        self.n = n
        self._rng = rng or np.random.default_rng()

    def sample(self):
        return int(self._rng.integers(0, self.n))

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)


class DevicePool:
    """A mel_episode_pool resident in HBM."""

    def __init__(self, packed: dict, n: int, device):
        self.tensors = {k: torch.from_numpy(np.ascontiguousarray(v).view(np.int64) if v.dtype == np.uint64 else
                                            np.ascontiguousarray(v)).to(device) for k, v in packed.items()}
        self.struct = _lib.MelEpisodePool()
        self.struct.n_episodes = packed["origin"].shape[0]
        self.struct.n_nodes = n
        self.struct.max_moves = packed["moves"].shape[1]
        self.snapshot_env = self.snapshot_state = None
        self.refresh()

    @classmethod
    def empty(cls, n_episodes: int, n: int, max_moves: int, dynamic: bool, device) -> "DevicePool":
        """An all-zero pool allocated ON the device (the ring of an episode stream: hundreds of MB of movement offsets that
        the device sampler fills - nothing to upload)."""
        self = cls.__new__(cls)
        z = lambda *shape, dtype: torch.zeros(*shape, dtype=dtype, device=device)
        ws = set_shape(n)                          # node sets: one int64 word per set up to 64 nodes, [W] beyond
        self.tensors = dict(pos=z(n_episodes, n, 2, dtype=torch.float64), one_hop=z(n_episodes, n, *ws, dtype=torch.int64),
                            interested=z(n_episodes, *ws, dtype=torch.int64), origin=z(n_episodes, dtype=torch.int32),
                            moves=z(n_episodes, max_moves if dynamic else 1, 2, n, dtype=torch.float64),
                            scripted=z(n_episodes, *ws, dtype=torch.int64))
        self.struct = _lib.MelEpisodePool()
        self.struct.n_episodes, self.struct.n_nodes = n_episodes, n
        self.struct.max_moves = max_moves if dynamic else 1
        self.snapshot_env = self.snapshot_state = None
        self.refresh()
        return self

    def alloc_snapshots(self, lib, like_env):
        """A snapshot batch (one env per pool slot, ``like_env``'s settings) whose resets somebody else runs
        (mel_episode_refill)."""
        e, n = int(self.struct.n_episodes), int(self.struct.n_nodes)
        self.drop_snapshots()
        self.snapshot_state = torch.zeros(int(lib.mel_env_state_bytes(e, n)), dtype=torch.uint8,
                                          device=self.tensors["origin"].device)
        env = _lib.MelEnvBatch()
        for name in ("dynamic_graph", "has_local_ratio", "local_ratio", "heuristic", "is_testing"):
            setattr(env, name, getattr(like_env, name))
        _lib.check(lib.mel_env_bind(C.byref(env), e, n, self.snapshot_state.data_ptr()), "mel_env_bind")
        self.snapshot_env = env
        self.struct.snapshot = C.addressof(env)

    def refresh(self):
        t = self.tensors
        s = self.struct
        s.pos, s.one_hop, s.interested = t["pos"].data_ptr(), t["one_hop"].data_ptr(), t["interested"].data_ptr()
        s.origin, s.moves = t["origin"].data_ptr(), t["moves"].data_ptr()
        s.scripted = t["scripted"].data_ptr() if "scripted" in t else None

    def drop_snapshots(self):
        self.struct.snapshot = None
        self.snapshot_env = self.snapshot_state = None

    def build_snapshots(self, lib, like_env, stream_ptr):
        """Reset snapshots (mel_episode_pool.snapshot): run mel_env_reset once per episode into a private env batch with
        ``like_env``'s settings; mel_env_round then loads an ending env's next episode from there instead of recomputing
        GraphEnv.reset + World.reset (a pure function of the pre-drawn episode)."""
        e, n = int(self.struct.n_episodes), int(self.struct.n_nodes)
        dev = self.tensors["origin"].device
        self.drop_snapshots()
        self.snapshot_state = torch.zeros(int(lib.mel_env_state_bytes(e, n)), dtype=torch.uint8, device=dev)
        env = _lib.MelEnvBatch()
        for name in ("dynamic_graph", "has_local_ratio", "local_ratio", "heuristic", "is_testing"):
            setattr(env, name, getattr(like_env, name))
        _lib.check(lib.mel_env_bind(C.byref(env), e, n, self.snapshot_state.data_ptr()), "mel_env_bind")
        ids = torch.arange(e, dtype=torch.int32, device=dev)
        _lib.check(lib.mel_env_reset(C.byref(env), C.byref(self.struct), None, ids.data_ptr(), e, 0, None, stream_ptr),
                   "mel_env_reset (snapshots)")
        torch.cuda.synchronize(dev)
        self.snapshot_env = env                                  # keeps the host struct alive
        self.struct.snapshot = C.addressof(env)

    def write(self, slot_ids: np.ndarray, packed: dict):
        self.drop_snapshots()                                    # the episodes change: snapshots would be stale
        idx = torch.as_tensor(slot_ids, dtype=torch.long, device=self.tensors["origin"].device)
        for k, v in packed.items():
            src = np.ascontiguousarray(v).view(np.int64) if v.dtype == np.uint64 else np.ascontiguousarray(v)
            self.tensors[k][idx] = torch.from_numpy(src).to(idx.device)


class ObsBuffers:
    """Device outputs of last() (+ the ctypes view handed to the C ABI)."""

    def __init__(self, rows: int, n: int, device, obs: torch.Tensor | None = None):
        w = 8 * n + 1
        self.obs = obs if obs is not None else torch.empty(rows, w, dtype=torch.float32, device=device)
        self.agent_id = torch.empty(rows, dtype=torch.int32, device=device)
        self.action_mask = torch.empty(rows, 2, dtype=torch.uint8, device=device)
        self.rew = torch.empty(rows, n, dtype=torch.float64, device=device)
        self.terminated = torch.empty(rows, dtype=torch.uint8, device=device)
        self.flags = torch.empty(rows, 4, dtype=torch.int32, device=device)
        self.active_nb = torch.empty(rows, *set_shape(n), dtype=torch.int64, device=device)
        self.stats = torch.empty(rows, _lib.ENV_LOGGER_STATS, dtype=torch.float64, device=device)
        s = _lib.MelEnvObs()
        s.obs, s.obs_stride = self.obs.data_ptr(), self.obs.stride(0)
        s.agent_id, s.action_mask, s.rew = self.agent_id.data_ptr(), self.action_mask.data_ptr(), self.rew.data_ptr()
        s.terminated, s.flags = self.terminated.data_ptr(), self.flags.data_ptr()
        s.active_nb, s.stats = self.active_nb.data_ptr(), self.stats.data_ptr()
        self.struct = s


class HipGraphVectorEnv:
    is_async = False

    def __init__(self, env_num: int, number_of_agents: int, graph_pool=None, graph: Graph | None = None,
                 dynamic_graph: bool = False, local_ratio=None, device="cuda", max_moves: int = 64,
                 seed=None, fixed_interest_density=None, construct_like_reference: "bool | int" = True,
                 is_testing: bool = False, num_test_episodes: int = 10, scripted_agents_ratio: float = 0.0,
                 heuristic: str | None = None):
        """``graph`` fixes one graph for every episode (GraphEnv(graph=...)); ``graph_pool`` is a list of
        ``Graph`` standing for the ``graph_topologies/training_N/*`` files.  ``seed`` seeds env k's
        generator with ``seed + k`` (tianshou ``BaseVectorEnv.seed``).  ``construct_like_reference``
        replays the episode samplings the reference performs while CONSTRUCTING an env, so that RNG streams
        line up with a reference run: ``True`` (= 3) for ``PettingZooEnv(GraphEnv(...))`` as ``get_env`` builds it
        (common.py:100-129: World.__init__ -> reset core.py:190, GraphEnv.__init__ -> reset graph.py:118, [3P]
        tianshou PettingZooEnv.__init__ -> env.reset()), ``2`` for a bare ``GraphEnv`` (what the golden traces
        drive), ``False`` for none.  ``is_testing``: the
        reference's evaluation schedule (GraphEnv(is_testing=True, num_test_episodes=...), core.py:182-187,348-370);
        ``graph_pool`` then stands for the sorted ``graph_topologies/testing_N/*`` files."""
        _lib.check_n_nodes(number_of_agents, "HipGraphVectorEnv")
        if is_testing and graph is not None:
            raise ValueError("testing mode draws its graph from graph_pool (core.py:355-359)")
        # core.py:143-148: the same argument checks, same messages
        if not (0.0 <= scripted_agents_ratio <= 1.0):
            raise ValueError("`scripted_agents_ratio` must be in [0.0, 1.0].")
        if scripted_agents_ratio == 0.0 and heuristic is not None:
            raise ValueError("If `scripted_agents_ratio` is 0.0, no heuristic can be set.")
        if heuristic not in _lib.HEURISTICS:         # None with a ratio > 0: scripted nodes never act (no callback)
            raise ValueError(f"Unknown heuristic policy: {heuristic} (the HIP env offers the deterministic ones: "
                             f"{[k for k in _lib.HEURISTICS if k]})")
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("HipGraphVectorEnv needs a ROCm GPU; there is no CPU fallback")
        assert (graph is None) != (graph_pool is None), "give exactly one of graph / graph_pool"
        self.env_num, self.n = int(env_num), int(number_of_agents)
        self.device = torch.device(device)
        self.dynamic_graph, self.local_ratio, self.max_moves = bool(dynamic_graph), local_ratio, int(max_moves)
        self.fixed_graph = graph is not None
        self.graphs = [graph] if graph is not None else list(graph_pool)
        self.action_space = [Discrete(2) for _ in range(self.env_num)]
        seeds = [None] * self.env_num if seed is None else [seed + k for k in range(self.env_num)]
        self._sampler_kw = dict(fixed_interest_density=fixed_interest_density, is_testing=is_testing,
                                num_test_episodes=num_test_episodes, scripted_agents_ratio=scripted_agents_ratio)
        self.samplers = [self.make_sampler(s) for s in seeds]
        # device state
        nbytes = int(self.lib.mel_env_state_bytes(self.env_num, self.n))
        self.state = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        self.env = _lib.MelEnvBatch()
        self.env.dynamic_graph = int(self.dynamic_graph)
        self.env.has_local_ratio = int(local_ratio is not None)
        self.env.local_ratio = float(local_ratio) if local_ratio is not None else 0.0
        self.env.heuristic, self.env.is_testing = _lib.HEURISTICS[heuristic], int(bool(is_testing))
        _lib.check(self.lib.mel_env_bind(C.byref(self.env), self.env_num, self.n, self.state.data_ptr()), "mel_env_bind")
        # one staging pool slot per env for the host-sampled ("tianshou compatible") reset path
        empty = pack_episodes([], self.graphs, self.n, self.max_moves, self.dynamic_graph)
        empty = {k: np.zeros((self.env_num,) + v.shape[1:], dtype=v.dtype) for k, v in empty.items()}
        self.staging = DevicePool(empty, self.n, self.device)
        self.out = ObsBuffers(self.env_num, self.n, self.device)
        self._graph_loaded = np.zeros(self.env_num, dtype=bool)
        self._ids_all = torch.arange(self.env_num, dtype=torch.int32, device=self.device)
        for _ in range(3 if construct_like_reference is True else int(construct_like_reference)):
            self._reset_rows(np.arange(self.env_num), observe=False)

    def enable_episode_log(self, capacity: int):
        """Device-side log of finished episodes (on-device resets only): one row of the reference's ``logger_stats``
        (graph.py:166-178) per episode, what ``MultiAgentCollector`` gathers into ``episode_info``."""
        self.log_cursor = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.log_stats = torch.zeros(capacity, _lib.ENV_LOGGER_STATS, dtype=torch.float64, device=self.device)
        self.log_meta = torch.zeros(capacity, 3, dtype=torch.int32, device=self.device)
        self.env.log_capacity = int(capacity)
        self.env.log_cursor, self.env.log_stats = self.log_cursor.data_ptr(), self.log_stats.data_ptr()
        self.env.log_meta = self.log_meta.data_ptr()

    def read_episode_log(self, reset: bool = False):
        """-> (stats float64 [k, 10] in _lib.LOGGER_KEYS order, meta int32 [k, 3] = env, pool episode, num_moves,
        total episodes ended incl. rows dropped beyond the capacity).  Synchronises."""
        total = int(self.log_cursor.item())
        k = min(total, int(self.env.log_capacity))
        out = self.log_stats[:k].cpu().numpy(), self.log_meta[:k].cpu().numpy(), total
        if reset:
            self.log_cursor.zero_()
        return out

    def make_sampler(self, seed) -> EpisodeSampler:
        """An episode sampler with this env's settings (graph pool size, evaluation schedule, scripted ratio, ...)
        and its own generator seeded ``seed`` - what the device-resident loops pre-draw their episode pools with."""
        return EpisodeSampler(self.n, np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed))),
                              len(self.graphs), self.fixed_graph, **self._sampler_kw)

    # ------------------------------------------------------------------ plumbing
    def __len__(self):
        return self.env_num

    def _stream(self):
        return _lib.current_stream_ptr(self.device)

    def _ids(self, id):
        if id is None:
            return np.arange(self.env_num)
        return np.atleast_1d(np.asarray(id)).astype(np.int64)

    def seed(self, seed=None):
        seeds = [None] * self.env_num if seed is None else ([seed + k for k in range(self.env_num)]
                                                            if np.isscalar(seed) else list(seed))
        for s, sampler in zip(seeds, self.samplers):
            sampler.seed(s)
        return seeds

    def scalars(self) -> torch.Tensor:
        """[B, 16] int32 view of the MEL_S_* scalars (device)."""
        off = self.env.scalars - self.state.data_ptr()
        return self.state[off: off + self.env_num * _lib.ENV_SCALARS * 4].view(torch.int32).view(self.env_num, -1)

    def node_sets(self) -> torch.Tensor:
        """[B, 8] int64 bit patterns of the MEL_SET_* node sets (device); [B, 8, W] beyond 64 nodes."""
        off = self.env.node_sets - self.state.data_ptr()
        w = set_words(self.n)
        return self.state[off: off + self.env_num * 8 * 8 * w].view(torch.int64).view(self.env_num, 8, *set_shape(self.n))

    def _field(self, ptr, per_env, dtype):
        off = ptr - self.state.data_ptr()
        nbytes = self.env_num * per_env * torch.empty((), dtype=dtype).element_size()
        return self.state[off: off + nbytes].view(dtype).view(self.env_num, per_env)

    def obs_matrix(self) -> torch.Tensor:
        """[B, 8N] fp32 view of GraphEnv.obs_matrix of every env (device; what the round-batched forward reads)."""
        return self._field(self.env.obs_matrix, 8 * self.n, torch.float32)

    def positions(self):
        return self._field(self.env.pos, 2 * self.n, torch.float64).view(self.env_num, self.n, 2)

    def one_hop(self):
        """[B, N] int64 bit patterns ([B, N, W] beyond 64 nodes)."""
        return self._field(self.env.one_hop, self.n * set_words(self.n), torch.int64).view(self.env_num, self.n, *set_shape(self.n))

    def two_hop(self):
        return self._field(self.env.two_hop, self.n * set_words(self.n), torch.int64).view(self.env_num, self.n, *set_shape(self.n))

    # ------------------------------------------------------------------ reset / step
    def _reset_rows(self, ids: np.ndarray, observe: bool = True):
        episodes = [self.samplers[i].sample() for i in ids]
        self.staging.write(ids, pack_episodes(episodes, self.graphs, self.n, self.max_moves, self.dynamic_graph))
        ids_t = torch.as_tensor(ids, dtype=torch.int32, device=self.device)
        if self.fixed_graph:
            # the reference mutates its one graph object across episodes (core.py:130,303-314): only the
            # first load takes positions/edges from the pool, later resets keep the env's current graph
            fresh = ids[~self._graph_loaded[ids]]
            kept = ids[self._graph_loaded[ids]]
            groups = [(fresh, 0), (kept, 1)]
        else:
            groups = [(ids, 0)]
        for rows, keep in groups:
            if len(rows) == 0:
                continue
            rows_t = torch.as_tensor(rows, dtype=torch.int32, device=self.device)
            pos_in_ids = torch.as_tensor(np.searchsorted(ids, rows) if np.all(np.diff(ids) > 0)
                                         else [list(ids).index(r) for r in rows], dtype=torch.long, device=self.device)
            out = None
            if observe:
                out = self._sub_out(pos_in_ids)
            _lib.check(self.lib.mel_env_reset(C.byref(self.env), C.byref(self.staging.struct), rows_t.data_ptr(),
                                              rows_t.data_ptr(), len(rows), keep,
                                              C.byref(out.struct) if out is not None else None, self._stream()),
                       "mel_env_reset")
            if observe:
                self._scatter_out(out, pos_in_ids)
        self._graph_loaded[ids] = True
        del ids_t

    def _sub_out(self, pos_in_ids):
        return ObsBuffers(len(pos_in_ids), self.n, self.device)

    def _scatter_out(self, sub, pos_in_ids):
        for name in ("obs", "agent_id", "action_mask", "rew", "terminated", "flags", "active_nb", "stats"):
            getattr(self.out, name)[pos_in_ids] = getattr(sub, name)

    def reset(self, id=None, **kwargs):
        ids = self._ids(id)
        self._reset_rows(ids, observe=True)
        obs, _rew, _term, info = self._to_host(ids)
        return obs, info

    def step(self, action, id=None):
        ids = self._ids(id)
        act = torch.as_tensor(np.asarray(action).astype(np.int32), device=self.device)
        ids_t = torch.as_tensor(ids, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.mel_env_step(C.byref(self.env), C.byref(self.staging.struct), act.data_ptr(),
                                         ids_t.data_ptr(), len(ids), C.byref(self.out.struct), None, 0,
                                         self._stream()), "mel_env_step")
        obs, rew, term, info = self._to_host(ids)
        return obs, rew, term, np.zeros(len(ids), dtype=bool), info

    def _to_host(self, ids):
        k = len(ids)
        o = self.out
        obs_h = o.obs[:k].cpu().numpy()                # the copy synchronises the stream
        agent = o.agent_id[:k].cpu().numpy()
        mask = o.action_mask[:k].cpu().numpy().astype(bool)
        rew = o.rew[:k].cpu().numpy()
        term = o.terminated[:k].cpu().numpy().astype(bool)
        flags = o.flags[:k].cpu().numpy()
        nb = sets_to_bool(o.active_nb[:k].cpu().numpy().view(np.uint64), self.n)
        stats = o.stats[:k].cpu().numpy()
        obs, info = [], []
        for r in range(k):
            obs.append({"agent_id": str(int(agent[r])), "obs": obs_h[r], "mask": mask[r]})
            d = {"env_id": int(ids[r]), "env_step": int(flags[r, 0]), "environment_step": bool(flags[r, 1]),
                 "explicit_reset": bool(flags[r, 2]),
                 "active_one_hop_neighbors": nb[r]}
            if flags[r, 3]:
                d["logger_stats"] = {key: stats[r, j] for j, key in enumerate(LOGGER_KEYS)}
            info.append(d)
        return np.array(obs, dtype=object), rew, term, np.array(info, dtype=object)

    # ------------------------------------------------------------------ device-resident path
    def load_pool(self, packed: dict, reset_snapshots: bool = False) -> DevicePool:
        """Upload a pre-sampled episode pool (pack_episodes layout) to HBM.  ``reset_snapshots``: also precompute every
        episode's post-reset state for mel_env_round (DevicePool.build_snapshots)."""
        pool = DevicePool(packed, self.n, self.device)
        if reset_snapshots:
            pool.build_snapshots(self.lib, self.env, self._stream())
        return pool

    def reset_device(self, pool: DevicePool, episode_ids: torch.Tensor, out: ObsBuffers | None):
        _lib.check(self.lib.mel_env_reset(C.byref(self.env), C.byref(pool.struct), None, episode_ids.data_ptr(),
                                          self.env_num, 0, C.byref(out.struct) if out is not None else None,
                                          self._stream()), "mel_env_reset")
        self._graph_loaded[:] = True

    def round_device(self, pool: DevicePool, actions: torch.Tensor | None, row_offsets: torch.Tensor | None,
                     live: torch.Tensor, episode_table: torch.Tensor | None, first: bool = False,
                     round_counter: torch.Tensor | None = None, replay=None):
        """One whole env round for every env (mel_env_round).  ``live`` int64 [B] ([B, W] beyond 64 nodes) is read (the active sets the
        actions were computed for) and overwritten with the next round's active sets."""
        _lib.check(self.lib.mel_env_round(
            C.byref(self.env), C.byref(pool.struct), actions.data_ptr() if actions is not None else None,
            row_offsets.data_ptr() if row_offsets is not None else None, live.data_ptr(),
            episode_table.data_ptr() if episode_table is not None else None,
            episode_table.shape[1] if episode_table is not None else 0, int(first),
            round_counter.data_ptr() if round_counter is not None else None,
            C.byref(replay.struct) if replay is not None else None, self._stream()), "mel_env_round")

    def step_device(self, pool: DevicePool, actions: torch.Tensor, out: ObsBuffers | None,
                    episode_table: torch.Tensor | None = None):
        """One AEC step for every env, all on the current stream; ``actions`` int32 [B] on device.
        With ``episode_table`` (int32 [B, K]) finished episodes are re-seeded on device."""
        _lib.check(self.lib.mel_env_step(C.byref(self.env), C.byref(pool.struct), actions.data_ptr(), None,
                                         self.env_num, C.byref(out.struct) if out is not None else None,
                                         episode_table.data_ptr() if episode_table is not None else None,
                                         episode_table.shape[1] if episode_table is not None else 0,
                                         self._stream()), "mel_env_step")
