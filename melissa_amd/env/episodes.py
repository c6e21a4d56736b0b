"""Host-side episode sampling and packing.

``World.reset`` (graph_env/env/utils/core.py:343-437) draws, per episode, an episode seed from the
env's PCG64 generator, then - from ``RandomState(episode_seed)`` - the movement seed, the source node,
the interest density and the interested set; node movement later consumes
``RandomState(movement_seed).uniform(-1, 1)`` N x-draws then N y-draws per world step
(core.py:316-319).  ``EpisodeSampler.sample()`` performs exactly those calls in that order and
``pack_episodes`` lays the result out as the device-resident pool of include/melissa_hip.h
(``mel_episode_pool``): the packed replacement of the reference's per-episode ``pickle.load`` of an
``nx.Graph`` (core.py:450-452).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

RADIUS_OF_INFLUENCE = 0.20
NODES_MOVEMENT_STEP = 0.06      # constants.py:4


# ---- node sets (include/melissa_hip.h, MEL_SET_WORDS): a set of nodes of an N-node graph is W = ceil(N / 64) uint64 words,
# word k = nodes 64 k .. 64 k + 63.  N <= 64: one uint64 per set and arrays of sets have exactly the shape they always had
# ([N], [B], ...); 64 < N <= 128: a trailing [2] axis.
def set_words(n: int) -> int:
    return (int(n) + 63) // 64


def set_shape(n: int) -> tuple:
    """Trailing shape of an array of node sets: () for N <= 64, (W,) beyond."""
    w = set_words(n)
    return () if w == 1 else (w,)


def int_to_set(mask: int, n: int):
    """Python int bit mask -> np.uint64 scalar (N <= 64) or uint64 [W]."""
    w = set_words(n)
    if w == 1:
        return np.uint64(mask & 0xFFFFFFFFFFFFFFFF)
    return np.array([(mask >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(w)], dtype=np.uint64)


def set_to_int(words) -> int:
    """np.uint64 scalar / uint64 [W] (or their int64 bit patterns) -> Python int bit mask."""
    a = np.atleast_1d(np.asarray(words))
    if a.dtype != np.uint64:
        a = a.astype(np.int64).view(np.uint64)
    out = 0
    for k, v in enumerate(a):
        out |= int(v) << (64 * k)
    return out


def sets_to_bool(words: np.ndarray, n: int) -> np.ndarray:
    """uint64 [..., (W)] node sets -> bool [..., n] membership."""
    a = np.asarray(words)
    if a.dtype != np.uint64:
        a = a.astype(np.int64).view(np.uint64)
    w = set_words(n)
    if w == 1:
        a = a[..., None]
    node = np.arange(n)
    return ((a[..., node // 64] >> (node % 64).astype(np.uint64)) & np.uint64(1)).astype(bool)


@dataclass
class Graph:
    """A graph as the env consumes it: float64 node positions + one-hop node sets (uint64 per node; [N, 2] beyond 64 nodes)."""
    pos: np.ndarray              # [N, 2] float64
    one_hop: np.ndarray          # [N] uint64 ([N, W] for N > 64)

    @staticmethod
    def from_positions(pos, radius: float = RADIUS_OF_INFLUENCE) -> "Graph":
        """nx.random_geometric_graph / geometric_edges rule: edge iff dx^2 + dy^2 <= radius^2 (float64)."""
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 2)
        dx = pos[:, None, 0] - pos[None, :, 0]
        dy = pos[:, None, 1] - pos[None, :, 1]
        within = (dx * dx + dy * dy) <= radius ** 2
        np.fill_diagonal(within, False)
        return Graph(pos, adjacency_to_masks(within))

    @staticmethod
    def from_edges(n: int, edges, pos=None) -> "Graph":
        adj = np.zeros((n, n), dtype=bool)
        for u, v in edges:
            adj[u, v] = adj[v, u] = True
        return Graph(np.zeros((n, 2)) if pos is None else np.asarray(pos, dtype=np.float64), adjacency_to_masks(adj))

    @staticmethod
    def from_networkx(g) -> "Graph":
        n = g.number_of_nodes()
        pos = np.array([g.nodes[i]["pos"] for i in range(n)], dtype=np.float64)
        return Graph.from_edges(n, list(g.edges()), pos)

    def is_connected(self) -> bool:
        n = len(self.one_hop)
        seen, frontier = 1, 1
        masks = [set_to_int(m) for m in self.one_hop]
        while frontier:
            nxt = 0
            for i in range(n):
                if (frontier >> i) & 1:
                    nxt |= masks[i]
            frontier = nxt & ~seen
            seen |= nxt
        return seen == (1 << n) - 1


def adjacency_to_masks(adj: np.ndarray) -> np.ndarray:
    """bool [N, N] -> uint64 [N] (N <= 64) or [N, W] node sets (bit j of row i = adj[i, j])."""
    n = adj.shape[0]
    w = set_words(n)
    out = np.zeros((n, w), dtype=np.uint64)
    for k in range(w):
        cols = adj[:, 64 * k: 64 * (k + 1)]
        weights = (np.uint64(1) << np.arange(cols.shape[1], dtype=np.uint64))
        out[:, k] = (cols.astype(np.uint64) * weights[None, :]).sum(axis=1, dtype=np.uint64)
    return out[:, 0] if w == 1 else out


def synthetic_graph_pool(n: int, count: int, first_seed: int = 0, radius: float = RADIUS_OF_INFLUENCE):
    """Connected random geometric graphs in the unit square (recipe of core.py:440-447).
    Positions follow ``nx.random_geometric_graph(n, radius, seed=s)`` when networkx is importable
    (python ``random.Random(s)``: x then y per node), which is what the recipe names."""
    import random
    out, s = [], first_seed
    while len(out) < count:
        rng = random.Random(s)
        pos = np.array([[rng.random(), rng.random()] for _ in range(n)], dtype=np.float64)
        g = Graph.from_positions(pos, radius)
        if g.is_connected():
            out.append(g)
        s += 1
    return out


def packed_graph_pool(n: int, count: int, first_seed: int = 0, radius: float = RADIUS_OF_INFLUENCE):
    """The same accepted graphs as :func:`synthetic_graph_pool` (seeds first_seed, first_seed + 1, ... keeping the
    connected ones) as packed arrays - ``pos`` float64 [G, N, 2], ``one_hop`` uint64 [G, N], ``seeds`` int64 [G].  This is
    the on-disk / in-HBM dataset format that replaces the per-episode ``pickle.load`` of an ``nx.Graph``
    (core.py:165-175,450-452; the reference trains on 50 000-graph pools, README.md:92-93): see :func:`save_graph_pool`.
    About 0.5 ms per accepted graph at N = 50 (one in four candidates is connected)."""
    import random
    pos_out = np.empty((count, n, 2), dtype=np.float64)
    hop_out = np.empty((count, n) + set_shape(n), dtype=np.uint64)
    seeds = np.empty(count, dtype=np.int64)
    got, s = 0, first_seed
    while got < count:
        rng = random.Random(s)
        g = Graph.from_positions(np.array([[rng.random(), rng.random()] for _ in range(n)], dtype=np.float64), radius)
        if g.is_connected():
            pos_out[got], hop_out[got], seeds[got] = g.pos, g.one_hop, s
            got += 1
        s += 1
    return pos_out, hop_out, seeds


def save_graph_pool(path: str, pos: np.ndarray, one_hop: np.ndarray, seeds=None):
    """Packed graph dataset on disk: ``pos`` f64 [G, N, 2] + ``adj`` u64 [G, N] (bit j of adj[g, i] = edge i - j;
    [G, N, 2] words beyond 64 nodes)."""
    np.savez(path, pos=np.ascontiguousarray(pos, dtype=np.float64), adj=np.ascontiguousarray(one_hop, dtype=np.uint64),
             **({"seeds": np.asarray(seeds)} if seeds is not None else {}))


def load_graph_pool(path: str):
    """-> list of :class:`Graph` views over the packed arrays of :func:`save_graph_pool` (no per-graph copies)."""
    with np.load(path, allow_pickle=False) as z:
        pos, adj = z["pos"], z["adj"]
    return [Graph(pos[g], adj[g]) for g in range(pos.shape[0])]


def cached_graph_pool(n: int, count: int, first_seed: int = 0, cache_dir: str | None = None):
    """:func:`packed_graph_pool` through an on-disk cache (``graph_pool_n{n}_{count}_{first_seed}.npz``, default
    directory ``$MEL_GRAPH_CACHE`` or the system temp dir): a 50 000-graph pool takes tens of seconds to draw once and
    a fraction of a second to load afterwards."""
    import os
    import tempfile
    cache_dir = cache_dir or os.environ.get("MEL_GRAPH_CACHE") or os.path.join(tempfile.gettempdir(), "melissa_graph_pools")
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, f"graph_pool_n{n}_{count}_{first_seed}.npz")
    if not os.path.exists(path):
        pos, hop, seeds = packed_graph_pool(n, count, first_seed)
        tmp = f"{path}.tmp.{os.getpid()}.npz"
        save_graph_pool(tmp, pos, hop, seeds)
        os.replace(tmp, path)
    return load_graph_pool(path)


@dataclass
class Episode:
    graph_index: int
    origin: int
    interested: int              # bit mask
    movement_seed: int
    scripted: int = 0            # bit mask (World.scripted_indices, core.py:197-221)


class EpisodeSampler:
    """The RNG protocol of World.reset (core.py:343-395), one instance per env.  Training mode draws the episode
    seed and the graph from the env's generator; testing mode (``is_testing``, core.py:182-187,348-370) walks a
    fixed list of ``num_test_episodes`` seeds derived from ``RandomState(17)`` in strict order, picks the graph
    with the episode's own RNG from the (sorted) test pool and cycles the interest density through 0.1 .. 1.0."""

    def __init__(self, n: int, np_random: np.random.Generator, pool_size: int, fixed_graph: bool,
                 fixed_interest_density=None, is_testing: bool = False, num_test_episodes: int = 10,
                 scripted_agents_ratio: float = 0.0):
        if not (0.0 <= scripted_agents_ratio <= 1.0):                                   # core.py:143-144
            raise ValueError("`scripted_agents_ratio` must be in [0.0, 1.0].")
        self.scripted_agents_ratio = float(scripted_agents_ratio)
        self.n, self.np_random, self.pool_size, self.fixed_graph = n, np_random, pool_size, fixed_graph
        self.fixed_interest_density = fixed_interest_density
        self.is_testing, self.num_test_episodes = bool(is_testing), int(num_test_episodes)
        self.test_episode_index = 0
        self.test_seeds_list = []
        if self.is_testing:
            gen = np.random.RandomState(17)                                             # core.py:184-187
            self.test_seeds_list = [gen.randint(0, 1e9) for _ in range(self.num_test_episodes)]

    def seed(self, seed=None):
        self.np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))   # graph.py:145-146

    def _sample_scripted(self, origin: int) -> int:
        """World._sample_scripted_agents (core.py:197-215): ``round(ratio * N)`` vertices drawn without replacement
        from the ENV's generator; the origin is never scripted unless every agent is (ratio == 1)."""
        n_scripted = int(round(self.scripted_agents_ratio * self.n))
        chosen = self.np_random.choice(self.n, size=n_scripted, replace=False)
        mask = 0
        for i in chosen:
            mask |= 1 << int(i)
        if self.scripted_agents_ratio < 1.0:
            mask &= ~(1 << origin)
        return mask

    def _sample_testing(self) -> Episode:
        n = self.n
        if not self.test_seeds_list:                                                    # core.py:349-350
            raise ValueError("No test seeds have been generated! Check num_test_episodes.")
        episode_seed = self.test_seeds_list[self.test_episode_index]                    # :351
        self.test_episode_index = (self.test_episode_index + 1) % self.num_test_episodes   # :352
        ep_rng = np.random.RandomState(episode_seed)                                    # :353
        graph_index = int(ep_rng.randint(0, self.pool_size))      # :357 ep_rng.choice(test_graphs): same draw
        movement_seed = ep_rng.randint(0, 1e9)                                          # :361
        origin = int(ep_rng.randint(0, n))                                              # :364
        density = [i / 10.0 for i in range(1, 11)][self.test_episode_index % 10]        # :365-366 (index already bumped)
        chosen = ep_rng.choice(n, size=int(density * n), replace=False)                 # :393-394
        scripted = self._sample_scripted(origin)                                        # :395
        mask = 0
        for i in chosen:
            mask |= 1 << int(i)
        return Episode(graph_index, origin, mask, int(movement_seed), scripted)

    def sample(self) -> Episode:
        if self.is_testing:
            return self._sample_testing()
        n = self.n
        episode_seed = self.np_random.integers(0, 1e9)                                  # core.py:372
        ep_rng = np.random.RandomState(episode_seed)                                    # :373
        graph_index = 0
        if not self.fixed_graph:                                                        # :377-379
            graph_index = int(self.np_random.choice(self.pool_size, replace=True))
        movement_seed = ep_rng.randint(0, 1e9)                                          # :381
        origin = int(ep_rng.randint(0, n))                                              # :384
        density = (ep_rng.uniform(0.1, 1.0) if self.fixed_interest_density is None
                   else self.fixed_interest_density)                                    # :385
        chosen = ep_rng.choice(n, size=int(density * n), replace=False)                 # :393-394
        scripted = self._sample_scripted(origin)                                        # :395
        mask = 0
        for i in chosen:
            mask |= 1 << int(i)
        return Episode(graph_index, origin, mask, int(movement_seed), scripted)


def movement_offsets(movement_seed: int, n: int, max_moves: int) -> np.ndarray:
    """[max_moves, 2, N] float64: per world step N x-offsets then N y-offsets, each
    ``0.06 * RandomState(seed).uniform(-1, 1)`` (core.py:316-319)."""
    rs = np.random.RandomState(movement_seed)
    return NODES_MOVEMENT_STEP * rs.uniform(-1, 1, size=(max_moves, 2, n))


def pack_episodes(episodes, graphs, n: int, max_moves: int, dynamic: bool):
    """-> dict of numpy arrays in mel_episode_pool layout."""
    e = len(episodes)
    pos = np.zeros((e, n, 2), dtype=np.float64)
    one_hop = np.zeros((e, n) + set_shape(n), dtype=np.uint64)
    interested = np.zeros((e,) + set_shape(n), dtype=np.uint64)
    scripted = np.zeros((e,) + set_shape(n), dtype=np.uint64)
    origin = np.zeros(e, dtype=np.int32)
    moves = np.zeros((e, max_moves if dynamic else 1, 2, n), dtype=np.float64)
    for k, ep in enumerate(episodes):
        g = graphs[ep.graph_index]
        pos[k], one_hop[k] = g.pos, g.one_hop
        interested[k], origin[k], scripted[k] = int_to_set(ep.interested, n), ep.origin, int_to_set(ep.scripted, n)
        if dynamic:
            moves[k] = movement_offsets(ep.movement_seed, n, max_moves)
    return dict(pos=pos, one_hop=one_hop, interested=interested, origin=origin, moves=moves, scripted=scripted)
