"""CPU restatement of the reference environment half (TEST INFRASTRUCTURE - the oracle).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``melissa_amd``) never does.

What is restated, step for step, without networkx / pettingzoo / gymnasium / tianshou:

* ``GraphEnv``        graph_env/env/graph.py:18-463      -> :class:`OracleGraphEnv`
* ``World`` / ``Agent`` / ``State``  graph_env/env/utils/core.py:11-437 -> fields of OracleGraphEnv
* ``CustomSelector``  graph_env/env/utils/selector.py:1-52 -> ``sel_*`` fields
* [3P] pettingzoo ``AECEnv`` helpers used by graph.py:325-326,359 (``_accumulate_rewards``,
  ``_clear_rewards``, ``_deads_step_first``, ``last``)
* [3P] tianshou 1.0.0 ``PettingZooEnv.reset/step`` -> :class:`OraclePettingZooEnv`
* [3P] tianshou ``DummyVectorEnv.reset/step`` with env ids   -> :class:`OracleVectorEnv`

Pinning: ``tests/golden/make_env_golden.py`` runs the REAL reference ``GraphEnv`` (imported from
``/root/reference`` through the stand-in base modules of ``tests/golden/ref_standins.py``) on seeded
graphs / action tapes and stores every per-step output in ``tests/golden/env_trace_*.npz``;
``tests/test_env_oracle.py`` replays the same tapes through this restatement and demands equality
(bit-exact for masks/ints/obs, exact float64 for rewards).  The reference's own known-answer tests
(tests/unit/graph_env/env/utils/test_core.py:97-169) are replayed in the same test file.  The [3P]
sliver (pettingzoo/tianshou wrappers, versions unpinned/absent) is "parity unpinned".

Scripted agents (``scripted_agents_ratio`` > 0) run the deterministic heuristics of heuristics/core.py; with the
default ratio 0 ``dm_flag`` is 1 for every node and ``World.step``'s scripted branches (core.py:226-243,264-266)
are empty loops.

Representation: node sets are Python ints used as N-bit masks (bit i = node i); this is exactly the
layout the HIP kernels use (one uint64 per set for N <= 64, two 64-bit words - low word first - for the reference's
100-node size).
"""
from __future__ import annotations

import numpy as np

RADIUS_OF_INFLUENCE = 0.20      # constants.py:1
NUMBER_OF_FEATURES = 5          # constants.py:2
NODES_MOVEMENT_STEP = 0.06      # constants.py:4
MAX_AGENT_STEPS = 4             # graph.py:332, selector.py:44

NONE = -1                       # "None" action / "False" agent selection
SKIP_NONE = -2                  # _skip_agent_selection is None

LOGGER_KEYS = (                 # graph.py:167-177, in dict order
    "total_messages_transmitted", "coverage", "messages_sent", "messages_received",
    "n_neighbours", "interested_agents", "coverage_interested_fraction",
    "coverage_interested_count", "uninterested_with_message", "episode_rewards_sum",
)


def popcount(x: int) -> int:
    return bin(x).count("1")


def bits(x: int):
    i = 0
    while x:
        if x & 1:
            yield i
        x >>= 1
        i += 1


def geometric_adjacency(pos: np.ndarray, radius: float = RADIUS_OF_INFLUENCE) -> list:
    """networkx ``geometric_edges(G, radius)`` (core.py:311) for p=2: edge iff dx^2+dy^2 <= r^2 in
    float64 (both the scipy-KDTree path and the pure-python fallback compare the sum of squares with
    ``radius**2``).  Returns one bitmask per node."""
    n = pos.shape[0]
    r2 = radius ** 2
    dx = pos[:, None, 0] - pos[None, :, 0]
    dy = pos[:, None, 1] - pos[None, :, 1]
    within = (dx * dx + dy * dy) <= r2          # elementwise float64: identical to the scalar expression
    np.fill_diagonal(within, False)
    weights = [1 << j for j in range(n)]
    return [sum(w for w, hit in zip(weights, row) if hit) for row in within.tolist()]


def two_hop_masks(adj: list) -> list:
    """core.py:334-341: one-hop OR neighbours' one-hop, minus self."""
    out = []
    for i, a in enumerate(adj):
        m = a
        for j in bits(a):
            m |= adj[j]
        out.append(m & ~(1 << i))
    return out


class GraphSpec:
    """A graph as the env consumes it: float64 positions + one-hop bitmasks (the packed form of a
    pickled ``nx.Graph`` with ``pos`` node attributes, core.py:450-452)."""

    def __init__(self, pos, adj=None):
        self.pos = np.array(pos, dtype=np.float64).reshape(-1, 2)
        self.adj = list(adj) if adj is not None else geometric_adjacency(self.pos)

    def copy(self):
        return GraphSpec(self.pos.copy(), list(self.adj))

    @staticmethod
    def from_edges(n, edges, pos=None):
        adj = [0] * n
        for u, v in edges:
            adj[u] |= 1 << v
            adj[v] |= 1 << u
        return GraphSpec(np.zeros((n, 2)) if pos is None else pos, adj)


class OracleGraphEnv:
    """``GraphEnv`` + ``World`` + ``CustomSelector`` (training mode, no scripted agents).

    Constructor mirrors graph.py:25-41 for the arguments on the hot path.  ``graph`` fixes the graph
    (core.py:130 ``is_graph_fixed``; note the reference then mutates it in place across episodes when
    ``dynamic_graph`` is on - reproduced here); ``graph_pool`` stands for the
    ``graph_topologies/training_N/*`` files (core.py:175,377-378).
    """

    def __init__(self, number_of_agents, graph: GraphSpec | None = None, graph_pool=None,
                 radius=RADIUS_OF_INFLUENCE, local_ratio=None, dynamic_graph=False,
                 np_random: np.random.Generator | None = None, fixed_interest_density=None,
                 is_testing=False, num_test_episodes=10, scripted_agents_ratio=0.0, heuristic=None):
        self.n = int(number_of_agents)
        # core.py:143-163: scripted agents run one of the heuristics of heuristics/core.py (the deterministic ones;
        # the probabilistic ones draw from the process-global np.random and cannot be pinned)
        if not (0.0 <= scripted_agents_ratio <= 1.0):
            raise ValueError("`scripted_agents_ratio` must be in [0.0, 1.0].")
        elif scripted_agents_ratio == 0.0 and heuristic is not None:
            raise ValueError("If `scripted_agents_ratio` is 0.0, no heuristic can be set.")
        if heuristic not in (None, "simple_broadcast", "broadcast_if_any_interested", "silent"):
            raise ValueError(f"Unknown heuristic policy: {heuristic}")
        self.scripted_agents_ratio, self.heuristic = scripted_agents_ratio, heuristic
        # testing mode (core.py:178-187): fixed list of per-episode seeds, walked in strict order
        self.is_testing, self.num_test_episodes = bool(is_testing), int(num_test_episodes)
        self.test_episode_index = 0
        self.test_seeds_list = []
        if self.is_testing:
            testing_generator = np.random.RandomState(17)
            self.test_seeds_list = [testing_generator.randint(0, 1e9) for _ in range(self.num_test_episodes)]
        assert 1 <= self.n <= 128
        self.full = (1 << self.n) - 1
        self.radius = radius
        self.local_ratio = local_ratio
        self.dynamic_graph = dynamic_graph
        self.fixed_interest_density = fixed_interest_density
        self.is_graph_fixed = graph is not None
        self.graph = graph.copy() if graph is not None else None
        self.graph_pool = graph_pool
        self.selected_graph = None
        # graph.py:44 self.seed() -> gymnasium seeding.np_random(None); tests pass a seeded Generator
        self.np_random = np_random if np_random is not None else np.random.default_rng()
        self.obs_matrix = np.zeros((self.n, 2 + NUMBER_OF_FEATURES + 1), dtype=np.float32)
        self.is_new_round = None
        self.movement_np_random = None
        # World.__init__ ends with self.reset() (core.py:190), then GraphEnv.__init__ calls reset()
        # again (graph.py:118): two episode samplings per construction.
        self._world_reset()
        self.reset()

    # ------------------------------------------------------------------ seeding (graph.py:145-146)
    def seed(self, seed=None):
        ss = np.random.SeedSequence(seed)
        self.np_random = np.random.Generator(np.random.PCG64(ss))

    # ------------------------------------------------------------------ World.reset core.py:343-437
    def _world_reset(self):
        n = self.n
        if self.is_testing:                                                  # core.py:348-370
            if not self.test_seeds_list:
                raise ValueError("No test seeds have been generated! Check num_test_episodes.")
            episode_seed = self.test_seeds_list[self.test_episode_index]
            self.test_episode_index = (self.test_episode_index + 1) % self.num_test_episodes
            ep_rng = np.random.RandomState(episode_seed)
            # :357 ep_rng.choice(self.test_graphs) over the sorted file list = one randint over the pool
            self.selected_graph = ep_rng.choice(len(self.graph_pool))
            self.graph = self.graph_pool[int(self.selected_graph)].copy()
            movement_seed = ep_rng.randint(0, 1e9)                           # :361
            self.movement_np_random = np.random.RandomState(movement_seed)
            chosen_source_id = ep_rng.randint(0, n)                          # :364
            fixed_interest_densities = [i / 10.0 for i in range(1, 11)]     # :365-366
            interest_density = fixed_interest_densities[self.test_episode_index % len(fixed_interest_densities)]
        else:
            episode_seed = self.np_random.integers(0, 1e9)                   # core.py:372
            ep_rng = np.random.RandomState(episode_seed)                     # :373
            if not self.is_graph_fixed:                                      # :377-379
                self.selected_graph = self.np_random.choice(len(self.graph_pool), replace=True)
                self.graph = self.graph_pool[int(self.selected_graph)].copy()
            movement_seed = ep_rng.randint(0, 1e9)                           # :381
            self.movement_np_random = np.random.RandomState(movement_seed)   # :382
            chosen_source_id = ep_rng.randint(0, n)                          # :384
            interest_density = (ep_rng.uniform(0.1, 1.0) if self.fixed_interest_density is None
                                else self.fixed_interest_density)            # :385
        self.messages_transmitted = 0                                        # :389
        self.origin_agent = int(chosen_source_id)                            # :390
        num_interested = int(interest_density * n)                           # :393
        interested_indices = ep_rng.choice(n, size=num_interested, replace=False)  # :394
        # :395 _apply_scripted_mask -> _sample_scripted_agents (core.py:197-215), from the ENV's generator
        n_scripted = int(round(self.scripted_agents_ratio * n))
        scripted_draw = self.np_random.choice(n, size=n_scripted, replace=False)
        self.interested = 0
        for i in interested_indices:
            self.interested |= 1 << int(i)
        self.scripted = 0
        for i in scripted_draw:
            self.scripted |= 1 << int(i)
        if self.scripted_agents_ratio < 1.0:                                  # :212-214 origin is never scripted
            self.scripted &= ~(1 << self.origin_agent)
        # Agent(...) / state.reset / agent.reset (core.py:398-425)
        self.pos = self.graph.pos            # shared with the graph object (mutated in place)
        self.adj = list(self.graph.adj)                                       # update_one_hop :321-332
        self.two_hop = two_hop_masks(self.adj)                                # :334-341
        self.has_message = 0
        self.message_origin = 0
        self.has_taken_action = 0
        self.received_count = [0] * n       # sum(received_from) per agent (State.received_from :22)
        self.agent_msgs = [0] * n           # Agent.messages_transmitted
        self.agent_action = [NONE] * n      # Agent.action (None)
        self.steps_taken = [0] * n          # :424
        self.truncated = 0                  # Agent.truncated = False :425
        self.two_hop_cover = [0] * n
        self.gained_two_hop_cover = [0] * n
        # update_one_hop_neighbors_info counts them (:401,:326-327) but agent.reset() re-runs Agent.__init__ right
        # after (:412-416 -> core.py:68), so the count is 0 until the first move_graph of the episode recomputes it
        self.number_interested_neighbors = [0] * n
        # source (core.py:432-435)
        s = self.origin_agent
        self.message_origin |= 1 << s
        self.has_message |= 1 << s
        self.steps_taken[s] = 1
        self._world_step()                                                    # :437

    # ------------------------------------------------------------------ World.step core.py:225-266
    def _world_step(self):
        n = self.n
        s = self.origin_agent
        callback = self.heuristic is not None                                 # Agent.action_callback, core.py:428-429
        if callback:
            for i in bits(self.scripted):                                     # :226-234 (no relay masks, so :236-243 is idle)
                if self.heuristic == "simple_broadcast":                      # heuristics/core.py:13-18
                    self.agent_action[i] = 0 if (self.has_taken_action >> i) & 1 else 1
                elif self.heuristic == "broadcast_if_any_interested":         # :45-53
                    self.agent_action[i] = 1 if self.number_interested_neighbors[i] > 0 else 0
                else:                                                         # silent :56-62
                    self.agent_action[i] = 0
        if self.agent_msgs[s] == 0:                                           # :246
            self.agent_action[s] = 1
        for i in range(n):                                                    # :249-254 (id order)
            if self.agent_action[i] not in (NONE, 0) and (self.has_message >> i) & 1:
                self._relay_message(i)
        if self.dynamic_graph:                                                # :256-257
            self._move_graph()
        for i in range(n):                                                    # :260-261 -> :94-102
            cover = popcount(self.two_hop[i] & (self.has_message | self.message_origin))
            self.gained_two_hop_cover[i] = cover - self.two_hop_cover[i]
            self.two_hop_cover[i] = cover
        if callback:
            for i in bits(self.scripted):                                     # :264-266
                self.agent_action[i] = 0

    def _relay_message(self, i):                                              # core.py:268-279
        self.messages_transmitted += 1
        self.agent_msgs[i] += 1
        self.has_taken_action |= 1 << i
        for j in bits(self.adj[i]):
            self.received_count[j] += 1
        self.has_message |= self.adj[i]

    def _move_graph(self):                                                    # core.py:281-319
        n = self.n
        step = NODES_MOVEMENT_STEP
        ox = [step * self.movement_np_random.uniform(-1, 1) for _ in range(n)]   # :317
        oy = [step * self.movement_np_random.uniform(-1, 1) for _ in range(n)]   # :318
        for k in range(n):                                                    # :305-306
            self.pos[k, 0] = self.pos[k, 0] + ox[k]
            self.pos[k, 1] = self.pos[k, 1] + oy[k]
        self.adj = geometric_adjacency(self.pos, RADIUS_OF_INFLUENCE)         # :311-314
        self.graph.adj = list(self.adj)
        self.number_interested_neighbors = [popcount(a & self.interested) for a in self.adj]   # :286-287 -> :321-327
        self.two_hop = two_hop_masks(self.adj)

    # ------------------------------------------------------------------ GraphEnv.reset graph.py:222-248
    def reset(self, seed=None):
        n = self.n
        if seed is not None:
            self.seed(seed)
        # selector.reinit (selector.py:9-16)
        self.sel_steps = [0] * n
        self.sel_active = 0
        self.sel_selected = 0
        self.rewards = [0.0] * n
        self.cum_rewards = [0.0] * n
        self.alive = self.full              # keys present in terminations/truncations/rewards/infos
        self.terminated = 0
        self.infos = [dict() for _ in range(n)]
        self.num_moves = 0
        self._world_reset()
        self.episode_rewards_sum = 0.0
        self._update_obs_matrix()
        self.agents = self.has_message & (self.full if self.is_testing else ~self.scripted)   # :242-245
        # selector.enable(on_reset=True) (selector.py:39-44)
        self.sel_steps[self.origin_agent] += 1
        self._selector_enable(self.agents)
        self.agent_selection = self._selector_next()                          # :247
        self.skip_selection = getattr(self, "skip_selection", SKIP_NONE)      # attribute survives resets
        self.current_actions = [NONE] * n                                     # :248

    # ------------------------------------------------------------------ selector.py
    def _selector_enable(self, agents_mask):
        for i in bits(agents_mask):
            if self.sel_steps[i] < MAX_AGENT_STEPS:
                self.sel_active |= 1 << i
            else:
                self.sel_active &= ~(1 << i)

    def _selector_next(self):
        cand = self.sel_active & ~self.sel_selected
        if cand:
            i = (cand & -cand).bit_length() - 1
            self.sel_steps[i] += 1
            self.sel_selected |= 1 << i
            return i
        return NONE

    # ------------------------------------------------------------------ obs matrix graph.py:250-271
    def _update_obs_matrix(self):
        m = self.obs_matrix
        for i in range(self.n):
            m[i, 0] = self.pos[i, 0]
            m[i, 1] = self.pos[i, 1]
            m[i, 2] = popcount(self.adj[i])
            m[i, 3] = self.agent_msgs[i]
            a = self.agent_action[i]
            m[i, 4] = a if a != NONE else 0
            m[i, 5] = 1.0 if (self.interested >> i) & 1 else 0.0
            m[i, 6] = 1.0 if ((self.has_message | self.message_origin) >> i) & 1 else 0.0
            m[i, 7] = 0.0 if (self.scripted >> i) & 1 else 1.0

    # ------------------------------------------------------------------ get_info graph.py:149-179
    def get_info(self):
        n = self.n
        num_interested = popcount(self.interested)
        cov_int = popcount(self.has_message & self.interested)
        return {"logger_stats": {
            "total_messages_transmitted": self.messages_transmitted,
            "coverage": popcount(self.has_message) / n,
            "messages_sent": sum(self.agent_msgs),
            "messages_received": float(sum(self.received_count)),
            "n_neighbours": float(sum(popcount(a) for a in self.adj)),
            "interested_agents": num_interested,
            "coverage_interested_fraction": (cov_int / num_interested if num_interested > 0 else 0.0),
            "coverage_interested_count": cov_int,
            "uninterested_with_message": popcount(self.has_message & ~self.interested & self.full),
            "episode_rewards_sum": self.episode_rewards_sum,
        }}

    # ------------------------------------------------------------------ observe graph.py:181-216
    def observe(self, agent):
        n = self.n
        obs = np.concatenate([self.obs_matrix.reshape(-1), [agent]]).astype(np.float32)
        dead = (self.terminated >> agent) & 1
        action_mask = np.array([0, 0] if dead else [1, 1], dtype=np.int8)
        info = self.infos[agent]
        info["env_step"] = self.num_moves
        info["environment_step"] = False
        info["explicit_reset"] = False
        # :198-203 neighbours that are truncated AND no longer listed in self.agents are masked out
        nb = self.adj[agent] & ~(self.truncated & ~self.agents)
        info["active_one_hop_neighbors"] = np.array([(nb >> i) & 1 for i in range(n)], dtype=np.bool_)
        if popcount(self.agents) == 1 and (self.agents & ~self.terminated) == 0:   # :205-207
            self.is_new_round = False
            info["explicit_reset"] = True
        if self.is_new_round:                                                 # :209-211
            info["environment_step"] = True
            self.is_new_round = False
        return {"observation": obs, "action_mask": action_mask}

    def last(self):
        """[3P] AECEnv.last: (obs, cumulative reward, terminated, truncated, info)."""
        a = self.agent_selection
        obs = self.observe(a)
        return obs, self.cum_rewards[a], bool((self.terminated >> a) & 1), False, self.infos[a]

    # ------------------------------------------------------------------ _was_dead_step graph.py:274-301
    def _was_dead_step(self):
        a = self.agent_selection
        assert (self.terminated >> a) & 1, "an agent that was not dead attempted to be removed"
        self.alive &= ~(1 << a)
        self.terminated &= ~(1 << a)
        self.infos[a] = None
        self.agents &= ~(1 << a)
        dead = self.agents & self.terminated
        if dead:
            if self.skip_selection == SKIP_NONE:
                self.skip_selection = self.agent_selection
            self.agent_selection = (dead & -dead).bit_length() - 1
        else:
            if self.skip_selection != SKIP_NONE:
                self.agent_selection = self.skip_selection
            self.skip_selection = SKIP_NONE

    # ------------------------------------------------------------------ step graph.py:303-359
    def step(self, action):
        a = self.agent_selection
        if (self.terminated >> a) & 1:                                        # :304-310
            self.sel_active &= ~(1 << a)
            self._was_dead_step()
            return
        self.current_actions[a] = action                                      # :314
        self.steps_taken[a] += 1                                              # :316-318
        self.cum_rewards[a] = 0                                               # :320
        self.agent_selection = self._selector_next()                          # :321
        if self.agent_selection == NONE:                                      # :324
            for i in bits(self.alive):                                        # _accumulate_rewards
                self.cum_rewards[i] += self.rewards[i]
            for i in bits(self.alive):                                        # _clear_rewards
                self.rewards[i] = 0
            self._execute_world_step()
            self.num_moves += 1
            for i in bits(self.agents):                                       # :330-334
                if self.steps_taken[i] >= MAX_AGENT_STEPS and not (self.truncated >> i) & 1:
                    self.truncated |= 1 << i
                    self.terminated |= 1 << i
            self.agents = self.has_message & self.alive & (self.full if self.is_testing else ~self.scripted)   # :336-341
            self._selector_enable(self.agents)                                # :342
            self.sel_selected = 0                                             # :343
            self.is_new_round = True                                          # :344
            self.agent_selection = self._selector_next()                      # :345
            self.current_actions = [NONE] * self.n                            # :347
        if self.agent_selection != NONE:                                      # :358
            self.infos[self.agent_selection] = self.get_info()
        # _deads_step_first (graph.py:359, [3P])
        dead = self.agents & self.terminated
        if dead:
            self.skip_selection = self.agent_selection
            self.agent_selection = (dead & -dead).bit_length() - 1

    # ------------------------------------------------------------------ graph.py:361-389
    def _execute_world_step(self):
        for i in range(self.n):                                               # :362-365
            self.agent_action[i] = self.current_actions[i]
        self._world_step()
        self._update_obs_matrix()                                             # :370-371
        for i in bits(self.agents):                                           # :378 (id order)
            r = float(self.reward(i))
            if self.local_ratio is not None:                                  # :380-384 (global reward = 0.0)
                r = 0.0 * (1 - self.local_ratio) + r * self.local_ratio
            self.rewards[i] = r
            self.episode_rewards_sum += r

    # ------------------------------------------------------------------ reward graph.py:402-463
    def reward(self, i):
        one_hop = self.adj[i]
        two_hop = self.two_hop[i]
        covered = self.has_message | self.message_origin
        total_int_2hop = popcount(two_hop & self.interested)
        cov_int_2hop = popcount(two_hop & self.interested & covered)
        reward = (cov_int_2hop / total_int_2hop) if total_int_2hop > 0 else 0.0
        deg = popcount(one_hop)
        if self.agent_action[i] not in (NONE, 0):                             # :430 transmits
            if deg > 0:
                pen_unint = popcount(one_hop & ~self.interested) / deg
            else:
                pen_unint = 0
            pen_cov = popcount(one_hop & self.has_message) / deg if deg else 0
            reward -= (pen_unint + pen_cov)
        else:                                                                 # :452-461
            one_int = popcount(one_hop & self.interested)
            unc = popcount(one_hop & self.interested & ~self.has_message & ~self.message_origin)
            if unc > 0:
                reward -= unc / one_int
        return reward


class OraclePettingZooEnv:
    """[3P] tianshou 1.0.0 ``PettingZooEnv``: sticky length-N reward vector, obs dict
    ``{agent_id, obs, mask}``."""

    def __init__(self, env: OracleGraphEnv):
        self.env = env
        self.n = env.n
        self.rewards = [0] * env.n
        self.reset()

    def _pack(self):
        observation, _rew, term, trunc, info = self.env.last()
        obs = {"agent_id": str(self.env.agent_selection),
               "obs": observation["observation"],
               "mask": [bool(m == 1) for m in observation["action_mask"]]}
        return obs, term, trunc, info

    def reset(self, seed=None):
        self.env.reset(seed=seed)
        obs, _t, _tr, info = self._pack()
        return obs, info

    def step(self, action):
        self.env.step(action)
        obs, term, trunc, info = self._pack()
        for i in bits(self.env.alive):
            self.rewards[i] = self.env.rewards[i]
        return obs, list(self.rewards), term, trunc, info


class OracleVectorEnv:
    """[3P] tianshou ``DummyVectorEnv`` reduced to what the collectors use
    (multi_agent_collector.py:119,192-195,296): ``__len__``, ``is_async``, ``reset(ids)``,
    ``step(actions, ids)`` returning stacked numpy, ``info['env_id']``."""

    is_async = False

    def __init__(self, envs):
        self.workers = [OraclePettingZooEnv(e) for e in envs]
        self.env_num = len(envs)

    def __len__(self):
        return self.env_num

    def _ids(self, ids):
        return list(range(self.env_num)) if ids is None else [int(i) for i in np.atleast_1d(ids)]

    def reset(self, ids=None):
        ids = self._ids(ids)
        out = [self.workers[i].reset() for i in ids]
        infos = []
        for i, (_, info) in zip(ids, out):
            info = dict(info)
            info["env_id"] = i
            infos.append(info)
        return np.array([o for o, _ in out], dtype=object), np.array(infos, dtype=object)

    def step(self, actions, ids=None):
        ids = self._ids(ids)
        res = [self.workers[i].step(int(a)) for i, a in zip(ids, actions)]
        obs = np.array([r[0] for r in res], dtype=object)
        rew = np.array([r[1] for r in res], dtype=np.float64)
        term = np.array([r[2] for r in res], dtype=bool)
        trunc = np.array([r[3] for r in res], dtype=bool)
        infos = []
        for i, r in zip(ids, res):
            info = dict(r[4])
            info["env_id"] = i
            infos.append(info)
        return obs, rew, term, trunc, np.array(infos, dtype=object)
