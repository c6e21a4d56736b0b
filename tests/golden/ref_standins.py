"""Stand-in base modules so the *real* reference ``graph_env.env.graph`` can be imported
in the build container (gymnasium / pettingzoo are not installed there).

TEST INFRASTRUCTURE ONLY - used by the golden-vector generators in this directory, which
run in the build container where ``/root/reference`` exists.  Nothing here travels into
the product path, and nothing here is reference source: these ~60 lines restate the
third-party base-class behaviour that ``GraphEnv`` relies on:

* ``gymnasium.spaces.{Dict,Box,Discrete}``   - shape holders only (graph.py:88-111)
* ``gymnasium.utils.seeding.np_random``      - PCG64 Generator from a SeedSequence (graph.py:146)
* ``pettingzoo.AECEnv``                      - ``_accumulate_rewards`` / ``_clear_rewards`` /
  ``_deads_step_first`` / ``last`` (graph.py:325-326,359) restated from PettingZoo 1.24
* ``pettingzoo.utils.wrappers``              - identity wrappers (graph.py:487-493)

PettingZoo's version is not pinned by the reference (requirements.txt:12-13), so this sliver
is "parity unpinned"; everything else in the env half is the reference's own code.
"""
import sys
import types

import numpy as np

DEFAULT_SEED = None


def install():
    if "gymnasium" in sys.modules and "pettingzoo" in sys.modules:
        return
    gym = types.ModuleType("gymnasium")
    spaces = types.ModuleType("gymnasium.spaces")
    utils = types.ModuleType("gymnasium.utils")
    seeding = types.ModuleType("gymnasium.utils.seeding")
    logger = types.ModuleType("gymnasium.logger")

    class _Space:
        def __init__(self, *a, **k):
            self.args, self.kwargs = a, k
            self.shape = k.get("shape")
            self.n = a[0] if a and isinstance(a[0], int) else None

    class Dict(_Space):
        pass

    class Box(_Space):
        pass

    class Discrete(_Space):
        pass

    spaces.Dict, spaces.Box, spaces.Discrete = Dict, Box, Discrete

    def np_random(seed=None):
        # real gymnasium draws OS entropy for seed=None; the golden generators need the two
        # constructor-time episode samplings (core.py:190, graph.py:118) to be reproducible, so
        # None maps to the module-level DEFAULT_SEED (set by the generator before building an env)
        if seed is None:
            seed = sys.modules[__name__].DEFAULT_SEED
        ss = np.random.SeedSequence(seed)
        return np.random.Generator(np.random.PCG64(ss)), ss.entropy

    seeding.np_random = np_random
    logger.warn = lambda *a, **k: None
    gym.spaces, gym.utils, gym.logger = spaces, utils, logger
    utils.seeding = seeding

    pz = types.ModuleType("pettingzoo")
    pz_utils = types.ModuleType("pettingzoo.utils")
    wrappers = types.ModuleType("pettingzoo.utils.wrappers")

    class AECEnv:
        def __init__(self):
            pass

        def _deads_step_first(self):
            dead = [a for a in self.agents if (self.terminations[a] or self.truncations[a])]
            if dead:
                self._skip_agent_selection = self.agent_selection
                self.agent_selection = dead[0]
            return self.agent_selection

        def _clear_rewards(self):
            for a in self.rewards:
                self.rewards[a] = 0

        def _accumulate_rewards(self):
            for a, r in self.rewards.items():
                self._cumulative_rewards[a] += r

        def last(self, observe=True):
            a = self.agent_selection
            obs = self.observe(a) if observe else None
            return (obs, self._cumulative_rewards[a], self.terminations[a],
                    self.truncations[a], self.infos[a])

    wrappers.AssertOutOfBoundsWrapper = lambda e: e
    wrappers.OrderEnforcingWrapper = lambda e: e
    pz.AECEnv, pz.utils = AECEnv, pz_utils
    pz_utils.wrappers = wrappers

    # matplotlib is only used by draw_graph (rendering, out of scope); avoid importing a backend
    if "matplotlib" not in sys.modules:
        try:
            import matplotlib  # noqa: F401
            matplotlib.use("Agg")
        except Exception:
            mpl = types.ModuleType("matplotlib")
            plt = types.ModuleType("matplotlib.pyplot")
            mpl.pyplot = plt
            sys.modules["matplotlib"] = mpl
            sys.modules["matplotlib.pyplot"] = plt

    for name, mod in {
        "gymnasium": gym, "gymnasium.spaces": spaces, "gymnasium.utils": utils,
        "gymnasium.utils.seeding": seeding, "gymnasium.logger": logger,
        "pettingzoo": pz, "pettingzoo.utils": pz_utils, "pettingzoo.utils.wrappers": wrappers,
    }.items():
        sys.modules[name] = mod


def import_reference(ref_root="/root/reference"):
    """Import the real reference env modules (build container only)."""
    install()
    sys.dont_write_bytecode = True
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)
    import graph_env.env.graph as ref_graph
    import graph_env.env.utils.core as ref_core
    return ref_graph, ref_core
