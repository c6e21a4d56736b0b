from .episodes import (EpisodeSampler, Graph, cached_graph_pool, load_graph_pool, pack_episodes, packed_graph_pool,
                       save_graph_pool, synthetic_graph_pool)
from .vector_env import HipGraphVectorEnv

__all__ = ["HipGraphVectorEnv", "Graph", "EpisodeSampler", "pack_episodes", "synthetic_graph_pool", "packed_graph_pool",
           "save_graph_pool", "load_graph_pool", "cached_graph_pool"]
