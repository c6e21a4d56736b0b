from .episodes import EpisodeSampler, Graph, pack_episodes, synthetic_graph_pool
from .vector_env import HipGraphVectorEnv

__all__ = ["HipGraphVectorEnv", "Graph", "EpisodeSampler", "pack_episodes", "synthetic_graph_pool"]
