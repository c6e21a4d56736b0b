"""melissa_amd - MI355X-native hot path of Melissa (L-DGN / HL-DGN forward + batched graph env).

Drop-in surface (names follow the reference):
    melissa_amd.networks.LDGNNetwork / HLDGNNetwork   <- graph_env/env/utils/networks/{l_dgn,hl_dgn}.py
    melissa_amd.env.HipGraphVectorEnv                 <- tianshou vector env over graph_env.env.graph.GraphEnv
    melissa_amd.policy.DQNPolicy / MultiAgentSharedPolicy
    melissa_amd.collect.DecisionLoop                  <- the collector hot loop, device resident
The arithmetic lives in hand-written HIP (melissa_amd/csrc) behind the C ABI of include/melissa_hip.h.
"""
__version__ = "0.1.0"
