"""GPU parity of the continuous episode supply (csrc/episode_stream.hpp, melissa_amd/env/stream.py): the device sampler
draws, bit for bit, what World.reset draws (core.py:343-395) - checked against ``EpisodeSampler``, the numpy
restatement of that protocol which the golden env traces pin to the real reference - for episodes far beyond the first
ring, and the loops that consume it walk exactly the oracle env's trajectory through many resets."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.trace_replay import set_int      # node sets (uint64 scalar, or words beyond 64 nodes) -> int
DUEL = lambda: ({"hidden_sizes": [128, 128]}, {"hidden_sizes": [128, 128]})


def _expected(venv, seed, count):
    from melissa_amd.env.episodes import movement_offsets
    out = []
    for b in range(venv.env_num):
        sampler = venv.make_sampler(seed + b)
        out.append([sampler.sample() for _ in range(count)])
    return out, movement_offsets


@pytest.mark.parametrize("n,n_graphs,dynamic,density,fixed", [(20, 5, True, None, False), (50, 64, True, None, False),
                                                               (64, 3, True, None, False), (12, 7, False, 0.35, False),
                                                               (20, 1, False, None, True), (50, 50000, True, None, False),
                                                               (100, 4, True, None, False), (70, 3, False, 0.6, False)])
def test_device_sampler_matches_numpy_protocol(n, n_graphs, dynamic, density, fixed):
    from melissa_amd import _lib as L
    from melissa_amd.env import Graph, HipGraphVectorEnv, synthetic_graph_pool
    from melissa_amd.env.stream import EpisodeStream
    B, K, seed, max_moves, discard = 9, 5, 123, 7, 2
    base = synthetic_graph_pool(n, min(n_graphs, 6), first_seed=3)
    if n_graphs > len(base):                   # a big pool (the graph draw's range matters, not the graphs): repeat
        graphs = [base[g % len(base)] for g in range(n_graphs)]
    else:
        graphs = base
    kw = dict(graph=graphs[0]) if fixed else dict(graph_pool=graphs)
    venv = HipGraphVectorEnv(B, n, dynamic_graph=dynamic, device="cuda", max_moves=max_moves, fixed_interest_density=density,
                             construct_like_reference=False, **kw)
    total = 4 * K + 3
    want, movement_offsets = _expected(venv, seed, total + discard)
    st = EpisodeStream(venv, seed, ring=K, discard=discard)
    cursor = venv.scalars()[:, L.S_EP_CURSOR]
    checked = 0
    for cur in range(0, total - K + 2):
        cursor.fill_(cur)                      # pretend every env has started `cur` episodes
        st.refill()
        torch.cuda.synchronize()
        produced = st.produced.cpu().numpy()
        assert (produced == cur + K - 1).all()
        t = {k: v.cpu().numpy() for k, v in st.pool.tensors.items()}
        for b in range(B):
            for j in range(max(0, cur - 1), cur + K - 1):          # every live slot of the ring
                slot = b * K + j % K
                ep = want[b][j + discard]
                g = graphs[ep.graph_index]
                assert t["origin"][slot] == ep.origin and set_int(t["interested"][slot]) == ep.interested, (b, j)
                np.testing.assert_array_equal(t["pos"][slot], g.pos)
                np.testing.assert_array_equal(t["one_hop"][slot].view(np.uint64), g.one_hop)
                if dynamic:
                    np.testing.assert_array_equal(t["moves"][slot], movement_offsets(ep.movement_seed, n, max_moves))
                checked += 1
    assert checked > 100
    # the generators themselves: numpy's PCG64 state after the same number of samplings
    pcg = st.pcg.cpu().numpy().view(np.uint64)
    half = st.pcg_half.cpu().numpy().view(np.uint32)
    for b in range(B):
        sampler = venv.make_sampler(seed + b)
        for _ in range(int(st.produced[b]) + discard):
            sampler.sample()
        ref = sampler.np_random.bit_generator.state
        assert (int(pcg[b, 1]) << 64 | int(pcg[b, 0])) == ref["state"]["state"]
        assert int(half[b, 0]) == ref["has_uint32"] and (not ref["has_uint32"] or int(half[b, 1]) == ref["uinteger"])


def test_stream_rejects_modes_the_device_sampler_does_not_cover():
    from melissa_amd.env import HipGraphVectorEnv, synthetic_graph_pool
    from melissa_amd.env.stream import EpisodeStream, make_supply, StaticSupply
    graphs = synthetic_graph_pool(12, 3, first_seed=0)
    venv = HipGraphVectorEnv(4, 12, graph_pool=graphs, dynamic_graph=True, device="cuda", construct_like_reference=False,
                             is_testing=True)
    with pytest.raises(ValueError, match="device sampler"):
        EpisodeStream(venv, 0)
    assert isinstance(make_supply(venv, 0, episodes_per_env=10), StaticSupply)      # falls back to a host-drawn table
    venv = HipGraphVectorEnv(4, 12, graph=graphs[0], dynamic_graph=True, device="cuda", construct_like_reference=False)
    with pytest.raises(ValueError, match="device sampler"):
        EpisodeStream(venv, 0)


@pytest.mark.parametrize("model", ["l_dgn", "hl_dgn"])
def test_long_run_never_replays_an_episode(model):
    """Thousands of resets through a 7-slot ring: no underrun flag, and at the end every env's ring holds exactly the
    episodes the numpy protocol gives for its LAST ordinals - a skipped, repeated or re-ordered draw anywhere in the run
    would leave the (sequential) generator somewhere else."""
    from melissa_amd import _lib as L
    from melissa_amd.collect import RoundLoop
    from melissa_amd.env import HipGraphVectorEnv, synthetic_graph_pool
    from melissa_amd.env.episodes import movement_offsets
    from melissa_amd.networks import HLDGNNetwork, LDGNNetwork
    from melissa_amd.policy import DQNPolicy
    n, B, K, seed, rounds = 20, 96, 7, 31, 700
    graphs = synthetic_graph_pool(n, 9, first_seed=1)
    torch.manual_seed(1)
    if model == "l_dgn":
        net = LDGNNetwork(5, 128, 2, 4, n, dueling_param=DUEL(), device="cuda", backend="hip")
    else:
        net = HLDGNNetwork(5, 128, 2, 4, n, aggregator="max", dueling_param=DUEL(), device="cuda", backend="hip")
    venv = HipGraphVectorEnv(B, n, graph_pool=graphs, dynamic_graph=True, device="cuda", max_moves=40,
                             construct_like_reference=False)
    loop = RoundLoop(venv, DQNPolicy(net), seed=seed, eps=0.2, ring=K, use_graph=True)
    loop.run(rounds)
    torch.cuda.synchronize()
    c = loop.counters()
    assert c["errors"] == 0 and c["episodes"] > 20 * B
    sc = venv.scalars().cpu().numpy()
    produced = loop.supply.produced.cpu().numpy()
    assert (sc[:, L.S_EP_CURSOR] == sc[:, L.S_EPISODES_DONE] + 1).all()
    assert (produced > sc[:, L.S_EP_CURSOR]).all() and (produced <= sc[:, L.S_EP_CURSOR] + K - 1).all()
    t = {k: v.cpu().numpy() for k, v in loop.supply.pool.tensors.items()}
    for b in range(0, B, 7):
        sampler = venv.make_sampler(seed + b)
        eps = [sampler.sample() for _ in range(int(produced[b]))]
        for j in range(int(sc[b, L.S_EP_CURSOR]) - 1, int(produced[b])):          # the running episode and the ones ahead
            slot = b * K + j % K
            assert t["origin"][slot] == eps[j].origin and int(t["interested"][slot].view(np.uint64)) == eps[j].interested
            np.testing.assert_array_equal(t["moves"][slot], movement_offsets(eps[j].movement_seed, n, 40))
    assert loop.supply.describe()["refills"] >= rounds // loop.supply.period
