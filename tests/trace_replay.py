"""Shared replay harness: drive any env exposing the PettingZooEnv-level surface through a golden
trace (tests/golden/env_trace_*.npz, produced by the real reference) and compare every output."""
import numpy as np

LOGGER_KEYS = ("total_messages_transmitted", "coverage", "messages_sent", "messages_received",
               "n_neighbours", "interested_agents", "coverage_interested_fraction",
               "coverage_interested_count", "uninterested_with_message", "episode_rewards_sum")


def set_int(x) -> int:
    """A node set in any of its forms - Python int (the oracle), np.uint64 / int64 scalar, or an array of 64-bit words
    (graphs beyond 64 nodes: word k = nodes 64 k .. 64 k + 63) - as a Python int bit mask."""
    if isinstance(x, int):
        return x
    a = np.atleast_1d(np.asarray(x))
    if a.dtype == object:
        return int(a[0])
    if a.dtype != np.uint64:
        a = a.astype(np.int64).view(np.uint64)
    out = 0
    for k, v in enumerate(a):
        out |= int(v) << (64 * k)
    return out


def set_ints(rows):
    return [set_int(v) for v in rows]


def check_row(tr, r, obs, rew, term, trunc, info, state=None):
    ctx = f"row {r}"
    assert int(obs["agent_id"]) == int(tr["agent_id"][r]), ctx
    np.testing.assert_array_equal(np.asarray(obs["obs"], dtype=np.float32), tr["obs"][r], err_msg=ctx)
    np.testing.assert_array_equal(np.asarray(obs["mask"], dtype=bool), tr["mask"][r], err_msg=ctx)
    # rewards: float64, same operation order as graph.py:402-463 -> exact
    np.testing.assert_array_equal(np.asarray(rew, dtype=np.float64), tr["rew"][r], err_msg=ctx)
    assert bool(term) == bool(tr["term"][r]), ctx
    assert bool(trunc) == bool(tr["trunc"][r]), ctx
    assert int(info["env_step"]) == int(tr["env_step"][r]), ctx
    assert bool(info["environment_step"]) == bool(tr["environment_step"][r]), ctx
    assert bool(info["explicit_reset"]) == bool(tr["explicit_reset"][r]), ctx
    np.testing.assert_array_equal(np.asarray(info["active_one_hop_neighbors"], dtype=bool),
                                  tr["active_nb"][r], err_msg=ctx)
    stats = info.get("logger_stats")
    assert (stats is not None) == bool(tr["has_stats"][r]), ctx
    if stats is not None:
        got = np.array([float(stats[k]) for k in LOGGER_KEYS])
        np.testing.assert_array_equal(got, tr["stats"][r], err_msg=ctx)
    if state is not None:
        for key in ("agents_mask", "alive_mask", "terminated_mask", "has_message_mask",
                    "interested_mask"):
            assert set_int(state[key]) == set_int(tr[key][r]), f"{ctx} {key}"
        if "scripted_mask" in tr.files and "scripted_mask" in state:
            assert set_int(state["scripted_mask"]) == set_int(tr["scripted_mask"][r]), f"{ctx} scripted_mask"
        assert int(state["origin"]) == int(tr["origin"][r]), ctx
        np.testing.assert_array_equal(np.asarray(state["pos"], dtype=np.float64), tr["pos"][r], err_msg=ctx)
        assert set_ints(state["one_hop"]) == set_ints(tr["one_hop"][r]), f"{ctx} one_hop"
        assert set_ints(state["two_hop"]) == set_ints(tr["two_hop"][r]), f"{ctx} two_hop"


def replay(tr, pz, state_fn=None):
    """pz: object with reset() -> (obs, info) and step(a) -> (obs, rew, term, trunc, info) and a
    sticky ``rewards`` list (tianshou PettingZooEnv surface).  Returns number of rows checked."""
    n = int(tr["n"])
    tape = tr["tape"]
    r = 0
    obs, info = pz.reset()
    check_row(tr, r, obs, pz.rewards, False if not tr["term"][r] else True, False, info,
              state_fn() if state_fn else None)
    r += 1
    done_count = 0
    for t in range(len(tape)):
        obs, rew, term, trunc, info = pz.step(int(tape[t]))
        check_row(tr, r, obs, rew, term, trunc, info, state_fn() if state_fn else None)
        r += 1
        if term or trunc:
            done_count += 1
            if done_count == n or info.get("explicit_reset", False):
                obs, info = pz.reset()
                assert bool(tr["was_reset"][r])
                check_row(tr, r, obs, pz.rewards, bool(tr["term"][r]), False, info,
                          state_fn() if state_fn else None)
                r += 1
                done_count = 0
    assert r == len(tr["agent_id"])
    return r


def scripted_kwargs(tr):
    """scripted_agents_ratio / heuristic a trace was recorded with (absent in the older traces: ratio 0)."""
    if "scripted_agents_ratio" not in tr.files or float(tr["scripted_agents_ratio"]) == 0.0:
        return {}
    return dict(scripted_agents_ratio=float(tr["scripted_agents_ratio"]), heuristic=str(tr["heuristic"]) or None)
