// Batched message-dissemination environment for MI355X (gfx950).
//
// One wavefront steps one env; lane i is node/agent i (N <= 64); every node set is one 64-bit mask
// held wave-uniformly, so the reference's Python list/dict walks become popcounts, ballots and a
// handful of cross-lane reads.  Integer state is bit-exact with the reference; float64 arithmetic
// (positions, rewards, logger stats) is written operation-for-operation like the Python expressions
// and the library is built with -ffp-contract=off, so it is bit-exact too.
//
// Reference semantics (paths relative to the reference repo):
//   GraphEnv.step / _execute_world_step / reward / observe / get_info   graph_env/env/graph.py:149-463
//   World.step / relay_message / move_graph / update_*_hop / reset       graph_env/env/utils/core.py:225-437
//   CustomSelector                                                       graph_env/env/utils/selector.py
//   [3P] AECEnv._deads_step_first / _was_dead_step, tianshou PettingZooEnv.step
// Scripted agents run the deterministic heuristics of heuristics/core.py (MEL_HEURISTIC_*).
#include "common.hpp"
#include "plan_masks.hpp"

namespace mel {

constexpr int NONE = -1;        // Python None action / False agent selection
constexpr int SKIP_NONE = -2;   // _skip_agent_selection is None
constexpr int MAX_AGENT_STEPS = 4;          // graph.py:332, selector.py:44
constexpr double R2_F64 = 0.04000000000000001;   // 0.2 ** 2 (nx.geometric_edges, core.py:311)

// every cross-lane read in this file has a wave-uniform source lane (loop counter / selected agent)
__device__ __forceinline__ double shfl_f64(double v, int src) { return lane_f64(v, src); }
__device__ __forceinline__ int wave_sum_i32(int v) { return wave_sum_i32_dpp(v); }
__device__ __forceinline__ uint64_t bit(int i) { return 1ull << i; }

// Wave-uniform working copy of one env's masks/scalars + this lane's per-node values.
struct Env {
    // uniform
    uint64_t has_msg, origin_set, interested, scripted, truncated, alive, terminated, agents;
    uint64_t sel_active, sel_selected, info_valid, taken_action;
    int origin, sel, skip, num_moves, world_msgs, new_round, episode, move_cursor, decisions, done_count,
        episodes_done, error, ep_cursor;
    double episode_rewards;
    // per lane (node)
    double px, py, reward, pz_reward;
    uint64_t one_hop, two_hop;
    int msgs, received, cover;
    int act, cur_act, steps, sel_steps;
    // not stored: get_info's ten values (graph.py:166-178), lane k holds value k; they only change in World.step / reset, so
    // the AEC sub-steps of a round reuse them (three wave sums and two float64 divisions once instead of per sub-step)
    double info_val;
    int info_valid_cache;
};

__device__ __forceinline__ void env_load(const mel_env_batch& e, int b, int lane, Env& s) {
    const int n = e.n_nodes;
    // wave-uniform state: forced into SGPRs so mask arithmetic and branches run on the scalar unit
    const uint64_t* ns = e.node_sets + (size_t)b * 8;
    s.has_msg = uniform_u64(ns[0]), s.origin_set = uniform_u64(ns[1]), s.interested = uniform_u64(ns[2]);
    s.scripted = uniform_u64(ns[3]), s.truncated = uniform_u64(ns[4]), s.alive = uniform_u64(ns[5]);
    s.terminated = uniform_u64(ns[6]), s.agents = uniform_u64(ns[7]);
    const uint64_t* ss = e.sel_sets + (size_t)b * 4;
    s.sel_active = uniform_u64(ss[0]), s.sel_selected = uniform_u64(ss[1]);
    s.info_valid = uniform_u64(ss[2]), s.taken_action = uniform_u64(ss[3]);
    const int32_t* sc = e.scalars + (size_t)b * MEL_ENV_SCALARS;
    s.origin = uniform_i32(sc[MEL_S_ORIGIN]), s.sel = uniform_i32(sc[MEL_S_SELECTION]), s.skip = uniform_i32(sc[MEL_S_SKIP]);
    s.num_moves = uniform_i32(sc[MEL_S_NUM_MOVES]), s.world_msgs = uniform_i32(sc[MEL_S_WORLD_MSGS]);
    s.new_round = uniform_i32(sc[MEL_S_NEW_ROUND]), s.episode = uniform_i32(sc[MEL_S_EPISODE]);
    s.move_cursor = uniform_i32(sc[MEL_S_MOVE_CURSOR]), s.decisions = uniform_i32(sc[MEL_S_DECISIONS]);
    s.done_count = uniform_i32(sc[MEL_S_DONE_COUNT]), s.episodes_done = uniform_i32(sc[MEL_S_EPISODES_DONE]);
    s.error = uniform_i32(sc[MEL_S_ERROR]), s.ep_cursor = uniform_i32(sc[MEL_S_EP_CURSOR]);
    s.episode_rewards = e.episode_rewards[b];
    const size_t k = (size_t)b * n + lane;
    const bool on = lane < n;
    s.px = on ? e.pos[2 * k] : 0.0, s.py = on ? e.pos[2 * k + 1] : 0.0;
    s.reward = on ? e.rewards[k] : 0.0, s.pz_reward = on ? e.pz_rewards[k] : 0.0;
    s.one_hop = on ? e.one_hop[k] : 0ull, s.two_hop = on ? e.two_hop[k] : 0ull;
    s.msgs = on ? e.agent_msgs[k] : 0, s.received = on ? e.received[k] : 0, s.cover = on ? e.two_hop_cover[k] : 0;
    s.act = on ? e.agent_action[k] : NONE, s.cur_act = on ? e.current_actions[k] : NONE;
    s.steps = on ? e.steps_taken[k] : 0, s.sel_steps = on ? e.sel_steps[k] : 0;
    s.info_val = 0.0, s.info_valid_cache = 0;
}

__device__ __forceinline__ void env_store(const mel_env_batch& e, int b, int lane, const Env& s) {
    const int n = e.n_nodes;
    if (lane == 0) {
        uint64_t* ns = e.node_sets + (size_t)b * 8;
        ns[0] = s.has_msg, ns[1] = s.origin_set, ns[2] = s.interested, ns[3] = s.scripted;
        ns[4] = s.truncated, ns[5] = s.alive, ns[6] = s.terminated, ns[7] = s.agents;
        uint64_t* ss = e.sel_sets + (size_t)b * 4;
        ss[0] = s.sel_active, ss[1] = s.sel_selected, ss[2] = s.info_valid, ss[3] = s.taken_action;
        int32_t* sc = e.scalars + (size_t)b * MEL_ENV_SCALARS;
        sc[MEL_S_ORIGIN] = s.origin, sc[MEL_S_SELECTION] = s.sel, sc[MEL_S_SKIP] = s.skip;
        sc[MEL_S_NUM_MOVES] = s.num_moves, sc[MEL_S_WORLD_MSGS] = s.world_msgs, sc[MEL_S_NEW_ROUND] = s.new_round;
        sc[MEL_S_EPISODE] = s.episode, sc[MEL_S_MOVE_CURSOR] = s.move_cursor, sc[MEL_S_DECISIONS] = s.decisions;
        sc[MEL_S_DONE_COUNT] = s.done_count, sc[MEL_S_EPISODES_DONE] = s.episodes_done, sc[MEL_S_ERROR] = s.error;
        sc[MEL_S_EP_CURSOR] = s.ep_cursor;
        e.episode_rewards[b] = s.episode_rewards;
    }
    if (lane < n) {
        const size_t k = (size_t)b * n + lane;
        e.pos[2 * k] = s.px, e.pos[2 * k + 1] = s.py;
        e.rewards[k] = s.reward, e.pz_rewards[k] = s.pz_reward;
        e.one_hop[k] = s.one_hop, e.two_hop[k] = s.two_hop;
        e.agent_msgs[k] = s.msgs, e.received[k] = s.received, e.two_hop_cover[k] = s.cover;
        e.agent_action[k] = (int8_t)s.act, e.current_actions[k] = (int8_t)s.cur_act;
        e.steps_taken[k] = (int8_t)s.steps, e.sel_steps[k] = (int8_t)s.sel_steps;
    }
}

// core.py:334-341: one-hop OR neighbours' one-hop, minus self
__device__ __forceinline__ uint64_t two_hop_of(uint64_t one_hop, int lane, int n) {
    // every lane walks ITS OWN neighbours (a handful, not all n nodes) and ORs their rows in, fetched from LDS by a
    // per-lane gather; the wave iterates max-degree times instead of n times with two v_readlane each
    __shared__ uint64_t rows[4][64];
    const int w = (threadIdx.x >> 6) & 3;
    (void)n;
    rows[w][lane] = one_hop;                    // (lanes >= n hold 0)
    uint64_t m = one_hop, rest = one_hop;
    while (__ballot(rest != 0ull)) {
        if (rest) {
            const int j = __ffsll((long long)rest) - 1;
            rest &= rest - 1ull;
            m |= rows[w][j];
        }
    }
    return m & ~bit(lane);
}

// nx.geometric_edges (core.py:311): edge iff dx*dx + dy*dy <= 0.2**2 in float64.
// Node j's position reaches the lanes as an LDS broadcast (every lane reads the same address) instead of four
// v_readlane into SGPRs: the loop body is then pure VALU on VGPR operands, without the SGPR write -> VALU read wait
// states that dominated it with one wavefront per SIMD (workgroups of the env kernels are 4 wavefronts).
__device__ __forceinline__ uint64_t geometric_one_hop(double px, double py, int lane, int n) {
    __shared__ double sx[4][64], sy[4][64];
    const int w = (threadIdx.x >> 6) & 3;
    sx[w][lane] = px, sy[w][lane] = py;
    uint64_t m = 0;
#pragma unroll 5
    for (int j = 0; j < n; ++j) {
        const double dx = px - sx[w][j], dy = py - sy[w][j];
        const double d2 = dx * dx + dy * dy;
        if (d2 <= R2_F64) m |= bit(j);
    }
    return (lane < n) ? (m & ~bit(lane)) : 0ull;
}

// selector.py:25-34
__device__ __forceinline__ int selector_next(Env& s, int lane) {
    const uint64_t cand = s.sel_active & ~s.sel_selected;
    if (!cand) return NONE;
    const int i = lowest_bit(cand);
    if (lane == i) s.sel_steps += 1;
    s.sel_selected |= bit(i);
    return i;
}
// selector.py:43-44
__device__ __forceinline__ void selector_enable(Env& s, uint64_t agents, int lane) {
    const uint64_t can = __ballot(s.sel_steps < MAX_AGENT_STEPS);
    s.sel_active = (s.sel_active & ~agents) | (agents & can);
}

#ifdef MEL_ENV_PROF
// finer split of the world step (wave 0 lane 0 of every eighth env adds): [0] relay + scripted, [1] move + all-pairs edges,
// [2] two-hop masks, [3] calls
__device__ unsigned long long g_world_prof[4];
#endif

// World.step core.py:225-266
__device__ __forceinline__ void world_step(const mel_env_batch& e, const mel_episode_pool& pool, Env& s,
                                           int lane) {
    const int n = e.n_nodes;
#ifdef MEL_ENV_PROF
    const unsigned long long wp0 = __builtin_readcyclecounter();
#endif
    s.info_valid_cache = 0;                                  // message counters, coverage and (dynamic graph) degrees change here
    // :226-234 scripted agents: action = heuristic(agent) (the heuristics offered return no relay mask, so the
    // relays_for pass :236-243 never fires)
    // (no heuristic: Agent.action_callback stays None, World.scripted_agents is empty, nothing is overridden)
    const bool scripted_lane = e.heuristic != MEL_HEURISTIC_NONE && lane < n && ((s.scripted >> lane) & 1ull);
    if (scripted_lane) {
        if (e.heuristic == MEL_HEURISTIC_SIMPLE_BROADCAST) s.act = ((s.taken_action >> lane) & 1ull) ? 0 : 1;
        // Agent.number_interested_neighbors: counted at reset (core.py:401) but zeroed again by agent.reset() ->
        // Agent.__init__ (:412-416, :68); only move_graph recomputes it (:286-287), so it is 0 until the episode's
        // first move and tracks the current graph afterwards (one_hop changes only in moves)
        else if (e.heuristic == MEL_HEURISTIC_BROADCAST_IF_INTERESTED)
            s.act = (e.dynamic_graph && s.move_cursor > 0 && (s.one_hop & s.interested)) ? 1 : 0;
        else s.act = 0;                                          // silent
    }
    // :246 the source always transmits on its first opportunity
    const int origin_msgs = lane_i32(s.msgs, s.origin);
    if (lane == s.origin && origin_msgs == 0) s.act = 1;
    // :249-254 relay in id order; has_message is re-read at each agent's turn
    uint64_t cand = __ballot(lane < n && s.act != NONE && s.act != 0);
    while (cand) {
        const int i = lowest_bit(cand);
        cand &= cand - 1;
        if ((s.has_msg >> i) & 1ull) {                       // relay_message core.py:268-279
            const uint64_t nb = lane_u64(s.one_hop, i);
            s.world_msgs += 1;
            if (lane == i) s.msgs += 1;
            s.taken_action |= bit(i);
            s.received += (int)((nb >> lane) & 1ull);
            s.has_msg |= nb;
        }
    }
#ifdef MEL_ENV_PROF
    asm volatile("s_nop 0" ::"s"(s.has_msg), "v"(s.received));
    const unsigned long long wp1 = __builtin_readcyclecounter();
    unsigned long long wp2 = wp1, wp3 = wp1;
#endif
    // :256-257 move_graph -> update_position + one/two hop recompute (core.py:281-341)
    if (e.dynamic_graph) {
        int mv = s.move_cursor;
        if (mv >= pool.max_moves) {
            mv = pool.max_moves - 1;
            s.error |= MEL_ENV_ERR_MOVES_EXHAUSTED;
        }
        if (lane < n) {
            const double* off = pool.moves + ((size_t)s.episode * pool.max_moves + mv) * 2 * n;
            s.px = s.px + off[lane];
            s.py = s.py + off[n + lane];
        }
        s.move_cursor += 1;
        s.one_hop = geometric_one_hop(s.px, s.py, lane, n);
#ifdef MEL_ENV_PROF
        asm volatile("s_nop 0" ::"v"(s.one_hop));
        wp2 = __builtin_readcyclecounter();
#endif
        s.two_hop = two_hop_of(s.one_hop, lane, n);
#ifdef MEL_ENV_PROF
        asm volatile("s_nop 0" ::"v"(s.two_hop));
        wp3 = __builtin_readcyclecounter();
#endif
    }
#ifdef MEL_ENV_PROF
    if (lane == 0 && (blockIdx.x & 1) == 0 && (threadIdx.x >> 6) == 0) {
        atomicAdd(&g_world_prof[0], wp1 - wp0), atomicAdd(&g_world_prof[1], wp2 - wp1), atomicAdd(&g_world_prof[2], wp3 - wp2);
        atomicAdd(&g_world_prof[3], 1ull);
    }
#endif
    // :260-261 -> Agent.update_two_hop_cover_from_one_hopper (core.py:94-102)
    s.cover = __popcll(s.two_hop & (s.has_msg | s.origin_set));
    if (scripted_lane) s.act = 0;                                // :264-266
}

// graph.py:254-271 (copy: optional second destination, the replay's obs_next slot)
__device__ __forceinline__ void write_obs_matrix(const mel_env_batch& e, int b, const Env& s, int lane,
                                                 float* copy = nullptr) {
    if (lane >= e.n_nodes) return;
    float4* row = reinterpret_cast<float4*>(e.obs_matrix + ((size_t)b * e.n_nodes + lane) * 8);
    const float act = (s.act != NONE) ? (float)s.act : 0.f;
    const float interested = ((s.interested >> lane) & 1ull) ? 1.f : 0.f;
    const float has = (((s.has_msg | s.origin_set) >> lane) & 1ull) ? 1.f : 0.f;
    const float dm = ((s.scripted >> lane) & 1ull) ? 0.f : 1.f;
    row[0] = make_float4((float)s.px, (float)s.py, (float)__popcll(s.one_hop), (float)s.msgs);
    row[1] = make_float4(act, interested, has, dm);
    if (copy) {
        float4* c = reinterpret_cast<float4*>(copy + (size_t)lane * 8);
        c[0] = row[0];
        c[1] = row[1];
    }
}

// graph.py:402-463, float64, same operation order
__device__ __forceinline__ double agent_reward(const Env& s) {
    const uint64_t covered = s.has_msg | s.origin_set;
    const int total = __popcll(s.two_hop & s.interested);
    const int cov = __popcll(s.two_hop & s.interested & covered);
    double reward = total > 0 ? (double)cov / (double)total : 0.0;
    const int deg = __popcll(s.one_hop);
    if (s.act != NONE && s.act != 0) {
        const double pen_unint = deg > 0 ? (double)__popcll(s.one_hop & ~s.interested) / (double)deg : 0.0;
        const double pen_cov = deg > 0 ? (double)__popcll(s.one_hop & s.has_msg) / (double)deg : 0.0;
        const double penalty = pen_unint + pen_cov;
        reward -= penalty;
    } else {
        const int one_int = __popcll(s.one_hop & s.interested);
        const int unc = __popcll(s.one_hop & s.interested & ~s.has_msg & ~s.origin_set);
        if (unc > 0) reward -= (double)unc / (double)one_int;
    }
    return reward;
}

// graph.py:149-179 -> infos[agent]['logger_stats'] (10 float64 in dict order)
__device__ __forceinline__ void write_info_stats(const mel_env_batch& e, int b, int agent, Env& s, int lane) {
    const int n = e.n_nodes;
    if (!s.info_valid_cache) {
        const int sent = wave_sum_i32(lane < n ? s.msgs : 0);
        const int recv = wave_sum_i32(lane < n ? s.received : 0);
        const int nbrs = wave_sum_i32(lane < n ? __popcll(s.one_hop) : 0);
        const int n_int = __popcll(s.interested);
        const int cov_int = __popcll(s.has_msg & s.interested);
        const double st1 = (double)__popcll(s.has_msg) / (double)n;
        const double st6 = n_int > 0 ? (double)cov_int / (double)n_int : 0.0;
        double v = (double)s.world_msgs;                                   // lane 0
        v = lane == 1 ? st1 : v;
        v = lane == 2 ? (double)sent : v;
        v = lane == 3 ? (double)recv : v;
        v = lane == 4 ? (double)nbrs : v;
        v = lane == 5 ? (double)n_int : v;
        v = lane == 6 ? st6 : v;
        v = lane == 7 ? (double)cov_int : v;
        v = lane == 8 ? (double)__popcll(s.has_msg & ~s.interested) : v;
        v = lane == 9 ? s.episode_rewards : v;
        s.info_val = v;
        s.info_valid_cache = 1;
    }
    if (lane < MEL_ENV_LOGGER_STATS) e.info_stats[((size_t)b * n + agent) * MEL_ENV_LOGGER_STATS + lane] = s.info_val;
}

// one row of the episode log: the logger_stats of the final observation of the episode that just ended
__device__ __forceinline__ void log_episode(const mel_env_batch& e, int b, const Env& s, int lane) {
    if (e.log_capacity <= 0) return;
    int slot = 0;
    if (lane == 0) slot = atomicAdd(e.log_cursor, 1);
    slot = uniform_i32(slot);
    if (slot >= e.log_capacity) return;
    // the info the collectors see is infos[agent_selection] AS STORED (graph.py:358 refreshes it only when that agent
    // becomes the selection after a live step; a dead step returns before that, graph.py:304-310), so the row is a
    // copy of the selected agent's slot, not a recomputation from the current state
    if (lane < MEL_ENV_LOGGER_STATS) {
        const int a = s.sel >= 0 ? s.sel : 0;
        const bool valid = s.sel >= 0 && ((s.info_valid >> a) & 1ull);
        e.log_stats[(size_t)slot * MEL_ENV_LOGGER_STATS + lane] =
            valid ? e.info_stats[((size_t)b * e.n_nodes + a) * MEL_ENV_LOGGER_STATS + lane] : 0.0;
    }
    if (lane == 0) {
        int32_t* m = e.log_meta + (size_t)slot * 3;
        m[0] = b, m[1] = s.episode, m[2] = s.num_moves;
    }
}

// GraphEnv.step graph.py:303-359 (+ the sticky reward copy of [3P] PettingZooEnv.step)
// returns true when this step completed the round and ran the world step
__device__ __forceinline__ bool env_step(const mel_env_batch& e, const mel_episode_pool& pool, int b, Env& s,
                                         int action, int lane, float* obs_next_copy = nullptr) {
    const int n = e.n_nodes;
    const int a = s.sel;
    bool world = false;
    if (a < 0) {                      // reference would raise KeyError; flag it and stay put
        s.error |= 2;
        return false;
    }
    if ((s.terminated >> a) & 1ull) {                                   // :304-310 dead step
        s.sel_active &= ~bit(a);
        s.alive &= ~bit(a);                                             // _was_dead_step :274-301
        s.terminated &= ~bit(a);
        s.info_valid &= ~bit(a);
        s.agents &= ~bit(a);
        const uint64_t dead = s.agents & s.terminated;
        if (dead) {
            if (s.skip == SKIP_NONE) s.skip = a;
            s.sel = lowest_bit(dead);
        } else {
            if (s.skip != SKIP_NONE) s.sel = s.skip;
            s.skip = SKIP_NONE;
        }
    } else {
        s.decisions += 1;
        if (lane == a) {
            s.cur_act = action;                                         // :314
            s.steps += 1;                                               // :316-318
        }
        s.sel = selector_next(s, lane);                                 // :321
        if (s.sel == NONE) {                                            // :324 round complete
            world = true;
            if ((s.alive >> lane) & 1ull) s.reward = 0.0;               // _clear_rewards
            s.act = s.cur_act;                                          // :362-365
            world_step(e, pool, s, lane);
            write_obs_matrix(e, b, s, lane, obs_next_copy);             // :370-371
            const double r0 = agent_reward(s);
            double r = r0;
            if (e.has_local_ratio) r = 0.0 * (1.0 - e.local_ratio) + r0 * e.local_ratio;   // :380-384
            if ((s.agents >> lane) & 1ull) s.reward = r;                // :386
            uint64_t rest = s.agents;                                   // :387 summed in id order
            while (rest) {
                const int i = lowest_bit(rest);
                rest &= rest - 1;
                s.episode_rewards += shfl_f64(r, i);
            }
            s.num_moves += 1;                                           // :328
            const uint64_t expire = __ballot(lane < n && s.steps >= MAX_AGENT_STEPS) & s.agents & ~s.truncated;
            s.truncated |= expire;                                      // :330-334
            s.terminated |= expire;
            s.agents = s.has_msg & s.alive & (e.is_testing ? ~0ull : ~s.scripted);   // :336-341
            selector_enable(s, s.agents, lane);                         // :342
            s.sel_selected = 0;                                         // :343
            s.new_round = 1;                                            // :344
            s.sel = selector_next(s, lane);                             // :345
            s.cur_act = NONE;                                           // :347
        }
        if (s.sel != NONE) {                                            // :358
            write_info_stats(e, b, s.sel, s, lane);
            s.info_valid |= bit(s.sel);
        }
        const uint64_t dead = s.agents & s.terminated;                  // :359 _deads_step_first
        if (dead) {
            s.skip = s.sel;
            s.sel = lowest_bit(dead);
        }
    }
    if ((s.alive >> lane) & 1ull) s.pz_reward = s.reward;               // PettingZooEnv.step (A.6)
    return world;
}

// GraphEnv.reset + World.reset, graph.py:222-248 / core.py:343-437, from a pre-sampled pool episode
__device__ __forceinline__ void env_reset(const mel_env_batch& e, const mel_episode_pool& pool, int b, Env& s,
                                          int episode, int keep_graph, int lane) {
    const int n = e.n_nodes;
    const uint64_t full = (n == 64) ? ~0ull : (bit(n) - 1ull);
    s.episode = episode;
    s.move_cursor = 0;
    if (!keep_graph) {
        const size_t k = (size_t)episode * n + lane;
        s.px = lane < n ? pool.pos[2 * k] : 0.0;
        s.py = lane < n ? pool.pos[2 * k + 1] : 0.0;
        s.one_hop = lane < n ? pool.one_hop[k] : 0ull;
    }
    s.two_hop = two_hop_of(s.one_hop, lane, n);                         // core.py:421
    s.origin = uniform_i32(pool.origin[episode]);
    s.interested = uniform_u64(pool.interested[episode]) & full;
    s.scripted = pool.scripted ? (uniform_u64(pool.scripted[episode]) & full) : 0ull;   // core.py:395,404
    s.world_msgs = 0;
    s.has_msg = bit(s.origin);                                          // :432-434
    s.origin_set = bit(s.origin);
    s.taken_action = 0;
    s.truncated = 0;
    s.msgs = 0, s.received = 0, s.cover = 0;
    s.act = NONE;
    s.steps = (lane == s.origin) ? 1 : 0;                               // :424,435
    world_step(e, pool, s, lane);                                       // :437
    // GraphEnv.reset
    s.sel_steps = (lane == s.origin) ? 1 : 0;                           // selector.reinit + enable(on_reset)
    s.sel_active = 0, s.sel_selected = 0;
    s.reward = 0.0;
    s.alive = full, s.terminated = 0, s.info_valid = 0;
    s.num_moves = 0;
    s.episode_rewards = 0.0;
    s.done_count = 0;
    write_obs_matrix(e, b, s, lane);
    s.agents = s.has_msg & (e.is_testing ? ~0ull : ~s.scripted);        // :242-245
    selector_enable(s, s.agents, lane);
    s.sel = selector_next(s, lane);                                     // :247
    s.skip = SKIP_NONE;
    s.cur_act = NONE;                                                   // :248
    s.ep_cursor += 1;
}

// GraphEnv.observe / last() graph.py:181-216 + PettingZooEnv packing; returns bit0 terminated,
// bit1 explicit_reset, bit2 environment_step
__device__ __forceinline__ int env_observe(const mel_env_batch& e, int b, Env& s, const mel_env_obs& o,
                                           int64_t row, int lane) {
    const int n = e.n_nodes;
    const int a = s.sel;
    const int valid = a >= 0;
    const int dead = valid ? (int)((s.terminated >> a) & 1ull) : 0;
    int explicit_reset = 0, environment_step = 0;
    if (__popcll(s.agents) == 1 && (s.agents & ~s.terminated) == 0) {   // :205-207
        s.new_round = 0;
        explicit_reset = 1;
    }
    if (s.new_round == 1) {                                             // :209-211
        environment_step = 1;
        s.new_round = 0;
    }
    if (o.obs) {
        float* dst = o.obs + row * o.obs_stride;
        const float* src = e.obs_matrix + (size_t)b * n * 8;
        for (int c = lane; c < n * 8; c += 64) dst[c] = src[c];
        if (lane == 0) dst[n * 8] = (float)a;
    }
    if (o.rew && lane < n) o.rew[row * n + lane] = s.pz_reward;
    if (o.stats && lane < MEL_ENV_LOGGER_STATS) {
        const int ok = valid && ((s.info_valid >> a) & 1ull);
        o.stats[row * MEL_ENV_LOGGER_STATS + lane] =
            ok ? e.info_stats[((size_t)b * n + a) * MEL_ENV_LOGGER_STATS + lane] : 0.0;
    }
    const uint64_t nb = lane_u64(s.one_hop, valid ? a : 0);
    if (lane == 0) {
        if (o.agent_id) o.agent_id[row] = a;
        if (o.action_mask) o.action_mask[2 * row] = o.action_mask[2 * row + 1] = dead ? 0 : 1;   // :190-192
        if (o.terminated) o.terminated[row] = (uint8_t)dead;
        if (o.flags) {
            o.flags[4 * row + 0] = s.num_moves;
            o.flags[4 * row + 1] = environment_step;
            o.flags[4 * row + 2] = explicit_reset;
            o.flags[4 * row + 3] = valid ? (int)((s.info_valid >> a) & 1ull) : 0;
        }
        if (o.active_nb) o.active_nb[row] = valid ? (nb & ~(s.truncated & ~s.agents)) : 0ull;   // :198-203
    }
    return dead | (explicit_reset << 1) | (environment_step << 2);
}

// One whole env round per launch (round-batched loop): the AEC steps of GraphEnv.step are replayed in the
// reference's order - pending dead agents first (graph.py:304-310,359), then every active agent in id
// order with ITS action (selector.py:25-34) - until the world step fires (graph.py:324-347) or the episode
// ends.  State evolution is identical to calling env_step + env_observe once per agent.
struct RoundArgs {
    mel_env_batch env;
    mel_episode_pool pool;
    const int32_t* actions;        // [rows] one action per (env, active agent), rows ordered by env, agent id
    const int32_t* row_offsets;    // [B+1] first row of each env (as mel_ldgn_forward_agents wrote them)
    uint64_t* live;                // [B] in: the active set the actions were computed for; out: next round's
    const int32_t* episode_table;
    int table_stride;
    int first;                     // 1: only publish the active set (no step) - used right after a reset
    uint32_t* round_counter;       // optional: rounds played (device), drives the device-side RNG step
    mel_round_replay replay;       // optional (capacity 0 = off)
    mel_env_batch snap;            // optional reset snapshots (mel_episode_pool::snapshot), env e = episode e after its reset
    int has_snap;
};

// GraphEnv.reset from a snapshot: the state env_reset would compute for this episode was computed once when the pool was
// loaded (same kernel, same settings); what a reset does NOT touch is carried over from the live env.
__device__ __forceinline__ void env_reset_from_snapshot(const mel_env_batch& e, const mel_env_batch& snap, int b, Env& s,
                                                        int episode, int lane) {
    Env t;
    env_load(snap, episode, lane, t);
    t.new_round = s.new_round, t.decisions = s.decisions, t.episodes_done = s.episodes_done;
    t.error = s.error | t.error, t.ep_cursor = s.ep_cursor + 1;
    t.pz_reward = s.pz_reward;                 // the sticky Tianshou reward vector survives a reset
    s = t;
    write_obs_matrix(e, b, s, lane);
}

#ifdef MEL_ENV_PROF
// tuning builds: cycles a sample of the env wavefronts spends in [0] state load, [1] the round loop, [2] the env_step call
// that runs the world step, [3] the other env_step calls, [4] env_observe, [5] episode end (log + reset), [6] state store;
// [7] wavefronts sampled, [8] loop iterations                                               (tools/env_prof.py)
__device__ unsigned long long g_env_prof[9];
#define ENV_T() __builtin_readcyclecounter()
#define ENV_MARK(x) asm volatile("s_nop 0" ::"s"(x))
#endif

__global__ __launch_bounds__(256) void env_round_kernel(RoundArgs a) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= a.env.n_envs) return;
    const int lane = lane_id();
    const int n = a.env.n_nodes;
    if (a.round_counter && b == 0 && lane == 0 && !a.first) atomicAdd(a.round_counter, 1u);
    Env s;
#ifdef MEL_ENV_PROF
    unsigned long long pw = 0, ps = 0, po = 0, pr = 0, pit = 0;
    const unsigned long long p0 = ENV_T();
#endif
    env_load(a.env, b, lane, s);
#ifdef MEL_ENV_PROF
    ENV_MARK(s.sel);
    asm volatile("s_nop 0" ::"v"(s.px), "v"(s.steps));
    const unsigned long long p1 = ENV_T();
#endif
    const mel_env_obs none{};
    if (!a.first) {
        const uint64_t live_in = uniform_u64(a.live[b]);
        // every agent's action in one coalesced load (lane = agent), read back with v_readlane; actions are
        // either packed rows (row_offsets) or the dense [B, N] layout
        int my_action = 0;
        if ((live_in >> lane) & 1ull)
            my_action = a.row_offsets ? a.actions[uniform_i32(a.row_offsets[b]) + rank_below(live_in, lane)]
                                      : a.actions[(size_t)b * n + lane];
        // replay record of this round (only envs with acting agents): pre-state now, outcome after the world step
        float* rec_next = nullptr;
        size_t rec = 0;
        if (a.replay.capacity > 0 && live_in) {
            const int cur = uniform_i32(a.replay.cursor[b]);
            rec = (size_t)b * a.replay.capacity + (cur % a.replay.capacity);
            const float* src = a.env.obs_matrix + (size_t)b * n * 8;
            float* dst = a.replay.obs + rec * n * 8;
            for (int c = lane; c < n * 8; c += 64) dst[c] = src[c];
            if (lane < n) a.replay.act[rec * n + lane] = (int8_t)my_action;
            if (lane == 0) {
                a.replay.acted[rec] = live_in;
                a.replay.episode[rec] = s.ep_cursor;
                a.replay.cursor[b] = cur + 1;
            }
            rec_next = a.replay.obs_next + rec * n * 8;
        }
        // ---- fast-forward of the bookkeeping-only AEC sub-steps of the round ------------------------------------------
        // (1) Pending dead agents (GraphEnv.step :304-310 + _deads_step_first :359: the selection is the first terminated
        //     agent, the real one waits in skip).  Each dead step only removes its agent and moves the selection to the
        //     next dead agent, at last back to skip; every observe that lands on a dead agent counts one done.  Done in one
        //     go unless the episode could end inside the sequence (every active agent dead, or the done count reaching n):
        //     those rare rounds take the sequential path.
        {
            const uint64_t dead = s.agents & s.terminated;
            if (dead && s.skip >= 0 && s.sel == lowest_bit(dead) && (s.agents & ~dead) != 0ull &&
                s.done_count + __popcll(dead) - 1 < n && s.error == 0) {
                s.sel_active &= ~dead, s.alive &= ~dead, s.terminated &= ~dead, s.info_valid &= ~dead, s.agents &= ~dead;
                s.done_count += __popcll(dead) - 1;
                s.sel = s.skip;
                s.skip = SKIP_NONE;
                if ((s.alive >> lane) & 1ull) s.pz_reward = s.reward;   // PettingZooEnv.step (A.6), idempotent
            }
        }
        // (2) No pending dead agent, the selection is the first active agent and the forward covered exactly the active
        //     set: the sub-steps of every acting agent but the LAST change nothing but bookkeeping - its action and step
        //     count, the selector's marks, the stored info of the next selection (the same ten values: nothing moves
        //     before the world step) - so they are applied in one go (graph.py:312-321,358, k - 1 times); the last agent
        //     takes the sequential path below, which runs the world step.
        if (s.skip == SKIP_NONE && (s.agents & s.terminated) == 0ull && live_in != 0ull && live_in == s.sel_active &&
            (live_in & ~s.agents) == 0ull && s.sel >= 0 && s.sel == lowest_bit(live_in) && s.sel_selected == bit(s.sel) &&
            s.error == 0) {
            const int last = 63 - __clzll((long long)live_in);
            const uint64_t early = live_in & ~bit(last);             // agents whose sub-step is fast-forwarded
            if (early) {
                s.decisions += __popcll(early);
                if ((early >> lane) & 1ull) {                        // :314-318
                    s.cur_act = my_action;
                    s.steps += 1;
                }
                const uint64_t later = live_in & ~bit(s.sel);        // every agent that becomes the selection: a_2 .. a_k
                if ((later >> lane) & 1ull) s.sel_steps += 1;       // selector.py:25-34
                s.sel_selected = live_in;
                uint64_t rest = later;                               // :358 infos[next] = get_info()
                while (rest) {
                    const int nxt = lowest_bit(rest);
                    rest &= rest - 1ull;
                    write_info_stats(a.env, b, nxt, s, lane);
                }
                s.info_valid |= later;
                if ((s.alive >> lane) & 1ull) s.pz_reward = s.reward;
                s.new_round = 0;                                     // the first observe of the round (:209-211)
                s.sel = last;
            }
        }
        for (int it = 0; it < 3 * n + 4; ++it) {
            const int sel = s.sel;
            if (sel < 0) {
                s.error |= 2;
                break;
            }
            int action = 0;
            if (!((s.terminated >> sel) & 1ull)) {
                if (!((live_in >> sel) & 1ull)) {          // an agent acts that the forward did not cover
                    s.error |= 4;
                    break;
                }
                action = lane_i32(my_action, sel);
            }
#ifdef MEL_ENV_PROF
            const unsigned long long q0 = ENV_T();
#endif
            const bool world = env_step(a.env, a.pool, b, s, action, lane, rec_next);
#ifdef MEL_ENV_PROF
            ENV_MARK(s.sel);
            asm volatile("s_nop 0" ::"v"(s.reward), "v"(s.one_hop));
            const unsigned long long q1 = ENV_T();
#endif
            const int r = env_observe(a.env, b, s, none, 0, lane);
#ifdef MEL_ENV_PROF
            ENV_MARK(r);
            const unsigned long long q2 = ENV_T();
            (world ? pw : ps) += q1 - q0, po += q2 - q1, pit += 1;
#endif
            if (world && rec_next) {          // the world step just ran: rewards / terminations of this round
                if (lane < n) a.replay.rew[rec * n + lane] = ((live_in >> lane) & 1ull) ? (float)s.reward : 0.f;
                if (lane == 0) a.replay.done[rec] = s.terminated & live_in;
            }
            if (r & 1) s.done_count += 1;
            if ((r & 1) && ((r & 2) || s.done_count == n)) {                  // episode over
                s.episodes_done += 1;
#ifdef MEL_ENV_PROF
                const unsigned long long r0 = ENV_T();
#endif
                log_episode(a.env, b, s, lane);
                if (a.pool.produced && s.ep_cursor >= uniform_i32(a.pool.produced[b])) s.error |= MEL_ENV_ERR_EPISODE_UNDERRUN;
                const int ep = uniform_i32(a.episode_table[(size_t)b * a.table_stride + (s.ep_cursor % a.table_stride)]);
                if (a.has_snap) env_reset_from_snapshot(a.env, a.snap, b, s, ep, lane);
                else env_reset(a.env, a.pool, b, s, ep, 0, lane);
                env_observe(a.env, b, s, none, 0, lane);
#ifdef MEL_ENV_PROF
                ENV_MARK(s.sel);
                asm volatile("s_nop 0" ::"v"(s.px), "v"(s.one_hop));
                pr += ENV_T() - r0;
#endif
                break;
            }
            if (r & 4) break;                                                  // world step done: new round
        }
    }
#ifdef MEL_ENV_PROF
    const unsigned long long p2 = ENV_T();
#endif
    // agents that will act in the coming round: exactly the selector's active set (selector.py:22-34,43-44)
    if (lane == 0) a.live[b] = s.sel_active;
    env_store(a.env, b, lane, s);
    // optional plan sink: what the next forward's plan_masks launch would compute from the obs this launch wrote (the
    // obs positions are (float)px, (float)py: write_obs_matrix) and the active set it just published
    if (a.env.plan_adj) {
        const uint64_t full = (n == 64) ? ~0ull : (bit(n) - 1ull);
        const float x = lane < n ? (float)s.px : 0.f, y = lane < n ? (float)s.py : 0.f;
        plan_masks_env(x, y, s.sel_active & full, a.env.plan_u1 ? 1 : -1, b, a.env.n_envs, n, lane,
                       PlanSink{a.env.plan_adj, a.env.plan_live, a.env.plan_u1, a.env.plan_u2, a.env.plan_cnt});
    }
#ifdef MEL_ENV_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long p3 = ENV_T();
    if (lane == 0 && (b & 7) == 0 && !a.first) {
        atomicAdd(&g_env_prof[0], p1 - p0), atomicAdd(&g_env_prof[1], p2 - p1), atomicAdd(&g_env_prof[2], pw);
        atomicAdd(&g_env_prof[3], ps), atomicAdd(&g_env_prof[4], po), atomicAdd(&g_env_prof[5], pr);
        atomicAdd(&g_env_prof[6], p3 - p2), atomicAdd(&g_env_prof[7], 1ull), atomicAdd(&g_env_prof[8], pit);
    }
#endif
}

struct StepArgs {
    mel_env_batch env;
    mel_episode_pool pool;
    const int32_t* actions;
    const int32_t* env_ids;
    const int32_t* episode_ids;
    int64_t n;
    mel_env_obs out;
    int has_out;
    int keep_graph;
    const int32_t* episode_table;
    int table_stride;
};

enum { OP_RESET = 0, OP_STEP = 1, OP_OBSERVE = 2 };

template <int OP>
__global__ __launch_bounds__(256) void env_kernel(StepArgs a) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.n) return;
    const int lane = lane_id();
    const int b = a.env_ids ? a.env_ids[row] : (int)row;
    Env s;
    env_load(a.env, b, lane, s);
    if (OP == OP_RESET) {
        env_reset(a.env, a.pool, b, s, a.episode_ids[row], a.keep_graph, lane);
        if (a.has_out) env_observe(a.env, b, s, a.out, row, lane);
    } else if (OP == OP_STEP) {
        env_step(a.env, a.pool, b, s, a.actions[row], lane);
        if (a.has_out) {
            const int r = env_observe(a.env, b, s, a.out, row, lane);
            if (a.episode_table) {
                if (r & 1) s.done_count += 1;
                // multi_agent_collector.py:261-264: episode over -> reset this env
                if ((r & 1) && ((r & 2) || s.done_count == a.env.n_nodes)) {
                    s.episodes_done += 1;
                    log_episode(a.env, b, s, lane);
                    if (a.pool.produced && s.ep_cursor >= a.pool.produced[b]) s.error |= MEL_ENV_ERR_EPISODE_UNDERRUN;
                    const int ep = a.episode_table[(size_t)b * a.table_stride + (s.ep_cursor % a.table_stride)];
                    env_reset(a.env, a.pool, b, s, ep, 0, lane);
                    env_observe(a.env, b, s, a.out, row, lane);
                }
            }
        }
    } else {
        env_observe(a.env, b, s, a.out, row, lane);
    }
    env_store(a.env, b, lane, s);
}

static mel_status check_env(const mel_env_batch* env, int64_t n) {
    if (!env) return fail(MEL_ERR_INVALID_ARG, "env batch is null");
    if (env->n_nodes < 1 || env->n_nodes > MEL_MAX_NODES) return fail(MEL_ERR_INVALID_ARG, "n_nodes=%d outside [1, 64]", env->n_nodes);
    if (n < 0 || n > env->n_envs) return fail(MEL_ERR_INVALID_ARG, "n=%ld outside [0, n_envs=%d]", (long)n, env->n_envs);
    if (!env->pos || !env->scalars) return fail(MEL_ERR_INVALID_ARG, "env batch is not bound (mel_env_bind)");
    if (env->log_capacity < 0 || (env->log_capacity > 0 && (!env->log_cursor || !env->log_stats || !env->log_meta)))
        return fail(MEL_ERR_INVALID_ARG, "episode log of capacity %d has null buffers", env->log_capacity);
    return MEL_OK;
}

static mel_status check_pool(const mel_env_batch* env, const mel_episode_pool* pool) {
    if (!pool) return fail(MEL_ERR_INVALID_ARG, "episode pool is null");
    if (pool->n_nodes != env->n_nodes) return fail(MEL_ERR_INVALID_ARG, "pool has %d nodes, env %d", pool->n_nodes, env->n_nodes);
    if (pool->n_episodes < 1 || !pool->origin || !pool->interested) return fail(MEL_ERR_INVALID_ARG, "empty episode pool");
    if (env->dynamic_graph && (pool->max_moves < 1 || !pool->moves)) return fail(MEL_ERR_INVALID_ARG, "dynamic graph needs movement offsets");
    return MEL_OK;
}

struct EnvLayout {
    mel_env_batch e;
    size_t bytes;
};

static EnvLayout carve_env(int32_t B, int32_t n, void* state) {
    EnvLayout L{};
    Carver c(state);
    const size_t BN = (size_t)B * n;
    L.e.n_envs = B, L.e.n_nodes = n;
    L.e.pos = c.take<double>(BN * 2);
    L.e.one_hop = c.take<uint64_t>(BN);
    L.e.two_hop = c.take<uint64_t>(BN);
    L.e.node_sets = c.take<uint64_t>((size_t)B * 8);
    L.e.sel_sets = c.take<uint64_t>((size_t)B * 4);
    L.e.scalars = c.take<int32_t>((size_t)B * MEL_ENV_SCALARS);
    L.e.agent_msgs = c.take<int32_t>(BN);
    L.e.received = c.take<int32_t>(BN);
    L.e.two_hop_cover = c.take<int32_t>(BN);
    L.e.agent_action = c.take<int8_t>(BN);
    L.e.current_actions = c.take<int8_t>(BN);
    L.e.steps_taken = c.take<int8_t>(BN);
    L.e.sel_steps = c.take<int8_t>(BN);
    L.e.rewards = c.take<double>(BN);
    L.e.pz_rewards = c.take<double>(BN);
    L.e.episode_rewards = c.take<double>(B);
    L.e.obs_matrix = c.take<float>(BN * 8);
    L.e.info_stats = c.take<double>(BN * MEL_ENV_LOGGER_STATS);
    L.bytes = c.off;
    return L;
}

}  // namespace mel

#include "episode_stream.hpp"

using namespace mel;

extern "C" {

size_t mel_env_state_bytes(int32_t n_envs, int32_t n_nodes) {
    if (n_envs < 1 || n_nodes < 1 || n_nodes > MEL_MAX_NODES) return 0;
    return carve_env(n_envs, n_nodes, nullptr).bytes;
}

mel_status mel_env_bind(mel_env_batch* env, int32_t n_envs, int32_t n_nodes, void* state) {
    if (!env || !state) return fail(MEL_ERR_INVALID_ARG, "null env/state");
    if (n_envs < 1 || n_nodes < 1 || n_nodes > MEL_MAX_NODES) return fail(MEL_ERR_INVALID_ARG, "n_envs=%d n_nodes=%d", n_envs, n_nodes);
    if (reinterpret_cast<uintptr_t>(state) & 255) return fail(MEL_ERR_INVALID_ARG, "state must be 256-byte aligned");
    const EnvLayout L = carve_env(n_envs, n_nodes, state);
    const int32_t dyn = env->dynamic_graph, hlr = env->has_local_ratio, heu = env->heuristic, tst = env->is_testing;
    const double lr = env->local_ratio;
    const mel_env_batch keep = *env;               // the caller-owned episode log survives a (re)bind
    if (heu < MEL_HEURISTIC_NONE || heu > MEL_HEURISTIC_SILENT) return fail(MEL_ERR_INVALID_ARG, "heuristic %d", heu);
    *env = L.e;
    env->dynamic_graph = dyn, env->has_local_ratio = hlr, env->local_ratio = lr, env->heuristic = heu, env->is_testing = tst;
    env->log_capacity = keep.log_capacity, env->log_cursor = keep.log_cursor, env->log_stats = keep.log_stats,
    env->log_meta = keep.log_meta;
    env->plan_adj = keep.plan_adj, env->plan_live = keep.plan_live, env->plan_u1 = keep.plan_u1, env->plan_u2 = keep.plan_u2,
    env->plan_cnt = keep.plan_cnt;
    return MEL_OK;
}

mel_status mel_env_reset(mel_env_batch* env, const mel_episode_pool* pool, const int32_t* env_ids,
                         const int32_t* episode_ids, int64_t n, int32_t keep_graph, const mel_env_obs* out,
                         void* stream) {
    if (mel_status st = check_env(env, n)) return st;
    if (mel_status st = check_pool(env, pool)) return st;
    if (!episode_ids) return fail(MEL_ERR_INVALID_ARG, "episode_ids is null");
    if (!keep_graph && (!pool->pos || !pool->one_hop)) return fail(MEL_ERR_INVALID_ARG, "pool has no graphs");
    if (n == 0) return MEL_OK;
    clear_stale_error();
    StepArgs a{};
    a.env = *env, a.pool = *pool, a.env_ids = env_ids, a.episode_ids = episode_ids, a.n = n, a.keep_graph = keep_graph;
    if (out) a.out = *out, a.has_out = 1;
    StageScope t(MEL_STAGE_ENV_RESET, static_cast<hipStream_t>(stream));
    MEL_LAUNCH(env_kernel<OP_RESET>, dim3((n + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("env_reset");
}

mel_status mel_env_step(mel_env_batch* env, const mel_episode_pool* pool, const int32_t* actions,
                        const int32_t* env_ids, int64_t n, const mel_env_obs* out, const int32_t* episode_table,
                        int32_t table_stride, void* stream) {
    if (mel_status st = check_env(env, n)) return st;
    if (mel_status st = check_pool(env, pool)) return st;
    if (!actions) return fail(MEL_ERR_INVALID_ARG, "actions is null");
    if (episode_table && (table_stride < 1 || !out)) return fail(MEL_ERR_INVALID_ARG, "auto-reset needs table_stride >= 1 and an output block");
    if (n == 0) return MEL_OK;
    clear_stale_error();
    StepArgs a{};
    a.env = *env, a.pool = *pool, a.actions = actions, a.env_ids = env_ids, a.n = n;
    a.episode_table = episode_table, a.table_stride = table_stride;
    if (out) a.out = *out, a.has_out = 1;
    StageScope t(MEL_STAGE_ENV_STEP, static_cast<hipStream_t>(stream));
    MEL_LAUNCH(env_kernel<OP_STEP>, dim3((n + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("env_step");
}

mel_status mel_env_round(mel_env_batch* env, const mel_episode_pool* pool, const int32_t* actions,
                         const int32_t* row_offsets, uint64_t* live, const int32_t* episode_table,
                         int32_t table_stride, int32_t first, uint32_t* round_counter,
                         const mel_round_replay* replay, void* stream) {
    if (mel_status st = check_env(env, env->n_envs)) return st;
    if (mel_status st = check_pool(env, pool)) return st;
    if (!live) return fail(MEL_ERR_INVALID_ARG, "live mask buffer is null");
    if (!first && (!actions || !episode_table || table_stride < 1))
        return fail(MEL_ERR_INVALID_ARG, "round step needs actions and an episode table");
    if (env->plan_adj && env->plan_u1 && (!env->plan_live || !env->plan_u2 || !env->plan_cnt))
        return fail(MEL_ERR_INVALID_ARG, "incomplete plan sink (plan_live / plan_u2 / plan_cnt)");
    clear_stale_error();
    RoundArgs a{};
    a.env = *env, a.pool = *pool, a.actions = actions, a.row_offsets = row_offsets, a.live = live;
    a.episode_table = episode_table, a.table_stride = table_stride, a.first = first, a.round_counter = round_counter;
    if (pool->snapshot) {
        const mel_env_batch* sn = pool->snapshot;
        if (sn->n_nodes != env->n_nodes || sn->n_envs < pool->n_episodes || sn->dynamic_graph != env->dynamic_graph ||
            sn->heuristic != env->heuristic || sn->is_testing != env->is_testing)
            return fail(MEL_ERR_INVALID_ARG, "reset snapshots do not fit the env batch / pool");
        if (mel_status st = check_env(sn, sn->n_envs)) return st;
        a.snap = *sn, a.has_snap = 1;
    }
    if (replay) {
        if (replay->capacity < 1 || !replay->obs || !replay->obs_next || !replay->acted || !replay->done ||
            !replay->act || !replay->rew || !replay->episode || !replay->cursor)
            return fail(MEL_ERR_INVALID_ARG, "incomplete replay block");
        a.replay = *replay;
    }
    StageScope t(MEL_STAGE_ENV_STEP, static_cast<hipStream_t>(stream));
    MEL_LAUNCH(env_round_kernel, dim3((env->n_envs + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("env_round");
}

mel_status mel_episode_refill(const mel_episode_stream* st, const mel_graph_pool* graphs, const mel_episode_pool* pool,
                              const mel_env_batch* env, int32_t max_new, int32_t discard, void* stream) {
    return launch_episode_refill(st, graphs, pool, env, max_new, discard, static_cast<hipStream_t>(stream));
}

mel_status mel_wait_counter(const uint32_t* counter, uint32_t target, uint32_t timeout_us, void* stream) {
    if (!counter) return fail(MEL_ERR_INVALID_ARG, "mel_wait_counter: counter is null");
    clear_stale_error();
    MEL_LAUNCH(wait_counter_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), counter, target, timeout_us);
    return check_launch("wait_counter");
}

// tuning builds only (-DMEL_ENV_PROF): read and reset the round kernel's cycle counters (tools/env_prof.py)
void mel_debug_world_prof(unsigned long long* out4) {
#ifdef MEL_ENV_PROF
    (void)hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_world_prof), 4 * sizeof(unsigned long long));
    unsigned long long z[4] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_world_prof), z, sizeof(z));
#else
    for (int i = 0; i < 4; ++i) out4[i] = 0;
#endif
}
void mel_debug_env_prof(unsigned long long* out9) {
#ifdef MEL_ENV_PROF
    (void)hipMemcpyFromSymbol(out9, HIP_SYMBOL(g_env_prof), 9 * sizeof(unsigned long long));
    unsigned long long z[9] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_env_prof), z, sizeof(z));
#else
    for (int i = 0; i < 9; ++i) out9[i] = 0;
#endif
}

mel_status mel_env_observe(mel_env_batch* env, const int32_t* env_ids, int64_t n, const mel_env_obs* out,
                           void* stream) {
    if (mel_status st = check_env(env, n)) return st;
    if (!out) return fail(MEL_ERR_INVALID_ARG, "output block is null");
    if (n == 0) return MEL_OK;
    clear_stale_error();
    StepArgs a{};
    a.env = *env, a.env_ids = env_ids, a.n = n, a.out = *out, a.has_out = 1;
    StageScope t(MEL_STAGE_ENV_OBSERVE, static_cast<hipStream_t>(stream));
    MEL_LAUNCH(env_kernel<OP_OBSERVE>, dim3((n + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("env_observe");
}

}  // extern "C"
