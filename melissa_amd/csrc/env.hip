// Batched message-dissemination environment for MI355X (gfx950).
//
// One wavefront steps one env; lane i is node/agent i (N <= 64); every node set is one 64-bit mask
// held wave-uniformly, so the reference's Python list/dict walks become popcounts, ballots and a
// handful of cross-lane reads.  Graphs of 65 .. 128 nodes (--n-agents 100, common.py:49) run the same code with
// W = 2: a lane holds the nodes lane and lane + 64, a node set is two words (NodeSet<W>, common.hpp).
// Integer state is bit-exact with the reference; float64 arithmetic
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

// every cross-lane read in this file has a wave-uniform source node (loop counter / selected agent)
__device__ __forceinline__ int wave_sum_i32(int v) { return wave_sum_i32_dpp(v); }

// Wave-uniform working copy of one env's masks/scalars + this lane's per-node values ([h]: node lane + 64 h).
template <int W>
struct Env {
    // uniform
    NodeSet<W> has_msg, origin_set, interested, scripted, truncated, alive, terminated, agents;
    NodeSet<W> sel_active, sel_selected, info_valid, taken_action;
    int origin, sel, skip, num_moves, world_msgs, new_round, episode, move_cursor, decisions, done_count,
        episodes_done, error, ep_cursor;
    double episode_rewards;
    // per lane (node lane + 64 h)
    double px[W], py[W], reward[W], pz_reward[W];
    NodeSet<W> one_hop[W], two_hop[W];
    int msgs[W], received[W], cover[W];
    int act[W], cur_act[W], steps[W], sel_steps[W];
    // not stored: get_info's ten values (graph.py:166-178), lane k holds value k; they only change in World.step / reset, so
    // the AEC sub-steps of a round reuse them (three wave sums and two float64 divisions once instead of per sub-step)
    double info_val;
    int info_valid_cache;
};

template <int W>
__device__ __forceinline__ void env_load(const mel_env_batch& e, int b, int lane, Env<W>& s) {
    const int n = e.n_nodes;
    // wave-uniform state: forced into SGPRs so mask arithmetic and branches run on the scalar unit
    const uint64_t* ns = e.node_sets + (size_t)b * 8 * W;
    s.has_msg = ns_uniform(ns_load<W>(ns, 0)), s.origin_set = ns_uniform(ns_load<W>(ns, 1));
    s.interested = ns_uniform(ns_load<W>(ns, 2)), s.scripted = ns_uniform(ns_load<W>(ns, 3));
    s.truncated = ns_uniform(ns_load<W>(ns, 4)), s.alive = ns_uniform(ns_load<W>(ns, 5));
    s.terminated = ns_uniform(ns_load<W>(ns, 6)), s.agents = ns_uniform(ns_load<W>(ns, 7));
    const uint64_t* ss = e.sel_sets + (size_t)b * 4 * W;
    s.sel_active = ns_uniform(ns_load<W>(ss, 0)), s.sel_selected = ns_uniform(ns_load<W>(ss, 1));
    s.info_valid = ns_uniform(ns_load<W>(ss, 2)), s.taken_action = ns_uniform(ns_load<W>(ss, 3));
    const int32_t* sc = e.scalars + (size_t)b * MEL_ENV_SCALARS;
    s.origin = uniform_i32(sc[MEL_S_ORIGIN]), s.sel = uniform_i32(sc[MEL_S_SELECTION]), s.skip = uniform_i32(sc[MEL_S_SKIP]);
    s.num_moves = uniform_i32(sc[MEL_S_NUM_MOVES]), s.world_msgs = uniform_i32(sc[MEL_S_WORLD_MSGS]);
    s.new_round = uniform_i32(sc[MEL_S_NEW_ROUND]), s.episode = uniform_i32(sc[MEL_S_EPISODE]);
    s.move_cursor = uniform_i32(sc[MEL_S_MOVE_CURSOR]), s.decisions = uniform_i32(sc[MEL_S_DECISIONS]);
    s.done_count = uniform_i32(sc[MEL_S_DONE_COUNT]), s.episodes_done = uniform_i32(sc[MEL_S_EPISODES_DONE]);
    s.error = uniform_i32(sc[MEL_S_ERROR]), s.ep_cursor = uniform_i32(sc[MEL_S_EP_CURSOR]);
    s.episode_rewards = e.episode_rewards[b];
    MEL_W_FOR(h) {
        const size_t k = (size_t)b * n + lane + 64 * h;
        const bool on = lane + 64 * h < n;
        s.px[h] = on ? e.pos[2 * k] : 0.0, s.py[h] = on ? e.pos[2 * k + 1] : 0.0;
        s.reward[h] = on ? e.rewards[k] : 0.0, s.pz_reward[h] = on ? e.pz_rewards[k] : 0.0;
        s.one_hop[h] = on ? ns_load<W>(e.one_hop, k) : ns_zero<W>(), s.two_hop[h] = on ? ns_load<W>(e.two_hop, k) : ns_zero<W>();
        s.msgs[h] = on ? e.agent_msgs[k] : 0, s.received[h] = on ? e.received[k] : 0, s.cover[h] = on ? e.two_hop_cover[k] : 0;
        s.act[h] = on ? e.agent_action[k] : NONE, s.cur_act[h] = on ? e.current_actions[k] : NONE;
        s.steps[h] = on ? e.steps_taken[k] : 0, s.sel_steps[h] = on ? e.sel_steps[k] : 0;
    }
    s.info_val = 0.0, s.info_valid_cache = 0;
}

template <int W>
__device__ __forceinline__ void env_store(const mel_env_batch& e, int b, int lane, const Env<W>& s) {
    const int n = e.n_nodes;
    if (lane == 0) {
        uint64_t* ns = e.node_sets + (size_t)b * 8 * W;
        ns_store<W>(ns, 0, s.has_msg), ns_store<W>(ns, 1, s.origin_set), ns_store<W>(ns, 2, s.interested);
        ns_store<W>(ns, 3, s.scripted), ns_store<W>(ns, 4, s.truncated), ns_store<W>(ns, 5, s.alive);
        ns_store<W>(ns, 6, s.terminated), ns_store<W>(ns, 7, s.agents);
        uint64_t* ss = e.sel_sets + (size_t)b * 4 * W;
        ns_store<W>(ss, 0, s.sel_active), ns_store<W>(ss, 1, s.sel_selected);
        ns_store<W>(ss, 2, s.info_valid), ns_store<W>(ss, 3, s.taken_action);
        int32_t* sc = e.scalars + (size_t)b * MEL_ENV_SCALARS;
        sc[MEL_S_ORIGIN] = s.origin, sc[MEL_S_SELECTION] = s.sel, sc[MEL_S_SKIP] = s.skip;
        sc[MEL_S_NUM_MOVES] = s.num_moves, sc[MEL_S_WORLD_MSGS] = s.world_msgs, sc[MEL_S_NEW_ROUND] = s.new_round;
        sc[MEL_S_EPISODE] = s.episode, sc[MEL_S_MOVE_CURSOR] = s.move_cursor, sc[MEL_S_DECISIONS] = s.decisions;
        sc[MEL_S_DONE_COUNT] = s.done_count, sc[MEL_S_EPISODES_DONE] = s.episodes_done, sc[MEL_S_ERROR] = s.error;
        sc[MEL_S_EP_CURSOR] = s.ep_cursor;
        e.episode_rewards[b] = s.episode_rewards;
    }
    MEL_W_FOR(h) {
        if (lane + 64 * h < n) {
            const size_t k = (size_t)b * n + lane + 64 * h;
            e.pos[2 * k] = s.px[h], e.pos[2 * k + 1] = s.py[h];
            e.rewards[k] = s.reward[h], e.pz_rewards[k] = s.pz_reward[h];
            ns_store<W>(e.one_hop, k, s.one_hop[h]), ns_store<W>(e.two_hop, k, s.two_hop[h]);
            e.agent_msgs[k] = s.msgs[h], e.received[k] = s.received[h], e.two_hop_cover[k] = s.cover[h];
            e.agent_action[k] = (int8_t)s.act[h], e.current_actions[k] = (int8_t)s.cur_act[h];
            e.steps_taken[k] = (int8_t)s.steps[h], e.sel_steps[k] = (int8_t)s.sel_steps[h];
        }
    }
}

// core.py:334-341: one-hop OR neighbours' one-hop, minus self
template <int W>
__device__ __forceinline__ void two_hop_of(const NodeSet<W> (&one_hop)[W], int lane, int n, NodeSet<W> (&two_hop)[W]) {
    // every lane walks ITS OWN neighbours (a handful, not all n nodes) and ORs their rows in, fetched from LDS by a
    // per-lane gather; the wave iterates max-degree times instead of n times with two v_readlane each
    __shared__ uint64_t rows[4][64 * W][W];
    const int w = (threadIdx.x >> 6) & 3;
    (void)n;
    MEL_W_FOR(h) MEL_W_FOR(k) rows[w][lane + 64 * h][k] = one_hop[h].w[k];          // (nodes >= n hold 0)
    MEL_W_FOR(h) {
        NodeSet<W> m = one_hop[h], rest = one_hop[h];
        while (__ballot(ns_any(rest))) {
            if (ns_any(rest)) {
                const int j = ns_lowest(rest);
                ns_clear_lowest(rest);
                MEL_W_FOR(k) m.w[k] |= rows[w][j][k];
            }
        }
        two_hop[h] = m & ~ns_bit<W>(lane + 64 * h);
    }
}

// nx.geometric_edges (core.py:311): edge iff dx*dx + dy*dy <= 0.2**2 in float64.
// Node j's position reaches the lanes as an LDS broadcast (every lane reads the same address) instead of four
// v_readlane into SGPRs: the loop body is then pure VALU on VGPR operands, without the SGPR write -> VALU read wait
// states that dominated it with one wavefront per SIMD (workgroups of the env kernels are 4 wavefronts).
template <int W>
__device__ __forceinline__ void geometric_one_hop(const double (&px)[W], const double (&py)[W], int lane, int n,
                                                  NodeSet<W> (&one_hop)[W]) {
    __shared__ double sx[4][64 * W], sy[4][64 * W];
    const int w = (threadIdx.x >> 6) & 3;
    MEL_W_FOR(h) sx[w][lane + 64 * h] = px[h], sy[w][lane + 64 * h] = py[h];
    NodeSet<W> m[W];
    MEL_W_FOR(h) m[h] = ns_zero<W>();
    MEL_W_FOR(k) {
        const int cnt = n - 64 * k < 64 ? n - 64 * k : 64;
#pragma unroll 5
        for (int jj = 0; jj < cnt; ++jj) {
            const double xj = sx[w][64 * k + jj], yj = sy[w][64 * k + jj];
            MEL_W_FOR(h) {
                const double dx = px[h] - xj, dy = py[h] - yj;
                const double d2 = dx * dx + dy * dy;
                if (d2 <= R2_F64) m[h].w[k] |= 1ull << jj;
            }
        }
    }
    MEL_W_FOR(h) one_hop[h] = (lane + 64 * h < n) ? (m[h] & ~ns_bit<W>(lane + 64 * h)) : ns_zero<W>();
}

// selector.py:25-34
template <int W>
__device__ __forceinline__ int selector_next(Env<W>& s, int lane) {
    const NodeSet<W> cand = s.sel_active & ~s.sel_selected;
    if (!ns_any(cand)) return NONE;
    const int i = ns_lowest(cand);
    MEL_W_FOR(h) if (lane + 64 * h == i) s.sel_steps[h] += 1;
    s.sel_selected |= ns_bit<W>(i);
    return i;
}
// selector.py:43-44
template <int W>
__device__ __forceinline__ void selector_enable(Env<W>& s, const NodeSet<W>& agents, int lane) {
    (void)lane;
    NodeSet<W> can;
    MEL_W_FOR(h) can.w[h] = __ballot(s.sel_steps[h] < MAX_AGENT_STEPS);
    s.sel_active = (s.sel_active & ~agents) | (agents & can);
}

#ifdef MEL_ENV_PROF
// finer split of the world step (wave 0 lane 0 of every eighth env adds): [0] relay + scripted, [1] move + all-pairs edges,
// [2] two-hop masks, [3] calls
__device__ unsigned long long g_world_prof[4];
#endif

// World.step core.py:225-266
template <int W>
__device__ __forceinline__ void world_step(const mel_env_batch& e, const mel_episode_pool& pool, Env<W>& s,
                                           int lane) {
    const int n = e.n_nodes;
#ifdef MEL_ENV_PROF
    const unsigned long long wp0 = __builtin_readcyclecounter();
#endif
    s.info_valid_cache = 0;                                  // message counters, coverage and (dynamic graph) degrees change here
    // :226-234 scripted agents: action = heuristic(agent) (the heuristics offered return no relay mask, so the
    // relays_for pass :236-243 never fires)
    // (no heuristic: Agent.action_callback stays None, World.scripted_agents is empty, nothing is overridden)
    bool scripted_lane[W];
    MEL_W_FOR(h) {
        scripted_lane[h] = e.heuristic != MEL_HEURISTIC_NONE && lane + 64 * h < n && ns_mine(s.scripted, lane, h);
        if (scripted_lane[h]) {
            if (e.heuristic == MEL_HEURISTIC_SIMPLE_BROADCAST) s.act[h] = ns_mine(s.taken_action, lane, h) ? 0 : 1;
            // Agent.number_interested_neighbors: counted at reset (core.py:401) but zeroed again by agent.reset() ->
            // Agent.__init__ (:412-416, :68); only move_graph recomputes it (:286-287), so it is 0 until the episode's
            // first move and tracks the current graph afterwards (one_hop changes only in moves)
            else if (e.heuristic == MEL_HEURISTIC_BROADCAST_IF_INTERESTED)
                s.act[h] = (e.dynamic_graph && s.move_cursor > 0 && ns_any(s.one_hop[h] & s.interested)) ? 1 : 0;
            else s.act[h] = 0;                                       // silent
        }
    }
    // :246 the source always transmits on its first opportunity
    const int origin_msgs = node_i32<W>(s.msgs, s.origin);
    MEL_W_FOR(h) if (lane + 64 * h == s.origin && origin_msgs == 0) s.act[h] = 1;
    // :249-254 relay in id order; has_message is re-read at each agent's turn
    NodeSet<W> cand;
    MEL_W_FOR(h) cand.w[h] = __ballot(lane + 64 * h < n && s.act[h] != NONE && s.act[h] != 0);
    while (ns_any(cand)) {
        const int i = ns_lowest(cand);
        ns_clear_lowest(cand);
        if (ns_test(s.has_msg, i)) {                         // relay_message core.py:268-279
            const NodeSet<W> nb = node_set<W>(s.one_hop, i);
            s.world_msgs += 1;
            MEL_W_FOR(h) if (lane + 64 * h == i) s.msgs[h] += 1;
            s.taken_action |= ns_bit<W>(i);
            MEL_W_FOR(h) s.received[h] += (int)ns_mine(nb, lane, h);
            s.has_msg |= nb;
        }
    }
#ifdef MEL_ENV_PROF
    asm volatile("s_nop 0" ::"s"(s.has_msg.w[0]), "v"(s.received[0]));
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
        const double* off = pool.moves + ((size_t)s.episode * pool.max_moves + mv) * 2 * n;
        MEL_W_FOR(h) {
            if (lane + 64 * h < n) {
                s.px[h] = s.px[h] + off[lane + 64 * h];
                s.py[h] = s.py[h] + off[n + lane + 64 * h];
            }
        }
        s.move_cursor += 1;
        geometric_one_hop<W>(s.px, s.py, lane, n, s.one_hop);
#ifdef MEL_ENV_PROF
        asm volatile("s_nop 0" ::"v"(s.one_hop[0].w[0]));
        wp2 = __builtin_readcyclecounter();
#endif
        two_hop_of<W>(s.one_hop, lane, n, s.two_hop);
#ifdef MEL_ENV_PROF
        asm volatile("s_nop 0" ::"v"(s.two_hop[0].w[0]));
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
    MEL_W_FOR(h) {
        s.cover[h] = ns_count(s.two_hop[h] & (s.has_msg | s.origin_set));
        if (scripted_lane[h]) s.act[h] = 0;                          // :264-266
    }
}

// graph.py:254-271 (copy: optional second destination, the replay's obs_next slot)
template <int W>
__device__ __forceinline__ void write_obs_matrix(const mel_env_batch& e, int b, const Env<W>& s, int lane,
                                                 float* copy = nullptr) {
    MEL_W_FOR(h) {
        const int node = lane + 64 * h;
        if (node >= e.n_nodes) continue;
        float4* row = reinterpret_cast<float4*>(e.obs_matrix + ((size_t)b * e.n_nodes + node) * 8);
        const float act = (s.act[h] != NONE) ? (float)s.act[h] : 0.f;
        const float interested = ns_mine(s.interested, lane, h) ? 1.f : 0.f;
        const float has = ns_mine(s.has_msg | s.origin_set, lane, h) ? 1.f : 0.f;
        const float dm = ns_mine(s.scripted, lane, h) ? 0.f : 1.f;
        row[0] = make_float4((float)s.px[h], (float)s.py[h], (float)ns_count(s.one_hop[h]), (float)s.msgs[h]);
        row[1] = make_float4(act, interested, has, dm);
        if (copy) {
            float4* c = reinterpret_cast<float4*>(copy + (size_t)node * 8);
            c[0] = row[0];
            c[1] = row[1];
        }
    }
}

// graph.py:402-463, float64, same operation order (h: the lane's node lane + 64 h)
template <int W>
__device__ __forceinline__ double agent_reward(const Env<W>& s, int h) {
    const NodeSet<W> covered = s.has_msg | s.origin_set;
    const int total = ns_count(s.two_hop[h] & s.interested);
    const int cov = ns_count(s.two_hop[h] & s.interested & covered);
    double reward = total > 0 ? (double)cov / (double)total : 0.0;
    const int deg = ns_count(s.one_hop[h]);
    if (s.act[h] != NONE && s.act[h] != 0) {
        const double pen_unint = deg > 0 ? (double)ns_count(s.one_hop[h] & ~s.interested) / (double)deg : 0.0;
        const double pen_cov = deg > 0 ? (double)ns_count(s.one_hop[h] & s.has_msg) / (double)deg : 0.0;
        const double penalty = pen_unint + pen_cov;
        reward -= penalty;
    } else {
        const int one_int = ns_count(s.one_hop[h] & s.interested);
        const int unc = ns_count(s.one_hop[h] & s.interested & ~s.has_msg & ~s.origin_set);
        if (unc > 0) reward -= (double)unc / (double)one_int;
    }
    return reward;
}

// graph.py:149-179 -> infos[agent]['logger_stats'] (10 float64 in dict order)
template <int W>
__device__ __forceinline__ void write_info_stats(const mel_env_batch& e, int b, int agent, Env<W>& s, int lane) {
    const int n = e.n_nodes;
    if (!s.info_valid_cache) {
        int my_sent = 0, my_recv = 0, my_nbrs = 0;
        MEL_W_FOR(h) {
            const bool on = lane + 64 * h < n;
            my_sent += on ? s.msgs[h] : 0, my_recv += on ? s.received[h] : 0, my_nbrs += on ? ns_count(s.one_hop[h]) : 0;
        }
        const int sent = wave_sum_i32(my_sent);
        const int recv = wave_sum_i32(my_recv);
        const int nbrs = wave_sum_i32(my_nbrs);
        const int n_int = ns_count(s.interested);
        const int cov_int = ns_count(s.has_msg & s.interested);
        const double st1 = (double)ns_count(s.has_msg) / (double)n;
        const double st6 = n_int > 0 ? (double)cov_int / (double)n_int : 0.0;
        double v = (double)s.world_msgs;                                   // lane 0
        v = lane == 1 ? st1 : v;
        v = lane == 2 ? (double)sent : v;
        v = lane == 3 ? (double)recv : v;
        v = lane == 4 ? (double)nbrs : v;
        v = lane == 5 ? (double)n_int : v;
        v = lane == 6 ? st6 : v;
        v = lane == 7 ? (double)cov_int : v;
        v = lane == 8 ? (double)ns_count(s.has_msg & ~s.interested) : v;
        v = lane == 9 ? s.episode_rewards : v;
        s.info_val = v;
        s.info_valid_cache = 1;
    }
    if (lane < MEL_ENV_LOGGER_STATS) e.info_stats[((size_t)b * n + agent) * MEL_ENV_LOGGER_STATS + lane] = s.info_val;
}

// one row of the episode log: the logger_stats of the final observation of the episode that just ended
template <int W>
__device__ __forceinline__ void log_episode(const mel_env_batch& e, int b, const Env<W>& s, int lane) {
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
        const bool valid = s.sel >= 0 && ns_test(s.info_valid, a);
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
template <int W>
__device__ __forceinline__ bool env_step(const mel_env_batch& e, const mel_episode_pool& pool, int b, Env<W>& s,
                                         int action, int lane, float* obs_next_copy = nullptr) {
    const int n = e.n_nodes;
    const int a = s.sel;
    bool world = false;
    if (a < 0) {                      // reference would raise KeyError; flag it and stay put
        s.error |= 2;
        return false;
    }
    if (ns_test(s.terminated, a)) {                                     // :304-310 dead step
        const NodeSet<W> not_a = ~ns_bit<W>(a);
        s.sel_active &= not_a;
        s.alive &= not_a;                                               // _was_dead_step :274-301
        s.terminated &= not_a;
        s.info_valid &= not_a;
        s.agents &= not_a;
        const NodeSet<W> dead = s.agents & s.terminated;
        if (ns_any(dead)) {
            if (s.skip == SKIP_NONE) s.skip = a;
            s.sel = ns_lowest(dead);
        } else {
            if (s.skip != SKIP_NONE) s.sel = s.skip;
            s.skip = SKIP_NONE;
        }
    } else {
        s.decisions += 1;
        MEL_W_FOR(h) {
            if (lane + 64 * h == a) {
                s.cur_act[h] = action;                                  // :314
                s.steps[h] += 1;                                        // :316-318
            }
        }
        s.sel = selector_next(s, lane);                                 // :321
        if (s.sel == NONE) {                                            // :324 round complete
            world = true;
            MEL_W_FOR(h) {
                if (ns_mine(s.alive, lane, h)) s.reward[h] = 0.0;       // _clear_rewards
                s.act[h] = s.cur_act[h];                                // :362-365
            }
            world_step(e, pool, s, lane);
            write_obs_matrix(e, b, s, lane, obs_next_copy);             // :370-371
            double r[W];
            MEL_W_FOR(h) {
                const double r0 = agent_reward(s, h);
                r[h] = r0;
                if (e.has_local_ratio) r[h] = 0.0 * (1.0 - e.local_ratio) + r0 * e.local_ratio;   // :380-384
                if (ns_mine(s.agents, lane, h)) s.reward[h] = r[h];     // :386
            }
            NodeSet<W> rest = s.agents;                                 // :387 summed in id order
            while (ns_any(rest)) {
                const int i = ns_lowest(rest);
                ns_clear_lowest(rest);
                s.episode_rewards += node_f64<W>(r, i);
            }
            s.num_moves += 1;                                           // :328
            NodeSet<W> expire;
            MEL_W_FOR(h) expire.w[h] = __ballot(lane + 64 * h < n && s.steps[h] >= MAX_AGENT_STEPS);
            expire = expire & s.agents & ~s.truncated;
            s.truncated |= expire;                                      // :330-334
            s.terminated |= expire;
            s.agents = s.has_msg & s.alive;                             // :336-341
            if (!e.is_testing) s.agents &= ~s.scripted;
            selector_enable(s, s.agents, lane);                         // :342
            s.sel_selected = ns_zero<W>();                              // :343
            s.new_round = 1;                                            // :344
            s.sel = selector_next(s, lane);                             // :345
            MEL_W_FOR(h) s.cur_act[h] = NONE;                           // :347
        }
        if (s.sel != NONE) {                                            // :358
            write_info_stats(e, b, s.sel, s, lane);
            s.info_valid |= ns_bit<W>(s.sel);
        }
        const NodeSet<W> dead = s.agents & s.terminated;                // :359 _deads_step_first
        if (ns_any(dead)) {
            s.skip = s.sel;
            s.sel = ns_lowest(dead);
        }
    }
    MEL_W_FOR(h) if (ns_mine(s.alive, lane, h)) s.pz_reward[h] = s.reward[h];   // PettingZooEnv.step (A.6)
    return world;
}

// GraphEnv.reset + World.reset, graph.py:222-248 / core.py:343-437, from a pre-sampled pool episode
template <int W>
__device__ __forceinline__ void env_reset(const mel_env_batch& e, const mel_episode_pool& pool, int b, Env<W>& s,
                                          int episode, int keep_graph, int lane) {
    const int n = e.n_nodes;
    const NodeSet<W> full = ns_full<W>(n);
    s.episode = episode;
    s.move_cursor = 0;
    if (!keep_graph) {
        MEL_W_FOR(h) {
            const size_t k = (size_t)episode * n + lane + 64 * h;
            const bool on = lane + 64 * h < n;
            s.px[h] = on ? pool.pos[2 * k] : 0.0;
            s.py[h] = on ? pool.pos[2 * k + 1] : 0.0;
            s.one_hop[h] = on ? ns_load<W>(pool.one_hop, k) : ns_zero<W>();
        }
    }
    two_hop_of<W>(s.one_hop, lane, n, s.two_hop);                       // core.py:421
    s.origin = uniform_i32(pool.origin[episode]);
    s.interested = ns_uniform(ns_load<W>(pool.interested, episode)) & full;
    s.scripted = pool.scripted ? (ns_uniform(ns_load<W>(pool.scripted, episode)) & full) : ns_zero<W>();   // core.py:395,404
    s.world_msgs = 0;
    s.has_msg = ns_bit<W>(s.origin);                                    // :432-434
    s.origin_set = ns_bit<W>(s.origin);
    s.taken_action = ns_zero<W>();
    s.truncated = ns_zero<W>();
    MEL_W_FOR(h) {
        s.msgs[h] = 0, s.received[h] = 0, s.cover[h] = 0;
        s.act[h] = NONE;
        s.steps[h] = (lane + 64 * h == s.origin) ? 1 : 0;               // :424,435
    }
    world_step(e, pool, s, lane);                                       // :437
    // GraphEnv.reset
    MEL_W_FOR(h) {
        s.sel_steps[h] = (lane + 64 * h == s.origin) ? 1 : 0;           // selector.reinit + enable(on_reset)
        s.reward[h] = 0.0;
    }
    s.sel_active = ns_zero<W>(), s.sel_selected = ns_zero<W>();
    s.alive = full, s.terminated = ns_zero<W>(), s.info_valid = ns_zero<W>();
    s.num_moves = 0;
    s.episode_rewards = 0.0;
    s.done_count = 0;
    write_obs_matrix(e, b, s, lane);
    s.agents = s.has_msg;                                               // :242-245
    if (!e.is_testing) s.agents &= ~s.scripted;
    selector_enable(s, s.agents, lane);
    s.sel = selector_next(s, lane);                                     // :247
    s.skip = SKIP_NONE;
    MEL_W_FOR(h) s.cur_act[h] = NONE;                                   // :248
    s.ep_cursor += 1;
}

// GraphEnv.observe / last() graph.py:181-216 + PettingZooEnv packing; returns bit0 terminated,
// bit1 explicit_reset, bit2 environment_step
template <int W>
__device__ __forceinline__ int env_observe(const mel_env_batch& e, int b, Env<W>& s, const mel_env_obs& o,
                                           int64_t row, int lane) {
    const int n = e.n_nodes;
    const int a = s.sel;
    const int valid = a >= 0;
    const int dead = valid ? (int)ns_test(s.terminated, a) : 0;
    int explicit_reset = 0, environment_step = 0;
    if (ns_count(s.agents) == 1 && !ns_any(s.agents & ~s.terminated)) {   // :205-207
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
    if (o.rew) MEL_W_FOR(h) if (lane + 64 * h < n) o.rew[row * n + lane + 64 * h] = s.pz_reward[h];
    if (o.stats && lane < MEL_ENV_LOGGER_STATS) {
        const int ok = valid && ns_test(s.info_valid, a);
        o.stats[row * MEL_ENV_LOGGER_STATS + lane] =
            ok ? e.info_stats[((size_t)b * n + a) * MEL_ENV_LOGGER_STATS + lane] : 0.0;
    }
    const NodeSet<W> nb = node_set<W>(s.one_hop, valid ? a : 0);
    if (lane == 0) {
        if (o.agent_id) o.agent_id[row] = a;
        if (o.action_mask) o.action_mask[2 * row] = o.action_mask[2 * row + 1] = dead ? 0 : 1;   // :190-192
        if (o.terminated) o.terminated[row] = (uint8_t)dead;
        if (o.flags) {
            o.flags[4 * row + 0] = s.num_moves;
            o.flags[4 * row + 1] = environment_step;
            o.flags[4 * row + 2] = explicit_reset;
            o.flags[4 * row + 3] = valid ? (int)ns_test(s.info_valid, a) : 0;
        }
        if (o.active_nb) ns_store<W>(o.active_nb, row, valid ? (nb & ~(s.truncated & ~s.agents)) : ns_zero<W>());   // :198-203
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
    uint64_t* live;                // [B] node sets; in: the active set the actions were computed for; out: next round's
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
template <int W>
__device__ __forceinline__ void env_reset_from_snapshot(const mel_env_batch& e, const mel_env_batch& snap, int b, Env<W>& s,
                                                        int episode, int lane) {
    Env<W> t;
    env_load(snap, episode, lane, t);
    t.new_round = s.new_round, t.decisions = s.decisions, t.episodes_done = s.episodes_done;
    t.error = s.error | t.error, t.ep_cursor = s.ep_cursor + 1;
    MEL_W_FOR(h) t.pz_reward[h] = s.pz_reward[h];          // the sticky Tianshou reward vector survives a reset
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

template <int W>
__global__ __launch_bounds__(256) void env_round_kernel(RoundArgs a) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= a.env.n_envs) return;
    const int lane = lane_id();
    const int n = a.env.n_nodes;
    if (a.round_counter && b == 0 && lane == 0 && !a.first) atomicAdd(a.round_counter, 1u);
    Env<W> s;
#ifdef MEL_ENV_PROF
    unsigned long long pw = 0, ps = 0, po = 0, pr = 0, pit = 0;
    const unsigned long long p0 = ENV_T();
#endif
    env_load(a.env, b, lane, s);
#ifdef MEL_ENV_PROF
    ENV_MARK(s.sel);
    asm volatile("s_nop 0" ::"v"(s.px[0]), "v"(s.steps[0]));
    const unsigned long long p1 = ENV_T();
#endif
    const mel_env_obs none{};
    if (!a.first) {
        const NodeSet<W> live_in = ns_uniform(ns_load<W>(a.live, b));
        // every agent's action in one coalesced load (lane = agent), read back with v_readlane; actions are
        // either packed rows (row_offsets) or the dense [B, N] layout
        int my_action[W];
        MEL_W_FOR(h) {
            my_action[h] = 0;
            if (ns_mine(live_in, lane, h))
                my_action[h] = a.row_offsets ? a.actions[uniform_i32(a.row_offsets[b]) + ns_rank_below(live_in, lane + 64 * h)]
                                             : a.actions[(size_t)b * n + lane + 64 * h];
        }
        // replay record of this round (only envs with acting agents): pre-state now, outcome after the world step
        float* rec_next = nullptr;
        size_t rec = 0;
        if (a.replay.capacity > 0 && ns_any(live_in)) {
            const int cur = uniform_i32(a.replay.cursor[b]);
            rec = (size_t)b * a.replay.capacity + (cur % a.replay.capacity);
            const float* src = a.env.obs_matrix + (size_t)b * n * 8;
            float* dst = a.replay.obs + rec * n * 8;
            for (int c = lane; c < n * 8; c += 64) dst[c] = src[c];
            MEL_W_FOR(h) if (lane + 64 * h < n) a.replay.act[rec * n + lane + 64 * h] = (int8_t)my_action[h];
            if (lane == 0) {
                ns_store<W>(a.replay.acted, rec, live_in);
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
            const NodeSet<W> dead = s.agents & s.terminated;
            if (ns_any(dead) && s.skip >= 0 && s.sel == ns_lowest(dead) && ns_any(s.agents & ~dead) &&
                s.done_count + ns_count(dead) - 1 < n && s.error == 0) {
                const NodeSet<W> keep = ~dead;
                s.sel_active &= keep, s.alive &= keep, s.terminated &= keep, s.info_valid &= keep, s.agents &= keep;
                s.done_count += ns_count(dead) - 1;
                s.sel = s.skip;
                s.skip = SKIP_NONE;
                MEL_W_FOR(h) if (ns_mine(s.alive, lane, h)) s.pz_reward[h] = s.reward[h];   // PettingZooEnv.step (A.6), idempotent
            }
        }
        // (2) No pending dead agent, the selection is the first active agent and the forward covered exactly the active
        //     set: the sub-steps of every acting agent but the LAST change nothing but bookkeeping - its action and step
        //     count, the selector's marks, the stored info of the next selection (the same ten values: nothing moves
        //     before the world step) - so they are applied in one go (graph.py:312-321,358, k - 1 times); the last agent
        //     takes the sequential path below, which runs the world step.
        if (s.skip == SKIP_NONE && !ns_any(s.agents & s.terminated) && ns_any(live_in) && live_in == s.sel_active &&
            !ns_any(live_in & ~s.agents) && s.sel >= 0 && s.sel == ns_lowest(live_in) && s.sel_selected == ns_bit<W>(s.sel) &&
            s.error == 0) {
            const int last = ns_highest(live_in);
            const NodeSet<W> early = live_in & ~ns_bit<W>(last);     // agents whose sub-step is fast-forwarded
            if (ns_any(early)) {
                s.decisions += ns_count(early);
                const NodeSet<W> later = live_in & ~ns_bit<W>(s.sel);    // every agent that becomes the selection: a_2 .. a_k
                MEL_W_FOR(h) {
                    if (ns_mine(early, lane, h)) {                   // :314-318
                        s.cur_act[h] = my_action[h];
                        s.steps[h] += 1;
                    }
                    if (ns_mine(later, lane, h)) s.sel_steps[h] += 1;    // selector.py:25-34
                }
                s.sel_selected = live_in;
                NodeSet<W> rest = later;                             // :358 infos[next] = get_info()
                while (ns_any(rest)) {
                    const int nxt = ns_lowest(rest);
                    ns_clear_lowest(rest);
                    write_info_stats(a.env, b, nxt, s, lane);
                }
                s.info_valid |= later;
                MEL_W_FOR(h) if (ns_mine(s.alive, lane, h)) s.pz_reward[h] = s.reward[h];
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
            if (!ns_test(s.terminated, sel)) {
                if (!ns_test(live_in, sel)) {              // an agent acts that the forward did not cover
                    s.error |= 4;
                    break;
                }
                action = node_i32<W>(my_action, sel);
            }
#ifdef MEL_ENV_PROF
            const unsigned long long q0 = ENV_T();
#endif
            const bool world = env_step(a.env, a.pool, b, s, action, lane, rec_next);
#ifdef MEL_ENV_PROF
            ENV_MARK(s.sel);
            asm volatile("s_nop 0" ::"v"(s.reward[0]), "v"(s.one_hop[0].w[0]));
            const unsigned long long q1 = ENV_T();
#endif
            const int r = env_observe(a.env, b, s, none, 0, lane);
#ifdef MEL_ENV_PROF
            ENV_MARK(r);
            const unsigned long long q2 = ENV_T();
            (world ? pw : ps) += q1 - q0, po += q2 - q1, pit += 1;
#endif
            if (world && rec_next) {          // the world step just ran: rewards / terminations of this round
                MEL_W_FOR(h) if (lane + 64 * h < n)
                    a.replay.rew[rec * n + lane + 64 * h] = ns_mine(live_in, lane, h) ? (float)s.reward[h] : 0.f;
                if (lane == 0) ns_store<W>(a.replay.done, rec, s.terminated & live_in);
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
                asm volatile("s_nop 0" ::"v"(s.px[0]), "v"(s.one_hop[0].w[0]));
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
    if (lane == 0) ns_store<W>(a.live, b, s.sel_active);
    env_store(a.env, b, lane, s);
    // optional plan sink: what the next forward's plan_masks launch would compute from the obs this launch wrote (the
    // obs positions are (float)px, (float)py: write_obs_matrix) and the active set it just published
    if (a.env.plan_adj) {
        float x[W], y[W];
        MEL_W_FOR(h) x[h] = lane + 64 * h < n ? (float)s.px[h] : 0.f, y[h] = lane + 64 * h < n ? (float)s.py[h] : 0.f;
        plan_masks_env<W>(x, y, s.sel_active & ns_full<W>(n), a.env.plan_u1 ? 1 : -1, b, a.env.n_envs, n, lane,
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

template <int OP, int W>
__global__ __launch_bounds__(256) void env_kernel(StepArgs a) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.n) return;
    const int lane = lane_id();
    const int b = a.env_ids ? a.env_ids[row] : (int)row;
    Env<W> s;
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
    if (env->n_nodes < 1 || env->n_nodes > MEL_MAX_NODES) return fail(MEL_ERR_INVALID_ARG, "n_nodes=%d outside [1, %d]", env->n_nodes, MEL_MAX_NODES);
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
    const size_t SW = (size_t)set_words(n);            // words per node set
    L.e.n_envs = B, L.e.n_nodes = n;
    L.e.pos = c.take<double>(BN * 2);
    L.e.one_hop = c.take<uint64_t>(BN * SW);
    L.e.two_hop = c.take<uint64_t>(BN * SW);
    L.e.node_sets = c.take<uint64_t>((size_t)B * 8 * SW);
    L.e.sel_sets = c.take<uint64_t>((size_t)B * 4 * SW);
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
    if (env->n_nodes > 64) MEL_LAUNCH((env_kernel<OP_RESET, 2>), dim3((n + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    else MEL_LAUNCH((env_kernel<OP_RESET, 1>), dim3((n + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), a);
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
    if (env->n_nodes > 64) MEL_LAUNCH((env_kernel<OP_STEP, 2>), dim3((n + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    else MEL_LAUNCH((env_kernel<OP_STEP, 1>), dim3((n + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), a);
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
    if (env->n_nodes > 64) MEL_LAUNCH(env_round_kernel<2>, dim3((env->n_envs + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    else MEL_LAUNCH(env_round_kernel<1>, dim3((env->n_envs + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("env_round");
}

// ---- replay sampling: one launch instead of ~150 small index launches (replay.RoundReplay.sample) ---------------------------------
// Uniform over (record, acting agent) pairs with replacement, then the n-step walk of [3P] tianshou's compute_nstep_return as the
// reference configures it (l_dgn.py:246-261, estimation_step): follow the agent through consecutive ring slots while the
// episode is the same and it keeps acting.  One 1 024-thread workgroup: (1) per-record pair counts and their exclusive prefix
// sums (chunks of 1 024 consecutive records: coalesced loads, shuffle scans), (2) one thread per sample: a counter-based draw (splitmix64 of seed, the device-side draw counter, the sample index),
// binary search in the prefix sums, the rank-th member of the record's acted set, the walk, (3) all threads copy the sampled
// observation rows.  No host value enters but the seed: replayable from a HIP graph, the draw counter advances on the device.
struct ReplaySampleArgs {
    mel_round_replay rp;
    mel_replay_batch out;
    int B, n, W, batch, n_step;
    float disc[MEL_REPLAY_MAX_NSTEP + 1];
    unsigned long long seed;
    unsigned long long* counter;
    int* prefix;            // [B * K + 1]
};

__device__ __forceinline__ int replay_pairs(const ReplaySampleArgs& a, int rec) {
    const int K = a.rp.capacity, e = rec / K, k = rec - e * K;
    const int filled = min(a.rp.cursor[e], K);
    if (k >= filled) return 0;
    int c = 0;
    for (int w = 0; w < a.W; ++w) c += __popcll(a.rp.acted[(size_t)rec * a.W + w]);
    return c;
}

__global__ __launch_bounds__(1024) void replay_sample_kernel(ReplaySampleArgs a) {
    __shared__ int wave_tot[16];
    __shared__ int s_env[1024], s_slot[1024], s_agent[1024], s_boot[1024];
    const int tid = threadIdx.x, K = a.rp.capacity, BK = a.B * K;
    const int lane = tid & 63, wv = tid >> 6;
    // (1) pair counts of consecutive records by consecutive threads (coalesced, sixteen chunks of 1 024 records in flight per
    // thread), then per chunk an exclusive scan: inside the wave by shuffles, across the 16 waves through LDS
    int run = 0;
    for (int c0 = 0; c0 < BK; c0 += 16 * 1024) {
        int cnt[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int i = c0 + j * 1024 + tid;
            cnt[j] = i < BK ? replay_pairs(a, i) : 0;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (c0 + j * 1024 >= BK) break;                 // (uniform)
            int incl = cnt[j];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(incl, d);
                if (lane >= d) incl += up;
            }
            if (lane == 63) wave_tot[wv] = incl;
            __syncthreads();
            int before = 0, total_c = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int t = wave_tot[k];
                before += k < wv ? t : 0;
                total_c += t;
            }
            const int i = c0 + j * 1024 + tid;
            if (i < BK) a.prefix[i] = run + before + incl - cnt[j];
            run += total_c;
            __syncthreads();
        }
    }
    const int total = run;
    if (tid == 0) a.prefix[BK] = total;
    __threadfence_block();
    __syncthreads();
    const unsigned long long draw = *a.counter;
    if (tid < a.batch) {
        int e = 0, k = 0, agent = 0, boot = 0;
        float ret = 0.f, bw = 1.f;
        if (total > 0) {
            unsigned long long z = a.seed + draw * 0x9E3779B97F4A7C15ull + (unsigned long long)(tid + 1) * 0xD1B54A32D192ED03ull;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            const int pick = (int)(((z >> 32) * (unsigned long long)total) >> 32);       // uniform in [0, total)
            int l = 0, h = BK;                             // last record whose prefix is <= pick (records without pairs share
            while (h - l > 1) {                            // their successor's prefix and are skipped)
                const int m = (l + h) >> 1;
                if (a.prefix[m] <= pick) l = m;
                else h = m;
            }
            int rank = pick - a.prefix[l];
            e = l / K, k = l - e * K;
            for (int w = 0; w < a.W; ++w) {                // the rank-th acting agent in id order
                unsigned long long m = a.rp.acted[(size_t)l * a.W + w];
                const int c = __popcll(m);
                if (rank >= c) { rank -= c; continue; }
                for (int r = 0; r < rank; ++r) m &= m - 1;
                agent = 64 * w + __ffsll((long long)m) - 1;
                break;
            }
            // the n-step walk (replay.RoundReplay.sample's loop, same operation order in fp32)
            const int filled = min(a.rp.cursor[e], K), newest = (int)(((long long)a.rp.cursor[e] - 1) % K);
            const int ep0 = a.rp.episode[(size_t)e * K + k];
            const int aw = agent >> 6;
            const unsigned long long abit = 1ull << (agent & 63);
            bool alive = true;
            int kk = k;
            boot = k;
            for (int j = 0; j < a.n_step; ++j) {
                const size_t rec = (size_t)e * K + kk;
                const bool ok = alive && a.rp.episode[rec] == ep0 && (a.rp.acted[rec * a.W + aw] & abit) && kk < filled;
                if (ok) {
                    ret = ret + a.disc[j] * a.rp.rew[rec * a.n + agent];
                    boot = kk;
                    bw = a.disc[j + 1];
                }
                const bool finished = ok && (a.rp.done[rec * a.W + aw] & abit);
                if (finished) bw = 0.f;
                alive = ok && !finished && kk != newest;
                kk = kk + 1 == K ? 0 : kk + 1;
            }
        }
        s_env[tid] = e, s_slot[tid] = k, s_agent[tid] = agent, s_boot[tid] = boot;
        a.out.env[tid] = e, a.out.slot[tid] = k, a.out.agent[tid] = agent;
        a.out.act[tid] = (long long)a.rp.act[((size_t)e * K + k) * a.n + agent];
        a.out.ret[tid] = ret, a.out.boot_w[tid] = bw;
    }
    __syncthreads();
    const int width = 8 * a.n, row = width + 1;
    for (int i = tid; i < a.batch * row; i += 1024) {
        const int sidx = i / row, c = i - sidx * row;
        const size_t base = (size_t)s_env[sidx] * K;
        a.out.obs[i] = c < width ? a.rp.obs[(base + s_slot[sidx]) * width + c] : (float)s_agent[sidx];
        a.out.boot_obs[i] = c < width ? a.rp.obs_next[(base + s_boot[sidx]) * width + c] : (float)s_agent[sidx];
    }
    if (tid == 0) *a.counter = draw + 1;
}

mel_status mel_replay_sample(const mel_round_replay* replay, int64_t n_envs, int32_t n_nodes, int32_t batch, int32_t n_step,
                             const float* discount, uint64_t seed, uint64_t* draw_counter, int32_t* scratch,
                             const mel_replay_batch* out, void* stream) {
    if (!replay || !discount || !draw_counter || !scratch || !out) return fail(MEL_ERR_INVALID_ARG, "mel_replay_sample: null argument");
    if (replay->capacity < 1 || !replay->obs || !replay->obs_next || !replay->acted || !replay->done || !replay->act || !replay->rew ||
        !replay->episode || !replay->cursor)
        return fail(MEL_ERR_INVALID_ARG, "incomplete replay block");
    if (!out->obs || !out->boot_obs || !out->act || !out->ret || !out->boot_w || !out->env || !out->slot || !out->agent)
        return fail(MEL_ERR_INVALID_ARG, "incomplete replay batch");
    if (n_envs < 1 || n_nodes < 1 || n_nodes > MEL_MAX_NODES || batch < 1 || batch > 1024 || n_step < 1 || n_step > MEL_REPLAY_MAX_NSTEP ||
        n_envs * (int64_t)replay->capacity > (1ll << 24))
        return fail(MEL_ERR_INVALID_ARG, "mel_replay_sample: batch in [1, 1024], n_step in [1, %d], n_envs * capacity <= 2^24", MEL_REPLAY_MAX_NSTEP);
    clear_stale_error();
    ReplaySampleArgs a{};
    a.rp = *replay, a.out = *out, a.B = (int)n_envs, a.n = n_nodes, a.W = MEL_SET_WORDS(n_nodes), a.batch = batch, a.n_step = n_step;
    for (int j = 0; j <= n_step; ++j) a.disc[j] = discount[j];
    a.seed = seed, a.counter = reinterpret_cast<unsigned long long*>(draw_counter), a.prefix = scratch;
    MEL_LAUNCH(replay_sample_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("replay_sample");
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
    if (env->n_nodes > 64) MEL_LAUNCH((env_kernel<OP_OBSERVE, 2>), dim3((n + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    else MEL_LAUNCH((env_kernel<OP_OBSERVE, 1>), dim3((n + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("env_observe");
}

}  // extern "C"
