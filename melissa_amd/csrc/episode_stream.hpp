// Continuous episode supply: World.reset's sampling (graph_env/env/utils/core.py:343-395) on the device, bit for bit.
//
// Included by env.hip inside namespace mel (it calls env_reset / env_store).  Three launches per refill:
//   episode_draw_kernel    one thread per env: the env's own generator (numpy Generator(PCG64)) draws episode_seed and
//                          the graph for each free ring slot, in order                             core.py:372,378
//   episode_fill_kernel    one wavefront per new episode: RandomState(episode_seed) (MT19937) draws movement seed, source,
//                          interest density and the interested set; RandomState(movement_seed) fills the movement offsets;
//                          the graph is copied from the packed dataset; GraphEnv.reset + World.reset run into the snapshot
//                          batch                                                          core.py:381-394,316-319,398-437
//   episode_publish_kernel produced[b] += new_count[b]
//
// numpy algorithms restated here (numpy/random/src: pcg64.h, mt19937.c, distributions.c, legacy):
//   PCG64       state' = state * 0x2360ED051FC65DA44385DF649FCCF645 + inc (mod 2^128); out = rotr64(hi ^ lo, hi >> 58)
//               next_uint32 returns the low half of a fresh 64-bit output and keeps the high half for the next call
//   Generator.integers(0, hi) / choice(n)  (range < 2^32): Lemire's multiply-shift on next_uint32 with rejection
//   RandomState(seed)     init_genrand: key[0] = seed; key[i] = 1812433253 * (key[i-1] ^ (key[i-1] >> 30)) + i
//   RandomState.randint   masked rejection: draw & mask until <= range
//   RandomState.uniform   low + (high - low) * ((a >> 5) * 2^26 + (b >> 6)) / 2^53, a, b consecutive 32-bit outputs
//   RandomState.choice(n, k, replace=False) = permutation(n)[:k]: Fisher-Yates from the top, j = random_interval(i)
#pragma once

namespace mel {

// ---- PCG64 (one thread) ------------------------------------------------------------------------------------------
struct Pcg64 {
    uint64_t lo, hi, inc_lo, inc_hi;
    uint32_t has32, half;
};

__device__ __forceinline__ uint64_t pcg64_next64(Pcg64& g) {
    constexpr uint64_t MH = 2549297995355413924ull, ML = 4865540595714422341ull;
    // (hi:lo) * (MH:ML) mod 2^128, then + inc
    uint64_t lo = g.lo * ML;
    uint64_t hi = __umul64hi(g.lo, ML) + g.hi * ML + g.lo * MH;
    const uint64_t lo2 = lo + g.inc_lo;
    hi = hi + g.inc_hi + (lo2 < lo ? 1ull : 0ull);
    g.lo = lo2, g.hi = hi;
    const uint64_t x = hi ^ lo2;
    const unsigned r = (unsigned)(hi >> 58);
    return (x >> r) | (x << ((64u - r) & 63u));
}
__device__ __forceinline__ uint32_t pcg64_next32(Pcg64& g) {
    if (g.has32) {
        g.has32 = 0;
        return g.half;
    }
    const uint64_t v = pcg64_next64(g);
    g.has32 = 1;
    g.half = (uint32_t)(v >> 32);
    return (uint32_t)v;
}
// uniform integer in [0, rng] (rng < 2^32 - 1): buffered_bounded_lemire_uint32
__device__ __forceinline__ uint32_t pcg64_bounded(Pcg64& g, uint32_t rng) {
    if (rng == 0) return 0;
    const uint32_t excl = rng + 1u;
    uint64_t m = (uint64_t)pcg64_next32(g) * excl;
    uint32_t left = (uint32_t)m;
    if (left < excl) {
        const uint32_t thr = (0xFFFFFFFFu - rng) % excl;
        while (left < thr) {
            m = (uint64_t)pcg64_next32(g) * excl;
            left = (uint32_t)m;
        }
    }
    return (uint32_t)(m >> 32);
}

struct StreamArgs {
    mel_episode_stream st;
    mel_graph_pool graphs;
    mel_episode_pool pool;      // ring arrays (written here)
    mel_env_batch env;          // live envs: ep_cursor
    mel_env_batch snap;         // reset snapshots, one "env" per ring slot
    int max_new, discard;
};

__global__ __launch_bounds__(256) void episode_draw_kernel(StreamArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.st.n_envs) return;
    const int K = a.st.ring;
    Pcg64 g;
    g.lo = a.st.pcg[4 * (size_t)b], g.hi = a.st.pcg[4 * (size_t)b + 1];
    g.inc_lo = a.st.pcg[4 * (size_t)b + 2], g.inc_hi = a.st.pcg[4 * (size_t)b + 3];
    g.has32 = a.st.pcg_half[2 * (size_t)b], g.half = a.st.pcg_half[2 * (size_t)b + 1];
    const uint32_t graph_rng = (uint32_t)(a.graphs.n_graphs - 1);
    for (int d = 0; d < a.discard; ++d) {                       // episodes the reference samples and throws away
        (void)pcg64_bounded(g, 999999999u);
        if (!a.st.fixed_graph) (void)pcg64_bounded(g, graph_rng);
    }
    const int cur = a.env.scalars[(size_t)b * MEL_ENV_SCALARS + MEL_S_EP_CURSOR];
    const int first = a.st.produced[b];
    int cnt = 0;
    while (cnt < a.max_new && first + cnt <= cur + K - 2) {
        const int slot = b * K + (first + cnt) % K;
        a.st.draw_seed[slot] = pcg64_bounded(g, 999999999u);                                  // core.py:372
        a.st.draw_graph[slot] = a.st.fixed_graph ? 0 : (int)pcg64_bounded(g, graph_rng);      // core.py:377-379
        const int w = atomicAdd(a.st.work, 1);
        a.st.work[1 + w] = slot;
        ++cnt;
    }
    a.st.pcg[4 * (size_t)b] = g.lo, a.st.pcg[4 * (size_t)b + 1] = g.hi;
    a.st.pcg_half[2 * (size_t)b] = g.has32, a.st.pcg_half[2 * (size_t)b + 1] = g.half;
    a.st.new_count[b] = cnt;
}

__global__ __launch_bounds__(256) void episode_publish_kernel(StreamArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.st.n_envs) return;
    a.st.produced[b] += a.st.new_count[b];
    if (b == 0) a.st.work[0] = 0;
}

// ---- MT19937 held by one wavefront: the 624-word key in LDS, the position wave-uniform ------------------------------
struct Mt {
    uint32_t* key;      // LDS [624]
    int pos;            // wave-uniform
};

// init_genrand (mt19937_seed): a serial recurrence.  It is wave-uniform, so it is written for the SCALAR unit: the chain
// s -> 1812433253 * (s ^ (s >> 30)) + i runs in SGPRs (four dependent SALU operations per word instead of a VALU chain with
// a quarter-rate multiply), lane (i % 64) of a VGPR picks word i up with a compare + select, and 64 words go to LDS with one
// ds_write.
__device__ __forceinline__ void mt_seed(Mt& m, uint32_t seed, int lane) {
    uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)seed);
    for (int c = 0; c < 624; c += 64) {
        int mine = 0;
#pragma unroll 16
        for (int j = 0; j < 64; ++j) {
            mine = (lane == j) ? (int)s : mine;                 // lane j keeps word c + j (s is an SGPR)
            s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)(c + j + 1);
        }
        if (c + lane < 624) m.key[c + lane] = (uint32_t)mine;
    }
    m.pos = 624;
    __syncthreads();
}

__device__ __forceinline__ uint32_t mt_twist(uint32_t cur, uint32_t nxt, uint32_t far) {
    const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7FFFFFFFu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
}
// mt19937_gen: key[k] = key[(k + 397) % 624] ^ twist(key[k], key[k + 1]) in place.  k < 227 reads only old words;
// 227 <= k < 454 reads new words [0, 227); 454 <= k < 623 reads new words [227, 396); k = 623 reads the NEW key[0].
__device__ __forceinline__ void mt_regen(Mt& m, int lane) {
    uint32_t v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int k = c * 64 + lane;
        if (k < 227) v[c] = mt_twist(m.key[k], m.key[k + 1], m.key[k + 397]);
    }
    __syncthreads();                       // every old word of phase 1 has been read (key[227] by k = 226 included)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int k = c * 64 + lane;
        if (k < 227) m.key[k] = v[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int k = 227 + c * 64 + lane;
        if (k < 454) v[c] = mt_twist(m.key[k], m.key[k + 1], m.key[k - 227]);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int k = 227 + c * 64 + lane;
        if (k < 454) m.key[k] = v[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int k = 454 + c * 64 + lane;
        if (k < 623) v[c] = mt_twist(m.key[k], m.key[k + 1], m.key[k - 227]);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int k = 454 + c * 64 + lane;
        if (k < 623) m.key[k] = v[c];
    }
    __syncthreads();
    if (lane == 0) m.key[623] = mt_twist(m.key[623], m.key[0], m.key[396]);
    __syncthreads();
    m.pos = 0;
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9D2C5680u;
    y ^= (y << 15) & 0xEFC60000u;
    y ^= y >> 18;
    return y;
}
// next 32-bit output, the same value in every lane (control flow stays wave-uniform: pos is uniform)
__device__ __forceinline__ uint32_t mt_next32(Mt& m, int lane) {
    if (m.pos == 624) mt_regen(m, lane);
    const uint32_t y = m.key[m.pos];                 // LDS broadcast
    m.pos += 1;
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)mt_temper(y));
}
__device__ __forceinline__ double mt_double(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}
// random_interval / buffered_bounded_masked_uint32: uniform in [0, rng] by masked rejection
__device__ __forceinline__ uint32_t mt_masked(Mt& m, uint32_t rng, int lane) {
    if (rng == 0) return 0;
    uint32_t mask = rng;
    mask |= mask >> 1, mask |= mask >> 2, mask |= mask >> 4, mask |= mask >> 8, mask |= mask >> 16;
    uint32_t v;
    do {
        v = mt_next32(m, lane) & mask;
    } while (v > rng);
    return v;
}

// One wavefront (a 64-thread workgroup) per new episode.
template <int W>
__global__ __launch_bounds__(64) void episode_fill_kernel(StreamArgs a) {
    __shared__ uint32_t key[624];
    const int lane = threadIdx.x;
    const int n = a.pool.n_nodes;
    const int n_work = a.st.work[0];
    for (int w = blockIdx.x; w < n_work; w += gridDim.x) {
        const int slot = a.st.work[1 + w];
        Mt m{key, 624};
        // ---- ep_rng = RandomState(episode_seed)                                                    core.py:373
        mt_seed(m, a.st.draw_seed[slot], lane);
        const uint32_t movement_seed = mt_masked(m, 999999999u, lane);                              // :381
        const int origin = (int)mt_masked(m, (uint32_t)(n - 1), lane);                              // :384
        double density = a.st.fixed_interest_density;
        if (!a.st.has_density) {                                                                    // :385
            const uint32_t d0 = mt_next32(m, lane);
            const uint32_t d1 = mt_next32(m, lane);
            const double range = 1.0 - 0.1;
            density = 0.1 + range * mt_double(d0, d1);
        }
        const int k_int = (int)(density * (double)n);                                               // :392
        // permutation(n)[:k] (:393): the lane of node i holds arr[i]; swap(i, j) for i = n-1 .. 1, j = random_interval(i)
        int arr[W];
        MEL_W_FOR(h) arr[h] = lane + 64 * h;
        for (int i = n - 1; i >= 1; --i) {
            const int j = (int)mt_masked(m, (uint32_t)i, lane);
            const int vi = node_i32<W>(arr, i), vj = node_i32<W>(arr, j);
            MEL_W_FOR(h) {
                if (lane + 64 * h == i) arr[h] = vj;
                if (lane + 64 * h == j) arr[h] = vi;
            }
        }
        NodeSet<W> mine = ns_zero<W>();
        MEL_W_FOR(h) if (lane + 64 * h < k_int) mine |= ns_bit<W>(arr[h]);
        const NodeSet<W> interested = ns_wave_or(mine);
        // ---- the pool slot
        const int g = a.st.draw_graph[slot];
        MEL_W_FOR(h) {
            if (lane + 64 * h < n) {
                const size_t src = (size_t)g * n + lane + 64 * h, dst = (size_t)slot * n + lane + 64 * h;
                const_cast<double*>(a.pool.pos)[2 * dst] = a.graphs.pos[2 * src];
                const_cast<double*>(a.pool.pos)[2 * dst + 1] = a.graphs.pos[2 * src + 1];
                ns_store<W>(const_cast<uint64_t*>(a.pool.one_hop), dst, ns_load<W>(a.graphs.one_hop, src));
            }
        }
        if (lane == 0) {
            const_cast<int32_t*>(a.pool.origin)[slot] = origin;
            ns_store<W>(const_cast<uint64_t*>(a.pool.interested), slot, interested);
            if (a.pool.scripted) ns_store<W>(const_cast<uint64_t*>(a.pool.scripted), slot, ns_zero<W>());      // scripted_agents_ratio == 0
        }
        // ---- movement_np_random = RandomState(movement_seed): step * uniform(-1, 1), x's then y's per move   :316-319,382
        if (a.env.dynamic_graph) {
            __syncthreads();
            mt_seed(m, movement_seed, lane);
            const int total = a.pool.max_moves * 2 * n;                   // doubles; 312 per regeneration of the key
            double* out = const_cast<double*>(a.pool.moves) + (size_t)slot * total;
            for (int base = 0; base < total; base += 312) {
                mt_regen(m, lane);
                for (int t = lane; t < 312 && base + t < total; t += 64) {
                    const double u = mt_double(mt_temper(key[2 * t]), mt_temper(key[2 * t + 1]));
                    const double x = -1.0 + 2.0 * u;                      // random_uniform: low + range * u
                    out[base + t] = 0.06 * x;                             // NODES_MOVEMENT_STEP * ... (constants.py:4)
                }
            }
        }
        __syncthreads();                                                  // the slot is complete and visible to this wave
        // ---- GraphEnv.reset + World.reset for this episode into its snapshot                        core.py:398-437
        Env<W> s{};
        s.skip = SKIP_NONE, s.sel = NONE;
        MEL_W_FOR(h) s.act[h] = NONE, s.cur_act[h] = NONE;
        env_reset(a.snap, a.pool, slot, s, slot, 0, lane);
        s.ep_cursor = 0;
        env_store(a.snap, slot, lane, s);
        __syncthreads();                                                  // key[] is reused by the next work item
    }
}

// Pacing gate for a side stream: one wavefront polls a device counter that the main stream's kernels advance (the round
// counter of mel_env_round) until it reaches `target`, so that what follows on the side stream runs when the main stream has
// got that far - WITHOUT any event on the main stream (an event record / wait between HIP-graph replays costs the replayed
// step ~8 us on this stack).  Bounded: after timeout_us the gate opens anyway (what follows must be safe at any time; the
// episode refill is - it only ever writes ring slots no env can be using, judged from the cursors it reads).
__global__ __launch_bounds__(64) void wait_counter_kernel(const uint32_t* counter, uint32_t target, uint32_t timeout_us) {
    if (threadIdx.x != 0) return;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();            // 100 MHz
    const uint64_t limit = (uint64_t)timeout_us * 100ull;
    for (;;) {
        const uint32_t c = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int32_t)(c - target) >= 0) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > limit) break;
        __builtin_amdgcn_s_sleep(64);
    }
}

static mel_status launch_episode_refill(const mel_episode_stream* st, const mel_graph_pool* graphs,
                                        const mel_episode_pool* pool, const mel_env_batch* env, int32_t max_new,
                                        int32_t discard, hipStream_t stream) {
    if (!st || !graphs || !pool || !env) return fail(MEL_ERR_INVALID_ARG, "null argument");
    const int B = st->n_envs, K = st->ring;
    if (B != env->n_envs || K < 3) return fail(MEL_ERR_INVALID_ARG, "stream of %d envs x %d slots for %d envs (ring >= 3)", B, K, env->n_envs);
    if (!st->pcg || !st->pcg_half || !st->produced || !st->draw_seed || !st->draw_graph || !st->work || !st->new_count)
        return fail(MEL_ERR_INVALID_ARG, "episode stream has null buffers");
    if (graphs->n_nodes != env->n_nodes || graphs->n_graphs < 1 || !graphs->pos || !graphs->one_hop)
        return fail(MEL_ERR_INVALID_ARG, "graph pool of %d graphs x %d nodes for %d-node envs", graphs->n_graphs, graphs->n_nodes, env->n_nodes);
    if (st->fixed_graph && (graphs->n_graphs != 1 || env->dynamic_graph))
        return fail(MEL_ERR_UNSUPPORTED, "a fixed graph streams only when it is static (a moving fixed graph carries its positions over)");
    if (env->is_testing || env->heuristic != MEL_HEURISTIC_NONE)
        return fail(MEL_ERR_UNSUPPORTED, "the device sampler covers training mode without scripted agents");
    if (pool->n_episodes != B * K || pool->n_nodes != env->n_nodes || !pool->pos || !pool->one_hop || !pool->origin ||
        !pool->interested || (env->dynamic_graph && (!pool->moves || pool->max_moves < 1)))
        return fail(MEL_ERR_INVALID_ARG, "the ring pool must hold n_envs * ring = %d episodes", B * K);
    if (pool->produced != st->produced) return fail(MEL_ERR_INVALID_ARG, "pool->produced must be the stream's counter");
    const mel_env_batch* sn = pool->snapshot;
    if (!sn || sn->n_envs < B * K || sn->n_nodes != env->n_nodes || sn->dynamic_graph != env->dynamic_graph ||
        sn->has_local_ratio != env->has_local_ratio || !sn->pos || !sn->scalars)
        return fail(MEL_ERR_INVALID_ARG, "the ring pool needs a snapshot batch of n_envs * ring envs with the env's settings");
    if (max_new < 0 || discard < 0) return fail(MEL_ERR_INVALID_ARG, "max_new=%d discard=%d", max_new, discard);
    clear_stale_error();
    StreamArgs a{};
    a.st = *st, a.graphs = *graphs, a.pool = *pool, a.env = *env, a.snap = *sn;
    a.max_new = max_new > K ? K : max_new, a.discard = discard;
    StageScope t(MEL_STAGE_ENV_RESET, stream);
    MEL_LAUNCH(episode_draw_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, a);
    // the work-item count lives on the device: a fixed grid loops over it (surplus workgroups leave at once)
    const long expect = (long)B * (a.max_new < 4 ? a.max_new : 4);
#ifdef MEL_FILL_GRID
    const int grid = MEL_FILL_GRID;                  // tuning: fewer, longer-lived fill waves
#else
    const int grid = (int)(expect < 256 ? 256 : (expect > 4096 ? 4096 : expect));
#endif
    if (env->n_nodes > 64) MEL_LAUNCH(episode_fill_kernel<2>, dim3(grid), dim3(64), 0, stream, a);
    else MEL_LAUNCH(episode_fill_kernel<1>, dim3(grid), dim3(64), 0, stream, a);
    MEL_LAUNCH(episode_publish_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, a);
    return check_launch("episode_refill");
}

}  // namespace mel
