// Device functions behind the forward's plan masks (fp32 radius adjacency; agent / one-hop / two-hop node sets), shared by
// plan_masks_kernel (plan.hpp, fwd.hip) and the env round kernel's plan sink (env.hip, mel_env_batch.plan_*).  No kernels
// here: the header is included by two translation units.
#pragma once
#include "common.hpp"

namespace mel {

// [3P] torch_cluster radius_graph(pos, r=0.2, loop=False, max_num_neighbors=32) on the fp32 obs
// positions (common.py:47-48): d2 = dx*dx + dy*dy < float(0.2*0.2), no fma; per target
// the first 33 hits in index order (self included) survive, then self is dropped.
// x[h], y[h] / src[h]: node lane + 64 h (zeros / empty for nodes >= n).
template <int W>
__device__ __forceinline__ void radius_sources(const float (&x)[W], const float (&y)[W], int lane, int n,
                                               NodeSet<W> (&src)[W]) {
    // node j's position as an LDS broadcast instead of two v_readlane (see geometric_one_hop in env.hip); the callers'
    // workgroups are 4 wavefronts
    __shared__ float sx[4][64 * W], sy[4][64 * W];
    const int w = (threadIdx.x >> 6) & 3;
    MEL_W_FOR(h) sx[w][lane + 64 * h] = x[h], sy[w][lane + 64 * h] = y[h];
    const float r2 = (float)(0.2 * 0.2);
    MEL_W_FOR(h) src[h] = ns_zero<W>();
    MEL_W_FOR(k) {
        const int cnt = n - 64 * k < 64 ? n - 64 * k : 64;
#pragma unroll 5
        for (int jj = 0; jj < cnt; ++jj) {
            const float xj = sx[w][64 * k + jj], yj = sy[w][64 * k + jj];
            MEL_W_FOR(h) {
                const float dx = x[h] - xj, dy = y[h] - yj;
                const float d2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
                if (d2 < r2) src[h].w[k] |= 1ull << jj;
            }
        }
    }
    MEL_W_FOR(h) {
        while (ns_count(src[h]) > 33) src[h] = src[h] & ~ns_bit<W>(ns_highest(src[h]));
        src[h] = (lane + 64 * h < n) ? (src[h] & ~ns_bit<W>(lane + 64 * h)) : ns_zero<W>();
    }
}

// ---- node-feature table (MEL_FWD_INTEGER_FEATURES) ---------------------------------------------------------------------
// The five node features the encoder reads are small integers in the env's observations (graph.py:261-269: degree < N,
// messages transmitted <= 4 for a policy agent - at most four acting steps, graph.py:330-334, the source's forced first
// transmission included -, last action, interested, has message as 0 / 1), so a node's encoder row and its conv1
// projections are functions of one of N * 40 TUPLES.  With the flag set the forward evaluates the encoder and the conv1
// projections once per tuple (table rows, every call - nothing is kept between calls) and the attention kernels fetch rows
// by tuple id instead of by packed receptive-field row.  Per row it is the same arithmetic in the same order, so logits are
// bit-identical to the row-list path.  tuple id = degree * 40 + messages * 8 + action * 4 + interested * 2 + has_message.
// (Scripted agents relay without a step budget: the loops leave the flag off for envs that have them.)
constexpr int FEATURE_TUPLES_PER_DEGREE = 40;      // 5 message counts x 8 flag combinations
__device__ __forceinline__ int node_feature_id(const float* f, int n, int* bad) {
    const float deg = f[0], msg = f[1], act = f[2], itr = f[3], has = f[4];
    int d = (int)deg, m = (int)msg, a = (int)act, i = (int)itr, h = (int)has;
    const bool ok = (float)d == deg && (float)m == msg && (float)a == act && (float)i == itr && (float)h == has &&
                    d >= 0 && d < n && m >= 0 && m < 5 && a >= 0 && a < 2 && i >= 0 && i < 2 && h >= 0 && h < 2;
    if (!ok) {                                   // not an observation of this env family: flagged, clamped into the table
        *bad = 1;
        d = d < 0 ? 0 : (d >= n ? n - 1 : d), m = m < 0 ? 0 : (m > 4 ? 4 : m);
        a = a != 0, i = i != 0, h = h != 0;
    }
    return d * FEATURE_TUPLES_PER_DEGREE + m * 8 + a * 4 + i * 2 + h;
}

// the plan masks of one env from its fp32 node positions (x[h], y[h]: node lane + 64 h) and its agent set;
// want_receptive < 0: adjacency only (HL-DGN), 0: adjacency + agent set, 1: + one- / two-hop sets and sizes.  Shared by
// plan_masks_kernel and the env round kernel's plan sink (mel_env_batch.plan_*), which therefore write bit-identical
// buffers.  Sets are W words each (MEL_SET_WORDS).
struct PlanSink {
    uint64_t* adj;
    uint64_t* live;
    uint64_t* u1;
    uint64_t* u2;
    int32_t* cnt;
};
template <int W>
__device__ __forceinline__ void plan_masks_env(const float (&x)[W], const float (&y)[W], const NodeSet<W>& live,
                                               int want_receptive, int b, int bs, int n, int lane, const PlanSink& p) {
    NodeSet<W> src[W];
    radius_sources<W>(x, y, lane, n, src);
    MEL_W_FOR(h) if (lane + 64 * h < n) ns_store<W>(p.adj, (size_t)b * n + lane + 64 * h, src[h]);
    if (want_receptive < 0) return;
    if (!want_receptive) {
        if (lane == 0) ns_store<W>(p.live, b, live);
        return;
    }
    NodeSet<W> closed[W];                                                // sources incl. self-loop
    MEL_W_FOR(h) closed[h] = (lane + 64 * h < n) ? (src[h] | ns_bit<W>(lane + 64 * h)) : ns_zero<W>();
    NodeSet<W> acc = ns_zero<W>();
    MEL_W_FOR(h) if (ns_mine(live, lane, h)) acc |= closed[h];
    const NodeSet<W> u1 = ns_wave_or(acc);
    acc = ns_zero<W>();
    MEL_W_FOR(h) if (ns_mine(u1, lane, h)) acc |= closed[h];
    const NodeSet<W> u2 = ns_wave_or(acc);
    if (lane == 0) {
        ns_store<W>(p.live, b, live);
        ns_store<W>(p.u1, b, u1);
        ns_store<W>(p.u2, b, u2);
        p.cnt[b] = ns_count(live);
        p.cnt[bs + b] = ns_count(u1);
        p.cnt[2 * bs + b] = ns_count(u2);
    }
}

}  // namespace mel
