// Device functions behind the forward's plan masks (fp32 radius adjacency; agent / one-hop / two-hop node sets), shared by
// plan_masks_kernel (plan.hpp, fwd.hip) and the env round kernel's plan sink (env.hip, mel_env_batch.plan_*).  No kernels
// here: the header is included by two translation units.
#pragma once
#include "common.hpp"

namespace mel {

// [3P] torch_cluster radius_graph(pos, r=0.2, loop=False, max_num_neighbors=32) on the fp32 obs
// positions (common.py:47-48): d2 = dx*dx + dy*dy < float(0.2*0.2), no fma; per target
// the first 33 hits in index order (self included) survive, then self is dropped.
__device__ __forceinline__ uint64_t radius_sources(float x, float y, int lane, int n) {
    // node j's position as an LDS broadcast instead of two v_readlane (see geometric_one_hop in env.hip); the callers'
    // workgroups are 4 wavefronts
    __shared__ float sx[4][64], sy[4][64];
    const int w = (threadIdx.x >> 6) & 3;
    sx[w][lane] = x, sy[w][lane] = y;
    const float r2 = (float)(0.2 * 0.2);
    uint64_t m = 0;
#pragma unroll 5
    for (int j = 0; j < n; ++j) {
        const float dx = x - sx[w][j], dy = y - sy[w][j];
        const float d2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
        if (d2 < r2) m |= 1ull << j;
    }
    while (__popcll(m) > 33) m &= ~(1ull << (63 - __clzll((long long)m)));
    return (lane < n) ? (m & ~(1ull << lane)) : 0ull;
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

// the plan masks of one env from its fp32 node positions (lane = node) and its agent set; want_receptive < 0: adjacency
// only (HL-DGN), 0: adjacency + agent set, 1: + one- / two-hop sets and sizes.  Shared by plan_masks_kernel and the env
// round kernel's plan sink (mel_env_batch.plan_*), which therefore write bit-identical buffers.
struct PlanSink {
    uint64_t* adj;
    uint64_t* live;
    uint64_t* u1;
    uint64_t* u2;
    int32_t* cnt;
};
__device__ __forceinline__ void plan_masks_env(float x, float y, uint64_t live, int want_receptive, int b, int bs, int n,
                                               int lane, const PlanSink& p) {
    const uint64_t src = radius_sources(x, y, lane, n);
    if (lane < n) p.adj[(size_t)b * n + lane] = src;
    if (want_receptive < 0) return;
    if (!want_receptive) {
        if (lane == 0) p.live[b] = live;
        return;
    }
    const uint64_t closed = (lane < n) ? (src | (1ull << lane)) : 0ull;   // sources incl. self-loop
    const uint64_t u1 = wave_or_u64(((live >> lane) & 1ull) ? closed : 0ull);
    const uint64_t u2 = wave_or_u64(((u1 >> lane) & 1ull) ? closed : 0ull);
    if (lane == 0) {
        p.live[b] = live;
        p.u1[b] = u1;
        p.u2[b] = u2;
        p.cnt[b] = __popcll(live);
        p.cnt[bs + b] = __popcll(u1);
        p.cnt[2 * bs + b] = __popcll(u2);
    }
}

}  // namespace mel
