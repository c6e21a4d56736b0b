// Forward plan kernels: fp32 radius adjacency, agent / receptive-field sets, packed row lists and per-target
// descriptors (networks/common.py:6-64; the row lists serve l_dgn.py:117-135).  Included by fwd.hip.
#pragma once
#include "common.hpp"
#include "plan_masks.hpp"

namespace mel {

// ------------------------------------------------------------------------------------------------
// plan: one wavefront per env (= observation row), lane = node
//
// The forward is evaluated for a SET of controlling agents per env (mask L).  The reference's collector
// presents one agent per observation row (L = {obs[:, -1]}, common.py:63); within one env round every
// active agent sees the same obs_matrix (graph.py:186-188: rows differ only in the last column), so the
// round-batched loop passes all of a round's agents at once and the encoder / conv1 work is shared:
//   U1 = union over g in L of closed one-hop(g)   - conv1 targets that can reach some agent's logits
//   U2 = union over t in U1 of closed one-hop(t)  - their sources
// Rows are packed per env in id order, so the packed position of node j is popcount(mask below j).
// ------------------------------------------------------------------------------------------------
// everything the attention kernel needs to know about one target row, in one 32-byte load
// (W = 1: 32 bytes; W = 2, graphs of 65 .. 128 nodes: 48 bytes)
template <int W>
struct TargetDesc {
    NodeSet<W> sources; // source nodes of the target (closed neighbourhood for GATv2, open for TransformerConv)
    NodeSet<W> smask;   // node set the source rows are packed by
    int32_t soff;       // first source row of the env
    int32_t env;
    int32_t node;
    int32_t cat_row;    // conv1: agent row whose head input takes x_1 / x_2 from this target, or -1
};
inline size_t target_desc_bytes(int n_nodes) { return n_nodes > 64 ? sizeof(TargetDesc<2>) : sizeof(TargetDesc<1>); }

struct PlanBuffers {
    // node sets: MEL_SET_WORDS(N) words each
    uint64_t* adj;      // [bs*N] sources of target i (radius rule, self excluded)
    uint64_t* live;     // [bs]   L: controlling agents of the env
    uint64_t* u1;       // [bs]
    uint64_t* u2;       // [bs]
    int32_t* cnt;       // [3*bs] |L|, |U1|, |U2|
    int32_t* offL;      // [bs+1] exclusive scans (last entry = total)
    int32_t* off1;      // [bs+1]
    int32_t* off2;      // [bs+1]
    int32_t* nid2;      // [sum|U2|] global node id (b*N + i) of packed row
    int32_t* arow1;     // [sum|U1|] row of the U2 list holding the same node
    float* dm1;         // [sum|U1|] decision-maker flag of the node (l_dgn.py:128)
    void* desc1;        // [sum|U1|] TargetDesc<W>: conv1 target rows
    void* desc2;        // [R]       TargetDesc<W>: conv2 target rows (one per agent row)
    int32_t* row_env;   // [R] env of agent row r
    int32_t* row_agent; // [R] agent (node id) of agent row r
    int32_t* arow_g;    // [R] row of the U1 list holding the agent
    float* dm_g;        // [R]
    // node-feature table mode (plan_masks.hpp): tuple id of every node, per-env "a feature was not an integer in range"
    // flags, and [0] = table rows the forward used (0 = row-list path)
    int32_t* fid;       // [bs*N]
    int32_t* fbad;      // [bs]
    int32_t* fmeta;     // [4]
};

// standalone adjacency for the learn path (one wave per observation row)
template <int W>
__global__ __launch_bounds__(256) void radius_graph_kernel(const float* __restrict__ obs, int bs, int n, int obs_stride,
                                                           int node_cols, uint64_t* __restrict__ adj) {
    const int lane = lane_id();
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= bs) return;
    float x[W], y[W];
    MEL_W_FOR(h) {
        x[h] = 0.f, y[h] = 0.f;
        if (lane + 64 * h < n) {
            const float* p = obs + (size_t)b * obs_stride + (lane + 64 * h) * node_cols;
            x[h] = p[0], y[h] = p[1];
        }
    }
    NodeSet<W> m[W];
    radius_sources<W>(x, y, lane, n, m);
    MEL_W_FOR(h) if (lane + 64 * h < n) ns_store<W>(adj, (size_t)b * n + lane + 64 * h, m[h]);
}

// agent_mask == null: one agent per row, taken from the index column (common.py:63)
template <int W>
__global__ __launch_bounds__(256) void plan_masks_kernel(const float* __restrict__ obs, int bs, int n,
                                                         int obs_stride, int node_cols,
                                                         const uint64_t* __restrict__ agent_mask, PlanBuffers p,
                                                         int want_receptive) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= bs) return;
    const int lane = lane_id();
    const float* row = obs + (size_t)b * obs_stride;
    float x[W], y[W];
    MEL_W_FOR(h) {
        x[h] = 0.f, y[h] = 0.f;
        if (lane + 64 * h < n) {
            x[h] = row[(lane + 64 * h) * node_cols];
            y[h] = row[(lane + 64 * h) * node_cols + 1];
        }
    }
    NodeSet<W> live = ns_zero<W>();
    if (want_receptive >= 0) {
        if (agent_mask) {
            live = ns_load<W>(agent_mask, b) & ns_full<W>(n);
        } else {                               // obs[:, -1].clamp(0, N-1).long()
            float gf = row[n * node_cols];
            gf = fminf(fmaxf(gf, 0.f), (float)(n - 1));
            live = ns_bit<W>((int)gf);
        }
    }
    plan_masks_env<W>(x, y, live, want_receptive, b, bs, n, lane, PlanSink{p.adj, p.live, p.u1, p.u2, p.cnt});
}

// exclusive scans of |L|, |U1|, |U2| over the batch (single workgroup, any bs)
__global__ __launch_bounds__(1024) void plan_scan_kernel(int bs, PlanBuffers p) {
    __shared__ int32_t part[3][1024];
    const int tid = threadIdx.x;
    const int per = (bs + 1023) / 1024;
    const int lo = min(tid * per, bs), hi = min(lo + per, bs);
    int32_t s[3] = {0, 0, 0};
    for (int b = lo; b < hi; ++b)
#pragma unroll
        for (int k = 0; k < 3; ++k) s[k] += p.cnt[k * bs + b];
#pragma unroll
    for (int k = 0; k < 3; ++k) part[k][tid] = s[k];
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        int32_t a[3] = {0, 0, 0};
        if (tid >= d)
#pragma unroll
            for (int k = 0; k < 3; ++k) a[k] = part[k][tid - d];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 3; ++k) part[k][tid] += a[k];
        __syncthreads();
    }
    int32_t o[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = part[k][tid] - s[k];
    for (int b = lo; b < hi; ++b) {
        p.offL[b] = o[0], p.off1[b] = o[1], p.off2[b] = o[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] += p.cnt[k * bs + b];
    }
    if (tid == 1023) p.offL[bs] = part[0][1023], p.off1[bs] = part[1][1023], p.off2[bs] = part[2][1023];
}

struct PlanListsArgs {
    const float* obs;
    int bs, n, obs_stride, node_cols;
    PlanBuffers p;
    int32_t* row_offsets_out;
    int self_loops, inline_scan, table_rows;
};

// body of plan_lists_kernel for workgroup `block` (four envs): a device function so that plan_enc_kernel (fwd.hip) can run
// it beside the encoder rows of the node-feature table in one launch
template <int W>
__device__ __forceinline__ void plan_lists_body(const PlanListsArgs& a, const int block) {
    const float* __restrict__ obs = a.obs;
    const int bs = a.bs, n = a.n, obs_stride = a.obs_stride, node_cols = a.node_cols;
    const PlanBuffers& p = a.p;
    int32_t* __restrict__ row_offsets_out = a.row_offsets_out;
    const int self_loops = a.self_loops, inline_scan = a.inline_scan, table_rows = a.table_rows;
    const int b = block * 4 + (threadIdx.x >> 6);
    if (b >= bs) return;
    const int lane = lane_id();
    if (b == 0 && lane == 0) p.fmeta[0] = table_rows;
    if (table_rows > 0) {                           // tuple ids for the node-feature table path
        int bad = 0;
        MEL_W_FOR(h) {
            const int node = lane + 64 * h;
            if (node < n) p.fid[(size_t)b * n + node] = node_feature_id(obs + (size_t)b * obs_stride + node * node_cols + 2, n, &bad);
        }
        const int any_bad = __ballot(bad != 0) != 0ull;
        if (lane == 0) p.fbad[b] = any_bad;
    } else if (lane == 0) {
        p.fbad[b] = 0;
    }
    const NodeSet<W> live = ns_load<W>(p.live, b), u1 = ns_load<W>(p.u1, b), u2 = ns_load<W>(p.u2, b);
    int oL, o1, o2;
    if (inline_scan) {
        // exclusive prefix of the three per-env counts, recomputed by every wave from the cnt array (a few KB out
        // of L2): cheaper than a separate single-workgroup scan launch between the two plan kernels
        int sL = 0, s1 = 0, s2 = 0;
        for (int i = lane; i < b; i += 64) sL += p.cnt[i], s1 += p.cnt[bs + i], s2 += p.cnt[2 * bs + i];
        oL = wave_sum_i32_dpp(sL), o1 = wave_sum_i32_dpp(s1), o2 = wave_sum_i32_dpp(s2);
        if (lane == 0) {
            p.offL[b] = oL, p.off1[b] = o1, p.off2[b] = o2;
            if (b == bs - 1)
                p.offL[bs] = oL + p.cnt[b], p.off1[bs] = o1 + p.cnt[bs + b], p.off2[bs] = o2 + p.cnt[2 * bs + b];
        }
    } else {
        oL = p.offL[b], o1 = p.off1[b], o2 = p.off2[b];
    }
    const float* row = obs + (size_t)b * obs_stride;
    TargetDesc<W>* desc1 = static_cast<TargetDesc<W>*>(p.desc1);
    TargetDesc<W>* desc2 = static_cast<TargetDesc<W>*>(p.desc2);
    MEL_W_FOR(h) {
        const int node = lane + 64 * h;
        const float dm = (node < n) ? row[node * node_cols + node_cols - 1] : 0.f;
        if (ns_mine(u2, lane, h)) p.nid2[o2 + ns_rank_below(u2, node)] = b * n + node;
        if (ns_mine(u1, lane, h)) {
            const int r1 = o1 + ns_rank_below(u1, node);
            p.arow1[r1] = o2 + ns_rank_below(u2, node);
            p.dm1[r1] = dm;
        }
        NodeSet<W> mine = ns_zero<W>();
        if (node < n) {
            mine = ns_load<W>(p.adj, (size_t)b * n + node);
            if (self_loops) mine |= ns_bit<W>(node);
        }
        const bool is_agent = ns_mine(live, lane, h);
        const int rL = oL + ns_rank_below(live, node);
        if (ns_mine(u1, lane, h)) {
            TargetDesc<W> d;
            d.sources = mine, d.smask = u2, d.soff = o2, d.env = b, d.node = node, d.cat_row = is_agent ? rL : -1;
            desc1[o1 + ns_rank_below(u1, node)] = d;
        }
        if (is_agent) {
            p.row_env[rL] = b;
            p.row_agent[rL] = node;
            p.arow_g[rL] = o1 + ns_rank_below(u1, node);
            p.dm_g[rL] = dm;
            TargetDesc<W> d;
            d.sources = mine, d.smask = u1, d.soff = o1, d.env = b, d.node = node, d.cat_row = rL;
            desc2[rL] = d;
        }
    }
    if (row_offsets_out && lane == 0) {
        row_offsets_out[b] = oL;
        if (b == bs - 1) row_offsets_out[bs] = oL + p.cnt[b];
    }
}

template <int W>
__global__ __launch_bounds__(256) void plan_lists_kernel(PlanListsArgs a) { plan_lists_body<W>(a, (int)blockIdx.x); }

// HL-DGN has no row lists: the tuple ids alone (one wave per env)
__device__ __forceinline__ void feature_ids_body(const float* __restrict__ obs, int bs, int n, int obs_stride, int node_cols,
                                                 const PlanBuffers& p, int table_rows, const int block) {
    const int b = block * 4 + (threadIdx.x >> 6);
    if (b >= bs) return;
    const int lane = lane_id();
    if (b == 0 && lane == 0) p.fmeta[0] = table_rows;
    if (table_rows <= 0) {
        if (lane == 0) p.fbad[b] = 0;
        return;
    }
    int bad = 0;
    for (int node = lane; node < n; node += 64)
        p.fid[(size_t)b * n + node] = node_feature_id(obs + (size_t)b * obs_stride + node * node_cols + 2, n, &bad);
    const int any_bad = __ballot(bad != 0) != 0ull;
    if (lane == 0) p.fbad[b] = any_bad;
}
__global__ __launch_bounds__(256) void feature_ids_kernel(const float* __restrict__ obs, int bs, int n, int obs_stride,
                                                          int node_cols, PlanBuffers p, int table_rows) {
    feature_ids_body(obs, bs, n, obs_stride, node_cols, p, table_rows, (int)blockIdx.x);
}

}  // namespace mel
