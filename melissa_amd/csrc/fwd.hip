// L-DGN / HL-DGN forward for MI355X (gfx950): plan (obs unpack + fp32 radius adjacency + receptive
// field), fp32-MFMA row GEMMs (gemm_f32.hpp), GATv2 edge-softmax/aggregate, segmented pool, dueling
// tail, DQN action selection.  Reference semantics: networks/common.py:6-64, l_dgn.py:92-151,
// hl_dgn.py:82-119 and SURVEY.md Appendix A for the third-party operators.
#include "common.hpp"
#include "gemm_f32.hpp"
#include "gemm_bf16.hpp"
#include "gemm_split.hpp"

namespace mel {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static thread_local Profiler* g_prof = nullptr;
Profiler* current_profiler() { return g_prof; }

// ------------------------------------------------------------------------------------------------
// GEMM dispatch
// ------------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN>
static void gemm_launch_t(const GemmArgs* gs, int count, int mode, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    GemmBatch batch{};
    batch.count = count;
    int total = 0;
    for (int i = 0; i < count; ++i) {
        batch.p[i] = gs[i];
        batch.start[i] = total;
        const int tiles = ((gs[i].M + BM - 1) / BM) * (gs[i].N / BN);
        total += (tiles + 7) & ~7;               // keep every problem's ids aligned to the 8 XCDs
    }
    batch.start[count] = total;
    if (mode == GEMM_MODE_ENC)
        hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, TM, TN, GEMM_MODE_ENC>), dim3(total), dim3(64 * WM * WN), 0, s, batch);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, TM, TN, GEMM_MODE_PLAIN>), dim3(total), dim3(64 * WM * WN), 0, s, batch);
}

template <int WM, int WN, int TM, int TN, int TAG = 0>
static void gemm_launch_persistent(const GemmArgs* gs, int count, int mode, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int LDS_STAGE = (BM + BN) * GEMM_LDS_STRIDE * 4 * 2;
    constexpr int PER_CU = (160 * 1024) / LDS_STAGE > 4 ? 4 : (160 * 1024) / LDS_STAGE;    // workgroups an LDS-limited CU holds
    GemmBatch batch{};
    batch.count = count;
    long tiles = 0;
    for (int i = 0; i < count; ++i) {
        batch.p[i] = gs[i];
        tiles += ((long)((gs[i].M + BM - 1) / BM) * (gs[i].N / BN) + 7) & ~7L;
    }
    long grid = 256L * PER_CU;                    // 256 CUs; a multiple of 8 (XCD affinity of tile ids)
    if (grid > tiles) grid = tiles;
    if (mode == GEMM_MODE_ENC)
        hipLaunchKernelGGL((gemm_f32_persistent_kernel<WM, WN, TM, TN, GEMM_MODE_ENC>), dim3((int)grid), dim3(64 * WM * WN), 0, s, batch);
    else
        hipLaunchKernelGGL((gemm_f32_persistent_kernel<WM, WN, TM, TN, GEMM_MODE_PLAIN, TAG>), dim3((int)grid), dim3(64 * WM * WN), 0, s, batch);
}

// ragged 64 x 64 launches of the round step, named per call site
static void gemm_launch_ragged(const GemmArgs* gs, int count, hipStream_t s, int tag) {
    switch (tag) {
        case 1: gemm_launch_persistent<2, 2, 1, 1, 1>(gs, count, GEMM_MODE_PLAIN, s); break;
        case 2: gemm_launch_persistent<2, 2, 1, 1, 2>(gs, count, GEMM_MODE_PLAIN, s); break;
        case 3: gemm_launch_persistent<2, 2, 1, 1, 3>(gs, count, GEMM_MODE_PLAIN, s); break;
        default: gemm_launch_persistent<2, 2, 1, 1, 0>(gs, count, GEMM_MODE_PLAIN, s); break;
    }
}

template <int WM, int WN, int TM, int TN>
static void gemm_launch_glds(const GemmArgs* gs, int count, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    GemmBatch batch{};
    batch.count = count;
    int total = 0;
    for (int i = 0; i < count; ++i) {
        batch.p[i] = gs[i];
        batch.start[i] = total;
        const int tiles = ((gs[i].M + BM - 1) / BM) * (gs[i].N / BN);
        total += (tiles + 7) & ~7;
    }
    batch.start[count] = total;
    hipLaunchKernelGGL((gemm_f32_glds_kernel<WM, WN, TM, TN>), dim3(total), dim3(64 * WM * WN), 0, s, batch);
}

// bf16 feature path: one persistent launch for every case (ragged or not)
template <int WM, int WN, int TM, int TN>
static void gemm_launch_bf16(const GemmArgs* gs, int count, int mode, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int LDS_WG = (BM + BN) * GEMB_ROW * 16 * 2;
    constexpr int PER_CU = (160 * 1024) / LDS_WG > 4 ? 4 : (160 * 1024) / LDS_WG;
    GemmBatch batch{};
    batch.count = count;
    long tiles = 0;
    for (int i = 0; i < count; ++i) {
        batch.p[i] = gs[i];
        tiles += ((long)((gs[i].M + BM - 1) / BM) * (gs[i].N / BN) + 7) & ~7L;
    }
    long grid = 256L * PER_CU;
    if (grid > tiles) grid = tiles;
    if (mode == GEMM_MODE_ENC)
        hipLaunchKernelGGL((gemm_bf16_kernel<WM, WN, TM, TN, GEMM_MODE_ENC>), dim3((int)grid), dim3(64 * WM * WN), 0, s, batch);
    else
        hipLaunchKernelGGL((gemm_bf16_kernel<WM, WN, TM, TN, GEMM_MODE_PLAIN>), dim3((int)grid), dim3(64 * WM * WN), 0, s, batch);
}

// split path: one persistent 64 x 64 kernel, named per call site like the fp32 one
template <int TAG>
static void gemm_launch_split_t(const GemmArgs* gs, int count, int mode, hipStream_t s) {
    GemmBatch batch{};
    batch.count = count;
    long tiles = 0;
    for (int i = 0; i < count; ++i) {
        batch.p[i] = gs[i];
        tiles += ((long)((gs[i].M + 63) / 64) * (gs[i].N / 64) + 7) & ~7L;
    }
    long grid = 256L * 3;                          // 53 KB of LDS per workgroup: three per CU
    if (grid > tiles) grid = tiles;
    if (mode == GEMM_MODE_ENC)
        hipLaunchKernelGGL((gemm_split_kernel<GEMM_MODE_ENC, TAG>), dim3((int)grid), dim3(256), 0, s, batch);
    else
        hipLaunchKernelGGL((gemm_split_kernel<GEMM_MODE_PLAIN, TAG>), dim3((int)grid), dim3(256), 0, s, batch);
}
static void gemm_launch_split(const GemmArgs* gs, int count, int mode, hipStream_t s, int tag) {
    switch (tag) {
        case 1: gemm_launch_split_t<1>(gs, count, mode, s); break;
        case 2: gemm_launch_split_t<2>(gs, count, mode, s); break;
        case 3: gemm_launch_split_t<3>(gs, count, mode, s); break;
        default: gemm_launch_split_t<0>(gs, count, mode, s); break;
    }
}

static mel_status check_gemm_shape(const GemmArgs& g, const char* what) {
    if (g.split && g.K < 128) return fail(MEL_ERR_UNSUPPORTED, "%s: the split path needs K >= 128 (K=%d)", what, g.K);
    const int bk = g.bf16 ? GEMB_BK : GEMM_BK;
    if (g.K % bk != 0 || g.N % 64 != 0)
        return fail(MEL_ERR_UNSUPPORTED, "%s: GEMM needs K %% %d == 0 and N %% 64 == 0 (K=%d N=%d)", what, bk, g.K, g.N);
    if (g.bf16 && (g.lda % 8 != 0 && g.A))
        return fail(MEL_ERR_UNSUPPORTED, "%s: bf16 GEMM needs lda %% 8 == 0 (lda=%d)", what, g.lda);
    return MEL_OK;
}

mel_status launch_gemm(const GemmArgs& g, int mode, hipStream_t stream, const char* what, long m_hint, int force_tile,
                       int tag) {
    if (g.M <= 0) return MEL_OK;
    if (mel_status st = check_gemm_shape(g, what)) return st;
    if (m_hint < 0 || m_hint > g.M) m_hint = g.M;
    if (g.split) {
        gemm_launch_split(&g, 1, mode, stream, tag);
        return check_launch(what);
    }
    // encoder (ENC producer): a 64 x 128 tile spans the whole hidden width, so the first layer (VALU work inside the
    // A-tile producer) is evaluated once per row instead of once per 64-column tile
    const bool enc_wide = g.bf16 && mode == GEMM_MODE_ENC && force_tile == 0 && g.N % 128 == 0;   // fp32: measured slower (21 vs 18 us)
    if (g.bf16) {
        if (force_tile == 2 && g.N % 128 == 0) gemm_launch_bf16<2, 2, 2, 2>(&g, 1, mode, stream);
        else if (enc_wide) gemm_launch_bf16<2, 2, 1, 2>(&g, 1, mode, stream);
        else gemm_launch_bf16<2, 2, 1, 1>(&g, 1, mode, stream);
        return check_launch(what);
    }
    if (enc_wide) {
        gemm_launch_persistent<2, 2, 1, 2>(&g, 1, mode, stream);
        return check_launch(what);
    }
    if (force_tile == 1 || (force_tile >= 2 && g.N % 128 == 0)) {
        switch (force_tile) {
            case 1: gemm_launch_t<2, 2, 1, 1>(&g, 1, mode, stream); break;     //  64 x  64, 4 waves
            case 2: gemm_launch_t<2, 2, 2, 2>(&g, 1, mode, stream); break;     // 128 x 128, 4 waves
            case 3: gemm_launch_t<2, 4, 2, 1>(&g, 1, mode, stream); break;     // 128 x 128, 8 waves (64x32 each)
            case 4: gemm_launch_t<4, 2, 1, 2>(&g, 1, mode, stream); break;     // 128 x 128, 8 waves (32x64 each)
            case 5: gemm_launch_t<2, 2, 2, 1>(&g, 1, mode, stream); break;     // 128 x  64, 4 waves
            case 6: gemm_launch_t<2, 2, 1, 2>(&g, 1, mode, stream); break;     //  64 x 128, 4 waves
            case 21: gemm_launch_glds<2, 2, 1, 1>(&g, 1, stream); break;               // LDS-DMA  64 x  64
            case 22: gemm_launch_glds<2, 2, 2, 2>(&g, 1, stream); break;               // LDS-DMA 128 x 128
            case 11: gemm_launch_persistent<2, 2, 1, 1>(&g, 1, mode, stream); break;   // persistent  64 x  64
            case 12: gemm_launch_persistent<2, 2, 2, 2>(&g, 1, mode, stream); break;   // persistent 128 x 128
            default: return fail(MEL_ERR_INVALID_ARG, "unknown tile %d", force_tile);
        }
        return check_launch(what);
    }
    // Measured (tools/gemm_bench.py): the 64x64 tile (4x the workgroups, a quarter of the per-wave MFMA
    // chain, 4 workgroups per CU) wins or ties everywhere except long-K problems with thousands of tiles.
    const long big = ((m_hint + 127) / 128) * (g.N / 128);
    if (g.M_dev && mode == GEMM_MODE_PLAIN)
        gemm_launch_ragged(&g, 1, stream, tag);
    else if (g.M_dev)
        gemm_launch_persistent<2, 2, 1, 1>(&g, 1, mode, stream);
    else if (g.N % 128 == 0 && big >= 1536 && g.K >= 512)
        gemm_launch_t<2, 2, 2, 2>(&g, 1, mode, stream);
    else
        gemm_launch_t<2, 2, 1, 1>(&g, 1, mode, stream);
    return check_launch(what);
}

mel_status launch_gemm_group(const GemmArgs* gs, const long* hints, int count, hipStream_t stream, const char* what,
                             int tag) {
    if (count < 1 || count > GEMM_MAX_GROUP) return fail(MEL_ERR_INVALID_ARG, "%s: group of %d", what, count);
    long big = 0;
    bool n128 = true, long_k = true;
    for (int i = 0; i < count; ++i) {
        if (gs[i].M <= 0) return fail(MEL_ERR_INVALID_ARG, "%s: empty problem in group", what);
        if (mel_status st = check_gemm_shape(gs[i], what)) return st;
        const long h = (hints && hints[i] >= 0 && hints[i] <= gs[i].M) ? hints[i] : gs[i].M;
        big += ((h + 127) / 128) * (gs[i].N / 128);
        n128 = n128 && gs[i].N % 128 == 0;
        long_k = long_k && gs[i].K >= 512;
    }
    if (gs[0].split) {
        for (int i = 1; i < count; ++i)
            if (!gs[i].split) return fail(MEL_ERR_INVALID_ARG, "%s: mixed precisions in one group", what);
        gemm_launch_split(gs, count, GEMM_MODE_PLAIN, stream, tag);
        return check_launch(what);
    }
    if (gs[0].bf16) {
        for (int i = 1; i < count; ++i)
            if (!gs[i].bf16) return fail(MEL_ERR_INVALID_ARG, "%s: mixed precisions in one group", what);
        gemm_launch_bf16<2, 2, 1, 1>(gs, count, GEMM_MODE_PLAIN, stream);
        return check_launch(what);
    }
    bool ragged = false;
    for (int i = 0; i < count; ++i) ragged = ragged || gs[i].M_dev != nullptr;
    if (ragged)       // device-side row counts: a fixed grid walks the tiles instead of a worst-case grid exiting
        gemm_launch_ragged(gs, count, stream, tag);
    else if (n128 && long_k && big >= 1536)
        gemm_launch_t<2, 2, 2, 2>(gs, count, GEMM_MODE_PLAIN, stream);
    else
        gemm_launch_t<2, 2, 1, 1>(gs, count, GEMM_MODE_PLAIN, stream);
    return check_launch(what);
}

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
struct TargetDesc {
    uint64_t sources;   // source nodes of the target (closed neighbourhood for GATv2, open for TransformerConv)
    uint64_t smask;     // node set the source rows are packed by
    int32_t soff;       // first source row of the env
    int32_t env;
    int32_t node;
    int32_t cat_row;    // conv1: agent row whose head input takes x_1 / x_2 from this target, or -1
};

struct PlanBuffers {
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
    TargetDesc* desc1;  // [sum|U1|] conv1 target rows
    TargetDesc* desc2;  // [R]       conv2 target rows (one per agent row)
    int32_t* row_env;   // [R] env of agent row r
    int32_t* row_agent; // [R] agent (node id) of agent row r
    int32_t* arow_g;    // [R] row of the U1 list holding the agent
    float* dm_g;        // [R]
};

// [3P] torch_cluster radius_graph(pos, r=0.2, loop=False, max_num_neighbors=32) on the fp32 obs
// positions (common.py:47-48): d2 = dx*dx + dy*dy < float(0.2*0.2), no fma; per target
// the first 33 hits in index order (self included) survive, then self is dropped.
__device__ __forceinline__ uint64_t radius_sources(float x, float y, int lane, int n) {
    const float r2 = (float)(0.2 * 0.2);
    uint64_t m = 0;
    for (int j = 0; j < n; ++j) {
        const float xj = lane_f32(x, j), yj = lane_f32(y, j);     // j is the loop counter: v_readlane
        const float dx = x - xj, dy = y - yj;
        const float d2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
        if (d2 < r2) m |= 1ull << j;
    }
    while (__popcll(m) > 33) m &= ~(1ull << (63 - __clzll((long long)m)));
    return (lane < n) ? (m & ~(1ull << lane)) : 0ull;
}

// standalone adjacency for the learn path (one wave per observation row)
__global__ __launch_bounds__(256) void radius_graph_kernel(const float* __restrict__ obs, int bs, int n, int obs_stride,
                                                           int node_cols, uint64_t* __restrict__ adj) {
    const int lane = lane_id();
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= bs) return;
    float x = 0.f, y = 0.f;
    if (lane < n) {
        const float* p = obs + (size_t)b * obs_stride + lane * node_cols;
        x = p[0], y = p[1];
    }
    const uint64_t m = radius_sources(x, y, lane, n);
    if (lane < n) adj[(size_t)b * n + lane] = m;
}

// agent_mask == null: one agent per row, taken from the index column (common.py:63)
__global__ __launch_bounds__(256) void plan_masks_kernel(const float* __restrict__ obs, int bs, int n,
                                                         int obs_stride, int node_cols,
                                                         const uint64_t* __restrict__ agent_mask, PlanBuffers p,
                                                         int want_receptive) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= bs) return;
    const int lane = lane_id();
    const float* row = obs + (size_t)b * obs_stride;
    float x = 0.f, y = 0.f;
    if (lane < n) {
        x = row[lane * node_cols];
        y = row[lane * node_cols + 1];
    }
    const uint64_t src = radius_sources(x, y, lane, n);
    if (lane < n) p.adj[(size_t)b * n + lane] = src;
    const uint64_t full = (n == 64) ? ~0ull : ((1ull << n) - 1ull);
    uint64_t live;
    if (want_receptive < 0) {              // adjacency only: the row has no index column
        return;
    } else if (agent_mask) {
        live = agent_mask[b] & full;
    } else {                               // obs[:, -1].clamp(0, N-1).long()
        float gf = row[n * node_cols];
        gf = fminf(fmaxf(gf, 0.f), (float)(n - 1));
        live = 1ull << (int)gf;
    }
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

__global__ __launch_bounds__(256) void plan_lists_kernel(const float* __restrict__ obs, int bs, int n,
                                                         int obs_stride, int node_cols, PlanBuffers p,
                                                         int32_t* __restrict__ row_offsets_out, int self_loops,
                                                         int inline_scan) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= bs) return;
    const int lane = lane_id();
    const uint64_t live = p.live[b], u1 = p.u1[b], u2 = p.u2[b];
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
    const float dm = (lane < n) ? row[lane * node_cols + node_cols - 1] : 0.f;
    if ((u2 >> lane) & 1ull) p.nid2[o2 + rank_below(u2, lane)] = b * n + lane;
    if ((u1 >> lane) & 1ull) {
        const int r1 = o1 + rank_below(u1, lane);
        p.arow1[r1] = o2 + rank_below(u2, lane);
        p.dm1[r1] = dm;
    }
    const uint64_t mine = (lane < n) ? (p.adj[(size_t)b * n + lane] | (self_loops ? (1ull << lane) : 0ull)) : 0ull;
    const bool is_agent = (live >> lane) & 1ull;
    const int rL = oL + rank_below(live, lane);
    if ((u1 >> lane) & 1ull) {
        TargetDesc d;
        d.sources = mine, d.smask = u2, d.soff = o2, d.env = b, d.node = lane, d.cat_row = is_agent ? rL : -1;
        p.desc1[o1 + rank_below(u1, lane)] = d;
    }
    if (is_agent) {
        p.row_env[rL] = b;
        p.row_agent[rL] = lane;
        p.arow_g[rL] = o1 + rank_below(u1, lane);
        p.dm_g[rL] = dm;
        TargetDesc d;
        d.sources = mine, d.smask = u1, d.soff = o1, d.env = b, d.node = lane, d.cat_row = rL;
        p.desc2[rL] = d;
    }
    if (row_offsets_out && lane == 0) {
        row_offsets_out[b] = oL;
        if (b == bs - 1) row_offsets_out[bs] = oL + p.cnt[b];
    }
}

// ------------------------------------------------------------------------------------------------
// GATv2 edge softmax + aggregation.  One wavefront per target node; the 64 lanes
// span the heads*C output channels (VPL contiguous channels per lane, so a head is C/VPL adjacent
// lanes and the per-head score reduction is a few xor-shuffles).  Sources are streamed once with an
// online softmax: out = sum_j exp(e_j - m) x_l[j] / (sum_j exp(e_j - m) + 1e-16).
// ------------------------------------------------------------------------------------------------
enum { ATT_ROWS = 0, ATT_POOL = 1, ATT_SINGLE = 2 };

struct AttArgs {
    const float* xl;        // source rows
    int ld_l;
    const float* xr;        // target rows
    int ld_r;
    const float* att;       // [heads*C]
    const float* bias;      // [heads*C]
    const uint64_t* adj;    // [bs*N]
    const uint64_t* live;   // [bs] controlling agents (ATT_ROWS: whose x_1 / x_2 go to the head input)
    const uint64_t* tmask;  // [bs] targets, or null = all nodes
    const uint64_t* smask;  // [bs] set the source rows are packed by, or null = all nodes
    const int32_t* toff;    // [bs] first target row, or null = b*N
    const int32_t* soff;    // [bs] first source row, or null = b*N
    const int32_t* loff;    // [bs+1] first agent row of the env (ATT_ROWS) / row count at [bs] (ATT_SINGLE)
    int bs, n, lanes_per_head;
    int kind;               // MEL_CONV_*
    float score_scale;      // TransformerConv: 1 / sqrt(C)
    // ATT_ROWS
    float* out;             // [rows, ldo] relu(out + bias)
    int ldo;
    float* xcat;            // [R, ld_cat] head input: x_1 | x_2 | x_3 (l_dgn.py:139)
    int ld_cat, hidden;
    const float* h0;        // encoder rows (packed by smask), [*, hidden]
    // ATT_POOL
    const float* obs;       // dm flag source
    int obs_stride, node_cols, aggregator;
    float* pooled;          // [bs, heads*C]
    // ATT_ROWS / ATT_SINGLE: one wavefront per target row
    const TargetDesc* desc;     // [rows] per-target descriptor
    const int32_t* rows_dev;    // device-side row count
    long rows_hint;             // expected rows (grid sizing only)
    int rows_cap, cat_off;
    int bf16;                   // bf16 feature path: xl / xr / out / xcat / h0 hold bf16 rows (att / bias stay fp32)
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int VPL>
struct Vec {
    float v[VPL];
};

template <int VPL>
__device__ __forceinline__ Vec<VPL> load_vec(const float* p) {
    Vec<VPL> r;
    if constexpr (VPL >= 4) {
#pragma unroll
        for (int i = 0; i < VPL / 4; ++i) {
            const float4 t = reinterpret_cast<const float4*>(p)[i];
            r.v[4 * i] = t.x, r.v[4 * i + 1] = t.y, r.v[4 * i + 2] = t.z, r.v[4 * i + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < VPL; ++i) r.v[i] = p[i];
    }
    return r;
}

template <int VPL>
__device__ __forceinline__ void store_vec(float* p, const Vec<VPL>& r) {
    if constexpr (VPL >= 4) {
#pragma unroll
        for (int i = 0; i < VPL / 4; ++i)
            reinterpret_cast<float4*>(p)[i] = make_float4(r.v[4 * i], r.v[4 * i + 1], r.v[4 * i + 2], r.v[4 * i + 3]);
    } else {
#pragma unroll
        for (int i = 0; i < VPL; ++i) p[i] = r.v[i];
    }
}

// feature rows: fp32, or bf16 on the bf16 feature path (math stays fp32 either way).  idx in elements.
template <int VPL, bool BF>
__device__ __forceinline__ Vec<VPL> load_row(const float* base, size_t idx) {
    if constexpr (!BF) {
        return load_vec<VPL>(base + idx);
    } else {
        const uint16_t* p = reinterpret_cast<const uint16_t*>(base) + idx;
        Vec<VPL> r;
        if constexpr (VPL >= 8) {
#pragma unroll
            for (int c = 0; c < VPL / 8; ++c) {
                const u32x4 w = reinterpret_cast<const u32x4*>(p)[c];
#pragma unroll
                for (int i = 0; i < 4; ++i) r.v[8 * c + 2 * i] = bf16_lo(w[i]), r.v[8 * c + 2 * i + 1] = bf16_hi(w[i]);
            }
        } else if constexpr (VPL == 4) {
            const u32x2 w = *reinterpret_cast<const u32x2*>(p);
            r.v[0] = bf16_lo(w[0]), r.v[1] = bf16_hi(w[0]), r.v[2] = bf16_lo(w[1]), r.v[3] = bf16_hi(w[1]);
        } else {
            static_assert(VPL == 2, "VPL");
            const uint32_t w = *reinterpret_cast<const uint32_t*>(p);
            r.v[0] = bf16_lo(w), r.v[1] = bf16_hi(w);
        }
        return r;
    }
}

template <int VPL, bool BF>
__device__ __forceinline__ void store_row(float* base, size_t idx, const Vec<VPL>& r) {
    if constexpr (!BF) {
        store_vec<VPL>(base + idx, r);
    } else {
        uint16_t* p = reinterpret_cast<uint16_t*>(base) + idx;
        if constexpr (VPL >= 8) {
#pragma unroll
            for (int c = 0; c < VPL / 8; ++c) {
                const u32x4 w = {pack_bf16x2(r.v[8 * c], r.v[8 * c + 1]), pack_bf16x2(r.v[8 * c + 2], r.v[8 * c + 3]),
                                 pack_bf16x2(r.v[8 * c + 4], r.v[8 * c + 5]), pack_bf16x2(r.v[8 * c + 6], r.v[8 * c + 7])};
                reinterpret_cast<u32x4*>(p)[c] = w;
            }
        } else if constexpr (VPL == 4) {
            const u32x2 w = {pack_bf16x2(r.v[0], r.v[1]), pack_bf16x2(r.v[2], r.v[3])};
            *reinterpret_cast<u32x2*>(p) = w;
        } else {
            *reinterpret_cast<uint32_t*>(p) = pack_bf16x2(r.v[0], r.v[1]);
        }
    }
}

// sum over the lanes of one head (lanes_per_head adjacent lanes).  The common case (16 lanes: C = 128,
// 8 channels per lane) is four DPP moves inside a 16-lane row; anything else falls back to shuffles.
__device__ __forceinline__ float head_sum(float s, int lanes_per_head) {
    if (lanes_per_head == 16) {
        s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
        s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
        s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x141, 0xF, 0xF, true));  // row_half_mirror
        s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x140, 0xF, 0xF, true));  // row_mirror
        return s;
    }
    for (int o = lanes_per_head >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    return s;
}

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

// attention output of one target for this lane's VPL channels: relu(out + bias).  The source rows are
// streamed once with an online softmax, FOUR sources per step: their row loads, score dot products and
// per-head reductions are independent chains (the single-source form is one long dependent chain per source
// and the launch is latency bound), then one rescale per step:
//   m' = max(m, s_0..s_3); l = l e^(m-m') + sum_k e^(s_k-m'); acc = acc e^(m-m') + sum_k e^(s_k-m') row_k
// KIND = MEL_CONV_GATV2:       e = att . leaky_relu(x_r[i] + x_l[j]),          out = sum alpha x_l[j]
// KIND = MEL_CONV_TRANSFORMER: e = (q[i] . k[j]) / sqrt(C), k | v side by side, out = sum alpha v[j]
template <int VPL, int KIND, bool BF>
__device__ __forceinline__ Vec<VPL> attend_target(const AttArgs& a, size_t xr_row, uint64_t sources,
                                                  uint64_t smask, int soff, const Vec<VPL>& att,
                                                  const Vec<VPL>& bias, int lane) {
    constexpr int HC = 64 * VPL;
#ifndef MEL_ATT_G
#define MEL_ATT_G 4      // measured 2 / 3 / 4 / 8 sources per step: 24.9 / 25.9 / 25.7 / 29.9 us (conv1, round loop)
#endif
    constexpr int G = MEL_ATT_G;
    const Vec<VPL> xr = load_row<VPL, BF>(a.xr, xr_row * a.ld_r + lane * VPL);
    float m = -INFINITY, l = 0.f;
    Vec<VPL> acc;
#pragma unroll
    for (int i = 0; i < VPL; ++i) acc.v[i] = 0.f;
    while (sources) {                            // TransformerConv adds no self-loop: a target may be isolated
        size_t row[G];                           // element index of this lane's slice of the source row
        bool on[G];
#pragma unroll
        for (int k = 0; k < G; ++k) {
            on[k] = sources != 0;
            const int j = on[k] ? lowest_bit(sources) : 0;
            sources &= sources - 1;              // 0 stays 0
            row[k] = (size_t)(soff + (on[k] ? rank_below(smask, j) : 0)) * a.ld_l + lane * VPL;    // off slots re-read a valid row
        }
        Vec<VPL> xl[G], xv[G];
#pragma unroll
        for (int k = 0; k < G; ++k) {
            xl[k] = load_row<VPL, BF>(a.xl, row[k]);
            if constexpr (KIND == MEL_CONV_TRANSFORMER) xv[k] = load_row<VPL, BF>(a.xl, row[k] + HC);
        }
        float sc_[G];
#pragma unroll
        for (int k = 0; k < G; ++k) {
            float t = 0.f;
            if constexpr (KIND == MEL_CONV_GATV2 && VPL % 2 == 0) {
                // leaky_relu(negative_slope=0.2) as max(z, 0.2 z): same value for every finite z and one op
                // fewer than compare + select; channel pairs so that add / scale / fma issue as v_pk_*_f32
                f32x2 t2 = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < VPL; i += 2) {
                    const f32x2 z = f32x2{xr.v[i], xr.v[i + 1]} + f32x2{xl[k].v[i], xl[k].v[i + 1]};
                    const f32x2 zs = z * 0.2f;
                    const f32x2 zm = {fmaxf(z.x, zs.x), fmaxf(z.y, zs.y)};
                    t2 = __builtin_elementwise_fma(f32x2{att.v[i], att.v[i + 1]}, zm, t2);
                }
                t = t2.x + t2.y;
            } else if constexpr (KIND == MEL_CONV_GATV2) {
#pragma unroll
                for (int i = 0; i < VPL; ++i) {
                    const float z = xr.v[i] + xl[k].v[i];
                    t = fmaf(att.v[i], fmaxf(z, 0.2f * z), t);
                }
            } else {
#pragma unroll
                for (int i = 0; i < VPL; ++i) t = fmaf(xr.v[i], xl[k].v[i], t);
            }
            sc_[k] = t;
        }
#pragma unroll
        for (int k = 0; k < G; ++k) {
            sc_[k] = head_sum(sc_[k], a.lanes_per_head);
            if constexpr (KIND == MEL_CONV_TRANSFORMER) sc_[k] *= a.score_scale;
            if (!on[k]) sc_[k] = -INFINITY;
        }
        float mn = m;
#pragma unroll
        for (int k = 0; k < G; ++k) mn = fmaxf(mn, sc_[k]);
        // e^x as v_exp_f32(x log2 e): ~1e-6 relative on softmax weights that are later normalised (the libm
        // expansion was a quarter of this kernel's VALU work, and the kernel is VALU / latency bound)
        const float rs = fast_exp(m - mn);       // slot 0 is always on, so mn is finite
        float pe[G];
#pragma unroll
        for (int k = 0; k < G; ++k) pe[k] = fast_exp(sc_[k] - mn);   // exp(-inf) = 0 for the off slots
        float ps = 0.f;
#pragma unroll
        for (int k = 0; k < G; ++k) ps += pe[k];
        l = l * rs + ps;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            float t = acc.v[i] * rs;
#pragma unroll
            for (int k = 0; k < G; ++k) t = fmaf(pe[k], (KIND == MEL_CONV_TRANSFORMER ? xv[k].v[i] : xl[k].v[i]), t);
            acc.v[i] = t;
        }
        m = mn;
    }
    const float inv = __builtin_amdgcn_rcpf(l + 1e-16f);
    Vec<VPL> out;
#pragma unroll
    for (int i = 0; i < VPL; ++i) out.v[i] = fmaxf(acc.v[i] * inv + bias.v[i], 0.f);
    return out;
}

template <int VPL>
__device__ __forceinline__ Vec<VPL> load_vec_or_zero(const float* p, int lane) {
    Vec<VPL> r;
    if (p) return load_vec<VPL>(p + lane * VPL);
#pragma unroll
    for (int i = 0; i < VPL; ++i) r.v[i] = 0.f;
    return r;
}

// ATT_ROWS / ATT_SINGLE: ONE WAVEFRONT PER TARGET ROW over the whole batch (envs differ a lot in how many
// targets they have - a workgroup per env leaves the launch waiting for the few crowded envs).
//   ATT_ROWS   conv1 of L-DGN: row r of the U1 list -> h1[r]; the agents' x_1 / x_2 go to the head input
//   ATT_SINGLE conv2 of L-DGN: one target per agent row (only the controlling agent's row can reach its
//              logits, l_dgn.py:135), sources = its closed neighbourhood inside U1 -> x_3
// (256, 2): with the bare bound the register allocator aims at 6 waves per SIMD and SPILLS the source-row
// pointers (88 B of scratch in front of every row load); two blocks per CU lets it keep ~100 VGPRs.
template <int VPL, int MODE, int KIND, bool BF>
__global__ __launch_bounds__(256, 2) void gat_attend_rows_kernel(AttArgs a) {
    const int lane = lane_id();
    const int rows = min(*a.rows_dev, a.rows_cap);
    const Vec<VPL> att = load_vec_or_zero<VPL>(a.att, lane);
    const Vec<VPL> bias = load_vec_or_zero<VPL>(a.bias, lane);
    // grid-stride over the target rows: the grid is sized from the expected row count, not the worst case
    // (a surplus workgroup costs a global-load latency and a CU slot before it can exit)
    // XCD-aware order: consecutive target rows belong to one env and share their source rows, so each XCD
    // (block id % 8 under round-robin dispatch; the grid is a multiple of 8) walks a CONTIGUOUS range of rows
    // and the shared rows hit in that XCD's L2 instead of being fetched by all eight (measured before the
    // remap: 54 % L2 misses in this kernel).
    const int per_xcd = gridDim.x >> 3;
    const int vblock = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int rows_pad = ((rows + 4 * (int)gridDim.x - 1) / (4 * (int)gridDim.x)) * (4 * (int)gridDim.x);
    const int span = rows_pad >> 3;              // rows each XCD owns (multiple of 4 * per_xcd)
    for (int i = (vblock % per_xcd) * 4 + (threadIdx.x >> 6); i < span; i += per_xcd * 4) {
        const int r = (blockIdx.x & 7) * span + i;
        if (r >= rows) continue;
        const TargetDesc d = a.desc[r];          // one 32-byte record: no chain of dependent index loads
        const Vec<VPL> o = attend_target<VPL, KIND, BF>(a, (size_t)r, d.sources, d.smask, d.soff, att, bias, lane);
        if constexpr (MODE == ATT_SINGLE) {
            store_row<VPL, BF>(a.xcat, (size_t)r * a.ld_cat + a.cat_off + lane * VPL, o);
        } else {
            store_row<VPL, BF>(a.out, (size_t)r * a.ldo + lane * VPL, o);
            if (d.cat_row >= 0) {
                const size_t cat = (size_t)d.cat_row * a.ld_cat;
                // x_2: the controlling agent's conv1 row BEFORE the decision-maker mask (l_dgn.py:127)
                store_row<VPL, BF>(a.xcat, cat + a.hidden + lane * VPL, o);
                // x_1: its encoder row (l_dgn.py:122)
                const size_t h0 = (size_t)(d.soff + rank_below(d.smask, d.node)) * a.hidden;
                if constexpr (BF) {
                    uint16_t* dst = reinterpret_cast<uint16_t*>(a.xcat) + cat;
                    const uint16_t* src = reinterpret_cast<const uint16_t*>(a.h0) + h0;
                    for (int c = lane; c < a.hidden; c += 64) dst[c] = src[c];
                } else {
                    for (int c = lane; c < a.hidden; c += 64) a.xcat[cat + c] = a.h0[h0 + c];
                }
            }
        }
    }
}

// ATT_POOL (HL-DGN): one workgroup per env (every env has exactly N targets, so this is balanced):
// conv1 attention for all nodes, decision-maker mask, max / mean / add pool over the graph.
template <int VPL, bool BF>
__global__ __launch_bounds__(256, 2) void gat_attend_pool_kernel(AttArgs a) {
    constexpr int HC = 64 * VPL;
    __shared__ float part[4][HC];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.x;
    const Vec<VPL> att = load_vec<VPL>(a.att + lane * VPL);
    const Vec<VPL> bias = load_vec<VPL>(a.bias + lane * VPL);
    const uint64_t full = (a.n == 64) ? ~0ull : ((1ull << a.n) - 1ull);
    Vec<VPL> pool;
#pragma unroll
    for (int i = 0; i < VPL; ++i) pool.v[i] = (a.aggregator == MEL_AGG_MAX) ? -INFINITY : 0.f;
    for (int t = wave; t < a.n; t += 4) {
        const uint64_t sources = a.adj[(size_t)b * a.n + t] | (1ull << t);
        const Vec<VPL> o = attend_target<VPL, MEL_CONV_GATV2, BF>(a, (size_t)(b * a.n + t), sources, full, b * a.n,
                                                                  att, bias, lane);
        // hl_dgn.py:105-108: mask out non-decision-makers, then pool over the graph
        const float dm = a.obs[(size_t)b * a.obs_stride + t * a.node_cols + a.node_cols - 1];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const float v = o.v[i] * dm;
            pool.v[i] = (a.aggregator == MEL_AGG_MAX) ? fmaxf(pool.v[i], v) : pool.v[i] + v;
        }
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) part[wave][lane * VPL + i] = pool.v[i];
    __syncthreads();
    for (int c = threadIdx.x; c < HC; c += 256) {
        float v;
        if (a.aggregator == MEL_AGG_MAX) {
            v = fmaxf(fmaxf(part[0][c], part[1][c]), fmaxf(part[2][c], part[3][c]));
        } else {
            v = ((part[0][c] + part[1][c]) + part[2][c]) + part[3][c];
            if (a.aggregator == MEL_AGG_MEAN) v /= (float)a.n;
        }
        if constexpr (BF) reinterpret_cast<uint16_t*>(a.pooled)[(size_t)b * HC + c] = (uint16_t)pack_bf16x2(v, 0.f);
        else a.pooled[(size_t)b * HC + c] = v;
    }
}

template <int MODE>
static mel_status launch_attend(const AttArgs& a, int hc, hipStream_t s, const char* what) {
    if constexpr (MODE == ATT_POOL) {
        switch (hc / 64) {
#define MEL_POOL_LAUNCH(V)                                                                              \
    if (a.bf16) hipLaunchKernelGGL((gat_attend_pool_kernel<V, true>), dim3(a.bs), dim3(256), 0, s, a);  \
    else hipLaunchKernelGGL((gat_attend_pool_kernel<V, false>), dim3(a.bs), dim3(256), 0, s, a);
            case 2: MEL_POOL_LAUNCH(2) break;
            case 4: MEL_POOL_LAUNCH(4) break;
            case 8: MEL_POOL_LAUNCH(8) break;
            case 16: MEL_POOL_LAUNCH(16) break;
#undef MEL_POOL_LAUNCH
            default: return fail(MEL_ERR_UNSUPPORTED, "%s: heads*C = %d not in {128,256,512,1024}", what, hc);
        }
    } else {
        long want = ((a.rows_hint > 0 ? a.rows_hint : a.rows_cap) * 5 / 4 + 3) / 4;     // 25 % head-room, loop covers the rest
        if (want > (a.rows_cap + 3) / 4) want = (a.rows_cap + 3) / 4;
        if (want < 256) want = 256;
        const int grid = (int)((want + 7) & ~7L);       // multiple of 8: block id % 8 = XCD
#define MEL_ATT_LAUNCH(V)                                                                                             \
    if (a.kind == MEL_CONV_TRANSFORMER && a.bf16)                                                                     \
        hipLaunchKernelGGL((gat_attend_rows_kernel<V, MODE, MEL_CONV_TRANSFORMER, true>), dim3(grid), dim3(256), 0, s, a);  \
    else if (a.kind == MEL_CONV_TRANSFORMER)                                                                          \
        hipLaunchKernelGGL((gat_attend_rows_kernel<V, MODE, MEL_CONV_TRANSFORMER, false>), dim3(grid), dim3(256), 0, s, a); \
    else if (a.bf16)                                                                                                  \
        hipLaunchKernelGGL((gat_attend_rows_kernel<V, MODE, MEL_CONV_GATV2, true>), dim3(grid), dim3(256), 0, s, a);        \
    else                                                                                                              \
        hipLaunchKernelGGL((gat_attend_rows_kernel<V, MODE, MEL_CONV_GATV2, false>), dim3(grid), dim3(256), 0, s, a);
        switch (hc / 64) {
            case 2: MEL_ATT_LAUNCH(2) break;
            case 4: MEL_ATT_LAUNCH(4) break;
            case 8: MEL_ATT_LAUNCH(8) break;
            case 16: MEL_ATT_LAUNCH(16) break;
            default: return fail(MEL_ERR_UNSUPPORTED, "%s: heads*C = %d not in {128,256,512,1024}", what, hc);
        }
#undef MEL_ATT_LAUNCH
    }
    return check_launch(what);
}

// counter-based uniform stream for the on-device exploration noise
__device__ __forceinline__ uint32_t mix32(uint32_t x) {     // lowbias32 integer hash
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float u01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }


// ------------------------------------------------------------------------------------------------
// dueling tail: last Linear of Q and V + q - mean(q) + v  (l_dgn.py:142-147); one wave per row
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dueling_tail_kernel(const float* __restrict__ hq, int ldq, int kq,
                                                           const float* __restrict__ hv, int ldv, int kv,
                                                           mel_linear q_last, mel_linear v_last, int bs,
                                                           const int32_t* __restrict__ rows_dev, int dueling,
                                                           float* __restrict__ logits, mel_select sel) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= bs || (rows_dev && b >= *rows_dev)) return;
    const int lane = lane_id();
    const int na = q_last.out_dim;
    float q[8];
    float qsum = 0.f;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        q[a] = 0.f;
        if (a < na) {
            float s = 0.f;
            for (int k = lane; k < kq; k += 64) s = fmaf(hq[(size_t)b * ldq + k], q_last.weight[(size_t)a * kq + k], s);
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            q[a] = s + q_last.bias[a];
            qsum += q[a];
        }
    }
    float v = 0.f, mean = 0.f;
    if (dueling) {
        for (int k = lane; k < kv; k += 64) v = fmaf(hv[(size_t)b * ldv + k], v_last.weight[k], v);
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        v += v_last.bias[0];
        mean = qsum / (float)na;
    }
#pragma unroll
    for (int a = 0; a < 8; ++a)
        if (a < na && lane == a) logits[(size_t)b * na + a] = q[a] - mean + v;
    if (sel.act && lane == 0) {                 // fused DQN action selection
        int best = 0;
        float bv = -INFINITY;
#pragma unroll
        for (int a = 0; a < 8; ++a)
            if (a < na && q[a] - mean + v > bv) bv = q[a] - mean + v, best = a;
        if (sel.eps > 0.f) {
            const uint32_t step = sel.step_dev ? *sel.step_dev : 0u;
            const uint32_t base = mix32(sel.seed ^ mix32(step * 0x9e3779b9U + (uint32_t)b));
            if (u01(base) < sel.eps) {
                best = 0, bv = -1.f;
                for (int a = 0; a < na; ++a) {
                    const float u = u01(mix32(base + 0x85ebca6bU * (uint32_t)(a + 1)));
                    if (u > bv) bv = u, best = a;
                }
            }
        }
        sel.act[b] = best;
    }
}

// ------------------------------------------------------------------------------------------------
// [3P] DQNPolicy.forward / exploration_noise
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void minmax_kernel(const float* __restrict__ x, long count, float* out) {
    __shared__ float smin[16], smax[16];
    float lo = INFINITY, hi = -INFINITY;
    for (long i = threadIdx.x; i < count; i += 1024) {
        lo = fminf(lo, x[i]);
        hi = fmaxf(hi, x[i]);
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o, 64));
        hi = fmaxf(hi, __shfl_xor(hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) smin[threadIdx.x >> 6] = lo, smax[threadIdx.x >> 6] = hi;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) lo = fminf(lo, smin[w]), hi = fmaxf(hi, smax[w]);
        out[0] = lo;
        out[1] = hi;
    }
}

__global__ __launch_bounds__(256) void select_action_kernel(const float* __restrict__ logits,
                                                            const uint8_t* __restrict__ mask, long bs, int na,
                                                            float eps, const float* __restrict__ rand_u,
                                                            const float* __restrict__ rand_q,
                                                            const float* __restrict__ minmax,
                                                            int32_t* __restrict__ act) {
    const long b = (long)blockIdx.x * 256 + threadIdx.x;
    if (b >= bs) return;
    const float shift = mask ? (minmax[0] - minmax[1] - 1.0f) : 0.f;
    int best = 0;
    float bv = -INFINITY;
    for (int a = 0; a < na; ++a) {
        float q = logits[b * na + a];
        if (mask) q = q + (1.0f - (float)mask[b * na + a]) * shift;
        if (q > bv) bv = q, best = a;          // first maximum, as argmax
    }
    if (rand_u && rand_q && rand_u[b] < eps) {
        best = 0, bv = -INFINITY;
        for (int a = 0; a < na; ++a) {
            float q = rand_q[b * na + a];
            if (mask) q += (float)mask[b * na + a];
            if (q > bv) bv = q, best = a;
        }
    }
    act[b] = best;
}

// ------------------------------------------------------------------------------------------------
// row-wise action selection with a counter-based RNG (round-batched loop: row count lives on the device)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void select_rows_kernel(const float* __restrict__ logits,
                                                          const int32_t* __restrict__ logit_row, long rows_cap,
                                                          const int32_t* __restrict__ rows_dev, int na, float eps,
                                                          uint32_t seed, uint32_t step,
                                                          const uint32_t* __restrict__ step_dev,
                                                          int32_t* __restrict__ act) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows_cap || (rows_dev && r >= *rows_dev)) return;
    if (step_dev) step += *step_dev;            // device-side counter: advances under hipGraph replay
    const float* q = logits + (size_t)(logit_row ? logit_row[r] : r) * na;
    int best = 0;
    float bv = -INFINITY;
    for (int a = 0; a < na; ++a)
        if (q[a] > bv) bv = q[a], best = a;
    if (eps > 0.f) {                                   // exploration_noise, on-device stream
        const uint32_t base = mix32(seed ^ mix32(step * 0x9e3779b9U + (uint32_t)r));
        if (u01(base) < eps) {
            best = 0, bv = -1.f;
            for (int a = 0; a < na; ++a) {
                const float u = u01(mix32(base + 0x85ebca6bU * (uint32_t)(a + 1)));
                if (u > bv) bv = u, best = a;
            }
        }
    }
    act[r] = best;
}

__global__ __launch_bounds__(256) void select_envs_kernel(const float* __restrict__ logits,
                                                          const uint64_t* __restrict__ live, long bs, int n, int na,
                                                          float eps, uint32_t seed, const uint32_t* __restrict__ step_dev,
                                                          int32_t* __restrict__ act) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= bs * n) return;
    const long b = t / n;
    const int i = (int)(t - b * n);
    if (!((live[b] >> i) & 1ull)) return;
    const float* q = logits + b * na;
    int best = 0;
    float bv = -INFINITY;
    for (int a = 0; a < na; ++a)
        if (q[a] > bv) bv = q[a], best = a;
    if (eps > 0.f) {
        const uint32_t step = step_dev ? *step_dev : 0u;
        const uint32_t base = mix32(seed ^ mix32(step * 0x9e3779b9U + (uint32_t)(b * 64 + i)));
        if (u01(base) < eps) {
            best = 0, bv = -1.f;
            for (int a = 0; a < na; ++a) {
                const float u = u01(mix32(base + 0x85ebca6bU * (uint32_t)(a + 1)));
                if (u > bv) bv = u, best = a;
            }
        }
    }
    act[t] = best;
}

// ------------------------------------------------------------------------------------------------
// workspace layout
// ------------------------------------------------------------------------------------------------
struct Dims {
    int64_t bs;        // envs / observation rows
    int n;
    int64_t rows_cap;  // most agent rows (AEC: bs; round-batched: up to bs*n)
    int64_t u1_cap;    // most conv1 targets
    int64_t u2_cap;    // most conv1 sources
};

static Dims make_dims(int64_t bs, int n, int64_t rows_cap, bool single_agent) {
    Dims d;
    d.bs = bs, d.n = n, d.rows_cap = rows_cap;
    d.u1_cap = single_agent ? bs * (n < 34 ? n : 34) : bs * n;    // |closed one-hop| <= 33 sources + self
    d.u2_cap = bs * n;
    return d;
}

struct FwdLayout {
    PlanBuffers plan;
    float* h0;      // encoder rows
    float* xl1;     // conv1 lin_l rows (HL-DGN: xl | xr in one [M, 2*HC] matrix)
    float* xr1;
    float* h1;      // conv1 output rows
    float* xl2;
    float* xr2;
    float* xcat;    // head input [rows, latent]
    float* hq[2];   // head hidden ping-pong [rows, qw + vw]
    float* minmax;
    uint16_t* wb;   // bf16 copies of the projection weights (bf16 feature path)
    size_t wb_elems;
    size_t bytes;
};

// the weight matrices the dense projections read, in launch order; fp32 path: the nn.Parameter storages
// themselves, bf16 path: their bf16 copies in the workspace (refreshed by every call)
struct ProjWeights {
    const float* enc1;
    const float* c1l; const float* c1r; const float* c1v;
    const float* c2l; const float* c2r; const float* c2v;
    const float* q[MEL_MAX_HEAD_LAYERS];
    const float* v[MEL_MAX_HEAD_LAYERS];
};

static size_t lin_elems(const mel_linear& l) { return l.weight ? (size_t)l.in_dim * l.out_dim : 0; }

// visits every projection weight: f(const mel_linear&, const float** slot)
template <class F>
static void for_each_projection(const mel_weights* w, ProjWeights& pw, F f) {
    f(w->encoder.layer[1], &pw.enc1);
    f(w->conv1.lin_l, &pw.c1l), f(w->conv1.lin_r, &pw.c1r);
    if (w->conv1.kind == MEL_CONV_TRANSFORMER) f(w->conv1.lin_v, &pw.c1v);
    if (w->model != MEL_MODEL_HLDGN) {
        f(w->conv2.lin_l, &pw.c2l), f(w->conv2.lin_r, &pw.c2r);
        if (w->conv2.kind == MEL_CONV_TRANSFORMER) f(w->conv2.lin_v, &pw.c2v);
    }
    for (int i = 0; i + 1 < w->q_head.n_layers; ++i) f(w->q_head.layer[i], &pw.q[i]);
    if (w->dueling)
        for (int i = 0; i + 1 < w->v_head.n_layers; ++i) f(w->v_head.layer[i], &pw.v[i]);
}

static size_t projection_elems(const mel_weights* w) {
    ProjWeights pw{};
    size_t total = 0;
    for_each_projection(w, pw, [&](const mel_linear& l, const float**) { total += (lin_elems(l) + 7) & ~(size_t)7; });
    return total;
}

static int head_hidden_width(const mel_mlp& m) {
    int w = 0;
    for (int i = 0; i + 1 < m.n_layers; ++i) w = w > m.layer[i].out_dim ? w : m.layer[i].out_dim;
    return w;
}

static FwdLayout carve(const mel_weights* w, const Dims& d, void* ws) {
    FwdLayout L{};
    Carver c(ws);
    const int hidden = w->encoder.layer[1].out_dim;
    const int hc = w->conv1.heads * w->conv1.channels;
    const size_t M = (size_t)d.bs * d.n;
    const size_t R = (size_t)d.rows_cap;
    L.plan.adj = c.take<uint64_t>(M);
    L.plan.live = c.take<uint64_t>(d.bs);
    const int latent = w->q_head.layer[0].in_dim;
    if (w->model != MEL_MODEL_HLDGN) {
        L.plan.u1 = c.take<uint64_t>(d.bs);
        L.plan.u2 = c.take<uint64_t>(d.bs);
        L.plan.cnt = c.take<int32_t>(3 * d.bs);
        L.plan.offL = c.take<int32_t>(d.bs + 1);
        L.plan.off1 = c.take<int32_t>(d.bs + 1);
        L.plan.off2 = c.take<int32_t>(d.bs + 1);
        L.plan.nid2 = c.take<int32_t>(d.u2_cap);
        L.plan.arow1 = c.take<int32_t>(d.u1_cap);
        L.plan.dm1 = c.take<float>(d.u1_cap);
        L.plan.desc1 = c.take<TargetDesc>(d.u1_cap);
        L.plan.desc2 = c.take<TargetDesc>(R);
        L.plan.row_env = c.take<int32_t>(R);
        L.plan.row_agent = c.take<int32_t>(R);
        L.plan.arow_g = c.take<int32_t>(R);
        L.plan.dm_g = c.take<float>(R);
        L.h0 = c.take<float>((size_t)d.u2_cap * hidden);
        const int srcw = (w->conv1.kind == MEL_CONV_TRANSFORMER) ? 2 * hc : hc;     // key | value side by side
        L.xl1 = c.take<float>((size_t)d.u2_cap * srcw);
        L.xr1 = c.take<float>((size_t)d.u1_cap * hc);
        L.h1 = c.take<float>((size_t)d.u1_cap * hc);
        L.xl2 = c.take<float>((size_t)d.u1_cap * srcw);
        L.xr2 = c.take<float>(R * hc);
    } else {
        L.h0 = c.take<float>(M * hidden);
        L.xl1 = c.take<float>(M * 2 * hc);
    }
    L.xcat = c.take<float>(R * latent);
    const int hw = head_hidden_width(w->q_head) + head_hidden_width(w->v_head);
    L.hq[0] = c.take<float>(R * (hw > 0 ? hw : 1));
    L.hq[1] = c.take<float>(R * (hw > 0 ? hw : 1));
    L.minmax = c.take<float>(64);
    L.wb_elems = (w->precision == MEL_PREC_BF16) ? projection_elems(w)
                 : (w->precision == MEL_PREC_F32_SPLIT) ? 3 * projection_elems(w) : 0;
    L.wb = c.take<uint16_t>(L.wb_elems ? L.wb_elems : 8);
    L.bytes = c.off;
    return L;
}

static mel_status validate(const mel_weights* w, int model, int64_t bs, int n, int obs_stride, bool index_col) {
    if (!w) return fail(MEL_ERR_INVALID_ARG, "weights pointer is null");
    if (w->model != model) return fail(MEL_ERR_INVALID_ARG, "weights are for model %d, entry point is for %d", w->model, model);
    if (bs <= 0 || bs > (1 << 24)) return fail(MEL_ERR_INVALID_ARG, "bs=%ld out of range", (long)bs);
    if (n < 1 || n > MEL_MAX_NODES) return fail(MEL_ERR_INVALID_ARG, "n_nodes=%d outside [1, %d]", n, MEL_MAX_NODES);
    if (w->in_dim < 1 || w->in_dim > 8) return fail(MEL_ERR_UNSUPPORTED, "in_dim=%d outside [1, 8]", w->in_dim);
    if (w->precision < MEL_PREC_F32 || w->precision > MEL_PREC_F32_SPLIT) return fail(MEL_ERR_INVALID_ARG, "precision=%d", w->precision);
    const int expected = n * (w->in_dim + 3);
    if (index_col) {
        if (obs_stride - 1 != expected)      // networks/common.py:24-29
            return fail(MEL_ERR_SHAPE, "Expected %d feature cols for nodes, got %d", expected, obs_stride - 1);
    } else if (obs_stride < expected) {
        return fail(MEL_ERR_SHAPE, "Expected at least %d feature cols for nodes, got %d", expected, obs_stride);
    }
    if (w->encoder.n_layers != 2) return fail(MEL_ERR_UNSUPPORTED, "encoder must be a 2-layer MLP");
    const int hidden = w->encoder.layer[1].out_dim;
    if (w->encoder.layer[0].in_dim != w->in_dim || w->encoder.layer[1].in_dim != w->encoder.layer[0].out_dim)
        return fail(MEL_ERR_INVALID_ARG, "encoder layer shapes inconsistent");
    if (w->encoder.layer[0].out_dim > 256)
        return fail(MEL_ERR_UNSUPPORTED, "encoder hidden width %d > 256", w->encoder.layer[0].out_dim);
    if (w->encoder.layer[0].out_dim % 32 || hidden % 64)
        return fail(MEL_ERR_UNSUPPORTED, "encoder widths must be multiples of 64 (got %d, %d)", w->encoder.layer[0].out_dim, hidden);
    const int hc = w->conv1.heads * w->conv1.channels;
    if (hc != 128 && hc != 256 && hc != 512 && hc != 1024)
        return fail(MEL_ERR_UNSUPPORTED, "heads*channels = %d not in {128,256,512,1024}", hc);
    const int lph = hc / 64 ? w->conv1.channels / (hc / 64) : 0;
    if (lph < 1 || (lph & (lph - 1)) || lph * (hc / 64) != w->conv1.channels)
        return fail(MEL_ERR_UNSUPPORTED, "channels per head (%d) must be a power-of-two multiple of %d", w->conv1.channels, hc / 64);
    if (w->conv1.lin_l.in_dim != hidden || w->conv1.lin_l.out_dim != hc || w->conv1.lin_r.out_dim != hc)
        return fail(MEL_ERR_INVALID_ARG, "conv1 projection shapes inconsistent");
    const int want_kind = (model == MEL_MODEL_DGNR) ? MEL_CONV_TRANSFORMER : MEL_CONV_GATV2;
    if (w->conv1.kind != want_kind) return fail(MEL_ERR_INVALID_ARG, "conv1 kind %d does not fit model %d", w->conv1.kind, model);
    if (want_kind == MEL_CONV_GATV2 && (!w->conv1.att || !w->conv1.bias)) return fail(MEL_ERR_INVALID_ARG, "conv1 att/bias null");
    if (want_kind == MEL_CONV_TRANSFORMER && (w->conv1.lin_v.in_dim != hidden || w->conv1.lin_v.out_dim != hc))
        return fail(MEL_ERR_INVALID_ARG, "conv1 value projection shape inconsistent");
    int latent = hc;
    if (model != MEL_MODEL_HLDGN) {
        if (w->conv2.heads != w->conv1.heads || w->conv2.channels != w->conv1.channels || w->conv2.kind != want_kind ||
            w->conv2.lin_l.in_dim != hc || w->conv2.lin_l.out_dim != hc || w->conv2.lin_r.out_dim != hc)
            return fail(MEL_ERR_INVALID_ARG, "conv2 projection shapes inconsistent");
        if (want_kind == MEL_CONV_GATV2 && (!w->conv2.att || !w->conv2.bias)) return fail(MEL_ERR_INVALID_ARG, "conv2 att/bias null");
        if (want_kind == MEL_CONV_TRANSFORMER && (w->conv2.lin_v.in_dim != hc || w->conv2.lin_v.out_dim != hc))
            return fail(MEL_ERR_INVALID_ARG, "conv2 value projection shape inconsistent");
        latent = hidden + 2 * hc;                    // l_dgn.py:44, dgn_r.py:63
    }
    const mel_mlp* heads[2] = {&w->q_head, &w->v_head};
    for (int k = 0; k < (w->dueling ? 2 : 1); ++k) {
        const mel_mlp& m = *heads[k];
        if (m.n_layers < 1 || m.n_layers > MEL_MAX_HEAD_LAYERS) return fail(MEL_ERR_UNSUPPORTED, "dueling head depth %d", m.n_layers);
        if (m.layer[0].in_dim != latent) return fail(MEL_ERR_INVALID_ARG, "dueling head expects %d inputs, network produces %d", m.layer[0].in_dim, latent);
        for (int i = 0; i + 1 < m.n_layers; ++i)
            if (m.layer[i].out_dim % 64 || m.layer[i].in_dim % 32 || m.layer[i + 1].in_dim != m.layer[i].out_dim)
                return fail(MEL_ERR_UNSUPPORTED, "dueling hidden layer %d shape %dx%d unsupported", i, m.layer[i].out_dim, m.layer[i].in_dim);
    }
    if (w->dueling && w->q_head.n_layers != w->v_head.n_layers) return fail(MEL_ERR_UNSUPPORTED, "Q and V heads must have equal depth");
    if (w->q_head.layer[w->q_head.n_layers - 1].out_dim != w->n_actions || w->n_actions > 8 || w->n_actions < 1)
        return fail(MEL_ERR_UNSUPPORTED, "n_actions=%d outside [1, 8] or inconsistent", w->n_actions);
    if (w->dueling && w->v_head.layer[w->v_head.n_layers - 1].out_dim != 1) return fail(MEL_ERR_INVALID_ARG, "V head must end in 1 output");
    return MEL_OK;
}

// fp32: alias the parameters.  bf16: convert all projection weights into L.wb with one launch.
static mel_status resolve_projections(const mel_weights* w, const FwdLayout& L, ProjWeights& pw, hipStream_t s) {
    pw = ProjWeights{};
    if (w->precision == MEL_PREC_F32_SPLIT) {        // [rows][3][K] bf16 planes of every projection weight
        SplitBatch b{};
        size_t off = 0;
        int blocks = 0;
        bool bad = false;
        for_each_projection(w, pw, [&](const mel_linear& l, const float** slot) {
            const size_t cnt = lin_elems(l);
            if (b.n >= CVT_MAX_SEG || cnt % 8 != 0 || l.in_dim % 32 != 0 || !l.weight) { bad = true; return; }
            b.src[b.n] = l.weight, b.dst[b.n] = L.wb + off, b.count[b.n] = (int)cnt, b.K[b.n] = l.in_dim, b.start[b.n] = blocks;
            *slot = reinterpret_cast<const float*>(L.wb + off);
            blocks += (int)((cnt / 4 + 255) / 256);
            off += 3 * ((cnt + 7) & ~(size_t)7);
            ++b.n;
        });
        if (bad) return fail(MEL_ERR_UNSUPPORTED, "split path: a projection weight is null or its shape is unsupported");
        b.start[b.n] = blocks;
        hipLaunchKernelGGL(split_weights_kernel, dim3(blocks), dim3(256), 0, s, b);
        return check_launch("weights -> bf16 planes");
    }
    if (w->precision != MEL_PREC_BF16) {
        for_each_projection(w, pw, [&](const mel_linear& l, const float** slot) { *slot = l.weight; });
        return MEL_OK;
    }
    CvtBatch b{};
    size_t off = 0;
    int blocks = 0;
    bool bad = false;
    for_each_projection(w, pw, [&](const mel_linear& l, const float** slot) {
        const size_t cnt = lin_elems(l);
        if (b.n >= CVT_MAX_SEG || cnt % 8 != 0 || !l.weight) { bad = true; return; }
        b.src[b.n] = l.weight, b.dst[b.n] = L.wb + off, b.count[b.n] = (int)cnt, b.start[b.n] = blocks;
        *slot = reinterpret_cast<const float*>(L.wb + off);
        blocks += (int)((cnt / 8 + 255) / 256);
        off += (cnt + 7) & ~(size_t)7;
        ++b.n;
    });
    if (bad) return fail(MEL_ERR_UNSUPPORTED, "bf16 path: a projection weight is null or its size is not a multiple of 8");
    b.start[b.n] = blocks;
    hipLaunchKernelGGL(cvt_bf16_kernel, dim3(blocks), dim3(256), 0, s, b);
    return check_launch("weights -> bf16");
}

// dueling heads: hidden layers through the GEMM (Q | V stacked along n), last layer + combine in the tail.
// rows_dev (device row count) may be null.
static mel_status run_heads(const mel_weights* w, const ProjWeights& pw, const FwdLayout& L, int64_t rows,
                            const int32_t* rows_dev, long rows_hint, float* logits, hipStream_t s,
                            const mel_select* select = nullptr) {
    const int nl = w->q_head.n_layers;
    const int bf = w->precision == MEL_PREC_BF16;
    const int sp = w->precision == MEL_PREC_F32_SPLIT;
    const float* in_q = L.xcat;
    const float* in_v = L.xcat;
    int ld_q = w->q_head.layer[0].in_dim, ld_v = ld_q;
    for (int i = 0; i + 1 < nl; ++i) {
        StageScope t(MEL_STAGE_HEAD_HIDDEN, s);
        const mel_linear& q = w->q_head.layer[i];
        const mel_linear& v = w->v_head.layer[i];
        float* out = L.hq[i & 1];
        const int ldo = q.out_dim + v.out_dim;
        if (i == 0 && q.in_dim == v.in_dim) {       // shared input: one launch, weights split along n
            GemmArgs g;
            g.A = in_q, g.lda = ld_q;
            g.W = pw.q[i], g.W_hi = pw.v[i], g.bias = q.bias, g.bias_hi = v.bias, g.split_n = q.out_dim;
            g.Y = out, g.ldy = ldo, g.M = (int)rows, g.M_dev = rows_dev, g.N = ldo, g.K = q.in_dim, g.relu = 1;
            g.bf16 = bf, g.split = sp, g.y_f32 = (i + 2 == nl);       // the tail reads fp32
            if (mel_status st = launch_gemm(g, GEMM_MODE_PLAIN, s, "head hidden (Q|V)", rows_hint, 0, 3)) return st;
        } else {
            GemmArgs g[2];
            // element offset of the V half inside a row: in elements of the buffer's type (bf16 halves the bytes)
            const bool in16 = bf, out32 = !bf || (i + 2 == nl);
            g[0].A = in_q, g[0].lda = ld_q, g[0].W = pw.q[i], g[0].bias = q.bias;
            g[0].Y = out, g[0].ldy = ldo, g[0].M = (int)rows, g[0].M_dev = rows_dev, g[0].N = q.out_dim, g[0].K = q.in_dim, g[0].relu = 1;
            g[1].A = in_v, g[1].lda = ld_v, g[1].W = pw.v[i], g[1].bias = v.bias;
            g[1].Y = out32 ? out + q.out_dim : reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(out) + q.out_dim);
            g[1].ldy = ldo, g[1].M = (int)rows, g[1].M_dev = rows_dev, g[1].N = v.out_dim, g[1].K = v.in_dim, g[1].relu = 1;
            g[0].bf16 = g[1].bf16 = bf, g[0].split = g[1].split = sp, g[0].y_f32 = g[1].y_f32 = (bf && out32);
            (void)in16;
            const long hints[2] = {rows_hint, rows_hint};
            if (mel_status st = launch_gemm_group(g, hints, w->dueling ? 2 : 1, s, "Q + V hidden", 3)) return st;
        }
        const bool out16 = bf && !(i + 2 == nl);
        in_q = out, ld_q = ld_v = ldo;
        in_v = out16 ? reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(out) + q.out_dim) : out + q.out_dim;
    }
    if (bf && nl < 2) return fail(MEL_ERR_UNSUPPORTED, "bf16 path needs at least one hidden layer in the heads");
    const mel_linear& ql = w->q_head.layer[nl - 1];
    const mel_linear& vl = w->v_head.layer[nl - 1];
    StageScope t(MEL_STAGE_HEAD_TAIL, s);
    hipLaunchKernelGGL(dueling_tail_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, in_q, ld_q, ql.in_dim, in_v, ld_v,
                       vl.in_dim, ql, vl, (int)rows, rows_dev, w->dueling, logits, select ? *select : mel_select{});
    return check_launch("dueling tail");
}

// L-DGN for a set of controlling agents per env (agent_mask == null: the index column names one)
static mel_status ldgn_forward_impl(const mel_weights* w, const float* obs, const Dims& d, int obs_stride,
                                    const uint64_t* agent_mask, float* logits, int32_t* row_offsets_out,
                                    const mel_select* select, void* workspace, size_t ws_bytes, hipStream_t s) {
    if (!obs || !logits || !workspace) return fail(MEL_ERR_INVALID_ARG, "null obs/logits/workspace");
    const FwdLayout L = carve(w, d, workspace);
    if (ws_bytes < L.bytes) return fail(MEL_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, L.bytes);
    clear_stale_error();
    const int64_t bs = d.bs;
    const int n = d.n;
    const int node_cols = w->in_dim + 3;
    const int hidden = w->encoder.layer[1].out_dim, hc = w->conv1.heads * w->conv1.channels;
    const int latent = hidden + 2 * hc;
    const bool tconv = w->conv1.kind == MEL_CONV_TRANSFORMER;
    const int srcw = tconv ? 2 * hc : hc;          // source-side projection width (key | value)
    const int R = (int)d.rows_cap, U1 = (int)d.u1_cap, U2 = (int)d.u2_cap;
    const int32_t* nL = L.plan.offL + bs;
    const int32_t* n1 = L.plan.off1 + bs;
    const int32_t* n2 = L.plan.off2 + bs;
    // expected sizes of the ragged row lists, for tile selection only.  Measured means on r = 0.2 disc graphs
    // (N = 50): one agent per row: |U1| ~ 0.13 N, |U2| ~ 0.25 N; a whole round (about 0.1 N active agents
    // with overlapping neighbourhoods): |U1| ~ 0.21 N, |U2| ~ 0.32 N.
    const bool single = agent_mask == nullptr;
    const int bf = w->precision == MEL_PREC_BF16;
    const int sp = w->precision == MEL_PREC_F32_SPLIT;
    ProjWeights pw;
    if (mel_status st = resolve_projections(w, L, pw, s)) return st;
    const long hintL = single ? bs : bs * (long)(n < 10 ? 1 : n / 10);
    const long hint1 = single ? bs * (long)(n < 8 ? n : 2 + n / 8) : bs * (long)(n < 5 ? n : 1 + n / 5);
    const long hint2 = single ? bs * (long)(n < 8 ? n : 1 + n / 4) : bs * (long)(n < 3 ? n : 1 + n / 3);

    {
        StageScope t(MEL_STAGE_PLAN, s);
        hipLaunchKernelGGL(plan_masks_kernel, dim3((bs + 3) / 4), dim3(256), 0, s, obs, (int)bs, n, obs_stride, node_cols, agent_mask, L.plan, 1);
        if (mel_status st = check_launch("plan_masks")) return st;
        const int inline_scan = bs <= 8192;           // beyond that the per-wave re-scan (O(bs^2 / 64) loads) loses
        if (!inline_scan) {
            hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(1024), 0, s, (int)bs, L.plan);
            if (mel_status st = check_launch("plan_scan")) return st;
        }
        hipLaunchKernelGGL(plan_lists_kernel, dim3((bs + 3) / 4), dim3(256), 0, s, obs, (int)bs, n, obs_stride, node_cols, L.plan, row_offsets_out,
                           tconv ? 0 : 1, inline_scan);
        if (mel_status st = check_launch("plan_lists")) return st;
    }
    {   // encoder on the U2 rows: relu(W1 relu(W0 x + b0) + b1)      (l_dgn.py:117-118)
        GemmArgs g;
        g.obs = obs, g.obs_width = obs_stride, g.n_nodes = n, g.in_dim = w->in_dim, g.node_cols = node_cols;
        g.nid = L.plan.nid2, g.enc_w = w->encoder.layer[0].weight, g.enc_b = w->encoder.layer[0].bias;
        g.W = pw.enc1, g.bias = w->encoder.layer[1].bias, g.bf16 = bf, g.split = sp;
        g.Y = L.h0, g.ldy = hidden, g.M = U2, g.M_dev = n2, g.N = hidden;
        g.K = w->encoder.layer[0].out_dim, g.relu = 1;
        StageScope t(MEL_STAGE_ENCODER, s);
        if (mel_status st = launch_gemm(g, GEMM_MODE_ENC, s, "encoder", hint2)) return st;
    }
    {   // conv1.lin_l on the U2 rows + conv1.lin_r on the U1 rows, one grouped launch
        GemmArgs g[2];
        g[0].bf16 = g[1].bf16 = bf, g[0].split = g[1].split = sp;
        g[0].A = L.h0, g[0].lda = hidden, g[0].W = pw.c1l, g[0].bias = w->conv1.lin_l.bias;
        g[0].Y = L.xl1, g[0].ldy = srcw, g[0].M = U2, g[0].M_dev = n2, g[0].N = srcw, g[0].K = hidden;
        if (tconv)       // key | value of the sources in one problem (weights split along n)
            g[0].W_hi = pw.c1v, g[0].bias_hi = w->conv1.lin_v.bias, g[0].split_n = hc;
        g[1].A = L.h0, g[1].lda = hidden, g[1].arow = L.plan.arow1;
        g[1].W = pw.c1r, g[1].bias = w->conv1.lin_r.bias;
        g[1].Y = L.xr1, g[1].ldy = hc, g[1].M = U1, g[1].M_dev = n1, g[1].N = hc, g[1].K = hidden;
        const long hints[2] = {hint2, hint1};
        StageScope t(MEL_STAGE_CONV1_LIN, s);
        if (mel_status st = launch_gemm_group(g, hints, 2, s, "conv1.lin_l + lin_r", 1)) return st;
    }
    {   // conv1 attention for the U1 targets; also drops x_1 and x_2 of every agent into the head input
        AttArgs a{};
        a.xl = L.xl1, a.ld_l = srcw, a.xr = L.xr1, a.ld_r = hc, a.att = w->conv1.att, a.bias = w->conv1.bias;
        a.kind = w->conv1.kind, a.score_scale = 1.0f / sqrtf((float)w->conv1.channels), a.bf16 = bf;
        a.adj = L.plan.adj, a.live = L.plan.live, a.smask = L.plan.u2;
        a.soff = L.plan.off2, a.loff = L.plan.offL, a.bs = (int)bs, a.n = n;
        a.desc = L.plan.desc1, a.rows_dev = n1, a.rows_cap = U1, a.rows_hint = hint1;
        a.lanes_per_head = w->conv1.channels / (hc / 64);
        a.out = L.h1, a.ldo = hc, a.xcat = L.xcat, a.ld_cat = latent, a.hidden = hidden, a.h0 = L.h0;
        StageScope t(MEL_STAGE_CONV1_ATT, s);
        if (mel_status st = launch_attend<ATT_ROWS>(a, hc, s, "conv1 attention")) return st;
    }
    {   // conv2.lin_l on the U1 rows + conv2.lin_r on the agent rows, one grouped launch; the decision-maker
        // mask (l_dgn.py:128) rides along as a row scale
        GemmArgs g[2];
        g[0].bf16 = g[1].bf16 = bf, g[0].split = g[1].split = sp;
        g[0].A = L.h1, g[0].lda = hc, g[0].rscale = L.plan.dm1;
        g[0].W = pw.c2l, g[0].bias = w->conv2.lin_l.bias;
        g[0].Y = L.xl2, g[0].ldy = srcw, g[0].M = U1, g[0].M_dev = n1, g[0].N = srcw, g[0].K = hc;
        if (tconv) g[0].W_hi = pw.c2v, g[0].bias_hi = w->conv2.lin_v.bias, g[0].split_n = hc;
        g[1].A = L.h1, g[1].lda = hc, g[1].arow = L.plan.arow_g, g[1].rscale = L.plan.dm_g;
        g[1].W = pw.c2r, g[1].bias = w->conv2.lin_r.bias;
        g[1].Y = L.xr2, g[1].ldy = hc, g[1].M = R, g[1].M_dev = nL, g[1].N = hc, g[1].K = hc;
        const long hints[2] = {hint1, hintL};
        StageScope t(MEL_STAGE_CONV2_LIN, s);
        if (mel_status st = launch_gemm_group(g, hints, 2, s, "conv2.lin_l + lin_r", 2)) return st;
    }
    {   // conv2 attention, one target per agent row -> x_3
        AttArgs a{};
        a.xl = L.xl2, a.ld_l = srcw, a.xr = L.xr2, a.ld_r = hc, a.att = w->conv2.att, a.bias = w->conv2.bias;
        a.kind = w->conv2.kind, a.score_scale = 1.0f / sqrtf((float)w->conv2.channels), a.bf16 = bf;
        a.adj = L.plan.adj, a.smask = L.plan.u1, a.soff = L.plan.off1;
        a.bs = (int)bs, a.n = n, a.lanes_per_head = w->conv2.channels / (hc / 64);
        a.desc = L.plan.desc2, a.rows_dev = nL, a.rows_cap = R, a.rows_hint = hintL;
        a.xcat = L.xcat, a.ld_cat = latent, a.cat_off = hidden + hc;
        StageScope t(MEL_STAGE_CONV2_ATT, s);
        if (mel_status st = launch_attend<ATT_SINGLE>(a, hc, s, "conv2 attention")) return st;
    }
    return run_heads(w, pw, L, R, nL, hintL, logits, s, select);
}

}  // namespace mel

using namespace mel;

extern "C" {

const char* mel_last_error(void) { return g_err; }
size_t mel_abi_sizeof(int32_t which) {
    switch (which) {
        case 0: return sizeof(mel_linear);
        case 1: return sizeof(mel_gatv2);
        case 2: return sizeof(mel_mlp);
        case 3: return sizeof(mel_weights);
        case 4: return sizeof(mel_select);
        case 5: return sizeof(mel_env_batch);
        case 6: return sizeof(mel_episode_pool);
        case 7: return sizeof(mel_env_obs);
        case 8: return sizeof(mel_round_replay);
        default: return 0;
    }
}
const char* mel_version(void) { return "melissa_hip 0.2 (gfx950)"; }

size_t mel_workspace_bytes(const mel_weights* w, int64_t bs, int32_t n_nodes) {
    if (!w || bs <= 0 || n_nodes < 1 || n_nodes > MEL_MAX_NODES) return 0;
    return carve(w, make_dims(bs, n_nodes, bs, true), nullptr).bytes;
}

size_t mel_workspace_bytes_agents(const mel_weights* w, int64_t bs, int32_t n_nodes, int64_t rows_cap) {
    if (!w || bs <= 0 || n_nodes < 1 || n_nodes > MEL_MAX_NODES || rows_cap < 1) return 0;
    return carve(w, make_dims(bs, n_nodes, rows_cap, false), nullptr).bytes;
}

mel_status mel_ldgn_forward(const mel_weights* w, const float* obs, int64_t bs, int32_t n, int32_t obs_width,
                            float* logits, void* workspace, size_t ws_bytes, void* stream) {
    if (mel_status st = validate(w, MEL_MODEL_LDGN, bs, n, obs_width, true)) return st;
    return ldgn_forward_impl(w, obs, make_dims(bs, n, bs, true), obs_width, nullptr, logits, nullptr, nullptr,
                             workspace, ws_bytes, static_cast<hipStream_t>(stream));
}

mel_status mel_ldgn_forward_agents(const mel_weights* w, const float* obs, int64_t bs, int32_t n, int32_t obs_stride,
                                   const uint64_t* agent_mask, int64_t rows_cap, float* logits,
                                   int32_t* row_offsets, const mel_select* select, void* workspace,
                                   size_t ws_bytes, void* stream) {
    if (mel_status st = validate(w, MEL_MODEL_LDGN, bs, n, obs_stride, false)) return st;
    if (!agent_mask) return fail(MEL_ERR_INVALID_ARG, "agent_mask is null");
    if (rows_cap < 1 || rows_cap > bs * (int64_t)n) return fail(MEL_ERR_INVALID_ARG, "rows_cap=%ld outside [1, bs*n]", (long)rows_cap);
    if (select && !select->act) return fail(MEL_ERR_INVALID_ARG, "select->act is null");
    return ldgn_forward_impl(w, obs, make_dims(bs, n, rows_cap, false), obs_stride, agent_mask, logits, row_offsets,
                             select, workspace, ws_bytes, static_cast<hipStream_t>(stream));
}

mel_status mel_dgnr_forward(const mel_weights* w, const float* obs, int64_t bs, int32_t n, int32_t obs_width,
                            float* logits, void* workspace, size_t ws_bytes, void* stream) {
    if (mel_status st = validate(w, MEL_MODEL_DGNR, bs, n, obs_width, true)) return st;
    return ldgn_forward_impl(w, obs, make_dims(bs, n, bs, true), obs_width, nullptr, logits, nullptr, nullptr,
                             workspace, ws_bytes, static_cast<hipStream_t>(stream));
}

mel_status mel_dgnr_forward_agents(const mel_weights* w, const float* obs, int64_t bs, int32_t n, int32_t obs_stride,
                                   const uint64_t* agent_mask, int64_t rows_cap, float* logits,
                                   int32_t* row_offsets, const mel_select* select, void* workspace,
                                   size_t ws_bytes, void* stream) {
    if (mel_status st = validate(w, MEL_MODEL_DGNR, bs, n, obs_stride, false)) return st;
    if (!agent_mask) return fail(MEL_ERR_INVALID_ARG, "agent_mask is null");
    if (rows_cap < 1 || rows_cap > bs * (int64_t)n) return fail(MEL_ERR_INVALID_ARG, "rows_cap=%ld outside [1, bs*n]", (long)rows_cap);
    if (select && !select->act) return fail(MEL_ERR_INVALID_ARG, "select->act is null");
    return ldgn_forward_impl(w, obs, make_dims(bs, n, rows_cap, false), obs_stride, agent_mask, logits, row_offsets,
                             select, workspace, ws_bytes, static_cast<hipStream_t>(stream));
}

static mel_status hldgn_forward_impl(const mel_weights* w, int32_t aggregator, const float* obs, int64_t bs, int32_t n,
                                     int32_t obs_width, bool index_col, float* logits, void* workspace,
                                     size_t ws_bytes, void* stream) {
    if (mel_status st = validate(w, MEL_MODEL_HLDGN, bs, n, obs_width, index_col)) return st;
    if (aggregator < MEL_AGG_MAX || aggregator > MEL_AGG_ADD) return fail(MEL_ERR_INVALID_ARG, "aggregator %d", aggregator);
    if (!obs || !logits || !workspace) return fail(MEL_ERR_INVALID_ARG, "null obs/logits/workspace");
    const Dims d = make_dims(bs, n, bs, true);
    const FwdLayout L = carve(w, d, workspace);
    if (ws_bytes < L.bytes) return fail(MEL_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, L.bytes);
    hipStream_t s = static_cast<hipStream_t>(stream);
    clear_stale_error();
    const int node_cols = w->in_dim + 3;
    const int hidden = w->encoder.layer[1].out_dim, hc = w->conv1.heads * w->conv1.channels;
    const int M = (int)(bs * n);
    const int bf = w->precision == MEL_PREC_BF16;
    const int sp = w->precision == MEL_PREC_F32_SPLIT;
    ProjWeights pw;
    if (mel_status st = resolve_projections(w, L, pw, s)) return st;

    {
        StageScope t(MEL_STAGE_PLAN, s);
        // hl_dgn.py:108 pools over the whole graph: the controlling index is read (and clamped) but unused
        hipLaunchKernelGGL(plan_masks_kernel, dim3((bs + 3) / 4), dim3(256), 0, s, obs, (int)bs, n, obs_width, node_cols,
                           (const uint64_t*)nullptr, L.plan, index_col ? 0 : -1);
        if (mel_status st = check_launch("plan_masks")) return st;
    }
    {
        GemmArgs g;
        g.obs = obs, g.obs_width = obs_width, g.n_nodes = n, g.in_dim = w->in_dim, g.node_cols = node_cols;
        g.enc_w = w->encoder.layer[0].weight, g.enc_b = w->encoder.layer[0].bias;
        g.W = pw.enc1, g.bias = w->encoder.layer[1].bias, g.bf16 = bf, g.split = sp;
        g.Y = L.h0, g.ldy = hidden, g.M = M, g.N = hidden, g.K = w->encoder.layer[0].out_dim, g.relu = 1;
        StageScope t(MEL_STAGE_ENCODER, s);
        if (mel_status st = launch_gemm(g, GEMM_MODE_ENC, s, "encoder")) return st;
    }
    {   // x_l | x_r for every node in one GEMM (weights split along n)
        GemmArgs g;
        g.A = L.h0, g.lda = hidden;
        g.W = pw.c1l, g.W_hi = pw.c1r, g.bf16 = bf, g.split = sp;
        g.bias = w->conv1.lin_l.bias, g.bias_hi = w->conv1.lin_r.bias, g.split_n = hc;
        g.Y = L.xl1, g.ldy = 2 * hc, g.M = M, g.N = 2 * hc, g.K = hidden;
        StageScope t(MEL_STAGE_CONV1_LIN, s);
        if (mel_status st = launch_gemm(g, GEMM_MODE_PLAIN, s, "conv1.lin_l|lin_r")) return st;
    }
    {
        AttArgs a{};
        a.xl = L.xl1, a.ld_l = 2 * hc, a.ld_r = 2 * hc, a.bf16 = bf;
        a.xr = bf ? reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(L.xl1) + hc) : L.xl1 + hc;
        a.att = w->conv1.att, a.bias = w->conv1.bias, a.adj = L.plan.adj, a.kind = MEL_CONV_GATV2;
        a.bs = (int)bs, a.n = n, a.lanes_per_head = w->conv1.channels / (hc / 64);
        a.obs = obs, a.obs_stride = obs_width, a.node_cols = node_cols, a.aggregator = aggregator;
        a.pooled = L.xcat;
        StageScope t(MEL_STAGE_CONV1_ATT, s);
        if (mel_status st = launch_attend<ATT_POOL>(a, hc, s, "conv1 attention + pool")) return st;
    }
    return run_heads(w, pw, L, bs, nullptr, bs, logits, s);
}

mel_status mel_hldgn_forward(const mel_weights* w, int32_t aggregator, const float* obs, int64_t bs, int32_t n,
                             int32_t obs_width, float* logits, void* workspace, size_t ws_bytes, void* stream) {
    return hldgn_forward_impl(w, aggregator, obs, bs, n, obs_width, true, logits, workspace, ws_bytes, stream);
}

mel_status mel_hldgn_forward_envs(const mel_weights* w, int32_t aggregator, const float* obs, int64_t bs, int32_t n,
                                  int32_t obs_stride, float* logits, void* workspace, size_t ws_bytes, void* stream) {
    return hldgn_forward_impl(w, aggregator, obs, bs, n, obs_stride, false, logits, workspace, ws_bytes, stream);
}

mel_status mel_gemm_f32(const float* A, int32_t lda, const float* W, const float* bias, float* Y, int32_t ldy,
                        int64_t M, int32_t N, int32_t K, int32_t relu, int32_t tile, void* stream) {
    if (!A || !W || !Y || M < 0 || M > (1ll << 30) || lda < K || ldy < N)
        return fail(MEL_ERR_INVALID_ARG, "bad gemm arguments");
    clear_stale_error();
    GemmArgs g;
    g.A = A, g.lda = lda, g.W = W, g.bias = bias, g.Y = Y, g.ldy = ldy, g.M = (int)M, g.N = N, g.K = K, g.relu = relu;
    return launch_gemm(g, GEMM_MODE_PLAIN, static_cast<hipStream_t>(stream), "mel_gemm_f32", -1, tile);
}

mel_status mel_radius_graph(const float* obs, int64_t bs, int32_t n, int32_t obs_stride, int32_t in_dim, uint64_t* adj,
                            void* stream) {
    if (!obs || !adj || bs < 1 || bs > (1 << 24) || n < 1 || n > MEL_MAX_NODES || in_dim < 1 || in_dim > 8 ||
        obs_stride < n * (in_dim + 3))
        return fail(MEL_ERR_INVALID_ARG, "mel_radius_graph: bad arguments");
    clear_stale_error();
    hipLaunchKernelGGL(radius_graph_kernel, dim3((bs + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), obs, (int)bs, n,
                       obs_stride, in_dim + 3, adj);
    return check_launch("mel_radius_graph");
}

mel_status mel_gemm_bf16(const void* A, int32_t lda, const void* W, const float* bias, void* Y, int32_t ldy,
                         int64_t M, int32_t N, int32_t K, int32_t relu, int32_t y_f32, int32_t tile, void* stream) {
    if (!A || !W || !Y || M < 0 || M > (1ll << 30) || lda < K || ldy < N)
        return fail(MEL_ERR_INVALID_ARG, "bad gemm arguments");
    clear_stale_error();
    GemmArgs g;
    g.A = static_cast<const float*>(A), g.lda = lda, g.W = static_cast<const float*>(W), g.bias = bias;
    g.Y = static_cast<float*>(Y), g.ldy = ldy, g.M = (int)M, g.N = N, g.K = K, g.relu = relu, g.bf16 = 1, g.y_f32 = y_f32;
    return launch_gemm(g, GEMM_MODE_PLAIN, static_cast<hipStream_t>(stream), "mel_gemm_bf16", -1, tile);
}

mel_status mel_convert_bf16(const float* src, void* dst, int64_t count, void* stream) {
    if (!src || !dst || count < 0 || count % 8 != 0 || count > (1ll << 30))
        return fail(MEL_ERR_INVALID_ARG, "mel_convert_bf16: count must be a multiple of 8");
    if (count == 0) return MEL_OK;
    clear_stale_error();
    CvtBatch b{};
    b.n = 1, b.src[0] = src, b.dst[0] = static_cast<uint16_t*>(dst), b.count[0] = (int)count, b.start[0] = 0;
    b.start[1] = (int)((count / 8 + 255) / 256);
    hipLaunchKernelGGL(cvt_bf16_kernel, dim3(b.start[1]), dim3(256), 0, static_cast<hipStream_t>(stream), b);
    return check_launch("mel_convert_bf16");
}

mel_status mel_forward_tap(const mel_weights* w, int32_t kind, int64_t bs, int32_t n, int64_t rows_cap,
                           const void* workspace, void* out, void* stream) {
    if (!w || !workspace || !out || bs <= 0 || n < 1 || n > MEL_MAX_NODES) return fail(MEL_ERR_INVALID_ARG, "bad tap arguments");
    const bool single = rows_cap <= 0;
    const Dims d = make_dims(bs, n, single ? bs : rows_cap, single);
    const FwdLayout L = carve(w, d, const_cast<void*>(workspace));
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e;
    if (kind == 0)
        e = hipMemcpyAsync(out, L.plan.adj, (size_t)bs * n * sizeof(uint64_t), hipMemcpyDeviceToDevice, s);
    else if (kind == 1)
        e = hipMemcpyAsync(out, L.xcat, (size_t)d.rows_cap * w->q_head.layer[0].in_dim *
                           (w->precision == MEL_PREC_BF16 ? sizeof(uint16_t) : sizeof(float)), hipMemcpyDeviceToDevice, s);
    else if (kind == 2 && w->model != MEL_MODEL_HLDGN) {
        int32_t* o = static_cast<int32_t*>(out);
        e = hipMemcpyAsync(o, L.plan.off1 + bs, sizeof(int32_t), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(o + 1, L.plan.off2 + bs, sizeof(int32_t), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(o + 2, L.plan.offL + bs, sizeof(int32_t), hipMemcpyDeviceToDevice, s);
    } else
        return fail(MEL_ERR_INVALID_ARG, "unknown tap kind %d", kind);
    if (e != hipSuccess) return fail(MEL_ERR_LAUNCH, "tap copy: %s", hipGetErrorString(e));
    return MEL_OK;
}

void* mel_prof_create(int32_t capacity) {
    if (capacity < 1) return nullptr;
    Profiler* p = new Profiler();
    p->capacity = capacity;
    p->ev = new hipEvent_t[2 * (size_t)capacity];
    p->stage = new int[capacity];
    for (int i = 0; i < 2 * capacity; ++i)
        if (hipEventCreate(&p->ev[i]) != hipSuccess) {
            for (int k = 0; k < i; ++k) (void)hipEventDestroy(p->ev[k]);
            delete[] p->ev;
            delete[] p->stage;
            delete p;
            set_error("hipEventCreate failed");
            return nullptr;
        }
    return p;
}

void mel_prof_destroy(void* prof) {
    Profiler* p = static_cast<Profiler*>(prof);
    if (!p) return;
    if (g_prof == p) g_prof = nullptr;
    for (int i = 0; i < 2 * p->capacity; ++i) (void)hipEventDestroy(p->ev[i]);
    delete[] p->ev;
    delete[] p->stage;
    delete p;
}

void mel_prof_attach(void* prof) { g_prof = static_cast<Profiler*>(prof); }

void mel_prof_reset(void* prof) {
    if (prof) static_cast<Profiler*>(prof)->count = 0;
}

int32_t mel_prof_read(void* prof, double* ms_sum, int64_t* count) {
    Profiler* p = static_cast<Profiler*>(prof);
    if (!p || !ms_sum || !count) return fail(MEL_ERR_INVALID_ARG, "bad profiler arguments");
    for (int k = 0; k < MEL_N_STAGES; ++k) ms_sum[k] = 0.0, count[k] = 0;
    for (int i = 0; i < p->count; ++i) {
        if (hipEventSynchronize(p->ev[2 * i + 1]) != hipSuccess) return fail(MEL_ERR_LAUNCH, "event sync failed");
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p->ev[2 * i], p->ev[2 * i + 1]) != hipSuccess) return fail(MEL_ERR_LAUNCH, "event elapsed failed");
        const int st = p->stage[i];
        if (st >= 0 && st < MEL_N_STAGES) ms_sum[st] += ms, count[st] += 1;
    }
    return p->count;
}

mel_status mel_select_action(const float* logits, const uint8_t* mask, int64_t bs, int32_t na, float eps,
                             const float* rand_u, const float* rand_q, int32_t* act, void* scratch, void* stream) {
    if (!logits || !act || bs <= 0 || na < 1) return fail(MEL_ERR_INVALID_ARG, "bad select_action arguments");
    if (mask && !scratch) return fail(MEL_ERR_INVALID_ARG, "masking needs 8 bytes of scratch");
    hipStream_t s = static_cast<hipStream_t>(stream);
    clear_stale_error();
    StageScope t(MEL_STAGE_SELECT, s);
    if (mask) {
        hipLaunchKernelGGL(minmax_kernel, dim3(1), dim3(1024), 0, s, logits, (long)bs * na, static_cast<float*>(scratch));
        if (mel_status st = check_launch("minmax")) return st;
    }
    hipLaunchKernelGGL(select_action_kernel, dim3((bs + 255) / 256), dim3(256), 0, s, logits, mask, (long)bs, na, eps,
                       rand_u, rand_q, static_cast<const float*>(scratch), act);
    return check_launch("select_action");
}

mel_status mel_select_action_envs(const float* logits, const uint64_t* live, int64_t bs, int32_t n, int32_t na,
                                  float eps, uint32_t seed, const uint32_t* step_dev, int32_t* act, void* stream) {
    if (!logits || !live || !act || bs <= 0 || n < 1 || n > MEL_MAX_NODES || na < 1)
        return fail(MEL_ERR_INVALID_ARG, "bad select_action_envs arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    clear_stale_error();
    StageScope t(MEL_STAGE_SELECT, s);
    hipLaunchKernelGGL(select_envs_kernel, dim3((bs * n + 255) / 256), dim3(256), 0, s, logits, live, (long)bs, n, na, eps,
                       seed, step_dev, act);
    return check_launch("select_action_envs");
}

mel_status mel_select_action_rows(const float* logits, const int32_t* logit_row, int64_t rows_cap,
                                  const int32_t* rows_dev, int32_t na, float eps, uint32_t seed, uint32_t step,
                                  const uint32_t* step_dev, int32_t* act, void* stream) {
    if (!logits || !act || rows_cap <= 0 || na < 1) return fail(MEL_ERR_INVALID_ARG, "bad select_action_rows arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    clear_stale_error();
    StageScope t(MEL_STAGE_SELECT, s);
    hipLaunchKernelGGL(select_rows_kernel, dim3((rows_cap + 255) / 256), dim3(256), 0, s, logits, logit_row, (long)rows_cap,
                       rows_dev, na, eps, seed, step, step_dev, act);
    return check_launch("select_action_rows");
}

}  // extern "C"
