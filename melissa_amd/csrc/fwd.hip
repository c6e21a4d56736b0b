// L-DGN / HL-DGN forward for MI355X (gfx950): plan (obs unpack + fp32 radius adjacency + receptive
// field), fp32-MFMA row GEMMs (gemm_f32.hpp), GATv2 edge-softmax/aggregate, segmented pool, dueling
// tail, DQN action selection.  Reference semantics: networks/common.py:6-64, l_dgn.py:92-151,
// hl_dgn.py:82-119 and SURVEY.md Appendix A for the third-party operators.
#include <cstdlib>
#include "common.hpp"
#include "gemm_f32.hpp"
#include "gemm_bf16.hpp"
#include "gemm_split.hpp"
#include "gemm_ring.hpp"

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
        MEL_LAUNCH((gemm_f32_kernel<WM, WN, TM, TN, GEMM_MODE_ENC>), dim3(total), dim3(64 * WM * WN), 0, s, batch);
    else
        MEL_LAUNCH((gemm_f32_kernel<WM, WN, TM, TN, GEMM_MODE_PLAIN>), dim3(total), dim3(64 * WM * WN), 0, s, batch);
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
        MEL_LAUNCH((gemm_f32_persistent_kernel<WM, WN, TM, TN, GEMM_MODE_ENC>), dim3((int)grid), dim3(64 * WM * WN), 0, s, batch);
    else
        MEL_LAUNCH((gemm_f32_persistent_kernel<WM, WN, TM, TN, GEMM_MODE_PLAIN, TAG>), dim3((int)grid), dim3(64 * WM * WN), 0, s, batch);
}

// specialised-wavefront kernel (gemm_ring.hpp): two 8-wave workgroups per CU
template <int TAG, int BK = GEMM_BK, int WMC = 2>
static void gemm_launch_ring_t(const GemmArgs* gs, int count, hipStream_t s) {
    using Cfg = RingCfg<BK, WMC>;
    GemmBatch batch{};
    batch.count = count;
    long tiles = 0;
    for (int i = 0; i < count; ++i) {
        batch.p[i] = gs[i];
        tiles += ((long)((gs[i].M + Cfg::BM - 1) / Cfg::BM) * (gs[i].N / 64) * (gs[i].ksplit > 1 ? gs[i].ksplit : 1) + 7) & ~7L;
    }
    long grid = 256L * Cfg::WG_PER_CU;
    if (grid > tiles) grid = tiles;
    MEL_LAUNCH((gemm_f32_ring_kernel<TAG, BK, WMC>), dim3((int)grid), dim3(Cfg::THREADS), 0, s, batch);
}

// ragged 64 x 64 launches of the round step, named per call site
static void gemm_launch_ragged(const GemmArgs* gs, int count, hipStream_t s, int tag) {
    // Long-K problems (the heads' first layer, K = 1152: 36 K steps per tile) go to the specialised-wavefront kernel:
    // measured 27.4 vs 30.7 us for the head launches; at K <= 512 the one-role kernel is as fast or faster
    // (conv2 84 vs 85 us, conv1 46 vs 53 us), see NOTES.md.
    bool long_k = true;
    for (int i = 0; i < count; ++i) long_k = long_k && gs[i].K >= 1024 && gs[i].ldy % 4 == 0;
    if (long_k) {
        switch (tag) {
            case 1: gemm_launch_ring_t<1>(gs, count, s); break;
            case 2: gemm_launch_ring_t<2>(gs, count, s); break;
            case 3: gemm_launch_ring_t<3>(gs, count, s); break;
            default: gemm_launch_ring_t<0>(gs, count, s); break;
        }
        return;
    }
    switch (tag) {
        case 1: gemm_launch_persistent<2, 2, 1, 1, 1>(gs, count, GEMM_MODE_PLAIN, s); break;
        case 2: gemm_launch_persistent<2, 2, 1, 1, 2>(gs, count, GEMM_MODE_PLAIN, s); break;
        case 3: gemm_launch_persistent<2, 2, 1, 1, 3>(gs, count, GEMM_MODE_PLAIN, s); break;
        default: gemm_launch_persistent<2, 2, 1, 1, 0>(gs, count, GEMM_MODE_PLAIN, s); break;
    }
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
        tiles += ((long)((gs[i].M + BM - 1) / BM) * (gs[i].N / BN) * (gs[i].ksplit > 1 ? gs[i].ksplit : 1) + 7) & ~7L;
    }
    long grid = 256L * PER_CU;
    if (grid > tiles) grid = tiles;
    if (mode == GEMM_MODE_ENC)
        MEL_LAUNCH((gemm_bf16_kernel<WM, WN, TM, TN, GEMM_MODE_ENC>), dim3((int)grid), dim3(64 * WM * WN), 0, s, batch);
    else
        MEL_LAUNCH((gemm_bf16_kernel<WM, WN, TM, TN, GEMM_MODE_PLAIN>), dim3((int)grid), dim3(64 * WM * WN), 0, s, batch);
}

// split path: a persistent 64 x 64 kernel for small launches, 128 x 128 tiles from MEL_SPLIT_BIG_FROM expected big tiles on
// (both named per call site like the fp32 one)
#ifndef MEL_PLANES_FROM
#define MEL_PLANES_FROM 129          // expected 128 x 256 work items from which conv2 runs on gemm_planes_kernel: up to 256 tiles of
                                     // 128 x 128 the older kernel has a CU per tile (32 us), beyond it doubles up (42 us at 288); this one
                                     // takes 33 us for anything up to 256 items
#endif
#ifndef MEL_SPLIT_BIG_FROM
#define MEL_SPLIT_BIG_FROM 192
#endif
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
        MEL_LAUNCH((gemm_split_kernel<GEMM_MODE_ENC, TAG>), dim3((int)grid), dim3(256), 0, s, batch);
    else
        MEL_LAUNCH((gemm_split_kernel<GEMM_MODE_PLAIN, TAG>), dim3((int)grid), dim3(256), 0, s, batch);
}
// 128 x 128 tiles (gemm_split_big_kernel): two workgroups per CU, PLAIN mode, every N a multiple of 128
template <int TAG>
static void gemm_launch_split_big_t(const GemmArgs* gs, int count, hipStream_t s) {
    GemmBatch batch{};
    batch.count = count;
    long items = 0;
    for (int i = 0; i < count; ++i) {
        batch.p[i] = gs[i];
        items += ((long)((gs[i].M + 127) / 128) * (gs[i].N / 128) * (gs[i].ksplit > 1 ? gs[i].ksplit : 1) + 7) & ~7L;
    }
    long grid = 256L * 2;
    if (grid > items) grid = items;
    MEL_LAUNCH((gemm_split_big_kernel<TAG>), dim3((int)grid), dim3(256), 0, s, batch);
}
static bool split_big_fits(const GemmArgs* gs, int count, int mode) {
    if (mode != GEMM_MODE_PLAIN) return false;
    for (int i = 0; i < count; ++i) {
        const int S = gs[i].ksplit > 1 ? gs[i].ksplit : 1;
        if (gs[i].N % 128 || gs[i].K % (GEMS2_BK * S) || gs[i].K / GEMS2_BK / S < 4 || gs[i].lda % 4) return false;
    }
    return true;
}
// big_tiles: expected 128 x 128 tiles of the launch (from the row hints); < 0: the caller forces the 64 x 64 kernel
static void gemm_launch_split(const GemmArgs* gs, int count, int mode, hipStream_t s, int tag, long big_tiles) {
    if (big_tiles >= MEL_SPLIT_BIG_FROM && split_big_fits(gs, count, mode)) {
        switch (tag) {
            case 2: gemm_launch_split_big_t<2>(gs, count, s); break;
            case 3: gemm_launch_split_big_t<3>(gs, count, s); break;
            default: gemm_launch_split_big_t<0>(gs, count, s); break;
        }
        return;
    }
    switch (tag) {
        case 1: gemm_launch_split_t<1>(gs, count, mode, s); break;
        case 2: gemm_launch_split_t<2>(gs, count, mode, s); break;
        case 3: gemm_launch_split_t<3>(gs, count, mode, s); break;
        default: gemm_launch_split_t<0>(gs, count, mode, s); break;
    }
}

// gemm_planes_kernel (gemm_split.hpp): A as [rows][K / 16][3][16] bf16 planes, W / W_hi as [N / 256][K / 16][256][3][16]
static bool planes_fit(const GemmArgs* gs, int count) {
    long ncols = 0;
    for (int i = 0; i < count; ++i) {
        const GemmArgs& g = gs[i];
        if (g.N % GEMP_BN || g.K % GEMS2_BK || g.K / GEMS2_BK < 4 || g.ldy % 4 || g.rscale || g.ksplit > 1) return false;
        if (g.W_hi && g.split_n % GEMP_BN) return false;
        if ((size_t)g.M * (size_t)g.K * 6 >= ((size_t)1 << 32)) return false;      // 32-bit operand offsets
        ncols += g.N;
    }
    return count >= 1 && count <= GEMM_MAX_GROUP && ncols <= GEMP_BIAS_FLOATS;
}
template <int TAG>
static void gemm_launch_planes_t(const GemmArgs* gs, int count, hipStream_t s) {
    GemmBatch batch{};
    batch.count = count;
    long items = 0;
    for (int i = 0; i < count; ++i) {
        batch.p[i] = gs[i];
        items += ((long)((gs[i].M + 127) / 128) * (gs[i].N / GEMP_BN) + 7) & ~7L;
    }
    const long grid = items < 256 ? items : 256;          // one 768-thread workgroup (146 KB of LDS) per CU
    MEL_LAUNCH((gemm_planes_kernel<TAG>), dim3((int)grid), dim3(768), 0, s, batch);
}
static void gemm_launch_planes(const GemmArgs* gs, int count, hipStream_t s, int tag) {
    switch (tag) {
        case 2: gemm_launch_planes_t<2>(gs, count, s); break;
        default: gemm_launch_planes_t<0>(gs, count, s); break;
    }
}

// gemm_bf16_wide_kernel (gemm_split.hpp): bf16 rows on both sides, 128 x 256 tiles, 64-k stages
static bool bf16_wide_fit(const GemmArgs* gs, int count) {
    long ncols = 0;
    for (int i = 0; i < count; ++i) {
        const GemmArgs& g = gs[i];
        if (!g.bf16 || !g.A || g.N % GEMP_BN || g.K % GEMW_BK || g.K / GEMW_BK < 2 || g.lda % 8 || g.rscale || g.ksplit > 1) return false;
        if (g.W_hi && g.split_n % GEMP_BN) return false;
        if ((size_t)g.M * (size_t)g.lda * 2 >= ((size_t)1 << 32) || (size_t)GEMP_BN * g.K * 2 >= ((size_t)1 << 32)) return false;
        ncols += g.N;
    }
    return count >= 1 && count <= GEMM_MAX_GROUP && ncols <= GEMP_BIAS_FLOATS;
}
template <int TAG>
static void gemm_launch_bf16_wide_t(const GemmArgs* gs, int count, hipStream_t s) {
    GemmBatch batch{};
    batch.count = count;
    long items = 0;
    for (int i = 0; i < count; ++i) {
        batch.p[i] = gs[i];
        items += ((long)((gs[i].M + 127) / 128) * (gs[i].N / GEMP_BN) + 7) & ~7L;
    }
    const long grid = items < 256 ? items : 256;
    MEL_LAUNCH((gemm_bf16_wide_kernel<TAG>), dim3((int)grid), dim3(768), 0, s, batch);
}
static void gemm_launch_bf16_wide(const GemmArgs* gs, int count, hipStream_t s, int tag) {
    switch (tag) {
        case 2: gemm_launch_bf16_wide_t<2>(gs, count, s); break;
        default: gemm_launch_bf16_wide_t<0>(gs, count, s); break;
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
        const long big = force_tile == 1 ? -1 : force_tile == 2 ? (1L << 30) : ((m_hint + 127) / 128) * (g.N / 128);
        if (force_tile == 2 && !split_big_fits(&g, 1, mode)) return fail(MEL_ERR_UNSUPPORTED, "%s: shape does not fit the 128 x 128 split tile", what);
        gemm_launch_split(&g, 1, mode, stream, tag, big);
        return check_launch(what);
    }
    // encoder (ENC producer): a 64 x 128 tile spans the whole hidden width, so the first layer (VALU work inside the
    // A-tile producer) is evaluated once per row instead of once per 64-column tile
    // (fp32: the wide tile wins from about 20 000 rows - HL-DGN's 25 600: 24.8 -> 19.9 us - and loses below - L-DGN's
    // 16 000: 17 -> 19 us)
    const bool enc_wide = mode == GEMM_MODE_ENC && force_tile == 0 && g.N % 128 == 0 && (g.bf16 || m_hint >= 20000);
    if (g.bf16) {
        if (force_tile == 3) {
            if (mode != GEMM_MODE_PLAIN || !bf16_wide_fit(&g, 1))
                return fail(MEL_ERR_UNSUPPORTED, "%s: shape does not fit the 128 x 256 bf16 kernel (N %% 256, K %% 64, lda %% 8)", what);
            gemm_launch_bf16_wide(&g, 1, stream, tag);
        } else
        if (force_tile == 2 && g.N % 128 == 0) gemm_launch_bf16<2, 2, 2, 2>(&g, 1, mode, stream);
        else if (enc_wide) gemm_launch_bf16<2, 2, 1, 2>(&g, 1, mode, stream);
        else gemm_launch_bf16<2, 2, 1, 1>(&g, 1, mode, stream);
        return check_launch(what);
    }
    if (enc_wide) {
        gemm_launch_persistent<2, 2, 1, 2>(&g, 1, mode, stream);
        return check_launch(what);
    }
    if (force_tile == 31 && mode == GEMM_MODE_PLAIN && g.ldy % 4 == 0 && g.K >= 64) {      // specialised-wavefront kernel, 64 x 64
        gemm_launch_ring_t<0>(&g, 1, stream);
        return check_launch(what);
    }
    if (force_tile == 1 || (force_tile >= 2 && g.N % 128 == 0)) {
        switch (force_tile) {
            case 1: gemm_launch_t<2, 2, 1, 1>(&g, 1, mode, stream); break;     //  64 x  64, 4 waves
            case 2: gemm_launch_t<2, 2, 2, 2>(&g, 1, mode, stream); break;     // 128 x 128, 4 waves
            case 11: gemm_launch_persistent<2, 2, 1, 1>(&g, 1, mode, stream); break;   // persistent  64 x  64
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

// Skinny long-K problems (the dueling heads' first layer: 4 820 x 256 outputs over K = 1 152 are 304 tiles of 64 x 64 for
// 512 workgroup slots, one 36-step tile each and half of the slots empty): cut K into S chunks so that the work items
// fill the chip evenly.  Chunk s writes raw partial products to plane s of `parts`; splitk_finish_kernel sums the planes
// in order and applies scale / bias / ReLU.  Model of the launch in K steps of one workgroup: (workgroups sharing a
// CU) x (items per workgroup) x (steps per item + 2 for the hand-over), over the 512 slots of the ring kernel.
int choose_ksplit(const GemmArgs& g, long m_hint, int max_split) {
    if (g.K < 512 || g.ldy % 4 || g.N % 64 || g.K % GEMB_BK) return 1;
    if (g.split) {
        // 128 x 128 split kernel: 512 slots (two workgroups per CU); steps of 16 k, ~4 steps of hand-over per work item
        if (g.N % 128 || g.lda % 4) return 1;
        const long tiles = ((m_hint + 127) / 128) * (g.N / 128);
        const int KT = g.K / GEMS2_BK;
        int best = 1;
        long best_cost = 0;
        for (int S = 1; S <= max_split; ++S) {
            if (KT % S || KT / S < 4) continue;
            const long items = tiles * S, slots = items < 512 ? items : 512;
            const long cost = ((slots + 255) / 256) * ((items + slots - 1) / slots) * (KT / S + 4);
            if (S == 1 || cost < best_cost) best = S, best_cost = cost;
        }
        return best * tiles >= MEL_SPLIT_BIG_FROM ? best : 1;
    }
    const long tiles = ((m_hint + 63) / 64) * (g.N / 64);
    if (g.bf16) {
        // bf16 one-role kernel: 1 024 slots, latency bound at these sizes - items per slot x (steps per item + 2)
        const int KT = g.K / GEMB_BK;
        int best = 1;
        long best_cost = 0;
        for (int S = 1; S <= max_split; ++S) {
            if (KT % S || KT / S < 2) continue;
            const long items = tiles * S, slots = items < 1024 ? items : 1024;
            const long cost = ((items + slots - 1) / slots) * (KT / S + 2);
            if (S == 1 || cost < best_cost) best = S, best_cost = cost;
        }
        return best;
    }
    const int KT = g.K / GEMM_BK;
    int best = 1;
    long best_cost = 0;
    for (int S = 1; S <= max_split; ++S) {
        if (KT % S || KT / S < 2) continue;        // (the hand-over buffer needs two steps between tiles)
        const long items = tiles * S, slots = items < 512 ? items : 512;
        const long cost = ((slots + 255) / 256) * ((items + slots - 1) / slots) * (KT / S + 2);
        if (S == 1 || cost < best_cost) best = S, best_cost = cost;
    }
    return best;
}

mel_status launch_gemm_splitk(const GemmArgs& g, int S, float* parts, long part_stride, hipStream_t stream,
                              const char* what, long m_hint, int tag, bool finish = true) {
    if (g.M <= 0) return MEL_OK;
    if (mel_status st = check_gemm_shape(g, what)) return st;
    if (S < 2 || (g.K / (g.bf16 ? GEMB_BK : g.split ? GEMS2_BK : GEMM_BK)) % S || !parts || part_stride < (long)g.M * g.N)
        return fail(MEL_ERR_INVALID_ARG, "%s: bad split-K request (S=%d)", what, S);
    GemmArgs p = g;
    p.Y = parts, p.ldy = g.N, p.ksplit = S, p.part_stride = part_stride;
    if (g.split) {
        if (!split_big_fits(&p, 1, GEMM_MODE_PLAIN)) return fail(MEL_ERR_UNSUPPORTED, "%s: shape does not fit the 128 x 128 split tile", what);
        gemm_launch_split(&p, 1, GEMM_MODE_PLAIN, stream, tag, 1L << 30);
    } else if (g.bf16) {
        p.y_f32 = 1;
        gemm_launch_bf16<2, 2, 1, 1>(&p, 1, GEMM_MODE_PLAIN, stream);
    } else {
        switch (tag) {
            case 3: gemm_launch_ring_t<3>(&p, 1, stream); break;
            default: gemm_launch_ring_t<0>(&p, 1, stream); break;
        }
    }
    if (mel_status st = check_launch(what)) return st;
    if (!finish) return MEL_OK;               // the caller's next launch sums the planes itself
    SplitKFinish f{parts, part_stride, S, g.N, g.M, g.M_dev, g.bias, g.bias_hi, g.split_n, g.rscale, g.relu, g.Y, g.ldy};
    if (m_hint < 0 || m_hint > g.M) m_hint = g.M;
    long blocks = (m_hint * (g.N / 4) + 255) / 256;
    blocks = blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks;
    MEL_LAUNCH(splitk_finish_kernel, dim3((int)blocks), dim3(256), 0, stream, f);
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
    if (!gs[0].split && !gs[0].bf16 && big >= MEL_SPLIT_BIG_FROM) {
        // MEL_PREC_F32_AUTO: every problem carries the bf16 planes of its weights and the launch is large enough for the
        // 128 x 128 split kernel to win (conv2 in the L-DGN step: 52 against 83 us) - same fp32-accurate results
        GemmArgs t[GEMM_MAX_GROUP];
        bool alt = true;
        for (int i = 0; i < count; ++i) {
            t[i] = gs[i];
            alt = alt && gs[i].Ws && !gs[i].split && !gs[i].bf16 && (!gs[i].W_hi || gs[i].Ws_hi);
            t[i].W = gs[i].Ws, t[i].W_hi = gs[i].W_hi ? gs[i].Ws_hi : nullptr, t[i].split = 1;
        }
        if (alt && split_big_fits(t, count, GEMM_MODE_PLAIN)) {
            gemm_launch_split(t, count, GEMM_MODE_PLAIN, stream, tag, big);
            return check_launch(what);
        }
    }
    if (gs[0].split) {
        for (int i = 1; i < count; ++i)
            if (!gs[i].split) return fail(MEL_ERR_INVALID_ARG, "%s: mixed precisions in one group", what);
        gemm_launch_split(gs, count, GEMM_MODE_PLAIN, stream, tag, big);
        return check_launch(what);
    }
    if (gs[0].bf16) {
        for (int i = 1; i < count; ++i)
            if (!gs[i].bf16) return fail(MEL_ERR_INVALID_ARG, "%s: mixed precisions in one group", what);
        // large launches: 128 x 256 tiles (conv2 of the bf16 feature path: 97 MB of operands through the L2 instead of 0.5 GB)
        static const bool wide_off = getenv("MEL_NO_BF16_WIDE") != nullptr;
        long items = 0;
        for (int i = 0; i < count; ++i) {
            const long h = (hints && hints[i] >= 0 && hints[i] <= gs[i].M) ? hints[i] : gs[i].M;
            items += ((h + 127) / 128) * (gs[i].N / GEMP_BN);
        }
        if (!wide_off && items >= MEL_PLANES_FROM && bf16_wide_fit(gs, count)) gemm_launch_bf16_wide(gs, count, stream, tag);
        else gemm_launch_bf16<2, 2, 1, 1>(gs, count, GEMM_MODE_PLAIN, stream);
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

}  // namespace mel

#include "plan.hpp"
#include "attention.hpp"
#include "heads.hpp"

namespace mel {

// The row lists of a forward and the encoder rows of its node-feature table in ONE launch: the first `gemm_blocks`
// workgroups are 64 x 64 tiles of the (feature-domain) encoder GEMM, the rest run plan_lists.  The two are independent (the
// table depends on the weights only) and each is a latency-bound launch of a few hundred workgroups on its own.
template <int W>
__global__ __launch_bounds__(256, 2) void plan_enc_kernel(PlanListsArgs pa, GemmBatch batch, int gemm_blocks) {
    if ((int)blockIdx.x < gemm_blocks) gemm_f32_tile<2, 2, 1, 1, GEMM_MODE_ENC>(batch, (int)blockIdx.x);
    else plan_lists_body<W>(pa, (int)blockIdx.x - gemm_blocks);
}

// HL-DGN's counterpart: its tuple ids (feature_ids) beside the encoder rows of the table, one launch instead of two
__global__ __launch_bounds__(256, 2) void fid_enc_kernel(const float* __restrict__ obs, int bs, int n, int obs_stride, int node_cols,
                                                         PlanBuffers p, int table_rows, GemmBatch batch, int gemm_blocks) {
    if ((int)blockIdx.x < gemm_blocks) gemm_f32_tile<2, 2, 1, 1, GEMM_MODE_ENC>(batch, (int)blockIdx.x);
    else feature_ids_body(obs, bs, n, obs_stride, node_cols, p, table_rows, (int)blockIdx.x - gemm_blocks);
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
    float* hpart;   // split-K partial planes of the heads' first layer [HEAD_KSPLIT_MAX][rows, qw + vw] (fp32 path)
    float* minmax;
    uint16_t* wb;   // bf16 copies of the projection weights (bf16 feature path)
    size_t wb_elems;
    size_t bytes;
    size_t rows_cap;
};

// the weight matrices the dense projections read, in launch order; fp32 path: the nn.Parameter storages
// themselves, bf16 path: their bf16 copies in the workspace (refreshed by every call)
struct ProjWeights {
    const float* enc1;
    const float* c1l; const float* c1r; const float* c1v;
    const float* c2l; const float* c2r; const float* c2v;
    // conv2's matrices once more in 256-row blocks (planes only): what gemm_planes_kernel reads; null when the shape does not fit
    const float* c2l_blk; const float* c2r_blk; const float* c2v_blk;
    const float* q[MEL_MAX_HEAD_LAYERS];
    const float* v[MEL_MAX_HEAD_LAYERS];
    const ProjWeights* alt;        // MEL_PREC_F32_AUTO: the same matrices as bf16 planes (the slots above hold the fp32 ones)
};

static bool has_planes(const mel_weights* w) { return w->precision == MEL_PREC_F32_SPLIT || w->precision == MEL_PREC_F32_AUTO; }
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

// conv2's projection weights in gemm_planes_kernel's block layout, kept beside the row-major planes (small launches stay on
// the kernels that read those): f(const mel_linear&, const float* ProjWeights::* slot)
static bool conv2_blocks_fit(const mel_weights* w) {
    return w->model != MEL_MODEL_HLDGN && w->conv2.lin_l.weight && w->conv2.lin_l.out_dim % GEMP_BN == 0 &&
           w->conv2.lin_l.in_dim % GEMS2_BK == 0 && w->conv2.lin_l.in_dim / GEMS2_BK >= 4;
}
template <class F>
static void for_each_conv2_block(const mel_weights* w, ProjWeights& pw, F f) {
    if (!conv2_blocks_fit(w)) return;
    f(w->conv2.lin_l, &pw.c2l_blk), f(w->conv2.lin_r, &pw.c2r_blk);
    if (w->conv2.kind == MEL_CONV_TRANSFORMER) f(w->conv2.lin_v, &pw.c2v_blk);
}
static size_t plane_elems(const mel_weights* w) {       // bf16 elements of all planes of a model (row-major + conv2's blocks)
    ProjWeights pw{};
    size_t total = 3 * projection_elems(w);
    for_each_conv2_block(w, pw, [&](const mel_linear& l, const float**) { total += 3 * ((lin_elems(l) + 7) & ~(size_t)7); });
    return total;
}

constexpr int HEAD_KSPLIT_MAX = 4;

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
    const size_t SW = (size_t)set_words(d.n);          // words per node set
    L.plan.adj = c.take<uint64_t>(M * SW);
    L.plan.live = c.take<uint64_t>(d.bs * SW);
    const int latent = w->q_head.layer[0].in_dim;
    // node-feature table mode: the encoder / conv1-projection buffers must also hold the N * 40 table rows
    const size_t T = (size_t)d.n * FEATURE_TUPLES_PER_DEGREE;
    const size_t rows2 = (size_t)d.u2_cap > T ? (size_t)d.u2_cap : T, rows1 = (size_t)d.u1_cap > T ? (size_t)d.u1_cap : T;
    const size_t rowsM = M > T ? M : T;
    if (w->model != MEL_MODEL_HLDGN) {
        L.plan.u1 = c.take<uint64_t>(d.bs * SW);
        L.plan.u2 = c.take<uint64_t>(d.bs * SW);
        L.plan.cnt = c.take<int32_t>(3 * d.bs);
        L.plan.offL = c.take<int32_t>(d.bs + 1);
        L.plan.off1 = c.take<int32_t>(d.bs + 1);
        L.plan.off2 = c.take<int32_t>(d.bs + 1);
        L.plan.nid2 = c.take<int32_t>(d.u2_cap);
        L.plan.arow1 = c.take<int32_t>(d.u1_cap);
        L.plan.dm1 = c.take<float>(d.u1_cap);
        L.plan.desc1 = c.take<char>((size_t)d.u1_cap * target_desc_bytes(d.n));
        L.plan.desc2 = c.take<char>(R * target_desc_bytes(d.n));
        L.plan.row_env = c.take<int32_t>(R);
        L.plan.row_agent = c.take<int32_t>(R);
        L.plan.arow_g = c.take<int32_t>(R);
        L.plan.dm_g = c.take<float>(R);
        L.h0 = c.take<float>(rows2 * hidden);
        const int srcw = (w->conv1.kind == MEL_CONV_TRANSFORMER) ? 2 * hc : hc;     // key | value side by side
        L.xl1 = c.take<float>(rows2 * srcw);
        L.xr1 = c.take<float>(rows1 * hc);
        // (fp32 rows, or - conv2 on gemm_planes_kernel - three bf16 planes: 6 bytes per value)
        L.h1 = c.take<float>(((size_t)d.u1_cap * hc * 3 + 1) / 2);
        L.xl2 = c.take<float>((size_t)d.u1_cap * srcw);
        L.xr2 = c.take<float>(R * hc);
    } else {
        L.h0 = c.take<float>(rowsM * hidden);
        L.xl1 = c.take<float>(rowsM * 2 * hc);
    }
    L.plan.fid = c.take<int32_t>(M);
    L.plan.fbad = c.take<int32_t>(d.bs);
    L.plan.fmeta = c.take<int32_t>(4);
    L.xcat = c.take<float>(R * latent);
    const int hw = head_hidden_width(w->q_head) + head_hidden_width(w->v_head);
    L.hq[0] = c.take<float>(R * (hw > 0 ? hw : 1));
    L.hq[1] = c.take<float>(R * (hw > 0 ? hw : 1));
    L.hpart = c.take<float>(hw > 0 ? (size_t)HEAD_KSPLIT_MAX * R * hw : 8);
    L.minmax = c.take<float>(64);
    L.wb_elems = (w->precision == MEL_PREC_BF16) ? projection_elems(w) : has_planes(w) ? plane_elems(w) : 0;
    if (w->prepared) L.wb_elems = 0;             // the caller holds the converted weights (mel_prepare_weights)
    L.wb = c.take<uint16_t>(L.wb_elems ? L.wb_elems : 8);
    L.bytes = c.off;
    L.rows_cap = R;
    return L;
}

static mel_status validate(const mel_weights* w, int model, int64_t bs, int n, int obs_stride, bool index_col) {
    if (!w) return fail(MEL_ERR_INVALID_ARG, "weights pointer is null");
    if (w->model != model) return fail(MEL_ERR_INVALID_ARG, "weights are for model %d, entry point is for %d", w->model, model);
    if (bs <= 0 || bs > (1 << 24)) return fail(MEL_ERR_INVALID_ARG, "bs=%ld out of range", (long)bs);
    if (n < 1 || n > MEL_MAX_NODES) return fail(MEL_ERR_INVALID_ARG, "n_nodes=%d outside [1, %d]", n, MEL_MAX_NODES);
    if (w->in_dim < 1 || w->in_dim > 8) return fail(MEL_ERR_UNSUPPORTED, "in_dim=%d outside [1, 8]", w->in_dim);
    if (w->precision < MEL_PREC_F32 || w->precision > MEL_PREC_F32_AUTO) return fail(MEL_ERR_INVALID_ARG, "precision=%d", w->precision);
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

// The projection weights as the GEMMs of this precision read them.  fp32: the nn.Parameter storages themselves.  bf16 /
// split: bf16 copies (planes) - taken from the caller's PREPARED buffer (mel_prepare_weights: converted once per weight
// version, nothing launched here) or, when mel_weights.prepared is null, converted into `dst` (the workspace) by one launch
// on every call, which keeps a caller that never prepares correct.  convert = false only lays the pointers out.
static mel_status resolve_projections(const mel_weights* w, uint16_t* dst, ProjWeights& pw, ProjWeights& alt, hipStream_t s,
                                      bool convert) {
    pw = ProjWeights{}, alt = ProjWeights{};
    if (w->precision == MEL_PREC_F32_SPLIT || w->precision == MEL_PREC_F32_AUTO) {
        // [rows][K / 16][3][16] bf16 planes of every projection weight; AUTO: beside the fp32 matrices themselves
        const bool both = w->precision == MEL_PREC_F32_AUTO;
        ProjWeights& planes = both ? alt : pw;
        if (both) {
            for_each_projection(w, pw, [&](const mel_linear& l, const float** slot) { *slot = l.weight; });
            pw.alt = &alt;
        }
        SplitBatch b{};
        size_t off = 0;
        int blocks = 0;
        bool bad = false;
        for_each_projection(w, planes, [&](const mel_linear& l, const float** slot) {
            const size_t cnt = lin_elems(l);
            if (b.n >= CVT_MAX_SEG || cnt % 8 != 0 || l.in_dim % 32 != 0 || !l.weight) { bad = true; return; }
            b.src[b.n] = l.weight, b.dst[b.n] = dst + off, b.count[b.n] = (int)cnt, b.K[b.n] = l.in_dim, b.start[b.n] = blocks;
            b.rb[b.n] = 0;
            *slot = reinterpret_cast<const float*>(dst + off);
            blocks += (int)((cnt / 4 + 255) / 256);
            off += 3 * ((cnt + 7) & ~(size_t)7);
            ++b.n;
        });
        if (!bad)
            for_each_conv2_block(w, planes, [&](const mel_linear& l, const float** slot) {
                const size_t cnt = lin_elems(l);
                if (b.n >= CVT_MAX_SEG || !l.weight || l.out_dim % GEMP_BN) return;       // (the slot stays null: no planes kernel)
                b.src[b.n] = l.weight, b.dst[b.n] = dst + off, b.count[b.n] = (int)cnt, b.K[b.n] = l.in_dim, b.start[b.n] = blocks;
                b.rb[b.n] = GEMP_BN;
                *slot = reinterpret_cast<const float*>(dst + off);
                blocks += (int)((cnt / 4 + 255) / 256);
                off += 3 * ((cnt + 7) & ~(size_t)7);
                ++b.n;
            });
        if (bad && both) {                       // AUTO: shapes the split kernels cannot take simply stay on the exact-fp32 path
            alt = ProjWeights{}, pw.alt = nullptr;
            return MEL_OK;
        }
        if (bad) return fail(MEL_ERR_UNSUPPORTED, "split path: a projection weight is null or its shape is unsupported");
        if (!convert) return MEL_OK;
        b.start[b.n] = blocks;
        MEL_LAUNCH(split_weights_kernel, dim3(blocks), dim3(256), 0, s, b);
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
        b.src[b.n] = l.weight, b.dst[b.n] = dst + off, b.count[b.n] = (int)cnt, b.start[b.n] = blocks;
        *slot = reinterpret_cast<const float*>(dst + off);
        blocks += (int)((cnt / 8 + 255) / 256);
        off += (cnt + 7) & ~(size_t)7;
        ++b.n;
    });
    if (bad) return fail(MEL_ERR_UNSUPPORTED, "bf16 path: a projection weight is null or its size is not a multiple of 8");
    if (!convert) return MEL_OK;
    b.start[b.n] = blocks;
    MEL_LAUNCH(cvt_bf16_kernel, dim3(blocks), dim3(256), 0, s, b);
    return check_launch("weights -> bf16");
}
static size_t prepared_elems(const mel_weights* w) {
    return w->precision == MEL_PREC_BF16 ? projection_elems(w) : has_planes(w) ? plane_elems(w) : 0;
}
static mel_status resolve_projections(const mel_weights* w, const FwdLayout& L, ProjWeights& pw, ProjWeights& alt, hipStream_t s) {
    if (w->prepared && w->precision != MEL_PREC_F32)
        return resolve_projections(w, static_cast<uint16_t*>(const_cast<void*>(w->prepared)), pw, alt, s, false);
    return resolve_projections(w, L.wb, pw, alt, s, true);
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
    // The standard heads (hidden [128, 128] for Q and V, l_dgn.py:66-84 with the CLI defaults): the first layer's RAW
    // products go to fp32 planes (fp32 path: split-K, one plane per K chunk; bf16 / split paths: one plane), then
    // everything after them runs in ONE launch (head_finish_kernel: plane sum + bias + ReLU, hidden layer 1 in fp32 MFMA,
    // last layer, dueling combine, selection).
    if (nl == 3 && w->dueling && w->v_head.n_layers == 3) {
        const mel_linear& q = w->q_head.layer[0];
        const mel_linear& v = w->v_head.layer[0];
        const mel_linear& q1 = w->q_head.layer[1];
        const mel_linear& v1 = w->v_head.layer[1];
        GemmArgs g;
        g.A = in_q, g.lda = ld_q, g.W = pw.q[0], g.W_hi = pw.v[0], g.split_n = q.out_dim;
        g.Y = L.hpart, g.ldy = 2 * HF_W, g.M = (int)rows, g.M_dev = rows_dev, g.N = 2 * HF_W, g.K = q.in_dim;
        g.bf16 = bf, g.split = sp, g.y_f32 = 1;
        const long hint = rows_hint < 0 || rows_hint > rows ? rows : rows_hint;
        int S = choose_ksplit(g, hint, HEAD_KSPLIT_MAX);
        if (pw.alt && pw.alt->q[0] && pw.alt->v[0]) {     // MEL_PREC_F32_AUTO: the split kernel when its work items fill the chip
            GemmArgs t = g;
            t.W = pw.alt->q[0], t.W_hi = pw.alt->v[0], t.split = 1;
            const int St = choose_ksplit(t, hint, HEAD_KSPLIT_MAX);
            const long tiles = ((hint + 127) / 128) * (t.N / 128);
            t.ksplit = St > 1 ? St : 0;
            if ((St > 1 || tiles >= MEL_SPLIT_BIG_FROM) && split_big_fits(&t, 1, GEMM_MODE_PLAIN)) {
                t.ksplit = 0;
                g = t, S = St;
            }
        }
        if (q.in_dim == v.in_dim && q.out_dim == HF_W && v.out_dim == HF_W && q1.in_dim == HF_W &&
            q1.out_dim == HF_W && v1.in_dim == HF_W && v1.out_dim == HF_W && w->q_head.layer[2].out_dim <= HF_MAX_ACTIONS &&
            w->v_head.layer[2].out_dim == 1 && q1.weight && v1.weight) {
            const long ps = (long)L.rows_cap * 2 * HF_W;
            {
                StageScope t(MEL_STAGE_HEAD_HIDDEN, s);
                if (S > 1) {
                    if (mel_status st = launch_gemm_splitk(g, S, L.hpart, ps, s, "head hidden (Q|V), split-K", hint, 3, false)) return st;
                } else if (mel_status st = launch_gemm(g, GEMM_MODE_PLAIN, s, "head hidden (Q|V), raw", hint, 0, 3)) return st;
            }
            StageScope t(MEL_STAGE_HEAD_TAIL, s);
            HeadFinish f{L.hpart, ps, S, (int)rows, rows_dev, q.bias, v.bias, q1, v1, w->q_head.layer[2], w->v_head.layer[2],
                         logits, select ? *select : mel_select{}, 0};
            // grid: twice the expected row count (the hint is a rough mean; a workgroup that has to loop doubles the
            // launch), surplus workgroups leave after one load
            // 16 rows per workgroup while that leaves at most one workgroup per CU, 32 beyond
            const int per = hint > 16 * 256 ? 32 : 16;
            const long need = (rows + per - 1) / per;
            const long likely = (hint + per - 1) / per;
            long blocks = rows_dev ? 2 * likely + 8 : need;
            blocks = blocks > need ? need : blocks < 1 ? 1 : blocks;
            f.likely_blocks = (int)(likely < blocks ? likely : blocks);
            if (per == 16) MEL_LAUNCH(head_finish_kernel<1>, dim3((int)blocks), dim3(512), 0, s, f);
            else MEL_LAUNCH(head_finish_kernel<2>, dim3((int)blocks), dim3(512), 0, s, f);
            return check_launch("head finish");
        }
    }
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
            // (the plane sum writes fp32: on the bf16 path only where the consumer reads fp32)
            const int S = (bf && !g.y_f32) ? 1 : choose_ksplit(g, rows_hint < 0 || rows_hint > rows ? rows : rows_hint, HEAD_KSPLIT_MAX);
            if (S > 1) {
                if (mel_status st = launch_gemm_splitk(g, S, L.hpart, (long)L.rows_cap * ldo, s, "head hidden (Q|V), split-K",
                                                       rows_hint, 3)) return st;
            } else if (mel_status st = launch_gemm(g, GEMM_MODE_PLAIN, s, "head hidden (Q|V)", rows_hint, 0, 3)) return st;
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
    MEL_LAUNCH(dueling_tail_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, in_q, ld_q, ql.in_dim, in_v, ld_v,
                       vl.in_dim, ql, vl, (int)rows, rows_dev, w->dueling, logits, select ? *select : mel_select{});
    return check_launch("dueling tail");
}

// The node-feature table (plan_masks.hpp): encoder row of every feature tuple, then its conv1 projections - two launches
// into t_h0 [T, hidden], t_xl [T, srcw (HL-DGN: 2 hc, lin_l | lin_r)], t_xr [T, hc] (unused for HL-DGN).
struct FeatureTables {
    float* h0;
    float* xl;
    float* xr;
};
static size_t table_elem_bytes(const mel_weights* w) { return w->precision == MEL_PREC_BF16 ? 2 : 4; }
static FeatureTables carve_tables(const mel_weights* w, int n, void* buf, size_t* bytes) {
    const size_t T = (size_t)n * FEATURE_TUPLES_PER_DEGREE, es = table_elem_bytes(w);
    const int hidden = w->encoder.layer[1].out_dim, hc = w->conv1.heads * w->conv1.channels;
    const bool hl = w->model == MEL_MODEL_HLDGN;
    const int srcw = hl ? 2 * hc : (w->conv1.kind == MEL_CONV_TRANSFORMER ? 2 * hc : hc);
    Carver c(buf);
    FeatureTables t;
    t.h0 = reinterpret_cast<float*>(c.take<char>(T * hidden * es));
    t.xl = reinterpret_cast<float*>(c.take<char>(T * srcw * es));
    t.xr = hl ? nullptr : reinterpret_cast<float*>(c.take<char>(T * hc * es));
    if (bytes) *bytes = c.off;
    return t;
}
static mel_status run_feature_tables(const mel_weights* w, const ProjWeights& pw, int n, const FeatureTables& t, hipStream_t s,
                                     bool encoder_done = false) {
    const int T = n * FEATURE_TUPLES_PER_DEGREE;
    const int hidden = w->encoder.layer[1].out_dim, hc = w->conv1.heads * w->conv1.channels;
    const int bf = w->precision == MEL_PREC_BF16, sp = w->precision == MEL_PREC_F32_SPLIT;
    const bool tconv = w->conv1.kind == MEL_CONV_TRANSFORMER, hl = w->model == MEL_MODEL_HLDGN;
    if (!encoder_done) {
        GemmArgs g;
        g.feat_domain = 1, g.in_dim = w->in_dim, g.enc_w = w->encoder.layer[0].weight, g.enc_b = w->encoder.layer[0].bias;
        g.W = pw.enc1, g.bias = w->encoder.layer[1].bias, g.bf16 = bf, g.split = sp;
        // bf16 feature path: the table's encoder rows come from the exact-fp32 tile on the fp32 weights, stored as bf16 - the form
        // the forward's fused launch (plan_enc_kernel / fid_enc_kernel) evaluates, so prepared and per-call tables are identical
        if (bf) g.W = w->encoder.layer[1].weight, g.bf16 = 0, g.y_bf16 = 1;
        g.Y = t.h0, g.ldy = hidden, g.M = T, g.N = hidden, g.K = w->encoder.layer[0].out_dim, g.relu = 1;
        StageScope sc(MEL_STAGE_ENCODER, s);
        if (mel_status st = launch_gemm(g, GEMM_MODE_ENC, s, "encoder (feature tuples)", T)) return st;
    }
    StageScope sc(MEL_STAGE_CONV1_LIN, s);
    if (hl) {
        GemmArgs g;
        g.A = t.h0, g.lda = hidden, g.W = pw.c1l, g.W_hi = pw.c1r, g.bf16 = bf, g.split = sp;
        g.bias = w->conv1.lin_l.bias, g.bias_hi = w->conv1.lin_r.bias, g.split_n = hc;
        g.Y = t.xl, g.ldy = 2 * hc, g.M = T, g.N = 2 * hc, g.K = hidden;
        return launch_gemm(g, GEMM_MODE_PLAIN, s, "conv1.lin_l|lin_r (feature tuples)");
    }
    const int srcw = tconv ? 2 * hc : hc;
    GemmArgs g[2];
    g[0].bf16 = g[1].bf16 = bf, g[0].split = g[1].split = sp;
    g[0].A = t.h0, g[0].lda = hidden, g[0].W = pw.c1l, g[0].bias = w->conv1.lin_l.bias;
    g[0].Y = t.xl, g[0].ldy = srcw, g[0].M = T, g[0].N = srcw, g[0].K = hidden;
    if (tconv) g[0].W_hi = pw.c1v, g[0].bias_hi = w->conv1.lin_v.bias, g[0].split_n = hc;
    g[1].A = t.h0, g[1].lda = hidden, g[1].W = pw.c1r, g[1].bias = w->conv1.lin_r.bias;
    g[1].Y = t.xr, g[1].ldy = hc, g[1].M = T, g[1].N = hc, g[1].K = hidden;
    const long hints[2] = {T, T};
    return launch_gemm_group(g, hints, 2, s, "conv1.lin_l + lin_r (feature tuples)", 1);
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
    ProjWeights pw, pw_alt;
    if (mel_status st = resolve_projections(w, L, pw, pw_alt, s)) return st;
    // round-batched loop, measured on the bench workload (per env, N = 20 / 50): |L| = 3.1 / 4.7, |U1| = 5.8 / 10.4,
    // |U2| = 7.6 / 15.8 - linear fits below
    const long hintL = single ? bs : (n < 10 ? bs : bs * (long)(36 + n) / 18);
    const long hint1 = single ? bs * (long)(n < 8 ? n : 2 + n / 8) : (n < 10 ? bs * (long)n : bs * 2L * (18 + n) / 13);
    const long hint2 = single ? bs * (long)(n < 8 ? n : 1 + n / 4) : (n < 10 ? bs * (long)n : bs * 3L * (8 + n) / 11);
    // Node-feature table (plan_masks.hpp): worth it when the row lists are much longer than the N * 40 tuples
    const int T = n * FEATURE_TUPLES_PER_DEGREE;
    const bool table = (w->flags & MEL_FWD_INTEGER_FEATURES) && w->in_dim == 5 && hint1 + hint2 >= 2L * T;
    bool fused_enc = false;
    // conv2's projections on gemm_planes_kernel (both operands as bf16 planes in blocks, gemm_split.hpp): decided HERE, because
    // the conv1 attention then stores h1 already split, as that kernel's A operand.  Large launches of the fp32-accurate
    // paths only: a launch of fewer work items than CUs stays on the 128 x 128 / 64 x 64 kernels.
    const ProjWeights* planes = sp ? &pw : pw.alt;
    GemmArgs c2[2];
    {
        c2[0].A = L.h1, c2[0].lda = hc, c2[0].bias = w->conv2.lin_l.bias, c2[0].split = 1;
        c2[0].Y = L.xl2, c2[0].ldy = srcw, c2[0].M = U1, c2[0].M_dev = n1, c2[0].N = srcw, c2[0].K = hc;
        if (tconv) c2[0].bias_hi = w->conv2.lin_v.bias, c2[0].split_n = hc;
        c2[1].A = L.h1, c2[1].lda = hc, c2[1].arow = L.plan.arow_g, c2[1].bias = w->conv2.lin_r.bias, c2[1].split = 1;
        c2[1].Y = L.xr2, c2[1].ldy = hc, c2[1].M = R, c2[1].M_dev = nL, c2[1].N = hc, c2[1].K = hc;
        if (planes) {
            c2[0].W = planes->c2l_blk, c2[1].W = planes->c2r_blk;
            if (tconv) c2[0].W_hi = planes->c2v_blk;
        }
    }
    const long c2_items = ((hint1 + 127) / 128) * (srcw / GEMP_BN) + ((hintL + 127) / 128) * (hc / GEMP_BN);
    static const bool planes_off = getenv("MEL_NO_PLANES_GEMM") != nullptr;         // A/B switch for bench and tests
    static const long planes_from = getenv("MEL_PLANES_FROM") ? atol(getenv("MEL_PLANES_FROM")) : MEL_PLANES_FROM;      // (tuning)
    const bool conv2_planes = !planes_off && !bf && planes && c2[0].W && c2[1].W && (!tconv || c2[0].W_hi) && hc % 4 == 0 &&
                              hc / 64 >= 4 && c2_items >= planes_from && planes_fit(c2, 2);

    {
        StageScope t(MEL_STAGE_PLAN, s);
        if (w->flags & MEL_FWD_PLAN_READY) {       // mel_env_round's plan sink wrote the masks of this call
            if (!agent_mask) return fail(MEL_ERR_INVALID_ARG, "MEL_FWD_PLAN_READY needs the agent-set entry points");
        } else {
            if (n > 64) MEL_LAUNCH(plan_masks_kernel<2>, dim3((bs + 3) / 4), dim3(256), 0, s, obs, (int)bs, n, obs_stride, node_cols, agent_mask, L.plan, 1);
            else MEL_LAUNCH(plan_masks_kernel<1>, dim3((bs + 3) / 4), dim3(256), 0, s, obs, (int)bs, n, obs_stride, node_cols, agent_mask, L.plan, 1);
            if (mel_status st = check_launch("plan_masks")) return st;
        }
        const int inline_scan = bs <= 8192;           // beyond that the per-wave re-scan (O(bs^2 / 64) loads) loses
        if (!inline_scan) {
            MEL_LAUNCH(plan_scan_kernel, dim3(1), dim3(1024), 0, s, (int)bs, L.plan);
            if (mel_status st = check_launch("plan_scan")) return st;
        }
        // fp32 + node-feature table: the row lists and the encoder rows of the feature tuples (which depend on the weights
        // only) are independent - ONE launch runs both (plan_enc_kernel) instead of two latency-bound ones back to back
        // (bf16 feature path: the same fp32 tile on the fp32 encoder weights, its rows stored as bf16 - the bf16 ENC launch of the
        //  2 000 tuples alone took 15 us)
        fused_enc = table && !sp && !(w->tables && w->tables_nodes == n);
        const PlanListsArgs pa{obs, (int)bs, n, obs_stride, node_cols, L.plan, row_offsets_out, tconv ? 0 : 1, inline_scan, table ? T : 0};
        if (fused_enc) {
            GemmArgs g;
            g.feat_domain = 1, g.in_dim = w->in_dim, g.enc_w = w->encoder.layer[0].weight, g.enc_b = w->encoder.layer[0].bias;
            g.W = bf ? w->encoder.layer[1].weight : pw.enc1, g.bias = w->encoder.layer[1].bias, g.y_bf16 = bf;
            g.Y = L.h0, g.ldy = hidden, g.M = T, g.N = hidden, g.K = w->encoder.layer[0].out_dim, g.relu = 1;
            if (mel_status st = check_gemm_shape(g, "encoder (feature tuples)")) return st;
            GemmBatch batch{};
            batch.count = 1, batch.p[0] = g, batch.start[0] = 0;
            const int tiles = ((((T + 63) / 64) * (hidden / 64)) + 7) & ~7;
            batch.start[1] = tiles;
            if (n > 64) MEL_LAUNCH(plan_enc_kernel<2>, dim3(tiles + (int)((bs + 3) / 4)), dim3(256), 0, s, pa, batch, tiles);
            else MEL_LAUNCH(plan_enc_kernel<1>, dim3(tiles + (int)((bs + 3) / 4)), dim3(256), 0, s, pa, batch, tiles);
        } else if (n > 64) {
            MEL_LAUNCH(plan_lists_kernel<2>, dim3((bs + 3) / 4), dim3(256), 0, s, pa);
        } else {
            MEL_LAUNCH(plan_lists_kernel<1>, dim3((bs + 3) / 4), dim3(256), 0, s, pa);
        }
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
        if (!table)
            if (mel_status st = launch_gemm(g, GEMM_MODE_ENC, s, "encoder", hint2)) return st;
    }
    // table mode: one row per feature tuple, evaluated here (every call) unless the caller prepared them for these weights
    FeatureTables ft{L.h0, L.xl1, L.xr1};
    if (table) {
        if (w->tables && w->tables_nodes == n) ft = carve_tables(w, n, const_cast<void*>(w->tables), nullptr);
        else if (mel_status st = run_feature_tables(w, pw, n, ft, s, /*encoder_done=*/fused_enc)) return st;
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
        if (!table)
            if (mel_status st = launch_gemm_group(g, hints, 2, s, "conv1.lin_l + lin_r", 1)) return st;
    }
    {   // conv1 attention for the U1 targets; also drops x_1 and x_2 of every agent into the head input
        AttArgs a{};
        a.xl = ft.xl, a.ld_l = srcw, a.xr = ft.xr, a.ld_r = hc, a.att = w->conv1.att, a.bias = w->conv1.bias;
        a.kind = w->conv1.kind, a.score_scale = 1.0f / sqrtf((float)w->conv1.channels), a.bf16 = bf;
        a.adj = L.plan.adj, a.bs = (int)bs, a.n = n;
        a.desc = L.plan.desc1, a.rows_dev = n1, a.rows_cap = U1, a.rows_hint = hint1;
        a.lanes_per_head = w->conv1.channels / (hc / 64);
        a.out = L.h1, a.ldo = hc, a.xcat = L.xcat, a.ld_cat = latent, a.hidden = hidden, a.h0 = ft.h0;
        if (conv2_planes) a.out_planes = reinterpret_cast<uint16_t*>(L.h1);
        a.out_scale = L.plan.dm1;       // the decision-maker mask (l_dgn.py:128) is applied as h1 is stored; x_2 is taken before it
        a.fid = table ? L.plan.fid : nullptr;
        StageScope t(MEL_STAGE_CONV1_ATT, s);
        if (mel_status st = launch_attend<ATT_ROWS>(a, hc, s, "conv1 attention")) return st;
    }
    {   // conv2.lin_l on the U1 rows + conv2.lin_r on the agent rows, one grouped launch (h1 rows are already masked by the
        // decision-maker flag, l_dgn.py:128: the conv1 attention applied it as it stored them)
        GemmArgs g[2];
        g[0].bf16 = g[1].bf16 = bf, g[0].split = g[1].split = sp;
        g[0].A = L.h1, g[0].lda = hc;
        g[0].W = pw.c2l, g[0].bias = w->conv2.lin_l.bias;
        g[0].Y = L.xl2, g[0].ldy = srcw, g[0].M = U1, g[0].M_dev = n1, g[0].N = srcw, g[0].K = hc;
        if (tconv) g[0].W_hi = pw.c2v, g[0].bias_hi = w->conv2.lin_v.bias, g[0].split_n = hc;
        g[1].A = L.h1, g[1].lda = hc, g[1].arow = L.plan.arow_g;
        g[1].W = pw.c2r, g[1].bias = w->conv2.lin_r.bias;
        if (pw.alt) g[0].Ws = pw.alt->c2l, g[0].Ws_hi = tconv ? pw.alt->c2v : nullptr, g[1].Ws = pw.alt->c2r;
        g[1].Y = L.xr2, g[1].ldy = hc, g[1].M = R, g[1].M_dev = nL, g[1].N = hc, g[1].K = hc;
        const long hints[2] = {hint1, hintL};
        StageScope t(MEL_STAGE_CONV2_LIN, s);
        if (conv2_planes) {
            gemm_launch_planes(c2, 2, s, 2);
            if (mel_status st = check_launch("conv2.lin_l + lin_r (planes)")) return st;
        } else if (mel_status st = launch_gemm_group(g, hints, 2, s, "conv2.lin_l + lin_r", 2)) return st;
    }
    {   // conv2 attention, one target per agent row -> x_3
        AttArgs a{};
        a.xl = L.xl2, a.ld_l = srcw, a.xr = L.xr2, a.ld_r = hc, a.att = w->conv2.att, a.bias = w->conv2.bias;
        a.kind = w->conv2.kind, a.score_scale = 1.0f / sqrtf((float)w->conv2.channels), a.bf16 = bf;
        a.adj = L.plan.adj;
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
// tuning builds only (-DMEL_FIN_PROF): read and reset the head finish kernel's cycle counters
void mel_debug_fin_prof(unsigned long long* out5) {
#ifdef MEL_FIN_PROF
    (void)hipMemcpyFromSymbol(out5, HIP_SYMBOL(g_fin_prof), 5 * sizeof(unsigned long long));
    unsigned long long z[5] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fin_prof), z, sizeof(z));
#else
    for (int i = 0; i < 5; ++i) out5[i] = 0;
#endif
}

#ifdef MEL_RING_PROF
// tuning builds only (-DMEL_RING_PROF=<tag>): read and reset the ring kernel's in-kernel cycle counters (tools/ring_prof.py)
void mel_debug_ring_prof(unsigned long long* out8) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_ring_prof), 8 * sizeof(unsigned long long));
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ring_prof), z, sizeof(z));
}
#endif
#ifdef MEL_GEMM_PROF
// tuning builds only: read and reset the one-role persistent kernel's cycle counters (tools/gemm_prof.py)
void mel_debug_gemm_prof(unsigned long long* out8) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_gemm_prof), 8 * sizeof(unsigned long long));
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_prof), z, sizeof(z));
}
#endif
#ifdef MEL_SPLIT_PROF
void mel_debug_split_prof(unsigned long long* out16) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_split_prof), 16 * sizeof(unsigned long long));
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_split_prof), z, sizeof(z));
}
#endif
#ifdef MEL_ATT_PROF
// tuning builds only: read and reset the attention rows kernel's cycle counters (tools/att_prof.py)
void mel_debug_att_prof(unsigned long long* out8) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_att_prof), 8 * sizeof(unsigned long long));
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_att_prof), z, sizeof(z));
}
#endif
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
        case 9: return sizeof(mel_graph_pool);
        case 10: return sizeof(mel_episode_stream);
        case 11: return sizeof(mel_replay_batch);
        case 12: return sizeof(mel_adam_tensors);
        default: return 0;
    }
}
const char* mel_version(void) { return "melissa_hip 0.5 (gfx950)"; }

size_t mel_prepared_weights_bytes(const mel_weights* w) {
    if (!w || w->precision < MEL_PREC_F32 || w->precision > MEL_PREC_F32_AUTO) return 0;
    return prepared_elems(w) * sizeof(uint16_t);
}

size_t mel_feature_tables_bytes(const mel_weights* w, int32_t n_nodes) {
    if (!w || n_nodes < 1 || n_nodes > MEL_MAX_NODES || w->in_dim != 5 || w->encoder.n_layers != 2) return 0;
    size_t bytes = 0;
    (void)carve_tables(w, n_nodes, nullptr, &bytes);
    return bytes;
}

mel_status mel_prepare_feature_tables(const mel_weights* w, int32_t n_nodes, void* tables, size_t bytes, void* stream) {
    if (mel_status st = validate(w, w ? w->model : 0, 1, n_nodes, n_nodes * ((w ? w->in_dim : 5) + 3), false)) return st;
    if (w->in_dim != 5) return fail(MEL_ERR_UNSUPPORTED, "the node-feature table is defined for the 5 GraphEnv features");
    const size_t need = mel_feature_tables_bytes(w, n_nodes);
    if (!tables || bytes < need) return fail(MEL_ERR_WORKSPACE, "feature-table buffer %zu < %zu bytes", bytes, need);
    if (w->precision != MEL_PREC_F32 && !w->prepared)
        return fail(MEL_ERR_INVALID_ARG, "bf16 / split precision: prepare the weights first (mel_prepare_weights)");
    clear_stale_error();
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProjWeights pw, pw_alt;
    FwdLayout none{};
    if (mel_status st = resolve_projections(w, none, pw, pw_alt, s)) return st;
    return run_feature_tables(w, pw, n_nodes, carve_tables(w, n_nodes, tables, nullptr), s);
}

mel_status mel_prepare_weights(const mel_weights* w, void* prepared, size_t bytes, void* stream) {
    if (!w) return fail(MEL_ERR_INVALID_ARG, "weights pointer is null");
    if (w->precision == MEL_PREC_F32) return MEL_OK;                 // the fp32 path reads the parameters themselves
    if (w->precision != MEL_PREC_BF16 && !has_planes(w)) return fail(MEL_ERR_INVALID_ARG, "precision=%d", w->precision);
    const size_t need = prepared_elems(w) * sizeof(uint16_t);
    if (!prepared || bytes < need) return fail(MEL_ERR_WORKSPACE, "prepared-weights buffer %zu < %zu bytes", bytes, need);
    clear_stale_error();
    ProjWeights pw, pw_alt;
    return resolve_projections(w, static_cast<uint16_t*>(prepared), pw, pw_alt, static_cast<hipStream_t>(stream), true);
}

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
                                     size_t ws_bytes, void* stream, const mel_select* select = nullptr) {
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
    ProjWeights pw, pw_alt;
    if (mel_status st = resolve_projections(w, L, pw, pw_alt, s)) return st;
    const int T = n * FEATURE_TUPLES_PER_DEGREE;      // node-feature table (plan_masks.hpp): every node is a row here
    const bool table = (w->flags & MEL_FWD_INTEGER_FEATURES) && w->in_dim == 5 && M >= 2L * T;
    bool fused_enc = false;

    {
        StageScope t(MEL_STAGE_PLAN, s);
        // hl_dgn.py:108 pools over the whole graph: the controlling index is read (and clamped) but unused
        if (!((w->flags & MEL_FWD_PLAN_READY) && !index_col)) {      // (else: written by mel_env_round's plan sink)
            if (n > 64) MEL_LAUNCH(plan_masks_kernel<2>, dim3((bs + 3) / 4), dim3(256), 0, s, obs, (int)bs, n, obs_width, node_cols,
                                   (const uint64_t*)nullptr, L.plan, index_col ? 0 : -1);
            else MEL_LAUNCH(plan_masks_kernel<1>, dim3((bs + 3) / 4), dim3(256), 0, s, obs, (int)bs, n, obs_width, node_cols,
                               (const uint64_t*)nullptr, L.plan, index_col ? 0 : -1);
            if (mel_status st = check_launch("plan_masks")) return st;
        }
        // fp32 + node-feature table evaluated by this call: the tuple ids and the encoder rows of the tuples are independent
        // (the table depends on the weights only) - one launch runs both, as plan_enc_kernel does for L-DGN
        fused_enc = table && !sp && !(w->tables && w->tables_nodes == n);
        if (fused_enc) {
            GemmArgs g;
            g.feat_domain = 1, g.in_dim = w->in_dim, g.enc_w = w->encoder.layer[0].weight, g.enc_b = w->encoder.layer[0].bias;
            g.W = bf ? w->encoder.layer[1].weight : pw.enc1, g.bias = w->encoder.layer[1].bias, g.y_bf16 = bf;
            g.Y = L.h0, g.ldy = hidden, g.M = T, g.N = hidden, g.K = w->encoder.layer[0].out_dim, g.relu = 1;
            if (mel_status st = check_gemm_shape(g, "encoder (feature tuples)")) return st;
            GemmBatch batch{};
            batch.count = 1, batch.p[0] = g, batch.start[0] = 0;
            const int tiles = ((((T + 63) / 64) * (hidden / 64)) + 7) & ~7;
            batch.start[1] = tiles;
            MEL_LAUNCH(fid_enc_kernel, dim3(tiles + (int)((bs + 3) / 4)), dim3(256), 0, s, obs, (int)bs, n, obs_width, node_cols, L.plan, T,
                       batch, tiles);
        } else {
            MEL_LAUNCH(feature_ids_kernel, dim3((bs + 3) / 4), dim3(256), 0, s, obs, (int)bs, n, obs_width, node_cols, L.plan,
                       table ? T : 0);
        }
        if (mel_status st = check_launch("feature_ids")) return st;
    }
    {
        GemmArgs g;
        g.obs = obs, g.obs_width = obs_width, g.n_nodes = n, g.in_dim = w->in_dim, g.node_cols = node_cols;
        g.enc_w = w->encoder.layer[0].weight, g.enc_b = w->encoder.layer[0].bias;
        g.W = pw.enc1, g.bias = w->encoder.layer[1].bias, g.bf16 = bf, g.split = sp;
        g.Y = L.h0, g.ldy = hidden, g.M = M, g.N = hidden, g.K = w->encoder.layer[0].out_dim, g.relu = 1;
        StageScope t(MEL_STAGE_ENCODER, s);
        if (!table)
            if (mel_status st = launch_gemm(g, GEMM_MODE_ENC, s, "encoder")) return st;
    }
    {   // x_l | x_r for every node in one GEMM (weights split along n)
        GemmArgs g;
        g.A = L.h0, g.lda = hidden;
        g.W = pw.c1l, g.W_hi = pw.c1r, g.bf16 = bf, g.split = sp;
        g.bias = w->conv1.lin_l.bias, g.bias_hi = w->conv1.lin_r.bias, g.split_n = hc;
        g.Y = L.xl1, g.ldy = 2 * hc, g.M = M, g.N = 2 * hc, g.K = hidden;
        StageScope t(MEL_STAGE_CONV1_LIN, s);
        if (!table)
            if (mel_status st = launch_gemm(g, GEMM_MODE_PLAIN, s, "conv1.lin_l|lin_r")) return st;
    }
    FeatureTables ft{L.h0, L.xl1, nullptr};
    if (table) {
        if (w->tables && w->tables_nodes == n) ft = carve_tables(w, n, const_cast<void*>(w->tables), nullptr);
        else if (mel_status st = run_feature_tables(w, pw, n, ft, s, /*encoder_done=*/fused_enc)) return st;
    }
    {
        AttArgs a{};
        a.xl = ft.xl, a.ld_l = 2 * hc, a.ld_r = 2 * hc, a.bf16 = bf;
        a.xr = bf ? reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(ft.xl) + hc) : ft.xl + hc;
        a.att = w->conv1.att, a.bias = w->conv1.bias, a.adj = L.plan.adj, a.kind = MEL_CONV_GATV2;
        a.bs = (int)bs, a.n = n, a.lanes_per_head = w->conv1.channels / (hc / 64);
        a.obs = obs, a.obs_stride = obs_width, a.node_cols = node_cols, a.aggregator = aggregator;
        a.pooled = L.xcat, a.fid = table ? L.plan.fid : nullptr;
        StageScope t(MEL_STAGE_CONV1_ATT, s);
        if (mel_status st = launch_attend<ATT_POOL>(a, hc, s, "conv1 attention + pool")) return st;
    }
    return run_heads(w, pw, L, bs, nullptr, bs, logits, s, select);
}

mel_status mel_hldgn_forward(const mel_weights* w, int32_t aggregator, const float* obs, int64_t bs, int32_t n,
                             int32_t obs_width, float* logits, void* workspace, size_t ws_bytes, void* stream) {
    return hldgn_forward_impl(w, aggregator, obs, bs, n, obs_width, true, logits, workspace, ws_bytes, stream);
}

mel_status mel_hldgn_forward_envs(const mel_weights* w, int32_t aggregator, const float* obs, int64_t bs, int32_t n,
                                  int32_t obs_stride, float* logits, void* workspace, size_t ws_bytes, void* stream) {
    return hldgn_forward_impl(w, aggregator, obs, bs, n, obs_stride, false, logits, workspace, ws_bytes, stream);
}

mel_status mel_plan_pointers(const mel_weights* w, int64_t bs, int32_t n, int64_t rows_cap, void* workspace, void** out5) {
    if (!w || !workspace || !out5 || bs <= 0 || n < 1 || n > MEL_MAX_NODES || rows_cap < 0)
        return fail(MEL_ERR_INVALID_ARG, "bad mel_plan_pointers arguments");
    const bool single = rows_cap == 0;
    const FwdLayout L = carve(w, make_dims(bs, n, single ? bs : rows_cap, single), workspace);
    const bool hl = w->model == MEL_MODEL_HLDGN;
    out5[0] = L.plan.adj, out5[1] = hl ? nullptr : (void*)L.plan.live;
    out5[2] = hl ? nullptr : (void*)L.plan.u1, out5[3] = hl ? nullptr : (void*)L.plan.u2, out5[4] = hl ? nullptr : (void*)L.plan.cnt;
    return MEL_OK;
}

mel_status mel_hldgn_forward_envs_select(const mel_weights* w, int32_t aggregator, const float* obs, int64_t bs, int32_t n,
                                         int32_t obs_stride, float* logits, const mel_select* select, void* workspace,
                                         size_t ws_bytes, void* stream) {
    if (select && select->act) {
        if (!select->live || select->n_nodes != n)
            return fail(MEL_ERR_INVALID_ARG, "per-env selection needs select->live and select->n_nodes == n_nodes");
    }
    return hldgn_forward_impl(w, aggregator, obs, bs, n, obs_stride, false, logits, workspace, ws_bytes, stream, select);
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

mel_status mel_gemm_f32_t(const float* A, int32_t lda, int32_t a_t, const float* W, int32_t ldw, int32_t w_t, float* Y, int32_t ldy,
                          int64_t M, int32_t N, int32_t K, void* stream) {
    if (!A || !W || !Y || M < 1 || M > (1ll << 30) || N < 64 || K < 32 || ldy < N)
        return fail(MEL_ERR_INVALID_ARG, "bad transposed-operand gemm arguments");
    if (N % 64 || K % GEMM_BK || lda % 4 || ldw % 4 || (a_t ? (M % 4 || lda < M) : lda < K) || (w_t ? ldw < N : ldw < K) || !w_t)
        return fail(MEL_ERR_UNSUPPORTED, "transposed-operand gemm: N %% 64 == 0, K %% 32 == 0, lda / ldw %% 4 == 0, M %% 4 == 0 with a "
                                         "transposed A, W transposed (the plain form is mel_gemm_f32)");
    clear_stale_error();
    GemmTArgs g{A, W, Y, lda, ldw, ldy, (int)M, N, K};
    const long grid = ((M + 63) / 64) * (N / 64);
    if (grid > (1l << 30)) return fail(MEL_ERR_INVALID_ARG, "transposed-operand gemm: too many tiles");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a_t) MEL_LAUNCH((gemm_f32_t_kernel<true, true>), dim3((int)grid), dim3(256), 0, s, g);
    else MEL_LAUNCH((gemm_f32_t_kernel<false, true>), dim3((int)grid), dim3(256), 0, s, g);
    return check_launch("mel_gemm_f32_t");
}

mel_status mel_gemm_f32_splitk(const float* A, int32_t lda, const float* W, const float* bias, float* Y, int32_t ldy,
                               int64_t M, int32_t N, int32_t K, int32_t relu, int32_t ksplit, float* parts, int64_t parts_floats,
                               void* stream) {
    if (!A || !W || !Y || !parts || M <= 0 || M > (1ll << 30) || lda < K || ldy < N)
        return fail(MEL_ERR_INVALID_ARG, "bad split-K gemm arguments");
    if (ksplit < 2 || K % GEMM_BK || (K / GEMM_BK) % ksplit || (K / GEMM_BK) / ksplit < 2 || N % 64 || ldy % 4 ||
        parts_floats < (int64_t)ksplit * M * N)
        return fail(MEL_ERR_UNSUPPORTED, "split-K gemm: K / 32 = %d must split into %d chunks of >= 2 steps, N %% 64 == 0, "
                                         "ldy %% 4 == 0, parts >= ksplit * M * N floats", K / GEMM_BK, ksplit);
    clear_stale_error();
    GemmArgs g;
    g.A = A, g.lda = lda, g.W = W, g.bias = bias, g.Y = Y, g.ldy = ldy, g.M = (int)M, g.N = N, g.K = K, g.relu = relu;
    return launch_gemm_splitk(g, ksplit, parts, (long)M * N, static_cast<hipStream_t>(stream), "mel_gemm_f32_splitk", -1, 0);
}

mel_status mel_gemm_f32_split(const float* A, int32_t lda, const float* W, const float* bias, float* Y, int32_t ldy,
                              int64_t M, int32_t N, int32_t K, int32_t relu, int32_t tile, int32_t ksplit, void* scratch,
                              int64_t scratch_bytes, void* stream) {
    if (!A || !W || !Y || !scratch || M <= 0 || M > (1ll << 30) || lda < K || ldy < N || N < 64 || K < 128)
        return fail(MEL_ERR_INVALID_ARG, "bad split-precision gemm arguments");
    const int64_t plane_bytes = ((int64_t)6 * N * K + 255) & ~255ll;
    const bool blocks = tile % 100 == 3;                // gemm_planes_kernel: A goes through bf16 planes in scratch as well
    const int64_t a_bytes = blocks ? M * K * 6 : 0;
    const int64_t need = plane_bytes + a_bytes + (ksplit > 1 ? (int64_t)ksplit * M * N * 4 : 0);
    if (K % 32 || N % 64 || scratch_bytes < need)
        return fail(MEL_ERR_UNSUPPORTED, "split-precision gemm: K %% 32 == 0, N %% 64 == 0, scratch >= %lld bytes", (long long)need);
    clear_stale_error();
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (blocks) {
        GemmArgs g;
        g.A = reinterpret_cast<const float*>(static_cast<char*>(scratch) + plane_bytes), g.lda = K;
        g.W = static_cast<const float*>(scratch), g.bias = bias, g.Y = Y, g.ldy = ldy, g.M = (int)M, g.N = N, g.K = K, g.relu = relu, g.split = 1;
        if (ksplit > 1 || lda != K || M * K >= (1ll << 31) || !planes_fit(&g, 1))
            return fail(MEL_ERR_UNSUPPORTED, "split-precision gemm, 128 x 256 planes kernel: N %% 256 == 0, N <= %d, K / 16 >= 4, lda == K, "
                                             "ldy %% 4 == 0, no split-K", GEMP_BIAS_FLOATS);
        SplitBatch b{};
        b.n = 2, b.start[0] = 0;
        b.src[0] = W, b.dst[0] = static_cast<uint16_t*>(scratch), b.count[0] = N * K, b.K[0] = K, b.rb[0] = GEMP_BN;
        b.start[1] = (int)(((int64_t)N * K / 4 + 255) / 256);
        b.src[1] = A, b.dst[1] = reinterpret_cast<uint16_t*>(static_cast<char*>(scratch) + plane_bytes), b.count[1] = (int)(M * K), b.K[1] = K;
        b.rb[1] = 0;
        b.start[2] = b.start[1] + (int)((M * K / 4 + 255) / 256);
        MEL_LAUNCH(split_weights_kernel, dim3(b.start[2]), dim3(256), 0, s, b);
        if (mel_status st = check_launch("operands -> bf16 planes")) return st;
        gemm_launch_planes(&g, 1, s, 0);
        return check_launch("mel_gemm_f32_split (planes)");
    }
    SplitBatch b{};
    b.n = 1, b.src[0] = W, b.dst[0] = static_cast<uint16_t*>(scratch), b.count[0] = N * K, b.K[0] = K, b.start[0] = 0;
    b.start[1] = (int)(((int64_t)N * K / 4 + 255) / 256);
    if (tile < 100) {                                   // tile + 100: the planes of an earlier call are still in scratch
        MEL_LAUNCH(split_weights_kernel, dim3(b.start[1]), dim3(256), 0, s, b);
        if (mel_status st = check_launch("weights -> bf16 planes")) return st;
    } else {
        tile -= 100;
    }
    GemmArgs g;
    g.A = A, g.lda = lda, g.W = static_cast<const float*>(scratch), g.bias = bias, g.Y = Y, g.ldy = ldy, g.M = (int)M, g.N = N, g.K = K;
    g.relu = relu, g.split = 1;
    if (ksplit > 1)
        return launch_gemm_splitk(g, ksplit, reinterpret_cast<float*>(static_cast<char*>(scratch) + plane_bytes), (long)M * N, s,
                                  "mel_gemm_f32_split", -1, 0);
    return launch_gemm(g, GEMM_MODE_PLAIN, s, "mel_gemm_f32_split", -1, tile);
}

mel_status mel_radius_graph(const float* obs, int64_t bs, int32_t n, int32_t obs_stride, int32_t in_dim, uint64_t* adj,
                            void* stream) {
    if (!obs || !adj || bs < 1 || bs > (1 << 24) || n < 1 || n > MEL_MAX_NODES || in_dim < 1 || in_dim > 8 ||
        obs_stride < n * (in_dim + 3))
        return fail(MEL_ERR_INVALID_ARG, "mel_radius_graph: bad arguments");
    clear_stale_error();
    if (n > 64) MEL_LAUNCH(radius_graph_kernel<2>, dim3((bs + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), obs, (int)bs, n,
                           obs_stride, in_dim + 3, adj);
    else MEL_LAUNCH(radius_graph_kernel<1>, dim3((bs + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), obs, (int)bs, n,
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
    MEL_LAUNCH(cvt_bf16_kernel, dim3(b.start[1]), dim3(256), 0, static_cast<hipStream_t>(stream), b);
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
        e = hipMemcpyAsync(out, L.plan.adj, (size_t)bs * n * set_words(n) * sizeof(uint64_t), hipMemcpyDeviceToDevice, s);
    else if (kind == 1)
        e = hipMemcpyAsync(out, L.xcat, (size_t)d.rows_cap * w->q_head.layer[0].in_dim *
                           (w->precision == MEL_PREC_BF16 ? sizeof(uint16_t) : sizeof(float)), hipMemcpyDeviceToDevice, s);
    else if (kind == 2 && w->model != MEL_MODEL_HLDGN) {
        int32_t* o = static_cast<int32_t*>(out);
        e = hipMemcpyAsync(o, L.plan.off1 + bs, sizeof(int32_t), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(o + 1, L.plan.off2 + bs, sizeof(int32_t), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(o + 2, L.plan.offL + bs, sizeof(int32_t), hipMemcpyDeviceToDevice, s);
    } else if (kind == 3) {
        int32_t* o = static_cast<int32_t*>(out);
        e = hipMemcpyAsync(o, L.plan.fmeta, sizeof(int32_t), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(o + 1, L.plan.fbad, (size_t)bs * sizeof(int32_t), hipMemcpyDeviceToDevice, s);
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
        MEL_LAUNCH(minmax_kernel, dim3(1), dim3(1024), 0, s, logits, (long)bs * na, static_cast<float*>(scratch));
        if (mel_status st = check_launch("minmax")) return st;
    }
    MEL_LAUNCH(select_action_kernel, dim3((bs + 255) / 256), dim3(256), 0, s, logits, mask, (long)bs, na, eps,
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
    MEL_LAUNCH(select_envs_kernel, dim3((bs * n + 255) / 256), dim3(256), 0, s, logits, live, (long)bs, n, na, eps,
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
    MEL_LAUNCH(select_rows_kernel, dim3((rows_cap + 255) / 256), dim3(256), 0, s, logits, logit_row, (long)rows_cap,
                       rows_dev, na, eps, seed, step, step_dev, act);
    return check_launch("select_action_rows");
}

}  // extern "C"
