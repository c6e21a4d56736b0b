// Row GEMM on the exact-fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32):
//
//     Y[m, n] = act( rscale[m] * sum_k A(m, k) * W[n, k] + bias[n] )
//
// This is every dense projection of the hot path: the encoder MLP (l_dgn.py:49-54,117), lin_l / lin_r
// of both GATv2Conv layers (l_dgn.py:56-65) and the hidden layers of the dueling heads
// (l_dgn.py:85-86).  Both operands are K-contiguous exactly as PyTorch stores them (activations
// [rows, K], nn.Linear.weight [out, K]), so no transposed copies exist anywhere.
//
// Layout / mapping:
//   * one wavefront owns TM x TN MFMA tiles of 32x32 (16 accumulator VGPRs each); a workgroup is
//     WM x WN wavefronts -> BM = 32*TM*WM rows by BN = 32*TN*WN columns, K step 32.  Two shapes are
//     built: 128x128 (2x2 waves of 64x64) for the big row lists, 64x64 (2x2 waves of 32x32) when the
//     big tile would leave CUs idle (fp32 MFMA is slow enough - 64 cycles per 32x32x2 - that the short
//     per-wave MFMA chain of the small tile, not operand re-use, sets the latency of small problems);
//   * tiles are staged global -> registers -> LDS with 16-byte accesses into a two-stage LDS ring (one
//     barrier per K step); LDS rows are padded to 36 floats (144 B) so the 16 lanes of a ds_read_b128
//     group hit 16 distinct 16-B slots;
//   * the k order inside a K step is permuted identically for A and W (lane half h takes k = 8q+4h..+3
//     of sub-step q), which lets one ds_read_b128 per operand feed four MFMAs;
//   * the next K step's global loads are issued before the current step's MFMAs (register prefetch);
//   * optional A-row gather (arow), per-row scale (the decision-maker mask of l_dgn.py:128 commutes
//     with the projection), split weight/bias pointers (lin_l | lin_r, Q | V stacked along n), and an
//     on-the-fly producer for the first encoder layer (K = in_dim is far too short for an MFMA).
//   * the row count may live on the device (M_dev): ragged receptive-field row lists are sized by a
//     device-side scan, the grid is sized for the worst case and surplus workgroups exit at once.
#pragma once
#include "common.hpp"

namespace mel {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // first-class SSA value: HIP's float4 struct
                                                           // made the staging arrays land in scratch

struct GemmArgs {
    // A operand
    const float* A = nullptr;      // [rows, lda]
    int lda = 0;
    const int32_t* arow = nullptr;  // optional: A row of output row m (identity when null)
    const float* rscale = nullptr;  // optional: per output row scale (applied before the bias)
    // encoder-layer-0 producer (MODE_ENC): A(m, k) = relu(enc_b[k] + sum_f enc_w[k, f] * x_f(node m))
    const float* obs = nullptr;     // [bs, obs_width]
    int obs_width = 0, n_nodes = 0, in_dim = 0, node_cols = 0;
    const int32_t* nid = nullptr;   // optional: global node id (b*N + i) of row m (identity when null)
    int feat_domain = 0;            // 1: row m IS a node-feature tuple (feature_tuple_of, plan_masks.hpp): no obs is read
    const float* enc_w = nullptr;   // [K, in_dim]
    const float* enc_b = nullptr;   // [K]
    // W operand, rows [0, split_n) from W / bias, rows [split_n, N) from W_hi / bias_hi
    const float* W = nullptr;
    const float* W_hi = nullptr;
    const float* bias = nullptr;
    const float* bias_hi = nullptr;
    int split_n = 0;
    // output
    float* Y = nullptr;
    int ldy = 0;
    int M = 0;                      // rows the grid covers
    const int32_t* M_dev = nullptr; // optional device-side row count (<= M)
    int N = 0, K = 0;
    int relu = 0;
    // bf16 feature path (gemm_bf16.hpp): A / W / W_hi point at bf16 data (lda, K in elements), Y at bf16 - or at
    // fp32 when y_f32 is set.  bias / rscale / obs / enc_* are fp32 on both paths.
    int bf16 = 0;
    int y_f32 = 0;
    // exact-fp32 kernels only (gemm_f32_tile's epilogue): the result rows are stored as bf16 (Y: bf16 elements, ldy in elements) -
    // the bf16 feature path's node-feature table, whose encoder rows are evaluated by the fp32 tile that rides with the plan lists
    int y_bf16 = 0;
    // split path (gemm_split.hpp): A / Y fp32 as usual, W / W_hi point at [N][K / 16][3][16] bf16 planes (hi | mid | lo per 16 k)
    int split = 0;
    // MEL_PREC_F32_AUTO: W / W_hi are the fp32 matrices as usual and Ws / Ws_hi their bf16 planes; the launcher takes the split
    // kernel when the launch is large enough for it to win and the exact-fp32 kernel otherwise
    const float* Ws = nullptr;
    const float* Ws_hi = nullptr;
    // split-K (specialised-wavefront kernel only): the K range is cut into `ksplit` equal chunks, chunk s writes its RAW
    // partial products (no scale / bias / ReLU) to Y + s * part_stride; splitk_finish_kernel sums the planes in order
    int ksplit = 0;
    long part_stride = 0;
};

// Up to four independent problems in one launch (e.g. conv.lin_l + conv.lin_r, or the Q and V hidden layers):
// small problems ride in the shadow of the big one instead of paying their own latency-bound launch.
constexpr int GEMM_MAX_GROUP = 4;
struct GemmBatch {
    GemmArgs p[GEMM_MAX_GROUP];
    int start[GEMM_MAX_GROUP + 1];   // first workgroup id of each problem (multiples of 8), total at [count]
    int count;
};

constexpr int GEMM_BK = 32;
constexpr int GEMM_LDS_STRIDE = GEMM_BK + 4;   // floats; 144-byte rows

enum { GEMM_MODE_PLAIN = 0, GEMM_MODE_ENC = 1 };

// Per-thread staging coordinates of one 16-byte chunk (constant across K steps).
struct AChunk {
    const float* src;    // PLAIN: &A[arow(row)][chunk*4]
    float x[8];          // ENC:   node features
};
// Rows beyond M are clamped to row M-1 instead of being predicated: a predicated load turns into a branch
// plus a merge, and the merge makes the compiler wait for the load right where it was issued (no prefetch
// overlap).  The clamped rows only feed accumulator rows that the epilogue never stores.

// The 5 node features of output row `row` for the ENC producer: read from the observation (optionally through the row ->
// node list), or - feat_domain - decoded from the row index itself: the features are small integers (degree, messages
// sent, last action, interested, has message; graph.py:261-269), so the encoder and conv1 projections can be evaluated
// once per distinct TUPLE instead of once per node row (node-feature table, plan_masks.hpp).
__device__ __forceinline__ void enc_features(const GemmArgs& g, int row, float x[8]) {
#pragma unroll
    for (int f = 0; f < 8; ++f) x[f] = 0.f;
    if (g.feat_domain) {
        const int deg = row / 40, rem = row - deg * 40;               // tuple id = degree * 40 + messages * 8 + flags
        x[0] = (float)deg, x[1] = (float)(rem >> 3), x[2] = (float)((rem >> 2) & 1);
        x[3] = (float)((rem >> 1) & 1), x[4] = (float)(rem & 1);
        return;
    }
    const int id = g.nid ? g.nid[row] : row;
    const int b = id / g.n_nodes, node = id - b * g.n_nodes;
    const float* src = g.obs + (size_t)b * g.obs_width + node * g.node_cols + 2;
#pragma unroll
    for (int f = 0; f < 8; ++f)
        if (f < g.in_dim) x[f] = src[f];
}

// enc: layer-0 weights of the encoder staged in LDS as [K][9] = {w[k][0..7], b[k]} (ENC mode only)
template <int MODE>
__device__ __forceinline__ f32x4 fetch_a(const GemmArgs& g, const AChunk& c, int k0, int kc, const float* enc) {
    if constexpr (MODE == GEMM_MODE_PLAIN) {
        return *reinterpret_cast<const f32x4*>(c.src + k0);
    } else {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* wrow = enc + (k0 + kc + e) * 9;
            float s = wrow[8];
#pragma unroll
            for (int f = 0; f < 8; ++f) s = fmaf(wrow[f], c.x[f], s);       // w is zero-padded beyond in_dim
            v[e] = fmaxf(s, 0.f);
        }
        const f32x4 out = {v[0], v[1], v[2], v[3]};
        return out;
    }
}

// Epilogue of one 32x32 accumulator block (C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) +
// 8*(reg >> 2) + 4*(lane >> 5)):   Y[m, n] = act(rscale[m] * acc + bias[n]),  m = m_lane + (e & 3) + 8*(e >> 2).
// Written so that the 16 row-scale loads go out together, everything is computed, and then the 16 stores go out
// back to back.  The obvious per-element loop compiles to: reload Y / ldy / relu from the kernel-argument segment
// (s_load + lgkmcnt wait), load rscale[m], s_waitcnt vmcnt(0) - which also waits for the PREVIOUS element's store -
// and only then the next store: sixteen serialised memory round trips per block (measured: a third of the kernel).
// The operands the epilogue reads from memory (bias of this lane's column, row scales of its 16 rows).  (Fetching them
// a couple of K steps before the tile ends was measured and is NOT worth it: the live range costs the one-role kernel
// a wavefront of occupancy - conv2 103 vs 84 us - and changes nothing where registers are free.)
struct EpilogueOperands {
    float bias;
    float sc[16];
};
__device__ __forceinline__ EpilogueOperands fetch_epilogue_operands(const GemmArgs& g, int m_lane, int n, int M) {
    EpilogueOperands o;
    const float* __restrict__ rs = g.rscale;
    o.bias = (g.bias_hi && n >= g.split_n) ? g.bias_hi[n - g.split_n] : (g.bias ? g.bias[n] : 0.f);
#pragma unroll
    for (int e = 0; e < 16; ++e) o.sc[e] = rs ? rs[min(m_lane + (e & 3) + 8 * (e >> 2), M - 1)] : 1.f;
    return o;
}

__device__ __forceinline__ void store_block_f32(const GemmArgs& g, const f32x16& acc, int m_lane, int n, int M,
                                                const EpilogueOperands& o) {
    float* __restrict__ Y = g.Y;
    const int ldy = g.ldy, relu = g.relu;
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        v[e] = acc[e] * o.sc[e] + o.bias;
        if (relu) v[e] = fmaxf(v[e], 0.f);
    }
    if (g.y_bf16) {                           // round to nearest even, as v_cvt_pk_bf16_f32 does
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        uint16_t* col16 = reinterpret_cast<uint16_t*>(Y) + n;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m_lane + (e & 3) + 8 * (e >> 2);
            const uint32_t w = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{v[e], 0.f}, bf16x2_t));
            if (m < M) col16[(size_t)m * ldy] = (uint16_t)w;
        }
        return;
    }
    float* col = Y + n;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = m_lane + (e & 3) + 8 * (e >> 2);
        if (m < M) col[(size_t)m * ldy] = v[e];
    }
}
__device__ __forceinline__ void store_block_f32(const GemmArgs& g, const f32x16& acc, int m_lane, int n, int M) {
    store_block_f32(g, acc, m_lane, n, M, fetch_epilogue_operands(g, m_lane, n, M));
}

// body of gemm_f32_kernel for workgroup `block` of the launch's tile list (a device function so that another kernel can run
// GEMM tiles and unrelated work in ONE launch: plan_enc_kernel, fwd.hip)
template <int WM, int WN, int TM, int TN, int MODE>
__device__ __forceinline__ void gemm_f32_tile(const GemmBatch& batch, const int block) {
    int pi = 0;
#pragma unroll
    for (int k = 1; k < GEMM_MAX_GROUP; ++k)
        if (k < batch.count && block >= batch.start[k]) pi = k;
    const GemmArgs& g = batch.p[pi];
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, T = 64 * WM * WN;
    constexpr int A_CHUNKS = BM * (GEMM_BK / 4) / T;   // float4 chunks per thread per K step
    constexpr int W_CHUNKS = BN * (GEMM_BK / 4) / T;
    constexpr int BUF = (BM + BN) * GEMM_LDS_STRIDE;   // floats per LDS stage
    constexpr int ENC_MAX_K = 256;                     // encoder hidden width the ENC producer supports
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF + (MODE == GEMM_MODE_ENC ? ENC_MAX_K * 9 : 0)];
    float* enc = lds + 2 * BUF;

    // Tile order.  The row count may be ragged and device-side: only the first `active` workgroup ids
    // have work (the dispatcher deals consecutive ids round-robin over the 8 XCDs, so they are spread
    // evenly) and the rest exit at once.  Inside the active range ids are remapped (bijectively) so that
    // the workgroups sharing an A row panel (same m tile, different n tile) sit on one XCD's L2.
    const int nbn = g.N / BN;
    const int M = g.M_dev ? min(*g.M_dev, g.M) : g.M;
    const int active = ((M + BM - 1) / BM) * nbn;
    int wg = block - batch.start[pi];
    if (wg >= active) return;
    {
        const int q = active >> 3, r8 = active & 7, xcd = wg & 7, local = wg >> 3;
        wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + local;
    }
    const int m0 = (wg / nbn) * BM;
    const int n0 = (wg % nbn) * BN;

    const int tid = threadIdx.x;
    if constexpr (MODE == GEMM_MODE_ENC) {
        for (int i = tid; i < g.K * 9; i += T) {
            const int k = i / 9, f = i - k * 9;
            enc[i] = f == 8 ? g.enc_b[k] : (f < g.in_dim ? g.enc_w[(size_t)k * g.in_dim + f] : 0.f);
        }
        __syncthreads();
    }
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int r = lane & 31, h = lane >> 5;
    const int crow = tid >> 3;            // staging: 8 threads per 128-byte row slice
    const int kc = (tid & 7) * 4;

    AChunk ac[A_CHUNKS];
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) {
        const int row = min(m0 + crow + i * (T / 8), M - 1);
        ac[i].src = nullptr;
#pragma unroll
        for (int f = 0; f < 8; ++f) ac[i].x[f] = 0.f;
        if constexpr (MODE == GEMM_MODE_PLAIN) {
            const int ar = g.arow ? g.arow[row] : row;
            ac[i].src = g.A + (size_t)ar * g.lda + kc;
        } else {
            enc_features(g, row, ac[i].x);
        }
    }
    const float* w_src[W_CHUNKS];
#pragma unroll
    for (int i = 0; i < W_CHUNKS; ++i) {
        const int n = n0 + crow + i * (T / 8);
        const float* base = (g.W_hi && n >= g.split_n) ? g.W_hi + (size_t)(n - g.split_n) * g.K
                                                        : g.W + (size_t)n * g.K;
        w_src[i] = base + kc;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int st_off = crow * GEMM_LDS_STRIDE + kc;                       // this thread's staging slot
    const int a_off = (wm * 32 * TM + r) * GEMM_LDS_STRIDE + 4 * h;       // this lane's fragment rows
    const int w_off = BM * GEMM_LDS_STRIDE + (wn * 32 * TN + r) * GEMM_LDS_STRIDE + 4 * h;

    f32x4 a_reg[A_CHUNKS], w_reg[W_CHUNKS];
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) a_reg[i] = fetch_a<MODE>(g, ac[i], 0, kc, enc);
#pragma unroll
    for (int i = 0; i < W_CHUNKS; ++i) w_reg[i] = *reinterpret_cast<const f32x4*>(w_src[i]);
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i)
        *reinterpret_cast<f32x4*>(lds + st_off + i * (T / 8) * GEMM_LDS_STRIDE) = a_reg[i];
#pragma unroll
    for (int i = 0; i < W_CHUNKS; ++i)
        *reinterpret_cast<f32x4*>(lds + BM * GEMM_LDS_STRIDE + st_off + i * (T / 8) * GEMM_LDS_STRIDE) = w_reg[i];
    __syncthreads();

    const int KT = g.K / GEMM_BK;
    for (int kt = 0; kt < KT; ++kt) {
        const float* cur = lds + (kt & 1) * BUF;
        float* nxt = lds + ((kt + 1) & 1) * BUF;
        const bool more = kt + 1 < KT;
        if (more) {                       // next K step: global loads stay in flight under the MFMAs
            const int k0 = (kt + 1) * GEMM_BK;
#pragma unroll
            for (int i = 0; i < A_CHUNKS; ++i) a_reg[i] = fetch_a<MODE>(g, ac[i], k0, kc, enc);
#pragma unroll
            for (int i = 0; i < W_CHUNKS; ++i) w_reg[i] = *reinterpret_cast<const f32x4*>(w_src[i] + k0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t)
                a[t] = *reinterpret_cast<const f32x4*>(cur + a_off + t * 32 * GEMM_LDS_STRIDE + q * 8);
#pragma unroll
            for (int t = 0; t < TN; ++t)
                b[t] = *reinterpret_cast<const f32x4*>(cur + w_off + t * 32 * GEMM_LDS_STRIDE + q * 8);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                            a[i][kk], b[j][kk], acc[i][j], 0, 0, 0);
        }
        if (more) {                       // the other LDS stage was last read one barrier ago
#pragma unroll
            for (int i = 0; i < A_CHUNKS; ++i)
                *reinterpret_cast<f32x4*>(nxt + st_off + i * (T / 8) * GEMM_LDS_STRIDE) = a_reg[i];
#pragma unroll
            for (int i = 0; i < W_CHUNKS; ++i)
                *reinterpret_cast<f32x4*>(nxt + BM * GEMM_LDS_STRIDE + st_off + i * (T / 8) * GEMM_LDS_STRIDE) = w_reg[i];
        }
        __syncthreads();
    }

    // epilogue
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
            store_block_f32(g, acc[i][j], m0 + wm * 32 * TM + i * 32 + 4 * h, n0 + wn * 32 * TN + j * 32 + r, M);
}

template <int WM, int WN, int TM, int TN, int MODE>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN >= 8) ? 1 : 2) void gemm_f32_kernel(GemmBatch batch) {
    gemm_f32_tile<WM, WN, TM, TN, MODE>(batch, (int)blockIdx.x);
}

// ---- the learn path's backward products WITHOUT transposed copies of their operands -------------------------------------------
// Y[m, n] = sum_k A'[m, k] W'[n, k] as above, with either operand read TRANSPOSED from memory: A_T: A'[m, k] = A[k * lda + m],
// W_T: W'[n, k] = W[k * ldw + n].  dX = dY W is <false, true> (W' = W^T), dW = dY^T X is <true, true> (A' = dY^T, W' = X^T):
// autograd_ops.py used to launch mel_transpose_f32 for each of them (21 launches of a ~150-launch update).  Same 64 x 64 tile,
// LDS image, K order and MFMA sequence as gemm_f32_tile: the sums are bit-identical to the transposed-copy form.  A transposed
// tile is staged by 4-row chunks along the operand's contiguous dimension (coalesced 16-byte loads) and scattered into LDS.
struct GemmTArgs {
    const float* A;
    const float* W;
    float* Y;
    int lda, ldw, ldy, M, N, K;
};

template <bool A_T, bool W_T>
__global__ __launch_bounds__(256, 2) void gemm_f32_t_kernel(GemmTArgs g) {
    constexpr int BM = 64, BN = 64, T = 256;
    constexpr int BUF = (BM + BN) * GEMM_LDS_STRIDE;
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];
    const int nbn = g.N / BN;
    const int m0 = ((int)blockIdx.x / nbn) * BM, n0 = ((int)blockIdx.x % nbn) * BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int r = lane & 31, h = lane >> 5;
    // row-major staging (as gemm_f32_tile): 8 threads per 128-byte row slice, two rows per thread
    const int crow = tid >> 3, kc = (tid & 7) * 4;
    // transposed staging: chunk c of 512 = (k = c >> 4, rows 4 (c & 15) .. + 3), two chunks per thread
    const int tk0 = tid >> 4, tr0 = (tid & 15) * 4;
    f32x4 a_reg[2], w_reg[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (A_T) {
                const int row = min(m0 + tr0, g.M - 4);                  // (M % 4 == 0: whole chunks)
                a_reg[i] = *reinterpret_cast<const f32x4*>(g.A + (size_t)(k0 + tk0 + 16 * i) * g.lda + row);
            } else {
                const int row = min(m0 + crow + 32 * i, g.M - 1);
                a_reg[i] = *reinterpret_cast<const f32x4*>(g.A + (size_t)row * g.lda + k0 + kc);
            }
            if constexpr (W_T) w_reg[i] = *reinterpret_cast<const f32x4*>(g.W + (size_t)(k0 + tk0 + 16 * i) * g.ldw + n0 + tr0);
            else w_reg[i] = *reinterpret_cast<const f32x4*>(g.W + (size_t)(n0 + crow + 32 * i) * g.ldw + k0 + kc);
        }
    };
    auto stage = [&](float* buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (A_T) {
#pragma unroll
                for (int e = 0; e < 4; ++e) buf[(tr0 + e) * GEMM_LDS_STRIDE + tk0 + 16 * i] = a_reg[i][e];
            } else {
                *reinterpret_cast<f32x4*>(buf + (crow + 32 * i) * GEMM_LDS_STRIDE + kc) = a_reg[i];
            }
            float* wb = buf + BM * GEMM_LDS_STRIDE;
            if constexpr (W_T) {
#pragma unroll
                for (int e = 0; e < 4; ++e) wb[(tr0 + e) * GEMM_LDS_STRIDE + tk0 + 16 * i] = w_reg[i][e];
            } else {
                *reinterpret_cast<f32x4*>(wb + (crow + 32 * i) * GEMM_LDS_STRIDE + kc) = w_reg[i];
            }
        }
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int a_off = (wm * 32 + r) * GEMM_LDS_STRIDE + 4 * h;
    const int w_off = BM * GEMM_LDS_STRIDE + (wn * 32 + r) * GEMM_LDS_STRIDE + 4 * h;
    fetch(0);
    stage(lds);
    __syncthreads();
    const int KT = g.K / GEMM_BK;
    for (int kt = 0; kt < KT; ++kt) {
        const float* cur = lds + (kt & 1) * BUF;
        const bool more = kt + 1 < KT;
        if (more) fetch((kt + 1) * GEMM_BK);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(cur + a_off + q * 8);
            const f32x4 b = *reinterpret_cast<const f32x4*>(cur + w_off + q * 8);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[kk], acc, 0, 0, 0);
        }
        if (more) stage(lds + ((kt + 1) & 1) * BUF);
        __syncthreads();
    }
    float* col = g.Y + n0 + wn * 32 + r;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 32 + 4 * h + (e & 3) + 8 * (e >> 2);
        if (m < g.M) col[(size_t)m * g.ldy] = acc[e];
    }
}

// ------------------------------------------------------------------------------------------------
// Persistent variant: a fixed grid of workgroups walks the tile list of all problems of the launch.
// What it buys at the sizes of this path (a few hundred to a few thousand tiles, K as short as 128):
//   * the first K step of the NEXT tile (pointer set-up incl. the row-gather indices, then the global
//     loads) is issued under the MFMAs of the current tile's last K steps, and its LDS stage is filled
//     before the current tile's epilogue - so no tile after the first pays the cold prologue;
//   * the epilogue's stores drain while the next tile is already computing;
//   * no per-tile workgroup launch / exit.
// Tile ids are padded per problem to multiples of 8 so that id % 8 (= XCD under round-robin dispatch, the
// grid is a multiple of 8) keeps meaning "same XCD" for the A-panel-sharing remap.
// ------------------------------------------------------------------------------------------------
#ifdef MEL_GEMM_PROF
// Tuning builds (-DMEL_GEMM_PROF=<TAG>): cycles wave 0 of every workgroup of the launches tagged TAG spends in
// [0] issuing the prefetch (+ next-tile setup), [1] LDS fragment reads + MFMA chain, [2] waiting for the prefetch and
// filling the other LDS stage, [3] the step barrier, [4] the epilogue, [5] whole kernel, [6] workgroups counted
__device__ unsigned long long g_gemm_prof[8];
#define GEMM_T() __builtin_readcyclecounter()
#endif

template <int A_CHUNKS, int W_CHUNKS>
struct TileCtx {
    AChunk ac[A_CHUNKS];
    const float* w_src[W_CHUNKS];
    int m0, n0, M, pi, KT;
};

// TAG only names the call site (1 = conv1, 2 = conv2, 3 = head hidden layers, 0 = anything else) so that profiler
// summaries list the launches of one step separately instead of averaging them under one kernel name.
template <int WM, int WN, int TM, int TN, int MODE, int TAG = 0>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_f32_persistent_kernel(GemmBatch batch) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, T = 64 * WM * WN;
    constexpr int A_CHUNKS = BM * (GEMM_BK / 4) / T;
    constexpr int W_CHUNKS = BN * (GEMM_BK / 4) / T;
    constexpr int BUF = (BM + BN) * GEMM_LDS_STRIDE;
    constexpr int ENC_MAX_K = 256;
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF + (MODE == GEMM_MODE_ENC ? ENC_MAX_K * 9 : 0)];
    float* enc = lds + 2 * BUF;

    // tile bookkeeping (wave-uniform): active tiles and padded prefix per problem
    int act[GEMM_MAX_GROUP], pre[GEMM_MAX_GROUP + 1], rows[GEMM_MAX_GROUP];
    pre[0] = 0;
#pragma unroll
    for (int i = 0; i < GEMM_MAX_GROUP; ++i) {
        act[i] = 0, rows[i] = 0;
        if (i < batch.count) {
            const GemmArgs& q = batch.p[i];
            rows[i] = q.M_dev ? min(*q.M_dev, q.M) : q.M;
            act[i] = ((rows[i] + BM - 1) / BM) * (q.N / BN);
        }
        pre[i + 1] = pre[i] + ((act[i] + 7) & ~7);
    }
    const int total = pre[GEMM_MAX_GROUP];
    const int stride = gridDim.x;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int r = lane & 31, h = lane >> 5;
    const int crow = tid >> 3;
    const int kc = (tid & 7) * 4;

    // first valid tile at or after t (skips the per-problem padding)
    auto next_valid = [&](int t) {
        for (; t < total; t += stride) {
            int pi = 0;
#pragma unroll
            for (int k = 1; k < GEMM_MAX_GROUP; ++k)
                if (t >= pre[k]) pi = k;
            if (t - pre[pi] < act[pi]) return t;
        }
        return total;
    };
    auto setup = [&](TileCtx<A_CHUNKS, W_CHUNKS>& c, int t) {
        int pi = 0;
#pragma unroll
        for (int k = 1; k < GEMM_MAX_GROUP; ++k)
            if (t >= pre[k]) pi = k;
        const GemmArgs& g = batch.p[pi];
        const int nbn = g.N / BN;
        int wg = t - pre[pi];
        {
            const int active = act[pi];
            const int q = active >> 3, r8 = active & 7, xcd = wg & 7, local = wg >> 3;
            wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + local;
        }
        c.pi = pi, c.M = rows[pi], c.KT = g.K / GEMM_BK;
        c.m0 = (wg / nbn) * BM, c.n0 = (wg % nbn) * BN;
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) {
            const int row = min(c.m0 + crow + i * (T / 8), c.M - 1);
            c.ac[i].src = nullptr;
#pragma unroll
            for (int f = 0; f < 8; ++f) c.ac[i].x[f] = 0.f;
            if constexpr (MODE == GEMM_MODE_PLAIN) {
                const int ar = g.arow ? g.arow[row] : row;
                c.ac[i].src = g.A + (size_t)ar * g.lda + kc;
            } else {
                enc_features(g, row, c.ac[i].x);
            }
        }
#pragma unroll
        for (int i = 0; i < W_CHUNKS; ++i) {
            const int n = c.n0 + crow + i * (T / 8);
            const float* base = (g.W_hi && n >= g.split_n) ? g.W_hi + (size_t)(n - g.split_n) * g.K
                                                            : g.W + (size_t)n * g.K;
            c.w_src[i] = base + kc;
        }
    };

    int t = next_valid(blockIdx.x);
    if (t >= total) return;
    if constexpr (MODE == GEMM_MODE_ENC) {
        const GemmArgs& g = batch.p[0];
        for (int i = tid; i < g.K * 9; i += T) {
            const int k = i / 9, f = i - k * 9;
            enc[i] = f == 8 ? g.enc_b[k] : (f < g.in_dim ? g.enc_w[(size_t)k * g.in_dim + f] : 0.f);
        }
        __syncthreads();
    }

    const int st_off = crow * GEMM_LDS_STRIDE + kc;
    const int a_off = (wm * 32 * TM + r) * GEMM_LDS_STRIDE + 4 * h;
    const int w_off = BM * GEMM_LDS_STRIDE + (wn * 32 * TN + r) * GEMM_LDS_STRIDE + 4 * h;

    TileCtx<A_CHUNKS, W_CHUNKS> cur, nxt;
    setup(cur, t);
    nxt = cur;
    f32x4 a_reg[A_CHUNKS], w_reg[W_CHUNKS];
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) a_reg[i] = fetch_a<MODE>(batch.p[0], cur.ac[i], 0, kc, enc);
#pragma unroll
    for (int i = 0; i < W_CHUNKS; ++i) w_reg[i] = *reinterpret_cast<const f32x4*>(cur.w_src[i]);
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i)
        *reinterpret_cast<f32x4*>(lds + st_off + i * (T / 8) * GEMM_LDS_STRIDE) = a_reg[i];
#pragma unroll
    for (int i = 0; i < W_CHUNKS; ++i)
        *reinterpret_cast<f32x4*>(lds + BM * GEMM_LDS_STRIDE + st_off + i * (T / 8) * GEMM_LDS_STRIDE) = w_reg[i];
    __syncthreads();
    int stage = 0;
#ifdef MEL_GEMM_PROF
    unsigned long long pf = 0, pm = 0, pw = 0, pb = 0, pe = 0;
    const unsigned long long pk0 = GEMM_T();
#endif

    for (;;) {
        const int tn = next_valid(t + stride);
        const bool has_next = tn < total;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const int KT = cur.KT;
        for (int kt = 0; kt < KT; ++kt) {
            const float* cst = lds + stage * BUF;
            float* nst = lds + (stage ^ 1) * BUF;
            const bool last = kt + 1 == KT;
#ifdef MEL_GEMM_PROF
            const unsigned long long q0 = GEMM_T();
#endif
            // one K step ahead of its first load, resolve the next tile's pointers (row-gather indices)
            if (has_next && (kt + 2 == KT || (KT == 1 && last))) setup(nxt, tn);
            const bool fill = !last || has_next;
            if (!last) {
                const int k0 = (kt + 1) * GEMM_BK;
#pragma unroll
                for (int i = 0; i < A_CHUNKS; ++i) a_reg[i] = fetch_a<MODE>(batch.p[0], cur.ac[i], k0, kc, enc);
#pragma unroll
                for (int i = 0; i < W_CHUNKS; ++i) w_reg[i] = *reinterpret_cast<const f32x4*>(cur.w_src[i] + k0);
            } else if (has_next) {
#pragma unroll
                for (int i = 0; i < A_CHUNKS; ++i) a_reg[i] = fetch_a<MODE>(batch.p[0], nxt.ac[i], 0, kc, enc);
#pragma unroll
                for (int i = 0; i < W_CHUNKS; ++i) w_reg[i] = *reinterpret_cast<const f32x4*>(nxt.w_src[i]);
            }
#ifdef MEL_GEMM_PROF
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long q1 = GEMM_T();
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 a[TM], b[TN];
#pragma unroll
                for (int u = 0; u < TM; ++u)
                    a[u] = *reinterpret_cast<const f32x4*>(cst + a_off + u * 32 * GEMM_LDS_STRIDE + q * 8);
#pragma unroll
                for (int u = 0; u < TN; ++u)
                    b[u] = *reinterpret_cast<const f32x4*>(cst + w_off + u * 32 * GEMM_LDS_STRIDE + q * 8);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk], b[j][kk], acc[i][j], 0, 0, 0);
            }
#ifdef MEL_GEMM_PROF
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 0" ::"v"(acc[0][0][0]));      // the MFMA chain has retired
            const unsigned long long q2 = GEMM_T();
            __builtin_amdgcn_sched_barrier(0);
#endif
            if (fill) {
#pragma unroll
                for (int i = 0; i < A_CHUNKS; ++i)
                    *reinterpret_cast<f32x4*>(nst + st_off + i * (T / 8) * GEMM_LDS_STRIDE) = a_reg[i];
#pragma unroll
                for (int i = 0; i < W_CHUNKS; ++i)
                    *reinterpret_cast<f32x4*>(nst + BM * GEMM_LDS_STRIDE + st_off + i * (T / 8) * GEMM_LDS_STRIDE) = w_reg[i];
            }
#ifdef MEL_GEMM_PROF
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long q3 = GEMM_T();
#endif
            __syncthreads();
#ifdef MEL_GEMM_PROF
            const unsigned long long q4 = GEMM_T();
            pf += q1 - q0, pm += q2 - q1, pw += q3 - q2, pb += q4 - q3;
#endif
            stage ^= 1;
        }
#ifdef MEL_GEMM_PROF
        const unsigned long long e0 = GEMM_T();
#endif
        {   // epilogue of the finished tile (the next tile's first K step already sits in LDS)
            const GemmArgs& g = batch.p[cur.pi];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    store_block_f32(g, acc[i][j], cur.m0 + wm * 32 * TM + i * 32 + 4 * h, cur.n0 + wn * 32 * TN + j * 32 + r,
                                    cur.M);
        }
#ifdef MEL_GEMM_PROF
        pe += GEMM_T() - e0;
#endif
        if (!has_next) break;
        cur = nxt;
        t = tn;
    }
#ifdef MEL_GEMM_PROF
    if (tid == 0 && TAG == MEL_GEMM_PROF && MODE == GEMM_MODE_PLAIN) {
        atomicAdd(&g_gemm_prof[0], pf), atomicAdd(&g_gemm_prof[1], pm), atomicAdd(&g_gemm_prof[2], pw);
        atomicAdd(&g_gemm_prof[3], pb), atomicAdd(&g_gemm_prof[4], pe), atomicAdd(&g_gemm_prof[5], GEMM_T() - pk0);
        atomicAdd(&g_gemm_prof[6], 1ull);
    }
#endif
}

// Host-side dispatch: picks the tile so that the launch keeps the 256 CUs busy.  `m_hint` is the row
// count the caller expects (ragged lists are sized on the device; the grid still covers g.M rows).
mel_status launch_gemm(const GemmArgs& g, int mode, hipStream_t stream, const char* what, long m_hint = -1,
                       int force_tile = 0, int tag = 0);
// Several PLAIN problems in one launch; hints[i] = expected rows of problem i (-1 = g.M).
mel_status launch_gemm_group(const GemmArgs* gs, const long* hints, int count, hipStream_t stream, const char* what,
                             int tag = 0);

}  // namespace mel
