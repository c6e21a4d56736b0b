// fp32 row GEMM evaluated on the bf16 matrix cores by OPERAND SPLITTING (MEL_PREC_F32_SPLIT):
//
//     x = x_hi + x_mid + x_lo        three bf16 pieces, 8 significant bits each: the fp32 value EXACTLY
//     a * w ~= ah*wh + (ah*wm + am*wh) + (ah*wl + al*wh + am*wm)          six of the nine partial products
//
// Every partial product of two bf16 values is exact in fp32 and the MFMA accumulates in fp32, so the only error is the
// three dropped products (<= 2^-24 relative each): the result is as close to the exact dot product as a native fp32
// GEMM is (measured on random 512^3 problems: max error 1.1e-6 against float64, native fp32 matmul 2.8e-6).  The
// exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) retires 64 FLOP/clk/SIMD, v_mfma_f32_32x32x16_bf16 1024: six bf16 MFMAs
// per 16 k replace eight fp32 MFMAs (512 cycles) with 192 cycles of matrix-pipe time.
//
// Data flow: activations stay fp32 in HBM (no other kernel changes): the A tile is fetched exactly like the fp32
// kernel's (same GemmArgs, same row gather / encoder producer) and split in registers on its way into LDS; the
// weights are split once per weight version (mel_prepare_weights) or per forward call into bf16 planes interleaved per
// 16 k - [N][K / 16][3][16]: the 96 bytes a row contributes to a 16-k step are contiguous (split_weights_kernel).
// LDS row = 3 planes x 64 B (K step 32) + 16 B pad = 208 B: the 16-byte fragment reads of 8 consecutive rows land on
// 8 distinct 4-bank groups.  One ds_read_b128 = one MFMA operand; 12 reads feed the 12 MFMAs of a K step.
// Same persistent tile loop, XCD-aware tile order and next-tile prefetch as gemm_f32_persistent_kernel.
#pragma once
#include "gemm_bf16.hpp"
#include "gemm_ring.hpp"      // wait_lds_done
#include <type_traits>

namespace mel {

constexpr int GEMS_ROW = 13;                 // 16-byte chunks per LDS row: 3 planes x 4 chunks + 1 pad

// 4 fp32 -> their hi / mid / lo bf16 pieces, 8 bytes per plane
__device__ __forceinline__ void split4(const f32x4 x, u32x2& hi, u32x2& mid, u32x2& lo) {
    float r[4];
    uint32_t h[2], m[2], l[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        h[p] = pack_bf16x2(x[2 * p], x[2 * p + 1]);
        r[2 * p] = x[2 * p] - bf16_lo(h[p]);
        r[2 * p + 1] = x[2 * p + 1] - bf16_hi(h[p]);
        m[p] = pack_bf16x2(r[2 * p], r[2 * p + 1]);
        r[2 * p] -= bf16_lo(m[p]);
        r[2 * p + 1] -= bf16_hi(m[p]);
        l[p] = pack_bf16x2(r[2 * p], r[2 * p + 1]);
    }
    hi = u32x2{h[0], h[1]}, mid = u32x2{m[0], m[1]}, lo = u32x2{l[0], l[1]};
}

template <int A_CHUNKS, int W_CHUNKS>
struct STileCtx {
    AChunk ac[A_CHUNKS];
    const u32x4* w_src[W_CHUNKS];      // this thread's 16-byte chunk of plane p = i of its W row, K step 0
    int m0, n0, M, pi, KT;
};

template <int MODE, int TAG = 0>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(GemmBatch batch) {
    constexpr int BM = 64, BN = 64, T = 256;
    constexpr int A_CHUNKS = 2;                       // 4-float chunks per thread per K step (as the fp32 kernel)
    constexpr int W_CHUNKS = 3;                       // one 16-byte chunk of each plane
    constexpr int BUF = (BM + BN) * GEMS_ROW;         // 16-byte chunks per LDS stage
    constexpr int ENC_MAX_K = 256;
    __shared__ u32x4 lds[2 * BUF];
    __shared__ float enc_s[MODE == GEMM_MODE_ENC ? ENC_MAX_K * 9 : 1];
    const float* enc = enc_s;

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
    const int wm = wid >> 1, wn = wid & 1;
    const int r = lane & 31, h = lane >> 5;
    const int crow = tid >> 3;            // A staging: 8 threads per 128-byte fp32 row slice, 32 rows per pass
    const int kc = (tid & 7) * 4;         // first of this thread's 4 consecutive k
    const int wrow = tid >> 2;            // W staging: 4 threads per 64-byte plane slice, all 64 rows in one pass
    const int wch = tid & 3;

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
    auto setup = [&](STileCtx<A_CHUNKS, W_CHUNKS>& c, int t) {
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
            const int row = min(c.m0 + crow + i * 32, c.M - 1);
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
        {   // weights: [N][K / 16][3][16] bf16 planes (split_weights_kernel); W / W_hi point at row 0.  This thread's chunk
            // wch of plane p (k = 8 wch .. + 7 of the 32-k step) sits in 16-k step wch >> 1, half wch & 1
            const int n = c.n0 + wrow;
            const uint16_t* base = (g.W_hi && n >= g.split_n)
                                       ? reinterpret_cast<const uint16_t*>(g.W_hi) + (size_t)(n - g.split_n) * 3 * g.K
                                       : reinterpret_cast<const uint16_t*>(g.W) + (size_t)n * 3 * g.K;
#pragma unroll
            for (int p = 0; p < 3; ++p)
                c.w_src[p] = reinterpret_cast<const u32x4*>(base + ((wch >> 1) * 3 + p) * 16 + (wch & 1) * 8);
        }
    };

    int t = next_valid(blockIdx.x);
    if (t >= total) return;
    int nsteps = 0;                           // K steps of this workgroup's whole stream
    for (int tt = t; tt < total; tt = next_valid(tt + stride)) {
        int pi = 0;
#pragma unroll
        for (int k = 1; k < GEMM_MAX_GROUP; ++k)
            if (tt >= pre[k]) pi = k;
        nsteps += batch.p[pi].K / GEMM_BK;
    }
    if constexpr (MODE == GEMM_MODE_ENC) {
        const GemmArgs& g = batch.p[0];
        float* e = enc_s;
        for (int i = tid; i < g.K * 9; i += T) {
            const int k = i / 9, f = i - k * 9;
            e[i] = f == 8 ? g.enc_b[k] : (f < g.in_dim ? g.enc_w[(size_t)k * g.in_dim + f] : 0.f);
        }
        __syncthreads();
    }

    // LDS addressing in 8-byte units for the A pieces (row * 26 + plane * 8 + kc / 4), 16-byte chunks elsewhere
    u32x2* lds8 = reinterpret_cast<u32x2*>(lds);
    const int a_st = crow * (2 * GEMS_ROW) + (kc >> 2);                   // + i * 32 rows, + plane * 8
    const int w_st = (BM + wrow) * GEMS_ROW + wch;                        // + plane * 4
    const int a_off = (wm * 32 + r) * GEMS_ROW + h;                       // + plane * 4 + 2 * q
    const int w_off = (BM + wn * 32 + r) * GEMS_ROW + h;

    // ---- flat stream of K steps over this workgroup's tiles, register prefetch TWO steps ahead -------------------
    // At this MFMA rate one K step is ~400 cycles of matrix-pipe time per wave, far less than a global-load round
    // trip, so the loads of step s+3 are issued at the end of step s and only consumed (split + LDS fill) at the end
    // of step s+2.  Two register sets alternate; requires K >= 128 (the prefetch then never runs more than one tile
    // ahead of the arithmetic, so one pending tile header is enough).
    struct Regs {
        f32x4 a[A_CHUNKS];
        u32x4 w[W_CHUNKS];
    };
    struct Meta {
        int m0, n0, M, pi, KT;
    };
    STileCtx<A_CHUNKS, W_CHUNKS> pf;          // where the prefetch stands
    int pf_t = t, pf_kt = 0;
    bool pf_valid = true;
    setup(pf, t);
    Meta cm{pf.m0, pf.n0, pf.M, pf.pi, pf.KT}, nm{};
    bool nm_valid = false;

    // loads of the next step of the stream.  UNCONDITIONAL: once the stream is over the last step is simply fetched again
    // (and filled into a stage nobody reads).  A load that one path through the loop skips makes the compiler's wait-count
    // bookkeeping assume the worst at the join - "the other register set's loads may not have been issued, so mine are the
    // youngest" - and every fill then waits for ALL outstanding loads, the ones issued a step ago included: the prefetch
    // distance collapses to zero (measured on the 128 x 128 kernel: 163 TF with the skip).
    auto issue = [&](Regs& R) {
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) R.a[i] = fetch_a<MODE>(batch.p[0], pf.ac[i], pf_kt * GEMM_BK, kc, enc);
#pragma unroll
        for (int p = 0; p < 3; ++p) R.w[p] = pf.w_src[p][pf_kt * 12];          // 192 bytes per row and 32-k step
        if (pf_valid && ++pf_kt == pf.KT) {   // cross into this workgroup's next tile
            const int tn = next_valid(pf_t + stride);
            if (tn < total) {
                setup(pf, tn);
                pf_t = tn, pf_kt = 0;
                nm = Meta{pf.m0, pf.n0, pf.M, pf.pi, pf.KT}, nm_valid = true;
            } else {
                pf_valid = false, pf_kt = pf.KT - 1;
            }
        }
    };
    auto fill_stage = [&](int stage, const Regs& R) {
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) {
            u32x2 hi, mid, lo;
            split4(R.a[i], hi, mid, lo);
            u32x2* dst = lds8 + stage * (2 * BUF) + a_st + i * 32 * (2 * GEMS_ROW);
            dst[0] = hi, dst[8] = mid, dst[16] = lo;
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) lds[stage * BUF + w_st + p * 4] = R.w[p];
    };

    Regs R0, R1;
    issue(R0);                                 // step 0
    issue(R1);                                 // step 1
    fill_stage(0, R0);
    __syncthreads();
    issue(R0);                                 // step 2
    int stage = 0, ckt = 0;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    // one K step: MFMAs on `stage`, then Ra (step s+1) -> the other stage, barrier, then Ra <- loads of step s+3.
    // returns false when the stream is finished
    auto step = [&](Regs& Ra) {
        const u32x4* cst = lds + stage * BUF;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bf16x8 a[3], b[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[p] = __builtin_bit_cast(bf16x8, cst[a_off + p * 4 + 2 * q]);
                b[p] = __builtin_bit_cast(bf16x8, cst[w_off + p * 4 + 2 * q]);
            }
            // smallest products first: mid*mid, hi*lo, lo*hi, hi*mid, mid*hi, hi*hi
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
        }
        fill_stage(stage ^ 1, Ra);
        __syncthreads();
        stage ^= 1;
        issue(Ra);
        if (++ckt == cm.KT) {                 // the tile is complete: epilogue, the fp32 kernel's
            store_block_f32(batch.p[cm.pi], acc, cm.m0 + wm * 32 + 4 * h, cm.n0 + wn * 32 + r, cm.M);
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            cm = nm, ckt = 0;
            if (!nm_valid) cm.KT = 1 << 30;   // (the padding step of an odd stream ends no tile)
            nm_valid = false;
        }
    };
    // the loop is a plain counted one over PAIRS of steps (an odd stream gets one padding step on stale operands): with
    // an exit between the two steps the compiler's control-flow restructuring hands the wait-count pass edges that never
    // run, and the first step of the pair then waits for every outstanding load
    for (int it = 0; it < (nsteps + 1) >> 1; ++it) {
        step(R1);
        step(R0);
    }
}

// ---- 128 x 128 tiles ------------------------------------------------------------------------------------------------
// The 64 x 64 kernel above moves 20 KB from L2 into LDS per 12 MFMAs of each of its waves (384 matrix-pipe cycles): at
// the pipe's rate that is more than a CU's vector-memory path delivers (64 B / clk), so it runs at ~30 % of the split
// path's peak.  Here a workgroup of 2 x 2 waves owns 128 x 128 outputs and every wave 2 x 2 accumulator blocks: 24 MFMAs
// (768 cycles) per wave and 16-k step for 20 KB per workgroup - a quarter of the operand traffic per MFMA - and the 12
// fragment reads of a step feed 24 MFMAs.  K step 16: an LDS row is 3 planes x 32 B + 16 B pad = 112 B (the 16 lanes of a
// ds_read_b128 group land on 16 distinct 16-byte slots), a stage 28 KB, two stages 56 KB: two workgroups per CU, so a
// SIMD holds two waves of different workgroups and one's barrier / epilogue is the other's matrix time.  Same flat
// stream of K steps across the workgroup's work items with the register prefetch two steps ahead; PLAIN mode only.
// Split-K (GemmArgs::ksplit, as in gemm_bf16.hpp): work item = (tile, K chunk), raw products to plane ks.
#ifdef MEL_SPLIT_PROF
__device__ unsigned long long g_split_prof[16];      // issue-time stamps inside a K step (tools/split_prof.py)
#endif
constexpr int GEMS2_ROW = 7;                 // 16-byte chunks per LDS row: 3 planes x 2 chunks + 1 pad
constexpr int GEMS2_BK = 16;

struct S2TileCtx {
    const float* a_src[2];             // this thread's 4 floats of K step 0 of its two A rows
    const u32x4* w_src[3];             // this thread's three 16-byte chunks of the W tile's K step 0
    int m0, n0, M, pi, KT, ks;
};

template <int TAG = 0>
__global__ __launch_bounds__(256, 2) void gemm_split_big_kernel(GemmBatch batch) {
    constexpr int BM = 128, BN = 128;
    constexpr int BUF = (BM + BN) * GEMS2_ROW;        // 16-byte chunks per LDS stage
    __shared__ u32x4 lds[2 * BUF];

    int act[GEMM_MAX_GROUP], pre[GEMM_MAX_GROUP + 1], rows[GEMM_MAX_GROUP];
    pre[0] = 0;
#pragma unroll
    for (int i = 0; i < GEMM_MAX_GROUP; ++i) {
        act[i] = 0, rows[i] = 0;
        if (i < batch.count) {
            const GemmArgs& q = batch.p[i];
            rows[i] = q.M_dev ? min(*q.M_dev, q.M) : q.M;
            act[i] = ((rows[i] + BM - 1) / BM) * (q.N / BN) * (q.ksplit > 1 ? q.ksplit : 1);
        }
        pre[i + 1] = pre[i] + ((act[i] + 7) & ~7);
    }
    const int total = pre[GEMM_MAX_GROUP];
    const int stride = gridDim.x;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int r = lane & 31, h = lane >> 5;
    const int crow = tid >> 2;            // A staging: 4 threads per 64-byte fp32 row slice, 64 rows per pass, 2 passes
    const int kq = tid & 3;               // this thread's 4 consecutive k of the step

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
    auto setup = [&](S2TileCtx& c, int t) {
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
        const int S = g.ksplit > 1 ? g.ksplit : 1;
        c.pi = pi, c.M = rows[pi], c.KT = g.K / GEMS2_BK / S, c.ks = (wg / nbn) % S;
        c.m0 = (wg / (nbn * S)) * BM, c.n0 = (wg % nbn) * BN;
        const int step0 = c.ks * c.KT;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = min(c.m0 + crow + i * 64, c.M - 1);                  // clamped, never predicated
            const int ar = g.arow ? g.arow[row] : row;
            c.a_src[i] = g.A + (size_t)ar * g.lda + step0 * GEMS2_BK + kq * 4;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {       // [N][K / 16][3][16] planes: 6 chunks per row and step, 768 per tile and step
            const int ch = tid + i * 256, wrow = ch / 6, wch = ch - wrow * 6;
            const int n = c.n0 + wrow;
            const uint16_t* base = (g.W_hi && n >= g.split_n)
                                       ? reinterpret_cast<const uint16_t*>(g.W_hi) + (size_t)(n - g.split_n) * 3 * g.K
                                       : reinterpret_cast<const uint16_t*>(g.W) + (size_t)n * 3 * g.K;
            c.w_src[i] = reinterpret_cast<const u32x4*>(base + (size_t)step0 * 48 + wch * 8);
        }
    };

    int t = next_valid(blockIdx.x);
    if (t >= total) return;
    int nsteps = 0;                           // K steps of this workgroup's whole stream
    for (int tt = t; tt < total; tt = next_valid(tt + stride)) {
        int pi = 0;
#pragma unroll
        for (int k = 1; k < GEMM_MAX_GROUP; ++k)
            if (tt >= pre[k]) pi = k;
        nsteps += batch.p[pi].K / GEMS2_BK / (batch.p[pi].ksplit > 1 ? batch.p[pi].ksplit : 1);
    }

    // LDS addressing: 8-byte units for the A pieces (row * 14 + plane * 4 + kq), 16-byte chunks elsewhere
    u32x2* lds8 = reinterpret_cast<u32x2*>(lds);
    const int a_st = crow * (2 * GEMS2_ROW) + kq;                          // + i * 64 rows, + plane * 4
    int w_st[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int ch = tid + i * 256, wrow = ch / 6;
        w_st[i] = (BM + wrow) * GEMS2_ROW + (ch - wrow * 6);
    }
    const int a_off = (wm * 64 + r) * GEMS2_ROW + h;                       // + i * 32 rows, + plane * 2
    const int w_off = (BM + wn * 64 + r) * GEMS2_ROW + h;

    struct Regs {
        f32x4 a[2];
        u32x4 w[3];
    };
    struct Meta {
        int m0, n0, M, pi, KT, ks;
    };
    S2TileCtx pf;                             // where the prefetch stands
    int pf_t = t, pf_kt = 0;
    bool pf_valid = true;
    setup(pf, t);
    Meta cm{pf.m0, pf.n0, pf.M, pf.pi, pf.KT, pf.ks}, nm{};
    bool nm_valid = false;

    // loads of the next step of the stream, unconditional (see gemm_split_kernel); in pieces, so that a step can deal them
    // out between its MFMAs
    auto issue_a = [&](Regs& R) {
        const int kk = pf_kt;
#pragma unroll
        for (int i = 0; i < 2; ++i) R.a[i] = *reinterpret_cast<const f32x4*>(pf.a_src[i] + kk * GEMS2_BK);
    };
    auto issue_w = [&](Regs& R) {
        const int kk = pf_kt;
#pragma unroll
        for (int i = 0; i < 3; ++i) R.w[i] = pf.w_src[i][kk * 6];
    };
    auto advance = [&]() {
        if (pf_valid && ++pf_kt == pf.KT) {   // cross into this workgroup's next work item
            const int tn = next_valid(pf_t + stride);
            if (tn < total) {
                setup(pf, tn);
                pf_t = tn, pf_kt = 0;
                nm = Meta{pf.m0, pf.n0, pf.M, pf.pi, pf.KT, pf.ks}, nm_valid = true;
            } else {
                pf_valid = false, pf_kt = pf.KT - 1;
            }
        }
    };
    auto issue = [&](Regs& R) { issue_a(R), issue_w(R), advance(); };
    auto fill_a = [&](int stage, const Regs& R, int i) {
        u32x2 hi, mid, lo;
        split4(R.a[i], hi, mid, lo);
        u32x2* dst = lds8 + stage * (2 * BUF) + a_st + i * 64 * (2 * GEMS2_ROW);
        dst[0] = hi, dst[4] = mid, dst[8] = lo;
    };
    auto fill_w = [&](int stage, const Regs& R) {
#pragma unroll
        for (int i = 0; i < 3; ++i) lds[stage * BUF + w_st[i]] = R.w[i];
    };
    auto fill_stage = [&](int stage, const Regs& R) { fill_a(stage, R, 0), fill_a(stage, R, 1), fill_w(stage, R); };

    Regs R0, R1;
    issue(R0);                                 // step 0
    issue(R1);                                 // step 1
    fill_stage(0, R0);
    __syncthreads();
    issue(R0);                                 // step 2
    int stage = 0, ckt = 0;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // one K step: MFMAs on `stage`, then Ra (step s+1) -> the other stage, barrier, then Ra <- loads of step s+3
#ifdef MEL_SPLIT_PROF
    unsigned long long pm = 0, pw = 0, pb = 0, pi_ = 0, pe = 0, fine[13] = {0};
    const unsigned long long pk0 = GEMM_T();
#endif
    auto step = [&](Regs& Ra) {
#ifdef MEL_SPLIT_PROF
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long q0 = GEMM_T();
        __builtin_amdgcn_sched_barrier(0);
#endif
        const u32x4* cst = lds + stage * BUF;
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[u][p] = __builtin_bit_cast(bf16x8, cst[a_off + u * 32 * GEMS2_ROW + 2 * p]);
                b[u][p] = __builtin_bit_cast(bf16x8, cst[w_off + u * 32 * GEMS2_ROW + 2 * p]);
            }
        // per block smallest products first (mid*mid, hi*lo, lo*hi, hi*mid, mid*hi, hi*hi), the four blocks interleaved
        // The rest of the step's work is dealt out between the six groups of four MFMAs, so that a wave's instruction stream
        // never has a phase without matrix work (cycle stamps of the phase-by-phase version, per K step and wave: fragment
        // reads + MFMAs 1 109, prefetch wait + split + fill 713, barrier 188, issuing the five loads behind the barrier 603 -
        // the four waves of a workgroup all queue at the CU's address path at once -, 3 076 in all against 768 of MFMAs):
        // Ra (step s+1) is split and written to the other stage behind groups 0-2, and as soon as its registers are free
        // they are reloaded with step s+3 behind groups 3-4.
        constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#ifdef MEL_SPLIT_PROF
        unsigned long long ts[13];
        __builtin_amdgcn_sched_barrier(0);
        ts[0] = GEMM_T();                      // the 12 fragment reads are issued
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int k = 0; k < 6; ++k) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[k]], b[j][PB[k]], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#ifdef MEL_SPLIT_PROF
            ts[1 + 2 * k] = GEMM_T();          // MFMA group k is issued
            __builtin_amdgcn_sched_barrier(0);
#endif
            if (k == 0) fill_a(stage ^ 1, Ra, 0);
            if (k == 1) fill_a(stage ^ 1, Ra, 1);
            if (k == 2) fill_w(stage ^ 1, Ra);
            if (k == 3) issue_a(Ra);
            if (k == 4) issue_w(Ra);
            __builtin_amdgcn_sched_barrier(0);
#ifdef MEL_SPLIT_PROF
            ts[2 + 2 * k] = GEMM_T();          // the piece behind group k is issued
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
#ifdef MEL_SPLIT_PROF
        fine[0] += ts[0] - q0;
#pragma unroll
        for (int k = 1; k < 13; ++k) fine[k] += ts[k] - ts[k - 1];
#endif
#ifdef MEL_SPLIT_PROF
        asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[1][1][0]));      // the MFMA chains have retired
        const unsigned long long q1 = GEMM_T();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long q2 = GEMM_T();
#endif
        __syncthreads();
#ifdef MEL_SPLIT_PROF
        const unsigned long long q3 = GEMM_T();
        __builtin_amdgcn_sched_barrier(0);
#endif
        stage ^= 1;
        advance();
#ifdef MEL_SPLIT_PROF
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long q4 = GEMM_T();
        __builtin_amdgcn_sched_barrier(0);
        pm += q1 - q0, pw += q2 - q1, pb += q3 - q2, pi_ += q4 - q3;
#endif
        if (++ckt == cm.KT) {                  // the work item is complete
            const GemmArgs& g = batch.p[cm.pi];
            if (g.ksplit > 1) {                // raw partial products into this chunk's fp32 plane
                float* P = g.Y + (size_t)cm.ks * g.part_stride;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int n = cm.n0 + wn * 64 + j * 32 + r;
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int m = cm.m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                            if (m < cm.M) P[(size_t)m * g.ldy + n] = acc[i][j][e];
                        }
                    }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        store_block_f32(g, acc[i][j], cm.m0 + wm * 64 + i * 32 + 4 * h, cm.n0 + wn * 64 + j * 32 + r, cm.M);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            cm = nm, ckt = 0;
            if (!nm_valid) cm.KT = 1 << 30;    // (the padding step of an odd stream ends no work item)
            nm_valid = false;
#ifdef MEL_SPLIT_PROF
            pe += GEMM_T() - q4;
#endif
        }
    };
    for (int it = 0; it < (nsteps + 1) >> 1; ++it) {       // counted loop over pairs of steps (see gemm_split_kernel)
        step(R1);
        step(R0);
    }
#ifdef MEL_SPLIT_PROF
    // tuning builds (-DMEL_GEMM_PROF=99 -DMEL_SPLIT_PROF, tools/split_prof.py): cycles of wave 0 of every workgroup in [0] stream
    // bookkeeping, [1] fragment reads + MFMAs with fill / prefetch between them, [2] LDS writes landing, [3] barrier, [4] epilogue, [5] kernel
    if (tid == 0) {
        atomicAdd(&g_gemm_prof[0], pi_), atomicAdd(&g_gemm_prof[1], pm), atomicAdd(&g_gemm_prof[2], pw);
        atomicAdd(&g_gemm_prof[3], pb), atomicAdd(&g_gemm_prof[4], pe), atomicAdd(&g_gemm_prof[5], GEMM_T() - pk0);
        atomicAdd(&g_gemm_prof[6], 1ull), atomicAdd(&g_gemm_prof[7], (unsigned long long)nsteps);
#pragma unroll
        for (int k = 0; k < 13; ++k) atomicAdd(&g_split_prof[k], fine[k]);
    }
#endif
}

// Round 3 built three more forms of this kernel - specialised wavefronts (4 MFMA waves + one or two teams of loader waves, 3- /
// 4-stage LDS rings), a 128 x 256 tile, three workgroups per CU - all bit-identical to it and none faster; they are parked with
// their probes and the reason (the CU's vector-memory path delivers ~17 B / clk while its matrix pipe is saturated: this tile
// needs 26) in tools/experiments/gemm_split_variants.hpp.

// ---- 128 x 256 tiles fed from bf16 PLANES on both sides, specialised wavefronts (round 3) -------------------------------------
// What bounded the kernels above, found one layer at a time (tools/planes_probe.hip, profiles/r03k_*):
//   * a CU's vector-memory path delivers ~17 B / clk of 64- / 96-byte row pieces while its matrix pipe is busy
//     (tools/overlap_probe.hip), and a 128 x 128 tile needs 26: so 128 x 256 outputs per workgroup (eight MFMA waves of 64 x 64,
//     two per SIMD), and W stored the way a K step reads it - [N / 256][K / 16][256][3][16] (mel_prepare_weights), the 24 KB of
//     a step one contiguous run, every load instruction of a wave one whole KB; A as [rows][K / 16][3][16] planes, written by
//     the producer of the rows (the conv1 attention's store; 128-row blocks for A as well measured the same here and cost
//     that kernel 13 us at N = 100 in scattered stores);
//   * A arrives ALREADY SPLIT: no vector arithmetic is left in the GEMM, nothing competes with the MFMA stream for issue slots;
//     four loader waves (one per SIMD) only move bytes - nine 16-byte loads per thread and step, three steps ahead of their LDS
//     write (three register sets);
//   * all eight MFMA waves reading their 96 KB of fragments right behind the step's barrier left the matrix pipes idle for a
//     third of the step: they now refresh their fragments plane by plane WHILE the step's MFMAs run (below) - no second
//     register set, nothing to wait for behind the barrier.  (Replacing the barrier by LDS counters between the two roles,
//     with s_setprio keeping the two MFMA waves of a SIMD level, came out the same: with loads, LDS writes and fragment reads
//     hidden, 48 MFMAs per SIMD and step take 1 800 cycles at the clock the chip holds under this load.)
//   * the MFMA waves write a finished tile out through a private 4 KB LDS transposition as 16-byte row stores, bias from LDS.
// LDS: 3 x 36 KB stages (96-byte rows, chunk c of row r in slot c ^ ((r >> 3) & 1): conflict-free fragment reads without a pad
// chunk) + 8 x 4 KB + 6 KB of biases = 146 KB, one 768-thread workgroup per CU.  No row scale, no split-K.
constexpr int GEMP_BN = 256;
constexpr int GEMP_STAGES = 3;
constexpr int GEMP_RC = 6;                            // 16-byte chunks per LDS row
constexpr int GEMP_BIAS_FLOATS = 1536;

#ifdef MEL_PLANES_STAMPS
__device__ long long planes_stamps[32];
__device__ long long planes_block[512][2];
#endif

template <int TAG = 0>
__global__ __launch_bounds__(768, 3) void gemm_planes_kernel(GemmBatch batch) {
    constexpr int BM = 128, BN = GEMP_BN, RC = GEMP_RC;
    constexpr int BUF = (BM + BN) * RC;               // 16-byte chunks per LDS stage
    constexpr int XP = 8 * 256;                       // transposition buffers: 4 KB per MFMA wave
    __shared__ u32x4 lds[GEMP_STAGES * BUF + XP + GEMP_BIAS_FLOATS / 4];
    float* xpose = reinterpret_cast<float*>(lds + GEMP_STAGES * BUF);
    float* bias_s = xpose + 4 * XP;

    int act[GEMM_MAX_GROUP], pre[GEMM_MAX_GROUP + 1], rows[GEMM_MAX_GROUP], boff[GEMM_MAX_GROUP];
    pre[0] = 0;
    {
        int o = 0;
#pragma unroll
        for (int i = 0; i < GEMM_MAX_GROUP; ++i) {
            act[i] = 0, rows[i] = 0, boff[i] = o;
            if (i < batch.count) {
                const GemmArgs& q = batch.p[i];
                rows[i] = q.M_dev ? min(*q.M_dev, q.M) : q.M;
                act[i] = ((rows[i] + BM - 1) / BM) * (q.N / BN);
                for (int n = threadIdx.x; n < q.N; n += 768)
                    bias_s[o + n] = (q.bias_hi && n >= q.split_n) ? q.bias_hi[n - q.split_n] : (q.bias ? q.bias[n] : 0.f);
                o += q.N;
            }
            pre[i + 1] = pre[i] + ((act[i] + 7) & ~7);
        }
    }
    const int total = pre[GEMM_MAX_GROUP];
    const int stride = gridDim.x;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // 0-7 MFMA waves, 8-11 loader waves

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
    struct Meta {
        int m0, n0, M, pi, KT;
    };
    auto meta_of = [&](int t) {
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
        Meta m;
        m.pi = pi, m.M = rows[pi], m.KT = g.K / GEMS2_BK;
        m.m0 = (wg / nbn) * BM, m.n0 = (wg % nbn) * BN;
        return m;
    };

    const int t0 = next_valid(blockIdx.x);
    __syncthreads();                          // the biases are staged
    if (t0 >= total) return;
#ifdef MEL_PLANES_STAMPS
    const long long k_c0 = clock64(), k_w0 = wall_clock64();
#endif
    int nsteps = 0;                           // K steps of this workgroup's whole stream of work items
    for (int tt = t0; tt < total; tt = next_valid(tt + stride)) nsteps += meta_of(tt).KT;

    if (wid < 8) {
        // ---- MFMA waves ------------------------------------------------------------------------------------------------
        const int wm = wid >> 2, wn = wid & 3;
        const int r = lane & 31, h = lane >> 5;
        const int sw = (r >> 3) & 1;                                           // (64 wm, 64 wn, 32 u are multiples of 16)
        const int a_off = (wm * 64 + r) * RC + (h ^ sw);                       // + u * 32 rows, + plane * 2
        const int w_off = (BM + wn * 64 + r) * RC + (h ^ sw);
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        float* xp = xpose + wid * 1024;
        auto write_out = [&](const Meta& m) {
            const GemmArgs& g = batch.p[m.pi];
            float* __restrict__ Y = g.Y;
            const int relu = g.relu, ldy = g.ldy;
            int bo = 0;
#pragma unroll
            for (int k = 1; k < GEMM_MAX_GROUP; ++k)
                if (m.pi >= k) bo = boff[k];
            const int rr = lane >> 3, c4 = (lane & 7) * 4;          // read side: row rr + 8 q, floats c4 .. c4 + 3
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = m.n0 + wn * 64 + j * 32 + c4;
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_s + bo + n);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        xp[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[i][j][e];
                        acc[i][j][e] = 0.f;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = rr + 8 * q, mrow = m.m0 + wm * 64 + i * 32 + row;
                        const f32x4 a = *reinterpret_cast<const f32x4*>(xp + row * 32 + c4);
                        f32x4 o = {a[0] + b4[0], a[1] + b4[1], a[2] + b4[2], a[3] + b4[3]};
                        if (relu) o = f32x4{fmaxf(o[0], 0.f), fmaxf(o[1], 0.f), fmaxf(o[2], 0.f), fmaxf(o[3], 0.f)};
                        if (mrow < m.M) *reinterpret_cast<f32x4*>(Y + (size_t)mrow * ldy + n) = o;
                    }
                }
            }
        };
        int t = t0, kt = 0;
        Meta cm = meta_of(t0);
        bf16x8 a[2][3], b[2][3];
        // A step's 24 MFMAs, per block smallest products first (mid*mid, hi*lo, lo*hi, hi*mid, mid*hi, hi*hi), the four blocks
        // interleaved.  The fragments of step g + 1 are read WHILE step g's MFMAs run, each plane into the registers of the
        // plane it replaces as soon as that one has had its last use (b.lo after group 1, a.lo after 2, b.mid after 3, a.mid
        // after 4, the hi planes after 5).
        auto mf = [&](auto PAc, auto PBc) {
            constexpr int pa = decltype(PAc)::value, pb = decltype(PBc)::value;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa], b[j][pb], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        using P2 = std::integral_constant<int, 2>;
        __builtin_amdgcn_s_barrier();         // B(-1): stages 0 and 1 hold steps 0 and 1
#ifdef MEL_PLANES_STAMPS
        if (wid == 0 && lane == 0) planes_stamps[26] = clock64() - k_c0;
#endif
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                a[u][pl] = __builtin_bit_cast(bf16x8, lds[a_off + u * 32 * RC + 2 * pl]);
                b[u][pl] = __builtin_bit_cast(bf16x8, lds[w_off + u * 32 * RC + 2 * pl]);
            }
        // Interval g (between B(g - 1) and B(g)): the MFMAs of step g out of registers, the fragments of step g + 1 out of
        // stage (g + 1) % 3 (filled in interval g - 1); the loaders fill stage (g + 2) % 3, last read in interval g - 2.
        for (int g = 0; g < nsteps; ++g) {
            const u32x4* nx = lds + ((g + 1) % GEMP_STAGES) * BUF;
            auto ra = [&](auto Pc) {
                constexpr int pl = decltype(Pc)::value;
                a[0][pl] = __builtin_bit_cast(bf16x8, nx[a_off + 2 * pl]);
                a[1][pl] = __builtin_bit_cast(bf16x8, nx[a_off + 32 * RC + 2 * pl]);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto rb = [&](auto Pc) {
                constexpr int pl = decltype(Pc)::value;
                b[0][pl] = __builtin_bit_cast(bf16x8, nx[w_off + 2 * pl]);
                b[1][pl] = __builtin_bit_cast(bf16x8, nx[w_off + 32 * RC + 2 * pl]);
                __builtin_amdgcn_sched_barrier(0);
            };
            mf(P1{}, P1{});
            mf(P0{}, P2{});
            rb(P2{});
            mf(P2{}, P0{});
            ra(P2{});
            mf(P0{}, P1{});
            rb(P1{});
            mf(P1{}, P0{});
            ra(P1{});
            mf(P0{}, P0{});
            ra(P0{});
            rb(P0{});
            if (++kt == cm.KT) {              // the work item is complete
                write_out(cm);
                t = next_valid(t + stride), kt = 0;
                if (t < total) cm = meta_of(t);
                else cm.KT = 1 << 30;
            }
            __builtin_amdgcn_s_barrier();     // B(g)
        }
#ifdef MEL_PLANES_STAMPS
        if (wid == 0 && lane == 0) {
            planes_stamps[24] = clock64() - k_c0, planes_stamps[25] = wall_clock64() - k_w0;
            planes_block[blockIdx.x][0] = k_w0, planes_block[blockIdx.x][1] = wall_clock64();
        }
#endif
        return;
    }

    // ---- loader waves: nine 16-byte chunks per thread and step (A: 128 rows x 6, W: 256 rows x 6), nothing else ------------------
    const int tid = threadIdx.x - 512;
    // addresses as 32-bit byte offsets from two wave-uniform bases (the A blocks and the W blocks of the work item: scalar
    // registers): nine 64-bit pointers per thread would cost the loaders their third register set
    struct Ctx {
        const char* a_base;
        const char* w_base;
        uint32_t off[9];                   // chunk i of this thread, K step 0 of the work item
        int KT;
    };
    int dst[9];                            // LDS slot of chunk i (16-byte units inside a stage)
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int ch = tid + i * 256, row = ch / 6, c = ch - row * 6;          // rows 0-127: A, 128-383: W
        dst[i] = row * RC + (c ^ ((row >> 3) & 1));
    }
    auto setup = [&](Ctx& c, int t) {
        const Meta m = meta_of(t);
        const GemmArgs& g = batch.p[m.pi];
        c.KT = m.KT;
        c.a_base = reinterpret_cast<const char*>(g.A);
        // a 256-column tile lies inside ONE weight matrix (split_n is a multiple of 256: the launcher checks)
        c.w_base = (g.W_hi && m.n0 >= g.split_n)
                       ? reinterpret_cast<const char*>(g.W_hi) + (size_t)(m.n0 - g.split_n) * 6 * g.K
                       : reinterpret_cast<const char*>(g.W) + (size_t)m.n0 * 6 * g.K;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int ch = tid + i * 256, row = ch / 6, cc = ch - row * 6;
            if (i < 3) {                       // A planes [rows][K / 16][3][16]; rows beyond M clamped, never predicated
                const int gr = min(m.m0 + row, m.M - 1);
                const int ar = g.arow ? g.arow[gr] : gr;
                c.off[i] = (uint32_t)ar * (uint32_t)(g.K * 6) + cc * 16;
            } else {                           // W blocks [N / 256][K / 16][256][3][16]: a step's 24 KB are contiguous
                c.off[i] = (uint32_t)(row - BM) * 96u + cc * 16;
            }
        }
    };
    struct Regs {
        u32x4 v[9];
    };
    Ctx pf;
    int pf_t = t0, pf_kt = 0;
    bool pf_valid = true;
    setup(pf, t0);
    auto issue = [&](Regs& R) {               // loads of the next step of the stream, unconditional (see gemm_split_kernel)
        const uint32_t ka = (uint32_t)pf_kt * 96u;             // a K step: 96 bytes further in every A row,
        const uint32_t kw = (uint32_t)pf_kt * 24576u;          // the next 24 KB of the W block
#pragma unroll
        for (int i = 0; i < 9; ++i)
            R.v[i] = *reinterpret_cast<const u32x4*>((i < 3 ? pf.a_base : pf.w_base) + (pf.off[i] + (i < 3 ? ka : kw)));
        if (pf_valid && ++pf_kt == pf.KT) {
            const int tn = next_valid(pf_t + stride);
            if (tn < total) {
                setup(pf, tn);
                pf_t = tn, pf_kt = 0;
            } else {
                pf_valid = false, pf_kt = pf.KT - 1;
            }
        }
    };
    auto fill = [&](int st, const Regs& R) {
        u32x4* stage = lds + (st % GEMP_STAGES) * BUF;
#pragma unroll
        for (int i = 0; i < 9; ++i) stage[dst[i]] = R.v[i];
    };
    Regs R0, R1, R2;
    issue(R0);                                 // step 0
    issue(R1);                                 // step 1
    issue(R2);                                 // step 2
    fill(0, R0);
    issue(R0);                                 // step 3
    fill(1, R1);
    issue(R1);                                 // step 4
    wait_lds_done();
    __builtin_amdgcn_s_barrier();              // B(-1)
    // interval g: Ra (step g + 2) -> stage (g + 2) % 3, Ra <- step g + 5; every wave of the workgroup runs nsteps + 1 barriers
    auto step = [&](int g, Regs& Ra) {
        fill(g + 2, Ra);
        issue(Ra);
        wait_lds_done();
        __builtin_amdgcn_s_barrier();          // B(g)
    };
    for (int g = 0; g < nsteps; g += 3) {
        step(g, R2);
        if (g + 1 < nsteps) step(g + 1, R0);
        if (g + 2 < nsteps) step(g + 2, R1);
    }
}

// ---- the bf16 feature path's large projections: the same workgroup (128 x 256 tile, 8 MFMA + 4 loader waves) on bf16 rows ---------
// BASELINE's "bf16 feature path" (MEL_PREC_BF16): A [rows, lda] and W [N, K] are bf16 rows as the producers store them (a 64-k
// stage takes one whole 128-byte line of every row), ONE product per term.  The 64 x 64 one-role kernel streams 0.5 GB of
// operands through the L2 for conv2 (16 TB/s: its ceiling); this tile needs 97 MB.  Per 64-k stage: 48 KB of operands for 16
// MFMAs per wave (1 024 matrix cycles per SIMD) - the load path (~20 B / clk), not the pipe, sets the pace.  Three 48 KB LDS
// stages (128-byte rows, chunk c of row r in slot c ^ ((r >> 1) & 7): conflict-free fragment reads), the loaders two stages
// ahead (two register sets of twelve 16-byte chunks per thread), fragments of the next stage refreshed sub-step by sub-step while
// this one's MFMAs run, one barrier per stage; the epilogue stores straight from the accumulators (bf16, or fp32 with y_f32).
constexpr int GEMW_BK = 64;
constexpr int GEMW_RC = 8;                            // 16-byte chunks per LDS row and stage

template <int TAG = 0>
__global__ __launch_bounds__(768, 3) void gemm_bf16_wide_kernel(GemmBatch batch) {
    constexpr int BM = 128, BN = GEMP_BN, RC = GEMW_RC;
    constexpr int BUF = (BM + BN) * RC;               // 16-byte chunks per LDS stage
    __shared__ u32x4 lds[GEMP_STAGES * BUF + GEMP_BIAS_FLOATS / 4];
    float* bias_s = reinterpret_cast<float*>(lds + GEMP_STAGES * BUF);

    int act[GEMM_MAX_GROUP], pre[GEMM_MAX_GROUP + 1], rows[GEMM_MAX_GROUP], boff[GEMM_MAX_GROUP];
    pre[0] = 0;
    {
        int o = 0;
#pragma unroll
        for (int i = 0; i < GEMM_MAX_GROUP; ++i) {
            act[i] = 0, rows[i] = 0, boff[i] = o;
            if (i < batch.count) {
                const GemmArgs& q = batch.p[i];
                rows[i] = q.M_dev ? min(*q.M_dev, q.M) : q.M;
                act[i] = ((rows[i] + BM - 1) / BM) * (q.N / BN);
                for (int n = threadIdx.x; n < q.N; n += 768)
                    bias_s[o + n] = (q.bias_hi && n >= q.split_n) ? q.bias_hi[n - q.split_n] : (q.bias ? q.bias[n] : 0.f);
                o += q.N;
            }
            pre[i + 1] = pre[i] + ((act[i] + 7) & ~7);
        }
    }
    const int total = pre[GEMM_MAX_GROUP];
    const int stride = gridDim.x;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // 0-7 MFMA waves, 8-11 loader waves

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
    struct Meta {
        int m0, n0, M, pi, KT;
    };
    auto meta_of = [&](int t) {
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
        Meta m;
        m.pi = pi, m.M = rows[pi], m.KT = g.K / GEMW_BK;
        m.m0 = (wg / nbn) * BM, m.n0 = (wg % nbn) * BN;
        return m;
    };

    const int t0 = next_valid(blockIdx.x);
    __syncthreads();                          // the biases are staged
    if (t0 >= total) return;
    int nsteps = 0;                           // 64-k stages of this workgroup's whole stream of work items
    for (int tt = t0; tt < total; tt = next_valid(tt + stride)) nsteps += meta_of(tt).KT;

    if (wid < 8) {
        // ---- MFMA waves ------------------------------------------------------------------------------------------------
        const int wm = wid >> 2, wn = wid & 3;
        const int r = lane & 31, h = lane >> 5;
        const int sw = (r >> 1) & 7;
        const int a_row = (wm * 64 + r) * RC, w_row = (BM + wn * 64 + r) * RC;      // + u * 32 rows; + slot of the sub-step's chunk
        int cx[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) cx[ks] = (2 * ks + h) ^ sw;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        auto write_out = [&](const Meta& m) {
            const GemmArgs& g = batch.p[m.pi];
            const int relu = g.relu, ldy = g.ldy, y32 = g.y_f32;
            uint16_t* Y16 = reinterpret_cast<uint16_t*>(g.Y);
            int bo = 0;
#pragma unroll
            for (int k = 1; k < GEMM_MAX_GROUP; ++k)
                if (m.pi >= k) bo = boff[k];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = m.n0 + wn * 64 + j * 32 + r;
                const float bias = bias_s[bo + n];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = m.m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                        float v = acc[i][j][e] + bias;
                        if (relu) v = fmaxf(v, 0.f);
                        acc[i][j][e] = 0.f;
                        if (row < m.M) {
                            if (y32) g.Y[(size_t)row * ldy + n] = v;
                            else Y16[(size_t)row * ldy + n] = (uint16_t)pack_bf16x2(v, 0.f);
                        }
                    }
                }
            }
        };
        int t = t0, kt = 0;
        Meta cm = meta_of(t0);
        bf16x8 a[4][2], b[4][2];
        __builtin_amdgcn_s_barrier();         // B(-1): stages 0 and 1 hold the first two stages of the stream
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                a[ks][u] = __builtin_bit_cast(bf16x8, lds[a_row + u * 32 * RC + cx[ks]]);
                b[ks][u] = __builtin_bit_cast(bf16x8, lds[w_row + u * 32 * RC + cx[ks]]);
            }
        // interval g: the MFMAs of stage g out of registers, the fragments of stage g + 1 out of LDS stage (g + 1) % 3 (filled in
        // interval g - 1); the loaders fill LDS stage (g + 2) % 3, last read in interval g - 2
        for (int g = 0; g < nsteps; ++g) {
            const u32x4* nx = lds + ((g + 1) % GEMP_STAGES) * BUF;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    a[ks][u] = __builtin_bit_cast(bf16x8, nx[a_row + u * 32 * RC + cx[ks]]);
                    b[ks][u] = __builtin_bit_cast(bf16x8, nx[w_row + u * 32 * RC + cx[ks]]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (++kt == cm.KT) {              // the work item is complete
                write_out(cm);
                t = next_valid(t + stride), kt = 0;
                if (t < total) cm = meta_of(t);
                else cm.KT = 1 << 30;
            }
            __builtin_amdgcn_s_barrier();     // B(g)
        }
        return;
    }

    // ---- loader waves: twelve 16-byte chunks per thread and stage (A: 128 rows x 8, W: 256 rows x 8), nothing else -----------------
    const int tid = threadIdx.x - 512;
    struct Ctx {
        const char* a_base;
        const char* w_base;
        uint32_t off[12];                  // chunk i of this thread, stage 0 of the work item
        int KT;
    };
    int dst[12];                           // LDS slot of chunk i (16-byte units inside a stage)
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int ch = tid + i * 256, row = ch >> 3, c = ch & 7;               // rows 0-127: A, 128-383: W
        dst[i] = row * RC + (c ^ ((row >> 1) & 7));
    }
    auto setup = [&](Ctx& c, int t) {
        const Meta m = meta_of(t);
        const GemmArgs& g = batch.p[m.pi];
        c.KT = m.KT;
        c.a_base = reinterpret_cast<const char*>(g.A);
        // a 256-column tile lies inside ONE weight matrix (split_n is a multiple of 256: the launcher checks)
        c.w_base = (g.W_hi && m.n0 >= g.split_n)
                       ? reinterpret_cast<const char*>(g.W_hi) + (size_t)(m.n0 - g.split_n) * 2 * g.K
                       : reinterpret_cast<const char*>(g.W) + (size_t)m.n0 * 2 * g.K;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int ch = tid + i * 256, row = ch >> 3, cc = ch & 7;
            if (i < 4) {                       // A rows (bf16, lda elements apart); rows beyond M clamped, never predicated
                const int gr = min(m.m0 + row, m.M - 1);
                const int ar = g.arow ? g.arow[gr] : gr;
                c.off[i] = (uint32_t)ar * (uint32_t)(g.lda * 2) + cc * 16;
            } else {
                c.off[i] = (uint32_t)(row - BM) * (uint32_t)(g.K * 2) + cc * 16;
            }
        }
    };
    struct Regs {
        u32x4 v[12];
    };
    Ctx pf;
    int pf_t = t0, pf_kt = 0;
    bool pf_valid = true;
    setup(pf, t0);
    auto issue = [&](Regs& R) {               // loads of the next stage of the stream, unconditional (see gemm_split_kernel)
        const uint32_t ks = (uint32_t)pf_kt * 128u;            // a stage: 128 bytes further in every row
#pragma unroll
        for (int i = 0; i < 12; ++i)
            R.v[i] = *reinterpret_cast<const u32x4*>((i < 4 ? pf.a_base : pf.w_base) + (pf.off[i] + ks));
        if (pf_valid && ++pf_kt == pf.KT) {
            const int tn = next_valid(pf_t + stride);
            if (tn < total) {
                setup(pf, tn);
                pf_t = tn, pf_kt = 0;
            } else {
                pf_valid = false, pf_kt = pf.KT - 1;
            }
        }
    };
    auto fill = [&](int st, const Regs& R) {
        u32x4* stage = lds + (st % GEMP_STAGES) * BUF;
#pragma unroll
        for (int i = 0; i < 12; ++i) stage[dst[i]] = R.v[i];
    };
    Regs R0, R1;
    issue(R0);                                 // stage 0
    issue(R1);                                 // stage 1
    fill(0, R0);
    issue(R0);                                 // stage 2
    fill(1, R1);
    issue(R1);                                 // stage 3
    wait_lds_done();
    __builtin_amdgcn_s_barrier();              // B(-1)
    // interval g: Ra (stage g + 2) -> LDS stage (g + 2) % 3, Ra <- stage g + 4; every wave of the workgroup runs nsteps + 1 barriers
    auto step = [&](int g, Regs& Ra) {
        fill(g + 2, Ra);
        issue(Ra);
        wait_lds_done();
        __builtin_amdgcn_s_barrier();          // B(g)
    };
    for (int g = 0; g < nsteps; g += 2) {
        step(g, R0);
        if (g + 1 < nsteps) step(g + 1, R1);
    }
}

// [rows][K / 16][3][16] planes -> blocks of RB rows, [rows / RB][K / 16][RB][3][16]: what a 16-k step of gemm_planes_kernel reads
// from one operand (12 KB of 128 A rows, 24 KB of 256 W rows) is one contiguous run.  Rows beyond `rows` are written as zeros.
__global__ __launch_bounds__(256) void planes_to_blocks_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int rows, int K, int RB) {
    const size_t ch = (size_t)blockIdx.x * 256 + threadIdx.x;              // 16-byte chunk of the destination
    const int KT = K / 16;
    const size_t per_block = (size_t)KT * RB * 6;
    const size_t total = (size_t)((rows + RB - 1) / RB) * per_block;
    if (ch >= total) return;
    const int blk = (int)(ch / per_block);
    const int rem = (int)(ch - (size_t)blk * per_block);
    const int kt = rem / (RB * 6), r = (rem - kt * RB * 6) / 6, c = rem % 6;
    const int row = blk * RB + r;
    dst[ch] = row < rows ? src[((size_t)row * KT + kt) * 6 + c] : u32x4{0, 0, 0, 0};
}

// fp32 [rows, K] weight matrices -> [rows][K / 16][3][16] bf16 planes (hi | mid | lo per 16 k), all matrices of the model in one launch.
// rb[s] > 0: segment s is written in blocks of rb rows instead, [rows / rb][K / 16][rb][3][16] (gemm_planes_kernel's operands:
// rb = 256 for W, 128 for A; rows a multiple of rb or the buffer rounded up to it - rows beyond the matrix are not written).
struct SplitBatch {
    const float* src[CVT_MAX_SEG];
    uint16_t* dst[CVT_MAX_SEG];
    int start[CVT_MAX_SEG + 1];     // first workgroup of each segment
    int count[CVT_MAX_SEG];         // elements (multiple of 4)
    int K[CVT_MAX_SEG];
    int rb[CVT_MAX_SEG];
    int n;
};

__global__ __launch_bounds__(256) void split_weights_kernel(SplitBatch b) {
    int s = 0;
#pragma unroll
    for (int k = 1; k < CVT_MAX_SEG; ++k)
        if (k < b.n && (int)blockIdx.x >= b.start[k]) s = k;
    const int i = ((blockIdx.x - b.start[s]) * 256 + threadIdx.x) * 4;
    if (i >= b.count[s]) return;
    const int K = b.K[s];
    const int row = i / K, k = i - row * K;
    u32x2 hi, mid, lo;
    split4(*reinterpret_cast<const f32x4*>(b.src[s] + i), hi, mid, lo);
    const int rb = b.rb[s];
    uint16_t* d = rb ? b.dst[s] + ((size_t)(row / rb) * (K >> 4) + (k >> 4)) * ((size_t)rb * 48) + (size_t)(row % rb) * 48 + (k & 15)
                     : b.dst[s] + (size_t)row * 3 * K + (k >> 4) * 48 + (k & 15);
    *reinterpret_cast<u32x2*>(d) = hi;
    *reinterpret_cast<u32x2*>(d + 16) = mid;
    *reinterpret_cast<u32x2*>(d + 32) = lo;
}

}  // namespace mel
