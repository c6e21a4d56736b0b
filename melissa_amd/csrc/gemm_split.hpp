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
// weights are split once per forward call into [N][3][K] bf16 planes in the workspace (split_weights_kernel).
// LDS row = 3 planes x 64 B (K step 32) + 16 B pad = 208 B: the 16-byte fragment reads of 8 consecutive rows land on
// 8 distinct 4-bank groups.  One ds_read_b128 = one MFMA operand; 12 reads feed the 12 MFMAs of a K step.
// Same persistent tile loop, XCD-aware tile order and next-tile prefetch as gemm_f32_persistent_kernel.
#pragma once
#include "gemm_bf16.hpp"

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
        {   // weights: [N][3][K] bf16 planes (split_weights_kernel); W / W_hi point at plane 0 of row 0
            const int n = c.n0 + wrow;
            const uint16_t* base = (g.W_hi && n >= g.split_n)
                                       ? reinterpret_cast<const uint16_t*>(g.W_hi) + (size_t)(n - g.split_n) * 3 * g.K
                                       : reinterpret_cast<const uint16_t*>(g.W) + (size_t)n * 3 * g.K;
#pragma unroll
            for (int p = 0; p < 3; ++p) c.w_src[p] = reinterpret_cast<const u32x4*>(base + (size_t)p * g.K + wch * 8);
        }
    };

    int t = next_valid(blockIdx.x);
    if (t >= total) return;
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

    auto issue = [&](Regs& R) -> bool {       // loads of the next step of the stream; false when the stream is over
        if (!pf_valid) return false;
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) R.a[i] = fetch_a<MODE>(batch.p[0], pf.ac[i], pf_kt * GEMM_BK, kc, enc);
#pragma unroll
        for (int p = 0; p < 3; ++p) R.w[p] = pf.w_src[p][pf_kt * 4];
        if (++pf_kt == pf.KT) {               // cross into this workgroup's next tile
            const int tn = next_valid(pf_t + stride);
            if (tn < total) {
                setup(pf, tn);
                pf_t = tn, pf_kt = 0;
                nm = Meta{pf.m0, pf.n0, pf.M, pf.pi, pf.KT}, nm_valid = true;
            } else {
                pf_valid = false;
            }
        }
        return true;
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
    bool v0 = issue(R0);                       // step 0
    bool v1 = issue(R1);                       // step 1
    fill_stage(0, R0);
    __syncthreads();
    v0 = issue(R0);                            // step 2
    int stage = 0, ckt = 0;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    // one K step: MFMAs on `stage`, then Ra (step s+1) -> the other stage, barrier, then Ra <- loads of step s+3.
    // returns false when the stream is finished
    auto step = [&](Regs& Ra, bool& va) -> bool {
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
        if (va) fill_stage(stage ^ 1, Ra);
        __syncthreads();
        stage ^= 1;
        va = issue(Ra);
        if (++ckt < cm.KT) return true;
        // the tile is complete: epilogue, the fp32 kernel's
        store_block_f32(batch.p[cm.pi], acc, cm.m0 + wm * 32 + 4 * h, cm.n0 + wn * 32 + r, cm.M);
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        if (!nm_valid) return false;
        cm = nm, nm_valid = false, ckt = 0;
        return true;
    };
    for (;;) {
        if (!step(R1, v1)) break;
        if (!step(R0, v0)) break;
    }
}

// fp32 [rows, K] weight matrices -> [rows][3][K] bf16 planes (hi | mid | lo), all matrices of the model in one launch
struct SplitBatch {
    const float* src[CVT_MAX_SEG];
    uint16_t* dst[CVT_MAX_SEG];
    int start[CVT_MAX_SEG + 1];     // first workgroup of each segment
    int count[CVT_MAX_SEG];         // elements (multiple of 4)
    int K[CVT_MAX_SEG];
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
    uint16_t* d = b.dst[s] + (size_t)row * 3 * K + k;
    *reinterpret_cast<u32x2*>(d) = hi;
    *reinterpret_cast<u32x2*>(d + K) = mid;
    *reinterpret_cast<u32x2*>(d + 2 * K) = lo;
}

}  // namespace mel
