// Row GEMM of the bf16 FEATURE PATH (BASELINE config "L-DGN 50-node, 1024 vectorised envs, bf16 feature path"):
//
//     Y[m, n] = act( rscale[m] * sum_k A(m, k) * W[n, k] + bias[n] )      A, W bf16; accumulate fp32
//
// on v_mfma_f32_32x32x16_bf16 (gfx950: 16x the rate of the exact-fp32 MFMA of gemm_f32.hpp).  Same problem
// description (GemmArgs), same roles as the fp32 kernel; in this kernel A / W / W_hi point at bf16 data and Y at
// bf16 (or fp32 when y_f32 is set: the last hidden layer feeds the fp32 tail).  bias / rscale / the encoder's
// first layer stay fp32.
//
// Layout / mapping (byte-for-byte the fp32 kernel's, with 2-byte elements):
//   * K step = 64 elements = 128 B per row; LDS rows padded to 144 B (9 chunks of 16 B) -> the 16 lanes of a
//     ds_read_b128 group hit 16 distinct 16-B slots;
//   * one ds_read_b128 per operand IS one MFMA operand (8 bf16 per lane): lane half h of sub-step q takes chunk
//     2q + h of its row, for A and W alike, so the k permutation cancels in the contraction;
//   * at this MFMA rate a 64x64 tile's arithmetic is ~0.4 us per CU: the kernel lives on keeping many tiles in
//     flight.  It is persistent (a fixed grid walks the tile list of all problems of the launch, ragged
//     device-side row counts included) and prefetches the next tile's first K step under the current tile's
//     last one, so no tile after the first pays a cold prologue.
#pragma once
#include "gemm_f32.hpp"

namespace mel {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// round-to-nearest-even pair conversion (v_cvt_pk_bf16_f32); lo lands in bits [0,16)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

constexpr int GEMB_BK = 64;        // elements per K step
constexpr int GEMB_ROW = 9;        // 16-byte chunks per LDS row (8 data + 1 pad)

struct BChunk {
    const u32x4* src;    // PLAIN: this thread's 16-byte chunk of K step 0 of its A row
    float x[8];          // ENC:   node features
};

// one 16-byte chunk (8 bf16) of A for K step kt; ENC: k = kt*64 + kc .. +7 of relu(enc_w x + enc_b)
template <int MODE>
__device__ __forceinline__ u32x4 fetch_a_bf16(const BChunk& c, int kt, int kc, const float* enc) {
    if constexpr (MODE == GEMM_MODE_PLAIN) {
        return c.src[kt * 8];
    } else {
        uint32_t w[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float* wrow = enc + (kt * GEMB_BK + kc + 2 * p + e) * 9;
                float s = wrow[8];
#pragma unroll
                for (int f = 0; f < 8; ++f) s = fmaf(wrow[f], c.x[f], s);
                v[e] = fmaxf(s, 0.f);
            }
            w[p] = pack_bf16x2(v[0], v[1]);
        }
        const u32x4 out = {w[0], w[1], w[2], w[3]};
        return out;
    }
}

template <int A_CHUNKS, int W_CHUNKS>
struct BTileCtx {
    BChunk ac[A_CHUNKS];
    const u32x4* w_src[W_CHUNKS];
    int m0, n0, M, pi, KT, ks;          // ks: split-K chunk of this work item (GemmArgs::ksplit)
};

template <int WM, int WN, int TM, int TN, int MODE>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_bf16_kernel(GemmBatch batch) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, T = 64 * WM * WN;
    constexpr int A_CHUNKS = BM * 8 / T;               // 16-byte chunks per thread per K step
    constexpr int W_CHUNKS = BN * 8 / T;
    constexpr int BUF = (BM + BN) * GEMB_ROW;          // chunks per LDS stage
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
            act[i] = ((rows[i] + BM - 1) / BM) * (q.N / BN) * (q.ksplit > 1 ? q.ksplit : 1);
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
    const int crow = tid >> 3;            // staging: 8 threads per 128-byte row slice
    const int cch = tid & 7;              // this thread's chunk of the slice
    const int kc = cch * 8;               // first element of the chunk

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
    auto setup = [&](BTileCtx<A_CHUNKS, W_CHUNKS>& c, int t) {
        int pi = 0;
#pragma unroll
        for (int k = 1; k < GEMM_MAX_GROUP; ++k)
            if (t >= pre[k]) pi = k;
        const GemmArgs& g = batch.p[pi];
        const int nbn = g.N / BN;
        int wg = t - pre[pi];
        {   // workgroups sharing an A row panel sit on one XCD (see gemm_f32.hpp)
            const int active = act[pi];
            const int q = active >> 3, r8 = active & 7, xcd = wg & 7, local = wg >> 3;
            wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + local;
        }
        // split-K (skinny long-K problems, see gemm_ring.hpp): work items of a row panel are ordered chunk-major; chunk ks
        // covers K columns [k0, k0 + KT * GEMB_BK) and writes raw fp32 products to plane ks
        const int S = g.ksplit > 1 ? g.ksplit : 1;
        c.pi = pi, c.M = rows[pi], c.KT = g.K / GEMB_BK / S, c.ks = (wg / nbn) % S;
        c.m0 = (wg / (nbn * S)) * BM, c.n0 = (wg % nbn) * BN;
        const int k0 = c.ks * c.KT * GEMB_BK;
        const uint16_t* A16 = reinterpret_cast<const uint16_t*>(g.A) + k0;
        const uint16_t* W16 = reinterpret_cast<const uint16_t*>(g.W) + k0;
        const uint16_t* Wh16 = g.W_hi ? reinterpret_cast<const uint16_t*>(g.W_hi) + k0 : nullptr;
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) {
            const int row = min(c.m0 + crow + i * (T / 8), c.M - 1);      // clamped, never predicated
            c.ac[i].src = nullptr;
#pragma unroll
            for (int f = 0; f < 8; ++f) c.ac[i].x[f] = 0.f;
            if constexpr (MODE == GEMM_MODE_PLAIN) {
                const int ar = g.arow ? g.arow[row] : row;
                c.ac[i].src = reinterpret_cast<const u32x4*>(A16 + (size_t)ar * g.lda + kc);
            } else {
                enc_features(g, row, c.ac[i].x);
            }
        }
#pragma unroll
        for (int i = 0; i < W_CHUNKS; ++i) {
            const int n = c.n0 + crow + i * (T / 8);
            const uint16_t* base = (Wh16 && n >= g.split_n) ? Wh16 + (size_t)(n - g.split_n) * g.K
                                                            : W16 + (size_t)n * g.K;
            c.w_src[i] = reinterpret_cast<const u32x4*>(base + kc);
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

    const int st_off = crow * GEMB_ROW + cch;                             // this thread's staging slot (chunks)
    const int a_off = (wm * 32 * TM + r) * GEMB_ROW + h;                  // this lane's fragment rows
    const int w_off = BM * GEMB_ROW + (wn * 32 * TN + r) * GEMB_ROW + h;

    BTileCtx<A_CHUNKS, W_CHUNKS> cur, nxt;
    setup(cur, t);
    nxt = cur;
    u32x4 a_reg[A_CHUNKS], w_reg[W_CHUNKS];
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) a_reg[i] = fetch_a_bf16<MODE>(cur.ac[i], 0, kc, enc);
#pragma unroll
    for (int i = 0; i < W_CHUNKS; ++i) w_reg[i] = cur.w_src[i][0];
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) lds[st_off + i * (T / 8) * GEMB_ROW] = a_reg[i];
#pragma unroll
    for (int i = 0; i < W_CHUNKS; ++i) lds[BM * GEMB_ROW + st_off + i * (T / 8) * GEMB_ROW] = w_reg[i];
    __syncthreads();
    int stage = 0;

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
            const u32x4* cst = lds + stage * BUF;
            u32x4* nst = lds + (stage ^ 1) * BUF;
            const bool last = kt + 1 == KT;
            if (has_next && (kt + 2 == KT || (KT == 1 && last))) setup(nxt, tn);
            const bool fill = !last || has_next;
            if (!last) {
#pragma unroll
                for (int i = 0; i < A_CHUNKS; ++i) a_reg[i] = fetch_a_bf16<MODE>(cur.ac[i], kt + 1, kc, enc);
#pragma unroll
                for (int i = 0; i < W_CHUNKS; ++i) w_reg[i] = cur.w_src[i][(kt + 1) * 8];
            } else if (has_next) {
#pragma unroll
                for (int i = 0; i < A_CHUNKS; ++i) a_reg[i] = fetch_a_bf16<MODE>(nxt.ac[i], 0, kc, enc);
#pragma unroll
                for (int i = 0; i < W_CHUNKS; ++i) w_reg[i] = nxt.w_src[i][0];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bf16x8 a[TM], b[TN];
#pragma unroll
                for (int u = 0; u < TM; ++u)
                    a[u] = __builtin_bit_cast(bf16x8, cst[a_off + u * 32 * GEMB_ROW + 2 * q]);
#pragma unroll
                for (int u = 0; u < TN; ++u)
                    b[u] = __builtin_bit_cast(bf16x8, cst[w_off + u * 32 * GEMB_ROW + 2 * q]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            if (fill) {
#pragma unroll
                for (int i = 0; i < A_CHUNKS; ++i) nst[st_off + i * (T / 8) * GEMB_ROW] = a_reg[i];
#pragma unroll
                for (int i = 0; i < W_CHUNKS; ++i) nst[BM * GEMB_ROW + st_off + i * (T / 8) * GEMB_ROW] = w_reg[i];
            }
            __syncthreads();
            stage ^= 1;
        }
        {   // epilogue (C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5))
            const GemmArgs& g = batch.p[cur.pi];
            uint16_t* Y16 = reinterpret_cast<uint16_t*>(g.Y);
            if (g.ksplit > 1) {                // raw partial products into this chunk's fp32 plane
                float* P = g.Y + (size_t)cur.ks * g.part_stride;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = cur.n0 + wn * 32 * TN + j * 32 + r;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int m = cur.m0 + wm * 32 * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                            if (m < cur.M) P[(size_t)m * g.ldy + n] = acc[i][j][e];
                        }
                }
            } else
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = cur.n0 + wn * 32 * TN + j * 32 + r;
                const float bias = (g.bias_hi && n >= g.split_n) ? g.bias_hi[n - g.split_n]
                                                                 : (g.bias ? g.bias[n] : 0.f);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int m = cur.m0 + wm * 32 * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                        if (m < cur.M) {
                            float v = acc[i][j][e];
                            if (g.rscale) v *= g.rscale[m];
                            v += bias;
                            if (g.relu) v = fmaxf(v, 0.f);
                            if (g.y_f32) g.Y[(size_t)m * g.ldy + n] = v;
                            else Y16[(size_t)m * g.ldy + n] = (uint16_t)pack_bf16x2(v, 0.f);
                        }
                    }
                }
            }
        }
        if (!has_next) break;
        cur = nxt;
        t = tn;
    }
}

// fp32 -> bf16 copies of the weight matrices the bf16 GEMMs read, one launch for all of them (the library keeps
// no state between calls: the copies live in the caller's workspace and are refreshed by every forward, ~1 M
// elements, so optimizer steps and load_state_dict are seen immediately, as on the fp32 path)
constexpr int CVT_MAX_SEG = 20;
struct CvtBatch {
    const float* src[CVT_MAX_SEG];
    uint16_t* dst[CVT_MAX_SEG];
    int start[CVT_MAX_SEG + 1];     // first workgroup of each segment
    int count[CVT_MAX_SEG];         // elements (multiple of 8)
    int n;
};

__global__ __launch_bounds__(256) void cvt_bf16_kernel(CvtBatch b) {
    int s = 0;
#pragma unroll
    for (int k = 1; k < CVT_MAX_SEG; ++k)
        if (k < b.n && (int)blockIdx.x >= b.start[k]) s = k;
    const int i = ((blockIdx.x - b.start[s]) * 256 + threadIdx.x) * 8;
    if (i >= b.count[s]) return;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(b.src[s] + i);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(b.src[s] + i + 4);
    const u32x4 o = {pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
    *reinterpret_cast<u32x4*>(b.dst[s] + i) = o;
}

}  // namespace mel
