// Dueling tail (last Linear of Q / V + q - mean(q) + v, l_dgn.py:142-147) with the fused action selection, and the
// standalone DQN action-selection kernels ([3P] tianshou DQNPolicy.forward / exploration_noise).
// Included by fwd.hip.
#pragma once
#include "common.hpp"
#include "gemm_f32.hpp"

namespace mel {

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


// fused DQN action selection of row b from its dueling-combined values q[a] - mean + v
__device__ __forceinline__ void select_fused(const float (&q)[8], float mean, float v, int na, const mel_select& sel, int b) {
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

// last Linear of Q and V + dueling combine + selection for ONE row held by one wave; hq / hv: the row's hidden
// activations (global or LDS)
__device__ __forceinline__ void dueling_row(const float* hq, int kq, const float* hv, int kv, const mel_linear& q_last,
                                            const mel_linear& v_last, int dueling, int b, float* __restrict__ logits,
                                            const mel_select& sel) {
    const int lane = lane_id();
    const int na = q_last.out_dim;
    float q[8];
    float qsum = 0.f;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        q[a] = 0.f;
        if (a < na) {
            float s = 0.f;
            for (int k = lane; k < kq; k += 64) s = fmaf(hq[k], q_last.weight[(size_t)a * kq + k], s);
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            q[a] = s + q_last.bias[a];
            qsum += q[a];
        }
    }
    float v = 0.f, mean = 0.f;
    if (dueling) {
        for (int k = lane; k < kv; k += 64) v = fmaf(hv[k], v_last.weight[k], v);
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        v += v_last.bias[0];
        mean = qsum / (float)na;
    }
#pragma unroll
    for (int a = 0; a < 8; ++a)
        if (a < na && lane == a) logits[(size_t)b * na + a] = q[a] - mean + v;
    if (sel.act && lane == 0) select_fused(q, mean, v, na, sel, b);
}

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
    dueling_row(hq + (size_t)b * ldq, kq, hv + (size_t)b * ldv, kv, q_last, v_last, dueling, b, logits, sel);
}

// ------------------------------------------------------------------------------------------------
// fused finish of the standard dueling heads (hidden [128, 128] for Q and for V) after a split-K first layer:
//   h0 = relu(P_0 + P_1 + ... + bias0)  ->  h1 = relu(W1 h0 + b1) for Q and V (fp32 MFMA)  ->  last layer + combine + selection
// One workgroup of 8 wavefronts per 32 rows; h0 / h1 live in LDS; the arithmetic of each stage is the one the unfused
// launches perform (splitk_finish_kernel, the 64 x 64 GEMM's K order, dueling_row).
// ------------------------------------------------------------------------------------------------
constexpr int HF_W = 128;                  // hidden width per head
constexpr int HF_LD = 2 * HF_W + 4;        // LDS row stride (floats)
struct HeadFinish {
    const float* parts;                    // [S][rows_cap][2 * HF_W] raw first-layer products (Q | V)
    long part_stride;
    int S, M;
    const int32_t* M_dev;
    const float* bias0_q;
    const float* bias0_v;
    mel_linear q1, v1;                     // hidden layer 1
    mel_linear q_last, v_last;
    float* logits;
    mel_select sel;
    int likely_blocks;                     // workgroups below this index expect rows: they fetch W1 before the row count is known
};
__global__ __launch_bounds__(512) void head_finish_kernel(HeadFinish f) {
    __shared__ __attribute__((aligned(16))) float h0[32 * HF_LD];
    __shared__ __attribute__((aligned(16))) float h1[32 * HF_LD];
    const int rows = f.M_dev ? min(*f.M_dev, f.M) : f.M;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int hd = wid >> 2, cb = wid & 3;                 // head (0 Q, 1 V) and 32-column block of this wave's h1 tile
    // this lane's W1 fragments: row n = cb*32 + r of the head's [128, 128] matrix, chunks (2q + h) of four k each
    const mel_linear& l1 = hd ? f.v1 : f.q1;
    const float* wrow = l1.weight + (size_t)(cb * 32 + r) * HF_W;
    f32x4 wf[16];
    // the grid is sized generously from the EXPECTED row count (the real one lives on the device): a surplus workgroup
    // must not pull its 128 KB of W1 through the L2 before it finds out that it has nothing to do
    const bool likely = (int)blockIdx.x < f.likely_blocks;
    if (likely) {
#pragma unroll
        for (int q = 0; q < 16; ++q) wf[q] = *reinterpret_cast<const f32x4*>(wrow + (2 * q + h) * 4);
    } else {
        if ((int)blockIdx.x * 32 >= rows) return;
#pragma unroll
        for (int q = 0; q < 16; ++q) wf[q] = *reinterpret_cast<const f32x4*>(wrow + (2 * q + h) * 4);
    }
    const float b1 = l1.bias[cb * 32 + r];
    // last layer: 16 threads per row, thread j of a row holds columns k = j + 16 i of every output's weight row
    const int na = f.q_last.out_dim, j16 = tid & 15, trow = tid >> 4;
    float wq[8][8], wv[8], bq[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        bq[a] = a < na ? f.q_last.bias[a] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) wq[a][i] = a < na ? f.q_last.weight[(size_t)a * HF_W + j16 + 16 * i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) wv[i] = f.v_last.weight[j16 + 16 * i];
    const float bv = f.v_last.bias[0];
    for (int m0 = blockIdx.x * 32; m0 < rows; m0 += gridDim.x * 32) {
        // 1. h0 = relu(sum of the planes in order + bias0): 32 rows x 64 chunks of four columns, four per thread
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * 512, row = c >> 6, col = (c & 63) * 4;
            const int m = min(m0 + row, f.M - 1);
            const float* p = f.parts + (size_t)m * (2 * HF_W) + col;
            f32x4 pl[4];
#pragma unroll
            for (int s = 0; s < 4; ++s)
                pl[s] = s < f.S ? *reinterpret_cast<const f32x4*>(p + (size_t)s * f.part_stride) : f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 acc = pl[0];
#pragma unroll
            for (int s = 1; s < 4; ++s)
                if (s < f.S) acc = f32x4{acc[0] + pl[s][0], acc[1] + pl[s][1], acc[2] + pl[s][2], acc[3] + pl[s][3]};
            const f32x4 b = *reinterpret_cast<const f32x4*>(col < HF_W ? f.bias0_q + col : f.bias0_v + (col - HF_W));
            *reinterpret_cast<f32x4*>(h0 + row * HF_LD + col) =
                f32x4{fmaxf(acc[0] + b[0], 0.f), fmaxf(acc[1] + b[1], 0.f), fmaxf(acc[2] + b[2], 0.f), fmaxf(acc[3] + b[3], 0.f)};
        }
        __syncthreads();
        // 2. hidden layer 1: this wave's 32 x 32 block over K = 128
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const float* arow = h0 + r * HF_LD + hd * HF_W;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(arow + (2 * q + h) * 4);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], wf[q][kk], acc, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e)        // C/D layout: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * h
            h1[((e & 3) + 8 * (e >> 2) + 4 * h) * HF_LD + hd * HF_W + cb * 32 + r] = fmaxf(acc[e] + b1, 0.f);
        __syncthreads();
        // 3. last layer + dueling combine + selection: 16 threads per row
        {
            const float* hq = h1 + trow * HF_LD;
            float q[8], v = 0.f;
#pragma unroll
            for (int a = 0; a < 8; ++a) q[a] = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xq = hq[j16 + 16 * i], xv = hq[HF_W + j16 + 16 * i];
                v = fmaf(xv, wv[i], v);
#pragma unroll
                for (int a = 0; a < 8; ++a)
                    if (a < na) q[a] = fmaf(xq, wq[a][i], q[a]);
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                v += __shfl_xor(v, o, 64);
#pragma unroll
                for (int a = 0; a < 8; ++a)
                    if (a < na) q[a] += __shfl_xor(q[a], o, 64);
            }
            const int b = m0 + trow;
            if (b < rows) {
                float qsum = 0.f;
#pragma unroll
                for (int a = 0; a < 8; ++a)
                    if (a < na) q[a] += bq[a], qsum += q[a];
                v += bv;
                const float mean = qsum / (float)na;
#pragma unroll
                for (int a = 0; a < 8; ++a)
                    if (a < na && j16 == a) f.logits[(size_t)b * na + a] = q[a] - mean + v;
                if (f.sel.act && j16 == 0) select_fused(q, mean, v, na, f.sel, b);
            }
        }
        // no barrier here: the next iteration's h0 writes come after this one's h0 reads (barrier 2 above), and its h1 writes
        // come after its own barrier 1, which no wave passes before it has finished these h1 reads
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

}  // namespace mel
