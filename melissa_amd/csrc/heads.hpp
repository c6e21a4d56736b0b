// Dueling tail (last Linear of Q / V + q - mean(q) + v, l_dgn.py:142-147) with the fused action selection, and the
// standalone DQN action-selection kernels ([3P] tianshou DQNPolicy.forward / exploration_noise).
// Included by fwd.hip.
#pragma once
#include "common.hpp"

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

}  // namespace mel
