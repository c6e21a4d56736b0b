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

// per-env logits (HL-DGN): agent i of env b takes argmax / eps-greedy from the env's values with the stream of
// select_envs_kernel (key b * 64 * MEL_SET_WORDS(n_nodes) + i); act is the dense [bs, n_nodes] layout
__device__ __forceinline__ bool live_bit(const uint64_t* live, long b, int n_nodes, int i) {
    return (live[b * ((n_nodes + 63) >> 6) + (i >> 6)] >> (i & 63)) & 1ull;
}
__device__ __forceinline__ uint32_t env_agent_key(long b, int n_nodes, int i) {
    return (uint32_t)(b * (64 * ((n_nodes + 63) >> 6)) + i);
}
__device__ __forceinline__ void select_env_agent(const float (&q)[8], float mean, float v, int na, const mel_select& sel,
                                                 int b, int i) {
    int best = 0;
    float bv = -INFINITY;
#pragma unroll
    for (int a = 0; a < 8; ++a)
        if (a < na && q[a] - mean + v > bv) bv = q[a] - mean + v, best = a;
    if (sel.eps > 0.f) {
        const uint32_t step = sel.step_dev ? *sel.step_dev : 0u;
        const uint32_t base = mix32(sel.seed ^ mix32(step * 0x9e3779b9U + env_agent_key(b, sel.n_nodes, i)));
        if (u01(base) < sel.eps) {
            best = 0, bv = -1.f;
            for (int a = 0; a < na; ++a) {
                const float u = u01(mix32(base + 0x85ebca6bU * (uint32_t)(a + 1)));
                if (u > bv) bv = u, best = a;
            }
        }
    }
    sel.act[(size_t)b * sel.n_nodes + i] = best;
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
    if (sel.act && sel.live) {                // every lane holds the row's values after the butterfly sums
        for (int i = lane; i < sel.n_nodes; i += 64)
            if (live_bit(sel.live, b, sel.n_nodes, i)) select_env_agent(q, mean, v, na, sel, b, i);
    } else if (sel.act && lane == 0) {
        select_fused(q, mean, v, na, sel, b);
    }
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
// fused finish of the standard dueling heads (hidden [128, 128] for Q and for V) after the first layer's raw products:
//   h0 = relu(P_0 + P_1 + ... + bias0)  ->  h1 = relu(W1 h0 + b1) for Q and V (fp32 MFMA)  ->  last layer + combine + selection
// One workgroup of 8 wavefronts per 16 * MB rows (v_mfma_f32_16x16x4_f32, MB row blocks share a wave's W1 fragments);
// h0 / h1 live in LDS.  MB = 1 for small batches (half the matrix-pipe time per workgroup, twice the workgroups, two per
// CU covering each other's load / barrier phases: N = 20 config 16.3 -> 13.2 us); MB = 2 once there is more than one
// 16-row workgroup per CU - every workgroup pulls the 128 KB of W1 through the L2 (N = 50: 16.5 us against 18.1).
// ------------------------------------------------------------------------------------------------
constexpr int HF_W = 128;                  // hidden width per head
constexpr int HF_LD = 2 * HF_W + 4;        // LDS row stride (floats)
constexpr int HF_MAX_ACTIONS = 4;          // last-layer weights of Q live in registers (the reference has 2 actions)
typedef float f32x4_acc __attribute__((ext_vector_type(4)));
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
#ifdef MEL_FIN_PROF
// tuning builds: cycles of wave 0 of the busy workgroups from kernel start to [0] first barrier (operands fetched, h0 in LDS),
// [1] second barrier (hidden layer 1 done), [2] end; [3] workgroups counted; [4] kernel start to the row count known
__device__ unsigned long long g_fin_prof[5];
#endif
template <int MB>
__global__ __launch_bounds__(512, 4) void head_finish_kernel(HeadFinish f) {
    constexpr int HF_ROWS = 16 * MB;
#ifdef MEL_FIN_PROF
    const unsigned long long fp0 = __builtin_readcyclecounter();
    unsigned long long fp1 = fp0, fp2 = fp0, fpr = fp0;
#endif
    __shared__ __attribute__((aligned(16))) float h0[HF_ROWS * HF_LD];
    __shared__ __attribute__((aligned(16))) float h1[HF_ROWS * HF_LD];
    __shared__ float wl[(HF_MAX_ACTIONS + 1) * HF_W + HF_MAX_ACTIONS + 1];        // last-layer weights + biases
    const int rows = f.M_dev ? min(*f.M_dev, f.M) : f.M;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;              // MFMA 16x16x4: row / column index and k group of this lane
    const int hd = wid >> 2, cb = wid & 3;                 // head (0 Q, 1 V) and 32-column block of this wave's h1 tile
    // K order of the 32 MFMA steps: step 4t + u multiplies k = 16t + 4g + u of lane group g, so that a lane's operands of
    // four consecutive steps are ONE 16-byte chunk (column 16t + 4g) of its h0 row (A) and of its W1 row (B)
    const mel_linear& l1 = hd ? f.v1 : f.q1;
    const float* wrow0 = l1.weight + (size_t)(cb * 32 + li) * HF_W + 4 * lg;            // column block 0: n = cb*32 + li
    const float* wrow1 = wrow0 + (size_t)16 * HF_W;                                      // column block 1: n + 16
    f32x4 wf[2][8];
    // the grid is sized generously from the EXPECTED row count (the real one lives on the device): a surplus workgroup
    // must not pull its 128 KB of W1 through the L2 before it finds out that it has nothing to do
    const bool likely = (int)blockIdx.x < f.likely_blocks;
    if (!likely && (int)blockIdx.x * HF_ROWS >= rows) return;
#ifdef MEL_FIN_PROF
    asm volatile("s_nop 0" ::"s"(rows));
    fpr = __builtin_readcyclecounter();
#endif
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        wf[0][t] = *reinterpret_cast<const f32x4*>(wrow0 + 16 * t);
        wf[1][t] = *reinterpret_cast<const f32x4*>(wrow1 + 16 * t);
    }
    const float b1_0 = l1.bias[cb * 32 + li], b1_1 = l1.bias[cb * 32 + 16 + li];
    // last layer: 16 threads per row, thread j of a row takes columns k = j + 16 i of every output's weight row; the
    // weight rows (Q's na rows, then V's) and biases wait in LDS (visible after the loop's first barrier)
    const int na = f.q_last.out_dim, j16 = tid & 15, trow = tid >> 4;
    for (int i = tid; i < (na + 1) * HF_W; i += 512)
        wl[i] = i < na * HF_W ? f.q_last.weight[i] : f.v_last.weight[i - na * HF_W];
    if (tid <= na) wl[(HF_MAX_ACTIONS + 1) * HF_W + tid] = tid < na ? f.q_last.bias[tid] : f.v_last.bias[0];
    for (int m0 = blockIdx.x * HF_ROWS; m0 < rows; m0 += gridDim.x * HF_ROWS) {
        // 1. h0 = relu(sum of the planes in order + bias0): 16 MB rows x 64 chunks of four columns, 2 MB per thread
#pragma unroll
        for (int i = 0; i < 2 * MB; ++i) {
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
#ifdef MEL_FIN_PROF
        fp1 = __builtin_readcyclecounter();
#endif
        // 2. hidden layer 1: this wave's (16 MB) x 32 block (two 16 x 16 accumulators per row block) over K = 128
        f32x4_acc acc0[MB], acc1[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc0[mb] = f32x4_acc{0.f, 0.f, 0.f, 0.f}, acc1[mb] = f32x4_acc{0.f, 0.f, 0.f, 0.f};
        const float* arow = h0 + li * HF_LD + hd * HF_W + 4 * lg;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            f32x4 a[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) a[mb] = *reinterpret_cast<const f32x4*>(arow + mb * 16 * HF_LD + 16 * t);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    acc0[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb][u], wf[0][t][u], acc0[mb], 0, 0, 0);
                    acc1[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb][u], wf[1][t][u], acc1[mb], 0, 0, 0);
                }
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {   // C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + e
                float* dst = h1 + (mb * 16 + 4 * lg + e) * HF_LD + hd * HF_W + cb * 32 + li;
                dst[0] = fmaxf(acc0[mb][e] + b1_0, 0.f);
                dst[16] = fmaxf(acc1[mb][e] + b1_1, 0.f);
            }
        __syncthreads();
#ifdef MEL_FIN_PROF
        fp2 = __builtin_readcyclecounter();
#endif
        // 3. last layer + dueling combine + selection: 16 threads per row
        if (trow < HF_ROWS) {
            const float* hq = h1 + trow * HF_LD;
            float q[8], v = 0.f;
#pragma unroll
            for (int a = 0; a < 8; ++a) q[a] = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xq = hq[j16 + 16 * i], xv = hq[HF_W + j16 + 16 * i];
                v = fmaf(xv, wl[na * HF_W + j16 + 16 * i], v);
#pragma unroll
                for (int a = 0; a < HF_MAX_ACTIONS; ++a)
                    if (a < na) q[a] = fmaf(xq, wl[a * HF_W + j16 + 16 * i], q[a]);
            }
            v = row16_sum_f32(v);                  // the row's 16 lanes are one DPP row
#pragma unroll
            for (int a = 0; a < HF_MAX_ACTIONS; ++a)
                if (a < na) q[a] = row16_sum_f32(q[a]);
            const int b = m0 + trow;
            if (b < rows) {
                float qsum = 0.f;
#pragma unroll
                for (int a = 0; a < HF_MAX_ACTIONS; ++a)
                    if (a < na) q[a] += wl[(HF_MAX_ACTIONS + 1) * HF_W + a], qsum += q[a];
                v += wl[(HF_MAX_ACTIONS + 1) * HF_W + na];
                const float mean = qsum / (float)na;
#pragma unroll
                for (int a = 0; a < HF_MAX_ACTIONS; ++a)
                    if (a < na && j16 == a) f.logits[(size_t)b * na + a] = q[a] - mean + v;
                if (f.sel.act && f.sel.live) {        // the row's 16 lanes share its agents (all of them hold q and v)
                    for (int i = j16; i < f.sel.n_nodes; i += 16)
                        if (live_bit(f.sel.live, b, f.sel.n_nodes, i)) select_env_agent(q, mean, v, na, f.sel, b, i);
                } else if (f.sel.act && j16 == 0) {
                    select_fused(q, mean, v, na, f.sel, b);
                }
            }
        }
        // no barrier here: the next iteration's h0 writes come after this one's h0 reads (barrier 2 above), and its h1 writes
        // come after its own barrier 1, which no wave passes before it has finished these h1 reads
    }
#ifdef MEL_FIN_PROF
    if (tid == 0 && (int)blockIdx.x * HF_ROWS < rows) {
        const unsigned long long fp3 = __builtin_readcyclecounter();
        atomicAdd(&g_fin_prof[0], fp1 - fp0), atomicAdd(&g_fin_prof[1], fp2 - fp0), atomicAdd(&g_fin_prof[2], fp3 - fp0);
        atomicAdd(&g_fin_prof[3], 1ull), atomicAdd(&g_fin_prof[4], fpr - fp0);
    }
#endif
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
    if (!live_bit(live, b, n, i)) return;
    const float* q = logits + b * na;
    int best = 0;
    float bv = -INFINITY;
    for (int a = 0; a < na; ++a)
        if (q[a] > bv) bv = q[a], best = a;
    if (eps > 0.f) {
        const uint32_t step = step_dev ? *step_dev : 0u;
        const uint32_t base = mix32(seed ^ mix32(step * 0x9e3779b9U + env_agent_key(b, n, i)));
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
