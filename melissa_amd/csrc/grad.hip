// Learn-path kernels: the edge-softmax + aggregate of GATv2Conv / TransformerConv over FULL
// graphs with a hand-written backward, and the global max / mean / add pool with its backward.  The Python side
// wraps them in torch.autograd.Function (melissa_amd/networks/autograd_ops.py); the dense projections around
// them are plain library GEMMs under torch autograd.
//
//   forward   out[i] = relu( sum_j alpha_ij s_j + bias )          alpha = softmax_j(e_ij), eps 1e-16 (PyG)
//     GATv2        e_ij = att . leaky_relu(x_r[i] + x_l[j], 0.2),  s_j = x_l[j],  j in adj(i) + {i}   (A.1)
//     Transformer  e_ij = (q[i] . k[j]) / sqrt(C),                 s_j = v[j],    j in adj(i)         (A.2)
//   backward (g = grad_out * (out > 0)):
//     d alpha_ij = g . s_j (per head)        S_i = sum_j alpha_ij d alpha_ij       d e_ij = alpha_ij (d alpha_ij - S_i)
//     GATv2        dz = d e_ij * att * leaky'(z),  d x_r[i] += dz,  d x_l[j] += alpha_ij g + dz,
//                  d att += d e_ij * leaky(z),     d bias += g
//     Transformer  d q[i] += d e_ij k[j] / sqrt(C),  d k[j] += d e_ij q[i] / sqrt(C),  d v[j] += alpha_ij g
//
// One wavefront per row, 64 lanes x VPL channels = heads x C (same mapping as the inference kernels of fwd.hip).  The
// backward stores no per-edge coefficient and uses NO atomics on the row gradients:
//   targets kernel  (a wave per target i)  three passes over i's source rows - softmax statistics m_i, 1 / l_i; S_i; then
//                   d x_r[i] (d q[i]), d att, d bias - and the three per-head scalars go to `stats`;
//   sources kernel  (a wave per source j)  walks the targets that read j (bit j of their source sets: a wave-uniform scan of
//                   the graph's n adjacency rows), recomputes e_ij / alpha_ij from stats[i] exactly as the targets kernel
//                   did, and sums j's gradient in target order into registers: d x_l[j] (d k[j]) and d v[j] are WRITTEN,
//                   once, deterministically.
// (Round 2's first version accumulated the source-row gradients with fp32 global atomics from the target side: ~190 M
// atomics for a DGN-R update of 673 graphs - 15 of the update's 20 ms.)
#include "common.hpp"

namespace mel {

template <int VPL>
struct GVec {
    float v[VPL];
};

template <int VPL>
__device__ __forceinline__ GVec<VPL> gload(const float* p) {
    GVec<VPL> r;
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
__device__ __forceinline__ void gstore(float* p, const GVec<VPL>& r) {
    if constexpr (VPL >= 4) {
#pragma unroll
        for (int i = 0; i < VPL / 4; ++i)
            reinterpret_cast<float4*>(p)[i] = make_float4(r.v[4 * i], r.v[4 * i + 1], r.v[4 * i + 2], r.v[4 * i + 3]);
    } else {
#pragma unroll
        for (int i = 0; i < VPL; ++i) p[i] = r.v[i];
    }
}

// sum over the lanes of one head (lanes_per_head adjacent lanes, a power of two)
__device__ __forceinline__ float head_total(float s, int lanes_per_head) {
    for (int o = lanes_per_head >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    return s;
}

struct GradArgs {
    const float* xl;        // GATv2: x_l (sources);  Transformer: keys
    const float* xv;        // Transformer: values (GATv2: = xl)
    const float* xr;        // GATv2: x_r (targets);  Transformer: queries
    const float* att;       // [HC] GATv2 only
    const float* bias;      // [HC] or null
    const uint64_t* adj;    // [rows, nw] sources of target row (bit j = node j of the same graph), self excluded
    int rows, n, nw, lanes_per_head, kind;       // nw = MEL_SET_WORDS(n)
    float scale;            // Transformer: 1 / sqrt(C)
    float* out;             // [rows, HC] relu(out + bias)
    // backward
    const float* gout;      // [rows, HC]
    float* dxl;             // written by the sources kernel
    float* dxv;
    float* dxr;             // written by the targets kernel
    float* stats;           // [rows, heads, 4] scratch: m_i, 1 / (l_i + eps), S_i per head (targets kernel -> sources kernel)
    int heads;
    float* datt;            // [HC] written by gat_reduce_partials_kernel
    float* dbias;           // [HC] likewise, or null
    float* partials;        // [workgroups of the targets kernel][2][HC]: their sums of d att | d bias
};

template <int VPL, int KIND>
__device__ __forceinline__ float edge_score(const GradArgs& a, const GVec<VPL>& xr, const GVec<VPL>& xl, const GVec<VPL>& att) {
    float t = 0.f;
    if constexpr (KIND == MEL_CONV_GATV2) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const float z = xr.v[i] + xl.v[i];
            t = fmaf(att.v[i], fmaxf(z, 0.2f * z), t);
        }
        return head_total(t, a.lanes_per_head);
    } else {
#pragma unroll
        for (int i = 0; i < VPL; ++i) t = fmaf(xr.v[i], xl.v[i], t);
        return head_total(t, a.lanes_per_head) * a.scale;
    }
}

// closed neighbourhood for GATv2 (self-loop added, A.1), open for TransformerConv (A.2)
// (the learn path is not the hot path: one two-word code path serves every graph size, the upper word is empty for n <= 64)
template <int KIND>
__device__ __forceinline__ NodeSet<2> sources_of(const GradArgs& a, int r) {
    NodeSet<2> m;
    m.w[0] = a.adj[(size_t)r * a.nw];
    m.w[1] = a.nw > 1 ? a.adj[(size_t)r * a.nw + 1] : 0ull;
    if (KIND == MEL_CONV_GATV2) m |= ns_bit<2>(r % a.n);
    return m;
}

template <int VPL, int KIND>
__global__ __launch_bounds__(256) void gat_full_forward_kernel(GradArgs a) {
    constexpr int HC = 64 * VPL;
    const int lane = lane_id();
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.rows) return;
    const int g0 = (r / a.n) * a.n;                          // first row of this graph
    GVec<VPL> att, bias;
#pragma unroll
    for (int i = 0; i < VPL; ++i) att.v[i] = 0.f, bias.v[i] = 0.f;
    if (KIND == MEL_CONV_GATV2) att = gload<VPL>(a.att + lane * VPL);
    if (a.bias) bias = gload<VPL>(a.bias + lane * VPL);
    const GVec<VPL> xr = gload<VPL>(a.xr + (size_t)r * HC + lane * VPL);
    float m = -INFINITY, l = 0.f;
    GVec<VPL> acc;
#pragma unroll
    for (int i = 0; i < VPL; ++i) acc.v[i] = 0.f;
    for (NodeSet<2> s = sources_of<KIND>(a, r); ns_any(s); ns_clear_lowest(s)) {
        const size_t row = (size_t)(g0 + ns_lowest(s)) * HC + lane * VPL;
        const GVec<VPL> xl = gload<VPL>(a.xl + row);
        const float e = edge_score<VPL, KIND>(a, xr, xl, att);
        const float mn = fmaxf(m, e);
        const float rs = expf(m - mn), pe = expf(e - mn);
        l = l * rs + pe;
        const GVec<VPL> sv = (KIND == MEL_CONV_GATV2) ? xl : gload<VPL>(a.xv + row);
#pragma unroll
        for (int i = 0; i < VPL; ++i) acc.v[i] = fmaf(pe, sv.v[i], acc.v[i] * rs);
        m = mn;
    }
    const float inv = 1.f / (l + 1e-16f);
    GVec<VPL> o;
#pragma unroll
    for (int i = 0; i < VPL; ++i) o.v[i] = fmaxf(acc.v[i] * inv + bias.v[i], 0.f);
    gstore<VPL>(a.out + (size_t)r * HC + lane * VPL, o);
}

template <int VPL, int KIND>
__global__ __launch_bounds__(256) void gat_backward_targets_kernel(GradArgs a) {
    constexpr int HC = 64 * VPL;
    const int lane = lane_id();
    GVec<VPL> att, datt, dbias;
#pragma unroll
    for (int i = 0; i < VPL; ++i) att.v[i] = 0.f, datt.v[i] = 0.f, dbias.v[i] = 0.f;
    if (KIND == MEL_CONV_GATV2) att = gload<VPL>(a.att + lane * VPL);
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < a.rows; r += gridDim.x * 4) {
        const int g0 = (r / a.n) * a.n;
        const size_t me = (size_t)r * HC + lane * VPL;
        const GVec<VPL> xr = gload<VPL>(a.xr + me);
        GVec<VPL> g = gload<VPL>(a.gout + me);
        {
            const GVec<VPL> o = gload<VPL>(a.out + me);
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                g.v[i] = o.v[i] > 0.f ? g.v[i] : 0.f;          // relu'
                dbias.v[i] += g.v[i];
            }
        }
        const NodeSet<2> src = sources_of<KIND>(a, r);
        // pass 1: softmax statistics
        float m = -INFINITY, l = 0.f;
        for (NodeSet<2> s = src; ns_any(s); ns_clear_lowest(s)) {
            const GVec<VPL> xl = gload<VPL>(a.xl + (size_t)(g0 + ns_lowest(s)) * HC + lane * VPL);
            const float e = edge_score<VPL, KIND>(a, xr, xl, att);
            const float mn = fmaxf(m, e);
            l = l * expf(m - mn) + expf(e - mn);
            m = mn;
        }
        const float inv = 1.f / (l + 1e-16f);
        // pass 2: S = sum_j alpha_ij (g . s_j)
        float S = 0.f;
        for (NodeSet<2> s = src; ns_any(s); ns_clear_lowest(s)) {
            const size_t row = (size_t)(g0 + ns_lowest(s)) * HC + lane * VPL;
            const GVec<VPL> xl = gload<VPL>(a.xl + row);
            const float alpha = expf(edge_score<VPL, KIND>(a, xr, xl, att) - m) * inv;
            const GVec<VPL> sv = (KIND == MEL_CONV_GATV2) ? xl : gload<VPL>(a.xv + row);
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) d = fmaf(g.v[i], sv.v[i], d);
            S = fmaf(alpha, head_total(d, a.lanes_per_head), S);
        }
        // pass 3: gradients
        GVec<VPL> dxr;
#pragma unroll
        for (int i = 0; i < VPL; ++i) dxr.v[i] = 0.f;
        for (NodeSet<2> s = src; ns_any(s); ns_clear_lowest(s)) {
            const size_t row = (size_t)(g0 + ns_lowest(s)) * HC + lane * VPL;
            const GVec<VPL> xl = gload<VPL>(a.xl + row);
            const float alpha = expf(edge_score<VPL, KIND>(a, xr, xl, att) - m) * inv;
            const GVec<VPL> sv = (KIND == MEL_CONV_GATV2) ? xl : gload<VPL>(a.xv + row);
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) d = fmaf(g.v[i], sv.v[i], d);
            const float de = alpha * (head_total(d, a.lanes_per_head) - S);
            if constexpr (KIND == MEL_CONV_GATV2) {
#pragma unroll
                for (int i = 0; i < VPL; ++i) {
                    const float z = xr.v[i] + xl.v[i];
                    const float dz = de * att.v[i] * (z > 0.f ? 1.f : 0.2f);
                    datt.v[i] = fmaf(de, fmaxf(z, 0.2f * z), datt.v[i]);
                    dxr.v[i] += dz;
                }
            } else {
                const float des = de * a.scale;
#pragma unroll
                for (int i = 0; i < VPL; ++i) dxr.v[i] = fmaf(des, xl.v[i], dxr.v[i]);
            }
        }
        gstore<VPL>(a.dxr + me, dxr);
        if (lane % a.lanes_per_head == 0) {                   // the head's three scalars (equal in all of its lanes)
            float* st = a.stats + ((size_t)r * a.heads + lane / a.lanes_per_head) * 4;
            st[0] = m, st[1] = inv, st[2] = S;
        }
    }
    // d att / d bias: the workgroup's four waves are summed in LDS (in wave order) and the sum goes to THIS workgroup's row of
    // the partials; gat_reduce_partials_kernel adds the rows in workgroup order.  (One atomicAdd per lane and wave onto the same
    // 2 HC addresses was 1.6 M contended atomics for a batch of 32 graphs: 317 us of a 1.1 ms update - and their order of
    // arrival made the last bits of these two gradients differ from run to run.)
    __shared__ float red[4][2][HC];
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < VPL; ++i) red[w][0][lane * VPL + i] = datt.v[i], red[w][1][lane * VPL + i] = dbias.v[i];
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * HC; c += 256) {
        const int which = c / HC, k = c - which * HC;
        a.partials[((size_t)blockIdx.x * 2 + which) * HC + k] = ((red[0][which][k] + red[1][which][k]) + red[2][which][k]) + red[3][which][k];
    }
}

// rows of [workgroups][2][HC] partial sums -> datt, dbias (fixed order: deterministic)
__global__ __launch_bounds__(256) void gat_reduce_partials_kernel(const float* __restrict__ partials, int groups, int hc, float* datt, float* dbias) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= 2 * hc) return;
    const int which = c / hc, k = c - which * hc;
    float* dst = which ? dbias : datt;
    if (!dst) return;
    float sum = 0.f;
    for (int g = 0; g < groups; ++g) sum += partials[((size_t)g * 2 + which) * hc + k];
    dst[k] = sum;
}

// gradient of the SOURCE rows: d x_l[j] (GATv2) / d k[j], d v[j] (TransformerConv), one wave per source row j
template <int VPL, int KIND>
__global__ __launch_bounds__(256) void gat_backward_sources_kernel(GradArgs a) {
    constexpr int HC = 64 * VPL;
    const int lane = lane_id();
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.rows) return;
    const int g0 = (r / a.n) * a.n, jn = r - g0;
    const int head = lane / a.lanes_per_head;
    GVec<VPL> att;
#pragma unroll
    for (int i = 0; i < VPL; ++i) att.v[i] = 0.f;
    if (KIND == MEL_CONV_GATV2) att = gload<VPL>(a.att + lane * VPL);
    const size_t me = (size_t)r * HC + lane * VPL;
    const GVec<VPL> xl = gload<VPL>(a.xl + me);                                     // x_l[j] / k[j]
    const GVec<VPL> sv = (KIND == MEL_CONV_GATV2) ? xl : gload<VPL>(a.xv + me);     // what the targets aggregate: x_l[j] / v[j]
    GVec<VPL> acc_l, acc_v;
#pragma unroll
    for (int i = 0; i < VPL; ++i) acc_l.v[i] = 0.f, acc_v.v[i] = 0.f;
    for (int t = 0; t < a.n; ++t) {                         // targets in id order: a deterministic sum
        const int ti = g0 + t;
        const uint64_t word = a.adj[(size_t)ti * a.nw + (jn >> 6)];                 // wave-uniform
        const bool reads_j = ((word >> (jn & 63)) & 1ull) || (KIND == MEL_CONV_GATV2 && t == jn);   // GATv2: self-loop
        if (!reads_j) continue;
        const size_t ti_off = (size_t)ti * HC + lane * VPL;
        const GVec<VPL> xr = gload<VPL>(a.xr + ti_off);
        GVec<VPL> g = gload<VPL>(a.gout + ti_off);
        {
            const GVec<VPL> o = gload<VPL>(a.out + ti_off);
#pragma unroll
            for (int i = 0; i < VPL; ++i) g.v[i] = o.v[i] > 0.f ? g.v[i] : 0.f;     // relu'
        }
        const float* st = a.stats + ((size_t)ti * a.heads + head) * 4;
        const float m = st[0], inv = st[1], S = st[2];
        const float alpha = expf(edge_score<VPL, KIND>(a, xr, xl, att) - m) * inv;  // the targets kernel's expression
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) d = fmaf(g.v[i], sv.v[i], d);
        const float de = alpha * (head_total(d, a.lanes_per_head) - S);
        if constexpr (KIND == MEL_CONV_GATV2) {
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const float z = xr.v[i] + xl.v[i];
                const float dz = de * att.v[i] * (z > 0.f ? 1.f : 0.2f);
                acc_l.v[i] += fmaf(alpha, g.v[i], dz);
            }
        } else {
            const float des = de * a.scale;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                acc_l.v[i] = fmaf(des, xr.v[i], acc_l.v[i]);
                acc_v.v[i] = fmaf(alpha, g.v[i], acc_v.v[i]);
            }
        }
    }
    gstore<VPL>(a.dxl + me, acc_l);
    if constexpr (KIND == MEL_CONV_TRANSFORMER) gstore<VPL>(a.dxv + me, acc_v);
}

// ---- global pool over the graph of (x * dm): hl_dgn.py:105-108 -------------------------------------------
// one thread per (graph, channel); arg = node of the first maximum (max only)
__global__ __launch_bounds__(256) void pool_forward_kernel(const float* __restrict__ x, const float* __restrict__ dm,
                                                           int bs, int n, int hc, int agg, float* __restrict__ pooled,
                                                           int32_t* __restrict__ arg) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)bs * hc) return;
    const int b = (int)(t / hc), c = (int)(t % hc);
    float best = (agg == MEL_AGG_MAX) ? -INFINITY : 0.f;
    int at = 0;
    for (int i = 0; i < n; ++i) {
        const float v = x[((size_t)b * n + i) * hc + c] * dm[(size_t)b * n + i];
        if (agg == MEL_AGG_MAX) {
            if (v > best) best = v, at = i;
        } else {
            best += v;
        }
    }
    if (agg == MEL_AGG_MEAN) best /= (float)n;
    pooled[t] = best;
    if (arg) arg[t] = at;
}

__global__ __launch_bounds__(256) void pool_backward_kernel(const float* __restrict__ g, const float* __restrict__ dm,
                                                            const int32_t* __restrict__ arg, int bs, int n, int hc, int agg,
                                                            float* __restrict__ dx) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)bs * n * hc) return;
    const int c = (int)(t % hc);
    const long row = t / hc;
    const int b = (int)(row / n), i = (int)(row % n);
    const float gv = g[(size_t)b * hc + c];
    float v;
    if (agg == MEL_AGG_MAX) v = (arg[(size_t)b * hc + c] == i) ? gv : 0.f;
    else if (agg == MEL_AGG_MEAN) v = gv / (float)n;
    else v = gv;
    dx[t] = v * dm[row];
}

static mel_status check_gat(const float* xl, const float* xr, const uint64_t* adj, int64_t bs, int n, int heads, int channels,
                            int kind, const float* att, const float* xv) {
    if (!xl || !xr || !adj) return fail(MEL_ERR_INVALID_ARG, "gat: null pointer");
    if (bs < 1 || n < 1 || n > MEL_MAX_NODES || bs * n > (1ll << 28)) return fail(MEL_ERR_INVALID_ARG, "gat: bs=%ld n=%d", (long)bs, n);
    const int hc = heads * channels;
    if (hc != 128 && hc != 256 && hc != 512 && hc != 1024)
        return fail(MEL_ERR_UNSUPPORTED, "gat: heads*channels = %d not in {128,256,512,1024}", hc);
    const int vpl = hc / 64;
    if (channels % vpl != 0 || ((channels / vpl) & (channels / vpl - 1)))
        return fail(MEL_ERR_UNSUPPORTED, "gat: channels per head (%d) must be a power-of-two multiple of %d", channels, vpl);
    if (kind == MEL_CONV_GATV2 && !att) return fail(MEL_ERR_INVALID_ARG, "gat: att is null");
    if (kind == MEL_CONV_TRANSFORMER && !xv) return fail(MEL_ERR_INVALID_ARG, "gat: values are null");
    if (kind != MEL_CONV_GATV2 && kind != MEL_CONV_TRANSFORMER) return fail(MEL_ERR_INVALID_ARG, "gat: kind %d", kind);
    return MEL_OK;
}

#define MEL_GRAD_DISPATCH(KERNEL, grid)                                                                      \
    switch (hc / 64) {                                                                                        \
        case 2: if (kind == MEL_CONV_GATV2) MEL_LAUNCH((KERNEL<2, MEL_CONV_GATV2>), dim3(grid), dim3(256), 0, s, a);   \
                else MEL_LAUNCH((KERNEL<2, MEL_CONV_TRANSFORMER>), dim3(grid), dim3(256), 0, s, a); break;            \
        case 4: if (kind == MEL_CONV_GATV2) MEL_LAUNCH((KERNEL<4, MEL_CONV_GATV2>), dim3(grid), dim3(256), 0, s, a);   \
                else MEL_LAUNCH((KERNEL<4, MEL_CONV_TRANSFORMER>), dim3(grid), dim3(256), 0, s, a); break;            \
        case 8: if (kind == MEL_CONV_GATV2) MEL_LAUNCH((KERNEL<8, MEL_CONV_GATV2>), dim3(grid), dim3(256), 0, s, a);   \
                else MEL_LAUNCH((KERNEL<8, MEL_CONV_TRANSFORMER>), dim3(grid), dim3(256), 0, s, a); break;            \
        default: if (kind == MEL_CONV_GATV2) MEL_LAUNCH((KERNEL<16, MEL_CONV_GATV2>), dim3(grid), dim3(256), 0, s, a); \
                 else MEL_LAUNCH((KERNEL<16, MEL_CONV_TRANSFORMER>), dim3(grid), dim3(256), 0, s, a); break;          \
    }

}  // namespace mel

using namespace mel;

namespace mel {
// dst[c, r] = src[r, c] through a 32 x 33 LDS tile (both sides coalesced).  The learn path's dense backward needs both
// operands of a product contiguous along the contraction index: dX = dY W wants W^T, dW = dY^T X wants dY^T and X^T.
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ src, int ld_src, int rows, int cols,
                                                            float* __restrict__ dst, int ld_dst) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8 threads
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (r < rows && c < cols) ? src[(size_t)r * ld_src + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (c < cols && r < rows) dst[(size_t)c * ld_dst + r] = tile[tx][ty + 8 * k];
    }
}
// ---- Adam, every parameter tensor of a group in ONE launch ([3P] torch.optim.Adam's update, l_dgn.py:207: the optimizer the
// reference trains with).  torch's capturable form - the one a HIP-graph capture needs - evaluates the bias corrections with a
// dozen one-element launches PER PARAMETER TENSOR (~200 launches of a 300-launch update); here the step counters stay torch's
// own state tensors, read by this kernel and advanced by a second one-workgroup launch.
struct AdamArgs {
    mel_adam_tensors t;
    float lr, beta1, beta2, eps, weight_decay;
    float bc1, bc2_sqrt;          // host-evaluated (eager form: the step counter is a host value), unused with device counters
};

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
    const int ti = blockIdx.y;
    float* __restrict__ p = a.t.param[ti];
    const float* __restrict__ g = a.t.grad[ti];
    float* __restrict__ m = a.t.exp_avg[ti];
    float* __restrict__ v = a.t.exp_avg_sq[ti];
    float bc1 = a.bc1, bc2s = a.bc2_sqrt;
    if (a.t.step[ti]) {
        const float step = *a.t.step[ti] + 1.0f;
        bc1 = 1.0f - powf(a.beta1, step);
        bc2s = sqrtf(1.0f - powf(a.beta2, step));
    }
    const float step_size = a.lr / bc1;
    const long n = a.t.numel[ti];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float gi = g[i];
        const float pi = p[i];
        if (a.weight_decay != 0.f) gi = gi + a.weight_decay * pi;
        const float mi = m[i] + (gi - m[i]) * (1.0f - a.beta1);                 // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * a.beta2 + ((1.0f - a.beta2) * gi) * gi;         // mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        const float denom = sqrtf(vi) / bc2s + a.eps;
        m[i] = mi, v[i] = vi;
        p[i] = pi + (-step_size) * (mi / denom);                                // addcdiv_(exp_avg, denom, value=-step_size)
    }
}

__global__ void adam_bump_kernel(mel_adam_tensors t) {
    const int i = threadIdx.x;
    if (i < t.count && t.step[i]) {
        bool first = true;                                   // (tensors may share one counter: bump it once)
        for (int k = 0; k < i; ++k) first = first && t.step[k] != t.step[i];
        if (first) *t.step[i] += 1.0f;
    }
}

}  // namespace mel

extern "C" {

mel_status mel_adam_step(const mel_adam_tensors* t, float lr, float beta1, float beta2, float eps, float weight_decay,
                         double host_step, void* stream) {
    if (!t || t->count < 1 || t->count > MEL_ADAM_MAX_TENSORS) return fail(MEL_ERR_INVALID_ARG, "mel_adam_step: 1 .. %d tensors per call", MEL_ADAM_MAX_TENSORS);
    bool dev_steps = t->step[0] != nullptr;
    for (int i = 0; i < t->count; ++i) {
        if (!t->param[i] || !t->grad[i] || !t->exp_avg[i] || !t->exp_avg_sq[i] || t->numel[i] < 1)
            return fail(MEL_ERR_INVALID_ARG, "mel_adam_step: tensor %d incomplete", i);
        if ((t->step[i] != nullptr) != dev_steps) return fail(MEL_ERR_INVALID_ARG, "mel_adam_step: step counters on the device for all tensors or for none");
    }
    if (!dev_steps && host_step < 1.0) return fail(MEL_ERR_INVALID_ARG, "mel_adam_step: host_step is the number of THIS update (>= 1)");
    clear_stale_error();
    AdamArgs a{};
    a.t = *t, a.lr = lr, a.beta1 = beta1, a.beta2 = beta2, a.eps = eps, a.weight_decay = weight_decay;
    if (!dev_steps) {                                        // as torch's eager form: Python doubles, then fp32 tensors ops
        a.bc1 = (float)(1.0 - pow((double)beta1, host_step));
        a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, host_step));
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    MEL_LAUNCH(adam_kernel, dim3(64, t->count), dim3(256), 0, s, a);
    if (mel_status st = check_launch("mel_adam_step")) return st;
    if (dev_steps) {
        MEL_LAUNCH(adam_bump_kernel, dim3(1), dim3(MEL_ADAM_MAX_TENSORS), 0, s, a.t);
        return check_launch("mel_adam_step (counters)");
    }
    return MEL_OK;
}

mel_status mel_transpose_f32(const float* src, int32_t ld_src, int64_t rows, int32_t cols, float* dst, int32_t ld_dst,
                             void* stream) {
    if (!src || !dst || rows < 0 || cols < 1 || ld_src < cols || ld_dst < rows || rows > (1ll << 30))
        return fail(MEL_ERR_INVALID_ARG, "mel_transpose_f32: bad arguments");
    if (rows == 0) return MEL_OK;
    clear_stale_error();
    MEL_LAUNCH(transpose_f32_kernel, dim3((cols + 31) / 32, (unsigned)((rows + 31) / 32)), dim3(256), 0,
               static_cast<hipStream_t>(stream), src, ld_src, (int)rows, cols, dst, ld_dst);
    return check_launch("mel_transpose_f32");
}

mel_status mel_gat_forward(const float* xl, const float* xv, const float* xr, const float* att, const float* bias,
                           const uint64_t* adj, int64_t bs, int32_t n, int32_t heads, int32_t channels, int32_t kind,
                           float* out, void* stream) {
    if (mel_status st = check_gat(xl, xr, adj, bs, n, heads, channels, kind, att, xv)) return st;
    if (!out) return fail(MEL_ERR_INVALID_ARG, "gat: out is null");
    clear_stale_error();
    const int hc = heads * channels;
    GradArgs a{};
    a.xl = xl, a.xv = (kind == MEL_CONV_GATV2) ? xl : xv, a.xr = xr, a.att = att, a.bias = bias, a.adj = adj;
    a.rows = (int)(bs * n), a.n = n, a.nw = set_words(n), a.lanes_per_head = channels / (hc / 64), a.kind = kind;
    a.scale = 1.0f / sqrtf((float)channels), a.out = out;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = (a.rows + 3) / 4;
    MEL_GRAD_DISPATCH(gat_full_forward_kernel, grid)
    return check_launch("mel_gat_forward");
}

mel_status mel_gat_backward(const float* xl, const float* xv, const float* xr, const float* att, const uint64_t* adj,
                            const float* out, const float* grad_out, int64_t bs, int32_t n, int32_t heads,
                            int32_t channels, int32_t kind, float* dxl, float* dxv, float* dxr, float* datt,
                            float* dbias, float* stats, void* stream) {
    if (mel_status st = check_gat(xl, xr, adj, bs, n, heads, channels, kind, att, xv)) return st;
    if (!out || !grad_out || !dxl || !dxr || !stats || (kind == MEL_CONV_GATV2 && !datt) || (kind == MEL_CONV_TRANSFORMER && !dxv))
        return fail(MEL_ERR_INVALID_ARG, "gat backward: null gradient / scratch buffer");
    clear_stale_error();
    const int hc = heads * channels;
    GradArgs a{};
    a.xl = xl, a.xv = (kind == MEL_CONV_GATV2) ? xl : xv, a.xr = xr, a.att = att, a.adj = adj;
    a.rows = (int)(bs * n), a.n = n, a.nw = set_words(n), a.lanes_per_head = channels / (hc / 64), a.kind = kind;
    a.scale = 1.0f / sqrtf((float)channels);
    a.out = const_cast<float*>(out), a.gout = grad_out, a.dxl = dxl, a.dxv = dxv, a.dxr = dxr, a.datt = datt, a.dbias = dbias;
    a.stats = stats, a.heads = heads;
    a.partials = stats + (size_t)a.rows * heads * 4;           // [MEL_GAT_PARTIAL_GROUPS][2][HC] behind the per-target statistics
    hipStream_t s = static_cast<hipStream_t>(stream);
    int grid = (a.rows + 3) / 4;
    if (grid > MEL_GAT_PARTIAL_GROUPS) grid = MEL_GAT_PARTIAL_GROUPS;      // grid-stride over the target rows
    MEL_GRAD_DISPATCH(gat_backward_targets_kernel, grid)
    if (mel_status st = check_launch("mel_gat_backward (targets)")) return st;
    MEL_LAUNCH(gat_reduce_partials_kernel, dim3((2 * hc + 255) / 256), dim3(256), 0, s, a.partials, grid, hc, datt, dbias);
    if (mel_status st = check_launch("mel_gat_backward (d att, d bias)")) return st;
    const int grid_s = (a.rows + 3) / 4;
    MEL_GRAD_DISPATCH(gat_backward_sources_kernel, grid_s)
    return check_launch("mel_gat_backward (sources)");
}

mel_status mel_pool_forward(const float* x, const float* dm, int64_t bs, int32_t n, int32_t hc, int32_t aggregator,
                            float* pooled, int32_t* arg, void* stream) {
    if (!x || !dm || !pooled || bs < 1 || n < 1 || hc < 1 || bs * n * (int64_t)hc > (1ll << 31))
        return fail(MEL_ERR_INVALID_ARG, "pool: bad arguments");
    if (aggregator < MEL_AGG_MAX || aggregator > MEL_AGG_ADD) return fail(MEL_ERR_INVALID_ARG, "aggregator %d", aggregator);
    if (aggregator == MEL_AGG_MAX && !arg) return fail(MEL_ERR_INVALID_ARG, "pool: max needs the arg buffer");
    clear_stale_error();
    const long total = (long)bs * hc;
    MEL_LAUNCH(pool_forward_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), x, dm,
                       (int)bs, n, hc, aggregator, pooled, arg);
    return check_launch("mel_pool_forward");
}

mel_status mel_pool_backward(const float* grad_pooled, const float* dm, const int32_t* arg, int64_t bs, int32_t n,
                             int32_t hc, int32_t aggregator, float* dx, void* stream) {
    if (!grad_pooled || !dm || !dx || bs < 1 || n < 1 || hc < 1 || bs * n * (int64_t)hc > (1ll << 31))
        return fail(MEL_ERR_INVALID_ARG, "pool backward: bad arguments");
    if (aggregator < MEL_AGG_MAX || aggregator > MEL_AGG_ADD) return fail(MEL_ERR_INVALID_ARG, "aggregator %d", aggregator);
    if (aggregator == MEL_AGG_MAX && !arg) return fail(MEL_ERR_INVALID_ARG, "pool backward: max needs the arg buffer");
    clear_stale_error();
    const long total = (long)bs * n * hc;
    MEL_LAUNCH(pool_backward_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       grad_pooled, dm, arg, (int)bs, n, hc, aggregator, dx);
    return check_launch("mel_pool_backward");
}

}  // extern "C"
