// GATv2 / TransformerConv edge softmax + aggregation kernels of the inference path and the
// HL-DGN attention + graph pool (hl_dgn.py:101-108).  Included by fwd.hip after plan.hpp (TargetDesc).
#pragma once
#include "plan.hpp"
#include "gemm_bf16.hpp"   // bf16 pack / unpack helpers

namespace mel {

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
    const uint64_t* adj;    // [bs*N] node sets (MEL_SET_WORDS(n) words each): ATT_POOL; the row kernels read TargetDesc
    int bs, n, lanes_per_head;
    int kind;               // MEL_CONV_*
    float score_scale;      // TransformerConv: 1 / sqrt(C)
    // ATT_ROWS
    float* out;             // [rows, ldo] relu(out + bias)
    int ldo;
    const float* out_scale; // optional [rows]: row r of `out` is stored times out_scale[r] (the decision-maker mask of
                            // l_dgn.py:128 - the x_2 snapshot in xcat is taken before it, l_dgn.py:127)
    uint16_t* out_planes;   // optional (fp32 path, 4+ features per lane): the rows go out ALREADY SPLIT into bf16 planes,
                            // [rows][ldo / 16][3][16] - gemm_planes_kernel's A operand - instead of fp32 rows to `out`: the same
                            // split4() the split GEMMs apply to fp32 rows on their way into LDS
    float* xcat;            // [R, ld_cat] head input: x_1 | x_2 | x_3 (l_dgn.py:139)
    int ld_cat, hidden;
    const float* h0;        // encoder rows (packed by smask), [*, hidden]
    // ATT_POOL
    const float* obs;       // dm flag source
    int obs_stride, node_cols, aggregator;
    float* pooled;          // [bs, heads*C]
    // ATT_ROWS / ATT_SINGLE: one wavefront per target row
    const void* desc;           // [rows] per-target descriptor (TargetDesc<W>)
    const int32_t* rows_dev;    // device-side row count
    long rows_hint;             // expected rows (grid sizing only)
    int rows_cap, cat_off;
    int bf16;                   // bf16 feature path: xl / xr / out / xcat / h0 hold bf16 rows (att / bias stay fp32)
    // node-feature table mode (plan_masks.hpp): xl / xr / h0 are TABLES indexed by a node's tuple id fid[b*N + i] instead of
    // row lists indexed by packed position (null = row lists)
    const int32_t* fid;
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
// my_fid: table mode - the tuple ids of this lane's nodes (node lane + 64 h of the target's env); a source row is then the
// table row of the source's tuple (one v_readlane) instead of its packed position in the row list.
template <int VPL, int KIND, bool BF, int W, int G>
__device__ __forceinline__ Vec<VPL> attend_target(const AttArgs& a, size_t xr_row, NodeSet<W> sources,
                                                  const NodeSet<W>& smask, int soff, const Vec<VPL>& att,
                                                  const Vec<VPL>& bias, int lane, const int (&my_fid)[W]) {
    const bool table = a.fid != nullptr;
    constexpr int HC = 64 * VPL;
    // G = sources per online-softmax step.  Row kernels: 2 - with 4 the kernel needs 112 VGPRs (four waves per SIMD), with 2 it
    // fits 96 (five): conv1 attention 24.3 -> 22.8 us, conv2 attention 13.4 -> 13.0 us, step 0.2180 -> 0.2142 ms (two A/B
    // pairs; 3: 24.1 us).  The HL-DGN pool kernel: 2 as well (MEL_ATT_GP: 39.9 -> 33.5 us; 1: 36.4, 3: 34.5).
    const Vec<VPL> xr = load_row<VPL, BF>(a.xr, xr_row * a.ld_r + lane * VPL);
    float m = -INFINITY, l = 0.f;
    Vec<VPL> acc;
#pragma unroll
    for (int i = 0; i < VPL; ++i) acc.v[i] = 0.f;
    while (ns_any(sources)) {                    // TransformerConv adds no self-loop: a target may be isolated
        size_t row[G];                           // element index of this lane's slice of the source row
        bool on[G];
#pragma unroll
        for (int k = 0; k < G; ++k) {
            on[k] = ns_any(sources);
            const int j = on[k] ? ns_lowest(sources) : 0;
            ns_clear_lowest(sources);            // empty stays empty
            const int srow = table ? node_i32<W>(my_fid, j) : soff + (on[k] ? ns_rank_below(smask, j) : 0);
            row[k] = (size_t)srow * a.ld_l + lane * VPL;                     // off slots re-read a valid row
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

// row r of a [rows, K] matrix as bf16 planes (gemm_planes_kernel's A operand): this lane's VPL consecutive features
// k = lane * VPL ..; the 96 bytes of 16-k step k >> 4 hold hi | mid | lo of its 16 values
template <int VPL>
__device__ __forceinline__ void store_row_planes(uint16_t* base, int r, int K, int lane, const Vec<VPL>& o) {
    static_assert(VPL % 4 == 0, "planes need whole groups of four features per lane");
    uint16_t* row = base + (size_t)r * 3 * K;
    if constexpr (VPL % 8 == 0) {                 // eight features = half of a 16-k step: one 16-byte store per plane
#pragma unroll
        for (int i = 0; i < VPL; i += 8) {
            const int k = lane * VPL + i;
            u32x2 h0, m0, l0, h1, m1, l1;
            split4(f32x4{o.v[i], o.v[i + 1], o.v[i + 2], o.v[i + 3]}, h0, m0, l0);
            split4(f32x4{o.v[i + 4], o.v[i + 5], o.v[i + 6], o.v[i + 7]}, h1, m1, l1);
            uint16_t* d = row + (k >> 4) * 48 + (k & 15);
            *reinterpret_cast<u32x4*>(d) = u32x4{h0[0], h0[1], h1[0], h1[1]};
            *reinterpret_cast<u32x4*>(d + 16) = u32x4{m0[0], m0[1], m1[0], m1[1]};
            *reinterpret_cast<u32x4*>(d + 32) = u32x4{l0[0], l0[1], l1[0], l1[1]};
        }
    } else {
#pragma unroll
        for (int i = 0; i < VPL; i += 4) {
            const int k = lane * VPL + i;
            u32x2 hi, mid, lo;
            split4(f32x4{o.v[i], o.v[i + 1], o.v[i + 2], o.v[i + 3]}, hi, mid, lo);
            uint16_t* d = row + (k >> 4) * 48 + (k & 15);
            *reinterpret_cast<u32x2*>(d) = hi;
            *reinterpret_cast<u32x2*>(d + 16) = mid;
            *reinterpret_cast<u32x2*>(d + 32) = lo;
        }
    }
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
#ifdef MEL_ATT_PROF
// Tuning builds (-DMEL_ATT_PROF=<MODE>): cycles per wave of the rows kernel with that MODE (0 = conv1, 2 = conv2) in
// [0] prologue (row count, att / bias), [1] descriptor load, [2] attend_target (row loads + scores + softmax + sum),
// [3] stores, [4] whole wave, [5] waves counted, [6] rows processed
__device__ unsigned long long g_att_prof[8];
#endif

#ifndef MEL_ATT_MINB
#define MEL_ATT_MINB 2
#endif
#ifndef MEL_ATT_G
#define MEL_ATT_G 2      // sources per online-softmax step of the row kernels (see attend_target)
#endif
#ifndef MEL_POOL_NW
#define MEL_POOL_NW 8    // waves per env of the pool kernel below 2 048 envs (16 -> 8 with two sources per step: 33.4 -> 31.6 us; 12: 35.1)
#endif
#ifndef MEL_ATT_GP
#define MEL_ATT_GP 2     // ... of the HL-DGN pool kernel (4 -> 2: pool attention 39.9 -> 33.5 us, HL-DGN 512 envs 24.4 -> 26.1 M/s)
#endif
template <int VPL, int MODE, int KIND, bool BF, int W>
__global__ __launch_bounds__(256, MEL_ATT_MINB) void gat_attend_rows_kernel(AttArgs a) {
#ifdef MEL_ATT_PROF
    const unsigned long long p0 = __builtin_readcyclecounter();
    unsigned long long pd = 0, pa = 0, ps = 0, prows = 0;
#endif
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
#ifdef MEL_ATT_PROF
    asm volatile("s_nop 0" ::"v"(att.v[0]), "v"(bias.v[0]), "s"(rows));
    const unsigned long long p1 = __builtin_readcyclecounter();
#endif
    const int per_xcd = gridDim.x >> 3;
    const int vblock = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int rows_pad = ((rows + 4 * (int)gridDim.x - 1) / (4 * (int)gridDim.x)) * (4 * (int)gridDim.x);
    const int span = rows_pad >> 3;              // rows each XCD owns (multiple of 4 * per_xcd)
    for (int i = (vblock % per_xcd) * 4 + (threadIdx.x >> 6); i < span; i += per_xcd * 4) {
        const int r = (blockIdx.x & 7) * span + i;
        if (r >= rows) continue;
#ifdef MEL_ATT_PROF
        const unsigned long long q0 = __builtin_readcyclecounter();
#endif
        const TargetDesc<W> d = static_cast<const TargetDesc<W>*>(a.desc)[r];   // one 32-byte record: no chain of dependent index loads
#ifdef MEL_ATT_PROF
        asm volatile("s_nop 0" ::"s"(d.sources.w[0]), "s"(d.soff));
        const unsigned long long q1 = __builtin_readcyclecounter();
#endif
        int my_fid[W];
        MEL_W_FOR(h) my_fid[h] = 0;
        size_t xr_row = (size_t)r;
        if (MODE == ATT_ROWS && a.fid) {             // table mode: tuple ids of the target's env (one dependent load)
            MEL_W_FOR(h) my_fid[h] = lane + 64 * h < a.n ? a.fid[(size_t)d.env * a.n + lane + 64 * h] : 0;
            xr_row = (size_t)node_i32<W>(my_fid, d.node);
        }
        const Vec<VPL> o = attend_target<VPL, KIND, BF, W, MEL_ATT_G>(a, xr_row, d.sources, d.smask, d.soff, att, bias, lane, my_fid);
#ifdef MEL_ATT_PROF
        asm volatile("s_nop 0" ::"v"(o.v[0]));
        const unsigned long long q2 = __builtin_readcyclecounter();
#endif
        if constexpr (MODE == ATT_SINGLE) {
            store_row<VPL, BF>(a.xcat, (size_t)r * a.ld_cat + a.cat_off + lane * VPL, o);
        } else {
            Vec<VPL> om = o;
            if (a.out_scale) {
                const float dmv = a.out_scale[r];
#pragma unroll
                for (int i = 0; i < VPL; ++i) om.v[i] = o.v[i] * dmv;
            }
            bool as_rows = true;
            if constexpr (!BF && VPL % 4 == 0) {
                if (a.out_planes) store_row_planes<VPL>(a.out_planes, r, a.ldo, lane, om), as_rows = false;
            }
            if (as_rows) store_row<VPL, BF>(a.out, (size_t)r * a.ldo + lane * VPL, om);
            if (d.cat_row >= 0) {
                const size_t cat = (size_t)d.cat_row * a.ld_cat;
                // x_2: the controlling agent's conv1 row BEFORE the decision-maker mask (l_dgn.py:127)
                store_row<VPL, BF>(a.xcat, cat + a.hidden + lane * VPL, o);
                // x_1: its encoder row (l_dgn.py:122)
                const size_t h0 = (size_t)(a.fid ? node_i32<W>(my_fid, d.node) : d.soff + ns_rank_below(d.smask, d.node)) * a.hidden;
                if constexpr (BF) {
                    uint16_t* dst = reinterpret_cast<uint16_t*>(a.xcat) + cat;
                    const uint16_t* src = reinterpret_cast<const uint16_t*>(a.h0) + h0;
                    for (int c = lane; c < a.hidden; c += 64) dst[c] = src[c];
                } else {
                    for (int c = lane; c < a.hidden; c += 64) a.xcat[cat + c] = a.h0[h0 + c];
                }
            }
        }
#ifdef MEL_ATT_PROF
        const unsigned long long q3 = __builtin_readcyclecounter();
        pd += q1 - q0, pa += q2 - q1, ps += q3 - q2, prows += 1;
#endif
    }
#ifdef MEL_ATT_PROF
    if (threadIdx.x == 0 && (blockIdx.x & 31) == 0 && MODE == MEL_ATT_PROF && !BF) {      // a sample: few atomics
        atomicAdd(&g_att_prof[0], p1 - p0), atomicAdd(&g_att_prof[1], pd), atomicAdd(&g_att_prof[2], pa);
        atomicAdd(&g_att_prof[3], ps), atomicAdd(&g_att_prof[4], __builtin_readcyclecounter() - p0);
        atomicAdd(&g_att_prof[5], 1ull), atomicAdd(&g_att_prof[6], prows);
    }
#endif
}

// ATT_POOL (HL-DGN): one workgroup per env (every env has exactly N targets, so this is balanced):
// conv1 attention for all nodes, decision-maker mask, max / mean / add pool over the graph.
// NW wavefronts per env: at 512 envs per GPU four waves per workgroup leave a CU with 8 resident waves (two workgroups),
// too few to hide the source-row latency; eight waves (MEL_POOL_NW; six or seven targets each at N = 50) with two source rows
// per step measure best (sixteen: 33.4 us, twelve: 35.1, eight: 31.6).
template <int VPL, bool BF, int NW, int W>
__global__ __launch_bounds__(64 * NW) void gat_attend_pool_kernel(AttArgs a) {
    constexpr int HC = 64 * VPL;
    __shared__ float part[NW][HC];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.x;
    const Vec<VPL> att = load_vec<VPL>(a.att + lane * VPL);
    const Vec<VPL> bias = load_vec<VPL>(a.bias + lane * VPL);
    const NodeSet<W> full = ns_full<W>(a.n);
    Vec<VPL> pool;
#pragma unroll
    for (int i = 0; i < VPL; ++i) pool.v[i] = (a.aggregator == MEL_AGG_MAX) ? -INFINITY : 0.f;
    int my_fid[W];                                                                    // table mode (null: rows b*N + i)
    MEL_W_FOR(h) my_fid[h] = (a.fid && lane + 64 * h < a.n) ? a.fid[(size_t)b * a.n + lane + 64 * h] : 0;
    for (int t = wave; t < a.n; t += NW) {
        const NodeSet<W> sources = ns_load<W>(a.adj, (size_t)b * a.n + t) | ns_bit<W>(t);
        const size_t xr_row = a.fid ? (size_t)node_i32<W>(my_fid, t) : (size_t)(b * a.n + t);
        const Vec<VPL> o = attend_target<VPL, MEL_CONV_GATV2, BF, W, MEL_ATT_GP>(a, xr_row, sources, full, b * a.n, att, bias, lane, my_fid);
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
    for (int c = threadIdx.x; c < HC; c += 64 * NW) {
        float v = part[0][c];
        if (a.aggregator == MEL_AGG_MAX) {
#pragma unroll
            for (int w = 1; w < NW; ++w) v = fmaxf(v, part[w][c]);
        } else {
#pragma unroll
            for (int w = 1; w < NW; ++w) v += part[w][c];
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
#define MEL_POOL_LAUNCH(V)                                                                                         \
    if (a.n > 64) {             /* graphs of 65 .. 128 nodes: two-word node sets */                               \
        if (a.bf16) MEL_LAUNCH((gat_attend_pool_kernel<V, true, 16, 2>), dim3(a.bs), dim3(1024), 0, s, a);        \
        else MEL_LAUNCH((gat_attend_pool_kernel<V, false, 16, 2>), dim3(a.bs), dim3(1024), 0, s, a);              \
    } else if (a.bs >= 2048) {                                                                                    \
        if (a.bf16) MEL_LAUNCH((gat_attend_pool_kernel<V, true, 4, 1>), dim3(a.bs), dim3(256), 0, s, a);  \
        else MEL_LAUNCH((gat_attend_pool_kernel<V, false, 4, 1>), dim3(a.bs), dim3(256), 0, s, a);        \
    } else {                                                                                                      \
        if (a.bf16) MEL_LAUNCH((gat_attend_pool_kernel<V, true, MEL_POOL_NW, 1>), dim3(a.bs), dim3(64 * MEL_POOL_NW), 0, s, a);   \
        else MEL_LAUNCH((gat_attend_pool_kernel<V, false, MEL_POOL_NW, 1>), dim3(a.bs), dim3(64 * MEL_POOL_NW), 0, s, a);         \
    }
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
#ifdef MEL_ATT_GRID_CAP
        if (want > MEL_ATT_GRID_CAP) want = MEL_ATT_GRID_CAP;      // tuning: resident workgroups only, every wave loops
#endif
#ifdef MEL_ATT2_GRID_CAP
        if (MODE == ATT_SINGLE && want > MEL_ATT2_GRID_CAP) want = MEL_ATT2_GRID_CAP;
#endif
        const int grid = (int)((want + 7) & ~7L);       // multiple of 8: block id % 8 = XCD
#define MEL_ATT_LAUNCH_W(V, WW)                                                                                          \
    if (a.kind == MEL_CONV_TRANSFORMER && a.bf16)                                                                     \
        MEL_LAUNCH((gat_attend_rows_kernel<V, MODE, MEL_CONV_TRANSFORMER, true, WW>), dim3(grid), dim3(256), 0, s, a);  \
    else if (a.kind == MEL_CONV_TRANSFORMER)                                                                          \
        MEL_LAUNCH((gat_attend_rows_kernel<V, MODE, MEL_CONV_TRANSFORMER, false, WW>), dim3(grid), dim3(256), 0, s, a); \
    else if (a.bf16)                                                                                                  \
        MEL_LAUNCH((gat_attend_rows_kernel<V, MODE, MEL_CONV_GATV2, true, WW>), dim3(grid), dim3(256), 0, s, a);        \
    else                                                                                                              \
        MEL_LAUNCH((gat_attend_rows_kernel<V, MODE, MEL_CONV_GATV2, false, WW>), dim3(grid), dim3(256), 0, s, a);
#define MEL_ATT_LAUNCH(V)                                                                                             \
    if (a.n > 64) {                                                                                                   \
        MEL_ATT_LAUNCH_W(V, 2)                                                                                        \
    } else {                                                                                                          \
        MEL_ATT_LAUNCH_W(V, 1)                                                                                        \
    }
        switch (hc / 64) {
            case 2: MEL_ATT_LAUNCH(2) break;
            case 4: MEL_ATT_LAUNCH(4) break;
            case 8: MEL_ATT_LAUNCH(8) break;
            case 16: MEL_ATT_LAUNCH(16) break;
            default: return fail(MEL_ERR_UNSUPPORTED, "%s: heads*C = %d not in {128,256,512,1024}", what, hc);
        }
#undef MEL_ATT_LAUNCH
#undef MEL_ATT_LAUNCH_W
    }
    return check_launch(what);
}

}  // namespace mel
