// NOT COMPILED INTO THE LIBRARY - round 3's two other forms of the GATv2 attention launch (csrc/attention.hpp), both BIT-IDENTICAL to
// the shipped one-wave-per-target row kernel and both slower (L-DGN 50-node, 1024 envs, conv1 / conv2 attention, us;
// profiles/r03f_attention_forms.log):
//
//   gat_attend_rows_kernel (ships)                                                      22.6 / 12.6
//   the same with software-pipelined row loads (step s + 1's rows in flight under step s)  25.4 / 13.5   (104 VGPRs: 4 waves / SIMD)
//   gat_attend_env_kernel below: a workgroup per env, its source rows staged in LDS      29.3 / 16.3
//
// Why neither helps: the launch is bound by VECTOR ARITHMETIC, not by the L2 gather - per (target, source) pair and 8 channels
// a lane issues ~65 VALU instructions (add, 0.2 x, max, fma for the score; two fmas for the weighted sum; DPP head sums, exp);
// 10.7 K targets x ~10 sources x 65 = 7 M wave-instructions over 1 024 SIMDs at 4-5 cycles each is ~15 us of pure VALU issue
// for conv1.  Fewer bytes (env staging: ~6 x fewer from L2) or shorter dependent chains (pipelining) buy nothing, and both cost
// occupancy or balance (envs differ 4-8 x in target count: a workgroup per env waits for its stragglers).
//
// The pipelined form was a template flag of attend_target (fetch of step s + 1 before the arithmetic of step s, two row sets);
// the env kernel as it was measured:

// ---- ATT_ROWS / ATT_SINGLE with the env's source rows staged in LDS (round 3) -----------------------------------------------
// The row kernel above gathers every target's ~10 source rows (2 KB each) from L2 again - 10.7 K targets x ~10 rows = 214 MB per
// conv1 launch - although the targets of one env share a handful of distinct rows (|U2| ~ 16 per env, |U1| ~ 10).  Here a
// workgroup owns an ENV: its NW waves first copy the env's source rows (the rows of the set the sources are packed by: U2 for
// conv1, U1 for conv2; by tuple id from the node-feature table or by packed position) into LDS - one coalesced 2 KB row per wave
// and instruction, all in flight together - and then walk the env's targets, wave by wave, reading source rows from LDS (row of
// source j = rank of j in that set): ~6 x fewer bytes from L2 and an LDS round trip instead of an L2 round trip in the dependent
// chain of every softmax step.  Per row it is the arithmetic of attend_target in the same order: results are BIT-IDENTICAL to the
// row kernel (tests/test_gpu_forward.py).  GATv2, fp32 rows, heads * C <= 512; an env whose source set has more than ATT_ENV_SMAX
// rows takes the row kernel's path (global loads) inside this kernel.
constexpr int ATT_ENV_SMAX = 32;      // source rows an env can stage: 32 x 2 KB = 64 KB of LDS, two 8-wave workgroups per CU
constexpr int ATT_ENV_NW = 8;

template <int VPL, int W>
__device__ __forceinline__ Vec<VPL> attend_target_lds(const AttArgs& a, const Vec<VPL>& xr, NodeSet<W> sources, const NodeSet<W>& smask,
                                                      const float* __restrict__ rows, const Vec<VPL>& att, const Vec<VPL>& bias, int lane) {
    constexpr int HC = 64 * VPL, G = MEL_ATT_G;
    float m = -INFINITY, l = 0.f;
    Vec<VPL> acc;
#pragma unroll
    for (int i = 0; i < VPL; ++i) acc.v[i] = 0.f;
    while (ns_any(sources)) {
        int row[G];
        bool on[G];
#pragma unroll
        for (int k = 0; k < G; ++k) {
            on[k] = ns_any(sources);
            const int j = on[k] ? ns_lowest(sources) : 0;
            ns_clear_lowest(sources);
            row[k] = (on[k] ? ns_rank_below(smask, j) : 0) * HC + lane * VPL;          // off slots re-read row 0
        }
        Vec<VPL> xl[G];
#pragma unroll
        for (int k = 0; k < G; ++k) xl[k] = load_vec<VPL>(rows + row[k]);
        float sc_[G];
#pragma unroll
        for (int k = 0; k < G; ++k) {
            float t = 0.f;
            if constexpr (VPL % 2 == 0) {
                f32x2 t2 = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < VPL; i += 2) {
                    const f32x2 z = f32x2{xr.v[i], xr.v[i + 1]} + f32x2{xl[k].v[i], xl[k].v[i + 1]};
                    const f32x2 zs = z * 0.2f;
                    const f32x2 zm = {fmaxf(z.x, zs.x), fmaxf(z.y, zs.y)};
                    t2 = __builtin_elementwise_fma(f32x2{att.v[i], att.v[i + 1]}, zm, t2);
                }
                t = t2.x + t2.y;
            } else {
#pragma unroll
                for (int i = 0; i < VPL; ++i) {
                    const float z = xr.v[i] + xl[k].v[i];
                    t = fmaf(att.v[i], fmaxf(z, 0.2f * z), t);
                }
            }
            sc_[k] = t;
        }
#pragma unroll
        for (int k = 0; k < G; ++k) {
            sc_[k] = head_sum(sc_[k], a.lanes_per_head);
            if (!on[k]) sc_[k] = -INFINITY;
        }
        float mn = m;
#pragma unroll
        for (int k = 0; k < G; ++k) mn = fmaxf(mn, sc_[k]);
        const float rs = fast_exp(m - mn);
        float pe[G];
#pragma unroll
        for (int k = 0; k < G; ++k) pe[k] = fast_exp(sc_[k] - mn);
        float ps = 0.f;
#pragma unroll
        for (int k = 0; k < G; ++k) ps += pe[k];
        l = l * rs + ps;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            float t = acc.v[i] * rs;
#pragma unroll
            for (int k = 0; k < G; ++k) t = fmaf(pe[k], xl[k].v[i], t);
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

// off_t: [bs + 1] first target row of every env (off1 for conv1, offL for conv2)
template <int VPL, int MODE, int W>
__global__ __launch_bounds__(64 * ATT_ENV_NW, 4) void gat_attend_env_kernel(AttArgs a, const int32_t* __restrict__ off_t) {
    constexpr int HC = 64 * VPL, NW = ATT_ENV_NW;
    __shared__ __attribute__((aligned(16))) float rows[ATT_ENV_SMAX * HC];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    const int t0 = off_t[b], nt = off_t[b + 1] - t0;
    if (nt <= 0) return;                                   // (uniform over the workgroup)
    const Vec<VPL> att = load_vec_or_zero<VPL>(a.att, lane);
    const Vec<VPL> bias = load_vec_or_zero<VPL>(a.bias, lane);
    const TargetDesc<W>* desc = static_cast<const TargetDesc<W>*>(a.desc) + t0;
    const TargetDesc<W> d0 = desc[0];                       // every target of the env has the same source packing
    const NodeSet<W> smask = d0.smask;
    const int soff = d0.soff, ns = ns_count(smask);
    int my_fid[W];
    MEL_W_FOR(h) my_fid[h] = 0;
    const bool table = MODE == ATT_ROWS && a.fid != nullptr;
    if (table) MEL_W_FOR(h) my_fid[h] = lane + 64 * h < a.n ? a.fid[(size_t)b * a.n + lane + 64 * h] : 0;
    const bool staged = ns <= ATT_ENV_SMAX;
    if (staged) {
        // source row k (the k-th member of smask) -> LDS row k: wave w copies rows w, w + NW, ..
        NodeSet<W> rest = smask;
        for (int k = 0; k < ns; ++k) {
            const int j = ns_lowest(rest);
            ns_clear_lowest(rest);
            if ((k % NW) != wave) continue;
            const size_t src = (size_t)(table ? node_i32<W>(my_fid, j) : soff + k) * a.ld_l + lane * VPL;
            store_vec<VPL>(rows + k * HC + lane * VPL, load_vec<VPL>(a.xl + src));
        }
        __syncthreads();
    }
    for (int t = wave; t < nt; t += NW) {
        const int r = t0 + t;
        const TargetDesc<W> d = desc[t];
        size_t xr_row = (size_t)r;
        if (table) xr_row = (size_t)node_i32<W>(my_fid, d.node);
        Vec<VPL> o;
        if (staged) {
            const Vec<VPL> xr = load_vec<VPL>(a.xr + xr_row * a.ld_r + lane * VPL);
            o = attend_target_lds<VPL, W>(a, xr, d.sources, smask, rows, att, bias, lane);
        } else {
            o = attend_target<VPL, MEL_CONV_GATV2, false, W, MEL_ATT_G>(a, xr_row, d.sources, d.smask, d.soff, att, bias, lane, my_fid);
        }
        if constexpr (MODE == ATT_SINGLE) {
            store_vec<VPL>(a.xcat + (size_t)r * a.ld_cat + a.cat_off + lane * VPL, o);
        } else {
            if (a.out_scale) {
                const float dmv = a.out_scale[r];
                Vec<VPL> om;
#pragma unroll
                for (int i = 0; i < VPL; ++i) om.v[i] = o.v[i] * dmv;
                store_vec<VPL>(a.out + (size_t)r * a.ldo + lane * VPL, om);
            } else {
                store_vec<VPL>(a.out + (size_t)r * a.ldo + lane * VPL, o);
            }
            if (d.cat_row >= 0) {
                const size_t cat = (size_t)d.cat_row * a.ld_cat;
                store_vec<VPL>(a.xcat + cat + a.hidden + lane * VPL, o);                    // x_2, before the mask (l_dgn.py:127)
                const size_t h0 = (size_t)(table ? node_i32<W>(my_fid, d.node) : d.soff + ns_rank_below(d.smask, d.node)) * a.hidden;
                for (int c = lane; c < a.hidden; c += 64) a.xcat[cat + c] = a.h0[h0 + c];  // x_1 (l_dgn.py:122)
            }
        }
    }
}

