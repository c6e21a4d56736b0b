// NOT COMPILED INTO THE LIBRARY - round 3's experiments on the 128 x 128 split-bf16 GEMM (csrc/gemm_split.hpp), kept with the
// probes that measured them (tools/roles_probe.hip, tools/ldpath_probe.hip, tools/overlap_probe.hip; logs under profiles/r03c_*).
// All three kernels are CORRECT (bit-identical to gemm_split_big_kernel: same summation order) and none is faster:
//
//   conv2 of the L-DGN step (488 items), us      one-role 128 x 128 (ships)   57 - 60
//   gemm_split_wide_kernel<WN = 4>  128 x 256, 8 waves, one workgroup per CU, LDS-transposed 16-byte epilogue        59.6
//   gemm_split_wide_kernel<WN = 2>  128 x 128 with that epilogue, two workgroups per CU                               61.7
//   gemm_split_wide_kernel<2, DENSE> three workgroups per CU (96-byte swizzled LDS rows, 168 VGPRs, 104 B scratch)    79
//   gemm_split_roles_kernel          4 MFMA waves + 2 x 4 loader waves (teams on alternate steps), 4-stage ring      66
//   (its first form: 4 MFMA + 4 loader waves, 3-stage ring, hand-over epilogue in the loaders)                       72
//
// WHY (tools/overlap_probe.hip): with the matrix pipe of a CU saturated (24 MFMAs per 768 cycles per SIMD, register operands, no
// LDS) four loader waves on the same CU stream conv2's operand shape at 20 KB per ~1 200 cycles - 17 B / clk - against 562 cycles
// (36 B / clk) with the pipe idle; LDS-DMA instead of VGPR loads: 1 484 against 714.  The matrix work keeps its rate (769 cycles per
// step) either way.  A 128 x 128 split tile needs 20 KB per 768 cycles = 26 B / clk: the kernel is bound by the vector-memory path
// UNDER MFMA LOAD at ~1 200 cycles per tile step = 0.64 of the pipe (~40 us for conv2 at the 1.8 GHz the chip holds), whatever the
// wavefront structure.  What the stamps of the role-split kernels show is the same thing from the inside: MFMA waves at 860 cycles per
// step when fed, loader waves at 1 300 - 1 700 per step (waiting for loads issued four steps earlier), the split's VALU work
// (1 240 -> 680 cycles of a loader's step without MFMAs beside it) zero-sum against the MFMA stream of its SIMD (s_setprio moves the
// loss from one to the other).  The lever that is left is FEWER OPERAND BYTES PER MFMA: fp32 W split in the kernel (4 instead of 6
// bytes per element: 16 KB per 128 x 128 step, 12 KB per tile step at 128 x 256) in a structure that keeps the load path busy
// continuously - the 128 x 256 kernel below moves 20 % fewer bytes and is not faster because its single 8-wave workgroup per CU runs
// in lock-step.  Not built this round.
//
// To build the probes again: paste this file's kernels back into csrc/gemm_split.hpp in front of split_weights_kernel (it needs
// `#include "gemm_ring.hpp"` there for wait_lds_done) - tools/roles_probe.hip includes that header.

// ---- 128 x 256 tiles (round 3) -------------------------------------------------------------------------------------------
// What bounds the 128 x 128 kernel above is the CU's vector-memory path: 20 KB of operands per 768 cycles of matrix work is
// 26 B / clk, and four loader waves with NOTHING else on the CU stream exactly that shape at 33 B / clk (tools/ldpath_probe.hip:
// 626 cycles per step, whatever the access shape) - with two workgroups per CU the path is ~80 % busy at the pipe's rate and every
// wave's loads queue behind the others' (tools/roles_probe.hip's cycle stamps: loaders spend 1 000+ cycles per step however the
// work is dealt out between wavefronts).  So: fewer operand bytes per MFMA.  One 512-thread workgroup per CU owns 128 x 256
// outputs - 2 x 4 waves of 64 x 64, the same wave tile - and a 16-k step moves 8 KB of A + 24 KB of W planes for 2 x 768 cycles of
// matrix work per SIMD: 21 B / clk.  A is fetched from L2 twice instead of four times, and a thread splits ONE f32x4 per step
// instead of two (half the VALU work that competes with the MFMAs for issue slots).  Same flat stream of K steps with the register
// prefetch two steps ahead as gemm_split_big_kernel.
// Epilogue: no row scale (the launcher keeps such problems on the 128 x 128 kernel), bias from LDS (staged once per launch), and
// each wave turns its four 32 x 32 accumulator blocks into 16-byte row stores through a private 4 KB LDS transposition - 16
// store instructions per wave and tile, none behind a dependent load (the 128 x 128 kernel: 64 dword stores behind the scale
// loads, ~10 000 cycles per tile by its cycle stamps).
// WN = wave columns: 4 = the 128 x 256 tile (512 threads, one workgroup per CU); 2 = a 128 x 128 tile with the same epilogue
// (256 threads, 80 KB of LDS: two workgroups per CU) for launches whose N is no multiple of 256.
constexpr int GEMW_BN = 256;
constexpr int GEMW_BIAS_FLOATS = 2048;

// DENSE (WN = 2 only): THREE workgroups per CU - LDS rows without the pad chunk (96 bytes; chunk c of row r sits in slot
// c ^ ((r >> 3) & 1), which keeps the fragment reads conflict-free), 48 KB of stages and nothing else in LDS (the epilogue
// stores straight from the accumulators), at most 168 VGPRs.  A wave's K step is ~3 000 cycles of latencies around 768 cycles of
// matrix work (NOTES.md): a third wave per SIMD is a third more of them in flight.
template <int TAG = 0, int WN = 4, bool DENSE = false>
__global__ __launch_bounds__(128 * WN, DENSE ? 3 : 2) void gemm_split_wide_kernel(GemmBatch batch) {
    static_assert(!DENSE || WN == 2, "DENSE is the 128 x 128 form");
    constexpr int BM = 128, BN = 64 * WN, T = 128 * WN;
    constexpr int APT = 512 / T;                      // f32x4 pieces of A per thread and step
    constexpr int RC = DENSE ? 6 : GEMS2_ROW;         // 16-byte chunks per LDS row
    constexpr int BUF = (BM + BN) * RC;               // 16-byte chunks per LDS stage: 42 KB (WN = 2: 28 KB, DENSE 24 KB)
    constexpr int XP = DENSE ? 0 : 2 * WN * 256;      // transposition buffers: 4 KB per wave, in 16-byte chunks
    __shared__ u32x4 lds[2 * BUF + XP + (DENSE ? 0 : GEMW_BIAS_FLOATS / 4)];      // 84 KB + 32 KB + 8 KB (WN = 2: 56 + 16 + 8 = 80 KB)
    float* xpose = reinterpret_cast<float*>(lds + 2 * BUF);
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
                act[i] = ((rows[i] + BM - 1) / BM) * (q.N / BN) * (q.ksplit > 1 ? q.ksplit : 1);
                if constexpr (!DENSE)
                    for (int n = threadIdx.x; n < q.N; n += T)
                        bias_s[o + n] = (q.bias_hi && n >= q.split_n) ? q.bias_hi[n - q.split_n] : (q.bias ? q.bias[n] : 0.f);
                o += q.N;
            }
            pre[i + 1] = pre[i] + ((act[i] + 7) & ~7);
        }
    }
    const int total = pre[GEMM_MAX_GROUP];
    const int stride = gridDim.x;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int r = lane & 31, h = lane >> 5;
    const int crow = tid >> 2;            // A staging: 4 threads per 64-byte fp32 row slice, T / 4 rows per pass, APT passes
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
    struct Ctx {
        const float* a_src[APT];       // this thread's 4 floats of K step 0 of its A rows
        const u32x4* w_src[3];         // this thread's three 16-byte chunks of the W tile's K step 0
        int m0, n0, M, pi, KT, ks;
    };
    auto setup = [&](Ctx& c, int t) {
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
        for (int i = 0; i < APT; ++i) {
            const int row = min(c.m0 + crow + i * (T / 4), c.M - 1);              // clamped, never predicated
            const int ar = g.arow ? g.arow[row] : row;
            c.a_src[i] = g.A + (size_t)ar * g.lda + step0 * GEMS2_BK + kq * 4;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {       // [N][K / 16][3][16] planes: 6 chunks per row and step, 1 536 per tile and step
            const int ch = tid + i * T, wrow = ch / 6, wch = ch - wrow * 6;
            const int n = c.n0 + wrow;
            const uint16_t* base = (g.W_hi && n >= g.split_n)
                                       ? reinterpret_cast<const uint16_t*>(g.W_hi) + (size_t)(n - g.split_n) * 3 * g.K
                                       : reinterpret_cast<const uint16_t*>(g.W) + (size_t)n * 3 * g.K;
            c.w_src[i] = reinterpret_cast<const u32x4*>(base + (size_t)step0 * 48 + wch * 8);
        }
    };

    int t = next_valid(blockIdx.x);
    __syncthreads();                          // the biases are staged
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
    // DENSE: slot of chunk c in row r = c ^ ((r >> 3) & 1); T / 4, 32 and 64 rows are multiples of 16, so the bit is the same for
    // every row a thread / lane touches
    const int sw_a = DENSE ? (crow >> 3) & 1 : 0, sw_r = DENSE ? (r >> 3) & 1 : 0;
    const int a_st = crow * (2 * RC) + (kq ^ (sw_a << 1));                 // + i * (T / 4) rows, + plane * 4
    int w_st[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int ch = tid + i * T, wrow = ch / 6;
        w_st[i] = (BM + wrow) * RC + ((ch - wrow * 6) ^ (DENSE ? (wrow >> 3) & 1 : 0));
    }
    const int a_off = (wm * 64 + r) * RC + (h ^ sw_r);                     // + i * 32 rows, + plane * 2
    const int w_off = (BM + wn * 64 + r) * RC + (h ^ sw_r);

    struct Regs {
        f32x4 a[APT];
        u32x4 w[3];
    };
    struct Meta {
        int m0, n0, M, pi, KT, ks;
    };
    Ctx pf;                                   // where the prefetch stands
    int pf_t = t, pf_kt = 0;
    bool pf_valid = true;
    setup(pf, t);
    Meta cm{pf.m0, pf.n0, pf.M, pf.pi, pf.KT, pf.ks}, nm{};
    bool nm_valid = false;

    // loads of the next step of the stream, unconditional (see gemm_split_kernel)
    auto issue_a = [&](Regs& R) {
#pragma unroll
        for (int i = 0; i < APT; ++i) R.a[i] = *reinterpret_cast<const f32x4*>(pf.a_src[i] + pf_kt * GEMS2_BK);
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
    auto fill_a = [&](int stage, const Regs& R) {
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            u32x2 hi, mid, lo;
            split4(R.a[i], hi, mid, lo);
            u32x2* dst = lds8 + stage * (2 * BUF) + a_st + i * (T / 4) * (2 * RC);
            dst[0] = hi, dst[4] = mid, dst[8] = lo;
        }
    };
    auto fill_w = [&](int stage, const Regs& R) {
#pragma unroll
        for (int i = 0; i < 3; ++i) lds[stage * BUF + w_st[i]] = R.w[i];
    };

    Regs R0, R1;
    issue(R0);                                 // step 0
    issue(R1);                                 // step 1
    fill_a(0, R0), fill_w(0, R0);
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

    // epilogue of one finished work item: each 32 x 32 block through this wave's private 4 KB of LDS ([32 rows][32 floats],
    // written one register = two 128-byte row pieces at a time, read back as 16-byte row chunks: both conflict-free unpadded)
    float* xp = xpose + wid * 1024;
    auto write_out = [&](const Meta& m) {
        const GemmArgs& g = batch.p[m.pi];
        const bool raw = g.ksplit > 1;
        if constexpr (DENSE) {                 // straight from the accumulators: one dword per lane and store, bias from memory
            float* __restrict__ Yd = raw ? g.Y + (size_t)m.ks * g.part_stride : g.Y;
            float bj[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = m.n0 + wn * 64 + j * 32 + r;
                bj[j] = raw ? 0.f : (g.bias_hi && n >= g.split_n) ? g.bias_hi[n - g.split_n] : (g.bias ? g.bias[n] : 0.f);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float* col = Yd + m.n0 + wn * 64 + j * 32 + r;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int mrow = m.m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                        float v = acc[i][j][e] + bj[j];
                        if (!raw && g.relu) v = fmaxf(v, 0.f);
                        if (mrow < m.M) col[(size_t)mrow * g.ldy] = v;
                    }
                }
            return;
        }
        float* __restrict__ Y = raw ? g.Y + (size_t)m.ks * g.part_stride : g.Y;
        const int relu = raw ? 0 : g.relu, ldy = g.ldy;
        int bo = 0;
#pragma unroll
        for (int k = 1; k < GEMM_MAX_GROUP; ++k)
            if (m.pi >= k) bo = boff[k];
        const int rr = lane >> 3, c4 = (lane & 7) * 4;          // read side: row rr + 8 q, floats c4 .. c4 + 3
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = m.n0 + wn * 64 + j * 32 + c4;
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (!raw) b4 = *reinterpret_cast<const f32x4*>(bias_s + bo + n);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) xp[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[i][j][e];
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

    // one K step: MFMAs on `stage`, Ra (step s+1) -> the other stage and, as soon as its registers are free, the loads of
    // step s+3, dealt out between the six groups of four MFMAs (as in gemm_split_big_kernel)
    auto step = [&](Regs& Ra) {
        const u32x4* cst = lds + stage * BUF;
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[u][p] = __builtin_bit_cast(bf16x8, cst[a_off + u * 32 * RC + 2 * p]);
                b[u][p] = __builtin_bit_cast(bf16x8, cst[w_off + u * 32 * RC + 2 * p]);
            }
        constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[k]], b[j][PB[k]], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (k == 0) fill_a(stage ^ 1, Ra);
            if (k == 1) fill_w(stage ^ 1, Ra);
            if (k == 2) issue_a(Ra);
            if (k == 3) issue_w(Ra);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        stage ^= 1;
        advance();
        if (++ckt == cm.KT) {                  // the work item is complete
            write_out(cm);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            cm = nm, ckt = 0;
            if (!nm_valid) cm.KT = 1 << 30;    // (the padding step of an odd stream ends no work item)
            nm_valid = false;
        }
    };
    for (int it = 0; it < (nsteps + 1) >> 1; ++it) {       // counted loop over pairs of steps (see gemm_split_kernel)
        step(R1);
        step(R0);
    }
}

// ---- 128 x 128 tiles, SPECIALISED wavefronts with TWO loader teams (round 3) -----------------------------------------------
// One 768-thread workgroup per CU:
//   waves 0-3    MFMA waves  2 x 2, a 64 x 64 block set each: 24 v_mfma_f32_32x32x16_bf16 per 16-k step on fragments they read from
//                            LDS ONE STEP EARLIER (two fragment register sets) and nothing else in their instruction stream - measured
//                            860 cycles per step for 768 of matrix work when the operands keep coming (tools/roles_probe.hip).  A
//                            finished tile goes out through the wave's private 4 KB LDS transposition as 16-byte row stores.
//   waves 4-7    loader team 0: the even steps of the workgroup's stream      } global loads four steps ahead (two register sets per
//   waves 8-11   loader team 1: the odd steps                                 } team), fp32 -> 3 x bf16 split, LDS fill
// Why two teams: ONE loader wave per SIMD needs 1 300 - 1 700 cycles for the chain of a step (wait for the prefetch, ~45 VALU
// instructions that compete with the MFMA wave of their SIMD for issue slots, 9 LDS writes, 5 loads through an address path that
// is 80 % busy, LDS wait, barrier) - measured with the role split at one loader wave per SIMD: MFMA waves waiting at the step
// barrier for half of the kernel (142 TF at M = 65 536 against 178 for the one-role kernel).  Two teams have two steps each.
// Four-stage LDS ring.  Barriers B(-2) .. B(n-1), one per step, all twelve waves: B(g) ends step g - every MFMA wave has issued
// the MFMAs of step g and holds the fragments of step g + 1.  A team's iteration for step s: fill(s) - its stage held step s - 4,
// read during step s - 5 - | B(s-3) | loads of step s + 4, LDS writes landed | B(s-2).
#ifdef MEL_ROLES_PROF
// tuning builds (-DMEL_ROLES_PROF, tools/roles_probe.hip): cycles of MFMA wave 0 in [0] fragment reads + MFMAs, [1] epilogue,
// [2] LDS wait + step barrier; of loader wave 0 of team 0 in [3] prefetch wait + split + fill, [4] first barrier, [5] issuing the
// loads (+ next work item), [6] LDS wait + second barrier; [7] kernel (MFMA wave 0), [8] workgroups, [9] steps
__device__ unsigned long long g_roles_prof[16];
#define ROLES_T() __builtin_readcyclecounter()
#define ROLES_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
constexpr int GEMR_STAGES = 4;
constexpr int GEMR_BIAS_FLOATS = 2048;                // the launch's bias vectors, problem after problem (sum of N <= 2 048)

template <int TAG = 0>
__global__ __launch_bounds__(768, 3) void gemm_split_roles_kernel(GemmBatch batch) {
    constexpr int BM = 128, BN = 128;
    constexpr int BUF = (BM + BN) * GEMS2_ROW;        // 16-byte chunks per LDS stage
    constexpr int XP = 4 * 512;                       // the MFMA waves' transposition buffers: 8 KB each (two 32 x 32 blocks)
    __shared__ u32x4 lds[GEMR_STAGES * BUF + XP + GEMR_BIAS_FLOATS / 4];       // 112 KB ring + 32 KB + 8 KB (one LDS object)
    float* xpose = reinterpret_cast<float*>(lds + GEMR_STAGES * BUF);
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
                act[i] = ((rows[i] + BM - 1) / BM) * (q.N / BN) * (q.ksplit > 1 ? q.ksplit : 1);
                for (int n = threadIdx.x; n < q.N; n += 768)
                    bias_s[o + n] = (q.bias_hi && n >= q.split_n) ? q.bias_hi[n - q.split_n] : (q.bias ? q.bias[n] : 0.f);
                o += q.N;
            }
            pre[i + 1] = pre[i] + ((act[i] + 7) & ~7);
        }
    }
    const int total = pre[GEMM_MAX_GROUP];
    const int stride = gridDim.x;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);    // 0: MFMA waves, 1 / 2: loader teams 0 / 1
    const int tid = threadIdx.x & 255;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

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
        int m0, n0, M, pi, KT, ks;
    };
    auto meta_of = [&](int t) {           // the work item behind list position t (scalar arithmetic only)
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
        Meta m;
        m.pi = pi, m.M = rows[pi], m.KT = g.K / GEMS2_BK / S, m.ks = (wg / nbn) % S;
        m.m0 = (wg / (nbn * S)) * BM, m.n0 = (wg % nbn) * BN;
        return m;
    };

    const int t0 = next_valid(blockIdx.x);
    __syncthreads();                          // the biases are staged
    if (t0 >= total) return;
    int nsteps = 0;                           // K steps of this workgroup's whole stream
    for (int tt = t0; tt < total; tt = next_valid(tt + stride)) nsteps += meta_of(tt).KT;
    const int npad = (nsteps + 3) & ~3;       // every wave executes the barriers B(-2) .. B(npad - 1)

    if (role == 0) {
        // ---- MFMA waves ------------------------------------------------------------------------------------------------
        const int wm = wid >> 1, wn = wid & 1;
        const int r = lane & 31, h = lane >> 5;
        const int a_off = (wm * 64 + r) * GEMS2_ROW + h;                       // + i * 32 rows, + plane * 2
        const int w_off = (BM + wn * 64 + r) * GEMS2_ROW + h;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        struct Frags {
            bf16x8 a[2][3], b[2][3];
        };
        auto read_frags = [&](int g, Frags& f) {
            const u32x4* cst = lds + (g & (GEMR_STAGES - 1)) * BUF;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    f.a[u][p] = __builtin_bit_cast(bf16x8, cst[a_off + u * 32 * GEMS2_ROW + 2 * p]);
                    f.b[u][p] = __builtin_bit_cast(bf16x8, cst[w_off + u * 32 * GEMS2_ROW + 2 * p]);
                }
        };
        // epilogue of one finished work item: each 32 x 32 block through this wave's private 4 KB of LDS ([32 rows][32 floats],
        // written one register = two 128-byte row pieces at a time, read back as 16-byte row chunks: both conflict-free unpadded)
        float* xp = xpose + wid * 2048;
        auto write_out = [&](const Meta& m) {
            const GemmArgs& g = batch.p[m.pi];
            const bool raw = g.ksplit > 1;
            float* __restrict__ Y = raw ? g.Y + (size_t)m.ks * g.part_stride : g.Y;
            const int relu = raw ? 0 : g.relu, ldy = g.ldy;
            int bo = 0;
#pragma unroll
            for (int k = 1; k < GEMM_MAX_GROUP; ++k)
                if (m.pi >= k) bo = boff[k];
            const int rr = lane >> 3, c4 = (lane & 7) * 4;          // read side: row rr + 8 q, floats c4 .. c4 + 3
            // two blocks (the two row halves i of a column half j) per LDS round trip
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = m.n0 + wn * 64 + j * 32 + c4;
                f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
                if (!raw) b4 = *reinterpret_cast<const f32x4*>(bias_s + bo + n);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        xp[i * 1024 + ((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[i][j][e];
                        acc[i][j][e] = 0.f;
                    }
                f32x4 v[2][4];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[i][q] = *reinterpret_cast<const f32x4*>(xp + i * 1024 + (rr + 8 * q) * 32 + c4);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int mrow = m.m0 + wm * 64 + i * 32 + rr + 8 * q;
                        const f32x4 a = v[i][q];
                        f32x4 o = {a[0] + b4[0], a[1] + b4[1], a[2] + b4[2], a[3] + b4[3]};
                        if (relu) o = f32x4{fmaxf(o[0], 0.f), fmaxf(o[1], 0.f), fmaxf(o[2], 0.f), fmaxf(o[3], 0.f)};
                        if (mrow < m.M) *reinterpret_cast<f32x4*>(Y + (size_t)mrow * ldy + n) = o;
                    }
            }
        };
        int t = t0, kt = 0;
        Meta cm = meta_of(t0);
        Frags F0, F1;
#ifdef MEL_ROLES_PROF
        unsigned long long pc0 = 0, pc1 = 0, pc2 = 0;
        const unsigned long long pk0 = ROLES_T();
#endif
        __builtin_amdgcn_s_barrier();         // B(-2): stage 0 holds step 0
        read_frags(0, F0);
        wait_lds_done();
        __builtin_amdgcn_s_barrier();         // B(-1): stage 1 holds step 1
        // one step: the fragments of step g + 1 are read while the MFMAs of step g (fragments `cur`) run
        auto step = [&](int g, const Frags& cur, Frags& nxt) {
#ifdef MEL_ROLES_PROF
            ROLES_FENCE();
            const unsigned long long q0 = ROLES_T();
            ROLES_FENCE();
#endif
            read_frags(g + 1, nxt);           // (past the end: a stage of padding, never multiplied)
            // per block smallest products first (mid*mid, hi*lo, lo*hi, hi*mid, mid*hi, hi*hi), the four blocks interleaved
            constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
            for (int k = 0; k < 6; ++k)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.a[i][PA[k]], cur.b[j][PB[k]], acc[i][j], 0, 0, 0);
            // the twelve fragment reads of the NEXT step go out two per group of four MFMAs, from the first group on (left alone
            // the scheduler sinks them below the last MFMA that reads the registers they reuse, and the next step starts by
            // waiting for them)
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // 2 x ds_read
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);      // 4 x MFMA
            }
#ifdef MEL_ROLES_PROF
            ROLES_FENCE();
            asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[1][1][0]));      // the MFMA chains have retired
            const unsigned long long q1 = ROLES_T();
            ROLES_FENCE();
#endif
            if (++kt == cm.KT) {              // the work item is complete
                wait_lds_done();              // (the next step's fragments first: the transposition reuses the wait counter)
                write_out(cm);
                t = next_valid(t + stride), kt = 0;
                if (t < total) cm = meta_of(t);
                else cm.KT = 1 << 30;
            }
#ifdef MEL_ROLES_PROF
            ROLES_FENCE();
            const unsigned long long q2 = ROLES_T();
            ROLES_FENCE();
#endif
            wait_lds_done();                  // the next step's fragments are in registers
            __builtin_amdgcn_s_barrier();     // B(g)
#ifdef MEL_ROLES_PROF
            ROLES_FENCE();
            const unsigned long long q3 = ROLES_T();
            pc0 += q1 - q0, pc1 += q2 - q1, pc2 += q3 - q2;
#endif
        };
        int g = 0;
        for (; g + 1 < nsteps; g += 2) {
            step(g, F0, F1);
            step(g + 1, F1, F0);
        }
        if (g < nsteps) step(g, F0, F1);
#ifdef MEL_ROLES_PROF
        if (wid == 0 && lane == 0) {
            atomicAdd(&g_roles_prof[0], pc0), atomicAdd(&g_roles_prof[1], pc1), atomicAdd(&g_roles_prof[2], pc2);
            atomicAdd(&g_roles_prof[7], ROLES_T() - pk0), atomicAdd(&g_roles_prof[8], 1ull), atomicAdd(&g_roles_prof[9], (unsigned long long)nsteps);
        }
#endif
        for (int pad = npad - nsteps; pad > 0; --pad) __builtin_amdgcn_s_barrier();      // the loaders' padding steps
        return;
    }

    // ---- loader teams: the staging of gemm_split_big_kernel, thread for thread, every other step ---------------------------
    const int team = role - 1;            // steps team, team + 2, ...
#ifndef MEL_ROLES_LOADER_PRIO
#define MEL_ROLES_LOADER_PRIO 3
#endif
    __builtin_amdgcn_s_setprio(MEL_ROLES_LOADER_PRIO);    // the loaders' few vector instructions go ahead of the MFMA stream of their SIMD
    const int crow = tid >> 2;            // A staging: 4 threads per 64-byte fp32 row slice, 64 rows per pass, 2 passes
    const int kq = tid & 3;               // this thread's 4 consecutive k of the step
    struct Ctx {
        const float* a_src[2];
        const u32x4* w_src[3];
        int KT;
    };
    auto setup = [&](Ctx& c, int t) {
        const Meta m = meta_of(t);
        const GemmArgs& g = batch.p[m.pi];
        c.KT = m.KT;
        const int step0 = m.ks * m.KT;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = min(m.m0 + crow + i * 64, m.M - 1);                    // clamped, never predicated
            const int ar = g.arow ? g.arow[row] : row;
            c.a_src[i] = g.A + (size_t)ar * g.lda + step0 * GEMS2_BK + kq * 4;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {       // [N][K / 16][3][16] planes: 6 chunks per row and step, 768 per tile and step
            const int ch = tid + i * 256, wrow = ch / 6, wch = ch - wrow * 6;
            const int n = m.n0 + wrow;
            const uint16_t* base = (g.W_hi && n >= g.split_n)
                                       ? reinterpret_cast<const uint16_t*>(g.W_hi) + (size_t)(n - g.split_n) * 3 * g.K
                                       : reinterpret_cast<const uint16_t*>(g.W) + (size_t)n * 3 * g.K;
            c.w_src[i] = reinterpret_cast<const u32x4*>(base + (size_t)step0 * 48 + wch * 8);
        }
    };
    u32x2* lds8 = reinterpret_cast<u32x2*>(lds);
    const int a_st = crow * (2 * GEMS2_ROW) + kq;                          // + i * 64 rows, + plane * 4
    int w_st[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int ch = tid + i * 256, wrow = ch / 6;
        w_st[i] = (BM + wrow) * GEMS2_ROW + (ch - wrow * 6);
    }
    struct Regs {
        f32x4 a[2];
        u32x4 w[3];
    };
    Ctx pf;                                   // where this team's prefetch stands: step pf_kt of work item pf_t
    int pf_t = t0, pf_kt = team;
    bool pf_valid = true;
    setup(pf, t0);
    auto normalise = [&]() {                  // pf_kt may have run past its work item: cross into the next one(s)
        while (pf_valid && pf_kt >= pf.KT) {
            const int tn = next_valid(pf_t + stride);
            if (tn < total) {
                pf_kt -= pf.KT;
                setup(pf, tn);
                pf_t = tn;
            } else {
                pf_valid = false, pf_kt = pf.KT - 1;      // the stream is over: the last step again (into stages nobody reads)
            }
        }
    };
    normalise();
    auto issue = [&](Regs& R) {               // loads of this team's next step, unconditional (see gemm_split_kernel)
#pragma unroll
        for (int i = 0; i < 2; ++i) R.a[i] = *reinterpret_cast<const f32x4*>(pf.a_src[i] + pf_kt * GEMS2_BK);
#pragma unroll
        for (int i = 0; i < 3; ++i) R.w[i] = pf.w_src[i][pf_kt * 6];
        if (pf_valid) {
            pf_kt += 2;
            normalise();
        }
    };
    // the fill of a step in two halves, one on each side of the iteration's first barrier (so that the two teams' fills overlap
    // instead of taking turns between consecutive barriers): A rows 0-63 + W, then A rows 64-127
    auto fill_a = [&](int s, const Regs& R, int i) {
        const int stage = s & (GEMR_STAGES - 1);
        u32x2 hi, mid, lo;
        split4(R.a[i], hi, mid, lo);
        u32x2* dst = lds8 + stage * (2 * BUF) + a_st + i * 64 * (2 * GEMS2_ROW);
        dst[0] = hi, dst[4] = mid, dst[8] = lo;
    };
    auto fill_w = [&](int s, const Regs& R) {
        const int stage = s & (GEMR_STAGES - 1);
#pragma unroll
        for (int i = 0; i < 3; ++i) lds[stage * BUF + w_st[i]] = R.w[i];
    };
#ifdef MEL_ROLES_PROF
    unsigned long long pl0 = 0, pl1 = 0, pl2 = 0, pl3 = 0;
#endif
    Regs R0, R1;
    issue(R0);                                 // step team
    issue(R1);                                 // step team + 2
    int s = team;
    // one iteration: this team's step s.  first: team 0's step 0 has no barrier between its fill and the end of the iteration
    auto iter = [&](Regs& Ra, bool first) {
#ifdef MEL_ROLES_PROF
        ROLES_FENCE();
        const unsigned long long q0 = ROLES_T();
        ROLES_FENCE();
#endif
        fill_a(s, Ra, 0);
        fill_w(s, Ra);
#ifdef MEL_ROLES_PROF
        ROLES_FENCE();
        const unsigned long long q1 = ROLES_T();
        ROLES_FENCE();
#endif
        if (!first) __builtin_amdgcn_s_barrier();      // B(s - 3)
#ifdef MEL_ROLES_PROF
        ROLES_FENCE();
        const unsigned long long q2 = ROLES_T();
        ROLES_FENCE();
#endif
        fill_a(s, Ra, 1);
        issue(Ra);                             // step s + 4
#ifdef MEL_ROLES_PROF
        ROLES_FENCE();
        const unsigned long long q3 = ROLES_T();
        ROLES_FENCE();
#endif
        wait_lds_done();
        __builtin_amdgcn_s_barrier();          // B(s - 2)
#ifdef MEL_ROLES_PROF
        ROLES_FENCE();
        const unsigned long long q4 = ROLES_T();
        if (s < nsteps) pl0 += q1 - q0, pl1 += q2 - q1, pl2 += q3 - q2, pl3 += q4 - q3;
#endif
        s += 2;
    };
    if (team == 0) {
        // steps 0 (ends at B(-2)), then 2, 4, .. npad (ends at B(npad - 2)), then B(npad - 1)
        iter(R0, true);
        for (int it = 0; it < npad / 4; ++it) {
            iter(R1, false);
            iter(R0, false);
        }
        __builtin_amdgcn_s_barrier();          // B(npad - 1)
    } else {
        // steps 1 (B(-2), B(-1)), 3, .. npad + 1 (B(npad - 2), B(npad - 1))
        for (int it = 0; it < npad / 4; ++it) {
            iter(R0, false);
            iter(R1, false);
        }
        iter(R0, false);
    }
#ifdef MEL_ROLES_PROF
    if (team == 0 && wid == 0 && lane == 0) {
        atomicAdd(&g_roles_prof[3], pl0), atomicAdd(&g_roles_prof[4], pl1), atomicAdd(&g_roles_prof[5], pl2);
        atomicAdd(&g_roles_prof[6], pl3);
    }
#endif
}

