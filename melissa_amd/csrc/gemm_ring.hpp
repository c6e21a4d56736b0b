// fp32-MFMA row GEMM with SPECIALISED wavefronts (PLAIN problems, 64 x 64 tiles, persistent):
//
//   waves 0-3  consumers  2 x 2 wavefronts of 32 x 32: v_mfma_f32_32x32x2_f32 on the fragments read (ds_read_b128) one
//                         step earlier, while the reads of the next step are in flight; a finished accumulator tile is
//                         dropped into an LDS hand-over buffer (16 ds_write_b32) and the wave goes on
//   waves 4-7  loaders    global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write), running
//                         FOUR K steps ahead of the arithmetic through a four-stage LDS ring; they also run the
//                         epilogue (scale, bias, ReLU, 16-byte row stores) of the tile the consumers just finished
//
// Why: in the one-role kernels (gemm_f32.hpp) the workgroups that share a CU move through load -> LDS -> MFMA phases
// in step, so the data-movement skeleton and the MFMA chain of a launch add up instead of overlapping (NOTES.md,
// "GEMM status": conv2 38 us + 51.5 us ~ the measured 87).  Here the two never wait for each other inside a K step:
// a consumer wave's step is 8 fragment reads + 16 MFMAs and nothing else, a loader wave's step is "issue the DMAs of
// step g+4, make sure step g+2 has landed", and one s_barrier per step hands a stage over in each direction.
// Long-K skinny problems can be cut along K (GemmArgs::ksplit): a work item is then (tile, K chunk) and writes raw
// partial products to its chunk's plane; splitk_finish_kernel (or the heads' fused finish) sums the planes in order.
//
// LDS stage = 64 A rows + 64 W rows x 128 B, unpadded (a DMA instruction writes 1 KiB linearly); bank conflicts of the
// fragment reads are removed by XOR-ing the 16-byte chunk index with (row >> 1) & 7 on the DMA's SOURCE side and on
// the read side.  4 stages x 16 KiB = 64 KiB per workgroup -> two workgroups (16 wavefronts) per CU.
#pragma once
#include "gemm_f32.hpp"

namespace mel {

// 64 x 64 tiles, four-stage ring; the loaders run FOUR K steps ahead: a consumer wave holds the fragments of the step it
// multiplies in registers (read from LDS one step earlier, while the previous step's MFMAs ran), so a stage is free again as
// soon as the step BEFORE the one it feeds has been multiplied.  Barrier B(g) therefore means: every consumer has the
// fragments of step g + 1 in registers and has issued the MFMAs of step g; the loaders guarantee before B(g) that step
// g + 2 has landed.
// BK = 32: 16 KiB stages, 80 KiB of LDS per workgroup, two workgroups per CU.  BK = 16: 8 KiB stages (rows of 64 B,
// sixteen rows per 1 KiB DMA piece), 48 KiB per workgroup, THREE workgroups per CU - a third unsynchronised consumer wave
// per SIMD to fill the matrix pipe while the other two sit at their step barriers or hand a tile over - at the price of a
// barrier every 8 instead of 16 MFMAs.
// WMC = 4: 128 x 64 tiles, eight consumer waves (4 x 2 of 32 x 32) + four loaders = 768 threads; with BK = 16 a stage is
// 12 KiB and the hand-over buffer 32 KiB: 80 KiB, two workgroups per CU = four consumer waves per SIMD and a quarter
// less operand traffic per FLOP than the 64 x 64 tile.
template <int BK, int WMC = 2>
struct RingCfg {
    static constexpr int BM = 32 * WMC, BN = 64;
    static constexpr int CONSUMERS = 2 * WMC, THREADS = 64 * (CONSUMERS + 4);
    static constexpr int STAGES = 4;
    static constexpr int AHEAD = STAGES;                           // K steps the loaders run ahead
    static constexpr int STAGE_FLOATS = (BM + BN) * BK;
    static constexpr int CPR = BK / 4;                             // 16-byte chunks per row
    static constexpr int RP = 256 / BK;                            // rows per 1 KiB DMA piece
    static constexpr int SWZ_SHIFT = BK == 32 ? 1 : 2;             // chunk c of row r sits in slot c ^ ((r >> SHIFT) & (CPR - 1))
    static constexpr int PPW = (BM + BN) / RP / 4;                 // DMA pieces per loader wave per stage
    static constexpr int WG_PER_CU = (BK == 32 || WMC == 4) ? 2 : 3;
};
constexpr int RING_OUT_STRIDE = 64;                // floats per row of the accumulator hand-over buffer (unpadded: 64 KiB
                                                   // ring + 16 KiB = exactly half of the CU's LDS); 16-byte chunk c of row
                                                   // r sits at chunk c ^ ((r & 3) << 1) so that the loaders' four-row reads
                                                   // spread over all banks
__device__ __forceinline__ int ring_out_index(int row, int col) {
    return row * RING_OUT_STRIDE + ((((col >> 2) ^ ((row & 3) << 1))) << 2) + (col & 3);
}
// hipcc also parses kernel bodies in its host pass; the LDS address-space cast and the counted waits only exist for the
// device pass, so they live behind the pass macro (the host pass needs just the launch stub).
__device__ __forceinline__ void dma_16B_to_lds(const float* src, float* lds_dst_uniform) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)lds_dst_uniform, 16, 0, 0);
#endif
}
template <int N>
__device__ __forceinline__ void wait_vmcnt_le() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
__device__ __forceinline__ void wait_lds_done() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

#ifdef MEL_RING_PROF
// Tuning builds (-DMEL_RING_PROF=<TAG>): cycles a consumer wave 0 of the launches tagged TAG spends in [0] MFMA sections, [1] step barriers, [2] epilogues, [3] whole kernel,
// [4] workgroups counted; loader wave 4: [5] issue (+ epilogue of a finished tile), [6] wait_landed, [7] barrier
__device__ unsigned long long g_ring_prof[8];
#define RING_T() __builtin_readcyclecounter()
#endif

template <int TAG = 0, int BK = GEMM_BK, int WMC = 2>
__global__ __launch_bounds__((RingCfg<BK, WMC>::THREADS), (RingCfg<BK, WMC>::WG_PER_CU)) void gemm_f32_ring_kernel(GemmBatch batch) {
    using Cfg = RingCfg<BK, WMC>;
    constexpr int NC = Cfg::CONSUMERS;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, CPR = Cfg::CPR, RP = Cfg::RP, SWZ = Cfg::SWZ_SHIFT;
    constexpr int RING_STAGES = Cfg::STAGES, RING_AHEAD = Cfg::AHEAD, RING_STAGE_FLOATS = Cfg::STAGE_FLOATS, RING_PPW = Cfg::PPW;
    __shared__ __attribute__((aligned(16))) float lds[RING_STAGES * RING_STAGE_FLOATS];
    // finished accumulator tile, handed from the consumers to the loaders
    __shared__ __attribute__((aligned(16))) float outbuf[BM * RING_OUT_STRIDE];

    // tile bookkeeping (wave-uniform), identical for every wave of the workgroup
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
    struct Tile {
        int pi, m0, n0, M, KT, ks, k0;           // ks / k0: split-K chunk of this work item and its first K column
    };
    auto tile_of = [&](int t) {
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
        // split-K: the work items of one row panel are ordered chunk-major, so the nbn items that share an A chunk are
        // neighbours
        const int S = g.ksplit > 1 ? g.ksplit : 1;
        const int ks = (wg / nbn) % S, KT = g.K / BK / S;
        return Tile{pi, (wg / (nbn * S)) * BM, (wg % nbn) * BN, rows[pi], KT, ks, ks * KT * BK};
    };

    const int first = next_valid(blockIdx.x);
    if (first >= total) return;
    // total K steps of this workgroup's stream: every wave executes exactly that many step barriers
    int steps_total = 0;
    for (int t = first; t < total; t = next_valid(t + stride)) steps_total += tile_of(t).KT;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

    if (wid >= NC) {
        // ------------------------------------------------------------------ loaders
        const int lw = wid - NC;
        const float* src[RING_PPW];            // this lane's 16-byte chunk of each piece, K step 0 of the loader's tile
        int lt = first, lkt = 0, lKT = 0;
        auto load_tile = [&](int t) {
            const Tile c = tile_of(t);
            const GemmArgs& g = batch.p[c.pi];
            lKT = c.KT;
#pragma unroll
            for (int i = 0; i < RING_PPW; ++i) {
                const int row = (lw * RING_PPW + i) * RP + lane / CPR;            // tile-local: A rows then W rows
                const int chunk = (lane % CPR) ^ ((row >> SWZ) & (CPR - 1));      // source chunk of this LDS slot
                if (row < BM) {
                    const int gr = min(c.m0 + row, c.M - 1);                      // clamped, never predicated
                    const int ar = g.arow ? g.arow[gr] : gr;
                    src[i] = g.A + (size_t)ar * g.lda + c.k0 + chunk * 4;
                } else {
                    const int n = c.n0 + (row - BM);
                    const float* base = (g.W_hi && n >= g.split_n) ? g.W_hi + (size_t)(n - g.split_n) * g.K
                                                                    : g.W + (size_t)n * g.K;
                    src[i] = base + c.k0 + chunk * 4;
                }
            }
        };
        load_tile(first);
        int issued = 0;
        // the tile the CONSUMERS are working on (the loaders write its result out once it is complete)
        int ct = first;
        Tile cc = tile_of(first);
        int ckt = 0;
        constexpr int WR = BM / 16;            // row groups of four per loader wave (BM / 4 rows each)
        auto write_out = [&]() {               // epilogue of tile cc from the hand-over buffer: BM / 4 rows per loader wave
            const GemmArgs& g = batch.p[cc.pi];
            float* __restrict__ Y = g.Y;
            const float* __restrict__ rs = g.rscale;
            const int ldy = g.ldy, relu = g.relu;
            const int col = (lane & 15) * 4, n = cc.n0 + col;
            if (g.ksplit > 1) {                // split-K: raw partial products into this chunk's plane
                float* __restrict__ P = Y + (size_t)cc.ks * g.part_stride;
#pragma unroll
                for (int i = 0; i < WR; ++i) {
                    const int row = lw * (BM / 4) + i * 4 + (lane >> 4), m = cc.m0 + row;
                    const f32x4 a = *reinterpret_cast<const f32x4*>(outbuf + ring_out_index(row, col));
                    if (m < cc.M) *reinterpret_cast<f32x4*>(P + (size_t)m * ldy + n) = a;
                }
                return;
            }
            f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
            if (g.bias_hi && n >= g.split_n) bias4 = *reinterpret_cast<const f32x4*>(g.bias_hi + (n - g.split_n));
            else if (g.bias) bias4 = *reinterpret_cast<const f32x4*>(g.bias + n);
            float sc[WR];
#pragma unroll
            for (int i = 0; i < WR; ++i) sc[i] = rs ? rs[min(cc.m0 + lw * (BM / 4) + i * 4 + (lane >> 4), cc.M - 1)] : 1.f;
#pragma unroll
            for (int i = 0; i < WR; ++i) {
                const int row = lw * (BM / 4) + i * 4 + (lane >> 4);
                const f32x4 a = *reinterpret_cast<const f32x4*>(outbuf + ring_out_index(row, col));
                f32x4 o = {a[0] * sc[i] + bias4[0], a[1] * sc[i] + bias4[1], a[2] * sc[i] + bias4[2], a[3] * sc[i] + bias4[3]};
                if (relu) o = f32x4{fmaxf(o[0], 0.f), fmaxf(o[1], 0.f), fmaxf(o[2], 0.f), fmaxf(o[3], 0.f)};
                const int m = cc.m0 + row;
                if (m < cc.M) *reinterpret_cast<f32x4*>(Y + (size_t)m * ldy + n) = o;
            }
        };
        auto issue = [&]() {                   // DMAs of the next step of the stream into its ring stage
            if (issued >= steps_total) return;
            float* stage = lds + (issued % RING_STAGES) * RING_STAGE_FLOATS;
#pragma unroll
            for (int i = 0; i < RING_PPW; ++i)
                dma_16B_to_lds(src[i] + lkt * BK, stage + (lw * RING_PPW + i) * 256);
            ++issued;
            if (++lkt == lKT) {
                lt = next_valid(lt + stride);
                lkt = 0;
                if (lt < total) load_tile(lt);
            }
        };
        auto wait_landed = [&](int g) {        // every DMA of steps <= g has landed (in-order completion)
            const int after = issued - (g + 1);    // steps issued later than g
            if (after >= 2) wait_vmcnt_le<2 * RING_PPW>();
            else if (after >= 1) wait_vmcnt_le<RING_PPW>();
            else wait_vmcnt_le<0>();
        };
#pragma unroll
        for (int i = 0; i < RING_AHEAD; ++i) issue();
        wait_landed(1);
        __builtin_amdgcn_s_barrier();          // B(-2): steps 0 and 1 are in their stages
        __builtin_amdgcn_s_barrier();          // B(-1): the consumers hold step 0's fragments
#ifdef MEL_RING_PROF
        unsigned long long li = 0, lw_ = 0, lb = 0;
#endif
        for (int g = 0; g < steps_total; ++g) {
#ifdef MEL_RING_PROF
            const unsigned long long t0 = RING_T();
#endif
            issue();                           // step g + 4 -> the stage of step g (in registers since B(g-1))
#ifdef MEL_RING_PROF
            const unsigned long long t1 = RING_T();
#endif
            if (g + 2 < steps_total) wait_landed(g + 2);
#ifdef MEL_RING_PROF
            const unsigned long long t2 = RING_T();
#endif
            __builtin_amdgcn_s_barrier();      // B(g)
#ifdef MEL_RING_PROF
            const unsigned long long t3 = RING_T();
            li += t1 - t0, lw_ += t2 - t1, lb += t3 - t2;
#endif
            if (++ckt == cc.KT) {              // step g completed a tile: its accumulators are in outbuf (written before B(g))
                write_out();                   // done before this wave reaches B(g+1); the consumers' next hand-over is >= 2 steps away
                ct = next_valid(ct + stride);
                ckt = 0;
                if (ct < total) cc = tile_of(ct);
            }
        }
#ifdef MEL_RING_PROF
        if (wid == NC && lane == 0 && TAG == MEL_RING_PROF) {
            atomicAdd(&g_ring_prof[5], li), atomicAdd(&g_ring_prof[6], lw_), atomicAdd(&g_ring_prof[7], lb);
        }
#endif
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int wm = wid >> 1, wn = wid & 1;
    const int r = lane & 31, h = lane >> 5;
    const int ar = wm * 32 + r, wr = BM + wn * 32 + r;
    const int a_row = ar * BK, a_x = (ar >> SWZ) & (CPR - 1);     // fragment rows of this lane (floats into a stage) + swizzle
    const int w_row = wr * BK, w_x = (wr >> SWZ) & (CPR - 1);
    constexpr int NQ = BK / 8;                                    // fragment pairs (four MFMAs each) per K step
    auto read_frags = [&](int g, f32x4 (&fa)[NQ], f32x4 (&fb)[NQ]) {
        const float* cur = lds + (g % RING_STAGES) * RING_STAGE_FLOATS;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            fa[q] = *reinterpret_cast<const f32x4*>(cur + a_row + (((2 * q + h) ^ a_x) << 2));
            fb[q] = *reinterpret_cast<const f32x4*>(cur + w_row + (((2 * q + h) ^ w_x) << 2));
        }
    };

    int t = first;
    Tile c = tile_of(t);
    int kt = 0;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#ifdef MEL_RING_PROF
    unsigned long long cm_ = 0, cb = 0, ce = 0;
    const unsigned long long tk0 = RING_T();
#endif
    f32x4 fa[2][NQ], fb[2][NQ];                // fragments of the step being multiplied / of the next one
    __builtin_amdgcn_s_barrier();              // B(-2)
    read_frags(0, fa[0], fb[0]);
    wait_lds_done();
    __builtin_amdgcn_s_barrier();              // B(-1)
    auto step = [&](int g, f32x4 (&ca)[NQ], f32x4 (&cbf)[NQ], f32x4 (&na)[NQ], f32x4 (&nb)[NQ]) {
#ifdef MEL_RING_PROF
        const unsigned long long t0 = RING_T();
#endif
        if (g + 1 < steps_total) read_frags(g + 1, na, nb);       // landed before B(g-1); overlaps the MFMAs below
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[q][kk], cbf[q][kk], acc, 0, 0, 0);
#ifdef MEL_RING_PROF
        asm volatile("s_nop 0" ::"v"(acc[0]));    // the MFMA chain has retired
        const unsigned long long t1 = RING_T();
#endif
        if (++kt == c.KT) {                    // tile complete: hand the accumulator block to the loaders
            // (C/D layout: col = lane & 31, row = (e & 3) + 8*(e >> 2) + 4*h)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                outbuf[ring_out_index(wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, wn * 32 + r)] = acc[e];
                acc[e] = 0.f;
            }
            t = next_valid(t + stride);
            kt = 0;
            if (t < total) c = tile_of(t);
        }
#ifdef MEL_RING_PROF
        const unsigned long long t2 = RING_T();
#endif
        wait_lds_done();                       // the next step's fragments are in registers, the hand-over is written
        __builtin_amdgcn_s_barrier();          // B(g)
#ifdef MEL_RING_PROF
        const unsigned long long t3 = RING_T();
        cm_ += t1 - t0, ce += t2 - t1, cb += t3 - t2;
#endif
    };
    int g = 0;
    for (; g + 1 < steps_total; g += 2) {
        step(g, fa[0], fb[0], fa[1], fb[1]);
        step(g + 1, fa[1], fb[1], fa[0], fb[0]);
    }
    if (g < steps_total) step(g, fa[0], fb[0], fa[1], fb[1]);
#ifdef MEL_RING_PROF
    if (wid == 0 && lane == 0 && TAG == MEL_RING_PROF) {
        atomicAdd(&g_ring_prof[0], cm_), atomicAdd(&g_ring_prof[1], cb), atomicAdd(&g_ring_prof[2], ce);
        atomicAdd(&g_ring_prof[3], RING_T() - tk0), atomicAdd(&g_ring_prof[4], 1ull);
    }
#endif
}

// split-K finish: Y = act(scale * (P_0 + P_1 + ... in plane order) + bias) over the row range the launch covered; the
// fixed summation order keeps the result independent of how the work items were scheduled.
struct SplitKFinish {
    const float* parts;
    long part_stride;
    int S, N, M;
    const int32_t* M_dev;
    const float* bias;
    const float* bias_hi;
    int split_n;
    const float* rscale;
    int relu;
    float* Y;
    int ldy;
};
__global__ __launch_bounds__(256) void splitk_finish_kernel(SplitKFinish f) {
    const int rows = f.M_dev ? min(*f.M_dev, f.M) : f.M;
    const int n4 = f.N >> 2;
    const long total = (long)rows * n4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / n4), n = (int)(i % n4) * 4;
        const float* p = f.parts + (size_t)m * f.N + n;
        f32x4 acc = *reinterpret_cast<const f32x4*>(p);
        for (int s = 1; s < f.S; ++s) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(p + (size_t)s * f.part_stride);
            acc = f32x4{acc[0] + q[0], acc[1] + q[1], acc[2] + q[2], acc[3] + q[3]};
        }
        const float sc = f.rscale ? f.rscale[m] : 1.f;
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (f.bias_hi && n >= f.split_n) b = *reinterpret_cast<const f32x4*>(f.bias_hi + (n - f.split_n));
        else if (f.bias) b = *reinterpret_cast<const f32x4*>(f.bias + n);
        f32x4 o = {acc[0] * sc + b[0], acc[1] * sc + b[1], acc[2] * sc + b[2], acc[3] * sc + b[3]};
        if (f.relu) o = f32x4{fmaxf(o[0], 0.f), fmaxf(o[1], 0.f), fmaxf(o[2], 0.f), fmaxf(o[3], 0.f)};
        *reinterpret_cast<f32x4*>(f.Y + (size_t)m * f.ldy + n) = o;
    }
}

}  // namespace mel
