// NOT COMPILED INTO THE LIBRARY - a starting point for the next round (end of round 2).
//
// gemm_split_big_kernel (csrc/gemm_split.hpp) with specialised wavefronts, as it stood when the round's GPU budget ran out.
// tools/split_shapes_probe.hip measures the STEP of this structure at 273-287 TF of fp32-accurate FLOPs against 144-164 for the
// one-role step (the library's kernel: 160-177 TF).  This kernel, dropped into csrc/gemm_split.hpp behind
// gemm_launch_split_big_t (512 threads, grid 256 x 2), is CORRECT (tests/test_gpu_bf16.py: split GEMM against float64, both
// precisions' forwards) but SLOWER than the one-role kernel as written: M = 65 536, N = K = 512: 149 TF against 177; conv2 in
// the step 69 us against 52.6.  Known differences from the probe that have not been taken apart yet: the 128-VGPR cap of
// four waves per SIMD (124 bytes of scratch, all in the per-tile code: epilogue and tile set-up), the epilogue running in
// the MFMA waves while the loaders sit at the step barrier, the tile-crossing set-up inside the loaders' stream.  First
// things to try: the epilogue handed to the loader waves through LDS (as gemm_ring.hpp does), cycle stamps per role
// (tools/split_prof.py's scheme).  Already tried, on the last GPU minutes of the round: launch_bounds(512, 2) with ONE
// workgroup per CU (174 VGPRs, no scratch; grid 256): 125 TF at M = 65 536, and the same with the loads FOUR steps ahead
// of their fill (four register sets, the version below): 134 TF, conv2's lin_l shape 61 us - neither the register cap nor
// the prefetch distance is what separates this kernel from the probe's 255-287 TF.  Nor is HBM streaming: with a NEW A panel
// per 32 steps out of 2 GB (every A byte from HBM once - harsher than conv2, where four column tiles share a panel) the
// probe's roles step still runs at 209 TF.  What the probe does not have is the EPILOGUE: ~10 000 cycles per tile in the MFMA
// waves (tools/split_prof.py on the one-role kernel) against 32 steps x ~850 cycles - 209 / 1.37 = 152 TF, which is what this
// kernel measures.  So: the epilogue out of the MFMA waves (accumulators dropped into an LDS hand-over buffer and stored by
// the loader waves, gemm_ring.hpp's scheme) or overlapped with the next tile's first steps (a second accumulator set).
//
// ---- 128 x 128 tiles, specialised wavefronts -------------------------------------------------------------------------
// gemm_split_big_kernel with its work dealt to two kinds of wavefront: a 512-thread workgroup whose waves 0-3 (2 x 2, one
// 64 x 64 block set each) only read fragments and multiply, and whose waves 4-7 only load, split and fill - the staging of
// the kernel above, thread for thread.  One barrier per K step hands a stage over in each direction.  Why: a global load
// that has to wait at the CU's address path blocks the instruction stream it sits in; in the one-role kernel that stream
// also carries the MFMAs.  Measured on the step alone (tools/split_shapes_probe.hip, fp32-accurate TF): fragment reads +
// MFMAs 303, + barrier 299, + split and fill 260, + the step's five loads 144-164 (189 cache resident) - and 273-287 with
// the loads in loader waves.  <= 128 VGPRs: two workgroups (16 waves) per CU.
template <int TAG = 0>
__global__ __launch_bounds__(512, 2) void gemm_split_roles_kernel(GemmBatch batch) {
    constexpr int BM = 128, BN = 128;
    constexpr int BUF = (BM + BN) * GEMS2_ROW;        // 16-byte chunks per LDS stage
    __shared__ u32x4 lds[2 * BUF];

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
    const int role = threadIdx.x >> 8;    // 0: MFMA waves, 1: loader waves
    const int tid = threadIdx.x & 255;
    const int lane = tid & 63;
    const int wid = tid >> 6;

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
    if (t0 >= total) return;
    int nsteps = 0;                           // K steps of this workgroup's whole stream
    for (int tt = t0; tt < total; tt = next_valid(tt + stride)) nsteps += meta_of(tt).KT;

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
        int stage = 0;
        __syncthreads();                      // stage 0 holds step 0
        for (int t = t0; t < total; t = next_valid(t + stride)) {
            const Meta cm = meta_of(t);
            for (int kt = 0; kt < cm.KT; ++kt) {
                const u32x4* cst = lds + stage * BUF;
                bf16x8 a[2][3], b[2][3];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        a[u][p] = __builtin_bit_cast(bf16x8, cst[a_off + u * 32 * GEMS2_ROW + 2 * p]);
                        b[u][p] = __builtin_bit_cast(bf16x8, cst[w_off + u * 32 * GEMS2_ROW + 2 * p]);
                    }
                // per block smallest products first (mid*mid, hi*lo, lo*hi, hi*mid, mid*hi, hi*hi), the four blocks interleaved
                constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
                for (int k = 0; k < 6; ++k)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[k]], b[j][PB[k]], acc[i][j], 0, 0, 0);
                __syncthreads();
                stage ^= 1;
            }
            const GemmArgs& g = batch.p[cm.pi];
            if (g.ksplit > 1) {                // raw partial products into this chunk's fp32 plane
                float* P = g.Y + (size_t)cm.ks * g.part_stride;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int n = cm.n0 + wn * 64 + j * 32 + r;
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int m = cm.m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                            if (m < cm.M) P[(size_t)m * g.ldy + n] = acc[i][j][e];
                        }
                    }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        store_block_f32(g, acc[i][j], cm.m0 + wm * 64 + i * 32 + 4 * h, cm.n0 + wn * 64 + j * 32 + r, cm.M);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        }
        for (int pad = (4 - (nsteps & 3)) & 3; pad > 0; --pad) __syncthreads();      // the loaders' padding steps
        return;
    }

    // ---- loader waves: the staging of gemm_split_big_kernel ---------------------------------------------------------------
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
    Ctx pf;                                   // where the prefetch stands
    int pf_t = t0, pf_kt = 0;
    bool pf_valid = true;
    setup(pf, t0);
    auto issue = [&](Regs& R) {               // loads of the next step of the stream, unconditional (see gemm_split_kernel)
#pragma unroll
        for (int i = 0; i < 2; ++i) R.a[i] = *reinterpret_cast<const f32x4*>(pf.a_src[i] + pf_kt * GEMS2_BK);
#pragma unroll
        for (int i = 0; i < 3; ++i) R.w[i] = pf.w_src[i][pf_kt * 6];
        if (pf_valid && ++pf_kt == pf.KT) {   // cross into this workgroup's next work item
            const int tn = next_valid(pf_t + stride);
            if (tn < total) {
                setup(pf, tn);
                pf_t = tn, pf_kt = 0;
            } else {
                pf_valid = false, pf_kt = pf.KT - 1;
            }
        }
    };
    auto fill_stage = [&](int stage, const Regs& R) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u32x2 hi, mid, lo;
            split4(R.a[i], hi, mid, lo);
            u32x2* dst = lds8 + stage * (2 * BUF) + a_st + i * 64 * (2 * GEMS2_ROW);
            dst[0] = hi, dst[4] = mid, dst[8] = lo;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) lds[stage * BUF + w_st[i]] = R.w[i];
    };
    // FOUR register sets: a step of this kernel is short, so the loads run four steps ahead of their fill
    Regs R0, R1, R2, R3;
    issue(R0);                                 // step 0
    issue(R1);                                 // step 1
    issue(R2);                                 // step 2
    issue(R3);                                 // step 3
    fill_stage(0, R0);
    __syncthreads();                           // stage 0 is ready
    issue(R0);                                 // step 4
    int stage = 0;
    // during step s (the MFMA waves are on `stage`): Ra (step s+1) -> the other stage, then Ra <- loads of step s+5
    auto step = [&](Regs& Ra) {
        fill_stage(stage ^ 1, Ra);
        issue(Ra);
        __syncthreads();
        stage ^= 1;
    };
    for (int it = 0; it < (nsteps + 3) >> 2; ++it) {       // counted loop over quads of steps (see gemm_split_kernel)
        step(R1);
        step(R2);
        step(R3);
        step(R0);
    }
}

