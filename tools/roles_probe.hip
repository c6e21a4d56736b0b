// Standalone driver for the 128 x 128 split-bf16 GEMM kernels (csrc/gemm_split.hpp): the one-role kernel against the
// specialised-wavefront kernel on the conv2 launch of the L-DGN step (two grouped problems, the second with a row gather) and
// on one large problem, interleaved rounds in one process, results compared element for element; with -DMEL_ROLES_PROF also
// the specialised-wavefront kernel's in-kernel cycle breakdown.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off [-DMEL_ROLES_PROF] tools/roles_probe.hip -o tools/bin/roles_probe
#include "../melissa_amd/csrc/gemm_split.hpp"      // + the kernels of tools/experiments/gemm_split_variants.hpp pasted back in (see its header)
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>

namespace mel {
void set_error(const char*, ...) {}
Profiler* current_profiler() { return nullptr; }
}
using namespace mel;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static float* dev_random(size_t n, float scale, unsigned seed) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = scale * ((float)(s >> 8) / 8388608.0f - 1.0f);
    }
    float* d;
    CK(hipMalloc(&d, n * 4));
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}

struct Problem {
    int M, N, K;
    float *A, *W, *bias, *Y[3];
    uint16_t* planes;
    int32_t* arow;
};

static Problem make_problem(int M, int N, int K, int a_rows, bool gather, unsigned seed) {
    Problem p{};
    p.M = M, p.N = N, p.K = K;
    p.A = dev_random((size_t)a_rows * K, 1.0f, seed);
    p.W = dev_random((size_t)N * K, 1.0f / sqrtf((float)K), seed + 1);
    p.bias = dev_random(N, 1.0f, seed + 2);
    for (int i = 0; i < 3; ++i) CK(hipMalloc(&p.Y[i], (size_t)M * N * 4));
    CK(hipMalloc(&p.planes, (size_t)3 * N * K * 2));
    SplitBatch b{};
    b.n = 1, b.src[0] = p.W, b.dst[0] = p.planes, b.count[0] = N * K, b.K[0] = K, b.start[0] = 0;
    b.start[1] = (int)(((size_t)N * K / 4 + 255) / 256);
    hipLaunchKernelGGL(split_weights_kernel, dim3(b.start[1]), dim3(256), 0, 0, b);
    if (gather) {
        std::vector<int32_t> idx(a_rows);
        for (int i = 0; i < a_rows; ++i) idx[i] = i;
        unsigned s = seed;
        for (int i = a_rows - 1; i > 0; --i) { s = s * 1664525u + 1013904223u; std::swap(idx[i], idx[(s >> 8) % (i + 1)]); }
        idx.resize(M);
        std::sort(idx.begin(), idx.end());
        CK(hipMalloc(&p.arow, M * 4));
        CK(hipMemcpy(p.arow, idx.data(), M * 4, hipMemcpyHostToDevice));
    }
    return p;
}

static GemmBatch make_batch(const std::vector<Problem>& ps, int which, int ksplit, float* parts) {
    GemmBatch b{};
    b.count = (int)ps.size();
    for (int i = 0; i < b.count; ++i) {
        GemmArgs g;
        g.A = ps[i].A, g.lda = ps[i].K, g.arow = ps[i].arow, g.W = reinterpret_cast<const float*>(ps[i].planes), g.bias = ps[i].bias;
        g.Y = ps[i].Y[which], g.ldy = ps[i].N, g.M = ps[i].M, g.N = ps[i].N, g.K = ps[i].K, g.split = 1, g.relu = 0;
        if (ksplit > 1) g.ksplit = ksplit, g.part_stride = (long)ps[i].M * ps[i].N, g.Y = parts + (size_t)which * ksplit * ps[i].M * ps[i].N;
        b.p[i] = g;
    }
    return b;
}

static long items_of(const std::vector<Problem>& ps, int ksplit) {
    long items = 0;
    for (auto& p : ps) items += ((long)((p.M + 127) / 128) * (p.N / 128) * (ksplit > 1 ? ksplit : 1) + 7) & ~7L;
    return items;
}

static void run_case(const char* name, std::vector<Problem> ps, int ksplit, int rounds) {
    float* parts = nullptr;
    if (ksplit > 1) CK(hipMalloc(&parts, (size_t)3 * ksplit * ps[0].M * ps[0].N * 4));
    const GemmBatch b0 = make_batch(ps, 0, ksplit, parts), b1 = make_batch(ps, 1, ksplit, parts), b2 = make_batch(ps, 2, ksplit, parts);
    const long items = items_of(ps, ksplit);
    long witems = 0;
    bool wide_ok = true;
    for (auto& p : ps) witems += ((long)((p.M + 127) / 128) * (p.N / 256) * (ksplit > 1 ? ksplit : 1) + 7) & ~7L, wide_ok = wide_ok && p.N % 256 == 0;
    const int grid0 = (int)std::min(512L, items), grid1 = (int)std::min(256L, items), grid2 = (int)std::min(256L, witems);
    auto one = [&]() { hipLaunchKernelGGL((gemm_split_big_kernel<0>), dim3(grid0), dim3(256), 0, 0, b0); };
#ifdef PROBE_WIDE
    auto roles = [&]() { if (wide_ok) hipLaunchKernelGGL((gemm_split_wide_kernel<0>), dim3(grid2), dim3(512), 0, 0, b1); };
#elif defined(PROBE_WIDE2)
    auto roles = [&]() { hipLaunchKernelGGL((gemm_split_wide_kernel<0, 2>), dim3(grid0), dim3(256), 0, 0, b1); };
    (void)grid2, (void)b2, (void)grid1;
#elif defined(PROBE_DENSE)
    const int grid3 = (int)std::min(768L, items);
    auto roles = [&]() { hipLaunchKernelGGL((gemm_split_wide_kernel<0, 2, true>), dim3(grid3), dim3(256), 0, 0, b1); };
    (void)grid2, (void)b2, (void)grid1;
#else
    auto roles = [&]() { hipLaunchKernelGGL((gemm_split_roles_kernel<0>), dim3(grid1), dim3(768), 0, 0, b1); };
    (void)grid2, (void)b2;
#endif
    one(), roles();
    CK(hipDeviceSynchronize());
    // compare
    double maxdiff = 0;
    for (auto& p : ps) {
        const size_t n = (size_t)p.M * p.N * (ksplit > 1 ? ksplit : 1);
        std::vector<float> h0(n), h1(n);
        const float* y0 = ksplit > 1 ? parts : p.Y[0];
        const float* y1 = ksplit > 1 ? parts + (size_t)ksplit * p.M * p.N : p.Y[1];
        CK(hipMemcpy(h0.data(), y0, n * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h1.data(), y1, n * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) maxdiff = std::max(maxdiff, (double)fabsf(h0[i] - h1[i]));
    }
#ifdef MEL_ROLES_PROF
    unsigned long long z[16] = {0};
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_roles_prof), z, sizeof(z)));
#endif
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t0s, t1s;
    for (int r = 0; r < rounds; ++r) {
        float ms;
        CK(hipEventRecord(e0)); for (int i = 0; i < 4; ++i) one(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); t0s.push_back(ms / 4 * 1e3f);
        CK(hipEventRecord(e0)); for (int i = 0; i < 4; ++i) roles(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); t1s.push_back(ms / 4 * 1e3f);
    }
    std::sort(t0s.begin(), t0s.end()), std::sort(t1s.begin(), t1s.end());
    double flop = 0;
    for (auto& p : ps) flop += 2.0 * p.M * p.N * p.K;
    const float m0 = t0s[t0s.size() / 2], m1 = t1s[t1s.size() / 2];
    printf("%-28s %6.2f GF  items %4ld | one-role %7.1f us %6.1f TF (min %6.1f) | " 
#ifdef PROBE_WIDE
           "128x256"
#else
           "roles"
#endif
           " %7.1f us %6.1f TF (min %6.1f) | max |diff| %.1e\n", name,
           flop / 1e9, items, m0, flop / m0 / 1e6, t0s[0], m1, flop / m1 / 1e6, t1s[0], maxdiff);
#ifdef MEL_ROLES_PROF
    unsigned long long v[16];
    CK(hipMemcpyFromSymbol(v, HIP_SYMBOL(g_roles_prof), sizeof(v)));
    const double steps = (double)std::max(v[9], 1ull), wgs = (double)std::max(v[8], 1ull);
    printf("    per step, MFMA wave 0: reads+MFMAs %.0f, epilogue %.0f, lds wait + barrier %.0f | loader wave 0 of team 0, per step of ITS (every other): wait+split+fill %.0f, "
           "barrier %.0f, issue %.0f, lds wait + barrier %.0f | kernel %.0f cycles per workgroup, %.1f steps per workgroup\n",
           v[0] / steps, v[1] / steps, v[2] / steps, 2 * v[3] / steps, 2 * v[4] / steps, 2 * v[5] / steps, 2 * v[6] / steps, v[7] / wgs, steps / wgs);
#endif
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 15;
    run_case("conv2 (lin_l + gathered lin_r)", {make_problem(10653, 512, 512, 10653, false, 1), make_problem(4820, 512, 512, 10653, true, 5)}, 0, rounds);
    run_case("conv2 lin_l alone", {make_problem(10653, 512, 512, 10653, false, 11)}, 0, rounds);
    run_case("heads' first layer, split-K 3", {make_problem(4820, 256, 1152, 4820, false, 21)}, 3, rounds);
    run_case("big 65536 x 512 x 512", {make_problem(65536, 512, 512, 65536, false, 31)}, 0, rounds);
    return 0;
}
