// gemm_planes_kernel (csrc/gemm_split.hpp: 128 x 256 tiles, both operands as bf16 planes, specialised wavefronts) against the
// shipped gemm_split_big_kernel on conv2's launch: results compared element for element, interleaved timing.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off tools/planes_probe.hip -o tools/bin/planes_probe
#include "../melissa_amd/csrc/gemm_split.hpp"
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
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((float)(s >> 8) / 8388608.0f - 1.0f); }
    float* d;
    CK(hipMalloc(&d, n * 4));
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}
static uint16_t* planes_of(const float* src, int rows, int K) {
    uint16_t* d;
    CK(hipMalloc(&d, (size_t)3 * rows * K * 2));
    SplitBatch b{};
    b.n = 1, b.src[0] = src, b.dst[0] = d, b.count[0] = rows * K, b.K[0] = K, b.start[0] = 0;
    b.start[1] = (int)(((size_t)rows * K / 4 + 255) / 256);
    hipLaunchKernelGGL(split_weights_kernel, dim3(b.start[1]), dim3(256), 0, 0, b);
    return d;
}
static uint16_t* blocks_of(const uint16_t* planes, int rows, int K, int RB) {
    uint16_t* d;
    const size_t chunks = (size_t)((rows + RB - 1) / RB) * (K / 16) * RB * 6;
    CK(hipMalloc(&d, chunks * 16));
    hipLaunchKernelGGL(planes_to_blocks_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, 0, reinterpret_cast<const u32x4*>(planes),
                       reinterpret_cast<u32x4*>(d), rows, K, RB);
    return d;
}
struct Problem { int M, N, K, a_rows; float *A, *W, *bias, *Y[2]; uint16_t *Ap, *Wp, *Ab, *Wb; int32_t* arow; };
static Problem make_problem(int M, int N, int K, int a_rows, bool gather, unsigned seed) {
    Problem p{};
    p.M = M, p.N = N, p.K = K, p.a_rows = a_rows;
    p.A = dev_random((size_t)a_rows * K, 1.0f, seed);
    p.W = dev_random((size_t)N * K, 1.0f / sqrtf((float)K), seed + 1);
    p.bias = dev_random(N, 1.0f, seed + 2);
    for (int i = 0; i < 2; ++i) CK(hipMalloc(&p.Y[i], (size_t)M * N * 4));
    p.Ap = planes_of(p.A, a_rows, K), p.Wp = planes_of(p.W, N, K);
    p.Ab = blocks_of(p.Ap, a_rows, K, 128), p.Wb = blocks_of(p.Wp, N, K, 256);
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
static void run_case(const char* name, std::vector<Problem> ps, int rounds) {
    GemmBatch b0{}, b1{};
    b0.count = b1.count = (int)ps.size();
    long items0 = 0, items1 = 0;
    double flop = 0;
    for (int i = 0; i < b0.count; ++i) {
        GemmArgs g;
        g.A = ps[i].A, g.lda = ps[i].K, g.arow = ps[i].arow, g.W = reinterpret_cast<const float*>(ps[i].Wp), g.bias = ps[i].bias;
        g.Y = ps[i].Y[0], g.ldy = ps[i].N, g.M = ps[i].M, g.N = ps[i].N, g.K = ps[i].K, g.split = 1, g.relu = 1;
        b0.p[i] = g;
        g.A = reinterpret_cast<const float*>(ps[i].Ap);
        g.W = reinterpret_cast<const float*>(ps[i].Wb), g.lda = 3 * ps[i].K, g.Y = ps[i].Y[1];
        b1.p[i] = g;
        items0 += ((long)((ps[i].M + 127) / 128) * (ps[i].N / 128) + 7) & ~7L;
        items1 += ((long)((ps[i].M + 127) / 128) * (ps[i].N / 256) + 7) & ~7L;
        flop += 2.0 * ps[i].M * ps[i].N * ps[i].K;
    }
    const int grid0 = (int)std::min(512L, items0), grid1 = (int)std::min(256L, items1);
    auto one = [&]() { hipLaunchKernelGGL((gemm_split_big_kernel<0>), dim3(grid0), dim3(256), 0, 0, b0); };
    auto pl = [&]() { hipLaunchKernelGGL((gemm_planes_kernel<0>), dim3(grid1), dim3(768), 0, 0, b1); };
    one(), pl();
    CK(hipDeviceSynchronize());
    double maxdiff = 0;
    for (auto& p : ps) {
        const size_t n = (size_t)p.M * p.N;
        std::vector<float> h0(n), h1(n);
        CK(hipMemcpy(h0.data(), p.Y[0], n * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h1.data(), p.Y[1], n * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) maxdiff = std::max(maxdiff, (double)fabsf(h0[i] - h1[i]));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t0s, t1s;
    for (int r = 0; r < rounds; ++r) {
        float ms;
        CK(hipEventRecord(e0)); for (int i = 0; i < 4; ++i) one(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); t0s.push_back(ms / 4 * 1e3f);
        CK(hipEventRecord(e0)); for (int i = 0; i < 4; ++i) pl(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); t1s.push_back(ms / 4 * 1e3f);
    }
#ifdef MEL_PLANES_STAMPS
    {
        long long z[32] = {0}, st[32];
        CK(hipMemcpyToSymbol(HIP_SYMBOL(planes_stamps), z, sizeof z));
        pl();
        CK(hipDeviceSynchronize());
        CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(planes_stamps), sizeof st));
        auto per = [&](int i, int n) { return st[n] ? (double)st[i] / st[n] : 0.0; };
        printf("   block 0: %lld clock64 ticks in %lld wall ticks (100 MHz): clock64 runs at %.0f MHz; %.2f us inside the kernel\n", st[24], st[25],
               st[25] ? 100.0 * st[24] / st[25] : 0.0, st[25] / 100.0);
        {
            static long long pb[512][2];
            CK(hipMemcpyFromSymbol(pb, HIP_SYMBOL(planes_block), sizeof pb));
            long long s0 = pb[0][0], s1 = pb[0][0], e0 = pb[0][1], e1 = pb[0][1];
            double dur = 0;
            for (int i = 0; i < grid1; ++i) {
                s0 = std::min(s0, pb[i][0]), s1 = std::max(s1, pb[i][0]), e0 = std::min(e0, pb[i][1]), e1 = std::max(e1, pb[i][1]);
                dur += (pb[i][1] - pb[i][0]) / 100.0;
            }
            int late = 0;
            for (int i = 0; i < grid1; ++i) late += pb[i][0] - s0 > 300;
            printf("   %d blocks: starts spread %.2f us (%d start > 3 us after the first), ends %.2f .. %.2f us after the first start, mean in-block %.2f us\n",
                   grid1, (s1 - s0) / 100.0, late, (e0 - s0) / 100.0, (e1 - s0) / 100.0, dur / grid1);
        }
        printf("   block 0: first barrier passed at %lld ticks, step loop left at %lld, end %lld\n", st[26], st[27], st[24]);
    }
#endif
    std::sort(t0s.begin(), t0s.end()), std::sort(t1s.begin(), t1s.end());
    const float m0 = t0s[t0s.size() / 2], m1 = t1s[t1s.size() / 2];
    printf("%-30s %6.2f GF | one-role 128x128 (%4ld items) %7.1f us %6.1f TF (min %6.1f) | planes 128x256 (%4ld items) %7.1f us %6.1f TF (min %6.1f) | max |diff| %.1e\n",
           name, flop / 1e9, items0, m0, flop / m0 / 1e6, t0s[0], items1, m1, flop / m1 / 1e6, t1s[0], maxdiff);
    fflush(stdout);
}
int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 15;
    run_case("conv2 (lin_l + gathered lin_r)", {make_problem(10653, 512, 512, 10653, false, 1), make_problem(4820, 512, 512, 10653, true, 5)}, rounds);
    run_case("conv2 lin_l alone", {make_problem(10653, 512, 512, 10653, false, 11)}, rounds);
    run_case("ragged: 1000 rows", {make_problem(1000, 512, 512, 1000, false, 21)}, rounds);
    run_case("big 65536 x 512 x 512", {make_problem(65536, 512, 512, 65536, false, 31)}, rounds);
    run_case("dgn-r conv2 (1024 + 512 cols)", {make_problem(10653, 1024, 512, 10653, false, 41), make_problem(4820, 512, 512, 10653, true, 45)}, rounds);
    return 0;
}
